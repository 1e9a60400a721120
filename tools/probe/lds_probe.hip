// LDS micro-probe for gfx950 (lean inner loops: 2-3 VALU ops per LDS op so that the LDS pipe is what is measured).
// Prints clocks (at 2.4 GHz) per wave-level LDS instruction group per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// all index spaces are powers of two; idx advances by an LCG step (mul+add+and)
// MODE 0: random b64 read over TABN doubles
// MODE 1: clustered b64 read: loc shared by groups of 8 lanes (changes per iteration), row random in 16, row stride RS
// MODE 2: u64 atomic add, random over ACCN
// MODE 3: u64 atomic add, groups of 4 adjacent lanes hit the same address
// MODE 4: 2 clustered reads + 2 atomics (engine-3 pattern)
// MODE 5: 1 b16-ish read + 2 random b64 reads (current pattern)
// MODE 6: u64 atomic add random, returnless, 2 per iteration to adjacent slots (16 B apart? same slot pair)
template <int MODE, int RS>
__global__ __launch_bounds__(1024) void probe(int iters, uint32_t tab_mask, uint32_t acc_mask, unsigned long long *out)
{
    extern __shared__ unsigned long long lds[];
    const uint32_t tab_n = tab_mask + 1, acc_n = acc_mask + 1;
    unsigned long long *tab = lds, *acc = lds + 16 * RS;
    for (uint32_t i = threadIdx.x; i < 16 * RS + acc_n; i += blockDim.x) lds[i] = i;
    __syncthreads();
    uint32_t h = mix(threadIdx.x + blockIdx.x * 1024u + 12345u);
    const uint32_t lane = threadIdx.x & 63;
    uint32_t g = mix(blockIdx.x * 64u + lane / 8u);  // shared by 8 adjacent lanes
    unsigned long long s = 0;
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        if (MODE == 0) { s += tab[(h >> 8) & tab_mask]; }
        else if (MODE == 1) {
            g = g * 1664525u + 1013904223u;
            s += tab[((h >> 12) & 15u) * RS + ((g >> 10) & 255u)];
        } else if (MODE == 2) { atomicAdd(&acc[(h >> 8) & acc_mask], 1ull); }
        else if (MODE == 3) {
            g = g * 1664525u + 1013904223u;
            atomicAdd(&acc[((g >> 8) + (lane >> 2)) & acc_mask], 1ull);
        } else if (MODE == 4) {
            g = g * 1664525u + 1013904223u;
            const uint32_t loc = (g >> 10) & 255u, row = (h >> 12) & 15u;
            const unsigned long long v = tab[row * RS + loc], w = tab[(row & 3u) * RS + loc + 1];
            const uint32_t a = (h >> 16) & (acc_mask >> 1);
            atomicAdd(&acc[2 * a], v);
            atomicAdd(&acc[2 * a + 1], w);
        } else if (MODE == 5) {
            const uint32_t e = ((const unsigned short *)tab)[(threadIdx.x * 5u + it) & 4095u];
            s += tab[((h >> 8) + e) & tab_mask];
            s += tab[((h >> 10) + e) & tab_mask];
        } else if (MODE == 7) {  // random b128 read
            const ulonglong2 v = ((const ulonglong2 *)tab)[(h >> 8) & (tab_mask >> 1)];
            s += v.x ^ v.y;
        } else if (MODE == 8) {  // u16 read + random b128 read
            const uint32_t e = ((const unsigned short *)tab)[(threadIdx.x * 5u + it) & 4095u];
            const ulonglong2 v = ((const ulonglong2 *)tab)[((h >> 8) + e) & (tab_mask >> 1)];
            s += v.x ^ v.y;
        } else if (MODE == 9) {  // u16 read alone
            s += ((const unsigned short *)tab)[(threadIdx.x * 5u + it + (h & 1u)) & 4095u];
        } else if (MODE == 10) {  // u16 read + 1 random b64
            const uint32_t e = ((const unsigned short *)tab)[(threadIdx.x * 5u + it) & 4095u];
            s += tab[((h >> 8) + e) & tab_mask];
        } else if (MODE == 6) {
            const uint32_t a = (h >> 16) & (acc_mask >> 1);
            atomicAdd(&acc[2 * a], 3ull);
            atomicAdd(&acc[2 * a + 1], 5ull);
        }
    }
    __syncthreads();
    if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6) for (uint32_t i = threadIdx.x; i < acc_n; i += blockDim.x) s += acc[i];
    out[blockIdx.x * 1024u + threadIdx.x] = s + tab_n;
}

template <int MODE, int RS> int run(const char *name, int iters, uint32_t tab_n, uint32_t acc_n, int wg_per_cu)
{
    unsigned long long *out;
    const int ncu = 256, grid = ncu * wg_per_cu;
    CHECK(hipMalloc(&out, (size_t)grid * 1024 * 8));
    const size_t lds = (size_t)(16 * RS + acc_n) * 8;
    CHECK(hipFuncSetAttribute((const void *)probe<MODE, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<MODE, RS>), dim3(grid), dim3(1024), lds, 0, iters, tab_n - 1, acc_n - 1, out);
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((probe<MODE, RS>), dim3(grid), dim3(1024), lds, 0, iters, tab_n - 1, acc_n - 1, out);
    CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double clk = 2.4e9 * ms * 1e-3, per = clk / ((double)wg_per_cu * 16 * iters);
    printf("%-52s lds=%4zu KB wg/cu=%d %7.3f ms -> %5.1f clk per wave-iteration per CU\n", name, lds / 1024, wg_per_cu, ms, per);
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    const int it = 20000;
    run<0, 256>("random ds_read_b64 over 4096 doubles", it, 4096, 0, 2);
    run<1, 256>("clustered read, 8 lanes/locus, row stride 256", it, 4096, 0, 2);
    run<1, 257>("clustered read, 8 lanes/locus, row stride 257", it, 4096, 0, 2);
    run<2, 256>("ds_add_u64 random over 1024 acc", it, 4096, 1024, 2);
    run<2, 256>("ds_add_u64 random over 4096 acc", it, 4096, 4096, 2);
    run<3, 256>("ds_add_u64, 4 adjacent lanes same address", it, 4096, 2048, 2);
    run<6, 256>("2 x ds_add_u64 random pair (4096 acc)", it, 4096, 4096, 2);
    run<4, 257>("2 clustered reads (stride 257) + 2 atomics", it, 4096, 4096, 2);
    run<4, 257>("same, 1 wg/cu", it, 4096, 4096, 1);
    run<5, 256>("u16 read + 2 random b64 reads (current pattern)", it, 4096, 0, 2);
    run<7, 256>("random ds_read_b128", it, 4096, 0, 2);
    run<7, 256>("random ds_read_b128, 1 wg/cu", it, 4096, 0, 1);
    run<8, 256>("u16 read + random b128", it, 4096, 0, 2);
    run<8, 256>("u16 read + random b128, 1 wg/cu", it, 4096, 0, 1);
    run<9, 256>("u16 read alone", it, 4096, 0, 2);
    run<10, 256>("u16 read + 1 random b64", it, 4096, 0, 2);
    run<5, 256>("u16 read + 2 random b64, 1 wg/cu", it, 4096, 0, 1);
    return 0;
}
