// how long do large hipMalloc / hipFree (and the stream-ordered pool) take on this box?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(0);
    const size_t GB = 1ull << 30;
    for (size_t sz : {1 * GB, 8 * GB, 16 * GB, 32 * GB}) {
        void *p = nullptr;
        double t0 = now();
        if (hipMalloc(&p, sz) != hipSuccess) { printf("hipMalloc %zu GB failed\n", sz / GB); continue; }
        double t1 = now();
        hipMemset(p, 1, sz); hipDeviceSynchronize();
        double t2 = now();
        hipFree(p);
        double t3 = now();
        printf("hipMalloc %2zu GB: alloc %.3f s, memset %.3f s, free %.3f s\n", sz / GB, t1 - t0, t2 - t1, t3 - t2);
    }
    hipMemPool_t pool;
    hipDeviceGetDefaultMemPool(&pool, 0);
    uint64_t thr = UINT64_MAX;
    hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    for (int rep = 0; rep < 3; rep++) {
        void *p = nullptr;
        double t0 = now();
        if (hipMallocAsync(&p, 16 * GB, 0) != hipSuccess) { printf("hipMallocAsync failed\n"); break; }
        hipStreamSynchronize(0);
        double t1 = now();
        hipFreeAsync(p, 0);
        hipStreamSynchronize(0);
        double t2 = now();
        printf("hipMallocAsync 16 GB (rep %d): alloc %.3f s, free %.3f s\n", rep, t1 - t0, t2 - t1);
    }
    // mixed: stream-ordered allocation, plain hipFree; and a growing footprint like the ingest's (5 x 16 GB live)
    for (int rep = 0; rep < 3; rep++) {
        void *p = nullptr;
        double t0 = now();
        if (hipMallocAsync(&p, 16 * GB, 0) != hipSuccess) { printf("hipMallocAsync failed\n"); break; }
        hipStreamSynchronize(0);
        double t1 = now();
        hipError_t e = hipFree(p);
        double t2 = now();
        printf("hipMallocAsync 16 GB + hipFree (rep %d): alloc %.3f s, free %.3f s (%s)\n", rep, t1 - t0, t2 - t1, hipGetErrorString(e));
    }
    {
        void *q[8];
        double t0 = now();
        for (int i = 0; i < 8; i++) if (hipMalloc(&q[i], 16 * GB) != hipSuccess) printf("hipMalloc live %d failed\n", i);
        double t1 = now();
        for (int i = 0; i < 8; i++) hipFree(q[i]);
        double t2 = now();
        for (int i = 0; i < 8; i++) if (hipMalloc(&q[i], 16 * GB) != hipSuccess) printf("hipMalloc live %d failed\n", i);
        double t3 = now();
        for (int i = 0; i < 8; i++) hipFree(q[i]);
        printf("8 x hipMalloc 16 GB live: %.3f s, free all %.3f s, again %.3f s\n", t1 - t0, t2 - t1, t3 - t2);
        t0 = now();
        for (int i = 0; i < 8; i++) { if (hipMallocAsync(&q[i], 16 * GB, 0) != hipSuccess) printf("async live %d failed\n", i); }
        hipStreamSynchronize(0);
        t1 = now();
        for (int i = 0; i < 8; i++) hipFreeAsync(q[i], 0);
        hipStreamSynchronize(0);
        t2 = now();
        for (int i = 0; i < 8; i++) { if (hipMallocAsync(&q[i], 16 * GB, 0) != hipSuccess) printf("async live %d failed\n", i); }
        hipStreamSynchronize(0);
        t3 = now();
        printf("8 x hipMallocAsync 16 GB live: %.3f s, free all %.3f s, again %.3f s\n", t1 - t0, t2 - t1, t3 - t2);
    }
    return 0;
}
