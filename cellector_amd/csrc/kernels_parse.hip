// Device-side parse of the vartrix alt.mtx / ref.mtx text (rows A1 and f2 of the scope table).
//
// The host only brings bytes (mmap for plain files, zlib inflate for ".gz", multi-member) and reads the three header
// lines; the data lines are tokenised and converted on the GPU:
//   k_nl_count               newlines per 128-byte segment, scanned: the index of the first line that starts in a segment;
//   k_parse_lines            a thread per segment, for every line that starts there: split_whitespace + parse::<usize>() of the tokens the reference reads
//                            (load_data.rs:190-204): alt file tokens 0,1,2 (locus, cell, alt count), ref file token 2 only
//                            (its indices are never read); any failure records the smallest offending line;
//   k_pair_check / k_pair_fill   zip the two files (shorter one wins, like izip!), range checks, this shard's cell range,
//                            ordered compaction into the staged COO (locus u32, cell_local u32, alt u16, ref u16).
// The text contract is the reference's; errors come back as CELLECTOR_EPARSE / CELLECTOR_EINVAL with the line number.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "ctx.h"
#include "device_math.h"

#define PB 256

__device__ __forceinline__ bool is_ws(uint8_t ch)
{
    return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n' || ch == '\f' || ch == '\v';
}

// ---- line starts ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t newline_mask16(const uint8_t *__restrict__ text, uint64_t n, uint64_t base)
{
    uint32_t m = 0;
    if (base + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + base);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (((w[q] >> (8 * b)) & 0xffu) == '\n') m |= 1u << (4 * q + b);
    } else {
        for (uint64_t i = base; i < n; i++)
            if (text[i] == '\n') m |= 1u << (uint32_t)(i - base);
    }
    return m;
}

// (a thread takes NL_SEG bytes = 8 units of 16: the count array is 1/16 of the text instead of 1/2 — at 30 GB of text per
// file that is 15 GB of VRAM less to map)
#define NL_SEG 128
__global__ __launch_bounds__(PB) void k_nl_count(const uint8_t *__restrict__ text, uint64_t n, uint64_t *__restrict__ cnt)
{
    const uint64_t t = (uint64_t)blockIdx.x * PB + threadIdx.x;
    const uint64_t base = t * NL_SEG;
    if (base >= n) return;
    uint32_t k = 0;
#pragma unroll
    for (int u = 0; u < NL_SEG / 16; u++)
        if (base + 16 * u < n) k += __popc(newline_mask16(text, n, base + 16 * u));
    cnt[t] = k;
}

// ---- tokens ---------------------------------------------------------------------------------------------------
// parse token `idx` of [p, e) as an unsigned integer (optional leading '+'); false on missing / malformed / > 2^32-1
// parse tokens first_idx .. first_idx + n_tok - 1 of the LINE that starts at p (it ends at the next '\n'; e bounds the text)
// as unsigned integers (optional leading '+'); false on a missing / malformed token or a value above 2^32-1
__device__ bool token_u32(const uint8_t *__restrict__ p, const uint8_t *__restrict__ e, int first_idx, int n_tok,
                          uint32_t *out)
{
    int t = 0;
    while (true) {
        while (p < e && *p != '\n' && is_ws(*p)) p++;
        if (p >= e || *p == '\n') return false;
        const uint8_t *s = p;
        while (p < e && !is_ws(*p)) p++;
        if (t >= first_idx) {
            if (*s == '+') s++;
            if (s == p) return false;
            uint64_t v = 0;
            for (; s < p; s++) {
                if (*s < '0' || *s > '9') return false;
                v = v * 10 + (uint64_t)(*s - '0');
                if (v > 0xffffffffull) return false;
            }
            out[t - first_idx] = (uint32_t)v;
            if (t - first_idx + 1 == n_tok) return true;
        }
        t++;
    }
}

// ALT file: tokens 0,1,2 -> (locus1, cell1, count); REF file: token 2 -> count.  A thread takes the lines that START in its
// NL_SEG bytes (line 0 at byte 0, line k after the k-th newline; off[t] = newlines before the segment, from the scan of
// k_nl_count) — no array of line starts is ever materialised (at 2e9 lines per file that array was 16 GB of VRAM each).
#define PARSE_TAIL 64  // bytes staged beyond the block's segments: a line that starts near the end finishes in there
// Windowed use (big files, parse_windowed below): `text` is one window of the data section plus look-ahead bytes, n_count
// the bytes whose newlines belong to this window, line_base the number of newlines before the window, first = the window
// starts at the first data byte (line 0 has no newline before it), more = the file goes on beyond the n bytes given.
template <bool ALT>
__global__ __launch_bounds__(PB) void k_parse_lines(const uint8_t *__restrict__ text, uint64_t n, uint64_t n_count, uint64_t n_lines,
                                                    uint64_t line_base, bool first, bool more,
                                                    const uint64_t *__restrict__ off, uint32_t *__restrict__ o0,
                                                    uint32_t *__restrict__ o1, uint32_t *__restrict__ o2,
                                                    unsigned long long *__restrict__ first_bad)
{
    // the block's PB segments (+ a tail) go through LDS: coalesced 16-byte loads instead of byte-wise global reads
    __shared__ __attribute__((aligned(16))) uint8_t s_text[PB * NL_SEG + PARSE_TAIL];
    const uint64_t blk = (uint64_t)blockIdx.x * PB * NL_SEG;
    const uint64_t avail = blk < n ? min((uint64_t)(PB * NL_SEG + PARSE_TAIL), n - blk) : 0;  // text has 16 bytes of padding
    for (uint32_t i = threadIdx.x * 16; i < avail; i += PB * 16)
        *reinterpret_cast<uint4 *>(s_text + i) = *reinterpret_cast<const uint4 *>(text + blk + i);
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * PB + threadIdx.x;
    const uint64_t base = t * NL_SEG;
    if (base >= n_count) return;
    auto line = [&](uint64_t i, uint64_t start) {
        if (i >= n_lines || start >= n) return;
        uint32_t v[3] = {0, 0, 0};
        // inside the staged window (a line is far shorter than the tail) parse from LDS, else from the text itself
        const uint64_t rel = start - blk;
        const uint8_t *p = text + start, *e = text + n;
        bool closed = false;
        if (rel + PARSE_TAIL <= avail) {
            const uint8_t *q = s_text + rel, *qe = q + PARSE_TAIL;
            for (const uint8_t *z = q; z < qe; z++)
                if (*z == '\n') { closed = true; break; }
            if (closed) { p = q; e = qe; }
        }
        if (!closed && more) {  // a line that runs out of the window's look-ahead: not supported (PW_LOOK bytes)
            for (const uint8_t *z = p; z < e; z++)
                if (*z == '\n') { closed = true; break; }
            if (!closed) {
                atomicMin(first_bad, (unsigned long long)i);
                return;
            }
        }
        const bool ok = ALT ? token_u32(p, e, 0, 3, v) : token_u32(p, e, 2, 1, v);
        if (!ok) {
            atomicMin(first_bad, (unsigned long long)i);
            return;
        }
        if (ALT) { o0[i] = v[0]; o1[i] = v[1]; o2[i] = v[2]; } else o2[i] = v[0];
    };
    if (t == 0 && first) line(line_base, 0);
    uint64_t k = line_base + off[t];
    for (int u = 0; u < NL_SEG / 16; u++) {
        const uint64_t ub = base + 16 * u;
        if (ub >= n_count) break;
        uint32_t m = newline_mask16(text, n_count, ub);
        while (m) {
            const int b = __ffs((int)m) - 1;
            m &= m - 1;
            line(++k, ub + b + 1);
        }
    }
}

// ---- zip, validate, shard filter ------------------------------------------------------------------------------
enum { PE_NONE = 0, PE_INDEX0 = 1, PE_LOCUS = 2, PE_CELL = 3, PE_COUNT = 4 };

__global__ __launch_bounds__(PB) void k_pair_check(uint64_t n, const uint32_t *__restrict__ l1, const uint32_t *__restrict__ c1,
                                                   const uint32_t *__restrict__ a, const uint32_t *__restrict__ r,
                                                   uint64_t total_loci, uint64_t total_cells, uint64_t cb, uint64_t ce,
                                                   uint64_t *__restrict__ keep, unsigned long long *__restrict__ first_bad,
                                                   uint32_t *__restrict__ bad_kind, uint32_t *__restrict__ unsorted)
{
    const uint64_t i = (uint64_t)blockIdx.x * PB + threadIdx.x;
    if (i > n) return;
    if (i == n) { if (keep) keep[i] = 0; return; }
    uint32_t kind = PE_NONE;
    if (l1[i] == 0 || c1[i] == 0) kind = PE_INDEX0;          // `tok - 1` underflows in the reference
    else if (l1[i] > total_loci) kind = PE_LOCUS;
    else if (c1[i] > total_cells) kind = PE_CELL;
    else if (a[i] > CELLECTOR_MAX_COUNT || r[i] > CELLECTOR_MAX_COUNT) kind = PE_COUNT;
    if (kind != PE_NONE) {
        if (atomicMin(first_bad, (unsigned long long)i) > i) *bad_kind = kind;  // best effort: kind of the smallest seen
        if (keep) keep[i] = 0;
        return;
    }
    const uint64_t c0 = c1[i] - 1;
    const bool mine = c0 >= cb && c0 < ce;
    if (keep) keep[i] = mine ? 1 : 0;  // (null: the shard holds every cell, every valid entry stays where it is)
    if (i + 1 < n && l1[i + 1] < l1[i]) *unsorted = 1;  // file order not locus-major (all entries, a superset check)
}

__global__ __launch_bounds__(PB) void k_pair_fill(uint64_t n, const uint32_t *__restrict__ l1, const uint32_t *__restrict__ c1,
                                                  const uint32_t *__restrict__ a, const uint32_t *__restrict__ r, uint64_t cb,
                                                  uint64_t ce, const uint64_t *__restrict__ pos, uint32_t *__restrict__ o_locus,
                                                  uint32_t *__restrict__ o_cell, uint16_t *__restrict__ o_alt,
                                                  uint16_t *__restrict__ o_ref)
{
    const uint64_t i = (uint64_t)blockIdx.x * PB + threadIdx.x;
    if (i >= n) return;
    const uint64_t c0 = (uint64_t)c1[i] - 1;
    if (c1[i] == 0 || c0 < cb || c0 >= ce) return;
    const uint64_t p = pos[i];
    o_locus[p] = l1[i] - 1;
    o_cell[p] = (uint32_t)(c0 - cb);
    o_alt[p] = (uint16_t)a[i];
    o_ref[p] = (uint16_t)r[i];
}

// a shard that holds every cell: the token arrays BECOME the staged COO (indices made 0-based in place, counts narrowed
// into two new u16 arrays) — no keep flags, no scan, no second copy of the indices: 32 GB less fresh VRAM at 2e9 entries
__global__ __launch_bounds__(PB) void k_pair_take(uint64_t n, uint32_t *__restrict__ l1, uint32_t *__restrict__ c1,
                                                  const uint32_t *__restrict__ a, const uint32_t *__restrict__ r,
                                                  uint16_t *__restrict__ o_alt, uint16_t *__restrict__ o_ref)
{
    const uint64_t i = (uint64_t)blockIdx.x * PB + threadIdx.x;
    if (i >= n) return;
    l1[i] -= 1;
    c1[i] -= 1;
    o_alt[i] = (uint16_t)a[i];
    o_ref[i] = (uint16_t)r[i];
}

// ===============================================================================================================
namespace {

// the bytes of one input file: inflated into `owned` (.gz), mapped (plain, below FB_UNMAPPED), or — a plain file of
// FB_UNMAPPED bytes and more — not mapped at all: its windows are pread() straight into the pinned upload buffers.  (Mapping 2 x 31 GB meant
// 15 M page-table entries to fault in and to tear down again: the munmap alone took 0.7 s, during which the runtime's
// own allocations queue for the address-space lock.)
#define FB_UNMAPPED (1ull << 30)
struct FileBytes {
    const uint8_t *data = nullptr;  // null: unmapped, use read()
    size_t size = 0;
    void *map = nullptr;
    size_t map_len = 0;
    int fd = -1;
    std::vector<uint8_t> owned;  // .gz: the inflated file; unmapped: its first FB_HEAD bytes (the header lines)
    size_t head_len = 0;
    ~FileBytes()
    {
        if (map) munmap(map, map_len);
        if (fd >= 0) close(fd);
    }
    const uint8_t *head() const { return data ? data : owned.data(); }
    size_t head_size() const { return data ? size : head_len; }
    bool read(size_t off, size_t len, uint8_t *dst) const
    {
        if (data) {
            memcpy(dst, data + off, len);
            return true;
        }
        while (len) {
            const ssize_t got = pread(fd, dst, len, (off_t)off);
            if (got <= 0) return false;
            dst += got; off += (size_t)got; len -= (size_t)got;
        }
        return true;
    }
};
#define FB_HEAD (1u << 20)

// A ".gz" whose members all carry the BGZF extra field (bgzip: blocks of at most 64 KB, each a gzip member with its own
// compressed size in a 'B','C' subfield — the reference's MultiGzDecoder reads such a file like any multi-member gzip,
// load_data.rs:246) is inflated block-parallel: the member boundaries are found by hopping over the size fields, the
// output offsets are the prefix sums of the members' ISIZE trailers, and host threads inflate ranges of blocks straight
// into place (raw deflate, CRC-32 and length of every block checked like gzread does).  A single zlib stream inflates at
// ~0.35 GB/s of text; a plain gzip file has no such index and keeps the serial path below.
struct BgzfBlock { size_t off, clen; uint32_t xlen, isize; size_t out; };
bool bgzf_index(const uint8_t *f, size_t n, std::vector<BgzfBlock> *blocks, size_t *total)
{
    size_t pos = 0, out = 0;
    while (pos < n) {
        if (n - pos < 18 || f[pos] != 0x1f || f[pos + 1] != 0x8b || f[pos + 2] != 8 || !(f[pos + 3] & 4)) return false;
        if (f[pos + 3] & ~4u) return false;  // (name / comment / header CRC: not what bgzip writes — leave it to zlib)
        const uint32_t xlen = f[pos + 10] | ((uint32_t)f[pos + 11] << 8);
        if (n - pos < 12 + (size_t)xlen + 8) return false;
        uint32_t bsize = 0;
        bool have = false;
        for (size_t q = pos + 12, e = pos + 12 + xlen; q + 4 <= e;) {
            const uint32_t slen = f[q + 2] | ((uint32_t)f[q + 3] << 8);
            if (f[q] == 'B' && f[q + 1] == 'C' && slen == 2 && q + 6 <= e) { bsize = f[q + 4] | ((uint32_t)f[q + 5] << 8); have = true; }
            q += 4 + slen;
        }
        const size_t clen = (size_t)bsize + 1;
        if (!have || clen < 12 + (size_t)xlen + 8 || n - pos < clen) return false;
        const uint8_t *tr = f + pos + clen - 4;
        const uint32_t isize = tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        if (isize > 65536u) return false;  // (bgzip never puts more than 64 KB into a block: not BGZF, leave it to zlib)
        blocks->push_back({pos, clen, xlen, isize, out});
        out += isize;
        pos += clen;
    }
    *total = out;
    return !blocks->empty();
}
bool bgzf_inflate(const uint8_t *f, const std::vector<BgzfBlock> &blocks, uint8_t *dst)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    if ((size_t)nt > blocks.size()) nt = (unsigned)blocks.size();
    std::atomic<bool> ok(true);
    auto work = [&](size_t b0, size_t b1) {
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) { ok = false; return; }
        for (size_t b = b0; b < b1 && ok; b++) {
            const BgzfBlock &k = blocks[b];
            z.next_in = const_cast<Bytef *>(f + k.off + 12 + k.xlen);
            z.avail_in = (uInt)(k.clen - 12 - k.xlen - 8);
            z.next_out = dst + k.out;
            z.avail_out = k.isize;
            const int r = k.isize || z.avail_in ? inflate(&z, Z_FINISH) : Z_STREAM_END;
            const uint8_t *tr = f + k.off + k.clen - 8;
            const uint32_t crc = tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
            if (r != Z_STREAM_END || z.avail_out != 0 || (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst + k.out, k.isize) != crc) ok = false;
            inflateReset(&z);
        }
        inflateEnd(&z);
    };
    std::vector<std::thread> th;
    const size_t per = (blocks.size() + nt - 1) / nt;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, std::min(blocks.size(), t * per), std::min(blocks.size(), (t + 1) * per));
    work(0, std::min(blocks.size(), per));
    for (auto &t : th) t.join();
    return ok;
}

// reader (load_data.rs:240-251): ".gz" by extension (multi-member), plain otherwise
bool load_bytes(const char *path, FileBytes *fb)
{
    const size_t n = strlen(path);
    if (n >= 3 && strcmp(path + n - 3, ".gz") == 0) {
        {   // block-compressed (bgzip)?  then in parallel
            const int fd = open(path, O_RDONLY);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && st.st_size > 0 && !getenv("CELLECTOR_NO_BGZF")) {
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    std::vector<BgzfBlock> blocks;
                    size_t total = 0;
                    bool done = false;
                    try {
                        if (bgzf_index((const uint8_t *)m, (size_t)st.st_size, &blocks, &total)) {
                            fb->owned.resize(total ? total : 1);
                            done = bgzf_inflate((const uint8_t *)m, blocks, fb->owned.data());
                            if (done) { fb->data = fb->owned.data(); fb->size = total; }
                        }
                    } catch (const std::exception &) {  // (no room for the index or the text: the serial reader decides)
                        done = false;
                    }
                    munmap(m, (size_t)st.st_size);
                    if (done) { close(fd); return true; }
                    fb->owned.clear();  // (a damaged block: the serial reader below reports what zlib makes of the file)
                }
            }
            if (fd >= 0) close(fd);
        }
        gzFile gz = gzopen(path, "rb");
        if (!gz) return false;
        gzbuffer(gz, 1 << 20);
        size_t cap = 1 << 24, len = 0;
        fb->owned.resize(cap);
        for (;;) {
            if (len == cap) fb->owned.resize(cap *= 2);
            const int got = gzread(gz, fb->owned.data() + len, (unsigned)std::min<size_t>(cap - len, 1u << 30));
            if (got < 0) {  // a damaged stream (CRC, truncated member): the reference's decoder fails the read as well
                gzclose(gz);
                return false;
            }
            if (got == 0) break;
            len += (size_t)got;
        }
        gzclose(gz);
        fb->data = fb->owned.data();
        fb->size = len;
        return true;
    }
    fb->fd = open(path, O_RDONLY);
    if (fb->fd < 0) return false;
    struct stat st;
    if (fstat(fb->fd, &st) != 0) return false;
    fb->size = (size_t)st.st_size;
    const char *um = getenv("CELLECTOR_UNMAPPED_MIN");  // (tests: the unmapped path on small files)
    if (fb->size >= (um ? (size_t)strtoull(um, nullptr, 10) : (size_t)FB_UNMAPPED) && fb->size > 0) {
        fb->head_len = std::min<size_t>(fb->size, FB_HEAD);
        fb->owned.resize(fb->head_len);
        return fb->read(0, fb->head_len, fb->owned.data());
    }
    if (fb->size) {
        fb->map = mmap(nullptr, fb->size, PROT_READ, MAP_PRIVATE, fb->fd, 0);
        if (fb->map == MAP_FAILED) { fb->map = nullptr; return false; }
        fb->map_len = fb->size;
        madvise(fb->map, fb->size, MADV_SEQUENTIAL);
        fb->data = (const uint8_t *)fb->map;
    }
    return true;
}

// consume_mtx_header (load_data.rs:206-223): exactly three lines; returns the offset of the first data byte
size_t skip_header(const FileBytes &fb, std::string *third)
{
    size_t pos = 0;
    for (int x = 0; x < 3; x++) {
        // (an unmapped file: the three lines are looked for in its first FB_HEAD bytes)
        const uint8_t *hd = fb.head();
        const size_t hn = fb.head_size();
        const void *nl = pos < hn ? memchr(hd + pos, '\n', hn - pos) : nullptr;
        const size_t end = nl ? (size_t)((const uint8_t *)nl - hd) : hn;
        if (x == 2 && third) third->assign((const char *)hd + pos, end - pos);
        pos = nl ? end + 1 : hn;
    }
    return pos;
}

bool host_tok_u64(const std::string &s, int idx, uint64_t *out)
{
    size_t p = 0;
    for (int t = 0;; t++) {
        while (p < s.size() && isspace((unsigned char)s[p])) p++;
        if (p >= s.size()) return false;
        size_t b = p;
        while (p < s.size() && !isspace((unsigned char)s[p])) p++;
        if (t == idx) {
            if (s[b] == '+') b++;
            if (b == p) return false;
            uint64_t v = 0;
            for (; b < p; b++) {
                if (s[b] < '0' || s[b] > '9') return false;
                v = v * 10 + (uint64_t)(s[b] - '0');
            }
            *out = v;
            return true;
        }
    }
}

struct DevText {
    uint8_t *text = nullptr;
    uint64_t n = 0, n_lines = 0;
    uint64_t *seg_off = nullptr;  // newlines before each NL_SEG-byte segment (exclusive scan of k_nl_count)
    uint64_t n_seg = 0;
};

inline unsigned pgrid(uint64_t n) { return (unsigned)((n + PB - 1) / PB ? (n + PB - 1) / PB : 1); }

// bytes of the data section -> device.  A plain hipMemcpy out of the mapped file is staged by ONE runtime thread and
// swings between 8 and 25 GB/s from box to box (where the page cache sits relative to the GPU); here UP_THREADS host
// threads copy each piece into one of two pinned buffers while the previous piece is on its way over PCIe.
#define UP_PIECE (256ull << 20)
#define UP_THREADS 8
#define UP_MIN (4ull << 30)
cellector_status upload_text(cellector_ctx *c, const FileBytes &fb, size_t data_off, DevText *dt)
{
    HIPCHK(c, hipSetDevice(c->device));
    dt->n = fb.size - data_off;
    // a final line without '\n' still counts (BufRead::lines); normalise by treating the end of data as a terminator
    CHK(dev_alloc(c, &dt->text, dt->n + 16));
    // (everything below is ordered on the ctx's stream, which may be a non-blocking one: no null-stream calls)
    HIPCHK(c, hipMemsetAsync(dt->text + dt->n, '\n', 16, c->stream));
    if (dt->n < UP_MIN) {  // pinning the two buffers costs ~0.1 s: only worth it for multi-GB files
        const size_t piece = 1ull << 30;
        for (size_t o = 0; o < dt->n; o += piece)
            HIPCHK(c, hipMemcpyAsync(dt->text + o, fb.data + data_off + o, std::min(piece, (size_t)dt->n - o), hipMemcpyHostToDevice,
                                     c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return CELLECTOR_OK;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (the padding is in place before the upload stream's copies are consumed)
    uint8_t *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t up = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        e = hipHostMalloc((void **)&pin[b], UP_PIECE);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
    }
    const uint8_t *src = fb.data + data_off;
    uint64_t piece = 0;
    for (size_t o = 0; o < dt->n && e == hipSuccess; o += UP_PIECE, piece++) {
        const int b = (int)(piece & 1);
        const size_t len = std::min((size_t)UP_PIECE, (size_t)dt->n - o);
        if (piece >= 2) e = hipEventSynchronize(ev[b]);  // the copy that last read this buffer is done
        if (e != hipSuccess) break;
        std::thread th[UP_THREADS];
        const size_t slice = (len + UP_THREADS - 1) / UP_THREADS;
        for (int t = 0; t < UP_THREADS; t++)
            th[t] = std::thread([=] {
                const size_t b0 = std::min(len, (size_t)t * slice), b1 = std::min(len, b0 + slice);
                if (b1 > b0) memcpy(pin[b] + b0, src + o + b0, b1 - b0);
            });
        for (int t = 0; t < UP_THREADS; t++) th[t].join();
        e = hipMemcpyAsync(dt->text + o, pin[b], len, hipMemcpyHostToDevice, up);
        if (e == hipSuccess) e = hipEventRecord(ev[b], up);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(up);
    for (int b = 0; b < 2; b++) {
        if (ev[b]) (void)hipEventDestroy(ev[b]);
        if (pin[b]) (void)hipHostFree(pin[b]);
    }
    if (up) (void)hipStreamDestroy(up);
    if (e != hipSuccess) return ctx_fail(c, CELLECTOR_EDEVICE, "text upload: %s", hipGetErrorString(e));
    return CELLECTOR_OK;
}

cellector_status split_lines(cellector_ctx *c, const FileBytes &fb, DevText *dt)
{
    const bool unterminated = dt->n > 0 && fb.data[fb.size - 1] != '\n';
    const uint64_t n_scan = dt->n + (unterminated ? 1 : 0);  // include one padding '\n' as the terminator
    const uint64_t nthreads = (n_scan + NL_SEG - 1) / NL_SEG;
    CHK(dev_alloc(c, &dt->seg_off, nthreads + 1));
    HIPCHK(c, hipMemsetAsync(dt->seg_off, 0, (nthreads + 1) * 8, c->stream));
    if (nthreads) hipLaunchKernelGGL(k_nl_count, dim3(pgrid(nthreads)), dim3(PB), 0, c->stream, dt->text, n_scan, dt->seg_off);
    HIPCHK(c, hipGetLastError());
    uint64_t n_nl = 0;
    CHK(dev_exclusive_scan_u64(c, dt->seg_off, nthreads + 1, &n_nl));
    dt->n_lines = n_nl;  // every line is terminated now
    dt->n_seg = nthreads;
    dt->n = n_scan;
    return CELLECTOR_OK;
}

// ---- windowed parse of one file -------------------------------------------------------------------------------
// A multi-GB data section never resides on the device as a whole: a producer thread copies windows of it (plus PW_LOOK
// bytes of look-ahead: a line that starts in a window may end behind it) through pinned buffers into one of PW_NB device
// buffers, the caller's thread counts the window's newlines, scans them and tokenises the lines that start there straight
// into the token arrays, at the running line count.  The text of BASELINE configs[4] is 2 x 31 GB: as one allocation per
// file it cost 65 GB of fresh VRAM (30-50 ms per GB to map) before the first line was parsed.
#define PW_LOOK (1ull << 20)  // the longest line a windowed file may hold
#define PW_NB 3
#define PW_WINDOW (256ull << 20)
#define PW_SPLIT_WINDOW (32ull << 20)  // largest window of the split ingest of a multi-device ctx (every shard has its own ring)
#define PW_MIN (1ull << 30)  // data sections from this size on go through the windows
#define PW_THREADS 8  // host threads filling a pinned buffer (pread out of the page cache; 16 threads measured no faster)

// the windows' buffers, shared by the two files of a pair (pinning 3 x 257 MB takes ~0.05 s)
struct PwBuffers {
    uint8_t *pin[PW_NB] = {}, *dev[PW_NB] = {};
    hipEvent_t ev_up[PW_NB] = {}, ev_free[PW_NB] = {};
    hipStream_t up = nullptr;
    uint64_t *seg = nullptr;
    uint64_t window = 0;
    int n = 0;
    cellector_status make(cellector_ctx *c, uint64_t win, int count)
    {
        window = win;
        const size_t buf_bytes = window + PW_LOOK + 64;
        hipError_t e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
        for (int b = 0; b < count && e == hipSuccess; b++) {
            e = hipHostMalloc((void **)&pin[b], buf_bytes);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_up[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_free[b], hipEventDisableTiming);
            if (e == hipSuccess && dev_alloc(c, &dev[b], buf_bytes) != CELLECTOR_OK) e = hipErrorOutOfMemory;
            if (e == hipSuccess) n = b + 1;
        }
        if (e == hipSuccess && dev_alloc(c, &seg, window / NL_SEG + 2) != CELLECTOR_OK) e = hipErrorOutOfMemory;
        return e == hipSuccess ? CELLECTOR_OK : ctx_fail(c, CELLECTOR_EDEVICE, "parse buffers: %s", hipGetErrorString(e));
    }
    ~PwBuffers()
    {
        if (up) (void)hipStreamSynchronize(up);
        for (int b = 0; b < PW_NB; b++) {
            if (ev_up[b]) (void)hipEventDestroy(ev_up[b]);
            if (ev_free[b]) (void)hipEventDestroy(ev_free[b]);
            if (pin[b]) (void)hipHostFree(pin[b]);
            dev_free(dev[b]);
        }
        dev_free(seg);
        if (up) (void)hipStreamDestroy(up);
    }
};

template <typename T>
cellector_status grow_tokens(cellector_ctx *c, T **arr, uint64_t used, uint64_t new_cap)
{
    T *bigger = nullptr;
    CHK(dev_alloc(c, &bigger, new_cap));
    hipError_t e = hipSuccess;
    if (used) e = hipMemcpyAsync(bigger, *arr, used * sizeof(T), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        dev_free(bigger);
        return ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e));
    }
    dev_free(*arr);
    *arr = bigger;
    return CELLECTOR_OK;
}

// tokens of every line of the data section: ALT -> o0, o1, o2 (locus, cell, count), else o2 only (count); arrays are
// allocated here (capacity from the header's entry count, grown if the file holds more lines)
template <bool ALT>
cellector_status parse_windowed(cellector_ctx *c, const FileBytes &fb, size_t data_off, PwBuffers &B, uint64_t cap_hint,
                                uint32_t **o0, uint32_t **o1, uint32_t **o2, uint64_t *n_lines, unsigned long long *bad,
                                uint64_t w_begin = 0, uint64_t w_end = ~0ull)
{
    // [w_begin, w_end): the windows this call takes (a multi-device ingest gives every GPU a range of them).  Line indices
    // are then LOCAL: the number of newlines before the line inside the range; the line that starts right behind the range's
    // last newline — in the next range's bytes — is still this call's (look-ahead).  *n_lines = newlines in the range.
    const uint64_t window = B.window;
    uint8_t *const *pin = B.pin, *const *dev = B.dev;
    hipEvent_t *ev_up = B.ev_up, *ev_free = B.ev_free;
    hipStream_t up = B.up;
    uint64_t *seg = B.seg;
    const uint64_t n_ring = (uint64_t)B.n;  // buffers in the ring
    const uint64_t nb = fb.size - data_off;
    *n_lines = 0;
    uint64_t cap = (cap_hint ? cap_hint : nb / 12) + 16;
    if (ALT) { CHK(dev_alloc(c, o0, cap)); CHK(dev_alloc(c, o1, cap)); }
    CHK(dev_alloc(c, o2, cap));
    if (nb == 0) return CELLECTOR_OK;
    uint8_t last_byte = '\n';
    if (!fb.read(data_off + nb - 1, 1, &last_byte)) return ctx_fail(c, CELLECTOR_EIO, "cannot read the input file");
    const bool unterminated = last_byte != '\n';
    const uint64_t n_win = (nb + window - 1) / window;
    if (w_end > n_win) w_end = n_win;
    if (w_begin >= w_end) return CELLECTOR_OK;
    hipError_t e = hipSuccess;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t produced = 0, consumed = 0;
    bool stop = false;
    hipError_t perr = hipSuccess;
    cellector_status st = CELLECTOR_OK;
    {
        std::thread producer([&] {
            hipError_t pe = hipSetDevice(c->device);
            for (uint64_t w = w_begin; w < w_end && pe == hipSuccess; w++) {
                const uint64_t wi = w - w_begin;
                const int b = (int)(wi % n_ring);
                if (wi >= n_ring) {  // the buffer's previous window has been tokenised
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return consumed + n_ring > wi || stop; });
                    if (stop) return;
                    lk.unlock();
                    pe = hipEventSynchronize(ev_free[b]);
                    if (pe != hipSuccess) break;
                }
                const uint64_t o = w * window;
                const size_t len = (size_t)std::min<uint64_t>(window + PW_LOOK, nb - o);
                std::thread th[PW_THREADS];
                bool got[PW_THREADS];
                const size_t slice = (len + PW_THREADS - 1) / PW_THREADS;
                for (int t = 0; t < PW_THREADS; t++)
                    th[t] = std::thread([&, t] {
                        const size_t b0 = std::min(len, (size_t)t * slice), b1 = std::min(len, b0 + slice);
                        got[t] = b1 <= b0 || fb.read(data_off + o + b0, b1 - b0, pin[b] + b0);
                    });
                for (int t = 0; t < PW_THREADS; t++) th[t].join();
                for (int t = 0; t < PW_THREADS; t++)
                    if (!got[t]) pe = hipErrorFileNotFound;  // (reported as the upload's failure)
                if (pe != hipSuccess) break;
                memset(pin[b] + len, '\n', 32);  // the end of the data terminates a last line without '\n' (BufRead::lines)
                pe = hipMemcpyAsync(dev[b], pin[b], len + 32, hipMemcpyHostToDevice, up);
                if (pe == hipSuccess) pe = hipEventRecord(ev_up[b], up);
                if (pe != hipSuccess) break;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    produced = wi + 1;
                }
                cv.notify_all();
            }
            if (pe != hipSuccess) {
                std::lock_guard<std::mutex> lk(mu);
                perr = pe;
                stop = true;
            }
            cv.notify_all();
        });
        uint64_t line_base = 0;
        for (uint64_t w = w_begin; w < w_end && st == CELLECTOR_OK; w++) {
            const uint64_t wi = w - w_begin;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return produced > wi || stop; });
                if (produced <= wi) {
                    st = ctx_fail(c, CELLECTOR_EDEVICE, "text upload: %s", hipGetErrorString(perr));
                    break;
                }
            }
            const int b = (int)(wi % n_ring);
            const uint64_t o = w * window, in_win = std::min<uint64_t>(window, nb - o);
            const uint64_t len = std::min<uint64_t>(window + PW_LOOK, nb - o);
            const bool last = w + 1 == n_win, more = o + len < nb;
            const uint64_t n_count = in_win + (last && unterminated ? 1 : 0);  // (the padding newline closes the last line)
            const uint64_t n_buf = len + (last && unterminated ? 1 : 0);
            const uint64_t nseg = (n_count + NL_SEG - 1) / NL_SEG;
            e = hipStreamWaitEvent(c->stream, ev_up[b], 0);
            if (e == hipSuccess) e = hipMemsetAsync(seg, 0, (nseg + 1) * 8, c->stream);
            if (e != hipSuccess) { st = ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e)); break; }
            hipLaunchKernelGGL(k_nl_count, dim3(pgrid(nseg)), dim3(PB), 0, c->stream, dev[b], n_count, seg);
            uint64_t n_nl = 0;
            st = dev_exclusive_scan_u64(c, seg, nseg + 1, &n_nl);
            if (st != CELLECTOR_OK) break;
            if (line_base + n_nl + 1 > cap) {
                const uint64_t want = std::max(line_base + n_nl + 16, cap + cap / 2);
                if (ALT) { st = grow_tokens(c, o0, line_base + 1, want); if (st == CELLECTOR_OK) st = grow_tokens(c, o1, line_base + 1, want); }
                if (st == CELLECTOR_OK) st = grow_tokens(c, o2, line_base + 1, want);
                if (st != CELLECTOR_OK) break;
                cap = want;
            }
            hipLaunchKernelGGL(k_parse_lines<ALT>, dim3(pgrid(nseg)), dim3(PB), 0, c->stream, dev[b], n_buf, n_count, cap, line_base,
                               w == 0, more, seg, ALT ? *o0 : (uint32_t *)nullptr, ALT ? *o1 : (uint32_t *)nullptr, *o2, bad);
            e = hipEventRecord(ev_free[b], c->stream);
            if (e != hipSuccess) { st = ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e)); break; }
            {
                std::lock_guard<std::mutex> lk(mu);
                consumed = wi + 1;
            }
            cv.notify_all();
            line_base += n_nl;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            if (st != CELLECTOR_OK) stop = true;
        }
        cv.notify_all();
        producer.join();
        e = hipStreamSynchronize(c->stream);
        if (st == CELLECTOR_OK && e != hipSuccess) st = ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e));
        if (st == CELLECTOR_OK) *n_lines = line_base;
    }
    if (up) (void)hipStreamSynchronize(up);  // (the buffers go on to the pair's other file)
    return st;
}

}  // namespace

struct MtxInput {
    FileBytes fa, fr;
    size_t off_a = 0, off_r = 0;
    uint64_t total_loci = 0, total_cells = 0;
    uint64_t nnz_hint = 0;  // third number of the size line (0: absent); a capacity hint, never trusted
};

// open both files (bytes only) and read the dims from the REF file's third header line (load_data.rs:216-220)
cellector_status mtx_input_open(const cellector_ctx *c, const char *alt_path, const char *ref_path, MtxInput **out,
                                uint64_t *total_loci, uint64_t *total_cells)
{
    MtxInput *in = new (std::nothrow) MtxInput();
    if (!in) return ctx_fail(c, CELLECTOR_ENOMEM, "out of host memory");
    cellector_status st = CELLECTOR_OK;
    std::string third;
    bool ok_a = false, ok_r = false;
    {   // the two files are independent byte streams: inflate / map them concurrently
        std::thread ta([&] { ok_a = load_bytes(alt_path, &in->fa); });
        ok_r = load_bytes(ref_path, &in->fr);
        ta.join();
    }
    if (!ok_a) st = ctx_fail(c, CELLECTOR_EIO, "couldn't open file %s", alt_path);
    else if (!ok_r) st = ctx_fail(c, CELLECTOR_EIO, "couldn't open file %s", ref_path);
    else {
        in->off_a = skip_header(in->fa, nullptr);
        in->off_r = skip_header(in->fr, &third);
        if (!host_tok_u64(third, 0, &in->total_loci) || !host_tok_u64(third, 1, &in->total_cells))
            st = ctx_fail(c, CELLECTOR_EPARSE, "cannot parse the matrix market size line of %s", ref_path);
        else if (!host_tok_u64(third, 2, &in->nnz_hint) || in->nnz_hint > (in->fr.size - in->off_r) / 4)
            in->nnz_hint = 0;  // (a line holds at least "1 1 1": a hint beyond the bytes there are is nonsense)
    }
    if (st != CELLECTOR_OK) {
        delete in;
        return st;
    }
    *total_loci = in->total_loci;
    *total_cells = in->total_cells;
    *out = in;
    return CELLECTOR_OK;
}

void mtx_input_close(MtxInput *in) { delete in; }

// a file small enough to sit on the device whole: one upload, one scan of its newlines, one tokeniser launch
template <bool ALT>
static cellector_status parse_whole(cellector_ctx *c, const FileBytes &fb, size_t data_off, uint32_t **o0, uint32_t **o1,
                                    uint32_t **o2, uint64_t *n_lines, unsigned long long *bad)
{
    DevText t;
    cellector_status st = upload_text(c, fb, data_off, &t);
    if (st == CELLECTOR_OK) st = split_lines(c, fb, &t);
    const uint64_t n = t.n_lines;
    if (st == CELLECTOR_OK && ALT) { st = dev_alloc(c, o0, n); if (st == CELLECTOR_OK) st = dev_alloc(c, o1, n); }
    if (st == CELLECTOR_OK) st = dev_alloc(c, o2, n);
    if (st == CELLECTOR_OK && n) {
        hipLaunchKernelGGL(k_parse_lines<ALT>, dim3(pgrid(t.n_seg)), dim3(PB), 0, c->stream, t.text, t.n, t.n, n, 0ull, true, false,
                           t.seg_off, ALT ? *o0 : (uint32_t *)nullptr, ALT ? *o1 : (uint32_t *)nullptr, *o2, bad);
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e));
    }
    dev_free(t.text); dev_free(t.seg_off);
    *n_lines = n;
    return st;
}

// ---- split ingest of a multi-device ctx -------------------------------------------------------------------------------------
// Every shard tokenises a RANGE OF WINDOWS of both files on its own GPU (its own PCIe link, its own reader threads), then the
// shards line the two files up and route every entry to the shard that owns its cell:
//   1. parse: alt windows [k n_win / n, (k+1) n_win / n) -> (locus, cell, count) tokens, ref windows likewise -> counts; a
//      line's LOCAL index is the number of newlines before it inside the range, so its global index (= its position in the
//      zip, load_data.rs:190-204) is known once every shard's newline count is: exclusive sums over the ranks;
//   2. the two files' byte ranges do not cut at the same lines (their lines differ in length): every shard fetches the ref
//      counts of ITS alt lines from whichever shards tokenised them (contiguous pieces, peer copies);
//   3. zip + validation of its lines (same kernels as the single-device path), in place -> all-cells COO of those lines;
//   4. cut by owning cell range (order kept), one piece per destination shard;
//   5. every shard concatenates the pieces meant for it IN RANK ORDER = file order, which is what load_cell_data's per-cell
//      lists need (load_data.rs:151-174).
// The threads meet at a barrier between the steps; a failing shard releases the others (LocalGroup::fail).
struct MtxSplit {
    int n = 0;
    LocalGroup *bar = nullptr;
    uint64_t ma[CELLECTOR_MAX_SHARDS] = {}, mr[CELLECTOR_MAX_SHARDS] = {};
    unsigned long long bad_a[CELLECTOR_MAX_SHARDS] = {}, bad_r[CELLECTOR_MAX_SHARDS] = {}, bad_z[CELLECTOR_MAX_SHARDS] = {};
    uint32_t bad_kind[CELLECTOR_MAX_SHARDS] = {}, unsorted[CELLECTOR_MAX_SHARDS] = {};
    uint32_t first_locus[CELLECTOR_MAX_SHARDS] = {}, last_locus[CELLECTOR_MAX_SHARDS] = {};
    uint64_t lines[CELLECTOR_MAX_SHARDS] = {};
    uint32_t *r_dev[CELLECTOR_MAX_SHARDS] = {};
    int device[CELLECTOR_MAX_SHARDS] = {};
    uint32_t *pl[CELLECTOR_MAX_SHARDS][CELLECTOR_MAX_SHARDS] = {}, *pc[CELLECTOR_MAX_SHARDS][CELLECTOR_MAX_SHARDS] = {};
    uint16_t *pa[CELLECTOR_MAX_SHARDS][CELLECTOR_MAX_SHARDS] = {}, *pr[CELLECTOR_MAX_SHARDS][CELLECTOR_MAX_SHARDS] = {};
    uint64_t cnt[CELLECTOR_MAX_SHARDS][CELLECTOR_MAX_SHARDS] = {};
    bool balance = false;                             // cut the cells so that every shard gets about the same number of entries
    std::vector<uint32_t> hist[CELLECTOR_MAX_SHARDS];  // ... from every parser's entries per cell
};
MtxSplit *mtx_split_new(int n, LocalGroup *bar, bool balance)
{
    MtxSplit *S = new (std::nothrow) MtxSplit();
    if (S) { S->n = n; S->bar = bar; S->balance = balance; }
    return S;
}
void mtx_split_delete(MtxSplit *S) { delete S; }
// both files go through the windowed parser (big, or the parse_window option forces it): the split ingest needs that
bool mtx_input_windowed(const MtxInput *in, int64_t parse_window_opt)
{
    const uint64_t na = in->fa.size - in->off_a, nr = in->fr.size - in->off_r;
    return na > 0 && nr > 0 && (parse_window_opt > 0 || (na >= PW_MIN && nr >= PW_MIN));
}

static cellector_status copy_between(cellector_ctx *c, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes)
{
    if (!bytes) return CELLECTOR_OK;
    const hipError_t e = dev_copy_sync(c->stream, dst, dst_dev, src, src_dev, bytes);  // (c = the receiving shard, its device current)
    if (e != hipSuccess) return ctx_fail(c, CELLECTOR_EDEVICE, "copy between shards failed: %s", hipGetErrorString(e));
    return CELLECTOR_OK;
}

cellector_status ingest_stage_mtx_split(cellector_ctx *c, MtxInput *in, MtxSplit *S, int rank, uint64_t parse_window, uint32_t **o_locus,
                                        uint32_t **o_cell, uint16_t **o_alt, uint16_t **o_ref, uint64_t *o_n, bool *o_sorted)
{
    const int n = S->n;
    const uint64_t TL = in->total_loci, TC = in->total_cells;
    uint32_t *l1 = nullptr, *c1 = nullptr, *a = nullptr, *r = nullptr, *rk = nullptr, *flags = nullptr;
    uint16_t *alt16 = nullptr, *ref16 = nullptr;
    unsigned long long *bad = nullptr;
    uint64_t *keep = nullptr;
    auto cleanup = [&]() {
        dev_free(l1); dev_free(c1); dev_free(a); dev_free(r); dev_free(rk); dev_free(flags); dev_free(bad); dev_free(keep);
        dev_free(alt16); dev_free(ref16);
        for (int d = 0; d < n; d++) {  // (the pieces this shard cut for the others, if it got that far)
            dev_free(S->pl[rank][d]); dev_free(S->pc[rank][d]); dev_free(S->pa[rank][d]); dev_free(S->pr[rank][d]);
            S->pl[rank][d] = S->pc[rank][d] = nullptr; S->pa[rank][d] = S->pr[rank][d] = nullptr;
        }
    };
#define SCHK(expr)                     \
    do {                               \
        cellector_status s__ = (expr); \
        if (s__ != CELLECTOR_OK) {     \
            cleanup();                 \
            S->bar->fail();            \
            return s__;                \
        }                              \
    } while (0)
#define SBARRIER()                                                                                       \
    do {                                                                                                 \
        if (!S->bar->barrier()) {                                                                        \
            cleanup();                                                                                   \
            return ctx_fail(c, CELLECTOR_ECOMM, "another shard of this ctx failed during the ingest");   \
        }                                                                                                \
    } while (0)
    if (hipSetDevice(c->device) != hipSuccess) {  // (through the barrier group like every other failure: the peers must not wait)
        S->bar->fail();
        return ctx_fail(c, CELLECTOR_EDEVICE, "hipSetDevice(%d) failed", c->device);
    }
    S->device[rank] = c->device;
    const bool timing = rank == 0 && getenv("CELLECTOR_TIMING") != nullptr;  // phase wall times of shard 0 on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing]     split: %-26s %8.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    SCHK(dev_alloc(c, &bad, 3)); SCHK(dev_alloc(c, &flags, 4));
    unsigned long long h_bad[3] = {~0ull, ~0ull, ~0ull};
    uint32_t h_flags[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(bad, h_bad, sizeof h_bad, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(flags, h_flags, sizeof h_flags, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) SCHK(ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e)));
    // ---- 1. this shard's windows of both files
    // (n shards pin their upload buffers at the same time and the driver pins them one after the other: 4 x 3 x 256 MB took
    //  0.2-0.5 s per shard on one box — 4 logical shards, 2 x 2.9 GB: 1.1 s with 256 MB windows, 0.32 s with 16 MB)
    uint64_t win = std::max(in->fa.size - in->off_a, in->fr.size - in->off_r) / ((uint64_t)n * 32);
    win = std::min<uint64_t>(PW_SPLIT_WINDOW, std::max<uint64_t>(16ull << 20, win));
    if (const char *e_mb = getenv("CELLECTOR_SPLIT_WINDOW_MB")) win = (uint64_t)atoll(e_mb) << 20;  // (tools/split_check.sh)
    if (parse_window > 0) win = parse_window;
    if (win < 4 * NL_SEG) win = 4 * NL_SEG;
    win &= ~(uint64_t)(NL_SEG - 1);
    uint64_t ma = 0, mr = 0;
    {
        const uint64_t nb_a = in->fa.size - in->off_a, nb_r = in->fr.size - in->off_r;
        const uint64_t nw_a = (nb_a + win - 1) / win, nw_r = (nb_r + win - 1) / win;
        const uint64_t hint = in->nnz_hint ? in->nnz_hint / (uint64_t)n + in->nnz_hint / (uint64_t)(8 * n) + 1024 : 0;
        PwBuffers B;
        const uint64_t longest = std::max(nw_a, nw_r) / (uint64_t)n + 1;
        SCHK(B.make(c, win, (int)std::min<uint64_t>(PW_NB, std::max<uint64_t>(1, longest))));
        lap("upload buffers");
        // ranges of ceil(n_win / n) windows: rank 0 always holds window 0 (= line 0); late ranks of a short file may hold none
        const uint64_t pa = (nw_a + n - 1) / n, pr = (nw_r + n - 1) / n;
        SCHK((parse_windowed<true>(c, in->fa, in->off_a, B, hint ? hint : nb_a / (12 * (uint64_t)n) + 1024, &l1, &c1, &a, &ma, bad,
                                   std::min(nw_a, pa * rank), std::min(nw_a, pa * (rank + 1)))));
        lap("alt range (upload + tokens)");
        SCHK((parse_windowed<false>(c, in->fr, in->off_r, B, hint ? hint : nb_r / (12 * (uint64_t)n) + 1024, (uint32_t **)nullptr,
                                    (uint32_t **)nullptr, &r, &mr, bad + 1, std::min(nw_r, pr * rank), std::min(nw_r, pr * (rank + 1)))));
        lap("ref range (upload + tokens)");
    }
    e = hipMemcpyAsync(h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) SCHK(ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e)));
    S->ma[rank] = ma; S->mr[rank] = mr; S->bad_a[rank] = h_bad[0]; S->bad_r[rank] = h_bad[1]; S->r_dev[rank] = r;
    lap("release buffers");
    SBARRIER();
    lap("wait for the other shards");
    // ---- 2. global line numbers; the ref counts of this shard's alt lines
    uint64_t abase[CELLECTOR_MAX_SHARDS + 1] = {0}, rbase[CELLECTOR_MAX_SHARDS + 1] = {0};
    for (int k = 0; k < n; k++) { abase[k + 1] = abase[k] + S->ma[k]; rbase[k + 1] = rbase[k] + S->mr[k]; }
    const uint64_t nlines = std::min(abase[n], rbase[n]);  // izip!: stops at the shorter file
    {
        unsigned long long first = ~0ull;  // (a line beyond the shorter file is never read by the reference)
        for (int k = 0; k < n; k++) {
            if (S->bad_a[k] != ~0ull) first = std::min<unsigned long long>(first, abase[k] + S->bad_a[k]);
            if (S->bad_r[k] != ~0ull) first = std::min<unsigned long long>(first, rbase[k] + S->bad_r[k]);
        }
        if (first < nlines) {
            cleanup();
            (void)S->bar->barrier();  // (every shard sees the same numbers and leaves here: nobody is left waiting)
            return ctx_fail(c, CELLECTOR_EPARSE, "cannot parse mtx entry %llu (line %llu of the data section)", first, first + 1);
        }
    }
    // lines held by rank k of a file with bases b: global (b[k], b[k] + m[k]] and, for rank 0, line 0; local index = global - b[k]
    auto g_lo = [&](const uint64_t *b, int k) -> uint64_t { return k == 0 ? (uint64_t)0 : b[k] + 1; };
    auto g_hi = [&](const uint64_t *b, int k) -> uint64_t { return std::min<uint64_t>(nlines, b[k + 1] + 1); };  // exclusive
    const uint64_t glo = std::min(g_lo(abase, rank), nlines), ghi = std::max(glo, g_hi(abase, rank));
    const uint64_t count = ghi - glo, lo = count ? glo - abase[rank] : 0;  // (a rank whose alt range lies beyond a shorter ref file holds nothing)
    SCHK(dev_alloc(c, &rk, ma + 2));
    for (int k = 0; k < n; k++) {
        const uint64_t s0 = std::max(glo, g_lo(rbase, k)), s1 = std::min(ghi, g_hi(rbase, k));
        if (s0 >= s1) continue;
        SCHK(copy_between(c, rk + (s0 - abase[rank]), c->device, S->r_dev[k] + (s0 - rbase[k]), S->device[k], (s1 - s0) * sizeof(uint32_t)));
    }
    SBARRIER();  // (every shard has fetched what it needs out of the others' ref counts)
    lap("fetch ref counts");
    dev_free(r);
    S->r_dev[rank] = nullptr;
    // ---- 3. zip + validation of this shard's lines, in place
    hipLaunchKernelGGL(k_pair_check, dim3(pgrid(count + 1)), dim3(PB), 0, c->stream, count, l1 + lo, c1 + lo, a + lo, rk + lo, TL, TC,
                       (uint64_t)0, TC, (uint64_t *)nullptr, bad + 2, flags, flags + 1);
    uint32_t edge[2] = {0, 0};
    e = hipMemcpyAsync(h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_flags, flags, sizeof h_flags, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && count) e = hipMemcpyAsync(&edge[0], l1 + lo, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && count) e = hipMemcpyAsync(&edge[1], l1 + lo + count - 1, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) SCHK(ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e)));
    S->bad_z[rank] = h_bad[2] == ~0ull ? ~0ull : glo + h_bad[2];
    S->bad_kind[rank] = h_flags[0]; S->unsorted[rank] = h_flags[1];
    S->first_locus[rank] = edge[0]; S->last_locus[rank] = edge[1]; S->lines[rank] = count;
    SBARRIER();
    {
        int worst = -1;
        for (int k = 0; k < n; k++)
            if (S->bad_z[k] != ~0ull && (worst < 0 || S->bad_z[k] < S->bad_z[worst])) worst = k;
        if (worst >= 0) {
            static const char *what[] = {"", "index 0 (indices are 1-based)", "locus index out of range", "cell index out of range",
                                         "count above 65535 not supported"};
            cleanup();
            (void)S->bar->barrier();
            return ctx_fail(c, CELLECTOR_EINVAL, "mtx entry %llu: %s", S->bad_z[worst], what[S->bad_kind[worst] <= 4 ? S->bad_kind[worst] : 0]);
        }
        bool sorted = true;
        uint32_t prev_last = 0;
        bool have_prev = false;
        for (int k = 0; k < n; k++) {
            if (S->unsorted[k]) sorted = false;
            if (!S->lines[k]) continue;
            if (have_prev && S->first_locus[k] < prev_last) sorted = false;  // (1-based tokens on both sides)
            prev_last = S->last_locus[k];
            have_prev = true;
        }
        *o_sorted = sorted;
    }
    SCHK(dev_alloc(c, &alt16, count)); SCHK(dev_alloc(c, &ref16, count));
    if (count) hipLaunchKernelGGL(k_pair_take, dim3(pgrid(count)), dim3(PB), 0, c->stream, count, l1 + lo, c1 + lo, a + lo, rk + lo, alt16, ref16);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) SCHK(ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e)));
    // ---- 4. one piece per destination shard (cells of its range, order kept, cell index local to it)
    {
        // (ingest_split_coo reads the staged COO of a ctx: lend it the arrays for the calls; this ctx has staged nothing yet)
        c->coo_locus = l1 + lo; c->coo_cell = c1 + lo; c->coo_alt = alt16; c->coo_ref = ref16; c->coo_n = count;
        cellector_status st = dev_alloc(c, &keep, count + 1);
        if (S->balance && st == CELLECTOR_OK) {
            // nnz-balanced ranges: every parser counts its entries per cell, all of them add the n histograms up (the same
            // integers on every shard) and cut the cells where the running sum crosses k / n of the entries
            st = ingest_cell_histogram(c, c->coo_cell, count, TC, &S->hist[rank]);
            if (st != CELLECTOR_OK) { c->coo_locus = c->coo_cell = nullptr; c->coo_alt = c->coo_ref = nullptr; c->coo_n = 0; SCHK(st); }
            if (!S->bar->barrier()) {
                c->coo_locus = c->coo_cell = nullptr; c->coo_alt = c->coo_ref = nullptr; c->coo_n = 0;
                cleanup();
                return ctx_fail(c, CELLECTOR_ECOMM, "another shard of this ctx failed during the ingest");
            }
            std::vector<uint32_t> epc(TC, 0u);
            for (int k = 0; k < n; k++)
                for (uint64_t i = 0; i < TC; i++) epc[i] += S->hist[k][i];
            comm_balanced_bounds(epc.data(), TC, n, c->comm.bounds);
            c->comm.has_bounds = true;
        }
        for (int d = 0; d < n && st == CELLECTOR_OK; d++) {
            uint64_t cb, ce;
            comm_range(c->comm, TC, d, &cb, &ce);
            st = ingest_split_coo(c, cb, ce, keep, &S->pl[rank][d], &S->pc[rank][d], &S->pa[rank][d], &S->pr[rank][d], &S->cnt[rank][d]);
        }
        c->coo_locus = c->coo_cell = nullptr; c->coo_alt = c->coo_ref = nullptr; c->coo_n = 0;
        SCHK(st);
    }
    lap("zip + cut by owner");
    SBARRIER();
    // ---- 5. this shard's entries = the pieces meant for it, in rank order (= file order)
    uint64_t total = 0;
    for (int k = 0; k < n; k++) total += S->cnt[k][rank];
    uint32_t *fl = nullptr, *fc = nullptr;
    uint16_t *fa16 = nullptr, *fr16 = nullptr;
    cellector_status st = dev_alloc(c, &fl, total);
    if (st == CELLECTOR_OK) st = dev_alloc(c, &fc, total);
    if (st == CELLECTOR_OK) st = dev_alloc(c, &fa16, total);
    if (st == CELLECTOR_OK) st = dev_alloc(c, &fr16, total);
    uint64_t at = 0;
    for (int k = 0; k < n && st == CELLECTOR_OK; k++) {
        const uint64_t m = S->cnt[k][rank];
        st = copy_between(c, fl + at, c->device, S->pl[k][rank], S->device[k], m * 4);
        if (st == CELLECTOR_OK) st = copy_between(c, fc + at, c->device, S->pc[k][rank], S->device[k], m * 4);
        if (st == CELLECTOR_OK) st = copy_between(c, fa16 + at, c->device, S->pa[k][rank], S->device[k], m * 2);
        if (st == CELLECTOR_OK) st = copy_between(c, fr16 + at, c->device, S->pr[k][rank], S->device[k], m * 2);
        at += m;
    }
    if (st != CELLECTOR_OK) { dev_free(fl); dev_free(fc); dev_free(fa16); dev_free(fr16); SCHK(st); }
    if (!S->bar->barrier()) {  // (the pieces this shard made have been fetched by their destinations)
        dev_free(fl); dev_free(fc); dev_free(fa16); dev_free(fr16);
        cleanup();
        return ctx_fail(c, CELLECTOR_ECOMM, "another shard of this ctx failed during the ingest");
    }
    cleanup();
    lap("gather own pieces");
    *o_locus = fl; *o_cell = fc; *o_alt = fa16; *o_ref = fr16; *o_n = total;
#undef SCHK
#undef SBARRIER
    return CELLECTOR_OK;
}

// Stage this shard's entries of the alt/ref pair on the device; dims / shard range must already be set on the ctx.
// The ref file's count tokens parsed on ANOTHER ctx's device and stream (multi-device ingest: the two files are independent
// byte streams until they are zipped, so a second GPU takes the second file over its own PCIe link while the first one
// tokenises the alt file; two files together also read faster from the page cache than one: 76 vs 43 GB/s measured).
static cellector_status parse_ref_on(cellector_ctx *h, const FileBytes &fr, size_t off_r, uint64_t win, bool windowed, uint64_t nnz_hint,
                                     uint32_t **r_out, uint64_t *n_r, unsigned long long *bad_host)
{
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long *bad = nullptr;
    CHK(dev_alloc(h, &bad, 1));
    unsigned long long none = ~0ull;
    hipError_t e = hipMemcpyAsync(bad, &none, sizeof none, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cellector_status st = e == hipSuccess ? CELLECTOR_OK : ctx_fail(h, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e));
    if (st == CELLECTOR_OK) {
        if (windowed) {
            PwBuffers B;
            st = B.make(h, win, (int)std::min<uint64_t>(PW_NB, std::max<uint64_t>(1, (fr.size - off_r + win - 1) / win)));
            if (st == CELLECTOR_OK)
                st = parse_windowed<false>(h, fr, off_r, B, nnz_hint, (uint32_t **)nullptr, (uint32_t **)nullptr, r_out, n_r, bad);
        } else {
            st = parse_whole<false>(h, fr, off_r, (uint32_t **)nullptr, (uint32_t **)nullptr, r_out, n_r, bad);
        }
    }
    if (st == CELLECTOR_OK) {
        e = hipMemcpyAsync(bad_host, bad, sizeof none, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) st = ctx_fail(h, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e));
    }
    dev_free(bad);
    return st;
}

cellector_status ingest_stage_mtx_device(cellector_ctx *c, MtxInput *in, cellector_ctx *helper)
{
    FileBytes &fa = in->fa, &fr = in->fr;
    const size_t off_a = in->off_a, off_r = in->off_r;
    const bool timing = getenv("CELLECTOR_TIMING") != nullptr;  // phase wall times on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing]     %-22s %8.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    uint32_t *l1 = nullptr, *c1 = nullptr, *a = nullptr, *r = nullptr, *flags = nullptr;
    unsigned long long *bad = nullptr;  // [0] alt file, [1] ref file: smallest line that does not parse; [2] zip stage
    uint64_t *keep = nullptr;
    auto cleanup = [&]() {
        dev_free(l1); dev_free(c1); dev_free(a); dev_free(r); dev_free(flags); dev_free(bad); dev_free(keep);
    };
#define PCHK(expr)                     \
    do {                               \
        cellector_status s__ = (expr); \
        if (s__ != CELLECTOR_OK) {     \
            cleanup();                 \
            return s__;                \
        }                              \
    } while (0)
    HIPCHK(c, hipSetDevice(c->device));
    PCHK(dev_alloc(c, &bad, 3)); PCHK(dev_alloc(c, &flags, 4));
    unsigned long long h_bad[3] = {~0ull, ~0ull, ~0ull};
    uint32_t h_flags[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(bad, h_bad, sizeof h_bad, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(flags, h_flags, sizeof h_flags, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { cleanup(); return ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e)); }
    // a multi-GB file goes through the device in windows (option parse_window forces a window size: tests); the token
    // arrays' capacity comes from the size line's entry count (a hint only: the reference never reads it)
    uint64_t n_a = 0, n_r = 0;
    // window: 1/96 of the bigger file, 32 MB .. PW_WINDOW.  (Pinning and releasing 3 x 256 MB of upload buffers is 0.1-0.15 s
    // of a 0.39 s ingest of 2 x 2.9 GB: 32 MB windows take 0.245 s there; at 2 x 31 GB the window size makes no difference,
    // 256 MB stays.  tools/window_sweep.sh)
    uint64_t win = std::max(fa.size - off_a, fr.size - off_r) / 96;
    win = std::min<uint64_t>(PW_WINDOW, std::max<uint64_t>(32ull << 20, win));
    if (const char *e_mb = getenv("CELLECTOR_WINDOW_MB")) win = (uint64_t)atoll(e_mb) << 20;  // (the sweep)
    if (c->parse_window_opt > 0) win = (uint64_t)c->parse_window_opt;
    if (win < 4 * NL_SEG) win = 4 * NL_SEG;
    win &= ~(uint64_t)(NL_SEG - 1);
    const bool win_a = c->parse_window_opt > 0 || !fa.data || fa.size - off_a >= PW_MIN;
    const bool win_r = c->parse_window_opt > 0 || !fr.data || fr.size - off_r >= PW_MIN;
    if (helper) {  // the ref file on the helper's device, concurrently (its own ring of window buffers, its own stream)
        uint32_t *r_h = nullptr;
        unsigned long long bad_r = ~0ull;
        cellector_status st_r = CELLECTOR_OK;
        std::thread th([&] { st_r = parse_ref_on(helper, fr, off_r, win, win_r, in->nnz_hint, &r_h, &n_r, &bad_r); });
        cellector_status st_a = CELLECTOR_OK;
        {
            PwBuffers B;
            if (win_a) st_a = B.make(c, win, (int)std::min<uint64_t>(PW_NB, std::max<uint64_t>(1, (fa.size - off_a + win - 1) / win)));
            if (st_a == CELLECTOR_OK)
                st_a = win_a ? parse_windowed<true>(c, fa, off_a, B, in->nnz_hint, &l1, &c1, &a, &n_a, bad)
                             : parse_whole<true>(c, fa, off_a, &l1, &c1, &a, &n_a, bad);
        }
        th.join();
        (void)hipSetDevice(c->device);
        if (st_a == CELLECTOR_OK && st_r != CELLECTOR_OK) st_a = ctx_fail(c, st_r, "%s", helper->err.c_str());
        if (st_a == CELLECTOR_OK) st_a = dev_alloc(c, &r, n_r);
        if (st_a == CELLECTOR_OK && n_r && dev_copy_sync(c->stream, r, c->device, r_h, helper->device, n_r * sizeof(uint32_t)) != hipSuccess)
            st_a = ctx_fail(c, CELLECTOR_EDEVICE, "peer copy of the ref counts failed: %s", hipGetErrorString(hipGetLastError()));
        if (st_a == CELLECTOR_OK &&
            (hipMemcpyAsync(bad + 1, &bad_r, sizeof bad_r, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
             hipStreamSynchronize(c->stream) != hipSuccess))
            st_a = ctx_fail(c, CELLECTOR_EDEVICE, "parse: copy failed");
        dev_free(r_h);
        PCHK(st_a);
        lap("alt + ref files (two devices)");
    } else {
        // While the host reads and the device tokenises (host-bound: ~1.5 s for 2 x 31 GB), a helper thread maps the VRAM the
        // CSC / CSR build will ask for right afterwards — three arrays of 8 bytes per entry, sized from the header's entry
        // count — and parks the blocks in the caching layer: mapping fresh VRAM costs 10-60 ms per GB at this footprint and
        // would otherwise be paid serially behind the parse (0.5-1.5 s of the 1M x 200k ingest).  Unused blocks go back at the
        // end of the ingest (dev_cache_trim).
        std::thread premap;
        struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } premap_joiner{premap};
        {
            const char *env = getenv("CELLECTOR_PREMAP");
            const bool all = c->cell_begin == 0 && c->cell_end >= c->total_cells;
            const uint64_t hint = in->nnz_hint <= (fa.size - off_a) / 6 ? in->nnz_hint : 0;  // (a line has at least 6 bytes: else the header lies)
            if (win_a && all && !c->ingest_all_cells && hint >= (1ull << 26) && (!env || atoi(env) != 0)) {  // (not for a multi-device ctx's parser: its shards build smaller matrices)
                const int dev = c->device;
                premap = std::thread([dev, hint] {
                    if (hipSetDevice(dev) != hipSuccess) return;
                    // (a little more than 8 bytes per entry: the 4-byte token arrays, asked for meanwhile, must not match
                    //  these blocks — the caching layer hands out blocks of up to twice the request)
                    // in the order they are asked for: the ref file's count tokens (4 B per entry, as parse_windowed sizes
                    // them), the two 16-bit count arrays of the staged entries, then the CSC / CSR build's three arrays
                    const size_t sizes[6] = {((size_t)hint + 16) * 4, (size_t)hint * 2 + 4096, (size_t)hint * 2 + 4096,
                                             (size_t)hint * 8 + (1u << 16), (size_t)hint * 8 + (1u << 16), (size_t)hint * 8 + (1u << 16)};
                    for (size_t bytes : sizes) {
                        void *p = nullptr;
                        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return; }
                        dev_cache_park(p, bytes, dev);
                    }
                });
            }
        }
        PwBuffers B;  // one ring of window buffers for both files
        if (win_a || win_r) {
            const uint64_t longest = std::max(win_a ? fa.size - off_a : 0, win_r ? fr.size - off_r : 0);
            PCHK(B.make(c, win, (int)std::min<uint64_t>(PW_NB, std::max<uint64_t>(1, (longest + win - 1) / win))));
        }
        if (win_a) PCHK((parse_windowed<true>(c, fa, off_a, B, in->nnz_hint, &l1, &c1, &a, &n_a, bad)));
        else PCHK((parse_whole<true>(c, fa, off_a, &l1, &c1, &a, &n_a, bad)));
        lap("alt file (upload + tokens)");
        if (win_r)
            PCHK((parse_windowed<false>(c, fr, off_r, B, in->nnz_hint, (uint32_t **)nullptr, (uint32_t **)nullptr, &r, &n_r, bad + 1)));
        else
            PCHK((parse_whole<false>(c, fr, off_r, (uint32_t **)nullptr, (uint32_t **)nullptr, &r, &n_r, bad + 1)));
        lap("ref file (upload + tokens)");
        if (premap.joinable()) premap.join();
        lap("wait for the pre-mapped blocks");
    }
    const uint64_t n = std::min(n_a, n_r);  // izip!: stops at the shorter file
    e = hipMemcpyAsync(h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { cleanup(); return ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e)); }
    {   // (a line beyond the shorter file is never read by the reference: only failures among the first n count)
        const unsigned long long first = std::min(h_bad[0], h_bad[1]);
        if (first < n) {
            cleanup();
            return ctx_fail(c, CELLECTOR_EPARSE, "cannot parse mtx entry %llu (line %llu of the data section)", first, first + 1);
        }
    }
    const bool all_cells = c->cell_begin == 0 && c->cell_end >= c->total_cells;
    if (!all_cells) PCHK(dev_alloc(c, &keep, n + 1));
    hipLaunchKernelGGL(k_pair_check, dim3(pgrid(n + 1)), dim3(PB), 0, c->stream, n, l1, c1, a, r, c->total_loci, c->total_cells,
                       c->cell_begin, c->cell_end, keep, bad + 2, flags, flags + 1);
    // the validation result is read BEHIND k_pair_check on the ctx's stream (a caller-supplied non-blocking stream does
    // not order against null-stream copies: the range check could be read before the kernel ran)
    e = hipMemcpyAsync(h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_flags, flags, sizeof h_flags, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { cleanup(); return ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e)); }
    if (h_bad[2] != ~0ull) {
        cleanup();
        static const char *what[] = {"", "index 0 (indices are 1-based)", "locus index out of range", "cell index out of range",
                                     "count above 65535 not supported"};
        return ctx_fail(c, CELLECTOR_EINVAL, "mtx entry %llu: %s", h_bad[2], what[h_flags[0] <= 4 ? h_flags[0] : 0]);
    }
    c->coo_sorted = h_flags[1] == 0;
    if (all_cells) {
        c->coo_n = n;
        PCHK(dev_alloc(c, &c->coo_alt, n)); PCHK(dev_alloc(c, &c->coo_ref, n));
        if (n) hipLaunchKernelGGL(k_pair_take, dim3(pgrid(n)), dim3(PB), 0, c->stream, n, l1, c1, a, r, c->coo_alt, c->coo_ref);
        c->coo_locus = l1;
        c->coo_cell = c1;
        l1 = c1 = nullptr;  // (owned by the ctx now)
    } else {
        uint64_t kept = 0;
        PCHK(dev_exclusive_scan_u64(c, keep, n + 1, &kept));
        c->coo_n = kept;
        PCHK(dev_alloc(c, &c->coo_locus, kept)); PCHK(dev_alloc(c, &c->coo_cell, kept));
        PCHK(dev_alloc(c, &c->coo_alt, kept)); PCHK(dev_alloc(c, &c->coo_ref, kept));
        if (n)
            hipLaunchKernelGGL(k_pair_fill, dim3(pgrid(n)), dim3(PB), 0, c->stream, n, l1, c1, a, r, c->cell_begin, c->cell_end,
                               keep, c->coo_locus, c->coo_cell, c->coo_alt, c->coo_ref);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    lap("tokenise + zip + filter");
    cleanup();
    if (e != hipSuccess) return ctx_fail(c, CELLECTOR_EDEVICE, "parse: %s", hipGetErrorString(e));
#undef PCHK
    return CELLECTOR_OK;
}
