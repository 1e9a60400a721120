// Host-side reader of the vartrix alt.mtx / ref.mtx pair: the text contract of
// reader (load_data.rs:240-251), consume_mtx_header (:206-223) and read_mtx_lines (:190-204):
//   * ".gz" by file extension (multi-member), plain text otherwise;
//   * exactly three header lines per file, dims = first two tokens of the REF file's third line;
//   * data lines are zipped pairwise until the shorter file ends; locus, cell (1-based) and the alt
//     count come from the alt line, the ref count from the ref line's third token (its indices are
//     never read); tokens must parse as unsigned integers ("1.0" is an error, like parse::<usize>()).
// One pass over the text; the three passes of the reference become device kernels over the staged COO.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstring>

#include "ctx.h"

namespace {

struct LineSource {
    // plain
    const char *map = nullptr;
    size_t map_len = 0, pos = 0;
    int fd = -1;
    // gz
    gzFile gz = nullptr;
    std::vector<char> buf;
    size_t b_beg = 0, b_end = 0;
    bool gz_eof = false;
    std::string cur;  // gz mode: the current line

    bool open(const char *path)
    {
        const size_t n = strlen(path);
        if (n >= 3 && strcmp(path + n - 3, ".gz") == 0) {
            gz = gzopen(path, "rb");
            if (!gz) return false;
            gzbuffer(gz, 1 << 20);
            buf.resize(1 << 20);
            return true;
        }
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        map_len = (size_t)st.st_size;
        if (map_len) {
            void *m = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) return false;
            map = (const char *)m;
            madvise((void *)map, map_len, MADV_SEQUENTIAL);
        }
        return true;
    }
    ~LineSource()
    {
        if (map) munmap((void *)map, map_len);
        if (fd >= 0) ::close(fd);
        if (gz) gzclose(gz);
    }
    // next line without its terminator; false at end of input (BufRead::lines semantics)
    bool next(const char *&b, const char *&e)
    {
        if (!gz) {
            if (pos >= map_len) return false;
            const char *s = map + pos;
            const char *nl = (const char *)memchr(s, '\n', map_len - pos);
            const char *end = nl ? nl : map + map_len;
            pos = (size_t)(end - map) + (nl ? 1 : 0);
            b = s;
            e = end;
            return true;
        }
        cur.clear();
        for (;;) {
            if (b_beg < b_end) {
                const char *s = buf.data() + b_beg;
                const char *nl = (const char *)memchr(s, '\n', b_end - b_beg);
                if (nl) {
                    cur.append(s, (size_t)(nl - s));
                    b_beg = (size_t)(nl - buf.data()) + 1;
                    b = cur.data();
                    e = b + cur.size();
                    return true;
                }
                cur.append(s, b_end - b_beg);
                b_beg = b_end;
            }
            if (gz_eof) {
                if (cur.empty()) return false;
                b = cur.data();
                e = b + cur.size();
                return true;
            }
            const int got = gzread(gz, buf.data(), (unsigned)buf.size());
            if (got <= 0) {
                gz_eof = true;
                continue;
            }
            b_beg = 0;
            b_end = (size_t)got;
        }
    }
};

inline bool is_ws(char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n' || ch == '\f' || ch == '\v'; }

// split_whitespace + tokens[idx].parse::<usize>()
inline bool tok_u64(const char *b, const char *e, int idx, uint64_t *out)
{
    const char *p = b;
    for (int t = 0;; t++) {
        while (p < e && is_ws(*p)) p++;
        if (p >= e) return false;
        const char *s = p;
        while (p < e && !is_ws(*p)) p++;
        if (t == idx) {
            if (*s == '+') s++;
            if (s == p) return false;
            uint64_t v = 0;
            for (; s < p; s++) {
                if (*s < '0' || *s > '9') return false;
                v = v * 10 + (uint64_t)(*s - '0');
            }
            *out = v;
            return true;
        }
    }
}

// parse the first three whitespace-separated unsigned tokens of a line in one sweep
inline bool three_u64(const char *b, const char *e, uint64_t v[3])
{
    const char *p = b;
    for (int t = 0; t < 3; t++) {
        while (p < e && is_ws(*p)) p++;
        if (p >= e) return false;
        if (*p == '+') p++;
        if (p >= e || *p < '0' || *p > '9') return false;
        uint64_t x = 0;
        while (p < e && *p >= '0' && *p <= '9') x = x * 10 + (uint64_t)(*p++ - '0');
        if (p < e && !is_ws(*p)) return false;
        v[t] = x;
    }
    return true;
}

}  // namespace

cellector_status read_mtx_pair(const cellector_ctx *c, const char *alt_path, const char *ref_path, HostCoo *out)
{
    LineSource fa, fr;
    if (!fa.open(alt_path)) return ctx_fail(c, CELLECTOR_EIO, "couldn't open file %s", alt_path);
    if (!fr.open(ref_path)) return ctx_fail(c, CELLECTOR_EIO, "couldn't open file %s", ref_path);
    const char *ab, *ae, *rb, *re;
    for (int x = 0; x < 3; x++) {
        fa.next(ab, ae);
        const bool have = fr.next(rb, re);
        if (x == 2) {
            if (!have || !tok_u64(rb, re, 0, &out->total_loci) || !tok_u64(rb, re, 1, &out->total_cells))
                return ctx_fail(c, CELLECTOR_EPARSE, "cannot parse the matrix market size line of %s", ref_path);
        }
    }
    uint64_t line = 0;
    for (;;) {
        if (!fa.next(ab, ae)) break;
        if (!fr.next(rb, re)) break;
        uint64_t a[3];
        // the ref line's first two tokens are never parsed by the reference (only tokens[2])
        uint64_t rcount;
        if (!three_u64(ab, ae, a) || !tok_u64(rb, re, 2, &rcount) || a[0] == 0 || a[1] == 0)
            return ctx_fail(c, CELLECTOR_EPARSE, "cannot parse mtx entry %llu (alt '%.*s' / ref '%.*s')",
                            (unsigned long long)line, (int)(ae - ab), ab, (int)(re - rb), rb);
        if (a[0] - 1 > 0xfffffffeull || a[1] - 1 > 0xfffffffeull || a[2] > 0xffffffffull || rcount > 0xffffffffull)
            return ctx_fail(c, CELLECTOR_EINVAL, "mtx entry %llu: value too large", (unsigned long long)line);
        out->locus.push_back((uint32_t)(a[0] - 1));
        out->cell.push_back((uint32_t)(a[1] - 1));
        out->alt.push_back((uint32_t)a[2]);
        out->ref.push_back((uint32_t)rcount);
        line++;
    }
    return CELLECTOR_OK;
}
