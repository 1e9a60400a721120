/* Fast MatrixMarket text writer for the benchmark tools (tools/ingest_bench.py): `locus cell count` lines, 1-based.
 * Built on the fly with gcc; not part of the product. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline char *put_u64(char *p, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

/* returns 0 on success */
int fastmtx_write(const char *path, const char *header, uint64_t n, const int64_t *locus1, const int64_t *cell1,
                  const int64_t *val)
{
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    fputs(header, f);
    const size_t cap = 1u << 24;
    char *buf = (char *)malloc(cap + 128);
    if (!buf) { fclose(f); return 2; }
    char *p = buf;
    for (uint64_t i = 0; i < n; i++) {
        p = put_u64(p, (uint64_t)locus1[i]); *p++ = ' ';
        p = put_u64(p, (uint64_t)cell1[i]); *p++ = ' ';
        p = put_u64(p, (uint64_t)val[i]); *p++ = '\n';
        if ((size_t)(p - buf) > cap) { fwrite(buf, 1, (size_t)(p - buf), f); p = buf; }
    }
    fwrite(buf, 1, (size_t)(p - buf), f);
    free(buf);
    return fclose(f) != 0;
}
