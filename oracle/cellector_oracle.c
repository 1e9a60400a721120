/*
 * cellector_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 * See cellector_oracle.h for scope and the "PARITY UNPINNED" caveat.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/cellector/src/).  Operation order follows the reference so
 * that f64 rounding matches a glibc-linked Rust build (Rust's f64::ln/exp/
 * log10 resolve to the platform libm).
 */
#define _GNU_SOURCE
#include "cellector_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ========================================================================= */
/* statrs 0.16.0 restatements                         */
/* ========================================================================= */

static const double GAMMA_R = 10.900511;
static const double GAMMA_DK[11] = {
    2.48574089138753565546e-5,  1.05142378581721974210,   -3.45687097222016235469,
    4.51227709466894823700,     -2.98285225323576655721,  1.05639711577126713077,
    -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
    4.63399473359905636708e-6,  -2.71994908488607703910e-9};
static const double LN_2_SQRT_E_OVER_PI = 0.6207822376352452223455184457816472122518527279025978;
static const double LN_PI = 1.1447298858494001741434273513530587116472948129153;
#define ORC_PI 3.14159265358979323846264338327950288
#define ORC_E 2.71828182845904523536028747135266250

/* statrs::function::gamma::ln_gamma — Appendix B.1; call sites stats.rs:49-51 */
double orc_ln_gamma(double x)
{
    if (x < 0.5) {
        double s = GAMMA_DK[0];
        for (int i = 1; i <= 10; i++) s += GAMMA_DK[i] / ((double)i - x);
        return LN_PI - log(sin(ORC_PI * x)) - log(s) - LN_2_SQRT_E_OVER_PI -
               (0.5 - x) * log((0.5 - x + GAMMA_R) / ORC_E);
    }
    double s = GAMMA_DK[0];
    for (int i = 1; i <= 10; i++) s += GAMMA_DK[i] / (x + (double)i - 1.0);
    return log(s) + LN_2_SQRT_E_OVER_PI + (x - 0.5) * log((x - 0.5 + GAMMA_R) / ORC_E);
}

/* statrs::function::factorial — Appendix B.2 (FCACHE = f64 running product, 0..=170) */
static double FCACHE[171];
static int fcache_ready = 0;
static void fcache_init(void)
{
    if (fcache_ready) return;
    FCACHE[0] = 1.0;
    for (int i = 1; i <= 170; i++) FCACHE[i] = FCACHE[i - 1] * (double)i;
    fcache_ready = 1;
}

double orc_ln_factorial(uint64_t x)
{
    fcache_init();
    if (x <= 170) return log(FCACHE[x]);
    return orc_ln_gamma((double)x + 1.0);
}

/* call sites stats.rs:15,60; load_data.rs:163 */
double orc_ln_binomial(uint64_t n, uint64_t k)
{
    if (k > n) return -INFINITY;
    return orc_ln_factorial(n) - orc_ln_factorial(k) - orc_ln_factorial(n - k);
}

static int cmp_f64(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* Data::new(v): a sorted private copy gives the exact order statistics that
 * statrs' select_inplace (quickselect) returns. */
static double *sorted_copy(const double *x, size_t n)
{
    double *s = (double *)malloc((n ? n : 1) * sizeof(double));
    memcpy(s, x, n * sizeof(double));
    qsort(s, n, sizeof(double), cmp_f64);
    return s;
}

/* Data::median — Appendix B.3; call sites main.rs:325,443 */
double orc_median(const double *x, size_t n)
{
    if (n == 0) return NAN; /* select_inplace(0) -> min() of empty -> NaN */
    double *s = sorted_copy(x, n);
    size_t k = n / 2;
    double r = (n % 2 != 0) ? s[k] : (s[k ? k - 1 : 0] + s[k]) / 2.0;
    free(s);
    return r;
}

/* Data::quantile (R-8) — Appendix B.3; lower/upper_quartile at main.rs:326-327 */
double orc_quantile(const double *x, size_t n, double tau)
{
    if (!(tau >= 0.0 && tau <= 1.0) || n == 0) return NAN;
    double *s = sorted_copy(x, n);
    double h = ((double)n + 1.0 / 3.0) * tau + 1.0 / 3.0;
    int64_t hf = (int64_t)h;
    double r;
    if (hf <= 0 || tau == 0.0) r = s[0];
    else if (hf >= (int64_t)n || tau == 1.0) r = s[n - 1];
    else {
        double a = s[hf - 1], b = s[hf];
        r = a + (h - (double)hf) * (b - a);
    }
    free(s);
    return r;
}

/* statrs::distribution::Binomial::pmf — Appendix B.4; call sites main.rs:92-97 */
double orc_binomial_pmf(double p, uint64_t n, uint64_t k)
{
    if (k > n) return 0.0;
    if (p == 0.0) return k == 0 ? 1.0 : 0.0;
    if (fabs(p - 1.0) <= 4 * 2.220446049250313e-16) return k == n ? 1.0 : 0.0;
    return exp(orc_ln_binomial(n, k) + (double)k * log(p) + (double)(n - k) * log(1.0 - p));
}

/* ========================================================================= */
/* stats.rs                                                                  */
/* ========================================================================= */

/* stats.rs:35-39 */
double orc_logsumexp(double a, double b)
{
    double m = fmax(a, b);
    double sum = exp(a - m) + exp(b - m);
    return m + log(sum);
}

/* stats.rs:48-53 */
double orc_log_beta_calc(double a, double b)
{
    double lga = orc_ln_gamma(a);
    double lgb = orc_ln_gamma(b);
    double lgab = orc_ln_gamma(a + b);
    return lga + lgb - lgab;
}

/* stats.rs:41-46 */
double orc_log_beta_binomial_pmf(double alt, double ref, double alpha, double beta, double lnc)
{
    double num = orc_log_beta_calc(alt + alpha, ref + beta);
    double den = orc_log_beta_calc(alpha, beta);
    return lnc + num - den;
}

/* stats.rs:55-65 (max_n = 100, load_data.rs:149-150) */
#define LBC_MAX_N 100
static double LBC[LBC_MAX_N + 1][LBC_MAX_N + 1];
static int lbc_ready = 0;
static void lbc_init(void)
{
    if (lbc_ready) return;
    for (int n = 0; n <= LBC_MAX_N; n++)
        for (int k = 0; k <= n; k++) LBC[n][k] = orc_ln_binomial((uint64_t)n, (uint64_t)k);
    lbc_ready = 1;
}

/* stats.rs:8-33 */
void orc_expected_log_beta_binomial_pmf(size_t total, double alpha, double beta, double *expected,
                                        double *variance)
{
    lbc_init();
    double stackbuf[64];
    double *lls = total + 1 <= 64 ? stackbuf : (double *)malloc((total + 1) * sizeof(double));
    for (size_t k = 0; k <= total; k++) {
        double lbc = (total < LBC_MAX_N + 1) ? LBC[total][k]
                                              : orc_ln_binomial((uint64_t)total, (uint64_t)k);
        lls[k] = orc_log_beta_binomial_pmf((double)k, (double)(total - k), alpha, beta, lbc);
    }
    double e = 2.0 * lls[0];
    for (size_t k = 1; k <= total; k++) e = orc_logsumexp(e, 2.0 * lls[k]);
    double var = 0.0;
    for (size_t k = 0; k <= total; k++) {
        double d = lls[k] - e;
        var += exp(lls[k]) * (d * d);
    }
    if (lls != stackbuf) free(lls);
    *expected = e;
    if (variance) *variance = var;
}

/* ========================================================================= */
/* data model (load_data.rs)                                                 */
/* ========================================================================= */

/* load_data.rs:13-20 CellLocusData (48 bytes, AoS like the reference) */
typedef struct {
    size_t locus_index;
    size_t locus_id;
    double log_binomial_coefficient;
    double alt_count;
    double ref_count;
    size_t total;
} cell_locus;

/* main.rs:527-539 PMFData (88 bytes) */
typedef struct {
    size_t cell_id, locus_index, locus;
    uint64_t excluded; /* bool, padded */
    double log_pmf;
    size_t alt_count, ref_count;
    double alpha, beta, expected_log_pmf, expected_log_variance;
} pmf_data;

struct orc_ctx {
    uint64_t total_loci, total_cells, L, nnz;
    uint64_t *row_ptr;   /* [N+1] per-cell lists in file order */
    cell_locus *entries; /* [nnz] */
    double *locus_counts; /* [L][2] = {ref, alt}  (load_data.rs:157-158) */
    uint64_t *locus_ids;  /* [L] */
    /* loop state */
    uint8_t *loci_used; /* [L] */
    uint8_t *excluded;  /* [N] */
    uint64_t iteration;
    /* last-iteration outputs */
    double *ll, *ell, *evar, *nloci, *norm; /* [N] */
    double *c_min, *c_maj;                  /* [L] */
    uint64_t *n_min, *n_maj, *a_min, *r_min, *a_maj, *r_maj; /* [L] */
    pmf_data *pmfs;
    uint64_t n_pmfs, cap_pmfs;
};

static void *xcalloc(size_t n, size_t sz)
{
    void *p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

static void alloc_state(orc_ctx *c)
{
    uint64_t N = c->total_cells, L = c->L;
    c->loci_used = (uint8_t *)xcalloc(L, 1);
    memset(c->loci_used, 1, L); /* load_data.rs:176-179 */
    c->excluded = (uint8_t *)xcalloc(N, 1);
    c->ll = (double *)xcalloc(N, 8);  c->ell = (double *)xcalloc(N, 8);
    c->evar = (double *)xcalloc(N, 8); c->nloci = (double *)xcalloc(N, 8);
    c->norm = (double *)xcalloc(N, 8);
    c->c_min = (double *)xcalloc(L, 8); c->c_maj = (double *)xcalloc(L, 8);
    c->n_min = (uint64_t *)xcalloc(L, 8); c->n_maj = (uint64_t *)xcalloc(L, 8);
    c->a_min = (uint64_t *)xcalloc(L, 8); c->r_min = (uint64_t *)xcalloc(L, 8);
    c->a_maj = (uint64_t *)xcalloc(L, 8); c->r_maj = (uint64_t *)xcalloc(L, 8);
}

void orc_free(orc_ctx *c)
{
    if (!c) return;
    free(c->row_ptr); free(c->entries); free(c->locus_counts); free(c->locus_ids);
    free(c->loci_used); free(c->excluded);
    free(c->ll); free(c->ell); free(c->evar); free(c->nloci); free(c->norm);
    free(c->c_min); free(c->c_maj); free(c->n_min); free(c->n_maj);
    free(c->a_min); free(c->r_min); free(c->a_maj); free(c->r_maj);
    free(c->pmfs);
    free(c);
}

/* ln C(n,k) per entry: load_data.rs:159-164 */
static double entry_lnc(uint64_t alt, uint64_t ref)
{
    lbc_init();
    uint64_t n = alt + ref;
    if (n <= LBC_MAX_N) return LBC[n][alt];
    return orc_ln_binomial(n, alt);
}

/* get_loci_used (load_data.rs:254-280) + load_cell_data (load_data.rs:134-181)
 * over in-memory COO triplets in file order. */
orc_ctx *orc_from_coo(uint64_t total_loci, uint64_t total_cells, uint64_t nnz,
                      const uint32_t *locus0, const uint32_t *cell0, const uint32_t *alt,
                      const uint32_t *ref, uint64_t min_alt, uint64_t min_ref, char *err,
                      size_t errlen)
{
    /* PASS 1: per-locus number of cells with ref>0 / alt>0 (load_data.rs:265-270) */
    uint64_t *cnt = (uint64_t *)xcalloc(total_loci * 2, 8);
    for (uint64_t i = 0; i < nnz; i++) {
        if (locus0[i] >= total_loci) {
            if (err) snprintf(err, errlen, "locus index %u out of range (total_loci %llu)",
                              locus0[i] + 1, (unsigned long long)total_loci);
            free(cnt);
            return NULL;
        }
        if (ref[i] > 0) cnt[2 * (uint64_t)locus0[i] + 0] += 1;
        if (alt[i] > 0) cnt[2 * (uint64_t)locus0[i] + 1] += 1;
    }
    /* filter + compaction (load_data.rs:271-278) */
    uint64_t *to_used = (uint64_t *)xcalloc(total_loci, 8);
    uint64_t L = 0;
    for (uint64_t l = 0; l < total_loci; l++) {
        if (cnt[2 * l] >= min_ref && cnt[2 * l + 1] >= min_alt) to_used[l] = L++;
        else to_used[l] = UINT64_MAX;
    }
    free(cnt);

    orc_ctx *c = (orc_ctx *)xcalloc(1, sizeof(orc_ctx));
    c->total_loci = total_loci; c->total_cells = total_cells; c->L = L;
    c->locus_ids = (uint64_t *)xcalloc(L, 8);
    for (uint64_t l = 0; l < total_loci; l++)
        if (to_used[l] != UINT64_MAX) c->locus_ids[to_used[l]] = l; /* load_data.rs:144-147 */
    c->locus_counts = (double *)xcalloc(L * 2, 8);

    /* PASS 2 (load_data.rs:151-174): per-cell lists in file order. */
    c->row_ptr = (uint64_t *)xcalloc(total_cells + 1, 8);
    uint64_t used_nnz = 0;
    for (uint64_t i = 0; i < nnz; i++) {
        if (to_used[locus0[i]] == UINT64_MAX) continue;
        if (cell0[i] >= total_cells) {
            if (err) snprintf(err, errlen, "cell index %u out of range (total_cells %llu)",
                              cell0[i] + 1, (unsigned long long)total_cells);
            free(to_used); orc_free(c);
            return NULL;
        }
        c->row_ptr[cell0[i] + 1]++;
        used_nnz++;
    }
    for (uint64_t i = 0; i < total_cells; i++) c->row_ptr[i + 1] += c->row_ptr[i];
    c->nnz = used_nnz;
    c->entries = (cell_locus *)xcalloc(used_nnz, sizeof(cell_locus));
    uint64_t *cursor = (uint64_t *)xcalloc(total_cells, 8);
    for (uint64_t i = 0; i < nnz; i++) {
        uint64_t u = to_used[locus0[i]];
        if (u == UINT64_MAX) continue;
        c->locus_counts[2 * u + 0] += (double)ref[i];
        c->locus_counts[2 * u + 1] += (double)alt[i];
        cell_locus *e = &c->entries[c->row_ptr[cell0[i]] + cursor[cell0[i]]++];
        e->locus_index = u; e->locus_id = locus0[i];
        e->alt_count = (double)alt[i]; e->ref_count = (double)ref[i];
        e->total = (size_t)alt[i] + (size_t)ref[i];
        e->log_binomial_coefficient = entry_lnc(alt[i], ref[i]);
    }
    free(cursor); free(to_used);
    alloc_state(c);
    return c;
}

/* ---- text ingest: reader (load_data.rs:240-251), consume_mtx_header
 * (:206-223), read_mtx_lines (:190-204) ------------------------------------ */
typedef struct { gzFile gz; char *buf; size_t cap; } linereader;

static int lr_open(linereader *r, const char *path)
{
    /* gzopen reads plain files transparently and concatenated members like
     * MultiGzDecoder; the reference switches on the ".gz" extension. */
    r->gz = gzopen(path, "rb");
    if (!r->gz) return -1;
    gzbuffer(r->gz, 1 << 20);
    r->cap = 256; r->buf = (char *)malloc(r->cap);
    return 0;
}
static void lr_close(linereader *r) { if (r->gz) gzclose(r->gz); free(r->buf); }
/* returns 1 and a NUL-terminated line (newline stripped), or 0 at EOF */
static int lr_line(linereader *r)
{
    size_t len = 0;
    for (;;) {
        if (!gzgets(r->gz, r->buf + len, (int)(r->cap - len))) return len > 0;
        len += strlen(r->buf + len);
        if (len && r->buf[len - 1] == '\n') { r->buf[--len] = 0; break; }
        if (len + 1 < r->cap) return 1; /* EOF without newline */
        r->cap *= 2; r->buf = (char *)realloc(r->buf, r->cap);
    }
    if (len && r->buf[len - 1] == '\r') r->buf[len - 1] = 0;
    return 1;
}
/* split_whitespace + nth token parse::<usize>(); returns 0 on failure */
static int tok_usize(const char *line, int idx, uint64_t *out)
{
    const char *p = line;
    for (int t = 0;; t++) {
        while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\f' || *p == '\v') p++;
        if (!*p) return 0;
        const char *s = p;
        while (*p && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\f' || *p == '\v')) p++;
        if (t == idx) {
            if (*s == '+') s++;
            if (s == p) return 0;
            uint64_t v = 0;
            for (; s < p; s++) {
                if (*s < '0' || *s > '9') return 0;
                v = v * 10 + (uint64_t)(*s - '0');
            }
            *out = v;
            return 1;
        }
    }
}

orc_ctx *orc_load_mtx(const char *alt_path, const char *ref_path, uint64_t min_alt,
                      uint64_t min_ref, char *err, size_t errlen)
{
    linereader ra = {0}, rr = {0};
    if (lr_open(&ra, alt_path)) { if (err) snprintf(err, errlen, "couldn't open file %s", alt_path); return NULL; }
    if (lr_open(&rr, ref_path)) { if (err) snprintf(err, errlen, "couldn't open file %s", ref_path); lr_close(&ra); return NULL; }
    uint64_t total_loci = 0, total_cells = 0;
    int ok = 1;
    for (int x = 0; x < 3 && ok; x++) { /* load_data.rs:212-221: dims from the REF file's 3rd line */
        lr_line(&ra);
        if (!lr_line(&rr)) rr.buf[0] = 0;
        if (x == 2) ok = tok_usize(rr.buf, 0, &total_loci) && tok_usize(rr.buf, 1, &total_cells);
    }
    if (!ok) { if (err) snprintf(err, errlen, "cannot parse matrix market size line of %s", ref_path); lr_close(&ra); lr_close(&rr); return NULL; }
    size_t cap = 1 << 16, n = 0;
    uint32_t *lo = (uint32_t *)malloc(cap * 4), *ce = (uint32_t *)malloc(cap * 4);
    uint32_t *al = (uint32_t *)malloc(cap * 4), *re = (uint32_t *)malloc(cap * 4);
    while (lr_line(&ra) && lr_line(&rr)) { /* izip!: stops at the shorter file */
        uint64_t l, cidx, a, r;
        /* load_data.rs:194-197: indices from the alt line, ref line gives only the count */
        if (!tok_usize(ra.buf, 0, &l) || !tok_usize(rr.buf, 2, &r) || !tok_usize(ra.buf, 2, &a) ||
            !tok_usize(ra.buf, 1, &cidx) || l == 0 || cidx == 0) {
            if (err) snprintf(err, errlen, "cannot parse mtx entry %zu: '%s' / '%s'", n, ra.buf, rr.buf);
            free(lo); free(ce); free(al); free(re); lr_close(&ra); lr_close(&rr);
            return NULL;
        }
        if (n == cap) {
            cap *= 2;
            lo = (uint32_t *)realloc(lo, cap * 4); ce = (uint32_t *)realloc(ce, cap * 4);
            al = (uint32_t *)realloc(al, cap * 4); re = (uint32_t *)realloc(re, cap * 4);
        }
        lo[n] = (uint32_t)(l - 1); ce[n] = (uint32_t)(cidx - 1); al[n] = (uint32_t)a; re[n] = (uint32_t)r;
        n++;
    }
    lr_close(&ra); lr_close(&rr);
    orc_ctx *c = orc_from_coo(total_loci, total_cells, n, lo, ce, al, re, min_alt, min_ref, err, errlen);
    free(lo); free(ce); free(al); free(re);
    return c;
}

orc_ctx *orc_from_csr(uint64_t n_cells, uint64_t n_loci, const uint64_t *row_ptr,
                      const uint64_t *packed, const double *locus_counts)
{
    orc_ctx *c = (orc_ctx *)xcalloc(1, sizeof(orc_ctx));
    c->total_loci = n_loci; c->total_cells = n_cells; c->L = n_loci; c->nnz = row_ptr[n_cells];
    c->row_ptr = (uint64_t *)xcalloc(n_cells + 1, 8);
    memcpy(c->row_ptr, row_ptr, (n_cells + 1) * 8);
    c->locus_ids = (uint64_t *)xcalloc(n_loci, 8);
    for (uint64_t l = 0; l < n_loci; l++) c->locus_ids[l] = l;
    c->locus_counts = (double *)xcalloc(n_loci * 2, 8);
    memcpy(c->locus_counts, locus_counts, n_loci * 16);
    c->entries = (cell_locus *)xcalloc(c->nnz, sizeof(cell_locus));
    for (uint64_t i = 0; i < c->nnz; i++) {
        uint64_t p = packed[i];
        uint64_t l = p & 0xffffffffu, a = (p >> 32) & 0xffff, r = p >> 48;
        cell_locus *e = &c->entries[i];
        e->locus_index = l; e->locus_id = l; e->alt_count = (double)a; e->ref_count = (double)r;
        e->total = a + r; e->log_binomial_coefficient = entry_lnc(a, r);
    }
    alloc_state(c);
    return c;
}

void orc_dims(const orc_ctx *c, uint64_t *tc, uint64_t *tl, uint64_t *lu, uint64_t *nnz)
{
    if (tc) *tc = c->total_cells;
    if (tl) *tl = c->total_loci;
    if (lu) *lu = c->L;
    if (nnz) *nnz = c->nnz;
}
void orc_locus_ids(const orc_ctx *c, uint64_t *out) { memcpy(out, c->locus_ids, c->L * 8); }
void orc_locus_counts(const orc_ctx *c, double *out) { memcpy(out, c->locus_counts, c->L * 16); }
void orc_row_ptr(const orc_ctx *c, uint64_t *out) { memcpy(out, c->row_ptr, (c->total_cells + 1) * 8); }
void orc_entries_per_cell(const orc_ctx *c, uint32_t *out)
{
    for (uint64_t i = 0; i < c->total_cells; i++) out[i] = (uint32_t)(c->row_ptr[i + 1] - c->row_ptr[i]);
}
void orc_entries(const orc_ctx *c, uint32_t *li, uint32_t *alt, uint32_t *ref, double *lnc)
{
    for (uint64_t i = 0; i < c->nnz; i++) {
        if (li) li[i] = (uint32_t)c->entries[i].locus_index;
        if (alt) alt[i] = (uint32_t)c->entries[i].alt_count;
        if (ref) ref[i] = (uint32_t)c->entries[i].ref_count;
        if (lnc) lnc[i] = c->entries[i].log_binomial_coefficient;
    }
}

/* ========================================================================= */
/* scoring loop (main.rs)                                                    */
/* ========================================================================= */

/* init_alpha_betas — main.rs:598-611 */
static void init_alpha_betas(const orc_ctx *c, const uint8_t *excluded, double *alpha, double *beta)
{
    for (uint64_t l = 0; l < c->L; l++) {
        alpha[l] = c->locus_counts[2 * l + 1] + 1.0;
        beta[l] = c->locus_counts[2 * l + 0] + 1.0;
    }
    for (uint64_t cell = 0; cell < c->total_cells; cell++) {
        if (!excluded[cell]) continue;
        for (uint64_t i = c->row_ptr[cell]; i < c->row_ptr[cell + 1]; i++) {
            const cell_locus *e = &c->entries[i];
            alpha[e->locus_index] -= e->alt_count;
            beta[e->locus_index] -= e->ref_count;
        }
    }
}

void orc_alpha_betas(const orc_ctx *c, double *alpha, double *beta)
{
    init_alpha_betas(c, c->excluded, alpha, beta);
}

static void pmfs_reserve(orc_ctx *c, uint64_t n)
{
    if (c->cap_pmfs >= n) return;
    free(c->pmfs);
    c->pmfs = (pmf_data *)malloc((n ? n : 1) * sizeof(pmf_data));
    if (!c->pmfs) { fprintf(stderr, "oracle: out of memory (all_pmfs)\n"); abort(); }
    c->cap_pmfs = n;
}

/* Worker threads for the per-cell loop below.  Default 1: the reference is single-threaded and the
 * cpu_baseline leg times one core.  Cells are independent and each cell's sums stay sequential in file
 * order, so any thread count gives bit-identical results; the large parity tests use all host cores. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int orc_get_threads(void) { return g_threads; }

/* get_cell_log_likelihoods — main.rs:541-591 */
static void get_cell_log_likelihoods(orc_ctx *c, const uint8_t *loci_used, const double *alpha,
                                     const double *beta, const uint8_t *excluded, double *ll,
                                     double *ell, double *evar, double *nloci)
{
    pmfs_reserve(c, c->nnz);
    lbc_init(); /* static tables are built before the loop (no lazy init inside worker threads) */
    const uint64_t N = c->total_cells;
    /* position of each cell's first PMFData in all_pmfs (the reference pushes them in cell order) */
    uint64_t *first = (uint64_t *)xcalloc(N + 1, 8);
    for (uint64_t cell = 0; cell < N; cell++) {
        uint64_t used = 0;
        for (uint64_t i = c->row_ptr[cell]; i < c->row_ptr[cell + 1]; i++)
            used += loci_used[c->entries[i].locus_index] ? 1 : 0;
        first[cell + 1] = first[cell] + used;
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads) if (g_threads > 1)
#endif
    for (uint64_t cell = 0; cell < N; cell++) {
        double log_likelihood = 0.0, expected_ll = 0.0, expected_var = 0.0, used = 0.0;
        uint64_t np = first[cell];
        for (uint64_t i = c->row_ptr[cell]; i < c->row_ptr[cell + 1]; i++) {
            const cell_locus *e = &c->entries[i];
            if (!loci_used[e->locus_index]) continue;
            double a = alpha[e->locus_index], b = beta[e->locus_index];
            double log_pmf = orc_log_beta_binomial_pmf(e->alt_count, e->ref_count, a, b,
                                                       e->log_binomial_coefficient);
            log_likelihood += log_pmf;
            double ex, var;
            orc_expected_log_beta_binomial_pmf(e->total, a, b, &ex, &var);
            expected_ll += ex;
            expected_var += var;
            pmf_data *p = &c->pmfs[np++];
            p->log_pmf = log_pmf; p->cell_id = cell; p->excluded = excluded ? excluded[cell] : 0;
            p->locus = e->locus_id; p->locus_index = e->locus_index;
            p->alt_count = (size_t)e->alt_count; p->ref_count = (size_t)e->ref_count;
            p->alpha = a; p->beta = b; p->expected_log_pmf = ex; p->expected_log_variance = var;
            used += 1.0;
        }
        ll[cell] = log_likelihood; nloci[cell] = used;
        if (ell) ell[cell] = expected_ll;
        if (evar) evar[cell] = expected_var;
    }
    c->n_pmfs = first[N];
    free(first);
}

void orc_cell_log_likelihoods(orc_ctx *c, const double *alpha, const double *beta,
                              const uint8_t *mask, double *ll, double *expected_ll,
                              double *loci_used_per_cell)
{
    get_cell_log_likelihoods(c, mask, alpha, beta, c->excluded, ll, expected_ll, NULL,
                             loci_used_per_cell);
}

/* get_locus_log_likelihoods — main.rs:368-420 */
static void get_locus_log_likelihoods(orc_ctx *c, const uint8_t *new_excluded)
{
    uint64_t L = c->L;
    memset(c->c_min, 0, L * 8); memset(c->c_maj, 0, L * 8);
    memset(c->n_min, 0, L * 8); memset(c->n_maj, 0, L * 8);
    memset(c->a_min, 0, L * 8); memset(c->r_min, 0, L * 8);
    memset(c->a_maj, 0, L * 8); memset(c->r_maj, 0, L * 8);
    for (uint64_t i = 0; i < c->n_pmfs; i++) {
        const pmf_data *p = &c->pmfs[i];
        if (new_excluded[p->cell_id]) {
            c->c_min[p->locus_index] += p->log_pmf; c->n_min[p->locus_index] += 1;
            c->r_min[p->locus_index] += p->ref_count; c->a_min[p->locus_index] += p->alt_count;
        } else {
            c->c_maj[p->locus_index] += p->log_pmf; c->n_maj[p->locus_index] += 1;
            c->r_maj[p->locus_index] += p->ref_count; c->a_maj[p->locus_index] += p->alt_count;
        }
    }
}

/* compute_new_excluded — main.rs:308-347, with the locus filter of
 * locus_filter_and_output_locus_data — main.rs:428-451 */
void orc_em_iteration(orc_ctx *c, double iqr_multiple, orc_iter_summary *out)
{
    uint64_t N = c->total_cells, L = c->L;
    double *alpha = (double *)xcalloc(L, 8), *beta = (double *)xcalloc(L, 8);
    init_alpha_betas(c, c->excluded, alpha, beta);                           /* main.rs:309 */
    get_cell_log_likelihoods(c, c->loci_used, alpha, beta, c->excluded, c->ll, c->ell, c->evar,
                             c->nloci);                                       /* main.rs:312 */
    for (uint64_t i = 0; i < N; i++)                                          /* main.rs:314-323 */
        c->norm[i] = c->nloci[i] > 0.0 ? c->ll[i] / c->nloci[i] : 0.0;
    double median = orc_median(c->norm, N);                                   /* main.rs:325 */
    double q1 = orc_quantile(c->norm, N, 0.25), q3 = orc_quantile(c->norm, N, 0.75);
    double iqr = q3 - q1;
    double threshold = q1 - iqr_multiple * iqr;                               /* main.rs:329 */
    uint8_t *new_excluded = (uint8_t *)xcalloc(N, 1);
    uint64_t n_new = 0, n_resc = 0;
    for (uint64_t i = 0; i < N; i++) {
        new_excluded[i] = c->norm[i] < threshold;                             /* main.rs:331 */
        if (new_excluded[i] && !c->excluded[i]) n_new++;
        if (!new_excluded[i] && c->excluded[i]) n_resc++;
    }
    get_locus_log_likelihoods(c, new_excluded);                               /* main.rs:343 */
    /* locus filter, main.rs:428-451 */
    double *sub = (double *)xcalloc(L, 8);
    size_t nsub = 0;
    uint64_t nfilt = 0;
    for (uint64_t l = 0; l < L; l++)
        if (c->n_min[l] != 0) sub[nsub++] = c->c_min[l] / (double)c->n_min[l];
    double lmedian = orc_median(sub, nsub);
    for (uint64_t l = 0; l < L; l++) {
        double per_cell = c->n_min[l] != 0 ? c->c_min[l] / (double)c->n_min[l] : 0.0;
        if (per_cell < -80.0) { c->loci_used[l] = 0; nfilt++; }               /* main.rs:444-447 */
    }
    free(sub);
    memcpy(c->excluded, new_excluded, N);
    free(new_excluded); free(alpha); free(beta);
    c->iteration++;
    if (out) {
        out->any_change = (n_new > 0 || n_resc > 0);                          /* main.rs:335 */
        out->n_new_excluded = n_new; out->n_rescued = n_resc; out->n_loci_filtered = nfilt;
        out->median = median; out->iqr = iqr; out->threshold = threshold;
        out->locus_median = lmedian;
    }
}

#define COPY_IF(dst, src, n) do { if (dst) memcpy(dst, src, (n) * 8); } while (0)
void orc_iter_cell_outputs(const orc_ctx *c, double *ll, double *ell, double *nloci, double *norm)
{
    COPY_IF(ll, c->ll, c->total_cells); COPY_IF(ell, c->ell, c->total_cells);
    COPY_IF(nloci, c->nloci, c->total_cells); COPY_IF(norm, c->norm, c->total_cells);
}
void orc_iter_locus_outputs(const orc_ctx *c, double *cmin, double *cmaj, uint64_t *nmin,
                            uint64_t *nmaj, uint64_t *amin, uint64_t *rmin, uint64_t *amaj,
                            uint64_t *rmaj)
{
    COPY_IF(cmin, c->c_min, c->L); COPY_IF(cmaj, c->c_maj, c->L);
    COPY_IF(nmin, c->n_min, c->L); COPY_IF(nmaj, c->n_maj, c->L);
    COPY_IF(amin, c->a_min, c->L); COPY_IF(rmin, c->r_min, c->L);
    COPY_IF(amaj, c->a_maj, c->L); COPY_IF(rmaj, c->r_maj, c->L);
}
void orc_loci_mask(const orc_ctx *c, uint8_t *out) { memcpy(out, c->loci_used, c->L); }
void orc_excluded(const orc_ctx *c, uint8_t *out) { memcpy(out, c->excluded, c->total_cells); }
void orc_set_excluded(orc_ctx *c, const uint8_t *in) { memcpy(c->excluded, in, c->total_cells); }

/* get_cell_log_likelihoods + get_locus_log_likelihoods under caller alpha/beta/mask and a caller exclusion set:
 * the per-locus half of compute_new_excluded (main.rs:312,343) for a shard of cells (multi-shard host tests). */
void orc_locus_stats(orc_ctx *c, const double *alpha, const double *beta, const uint8_t *mask,
                     const uint8_t *new_excluded, double *cmin, double *cmaj, uint64_t *nmin, uint64_t *nmaj,
                     uint64_t *amin, uint64_t *rmin, uint64_t *amaj, uint64_t *rmaj)
{
    uint64_t N = c->total_cells;
    double *ll = (double *)xcalloc(N, 8), *nl = (double *)xcalloc(N, 8);
    get_cell_log_likelihoods(c, mask, alpha, beta, c->excluded, ll, NULL, NULL, nl);
    get_locus_log_likelihoods(c, new_excluded);
    orc_iter_locus_outputs(c, cmin, cmaj, nmin, nmaj, amin, rmin, amaj, rmaj);
    free(ll); free(nl);
}

/* calculate_posteriors — main.rs:228-280 (get_loci_used_for_posterior_calc,
 * main.rs:282-306, returns all-true: quirk Q1) */
void orc_posteriors(orc_ctx *c, double *posterior, double *doublet_posterior, double *ll_majority,
                    double *ll_minority)
{
    uint64_t N = c->total_cells, L = c->L;
    uint8_t *included = (uint8_t *)xcalloc(N, 1);
    uint64_t n_excl = 0;
    for (uint64_t i = 0; i < N; i++) { included[i] = !c->excluded[i]; n_excl += c->excluded[i]; }
    double *a_maj = (double *)xcalloc(L, 8), *b_maj = (double *)xcalloc(L, 8);
    double *a_min = (double *)xcalloc(L, 8), *b_min = (double *)xcalloc(L, 8);
    double *a_dbl = (double *)xcalloc(L, 8), *b_dbl = (double *)xcalloc(L, 8);
    init_alpha_betas(c, c->excluded, a_maj, b_maj);                           /* main.rs:239 */
    double mf = ((double)n_excl + 1.0) / ((double)N + 1.0);                   /* main.rs:240 */
    init_alpha_betas(c, included, a_min, b_min);                              /* main.rs:241 */
    for (uint64_t l = 0; l < L; l++) {                                        /* main.rs:244-248 */
        a_dbl[l] = (a_maj[l] - 1.0) * mf + (a_min[l] - 1.0) + 1.0;
        b_dbl[l] = (b_maj[l] - 1.0) * mf + (b_min[l] - 1.0) + 1.0;
    }
    mf = fmax(mf, 0.01);                                                      /* main.rs:250 */
    for (uint64_t l = 0; l < L; l++) {                                        /* main.rs:251-254 */
        a_maj[l] = (a_maj[l] - 1.0) * mf + 1.0;
        b_maj[l] = (b_maj[l] - 1.0) * mf + 1.0;
    }
    uint8_t *all = (uint8_t *)xcalloc(L, 1);
    memset(all, 1, L);                                                        /* main.rs:301-303 */
    double *l_min = (double *)xcalloc(N, 8), *l_maj = (double *)xcalloc(N, 8);
    double *l_dbl = (double *)xcalloc(N, 8), *scratch = (double *)xcalloc(N, 8);
    double *scratch2 = (double *)xcalloc(N, 8);
    get_cell_log_likelihoods(c, all, a_min, b_min, c->excluded, l_min, scratch2, NULL, scratch);
    get_cell_log_likelihoods(c, all, a_maj, b_maj, included, l_maj, scratch2, NULL, scratch);
    get_cell_log_likelihoods(c, all, a_dbl, b_dbl, included, l_dbl, scratch2, NULL, scratch);
    double log_prior_doublet = log((double)N / 1000.0 / 100.0 * fmax(mf, 0.1)); /* main.rs:259 */
    double log_prior_minority = log(mf);
    double log_prior_majority = log(1.0 - mf);
    for (uint64_t i = 0; i < N; i++) {                                        /* main.rs:266-278 */
        double log_num = log_prior_minority + l_min[i];
        double log_den = orc_logsumexp(log_num, log_prior_majority + l_maj[i]);
        double log_dbl_num = log_prior_doublet + l_dbl[i];
        log_den = orc_logsumexp(log_den, log_dbl_num);
        posterior[i] = exp(log_num - log_den);
        doublet_posterior[i] = exp(log_dbl_num - log_den);
        if (ll_majority) ll_majority[i] = l_maj[i];
        if (ll_minority) ll_minority[i] = l_min[i];
    }
    free(included); free(a_maj); free(b_maj); free(a_min); free(b_min); free(a_dbl); free(b_dbl);
    free(all); free(l_min); free(l_maj); free(l_dbl); free(scratch); free(scratch2);
}

/* output_final_assignments rule — main.rs:141-171 */
void orc_assignments(const orc_ctx *c, const double *posterior, const double *doublet_posterior,
                     double posterior_threshold, uint64_t min_loci_used, uint8_t *pa, uint8_t *aa,
                     uint64_t *qual)
{
    for (uint64_t i = 0; i < c->total_cells; i++) {
        uint8_t a = 3;
        if (posterior[i] > posterior_threshold) a = 0;
        else if (1.0 - posterior[i] > posterior_threshold) a = 1;
        if (doublet_posterior[i] > 0.5) a = 2;
        if (c->row_ptr[i + 1] - c->row_ptr[i] < min_loci_used) a = 3;        /* main.rs:153, quirk Q5 */
        pa[i] = a;
        aa[i] = c->excluded[i] ? 0 : 1;                                      /* main.rs:161-163 */
        double post = fmax(posterior[i], 1.0 - posterior[i]);
        double q = fmin(-10.0 * log10(1.0 - post), 255.0);                   /* f64::min ignores NaN */
        qual[i] = (q != q || q < 0.0) ? 0 : (uint64_t)q;                     /* `as usize` saturates */
    }
}

/* load_mtx_final — load_data.rs:109-132 (over ALL loci, by final exclusion set) */
void orc_final_tallies_coo(uint64_t total_loci, uint64_t nnz, const uint32_t *locus0,
                           const uint32_t *cell0, const uint32_t *alt, const uint32_t *ref,
                           const uint8_t *excluded, uint64_t *alt_min, uint64_t *ref_min,
                           uint64_t *alt_maj, uint64_t *ref_maj)
{
    memset(alt_min, 0, total_loci * 8); memset(ref_min, 0, total_loci * 8);
    memset(alt_maj, 0, total_loci * 8); memset(ref_maj, 0, total_loci * 8);
    for (uint64_t i = 0; i < nnz; i++) {
        if (excluded[cell0[i]]) { alt_min[locus0[i]] += alt[i]; ref_min[locus0[i]] += ref[i]; }
        else { alt_maj[locus0[i]] += alt[i]; ref_maj[locus0[i]] += ref[i]; }
    }
}

/* genotype call of output_final_vcf — main.rs:79-124 */
void orc_vcf_genotype(uint64_t minority_alt, uint64_t minority_ref, uint64_t majority_alt,
                      uint64_t majority_ref, uint8_t *gt_maj, double *maxpost_maj, uint8_t *gt_min,
                      double *maxpost_min)
{
    const double ambient_percent = 0.03, gt_threshold = 0.99;
    uint64_t total_alt = minority_alt + majority_alt, total_ref = minority_ref + majority_ref;
    double soup_frac = 0.5;
    if (total_alt + total_ref > 0) soup_frac = (double)total_alt / (double)(total_alt + total_ref);
    double p_hom_alt = (1.0 - ambient_percent) * 0.99 + ambient_percent * soup_frac;
    double p_het = (1.0 - ambient_percent) * 0.5 + ambient_percent * soup_frac;
    double p_hom_ref = (1.0 - ambient_percent) * 0.01 + ambient_percent * soup_frac;
    for (int which = 0; which < 2; which++) {
        uint64_t a = which ? majority_alt : minority_alt, r = which ? majority_ref : minority_ref;
        double l_alt = orc_binomial_pmf(p_hom_alt, a + r, a);
        double l_het = orc_binomial_pmf(p_het, a + r, a);
        double l_ref = orc_binomial_pmf(p_hom_ref, a + r, a);
        double denom = 1.0 / 3.0 * l_alt + 1.0 / 3.0 * l_het + 1.0 / 3.0 * l_ref;
        double p_alt = l_alt * 1.0 / 3.0 / denom;
        double p_hetp = l_het * 1.0 / 3.0 / denom;
        double p_ref = l_ref * 1.0 / 3.0 / denom;
        double mx = fmax(fmax(p_alt, p_hetp), p_ref);
        uint8_t gt = 0;
        if (p_alt > gt_threshold) gt = 1;
        else if (p_hetp > gt_threshold) gt = 2;
        else if (p_ref > gt_threshold) gt = 3;
        if (which) { *gt_maj = gt; *maxpost_maj = mx; } else { *gt_min = gt; *maxpost_min = mx; }
    }
}
