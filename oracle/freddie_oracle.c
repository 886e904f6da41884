/*
 * ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's canonical-segmentation algorithm
 * (reference: py/freddie_segment.py, function segment() :738-844 and its helpers).
 * It exists so that the HIP path can be checked against something that follows the
 * reference step by step.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (freddie_amd/) never does.
 *
 * Parity status: the reference ships no tests or golden vectors of its own
 * ("parity unpinned" by the reference).  This oracle is pinned
 * instead against outputs of the reference itself, run in the build container
 * (tests/golden/make_golden.py -> the fixtures under tests/golden, checked by
 * tests/test_oracle_golden.py, plus a live differential test when /root/reference
 * is present).
 *
 * Third-party numerics restated here (scipy 1.15.3 / numpy 2.2.6 are not vendored by
 * the reference; behaviour recovered by black-box probes):
 *   scipy.ndimage.gaussian_filter1d  -> fo_gaussian
 *   scipy.signal.find_peaks          -> fo_local_maxima, fo_select_by_distance
 *   numpy sum/mean/std               -> fo_np_sum
 * Gaussian weights and the smoothed-threshold table are taken as DATA from the caller
 * (they come out of numpy's exp / Python's round in the reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double sigma;
    double threshold_rate;
    double variance_factor;
    int32_t max_problem_size;
    int32_t min_read_support_outside;
    int32_t ignore_ends;
    int32_t radius_main;        /* int(4.0*sigma+0.5) */
    const double *w_main;       /* centre first, radius_main+1 values */
    int32_t radius_refine;      /* int(1.0*sigma+0.5) */
    const double *w_refine;
    int32_t h_len;
    const double *h_table;      /* smooth_threshold(threshold_rate) */
} fo_params;

typedef struct {
    int32_t K, R;
    int64_t P;
    int64_t *pos_off;           /* K+1 */
    double *Y_raw, *Y;          /* P */
    double threshold;
    int64_t n_vals;             /* number of Y>0 values */
    int64_t *cand_off;          /* K+1 */
    int32_t *cands;             /* y indices */
    int64_t *fixed_off;         /* K+1 */
    int32_t *fixed;             /* candidate indices (after break_large_problems) */
    int64_t n_problems;
    int32_t *prob_interval, *prob_start, *prob_end, *prob_nchain; /* chain = number of triples on the backtrack */
    int64_t *finalc_off;        /* K+1 */
    int32_t *finalc;            /* candidate indices chosen by run_optimize */
    int64_t *refine_off;        /* K+1 */
    int32_t *refine;            /* y indices added by refine_segmentation */
    int64_t *final_off;         /* K+1 */
    int32_t *final_y;           /* final y indices per interval (sorted) */
    int32_t *final_pos;         /* genomic */
    int64_t F;
    uint8_t *labels;            /* R x (F-1), values 0/1/2 */
    int32_t error;              /* nonzero: a reference assert would have fired */
    char errmsg[256];
} fo_result;

static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return p; }
static void *xcalloc(size_t n, size_t s) { void *p = calloc(n ? n : 1, s); if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return p; }

/* ---- scipy.ndimage.gaussian_filter1d (call sites py/freddie_segment.py:755, :260) -------
 * correlate1d with a symmetric kernel: out[l] = x[l]*w[0], then for j = radius..1:
 * out += (x[l-j] + x[l+j]) * w[j]   (farthest pair first; separate multiply and add).
 * mode 0 = 'reflect' (d c b a | a b c d | d c b a), mode 1 = 'constant' cval 0. */
static inline int64_t reflect_idx(int64_t i, int64_t n) {
    if (n == 1) return 0;
    int64_t p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}
#if defined(__GNUC__)
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
#endif
void fo_gaussian(const double *x, int64_t n, const double *w, int32_t radius, int32_t mode, double *out) {
    for (int64_t l = 0; l < n; ++l) {
        double t = x[l] * w[0];
        for (int32_t j = radius; j >= 1; --j) {
            int64_t a = l - j, b = l + j;
            double xa, xb;
            if (mode == 0) { xa = x[reflect_idx(a, n)]; xb = x[reflect_idx(b, n)]; }
            else { xa = a >= 0 ? x[a] : 0.0; xb = b < n ? x[b] : 0.0; }
            double s = xa + xb;
            double m = s * w[j];
            t = t + m;
        }
        out[l] = t;
    }
}

/* ---- numpy add.reduce on a contiguous f64 array (py/freddie_segment.py:758-759) ----------
 * buffered in chunks of 8192; each chunk by pairwise summation with an unrolled 8-lane leaf. */
static double pairwise(const double *a, int64_t n) {
    if (n < 8) { double r = 0.0; for (int64_t i = 0; i < n; ++i) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2; n2 -= n2 % 8;
    return pairwise(a, n2) + pairwise(a + n2, n - n2);
}
double fo_np_sum(const double *a, int64_t n) {
    double res = 0.0; int first = 1;
    for (int64_t s = 0; s < n; s += 8192) {
        int64_t m = n - s < 8192 ? n - s : 8192;
        double v = pairwise(a + s, m);
        if (first) { res = v; first = 0; } else res = res + v;
    }
    return res;
}
/* mean + vf*std over the Y>0 values in (interval, position) order; empty -> NaN */
double fo_variance_threshold(const double *Y, int64_t P, double vf, int64_t *n_out) {
    double *v = (double *)xmalloc(sizeof(double) * (size_t)P);
    int64_t n = 0;
    for (int64_t i = 0; i < P; ++i) if (Y[i] > 0) v[n++] = Y[i];
    if (n_out) *n_out = n;
    if (n == 0) { free(v); return NAN; }
    double mean = fo_np_sum(v, n) / (double)n;
    for (int64_t i = 0; i < n; ++i) { double d = v[i] - mean; v[i] = d * d; }
    double var = fo_np_sum(v, n) / (double)n;
    free(v);
    double sd = sqrt(var);
    double t = vf * sd;
    return mean + t;
}
#if defined(__GNUC__)
#pragma GCC pop_options
#endif

/* ---- scipy.signal.find_peaks(y) with no options = _local_maxima_1d (py/freddie_segment.py:616) */
int64_t fo_local_maxima(const double *x, int64_t n, int32_t *out) {
    int64_t m = 0, i = 1, i_max = n - 1;
    while (i < i_max) {
        if (x[i - 1] < x[i]) {
            int64_t ia = i + 1;
            while (ia < i_max && x[ia] == x[i]) ++ia;
            if (x[ia] < x[i]) {
                out[m++] = (int32_t)((i + ia - 1) / 2);
                i = ia;
            }
        }
        ++i;
    }
    return m;
}
/* find_peaks(..., distance=d): _select_by_peak_distance; priority = height, processed from the
 * highest down, ties resolved as a stable ascending argsort iterated from the end (the later
 * peak wins).  keep[] is 0/1 on return. */
void fo_select_by_distance(const int32_t *peaks, const double *prio, int64_t m, int32_t distance, uint8_t *keep) {
    int64_t *order = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)m);
    for (int64_t i = 0; i < m; ++i) order[i] = i;
    /* stable insertion sort ascending by priority */
    for (int64_t i = 1; i < m; ++i) {
        int64_t v = order[i], j = i - 1;
        while (j >= 0 && prio[order[j]] > prio[v]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = v;
    }
    for (int64_t i = 0; i < m; ++i) keep[i] = 1;
    for (int64_t i = m - 1; i >= 0; --i) {
        int64_t j = order[i];
        if (!keep[j]) continue;
        int64_t k = j - 1;
        while (k >= 0 && peaks[j] - peaks[k] < distance) { keep[k] = 0; --k; }
        k = j + 1;
        while (k < m && peaks[k] - peaks[j] < distance) { keep[k] = 0; ++k; }
    }
    free(order);
}

/* bisect.bisect_right(a, x, lo) */
static int64_t bisect_right(const int32_t *a, int64_t n, int32_t x, int64_t lo) {
    int64_t hi = n;
    while (lo < hi) { int64_t mid = (lo + hi) / 2; if (x < a[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}

static void set_err(fo_result *r, const char *msg) {
    if (!r->error) { r->error = 1; snprintf(r->errmsg, sizeof r->errmsg, "%s", msg); }
}

/* ---- get_cumulative_coverage (py/freddie_segment.py:188-246) --------------------------------
 * C is (n_c+1) x R uint32, row-major.  Exons are given as y indices of interval k. */
static uint32_t *cumulative_coverage(fo_result *res, int32_t R, const int64_t *rep_exon_off, const int32_t *ex_iv,
                                     const int32_t *ex_ys, const int32_t *ex_ye, int32_t k,
                                     const int32_t *cands, int64_t nc) {
    uint32_t *C = (uint32_t *)xcalloc((size_t)(nc + 1) * (size_t)R, sizeof(uint32_t));
    for (int32_t r = 0; r < R; ++r) {
        for (int64_t e = rep_exon_off[r]; e < rep_exon_off[r + 1]; ++e) {
            if (ex_iv[e] != k) continue;
            int32_t ys = ex_ys[e], ye = ex_ye[e];
            int64_t cs = bisect_right(cands, nc, ys, 0);          /* :207-210 */
            int64_t ce = bisect_right(cands, nc, ye, cs);         /* :212-216 */
            if (!(0 < cs && cs <= ce && ce <= nc)) { set_err(res, "coverage: C index out of range (:223)"); continue; }
            if (cs == ce) { C[cs * R + r] += (uint32_t)(ye - ys + 1); continue; }   /* :224-226 */
            C[cs * R + r] += (uint32_t)(cands[cs] - ys);          /* :228,:233 */
            C[ce * R + r] += (uint32_t)(ye - cands[ce - 1] + 1);  /* :230,:235 */
            for (int64_t c = cs + 1; c < ce; ++c) C[c * R + r] += (uint32_t)(cands[c] - cands[c - 1]); /* :237-240 */
        }
    }
    for (int64_t c = 1; c <= nc; ++c)                              /* :244-245 */
        for (int32_t r = 0; r < R; ++r) C[c * R + r] += C[(c - 1) * R + r];
    return C;
}

static inline double high_threshold(const fo_params *p, int64_t seg_len) {   /* get_high_threshold :269-274 */
    return seg_len < p->h_len ? p->h_table[seg_len] : p->threshold_rate;
}

/* ---- optimize (py/freddie_segment.py:475-568) ----------------------------------------------- */
#define NEG_INF INT64_MIN
typedef struct {
    int32_t start, end, n, R;
    int64_t words;
    const int32_t *cands;
    uint64_t *yea, *nay;          /* [n*n][words], pair (i,j) relative indices */
    const int32_t *W; int unit_w;
    int64_t *in_mem; uint8_t *in_set;
    int64_t *out_mem; uint8_t *out_set;
    int64_t *D; int32_t *B; uint8_t *set;    /* n^3 */
    int64_t support;
} opt_ctx;

static inline int64_t wsum(const opt_ctx *c, const uint64_t *bits_a, const uint64_t *bits_b, int mode) {
    /* mode 0: sum W over (a & b); mode 1: sum W over ~(a | b) restricted to valid reps */
    int64_t s = 0;
    for (int64_t w = 0; w < c->words; ++w) {
        uint64_t x = mode == 0 ? (bits_a[w] & bits_b[w]) : ~(bits_a[w] | bits_b[w]);
        if (mode == 1 && w == c->words - 1 && (c->R & 63)) x &= (~0ULL) >> (64 - (c->R & 63));
        if (c->unit_w) s += __builtin_popcountll(x);
        else while (x) { int b = __builtin_ctzll(x); s += c->W[w * 64 + b]; x &= x - 1; }
    }
    return s;
}
static int64_t opt_inside(opt_ctx *c, int32_t i, int32_t j) {      /* :500-506 */
    int64_t idx = (int64_t)i * c->n + j;
    if (!c->in_set[idx]) {
        c->in_mem[idx] = (i == j) ? 0 : -wsum(c, c->yea + idx * c->words, c->nay + idx * c->words, 1);
        c->in_set[idx] = 1;
    }
    return c->in_mem[idx];
}
static int64_t opt_outside(opt_ctx *c, int32_t i, int32_t j, int32_t k) {   /* :509-528 */
    int64_t idx = ((int64_t)i * c->n + j) * c->n + k;
    if (!c->out_set[idx]) {
        int64_t v;
        if (i == j || j == k) v = 0;
        else {
            int64_t ij = (int64_t)i * c->n + j, jk = (int64_t)j * c->n + k;
            /* X1 = yea_ij & nay_jk, X2 = nay_ij & yea_jk; yea and nay are exclusive so X1|X2 is a disjoint union */
            v = wsum(c, c->yea + ij * c->words, c->nay + jk * c->words, 0) +
                wsum(c, c->nay + ij * c->words, c->yea + jk * c->words, 0);
            if (v < c->support) v = NEG_INF;
        }
        c->out_mem[idx] = v; c->out_set[idx] = 1;
    }
    return c->out_mem[idx];
}
static inline int64_t add3(int64_t a, int64_t b, int64_t d) {
    if (a == NEG_INF || b == NEG_INF || d == NEG_INF) return NEG_INF;
    return a + b + d;
}
static int64_t opt_dp(opt_ctx *c, int32_t i, int32_t j, int32_t k) {        /* :532-558 */
    int64_t idx = ((int64_t)i * c->n + j) * c->n + k;
    if (c->set[idx]) return c->D[idx];
    const int32_t *cy = c->cands + c->start;
    int64_t max_d = NEG_INF; int32_t bj = -1, bk = -1, bk2 = -1;
    if (cy[j] - cy[i] < 5 || cy[k] - cy[j] < 5) {                            /* :540-543 */
    } else if (k == c->n - 1) {                                              /* :545-548 */
        max_d = add3(opt_inside(c, i, j), opt_outside(c, i, j, k), opt_inside(c, j, k));
    } else {
        for (int32_t k2 = k + 1; k2 < c->n; ++k2) {                          /* :550-555 */
            int64_t cur = add3(opt_inside(c, i, j), opt_outside(c, i, j, k), opt_dp(c, j, k, k2));
            if (cur > max_d) { max_d = cur; bj = j; bk = k; bk2 = k2; }
        }
    }
    c->D[idx] = max_d; c->B[3 * idx] = bj; c->B[3 * idx + 1] = bk; c->B[3 * idx + 2] = bk2; c->set[idx] = 1;
    return max_d;
}

/* Runs one problem; marks chosen candidate indices in chosen[] (absolute c idx); returns chain length. */
static int32_t optimize_problem(const fo_params *p, const int32_t *cands, const uint32_t *C, int32_t R, const int32_t *W,
                                int unit_w, int32_t start, int32_t end, uint8_t *chosen) {
    opt_ctx c; memset(&c, 0, sizeof c);
    c.start = start; c.end = end; c.n = end - start + 1; c.R = R; c.words = (R + 63) / 64;
    c.cands = cands; c.W = W; c.unit_w = unit_w; c.support = p->min_read_support_outside;
    const int32_t n = c.n;
    size_t nn = (size_t)n * n, nnn = nn * n;
    c.yea = (uint64_t *)xcalloc(nn * (size_t)c.words, 8);
    c.nay = (uint64_t *)xcalloc(nn * (size_t)c.words, 8);
    for (int32_t i = 0; i < n - 1; ++i) {                                     /* :488-497 */
        for (int32_t j = i; j < n; ++j) {
            int64_t seg_len = (int64_t)cands[start + j] - cands[start + i] + 1;
            double h = high_threshold(p, seg_len);
            double l = 1 - h;
            const uint32_t *Ci = C + (size_t)(start + i) * R, *Cj = C + (size_t)(start + j) * R;
            uint64_t *y = c.yea + ((size_t)i * n + j) * c.words, *z = c.nay + ((size_t)i * n + j) * c.words;
            for (int32_t r = 0; r < R; ++r) {
                double cv = (double)(uint32_t)(Cj[r] - Ci[r]) / (double)seg_len;
                if (cv > h) y[r >> 6] |= 1ULL << (r & 63);
                if (cv < l) z[r >> 6] |= 1ULL << (r & 63);
            }
        }
    }
    /* pairs (i, n-1) with i = n-1 are never built by the reference loop (:488 stops at end-1);
     * they are only ever asked for with i == j, where inside() returns 0. */
    c.in_mem = (int64_t *)xcalloc(nn, 8); c.in_set = (uint8_t *)xcalloc(nn, 1);
    c.out_mem = (int64_t *)xcalloc(nnn, 8); c.out_set = (uint8_t *)xcalloc(nnn, 1);
    c.D = (int64_t *)xcalloc(nnn, 8); c.B = (int32_t *)xcalloc(nnn * 3, 4); c.set = (uint8_t *)xcalloc(nnn, 1);
    int64_t max_d = opt_inside(&c, 0, n - 1);                                 /* :560 */
    int32_t b0 = -1, b1 = -1, b2 = -1;
    for (int32_t j = 1; j < n - 1; ++j)                                      /* :562-566 */
        for (int32_t k = j + 1; k < n; ++k) {
            int64_t d = opt_dp(&c, 0, j, k);
            if (d > max_d) { b0 = 0; b1 = j; b2 = k; max_d = d; }
        }
    int32_t chain = 0;
    while (b0 != -1) {                                                        /* run_optimize :592-594 */
        chosen[start + b0] = 1; chosen[start + b1] = 1; chosen[start + b2] = 1; ++chain;
        size_t idx = ((size_t)b0 * n + b1) * n + b2;
        int32_t nb0 = c.B[3 * idx], nb1 = c.B[3 * idx + 1], nb2 = c.B[3 * idx + 2];
        b0 = nb0; b1 = nb1; b2 = nb2;
    }
    free(c.yea); free(c.nay); free(c.in_mem); free(c.in_set); free(c.out_mem); free(c.out_set);
    free(c.D); free(c.B); free(c.set);
    return chain;
}

/* Python round(): half to even */
static inline int64_t py_round(double x) { return (int64_t)nearbyint(x); }

/* ---- refine_segmentation (py/freddie_segment.py:249-266) ------------------------------------ */
static int64_t refine_segmentation(const fo_params *p, const double *y_raw, const int32_t *y_idxs, int64_t m,
                                   int32_t *out) {
    const int32_t skip = 20; const double min_internal = 20;
    int64_t n_out = 0;
    for (int64_t t = 0; t + 1 < m; ++t) {
        int32_t s = y_idxs[t], e = y_idxs[t + 1];
        if (e - s <= 2 * skip) continue;
        int64_t len = e - s;
        double *v = (double *)xmalloc(sizeof(double) * (size_t)len);
        for (int64_t i = 0; i < len; ++i) v[i] = y_raw[s + i];
        for (int i = 0; i < skip; ++i) { v[i] = 0.0; v[len - i - 1] = 0.0; }
        double tot = 0.0;                                                    /* sum() from int 0, left to right */
        for (int64_t i = 0; i < len; ++i) tot += v[i];
        if (tot < min_internal) { free(v); continue; }
        double *g = (double *)xmalloc(sizeof(double) * (size_t)len);
        fo_gaussian(v, len, p->w_refine, p->radius_refine, 1, g);
        int32_t *pk = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)len);
        int64_t np_ = fo_local_maxima(g, len, pk);
        double *pr = (double *)xmalloc(sizeof(double) * (size_t)(np_ ? np_ : 1));
        uint8_t *keep = (uint8_t *)xmalloc((size_t)(np_ ? np_ : 1));
        for (int64_t i = 0; i < np_; ++i) pr[i] = g[pk[i]];
        fo_select_by_distance(pk, pr, np_, skip, keep);
        for (int64_t q = 0; q < np_; ++q) {
            if (!keep[q]) continue;
            int64_t i = pk[q];
            int64_t a = py_round((double)i - p->sigma), b = py_round((double)i + p->sigma + 1);
            /* Python slice semantics */
            if (a < 0) { a += len; if (a < 0) a = 0; } else if (a > len) a = len;
            if (b < 0) { b += len; if (b < 0) b = 0; } else if (b > len) b = len;
            double sm = 0.0;
            for (int64_t x = a; x < b; ++x) sm += g[x];
            if (sm < min_internal) continue;
            out[n_out++] = (int32_t)(i + s);
        }
        free(v); free(g); free(pk); free(pr); free(keep);
    }
    return n_out;
}

static int cmp_i32(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }

void fo_free(fo_result *r) {
    if (!r) return;
    free(r->pos_off); free(r->Y_raw); free(r->Y); free(r->cand_off); free(r->cands);
    free(r->fixed_off); free(r->fixed); free(r->prob_interval); free(r->prob_start); free(r->prob_end);
    free(r->prob_nchain); free(r->finalc_off); free(r->finalc); free(r->refine_off); free(r->refine);
    free(r->final_off); free(r->final_y); free(r->final_pos); free(r->labels);
    free(r);
}

/* ---- segment() (py/freddie_segment.py:738-844), numeric part (everything before gaps/polyA) ---
 * Inputs: K tint intervals [iv_start, iv_end] (both ends are positions, :652-659);
 * R read reps with multiplicity rep_weight and exon lists (genomic ts, te), CSR by rep_exon_off.
 * stop_after: 0 = full run; 1 = stop after smoothing/threshold/candidates/fixed (cheap stages). */
fo_result *fo_segment(const fo_params *p, int32_t K, const int32_t *iv_start, const int32_t *iv_end, int32_t R,
                      const int32_t *rep_weight, const int64_t *rep_exon_off, const int32_t *ex_ts,
                      const int32_t *ex_te, int32_t stop_after) {
    fo_result *res = (fo_result *)xcalloc(1, sizeof(fo_result));
    res->K = K; res->R = R;
    /* process_splicing_data :648-678 */
    res->pos_off = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(K + 1));
    res->pos_off[0] = 0;
    for (int32_t k = 0; k < K; ++k) res->pos_off[k + 1] = res->pos_off[k] + (iv_end[k] - iv_start[k] + 1);
    const int64_t P = res->P = res->pos_off[K];
    res->Y_raw = (double *)xcalloc((size_t)P, sizeof(double));
    res->Y = (double *)xcalloc((size_t)P, sizeof(double));
    const int64_t I = rep_exon_off[R];
    int32_t *ex_iv = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)I);
    int32_t *ex_ys = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)I);
    int32_t *ex_ye = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)I);
    for (int32_t r = 0; r < R; ++r) {
        int64_t e0 = rep_exon_off[r], e1 = rep_exon_off[r + 1];
        for (int64_t e = e0; e < e1; ++e) {
            /* pos_to_Yy_idx lookup (:666-667): the interval containing ts must also contain te (:668) */
            int32_t lo = 0, hi = K - 1, k = -1;
            while (lo <= hi) { int32_t mid = (lo + hi) / 2; if (ex_ts[e] < iv_start[mid]) hi = mid - 1; else if (ex_ts[e] > iv_end[mid]) lo = mid + 1; else { k = mid; break; } }
            if (k < 0 || ex_te[e] < iv_start[k] || ex_te[e] > iv_end[k]) { set_err(res, "exon not inside one tint interval (:666-668)"); ex_iv[e] = -1; ex_ys[e] = ex_ye[e] = 0; continue; }
            ex_iv[e] = k; ex_ys[e] = ex_ts[e] - iv_start[k]; ex_ye[e] = ex_te[e] - iv_start[k];
            if (!(p->ignore_ends && e == e0)) res->Y_raw[res->pos_off[k] + ex_ys[e]] += (double)rep_weight[r];       /* :670-671 */
            if (!(p->ignore_ends && e == e1 - 1)) res->Y_raw[res->pos_off[k] + ex_ye[e]] += (double)rep_weight[r];   /* :672-673 */
        }
    }
    if (res->error) goto done_early;
    /* :755 */
    for (int32_t k = 0; k < K; ++k)
        fo_gaussian(res->Y_raw + res->pos_off[k], res->pos_off[k + 1] - res->pos_off[k], p->w_main, p->radius_main, 0,
                    res->Y + res->pos_off[k]);
    /* :757-759 */
    res->threshold = fo_variance_threshold(res->Y, P, p->variance_factor, &res->n_vals);

    res->cand_off = (int64_t *)xcalloc((size_t)(K + 1), 8);
    res->fixed_off = (int64_t *)xcalloc((size_t)(K + 1), 8);
    res->finalc_off = (int64_t *)xcalloc((size_t)(K + 1), 8);
    res->refine_off = (int64_t *)xcalloc((size_t)(K + 1), 8);
    res->final_off = (int64_t *)xcalloc((size_t)(K + 1), 8);
    res->cands = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(P + 2 * K));
    res->fixed = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(P + 2 * K));
    res->finalc = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(P + 2 * K));
    res->refine = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(P + 2 * K));
    res->final_y = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(2 * P + 4 * K));
    res->final_pos = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(2 * P + 4 * K));
    int64_t prob_cap = P + 2 * K;
    res->prob_interval = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)prob_cap);
    res->prob_start = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)prob_cap);
    res->prob_end = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)prob_cap);
    res->prob_nchain = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)prob_cap);
    int unit_w = 1;
    for (int32_t r = 0; r < R; ++r) if (rep_weight[r] != 1) unit_w = 0;

    /* pass 1: candidates + fixed per interval (needed before labels can be sized) */
    for (int32_t k = 0; k < K; ++k) {
        const double *y = res->Y + res->pos_off[k];
        const int64_t len = res->pos_off[k + 1] - res->pos_off[k];
        /* candidates_from_peaks :615-621 */
        int32_t *cands = res->cands + res->cand_off[k];
        int64_t nc = 0;
        cands[nc++] = 0;
        {
            int32_t *pk = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)len);
            int64_t m = fo_local_maxima(y, len, pk);
            for (int64_t i = 0; i < m; ++i) cands[nc++] = pk[i];   /* peaks are in (0, len-1), already sorted */
            free(pk);
        }
        if (len - 1 != 0) cands[nc++] = (int32_t)(len - 1);
        res->cand_off[k + 1] = res->cand_off[k] + nc;
        /* fixing :776-783 */
        uint8_t *fx = (uint8_t *)xcalloc((size_t)nc, 1);
        fx[0] = 1; fx[nc - 1] = 1;
        for (int64_t c = 0; c < nc; ++c) if (y[cands[c]] > res->threshold) fx[c] = 1;
        /* break_large_problems :623-645 (pairs computed once, before any insertion) */
        {
            int32_t *f0 = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)nc); int64_t nf = 0;
            for (int64_t c = 0; c < nc; ++c) if (fx[c]) f0[nf++] = (int32_t)c;
            for (int64_t q = 0; q + 1 < nf; ++q) {
                int32_t cs = f0[q], ce = f0[q + 1];
                int32_t size = ce - cs + 1;
                if (size <= p->max_problem_size) continue;
                int32_t cnt = (int32_t)ceil((double)size / (double)p->max_problem_size);
                double step = (double)size / (double)cnt;
                for (int32_t i = 1; i < cnt; ++i) {
                    int32_t anchor = (int32_t)((double)cs + (double)i * step);
                    double best = -INFINITY; int32_t best_c = -1;
                    for (int32_t c = anchor - 5; c < anchor + 5; ++c) {
                        int64_t cc = c;
                        if (cc < 0) cc += nc;                     /* Python negative-index wraparound */
                        if (cc < 0 || cc >= nc) { set_err(res, "break_large_problems: index out of range (:640)"); continue; }
                        if (y[cands[cc]] > best) { best = y[cands[cc]]; best_c = c; }
                    }
                    if (!(best > 0)) { set_err(res, "break_large_problems: assert max_c_idx_y_v > 0 (:643)"); continue; }
                    /* the reference adds max_c_idx as found (possibly negative); a negative member would
                     * break sorted()/indexing later, treat as error */
                    if (best_c < 0) { set_err(res, "break_large_problems: negative anchor"); continue; }
                    fx[best_c] = 1;
                }
            }
            free(f0);
        }
        int64_t nf = 0;
        for (int64_t c = 0; c < nc; ++c) if (fx[c]) res->fixed[res->fixed_off[k] + nf++] = (int32_t)c;
        res->fixed_off[k + 1] = res->fixed_off[k] + nf;
        free(fx);
    }
    if (stop_after == 1 || res->error) goto done_early;

    /* pass 2: DP, refinement, final positions */
    for (int32_t k = 0; k < K; ++k) {
        const int64_t len = res->pos_off[k + 1] - res->pos_off[k];
        const int32_t *cands = res->cands + res->cand_off[k];
        const int64_t nc = res->cand_off[k + 1] - res->cand_off[k];
        const int32_t *fixed = res->fixed + res->fixed_off[k];
        const int64_t nf = res->fixed_off[k + 1] - res->fixed_off[k];
        uint32_t *C = cumulative_coverage(res, R, rep_exon_off, ex_iv, ex_ys, ex_ye, k, cands, nc);   /* :769 */
        uint8_t *chosen = (uint8_t *)xcalloc((size_t)nc, 1);
        for (int64_t q = 0; q < nf; ++q) chosen[fixed[q]] = 1;                                        /* :580 */
        for (int64_t q = 0; q + 1 < nf; ++q) {                                                        /* :581 */
            int64_t pi = res->n_problems++;
            res->prob_interval[pi] = k; res->prob_start[pi] = fixed[q]; res->prob_end[pi] = fixed[q + 1];
            res->prob_nchain[pi] = optimize_problem(p, cands, C, R, rep_weight, unit_w, fixed[q], fixed[q + 1], chosen);
        }
        free(C);
        int64_t nfc = 0;
        int32_t *fy = res->final_y + res->final_off[k];
        int64_t nfy = 0;
        for (int64_t c = 0; c < nc; ++c) if (chosen[c]) { res->finalc[res->finalc_off[k] + nfc++] = (int32_t)c; fy[nfy++] = cands[c]; }
        res->finalc_off[k + 1] = res->finalc_off[k] + nfc;
        free(chosen);
        int64_t nr = refine_segmentation(p, res->Y_raw + res->pos_off[k], fy, nfy, res->refine + res->refine_off[k]);  /* :803 */
        res->refine_off[k + 1] = res->refine_off[k] + nr;
        for (int64_t i = 0; i < nr; ++i) fy[nfy++] = res->refine[res->refine_off[k] + i];             /* :804 */
        qsort(fy, (size_t)nfy, sizeof(int32_t), cmp_i32);                                              /* :805 */
        res->final_off[k + 1] = res->final_off[k] + nfy;
        for (int64_t i = 0; i < nfy; ++i) res->final_pos[res->final_off[k] + i] = iv_start[k] + fy[i]; /* :806-807 */
        (void)len;
    }
    res->F = res->final_off[K];
    /* labels :808-830, sentinel :829-830, pop :840 */
    {
        const int64_t S = res->F - 1;
        res->labels = (uint8_t *)xcalloc((size_t)R * (size_t)(S > 0 ? S : 1), 1);
        int64_t col = 0;
        for (int32_t k = 0; k < K; ++k) {
            const int32_t *fy = res->final_y + res->final_off[k];
            const int64_t nfy = res->final_off[k + 1] - res->final_off[k];
            uint32_t *C = cumulative_coverage(res, R, rep_exon_off, ex_iv, ex_ys, ex_ye, k, fy, nfy);  /* :808 */
            for (int64_t t = 0; t + 1 < nfy; ++t) {
                int64_t seg_len = (int64_t)fy[t + 1] - fy[t] + 1;
                double h = high_threshold(p, seg_len);
                double l = 1 - h;
                for (int32_t r = 0; r < R; ++r) {
                    double ratio = (double)(uint32_t)(C[(t + 1) * R + r] - C[t * R + r]) / (double)seg_len;
                    if (!(0 <= ratio && ratio <= 1)) set_err(res, "label: ratio out of [0,1] (:821)");
                    res->labels[(size_t)r * S + col] = ratio > h ? 1 : (ratio < l ? 0 : 2);
                }
                ++col;
            }
            free(C);
            if (k + 1 < K) ++col;        /* sentinel 0 between intervals; the last one is popped */
        }
    }
done_early:
    free(ex_iv); free(ex_ys); free(ex_ye);
    return res;
}

/* getters for ctypes */
#define GETTER(type, name, field) type name(const fo_result *r) { return r->field; }
GETTER(int64_t, fo_P, P)
GETTER(int64_t, fo_F, F)
GETTER(int64_t, fo_n_problems, n_problems)
GETTER(int64_t, fo_n_vals, n_vals)
GETTER(double, fo_threshold, threshold)
GETTER(int32_t, fo_error, error)
GETTER(const char *, fo_errmsg, errmsg)
GETTER(const int64_t *, fo_pos_off, pos_off)
GETTER(const double *, fo_Y_raw, Y_raw)
GETTER(const double *, fo_Y, Y)
GETTER(const int64_t *, fo_cand_off, cand_off)
GETTER(const int32_t *, fo_cands, cands)
GETTER(const int64_t *, fo_fixed_off, fixed_off)
GETTER(const int32_t *, fo_fixed, fixed)
GETTER(const int32_t *, fo_prob_interval, prob_interval)
GETTER(const int32_t *, fo_prob_start, prob_start)
GETTER(const int32_t *, fo_prob_end, prob_end)
GETTER(const int32_t *, fo_prob_nchain, prob_nchain)
GETTER(const int64_t *, fo_finalc_off, finalc_off)
GETTER(const int32_t *, fo_finalc, finalc)
GETTER(const int64_t *, fo_refine_off, refine_off)
GETTER(const int32_t *, fo_refine, refine)
GETTER(const int64_t *, fo_final_off, final_off)
GETTER(const int32_t *, fo_final_y, final_y)
GETTER(const int32_t *, fo_final_pos, final_pos)
GETTER(const uint8_t *, fo_labels, labels)

#ifdef FREDDIE_SOURCE_HASH
/* what this binary was built from (freddie_amd/build.py looks for the marker in the file) */
static const char freddie_source_stamp[] __attribute__((used)) = "FREDDIE_SRC_HASH=" FREDDIE_SOURCE_HASH;
#endif
