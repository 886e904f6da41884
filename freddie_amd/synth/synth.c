/*
 * Deterministic synthetic split-partition generator (test + bench infrastructure).
 *
 * Produces the INPUT of the segmentation stage: one "tint" partition as written by
 * the split stage (format: reference py/freddie_split.py:445-481 for split_*.tsv and
 * :395-401 for reads_*.tsv; interval merging rule: py/freddie_split.py:295-323).
 * The generator itself has no counterpart in the reference (the reference ships no
 * test data); its recipe follows SURVEY.md section 8(d).
 *
 * Everything is integer arithmetic + IEEE add/mul on doubles driven by splitmix64,
 * so the same seed gives byte-identical partitions on every machine.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t seed;
    int32_t n_reads;
    int32_t n_exons;      /* exons in the gene model */
    int32_t n_isoforms;   /* default 8 */
    int32_t max_span;     /* max exons per read, 0 = unbounded */
    int32_t origin;       /* genomic origin, default 100000 */
    int32_t tint_id;
    int32_t rid_base;     /* first read id */
    int32_t indel_permille; /* chance per exon of an I/D op inside the CIGAR */
    double keep_p;        /* isoform keeps an exon w.p. keep_p (0.7) */
    double rp;            /* intron-retention prob per adjacent exon pair */
    double jp;            /* boundary jitter prob */
    double jsd;           /* boundary jitter sd */
} fsynth_params;

typedef struct {
    fsynth_params p;
    int32_t n_reads;
    int32_t n_intervals;
    int32_t *iv_start, *iv_end;      /* tint intervals [s,e) as split writes them */
    int64_t n_exons_total;
    int64_t *read_exon_off;          /* n_reads+1 */
    int32_t *ex_ts, *ex_te, *ex_qs, *ex_qe;
    int64_t *ex_cig_off;             /* n_exons_total+1 */
    int64_t n_cig;
    int32_t *cig_len;
    char *cig_op;
    char *strand;                    /* n_reads */
    int64_t *seq_off;                /* n_reads+1 */
    char *seq;
} fsynth;

static inline uint64_t sm64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline int32_t urange(uint64_t *s, int32_t lo, int32_t hi) { /* inclusive */
    return lo + (int32_t)(sm64(s) % (uint64_t)(hi - lo + 1));
}
static inline double udouble(uint64_t *s) { return (double)(sm64(s) >> 11) * (1.0 / 9007199254740992.0); }
/* Irwin-Hall(12) - 6: mean 0, variance 1; only adds, so bit-reproducible */
static inline double gauss(uint64_t *s) {
    double a = 0.0;
    for (int i = 0; i < 12; ++i) a += udouble(s);
    return a - 6.0;
}
static inline int32_t iround(double x) { return (int32_t)(x < 0 ? x - 0.5 : x + 0.5); }

#define GROW(ptr, cap, need, type)                                   \
    do {                                                             \
        if ((need) > (cap)) {                                        \
            while ((need) > (cap)) (cap) = (cap) ? (cap) * 2 : 1024; \
            (ptr) = (type *)realloc((ptr), (size_t)(cap) * sizeof(type)); \
        }                                                            \
    } while (0)

static int cmp_iv(const void *a, const void *b) {
    const int32_t *x = (const int32_t *)a, *y = (const int32_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
    return 0;
}

void fsynth_free(fsynth *g) {
    if (!g) return;
    free(g->iv_start); free(g->iv_end); free(g->read_exon_off);
    free(g->ex_ts); free(g->ex_te); free(g->ex_qs); free(g->ex_qe);
    free(g->ex_cig_off); free(g->cig_len); free(g->cig_op);
    free(g->strand); free(g->seq_off); free(g->seq);
    free(g);
}

fsynth *fsynth_generate(const fsynth_params *pp, int32_t with_seq) {
    fsynth *g = (fsynth *)calloc(1, sizeof(fsynth));
    g->p = *pp;
    const fsynth_params p = *pp;
    uint64_t st = p.seed;
    uint64_t sq_st = p.seed ^ 0xA5A5A5A5C3C3C3C3ULL;   /* sequence letters: own stream, so that the
                                                           partition structure does not depend on with_seq */
    const int E = p.n_exons, NI = p.n_isoforms > 0 ? p.n_isoforms : 8;
    int32_t *gs = (int32_t *)malloc(sizeof(int32_t) * E), *ge = (int32_t *)malloc(sizeof(int32_t) * E);
    int32_t pos = p.origin;
    for (int e = 0; e < E; ++e) {
        int32_t len = urange(&st, 60, 300);
        gs[e] = pos; ge[e] = pos + len;
        pos = ge[e] + urange(&st, 100, 400);
    }
    int32_t **iso = (int32_t **)malloc(sizeof(int32_t *) * NI);
    int32_t *iso_n = (int32_t *)malloc(sizeof(int32_t) * NI);
    for (int u = 0; u < NI; ++u) {
        iso[u] = (int32_t *)malloc(sizeof(int32_t) * E);
        int m = 0;
        for (int e = 0; e < E; ++e) {
            double x = udouble(&st);
            if (e == 0 || e == E - 1 || x < p.keep_p) iso[u][m++] = e;
        }
        iso_n[u] = m;
    }
    const int n = p.n_reads;
    g->n_reads = n;
    g->read_exon_off = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    g->strand = (char *)malloc(n);
    g->seq_off = (int64_t *)calloc(n + 1, sizeof(int64_t));
    int64_t ex_cap = 0, cig_cap = 0, seq_cap = 0, n_ex = 0, n_cig = 0, n_seq = 0;
    int32_t *tmp_ts = (int32_t *)malloc(sizeof(int32_t) * (E + 1)), *tmp_te = (int32_t *)malloc(sizeof(int32_t) * (E + 1));
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    g->ex_cig_off = NULL;
    int64_t cigoff_cap = 0;
    for (int i = 0; i < n; ++i) {
        int u = urange(&st, 0, NI - 1);
        int m = iso_n[u];
        int hi = (p.max_span > 0 && p.max_span < m) ? p.max_span : m;
        if (hi < 2) hi = m < 2 ? m : 2;
        int span = m < 2 ? m : urange(&st, 2, hi);
        int a = urange(&st, 0, m - span);
        /* exon list with intron retention */
        int k = 0;
        for (int j = 0; j < span; ++j) {
            int e = iso[u][a + j];
            if (k > 0 && udouble(&st) < p.rp) {
                tmp_te[k - 1] = ge[e];
            } else {
                tmp_ts[k] = gs[e]; tmp_te[k] = ge[e]; ++k;
            }
        }
        /* boundary jitter, clipped so that exons stay ordered and non-empty */
        for (int j = 0; j < k; ++j) {
            for (int side = 0; side < 2; ++side) {
                if (udouble(&st) < p.jp) {
                    int32_t d = iround(gauss(&st) * p.jsd);
                    if (d > 25) d = 25;
                    if (d < -25) d = -25;
                    if (side == 0) tmp_ts[j] += d; else tmp_te[j] += d;
                }
            }
        }
        g->strand[i] = (sm64(&st) & 1) ? '+' : '-';
        int lead = urange(&st, 0, 12);
        int tail_kind = urange(&st, 0, 2);   /* 0: polyA at end, 1: polyT at start, 2: none */
        int poly_len = urange(&st, 0, 40);
        int trail = urange(&st, 0, 4);
        int lead_total = lead + (tail_kind == 1 ? poly_len : 0);
        int tail_total = trail + (tail_kind == 0 ? poly_len : 0);
        g->read_exon_off[i] = n_ex;
        GROW(g->ex_ts, ex_cap, n_ex + k, int32_t);
        g->ex_te = (int32_t *)realloc(g->ex_te, (size_t)ex_cap * sizeof(int32_t));
        g->ex_qs = (int32_t *)realloc(g->ex_qs, (size_t)ex_cap * sizeof(int32_t));
        g->ex_qe = (int32_t *)realloc(g->ex_qe, (size_t)ex_cap * sizeof(int32_t));
        GROW(g->ex_cig_off, cigoff_cap, n_ex + k + 1, int64_t);
        int32_t q = lead_total;
        for (int j = 0; j < k; ++j) {
            int32_t tlen = tmp_te[j] - tmp_ts[j];
            GROW(g->cig_len, cig_cap, n_cig + 3, int32_t);
            g->cig_op = (char *)realloc(g->cig_op, (size_t)cig_cap);
            g->ex_cig_off[n_ex] = n_cig;
            int32_t qlen;
            if (p.indel_permille > 0 && urange(&st, 0, 999) < p.indel_permille && tlen > 50) {
                int32_t c = urange(&st, 1, 20);
                int is_del = (int)(sm64(&st) & 1);
                if (is_del) {
                    int32_t a1 = urange(&st, 5, tlen - c - 5);
                    g->cig_len[n_cig] = a1; g->cig_op[n_cig++] = 'M';
                    g->cig_len[n_cig] = c; g->cig_op[n_cig++] = 'D';
                    g->cig_len[n_cig] = tlen - c - a1; g->cig_op[n_cig++] = 'M';
                    qlen = tlen - c;
                } else {
                    int32_t a1 = urange(&st, 5, tlen - 5);
                    g->cig_len[n_cig] = a1; g->cig_op[n_cig++] = 'M';
                    g->cig_len[n_cig] = c; g->cig_op[n_cig++] = 'I';
                    g->cig_len[n_cig] = tlen - a1; g->cig_op[n_cig++] = 'M';
                    qlen = tlen + c;
                }
            } else {
                g->cig_len[n_cig] = tlen; g->cig_op[n_cig++] = 'M';
                qlen = tlen;
            }
            /* occasional unaligned query bases between exons */
            if (j > 0 && urange(&st, 0, 19) == 0) q += urange(&st, 1, 15);
            g->ex_ts[n_ex] = tmp_ts[j]; g->ex_te[n_ex] = tmp_te[j];
            g->ex_qs[n_ex] = q; g->ex_qe[n_ex] = q + qlen;
            q += qlen;
            ++n_ex;
        }
        g->ex_cig_off[n_ex] = n_cig;
        int32_t total = q + tail_total;
        g->seq_off[i + 1] = g->seq_off[i] + (with_seq ? total : 0);
        if (with_seq) {
            GROW(g->seq, seq_cap, n_seq + total, char);
            char *sq = g->seq + n_seq;
            for (int32_t x = 0; x < total; ++x) sq[x] = ACGT[sm64(&sq_st) & 3];
            /* poly tails with ~4% impurities */
            if (tail_kind == 0)
                for (int32_t x = 0; x < poly_len; ++x)
                    if (urange(&sq_st, 0, 24) != 0) sq[q + x] = 'A';
            if (tail_kind == 1)
                for (int32_t x = 0; x < poly_len; ++x)
                    if (urange(&sq_st, 0, 24) != 0) sq[lead + x] = 'T';
            n_seq += total;
        }
    }
    g->read_exon_off[n] = n_ex;
    g->n_exons_total = n_ex;
    g->n_cig = n_cig;
    /* tint intervals: sweep-merge of all exons, new interval when s > running end */
    int32_t *srt = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(n_ex ? n_ex : 1));
    for (int64_t x = 0; x < n_ex; ++x) { srt[2 * x] = g->ex_ts[x]; srt[2 * x + 1] = g->ex_te[x]; }
    qsort(srt, (size_t)n_ex, 2 * sizeof(int32_t), cmp_iv);
    int64_t iv_cap = 0; int32_t niv = 0;
    int32_t cs = 0, ce = 0; int have = 0;
    for (int64_t x = 0; x < n_ex; ++x) {
        int32_t s = srt[2 * x], e = srt[2 * x + 1];
        if (!have) { cs = s; ce = e; have = 1; }
        if (s > ce) {
            GROW(g->iv_start, iv_cap, niv + 1, int32_t);
            g->iv_end = (int32_t *)realloc(g->iv_end, (size_t)iv_cap * sizeof(int32_t));
            g->iv_start[niv] = cs; g->iv_end[niv] = ce; ++niv;
            cs = s; ce = e;
        }
        if (e > ce) ce = e;
    }
    if (have) {
        GROW(g->iv_start, iv_cap, niv + 1, int32_t);
        g->iv_end = (int32_t *)realloc(g->iv_end, (size_t)iv_cap * sizeof(int32_t));
        g->iv_start[niv] = cs; g->iv_end[niv] = ce; ++niv;
    }
    g->n_intervals = niv;
    free(srt); free(tmp_ts); free(tmp_te); free(gs); free(ge);
    for (int u = 0; u < NI; ++u) free(iso[u]);
    free(iso); free(iso_n);
    return g;
}

int32_t fsynth_n_reads(const fsynth *g) { return g->n_reads; }
int32_t fsynth_n_intervals(const fsynth *g) { return g->n_intervals; }
int64_t fsynth_n_exons(const fsynth *g) { return g->n_exons_total; }
int64_t fsynth_n_cigar(const fsynth *g) { return g->n_cig; }
int64_t fsynth_seq_bytes(const fsynth *g) { return g->seq_off[g->n_reads]; }

void fsynth_copy_intervals(const fsynth *g, int32_t *s, int32_t *e) {
    memcpy(s, g->iv_start, sizeof(int32_t) * g->n_intervals);
    memcpy(e, g->iv_end, sizeof(int32_t) * g->n_intervals);
}
void fsynth_copy_exons(const fsynth *g, int64_t *off, int32_t *ts, int32_t *te, int32_t *qs, int32_t *qe) {
    memcpy(off, g->read_exon_off, sizeof(int64_t) * (g->n_reads + 1));
    size_t b = sizeof(int32_t) * (size_t)g->n_exons_total;
    memcpy(ts, g->ex_ts, b); memcpy(te, g->ex_te, b);
    if (qs) memcpy(qs, g->ex_qs, b);
    if (qe) memcpy(qe, g->ex_qe, b);
}
void fsynth_copy_cigar(const fsynth *g, int64_t *ex_cig_off, int32_t *len, char *op) {
    memcpy(ex_cig_off, g->ex_cig_off, sizeof(int64_t) * (size_t)(g->n_exons_total + 1));
    memcpy(len, g->cig_len, sizeof(int32_t) * (size_t)g->n_cig);
    memcpy(op, g->cig_op, (size_t)g->n_cig);
}
void fsynth_copy_reads(const fsynth *g, char *strand, int64_t *seq_off, char *seq) {
    memcpy(strand, g->strand, (size_t)g->n_reads);
    memcpy(seq_off, g->seq_off, sizeof(int64_t) * (g->n_reads + 1));
    if (seq && g->seq) memcpy(seq, g->seq, (size_t)g->seq_off[g->n_reads]);
}

/* Write split_<contig>_<tint>.tsv and reads_<contig>_<tint>.tsv. Returns 0 on success. */
int fsynth_write_tsv(const fsynth *g, const char *split_path, const char *reads_path, const char *contig) {
    FILE *f = fopen(split_path, "w");
    if (!f) return -1;
    fprintf(f, "#%s\t%d\t", contig, g->p.tint_id);
    for (int i = 0; i < g->n_intervals; ++i)
        fprintf(f, "%s%d-%d", i ? "," : "", g->iv_start[i], g->iv_end[i]);
    fprintf(f, "\t%d\n", g->n_reads);
    for (int i = 0; i < g->n_reads; ++i) {
        fprintf(f, "%d\tsynth_%d_%d\t%s\t%c\t%d", g->p.rid_base + i, g->p.tint_id, i, contig, g->strand[i], g->p.tint_id);
        for (int64_t x = g->read_exon_off[i]; x < g->read_exon_off[i + 1]; ++x) {
            fprintf(f, "\t%d-%d:%d-%d:", g->ex_ts[x], g->ex_te[x], g->ex_qs[x], g->ex_qe[x]);
            for (int64_t c = g->ex_cig_off[x]; c < g->ex_cig_off[x + 1]; ++c)
                fprintf(f, "%d%c", g->cig_len[c], g->cig_op[c]);
        }
        fputc('\n', f);
    }
    fclose(f);
    if (reads_path && g->seq) {
        f = fopen(reads_path, "w");
        if (!f) return -2;
        for (int i = 0; i < g->n_reads; ++i) {
            fprintf(f, "%d\t%s\t%d\t", g->p.rid_base + i, contig, g->p.tint_id);
            fwrite(g->seq + g->seq_off[i], 1, (size_t)(g->seq_off[i + 1] - g->seq_off[i]), f);
            fputc('\n', f);
        }
        fclose(f);
    }
    return 0;
}

#ifdef FREDDIE_SOURCE_HASH
/* what this binary was built from (freddie_amd/build.py looks for the marker in the file) */
static const char freddie_source_stamp[] __attribute__((used)) = "FREDDIE_SRC_HASH=" FREDDIE_SOURCE_HASH;
#endif
