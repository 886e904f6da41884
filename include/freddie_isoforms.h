/*
 * freddie_isoforms.h -- C-ABI of the GPU part of the isoform-consensus stage.
 *
 * It replaces the two per-read loops of the reference's py/freddie_isoforms.py:
 *   - isoforms_cons()       :203-250   per isoform and segment: how many member reads span the segment (cov) and how
 *                                      many of those cover it (cons); per isoform the poly-tail categories of its reads
 *   - correct_boundaries()  :122-140   per isoform boundary: votes of the member reads' alignment boundaries that lie
 *                                      within +-correction_window of it, by offset
 * The decisions taken from these integers (x/c > 0.5 with x >= 3, v/N >= majority_threshold, strand, exon runs) are
 * float / control logic of a few operations per isoform and stay on the host (freddie_amd/isoforms.py), as do the
 * cluster / split TSV readers (:143-201) and the GTF writer (:72-119).
 *
 * Layout (caller-owned host arrays): reads are grouped by isoform, isoform i owns reads iso_read_off[i] ..
 * iso_read_off[i+1]; read r's corrected labels are the n_seg[i] ASCII bytes at labels[read_lab_off[r]] and its
 * poly-tail category 'N','S','E' is tail[r] = 0,1,2.  Results for isoform i start at iso_seg_off[i] (one int32 per
 * segment) and at 3 * i (tails).
 */
#ifndef FREDDIE_ISOFORMS_H
#define FREDDIE_ISOFORMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fiso_ctx fiso_ctx;

enum { FISO_OK = 0, FISO_ERR_ARG = 1, FISO_ERR_HIP = 2 };

int fiso_abi_version(void);
int fiso_create(int device, fiso_ctx **out);      /* fails when no HIP device is usable: there is no CPU fallback */
void fiso_destroy(fiso_ctx *c);
const char *fiso_last_error(const fiso_ctx *c);   /* c may be NULL: error of the last failed fiso_create() */

/* isoforms_cons() counts (py/freddie_isoforms.py:203-232).  A read without any '1' is skipped altogether (:215-216);
 * a read with tail 'S' spans every segment (:217-224, as written in the reference: both tests are on 'S'). */
int fiso_consensus(fiso_ctx *c, int32_t n_iso, const int64_t *iso_read_off, const int32_t *n_seg, const int64_t *iso_seg_off,
                   const int64_t *read_lab_off, const uint8_t *labels, const uint8_t *tail,
                   int32_t *cons_out, int32_t *cov_out, int32_t *tails_out);

/* The same with the labels at two bits each, in the layout fseg_results_packed() (include/freddie_seg.h) delivers: label g of
 * the arena = bits 2(g & 3).. of labels2[g >> 2], code = ASCII & 3; read_lab_off still counts labels.  A quarter of the bytes
 * across PCIe and out of HBM; for a producer that holds the packed form (the text TSVs the stages exchange hold bytes). */
int fiso_consensus_packed(fiso_ctx *c, int32_t n_iso, const int64_t *iso_read_off, const int32_t *n_seg, const int64_t *iso_seg_off,
                          const int64_t *read_lab_off, const uint8_t *labels2, const uint8_t *tail,
                          int32_t *cons_out, int32_t *cov_out, int32_t *tails_out);

/* correct_boundaries() votes for one side (py/freddie_isoforms.py:129-137).  Isoform i owns the (ascending) boundaries
 * iso_bound[iso_b_off[i] .. iso_b_off[i+1]); read r owns read_bound[read_b_off[r] .. read_b_off[r+1]).
 * votes_out[(iso_b_off[i] + idx) * (2 * window + 1) + (x + window)] = number of (read, boundary) of isoform i with
 * boundary - iso_bound == x.  window in [1, 20] (:45). */
int fiso_boundary_votes(fiso_ctx *c, int32_t n_iso, const int64_t *iso_read_off, const int64_t *iso_b_off, const int32_t *iso_bound,
                        const int64_t *read_b_off, const int32_t *read_bound, int32_t window, int32_t *votes_out);

/* Kernel time of the last call (HIP events on the library's stream), ms. */
int fiso_last_kernel_ms(fiso_ctx *c, float *ms);

#ifdef __cplusplus
}
#endif
#endif
