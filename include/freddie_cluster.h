/*
 * freddie_cluster.h -- C-ABI of the GPU part of the clustering stage's pre-ILP work.
 *
 * It replaces the two quadratic loops of the reference's partition_reads() (py/freddie_cluster.py:196-274):
 *   - the pairwise read compatibility test            py/freddie_cluster.py:217-234
 *   - the iterated edge pruning of the compatibility graph   py/freddie_cluster.py:240-255
 * for a batch of transcriptional intervals ("tints") at once.  Everything else of partition_reads() -- the dedupe of
 * reads with the same structure (:207-215), connected components, the even split and the incompatible-pair lists
 * (:256-274) -- is host code (freddie_amd/cluster_prep.py), as are read_segment() (:119-172) and preprocess_ilp()
 * (:277-328).  The ILP itself (run_ilp, Gurobi) is out of scope.
 *
 * Data layout (caller-owned host arrays, copied by the call):
 *   tint t owns the unique reads row_off[t] .. row_off[t+1]  (N_t of them; "unique" = py/freddie_cluster.py:207-215)
 *   every unique read of tint t is a row of W_t = ceil(M_t / 32) uint32 words at bits[bits_off[t] + r * W_t]:
 *     bit s of the row = I[read][s] (1 = the read covers segment s; label 2 counts as 0, :287-288)
 *   first[r], last[r] = FL[read] (:301), tail[r] = poly_tail_category 'N','S','E' as 0,1,2 (:291-300)
 * Result: for tint t a symmetric N_t x N_t bit matrix, row r at adj[adj_off[t] + r * AW_t], AW_t = ceil(N_t / 64)
 * uint64 words, bit c of the row = the graph has the edge (r, c) after pruning (prune != 0) or before it (prune == 0).
 */
#ifndef FREDDIE_CLUSTER_H
#define FREDDIE_CLUSTER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fclu_ctx fclu_ctx;

enum { FCLU_OK = 0, FCLU_ERR_ARG = 1, FCLU_ERR_HIP = 2, FCLU_ERR_UNSUPPORTED = 3 };

int fclu_abi_version(void);
/* One context per device and host thread.  Fails (no CPU fallback) when no HIP device is usable. */
int fclu_create(int device, fclu_ctx **out);
void fclu_destroy(fclu_ctx *c);
const char *fclu_last_error(const fclu_ctx *c);   /* c may be NULL: error of the last failed fclu_create() */

typedef struct fclu_batch {
    int32_t n_tint;
    const int64_t *row_off;    /* n_tint + 1 */
    const int32_t *n_seg;      /* n_tint: M_t */
    const int64_t *bits_off;   /* n_tint + 1, in uint32 words */
    const uint32_t *bits;
    const int32_t *first;      /* row_off[n_tint] */
    const int32_t *last;
    const uint8_t *tail;
    const int64_t *adj_off;    /* n_tint + 1, in uint64 words */
} fclu_batch;

/* Compatibility graph of every tint of the batch.  adj_out: adj_off[n_tint] uint64 words (caller-owned).
 * rounds_out (may be NULL): per tint, the number of pruning passes that removed at least one edge. */
int fclu_compat_graph(fclu_ctx *c, const fclu_batch *b, int32_t prune, uint64_t *adj_out, int32_t *rounds_out);

/* Duration of the kernels of the last fclu_compat_graph() call, from HIP events on the library's stream (ms). */
int fclu_last_timing(fclu_ctx *c, float *compat_ms, float *prune_ms);

#ifdef __cplusplus
}
#endif
#endif
