/*
 * freddie_seg.h -- C-ABI of the MI355X (gfx950) canonical-segmentation library.
 *
 * The reference (vpc-ccg/freddie) has no FFI: its stage boundary is the process CLI
 * (py/freddie_segment.py:53-110, :847-885) and, in-process, the function
 *
 *     segment(tint, sigma, smoothed_threshold, threshold_rate, variance_factor,
 *             max_problem_size, min_read_support_outside, ignore_ends)     py/freddie_segment.py:738-747
 *
 * which consumes tint['intervals'], tint['read_reps'] (py/freddie_segment.py:127-135,:165-170)
 * and produces tint['final_positions'] (:761,:806) and one 0/1/2 label list per read rep
 * (:815-830).  This header is the batched, flat-array form of that seam: what a ctypes /
 * cgo / JNI binding of the segmentation path binds.  No torch types, no C++ types; every
 * call returns an int status (0 = ok) and never throws or aborts; the message of the last
 * failure of a context is available through fseg_last_error().
 *
 * Threading: a context belongs to one host thread; different contexts (one per GPU) may
 * be used concurrently.  All device work of a context runs on its own HIP stream.
 */
#ifndef FREDDIE_SEG_H
#define FREDDIE_SEG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSEG_ABI_VERSION 2

/* status codes */
#define FSEG_OK 0
#define FSEG_ERR_ARG 1        /* bad argument / state */
#define FSEG_ERR_HIP 2        /* a HIP runtime call failed */
#define FSEG_ERR_INPUT 3      /* input violates an invariant the reference asserts (see message) */
#define FSEG_ERR_UNSUPPORTED 4

typedef struct fseg_ctx fseg_ctx;

/* The scalar arguments of segment() (py/freddie_segment.py:738-747) plus the two tables the
 * reference obtains from numpy/scipy/Python at run time and that must be supplied as data:
 *   w_main   = Gaussian half kernel (centre first) of gaussian_filter1d(y, sigma, truncate=4.0)   (:755)
 *   w_refine = the same for gaussian_filter1d(..., mode='constant', truncate=1.0)                 (:260-261)
 *   h_table  = smooth_threshold(threshold_rate)                                                   (:277-286, :864) */
typedef struct {
    double sigma;
    double threshold_rate;
    double variance_factor;
    int32_t max_problem_size;
    int32_t min_read_support_outside;
    int32_t ignore_ends;
    int32_t radius_main;
    const double *w_main;        /* radius_main + 1 values */
    int32_t radius_refine;
    const double *w_refine;      /* radius_refine + 1 values */
    int32_t h_len;
    const double *h_table;       /* h_len values */
} fseg_params;

/* A batch of independent partitions ("tints", one split_<contig>_<tint>.tsv each) as flat
 * CSR arrays -- the flat form of the tint dict (py/freddie_segment.py:127-135):
 *   intervals of partition p : iv_start/iv_end[part_iv_off[p] .. part_iv_off[p+1])   tint['intervals']
 *   read reps of partition p : [part_rep_off[p] .. part_rep_off[p+1])               tint['read_reps']
 *   rep r: multiplicity rep_weight[r] (= len(read indices), :797), exons
 *          (ex_ts[e], ex_te[e]) for e in [rep_exon_off[r] .. rep_exon_off[r+1])     the rep's key tuple (:166)
 * Coordinates are the genomic integers of the split file.  All pointers are host pointers. */
typedef struct {
    int32_t n_part;
    const int64_t *part_iv_off;   /* n_part + 1 */
    const int32_t *iv_start;
    const int32_t *iv_end;
    const int64_t *part_rep_off;  /* n_part + 1 */
    const int32_t *rep_weight;
    const int64_t *rep_exon_off;  /* n_rep + 1 */
    const int32_t *ex_ts;
    const int32_t *ex_te;
} fseg_batch;

/* Sizes of the result of the last fseg_run(). */
typedef struct {
    int64_t n_final;       /* total number of final positions (sum over partitions)            */
    int64_t label_bytes;   /* sum over partitions of n_rep_p * (n_final_p - 1)                  */
    int64_t n_cand;        /* total candidates (debug)                                          */
    int64_t n_problems;    /* DP problems with at least 3 candidates (debug)                    */
    int64_t n_positions;   /* sum of interval lengths (debug)                                   */
    int64_t max_problem_size;   /* candidates of the largest DP problem (debug)                 */
    int64_t max_problem_reads;  /* reads examined by the widest DP problem (debug; >= 65536 selects 32-bit DP counts) */
} fseg_sizes;

int fseg_abi_version(void);
/* Hash of the sources this binary was built from (freddie_amd/build.py compares it with the tree's). */
const char *fseg_source_hash(void);

/* Create / destroy a context bound to HIP device `device`.  Fails (FSEG_ERR_HIP) when no
 * usable GPU is present: there is no CPU fallback. */
int fseg_create(int device, fseg_ctx **out);
void fseg_destroy(fseg_ctx *ctx);
const char *fseg_last_error(const fseg_ctx *ctx);   /* never NULL; ctx may be NULL after a failed create */

int fseg_set_params(fseg_ctx *ctx, const fseg_params *params);

/* Copy a batch into HBM (replaces the resident batch): one pinned staging image, one host-to-device copy; the
 * per-read preparation (validation, ordering of the reads by first position) runs on the device behind the copy and
 * the call returns without waiting for it.  Arrays are validated the way read_split() asserts them
 * (py/freddie_segment.py:138-140,:158-161); what the device-side part of that finds is reported by the next call
 * that waits for the device (fseg_run, fseg_sync, ...). */
int fseg_upload(fseg_ctx *ctx, const fseg_batch *batch);

/* Run the whole segmentation path over the resident batch: splice histogram (:648-678),
 * Gaussian smoothing (:755), variance threshold (:757-759), candidates (:615-621), fixing and
 * problem splitting (:776-788,:623-645), interval scoring + DP (:475-596), refinement
 * (:249-266), final positions (:802-807) and labels (:808-830).  Results stay in HBM until fetched.
 * The first run of a batch waits twice for the device inside the call (it reads the sizes of the problem list and
 * of the label matrix to size their arenas exactly); its last stage, and every later run of the same batch (replayed
 * as a hipGraph), is asynchronous until fseg_sync()/fseg_get_sizes()/fseg_results()/fseg_download(). */
int fseg_run(fseg_ctx *ctx);
int fseg_sync(fseg_ctx *ctx);

int fseg_get_sizes(fseg_ctx *ctx, fseg_sizes *out);

/* Results, zero-copy: pointers into the context's own pinned host buffers, filled by one device-to-host copy per
 * array (layout as for fseg_download below).  They stay valid until the next fseg_results / fseg_results_packed call
 * on this context (or fseg_destroy): the pinned result buffers are written by these two calls and by nothing else, so
 * fseg_upload and fseg_run of the NEXT batch may be issued while a writer thread still reads the previous batch's
 * results -- which is what the CLI's pipeline does (freddie_amd/segment.py: upload + run of batch i+1, then wait for
 * the writer of batch i, then fseg_results_packed).  fseg_download copies into caller memory instead.  Any pointer may
 * be NULL. */
int fseg_results(fseg_ctx *ctx, const int64_t **part_final_off, const int32_t **final_pos, const int64_t **label_off,
                 const uint8_t **labels);

/* The same with the label matrix at two bits per label (it is most of what crosses PCIe: about 300 labels per read):
 * label byte g of the arena (g as counted by label_off) is bits 2(g & 3) .. 2(g & 3)+1 of labels2[g >> 2], values 0 / 1 / 2.
 * fhost_write_packed() (include/freddie_host.h) takes this form. */
int fseg_results_packed(fseg_ctx *ctx, const int64_t **part_final_off, const int32_t **final_pos, const int64_t **label_off,
                        const uint8_t **labels2);

/* Results.  part_final_off: n_part+1 offsets into final_pos; final_pos: n_final genomic
 * positions (tint['final_positions'] of each partition, concatenated).  labels: for partition
 * p, rep r (local index) the bytes labels[label_off[p] + r*(F_p-1) ..] are the ASCII digits
 * '0','1','2' of the rep's label list (the 6th column of segment_*.tsv); label_off has
 * n_part+1 entries.  Any pointer may be NULL to skip that array. */
int fseg_download(fseg_ctx *ctx, int64_t *part_final_off, int32_t *final_pos, int64_t *label_off, uint8_t *labels);

/* Debug taps: intermediates of the last run, for parity tests.  `what` selects the array;
 * dst receives up to cap_bytes; *n_bytes is set to the full size in bytes. */
enum {
    FSEG_TAP_POS_OFF = 1,     /* int64[K+1]   offsets of each interval's positions                         */
    FSEG_TAP_Y_RAW = 2,       /* int32[P]     splice histogram                                              */
    FSEG_TAP_Y = 3,           /* double[P]    smoothed signal                                               */
    FSEG_TAP_THRESHOLD = 4,   /* double[n_part]                                                             */
    FSEG_TAP_CAND_OFF = 5,    /* int64[K+1]                                                                 */
    FSEG_TAP_CAND_Y = 6,      /* int32[n_cand] candidate y indices                                          */
    FSEG_TAP_FIXED = 7,       /* uint8[n_cand] 1 = fixed after break_large_problems                         */
    FSEG_TAP_CHOSEN = 8,      /* uint8[n_cand] 1 = in final_c_idxs after run_optimize                       */
    FSEG_TAP_FINAL_OFF = 9,   /* int64[K+1]                                                                 */
    FSEG_TAP_FINAL_Y = 10,    /* int32[n_final] final y indices per interval                                */
    FSEG_TAP_PROBLEMS = 11,   /* int32[n_problems][4] = (interval, start, n, chain_len)                     */
    /* the upload's device-side preparation: every rep repeated rep_weight times ("lanes"), ordered inside its
     * partition by first position */
    FSEG_TAP_LANE_START = 12, /* int32[lanes]  first position of the lane's rep                                      */
    FSEG_TAP_LANE_PMAX = 13,  /* int32[lanes]  running maximum (inside the partition) of the reps' last positions    */
    FSEG_TAP_LANE_EXONS = 14, /* int64[lanes][2] exon range (into ex_ts / ex_te) of the lane's rep                   */
    FSEG_TAP_LANE_STREAM = 15, /* int32[lanes][2] the lane's exon range in the lane-ordered exon stream                */
    FSEG_TAP_EXON_STREAM = 16, /* int32[I][2]    (ts, te) of every exon, the reps of a partition in lane order        */
    /* the scoring stage's device-side fork / join: stages enqueued with waiters so far, whether waiters are still in use
     * (0 after one timed out: events from then on), and the device's words (last problem-list generation, last generation of
     * side streams 0 / 1, workgroup counter) */
    FSEG_TAP_SYNC = 17         /* int32[6]                                                                             */
};
int fseg_tap(fseg_ctx *ctx, int what, void *dst, int64_t cap_bytes, int64_t *n_bytes);

/* Timing support for benchmarks: HIP-event time of the last run in milliseconds, per stage.
 * Stage names are returned by fseg_stage_name(i); n_stages by fseg_n_stages().  Only filled
 * when profiling was enabled with fseg_set_profiling(ctx, 1) (adds event records: two per stage; on replays of a resident
 * batch only the interval-scoring stage and the two graphs around it are bracketed); with 2 only the interval-scoring
 * stage is bracketed (two records per run; the other stages report 0); with 3 every stage is bracketed on replays too
 * (plain launches instead of the graph: what bench.py's config2 leg sums coverage + scoring + DP from). */
int fseg_set_profiling(fseg_ctx *ctx, int on);
int fseg_n_stages(void);
const char *fseg_stage_name(int i);
int fseg_stage_ms(fseg_ctx *ctx, float *ms /* n_stages */);
/* Algorithmic bytes of the interval-scoring stage for the resident batch after a run:
 * 4*(N+K)*R summed over partitions + 4*R. */
int64_t fseg_scoring_algorithmic_bytes(fseg_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
