// seg_kernels.h -- declarations of every kernel of libfreddie_seg.so (defined in the stage families' translation units, launched
// from freddie_seg.hip).  Template kernels are instantiated where they are defined (the table at the end of each unit).
#pragma once
#include "seg_common.h"

namespace fseg {

// seg_front.hip
__global__ void __launch_bounds__(256) k_thr_table(const double *h_table, int h_len, double tau, int2 *tab);

// seg_front.hip
__global__ void __launch_bounds__(512) k_hist(int n_chunks, const int *chunk_part, const i64 *chunk_p0, const int *chunk_n,
                                              const int *chunk_glo, const int *chunk_ghi, const i64 *chunk_lane_lo,
                                              const i64 *chunk_lane_hi, const i64 *part_iv_off,
                                              const int *iv_start, const int *iv_end, const i64 *pos_off,
                                              const i64 *part_lane_off, const int2 *__restrict__ lane_lx, const int *lane_start,
                                              const int *lane_pmax, const int2 *__restrict__ lex,
                                              int ignore_ends, int *y_raw, Status *st, u64 *zero_ptr, i64 zero_n);

// seg_front.hip
template <int R>
#ifndef FSEG_SMOOTH_OCC
#define FSEG_SMOOTH_OCC 6
#endif
__global__ void __launch_bounds__(kSmoothThreads, FSEG_SMOOTH_OCC) k_smooth(int n_tiles, const TileDesc *__restrict__ tiles,
                                                const int *__restrict__ y_raw, const double *__restrict__ w_g, int radius_rt,
                                                double *y_out, unsigned *flag_pos, unsigned *flag_cand, int *blk_pre, int *tile_tot,
                                                int *tile_defer);

// seg_front.hip
__global__ void __launch_bounds__(256) k_clear(Status *st, unsigned *bits, i64 n_words);

// seg_front.hip
__global__ void __launch_bounds__(256) k_scan1(const unsigned *flags, i64 n, int *bsum);

// seg_front.hip
__global__ void __launch_bounds__(kScan2Threads) k_scan2(int *bsum, i64 nb, u64 *total_out, i64 *off_last /* may be null */);

// seg_front.hip
template <int MODE>
__global__ void __launch_bounds__(256) k_scan_emit(const unsigned *flags, i64 n, const int *bsum /* or null */,
                                                   u64 *state, u64 *total_out, i64 *off_last /* may be null */,
                                                   unsigned *err, const double *y,
                                                   double *v, i64 K, const i64 *pos_off, const int *iv_start,
                                                   const int *blk_iv0, int *out_y, int *out_pos, i64 *out_off,
                                                   int force_stall /* tests: report a look-back stall */, int *out_iv /* may be null */);

// seg_front.hip
__global__ void __launch_bounds__(64) k_voff(int n_part, const i64 *part_iv_off, const i64 *pos_off, i64 n_pos,
                                             const unsigned *flags, const int *bsum /* or null */, const u64 *state, const u64 *total, i64 *voff);

// seg_front.hip
__global__ void __launch_bounds__(256) k_vplan(int n_part, const i64 *part_iv_off, const i64 *pos_off, i64 n_pos,
                                               i64 *voff, i64 *chunk_off, Status *st, i64 chunk_cap);

// seg_front.hip
__global__ void __launch_bounds__(512) k_vsum_chunks(int n_part, const i64 *voff, const i64 *chunk_off, const double *v,
                                                     const double *csum0, int pass, double *csum, i64 chunk_cap);

// seg_front.hip
__global__ void __launch_bounds__(512) k_thr_part(int n_part, const i64 *part_iv_off, const i64 *pos_off, i64 n_pos, const unsigned *flags,
                                                  const double *__restrict__ y, double *v, double vf, double *mean, double *thr);

// seg_front.hip
__global__ void k_vsum_part(int n_part, const i64 *voff, const i64 *chunk_off, const double *csum0, const double *csum1,
                            double vf, double *mean, double *thr, i64 chunk_cap);

// seg_front.hip
__global__ void __launch_bounds__(256) k_peaks_edges(int n_tiles, const TileDesc *tiles, const int *tile_defer, const double *x,
                                                     unsigned *flag, int *part_has2, int n_part);

// seg_problems.hip
__global__ void k_fix(i64 K, const i64 *pos_off, const int *iv_part, const i64 *cand_off, const int *cand_y,
                      const double *yv, const double *thr_part, int mps, unsigned char *fixed0, unsigned char *added,
                      unsigned char *fixed, unsigned char *chosen, int *cand_pn, int *cand_iv, Status *st);

// seg_problems.hip
__global__ void __launch_bounds__(kRangeThreads) k_prob_range(Status *st, const int *cand_pn, const int *cand_iv, const int *cand_y,
                             const int *iv_part, const int *iv_start, const i64 *part_lane_off, const int *lane_start,
                             const int *lane_pmax, int *cand_ll, int *cand_ln, unsigned char *cand_wide,
                             const int2 *__restrict__ lane_lx, const int2 *__restrict__ lex, int wide_by_seen, int fuse_lanes,
                             i64 *bs, ProbSplit sp);

// seg_problems.hip
__global__ void __launch_bounds__(256) k_prob_scan1(Status *st, const int *cand_pn, const int *cand_ln, const unsigned char *cand_wide, i64 *bs, ProbSplit sp);

// seg_problems.hip
__global__ void __launch_bounds__(256) k_prob_scan2(Status *st, i64 *bs);

// seg_problems.hip
__global__ void __launch_bounds__(256) k_prob_emit(Status *st, const int *cand_pn, const int *cand_ll, const int *cand_ln,
                                                   const int *cand_iv, const i64 *cand_off, const i64 *bs,
                                                   ProblemArrays pr, i64 prob_cap, int2 *work_pc, int4 *cls_items,
                                                   i64 work_cap, int *dp_items, ProbDesc *desc, const int *iv_start,
                                                   const int *iv_part, const i64 *part_lane_off, ProbSplit sp, int *solve_items,
                                                   ProbDesc *solve_desc, int *wide_items, int *wide_all, const unsigned char *cand_wide,
                                                   SyncWords *sw, unsigned sync_gen);

// seg_score_arena.hip
__global__ void __launch_bounds__(256) k_pair_thresholds(const Status *st, ProblemArrays pr, const ProbDesc *desc, i64 prob_cap,
                                                         const i64 *cand_off, const int *cand_y, const double *h_table,
                                                         int h_len, double tau, int2 *pair_thr, i64 pair_cap,
                                                         unsigned *amb_g, unsigned *out_g, i64 tri_cap);

// seg_score_arena.hip
__global__ void __launch_bounds__(kLaneChunk) k_cov(Status *st, const ProbDesc *desc, i64 prob_cap, const int2 *work_pc,
                                                    i64 work_cap, const i64 *cand_off,
                                                    const int *cand_y, const int *iv_start, const int2 *__restrict__ lane_lx,
                                                    const int2 *__restrict__ lex,
                                                    unsigned *cov_g, i64 cov_cap, unsigned char *work_active,
                                                    int cov_blocks, ProblemArrays pr, const double *h_table, int h_len, double tau,
                                                    int2 *pair_thr, i64 pair_cap, unsigned *amb_g, unsigned *out_g, i64 tri_cap);

// seg_score_arena.hip
template <int NM>
__global__ void __launch_bounds__(ScoreCfg<NM>::kThreads) k_score(Status *st, int cls, int nm, ProblemArrays pr, i64 prob_cap,
                                                                  const int4 *cls_items, const ProbDesc *desc,
                                                                  i64 work_cap, const i64 *cand_off,
                                                                  const int *cand_y, const unsigned char *work_active,
                                                                  const unsigned *cov_g, i64 cov_cap, const int2 *pair_thr,
                                                                  i64 pair_cap, unsigned *out_g, i64 tri_cap,
                                                                  unsigned *amb_g FSEG_TPARAM);

// seg_score_arena.hip
template <int NM, int T, typename OutT>
__global__ void __launch_bounds__(T) k_dp(Status *st, int dp_class, int nm, const int *dp_items, ProblemArrays pr, const ProbDesc *desc, i64 prob_cap,
                                            const i64 *cand_off, const int *cand_y, const int *iv_part,
                                            const i64 *part_lane_off, const unsigned *out_g, i64 tri_cap,
                                            const unsigned *amb_g, const int2 *pair_thr, i64 pair_cap, int support,
                                            unsigned char *chosen, int n_lo, int dp_blocks, uint4 *labels16, i64 labels_n16 FSEG_TPARAM);

// seg_score_arena.hip
template <typename OutT>
__global__ void __launch_bounds__(256, 6) k_dp_waves(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc, i64 prob_cap,
                                                  const int *cand_y, const unsigned *out_g, i64 tri_cap, const unsigned *amb_g,
                                                  const int2 *pair_thr, i64 pair_cap, int support, unsigned char *chosen, int coop);

// seg_score_fused.hip
#ifndef FSEG_TINY_OCC
#define FSEG_TINY_OCC 5
#endif
__global__ void __launch_bounds__(256, FSEG_TINY_OCC) k_tiny(Status *st, const ProbDesc *desc, i64 prob_cap, int tiny_max, ProblemArrays pr,
                                              const int *cand_y, const longlong2 *lane_ex, const int *ex_ts,
                                              const int *ex_te, const double *h_table, int h_len, double tau, const int2 *thr_tab,
                                              int support, unsigned char *chosen, i64 lb_h, i64 ln_h FSEG_TPARAM);

// seg_score_fused.hip
template <int NM, typename V>
__global__ void __launch_bounds__(256, WaveCfg<NM>::kOcc) k_wave(Status *st, const ProbDesc *desc, i64 prob_cap, int list, ProblemArrays pr,
                                                                 const int *__restrict__ cand_y, const int2 *__restrict__ lane_lx, const int2 *__restrict__ lex,
                                                                 const double *h_table, int h_len, double tau, const int2 *__restrict__ thr_tab,
                                                                 int support, unsigned char *chosen, i64 lb_h, i64 ln_h FSEG_TPARAM);

// seg_score_fused.hip
__global__ void __launch_bounds__(64) k_gate(Status *st, int which, unsigned grid, unsigned max_ticks);

// seg_score_fused.hip
__global__ void __launch_bounds__(64) k_wait_word(Status *st, const unsigned *words, unsigned mask, unsigned gen, unsigned max_ticks);

// seg_score_fused.hip
__global__ void __launch_bounds__(64) k_signal(unsigned *word, unsigned gen);
__global__ void __launch_bounds__(64) k_probe_wait(const unsigned *word, unsigned gen, unsigned *result, unsigned max_ticks);

// seg_score_fused.hip
template <int NM, typename CntT, typename V, bool SPLIT>
__global__ void __launch_bounds__(SolveCfg<NM>::kThreads, SolveCfg<NM>::kMinBlocks) k_solve(Status *st, int cls, int nm, i64 lb_h, i64 ln_h, ProblemArrays pr,
                                                                  const ProbDesc *desc, i64 prob_cap, const int *cand_y,
                                                                  const int2 *__restrict__ lane_lx, const int2 *__restrict__ lex,
                                                                  const double *h_table, int h_len, double tau, const int2 *thr_tab,
                                                                  int support, unsigned char *chosen,
                                                                  unsigned char *dpx, i64 dpx_stride,
                                                                  const int *__restrict__ wide_items FSEG_TPARAM);

// seg_score_fused.hip
template <int NM, typename OutT, typename V, int T>
__global__ void __launch_bounds__(T) k_dpw(Status *st, int nm, i64 list_base, i64 list_n, ProblemArrays pr, const ProbDesc *desc,
                                            const unsigned char *dpx, i64 dpx_stride,
                                            int support, unsigned char *chosen, const int *__restrict__ wide_items FSEG_TPARAM);

// seg_score_arena.hip
__global__ void __launch_bounds__(512) k_score_huge(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc,
                                                    i64 prob_cap, i64 work_cap, const int *cand_y, const unsigned *cov_g,
                                                    i64 cov_cap, const int2 *pair_thr, i64 pair_cap, unsigned *out_g,
                                                    i64 tri_cap, unsigned *amb_g);

// seg_score_arena.hip
__global__ void __launch_bounds__(512) k_dp_huge(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc,
                                                 i64 prob_cap, const int *cand_y, const unsigned *out_g, i64 tri_cap,
                                                 const unsigned *amb_g, const int2 *pair_thr, i64 pair_cap, int support,
                                                 unsigned char *chosen);

// seg_score_arena.hip
__global__ void __launch_bounds__(512) k_score_giant(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc,
                                                     i64 prob_cap, i64 work_cap, const int *cand_y, const unsigned *cov_g,
                                                     i64 cov_cap, const int2 *pair_thr, i64 pair_cap, unsigned *out_g,
                                                     i64 tri_cap, unsigned *amb_g, int nm, unsigned char *scratch, i64 scratch_stride);

// seg_score_arena.hip
__global__ void __launch_bounds__(512) k_dp_giant(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc,
                                                  i64 prob_cap, const int *cand_y, const unsigned *out_g, i64 tri_cap,
                                                  const unsigned *amb_g, const int2 *pair_thr, i64 pair_cap, int support,
                                                  unsigned char *chosen, int nm, unsigned char *scratch, i64 scratch_stride);

// seg_tail.hip
__global__ void k_segments(i64 K, const i64 *pos_off, const i64 *cand_off, const int *cand_y, const int *__restrict__ y_raw,
                           const int *__restrict__ blk_pre, const int *tile_tot, const int *iv_tile0, const unsigned char *chosen, unsigned *final_flag, int *rseg_c, int *rseg_prev,
                           Status *st);

// seg_tail.hip
__global__ void __launch_bounds__(64) k_refine(const Status *st, const int *cand_iv, const int *rseg_c,
                                               const int *rseg_prev, const int *cand_y, const i64 *pos_off,
                                               const int *y_raw, const double *w_g, int radius, double sigma,
                                               double *g_scr, int *pk_scr, unsigned char *flag_scr,
                                               unsigned char *keep_scr, unsigned *final_flag);

// seg_tail.hip
__global__ void __launch_bounds__(256) k_label_cols(i64 K, const i64 *final_off, const int *final_y, const int *final_iv, const int *iv_part,
                                                    const double *h_table, int h_len, double tau, const int2 *thr_tab, int2 *col_thr,
                                                    unsigned char *col_zero, int *part_has2, int n_part,
                                                    const i64 *part_iv_off, const i64 *part_rep_off, i64 *label_off,
                                                    Status *st, i64 label_cap);

// seg_tail.hip
__global__ void __launch_bounds__(256) k_label_zero(uint4 *labels16, i64 n16);

// seg_tail.hip
__global__ void __launch_bounds__(256) k_label_reads(int n_blocks, const int *rb_part, const int *rb_r0,
                                                     const i64 *label_off, i64 label_cap, int n_part,
                                                     const i64 *part_iv_off, const i64 *part_rep_off,
                                                     const i64 *final_off, const int *final_pos, const int2 *col_thr,
                                                     const i64 *rep_exon_off, const int *ex_ts, const int *ex_te,
                                                     const unsigned char *col_zero, const int *part_has2,
                                                     unsigned char *labels, unsigned *packed);
__global__ void __launch_bounds__(256) k_unpack_labels(const unsigned *__restrict__ packed, uint4 *__restrict__ labels16, i64 n16);

// seg_tail.hip
__global__ void __launch_bounds__(256) k_pack_labels(const uint4 *__restrict__ labels16, unsigned *__restrict__ packed, i64 n16);

// seg_upload.hip
__global__ void __launch_bounds__(256) k_prep_reps(int n_blocks, const int *rb_part, const int *rb_r0, const i64 *part_rep_off,
                                                   const i64 *part_iv_off, const int *iv_start, const int *iv_end,
                                                   const i64 *rep_exon_off, const int *ex_ts, const int *ex_te, u64 *key,
                                                   int *val, int *rep_last, PrepStatus *ps);

// seg_upload.hip
__global__ void __launch_bounds__(256) k_lanes(int n_part, const i64 *part_rep_off, const i64 *part_lane_off, const u64 *key_sorted,
                                               const int *val_sorted, const int *rep_weight, const int *rep_last,
                                               const i64 *rep_exon_off, longlong2 *lane_ex, int *lane_start, int *lane_pmax,
                                               int sort_here, const u64 *key_unsorted, const int *ex_ts, const int *ex_te,
                                               int2 *lane_lx, int2 *lex);

// seg_upload.hip
__global__ void __launch_bounds__(256) k_lane_blocks(int n_blocks, const int *rb_part, const int *rb_r0, const i64 *part_rep_off,
                                                     const int *val_sorted, const int *rep_weight, const int *rep_last,
                                                     i64 *rb_sum, int *rb_max, const i64 *rep_exon_off, i64 *rb_esum);

// seg_upload.hip
__global__ void __launch_bounds__(256) k_lane_block_scan(int n_part, int n_blocks, const int *rb_part, const i64 *part_lane_off,
                                                         const i64 *rb_sum, const int *rb_max, i64 *rb_base, int *rb_cmax,
                                                         const i64 *part_rep_off, const i64 *rep_exon_off, const i64 *rb_esum, i64 *rb_ebase);

// seg_upload.hip
__global__ void __launch_bounds__(256) k_lane_emit(int n_blocks, const int *rb_part, const int *rb_r0, const i64 *part_rep_off,
                                                   const u64 *key_sorted, const int *val_sorted, const int *rep_weight, const int *rep_last,
                                                   const i64 *rep_exon_off, const i64 *rb_base, const int *rb_cmax,
                                                   longlong2 *lane_ex, int *lane_start, int *lane_pmax, const i64 *rb_ebase,
                                                   const int *ex_ts, const int *ex_te, int2 *lane_lx, int2 *lex);

// seg_upload.hip
__global__ void __launch_bounds__(256) k_hist_ranges(int n_chunks, const int *hc_part, const int *hc_glo, const int *hc_ghi,
                                                     const i64 *part_lane_off, const int *lane_start, const int *lane_pmax,
                                                     i64 *hc_llo, i64 *hc_lhi);

}  // namespace fseg
