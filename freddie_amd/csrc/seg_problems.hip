// seg_problems.hip -- S4 fixing / break_large_problems (:623-645, :776-788) and the problem list (ranges of reads, sizes, records, solve lists).
// Part of libfreddie_seg.so (gfx950); shared definitions: seg_common.h, declarations: seg_kernels.h, launches: freddie_seg.hip.
#include "seg_kernels.h"

namespace fseg {


__global__ void k_fix(i64 K, const i64 *pos_off, const int *iv_part, const i64 *cand_off, const int *cand_y,
                      const double *yv, const double *thr_part, int mps, unsigned char *fixed0, unsigned char *added,
                      unsigned char *fixed, unsigned char *chosen, int *cand_pn, int *cand_iv, Status *st) {
    __shared__ int lds[16];
    const int T = blockDim.x;
    // (Round 5, measured and not kept: eight intervals per wave -- lanes 8g .. 8g+7 the candidates of interval kb + g -- while an interval
    // holds at most eight candidates (most hold two or three) and the longer ones one by one behind them: 17 us against 14, the longer
    // intervals of a group then run one after the other in ONE wave instead of side by side in their own.)
    for (i64 k = blockIdx.x; k < K; k += gridDim.x) {
        i64 c0 = cand_off[k];
        int N = (int)(cand_off[k + 1] - c0);
        const double *y = yv + pos_off[k];
        const int *cy = cand_y + c0;
        double thr = thr_part[iv_part[k]];
        if (T == 64 && N <= 64 && N <= mps) {
            // short interval, one wave: no gap can exceed max_problem_size, so the fixed set is final at once and the
            // previous fixed candidate comes from the ballot -- three dependent load rounds, no barrier
            const int c = threadIdx.x;
            const bool in = c < N;
            const bool f = in && (c == 0 || c == N - 1 || y[cy[c]] > thr);
            const u64 mask = __ballot(f);
            int n = 0;
            if (in) {
                const u64 below = mask & ((1ULL << c) - 1ULL);
                const int prev = below ? 63 - __clzll((long long)below) : -1;
                n = (f && prev >= 0 && c - prev + 1 >= 3) ? c - prev + 1 : 0;
                fixed0[c0 + c] = f; added[c0 + c] = 0; cand_iv[c0 + c] = (int)k;
                fixed[c0 + c] = f; chosen[c0 + c] = f;
                cand_pn[c0 + c] = n;
            }
            continue;
        }
        for (int c = threadIdx.x; c < N; c += T) {
            fixed0[c0 + c] = (c == 0 || c == N - 1 || y[cy[c]] > thr) ? 1 : 0;
            added[c0 + c] = 0;
            cand_iv[c0 + c] = (int)k;
        }
        __syncthreads();
        // break_large_problems over the original consecutive fixed pairs; the thread that owns the
        // right end of an oversized gap places its anchors
        int carry = -1;
        for (int t0 = 0; t0 < N; t0 += T) {
            int c = t0 + threadIdx.x;
            bool f = c < N && fixed0[c0 + c];
            int prev = wg_prev_flagged(f, c, carry, lds);
            if (f && prev >= 0) {
                int size = c - prev + 1;
                if (size > mps) {
                    int cnt = (int)ceil((double)size / (double)mps);
                    double step = (double)size / (double)cnt;
                    for (int i = 1; i < cnt; ++i) {
                        int anchor = (int)((double)prev + __dmul_rn((double)i, step));
                        double best = -INFINITY;
                        int best_c = -1;
                        bool bad = false;
                        for (int cc = anchor - 5; cc < anchor + 5; ++cc) {
                            int ci = cc < 0 ? cc + N : cc;          // Python negative-index wraparound
                            if (ci < 0 || ci >= N) { bad = true; continue; }
                            double val = y[cy[ci]];
                            if (val > best) { best = val; best_c = cc; }
                        }
                        if (bad || !(best > 0.0) || best_c < 0) atomicOr(&st->err, kErrBreakAssert);
                        else added[c0 + best_c] = 1;
                    }
                }
            }
        }
        __syncthreads();
        // final fixed set; the right end of every problem records the problem's size
        carry = -1;
        for (int t0 = 0; t0 < N; t0 += T) {
            int c = t0 + threadIdx.x;
            bool f = c < N && (fixed0[c0 + c] | added[c0 + c]);
            int prev = wg_prev_flagged(f, c, carry, lds);
            if (c < N) {
                fixed[c0 + c] = f; chosen[c0 + c] = f;
                int n = (f && prev >= 0 && c - prev + 1 >= 3) ? c - prev + 1 : 0;
                if (n > kNGiant) atomicOr(&st->err, kErrProblemTooLarge);
                cand_pn[c0 + c] = n;
            }
        }
        __syncthreads();
    }
}

// bs != nullptr: the workgroup's 1 024 candidates are a block of the problem scan, and it adds up the block's sizes itself when its
// ranges are known -- what k_prob_scan1 (25 us of a config4 batch, most of it 147 workgroups starting, loading and ending) then
// need not do.
__global__ void __launch_bounds__(kRangeThreads) k_prob_range(Status *st, const int *cand_pn, const int *cand_iv, const int *cand_y,
                             const int *iv_part, const int *iv_start, const i64 *part_lane_off, const int *lane_start,
                             const int *lane_pmax, int *cand_ll, int *cand_ln, unsigned char *cand_wide,
                             const int2 *__restrict__ lane_lx, const int2 *__restrict__ lex, int wide_by_seen, int fuse_lanes,
                             i64 *bs, ProbSplit sp) {
    static_assert(kRangeThreads == kProbBlock, "a workgroup of k_prob_range is a block of the problem scan");
    __shared__ int l_wide[kRangeThreads], l_n, l_red[kRangeThreads / 64];
    __shared__ i64 scan_lds[4 * kProbCols];
    __shared__ int l_mx[8];
    const i64 n_cand = (i64)st->n_cand;
    for (i64 c0 = (i64)blockIdx.x * blockDim.x; c0 < n_cand; c0 += (i64)gridDim.x * blockDim.x) {    // (workgroup-uniform)
        const i64 c = c0 + threadIdx.x;
        if (threadIdx.x == 0) l_n = 0;
        __syncthreads();
        if (c < n_cand) {
            int n = cand_pn[c];
            int lo_lane = 0, n_lanes = 0;
            if (n > 0) {
                int k = cand_iv[c];
                int part = iv_part[k];
                int g0 = iv_start[k] + cand_y[c - (n - 1)], g1 = iv_start[k] + cand_y[c];
                i64 a = part_lane_off[part], L1 = part_lane_off[part + 1], b = L1;
                while (a < b) { i64 m = (a + b) >> 1; if (lane_pmax[m] < g0) a = m + 1; else b = m; }
                i64 lo = a;
                b = L1;
                while (a < b) { i64 m = (a + b) >> 1; if (lane_start[m] < g1) a = m + 1; else b = m; }
                lo_lane = (int)lo; n_lanes = (int)(a - lo);
            }
            cand_ll[c] = lo_lane; cand_ln[c] = n_lanes;
            // (fuse_lanes: what a problem may see to be solved whole, -1 in a batch that is not -- there nobody asks, and a
            // batch of deep problems would pay a workgroup's walk over up to 1 023 reads for every one of them)
            const bool cand = n > 0 && n <= kNMax && n_lanes > kFuseLanes && n_lanes <= kFuseLanesWide && n_lanes <= fuse_lanes;
            cand_wide[c] = (unsigned char)((cand && wide_by_seen) ? 1 : 0);     // (FSEG_WIDE_BY_SEEN=1, tests: whatever SEES more than kFuseLanes reads)
            if (cand && !wide_by_seen) l_wide[atomicAdd(&l_n, 1)] = (int)threadIdx.x;
        }
        __syncthreads();
        const int nw = l_n;
        for (int w = 0; w < nw; ++w) {
            const i64 cw = c0 + l_wide[w];
            const int n = cand_pn[cw], k = cand_iv[cw], ll = cand_ll[cw], ln = cand_ln[cw];
            const int cp0 = iv_start[k] + cand_y[cw - (n - 1)], c_last = iv_start[k] + cand_y[cw];
            int kept = 0;
            for (int l = threadIdx.x; l < ln; l += kRangeThreads) {
                int first_rel, cnt;
                window_exons(lex, lane_lx[ll + l], cp0, c_last, &first_rel, &cnt);
                kept += cnt > 0;
            }
            for (int d = 32; d >= 1; d >>= 1) kept += __shfl_xor(kept, d);
            if (lane_id() == 0) l_red[threadIdx.x >> 6] = kept;
            __syncthreads();
            if (threadIdx.x == 0) {
                int tot = 0;
                for (int q = 0; q < kRangeThreads / 64; ++q) tot += l_red[q];
                if (tot > kFuseLanes) cand_wide[cw] = 1;
            }
            __syncthreads();
        }
        if (bs) {                                            // (workgroup-uniform) the block's sums, as k_prob_scan1 makes them
            __syncthreads();                                 // (the block's ranges and wide marks are written)
            const i64 b = c0 / kProbBlock;
            // (the scan's own shape: the first four waves, four candidates a thread; the other waves only meet the barriers)
            const bool scan_t = threadIdx.x < 256;
            const int lane = lane_id(), wave = threadIdx.x >> 6;
            ProbSizes acc;
            if (scan_t) {
                acc = prob_block_sizes(cand_pn, cand_ln, b, n_cand, nullptr, sp);
                prob_block_maxima(st, cand_pn, cand_ln, b, n_cand, l_mx);
                prob_block_wide(st, cand_pn, cand_ln, cand_wide, b, n_cand, sp);
                for (int q = 0; q < kProbCols; ++q) {
                    i64 x = acc.v[q];
                    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
                    if (lane == 0) scan_lds[wave * kProbCols + q] = x;
                }
            }
            __syncthreads();
            if (threadIdx.x < kProbCols)
                bs[b * kProbCols + threadIdx.x] = scan_lds[threadIdx.x] + scan_lds[kProbCols + threadIdx.x] + scan_lds[2 * kProbCols + threadIdx.x] +
                                                  scan_lds[3 * kProbCols + threadIdx.x];
            if (threadIdx.x == 0) prob_publish_maxima(st, l_mx);
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(256) k_prob_scan1(Status *st, const int *cand_pn, const int *cand_ln, const unsigned char *cand_wide, i64 *bs, ProbSplit sp) {
    __shared__ i64 lds[4 * kProbCols];
    __shared__ int l_mx[8];
    i64 n = (i64)st->n_cand;
    i64 nb = (n + kProbBlock - 1) / kProbBlock;
    for (i64 b = blockIdx.x; b < nb; b += gridDim.x) {
        ProbSizes acc = prob_block_sizes(cand_pn, cand_ln, b, n, nullptr, sp), ex, tot;
        prob_block_maxima(st, cand_pn, cand_ln, b, n, l_mx);
        prob_block_wide(st, cand_pn, cand_ln, cand_wide, b, n, sp);
        wg_scan_cols(acc, ex, tot, lds);
        if (threadIdx.x < kProbCols) bs[b * kProbCols + threadIdx.x] = tot.v[threadIdx.x];
        if (threadIdx.x == 0) prob_publish_maxima(st, l_mx);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_prob_scan2(Status *st, i64 *bs) {
    __shared__ i64 lds[4 * kProbCols];
    i64 n = (i64)st->n_cand;
    i64 nb = (n + kProbBlock - 1) / kProbBlock;
    ProbSizes carry;
    for (int q = 0; q < kProbCols; ++q) carry.v[q] = 0;
    for (i64 b0 = 0; b0 < nb; b0 += blockDim.x) {
        i64 b = b0 + threadIdx.x;
        ProbSizes v, ex, tot;
        for (int q = 0; q < kProbCols; ++q) v.v[q] = b < nb ? bs[b * kProbCols + q] : 0;
        wg_scan_cols(v, ex, tot, lds);
        for (int q = 0; q < kProbCols; ++q) {
            if (b < nb) bs[b * kProbCols + q] = carry.v[q] + ex.v[q];
            carry.v[q] += tot.v[q];
        }
    }
    if (threadIdx.x == 0) prob_store_totals(st, carry);
}

__global__ void __launch_bounds__(256) k_prob_emit(Status *st, const int *cand_pn, const int *cand_ll, const int *cand_ln,
                                                   const int *cand_iv, const i64 *cand_off, const i64 *bs,
                                                   ProblemArrays pr, i64 prob_cap, int2 *work_pc, int4 *cls_items,
                                                   i64 work_cap, int *dp_items, ProbDesc *desc, const int *iv_start,
                                                   const int *iv_part, const i64 *part_lane_off, ProbSplit sp, int *solve_items,
                                                   ProbDesc *solve_desc, int *wide_items, int *wide_all, const unsigned char *cand_wide,
                                                   SyncWords *sw, unsigned sync_gen) {
    __shared__ i64 lds[4 * kProbCols];
    __shared__ int l_slot[kProbBlock], l_cnt[kProbBlock];
    __shared__ int l_mx[8];
    __shared__ i64 l_w0[kProbBlock], l_c0[kProbBlock];
    __shared__ int l_n;
    // The solve lists are filled from BOTH ends: a class's expensive problems (many candidates) from the front, the cheap ones
    // from the back.  A class's kernel runs one problem per workgroup, what does not fit the chip at once starts when something
    // else ends, and a 40 us problem that starts late is the kernel's tail: with the expensive half first the dispatcher's
    // in-order placement is a longest-first schedule (k_solve<32> 69 -> 61 us, <60> 73 -> 68, <16> 38 -> 34 on config4).  A
    // workgroup reserves its share of each end with one atomic per end; the order inside an end is whatever the atomics make
    // it -- problems do not depend on one another.
    __shared__ int l_cnt2[8], l_base2[8], l_cur2[8];
    auto list_end_of = [&](int nn, int lanes) -> int {            // 2 * list + (cheap ? 1 : 0), or -1: not in a solve list
        const int kind = prob_kind(nn, lanes, sp);
        if (kind == kKindTiny) return 6 + (nn >= 5 ? 0 : 1);
        if (kind != kKindFused) return -1;
        const int sc = size_class(nn);
        return 2 * sc + (nn >= (sc == 0 ? 12 : (sc == 1 ? 23 : 42)) ? 0 : 1);
    };
    i64 n = (i64)st->n_cand;
    i64 nb = (n + kProbBlock - 1) / kProbBlock;
    for (i64 b = blockIdx.x; b < nb; b += gridDim.x) {
        ProbSizes sz[4], ex, tot, before, grand;
        if (threadIdx.x == 0) l_n = 0;
        if (threadIdx.x < 8) { l_cnt2[threadIdx.x] = 0; l_cur2[threadIdx.x] = 0; }
        if (bs) {
            for (int q = 0; q < kProbCols; ++q) before.v[q] = bs[b * kProbCols + q];
            grand.v[4] = (i64)st->cls_work[0] + ((i64)st->cls_work[1] << 32); grand.v[0] = (i64)st->dp_cls[0] << 32;
            grand.v[5] = (i64)st->cls_work[2] + ((i64)st->dp_cls[1] << 32);
            grand.v[7] = (i64)st->solve_cls[0] + ((i64)st->solve_cls[1] << 32);
            grand.v[8] = (i64)st->solve_cls[2];
        } else {
            for (int q = 0; q < kProbCols; ++q) { before.v[q] = 0; grand.v[q] = 0; }
            for (i64 bb = 0; bb < nb; ++bb) {
                if (bb == b) continue;
                ProbSizes acc = prob_block_sizes(cand_pn, cand_ln, bb, n, nullptr, sp), e2, t2;
                wg_scan_cols(acc, e2, t2, lds);
                for (int q = 0; q < kProbCols; ++q) { if (bb < b) before.v[q] += t2.v[q]; grand.v[q] += t2.v[q]; }
            }
        }
        ProbSizes acc = prob_block_sizes(cand_pn, cand_ln, b, n, sz, sp);
        if (!bs) prob_block_maxima(st, cand_pn, cand_ln, b, n, l_mx);   // (with block sums, k_prob_scan1 has done it; read after the barriers below)
        if (!bs) prob_block_wide(st, cand_pn, cand_ln, cand_wide, b, n, sp);
        wg_scan_cols(acc, ex, tot, lds);
        if (!bs && threadIdx.x == 0) prob_publish_maxima(st, l_mx);
        for (int q = 0; q < kProbCols; ++q) ex.v[q] += before.v[q];
        if (!bs) {
            for (int q = 0; q < kProbCols; ++q) grand.v[q] += tot.v[q];
            if (b == nb - 1 && threadIdx.x == 0) prob_store_totals(st, grand);
        }
        const i64 i0 = b * kProbBlock + (i64)threadIdx.x * 4;
        const i64 g_cls0 = col_lo(grand.v[4]), g_cls1 = col_hi(grand.v[4]), g_cls2 = col_lo(grand.v[5]);
        const i64 g_dp0 = col_hi(grand.v[0]), g_dp1 = col_hi(grand.v[5]);
        const i64 g_sol0 = col_lo(grand.v[7]), g_sol1 = col_hi(grand.v[7]), g_sol2 = col_lo(grand.v[8]);
        // what a problem's records need, for this thread's four candidates, in three rounds of loads instead of one
        // chain per candidate (a load under `if (problem)` is a branch with its own wait): candidate -> interval -> partition
        int pn4[4], iv4[4], ll4[4], ln4[4], is4[4], part4[4], lanes4[4];
        i64 co4[4];
        for (int e = 0; e < 4; ++e) {
            const i64 cc = i0 + e < n ? i0 + e : n - 1;
            pn4[e] = cand_pn[cc]; iv4[e] = cand_iv[cc]; ll4[e] = cand_ll[cc]; ln4[e] = cand_ln[cc];
        }
        for (int e = 0; e < 4; ++e) { co4[e] = cand_off[iv4[e]]; is4[e] = iv_start[iv4[e]]; part4[e] = iv_part[iv4[e]]; }
        for (int e = 0; e < 4; ++e) lanes4[e] = (int)(part_lane_off[part4[e] + 1] - part_lane_off[part4[e]]);
        // this workgroup's share of the two ends of every solve list (the barriers of wg_scan_cols above have published the zeroed counters)
        int le4[4];
        for (int e = 0; e < 4; ++e) {
            le4[e] = (sz[e].v[0] && i0 + e < n) ? list_end_of(pn4[e], ln4[e]) : -1;
            if (le4[e] >= 0) atomicAdd(&l_cnt2[le4[e]], 1);
        }
        __syncthreads();
        if (threadIdx.x < 8 && l_cnt2[threadIdx.x]) l_base2[threadIdx.x] = (int)atomicAdd(&st->list_cur[threadIdx.x], (unsigned)l_cnt2[threadIdx.x]);
        __syncthreads();
        const i64 g_tiny = bs ? (i64)st->n_tiny : col_hi(grand.v[8]);
        for (int e = 0; e < 4; ++e) {
            if (sz[e].v[0]) {
                i64 c = i0 + e;
                i64 slot = col_lo(ex.v[0]);
                int nn = pn4[e];
                if (slot < prob_cap) {
                    int k = iv4[e];
                    pr.iv[slot] = k; pr.start[slot] = (int)(c - co4[e]) - (nn - 1); pr.n[slot] = nn;
                    pr.pair_off[slot] = ex.v[1]; pr.tri_off[slot] = ex.v[2]; pr.cov_off[slot] = ex.v[3];
                    pr.flags[slot] = 0; pr.chain[slot] = 0;
                    pr.lane_lo[slot] = ll4[e]; pr.lane_n[slot] = ln4[e];
                    ProbDesc d;
                    d.c0 = c - (nn - 1); d.pair_off = ex.v[1]; d.tri_off = ex.v[2]; d.cov_off = ex.v[3];
                    d.n = nn; d.lane_lo = ll4[e]; d.lane_n = ln4[e]; d.g0 = is4[e];
                    d.outside = lanes4[e] - d.lane_n;
                    d.iv = k; d.w0 = (int)(col_lo(ex.v[4]) + col_hi(ex.v[4]) + col_lo(ex.v[5]) + col_lo(ex.v[6]));
                    const int kind = prob_kind(nn, ln4[e], sp);
                    const bool keeps_wide = kind == kKindFused && cand_wide[c] != 0;
                    d.kind = keeps_wide ? kKindFusedWide : kind;
                    desc[slot] = d;
                    if (le4[e] >= 0) {          // solve lists: class 0, then class 1, then class 2, then k_tiny's problems
                        const int li = le4[e] >> 1;
                        const i64 lbase = li == 0 ? 0 : (li == 1 ? g_sol0 : (li == 2 ? g_sol0 + g_sol1 : g_sol0 + g_sol1 + g_sol2));
                        const i64 llen = li == 0 ? g_sol0 : (li == 1 ? g_sol1 : (li == 2 ? g_sol2 : g_tiny));
                        const i64 off = (i64)l_base2[le4[e]] + atomicAdd(&l_cur2[le4[e]], 1);
                        const i64 si = (le4[e] & 1) ? lbase + llen - 1 - off : lbase + off;
                        // the kernels of the solve lists read the record from the list itself (one load less in every problem's
                        // chain of dependent loads); w0, the arena path's work item, is the problem's slot there
                        if (si >= lbase && si < lbase + llen && si < prob_cap) {
                            solve_items[si] = (int)slot; d.w0 = (int)slot; solve_desc[si] = d;
                            // the problems of a solve list that keep more than kFuseLanes reads, as list positions: what the
                            // 16-bit-counter instances are launched over (a class has a handful; as launches over the whole
                            // list their 90 KB workgroups waited for room behind everything else: config3, one such problem
                            // started 127 us into the stage)
                            if (li < 3 && keeps_wide) {
                                const i64 wp = lbase + (i64)atomicAdd(&st->wide_cur[li], 1u);
                                if (wp < lbase + llen && wp < prob_cap) wide_items[wp] = (int)(si - lbase);
                                // ... and of the three lists together, as positions from the first list's start: a batch with a
                                // handful of them has one launch for them all (plan 'W')
                                const i64 wa = (i64)atomicAdd(&st->wide_cur[3], 1u);
                                if (wa < prob_cap) wide_all[wa] = (int)si;
                            }
                        }
                    }
                    if (kind == kKindArena) {   // DP problem lists: the small problems first, then the big ones
                        // (then the huge ones); k_tiny's and k_solve's problems are in no DP list
                        i64 di = nn <= kDpSmall ? col_hi(ex.v[0]) : (nn <= kNMax ? g_dp0 + col_hi(ex.v[5]) : g_dp0 + g_dp1 + col_hi(ex.v[6]));
                        if (di < prob_cap) dp_items[di] = (int)slot;
                    }
                    int cls = size_class(nn);
                    const i64 e_cls0 = col_lo(ex.v[4]), e_cls1 = col_hi(ex.v[4]), e_cls2 = col_lo(ex.v[5]), e_cls3 = col_lo(ex.v[6]);
                    const i64 w0 = e_cls0 + e_cls1 + e_cls2 + e_cls3;   // work items before this problem
                    i64 cbase = cls == 0 ? e_cls0 : (cls == 1 ? g_cls0 + e_cls1 : (cls == 2 ? g_cls0 + g_cls1 + e_cls2
                                                                                    : g_cls0 + g_cls1 + g_cls2 + e_cls3));
                    const i64 cnt = col_lo(sz[e].v[4]) + col_hi(sz[e].v[4]) + col_lo(sz[e].v[5]) + col_lo(sz[e].v[6]);
                    if (w0 + cnt > work_cap || cbase + cnt > work_cap) atomicOr(&st->err, kErrOverflowWork);
                    else if (cnt <= kProbDirect) {
                        for (i64 q = 0; q < cnt; ++q) {
                            work_pc[w0 + q] = make_int2((int)slot, (int)q);
                            cls_items[cbase + q] = make_int4((int)(w0 + q), (int)slot, (int)q, 0);
                        }
                    } else {
                        int li = atomicAdd(&l_n, 1);
                        l_slot[li] = (int)slot; l_cnt[li] = (int)cnt; l_w0[li] = w0; l_c0[li] = cbase;
                    }
                } else atomicOr(&st->err, kErrOverflowProblems);
            }
            for (int q = 0; q < kProbCols; ++q) ex.v[q] += sz[e].v[q];
        }
        __syncthreads();
        const int ln = l_n;
        for (int li = threadIdx.x >> 6; li < ln; li += 4) {        // one wave per long list
            const int slot = l_slot[li], cnt = l_cnt[li];
            const i64 w0 = l_w0[li], c0 = l_c0[li];
            for (int q = lane_id(); q < cnt; q += 64) {
                work_pc[w0 + q] = make_int2(slot, q);
                cls_items[c0 + q] = make_int4((int)(w0 + q), slot, q, 0);
            }
        }
        __syncthreads();
    }
    emit_done(sw, sync_gen);
}

}  // namespace fseg
