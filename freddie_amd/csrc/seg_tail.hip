// seg_tail.hip -- S6 refinement (refine_segmentation :249-266) and S7 labels (:808-830): segments, refine, label columns / fill / reads, packing.
// Part of libfreddie_seg.so (gfx950); shared definitions: seg_common.h, declarations: seg_kernels.h, launches: freddie_seg.hip.
#include "seg_kernels.h"

namespace fseg {

__global__ void k_segments(i64 K, const i64 *pos_off, const i64 *cand_off, const int *cand_y, const int *__restrict__ y_raw,
                           const int *__restrict__ blk_pre, const int *tile_tot, const int *iv_tile0, const unsigned char *chosen, unsigned *final_flag, int *rseg_c, int *rseg_prev,
                           Status *st) {
    __shared__ int lds[16];
    __shared__ int cnt_s[16];
    __shared__ u64 base_s;
    const int T = blockDim.x;
    int lane = lane_id(), wave = threadIdx.x >> 6, nw = (T + 63) >> 6;
    // the inner-positions test of refine_segmentation (:258) for the segment (py, y] of interval k, whose first tile is tile0
    auto inner_sum_ok = [&](i64 base, int tile0, int py, int y) -> bool {
        // refine_segmentation's `sum(i_vals) < 20 -> continue` (:258), exactly, over the inner positions [py+20, y-21]:
        // k_smooth's prefix of the histogram at the start of a's and of b's block (inside their tiles), the tiles
        // between them, and the positions of those two blocks up to a (exclusive) / up to b (inclusive)
        const int a = py + 20, b = y - 21;
        const int ta = a >> kSmoothShift, tb = b >> kSmoothShift;
        if (tb - ta > 64) return true;                            // (very long segments: k_refine sums them itself)
        constexpr int kBlocks = kSmoothTile / kSumBlock;
        const int *tt = tile_tot + tile0;
        const int *yr = y_raw + base;
        const int a0 = a & ~(kSumBlock - 1), b0 = b & ~(kSumBlock - 1);
        i64 tot = (i64)blk_pre[(i64)(tile0 + tb) * kBlocks + ((b & (kSmoothTile - 1)) >> kSumShift)]
                - (i64)blk_pre[(i64)(tile0 + ta) * kBlocks + ((a & (kSmoothTile - 1)) >> kSumShift)];
        // (the two blocks as 16-byte loads from dword-aligned addresses; a block of b's may reach beyond the interval's
        // last position -- into the next interval's counts or the slab's padding: masked)
        int4u va[kSumBlock / 4], vb[kSumBlock / 4];
#pragma unroll
        for (int e = 0; e < kSumBlock / 4; ++e) { va[e] = *reinterpret_cast<const int4u *>(yr + a0 + 4 * e); vb[e] = *reinterpret_cast<const int4u *>(yr + b0 + 4 * e); }
#pragma unroll
        for (int e = 0; e < kSumBlock / 4; ++e) {
            const int pa = a0 + 4 * e, pb = b0 + 4 * e;
            tot += (pb <= b ? vb[e].x : 0) + (pb + 1 <= b ? vb[e].y : 0) + (pb + 2 <= b ? vb[e].z : 0) + (pb + 3 <= b ? vb[e].w : 0);
            tot -= (pa < a ? va[e].x : 0) + (pa + 1 < a ? va[e].y : 0) + (pa + 2 < a ? va[e].z : 0) + (pa + 3 < a ? va[e].w : 0);
        }
        for (int q = ta; q < tb; ++q) tot += tt[q];
        return tot >= 20;
    };
    for (i64 k = blockIdx.x; k < K; k += gridDim.x) {
        i64 c0 = cand_off[k];
        int N = (int)(cand_off[k + 1] - c0);
        i64 base = pos_off[k];
        if (T == 64 && N <= 64 * kSegChunks) {
            // an interval of at most 256 candidates, one wave (round 5): the chosen flags and the candidates' positions of all its
            // 64-candidate chunks are asked for together, the previous chosen candidate's position comes from its lane (or the
            // chunk before), and the chunks' inner-sum tests are in flight together -- three rounds of loads per interval whatever
            // its length and no barrier.  (The kernel is a chain of dependent loads per interval; with six rounds per 64
            // candidates the few long intervals of a batch -- 100 to 300 candidates -- were what it took: 26 us.)
            const int tile0 = iv_tile0[k];
            unsigned char ch[kSegChunks];
            int yv[kSegChunks];
#pragma unroll
            for (int u = 0; u < kSegChunks; ++u) {
                const int c = u * 64 + lane;
                ch[u] = chosen[c0 + (c < N ? c : 0)];
                yv[u] = cand_y[c0 + (c < N ? c : 0)];
            }
            int pyv[kSegChunks];
            bool need[kSegChunks];
            int carry_y = -1;                                          // position of the last chosen candidate of the chunks before (wave-uniform)
#pragma unroll
            for (int u = 0; u < kSegChunks; ++u) {
                const int c = u * 64 + lane;
                const bool f = c < N && ch[u];
                const u64 mask = __ballot(f);
                const u64 below = mask & ((1ULL << lane) - 1ULL);
                const int prev = below ? 63 - __clzll((long long)below) : -1;
                const int py_in = __shfl(yv[u], prev >= 0 ? prev : 0);
                const int py = prev >= 0 ? py_in : carry_y;
                if (f) set_flag(final_flag, base + yv[u]);
                pyv[u] = py;
                need[u] = f && py >= 0 && yv[u] - py > 40;              // :252
                if (mask) carry_y = __shfl(yv[u], 63 - __clzll((long long)mask));
            }
#pragma unroll
            for (int u = 0; u < kSegChunks; ++u) if (need[u]) need[u] = inner_sum_ok(base, tile0, pyv[u], yv[u]);
#pragma unroll
            for (int u = 0; u < kSegChunks; ++u) {
                const u64 m = __ballot(need[u]);
                if (m) {                                               // (wave-uniform)
                    u64 slot0 = 0;
                    if (lane == 0) slot0 = atomicAdd(&st->n_rseg, (u64)__popcll(m));
                    slot0 = __shfl(slot0, 0);
                    if (need[u]) {
                        const u64 slot = slot0 + __popcll(m & ((1ULL << lane) - 1ULL));
                        rseg_c[slot] = (int)(c0 + u * 64 + lane); rseg_prev[slot] = pyv[u];
                    }
                }
            }
            continue;
        }
        int carry = -1;
        for (int t0 = 0; t0 < N; t0 += T) {
            int c = t0 + threadIdx.x;
            bool f = c < N && chosen[c0 + c];
            int prev = wg_prev_flagged(f, c, carry, lds);
            int y = 0, py = -1;
            if (f) {
                y = cand_y[c0 + c];
                set_flag(final_flag, base + y);
                if (prev >= 0) py = cand_y[c0 + prev];
            }
            bool need = f && py >= 0 && y - py > 40;                  // :252
            if (need) need = inner_sum_ok(base, iv_tile0[k], py, y);
            u64 m = __ballot(need);
            if (lane == 0) cnt_s[wave] = __popcll(m);
            __syncthreads();
            if (threadIdx.x == 0) {
                int tot = 0;
                for (int w = 0; w < nw; ++w) { int v = cnt_s[w]; cnt_s[w] = tot; tot += v; }
                base_s = tot ? atomicAdd(&st->n_rseg, (u64)tot) : 0;
            }
            __syncthreads();
            if (need) {
                u64 slot = base_s + cnt_s[wave] + __popcll(m & ((1ULL << lane) - 1ULL));
                rseg_c[slot] = (int)(c0 + c); rseg_prev[slot] = py;
            }
            __syncthreads();
        }
    }
}

// (Round 5, measured and not kept: the same by candidates instead of intervals -- a thread per candidate of the batch, its predecessor
// from speculative loads of the four candidates below it, one slot atomic per workgroup: 2 300 waves instead of 61 000 and 20-21 us
// either way.  With the pieces taken out: 11 us without the final flags and the inner sums, the sums 8, the flags 2 -- the kernel is four
// or five levels of dependent L2 round trips and a drain whatever its shape.)
__global__ void __launch_bounds__(64) k_refine(const Status *st, const int *cand_iv, const int *rseg_c,
                                               const int *rseg_prev, const int *cand_y, const i64 *pos_off,
                                               const int *y_raw, const double *w_g, int radius, double sigma,
                                               double *g_scr, int *pk_scr, unsigned char *flag_scr,
                                               unsigned char *keep_scr, unsigned *final_flag) {
    __shared__ double ws[kMaxRadius + 1];
    __shared__ int xl[kRefCap], pkl[kRefCap];
    __shared__ double gl[kRefCap];
    __shared__ unsigned char pfl[kRefCap], kpl[kRefCap];
    const int skip = 20;
    int lane = lane_id();
    for (int j = lane; j <= radius; j += 64) ws[j] = w_g[j];
    __syncthreads();
    i64 n_seg = (i64)st->n_rseg;
    for (i64 si = blockIdx.x; si < n_seg; si += gridDim.x) {
      {
        i64 sg = rseg_c[si];
        int s = rseg_prev[si];
        int e = cand_y[sg];
        int len = e - s;
        i64 base = pos_off[cand_iv[sg]] + s;
        const int *xr = y_raw + base;
        if (len <= kRefCap) {
            // ---- the segment fits in LDS (nearly all do): its counts are fetched once, four coalesced rows at a time
            // from clamped addresses (a load under a condition is a branch with its own wait -- and the filter below
            // would do 2 * radius of them per position), then everything runs out of LDS --------------------------------
            i64 tot_l = 0;
            for (int i0 = 0; i0 < len; i0 += 256) {
                int v[4];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) { const int t = i0 + e4 * 64 + lane; v[e4] = xr[t < len ? t : len - 1]; }
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int t = i0 + e4 * 64 + lane;
                    if (t < len) { const int m = (t < skip || t >= len - skip) ? 0 : v[e4]; xl[t] = m; tot_l += m; }   // zeroed ends (:256-257)
                }
            }
            for (int d = 32; d >= 1; d >>= 1) tot_l += __shfl_xor(tot_l, d);
            if (tot_l < 20) continue;                                     // sum(i_vals) < 20 -> skip (:258)
            __syncthreads();
            for (int i = lane; i < len; i += 64) {                         // gaussian_filter1d(..., mode='constant', truncate=1.0) (:260-261)
                double acc = __dmul_rn((double)xl[i], ws[0]);
                for (int j = radius; j >= 1; --j) {
                    const int a = i - j, b = i + j;
                    const int sv = (a >= 0 ? xl[a] : 0) + (b < len ? xl[b] : 0);
                    acc = __dadd_rn(acc, __dmul_rn((double)sv, ws[j]));
                }
                gl[i] = acc;
                pfl[i] = 0;
            }
            __syncthreads();
            for (int i = 1 + lane; i < len - 1; i += 64) {                 // scipy _local_maxima_1d
                const double gi = gl[i];
                if (gl[i - 1] < gi) {
                    int ia = i + 1;
                    while (ia < len - 1 && gl[ia] == gi) ++ia;
                    if (gl[ia] < gi) pfl[(i + ia - 1) / 2] = 1;
                }
            }
            __syncthreads();
            int m = 0;
            for (int t0 = 0; t0 < len; t0 += 64) {
                const int i = t0 + lane;
                const bool f = i < len && pfl[i];
                const u64 mask = __ballot(f);
                if (f) { const int rank = __popcll(mask & ((1ULL << lane) - 1ULL)); pkl[m + rank] = i; kpl[m + rank] = 1; }
                m += __popcll(mask);
            }
            __syncthreads();
            for (;;) {                                                     // find_peaks(distance=20), as in the general path below
                double bv = -INFINITY; int bq = -1;
                for (int q = lane; q < m; q += 64)
                    if (kpl[q] == 1) { const double v = gl[pkl[q]]; if (v > bv || (v == bv && q > bq)) { bv = v; bq = q; } }
                for (int d = 32; d >= 1; d >>= 1) {
                    const double ov = __shfl_xor(bv, d); const int oq = __shfl_xor(bq, d);
                    if (oq >= 0 && (bq < 0 || ov > bv || (ov == bv && oq > bq))) { bv = ov; bq = oq; }
                }
                if (bq < 0) break;
                if (lane == 0) {
                    kpl[bq] = 2;
                    const int pj = pkl[bq];
                    for (int q = bq - 1; q >= 0 && pj - pkl[q] < skip; --q) kpl[q] = 0;
                    for (int q = bq + 1; q < m && pkl[q] - pj < skip; ++q) kpl[q] = 0;
                }
                __syncthreads();
            }
            for (int q = lane; q < m; q += 64) {
                if (kpl[q] != 2) continue;
                const int i = pkl[q];
                i64 a = (i64)rint((double)i - sigma), b = (i64)rint((double)i + sigma + 1.0);   // Python round(): half even
                if (a < 0) { a += len; if (a < 0) a = 0; } else if (a > len) a = len;           // slice semantics (:263)
                if (b < 0) { b += len; if (b < 0) b = 0; } else if (b > len) b = len;
                double sm = 0.0;
                for (i64 x = a; x < b; ++x) sm = __dadd_rn(sm, gl[x]);
                if (!(sm < 20.0)) set_flag(final_flag, base + i);
            }
            __syncthreads();
            continue;
        }
        // ---- general path (segments longer than kRefCap): scratch in global memory -------------------------------
        // sum(i_vals) < 20 -> skip (:258); values are exact integers
        i64 tot = 0;
        for (int i = skip + lane; i < len - skip; i += 64) tot += xr[i];
        for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
        if (tot < 20) continue;
        double *g = g_scr + base;
        int *pk = pk_scr + base;
        unsigned char *pf = flag_scr + base, *kp = keep_scr + base;
        // gaussian_filter1d(i_vals, sigma, mode='constant', cval=0, truncate=1.0)  (:260-261)
        for (int i = lane; i < len; i += 64) {
#define FSEG_V(t) (((t) < skip || (t) >= len - skip) ? 0 : xr[t])
            double acc = __dmul_rn((double)FSEG_V(i), ws[0]);
            for (int j = radius; j >= 1; --j) {
                int a = i - j, b = i + j;
                int sv = (a >= 0 ? FSEG_V(a) : 0) + (b < len ? FSEG_V(b) : 0);
                acc = __dadd_rn(acc, __dmul_rn((double)sv, ws[j]));
            }
#undef FSEG_V
            g[i] = acc;
            pf[i] = 0;
        }
        __syncthreads();
        for (int i = 1 + lane; i < len - 1; i += 64) {
            double gi = g[i];
            if (g[i - 1] < gi) {
                int ia = i + 1;
                while (ia < len - 1 && g[ia] == gi) ++ia;
                if (g[ia] < gi) pf[(i + ia - 1) / 2] = 1;
            }
        }
        __syncthreads();
        int m = 0;
        for (int t0 = 0; t0 < len; t0 += 64) {
            int i = t0 + lane;
            bool f = i < len && pf[i];
            u64 mask = __ballot(f);
            if (f) { int rank = __popcll(mask & ((1ULL << lane) - 1ULL)); pk[m + rank] = i; kp[m + rank] = 1; }
            m += __popcll(mask);
        }
        __syncthreads();
        // find_peaks(distance=20): highest peak first, ties -> later peak first; state 1 = kept and
        // unprocessed, 2 = kept and processed, 0 = removed
        for (;;) {
            double bv = -INFINITY; int bq = -1;
            for (int q = lane; q < m; q += 64)
                if (kp[q] == 1) { double v = g[pk[q]]; if (v > bv || (v == bv && q > bq)) { bv = v; bq = q; } }
            for (int d = 32; d >= 1; d >>= 1) {
                double ov = __shfl_xor(bv, d); int oq = __shfl_xor(bq, d);
                if (oq >= 0 && (bq < 0 || ov > bv || (ov == bv && oq > bq))) { bv = ov; bq = oq; }
            }
            if (bq < 0) break;
            if (lane == 0) {
                kp[bq] = 2;
                int pj = pk[bq];
                for (int q = bq - 1; q >= 0 && pj - pk[q] < skip; --q) kp[q] = 0;
                for (int q = bq + 1; q < m && pk[q] - pj < skip; ++q) kp[q] = 0;
            }
            __syncthreads();
        }
        for (int q = lane; q < m; q += 64) {
            if (kp[q] != 2) continue;
            int i = pk[q];
            i64 a = (i64)rint((double)i - sigma), b = (i64)rint((double)i + sigma + 1.0);   // Python round(): half even
            if (a < 0) { a += len; if (a < 0) a = 0; } else if (a > len) a = len;           // slice semantics (:263)
            if (b < 0) { b += len; if (b < 0) b = 0; } else if (b > len) b = len;
            double sm = 0.0;
            for (i64 x = a; x < b; ++x) sm = __dadd_rn(sm, g[x]);
            if (!(sm < 20.0)) set_flag(final_flag, base + i);
        }
        __syncthreads();
      }
    }
}


// ---------------------------------------------------------------------------------------------
// S7  labels   (py/freddie_segment.py:808-830, sentinel :829-830, pop :840)
// The label matrix of a partition is R x (F-1) bytes ('0','1','2').  A read overlaps only a few of
// the F-1 segments, so the matrix is first filled with each column's zero-coverage label (k_label_fill,
// a pure streaming store) and then every read rewrites just the columns its exons can reach
// (k_label_reads).
// ---------------------------------------------------------------------------------------------
// label arena offsets of the partitions (one workgroup of 256 threads; part of k_label_cols)
__device__ void label_plan(int n_part, const i64 *part_iv_off, const i64 *part_rep_off,
                           const i64 *final_off, i64 *label_off, Status *st, i64 label_cap) {
    // (a workgroup scan per 256 partitions; until round 5 thread 0 added the 256 sizes up one by one -- a chain of 256 LDS round
    // trips, 12 us per 256 partitions: it was what k_label_cols took, 23 us for the 500 partitions of a config4 batch)
    __shared__ i64 carry_s;
    __shared__ i64 scan_lds[16];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int p0 = 0; p0 < n_part; p0 += blockDim.x) {
        int p = p0 + threadIdx.x;
        i64 bytes = 0;
        if (p < n_part) {
            i64 F = final_off[part_iv_off[p + 1]] - final_off[part_iv_off[p]];
            bytes = (part_rep_off[p + 1] - part_rep_off[p]) * (F > 0 ? F - 1 : 0);
        }
        i64 tot;
        const i64 ex = wg_exclusive_scan64(bytes, scan_lds, &tot);
        const i64 carry = carry_s;
        if (p < n_part) label_off[p] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        label_off[n_part] = carry_s;
        st->label_bytes = (u64)carry_s;
        if (carry_s > label_cap) atomicOr(&st->err, kErrOverflowLabels);
    }
}

// per final index f (= column): integer thresholds of the segment [final_f, final_f+1) and the label of
// a read without coverage there; the last index of an interval is the sentinel column (hi = INT_MAX)
__global__ void __launch_bounds__(256) k_label_cols(i64 K, const i64 *final_off, const int *final_y, const int *final_iv, const int *iv_part,
                                                    const double *h_table, int h_len, double tau, const int2 *thr_tab, int2 *col_thr,
                                                    unsigned char *col_zero, int *part_has2, int n_part,
                                                    const i64 *part_iv_off, const i64 *part_rep_off, i64 *label_off,
                                                    Status *st, i64 label_cap) {
    if (blockIdx.x == 0) label_plan(n_part, part_iv_off, part_rep_off, final_off, label_off, st, label_cap);
    i64 F = final_off[K];
    for (i64 f = (i64)blockIdx.x * blockDim.x + threadIdx.x; f < F; f += (i64)gridDim.x * blockDim.x) {
        const i64 k = final_iv[f];                           // the interval of every final position, noted by the compaction that emitted it
        if (f + 1 == final_off[k + 1]) { col_thr[f] = make_int2(0x7fffffff, 0x7fffffff); col_zero[f] = '0'; continue; }
        int hi, lo;
        label_thresholds_tab((i64)final_y[f + 1] - final_y[f] + 1, thr_tab, h_table, h_len, tau, &hi, &lo);
        col_thr[f] = make_int2(hi, lo);
        col_zero[f] = lo >= 0 ? '0' : '2';
        if (lo < 0) atomicOr(&part_has2[iv_part[k]], 1);
    }
}

// The label arena is pre-filled with '0' (the label of a read without coverage) by one streaming kernel; in
// partitions in which a zero-coverage read is ambiguous for some segment (lo < 0: h >= 1 -- threshold_rate = 1, or a table entry rounded to 1.0) every rep
// first rewrites its row with the columns' defaults (k_label_reads).
__global__ void __launch_bounds__(256) k_label_zero(uint4 *labels16, i64 n16) {
    fill_labels(labels16, n16, (i64)blockIdx.x * blockDim.x + threadIdx.x, (i64)gridDim.x * blockDim.x);
}

__global__ void __launch_bounds__(256) k_label_reads(int n_blocks, const int *rb_part, const int *rb_r0,
                                                     const i64 *label_off, i64 label_cap, int n_part,
                                                     const i64 *part_iv_off, const i64 *part_rep_off,
                                                     const i64 *final_off, const int *final_pos, const int2 *col_thr,
                                                     const i64 *rep_exon_off, const int *ex_ts, const int *ex_te,
                                                     const unsigned char *col_zero, const int *part_has2,
                                                     unsigned char *labels, unsigned *packed) {
    // packed != nullptr: the labels are written at two bits each into a cleared arena (label g at bits 2 (g & 15) .. of word g >> 4,
    // the layout fseg_results_packed delivers): a thread walks consecutive columns of its rep's row, collects the codes of a word
    // in a register and ORs the word in when it moves on.  Only where no column's default is '2' (the host: no table entry and no rate >= 1, label_has2).
    __shared__ int fp_s[kLabelCols + 1];
    __shared__ int2 th_s[kLabelCols];
    if (label_off[n_part] > label_cap) return;
    for (i64 unit = blockIdx.x; unit < (i64)n_blocks * kLabelSplit; unit += gridDim.x) {
        const int blk = (int)(unit / kLabelSplit), sub = (int)(unit % kLabelSplit);
        int p = rb_part[blk];
        i64 f0 = final_off[part_iv_off[p]];
        i64 F = final_off[part_iv_off[p + 1]] - f0;
        i64 S = F - 1;
        if (S <= 0) continue;
        if ((i64)rb_r0[blk] + sub * (256 / kLabelSplit) >= part_rep_off[p + 1]) continue;
        const int *fp = final_pos + f0;                      // ascending over the whole partition
        const int2 *th = col_thr + f0;
        __syncthreads();
        // short column tables are staged in LDS; a long table stays in global memory (a read only visits the few
        // columns around its exons, and staging the whole table per workgroup would cost more than it saves)
        if (S <= kLabelStage) {
            for (int x = threadIdx.x; x <= S; x += blockDim.x) fp_s[x] = fp[x];
            for (int x = threadIdx.x; x < S; x += blockDim.x) th_s[x] = th[x];
            fp = fp_s; th = th_s;
        }
        __syncthreads();
        i64 r = (i64)rb_r0[blk] + sub * (256 / kLabelSplit) + (threadIdx.x / kLabelSplit);
        const int q = threadIdx.x % kLabelSplit;
#ifdef FSEG_LAB_NOSEARCH
        if (fp[0] != -12345) continue;                       // diagnostic build (wrong results): the workgroup's staging only
#endif
        if (!packed && part_has2[p]) {                       // uniform over the workgroup: the rows' defaults are not all '0'
            if (r < part_rep_off[p + 1]) {
                unsigned char *row0 = labels + label_off[p] + (r - part_rep_off[p]) * S;
                const unsigned char *cz = col_zero + f0;
                for (i64 x = S * q / kLabelSplit; x < S * (q + 1) / kLabelSplit; ++x) row0[x] = cz[x];
            }
            __threadfence_block();
            __syncthreads();                                 // the label stores below may hit bytes another thread just wrote
        }
        if (r >= part_rep_off[p + 1]) continue;
        const i64 row_g0 = label_off[p] + (r - part_rep_off[p]) * S;
        unsigned char *row = labels + row_g0;
        unsigned acc = 0;
        i64 acc_w = -1;
        i64 e = rep_exon_off[r], e1 = rep_exon_off[r + 1];
        if (e >= e1) continue;
        int first_ts = ex_ts[e], last_te = ex_te[e1 - 1];
        // first column whose segment [fp[c], fp[c+1]) ends after first_ts
        int lo = 0, hi = (int)S;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (fp[mid + 1] <= first_ts) lo = mid + 1; else hi = mid; }
        // first column that starts after last_te
        int c_hi = lo; hi = (int)S;
        while (c_hi < hi) { int mid = (c_hi + hi) >> 1; if (fp[mid] <= last_te) c_hi = mid + 1; else hi = mid; }
        // this thread's share of [lo, c_hi)
        const int span = c_hi - lo;
        const int c_a = lo + (int)((i64)span * q / kLabelSplit), c_b = lo + (int)((i64)span * (q + 1) / kLabelSplit);
        if (c_a >= c_b) continue;
#ifdef FSEG_LAB_NOLOOP
        if (c_a != -12345) { if (c_b == 0x7ffffff0) atomicOr(&packed[0], 1u); continue; }    // diagnostic build (wrong results): no column loop
#endif
        if (q) {                                             // first exon that reaches the first column of the share
            const int g = fp[c_a];
            i64 a = e, b = e1;
            while (a < b) { i64 mid = (a + b) >> 1; if (ex_te[mid] < g) a = mid + 1; else b = mid; }
            e = a;
        }
        int ts = 0, te = 0;
        if (e < e1) { ts = ex_ts[e]; te = ex_te[e]; }
        for (int c = c_a; c < c_b; ++c) {
            int2 t2 = th[c];
            if (t2.x == 0x7fffffff) continue;                 // sentinel column between two intervals
            int g0 = fp[c], g1 = fp[c + 1];
            while (e < e1 && te < g0) { ++e; if (e < e1) { ts = ex_ts[e]; te = ex_te[e]; } }   // exons before the segment
            int cov = 0;
            if (e < e1 && ts < g1) {
                int a = ts > g0 ? ts : g0, b2 = te + 1 < g1 ? te + 1 : g1;
                if (b2 > a) cov += b2 - a;
                for (i64 x = e + 1; x < e1 && ex_ts[x] < g1; ++x) {
                    int a3 = ex_ts[x] > g0 ? ex_ts[x] : g0;
                    int b3 = ex_te[x] + 1 < g1 ? ex_te[x] + 1 : g1;
                    if (b3 > a3) cov += b3 - a3;
                }
            }
            // the arena already holds the zero-coverage label of the column ('0', or '2' when lo < 0): store only what differs
            const unsigned char lab = cov >= t2.x ? '1' : (cov <= t2.y ? '0' : '2');
            if (lab != (t2.y < 0 ? '2' : '0')) {
                if (packed) {
                    const i64 g = row_g0 + c, w = g >> 4;
#ifdef FSEG_LAB_NOSTORE
                    if (w != acc_w) { acc_w = w; }           // diagnostic build (wrong results): one store per thread
#else
                    if (w != acc_w) { if (acc) atomicOr(&packed[acc_w], acc); acc_w = w; acc = 0; }
#endif
                    acc |= (unsigned)(lab - '0') << (2 * (int)(g & 15));
                } else row[c] = lab;
            }
        }
        if (acc) atomicOr(&packed[acc_w], acc);
    }
}


// (Round 5, measured and not kept: the (rep, column) pairs of a workgroup's 64 reps as ONE list walked by its 256 threads -- every
// thread the same number of pairs, the reps' exons in LDS, a word's codes OR-ed across lanes before one atomic -- parity-green,
// 78 us against 73: finding a pair's rep, walking its exons from the first and the cross-lane OR cost more per pair than the
// balance saves.  Of k_label_reads' 73 us 41 are the column loop, 13 its stores, 9 the bisections, 10 the workgroup's staging
// (tools/probes/label_ablate.sh with the FSEG_LAB_NO* builds).)


// ---------------------------------------------------------------------------------------------
// Results to the host.  The label matrix is by far the largest thing that crosses PCIe (about 300 bytes per read, 75 MB per
// 250 k-read batch, 1.4 ms at 55 GB/s -- more than the whole device pipeline), and a label has three values: the arena is
// packed to two bits per label before it leaves (k_pack_labels; label byte g of the arena = bits 2(g & 3) .. of packed byte
// g >> 2) and the host writer unpacks rows straight into the TSV it is assembling (fhost_write_packed).
// (A copy kernel of our own that streams to pinned memory with a small grid was tried instead of the runtime's copy: it
// slows kernels of the other contexts of the pipeline 3x while it runs.  The runtime's own large copies are kernels too --
// see sdma_d2h() for what replaces them.)
// ---------------------------------------------------------------------------------------------
// ... and back: 32 bits -> 16 ASCII labels (fseg_results / fseg_download of a run that wrote its labels packed)
__global__ void __launch_bounds__(256) k_unpack_labels(const unsigned *__restrict__ packed, uint4 *__restrict__ labels16, i64 n16) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (i64)gridDim.x * blockDim.x) {
        const unsigned v = packed[i];
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned b = (v >> (8 * q)) & 0xffu;      // four labels
            w[q] = 0x30303030u | (b & 3u) | ((b & 0xcu) << 6) | ((b & 0x30u) << 12) | ((b & 0xc0u) << 18);
        }
        labels16[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
__global__ void __launch_bounds__(256) k_pack_labels(const uint4 *__restrict__ labels16, unsigned *__restrict__ packed, i64 n16) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (i64)gridDim.x * blockDim.x) {
        const uint4 v = labels16[i];                    // 16 ASCII labels ('0' + 0 .. 2) -> 32 bits
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        unsigned out = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned x = w[q] & 0x03030303u;      // the two low bits of each byte are the label
            out |= ((x & 3u) | ((x >> 6) & 0xcu) | ((x >> 12) & 0x30u) | ((x >> 18) & 0xc0u)) << (8 * q);
        }
        packed[i] = out;
    }
}

}  // namespace fseg
