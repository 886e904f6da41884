// seg_upload.hip -- the upload's device-side preparation: per-rep validation and sort keys, the lane list and the lane-ordered exon stream,
// the histogram chunks' lane ranges.
// Part of libfreddie_seg.so (gfx950); shared definitions: seg_common.h, declarations: seg_kernels.h, launches: freddie_seg.hip.
#include "seg_kernels.h"

namespace fseg {


__global__ void __launch_bounds__(256) k_prep_reps(int n_blocks, const int *rb_part, const int *rb_r0, const i64 *part_rep_off,
                                                   const i64 *part_iv_off, const int *iv_start, const int *iv_end,
                                                   const i64 *rep_exon_off, const int *ex_ts, const int *ex_te, u64 *key,
                                                   int *val, int *rep_last, PrepStatus *ps) {
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int p = rb_part[blk];
        const i64 r = (i64)rb_r0[blk] + threadIdx.x;
        if (r >= part_rep_off[p + 1]) continue;
        const i64 e0 = rep_exon_off[r], e1 = rep_exon_off[r + 1];
        unsigned bad = 0;
        int first = 0, last = 0;
        if (e1 <= e0) bad = kPrepNoExons;
        else {
            const i64 k0 = part_iv_off[p], k1 = part_iv_off[p + 1];
            first = ex_ts[e0]; last = ex_te[e1 - 1];
            i64 kk = k0;
            {   // first interval that ends at or after the read's first position; exons and intervals are both ordered
                i64 lo = k0, hi = k1;
                while (lo < hi) { const i64 mid = (lo + hi) >> 1; if (iv_end[mid] < first) lo = mid + 1; else hi = mid; }
                kk = lo;
            }
            int prev_te = 0;
            for (i64 e = e0; e < e1; ++e) {
                const int ts = ex_ts[e], te = ex_te[e];
                if (!(ts < te)) bad |= kPrepExonEnds;                                   // :160
                if (e > e0 && !(prev_te <= ts)) bad |= kPrepExonOrder;                  // :158
                while (kk < k1 && iv_end[kk] < ts) ++kk;
                if (kk >= k1 || ts < iv_start[kk] || te > iv_end[kk]) bad |= kPrepExonInterval;   // :666-668
                prev_te = te;
            }
        }
        key[r] = ((u64)(unsigned)p << 32) | (u64)((unsigned)first ^ 0x80000000u);   // signed order of the position
        val[r] = (int)r;
        rep_last[r] = last;
        if (bad) {
            atomicOr(&ps->err, bad);
            for (int q = 0; q < 4; ++q) if ((bad >> q) & 1u) atomicMin((unsigned long long *)&ps->bad_rep[q], (unsigned long long)r);
        }
    }
}

__global__ void __launch_bounds__(256) k_lanes(int n_part, const i64 *part_rep_off, const i64 *part_lane_off, const u64 *key_sorted,
                                               const int *val_sorted, const int *rep_weight, const int *rep_last,
                                               const i64 *rep_exon_off, longlong2 *lane_ex, int *lane_start, int *lane_pmax,
                                               int sort_here, const u64 *key_unsorted, const int *ex_ts, const int *ex_te,
                                               int2 *lane_lx, int2 *lex) {
    __shared__ int lds[16];
    __shared__ int wmax[4];
    __shared__ int carry_max_s;
    __shared__ i64 carry_lane_s;
    __shared__ i64 carry_ex_s;
    __shared__ unsigned char owner_s[kLexChunk];
    __shared__ int eb_s[256];
    __shared__ i64 src_s[256];
    __shared__ u64 skey[kLaneSortMax];             // (biased first position << 32 | rep index inside the partition): unique, so
                                                   // the order is the stable order by position
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int p = blockIdx.x; p < n_part; p += gridDim.x) {
        const i64 r0 = part_rep_off[p], r1 = part_rep_off[p + 1];
        __syncthreads();
        if (threadIdx.x == 0) { carry_max_s = -0x7fffffff - 1; carry_lane_s = part_lane_off[p]; carry_ex_s = rep_exon_off[r0]; }
        if (sort_here) {
            const int nr = (int)(r1 - r0);
            int N = 1;
            while (N < nr) N <<= 1;
            for (int i = threadIdx.x; i < N; i += 256)
                skey[i] = i < nr ? ((key_unsorted[r0 + i] & 0xffffffffULL) << 32) | (u64)(unsigned)i : ~0ULL;     // padding sorts last
            __syncthreads();
            for (int k = 2; k <= N; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = threadIdx.x; t < N / 2; t += 256) {
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;      // the pair (i, i ^ j) with bit j clear in i
                        const u64 a = skey[i], b = skey[ixj];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { skey[i] = b; skey[ixj] = a; }
                    }
                    __syncthreads();
                }
        }
        __syncthreads();
        for (i64 t0 = r0; t0 < r1; t0 += 256) {
            const i64 i = t0 + threadIdx.x;
            const bool in = i < r1;
            int r = 0, first = 0;
            if (in) {
                if (sort_here) { const u64 k2 = skey[i - r0]; r = (int)(r0 + (i64)(k2 & 0xffffffffULL)); first = (int)((unsigned)(k2 >> 32) ^ 0x80000000u); }
                else { r = val_sorted[i]; first = (int)((unsigned)(key_sorted[i] & 0xffffffffULL) ^ 0x80000000u); }
            }
            const int w = in ? rep_weight[r] : 0;
            int m = in ? rep_last[r] : -0x7fffffff - 1;
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(m, d); if (lane >= d) m = max(m, y); }
            int tot;
            const int ex = wg_exclusive_scan(w, lds, &tot);
            if (lane == 63) wmax[wave] = m;
            __syncthreads();
            int run = carry_max_s;
            for (int w2 = 0; w2 < wave; ++w2) run = max(run, wmax[w2]);
            m = max(m, run);
            const i64 base = carry_lane_s + ex;
            // the partition's exons again, in lane order (the exon stream `lex`): a rep's exons start where the exons of
            // the reps sorted before it end, inside the partition's own exon range
            const longlong2 exr = in ? make_longlong2(rep_exon_off[r], rep_exon_off[r + 1]) : make_longlong2(0, 0);
            const int ne = (int)(exr.y - exr.x);
            int tot_e;
            const int ex_e = wg_exclusive_scan(ne, lds, &tot_e);
            const i64 ebase = carry_ex_s + ex_e;
            __syncthreads();
            if (threadIdx.x == 255) { carry_max_s = m; carry_lane_s = base + w; carry_ex_s = ebase + ne; }
            if (in) {
                const int2 lx = make_int2((int)ebase, (int)(ebase + ne));
                for (int q = 0; q < w; ++q) { lane_ex[base + q] = exr; lane_start[base + q] = first; lane_pmax[base + q] = m; lane_lx[base + q] = lx; }
            }
            lex_copy_tile(ex_e, ne, exr.x, tot_e, ebase - ex_e, ex_ts, ex_te, lex, owner_s, eb_s, src_s);
        }
    }
}


// The same for a batch that went through the batch-wide sort (it holds a partition of more than kLaneSortMax reps, e.g. one
// 50 000-read partition): a workgroup per partition would walk such a partition 256 reps at a time, alone (330 us for 50 k
// reps).  Instead every block of 256 sorted reps is a workgroup of its own, in three launches: block totals (weights, last
// positions), an exclusive scan of the totals inside each partition (one wave per partition), and the lanes themselves.
__global__ void __launch_bounds__(256) k_lane_blocks(int n_blocks, const int *rb_part, const int *rb_r0, const i64 *part_rep_off,
                                                     const int *val_sorted, const int *rep_weight, const int *rep_last,
                                                     i64 *rb_sum, int *rb_max, const i64 *rep_exon_off, i64 *rb_esum) {
    __shared__ int lds[16];
    __shared__ int wmax[4];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const i64 i = (i64)rb_r0[blk] + threadIdx.x;
        const bool in = i < part_rep_off[rb_part[blk] + 1];
        const int r = in ? val_sorted[i] : 0;
        const int w = in ? rep_weight[r] : 0;
        int m = in ? rep_last[r] : -0x7fffffff - 1;
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
        __syncthreads();
        int tot;
        (void)wg_exclusive_scan(w, lds, &tot);
        int tot_e;
        (void)wg_exclusive_scan(in ? (int)(rep_exon_off[r + 1] - rep_exon_off[r]) : 0, lds, &tot_e);
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        if (threadIdx.x == 0) { rb_sum[blk] = tot; rb_esum[blk] = tot_e; rb_max[blk] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])); }
    }
}

__global__ void __launch_bounds__(256) k_lane_block_scan(int n_part, int n_blocks, const int *rb_part, const i64 *part_lane_off,
                                                         const i64 *rb_sum, const int *rb_max, i64 *rb_base, int *rb_cmax,
                                                         const i64 *part_rep_off, const i64 *rep_exon_off, const i64 *rb_esum, i64 *rb_ebase) {
    const int lane = lane_id();
    for (int p = blockIdx.x * 4 + (int)(threadIdx.x >> 6); p < n_part; p += gridDim.x * 4) {
        // the partition's blocks are consecutive in the block list: [first block of p, first block of p + 1)
        int b0 = 0, b1 = n_blocks;
        { int lo = 0, hi = n_blocks; while (lo < hi) { const int mid = (lo + hi) >> 1; if (rb_part[mid] < p) lo = mid + 1; else hi = mid; } b0 = lo; }
        { int lo = b0, hi = n_blocks; while (lo < hi) { const int mid = (lo + hi) >> 1; if (rb_part[mid] <= p) lo = mid + 1; else hi = mid; } b1 = lo; }
        i64 carry = part_lane_off[p], carry_e = rep_exon_off[part_rep_off[p]];
        int carry_max = -0x7fffffff - 1;
        for (int c0 = b0; c0 < b1; c0 += 64) {
            const int b = c0 + lane;
            const bool in = b < b1;
            const i64 v = in ? rb_sum[b] : 0;
            int m = in ? rb_max[b] : -0x7fffffff - 1;
            i64 tot, tot_e;
            const i64 ex = wave_excl_scan(v, &tot);
            const i64 ex_e = wave_excl_scan(in ? rb_esum[b] : 0, &tot_e);
            if (in) rb_ebase[b] = carry_e + ex_e;
            carry_e += tot_e;
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(m, d); if (lane >= d) m = max(m, y); }      // inclusive running maximum
            int before = __shfl_up(m, 1);
            if (lane == 0) before = -0x7fffffff - 1;
            if (in) { rb_base[b] = carry + ex; rb_cmax[b] = max(carry_max, before); }
            carry += tot;
            carry_max = max(carry_max, __shfl(m, 63));
        }
    }
}

__global__ void __launch_bounds__(256) k_lane_emit(int n_blocks, const int *rb_part, const int *rb_r0, const i64 *part_rep_off,
                                                   const u64 *key_sorted, const int *val_sorted, const int *rep_weight, const int *rep_last,
                                                   const i64 *rep_exon_off, const i64 *rb_base, const int *rb_cmax,
                                                   longlong2 *lane_ex, int *lane_start, int *lane_pmax, const i64 *rb_ebase,
                                                   const int *ex_ts, const int *ex_te, int2 *lane_lx, int2 *lex) {
    __shared__ int lds[16];
    __shared__ int wmax[4];
    __shared__ unsigned char owner_s[kLexChunk];
    __shared__ int eb_s[256];
    __shared__ i64 src_s[256];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const i64 i = (i64)rb_r0[blk] + threadIdx.x;
        const bool in = i < part_rep_off[rb_part[blk] + 1];
        const int r = in ? val_sorted[i] : 0;
        const int first = in ? (int)((unsigned)(key_sorted[i] & 0xffffffffULL) ^ 0x80000000u) : 0;
        const int w = in ? rep_weight[r] : 0;
        int m = in ? rep_last[r] : -0x7fffffff - 1;
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(m, d); if (lane >= d) m = max(m, y); }
        __syncthreads();
        int tot;
        const int ex = wg_exclusive_scan(w, lds, &tot);
        if (lane == 63) wmax[wave] = m;
        __syncthreads();
        int run = rb_cmax[blk];
        for (int w2 = 0; w2 < wave; ++w2) run = max(run, wmax[w2]);
        m = max(m, run);
        const longlong2 exr = in ? make_longlong2(rep_exon_off[r], rep_exon_off[r + 1]) : make_longlong2(0, 0);
        const int ne = (int)(exr.y - exr.x);
        int tot_e;
        const int ex_e = wg_exclusive_scan(ne, lds, &tot_e);
        if (in) {
            const i64 base = rb_base[blk] + ex, ebase = rb_ebase[blk] + ex_e;
            const int2 lx = make_int2((int)ebase, (int)(ebase + ne));
            for (int q = 0; q < w; ++q) { lane_ex[base + q] = exr; lane_start[base + q] = first; lane_pmax[base + q] = m; lane_lx[base + q] = lx; }
        }
        lex_copy_tile(ex_e, ne, exr.x, tot_e, rb_ebase[blk], ex_ts, ex_te, lex, owner_s, eb_s, src_s);
    }
}


// lanes of the chunk's partition whose [first, last] position range meets the chunk's genomic range [glo, ghi]
__global__ void __launch_bounds__(256) k_hist_ranges(int n_chunks, const int *hc_part, const int *hc_glo, const int *hc_ghi,
                                                     const i64 *part_lane_off, const int *lane_start, const int *lane_pmax,
                                                     i64 *hc_llo, i64 *hc_lhi) {
    for (int ch = blockIdx.x * blockDim.x + threadIdx.x; ch < n_chunks; ch += gridDim.x * blockDim.x) {
        const int p = hc_part[ch], glo = hc_glo[ch], ghi = hc_ghi[ch];
        const i64 L0 = part_lane_off[p], L1 = part_lane_off[p + 1];
        i64 a = L0, b = L1;
        while (a < b) { const i64 m = (a + b) >> 1; if (lane_pmax[m] < glo) a = m + 1; else b = m; }
        const i64 llo = a;
        b = L1;
        while (a < b) { const i64 m = (a + b) >> 1; if (lane_start[m] <= ghi) a = m + 1; else b = m; }
        hc_llo[ch] = llo; hc_lhi[ch] = a;
    }
}

}  // namespace fseg
