// seg_score_arena.hip -- S5 for problems that see many reads -- the arena path: pair thresholds, window coverage tiles, k_score per size class, the
// DP kernels, and the slow-but-complete kernels for problems of 61 .. 128 (huge) and 129 .. 1024 (giant) candidates.
// Part of libfreddie_seg.so (gfx950); shared definitions: seg_common.h, declarations: seg_kernels.h, launches: freddie_seg.hip.
#include "seg_kernels.h"

namespace fseg {

__global__ void __launch_bounds__(256) k_pair_thresholds(const Status *st, ProblemArrays pr, const ProbDesc *desc, i64 prob_cap,
                                                         const i64 *cand_off, const int *cand_y, const double *h_table,
                                                         int h_len, double tau, int2 *pair_thr, i64 pair_cap,
                                                         unsigned *amb_g, unsigned *out_g, i64 tri_cap) {
    pair_thresholds_blocks(blockIdx.x, gridDim.x, st, pr, desc, prob_cap, cand_y, h_table, h_len, tau, pair_thr, pair_cap, amb_g, out_g,
                           tri_cap);
}



// ---------------------------------------------------------------------------------------------
// S5b  window coverage of every (problem, read)      get_cumulative_coverage (:188-246)
// cov[j] = #positions of the read's closed exons in [cand_0, cand_j) = C[start+j] - C[start].
// One thread per read of the problem's read range; the read's (ordered) exon list is merged
// against the problem's n candidates.  Layout of a work item's block: [j][kLaneChunk reads], so
// both this kernel's stores and the scoring kernel's loads are coalesced.  Also records, per work
// item, which 64-read sub-chunks contain a read with any coverage in the window.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kLaneChunk) k_cov(Status *st, const ProbDesc *desc, i64 prob_cap, const int2 *work_pc,
                                                    i64 work_cap, const i64 *cand_off,
                                                    const int *cand_y, const int *iv_start, const int2 *__restrict__ lane_lx,
                                                    const int2 *__restrict__ lex,
                                                    unsigned *cov_g, i64 cov_cap, unsigned char *work_active,
                                                    int cov_blocks, ProblemArrays pr, const double *h_table, int h_len, double tau,
                                                    int2 *pair_thr, i64 pair_cap, unsigned *amb_g, unsigned *out_g, i64 tri_cap) {
    // One-partition batches are chains of launch-latency-sized kernels: there the pair thresholds (which, like the
    // coverage, need only the problem list) ride along as the workgroups behind the coverage ones -- one graph node less,
    // and the two overlap (cov_blocks == gridDim.x: no such workgroups, k_pair_thresholds was launched on its own).
    if ((int)blockIdx.x >= cov_blocks) {
        pair_thresholds_blocks(blockIdx.x - cov_blocks, gridDim.x - cov_blocks, st, pr, desc, prob_cap, cand_y, h_table, h_len, tau,
                               pair_thr, pair_cap, amb_g, out_g, tri_cap);
        return;
    }
    __shared__ int cp[kNGiant + 4];
    __shared__ u64 work_s;
    __shared__ unsigned active_s;
    i64 n_work = (i64)st->n_work;
    if (n_work > work_cap || (i64)st->n_prob > prob_cap) return;   // lists incomplete: this run only sizes the arenas
    for (i64 w = blockIdx.x; w < n_work; w += cov_blocks) {     // static stride: a shared work counter saturates near 90 pops/us
        __syncthreads();
        if (threadIdx.x == 0) active_s = 0;
        __syncthreads();
        const int2 pc = work_pc[w];
        const int chunk = pc.y;
        const ProbDesc d = load_desc(desc + pc.x);
        const int n = d.n;
        i64 coff = d.cov_off + (i64)chunk * kLaneChunk * n;
        if (n > kNGiant || coff + (i64)kLaneChunk * n > cov_cap) { if (threadIdx.x == 0) work_active[w] = 0; continue; }
        const int *cy = cand_y + d.c0;
        const int g0 = d.g0;
        int t = threadIdx.x;
        int li = chunk * kLaneChunk + t;
        // the read's exon range needs only the descriptor: in flight together with the candidate positions
        const int2 ex = lane_lx[d.lane_lo + (li < d.lane_n ? li : 0)];      // (its piece of the lane-ordered (ts, te) stream)
        for (int j = threadIdx.x; j < n; j += blockDim.x) cp[j] = g0 + cy[j];
        __syncthreads();
        const int cp0 = cp[0];
        unsigned *dst = cov_g + coff + t;
        unsigned last = 0;
        if (li < d.lane_n) {
            i64 e = ex.x, e1 = ex.y;
            {   // first exon whose closed interval reaches cand_0 (exons of a read are ordered, :158)
                i64 lo = e, hi = e1;
                while (lo < hi) { i64 mid = (lo + hi) >> 1; if (lex[mid].y < cp0) lo = mid + 1; else hi = mid; }
                e = lo;
            }
            unsigned acc = 0;
            int ts = 0, te = -1;
            if (e < e1) { const int2 x = lex[e]; ts = x.x; te = x.y; }
            dst[0] = 0;
            for (int j = 1; j < n; ++j) {
                int cj = cp[j];
                while (e < e1 && te < cj) {
                    acc += (unsigned)(te + 1 - (ts > cp0 ? ts : cp0));
                    ++e;
                    if (e < e1) { const int2 x = lex[e]; ts = x.x; te = x.y; }
                }
                unsigned part_cov = 0;
                if (e < e1 && ts < cj) part_cov = (unsigned)(cj - (ts > cp0 ? ts : cp0));
                last = acc + part_cov;
                dst[(i64)j * kLaneChunk] = last;
            }
        } else {
            for (int j = 0; j < n; ++j) dst[(i64)j * kLaneChunk] = 0;
        }
        u64 any = __ballot(last > 0);
        if (lane_id() == 0 && any) atomicOr(&active_s, 1u << (threadIdx.x >> 6));
        __syncthreads();
        if (threadIdx.x == 0) work_active[w] = (unsigned char)active_s;
    }
}

template <int NM>
__global__ void __launch_bounds__(ScoreCfg<NM>::kThreads) k_score(Status *st, int cls, int nm, ProblemArrays pr, i64 prob_cap,
                                                                  const int4 *cls_items, const ProbDesc *desc,
                                                                  i64 work_cap, const i64 *cand_off,
                                                                  const int *cand_y, const unsigned char *work_active,
                                                                  const unsigned *cov_g, i64 cov_cap, const int2 *pair_thr,
                                                                  i64 pair_cap, unsigned *out_g, i64 tri_cap,
                                                                  unsigned *amb_g FSEG_TPARAM) {
    using C = ScoreCfg<NM>;
    constexpr int T = C::kThreads;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int cy_s[NM + 4];
    __shared__ int iend_s[NM + 4];
    __shared__ u64 work_s;
    // LDS carve-up for problems of at most nm candidates (nm <= NM is chosen by the host from the previous run's
    // largest problem, so that a batch of moderately sized problems gets more workgroups per CU)
    const int rt_pairs = nm * (nm - 1) / 2;
    constexpr int rt_stride = C::kCovStride;      // compile-time row stride: LDS addresses fold into instruction offsets
    uint4 *planes = reinterpret_cast<uint4 *>(smem);                                         // rt_pairs * 16 B
    unsigned *cov = reinterpret_cast<unsigned *>(smem + (size_t)rt_pairs * 16);              // kSub * rt_stride * 4 B
    unsigned short *pair_ij = reinterpret_cast<unsigned short *>(cov + kSub * rt_stride);    // rt_pairs * 2 B
    unsigned short *out16 = pair_ij + ((rt_pairs + 7) & ~7);                                 // C(nm,3) * 2 B
    for (int q = threadIdx.x; q < rt_pairs; q += T) pair_ij[q] = g_pair_ij[q];
    // cls < 0: this launch takes the work items of every size class (small batches: one launch instead of three)
    i64 cls_base = cls < 0 ? 0 : (cls >= 1 ? (i64)st->cls_work[0] : 0) + (cls >= 2 ? (i64)st->cls_work[1] : 0);
    i64 n_items = cls < 0 ? (i64)st->n_work - (i64)st->cls_work[3] : (i64)st->cls_work[cls];   // never the huge class
    u64 *queue = &st->cls_queue[cls < 0 ? 0 : cls];
    if ((i64)st->n_work > work_cap || (i64)st->n_prob > prob_cap) n_items = 0;   // lists incomplete: sizing run
    FSEG_T0;
    // a shared work counter saturates near 90 pops/us, so the small classes claim several items per pop
    constexpr int kPop = NM <= 16 ? 8 : (NM <= 32 ? 4 : 1);
#ifndef FSEG_SCORE_STATIC
#define FSEG_SCORE_STATIC 1
#endif
    // The small classes hold tens of thousands of short items: even batched pops serialise on the one counter, so
    // they take a static stride (neighbouring items are of similar size); the big class keeps the counter.
    constexpr bool kStatic = FSEG_SCORE_STATIC && NM <= 32;
    i64 wi_base = 0;
    int wi_left = 0;
    i64 wi_static = blockIdx.x;
    for (;;) {
        i64 wi;
        if (kStatic) {
            wi = cls < 0 ? -1 : wi_static;
            wi_static += gridDim.x;
        }
        if (!kStatic || wi < 0) {
            if (wi_left == 0) {
                __syncthreads();
                if (threadIdx.x == 0) work_s = atomicAdd(queue, (u64)kPop);
                __syncthreads();
                wi_base = (i64)work_s;
                wi_left = kPop;
            }
            wi = wi_base + (kPop - wi_left);
            --wi_left;
        }
        __syncthreads();
        FSEG_TICK(0);
        if (wi >= n_items) break;
        const int4 item = cls_items[cls_base + wi];
        const i64 w = item.x;
        const int p = item.y, chunk = item.z;
        const ProbDesc d = load_desc(desc + p);
        const int n = d.n;
        i64 poff = d.pair_off, toff = d.tri_off;
        i64 coff = d.cov_off + (i64)chunk * kLaneChunk * n;
        int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
        if (n > nm) { if (threadIdx.x == 0) atomicOr(&st->err, kErrOverflowNm); continue; }
        if (poff + npairs > pair_cap || toff + ntri > tri_cap || coff + (i64)kLaneChunk * n > cov_cap) continue;
        bool zero_ambiguous = (pr.flags[p] & 1) != 0;
        unsigned active = zero_ambiguous ? 0xfu : work_active[w];
        int lanes_here = d.lane_n - chunk * kLaneChunk;
        if (lanes_here > kLaneChunk) lanes_here = kLaneChunk;
        if (lanes_here < kLaneChunk) active &= (1u << ((lanes_here + kSub - 1) / kSub)) - 1u;
        if (active == 0) continue;                       // no read of this chunk touches the window
        const int *cy = cand_y + d.c0;
        for (int j = threadIdx.x; j < n; j += T) cy_s[j] = cy[j];
        {
            uint4 *z = reinterpret_cast<uint4 *>(out16);              // 8 counters per store
            for (int x = threadIdx.x; x < (ntri + 7) / 8; x += T) z[x] = make_uint4(0, 0, 0, 0);
        }
        unsigned amb_acc[C::kSlots];
#pragma unroll
        for (int s = 0; s < C::kSlots; ++s) amb_acc[s] = 0;
        // coverage tile of the first active sub-chunk into registers (global layout is [j][reads])
        unsigned stage[C::kStage];
        int sub = __ffs(active) - 1;
        const int tile_words = n * kSub;
#pragma unroll
        for (int e = 0; e < C::kStage; ++e) {
            int x = e * T + threadIdx.x;
            stage[e] = x < tile_words ? cov_g[coff + (i64)(x >> 6) * kLaneChunk + sub * kSub + (x & 63)] : 0;
        }
        __syncthreads();
        if (threadIdx.x < n) {
            // iend_s[j] = number of i < j with cand_j - cand_i >= 5 (candidates ascending): binary search
            int j = threadIdx.x, lim = cy_s[j] - 5, lo = 0, hi = j;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (cy_s[mid] <= lim) lo = mid + 1; else hi = mid; }
            iend_s[j] = lo;
        }
        __syncthreads();
        FSEG_TICK(1);
        while (sub >= 0) {
            int n_valid = lanes_here - sub * kSub;
            if (n_valid > kSub) n_valid = kSub;
            // ---- A: registers -> LDS cov[r][j]; start fetching the next active tile ---------------------
#pragma unroll
            for (int e = 0; e < C::kStage; ++e) {
                int x = e * T + threadIdx.x;
                if (x < tile_words) cov[(x & 63) * rt_stride + (x >> 6)] = stage[e];
            }
            unsigned rest = active & ~((2u << sub) - 1u);
            int next_sub = rest ? __ffs(rest) - 1 : -1;
            if (next_sub >= 0) {
#pragma unroll
                for (int e = 0; e < C::kStage; ++e) {
                    int x = e * T + threadIdx.x;
                    stage[e] = x < tile_words ? cov_g[coff + (i64)(x >> 6) * kLaneChunk + next_sub * kSub + (x & 63)] : 0;
                }
            }
            lds_barrier();
            FSEG_TICK(2);
            // ---- B: pair planes ---------------------------------------------------------------------
            const int nv1 = n_valid - 32;
            unsigned valid0 = n_valid >= 32 ? 0xffffffffu : (n_valid > 0 ? ~(0xffffffffu >> n_valid) : 0u);
            unsigned valid1 = nv1 >= 32 ? 0xffffffffu : (nv1 > 0 ? ~(0xffffffffu >> nv1) : 0u);
#pragma unroll
            for (int s = 0; s < C::kSlots; ++s) {
                int q = s * T + threadIdx.x;
                if (q < npairs) {
                    int i = pair_ij[q] & 255, j = pair_ij[q] >> 8;
                    int2 th = pair_thr[poff + q];
                    // shift the compare result into the plane word through the carry: acc = 2*acc + (d >= hi).
                    // Read b of a word therefore lands on bit 31-b (the valid masks below use the same order).
                    unsigned y0 = 0, z0 = 0, y1 = 0, z1 = 0;
#define FSEG_SHIFT_IN(acc, cmp, a, b) asm("v_cmp_" cmp "_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc")
#pragma unroll
                    for (int b = 0; b < 32; ++b) {
                        int d = (int)(cov[b * rt_stride + j] - cov[b * rt_stride + i]);
                        FSEG_SHIFT_IN(y0, "ge", d, th.x);
                        FSEG_SHIFT_IN(z0, "le", d, th.y);
                    }
#pragma unroll
                    for (int b = 0; b < 32; ++b) {
                        int d = (int)(cov[(32 + b) * rt_stride + j] - cov[(32 + b) * rt_stride + i]);
                        FSEG_SHIFT_IN(y1, "ge", d, th.x);
                        FSEG_SHIFT_IN(z1, "le", d, th.y);
                    }
#undef FSEG_SHIFT_IN
                    planes[q] = make_uint4(y0, y1, z0, z1);
                    amb_acc[s] += __popc(~(y0 | z0) & valid0) + __popc(~(y1 | z1) & valid1);
                }
            }
            lds_barrier();
            FSEG_TICK(3);
            // ---- C: triples.  B pairs (j,k) are enumerated with j descending so that the 64 lanes of a wave
            // have (nearly) the same trip count j and mostly share the (i,j) plane they read --------------
            for (int r = threadIdx.x; r < npairs; r += T) {
                int m = pair_ij[r] >> 8, x = pair_ij[r] & 255;        // m = n-1-j in [1, n-1], x = k-j-1 in [0, m)
                int j = n - 1 - m, kk = j + 1 + x;
                if (j == 0 || cy_s[kk] - cy_s[j] < 5) continue;       // dp(): segment too small (:540)
                uint4 B = planes[kk * (kk - 1) / 2 + j];
                if ((B.x | B.y | B.z | B.w) == 0) continue;
                int tbase = kk * (kk - 1) * (kk - 2) / 6 + j * (j - 1) / 2;
                int abase = j * (j - 1) / 2;
                const int i_end = iend_s[j];                          // i with cand_j - cand_i >= 5 (:540), a prefix
        // (a read is never yea AND nay of one pair -- lo < hi --, so the two cross terms of a plane word are disjoint: one popcount of their union)
#define FSEG_TRI_CNT(A) (__popc(((A).x & B.z) | ((A).z & B.x)) + __popc(((A).y & B.w) | ((A).w & B.y)))
                unsigned short *o16 = out16 + tbase;
                int i = 0;
                if ((tbase & 1) && i_end > 0) {                       // align to a counter pair
                    uint4 A = planes[abase];
                    o16[0] += (unsigned short)FSEG_TRI_CNT(A);
                    i = 1;
                }
                for (; i + 1 < i_end; i += 2) {                       // two u16 counters per 32-bit read-modify-write
                    uint4 A0 = planes[abase + i], A1 = planes[abase + i + 1];
                    unsigned add = FSEG_TRI_CNT(A0) | (FSEG_TRI_CNT(A1) << 16);
                    *reinterpret_cast<unsigned *>(o16 + i) += add;    // a counter never exceeds the reads of a work item (< 65536)
                }
                if (i < i_end) {
                    uint4 A = planes[abase + i];
                    o16[i] += (unsigned short)FSEG_TRI_CNT(A);
                }
#undef FSEG_TRI_CNT
            }
            lds_barrier();
            FSEG_TICK(4);
            sub = next_sub;
        }
        // ---- flush ---------------------------------------------------------------------------------
        for (int x = threadIdx.x; x < ntri; x += T) {
            unsigned v = out16[x];
            if (v) atomicAdd(&out_g[toff + x], v);
        }
#pragma unroll
        for (int s = 0; s < C::kSlots; ++s) {
            int q = s * T + threadIdx.x;
            if (q < npairs && amb_acc[s]) atomicAdd(&amb_g[poff + q], amb_acc[s]);
        }
        FSEG_TICK(5);
    }
}


template <int NM, int T, typename OutT>
__global__ void __launch_bounds__(T) k_dp(Status *st, int dp_class, int nm, const int *dp_items, ProblemArrays pr, const ProbDesc *desc, i64 prob_cap,
                                            const i64 *cand_off, const int *cand_y, const int *iv_part,
                                            const i64 *part_lane_off, const unsigned *out_g, i64 tri_cap,
                                            const unsigned *amb_g, const int2 *pair_thr, i64 pair_cap, int support,
                                            unsigned char *chosen, int n_lo, int dp_blocks, uint4 *labels16, i64 labels_n16 FSEG_TPARAM) {
    if ((int)blockIdx.x >= dp_blocks) {                 // the workgroups behind the DP ones: label arena fill
        fill_labels(labels16, labels_n16, (i64)(blockIdx.x - dp_blocks) * T + threadIdx.x, (i64)(gridDim.x - dp_blocks) * T);
        return;
    }
    // handles problems with n_lo < n <= NM that the scoring kernel did not finish itself (more than one
    // work item); the out table of the problem is staged in LDS first
    // LDS carve-up for problems of at most nm <= NM candidates (nm from the previous run's largest problem)
    const int kTri = nm * (nm - 1) * (nm - 2) / 6, kPairs = nm * (nm - 1) / 2;
    extern __shared__ __align__(16) unsigned char smem[];
    i64 *M = reinterpret_cast<i64 *>(smem);                          // M(b,c), b < c, at c*(c-1)/2 + b
    int *in_s = reinterpret_cast<int *>(M + kPairs);
    OutT *out_s = reinterpret_cast<OutT *>(in_s + kPairs);            // counts: 16 bit when every problem sees < 65536 reads
    unsigned char *A = reinterpret_cast<unsigned char *>(out_s + ((kTri + 3) & ~3));
    __shared__ int cy_s[NM];
    i64 n_prob = (i64)st->n_prob;
    if (n_prob > prob_cap) return;                                  // sizing run
    // dp_class 0 / 1: the small / big problems of the per-class list; -1: every problem
    const i64 list_base = dp_class == 1 ? (i64)st->dp_cls[0] : 0;
    const i64 list_n = dp_class < 0 ? n_prob : (i64)st->dp_cls[dp_class];
#ifdef FSEG_SCORE_TIMING
    unsigned long long *dp_tacc = tacc; unsigned long long dt_prev = wall_clock64();
#endif
    for (i64 t = blockIdx.x; t < list_n; t += dp_blocks) {          // static stride (no shared work counter)
        __syncthreads();
        FSEG_DTICK(8);
        const i64 p = dp_class < 0 ? t : (i64)dp_items[list_base + t];
        const ProbDesc d = load_desc(desc + p);
        int n = d.n;
        if (n > NM || n <= n_lo || d.kind != kKindArena) continue;   // (when this launch walks every problem: k_tiny's and k_solve's are not its own)
        if (n > nm) { if (threadIdx.x == 0) atomicOr(&st->err, kErrOverflowNm); continue; }
        if (sizeof(OutT) == 2 && d.lane_n >= 65536) { if (threadIdx.x == 0) atomicOr(&st->err, kErrNeedWideDp); continue; }
        i64 poff = d.pair_off, toff = d.tri_off;
        int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
        if (poff + npairs > pair_cap || toff + ntri > tri_cap) continue;
        const i64 c0 = d.c0;
        const i64 outside = d.outside;
        const bool zamb = (pr.flags[p] & 1) != 0;
        for (int j = threadIdx.x; j < n; j += blockDim.x) cy_s[j] = cand_y[c0 + j];
        for (int x0 = threadIdx.x; x0 < ntri; x0 += T * 8) {       // 8 loads in flight per thread
            unsigned v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { int x = x0 + e * T; v[e] = x < ntri ? out_g[toff + x] : 0u; }
#pragma unroll
            for (int e = 0; e < 8; ++e) { int x = x0 + e * T; if (x < ntri) out_s[x] = (OutT)v[e]; }
        }
        for (int q = threadIdx.x; q < npairs; q += blockDim.x)
            in_s[q] = -(int)((i64)amb_g[poff + q] + ((zamb && pair_thr[poff + q].y < 0) ? outside : 0));
        __syncthreads();
        FSEG_DTICK(9);
        int chain = dp_solve_push<T, NM>(n, out_s, in_s, M, A, cy_s, support, chosen + c0 FSEG_DARG);
        if (threadIdx.x == 0) pr.chain[p] = chain;
    }
}

template <typename OutT>
__global__ void __launch_bounds__(256, 6) k_dp_waves(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc, i64 prob_cap,
                                                  const int *cand_y, const unsigned *out_g, i64 tri_cap, const unsigned *amb_g,
                                                  const int2 *pair_thr, i64 pair_cap, int support, unsigned char *chosen, int coop) {
    constexpr int T = 256, NM = kDpSmall;
    constexpr int kTri = NM * (NM - 1) * (NM - 2) / 6, kPairs = NM * (NM - 1) / 2;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int cy_s[NM];
    __shared__ int cy_w[4][kDpWave];
    __shared__ int big_s[4];
    const i64 n_prob = (i64)st->n_prob;
    if (n_prob > prob_cap) return;                                  // sizing run
    const i64 list_n = (i64)st->dp_cls[0];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
#ifdef FSEG_SCORE_TIMING
    __shared__ unsigned long long tick_sink[16];                    // the diagnostic build's ticks of this kernel are dropped
    unsigned long long *dp_tacc = tick_sink; unsigned long long dt_prev = 0;
#endif
    // a workgroup's four entries are a grid apart, not neighbours: the list is in candidate order, neighbouring problems
    // come from the same gene and are of similar size, and four large ones in one workgroup would be solved one after
    // the other while the rest of the GPU is already idle
    for (i64 g = (i64)blockIdx.x; g < list_n; g += (i64)gridDim.x * 4) {
        __syncthreads();
        {   // ---- every wave: its own entry ------------------------------------------------------------------
            const i64 t = g + (i64)wave * gridDim.x;
            int big = -1;
            if (t < list_n) {
                const i64 p = dp_items[t];
                const ProbDesc d = load_desc(desc + p);
                const int n = d.n;
                const i64 poff = d.pair_off, toff = d.tri_off;
                const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
                const bool usable = n <= NM && poff + npairs <= pair_cap && toff + ntri <= tri_cap;
                if (sizeof(OutT) == 2 && d.lane_n >= 65536) { if (lane == 0) atomicOr(&st->err, kErrNeedWideDp); }
                else if (usable && n > kDpWave) big = coop ? (int)p : -1;   // coop == 0: a k_dp launch of its own takes these
                else if (usable) {
                    unsigned char *w_mem = smem + (size_t)wave * dp_wave_bytes<OutT>();
                    i64 *M = reinterpret_cast<i64 *>(w_mem);
                    int *in_s = reinterpret_cast<int *>(M + kDpWavePairs);
                    OutT *out_s = reinterpret_cast<OutT *>(in_s + kDpWavePairs);
                    unsigned char *A = reinterpret_cast<unsigned char *>(out_s + ((kDpWaveTri + 3) & ~3));
                    const bool zamb = (pr.flags[p] & 1) != 0;
                    if (lane < n) cy_w[wave][lane] = cand_y[d.c0 + lane];
                    for (int x0 = lane; x0 < ntri; x0 += 64 * 4) {
                        unsigned v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { int x = x0 + e * 64; v[e] = x < ntri ? out_g[toff + x] : 0u; }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { int x = x0 + e * 64; if (x < ntri) out_s[x] = (OutT)v[e]; }
                    }
                    for (int q = lane; q < npairs; q += 64)
                        in_s[q] = -(int)((i64)amb_g[poff + q] + ((zamb && pair_thr[poff + q].y < 0) ? (i64)d.outside : 0));
                    dp_sync<64>();
                    int chain = dp_solve_push<64, kDpWave>(n, out_s, in_s, M, A, cy_w[wave], support, chosen + d.c0 FSEG_DARG);
                    if (lane == 0) pr.chain[p] = chain;
                }
            }
            if (lane == 0) big_s[wave] = big;
        }
        __syncthreads();
        // ---- the workgroup: entries with more than kDpWave candidates, one after the other ----------------------
        for (int w = 0; w < 4; ++w) {
            const int pb = big_s[w];
            if (pb < 0) continue;                                       // uniform: big_s is shared
            i64 *M = reinterpret_cast<i64 *>(smem);
            int *in_s = reinterpret_cast<int *>(M + kPairs);
            OutT *out_s = reinterpret_cast<OutT *>(in_s + kPairs);
            unsigned char *A = reinterpret_cast<unsigned char *>(out_s + ((kTri + 3) & ~3));
            const ProbDesc d = load_desc(desc + pb);
            const int n = d.n;
            const i64 poff = d.pair_off, toff = d.tri_off;
            const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
            const bool zamb = (pr.flags[pb] & 1) != 0;
            __syncthreads();
            for (int j = threadIdx.x; j < n; j += T) cy_s[j] = cand_y[d.c0 + j];
            for (int x0 = threadIdx.x; x0 < ntri; x0 += T * 8) {
                unsigned v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { int x = x0 + e * T; v[e] = x < ntri ? out_g[toff + x] : 0u; }
#pragma unroll
                for (int e = 0; e < 8; ++e) { int x = x0 + e * T; if (x < ntri) out_s[x] = (OutT)v[e]; }
            }
            for (int q = threadIdx.x; q < npairs; q += T)
                in_s[q] = -(int)((i64)amb_g[poff + q] + ((zamb && pair_thr[poff + q].y < 0) ? (i64)d.outside : 0));
            __syncthreads();
            int chain = dp_solve_push<T, NM>(n, out_s, in_s, M, A, cy_s, support, chosen + d.c0 FSEG_DARG);
            if (threadIdx.x == 0) pr.chain[pb] = chain;
        }
    }
}

__global__ void __launch_bounds__(512) k_score_huge(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc,
                                                    i64 prob_cap, i64 work_cap, const int *cand_y, const unsigned *cov_g,
                                                    i64 cov_cap, const int2 *pair_thr, i64 pair_cap, unsigned *out_g,
                                                    i64 tri_cap, unsigned *amb_g) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint2 *planes = reinterpret_cast<uint2 *>(smem);                                   // {yea, nay} per pair
    unsigned *cov = reinterpret_cast<unsigned *>(smem + (size_t)(kNHuge * (kNHuge - 1) / 2) * 8);   // [read][j], stride kNHuge + 1
    __shared__ int cy_s[kNHuge];
    __shared__ int iend_s[kNHuge];
    constexpr int T = 512, stride = kNHuge + 1;
    if ((i64)st->n_work > work_cap || (i64)st->n_prob > prob_cap) return;              // sizing run
    const i64 list_base = (i64)st->dp_cls[0] + (i64)st->dp_cls[1], list_n = (i64)st->dp_cls[2];
    for (i64 t = blockIdx.x; t < list_n; t += gridDim.x) {
        __syncthreads();
        const int p = dp_items[list_base + t];
        const ProbDesc d = load_desc(desc + p);
        const int n = d.n;
        if (n > kNHuge) continue;
        const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
        const i64 poff = d.pair_off, toff = d.tri_off;
        const int n_chunks = (d.lane_n + kLaneChunk - 1) / kLaneChunk;
        if (poff + npairs > pair_cap || toff + ntri > tri_cap || d.cov_off + (i64)n_chunks * kLaneChunk * n > cov_cap) continue;
        for (int j = threadIdx.x; j < n; j += T) cy_s[j] = cand_y[d.c0 + j];
        __syncthreads();
        if (threadIdx.x < n) {
            int j = threadIdx.x, lim = cy_s[j] - 5, lo = 0, hi = j;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (cy_s[mid] <= lim) lo = mid + 1; else hi = mid; }
            iend_s[j] = lo;                                        // number of i < j with cand_j - cand_i >= 5 (:540)
        }
        const bool zero_ambiguous = (pr.flags[p] & 1) != 0;
        for (int r0 = 0; r0 < d.lane_n; r0 += kHugeSub) {
            const int nr = d.lane_n - r0 < kHugeSub ? d.lane_n - r0 : kHugeSub;
            const int chunk = r0 / kLaneChunk, in_chunk = r0 % kLaneChunk;
            const unsigned *src = cov_g + d.cov_off + (i64)chunk * kLaneChunk * n + in_chunk;   // [j][256 reads]
            __syncthreads();
            for (int x = threadIdx.x; x < n * kHugeSub; x += T) {
                const int j = x / kHugeSub, b = x % kHugeSub;
                cov[b * stride + j] = b < nr ? src[(i64)j * kLaneChunk + b] : 0u;
            }
            __syncthreads();
            const unsigned valid = nr >= 32 ? 0xffffffffu : ((1u << nr) - 1u);
            for (int q = threadIdx.x; q < npairs; q += T) {
                int i, j;
                pair_decode(q, &i, &j);
                const int2 th = pair_thr[poff + q];
                unsigned y = 0, z = 0;
                for (int b = 0; b < kHugeSub; ++b) {
                    const int dd = (int)(cov[b * stride + j] - cov[b * stride + i]);
                    y |= (unsigned)(dd >= th.x) << b;              // yea: covered fraction above the high threshold
                    z |= (unsigned)(dd <= th.y) << b;              // nay: below the low threshold
                }
                y &= valid; z &= valid;
                planes[q] = make_uint2(y, z);
                const unsigned amb = __popc(~(y | z) & valid);
                if (amb) amb_g[poff + q] += amb;                    // this workgroup owns the problem: plain update
            }
            __syncthreads();
            // triples: thread = (j,k), loop over the i with cand_j - cand_i >= 5; counters of a (j,k) are contiguous
            for (int r = threadIdx.x; r < npairs; r += T) {
                int j, kk;
                pair_decode(r, &j, &kk);
                if (j == 0 || cy_s[kk] - cy_s[j] < 5) continue;
                const uint2 B = planes[r];
                if ((B.x | B.y) == 0) continue;
                unsigned *o = out_g + toff + (i64)kk * (kk - 1) * (kk - 2) / 6 + j * (j - 1) / 2;
                const int abase = j * (j - 1) / 2, i_end = iend_s[j];
                for (int i = 0; i < i_end; ++i) {
                    const uint2 A = planes[abase + i];
                    const unsigned cnt = __popc((A.x & B.y) | (A.y & B.x));          // (disjoint: a read is never yea and nay of one pair)
                    if (cnt) o[i] += cnt;
                }
            }
        }
        (void)zero_ambiguous;                                       // zero-coverage reads outside the lane range: added in the DP
    }
}

__global__ void __launch_bounds__(512) k_dp_huge(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc,
                                                 i64 prob_cap, const int *cand_y, const unsigned *out_g, i64 tri_cap,
                                                 const unsigned *amb_g, const int2 *pair_thr, i64 pair_cap, int support,
                                                 unsigned char *chosen) {
    constexpr int T = 512, kSlices = 4, kB = T / kSlices;          // thread = (b, c2 slice); kB == kNHuge
    constexpr int kPairs = kNHuge * (kNHuge - 1) / 2;
    extern __shared__ __align__(16) unsigned char smem[];
    i64 *M = reinterpret_cast<i64 *>(smem);
    int *in_s = reinterpret_cast<int *>(M + kPairs);
    unsigned char *A = reinterpret_cast<unsigned char *>(in_s + kPairs);
    __shared__ int cy_s[kNHuge];
    __shared__ i64 part_v[T];
    __shared__ unsigned char part_a[T];
    __shared__ i64 top_v[T / 64];
    __shared__ int top_key[T / 64];
    if ((i64)st->n_prob > prob_cap) return;
    const i64 list_base = (i64)st->dp_cls[0] + (i64)st->dp_cls[1], list_n = (i64)st->dp_cls[2];
    const int b = 1 + threadIdx.x % kB, slice = threadIdx.x / kB;
    for (i64 t = blockIdx.x; t < list_n; t += gridDim.x) {
        __syncthreads();
        const int p = dp_items[list_base + t];
        const ProbDesc d = load_desc(desc + p);
        const int n = d.n;
        if (n > kNHuge) continue;
        const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6, end = n - 1;
        const i64 poff = d.pair_off, toff = d.tri_off;
        if (poff + npairs > pair_cap || toff + ntri > tri_cap) continue;
        const unsigned *out_p = out_g + toff;
        const i64 outside = d.outside;
        const bool zamb = (pr.flags[p] & 1) != 0;
        for (int j = threadIdx.x; j < n; j += T) cy_s[j] = cand_y[d.c0 + j];
        for (int q = threadIdx.x; q < npairs; q += T)
            in_s[q] = -(int)((i64)amb_g[poff + q] + ((zamb && pair_thr[poff + q].y < 0) ? outside : 0));
        __syncthreads();
#define FSEG_IN(a, bb) ((i64)in_s[(bb) * ((bb) - 1) / 2 + (a)])
#define FSEG_M(a, bb) M[(bb) * ((bb) - 1) / 2 + (a)]
        for (int x = threadIdx.x; x < end; x += T) {
            FSEG_M(x, end) = cy_s[end] - cy_s[x] >= 5 ? FSEG_IN(x, end) : kNegInf;
            A[end * (end - 1) / 2 + x] = 255;
        }
        __syncthreads();
        for (int c = end - 1; c >= 2; --c) {
            i64 best = kNegInf; int arg = 255;
            const bool live = b < c && cy_s[c] - cy_s[b] >= 5;
            if (live) {
                const int base = c * (c - 1) / 2 + b;
                for (int c2 = c + 1 + slice; c2 <= end; c2 += kSlices) {
                    const i64 tail = FSEG_M(c, c2);
                    const unsigned o = out_p[(i64)c2 * (c2 - 1) * (c2 - 2) / 6 + base];
                    const bool ok = (tail != kNegInf) & ((i64)o >= (i64)support);
                    const i64 cur = ok ? (i64)o + tail : kNegInf;
                    const bool take = cur > best;
                    best = take ? cur : best; arg = take ? c2 : arg;
                }
            }
            part_v[threadIdx.x] = best; part_a[threadIdx.x] = (unsigned char)arg;
            __syncthreads();
            if (slice == 0 && b < c) {
                i64 bv = best; int ba = arg;
                for (int s2 = 1; s2 < kSlices; ++s2) {
                    const i64 v = part_v[s2 * kB + b - 1]; const int a2 = part_a[s2 * kB + b - 1];
                    if (v > bv || (v == bv && v != kNegInf && a2 < ba)) { bv = v; ba = a2; }
                }
                FSEG_M(b, c) = (live && bv != kNegInf) ? bv + FSEG_IN(b, c) : kNegInf;
                A[c * (c - 1) / 2 + b] = (unsigned char)ba;
            }
            __syncthreads();
        }
        // top level (:560-566): first maximiser in (j, k) order, taken only if strictly better than no cut
        i64 bv = kNegInf; int bkey = 0x7fffffff;
        for (int q = threadIdx.x; q < npairs; q += T) {
            int j, kx;
            pair_decode(q, &j, &kx);
            if (j < 1) continue;
            if (cy_s[j] - cy_s[0] < 5 || cy_s[kx] - cy_s[j] < 5) continue;
            const i64 tail = FSEG_M(j, kx);
            const unsigned o = out_p[(i64)kx * (kx - 1) * (kx - 2) / 6 + j * (j - 1) / 2];
            if (tail == kNegInf || (i64)o < (i64)support) continue;
            const i64 cur = FSEG_IN(0, j) + (i64)o + tail;
            const int key = j * 256 + kx;
            if (cur > bv || (cur == bv && key < bkey)) { bv = cur; bkey = key; }
        }
        for (int dd = 32; dd >= 1; dd >>= 1) {
            i64 ov = __shfl_xor(bv, dd); int ok2 = __shfl_xor(bkey, dd);
            if (ov > bv || (ov == bv && ok2 < bkey)) { bv = ov; bkey = ok2; }
        }
        if (lane_id() == 0) { top_v[threadIdx.x >> 6] = bv; top_key[threadIdx.x >> 6] = bkey; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < T / 64; ++w)
                if (top_v[w] > bv || (top_v[w] == bv && top_key[w] < bkey)) { bv = top_v[w]; bkey = top_key[w]; }
            int chain = 0;
            if (bv != kNegInf && bv > FSEG_IN(0, end)) {
                int j = bkey >> 8, k = bkey & 255;
                unsigned char *ch = chosen + d.c0;
                ch[0] = 1;
                for (;;) {
                    ch[j] = 1; ch[k] = 1; ++chain;
                    if (k == end) break;
                    int k2 = A[k * (k - 1) / 2 + j];
                    if (k2 == 255) break;
                    j = k; k = k2;
                }
            }
            pr.chain[p] = chain;
        }
#undef FSEG_IN
#undef FSEG_M
    }
}

__global__ void __launch_bounds__(512) k_score_giant(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc,
                                                     i64 prob_cap, i64 work_cap, const int *cand_y, const unsigned *cov_g,
                                                     i64 cov_cap, const int2 *pair_thr, i64 pair_cap, unsigned *out_g,
                                                     i64 tri_cap, unsigned *amb_g, int nm, unsigned char *scratch, i64 scratch_stride) {
    extern __shared__ __align__(16) unsigned char smem[];
    int *cy_s = reinterpret_cast<int *>(smem);
    int *iend_s = cy_s + nm;
    unsigned *cov = reinterpret_cast<unsigned *>(iend_s + nm);                 // [read][j], stride nm + 1
    uint2 *planes = reinterpret_cast<uint2 *>(scratch + (i64)blockIdx.x * scratch_stride);   // {yea, nay} per pair
    constexpr int T = 512;
    const int stride = nm + 1;
    if ((i64)st->n_work > work_cap || (i64)st->n_prob > prob_cap) return;              // sizing run
    const i64 list_base = (i64)st->dp_cls[0] + (i64)st->dp_cls[1], list_n = (i64)st->dp_cls[2];
    for (i64 t = blockIdx.x; t < list_n; t += gridDim.x) {
        __syncthreads();
        const int p = dp_items[list_base + t];
        const ProbDesc d = load_desc(desc + p);
        const int n = d.n;
        if (n <= kNHuge || n > nm) continue;                                          // (the huge kernels' problems; nm covers the run's largest)
        const int npairs = n * (n - 1) / 2;
        const i64 ntri = (i64)n * (n - 1) * (n - 2) / 6;
        const i64 poff = d.pair_off, toff = d.tri_off;
        const int n_chunks = (d.lane_n + kLaneChunk - 1) / kLaneChunk;
        if (poff + npairs > pair_cap || toff + ntri > tri_cap || d.cov_off + (i64)n_chunks * kLaneChunk * n > cov_cap) continue;
        for (int j = threadIdx.x; j < n; j += T) cy_s[j] = cand_y[d.c0 + j];
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += T) {
            int lim = cy_s[j] - 5, lo = 0, hi = j;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (cy_s[mid] <= lim) lo = mid + 1; else hi = mid; }
            iend_s[j] = lo;                                        // number of i < j with cand_j - cand_i >= 5 (:540)
        }
        for (int r0 = 0; r0 < d.lane_n; r0 += kHugeSub) {
            const int nr = d.lane_n - r0 < kHugeSub ? d.lane_n - r0 : kHugeSub;
            const int chunk = r0 / kLaneChunk, in_chunk = r0 % kLaneChunk;
            const unsigned *src = cov_g + d.cov_off + (i64)chunk * kLaneChunk * n + in_chunk;   // [j][256 reads]
            __syncthreads();
            for (int x = threadIdx.x; x < n * kHugeSub; x += T) {
                const int j = x / kHugeSub, b = x % kHugeSub;
                cov[b * stride + j] = b < nr ? src[(i64)j * kLaneChunk + b] : 0u;
            }
            __syncthreads();
            const unsigned valid = nr >= 32 ? 0xffffffffu : ((1u << nr) - 1u);
            for (int q = threadIdx.x; q < npairs; q += T) {
                int i, j;
                pair_decode(q, &i, &j);
                const int2 th = pair_thr[poff + q];
                unsigned y = 0, z = 0;
                for (int b = 0; b < kHugeSub; ++b) {
                    const int dd = (int)(cov[b * stride + j] - cov[b * stride + i]);
                    y |= (unsigned)(dd >= th.x) << b;              // yea: covered fraction above the high threshold
                    z |= (unsigned)(dd <= th.y) << b;              // nay: below the low threshold
                }
                y &= valid; z &= valid;
                planes[q] = make_uint2(y, z);
                const unsigned amb = __popc(~(y | z) & valid);
                if (amb) amb_g[poff + q] += amb;                    // this workgroup owns the problem: plain update
            }
            __syncthreads();
            // triples: thread = (j,k), loop over the i with cand_j - cand_i >= 5; counters of a (j,k) are contiguous
            for (int r = threadIdx.x; r < npairs; r += T) {
                int j, kk;
                pair_decode(r, &j, &kk);
                if (j == 0 || cy_s[kk] - cy_s[j] < 5) continue;
                const uint2 B = planes[r];
                if ((B.x | B.y) == 0) continue;
                unsigned *o = out_g + toff + (i64)kk * (kk - 1) * (kk - 2) / 6 + j * (j - 1) / 2;
                const int abase = j * (j - 1) / 2, i_end = iend_s[j];
                for (int i = 0; i < i_end; ++i) {
                    const uint2 A = planes[abase + i];
                    const unsigned cnt = __popc((A.x & B.y) | (A.y & B.x));          // (disjoint: a read is never yea and nay of one pair)
                    if (cnt) o[i] += cnt;
                }
            }
        }
    }
}

__global__ void __launch_bounds__(512) k_dp_giant(Status *st, const int *dp_items, ProblemArrays pr, const ProbDesc *desc,
                                                  i64 prob_cap, const int *cand_y, const unsigned *out_g, i64 tri_cap,
                                                  const unsigned *amb_g, const int2 *pair_thr, i64 pair_cap, int support,
                                                  unsigned char *chosen, int nm, unsigned char *scratch, i64 scratch_stride) {
    constexpr int T = 512;
    constexpr unsigned short kNoArg = 0xffffu;
    extern __shared__ __align__(16) unsigned char smem[];
    int *cy_s = reinterpret_cast<int *>(smem);
    __shared__ i64 top_v[T / 64];
    __shared__ i64 top_key[T / 64];
    const i64 rt_pairs = (i64)nm * (nm - 1) / 2;
    i64 *M = reinterpret_cast<i64 *>(scratch + (i64)blockIdx.x * scratch_stride);
    int *in_s = reinterpret_cast<int *>(M + rt_pairs);
    unsigned short *A = reinterpret_cast<unsigned short *>(in_s + rt_pairs);
    if ((i64)st->n_prob > prob_cap) return;
    const i64 list_base = (i64)st->dp_cls[0] + (i64)st->dp_cls[1], list_n = (i64)st->dp_cls[2];
    for (i64 t = blockIdx.x; t < list_n; t += gridDim.x) {
        __syncthreads();
        const int p = dp_items[list_base + t];
        const ProbDesc d = load_desc(desc + p);
        const int n = d.n;
        if (n <= kNHuge || n > nm) continue;
        const int npairs = n * (n - 1) / 2, end = n - 1;
        const i64 ntri = (i64)n * (n - 1) * (n - 2) / 6;
        const i64 poff = d.pair_off, toff = d.tri_off;
        if (poff + npairs > pair_cap || toff + ntri > tri_cap) continue;
        const unsigned *out_p = out_g + toff;
        const i64 outside = d.outside;
        const bool zamb = (pr.flags[p] & 1) != 0;
        for (int j = threadIdx.x; j < n; j += T) cy_s[j] = cand_y[d.c0 + j];
        for (int q = threadIdx.x; q < npairs; q += T)
            in_s[q] = -(int)((i64)amb_g[poff + q] + ((zamb && pair_thr[poff + q].y < 0) ? outside : 0));
        __syncthreads();
#define FSEG_IN(a, bb) ((i64)in_s[(bb) * ((bb) - 1) / 2 + (a)])
#define FSEG_M(a, bb) M[(bb) * ((bb) - 1) / 2 + (a)]
        for (int x = threadIdx.x; x < end; x += T) {
            FSEG_M(x, end) = cy_s[end] - cy_s[x] >= 5 ? FSEG_IN(x, end) : kNegInf;
            A[end * (end - 1) / 2 + x] = kNoArg;
        }
        __syncthreads();
        for (int c = end - 1; c >= 2; --c) {
            // M(b,c) = in(b,c) + max over c2 > c of out(b,c,c2) + M(c,c2), first maximiser (:550-555); a thread per b
            for (int b = 1 + threadIdx.x; b < c; b += T) {
                i64 best = kNegInf; int arg = kNoArg;
                const bool live = cy_s[c] - cy_s[b] >= 5;
                if (live) {
                    const int base = c * (c - 1) / 2 + b;
                    for (int c2 = c + 1; c2 <= end; ++c2) {
                        const i64 tail = FSEG_M(c, c2);
                        const unsigned o = out_p[(i64)c2 * (c2 - 1) * (c2 - 2) / 6 + base];
                        const bool ok = (tail != kNegInf) & ((i64)o >= (i64)support);
                        const i64 cur = ok ? (i64)o + tail : kNegInf;
                        const bool take = cur > best;
                        best = take ? cur : best; arg = take ? c2 : arg;
                    }
                }
                FSEG_M(b, c) = (live && best != kNegInf) ? best + FSEG_IN(b, c) : kNegInf;
                A[c * (c - 1) / 2 + b] = (unsigned short)arg;
            }
            __syncthreads();
        }
        // top level (:560-566): first maximiser in (j, k) order, taken only if strictly better than no cut
        i64 bv = kNegInf, bkey = 0x7fffffffffffffffLL;
        for (int q = threadIdx.x; q < npairs; q += T) {
            int j, kx;
            pair_decode(q, &j, &kx);
            if (j < 1) continue;
            if (cy_s[j] - cy_s[0] < 5 || cy_s[kx] - cy_s[j] < 5) continue;
            const i64 tail = FSEG_M(j, kx);
            const unsigned o = out_p[(i64)kx * (kx - 1) * (kx - 2) / 6 + j * (j - 1) / 2];
            if (tail == kNegInf || (i64)o < (i64)support) continue;
            const i64 cur = FSEG_IN(0, j) + (i64)o + tail;
            const i64 key = ((i64)j << 20) | (i64)kx;
            if (cur > bv || (cur == bv && key < bkey)) { bv = cur; bkey = key; }
        }
        for (int dd = 32; dd >= 1; dd >>= 1) {
            i64 ov = __shfl_xor(bv, dd); i64 ok2 = __shfl_xor(bkey, dd);
            if (ov > bv || (ov == bv && ok2 < bkey)) { bv = ov; bkey = ok2; }
        }
        if (lane_id() == 0) { top_v[threadIdx.x >> 6] = bv; top_key[threadIdx.x >> 6] = bkey; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < T / 64; ++w)
                if (top_v[w] > bv || (top_v[w] == bv && top_key[w] < bkey)) { bv = top_v[w]; bkey = top_key[w]; }
            int chain = 0;
            if (bv != kNegInf && bv > FSEG_IN(0, end)) {
                int j = (int)(bkey >> 20), k = (int)(bkey & 0xfffff);
                unsigned char *ch = chosen + d.c0;
                ch[0] = 1;
                for (;;) {
                    ch[j] = 1; ch[k] = 1; ++chain;
                    if (k == end) break;
                    const int k2 = A[k * (k - 1) / 2 + j];
                    if (k2 == kNoArg) break;
                    j = k; k = k2;
                }
            }
            pr.chain[p] = chain;
        }
#undef FSEG_IN
#undef FSEG_M
    }
}


// the instances the host launches (freddie_seg.hip sees the declarations only: taking an instance's address here is what
// instantiates it -- host stub and device code -- in this translation unit)
__attribute__((used)) static const void *const kInstances[] = {
    reinterpret_cast<const void *>(&k_score<kClsSmall>),
    reinterpret_cast<const void *>(&k_score<kClsMid>),
    reinterpret_cast<const void *>(&k_score<kNMax>),
    reinterpret_cast<const void *>(&k_dp<kNMax, 512, unsigned>),
    reinterpret_cast<const void *>(&k_dp<kNMax, 256, unsigned>),
    reinterpret_cast<const void *>(&k_dp<kNMax, 512, unsigned short>),
    reinterpret_cast<const void *>(&k_dp<kDpSmall, 256, unsigned>),
    reinterpret_cast<const void *>(&k_dp<kDpSmall, 256, unsigned short>),
    reinterpret_cast<const void *>(&k_dp_waves<unsigned>),
    reinterpret_cast<const void *>(&k_dp_waves<unsigned short>),
};

}  // namespace fseg
