// seg_solve.h -- the definition of k_solve (S5 whole for a problem that sees few reads: a workgroup takes it from its candidates
// to its count table without leaving LDS).  A header because its eighteen instances are compiled by three translation units
// (seg_solve16 / 32 / 60 .hip: a size class each), side by side: the kernel is most of the library's build time.
#pragma once
#include "seg_kernels.h"

namespace fseg {

// SPLIT: the workgroup ends when its rounds are over -- the problem's count table and in() go to its slot of the hand-over
// arena (dpx_slot) and k_dpw, the next launch on the stream, does the DP with one wave and a fraction of the LDS.
template <int NM, typename CntT, typename V, bool SPLIT>
__global__ void __launch_bounds__(SolveCfg<NM>::kThreads, SolveCfg<NM>::kMinBlocks) k_solve(Status *st, int cls, int nm, i64 lb_h, i64 ln_h, ProblemArrays pr,
                                                                  const ProbDesc *desc, i64 prob_cap, const int *cand_y,
                                                                  const int2 *__restrict__ lane_lx, const int2 *__restrict__ lex,
                                                                  const double *h_table, int h_len, double tau, const int2 *thr_tab,
                                                                  int support, unsigned char *chosen,
                                                                  unsigned char *dpx, i64 dpx_stride,
                                                                  const int *__restrict__ wide_items FSEG_TPARAM) {
    using C = SolveCfg<NM>;
    constexpr int T = C::kThreads, NR = C::kRanges;
    constexpr int PACK = 4 / (int)sizeof(CntT);                    // counters per 32-bit read-modify-write
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int cy_s[NM + 4];
    __shared__ int iend_s[NM + 4];
    __shared__ int2 act_s[sizeof(CntT) == 1 ? kFuseLanes + 1 : kFuseLanesWide + 1];   // reads with coverage in the window: (first exon that meets it, how many do)
    __shared__ int act_wave[T / 64];
    const int rt_pairs = nm * (nm - 1) / 2;
    constexpr int rt_stride = NM + 1;             // compile-time row stride (odd: rows do not collide on LDS banks)
    uint4 *planes = reinterpret_cast<uint4 *>(smem);                                         // rt_pairs * 16 B; later M | in | A
    unsigned *cov = reinterpret_cast<unsigned *>(smem + (size_t)rt_pairs * 16);              // kSub * rt_stride * 4 B
    const unsigned shared_b = (unsigned)rt_pairs * 16 + (unsigned)(kSub * rt_stride * 4);
    CntT *cnt = reinterpret_cast<CntT *>(smem + ((shared_b + 15) & ~15u));                   // C(nm,3) counters (solve_shared_bytes)
    V *M = reinterpret_cast<V *>(smem);
    int *in_s = reinterpret_cast<int *>(M + rt_pairs);
    unsigned char *A = reinterpret_cast<unsigned char *>(in_s + rt_pairs);
    // the DP is one wave's (dp_solve_wave) except where its registers would not fit: the large class with 64-bit sums
    constexpr bool kWaveDp = NM <= 32 || sizeof(V) == 4;
    if (NM == kNMax && threadIdx.x == 0) { atomicAdd(&st->gate, 1u); if (sizeof(CntT) != 1) atomicAdd(&st->gate_wide, 1u); }   // placed: see k_gate
    // (lb_h >= 0: the host knows the lists' sizes -- the batch has been sized --, and the status record is not on the way to the first problem)
    if (lb_h < 0 && (i64)st->n_prob > prob_cap) return;              // lists incomplete (a run that only sizes the arenas)
    // cls < 0: every solve list (batches of few problems: one launch instead of three)
    const i64 list_base = lb_h >= 0 ? lb_h : (cls <= 0 ? 0 : (cls == 1 ? (i64)st->solve_cls[0] : (i64)st->solve_cls[0] + (i64)st->solve_cls[1]));
    const i64 list_n = lb_h >= 0 ? ln_h : (cls < 0 ? (i64)st->solve_cls[0] + (i64)st->solve_cls[1] + (i64)st->solve_cls[2] : (i64)st->solve_cls[cls]);
    const int r_lane = threadIdx.x & 63, w_rng = wave_id();
    const bool own_wg = (i64)gridDim.x >= list_n;                    // a workgroup per problem (workgroup-uniform)
    const unsigned aborted = stage_aborted(st);                      // (a waiter in front of this launch gave up: the lists may not exist)
#ifdef FSEG_SCORE_TIMING
    // diagnostic build: phase clocks of the class given by tacc[15] (slots 0..5 scoring phases, 8..12 the DP's)
    __shared__ unsigned long long tick_sink[16];
    const bool timed = (int)tacc[15] == cls;
    unsigned long long *tk = timed ? tacc : tick_sink;
    unsigned long long *dp_tacc = tk; unsigned long long dt_prev = wall_clock64();
#define FSEG_STICK(i) FSEG_DTICK(i)
#else
#define FSEG_STICK(i)
#endif
    for (i64 tt = blockIdx.x; tt < list_n; tt += gridDim.x) {         // static stride; the lists are in candidate order
#ifdef FSEG_SCORE_TIMING
        const unsigned long long t_prob0 = wall_clock64();
#endif
        // (wide_items: this launch goes over the list's problems that see more than kFuseLanes reads only -- list_n of them)
        if (wide_items && aborted) return;                           // (the list of wide problems is an index into the records: not followed blindly)
        const i64 t = wide_items ? (i64)uni(wide_items[list_base + tt]) : tt;
        const ProbDesc d = FSEG_LOAD_DESC(desc + list_base + t);    // (the list's own copy of the record: k_prob_emit)
        if (aborted) return;                                         // (workgroup-uniform)
        const int p = d.w0;
        const int n = d.n;
        __syncthreads();                                             // the previous problem's DP is done with LDS
        FSEG_STICK(0);
        unsigned char *slot = SPLIT ? dpx + t * dpx_stride : nullptr;           // (dpx: the class's first slot)
        if (n > nm || n > NM) { if (threadIdx.x == 0) atomicOr(&st->err, kErrOverflowNm); continue; }
        // a list's problems are shared by two launches: the 8-bit counters take those that KEEP at most 255 reads (a counter
        // counts reads with coverage in the window: about two thirds of those the problem sees), the 16-bit ones the rest --
        // k_prob_range has counted, the record says whose the problem is.
        if ((d.kind == kKindFusedWide) != (sizeof(CntT) != 1)) continue;
        if (d.lane_n > kFuseLanesWide) { if (threadIdx.x == 0) atomicOr(&st->err, kErrNeedWideDp); continue; }
        const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
        const int *cy = cand_y + d.c0;
        for (int j = threadIdx.x; j < n; j += T) cy_s[j] = cy[j];
        {
            uint4 *z = reinterpret_cast<uint4 *>(cnt);
            for (int x = threadIdx.x; x < (ntri * (int)sizeof(CntT) + 15) / 16; x += T) z[x] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        // this thread's pairs: (i, j), integer label bounds, ambiguity count -- in registers for the whole problem
        // (Tried: the pairs loaded once per workgroup, the reads' exon ranges requested with the candidates and their first exon
        // blocks with the threshold table -- three dependent loads instead of seven before the first round.  The registers that
        // keeps alive spill (mid class 61 -> 77 us), and a problem alone on the chip is no faster for it: its time is the LDS
        // phases and the DP, not these loads.)
        int pi[C::kSlots], pj[C::kSlots], th_hi[C::kSlots], th_lo[C::kSlots];
        unsigned amb_acc[C::kSlots];
#pragma unroll
        for (int s = 0; s < C::kSlots; ++s) {
            const int q = s * T + threadIdx.x;
            amb_acc[s] = 0; pi[s] = 0; pj[s] = 1; th_hi[s] = 0x7fffffff; th_lo[s] = -1;
            if (q < npairs) {
                const unsigned short ij = g_pair_ij[q];
                pi[s] = ij & 255; pj[s] = ij >> 8;
                label_thresholds_tab((i64)cy_s[pj[s]] - cy_s[pi[s]] + 1, thr_tab, h_table, h_len, tau, &th_hi[s], &th_lo[s]);
            }
        }
        if (threadIdx.x < n) {
            // iend_s[j] = number of i < j with cand_j - cand_i >= 5 (candidates ascending): binary search
            int j = threadIdx.x, lim = cy_s[j] - 5, lo = 0, hi = j;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (cy_s[mid] <= lim) lo = mid + 1; else hi = mid; }
            iend_s[j] = lo;
        }
        // ---- the reads that matter.  The lane range is a superset (reads sorted by first position, cut by a running maximum of
        //      last positions): about a third of its reads have no exon in the window at all.  Such a read is `nay` for every pair
        //      -- it adds nothing to out(), and to in() only where lo < 0 -- so the scoring rounds run over the others only, packed
        //      densely (fewer 64-read rounds), and each of those arrives with the exons that meet the window already located:
        //      exons are ordered, so they are consecutive -- the first with te >= cand_0 up to the last with ts < cand_{n-1}.
        const int cp0 = d.g0 + cy_s[0], c_last = d.g0 + cy_s[n - 1];
        int n_act = 0;
        for (int l0 = 0; l0 < d.lane_n; l0 += T) {
            const int l = l0 + (int)threadIdx.x;
            const bool in = l < d.lane_n;
#ifdef FSEG_ABLATE_COV
            const int2 ex = make_int2(0, 0);                         // diagnostic (wrong results): no exon access at all, two lanes in three kept
            int first_rel = 0, cnt = (l % 3) != 2;
#else
            const int2 ex = lane_lx[d.lane_lo + (in ? l : 0)];
            int first_rel, cnt;
            window_exons(lex, ex, cp0, c_last, &first_rel, &cnt);
#endif
            const int first = ex.x + first_rel;
            const bool act = in && cnt > 0;
            const u64 m = __ballot(act);
            if ((threadIdx.x & 63) == 0) act_wave[threadIdx.x >> 6] = __popcll(m);
            __syncthreads();
            int base = n_act, tot = 0;
            for (int w2 = 0; w2 < T / 64; ++w2) { const int v = act_wave[w2]; if (w2 < (int)(threadIdx.x >> 6)) base += v; tot += v; }
            {
                constexpr int kActCap = sizeof(CntT) == 1 ? kFuseLanes + 1 : kFuseLanesWide + 1;
                const int slot = base + __popcll(m & ((1ULL << (threadIdx.x & 63)) - 1ULL));
                if (act && slot < kActCap) act_s[slot] = make_int2(first, cnt);      // (beyond it: the other instance's problem)
            }
            n_act += tot;
            __syncthreads();
        }
        // (k_prob_range has counted the same reads by the same test: an 8-bit instance never meets more than its counters hold)
        if (sizeof(CntT) == 1 && n_act > kFuseLanes) { if (threadIdx.x == 0) atomicOr(&st->err, kErrWideMissed); continue; }      // (workgroup-uniform)
        FSEG_STICK(1);
        // this thread's share of a round's coverage: read r_lane, candidates [ja, jb) of 1 .. n-1 (at most kCovJ of them)
        constexpr int kCovJ = (NM - 1 + NR - 1) / NR;
        const int ja = 1 + (int)((i64)(n - 1) * w_rng / NR), jb = 1 + (int)((i64)(n - 1) * (w_rng + 1) / NR);
        int cjv[kCovJ];
#pragma unroll
        for (int u = 0; u < kCovJ; ++u) cjv[u] = ja + u < jb ? d.g0 + cy_s[ja + u] : cp0;     // beyond the share: an empty window
        for (int r0 = 0; r0 < n_act; r0 += kSub) {
            int n_valid = n_act - r0;
            if (n_valid > kSub) n_valid = kSub;
            // ---- A: window coverage cov[r][j] = positions of the read's closed exons in [cand_0, cand_j)
            //      (get_cumulative_coverage :188-246) = sum over its exons of |[ts, te] n [cand_0, cand_j)|, over the few exons that
            //      meet the window (two per step: their loads depend on nothing but the LDS record, so they fly together)
            {
                const bool valid = r_lane < n_valid;
                const int2 a = act_s[r0 + (valid ? r_lane : 0)];
                int acc[kCovJ];
#pragma unroll
                for (int u = 0; u < kCovJ; ++u) acc[u] = 0;
#ifdef FSEG_ABLATE_COV
                const int e_end = 0;
#else
                const int e_end = valid ? a.y : 0;
#endif
                for (int e = 0; e < e_end; e += 4) {
                    // four exons per round trip (what lies beyond the read's own exons is masked below; the arrays are padded);
                    // the second pair is worked on only if some read of the wave has it
                    const int4u x01 = *reinterpret_cast<const int4u *>(lex + a.x + e), x23 = *reinterpret_cast<const int4u *>(lex + a.x + e + 2);
                    {
                        const int a0 = max(x01.x, cp0), b0 = x01.y + 1;                     // closed exon -> half-open end
                        const int a1 = max(x01.z, cp0), b1 = e + 1 < e_end ? x01.w + 1 : a1;  // (an odd count: the second slot is empty)
#pragma unroll
                        for (int v = 0; v < kCovJ; ++v) acc[v] += max(0, min(b0, cjv[v]) - a0) + max(0, min(b1, cjv[v]) - a1);
                    }
                    if (e + 2 < e_end) {
                        const int a0 = max(x23.x, cp0), b0 = x23.y + 1;
                        const int a1 = max(x23.z, cp0), b1 = e + 3 < e_end ? x23.w + 1 : a1;
#pragma unroll
                        for (int v = 0; v < kCovJ; ++v) acc[v] += max(0, min(b0, cjv[v]) - a0) + max(0, min(b1, cjv[v]) - a1);
                    }
                }
#pragma unroll
                for (int u = 0; u < kCovJ; ++u) if (ja + u < jb) cov[r_lane * rt_stride + ja + u] = (unsigned)acc[u];
                if (w_rng == 0) cov[r_lane * rt_stride] = 0;
            }
            lds_barrier();
            FSEG_STICK(2);
            // ---- B: pair planes (read b of a plane word lands on bit 31-b, as in k_score) -----------------------------
            const int nv1 = n_valid - 32;
            const unsigned valid0 = n_valid >= 32 ? 0xffffffffu : (n_valid > 0 ? ~(0xffffffffu >> n_valid) : 0u);
            const unsigned valid1 = nv1 >= 32 ? 0xffffffffu : (nv1 > 0 ? ~(0xffffffffu >> nv1) : 0u);
#pragma unroll
            for (int s = 0; s < C::kSlots; ++s) {
                const int q = s * T + threadIdx.x;
                if (q < npairs) {
                    const int i = pi[s], j = pj[s], hi = th_hi[s], lo = th_lo[s];
                    unsigned y0 = 0, z0 = 0, y1 = 0, z1 = 0;
#define FSEG_SHIFT_IN(acc, cmp, a, b) asm("v_cmp_" cmp "_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc")
                    // (a round's last reads rarely fill a word: a problem keeps ~80 reads, 64 + 16 -- the partly filled word
                    // costs its reads, not thirty-two; its bits are then moved up to where the full words' are)
                    if (n_valid >= 32) {
#pragma unroll
                        for (int b = 0; b < 32; ++b) {
                            int dd = (int)(cov[b * rt_stride + j] - cov[b * rt_stride + i]);
                            FSEG_SHIFT_IN(y0, "ge", dd, hi);
                            FSEG_SHIFT_IN(z0, "le", dd, lo);
                        }
                    } else {
                        for (int b = 0; b < n_valid; ++b) {
                            int dd = (int)(cov[b * rt_stride + j] - cov[b * rt_stride + i]);
                            FSEG_SHIFT_IN(y0, "ge", dd, hi);
                            FSEG_SHIFT_IN(z0, "le", dd, lo);
                        }
                        y0 <<= 32 - n_valid; z0 <<= 32 - n_valid;          // (1 <= n_valid <= 31)
                    }
                    if (nv1 >= 32) {
#pragma unroll
                        for (int b = 0; b < 32; ++b) {
                            int dd = (int)(cov[(32 + b) * rt_stride + j] - cov[(32 + b) * rt_stride + i]);
                            FSEG_SHIFT_IN(y1, "ge", dd, hi);
                            FSEG_SHIFT_IN(z1, "le", dd, lo);
                        }
                    } else if (nv1 > 0) {
                        for (int b = 0; b < nv1; ++b) {
                            int dd = (int)(cov[(32 + b) * rt_stride + j] - cov[(32 + b) * rt_stride + i]);
                            FSEG_SHIFT_IN(y1, "ge", dd, hi);
                            FSEG_SHIFT_IN(z1, "le", dd, lo);
                        }
                        y1 <<= 32 - nv1; z1 <<= 32 - nv1;
                    }
#undef FSEG_SHIFT_IN
                    y0 &= valid0; z0 &= valid0; y1 &= valid1; z1 &= valid1;     // rows beyond the problem's reads hold nothing
                    planes[q] = make_uint4(y0, y1, z0, z1);
                    amb_acc[s] += __popc(~(y0 | z0) & valid0) + __popc(~(y1 | z1) & valid1);
                }
            }
            lds_barrier();
            FSEG_STICK(3);
            // ---- C: triples; the (j,k) pairs are enumerated with j descending (lanes of a wave share the trip count) ---
#pragma unroll
            for (int s = 0; s < C::kSlots; ++s) {
                const int r = s * T + threadIdx.x;
                if (r >= npairs) continue;
                const int m = pj[s], x = pi[s];                       // pair r = (x, m): m = n-1-j in [1, n-1], x = k-j-1 in [0, m)
                const int j = n - 1 - m, kk = j + 1 + x;
                if (j == 0 || cy_s[kk] - cy_s[j] < 5) continue;       // dp(): segment too small (:540)
                const uint4 B = planes[kk * (kk - 1) / 2 + j];
                if ((B.x | B.y | B.z | B.w) == 0) continue;
                const int tbase = kk * (kk - 1) * (kk - 2) / 6 + j * (j - 1) / 2;
                const int abase = j * (j - 1) / 2;
                const int i_end = iend_s[j];                          // i with cand_j - cand_i >= 5 (:540), a prefix
                // (a read is never yea AND nay of one pair -- lo < hi --, so the two cross terms of a plane word are disjoint: one
                // popcount of their union, six instructions per triple and round instead of eight)
                // (a round of at most 32 reads has nothing in the second words: half the instructions)
#define FSEG_TRI_CNT(Av) (HALF ? __popc(((Av).x & B.z) | ((Av).z & B.x)) \
                               : __popc(((Av).x & B.z) | ((Av).z & B.x)) + __popc(((Av).y & B.w) | ((Av).w & B.y)))
                CntT *o = cnt + tbase;
                auto row = [&](auto half_c) {
                    constexpr bool HALF = decltype(half_c)::value;
                    int i = 0;
                    for (; i < i_end && ((tbase + i) & (PACK - 1)); ++i) {          // up to a 32-bit boundary of the table
                        const uint4 Av = planes[abase + i];
                        o[i] = (CntT)(o[i] + FSEG_TRI_CNT(Av));
                    }
                    for (; i + PACK <= i_end; i += PACK) {                // PACK counters per 32-bit read-modify-write: a counter
                        unsigned add = 0;                                 // never exceeds the reads of the problem, so no carry
#pragma unroll
                        for (int u = 0; u < PACK; ++u) { const uint4 Av = planes[abase + i + u]; add |= (unsigned)FSEG_TRI_CNT(Av) << (8 * (int)sizeof(CntT) * u); }
                        *reinterpret_cast<unsigned *>(o + i) += add;
                    }
                    for (; i < i_end; ++i) {
                        const uint4 Av = planes[abase + i];
                        o[i] = (CntT)(o[i] + FSEG_TRI_CNT(Av));
                    }
                };
                if (n_valid <= 32) row(std::true_type{}); else row(std::false_type{});
#undef FSEG_TRI_CNT
            }
            lds_barrier();
            FSEG_STICK(4);
        }
        // ---- DP on the tables where they lie: the planes' LDS becomes M | in | A ---------------------------------------
        // a read outside the lane range, or dropped above, has no coverage in the window: ambiguous exactly where lo < 0 (h >= 1)
        int in_val[C::kSlots];
#pragma unroll
        for (int s = 0; s < C::kSlots; ++s) in_val[s] = -(int)((i64)amb_acc[s] + (th_lo[s] < 0 ? (i64)d.outside + (d.lane_n - n_act) : 0));
        __syncthreads();
        if constexpr (SPLIT) {
            // hand-over: in() per pair (kDeadPair where the segment is too small, :540; the pair (0, end) keeps its value: it is
            // "no cut", :560, and never a link) and the count table as it lies (whose counters these are follows from the reads
            // the problem keeps: k_dpw decides as this kernel did)
            int *g_in = reinterpret_cast<int *>(slot + kDpxHeader);
#pragma unroll
            for (int s = 0; s < C::kSlots; ++s) {
                const int q = s * T + threadIdx.x;
                if (q < npairs) {
                    const bool dead = cy_s[pj[s]] - cy_s[pi[s]] < 5 && !(pi[s] == 0 && pj[s] == n - 1);
                    g_in[q] = dead ? kDeadPair : in_val[s];
                }
            }
            {
                uint4 *g_out = reinterpret_cast<uint4 *>(slot + kDpxHeader + dpx_in_bytes(nm));
                const uint4 *l_out = reinterpret_cast<const uint4 *>(cnt);
                for (int x = threadIdx.x; x < (ntri * (int)sizeof(CntT) + 15) / 16; x += T) g_out[x] = l_out[x];
            }
            FSEG_STICK(9);
        } else if constexpr (kWaveDp) {
            // the pairs' owners hand each pair over whole: in(b,c), or kDeadPair where the segment is too small (:540; the pair
            // (0, end) keeps its value: it is "no cut", :560, and never a link), and c
#pragma unroll
            for (int s = 0; s < C::kSlots; ++s) {
                const int q = s * T + threadIdx.x;
                if (q < npairs) {
                    const bool dead = cy_s[pj[s]] - cy_s[pi[s]] < 5 && !(pi[s] == 0 && pj[s] == n - 1);
                    in_s[q] = dead ? kDeadPair : in_val[s];
                    A[q] = (unsigned char)pj[s];
                }
            }
            __syncthreads();
            FSEG_STICK(9);
            // The other waves are done with this problem.  When every problem of the list has a workgroup of its own (the usual
            // launch) they END here, and what they held is free for the next workgroup while wave 0 walks the DP's chain; else
            // they go on to the next problem's descriptor and wait at the barrier at the top of the loop.
            if (w_rng != 0) { if (own_wg) return; continue; }
#ifdef FSEG_ABLATE_DP
            if (own_wg) return;          // diagnostic (wrong results): what the stage takes when a workgroup's LDS is free once its rounds are over
#endif
            const int chain = dp_solve_wave<NM>(n, cnt, in_s, M, A, support, chosen + d.c0 FSEG_DARG);
            if (threadIdx.x == 0) pr.chain[p] = chain;
        } else {
#pragma unroll
        for (int s = 0; s < C::kSlots; ++s) { const int q = s * T + threadIdx.x; if (q < npairs) in_s[q] = in_val[s]; }
        __syncthreads();
        FSEG_STICK(9);
        const int chain = dp_solve_push<T, NM>(n, cnt, in_s, M, A, cy_s, support, chosen + d.c0 FSEG_DARG);
        if (threadIdx.x == 0) pr.chain[p] = chain;
        }
#ifdef FSEG_SCORE_TIMING
        if (threadIdx.x == 0) FSEG_PROB_TICK(p, t_prob0, d.lane_n, n_act);
#endif
    }
#undef FSEG_STICK
}

}  // namespace fseg
