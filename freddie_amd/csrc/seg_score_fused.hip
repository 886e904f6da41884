// seg_score_fused.hip -- S5 for problems that see few reads, whole: k_solve (a workgroup per problem; + k_dpw, its DP as a launch of one-wave
// workgroups), k_wave / k_tiny (a wave per tiny problem), and the stage's device-side gate / fork / join kernels.
// Part of libfreddie_seg.so (gfx950); shared definitions: seg_common.h, declarations: seg_kernels.h, launches: freddie_seg.hip.
#include "seg_kernels.h"

namespace fseg {


// (6 workgroups = 24 waves per CU asked of the register allocator: the kernel is a chain of dependent loads, and at the
// 120 registers it would otherwise take only 16 waves fit; measured 56 -> 47 us on config4, 8 spills and is slower)
#ifndef FSEG_TINY_OCC
#define FSEG_TINY_OCC 5
#endif
__global__ void __launch_bounds__(256, FSEG_TINY_OCC) k_tiny(Status *st, const ProbDesc *desc, i64 prob_cap, int tiny_max, ProblemArrays pr,
                                              const int *cand_y, const longlong2 *lane_ex, const int *ex_ts,
                                              const int *ex_te, const double *h_table, int h_len, double tau, const int2 *thr_tab,
                                              int support, unsigned char *chosen, i64 lb_h, i64 ln_h FSEG_TPARAM) {
    __shared__ u64 planes[4][kTinyPairs][2];            // [wave][pair]{yea, nay} of the current 64 reads
    __shared__ i64 M_s[4][kTinyPairs];
    __shared__ int in_s[4][kTinyPairs];
    __shared__ unsigned out_s[4][kTinyTri + 4];
    __shared__ unsigned char A_s[4][kTinyPairs + 4];
    __shared__ int cy_s[4][kTiny];
    __shared__ unsigned char tri_ijk[kTinyTri][4];
    __shared__ int2 act_w[4][128];                      // [wave] reads with coverage in the window, waiting for a round: (first exon, count)
    const int lane = lane_id(), wave = wave_id();
    const u64 lt_mask = (1ULL << lane) - 1ULL;
    if (threadIdx.x < kTinyTri) {                       // rank t = k(k-1)(k-2)/6 + j(j-1)/2 + i  ->  (i, j, k)
        int t = threadIdx.x, k = 2;
        while ((k + 1) * k * (k - 1) / 6 <= t) ++k;
        int i, j;
        pair_decode(t - k * (k - 1) * (k - 2) / 6, &i, &j);
        tri_ijk[t][0] = (unsigned char)i; tri_ijk[t][1] = (unsigned char)j; tri_ijk[t][2] = (unsigned char)k;
    }
    __syncthreads();
    // (lb_h >= 0: the host knows the lists' sizes -- the batch has been sized --, and the status record is not on the way to the first problem)
    if (lb_h < 0 && (i64)st->n_prob > prob_cap) return;                      // sizing run: the descriptors are incomplete
#ifdef FSEG_SCORE_TIMING
    __shared__ unsigned long long tick_sink[16];
    unsigned long long *dp_tacc = tick_sink; unsigned long long dt_prev = 0;
#endif
    // k_tiny's problems are a list of their own (behind the three solve lists): every wave of a workgroup has one, and a
    // workgroup's four are a grid apart (neighbours in the list are neighbours on the genome and of similar size)
    const i64 list_base = lb_h >= 0 ? lb_h : (i64)st->solve_cls[0] + (i64)st->solve_cls[1] + (i64)st->solve_cls[2];
    const i64 list_n = lb_h >= 0 ? ln_h : (i64)st->n_tiny;
    for (i64 t = (i64)blockIdx.x + (i64)wave * gridDim.x; t < list_n; t += (i64)gridDim.x * 4) {
#ifdef FSEG_SCORE_TIMING
        const unsigned long long t_prob0 = wall_clock64();
#endif
        const ProbDesc d = FSEG_LOAD_DESC(desc + list_base + t);          // (the list's own copy of the record: k_prob_emit)
        const int p = d.w0;
        const int n = d.n;
        if (n > tiny_max) continue;                     // wave-uniform
        const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
        // The problem is a chain of dependent loads (descriptor -> candidates -> thresholds; descriptor -> exon ranges -> exons):
        // the second branch needs nothing of the first, so the first 64 reads' exon ranges and first exon blocks are requested
        // now and arrive while the candidates and thresholds are being fetched.
        longlong2 pf_ex = make_longlong2(0, 0);
        int pf_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_te[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (d.lane_n > 0) {                             // (wave-uniform)
            pf_ex = lane_ex[d.lane_lo + (lane < d.lane_n ? lane : 0)];
            load_exons8(ex_ts + pf_ex.x, pf_ts); load_exons8(ex_te + pf_ex.x, pf_te);
        }
        dp_sync<64>();                                  // the previous problem's readers of the wave-private tables are done
        if (lane < n) cy_s[wave][lane] = cand_y[d.c0 + lane];
        dp_sync<64>();
        int hi_q = 0x7fffffff, lo_q = -1, pi = 0, pj = 1;
        if (lane < npairs) {
            pair_decode(lane, &pi, &pj);
            label_thresholds_tab((i64)cy_s[wave][pj] - cy_s[wave][pi] + 1, thr_tab, h_table, h_len, tau, &hi_q, &lo_q);   // :490-495 as integer bounds
        }
        int ti = 0, tj = 1, tk = 2;
        if (lane < ntri) { ti = tri_ijk[lane][0]; tj = tri_ijk[lane][1]; tk = tri_ijk[lane][2]; }
        const int cp0 = d.g0 + cy_s[wave][0], c_last = d.g0 + cy_s[wave][n - 1];
        int cj[kTiny];
#pragma unroll
        for (int j = 0; j < kTiny; ++j) cj[j] = j < n ? d.g0 + cy_s[wave][j] : cp0;      // beyond the problem: an empty window
        unsigned amb = 0, out = 0;
        // As in k_solve: only the reads with an exon in the window are scored (about 60 % of the lane range), packed into
        // full rounds of 64, each with its window exons located (they are consecutive: first with te >= cand_0 .. last with
        // ts < cand_{n-1}), so the coverage below is a sum of overlaps over two or three exons instead of a search and a walk.
        int raw = 0, fill = 0, n_act = 0;               // wave-uniform: lanes examined, records waiting, reads kept
        while (raw < d.lane_n || fill > 0) {
            while (fill < 64 && raw < d.lane_n) {
                const int li = raw + lane;
                const bool in = li < d.lane_n;
                longlong2 ex;
                int ts8[8], te8[8];
                if (raw == 0) {                                               // (wave-uniform) the block requested above
                    ex = pf_ex;
#pragma unroll
                    for (int u = 0; u < 8; ++u) { ts8[u] = pf_ts[u]; te8[u] = pf_te[u]; }
                } else {
                    ex = lane_ex[d.lane_lo + (in ? li : 0)];                  // unconditional: no branch around the load
                    load_exons8(ex_ts + ex.x, ts8); load_exons8(ex_te + ex.x, te8);
                }
                i64 first = ex.x;
                int cnt = 0;
                for (i64 eb = ex.x;;) {                                       // eight exons per round from clamped addresses
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool hit = eb + u < ex.y && te8[u] >= cp0 && ts8[u] < c_last;
                        if (hit && cnt == 0) first = eb + u;
                        cnt += hit;
                    }
                    if (ts8[7] >= c_last) break;
                    eb += 8;
                    if (eb >= ex.y) break;
                    load_exons8(ex_ts + eb, ts8); load_exons8(ex_te + eb, te8);
                }
                const bool act = in && cnt > 0;
                const u64 m = __ballot(act);
                if (act) act_w[wave][fill + __popcll(m & lt_mask)] = make_int2((int)first, cnt);
                fill += __popcll(m); n_act += __popcll(m);
                raw += 64;
            }
            dp_sync<64>();
            const int nv = fill < 64 ? fill : 64;
            const bool valid = lane < nv;
            // window coverage of this lane's read: cov[j] = positions of its closed exons in [cand_0, cand_j)
            int cov[kTiny];
#pragma unroll
            for (int j = 0; j < kTiny; ++j) cov[j] = 0;
            {
                const int2 a = act_w[wave][valid ? lane : 0];
                const int e_end = valid ? a.y : 0;
                for (int e = 0; e < e_end; e += 2) {
                    const int2 ts2 = load_exons2(ex_ts + a.x + e), te2 = load_exons2(ex_te + a.x + e);      // (the second may be the next read's: masked below)
                    const int tsa = ts2.x, tea = te2.x, tsb = ts2.y, teb = te2.y;
                    const int a0 = max(tsa, cp0), b0 = tea + 1;
                    const int a1 = max(tsb, cp0), b1 = e + 1 < e_end ? teb + 1 : a1;
#pragma unroll
                    for (int j = 1; j < kTiny; ++j) cov[j] += max(0, min(b0, cj[j]) - a0) + max(0, min(b1, cj[j]) - a1);
                }
            }
            const u64 vmask = __ballot(valid);
            // pair labels: one compare per plane, the 64 reads' bits arrive as the ballot; lane q keeps pair q's planes
            u64 my_yea = 0, my_nay = 0;
#pragma unroll
            for (int j = 1; j < kTiny; ++j) {
#pragma unroll
                for (int i = 0; i < j; ++i) {
                    const int q = j * (j - 1) / 2 + i;
                    if (q < npairs) {                   // wave-uniform
                        const int dd = cov[j] - cov[i];
                        const int hi = __builtin_amdgcn_readlane(hi_q, q), lo = __builtin_amdgcn_readlane(lo_q, q);
                        const u64 y = __ballot(valid && dd >= hi), z = __ballot(valid && dd <= lo);
                        if (lane == q) { my_yea = y; my_nay = z; }
                    }
                }
            }
            if (lane < npairs) {
                amb += (unsigned)__popcll(~(my_yea | my_nay) & vmask);       // neither label: ambiguous (:500-506)
                planes[wave][lane][0] = my_yea; planes[wave][lane][1] = my_nay;
            }
            // the records beyond this round move to the front of the list
            int2 keep = make_int2(0, 0);
            if (lane < fill - nv) keep = act_w[wave][64 + lane];
            dp_sync<64>();
            if (lane < fill - nv) act_w[wave][lane] = keep;
            fill -= nv;
            if (lane < ntri) {                          // out(i,j,k) (:509-528): the two labels exclude each other
                const int qa = tj * (tj - 1) / 2 + ti, qb = tk * (tk - 1) / 2 + tj;
                out += (unsigned)(__popcll(planes[wave][qa][0] & planes[wave][qb][1]) + __popcll(planes[wave][qa][1] & planes[wave][qb][0]));
            }
            dp_sync<64>();
        }
        const int dropped = d.lane_n - n_act;           // reads of the lane range without coverage: treated like those outside it
        // a read outside the lane range has no coverage in the window: ambiguous exactly where lo < 0 (h >= 1)
        if (lane < npairs) in_s[wave][lane] = -(int)((i64)amb + (lo_q < 0 ? (i64)d.outside + dropped : 0));
        if (lane < ntri) out_s[wave][lane] = out;
        dp_sync<64>();
        const int chain = dp_solve_push<64, kTiny>(n, out_s[wave], in_s[wave], M_s[wave], A_s[wave], cy_s[wave], support, chosen + d.c0 FSEG_DARG);
        if (lane == 0) pr.chain[p] = chain;
#ifdef FSEG_SCORE_TIMING
        if (lane == 0) FSEG_PROB_TICK(p, t_prob0, d.lane_n, n_act);
#endif
    }
}

template <int NM, typename V>
__global__ void __launch_bounds__(256, WaveCfg<NM>::kOcc) k_wave(Status *st, const ProbDesc *desc, i64 prob_cap, int list, ProblemArrays pr,
                                                                 const int *__restrict__ cand_y, const int2 *__restrict__ lane_lx, const int2 *__restrict__ lex,
                                                                 const double *h_table, int h_len, double tau, const int2 *__restrict__ thr_tab,
                                                                 int support, unsigned char *chosen, i64 lb_h, i64 ln_h FSEG_TPARAM) {
    using C = WaveCfg<NM>;
    __shared__ WaveLds<NM, V> lds4[4];
    __shared__ unsigned short tri_q[C::kTri + 2];      // triple rank t -> (pair (i,j)) | (pair (j,k)) << 8
    const int lane = lane_id(), wave = wave_id();
    WaveLds<NM, V> &L = lds4[wave];
    for (int t = threadIdx.x; t < C::kTri; t += 256) {
        int k = 2;
        while ((k + 1) * k * (k - 1) / 6 <= t) ++k;
        int i, j;
        pair_decode(t - k * (k - 1) * (k - 2) / 6, &i, &j);
        tri_q[t] = (unsigned short)((j * (j - 1) / 2 + i) | ((k * (k - 1) / 2 + j) << 8));
    }
    __syncthreads();
    if (lb_h < 0 && (i64)st->n_prob > prob_cap) return;             // sizing run: the descriptors are incomplete
#ifdef FSEG_SCORE_TIMING
    __shared__ unsigned long long tick_sink[16];
    unsigned long long *dp_tacc = tick_sink; unsigned long long dt_prev = 0;
#endif
    // list 3: k_tiny's problems (behind the three solve lists); list 0: the small class.  lb_h >= 0: bounds from the host (a sized batch)
    const i64 list_base = lb_h >= 0 ? lb_h : (list == 3 ? (i64)st->solve_cls[0] + (i64)st->solve_cls[1] + (i64)st->solve_cls[2] : 0);
    const i64 list_n = lb_h >= 0 ? ln_h : (list == 3 ? (i64)st->n_tiny : (i64)st->solve_cls[0]);
    const unsigned aborted = stage_aborted(st);                     // (a waiter in front of this launch gave up: the lists may not exist)
    for (i64 t = (i64)blockIdx.x + (i64)wave * gridDim.x; t < list_n; t += (i64)gridDim.x * 4) {
#ifdef FSEG_SCORE_TIMING
        const unsigned long long t_prob0 = wall_clock64();
#endif
        const ProbDesc d = FSEG_LOAD_DESC(desc + list_base + t);          // (the list's own copy of the record: k_prob_emit)
        if (aborted) return;
        const int p = d.w0;
        const int n = d.n;
        if (n > NM || n < 3) { if (lane == 0) atomicOr(&st->err, kErrOverflowNm); continue; }       // (wave-uniform)
        const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
        // two independent loads behind the descriptor: the first round's lane ranges and the candidates
        int2 lx = lane_lx[d.lane_lo + (lane < d.lane_n ? lane : 0)];
        const int cyv = cand_y[d.c0 + (lane < n ? lane : 0)];
        dp_sync<64>();                                  // the previous problem's DP is done with the wave's tables
        if (lane < n) L.cy[lane] = cyv;
        int hi_q[C::kPSlots], lo_q[C::kPSlots];
        unsigned amb[C::kPSlots], outc[C::kTSlots];
#pragma unroll
        for (int s = 0; s < C::kPSlots; ++s) {
            const int q = s * 64 + lane;
            hi_q[s] = 0x7fffffff; lo_q[s] = -1; amb[s] = 0;
            const unsigned short ij = g_pair_ij[q < npairs ? q : 0];
            const int ci = __shfl(cyv, ij & 255), cjv = __shfl(cyv, ij >> 8);
            if (q < npairs) label_thresholds_tab((i64)cjv - ci + 1, thr_tab, h_table, h_len, tau, &hi_q[s], &lo_q[s]);   // :490-495 as integer bounds
        }
#pragma unroll
        for (int s = 0; s < C::kTSlots; ++s) outc[s] = 0;
        // the candidates' genomic positions, wave-uniform
        const int cp0 = d.g0 + __builtin_amdgcn_readlane(cyv, 0), c_last = d.g0 + __builtin_amdgcn_readlane(cyv, n - 1);
        int cj[NM];
#pragma unroll
        for (int j = 0; j < NM; ++j) cj[j] = j < n ? d.g0 + __builtin_amdgcn_readlane(cyv, j) : cp0;      // beyond the problem: an empty window
        for (int l0 = 0; l0 < d.lane_n;) {
            const bool in = l0 + lane < d.lane_n;
            if (l0 > 0) lx = lane_lx[d.lane_lo + l0 + (in ? lane : 0)];
            // ---- the round's reads: as many of the next 64 lanes as own at most kStageCap exons together (all 64, usually);
            //      their exons are the stream's piece [base, end of the last one's)
            const int base = uni(lx.x) & ~1;                                     // (16-byte units)
            const u64 fm = __ballot(in && lx.y - base <= kStageCap);
            const int m = ~fm == 0 ? 64 : (int)__builtin_ctzll(~fm);            // the ranges ascend: a prefix of the lanes
            if (m == 0) { if (lane == 0) atomicOr(&st->err, kErrWaveStage); break; }      // (the host keeps such batches away: wave_on)
            const int total = __builtin_amdgcn_readlane(lx.y, m - 1) - base;
            {
                const int last2 = total & ~1;
                int4 sv[kStageCap / 128];
#pragma unroll
                for (int u = 0; u < kStageCap / 128; ++u) {
                    const int i = 2 * lane + 128 * u;
                    sv[u] = *reinterpret_cast<const int4 *>(lex + base + (i < last2 ? i : last2));
                }
#pragma unroll
                for (int u = 0; u < kStageCap / 128; ++u) {
                    const int i = 2 * lane + 128 * u;
                    if (i < total) *reinterpret_cast<int4 *>(&L.stage[i]) = sv[u];
                }
            }
            dp_sync<64>();
            const bool valid = lane < m;
            const u64 vmask = m == 64 ? ~0ULL : ((1ULL << m) - 1ULL);
            // ---- this lane's read: its exons that meet the window (consecutive: the first with te >= cand_0 up to the last with
            //      ts < cand_{n-1}), then its window coverage cov[j] = positions of its closed exons in [cand_0, cand_j)
            //      = sum over those exons of |[ts, te] n [cand_0, cand_j)|
            const int ea = valid ? lx.x - base : 0, eb = valid ? lx.y - base : 0;
            int first = ea, cnt = 0;
            for (int e = ea; e < eb; e += 4) {
                int2 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = L.stage[e + u];                 // (beyond the read: masked; the array has room)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool hit = e + u < eb && x[u].y >= cp0 && x[u].x < c_last;
                    if (hit && cnt == 0) first = e + u;
                    cnt += hit;
                }
                if (x[3].x >= c_last) break;                                       // the rest of the read lies beyond the window
            }
            int cov[NM];
#pragma unroll
            for (int j = 0; j < NM; ++j) cov[j] = 0;
            for (int e = 0; e < cnt; e += 2) {
                const int2 xa = L.stage[first + e], xb = L.stage[first + e + 1];
                const int a0 = max(xa.x, cp0), b0 = xa.y + 1;                      // closed exon -> half-open end
                const int a1 = max(xb.x, cp0), b1 = e + 1 < cnt ? xb.y + 1 : a1;   // (an odd count: the second slot is empty)
#pragma unroll
                for (int j = 1; j < NM; ++j)
                    if (j < n) cov[j] += max(0, min(b0, cj[j]) - a0) + max(0, min(b1, cj[j]) - a1);
            }
            // ---- pair labels: one compare per plane, the reads' bits arrive as the ballot; lane q keeps pair q's planes
            unsigned yl[C::kPSlots], yh[C::kPSlots], zl[C::kPSlots], zh[C::kPSlots];
#pragma unroll
            for (int s = 0; s < C::kPSlots; ++s) { yl[s] = 0; yh[s] = 0; zl[s] = 0; zh[s] = 0; }
            static_for<1, NM>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < n) {                                                       // (wave-uniform)
                    static_for<0, j>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        constexpr int q = j * (j - 1) / 2 + i, s = q >> 6, ql = q & 63;
                        const int dd = cov[j] - cov[i];
                        const int hi = __builtin_amdgcn_readlane(hi_q[s], ql), lo = __builtin_amdgcn_readlane(lo_q[s], ql);
                        const u64 y = __ballot(dd >= hi) & vmask, z = __ballot(dd <= lo) & vmask;
                        write_lane<ql>(yl[s], (unsigned)y); write_lane<ql>(yh[s], (unsigned)(y >> 32));
                        write_lane<ql>(zl[s], (unsigned)z); write_lane<ql>(zh[s], (unsigned)(z >> 32));
                    });
                }
            });
#pragma unroll
            for (int s = 0; s < C::kPSlots; ++s) {
                const int q = s * 64 + lane;
                if (q < npairs) {
                    amb[s] += (unsigned)(__popc(~(yl[s] | zl[s]) & (unsigned)vmask) + __popc(~(yh[s] | zh[s]) & (unsigned)(vmask >> 32)));   // neither label (:500-506)
                    L.planes[q] = make_uint4(yl[s], yh[s], zl[s], zh[s]);
                }
            }
            dp_sync<64>();
            // ---- out(i,j,k) (:509-528): the two labels exclude each other
#pragma unroll
            for (int s = 0; s < C::kTSlots; ++s) {
                const int tt = s * 64 + lane;
                if (tt < ntri) {
                    const unsigned tq = tri_q[tt];
                    const uint4 a = L.planes[tq & 255], b = L.planes[tq >> 8];
                    outc[s] += (unsigned)(__popc((a.x & b.z) | (a.z & b.x)) + __popc((a.y & b.w) | (a.w & b.y)));    // (disjoint: lo < hi)
                }
            }
            dp_sync<64>();
            l0 += m;
        }
        // a read outside the lane range has no coverage in the window: ambiguous exactly where lo < 0 (h >= 1)
#pragma unroll
        for (int s = 0; s < C::kPSlots; ++s) {
            const int q = s * 64 + lane;
            if (q < npairs) L.in[q] = -(int)((i64)amb[s] + (lo_q[s] < 0 ? (i64)d.outside : 0));
        }
#pragma unroll
        for (int s = 0; s < C::kTSlots; ++s) {
            const int tt = s * 64 + lane;
            if (tt < ntri) L.out[tt] = outc[s];
        }
        dp_sync<64>();
        const int chain = dp_solve_push<64, NM>(n, L.out, L.in, L.M, L.A, L.cy, support, chosen + d.c0 FSEG_DARG);
        if (lane == 0) pr.chain[p] = chain;
#ifdef FSEG_SCORE_TIMING
        if (lane == 0) FSEG_PROB_TICK(p, t_prob0, d.lane_n, d.lane_n);
#endif
    }
}


// ---------------------------------------------------------------------------------------------
// S5 whole, for problems that see few reads (at most kFuseLanes -- in batches of many partitions that is every problem: a
// DP window of a 500-read partition overlaps some 40 .. 130 of them): ONE WORKGROUP takes a problem from its candidates to
// its chosen breakpoints without leaving LDS --
//   pair thresholds (:490-495 as integer bounds) into registers, a thread keeps the pairs q = tid, tid + T, ..;
//   per 64 reads: window coverage (get_cumulative_coverage :188-246) by (read, candidate range) threads -- every read's
//     exon walk is cut into T/64 pieces that run side by side --, pair planes and ambiguity counts (:488-506), triple
//     counts (:509-528) into a table of CntT counters (8 bit when no problem of the launch sees more than 255 reads);
//   then the planes' LDS becomes M / in / A and dp_solve_push (:532-566, :592-594) runs on the count table where it lies.
// Nothing of such a problem exists in global memory between its descriptor and its chosen flags: no coverage tiles, no
// threshold / ambiguity / count arenas, no work items, no DP list entry.  (The arena path remains for problems that see
// thousands of reads, where one problem has to be spread over many workgroups.)
// ---------------------------------------------------------------------------------------------
// A large-class workgroup wants eight wave slots and 57-78 KB of LDS at once.  Beside kernels of small workgroups on other
// streams it is placed last, whatever the launch order (the dispatcher places what fits), and then the stage ends with the
// large class running alone on a mostly empty chip.  k_gate is what the side stream runs first: one wave that waits until the
// large class's workgroups have all started (they all fit the chip at once) or `max_ticks` of the 100 MHz clock have passed --
// an exit every launch reaches -- so the small classes fill the space the large one leaves instead of taking it first.
__global__ void __launch_bounds__(64) k_gate(Status *st, int which, unsigned grid, unsigned max_ticks) {
    // grid: the large class's workgroups the plan has launched (8-bit instance, and for the start gate the 16-bit one's too), at
    // most as many as fit the chip at once
    const unsigned want = grid;
    const unsigned long long t0 = wall_clock64();
    const unsigned *ctr = which == 2 ? &st->gate_wide : &st->gate;      // (2: the 16-bit instance's workgroups)
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && wall_clock64() - t0 < max_ticks && !stage_aborted(st))
        __builtin_amdgcn_s_sleep(16);
}

__global__ void __launch_bounds__(64) k_wait_word(Status *st, const unsigned *words, unsigned mask, unsigned gen, unsigned max_ticks) {
    // mask: which of words[0..31] must have reached the generation (lane i looks at word i)
    const int lane = lane_id();
    const bool mine = lane < 32 && ((mask >> lane) & 1u);
    const unsigned *w = words + (mine ? lane : 0);
    const unsigned long long t0 = wall_clock64();
    bool ok = false;
    for (;;) {
        ok = !mine || gen_reached(__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT), gen);
        if (__all(ok) || wall_clock64() - t0 >= max_ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (!__all(ok) && lane == 0) {
        __hip_atomic_store(&st->sync_abort, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        atomicOr(&st->err, kErrSyncTimeout);
    }
}

__global__ void __launch_bounds__(64) k_signal(unsigned *word, unsigned gen) {
    if (threadIdx.x == 0) __hip_atomic_store(word, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// The probe of a side stream's hardware queue (probe_side_queues, freddie_seg.hip): one wave that waits for a word a k_signal on the
// MAIN stream publishes behind it.  If it sees the word, the main stream runs beside this stream (result 1); if its time runs
// out, the main stream's packets sit BEHIND this wave in the same hardware queue (result 2) -- a k_wait_word here would wait for
// its own limit every time.
__global__ void __launch_bounds__(64) k_probe_wait(const unsigned *word, unsigned gen, unsigned *result, unsigned max_ticks) {
    const unsigned long long t0 = wall_clock64();
    bool ok = false;
    for (;;) {
        ok = gen_reached(__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT), gen);
        if (ok || wall_clock64() - t0 >= max_ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (threadIdx.x == 0) __hip_atomic_store(result, ok ? 1u : 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// (k_solve: seg_solve.h, instantiated by seg_solve16 / 32 / 60 .hip)

// T > 64 (round 6): the same DP by T threads (dp_solve_push: a column is owners' write, workgroup barrier, a pair or two per thread).
// The large class's launch takes T = 512: its problems' columns are up to 28 slots of a single wave, and eight waves bring the launch
// from 38 to 27 us alone, the stage from 0.136 to 0.133 ms on config3 and 0.121 to 0.118 on config5, whose closing chain it is (config4:
// 0.123 to 0.122).  The mid class stays with one wave: four take its launch from 20.4 to 17.8 us alone and the stage nowhere
// (tools/probes/dp_waves.sh; two and four waves for the large class: 0.125 / 0.122 against 0.122 / 0.122 with one / eight on config4).
template <int NM, typename OutT, typename V, int T>
__global__ void __launch_bounds__(T) k_dpw(Status *st, int nm, i64 list_base, i64 list_n, ProblemArrays pr, const ProbDesc *desc,
                                            const unsigned char *dpx, i64 dpx_stride,
                                            int support, unsigned char *chosen, const int *__restrict__ wide_items FSEG_TPARAM) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = T == 64 ? lane_id() : (int)threadIdx.x;
    if ((i64)blockIdx.x >= list_n) return;
    const unsigned aborted = stage_aborted(st);
    if (wide_items && aborted) return;                  // (the list of wide problems is an index into the records: not followed blindly)
    const i64 t = wide_items ? (i64)uni(wide_items[list_base + blockIdx.x]) : (i64)blockIdx.x;
#ifdef FSEG_SCORE_TIMING
    const unsigned long long t_dp0 = wall_clock64();
#endif
    const unsigned char *slot = dpx + t * dpx_stride;
    const ProbDesc d = FSEG_LOAD_DESC(desc + list_base + t);
    const int n = d.n;
    if (aborted) return;
    if (n > nm || n > NM || n < 3 || d.lane_n > kFuseLanesWide) return;         // (k_solve has raised the error)
    if ((d.kind == kKindFusedWide) != (sizeof(OutT) != 1)) return;               // whose problem (k_prob_range)
    const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
    const int rt_pairs = nm * (nm - 1) / 2;
    V *M = reinterpret_cast<V *>(smem);
    int *in_s = reinterpret_cast<int *>(smem + (((size_t)rt_pairs * sizeof(V) + 15) & ~(size_t)15));
    unsigned char *A = reinterpret_cast<unsigned char *>(in_s) + (((size_t)rt_pairs * 4 + 15) & ~(size_t)15);
    OutT *out_s = reinterpret_cast<OutT *>(A + (((size_t)rt_pairs + 15) & ~(size_t)15));
    {
        const int *g_in = reinterpret_cast<const int *>(slot + kDpxHeader);
        for (int q = lane; q < npairs; q += T) { in_s[q] = g_in[q]; A[q] = (unsigned char)(g_pair_ij[q] >> 8); }
        const uint4 *g_out = reinterpret_cast<const uint4 *>(slot + kDpxHeader + dpx_in_bytes(nm));
        uint4 *l_out = reinterpret_cast<uint4 *>(out_s);
        const int n16 = (ntri * (int)sizeof(OutT) + 15) / 16;
        constexpr int kInFlight = T == 64 ? 8 : 2;                               // 16-byte loads per thread in flight
        for (int x0 = 0; x0 < n16; x0 += T * kInFlight) {
            uint4 v[kInFlight];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) { const int x = x0 + u * T + lane; v[u] = g_out[x < n16 ? x : 0]; }
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) { const int x = x0 + u * T + lane; if (x < n16) l_out[x] = v[u]; }
        }
    }
    dp_sync<T>();
#ifdef FSEG_SCORE_TIMING
    __shared__ unsigned long long tick_sink[16];
    unsigned long long *dp_tacc = tick_sink; unsigned long long dt_prev = 0;
#endif
    // (s_setprio 3 for this wave -- a chain of dependent instructions that its class's chain ends with -- made the stage slower:
    // config3 0.210 -> 0.245-0.270 ms, config4 0.147 -> 0.151)
    int chain;
    if constexpr (T == 64) chain = dp_solve_wave<NM>(n, out_s, in_s, M, A, support, chosen + d.c0 FSEG_DARG);
    else chain = dp_solve_push<T, NM, OutT, V, true>(n, out_s, in_s, M, A, nullptr, support, chosen + d.c0 FSEG_DARG);
    if (lane == 0) pr.chain[d.w0] = chain;
#ifdef FSEG_SCORE_TIMING
    if (lane == 0 && (size_t)d.w0 < kTaccProbs) { unsigned long long *r_ = tacc + 16 + 4 * kTaccProbs + 4 * (size_t)d.w0; r_[0] = wall_clock64() - t_dp0; r_[1] = (unsigned long long)hw_where() << 32; r_[3] = t_dp0; }
#endif
}


// the instances the host launches (freddie_seg.hip sees the declarations only: taking an instance's address here is what
// instantiates it -- host stub and device code -- in this translation unit)
__attribute__((used)) static const void *const kInstances[] = {
    reinterpret_cast<const void *>(&k_wave<kTiny, int>),
    reinterpret_cast<const void *>(&k_wave<kTiny, i64>),
    reinterpret_cast<const void *>(&k_dpw<kClsSmall, unsigned char, int, 64>),
    reinterpret_cast<const void *>(&k_dpw<kClsSmall, unsigned char, i64, 64>),
    reinterpret_cast<const void *>(&k_dpw<kClsSmall, unsigned short, int, 64>),
    reinterpret_cast<const void *>(&k_dpw<kClsSmall, unsigned short, i64, 64>),
    reinterpret_cast<const void *>(&k_dpw<kClsMid, unsigned char, int, 64>),
    reinterpret_cast<const void *>(&k_dpw<kClsMid, unsigned char, i64, 64>),
    reinterpret_cast<const void *>(&k_dpw<kClsMid, unsigned short, int, 64>),
    reinterpret_cast<const void *>(&k_dpw<kClsMid, unsigned short, i64, 64>),
    reinterpret_cast<const void *>(&k_dpw<kNMax, unsigned char, int, 64>),
    reinterpret_cast<const void *>(&k_dpw<kNMax, unsigned char, i64, 64>),
    reinterpret_cast<const void *>(&k_dpw<kNMax, unsigned short, int, 64>),
    reinterpret_cast<const void *>(&k_dpw<kNMax, unsigned short, i64, 64>),
    reinterpret_cast<const void *>(&k_dpw<kNMax, unsigned char, int, 512>),
    reinterpret_cast<const void *>(&k_dpw<kNMax, unsigned char, i64, 512>),
    reinterpret_cast<const void *>(&k_dpw<kNMax, unsigned short, int, 512>),
    reinterpret_cast<const void *>(&k_dpw<kNMax, unsigned short, i64, 512>),
};

}  // namespace fseg
