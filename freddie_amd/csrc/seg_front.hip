// seg_front.hip -- S1 histogram, S2 Gaussian smoothing (+ flags, block prefixes), the compaction scans, S3a variance threshold (chunk
// kernels and the one-workgroup-per-partition form), S3b what the smoothing tiles could not decide about the peaks.
// Part of libfreddie_seg.so (gfx950); shared definitions: seg_common.h, declarations: seg_kernels.h, launches: freddie_seg.hip.
#include "seg_kernels.h"

#include <type_traits>

namespace fseg {

__global__ void __launch_bounds__(256) k_thr_table(const double *h_table, int h_len, double tau, int2 *tab) {
    for (int L = blockIdx.x * blockDim.x + threadIdx.x; L < kThrTab; L += gridDim.x * blockDim.x) {
        int hi = 0x7fffffff, lo = -1;
        if (L >= 1) label_thresholds((i64)L, h_table, h_len, tau, &hi, &lo);
        tab[L] = make_int2(hi, lo);
    }
}

__global__ void __launch_bounds__(512) k_hist(int n_chunks, const int *chunk_part, const i64 *chunk_p0, const int *chunk_n,
                                              const int *chunk_glo, const int *chunk_ghi, const i64 *chunk_lane_lo,
                                              const i64 *chunk_lane_hi, const i64 *part_iv_off,
                                              const int *iv_start, const int *iv_end, const i64 *pos_off,
                                              const i64 *part_lane_off, const int2 *__restrict__ lane_lx, const int *lane_start,
                                              const int *lane_pmax, const int2 *__restrict__ lex,
                                              int ignore_ends, int *y_raw, Status *st, u64 *zero_ptr, i64 zero_n) {
    __shared__ int hist[kHistChunk];
    __shared__ int ivs_s[kHistIv], ive_s[kHistIv], base_s[kHistIv];
    // first kernel of the run: also clears the look-back words of the three compactions (saves a memset node)
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += (i64)gridDim.x * blockDim.x) zero_ptr[i] = 0;
    for (int ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const int part = chunk_part[ch];
        const i64 p0 = chunk_p0[ch];
        const int np = chunk_n[ch];
        const int g_lo = chunk_glo[ch], g_hi = chunk_ghi[ch];          // genomic position of the first / last position
        const i64 k0 = part_iv_off[part], k1 = part_iv_off[part + 1];
        __syncthreads();
        for (int i = threadIdx.x; i < np; i += blockDim.x) hist[i] = 0;
        // lanes whose [first, last] position range meets [g_lo, g_hi]: found on upload (the chunks and the sorted
        // lanes are both fixed then), two dependent 16-step searches less per workgroup
        const i64 lo = chunk_lane_lo[ch], hi = chunk_lane_hi[ch];
        // the partition's interval table in LDS when it fits (the per-exon interval search then stays on chip)
        const int nk = (int)(k1 - k0);
        const bool cached = nk <= kHistIv;
        if (cached)
            for (int i = threadIdx.x; i < nk; i += blockDim.x) {
                ivs_s[i] = iv_start[k0 + i]; ive_s[i] = iv_end[k0 + i];
                base_s[i] = (int)(pos_off[k0 + i] - p0) - iv_start[k0 + i];   // chunk-local index = base + genomic position
            }
        __syncthreads();
        // 8 threads share a read: thread q of the group takes the read's exons q, q+8, ...  The walk is a chain of
        // dependent loads (lane -> exon range -> exon), so four reads per group are in flight: their exon ranges, then
        // their first exons, are loaded together from clamped addresses before any of them is used.  The exons come from
        // the lane-ordered (ts, te) stream: a group's eight threads read 64 consecutive bytes, consecutive groups consecutive
        // lanes' pieces (from the rep-ordered ex_ts / ex_te: two lines per read, anywhere)
        const int sub = threadIdx.x & 7;
        const int G8 = blockDim.x >> 3;
        auto count_exon = [&](i64 e, i64 e0, i64 e1, int ts, int te) {
            if (te < g_lo || ts > g_hi) return;
            // the interval that holds ts must hold te as well (:666-668; also validated on upload)
            int kl = 0;
            bool ok;
            int base;
            if (cached) {
                int a2 = 0, b2 = nk;
                while (b2 - a2 > 1) { int m = (a2 + b2) >> 1; if (ivs_s[m] <= ts) a2 = m; else b2 = m; }
                kl = a2;
                ok = ts >= ivs_s[kl] && ts <= ive_s[kl] && te <= ive_s[kl];
                base = base_s[kl];
            } else {
                ok = ts >= iv_start[k0];
                i64 k = k0;
                if (ok) { k = k0 + last_le(iv_start + k0, k1 - k0, ts); ok = ts <= iv_end[k] && te <= iv_end[k]; }
                base = (int)(pos_off[k] - p0) - iv_start[k];
            }
            if (!ok) { atomicOr(&st->err, kErrExonInterval); return; }
            if (!(ignore_ends && e == e0) && ts >= g_lo && ts <= g_hi) atomicAdd(&hist[base + ts], 1);       // :670-671
            if (!(ignore_ends && e == e1 - 1) && te >= g_lo && te <= g_hi) atomicAdd(&hist[base + te], 1);   // :672-673
        };
        for (i64 l0 = lo + (threadIdx.x >> 3); l0 < hi; l0 += 4 * (i64)G8) {
            int2 ex[4], x0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const i64 l = l0 + (i64)u * G8; ex[u] = lane_lx[l < hi ? l : l0]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = ex[u].x + sub < ex[u].y ? ex[u].x + sub : ex[u].x;       // a valid exon of the read (a read has at least one)
                x0[u] = lex[e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (l0 + (i64)u * G8 >= hi) break;
                const i64 e0 = ex[u].x, e1 = ex[u].y;
                if (e0 + sub < e1) count_exon(e0 + sub, e0, e1, x0[u].x, x0[u].y);
                for (i64 e = e0 + sub + 8; e < e1; e += 8) { const int2 x = lex[e]; count_exon(e, e0, e1, x.x, x.y); }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < np; i += blockDim.x) y_raw[p0 + i] = hist[i];
    }
}

template <int R>
#ifndef FSEG_SMOOTH_OCC
#define FSEG_SMOOTH_OCC 6
#endif
__global__ void __launch_bounds__(kSmoothThreads, FSEG_SMOOTH_OCC) k_smooth(int n_tiles, const TileDesc *__restrict__ tiles,
                                                const int *__restrict__ y_raw, const double *__restrict__ w_g, int radius_rt,
                                                double *y_out, unsigned *flag_pos, unsigned *flag_cand, int *blk_pre, int *tile_tot,
                                                int *tile_defer) {
    __shared__ int xs[kSmoothTile + 2 * kMaxRadius];
    __shared__ unsigned tw_c[kSmoothTile / 32 + 1], tw_p[kSmoothTile / 32 + 1];   // the tile's candidate / Y > 0 flags, a bit per position
    __shared__ int defer_s;
    __shared__ int blk_s[kSmoothTile / kSumBlock];
    __shared__ double ws[kMaxRadius + 1];
    // the tile's smoothed values: what the candidate test reads of its neighbours (the results themselves leave from
    // registers: a thread's four consecutive positions are 32 / 16 / 4 contiguous bytes of the output arrays, a wave's 256
    // positions one contiguous run per store instruction -- as five arrays of single elements per lane the kernel took
    // 183 us per 250 k-read batch with or without its arithmetic)
    __shared__ double ys[kSmoothTile];
    const int radius = R > 0 ? R : radius_rt;
    const int span = kSmoothTile + 2 * radius;
    constexpr int kStage = R > 0 ? (kSmoothTile + 2 * R + kSmoothThreads - 1) / kSmoothThreads : kSmoothStage;
    for (int j = threadIdx.x; j <= radius; j += blockDim.x) ws[j] = w_g[j];
    // A tile is a short chain of dependent loads (record -> counts) followed by barriers, and a workgroup walks several
    // tiles: the chain of the NEXT tile is issued before the arithmetic of the current one (counts into registers,
    // the record one tile further ahead), so its latency hides behind the taps and the stores.
    const int G = gridDim.x;
    auto load_counts = [&](const TileDesc &d, int *v) {
        // 'reflect': (d c b a | a b c d | d c b a).  One reflection does unless the interval is shorter than the radius;
        // only then the general index (a 64-bit modulo) is evaluated.
        // (the record is the same in every lane: in scalar registers the tests below are branches of the whole wave)
        const int len_d = __builtin_amdgcn_readfirstlane(d.len), y0_d = __builtin_amdgcn_readfirstlane(d.y0);
        const int yb = y0_d - radius + (int)threadIdx.x;
        const int *src = y_raw + (((i64)__builtin_amdgcn_readfirstlane((int)(d.base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)d.base));
        if (R > 0 && y0_d + kSmoothTile + radius <= len_d) {
            // nothing of the window lies beyond the interval's end (two tiles in three): only the first radius positions of the
            // interval's first tile reflect, and they are all in the first staged element (R <= kSmoothThreads)
            static_assert(R <= kSmoothThreads, "a tile's left halo lies in the first staged element of each thread");
#pragma unroll
            for (int e = 0; e < kStage; ++e) {
                const int idx = e * kSmoothThreads + threadIdx.x;
                const int y = yb + e * kSmoothThreads;
                const int r = e == 0 && y < 0 ? -1 - y : y;
                v[e] = (e + 1) * kSmoothThreads <= kSmoothTile + 2 * R || idx < span ? src[r] : 0;
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < kStage; ++e) {
            const int idx = e * kSmoothThreads + threadIdx.x;
            const int y = yb + e * kSmoothThreads;
            int r = y < 0 ? -1 - y : (y >= len_d ? 2 * len_d - 1 - y : y);
            if ((unsigned)r >= (unsigned)len_d) r = (int)reflect_index((i64)y, (i64)len_d);
            v[e] = idx < span ? src[r] : 0;
        }
    };
    int t = blockIdx.x;
    TileDesc d_cur = {0, 0, 1}, d_next = {0, 0, 1};
    int v_cur[kStage];
    if (t < n_tiles) { d_cur = tiles[t]; load_counts(d_cur, v_cur); }
    if (t + G < n_tiles) d_next = tiles[t + G];
    for (; t < n_tiles; t += G) {
        const int y0 = d_cur.y0, len = d_cur.len;                    // (positions inside one interval: 32 bits)
        const i64 base = d_cur.base;
        lds_barrier();
#pragma unroll
        for (int e = 0; e < kStage; ++e) { const int idx = e * kSmoothThreads + threadIdx.x; if (idx < span) xs[idx] = v_cur[e]; }
        TileDesc d_n2 = {0, 0, 1};
        if (t + 2 * G < n_tiles) d_n2 = tiles[t + 2 * G];
        if (t + G < n_tiles) load_counts(d_next, v_cur);             // in flight during this tile's work
        lds_barrier();
        {   // the histogram's sums over blocks of kSumBlock positions of the tile (four threads' positions each; what lies
            // beyond the interval counts nothing): with their exclusive prefix inside the tile and the tile's total they let
            // k_segments answer refine_segmentation's `sum(i_vals) < 20` test (:258) exactly with two look-ups and at most
            // 31 positions of the histogram itself.  (Round 3 kept an inclusive prefix PER POSITION: a workgroup scan per tile
            // and 114 MB written per batch -- 25 of this kernel's 158 us.)
            const int o4 = threadIdx.x * 4;
            int run = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) run += (y0 + o4 + e < len) ? xs[radius + o4 + e] : 0;
            run += __builtin_amdgcn_update_dpp(0, run, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]: + the neighbour's
            run += __builtin_amdgcn_update_dpp(0, run, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]: + the other pair's
            if ((threadIdx.x & 3) == 0) blk_s[threadIdx.x >> 2] = run;
        }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;    // this thread's four outputs (kept for the candidate test below)
        {   // every thread computes 4 consecutive outputs; the two 4-wide input windows of tap j slide by one
            // position per tap, so each tap costs two LDS reads for four outputs
            const int o4 = threadIdx.x * 4;
            if (y0 + o4 < len) {
                const int c = o4 + radius;
                const double w0 = ws[0];
                a0 = __dmul_rn((double)(xs[c]), w0); a1 = __dmul_rn((double)(xs[c + 1]), w0);
                a2 = __dmul_rn((double)(xs[c + 2]), w0); a3 = __dmul_rn((double)(xs[c + 3]), w0);
                // R > 0: the windows hold the counts as doubles: a count is converted once, when it enters a window, and a pair's sum is
                // an fp64 addition of two integers (exact: the same value as the sum converted) -- 3.5 instead of 4 instructions
                // per output and tap.  (Any radius: integer windows, the sum converted: the doubles would spill there.)
                typedef typename std::conditional<(R > 0), double, int>::type Win;
                auto pair_sum = [](Win l, Win r) -> double { if constexpr (R > 0) return __dadd_rn(l, r); else return (double)(l + r); };
                Win l0 = (Win)xs[c - radius], l1 = (Win)xs[c - radius + 1], l2 = (Win)xs[c - radius + 2], l3 = (Win)xs[c - radius + 3];
                Win r0 = (Win)xs[c + radius], r1 = (Win)xs[c + radius + 1], r2 = (Win)xs[c + radius + 2], r3 = (Win)xs[c + radius + 3];
#define FSEG_TAP(W)                                                                                        \
                    a0 = __dadd_rn(a0, __dmul_rn(pair_sum(l0, r0), (W)));                                          \
                    a1 = __dadd_rn(a1, __dmul_rn(pair_sum(l1, r1), (W)));                                          \
                    a2 = __dadd_rn(a2, __dmul_rn(pair_sum(l2, r2), (W)));                                          \
                    a3 = __dadd_rn(a3, __dmul_rn(pair_sum(l3, r3), (W)));                                          \
                    l0 = l1; l1 = l2; l2 = l3; l3 = (Win)xs[c - j + 4];     /* left window moves right */          \
                    r3 = r2; r2 = r1; r1 = r0; r0 = (Win)xs[c + j - 1];     /* right window moves left */
                if (R > 0) {
#pragma unroll
                    for (int j = R; j >= 1; --j) { FSEG_TAP(w_g[j]) }
                } else {
                    for (int j = radius; j >= 1; --j) { FSEG_TAP(ws[j]) }
                }
#undef FSEG_TAP
                ys[o4] = a0; ys[o4 + 1] = a1; ys[o4 + 2] = a2; ys[o4 + 3] = a3;
            }
        }
        if (threadIdx.x == 0) defer_s = -1;
        lds_barrier();
        if (threadIdx.x < kSmoothTile / kSumBlock) {                  // (one wave: the tile's block sums -> exclusive prefixes, total)
            const int v = blk_s[threadIdx.x];
            int x = v;
#pragma unroll
            for (int dd = 1; dd < kSmoothTile / kSumBlock; dd <<= 1) { const int y = __shfl_up(x, dd); if ((int)threadIdx.x >= dd) x += y; }
            blk_pre[(i64)t * (kSmoothTile / kSumBlock) + threadIdx.x] = x - v;
            if (threadIdx.x == kSmoothTile / kSumBlock - 1) tile_tot[t] = x;
        }
        // S3b candidates (candidates_from_peaks :615-621 = scipy's _local_maxima_1d + the interval's first and last position),
        // decided here while the tile's smoothed values are at hand -- a pass of its own over the signal read all of it back
        // from HBM.  A strict maximum, or the midpoint of a plateau that rises on its left and falls on its right
        // ((first + last) / 2), counts.  A thread tests its own four outputs (registers; its two outer neighbours from LDS) and
        // writes their flags as one word.  What this tile cannot see is left to k_peaks_edges: whether its first and its last
        // position start a peak (their outer neighbours belong to other tiles) and the one plateau that may run into the
        // tile's last position (its start goes to tile_defer).
        int mid0 = -1, mid1 = -1;        // plateau midpoints found by this thread (four consecutive positions hold at most two
                                         // plateau peaks: rise, level, fall, rise, level)
        unsigned word = 0;               // this thread's four candidate flags, a byte each
#ifndef FSEG_SM_NOCAND
        {
            const int o4 = threadIdx.x * 4;
            const double v[6] = {ys[o4 > 0 ? o4 - 1 : 0], a0, a1, a2, a3, ys[o4 + 4 < kSmoothTile ? o4 + 4 : kSmoothTile - 1]};
            auto plateau = [&](int i, double a) {                        // position i rises to a and the next one is level with it
                int ia = i + 1;                                          // (scipy: extend while ia < len - 1 and y[ia] == y[i])
                while (ia < kSmoothTile - 1 && y0 + ia < len - 1 && ys[ia] == a) ++ia;
                if (ys[ia] == a && y0 + ia < len - 1) defer_s = i;       // still level at the tile's last position: not decidable here
                                                                         // (at most one run of equal values reaches the tile's end)
                else if (ys[ia] < a) { if (mid0 < 0) mid0 = (i + ia - 1) >> 1; else mid1 = (i + ia - 1) >> 1; }
            };
            // the wave's 256 positions lie strictly inside the interval (four waves in six of an average interval): no position is an
            // end of the interval or beyond it, and the tile's own first / last position need no test -- the first sees itself as its
            // left neighbour (no rise), the last itself as its right one (level: excluded below)
            const int w0 = __builtin_amdgcn_readfirstlane(y0) + (int)(threadIdx.x & ~63u) * 4;
            if (w0 >= 1 && w0 + 255 <= __builtin_amdgcn_readfirstlane(len) - 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double a = v[e + 1];
                    if (v[e] < a) {
                        if (v[e + 2] < a) word |= 1u << (8 * e);
                        else if (v[e + 2] == a && o4 + e != kSmoothTile - 1) plateau(o4 + e, a);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = o4 + e;
                    const int pos = y0 + i;
                    if (pos >= len) break;
                    if (pos == 0 || pos == len - 1) { word |= 1u << (8 * e); continue; }
                    if (i == 0 || i == kSmoothTile - 1) continue;        // k_peaks_edges
                    const double a = v[e + 1];
                    if (v[e] < a) {
                        if (v[e + 2] < a) word |= 1u << (8 * e);
                        else if (v[e + 2] == a) plateau(i, a);
                    }
                }
            }
        }
#else
        word = a1 > a2 ? 1u : 0u;      // diagnostic build (wrong results): no candidate test
#endif
#ifndef FSEG_SM_NOPACK
        // The tile's flags leave as bits of the batch-wide masks (cleared before this kernel).  A thread's four candidate flags and
        // its four Y > 0 flags are a nibble each; eight neighbouring lanes OR their nibbles into the word of their 32 positions
        // (three DPP steps) and its first lane leaves it in LDS -- sixteen words a tile and mask -- where seventeen lanes shift them to
        // where the tile starts in the batch, below.  (Round 4 kept a flag byte per position in LDS and had those seventeen lanes
        // gather thirty-two bytes each: 11 of the kernel's 128 us, with the other threads waiting at the barrier.)  A plateau's
        // midpoint is another thread's position: its bit goes to the mask directly.
        {
            const int o4 = threadIdx.x * 4, left = len - (y0 + o4);
            const unsigned nib_c = __builtin_amdgcn_udot4(word, 0x08040201u, 0u, false);
            const unsigned nib_p = (left > 0 && a0 > 0.0 ? 1u : 0u) | (left > 1 && a1 > 0.0 ? 2u : 0u) |
                                   (left > 2 && a2 > 0.0 ? 4u : 0u) | (left > 3 && a3 > 0.0 ? 8u : 0u);     // (what lies beyond the interval flags nothing)
            const int sh4 = 4 * (threadIdx.x & 7);
            unsigned vc = nib_c << sh4, vp = nib_p << sh4;
            vc |= __builtin_amdgcn_update_dpp(0, vc, 0xB1, 0xf, 0xf, false);  vp |= __builtin_amdgcn_update_dpp(0, vp, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
            vc |= __builtin_amdgcn_update_dpp(0, vc, 0x4E, 0xf, 0xf, false);  vp |= __builtin_amdgcn_update_dpp(0, vp, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
            vc |= __builtin_amdgcn_update_dpp(0, vc, 0x141, 0xf, 0xf, false); vp |= __builtin_amdgcn_update_dpp(0, vp, 0x141, 0xf, 0xf, false);   // row_half_mirror: the other quad of the eight
            if ((threadIdx.x & 7) == 0) { tw_c[threadIdx.x >> 3] = vc; tw_p[threadIdx.x >> 3] = vp; }
            const i64 p0 = base + y0;
            if (mid0 >= 0) atomicOr(&flag_cand[(p0 + mid0) >> 5], 1u << (int)((p0 + mid0) & 31));
            if (mid1 >= 0) atomicOr(&flag_cand[(p0 + mid1) >> 5], 1u << (int)((p0 + mid1) & 31));
        }
#endif
        lds_barrier();
        if (threadIdx.x == 0) tile_defer[t] = defer_s;
        {
            const int o4 = threadIdx.x * 4;
            const i64 p = base + y0 + o4;
            if (y0 + o4 + 3 < len) {                                 // the thread's four positions lie inside the interval
                typedef double double2u __attribute__((ext_vector_type(2), aligned(8)));
                double2u lo2, hi2; lo2.x = a0; lo2.y = a1; hi2.x = a2; hi2.y = a3;
                *reinterpret_cast<double2u *>(y_out + p) = lo2;
                *reinterpret_cast<double2u *>(y_out + p + 2) = hi2;
            } else {
                const double av[4] = {a0, a1, a2, a3};
                for (int e = 0; e < 4; ++e)
                    if (y0 + o4 + e < len) y_out[p + e] = av[e];
            }
        }
#ifndef FSEG_SM_NOPACK
        if (threadIdx.x <= kSmoothTile / 32) {
            // the tile starts at an arbitrary position of the batch -- bit s = (base + y0) & 31 of its first word --, so word j of the
            // masks is T[j] << s | T[j-1] >> (32 - s), seventeen of them, OR-ed in (the first and the last are shared with the
            // neighbouring tiles)
            const int j = threadIdx.x;
            const unsigned tc = j < kSmoothTile / 32 ? tw_c[j] : 0u, tp = j < kSmoothTile / 32 ? tw_p[j] : 0u;
            const unsigned pc = j > 0 ? tw_c[j - 1] : 0u, pp = j > 0 ? tw_p[j - 1] : 0u;
            const i64 p0 = base + y0;
            const int sh = (int)(p0 & 31);
            const unsigned gc = sh ? (tc << sh) | (pc >> (32 - sh)) : tc;
            const unsigned gp = sh ? (tp << sh) | (pp >> (32 - sh)) : tp;
            if (gc) atomicOr(&flag_cand[(p0 >> 5) + j], gc);
            if (gp) atomicOr(&flag_pos[(p0 >> 5) + j], gp);
        }
#endif
        d_cur = d_next; d_next = d_n2;
    }
}


// The start of a run: Status and the three flag planes (OR-ed into by k_smooth, k_peaks_edges, k_segments, k_refine) zeroed by ONE
// launch of 4 us (round 6; two hipMemsetAsync until then, which this runtime turns into three fill kernels of ~4 us each in front of
// k_hist: tools/probes/kernel_ab.sh).
static_assert(sizeof(Status) % 4 == 0, "k_clear zeroes Status word by word");
__global__ void __launch_bounds__(256) k_clear(Status *st, unsigned *bits, i64 n_words) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x, stride = (i64)gridDim.x * blockDim.x;
    uint4 *b4 = reinterpret_cast<uint4 *>(bits);                     // (the planes start an allocation: 256-byte aligned)
    const i64 n4 = n_words >> 2;
    for (i64 i = t; i < n4; i += stride) b4[i] = make_uint4(0u, 0u, 0u, 0u);
    for (i64 i = (n4 << 2) + t; i < n_words; i += stride) bits[i] = 0u;
    if (blockIdx.x == 0 && st) {                                     // (null: the two-bit label arena's clear, labels stage)
        unsigned *s32 = reinterpret_cast<unsigned *>(st);
        for (int i = threadIdx.x; i < (int)(sizeof(Status) / 4); i += blockDim.x) s32[i] = 0u;
    }
}

// many blocks: three passes (block sums, their scan by one workgroup, emission)
__global__ void __launch_bounds__(256) k_scan1(const unsigned *flags, i64 n, int *bsum) {
    __shared__ int lds[16];
    i64 nb = (n + kScanBlock - 1) / kScanBlock;
    for (i64 b = blockIdx.x; b < nb; b += gridDim.x) {
        i64 i0 = b * kScanBlock + (i64)threadIdx.x * 32;
        int s = i0 < n ? count_flags32(load_flags32(flags, i0, n)) : 0;
        int tot;
        wg_exclusive_scan<4>(s, lds, &tot);
        if (threadIdx.x == 0) bsum[b] = tot;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kScan2Threads) k_scan2(int *bsum, i64 nb, u64 *total_out, i64 *off_last /* may be null */) {
    __shared__ int lds[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (i64 b0 = 0; b0 < nb; b0 += (i64)kScan2Threads * kScan2Per) {
        const i64 base = b0 + (i64)threadIdx.x * kScan2Per;
        int v[kScan2Per], run = 0;
#pragma unroll
        for (int e = 0; e < kScan2Per; ++e) { v[e] = base + e < nb ? bsum[base + e] : 0; }
#pragma unroll
        for (int e = 0; e < kScan2Per; ++e) { const int x = v[e]; v[e] = run; run += x; }      // exclusive inside the thread
        int tot;
        const int ex = wg_exclusive_scan<kScan2Threads / 64>(run, lds, &tot);
        const int carry = carry_s;
#pragma unroll
        for (int e = 0; e < kScan2Per; ++e) if (base + e < nb) bsum[base + e] = carry + ex + v[e];
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { *total_out = (u64)carry_s; if (off_last) *off_last = (i64)carry_s; }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_scan_emit(const unsigned *flags, i64 n, const int *bsum /* or null */,
                                                   u64 *state, u64 *total_out, i64 *off_last /* may be null */,
                                                   unsigned *err, const double *y,
                                                   double *v, i64 K, const i64 *pos_off, const int *iv_start,
                                                   const int *blk_iv0, int *out_y, int *out_pos, i64 *out_off,
                                                   int force_stall /* tests: report a look-back stall */, int *out_iv /* may be null */) {
    if (force_stall && !bsum && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err, kErrScanStall);
    // A block is 4 waves x 2048 consecutive positions.  Each wave first counts its flags (16-byte loads), the wave
    // offsets come from LDS, then the wave walks its positions in rows of 64: ballot -> rank, so the loads of y and
    // the stores of the compacted output are coalesced.
    __shared__ int wave_cnt[4];
    __shared__ int lds[16];
    __shared__ i64 bcast;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const u64 lt_mask = (1ULL << lane) - 1ULL;
    i64 nb = (n + kScanBlock - 1) / kScanBlock;
    if (MODE == kEmitPositions) {
        // sparse flags (about one position in a hundred): every thread owns 32 consecutive positions and only the
        // threads that hold a flag do any work.  The intervals the block's positions lie in (first interval of this block ..
        // first interval of the next) are staged in LDS with one coalesced load: a flagged thread's interval search and its
        // three look-ups would otherwise be a chain of five or six dependent global loads, which is what this kernel ran at.
        constexpr int kIvStage = 768;
        __shared__ i64 po_s[kIvStage + 1];
        __shared__ int is_s[kIvStage];
        {
            const i64 b = blockIdx.x;                                // grid == nb
            const i64 ka0 = blk_iv0[b], kb0 = (i64)blk_iv0[b + 1] + 1;      // intervals [ka0, kb0) (kb0 <= K)
            const int niv = (int)(kb0 - ka0);
            const bool staged = niv <= kIvStage;
            if (staged) {
                for (int x = threadIdx.x; x <= niv; x += blockDim.x) po_s[x] = pos_off[ka0 + x];
                for (int x = threadIdx.x; x < niv; x += blockDim.x) is_s[x] = iv_start[ka0 + x];
            }
            i64 i0 = b * kScanBlock + (i64)threadIdx.x * 32;
            Flags32 f = 0;
            int s = 0;
            if (i0 < n) { f = load_flags32(flags, i0, n); s = count_flags32(f); }
            int tot;
            int ex = wg_exclusive_scan<4>(s, lds, &tot);             // (its barriers also publish the staged table)
            ex += bsum ? bsum[b] : (int)scan_lookback(state, b, nb, tot, &bcast, total_out, off_last, err);
            if (s) {
                i64 k = -1, k_end = 0, k_base = 0;
                int k_start = 0;
                {
                    unsigned w = f;
                    while (w) {
                        int e = __ffs(w) - 1;
                        w &= w - 1;
                        i64 i = i0 + e;
                        if (k < 0 || i >= k_end) {
                            // the interval of position i lies between the first intervals of this and the next block
                            const i64 ka = k < 0 ? ka0 : k + 1;
                            if (staged) {
                                const int a = (int)(ka - ka0);
                                const int kk = a + (int)last_le(po_s + a, (i64)(niv - a), i);
                                k = ka0 + kk; k_base = po_s[kk]; k_end = po_s[kk + 1]; k_start = is_s[kk];
                            } else {
                                k = ka + last_le(pos_off + ka, kb0 - ka, i);
                                k_base = pos_off[k]; k_end = pos_off[k + 1]; k_start = iv_start[k];
                            }
                        }
                        int yy = (int)(i - k_base);
                        out_y[ex] = yy;
                        if (out_pos) out_pos[ex] = k_start + yy;
                        if (out_iv) out_iv[ex] = (int)k;
                        if (yy == 0) out_off[k] = ex;
                        ++ex;
                    }
                }
            }
        }
        return;
    }
    {
        const i64 b = blockIdx.x;                                    // grid == nb
        const i64 w0 = b * kScanBlock + (i64)wave * 2048;            // first position of this wave
        int s = 0;
        unsigned fm = 0;                                             // bit j = flag of position w0 + lane * 32 + j
        {
            i64 i0 = w0 + (i64)lane * 32;
            if (i0 < n) {
                fm = load_flags32(flags, i0, n);
                s = count_flags32(fm);
            }
        }
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) wave_cnt[wave] = s;
        __syncthreads();
        const int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        int ex = bsum ? bsum[b] : (int)scan_lookback(state, b, nb, tot, &bcast, total_out, off_last, err);
        for (int w = 0; w < wave; ++w) ex += wave_cnt[w];
        if (s) {
            // The wave walks its 2048 positions in rows of 64 (lane = column), so the loads of y and the compacted
            // stores are coalesced.  Row q's 64 flags are the masks of lanes 2q and 2q+1 (two readlanes, no memory),
            // and the values of eight rows are loaded together from clamped addresses before any of them is used: a
            // load under a condition would be a branch with its own wait, one memory round trip per row.
            i64 k = -1, k_end = 0, k_base = 0;
            // the rows that hold a flag at all (bit q of `rows`): the values Y > 0 come in runs of 2 * radius + 1 around the
            // splice sites, so six rows in ten of a typical batch hold none and their values are not loaded
            u64 rows = 0;
            {
                const u64 lanes_set = __ballot(fm != 0);                 // bit l: lane l's 32 positions hold a flag
                u64 pairs = (lanes_set | (lanes_set >> 1)) & 0x5555555555555555ULL;     // bit 2q: row q
                pairs = (pairs | (pairs >> 1)) & 0x3333333333333333ULL;
                pairs = (pairs | (pairs >> 2)) & 0x0f0f0f0f0f0f0f0fULL;
                pairs = (pairs | (pairs >> 4)) & 0x00ff00ff00ff00ffULL;
                pairs = (pairs | (pairs >> 8)) & 0x0000ffff0000ffffULL;
                rows = (pairs | (pairs >> 16)) & 0x00000000ffffffffULL;
            }
            while (rows) {
                u64 m[8];
                double yv[8];
                int qs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {                            // the next eight rows that hold flags
                    const int q = rows ? (int)__builtin_ctzll(rows) : -1;
                    qs[e] = q;
                    rows = rows ? rows & (rows - 1) : 0;
                    const int qq = q < 0 ? 0 : q;
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)fm, 2 * qq);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)fm, 2 * qq + 1);
                    m[e] = q < 0 ? 0 : ((u64)lo | ((u64)hi << 32));
                    const i64 i = w0 + qq * 64 + lane;
                    if (MODE == kEmitValues) yv[e] = y[i < n ? i : n - 1];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (!m[e]) continue;
                    const i64 i = w0 + qs[e] * 64 + lane;
                    if ((m[e] >> lane) & 1ULL) {
                        const int d = ex + __popcll(m[e] & lt_mask);
                        if (MODE == kEmitValues) v[d] = yv[e];
                        else {
                            if (k < 0 || i >= k_end) {
                                // the interval of position i lies between the first intervals of this and the next block
                                const i64 ka = k < 0 ? blk_iv0[b] : k, kb = (i64)blk_iv0[b + 1] + 1;
                                k = ka + last_le(pos_off + ka, kb - ka, i);
                                k_base = pos_off[k]; k_end = pos_off[k + 1];
                            }
                            int yy = (int)(i - k_base);
                            out_y[d] = yy;
                            if (out_pos) out_pos[d] = iv_start[k] + yy;
                            if (out_iv) out_iv[d] = (int)k;
                            if (yy == 0) out_off[k] = d;
                        }
                    }
                    ex += __popcll(m[e]);
                }
            }
        }
    }
}

// rank of the first position of every partition in the compaction of the Y > 0 flags
__global__ void __launch_bounds__(64) k_voff(int n_part, const i64 *part_iv_off, const i64 *pos_off, i64 n_pos,
                                             const unsigned *flags, const int *bsum /* or null */, const u64 *state, const u64 *total, i64 *voff) {
    for (int p = blockIdx.x; p <= n_part; p += gridDim.x) {
        if (p == n_part) { if (threadIdx.x == 0) voff[p] = (i64)*total; continue; }
        i64 pos = pos_off[part_iv_off[p]];
        i64 b = pos / kScanBlock, start = b * kScanBlock;
        int cnt = 0;
        for (i64 i0 = start + (i64)threadIdx.x * 32; i0 < pos; i0 += 64 * 32) {
            Flags32 f = load_flags32(flags, i0, pos);      // positions at or after pos are masked out by the bound
            cnt += count_flags32(f);
        }
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        if (threadIdx.x == 0)                                       // look-back state: inclusive prefix of block b-1
            voff[p] = (bsum ? (i64)bsum[b] : (b ? (i64)(state[b - 1] & kScanValueMask) : 0)) + cnt;
    }
}


// ---------------------------------------------------------------------------------------------
// S3a  variance threshold   (py/freddie_segment.py:757-759)
// V = the Y > 0 values in (interval, position) order; thr = mean(V) + vf * std(V) with numpy's
// summation order: consecutive 8192-element chunks, each summed pairwise (8 strided accumulators
// below 129 elements, halves rounded down to a multiple of 8 above), chunk results added left to
// right.  Empty V gives NaN, which fixes nothing.
// ---------------------------------------------------------------------------------------------
// one workgroup: per-partition V ranges and chunk offsets
__global__ void __launch_bounds__(256) k_vplan(int n_part, const i64 *part_iv_off, const i64 *pos_off, i64 n_pos,
                                               i64 *voff, i64 *chunk_off, Status *st, i64 chunk_cap) {
    __shared__ int lds[16];
    __shared__ i64 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int p0 = 0; p0 < n_part; p0 += blockDim.x) {
        int p = p0 + threadIdx.x;
        int nch = 0;
        if (p < n_part) nch = (int)((voff[p + 1] - voff[p] + 8191) / 8192);
        int tot;
        int ex = wg_exclusive_scan(nch, lds, &tot);
        i64 carry = carry_s;
        if (p < n_part) chunk_off[p] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        chunk_off[n_part] = carry_s;
        st->n_vchunks = (u64)carry_s;
        if (carry_s > chunk_cap) atomicOr(&st->err, kErrOverflowChunks);
    }
}

// one 512-thread workgroup per chunk; pass 0 sums v, pass 1 sums (v-mean)^2.
// Thread (leaf, q) owns accumulator q of the 8-lane leaf of the pairwise recursion: r[q] = a[q] + a[8+q] +
// a[16+q] + ... in that order; the 8 accumulators are combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) by
// three xor-shuffles (IEEE addition is commutative, so both partners get identical bits), the tail (len%8)
// is added left to right, and the leaves are combined in recursion order.
__global__ void __launch_bounds__(512) k_vsum_chunks(int n_part, const i64 *voff, const i64 *chunk_off, const double *v,
                                                     const double *csum0, int pass, double *csum, i64 chunk_cap) {
    __shared__ VsumLds L;
    __shared__ double mu_s;
    i64 n_chunks = chunk_off[n_part];
    if (n_chunks > chunk_cap) n_chunks = chunk_cap;
    for (i64 c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        int p = (int)last_le(chunk_off, (i64)n_part + 1, c);
        i64 nv = voff[p + 1] - voff[p];
        i64 o0 = (c - chunk_off[p]) * 8192;
        int m = (int)((nv - o0) < 8192 ? (nv - o0) : 8192);
        const double *a = v + voff[p] + o0;
        __syncthreads();
        double mu = 0.0;
        if (pass) {
            // mean of the partition from the first pass' chunk sums, added left to right (numpy adds its 8192-element
            // blocks in order); every chunk of the partition repeats these few additions instead of a separate launch
            if (threadIdx.x == 0) {
                i64 c0 = chunk_off[p], c1 = chunk_off[p + 1];
                if (c1 > chunk_cap) c1 = chunk_cap;
                double sacc = 0.0;
                for (i64 cc = c0; cc < c1; ++cc) sacc = (cc == c0) ? csum0[cc] : __dadd_rn(sacc, csum0[cc]);
                mu_s = sacc / (double)nv;
            }
            __syncthreads();
            mu = mu_s;
        }
        const double x = vsum_chunk(a, m, pass, mu, L);
        if (threadIdx.x == 0) csum[c] = x;
    }
}

__global__ void __launch_bounds__(512) k_thr_part(int n_part, const i64 *part_iv_off, const i64 *pos_off, i64 n_pos, const unsigned *flags,
                                                  const double *__restrict__ y, double *v, double vf, double *mean, double *thr) {
    __shared__ VsumLds L;
    __shared__ int wave_cnt[8];
    __shared__ double cs[kThrPartMaxChunks];
    __shared__ double mu_s;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int p = blockIdx.x; p < n_part; p += gridDim.x) {
        const i64 pos0 = pos_off[part_iv_off[p]], pos1 = pos_off[part_iv_off[p + 1]];
        double *vp = v + pos0;
        // compaction: the partition's flag words in groups of 64 (2048 positions), every wave an eighth of the groups
        const i64 wbeg = pos0 >> 5, wend = (pos1 + 31) >> 5;
        const i64 groups = (wend - wbeg + 63) / 64, gpw = (groups + 7) / 8;
        const i64 g0 = (i64)wave * gpw, g1 = g0 + gpw < groups ? g0 + gpw : groups;
        auto flag_word = [&](i64 wd) -> unsigned {
            if (wd >= wend) return 0u;
            unsigned f = flags[wd];
            const i64 i0 = wd << 5;
            if (i0 < pos0) f &= ~0u << (int)(pos0 - i0);
            if (i0 + 32 > pos1) f &= (1u << (int)(pos1 - i0)) - 1u;
            return f;
        };
        int cnt = 0;
        for (i64 g = g0; g < g1; ++g) cnt += __popc(flag_word(wbeg + g * 64 + lane));
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        if (lane == 0) wave_cnt[wave] = cnt;
        __syncthreads();
        i64 ex = 0;
        int nv = 0;
        for (int w = 0; w < 8; ++w) { if (w < wave) ex += wave_cnt[w]; nv += wave_cnt[w]; }
        for (i64 g = g0; g < g1; ++g) ex += wave_emit_values((wbeg + g * 64) << 5, n_pos, flag_word(wbeg + g * 64 + lane), ex, y, vp);
        __builtin_amdgcn_s_waitcnt(0);                  // the values are read back by other waves of this workgroup
        __syncthreads();
        const int nch = (nv + 8191) / 8192;
        double s_acc[2] = {0.0, 0.0};
        for (int pass = 0; pass < 2; ++pass) {
            const double mu = pass ? mu_s : 0.0;
            for (int c = 0; c < nch; ++c) {
                const int m = nv - c * 8192 < 8192 ? nv - c * 8192 : 8192;
                const double x = vsum_chunk(vp + (i64)c * 8192, m, pass, mu, L);
                if (threadIdx.x == 0) cs[c] = x;
            }
            if (threadIdx.x == 0) {
                double sacc = 0.0;
                for (int c = 0; c < nch; ++c) sacc = c == 0 ? cs[0] : __dadd_rn(sacc, cs[c]);     // numpy adds its chunks left to right
                s_acc[pass] = sacc;
                if (pass == 0) mu_s = sacc / (double)nv;                                         // empty -> 0/0 = NaN like numpy
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const double mu = mu_s;
            mean[p] = mu;
            thr[p] = __dadd_rn(mu, __dmul_rn(vf, __dsqrt_rn(s_acc[1] / (double)nv)));             // :758-759
        }
        __syncthreads();
    }
}

__global__ void k_vsum_part(int n_part, const i64 *voff, const i64 *chunk_off, const double *csum0, const double *csum1,
                            double vf, double *mean, double *thr, i64 chunk_cap) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_part; p += gridDim.x * blockDim.x) {
        i64 c0 = chunk_off[p], c1 = chunk_off[p + 1];
        if (c1 > chunk_cap) c1 = chunk_cap;
        double n = (double)(voff[p + 1] - voff[p]);
        double s0 = 0.0, s1 = 0.0;
        for (i64 c = c0; c < c1; ++c) {
            s0 = (c == c0) ? csum0[c] : __dadd_rn(s0, csum0[c]);
            s1 = (c == c0) ? csum1[c] : __dadd_rn(s1, csum1[c]);
        }
        const double mu = s0 / n;                                       // empty -> 0/0 = NaN like numpy
        mean[p] = mu;
        thr[p] = __dadd_rn(mu, __dmul_rn(vf, __dsqrt_rn(s1 / n)));      // :758-759
    }
}

__global__ void __launch_bounds__(256) k_peaks_edges(int n_tiles, const TileDesc *tiles, const int *tile_defer, const double *x,
                                                     unsigned *flag, int *part_has2, int n_part) {
    // also clears the per-partition 'some default label is not 0' flags that k_label_cols sets much later
    if (blockIdx.x == 0) for (int p = threadIdx.x; p < n_part; p += blockDim.x) part_has2[p] = 0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += gridDim.x * blockDim.x) {
        const TileDesc d = tiles[t];
        const i64 first = d.base + d.y0, last = d.base + d.len - 1;
        const i64 tile_last = first + kSmoothTile - 1;
        if (d.y0 > 0 && first < last) peak_from(first, last, x, flag);              // (an interval's own ends are candidates already)
        if (tile_last < last && kSmoothTile > 1) peak_from(tile_last, last, x, flag);
        const int df = tile_defer[t];
        if (df >= 0) peak_from(first + df, last, x, flag);
    }
}


// the instances the host launches (freddie_seg.hip sees the declarations only: taking an instance's address here is what
// instantiates it -- host stub and device code -- in this translation unit)
__attribute__((used)) static const void *const kInstances[] = {
    reinterpret_cast<const void *>(&k_smooth<20>),
    reinterpret_cast<const void *>(&k_smooth<12>),
    reinterpret_cast<const void *>(&k_smooth<0>),
    reinterpret_cast<const void *>(&k_scan_emit<kEmitValues>),
    reinterpret_cast<const void *>(&k_scan_emit<kEmitPositions>),
};

}  // namespace fseg
