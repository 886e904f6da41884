// freddie_cluster.hip -- gfx950 kernels + C-ABI (include/freddie_cluster.h) for the pre-ILP work of the clustering
// stage: the pairwise read-compatibility graph of partition_reads() (py/freddie_cluster.py:217-234) and its iterated
// edge pruning (:240-255), for a batch of tints per call.
//
// Reads are bit rows (bit s = the read covers segment s), so the reference's two list comprehensions over the
// overlap [f, l] (:229, :232) become popcounts of (a & b & mask) and ((a ^ b) & mask).  The graph is a symmetric
// bit matrix; a 64 x 64 tile of it is one workgroup's unit of work, lane = column, so a row's 64 edge bits are one
// wave ballot and one 8-byte store.  Pruning keeps an edge when either end has no other neighbour or the two ends
// share a neighbour (:247-251): the columns that share a neighbour with row r are the OR of the rows of r's
// neighbours, one wave per row; every pass reads the previous pass' matrix only (the reference removes the edges
// of a pass together, :252) and passes repeat until one removes nothing (:254).
#include "freddie_cluster.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

typedef long long i64;
typedef unsigned long long u64;

constexpr int kTile = 64;           // rows and columns per tile
constexpr int kMaxWords = 300;      // uint32 words per read row the LDS staging of k_compat can hold (9600 segments)

struct TintDesc {
    i64 row0, bits_off, adj_off;
    int n, n_seg, w, aw;            // rows, segments, uint32 words per read row, uint64 words per adjacency row
    int in_lds, pad;                // the tint's pruning runs whole in one workgroup's LDS (k_prune_lds): the per-pass kernels skip its rows
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// bits [f, l] of the 32-bit word number w (f <= l, both inside the row)
__device__ __forceinline__ unsigned range_mask(int f, int l, int w) {
    const int lo = f - w * 32, hi = l - w * 32;
    unsigned m = 0xffffffffu;
    if (lo > 0) m &= 0xffffffffu << lo;
    if (hi < 31) m &= 0xffffffffu >> (31 - hi);
    return m;
}

// ---- pairwise compatibility (py/freddie_cluster.py:217-234) ------------------------------------------------------
// The relation is symmetric, so only the tiles on and above the diagonal are computed: a workgroup writes its tile's rows (a
// row's 64 edge bits are one ballot) AND the transposed tile (lane = column keeps its own bit of every row; the waves' sixteen
// rows each meet in LDS).
// RANK (rows of at most kRankWords words): besides the bit rows, the number of a row's bits in front of each of its words is
// staged (16 bits each; as 8-byte entries {word, rank} the big tint ran at two workgroups per CU instead of three and
// took 3.0 instead of 2.0 ms).  A read's bits lie inside [first, last] (checked on the host), so over the pair's overlap [f, l]
//   same = popcount(a & b) over the overlap's words, no mask (a & b has no bit outside [f, l]);
//   diff = bits of a in [f, l] + bits of b in [f, l] - 2 same, and "bits of a in [f, l]" is the ranks of the overlap's first and
//          last word and two masked popcounts of those words -- which the sum over the overlap reads anyway.
// Most overlaps lie in one or two words: the first and the last word are taken outside the loop (four LDS reads a pair), the
// loop runs over what lies between (its trip count is the longest overlap of the wave's 64 pairs).
// !RANK: longer rows (up to kMaxWords): both sums with a range mask per word.
constexpr int kRankWords = 207;     // (64 + 64) rows x (4 + 2) bytes x 207 words + the static arrays <= 160 KB of LDS

template <bool RANK>
__global__ void __launch_bounds__(256) k_compat(int n_tiles, const int4 *tiles, const TintDesc *tints, const unsigned *bits,
                                                const int *first, const int *last, const unsigned char *tail, u64 *adj) {
    extern __shared__ unsigned lds[];
    __shared__ int row_f[kTile], row_l[kTile], row_t[kTile];
    __shared__ unsigned col_part[4][kTile];
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
        const int4 tile = tiles[ti];
        const TintDesc d = tints[tile.x];
        const int W = d.w, Wp = W | 1;                  // odd row stride: lanes reading the same word of 64 rows spread over the banks
        unsigned *rows = lds, *cols = lds + kTile * Wp;
        unsigned short *rrank = reinterpret_cast<unsigned short *>(cols + kTile * Wp), *crank = rrank + kTile * Wp;
        const unsigned *B = bits + d.bits_off;
        const int r0 = tile.y * kTile, c0 = tile.z * kTile;
        __syncthreads();
        {
            const float inv_w = 1.0f / (float)W;        // x / W for x < 64 * kMaxWords: a float quotient and one correction
            for (int x = threadIdx.x; x < kTile * W; x += blockDim.x) {
                int q = (int)((float)x * inv_w), w = x - q * W;
                if (w >= W) { ++q; w -= W; } else if (w < 0) { --q; w += W; }
                rows[q * Wp + w] = r0 + q < d.n ? B[(i64)(r0 + q) * W + w] : 0u;
                cols[q * Wp + w] = c0 + q < d.n ? B[(i64)(c0 + q) * W + w] : 0u;
            }
        }
        // the tile's rows' first / last covered segment and tail, staged with the bit rows (round 5): the row loop below read them
        // from global memory, three wave-uniform loads in front of every row's 64 pairs -- a microsecond of latency per row
        if (threadIdx.x < kTile) {
            const int row = r0 + threadIdx.x;
            const bool ok = row < d.n;
            row_f[threadIdx.x] = ok ? first[d.row0 + row] : 0;
            row_l[threadIdx.x] = ok ? last[d.row0 + row] : -1;
            row_t[threadIdx.x] = ok ? tail[d.row0 + row] : 0;
        }
        __syncthreads();
        if (RANK) {
            if (threadIdx.x < 2 * kTile) {              // a thread per staged row: the bits in front of each of its words
                const unsigned *src = (threadIdx.x < kTile ? rows : cols) + (threadIdx.x & (kTile - 1)) * Wp;
                unsigned short *dst = (threadIdx.x < kTile ? rrank : crank) + (threadIdx.x & (kTile - 1)) * Wp;
                unsigned run = 0;
                for (int w = 0; w < W; ++w) { dst[w] = (unsigned short)run; run += __popc(src[w]); }
            }
            __syncthreads();
        }
        const int col = c0 + lane;
        const bool col_ok = col < d.n;
        int f2 = 0, l2 = -1, t2 = 0;
        if (col_ok) { f2 = first[d.row0 + col]; l2 = last[d.row0 + col]; t2 = tail[d.row0 + col]; }
        const unsigned *b = cols + lane * Wp;
        const unsigned short *rb = crank + lane * Wp;
        unsigned mine = 0;                               // this column's edge bits of the wave's sixteen rows
        for (int k = 0; k < kTile / 4; ++k) {
            const int rr = wave * (kTile / 4) + k, row = r0 + rr;
            if (row >= d.n) break;
            const int f1 = row_f[rr], l1 = row_l[rr], t1 = row_t[rr];
            const int f = f1 > f2 ? f1 : f2, l = l1 < l2 ? l1 : l2;         // overlap of the two reads (:224-226)
            const int o = l - f + 1;
            bool edge = false;
            // poly-A tails on different ends: incompatible (:222-223); f < 0 only when neither read covers any segment: then no
            // common segment either (:228-230).  (Straight-line code that computes every pair and drops the untested ones at the
            // end was slower: whole waves skip here -- 0.168 against 0.151 ms on 400 tints of 500 reads.)
            if (col_ok && col != row && !(t1 != 0 && t2 != 0 && t1 != t2) && o >= 1 && f >= 0) {
                const unsigned *a = rows + rr * Wp;
                int same = 0, diff = 0;
                if (RANK) {
                    const unsigned short *ra = rrank + rr * Wp;
                    const int wf = f >> 5, wl = l >> 5;
                    const unsigned a0 = a[wf], a1 = a[wl], b0 = b[wf], b1 = b[wl];
                    same = __popc(a0 & b0);                                             // segments both reads cover (:229)
                    if (wl > wf) {
                        same += __popc(a1 & b1);
                        for (int w = wf + 1; w < wl; ++w) same += __popc(a[w] & b[w]);
                    }
                    const unsigned below_f = (1u << (f & 31)) - 1u, upto_l = 0xffffffffu >> (31 - (l & 31));
                    const int in_a = (int)ra[wl] + __popc(a1 & upto_l) - (int)ra[wf] - __popc(a0 & below_f);
                    const int in_b = (int)rb[wl] + __popc(b1 & upto_l) - (int)rb[wf] - __popc(b0 & below_f);
                    diff = in_a + in_b - 2 * same;                                      // segments where they differ (:232)
                } else {
                    for (int w = f >> 5; w <= (l >> 5); ++w) {
                        const unsigned m = range_mask(f, l, w), x = a[w], y = b[w];
                        same += __popc(x & y & m);           // segments both reads cover (:229)
                        diff += __popc((x ^ y) & m);         // segments where they differ (:232)
                    }
                }
                edge = same >= 1 && ((o > 3 && diff < 3) || (o <= 3 && diff == 0));   // :230, :234
            }
            const u64 word = __ballot(edge);
            if (lane == 0) adj[d.adj_off + (i64)row * d.aw + tile.z] = word;
            mine |= (unsigned)edge << k;
        }
        if (tile.y != tile.z) {                          // (workgroup-uniform) the tile below the diagonal: this one transposed
            col_part[wave][lane] = mine;
            __syncthreads();
            if (threadIdx.x < kTile && col_ok)
                adj[d.adj_off + (i64)col * d.aw + tile.y] = (u64)col_part[0][lane] | ((u64)col_part[1][lane] << 16) |
                                                            ((u64)col_part[2][lane] << 32) | ((u64)col_part[3][lane] << 48);
        }
    }
}

// ---- degrees ---------------------------------------------------------------------------------------------------
// `gate` (all three kernels of a pass): the "some edge was removed" word of the PREVIOUS pass, or null for the first pass of
// a burst.  The host enqueues several passes back to back and reads the flags once per burst; the passes after the one that
// removed nothing find their gate at zero and return at once (py/freddie_cluster.py:240-255 loops until nothing changes).
__global__ void __launch_bounds__(256) k_degree(i64 n_rows_total, const int *row_tint, const TintDesc *tints, const u64 *adj, int *deg,
                                                const int *gate) {
    if (gate && *gate == 0) return;
    const int lane = lane_id();
    const i64 wave_g = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((i64)gridDim.x * blockDim.x) >> 6;
    for (i64 r = wave_g; r < n_rows_total; r += n_waves) {
        const TintDesc d = tints[row_tint[r]];
        if (d.in_lds) continue;
        const u64 *a = adj + d.adj_off + (r - d.row0) * d.aw;
        int c = 0;
        for (int w = lane; w < d.aw; w += 64) c += __popcll(a[w]);
        for (int s = 32; s >= 1; s >>= 1) c += __shfl_xor(c, s);
        if (lane == 0) deg[r] = c;
    }
}

// ---- one pruning pass (py/freddie_cluster.py:243-252) ------------------------------------------------------------
// new(r, c) = old(r, c) and (deg r == 1 or deg c == 1 or r and c share a neighbour).
// "r and c share a neighbour" for all c at once: H(r) = OR of the rows of r's neighbours (the matrix is symmetric, so
// bit c of neighbour k's row says k ~ c).  One wave per row, lanes = 64 consecutive words of the row, the loop runs over
// the set bits of row r (wave-uniform) and ORs the neighbour's words (a coalesced read of the neighbour's row): the
// work is sum(deg) * words instead of one LDS-staged 64 x 64 block pair per tile of the matrix.
__global__ void __launch_bounds__(256) k_deg1(int n_words_total, const int2 *word_tint, const TintDesc *tints, const int *deg, u64 *deg1,
                                              const int *gate) {
    if (gate && *gate == 0) return;
    // deg1[tint word z] bit c = column 64 z + c has exactly one neighbour
    const int lane = lane_id();
    const i64 wave_g = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((i64)gridDim.x * blockDim.x) >> 6;
    for (i64 x = wave_g; x < n_words_total; x += n_waves) {
        const int2 wt = word_tint[x];                    // (tint, word index inside the tint)
        const TintDesc d = tints[wt.x];
        if (d.in_lds) continue;
        const int col = wt.y * 64 + lane;
        const u64 m = __ballot(col < d.n && deg[d.row0 + col] == 1);
        if (lane == 0) deg1[x] = m;
    }
}
// The same pass, edge by edge (round 6; rows of at most kEdgeChunks x 64 words = 32 768 reads): only the columns c that ARE neighbours
// of r need an answer, and "r and c share a neighbour" is "row r AND row c is not empty" (the matrix is symmetric).  A wave takes a row,
// keeps it in registers (lane = word, kEdgeChunks words a lane), walks its set bits in order and for each neighbour c reads row c 64 words
// at a time until a word of the AND is not zero -- in these graphs (reads of one gene: triangles everywhere) the first 512 bytes nearly
// always answer, where the OR of all of N(r)'s rows (k_prune below) reads every row whole, once per 64-word chunk of the output:
// 19 539 reads, mean degree 239: 8.7 -> ~2 ms a pass.  Removed bits are cleared in the lane that holds their word.
constexpr int kEdgeChunks = 8;

__global__ void __launch_bounds__(256) k_prune_edges(i64 n_rows_total, int max_chunks, const int *row_tint, const TintDesc *tints, const u64 *old_adj, const int *deg,
                                                     u64 *new_adj, int *changed /* per tint */, int *pass_any, const int *gate) {
    if (gate && *gate == 0) return;
    const int lane = lane_id();
    const i64 wave_g = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((i64)gridDim.x * blockDim.x) >> 6;
    // a work item = (row r, 64-word chunk qc of it): the wave tests the neighbours whose bits lie in that chunk and writes that chunk of
    // the new row.  (A wave per ROW ended with its longest rows: degrees reach 7 000 where the mean is 239.)
    for (i64 item = wave_g; item < n_rows_total * max_chunks; item += n_waves) {
        const i64 r = item / max_chunks;
        const int qc = (int)(item - r * max_chunks);
        const int t = row_tint[r];
        const TintDesc d = tints[t];
        if (d.in_lds) continue;
        const u64 *A = old_adj + d.adj_off;
        const i64 rl = r - d.row0;
        const int aw = d.aw, n_chunks = (aw + 63) >> 6;
        if (qc >= n_chunks) continue;
        u64 mine[kEdgeChunks];
#pragma unroll
        for (int q = 0; q < kEdgeChunks; ++q) { const int z = q * 64 + lane; mine[q] = (q < n_chunks && z < aw) ? A[rl * aw + z] : 0ull; }
        u64 own = 0;                                          // this item's chunk of the row
#pragma unroll
        for (int q = 0; q < kEdgeChunks; ++q) if (q == qc) own = mine[q];
        u64 keep = own;
        const int deg_r = deg[r];
        bool any_change = false;
        if (deg_r > 1) {                                       // (deg 1: the edge stays; deg 0: nothing to do)
            // Neighbours are taken FOUR at a time: their degrees and the first 64 words of their rows are asked for together and
            // looked at afterwards -- one neighbour at a time the walk was a chain of two dependent loads (~1 us) per neighbour.
            i64 bc[4] = {0, 0, 0, 0}; int bl[4] = {0, 0, 0, 0}, bb[4] = {0, 0, 0, 0};
            int cnt = 0;
            auto flush = [&]() {
                int dg[4]; u64 cw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) dg[u] = deg[d.row0 + bc[u < cnt ? u : 0]];
#pragma unroll
                for (int u = 0; u < 4; ++u) cw[u] = lane < aw ? A[bc[u < cnt ? u : 0] * aw + lane] : 0ull;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u >= cnt) break;
                    bool stays = dg[u] == 1;
                    if (!stays) stays = __ballot((cw[u] & mine[0]) != 0ull) != 0ull;
                    if (!stays) {
                        const u64 *C = A + bc[u] * aw;
#pragma unroll
                        for (int q2 = 1; q2 < kEdgeChunks; ++q2) {
                            if (q2 >= n_chunks) break;
                            const int z = q2 * 64 + lane;
                            const u64 w2 = z < aw ? C[z] : 0ull;
                            if (__ballot((w2 & mine[q2]) != 0ull)) { stays = true; break; }
                        }
                    }
                    if (!stays) { if (lane == bl[u]) keep &= ~(1ull << bb[u]); any_change = true; }
                }
                cnt = 0;
            };
            u64 have = __ballot(own != 0ull);                  // the lanes whose word of this chunk holds a neighbour
            while (have) {
                const int L = __builtin_amdgcn_readfirstlane(__ffsll((long long)have) - 1);     // (wave-uniform: a scalar for readlane)
                have &= have - 1;
                u64 word = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(own >> 32), L) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane((int)own, L);
                while (word) {
                    const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)word) - 1);
                    word &= word - 1;
                    // (the newest neighbour enters at slot 0 and the others move up: static register indices, no scratch; the order
                    // inside a batch does not matter)
#pragma unroll
                    for (int u = 3; u > 0; --u) { bc[u] = bc[u - 1]; bl[u] = bl[u - 1]; bb[u] = bb[u - 1]; }
                    bc[0] = ((i64)qc * 64 + L) * 64 + b; bl[0] = L; bb[0] = b;                 // the neighbour (wave-uniform)
                    if (++cnt == 4) flush();
                }
            }
            if (cnt) flush();
        }
        { const int z = qc * 64 + lane; if (z < aw) new_adj[d.adj_off + rl * aw + z] = keep; }
        if (any_change && lane == 0) { changed[t] = 1; *pass_any = 1; }
    }
}

__global__ void __launch_bounds__(256) k_prune(i64 n_rows_total, const int *row_tint, const TintDesc *tints, const i64 *tint_word0,
                                               const u64 *old_adj, const int *deg, const u64 *deg1, u64 *new_adj,
                                               int *changed /* per tint */, int *pass_any, const int *gate) {
    if (gate && *gate == 0) return;
    const int lane = lane_id();
    const i64 wave_g = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((i64)gridDim.x * blockDim.x) >> 6;
    for (i64 r = wave_g; r < n_rows_total; r += n_waves) {
        const int t = row_tint[r];
        const TintDesc d = tints[t];
        if (d.in_lds) continue;
        const u64 *A = old_adj + d.adj_off;
        const i64 rl = r - d.row0;
        const int deg_r = deg[r];
        const u64 *d1 = deg1 + tint_word0[t];
        bool any_change = false;
        for (int z0 = 0; z0 < d.aw; z0 += 64) {
            const int z = z0 + lane;
            const bool zin = z < d.aw;
            const u64 oldw = zin ? A[rl * d.aw + z] : 0ull;
            u64 keep = ~0ull;
            if (deg_r != 1) {                            // wave-uniform
                u64 acc = 0;
                for (int wi = 0; wi < d.aw; ++wi) {
                    u64 word = A[rl * d.aw + wi];        // the same address in every lane
                    while (word) {
                        const int k = wi * 64 + __ffsll((long long)word) - 1;
                        word &= word - 1;
                        acc |= A[(i64)k * d.aw + (zin ? z : 0)];
                    }
                }
                keep = acc | (zin ? d1[z] : 0ull);
            }
            const u64 neww = oldw & keep;
            if (zin) new_adj[d.adj_off + rl * d.aw + z] = neww;
            any_change |= neww != oldw;
        }
        if (__ballot(any_change) && lane == 0) { changed[t] = 1; *pass_any = 1; }
    }
}

// ---- the whole pruning of a small tint by ONE workgroup, in LDS (round 6) ------------------------------------------
// A tint whose bit matrix fits LDS twice (rows x words <= kPruneLdsWords: 500 reads are 32 KB a copy) is pruned to its fixed point
// without leaving the workgroup: degrees, the "exactly one neighbour" column mask, one pass into the other copy, again until a pass
// removes nothing (:240-255) -- no launch per pass, no flags to the host, no pass over tints that are already done.  The per-pass
// kernels above are what tints too large for this use (and they skip the rows of the tints that are not theirs).
// A thread owns (row r, word z) pairs: new = old & (deg r == 1 ? all : H(r)[z] | deg1[z]) with H(r) = OR of the rows of r's
// neighbours; only the bits of old[r][z] that are not excused by deg1 need a common neighbour, and the walk over r's neighbours
// ends as soon as they all have one (in these graphs -- reads of one gene -- after a handful of neighbours).
constexpr int kPruneLdsWords = 7808;       // u64 words of one copy: 2 x 61 KB, two such workgroups share a CU's LDS with room for the rest

__global__ void __launch_bounds__(256) k_prune_lds(const int *small_tints, const TintDesc *tints, u64 *adj0, u64 *adj1, int *rounds) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ int s_changed;
    const int t = small_tints[blockIdx.x];
    const TintDesc d = tints[t];
    const int n = d.n, aw = d.aw, nw = n * aw;
    u64 *buf[2] = {reinterpret_cast<u64 *>(lds_raw), reinterpret_cast<u64 *>(lds_raw) + nw};
    u64 *d1 = buf[1] + nw;
    unsigned short *deg = reinterpret_cast<unsigned short *>(d1 + aw);
    for (int x = threadIdx.x; x < nw; x += blockDim.x) buf[0][x] = adj0[d.adj_off + x];
    __syncthreads();
    int cur = 0, n_rounds = 0;
    for (;;) {
        const u64 *old = buf[cur];
        u64 *nxt = buf[cur ^ 1];
        for (int r = threadIdx.x; r < n; r += blockDim.x) {
            int c = 0;
            for (int z = 0; z < aw; ++z) c += __popcll(old[r * aw + z]);
            deg[r] = (unsigned short)c;
        }
        if (threadIdx.x == 0) s_changed = 0;
        __syncthreads();
        for (int z = threadIdx.x; z < aw; z += blockDim.x) {
            u64 m = 0;
            for (int b = 0; b < 64; ++b) { const int col = z * 64 + b; if (col < n && deg[col] == 1) m |= 1ull << b; }
            d1[z] = m;
        }
        __syncthreads();
        bool ch = false;
        for (int x = threadIdx.x; x < nw; x += blockDim.x) {
            const int r = x / aw, z = x - r * aw;
            const u64 oldw = old[x];
            u64 neww = oldw;
            if (oldw && deg[r] != 1) {
                const u64 need = oldw & ~d1[z];                 // these bits stay only with a common neighbour
                u64 acc = 0;
                for (int wi = 0; wi < aw && (need & ~acc); ++wi) {
                    u64 word = old[r * aw + wi];
                    while (word && (need & ~acc)) {
                        const int k = wi * 64 + __ffsll((long long)word) - 1;
                        word &= word - 1;
                        acc |= old[k * aw + z];
                    }
                }
                neww = oldw & (acc | d1[z]);
            }
            nxt[x] = neww;
            ch |= neww != oldw;
        }
        if (ch) s_changed = 1;
        __syncthreads();
        const int any = s_changed;
        __syncthreads();                                        // (everybody has read the flag before the next pass clears it)
        cur ^= 1;
        if (!any) break;
        ++n_rounds;
    }
    for (int x = threadIdx.x; x < nw; x += blockDim.x) { const u64 v = buf[cur][x]; adj0[d.adj_off + x] = v; adj1[d.adj_off + x] = v; }
    if (threadIdx.x == 0) rounds[t] = n_rounds;
}

}  // namespace

struct GrowBuf {              // device buffer that lives with the context and only ever grows
    void *p = nullptr;
    size_t cap = 0;
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};
struct fclu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {};
    std::string err;
    float compat_ms = 0.f, prune_ms = 0.f;
    GrowBuf tints, tiles, row_tint, bits, first, last, tail, adj[2], deg, changed, word_tint, tint_word0, deg1, pass_any, small_tints, small_rounds;
    int *h_flags = nullptr;   // pinned: per-pass flags of a burst + per-tint flags
    size_t h_flags_cap = 0;
};

namespace {

std::string g_create_error;

int fail(fclu_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

hipError_t grow(GrowBuf &b, size_t bytes) {
    if (b.p && bytes <= b.cap) return hipSuccess;
    if (b.p) { hipError_t e = hipFree(b.p); b.p = nullptr; b.cap = 0; if (e != hipSuccess) return e; }
    const size_t want = (bytes ? bytes : 16) + bytes / 4;
    hipError_t e = hipMalloc(&b.p, want);
    if (e == hipSuccess) b.cap = want;
    return e;
}
constexpr int kBurst = 4;     // pruning passes enqueued per host round trip

#define HIP_TRY(c, expr)                                                                                     \
    do {                                                                                                     \
        hipError_t e__ = (expr);                                                                             \
        if (e__ != hipSuccess) return fail((c), FCLU_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__));      \
    } while (0)

}  // namespace

extern "C" {

int fclu_abi_version(void) { return 1; }

int fclu_create(int device, fclu_ctx **out) {
    if (!out) return FCLU_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, FCLU_ERR_HIP, "no HIP device available: %s (this library has no CPU fallback)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(nullptr, FCLU_ERR_ARG, "device ordinal out of range");
    fclu_ctx *c = new fclu_ctx();
    c->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < 3; ++i) e = hipEventCreate(&c->ev[i]);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_compat<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * kTile * (kMaxWords | 1) * 4);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_compat<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * kTile * (kRankWords | 1) * 6);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_prune_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) {
        fail(nullptr, FCLU_ERR_HIP, "context creation failed: %s", hipGetErrorString(e));
        delete c;
        return FCLU_ERR_HIP;
    }
    *out = c;
    return FCLU_OK;
}

void fclu_destroy(fclu_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (int i = 0; i < 3; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    GrowBuf *bufs[] = {&c->tints, &c->tiles, &c->row_tint, &c->bits, &c->first, &c->last, &c->tail, &c->adj[0], &c->adj[1], &c->deg,
                       &c->changed, &c->word_tint, &c->tint_word0, &c->deg1, &c->pass_any, &c->small_tints, &c->small_rounds};
    for (GrowBuf *b : bufs) if (b->p) (void)hipFree(b->p);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    delete c;
}

const char *fclu_last_error(const fclu_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int fclu_compat_graph(fclu_ctx *c, const fclu_batch *b, int32_t prune, uint64_t *adj_out, int32_t *rounds_out) {
    if (!c || !b || !adj_out) return FCLU_ERR_ARG;
    if (b->n_tint <= 0) return fail(c, FCLU_ERR_ARG, "fclu_compat_graph: empty batch");
    HIP_TRY(c, hipSetDevice(c->device));
    const int T = b->n_tint;
    const i64 R = b->row_off[T];
    // host-side shape checks: the kernels index with these and nothing else
    std::vector<TintDesc> tints((size_t)T);
    std::vector<int4> tiles;
    std::vector<int> row_tint((size_t)R);
    std::vector<int2> word_tint;                         // every adjacency word column of every tint: (tint, word)
    std::vector<i64> tint_word0((size_t)T + 1, 0);
    std::vector<int> small_tints;                        // pruned whole in LDS (k_prune_lds); FCLU_PRUNE_LDS=0: none (tests)
    const char *lds_env = getenv("FCLU_PRUNE_LDS");
    const bool lds_ok = !(lds_env && lds_env[0] == '0');
    const char *ldsw_env = getenv("FCLU_PRUNE_LDS_WORDS");                // (tests: a lower limit, so that small tints take both ways in one batch)
    const i64 lds_words = (ldsw_env && atoll(ldsw_env) > 0 && atoll(ldsw_env) < kPruneLdsWords) ? atoll(ldsw_env) : kPruneLdsWords;
    size_t small_lds = 0;
    bool any_large = false;
    int max_w = 1, max_aw_large = 0;
    for (int t = 0; t < T; ++t) {
        TintDesc &d = tints[(size_t)t];
        d.row0 = b->row_off[t];
        const i64 n = b->row_off[t + 1] - d.row0;
        if (n < 0 || n > (1 << 30) || b->n_seg[t] < 0) return fail(c, FCLU_ERR_ARG, "tint %d: bad row count or segment count", t);
        d.n = (int)n; d.n_seg = b->n_seg[t];
        d.w = (d.n_seg + 31) / 32; if (d.w < 1) d.w = 1;
        d.aw = (d.n + 63) / 64;
        d.bits_off = b->bits_off[t]; d.adj_off = b->adj_off[t];
        d.in_lds = (prune && lds_ok && d.n > 0 && d.n <= 65535 && (i64)d.n * d.aw <= lds_words) ? 1 : 0; d.pad = 0;
        if (d.in_lds) { small_tints.push_back(t); small_lds = std::max(small_lds, ((size_t)2 * d.n * d.aw + d.aw) * 8 + (size_t)d.n * 2 + 16); }
        else if (d.n > 0) { any_large = true; max_aw_large = std::max(max_aw_large, d.aw); }
        if (b->bits_off[t + 1] - d.bits_off != (i64)d.n * d.w) return fail(c, FCLU_ERR_ARG, "tint %d: bits_off does not match rows x words", t);
        if (b->adj_off[t + 1] - d.adj_off != (i64)d.n * d.aw) return fail(c, FCLU_ERR_ARG, "tint %d: adj_off does not match rows x words", t);
        if (d.w > kMaxWords) return fail(c, FCLU_ERR_UNSUPPORTED, "tint %d has %d segments; this build stages at most %d", t, d.n_seg, kMaxWords * 32);
        if (d.w > max_w) max_w = d.w;
        for (i64 r = 0; r < n; ++r) {
            row_tint[(size_t)(d.row0 + r)] = t;
            const int f = b->first[d.row0 + r], l = b->last[d.row0 + r];
            if (f < -1 || l >= (d.n_seg > 0 ? d.n_seg : 1) || b->tail[d.row0 + r] > 2) return fail(c, FCLU_ERR_ARG, "tint %d read %lld: first/last/tail out of range", t, r);
            // a read's bits lie inside [first, last] (first / last ARE its first and last covered segment, :175-183): k_compat counts on it
            const uint32_t *rw = b->bits + d.bits_off + r * d.w;
            for (int w = 0; w < d.w; ++w) {
                uint32_t allowed = 0;
                if (f >= 0 && l >= f && w >= (f >> 5) && w <= (l >> 5)) {
                    allowed = 0xffffffffu;
                    if (w == (f >> 5)) allowed &= 0xffffffffu << (f & 31);
                    if (w == (l >> 5)) allowed &= 0xffffffffu >> (31 - (l & 31));
                }
                if (rw[w] & ~allowed) return fail(c, FCLU_ERR_ARG, "tint %d read %lld: a covered segment outside [first, last]", t, r);
            }
        }
        for (int ti = 0; ti < d.aw; ++ti) for (int tj = ti; tj < d.aw; ++tj) tiles.push_back(make_int4(t, ti, tj, 0));   // (on and above the diagonal)
        for (int z = 0; z < d.aw; ++z) word_tint.push_back(make_int2(t, z));
        tint_word0[(size_t)t + 1] = (i64)word_tint.size();
    }
    const i64 n_bits = b->bits_off[T], n_adj = b->adj_off[T];
    const int n_tiles = (int)tiles.size();
    if (rounds_out) for (int t = 0; t < T; ++t) rounds_out[t] = 0;
    c->compat_ms = c->prune_ms = 0.f;
    if (n_tiles == 0 || R == 0) return FCLU_OK;

    GrowBuf &d_tints = c->tints, &d_tiles = c->tiles, &d_row_tint = c->row_tint, &d_bits = c->bits, &d_first = c->first, &d_last = c->last,
            &d_tail = c->tail, &d_deg = c->deg, &d_changed = c->changed, &d_word_tint = c->word_tint, &d_tint_word0 = c->tint_word0,
            &d_deg1 = c->deg1, &d_pass_any = c->pass_any;
    GrowBuf *d_adj = c->adj;
    HIP_TRY(c, grow(d_tints, tints.size() * sizeof(TintDesc)));
    HIP_TRY(c, grow(d_tiles, tiles.size() * sizeof(int4)));
    HIP_TRY(c, grow(d_row_tint, (size_t)R * 4));
    HIP_TRY(c, grow(d_bits, (size_t)n_bits * 4));
    HIP_TRY(c, grow(d_first, (size_t)R * 4));
    HIP_TRY(c, grow(d_last, (size_t)R * 4));
    HIP_TRY(c, grow(d_tail, (size_t)R));
    HIP_TRY(c, grow(d_adj[0], (size_t)n_adj * 8));
    HIP_TRY(c, grow(d_adj[1], (size_t)n_adj * 8));
    HIP_TRY(c, grow(d_deg, (size_t)R * 4));
    HIP_TRY(c, grow(d_changed, (size_t)kBurst * T * 4));
    HIP_TRY(c, grow(d_word_tint, word_tint.size() * sizeof(int2)));
    HIP_TRY(c, grow(d_tint_word0, tint_word0.size() * 8));
    HIP_TRY(c, grow(d_deg1, word_tint.size() * 8));
    HIP_TRY(c, grow(d_pass_any, (size_t)kBurst * 4));
    HIP_TRY(c, grow(c->small_tints, small_tints.size() * 4 + 4));
    HIP_TRY(c, grow(c->small_rounds, (size_t)T * 4));
    {
        const size_t need = ((size_t)kBurst * T + kBurst) * 4;
        if (need > c->h_flags_cap) {
            if (c->h_flags) HIP_TRY(c, hipHostFree(c->h_flags));
            c->h_flags = nullptr; c->h_flags_cap = 0;
            HIP_TRY(c, hipHostMalloc((void **)&c->h_flags, need + need / 4, hipHostMallocDefault));
            c->h_flags_cap = need + need / 4;
        }
    }
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(d_tints.p, tints.data(), tints.size() * sizeof(TintDesc), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_tiles.p, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_row_tint.p, row_tint.data(), (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_word_tint.p, word_tint.data(), word_tint.size() * sizeof(int2), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_tint_word0.p, tint_word0.data(), tint_word0.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_bits.p, b->bits, (size_t)n_bits * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_first.p, b->first, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_last.p, b->last, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_tail.p, b->tail, (size_t)R, hipMemcpyHostToDevice, s));

    const int grid = n_tiles < 8192 ? n_tiles : 8192;
    // rows of at most kRankWords words: the rank tables ride along (FCLU_RANK=0 keeps the masked sums: tests)
    const char *rank_env = getenv("FCLU_RANK");
    const bool rank = max_w <= kRankWords && !(rank_env && rank_env[0] == '0');
    const size_t lds = (size_t)2 * kTile * (max_w | 1) * (rank ? 6 : 4);
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    if (rank)
        hipLaunchKernelGGL(k_compat<true>, dim3(grid), dim3(256), lds, s, n_tiles, d_tiles.as<int4>(), d_tints.as<TintDesc>(),
                           d_bits.as<unsigned>(), d_first.as<int>(), d_last.as<int>(), d_tail.as<unsigned char>(), d_adj[0].as<u64>());
    else
        hipLaunchKernelGGL(k_compat<false>, dim3(grid), dim3(256), lds, s, n_tiles, d_tiles.as<int4>(), d_tints.as<TintDesc>(),
                           d_bits.as<unsigned>(), d_first.as<int>(), d_last.as<int>(), d_tail.as<unsigned char>(), d_adj[0].as<u64>());
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    int cur = 0;
    if (prune && !small_tints.empty()) {
        // (the final matrix of such a tint goes into BOTH copies: whichever the per-pass kernels of the large tints end on holds it)
        HIP_TRY(c, hipMemcpyAsync(c->small_tints.p, small_tints.data(), small_tints.size() * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_prune_lds, dim3((unsigned)small_tints.size()), dim3(256), small_lds, s, c->small_tints.as<int>(), d_tints.as<TintDesc>(),
                           d_adj[0].as<u64>(), d_adj[1].as<u64>(), c->small_rounds.as<int>());
    }
    if (prune && any_large) {
        // Passes are enqueued kBurst at a time; pass q of a burst is gated on pass q-1's "removed something" word, so the host
        // reads the flags once per burst instead of once per pass (the loop of :240-255 usually ends after two or three).
        const int deg_grid = (int)((R + 3) / 4 < 4096 ? (R + 3) / 4 : 4096);
        // the pass edge by edge (k_prune_edges) when every tint of the per-pass kernels has rows of at most kEdgeChunks x 64 words;
        // FCLU_PRUNE_EDGES=0: the OR-of-rows form whatever the shapes (tests)
        const char *edge_env = getenv("FCLU_PRUNE_EDGES");
        const bool edge_walk = max_aw_large <= kEdgeChunks * 64 && !(edge_env && edge_env[0] == '0');
        const int edge_chunks = (max_aw_large + 63) / 64 > 0 ? (max_aw_large + 63) / 64 : 1;
        const int n_words = (int)word_tint.size();
        int *h_any = c->h_flags, *h_changed = c->h_flags + kBurst;
        bool done = false;
        for (int burst = 0; burst < (1 << 18) && !done; ++burst) {
            HIP_TRY(c, hipMemsetAsync(d_changed.p, 0, (size_t)kBurst * T * 4, s));
            HIP_TRY(c, hipMemsetAsync(d_pass_any.p, 0, (size_t)kBurst * 4, s));
            for (int q = 0; q < kBurst; ++q) {
                const int *gate = q ? d_pass_any.as<int>() + (q - 1) : nullptr;
                const int from = cur ^ (q & 1), to = from ^ 1;
                hipLaunchKernelGGL(k_degree, dim3(deg_grid), dim3(256), 0, s, R, d_row_tint.as<int>(), d_tints.as<TintDesc>(),
                                   d_adj[from].as<u64>(), d_deg.as<int>(), gate);
                if (edge_walk)
                    hipLaunchKernelGGL(k_prune_edges, dim3((int)((R * edge_chunks + 3) / 4 < 65536 ? (R * edge_chunks + 3) / 4 : 65536)), dim3(256), 0, s, R, edge_chunks,
                                       d_row_tint.as<int>(), d_tints.as<TintDesc>(), d_adj[from].as<u64>(), d_deg.as<int>(), d_adj[to].as<u64>(),
                                       d_changed.as<int>() + (size_t)q * T, d_pass_any.as<int>() + q, gate);
                else {
                hipLaunchKernelGGL(k_deg1, dim3((n_words + 3) / 4 < 4096 ? (n_words + 3) / 4 : 4096), dim3(256), 0, s, n_words,
                                   d_word_tint.as<int2>(), d_tints.as<TintDesc>(), d_deg.as<int>(), d_deg1.as<u64>(), gate);
                hipLaunchKernelGGL(k_prune, dim3((int)((R + 3) / 4 < 16384 ? (R + 3) / 4 : 16384)), dim3(256), 0, s, R, d_row_tint.as<int>(),
                                   d_tints.as<TintDesc>(), d_tint_word0.as<i64>(), d_adj[from].as<u64>(), d_deg.as<int>(),
                                   d_deg1.as<u64>(), d_adj[to].as<u64>(), d_changed.as<int>() + (size_t)q * T, d_pass_any.as<int>() + q, gate);
                }
            }
            HIP_TRY(c, hipMemcpyAsync(h_any, d_pass_any.p, (size_t)kBurst * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(h_changed, d_changed.p, (size_t)kBurst * T * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            // the passes that ran: 0 .. first pass that removed nothing (inclusive); each of them wrote the other buffer
            int ran = 0;
            for (int q = 0; q < kBurst; ++q) { ++ran; if (!h_any[q]) { done = true; break; } }
            for (int q = 0; q < ran; ++q)
                if (rounds_out) for (int t = 0; t < T; ++t) if (h_changed[(size_t)q * T + t]) rounds_out[t] += 1;
            cur ^= ran & 1;
        }
    }
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    HIP_TRY(c, hipMemcpyAsync(adj_out, d_adj[cur].p, (size_t)n_adj * 8, hipMemcpyDeviceToHost, s));
    std::vector<int> small_rounds;
    if (prune && rounds_out && !small_tints.empty()) {
        small_rounds.resize((size_t)T);
        HIP_TRY(c, hipMemcpyAsync(small_rounds.data(), c->small_rounds.p, (size_t)T * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    if (!small_rounds.empty()) for (int t : small_tints) rounds_out[t] = small_rounds[(size_t)t];
    HIP_TRY(c, hipGetLastError());
    (void)hipEventElapsedTime(&c->compat_ms, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&c->prune_ms, c->ev[1], c->ev[2]);
    return FCLU_OK;
}

int fclu_last_timing(fclu_ctx *c, float *compat_ms, float *prune_ms) {
    if (!c) return FCLU_ERR_ARG;
    if (compat_ms) *compat_ms = c->compat_ms;
    if (prune_ms) *prune_ms = c->prune_ms;
    return FCLU_OK;
}

}  // extern "C"

#ifdef FREDDIE_SOURCE_HASH
/* what this binary was built from (freddie_amd/build.py looks for the marker in the file) */
static const char freddie_source_stamp[] __attribute__((used)) = "FREDDIE_SRC_HASH=" FREDDIE_SOURCE_HASH;
#endif
