// seg_common.h -- what the translation units of libfreddie_seg.so share: constants, the status record, the problem records,
// launch-configuration helpers (LDS sizes, threads per workgroup) and the small device functions (scans, flag words, label
// thresholds, the DP of a problem: dp_solve_push / dp_solve_wave).  Everything here is constexpr, a type, a template or an
// inline function; kernels live in the stage families' translation units (seg_front / seg_problems / seg_score_arena /
// seg_score_fused / seg_tail / seg_upload .hip), their declarations in seg_kernels.h, the host side in freddie_seg.hip.
// All citations `:N` are lines of the reference's py/freddie_segment.py.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "freddie_seg.h"

namespace fseg {


typedef long long i64;

typedef unsigned long long u64;


constexpr int kSmoothShift = 9;

constexpr int kSmoothTile = 1 << kSmoothShift;   // positions per smoothing tile: intervals average ~1.5 K positions in many-partition

                                                 // batches, and a tile never spans two intervals -- 512 keeps the tiles ~95 % full
constexpr int kSmoothThreads = kSmoothTile / 4;  // a thread computes 4 consecutive outputs

constexpr int kSumShift = 4, kSumBlock = 1 << kSumShift;     // positions per block of the histogram's in-tile prefix sums (k_smooth -> k_segments)

static_assert(kSmoothTile / kSumBlock <= 64 && kSumBlock == 16, "a tile's block sums are scanned by one wave; a block is four threads' positions");

constexpr int kMaxRadius = 200;        // sigma <= 50, truncate 4.0 (py/freddie_segment.py:106,:755)

constexpr int kScanBlock = 8192;       // elements per scan workgroup (256 threads x one 32-bit word of flags)

inline size_t flag_words(i64 n_pos) { return ((size_t)n_pos + 31) / 32 + 64; }      // words of one flag mask (+ room for a tile's last word and the scans' last block)

constexpr int kNMax = 60;              // largest DP problem handled by the LDS-resident scoring kernel

constexpr int kNHuge = 128;            // 60 < n <= 128 (max_problem_size up to ~115): the global-count-table kernels k_score_huge / k_dp_huge

                                       // (pair planes and the DP's tables in LDS)
constexpr int kNGiant = 1024;          // 128 < n <= 1024 (max_problem_size up to 1 000: what the CLI accepts): k_score_giant / k_dp_giant,

                                       // every per-pair table in global scratch -- slow, complete (the reference's own optimize() is
                                       // O(n^3 R) Python there: nobody runs it for long)
constexpr int kLaneChunk = 256;        // reads ("lanes") per scoring work item (u16 counters: must stay < 65536)

constexpr int kSub = 64;               // reads per scoring sub-chunk (two 32-bit plane words)

constexpr i64 kNegInf = (i64)(-0x7fffffffffffffffLL - 1);

constexpr i64 kKey32Reads = 1LL << 18;   // partitions of fewer reads take the DP's 32-bit keys (dp_solve_push / dp_solve_wave check the margin against NM)

constexpr int kFuseLanes = 255;   // reads a problem may see for 8-bit triple counters (four 64-read rounds at most)

constexpr int kFuseLanesDefault = 511;  // reads a problem may see for its batch to take the fused kernels (FSEG_FUSE_LANES): eight rounds

constexpr int kFuseLanesWide = 1023;   // ... and for the 16-bit instances: partitions of 1 000 reads have problems that see ~300 (one

                                       // in twenty-five of them more than 255); a problem that sees more than this is quicker spread
                                       // over the arena path's work items, and so is its whole batch

// error bits of Status::err
enum : unsigned {
    kErrExonInterval = 1u,     // an exon is not inside one tint interval (py/freddie_segment.py:668)
    kErrBreakAssert = 2u,      // break_large_problems: assert max_c_idx_y_v > 0 / index out of range (:640-643)
    kErrProblemTooLarge = 4u,  // a DP problem has more than kNGiant candidates
    kErrOverflowPairs = 8u,
    kErrOverflowTri = 16u,
    kErrOverflowWork = 32u,
    kErrOverflowLabels = 64u,
    kErrOverflowProblems = 128u,
    kErrOverflowChunks = 256u,
    kErrOverflowCov = 512u,
    kErrOverflowNm = 1024u,
    kErrWideMissed = 16384u,   // internal: a problem keeps more reads than k_prob_range counted for it (the 8-bit instance met it)
    kErrWaveStage = 8192u,     // a wave kernel (k_wave) met a read with more exons than its LDS stage holds: rerun without them
    kErrScanStall = 4096u,     // the look-back scan gave up waiting for a predecessor block: rerun with the three-pass scan
    kErrSyncTimeout = 32768u,  // a device-side waiter of the scoring stage (k_wait_word) gave up: the stage was skipped, rerun with events
    kErrNeedWideDp = 2048u,    // a problem sees >= 65536 reads: its DP needs the 32-bit count table    // a problem is larger than the LDS carve-up this launch was sized for
};


#ifdef FSEG_SCORE_TIMING
constexpr size_t kTaccProbs = 1u << 17, kTaccBytes = 128 + kTaccProbs * 64;     // (a record per problem for k_solve / k_wave, another for k_dpw)

#else
constexpr size_t kTaccBytes = 128;

#endif
struct Status {
    unsigned err;
    unsigned pad;
    u64 n_vals;        // number of Y > 0 values (all partitions)
    u64 n_vchunks;     // 8192-element chunks of the threshold reduction
    u64 n_cand;
    u64 n_prob;
    u64 n_work;
    u64 pair_used;
    u64 tri_used;
    u64 n_rseg;
    u64 n_final;
    u64 label_bytes;
    u64 cov_used;      // elements of the coverage arena
    unsigned max_n;    // largest DP problem of this run
    unsigned max_ln;   // most reads any DP problem of this run examines (>= 65536: the DP needs 32-bit counts)
    u64 cls_work[4];   // work items per problem-size class (n <= 16, <= 32, <= kNMax, <= kNHuge)
    u64 cls_queue[3];  // dynamic work counters of the scoring kernels
    u64 dp_cls[3];     // DP problems with n <= kDpSmall / <= kNMax / larger
    u64 cov_queue;
    u64 solve_cls[3];  // problems solved whole by k_solve (n <= 16 / <= 32 / <= kNMax)
    u64 n_tiny;        // problems solved whole by k_tiny (their list follows the three solve lists)
    unsigned list_cur[8];   // k_prob_emit's cursors into the four solve lists: [2 * list] from the front (expensive problems), [2 * list + 1] from the back
    unsigned wide_cls[4];   // solve-list problems per size class that see more than kFuseLanes reads (16-bit counters); [3] unused
    unsigned wide_cur[4];   // k_prob_emit's cursors into the per-class lists of those problems (wide_items); [3]: into the list of all of them (wide_all)
    unsigned gate_wide;     // workgroups of the large class's 16-bit instance that have started ('h' in a plan: the 8-bit instance waits for them)
    unsigned gate;          // large-class workgroups that have started (k_gate holds the small classes back until they are placed)
                            // (a counter of the mid class's 2 000 workgroups, bumped by each as it started, cost that kernel 10 of
                            // its 62 us: these two count a few hundred)
    unsigned sync_abort;    // a device-side waiter timed out: the scoring kernels behind it end at once (their input may not exist yet)
    unsigned pad2;
    // (Round 6 tried a start counter for the mid class here -- a gate 'i' that holds the small class back until the mid class's first
    // workgroups are placed.  Beside sync_abort, which every workgroup of the stage reads, eight counters took the stage from 0.125 to
    // 0.151 ms; on 128-byte lines of their own they cost nothing, and the gate gained nothing: 0.120-0.127 ms.  Removed.)
};

// Words of the device-side fork / join of the scoring stage (own allocation, zeroed once; generations only grow): see k_wait_word.
struct SyncWords {
    unsigned emit_gen;      // generation of the last scoring stage whose problem list is complete (published by the first launch behind
                            // k_prob_emit on the main stream): the side streams' waiters spin on it
    unsigned side_gen[4];   // generation of the last scoring stage whose chain on side stream k has ended (k_signal)
    unsigned emit_ctr;      // FSEG_EMIT_SIGNAL=1: workgroups of k_prob_emit that have finished (the last one publishes emit_gen and resets this)
    unsigned probe_word;    // probe_side_queues: generation published by a k_signal on the main stream ...
    unsigned probe_result;  // ... and what the side stream's k_probe_wait saw (1: the word, 2: its time limit)
};


// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }


// k_prob_emit's LAST workgroup publishes the stage's generation itself (FSEG_EMIT_SIGNAL=0: left to the first launch behind it on
// the main stream) -- the side streams start ~6 us earlier: config4 0.130 -> 0.123 ms, config3 0.148 -> 0.144, k_prob_emit as long as before.  Every workgroup pays one release to the device (its stores have reached L2: s_waitcnt +
// barrier; thread 0's release writes the XCD's dirty L2 lines back).  As __threadfence() in every wave it took the kernel from 17
// to 106 us (four write-backs AND four L2 invalidations per workgroup, in a kernel that lives on L2 hits).
__device__ __forceinline__ void emit_done(SyncWords *sw, unsigned gen) {
    if (!sw) return;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(&sw->emit_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);   // (acquire: the last workgroup's release of emit_gen then carries every workgroup's stores, not only its own)
        if (done == gridDim.x - 1) {
            __hip_atomic_store(&sw->emit_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sw->emit_gen, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// (asked beside the first descriptor load and tested behind it: no dependent load of its own in a problem's chain)
__device__ __forceinline__ unsigned stage_aborted(const Status *st) { return __hip_atomic_load(&st->sync_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }



// index of the last element of a[0..n) that is <= x  (a ascending, a[0] <= x)
template <typename T, typename X>
__device__ __forceinline__ i64 last_le(const T *a, i64 n, X x) {
    i64 lo = 0, hi = n;   // invariant: a[lo] <= x (if any), answer in [lo, hi)
    while (hi - lo > 1) {
        i64 mid = (lo + hi) >> 1;
        if ((X)a[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}


__device__ __forceinline__ i64 reflect_index(i64 i, i64 n) {
    if (n == 1) return 0;
    i64 p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}


// Integer forms of the reference's floating-point label tests (py/freddie_segment.py:490-495,
// :816-828): with c = cov/L in IEEE double, `c > h`  <=>  cov >= hi  and  `c < 1-h`  <=>  cov <= lo,
// where hi = min{v : fl(v/L) > h}, lo = max{v : fl(v/L) < fl(1-h)} (fl(v/L) is monotone in v).
__device__ __forceinline__ void label_thresholds(i64 L, const double *h_table, int h_len, double tau, int *hi_out,
                                                 int *lo_out) {
    double h = L < (i64)h_len ? h_table[L] : tau;   // get_high_threshold :269-274
    double l = 1.0 - h;
    double dL = (double)L;
    i64 v = (i64)floor(h * dL);
    if (v < 0) v = 0;
    while (v > 0 && (double)(v - 1) / dL > h) --v;
    while (!((double)v / dL > h) && v <= L + 1) ++v;
    *hi_out = (int)v;
    i64 u = (i64)ceil(l * dL);
    if (u > L) u = L;
    while (u >= 0 && !((double)u / dL < l)) --u;
    while ((double)(u + 1) / dL < l && u < L) ++u;
    *lo_out = (int)u;
}


// The bounds depend only on L (and the run's parameters): a table for the short segments, built once per parameter set,
// replaces the fp64 divisions in the kernels that evaluate them per problem (a DP window's pairs are mostly a few
// hundred positions apart).
constexpr int kThrTab = 8192;

__device__ __forceinline__ void label_thresholds_tab(i64 L, const int2 *tab, const double *h_table, int h_len, double tau,
                                                     int *hi_out, int *lo_out) {
    if (L < (i64)kThrTab) { const int2 t = tab[L]; *hi_out = t.x; *lo_out = t.y; }
    else label_thresholds(L, h_table, h_len, tau, hi_out, lo_out);
}


// ---------------------------------------------------------------------------------------------
// S1  splice histogram   (process_splicing_data, py/freddie_segment.py:648-678)
// One workgroup per chunk of kHistChunk consecutive positions of one partition.  The reads that can touch the
// chunk are a contiguous range of the position-sorted lane list (same two binary searches as k_prob_range); their
// exon ends falling into the chunk are counted in an LDS histogram (integer counts, so LDS atomics are
// order-free) which is then written out whole -- no global atomics and no memset of the histogram.
// A lane is one read (reps are repeated rep_weight times in the lane list), so every hit adds 1.
// ---------------------------------------------------------------------------------------------
constexpr int kHistChunk = 8192;

constexpr int kHistIv = 1024;      // intervals of a partition cached in LDS by k_hist


// ---------------------------------------------------------------------------------------------
// S2  Gaussian smoothing, fp64   (gaussian_filter1d(y, sigma, truncate=4.0), :755)
// out[l] = x[l]*w[0]; for j = radius..1: out += (x[l-j] + x[l+j]) * w[j]   -- farthest pair first,
// separate multiply and add (no FMA), 'reflect' boundary.  One workgroup per tile of positions; the
// tile plus its halo is staged in LDS as int32 (the histogram holds exact small integers).
// Also writes the flag Y > 0 used by the threshold stage.
// ---------------------------------------------------------------------------------------------
template <int NW = 0> __device__ __forceinline__ int wg_exclusive_scan(int v, int *lds, int *total);


// Workgroup barrier that orders LDS traffic only: global loads and stores issued before it (prefetches of the next
// tile, result stores) stay in flight, which a full __syncthreads() would wait for.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// wg_exclusive_scan() with LDS-only barriers.  NW: the workgroup's waves when the caller knows them (with the count read from
// blockDim the loop over the waves' totals is a general loop, unrolled sixteen-fold with masks: dozens of instructions for two values)
template <int NW = 0>
__device__ __forceinline__ int wg_exclusive_scan_lds(int v, int *lds /* >= 16 ints */, int *total) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = NW > 0 ? NW : (int)((blockDim.x + 63) >> 6);
    int x = v;
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    lds_barrier();
    if (lane == 63) lds[wave] = x;
    lds_barrier();
    int off = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        int sv = lds[w];
        if (w < wave) off += sv;
        tot += sv;
    }
    *total = tot;
    return off + x - v;
}


// R > 0: radius known at compile time (the tap loop is fully unrolled: no window moves, no loop control, weights
// in scalar registers); R == 0: any radius <= kMaxRadius.
// One 16-byte record per tile (built on upload) instead of tile -> interval -> offsets: the per-tile set-up is one load.
struct __align__(16) TileDesc {
    i64 base;       // position of the interval's first element (pos_off[interval])
    int y0;         // first position of the tile inside the interval
    int len;        // interval length
};

constexpr int kSmoothStage = (kSmoothTile + 2 * kMaxRadius + kSmoothThreads - 1) / kSmoothThreads;   // staged counts per thread, any radius


typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));      // (16 bytes from a dword-aligned address)


// ---------------------------------------------------------------------------------------------
// exclusive prefix sum of byte flags (three small kernels; used for the three compactions)
// ---------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ int wg_exclusive_scan(int v, int *lds /* >= 16 ints */, int *total) {
    int lane = lane_id(), wave = threadIdx.x >> 6;
    const int nw = NW > 0 ? NW : (int)((blockDim.x + 63) >> 6);
    int x = v;
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        int s = lds[w];
        if (w < wave) off += s;
        tot += s;
    }
    *total = tot;
    return off + x - v;
}


// The three flag sets of a run (Y > 0, candidate, final position) are BIT masks over the batch's positions: bit (p & 31) of word
// p >> 5 (round 4; a byte per position until then: 85 MB written by k_smooth per 250 k-read batch and read again by five scan
// launches).  Every thread of a scan owns the 32 positions of one word; i0 is a multiple of 32, positions at or beyond n count nothing.
typedef unsigned Flags32;

__device__ __forceinline__ Flags32 load_flags32(const unsigned *flags, i64 i0, i64 n) {
    unsigned w = flags[i0 >> 5];
    if (i0 + 32 > n) w &= n > i0 ? ((1u << (int)(n - i0)) - 1u) : 0u;
    return w;
}

__device__ __forceinline__ int count_flags32(Flags32 f) { return __popc(f); }

__device__ __forceinline__ void set_flag(unsigned *flags, i64 p) { atomicOr(&flags[p >> 5], 1u << (int)(p & 31)); }

// One workgroup of 1024 threads, eight block sums per thread per round (registers), so a batch's few thousand block sums are
// scanned in one round of one load, one workgroup scan and one store per thread (it was 256 threads x one element: a dozen
// latency-bound rounds, 12 us three times per run).
constexpr int kScan2Threads = 1024, kScan2Per = 8;

// few blocks: single-pass chained scan (decoupled look-back): block b publishes its flag count in state[b] as soon as it is
// known, then adds up its predecessors' words until it meets one that already holds an inclusive prefix.
// state word = (tag << 62) | value, tag 0 = empty (zeroed at the start of the run), 1 = block aggregate,
// 2 = inclusive prefix.  One workgroup per block: a block only ever waits for blocks with a smaller index, which
// the dispatcher started earlier.  Returns the exclusive prefix of the block (all threads); the last block also
// writes the grand total.
constexpr u64 kScanValueMask = (1ULL << 62) - 1ULL;

__device__ __forceinline__ i64 scan_lookback(u64 *state, i64 b, i64 nb, i64 agg, i64 *bcast /* LDS */, u64 *total_out,
                                             i64 *off_last, unsigned *err) {
    const int lane = lane_id();
    if (threadIdx.x < 64) {
        if (lane == 0)
            __hip_atomic_store(&state[b], ((b == 0 ? 2ULL : 1ULL) << 62) | (u64)agg, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        i64 excl = 0;
        if (b > 0) {
            i64 base = b - 1;                                        // lane l looks at block base - l
            // every spin is bounded: the chain is short (the host only picks this scan then) and all its blocks are
            // resident together, but nothing about dispatch order is guaranteed -- a stall is reported, not waited out
            for (int spins = 0;; ++spins) {
                if (spins > (1 << 20)) { if (lane == 0) atomicOr(err, kErrScanStall); break; }
                const i64 idx = base - lane;
                u64 sv = 2ULL << 62;                                 // before block 0: prefix 0
                if (idx >= 0) sv = __hip_atomic_load(&state[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned tag = (unsigned)(sv >> 62);
                const u64 m_pref = __ballot(tag == 2), m_empty = __ballot(tag == 0);
                u64 use;                                             // lanes whose value is added
                if (m_pref) {
                    const int first = __ffsll((long long)m_pref) - 1;
                    use = first == 63 ? ~0ULL : ((2ULL << first) - 1ULL);
                } else use = ~0ULL;
                if (m_empty & use) { __builtin_amdgcn_s_sleep(1); continue; }   // a predecessor has not published yet
                i64 v = ((use >> lane) & 1ULL) ? (i64)(sv & kScanValueMask) : 0;
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                excl += v;
                if (m_pref) break;
                base -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&state[b], (2ULL << 62) | (u64)(excl + agg), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            *bcast = excl;
            if (b == nb - 1) { *total_out = (u64)(excl + agg); if (off_last) *off_last = excl + agg; }
        }
    }
    __syncthreads();
    return *bcast;
}

// third pass fused with the consumer of the compaction:
//   kEmitValues:    v[rank] = y[i]                                   (threshold stage)
//   kEmitPositions: out_y[rank] = y index inside its interval, out_pos[rank] = genomic position,
//                   out_off[k] = rank of the interval's first position (always flagged)
enum { kEmitValues = 0, kEmitPositions = 1 };

// one 512-thread workgroup per chunk; pass 0 sums v, pass 1 sums (v-mean)^2.
// Thread (leaf, q) owns accumulator q of the 8-lane leaf of the pairwise recursion: r[q] = a[q] + a[8+q] +
// a[16+q] + ... in that order; the 8 accumulators are combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) by
// three xor-shuffles (IEEE addition is commutative, so both partners get identical bits), the tail (len%8)
// is added left to right, and the leaves are combined in recursion order.
struct VsumLds {
    // leaves of the pairwise recursion, left to right; a partial chunk's tree is kept in heap order (root 1,
    // children 2i / 2i+1; depth <= 7): node_kind 1 = leaf, 2 = inner node
    int leaf_off[128], leaf_len[128], leaf_heap[128];
    double node_val[256];
    unsigned char node_kind[256];
    int n_leaf_s, wave0_leaves;
};

// numpy's pairwise sum of ONE chunk (m <= 8192 values at a; pass 1: of (v - mu)^2) by a 512-thread workgroup; the result is
// valid in thread 0.  Starts with a barrier (the previous chunk is done with L).
__device__ __forceinline__ double vsum_chunk(const double *a, int m, int pass, double mu, VsumLds &L) {
    __syncthreads();
    if (m == 8192) {
        // perfect tree: 64 leaves of 128
        for (int t = threadIdx.x; t < 64; t += blockDim.x) { L.leaf_off[t] = t * 128; L.leaf_len[t] = 128; L.leaf_heap[t] = 64 + t; }
        if (threadIdx.x == 0) L.n_leaf_s = 64;
    } else {
        // every leaf but a lone one has at least 64 elements, so it holds exactly one x = 64 t with x - off < 64:
        // thread t walks the recursion (n2 = len/2 rounded down to a multiple of 8) down to the leaf of x
        if (threadIdx.x < 256) L.node_kind[threadIdx.x] = 0;
        __syncthreads();
        if (threadIdx.x < 128) {
            const int x = threadIdx.x * 64;
            int off = 0, len = m, h = 1;
            bool own = false;
            if (x < m) {
                while (len > 128) {
                    int n2 = len / 2; n2 -= n2 % 8;
                    if (x < off + n2) { len = n2; h = 2 * h; } else { off += n2; len -= n2; h = 2 * h + 1; }
                }
                own = x - off < 64;
            }
            const u64 mk = __ballot(own);
            if (threadIdx.x == 0) L.wave0_leaves = __popcll(mk);
            __syncthreads();
            if (own) {
                const int rank = __popcll(mk & ((1ULL << lane_id()) - 1ULL)) + (threadIdx.x >= 64 ? L.wave0_leaves : 0);
                L.leaf_off[rank] = off; L.leaf_len[rank] = len; L.leaf_heap[rank] = h;
                L.node_kind[h] = 1;
                for (int anc = h >> 1; anc >= 1; anc >>= 1) L.node_kind[anc] = 2;
            }
            if (threadIdx.x == 64) L.n_leaf_s = L.wave0_leaves + __popcll(mk);
        } else __syncthreads();
    }
    __syncthreads();
    int nl = L.n_leaf_s;
    const int q = threadIdx.x & 7;
#define FSEG_VAL(x) (pass ? __dmul_rn(__dsub_rn((x), mu), __dsub_rn((x), mu)) : (x))
    for (int t0 = 0; t0 < nl; t0 += 64) {
        int t = t0 + (threadIdx.x >> 3);
        double res = 0.0;
        if (t < nl) {
            const double *b = a + L.leaf_off[t];
            int len = L.leaf_len[t];
            if (len < 8) {
                for (int i = 0; i < len; ++i) res = __dadd_rn(res, FSEG_VAL(b[i]));      // from 0.0, left to right
            } else {
                int body = len - (len % 8);
                double x[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = (8 * i + q < body) ? b[8 * i + q] : 0.0;
                double r = FSEG_VAL(x[0]);
#pragma unroll
                for (int i = 1; i < 16; ++i) if (8 * i + q < body) r = __dadd_rn(r, FSEG_VAL(x[i]));
                r = __dadd_rn(r, __shfl_xor(r, 1));
                r = __dadd_rn(r, __shfl_xor(r, 2));
                r = __dadd_rn(r, __shfl_xor(r, 4));
                res = r;
                for (int i = body; i < len; ++i) res = __dadd_rn(res, FSEG_VAL(b[i]));
            }
        }
        if (t < nl && q == 0) L.node_val[L.leaf_heap[t]] = res;
    }
#undef FSEG_VAL
    __syncthreads();
    double out = 0.0;
    if (m == 8192) {
        // perfect tree over 64 leaves: adjacent pairs level by level = xor butterfly on one wave
        if (threadIdx.x < 64) {
            double x = L.node_val[64 + threadIdx.x];
            for (int d = 1; d < 64; d <<= 1) x = __dadd_rn(x, __shfl_xor(x, d));
            out = x;
        }
    } else {
        // inner nodes bottom-up, one tree level per step: sum(left) + sum(right)
        for (int lvl = 6; lvl >= 0; --lvl) {
            const int i = (1 << lvl) + threadIdx.x;
            if ((int)threadIdx.x < (1 << lvl) && L.node_kind[i] == 2) L.node_val[i] = __dadd_rn(L.node_val[2 * i], L.node_val[2 * i + 1]);
            __syncthreads();
        }
        if (threadIdx.x == 0) out = L.node_val[1];
    }
    return out;
}

// The whole threshold of a partition by ONE 512-thread workgroup (round 5; batches of many partitions of moderate size): the
// partition's Y > 0 values are compacted into its own piece of v (it starts where the partition's positions start: no batch-wide
// scan, no offsets), summed chunk by chunk in numpy's order (vsum_chunk, the same function the chunk kernel uses), the mean,
// the squared deviations likewise, the threshold.  One launch instead of seven to nine (k_scan1 / k_scan2 / k_scan_emit<values>,
// k_voff, k_vplan, k_vsum_chunks twice, k_vsum_part): 0.106 ms of launch-latency-sized pieces per 250 k-read batch.
constexpr int kThrPartMaxChunks = 128;     // chunk sums a workgroup keeps in LDS: partitions of up to 2^20 positions

// A wave compacts the flagged values of its 2048 positions w0 .. w0 + 2047 (fm: the flag word of lane l's 32 positions, w0 + 32 l ..)
// into v[ex ..) and returns their number.  Rows of 64 positions (lane = column), so the loads of y and the stores are coalesced;
// a row's 64 flags are the words of lanes 2q and 2q + 1 (two readlanes); rows without a flag are skipped (the values Y > 0 come
// in runs of 2 * radius + 1 around the splice sites) and the values of eight rows are loaded together from clamped addresses.
__device__ __forceinline__ int wave_emit_values(i64 w0, i64 n_pos, unsigned fm, i64 ex, const double *__restrict__ y, double *v) {
    const int lane = lane_id();
    const u64 lt_mask = (1ULL << lane) - 1ULL;
    u64 rows;
    {
        const u64 lanes_set = __ballot(fm != 0);                 // bit l: lane l's 32 positions hold a flag
        u64 pairs = (lanes_set | (lanes_set >> 1)) & 0x5555555555555555ULL;     // bit 2q: row q
        pairs = (pairs | (pairs >> 1)) & 0x3333333333333333ULL;
        pairs = (pairs | (pairs >> 2)) & 0x0f0f0f0f0f0f0f0fULL;
        pairs = (pairs | (pairs >> 4)) & 0x00ff00ff00ff00ffULL;
        pairs = (pairs | (pairs >> 8)) & 0x0000ffff0000ffffULL;
        rows = (pairs | (pairs >> 16)) & 0x00000000ffffffffULL;
    }
    int cnt = 0;
    while (rows) {
        u64 m[8];
        double yv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {                            // the next eight rows that hold flags
            const int q = rows ? (int)__builtin_ctzll(rows) : -1;
            rows = rows ? rows & (rows - 1) : 0;
            const int qq = q < 0 ? 0 : q;
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)fm, 2 * qq);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)fm, 2 * qq + 1);
            m[e] = q < 0 ? 0 : ((u64)lo | ((u64)hi << 32));
            const i64 i = w0 + qq * 64 + lane;
            yv[e] = y[i < n_pos ? i : n_pos - 1];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (!m[e]) continue;
            if ((m[e] >> lane) & 1ULL) v[ex + cnt + __popcll(m[e] & lt_mask)] = yv[e];
            cnt += __popcll(m[e]);
        }
    }
    return cnt;
}


// ---------------------------------------------------------------------------------------------
// S3b  candidates   (candidates_from_peaks :615-621; scipy _local_maxima_1d)
// strict local maxima with the plateau-midpoint rule, plus the first and last position.
// ---------------------------------------------------------------------------------------------
// edge[p]: bit 0 = p is the first position of its interval, bit 1 = the last one.  Built once per uploaded batch.
// What k_smooth's tiles could not decide about candidates_from_peaks (:615-621): one thread per tile looks, in the finished
// signal, at the tile's first position, its last position and the plateau start the tile deferred -- as possible STARTS of a
// peak (a strict maximum, or a plateau whose midpoint counts when it falls on its right; the walk stops at the interval's
// last position, as scipy's does).  Interval ends were flagged by the tiles.
__device__ __forceinline__ void peak_from(i64 p, i64 last /* the interval's last position */, const double *x, unsigned *flag) {
    const double xi = x[p];
    if (!(x[p - 1] < xi)) return;
    if (x[p + 1] < xi) { set_flag(flag, p); return; }
    if (x[p + 1] == xi) {
        i64 ia = p + 1;
        while (ia < last && x[ia] == xi) ++ia;
        if (x[ia] < xi) set_flag(flag, (p + ia - 1) / 2);     // plateau midpoint (positions of one interval are consecutive)
    }
}


// ---------------------------------------------------------------------------------------------
// S4  fixing, problem splitting, problem list
//   fixed = {0, N-1} U {c : Y[cand_c] > thr}                      py/freddie_segment.py:776-783
//   break_large_problems                                           :623-645 (pairs taken before insertion)
//   problems = consecutive fixed pairs                             :581
// One wave per tint interval, lane = candidate inside a 64-candidate tile; the "previous fixed
// candidate" of a lane comes from the tile's ballot mask or from the carry of earlier tiles.
// Problems with n == 2 have no (i,j,k) and can never add a breakpoint, so only n >= 3 are emitted.
// For each problem the kernel also finds the range of position-sorted reads that can overlap its
// window and carves its share of the arenas (one wave-aggregated atomic per arena and tile).
// ---------------------------------------------------------------------------------------------
struct ProblemArrays {
    int *iv;        // interval
    int *start;     // first candidate (index inside the interval)
    int *n;         // number of candidates
    i64 *pair_off;  // offset into the pair arenas (thresholds, ambiguity counts)
    i64 *tri_off;   // offset into the out-count arena
    int *flags;     // bit0: some pair has lo < 0 (a read with zero coverage is ambiguous there)
    int *chain;     // number of backtracked triples (debug)
    i64 *cov_off;   // offset into the coverage arena
    int *lane_lo;   // first lane (position-sorted read) that can overlap the problem's window
    int *lane_n;    // number of lanes examined: [lane_lo, lane_lo + lane_n)
};

// Everything the coverage / scoring / DP kernels need to know about a problem, in one 64-byte record: their
// per-item set-up is a chain of dependent loads, and one record load replaces three levels of it.
struct __align__(16) ProbDesc {
    i64 c0;         // global index of the problem's first candidate (cand_off[iv] + start)
    i64 pair_off, tri_off, cov_off;
    int n, lane_lo, lane_n;
    int g0;         // genomic start of the interval (iv_start[iv])
    int outside;    // lanes of the partition outside [lane_lo, lane_lo + lane_n)
    int iv;
    int w0;         // first work item (= chunk 0) of the problem
    int kind;       // kKindArena / kKindTiny / kKindFused: which kernels solve it
};

static_assert(sizeof(ProbDesc) == 64, "ProbDesc is one 64-byte record");

__device__ __forceinline__ ProbDesc load_desc(const ProbDesc *d) {
    const uint4 *q = reinterpret_cast<const uint4 *>(d);
    union { uint4 v[4]; ProbDesc p; } u;
    u.v[0] = q[0]; u.v[1] = q[1]; u.v[2] = q[2]; u.v[3] = q[3];
    return u.p;
}


// Eight (two) consecutive exon coordinates from a dword-aligned address as two 16-byte loads (one 8-byte load).  A lane that
// walks its own read's exons touches one or two cache lines per block whichever way it loads them, but the texture path works
// per instruction and lane: eight dword loads of 64 lanes are 512 line accesses, two 16-byte loads 128 -- and that rate, not
// HBM or the ALUs, is what the small problems' kernels run at.  The exon arrays are padded so that a block which starts at
// the batch's last exons stays inside them; elements beyond a read's own exons are masked by the callers.
typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));

typedef int int2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ void load_exons8(const int *p, int (&v)[8]) {
    const int4u a = *reinterpret_cast<const int4u *>(p), b = *reinterpret_cast<const int4u *>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

__device__ __forceinline__ int2 load_exons2(const int *p) {
    const int2u a = *reinterpret_cast<const int2u *>(p);
    return make_int2(a.x, a.y);
}

constexpr size_t kExonPad = 32;        // bytes behind ex_ts / ex_te

constexpr size_t kLexPad = 64;         // bytes behind the lane-ordered exon stream (it is read in aligned 16-byte units)


// A value every lane of the wave holds identically, moved to a scalar register: what is computed from it (triangular
// table offsets, loop bounds, LDS base addresses) then runs on the scalar unit instead of costing every lane a multiply.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int wave_id() { return uni((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ ProbDesc load_desc_uniform(const ProbDesc *d) {      // the whole wave loads the same record
    const uint4 *q = reinterpret_cast<const uint4 *>(d);
    union { uint4 v[4]; int w[16]; ProbDesc p; } u;
    u.v[0] = q[0]; u.v[1] = q[1]; u.v[2] = q[2]; u.v[3] = q[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) u.w[i] = uni(u.w[i]);
    return u.p;
}


#ifdef FSEG_DESC_VECTOR
#define FSEG_LOAD_DESC load_desc
#else
#define FSEG_LOAD_DESC load_desc_uniform
#endif

__device__ __forceinline__ i64 wave_excl_scan(i64 v, i64 *total) {
    int lane = lane_id();
    i64 x = v;
    for (int d = 1; d < 64; d <<= 1) {
        i64 y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    *total = __shfl(x, 63);
    return x - v;
}


// Workgroup-wide "previous flagged element": every thread holds one element (index idx, flag f) of a tile of
// blockDim.x consecutive elements; returns the index of the nearest flagged element before it (from this tile,
// else `carry`), and advances carry to the tile's last flagged element.  lds: >= 16 ints.
__device__ __forceinline__ int wg_prev_flagged(bool f, int idx, int &carry, int *lds) {
    int lane = lane_id(), wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    u64 mask = __ballot(f);
    __syncthreads();
    if (lane == 0) lds[wave] = mask ? idx + 63 - __clzll((long long)mask) : -1;
    __syncthreads();
    u64 below = mask & ((1ULL << lane) - 1ULL);
    int prev = carry;
    if (below) prev = idx - lane + 63 - __clzll((long long)below);
    else for (int w = wave - 1; w >= 0; --w) if (lds[w] >= 0) { prev = lds[w]; break; }
    for (int w = nw - 1; w >= 0; --w) if (lds[w] >= 0) { carry = lds[w]; break; }
    return prev;
}


// The exons of one read (ex = its range in the exon arrays) that meet the window [cp0, c_last): exons are ordered, so they
// are consecutive -- `cnt` of them from the read's `first_rel`-th.  k_solve keeps the reads with cnt > 0 (its rounds run over
// those only) and k_prob_range counts them ahead of it: the one definition of "keeps".
__device__ __forceinline__ void window_exons(const int2 *__restrict__ lex, int2 lx, int cp0, int c_last, int *first_rel, int *cnt_out) {
    // (the exons come from the lane-ordered (ts, te) stream: a read's exons are one contiguous piece of it and consecutive
    // lanes' pieces follow each other, so the lanes of a wave walk neighbouring cache lines -- from the rep-ordered ex_ts / ex_te
    // every lane's eight exons were two lines of their own, 128 line accesses per load instruction of a wave)
    const int n_ex = lx.y - lx.x;
    int fr = 0, cnt = 0;
    for (int eb = 0; eb < n_ex; eb += 8) {                   // eight exons per round (what lies beyond the read's own is masked; the stream is padded)
        int4u x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const int4u *>(lex + lx.x + eb + 2 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool hit0 = eb + 2 * u < n_ex && x[u].y >= cp0 && x[u].x < c_last;
            if (hit0 && cnt == 0) fr = eb + 2 * u;
            cnt += hit0;
            const bool hit1 = eb + 2 * u + 1 < n_ex && x[u].w >= cp0 && x[u].z < c_last;
            if (hit1 && cnt == 0) fr = eb + 2 * u + 1;
            cnt += hit1;
        }
        if (eb + 7 < n_ex && x[3].z >= c_last) break;        // the rest of the read lies beyond the window
    }
    *first_rel = fr; *cnt_out = cnt;
}


// For the right end of every problem: the range of position-sorted reads that can overlap the problem's window
// [g0, g1): reads are sorted by first position, lane_pmax is the running maximum of their last position.
// ... and, for a problem that sees more than kFuseLanes reads (8-bit counters hold 255), whether it KEEPS more than that
// many -- reads with an exon in the window, about two thirds of those it sees: cand_wide[c] = 1 sends it to the 16-bit
// instances of k_solve (kKindFusedWide in its record).  Such candidates are rare and expensive (up to kFuseLanesWide reads
// each): the workgroup collects its own in LDS and goes over them together, a read per thread.  (Round 4 until here: both
// instances located every such problem and counted its kept reads, the 16-bit ones to drop nearly all of them again --
// launches of 12-70 us in front of the classes behind them on config3 / config5.)
constexpr int kRangeThreads = 1024;    // = kProbBlock: a workgroup of k_prob_range is a block of the problem scan


// Problem list by a prefix sum over the candidates: problem slot, pair / triple / coverage arena offsets and
// work items come out in candidate order, so the arena layout is deterministic.
// Scanned columns.  Counters that stay below 2^32 over a whole batch share a 64-bit column (low | high << 32):
//   0: problem slot | DP problems of the small class     1: pairs     2: triples     3: coverage elements
//   4: work items of class 0 | class 1                   5: work items of class 2 | DP problems of the big class
//   6: work items of class 3 (huge) | DP problems of the huge class
//   7: fused problems of class 0 | class 1                   8: fused problems of class 2 | k_tiny's problems
// (work items overall = the four class counts)
constexpr int kProbCols = 9;

__device__ __forceinline__ i64 col_lo(i64 x) { return x & 0xffffffffLL; }

__device__ __forceinline__ i64 col_hi(i64 x) { return (i64)((u64)x >> 32); }

constexpr int kDpSmall = 32;

constexpr int kClsSmall = 16, kClsMid = 32;

struct ProbSizes { i64 v[kProbCols]; };

__device__ __forceinline__ int size_class(int n) { return n <= kClsSmall ? 0 : (n <= kClsMid ? 1 : (n <= kNMax ? 2 : 3)); }

// How the problems of a run are divided among the three ways of solving them:
//   n <= tiny_max (> 0 in batches of many problems): whole by k_tiny, one wave each; a problem slot and nothing else;
//   otherwise, lanes <= fuse_lanes and n <= kNMax: whole by k_solve, one workgroup each (coverage, pair labels, counts and
//     DP without leaving LDS); a problem slot and an entry in its size class's solve list;
//   otherwise: the arena path -- coverage tiles, one scoring work item per 256 reads, global count tables, DP kernels
//     (problems that see many reads need many workgroups to score them).
//   with the wave kernels (k_wave: the batch has its exon stream) the small class -- n <= wave_n -- is solved whole, one wave
//     per problem, whenever the problem sees at most wave_lanes reads, whatever the rest of the batch looks like.
struct ProbSplit { int tiny_max, fuse_lanes; };

enum { kKindArena = 0, kKindTiny = 1, kKindFused = 2, kKindFusedWide = 3 };   // (kKindFusedWide: a record's kind only -- a fused problem that KEEPS more than kFuseLanes reads, k_prob_range)

__device__ __forceinline__ int prob_kind(int n, int n_lanes, ProbSplit sp) {
    if (n <= sp.tiny_max) return kKindTiny;
    return (n_lanes <= sp.fuse_lanes && n <= kNMax) ? kKindFused : kKindArena;
}

__device__ __forceinline__ ProbSizes prob_sizes(int n, int n_lanes, ProbSplit sp) {
    ProbSizes s;
    for (int q = 0; q < kProbCols; ++q) s.v[q] = 0;
    if (n <= 0) return s;
    const int kind = prob_kind(n, n_lanes, sp);
    if (kind == kKindTiny) { s.v[0] = 1; s.v[8] = 1LL << 32; return s; }
    if (kind == kKindFused) {
        const int c = size_class(n);
        s.v[0] = 1;
        s.v[7] = c == 0 ? 1 : (c == 1 ? (1LL << 32) : 0);
        s.v[8] = c == 2 ? 1 : 0;
        return s;
    }
    i64 chunks = (n_lanes + kLaneChunk - 1) / kLaneChunk;
    const int cls = size_class(n);
    s.v[0] = 1 + (n <= kDpSmall ? (1LL << 32) : 0);
    s.v[1] = (i64)n * (n - 1) / 2; s.v[2] = (i64)n * (n - 1) * (n - 2) / 6;
    s.v[3] = chunks * kLaneChunk * n;
    s.v[4] = cls == 0 ? chunks : (cls == 1 ? chunks << 32 : 0);
    s.v[5] = (cls == 2 ? chunks : 0) + ((n > kDpSmall && n <= kNMax) ? (1LL << 32) : 0);
    s.v[6] = (cls == 3 ? chunks : 0) + (n > kNMax ? (1LL << 32) : 0);
    return s;
}

__device__ __forceinline__ i64 wg_exclusive_scan64(i64 v, i64 *lds /* >= 16 */, i64 *total) {
    int lane = lane_id(), wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    i64 x = v;
    for (int d = 1; d < 64; d <<= 1) {
        i64 y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    i64 off = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        i64 sv = lds[w];
        if (w < wave) off += sv;
        tot += sv;
    }
    *total = tot;
    return off + x - v;
}

constexpr int kProbBlock = 1024;   // candidates per workgroup of the problem scan (256 threads x 4)

// exclusive scan of kProbCols columns over the 256 threads of a workgroup (two barriers for all columns)
__device__ __forceinline__ void wg_scan_cols(const ProbSizes &v, ProbSizes &ex, ProbSizes &tot, i64 *lds /* 4 * kProbCols */) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int q = 0; q < kProbCols; ++q) {
        i64 x = v.v[q];
        if (__ballot(x != 0)) {                          // (a batch fills the arena path's columns or the solve lists', rarely both:
            for (int d = 1; d < 64; d <<= 1) {           //  the wave skips the columns in which it holds nothing)
                i64 y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
        }
        ex.v[q] = x - v.v[q];
        if (lane == 63) lds[wave * kProbCols + q] = x;
    }
    __syncthreads();
    for (int q = 0; q < kProbCols; ++q) {
        i64 off = 0, t = 0;
        for (int w = 0; w < 4; ++w) {
            i64 sv = lds[w * kProbCols + q];
            if (w < wave) off += sv;
            t += sv;
        }
        ex.v[q] += off; tot.v[q] = t;
    }
    __syncthreads();
}

__device__ __forceinline__ ProbSizes prob_block_sizes(const int *cand_pn, const int *cand_ln, i64 b, i64 n, ProbSizes *per_elem /* 4, may be null */, ProbSplit sp) {
    ProbSizes acc;
    for (int q = 0; q < kProbCols; ++q) acc.v[q] = 0;
    const i64 i0 = b * kProbBlock + (i64)threadIdx.x * 4;
    for (int e = 0; e < 4; ++e) {
        ProbSizes sz = i0 + e < n ? prob_sizes(cand_pn[i0 + e], cand_ln[i0 + e], sp) : prob_sizes(0, 0, sp);
        if (per_elem) per_elem[e] = sz;
        for (int q = 0; q < kProbCols; ++q) acc.v[q] += sz.v[q];
    }
    return acc;
}

__device__ __forceinline__ void prob_store_totals(Status *st, const ProbSizes &t) {
    st->n_prob = (u64)col_lo(t.v[0]); st->pair_used = (u64)t.v[1]; st->tri_used = (u64)t.v[2];
    st->n_work = (u64)(col_lo(t.v[4]) + col_hi(t.v[4]) + col_lo(t.v[5]) + col_lo(t.v[6])); st->cov_used = (u64)t.v[3];
    st->cls_work[0] = (u64)col_lo(t.v[4]); st->cls_work[1] = (u64)col_hi(t.v[4]); st->cls_work[2] = (u64)col_lo(t.v[5]);
    st->cls_work[3] = (u64)col_lo(t.v[6]);
    st->dp_cls[0] = (u64)col_hi(t.v[0]); st->dp_cls[1] = (u64)col_hi(t.v[5]); st->dp_cls[2] = (u64)col_hi(t.v[6]);
    st->solve_cls[0] = (u64)col_lo(t.v[7]); st->solve_cls[1] = (u64)col_hi(t.v[7]); st->solve_cls[2] = (u64)col_lo(t.v[8]);
    st->n_tiny = (u64)col_hi(t.v[8]);
}

// Largest problem (candidates) and widest problem (reads examined) of the run: they size the big-problem kernels' LDS
// and pick the DP's count width.  One atomic per block of 1024 candidates -- per-problem atomics on the one address
// serialise (~90 per us).  l_mx: 8 ints of LDS; the caller's next barrier orders them.
__device__ __forceinline__ void prob_block_maxima(Status *st, const int *cand_pn, const int *cand_ln, i64 b, i64 n, int *l_mx) {
    int mx = 0, ml = 0;
    for (int e = 0; e < 4; ++e) {
        const i64 cc = b * kProbBlock + (i64)threadIdx.x * 4 + e;
        if (cc < n) { const int pn = cand_pn[cc]; mx = max(mx, pn); if (pn > 0) ml = max(ml, cand_ln[cc]); }
    }
    for (int d = 32; d >= 1; d >>= 1) { mx = max(mx, __shfl_xor(mx, d)); ml = max(ml, __shfl_xor(ml, d)); }
    if (lane_id() == 0) { l_mx[threadIdx.x >> 6] = mx; l_mx[4 + (threadIdx.x >> 6)] = ml; }
}

// solve-list problems of the block that keep more than 255 reads (k_prob_range), per size class (they need the 16-bit-counter
// instances of k_solve: the host launches those only for classes that have any)
__device__ __forceinline__ void prob_block_wide(Status *st, const int *cand_pn, const int *cand_ln, const unsigned char *cand_wide, i64 b, i64 n, ProbSplit sp) {
    unsigned w = 0;                                            // one count per byte: class 0 | class 1 << 8 | class 2 << 16
    for (int e = 0; e < 4; ++e) {
        const i64 cc = b * kProbBlock + (i64)threadIdx.x * 4 + e;
        if (cc < n) {
            const int pn = cand_pn[cc], ln = pn > 0 ? cand_ln[cc] : 0;
            if (pn > 0 && cand_wide[cc] && prob_kind(pn, ln, sp) == kKindFused) w += 1u << (8 * size_class(pn));
        }
    }
    if (__ballot(w != 0) == 0) return;
    unsigned c0 = w & 255u, c1 = (w >> 8) & 255u, c2 = (w >> 16) & 255u;
    for (int d = 32; d >= 1; d >>= 1) { c0 += __shfl_xor(c0, d); c1 += __shfl_xor(c1, d); c2 += __shfl_xor(c2, d); }
    if (lane_id() == 0) {
        if (c0) atomicAdd(&st->wide_cls[0], c0);
        if (c1) atomicAdd(&st->wide_cls[1], c1);
        if (c2) atomicAdd(&st->wide_cls[2], c2);
    }
}

__device__ __forceinline__ void prob_publish_maxima(Status *st, const int *l_mx) {
    const int mx = max(max(l_mx[0], l_mx[1]), max(l_mx[2], l_mx[3])), ml = max(max(l_mx[4], l_mx[5]), max(l_mx[6], l_mx[7]));
    if (mx > 0 && (unsigned)mx > __hip_atomic_load(&st->max_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->max_n, (unsigned)mx);
    if (ml > 0 && (unsigned)ml > __hip_atomic_load(&st->max_ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->max_ln, (unsigned)ml);
}

// bs == nullptr: every workgroup adds up the blocks before it itself (and all of them for the class bases) -- one
// launch instead of three while the candidate list is a handful of blocks (the host picks the mode from the
// previous run; either is correct for any size).
constexpr int kProbDirect = 4;     // work items a problem's own thread writes itself; longer lists are written by the workgroup


// pair index q = j*(j-1)/2 + i (i < j);  triple rank = k*(k-1)*(k-2)/6 + j*(j-1)/2 + i (i < j < k)
__device__ __forceinline__ void pair_decode(int q, int *i, int *j) {
    int jj = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)q)) * 0.5f);
    while (jj * (jj - 1) / 2 > q) --jj;
    while ((jj + 1) * jj / 2 <= q) ++jj;
    *j = jj;
    *i = q - jj * (jj - 1) / 2;
}


// S5a  integer label thresholds of every candidate pair of every problem (:490-495)
__device__ __forceinline__ void pair_thresholds_blocks(i64 first, i64 stride, const Status *st, ProblemArrays pr, const ProbDesc *desc,
                                                       i64 prob_cap, const int *cand_y, const double *h_table, int h_len, double tau,
                                                       int2 *pair_thr, i64 pair_cap, unsigned *amb_g, unsigned *out_g, i64 tri_cap) {
    i64 n_prob = (i64)st->n_prob < prob_cap ? (i64)st->n_prob : prob_cap;
    for (i64 p = first; p < n_prob; p += stride) {
        const ProbDesc d = load_desc(desc + p);
        int n = d.n;
        if (d.kind != kKindArena) continue;                 // solved whole by k_tiny / k_solve: owns nothing in the arenas
        i64 poff = d.pair_off;
        int npairs = n * (n - 1) / 2;
        if (poff + npairs > pair_cap) continue;
        const int *cy = cand_y + d.c0;
        int any_neg = 0;
        for (int q = threadIdx.x; q < npairs; q += blockDim.x) {
            int i, j;
            pair_decode(q, &i, &j);
            i64 L = (i64)cy[j] - cy[i] + 1;
            int hi, lo;
            label_thresholds(L, h_table, h_len, tau, &hi, &lo);
            pair_thr[poff + q] = make_int2(hi, lo);
            amb_g[poff + q] = 0;
            if (lo < 0) any_neg = 1;
        }
        i64 toff = d.tri_off;
        int ntri = n * (n - 1) * (n - 2) / 6;
        if (toff + ntri <= tri_cap) for (int x = threadIdx.x; x < ntri; x += blockDim.x) out_g[toff + x] = 0;
        if (any_neg) atomicOr(&pr.flags[p], 1);
    }
}


// ---------------------------------------------------------------------------------------------
// S5  interval scoring   (optimize(): pair labels :488-497, inside :500-506, outside :509-528)
//
// Work item = (problem, chunk of <= kLaneChunk reads of the problem's read range).  For 64 reads at a
// time the workgroup
//   A. stages the coverage prefixes cov[r][j] of the 64 reads in LDS;
//   B. evaluates every pair (i,j): yea = cov_j-cov_i >= hi_ij, nay = cov_j-cov_i <= lo_ij, shifting the
//      64 results into two 32-bit plane words per label, kept in LDS as {yea0,yea1,nay0,nay1};
//      ambiguous reads (neither) are counted per pair for inside(i,j) = -sum(W*amb);
//   C. for every triple i<j<k that the DP can use adds
//      popc(yea_ij & nay_jk) + popc(nay_ij & yea_jk)  (the two conjunctions are disjoint, :515-523)
//      into a u16 counter table in LDS.
// At the end of the work item the non-zero counters go to the global table with one atomic each.
// Reads with multiplicity W are expanded into W lanes on upload, so every lane has weight 1.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// S5c  DP over one problem   (dp() :532-558, top level :560-566, backtrack :592-594)
// D(a,b,c) = in_ab + out_abc + M(b,c),  M(b,c) = max_{c2>c} D(b,c,c2) (first maximiser, strict >),
// M(b,end) := in_b,end closes the chain (base case :545-548).  The inner maximum depends only on (b,c), so
// filling M for c descending is O(n^3) and gives the reference's O(n^4) recursion's result
// (SURVEY.md App. A.7).  All tables live in LDS:
//   out_s[rank(a,b,c)] counts, in_s[pair] = -(ambiguous reads), M / A (argmax) per pair, cy_s = candidate y.
// Every thread of the workgroup must call it; returns the number of backtracked triples (valid on thread 0)
// and marks the chosen candidates.
// ---------------------------------------------------------------------------------------------
#ifdef FSEG_SCORE_TIMING
#define FSEG_DTICK(i) do { unsigned long long t_now = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&dp_tacc[i], t_now - dt_prev); dt_prev = t_now; } while (0)
#define FSEG_DPARAM , unsigned long long *dp_tacc, unsigned long long &dt_prev
#define FSEG_DARG , dp_tacc, dt_prev
#else
#define FSEG_DTICK(i)
#define FSEG_DPARAM
#define FSEG_DARG
#endif

// T == 64: the caller is ONE WAVE working on its own problem with wave-private tables (other waves of the workgroup
// may be inside their own dp_solve_push<64>), so synchronisation is wave-level and thread indices are lane indices.
// Either way only LDS traffic is ordered (the tables are in LDS): global loads issued before it -- the next phase's
// prefetches -- stay in flight, which a full fence would wait for.
template <int T>
__device__ __forceinline__ void dp_sync() {
    if (T == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    } else {
        lds_barrier();
    }
}

// V: the type the sums are kept in.  i64 in general; int where the caller knows that n * (reads of the partition) stays
// below 2^30 (k_solve's launches: every sum is a chain of at most n/2 counts and ambiguity terms, each bounded by the
// partition's reads) -- half the registers, LDS traffic and instructions of the chain.
template <typename V> __device__ __forceinline__ constexpr V dp_neg_inf() { return sizeof(V) == 8 ? (V)kNegInf : (V)(-0x40000000); }

// Pushed, not pulled: every thread OWNS pairs (b,c) -- pair q = s*T + tid, the ownership the scoring phase already uses --
// and keeps their running maximum in registers:
//   column c2 final  ->  its owners write M(.,c2)  ->  one barrier  ->  every pair (b,c) with c < c2 takes
//   out(b,c,c2) + M(c,c2) into its maximum.
// One barrier per candidate, no reduction over waves, no serial part, and the work of a step is spread over all the
// threads (out(.,.,c2) is one contiguous run of the count table: lane-consecutive bytes).  Pushes arrive with c2
// descending, so "first maximiser" (the smallest c2 among equals) is "the later push wins ties".  The row b = 0 is the top
// level (:560-566): M(0,j) = in(0,j) + max_k(out(0,j,k) + M(j,k)), then the first maximiser over j, taken only if it beats
// "no cut" = in(0,end).  (A gather formulation -- lanes = b, a loop over c2 per lane, blocks of four candidates with a serial
// in-block fix-up -- took 36 us for n = 49 against 22 us; tools/probes/dp_probe.hip.)
// A running maximum and its argument are ONE integer, the key  value * 64 + (63 - c2):  the larger value wins, among equal
// values the smaller c2 (the reference's first maximiser, :526-527), and the update of a pair is one v_max.  A pair's final
// M(b,c) is stored in the same form with (63 - c) in the low bits -- the tie-break it needs when it is the tail of a push
// from column c, and at the top level (first maximiser over j) -- or kKeyNone.  With 32-bit keys every |value| must stay
// below 2^24: k_solve / k_wave take them when the largest partition has fewer than kKey32Reads = 2^18 reads (a chain has at most
// NM - 1 <= 63 links of at most that many reads each: 63 * 2^18 < 2^24, checked at compile time), the 64-bit instances otherwise.
// f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>): a loop whose index is a compile-time constant in the body
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

// f(integral_constant<int, I>) ... f(integral_constant<int, 0>)
template <int I, typename F> __device__ __forceinline__ void static_for_down(F &&f) {
    f(std::integral_constant<int, I>{});
    if constexpr (I > 0) static_for_down<I - 1>(f);
}

template <typename V> __device__ __forceinline__ constexpr V dp_key_none() { return sizeof(V) == 8 ? (V)(-(1LL << 62)) : (V)(-0x7ff00000); }

template <typename V> __device__ __forceinline__ constexpr V dp_key_min() { return sizeof(V) == 8 ? (V)(-(1LL << 61)) : (V)(-0x40000000); }   // every key of a value is above it

// The push of column c2 into the first NS slots of a thread: all the slots' LDS loads first (none of them under a branch),
// one wait, then four instructions per pair.  Only the last of a wave's live slots can hold pairs at or beyond the column
// (q >= t2): it is the one that is masked.
template <int NS, int T, int SLOTS, typename OutT, typename V>
__device__ __forceinline__ void dp_push_slots(int tid, int t2, int t3, int support, const V *M, const OutT *out_s,
                                              const int (&pc)[SLOTS], V (&best)[SLOTS]) {
    constexpr V kNone = dp_key_none<V>();
    V tail[NS];
    unsigned o[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int q = s * T + tid;
        const bool act = s < NS - 1 || q < t2;
        tail[s] = M[t2 + (act ? pc[s] : 0)];
        o[s] = (unsigned)out_s[t3 + (act ? q : 0)];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int q = s * T + tid;
        const bool act = s < NS - 1 || q < t2;
        const V key = (V)o[s] * 64 + tail[s];                        // (a tail of kKeyNone stays below every key of a value)
        const bool ok = act & ((int)o[s] >= support);                // :540
        const V k2 = ok ? key : kNone;
        best[s] = k2 > best[s] ? k2 : best[s];
    }
}

// ... for the `ns` (wave-uniform) slots of this wave that still hold pairs below the column
template <int NS, int T, int SLOTS, typename OutT, typename V>
__device__ __forceinline__ void dp_push_dispatch(int ns, int tid, int t2, int t3, int support, const V *M, const OutT *out_s,
                                                 const int (&pc)[SLOTS], V (&best)[SLOTS]) {
    if (ns >= NS) dp_push_slots<NS, T, SLOTS>(tid, t2, t3, support, M, out_s, pc, best);
    else if constexpr (NS > 1) dp_push_dispatch<NS - 1, T, SLOTS>(ns, tid, t2, t3, support, M, out_s, pc, best);
}

// Pushed, not pulled: every thread OWNS pairs (b,c) -- pair q = s*T + tid, the ownership the scoring phase already uses --
// and keeps their running maximum in registers:
//   column c2 final  ->  its owners write M(.,c2)  ->  one barrier  ->  every pair (b,c) with c < c2 takes
//   out(b,c,c2) + M(c,c2) into its maximum.
// One barrier per candidate, no reduction over waves, no serial part, and the work of a step is spread over all the
// threads (out(.,.,c2) is one contiguous run of the count table: lane-consecutive bytes).  The row b = 0 is the top
// level (:560-566): M(0,j) = in(0,j) + max_k(out(0,j,k) + M(j,k)), then the first maximiser over j, taken only if it beats
// "no cut" = in(0,end).  A step is a chain -- owners' write, barrier, loads, update -- and the problem's DP is n of them in
// a row, so what counts is the number of dependent instructions in a step (tools/probes/dp_probe.hip: n = 49 took 19 us with
// value and argument kept apart and compare / select through the scalar unit; the bare write-barrier-read is 73 ns).
// (A gather formulation -- lanes = b, a loop over c2 per lane, blocks of four candidates with a serial in-block fix-up --
// took 36 us for n = 49.)
// IN_DEAD: in_s[] is a hand-over of k_solve<.., SPLIT> -- kDeadPair marks the pairs whose segment is too small (:540) and cy_s is not
// looked at (k_dpw: the candidates' positions are not among what a problem hands over).
template <int T, int NM, typename OutT, typename V, bool IN_DEAD = false>
__device__ __forceinline__ int dp_solve_push(int n, const OutT *out_s, const int *in_s, V *M, unsigned char *A, const int *cy_s, int support,
                                             unsigned char *chosen /* + first candidate of the problem */ FSEG_DPARAM) {
    constexpr int SLOTS = (NM * (NM - 1) / 2 + T - 1) / T;
    constexpr int LOG2T = T == 64 ? 6 : (T == 128 ? 7 : (T == 256 ? 8 : (T == 512 ? 9 : 10)));
    static_assert((1 << LOG2T) == T, "T is a power of two from 64 to 1024");
    static_assert(NM <= 64, "the top level is one lane per candidate; an argument is six bits of a key");
    // 32-bit keys: value * 64 + argument with |value| < 2^24 -- a chain has at most NM - 1 links of at most kKey32Reads reads each
    static_assert(sizeof(V) == 8 || (i64)(NM - 1) * kKey32Reads < (1LL << 24), "32-bit DP keys: the longest chain's sum must stay below 2^24");
    const int lane = lane_id(), wave = T == 64 ? 0 : wave_id();
    const int tid = T == 64 ? lane : (int)threadIdx.x;
    n = uni(n); support = uni(support);
    const int end = n - 1;
    const int npairs = n * (n - 1) / 2;
    constexpr V kNone = dp_key_none<V>(), kMin = dp_key_min<V>();
    int pc[SLOTS];
    V best[SLOTS], inv[SLOTS];
    bool live[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int q = s * T + tid;
        int b, c;
        pair_decode(q < npairs ? q : 0, &b, &c);
        pc[s] = c;
        const int in_q = in_s[q < npairs ? q : 0];
        if constexpr (IN_DEAD) live[s] = q < npairs && in_q != (int)0x80000000;    // (kDeadPair)
        else live[s] = q < npairs && cy_s[c] - cy_s[b] >= 5;       // "segment too small" (:540)
        inv[s] = (V)in_q;
        best[s] = (q < npairs && c == end) ? (V)0 : kNone;         // M(b,end) = in(b,end): the chain's last link (:545-548)
    }
    int t2 = end * (end - 1) / 2, t3 = end * (end - 1) * (end - 2) / 6;
    for (int c2 = end; c2 >= 1; --c2) {
        // column c2 is final: its pairs are q in [t2, t2 + c2), at most one of them this thread's
        {
            const int d0 = (tid - t2) & (T - 1);
            if (d0 < c2) {
                const int q0 = t2 + d0;
                auto finish = [&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    const V bb = best[s];
                    const bool ok = live[s] && bb > kMin;
                    const V val = (bb >> 6) + inv[s];
                    M[q0] = ok ? (V)(val * 64 + (V)(63 - c2)) : kNone;
                    A[q0] = ok ? (unsigned char)(63 - (int)(bb & 63)) : (unsigned char)255;
                };
                if constexpr (SLOTS == 1) finish(std::integral_constant<int, 0>{});
                else if constexpr (T == 64) {                        // (a wave's owners may sit in two slots)
                    static_for<0, SLOTS>([&](auto sc) { if ((q0 >> LOG2T) == decltype(sc)::value) finish(sc); });
                } else {                                             // the owners of a wave share the slot
                    const int so = uni(q0 >> LOG2T);
                    static_for<0, SLOTS>([&](auto sc) { if (so == decltype(sc)::value) finish(sc); });
                }
            }
        }
        dp_sync<T>();
        if (c2 == 1) break;
        // pairs (b, c) with c < c2: q < t2, a prefix of the pair order -- the slots of this wave that reach below t2 come first
        const int wave_q0 = T == 64 ? 0 : wave * 64;
        const int ns = t2 > wave_q0 ? uni((t2 - wave_q0 + T - 1) >> LOG2T) : 0;
        dp_push_dispatch<SLOTS, T, SLOTS>(ns < SLOTS ? ns : SLOTS, tid, t2, t3, support, M, out_s, pc, best);
        t2 -= c2 - 1; t3 -= t2;
    }
    FSEG_DTICK(10);
    int chain = 0;
    if (wave == 0) {
        // first maximiser over j of M(0,j) (larger value, then smaller j): the largest key, one candidate per lane
        const int j0 = lane >= 1 && lane < end ? lane : 1;
        V kv = (lane >= 1 && lane < end) ? M[j0 * (j0 - 1) / 2] : kNone;
        for (int d = 32; d >= 1; d >>= 1) {
            const V ov = __shfl_xor(kv, d);
            kv = ov > kv ? ov : kv;
        }
        FSEG_DTICK(11);
        // the chain is walked by one lane (dependent LDS loads only: the chosen candidates are collected in a mask) and
        // stored by the wave, one candidate per lane
        // (every lane holds the same key after the reduction: the walk is the whole wave's, on scalar registers -- a link is one
        // LDS byte and a few scalar instructions; walked by lane 0 alone under an execution mask it was 113 ns per link)
        u64 mask = 0;
        const bool cut = end >= 2 && kv > kMin && (kv >> 6) > (V)in_s[end * (end - 1) / 2];
        if (uni(cut ? 1 : 0)) {
            const int bj = uni(63 - (int)(kv & 63));
            int j = bj, k = uni((int)A[bj * (bj - 1) / 2]);
            mask = 1ULL;
            for (;;) {
                mask |= (1ULL << j) | (1ULL << k); ++chain;
                if (k == end) break;
                const int k2 = uni((int)A[k * (k - 1) / 2 + j]);
                if (k2 == 255) break;
                j = k; k = k2;
            }
        }
        if ((mask >> lane) & 1ULL) chosen[lane] = 1;
    }
    FSEG_DTICK(12);
    return chain;
}


// ---------------------------------------------------------------------------------------------
// The same DP by ONE WAVE (k_solve's tail, round 4).  A column of the push DP is a chain -- the owners' write, the
// loads of the others, a handful of arithmetic -- and with T threads every wave pays the chain's ~50 instructions
// for a pair or two each, plus a workgroup barrier per column, while seven of the large class's eight waves hold
// their registers for nothing (the DP was 42-45 % of a problem's time).  One wave holds every pair (slot s = pair
// s * 64 + lane), needs no barrier (a wave's LDS operations complete in order) and lets the workgroup's other waves
// END when the scoring rounds are over: their registers and wave slots go to the next workgroup while this one
// finishes on a sixteenth of what it held.
//   * The columns are visited in STAGES: stage S = the columns whose first pair lies in slot S (t2 >> 6 == S), S
//     descending.  Inside a stage the slots that finish (S and S + 1) and the slots that take the push (0 .. S) are
//     compile-time constants: no dispatch, no register indexing, and a column costs its own pairs only.
//   * A pair is two registers, its running key and c; the scoring owners of the pairs leave c in A[q] and in(b,c) in
//     in_s[q] -- kDeadPair where the segment is too small (:540) -- so nothing is decoded here, and in() is read when the
//     pair's column is finished (asked for a column ahead).
//   * A[q] becomes the chain's link: the argument, or kLinkNone at the chain's end (c == end).  The walk is one LDS byte
//     and three integer instructions per link, kept on the vector unit (values the same in every lane); the visited
//     candidates are collected one per lane and stored by the wave.
// 32-bit keys only when NM > 32 (64-bit keys would need 112 registers for the large class): k_solve keeps dp_solve_push for
// that instance.
// ---------------------------------------------------------------------------------------------
constexpr int kDeadPair = (int)0x80000000;

constexpr unsigned char kLinkNone = 255;

template <int NM, typename OutT, typename V>
__device__ __forceinline__ int dp_solve_wave_check() {
    static_assert(sizeof(V) == 8 || (i64)(NM - 1) * kKey32Reads < (1LL << 24), "32-bit DP keys: the longest chain's sum must stay below 2^24");
    return 0;
}

template <int NM, typename OutT, typename V>
__device__ __forceinline__ int dp_solve_wave(int n, const OutT *out_s, const int *in_s, V *M, unsigned char *A, int support,
                                             unsigned char *chosen /* + first candidate of the problem */ FSEG_DPARAM) {
    constexpr int SLOTS = (NM * (NM - 1) / 2 + 63) / 64;
    static_assert(NM <= 64, "a candidate per lane at the top level; an argument is six bits of a key");
    (void)dp_solve_wave_check<NM, OutT, V>();
    const int lane = lane_id();
    n = uni(n); support = uni(support);
    const int end = n - 1, npairs = n * (n - 1) / 2;
    constexpr V kNone = dp_key_none<V>(), kMin = dp_key_min<V>();
    int pc[SLOTS];                                  // c of pair (b,c)
    V best[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int q = s * 64 + lane;
        pc[s] = (int)A[q < npairs ? q : 0];
        best[s] = (q < npairs && pc[s] == end) ? (V)0 : kNone;       // M(b,end) = in(b,end): the chain's last link (:545-548)
    }
    int c2 = end, t2 = end * (end - 1) / 2, t3 = end * (end - 1) * (end - 2) / 6;
    // the pair this lane finishes in column c2 is q = t2 + ((lane - t2) & 63) if that offset is below c2; its in() is asked for
    // a column ahead (beside the push's loads), so that a column's chain holds no load of its own before the owners' write
    int in_nx = in_s[t2 + (((lane - t2) & 63) < c2 ? ((lane - t2) & 63) : 0)];
    auto stage = [&](auto Sc) {
        constexpr int S = decltype(Sc)::value;
        while (c2 >= 1 && (t2 >> 6) == S) {                         // (wave-uniform)
            const int d0 = (lane - t2) & 63, q0 = t2 + d0;
            if (d0 < c2) {
                V bb = best[S];
                if constexpr (S + 1 < SLOTS) bb = (q0 >> 6) == S ? bb : best[S + 1];
                const bool ok = in_nx != kDeadPair && bb > kMin;
                const V val = (bb >> 6) + (V)(ok ? in_nx : 0);
                M[q0] = ok ? (V)(val * 64 + (V)(63 - c2)) : kNone;
                A[q0] = (ok && c2 != end) ? (unsigned char)(63 - (int)(bb & 63)) : kLinkNone;
            }
            dp_sync<64>();
            if (c2 == 1) { c2 = 0; break; }
            const int t2n = t2 - (c2 - 1), d1 = (lane - t2n) & 63;
            in_nx = in_s[t2n + (d1 < c2 - 1 ? d1 : 0)];
            // pairs (b, c) with c < c2: q < t2 -- slots 0 .. S, the last one partly
            // (the loads of up to kBatch slots are in flight together: more would cost the registers the pairs live in)
            constexpr int kBatch = 8;
            static_for<0, (S + kBatch) / kBatch>([&](auto gc) {
                constexpr int s0 = decltype(gc)::value * kBatch, s1 = s0 + kBatch <= S + 1 ? s0 + kBatch : S + 1;
                V tail[s1 - s0];
                unsigned o[s1 - s0];
#pragma unroll
                for (int s = s0; s < s1; ++s) {
                    const int q = s * 64 + lane;
                    const bool act = s < S || q < t2;
                    tail[s - s0] = M[t2 + (act ? pc[s] : 0)];
                    o[s - s0] = (unsigned)out_s[t3 + (act ? q : 0)];
                }
#pragma unroll
                for (int s = s0; s < s1; ++s) {
                    const int q = s * 64 + lane;
                    const bool act = s < S || q < t2;
                    const V key = (V)o[s - s0] * 64 + tail[s - s0];          // (a tail of kKeyNone stays below every key of a value)
                    const bool ok = act & ((int)o[s - s0] >= support);       // :540
                    const V k2 = ok ? key : kNone;
                    best[s] = k2 > best[s] ? k2 : best[s];
                }
            });
            t2 = t2n; --c2; t3 -= t2;
        }
    };
    static_for_down<SLOTS - 1>(stage);
    FSEG_DTICK(10);
    // first maximiser over j of M(0,j) (larger value, then smaller j): the largest key, one candidate per lane
    const int j0 = lane >= 1 && lane < end ? lane : 1;
    V kv = (lane >= 1 && lane < end) ? M[j0 * (j0 - 1) / 2] : kNone;
    for (int d = 32; d >= 1; d >>= 1) {
        const V ov = __shfl_xor(kv, d);
        kv = ov > kv ? ov : kv;
    }
    FSEG_DTICK(11);
    int chain = 0;
    const bool cut = end >= 2 && kv > kMin && (kv >> 6) > (V)in_s[end * (end - 1) / 2];
    if (uni(cut ? 1 : 0)) {
        const int bj = 63 - (int)(kv & 63);
        int q = bj * (bj - 1) / 2, k = bj, rec = 0;                  // the state (j, k) is reached through pair q = (j, k)'s predecessor
#pragma nounroll
        for (; chain < 62; ++chain) {
            const int e = (int)A[q];
            if (e == (int)kLinkNone) break;
            rec = lane == chain ? e : rec;
            q = e * (e - 1) / 2 + k;                                 // pair (k, e): the next state
            k = e;
        }
        if (lane == 62) rec = 0;
        if (lane == 63) rec = bj;
        if (lane < chain || lane >= 62) chosen[rec] = 1;
    }
    FSEG_DTICK(12);
    return chain;
}



template <int NM> struct ScoreCfg {
    static constexpr int kPairs = NM * (NM - 1) / 2;
    static constexpr int kTri = NM * (NM - 1) * (NM - 2) / 6;
    static constexpr int kThreads = NM <= 16 ? 128 : (NM <= 32 ? 256 : 512);
    static constexpr int kSlots = (kPairs + kThreads - 1) / kThreads;
    static constexpr int kCovStride = NM + 1;      // odd: read-major rows do not collide on LDS banks
    static constexpr int kStage = (NM * kSub + kThreads - 1) / kThreads;   // coverage words per thread and tile
    static constexpr size_t kLds = (size_t)kPairs * 16 + (size_t)kSub * kCovStride * 4 + (size_t)((kPairs + 7) & ~7) * 2 +
                                   (size_t)((kTri + 7) & ~7) * 2;
};

inline size_t score_lds_for(int nm, int cov_stride) {
    size_t pairs = (size_t)nm * (nm - 1) / 2, tri = (size_t)nm * (nm - 1) * (nm - 2) / 6;
    return ((pairs * 16 + (size_t)kSub * cov_stride * 4 + ((pairs + 7) & ~(size_t)7) * 2 + ((tri + 7) & ~(size_t)7) * 2) + 15) & ~(size_t)15;
}


#ifdef FSEG_SCORE_TIMING
#define FSEG_TPARAM , unsigned long long *tacc
// diagnostic build: per-problem records behind the 16 phase slots -- (ticks, reads examined, reads with coverage, start tick)
// where the wave runs: HW_ID's low 16 bits (wave 3:0, SIMD 5:4, pipe 7:6, CU 11:8, SH 12, SE 15:13) | XCC_ID << 16 -- kept in the upper
// half of the record's second word (tools/prob_ticks.py: who is on which CU when)
__device__ __forceinline__ unsigned hw_where() {
    unsigned a, b;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(a));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(b));
    return (a & 0xffffu) | ((b & 0xfu) << 16);
}
#define FSEG_PROB_TICK(P, T0, LN, NA) do { if ((P) < kTaccProbs) { unsigned long long *r_ = tacc + 16 + 4 * (size_t)(P); \
        r_[0] = wall_clock64() - (T0); r_[1] = (unsigned long long)(LN) | ((unsigned long long)hw_where() << 32); r_[2] = (unsigned long long)(NA); r_[3] = (T0); } } while (0)
#define FSEG_T0 unsigned long long t_prev = wall_clock64()
#define FSEG_TICK(i) do { unsigned long long t_now = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&tacc[i], t_now - t_prev); t_prev = t_now; } while (0)
#else
#define FSEG_TPARAM
#define FSEG_T0
#define FSEG_TICK(i)
#endif


// The label arena starts every run filled with '0' (the label of a read without coverage, S7).  The fill depends on
// nothing but the arena's capacity, and the big-problem DP occupies a fraction of the GPU with latency-bound
// workgroups -- so the fill rides along as extra workgroups of that launch (k_label_zero when there is no DP launch).
__device__ __forceinline__ void fill_labels(uint4 *labels16, i64 n16, i64 first, i64 stride) {
    const uint4 z = make_uint4(0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u);
    for (i64 i = first; i < n16; i += stride) labels16[i] = z;
}


// DP of the small class (n <= kDpSmall) for batches of many partitions, where most problems have a handful of
// candidates: a workgroup takes four list entries at a time; every wave solves its own entry alone when it has at most
// kDpWave candidates (wave-private tables, wave-level synchronisation, no workgroup barrier on that path), and the
// entries above that are then solved one after the other by the whole workgroup as in k_dp.
constexpr int kDpWave = 16;

constexpr int kDpWavePairs = kDpWave * (kDpWave - 1) / 2, kDpWaveTri = kDpWave * (kDpWave - 1) * (kDpWave - 2) / 6;

template <typename OutT>
__host__ __device__ constexpr size_t dp_wave_bytes() {          // tables of one wave-private problem, 16-byte multiple
    return (((size_t)kDpWavePairs * (8 + 4 + 1) + (size_t)(kDpWaveTri + 4) * sizeof(OutT)) + 15) & ~(size_t)15;
}


// ---------------------------------------------------------------------------------------------
// Problems with at most kTiny candidates -- in batches of many partitions that is most of them (half have n = 3) -- are
// solved whole by ONE WAVE each: window coverage (get_cumulative_coverage :188-246), pair labels (:488-497), in / out
// counts (:500-528) and the DP (:532-566), without a work item, a coverage tile or an arena entry.  Lanes are the reads
// of the problem's lane range, 64 at a time: a pair's yea / nay plane for those reads is the result of one v_cmp (a
// ballot), lane q keeps pair q's planes and ambiguity count, lane t the count of triple t; the DP is dp_solve_push<64> on
// wave-private tables.  Four waves = four problems per workgroup, no workgroup barrier anywhere.
// ---------------------------------------------------------------------------------------------
constexpr int kTiny = 8;

constexpr int kTinyPairs = kTiny * (kTiny - 1) / 2, kTinyTri = kTiny * (kTiny - 1) * (kTiny - 2) / 6;


// ---------------------------------------------------------------------------------------------
// S5 whole by ONE WAVE per problem, for problems of at most NM candidates (NM = 8: the tiny list, NM = 16: solve list 0) --
// k_tiny's plan with the loads made cheap.  k_tiny fetches a read's exons with one gather per lane from the rep-ordered arrays (64
// cache lines per load instruction, and the texture path takes them one by one: that rate, not HBM or the ALUs, is what it runs
// at; k_solve gathers too, but since round 4 from the lane-ordered stream, where neighbouring lanes share lines) and every
// problem is a chain of such gathers.  Here the exons come from the lane-ordered stream `lex` (k_lanes): the reads a
// round examines -- up to 64 consecutive lanes -- own ONE contiguous piece of it, which the wave copies into LDS with
// lane-consecutive 16-byte loads; everything after that is LDS and registers:
//   per round: every lane finds the exons of its read that meet the window (ordered, so they are consecutive) and sums
//     their overlaps with [cand_0, cand_j) -- window coverage (get_cumulative_coverage :188-246) in registers, lane = read;
//     pair labels (:488-497): the 64 reads' bits of a pair's plane are one v_cmp, kept by lane q for pair q; in / out counts
//     (:500-528) with lane t owning triples t, t + 64, ..;
//   then dp_solve_push<64> (:532-566, :592-594) on wave-private tables.
// A read of the lane range without coverage in the window is scored like any other (all `nay`; ambiguous where lo < 0).
// No workgroup barrier anywhere; four waves = four problems per workgroup.
// ---------------------------------------------------------------------------------------------
// acc's lane `lane` := the wave-uniform value v (v_writelane_b32; this compiler has no builtin for it)
// (the lane number has to be an inline constant: a second scalar register would break the one-scalar-operand rule)
template <int LANE> __device__ __forceinline__ void write_lane(unsigned &acc, unsigned v) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(v), "n"(LANE));
}

constexpr int kWaveLanes = 1023;               // reads up to which a small problem is one wave's (16 rounds); beyond, the arena path

constexpr int kStageCap = 512;                 // exons of one round's reads staged in LDS (a round takes fewer reads if they own more)

constexpr int kWaveRepExons = kStageCap - 2;   // a batch with a rep of more exons than this keeps k_tiny / k_solve

template <int NM> struct WaveCfg {
    static constexpr int kPairs = NM * (NM - 1) / 2, kTri = NM * (NM - 1) * (NM - 2) / 6;
    static constexpr int kPSlots = (kPairs + 63) / 64, kTSlots = (kTri + 63) / 64;
#ifndef FSEG_WAVE_OCC8
#define FSEG_WAVE_OCC8 7        // (72 registers: all of a 250 k-read batch's ~7 000 tiny problems are resident at once, 30 -> 25 us)
#endif
#ifndef FSEG_WAVE_OCC16
#define FSEG_WAVE_OCC16 4
#endif
    static constexpr int kOcc = NM <= 8 ? FSEG_WAVE_OCC8 : FSEG_WAVE_OCC16;
};

template <int NM, typename V> struct __align__(16) WaveLds {
    int2 stage[kStageCap + 4];
    uint4 planes[WaveCfg<NM>::kPairs];             // {yea lo, yea hi, nay lo, nay hi} of the current round's reads
    V M[WaveCfg<NM>::kPairs];
    int in[WaveCfg<NM>::kPairs];
    unsigned out[WaveCfg<NM>::kTri + 4];
    unsigned char A[WaveCfg<NM>::kPairs + 8];
    int cy[NM];
};


// ---------------------------------------------------------------------------------------------
// Device-side fork and join of the scoring stage (round 5).  A dependency between two streams made of hipEventRecord +
// hipStreamWaitEvent costs 10-15 us on this runtime (a marker packet on one queue, a barrier packet on the other): with the
// stage's chains on three streams that was 31 of the 145 us between k_prob_emit's end and k_segments' start
// (profiles/r04_config4_stage_timeline.txt).  Instead:
//   fork: the side streams are forked EARLY by an event (before k_fix, or at the start of the piece that holds k_prob_emit: the
//         event's latency hides behind the kernels in front of the stage) and then run k_wait_word: one wave that sleeps until
//         the FIRST launch behind k_prob_emit on the main stream -- the plan's k_gate, else a k_signal -- has published this run's
//         generation.  (Published by k_prob_emit's own last workgroup the side streams started 5 us earlier, but a release
//         fence per workgroup -- buffer_wbl2 sc1: the XCD's whole L2 is searched for dirty lines, by 600 waves -- took the kernel
//         from 17 to 106 us: the release that costs nothing is the one at a kernel's end.)
//   join: the last launch of a side chain is k_signal (the chain's generation, stored with release order once the kernels in
//         front of it on that stream have ended), and the main stream runs k_wait_word on those words in front of k_segments.
// Every waiter has an exit every launch reaches: after `max_ticks` of the 100 MHz clock it raises kErrSyncTimeout and
// Status::sync_abort -- the scoring kernels behind it end at once (their lists may not exist yet) and the host reruns the batch
// with events (FSEG_DEV_SYNC=0).  A waiter must never sit on the hardware queue of the stream it waits for: the process's
// fourth stream shares a queue with the first, so only side streams 0 and 1 take waiters; a third keeps events.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool gen_reached(unsigned have, unsigned want) { return (int)(have - want) >= 0; }

template <int NM> struct SolveCfg {
    static constexpr int kThreads = ScoreCfg<NM>::kThreads;
    static constexpr int kSlots = ScoreCfg<NM>::kSlots;
    static constexpr int kRanges = kThreads / 64;                 // candidate ranges a read's coverage walk is cut into
    // waves per SIMD asked of the register allocator (HIP's second launch bound): the kernel is mostly waiting (descriptor
    // -> candidates -> exon block -> LDS phases -> DP chain), so what it needs is many problems in flight, not many
    // registers per thread; the big class must fit two 8-wave workgroups per CU
#ifndef FSEG_SOLVE_OCC
#define FSEG_SOLVE_OCC 1
#endif
#ifndef FSEG_SOLVE_OCC32
#define FSEG_SOLVE_OCC32 5      // (96 registers, five workgroups of the mid class per CU: 74 -> 70 us on config4; six spill and lose it again)
#endif
#ifndef FSEG_SOLVE_OCC16
#define FSEG_SOLVE_OCC16 5
#endif
    static constexpr int kMinBlocks = !FSEG_SOLVE_OCC ? 1 : (NM <= 16 ? FSEG_SOLVE_OCC16 : (NM <= 32 ? FSEG_SOLVE_OCC32 : 4));
};

// LDS of a k_solve workgroup: the pair planes (later the DP's M | in | A), a round's coverage rows, the count table.  (Round 4
// tried the planes IN the coverage rows' LDS -- 25 -> 16 KB for the mid class, 55 -> 39 KB for the large one: the planes then wait
// in registers across a barrier, the kernels sit at their register caps, and the spills cost 12-17 % per problem: DESIGN section 8.)
inline size_t solve_shared_bytes(int nm, int cov_stride) {
    const size_t planes = (size_t)nm * (nm - 1) / 2 * 16, cov = (size_t)kSub * cov_stride * 4;
    return (planes + cov + 15) & ~(size_t)15;
}

// a problem's slot of the hand-over arena (k_solve<.., SPLIT> -> k_dpw): in() per pair, the count table
constexpr int kDpxHeader = 0;

__host__ __device__ inline size_t dpx_in_bytes(int nm) { return ((size_t)nm * (nm - 1) / 2 * 4 + 15) & ~(size_t)15; }

inline size_t dpx_slot_bytes(int nm, int cnt_bytes) {
    const size_t tri = (size_t)nm * (nm - 1) * (nm - 2) / 6;
    return (kDpxHeader + dpx_in_bytes(nm) + ((tri * cnt_bytes + 15) & ~(size_t)15) + 255) & ~(size_t)255;
}

inline size_t solve_lds_for(int nm, int cov_stride, int cnt_bytes) {
    const size_t tri = (size_t)nm * (nm - 1) * (nm - 2) / 6;
    return (solve_shared_bytes(nm, cov_stride) + ((tri + 15) & ~(size_t)15) * cnt_bytes + 15) & ~(size_t)15;
}


// ---------------------------------------------------------------------------------------------
// The DP of the problems k_solve<.., SPLIT> has handed over: ONE WAVE per problem (a workgroup of one wave), the problem's
// in() and count table copied from its slot into LDS, then dp_solve_wave.  Why a launch of its own: a k_solve workgroup
// holds 25 KB (mid class) to 55-78 KB (large) of LDS and its DP needs a third of that and one wave of its four or eight;
// as the tail of the same workgroup (round 4's first version: the other waves ended early, which frees their registers --
// tools/probes/exit_probe.hip -- but not the workgroup's LDS) the large class's 385 workgroups sat on half of the chip's
// LDS for the 20-25 us of their DPs while the mid class waited for room (tools/prob_ticks.py: 250 mid-class problems in
// flight beside them, 1 250 once they were gone).  OutT says whose problems: the 8-bit instance's or the 16-bit one's.
// ---------------------------------------------------------------------------------------------
constexpr i64 kSplitGridCap = 1 << 20;      // workgroups of a split-path launch: k_dpw's workgroup b does problem b of its list, so lists beyond this are not split

inline size_t dpw_lds_for(int nm, int key_bytes, int cnt_bytes) {
    const size_t pairs = (size_t)nm * (nm - 1) / 2, tri = (size_t)nm * (nm - 1) * (nm - 2) / 6;
    return ((pairs * key_bytes + 15) & ~(size_t)15) + ((pairs * 4 + 15) & ~(size_t)15) + ((pairs + 15) & ~(size_t)15) + ((tri * cnt_bytes + 15) & ~(size_t)15);
}


// ---------------------------------------------------------------------------------------------
// Problems with kNMax < n <= kNHuge candidates (max_problem_size well above the default 50): the same scoring and
// DP with the triple counters left in the global arena.  One workgroup owns a problem outright and walks all of its
// coverage chunks itself, 32 reads at a time, so the counters are plain read-modify-writes (no atomics).  These are
// the slow-but-complete kernels; they are only launched when a previous run of the batch met such a problem.
// ---------------------------------------------------------------------------------------------
constexpr int kHugeSub = 32;           // reads per step of the huge-problem scoring kernel (one plane word per label)

constexpr size_t kHugeScoreLds = (size_t)(kNHuge * (kNHuge - 1) / 2) * 8 + (size_t)kHugeSub * (kNHuge + 1) * 4;


constexpr size_t kHugeDpLds = (size_t)(kNHuge * (kNHuge - 1) / 2) * (8 + 4 + 1) + 16;


// ---------------------------------------------------------------------------------------------
// Problems with kNHuge < n <= kNGiant candidates (round 5: max_problem_size beyond ~115 used to be refused, although the CLI
// -- like the reference's parse_args :108 -- accepts any value > 3 and optimize :475-568 has no size limit).  The same two
// kernels as the huge class with every per-pair table in GLOBAL scratch (a piece per workgroup, sized for the run's largest
// problem): the pair planes of the scoring kernel; M, in() and the argument (16 bits) of the DP.  Only the 32 reads' coverage
// rows and the candidates' positions stay in LDS (dynamic: (32 + 2) * (n + 1) words).  One workgroup owns a problem; values
// written by one wave and read by another are ordered by the workgroup's barriers (one CU, one L1).  Slow -- a thread walks its
// pairs' c2 loop from global memory -- and complete; checked against the oracle at max_problem_size 150 and 300.
// ---------------------------------------------------------------------------------------------
constexpr int kGiantWgs = 64;              // workgroups (and scratch pieces) of a giant-kernel launch

inline size_t giant_score_lds(int nm) { return (size_t)(2 * nm + kHugeSub * (nm + 1)) * 4; }

inline size_t giant_dp_lds(int nm) { return (size_t)nm * 4; }

inline size_t giant_scratch_bytes(int nm) {                 // per workgroup: max(planes, M + in + A)
    const size_t pairs = (size_t)nm * (nm - 1) / 2;
    return ((pairs * (8 + 4 + 2) + 255) & ~(size_t)255);
}


// ---------------------------------------------------------------------------------------------
// S6  refinement   (refine_segmentation :249-266) and final positions (:802-807)
// k_segments marks the chosen candidates as final positions and, for every chosen candidate whose
// previous chosen candidate is more than 40 positions away, records that segment; k_refine then
// visits the recorded segments (one wave each).
// ---------------------------------------------------------------------------------------------
constexpr int kSegChunks = 4;      // 64-candidate chunks of an interval that k_segments' one-wave path takes at once


constexpr int kRefCap = 1024;       // segment length up to which k_refine works out of LDS

// One workgroup per 64 read reps of one partition (a quarter of a 256-rep block).  The partition's column table
// (segment boundaries and integer thresholds) is staged in LDS when it fits; kLabelSplit threads share a rep: each
// merges the rep's exon list against a quarter of the columns the exons can reach (the walk is a chain of dependent
// loads, so shorter chains and more of them is what makes it faster).
constexpr int kLabelCols = 1024;

#ifndef FSEG_LABEL_STAGE
#define FSEG_LABEL_STAGE 1024
#endif
constexpr int kLabelStage = FSEG_LABEL_STAGE;

constexpr int kLabelSplit = 4;


// ---------------------------------------------------------------------------------------------
// upload-time preparation, once per batch, on the device (the inputs arrive in one copy; what used to be a host pass
// over every exon and a host sort now runs behind that copy on the context's stream)
//   k_prep_reps   the per-read assertions of read_split() (py/freddie_segment.py:158-161) and of process_splicing_data
//                 (:666-668: both ends of an exon are positions of one tint interval), and the sort key of every rep
//   (radix sort)  reps of a partition by first position (freddie_seg_sort.hip)
//   k_lanes       the lane list: every rep repeated rep_weight times, with the running maximum of the last position
//   k_hist_ranges the lanes that can reach each histogram chunk
// ---------------------------------------------------------------------------------------------
enum : unsigned { kPrepExonEnds = 1u, kPrepExonOrder = 2u, kPrepExonInterval = 4u, kPrepNoExons = 8u };

struct PrepStatus {
    unsigned err;
    unsigned pad;
    i64 bad_rep[4];    // smallest rep with error bit q
};


// One workgroup per partition walks the partition's reps in sorted order, 256 at a time: exclusive scan of the weights
// (lane offsets) and inclusive running maximum of the last positions, both with a carry from tile to tile.
// sort_here: no batch-wide sort ran (every partition has at most kLaneSortMax reps, the usual case): the workgroup sorts
// its partition's (first position, rep) keys itself, bitonic in LDS -- one launch instead of the radix sort's sixteen.
constexpr int kLaneSortMax = 2048;

// The exon stream of a tile of 256 sorted reps: thread u's rep owns [eb, eb + ne) of the tile's piece, which starts at
// lex[tile_e0]; its exons are exr.x .. in the caller's arrays.  Every thread marks its own range in an owner table (a
// byte per exon, kLexChunk exons at a time), then the workgroup copies the piece with lane-consecutive stores -- a thread
// copying its own rep's exons writes eight bytes every ~60 (the lane kernel: 17 -> 36 us per 250 k-read batch that way).
constexpr int kLexChunk = 4096;

__device__ __forceinline__ void lex_copy_tile(int eb, int ne, i64 src0, int tot_e, i64 tile_e0, const int *__restrict__ ex_ts,
                                              const int *__restrict__ ex_te, int2 *__restrict__ lex, unsigned char *owner_s /* kLexChunk */,
                                              int *eb_s /* 256 */, i64 *src_s /* 256 */) {
    eb_s[threadIdx.x] = eb; src_s[threadIdx.x] = src0;
    for (int c0 = 0; c0 < tot_e; c0 += kLexChunk) {
        __syncthreads();
        const int lo = max(eb, c0), hi = min(eb + ne, c0 + kLexChunk);
        for (int o = lo; o < hi; ++o) owner_s[o - c0] = (unsigned char)threadIdx.x;
        __syncthreads();
        const int end = min(tot_e, c0 + kLexChunk);
        for (int o = c0 + (int)threadIdx.x; o < end; o += 256) {
            const int u = owner_s[o - c0];
            const i64 src = src_s[u] + (o - eb_s[u]);
            lex[tile_e0 + o] = make_int2(ex_ts[src], ex_te[src]);
        }
    }
    __syncthreads();
}

// pair q = j*(j-1)/2 + i  <->  (i, j) as i | j << 8; independent of the problem size.  Built at COMPILE time: every translation
// unit that reads it (the scoring kernels' units) carries its own constant copy in its code object -- nothing to initialise,
// nothing to share across units (until the library was split into units a kernel filled one table per context).
struct PairTable { unsigned short v[kNMax * (kNMax - 1) / 2]; };
constexpr PairTable make_pair_table() {
    PairTable t{};
    int q = 0;
    for (int j = 1; j < kNMax; ++j) for (int i = 0; i < j; ++i) t.v[q++] = (unsigned short)(i | (j << 8));
    return t;
}
static __device__ const PairTable g_pair_table = make_pair_table();
#define g_pair_ij g_pair_table.v

}  // namespace fseg
