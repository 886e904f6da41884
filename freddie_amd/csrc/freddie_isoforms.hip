// freddie_isoforms.hip -- gfx950 kernels + C-ABI (include/freddie_isoforms.h) of the isoform-consensus stage's per-read
// loops: the consensus counts of isoforms_cons() (py/freddie_isoforms.py:203-232) and the boundary votes of
// correct_boundaries() (:129-137).  Integer work only; the layout is read-major label bytes, so the consensus kernel's
// lanes (= consecutive segments of one read) load consecutive bytes.
#include "freddie_isoforms.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

namespace {

typedef long long i64;
typedef unsigned long long u64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// k_consensus: rows of any length (the launch takes it when an isoform of the call has more than kRowMax segments; FISO_ROWS=0
// forces it).  One workgroup per isoform, its reads in chunks of kSpanCap.  Phase 1: the four waves split the chunk's reads and
// find every read's span = first / last segment holding '1' (ballot over 256 label bytes at a time), widened by the
// tail rule of :217-224, into LDS.  Phase 2: lanes = 4 consecutive segments each, the waves split the reads again and count
// coverage / consensus, partial sums meet in LDS.  The chunk's label bytes (tens of KB) are read from HBM once: the
// second phase finds them in L2.
constexpr int kSpanCap = 1024;
constexpr int kE = 8;               // reads a wave keeps in flight (16 is slower: registers)
// The labels of four consecutive segments of a read, raw: four ASCII bytes -- or, PACKED, the two bytes that hold their
// 2-bit codes (label index g at bits 2(g & 3).. of byte g >> 2: what fseg_results_packed() delivers), with the shift.
template <bool PACKED>
__device__ __forceinline__ unsigned load_labels4(const unsigned char *labels, i64 first_label) {
    if (!PACKED) { unsigned w; __builtin_memcpy(&w, labels + first_label, 4); return w; }
    unsigned short h;
    __builtin_memcpy(&h, labels + (first_label >> 2), 2);
    return ((unsigned)h >> (2 * (unsigned)(first_label & 3))) & 0xffu;
}
// bit s of the result: segment s of the four holds '1'
template <bool PACKED>
__device__ __forceinline__ unsigned ones_of4(unsigned w) {
    if (!PACKED) {
        unsigned m = 0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) m |= (unsigned)(((w >> (8 * s4)) & 0xffu) == '1') << s4;
        return m;
    }
    const unsigned one = w & ~(w >> 1) & 0x55u;                  // code 1 = low bit set, high bit clear; at bits 0, 2, 4, 6
    return (one & 1u) | ((one >> 1) & 2u) | ((one >> 2) & 4u) | ((one >> 3) & 8u);
}
template <bool PACKED>
__global__ void __launch_bounds__(256) k_consensus(int n_iso, const i64 *iso_read_off, const int *n_seg, const i64 *iso_seg_off,
                                                   const i64 *read_lab_off, const unsigned char *labels, const unsigned char *tail,
                                                   int *cons, int *cov, int *tails) {
    __shared__ int t_s[3];
    __shared__ int2 sp_s[kSpanCap];
    __shared__ i64 off_s[kSpanCap];
    __shared__ unsigned char tl_s[kSpanCap];
    __shared__ int part[4][256][2];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int i = blockIdx.x; i < n_iso; i += gridDim.x) {
        const i64 r0 = iso_read_off[i], r1 = iso_read_off[i + 1];
        const int M = n_seg[i];
        __syncthreads();
        if (threadIdx.x < 3) t_s[threadIdx.x] = 0;
        if (r0 == r1) for (int j = threadIdx.x; j < M; j += blockDim.x) { cons[iso_seg_off[i] + j] = 0; cov[iso_seg_off[i] + j] = 0; }
        for (i64 rb = r0; rb < r1; rb += kSpanCap) {
            const int n = (int)(r1 - rb < kSpanCap ? r1 - rb : kSpanCap);
            __syncthreads();
            for (int q = threadIdx.x; q < n; q += blockDim.x) { off_s[q] = read_lab_off[rb + q]; tl_s[q] = tail[rb + q]; }   // one coalesced pass
            __syncthreads();
            // A lane owns 4 consecutive segments (one 4-byte load, possibly unaligned: rows are M bytes long), so a wave
            // covers 256 segments of a read per load instruction; kE reads are in flight per wave.  Loads are
            // unconditional from clamped (always valid) addresses, the predicate is applied afterwards: a load under a
            // condition compiles to a branch with its own wait, which would serialise the kE loads.
            for (int q0 = wave * kE; q0 < n; q0 += 4 * kE) {              // ---- phase 1: spans
                int first[kE], last[kE];
#pragma unroll
                for (int e = 0; e < kE; ++e) { first[e] = -1; last[e] = -1; }
                for (int j0 = 0; j0 < M; j0 += 256) {
                    const int j = j0 + 4 * lane;
                    const int jc = j < M ? j : 0;                          // bytes past the row's end are masked below
                    unsigned w[kE];
#pragma unroll
                    for (int e = 0; e < kE; ++e) w[e] = load_labels4<PACKED>(labels, off_s[q0 + e < n ? q0 + e : q0] + jc);
                    const unsigned in_row = j + 3 < M ? 15u : (j < M ? (1u << (M - j)) - 1u : 0u);     // segments of the four inside the row
#pragma unroll
                    for (int e = 0; e < kE; ++e) {
                        const unsigned ones = ones_of4<PACKED>(w[e]) & in_row;     // bit s = segment j + s holds '1'
                        const u64 m = __ballot(q0 + e < n && ones != 0);
                        if (m) {
                            const int lf = __ffsll((long long)m) - 1, ll = 63 - __clzll((long long)m);
                            const unsigned of = (unsigned)__builtin_amdgcn_readlane((int)ones, lf);
                            const unsigned ol = (unsigned)__builtin_amdgcn_readlane((int)ones, ll);
                            if (first[e] < 0) first[e] = j0 + 4 * lf + __ffs(of) - 1;
                            last[e] = j0 + 4 * ll + 31 - __clz(ol);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < kE; ++e) {
                    if (q0 + e >= n) break;
                    if (first[e] >= 0 && tl_s[q0 + e] == 1) { first[e] = 0; last[e] = M - 1; }   // 'S': the whole tint (:217-224)
                    if (lane == 0) sp_s[q0 + e] = make_int2(first[e], last[e]);
                }
            }
            __syncthreads();
            for (int q = threadIdx.x; q < n; q += blockDim.x)
                if (sp_s[q].x >= 0) atomicAdd(&t_s[tl_s[q]], 1);      // reads without a '1' are not counted (:215-216)
            for (int j0 = 0; j0 < M; j0 += 256) {                          // ---- phase 2: counts
                const int j = j0 + 4 * lane;
                const int jc = j < M ? j : 0;
                int x[4] = {0, 0, 0, 0}, c[4] = {0, 0, 0, 0};
                for (int q0 = wave * kE; q0 < n; q0 += 4 * kE) {
                    unsigned w[kE];
                    int2 sp[kE];
#pragma unroll
                    for (int e = 0; e < kE; ++e) {
                        const int q = q0 + e < n ? q0 + e : q0;
                        sp[e] = q0 + e < n ? sp_s[q] : make_int2(1, 0);        // an empty span for the padding reads
                        w[e] = load_labels4<PACKED>(labels, off_s[q] + jc);
                    }
#pragma unroll
                    for (int e = 0; e < kE; ++e) {
                        const unsigned ones = ones_of4<PACKED>(w[e]);
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            const bool in = j + s4 >= sp[e].x && j + s4 <= sp[e].y;   // spans end below M
                            c[s4] += in;
                            x[s4] += in && ((ones >> s4) & 1u);
                        }
                    }
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { part[wave][4 * lane + s4][0] = x[s4]; part[wave][4 * lane + s4][1] = c[s4]; }
                __syncthreads();
                {
                    const int jj = j0 + threadIdx.x;                       // 256 threads = the 256 segments of this step
                    if (jj < M) {
                        const int t = threadIdx.x;
                        const int xs = part[0][t][0] + part[1][t][0] + part[2][t][0] + part[3][t][0];
                        const int cs = part[0][t][1] + part[1][t][1] + part[2][t][1] + part[3][t][1];
                        const i64 o = iso_seg_off[i] + jj;
                        cons[o] = (rb == r0 ? 0 : cons[o]) + xs;           // later chunks add to the first chunk's sums
                        cov[o] = (rb == r0 ? 0 : cov[o]) + cs;
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        if (threadIdx.x < 3) tails[3 * (i64)i + threadIdx.x] = t_s[threadIdx.x];
    }
}

// ---- rows of at most kRowMax segments (every tint seen so far): one pass, a read's whole row in one load instruction ----------
// A read's M labels are a group of G = ceil(M / 16) consecutive lanes, 16 labels (one 16-byte load; 32 bits when PACKED) per
// lane, and a wave-wide load instruction carries 64 / G reads; kRowE of them are in flight per wave.  With the row in
// registers the span (first / last '1': one ballot, the group's bits of it, two lane reads) comes from the same bytes: no
// second pass, no spans in LDS.  What is counted:
//   consensus  a '1' always lies inside its read's span (the span IS first '1' .. last '1', and the tail rule only widens it),
//              so cons[j] = the reads that hold '1' at j: the flag bytes themselves (bit 0 of a label's byte), added four
//              segments to a register as BYTE counters, which go to the isoform's sums in LDS before a byte could overflow
//              (every kRowFlush row loads) and at the end;
//   coverage   cov[j] = reads with first <= j <= last = the running sum of (+1 at first, -1 at last + 1): two LDS atomics per
//              read by its first lane and one scan per isoform instead of a span mask per lane and load.
constexpr int kRowMax = 1024;        // 64 lanes x 16 labels
#ifndef FISO_ROW_E
#define FISO_ROW_E 2
#endif
#ifndef FISO_ROW_OCC
#define FISO_ROW_OCC 8
#endif
constexpr int kRowE = FISO_ROW_E;    // row loads a wave keeps in flight
constexpr int kRowFlush = 255 / kRowE * kRowE;   // row loads between flushes of the byte counters (a byte holds 255; a multiple of kRowE)
// (24-bit multiplies and dot products: full rate, where a 32-bit multiply takes four issue slots)
__device__ __forceinline__ unsigned spread4(unsigned nib) { return __umul24(nib, 0x00204081u) & 0x01010101u; }     // bit k -> bit 8k (k < 4)

template <bool PACKED>
__global__ void __launch_bounds__(256, FISO_ROW_OCC) k_consensus_rows(int n_iso, const i64 *iso_read_off, const int *n_seg, const i64 *iso_seg_off,
                                                        const i64 *__restrict__ read_lab_off, const unsigned char *__restrict__ labels,
                                                        const unsigned char *__restrict__ tail, int *cons, int *cov, int *tails) {
    __shared__ int cons_s[kRowMax], diff_s[kRowMax + 4];
    __shared__ int t_s[4], wtot_s[4];
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = blockIdx.x; i < n_iso; i += gridDim.x) {
        const i64 r0 = iso_read_off[i], r1 = iso_read_off[i + 1];
        const int M = n_seg[i];
        __syncthreads();                                                  // (the previous isoform's sums have left)
        for (int j = threadIdx.x; j <= M && j <= kRowMax; j += blockDim.x) { if (j < kRowMax) cons_s[j] = 0; diff_s[j] = 0; }
        if (threadIdx.x < 4) t_s[threadIdx.x] = 0;
        __syncthreads();
        if (M > 0 && M <= kRowMax && r0 < r1) {
            const int G = (M + 15) >> 4, rpw = 64 / G;                    // lanes per read, reads per row load
            const int g = lane / G, lp = lane - g * G;                    // this lane's read of the load, its 16 labels of the row
            const bool lane_on = g < rpw;
            const int seg0 = 16 * lp;
            // labels of the sixteen inside the row (lp < G: M > seg0), as byte masks of the four words; nothing for an idle lane
            const unsigned row16 = !lane_on ? 0u : (M - seg0 >= 16 ? 0xffffu : (1u << (M - seg0)) - 1u);
            const unsigned rowm[4] = {spread4(row16 & 15u), spread4((row16 >> 4) & 15u), spread4((row16 >> 8) & 15u), spread4(row16 >> 12)};
            const int gsh = g * G;                                        // the group's first lane
            const u64 gmask = lane_on ? (G == 64 ? ~0ULL : (1ULL << G) - 1ULL) << gsh : 0ULL;     // the group's lanes, where they are in the wave
            unsigned x[4] = {0, 0, 0, 0}, tc = 0;                         // byte counters: consensus; tails (a byte per tail value)
            auto flush = [&]() {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int xv = (int)((x[k] >> (8 * b)) & 255u);
                        if (xv) atomicAdd(&cons_s[seg0 + 4 * k + b], xv);      // (xv > 0 only inside the row)
                    }
                    x[k] = 0;
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) { const int v = (int)((tc >> (8 * t)) & 255u); if (v) atomicAdd(&t_s[t], v); }
                tc = 0;
            };
            const i64 stride = (i64)4 * rpw * kRowE;                      // reads a workgroup takes per step
            i64 off[kRowE]; unsigned tl[kRowE];
            const unsigned n_reads = (unsigned)(r1 - r0 < 0x10000000 ? r1 - r0 : 0x10000000);     // (an isoform of more reads than this: the two-pass kernel, see the launch)
            const unsigned lane_r = lane_on ? (unsigned)(wave * rpw + g) : 0u;
            const i64 *off_p = read_lab_off + r0;
            const unsigned char *tail_p = tail + r0;
            auto fetch_meta = [&](unsigned rel) {                        // rel: the step's first read, counted from the isoform's
#pragma unroll
                for (int e = 0; e < kRowE; ++e) {
                    unsigned rc = rel + (unsigned)(e * 4 * rpw) + lane_r;
                    rc = rc < n_reads ? rc : n_reads - 1u;                // (a valid read: loads are unconditional, the predicate comes after)
                    off[e] = off_p[rc]; tl[e] = tail_p[rc];
                }
            };
            // one step: kRowE row loads of this wave.  FULL: every read of the step exists (all but the isoform's last step).
            // (Tried: the next step's rows requested before this step's are worked on -- 101 registers, 0.103 against 0.093 ms;
            // with the arithmetic taken out the kernel takes 0.088 ms at rows of 150 labels and 0.070 at 148: a row that starts
            // off a 4-byte boundary costs the load path what the arithmetic costs the ALUs.)
            auto step = [&](auto full_c, i64 base, bool more) {
                constexpr bool FULL = decltype(full_c)::value;
                unsigned fb[kRowE][4];                                    // a lane's sixteen labels: a flag byte where the label is '1'
                unsigned tle[kRowE];
#pragma unroll
                for (int e = 0; e < kRowE; ++e) {
                    tle[e] = tl[e];
                    const i64 first_label = off[e] + seg0;
                    if (!PACKED) {
                        // (a row starts anywhere.  Five words from the 4-byte boundary below, shifted into place with v_alignbyte,
                        // instead of this unaligned 16-byte load: no faster at rows of 150 labels, slower at 144)
                        uint4 v; __builtin_memcpy(&v, labels + first_label, 16);
                        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned t = w[k] ^ 0x31313131u;                                        // a zero byte where the label is '1'
                            fb[e][k] = (~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) >> 7) & rowm[k];          // (exact: no carry crosses a byte)
                        }
                    } else {
                        u64 q; __builtin_memcpy(&q, labels + (first_label >> 2), 8);
                        const unsigned codes = (unsigned)(q >> (2 * (unsigned)(first_label & 3)));         // sixteen 2-bit codes
                        const unsigned one = codes & ~(codes >> 1) & 0x55555555u;                          // code 1: bit 2s
#pragma unroll
                        for (int k = 0; k < 4; ++k) fb[e][k] = __umul24((one >> (8 * k)) & 0x55u, 0x00041041u) & rowm[k];   // bit 2s -> bit 8s
                    }
                }
                if (more) fetch_meta((unsigned)(base - r0 + stride));     // the next step's offsets fly during this step's arithmetic
#pragma unroll
                for (int e = 0; e < kRowE; ++e) {
                    if (!FULL) {
                        if (base + (i64)(e * 4 + wave) * rpw >= r1) break;                                 // (wave-uniform: none of this load's reads exists)
                        const bool valid = base + (i64)(e * 4 + wave) * rpw + g < r1;
#pragma unroll
                        for (int k = 0; k < 4; ++k) fb[e][k] = valid ? fb[e][k] : 0u;
                    }
                    // flag bytes -> bits: a dot product with the bits' weights gathers a word's four flags
                    const unsigned o16 = __builtin_amdgcn_udot4(fb[e][1], 0x80402010u, __builtin_amdgcn_udot4(fb[e][0], 0x08040201u, 0u, false), false) |
                                         (__builtin_amdgcn_udot4(fb[e][3], 0x80402010u, __builtin_amdgcn_udot4(fb[e][2], 0x08040201u, 0u, false), false) << 8);
                    const u64 m = __ballot(o16 != 0);
                    const u64 gm = m & gmask;                                                              // this read's lanes that hold a '1'
                    const bool has = gm != 0;                                                              // reads without a '1' are not counted (:215-216)
                    const int lf = __ffsll((long long)gm) - 1, ll = 63 - __clzll((long long)gm);           // (lanes of the wave)
                    const int myf = __ffs(o16) - 1 - 16 * gsh, myl = 31 - __clz(o16) - 16 * gsh;           // (the group's first lane taken off here: one subtraction)
                    int first = 16 * lf + __shfl(myf, lf & 63), last = 16 * ll + __shfl(myl, ll & 63);
                    if (tle[e] == 1) { first = 0; last = M - 1; }                                          // 'S': the whole tint (:217-224)
                    if (has && lp == 0) {
                        atomicAdd(&diff_s[first], 1); atomicAdd(&diff_s[last + 1], -1);
                        tc += 1u << (8 * tle[e]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[k] += fb[e][k];
                }
            };
            fetch_meta(0u);
            int since = 0;
            i64 base = r0;
            for (; base + stride <= r1; base += stride) {                 // steps whose reads all exist
                step(std::true_type{}, base, base + stride < r1);
                since += kRowE;
                if (since >= kRowFlush) { flush(); since = 0; }
            }
            if (base < r1) step(std::false_type{}, base, false);
            flush();
        }
        __syncthreads();
        {   // coverage = the running sum of diff_s (four segments per thread, the waves' totals through LDS); results out
            const int j4 = 4 * threadIdx.x;
            int v[4], run = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { run += j4 + k < M && j4 + k < kRowMax ? diff_s[j4 + k] : 0; v[k] = run; }
            int inc = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(inc, d); if (lane >= d) inc += y; }
            if (lane == 63) wtot_s[wave] = inc;
            __syncthreads();
            int before = inc - run;
            for (int w = 0; w < wave; ++w) before += wtot_s[w];
            if (M <= kRowMax) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (j4 + k < M) { const i64 o = iso_seg_off[i] + j4 + k; cons[o] = cons_s[j4 + k]; cov[o] = before + v[k]; }
            }
        }
        if (threadIdx.x < 3) tails[3 * (i64)i + threadIdx.x] = t_s[threadIdx.x];
    }
}

// one workgroup per isoform: threads = the member reads' boundaries; every boundary votes for the isoform boundaries
// within the window (they are ascending: lower bound, then walk)
__global__ void __launch_bounds__(256) k_votes(int n_iso, const i64 *iso_read_off, const i64 *iso_b_off, const int *iso_bound,
                                               const i64 *read_b_off, const int *read_bound, int window, int *votes) {
    for (int i = blockIdx.x; i < n_iso; i += gridDim.x) {
        const i64 b0 = iso_b_off[i], nb = iso_b_off[i + 1] - b0;
        if (nb == 0) continue;
        const i64 r0 = iso_read_off[i], r1 = iso_read_off[i + 1];
        const i64 q0 = read_b_off[r0], q1 = read_b_off[r1];                // the member reads' boundaries are contiguous
        const int *ib = iso_bound + b0;
        for (i64 q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
            const int v = read_bound[q];
            i64 lo = 0, hi = nb;
            while (lo < hi) { i64 mid = (lo + hi) >> 1; if (ib[mid] < v - window) lo = mid + 1; else hi = mid; }
            for (i64 k = lo; k < nb && ib[k] <= v + window; ++k)
                atomicAdd(&votes[(b0 + k) * (2 * window + 1) + (v - ib[k] + window)], 1);
        }
    }
}

}  // namespace

struct fiso_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {};
    std::string err;
    float kernel_ms = 0.f;
};

namespace {

std::string g_create_error;

int fail(fiso_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

struct Dev {
    void *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

#define HIP_TRY(c, expr)                                                                                     \
    do {                                                                                                     \
        hipError_t e__ = (expr);                                                                             \
        if (e__ != hipSuccess) return fail((c), FISO_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__));      \
    } while (0)

template <typename T>
int to_device(fiso_ctx *c, Dev &d, const T *src, size_t n) {
    HIP_TRY(c, hipMalloc(&d.p, n * sizeof(T) + 32));      // (the kernels' vector loads reach up to 19 bytes past a row's end)
    if (n) HIP_TRY(c, hipMemcpyAsync(d.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return FISO_OK;
}
#define TRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

int check_offsets(fiso_ctx *c, const char *what, const int64_t *off, i64 n) {
    if (off[0] != 0) return fail(c, FISO_ERR_ARG, "%s does not start at 0", what);
    for (i64 i = 0; i < n; ++i) if (off[i + 1] < off[i]) return fail(c, FISO_ERR_ARG, "%s is not monotone at %lld", what, i);
    return FISO_OK;
}

}  // namespace

extern "C" {

int fiso_abi_version(void) { return 1; }

int fiso_create(int device, fiso_ctx **out) {
    if (!out) return FISO_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, FISO_ERR_HIP, "no HIP device available: %s (this library has no CPU fallback)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(nullptr, FISO_ERR_ARG, "device ordinal out of range");
    fiso_ctx *c = new fiso_ctx();
    c->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < 2; ++i) e = hipEventCreate(&c->ev[i]);
    if (e != hipSuccess) {
        fail(nullptr, FISO_ERR_HIP, "context creation failed: %s", hipGetErrorString(e));
        delete c;
        return FISO_ERR_HIP;
    }
    *out = c;
    return FISO_OK;
}

void fiso_destroy(fiso_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (int i = 0; i < 2; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    delete c;
}

const char *fiso_last_error(const fiso_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

static int consensus_impl(fiso_ctx *c, int32_t n_iso, const int64_t *iso_read_off, const int32_t *n_seg, const int64_t *iso_seg_off,
                          const int64_t *read_lab_off, const uint8_t *labels, const uint8_t *tail,
                          int32_t *cons_out, int32_t *cov_out, int32_t *tails_out, bool packed);
int fiso_consensus(fiso_ctx *c, int32_t n_iso, const int64_t *iso_read_off, const int32_t *n_seg, const int64_t *iso_seg_off,
                   const int64_t *read_lab_off, const uint8_t *labels, const uint8_t *tail,
                   int32_t *cons_out, int32_t *cov_out, int32_t *tails_out) {
    return consensus_impl(c, n_iso, iso_read_off, n_seg, iso_seg_off, read_lab_off, labels, tail, cons_out, cov_out, tails_out, false);
}
int fiso_consensus_packed(fiso_ctx *c, int32_t n_iso, const int64_t *iso_read_off, const int32_t *n_seg, const int64_t *iso_seg_off,
                          const int64_t *read_lab_off, const uint8_t *labels2, const uint8_t *tail,
                          int32_t *cons_out, int32_t *cov_out, int32_t *tails_out) {
    return consensus_impl(c, n_iso, iso_read_off, n_seg, iso_seg_off, read_lab_off, labels2, tail, cons_out, cov_out, tails_out, true);
}
static int consensus_impl(fiso_ctx *c, int32_t n_iso, const int64_t *iso_read_off, const int32_t *n_seg, const int64_t *iso_seg_off,
                          const int64_t *read_lab_off, const uint8_t *labels, const uint8_t *tail,
                          int32_t *cons_out, int32_t *cov_out, int32_t *tails_out, bool packed) {
    if (!c || n_iso <= 0 || !iso_read_off || !n_seg || !iso_seg_off || !cons_out || !cov_out || !tails_out) return FISO_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(check_offsets(c, "iso_read_off", iso_read_off, n_iso));
    TRY(check_offsets(c, "iso_seg_off", iso_seg_off, n_iso));
    const i64 R = iso_read_off[n_iso], S = iso_seg_off[n_iso];
    i64 lab_bytes = 0;
    int max_seg = 0;
    i64 max_reads = 0;
    for (int i = 0; i < n_iso; ++i) {
        if (n_seg[i] > max_seg) max_seg = n_seg[i];
        if (iso_read_off[i + 1] - iso_read_off[i] > max_reads) max_reads = iso_read_off[i + 1] - iso_read_off[i];
        if (n_seg[i] < 0 || iso_seg_off[i + 1] - iso_seg_off[i] != n_seg[i]) return fail(c, FISO_ERR_ARG, "isoform %d: iso_seg_off does not match n_seg", i);
        for (i64 r = iso_read_off[i]; r < iso_read_off[i + 1]; ++r) {
            if (read_lab_off[r] < 0 || tail[r] > 2) return fail(c, FISO_ERR_ARG, "read %lld: bad label offset or tail", r);
            if (read_lab_off[r] + n_seg[i] > lab_bytes) lab_bytes = read_lab_off[r] + n_seg[i];
        }
    }
    c->kernel_ms = 0.f;
    Dev d_iro, d_ns, d_iso, d_rlo, d_lab, d_tail, d_cons, d_cov, d_tails;
    TRY(to_device(c, d_iro, iso_read_off, (size_t)n_iso + 1));
    TRY(to_device(c, d_ns, n_seg, (size_t)n_iso));
    TRY(to_device(c, d_iso, iso_seg_off, (size_t)n_iso + 1));
    TRY(to_device(c, d_rlo, read_lab_off, (size_t)R));
    TRY(to_device(c, d_lab, labels, packed ? (size_t)(lab_bytes + 3) / 4 : (size_t)lab_bytes));     // (lab_bytes counts labels)
    TRY(to_device(c, d_tail, tail, (size_t)R));
    HIP_TRY(c, hipMalloc(&d_cons.p, (size_t)S * 4 + 16));
    HIP_TRY(c, hipMalloc(&d_cov.p, (size_t)S * 4 + 16));
    HIP_TRY(c, hipMalloc(&d_tails.p, (size_t)n_iso * 12 + 16));
    hipStream_t s = c->stream;
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    // rows of at most kRowMax labels (every isoform of the call): the one-pass kernel; FISO_ROWS=0 keeps the two-pass one (tests)
    const char *rows_env = getenv("FISO_ROWS");
    const bool rows = max_seg <= kRowMax && max_reads < 0x10000000 && !(rows_env && rows_env[0] == '0');
    const dim3 grid(n_iso < 8192 ? n_iso : 8192);
    if (rows && packed)
        hipLaunchKernelGGL(k_consensus_rows<true>, grid, dim3(256), 0, s, n_iso, d_iro.as<i64>(), d_ns.as<int>(),
                           d_iso.as<i64>(), d_rlo.as<i64>(), d_lab.as<unsigned char>(), d_tail.as<unsigned char>(),
                           d_cons.as<int>(), d_cov.as<int>(), d_tails.as<int>());
    else if (rows)
        hipLaunchKernelGGL(k_consensus_rows<false>, grid, dim3(256), 0, s, n_iso, d_iro.as<i64>(), d_ns.as<int>(),
                           d_iso.as<i64>(), d_rlo.as<i64>(), d_lab.as<unsigned char>(), d_tail.as<unsigned char>(),
                           d_cons.as<int>(), d_cov.as<int>(), d_tails.as<int>());
    else if (packed)
        hipLaunchKernelGGL(k_consensus<true>, dim3(n_iso < 8192 ? n_iso : 8192), dim3(256), 0, s, n_iso, d_iro.as<i64>(), d_ns.as<int>(),
                           d_iso.as<i64>(), d_rlo.as<i64>(), d_lab.as<unsigned char>(), d_tail.as<unsigned char>(),
                           d_cons.as<int>(), d_cov.as<int>(), d_tails.as<int>());
    else
        hipLaunchKernelGGL(k_consensus<false>, dim3(n_iso < 8192 ? n_iso : 8192), dim3(256), 0, s, n_iso, d_iro.as<i64>(), d_ns.as<int>(),
                           d_iso.as<i64>(), d_rlo.as<i64>(), d_lab.as<unsigned char>(), d_tail.as<unsigned char>(),
                           d_cons.as<int>(), d_cov.as<int>(), d_tails.as<int>());
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    if (S) HIP_TRY(c, hipMemcpyAsync(cons_out, d_cons.p, (size_t)S * 4, hipMemcpyDeviceToHost, s));
    if (S) HIP_TRY(c, hipMemcpyAsync(cov_out, d_cov.p, (size_t)S * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(tails_out, d_tails.p, (size_t)n_iso * 12, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    (void)hipEventElapsedTime(&c->kernel_ms, c->ev[0], c->ev[1]);
    return FISO_OK;
}

int fiso_boundary_votes(fiso_ctx *c, int32_t n_iso, const int64_t *iso_read_off, const int64_t *iso_b_off, const int32_t *iso_bound,
                        const int64_t *read_b_off, const int32_t *read_bound, int32_t window, int32_t *votes_out) {
    if (!c || n_iso <= 0 || !iso_read_off || !iso_b_off || !read_b_off || !votes_out) return FISO_ERR_ARG;
    if (window < 1 || window > 20) return fail(c, FISO_ERR_ARG, "window must be in [1, 20] (py/freddie_isoforms.py:45)");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(check_offsets(c, "iso_read_off", iso_read_off, n_iso));
    TRY(check_offsets(c, "iso_b_off", iso_b_off, n_iso));
    const i64 R = iso_read_off[n_iso], B = iso_b_off[n_iso];
    TRY(check_offsets(c, "read_b_off", read_b_off, R));
    const i64 Q = read_b_off[R];
    for (int i = 0; i < n_iso; ++i)
        for (i64 k = iso_b_off[i] + 1; k < iso_b_off[i + 1]; ++k)
            if (iso_bound[k] < iso_bound[k - 1]) return fail(c, FISO_ERR_ARG, "isoform %d: boundaries are not ascending", i);
    c->kernel_ms = 0.f;
    const size_t n_votes = (size_t)B * (size_t)(2 * window + 1);
    if (n_votes == 0) return FISO_OK;
    Dev d_iro, d_ibo, d_ib, d_rbo, d_rb, d_votes;
    TRY(to_device(c, d_iro, iso_read_off, (size_t)n_iso + 1));
    TRY(to_device(c, d_ibo, iso_b_off, (size_t)n_iso + 1));
    TRY(to_device(c, d_ib, iso_bound, (size_t)B));
    TRY(to_device(c, d_rbo, read_b_off, (size_t)R + 1));
    TRY(to_device(c, d_rb, read_bound, (size_t)Q));
    HIP_TRY(c, hipMalloc(&d_votes.p, n_votes * 4 + 16));
    hipStream_t s = c->stream;
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    HIP_TRY(c, hipMemsetAsync(d_votes.p, 0, n_votes * 4, s));
    hipLaunchKernelGGL(k_votes, dim3(n_iso < 8192 ? n_iso : 8192), dim3(256), 0, s, n_iso, d_iro.as<i64>(), d_ibo.as<i64>(),
                       d_ib.as<int>(), d_rbo.as<i64>(), d_rb.as<int>(), window, d_votes.as<int>());
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    HIP_TRY(c, hipMemcpyAsync(votes_out, d_votes.p, n_votes * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    (void)hipEventElapsedTime(&c->kernel_ms, c->ev[0], c->ev[1]);
    return FISO_OK;
}

int fiso_last_kernel_ms(fiso_ctx *c, float *ms) {
    if (!c || !ms) return FISO_ERR_ARG;
    *ms = c->kernel_ms;
    return FISO_OK;
}

}  // extern "C"

#ifdef FREDDIE_SOURCE_HASH
/* what this binary was built from (freddie_amd/build.py looks for the marker in the file) */
static const char freddie_source_stamp[] __attribute__((used)) = "FREDDIE_SRC_HASH=" FREDDIE_SOURCE_HASH;
#endif
