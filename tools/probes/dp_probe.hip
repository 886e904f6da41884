// Developer probe: latency of one segmentation DP (dp_solve_push, per thread count and table width) on tables already in LDS,
// alone on the GPU and with every CU busy with the same work; results are checked against a plain host DP.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DFSEG_SCORE_TIMING -I include tools/probes/dp_probe.hip freddie_amd/csrc/freddie_seg_sort.hip -lhsa-runtime64 -o tools/probes/dp_probe.bin
#include "../../freddie_amd/csrc/freddie_seg.hip"
#include <vector>
#include <random>

template <int T, typename OutT, typename V, int VARIANT, int NM>
__global__ void __launch_bounds__(T) k_dp_probe(int n, const OutT *out_g, const int *in_g, const int *cy_g, int support, int reps,
                                               unsigned char *chosen_g, int *chain_g, long long *ticks, unsigned long long *dp_tacc) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
    V *M = reinterpret_cast<V *>(smem);
    int *in_s = reinterpret_cast<int *>(M + npairs);
    OutT *out_s = reinterpret_cast<OutT *>(in_s + npairs);
    unsigned char *A = reinterpret_cast<unsigned char *>(out_s + ((ntri + 3) & ~3));
    __shared__ int cy_s[64];
    for (int i = threadIdx.x; i < n; i += T) cy_s[i] = cy_g[i];
    for (int i = threadIdx.x; i < npairs; i += T) in_s[i] = in_g[i];
    for (int i = threadIdx.x; i < ntri; i += T) out_s[i] = out_g[i];
    __shared__ unsigned char c_s[NM * (NM - 1) / 2];
    if constexpr (VARIANT == 2) {
        __syncthreads();
        for (int q = threadIdx.x; q < npairs; q += T) {
            int b, c;
            pair_decode(q, &b, &c);
            c_s[q] = (unsigned char)c;
            in_s[q] = (cy_s[c] - cy_s[b] < 5 && !(b == 0 && c == n - 1)) ? kDeadPair : in_s[q];
        }
    }
    __syncthreads();
    long long t0 = wall_clock64();
    unsigned long long dt_prev = t0;
    int chain = 0;
    for (int r = 0; r < reps; ++r) {
        if constexpr (VARIANT == 2) {                                // k_solve's tail: one wave, the pairs handed over by their scoring owners
            for (int q = threadIdx.x; q < npairs; q += T) A[q] = c_s[q];
            __syncthreads();
            if (threadIdx.x < 64) chain = dp_solve_wave<NM>(n, out_s, in_s, M, A, support, chosen_g + (size_t)blockIdx.x * 64 FSEG_DARG);
        } else
        chain = dp_solve_push<T, NM>(n, out_s, in_s, M, A, cy_s, support, chosen_g + (size_t)blockIdx.x * 64 FSEG_DARG);
        __syncthreads();
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { ticks[blockIdx.x] = t1 - t0; chain_g[blockIdx.x] = chain; }
}

// calibration: what one "write LDS -> barrier -> read LDS -> a little arithmetic" step costs for a workgroup of T threads
template <int T, int MODE>
__global__ void __launch_bounds__(T) k_pingpong(int steps, int *out, long long *ticks) {
    __shared__ int buf[2048];
    int x = threadIdx.x;
    buf[threadIdx.x] = x; buf[threadIdx.x + T] = x;
    __syncthreads();
    long long t0 = wall_clock64();
    for (int i = 0; i < steps; ++i) {
        if (MODE == 0) { __syncthreads(); x += i; }                                                   // barrier only
        else if (MODE == 1) { buf[threadIdx.x] = x; __syncthreads(); x = buf[(threadIdx.x + 65) % T] + i; }      // write, barrier, read
        else if (MODE == 2) { x = buf[(x + 65) & (T - 1)] + i; }                                         // dependent LDS reads, no barrier
        else { buf[threadIdx.x] = x; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); x = buf[(threadIdx.x + 1) % 64 + (threadIdx.x & ~63)] + i; }
    }
    long long t1 = wall_clock64();
    out[blockIdx.x * T + threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int T, int MODE> static void pingpong(const char *name) {
    int *o; long long *t; CK_(hipMalloc(&o, 4096 * T * 4)); CK_(hipMalloc(&t, 4096 * 8));
    for (int grid : {1, 512, 4096}) {
        const int steps = 2000;
        for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL((k_pingpong<T, MODE>), dim3(grid), dim3(T), 0, 0, steps, o, t); CK_(hipDeviceSynchronize()); }
        std::vector<long long> h(grid); CK_(hipMemcpy(h.data(), t, grid * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += v; avg /= grid;
        printf("%-40s T=%4d grid=%4d: %.1f ns per step\n", name, T, grid, avg * 10.0 / steps);
    }
    (void)hipFree(o); (void)hipFree(t);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct HostDp { std::vector<unsigned char> chosen; int chain; };
static HostDp host_dp(int n, const std::vector<int> &out, const std::vector<int> &in, const std::vector<int> &cy, int support) {
    const long long NEG = -(1LL << 60);
    const int end = n - 1;
    auto P = [](int a, int b) { return b * (b - 1) / 2 + a; };
    auto O = [](int a, int b, int c) { return c * (c - 1) * (c - 2) / 6 + b * (b - 1) / 2 + a; };
    std::vector<long long> M(n * n, NEG); std::vector<int> A(n * n, 255);
    for (int b = 0; b < end; ++b) M[b * n + end] = cy[end] - cy[b] >= 5 ? in[P(b, end)] : NEG;
    for (int c = end - 1; c >= 2; --c)
        for (int b = 1; b < c; ++b) {
            if (cy[c] - cy[b] < 5) continue;
            long long best = NEG; int arg = 255;
            for (int c2 = c + 1; c2 <= end; ++c2) {
                long long tail = M[c * n + c2]; int o = out[O(b, c, c2)];
                if (tail == NEG || o < support) continue;
                if (o + tail > best) { best = o + tail; arg = c2; }
            }
            if (best != NEG) { M[b * n + c] = best + in[P(b, c)]; A[b * n + c] = arg; }
        }
    long long bv = NEG; int bj = -1, bk = -1;
    for (int j = 1; j < end; ++j)
        for (int k = j + 1; k <= end; ++k) {
            if (cy[j] - cy[0] < 5 || cy[k] - cy[j] < 5) continue;
            long long tail = M[j * n + k]; int o = out[O(0, j, k)];
            if (tail == NEG || o < support) continue;
            long long cur = in[P(0, j)] + o + tail;
            if (cur > bv) { bv = cur; bj = j; bk = k; }
        }
    HostDp r; r.chosen.assign(64, 0); r.chain = 0;
    if (bv != NEG && bv > in[P(0, end)]) {
        int j = bj, k = bk; r.chosen[0] = 1;
        for (;;) { r.chosen[j] = 1; r.chosen[k] = 1; ++r.chain; if (k == end) break; int k2 = A[j * n + k]; if (k2 == 255) break; j = k; k = k2; }
    }
    return r;
}

template <int T, typename OutT, typename V, int VARIANT, int NM>
static void run(const char *name, int n, const std::vector<int> &out, const std::vector<int> &in, const std::vector<int> &cy, int support, const HostDp &ref) {
    const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
    std::vector<OutT> o2(out.begin(), out.end());
    OutT *d_out; int *d_in, *d_cy, *d_chain; unsigned char *d_ch; long long *d_t; unsigned long long *d_acc;
    CK(hipMalloc(&d_acc, 16 * 8));
    const int GMAX = 4096;
#ifdef FSEG_SCORE_TIMING
    const bool acc_on = true;
#else
    const bool acc_on = false;
#endif
    CK(hipMalloc(&d_out, ntri * sizeof(OutT))); CK(hipMalloc(&d_in, npairs * 4)); CK(hipMalloc(&d_cy, n * 4)); CK(hipMalloc(&d_chain, GMAX * 4));
    CK(hipMalloc(&d_ch, GMAX * 64)); CK(hipMalloc(&d_t, GMAX * 8));
    CK(hipMemcpy(d_out, o2.data(), ntri * sizeof(OutT), hipMemcpyHostToDevice)); CK(hipMemcpy(d_in, in.data(), npairs * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cy, cy.data(), n * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)npairs * (sizeof(V) + 4 + 1) + (size_t)((ntri + 3) & ~3) * sizeof(OutT) + 64;
    CK(hipFuncSetAttribute((const void *)k_dp_probe<T, OutT, V, VARIANT, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int grid : {1, 512, 4096}) {
        const int reps = 20;
        CK(hipMemset(d_ch, 0, GMAX * 64));
        for (int w = 0; w < 2; ++w) {
            CK(hipMemset(d_acc, 0, 16 * 8));
            hipLaunchKernelGGL((k_dp_probe<T, OutT, V, VARIANT, NM>), dim3(grid), dim3(T), lds, 0, n, d_out, d_in, d_cy, support, reps, d_ch, d_chain, d_t, d_acc);
            CK(hipDeviceSynchronize());
        }
        std::vector<long long> t(grid); std::vector<unsigned char> ch(64); int chain;
        CK(hipMemcpy(t.data(), d_t, grid * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ch.data(), d_ch + (size_t)(grid - 1) * 64, 64, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&chain, d_chain + grid - 1, 4, hipMemcpyDeviceToHost));
        long long mx = 0; double avg = 0; for (auto v : t) { mx = v > mx ? v : mx; avg += v; } avg /= grid;
        bool ok = chain == ref.chain; for (int i = 0; i < n; ++i) ok = ok && ch[i] == ref.chosen[i];
        printf("%-34s n=%d grid=%3d: %.2f us per DP (slowest block %.2f)  chain %d %s\n", name, n, grid, avg / reps / 100.0, mx / (double)reps / 100.0, chain, ok ? "ok" : "MISMATCH");
        if (grid == 1 && acc_on) {
            unsigned long long acc[16]; CK(hipMemcpy(acc, d_acc, sizeof acc, hipMemcpyDeviceToHost));
            printf("      phases (us per DP): init %.2f  slices %.2f  fix-up %.2f  top level %.2f  backtrack %.2f\n", acc[6] / 100.0 / reps, acc[5] / 100.0 / reps,
                   acc[10] / 100.0 / reps, acc[11] / 100.0 / reps, acc[12] / 100.0 / reps);
        }
    }
    hipFree(d_out); hipFree(d_in); hipFree(d_cy); hipFree(d_chain); hipFree(d_ch); hipFree(d_t);
}

int main() {
    pingpong<512, 0>("barrier only"); pingpong<512, 1>("LDS write, barrier, LDS read"); pingpong<512, 2>("dependent LDS reads");
    pingpong<128, 0>("barrier only"); pingpong<128, 1>("LDS write, barrier, LDS read"); pingpong<64, 3>("LDS write, wave fence, LDS read");
    pingpong<1024, 1>("LDS write, barrier, LDS read");
    for (int n : {59, 49, 38, 32, 30, 22, 16, 14, 10, 8, 5, 3}) {
        std::mt19937 rng(7 + n);
        const int npairs = n * (n - 1) / 2, ntri = n * (n - 1) * (n - 2) / 6;
        std::vector<int> out(ntri), in(npairs), cy(n);
        for (auto &v : out) v = rng() % 120;
        for (auto &v : in) v = -(int)(rng() % 30u);
        cy[0] = 0; for (int i = 1; i < n; ++i) cy[i] = cy[i - 1] + 3 + rng() % 9;
        const int support = 3;
        HostDp ref = host_dp(n, out, in, cy, support);
        printf("host: n=%d chain %d\n", n, ref.chain);
        if (n > 32) {
            run<64, unsigned char, int, 2, 60>("dp_solve_wave<60,u8,int>", n, out, in, cy, support, ref);
            run<64, unsigned short, int, 2, 60>("dp_solve_wave<60,u16,int>", n, out, in, cy, support, ref);
            run<512, unsigned char, int, 1, 60>("dp_solve_push<512,60,u8,int>", n, out, in, cy, support, ref);
            run<512, unsigned short, i64, 1, 60>("dp_solve_push<512,60,u16,i64>", n, out, in, cy, support, ref);
            run<1024, unsigned short, i64, 1, 60>("dp_solve_push<1024,60,u16,i64>", n, out, in, cy, support, ref);
        } else if (n > 16) {
            run<64, unsigned char, int, 2, 32>("dp_solve_wave<32,u8,int>", n, out, in, cy, support, ref);
            run<64, unsigned char, i64, 2, 32>("dp_solve_wave<32,u8,i64>", n, out, in, cy, support, ref);
            run<256, unsigned char, int, 1, 32>("dp_solve_push<256,32,u8,int>", n, out, in, cy, support, ref);
            run<512, unsigned char, int, 1, 32>("dp_solve_push<512,32,u8,int>", n, out, in, cy, support, ref);
        } else if (n > 8) {
            run<64, unsigned char, int, 2, 16>("dp_solve_wave<16,u8,int>", n, out, in, cy, support, ref);
            run<128, unsigned char, int, 1, 16>("dp_solve_push<128,16,u8,int>", n, out, in, cy, support, ref);
            run<64, unsigned char, int, 1, 16>("dp_solve_push<64,16,u8,int>", n, out, in, cy, support, ref);
        } else {
            run<64, unsigned char, int, 1, 8>("dp_solve_push<64,8,u8,int>", n, out, in, cy, support, ref);
        }
    }
    return 0;
}
