// Developer probe: are a wave's registers and wave slot free for other workgroups as soon as the WAVE ends, or only when its
// whole workgroup has ended?  Workgroups of 512 threads at 128 VGPRs (two fit a CU); in mode 1 waves 1..7 end at once and
// wave 0 spins for `spin_us`; in mode 0 every wave spins.  With per-wave release, mode 1 runs many more workgroups at a time.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/exit_probe.hip -o tools/probes/exit_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int REGS>
__global__ void __launch_bounds__(512, REGS == 128 ? 4 : 8) k_probe(int mode, long long spin_ticks, int lds_words, long long *t_start, long long *t_end, float *sink) {
    extern __shared__ int lds[];
    // hold REGS registers alive
    float acc[REGS - 24];
#pragma unroll
    for (int i = 0; i < REGS - 24; ++i) acc[i] = (float)(threadIdx.x + i);
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) t_start[blockIdx.x] = t0;
    if (lds_words > 0 && threadIdx.x < lds_words) lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (mode == 1 && threadIdx.x >= 64) {
        float s = 0;
#pragma unroll
        for (int i = 0; i < REGS - 24; ++i) s += acc[i];
        if (s == -1.0f) sink[0] = s;
        return;
    }
    while (wall_clock64() - t0 < spin_ticks) {
#pragma unroll
        for (int i = 0; i < REGS - 24; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < REGS - 24; ++i) s += acc[i];
    if (s == -1.0f) sink[0] = s;
    if (threadIdx.x == 0) t_end[blockIdx.x] = wall_clock64();
}

template <int REGS> static void run(int mode, int grid, int lds_bytes, double spin_us) {
    long long *ts, *te; float *sink;
    CK(hipMalloc(&ts, grid * 8)); CK(hipMalloc(&te, grid * 8)); CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute((const void *)k_probe<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    for (int w = 0; w < 2; ++w) {
        hipLaunchKernelGGL(k_probe<REGS>, dim3(grid), dim3(512), lds_bytes, 0, mode, (long long)(spin_us * 100), lds_bytes / 4, ts, te, sink);
        CK(hipDeviceSynchronize());
    }
    std::vector<long long> vs(grid), ve(grid);
    CK(hipMemcpy(vs.data(), ts, grid * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ve.data(), te, grid * 8, hipMemcpyDeviceToHost));
    long long t0 = *std::min_element(vs.begin(), vs.end()), t1 = *std::max_element(ve.begin(), ve.end());
    int first = 0; for (auto v : vs) first += (v - t0) < (long long)(spin_us * 100 / 2);
    printf("regs %3d mode %d (%s) grid %5d lds %6d B: span %.1f us, %d workgroups started in the first %.0f us\n", REGS, mode,
           mode ? "waves 1..7 end at once" : "all eight waves spin", grid, lds_bytes, (t1 - t0) / 100.0, first, spin_us / 2);
    CK(hipFree(ts)); CK(hipFree(te)); CK(hipFree(sink));
}

int main() {
    for (int lds : {1024, 16 * 1024, 50 * 1024}) {
        for (int mode : {0, 1}) {
            run<128>(mode, 4096, lds, 20.0);
            run<64>(mode, 4096, lds, 20.0);
        }
    }
    return 0;
}
