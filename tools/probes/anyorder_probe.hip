// Developer probe: does hipExtAnyOrderLaunch let a kernel start before the previous kernel of the SAME stream has ended on gfx950?
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/anyorder_probe.hip -o tools/probes/anyorder_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_spin(long long ticks, long long *t) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
int main() {
    long long *t; CK(hipMalloc(&t, 64));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, 5000LL, t);            // 50 us
            if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, 1000LL, t + 2);
            else hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, 1000LL, t + 2);
            CK(hipStreamSynchronize(s));
        }
        long long h[4]; CK(hipMemcpy(h, t, 32, hipMemcpyDeviceToHost));
        printf("%s: first kernel %.1f us; second starts %.1f us after the first starts\n", mode ? "any-order launch" : "plain launch   ",
               (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0);
    }
    return 0;
}
