// Developer probe (round 5): what does it cost to bracket a chain of kernels with HIP events, and do events attached to kernel
// launches (hipExtLaunchKernelGGL's start / stop events: no marker packet of their own) give the same times?
//   chain on one stream:  A (20 us)  ->  B (20 us)  ->  C (20 us)
//   (1) hipEventRecord before B and after B             (2) B launched with startEvent / stopEvent
//   (3) startEvent on B, stopEvent on C  (an interval that spans two launches: what the scoring stage's bracket would be)
// and for each the device's own clock: end of A -> start of C (wall_clock64, 100 MHz), i.e. what the markers cost the chain.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/ext_event_probe.hip -o tools/probes/ext_event_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
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
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"no events", "hipEventRecord around B", "start/stop events ON B", "start on B, stop on C"};
    for (int mode = 0; mode < 4; ++mode) {
        float ms_sum = 0; double gap_ab = 0, gap_bc = 0, span = 0; const int reps = 20;
        for (int rep = 0; rep < reps + 2; ++rep) {
            hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, 2000LL, t);
            if (mode == 1) CK(hipEventRecord(e0, s));
            if (mode == 2) hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, e0, e1, 0, 2000LL, t + 2);
            else if (mode == 3) hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, e0, nullptr, 0, 2000LL, t + 2);
            else hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, 2000LL, t + 2);
            if (mode == 1) CK(hipEventRecord(e1, s));
            if (mode == 3) hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, nullptr, e1, 0, 2000LL, t + 4);
            else hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, 2000LL, t + 4);
            CK(hipStreamSynchronize(s));
            if (rep < 2) continue;
            long long h[6]; CK(hipMemcpy(h, t, 48, hipMemcpyDeviceToHost));
            gap_ab += (h[2] - h[1]) / 100.0; gap_bc += (h[4] - h[3]) / 100.0; span += (h[5] - h[0]) / 100.0;
            if (mode) { float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms_sum += ms; }
        }
        printf("%-26s A end -> B start %5.1f us, B end -> C start %5.1f us, A start -> C end %6.1f us; events say %6.1f us (B alone is 20.0, B + C 40.0 + gap)\n",
               names[mode], gap_ab / reps, gap_bc / reps, span / reps, ms_sum / reps * 1e3);
    }
    return 0;
}
