// Developer probe: how much does a device-to-host copy of a large buffer slow down kernels that run beside it, for the
// three ways of making that copy -- the runtime's hipMemcpyAsync, a small store-to-pinned-memory kernel, and the SDMA
// engine through hsa_amd_memory_async_copy.  Build: hipcc -O2 --offload-arch=gfx950 tools/probes/d2h_probe.hip -o gpurun_out/d2h_probe -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void k_copy_out(const v4u *src, v4u *dst, long long n16) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(src[i], dst + i);
}
// a latency-bound compute kernel: every thread chases pointers through a table (global loads that depend on each other)
__global__ void k_chase(const int *tab, int n, int steps, int *out) {
    int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
    for (int s = 0; s < steps; ++s) i = tab[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
}
// a streaming kernel (HBM bandwidth)
__global__ void k_stream(const v4u *a, v4u *b, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}
static hsa_agent_t g_gpu, g_cpu; static int g_ngpu = 0, g_ncpu = 0;
static hsa_status_t agent_cb(hsa_agent_t a, void *) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU) { if (g_ngpu++ == 0) g_gpu = a; } else if (t == HSA_DEVICE_TYPE_CPU) { if (g_ncpu++ == 0) g_cpu = a; }
    return HSA_STATUS_SUCCESS;
}
int main() {
    const size_t bytes = 75u << 20;
    void *d_src, *h_dst; int *tab, *out; v4u *sa, *sb;
    CK(hipMalloc(&d_src, bytes)); CK(hipHostMalloc(&h_dst, bytes, hipHostMallocDefault));
    const int N = 1 << 22; std::vector<int> h(N); for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 1103515245LL + 12345) % N);
    CK(hipMalloc(&tab, N * 4)); CK(hipMemcpy(tab, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMalloc(&out, 256 * 1024 * 4));
    const long long SN = 16 << 20; CK(hipMalloc(&sa, SN * 16)); CK(hipMalloc(&sb, SN * 16));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hsa_init(); hsa_iterate_agents(agent_cb, nullptr);
    hsa_signal_t sig; hsa_signal_create(1, 0, nullptr, &sig);
    auto compute = [&](const char *what) {
        float ms_c, ms_s;
        CK(hipEventRecord(e0, s1));
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k_chase, dim3(1024), dim3(256), 0, s1, tab, N, 64, out);
        CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_c, e0, e1));
        CK(hipEventRecord(e0, s1));
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s1, sa, sb, SN);
        CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_s, e0, e1));
        printf("%-28s chase x20: %7.3f ms   stream 4 x 512 MB: %7.3f ms (%.0f GB/s)\n", what, ms_c, ms_s, 4 * 2.0 * SN * 16 / ms_s / 1e6);
    };
    compute("alone (warm-up)"); compute("alone");
    for (int mode = 0; mode < 4; ++mode) {
        for (int blocks : {16, 48}) {
            if (mode != 1 && blocks != 16) continue;
            auto t0 = std::chrono::steady_clock::now();
            const int reps = 6;     // ~8 ms of copying: the compute sample (about 1 + 1.5 ms) falls inside it
            for (int r = 0; r < reps; ++r) {
                if (mode == 0) CK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s2));
                else if (mode == 1) hipLaunchKernelGGL(k_copy_out, dim3(blocks), dim3(256), 0, s2, (const v4u *)d_src, (v4u *)h_dst, (long long)(bytes / 16));
                else if (mode == 2) { hsa_signal_store_relaxed(sig, 1); if (hsa_amd_memory_async_copy(h_dst, g_cpu, d_src, g_gpu, bytes, 0, nullptr, sig) != HSA_STATUS_SUCCESS) { printf("hsa copy failed\n"); return 1; }
                                      if (r + 1 < reps) hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED); }
                else if (mode == 3) { hsa_signal_store_relaxed(sig, 1); hsa_amd_memory_async_copy_on_engine(h_dst, g_cpu, d_src, g_gpu, bytes, 0, nullptr, sig, HSA_AMD_SDMA_ENGINE_0, false);
                                      if (r + 1 < reps) hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED); }
                if (r == 0) { char name[64]; snprintf(name, sizeof name, mode == 0 ? "beside hipMemcpyAsync D2H" : mode == 1 ? "beside k_copy_out (%d WGs)" : mode == 2 ? "beside hsa async copy" : "beside hsa copy on SDMA 0", blocks);
                              if (mode >= 2) { /* first copy in flight (not waited) */ } compute(name); }
            }
            if (mode >= 2) hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
            CK(hipStreamSynchronize(s2));
            double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("    (%d copies of %zu MB in %.2f ms incl. the compute sample)\n", reps, bytes >> 20, dt);
        }
    }
    // copy rates alone
    for (int mode = 0; mode < 3; ++mode) {
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 4; ++r) {
            if (mode == 0) { CK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); }
            else if (mode == 1) { hipLaunchKernelGGL(k_copy_out, dim3(48), dim3(256), 0, s2, (const v4u *)d_src, (v4u *)h_dst, (long long)(bytes / 16)); CK(hipStreamSynchronize(s2)); }
            else { hsa_signal_store_relaxed(sig, 1); hsa_amd_memory_async_copy(h_dst, g_cpu, d_src, g_gpu, bytes, 0, nullptr, sig); hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED); }
        }
        double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 4;
        printf("copy alone, %s: %.3f ms per %zu MB = %.1f GB/s\n", mode == 0 ? "hipMemcpyAsync" : mode == 1 ? "k_copy_out(48)" : "hsa async copy", dt, bytes >> 20, bytes / dt / 1e6);
    }
    return 0;
}
