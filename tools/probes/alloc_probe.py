#!/usr/bin/env python3
"""Developer tool: what first-use allocations cost on this box (hipMalloc / hipHostMalloc / hipFree by size).
    python tools/probes/alloc_probe.py"""
import ctypes
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipHostFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
t0 = time.perf_counter(); hip.hipInit(0); hip.hipSetDevice(0); p = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), 256); hip.hipDeviceSynchronize()
print("hipInit + first hipMalloc: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for mb in (1, 16, 64, 256, 1024, 2048):
    ts = []
    for _ in range(3):
        q = ctypes.c_void_p(); t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(q), mb << 20); t1 = time.perf_counter()
        hip.hipMemset(q, 0, mb << 20); hip.hipDeviceSynchronize(); t2 = time.perf_counter(); hip.hipFree(q); t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    print("hipMalloc %5d MB: %s ms (malloc / first memset / free)" % (mb, " | ".join("%.2f / %.2f / %.2f" % tuple(x * 1e3 for x in t) for t in ts)))
for mb in (1, 16, 64, 128):
    ts = []
    for _ in range(3):
        q = ctypes.c_void_p(); t0 = time.perf_counter(); rc = hip.hipHostMalloc(ctypes.byref(q), mb << 20, 0); t1 = time.perf_counter(); hip.hipHostFree(q); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    print("hipHostMalloc %4d MB: %s ms (malloc / free)" % (mb, " | ".join("%.2f / %.2f" % tuple(x * 1e3 for x in t) for t in ts)))
