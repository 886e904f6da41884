"""Diagnostic only: per-phase time shares inside k_score, from a -DFSEG_SCORE_TIMING build
(hipcc ... -DFSEG_SCORE_TIMING -o freddie_amd/libfreddie_seg_timing.so).  Never used for reported numbers."""
import ctypes, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from freddie_amd import _lib, build, pack, synth, tables
build.SEG_SO = os.path.join(os.getcwd(), "freddie_amd", "libfreddie_seg_timing.so")
import bench
wl = sys.argv[1]
parts, n_reads = bench.build_batch(wl, 0, bench.per_gpu_partitions(wl, 1))
params = bench.PARAMS["default"]
tabs = dict(w_main=tables.gaussian_half_kernel(5.0, 4.0), w_refine=tables.gaussian_half_kernel(5.0, 1.0), h_table=np.asarray(tables.smooth_threshold(0.9)))
ctx = _lib.Context(0); ctx.set_params(**params, **tabs); ctx.upload(**pack.concat_batch(parts)); ctx.set_profiling(True); os.environ["X"]="1"
L = _lib.load(); L.fseg_debug_score_timing.argtypes=[ctypes.c_void_p, ctypes.c_void_p]
buf = np.zeros(16, np.uint64)
for i in range(3): ctx.run(); ctx.sync()
L.fseg_debug_score_timing(ctx._h, buf.ctypes.data)
N=5
for i in range(N): ctx.run(); ctx.sync()
L.fseg_debug_score_timing(ctx._h, buf.ctypes.data)
names=["queue/idle","setup","A copy","B pairs","C triples","flush"]
tot=buf[:6].sum()
print(wl, "score ms", ctx.stage_ms()["interval_scoring"], "sizes", ctx.sizes())
for n,v in zip(names, buf[:6]): print("  %-10s %6.1f%%  (%.1f us per WG per run at 100MHz clock)"%(n, 100.0*v/tot, v/N/256/100.0))
dn=["dp wait/top","dp stage","dp blocks","dp top-level","dp backtrack"]
dt=buf[8:13].sum()
print("  dp stage ms", ctx.stage_ms()["dp"])
for n,v in zip(dn, buf[8:13]): print("  %-12s %6.1f%%  (%.0f ticks total per run)"%(n, 100.0*v/max(dt,1), v/N))
