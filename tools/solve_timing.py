"""Diagnostic only: phase shares inside k_solve (and its DP) from a -DFSEG_SCORE_TIMING build.  Never used for reported numbers.
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC -I include -DFSEG_SCORE_TIMING -o freddie_amd/libfreddie_seg_timing.so \
        freddie_amd/csrc/freddie_seg.hip freddie_amd/csrc/freddie_seg_sort.hip -lhsa-runtime64
    FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_timing.so python tools/solve_timing.py [workload]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from freddie_amd import _lib, tables
wl = sys.argv[1] if len(sys.argv) > 1 else "config4"
params = bench.PARAMS["config5" if wl == "config5" else "default"]
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
per, _ = bench.plan_batches(wl, 1)
per = int(os.environ.get("PROB_TICKS_PARTS", per))     # a few partitions: every problem alone on a CU (unloaded latency)
w = dict(bench.synth.WORKLOADS[wl]); w.pop("n_partitions")
parts = []
for i in range(per):
    g = bench.synth.generate(i, with_seq=False, **w)
    parts.append(bench.pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
b = bench.Batch(parts)
ctx = _lib.Context(0); ctx.set_params(**params, **tabs); ctx.set_profiling(True)
L = _lib.load()
L.fseg_debug_score_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.fseg_debug_timed_class.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = {0: "wait/next", 1: "setup+thresholds", 2: "coverage", 3: "pairs", 4: "triples", 8: "k_dp: wait/next", 9: "in_s / hand-over", 10: "dp columns", 11: "dp top level", 12: "dp backtrack",
         6: "k_cover: descriptor", 7: "k_cover: thresholds", 13: "k_cover: pass 1", 5: "k_cover: prefix", 14: "k_cover: pass 2"}
buf = np.zeros(16, np.uint64)
for cls in ((2,) if wl == "config2" else (2, 1, 0)):
    L.fseg_debug_timed_class(ctx._h, cls)
    ctx.upload(**b.arrays); ctx.run(); ctx.sync()
    L.fseg_debug_score_timing(ctx._h, buf.ctypes.data)
    N = 5
    for _ in range(N):
        ctx.run(); ctx.sync()
    L.fseg_debug_score_timing(ctx._h, buf.ctypes.data)
    tot = float(buf[:15].sum())
    print("class %d: scoring stage %.3f ms; ticks per run %.0f (100 MHz => %.1f us summed over workgroups)" % (cls, ctx.stage_ms()["interval_scoring"], tot / N, tot / N / 100.0))
    for k in sorted(names):
        print("   %-18s %5.1f%%" % (names[k], 100.0 * float(buf[k]) / max(tot, 1.0)))
ctx.close()
