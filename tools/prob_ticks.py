"""Diagnostic only (-DFSEG_SCORE_TIMING build): how long every DP problem of a batch takes inside k_solve / k_tiny, and when it
starts relative to the first one -- what the scoring stage's critical path is made of.
    FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_timing.so python tools/prob_ticks.py [workload]"""
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
L.fseg_debug_prob_ticks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
L.fseg_debug_timed_class.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.fseg_debug_timed_class(ctx._h, -9)                      # no phase clocks: only the per-problem records
ctx.upload(**b.arrays); ctx.run(); ctx.sync()
for _ in range(3):
    ctx.run(); ctx.sync()
print("scoring stage %.3f ms" % ctx.stage_ms()["interval_scoring"])
prob = ctx.tap("problems").reshape(-1, 4)
n = prob[:, 2]
rec = np.zeros((len(n), 4), np.uint64)
L.fseg_debug_prob_ticks(ctx._h, rec.ctypes.data, len(n))
dur = rec[:, 0].astype(np.float64) / 100.0          # us
ln, na = rec[:, 1].astype(np.int64), rec[:, 2].astype(np.int64)
t0 = rec[:, 3].astype(np.float64) / 100.0
t0 -= t0[t0 > 0].min()
end = t0 + dur
print("stage span from the records: %.1f us" % end.max())
for name, lo, hi in (("tiny (n<=8)", 0, 8), ("n<=16", 9, 16), ("n<=32", 17, 32), ("n<=60", 33, 60)):
    m = (n >= lo) & (n <= hi) & (dur > 0)
    if not m.any():
        continue
    d = dur[m]
    print("%-12s %5d problems: duration us  mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f | start us p50 %.1f p99 %.1f max %.1f | end max %.1f"
          % (name, m.sum(), d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max(),
             np.percentile(t0[m], 50), np.percentile(t0[m], 99), t0[m].max(), end[m].max()))
    for q in (0.999,):
        i = np.argsort(d)[-5:]
        idx = np.flatnonzero(m)[i]
        print("             slowest:", [(int(n[j]), int(ln[j]), int(na[j]), round(dur[j], 1), round(t0[j], 1)) for j in idx], "(n, reads examined, reads kept, us, start)")
    A = np.stack([np.ones(m.sum()), n[m], ln[m], na[m], n[m].astype(float) ** 2], 1)
    coef, *_ = np.linalg.lstsq(A, d, rcond=None)
    print("             fit us = %.1f + %.2f n + %.3f reads_examined + %.3f reads_kept + %.4f n^2" % tuple(coef))
# k_dpw's records (the split path: the DP is a launch of its own)
L.fseg_debug_dp_ticks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
drec = np.zeros((len(n), 4), np.uint64)
L.fseg_debug_dp_ticks(ctx._h, drec.ctypes.data, len(n))
ddur = drec[:, 0].astype(np.float64) / 100.0
dt0 = drec[:, 3].astype(np.float64) / 100.0 - (rec[:, 3].astype(np.float64) / 100.0)[rec[:, 3] > 0].min()
dend = dt0 + ddur
have_dp = ddur > 0
if have_dp.any():
    for name, lo, hi in (("n<=16", 9, 16), ("n<=32", 17, 32), ("n<=60", 33, 60)):
        m = (n >= lo) & (n <= hi) & have_dp
        if m.any():
            print("DP %-9s %5d problems: duration us  mean %.1f  p50 %.1f  p99 %.1f  max %.1f | start us p50 %.1f max %.1f | end max %.1f"
                  % (name, m.sum(), ddur[m].mean(), np.percentile(ddur[m], 50), np.percentile(ddur[m], 99), ddur[m].max(),
                     np.percentile(dt0[m], 50), dt0[m].max(), dend[m].max()))
    print("stage span with the DPs: %.1f us" % max(end.max(), dend[have_dp].max()))
# who is on the chip when: problems in flight per class every 10 us (a workgroup of 1 / 2 / 4 / 8 waves each until its DP)
cls = (("tiny", 0, 8), ("small", 9, 16), ("mid", 17, 32), ("large", 33, 60))
print("in flight at t us:   " + " ".join("%6s" % c[0] for c in cls) + "   DP: " + " ".join("%6s" % c[0] for c in cls[1:]))
last = max(end.max(), dend[have_dp].max() if have_dp.any() else 0.0)
for t in np.arange(0.0, last + 10.0, 10.0):
    row, drow = [], []
    for name, lo, hi in cls:
        m = (n >= lo) & (n <= hi) & (dur > 0)
        row.append(int(((t0[m] <= t) & (end[m] > t)).sum()))
        if lo > 8:
            md = (n >= lo) & (n <= hi) & have_dp
            drow.append(int(((dt0[md] <= t) & (dend[md] > t)).sum()))
    print("            %6.0f   " % t + " ".join("%6d" % v for v in row) + "       " + " ".join("%6d" % v for v in drow))
ctx.close()
