"""Diagnostic only (-DFSEG_SCORE_TIMING build, round 6): WHERE the scoring stage's problems run -- every problem's record carries the
hardware id of the wave that wrote it (XCC, SE, SH, CU, SIMD) -- and what each CU holds at a moment: workgroups per class, LDS,
wave slots and registers, against what one more mid-class workgroup needs.  The question it answers: the mid class has 600-800 of
its 2 000 workgroups in flight for most of the stage while half the chip's registers are free (tools/prob_ticks.py) -- what is full?
    FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_timing.so python tools/cu_occupancy.py [workload]"""
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
w = dict(bench.synth.WORKLOADS[wl]); w.pop("n_partitions")
parts = []
for i in range(per):
    g = bench.synth.generate(i, with_seq=False, **w)
    parts.append(bench.pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
b = bench.Batch(parts)
ctx = _lib.Context(0); ctx.set_params(**params, **tabs); ctx.set_profiling(True)
L = _lib.load()
L.fseg_debug_prob_ticks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
L.fseg_debug_dp_ticks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
L.fseg_debug_timed_class.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.fseg_debug_timed_class(ctx._h, -9)
ctx.upload(**b.arrays); ctx.run(); ctx.sync()
for _ in range(3):
    ctx.run(); ctx.sync()
print("scoring stage %.3f ms" % ctx.stage_ms()["interval_scoring"])
prob = ctx.tap("problems").reshape(-1, 4)
n = prob[:, 2]
rec = np.zeros((len(n), 4), np.uint64); L.fseg_debug_prob_ticks(ctx._h, rec.ctypes.data, len(n))
drec = np.zeros((len(n), 4), np.uint64); L.fseg_debug_dp_ticks(ctx._h, drec.ctypes.data, len(n))
ctx.close()


def unpack(r):
    dur = r[:, 0].astype(np.float64) / 100.0
    t0 = r[:, 3].astype(np.float64) / 100.0
    hw = (r[:, 1] >> np.uint64(32)).astype(np.int64)
    cu = ((hw >> 16) & 0xf) * 1024 + ((hw >> 13) & 7) * 128 + ((hw >> 12) & 1) * 64 + ((hw >> 8) & 0xf)      # (xcc, se, sh, cu) as one key
    simd = (hw >> 4) & 3
    return dur, t0, cu, simd


dur, t0, cu, simd = unpack(rec)
ddur, dt0, dcu, dsimd = unpack(drec)
base = t0[dur > 0].min()
t0 -= base; dt0 -= base
cus = np.unique(cu[dur > 0])
print("CUs seen: %d (XCCs %d)" % (len(cus), len(np.unique(cus // 1024))))
# what a workgroup of each kind holds: (LDS KB, waves, registers per wave); the tiny class: a wave of a 4-wave workgroup (a quarter of its 21 KB)
# (registers from the code objects' notes, in the allocation granule of 8: k_wave<8> 70, k_solve<16|32|60> 87 / 96 / 128, k_dpw 14 / 43 / 107)
KIND = {"tiny": (21.0 / 4, 1, 72), "small": (9.5, 2, 88), "mid": (25.0, 4, 96), "large": (57.0, 8, 128),
        "dp16": (1.7, 1, 16), "dp32": (9.5, 1, 48), "dp60": (31.0, 1, 112)}
cls = np.where(n <= 8, 0, np.where(n <= 16, 1, np.where(n <= 32, 2, 3)))
names = ["tiny", "small", "mid", "large"]
dnames = [None, "dp16", "dp32", "dp60"]
print("per-class problems: " + ", ".join("%s %d" % (names[k], int(((cls == k) & (dur > 0)).sum())) for k in range(4)))
cu_index = {c: i for i, c in enumerate(cus)}
for t in np.arange(20.0, max((t0 + dur).max(), (dt0 + ddur)[ddur > 0].max() if (ddur > 0).any() else 0) + 1, 10.0):
    lds = np.zeros(len(cus)); waves = np.zeros(len(cus)); regs = np.zeros(len(cus)); cnt = {k: 0 for k in KIND}
    live = (dur > 0) & (t0 <= t) & (t0 + dur > t)
    for k in range(4):
        m = live & (cls == k)
        idx = np.array([cu_index.get(c, -1) for c in cu[m]], int)
        kb, wv, rg = KIND[names[k]]
        np.add.at(lds, idx, kb); np.add.at(waves, idx, wv); np.add.at(regs, idx, wv * rg)
        cnt[names[k]] = int(m.sum())
    dlive = (ddur > 0) & (dt0 <= t) & (dt0 + ddur > t)
    for k in (1, 2, 3):
        m = dlive & (cls == k)
        idx = np.array([cu_index.get(c, -1) for c in dcu[m]], int)
        kb, wv, rg = KIND[dnames[k]]
        np.add.at(lds, idx[idx >= 0], kb); np.add.at(waves, idx[idx >= 0], wv); np.add.at(regs, idx[idx >= 0], wv * rg)
        cnt[dnames[k]] = int(m.sum())
    waiting_mid = int(((cls == 2) & (dur > 0) & (t0 > t)).sum())
    # a CU: 160 KB of LDS, 4 SIMDs x 8 wave slots (32), 4 x 512 registers a lane (2048 per CU)
    fits = (160.0 - lds >= 25.0) & (32 - waves >= 4) & (2048 - regs >= 4 * 96)
    print("t=%5.0f us  in flight %s | mid waiting %4d | CUs with room for a mid workgroup %3d / %d | per CU: LDS %5.1f KB mean (max %5.1f), waves %4.1f (max %2d), registers %4.0f (max %4d)" % (
        t, " ".join("%s %d" % kv for kv in cnt.items() if kv[1]), waiting_mid, int(fits.sum()), len(cus), lds.mean(), lds.max(), waves.mean(), int(waves.max()), regs.mean(), int(regs.max())))
# the stage's resource-time against the chip's: what perfect packing would take
area = {"LDS": 0.0, "wave slots": 0.0, "registers": 0.0}
for k in range(4):
    m = (cls == k) & (dur > 0); kb, wv, rg = KIND[names[k]]
    area["LDS"] += dur[m].sum() * kb; area["wave slots"] += dur[m].sum() * wv; area["registers"] += dur[m].sum() * wv * rg
for k in (1, 2, 3):
    m = (cls == k) & (ddur > 0); kb, wv, rg = KIND[dnames[k]]
    area["LDS"] += ddur[m].sum() * kb; area["wave slots"] += ddur[m].sum() * wv; area["registers"] += ddur[m].sum() * wv * rg
cap = {"LDS": 160.0 * len(cus), "wave slots": 32.0 * len(cus), "registers": 2048.0 * len(cus)}
print("resource-time of the stage / the chip's capacity (us at perfect packing): " + ", ".join("%s %.0f" % (k, area[k] / cap[k]) for k in area))
# how evenly the mid class is spread over the XCCs (a launch's workgroups go to the XCCs round robin by workgroup id)
m = (cls == 2) & (dur > 0)
x = cu[m] // 1024
print("mid-class problems per XCC: " + " ".join("%d" % int((x == k).sum()) for k in np.unique(x)))
print("mid-class busy time per XCC (sum of problem durations, us): " + " ".join("%.0f" % dur[m][x == k].sum() for k in np.unique(x)))
