#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 --pmc passes of a profile round (tools/profile_round.sh): per workload, the
scoring stage's HBM bytes per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE;
KiB -> bytes) and its VALU utilisation (4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8), weighted by each
kernel's share of the stage's time).  The file is stamped with the source hash of the library the passes ran
(fseg_source_hash()): bench.py reports these figures only for that very library.  The raw counter CSVs stay in
gpurun_out/ (scratch); this file and the per-kernel summaries next to it are what is committed.
    python profiles/make_traffic.py gpurun_out/<tag> <round prefix>"""
import csv
import json
import os
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"      # (the prefix the summaries are committed under: tools/collect_profiles.sh)
STAGE = {"config4": ("k_solve<16", "k_solve<32", "k_solve<60", "k_dpw<", "k_wave<", "k_tiny"),
         "config2": ("k_cov", "k_score<60>", "k_dp<")}       # (config2: the arena path's coverage + scoring + DP, like config4's stage)


def load(path):
    out = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").replace("void ", "").split("(")[0]
        out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def mean_last(v, n=5):
    v = v[-n:]
    return sum(v) / len(v)


from freddie_amd import build  # noqa: E402
doc = {"_comment": "HBM traffic and VALU utilisation of the interval-scoring stage per launch, from rocprofv3 --pmc passes "
       "(FETCH_SIZE, WRITE_SIZE and the SQ counters each in their own run: tools/profile_round.sh).  bench.py copies these figures "
       "into its line only when source_hash is the hash of the library it has loaded: they are not measured in the benchmark run.",
       "source_hash": build.seg_hash(),
       "round": tag}
for w, kernels in STAGE.items():
    f = load(os.path.join(src, "pmc_%s_FETCH_SIZE" % w, "p_counter_collection.csv"))
    wr = load(os.path.join(src, "pmc_%s_WRITE_SIZE" % w, "p_counter_collection.csv"))
    sq = load(os.path.join(src, "pmc_%s_sq" % w, "p_counter_collection.csv"))
    per = {}
    traffic = 0.0
    util_num = util_den = 0.0
    for k in sorted(f):
        if not any(k.startswith(p) for p in kernels):
            continue
        fetch = mean_last(f[k]["FETCH_SIZE"]); write = mean_last(wr[k]["WRITE_SIZE"]) if k in wr else 0.0
        gui = mean_last(sq[k]["GRBM_GUI_ACTIVE"]); valu = mean_last(sq[k]["SQ_ACTIVE_INST_VALU"])
        insts = mean_last(sq[k]["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in sq[k] else None
        util = 4.0 * valu / (1024.0 * gui / 8.0)
        per[k] = {"fetch_size_kib_raw": fetch, "write_size_kib": write, "bytes": int((2 * fetch + write) * 1024), "valu_util": round(util, 3),
                  "gui_active_cycles_per_xcd": gui / 8.0, "valu_instructions": insts}
        traffic += (2 * fetch + write) * 1024
        util_num += util * gui; util_den += gui
    doc[w] = {"kernels": per, "traffic_bytes": int(traffic), "valu_util": round(util_num / util_den, 3) if util_den else None,
              "valu_source": "SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE pass of %s (time-weighted over the stage's kernels; profiles/%s_%s_sq_summary.txt)" % (tag, tag, w)}
json.dump(doc, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
print(json.dumps({w: {k: doc[w][k] for k in ("traffic_bytes", "valu_util")} for w in STAGE}, indent=1))
