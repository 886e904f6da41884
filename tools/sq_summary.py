#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of a rocprofv3 --pmc pass and the VALU utilisation they imply.
usage: sq_summary.py <counter_collection.csv> [<kernel_trace.csv>] [kernel substring]
SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over the waves (MI355X_MICROARCH.md); the chip has
256 CUs x 4 SIMDs.  valu_util = cycles in which a SIMD issued VALU work / cycles the kernel ran:
    4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)      (GRBM_GUI_ACTIVE is reported summed over the XCDs)."""
import csv, sys
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").split("(")[0].replace("void ", "")
    vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
args = [a for a in sys.argv[2:]]
pat = ""
for a in args:
    if not a.endswith(".csv"):
        pat = a
for k in sorted(vals):
    if pat and pat not in k:
        continue
    m = {cn: sum(v[-5:]) / len(v[-5:]) for cn, v in vals[k].items()}
    line = "%-34s" % k[:34]
    if "SQ_ACTIVE_INST_VALU" in m and m.get("GRBM_GUI_ACTIVE"):
        util = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
        line += " valu_util=%.3f" % util
    if m.get("SQ_WAVE_CYCLES"):
        line += " wait_any=%.2f wait_inst=%.2f active_valu=%.2f of wave cycles" % (
            m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_WAVE_CYCLES"])
    print(line)
    for cn, v in sorted(m.items()):
        print("   %-26s %16.0f" % (cn, v))
