import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    print("%s value=%.3gM reads/s ms/step=%.3f frac=%.3f score_ms=%.3f"%(d["config"]["workload"], d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms"]))
    print("  "+" ".join("%s=%.3f"%(k,v) for k,v in d["stage_ms"].items()))
