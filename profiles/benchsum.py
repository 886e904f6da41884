import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print("%s value=%.4gM reads/s ms/pass(job)=%.3f timed=%.3fs | h2h %.4gM (%.3f ms/pass) | path frac %.3f / %.3f | scoring stage alone %.3f ms frac=%.3f (concurrent %.3f ms) | waiter time-outs %s" % (
        d["config"]["workload"], d["value"] / 1e6, d.get("ms_per_pass", d["ms_per_step"]), d.get("timed_s", 0.0),
        d.get("value_h2h", 0.0) / 1e6, d.get("ms_per_pass_h2h", 0.0), (d.get("roofline_path") or {}).get("frac", 0.0), (d.get("roofline_path") or {}).get("frac_h2h", 0.0),
        r["launch_ms"], r["frac"], d.get("roofline_concurrent", {}).get("launch_ms", 0.0), d.get("sync_timeouts")))
    st = d.get("roofline_stages") or {}
    print("  " + " ".join("%s=%.3f%s" % (k, v["ms"], ("(%.2f)" % v["frac"]) if "frac" in v else "") for k, v in st.items()))
    for w in ("config2", "config3", "config5"):
        rw = d.get("roofline_" + w)
        if rw:
            print("  %s stage alone %.3f ms frac=%.3f%s" % (w, rw["launch_ms"], rw["frac"],
                  (" (k_score only %.3f)" % rw["k_score_only"]["frac"]) if "k_score_only" in rw else ""))
    rw = d.get("roofline_whole_job")
    if rw:
        print("  the job as ONE batch (%d reads): stage alone %.3f ms frac=%.3f; %.3f ms a replay = %.4gM reads/s on one context" % (
            rw["reads"], rw["launch_ms"], rw["frac"], rw["ms_per_replay"], rw["reads_per_s_one_context"] / 1e6))
    e = d.get("e2e") or {}
    if "value" in e:
        print("  e2e %.3gM reads/s (%s s)" % (e["value"] / 1e6, ",".join("%.2f" % x for x in e["wall_s"])))
    sc = d.get("e2e_sidecar") or {}
    if "value" in sc:
        print("  e2e with side-cars %.3gM reads/s (%s s; same bytes: %s)" % (sc["value"] / 1e6, ",".join("%.2f" % x for x in sc["wall_s"]), sc.get("same_bytes_as_plain")))
