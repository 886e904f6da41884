#!/usr/bin/env python3
"""The REFERENCE's CPU figure on the benchmark's own seeds, measured in the build container.

Runs the unmodified reference CLI, /root/reference/py/freddie_segment.py, as a child process (`-t 1` and `-t <cores>`) on
split directories written by this repository's generator (freddie_amd.synth) for the first partitions of the bench
workloads -- the same seeds bench.py times on the GPU -- and records reads/s with everything a reader needs to judge the
number: versions, the host's core count and CPU model, the generator arguments, sha256 over the input files and over the
reference's output files (`-t 1` and `-t N` must write the same bytes).  Nothing of the reference is imported or copied;
the reference never travels to the GPU box, so the record (profiles/reference_cpu.json) is what bench.py quotes next to
its own `cpu_baseline` (the C port, timed on the GPU box).

    python tools/ref_cpu.py [--config4 200] [--config5 40] [--threads 1,8] [--out profiles/reference_cpu.json]
"""
import argparse
import hashlib
import json
import os
import platform
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freddie_amd import synth  # noqa: E402

REF_CLI = "/root/reference/py/freddie_segment.py"
RUN_PARAMS = {"config4": [], "config5": ["-sd", "3.0", "-tp", "0.80"]}       # BASELINE.json configs[3] / configs[4]


def sha256_tree(root):
    """One digest over the (relative path, content) of every file under root, in sorted order."""
    h = hashlib.sha256()
    n = 0
    for dp, dn, fs in os.walk(root):
        dn.sort()
        for f in sorted(fs):
            p = os.path.join(dp, f)
            h.update(os.path.relpath(p, root).encode() + b"\0")
            with open(p, "rb") as fh:
                h.update(fh.read())
            n += 1
    return h.hexdigest(), n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config4", type=int, default=200, help="partitions of config4 (500 reads each), indices 0..n-1")
    ap.add_argument("--config5", type=int, default=40, help="partitions of config5 (1 000 reads each; run with -sd 3.0 -tp 0.80)")
    ap.add_argument("--threads", default="1,%d" % (os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_cpu.json"))
    args = ap.parse_args()
    if not os.path.exists(REF_CLI):
        raise SystemExit("the reference is not here (%s): this script runs in the build container only" % REF_CLI)
    import numpy
    import scipy
    threads = [int(x) for x in args.threads.split(",") if x]
    rec = dict(what="unmodified reference CLI (vpc-ccg/freddie py/freddie_segment.py) as a child process on split directories written by "
                    "freddie_amd.synth for the first partitions of the bench workloads; wall time of the whole process",
               where="build container", cores=os.cpu_count(), cpu=cpu_model(), python=platform.python_version(),
               numpy=numpy.__version__, scipy=scipy.__version__, date=time.strftime("%Y-%m-%d"), runs={})
    for wl, n_part in (("config4", args.config4), ("config5", args.config5)):
        if n_part <= 0:
            continue
        kw = dict(synth.WORKLOADS[wl]); kw.pop("n_partitions")
        work = tempfile.mkdtemp(prefix="ref_cpu_")
        try:
            split = os.path.join(work, "split")
            for i in range(n_part):
                synth.generate(i, write_dir=split, **kw)
            in_sha, in_files = sha256_tree(split)
            reads = n_part * kw["n_reads"]
            entry = dict(partitions=n_part, reads=reads, generator=kw, params=" ".join(RUN_PARAMS[wl]) or "defaults",
                         input_sha256=in_sha, input_files=in_files, by_threads={})
            out_sha = None
            for t in threads:
                out = os.path.join(work, "out_t%d" % t)
                cmd = [sys.executable, REF_CLI, "-s", split, "-o", out, "-t", str(t)] + RUN_PARAMS[wl]
                t0 = time.perf_counter()
                subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, cwd=work)
                wall = time.perf_counter() - t0
                sha, n_out = sha256_tree(out)
                if out_sha is None:
                    out_sha = sha
                entry["by_threads"][str(t)] = dict(wall_s=round(wall, 2), reads_per_s=round(reads / wall, 1), output_files=n_out,
                                                   output_sha256=sha, same_bytes_as_first=(sha == out_sha))
                print("%s -t %d: %d reads in %.1f s = %.0f reads/s (%d files, outputs %s)" % (
                    wl, t, reads, wall, reads / wall, n_out, "identical" if sha == out_sha else "DIFFER"), flush=True)
                shutil.rmtree(out, ignore_errors=True)
            rec["runs"][wl] = entry
        finally:
            shutil.rmtree(work, ignore_errors=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
