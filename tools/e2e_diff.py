#!/usr/bin/env python3
"""Developer tool: run the drop-in CLI on a golden's split directory and show where its segment TSV differs from the reference's
bytes (first differing fields of the first lines):  python tools/e2e_diff.py <golden> [ENV=VALUE ...]"""
import sys, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import goldens
from test_host_mirror import input_dir
import pathlib
name = sys.argv[1]
g = goldens.load(name); case = goldens.manifest()["cases"][name]; run = case["run"]
tmp = pathlib.Path(tempfile.mkdtemp())
d, contig, tid = input_dir(name, tmp)
out = str(tmp / "out")
cmd = [sys.executable, os.path.join(ROOT, "py", "freddie_segment.py"), "-s", d, "-o", out, "--gpus", "1", "-sd", str(run["sigma"]), "-tp", str(run["threshold_rate"]), "-vf", str(run["variance_factor"]), "-mps", str(run["max_problem_size"]), "-lo", str(run["min_read_support_outside"])]
env = dict(os.environ); env.update({k: v for k, v in (a.split("=") for a in sys.argv[2:])})
r = subprocess.run(cmd, capture_output=True, text=True, env=env); print(r.returncode, r.stderr[-500:])
got = open(os.path.join(out, contig, "segment_%s_%d.tsv" % (contig, tid)), "rb").read().split(b"\n")
exp = g["segment_tsv"].tobytes().split(b"\n")
print(len(got), len(exp))
nd = 0
for i, (a, b) in enumerate(zip(got, exp)):
    if a != b:
        nd += 1
        if nd <= 3:
            fa, fb = a.split(b"\t"), b.split(b"\t")
            for k, (x, y) in enumerate(zip(fa, fb)):
                if x != y:
                    j = next(q for q in range(min(len(x), len(y))) if x[q] != y[q]) if x[:min(len(x),len(y))] != y[:min(len(x),len(y))] else min(len(x), len(y))
                    print("line", i, "field", k, "first diff at", j, "got", x[max(0,j-20):j+20], "exp", y[max(0,j-20):j+20], len(x), len(y))
print("differing lines", nd)
