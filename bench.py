#!/usr/bin/env python3
"""Benchmark of the segmentation hot path on MI355X.

The JOB is fixed: BASELINE config 4 = 4 000 synthetic partitions x 500 reads = 2 000 000 reads (``--workload config5``:
5 000 x 1 000 reads, sigma 3, threshold rate 0.8), statically scattered over the N ranks (LPT on the partitions' reads,
freddie_amd/scatter.py -- no collective: partitions share nothing).  A STEP is one pass of the hot path over the rank's
whole share of the job as ONE batch of up to 2 M reads (where HBM allows eight contexts to hold it: BATCH_READS) that takes turns on the contexts of the
GPU (eight by default); since round 6 a step is PASSES_PER_STEP = 16 such passes (a timed region of 20 steps is then 1.2 s instead of
0.076 s; reads are counted per pass, so `value` is unchanged by it).  It is timed twice, K steps each, between barriers:
  value               with the inputs RESIDENT IN HBM when the timed part starts (every context holds a batch of the share;
                      a pass = ``fseg_run`` of every batch: histogram .. labels recomputed from the inputs, results left in HBM
                      and checked from one fetch per context afterwards) -- the contract's reading of ``value``;
  value_h2h           the way the drop-in CLI does it, PCIe inside: every batch ``fseg_upload`` (host arrays -> HBM) ->
                      ``fseg_run`` -> ``fseg_results_packed`` (final positions and the label matrix at two bits per label in host
                      memory), one batch's copies overlapping the others' kernels; nothing is replayed.  Rounds 1-4 reported this
                      rate as ``value``; it depends on the box's host as much as on the GPU (300-409 M reads/s box to box).
``value`` = 2 000 000 reads x steps / max-over-ranks wall time (reference unit of work: run_segment,
py/freddie_segment.py:681-735, minus the file I/O which the ``e2e`` leg adds), ``ms_per_step`` = the time of the whole job,
``scaling`` = "strong".

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config4|config2|config3|config5|config1]
  python bench.py --workload cluster-many|cluster-big|isoforms      (rows N3 / N4 of SURVEY 8f: their own metrics, 1 GPU)

Prints ONE JSON line (rank 0) with, besides the driver's fields:
  roofline              the interval-scoring stage ALONE on the GPU (one context, one resident batch of the job replayed, HIP
                        events around the stage on the library's own streams): algorithmic bytes 4*(N+K)*R + 4*R per partition
                         over that time -- the figure the per-kernel medians of profiles/ reproduce
  roofline_concurrent   the same bracket inside the timed steps, where the other contexts' kernels run beside the stage
  roofline_path         the fraction that belongs to `value` (and `value_h2h`): the algorithmic bytes of EVERY stage of one pass over
                        the job over the time of a pass, against 8 TB/s per GPU
  sync_timeouts         device-side waiters of the scoring stage that reached their time limit (must be 0)
  roofline_stages       every stage of the path alone on the GPU (first-run path of distinct batches, events around every
                        stage): SURVEY 8d's algorithmic bytes of the stage over its time
  roofline_config2      one 50 k-read x 2 k-candidate partition (BASELINE configs[1], the arena path): coverage + scoring + DP
                        (interval_scoring = k_cov + k_score, + the dp stage), like the other configs' stage; k_score alone as a sub-field
  roofline_config3 / roofline_config5   the stage of one resident batch of BASELINE configs[2] / [4] (config4 runs only)
  roofline_whole_job    the stage with the whole job as ONE resident batch (config4: 2 M reads, one launch): what the stage does when the
                        fork / join and its tail are paid once (config4 runs only)
  value_resident_replay replay of one resident batch (no copies, no sizing; plain launches on the library's streams -- a run
                        that forks is not replayed as a hipGraph: DESIGN.md section 3)
  value_hbm_resident    batches uploaded first, then each run once (first-run path, no copies in the timed part)
  cpu_baseline / cpu_baseline_all_cores   the C oracle on this box's host cores (1 thread / every core)
  e2e                   the drop-in CLI on a split directory of the same job in tmpfs: files in -> files out, through the CLI's N
                        worker processes, run by rank 0 before any rank touches a GPU
  valu_util, roofline.traffic   counter figures of the scoring kernels from the committed rocprofv3 --pmc passes
                        (profiles/traffic.json) -- null unless that file was made from the very library this run loads
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from freddie_amd import pack, synth, tables  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Reads of a batch.  With eight contexts (one box, reads/s resident / host memory -> host memory, the stage alone as a fraction of the HBM roofline):
# 250 k (rounds 2-6): 539 M / 368 M, 0.345;  500 k: 560 / 394;  1 M: 576 / 381-413, 0.354;  2 M = the whole job as ONE batch that all eight
# contexts hold: 578 / 398-415, 0.383 -- every launch pays its stages' tails and the scoring stage's fork / join once.  A context that holds the
# 2 M-read job takes 26.4 GB of HBM (13.2 KB a read: 116 B a position), eight of them 211 of the card's 309 GB: the default is the whole job
# where the cards' free memory (sysfs, no HIP call: this is decided before anything forks) allows it, half of it and so on otherwise.
HBM_BYTES_PER_READ = 14000          # (config4, rounded up; the 1 000-read partitions of config3 / config5 take half of it)


def hbm_free_bytes():
    """The least free VRAM over the GPUs this process could use, without a HIP call in this process: the KFD topology's GPU nodes whose
    render node is accessible (freddie_amd/devices.py: on a shared host sysfs shows every tenant's card), amdgpu's mem_info_vram_* of
    each; a child process that asks the runtime where sysfs does not say; None where neither does."""
    from freddie_amd import devices
    free = []
    for _, props in (devices._kfd_gpu_nodes() or []):
        minor = props.get("drm_render_minor", -1)
        if minor < 0 or not devices._usable(props):
            continue
        try:
            base = "/sys/class/drm/renderD%d/device/mem_info_vram_" % minor
            free.append(int(open(base + "total").read()) - int(open(base + "used").read()))
        except (OSError, ValueError):
            pass
    if free:
        return min(free)
    code = ("import ctypes\n"
            "L=ctypes.CDLL('libamdhip64.so'); n=ctypes.c_int(0); out=[]\n"
            "if L.hipGetDeviceCount(ctypes.byref(n))==0:\n"
            "    for d in range(n.value):\n"
            "        f=ctypes.c_size_t(0); t=ctypes.c_size_t(0)\n"
            "        if L.hipSetDevice(d)==0 and L.hipMemGetInfo(ctypes.byref(f),ctypes.byref(t))==0: out.append(f.value)\n"
            "print(min(out) if out else -1)\n")
    try:
        v = int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout.strip() or -1)
        return v if v > 0 else None
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return None


def default_batch_reads(contexts=8, want=2000000):
    free = hbm_free_bytes()
    if free is None:
        return 1000000                                   # (nothing known about the card: what 96 GB hold)
    while want > 250000 and want * HBM_BYTES_PER_READ * contexts > 0.8 * free:
        want //= 2
    return want


BATCH_READS = int(os.environ.get("FREDDIE_BENCH_BATCH_READS", "0")) or default_batch_reads()      # (the override: tuning runs and tests)
# A STEP is this many passes over the job (round 6; until round 5: one).  One pass of the 2 M-read job takes 3.8 ms on one GPU, so the
# driver's 20 steps were a timed region of 0.076 s -- too short for anything that samples the GPU from outside.  Reads are counted
# per pass: `value` does not depend on this number, `ms_per_step` is the time of PASSES_PER_STEP passes.
PASSES_PER_STEP = int(os.environ.get("FREDDIE_BENCH_PASSES", "16"))


def popcount_bytes(a):
    """Set bits of a uint8 array (numpy >= 2 has bitwise_count; a table look-up otherwise)."""
    a = np.asarray(a, np.uint8)
    if hasattr(np, "bitwise_count"):
        return int(np.bitwise_count(a).sum(dtype=np.int64))
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint8)
    return int(lut[a].sum(dtype=np.int64))

PARAMS = {
    "default": dict(sigma=5.0, threshold_rate=0.9, variance_factor=3.0, max_problem_size=50,
                    min_read_support_outside=3, ignore_ends=True),
    "config5": dict(sigma=3.0, threshold_rate=0.8, variance_factor=3.0, max_problem_size=50,
                    min_read_support_outside=3, ignore_ends=True),
}


class Batch:
    def __init__(self, parts):
        self.parts = parts
        self.arrays = pack.concat_batch(parts)
        self.n_reads = sum(p.n_reads for p in parts)
        self.alg_bytes = None
        self.label_popcount = None
        self.path_bytes = None          # SURVEY 8(d)'s algorithmic bytes of every stage of the path, summed (noted in the warm-up)


def plan_job(workload, rank, n_gpus):
    """(partition indices of this rank, partitions per batch): the workload's partitions are scattered over the ranks by LPT
    on their reads (freddie_amd/scatter.py; equal partitions end up striped) and a rank's share is cut into batches of about
    BATCH_READS reads."""
    from freddie_amd import scatter
    w = synth.WORKLOADS[workload]
    n_part = w["n_partitions"]
    mine = scatter.rank_share([w["n_reads"]] * n_part, rank, n_gpus) if n_gpus > 1 else list(range(n_part))
    per = max(1, min(len(mine) or 1, BATCH_READS // w["n_reads"]))
    return mine, per


def plan_batches(workload, n_gpus):
    """(partitions per batch, batches of rank 0) -- what the developer tools under tools/ size their one batch with."""
    mine, per = plan_job(workload, 0, n_gpus)
    return per, (len(mine) + per - 1) // per


def build_batches(workload, rank, n_gpus):
    w = dict(synth.WORKLOADS[workload])
    w.pop("n_partitions")
    mine, per = plan_job(workload, rank, n_gpus)
    batches = []
    for b0 in range(0, len(mine), per):
        parts = []
        for i in mine[b0:b0 + per]:
            g = synth.generate(i, with_seq=False, **w)
            parts.append(pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
        batches.append(Batch(parts))
    return batches


# ---- CPU legs (they fork or start processes: all of them run BEFORE this process touches the GPU) -------------------
def cpu_baseline(batches, params, tabs, min_s=10.0, max_s=25.0):
    """Time the CPU oracle on a bounded sample of the same workload: whole partitions in order, repeated
    until at least min_s seconds of CPU work have been measured (never more than max_s)."""
    from oracle import oracle
    parts = [p for b in batches for p in b.parts]
    t0 = time.perf_counter()
    reads = used = passes = 0
    while True:
        for p in parts:
            o = oracle.segment(p.iv_start, p.iv_end, p.rep_weight, p.rep_exon_off, p.ex_ts, p.ex_te, **params, **tabs)
            if o["error"]:
                raise RuntimeError("oracle failed: " + o["errmsg"])
            reads += p.n_reads
            used += 1
            if time.perf_counter() - t0 > max_s:
                break
        passes += 1
        if time.perf_counter() - t0 > min_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=reads / dt, unit="reads/s", cores=1, kind="port",
                sample="%d partition runs (%d reads) of the C oracle over the workload's partitions in order, %.1f s" % (used, reads, dt))


_POOL_JOB = None


def _oracle_one(i):
    from oracle import oracle
    parts, params, tabs = _POOL_JOB
    p = parts[i]
    o = oracle.segment(p.iv_start, p.iv_end, p.rep_weight, p.rep_exon_off, p.ex_ts, p.ex_te, **params, **tabs)
    return o["error"]


def host_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))           # the CPUs this process may use, not the machine's
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline_all_cores(batches, params, tabs, min_s=8.0):
    """The same oracle over whole partitions on every host core (one process per core, partitions are independent; the
    reference's own parallelism is a process pool over partitions, py/freddie_segment.py:871-876).  The pool is forked
    (the partitions are inherited, only indices travel) and warmed outside the timed region."""
    import multiprocessing as mp
    global _POOL_JOB
    parts = [p for b in batches for p in b.parts]
    if len(parts) < 2:
        return None
    cores = min(host_cores(), 64, len(parts))
    from oracle import oracle
    oracle.lib()          # built (if stale) and loaded HERE, once: the forked workers inherit it instead of racing to rebuild it
    _POOL_JOB = (parts, params, tabs)
    n_reads = sum(p.n_reads for p in parts)
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_oracle_one, range(min(cores, len(parts))))                  # start-up, page-in
        t0 = time.perf_counter()
        reads = passes = 0
        while time.perf_counter() - t0 < min_s:
            errs = pool.map(_oracle_one, range(len(parts)), chunksize=max(1, len(parts) // (cores * 8)))
            if any(errs):
                raise RuntimeError("oracle failed in the all-cores baseline")
            reads += n_reads
            passes += 1
        dt = time.perf_counter() - t0
    _POOL_JOB = None
    return dict(value=reads / dt, unit="reads/s", cores=cores, kind="port",
                sample="%d pass(es) over %d partitions (%d reads) of the C oracle on %d processes, %.1f s" % (
                    passes, len(parts), reads, cores, dt))


def _gen_split(job):
    idx, kw, d = job
    synth.generate(idx, write_dir=d, **kw)
    return idx


def e2e_leg(workload, params, n_reads_target, threads, gpus=1):
    """Files in -> files out through the drop-in CLI (py/freddie_segment.py): a split directory of the workload's
    partitions (with sequences) in tmpfs, the CLI as a child process (it creates its own GPU contexts -- one worker process
    per GPU when gpus > 1; this process has not touched the GPU yet), wall time of the whole process including interpreter
    and context start-up.  The job is the same at every GPU count (strong scaling).  Two runs: the first pages the
    libraries in, the second is reported."""
    import multiprocessing as mp
    w = dict(synth.WORKLOADS[workload])
    w.pop("n_partitions")
    n_part = max(1, n_reads_target // w["n_reads"])
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    need = n_part * w["n_reads"] * 2400                        # ~1.9 KB of TSV in + ~0.4 KB out per read
    if base and shutil.disk_usage(base).free < 2 * need:
        base = None
    work = tempfile.mkdtemp(prefix="freddie_e2e_", dir=base)
    try:
        split, out = os.path.join(work, "split"), os.path.join(work, "out")
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(min(host_cores(), 32)) as pool:
            pool.map(_gen_split, [(i, w, split) for i in range(n_part)], chunksize=8)
        t_gen = time.perf_counter() - t0
        size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(split) for f in fs)
        def cli(sidecar):
            return [sys.executable, os.path.join(ROOT, "py", "freddie_segment.py"), "-s", split, "-o", out, "-t", str(threads),
                    "--gpus", str(gpus), "--sidecar", sidecar, "-sd", str(params["sigma"]), "-tp", str(params["threshold_rate"])]
        walls = []
        for _ in range(2):
            shutil.rmtree(out, ignore_errors=True)
            t0 = time.perf_counter()
            subprocess.run(cli("off"), check=True, stdout=subprocess.DEVNULL)
            walls.append(time.perf_counter() - t0)
        n_out = sum(len(fs) for _, _, fs in os.walk(out))
        reads = n_part * w["n_reads"]
        res = dict(value=reads / walls[-1], unit="reads/s", reads=reads, partitions=n_part, wall_s=walls, threads=threads,
                   n_gpus=gpus, scaling="strong", input_mb=size / 1e6, output_files=n_out, tmp=base or tempfile.gettempdir(),
                   generate_s=t_gen,
                   what="py/freddie_segment.py -s <split> -o <out> -t %d --gpus %d --sidecar off (-t is per GPU worker): whole "
                        "process wall time, second of two runs" % (threads, gpus))
        # row N2: the same job with binary side-cars beside the TSVs (what a second pass over a split directory meets: a parameter
        # sweep, a re-run).  One untimed run writes them (--sidecar write), two timed runs load them (--sidecar auto); the output
        # files must be byte-identical to the side-car-free run's.
        try:
            import hashlib

            def digest():
                h = hashlib.sha256()
                for dp, _, fs in sorted(os.walk(out)):
                    for f in sorted(fs):
                        with open(os.path.join(dp, f), "rb") as fh:
                            h.update(f.encode()); h.update(fh.read())
                return h.hexdigest()
            plain = digest()
            shutil.rmtree(out, ignore_errors=True)
            t0 = time.perf_counter()
            subprocess.run(cli("write"), check=True, stdout=subprocess.DEVNULL)
            t_write = time.perf_counter() - t0
            sc_walls = []
            for _ in range(2):
                shutil.rmtree(out, ignore_errors=True)
                t0 = time.perf_counter()
                subprocess.run(cli("auto"), check=True, stdout=subprocess.DEVNULL)
                sc_walls.append(time.perf_counter() - t0)
            sc_mb = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(split) for f in fs if f.endswith(".fsc")) / 1e6
            res["sidecar"] = dict(value=reads / sc_walls[-1], unit="reads/s", wall_s=sc_walls, write_pass_s=t_write, sidecar_mb=sc_mb,
                                  same_bytes_as_plain=digest() == plain,
                                  what="the same command with --sidecar auto on a split directory whose .fsc side-cars exist (written by an "
                                       "untimed --sidecar write run): second of two runs")
        except Exception as exc:                          # (the plain figure must not depend on this leg)
            res["sidecar"] = dict(error="%s: %s" % (type(exc).__name__, exc))
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


# ---- rows N3 / N4 ------------------------------------------------------------------------------------------------------
def cpu_baseline_cluster(uniq, budget_s=12.0):
    """CPU leg of the row-N3 measurement (tools/cluster_bench.py): the oracle's Python restatement of the pairwise
    compatibility test + pruning on a bounded sample (the first 300 unique reads of successive tints)."""
    from oracle import cluster_oracle
    t0 = time.perf_counter()
    done = 0
    for u in uniq:
        sub = u[:300]
        cluster_oracle.prune(len(sub), cluster_oracle.compat_edges(sub))
        done += len(sub) * (len(sub) - 1) // 2
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="pairs/s", cores=1, kind="port",
                sample="%d pairs (first 300 unique reads of successive tints), Python oracle, %.1f s" % (done, dt))


def cpu_baseline_isoforms(per, n_seg, window, budget_s=10.0):
    """CPU leg of the row-N4 measurement (tools/isoforms_bench.py): the Python oracle's isoforms_cons + correct_boundaries
    on synthetic tints of the same shape until the budget is spent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import isoforms_util as iu
    from oracle import isoforms_oracle
    t0 = time.perf_counter()
    done = k = 0
    while time.perf_counter() - t0 < budget_s:
        isos, segments, reads = iu.random_job(100 + k, 8, per, n_seg)
        isoforms_oracle.isoforms_cons(isos, segments, reads)
        for side in ("starts", "ends"):
            isoforms_oracle.correct_boundaries(side, isos, reads, 0.5, window)
        done += len(reads)
        k += 1
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="reads/s", cores=1, kind="port",
                sample="%d reads in %d synthetic tints (generation included), Python oracle, %.1f s" % (done, k, dt))


NEXT_ROW_WORKLOADS = ("cluster-many", "cluster-big", "isoforms")     # their own metrics


def run_next_row(args):
    """The measurements of rows N3 / N4 live in tools/ (GPU side only); the CPU-baseline legs, which use the oracle,
    are here."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    if args.workload == "isoforms":
        import isoforms_bench
        out = isoforms_bench.run(steps=min(args.steps, 10), cpu_baseline=None if args.no_cpu_baseline else cpu_baseline_isoforms)
    else:
        import cluster_bench
        out = cluster_bench.run(args.workload.split("-")[1], steps=min(args.steps, 10),
                                cpu_baseline=None if args.no_cpu_baseline else cpu_baseline_cluster)
    print(json.dumps(out))


def next_rows_summary():
    """Short runs of the N3 / N4 measurements (tools/cluster_bench.py, tools/isoforms_bench.py; no CPU legs): value, unit, kernel times and
    the roofline fraction of each row's hot kernel."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    rows = {}
    try:
        import cluster_bench
        import isoforms_bench
        for name, fn in (("cluster-many", lambda: cluster_bench.run("many", steps=3)), ("cluster-big", lambda: cluster_bench.run("big", steps=2)),
                         ("isoforms", lambda: isoforms_bench.run(steps=3))):
            t0 = time.perf_counter()
            r = fn()
            rows[name] = {"metric": r.get("metric"), "value": r.get("value"), "unit": r.get("unit"), "config": r.get("config"),
                          "call_wall_ms": r.get("call_wall_ms"), "kernel_ms": r.get("kernel_ms"),
                          "roofline": {k: (r.get("roofline") or {}).get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac")},
                          "leg_s": time.perf_counter() - t0}
    except Exception as exc:                              # the headline must not depend on these
        rows["error"] = "%s: %s" % (type(exc).__name__, exc)
    return rows


def committed_counters(workload, source_hash=None):
    """Counter figures that need their own rocprofv3 --pmc passes (profiles/traffic.json, written by profiles/make_traffic.py
    from the passes of tools/profile_round.sh): HBM bytes per launch and VALU utilisation of the scoring kernels.  They are
    not measured in this run, so they are reported only when the file says it was made from the very library this run has
    loaded (``source_hash`` = fseg_source_hash()); otherwise None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None
    if source_hash is not None and doc.get("source_hash") != source_hash:
        return None
    return doc.get(workload)


def reference_cpu(workload, value, cpu_one, e2e=None):
    """The REFERENCE's own CPU figure for this workload, as recorded by tools/ref_cpu.py in the BUILD CONTAINER (the reference
    never travels to the GPU box): the unmodified reference CLI on the first partitions of the very split directories this
    benchmark's generator writes, `-t 1` and `-t <cores>`.  Quoted here next to cpu_baseline (the C port timed on this box),
    with the ratios a reader wants: this run's value over the reference at one core and at all of the container's cores, and
    the port's one-core rate over the reference's (what "kind: port" hides).  Different machines: the ratios are reported,
    not claimed as a same-box speed-up."""
    try:
        with open(os.path.join(ROOT, "profiles", "reference_cpu.json")) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None
    run = (doc.get("runs") or {}).get(workload)
    if not run:
        return None
    by = {int(t): r for t, r in run["by_threads"].items()}
    t1, tn = by.get(1), by.get(max(by))
    out = {"kind": "reference", "where": "%s, %d vCPU (%s)" % (doc.get("where", "build container"), doc.get("cores", 0), doc.get("cpu", "?")),
           "versions": "python %s, numpy %s, scipy %s" % (doc.get("python"), doc.get("numpy"), doc.get("scipy")),
           "sample": "%d partitions (%d reads) of %s, indices 0.., params %s; input sha256 %s" % (
               run["partitions"], run["reads"], workload, run["params"], run["input_sha256"][:16]),
           "unit": "reads/s", "by_threads": {str(t): r["reads_per_s"] for t, r in sorted(by.items())},
           "outputs_identical_across_threads": all(r.get("same_bytes_as_first") for r in by.values()),
           "source": "profiles/reference_cpu.json (tools/ref_cpu.py, %s)" % doc.get("date")}
    if t1:
        out["value"] = t1["reads_per_s"]; out["cores"] = 1
        out["this_run_over_reference_1_core"] = value / t1["reads_per_s"]
        if cpu_one and cpu_one.get("value"):
            out["port_over_reference_per_core"] = cpu_one["value"] / t1["reads_per_s"]
    if tn and max(by) > 1:
        out["this_run_over_reference_%d_cores" % max(by)] = value / tn["reads_per_s"]
        # like for like with the reference's CLI (files in, files out, process start-up included): the drop-in CLI's e2e leg
        if e2e and e2e.get("value"):
            out["e2e_cli_over_reference_%d_cores" % max(by)] = e2e["value"] / tn["reads_per_s"]
    return out


def stage_algorithmic_bytes(batch, sizes, part_final_off, scoring_bytes):
    """SURVEY.md 8(d) / BASELINE.md section 4: algorithmic bytes per stage of one batch, every input read once and every output
    written once in the reference's own dtypes (P positions, I exons, R read reps, N candidates, K intervals, F final
    positions; per partition where the formula is a product)."""
    P, N = sizes["n_positions"], sizes["n_cand"]
    I = len(batch.arrays["ex_ts"])
    R = len(batch.arrays["rep_weight"])
    labels = 0
    for p, part in enumerate(batch.parts):
        F, K, Rp = int(part_final_off[p + 1] - part_final_off[p]), len(part.iv_start), part.n_reps
        labels += 4 * (F + K) * Rp + max(F - 1, 0) * Rp
    return {"histogram": 8 * I + 4 * R + 8 * P,            # S1  8I + 4R -> 8P                  (:662-673)
            "smooth": 16 * P,                              # S2  16 B / position                (:755)
            "threshold": 8 * P,                            # S3a mean + vf * std of Y > 0       (:757-759)
            "candidates": 8 * P + 4 * N,                   # S3b 8P -> 4N                        (:615-621)
            "interval_scoring": scoring_bytes,             # S4 + S5  4(N+K)R + 4R per partition (:188-246, :475-568)
            "labels": labels}                              # S7  4(F+K)R -> (F-1)R               (:808-830)


# ---- GPU legs -----------------------------------------------------------------------------------------------------------
def one_shot_steps(ctxs, batches, order, collect=None):
    """The steps `order` (batch indices), alternating between the contexts, one host thread per context."""
    errors = []

    def worker(k):
        ctx = ctxs[k]
        try:
            for si in range(k, len(order), len(ctxs)):
                b = batches[order[si]]
                ctx.upload(**b.arrays)
                ctx.run()
                res = ctx.results(packed=True)
                if collect is not None:
                    collect(si, order[si], ctx, res)
        except BaseException as exc:                      # noqa: BLE001  (re-raised by the caller)
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(ctxs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def scoring_roofline(alg_bytes, score_ms, committed, extra=None):
    achieved = alg_bytes / (score_ms * 1e-3) / 1e9 if score_ms > 0 else 0.0
    r = {"kernel": "interval-scoring stage: k_solve<16|32|60> (coverage, pair labels, in/out counts of a problem, in LDS) + k_dpw<16|32|60> (its DP, "
                   "one wave) + k_wave<8> (tiny problems whole) (+ k_gate, one wave) where every problem of the batch sees <= 511 reads, "
                   "else the arena path: k_cov + k_score<16|32|60> + k_dp (+ k_wave<8>)",
         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": (committed or {}).get("traffic_bytes"), "traffic_source": "committed PMC pass (profiles/traffic.json)" if committed else None,
         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": score_ms}
    if extra:
        r.update(extra)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; a step is the whole job (config4 on one GPU: 8 batches, ~7 ms)")
    ap.add_argument("--warmup", type=int, default=3, help="untimed steps before them (at least one; more until the GPU has been busy for 0.5 s)")
    ap.add_argument("--workload", default="config4", choices=sorted(synth.WORKLOADS) + list(NEXT_ROW_WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip both CPU-oracle legs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the CLI end-to-end leg")
    ap.add_argument("--no-extras", action="store_true", help="skip config2's roofline, the replay and the HBM-resident legs")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the short runs of rows N3 / N4 (cluster-many, cluster-big, isoforms) in `next_rows`")
    ap.add_argument("--e2e-reads", type=int, default=2000000, help="reads of the end-to-end job (default: the whole 2 M-read job)")
    ap.add_argument("--contexts", type=int, default=8, help="contexts per GPU the steps alternate between (each on one stream while "
                    "the others have work in flight: two per hardware queue; 4: -20 %%, 16: -30 %%)")
    args = ap.parse_args()
    if args.workload in NEXT_ROW_WORKLOADS:
        return run_next_row(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    params = PARAMS["config5" if args.workload == "config5" else "default"]
    tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0),
                w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
                h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
    batches = build_batches(args.workload, rank, args.gpus)
    n_b = len(batches)
    wl = synth.WORKLOADS[args.workload]
    job_reads, job_parts = wl["n_partitions"] * wl["n_reads"], wl["n_partitions"]

    # everything that forks or starts a process comes before this process initialises the GPU
    cpu_one = cpu_all = e2e = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_all = cpu_baseline_all_cores(batches, params, tabs)
    # The end-to-end leg: the same job as files through the CLI, which scatters it over N worker processes (one per GPU).
    # Rank 0 runs it while no rank has touched a GPU yet; the others wait for it.
    if not args.no_e2e and args.workload != "config2":
        done = os.path.join(tempfile.gettempdir(), "freddie_bench_e2e_%s_%d.done" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))     # (the launcher is every rank's parent)
        if rank == 0:
            try:
                if os.path.exists(done):
                    os.remove(done)
                e2e = e2e_leg(args.workload, params, min(args.e2e_reads, job_reads), threads=max(1, min(16, host_cores() // world)), gpus=world)
            except Exception as exc:                     # the headline must not depend on this leg (e.g. no room in tmpfs)
                e2e = dict(error="%s: %s" % (type(exc).__name__, exc))
            finally:
                if world > 1:
                    open(done, "w").close()
        else:
            t_wait = time.time()
            while not os.path.exists(done) and time.time() - t_wait < 900:
                time.sleep(0.05)
    config2_batch = None
    extra_parts = {}
    if rank == 0 and not args.no_extras and args.workload != "config2":
        w2 = dict(synth.WORKLOADS["config2"]); w2.pop("n_partitions")
        g = synth.generate(0, with_seq=False, **w2)
        config2_batch = Batch([pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True)])
    if rank == 0 and not args.no_extras and args.workload == "config4":
        for wname in ("config3", "config5"):             # one batch of each (the line carries every config's stage fraction)
            wx = dict(synth.WORKLOADS[wname]); wx.pop("n_partitions")
            per_x = max(1, min(synth.WORKLOADS[wname]["n_partitions"], BATCH_READS // wx["n_reads"]))     # (config3's whole job is 500 partitions)
            extra_parts[wname] = []
            for i in range(per_x):
                g = synth.generate(i, with_seq=False, **wx)
                extra_parts[wname].append(pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))

    import torch
    from freddie_amd import _lib
    dist = None
    coll = None                                         # the group the barriers and the reductions of (time, reads, checksums) use
    coll_note = None
    # (FREDDIE_BENCH_FORCE_DIST=1: the process group also for one rank -- how the RCCL branch is rehearsed on a one-GPU box)
    if world > 1 or os.environ.get("FREDDIE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                   # (FREDDIE_BENCH_FORCE_DIST without a launcher: a group of one)
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29511")):
                os.environ.setdefault(k, v)
        # The job has no data-path collective: the process group carries the barriers around the timed regions and
        # one reduction of eleven numbers.  The default group is gloo (host tensors; it comes up wherever TCP to 127.0.0.1 does), and
        # the barriers and reductions go over RCCL, one GPU per rank, when every rank has seen an RCCL all-reduce of its own work:
        # a rank whose RCCL does not come up must not cost the node its whole run.  FREDDIE_BENCH_BACKEND=gloo: no RCCL at all -- the
        # rehearsal of the multi-rank flow on a box with fewer GPUs than ranks (the ranks share the GPUs there are).
        backend = os.environ.get("FREDDIE_BENCH_BACKEND", "nccl")
        if backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo")
        if backend != "gloo":
            ok, why = 1.0, ""
            try:
                coll = dist.new_group(backend="nccl")
                probe = torch.ones(1, dtype=torch.float64, device="cuda")
                dist.all_reduce(probe, op=dist.ReduceOp.SUM, group=coll)
                torch.cuda.synchronize()
                if int(probe[0].item()) != dist.get_world_size():
                    ok, why = 0.0, "RCCL all-reduce returned %r on rank %d" % (probe[0].item(), rank)
            except Exception as exc:                     # noqa: BLE001  (whatever RCCL raises here, the run goes on over gloo)
                ok, why = 0.0, "%s: %s" % (type(exc).__name__, str(exc)[:200])
            agreed = torch.tensor([ok], dtype=torch.float64)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)            # (over gloo: every rank takes the same branch)
            if agreed[0] < 1.0:
                coll = None
                coll_note = "gloo (RCCL did not come up on every rank%s)" % ((": " + why) if why else "")
                print("bench.py rank %d: %s" % (rank, coll_note), file=sys.stderr)
    else:
        torch.cuda.set_device(local_rank)

    ctxs = [_lib.Context(local_rank) for _ in range(max(1, args.contexts))]
    for ctx in ctxs:
        ctx.set_params(**params, **tabs)
        ctx.set_profiling(2)                            # two events per run: the bracket around the interval-scoring stage

    def barrier():
        if dist is not None:
            dist.barrier(group=coll)
        torch.cuda.synchronize()

    # warm-up: W untimed steps (every batch of the share passes W times: buffers reach their final sizes; a batch's
    # algorithmic bytes are noted here, outside the timed region: they are a property of the batch) ...
    def note_alg(si, bi, ctx, res):
        if batches[bi].alg_bytes is None:
            batches[bi].alg_bytes = ctx.scoring_algorithmic_bytes()
        if batches[bi].label_popcount is None:
            # the labels' CONTENT, out here where it costs nothing timed: set bits of the two-bit labels = labels that are '1' or
            # '2' (padding is zero), a property of the batch's partitions however they are batched
            batches[bi].label_popcount = popcount_bytes(res[3])
        if batches[bi].path_bytes is None:
            batches[bi].path_bytes = sum(stage_algorithmic_bytes(batches[bi], ctx.sizes(), res[0], batches[bi].alg_bytes).values())
    n_warm_steps = max(1, args.warmup)
    t_warm = time.perf_counter()
    if n_b:
        one_shot_steps(ctxs, batches, [i % n_b for i in range(max(n_warm_steps * PASSES_PER_STEP * n_b, 2 * len(ctxs)))], note_alg)
        # ... and until the GPU has been busy for half a second (a fresh box runs its first 100 ms at lower clocks)
        while time.perf_counter() - t_warm < 0.5 and n_warm_steps < 4096:
            one_shot_steps(ctxs, batches, list(range(n_b)), note_alg)
            n_warm_steps += 1

    score_ms_total = [0.0]
    alg_total = [0]
    acc_lock = threading.Lock()
    checksum = [0]

    def collect(si, bi, ctx, res):
        ms = ctx.stage_ms()
        with acc_lock:
            score_ms_total[0] += ms["interval_scoring"]
            alg_total[0] += batches[bi].alg_bytes
            # the batch's results, read where they lie in host memory: the number of final positions, the number of label bytes
            # and the SUM of the final positions (content; ~150 k integers per batch) -- sums over partitions, so the job's total
            # does not depend on how it is cut into batches or scattered over ranks.  The labels' content (19 MB per batch at
            # two bits per label) is summed in the warm-up pass instead: label_popcount below
            checksum[0] += int(res[0][-1]) + int(res[2][-1]) + int(res[1].sum(dtype=np.int64))

    # the timed steps: K passes over the share, batch after batch, as one stream of work for the contexts (step boundaries
    # are not barriers: the CLI does not stop between batches either); consecutive batch-steps go to consecutive contexts
    n_passes = args.steps * PASSES_PER_STEP
    order = [i % n_b for i in range(n_passes * n_b)] if n_b else []
    barrier()
    t0 = time.perf_counter()
    if order:
        one_shot_steps(ctxs, batches, order, collect)
    barrier()
    dt = time.perf_counter() - t0
    n_reads = sum(batches[bi].n_reads for bi in order)

    # ---- `value`: the same K passes with the inputs RESIDENT IN HBM when the timed part starts (the contract's reading: a rate that
    # carries the host buffers across PCIe is reported beside it -- the region above, `value_h2h` -- and is never `value`).
    # Every context holds a batch of the rank's share (context k: batch k mod n_b, so a rank with fewer batches than contexts --
    # N GPUs share the eight batches -- still keeps all its contexts busy); a pass = every batch of the share run once, everything
    # recomputed from the inputs (histogram .. labels), results left in HBM; K passes dealt out over the contexts that hold each
    # batch, one host thread per context.  Shares of more batches than contexts go group by group, the uploads between the timed
    # parts.  The checksum is taken from one fetch per context after the timed part.
    n_ctx = len(ctxs)
    dev_t = "cuda" if coll is not None else "cpu"
    n_groups = (n_b + n_ctx - 1) // n_ctx
    if dist is not None:
        tg = torch.tensor([float(n_groups)], dtype=torch.float64, device=dev_t)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX, group=coll)  # (every rank meets the same barriers)
        n_groups = int(tg[0])
    dt_res, reads_res, chk_res, pop_res, sync_timeouts, forked_runs = 0.0, 0, 0, 0, 0, 0
    for gi in range(n_groups):
        grp = list(range(gi * n_ctx, min(n_b, (gi + 1) * n_ctx)))
        runs = [0] * n_ctx
        if grp:
            for k, c2 in enumerate(ctxs):
                c2.upload(**batches[grp[k % len(grp)]].arrays)
                c2.run(); c2.sync()                               # (arenas sized, the replay set up: outside the timed part)
            for i in range(len(grp)):
                holders = [k for k in range(n_ctx) if k % len(grp) == i]
                for pos, k in enumerate(holders):
                    runs[k] = n_passes // len(holders) + (1 if pos < n_passes % len(holders) else 0)

        def run_many(k):
            for _ in range(runs[k]):
                ctxs[k].run()
                ctxs[k].sync()
        th = [threading.Thread(target=run_many, args=(k,)) for k in range(n_ctx) if runs[k]]
        barrier()
        t0 = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        barrier()
        dt_res += time.perf_counter() - t0
        for k in range(n_ctx):
            if runs[k]:
                res = ctxs[k].results(packed=True)
                bk = batches[grp[k % len(grp)]]
                reads_res += runs[k] * bk.n_reads
                chk_res += runs[k] * (int(res[0][-1]) + int(res[2][-1]) + int(res[1].sum(dtype=np.int64)))
                # the labels the LAST pass of the region left in HBM, by content: their popcount must be the batch's (warm-up pass,
                # host memory -> host memory on another context)
                pop = popcount_bytes(res[3])
                if pop != bk.label_popcount:
                    raise SystemExit("resident region: context %d left labels with popcount %d, the batch's is %d" % (k, pop, bk.label_popcount))
                pop_res += pop
            sy = ctxs[k].tap("sync")
            if len(sy) > 7:                                    # (a library of an earlier round under FSEG_LIB has no such counters)
                sync_timeouts += int(sy[6]); forked_runs += int(sy[7])

    label_pop = sum(b.label_popcount or 0 for b in batches)
    path_bytes = sum(b.path_bytes or 0 for b in batches)                 # one pass over this rank's share
    t = torch.tensor([dt, float(n_reads), float(checksum[0]), float(label_pop), dt_res, float(reads_res), float(chk_res), float(path_bytes),
                      float(pop_res), float(sync_timeouts), float(forked_runs)], dtype=torch.float64, device=dev_t)
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=coll)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM, group=coll)
        dt_h2h, reads_h2h, checksum_h2h, label_pop_all = float(tmax[0]), float(tsum[1]), int(tsum[2]), int(tsum[3])
        dt_max, total_reads, checksum_all = float(tmax[4]), float(tsum[5]), int(tsum[6])
        path_bytes_all, pop_res_all, sync_timeouts_all, forked_runs_all = float(tsum[7]), int(tsum[8]), int(tsum[9]), int(tsum[10])
    else:
        dt_h2h, reads_h2h, checksum_h2h, label_pop_all = dt, float(n_reads), checksum[0], label_pop
        dt_max, total_reads, checksum_all = dt_res, float(reads_res), chk_res
        path_bytes_all, pop_res_all, sync_timeouts_all, forked_runs_all = float(path_bytes), pop_res, sync_timeouts, forked_runs

    if rank == 0:
        lib_hash = _lib.load().fseg_source_hash().decode()
        committed = committed_counters(args.workload, lib_hash)
        sizes = ctxs[0].sizes()
        _, per = plan_job(args.workload, 0, args.gpus)
        reps = 20
        ctx = ctxs[0]
        for c2 in ctxs:
            c2.sync()

        # ---- the interval-scoring stage alone on the GPU: one context, one resident batch of the job, replayed
        ctx.set_profiling(2)
        ctx.upload(**batches[0].arrays)
        ctx.run(); ctx.sync()
        for _ in range(3):
            ctx.run(); ctx.sync()
        t0 = time.perf_counter()
        sc_all = []
        for _ in range(reps):
            ctx.run(); ctx.sync()
            sc_all.append(ctx.stage_ms()["interval_scoring"])
        dt_r = time.perf_counter() - t0
        sc_med = float(np.median(sc_all))                      # (the median, like the per-kernel medians of profiles/)
        roofline = scoring_roofline(batches[0].alg_bytes, sc_med, committed,
                                    {"measured": "HIP events around the stage's launches on the library's streams; one context, one resident "
                                                 "batch of the job (%d reads) replayed %d times: the stage alone on the GPU (median; mean %.4f ms). "
                                                 "The stream plan and k_gate apply to this case only: in the timed steps (value) eight contexts "
                                                 "take turns and each keeps its stage on one stream -- that bracket is roofline_concurrent"
                                                 % (batches[0].n_reads, reps, float(np.mean(sc_all)))})

        # ---- every stage alone on the GPU: distinct batches through the first-run path (plain launches, events around every stage)
        ctx.set_profiling(True)
        st_acc, st_bytes, n_st = {}, {}, 0
        for b in batches[:4]:
            for _ in range(2):                                        # (the second pass of a batch: arenas already sized)
                ctx.upload(**b.arrays)
                ctx.run(); ctx.sync()
            ms = ctx.stage_ms()
            res = ctx.results(packed=True)
            by = stage_algorithmic_bytes(b, ctx.sizes(), res[0], ctx.scoring_algorithmic_bytes())
            n_st += 1
            for k, v in ms.items():
                st_acc[k] = st_acc.get(k, 0.0) + v
            for k, v in by.items():
                st_bytes[k] = st_bytes.get(k, 0) + v
        roofline_stages = {}
        for k, v in st_acc.items():
            if k.startswith("graph_") or n_st == 0:
                continue
            ms_k = v / n_st
            e = {"ms": ms_k}
            if k == "interval_scoring":                               # (with its DP: fused batches have none of their own)
                e["ms"] = ms_k = (v + st_acc.get("dp", 0.0)) / n_st
            if k in st_bytes and ms_k > 0:
                gbs = st_bytes[k] / n_st / (ms_k * 1e-3) / 1e9
                e.update(algorithmic_bytes=st_bytes[k] // n_st, achieved=gbs, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
            roofline_stages[k] = e
        ctx.set_profiling(2)

        out = {
            "metric": "reads segmented/sec (whole node)",
            "value": total_reads / dt_max,
            "unit": "reads/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_steps_run": n_warm_steps,
            "ms_per_step": dt_max / args.steps * 1e3,
            "passes_per_step": PASSES_PER_STEP,
            "ms_per_pass": dt_max / max(1, n_passes) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "dtype_detail": "u32 bit-planes + popcount for scoring, int32 (int64 for very large partitions) DP sums, f64 Gaussian smoothing / threshold",
            "data": "synthetic",
            "value_is": "inputs resident in HBM when the timed part starts (the contract's reading; rounds 1-4 reported the host memory -> host "
                        "memory rate here, now `value_h2h`): a step = fseg_run of every batch of the rank's share of the fixed job, distinct "
                        "batches, %d contexts per GPU each holding a batch of the share, everything recomputed from the inputs each pass, results "
                        "left in HBM (checked: one fetch per context after the timed part)" % len(ctxs),
            "timed_s": dt_max,
            # rounds 1-4's `value` (pinned host memory -> results in host memory) continues HERE, at the top level
            "value_h2h": reads_h2h / dt_h2h if dt_h2h > 0 else 0.0,
            "ms_per_step_h2h": dt_h2h / args.steps * 1e3,
            "ms_per_pass_h2h": dt_h2h / max(1, n_passes) * 1e3,
            "timed_s_h2h": dt_h2h,
            "value_h2h_is": "the same passes host memory -> host memory: fseg_upload + fseg_run + fseg_results_packed of every batch of the share, "
                            "%d contexts per GPU taking turns (PCIe-inclusive; what the CLI's workers do per batch); the figure BENCH_r01..r04 "
                            "report as `value`" % len(ctxs),
            "result_checksum_h2h": checksum_h2h,
            # the fraction of the HBM roofline that belongs to `value` / `value_h2h`: SURVEY 8(d)'s algorithmic bytes of EVERY stage of the
            # path (histogram, smoothing, threshold, candidates, interval scoring, labels) of one pass over the job, over the time of a pass
            "roofline_path": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS * args.gpus, "algorithmic_bytes_per_pass": path_bytes_all,
                              "achieved": path_bytes_all * n_passes / dt_max / 1e9 if dt_max > 0 else 0.0,
                              "frac": path_bytes_all * n_passes / dt_max / 1e9 / (HBM_PEAK_GBS * args.gpus) if dt_max > 0 else 0.0,
                              "achieved_h2h": path_bytes_all * n_passes / dt_h2h / 1e9 if dt_h2h > 0 else 0.0,
                              "frac_h2h": path_bytes_all * n_passes / dt_h2h / 1e9 / (HBM_PEAK_GBS * args.gpus) if dt_h2h > 0 else 0.0,
                              "what": "sum of the stages' algorithmic bytes (SURVEY 8d, every input once, every output once, reference dtypes) of one pass "
                                      "over the job / time of a pass / (8 TB/s x GPUs): `frac` for `value` (inputs resident), `frac_h2h` for `value_h2h`"},
            # device-side waiters (k_wait_word) of the scoring stage that reached their time limit in this process's contexts (each costs the
            # limit + a rerun of its batch): must be 0; forked_runs = runs that owned the device (side streams in use)
            "sync_timeouts": sync_timeouts_all,
            "forked_runs": forked_runs_all,
            # what carried the barriers and the reduction of the ranks' (time, reads, checksums): RCCL one GPU per rank, or gloo
            "rank_sync": None if dist is None else (coll_note or ("rccl" if coll is not None else "gloo")),
            "result_label_popcount_resident": pop_res_all,
            "config": {"workload": args.workload, "partitions": job_parts, "reads": job_reads,
                       "batches_per_step_rank0": n_b, "partitions_per_batch": per, "reads_per_batch": batches[0].n_reads if n_b else 0,
                       "reads_per_partition": wl["n_reads"], "contexts_per_gpu": len(ctxs), "scatter": "static LPT over %d rank(s), no collective" % args.gpus,
                       "candidates_last_batch": sizes["n_cand"], "dp_problems_last_batch": sizes["n_problems"],
                       "positions_last_batch": sizes["n_positions"], "params": {k: params[k] for k in ("sigma", "threshold_rate")}},
            "roofline": roofline,
            "roofline_concurrent": scoring_roofline(alg_total[0] / max(1, len(order)), score_ms_total[0] / max(1, len(order)), committed,
                                                    {"measured": "the same bracket inside the timed steps (the other contexts' kernels run inside it)"}),
            "roofline_stages": roofline_stages,
            "roofline_stages_from": "%d distinct batches, one context alone on the GPU, first-run path (plain launches; events around every stage); "
                                    "interval_scoring includes the DP" % n_st,
            "valu_util": (committed or {}).get("valu_util"),
            "valu_util_source": (committed or {}).get("valu_source"),
            "library_source_hash": lib_hash,
            # counts of final positions and label bytes + the sum of the final positions, of every batch of every timed step,
            # summed over the ranks: a property of the job (every partition exactly once per step), whatever N and however a
            # rank's share is cut into batches; result_label_popcount: the labels that are '1' or '2' in one pass over the job
            # (counted in the warm-up pass, outside the timed region)
            "result_checksum": checksum_all,
            "result_checksum_per_pass": checksum_all // max(1, n_passes),
            "result_label_popcount": label_pop_all,
        }
        out["value_resident_replay"] = {"value": batches[0].n_reads * reps / dt_r, "unit": "reads/s", "ms_per_step": dt_r / reps * 1e3,
                                        "what": "replay of one resident batch of the job (%d reads): no copies, no arena sizing (a batch, not the job)" % batches[0].n_reads}
        if not args.no_extras:
            # inputs resident in HBM before the timed part, every batch run ONCE on the first-run path: one batch per
            # context uploaded, then all of them run (one host thread per context, as in the timed steps), results left in HBM
            n_h = min(len(ctxs), n_b)
            for c2, b in zip(ctxs[:n_h], batches):
                c2.upload(**b.arrays)
                c2.sync()
            torch.cuda.synchronize()

            def run_one(c2):
                c2.run()
                c2.sync()
            th = [threading.Thread(target=run_one, args=(c2,)) for c2 in ctxs[:n_h]]
            t0 = time.perf_counter()
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            dt_h = time.perf_counter() - t0
            out["value_hbm_resident"] = {"value": sum(b.n_reads for b in batches[:n_h]) / dt_h, "unit": "reads/s", "ms_per_batch": dt_h / n_h * 1e3,
                                         "what": "%d distinct batches uploaded first (one per context), then each run once, concurrently: "
                                                 "first-run path (sized arenas, plain launches), results left in HBM, no copies in the "
                                                 "timed part" % n_h}
            # BASELINE configs[1]: one 50 k-read partition.  Its problems take the arena path, where coverage (k_cov: inside the
            # interval_scoring bracket when the stages are bracketed on plain launches) and the DP (k_dp, the dp stage) are launches
            # of their own: the figure that compares with the other configs' stage (which holds all three) is interval_scoring + dp;
            # k_score alone (the bracket of the replay that runs everything else as graphs) is the sub-field.
            if config2_batch is not None:
                ctx.set_profiling(3)
                ctx.upload(**config2_batch.arrays)
                ctx.run(); ctx.sync()
                alg2 = ctx.scoring_algorithmic_bytes()
                for _ in range(3):
                    ctx.run(); ctx.sync()
                t0 = time.perf_counter()
                acc2 = []
                for _ in range(reps):
                    ctx.run(); ctx.sync()
                    ms = ctx.stage_ms()
                    acc2.append((ms["scoring_prep"], ms["interval_scoring"], ms["dp"]))
                dt_2 = time.perf_counter() - t0
                med2 = np.median(np.asarray(acc2), axis=0)
                ctx.set_profiling(2)                                  # (one stream: graph | events | k_score | graph)
                for _ in range(3):
                    ctx.run(); ctx.sync()
                ks = []
                for _ in range(reps):
                    ctx.run(); ctx.sync()
                    ks.append(ctx.stage_ms()["interval_scoring"])
                r2 = scoring_roofline(alg2, float(med2[1] + med2[2]), committed_counters("config2", lib_hash))
                r2.update(workload="config2", reads=config2_batch.n_reads, ms_per_step=dt_2 / reps * 1e3,
                          what="one partition, 50 k reads x ~2 k candidates, resident, %d runs with plain launches and events around every stage "
                               "(median): interval_scoring (pair thresholds + k_cov + k_score) + dp (k_dp); scoring_prep (the problem "
                               "list) is outside, as on the other configs" % reps,
                          stage_ms={"scoring_prep": float(med2[0]), "interval_scoring": float(med2[1]), "dp": float(med2[2])},
                          k_score_only=scoring_roofline(alg2, float(np.median(ks)), None))
                out["roofline_config2"] = r2
                ctx.set_profiling(2)
            # BASELINE configs[2] and [4]: one resident batch each (250 partitions of 1 000 reads; config5 under sigma 3, tau 0.8)
            for wname, parts_x in extra_parts.items():
                px = PARAMS["config5" if wname == "config5" else "default"]
                tx = dict(w_main=tables.gaussian_half_kernel(px["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(px["sigma"], 1.0),
                          h_table=np.asarray(tables.smooth_threshold(px["threshold_rate"]), np.float64))
                bx = Batch(parts_x)
                ctx.set_params(**px, **tx)
                ctx.upload(**bx.arrays)
                ctx.run(); ctx.sync()
                algx = ctx.scoring_algorithmic_bytes()
                for _ in range(3):
                    ctx.run(); ctx.sync()
                scx = []
                for _ in range(reps):
                    ctx.run(); ctx.sync()
                    scx.append(ctx.stage_ms()["interval_scoring"])
                rx = scoring_roofline(algx, float(np.median(scx)), committed_counters(wname, lib_hash))
                rx.update(workload=wname, reads=bx.n_reads, params={k: px[k] for k in ("sigma", "threshold_rate")},
                          what="one resident batch of %d partitions replayed %d times, the stage alone on the GPU (median)" % (len(parts_x), reps))
                out["roofline_" + wname] = rx
            ctx.set_params(**params, **tabs)
            # The same stage with the rank's WHOLE share as one launch (config4 on one GPU: the 2 M-read job, 4 000 partitions, one batch):
            # the fork / join and the tail of the stage are paid once per launch (250 k-read batches: 0.34-0.35 of the roofline, 1 M: 0.354,
            # the job: 0.38; `value` itself wants eight contexts that fill each other's gaps, and 8 x 2 M reads would be 190 GB of HBM).
            if args.workload == "config4" and n_b > 1:
                bw = Batch([p_ for b_ in batches for p_ in b_.parts])
                ctx.upload(**bw.arrays)
                ctx.run(); ctx.sync()
                algw = ctx.scoring_algorithmic_bytes()
                for _ in range(2):
                    ctx.run(); ctx.sync()
                scw, t0 = [], time.perf_counter()
                for _ in range(10):
                    ctx.run(); ctx.sync()
                    scw.append(ctx.stage_ms()["interval_scoring"])
                dt_w = (time.perf_counter() - t0) / 10
                rw = scoring_roofline(algw, float(np.median(scw)), None)
                rw.update(workload=args.workload, reads=bw.n_reads, partitions=len(bw.parts), ms_per_replay=dt_w * 1e3,
                          reads_per_s_one_context=bw.n_reads / dt_w if dt_w > 0 else 0.0,
                          what="the rank's whole share as ONE resident batch, one context, replayed 10 times: the stage alone on the GPU (median); "
                               "`roofline` is the same stage on one of the job's batches")
                out["roofline_whole_job"] = rw
                del bw
        if not args.no_cpu_baseline:
            cpu_one = cpu_baseline(batches, params, tabs)
        out["cpu_baseline"] = cpu_one
        out["cpu_baseline_all_cores"] = cpu_all
        # (the ratios against the reference's CLI use the PCIe-inclusive rate: the reference's figure includes its I/O)
        out["reference_cpu"] = reference_cpu(args.workload, out["value_h2h"], cpu_one, e2e)
        if out["reference_cpu"]:
            out["reference_cpu"]["ratios_use"] = "value_h2h (host memory -> host memory)"
        out["e2e"] = e2e
        out["e2e_sidecar"] = (e2e or {}).get("sidecar")
        if not args.no_next_rows and not args.no_extras and args.gpus == 1:
            # rows N3 / N4 of SURVEY 8(f), their own metrics, a few steps each on this GPU (the full runs: --workload cluster-many | cluster-big |
            # isoforms): the line that is recorded carries them, not only profiles/
            for ctx in ctxs:
                ctx.close()
            ctxs = []
            out["next_rows"] = next_rows_summary()
        print(json.dumps(out))
    for ctx in ctxs:
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
