#!/usr/bin/env python3
"""Benchmark of the segmentation hot path on MI355X.

A *step* is one pass of the whole device pipeline (splice histogram -> ... -> labels) over one
batch of synthetic partitions that is already resident in HBM.  At N GPUs every rank owns its own
partitions (static scatter, no collectives: partitions share nothing), so scaling is weak and
`value` = reads segmented by all ranks / max-over-ranks time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config3|config4|config5|config1]
  python bench.py --workload cluster-many|cluster-big|isoforms      (rows N3 / N4 of SURVEY 8f: their own metrics, 1 GPU)

Prints ONE JSON line (rank 0).  `roofline` is for the interval-scoring kernel: algorithmic bytes
4*(N+K)*R + 4*R per partition over its mean launch duration, measured with HIP events
on the library's stream inside the timed region.  `cpu_baseline` times the CPU oracle (a C port of
the reference algorithm, single thread) on the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from freddie_amd import _lib, pack, synth, tables  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

PARAMS = {
    "default": dict(sigma=5.0, threshold_rate=0.9, variance_factor=3.0, max_problem_size=50,
                    min_read_support_outside=3, ignore_ends=True),
    "config5": dict(sigma=3.0, threshold_rate=0.8, variance_factor=3.0, max_problem_size=50,
                    min_read_support_outside=3, ignore_ends=True),
}


def per_gpu_partitions(workload, n_gpus):
    """Number of partitions one rank processes: config2 = one 50k-read partition per GPU; the
    multi-partition configs give every GPU the 1/8 share of the 8-GPU whole-node job."""
    w = synth.WORKLOADS[workload]
    if workload in ("config4", "config5"):
        return w["n_partitions"] // 8
    return w["n_partitions"]


def build_batch(workload, rank, n_local):
    w = dict(synth.WORKLOADS[workload])
    w.pop("n_partitions")
    parts = []
    n_reads = 0
    for i in range(n_local):
        g = synth.generate(rank * n_local + i, with_seq=False, **w)
        n_reads += g.n_reads
        parts.append(pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
    return parts, n_reads


def cpu_baseline(parts, n_reads_of, params, tabs, min_s=10.0, max_s=25.0):
    """Time the CPU oracle on a bounded sample of the same workload: whole partitions in order, repeated
    until at least min_s seconds of CPU work have been measured (never more than max_s)."""
    from oracle import oracle
    t0 = time.perf_counter()
    reads = 0
    used = 0
    passes = 0
    while True:
        for p, nr in zip(parts, n_reads_of):
            o = oracle.segment(p.iv_start, p.iv_end, p.rep_weight, p.rep_exon_off, p.ex_ts, p.ex_te, **params, **tabs)
            if o["error"]:
                raise RuntimeError("oracle failed: " + o["errmsg"])
            reads += nr
            used += 1
            if time.perf_counter() - t0 > max_s:
                break
        passes += 1
        if time.perf_counter() - t0 > min_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=reads / dt, unit="reads/s", cores=1, kind="port",
                sample="%d partition runs (%d reads, %d pass(es) over %d partitions) of the C oracle, %.1f s" % (
                    used, reads, passes, len(parts), dt))


def _oracle_one(job):
    from oracle import oracle
    p, params, tabs = job
    o = oracle.segment(p.iv_start, p.iv_end, p.rep_weight, p.rep_exon_off, p.ex_ts, p.ex_te, **params, **tabs)
    return o["error"]


def cpu_baseline_all_cores(parts, n_reads_of, params, tabs, min_s=8.0):
    """The same oracle over whole partitions on every host core (one process per core, partitions are independent):
    only meaningful for the many-partition workloads.  The pool is started (and warmed) outside the timed region."""
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))                       # the CPUs this process may use, not the machine's
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    jobs = [(p, params, tabs) for p in parts]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_oracle_one, jobs[:cores])                       # start-up, page-in
        t0 = time.perf_counter()
        reads = passes = 0
        while time.perf_counter() - t0 < min_s:
            errs = pool.map(_oracle_one, jobs, chunksize=max(1, len(jobs) // (cores * 8)))
            if any(errs):
                raise RuntimeError("oracle failed in the all-cores baseline")
            reads += sum(n_reads_of)
            passes += 1
        dt = time.perf_counter() - t0
    return dict(value=reads / dt, unit="reads/s", cores=cores, kind="port",
                sample="%d pass(es) over %d partitions (%d reads) of the C oracle on %d processes, %.1f s" % (
                    passes, len(parts), reads, cores, dt))


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


def measured_traffic(workload):
    """HBM bytes per launch of the scoring kernel from the committed PMC passes (profiles/traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)[workload]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=sorted(synth.WORKLOADS) + list(NEXT_ROW_WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the CPU oracle on every host core (many-partition workloads; adds cpu_baseline_all_cores)")
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
    n_local = per_gpu_partitions(args.workload, args.gpus)
    parts, n_reads = build_batch(args.workload, rank, n_local)
    n_reads_of = [p.n_reads for p in parts]

    cpu_all = None
    if args.cpu_all_cores and rank == 0 and len(parts) > 1:
        cpu_all = cpu_baseline_all_cores(parts, n_reads_of, params, tabs)    # forks: before anything touches the GPU

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    ctx = _lib.Context(local_rank)
    ctx.set_params(**params, **tabs)
    ctx.upload(**pack.concat_batch(parts))       # inputs resident in HBM from here on
    ctx.set_profiling(True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.run()
        ctx.sync()
    barrier()
    t0 = time.perf_counter()
    stage_acc = {}
    for _ in range(args.steps):
        ctx.run()
        ctx.sync()
        for k, v in ctx.stage_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    barrier()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt, float(n_reads)], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, total_reads = float(tmax[0]), float(tsum[1])
    else:
        dt_max, total_reads = dt, float(n_reads)

    if rank == 0:
        sizes = ctx.sizes()
        alg_bytes = ctx.scoring_algorithmic_bytes()
        score_ms = stage_acc["interval_scoring"] / args.steps
        achieved = alg_bytes / (score_ms * 1e-3) / 1e9 if score_ms > 0 else 0.0
        out = {
            "metric": "reads segmented/sec (whole node)",
            "value": total_reads * args.steps / dt_max,
            "unit": "reads/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "dtype_detail": "u32 bit-planes + popcount for scoring, int64 DP, f64 Gaussian smoothing / threshold",
            "data": "synthetic",
            "config": {"workload": args.workload, "partitions_per_gpu": n_local, "reads_per_gpu": n_reads,
                       "candidates_rank0": sizes["n_cand"], "dp_problems_rank0": sizes["n_problems"],
                       "positions_rank0": sizes["n_positions"], "params": {k: params[k] for k in ("sigma", "threshold_rate")}},
            "roofline": {"kernel": "k_score (interval scoring)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload),
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": score_ms},
            "stage_ms": {k: v / args.steps for k, v in stage_acc.items()},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(parts, n_reads_of, params, tabs)
        else:
            out["cpu_baseline"] = None
        if cpu_all is not None:
            out["cpu_baseline_all_cores"] = cpu_all
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
