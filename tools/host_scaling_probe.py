#!/usr/bin/env python3
"""Where does the host side stop scaling?  (round 5; no GPU needed)

P worker processes x T threads parse (and optionally write) disjoint shares of one split directory with the native host
library alone -- no Python driver, no queues, no stand-in device --, and the probe reports wall time, reads/s and the CPU
seconds the workers spent in user and in system mode.  If P x T scales here and tools/host_ceiling.py does not, the driver
is at fault; if system time explodes with P, the kernel is (page faults on the mapped files, tmpfs, directory locks).

    python tools/host_scaling_probe.py <work_dir_with_split/> [--procs 1,2,4,8,16,32] [--threads 1,4,16] [--write] [--batch 250]
"""
import argparse
import multiprocessing as mp
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def worker(args):
    jobs, threads, write, out_dir, batch, pin = args
    from freddie_amd import _host
    import host_ceiling as hc
    if pin is not None:
        try:
            os.sched_setaffinity(0, pin)
        except OSError:
            pass
    n = 0
    t_load = t_write = 0.0
    for i in range(0, len(jobs), batch):
        chunk = jobs[i:i + batch]
        t0 = time.perf_counter()
        hb = _host.HostBatch([j[0] for j in chunk], [j[1] for j in chunk], n_threads=threads)
        t1 = time.perf_counter()
        n += hb.n_reads
        if write:
            ctx = hc.NullContext(0)
            ctx.upload(**hb.arrays())
            hb.write(*ctx.results(packed=True), [os.path.join(out_dir, "seg_%s" % os.path.basename(j[0])) for j in chunk], n_threads=threads, packed=True)
        t_write += time.perf_counter() - t1
        t_load += t1 - t0
        hb.close()
    ru = resource.getrusage(resource.RUSAGE_SELF)
    return n, ru.ru_utime, ru.ru_stime, ru.ru_minflt, t_load, t_write


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("work")
    ap.add_argument("--procs", default="1,2,4,8,16,32")
    ap.add_argument("--threads", default="1,4,16")
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--batch", type=int, default=250)
    ap.add_argument("--pin", action="store_true", help="give every process its own slice of the allowed cores")
    args = ap.parse_args()
    split = os.path.join(args.work, "split")
    jobs = []
    for contig in sorted(os.listdir(split)):
        d = os.path.join(split, contig)
        for f in sorted(os.listdir(d)):
            if f.startswith("split_") and f.endswith(".tsv"):
                jobs.append((os.path.join(d, f), os.path.join(d, "reads_" + f[6:])))
    out_dir = os.path.join(args.work, "out_probe")
    os.makedirs(out_dir, exist_ok=True)
    cores = sorted(os.sched_getaffinity(0))
    print("%d partitions, %d cores allowed%s%s" % (len(jobs), len(cores), ", with writes" if args.write else "", ", pinned" if args.pin else ""))
    for T in [int(x) for x in args.threads.split(",")]:
        for P in [int(x) for x in args.procs.split(",")]:
            if P * T > 2 * len(cores):
                continue
            share = max(1, len(cores) // P)
            tasks = [(jobs[w::P], T, args.write, out_dir, args.batch, cores[w * share:(w + 1) * share] if args.pin else None) for w in range(P)]
            t0 = time.perf_counter()
            with mp.get_context("fork").Pool(P) as pool:
                res = pool.map(worker, tasks, chunksize=1)
            wall = time.perf_counter() - t0
            n = sum(r[0] for r in res); ut = sum(r[1] for r in res); st = sum(r[2] for r in res); mf = sum(r[3] for r in res)
            print("P=%3d x T=%2d: %.3f s  %6.2f M reads/s  cpu user %.2f s sys %.2f s  (%.2f us/read user, %.2f sys)  minor faults %d  in-lib load %.2f s write %.2f s (sum over processes)" % (
                P, T, wall, n / wall / 1e6, ut, st, ut / n * 1e6, st / n * 1e6, mf, sum(r[4] for r in res), sum(r[5] for r in res)), flush=True)


if __name__ == "__main__":
    main()
