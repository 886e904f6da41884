"""The multi-rank flow of bench.py on the one GPU there is (``-m gpu``): two ranks under torch.distributed.run with the gloo
backend share the card (the driver's runs use RCCL, one GPU per rank -- no 8-GPU node has been available to any round, so
this rehearsal is what covers rank scatter, the barrier / max / sum reductions and the one JSON line of rank 0).  The job is
fixed (strong scaling), so the line must say so whatever N is, and the job's result checksum must not depend on N.
Reference: the partition-parallel map this replaces, py/freddie_segment.py:871-876."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-extras", "--contexts", "2"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_report_the_same_job():
    # (batches of 1 M reads: by default a batch is the whole job where the card has the memory for eight contexts that hold it)
    env = dict(os.environ, FREDDIE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", FREDDIE_BENCH_BATCH_READS="1000000")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON,
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    r2 = _last_json(two.stdout)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + COMMON,
                         cwd=ROOT, env=dict(os.environ, FREDDIE_BENCH_BATCH_READS="1000000"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    r1 = _last_json(one.stdout)
    assert r2["n_gpus"] == 2 and r1["n_gpus"] == 1
    assert r2["rank_sync"] == "gloo" and r1["rank_sync"] is None
    for r in (r1, r2):
        assert r["metric"] == "reads segmented/sec (whole node)" and r["unit"] == "reads/s"
        assert r["scaling"] == "strong" and r["config"]["workload"] == "config4" and r["config"]["reads"] == 2000000
        assert r["steps"] == 2 and r["value"] > 0
    assert r2["config"]["batches_per_step_rank0"] == 1 and r1["config"]["batches_per_step_rank0"] == 2     # (batches of 1 M reads; two contexts)
    assert r2["result_checksum"] == r1["result_checksum"] and r1["result_checksum"] > 0
    # (both timed regions pass over the same job K times: resident inputs -- `value` -- and host memory -> host memory)
    for r in (r1, r2):
        assert r["result_checksum_h2h"] == r1["result_checksum"] and r["value_h2h"] > 0
        assert r["sync_timeouts"] == 0
        assert r["result_label_popcount_resident"] > 0        # (per resident context: its batch's popcount, checked inside bench.py)
        assert 0 < r["roofline_path"]["frac_h2h"] <= r["roofline_path"]["frac"] < 1
    # (the checksum holds the SUM of the final positions, and the labels' content is compared too: '1' / '2' labels of the job)
    assert r2["result_label_popcount"] == r1["result_label_popcount"] and r1["result_label_popcount"] > 0


def _ranks(n, workload, env_extra):
    """bench.py under torch.distributed.run with n gloo ranks on the one card, and the one-rank run of the same job."""
    env = dict(os.environ, FREDDIE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", FREDDIE_BENCH_PASSES="2", **env_extra)
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-extras", "--workload", workload, "--contexts", "8"]
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
                           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + common,
                          cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert many.returncode == 0, many.stderr[-3000:]
    env1 = {k: v for k, v in env.items() if k != "FREDDIE_BENCH_BACKEND"}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                         cwd=ROOT, env=env1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    return _last_json(many.stdout), _last_json(one.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,batch_reads,batches_rank0", [("config4", "500000", 1), ("config5", "250000", 5)],
                         ids=["one-batch-per-rank-eight-contexts-share-it", "five-batches-on-eight-contexts"])
def test_the_shape_an_eight_gpu_run_has_per_rank(workload, batch_reads, batches_rank0):
    """What the driver's N = 8 run will be the first to execute on hardware, rehearsed within this pool's limit of six processes on
    a card: FOUR gloo ranks on the one GPU, eight contexts each.  (a) config4 with batches of 500 k reads: every rank's share is
    ONE batch and its eight contexts all hold it -- with the default batches of 1 M reads every rank of an N >= 2 run is in exactly this
    position (bench.py: `n_b < n_ctx`, context k holds batch k mod n_b, a pass's runs dealt out over a batch's holders);
    (b) config5: five batches on eight contexts, three of them held twice.  The job is fixed: checksum of final positions and the
    labels' popcount must be the one-rank run's, no waiter may time out, the line says strong scaling over 4 GPUs."""
    rn, r1 = _ranks(4, workload, {"FREDDIE_BENCH_BATCH_READS": batch_reads})
    assert rn["n_gpus"] == 4 and r1["n_gpus"] == 1
    for r in (rn, r1):
        assert r["scaling"] == "strong" and r["config"]["workload"] == workload and r["config"]["contexts_per_gpu"] == 8
        assert r["value"] > 0 and r["value_h2h"] > 0 and r["sync_timeouts"] == 0 and r["passes_per_step"] == 2
    assert rn["config"]["batches_per_step_rank0"] == batches_rank0
    assert rn["result_checksum"] == r1["result_checksum"] == rn["result_checksum_h2h"] == r1["result_checksum_h2h"] > 0
    assert rn["result_label_popcount"] == r1["result_label_popcount"] > 0
    # (the resident region's label popcounts are compared with the warm-up's per batch INSIDE bench.py; here: every context fetched one)
    assert rn["result_label_popcount_resident"] > 0 and r1["result_label_popcount_resident"] > 0


@pytest.mark.gpu
def test_the_rccl_branch_with_the_one_rank_a_one_gpu_box_allows():
    """The driver's N > 1 runs carry their barriers and the reduction of (time, reads, checksums) over RCCL, one GPU per rank; two ranks
    cannot share a card over RCCL, so what can be executed here is that branch with ONE rank (FREDDIE_BENCH_FORCE_DIST=1 under
    torch.distributed.run): the gloo default group, the RCCL group made beside it, its probe all-reduce, `barrier(group=...)`, the MAX /
    SUM reductions on device tensors.  The line must say the ranks were synchronised over RCCL and report the job as the plain run does."""
    env = dict(os.environ, FREDDIE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", FREDDIE_BENCH_PASSES="2")
    env.pop("FREDDIE_BENCH_BACKEND", None)
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + COMMON,
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    r = _last_json(run.stdout)
    assert r["rank_sync"] == "rccl", (r["rank_sync"], run.stderr[-2000:])
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["value_h2h"] > 0 and r["sync_timeouts"] == 0
    assert r["result_checksum"] == r["result_checksum_h2h"] > 0
