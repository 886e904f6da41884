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
    env = dict(os.environ, FREDDIE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON,
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    r2 = _last_json(two.stdout)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + COMMON,
                         cwd=ROOT, env=os.environ.copy(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    r1 = _last_json(one.stdout)
    assert r2["n_gpus"] == 2 and r1["n_gpus"] == 1
    for r in (r1, r2):
        assert r["metric"] == "reads segmented/sec (whole node)" and r["unit"] == "reads/s"
        assert r["scaling"] == "strong" and r["config"]["workload"] == "config4" and r["config"]["reads"] == 2000000
        assert r["steps"] == 2 and r["value"] > 0
    assert r2["config"]["batches_per_step_rank0"] == 4 and r1["config"]["batches_per_step_rank0"] == 8
    assert r2["result_checksum"] == r1["result_checksum"] and r1["result_checksum"] > 0
    # (both timed regions pass over the same job K times: resident inputs -- `value` -- and host memory -> host memory)
    for r in (r1, r2):
        assert r["value_host_to_host"]["result_checksum"] == r1["result_checksum"] and r["value_host_to_host"]["value"] > 0
    # (the checksum holds the SUM of the final positions, and the labels' content is compared too: '1' / '2' labels of the job)
    assert r2["result_label_popcount"] == r1["result_label_popcount"] and r1["result_label_popcount"] > 0
