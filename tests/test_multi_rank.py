"""The N>1 path on CPU: two gloo ranks take their static share of the partitions (no data-path collective)
and the shares cover the work exactly once."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from freddie_amd import scatter, synth
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
costs = [1000 + 37 * (i %% 11) for i in range(40)]            # every rank derives the same static plan
mine = scatter.rank_share(costs, rank, world)
# each rank touches only its own partitions: generate them and count the reads
reads = sum(synth.generate(i, n_reads=20 + i, n_exons=10, with_seq=False).n_reads for i in mine[:5])
t = torch.tensor([len(mine), sum(costs[i] for i in mine), reads], dtype=torch.int64)
gathered = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(gathered, t)                                   # bookkeeping only (what bench.py reduces)
flags = torch.zeros(len(costs), dtype=torch.int64)
flags[mine] = 1
dist.all_reduce(flags)
if rank == 0:
    assert int(sum(g[0] for g in gathered)) == len(costs)
    assert bool((flags == 1).all()), "partitions must be owned exactly once"
    loads = [int(g[1]) for g in gathered]
    assert max(loads) - min(loads) <= max(costs)
    print("OK", loads)
dist.destroy_process_group()
'''


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert "OK" in res.stdout
