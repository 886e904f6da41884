"""The drop-in CLI's multi-GPU path on CPU: main() with two worker processes (--devices 0,1), the LPT scatter, the
pipelined driver (two contexts per worker, loader / device / writer threads) and the progress protocol -- everything
but the GPU, whose context is replaced by a stand-in that answers with the CPU oracle's results.  The outputs must be
byte-identical to a one-worker run and cover every partition exactly once (reference: main() :847-885 maps the
partitions over a process pool)."""
import os

import numpy as np
import pytest

import util
from freddie_amd import _lib, devices, pack, segment, synth


class OracleContext:
    """Stand-in for _lib.Context: same calls the driver makes, results from the CPU oracle."""

    def __init__(self, device):
        self.device = device
        self.n_part = 0

    def set_params(self, sigma, threshold_rate, variance_factor, max_problem_size, min_read_support_outside, ignore_ends,
                   w_main, w_refine, h_table):
        self.params = dict(sigma=sigma, threshold_rate=threshold_rate, variance_factor=variance_factor,
                           max_problem_size=max_problem_size, min_read_support_outside=min_read_support_outside,
                           ignore_ends=ignore_ends)

    def upload(self, **a):
        self.a = {k: np.array(v) for k, v in a.items()}
        self.n_part = len(a["part_iv_off"]) - 1

    def run(self):
        a = self.a
        pfo, lo, fps, labs = [0], [0], [], []
        for p in range(self.n_part):
            k0, k1 = a["part_iv_off"][p], a["part_iv_off"][p + 1]
            r0, r1 = a["part_rep_off"][p], a["part_rep_off"][p + 1]
            e0, e1 = a["rep_exon_off"][r0], a["rep_exon_off"][r1]
            part = pack.PackedPartition(a["iv_start"][k0:k1], a["iv_end"][k0:k1], a["rep_weight"][r0:r1],
                                        a["rep_exon_off"][r0:r1 + 1] - e0, a["ex_ts"][e0:e1], a["ex_te"][e0:e1], np.zeros(0, np.int32))
            o = util.run_oracle(part, self.params)
            assert not o["error"], o["errmsg"]
            fps.append(o["final_pos"]); labs.append((o["labels"] + 48).astype(np.uint8).ravel())
            pfo.append(pfo[-1] + len(o["final_pos"])); lo.append(lo[-1] + labs[-1].size)
        self.res = (np.array(pfo, np.int64), np.concatenate(fps).astype(np.int32), np.array(lo, np.int64), np.concatenate(labs))

    def results(self, packed=False):
        return self.res[:3] + (util.pack_labels(self.res[3]),) if packed else self.res

    def close(self):
        pass


def fake_open_contexts(device, n=2):
    return [OracleContext(device) for _ in range(n)]


def read_tree(root):
    out = {}
    for dp, _, fs in os.walk(root):
        for f in fs:
            out[os.path.relpath(os.path.join(dp, f), root)] = open(os.path.join(dp, f), "rb").read()
    return out


def test_main_two_workers_equals_one_worker(tmp_path, monkeypatch, capsys):
    split = str(tmp_path / "split")
    n = 14
    for i in range(n):
        synth.generate(300 + i, n_reads=40 + 25 * (i % 5), n_exons=25, rp=0.1, write_dir=split, contig="chr%d" % (i % 3))
    monkeypatch.setattr(segment, "open_contexts", fake_open_contexts)
    monkeypatch.setattr(segment, "WORKER_START_METHOD", "fork")
    outs = []
    for devs in ("0", "0,1"):
        out = str(tmp_path / ("out_" + devs.replace(",", "_")))
        segment.main(["-s", split, "-o", out, "-t", "2", "--devices", devs, "--batch-reads", "150", "--sidecar", "off"])
        outs.append(read_tree(out))
        printed = capsys.readouterr().out
        assert "Done with 0/%d tints" % n in printed            # the reference's progress line (:877-878)
    assert outs[0] == outs[1]
    tsvs = [k for k in outs[0] if k.endswith(".tsv")]
    assert len(tsvs) == n and len([k for k in outs[0] if k.endswith(".log")]) == n
    assert all(len(v) > 0 for k, v in outs[0].items() if k.endswith(".tsv"))


class SmallContext(OracleContext):
    """A context that takes at most three partitions per upload, like a device whose limits (2^31 positions per upload,
    memory) a batch of many low-coverage partitions exceeds."""
    refused = 0

    def upload(self, **a):
        if len(a["part_iv_off"]) - 1 > 3:
            SmallContext.refused += 1
            raise _lib.SegError("fseg_upload failed (4): batch has 9999999999 positions; split it (limit 2^31-1 per upload)", 4)
        super().upload(**a)


def test_a_batch_the_device_refuses_is_halved_and_retried(tmp_path, monkeypatch):
    """Batches are cut by split-file bytes; the driver must not die when one of them is more than an upload takes (the
    reference works partition by partition): same output files as with batches that fit."""
    split = str(tmp_path / "split")
    for i in range(11):
        synth.generate(700 + i, n_reads=30 + 10 * (i % 4), n_exons=20, rp=0.1, write_dir=split, contig="chr%d" % (i % 2))
    outs = []
    for cls, batch_reads in ((OracleContext, "60"), (SmallContext, "100000")):
        monkeypatch.setattr(segment, "open_contexts", lambda device, n=2, cls=cls: [cls(device) for _ in range(n)])
        out = str(tmp_path / ("out_" + cls.__name__))
        segment.main(["-s", split, "-o", out, "-t", "2", "--devices", "0", "--batch-reads", batch_reads, "--sidecar", "off"])
        outs.append(read_tree(out))
    assert SmallContext.refused >= 2                      # 11 partitions in one batch: halved more than once
    assert outs[0] == outs[1] and len([k for k in outs[0] if k.endswith(".tsv")]) == 11
    with pytest.raises(_lib.SegError):                    # an error splitting cannot cure is still an error
        class Broken(OracleContext):
            def run(self):
                raise _lib.SegError("fseg_run failed (4): a problem has 200 candidates", 4)
        monkeypatch.setattr(segment, "open_contexts", lambda device, n=2: [Broken(device) for _ in range(n)])
        segment.main(["-s", split, "-o", str(tmp_path / "out_broken"), "-t", "2", "--devices", "0", "--sidecar", "off"])


def test_device_count_without_the_runtime(tmp_path):
    """devices.visible_gpu_count(): the runtime's own variables first, then the KFD topology (GPU nodes = simd_count > 0)."""
    root = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):
        d = root / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n" % (0 if simd else 64, simd, 128 + i))
    assert devices.visible_gpu_count(env={}, kfd_root=str(root)) == 3
    assert devices.visible_gpu_count(env={"HIP_VISIBLE_DEVICES": "0,2"}, kfd_root=str(root)) == 2
    assert devices.visible_gpu_count(env={"ROCR_VISIBLE_DEVICES": "0"}, kfd_root=str(root)) == 1
    assert devices.visible_gpu_count(env={"HIP_VISIBLE_DEVICES": ""}, kfd_root=str(root)) == 0
    assert devices.visible_gpu_count(env={"HIP_VISIBLE_DEVICES": "0,-1,1"}, kfd_root=str(root)) == 1
    cpus = devices.cpus_near_gpu(1, 2, kfd_root=str(root))
    assert cpus and set(cpus) <= set(os.sched_getaffinity(0))
    both = [devices.cpus_near_gpu(k, 2, kfd_root=str(root)) for k in (0, 1)]
    assert not (set(both[0]) & set(both[1])) or len(os.sched_getaffinity(0)) < 2


def test_eight_workers_get_disjoint_cores(tmp_path, monkeypatch):
    """devices.cpus_near_gpu() for the 8-GPU node no round has had: eight workers under ``--devices 0,...,7`` (ordinal ==
    worker, the KFD topology says where each GPU sits) and under a ``HIP_VISIBLE_DEVICES`` remap (nothing is known: even
    slices by worker index) never share a core, and every worker gets some."""
    root = tmp_path / "nodes"
    for i, simd in enumerate([0, 0] + [1024] * 8):
        d = root / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n" % (0 if simd else 64, simd, 128 + i))
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)))
    for env in ({}, {"HIP_VISIBLE_DEVICES": "7,6,5,4,3,2,1,0"}):
        cores = [devices.cpus_near_gpu(k, 8, ordinal=k, kfd_root=str(root), env=env) for k in range(8)]
        assert all(cores) and all(set(c) <= set(range(128)) for c in cores)
        for a in range(8):
            for b in range(a + 1, 8):
                assert not (set(cores[a]) & set(cores[b])), (env, a, b)
        assert sum(len(c) for c in cores) <= 128
    # a worker whose GPU is not "its" ordinal (--devices 2,4): slices by worker index, disjoint as well
    pair = [devices.cpus_near_gpu(k, 2, ordinal=o, kfd_root=str(root), env={}) for k, o in ((0, 2), (1, 4))]
    assert pair[0] and pair[1] and not (set(pair[0]) & set(pair[1]))


def test_host_ceiling_tool_runs_the_real_main_without_a_device(tmp_path):
    """tools/host_ceiling.py: main() with a stand-in context that answers at once (both ends of every interval as final
    positions, a constant label): every partition gets its segment TSV, with one label per segment and read."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("host_ceiling", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "host_ceiling.py"))
    hc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hc)
    split, out = str(tmp_path / "split"), str(tmp_path / "out")
    for i in range(5):
        synth.generate(700 + i, n_reads=30 + 10 * i, n_exons=12, rp=0.1, write_dir=split)
    saved = (segment.open_contexts, segment.WORKER_START_METHOD)
    try:
        for label, workers in ((0, 1), (1, 2)):
            assert hc.run(split, out, workers, 2, label) > 0
            files = sorted(f for _, _, fs in os.walk(out) for f in fs if f.endswith(".tsv"))
            assert len(files) == 5
            for dp, _, fs in os.walk(out):
                for f in fs:
                    if not f.endswith(".tsv"):
                        continue
                    lines = open(os.path.join(dp, f)).read().split("\n")
                    n_seg = len(lines[0].split("\t")[2].split(",")) - 1
                    for ln in lines[1:]:
                        if ln:
                            assert ln.split("\t")[5] == str(label) * n_seg
    finally:
        segment.open_contexts, segment.WORKER_START_METHOD = saved
