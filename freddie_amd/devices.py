"""Which GPUs may this process use, and which host cores sit next to each of them -- found WITHOUT touching the HIP
runtime, so a parent that only counts devices can still start one worker process per GPU (a process that has
initialised HIP must not fork+exec children; the reference simply forks a ``multiprocessing.Pool``,
py/freddie_segment.py:871-876).

Sources, in order: the ``*_VISIBLE_DEVICES`` variables the ROCm runtime itself honours, then the KFD topology in
sysfs (``/sys/class/kfd/kfd/topology/nodes/*/properties``: a node with ``simd_count`` > 0 is a GPU; one whose render
node cannot be opened by this user does not count), then -- only if sysfs is absent -- a short-lived helper process
that asks the runtime.
"""
import os
import subprocess
import sys

_KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def _visible_from_env(env):
    """Number of devices selected by the runtime's own variables, or None when none of them is set."""
    for name in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        v = env.get(name)
        if v is None:
            continue
        items = [x for x in v.split(",") if x.strip() != ""]
        n = 0
        for x in items:                      # the runtime stops at the first invalid entry (e.g. -1)
            if x.strip().lstrip("+").isdigit() or x.strip().upper().startswith("GPU-"):
                n += 1
            else:
                break
        return n
    return None


def _kfd_gpu_nodes(root=_KFD_NODES):
    """[(node index, {property: int})] of the GPU nodes of the KFD topology, in node order."""
    out = []
    try:
        names = sorted(os.listdir(root), key=lambda s: int(s) if s.isdigit() else 1 << 30)
    except OSError:
        return None
    for name in names:
        props = {}
        try:
            with open(os.path.join(root, name, "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    try:
                        props[k] = int(v)
                    except ValueError:
                        pass
        except OSError:
            continue
        if props.get("simd_count", 0) > 0:
            out.append((int(name) if name.isdigit() else len(out), props))
    return out


def _usable(props, dev_root="/dev/dri"):
    minor = props.get("drm_render_minor", -1)
    if minor < 0:
        return True
    path = os.path.join(dev_root, "renderD%d" % minor)
    return (not os.path.exists(dev_root)) or os.access(path, os.R_OK | os.W_OK)


def _ask_runtime():
    """Last resort: a child process loads the library and counts; the parent's HIP state stays untouched."""
    code = ("import ctypes,sys\n"
            "try:\n"
            "    L=ctypes.CDLL('libamdhip64.so'); n=ctypes.c_int(0)\n"
            "    print(n.value if L.hipGetDeviceCount(ctypes.byref(n))==0 else 0)\n"
            "except OSError:\n"
            "    print(0)\n")
    try:
        return int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60).stdout.strip() or 0)
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return 0


def visible_gpu_count(env=None, kfd_root=_KFD_NODES):
    """GPUs a HIP context could be created on, without initialising HIP in this process."""
    env = os.environ if env is None else env
    n_env = _visible_from_env(env)
    nodes = _kfd_gpu_nodes(kfd_root)
    if nodes is None:
        return n_env if n_env is not None else _ask_runtime()
    n_sys = sum(1 for _, p in nodes if _usable(p))
    return n_sys if n_env is None else min(n_env, n_sys) if n_sys else n_env


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def cpus_near_gpu(worker, n_workers, ordinal=None, kfd_root=_KFD_NODES, env=None):
    """Host cores for worker ``worker`` (of ``n_workers``, one per GPU).  ``ordinal`` is the HIP ordinal of its GPU: when
    it equals the worker index and no ``*_VISIBLE_DEVICES`` variable remaps the ordinals, ordinal k is the k-th usable GPU of
    the KFD topology and the worker gets its share of the cores of that GPU's NUMA node; otherwise (``--devices 2,4``, a
    remapped environment) nothing is known about where the GPU sits and the worker gets an even slice of the allowed
    cores -- by WORKER index, so that no two workers share cores."""
    env = os.environ if env is None else env
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if n_workers <= 1 or not allowed:
        return allowed
    ordinal = worker if ordinal is None else ordinal
    if ordinal == worker and _visible_from_env(env) is None:
        nodes = _kfd_gpu_nodes(kfd_root) or []
        usable = [p for _, p in nodes if _usable(p)]
        numa_of = []
        for p in usable:
            node = -1
            minor = p.get("drm_render_minor", -1)
            try:
                with open("/sys/class/drm/renderD%d/device/numa_node" % minor) as f:
                    node = int(f.read().strip())
            except (OSError, ValueError):
                pass
            numa_of.append(node)
        if worker < len(numa_of) and numa_of[worker] >= 0:
            try:
                with open("/sys/devices/system/node/node%d/cpulist" % numa_of[worker]) as f:
                    local = [c for c in _parse_cpulist(f.read()) if c in set(allowed)]
            except OSError:
                local = []
            peers = [i for i, n in enumerate(numa_of[:n_workers]) if n == numa_of[worker]]
            if local and worker in peers:
                k = peers.index(worker)
                share = len(local) // len(peers)
                if share > 0:
                    return local[k * share:(k + 1) * share]
    share = max(1, len(allowed) // n_workers)
    return allowed[(worker % n_workers) * share:(worker % n_workers + 1) * share] or allowed


def pin_worker(worker, n_workers, ordinal=None):
    """Bind the calling process to its share of the host cores (best effort; returns the core list or None)."""
    cpus = cpus_near_gpu(worker, n_workers, ordinal)
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
            return cpus
        except OSError:
            return None
    return None
