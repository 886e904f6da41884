"""In-tree build of the native pieces.

* ``libfreddie_seg.so``  -- the product: gfx950 HIP kernels + C-ABI (hipcc, cross-compiles without a GPU)
* ``libfreddie_host.so`` -- the product's native host I/O: TSV parser, gaps/poly-A, writer (g++)
* ``libfreddie_cluster.so`` -- gfx950 kernels + C-ABI of the clustering stage's pre-ILP graph work (hipcc; built by
  ``freddie_amd.cluster_prep.build``)
* ``libfreddie_isoforms.so`` -- gfx950 kernels + C-ABI of the isoform-consensus stage's per-read loops (hipcc; built by
  ``freddie_amd.isoforms.build``)
* ``synth/libfreddie_synth.so`` -- synthetic split-partition generator (gcc; test/bench infrastructure)

The oracle (``oracle/``) is test infrastructure and is built by ``oracle/Makefile``; it is never
linked into or loaded by anything in this package.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SEG_SO = os.path.join(_HERE, "libfreddie_seg.so")
SEG_SRC = [os.path.join(_HERE, "csrc", "freddie_seg.hip")]
HOST_SO = os.path.join(_HERE, "libfreddie_host.so")
HOST_SRC = [os.path.join(_HERE, "csrc", "freddie_host.cpp")]
INCLUDE = os.path.join(ROOT, "include")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_seg(force=False, verbose=False):
    deps = SEG_SRC + [os.path.join(INCLUDE, "freddie_seg.h")]
    if force or _stale(SEG_SO, deps):
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-shared", "-fPIC", "-I", INCLUDE,
               "-o", SEG_SO] + SEG_SRC
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SEG_SO


def build_host(force=False, verbose=False):
    """Native host I/O (parser, gaps/poly-A, writer): plain C++17, no GPU code."""
    deps = HOST_SRC + [os.path.join(INCLUDE, "freddie_host.h")]
    if force or _stale(HOST_SO, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wall", "-I", INCLUDE, "-o", HOST_SO] + HOST_SRC
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_SO


def build_all(force=False, verbose=False):
    from . import cluster_prep, isoforms, synth
    build_seg(force, verbose)
    build_host(force, verbose)
    cluster_prep.build(force, verbose)
    isoforms.build(force, verbose)
    synth.build(force)
    return SEG_SO
