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
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SEG_SO = os.path.join(_HERE, "libfreddie_seg.so")
SEG_SRC = [os.path.join(_HERE, "csrc", "freddie_seg.hip"), os.path.join(_HERE, "csrc", "freddie_seg_sort.hip")]
HOST_SO = os.path.join(_HERE, "libfreddie_host.so")
HOST_SRC = [os.path.join(_HERE, "csrc", "freddie_host.cpp")]
INCLUDE = os.path.join(ROOT, "include")


# Every native library carries the hash of what it was built from ("FREDDIE_SRC_HASH=<hex>" somewhere in its bytes):
# the built .so files are git-ignored but travel to the GPU box with the tree, so "is this binary current?" must not
# depend on file times (a checkout or a copy resets them).
_STAMP = b"FREDDIE_SRC_HASH="


def source_hash(sources, cmd):
    """sha256 over the build command and the contents of every source / header, first 32 hex digits."""
    # the tree may live anywhere (the GPU box unpacks it under another path): paths inside the tree count relative to it
    h = hashlib.sha256(" ".join(cmd).replace(ROOT + os.sep, "").replace(ROOT, ".").encode())
    for path in sources:
        with open(path, "rb") as f:
            h.update(b"\0" + os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:32]


def embedded_hash(so_path):
    """The hash a library was built with, or None (missing file, or a build from before the stamp existed)."""
    try:
        with open(so_path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(_STAMP)
    return blob[i + len(_STAMP):i + len(_STAMP) + 32].decode("ascii", "replace") if i >= 0 else None


def build_stamped(target, cmd, sources, force=False, verbose=False):
    """Run ``cmd`` (+ the stamp definition) unless ``target`` already carries the hash of (cmd, sources)."""
    want = source_hash(sources, cmd)
    if not force and embedded_hash(target) == want:
        return False
    full = cmd + ['-DFREDDIE_SOURCE_HASH="%s"' % want]
    if verbose:
        print(" ".join(full))
    subprocess.check_call(full)
    if embedded_hash(target) != want:
        raise RuntimeError("%s was built but does not carry its source hash" % target)
    return True


def is_current(target, cmd, sources):
    return embedded_hash(target) == source_hash(sources, cmd)


def seg_command():
    return ["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-shared", "-fPIC", "-I", INCLUDE,
            "-o", SEG_SO] + SEG_SRC + ["-lhsa-runtime64"]


def build_seg(force=False, verbose=False):
    build_stamped(SEG_SO, seg_command(), SEG_SRC + [os.path.join(INCLUDE, "freddie_seg.h")], force, verbose)
    return SEG_SO


def host_command():
    return ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wall", "-I", INCLUDE, "-o", HOST_SO] + HOST_SRC


def build_host(force=False, verbose=False):
    """Native host I/O (parser, gaps/poly-A, writer): plain C++17, no GPU code."""
    build_stamped(HOST_SO, host_command(), HOST_SRC + [os.path.join(INCLUDE, "freddie_host.h")], force, verbose)
    return HOST_SO


def build_all(force=False, verbose=False):
    from . import cluster_prep, isoforms, synth
    build_seg(force, verbose)
    build_host(force, verbose)
    cluster_prep.build(force, verbose)
    isoforms.build(force, verbose)
    synth.build(force)
    return SEG_SO
