"""In-tree build of the native pieces.

* ``libfreddie_seg.so``  -- the product: gfx950 HIP kernels + C-ABI (hipcc, cross-compiles without a GPU; a translation unit per
  stage family under csrc/, compiled side by side: build_seg())
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
CSRC = os.path.join(_HERE, "csrc")
# libfreddie_seg.so: one translation unit per stage family (kernels), the host side (freddie_seg.hip: contexts, launches, C-ABI)
# and the radix-sort unit; what they share is in two headers of csrc/ plus the ABI header.  The units are compiled side by side
# (k_solve's three size classes are a unit each: the kernel is most of the build time -- 71 s as one file, ~25 s now on 8 cores).
SEG_UNITS = ["freddie_seg", "seg_front", "seg_problems", "seg_score_arena", "seg_score_fused", "seg_solve16", "seg_solve32", "seg_solve60",
             "seg_tail", "seg_upload", "freddie_seg_sort"]
SEG_HEADERS = [os.path.join(CSRC, "seg_common.h"), os.path.join(CSRC, "seg_kernels.h"), os.path.join(CSRC, "seg_solve.h")]
SEG_SRC = [os.path.join(CSRC, u + ".hip") for u in SEG_UNITS] + SEG_HEADERS
SEG_OBJ_DIR = os.path.join(CSRC, ".obj")
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


SEG_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC"]


def seg_command(extra=()):
    """What the library is built from, as ONE command line (it is what the source hash covers, and it builds the library as it
    stands -- one unit after the other; build_seg() runs the same compilations side by side)."""
    return (["hipcc"] + SEG_FLAGS + list(extra) + ["-shared", "-I", INCLUDE, "-o", SEG_SO] +
            [os.path.join(CSRC, u + ".hip") for u in SEG_UNITS] + ["-lhsa-runtime64"])


def seg_hash(extra=()):
    """The hash a current libfreddie_seg.so carries (fseg_source_hash()): sources, headers and the build command."""
    return source_hash(SEG_SRC + [os.path.join(INCLUDE, "freddie_seg.h")], seg_command(extra))


def build_seg(force=False, verbose=False, extra=(), target=None):
    """Compile the units that changed (an object per unit under csrc/.obj/, each with the hash of what it was compiled from),
    side by side, and link.  ``extra``: more compiler flags (diagnostic builds: tools/build_variant.sh); ``target``: the library
    to write (default: the product, libfreddie_seg.so)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    target = target or SEG_SO
    want = seg_hash(extra)
    if not force and embedded_hash(target) == want:
        return target
    t0 = time.perf_counter()
    tag = hashlib.sha256((" ".join(extra) + "|" + os.path.basename(target)).encode()).hexdigest()[:8] if (extra or target != SEG_SO) else "main"
    objdir = os.path.join(SEG_OBJ_DIR, tag)
    os.makedirs(objdir, exist_ok=True)
    shared = SEG_HEADERS + [os.path.join(INCLUDE, "freddie_seg.h")]

    def compile_unit(u):
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(objdir, u + ".o")
        cmd = ["hipcc"] + SEG_FLAGS + list(extra) + ["-I", INCLUDE, "-c", src, "-o", obj]
        if u == "freddie_seg":
            cmd.append('-DFREDDIE_SOURCE_HASH="%s"' % want)         # (the stamp lives in the host unit: it is recompiled whenever anything changed)
        h = source_hash([src] + shared, cmd)
        try:
            with open(obj + ".hash") as f:
                if not force and f.read().strip() == h and os.path.exists(obj):
                    return u, 0.0
        except OSError:
            pass
        t = time.perf_counter()
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(obj + ".hash", "w") as f:
            f.write(h)
        return u, time.perf_counter() - t

    with ThreadPoolExecutor(max_workers=max(1, min(len(SEG_UNITS), os.cpu_count() or 1))) as pool:
        times = list(pool.map(compile_unit, SEG_UNITS))
    link = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + [os.path.join(objdir, u + ".o") for u in SEG_UNITS] + ["-lhsa-runtime64"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    if embedded_hash(target) != want:
        raise RuntimeError("%s was built but does not carry its source hash" % target)
    if verbose:
        print("libfreddie_seg: %.1f s (units: %s)" % (time.perf_counter() - t0, ", ".join("%s %.1f" % (u, t) for u, t in times if t > 0)))
    return target


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


if __name__ == "__main__":
    # python -m freddie_amd.build [--variant NAME] [-DFOO=1 ...]: the product, or the same library under other switches as
    # freddie_amd/libfreddie_seg_<NAME>.so (FSEG_LIB=... selects it)
    import sys
    argv = sys.argv[1:]
    name = None
    if argv[:1] == ["--variant"]:
        name, argv = argv[1], argv[2:]
    if name:
        print(build_seg(force=False, verbose=True, extra=tuple(argv), target=os.path.join(_HERE, "libfreddie_seg_%s.so" % name)))
    else:
        build_all(force=False, verbose=True)
