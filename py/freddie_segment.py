#!/usr/bin/env python3
"""Drop-in for the reference's py/freddie_segment.py: same command line, same files in and out.
The work is done by freddie_amd (gfx950 HIP library behind a C-ABI); see INTEGRATION.md."""
import os
import sys
import threading
import time

_T0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _preload():
    # Mapping the native libraries (and the HIP runtime they link) takes as long as importing numpy does: both at once.
    # Loading is not initialising: no HIP call is made here, so a parent that only scatters work over worker processes
    # stays clear of the runtime (freddie_amd/devices.py).
    import ctypes
    # (the GPU library only when the command line says this process will drive the one GPU itself: --gpus 1 / --devices K)
    a = sys.argv
    single = any((x == "--gpus" and a[i + 1:i + 2] == ["1"]) or x == "--gpus=1" or (x == "--devices" and "," not in "".join(a[i + 1:i + 2]))
                 for i, x in enumerate(a))
    for name in ("libfreddie_host.so", "libfreddie_seg.so") if single else ("libfreddie_host.so",):
        try:
            ctypes.CDLL(os.environ.get("FSEG_LIB") if name == "libfreddie_seg.so" and os.environ.get("FSEG_LIB")
                        else os.path.join(ROOT, "freddie_amd", name))
        except OSError:
            pass            # the loaders in freddie_amd report a missing / stale library with a proper message


def _one_device():
    """The device this process will drive itself, or None: --gpus 1 / --devices K on the command line (anything else is
    decided by main(), which may have to start worker processes and must then not have touched the HIP runtime)."""
    a = sys.argv
    for i, x in enumerate(a):
        if x == "--devices" and i + 1 < len(a) and a[i + 1].isdigit():
            return int(a[i + 1])
        if x.startswith("--devices=") and x[10:].isdigit():
            return int(x[10:])
    if any((x == "--gpus" and a[i + 1:i + 2] == ["1"]) or x == "--gpus=1" for i, x in enumerate(a)) and not any(
            x == "--devices" or x.startswith("--devices=") for x in a):
        return 0
    return None


if __name__ == "__main__":
    threading.Thread(target=_preload, daemon=True).start()
    _dev = _one_device()
    if _dev is not None:
        # the GPU's contexts come up (0.3 s) beside the imports, the directory scan and the first parse
        from freddie_amd import _early
        _early.start(_dev)

if __name__ == "__main__":
    try:
        # (inside the try: an import error must not end the interpreter while the start-up thread is inside the HIP runtime)
        from freddie_amd.segment import main
        from freddie_amd._early import fast_exit_allowed
        _t_import = time.perf_counter()
        main(leave_contexts=fast_exit_allowed())
    finally:
        if _dev is not None:
            _early.finish()
    if os.environ.get("FREDDIE_TIMING") == "1":
        print("[freddie_segment] script: imports done %.3f s after its first line, main() returned at %.3f s (interpreter start-up and exit "
              "come on top)" % (_t_import - _T0, time.perf_counter() - _T0), file=sys.stderr)
    # The work is done and every output file is closed: leave without the interpreter's and the HIP runtime's tear-down (module
    # clean-up, the runtime's static destructors, unmapping the code objects: 0.15 s of a 2 M-read job's 0.84 s wall).  Only on
    # success -- an exception takes the ordinary way out above --, and never under a profiler, tracer or coverage run, whose output is
    # written by exit hooks (freddie_amd/_early.py: fast_exit_allowed); FREDDIE_CLEAN_EXIT=1 keeps the ordinary exit.
    if fast_exit_allowed():
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
else:
    from freddie_amd.segment import main  # noqa: E402,F401
