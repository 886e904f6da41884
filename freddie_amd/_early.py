"""Start-up of the one GPU a process drives itself, begun before anything else is imported.

Bringing a context up (HIP start-up, code object, streams, pinned buffers) takes about 0.3 s, importing numpy and this
package 0.1-0.2 s, and a 2 M-read job's whole wall time is under a second: ``py/freddie_segment.py`` calls ``start()`` as
its first statement when its command line names one GPU, and ``segment.open_contexts()`` takes the contexts over.  This
module imports nothing but ctypes / os / threading, and nothing here runs in a process that scatters work over worker
processes (such a parent must stay clear of the HIP runtime: devices.py)."""
import ctypes
import os
import threading

_state = None


def start(device, n=2):
    """Create ``n`` contexts on ``device`` in a background thread (the ctypes call releases the GIL)."""
    global _state
    if _state is not None:
        return
    path = os.environ.get("FSEG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfreddie_seg.so")
    st = {"device": int(device), "handles": [], "lib": None}

    def run():
        try:
            L = ctypes.CDLL(path)
            L.fseg_create.restype = ctypes.c_int
            L.fseg_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
            st["lib"] = L
            for _ in range(n):
                h = ctypes.c_void_p()
                if L.fseg_create(st["device"], ctypes.byref(h)) != 0:
                    break                      # the regular path reports the error (Context())
                st["handles"].append(h)
        except OSError:
            pass                               # a missing library: reported by _lib.load()

    st["thread"] = threading.Thread(target=run, daemon=True)
    st["thread"].start()
    _state = st


def started():
    """True once start() has been called in this process (HIP's start-up is under way or done)."""
    return _state is not None


def take(device):
    """The context handles start() has created for ``device`` (waits for it), or [] -- each handle is handed out once."""
    st = _state
    if st is None:
        return []
    st["thread"].join()
    hs, st["handles"] = st["handles"], []
    if st["device"] != int(device):
        discard(hs, st["lib"])
        return []
    return hs


def finish():
    """End of the process that called start(): contexts nobody took over (main() stopped before its batches) are destroyed --
    after the thread that makes them has ended, so the interpreter never exits with a thread inside the runtime's start-up."""
    st = _state
    if st is None:
        return
    st["thread"].join()
    hs, st["handles"] = st["handles"], []
    discard(hs, st["lib"])


def discard(handles, lib=None):
    lib = lib or (_state or {}).get("lib")
    if lib is None:
        return
    lib.fseg_destroy.restype = None
    lib.fseg_destroy.argtypes = [ctypes.c_void_p]
    for h in handles:
        lib.fseg_destroy(h)


def fast_exit_allowed():
    """May a finished CLI process (or worker) leave through os._exit(0), skipping the interpreter's and the HIP runtime's tear-down
    (0.15 s of a 2 M-read job's 0.8 s)?  Not when somebody is watching the process through an exit hook: a profiler's tool library
    (rocprofv3 sets ROCP_TOOL_LIBRARIES / preloads librocprofiler), coverage, a Python tracer or profiler, or FREDDIE_CLEAN_EXIT=1
    -- their output is written by finalisers that a hard exit skips."""
    import os
    import sys
    if os.environ.get("FREDDIE_CLEAN_EXIT") == "1":
        return False
    if os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or os.environ.get("COVERAGE_PROCESS_START") \
            or os.environ.get("HSA_TOOLS_LIB"):
        return False
    if any(k in os.environ.get("LD_PRELOAD", "") for k in ("rocprof", "roctracer", "rocprofiler", "asan", "tsan")):
        return False
    if sys.gettrace() is not None or sys.getprofile() is not None or "coverage" in sys.modules:
        return False
    return True
