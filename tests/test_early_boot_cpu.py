"""The drop-in script's early context start-up (freddie_amd/_early.py) on a box without a GPU: it must never stand between
the user and an error message, and open_contexts() must work whether or not anything was started."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "py", "freddie_segment.py")


def run(*args):
    return subprocess.run([sys.executable, SHIM, *args], capture_output=True, text=True, timeout=120)


def test_missing_split_directory_is_reported_with_one_gpu_named(tmp_path):
    r = run("--gpus", "1", "-s", str(tmp_path / "no_such_dir"), "-o", str(tmp_path / "out"))
    assert r.returncode != 0
    assert "no_such_dir" in r.stderr


def test_argument_errors_surface_at_once():
    r = run("--gpus", "1", "--no-such-flag")
    assert r.returncode == 2 and "usage" in r.stderr.lower()


def test_take_without_start_and_for_another_device():
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from freddie_amd import _early\n"
            "assert _early.take(0) == []\n"
            "_early.start(0); _early.start(0)\n"          # the second call is a no-op
            "hs = _early.take(1)\n"                       # another device: nothing to take (and nothing leaks)
            "assert hs == [] and _early.take(0) == []\n"
            "_early.finish()\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_one_device_detection():
    code = ("import sys, runpy; sys.path.insert(0, %r)\n"
            "ns = runpy.run_path(%r, run_name='shim')\n"
            "f = ns['_one_device']\n"
            "def d(*a):\n"
            "    sys.argv = ['x', *a]; return f()\n"
            "assert d('--gpus', '1') == 0 and d('--gpus=1') == 0\n"
            "assert d('--devices', '3') == 3 and d('--devices=5', '--gpus', '1') == 5\n"
            "assert d('--gpus', '2') is None and d() is None and d('--devices', '0,1') is None\n"
            "assert d('--gpus', '1', '--devices', '0,1') is None\n") % (ROOT, SHIM)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
