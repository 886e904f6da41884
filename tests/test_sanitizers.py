"""AddressSanitizer + UndefinedBehaviorSanitizer builds of the native CPU code (the host I/O library and the C oracle),
exercised by the same tests as the regular builds, in a child interpreter with the sanitizer runtimes preloaded
(sanitizers run on the CPU build; the GPU pool has no ASan).  Any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build_asan")
FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-fPIC"]


def runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.fixture(scope="module")
def sanitized():
    asan, ubsan = runtime("libasan.so"), runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("the sanitizer runtimes are not installed")
    os.makedirs(OUT, exist_ok=True)
    host = os.path.join(OUT, "libfreddie_host_asan.so")
    oracle = os.path.join(OUT, "libfreddie_oracle_asan.so")
    subprocess.check_call(["g++", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include")] + FLAGS +
                          ["-o", host, os.path.join(ROOT, "freddie_amd", "csrc", "freddie_host.cpp")])
    subprocess.check_call(["gcc", "-ffp-contract=off", "-fno-fast-math"] + FLAGS +
                          ["-o", oracle, os.path.join(ROOT, "oracle", "freddie_oracle.c"), "-lm"])
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, FHOST_LIB=host, FREDDIE_ORACLE_SO=oracle,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the child really runs the instrumented builds (not the regular ones next to the sources)
    probe = ("import sys; sys.path.insert(0, %r); from freddie_amd import _host; from oracle import oracle; _host.load(); oracle.lib(); "
             "print(open('/proc/self/maps').read())" % ROOT)
    maps = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=300).stdout
    assert "libfreddie_host_asan.so" in maps and "libfreddie_oracle_asan.so" in maps and "libasan" in maps
    assert "/freddie_amd/libfreddie_host.so" not in maps and "/oracle/libfreddie_oracle.so" not in maps
    return env


def run_tests(env, *args):
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + list(args), cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    tail = (res.stdout[-3000:] + res.stderr[-3000:])
    assert res.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    return res.stdout


def test_host_library_under_asan_ubsan(sanitized):
    out = run_tests(sanitized, "tests/test_host_native.py")
    assert " passed" in out


def test_oracle_under_asan_ubsan(sanitized):
    out = run_tests(sanitized, "tests/test_oracle_golden.py", "tests/test_oracle_units.py", "-k", "not config2")
    assert " passed" in out
