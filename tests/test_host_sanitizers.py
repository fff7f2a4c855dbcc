"""Sanitizer leg of the host code ("Race detection / sanitizers": the reference
has none; its per-sample thread pool, src/concurrent, is replaced here by the block scheduler's
parse-ahead workers and the staging ring, which is where host races would live).

Builds host/host_tests.cpp + the host mirror with -fsanitize=address,undefined and again with
-fsanitize=thread (CPU build only: GPU sanitizers are not available on the pool), then runs the CPU
checks and the multi-threaded `ingest` / `stream` sub-commands at 8 threads on a generated libffm
file.  A clean run = exit code 0 and no sanitizer report in the output."""
import os
import subprocess

import numpy as np
import pytest

import ftrl_ffm_amd as fa
fbuild = fa._build  # (ftrl-ffm_amd/build.py: the host sources and paths)

REPORTS = ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "WARNING: ThreadSanitizer", "runtime error:")


def _build(tmp_path, name, flags):
    lib = fa.build()
    out = str(tmp_path / ("host_tests_" + name))
    srcs = [os.path.join(fbuild.HOST, f) for f in fbuild.HOST_SRCS]
    cmd = (["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fopenmp", "-pthread"] + flags +
           ["-o", out, os.path.join(fbuild.HOST, "host_tests.cpp")] + srcs +
           [lib, "-ldl", "-Wl,-rpath," + os.path.dirname(lib)])
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    return out


def _libffm_file(tmp_path, rows=20000, fields=12, per=500):
    rng = np.random.default_rng(5)
    ids = rng.integers(0, per, (rows, fields)) + np.arange(fields) * per
    vals = np.round(rng.uniform(0.0001, 1.0, (rows, fields)), 4)
    y = rng.integers(0, 2, rows)
    p = tmp_path / "san.ffm"
    with open(p, "w") as f:
        for r in range(rows):
            f.write(str(y[r]) + " " + " ".join("%d:%d:%s" % (j, ids[r, j], vals[r, j]) for j in range(fields)) + "\n")
    return str(p)


@pytest.mark.parametrize("name,flags,env", [
    ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
     {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"}),
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=0:second_deadlock_stack=1"}),
])
def test_host_code_is_clean_under_sanitizers(tmp_path, name, flags, env):
    exe = _build(tmp_path, name, flags)
    data = _libffm_file(tmp_path)
    run_env = dict(os.environ, OMP_NUM_THREADS="8", **env)
    for args in (["cpu"], ["ingest", data, "libffm", "8"], ["stream", data, "libffm", "8"]):
        out = subprocess.run([exe] + args, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=run_env)
        log = out.stdout + out.stderr
        assert out.returncode == 0, name + " " + " ".join(args) + "\n" + log[-4000:]
        for mark in REPORTS:
            assert mark not in log, name + " " + " ".join(args) + "\n" + log[-4000:]
        if args == ["cpu"]:
            assert "0 failed" in out.stdout
