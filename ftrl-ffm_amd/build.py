"""Builds libffm_engine.so (hand-written HIP for gfx950 + the C ABI) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libffm_engine.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: the reference's x86-64 build has no FMA, and parity is bit-level (ftrl_math.h).
# fp32 divide/sqrt stay correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def sources():
    return [os.path.join(CSRC, "engine.hip")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "ffm_engine.h"))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


HOST = os.path.join(HERE, "host")
HOST_SRCS = ["cmd_option.cpp", "parser.cpp", "reader.cpp", "ftrl_model.cpp", "trainer.cpp",
             "persist.cpp", "csr_reader.cpp", "csr_stream.cpp"]
MAIN_BIN = os.path.join(HOST, "ftrl_ffm_main")
TEST_BIN = os.path.join(HOST, "host_tests")


def build_host(force=False, verbose=False):
    """The C++17 host mirror of the reference's trainer (host/) linked against libffm_engine.so:
    the CLI `ftrl_ffm_main` and the `host_tests` executable."""
    build(force=False, verbose=verbose)
    srcs = [os.path.join(HOST, f) for f in HOST_SRCS]
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith((".h", ".cpp"))] + [LIB]
    for out, main in ((MAIN_BIN, "main.cpp"), (TEST_BIN, "host_tests.cpp")):
        if (not force and os.path.exists(out)
                and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps)):
            continue
        cmd = ["g++", "-std=c++17", "-O3", "-fopenmp", "-Wall", "-pthread", "-o", out,
               os.path.join(HOST, main)] + srcs + [LIB, "-ldl", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + HERE]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return MAIN_BIN, TEST_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_host(force="--force" in sys.argv, verbose=True)
