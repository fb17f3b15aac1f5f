"""Builds libscann_hip.so (gfx950) in-tree with hipcc.  No GPU needed to build."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libscann_hip.so")
SOURCES = ["api.hip", "txh.hip", "bf.hip", "index_file.hip"]
HEADERS = ["common.h", "txh.h", "bf.h", os.path.join("..", "..", "include", "scann_hip.h")]
# -ffp-contract=off: the reference never contracts a*b+c (Rust); FMA is used only via
# explicit fmaf()/MFMA where the reference uses _mm256_fmadd_ps.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False, extra_flags=()):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra_flags) + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


HOST = os.path.join(HERE, "host")
HOST_PROGRAMS = ["host_test", "ann_benchmark"]   # C++ mirror of the reference API + its benchmark CLI


def build_host(verbose=False):
    """Compile the host-side C++ programs (g++, link against libscann_hip.so)."""
    build()
    out = []
    for name in HOST_PROGRAMS:
        exe, src = os.path.join(HOST, name), os.path.join(HOST, name + ".cpp")
        deps = [src, os.path.join(HOST, "scann.hpp"), os.path.join(HERE, "..", "include", "scann_hip.h")]
        if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
            cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, src, "-L" + HERE, "-lscann_hip",
                   "-Wl,-rpath," + HERE, "-lpthread"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        out.append(exe)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_host(verbose=True)
    print(LIB)
