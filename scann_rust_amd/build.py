"""Builds libscann_hip.so (gfx950) in-tree with hipcc.  No GPU needed to build."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libscann_hip.so")
SOURCES = ["api.hip", "txh.hip", "bf.hip"]
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


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
