"""Builds libscann_hip.so (gfx950) in-tree with hipcc.  No GPU needed to build.

Every source is compiled to its own object (in parallel, rebuilt only when it or a header
changed) and the objects are linked into the shared library, so editing one kernel file costs one
compile."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libscann_hip.so")
SOURCES = ["api.hip", "txh.hip", "bf.hip", "index_file.hip", "comm.hip"]
HEADERS = ["common.h", "txh.h", "bf.h", "comm.h", os.path.join("..", "..", "include", "scann_hip.h")]
# -ffp-contract=off: the reference never contracts a*b+c (Rust); FMA is used only via
# explicit fmaf()/MFMA where the reference uses _mm256_fmadd_ps.
CFLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-Wall",
          "-Wno-unused-function"]
LDFLAGS = ["-shared", "-fPIC", "--offload-arch=gfx950", "-ldl"]


def src_sha256():
    """sha256 over the library's sources, headers and compile flags: identifies a build of the library whatever
    directory or machine it was compiled in (the .so itself embeds build paths, so its own hash changes with them)."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(SOURCES) + sorted(os.path.normpath(os.path.join(CSRC, f)) for f in HEADERS)
    for f in names:
        p = f if os.path.isabs(f) else os.path.join(CSRC, f)
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(CFLAGS + LDFLAGS).encode())
    return h.hexdigest()


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _newest_header():
    t = 0.0
    for f in HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            t = max(t, os.path.getmtime(p))
    return max(t, os.path.getmtime(os.path.abspath(__file__)))


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale_objects(force, extra_flags):
    th = _newest_header()
    out = []
    for s in _sources():
        o, p = _obj(s), os.path.join(CSRC, s)
        if force or extra_flags or not os.path.exists(o) or os.path.getmtime(o) < max(th, os.path.getmtime(p)):
            out.append(s)
    return out


def build_variant(name, extra_flags, only=("txh.hip",)):
    """A tuning variant of the library: `only` sources recompiled with extra -D flags, the other objects
    reused; written to libscann_hip_<name>.so (select with SCANN_HIP_LIB)."""
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in _sources():
        if s in only:
            o = os.path.join(OBJ, os.path.splitext(s)[0] + "_" + name + ".o")
            subprocess.check_call([hipcc] + CFLAGS + list(extra_flags) + ["-c", "-o", o, os.path.join(CSRC, s)], cwd=CSRC)
            objs.append(o)
        else:
            objs.append(_obj(s))
    out = os.path.join(HERE, "libscann_hip_%s.so" % name)
    subprocess.check_call([hipcc] + LDFLAGS + ["-o", out] + objs, cwd=CSRC)
    return out


def build(force=False, verbose=False, extra_flags=()):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    todo = _stale_objects(force, extra_flags)
    objs = [_obj(s) for s in _sources()]
    if not todo and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(o) for o in objs):
        return LIB

    def compile_one(s):
        cmd = [hipcc] + CFLAGS + list(extra_flags) + ["-c", "-o", _obj(s), os.path.join(CSRC, s)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc] + LDFLAGS + ["-o", LIB] + objs
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
