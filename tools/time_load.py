#!/usr/bin/env python3
"""Times scann_hip_index_load_file (mmap -> pinned -> device) against create-from-host-arrays on a
1M x 128 Tree-X-Hybrid-shaped index (random codes: the load path does not care)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scann_rust_amd import hip, synth  # noqa: E402


def main():
    n, dim, L, S = int(os.environ.get("N", 1000000)), 128, 1000, 32
    rows = synth.uniform_f32(n, dim, 42)
    rng = np.random.default_rng(1)
    kw = dict(data=rows, n_rows=n, dim=dim, stride=dim, centers=rng.random((L, dim), dtype=np.float32),
              leaf_offsets=np.linspace(0, n, L + 1).astype(np.uint32), leaf_ids=rng.permutation(n).astype(np.uint32),
              codebook=rng.random((S, 16, dim // S), dtype=np.float32),
              codes=rng.integers(0, 16, (n, S), dtype=np.uint8))
    hip.context(0)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.scannidx")
        t = time.perf_counter(); hip.txh_write_file(p, **kw); tw = time.perf_counter() - t
        size = os.path.getsize(p)
        t = time.perf_counter(); a = hip.txh_create(**kw); tc = time.perf_counter() - t
        del a
        out = {}
        for pin in ("1", "0", "1"):
            os.environ["SCANN_HIP_LOAD_PIN"] = pin
            t = time.perf_counter(); b = hip.load_file(p); out.setdefault(pin, []).append(time.perf_counter() - t)
            del b
        print("file %.1f MB  write %.3f s  create(host arrays) %.3f s  load pinned %s s  load unpinned %s s" % (
            size / 1e6, tw, tc, ["%.3f" % v for v in out["1"]], ["%.3f" % v for v in out["0"]]))


if __name__ == "__main__":
    main()
