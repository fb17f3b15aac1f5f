#!/usr/bin/env python3
"""SearchMode::Partitioned at scale: GPU k-means partition of N x dim uniform rows, one batch of
queries through leaf_exact_scan_kernel.  Prints per-batch ms (HIP-event kernel time + wall)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scann_rust_amd import hip, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--leaves", type=int, default=1000)
    ap.add_argument("--P", type=int, default=20)
    ap.add_argument("--nq", type=int, default=2048)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--measure", type=int, default=0)
    a = ap.parse_args()
    rows = synth.uniform_f32(a.n, a.dim, 42)
    q = synth.uniform_f32(a.nq, a.dim, 123)
    bf = hip.bf_create(rows, a.n, a.dim, a.dim, hip.SQUARED_L2)
    c, assign = hip.kmeans_lloyd(bf, hip.kmeans_init_pp(bf, a.leaves, seed=42), max_iterations=10)[:2]
    del bf
    order = np.argsort(assign, kind="stable").astype(np.uint32)
    off = np.zeros(c.shape[0] + 1, np.uint32)
    off[1:] = np.cumsum(np.bincount(assign, minlength=c.shape[0]))
    ix = hip.txh_create(data=rows, n_rows=a.n, dim=a.dim, stride=a.dim, centers=c, leaf_offsets=off, leaf_ids=order,
                        codebook=None, codes=None, partitions_to_search=a.P, distance_measure=a.measure)
    ix.enable_timing()
    ix.search_batched(q, a.k)
    best = (1e9, 1e9)
    for _ in range(a.reps):
        t = time.perf_counter()
        ix.search_batched(q, a.k)
        wall = (time.perf_counter() - t) * 1e3
        ms, name = ix.last_kernel_ms()
        best = min(best, (wall, ms))
    rows_scanned = a.nq * a.P * (a.n / a.leaves)
    print("partitioned n=%d dim=%d L=%d P=%d nq=%d: wall %.3f ms (%.0f QPS)  %s %.3f ms = %.2f TFLOP/s (3 flop per dim)"
          % (a.n, a.dim, a.leaves, a.P, a.nq, best[0], a.nq / best[0] * 1e3, name, best[1],
             rows_scanned * a.dim * 3 / best[1] / 1e9))


if __name__ == "__main__":
    main()
