#!/usr/bin/env python3
"""Times the GPU index build blocks on synthetic data: partitioner k-means
(k-means++ + Lloyd, trees/kmeans.rs), per-subspace codebook k-means on residuals
(hashes/codebook.rs:146-202) and the residual encode.  libscann_hip.so only.

    python tools/time_build.py --num-points 1000000 --leaves 1000
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-points", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--leaves", type=int, default=1000)
    ap.add_argument("--S", type=int, default=32)
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--kmeans-iters", type=int, default=100)
    ap.add_argument("--pq-iters", type=int, default=25)
    a = ap.parse_args()
    from scann_rust_amd import hip, synth
    rows, _ = synth.clustered_f32(a.n, a.dim, 7, n_clusters=a.leaves)
    stride = hip.compute_stride(a.dim)
    data = np.zeros((a.n, stride), np.float32)
    data[:, :a.dim] = rows
    t0 = time.time()
    bf = hip.bf_create(data, a.n, a.dim, stride, hip.SQUARED_L2)
    t1 = time.time()
    init = hip.kmeans_init_pp(bf, a.leaves, seed=42)
    t2 = time.time()
    centers, assign, sizes, inertia, iters, conv = hip.kmeans_lloyd(bf, init, max_iterations=a.kmeans_iters)
    t3 = time.time()
    print("upload %.2fs  k-means++ (k=%d) %.2fs  Lloyd %d iterations %.2fs (%.1f ms/iter) converged=%s "
          "inertia=%.6g leaf sizes min/mean/max=%d/%d/%d"
          % (t1 - t0, a.leaves, t2 - t1, iters, t3 - t2, 1e3 * (t3 - t2) / max(1, iters), conv, inertia,
             sizes.min(), sizes.mean(), sizes.max()), flush=True)
    resid = np.zeros((a.n, stride), np.float32)
    resid[:, :a.dim] = rows - centers[assign]
    rbf = hip.bf_create(resid, a.n, a.dim, stride, hip.SQUARED_L2)
    dsub = a.dim // a.S
    cb = np.zeros((a.S, a.K, dsub), np.float32)
    t4 = time.time()
    tot_it = 0
    for s in range(a.S):
        c0 = hip.kmeans_init_pp(rbf, a.K, seed=42 + s, col_offset=s * dsub, sub_dim=dsub)
        c, _, _, _, it, _ = hip.kmeans_lloyd(rbf, c0, max_iterations=a.pq_iters, col_offset=s * dsub)
        cb[s] = c
        tot_it += it
    t5 = time.time()
    codes = hip.encode(cb, resid, stride=stride)
    t6 = time.time()
    print("codebook: %d subspaces x %d codes, %d Lloyd iterations total, %.2fs ; encode %.2fs ; "
          "build total %.2fs" % (a.S, a.K, tot_it, t5 - t4, t6 - t5, t6 - t0), flush=True)


if __name__ == "__main__":
    main()
