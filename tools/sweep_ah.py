#!/usr/bin/env python3
"""Recall / QPS sweep over pre_reorder_k for the AsymmetricHasher LUT16 workload
(development tool; prints one line per setting)."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--S", type=int, default=32)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ms", default="10,100,1000,2048,4096,6000,8192")
    ap.add_argument("--dist", default="uniform")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--pq-iters", type=int, default=8)
    ap.add_argument("--pq-sample", type=int, default=32768)
    a = ap.parse_args()
    import torch
    from scann_rust_amd import hip, synth, trainer
    L = hip.load()
    n, dim, S, Q, k = a.n, a.dim, a.S, a.batch, a.k
    stride = hip.compute_stride(dim)
    t0 = time.time()
    if a.dist == "uniform":
        rows = synth.uniform_f32(n, dim, 42)
        qs = synth.uniform_f32(Q, dim, 123)
    else:
        rows, _ = synth.clustered_f32(n, dim, 7, n_clusters=1000)
        qs, _ = synth.clustered_f32(Q, dim, 8, n_clusters=1000)
    data = np.zeros((n, stride), np.float32)
    data[:, :dim] = rows
    print("data %.1fs" % (time.time() - t0), flush=True)
    t0 = time.time()
    cb = trainer.train_codebook(rows[:: max(1, n // a.pq_sample)], S, 16, iters=a.pq_iters, seed=42,
                                sample=1 << 30)
    codes = hip.encode(cb, data, stride=stride)
    index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=None,
                           leaf_offsets=None, leaf_ids=None, codebook=cb, codes=codes,
                           use_residuals=False, partitions_to_search=1, pre_reorder_multiplier=1.0)
    bf = hip.bf_create(data, n, dim, stride, hip.SQUARED_L2)
    print("index %.1fs" % (time.time() - t0), flush=True)
    t0 = time.time()
    ti, td, tc = bf.search_batched(qs[:256], k)
    print("gt (gpu brute force, 256 q) %.2fs" % (time.time() - t0), flush=True)
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(qs).to(dev)
    oi = torch.empty((Q, k), dtype=torch.int32, device=dev)
    od = torch.empty((Q, k), dtype=torch.float32, device=dev)
    oc = torch.empty((Q,), dtype=torch.int32, device=dev)
    sptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for m in [int(x) for x in a.ms.split(",")]:
        o = hip.default_opts()
        o.pre_reorder_k = m
        gi, gd, gc = index.search_batched(qs[:256], k, o)
        rec = sum(len(set(gi[i].tolist()) & set(ti[i].tolist())) for i in range(256)) / (256.0 * k)
        hip.check(L.scann_hip_index_reserve(index.h, Q, k, ctypes.byref(o)))
        for _ in range(2):
            hip.check(L.scann_hip_search_batched_device(index.h, ctypes.c_void_p(qd.data_ptr()), Q, dim, k,
                                                        ctypes.byref(o), ctypes.c_void_p(oi.data_ptr()),
                                                        ctypes.c_void_p(od.data_ptr()),
                                                        ctypes.c_void_p(oc.data_ptr()), sptr))
        torch.cuda.synchronize()
        index.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            hip.check(L.scann_hip_search_batched_device(index.h, ctypes.c_void_p(qd.data_ptr()), Q, dim, k,
                                                        ctypes.byref(o), ctypes.c_void_p(oi.data_ptr()),
                                                        ctypes.c_void_p(od.data_ptr()),
                                                        ctypes.c_void_p(oc.data_ptr()), sptr))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        kms, kn = index.last_kernel_ms()
        index.enable_timing(False)
        hip.check(L.scann_hip_index_last_device_status(index.h, sptr))
        print("m=%5d recall10@10=%.4f  QPS=%.0f  ms/step=%.3f  %s=%.3f ms  (algo %.0f GB/s)"
              % (m, rec, Q * a.steps / el, el / a.steps * 1e3, kn, kms,
                 (n * S // 2 + S * 64 + k * 8) * Q / (kms * 1e-3) / 1e9 if kms else 0), flush=True)


if __name__ == "__main__":
    main()
