#!/usr/bin/env python3
"""Tree-X-Hybrid (north-star path) recall / QPS sweep over partitions_to_search and
pre_reorder_k.  Development / measurement tool: index building uses torch on the GPU as
harness plumbing (k-means assignment GEMMs); the SEARCH path is libscann_hip.so only.

    python tools/sweep_txh.py --num-points 1000000 --leaves 1000 --dist clustered
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kmeans_torch(X, k, iters, seed):
    import torch
    g = torch.Generator(device=X.device)
    g.manual_seed(seed)
    n = X.shape[0]
    C = X[torch.randperm(n, generator=g, device=X.device)[:k]].clone()
    assign = None
    for _ in range(iters + 1):
        cn = (C * C).sum(1)
        parts = []
        for r0 in range(0, n, 262144):
            x = X[r0:r0 + 262144]
            parts.append((cn[None, :] - 2.0 * (x @ C.T)).argmin(1))
        assign = torch.cat(parts)
        if _ == iters:
            break
        sums = torch.zeros_like(C).index_add_(0, assign, X)
        cnt = torch.bincount(assign, minlength=k).clamp(min=1).to(X.dtype)
        newC = sums / cnt[:, None]
        keep = torch.bincount(assign, minlength=k) == 0
        newC[keep] = C[keep]
        C = newC
    return C, assign


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-points", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--S", type=int, default=32)
    ap.add_argument("--leaves", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--dist", default="clustered", choices=["clustered", "uniform"])
    ap.add_argument("--Ps", default="10,25,50,100")
    ap.add_argument("--ms", default="30,100,300")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--kmeans-iters", type=int, default=8)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    import torch
    from scann_rust_amd import hip, synth, trainer
    Lh = hip.load()
    dev = torch.device("cuda", 0)
    n, dim, S, L, Q, k = a.n, a.dim, a.S, a.leaves, a.batch, a.k
    stride = hip.compute_stride(dim)
    t0 = time.time()
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    if a.dist == "uniform":
        X = torch.rand((n, dim), generator=g, device=dev)
        Xq = torch.rand((Q, dim), generator=g, device=dev)
    else:  # mixture of 1000 Gaussians, sigma = 0.1 * sqrt(dim / 6)
        cen = torch.rand((1000, dim), generator=g, device=dev)
        sig = 0.1 * (1.0 / 6.0) ** 0.5   # per-dim sigma: 0.1 x inter-centre distance sqrt(d/6), / sqrt(d)
        X = cen[torch.randint(0, 1000, (n,), generator=g, device=dev)] + \
            sig * torch.randn((n, dim), generator=g, device=dev)
        Xq = cen[torch.randint(0, 1000, (Q,), generator=g, device=dev)] + \
            sig * torch.randn((Q, dim), generator=g, device=dev)
    print("data %.1fs" % (time.time() - t0), flush=True)
    t0 = time.time()
    C, assign = kmeans_torch(X, L, a.kmeans_iters, 42)
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=L)
    leaf_off = np.zeros(L + 1, np.uint32)
    leaf_off[1:] = np.cumsum(counts.cpu().numpy())
    leaf_ids = order.cpu().numpy().astype(np.uint32)
    centers = C.cpu().numpy().astype(np.float32)
    leaf_of_row = assign[order].cpu().numpy().astype(np.uint32)
    rows_csr = X[order]
    res = (rows_csr - C[assign[order]])
    sample = res[torch.randperm(n, generator=g, device=dev)[:32768]].cpu().numpy()
    codebook = trainer.train_codebook(sample, S, 16, iters=8, seed=42, sample=1 << 30)
    data = np.zeros((n, stride), np.float32)
    data[:, :dim] = X.cpu().numpy()
    rows_csr_np = np.zeros((n, stride), np.float32)
    rows_csr_np[:, :dim] = rows_csr.cpu().numpy()
    codes = hip.encode(codebook, rows_csr_np, stride=stride, centers=centers, leaf_of_row=leaf_of_row)
    del rows_csr_np, res, rows_csr
    print("index build %.1fs  leaf sizes min/mean/max = %d/%.0f/%d"
          % (time.time() - t0, counts.min().item(), counts.float().mean().item(), counts.max().item()),
          flush=True)
    index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=centers,
                           leaf_offsets=leaf_off, leaf_ids=leaf_ids, codebook=codebook, codes=codes,
                           use_residuals=True, partitions_to_search=10, pre_reorder_multiplier=3.0)
    bf = hip.bf_create(data, n, dim, stride, hip.SQUARED_L2)
    qs = Xq.cpu().numpy().astype(np.float32)
    ne = 256
    ti, td, tc = bf.search_batched(qs[:ne], k)
    bf.close()
    sizes = (leaf_off[1:] - leaf_off[:-1]).astype(np.int64)
    qd = torch.from_numpy(qs).to(dev)
    oi = torch.empty((Q, k), dtype=torch.int32, device=dev)
    od = torch.empty((Q, k), dtype=torch.float32, device=dev)
    oc = torch.empty((Q,), dtype=torch.int32, device=dev)
    sptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    results = []
    for P in [int(x) for x in a.Ps.split(",")]:
        tok, _, _ = hip.txh_partition(index, qs, P)
        scanned = sizes[tok.astype(np.int64)].sum(1).mean()
        for m in [int(x) for x in a.ms.split(",")]:
            o = hip.default_opts()
            o.partitions_to_search = P
            o.pre_reorder_k = m
            gi, gd, gc = index.search_batched(qs[:ne], k, o)
            rec = sum(len(set(gi[i].tolist()) & set(ti[i].tolist())) for i in range(ne)) / (ne * float(k))
            hip.check(Lh.scann_hip_index_reserve(index.h, Q, k, ctypes.byref(o)))

            def run():
                hip.check(Lh.scann_hip_search_batched_device(
                    index.h, ctypes.c_void_p(qd.data_ptr()), Q, dim, k, ctypes.byref(o),
                    ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(od.data_ptr()),
                    ctypes.c_void_p(oc.data_ptr()), sptr))
            for _ in range(10):   # (also lets the clocks settle after the previous configuration's kernels)
                run()
            torch.cuda.synchronize()
            index.enable_timing(True)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                run()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            kms, kn = index.last_kernel_ms()
            index.enable_timing(False)
            hip.check(Lh.scann_hip_index_last_device_status(index.h, sptr))
            # SURVEY.md 8d, Tree-X-Hybrid per query, with the ACTUAL scanned leaf sizes
            algo = L * dim * 4 + scanned * (S // 2) + scanned * 4 + P * S * 64 + m * dim * 4 + k * 8
            scan_bytes = scanned * (S // 2) + P * S * 64
            r = dict(P=P, m=m, recall=rec, qps=Q * a.steps / el, ms_per_step=el / a.steps * 1e3,
                     scan_ms=kms, scanned_points=float(scanned), algo_bytes_per_query=float(algo),
                     scan_algo_gbps=scan_bytes * Q / (kms * 1e-3) / 1e9 if kms else 0.0)
            results.append(r)
            print("P=%4d m=%5d recall10@10=%.4f QPS=%9.0f ms/step=%7.3f scan=%.3f ms scanned/q=%.0f "
                  "scan algo %.0f GB/s" % (P, m, rec, r["qps"], r["ms_per_step"], kms, scanned,
                                           r["scan_algo_gbps"]), flush=True)
    if a.json:
        json.dump(dict(n=n, dim=dim, S=S, L=L, batch=Q, k=k, dist=a.dist, results=results),
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
