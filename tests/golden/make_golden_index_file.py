#!/usr/bin/env python3
"""Generates tests/golden/txh_small.scannidx (the SCANNIDX container, include/scann_hip.h "index
files") with the numpy writer, and txh_small_expected.npz: the CPU oracle's results on that index.

The fixture pins the FILE FORMAT: the GPU loader (scann_hip_index_load_file) and the numpy reader
(scann_rust_amd/index_file.py) must both read these committed bytes and reproduce the stored rows.

    python tests/golden/make_golden_index_file.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as orc  # noqa: E402
from scann_rust_amd import index_file, synth, trainer  # noqa: E402

N, DIM, L, S, K, P, MULT, NQ, KNN = 384, 32, 6, 8, 16, 3, 4.0, 24, 10


def main():
    rows = synth.uniform_f32(N, DIM, 31)
    queries = synth.uniform_f32(NQ, DIM, 32)
    data, stride = orc.to_strided(rows)
    ix = trainer.build_txh_index(rows, L, S, K=K, seed=5, kmeans_iters=4, pq_iters=4)
    path = os.path.join(HERE, "txh_small.scannidx")
    index_file.write(path, kind=1, n_rows=N, n_local=N, dim=DIM, stride=stride, num_partitions=L, num_subspaces=S,
                     num_codes=K, dims_per_subspace=DIM // S, use_residuals=1, partitions_to_search=P,
                     pre_reorder_multiplier=MULT,
                     sections=[("data", np.asarray(data, np.float32).reshape(-1)), ("centers", ix["centers"]),
                               ("leaf_offsets", ix["leaf_off"]), ("leaf_ids", ix["leaf_ids"]),
                               ("codebook", ix["codebook"]), ("codes", ix["codes"])])
    oix = orc.TxhIndex(data, stride, DIM, ix["centers"], ix["leaf_off"], ix["leaf_ids"], ix["codebook"],
                       ix["codes"], partitions_to_search=P, pre_reorder_multiplier=MULT)
    idx = np.full((NQ, KNN), 0xFFFFFFFF, np.uint32)
    dist = np.full((NQ, KNN), np.inf, np.float32)
    cnt = np.zeros(NQ, np.uint32)
    for i in range(NQ):
        oi, od = orc.txh_search(oix, queries[i], KNN)[:2]
        cnt[i] = oi.size
        idx[i, :oi.size], dist[i, :oi.size] = oi, od
    np.savez_compressed(os.path.join(HERE, "txh_small_expected.npz"), queries=queries, idx=idx, dist=dist, count=cnt)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
