#!/usr/bin/env python3
"""Generates the golden fixtures in this directory with the CPU oracle.

The reference ships no golden files and cannot be built here (Rust, no toolchain), so
these vectors are produced by the oracle restatement (oracle/scann_oracle.c), whose
arithmetic is pinned by the reference's own unit-test vectors
(tests/test_oracle_known_answers.py).  They freeze the oracle's stage-by-stage outputs so
that (a) the oracle cannot drift silently and (b) the HIP path is checked against files,
not only against a live oracle.

Stored per case: the trained index (centers, CSR, codebook, codes), CRC32 of the
regenerable inputs (rows, queries come from scann_rust_amd.synth), and per query the leaf
tokens + centre distances, the merged approximate candidates, and the final top-k.

    python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as orc  # noqa: E402
from scann_rust_amd import synth, trainer  # noqa: E402

N, L, K, NQ, KNN = 4096, 16, 16, 64, 10
CASES = [(seed, dim, S) for seed in (1, 2, 3) for dim, S in ((128, 32), (96, 24))]
SETTINGS = [(1, 3.0), (4, 3.0), (16, 10.0)]


def build_case(seed, dim, S):
    rows = synth.uniform_f32(N, dim, 1000 + seed)
    queries = synth.uniform_f32(NQ, dim, 2000 + seed)
    ix = trainer.build_txh_index(rows, L, S, K=K, seed=seed, kmeans_iters=4, pq_iters=4)
    return rows, queries, ix


def main():
    for seed, dim, S in CASES:
        rows, queries, ix = build_case(seed, dim, S)
        data, stride = orc.to_strided(rows)
        out = dict(centers=ix["centers"], leaf_off=ix["leaf_off"], leaf_ids=ix["leaf_ids"],
                   codebook=ix["codebook"], codes=ix["codes"],
                   rows_crc=np.uint32(zlib.crc32(rows.tobytes())),
                   queries_crc=np.uint32(zlib.crc32(queries.tobytes())))
        for P, mult in SETTINGS:
            m = orc.pre_reorder_k(KNN, mult)
            oix = orc.TxhIndex(data, stride, dim, ix["centers"], ix["leaf_off"], ix["leaf_ids"],
                               ix["codebook"], ix["codes"], partitions_to_search=P,
                               pre_reorder_multiplier=mult)
            tok = np.zeros((NQ, P), np.uint32); tokd = np.zeros((NQ, P), np.float32)
            ci = np.full((NQ, m), 0xFFFFFFFF, np.uint32); cd = np.full((NQ, m), np.inf, np.float32)
            cc = np.zeros(NQ, np.uint32)
            fi = np.full((NQ, KNN), 0xFFFFFFFF, np.uint32); fd = np.full((NQ, KNN), np.inf, np.float32)
            fc = np.zeros(NQ, np.uint32)
            for i in range(NQ):
                oi, od, otok, otokd, oci, ocd = orc.txh_search(oix, queries[i], KNN, stages=True)
                tok[i], tokd[i] = otok, otokd
                cc[i] = oci.size; ci[i, :oci.size] = oci; cd[i, :oci.size] = ocd
                fc[i] = oi.size; fi[i, :oi.size] = oi; fd[i, :oi.size] = od
            tag = "P%d_m%d" % (P, m)
            out.update({tag + "_tokens": tok, tag + "_token_dists": tokd, tag + "_cand_idx": ci,
                        tag + "_cand_dist": cd, tag + "_cand_count": cc, tag + "_idx": fi,
                        tag + "_dist": fd, tag + "_count": fc})
        path = os.path.join(HERE, "txh_seed%d_d%d_S%d.npz" % (seed, dim, S))
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path))

    # brute force + AsymmetricHasher vectors on one small dataset
    rows = synth.uniform_f32(2000, 64, 77)
    queries = synth.uniform_f32(16, 64, 78)
    data, stride = orc.to_strided(rows)
    out = dict(rows_crc=np.uint32(zlib.crc32(rows.tobytes())),
               queries_crc=np.uint32(zlib.crc32(queries.tobytes())))
    for name, meas in (("sql2", orc.SQUARED_L2), ("l2", orc.L2), ("dot", orc.DOT_PRODUCT)):
        oi, od, oc = orc.bf_search_batched(data, 2000, 64, stride, meas, queries, KNN)
        out["bf_%s_idx" % name] = oi
        out["bf_%s_dist" % name] = od
    ah = trainer.build_ah_index(rows, 8, K=16, seed=5, pq_iters=4)
    out["ah_codebook"] = ah["codebook"]
    out["ah_codes"] = ah["codes"]
    ai = np.zeros((16, KNN), np.uint32); ad = np.zeros((16, KNN), np.float32)
    ri = np.zeros((16, KNN), np.uint32); rd = np.zeros((16, KNN), np.float32)
    for i in range(16):
        ai[i], ad[i] = orc.ah_search(ah["codebook"], ah["codes"], queries[i], KNN)
        ri[i], rd[i] = orc.ah_search_with_reordering(ah["codebook"], ah["codes"], data, stride,
                                                     queries[i], KNN, 50)
    out.update(ah_idx=ai, ah_dist=ad, ahr_idx=ri, ahr_dist=rd)
    path = os.path.join(HERE, "bf_ah_n2000_d64.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
