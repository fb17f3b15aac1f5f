"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU: the oracle must reproduce them bit for bit.  GPU: the HIP path must too."""
import glob
import os
import re
import zlib

import numpy as np
import pytest

from oracle import pyoracle as orc
from scann_rust_amd import synth
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TXH_FILES = sorted(glob.glob(os.path.join(GOLD, "txh_seed*.npz")))
N, L, NQ, KNN = 4096, 16, 64, 10
SETTINGS = [(1, 3.0), (4, 3.0), (16, 10.0)]


def load_case(path):
    seed, dim, S = (int(x) for x in re.search(r"seed(\d+)_d(\d+)_S(\d+)", path).groups())
    g = np.load(path)
    rows = synth.uniform_f32(N, dim, 1000 + seed)
    queries = synth.uniform_f32(NQ, dim, 2000 + seed)
    assert np.uint32(zlib.crc32(rows.tobytes())) == g["rows_crc"], "synthetic generator drifted"
    assert np.uint32(zlib.crc32(queries.tobytes())) == g["queries_crc"]
    return seed, dim, S, g, rows, queries


def test_fixture_set_is_complete():
    assert len(TXH_FILES) == 6
    assert os.path.exists(os.path.join(GOLD, "bf_ah_n2000_d64.npz"))


@pytest.mark.parametrize("path", TXH_FILES, ids=[os.path.basename(p) for p in TXH_FILES])
def test_oracle_matches_golden_txh(path):
    seed, dim, S, g, rows, queries = load_case(path)
    data, stride = orc.to_strided(rows)
    for P, mult in SETTINGS:
        m = orc.pre_reorder_k(KNN, mult)
        tag = "P%d_m%d" % (P, m)
        oix = orc.TxhIndex(data, stride, dim, g["centers"], g["leaf_off"], g["leaf_ids"],
                           g["codebook"], g["codes"], partitions_to_search=P,
                           pre_reorder_multiplier=mult)
        for i in range(0, NQ, 3):
            oi, od, otok, otokd, oci, ocd = orc.txh_search(oix, queries[i], KNN, stages=True)
            assert np.array_equal(otok, g[tag + "_tokens"][i])
            assert np.array_equal(otokd.view(np.uint32), g[tag + "_token_dists"][i].view(np.uint32))
            c = g[tag + "_cand_count"][i]
            assert oci.size == c and np.array_equal(oci, g[tag + "_cand_idx"][i, :c])
            f = g[tag + "_count"][i]
            assert np.array_equal(oi, g[tag + "_idx"][i, :f])
            assert np.array_equal(od.view(np.uint32), g[tag + "_dist"][i, :f].view(np.uint32))


def test_oracle_matches_golden_bf_ah():
    g = np.load(os.path.join(GOLD, "bf_ah_n2000_d64.npz"))
    rows = synth.uniform_f32(2000, 64, 77)
    queries = synth.uniform_f32(16, 64, 78)
    assert np.uint32(zlib.crc32(rows.tobytes())) == g["rows_crc"]
    data, stride = orc.to_strided(rows)
    for name, meas in (("sql2", orc.SQUARED_L2), ("l2", orc.L2), ("dot", orc.DOT_PRODUCT)):
        oi, od, oc = orc.bf_search_batched(data, 2000, 64, stride, meas, queries, KNN)
        assert np.array_equal(oi, g["bf_%s_idx" % name])
        assert np.array_equal(od.view(np.uint32), g["bf_%s_dist" % name].view(np.uint32))
    for i in range(16):
        ai, ad = orc.ah_search(g["ah_codebook"], g["ah_codes"], queries[i], KNN)
        assert np.array_equal(ai, g["ah_idx"][i])
        ri, rd = orc.ah_search_with_reordering(g["ah_codebook"], g["ah_codes"], data, stride,
                                               queries[i], KNN, 50)
        assert np.array_equal(ri, g["ahr_idx"][i])
        assert np.array_equal(rd.view(np.uint32), g["ahr_dist"][i].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("path", TXH_FILES, ids=[os.path.basename(p) for p in TXH_FILES])
def test_hip_matches_golden_txh(path):
    from scann_rust_amd import hip
    seed, dim, S, g, rows, queries = load_case(path)
    data, stride = orc.to_strided(rows)
    index = hip.txh_create(data=data, n_rows=N, dim=dim, stride=stride, centers=g["centers"],
                           leaf_offsets=g["leaf_off"], leaf_ids=g["leaf_ids"],
                           codebook=g["codebook"], codes=g["codes"], partitions_to_search=4,
                           pre_reorder_multiplier=3.0)
    for P, mult in SETTINGS:
        m = orc.pre_reorder_k(KNN, mult)
        tag = "P%d_m%d" % (P, m)
        o = hip.default_opts()
        o.partitions_to_search = P
        o.pre_reorder_k = m
        idx, dist, cnt, (tok, tokd, ci, cd, cc) = index.search_batched(queries, KNN, o, stages=True)
        assert np.array_equal(tok[:, :P], g[tag + "_tokens"])
        assert np.array_equal(tokd[:, :P].view(np.uint32), g[tag + "_token_dists"].view(np.uint32))
        assert np.array_equal(cc, g[tag + "_cand_count"])
        assert np.array_equal(cnt, g[tag + "_count"])
        for i in range(NQ):
            H.assert_topk_equal_up_to_ties(ci[i, :cc[i]], cd[i, :cc[i]], g[tag + "_cand_idx"][i, :cc[i]],
                                           g[tag + "_cand_dist"][i, :cc[i]], what="cand q%d" % i)
            H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], g[tag + "_idx"][i, :cnt[i]],
                                           g[tag + "_dist"][i, :cnt[i]], what="final q%d" % i)


@pytest.mark.gpu
def test_hip_matches_golden_bf_ah():
    from scann_rust_amd import hip
    g = np.load(os.path.join(GOLD, "bf_ah_n2000_d64.npz"))
    rows = synth.uniform_f32(2000, 64, 77)
    queries = synth.uniform_f32(16, 64, 78)
    data, stride = orc.to_strided(rows)
    for name, meas in (("sql2", hip.SQUARED_L2), ("l2", hip.L2), ("dot", hip.DOT_PRODUCT)):
        index = hip.bf_create(data, 2000, 64, stride, meas)
        idx, dist, cnt = index.search_batched(queries, KNN)
        for i in range(16):
            H.assert_topk_equal_up_to_ties(idx[i], dist[i], g["bf_%s_idx" % name][i],
                                           g["bf_%s_dist" % name][i], what="%s q%d" % (name, i))
    index = hip.txh_create(data=data, n_rows=2000, dim=64, stride=stride, centers=None,
                           leaf_offsets=None, leaf_ids=None, codebook=g["ah_codebook"],
                           codes=g["ah_codes"], use_residuals=False, partitions_to_search=1,
                           pre_reorder_multiplier=1.0)
    o = hip.default_opts()
    o.exact_reorder = 0
    idx, dist, cnt = index.search_batched(queries, KNN, o)
    for i in range(16):
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], g["ah_idx"][i], g["ah_dist"][i], what="ah q%d" % i)
    o = hip.default_opts()
    o.pre_reorder_k = 50
    idx, dist, cnt = index.search_batched(queries, KNN, o)
    for i in range(16):
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], g["ahr_idx"][i], g["ahr_dist"][i], what="ahr q%d" % i)
