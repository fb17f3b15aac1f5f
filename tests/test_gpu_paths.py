"""Every scan / re-rank path the library can take, forced through its environment switches and compared with the
ORACLE (not with one another): the driver's own `pytest -m gpu` run sees the whole matrix.

    SCANN_HIP_MFMA         0 = f32 LDS-gather scan, 2 = 32-pair integer-MFMA prefilter, 3 = 16-pair prefilter
    SCANN_HIP_SMFMAC       (with MFMA = 2) 1 = v_smfmac_i32_32x32x64_i8 (default), 0 = dense v_mfma_i32_32x32x32_i8
    SCANN_HIP_RESIDENT     (with MFMA = 0) 2 = resident-table scan kernel
    SCANN_HIP_RERANK_I8    0 = exact re-rank of every candidate, 2 = 8-bit row filter in front of it (any size)
    SCANN_HIP_RERANK_STORE int8 | fp8 row store of that filter (read at index creation)
    SCANN_HIP_RERANK_UNIFORM (int8 store, read at index creation) 0 = a {scale, error} pair per row; default: one
                           scale and one error bound for all rows when their magnitudes allow it

The cases mirror test_txh_search_stages / test_ah_search_with_reordering of test_gpu_parity.py at sizes where a
filter bound is in force (the prefilter needs one), plus the 1M x 128 headline index."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from scann_rust_amd import hip, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

# (MFMA, SMFMAC, RESIDENT, RERANK_I8, STORE)
MATRIX = [
    ("0", None, "0", "0", "int8"),
    ("0", None, "0", "2", "int8-perrow"),
    ("2", "1", None, "2", "int8-perrow"),
    ("0", None, "2", "2", "fp8"),
    ("2", "1", None, "0", "int8"),
    ("2", "1", None, "2", "int8"),
    ("2", "1", None, "2", "fp8"),
    ("2", "0", None, "0", "fp8"),
    ("2", "0", None, "2", "int8"),
    ("3", None, None, "0", "int8"),
    ("3", None, None, "2", "fp8"),
]
IDS = ["mfma%s%s%s-i8_%s-%s" % (m, "" if s is None else "-smfmac" + s, "" if r is None else "-res" + r, i, st)
       for m, s, r, i, st in MATRIX]


def _force(monkeypatch, mfma, smfmac, resident, i8, store):
    monkeypatch.setenv("SCANN_HIP_MFMA", mfma)
    for name, val in (("SCANN_HIP_SMFMAC", smfmac), ("SCANN_HIP_RESIDENT", resident)):
        if val is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, val)
    monkeypatch.setenv("SCANN_HIP_RERANK_I8", i8)
    monkeypatch.setenv("SCANN_HIP_RERANK_I8_MIN", "1")
    monkeypatch.setenv("SCANN_HIP_RERANK_STORE", store.split("-")[0])
    # the int8 store with a {scale, error} pair per row, or with one pair for all rows whatever their magnitudes
    monkeypatch.setenv("SCANN_HIP_RERANK_UNIFORM", "0" if store.endswith("-perrow") else "2")


def _expected_kernel(mfma, smfmac, resident):
    if mfma == "0":
        return "adc_scan_res_kernel" if resident == "2" else "adc_scan_kernel"
    if mfma == "3":
        return "adc_mfma16_kernel"
    return "adc_mfma_kernel" if smfmac == "0" else "adc_smfmac_kernel"


@pytest.mark.parametrize("mfma,smfmac,resident,i8,store", MATRIX, ids=IDS)
def test_txh_stages_every_path_against_oracle(mfma, smfmac, resident, i8, store, monkeypatch):
    """TreeXHybridSearcher::search stage by stage (tokens, candidate distances and sets, final rows) on a clustered
    index whose leaves are scanned by many queries, with a filter bound in force."""
    _force(monkeypatch, mfma, smfmac, resident, i8, store)
    n, dim, L, S, P, m, k = 80000, 96, 20, 24, 6, 250, 10
    rows, data, stride, ix, oix, kw = H.make_txh_case(n, dim, L, S, seed=32, P=P, mult=m / k, kmeans_iters=3,
                                                      pq_iters=3, clustered=True)
    index = hip.txh_create(**kw)
    q = synth.clustered_f32(96, dim, 33, n_clusters=L)[0]
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = P, m
    index.enable_timing(True)
    idx, dist, cnt, (tok, tokd, ci, cd, cc) = index.search_batched(q, k, o, stages=True)
    assert index.last_kernel_ms()[1] == _expected_kernel(mfma, smfmac, resident)
    for i in range(0, q.shape[0], 3):
        H.check_txh_query(oix, q[i], k, idx[i, :cnt[i]], dist[i, :cnt[i]], tok[i], tokd[i], ci[i, :cc[i]],
                          cd[i, :cc[i]], what="%s q%d" % (IDS[MATRIX.index((mfma, smfmac, resident, i8, store))], i))
    # the same rows without the stage outputs (the unsorted fast path: int8 / FP8 filter + shortlist re-rank)
    idx2, dist2, cnt2 = index.search_batched(q, k, o)
    assert np.array_equal(cnt2, cnt) and np.array_equal(dist2.view(np.uint32), dist.view(np.uint32))
    for i in range(q.shape[0]):
        H.assert_topk_equal_up_to_ties(idx2[i, :cnt[i]], dist2[i, :cnt[i]], idx[i, :cnt[i]], dist[i, :cnt[i]], what="fast q%d" % i)


@pytest.mark.parametrize("mfma,smfmac,resident,i8,store", MATRIX, ids=IDS)
def test_ah_reordering_every_path_against_oracle(mfma, smfmac, resident, i8, store, monkeypatch):
    """AsymmetricHasher::search_with_reordering (hashes/hasher.rs:186-215) on a flat index."""
    _force(monkeypatch, mfma, smfmac, resident, i8, store)
    n, dim, S, k, pre_k = 60000, 128, 32, 10, 300
    rows, data, stride, ix, kw = H.make_ah_case(n, dim, S, seed=31, pq_iters=3)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(64, dim, 77)
    o = hip.default_opts()
    o.pre_reorder_k = pre_k
    index.enable_timing(True)
    idx, dist, cnt = index.search_batched(q, k, o)
    assert index.last_kernel_ms()[1] == _expected_kernel(mfma, smfmac, resident)
    for i in range(0, q.shape[0], 4):
        oi, od = orc.ah_search_with_reordering(ix["codebook"], ix["codes"], data, stride, q[i], k, pre_k)
        assert cnt[i] == oi.size
        H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], oi, od, what="ahr q%d" % i)


@pytest.fixture(scope="module")
def ah_1m():
    n, dim, S = 1_000_000, 128, 32
    rows = synth.uniform_f32(n, dim, 42)
    data, stride = orc.to_strided(rows)
    from scann_rust_amd import trainer
    ix = trainer.build_ah_index(rows[:100000], S, K=16, seed=42, pq_iters=3)   # codebook from a sample
    codes = hip.encode(ix["codebook"], data, stride=stride)
    return dict(data=data, stride=stride, codebook=ix["codebook"], codes=codes, q=synth.uniform_f32(1024, dim, 123))


@pytest.mark.parametrize("mfma,smfmac,resident,i8,store",
                         [MATRIX[0], MATRIX[2], MATRIX[5], MATRIX[7], MATRIX[8], MATRIX[10]],
                         ids=[IDS[0], IDS[2], IDS[5], IDS[7], IDS[8], IDS[10]])
def test_ah_1m_headline_every_path(ah_1m, mfma, smfmac, resident, i8, store, monkeypatch):
    """BASELINE configs[2] (1M x 128, S = 32, pre_reorder_k = 5000, batch 1024) under each forced path: three rows
    against the oracle, every row against exact re-computation and against the default path."""
    b = ah_1m
    k, pre_k = 10, 5000
    o = hip.default_opts()
    o.pre_reorder_k = pre_k
    kw = dict(data=b["data"], n_rows=1_000_000, dim=128, stride=b["stride"], centers=None, leaf_offsets=None, leaf_ids=None,
              codebook=b["codebook"], codes=b["codes"], codes_packed4=False, use_residuals=False, partitions_to_search=1,
              pre_reorder_multiplier=1.0)
    for name in ("SCANN_HIP_MFMA", "SCANN_HIP_SMFMAC", "SCANN_HIP_RESIDENT", "SCANN_HIP_RERANK_I8", "SCANN_HIP_RERANK_I8_MIN",
                 "SCANN_HIP_RERANK_STORE", "SCANN_HIP_RERANK_UNIFORM"):
        monkeypatch.delenv(name, raising=False)
    ref = hip.txh_create(**kw).search_batched(b["q"], k, o)      # the default heuristics
    _force(monkeypatch, mfma, smfmac, resident, i8, store)
    index = hip.txh_create(**kw)
    index.enable_timing(True)
    idx, dist, cnt = index.search_batched(b["q"], k, o)
    assert index.last_kernel_ms()[1] == _expected_kernel(mfma, smfmac, resident)
    assert np.array_equal(cnt, ref[2]) and np.array_equal(dist.view(np.uint32), ref[1].view(np.uint32))
    for i in range(0, 1024, 7):
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], ref[0][i], ref[1][i], what="vs default q%d" % i)
        want = orc.one_to_many(b["q"][i], b["data"][idx[i]].ravel(), 128, k, hip.SQUARED_L2)
        assert np.array_equal(want.view(np.uint32), dist[i].view(np.uint32))
    for i in (0, 511, 1023):
        oi, od = orc.ah_search_with_reordering(b["codebook"], b["codes"], b["data"], b["stride"], b["q"][i], k, pre_k)
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], oi, od, what="oracle q%d" % i)


def test_ah_1m_single_queries_take_the_wide_pipeline(ah_1m, monkeypatch):
    """One to four queries over the flat 1M hasher at pre_reorder_k = 5000 (the ann_benchmark operating point,
    bin/ann_benchmark.rs:172-178): the wide few-query pipeline (three launches), rows identical to the same queries'
    rows inside a batch of 1024 and to the oracle's."""
    b = ah_1m
    k, pre_k = 10, 5000
    o = hip.default_opts()
    o.pre_reorder_k = pre_k
    for name in ("SCANN_HIP_MFMA", "SCANN_HIP_SMFMAC", "SCANN_HIP_RESIDENT", "SCANN_HIP_RERANK_I8", "SCANN_HIP_RERANK_I8_MIN",
                 "SCANN_HIP_RERANK_STORE", "SCANN_HIP_RERANK_UNIFORM", "SCANN_HIP_SMALL", "SCANN_HIP_WIDE"):
        monkeypatch.delenv(name, raising=False)
    index = hip.txh_create(data=b["data"], n_rows=1_000_000, dim=128, stride=b["stride"], centers=None, leaf_offsets=None,
                           leaf_ids=None, codebook=b["codebook"], codes=b["codes"], codes_packed4=False, use_residuals=False,
                           partitions_to_search=1, pre_reorder_multiplier=1.0)
    index.enable_timing(True)
    ref = index.search_batched(b["q"], k, o)
    for lo, nq in ((0, 1), (1, 1), (100, 4), (511, 2), (1023, 1)):
        idx, dist, cnt = index.search_batched(b["q"][lo:lo + nq], k, o)
        assert index.last_kernel_ms()[1] == "wide_scan_kernel"
        assert np.array_equal(cnt, ref[2][lo:lo + nq])
        assert np.array_equal(dist.view(np.uint32), ref[1][lo:lo + nq].view(np.uint32))
        assert np.array_equal(idx, ref[0][lo:lo + nq])
    for i in (0, 1023):
        idx, dist, cnt = index.search_batched(b["q"][i:i + 1], k, o)
        oi, od = orc.ah_search_with_reordering(b["codebook"], b["codes"], b["data"], b["stride"], b["q"][i], k, pre_k)
        H.assert_topk_equal_up_to_ties(idx[0], dist[0], oi, od, what="oracle q%d" % i)
