"""Full-size (BASELINE.json: 1M x 128) checks through size-independent properties, plus
oracle spot checks on a few queries (the oracle needs ~0.1-1 s per query at this size).

Properties: ascending distances; exact re-rank distances of the returned indices are
bit-identical to the oracle's AVX2 arithmetic; batch-split invariance (a query's row does
not depend on which batch it travelled in); idempotence; monotone recall in
pre_reorder_k; no point outside the result beats the k-th distance (sampled)."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from scann_rust_amd import hip, synth, trainer
from tests import helpers as H

pytestmark = pytest.mark.gpu

N, DIM, S, K = 1_000_000, 128, 32, 10


@pytest.fixture(scope="module")
def big():
    rows = synth.uniform_f32(N, DIM, 42)
    stride = hip.compute_stride(DIM)
    data = np.ascontiguousarray(rows)          # dim 128 == stride 128
    assert stride == DIM
    sample = rows[:: N // 32768]
    cb = trainer.train_codebook(sample, S, 16, iters=6, seed=42, sample=1 << 30)
    codes = hip.encode(cb, data, stride=stride)
    q = synth.uniform_f32(64, DIM, 123)
    return dict(rows=rows, data=data, stride=stride, cb=cb, codes=codes, q=q)


def test_encode_full_size_matches_oracle_sample(big):
    sel = np.arange(0, N, 9973)
    assert np.array_equal(big["codes"][sel], orc.encode_many(big["cb"], big["rows"][sel]))


def test_ah_lut16_1m_properties_and_oracle(big):
    index = hip.txh_create(data=big["data"], n_rows=N, dim=DIM, stride=big["stride"], centers=None,
                           leaf_offsets=None, leaf_ids=None, codebook=big["cb"], codes=big["codes"],
                           use_residuals=False, partitions_to_search=1, pre_reorder_multiplier=1.0)
    q = big["q"]
    o = hip.default_opts()
    o.pre_reorder_k = 400
    idx, dist, cnt = index.search_batched(q, K, o)
    assert np.all(cnt == K)
    assert np.all(np.diff(dist, axis=1) >= 0)                       # ascending
    for i in range(q.shape[0]):                                      # exact distances, bit for bit
        for j in range(K):
            assert np.float32(orc.squared_l2_avx2(q[i], big["rows"][idx[i, j]])) == dist[i, j]
        assert len(set(idx[i].tolist())) == K
    # batch-split invariance + idempotence
    parts = [index.search_batched(q[a:a + 8], K, o) for a in range(0, 64, 8)]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), idx)
    assert np.array_equal(np.concatenate([p[1] for p in parts]).view(np.uint32), dist.view(np.uint32))
    idx2, dist2, _ = index.search_batched(q, K, o)
    assert np.array_equal(idx2, idx) and np.array_equal(dist2.view(np.uint32), dist.view(np.uint32))
    # oracle spot check: AsymmetricHasher::search_with_reordering on the same index
    for i in range(3):
        oi, od = orc.ah_search_with_reordering(big["cb"], big["codes"], big["data"], big["stride"],
                                               q[i], K, 400)
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], oi, od, what="q%d" % i)
    # approximate top-k without re-ordering (AsymmetricHasher::search)
    o2 = hip.default_opts()
    o2.exact_reorder = 0
    aidx, adist, _ = index.search_batched(q[:2], K, o2)
    for i in range(2):
        oi, od = orc.ah_search(big["cb"], big["codes"], q[i], K)
        H.assert_topk_equal_up_to_ties(aidx[i], adist[i], oi, od, what="approx q%d" % i)
    # recall is monotone in pre_reorder_k (candidate sets are nested)
    bf = hip.bf_create(big["data"], N, DIM, big["stride"], hip.SQUARED_L2)
    ti, td, _ = bf.search_batched(q, K)
    prev = -1.0
    for m in (10, 100, 1000, 5000):
        o.pre_reorder_k = m
        gi, _, _ = index.search_batched(q, K, o)
        rec = H.recall_at_k(gi, ti, K)
        assert rec >= prev - 1e-9
        prev = rec
    assert prev > 0.9


@pytest.mark.parametrize("measure,nq", [(hip.DOT_PRODUCT, 40), (hip.SQUARED_L2, 40), (hip.SQUARED_L2, 300),
                                        (hip.L2, 300)])
def test_brute_force_1m_properties_and_oracle(big, measure, nq):
    """nq = 40 runs the generic row-per-thread kernel (MFMA kernel for DotProduct), nq = 300 the
    one-query-per-lane kernel."""
    index = hip.bf_create(big["data"], N, DIM, big["stride"], measure)
    q = synth.uniform_f32(nq, DIM, 123)
    idx, dist, cnt = index.search_batched(q, K)
    assert np.all(cnt == K) and np.all(np.diff(dist, axis=1) >= 0)
    # oracle on 2 full queries (TopK over all 1M rows)
    for i in range(2):
        oi, od = orc.bf_search(big["data"], N, DIM, big["stride"], measure, q[i], K)
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], oi, od, what="bf q%d" % i)
    # returned distances are the oracle's arithmetic, and no sampled outsider beats the k-th
    sel = np.arange(7, N, 4099)
    for i in range(q.shape[0]):
        want = orc.one_to_many(q[i], big["rows"][idx[i]].ravel(), DIM, K, measure)
        assert np.array_equal(want.view(np.uint32), dist[i].view(np.uint32))
        other = orc.one_to_many(q[i], big["rows"][sel].ravel(), DIM, sel.size, measure)
        inside = np.isin(sel, idx[i])
        assert np.all(other[~inside] >= dist[i, -1])
    # batch-split invariance
    a, b, _ = index.search_batched(q[:5], K)
    assert np.array_equal(a, idx[:5]) and np.array_equal(b.view(np.uint32), dist[:5].view(np.uint32))


def test_txh_1m_gpu_build_and_oracle():
    """Tree-X-Hybrid at 1M x 128 with the index built by the library's GPU k-means (partitioner +
    per-subspace codebook on residuals + residual encode), searched on the GPU and checked stage
    by stage against the oracle on a few queries (the oracle scans only the selected leaves)."""
    n, dim, L, S, P, m, k = 1_000_000, 128, 1000, 32, 10, 300, 10
    rows, _ = synth.clustered_f32(n, dim, 7, n_clusters=1000)
    stride = hip.compute_stride(dim)
    data = np.ascontiguousarray(rows)
    bf = hip.bf_create(data, n, dim, stride, hip.SQUARED_L2)
    centers, assign, sizes, _, iters, _ = hip.kmeans_lloyd(bf, hip.kmeans_init_pp(bf, L, seed=42),
                                                            max_iterations=10)
    bf.close()
    assert sizes.sum() == n and iters >= 1
    order = np.argsort(assign, kind="stable").astype(np.uint32)
    leaf_off = np.zeros(L + 1, np.uint32)
    leaf_off[1:] = np.cumsum(sizes)
    resid = rows[order] - centers[assign[order]]
    sub = np.ascontiguousarray(resid[:: n // 131072])
    rbf = hip.bf_create(sub, sub.shape[0], dim, stride, hip.SQUARED_L2)
    dsub = dim // S
    cb = np.stack([hip.kmeans_lloyd(rbf, hip.kmeans_init_pp(rbf, 16, seed=42 + s, col_offset=s * dsub,
                                                            sub_dim=dsub),
                                    max_iterations=10, col_offset=s * dsub)[0] for s in range(S)])
    rbf.close()
    codes = hip.encode(cb, np.ascontiguousarray(resid), stride=stride)
    assert np.array_equal(codes[::9973], orc.encode_many(cb, resid[::9973]))
    index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=centers,
                           leaf_offsets=leaf_off, leaf_ids=order, codebook=cb, codes=codes,
                           use_residuals=True, partitions_to_search=P, pre_reorder_multiplier=m / k)
    q, _ = synth.clustered_f32(256, dim, 8, n_clusters=1000)
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = P, m
    idx, dist, cnt, (tok, tokd, ci, cd, cc) = index.search_batched(q, k, o, stages=True)
    assert np.all(cnt == k) and np.all(np.diff(dist, axis=1) >= 0)
    oix = orc.TxhIndex(data, stride, dim, centers, leaf_off, order, cb, codes, use_residuals=True,
                       partitions_to_search=P, pre_reorder_multiplier=m / k)
    for i in range(0, 256, 16):
        H.check_txh_query(oix, q[i], k, idx[i, :cnt[i]], dist[i, :cnt[i]], tok[i], tokd[i],
                          ci[i, :cc[i]], cd[i, :cc[i]], what="q%d" % i)
    a, b, _ = index.search_batched(q[:7], k, o)      # batch-split invariance
    assert np.array_equal(b.view(np.uint32), dist[:7].view(np.uint32))


@pytest.mark.parametrize("measure", [hip.SQUARED_L2, hip.DOT_PRODUCT])
def test_scann_partitioned_1m_all_leaves_equals_brute_force(big, measure):
    """SearchMode::Partitioned at full size: with every leaf searched the scanned stream is the whole
    database, so the rows must equal brute force over the same data -- the same distances bit for bit
    (same single-pair arithmetic), the same indices up to distance ties; a subset of the leaves can
    only be worse.  Also checks the loader on a 0.5 GB index file."""
    L = 64
    bf = hip.bf_create(big["data"], N, DIM, big["stride"], measure)
    sq = bf if measure == hip.SQUARED_L2 else hip.bf_create(big["data"], N, DIM, big["stride"], hip.SQUARED_L2)
    centers, assign = hip.kmeans_lloyd(sq, hip.kmeans_init_pp(sq, L, seed=42), max_iterations=3)[:2]
    order = np.argsort(assign, kind="stable").astype(np.uint32)
    off = np.zeros(L + 1, np.uint32)
    off[1:] = np.cumsum(np.bincount(assign, minlength=L))
    kw = dict(data=big["data"], n_rows=N, dim=DIM, stride=big["stride"], centers=centers, leaf_offsets=off,
              leaf_ids=order, codebook=None, codes=None, partitions_to_search=L, distance_measure=measure)
    index = hip.txh_create(**kw)
    q = big["q"][:6]
    idx, dist, cnt = index.search_batched(q, K)
    bi, bd, bc = bf.search_batched(q, K)
    assert np.all(cnt == K) and np.all(bc == K)
    assert np.array_equal(dist.view(np.uint32), bd.view(np.uint32))
    for i in range(q.shape[0]):
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], bi[i], bd[i], what="partitioned(all leaves) q%d" % i)
    o = hip.default_opts()
    o.partitions_to_search = 8
    idx8, dist8, cnt8 = index.search_batched(q, K, o)
    assert np.all(dist8 >= dist) and np.all(np.diff(dist8, axis=1) >= 0)
    if measure == hip.SQUARED_L2:     # index file round trip at full size
        import os
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "part.scannidx")
            hip.txh_write_file(p, **kw)
            loaded = hip.load_file(p)
            li, ld, lc = loaded.search_batched(q, K)
            assert np.array_equal(li, idx) and np.array_equal(ld.view(np.uint32), dist.view(np.uint32))


# ---- BASELINE.json configs[3] and the shard shape of configs[4] (untested configs) --------------
def _build_big_txh(n, dim, L, S, kmeans_iters=6):
    """Clustered synthetic set (mixture of 1000 Gaussians) generated with torch on the GPU
    (harness plumbing: numpy would need minutes at this size), index built by the LIBRARY's GPU k-means
    (partitioner on all rows, per-subspace codebooks on a residual sample, residual encode).  The
    dataset is stored in leaf order, so datapoint index == CSR row and leaf_ids is the identity."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    cen = torch.rand((1000, dim), generator=g, device=dev)
    sig = 0.1 * (1.0 / 6.0) ** 0.5
    X = cen[torch.randint(0, 1000, (n,), generator=g, device=dev)] + sig * torch.randn((n, dim), generator=g, device=dev)
    Q = cen[torch.randint(0, 1000, (256,), generator=g, device=dev)] + sig * torch.randn((256, dim), generator=g,
                                                                                        device=dev)
    stride = hip.compute_stride(dim)
    assert stride == dim
    x_np = X.cpu().numpy()
    bf = hip.bf_create(x_np, n, dim, stride, hip.SQUARED_L2)
    centers, assign, sizes, _, iters, _ = hip.kmeans_lloyd(bf, hip.kmeans_init_pp(bf, L, seed=42),
                                                            max_iterations=kmeans_iters)
    bf.close()
    del x_np
    assert sizes.sum() == n and iters >= 1
    a_t = torch.from_numpy(assign.astype(np.int64)).to(dev)
    order = torch.argsort(a_t, stable=True)
    X = X[order]                                         # the dataset, in leaf order
    leaf_of_row = a_t[order]
    leaf_off = np.zeros(L + 1, np.uint32)
    leaf_off[1:] = np.cumsum(sizes)
    c_t = torch.from_numpy(centers).to(dev)
    ns = 131072
    pick = torch.randperm(n, generator=g, device=dev)[:ns]
    sub = np.ascontiguousarray((X[pick] - c_t[leaf_of_row[pick]]).cpu().numpy())
    rbf = hip.bf_create(sub, ns, dim, stride, hip.SQUARED_L2)
    dsub = dim // S
    cb = np.stack([hip.kmeans_lloyd(rbf, hip.kmeans_init_pp(rbf, 16, seed=42 + s, col_offset=s * dsub, sub_dim=dsub),
                                    max_iterations=8, col_offset=s * dsub)[0] for s in range(S)])
    rbf.close()
    data = X.cpu().numpy()
    lor = leaf_of_row.cpu().numpy().astype(np.uint32)
    del X, a_t, order, leaf_of_row
    torch.cuda.empty_cache()
    codes = hip.encode(cb, data, stride=stride, centers=centers, leaf_of_row=lor)
    sel = np.arange(0, n, 99991)
    assert np.array_equal(codes[sel], orc.encode_many(cb, data[sel] - centers[lor[sel]]))
    ids = np.arange(n, dtype=np.uint32)
    index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=centers, leaf_offsets=leaf_off,
                           leaf_ids=ids, codebook=cb, codes=codes, use_residuals=True, partitions_to_search=10,
                           pre_reorder_multiplier=3.0)
    return dict(index=index, data=data, stride=stride, dim=dim, centers=centers, leaf_off=leaf_off, ids=ids, cb=cb,
                codes=codes, q=Q.cpu().numpy())


def _check_big_txh(b, k, P, m, oracle_rows):
    """Stage-by-stage oracle check on a few queries, size-independent properties on all of them."""
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = P, m
    q = b["q"]
    idx, dist, cnt, (tok, tokd, ci, cd, cc) = b["index"].search_batched(q, k, o, stages=True)
    assert np.all(cnt == k) and np.all(np.diff(dist, axis=1) >= 0)
    oix = orc.TxhIndex(b["data"], b["stride"], b["dim"], b["centers"], b["leaf_off"], b["ids"], b["cb"], b["codes"],
                       use_residuals=True, partitions_to_search=P, pre_reorder_multiplier=m / k)
    for i in oracle_rows:
        H.check_txh_query(oix, q[i], k, idx[i, :cnt[i]], dist[i, :cnt[i]], tok[i], tokd[i], ci[i, :cc[i]],
                          cd[i, :cc[i]], what="P=%d m=%d q%d" % (P, m, i))
    # the returned distances are the reference's exact arithmetic on the returned rows, bit for bit
    for i in range(0, q.shape[0], 5):
        want = orc.one_to_many(q[i], b["data"][idx[i]].ravel(), b["dim"], k, hip.SQUARED_L2)
        assert np.array_equal(want.view(np.uint32), dist[i].view(np.uint32))
        assert len(set(idx[i].tolist())) == k
    # batch-split invariance (rows do not depend on the batch they travel in) and idempotence
    parts = [b["index"].search_batched(q[a:a + 37], k, o) for a in range(0, q.shape[0], 37)]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), idx)
    assert np.array_equal(np.concatenate([p[1] for p in parts]).view(np.uint32), dist.view(np.uint32))
    return idx


def test_txh_10m_gpu_build_and_oracle():
    """BASELINE configs[3]: Tree-X-Hybrid, KMeansTree 1000 leaves + AH LUT16 (32 x 16) + exact re-rank,
    10M x 128 -- including pre_reorder_k = 8192, the library's hard cap."""
    b = _build_big_txh(10_000_000, 128, 1000, 32)
    k = 10
    got = {}
    for P, m, rows in ((10, 1000, range(0, 256, 32)), (10, 8192, range(0, 256, 64)), (50, 1000, range(0, 256, 64)),
                       (50, 8192, (0, 128))):
        got[(P, m)] = _check_big_txh(b, k, P, m, rows)
    # recall against exact brute force is monotone in m and in P (nested candidate streams)
    bf = hip.bf_create(b["data"], 10_000_000, 128, b["stride"], hip.SQUARED_L2)
    ti, _, _ = bf.search_batched(b["q"], k)
    bf.close()
    rec = {pm: H.recall_at_k(g, ti, k) for pm, g in got.items()}
    assert rec[(10, 8192)] >= rec[(10, 1000)] - 1e-9 and rec[(50, 8192)] >= rec[(50, 1000)] - 1e-9
    assert rec[(50, 8192)] >= rec[(10, 8192)] - 1e-9 and rec[(10, 8192)] > 0.9, rec
    # one more than the cap is refused, not truncated
    o = hip.default_opts()
    o.pre_reorder_k = 8193
    with pytest.raises(hip.ScannError) as e:
        b["index"].search_batched(b["q"][:2], k, o)
    assert e.value.code == hip.UNIMPLEMENTED
    # A batch of 4096 queries gives every leaf ~40 (query, leaf) pairs: the DEFAULT heuristic now takes the integer-MFMA
    # prefilter (32-pair tiles, sparse instruction) on this TREE index; stage by stage against the oracle.
    rng = np.random.default_rng(5)
    pick = rng.integers(0, 10_000_000, 4096)
    q4 = b["data"][pick] + np.float32(0.02) * rng.standard_normal((4096, 128)).astype(np.float32)
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = 10, 500
    b["index"].enable_timing(True)
    idx, dist, cnt, (tok, tokd, ci, cd, cc) = b["index"].search_batched(q4, k, o, stages=True)
    assert b["index"].last_kernel_ms()[1] == "adc_smfmac_kernel"
    b["index"].enable_timing(False)
    oix = orc.TxhIndex(b["data"], b["stride"], b["dim"], b["centers"], b["leaf_off"], b["ids"], b["cb"], b["codes"],
                       use_residuals=True, partitions_to_search=10, pre_reorder_multiplier=50.0)
    for i in (0, 777, 2048, 4095):
        H.check_txh_query(oix, q4[i], k, idx[i, :cnt[i]], dist[i, :cnt[i]], tok[i], tokd[i], ci[i, :cc[i]], cd[i, :cc[i]],
                          what="4096-query batch q%d" % i)
    for i in range(0, 4096, 97):
        want = orc.one_to_many(q4[i], b["data"][idx[i]].ravel(), 128, k, hip.SQUARED_L2)
        assert np.array_equal(want.view(np.uint32), dist[i].view(np.uint32))


def test_txh_c5_shard_shape():
    """The per-GPU shard of BASELINE configs[4] (100M x 96 over 8 GPUs): 12.5M x 96, S = 24, 1250 leaves."""
    b = _build_big_txh(12_500_000, 96, 1250, 24)
    k = 10
    got = {}
    for P, m, rows in ((10, 1000, range(0, 256, 32)), (10, 8192, range(0, 256, 64))):
        got[(P, m)] = _check_big_txh(b, k, P, m, rows)
    bf = hip.bf_create(b["data"], 12_500_000, 96, b["stride"], hip.SQUARED_L2)
    ti, _, _ = bf.search_batched(b["q"], k)
    bf.close()
    r1, r2 = H.recall_at_k(got[(10, 1000)], ti, k), H.recall_at_k(got[(10, 8192)], ti, k)
    assert r2 >= r1 - 1e-9 and r2 > 0.9, (r1, r2)
