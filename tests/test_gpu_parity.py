"""GPU parity tests proper: every row of SURVEY.md section 8a through the C ABI
(libscann_hip.so) against the CPU oracle on the same seeded inputs.

Bars: approximate (LUT-sum) distances, centroid distances, LUTs, codes and exact
re-rank / brute-force distances are BIT-EXACT (the kernels reproduce the reference's
operation order); result indices are equal up to exact distance ties."""
import threading

import numpy as np
import pytest

from oracle import pyoracle as orc
from scann_rust_amd import hip, synth, trainer
from tests import helpers as H

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ---- a2: TreePartitioner::partition ----------------------------------------------------
@pytest.mark.parametrize("L,dim,P", [(16, 128, 4), (37, 96, 37), (100, 32, 10), (1000, 128, 50),
                                     (5, 8, 3)])
def test_partition_bit_exact(L, dim, P):
    S = 8
    rows, data, stride, ix, oix, kw = H.make_txh_case(max(4 * L, 512), dim, L, S, seed=3, P=P,
                                                      kmeans_iters=2, pq_iters=2)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(33, dim, 123)
    tok, dist, cnt = hip.txh_partition(index, q, P)
    for i in range(q.shape[0]):
        ot, od = orc.partition(ix["centers"], q[i], P)
        assert cnt[i] == ot.size
        assert np.array_equal(bits(dist[i, :ot.size]), bits(od))
        assert np.array_equal(tok[i, :ot.size], ot)


# ---- a3/a4: residual + LookupTable::from_query ----------------------------------------------
@pytest.mark.parametrize("dim,S", [(128, 32), (96, 24), (64, 8), (128, 16)])
def test_lut_from_query_bit_exact(dim, S):
    rows, data, stride, ix, oix, kw = H.make_txh_case(2048, dim, 8, S, seed=5, kmeans_iters=2,
                                                      pq_iters=3)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(17, dim, 9)
    lut = hip.lut_from_query(index, q, S, 16)
    for i in range(q.shape[0]):
        assert np.array_equal(bits(lut[i]), bits(orc.lut_from_query(ix["codebook"], q[i])))
    leaves = (np.arange(17) % 8).astype(np.uint32)
    lut = hip.lut_from_query(index, q, S, 16, leaf_for_query=leaves)
    for i in range(q.shape[0]):
        res = q[i] - ix["centers"][leaves[i]]
        assert np.array_equal(bits(lut[i]), bits(orc.lut_from_query(ix["codebook"], res)))


# ---- a5/a13: LookupTable::compute_distance over packed 4-bit codes ------------------------------
@pytest.mark.parametrize("dim,S", [(128, 32), (96, 24), (64, 8), (128, 64), (96, 48)])
def test_adc_distances_bit_exact(dim, S):
    rows, data, stride, ix, kw = H.make_ah_case(3000, dim, S, seed=11, pq_iters=3)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(5, dim, 77)
    luts = np.stack([orc.lut_from_query(ix["codebook"], qq) for qq in q])
    got = hip.adc_distances(index, luts)
    for i in range(q.shape[0]):
        want = np.array([orc.lut_distance(luts[i], c) for c in ix["codes"]], np.float32)
        assert np.array_equal(bits(got[i]), bits(want))


# ---- a13: Lut16SimdTables::compute_distances_batch (u8 tables) -------------------------------------
@pytest.mark.parametrize("S,n", [(2, 4), (16, 1000), (32, 5000), (7, 333)])
def test_lut16_u8_batch_bit_exact(S, n):
    rng = np.random.default_rng(S * 1000 + n)
    tables = rng.random((S, 16), dtype=np.float32) * 3.0
    lut8, bias, mult = orc.lut16_quantize(tables)
    codes = rng.integers(0, 16, (n, S), dtype=np.uint8)
    packed = orc.pack4(codes)
    got = hip.lut16_distances_batch(packed, lut8, S, n, bias, mult)
    want = orc.lut16_distances_batch(packed, lut8, S, n, bias, mult)
    assert np.array_equal(bits(got), bits(want))


def test_lut16_reference_vectors():  # src/simd/tests.rs:237-264, src/hashes/lut16_simd.rs:354-375
    lut = np.zeros((2, 16), np.uint8)
    lut[0] = np.arange(16)
    lut[1] = 15 - np.arange(16)
    got = hip.lut16_distances_batch(np.array([0x00, 0x11, 0x0F, 0xF0], np.uint8), lut, 2, 4, 0.0, 1.0)
    assert list(got) == [15.0, 15.0, 30.0, 0.0]


# ---- a13: Lut16SimdTables::from_float_tables on the device (hashes/lut16_simd.rs:39-90) ----------------
@pytest.mark.parametrize("S,kind", [(1, "rand"), (2, "rand"), (16, "rand"), (32, "rand"), (64, "wide"), (7, "neg"),
                                    (3, "const"), (4, "tiny"), (5, "nan"), (32, "lut")])
def test_lut16_quantize_bit_exact(S, kind):
    rng = np.random.default_rng(S)
    t = rng.random((S, 16), dtype=np.float32) * 3.0
    if kind == "wide":
        t = (t * np.float32(1e6)).astype(np.float32)
    if kind == "neg":
        t = (t - np.float32(1.5)).astype(np.float32)
    if kind == "const":                       # degenerate range: scale = multiplier = 1, bias = the value
        t[:] = np.float32(2.5)
    if kind == "tiny":                        # range < 1e-10 but not zero
        t = (np.float32(1.0) + t * np.float32(1e-12)).astype(np.float32)
    if kind == "nan":                         # f32::min / f32::max ignore a NaN operand; NaN as u8 = 0
        t[1, 3] = np.nan
    if kind == "lut":                         # a real query table
        cb = trainer.train_codebook(synth.uniform_f32(2000, 128, 3), 32, 16, iters=2, seed=1)
        t = orc.lut_from_query(cb, synth.uniform_f32(1, 128, 9)[0])
    g8, gb, gm = hip.lut16_quantize(t)
    o8, ob, om = orc.lut16_quantize(t)
    assert np.array_equal(g8, o8)
    assert np.float32(gb).view(np.uint32) == np.float32(ob).view(np.uint32)
    assert np.float32(gm).view(np.uint32) == np.float32(om).view(np.uint32)


def test_lut16_quantize_reference_vectors_end_to_end():
    """hashes/lut16_simd.rs:306-375 run on the GPU end to end: quantise -> batch kernel -> dequantise."""
    t0 = np.arange(16, dtype=np.float32)
    t1 = (15 - np.arange(16)).astype(np.float32)
    lut8, bias, mult = hip.lut16_quantize(np.stack([t0, t1]))
    # test_quantization_roundtrip (:306-330): codes [0,0] -> 15, [15,15] -> 15, [5,10] -> 10, tol 0.1
    got = hip.lut16_distances_batch(orc.pack4(np.array([[0, 0], [15, 15], [5, 10]], np.uint8)), lut8, 2, 3, bias, mult)
    assert np.all(np.abs(got - np.array([15.0, 15.0, 10.0], np.float32)) < 0.1)
    # test_two_subspace_packed (:354-375): 0x00, 0x55, 0x0F, 0xF0 -> 15, 15, 30, 0, tol 0.5
    got = hip.lut16_distances_batch(np.array([0x00, 0x55, 0x0F, 0xF0], np.uint8), lut8, 2, 4, bias, mult)
    assert np.all(np.abs(got - np.array([15.0, 15.0, 30.0, 0.0], np.float32)) < 0.5)
    # test_batch_computation (:332-352): one subspace, 0x00 0x05 0x0A 0x0F -> 0, 5, 10, 15, tol 0.1
    l1, b1, m1 = hip.lut16_quantize(t0[None])
    got = hip.lut16_distances_batch(np.array([0x00, 0x05, 0x0A, 0x0F], np.uint8), l1, 1, 4, b1, m1)
    assert np.all(np.abs(got - np.array([0.0, 5.0, 10.0, 15.0], np.float32)) < 0.1)
    # S == 0 (:42-49)
    e8, eb, em = hip.lut16_quantize(np.zeros((0, 16), np.float32))
    assert e8.shape == (0, 16) and eb == 0.0 and em == 1.0


# ---- a14: Codebook::encode -------------------------------------------------------------------------
def test_encode_bit_exact():
    X = synth.uniform_f32(4000, 128, 21)
    cb = trainer.train_codebook(X, 32, 16, iters=3, seed=4)
    assert np.array_equal(hip.encode(cb, X), orc.encode_many(cb, X))
    centers = synth.uniform_f32(6, 128, 8)
    leaf = (np.arange(4000) % 6).astype(np.uint32)
    got = hip.encode(cb, X, centers=centers, leaf_of_row=leaf)
    assert np.array_equal(got, orc.encode_many(cb, X - centers[leaf]))


# ---- AsymmetricHasher::search / search_with_reordering ------------------------------------------------
@pytest.mark.parametrize("n,dim,S,k", [(5000, 128, 32, 10), (20000, 96, 24, 7), (300, 64, 8, 50)])
def test_ah_search(n, dim, S, k):
    rows, data, stride, ix, kw = H.make_ah_case(n, dim, S, seed=2, pq_iters=3)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(9, dim, 123)
    o = hip.default_opts()
    o.exact_reorder = 0
    idx, dist, cnt = index.search_batched(q, k, o)
    for i in range(q.shape[0]):
        oi, od = orc.ah_search(ix["codebook"], ix["codes"], q[i], k)
        assert cnt[i] == oi.size
        H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], oi, od, what="ah q%d" % i)


@pytest.mark.parametrize("n,dim,S,k,pre_k", [(5000, 128, 32, 10, 100), (20000, 96, 24, 5, 37)])
def test_ah_search_with_reordering(n, dim, S, k, pre_k):
    rows, data, stride, ix, kw = H.make_ah_case(n, dim, S, seed=6, pq_iters=3)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(9, dim, 123)
    o = hip.default_opts()
    o.pre_reorder_k = pre_k
    idx, dist, cnt = index.search_batched(q, k, o)
    for i in range(q.shape[0]):
        oi, od = orc.ah_search_with_reordering(ix["codebook"], ix["codes"], data, stride, q[i], k, pre_k)
        assert cnt[i] == oi.size
        H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], oi, od, what="ahr q%d" % i)


# ---- TreeXHybridSearcher::search (north-star path), stage by stage ----------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("dim,S", [(128, 32), (96, 24)])
@pytest.mark.parametrize("P,mult", [(1, 3.0), (4, 3.0), (16, 10.0)])
def test_txh_search_stages(seed, dim, S, P, mult):
    n, L, k, nq = 4096, 16, 10, 64
    rows, data, stride, ix, oix, kw = H.make_txh_case(n, dim, L, S, seed=seed, P=P, mult=mult,
                                                      kmeans_iters=3, pq_iters=3)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(nq, dim, 123 + seed)
    m = orc.pre_reorder_k(k, mult)
    o = hip.default_opts()
    o.partitions_to_search = P
    o.pre_reorder_k = m
    idx, dist, cnt, (tok, tokd, ci, cd, cc) = index.search_batched(q, k, o, stages=True)
    for i in range(nq):
        H.check_txh_query(oix, q[i], k, idx[i, :cnt[i]], dist[i, :cnt[i]], tok[i], tokd[i],
                          ci[i, :cc[i]], cd[i, :cc[i]], what="q%d" % i)


def test_txh_no_residuals_and_clustered():
    rows, data, stride, ix, oix, kw = H.make_txh_case(6000, 128, 24, 32, seed=9, P=6, mult=5.0,
                                                      use_residuals=False, clustered=True,
                                                      kmeans_iters=3, pq_iters=3)
    index = hip.txh_create(**kw)
    q = rows[::200] + np.float32(0.01)
    o = hip.default_opts()
    o.partitions_to_search = 6
    o.pre_reorder_k = 50
    idx, dist, cnt, (tok, tokd, ci, cd, cc) = index.search_batched(q, 10, o, stages=True)
    for i in range(q.shape[0]):   # tight clusters + global codebook: many approximate ties
        H.check_txh_query(oix, q[i], 10, idx[i, :cnt[i]], dist[i, :cnt[i]], tok[i], tokd[i],
                          ci[i, :cc[i]], cd[i, :cc[i]], what="q%d" % i)


def test_txh_ragged_and_short_results():
    """Empty leaves, leaves smaller than m, fewer candidates than k (mod.rs:360-363)."""
    dim, S = 64, 8
    rows = synth.uniform_f32(40, dim, 5)
    centers = synth.uniform_f32(6, dim, 6)
    assign = np.array([0] * 30 + [2] * 7 + [5] * 3)       # leaves 1, 3, 4 empty
    built = trainer.build_txh_index(rows, 6, S, centers=centers, assign=assign, pq_iters=2)
    data, stride = orc.to_strided(rows)
    oix = orc.TxhIndex(data, stride, dim, built["centers"], built["leaf_off"], built["leaf_ids"],
                       built["codebook"], built["codes"], partitions_to_search=3,
                       pre_reorder_multiplier=3.0)
    index = hip.txh_create(data=data, n_rows=40, dim=dim, stride=stride, centers=built["centers"],
                           leaf_offsets=built["leaf_off"], leaf_ids=built["leaf_ids"],
                           codebook=built["codebook"], codes=built["codes"],
                           partitions_to_search=3, pre_reorder_multiplier=3.0)
    q = synth.uniform_f32(12, dim, 7)
    idx, dist, cnt = index.search_batched(q, 10)
    for i in range(q.shape[0]):
        oi, od = orc.txh_search(oix, q[i], 10)
        assert cnt[i] == oi.size
        H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], oi, od, what="q%d" % i)


def test_txh_candidate_overflow_retry():
    """Adversarial order: the strided sample sees only far points, so the first-pass
    candidate buffer overflows and the host entry retries with a full-size buffer."""
    dim, S, n = 64, 8, 60000
    rows = synth.uniform_f32(n, dim, 31)
    q = synth.uniform_f32(3, dim, 32)
    cb = trainer.train_codebook(rows, S, 16, iters=3, seed=1)
    codes = trainer.encode(cb, rows)
    # order points so that every position divisible by the sample stride is far from q[0]
    lut = orc.lut_from_query(cb, q[0])
    d = np.array([orc.lut_distance(lut, c) for c in codes[:n]], np.float32)
    order = np.argsort(-d, kind="stable")
    ns = min(max(n // 16, 4096), 65536)      # sample_stride() of csrc/txh.h
    st = -(-n // ns)
    perm = np.empty(n, np.int64)
    far = list(order[: -(-n // st)])
    near = list(order[-(-n // st):])
    fi = ni = 0
    for pos in range(n):
        if pos % st == 0:
            perm[pos] = far[fi]; fi += 1
        else:
            perm[pos] = near[ni]; ni += 1
    rows2, codes2 = rows[perm], codes[perm]
    data, stride = orc.to_strided(rows2)
    index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=None,
                           leaf_offsets=None, leaf_ids=None, codebook=cb, codes=codes2,
                           use_residuals=False, partitions_to_search=1, pre_reorder_multiplier=1.0)
    o = hip.default_opts()
    o.exact_reorder = 0
    idx, dist, cnt = index.search_batched(q, 10, o)
    for i in range(3):
        oi, od = orc.ah_search(cb, codes2, q[i], 10)
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], oi, od, what="q%d" % i)


# ---- brute force -----------------------------------------------------------------------------------
@pytest.mark.parametrize("measure", [hip.DOT_PRODUCT, hip.SQUARED_L2, hip.L2])
@pytest.mark.parametrize("n,dim", [(1000, 128), (777, 96), (300, 50), (5000, 32)])
def test_bf_distances_bit_exact(measure, n, dim):
    rows = synth.uniform_f32(n, dim, 42)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, measure)
    q = synth.uniform_f32(19, dim, 123)
    got = hip.bf_distances(index, q)
    for i in range(q.shape[0]):
        want = orc.one_to_many(q[i], data, stride, n, measure)
        assert np.array_equal(bits(got[i]), bits(want)), "q%d" % i


@pytest.mark.parametrize("measure", [hip.DOT_PRODUCT, hip.SQUARED_L2])
@pytest.mark.parametrize("n,dim,k,nq", [(10000, 128, 10, 40), (30000, 64, 3, 130), (500, 96, 600, 5)])
def test_bf_search(measure, n, dim, k, nq):
    rows = synth.uniform_f32(n, dim, 42)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, measure)
    q = synth.uniform_f32(nq, dim, 123)
    idx, dist, cnt = index.search_batched(q, k)
    oi, od, oc = orc.bf_search_batched(data, n, dim, stride, measure, q, k)
    assert np.array_equal(cnt, oc)
    for i in range(nq):
        c = cnt[i]
        H.assert_topk_equal_up_to_ties(idx[i, :c], dist[i, :c], oi[i, :c], od[i, :c], what="bf q%d" % i)


def test_bf_reference_vectors():
    # src/brute_force/searcher.rs:280-376, tests/unit_tests.rs:204-259
    cube = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]], np.float32)
    data, stride = orc.to_strided(cube)
    index = hip.bf_create(data, 5, 3, stride, hip.SQUARED_L2)
    idx, dist, cnt = index.search_batched(np.array([[0, 0, 0]], np.float32), 3)
    assert cnt[0] == 3 and idx[0, 0] == 0 and abs(dist[0, 0]) < 1e-6
    idx, dist, cnt = index.search_batched(np.array([[0.5, 0.5, 0.5]], np.float32), 5)
    assert cnt[0] == 5 and np.all(np.diff(dist[0]) >= 0)
    idx, dist, cnt = index.search_batched(np.array([[0, 0, 0], [1, 1, 1]], np.float32), 2)
    assert list(cnt) == [2, 2]
    with pytest.raises(hip.ScannError) as e:      # dimension mismatch -> InvalidArgument
        index.search_batched(np.array([[1, 2]], np.float32), 5)
    assert e.value.code == hip.INVALID_ARGUMENT
    empty = hip.bf_create(np.zeros(0, np.float32), 0, 3, 16, hip.SQUARED_L2)
    idx, dist, cnt = empty.search_batched(np.array([[1, 2, 3]], np.float32), 5)
    assert cnt[0] == 0
    dot = hip.bf_create(*orc.to_strided(np.array([[1, 0], [0, 1], [1, 1]], np.float32))[:1], 3, 2, 16,
                        hip.DOT_PRODUCT)
    idx, dist, cnt = dot.search_batched(np.array([[1, 0]], np.float32), 3)
    assert list(dist[0]) == [-1.0, -1.0, 0.0] and sorted(idx[0, :2].tolist()) == [0, 2]


# ---- error behaviour of the boundary ----------------------------------------------------------------
def test_error_codes():
    rows, data, stride, ix, oix, kw = H.make_txh_case(600, 64, 4, 8, seed=1, kmeans_iters=2, pq_iters=2)
    index = hip.txh_create(**kw)
    with pytest.raises(hip.ScannError) as e:      # tree_x_hybrid/mod.rs:251-253
        index.search_batched(synth.uniform_f32(2, 32, 1), 5)
    assert e.value.code == hip.INVALID_ARGUMENT
    bad = dict(kw)
    bad["codes"] = kw["codes"][:0]
    with pytest.raises(hip.ScannError) as e:      # mod.rs:132-134 empty dataset
        hip.txh_create(**bad)
    assert e.value.code == hip.INVALID_ARGUMENT
    bad = dict(kw)
    bad["codebook"] = np.zeros((7, 16, 9), np.float32)   # 64 % 7 != 0 -> codebook.rs:154-159
    with pytest.raises(hip.ScannError) as e:
        hip.txh_create(**bad)
    assert e.value.code == hip.INVALID_ARGUMENT
    nostore = dict(kw)
    nostore["data"] = None                         # hasher.rs:194-197 "Dataset not stored"
    ns = hip.txh_create(**nostore)
    with pytest.raises(hip.ScannError) as e:
        ns.search_batched(synth.uniform_f32(2, 64, 1), 5)
    assert e.value.code == hip.FAILED_PRECONDITION
    ah = hip.txh_create(**H.make_ah_case(300, 64, 8, seed=2, pq_iters=2)[4])
    with pytest.raises(hip.ScannError) as e:      # tree_partitioner.rs:197-198
        hip.txh_partition(ah, synth.uniform_f32(2, 64, 1), 2)
    assert e.value.code == hip.FAILED_PRECONDITION


# ---- Searcher: Send + Sync (tests/stress_tests.rs:256-297) ---------------------------------------------
def test_concurrent_queries():
    rows, data, stride, ix, oix, kw = H.make_txh_case(4096, 64, 16, 8, seed=4, P=4, kmeans_iters=2,
                                                      pq_iters=2)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(40, 64, 99)
    want = [orc.txh_search(oix, q[i], 10) for i in range(40)]
    errs = []

    def worker(t):
        try:
            for i in range(t, 40, 4):
                idx, dist, cnt = index.search_batched(q[i:i + 1], 10)
                assert cnt[0] == want[i][0].size
                H.assert_topk_equal_up_to_ties(idx[0, :cnt[0]], dist[0, :cnt[0]], *want[i])
        except Exception as ex:  # noqa
            errs.append(ex)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


@pytest.mark.parametrize("kind", ["txh", "bf_dot", "bf_l2", "partitioned"])
def test_concurrent_queries_use_separate_slots(kind):
    """8 caller threads on one handle: concurrent host-side searches run on separate search slots
    (stream + workspace each) and must return exactly what the same calls return one at a time,
    for single queries and for batches of different sizes."""
    rows, data, stride, ix, oix, kw = H.make_txh_case(5000, 64, 12, 8, seed=14, P=3, kmeans_iters=2,
                                                      pq_iters=2)
    if kind == "txh":
        index = hip.txh_create(**kw)
    elif kind == "partitioned":
        index = hip.txh_create(**dict(kw, codebook=None, codes=None))
    else:
        index = hip.bf_create(data, 5000, 64, stride, hip.DOT_PRODUCT if kind == "bf_dot" else hip.L2)
    q = synth.uniform_f32(240, 64, 77)
    jobs = [(i, i + 1) for i in range(0, 96)] + [(96 + 9 * j, 96 + 9 * j + 9) for j in range(16)]
    want = [index.search_batched(q[a:b], 10) for a, b in jobs]
    errs = []

    def worker(t):
        try:
            for rep in range(3):
                for j in range(t, len(jobs), 8):
                    a, b = jobs[j]
                    idx, dist, cnt = index.search_batched(q[a:b], 10)
                    assert np.array_equal(cnt, want[j][2]) and np.array_equal(idx, want[j][0])
                    assert np.array_equal(bits(dist), bits(want[j][1]))
        except Exception as ex:  # noqa
            errs.append(ex)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:2]


# ---- multi-GPU stages on one GPU: two leaf shards -> local stage x2 -> merge ------------------------------
@pytest.mark.parametrize("world,k,m", [(2, 10, 60), (3, 10, 60), (2, 100, 400)])
def test_sharded_local_plus_merge_equals_single_index(world, k, m):
    import ctypes as C
    import torch
    from scann_rust_amd import sharding
    n, dim, L, S, P, nq = 6000, 128, 24, 32, 8, 50
    rows, data, stride, ix, oix, kw = H.make_txh_case(n, dim, L, S, seed=12, P=P, mult=m / k,
                                                      kmeans_iters=3, pq_iters=3)
    full = hip.txh_create(**kw)
    q = synth.uniform_f32(nq, dim, 55)
    o = hip.default_opts()
    o.partitions_to_search = P
    o.pre_reorder_k = m
    want_idx, want_dist, want_cnt = full.search_batched(q, k, o)

    Lh = hip.load()
    dev = torch.device("cuda", 0)
    sptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    qd = torch.from_numpy(q).to(dev)
    g_keys = torch.zeros((world, nq, m), dtype=torch.int64, device=dev)
    g_idx = torch.zeros((world, nq, m), dtype=torch.int32, device=dev)
    g_ex = torch.zeros((world, nq, m), dtype=torch.float32, device=dev)
    g_cnt = torch.zeros((world, nq), dtype=torch.int32, device=dev)
    shards = []
    for r in range(world):
        skw = sharding.shard_txh_index(ix, data, stride, r, world)
        sh = hip.txh_create(partitions_to_search=P, pre_reorder_multiplier=m / k, **skw)
        shards.append(sh)
        hip.check(Lh.scann_hip_txh_search_local_device(
            sh.h, C.c_void_p(qd.data_ptr()), nq, dim, k, C.byref(o),
            C.c_void_p(g_keys[r].data_ptr()), C.c_void_p(g_idx[r].data_ptr()),
            C.c_void_p(g_ex[r].data_ptr()), C.c_void_p(g_cnt[r].data_ptr()), sptr))
        hip.check(Lh.scann_hip_index_last_device_status(sh.h, sptr))
    out_idx = torch.zeros((nq, k), dtype=torch.int32, device=dev)
    out_dist = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    out_cnt = torch.zeros((nq,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    hip.check(Lh.scann_hip_txh_merge_device(hip.context(0), world, nq, m, m, k, 0,
                                            C.c_void_p(g_keys.data_ptr()), C.c_void_p(g_idx.data_ptr()),
                                            C.c_void_p(g_ex.data_ptr()), C.c_void_p(g_cnt.data_ptr()),
                                            C.c_void_p(out_idx.data_ptr()), C.c_void_p(out_dist.data_ptr()),
                                            C.c_void_p(out_cnt.data_ptr()), C.c_void_p(status.data_ptr()),
                                            sptr))
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    gi = out_idx.cpu().numpy().view(np.uint32)
    gd = out_dist.cpu().numpy()
    gc = out_cnt.cpu().numpy()
    assert np.array_equal(gc.astype(np.uint32), want_cnt)
    assert np.array_equal(gd.view(np.uint32), want_dist.view(np.uint32))
    assert np.array_equal(gi, want_idx)
    # and the numpy model of the merge agrees with the kernel
    mi, md, mc = sharding.merge_reference(g_keys.cpu().numpy().view(np.uint64),
                                          g_idx.cpu().numpy().view(np.uint32), g_ex.cpu().numpy(),
                                          g_cnt.cpu().numpy(), m, k)
    assert np.array_equal(mi, gi) and np.array_equal(md.view(np.uint32), gd.view(np.uint32))


def test_sharded_local_stage_prunes_by_prefix_dominance(monkeypatch):
    """Long local lists (m > 512, int8 row copy): the local stage computes exact distances only for the first 256
    entries of a rank's key-ordered list and for the entries whose int8 lower bound does not exceed the k-th smallest
    upper bound among those 256 (ShortArgs::local_head); the rest travel as +inf.  The merged rows must equal the
    single index's, the distances that ARE filled in must equal the unpruned run's bit for bit, and most entries must
    in fact have been pruned."""
    import ctypes as C
    import torch
    from scann_rust_amd import sharding
    monkeypatch.setenv("SCANN_HIP_RERANK_I8", "2")   # (shards of 45000 rows: below the default size for the int8 copy)
    n, dim, L, S, P, nq, k, m, world = 90000, 64, 30, 16, 6, 40, 10, 1500, 2
    rows, data, stride, ix, oix, kw = H.make_txh_case(n, dim, L, S, seed=14, P=P, mult=m / k, kmeans_iters=3, pq_iters=3)
    full = hip.txh_create(**kw)
    q = synth.uniform_f32(nq, dim, 56)
    q[3] = rows[77]
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = P, m
    want_idx, want_dist, want_cnt = full.search_batched(q, k, o)
    Lh = hip.load()
    dev = torch.device("cuda", 0)
    sptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    qd = torch.from_numpy(q).to(dev)
    shards = [hip.txh_create(partitions_to_search=P, pre_reorder_multiplier=m / k,
                             **sharding.shard_txh_index(ix, data, stride, r, world)) for r in range(world)]
    runs = {}
    for prune in ("1", "0"):
        monkeypatch.setenv("SCANN_HIP_LOCAL_PRUNE", prune)
        g_keys = torch.zeros((world, nq, m), dtype=torch.int64, device=dev)
        g_idx = torch.zeros((world, nq, m), dtype=torch.int32, device=dev)
        g_ex = torch.zeros((world, nq, m), dtype=torch.float32, device=dev)
        g_cnt = torch.zeros((world, nq), dtype=torch.int32, device=dev)
        for r in range(world):
            hip.check(Lh.scann_hip_txh_search_local_device(
                shards[r].h, C.c_void_p(qd.data_ptr()), nq, dim, k, C.byref(o),
                C.c_void_p(g_keys[r].data_ptr()), C.c_void_p(g_idx[r].data_ptr()),
                C.c_void_p(g_ex[r].data_ptr()), C.c_void_p(g_cnt[r].data_ptr()), sptr))
            hip.check(Lh.scann_hip_index_last_device_status(shards[r].h, sptr))
        out_idx = torch.zeros((nq, k), dtype=torch.int32, device=dev)
        out_dist = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        out_cnt = torch.zeros((nq,), dtype=torch.int32, device=dev)
        status = torch.zeros((1,), dtype=torch.int32, device=dev)
        hip.check(Lh.scann_hip_txh_merge_device(hip.context(0), world, nq, m, m, k, 0,
                                                C.c_void_p(g_keys.data_ptr()), C.c_void_p(g_idx.data_ptr()),
                                                C.c_void_p(g_ex.data_ptr()), C.c_void_p(g_cnt.data_ptr()),
                                                C.c_void_p(out_idx.data_ptr()), C.c_void_p(out_dist.data_ptr()),
                                                C.c_void_p(out_cnt.data_ptr()), C.c_void_p(status.data_ptr()), sptr))
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        assert np.array_equal(out_cnt.cpu().numpy().astype(np.uint32), want_cnt), prune
        assert np.array_equal(out_dist.cpu().numpy().view(np.uint32), want_dist.view(np.uint32)), prune
        assert np.array_equal(out_idx.cpu().numpy().view(np.uint32), want_idx), prune
        runs[prune] = (g_keys.cpu().numpy(), g_idx.cpu().numpy(), g_ex.cpu().numpy(), g_cnt.cpu().numpy())
    pk, pi, pe, pc = runs["1"]
    fk, fi, fe, fc = runs["0"]
    assert np.array_equal(pk, fk) and np.array_equal(pi, fi) and np.array_equal(pc, fc)
    valid = np.arange(m)[None, None, :] < pc[:, :, None]
    assert not np.isinf(fe[valid]).any()
    pruned = np.isinf(pe) & valid
    assert pruned.sum() > 0.5 * valid.sum(), (int(pruned.sum()), int(valid.sum()))
    assert not pruned[:, :, :256].any()                       # the head of every list is always re-ranked
    assert np.array_equal(pe[valid & ~pruned].view(np.uint32), fe[valid & ~pruned].view(np.uint32))


# ---- index-build helper: TreePartitioner::partition(x, 1) for every row -----------------------------------
@pytest.mark.parametrize("n,dim,k", [(3000, 128, 100), (1000, 96, 37), (700, 50, 5), (500, 7, 16), (64, 32, 1)])
def test_assign_nearest_bit_exact(n, dim, k):
    rows = synth.uniform_f32(n, dim, 17)
    data, stride = orc.to_strided(rows)
    centers = synth.uniform_f32(k, dim, 18)
    index = hip.bf_create(data, n, dim, stride, hip.SQUARED_L2)
    asg, dist = hip.bf_assign_nearest(index, centers)
    for i in range(0, n, 7):
        tok, d = orc.partition(centers, rows[i], 1)
        assert asg[i] == tok[0]
        assert np.float32(dist[i]) == d[0]
    # ties -> lowest centre index (stable sort, tree_partitioner.rs:212)
    dup = np.concatenate([centers, centers[:1]])
    asg2, _ = hip.bf_assign_nearest(index, dup)
    assert np.array_equal(asg2, asg)


@pytest.mark.parametrize("frac,k", [(0.5, 10), (0.05, 10), (0.002, 5)])
def test_txh_allow_bitmap_filter(frac, k):
    """search_with_filter(.., Some(allow-list)) (tree_x_hybrid/mod.rs:245-250, 327-332):
    disallowed datapoints are skipped before scoring; every stage still matches the oracle.
    frac=0.002 leaves fewer allowed points than pre_reorder_k (and for some queries than k)."""
    n, dim, L, S = 6000, 64, 24, 16
    rows, data, stride, ix, oix, kw = H.make_txh_case(n, dim, L, S, seed=91, P=6, mult=8.0)
    rng = np.random.default_rng(17)
    allowed = np.flatnonzero(rng.random(n) < frac)
    bits = hip.allow_bitmap(n, allowed)
    oix.allow = bits
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(33, dim, 92)
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = 6, int(np.float32(k) * np.float32(8.0))
    idx, dist, cnt, (tok, tokd, ci, cd, cc) = index.search_batched(q, k, opts=o, stages=True,
                                                                   allow=bits)
    aset = set(allowed.tolist())
    for i in range(q.shape[0]):
        assert set(idx[i, :cnt[i]].tolist()) <= aset
        H.check_txh_query(oix, q[i], k, idx[i, :cnt[i]], dist[i, :cnt[i]], tok[i], tokd[i],
                          ci[i, :cc[i]], cd[i, :cc[i]], what="filter q%d" % i)
    # no filter afterwards: the bitmap does not stick to the handle
    oix.allow = None
    idx2, dist2, cnt2 = index.search_batched(q[:4], k)
    for i in range(4):
        oi, od = orc.txh_search(oix, q[i], k)
        H.assert_topk_equal_up_to_ties(idx2[i, :cnt2[i]], dist2[i, :cnt2[i]], oi, od)


def test_ah_allow_bitmap_filter():
    """Same filter on the flat AsymmetricHasher index (no partitions)."""
    n, dim, S = 5000, 32, 8
    rows, data, stride, ix, kw = H.make_ah_case(n, dim, S, seed=93)
    index = hip.txh_create(**kw)
    allowed = np.arange(0, n, 7)
    bits = hip.allow_bitmap(n, allowed)
    q = synth.uniform_f32(9, dim, 94)
    o = hip.default_opts()
    o.exact_reorder, o.pre_reorder_k = 1, 50
    idx, dist, cnt = index.search_batched(q, 10, opts=o, allow=bits)
    sub = np.ascontiguousarray(ix["codes"][allowed])
    for i in range(q.shape[0]):
        assert cnt[i] == 10 and np.all(idx[i] % 7 == 0)
        # oracle on the allowed subset == filtered search (skipped rows are never pushed)
        oi, od = orc.ah_search_with_reordering(ix["codebook"], sub,
                                               np.ascontiguousarray(data.reshape(n, stride)[allowed]),
                                               stride, q[i], 10, 50)
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], allowed[oi].astype(np.uint32), od)
    # a bitmap shorter than the dataset: indices >= its capacity are not allowed (allowlist.rs:97-100)
    idx, dist, cnt = index.search_batched(q, 10, opts=o, allow=bits[:32])
    assert np.all(cnt == 10) and np.all(idx < 2048) and np.all(idx % 7 == 0)


# ---- 8-bit codes: 16 < num_codes <= 256 (the reference's default 256 x 8 codebooks) --------------
@pytest.mark.parametrize("dim,S,K", [(64, 8, 256), (32, 4, 64), (128, 16, 256), (64, 8, 100)])
def test_adc_distances_bit_exact_byte_codes(dim, S, K):
    rows, data, stride, ix, kw = H.make_ah_case(3000, dim, S, seed=12, K=K, pq_iters=2)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(5, dim, 78)
    luts = np.stack([orc.lut_from_query(ix["codebook"], qq) for qq in q])
    assert np.array_equal(bits(hip.lut_from_query(index, q, S, K)), bits(luts))
    got = hip.adc_distances(index, luts)
    for i in range(q.shape[0]):
        want = np.array([orc.lut_distance(luts[i], c) for c in ix["codes"]], np.float32)
        assert np.array_equal(bits(got[i]), bits(want))


@pytest.mark.parametrize("dim,S,K,P", [(64, 8, 256, 4), (128, 16, 256, 3), (32, 4, 64, 16), (64, 8, 17, 5)])
def test_txh_search_stages_byte_codes(dim, S, K, P):
    """TreeXHybridConfig::default() hashes with 256 codes x 8 subspaces (tree_x_hybrid/mod.rs:37-48,
    hashes/hasher.rs:36-46): same stage-by-stage parity as the LUT16 configuration."""
    n, L, k, nq, mult = 5000, 16, 10, 37, 4.0
    rows, data, stride, ix, oix, kw = H.make_txh_case(n, dim, L, S, seed=21, K=K, P=P, mult=mult,
                                                      kmeans_iters=3, pq_iters=2)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(nq, dim, 321)
    o = hip.default_opts()
    o.partitions_to_search = P
    o.pre_reorder_k = orc.pre_reorder_k(k, mult)
    idx, dist, cnt, (tok, tokd, ci, cd, cc) = index.search_batched(q, k, o, stages=True)
    for i in range(nq):
        H.check_txh_query(oix, q[i], k, idx[i, :cnt[i]], dist[i, :cnt[i]], tok[i], tokd[i],
                          ci[i, :cc[i]], cd[i, :cc[i]], what="q%d" % i)


def test_ah_search_byte_codes_default_config():
    """AsymmetricHasherConfig::default(): 256 codes x 8 subspaces (hashes/hasher.rs:36-46)."""
    n, dim, S, K, k = 20000, 64, 8, 256, 10
    rows, data, stride, ix, kw = H.make_ah_case(n, dim, S, seed=33, K=K, pq_iters=2)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(21, dim, 34)
    o = hip.default_opts()
    o.exact_reorder = 0
    idx, dist, cnt = index.search_batched(q, k, o)
    o.exact_reorder, o.pre_reorder_k = 1, 200
    idx2, dist2, cnt2 = index.search_batched(q, k, o)
    for i in range(q.shape[0]):
        oi, od = orc.ah_search(ix["codebook"], ix["codes"], q[i], k)
        H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], oi, od, what="ah q%d" % i)
        oi, od = orc.ah_search_with_reordering(ix["codebook"], ix["codes"], data, stride, q[i], k, 200)
        H.assert_topk_equal_up_to_ties(idx2[i, :cnt2[i]], dist2[i, :cnt2[i]], oi, od, what="ahr q%d" % i)


def test_byte_code_limits():
    rows, data, stride, ix, kw = H.make_ah_case(600, 96, 32, seed=35, K=17, pq_iters=1)
    with pytest.raises(hip.ScannError) as e:   # 32 subspaces x 256 slots does not fit the LDS tables
        hip.txh_create(**kw)
    assert e.value.code == hip.UNIMPLEMENTED


@pytest.mark.parametrize("measure", [hip.SQUARED_L2, hip.DOT_PRODUCT, hip.L2])
@pytest.mark.parametrize("n,dim", [(20000, 128), (3000, 50)])
def test_bf_search_radius(measure, n, dim):
    """BruteForceSearcher::search_radius (brute_force/searcher.rs:142-167)."""
    rows = synth.uniform_f32(n, dim, 61)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, measure)
    q = synth.uniform_f32(3, dim, 62)
    for i in range(3):
        d = orc.one_to_many(q[i], data, stride, n, measure)
        for frac in (0.0, 0.001, 0.6):     # empty, a few, most of the dataset (> LDS sort sizes)
            radius = float(np.sort(d)[int(frac * (n - 1))]) if frac else float(d.min()) - 1.0
            oi, od = orc.bf_search_radius(data, n, dim, stride, measure, q[i], radius)
            gi, gd, cnt = hip.bf_search_radius(index, q[i], radius)
            assert cnt == oi.size
            assert np.array_equal(bits(gd), bits(od))
            H.assert_topk_equal_up_to_ties(gi, gd, oi, od)
    d0 = np.sort(orc.one_to_many(q[0], data, stride, n, measure))
    gi, gd, cnt = hip.bf_search_radius(index, q[0], float(d0[100]), capacity=10)
    assert cnt == int(np.searchsorted(d0, d0[100], side="right")) and gi.size == 10
    with pytest.raises(hip.ScannError):
        hip.bf_search_radius(index, q[0][:dim - 1], 1.0)


# ---- index build: K-means on the GPU (trees/kmeans.rs:210-414) -------------------------------------
@pytest.mark.parametrize("n,dim,k,col,sub,thr", [
    (5000, 32, 16, 0, 32, 128), (20000, 96, 50, 0, 96, 128), (3000, 64, 16, 8, 4, 128), (4000, 64, 256, 61, 3, 128),
    (300, 7, 3, 0, 7, 128),
    # dim >= simd_threshold (the reference's default 128, trees/kmeans.rs:59): squared_l2_avx2's order
    (6000, 128, 40, 0, 128, 128), (3000, 256, 24, 0, 256, 128), (2500, 200, 10, 32, 131, 128),
    (2000, 140, 9, 0, 140, 128),
    # configurable threshold: 0 = always the AVX2 order, a huge one = always sequential
    (3000, 64, 16, 3, 37, 0), (2000, 160, 8, 0, 160, 1 << 30)])
def test_kmeans_lloyd_matches_oracle(n, dim, k, col, sub, thr):
    """Same initial centres -> bit-identical centres, assignments, iteration count AND f64 inertia as the
    CPU restatement of KMeans::fit_single, on both sides of simd_threshold (trees/kmeans.rs:419-431)."""
    rows, _ = synth.clustered_f32(n, dim, 71, n_clusters=max(4, k // 2))
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, hip.SQUARED_L2)
    pick = (synth.splitmix64(5, 0, k) % np.uint64(n)).astype(np.int64)
    init = np.ascontiguousarray(rows[pick][:, col:col + sub])
    gc, ga, gs, gi, git, gconv = hip.kmeans_lloyd(index, init, max_iterations=12, col_offset=col,
                                                  simd_threshold=thr)
    oc, oa, os_, oi, oit, oconv = orc.kmeans_lloyd(data, n, stride, sub, init, max_iterations=12,
                                                   col_offset=col, simd_threshold=thr)
    assert git == oit and gconv == oconv
    assert np.array_equal(bits(gc), bits(oc))
    assert np.array_equal(ga, oa) and np.array_equal(gs, os_)
    assert np.float64(gi).view(np.uint64) == np.float64(oi).view(np.uint64)   # the datapoint-order f64 sum


def test_kmeans_inertia_sequential_fallback():
    """Distances spanning > 2^53 ulps of the smallest one: the reduction tree is not provably exact, the
    library must take the datapoint-order chain (kmeans.rs:376) and still match the oracle bit for bit."""
    rng = np.random.default_rng(3)
    n, dim = 20000, 8
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[:64] *= np.float32(1e-9)            # tiny distances next to O(1e4) ones
    rows[64:128] *= np.float32(1e3)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, hip.SQUARED_L2)
    init = np.zeros((2, dim), np.float32)
    init[1] = 0.5
    g = hip.kmeans_lloyd(index, init, max_iterations=3)
    o = orc.kmeans_lloyd(data, n, stride, dim, init, max_iterations=3)
    assert np.array_equal(bits(g[0]), bits(o[0])) and np.array_equal(g[1], o[1]) and g[4] == o[4]
    assert np.float64(g[3]).view(np.uint64) == np.float64(o[3]).view(np.uint64)


def test_kmeans_init_pp_avx_order_min_distances():
    """Seeding with dim >= simd_threshold: deterministic, every seed is a row (the D^2 sampling itself is
    a documented deviation: f64 tree sums, scann_hip.h)."""
    rows, _ = synth.clustered_f32(3000, 128, 5, n_clusters=12)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, 3000, 128, stride, hip.SQUARED_L2)
    a = hip.kmeans_init_pp(index, 12, seed=9)
    assert np.array_equal(a, hip.kmeans_init_pp(index, 12, seed=9))
    assert all(any(np.array_equal(r, p) for p in rows) for r in a)


def test_kmeans_reference_unit_tests_on_gpu():
    """trees/kmeans.rs:461-510: three clear clusters; empty clusters."""
    x = np.array([[bx + np.float32(i) * np.float32(0.1), by + np.float32(i) * np.float32(0.05)]
                  for bx, by in ((0.0, 0.0), (10.0, 10.0), (0.0, 10.0)) for i in range(10)], np.float32)
    data, stride = orc.to_strided(x)
    index = hip.bf_create(data, 30, 2, stride, hip.SQUARED_L2)
    init = hip.kmeans_init_pp(index, 3, seed=42)
    assert init.shape == (3, 2) and all(any(np.array_equal(r, p) for p in x) for r in init)
    c, a, sizes, inertia, iters, conv = hip.kmeans_lloyd(index, init)
    assert c.shape == (3, 2) and a.size == 30 and sizes.sum() == 30 and (conv or iters > 0)
    far = np.array([[0, 0], [10, 10], [1000, 1000]], np.float32)
    c, a, sizes, *_ = hip.kmeans_lloyd(index, far, max_iterations=1)
    assert np.array_equal(c[2], x[2])                   # empty cluster c takes row c % n
    oc, *_ = orc.kmeans_lloyd(data, 30, stride, 2, far, max_iterations=1)
    assert np.array_equal(bits(c), bits(oc))


def test_kmeans_init_pp_spreads_seeds():
    rows, labels = synth.clustered_f32(20000, 32, 72, n_clusters=20)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, 20000, 32, stride, hip.SQUARED_L2)
    a = hip.kmeans_init_pp(index, 20, seed=7)
    assert np.array_equal(a, hip.kmeans_init_pp(index, 20, seed=7))   # deterministic in the seed
    # D^2 sampling lands in (almost) every well-separated cluster
    owner = [int(labels[np.flatnonzero((rows == r).all(1))[0]]) for r in a]
    assert len(set(owner)) >= 16


def test_pack_blocks_matches_reference():
    """scann_hip_txh_pack_blocks_device == sharding.pack_blocks_reference (the all_to_all layout)."""
    import ctypes
    import torch
    from scann_rust_amd import sharding
    world, nq, m = 4, 24, 37
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 2 ** 63, size=(nq, m), dtype=np.uint64)
    idx = rng.integers(0, 2 ** 32, size=(nq, m), dtype=np.uint32)
    exact = rng.random((nq, m), dtype=np.float32)
    cnt = rng.integers(0, m + 1, size=nq, dtype=np.uint32)
    want = sharding.pack_blocks_reference(keys, idx, exact, cnt, world)
    bb = want.shape[1]
    dev = torch.device("cuda", 0)
    tk, ti, te, tc = (torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(dev) for a in (keys, idx, exact, cnt))
    out = torch.zeros((world, bb), dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    hip.check(hip.load().scann_hip_txh_pack_blocks_device(hip.context(0), world, nq, m, p(tk), p(ti), p(te), p(tc),
                                                          p(out), bb, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    _, _, _, oc, _ = sharding.block_layout(nq, m, world)
    used = oc + (nq // world) * 4
    assert np.array_equal(got[:, :used], want[:, :used])
    with pytest.raises(hip.ScannError):
        hip.check(hip.load().scann_hip_txh_pack_blocks_device(hip.context(0), 5, nq, m, p(tk), p(ti), p(te), p(tc),
                                                              p(out), bb, None))


# ---- resident-table scan kernel (long leaves scanned by >= 128 queries) ----------------------------
@pytest.fixture
def force_resident(monkeypatch):
    """adc_scan_res_kernel is selected by a size heuristic (api.hip); SCANN_HIP_RESIDENT=2 selects it
    for every 4-bit S <= 32 index so that small test cases run it too."""
    monkeypatch.setenv("SCANN_HIP_RESIDENT", "2")
    monkeypatch.setenv("SCANN_HIP_RES_CL", "3")


@pytest.mark.parametrize("n,dim,S,nq,pre_k", [(9000, 128, 32, 160, 300), (30000, 64, 16, 256, 77), (5000, 32, 8, 130, 2000),
                                              (700, 32, 8, 5, 40)])
def test_ah_resident_scan_matches_oracle(force_resident, n, dim, S, nq, pre_k):
    """Resident-table scan kernel: same candidates, same results as the oracle for every query
    (incl. a partial last chunk, fewer quads than the kernel keeps resident, and a filter)."""
    k = 10
    rows, data, stride, ix, kw = H.make_ah_case(n, dim, S, seed=81, pq_iters=2)
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(nq, dim, 82)
    o = hip.default_opts()
    o.exact_reorder, o.pre_reorder_k = 1, pre_k
    idx, dist, cnt = index.search_batched(q, k, o)
    bits_ = hip.allow_bitmap(n, np.arange(0, n, 3))
    fidx, fdist, fcnt = index.search_batched(q, k, o, allow=bits_)
    sub = np.arange(0, n, 3)
    for i in range(0, nq, 7):
        oi, od = orc.ah_search_with_reordering(ix["codebook"], ix["codes"], data, stride, q[i], k, pre_k)
        H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], oi, od, what="res q%d" % i)
        oi, od = orc.ah_search_with_reordering(ix["codebook"], np.ascontiguousarray(ix["codes"][sub]),
                                               np.ascontiguousarray(data.reshape(n, stride)[sub]), stride,
                                               q[i], k, pre_k)
        H.assert_topk_equal_up_to_ties(fidx[i, :fcnt[i]], fdist[i, :fcnt[i]], sub[oi].astype(np.uint32), od,
                                       what="res filtered q%d" % i)


def test_txh_resident_scan_two_big_leaves(force_resident):
    """Two big partitions, every query searches both: 64 quads per leaf -> resident-table kernel."""
    n, dim, S, nq, k = 12000, 64, 16, 256, 10
    rows = synth.uniform_f32(n, dim, 83)
    centers = np.stack([np.full(dim, 0.25, np.float32), np.full(dim, 0.75, np.float32)])
    assign = (rows.mean(1) > 0.5).astype(np.int64)
    built = trainer.build_txh_index(rows, 2, S, centers=centers, assign=assign, pq_iters=2)
    data, stride = orc.to_strided(rows)
    oix = orc.TxhIndex(data, stride, dim, built["centers"], built["leaf_off"], built["leaf_ids"],
                       built["codebook"], built["codes"], partitions_to_search=2,
                       pre_reorder_multiplier=20.0)
    index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=built["centers"],
                           leaf_offsets=built["leaf_off"], leaf_ids=built["leaf_ids"],
                           codebook=built["codebook"], codes=built["codes"], partitions_to_search=2,
                           pre_reorder_multiplier=20.0)
    q = synth.uniform_f32(nq, dim, 84)
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = 2, orc.pre_reorder_k(k, 20.0)
    idx, dist, cnt, (tok, tokd, ci, cd, cc) = index.search_batched(q, k, o, stages=True)
    for i in range(0, nq, 9):
        H.check_txh_query(oix, q[i], k, idx[i, :cnt[i]], dist[i, :cnt[i]], tok[i], tokd[i],
                          ci[i, :cc[i]], cd[i, :cc[i]], what="res txh q%d" % i)


# ---- bf16 shortlist path of the brute-force searcher ----------------------------------------------
@pytest.fixture
def force_shortlist(monkeypatch):
    """The bf16-shortlist path is taken by big batches on big indexes; these knobs select it for
    small test cases (read at index creation / per search)."""
    monkeypatch.setenv("SCANN_HIP_BF_SHORTLIST_MIN_ROWS", "1")
    monkeypatch.setenv("SCANN_HIP_BF_SHORTLIST_MIN_QUERIES", "1")
    monkeypatch.setenv("SCANN_HIP_SMALL", "0")      # (batches of <= 16 queries would take the small-batch pipeline)


@pytest.mark.parametrize("measure", [hip.DOT_PRODUCT, hip.SQUARED_L2, hip.L2])
@pytest.mark.parametrize("n,dim,k,nq", [(20000, 128, 10, 40), (9000, 64, 3, 130), (30000, 96, 16, 33), (5000, 32, 1, 7),
                                        (12000, 256, 10, 20)])
def test_bf_shortlist_matches_oracle(force_shortlist, measure, n, dim, k, nq):
    """bf16 MFMA scores shortlist 4k rows, the reference's f32 arithmetic re-scores them, an error
    bound proves nothing else can enter the top k: results bit-identical to the oracle."""
    rows = synth.uniform_f32(n, dim, 42)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, measure)
    index.enable_timing(True)
    q = synth.uniform_f32(nq, dim, 123)
    idx, dist, cnt = index.search_batched(q, k)
    assert index.last_kernel_ms()[1] == "bf_bf16_kernel"
    for i in range(nq):
        oi, od = orc.bf_search(data, n, dim, stride, measure, q[i], k)
        assert cnt[i] == oi.size
        H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], oi, od, what="sl q%d" % i)
    o = hip.default_opts()
    o.bf_exact = 1
    idx2, dist2, _ = index.search_batched(q, k, o)      # the exact kernels give the same rows
    assert np.array_equal(bits(dist2), bits(dist))


@pytest.mark.parametrize("measure", [hip.DOT_PRODUCT, hip.SQUARED_L2])
def test_bf_shortlist_unverifiable_falls_back(force_shortlist, measure):
    """Near-duplicate rows: bf16 scores cannot separate them, the verification fails, the host
    entry point repeats the batch on the exact kernels (device entry: status Aborted)."""
    import ctypes
    import torch
    n, dim, k, nq = 8000, 64, 10, 48
    base = synth.uniform_f32(1, dim, 5)
    rows = (base + np.float32(1e-4) * synth.uniform_f32(n, dim, 6)).astype(np.float32)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, measure)
    q = synth.uniform_f32(nq, dim, 7)
    idx, dist, cnt = index.search_batched(q, k)
    for i in range(0, nq, 5):
        oi, od = orc.bf_search(data, n, dim, stride, measure, q[i], k)
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], oi, od, what="dup q%d" % i)
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(q).to(dev)
    oi_t = torch.empty((nq, k), dtype=torch.int32, device=dev)
    od_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    oc_t = torch.empty((nq,), dtype=torch.int32, device=dev)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    L = hip.load()
    hip.check(L.scann_hip_search_batched_device(index.h, p(qd), nq, dim, k, None, p(oi_t), p(od_t), p(oc_t), sp))
    with pytest.raises(hip.ScannError) as e:
        hip.check(L.scann_hip_index_last_device_status(index.h, sp))
    assert e.value.code == 10     # Aborted: repeat with bf_exact = 1
    o = hip.default_opts()
    o.bf_exact = 1
    hip.check(L.scann_hip_search_batched_device(index.h, p(qd), nq, dim, k, ctypes.byref(o), p(oi_t), p(od_t),
                                                p(oc_t), sp))
    hip.check(L.scann_hip_index_last_device_status(index.h, sp))
    assert np.array_equal(bits(od_t.cpu().numpy()), bits(dist))


def test_bf_shortlist_error_bound_holds(force_shortlist):
    """The verification relies on |split-bf16 score - exact| <= E for every row.  Restate the
    split-bf16 score on the CPU (f64 sum of the three product groups) for adversarial magnitudes
    and check it against the bound; the GPU's f32 accumulation is covered by the bound's second
    term, checked through the GPU result staying exact on the same data."""
    dim, n = 128, 4096
    rng = np.random.default_rng(3)
    rows = (rng.standard_normal((n, dim)) * np.exp(rng.uniform(-6, 6, (n, 1)))).astype(np.float32)
    q = (rng.standard_normal((8, dim)) * 100).astype(np.float32)

    def bf16(x):
        u = x.view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return r.astype(np.uint32).view(np.float32)

    xh = bf16(rows); xl = bf16(rows - xh)
    qh = bf16(q); ql = bf16(q - qh)
    approx = (qh.astype(np.float64) @ xh.T.astype(np.float64) + qh.astype(np.float64) @ xl.T.astype(np.float64)
              + ql.astype(np.float64) @ xh.T.astype(np.float64))
    exact = q.astype(np.float64) @ rows.T.astype(np.float64)
    c = 1.01 * (3.1 / 65536 + (3 * dim + 64) / 8388608)
    bound = c * np.linalg.norm(q.astype(np.float64), axis=1)[:, None] * np.linalg.norm(rows.astype(np.float64), axis=1)[None, :]
    assert np.all(np.abs(approx - exact) <= bound)
    assert np.abs(approx - exact).max() <= 0.5 * bound.max()        # the dropped-terms part alone: slack for f32
    data, stride = orc.to_strided(rows)
    for measure in (hip.DOT_PRODUCT, hip.SQUARED_L2):
        index = hip.bf_create(data, n, dim, stride, measure)
        idx, dist, cnt = index.search_batched(np.tile(q, (5, 1)), 10)
        for i in range(8):
            oi, od = orc.bf_search(data, n, dim, stride, measure, q[i], 10)
            H.assert_topk_equal_up_to_ties(idx[i], dist[i], oi, od, what="scaled q%d" % i)


# ---- Scann facade modes (scann.rs:181-294) as configurations of the same index ------------------
def _partition_case(n, dim, L, seed):
    rows = synth.uniform_f32(n, dim, seed)
    data, stride = orc.to_strided(rows)
    centers, assign = trainer.kmeans(rows, L, iters=3, seed=seed)
    order = np.argsort(assign, kind="stable").astype(np.uint32)
    leaf_off = np.zeros(centers.shape[0] + 1, np.uint32)
    leaf_off[1:] = np.cumsum(np.bincount(assign, minlength=centers.shape[0]))
    return rows, data, stride, centers, leaf_off, order


@pytest.mark.parametrize("measure", [hip.SQUARED_L2, hip.L2, hip.DOT_PRODUCT, hip.L1, hip.COSINE])
@pytest.mark.parametrize("n,dim,L,P,k", [(3000, 64, 20, 5, 10), (5000, 100, 37, 37, 25), (800, 33, 8, 3, 900),
                                         (20000, 128, 16, 4, 10)])
def test_scann_partitioned_mode(measure, n, dim, L, P, k):
    """search_partitioned (scann.rs:213-252): exact distances bit for bit, and -- both sides being a
    stable sort of the same stream -- identical indices, ties included."""
    rows, data, stride, centers, leaf_off, leaf_ids = _partition_case(n, dim, L, seed=21)
    index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=centers,
                           leaf_offsets=leaf_off, leaf_ids=leaf_ids, codebook=None, codes=None,
                           partitions_to_search=P, distance_measure=measure)
    q = synth.uniform_f32(70, dim, 123)
    q[3] = rows[17]   # an exact hit
    idx, dist, cnt = index.search_batched(q, k)
    for i in range(q.shape[0]):
        oi, od = orc.scann_search_partitioned(centers, leaf_off, leaf_ids, data, stride, measure, q[i], P, k)
        assert cnt[i] == oi.size
        assert np.array_equal(bits(dist[i, :oi.size]), bits(od)), (i, dist[i, :oi.size], od)
        assert np.array_equal(idx[i, :oi.size], oi)


def test_scann_partitioned_mode_duplicates():
    """Duplicate rows tie exactly: the stable order (token order, then leaf order) decides."""
    rows, data, stride, centers, leaf_off, leaf_ids = _partition_case(2000, 32, 6, seed=5)
    rows[1000:] = rows[:1000]
    data, stride = orc.to_strided(rows)
    centers, assign = trainer.kmeans(rows, 6, iters=3, seed=5)
    leaf_ids = np.argsort(assign, kind="stable").astype(np.uint32)
    leaf_off = np.zeros(7, np.uint32)
    leaf_off[1:] = np.cumsum(np.bincount(assign, minlength=6))
    index = hip.txh_create(data=data, n_rows=2000, dim=32, stride=stride, centers=centers,
                           leaf_offsets=leaf_off, leaf_ids=leaf_ids, codebook=None, codes=None,
                           partitions_to_search=3)
    q = synth.uniform_f32(20, 32, 9)
    idx, dist, cnt = index.search_batched(q, 15)
    for i in range(20):
        oi, od = orc.scann_search_partitioned(centers, leaf_off, leaf_ids, data, stride, 0, q[i], 3, 15)
        assert np.array_equal(idx[i, :cnt[i]], oi) and np.array_equal(bits(dist[i, :cnt[i]]), bits(od))


@pytest.mark.parametrize("measure,reorder", [(hip.SQUARED_L2, False), (hip.SQUARED_L2, True),
                                             (hip.DOT_PRODUCT, True), (hip.L2, True), (hip.L1, True),
                                             (hip.COSINE, True)])
@pytest.mark.parametrize("K,S", [(256, 8), (16, 16)])
def test_scann_tree_ah_mode(measure, reorder, K, S):
    """search_tree_ah (scann.rs:255-294): one non-residual table per query, stable sort of every
    scanned row's LUT sum, first k; then the exact reordering of the truncated list (:199-209)."""
    n, dim, L, P, k = 4000, 64, 16, 4, 12
    rows, data, stride, ix, oix, kw = H.make_txh_case(n, dim, L, S, seed=8, K=K, use_residuals=False, P=P,
                                                      kmeans_iters=3, pq_iters=3)
    kw["distance_measure"] = measure
    index = hip.txh_create(**kw)
    codes_dp = np.empty_like(ix["codes"])
    codes_dp[ix["leaf_ids"]] = ix["codes"]
    q = synth.uniform_f32(50, dim, 77)
    o = hip.default_opts()
    o.pre_reorder_k = k
    o.exact_reorder = 1 if reorder else 0
    idx, dist, cnt = index.search_batched(q, k, opts=o)
    for i in range(q.shape[0]):
        oi, od = orc.scann_search_tree_ah(ix["centers"], ix["leaf_off"], ix["leaf_ids"], ix["codebook"],
                                          codes_dp, data, stride, measure, reorder, q[i], P, k)
        assert cnt[i] == oi.size
        assert np.array_equal(bits(dist[i, :oi.size]), bits(od)), (i, dist[i], od)
        assert np.array_equal(idx[i, :oi.size], oi)


@pytest.mark.parametrize("measure", [hip.SQUARED_L2, hip.DOT_PRODUCT])
def test_scann_hashed_mode_reorder(measure):
    """SearchMode::Hashed + exact reordering: AsymmetricHasher::search's k rows re-scored with the
    configured measure (scann.rs:189-209)."""
    n, dim, S, k = 3000, 64, 8, 10
    rows, data, stride, ix, kw = H.make_ah_case(n, dim, S, seed=4, K=256, pq_iters=3)
    kw["distance_measure"] = measure
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(40, dim, 5)
    o = hip.default_opts()
    o.pre_reorder_k = k
    o.exact_reorder = 1
    idx, dist, cnt = index.search_batched(q, k, opts=o)
    for i in range(q.shape[0]):
        ai, ad = orc.ah_search(ix["codebook"], ix["codes"], q[i], k)
        oi, od = orc.reorder_measure(data, stride, dim, measure, q[i], ai, k)
        assert cnt[i] == oi.size
        assert np.array_equal(bits(dist[i, :oi.size]), bits(od))
        H.assert_topk_equal_up_to_ties(idx[i, :oi.size], dist[i, :oi.size], oi, od, what="hashed+reorder")


def test_scann_partitioned_mode_errors():
    rows, data, stride, centers, leaf_off, leaf_ids = _partition_case(500, 16, 4, seed=2)
    with pytest.raises(hip.ScannError) as e:   # needs the dataset rows
        hip.txh_create(data=None, n_rows=500, dim=16, stride=stride, centers=centers, leaf_offsets=leaf_off,
                       leaf_ids=leaf_ids, codebook=None, codes=None)
    assert e.value.code == hip.INVALID_ARGUMENT
    index = hip.txh_create(data=data, n_rows=500, dim=16, stride=stride, centers=centers,
                           leaf_offsets=leaf_off, leaf_ids=leaf_ids, codebook=None, codes=None)
    with pytest.raises(hip.ScannError) as e:
        index.search_batched(np.zeros((1, 8), np.float32), 5)
    assert e.value.code == hip.INVALID_ARGUMENT


# ---- small-batch brute force: bf_stream_kernel (a few queries, one coalesced database pass) -------
@pytest.mark.parametrize("measure", [hip.SQUARED_L2, hip.L2, hip.DOT_PRODUCT])
@pytest.mark.parametrize("n,dim", [(3000, 128), (70, 64), (5000, 100), (1000, 33), (4097, 8), (2500, 24),
                                   (900, 200), (1000, 512)])   # 1000 x 512: stress_tests.rs:300-323
def test_bf_small_batch_stream_kernel(measure, n, dim):
    """1..16 queries take the streaming kernel: all-pairs distances bitwise (ragged dims: slices of
    8/16/24 dims and scalar tails; row counts that are not multiples of 64) and exact top-k."""
    rows = synth.uniform_f32(n, dim, 42)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, measure)
    for nq in (1, 5, 8, 9, 16):
        q = synth.uniform_f32(nq, dim, 100 + nq)
        got = hip.bf_distances(index, q)
        for i in range(nq):
            assert np.array_equal(bits(got[i]), bits(orc.one_to_many(q[i], data, stride, n, measure))), (nq, i)
        k = min(10, n)
        idx, dist, cnt = index.search_batched(q, k)
        oi, od, oc = orc.bf_search_batched(data, n, dim, stride, measure, q, k)
        assert np.array_equal(cnt, oc)
        for i in range(nq):
            H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], oi[i, :cnt[i]], od[i, :cnt[i]],
                                           what="stream nq=%d q%d" % (nq, i))


@pytest.mark.parametrize("tail", ["0.9", "0.3", "1e-2"])
@pytest.mark.parametrize("measure", [hip.DOT_PRODUCT, hip.SQUARED_L2])
def test_bf_shortlist_aggressive_filter_bound_stays_exact(monkeypatch, force_shortlist, tail, measure):
    """A filter bound cut far too deep (tail probability 0.9: the best sample score) leaves shortlists
    shorter than kp, some shorter than k.  The former are proven against the bound itself, the latter
    are flagged and redone by the exact kernels: the rows returned by the host entry stay exact."""
    monkeypatch.setenv("SCANN_HIP_BF_SHORTLIST_TAIL", tail)
    n, dim, k, nq = 70000, 64, 10, 96
    rows = synth.uniform_f32(n, dim, 42)
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, measure)
    q = synth.uniform_f32(nq, dim, 123)
    idx, dist, cnt = index.search_batched(q, k)
    oi, od, oc = orc.bf_search_batched(data, n, dim, stride, measure, q, k)
    assert np.array_equal(cnt, oc)
    for i in range(nq):
        H.assert_topk_equal_up_to_ties(idx[i], dist[i], oi[i], od[i], what="tail %s q%d" % (tail, i))


# ---- multi-GPU exchange inside the library (csrc/comm.hip) ---------------------------------------------
def test_sharded_search_through_rccl_world1(monkeypatch):
    """scann_hip_txh_search_sharded_device with a 1-rank communicator: the whole path (local stage, block
    packing with batch padding, grouped ncclSend/ncclRecv to self, merge, in-place ncclAllGather, copy-out)
    runs through RCCL on the communicator's stream and must equal the plain search bit for bit; two
    alternating caller streams exercise the double-buffered event ordering."""
    import ctypes
    import torch
    rows, data, stride, ix, oix, kw = H.make_txh_case(6000, 64, 16, 8, seed=21, P=5, mult=4.0, kmeans_iters=3,
                                                      pq_iters=3)
    sizes = (kw["leaf_offsets"][1:] - kw["leaf_offsets"][:-1]).astype(np.uint32)
    # a "sharded" index holding every leaf: global sizes given, rows in CSR order
    order = kw["leaf_ids"].astype(np.int64)
    data2 = np.ascontiguousarray(np.asarray(data).reshape(6000, stride)[order])
    skw = dict(kw, data=data2, leaf_sizes_global=sizes, data_is_csr_order=True)
    index = hip.txh_create(**skw)
    comm = hip.Comm(hip.Comm.unique_id(), 0, 1)
    dev = torch.device("cuda", 0)
    k, nq = 10, 37
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = 5, 40
    L = hip.load()
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    outs = []
    qs = [synth.uniform_f32(nq, 64, 100 + i) for i in range(4)]
    hip.check(L.scann_hip_index_reserve(index.h, nq, k, ctypes.byref(o)))
    for i, q in enumerate(qs):
        st = streams[i & 1]
        qd = torch.from_numpy(q).to(dev)
        oi = torch.empty((nq, k), dtype=torch.int32, device=dev)
        od = torch.empty((nq, k), dtype=torch.float32, device=dev)
        oc = torch.empty((nq,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        hip.check(L.scann_hip_txh_search_sharded_device(index.h, comm.h, ctypes.c_void_p(qd.data_ptr()), nq, 64, k,
                                                        ctypes.byref(o), 0, ctypes.c_void_p(oi.data_ptr()),
                                                        ctypes.c_void_p(od.data_ptr()), ctypes.c_void_p(oc.data_ptr()),
                                                        ctypes.c_void_p(st.cuda_stream)))
        outs.append((qd, oi, od, oc))
    torch.cuda.synchronize()
    comm.last_status()
    plain = hip.txh_create(**kw)
    for q, (qd, oi, od, oc) in zip(qs, outs):
        wi, wd, wc = plain.search_batched(q, k, o)
        assert np.array_equal(oc.cpu().numpy().view(np.uint32), wc)
        assert np.array_equal(od.cpu().numpy().view(np.uint32), wd.view(np.uint32))
        assert np.array_equal(oi.cpu().numpy().view(np.uint32), wi)
    for i in range(0, nq, 9):                       # and the oracle on a few rows
        wi, wd = orc.txh_search(orc.TxhIndex(data, stride, 64, ix["centers"], ix["leaf_off"], ix["leaf_ids"],
                                             ix["codebook"], ix["codes"], partitions_to_search=5,
                                             pre_reorder_multiplier=4.0), qs[0][i], k)
        H.assert_topk_equal_up_to_ties(outs[0][1][i].cpu().numpy().view(np.uint32)[:wi.size],
                                       outs[0][2][i].cpu().numpy()[:wi.size], wi, wd, what="sharded q%d" % i)
    # m_local = m: compact destination blocks (a count per query, the entries one behind the other), same rows
    def call(m_local):
        hip.check(L.scann_hip_txh_search_sharded_device(index.h, comm.h, ctypes.c_void_p(outs[0][0].data_ptr()), nq, 64, k,
                                                        ctypes.byref(o), m_local, ctypes.c_void_p(outs[0][1].data_ptr()),
                                                        ctypes.c_void_p(outs[0][2].data_ptr()),
                                                        ctypes.c_void_p(outs[0][3].data_ptr()),
                                                        ctypes.c_void_p(streams[0].cuda_stream)))
        torch.cuda.synchronize()
    want0 = [t.cpu().numpy().copy() for t in outs[0][1:]]
    for t in outs[0][1:]:
        t.zero_()
    call(40)
    comm.last_status()
    for t, w in zip(outs[0][1:], want0):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), w.view(np.uint32))
    # blocks with room for a twentieth of the worst case: the overflow is flagged and reported as Aborted ...
    monkeypatch.setenv("SCANN_HIP_COMM_FILL", "0.05")
    assert hip.comm_layout(nq, 1, 40, k)["cap"] < nq * 40 // 10
    call(40)
    with pytest.raises(hip.ScannError) as e:
        comm.last_status()
    assert e.value.code == 10
    # ... and m_local = 0 (worst-case blocks whatever the fill factor) repairs it
    for t in outs[0][1:]:
        t.zero_()
    call(0)
    comm.last_status()
    for t, w in zip(outs[0][1:], want0):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), w.view(np.uint32))
    monkeypatch.delenv("SCANN_HIP_COMM_FILL")
    # m_local < m that is too short must be reported, not silently wrong
    hip.check(L.scann_hip_txh_search_sharded_device(index.h, comm.h, ctypes.c_void_p(outs[0][0].data_ptr()), nq, 64, k,
                                                    ctypes.byref(o), 12, ctypes.c_void_p(outs[0][1].data_ptr()),
                                                    ctypes.c_void_p(outs[0][2].data_ptr()),
                                                    ctypes.c_void_p(outs[0][3].data_ptr()),
                                                    ctypes.c_void_p(streams[0].cuda_stream)))
    torch.cuda.synchronize()
    with pytest.raises(hip.ScannError) as e:
        comm.last_status()
    assert e.value.code == 10
    comm.close()


# ---- the three scan kernels must produce the same candidate lists ----------------------------------------
@pytest.mark.parametrize("mode", ["ah", "txh"])
def test_scan_paths_agree(mode, monkeypatch):
    """adc_scan_kernel (f32 LDS gather per chunk), adc_scan_res_kernel (resident tables) and the
    integer-MFMA prefilter + exact refine (adc_mfma_kernel / adc_mfma16_kernel + adc_refine_kernel) are schedules of the
    same arithmetic: candidates (approximate distances bitwise, index sets) and final rows must be
    identical, with a filter bound in force (n >> m) and with an allow-bitmap."""
    if mode == "ah":
        rows, data, stride, ix, kw = H.make_ah_case(60000, 128, 32, seed=31, pq_iters=3)
        o = hip.default_opts()
        o.pre_reorder_k = 300
        n = 60000
    else:
        rows, data, stride, ix, oix, kw = H.make_txh_case(80000, 96, 20, 24, seed=32, P=6, kmeans_iters=3,
                                                          pq_iters=3, clustered=True)
        o = hip.default_opts()
        o.partitions_to_search, o.pre_reorder_k = 6, 250
        n = 80000
    index = hip.txh_create(**kw)
    q = synth.uniform_f32(96, kw["dim"], 77) if mode == "ah" else synth.clustered_f32(96, 96, 33, n_clusters=20)[0]
    allow = hip.allow_bitmap(n, np.arange(0, n, 3))
    got = {}
    for name, env in (("chunk", {"SCANN_HIP_MFMA": "0", "SCANN_HIP_RESIDENT": "0"}),
                      ("resident", {"SCANN_HIP_MFMA": "0", "SCANN_HIP_RESIDENT": "2"}),
                      ("smfmac", {"SCANN_HIP_MFMA": "2"}),        # 32-pair tiles on v_smfmac_i32_32x32x64_i8 (default form)
                      ("mfma", {"SCANN_HIP_MFMA": "2", "SCANN_HIP_SMFMAC": "0"}),   # 32-pair tiles (v_mfma_i32_32x32x32_i8)
                      ("mfma16", {"SCANN_HIP_MFMA": "3"})):       # 16-pair tiles (v_mfma_i32_16x16x64_i8)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        plain = index.search_batched(q, 10, o, stages=True)
        filt = index.search_batched(q, 10, o, allow=allow)
        o.allow_bitmap, o.allow_bitmap_bits = None, 0
        got[name] = (plain, filt)
        monkeypatch.delenv("SCANN_HIP_RESIDENT", raising=False)
        monkeypatch.delenv("SCANN_HIP_SMFMAC", raising=False)
    ref_plain, ref_filt = got["chunk"]
    # the reference path against the oracle (every other path is then compared with it bit for bit)
    for i in range(0, q.shape[0], 5):
        cnt_i = ref_plain[2][i]
        if mode == "ah":
            wi, wd = orc.ah_search_with_reordering(ix["codebook"], ix["codes"], data, stride, q[i], 10, o.pre_reorder_k)
            assert cnt_i == wi.size
            H.assert_topk_equal_up_to_ties(ref_plain[0][i, :cnt_i], ref_plain[1][i, :cnt_i], wi, wd, what="oracle q%d" % i)
        else:
            tok, tokd, ci, cd, cc = ref_plain[3]
            oix = orc.TxhIndex(data, stride, kw["dim"], ix["centers"], ix["leaf_off"], ix["leaf_ids"], ix["codebook"],
                               ix["codes"], partitions_to_search=6, pre_reorder_multiplier=25.0)   # m = 250
            H.check_txh_query(oix, q[i], 10, ref_plain[0][i, :cnt_i], ref_plain[1][i, :cnt_i], tok[i], tokd[i],
                              ci[i, :cc[i]], cd[i, :cc[i]], what="oracle q%d" % i)
    for name in ("resident", "smfmac", "mfma", "mfma16"):
        plain, filt = got[name]
        assert np.array_equal(plain[0], ref_plain[0]) and np.array_equal(bits(plain[1]), bits(ref_plain[1])), name
        assert np.array_equal(plain[2], ref_plain[2]), name
        (tok, tokd, ci, cd, cc), (rtok, rtokd, rci, rcd, rcc) = plain[3], ref_plain[3]
        assert np.array_equal(cc, rcc) and np.array_equal(bits(cd), bits(rcd)), name + ": candidate distances"
        for i in range(q.shape[0]):
            assert sorted(ci[i, :cc[i]].tolist()) == sorted(rci[i, :rcc[i]].tolist()), name + ": candidate set"
        assert np.array_equal(filt[0], ref_filt[0]) and np.array_equal(bits(filt[1]), bits(ref_filt[1])), name + " filtered"
    assert np.all(np.isin(ref_filt[0][ref_filt[0] != 0xFFFFFFFF], np.arange(0, n, 3)))


# ---- int8 row filter in front of the exact re-rank (txh.hip K8b) -------------------------------------------
@pytest.mark.parametrize("case", ["uniform", "duplicates", "scales", "txh"])
def test_rerank_i8_filter_matches_full_rerank(case, monkeypatch):
    """The int8 filter shortlists candidates by a proven bracket of their exact distances; rows, distances
    and tie order must equal the full re-rank's (and the oracle's), also with duplicated rows (exact ties
    at the k-th place), rows of wildly different magnitude, a zero row and m barely above k."""
    n, dim, S = 30000, 64, 16
    rng = np.random.default_rng(11)
    if case == "txh":
        rows, data, stride, ix, oix, kw = H.make_txh_case(40000, 96, 12, 24, seed=41, P=5, kmeans_iters=3, pq_iters=3)
        q = synth.uniform_f32(48, 96, 78)
        o = hip.default_opts()
        o.partitions_to_search, o.pre_reorder_k = 5, 700
    else:
        rows = synth.uniform_f32(n, dim, 5)
        if case == "duplicates":
            rows[1::3] = rows[0::3][: rows[1::3].shape[0]]          # every third row repeats its neighbour
        if case == "scales":
            rows = (rows * np.exp(rng.uniform(-4, 4, (n, 1))).astype(np.float32)).astype(np.float32)
            rows[7] = 0.0
        data, stride = orc.to_strided(rows)
        ixa = trainer.build_ah_index(rows, S, K=16, seed=3, pq_iters=3)
        kw = dict(data=data, n_rows=n, dim=dim, stride=stride, centers=None, leaf_offsets=None, leaf_ids=None,
                  codebook=ixa["codebook"], codes=ixa["codes"], codes_packed4=False, use_residuals=False,
                  partitions_to_search=1, pre_reorder_multiplier=1.0)
        q = synth.uniform_f32(48, dim, 79)
        if case == "scales":
            q = (q * np.float32(3.0)).astype(np.float32)
        o = hip.default_opts()
        o.pre_reorder_k = 900
    monkeypatch.setenv("SCANN_HIP_RERANK_I8", "0")
    plain = hip.txh_create(**kw)
    monkeypatch.setenv("SCANN_HIP_RERANK_I8", "2")
    monkeypatch.setenv("SCANN_HIP_RERANK_I8_MIN", "1")
    filt = hip.txh_create(**kw)
    for k, mm in ((10, o.pre_reorder_k), (10, 45), (1, 300), (40, 170)):
        o.pre_reorder_k = mm
        a = plain.search_batched(q, k, o)
        b = filt.search_batched(q, k, o)
        assert np.array_equal(a[2], b[2]), (case, k, mm)
        assert np.array_equal(bits(a[1]), bits(b[1])), (case, k, mm)
        assert np.array_equal(a[0], b[0]), (case, k, mm)
    if case == "uniform":
        o.pre_reorder_k = 900
        gi, gd, gc = filt.search_batched(q[:6], 10, o)
        for i in range(6):
            oi, od = orc.ah_search_with_reordering(kw["codebook"], kw["codes"], data, stride, q[i], 10, 900)
            H.assert_topk_equal_up_to_ties(gi[i], gd[i], oi, od, what="i8 filter vs oracle q%d" % i)


# ---- L1 / Cosine: DistanceMeasure::distance one pair at a time (brute_force/searcher.rs:131-137) ---------
@pytest.mark.parametrize("measure", [hip.L1, hip.COSINE])
@pytest.mark.parametrize("n,dim,nq", [(3000, 64, 5), (2000, 100, 40), (900, 33, 300), (5000, 128, 64), (700, 7, 3)])
def test_bf_l1_cosine(measure, n, dim, nq):
    """Brute force with the measures the reference scores through DistanceMeasure::distance: all-pairs
    distances bitwise against the oracle (l1_distance_avx2; the cosine kernel with wide 0.7's reduce_add
    order -- parity unpinned by the reference, see scann_hip.h), exact top-k, a zero row and a zero query."""
    rows = synth.uniform_f32(n, dim, 42) - np.float32(0.3)
    rows[11] = 0.0
    data, stride = orc.to_strided(rows)
    index = hip.bf_create(data, n, dim, stride, measure)
    q = synth.uniform_f32(nq, dim, 50 + nq) - np.float32(0.5)
    q[min(2, nq - 1)] = 0.0
    d = hip.bf_distances(index, q)
    for i in range(min(nq, 6)):
        assert np.array_equal(bits(d[i]), bits(orc.one_to_many(q[i], data, stride, n, measure))), i
    idx, dist, cnt = index.search_batched(q, 10)
    for i in range(min(nq, 12)):
        oi, od = orc.bf_search(data, n, dim, stride, measure, q[i], 10)
        assert cnt[i] == oi.size
        H.assert_topk_equal_up_to_ties(idx[i, :oi.size], dist[i, :oi.size], oi, od, what="bf measure %d q%d" % (measure, i))


def test_l1_cosine_reference_known_answers_on_gpu():
    """distance_measures/one_to_one.rs:664-718 through the GPU brute force: L1 (1,2,3)-(4,5,6) = 9;
    cosine distance 0 for equal axis vectors, 1 for orthogonal ones, 1 when a norm is zero."""
    rows = np.array([[4, 5, 6]], np.float32)
    data, stride = orc.to_strided(rows)
    d = hip.bf_distances(hip.bf_create(data, 1, 3, stride, hip.L1), np.array([[1, 2, 3]], np.float32))
    assert d[0, 0] == np.float32(9.0)
    rows = np.array([[1, 0], [0, 1], [0, 0]], np.float32)
    data, stride = orc.to_strided(rows)
    d = hip.bf_distances(hip.bf_create(data, 3, 2, stride, hip.COSINE), np.array([[1, 0]], np.float32))
    assert abs(d[0, 0]) < 1e-6 and abs(d[0, 1] - 1.0) < 1e-6 and d[0, 2] == np.float32(1.0)


# ---- small-batch pipeline (txh.hip "Small batches"): 1..16 queries, three launches ------------------------
def _small_cases():
    yield "txh_residual", None
    yield "txh_byte_codes", None
    yield "ah_flat", None
    for m_ in (hip.SQUARED_L2, hip.DOT_PRODUCT, hip.L2, hip.L1, hip.COSINE):
        yield "partitioned", m_
    for m_ in (hip.SQUARED_L2, hip.DOT_PRODUCT, hip.L2, hip.L1, hip.COSINE):
        yield "bf", m_


@pytest.mark.parametrize("kind", ["txh", "bf"])
def test_small_batch_flags_survive_changing_layouts(kind, monkeypatch):
    """The small-batch host path polls completion flags in pinned memory whose offset moves with nq / k: a flag word
    of one call is a result word (count, index or distance bits) of another.  More than 64 calls alternating nq and k
    on ONE handle (sequence numbers run through small integers such as 10 = a count at k = 10) must each return exactly
    the rows of the staged pipeline."""
    if kind == "txh":
        rows, data, stride, ix, oix, kw = H.make_txh_case(9000, 64, 12, 16, seed=91, P=4, mult=10.0, kmeans_iters=3, pq_iters=3)
        index = hip.txh_create(**kw)
        o = hip.default_opts()
        o.partitions_to_search, o.pre_reorder_k = 4, 100
    else:
        rows = synth.uniform_f32(6000, 64, 92)
        data, stride = orc.to_strided(rows)
        index = hip.bf_create(data, 6000, 64, stride, hip.SQUARED_L2)
        o = None
    q = synth.uniform_f32(16, 64, 93)
    monkeypatch.setenv("SCANN_HIP_SMALL", "0")
    want = {(nq, k): index.search_batched(q[:nq], k, o) for nq in (1, 2, 3, 16) for k in (1, 10, 11)}
    monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
    shapes = [(2, 10), (1, 10), (16, 11), (1, 1), (3, 10), (1, 11), (2, 1), (1, 10)]
    for it in range(96):
        nq, k = shapes[it % len(shapes)]
        got = index.search_batched(q[:nq], k, o)
        w = want[(nq, k)]
        assert np.array_equal(got[2], w[2]), (it, nq, k)
        assert np.array_equal(got[0], w[0]), (it, nq, k)
        assert np.array_equal(bits(got[1]), bits(w[1])), (it, nq, k)


def _small_case(kind, measure):
    """(index, opts, rows, n, oracle) of one searcher kind of the small-batch tests."""
    o = hip.default_opts()
    oracle = None
    if kind == "txh_residual":
        rows, data, stride, ix, oix, kw = H.make_txh_case(9000, 96, 30, 24, seed=61, P=7, mult=12.0, kmeans_iters=3,
                                                          pq_iters=3)
        o.partitions_to_search, o.pre_reorder_k = 7, 120
        n = 9000
        oracle = lambda qv, k: orc.txh_search(oix, qv, k)
        oix.partitions_to_search, oix.pre_reorder_multiplier = 7, 12.0
        index = hip.txh_create(**kw)
    elif kind == "txh_byte_codes":
        rows, data, stride, ix, oix, kw = H.make_txh_case(6000, 64, 16, 8, seed=62, K=256, P=4, mult=10.0,
                                                          kmeans_iters=3, pq_iters=2)
        o.partitions_to_search, o.pre_reorder_k = 4, 100
        n = 6000
        index = hip.txh_create(**kw)
    elif kind == "ah_flat":
        rows, data, stride, ix, kw = H.make_ah_case(12000, 128, 32, seed=63, pq_iters=3)
        o.pre_reorder_k, o.exact_reorder = 150, 1
        n = 12000
        oracle = lambda qv, k: orc.ah_search_with_reordering(ix["codebook"], ix["codes"], data, stride, qv, k, 150)
        index = hip.txh_create(**kw)
    elif kind == "partitioned":
        n, dim = 5000, 100
        rows, data, stride, centers, leaf_off, leaf_ids = _partition_case(n, dim, 23, seed=64)
        index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=centers, leaf_offsets=leaf_off,
                               leaf_ids=leaf_ids, codebook=None, codes=None, partitions_to_search=5,
                               distance_measure=measure)
        oracle = lambda qv, k: orc.scann_search_partitioned(centers, leaf_off, leaf_ids, data, stride, measure, qv, 5, k)
    else:
        n, dim = 7000, 72
        rows = synth.uniform_f32(n, dim, 65) - np.float32(0.4)
        rows[5] = 0.0
        data, stride = orc.to_strided(rows)
        index = hip.bf_create(data, n, dim, stride, measure)
        oracle = lambda qv, k: orc.bf_search(data, n, dim, stride, measure, qv, k)
        o = None
    return index, o, rows, n, oracle


def _wide_cases():
    yield "txh_residual", None
    yield "txh_byte_codes", None
    yield "ah_flat", None


@pytest.mark.parametrize("kind,measure", list(_wide_cases()))
def test_wide_pipeline_matches_staged_pipeline(kind, measure, monkeypatch):
    """Calls of <= 4 queries over long streams run the wide pipeline (wide_scan_kernel ->
    wide_filter_kernel -> wide_final_kernel): SCANN_HIP_WIDE=2 forces it onto the small test indexes, so that streams
    shorter than pre_reorder_k, k above the stream, one-point groups and every searcher kind pass through it.  Rows,
    distance bits and counts must equal the staged pipeline's (SCANN_HIP_SMALL=0), which the batched parity tests pin
    to the oracle; a few queries are checked against the oracle directly."""
    index, o, rows, n, oracle = _small_case(kind, measure)
    index.enable_timing(True)
    dim = index.dimensionality()
    q = synth.uniform_f32(8, dim, 67)
    q[2] = rows[321]
    variants = [None]
    if kind in ("txh_residual", "ah_flat"):
        variants = [None, (40, 1), (5000, 1), (0, 0)]   # (pre_reorder_k, exact_reorder): short lists, m above the stream, no re-ordering
    for var in variants:
        if var is not None:
            o.pre_reorder_k, o.exact_reorder = var
        for nq in (1, 3, 4):
            for k in (10, 1, 64):
                monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
                monkeypatch.setenv("SCANN_HIP_WIDE", "2")
                a = index.search_batched(q[:nq], k, o)
                assert index.last_kernel_ms()[1] == "wide_scan_kernel", (kind, var, nq, k)
                monkeypatch.setenv("SCANN_HIP_SMALL", "0")
                b = index.search_batched(q[:nq], k, o)
                assert np.array_equal(a[2], b[2]), (kind, var, nq, k)
                assert np.array_equal(bits(a[1]), bits(b[1])), (kind, var, nq, k)
                assert np.array_equal(a[0], b[0]), (kind, var, nq, k)
        if oracle is not None and var is None:
            monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
            gi, gd, gc = index.search_batched(q[:3], 10, o)
            for i in range(3):
                oi, od = oracle(q[i], 10)
                assert gc[i] == oi.size
                H.assert_topk_equal_up_to_ties(gi[i, :oi.size], gd[i, :oi.size], oi, od, what="%s wide q%d" % (kind, i))
    # search_with_filter (an allow-bitmap) and the token outputs through the wide pipeline (device buffers, no pinned staging)
    o.pre_reorder_k, o.exact_reorder = 120, 1
    allow = hip.allow_bitmap(n, np.arange(1, n, 5))
    res = []
    for env in (None, "0"):
        monkeypatch.setenv("SCANN_HIP_WIDE", "2")
        if env is None:
            monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
        else:
            monkeypatch.setenv("SCANN_HIP_SMALL", env)
        P = 1 if kind == "ah_flat" else o.partitions_to_search
        tok = np.zeros((3, P), np.uint32); tokd = np.zeros((3, P), np.float32)
        o2 = hip.default_opts()
        o2.partitions_to_search, o2.pre_reorder_k, o2.exact_reorder = o.partitions_to_search, 120, 1
        o2.tokens, o2.token_dists = hip.ptr(tok, hip.u32p), hip.ptr(tokd, hip.f32p)
        fa = index.search_batched(q[:3], 10, o2, allow=allow)
        res.append((fa, tok.copy(), tokd.copy()))
    (fa, ta, tda), (fb, tb, tdb) = res
    assert np.array_equal(fa[0], fb[0]) and np.array_equal(bits(fa[1]), bits(fb[1])) and np.array_equal(fa[2], fb[2])
    assert np.all(fa[0][fa[0] != 0xFFFFFFFF] % 5 == 1)
    assert np.array_equal(ta, tb) and np.array_equal(bits(tda), bits(tdb))
    # five queries are above the wide pipeline's batch: the small-batch pipeline takes them
    monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
    index.search_batched(q[:5], 10, o)
    assert index.last_kernel_ms()[1] == "small_scan_kernel"


def _random_tree_index(sizes, dim, S, seed, rows=None):
    """A tree index with the given leaf sizes: random centres, random 4-bit codes and codebook (the pipelines are
    compared with each other: nothing here needs a trained quantiser)."""
    rng = np.random.default_rng(seed)
    n = int(sum(sizes))
    if rows is None:
        rows = synth.uniform_f32(n, dim, seed)
    data, stride = orc.to_strided(rows)
    L = len(sizes)
    centers = np.ascontiguousarray(rows[rng.choice(n, L, replace=False)])
    leaf_off = np.zeros(L + 1, np.uint32)
    leaf_off[1:] = np.cumsum(np.asarray(sizes, np.uint32))
    leaf_ids = rng.permutation(n).astype(np.uint32)
    codebook = rng.random((S, 16, dim // S), dtype=np.float32)
    codes = rng.integers(0, 16, (n, S), dtype=np.uint8)
    kw = dict(data=data, n_rows=n, dim=dim, stride=stride, centers=centers, leaf_offsets=leaf_off, leaf_ids=leaf_ids,
              codebook=codebook, codes=codes, codes_packed4=False, use_residuals=True, partitions_to_search=1,
              pre_reorder_multiplier=1.0)
    return rows, centers, kw


def test_wide_pipeline_uneven_leaves_and_overflow(monkeypatch):
    """The wide pipeline sizes its groups of stream positions from each query's OWN stream (a leaf of 40000 points next
    to leaves of 3000).  Its compact candidate arrays overflow only when thousands of points tie at the pivot's
    distance -- a dataset of few distinct code rows: count row 0xFFFFFFFF in the pinned buffer, and the host entry
    repeats the call on the batched pipeline -- same rows either way."""
    rows, centers, kw = _random_tree_index([40000, 3000, 3000, 3000, 3000, 3000], 32, 8, seed=71)
    index = hip.txh_create(**kw)
    index.enable_timing(True)
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = 1, 300
    q = np.ascontiguousarray(centers + np.float32(0.001))
    monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
    monkeypatch.delenv("SCANN_HIP_WIDE", raising=False)   # (40000 points, leaves of 9000 on average: the default rule)

    def both(ix_, qq, k):
        a = ix_.search_batched(qq, k, o)
        name = ix_.last_kernel_ms()[1]
        monkeypatch.setenv("SCANN_HIP_SMALL", "0")
        b = ix_.search_batched(qq, k, o)
        monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
        return name

    for i in range(6):
        assert both(index, q[i:i + 1], 10) == "wide_scan_kernel", i
    assert both(index, q[:4], 10) == "wide_scan_kernel"       # long and short streams in one call
    # few distinct rows and code rows: ties everywhere (approximate and exact distances); the order is (exact, merge key)
    base = synth.uniform_f32(37, 32, 72)
    for ncodes, m, wide in ((53, 200, True), (53, 2000, True), (5, 200, False)):
        rows2 = np.ascontiguousarray(base[np.arange(36000) % 37])
        rows2, centers2, kw2 = _random_tree_index([36000], 32, 8, seed=73, rows=rows2)
        kw2["codes"] = np.ascontiguousarray(kw2["codes"][np.arange(36000) % ncodes])
        index2 = hip.txh_create(**kw2)
        index2.enable_timing(True)
        o.pre_reorder_k = m
        name = both(index2, q[:2], 25)
        # 5 code rows: 7200 points tie at the pivot, the compact arrays (2 m + 1024) overflow, the call is repeated
        assert (name == "wide_scan_kernel") == wide, (ncodes, m, name)


def test_wide_pipeline_many_leaves(monkeypatch):
    """4096 leaves: the leaf selection inside wide_scan_kernel needs more than 64 KB of dynamic LDS next to the kernel's
    static arrays (the LDS attribute must leave room for both); rows equal the staged pipeline's."""
    rows, centers, kw = _random_tree_index([15] * 4096, 32, 8, seed=81)
    kw["partitions_to_search"] = 40
    index = hip.txh_create(**kw)
    index.enable_timing(True)
    o = hip.default_opts()
    o.partitions_to_search, o.pre_reorder_k = 40, 100
    q = synth.uniform_f32(4, 32, 82)
    for nq in (1, 4):
        monkeypatch.setenv("SCANN_HIP_WIDE", "2")
        monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
        a = index.search_batched(q[:nq], 10, o)
        assert index.last_kernel_ms()[1] == "wide_scan_kernel"
        monkeypatch.setenv("SCANN_HIP_SMALL", "0")
        b = index.search_batched(q[:nq], 10, o)
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
    # the same for the one-launch small-batch kernel (80 KB of static arrays) on a Partitioned index of 4096 leaves
    monkeypatch.delenv("SCANN_HIP_WIDE", raising=False)
    part = hip.txh_create(data=kw["data"], n_rows=kw["n_rows"], dim=32, stride=kw["stride"], centers=centers,
                          leaf_offsets=kw["leaf_offsets"], leaf_ids=kw["leaf_ids"], codebook=None, codes=None,
                          partitions_to_search=40, distance_measure=hip.SQUARED_L2)
    for nq in (1, 3):
        monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
        a = part.search_batched(q[:nq], 10)
        monkeypatch.setenv("SCANN_HIP_SMALL", "0")
        b = part.search_batched(q[:nq], 10)
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])


def test_wide_pipeline_on_the_device_entry_points():
    """scann_hip_search_batched_device with 1-4 queries over a long stream: the wide pipeline on the caller's stream
    (a workspace per stream), rows equal to the host entry point's; calls on two streams back to back do not share
    the dense key lists or the compact arrays.  An overflow of the compact arrays (few distinct code rows) is reported
    by scann_hip_index_last_device_status with count rows 0 -- the host entry repeats such a call by itself."""
    import ctypes

    import torch
    dev = torch.device("cuda:0")
    k = 10
    rows, data, stride, ix, kw = H.make_ah_case(70000, 64, 16, seed=74, pq_iters=3)
    index = hip.txh_create(**kw)
    o = hip.default_opts()
    o.pre_reorder_k = 1500
    L = hip.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    nqs = [1, 4]
    qs = [synth.uniform_f32(nq, 64, 800 + i) for i, nq in enumerate(nqs)]
    want = [index.search_batched(q, k, o) for q in qs]
    qd = [torch.from_numpy(q).to(dev) for q in qs]
    outs = [(torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
             torch.empty((nq,), dtype=torch.int32, device=dev)) for nq in nqs]
    index.enable_timing(True)
    for rep in range(3):
        for oi_, od_, oc_ in outs:
            oi_.fill_(-1); od_.zero_(); oc_.zero_()
        torch.cuda.synchronize()
        for s in (0, 1, 0, 1):
            hip.check(L.scann_hip_search_batched_device(index.h, p(qd[s]), nqs[s], 64, k, ctypes.byref(o), p(outs[s][0]),
                                                        p(outs[s][1]), p(outs[s][2]), ctypes.c_void_p(streams[s].cuda_stream)))
        for s in (0, 1):
            hip.check(L.scann_hip_index_last_device_status(index.h, ctypes.c_void_p(streams[s].cuda_stream)))
        torch.cuda.synchronize()
        for s in (0, 1):
            wi, wd, wc = want[s]
            assert np.array_equal(outs[s][2].cpu().numpy().view(np.uint32), wc), (rep, s)
            assert np.array_equal(bits(outs[s][1].cpu().numpy()), bits(wd)), (rep, s)
            assert np.array_equal(outs[s][0].cpu().numpy().view(np.uint32), wi), (rep, s)
    for i in range(4):
        oi, od = orc.ah_search_with_reordering(ix["codebook"], ix["codes"], data, stride, qs[1][i], k, 1500)
        H.assert_topk_equal_up_to_ties(outs[1][0][i].cpu().numpy().view(np.uint32), outs[1][1][i].cpu().numpy(), oi, od,
                                       what="device wide q%d" % i)
    # overflow: 5 distinct code rows, thousands of points tied at the pivot
    base = synth.uniform_f32(37, 32, 72)
    rows2 = np.ascontiguousarray(base[np.arange(36000) % 37])
    rows2, centers2, kw2 = _random_tree_index([36000], 32, 8, seed=73, rows=rows2)
    kw2["codes"] = np.ascontiguousarray(kw2["codes"][np.arange(36000) % 5])
    index2 = hip.txh_create(**kw2)
    o2 = hip.default_opts()
    o2.partitions_to_search, o2.pre_reorder_k = 1, 200
    q2 = torch.from_numpy(synth.uniform_f32(1, 32, 75)).to(dev)
    o_i = torch.empty((1, k), dtype=torch.int32, device=dev); o_d = torch.empty((1, k), dtype=torch.float32, device=dev)
    o_c = torch.full((1,), 7, dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(streams[0].cuda_stream)
    hip.check(L.scann_hip_search_batched_device(index2.h, p(q2), 1, 32, k, ctypes.byref(o2), p(o_i), p(o_d), p(o_c), st))
    status = L.scann_hip_index_last_device_status(index2.h, st)
    torch.cuda.synchronize()
    assert status == hip.RESOURCE_EXHAUSTED, status
    assert int(o_c.cpu().numpy()[0]) == 0


@pytest.mark.parametrize("kind,measure", list(_small_cases()))
def test_small_batch_pipeline_matches_staged_pipeline(kind, measure, monkeypatch):
    """Calls of <= 16 queries run select_leaves (inline centroid scoring) -> small_scan -> small_finish -- as ONE
    launch (small_fused_kernel) when the grid is small, as three otherwise (SCANN_HIP_FUSED=0 forces three) -- with
    pinned zero-copy staging; SCANN_HIP_SMALL=0 sends the same call down the staged pipeline the batched
    parity tests pin to the oracle. Rows, distance bits and counts must be identical for every searcher
    kind and measure, for 1, 7 and 16 queries, for k above the scanned stream (short results), and with an
    allow-bitmap; a few queries are also checked against the oracle directly."""
    index, o, rows, n, oracle = _small_case(kind, measure)
    dim = index.dimensionality()
    q = synth.uniform_f32(16, dim, 66) - np.float32(0.2 if kind == "bf" else 0.0)
    q[4] = rows[123]
    allow = hip.allow_bitmap(n, np.arange(1, n, 5))
    for nq in (1, 7, 16):
        for k in (10, 1, 64):
            monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
            a = index.search_batched(q[:nq], k, o)              # one launch (small_fused_kernel) when the grid is small
            monkeypatch.setenv("SCANN_HIP_FUSED", "0")
            a3 = index.search_batched(q[:nq], k, o)             # the three-launch form
            monkeypatch.delenv("SCANN_HIP_FUSED", raising=False)
            monkeypatch.setenv("SCANN_HIP_SMALL", "0")
            b = index.search_batched(q[:nq], k, o)              # the staged pipeline
            for x, what in ((a, "fused"), (a3, "three launches")):
                assert np.array_equal(x[2], b[2]), (kind, nq, k, what)
                assert np.array_equal(bits(x[1]), bits(b[1])), (kind, nq, k, what)
                assert np.array_equal(x[0], b[0]), (kind, nq, k, what)
        if oracle is not None:
            monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
            gi, gd, gc = index.search_batched(q[:nq], 10, o)
            for i in range(min(nq, 3)):
                oi, od = oracle(q[i], 10)
                assert gc[i] == oi.size
                H.assert_topk_equal_up_to_ties(gi[i, :oi.size], gd[i, :oi.size], oi, od, what="%s small q%d" % (kind, i))
    if kind not in ("bf", "partitioned"):   # search_with_filter through the small pipeline (the partitioned
        monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)   # and brute-force searchers take no filter)
        fa = index.search_batched(q[:9], 10, o, allow=allow)
        o.allow_bitmap, o.allow_bitmap_bits = None, 0
        monkeypatch.setenv("SCANN_HIP_SMALL", "0")
        fb = index.search_batched(q[:9], 10, o, allow=allow)
        o.allow_bitmap, o.allow_bitmap_bits = None, 0
        assert np.array_equal(fa[0], fb[0]) and np.array_equal(bits(fa[1]), bits(fb[1])) and np.array_equal(fa[2], fb[2])
        assert np.all(fa[0][fa[0] != 0xFFFFFFFF] % 5 == 1)
    if kind != "bf":   # the token outputs through the small pipeline
        if kind != "ah_flat":
            P = o.partitions_to_search or 5
            toks = []
            for env in (None, "0"):
                if env is None:
                    monkeypatch.delenv("SCANN_HIP_SMALL", raising=False)
                else:
                    monkeypatch.setenv("SCANN_HIP_SMALL", env)
                tok = np.zeros((9, P), np.uint32); tokd = np.zeros((9, P), np.float32)
                o2 = hip.default_opts()
                o2.partitions_to_search, o2.pre_reorder_k, o2.exact_reorder = P, o.pre_reorder_k, o.exact_reorder
                o2.tokens, o2.token_dists = hip.ptr(tok, hip.u32p), hip.ptr(tokd, hip.f32p)
                index.search_batched(q[:9], 10, o2)
                toks.append((tok, tokd))
            assert np.array_equal(toks[0][0], toks[1][0]) and np.array_equal(bits(toks[0][1]), bits(toks[1][1]))


# ---- FP8: the reference's codec, its one-to-many kernels, and the FP8 row store of the re-rank filter ------
@pytest.mark.parametrize("fmt", [hip.FP8_E4M3, hip.FP8_E5M2])
def test_fp8_codec_bit_exact(fmt):
    """quantization/fp8.rs:80-268 on the device against the oracle restatement: every f32 exponent band,
    zeros, infinities, NaN, the mantissa-carry wrap, flush-to-zero, scales from calibrate_scale; all 256
    codes decoded (the reference's own 'subnormal' rule included)."""
    rng = np.random.default_rng(5 + fmt)
    vals = np.concatenate([
        (rng.standard_normal(4000) * np.exp(rng.uniform(-14, 14, 4000))).astype(np.float32),
        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.9375, -1.9375, 448.0, 449.0, 255.9, 256.0, 2.0 ** -6, 2.0 ** -7,
                  2.0 ** -14, 2.0 ** -15, 57344.0, 65536.0, 1e30, -1e30, 1e-30], np.float32),
        np.ldexp(np.float32(1.0), np.arange(-20, 20)).astype(np.float32)])
    for scale in (1.0, float(orc.fp8_calibrate_scale(3.7, fmt)), 0.01):
        got = hip.fp8_quantize(vals, scale, fmt)
        want = orc.fp8_quantize(vals, scale, fmt)
        assert np.array_equal(got, want), (fmt, scale, np.flatnonzero(got != want)[:5])
    codes = np.arange(256, dtype=np.uint8)
    for scale in (1.0, 224.0):
        assert np.array_equal(bits(hip.fp8_dequantize(codes, scale, fmt)), bits(orc.fp8_dequantize(codes, scale, fmt)))
    # reference unit tests on the device (fp8.rs:278-343)
    v4 = np.array([1.0, 2.0, 3.0, 4.0], np.float32)
    assert np.all(np.abs(hip.fp8_dequantize(hip.fp8_quantize(v4)) - v4) < 0.5)


@pytest.mark.parametrize("measure", [hip.SQUARED_L2, hip.DOT_PRODUCT])
@pytest.mark.parametrize("n,dim,stride", [(1, 4, 4), (3000, 128, 128), (700, 50, 64), (257, 7, 9)])
def test_fp8_one_to_many_bit_exact(measure, n, dim, stride):
    """one_to_many_fp8_float_{squared_l2,dot_product} (one_to_many_asymmetric.rs:327-377): arbitrary code
    bytes (every one of the 256 codes occurs), padded strides, sequential f32 sums bit for bit."""
    rng = np.random.default_rng(n + dim)
    db = rng.integers(0, 256, (n, stride), dtype=np.uint8)
    if n == 1:   # the reference's test_simd_operations: query == database == [1, 2, 3, 4]
        q = np.array([1.0, 2.0, 3.0, 4.0], np.float32)
        db = hip.fp8_quantize(q).reshape(1, 4)
    else:
        q = (rng.standard_normal(dim) * 3).astype(np.float32)
    got = hip.fp8_distances(q, db, stride, n, measure)
    want = orc.one_to_many_fp8(q, db, stride, n, measure)
    assert np.array_equal(bits(got), bits(want))
    if n == 1:
        assert (abs(-got[0] - 30.0) < 1.0) if measure == hip.DOT_PRODUCT else (got[0] < 1.0)
    with pytest.raises(hip.ScannError) as e:
        hip.fp8_distances(q, db, stride, n, hip.L1)
    assert e.value.code == hip.UNIMPLEMENTED


@pytest.mark.parametrize("case", ["uniform", "duplicates", "scales"])
def test_rerank_fp8_filter_matches_full_rerank(case, monkeypatch):
    """SCANN_HIP_RERANK_STORE=fp8: the re-rank filter over rows stored with the reference's E4M3 codec and a
    per-row calibrate_scale.  Same proof as the int8 store, so rows, distance bits and tie order must equal
    the full re-rank's -- with duplicated rows, rows spanning e^+-4 in magnitude and a zero row."""
    n, dim, S = 30000, 64, 16
    rng = np.random.default_rng(12)
    rows = synth.uniform_f32(n, dim, 6)
    if case == "duplicates":
        rows[1::3] = rows[0::3][: rows[1::3].shape[0]]
    if case == "scales":
        rows = (rows * np.exp(rng.uniform(-4, 4, (n, 1))).astype(np.float32)).astype(np.float32)
        rows[7] = 0.0
    data, stride = orc.to_strided(rows)
    ixa = trainer.build_ah_index(rows, S, K=16, seed=3, pq_iters=3)
    kw = dict(data=data, n_rows=n, dim=dim, stride=stride, centers=None, leaf_offsets=None, leaf_ids=None,
              codebook=ixa["codebook"], codes=ixa["codes"], codes_packed4=False, use_residuals=False,
              partitions_to_search=1, pre_reorder_multiplier=1.0)
    q = synth.uniform_f32(48, dim, 80)
    o = hip.default_opts()
    monkeypatch.setenv("SCANN_HIP_RERANK_I8", "0")
    plain = hip.txh_create(**kw)
    monkeypatch.setenv("SCANN_HIP_RERANK_I8", "2")
    monkeypatch.setenv("SCANN_HIP_RERANK_I8_MIN", "1")
    monkeypatch.setenv("SCANN_HIP_RERANK_STORE", "fp8")
    filt = hip.txh_create(**kw)
    for k, mm in ((10, 900), (10, 45), (1, 300), (40, 170)):
        o.pre_reorder_k = mm
        a = plain.search_batched(q, k, o)
        b = filt.search_batched(q, k, o)
        assert np.array_equal(a[2], b[2]), (case, k, mm)
        assert np.array_equal(bits(a[1]), bits(b[1])), (case, k, mm)
        assert np.array_equal(a[0], b[0]), (case, k, mm)


# ---- leaf selection among many leaves (L > 4096: the keys of all leaves staged in > 64 KB of LDS) ---------
@pytest.mark.parametrize("L,dim,P,kind", [(5000, 16, 10, "rand"), (9000, 32, 200, "rand"), (16384, 8, 512, "rand"),
                                          (6000, 16, 37, "dups"), (5000, 8, 10, "all_equal"), (7000, 16, 600, "rand")])
def test_partition_many_leaves(L, dim, P, kind):
    """TreePartitioner::partition (partitioning/tree_partitioner.rs:196-229) with thousands of leaves: tokens
    and distance bits against the oracle.  'dups': every centre occurs ~8 times (ties at the P-th place are
    broken by leaf index, as the stable sort does); 'all_equal': all centres identical (every key ties on the
    distance); P = 600: the sort-everything form; a NaN query (every distance NaN) must also come out in leaf
    order."""
    rng = np.random.default_rng(L + P)
    if kind == "rand":
        centers = rng.standard_normal((L, dim)).astype(np.float32)
    elif kind == "dups":
        base = rng.standard_normal((L // 8 + 1, dim)).astype(np.float32)
        centers = base[rng.integers(0, base.shape[0], L)]
    else:
        centers = np.tile(rng.standard_normal((1, dim)).astype(np.float32), (L, 1))
    n = L                      # one row per leaf: the partitioner only needs the centres and the leaf sizes
    rows = centers.copy()
    data, stride = orc.to_strided(rows)
    S = 8
    codebook = rng.standard_normal((S, 16, dim // S)).astype(np.float32)
    codes = rng.integers(0, 16, (n, S), dtype=np.uint8)
    index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=centers,
                           leaf_offsets=np.arange(L + 1, dtype=np.uint32), leaf_ids=np.arange(n, dtype=np.uint32),
                           codebook=codebook, codes=codes, codes_packed4=False, use_residuals=True,
                           partitions_to_search=P, pre_reorder_multiplier=3.0)
    q = rng.standard_normal((40, dim)).astype(np.float32)
    q[7] = centers[11]         # an exact hit (distance 0, with its duplicates)
    q[9] = np.nan
    tok, dist, cnt = hip.txh_partition(index, q, P)
    for i in range(q.shape[0]):
        ot, od = orc.partition(centers, q[i], P)
        assert cnt[i] == ot.size, (i, cnt[i], ot.size)
        if i != 9:
            assert np.array_equal(bits(dist[i, :ot.size]), bits(od)), i
        assert np.array_equal(tok[i, :ot.size], ot), i


# ---- the filter bound under massive ties of the approximate distance ---------------------------------------
@pytest.mark.parametrize("mode", ["ah", "txh"])
def test_threshold_ties_do_not_overflow_device_path(mode, monkeypatch):
    """Coarse codes over clustered data give thousands of points the SAME approximate distance (here: every
    row occurs 150 times).  The sampled filter bound is the J-th smallest sample KEY (distance, stream position),
    not the distance alone -- with the distance alone the whole tie group passes and the candidate buffers of
    the device entry point overflow (found with tools/sweep_txh.py on a 12.5M x 96 index).  The device path
    must report no miss, and its rows must equal the host entry point's (which would have retried)."""
    import ctypes
    import torch
    dim, S, copies, distinct = 32, 8, 3000, 60
    rng = np.random.default_rng(21)
    base = rng.standard_normal((distinct, dim)).astype(np.float32)
    rows = base[rng.permutation(np.repeat(np.arange(distinct), copies))]
    n = rows.shape[0]
    data, stride = orc.to_strided(rows)
    if mode == "ah":
        ixa = trainer.build_ah_index(rows, S, K=16, seed=3, pq_iters=3)
        kw = dict(data=data, n_rows=n, dim=dim, stride=stride, centers=None, leaf_offsets=None, leaf_ids=None,
                  codebook=ixa["codebook"], codes=ixa["codes"], codes_packed4=False, use_residuals=False,
                  partitions_to_search=1, pre_reorder_multiplier=1.0)
        P = 1
    else:
        ixt = trainer.build_txh_index(rows, 12, S, K=16, use_residuals=True, seed=5, kmeans_iters=3, pq_iters=3)
        kw = dict(data=data, n_rows=n, dim=dim, stride=stride, centers=ixt["centers"], leaf_offsets=ixt["leaf_off"],
                  leaf_ids=ixt["leaf_ids"], codebook=ixt["codebook"], codes=ixt["codes"], codes_packed4=False,
                  use_residuals=True, partitions_to_search=4, pre_reorder_multiplier=3.0)
        P = 4
    index = hip.txh_create(**kw)
    nq, k = 96, 10
    q = (base[rng.integers(0, distinct, nq)] + np.float32(0.05) * rng.standard_normal((nq, dim))).astype(np.float32)
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(q).to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = hip.load()
    for m in (300, 2000):
        o = hip.default_opts()
        o.partitions_to_search, o.pre_reorder_k, o.exact_reorder = P, m, 1
        want = index.search_batched(q, k, o)
        oi_t = torch.empty((nq, k), dtype=torch.int32, device=dev)
        od_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
        oc_t = torch.empty((nq,), dtype=torch.int32, device=dev)
        hip.check(L.scann_hip_search_batched_device(index.h, p(qd), nq, dim, k, ctypes.byref(o), p(oi_t), p(od_t),
                                                    p(oc_t), sp))
        hip.check(L.scann_hip_index_last_device_status(index.h, sp))        # no threshold / buffer miss
        assert np.array_equal(oc_t.cpu().numpy().view(np.uint32), want[2])
        assert np.array_equal(bits(od_t.cpu().numpy()), bits(want[1]))
    # the same call with the bound on the distance alone: the tie group at the bound floods the buffer
    monkeypatch.setenv("SCANN_HIP_THR_TIES", "0")
    hip.check(L.scann_hip_search_batched_device(index.h, p(qd), nq, dim, k, ctypes.byref(o), p(oi_t), p(od_t), p(oc_t), sp))
    with pytest.raises(hip.ScannError) as e:
        hip.check(L.scann_hip_index_last_device_status(index.h, sp))
    assert e.value.code == hip.RESOURCE_EXHAUSTED


# ---- device entry points: one workspace per caller stream ---------------------------------------------------
@pytest.mark.parametrize("kind", ["ah", "txh", "bf"])
def test_device_calls_on_two_streams_match_oracle(kind):
    """scann_hip_search_batched_device binds a workspace to the caller's stream (include/scann_hip.h "device entry
    points and streams"): calls issued back to back on two streams, with no synchronisation between them, must not share
    LUTs, candidate lists or counters.  Every row is compared with the host entry point (pinned to the oracle by the
    tests above) and a few with the oracle directly; a third stream takes over the least recently used workspace."""
    import ctypes

    import torch
    dev = torch.device("cuda:0")
    k = 10
    o = hip.default_opts()
    if kind == "ah":
        rows, data, stride, ix, kw = H.make_ah_case(70000, 64, 16, seed=71, pq_iters=3)
        index = hip.txh_create(**kw)
        o.pre_reorder_k = 600
        dim = 64
        oracle = lambda qv: orc.ah_search_with_reordering(ix["codebook"], ix["codes"], data, stride, qv, k, 600)
    elif kind == "txh":
        rows, data, stride, ix, oix, kw = H.make_txh_case(66000, 96, 24, 24, seed=72, P=5, mult=30.0, kmeans_iters=3, pq_iters=3)
        index = hip.txh_create(**kw)
        o.partitions_to_search, o.pre_reorder_k = 5, 300
        dim = 96
        oracle = lambda qv: orc.txh_search(oix, qv, k)
    else:
        rows = synth.uniform_f32(20000, 64, 73)
        data, stride = orc.to_strided(rows)
        index = hip.bf_create(data, 20000, 64, stride, hip.SQUARED_L2)
        dim = 64
        o = None
        oracle = lambda qv: orc.bf_search(data, 20000, 64, stride, hip.SQUARED_L2, qv, k)
    L = hip.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    nqs = [192, 77, 130]
    qs = [synth.uniform_f32(nq, dim, 700 + i) for i, nq in enumerate(nqs)]
    want = [index.search_batched(q, k, o) for q in qs]
    qd = [torch.from_numpy(q).to(dev) for q in qs]
    outs = [(torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
             torch.empty((nq,), dtype=torch.int32, device=dev)) for nq in nqs]
    torch.cuda.synchronize()
    ob = ctypes.byref(o) if o is not None else None
    for rep in range(4):
        order = [0, 1] if rep < 2 else [2, 0, 1, 2]      # rep >= 2: three streams over two workspaces
        for oi_, od_, oc_ in outs:
            oi_.fill_(-1); od_.zero_(); oc_.zero_()
        torch.cuda.synchronize()
        for s in order:
            hip.check(L.scann_hip_search_batched_device(index.h, p(qd[s]), nqs[s], dim, k, ob, p(outs[s][0]), p(outs[s][1]),
                                                        p(outs[s][2]), ctypes.c_void_p(streams[s].cuda_stream)))
        for s in set(order):
            hip.check(L.scann_hip_index_last_device_status(index.h, ctypes.c_void_p(streams[s].cuda_stream)))
        torch.cuda.synchronize()
        for s in set(order):
            wi, wd, wc = want[s]
            assert np.array_equal(outs[s][2].cpu().numpy().view(np.uint32), wc), (kind, rep, s)
            assert np.array_equal(bits(outs[s][1].cpu().numpy()), bits(wd)), (kind, rep, s)
            assert np.array_equal(outs[s][0].cpu().numpy().view(np.uint32), wi), (kind, rep, s)
    for s in range(3):
        for i in range(0, nqs[s], 37):
            wi, wd = oracle(qs[s][i])[:2]
            H.assert_topk_equal_up_to_ties(outs[s][0][i].cpu().numpy().view(np.uint32)[:wi.size],
                                           outs[s][1][i].cpu().numpy()[:wi.size], wi, wd, what="%s s%d q%d" % (kind, s, i))


# ---- Searcher::search_batched_with_params: one num_neighbors per query ------------------------------------------
@pytest.mark.parametrize("kind", ["txh", "ah", "bf"])
def test_search_batched_with_params_per_query_k(kind):
    """searcher.rs:148-186 / tree_x_hybrid/mod.rs:383-409: every query of a batch carries its own SearchParameters, of
    which the searchers on this path read num_neighbors.  Query i must get exactly the rows of search(query_i, k_i) --
    with the pre-reorder candidate count k_i * multiplier that goes with ITS k -- whatever the other queries ask for."""
    if kind == "txh":
        rows, data, stride, ix, oix, kw = H.make_txh_case(5000, 64, 12, 16, seed=81, P=4, mult=3.0, kmeans_iters=3, pq_iters=3)
        index = hip.txh_create(**kw)
        single = lambda qv, k: orc.txh_search(oix, qv, k)
    elif kind == "ah":
        rows, data, stride, ix, kw = H.make_ah_case(4000, 64, 16, seed=82, pq_iters=3)
        kw["pre_reorder_multiplier"] = 4.0
        index = hip.txh_create(**kw)
        single = lambda qv, k: orc.ah_search_with_reordering(ix["codebook"], ix["codes"], data, stride, qv, k, orc.pre_reorder_k(k, 4.0))
    else:
        rows = synth.uniform_f32(3000, 48, 83)
        data, stride = orc.to_strided(rows)
        index = hip.bf_create(data, 3000, 48, stride, hip.DOT_PRODUCT)
        single = lambda qv, k: orc.bf_search(data, 3000, 48, stride, hip.DOT_PRODUCT, qv, k)
    dim = index.dimensionality()
    q = synth.uniform_f32(23, dim, 84)
    ks = np.array([10, 1, 5, 10, 37, 0, 5, 64, 10, 2, 3, 10, 1, 100, 7, 10, 5, 5, 10, 20, 1, 10, 9], np.uint32)
    idx, dist, cnt = index.search_batched_with_params(q, ks)
    assert idx.shape[1] == 100
    for i in range(q.shape[0]):
        k = int(ks[i])
        if k == 0:
            assert cnt[i] == 0
            continue
        wi, wd = single(q[i], k)[:2]
        assert cnt[i] == wi.size, (i, k, cnt[i], wi.size)
        H.assert_topk_equal_up_to_ties(idx[i, :cnt[i]], dist[i, :cnt[i]], wi, wd, what="%s q%d k=%d" % (kind, i, k))
        assert np.all(idx[i, cnt[i]:] == 0xFFFFFFFF) and np.all(np.isinf(dist[i, cnt[i]:]))
