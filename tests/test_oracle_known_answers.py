"""Pins the CPU oracle against every known-answer unit test the reference holds
for the hot path.  Each test names the reference test it
transcribes (paths under /root/reference).  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as orc


def approx(a, b, tol=1e-5):
    return abs(a - b) < tol


# ---- f32 kernels ---------------------------------------------------------------
def test_dot_product_dispatch():  # src/simd/tests.rs:122-131, src/simd/x86.rs:485-497
    a = [1, 2, 3, 4, 5, 6, 7, 8]
    b = [1] * 8
    assert approx(orc.dot_product_avx2(a, b), 36.0)
    assert approx(orc.dot_product_portable(a, b), 36.0)


def test_dot_product_large():  # src/simd/tests.rs:133-143
    size = 128
    a = np.arange(size, dtype=np.float32) * np.float32(0.1)
    b = (size - np.arange(size)).astype(np.float32) * np.float32(0.1)
    expected = np.float32(0)
    for x, y in zip(a, b):
        expected = np.float32(expected + np.float32(x * y))
    assert abs(orc.dot_product_avx2(a, b) - float(expected)) < 0.1


def test_squared_l2_dispatch():  # src/simd/tests.rs:145-155, src/simd/x86.rs:499-513
    a = [1, 2, 3, 4, 5, 6, 7, 8]
    b = [2, 3, 4, 5, 6, 7, 8, 9]
    assert approx(orc.squared_l2_avx2(a, b), 8.0)
    assert approx(orc.squared_l2_portable(a, b), 8.0)
    assert approx(orc.squared_l2_sequential(a, b), 8.0)


def test_squared_l2_large():  # src/simd/tests.rs:157-167
    a = np.arange(256, dtype=np.float32)
    assert abs(orc.squared_l2_avx2(a, a + 1) - 256.0) < 0.01


def test_one_to_many_dot_product_simd():  # src/simd/tests.rs:189-208
    q = np.arange(1, 9, dtype=np.float32)
    db = np.concatenate([np.full(8, 1.0), np.full(8, 2.0), np.full(8, 0.5)]).astype(np.float32)
    r = orc.one_to_many(q, db, 8, 3, orc.DOT_PRODUCT)
    assert approx(r[0], -36.0) and approx(r[1], -72.0) and approx(r[2], -18.0)


def test_one_to_many_squared_l2_simd():  # src/simd/tests.rs:210-231
    q = np.arange(1, 9, dtype=np.float32)
    db = np.concatenate([q, np.zeros(8), q + 1]).astype(np.float32)
    r = orc.one_to_many(q, db, 8, 3, orc.SQUARED_L2)
    assert approx(r[0], 0.0) and approx(r[1], 204.0) and approx(r[2], 8.0)


def test_one_to_many_small():  # src/distance_measures/one_to_many.rs:380-413
    q = [1.0, 2.0, 3.0]
    db = np.array([[1, 2, 3], [2, 3, 4], [0, 0, 0]], np.float32)
    r = orc.one_to_many(q, db, 3, 3, orc.SQUARED_L2)
    assert approx(r[0], 0, 1e-6) and approx(r[1], 3, 1e-6) and approx(r[2], 14, 1e-6)
    db = np.array([[1, 1, 1], [2, 2, 2]], np.float32)
    r = orc.one_to_many(q, db, 3, 2, orc.DOT_PRODUCT)
    assert approx(r[0], -6, 1e-6) and approx(r[1], -12, 1e-6)


def test_one_to_many_strided():  # src/distance_measures/one_to_many.rs:415-430
    data = np.array([1, 2, 0, 0, 3, 4, 0, 0], np.float32)
    r = orc.one_to_many([1.0, 2.0], data, 4, 2, orc.SQUARED_L2)
    assert approx(r[0], 0.0, 1e-6) and approx(r[1], 8.0, 1e-6)


def test_one_to_one_distances():
    # src/distance_measures/one_to_one.rs:672-694, tests/unit_tests.rs:145-180
    assert approx(orc.squared_l2_avx2([1, 2, 3], [4, 5, 6]), 27.0, 1e-6)
    assert approx(orc.dot_product_avx2([1, 2, 3], [4, 5, 6]), 32.0, 1e-6)
    assert approx(np.sqrt(orc.squared_l2_avx2([0, 0, 0], [3, 4, 0])), 5.0, 1e-6)
    assert approx(orc.squared_l2_avx2([0, 0], [3, 4]), 25.0, 1e-6)


def test_compute_stride():  # src/data_format/dataset.rs:90-96
    assert orc.compute_stride(128) == 128
    assert orc.compute_stride(96) == 96
    assert orc.compute_stride(3) == 16
    assert orc.compute_stride(17) == 32


# ---- top-k structures -------------------------------------------------------------
def test_top_k_basic():  # src/brute_force/top_k.rs:399-422 (and FixedTopK :467-490)
    idx, dist = orc.topk_run(3, [0, 1, 2, 3, 4], [5.0, 3.0, 7.0, 4.0, 6.0])
    assert list(idx) == [1, 3, 0]
    assert list(dist) == [3.0, 4.0, 5.0]


def test_top_k_empty():  # src/brute_force/top_k.rs:424-429
    idx, _ = orc.topk_run(5, [], [])
    assert idx.size == 0
    idx, _ = orc.topk_run(0, [0, 1], [1.0, 2.0])
    assert idx.size == 0


def test_fast_top_neighbors():  # src/brute_force/top_k.rs:431-449
    idx, dist = orc.fast_top_neighbors_run(3, [0, 1, 2, 3], [5.0, 3.0, 7.0, 2.0])
    assert idx.size == 3
    assert idx[0] == 3 and idx[1] == 1
    assert list(dist) == [2.0, 3.0, 5.0]


def test_fast_top_neighbors_batch():  # src/brute_force/top_k.rs:451-465
    idx, _ = orc.fast_top_neighbors_push_batch(3, [0, 1, 2, 3, 4], [5.0, 3.0, 7.0, 1.0, 4.0])
    assert list(idx) == [3, 1, 4]


def test_fixed_top_k_stress_equivalent():  # src/brute_force/top_k.rs:492-515
    d = np.array([(i * 7) % 100 for i in range(100)], np.float32)
    idx, dist = orc.topk_run(10, np.arange(100), d)
    assert idx.size == 10
    assert np.all(np.diff(dist) >= 0)
    assert dist[-1] < 10.0


def test_top_k_tie_rule():
    """TopK keeps the k lexicographically smallest (dist, idx): strict '<' replace
    and a (dist, idx) max-heap (src/brute_force/top_k.rs:66-81)."""
    d = np.array([1, 1, 1, 1, 0, 1], np.float32)
    idx, _ = orc.topk_run(3, np.arange(6), d)
    assert sorted(idx.tolist()) == [0, 1, 4]


# ---- brute force ------------------------------------------------------------------
CUBE = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]], np.float32)


def _cube():
    data, st = orc.to_strided(CUBE)
    return data, st


def test_brute_force_search():  # src/brute_force/searcher.rs:280-291, tests/unit_tests.rs:204-216
    data, st = _cube()
    idx, dist = orc.bf_search(data, 5, 3, st, orc.SQUARED_L2, [0, 0, 0], 3)
    assert idx.size == 3 and idx[0] == 0 and abs(dist[0]) < 1e-6
    idx, dist = orc.bf_search(data, 5, 3, st, orc.SQUARED_L2, [0, 0, 0], 1)
    assert idx.size == 1 and idx[0] == 0


def test_brute_force_search_all():  # src/brute_force/searcher.rs:293-306
    data, st = _cube()
    idx, dist = orc.bf_search(data, 5, 3, st, orc.SQUARED_L2, [0.5, 0.5, 0.5], 5)
    assert idx.size == 5 and np.all(np.diff(dist) >= 0)
    # k > n clamps (searcher.rs:91)
    idx, _ = orc.bf_search(data, 5, 3, st, orc.SQUARED_L2, [0.5, 0.5, 0.5], 50)
    assert idx.size == 5


def test_brute_force_dot_product():  # src/brute_force/searcher.rs:308-325
    data, st = orc.to_strided(np.array([[1, 0], [0, 1], [1, 1]], np.float32))
    idx, dist = orc.bf_search(data, 3, 2, st, orc.DOT_PRODUCT, [1, 0], 3)
    assert idx.size == 3 and dist[0] <= dist[1]
    assert list(dist) == [-1.0, -1.0, 0.0]


def test_brute_force_radius():  # src/brute_force/searcher.rs:327-338, tests/unit_tests.rs:250-259
    data, st = _cube()
    idx, _ = orc.bf_search_radius(data, 5, 3, st, orc.SQUARED_L2, [0, 0, 0], 1.5)
    assert idx.size == 4


def test_brute_force_batched():  # src/brute_force/searcher.rs:340-354
    data, st = _cube()
    q = np.array([[0, 0, 0], [1, 1, 1]], np.float32)
    idx, dist, cnt = orc.bf_search_batched(data, 5, 3, st, orc.SQUARED_L2, q, 2)
    assert idx.shape == (2, 2) and list(cnt) == [2, 2]


def test_brute_force_empty_dataset():  # src/brute_force/searcher.rs:356-365
    idx, _ = orc.bf_search(np.zeros(0, np.float32), 0, 3, 16, orc.SQUARED_L2, [1, 2, 3], 5)
    assert idx.size == 0


def test_brute_force_dimension_mismatch():  # src/brute_force/searcher.rs:367-376
    data, st = _cube()
    with pytest.raises(ValueError):
        orc.bf_search(data, 5, 3, st, orc.SQUARED_L2, [1, 2], 5)


def test_stress_recall_verification():
    """tests/stress_tests.rs:325-363: brute force == naive full stable sort, index
    exact and |dd| < 1e-5 on 1000x32 U[0,1) (own RNG; the assertion is the pin)."""
    from scann_rust_amd import synth
    N, DIM, K = 1000, 32, 10
    rows = synth.uniform_f32(N, DIM, 42)
    q = synth.uniform_f32(1, DIM, 123)[0]
    data, st = orc.to_strided(rows)
    idx, dist = orc.bf_search(data, N, DIM, st, orc.SQUARED_L2, q, K)
    alld = np.array([orc.squared_l2_avx2(q, rows[i]) for i in range(N)], np.float32)
    order = np.argsort(alld, kind="stable")
    assert list(idx) == list(order[:K])
    assert np.all(np.abs(dist - alld[order[:K]]) < 1e-5)


# ---- re-rank ------------------------------------------------------------------------
def test_reordering():  # src/utils/reordering.rs:102-122
    data, st = orc.to_strided(np.array([[0, 0], [1, 0], [2, 0], [3, 0]], np.float32))
    idx, dist = orc.reorder(data, st, 2, [0, 0], [2, 1, 3, 0], 3)
    assert list(idx) == [0, 1, 2]
    assert list(dist) == [0.0, 1.0, 4.0]


# ---- partitioner ----------------------------------------------------------------------
def test_partition_sorted():  # src/partitioning/tree_partitioner.rs:289-304
    centers = np.array([[0.95, 0.475], [10.95, 10.475], [0.95, 10.475]], np.float32)
    tok, d = orc.partition(centers, [0.0, 0.0], 2)
    assert tok.size == 2 and d[1] >= d[0] and tok[0] == 0
    tok, d = orc.partition(centers, [0.0, 0.0], 10)  # min(P, L) :214
    assert tok.size == 3


def test_partition_tie_is_stable():
    """sort_by_key is stable: equal distances keep ascending partition id (:212)."""
    centers = np.array([[1, 0], [0, 1], [-1, 0], [0, -1], [0, 0.5]], np.float32)
    tok, d = orc.partition(centers, [0.0, 0.0], 5)
    assert list(tok) == [4, 0, 1, 2, 3]


# ---- LUT16 -------------------------------------------------------------------------------
def test_lut16_batch_portable():  # src/simd/tests.rs:237-264
    lut = np.zeros((2, 16), np.uint8)
    lut[0] = np.arange(16)
    lut[1] = 15 - np.arange(16)
    packed = np.array([0x00, 0x11, 0x0F, 0xF0], np.uint8)
    r = orc.lut16_distances_batch_raw(packed, lut, 2, 4)
    assert list(r) == [15.0, 15.0, 30.0, 0.0]


def test_quantization_roundtrip():  # src/hashes/lut16_simd.rs:306-330
    t = np.stack([np.arange(16), 15 - np.arange(16)]).astype(np.float32)
    lut8, bias, mult = orc.lut16_quantize(t)
    assert abs(orc.lut16_distance_single(lut8, 2, bias, mult, [0, 0]) - 15.0) < 0.1
    assert abs(orc.lut16_distance_single(lut8, 2, bias, mult, [15, 15]) - 15.0) < 0.1
    assert abs(orc.lut16_distance_single(lut8, 2, bias, mult, [5, 10]) - 10.0) < 0.1


def test_lut16_batch_computation():  # src/hashes/lut16_simd.rs:332-352
    t = np.arange(16, dtype=np.float32)[None]
    lut8, bias, mult = orc.lut16_quantize(t)
    r = orc.lut16_distances_batch(np.array([0x00, 0x05, 0x0A, 0x0F], np.uint8), lut8, 1, 4,
                                  bias, mult)
    for got, want in zip(r, [0.0, 5.0, 10.0, 15.0]):
        assert abs(got - want) < 0.1


def test_two_subspace_packed():  # src/hashes/lut16_simd.rs:354-375
    t = np.stack([np.arange(16), 15 - np.arange(16)]).astype(np.float32)
    lut8, bias, mult = orc.lut16_quantize(t)
    r = orc.lut16_distances_batch(np.array([0x00, 0x55, 0x0F, 0xF0], np.uint8), lut8, 2, 4,
                                  bias, mult)
    for got, want in zip(r, [15.0, 15.0, 30.0, 0.0]):
        assert abs(got - want) < 0.5


def test_batch_equals_single():  # src/hashes/lut16_simd.rs:377-411 (portable semantics, F7)
    t = (np.arange(16, dtype=np.float32) * 2.0)
    lut8, bias, mult = orc.lut16_quantize(np.stack([t, t]))
    n = 100
    lo = np.arange(n) % 16
    hi = (np.arange(n) + 5) % 16
    packed = (lo | (hi << 4)).astype(np.uint8)
    r = orc.lut16_distances_batch(packed, lut8, 2, n, bias, mult)
    for i in range(n):
        e = orc.lut16_distance_single(lut8, 2, bias, mult, [lo[i], hi[i]])
        assert abs(r[i] - e) < 0.5
        assert r[i] == np.float32(e)  # same arithmetic -> bit equal


def test_degenerate_quantisation():  # src/hashes/lut16_simd.rs:62-72 (range < 1e-10)
    lut8, bias, mult = orc.lut16_quantize(np.full((3, 16), 2.5, np.float32))
    assert mult == 1.0 and bias == 2.5 and not lut8.any()


def test_packed_codes_roundtrip():  # src/hashes/lut16.rs:312-328
    codes = np.array([[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]], np.uint8)
    packed = orc.pack4(codes)
    assert packed.shape == (3, 2)
    assert list(packed[0]) == [0x10, 0x32]
    assert np.array_equal(orc.unpack4(packed, 4), codes)
    odd = np.array([[1, 2, 3]], np.uint8)  # odd S pads the high nibble with 0 (:51)
    p = orc.pack4(odd)
    assert list(p[0]) == [0x21, 0x03]
    assert np.array_equal(orc.unpack4(p, 3), odd)


def test_lut16_lookup_tables():  # src/hashes/lut16.rs:339-366
    cb = np.zeros((2, 16, 2), np.float32)
    cb[0, :, 0] = np.arange(16)
    cb[1, :, 1] = np.arange(16)
    lut = orc.lut_from_query(cb, [5.0, 0.0, 0.0, 5.0])
    assert orc.lut_distance(lut, [5, 5]) < 0.01
    assert abs(orc.lut_distance(lut, [0, 0]) - 50.0) < 0.01
    assert abs(orc.lut16_distance_packed_f32(lut, orc.pack4(np.array([[0, 0]], np.uint8))[0]) - 50.0) < 0.01


# ---- codebook encode -----------------------------------------------------------------------
def test_encode_lowest_index_on_ties():  # src/hashes/codebook.rs:82-95 (strict '<')
    cb = np.zeros((1, 4, 2), np.float32)
    cb[0] = [[1, 0], [-1, 0], [0, 1], [0, -1]]
    assert list(orc.encode(cb, [0.0, 0.0])) == [0]
    assert list(orc.encode(cb, [0.0, 0.9])) == [2]


def test_trainer_encode_matches_oracle():
    from scann_rust_amd import synth, trainer
    X = synth.uniform_f32(500, 32, 7)
    cb = trainer.train_codebook(X, 8, 16, iters=3, seed=1)
    assert np.array_equal(trainer.encode(cb, X), orc.encode_many(cb, X))
    assert np.array_equal(trainer.pack4(trainer.encode(cb, X)), orc.pack4(orc.encode_many(cb, X)))


def test_txh_filter_equals_search_over_allowed_rows():
    """search_partition skips disallowed rows before scoring (tree_x_hybrid/mod.rs:327-332),
    so a filtered search == an unfiltered search of an index whose leaves hold only the
    allowed rows.  No reference test covers the filter path: parity unpinned by the reference,
    this pins the restatement against its own unfiltered path."""
    import helpers as H
    from scann_rust_amd import hip, synth
    n, dim = 1200, 32
    rows, data, stride, ix, oix, kw = H.make_txh_case(n, dim, 8, 8, seed=3, P=3, mult=4.0)
    allowed = np.arange(0, n, 3)
    keep = np.isin(ix["leaf_ids"], allowed)
    sizes = np.add.reduceat(keep.astype(np.uint32), ix["leaf_off"][:-1].astype(np.int64)) \
        * (np.diff(ix["leaf_off"].astype(np.int64)) > 0)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    sub = orc.TxhIndex(data, stride, dim, ix["centers"], off, ix["leaf_ids"][keep],
                       ix["codebook"], ix["codes"][keep], partitions_to_search=3,
                       pre_reorder_multiplier=4.0)
    oix.allow = hip.allow_bitmap(n, allowed)
    q = synth.uniform_f32(10, dim, 4)
    for i in range(10):
        a = orc.txh_search(oix, q[i], 5, stages=True)
        b = orc.txh_search(sub, q[i], 5, stages=True)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def _clustered_30():   # trees/kmeans.rs:438-459
    pts = []
    for bx, by in ((0.0, 0.0), (10.0, 10.0), (0.0, 10.0)):
        for i in range(10):
            pts.append([bx + np.float32(i) * np.float32(0.1), by + np.float32(i) * np.float32(0.05)])
    return np.array(pts, np.float32)


def test_kmeans_lloyd_reference_unit_tests():  # trees/kmeans.rs:461-498 (structural asserts)
    x = _clustered_30()
    c, a, sizes, inertia, iters, conv = orc.kmeans_lloyd(x, 30, 2, 2, x[[0, 10, 20]])
    assert c.shape == (3, 2) and a.size == 30 and sizes.sum() == 30 and (conv or iters > 0)
    assert sorted(sizes.tolist()) == [10, 10, 10]
    # centres = means of the three groups (f64 sum, cast to f32: update_centers :382-414)
    for g in range(3):
        want = (x[10 * g:10 * g + 10].astype(np.float64).sum(0) / 10).astype(np.float32)
        assert np.array_equal(c[a[10 * g]], want)
    y = np.array([[1.0, 2.0], [1.1, 2.1], [0.9, 1.9]], np.float32)   # test_kmeans_single_cluster
    c, a, sizes, *_ = orc.kmeans_lloyd(y, 3, 2, 2, y[:1])
    assert c.shape == (1, 2) and np.all(a == 0)


def test_kmeans_lloyd_empty_cluster_reseeds_from_row():  # trees/kmeans.rs:405-408
    x = _clustered_30()
    init = np.array([[0, 0], [10, 10], [1000, 1000]], np.float32)   # third centre attracts nothing
    c, a, sizes, inertia, iters, conv = orc.kmeans_lloyd(x, 30, 2, 2, init, max_iterations=1)
    assert np.array_equal(c[2], x[2 % 30])


def test_l1_and_cosine_known_answers():
    """distance_measures/one_to_one.rs:664-670 (L1 = 9), :697-718 (cosine similarity / distance of axis
    vectors), plus the structure of the restated kernels: l1 = 8 lane chains + the AVX2 hsum tree + tail;
    cosine = add(mul) lane chains reduced as wide 0.7 does without AVX (two sequential halves)."""
    assert orc.measure_distance(orc.L1, np.array([1, 2, 3], np.float32), np.array([4, 5, 6], np.float32)) == 9.0
    e0, e1 = np.array([1, 0], np.float32), np.array([0, 1], np.float32)
    assert abs(orc.measure_distance(orc.COSINE, e0, e0)) < 1e-6
    assert abs(orc.measure_distance(orc.COSINE, e0, e1) - 1.0) < 1e-6
    assert orc.measure_distance(orc.COSINE, np.zeros(2, np.float32), e1) == 1.0       # a zero norm: similarity 0
    rng = np.random.default_rng(1)
    a = rng.standard_normal(29).astype(np.float32)
    b = rng.standard_normal(29).astype(np.float32)
    lanes = np.zeros(8, np.float32)
    for c in range(3):
        lanes = (lanes + np.abs(a[8 * c:8 * c + 8] - b[8 * c:8 * c + 8])).astype(np.float32)
    s = [np.float32(lanes[j] + lanes[j + 4]) for j in range(4)]
    r = np.float32(np.float32(s[0] + s[1]) + np.float32(s[2] + s[3]))
    for j in range(24, 29):
        r = np.float32(r + np.float32(abs(np.float32(a[j] - b[j]))))
    assert np.float32(orc.measure_distance(orc.L1, a, b)) == r

    def wide(v):
        lo = np.float32(np.float32(np.float32(v[0] + v[1]) + v[2]) + v[3])
        hi = np.float32(np.float32(np.float32(v[4] + v[5]) + v[6]) + v[7])
        return np.float32(lo + hi)
    ab = np.zeros(8, np.float32); aa = np.zeros(8, np.float32); bb = np.zeros(8, np.float32)
    for c in range(3):
        x, y = a[8 * c:8 * c + 8], b[8 * c:8 * c + 8]
        ab = (ab + (x * y).astype(np.float32)).astype(np.float32)
        aa = (aa + (x * x).astype(np.float32)).astype(np.float32)
        bb = (bb + (y * y).astype(np.float32)).astype(np.float32)
    sab, saa, sbb = wide(ab), wide(aa), wide(bb)
    for j in range(24, 29):
        sab = np.float32(sab + np.float32(a[j] * b[j]))
        saa = np.float32(saa + np.float32(a[j] * a[j]))
        sbb = np.float32(sbb + np.float32(b[j] * b[j]))
    want = np.float32(1.0) - np.float32(sab / np.float32(np.sqrt(saa) * np.sqrt(sbb)))
    assert np.float32(orc.measure_distance(orc.COSINE, a, b)) == np.float32(want)


def test_fp8_codec_known_answers():
    """quantization/fp8.rs tests (:278-343): E4M3 / E5M2 round trips within the stated relative errors,
    Fp8Quantizer::e4m3 on [1, 2, 3, 4] within 0.5, the FP8 one-to-many kernels on query == database
    ([1, 2, 3, 4]: |dot - 30| < 1, squared L2 < 1; one_to_many_asymmetric.rs:327-377 negate the dot) --
    plus the codec's fixed points the restatement must keep: bias 7 / 15 layouts, max codes 0x7E / 0x7C for
    overflow, infinity and NaN, flush-to-zero below the smallest normal, calibrate_scale."""
    for val in (0.0, 1.0, -1.0, 0.5, 2.0, 100.0, -0.1):                         # test_fp8_e4m3_roundtrip
        rec = orc.fp8_to_f32(orc.fp8_from_f32(val))
        if val != 0.0:
            assert abs((val - rec) / val) < 0.2, (val, rec)
        else:
            assert rec == 0.0
    for val in (0.0, 1.0, -1.0, 0.5, 2.0, 4.0, -0.125):                         # test_fp8_e5m2_roundtrip
        rec = orc.fp8_to_f32(orc.fp8_from_f32(val, orc.FP8_E5M2), orc.FP8_E5M2)
        if abs(val) > 1e-6:
            assert abs((val - rec) / val) < 0.5, (val, rec)
    vals = np.array([1.0, 2.0, 3.0, 4.0], np.float32)                           # test_fp8_quantizer
    deq = orc.fp8_dequantize(orc.fp8_quantize(vals), 1.0)
    assert np.all(np.abs(vals - deq) < 0.5)
    db = orc.fp8_quantize(vals)                                                 # test_simd_operations
    dot = orc.one_to_many_fp8(vals, db, 4, 1, orc.DOT_PRODUCT)[0]
    assert abs(-dot - 30.0) < 1.0
    assert orc.one_to_many_fp8(vals, db, 4, 1, orc.SQUARED_L2)[0] < 1.0
    # exact layouts: 1.0 = exponent field 7 (E4M3) / 15 (E5M2), powers of two and 1.5 are exact
    assert orc.fp8_from_f32(1.0) == 0x38 and orc.fp8_from_f32(-2.0) == 0xC0 and orc.fp8_from_f32(1.5) == 0x3C
    assert orc.fp8_from_f32(1.0, orc.FP8_E5M2) == 0x3C and orc.fp8_to_f32(0x3C, orc.FP8_E5M2) == 1.0
    assert orc.fp8_to_f32(0x7E) == 448.0
    assert orc.fp8_to_f32(0x7C, orc.FP8_E5M2) == 65536.0   # (to_f32_e5m2 decodes its overflow code as 2^16, :185-203)
    for big in (1e9, np.inf, np.nan, 256.0):      # exponent field 15 is only ever the maximum (fp8.rs:108-111)
        assert orc.fp8_from_f32(big) == 0x7E
    assert orc.fp8_from_f32(-np.inf) == 0xFE and orc.fp8_from_f32(np.inf, orc.FP8_E5M2) == 0x7C
    assert orc.fp8_from_f32(2.0 ** -7) == 0 and orc.fp8_from_f32(-(2.0 ** -8)) == 0x80   # flush to (signed) zero
    assert orc.fp8_from_f32(2.0 ** -6) == 0x08                                            # smallest normal
    assert orc.fp8_from_f32(1.9375) == 0x38      # the mantissa carry wraps without bumping the exponent (:114)
    assert orc.fp8_calibrate_scale(2.0) == np.float32(224.0) and orc.fp8_calibrate_scale(0.0) == np.float32(448.0 / 1e-10)
    assert orc.fp8_calibrate_scale(4.0, orc.FP8_E5M2) == np.float32(14336.0)
