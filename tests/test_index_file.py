"""The SCANNIDX index container (include/scann_hip.h "index files"): the C writer /
loader of libscann_hip.so, the numpy reader / writer (scann_rust_amd/index_file.py) and the committed
fixture tests/golden/txh_small.scannidx must agree byte for byte and row for row."""
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from scann_rust_amd import hip, index_file, synth, trainer
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NOT_FOUND, DATA_LOSS = 5, 15


def _sections(kw):
    out = []
    if kw.get("data") is not None:
        out.append(("data", np.asarray(kw["data"], np.float32).reshape(-1)))
    if kw.get("centers") is not None:
        out += [("centers", kw["centers"]), ("leaf_offsets", kw["leaf_offsets"]), ("leaf_ids", kw["leaf_ids"])]
    if kw.get("codebook") is not None:
        out += [("codebook", kw["codebook"]), ("codes", kw["codes"])]
    return out


def _numpy_write(path, kw):
    cb = kw.get("codebook")
    S, K, dsub = cb.shape if cb is not None else (0, 0, 0)
    index_file.write(path, kind=1, n_rows=kw["n_rows"], n_local=kw["n_rows"], dim=kw["dim"], stride=kw["stride"],
                     num_partitions=0 if kw.get("centers") is None else kw["centers"].shape[0],
                     num_subspaces=S, num_codes=K, dims_per_subspace=dsub,
                     distance_measure=kw.get("distance_measure", 0),
                     use_residuals=1 if kw.get("use_residuals", True) else 0,
                     partitions_to_search=kw.get("partitions_to_search", 10),
                     pre_reorder_multiplier=kw.get("pre_reorder_multiplier", 3.0), sections=_sections(kw))


def _cases():
    rows, data, stride, ix, oix, kw = H.make_txh_case(700, 32, 5, 8, seed=3, P=2, kmeans_iters=2, pq_iters=2)
    yield "txh", kw
    rows, data, stride, ix, kw = H.make_ah_case(300, 64, 8, seed=4, K=256, pq_iters=2)
    yield "ah", kw
    kw = dict(kw, codebook=None, codes=None)
    rows, data, stride, ix, oix, kw2 = H.make_txh_case(500, 33, 4, 3, seed=6, P=2, kmeans_iters=2, pq_iters=1)
    yield "partitioned", dict(kw2, codebook=None, codes=None, distance_measure=hip.DOT_PRODUCT)


@pytest.mark.parametrize("name,kw", list(_cases()), ids=lambda v: v if isinstance(v, str) else "")
def test_c_writer_and_numpy_agree(tmp_path, name, kw):
    """The C writer's bytes == the numpy writer's bytes; both readers return the input arrays."""
    a, b = str(tmp_path / "a.scannidx"), str(tmp_path / "b.scannidx")
    hip.txh_write_file(a, **kw)
    _numpy_write(b, kw)
    assert open(a, "rb").read() == open(b, "rb").read()
    got = index_file.arrays(a)
    info = hip.index_file_info(a)
    for key in ("n_rows", "dim", "stride", "num_partitions", "num_subspaces", "num_codes", "distance_measure",
                "partitions_to_search", "use_residuals"):
        assert info[key] == got[key], key
    assert info["kind"] == 1 and info["file_bytes"] == os.path.getsize(a) and info["has_data"] == 1
    for name_, arr in _sections(kw):
        assert np.array_equal(np.asarray(got[name_]).reshape(-1), np.asarray(arr).reshape(-1)), name_


def test_bf_file_roundtrip(tmp_path):
    rows = synth.uniform_f32(100, 24, 1)
    data, stride = orc.to_strided(rows)
    p = str(tmp_path / "bf.scannidx")
    hip.bf_write_file(p, data, 100, 24, stride, hip.L2)
    got = index_file.arrays(p)
    assert got["kind"] == 0 and got["distance_measure"] == hip.L2 and got["stride"] == stride
    assert np.array_equal(got["data"], np.asarray(data, np.float32).reshape(100, stride))


def test_file_errors(tmp_path):
    with pytest.raises(hip.ScannError) as e:
        hip.index_file_info(str(tmp_path / "missing.scannidx"))
    assert e.value.code == NOT_FOUND
    gold = open(os.path.join(GOLD, "txh_small.scannidx"), "rb").read()
    bad = tmp_path / "bad.scannidx"
    bad.write_bytes(b"NOTANIDX" + gold[8:])
    with pytest.raises(hip.ScannError) as e:
        hip.index_file_info(str(bad))
    assert e.value.code == hip.INVALID_ARGUMENT
    bad.write_bytes(gold[:-100])           # truncated
    with pytest.raises(hip.ScannError) as e:
        hip.index_file_info(str(bad))
    assert e.value.code == DATA_LOSS
    bad.write_bytes(gold[:100])            # shorter than a header
    with pytest.raises(hip.ScannError) as e:
        hip.index_file_info(str(bad))
    assert e.value.code == DATA_LOSS
    with pytest.raises(ValueError):
        index_file.read(str(bad))
    v2 = bytearray(gold)
    v2[8] = 2                              # version 2
    bad.write_bytes(bytes(v2))
    with pytest.raises(hip.ScannError) as e:
        hip.index_file_info(str(bad))
    assert e.value.code == hip.INVALID_ARGUMENT


def test_oracle_reads_golden_index_file():
    """The CPU restatement searches the committed file (numpy reader) and reproduces the stored rows."""
    a = index_file.arrays(os.path.join(GOLD, "txh_small.scannidx"))
    exp = np.load(os.path.join(GOLD, "txh_small_expected.npz"))
    oix = orc.TxhIndex(np.ascontiguousarray(a["data"]).reshape(-1), a["stride"], a["dim"], np.array(a["centers"]),
                       np.array(a["leaf_offsets"]), np.array(a["leaf_ids"]), np.array(a["codebook"]),
                       np.array(a["codes"]), use_residuals=bool(a["use_residuals"]),
                       partitions_to_search=a["partitions_to_search"],
                       pre_reorder_multiplier=a["pre_reorder_multiplier"])
    for i, q in enumerate(exp["queries"]):
        oi, od = orc.txh_search(oix, q, exp["idx"].shape[1])[:2]
        n = exp["count"][i]
        assert oi.size == n and np.array_equal(oi, exp["idx"][i, :n])
        assert np.array_equal(od.view(np.uint32), exp["dist"][i, :n].view(np.uint32))


@pytest.mark.gpu
def test_hip_loads_golden_index_file():
    index = hip.load_file(os.path.join(GOLD, "txh_small.scannidx"))
    exp = np.load(os.path.join(GOLD, "txh_small_expected.npz"))
    idx, dist, cnt = index.search_batched(exp["queries"], exp["idx"].shape[1])
    assert np.array_equal(cnt, exp["count"])
    for i in range(cnt.size):
        n = cnt[i]
        H.assert_topk_equal_up_to_ties(idx[i, :n], dist[i, :n], exp["idx"][i, :n], exp["dist"][i, :n],
                                       what="golden file query %d" % i)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", list(_cases()), ids=lambda v: v if isinstance(v, str) else "")
@pytest.mark.parametrize("pin", ["1", "0"])
def test_loaded_index_equals_created_index(tmp_path, monkeypatch, name, kw, pin):
    """load_file(write_file(desc)) searches exactly like txh_create(desc), pinned mapping or not."""
    monkeypatch.setenv("SCANN_HIP_LOAD_PIN", pin)
    p = str(tmp_path / "x.scannidx")
    hip.txh_write_file(p, **kw)
    a, b = hip.txh_create(**kw), hip.load_file(p)
    q = synth.uniform_f32(40, kw["dim"], 9)
    ra, rb = a.search_batched(q, 7), b.search_batched(q, 7)
    for x, y in zip(ra, rb):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                              y.view(np.uint32) if y.dtype == np.float32 else y)


@pytest.mark.gpu
def test_loaded_bf_index_and_corrupt_sections(tmp_path):
    rows = synth.uniform_f32(3000, 48, 2)
    data, stride = orc.to_strided(rows)
    p = str(tmp_path / "bf.scannidx")
    hip.bf_write_file(p, data, 3000, 48, stride, hip.DOT_PRODUCT)
    a, b = hip.bf_create(data, 3000, 48, stride, hip.DOT_PRODUCT), hip.load_file(p)
    q = synth.uniform_f32(9, 48, 3)
    ra, rb = a.search_batched(q, 5), b.search_batched(q, 5)
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1].view(np.uint32), rb[1].view(np.uint32))
    # a header whose row count disagrees with the data section: DataLoss, not an out-of-bounds read
    raw = bytearray(open(p, "rb").read())
    raw[16:24] = (3001).to_bytes(8, "little")
    bad = tmp_path / "bad.scannidx"
    bad.write_bytes(bytes(raw))
    with pytest.raises(hip.ScannError) as e:
        hip.load_file(str(bad))
    assert e.value.code == DATA_LOSS


@pytest.mark.gpu
def test_inconsistent_contents_are_rejected(tmp_path):
    """a file (or descriptor) whose section SIZES are right but whose CONTENTS would make a
    search read out of bounds must be refused -- DataLoss from the loader, InvalidArgument from
    txh_create -- not accepted and searched."""
    rows, data, stride, ix, oix, kw = H.make_txh_case(700, 32, 5, 8, seed=3, P=2, kmeans_iters=2, pq_iters=2)

    def refused(mut, code_create=hip.INVALID_ARGUMENT):
        bad = dict(kw)
        mut(bad)
        with pytest.raises(hip.ScannError) as e:
            hip.txh_create(**bad)
        assert e.value.code == code_create, e.value
        p = str(tmp_path / "bad.scannidx")
        hip.txh_write_file(p, **bad)          # the writer copies bytes; it does not judge them
        with pytest.raises(hip.ScannError) as e:
            hip.load_file(p)
        assert e.value.code == DATA_LOSS, e.value

    def ids_out_of_range(b):
        ids = b["leaf_ids"].copy()
        ids[123] = 700                        # >= n_rows: the re-rank would read row 700 of 700
        b["leaf_ids"] = ids
    refused(ids_out_of_range)

    def offsets_not_monotone(b):
        off = b["leaf_offsets"].copy()
        off[2] = off[1] - 1 if off[1] > 0 else off[3] + 1
        b["leaf_offsets"] = off
    refused(offsets_not_monotone)

    def code_out_of_range(b):
        cb = b["codebook"][:, :12].copy()     # 12 trained centres, codes up to 15 remain
        b["codebook"] = cb
    if kw["codes"].max() >= 12:
        refused(code_out_of_range)

    def packed_code_out_of_range(b):
        b["codebook"] = b["codebook"][:, :12].copy()
        b["codes"] = orc.pack4(kw["codes"])
        b["codes_packed4"] = True
    if kw["codes"].max() >= 12:
        refused(packed_code_out_of_range)

    def global_sizes_too_small(b):
        sizes = (b["leaf_offsets"][1:] - b["leaf_offsets"][:-1]).astype(np.uint32)
        sizes[0] -= 1
        b["leaf_sizes_global"] = sizes
    refused(global_sizes_too_small)

    # AsymmetricHasher mode reads data row i for point i: fewer rows than points is refused
    rows2, data2, stride2, ix2, kw2 = H.make_ah_case(300, 64, 8, seed=4, pq_iters=2)
    bad = dict(kw2, data=np.ascontiguousarray(np.asarray(data2).reshape(300, stride2)[:200]), n_rows=200)
    with pytest.raises(hip.ScannError) as e:
        hip.txh_create(**bad)
    assert e.value.code == hip.INVALID_ARGUMENT
    # a header whose n_rows was lowered after writing (sizes of every section still consistent
    # with the data section being absent is not possible: tamper n_rows and the data section length)
    p = str(tmp_path / "ah.scannidx")
    hip.txh_write_file(p, **bad)
    with pytest.raises(hip.ScannError) as e:
        hip.load_file(p)
    assert e.value.code == DATA_LOSS


@pytest.mark.gpu
def test_pair_table_overflow_is_reported():
    """batch x partitions_to_search beyond 2^32 pair slots: ResourceExhausted, not a wrapped size."""
    rows, data, stride, ix, oix, kw = H.make_txh_case(9000, 16, 4200, 8, seed=2, P=4, kmeans_iters=1,
                                                      pq_iters=1)
    index = hip.txh_create(**kw)
    o = hip.default_opts()
    o.partitions_to_search = 4096
    with pytest.raises(hip.ScannError) as e:
        hip.check(hip.load().scann_hip_index_reserve(index.h, 1 << 21, 10, o))
    assert e.value.code == hip.RESOURCE_EXHAUSTED
