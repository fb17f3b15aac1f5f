"""CPU-only checks: the C-ABI library loads and exports every symbol include/scann_hip.h
declares (no compute calls without a GPU), and host-side logic."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "scann_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scann_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from scann_rust_amd import build, hip
    build.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libscann_hip.so does not export %s" % name
    assert sorted(hip.EXPORTS) == declared


def test_compute_stride_and_version_without_gpu():
    from scann_rust_amd import build, hip
    build.build()
    L = hip.load()
    assert L.scann_hip_compute_stride(128) == 128
    assert L.scann_hip_compute_stride(96) == 96
    assert L.scann_hip_compute_stride(3) == 16
    assert b"gfx950" in L.scann_hip_version()


def test_no_silent_fallback_without_gpu():
    """Without a GPU the product must fail loudly (Unavailable), never compute on the CPU."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    from scann_rust_amd import build, hip
    build.build()
    with pytest.raises(hip.ScannError) as e:
        hip.context(0)
    assert e.value.code in (hip.UNAVAILABLE, hip.INTERNAL)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "scann_rust_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "scann_oracle" not in text, \
                    "%s references the oracle" % os.path.join(dirpath, f)


def test_assign_leaves_balanced():
    from scann_rust_amd import build, hip
    build.build()
    sizes = np.array([100, 90, 80, 10, 10, 10, 5, 5], np.uint32)
    owner = hip.assign_leaves(sizes, 2)
    loads = [int(sizes[owner == g].sum()) for g in range(2)]
    assert abs(loads[0] - loads[1]) <= 40 and sum(loads) == int(sizes.sum())
    assert set(owner.tolist()) == {0, 1}


def test_synth_generators_are_deterministic():
    from scann_rust_amd import synth
    a = synth.uniform_f32(100, 16, 42)
    b = synth.uniform_f32(100, 16, 42)
    assert np.array_equal(a, b)
    assert a.min() >= 0.0 and a.max() < 1.0
    # chunked generation is position-consistent
    c = synth.uniform_f32(100, 16, 42, chunk_rows=7)
    assert np.array_equal(a, c)
    pts, assign = synth.clustered_f32(500, 8, 7, n_clusters=10)
    assert pts.shape == (500, 8) and assign.max() < 10


def test_trainer_index_is_consistent():
    from scann_rust_amd import synth, trainer
    X = synth.uniform_f32(2000, 32, 3)
    ix = trainer.build_txh_index(X, 10, 8, kmeans_iters=3, pq_iters=3)
    assert ix["leaf_off"][0] == 0 and ix["leaf_off"][-1] == 2000
    assert sorted(ix["leaf_ids"].tolist()) == list(range(2000))
    for l in range(10):  # ascending datapoint index inside each leaf
        ids = ix["leaf_ids"][ix["leaf_off"][l]:ix["leaf_off"][l + 1]]
        assert np.all(np.diff(ids.astype(np.int64)) > 0)
    assert ix["codes"].max() < 16
    with pytest.raises(ValueError):
        trainer.train_codebook(X, 7, 16)   # 32 % 7 != 0 (codebook.rs:154-159)


# ---- one struct, three statements: the C header, the ctypes binding, the Rust block of INTEGRATION.md --------
_C_WIDTH = {"uint64_t": 8, "uint32_t": 4, "int32_t": 4, "float": 4, "int": 4}
_RUST_WIDTH = {"u64": 8, "u32": 4, "i32": 4, "c_float": 4, "f32": 4, "c_int": 4}


def _header_structs():
    """{struct name: [(field, width in bytes)]} parsed from include/scann_hip.h (pointers are 8 bytes)."""
    text = open(os.path.join(ROOT, "include", "scann_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(scann_hip_\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(?:const\s+)?(\w+)\s*(\**)\s*(.*)$", decl, flags=re.S)
            ctype, stars, names = m.group(1), m.group(2), m.group(3)
            for nm in names.split(","):
                nm = nm.strip()
                ptr = bool(stars) or nm.startswith("*")
                fields.append((nm.lstrip("* "), 8 if ptr else _C_WIDTH[ctype]))
        out[name] = fields
    return out


def _rust_structs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out = {}
    for name, body in re.findall(r"#\[repr\(C\)\]\npub struct (scann_hip_\w+) \{\n(.*?)\n\}", text, flags=re.S):
        fields = []
        for nm, ty in re.findall(r"pub (\w+):\s*([^,\n]+),", body):
            ty = ty.strip()
            fields.append((nm, 8 if ty.startswith("*") else _RUST_WIDTH[ty]))
        if fields:
            out[name] = fields
    return out


def test_struct_layouts_agree_header_ctypes_rust():
    """INTEGRATION.md's scann_hip_txh_desc had drifted from the header (a missing trailing
    field = an out-of-bounds read by the library).  Field names, order and widths of every ABI struct
    must be the same in the header, in hip.py's ctypes classes and in INTEGRATION.md's #[repr(C)] block,
    and the compiled library's own sizeof/offsetof must equal the ctypes layout."""
    from scann_rust_amd import build, hip
    build.build()
    hdr, rust = _header_structs(), _rust_structs()
    classes = {"scann_hip_txh_desc": hip.TxhDesc, "scann_hip_search_opts": hip.SearchOpts,
               "scann_hip_file_info": hip.FileInfo}
    assert set(hdr) == set(classes), sorted(hdr)
    for name, cls in classes.items():
        ct = [(f, ctypes.sizeof(t)) for f, t in cls._fields_]
        assert ct == hdr[name], "%s: ctypes %s != header %s" % (name, ct, hdr[name])
        assert name in rust, "INTEGRATION.md lacks a #[repr(C)] %s" % name
        assert rust[name] == hdr[name], "%s: INTEGRATION.md %s != header %s" % (name, rust[name], hdr[name])
    lay = hip.abi_layout()
    assert lay == [ctypes.sizeof(hip.TxhDesc), hip.TxhDesc.distance_measure.offset,
                   ctypes.sizeof(hip.SearchOpts), hip.SearchOpts.bf_exact.offset,
                   ctypes.sizeof(hip.FileInfo), hip.FileInfo.has_data.offset]


def test_integration_md_lists_every_extern():
    import subprocess
    import sys
    # the Rust block is generated from the header; a stale block fails here
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_binding.py")]) == 0
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    bound = set(re.findall(r"\bfn (scann_hip_[a-z0-9_]+)\s*\(", text))
    missing = [s for s in _declared_symbols() if s not in bound]
    assert not missing, "INTEGRATION.md's extern block lacks %s" % missing


def test_comm_layout_python_mirror_matches_library_constants():
    """sharding.comm_layout is the Python statement of comm.hip's block layout (used by the gloo test)."""
    from scann_rust_amd import build, hip, sharding
    build.build()
    lay = sharding.comm_layout(nq=10, world=4, m_local=7, k=3)
    assert lay["qr"] == 3 and lay["nq_pad"] == 12
    # compact blocks: room for fill / world of the worst case (default fill 2.5), never less than one full list
    assert lay["cap"] == 14 and lay["blk_count"] == 0 and lay["blk_flag"] == 12 and lay["blk_keys"] == 16
    assert lay["blk_idx"] == 16 + 14 * 8 and lay["blk_exact"] == lay["blk_idx"] + 14 * 4
    assert lay["block_bytes"] % 16 == 0 and lay["block_bytes"] >= lay["blk_exact"] + 14 * 4
    big = sharding.comm_layout(1024, 8, 8192, 10)
    assert big["cap"] == 128 * 8192 * 5 // 16 and big["block_bytes"] < 0.32 * 128 * 8192 * 16   # bytes per link per step
    assert sharding.comm_layout(1024, 8, 8192, 10, worst_case=True)["cap"] == 128 * 8192
    for nq, world, m, k in ((10, 4, 7, 3), (1024, 8, 8192, 10), (1, 1, 1, 1), (12, 3, 40, 10), (5, 8, 30, 10), (1024, 2, 100, 10)):
        assert sharding.comm_layout(nq, world, m, k) == hip.comm_layout(nq, world, m, k)
