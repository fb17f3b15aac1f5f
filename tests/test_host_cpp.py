"""The C++ host-side mirror of the reference API (scann_rust_amd/host/scann.hpp): compiles
on CPU; on a GPU its test program replays the reference's own searcher unit tests
(brute_force/searcher.rs:280-376, hashes/hasher.rs:322-380, tree_x_hybrid/mod.rs:436-468)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "scann_rust_amd", "host")


def _compile(name="host_test"):
    from scann_rust_amd import build
    build.build_host()
    return os.path.join(HOST, name)


def test_host_mirror_compiles_and_fails_loudly_without_gpu():
    exe = _compile()
    import torch
    if torch.cuda.device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stdout   # Unavailable, no CPU fallback


@pytest.mark.gpu
def test_host_mirror_reference_unit_tests():
    exe = _compile()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_test ok" in r.stdout


# ---- ann_benchmark CLI (src/bin/ann_benchmark.rs) ------------------------------------------------
def _run_cli(*args, timeout=300):
    return subprocess.run([_compile("ann_benchmark")] + [str(a) for a in args], capture_output=True, text=True,
                          timeout=timeout)


def test_ann_benchmark_cli_arguments():
    """Flag handling of parse_args (ann_benchmark.rs:206-274); needs no GPU."""
    r = _run_cli("--help")
    assert r.returncode == 0 and "--partitions-to-search" in r.stdout and "brute-force|partitioned|hashed|tree-ah" in r.stdout
    r = _run_cli("--bogus", 1)
    assert r.returncode != 0 and "unknown argument: --bogus" in r.stderr
    r = _run_cli("--k")
    assert r.returncode != 0 and "missing value for --k" in r.stderr
    r = _run_cli("--algorithm", "annoy")
    assert r.returncode != 0 and "unsupported algorithm: annoy" in r.stderr
    r = _run_cli("--distance", "hamming")
    assert r.returncode != 0 and "unsupported distance: hamming" in r.stderr
    import torch
    if torch.cuda.device_count() == 0:   # fails loudly, no CPU fallback
        r = _run_cli("--synthetic-train", 100, "--synthetic-test", 4, "--dim", 8)
        assert r.returncode != 0 and "no HIP device" in r.stderr


def _report(stdout):
    import json
    line = [l for l in stdout.splitlines() if l.startswith("json: ")]
    assert len(line) == 1, stdout
    return json.loads(line[0][6:])


REPORT_KEYS = ["dataset", "algorithm", "distance", "k", "train_size", "test_size", "dimension", "build_seconds",
               "search_seconds", "qps", "recall_at_k", "index_rss_delta_bytes"]   # BenchmarkReport :119-133


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm,snake,min_recall", [("brute-force", "brute_force", 0.999),
                                                        ("partitioned", "partitioned", 0.3),
                                                        ("hashed", "hashed", 0.2), ("tree-ah", "tree_ah", 0.1)])
def test_ann_benchmark_cli_synthetic(algorithm, snake, min_recall):
    """The reference's default run (10 000 x 64 synthetic, 200 queries, k = 10) per algorithm."""
    r = _run_cli("--algorithm", algorithm)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "=== ANN-Benchmarks style report ==="
    assert lines[1] == "dataset: synthetic_n10000_q200_d64"
    assert lines[5] == "train/test/dim: 10000/200/64"
    rep = _report(r.stdout)
    assert list(rep.keys())[:len(REPORT_KEYS)] == REPORT_KEYS
    assert rep["algorithm"] == snake and rep["distance"] == "squared_l2" and rep["k"] == 10
    assert rep["train_size"] == 10000 and rep["test_size"] == 200 and rep["dimension"] == 64
    assert rep["recall_at_k"] >= min_recall, rep
    assert rep["batched_equals_sequential"] is True
    assert rep["qps"] > 0 and rep["batched_qps"] > 0


@pytest.mark.gpu
def test_ann_benchmark_cli_json_dataset(tmp_path):
    """--data-json with train/test/neighbors (ann_benchmark.rs:334-383), limits and errors."""
    import json
    import numpy as np
    rng = np.random.default_rng(3)
    train = rng.random((500, 16), dtype=np.float32)
    test = rng.random((20, 16), dtype=np.float32)
    d = ((test[:, None, :].astype(np.float64) - train[None].astype(np.float64)) ** 2).sum(-1)
    nb = np.argsort(d, axis=1, kind="stable")[:, :10]
    path = tmp_path / "ds.json"
    path.write_text(json.dumps({"train": train.tolist(), "test": test.tolist(), "neighbors": nb.tolist(),
                                "comment": {"ignored": [1, 2, "x"]}}))
    r = _run_cli("--data-json", path, "--algorithm", "brute-force", "--k", 10)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = _report(r.stdout)
    assert rep["dataset"] == str(path) and rep["train_size"] == 500 and rep["test_size"] == 20
    assert rep["recall_at_k"] >= 0.995
    r = _run_cli("--data-json", path, "--limit-train", 100, "--limit-test", 5, "--algorithm", "partitioned",
                 "--num-partitions", 4, "--partitions-to-search", 4, "--distance", "l2")
    rep = _report(r.stdout)
    assert rep["train_size"] == 100 and rep["test_size"] == 5 and rep["distance"] == "l2"
    r = _run_cli("--data-json", path, "--k", 11)
    assert r.returncode != 0 and "neighbors rows must have at least 11 entries" in r.stderr
    for dist_name in ("cosine", "l1"):      # DistanceMeasure::distance's measures run too
        r = _run_cli("--data-json", path, "--algorithm", "partitioned", "--num-partitions", 4,
                     "--partitions-to-search", 4, "--distance", dist_name)
        assert r.returncode == 0, r.stdout + r.stderr
        assert _report(r.stdout)["distance"] == dist_name
