"""bench.py's workload defaults (no GPU): N = 1 -> BASELINE configs[2] (AsymmetricHasher 1M x 128), N > 1 -> one
configs[4]-shaped Tree-X-Hybrid shard per GPU through the library's exchange, and the replica fallback the run
takes when the RCCL communicator cannot be created restores the N = 1 workload on every rank."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _parse(monkeypatch, *argv):
    monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
    return _bench().parse()


def test_single_gpu_default_is_the_headline_config(monkeypatch):
    a = _parse(monkeypatch)
    assert (a.workload, a.sharded, a.n, a.dim, a.subspaces, a.pre_reorder_k, a.batch, a.k) == \
        ("ah", False, 1_000_000, 128, 32, 5000, 1024, 10)
    assert a.steps >= 100 and a.warmup >= 1


def test_multi_gpu_default_is_the_sharded_tree(monkeypatch):
    a = _parse(monkeypatch, "--gpus", "8")
    assert (a.workload, a.sharded, a.n, a.dim, a.subspaces, a.leaves, a.partitions_to_search, a.pre_reorder_k) == \
        ("txh", True, 12_500_000, 96, 24, 1250, 10, 8192)
    r = _parse(monkeypatch, "--gpus", "8", "--multi-gpu", "replica")
    assert (r.workload, r.sharded, r.n, r.dim) == ("ah", False, 1_000_000, 128)


def test_replica_fallback_restores_the_single_gpu_workload(monkeypatch):
    mod = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--batch", "512"])
    a = mod.parse()
    assert a.sharded and a.workload == "txh"
    a.multi_gpu = "replica"            # what main() does when the communicator cannot be created
    mod.resolve_defaults(a)
    assert (a.workload, a.sharded, a.n, a.dim, a.subspaces, a.pre_reorder_k, a.batch) == \
        ("ah", False, 1_000_000, 128, 32, 5000, 512)
    # values given on the command line survive the fallback
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--workload", "txh", "--num-points", "2000000"])
    b = mod.parse()
    b.multi_gpu = "replica"
    mod.resolve_defaults(b)
    assert (b.workload, b.sharded, b.n) == ("txh", False, 2_000_000)
