"""Shared builders/checkers for the parity tests (test infrastructure)."""
import numpy as np

from oracle import pyoracle as orc
from scann_rust_amd import synth, trainer


def make_txh_case(n, dim, L, S, seed, K=16, use_residuals=True, P=4, mult=3.0, clustered=False,
                  kmeans_iters=5, pq_iters=5):
    """Build one trained index + the oracle view and the kwargs of hip.txh_create."""
    if clustered:
        rows, _ = synth.clustered_f32(n, dim, seed, n_clusters=max(4, L))
    else:
        rows = synth.uniform_f32(n, dim, seed)
    data, stride = orc.to_strided(rows)
    ix = trainer.build_txh_index(rows, L, S, K=K, use_residuals=use_residuals, seed=seed,
                                 kmeans_iters=kmeans_iters, pq_iters=pq_iters)
    oix = orc.TxhIndex(data, stride, dim, ix["centers"], ix["leaf_off"], ix["leaf_ids"],
                       ix["codebook"], ix["codes"], use_residuals=use_residuals,
                       partitions_to_search=P, pre_reorder_multiplier=mult)
    kwargs = dict(data=data, n_rows=n, dim=dim, stride=stride, centers=ix["centers"],
                  leaf_offsets=ix["leaf_off"], leaf_ids=ix["leaf_ids"], codebook=ix["codebook"],
                  codes=ix["codes"], codes_packed4=False, use_residuals=use_residuals,
                  partitions_to_search=P, pre_reorder_multiplier=mult)
    return rows, data, stride, ix, oix, kwargs


def make_ah_case(n, dim, S, seed, K=16, pq_iters=5):
    rows = synth.uniform_f32(n, dim, seed)
    data, stride = orc.to_strided(rows)
    ix = trainer.build_ah_index(rows, S, K=K, seed=seed, pq_iters=pq_iters)
    kwargs = dict(data=data, n_rows=n, dim=dim, stride=stride, centers=None, leaf_offsets=None,
                  leaf_ids=None, codebook=ix["codebook"], codes=ix["codes"], codes_packed4=False,
                  use_residuals=False, partitions_to_search=1, pre_reorder_multiplier=1.0)
    return rows, data, stride, ix, kwargs


def assert_topk_equal_up_to_ties(got_idx, got_dist, want_idx, want_dist, rel=0.0, what=""):
    """SURVEY.md 8c parity rule: same index multiset up to groups whose distances tie
    (bit-equal when rel == 0, else within rel); distances equal position by position."""
    got_idx = np.asarray(got_idx); want_idx = np.asarray(want_idx)
    got_dist = np.asarray(got_dist, np.float32); want_dist = np.asarray(want_dist, np.float32)
    assert got_idx.shape == want_idx.shape, "%s: length %s vs %s" % (what, got_idx.shape, want_idx.shape)
    if rel == 0.0:
        assert np.array_equal(got_dist.view(np.uint32), want_dist.view(np.uint32)), \
            "%s: distances differ bitwise\n got %s\nwant %s" % (what, got_dist, want_dist)
    else:
        assert np.allclose(got_dist, want_dist, rtol=rel, atol=0.0), \
            "%s: distances differ\n got %s\nwant %s" % (what, got_dist, want_dist)
    if np.array_equal(got_idx, want_idx):
        return
    # differing positions must sit inside tie groups; the last group may be cut by k,
    # so only compare groups fully inside the result.
    n = got_idx.size
    i = 0
    while i < n:
        j = i
        while j + 1 < n and (want_dist[j + 1] == want_dist[i] if rel == 0.0 else
                             abs(want_dist[j + 1] - want_dist[i]) <= rel * abs(want_dist[i])):
            j += 1
        if j == n - 1 and j > i or (j == n - 1 and got_idx[i] != want_idx[i]):
            # tail group: membership may legitimately differ (tie cut by k / m)
            if i == j:
                # a single differing last element is a tie only if distances are equal,
                # which the distance check above already established.
                pass
        elif sorted(got_idx[i:j + 1].tolist()) != sorted(want_idx[i:j + 1].tolist()):
            raise AssertionError("%s: indices differ outside ties at [%d,%d]\n got %s\nwant %s"
                                 % (what, i, j, got_idx, want_idx))
        i = j + 1


def recall_at_k(retrieved, gt, k):
    """bin/ann_benchmark.rs:452-471."""
    tot = 0.0
    for r, g in zip(retrieved, gt):
        tot += len(set(r[:k].tolist()) & set(g[:k].tolist())) / float(k)
    return tot / len(retrieved)


def check_txh_query(oix, query, k, got_idx, got_dist, got_tok, got_tokd, got_ci, got_cd, what=""):
    """Stage-aware parity check of one Tree-X-Hybrid query against the oracle.

    tokens / centre distances: bit-exact.  Candidates: the sorted approximate distances
    must be bitwise identical; memberships may differ only inside ties of the approximate
    distance (FastTopNeighbors' slot-order tie behaviour is not reproduced: SURVEY 8c "up
    to distance ties").  Final rows: equal to the oracle's if the candidate sets agree,
    otherwise equal to the oracle's exact re-rank OF THE GPU's candidate list."""
    oi, od, otok, otokd, oci, ocd = orc.txh_search(oix, query, k, stages=True)
    P = otok.size
    assert np.array_equal(got_tok[:P], otok), "%s tokens" % what
    assert np.array_equal(np.asarray(got_tokd[:P], np.float32).view(np.uint32), otokd.view(np.uint32))
    assert got_ci.size == oci.size, "%s candidate count %d vs %d" % (what, got_ci.size, oci.size)
    assert np.array_equal(np.asarray(got_cd, np.float32).view(np.uint32), ocd.view(np.uint32)), \
        "%s approximate distances differ" % what
    assert_topk_equal_up_to_ties(got_ci, got_cd, oci, ocd, what=what + " cand")
    if sorted(got_ci.tolist()) == sorted(oci.tolist()):
        assert got_idx.size == oi.size
        assert_topk_equal_up_to_ties(got_idx, got_dist, oi, od, what=what + " final")
    else:
        ri, rd = orc.reorder(oix.data, oix.stride, oix.dim, query, got_ci, k)
        assert got_idx.size == ri.size
        assert_topk_equal_up_to_ties(got_idx, got_dist, ri, rd, what=what + " final(gpu cands)")
