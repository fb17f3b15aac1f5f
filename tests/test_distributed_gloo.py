"""world_size-2 gloo test (CPU) of the multi-GPU protocol: leaf sharding, merge-key
construction, all_gather layout and merge rule.  Each rank plays its GPU with the CPU
oracle's arithmetic (numpy LUT sums), gathers over torch.distributed (gloo), merges, and
must reproduce the single-process oracle result exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N, DIM, L, S, NQ, K, P, M = 3000, 64, 12, 8, 12, 10, 5, 40


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_stage(kw, ix_full, queries, tokens_all):
    """What scann_hip_txh_search_local_device returns, computed with oracle arithmetic."""
    from oracle import pyoracle as orc
    sizes = kw["leaf_sizes_global"].astype(np.int64)
    off = kw["leaf_offsets"].astype(np.int64)
    codes, ids, rows = kw["codes"], kw["leaf_ids"], kw["data"]
    keys = np.full((NQ, M), np.iinfo(np.int64).max, np.int64)
    idx = np.zeros((NQ, M), np.int32)
    exact = np.zeros((NQ, M), np.float32)
    cnt = np.zeros(NQ, np.int32)
    for q in range(NQ):
        toks = tokens_all[q]
        vbase = np.concatenate([[0], np.cumsum(sizes[toks])])
        ks, ci = [], []
        for r, leaf in enumerate(toks):
            b, e = off[leaf], off[leaf + 1]
            if e == b:
                continue
            res = queries[q] - ix_full["centers"][leaf]
            lut = orc.lut_from_query(ix_full["codebook"], res)
            d = np.zeros(e - b, np.float32)
            for s in range(S):                      # sequential f32 sum, subspace order
                d = d + lut[s, codes[b:e, s]]
            bits = d.view(np.uint32).astype(np.uint64) | np.uint64(0x80000000)
            ks.append((bits << np.uint64(32)) | (np.uint64(vbase[r]) + np.arange(e - b, dtype=np.uint64)))
            ci.append(np.arange(b, e))
        if not ks:
            continue
        ks = np.concatenate(ks); ci = np.concatenate(ci)
        order = np.argsort(ks, kind="stable")[:M]
        c = order.size
        cnt[q] = c
        keys[q, :c] = ks[order].view(np.int64)
        idx[q, :c] = ids[ci[order]].astype(np.int32)
        exact[q, :c] = [orc.squared_l2_avx2(queries[q], rows[j, :DIM]) for j in ci[order]]
    return keys, idx, exact, cnt


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as orc
        from scann_rust_amd import sharding, synth, trainer
        rows = synth.uniform_f32(N, DIM, 11)
        queries = synth.uniform_f32(NQ, DIM, 12)
        data, stride = orc.to_strided(rows)
        ix = trainer.build_txh_index(rows, L, S, seed=3, kmeans_iters=3, pq_iters=3)
        kw = sharding.shard_txh_index(ix, data, stride, rank, world)
        tokens = np.stack([orc.partition(ix["centers"], q, P)[0] for q in queries]).astype(np.int64)
        keys, idx, exact, cnt = _local_stage(kw, ix, queries, tokens)
        g_keys = [torch.empty((NQ, M), dtype=torch.int64) for _ in range(world)]
        g_idx = [torch.empty((NQ, M), dtype=torch.int32) for _ in range(world)]
        g_ex = [torch.empty((NQ, M), dtype=torch.float32) for _ in range(world)]
        g_cnt = [torch.empty((NQ,), dtype=torch.int32) for _ in range(world)]
        dist.all_gather(g_keys, torch.from_numpy(keys))
        dist.all_gather(g_idx, torch.from_numpy(idx))
        dist.all_gather(g_ex, torch.from_numpy(exact))
        dist.all_gather(g_cnt, torch.from_numpy(cnt))
        gk = torch.stack(g_keys).numpy().view(np.uint64)
        gi = torch.stack(g_idx).numpy().view(np.uint32)
        ge = torch.stack(g_ex).numpy()
        gc = torch.stack(g_cnt).numpy()
        oi, od, oc = sharding.merge_reference(gk, gi, ge, gc, M, K)
        oix = orc.TxhIndex(data, stride, DIM, ix["centers"], ix["leaf_off"], ix["leaf_ids"],
                           ix["codebook"], ix["codes"], partitions_to_search=P,
                           pre_reorder_multiplier=M / K)
        ok = True
        for q in range(NQ):
            wi, wd = orc.txh_search(oix, queries[q], K)
            ok = ok and oc[q] == wi.size and np.array_equal(oi[q, :wi.size], wi) \
                and np.array_equal(od[q, :wi.size].view(np.uint32), wd.view(np.uint32))
        # every rank owns something and the shards partition the points
        tot = torch.tensor([kw["leaf_ids"].size], dtype=torch.int64)
        dist.all_reduce(tot)
        ok = ok and int(tot.item()) == N and kw["leaf_ids"].size > 0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_leaf_sharding_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {0: True, 1: True}


def test_assign_leaves_matches_c_abi():
    from scann_rust_amd import build, hip, sharding
    build.build()
    rng = np.random.default_rng(5)
    sizes = rng.integers(0, 500, 97).astype(np.uint32)
    for world in (1, 2, 3, 8):
        assert np.array_equal(sharding.assign_leaves(sizes, world), hip.assign_leaves(sizes, world))


def _worker_a2a(rank, world, port, ret):
    """The bench's exchange: destination blocks -> all_to_all_single -> merge of the rank's own
    NQ/world queries -> all_gather of the result rows."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as orc
        from scann_rust_amd import sharding, synth, trainer
        rows = synth.uniform_f32(N, DIM, 11)
        queries = synth.uniform_f32(NQ, DIM, 12)
        data, stride = orc.to_strided(rows)
        ix = trainer.build_txh_index(rows, L, S, seed=3, kmeans_iters=3, pq_iters=3)
        kw = sharding.shard_txh_index(ix, data, stride, rank, world)
        tokens = np.stack([orc.partition(ix["centers"], q, P)[0] for q in queries]).astype(np.int64)
        keys, idx, exact, cnt = _local_stage(kw, ix, queries, tokens)
        send = sharding.pack_blocks_reference(keys.view(np.uint64), idx.view(np.uint32), exact,
                                              cnt.view(np.uint32), world)
        recv = torch.empty(send.size, dtype=torch.uint8)
        dist.all_to_all_single(recv, torch.from_numpy(send.reshape(-1)))
        gk, gi, ge, gc = sharding.unpack_blocks(recv.numpy(), NQ, M, world)
        oi, od, oc = sharding.merge_reference(gk, gi, ge, gc, M, K)      # my NQ/world queries
        qr = NQ // world
        res = torch.from_numpy(np.concatenate([oi.view(np.uint8).reshape(-1), od.view(np.uint8).reshape(-1),
                                               oc.view(np.uint8).reshape(-1)]))
        parts = [torch.empty_like(res) for _ in range(world)]
        dist.all_gather(parts, res)
        rb = qr * K * 4
        all_i = np.concatenate([p.numpy()[:rb].view(np.uint32).reshape(qr, K) for p in parts])
        all_d = np.concatenate([p.numpy()[rb:2 * rb].view(np.float32).reshape(qr, K) for p in parts])
        all_c = np.concatenate([p.numpy()[2 * rb:].view(np.uint32) for p in parts])
        oix = orc.TxhIndex(data, stride, DIM, ix["centers"], ix["leaf_off"], ix["leaf_ids"],
                           ix["codebook"], ix["codes"], partitions_to_search=P,
                           pre_reorder_multiplier=M / K)
        ok = True
        for q in range(NQ):
            wi, wd = orc.txh_search(oix, queries[q], K)
            ok = ok and all_c[q] == wi.size and np.array_equal(all_i[q, :wi.size], wi) \
                and np.array_equal(all_d[q, :wi.size].view(np.uint32), wd.view(np.uint32))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_to_all_exchange_matches_single_process(world):
    assert NQ % world == 0
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_a2a, args=(r, world, port, ret)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _worker_comm(rank, world, port, ret, nq):
    """The protocol of scann_hip_txh_search_sharded_device (csrc/comm.hip), rank by rank on the CPU:
    the LIBRARY's block layout (scann_hip_comm_layout), the batch padded to a multiple of the ranks,
    all-to-all of destination blocks, merge of the rank's qr queries, in-place all-gather of the rows."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as orc
        from scann_rust_amd import hip, sharding, synth, trainer
        rows = synth.uniform_f32(N, DIM, 11)
        queries = synth.uniform_f32(NQ, DIM, 12)
        data, stride = orc.to_strided(rows)
        ix = trainer.build_txh_index(rows, L, S, seed=3, kmeans_iters=3, pq_iters=3)
        kw = sharding.shard_txh_index(ix, data, stride, rank, world)
        tokens = np.stack([orc.partition(ix["centers"], q, P)[0] for q in queries]).astype(np.int64)
        keys, idx, exact, cnt = (a[:nq] for a in _local_stage(kw, ix, queries, tokens))
        lay = hip.comm_layout(nq, world, M, K)                  # the library's constants
        assert lay == sharding.comm_layout(nq, world, M, K)
        retried = False
        while True:
            send = sharding.comm_pack_reference(keys.view(np.uint64), idx.view(np.uint32), exact,
                                                cnt.view(np.uint32), world, lay)
            recv = torch.empty(send.size, dtype=torch.uint8)
            dist.all_to_all_single(recv, torch.from_numpy(send.reshape(-1)))
            gk, gi, ge, gc, overflow = sharding.comm_unpack(recv.numpy(), world, M, lay)
            oi, od, oc = sharding.merge_reference(gk, gi, ge, gc, M, K)      # my qr queries (padding: count 0)
            qr = lay["qr"]
            outs = []
            status = np.full(1, 10 if overflow else 0, np.int32)             # this rank's step status, gathered with the rows
            for arr in (oi.view(np.int32), od, oc.view(np.int32), status):   # all-gathers, rank g at g * qr
                t = torch.from_numpy(np.ascontiguousarray(arr).reshape(-1))  # (gloo has no unsigned 32-bit type)
                parts = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(parts, t)
                outs.append(np.concatenate([p.numpy() for p in parts]))
            if outs[3].max() == 0:
                break
            # a compact block overflowed somewhere: EVERY rank sees Aborted and repeats with worst-case blocks (m_local = 0)
            assert not retried
            retried = True
            lay = sharding.comm_layout(nq, world, M, K, worst_case=True)
        ret["retried_%d" % rank] = retried
        outs[0], outs[2] = outs[0].view(np.uint32), outs[2].view(np.uint32)
        all_i, all_d, all_c = outs[0].reshape(-1, K)[:nq], outs[1].reshape(-1, K)[:nq], outs[2][:nq]
        oix = orc.TxhIndex(data, stride, DIM, ix["centers"], ix["leaf_off"], ix["leaf_ids"],
                           ix["codebook"], ix["codes"], partitions_to_search=P,
                           pre_reorder_multiplier=M / K)
        ok = outs[2].size == qr * world and np.all(outs[2][nq:] == 0)
        for q in range(nq):
            wi, wd = orc.txh_search(oix, queries[q], K)
            ok = ok and all_c[q] == wi.size and np.array_equal(all_i[q, :wi.size], wi) \
                and np.array_equal(all_d[q, :wi.size].view(np.uint32), wd.view(np.uint32))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nq", [(2, 11), (4, 9)])
def test_library_exchange_protocol_with_padding_matches_single_process(world, nq):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_comm, args=(r, world, port, ret, nq)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert all(ret.get(r) for r in range(world)), dict(ret)
    # world 2: blocks have room for everything; world 4 with 3 queries per rank: a block overflows, all ranks repeat
    assert len({ret["retried_%d" % r] for r in range(world)}) == 1
    assert ret["retried_0"] == (world == 4)
