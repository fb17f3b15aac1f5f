"""Host-side leaf sharding for multi-GPU Tree-X-Hybrid.

One process per GPU; rank g owns a subset of the k-means leaves (their codes, ids and
f32 rows).  Centroids, codebook and the GLOBAL leaf sizes are replicated so every rank
selects the same leaves and builds identical merge keys; the only exchange is one
all_gather of (merge key, index, exact distance) triples of each rank's best
pre_reorder_k candidates.  No reference counterpart (the reference is single-process).
"""
import numpy as np


def assign_leaves(leaf_sizes, world):
    """Greedy size-balanced leaf -> rank map (same rule as scann_hip_assign_leaves)."""
    leaf_sizes = np.asarray(leaf_sizes, np.int64)
    order = np.argsort(-leaf_sizes, kind="stable")
    load = np.zeros(world, np.int64)
    owner = np.zeros(leaf_sizes.size, np.uint32)
    for l in order:
        g = int(np.argmin(load))   # first minimum == lowest rank, as in the C loop
        owner[l] = g
        load[g] += leaf_sizes[l]
    return owner


def shard_txh_index(ix, data, stride, rank, world, owner=None):
    """Local view of a trained index for `rank`: kwargs of hip.txh_create.

    ix: dict(centers, leaf_off, leaf_ids, codebook, codes[, use_residuals]) as built by
    trainer.build_txh_index; data: [n][stride] rows by datapoint index."""
    leaf_off = np.asarray(ix["leaf_off"], np.int64)
    L = leaf_off.size - 1
    sizes = (leaf_off[1:] - leaf_off[:-1]).astype(np.uint32)
    if owner is None:
        owner = assign_leaves(sizes, world)
    mine = owner == rank
    local_sizes = np.where(mine, sizes, 0).astype(np.uint32)
    off = np.zeros(L + 1, np.uint32)
    off[1:] = np.cumsum(local_sizes)
    sel = np.concatenate([np.arange(leaf_off[l], leaf_off[l + 1]) for l in range(L) if mine[l]]
                         or [np.zeros(0, np.int64)]).astype(np.int64)
    ids = np.asarray(ix["leaf_ids"], np.uint32)[sel]
    return dict(data=np.ascontiguousarray(data[ids]), n_rows=int(ids.size), dim=int(ix["centers"].shape[1]),
                stride=stride, centers=ix["centers"], leaf_offsets=off, leaf_ids=ids,
                leaf_sizes_global=sizes, codebook=ix["codebook"],
                codes=np.ascontiguousarray(np.asarray(ix["codes"])[sel]),
                use_residuals=bool(ix.get("use_residuals", True)), data_is_csr_order=True)


def merge_reference(keys, idx, exact, counts, m, k):
    """Numpy statement of the merge rule (tree_x_hybrid/mod.rs:283-293, 360-361) on
    gathered triples [world][nq][m]: stable sort by key -> truncate m -> stable sort by
    exact -> truncate k.  Used by tests as the CPU model of scann_hip_txh_merge_device."""
    world, nq, _ = keys.shape
    out_idx = np.full((nq, k), 0xFFFFFFFF, np.uint32)
    out_dist = np.full((nq, k), np.inf, np.float32)
    out_cnt = np.zeros(nq, np.uint32)
    for q in range(nq):
        ks = np.concatenate([keys[g, q, :counts[g, q]] for g in range(world)])
        ii = np.concatenate([idx[g, q, :counts[g, q]] for g in range(world)])
        ee = np.concatenate([exact[g, q, :counts[g, q]] for g in range(world)])
        order = np.argsort(ks, kind="stable")[:m]
        ii, ee = ii[order], ee[order]
        o2 = np.argsort(ee, kind="stable")[:k]
        out_cnt[q] = o2.size
        out_idx[q, :o2.size] = ii[o2]
        out_dist[q, :o2.size] = ee[o2]
    return out_idx, out_dist, out_cnt


def block_layout(nq, m_local, world):
    """Byte offsets inside one destination block of the all_to_all exchange
    (scann_hip_txh_pack_blocks_device): (keys, idx, exact, count, block_bytes) for nq/world queries."""
    qr = nq // world
    per = qr * m_local
    need = per * 16 + qr * 4
    return 0, per * 8, per * 12, per * 16, (need + 255) // 256 * 256


def pack_blocks_reference(keys, idx, exact, counts, world):
    """Numpy statement of scann_hip_txh_pack_blocks_device: uint8 array [world][block_bytes]."""
    nq, m = keys.shape
    qr = nq // world
    ok, oi, oe, oc, bb = block_layout(nq, m, world)
    out = np.zeros((world, bb), np.uint8)
    for d in range(world):
        sl = slice(d * qr, (d + 1) * qr)
        out[d, ok:oi] = np.ascontiguousarray(keys[sl]).view(np.uint8).reshape(-1)
        out[d, oi:oe] = np.ascontiguousarray(idx[sl]).view(np.uint8).reshape(-1)
        out[d, oe:oc] = np.ascontiguousarray(exact[sl]).view(np.uint8).reshape(-1)
        out[d, oc:oc + qr * 4] = np.ascontiguousarray(counts[sl]).view(np.uint8).reshape(-1)
    return out


def unpack_blocks(buf, nq, m_local, world):
    """Views (keys, idx, exact, counts) [world][nq/world][...] of a received [world][block_bytes] buffer."""
    qr = nq // world
    ok, oi, oe, oc, bb = block_layout(nq, m_local, world)
    buf = np.ascontiguousarray(buf).reshape(world, bb)
    keys = np.stack([buf[g, ok:oi].view(np.uint64).reshape(qr, m_local) for g in range(world)])
    idx = np.stack([buf[g, oi:oe].view(np.uint32).reshape(qr, m_local) for g in range(world)])
    exact = np.stack([buf[g, oe:oc].view(np.float32).reshape(qr, m_local) for g in range(world)])
    cnt = np.stack([buf[g, oc:oc + qr * 4].view(np.uint32) for g in range(world)])
    return keys, idx, exact, cnt


# ---- the library's own exchange (csrc/comm.hip, scann_hip_txh_search_sharded_device) --------------
def comm_layout(nq, world, m_local, k):
    """Python statement of comm.hip::comm_layout (checked against scann_hip_comm_layout by the tests):
    the batch is padded to qr * world queries; queries past nq travel with count 0."""
    qr = (nq + world - 1) // world
    per = qr * m_local
    return dict(qr=qr, nq_pad=qr * world, block_bytes=(per * 16 + qr * 4 + 15) // 16 * 16,
                blk_idx=per * 8, blk_exact=per * 12, blk_count=per * 16,
                soa_bytes=nq * m_local * 16 + nq * 4, soa_idx=nq * m_local * 8, soa_exact=nq * m_local * 12,
                soa_count=nq * m_local * 16, res_bytes=qr * world * (2 * k + 1) * 4, res_dist=qr * world * k * 4)


def comm_pack_reference(keys, idx, exact, counts, world, lay):
    """Numpy statement of comm.hip::comm_pack_kernel: uint8 [world][block_bytes]."""
    nq, m = keys.shape
    qr = lay["qr"]
    out = np.zeros((world, lay["block_bytes"]), np.uint8)
    for d in range(world):
        q0, q1 = min(nq, d * qr), min(nq, (d + 1) * qr)
        n = q1 - q0
        kb = np.zeros((qr, m), np.uint64); ib = np.zeros((qr, m), np.uint32); eb = np.zeros((qr, m), np.float32)
        cb = np.zeros(qr, np.uint32)
        kb[:n], ib[:n], eb[:n], cb[:n] = keys[q0:q1], idx[q0:q1], exact[q0:q1], counts[q0:q1]
        out[d, :lay["blk_idx"]] = kb.view(np.uint8).reshape(-1)
        out[d, lay["blk_idx"]:lay["blk_exact"]] = ib.view(np.uint8).reshape(-1)
        out[d, lay["blk_exact"]:lay["blk_count"]] = eb.view(np.uint8).reshape(-1)
        out[d, lay["blk_count"]:lay["blk_count"] + qr * 4] = cb.view(np.uint8).reshape(-1)
    return out


def comm_unpack(buf, world, m_local, lay):
    """(keys, idx, exact, counts) [world][qr][...] of a received [world][block_bytes] buffer."""
    qr = lay["qr"]
    buf = np.ascontiguousarray(buf).reshape(world, lay["block_bytes"])
    keys = np.stack([buf[g, :lay["blk_idx"]].view(np.uint64).reshape(qr, m_local) for g in range(world)])
    idx = np.stack([buf[g, lay["blk_idx"]:lay["blk_exact"]].view(np.uint32).reshape(qr, m_local) for g in range(world)])
    exact = np.stack([buf[g, lay["blk_exact"]:lay["blk_count"]].view(np.float32).reshape(qr, m_local)
                      for g in range(world)])
    cnt = np.stack([buf[g, lay["blk_count"]:lay["blk_count"] + qr * 4].view(np.uint32) for g in range(world)])
    return keys, idx, exact, cnt
