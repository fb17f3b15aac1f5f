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
def comm_fill():
    import os
    f = float(os.environ.get("SCANN_HIP_COMM_FILL", "2.5"))
    return max(f, 0.0)


def comm_layout(nq, world, m_local, k, worst_case=False):
    """Python statement of comm.hip::comm_layout (checked against scann_hip_comm_layout by the tests):
    the batch is padded to qr * world queries; queries past nq travel with count 0.  Destination blocks are
    compact: [count u32[qr] | overflow flag | pad | keys u64[cap] | idx u32[cap] | exact f32[cap]]."""
    import math
    qr = (nq + world - 1) // world
    full = qr * m_local
    cap = full
    f = comm_fill()
    if not worst_case and f > 0.0:
        cap = min(full, max(m_local, int(math.ceil(f * full / world))))
    blk_flag = qr * 4
    blk_keys = (blk_flag + 4 + 15) // 16 * 16
    blk_idx = blk_keys + cap * 8
    blk_exact = blk_idx + cap * 4
    res_dist = qr * world * k * 4
    res_status = 2 * res_dist + qr * world * 4
    return dict(qr=qr, nq_pad=qr * world, block_bytes=(blk_exact + cap * 4 + 15) // 16 * 16,
                blk_idx=blk_idx, blk_exact=blk_exact, blk_count=0,
                soa_bytes=nq * m_local * 16 + nq * 4, soa_idx=nq * m_local * 8, soa_exact=nq * m_local * 12,
                soa_count=nq * m_local * 16, res_bytes=res_status + world * 4, res_dist=res_dist,
                blk_keys=blk_keys, cap=cap, blk_flag=blk_flag, res_status=res_status)


def comm_pack_reference(keys, idx, exact, counts, world, lay):
    """Numpy statement of comm.hip::comm_offsets_kernel + comm_pack_kernel: uint8 [world][block_bytes]."""
    nq, m = keys.shape
    qr, cap = lay["qr"], lay["cap"]
    out = np.zeros((world, lay["block_bytes"]), np.uint8)
    for d in range(world):
        hdr = np.zeros(qr, np.uint32)
        kb = np.zeros(cap, np.uint64); ib = np.zeros(cap, np.uint32); eb = np.zeros(cap, np.float32)
        base, overflow = 0, 0
        for ql in range(qr):
            q = d * qr + ql
            c = min(int(counts[q]), m) if q < nq else 0
            sent = c
            if base + c > cap:
                sent = max(0, cap - base)
                overflow = 1
            hdr[ql] = sent
            if sent:
                kb[base:base + sent], ib[base:base + sent], eb[base:base + sent] = keys[q, :sent], idx[q, :sent], exact[q, :sent]
            base += c
        out[d, :qr * 4] = hdr.view(np.uint8)
        out[d, lay["blk_flag"]:lay["blk_flag"] + 4] = np.array([overflow], np.uint32).view(np.uint8)
        out[d, lay["blk_keys"]:lay["blk_idx"]] = kb.view(np.uint8)
        out[d, lay["blk_idx"]:lay["blk_exact"]] = ib.view(np.uint8)
        out[d, lay["blk_exact"]:lay["blk_exact"] + cap * 4] = eb.view(np.uint8)
    return out


def comm_unpack(buf, world, m_local, lay):
    """(keys, idx, exact, counts, overflow) of a received [world][block_bytes] buffer, the lists expanded to
    [world][qr][m_local] (what comm_recv_offsets_kernel + merge_kernel read through the per-query offsets)."""
    qr, cap = lay["qr"], lay["cap"]
    buf = np.ascontiguousarray(buf).reshape(world, lay["block_bytes"])
    keys = np.zeros((world, qr, m_local), np.uint64); idx = np.zeros((world, qr, m_local), np.uint32)
    exact = np.zeros((world, qr, m_local), np.float32); cnt = np.zeros((world, qr), np.uint32)
    overflow = False
    for g in range(world):
        hdr = buf[g, :qr * 4].view(np.uint32)
        overflow = overflow or bool(buf[g, lay["blk_flag"]:lay["blk_flag"] + 4].view(np.uint32)[0])
        kb = buf[g, lay["blk_keys"]:lay["blk_idx"]].view(np.uint64)
        ib = buf[g, lay["blk_idx"]:lay["blk_exact"]].view(np.uint32)
        eb = buf[g, lay["blk_exact"]:lay["blk_exact"] + cap * 4].view(np.float32)
        base = 0
        for ql in range(qr):
            c = int(hdr[ql])
            keys[g, ql, :c], idx[g, ql, :c], exact[g, ql, :c] = kb[base:base + c], ib[base:base + c], eb[base:base + c]
            cnt[g, ql] = c
            base += c
    return keys, idx, exact, cnt, overflow
