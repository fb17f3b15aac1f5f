#!/usr/bin/env python3
"""bench.py -- headline measurement of the MI355X ScaNN hot path.

    python bench.py --gpus N --steps K --warmup W [--workload ah|bf_dot|txh] ...

Metric (BASELINE.json): QPS @ recall10@10 + achieved HBM GB/s, 1M x 128 f32.

N = 1 (default): workload `ah` = BASELINE.json configs[2]: AsymmetricHasher LUT16 (4-bit PQ, 32 blocks
x 16 centres) over 1M x 128 uniform U[0,1) vectors, k = 10, search_with_reordering (exact f32 re-rank
of the pre_reorder_k best approximate candidates).  A "step" is one search_batched call over one batch
of `--batch` queries already resident in HBM; value = queries/s of the whole job.

N > 1 (one process per GPU, launched by torch.distributed.run): the north-star split --
a leaf-sharded Tree-X-Hybrid index, ONE shard of the BASELINE configs[4] shape per GPU (12.5M x 96,
S = 24, 1250 leaves per GPU: the index grows with N, at N = 8 it is the 100M x 96 / 10 000-leaf
configuration), searched through the LIBRARY's own entry point scann_hip_txh_search_sharded_device:
local stage -> one RCCL all-to-all of candidate blocks over xGMI -> merge -> all-gather of the k
result rows, on the library's streams (csrc/comm.hip; no torch collective on the data path --
torch.distributed/gloo carries only the rendezvous, the barriers and the max over ranks).  Every
step searches the same `--batch` queries on all ranks; partitions_to_search is fixed, so the work of
a step is fixed and split over the ranks ("strong").  `--multi-gpu replica` times N query-parallel
copies of the N = 1 workload instead (reported as `secondary` by the default N > 1 run).

The CPU oracle is used here ONLY as (a) the checker of a few result rows and (b) the cpu_baseline
leg; the timed path is libscann_hip.so through its C ABI.
"""
import argparse
import ctypes
import hashlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0          # HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
LDS_PEAK_GBPS = 157286.4        # ds_read_b128: 256 B/clk/CU x 256 CUs x 2.4 GHz
LDS_MEASURED_GBPS = 150000.0    # "Aggregate with every CU streaming: ~150 TB/s for ds_read_b64/b128"
F32_MFMA_PEAK_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # bf16 MFMA, dense (no sparsity)
I8_MFMA_PEAK_TOPS = 5000.0      # i8 MFMA: 2x the bf16 rate per clock, dense (v_mfma_i32_32x32x32_i8: 32 cycles/SIMD)
I8_MFMA_MEASURED_TOPS = 3500.0  # tools/micro/mfma_rate.hip on this pool: 1.5-1.75 PMAC/s sustained on random
                                # operands (the chip lowers its clock under MFMA load), 2.3 PMAC/s on zeros

METRIC = "QPS @ recall10@10 + achieved HBM GB/s, 1M x 128 f32"


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=500)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default=None, choices=["ah", "bf_dot", "txh"],
                   help="default: ah at N = 1, the leaf-sharded Tree-X-Hybrid at N > 1")
    p.add_argument("--num-points", dest="n", type=int, default=None,
                   help="points (per GPU when sharded); default 1M (ah, bf_dot, txh) / 12.5M per GPU (sharded)")
    p.add_argument("--dim", type=int, default=None)
    p.add_argument("--subspaces", type=int, default=None)
    p.add_argument("--num-codes", type=int, default=16,
                   help="codes per subspace: <= 16 = LUT16 (4-bit), <= 256 = byte codes")
    p.add_argument("--batch", type=int, default=1024)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--pre-reorder-k", type=int, default=None)
    p.add_argument("--leaves", type=int, default=None, help="k-means leaves (per GPU when sharded)")
    p.add_argument("--partitions-to-search", type=int, default=None)
    p.add_argument("--dist", default="uniform", choices=["uniform", "clustered"])
    p.add_argument("--eval-queries", type=int, default=256)
    p.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-recall", action="store_true")
    p.add_argument("--no-batch-sweep", action="store_true")
    p.add_argument("--multi-gpu", default="shard", choices=["shard", "replica"],
                   help="N > 1: the leaf-sharded index through the library's RCCL exchange (default), or "
                        "query-parallel replicas of the N = 1 workload")
    p.add_argument("--no-secondary", action="store_true", help="N > 1 shard: skip the replica measurement")
    p.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    p.add_argument("--force-sharded", action="store_true",
                   help="rehearsal: run the sharded path (library RCCL exchange included) with however many "
                        "ranks there are, even one")
    p.add_argument("--m-local", type=int, default=-1,
                   help="sharded: candidates sent per (rank, query); -1 (default) = all m in compact destination blocks "
                        "(verified; the run repeats with 0 if a block overflows), 0 = all m in worst-case blocks")
    p.add_argument("--kmeans-iters", type=int, default=8)
    p.add_argument("--bf-exact", action="store_true",
                   help="bf_dot: exact f32-MFMA kernels only (no bf16 shortlist)")
    a = p.parse_args()
    a.raw = {key: getattr(a, key) for key in ("workload", "n", "dim", "subspaces", "leaves", "partitions_to_search",
                                              "pre_reorder_k")}
    return resolve_defaults(a)


def resolve_defaults(a):
    """Workload defaults from the launch shape (also called again when the sharded path cannot start and
    the run falls back to replicas: `a.raw` keeps what the command line itself gave)."""
    for key, val in a.raw.items():
        setattr(a, key, val)
    sharded = (a.gpus > 1 or a.force_sharded) and a.multi_gpu == "shard" and a.workload in (None, "txh")
    a.sharded = sharded
    if a.workload is None:
        a.workload = "txh" if sharded else "ah"
    if sharded:      # one shard of BASELINE configs[4] per GPU
        a.n = a.n or 12_500_000
        a.dim = a.dim or 96
        a.subspaces = a.subspaces or 24
        a.leaves = a.leaves or 1250
        a.partitions_to_search = a.partitions_to_search or 10
        a.pre_reorder_k = a.pre_reorder_k or 8192
    else:
        a.n = a.n or 1_000_000
        a.dim = a.dim or 128
        a.subspaces = a.subspaces or 32
        a.leaves = a.leaves or 1000
        a.partitions_to_search = a.partitions_to_search or 50
        a.pre_reorder_k = a.pre_reorder_k or 5000
    return a


def dev_ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def lib_sha256(hip):
    h = hashlib.sha256()
    with open(hip.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def src_sha256():
    from scann_rust_amd import build as hip_build
    return hip_build.src_sha256()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# =====================================================================================================
# index builders
# =====================================================================================================
def clustered_points(torch, device, seed, n, dim, n_clusters=1000, centre_seed=7):
    """Mixture of 1000 Gaussians: centres from `centre_seed` (identical on every rank),
    points from `seed`.  torch on the GPU = harness plumbing."""
    gc = torch.Generator(device=device)
    gc.manual_seed(centre_seed)
    cen = torch.rand((n_clusters, dim), generator=gc, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sig = 0.1 * (1.0 / 6.0) ** 0.5
    return cen[torch.randint(0, n_clusters, (n,), generator=g, device=device)] + \
        sig * torch.randn((n, dim), generator=g, device=device)


def train_residual_codebook(torch, hip, res_sample, dim, S, stride, device_index):
    ns = res_sample.shape[0]
    rs = np.zeros((ns, stride), np.float32)
    rs[:, :dim] = res_sample
    rbf = hip.bf_create(rs, ns, dim, stride, hip.SQUARED_L2, device=device_index)
    dsub = dim // S
    cb = np.zeros((S, 16, dsub), np.float32)
    for sidx in range(S):
        c0 = hip.kmeans_init_pp(rbf, 16, seed=42 + sidx, col_offset=sidx * dsub, sub_dim=dsub)
        cb[sidx] = hip.kmeans_lloyd(rbf, c0, max_iterations=25, col_offset=sidx * dsub)[0]
    rbf.close()
    return cb


def build_txh_single(args, torch, hip, device, local_rank, stride):
    """Tree-X-Hybrid index on the clustered set, built with the library's GPU k-means."""
    n, dim, S, L, Q, k = args.n, args.dim, args.subspaces, args.leaves, args.batch, args.k
    X = clustered_points(torch, device, 11, n, dim)
    nq = max(Q * 4, args.eval_queries)
    Xq = clustered_points(torch, device, 12, nq, dim)
    xs = np.zeros((n, stride), np.float32)
    xs[:, :dim] = X.cpu().numpy()
    bf = hip.bf_create(xs, n, dim, stride, hip.SQUARED_L2, device=local_rank)
    centers, a_np, sizes, _, it_done, _ = hip.kmeans_lloyd(bf, hip.kmeans_init_pp(bf, L, seed=42), max_iterations=25)
    bf.close()
    log("partitioner k-means: %d Lloyd iterations" % it_done)
    assign = torch.from_numpy(a_np.astype(np.int64)).to(device)
    order = torch.argsort(assign, stable=True)
    C = torch.from_numpy(centers).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(5)
    pick = torch.randperm(n, generator=g, device=device)[:min(n, 262144)]
    codebook = train_residual_codebook(torch, hip, (X[pick] - C[assign[pick]]).cpu().numpy(), dim, S, stride,
                                       local_rank)
    leaf_off = np.zeros(L + 1, np.uint32)
    leaf_off[1:] = np.cumsum(sizes)
    order_np = order.cpu().numpy()
    rows_csr = np.zeros((n, stride), np.float32)
    rows_csr[:, :dim] = X[order].cpu().numpy()
    leaf_of_row = np.repeat(np.arange(L, dtype=np.uint32), sizes)
    codes = hip.encode(codebook, rows_csr, stride=stride, centers=centers, leaf_of_row=leaf_of_row, device=local_rank)
    index = hip.txh_create(data=rows_csr, n_rows=n, dim=dim, stride=stride, centers=centers, leaf_offsets=leaf_off,
                           leaf_ids=order_np.astype(np.uint32), codebook=codebook,
                           codes=codes, use_residuals=True, partitions_to_search=args.partitions_to_search,
                           pre_reorder_multiplier=float(args.pre_reorder_k) / k, data_is_csr_order=True,
                           device=local_rank)
    queries = Xq.cpu().numpy().astype(np.float32)
    tok, _, _ = hip.txh_partition(index, np.ascontiguousarray(queries[:Q]), args.partitions_to_search)
    scanned = float(sizes[tok.astype(np.int64)].sum(1).mean())
    full = xs   # rows by datapoint index (the oracle's and the ground truth's view)

    def oracle_index(orc, m, kk):
        return orc.TxhIndex(full, stride, dim, centers, leaf_off, order_np.astype(np.uint32), codebook, codes,
                            use_residuals=True, partitions_to_search=args.partitions_to_search,
                            pre_reorder_multiplier=float(m) / kk)
    return dict(index=index, queries=queries, data=full, codebook=codebook, codes=codes, scanned_points=scanned,
                oracle_index=oracle_index)


def build_txh_shard(args, torch, dist, hip, device, local_rank, rank, world, stride):
    """One BASELINE configs[4]-shaped shard per rank.  Rank g draws its own n points of the common
    mixture and trains its own `leaves` centroids on them (the library's GPU k-means); the global
    partitioner is the concatenation of the ranks' centroid tables (leaf l belongs to rank
    l // leaves), the PQ codebook is trained by rank 0 on its residuals and broadcast.  Centroids,
    codebook and GLOBAL leaf sizes are replicated, so every rank selects the same leaves and forms
    identical merge keys.  Datapoint index = rank * n + row in leaf order."""
    n, dim, S, Lr, Q, k = args.n, args.dim, args.subspaces, args.leaves, args.batch, args.k
    L = Lr * world
    X = clustered_points(torch, device, 100 + rank, n, dim)
    nq = max(Q * 4, args.eval_queries)
    queries = clustered_points(torch, device, 12, nq, dim).cpu().numpy().astype(np.float32)   # same on all ranks
    xs = np.zeros((n, stride), np.float32)
    xs[:, :dim] = X.cpu().numpy()
    bf = hip.bf_create(xs, n, dim, stride, hip.SQUARED_L2, device=local_rank)
    c_loc, a_np, sizes_loc, _, it_done, _ = hip.kmeans_lloyd(bf, hip.kmeans_init_pp(bf, Lr, seed=42 + rank),
                                                             max_iterations=args.kmeans_iters)
    bf.close()
    del xs
    log("shard k-means: %d leaves, %d Lloyd iterations" % (Lr, it_done))
    assign = torch.from_numpy(a_np.astype(np.int64)).to(device)
    order = torch.argsort(assign, stable=True)
    C_loc = torch.from_numpy(c_loc).to(device)
    cb_t = torch.zeros((S, 16, dim // S))
    if rank == 0:
        g = torch.Generator(device=device)
        g.manual_seed(5)
        pick = torch.randperm(n, generator=g, device=device)[:min(n, 262144)]
        cb_t = torch.from_numpy(train_residual_codebook(torch, hip, (X[pick] - C_loc[assign[pick]]).cpu().numpy(),
                                                        dim, S, stride, local_rank))
    cen_all = [torch.zeros((Lr, dim)) for _ in range(world)]
    siz_all = [torch.zeros((Lr,), dtype=torch.int64) for _ in range(world)]
    if world > 1:
        dist.broadcast(cb_t, 0)
        dist.all_gather(cen_all, torch.from_numpy(c_loc))
        dist.all_gather(siz_all, torch.from_numpy(sizes_loc.astype(np.int64)))
    else:
        cen_all, siz_all = [torch.from_numpy(c_loc)], [torch.from_numpy(sizes_loc.astype(np.int64))]
    codebook = cb_t.numpy().astype(np.float32)
    centers = torch.cat(cen_all).numpy().astype(np.float32)                 # [L][dim], replicated
    sizes_global = torch.cat(siz_all).numpy().astype(np.uint32)            # [L]
    local_sizes = np.zeros(L, np.uint32)
    local_sizes[rank * Lr:(rank + 1) * Lr] = sizes_loc
    loc_off = np.zeros(L + 1, np.uint32)
    loc_off[1:] = np.cumsum(local_sizes)
    rows_csr = np.zeros((n, stride), np.float32)
    rows_csr[:, :dim] = X[order].cpu().numpy()
    del X
    torch.cuda.empty_cache()
    leaf_of_row = np.repeat(np.arange(L, dtype=np.uint32), local_sizes)
    codes = hip.encode(codebook, rows_csr, stride=stride, centers=centers, leaf_of_row=leaf_of_row, device=local_rank)
    ids = (np.arange(n, dtype=np.uint64) + np.uint64(rank) * np.uint64(n)).astype(np.uint32)
    index = hip.txh_create(data=rows_csr, n_rows=n, dim=dim, stride=stride, centers=centers, leaf_offsets=loc_off,
                           leaf_ids=ids, leaf_sizes_global=sizes_global, codebook=codebook, codes=codes,
                           use_residuals=True, partitions_to_search=args.partitions_to_search,
                           pre_reorder_multiplier=float(args.pre_reorder_k) / k, data_is_csr_order=True,
                           device=local_rank)
    tok, _, _ = hip.txh_partition(index, np.ascontiguousarray(queries[:Q]), args.partitions_to_search)
    scanned_local = float(local_sizes[tok.astype(np.int64)].sum(1).mean())
    scanned_global = float(sizes_global[tok.astype(np.int64)].sum(1).mean())
    return dict(index=index, queries=queries, data=rows_csr, ids=ids, codebook=codebook, codes=codes,
                scanned_points=scanned_local, scanned_points_global=scanned_global, L=L)


# =====================================================================================================
# roofline of the dominant kernel
# =====================================================================================================
def traffic_for(hip, workload, kernel_name):
    """PMC-measured HBM bytes per launch from profiles/traffic.json -- only when the entry was recorded
    for THIS kernel of THESE sources of the library (keyed by kernel name + sha256 over csrc/, the headers and the
    compile flags: scann_rust_amd/build.py src_sha256 -- a rebuild in another directory keeps the key, an edit of any
    kernel drops it); anything else would be a stale constant, so null."""
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        ent = json.load(open(tr)).get(workload)
        if isinstance(ent, dict) and ent.get("kernel") == kernel_name and ent.get("src_sha256") == src_sha256():
            return {"bytes": ent["bytes"], "source": ent.get("source")}
    except Exception:
        pass
    return None


def scan_roofline(hip, workload, kernel_name, kernel_ms, scanned_points, S, K, Q, k, pairs_per_query):
    """LUT16 / byte-code ADC scan.  Binding resource of the batched scan = the LDS table gather: one
    ds_read_b128 (16 B) per (point, subspace, quad of 4 queries), i.e. scanned_points * S * 4 B per
    query (MI355X_MICROARCH.md section LDS: 256 B/clk/CU).  The SURVEY 8(d) HBM figure is reported
    beside it as ALGORITHMIC bytes: the codes are read once per tile and shared by the whole batch, so
    it is not physical traffic and is not a fraction of the HBM roof."""
    code_bytes = S // 2 if K <= 16 else S
    algo_bytes = scanned_points * code_bytes + pairs_per_query * S * (16 if K <= 16 else 256) * 4 + k * 8
    t = kernel_ms * 1e-3
    if kernel_name == "adc_smfmac_kernel":
        # The prefilter on the 2:4-sparse MFMA: a 32 x 32 tile issues 2 dense v_mfma_i32_32x32x32_i8 (subspaces 0..3)
        # and (S - 4) / 4 v_smfmac_i32_32x32x64_i8 (four subspaces each), every one of them one 32-cycle slot of the
        # matrix pipe.  achieved = ISSUED slots x the dense slot's 2 x 32^3 integer op -- the share of the dense-i8
        # pipe rate the kernel keeps busy (the sparse instruction's doubled K is not counted: it is the same slot);
        # the one-hot product those slots evaluate is reported beside it.
        slots = scanned_points / 32.0 * Q / 32.0 * (2.0 + (S - 4) / 4.0)
        ops = slots * 2.0 * 32768.0
        ach = ops / t / 1e12 if t else 0.0
        dense_equiv = 2.0 * scanned_points * S * 16.0 * Q
        roof = {"bound": "mfma", "achieved": ach, "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                "frac": ach / I8_MFMA_PEAK_TOPS, "frac_of_measured_peak": ach / I8_MFMA_MEASURED_TOPS,
                "peak_measured": I8_MFMA_MEASURED_TOPS, "traffic": None, "kernel": kernel_name,
                "kernel_ms": kernel_ms,
                "useful_lookups_per_s": scanned_points * S * Q / t if t else 0.0,
                "one_hot_product_TOPs": dense_equiv / t / 1e12 if t else 0.0,
                "algorithmic": "matrix-pipe slots: scanned points (%.0f) / 32 x %d queries / 32 x (2 dense + %d sparse MFMAs "
                               "per tile) = %.4g slots x 2 x 32^3 integer op = %.4g op per launch; peak = dense i8 MFMA "
                               "(2 x the 2.5 PFLOP/s bf16 rate); peak_measured = sustained rate of either instruction on "
                               "random operands (tools/micro/mfma_rate.hip, smfmac_probe.hip); the one-hot product "
                               "evaluated is 2 x points x S*16 x queries = %.4g op (one_hot_product_TOPs), of which "
                               "points x S x queries table lookups are useful (useful_lookups_per_s)"
                               % (scanned_points, Q, (S - 4) // 4, slots, ops, dense_equiv)}
    elif kernel_name in ("adc_mfma_kernel", "adc_mfma16_kernel"):
        # integer-MFMA prefilter: one-hot(codes) [points x S*16] x u8 tables [S*16 x queries]; every
        # (point, query) costs S*16 multiply-adds on the matrix cores (16x the useful table adds: the price
        # of turning a gather into a product).  Bound: the i8 MFMA rate.
        ops = 2.0 * scanned_points * S * 16.0 * Q
        ach = ops / t / 1e12 if t else 0.0
        roof = {"bound": "mfma", "achieved": ach, "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                "frac": ach / I8_MFMA_PEAK_TOPS, "frac_of_measured_peak": ach / I8_MFMA_MEASURED_TOPS,
                "peak_measured": I8_MFMA_MEASURED_TOPS, "traffic": None, "kernel": kernel_name,
                "kernel_ms": kernel_ms,
                "algorithmic": "one-hot product: 2 x scanned points (%.0f) x S*16 (%d) x %d queries per launch = "
                               "%.4g integer op; peak = dense i8 MFMA (2 x the 2.5 PFLOP/s bf16 rate); "
                               "peak_measured = sustained v_mfma_i32_32x32x32_i8 rate on random operands "
                               "(tools/micro/mfma_rate.hip)" % (scanned_points, S * 16, Q, ops)}
    else:
        lds_bytes = scanned_points * S * 4.0 * Q
        ach = lds_bytes / t / 1e9 if t else 0.0
        roof = {"bound": "lds", "achieved": ach, "peak": LDS_PEAK_GBPS, "unit": "GB/s", "frac": ach / LDS_PEAK_GBPS,
                "frac_of_measured_peak": ach / LDS_MEASURED_GBPS, "peak_measured": LDS_MEASURED_GBPS,
                "traffic": None, "kernel": kernel_name, "kernel_ms": kernel_ms,
                "algorithmic": "LDS gather: scanned points (%.0f) x S (%d) x 16 B per quad of 4 queries x %d queries "
                               "per launch = %.4g B; peak = ds_read_b128 256 B/clk/CU x 256 CUs x 2.4 GHz"
                               % (scanned_points, S, Q, lds_bytes)}
    tr = traffic_for(hip, workload, kernel_name)
    if tr:
        roof["traffic"] = tr["bytes"]
        roof["traffic_source"] = tr["source"]
    algo = {"bytes_per_query": algo_bytes, "GBps": algo_bytes * Q / t / 1e9 if t else 0.0,
            "x_hbm_peak": (algo_bytes * Q / t / 1e9 / HBM_PEAK_GBPS) if t else 0.0,
            "note": "SURVEY 8(d) algorithmic bytes (scanned codes + LUTs + out) x queries / kernel time; the batch "
                    "shares one read of the codes, so this may exceed the 8 TB/s HBM roof and is not a roofline "
                    "fraction -- physical HBM bytes are `roofline.traffic`"}
    return roof, algo


# =====================================================================================================
def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from scann_rust_amd import hip, synth, trainer

    rank = int(os.environ.get("RANK", "0"))
    nproc = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if nproc != args.gpus and nproc == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if nproc > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (rendezvous, barriers, max over ranks, index broadcast): the data path's
        # collectives are the library's own RCCL calls
        dist.init_process_group("gloo")
    sharded = args.sharded and (nproc > 1 or args.force_sharded)
    replica = nproc > 1 and not sharded

    L = hip.load()
    n, dim, S, K, k, Q = args.n, args.dim, args.subspaces, args.num_codes, args.k, args.batch
    m = args.pre_reorder_k
    stride = hip.compute_stride(dim)
    t0 = time.time()

    # ---------------- index ------------------------------------------------------------------------
    st = None
    data = codebook = codes = None
    flops_per_query = None
    comm = None
    shard_fallback = None
    if sharded:
        # The communicator first: if RCCL cannot be brought up on this node (it has never seen this code
        # with more than one rank before the driver's scaling run), every rank falls back to the replica
        # layout together and the JSON line says so -- a number with a note instead of a dead run.
        err = ""
        # Step 1, every rank the same collectives whatever fails: rank 0 makes the id (or fails to: RCCL not loadable)
        # and broadcasts [ok, id]; a failed rank 0 still broadcasts, so no peer is left waiting in the broadcast.
        # SCANN_BENCH_FAIL_COMM rehearses the fallback: "rank0" fails the id on rank 0 only, anything else fails the
        # communicator on every rank.
        fail_mode = os.environ.get("SCANN_BENCH_FAIL_COMM", "")
        msg = torch.zeros(129, dtype=torch.uint8)
        if rank == 0:
            try:
                if fail_mode == "rank0":
                    raise RuntimeError("forced by SCANN_BENCH_FAIL_COMM=rank0")
                msg[1:] = torch.frombuffer(bytearray(hip.Comm.unique_id()), dtype=torch.uint8)
                msg[0] = 1
            except Exception as e:   # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, e)
        if nproc > 1:
            dist.broadcast(msg, 0)
        if int(msg[0].item()) == 0:
            err = err or "rank 0 could not create the RCCL unique id"
        else:
            try:
                if fail_mode and fail_mode != "rank0":
                    raise RuntimeError("forced by SCANN_BENCH_FAIL_COMM")
                # RCCL prints a version banner on STDOUT when the first communicator is created; rank 0's stdout
                # must carry exactly one JSON line, so the banner is sent to stderr
                sys.stdout.flush()
                saved_stdout = os.dup(1)
                os.dup2(2, 1)
                try:
                    comm = hip.Comm(msg[1:].numpy().tobytes(), rank, nproc, device=local_rank)
                finally:
                    sys.stdout.flush()
                    os.dup2(saved_stdout, 1)
                    os.close(saved_stdout)
            except Exception as e:   # noqa: BLE001 (any failure of the bring-up takes the fallback)
                err = "%s: %s" % (type(e).__name__, e)
        ok = torch.tensor([0 if err else 1], dtype=torch.int32)
        if nproc > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0].item()) == 0:
            shard_fallback = err or "another rank could not create the RCCL communicator"
            print("[bench] rank %d: sharded path unavailable (%s); falling back to replicas" % (rank, shard_fallback),
                  file=sys.stderr, flush=True)
            comm = None
            args.multi_gpu = "replica"
            resolve_defaults(args)
            sharded = False
            replica = nproc > 1
            n, dim, S, m = args.n, args.dim, args.subspaces, args.pre_reorder_k
            stride = hip.compute_stride(dim)
    if sharded:
        st = build_txh_shard(args, torch, dist, hip, device, local_rank, rank, nproc, stride)
        index, queries_all = st["index"], st["queries"]
    elif args.workload == "txh":
        st = build_txh_single(args, torch, hip, device, local_rank, stride)
        index, queries_all, data, codebook, codes = st["index"], st["queries"], st["data"], st["codebook"], st["codes"]
    else:
        if args.dist == "uniform":
            rows = synth.uniform_f32(n, dim, 42)
            queries_all = synth.uniform_f32(max(Q * 4, args.eval_queries), dim, 123)
        else:
            rows, _ = synth.clustered_f32(n, dim, 7, n_clusters=1000)
            queries_all, _ = synth.clustered_f32(max(Q * 4, args.eval_queries), dim, 8, n_clusters=1000)
        data = np.zeros((n, stride), np.float32)
        data[:, :dim] = rows
        if args.workload == "bf_dot":
            index = hip.bf_create(data, n, dim, stride, hip.DOT_PRODUCT, device=local_rank)
            flops_per_query = 2.0 * n * dim
        else:
            sample = rows[:: max(1, n // 65536)] if args.dist != "uniform" else \
                synth.uniform_rows((synth.splitmix64(0xC0DE, 0, 65536) % np.uint64(n)).astype(np.int64), dim, 42)
            codebook = trainer.train_codebook(sample, S, K, iters=25, seed=42, sample=1 << 30)
            codes = hip.encode(codebook, data, stride=stride, device=local_rank)
            index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=None, leaf_offsets=None,
                                   leaf_ids=None, codebook=codebook, codes=codes, use_residuals=False,
                                   partitions_to_search=1, pre_reorder_multiplier=float(m) / k, device=local_rank)
        del rows
    if replica and rank:                # every replica searches its own queries
        queries_all = np.roll(queries_all, -rank * Q, axis=0)
    log("index built in %.1fs" % (time.time() - t0))

    workload_name = {
        "ah": "AsymmetricHasher %s S=%d K=%d + exact re-rank" % ("LUT16" if K <= 16 else "byte codes", S, K),
        "bf_dot": "BruteForceSearcher.search_batched DotProduct (bf16-MFMA shortlist + exact f32 re-score, verified)"
                  if not args.bf_exact else "BruteForceSearcher.search_batched DotProduct (f32 MFMA)",
        "txh": "Tree-X-Hybrid L=%d P=%d LUT16 S=%d + exact re-rank"
               % (args.leaves * (nproc if sharded else 1), args.partitions_to_search, S)}[args.workload]

    opts = hip.default_opts()
    if args.workload != "bf_dot":
        opts.pre_reorder_k = m
        opts.exact_reorder = 1
    if args.workload == "txh":
        opts.partitions_to_search = args.partitions_to_search
    opts.bf_exact = 1 if args.bf_exact else 0

    # ---------------- device buffers (inputs resident in HBM before the timed region) ----------------
    nbatches = 4
    qdev = [torch.from_numpy(np.ascontiguousarray(queries_all[i * Q:(i + 1) * Q])).to(device) for i in range(nbatches)]
    n_streams = max(1, int(os.environ.get("SCANN_BENCH_STREAMS", "2")))   # caller streams that alternate (default two)
    n_bufs = max(2, n_streams)
    streams = [torch.cuda.Stream(device) for _ in range(n_bufs)]
    outs = [(torch.empty((Q, k), dtype=torch.int32, device=device), torch.empty((Q, k), dtype=torch.float32, device=device),
             torch.empty((Q,), dtype=torch.int32, device=device)) for _ in range(n_bufs)]
    hip.check(L.scann_hip_index_reserve(index.h, Q, k, ctypes.byref(opts)))

    two_streams = n_streams != 1
    if sharded and args.m_local < 0:
        args.m_local = m   # every rank's full list, compact destination blocks

    def step(i, qd=None, nq=Q):
        qd = qdev[i % nbatches] if qd is None else qd
        b = i % n_bufs
        oi, od, oc = outs[b]
        # Two caller streams alternate (SCANN_BENCH_STREAMS=1: one): the library binds a workspace to each caller stream
        # (scann_hip.h "device entry points and streams"), so step i+1's matrix-core-bound scan runs under step i's
        # HBM-/latency-bound select + re-rank kernels; sharded: step i's exchange overlaps step i+1's local stage.
        sp = ctypes.c_void_p(streams[i % n_streams if two_streams else 0].cuda_stream)
        if sharded:
            # two caller streams alternate: step i's exchange (on the library's stream) overlaps
            # step i+1's local stage (the library orders its own buffers with events)
            hip.check(L.scann_hip_txh_search_sharded_device(index.h, comm.h, dev_ptr(qd), nq, dim, k,
                                                            ctypes.byref(opts), args.m_local, dev_ptr(oi),
                                                            dev_ptr(od), dev_ptr(oc), sp))
        else:
            hip.check(L.scann_hip_search_batched_device(index.h, dev_ptr(qd), nq, dim, k, ctypes.byref(opts),
                                                        dev_ptr(oi), dev_ptr(od), dev_ptr(oc), sp))

    def device_status():
        if sharded:
            comm.last_status()
        for st_ in streams[:n_streams if two_streams else 1]:
            hip.check(L.scann_hip_index_last_device_status(index.h, ctypes.c_void_p(st_.cuda_stream)))

    def barrier():
        if nproc > 1:
            dist.barrier()

    # ---------------- warmup, then EXACTLY K timed steps ---------------------------------------------
    bf_retry = False
    while True:
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        try:
            device_status()
        except hip.ScannError as e:
            if args.workload == "bf_dot" and e.code == 10 and not bf_retry:
                log("a bf16-shortlist result could not be verified; repeating with the exact kernels")
                bf_retry, opts.bf_exact = True, 1
                continue
            if sharded and e.code == 10 and args.m_local:
                log("m_local=%d: a list was too short or a compact block overflowed; repeating with worst-case blocks" % args.m_local)
                args.m_local = 0
                continue
            raise
        index.enable_timing(True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms_timed, kernel_name = index.last_kernel_ms()
        index.enable_timing(False)
        device_status()
        # rows of the LAST TIMED step, for the oracle check (the passes below reuse the buffers)
        last_out = tuple(t.clone() for t in outs[(args.steps - 1) % n_bufs])
        # The dominant kernel ALONE on the machine, for the roofline: with two caller streams the timed region runs
        # kernels of consecutive batches side by side, so an event pair around one of them also measures its neighbours
        # (C3: 0.52 ms inside the region, 0.35 ms alone; rocprofv3 --kernel-trace agrees with each in its own run).  A
        # short single-stream pass after the timed region, same index, same batches, HIP events on the launch stream.
        kernel_ms = kernel_ms_timed
        if two_streams and not sharded:
            index.enable_timing(True)
            two_streams = False
            for i in range(min(args.steps, 40)):
                step(i)
            torch.cuda.synchronize()
            kernel_ms, _ = index.last_kernel_ms()
            index.enable_timing(False)
            two_streams = True
        break
    if nproc > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
    if elapsed < 0.2:
        log("WARNING: the timed region is only %.3f s (%d steps); raise --steps for a stable number" % (elapsed, args.steps))
    qps = Q * args.steps * (nproc if replica else 1) / elapsed
    last_q = queries_all[((args.steps - 1) % nbatches) * Q:][:Q]

    # ---------------- batch-size sweep (batches {1, 100, 1024}, >= 10 reps, median) -----------
    sweep = None
    if not args.no_batch_sweep and not replica:
        sweep = {}
        for bsz in (1, 100, Q):
            ts = []
            qd = qdev[0][:bsz].contiguous()
            for r in range(3 + 12):
                barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                step(r, qd, bsz)
                torch.cuda.synchronize()
                if r >= 3:
                    ts.append(time.perf_counter() - t1)
            med = statistics.median(ts)
            if nproc > 1:
                t = torch.tensor([med], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                med = float(t[0].item())
            sweep[str(bsz)] = {"ms_per_call": med * 1e3, "queries_per_s": bsz / med}
        device_status()

    # ---------------- recall10@10 (bin/ann_benchmark.rs:427-471 semantics) ------------------------------
    recall = checked = None
    ne = min(args.eval_queries, queries_all.shape[0])
    qe = np.ascontiguousarray(queries_all[:ne])
    if not args.no_recall and args.workload != "bf_dot" and not replica:
        if sharded:
            # ground truth = exact brute force over every rank's shard (library kernels), merged on rank 0
            qt = torch.from_numpy(qe).to(device)
            gi = torch.empty((ne, k), dtype=torch.int32, device=device)
            gd = torch.empty((ne, k), dtype=torch.float32, device=device)
            gc = torch.empty((ne,), dtype=torch.int32, device=device)
            for a0 in range(0, ne, Q):
                nb = min(Q, ne - a0)
                hip.check(L.scann_hip_txh_search_sharded_device(
                    index.h, comm.h, dev_ptr(qt[a0:a0 + nb]), nb, dim, k, ctypes.byref(opts), 0, dev_ptr(gi[a0:a0 + nb]),
                    dev_ptr(gd[a0:a0 + nb]), dev_ptr(gc[a0:a0 + nb]), ctypes.c_void_p(streams[0].cuda_stream)))
            torch.cuda.synchronize()
            comm.last_status()
            bf = hip.bf_create(st["data"], n, dim, stride, hip.SQUARED_L2, device=local_rank)
            ti, td, _ = bf.search_batched(qe, k)
            bf.close()
            tid = st["ids"][ti.astype(np.int64)].astype(np.int64)         # global datapoint indices
            parts_d = [torch.zeros((ne, k)) for _ in range(nproc)]
            parts_i = [torch.zeros((ne, k), dtype=torch.int64) for _ in range(nproc)]
            if nproc > 1:
                dist.all_gather(parts_d, torch.from_numpy(td.astype(np.float32)))
                dist.all_gather(parts_i, torch.from_numpy(tid))
            else:
                parts_d, parts_i = [torch.from_numpy(td.astype(np.float32))], [torch.from_numpy(tid)]
            if rank == 0:
                ad = torch.cat(parts_d, 1).numpy()
                ai = torch.cat(parts_i, 1).numpy()
                o2 = np.argsort(ad, axis=1, kind="stable")[:, :k]
                truth = np.take_along_axis(ai, o2, 1)
                got = gi.cpu().numpy().view(np.uint32).astype(np.int64)
                recall = sum(len(set(got[i].tolist()) & set(truth[i].tolist())) for i in range(ne)) / float(ne * k)
                # exact distances of the sharded rows must be ascending and bit-equal to the brute-force
                # distances of the same points wherever the two lists share a point
                gdn = gd.cpu().numpy()
                tdn = np.take_along_axis(ad, o2, 1)
                ok = bool(np.all(np.diff(gdn, axis=1) >= 0))
                for i in range(ne):
                    pos = {int(p): j for j, p in enumerate(truth[i])}
                    for j in range(k):
                        if int(got[i, j]) in pos:
                            ok = ok and gdn[i, j].view(np.uint32) == tdn[i, pos[int(got[i, j])]].view(np.uint32)
                checked = bool(ok)
        elif rank == 0:
            gi, gd, gc = index.search_batched(qe, k, opts)
            bf = hip.bf_create(data, n, dim, stride, hip.SQUARED_L2, device=local_rank)
            ti, td, tc = bf.search_batched(qe, k)
            recall = sum(len(set(gi[i].tolist()) & set(ti[i].tolist())) for i in range(ne)) / float(ne * k)
            bf.close()
    if rank == 0 and nproc == 1 and not sharded:
        # result rows of the LAST TIMED step checked against the oracle (checker only)
        from oracle import pyoracle as orc
        gi = last_out[0].cpu().numpy().view(np.uint32)
        gd = last_out[1].cpu().numpy()
        ok = True
        for i in range(min(4, Q)):
            if args.workload == "bf_dot":
                oi, od = orc.bf_search(data, n, dim, stride, orc.DOT_PRODUCT, last_q[i], k)
            elif args.workload == "txh":
                oi, od = orc.txh_search(st["oracle_index"](orc, m, k), last_q[i], k)
            else:
                oi, od = orc.ah_search_with_reordering(codebook, codes, data, stride, last_q[i], k, m)
            ok = ok and np.array_equal(gd[i].view(np.uint32), od.view(np.uint32)) \
                and sorted(gi[i].tolist()) == sorted(oi.tolist())
        checked = bool(ok)

    # ---------------- CPU baseline: the oracle on this box's host cores (rank 0, N = 1) -----------------
    cpu = None
    if rank == 0 and nproc == 1 and not sharded and not args.no_cpu_baseline:
        from oracle import pyoracle as orc
        threads = orc.max_threads()

        def run_cpu(qb, nthreads):
            t1 = time.perf_counter()
            if args.workload == "bf_dot":
                orc.bf_search_batched(data, n, dim, stride, orc.DOT_PRODUCT, qb, k, nthreads)
            elif args.workload == "txh":
                orc.txh_search_batched(st["oracle_index"](orc, m, k), qb, k, nthreads)
            else:
                orc.ah_search_batched(codebook, codes, data, stride, qb, k, m, True, nthreads)
            return time.perf_counter() - t1

        # one thread, sequential queries (bin/ann_benchmark.rs:172-178): a probe query sizes the sample
        t_one = run_cpu(np.ascontiguousarray(queries_all[:1]), 1)
        n1 = int(max(1, min(32, 0.25 * args.cpu_baseline_seconds / max(t_one, 1e-4))))
        t_seq = run_cpu(np.ascontiguousarray(queries_all[:n1]), 1)
        # all host threads, one task per query (tree_x_hybrid/mod.rs:404-408, brute_force/searcher.rs:204-207)
        nq1 = min(threads, queries_all.shape[0])
        t_run = run_cpu(np.ascontiguousarray(queries_all[:nq1]), threads)
        if 2.0 * t_run < 0.75 * args.cpu_baseline_seconds:   # cheap workload: a longer sample
            reps = int(max(2, min(64, 0.75 * args.cpu_baseline_seconds / max(t_run, 1e-3))))
            nq1 = min(queries_all.shape[0], nq1 * reps)
            t_run = run_cpu(np.ascontiguousarray(queries_all[:nq1]), threads)
        cpu = {"value": nq1 / t_run, "unit": "queries/s", "cores": threads, "kind": "port",
               "cpu_model": cpu_model(),
               "sample": "%d queries of the same workload (same index, k, pre_reorder_k), one OpenMP task per "
                         "query on %d threads, %.1f s" % (nq1, threads, t_run),
               "single_thread": {"value": n1 / t_seq, "unit": "queries/s", "cores": 1,
                                 "sample": "%d sequential queries, %.1f s" % (n1, t_seq)},
               "label": "CPU restatement of the reference path (oracle/; the Rust toolchain is unavailable)"}

    # ---------------- the replica layout as a secondary number of the sharded run ---------------------
    secondary = None
    if sharded and nproc > 1 and not args.no_secondary:
        secondary = replica_secondary(args, torch, dist, hip, synth, trainer, device, local_rank, rank, nproc)

    if rank == 0:
        algo = None
        if args.workload == "bf_dot" and kernel_name == "bf_stream_kernel":
            passes = (Q + 7) // 8
            by = n * dim * 4
            achieved = by * passes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": kernel_name, "kernel_ms": kernel_ms,
                    "algorithmic": "N*d*4 = %d B per database pass x %d passes (8 queries each) per launch" % (by, passes)}
        elif args.workload == "bf_dot":
            achieved = flops_per_query * Q / (kernel_ms * 1e-3) / 1e12 if kernel_ms else 0.0
            peak = BF16_MFMA_PEAK_TFLOPS if kernel_name == "bf_bf16_kernel" else F32_MFMA_PEAK_TFLOPS
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": None, "kernel": kernel_name, "kernel_ms": kernel_ms,
                    "algorithmic": "2*N*d flop per query x %d queries per launch (%s)"
                                   % (Q, "bf16 MFMA shortlist pass; the shortlisted rows are re-scored with the "
                                         "reference's f32 arithmetic and the result is verified"
                                      if kernel_name == "bf_bf16_kernel" else "f32 MFMA, exact")}
            tr = traffic_for(hip, args.workload, kernel_name)
            if tr:
                roof["traffic"], roof["traffic_source"] = tr["bytes"], tr["source"]
        else:
            scanned = float(n) if args.workload == "ah" else st["scanned_points"]
            pairs = 1 if args.workload == "ah" else args.partitions_to_search
            roof, algo = scan_roofline(hip, "txh_sharded" if sharded else args.workload, kernel_name, kernel_ms,
                                       scanned, S, K, Q, k, pairs)
        if roof["frac"] > 1.0:
            roof["warning"] = "fraction above 1: the assumed bound is not the binding one"
        if roof is not None:
            roof["kernel_ms_in_timed_region"] = kernel_ms_timed
            roof["timing"] = ("kernel_ms: HIP events around the kernel on its launch stream, mean over a single-stream pass "
                              "run right after the timed region (the kernel alone on the machine); kernel_ms_in_timed_region: "
                              "the same events inside the timed region, where two caller streams overlap consecutive batches")
        line = {
            "metric": METRIC, "value": qps, "unit": "queries/s", "n_gpus": nproc, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "timed_region_s": elapsed,
            "higher_is_better": True,
            "scaling": "weak" if (replica or nproc == 1) else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name, "n": n * (nproc if sharded else 1), "n_per_gpu": n, "dim": dim,
                       "k": k, "batch": Q, "global_batch": Q * (nproc if replica else 1), "caller_streams": n_streams,
                       "pre_reorder_k": m if args.workload != "bf_dot" else None,
                       "pre_reorder_k_per_rank": (args.m_local or m) if sharded else None,
                       "exchange_bytes_per_link_per_step": (hip.comm_layout(Q, nproc, args.m_local, k)["block_bytes"]
                                                            if args.m_local else Q // nproc * m * 16) if sharded else None,
                       "distribution": args.dist if args.workload != "txh" else "clustered (1000 Gaussians)",
                       "leaves": (args.leaves * (nproc if sharded else 1)) if args.workload == "txh" else None,
                       "partitions_to_search": args.partitions_to_search if args.workload == "txh" else None,
                       "scanned_points_per_query": (st["scanned_points_global"] if sharded else
                                                    st["scanned_points"]) if st else float(n),
                       "recall10@10": recall, "oracle_check": checked,
                       "parallelism": ("1 process/GPU; leaf-sharded index (one 12.5M x 96-shaped shard per GPU), "
                                       "library entry point scann_hip_txh_search_sharded_device: local stage -> ONE "
                                       "RCCL all-to-all (grouped ncclSend/ncclRecv) -> merge -> ncclAllGather; "
                                       "work per step fixed and split over %d ranks" % nproc) if sharded else
                                      ("1 process/GPU, %d query-parallel replicas of the index (no data-path "
                                       "collective)" % nproc) if replica else "single GPU"},
            "roofline": roof, "algorithmic_hbm": algo, "cpu_baseline": cpu, "batch_sweep": sweep,
            "secondary": secondary, "lib_sha256": lib_sha256(hip), "src_sha256": src_sha256(),
        }
        if shard_fallback:
            line["config"]["note"] = ("the leaf-sharded layout could not start (%s): replica layout measured "
                                      "instead" % shard_fallback)
        print(json.dumps(line), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if nproc > 1:
        dist.barrier()
        dist.destroy_process_group()


def replica_secondary(args, torch, dist, hip, synth, trainer, device, local_rank, rank, nproc):
    """N query-parallel copies of the N = 1 headline workload (AH LUT16 1M x 128): each rank searches its
    own batch, no data-path collective; reported beside the sharded number."""
    n, dim, S, K, k, Q, m = 1_000_000, 128, 32, 16, args.k, args.batch, 5000
    stride = hip.compute_stride(dim)
    rows = synth.uniform_f32(n, dim, 42)
    sample = synth.uniform_rows((synth.splitmix64(0xC0DE, 0, 65536) % np.uint64(n)).astype(np.int64), dim, 42)
    codebook = trainer.train_codebook(sample, S, K, iters=25, seed=42, sample=1 << 30)
    codes = hip.encode(codebook, rows, stride=stride, device=local_rank)
    index = hip.txh_create(data=rows, n_rows=n, dim=dim, stride=stride, centers=None, leaf_offsets=None,
                           leaf_ids=None, codebook=codebook, codes=codes, use_residuals=False,
                           partitions_to_search=1, pre_reorder_multiplier=float(m) / k, device=local_rank)
    L = hip.load()
    o = hip.default_opts()
    o.pre_reorder_k, o.exact_reorder = m, 1
    q = torch.from_numpy(np.roll(synth.uniform_f32(Q * 4, dim, 123), -rank * Q, axis=0)[:Q].copy()).to(device)
    oi = torch.empty((Q, k), dtype=torch.int32, device=device)
    od = torch.empty((Q, k), dtype=torch.float32, device=device)
    oc = torch.empty((Q,), dtype=torch.int32, device=device)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    hip.check(L.scann_hip_index_reserve(index.h, Q, k, ctypes.byref(o)))
    steps = 100

    def run(cnt):
        for _ in range(cnt):
            hip.check(L.scann_hip_search_batched_device(index.h, dev_ptr(q), Q, dim, k, ctypes.byref(o), dev_ptr(oi),
                                                        dev_ptr(od), dev_ptr(oc), sp))
    run(5)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    hip.check(L.scann_hip_index_last_device_status(index.h, sp))
    index.close()
    return {"layout": "replica", "workload": "AsymmetricHasher LUT16 S=32 K=16 + exact re-rank, 1M x 128 per GPU",
            "value": Q * steps * nproc / float(t[0].item()), "unit": "queries/s", "steps": steps,
            "note": "N independent copies of the N = 1 headline workload, one batch per rank per step"}


if __name__ == "__main__":
    main()
