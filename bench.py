#!/usr/bin/env python3
"""bench.py -- headline measurement of the MI355X ScaNN hot path.

    python bench.py --gpus N --steps K --warmup W [--workload ah|bf_dot|txh] ...

Metric (BASELINE.json): QPS @ recall10@10 + achieved HBM GB/s, 1M x 128 f32.
Default workload `ah` = BASELINE.json configs[2]: AsymmetricHasher LUT16 (4-bit PQ,
32 blocks x 16 centres) over 1M x 128 uniform U[0,1) vectors, k = 10, with
search_with_reordering (exact f32 re-rank of the pre_reorder_k best approximate
candidates).  A "step" is one search_batched call over one batch of `--batch` queries
already resident in HBM; value = queries/s of the whole job.

One process per GPU.  N > 1 (launched by torch.distributed.run), two layouts:

  --multi-gpu replica (default): the index of this configuration (1M x 128: 0.53 GB) fits one GPU
    many times over, so the data-parallel unit is the QUERY -- the reference's own parallelism
    (search_batched = one rayon task per query, tree_x_hybrid/mod.rs:404-408).  Every rank holds
    the full index and searches its own batch of `--batch` queries per step; no data-path
    collective.  Weak scaling: per-GPU work fixed, value = N * batch * steps / max-over-ranks time.
  --multi-gpu shard: for indexes that do not fit (BASELINE configs[4]).  The database is sharded
    across ranks (leaves; row ranges = leaves of a flat partition for the hasher), every rank scans
    its shard for all queries, ONE RCCL all_to_all per step sends each peer the (merge key, index,
    exact distance) triples of the queries that peer merges (xGMI is point-to-point: 1/world of
    the all_gather volume per link), each rank merges its batch/world queries and the k result
    rows are all_gathered (strong scaling: total work fixed).  Steps are software-pipelined: a
    step's exchange overlaps the next step's local stage.

The CPU oracle is used here ONLY as (a) the checker of a few result rows and (b) the
cpu_baseline leg; the timed path is libscann_hip.so through its C ABI.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (no sparsity)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="ah", choices=["ah", "bf_dot", "txh"])
    p.add_argument("--num-points", dest="n", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--subspaces", type=int, default=32)
    p.add_argument("--num-codes", type=int, default=16,
                   help="codes per subspace: <= 16 = LUT16 (4-bit), <= 256 = byte codes")
    p.add_argument("--batch", type=int, default=1024)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--pre-reorder-k", type=int, default=5000)
    p.add_argument("--leaves", type=int, default=1000)
    p.add_argument("--partitions-to-search", type=int, default=50)
    p.add_argument("--dist", default="uniform", choices=["uniform", "clustered"])
    p.add_argument("--eval-queries", type=int, default=256)
    p.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-recall", action="store_true")
    # rehearsal knobs (not used by the driver): gloo collectives / all ranks on one device
    p.add_argument("--multi-gpu", default="replica", choices=["replica", "shard"],
                   help="N > 1: query-parallel replicas of the index (default) or a leaf-sharded index "
                        "with one RCCL all_to_all per step")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    p.add_argument("--single-device", action="store_true")
    p.add_argument("--bf-exact", action="store_true",
                   help="bf_dot: exact f32-MFMA kernels only (no bf16 shortlist)")
    return p.parse_args()


def dev_ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def build_txh(args, torch, dist, hip, synth, trainer, device, local_rank, rank, world, stride):
    """Tree-X-Hybrid index on a clustered synthetic set (mixture of 1000 Gaussians, SURVEY.md
    8d).  The data is generated with torch on the GPU (harness plumbing); rank 0 builds the index
    with the library's GPU k-means and broadcasts it so every rank holds the identical index, then
    keeps only its leaves."""
    from scann_rust_amd import sharding
    n, dim, S, L, Q, k = args.n, args.dim, args.subspaces, args.leaves, args.batch, args.k
    g = torch.Generator(device=device)
    g.manual_seed(7)
    cen = torch.rand((1000, dim), generator=g, device=device)
    sig = 0.1 * (1.0 / 6.0) ** 0.5
    X = cen[torch.randint(0, 1000, (n,), generator=g, device=device)] + \
        sig * torch.randn((n, dim), generator=g, device=device)
    nq = max(Q * 4, args.eval_queries)
    Xq = cen[torch.randint(0, 1000, (nq,), generator=g, device=device)] + \
        sig * torch.randn((nq, dim), generator=g, device=device)
    C = torch.empty((L, dim), device=device)
    assign = torch.empty((n,), dtype=torch.int64, device=device)
    cb_t = torch.empty((S, 16, dim // S), device=device)
    if rank == 0:
        # index build with the library's own GPU k-means (KMeans::fit of trees/kmeans.rs: k-means++
        # seeding + Lloyd, scann_hip_kmeans_*): partitioner on all rows, codebook on a residual sample
        xs = np.zeros((n, stride), np.float32)
        xs[:, :dim] = X.cpu().numpy()
        bf = hip.bf_create(xs, n, dim, stride, hip.SQUARED_L2, device=local_rank)
        c_np, a_np, _, _, it_done, _ = hip.kmeans_lloyd(bf, hip.kmeans_init_pp(bf, L, seed=42),
                                                        max_iterations=25)
        bf.close()
        del xs
        C.copy_(torch.from_numpy(c_np).to(device))
        assign.copy_(torch.from_numpy(a_np.astype(np.int64)).to(device))
        res = X - C[assign]
        ns = min(n, 262144)
        rs = np.zeros((ns, stride), np.float32)
        rs[:, :dim] = res[torch.randperm(n, generator=g, device=device)[:ns]].cpu().numpy()
        rbf = hip.bf_create(rs, ns, dim, stride, hip.SQUARED_L2, device=local_rank)
        dsub = dim // S
        cb_np = np.zeros((S, 16, dsub), np.float32)
        for sidx in range(S):
            c0 = hip.kmeans_init_pp(rbf, 16, seed=42 + sidx, col_offset=sidx * dsub, sub_dim=dsub)
            cb_np[sidx] = hip.kmeans_lloyd(rbf, c0, max_iterations=25, col_offset=sidx * dsub)[0]
        rbf.close()
        cb_t.copy_(torch.from_numpy(cb_np).to(device))
        log("partitioner k-means: %d Lloyd iterations" % it_done)
    if world > 1:
        if args.backend == "nccl":
            for t in (C, assign, cb_t):
                dist.broadcast(t, 0)
        else:
            for t in (C, assign, cb_t):
                h = t.cpu()
                dist.broadcast(h, 0)
                t.copy_(h.to(device))
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=L).cpu().numpy()
    leaf_off = np.zeros(L + 1, np.uint32)
    leaf_off[1:] = np.cumsum(counts)
    sizes = counts.astype(np.uint32)
    owner = sharding.assign_leaves(sizes, world)
    mine = owner == rank
    centers = C.cpu().numpy().astype(np.float32)
    codebook = cb_t.cpu().numpy().astype(np.float32)
    order_np = order.cpu().numpy()
    sel = np.concatenate([order_np[leaf_off[l]:leaf_off[l + 1]] for l in range(L) if mine[l]]
                         or [np.zeros(0, np.int64)])
    local_sizes = np.where(mine, sizes, 0).astype(np.uint32)
    loc_off = np.zeros(L + 1, np.uint32)
    loc_off[1:] = np.cumsum(local_sizes)
    leaf_of_row = np.repeat(np.arange(L, dtype=np.uint32), local_sizes)
    rows_csr = np.zeros((sel.size, stride), np.float32)
    rows_csr[:, :dim] = X[torch.from_numpy(sel).to(device)].cpu().numpy()
    codes = hip.encode(codebook, rows_csr, stride=stride, centers=centers, leaf_of_row=leaf_of_row,
                       device=local_rank)
    index = hip.txh_create(data=rows_csr, n_rows=sel.size, dim=dim, stride=stride, centers=centers,
                           leaf_offsets=loc_off, leaf_ids=sel.astype(np.uint32),
                           leaf_sizes_global=sizes, codebook=codebook, codes=codes, use_residuals=True,
                           partitions_to_search=args.partitions_to_search,
                           pre_reorder_multiplier=float(args.pre_reorder_k) / k, data_is_csr_order=True,
                           device=local_rank)
    queries = Xq.cpu().numpy().astype(np.float32)
    tok, _, _ = hip.txh_partition(index, np.ascontiguousarray(queries[:Q]), args.partitions_to_search)
    scanned_local = local_sizes[tok.astype(np.int64)].sum(1).mean()
    state = dict(index=index, queries=queries, n_local=int(sel.size), data=rows_csr, codebook=codebook,
                 codes=codes,
                 scan_bytes_per_query=int(scanned_local * (S // 2) + args.partitions_to_search * S * 64))
    full = None
    if world == 1:
        full = np.zeros((n, stride), np.float32)
        full[:, :dim] = X.cpu().numpy()
    state["full_data"] = full

    def oracle_index(orc, m, kk):
        return orc.TxhIndex(full, stride, dim, centers, leaf_off, order_np.astype(np.uint32), codebook,
                            codes, use_residuals=True, partitions_to_search=args.partitions_to_search,
                            pre_reorder_multiplier=float(m) / kk)
    state["oracle_index"] = oracle_index
    return state


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from scann_rust_amd import hip, synth, trainer

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    nproc = world                       # processes (= GPUs) of the job
    replica = nproc > 1 and args.multi_gpu == "replica"
    if replica:
        world = 1                       # shards of the index: every rank holds all of it
    srank = 0 if replica else rank      # this rank's shard
    if nproc > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    def all_to_all_start(dst, src):
        """Enqueue the exchange of this step's destination blocks (one block per peer: xGMI is
        point-to-point, every rank sends each peer only the candidates of the queries that peer
        merges).  NCCL/RCCL: asynchronous on the process group's own stream, so the next step's
        local stage overlaps the transfer."""
        if args.backend == "nccl":
            return dist.all_to_all_single(dst.view(-1), src.view(-1), async_op=True)
        h = torch.empty(src.numel(), dtype=src.dtype)            # rehearsal path: through host memory
        dist.all_to_all_single(h, src.view(-1).cpu())
        dst.view(-1).copy_(h.to(dst.device))
        return None

    def wait_for(work):
        if work is not None:
            work.wait()   # the current stream waits for the collective; the host does not block

    def all_gather_start(dst, src):
        if args.backend == "nccl":
            return dist.all_gather_into_tensor(dst.view(-1), src.view(-1), async_op=True)
        else:
            parts = [torch.empty(src.numel(), dtype=src.dtype) for _ in range(world)]
            dist.all_gather(parts, src.view(-1).cpu())
            dst.view(-1).copy_(torch.cat(parts).to(dst.device))
            return None

    L = hip.load()
    n, dim, S, K, k, Q = args.n, args.dim, args.subspaces, args.num_codes, args.k, args.batch
    m = args.pre_reorder_k
    stride = hip.compute_stride(dim)
    stream = torch.cuda.current_stream().cuda_stream
    sptr = ctypes.c_void_p(stream)

    # ---------------- data (synthetic, reference-harness shape: U[0,1), seeds 42/123) ----
    t0 = time.time()
    lo, hi = (n * srank) // world, (n * (srank + 1)) // world
    n_loc = hi - lo
    txh_state = None
    if args.workload == "txh":
        txh_state = build_txh(args, torch, dist, hip, synth, trainer, device, local_rank, srank, world,
                              stride)
        rows = None
        queries_all = txh_state["queries"]
        n_loc = txh_state["n_local"]
    elif args.dist == "uniform":
        rows = synth.uniform_f32(n_loc, dim, 42, row_offset=lo)
        queries_all = synth.uniform_f32(max(Q * 4, args.eval_queries), dim, 123)
    else:
        if world > 1:
            raise SystemExit("clustered data is single-GPU only in this round")
        rows, _ = synth.clustered_f32(n, dim, 7, n_clusters=1000)
        qsrc, _ = synth.clustered_f32(max(Q * 4, args.eval_queries), dim, 8, n_clusters=1000)
        queries_all = qsrc
    if replica and rank:                # every replica searches its own queries
        queries_all = np.roll(queries_all, -rank * Q, axis=0)
    if rows is not None:
        data = np.zeros((n_loc, stride), np.float32)
        data[:, :dim] = rows
    else:
        data = txh_state["data"]
    log("data %.1fs (n_local=%d)" % (time.time() - t0, n_loc))

    workload_name = {"ah": "AsymmetricHasher %s S=%d K=%d + exact re-rank"
                           % ("LUT16" if K <= 16 else "byte codes", S, K),
                     "bf_dot": "BruteForceSearcher.search_batched DotProduct (bf16-MFMA shortlist + exact f32 "
                               "re-score, verified)" if not args.bf_exact else
                               "BruteForceSearcher.search_batched DotProduct (f32 MFMA)",
                     "txh": "Tree-X-Hybrid L=%d P=%d LUT16 S=%d + exact re-rank"
                            % (args.leaves, args.partitions_to_search, S)}[args.workload]

    opts = hip.default_opts()
    index = None
    algo_bytes_per_query = None
    flops_per_query = None
    codebook = codes = None
    t0 = time.time()
    if args.workload == "txh":
        index = txh_state["index"]
        codebook, codes = txh_state["codebook"], txh_state["codes"]
        algo_bytes_per_query = txh_state["scan_bytes_per_query"]
        opts.partitions_to_search = args.partitions_to_search
        opts.pre_reorder_k = m
        opts.exact_reorder = 1
    elif args.workload == "bf_dot":
        if world > 1:
            raise SystemExit("bf_dot is single-GPU only in this round")
        index = hip.bf_create(data, n, dim, stride, hip.DOT_PRODUCT)
        algo_bytes_per_query = n * dim * 4          # SURVEY.md 8d (per DB pass, B = batch)
        flops_per_query = 2.0 * n * dim
    else:
        # codebook: identical on every rank (trained on a deterministic global sample)
        if args.dist == "uniform":
            sel = (synth.splitmix64(0xC0DE, 0, 65536) % np.uint64(n)).astype(np.int64)
            sample = synth.uniform_rows(sel, dim, 42)
        else:
            sample = rows[:: max(1, n // 65536)]
        if args.workload == "ah":
            codebook = trainer.train_codebook(sample, S, K, iters=25, seed=42, sample=1 << 30)
            codes = hip.encode(codebook, data, stride=stride, device=local_rank)
            if world == 1:
                index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=None,
                                       leaf_offsets=None, leaf_ids=None, codebook=codebook,
                                       codes=codes, use_residuals=False, partitions_to_search=1,
                                       pre_reorder_multiplier=float(m) / k, device=local_rank)
            else:
                # flat partition: leaf g = row range of rank g; all leaves searched
                sizes = np.array([(n * (g + 1)) // world - (n * g) // world for g in range(world)],
                                 np.uint32)
                off = np.zeros(world + 1, np.uint32)
                off[srank + 1:] = n_loc
                index = hip.txh_create(
                    data=data, n_rows=n_loc, dim=dim, stride=stride,
                    centers=np.zeros((world, dim), np.float32), leaf_offsets=off,
                    leaf_ids=np.arange(lo, hi, dtype=np.uint32), leaf_sizes_global=sizes,
                    codebook=codebook, codes=codes, use_residuals=False,
                    partitions_to_search=world, pre_reorder_multiplier=float(m) / k,
                    data_is_csr_order=True, device=local_rank)
            # 16 002 128 B at N = 1; per launch a rank scans its n_loc points
            code_bytes = S // 2 if K <= 16 else S
            algo_bytes_per_query = n_loc * code_bytes + S * K * 4 + k * 8
        opts.pre_reorder_k = m
        opts.exact_reorder = 1
    log("index %.1fs" % (time.time() - t0))

    # ---------------- device buffers (inputs resident in HBM before the timed region) ----
    nbatches = 4
    qdev = [torch.from_numpy(np.ascontiguousarray(queries_all[i * Q:(i + 1) * Q])).to(device)
            for i in range(nbatches)]
    out_idx = torch.empty((Q, k), dtype=torch.int32, device=device)
    out_dist = torch.empty((Q, k), dtype=torch.float32, device=device)
    out_cnt = torch.empty((Q,), dtype=torch.int32, device=device)
    # Sharded runs: a rank sends its m_local best candidates.  A random row-range shard holds
    # Binomial(m, 1/world) of the global best m, so m_local = m/world + 6 sigma + 16 suffices;
    # the merge kernel VERIFIES it (status Aborted -> the whole measurement is repeated with
    # m_local = m, which is exact by construction).
    m_local = m
    if world > 1 and args.workload == "ah":
        m_local = min(m, int(m / world + 6.0 * (m / world) ** 0.5 + 16))
    elapsed = kernel_ms = 0.0
    kernel_name = ""
    bf_exact_retry = False

    def device_status_aborted():
        """Brute force: a bf16-shortlist result that could not be proven exact -> repeat the whole
        measurement on the exact kernels.  Any other failure raises."""
        nonlocal bf_exact_retry
        try:
            hip.check(L.scann_hip_index_last_device_status(index.h, sptr))
        except hip.ScannError as e:
            if args.workload == "bf_dot" and e.code == 10 and not bf_exact_retry:
                log("a bf16-shortlist result could not be verified; repeating with the exact kernels")
                bf_exact_retry = True
                return True
            raise
        return False

    while True:
        lopts = hip.default_opts()
        lopts.pre_reorder_k = m_local if world > 1 else opts.pre_reorder_k
        lopts.exact_reorder = opts.exact_reorder
        lopts.partitions_to_search = opts.partitions_to_search
        if world > 1:
            if Q % world:
                raise SystemExit("--batch must be a multiple of the number of ranks")
            from scann_rust_amd import sharding as _sh
            Qr = Q // world
            # local stage output (SoA, [Q][m_local]) -> destination blocks -> ONE all_to_all per
            # step -> merge of this rank's Qr queries -> all_gather of the k result rows (tiny)
            kb, ib = Q * m_local * 8, Q * m_local * 4
            soa = torch.zeros((kb + 2 * ib + Q * 4,), dtype=torch.uint8, device=device)
            ssec = [0, kb, kb + ib, kb + 2 * ib]
            bk, bi, be, bc, block_bytes = _sh.block_layout(Q, m_local, world)
            bsec = [bk, bi, be, bc]
            # double-buffered: step i+1's local stage runs while step i's exchange is in flight
            sends = [torch.zeros((world, block_bytes), dtype=torch.uint8, device=device) for _ in range(2)]
            recvs = [torch.zeros((world, block_bytes), dtype=torch.uint8, device=device) for _ in range(2)]
            rb = Qr * k * 4
            res_locals = [torch.zeros((2 * rb + Qr * 4,), dtype=torch.uint8, device=device) for _ in range(2)]
            res_alls = [torch.zeros((world, 2 * rb + Qr * 4), dtype=torch.uint8, device=device)
                        for _ in range(2)]
            res_work = [None, None]   # result gathers in flight (waited for one step later)
            mstatus = torch.zeros((1,), dtype=torch.int32, device=device)
            pending = []
            last_res = [0]

            def at(t, off):
                return ctypes.c_void_p(t.data_ptr() + off)

            def finish_step():
                work, b = pending.pop(0)
                wait_for(work)
                rv, res_local = recvs[b], res_locals[b]
                wait_for(res_work[b])          # the gather that last read res_locals[b]
                hip.check(L.scann_hip_txh_merge_device(hip.context(local_rank), world, Qr, m_local, m,
                                                       k, block_bytes, at(rv, bsec[0]), at(rv, bsec[1]),
                                                       at(rv, bsec[2]), at(rv, bsec[3]), at(res_local, 0),
                                                       at(res_local, rb), at(res_local, 2 * rb),
                                                       dev_ptr(mstatus), sptr))
                res_work[b] = all_gather_start(res_alls[b], res_local)
                last_res[0] = b
        hip.check(L.scann_hip_index_reserve(index.h, Q, k, ctypes.byref(lopts)))

        def step(i):
            qd = qdev[i % nbatches]
            if world == 1:
                hip.check(L.scann_hip_search_batched_device(index.h, dev_ptr(qd), Q, dim, k,
                                                            ctypes.byref(lopts), dev_ptr(out_idx),
                                                            dev_ptr(out_dist), dev_ptr(out_cnt), sptr))
            else:
                # software pipeline over steps: local stage(i) -> exchange(i) in flight ->
                # [merge(i-1) + result gather]; the last merge is drained by flush_steps()
                b = i & 1
                hip.check(L.scann_hip_txh_search_local_device(index.h, dev_ptr(qd), Q, dim, k,
                                                              ctypes.byref(lopts), at(soa, ssec[0]),
                                                              at(soa, ssec[1]), at(soa, ssec[2]),
                                                              at(soa, ssec[3]), sptr))
                hip.check(L.scann_hip_txh_pack_blocks_device(hip.context(local_rank), world, Q, m_local,
                                                             at(soa, ssec[0]), at(soa, ssec[1]),
                                                             at(soa, ssec[2]), at(soa, ssec[3]),
                                                             dev_ptr(sends[b]), block_bytes, sptr))
                work = all_to_all_start(recvs[b], sends[b])
                if pending:
                    finish_step()
                pending.append((work, b))

        def flush_steps():
            while world > 1 and pending:
                finish_step()
            if world > 1:
                for wk in res_work:
                    wait_for(wk)

        # ---------------- warmup, then EXACTLY K timed steps -------------------------------
        lopts.bf_exact = 1 if (args.bf_exact or bf_exact_retry) else 0
        for i in range(args.warmup):
            step(i)
        flush_steps()
        torch.cuda.synchronize()
        if device_status_aborted():
            continue
        index.enable_timing(True)
        if nproc > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        flush_steps()
        torch.cuda.synchronize()
        if nproc > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms, kernel_name = index.last_kernel_ms()
        index.enable_timing(False)
        if device_status_aborted():
            continue
        if replica:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0].item())
        if world > 1:
            t = torch.tensor([elapsed, float(mstatus.item())], dtype=torch.float64,
                             device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0].item())
            if t[1].item() != 0 and m_local < m:
                log("a shard's m_local=%d list was too short; repeating with m_local=m" % m_local)
                m_local = m
                continue
            if t[1].item() != 0:
                raise SystemExit("merge reported status %d" % int(t[1].item()))
        break
    qps = Q * args.steps * (nproc if replica else 1) / elapsed
    if world > 1:   # result rows of the last step, gathered from the ranks that merged them
        ra = res_alls[last_res[0]].cpu().numpy()
        rb_ = (Q // world) * k * 4
        out_idx = torch.from_numpy(np.concatenate([ra[g, :rb_].view(np.int32).reshape(-1, k)
                                                   for g in range(world)]))
        out_dist = torch.from_numpy(np.concatenate([ra[g, rb_:2 * rb_].view(np.float32).reshape(-1, k)
                                                    for g in range(world)]))

    # ---------------- recall10@10 (bin/ann_benchmark.rs:427-471 semantics) ------------------
    recall = None
    checked = None
    if not args.no_recall and world == 1 and rank == 0 and args.workload != "bf_dot":
        ne = min(args.eval_queries, queries_all.shape[0])
        qe = np.ascontiguousarray(queries_all[:ne])
        gi, gd, gc = index.search_batched(qe, k, opts)
        bf = hip.bf_create(txh_state["full_data"] if txh_state else data, n, dim, stride,
                           hip.SQUARED_L2, device=local_rank)
        ti, td, tc = bf.search_batched(qe, k)
        hits = sum(len(set(gi[i].tolist()) & set(ti[i].tolist())) for i in range(ne))
        recall = hits / float(ne * k)
        bf.close()
    if rank == 0 and world > 1 and args.workload == "ah":
        # the merged rows of the last timed step, checked against the oracle on the FULL
        # database (rank 0 regenerates it; checker only)
        from oracle import pyoracle as orc
        full = np.zeros((n, stride), np.float32)
        full[:, :dim] = synth.uniform_f32(n, dim, 42)
        fcodes = trainer.encode(codebook, full[:, :dim])
        last = np.ascontiguousarray(queries_all[((args.steps - 1) % nbatches) * Q:][:4])
        gi = out_idx[:4].cpu().numpy().view(np.uint32)
        gd = out_dist[:4].cpu().numpy()
        ok = True
        for i in range(4):
            oi, od = orc.ah_search_with_reordering(codebook, fcodes, full, stride, last[i], k, m)
            ok = ok and np.array_equal(gd[i].view(np.uint32), od.view(np.uint32)) \
                and sorted(gi[i].tolist()) == sorted(oi.tolist())
        checked = bool(ok)
    if rank == 0 and world == 1:
        # result rows of the timed path checked against the oracle (checker only)
        from oracle import pyoracle as orc
        nchk = 4
        qe = np.ascontiguousarray(queries_all[:nchk])
        gi, gd, gc = index.search_batched(qe, k, opts)
        ok = True
        for i in range(nchk):
            if args.workload == "bf_dot":
                oi, od = orc.bf_search(data, n, dim, stride, orc.DOT_PRODUCT, qe[i], k)
            elif args.workload == "txh":
                oi, od = orc.txh_search(txh_state["oracle_index"](orc, m, k), qe[i], k)
            else:
                oi, od = orc.ah_search_with_reordering(codebook, codes, data, stride, qe[i], k, m)
            ok = ok and np.array_equal(gd[i].view(np.uint32), od.view(np.uint32)) \
                and sorted(gi[i].tolist()) == sorted(oi.tolist())
        checked = bool(ok)

    # ---------------- CPU baseline: the oracle on this box's host cores (rank 0, N = 1) -------
    cpu = None
    if rank == 0 and nproc == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as orc
        threads = orc.max_threads()
        nq0 = min(threads, queries_all.shape[0])
        qs = np.ascontiguousarray(queries_all[:nq0])

        def run_cpu(qb):
            t1 = time.perf_counter()
            if args.workload == "bf_dot":
                orc.bf_search_batched(data, n, dim, stride, orc.DOT_PRODUCT, qb, k, threads)
            elif args.workload == "txh":
                orc.txh_search_batched(txh_state["oracle_index"](orc, m, k), qb, k, threads)
            else:
                orc.ah_search_batched(codebook, codes, data, stride, qb, k, m, True, threads)
            return time.perf_counter() - t1

        t_probe = run_cpu(qs)
        reps = int(max(1, min(64, args.cpu_baseline_seconds / max(t_probe, 1e-3))))
        nq1 = min(queries_all.shape[0], nq0 * reps)
        t_run = run_cpu(np.ascontiguousarray(queries_all[:nq1]))
        cpu = {"value": nq1 / t_run, "unit": "queries/s", "cores": threads, "kind": "port",
               "sample": "%d queries of the same workload (same index, k, pre_reorder_k), "
                         "one OpenMP task per query, %.1f s" % (nq1, t_run)}

    if rank == 0:
        if args.workload == "bf_dot" and kernel_name == "bf_stream_kernel":
            # a few queries: one coalesced pass over the database per 8 queries (N*d*4 B)
            passes = (Q + 7) // 8
            achieved = algo_bytes_per_query * passes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": kernel_name,
                    "kernel_ms": kernel_ms,
                    "algorithmic": "N*d*4 = %d B per database pass x %d passes (8 queries each) per launch"
                                   % (algo_bytes_per_query, passes)}
        elif args.workload == "bf_dot":
            achieved = flops_per_query * Q / (kernel_ms * 1e-3) / 1e12 if kernel_ms else 0.0
            peak = BF16_MFMA_PEAK_TFLOPS if kernel_name == "bf_bf16_kernel" else F32_MFMA_PEAK_TFLOPS
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                    "kernel": kernel_name, "kernel_ms": kernel_ms,
                    "algorithmic": "2*N*d flop per query x %d queries per launch (%s)"
                                   % (Q, "bf16 MFMA shortlist pass; the shortlisted rows are re-scored with "
                                         "the reference's f32 arithmetic and the result is verified"
                                      if kernel_name == "bf_bf16_kernel" else "f32 MFMA, exact")}
        else:
            achieved = algo_bytes_per_query * Q / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": kernel_name,
                    "kernel_ms": kernel_ms,
                    "algorithmic": ("%d B per query (scanned codes + LUTs of the selected leaves) x %d "
                                    "queries per launch (rank 0)" if args.workload == "txh" else
                                    "%d B per query (N_local*code_bytes + S*K*4 LUT + k*8 out) x %d "
                                    "queries per launch (rank 0)") % (algo_bytes_per_query, Q)}
        tr = os.path.join(ROOT, "profiles", "traffic.json")
        # PMC-measured HBM bytes per launch (profiles/traffic.json, collected with rocprofv3 --pmc
        # on this workload's default configuration only: single GPU, 1M x 128, batch 1024)
        default_cfg = world == 1 and n == 1_000_000 and dim == 128 and Q == 1024 and \
            (args.workload != "ah" or m == 5000) and args.workload != "txh"
        if os.path.exists(tr) and default_cfg:
            try:
                roof["traffic"] = json.load(open(tr)).get(args.workload)
            except Exception:
                pass
        line = {
            "metric": "QPS @ recall10@10 + achieved HBM GB/s, 1M x 128 f32",
            "value": qps, "unit": "queries/s", "n_gpus": nproc, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if (replica or nproc == 1) else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name, "n": n, "dim": dim, "k": k, "batch": Q,
                       "global_batch": Q * (nproc if replica else 1),
                       "pre_reorder_k": m if args.workload != "bf_dot" else None,
                       "pre_reorder_k_per_rank": m_local if world > 1 else None,
                       "distribution": args.dist if args.workload != "txh" else "clustered (1000 Gaussians)",
                       "leaves": args.leaves if args.workload == "txh" else None,
                       "partitions_to_search": args.partitions_to_search if args.workload == "txh" else None,
                       "recall10@10": recall,
                       "oracle_check": checked,
                       "parallelism": "1 process/GPU, leaf(row-range)-sharded x%d + RCCL all_to_all of candidates"
                                      % world if world > 1 else
                                      ("1 process/GPU, %d query-parallel replicas of the index (no data-path "
                                       "collective); --multi-gpu shard = leaf-sharded index + RCCL all_to_all"
                                       % nproc) if replica else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if nproc > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
