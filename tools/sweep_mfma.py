#!/usr/bin/env python3
"""Kernel-tuning sweep of the C3 workload (AH LUT16 1M x 128, batch 1024, m = 5000): one index build,
then step time and dominant-kernel time for each value of the launch-time knobs given as
NAME=v1,v2,... arguments (environment variables the library reads at launch)."""
import ctypes
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from scann_rust_amd import hip, synth, trainer
    knobs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
    n, dim, S, K, k, Q, m = 1_000_000, 128, 32, 16, 10, 1024, 5000
    if "--n" in sys.argv:
        n = int(sys.argv[sys.argv.index("--n") + 1])
    if "--batch" in sys.argv:
        Q = int(sys.argv[sys.argv.index("--batch") + 1])
    rows = synth.uniform_f32(n, dim, 42)
    sample = synth.uniform_rows((synth.splitmix64(0xC0DE, 0, 65536) % np.uint64(n)).astype(np.int64), dim, 42)
    cb = trainer.train_codebook(sample, S, K, iters=10, seed=42, sample=1 << 30)
    codes = hip.encode(cb, rows, stride=dim)
    index = hip.txh_create(data=rows, n_rows=n, dim=dim, stride=dim, centers=None, leaf_offsets=None, leaf_ids=None,
                           codebook=cb, codes=codes, use_residuals=False, partitions_to_search=1,
                           pre_reorder_multiplier=float(m) / k)
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(synth.uniform_f32(Q, dim, 123)).to(dev)
    oi = torch.empty((Q, k), dtype=torch.int32, device=dev)
    od = torch.empty((Q, k), dtype=torch.float32, device=dev)
    oc = torch.empty((Q,), dtype=torch.int32, device=dev)
    L = hip.load()
    o = hip.default_opts()
    o.pre_reorder_k, o.exact_reorder = m, 1
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    hip.check(L.scann_hip_index_reserve(index.h, Q, k, ctypes.byref(o)))

    def run(cnt):
        for _ in range(cnt):
            hip.check(L.scann_hip_search_batched_device(index.h, ctypes.c_void_p(q.data_ptr()), Q, dim, k,
                                                        ctypes.byref(o), ctypes.c_void_p(oi.data_ptr()),
                                                        ctypes.c_void_p(od.data_ptr()), ctypes.c_void_p(oc.data_ptr()), sp))
    names = [kv[0] for kv in knobs]
    for combo in itertools.product(*[kv[1].split(",") for kv in knobs]) if knobs else [()]:
        for nm, v in zip(names, combo):
            os.environ[nm] = v
        hip.check(L.scann_hip_index_reserve(index.h, Q, k, ctypes.byref(o)))
        run(5)
        torch.cuda.synchronize()
        index.enable_timing(True)
        t0 = time.perf_counter()
        run(50)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        kms, kname = index.last_kernel_ms()
        index.enable_timing(False)
        try:
            hip.check(L.scann_hip_index_last_device_status(index.h, sp))
            status = ""
        except hip.ScannError as e:
            status = "  [device status: %s]" % e
        print("%s lib=%s  step %.3f ms  %s %.3f ms  (%.0f QPS)" % (
            " ".join("%s=%s" % z for z in zip(names, combo)), os.path.basename(hip.LIB_PATH), dt * 1e3, kname, kms,
            Q / dt) + status, flush=True)


if __name__ == "__main__":
    main()
