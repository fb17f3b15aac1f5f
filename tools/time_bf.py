#!/usr/bin/env python3
"""Times BruteForceSearcher.search_batched on the GPU for each measure (development tool)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scann_rust_amd import hip, synth

n, dim, Q, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128, 1024, 10
rows = synth.uniform_f32(n, dim, 42)
qs = synth.uniform_f32(Q, dim, 123)
L = hip.load()
dev = torch.device("cuda", 0)
qd = torch.from_numpy(qs).to(dev)
oi = torch.empty((Q, k), dtype=torch.int32, device=dev)
od = torch.empty((Q, k), dtype=torch.float32, device=dev)
oc = torch.empty((Q,), dtype=torch.int32, device=dev)
sptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, meas in (("dot", hip.DOT_PRODUCT), ("sql2", hip.SQUARED_L2), ("l2", hip.L2)):
    ix = hip.bf_create(rows, n, dim, dim, meas)
    hip.check(L.scann_hip_index_reserve(ix.h, Q, k, None))
    def run():
        hip.check(L.scann_hip_search_batched_device(ix.h, ctypes.c_void_p(qd.data_ptr()), Q, dim, k, None,
                  ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(od.data_ptr()), ctypes.c_void_p(oc.data_ptr()), sptr))
    run(); torch.cuda.synchronize()
    ix.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 3
    kms, kn = ix.last_kernel_ms()
    print("%-5s n=%d  %.2f ms/step  %.0f QPS  main-pass kernel %.2f ms" % (name, n, el * 1e3, Q / el, kms), flush=True)
    ix.close()
