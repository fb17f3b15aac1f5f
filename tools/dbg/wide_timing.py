"""Phase timestamps inside the wide pipeline's kernels (library variant built with -DSCANN_WIDE_TIMING):
   python3 -c "from scann_rust_amd import build; build.build_variant('wt', ['-DSCANN_WIDE_TIMING'])"
   SCANN_HIP_LIB=$PWD/scann_rust_amd/libscann_hip_wt.so python3 tools/dbg/wide_timing.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scann_rust_amd import hip, synth, trainer
from oracle import pyoracle as orc

n, dim, S = 1_000_000, 128, 32
rows = synth.uniform_f32(n, dim, 42)
data, stride = orc.to_strided(rows)
ix = trainer.build_ah_index(rows[:100000], S, K=16, seed=42, pq_iters=2)
codes = hip.encode(ix["codebook"], data, stride=stride)
index = hip.txh_create(data=data, n_rows=n, dim=dim, stride=stride, centers=None, leaf_offsets=None, leaf_ids=None,
                       codebook=ix["codebook"], codes=codes, codes_packed4=False, use_residuals=False,
                       partitions_to_search=1, pre_reorder_multiplier=1.0)
o = hip.default_opts()
o.pre_reorder_k = 5000
q = synth.uniform_f32(8, dim, 123)
for i in range(6):
    index.search_batched(q[i:i + 1], 10, o)
