import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import helpers as H
from oracle import pyoracle as orc
from scann_rust_amd import hip, synth
rows, data, stride, ix, oix, kw = H.make_txh_case(80000, 96, 20, 24, seed=32, P=6, kmeans_iters=3, pq_iters=3, clustered=True)
o = hip.default_opts()
o.partitions_to_search, o.pre_reorder_k = 6, 250
index = hip.txh_create(**kw)
q = synth.clustered_f32(96, 96, 33, n_clusters=20)[0]
os.environ["SCANN_HIP_MFMA"] = "0"
os.environ["SCANN_HIP_RESIDENT"] = "0"
idx, dist, cnt, (tok, tokd, ci, cd, cc) = index.search_batched(q, 10, o, stages=True)
oix2 = orc.TxhIndex(data, stride, 96, ix["centers"], ix["leaf_off"], ix["leaf_ids"], ix["codebook"], ix["codes"],
                    partitions_to_search=6, pre_reorder_multiplier=25.0)
for i in (0, 5):
    oi, od, otok, otokd, oci, ocd = orc.txh_search(oix2, q[i], 10, stages=True)
    g = cd[i, :cc[i]]
    print(i, cc[i], oci.size, np.array_equal(tok[i], otok))
    d = np.nonzero(g.view(np.uint32) != ocd.view(np.uint32))[0]
    print(" ndiff", d.size, d[:10], g[d[:5]], ocd[d[:5]])
    print(" gpu tail", g[-5:], "oracle tail", ocd[-5:], "sorted?", np.all(np.diff(g) >= 0), np.all(np.diff(ocd) >= 0))
    print(" uniq", np.unique(g).size, np.unique(ocd).size)
