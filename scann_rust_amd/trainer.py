"""Trainer-lite: builds the flat index arrays the hot path consumes.

The reference trains with rand::StdRng-seeded k-means (trees/kmeans.rs:166-414,
hashes/codebook.rs:146-202, tree_x_hybrid/mod.rs:131-209); that RNG stream is
not reproducible here and is pinned by no reference test, so the
trained index is an INPUT to the search path, not part of the parity contract.
What IS part of the contract is the encode rule (hashes/codebook.rs:82-95:
sequential f32 squared distance, strict '<', lowest centre index wins), because
the codes feed the scan; `encode` below reproduces it bit for bit in numpy.

Host-side numpy only (small/medium inputs); no oracle imports.
"""
import numpy as np

from . import synth


def _sq_dists(X, C, chunk=8192):
    """argmin_c ||x - c||^2 via the expansion (training only; not parity maths)."""
    cn = (C.astype(np.float32) ** 2).sum(1)
    out = np.empty(X.shape[0], np.int64)
    for r0 in range(0, X.shape[0], chunk):
        x = X[r0:r0 + chunk]
        d = cn[None, :] - 2.0 * (x @ C.T)
        out[r0:r0 + chunk] = d.argmin(1)
    return out


def kmeans(X, k, iters=10, seed=42):
    """Plain Lloyd k-means.  Returns (centers f32 [k,d], assignment int64 [n])."""
    X = np.ascontiguousarray(X, np.float32)
    n, d = X.shape
    k = min(k, n)
    pick = (synth.splitmix64(seed, 0, 4 * k) % np.uint64(n)).astype(np.int64)
    _, first = np.unique(pick, return_index=True)
    pick = pick[np.sort(first)][:k]
    if pick.size < k:  # top up deterministically
        rest = np.setdiff1d(np.arange(n), pick)[: k - pick.size]
        pick = np.concatenate([pick, rest])
    C = X[pick].copy()
    a = _sq_dists(X, C)
    for _ in range(iters):
        sums = np.zeros((k, d), np.float64)
        np.add.at(sums, a, X)
        cnt = np.bincount(a, minlength=k)
        nz = cnt > 0
        C[nz] = (sums[nz] / cnt[nz, None]).astype(np.float32)
        a_new = _sq_dists(X, C)
        if np.array_equal(a_new, a):
            break
        a = a_new
    return C, a


def train_codebook(R, S, K, iters=10, seed=42, sample=65536):
    """Per-subspace k-means (hashes/codebook.rs:177-199, seed + s per subspace).
    Returns codebook f32 [S, K, dsub]."""
    R = np.ascontiguousarray(R, np.float32)
    n, d = R.shape
    if d % S != 0:
        raise ValueError("InvalidArgument: dimensionality %d must be divisible by "
                         "num_subspaces %d" % (d, S))  # codebook.rs:154-159
    dsub = d // S
    if n > sample:
        sel = (synth.splitmix64(seed ^ 0xC0DE, 0, sample) % np.uint64(n)).astype(np.int64)
        R = R[sel]
    cb = np.zeros((S, K, dsub), np.float32)
    for s in range(S):
        C, _ = kmeans(R[:, s * dsub:(s + 1) * dsub], K, iters=iters, seed=seed + s)
        cb[s, :C.shape[0]] = C
        if C.shape[0] < K:  # fewer points than codes: repeat the last centre
            cb[s, C.shape[0]:] = C[-1]
    return cb


def encode(codebook, R, chunk=65536):
    """Codebook::encode (hashes/codebook.rs:82-95, 205-215), vectorised but
    bit-identical: per subspace d = x - c; p = d*d; sum left-to-right in f32;
    argmin takes the first (lowest) index among equal minima."""
    codebook = np.ascontiguousarray(codebook, np.float32)
    R = np.ascontiguousarray(R, np.float32)
    S, K, dsub = codebook.shape
    n = R.shape[0]
    codes = np.empty((n, S), np.uint8)
    for r0 in range(0, n, chunk):
        x = R[r0:r0 + chunk].reshape(-1, S, 1, dsub)
        diff = x - codebook[None]                      # [c, S, K, dsub] f32
        p = diff * diff
        acc = p[..., 0].copy()
        for j in range(1, dsub):
            acc = acc + p[..., j]
        codes[r0:r0 + chunk] = acc.argmin(2).astype(np.uint8)
    return codes


def pack4(codes):
    """PackedCodes4Bit::from_codes (hashes/lut16.rs:43-61): byte j =
    code[2j] | code[2j+1] << 4; odd S pads the last high nibble with 0."""
    codes = np.ascontiguousarray(codes, np.uint8)
    n, S = codes.shape
    if S % 2:
        codes = np.concatenate([codes, np.zeros((n, 1), np.uint8)], 1)
    return ((codes[:, 0::2] & 0x0F) | ((codes[:, 1::2] & 0x0F) << 4)).astype(np.uint8)


def build_txh_index(data, L, S, K=16, use_residuals=True, kmeans_iters=10,
                    pq_iters=10, seed=42, centers=None, assign=None):
    """What TreeXHybridSearcher::build produces (tree_x_hybrid/mod.rs:131-209),
    as flat arrays: centers [L,d], CSR leaf_off [L+1] / leaf_ids [n] (ascending
    datapoint index inside each leaf, as partition_to_indices is filled in index
    order: tree_partitioner.rs:84-89), codebook [S,K,dsub], codes [n,S] in CSR
    row order (residual-encoded against the owning leaf's centre)."""
    data = np.ascontiguousarray(data, np.float32)
    n, d = data.shape
    if n == 0:
        raise ValueError("InvalidArgument: Cannot build from empty dataset")
    if centers is None:
        centers, assign = kmeans(data, L, iters=kmeans_iters, seed=seed)
    centers = np.ascontiguousarray(centers, np.float32)
    L = centers.shape[0]
    assign = np.asarray(assign, np.int64)
    order = np.argsort(assign, kind="stable").astype(np.uint32)
    counts = np.bincount(assign, minlength=L)
    leaf_off = np.zeros(L + 1, np.uint32)
    leaf_off[1:] = np.cumsum(counts)
    rows = data[order]
    if use_residuals:
        rows = rows - centers[assign[order]]
    codebook = train_codebook(rows, S, K, iters=pq_iters, seed=seed)
    codes = encode(codebook, rows)
    return dict(centers=centers, leaf_off=leaf_off, leaf_ids=order, codebook=codebook,
                codes=codes, use_residuals=bool(use_residuals))


def build_ah_index(data, S, K=16, pq_iters=10, seed=42):
    """AsymmetricHasher::build (hashes/hasher.rs:109-134): global codebook on the
    raw vectors, codes in datapoint order."""
    data = np.ascontiguousarray(data, np.float32)
    if data.shape[0] == 0:
        raise ValueError("InvalidArgument: Cannot build from empty dataset")
    codebook = train_codebook(data, S, K, iters=pq_iters, seed=seed)
    return dict(codebook=codebook, codes=encode(codebook, data))
