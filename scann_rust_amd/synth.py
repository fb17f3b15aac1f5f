"""Synthetic inputs with the reference harness' shape.

The reference draws every element i.i.d. U[0,1) f32 from rand::StdRng
(bin/ann_benchmark.rs:402-425, tests/stress_tests.rs:10-24, DB seed 42 /
query seed 123).  StdRng (ChaCha12) cannot be reproduced here (no Rust
toolchain, Cargo.lock not pinned), so we use a documented counter-based
generator instead: splitmix64 over a counter, top 24 bits -> (u >> 40) * 2^-24,
the same 24-bit-mantissa construction `rand` uses for gen::<f32>().
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, start, count):
    """count 64-bit outputs of splitmix64(seed) starting at stream position start."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_f32(n, dim, seed, chunk_rows=1 << 16, row_offset=0):
    """[n, dim] f32, i.i.d. U[0,1) with 24-bit mantissas: rows row_offset .. row_offset+n
    of the (unbounded) dataset `seed` -- the generator is counter based, so any row range
    can be produced independently (each rank of a sharded run generates only its rows)."""
    out = np.empty((n, dim), np.float32)
    for r0 in range(0, n, chunk_rows):
        r1 = min(n, r0 + chunk_rows)
        u = splitmix64(seed, (row_offset + r0) * dim, (r1 - r0) * dim)
        out[r0:r1] = ((u >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
                      ).reshape(r1 - r0, dim)
    return out


def uniform_rows(rows, dim, seed):
    """Selected rows (by index) of the dataset uniform_f32(., dim, seed)."""
    rows = np.asarray(rows, np.int64)
    out = np.empty((rows.size, dim), np.float32)
    for i, r in enumerate(rows):
        u = splitmix64(seed, int(r) * dim, dim)
        out[i] = (u >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return out


def normal_f32(n, dim, seed, chunk_rows=1 << 16):
    """[n, dim] f32 standard normal (Box-Muller over the same stream)."""
    out = np.empty((n, dim), np.float32)
    for r0 in range(0, n, chunk_rows):
        r1 = min(n, r0 + chunk_rows)
        cnt = (r1 - r0) * dim
        u = splitmix64(seed, 2 * r0 * dim, 2 * cnt)
        u1 = ((u[:cnt] >> np.uint64(11)).astype(np.float64) + 1.0) * (2.0 ** -53)
        u2 = (u[cnt:] >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        out[r0:r1] = z.astype(np.float32).reshape(r1 - r0, dim)
    return out


def clustered_f32(n, dim, seed, n_clusters=1000, sigma_frac=0.1):
    """Mixture of n_clusters isotropic Gaussians (second
    distribution): centres U[0,1)^dim, |noise| ~ sigma_frac * typical
    inter-centre distance sqrt(dim/6) (per-dimension sigma = sigma_frac / sqrt(6)).  Returns (points, assignment)."""
    centres = uniform_f32(n_clusters, dim, seed ^ 0x5EED)
    # total sigma = sigma_frac * sqrt(dim/6); per dimension divide by sqrt(dim)
    sigma = np.float32(sigma_frac * np.sqrt(1.0 / 6.0))
    a = (splitmix64(seed ^ 0xA551, 0, n) % np.uint64(n_clusters)).astype(np.int64)
    pts = centres[a] + sigma * normal_f32(n, dim, seed)
    return pts.astype(np.float32), a
