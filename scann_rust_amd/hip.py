"""Raw ctypes binding of libscann_hip.so (include/scann_hip.h).

Plumbing only: numpy arrays in, numpy arrays out.  There is NO fallback: if the
library is missing or no gfx950 device is present every call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCANN_HIP_LIB: alternative build of the same library (kernel-tuning sweeps)
LIB_PATH = os.environ.get("SCANN_HIP_LIB") or os.path.join(_HERE, "libscann_hip.so")

OK, INVALID_ARGUMENT, RESOURCE_EXHAUSTED, FAILED_PRECONDITION = 0, 3, 8, 9
OUT_OF_RANGE, UNIMPLEMENTED, INTERNAL, UNAVAILABLE = 11, 12, 13, 14
SQUARED_L2, L2, DOT_PRODUCT, L1, COSINE = 0, 1, 2, 3, 4

_CODE_NAMES = {
    0: "Ok", 1: "Cancelled", 2: "Unknown", 3: "InvalidArgument", 4: "DeadlineExceeded",
    5: "NotFound", 6: "AlreadyExists", 7: "PermissionDenied", 8: "ResourceExhausted",
    9: "FailedPrecondition", 10: "Aborted", 11: "OutOfRange", 12: "Unimplemented",
    13: "Internal", 14: "Unavailable", 15: "DataLoss", 16: "Unauthenticated",
}

EXPORTS = [
    "scann_hip_init", "scann_hip_shutdown", "scann_hip_last_error", "scann_hip_version",
    "scann_hip_compute_stride", "scann_hip_bf_create", "scann_hip_txh_create",
    "scann_hip_search_opts_default", "scann_hip_search_batched", "scann_hip_search_batched_params", "scann_hip_index_reserve",
    "scann_hip_search_batched_device", "scann_hip_index_last_device_status",
    "scann_hip_txh_search_local_device", "scann_hip_txh_merge_device",
    "scann_hip_assign_leaves", "scann_hip_txh_partition", "scann_hip_lut_from_query",
    "scann_hip_adc_distances", "scann_hip_lut16_distances_batch", "scann_hip_encode",
    "scann_hip_fp8_quantize", "scann_hip_fp8_dequantize", "scann_hip_fp8_distances",
    "scann_hip_bf_distances", "scann_hip_bf_search_radius", "scann_hip_bf_assign_nearest",
    "scann_hip_kmeans_init_pp", "scann_hip_kmeans_lloyd", "scann_hip_txh_pack_blocks_device", "scann_hip_index_size", "scann_hip_index_dimensionality",
    "scann_hip_index_destroy", "scann_hip_index_enable_timing",
    "scann_hip_index_last_kernel_ms",
    "scann_hip_txh_write_file", "scann_hip_bf_write_file", "scann_hip_index_file_info",
    "scann_hip_index_load_file",
    "scann_hip_abi_layout", "scann_hip_lut16_quantize",
    "scann_hip_comm_unique_id", "scann_hip_comm_create", "scann_hip_comm_destroy",
    "scann_hip_txh_search_sharded_device", "scann_hip_comm_last_status", "scann_hip_comm_layout",
]


class ScannError(RuntimeError):
    """error.rs:73-147: ScannError { code, message }."""

    def __init__(self, code, message):
        super().__init__("%s: %s" % (_CODE_NAMES.get(code, str(code)), message))
        self.code = code
        self.message = message


f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


class TxhDesc(C.Structure):
    _fields_ = [
        ("data", f32p), ("n_rows", C.c_uint64), ("dim", C.c_uint32), ("stride", C.c_uint32),
        ("data_is_csr_order", C.c_int32),
        ("centers", f32p), ("num_partitions", C.c_uint32),
        ("leaf_offsets", u32p), ("leaf_ids", u32p), ("leaf_sizes_global", u32p),
        ("n_local", C.c_uint64),
        ("codebook", f32p), ("num_subspaces", C.c_uint32), ("num_codes", C.c_uint32),
        ("dims_per_subspace", C.c_uint32),
        ("codes", u8p), ("codes_packed4", C.c_int32), ("use_residuals", C.c_int32),
        ("partitions_to_search", C.c_uint32), ("pre_reorder_multiplier", C.c_float),
        ("distance_measure", C.c_int32),
    ]


class FileInfo(C.Structure):
    _fields_ = [
        ("version", C.c_uint32), ("kind", C.c_uint32),
        ("n_rows", C.c_uint64), ("n_local", C.c_uint64), ("file_bytes", C.c_uint64),
        ("dim", C.c_uint32), ("stride", C.c_uint32), ("num_partitions", C.c_uint32),
        ("num_subspaces", C.c_uint32), ("num_codes", C.c_uint32), ("dims_per_subspace", C.c_uint32),
        ("distance_measure", C.c_int32), ("data_is_csr_order", C.c_int32), ("codes_packed4", C.c_int32),
        ("use_residuals", C.c_int32), ("partitions_to_search", C.c_uint32),
        ("pre_reorder_multiplier", C.c_float), ("has_data", C.c_int32),
    ]


class SearchOpts(C.Structure):
    _fields_ = [
        ("partitions_to_search", C.c_uint32), ("pre_reorder_k", C.c_uint32),
        ("exact_reorder", C.c_int32),
        ("tokens", u32p), ("token_dists", f32p),
        ("cand_idx", u32p), ("cand_dist", f32p), ("cand_count", u32p),
        ("allow_bitmap", u64p), ("allow_bitmap_bits", C.c_uint64),
        ("bf_exact", C.c_int32),
    ]


_lib = None


def load():
    """Load the shared library (no GPU needed just to load and look up symbols)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScannError(UNAVAILABLE, "libscann_hip.so is not built (run "
                         "`python -m scann_rust_amd.build`); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    L.scann_hip_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.scann_hip_shutdown.argtypes = [vp]
    L.scann_hip_shutdown.restype = None
    L.scann_hip_last_error.restype = C.c_char_p
    L.scann_hip_version.restype = C.c_char_p
    L.scann_hip_compute_stride.restype = C.c_uint32
    L.scann_hip_compute_stride.argtypes = [C.c_uint32]
    L.scann_hip_bf_create.argtypes = [vp, f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int,
                                      C.POINTER(vp)]
    L.scann_hip_txh_create.argtypes = [vp, C.POINTER(TxhDesc), C.POINTER(vp)]
    L.scann_hip_txh_write_file.argtypes = [C.c_char_p, C.POINTER(TxhDesc)]
    L.scann_hip_bf_write_file.argtypes = [C.c_char_p, f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    L.scann_hip_index_file_info.argtypes = [C.c_char_p, C.POINTER(FileInfo)]
    L.scann_hip_index_load_file.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.scann_hip_search_opts_default.argtypes = [C.POINTER(SearchOpts)]
    L.scann_hip_search_opts_default.restype = None
    L.scann_hip_search_batched.argtypes = [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_uint32, C.POINTER(SearchOpts), u32p, f32p, u32p]
    L.scann_hip_search_batched_params.argtypes = [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, u32p,
                                                  C.POINTER(SearchOpts), C.c_uint32, u32p, f32p, u32p]
    L.scann_hip_index_reserve.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(SearchOpts)]
    L.scann_hip_search_batched_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  C.POINTER(SearchOpts), vp, vp, vp, vp]
    L.scann_hip_index_last_device_status.argtypes = [vp, vp]
    L.scann_hip_txh_search_local_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                                    C.POINTER(SearchOpts), vp, vp, vp, vp, vp]
    L.scann_hip_txh_merge_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.scann_hip_assign_leaves.argtypes = [u32p, C.c_uint32, C.c_uint32, u32p]
    L.scann_hip_txh_partition.argtypes = [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, u32p, f32p, u32p]
    L.scann_hip_lut_from_query.argtypes = [vp, f32p, C.c_uint32, C.c_uint32, u32p, f32p]
    L.scann_hip_adc_distances.argtypes = [vp, f32p, C.c_uint32, f32p]
    L.scann_hip_lut16_distances_batch.argtypes = [vp, u8p, u8p, C.c_uint32, C.c_uint64,
                                                  C.c_float, C.c_float, f32p]
    L.scann_hip_encode.argtypes = [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p,
                                   C.c_uint64, C.c_uint32, f32p, u32p, u8p]
    L.scann_hip_bf_distances.argtypes = [vp, f32p, C.c_uint32, C.c_uint32, f32p]
    L.scann_hip_bf_assign_nearest.argtypes = [vp, f32p, C.c_uint32, u32p, f32p]
    L.scann_hip_txh_pack_blocks_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp,
                                                   C.c_uint64, vp]
    L.scann_hip_kmeans_init_pp.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
                                           f32p]
    L.scann_hip_kmeans_lloyd.argtypes = [vp, C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.c_uint32,
                                         C.c_double, C.c_uint32, u32p, u32p, C.POINTER(C.c_double), u32p,
                                         C.POINTER(C.c_int)]
    L.scann_hip_abi_layout.restype = C.c_uint32
    L.scann_hip_abi_layout.argtypes = [u32p, C.c_uint32]
    L.scann_hip_lut16_quantize.argtypes = [vp, f32p, C.c_uint32, u8p, f32p, f32p]
    L.scann_hip_fp8_quantize.argtypes = [vp, f32p, C.c_uint64, C.c_float, C.c_int, u8p]
    L.scann_hip_fp8_dequantize.argtypes = [vp, u8p, C.c_uint64, C.c_float, C.c_int, f32p]
    L.scann_hip_fp8_distances.argtypes = [vp, f32p, C.c_uint32, u8p, C.c_uint64, C.c_uint64, C.c_int, f32p]
    L.scann_hip_comm_unique_id.argtypes = [vp]
    L.scann_hip_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.scann_hip_comm_destroy.argtypes = [vp]
    L.scann_hip_comm_destroy.restype = None
    L.scann_hip_txh_search_sharded_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                                      C.POINTER(SearchOpts), C.c_uint32, vp, vp, vp, vp]
    L.scann_hip_comm_last_status.argtypes = [vp]
    L.scann_hip_comm_layout.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u64p]
    L.scann_hip_bf_search_radius.argtypes = [vp, f32p, C.c_uint32, C.c_float, u32p, f32p, C.c_uint64,
                                             C.POINTER(C.c_uint64)]
    L.scann_hip_index_size.restype = C.c_uint64
    L.scann_hip_index_size.argtypes = [vp]
    L.scann_hip_index_dimensionality.restype = C.c_uint32
    L.scann_hip_index_dimensionality.argtypes = [vp]
    L.scann_hip_index_destroy.argtypes = [vp]
    L.scann_hip_index_destroy.restype = None
    L.scann_hip_index_enable_timing.argtypes = [vp, C.c_int]
    L.scann_hip_index_enable_timing.restype = None
    L.scann_hip_index_last_kernel_ms.restype = C.c_float
    L.scann_hip_index_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_char_p)]
    _lib = L
    return L


def check(status):
    if status != OK:
        raise ScannError(status, (load().scann_hip_last_error() or b"").decode())


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


_ctx = {}


def context(device=0):
    """Process-wide context per device (scann_hip_init)."""
    if device not in _ctx:
        h = vp()
        check(load().scann_hip_init(device, C.byref(h)))
        _ctx[device] = h
    return _ctx[device]


def compute_stride(dim):
    return int(load().scann_hip_compute_stride(dim))


def default_opts():
    o = SearchOpts()
    load().scann_hip_search_opts_default(C.byref(o))
    return o


class Index:
    """Owns one scann_hip_index handle."""

    def __init__(self, handle, keep=()):
        self.h = handle
        self._keep = keep

    def close(self):
        if self.h:
            load().scann_hip_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return int(load().scann_hip_index_size(self.h))

    def dimensionality(self):
        return int(load().scann_hip_index_dimensionality(self.h))

    def search_batched(self, queries, k, opts=None, q_dim=None, stages=False, allow=None):
        """`allow`: optional uint64 allow-bitmap (see allow_bitmap()) = search_with_filter."""
        q = f32(queries)
        if q.ndim == 1:
            q = q[None]
        nq, qs = q.shape
        qd = qs if q_dim is None else q_dim
        out_idx = np.full((nq, max(k, 1)), 0xFFFFFFFF, np.uint32)
        out_dist = np.full((nq, max(k, 1)), np.inf, np.float32)
        out_cnt = np.zeros(nq, np.uint32)
        o = opts if opts is not None else default_opts()
        extra = None
        if allow is not None:
            allow = np.ascontiguousarray(allow, np.uint64)
            o.allow_bitmap, o.allow_bitmap_bits = ptr(allow, u64p), allow.size * 64
        if stages:
            P = o.partitions_to_search or 4096
            m = o.pre_reorder_k or 4096
            tok = np.zeros((nq, P), np.uint32); tokd = np.zeros((nq, P), np.float32)
            ci = np.zeros((nq, m), np.uint32); cd = np.zeros((nq, m), np.float32)
            cc = np.zeros(nq, np.uint32)
            o.tokens, o.token_dists = ptr(tok, u32p), ptr(tokd, f32p)
            o.cand_idx, o.cand_dist, o.cand_count = ptr(ci, u32p), ptr(cd, f32p), ptr(cc, u32p)
            extra = (tok, tokd, ci, cd, cc)
        try:
            check(load().scann_hip_search_batched(self.h, ptr(q, f32p), nq, qs, qd, k, C.byref(o),
                                                  ptr(out_idx, u32p), ptr(out_dist, f32p),
                                                  ptr(out_cnt, u32p)))
        finally:
            # the caller's opts must not keep pointers into this call's arrays: a later call with the same opts
            # would write its stage outputs / read its filter through them
            if stages:
                o.tokens = o.token_dists = o.cand_idx = o.cand_dist = o.cand_count = None
            if allow is not None:
                o.allow_bitmap, o.allow_bitmap_bits = None, 0
        if stages:
            return out_idx[:, :k], out_dist[:, :k], out_cnt, extra
        return out_idx[:, :k], out_dist[:, :k], out_cnt

    def search_batched_with_params(self, queries, ks, opts=None):
        """Searcher::search_batched_with_params: one num_neighbors per query; rows at pitch max(ks)."""
        q = f32(queries)
        nq, qs = q.shape
        ks = np.ascontiguousarray(ks, np.uint32)
        pitch = max(int(ks.max()), 1)
        out_idx = np.zeros((nq, pitch), np.uint32); out_dist = np.zeros((nq, pitch), np.float32)
        out_cnt = np.zeros(nq, np.uint32)
        check(load().scann_hip_search_batched_params(self.h, ptr(q, f32p), nq, qs, qs, ptr(ks, u32p),
                                                     C.byref(opts) if opts is not None else None, pitch,
                                                     ptr(out_idx, u32p), ptr(out_dist, f32p), ptr(out_cnt, u32p)))
        return out_idx, out_dist, out_cnt

    def enable_timing(self, on=True):
        load().scann_hip_index_enable_timing(self.h, 1 if on else 0)

    def last_kernel_ms(self):
        name = C.c_char_p()
        ms = load().scann_hip_index_last_kernel_ms(self.h, C.byref(name))
        return float(ms), (name.value or b"").decode()


def allow_bitmap(n, allowed):
    """uint64 bitmap with bit i set for every datapoint index in `allowed`
    (restricts/allowlist.rs semantics: listed indices are allowed)."""
    bits = np.zeros((n + 63) // 64, np.uint64)
    a = np.asarray(allowed, np.uint64)
    np.bitwise_or.at(bits, (a >> np.uint64(6)).astype(np.int64), np.uint64(1) << (a & np.uint64(63)))
    return bits


def bf_create(data, n, dim, stride, measure, device=0):
    d = f32(data)
    h = vp()
    check(load().scann_hip_bf_create(context(device), ptr(d, f32p) if n else None, n, dim,
                                     stride, measure, C.byref(h)))
    return Index(h)


def txh_create(*, data, n_rows, dim, stride, centers, leaf_offsets, leaf_ids, codebook, codes,
               codes_packed4=False, use_residuals=True, partitions_to_search=10,
               pre_reorder_multiplier=3.0, leaf_sizes_global=None, data_is_csr_order=False,
               distance_measure=SQUARED_L2, device=0):
    """codebook=None and codes=None: SearchMode::Partitioned (exact scan of the selected leaves)."""
    d, keep = _txh_desc(data=data, n_rows=n_rows, dim=dim, stride=stride, centers=centers,
                        leaf_offsets=leaf_offsets, leaf_ids=leaf_ids, codebook=codebook, codes=codes,
                        codes_packed4=codes_packed4, use_residuals=use_residuals,
                        partitions_to_search=partitions_to_search,
                        pre_reorder_multiplier=pre_reorder_multiplier, leaf_sizes_global=leaf_sizes_global,
                        data_is_csr_order=data_is_csr_order, distance_measure=distance_measure)
    h = vp()
    check(load().scann_hip_txh_create(context(device), C.byref(d), C.byref(h)))
    return Index(h)


def txh_write_file(path, *, device=None, **kw):
    """Write the index the same keyword arguments would create (no GPU needed)."""
    d, keep = _txh_desc(**kw)
    check(load().scann_hip_txh_write_file(os.fsencode(path), C.byref(d)))


def bf_write_file(path, data, n, dim, stride, measure):
    d = f32(data)
    check(load().scann_hip_bf_write_file(os.fsencode(path), ptr(d, f32p) if n else None, n, dim, stride, measure))


def index_file_info(path):
    info = FileInfo()
    check(load().scann_hip_index_file_info(os.fsencode(path), C.byref(info)))
    return {name: getattr(info, name) for name, _ in FileInfo._fields_}


def load_file(path, device=0):
    """mmap the index file and upload it (scann_hip_index_load_file)."""
    h = vp()
    check(load().scann_hip_index_load_file(context(device), os.fsencode(path), C.byref(h)))
    return Index(h)


def _txh_desc(*, data, n_rows, dim, stride, centers, leaf_offsets, leaf_ids, codebook, codes,
              codes_packed4=False, use_residuals=True, partitions_to_search=10,
              pre_reorder_multiplier=3.0, leaf_sizes_global=None, data_is_csr_order=False,
              distance_measure=SQUARED_L2):
    d = TxhDesc()
    keep = []

    def hold(a, dt):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a

    data = hold(data, np.float32)
    centers = hold(centers, np.float32)
    leaf_offsets = hold(leaf_offsets, np.uint32)
    leaf_ids = hold(leaf_ids, np.uint32)
    leaf_sizes_global = hold(leaf_sizes_global, np.uint32)
    codebook = hold(codebook, np.float32)
    codes = hold(codes, np.uint8)
    d.data = ptr(data, f32p)
    d.n_rows = n_rows
    d.dim = dim
    d.stride = stride
    d.data_is_csr_order = 1 if data_is_csr_order else 0
    d.centers = ptr(centers, f32p)
    d.num_partitions = 0 if centers is None else centers.shape[0]
    d.leaf_offsets = ptr(leaf_offsets, u32p)
    d.leaf_ids = ptr(leaf_ids, u32p)
    d.leaf_sizes_global = ptr(leaf_sizes_global, u32p)
    d.n_local = codes.shape[0] if codes is not None else leaf_ids.shape[0]
    d.codebook = ptr(codebook, f32p)
    if codebook is not None:
        d.num_subspaces, d.num_codes, d.dims_per_subspace = codebook.shape
    d.codes = ptr(codes, u8p)
    d.distance_measure = distance_measure
    d.codes_packed4 = 1 if codes_packed4 else 0
    d.use_residuals = 1 if use_residuals else 0
    d.partitions_to_search = partitions_to_search
    d.pre_reorder_multiplier = pre_reorder_multiplier
    return d, keep


def txh_partition(index, queries, num_partitions, q_dim=None):
    q = f32(queries)
    nq, qs = q.shape
    tok = np.zeros((nq, max(num_partitions, 1)), np.uint32)
    dist = np.zeros((nq, max(num_partitions, 1)), np.float32)
    cnt = np.zeros(nq, np.uint32)
    check(load().scann_hip_txh_partition(index.h, ptr(q, f32p), nq, qs, qs if q_dim is None else q_dim,
                                         num_partitions, ptr(tok, u32p), ptr(dist, f32p),
                                         ptr(cnt, u32p)))
    return tok, dist, cnt


def lut_from_query(index, queries, S, K, leaf_for_query=None):
    q = f32(queries)
    nq, qs = q.shape
    out = np.zeros((nq, S, K), np.float32)
    lf = None if leaf_for_query is None else np.ascontiguousarray(leaf_for_query, np.uint32)
    check(load().scann_hip_lut_from_query(index.h, ptr(q, f32p), nq, qs, ptr(lf, u32p),
                                          ptr(out, f32p)))
    return out


def adc_distances(index, luts):
    luts = f32(luts)
    nq = luts.shape[0]
    out = np.zeros((nq, index.size()), np.float32)
    check(load().scann_hip_adc_distances(index.h, ptr(luts, f32p), nq, ptr(out, f32p)))
    return out


def lut16_distances_batch(packed, lut8, S, n, bias, mult, device=0):
    packed = np.ascontiguousarray(packed, np.uint8)
    lut8 = np.ascontiguousarray(lut8, np.uint8)
    out = np.zeros(n, np.float32)
    check(load().scann_hip_lut16_distances_batch(context(device), ptr(packed, u8p), ptr(lut8, u8p),
                                                 S, n, bias, mult, ptr(out, f32p)))
    return out


def lut16_quantize(tables, device=0):
    """Lut16SimdTables::from_float_tables on the device: (lut8 [S][16] u8, bias, multiplier)."""
    t = f32(tables)
    S = t.shape[0]
    lut8 = np.zeros((S, 16), np.uint8)
    bias = C.c_float(0)
    mult = C.c_float(0)
    check(load().scann_hip_lut16_quantize(context(device), ptr(t, f32p) if S else None, S,
                                          ptr(lut8, u8p) if S else None, C.byref(bias), C.byref(mult)))
    return lut8, float(np.float32(bias.value)), float(np.float32(mult.value))


FP8_E4M3, FP8_E5M2 = 0, 1


def fp8_quantize(values, scale=1.0, fmt=FP8_E4M3, device=0):
    """Quantizer::quantize over Fp8Quantizer (quantization/fp8.rs:247-255) on the device."""
    v = f32(values)
    out = np.zeros(v.shape, np.uint8)
    check(load().scann_hip_fp8_quantize(context(device), ptr(v, f32p), v.size, float(np.float32(scale)), fmt,
                                        ptr(out, u8p)))
    return out


def fp8_dequantize(bits, scale=1.0, fmt=FP8_E4M3, device=0):
    b = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros(b.shape, np.float32)
    check(load().scann_hip_fp8_dequantize(context(device), ptr(b, u8p), b.size, float(np.float32(scale)), fmt,
                                          ptr(out, f32p)))
    return out


def fp8_distances(query, database, stride, n, measure, device=0):
    """one_to_many_fp8_float_{squared_l2,dot_product} (one_to_many_asymmetric.rs:327-377)."""
    q = f32(query)
    db = np.ascontiguousarray(database, np.uint8)
    out = np.zeros(n, np.float32)
    check(load().scann_hip_fp8_distances(context(device), ptr(q, f32p), q.size, ptr(db, u8p), stride, n, measure,
                                         ptr(out, f32p)))
    return out


def abi_layout():
    out = np.zeros(6, np.uint32)
    n = load().scann_hip_abi_layout(ptr(out, u32p), 6)
    return [int(v) for v in out[:n]]


_LAYOUT_KEYS = ["qr", "nq_pad", "block_bytes", "blk_idx", "blk_exact", "blk_count", "soa_bytes", "soa_idx",
                "soa_exact", "soa_count", "res_bytes", "res_dist", "blk_keys", "cap", "blk_flag", "res_status"]


def comm_layout(nq, world, m_local, k):
    """scann_hip_comm_layout as a dict (no GPU needed)."""
    out = np.zeros(16, np.uint64)
    check(load().scann_hip_comm_layout(nq, world, m_local, k, ptr(out, u64p)))
    return {n: int(v) for n, v in zip(_LAYOUT_KEYS, out)}


class Comm:
    """One RCCL communicator of the library (scann_hip_comm_*)."""

    def __init__(self, unique_id, rank, world, device=0):
        h = vp()
        self._id = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        check(load().scann_hip_comm_create(context(device), C.cast(self._id, vp), rank, world, C.byref(h)))
        self.h, self.rank, self.world = h, rank, world

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        check(load().scann_hip_comm_unique_id(C.cast(buf, vp)))
        return bytes(buf)

    def last_status(self):
        check(load().scann_hip_comm_last_status(self.h))

    def close(self):
        if self.h:
            load().scann_hip_comm_destroy(self.h)
            self.h = None


def encode(codebook, rows, stride=None, centers=None, leaf_of_row=None, device=0):
    cb = f32(codebook)
    rows = f32(rows)
    n = rows.shape[0]
    st = rows.shape[1] if stride is None else stride
    S, K, dsub = cb.shape
    out = np.zeros((n, S), np.uint8)
    cen = None if centers is None else f32(centers)
    lf = None if leaf_of_row is None else np.ascontiguousarray(leaf_of_row, np.uint32)
    check(load().scann_hip_encode(context(device), ptr(cb, f32p), S, K, dsub, ptr(rows, f32p), n,
                                  st, ptr(cen, f32p), ptr(lf, u32p), ptr(out, u8p)))
    return out


def bf_distances(index, queries):
    q = f32(queries)
    nq, qs = q.shape
    out = np.zeros((nq, index.size()), np.float32)
    check(load().scann_hip_bf_distances(index.h, ptr(q, f32p), nq, qs, ptr(out, f32p)))
    return out


def bf_search_radius(index, query, radius, capacity=None):
    """BruteForceSearcher::search_radius: (idx, dist) of every row with distance <= radius."""
    q = f32(query).reshape(-1)
    cap = index.size() if capacity is None else int(capacity)
    idx = np.zeros(max(cap, 1), np.uint32)
    dist = np.zeros(max(cap, 1), np.float32)
    cnt = C.c_uint64(0)
    check(load().scann_hip_bf_search_radius(index.h, ptr(q, f32p), q.size, C.c_float(radius),
                                            ptr(idx, u32p), ptr(dist, f32p), C.c_uint64(cap),
                                            C.byref(cnt)))
    n = min(int(cnt.value), cap)
    return idx[:n], dist[:n], int(cnt.value)


KMEANS_SIMD_THRESHOLD = 128   # KMeansConfig::default().simd_threshold (trees/kmeans.rs:59)


def kmeans_init_pp(index, k, seed, col_offset=0, sub_dim=None, simd_threshold=KMEANS_SIMD_THRESHOLD):
    """k-means++ seeding on the GPU over the rows of a brute-force index: centres [k][sub_dim]."""
    sd = index.dimensionality() if sub_dim is None else sub_dim
    out = np.zeros((k, sd), np.float32)
    check(load().scann_hip_kmeans_init_pp(index.h, col_offset, sd, k, C.c_uint64(seed), simd_threshold,
                                          ptr(out, f32p)))
    return out


def kmeans_lloyd(index, centers, max_iterations=100, convergence_threshold=1e-5, col_offset=0,
                 simd_threshold=KMEANS_SIMD_THRESHOLD):
    """KMeans::fit_single's Lloyd loop on the GPU from given centres.
    Returns (centers, assign, sizes, inertia, iterations, converged)."""
    c = np.array(centers, np.float32, copy=True, order="C")
    k, sd = c.shape
    n = index.size()
    assign = np.zeros(n, np.uint32); sizes = np.zeros(k, np.uint32)
    inertia = C.c_double(0); iters = C.c_uint32(0); conv = C.c_int(0)
    check(load().scann_hip_kmeans_lloyd(index.h, col_offset, sd, ptr(c, f32p), k, max_iterations,
                                        C.c_double(convergence_threshold), simd_threshold, ptr(assign, u32p),
                                        ptr(sizes, u32p), C.byref(inertia),
                                        C.cast(C.byref(iters), u32p), C.byref(conv)))
    return c, assign, sizes, inertia.value, iters.value, bool(conv.value)


def bf_assign_nearest(index, centers, want_dist=True):
    c = f32(centers)
    n = index.size()
    out = np.zeros(n, np.uint32)
    dist = np.zeros(n, np.float32) if want_dist else None
    check(load().scann_hip_bf_assign_nearest(index.h, ptr(c, f32p), c.shape[0], ptr(out, u32p),
                                             ptr(dist, f32p)))
    return (out, dist) if want_dist else out


def assign_leaves(sizes, world):
    sizes = np.ascontiguousarray(sizes, np.uint32)
    owner = np.zeros(sizes.size, np.uint32)
    check(load().scann_hip_assign_leaves(ptr(sizes, u32p), sizes.size, world, ptr(owner, u32p)))
    return owner
