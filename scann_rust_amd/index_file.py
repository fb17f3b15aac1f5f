"""numpy reader / writer of the SCANNIDX index container (include/scann_hip.h "index files").

The same bytes scann_hip_txh_write_file / scann_hip_index_load_file handle, so that the CPU oracle,
the golden fixtures and any offline tool can share index files with the GPU library without going
through it.  Arrays come back as read-only np.memmap views (nothing is copied).
"""
import struct

import numpy as np

MAGIC = b"SCANNIDX"
VERSION = 1
ALIGN = 4096
_DTYPES = {0: np.float32, 1: np.uint32, 2: np.uint8}
_CODES = {np.dtype(np.float32): 0, np.dtype(np.uint32): 1, np.dtype(np.uint8): 2}
# magic, version, kind, n_rows, n_local, file_bytes, dim, stride, L, S, K, dsub, measure, csr, packed4,
# residuals, partitions_to_search, multiplier, n_sections
_HEADER = struct.Struct("<8sIIQQQIIIIIIiiiiIfI")
_SECTION = struct.Struct("<24sIIQQ16x")
_FIELDS = ["version", "kind", "n_rows", "n_local", "file_bytes", "dim", "stride", "num_partitions",
           "num_subspaces", "num_codes", "dims_per_subspace", "distance_measure", "data_is_csr_order",
           "codes_packed4", "use_residuals", "partitions_to_search", "pre_reorder_multiplier", "n_sections"]


def read(path):
    """-> (header dict, {section name: np.memmap})."""
    with open(path, "rb") as f:
        raw = f.read(256)
        if len(raw) < 256:
            raise ValueError("DataLoss: %s is shorter than an index header" % path)
        vals = _HEADER.unpack_from(raw)
        if vals[0] != MAGIC:
            raise ValueError("InvalidArgument: %s is not a SCANNIDX file" % path)
        header = dict(zip(_FIELDS, vals[1:]))
        if header["version"] != VERSION:
            raise ValueError("InvalidArgument: unsupported index file version %d" % header["version"])
        table = f.read(64 * header["n_sections"])
        f.seek(0, 2)
        if f.tell() != header["file_bytes"]:
            raise ValueError("DataLoss: %s is %d bytes, header says %d" % (path, f.tell(), header["file_bytes"]))
    sections = {}
    for i in range(header["n_sections"]):
        name, dtype, _, off, nbytes = _SECTION.unpack_from(table, 64 * i)
        dt = np.dtype(_DTYPES[dtype])
        if off % ALIGN or off + nbytes > header["file_bytes"] or nbytes % dt.itemsize:
            raise ValueError("DataLoss: section table out of bounds")
        sections[name.rstrip(b"\0").decode()] = np.memmap(path, dt, "r", off, (nbytes // dt.itemsize,)) \
            if nbytes else np.empty(0, dt)
    return header, sections


def arrays(path):
    """The file's arrays in their natural shapes: the keyword arguments of hip.txh_create /
    the CPU checker's index view (kind 1) or (data, n, dim, stride, measure) pieces (kind 0)."""
    h, s = read(path)
    out = dict(h)
    if "data" in s:
        out["data"] = s["data"].reshape(h["n_rows"], h["stride"])
    if h["kind"] == 1:
        if h["num_partitions"]:
            out["centers"] = s["centers"].reshape(h["num_partitions"], h["dim"])
            out["leaf_offsets"] = s["leaf_offsets"]
            out["leaf_ids"] = s["leaf_ids"]
            if "leaf_sizes_global" in s:
                out["leaf_sizes_global"] = s["leaf_sizes_global"]
        if h["num_subspaces"]:
            out["codebook"] = s["codebook"].reshape(h["num_subspaces"], h["num_codes"], h["dims_per_subspace"])
            bpp = (h["num_subspaces"] + 1) // 2 if h["codes_packed4"] else h["num_subspaces"]
            out["codes"] = s["codes"].reshape(h["n_local"], bpp)
    return out


def write(path, *, kind, n_rows, n_local, dim, stride, num_partitions=0, num_subspaces=0, num_codes=0,
          dims_per_subspace=0, distance_measure=0, data_is_csr_order=0, codes_packed4=0, use_residuals=0,
          partitions_to_search=0, pre_reorder_multiplier=0.0, sections=()):
    """sections: ordered (name, array) pairs; dtype f32 / u32 / u8."""
    secs = [(n, np.ascontiguousarray(a)) for n, a in sections]
    off = 256 + 64 * len(secs)
    table = b""
    offsets = []
    for name, a in secs:
        off = (off + ALIGN - 1) // ALIGN * ALIGN
        table += _SECTION.pack(name.encode(), _CODES[a.dtype], 0, off, a.nbytes)
        offsets.append(off)
        off += a.nbytes
    head = _HEADER.pack(MAGIC, VERSION, kind, n_rows, n_local, off, dim, stride, num_partitions, num_subspaces,
                        num_codes, dims_per_subspace, distance_measure, data_is_csr_order, codes_packed4,
                        use_residuals, partitions_to_search, pre_reorder_multiplier, len(secs))
    with open(path, "wb") as f:
        f.write(head.ljust(256, b"\0"))
        f.write(table)
        for (name, a), o in zip(secs, offsets):
            f.write(b"\0" * (o - f.tell()))
            f.write(a.tobytes())
