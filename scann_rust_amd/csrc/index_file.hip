// index_file.hip -- the on-disk index container (include/scann_hip.h "index files"): writer,
// header inspection, and the mmap -> (pinned) -> device loader.  Host code only.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace {

using scann::fail;

constexpr char kMagic[8] = {'S', 'C', 'A', 'N', 'N', 'I', 'D', 'X'};
constexpr uint32_t kVersion = 1;
constexpr uint64_t kAlign = 4096;
enum { DT_F32 = 0, DT_U32 = 1, DT_U8 = 2 };

struct FileHeader {   // 256 bytes, little-endian, no implicit padding
    char magic[8];
    uint32_t version, kind;
    uint64_t n_rows, n_local, file_bytes;
    uint32_t dim, stride, num_partitions, num_subspaces, num_codes, dims_per_subspace;
    int32_t distance_measure, data_is_csr_order, codes_packed4, use_residuals;
    uint32_t partitions_to_search;
    float pre_reorder_multiplier;
    uint32_t n_sections;
    uint32_t reserved[41];
};
static_assert(sizeof(FileHeader) == 256, "header layout");

struct FileSection {  // 64 bytes
    char name[24];
    uint32_t dtype, reserved0;
    uint64_t offset, nbytes;
    uint64_t reserved1[2];
};
static_assert(sizeof(FileSection) == 64, "section layout");

struct PendingSection {
    const char *name;
    uint32_t dtype;
    const void *ptr;
    uint64_t nbytes;
};

int write_container(const char *path, FileHeader h, const std::vector<PendingSection> &secs) {
    std::vector<FileSection> table(secs.size());
    uint64_t off = sizeof(FileHeader) + sizeof(FileSection) * secs.size();
    for (size_t i = 0; i < secs.size(); ++i) {
        off = (off + kAlign - 1) / kAlign * kAlign;
        std::memset(&table[i], 0, sizeof(FileSection));
        std::strncpy(table[i].name, secs[i].name, sizeof(table[i].name) - 1);
        table[i].dtype = secs[i].dtype;
        table[i].offset = off;
        table[i].nbytes = secs[i].nbytes;
        off += secs[i].nbytes;
    }
    h.n_sections = (uint32_t)secs.size();
    h.file_bytes = off;
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(errno == EACCES ? SCANN_HIP_PERMISSION_DENIED : SCANN_HIP_NOT_FOUND,
                        std::string("cannot create ") + path + ": " + std::strerror(errno));
    bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1;
    if (ok && !table.empty()) ok = std::fwrite(table.data(), sizeof(FileSection), table.size(), f) == table.size();
    uint64_t pos = sizeof(FileHeader) + sizeof(FileSection) * secs.size();
    static const char zeros[4096] = {0};
    for (size_t i = 0; ok && i < secs.size(); ++i) {
        while (ok && pos < table[i].offset) {
            const size_t pad = (size_t)std::min<uint64_t>(sizeof(zeros), table[i].offset - pos);
            ok = std::fwrite(zeros, 1, pad, f) == pad;
            pos += pad;
        }
        if (ok && secs[i].nbytes) ok = std::fwrite(secs[i].ptr, 1, secs[i].nbytes, f) == secs[i].nbytes;
        pos += secs[i].nbytes;
    }
    if (std::fclose(f) != 0) ok = false;
    if (!ok) {
        std::remove(path);
        return fail(SCANN_HIP_DATA_LOSS, std::string("short write to ") + path);
    }
    return SCANN_HIP_OK;
}

FileHeader blank_header(uint32_t kind) {
    FileHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, kMagic, 8);
    h.version = kVersion;
    h.kind = kind;
    return h;
}

// a read-only mapping of a validated file
struct Mapping {
    void *base = MAP_FAILED;
    size_t bytes = 0;
    bool pinned = false;
    ~Mapping() {
        if (pinned) (void)hipHostUnregister(base);
        if (base != MAP_FAILED) munmap(base, bytes);
    }
    const FileHeader &header() const { return *static_cast<const FileHeader *>(base); }
    const FileSection *find(const char *name) const {
        const FileSection *t = reinterpret_cast<const FileSection *>(static_cast<const char *>(base) + sizeof(FileHeader));
        for (uint32_t i = 0; i < header().n_sections; ++i)
            if (std::strncmp(t[i].name, name, sizeof(t[i].name)) == 0) return &t[i];
        return nullptr;
    }
    const void *data(const FileSection *s) const { return s ? static_cast<const char *>(base) + s->offset : nullptr; }
};

int open_mapping(const char *path, Mapping *m) {
    if (!path) return fail(SCANN_HIP_INVALID_ARGUMENT, "path is null");
    const int fd = open(path, O_RDONLY);
    if (fd < 0)
        return fail(errno == EACCES ? SCANN_HIP_PERMISSION_DENIED : SCANN_HIP_NOT_FOUND,
                    std::string("cannot open ") + path + ": " + std::strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return fail(SCANN_HIP_INTERNAL, std::string("fstat failed on ") + path);
    }
    if ((uint64_t)st.st_size < sizeof(FileHeader)) {
        close(fd);
        return fail(SCANN_HIP_DATA_LOSS, std::string(path) + ": shorter than an index header");
    }
    m->bytes = (size_t)st.st_size;
    m->base = mmap(nullptr, m->bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m->base == MAP_FAILED) return fail(SCANN_HIP_RESOURCE_EXHAUSTED, std::string("mmap failed on ") + path);
    const FileHeader &h = m->header();
    if (std::memcmp(h.magic, kMagic, 8) != 0)
        return fail(SCANN_HIP_INVALID_ARGUMENT, std::string(path) + ": not a SCANNIDX file");
    if (h.version != kVersion)
        return fail(SCANN_HIP_INVALID_ARGUMENT,
                    std::string(path) + ": unsupported index file version " + std::to_string(h.version));
    if (h.file_bytes != (uint64_t)st.st_size)
        return fail(SCANN_HIP_DATA_LOSS, std::string(path) + ": file is " + std::to_string(st.st_size) +
                                             " bytes, header says " + std::to_string(h.file_bytes));
    if (h.kind > 1 || h.n_sections > 64 ||
        sizeof(FileHeader) + (uint64_t)h.n_sections * sizeof(FileSection) > (uint64_t)st.st_size)
        return fail(SCANN_HIP_DATA_LOSS, std::string(path) + ": corrupt header");
    const FileSection *t = reinterpret_cast<const FileSection *>(static_cast<const char *>(m->base) + sizeof(FileHeader));
    for (uint32_t i = 0; i < h.n_sections; ++i)
        if (t[i].offset % kAlign != 0 || t[i].offset > h.file_bytes || t[i].nbytes > h.file_bytes - t[i].offset)
            return fail(SCANN_HIP_DATA_LOSS, std::string(path) + ": section table out of bounds");
    return SCANN_HIP_OK;
}

// a * b * c, or UINT64_MAX (which no section of a validated file can have) on overflow
uint64_t bytes3(uint64_t a, uint64_t b, uint64_t c) {
    uint64_t ab, abc;
    if (__builtin_mul_overflow(a, b, &ab) || __builtin_mul_overflow(ab, c, &abc)) return ~0ull;
    return abc;
}

// section `name` must exist with exactly `want` bytes (or be absent when !required)
int section(const Mapping &m, const char *path, const char *name, uint64_t want, bool required,
            const void **out) {
    const FileSection *s = m.find(name);
    *out = nullptr;
    if (!s) {
        if (required) return fail(SCANN_HIP_DATA_LOSS, std::string(path) + ": section '" + name + "' is missing");
        return SCANN_HIP_OK;
    }
    if (s->nbytes != want)
        return fail(SCANN_HIP_DATA_LOSS, std::string(path) + ": section '" + name + "' has " +
                                             std::to_string(s->nbytes) + " bytes, expected " + std::to_string(want));
    *out = m.data(s);
    return SCANN_HIP_OK;
}

}  // namespace

extern "C" {

int scann_hip_txh_write_file(const char *path, const scann_hip_txh_desc *d) {
    if (!path || !d) return fail(SCANN_HIP_INVALID_ARGUMENT, "path/desc is null");
    const bool ah = d->num_partitions == 0;
    const bool exact = !d->codebook && !d->codes && d->num_subspaces == 0;
    if (d->n_local == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "Cannot build from empty dataset");
    if (!exact && (!d->codebook || !d->codes)) return fail(SCANN_HIP_INVALID_ARGUMENT, "codebook/codes null");
    if (!ah && (!d->centers || !d->leaf_offsets || !d->leaf_ids))
        return fail(SCANN_HIP_INVALID_ARGUMENT, "centers/leaf_offsets/leaf_ids null");
    FileHeader h = blank_header(1);
    h.n_rows = d->n_rows;
    h.n_local = d->n_local;
    h.dim = d->dim;
    h.stride = d->stride;
    h.num_partitions = d->num_partitions;
    h.num_subspaces = d->num_subspaces;
    h.num_codes = d->num_codes;
    h.dims_per_subspace = d->dims_per_subspace;
    h.distance_measure = d->distance_measure;
    h.data_is_csr_order = d->data_is_csr_order;
    h.codes_packed4 = d->codes_packed4;
    h.use_residuals = d->use_residuals;
    h.partitions_to_search = d->partitions_to_search;
    h.pre_reorder_multiplier = d->pre_reorder_multiplier;
    std::vector<PendingSection> secs;
    if (d->data) secs.push_back({"data", DT_F32, d->data, (uint64_t)d->n_rows * d->stride * 4});
    if (!ah) {
        secs.push_back({"centers", DT_F32, d->centers, (uint64_t)d->num_partitions * d->dim * 4});
        secs.push_back({"leaf_offsets", DT_U32, d->leaf_offsets, ((uint64_t)d->num_partitions + 1) * 4});
        secs.push_back({"leaf_ids", DT_U32, d->leaf_ids, d->n_local * 4});
        if (d->leaf_sizes_global)
            secs.push_back({"leaf_sizes_global", DT_U32, d->leaf_sizes_global, (uint64_t)d->num_partitions * 4});
    }
    if (!exact) {
        const uint64_t bpp = d->codes_packed4 ? (d->num_subspaces + 1) / 2 : d->num_subspaces;
        secs.push_back({"codebook", DT_F32, d->codebook,
                        (uint64_t)d->num_subspaces * d->num_codes * d->dims_per_subspace * 4});
        secs.push_back({"codes", DT_U8, d->codes, d->n_local * bpp});
    }
    return write_container(path, h, secs);
}

int scann_hip_bf_write_file(const char *path, const float *data, uint64_t n, uint32_t dim, uint32_t stride,
                            int measure) {
    if (!path || (n && !data)) return fail(SCANN_HIP_INVALID_ARGUMENT, "path/data is null");
    if (stride < dim) return fail(SCANN_HIP_INVALID_ARGUMENT, "stride < dim");
    FileHeader h = blank_header(0);
    h.n_rows = n;
    h.n_local = n;
    h.dim = dim;
    h.stride = stride;
    h.distance_measure = measure;
    std::vector<PendingSection> secs;
    secs.push_back({"data", DT_F32, data, n * stride * 4});
    return write_container(path, h, secs);
}

int scann_hip_index_file_info(const char *path, scann_hip_file_info *out) {
    if (!out) return fail(SCANN_HIP_INVALID_ARGUMENT, "out_info is null");
    Mapping m;
    SCANN_TRY(open_mapping(path, &m));
    const FileHeader &h = m.header();
    std::memset(out, 0, sizeof(*out));
    out->version = h.version;
    out->kind = h.kind;
    out->n_rows = h.n_rows;
    out->n_local = h.n_local;
    out->file_bytes = h.file_bytes;
    out->dim = h.dim;
    out->stride = h.stride;
    out->num_partitions = h.num_partitions;
    out->num_subspaces = h.num_subspaces;
    out->num_codes = h.num_codes;
    out->dims_per_subspace = h.dims_per_subspace;
    out->distance_measure = h.distance_measure;
    out->data_is_csr_order = h.data_is_csr_order;
    out->codes_packed4 = h.codes_packed4;
    out->use_residuals = h.use_residuals;
    out->partitions_to_search = h.partitions_to_search;
    out->pre_reorder_multiplier = h.pre_reorder_multiplier;
    out->has_data = m.find("data") ? 1 : 0;
    return SCANN_HIP_OK;
}

int scann_hip_index_load_file(scann_hip_ctx *ctx, const char *path, scann_hip_index **out) {
    if (!ctx || !out) return fail(SCANN_HIP_INVALID_ARGUMENT, "null ctx/out_index");
    Mapping m;
    SCANN_TRY(open_mapping(path, &m));
    const FileHeader &h = m.header();
    // Pin the mapping so the uploads DMA straight from the page cache.  Best effort: a driver
    // that cannot pin a read-only file mapping leaves the (staged) pageable copy path.
    const char *pin = std::getenv("SCANN_HIP_LOAD_PIN");
    if (!(pin && std::atoi(pin) == 0)) {
        if (hipHostRegister(m.base, m.bytes, hipHostRegisterReadOnly) == hipSuccess) m.pinned = true;
        else (void)hipGetLastError();
    }
    const void *p = nullptr;
    if (h.kind == 0) {
        SCANN_TRY(section(m, path, "data", bytes3(h.n_rows, h.stride, 4), true, &p));
        return scann_hip_bf_create(ctx, static_cast<const float *>(p), h.n_rows, h.dim, h.stride,
                                   h.distance_measure, out);
    }
    scann_hip_txh_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_rows = h.n_rows;
    d.n_local = h.n_local;
    d.dim = h.dim;
    d.stride = h.stride;
    d.num_partitions = h.num_partitions;
    d.num_subspaces = h.num_subspaces;
    d.num_codes = h.num_codes;
    d.dims_per_subspace = h.dims_per_subspace;
    d.distance_measure = h.distance_measure;
    d.data_is_csr_order = h.data_is_csr_order;
    d.codes_packed4 = h.codes_packed4;
    d.use_residuals = h.use_residuals;
    d.partitions_to_search = h.partitions_to_search;
    d.pre_reorder_multiplier = h.pre_reorder_multiplier;
    SCANN_TRY(section(m, path, "data", bytes3(h.n_rows, h.stride, 4), false, &p));
    d.data = static_cast<const float *>(p);
    if (h.num_partitions) {
        SCANN_TRY(section(m, path, "centers", bytes3(h.num_partitions, h.dim, 4), true, &p));
        d.centers = static_cast<const float *>(p);
        SCANN_TRY(section(m, path, "leaf_offsets", ((uint64_t)h.num_partitions + 1) * 4, true, &p));
        d.leaf_offsets = static_cast<const uint32_t *>(p);
        SCANN_TRY(section(m, path, "leaf_ids", bytes3(h.n_local, 4, 1), true, &p));
        d.leaf_ids = static_cast<const uint32_t *>(p);
        SCANN_TRY(section(m, path, "leaf_sizes_global", (uint64_t)h.num_partitions * 4, false, &p));
        d.leaf_sizes_global = static_cast<const uint32_t *>(p);
    }
    if (h.num_subspaces) {
        const uint64_t bpp = h.codes_packed4 ? (h.num_subspaces + 1) / 2 : h.num_subspaces;
        SCANN_TRY(section(m, path, "codebook", bytes3(bytes3(h.num_subspaces, h.num_codes, h.dims_per_subspace), 4, 1),
                          true, &p));
        d.codebook = static_cast<const float *>(p);
        SCANN_TRY(section(m, path, "codes", bytes3(h.n_local, bpp, 1), true, &p));
        d.codes = static_cast<const uint8_t *>(p);
    }
    return scann::txh_create_checked(ctx, &d, out, SCANN_HIP_DATA_LOSS);
}

}  // extern "C"
