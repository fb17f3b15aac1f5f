// api.hip -- C ABI of libscann_hip.so (include/scann_hip.h): contexts, index handles,
// workspace management and host orchestration.  No CPU compute fallback exists: every
// search entry point runs the HIP kernels or fails.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "bf.h"
#include "comm.h"
#include "common.h"
#include "txh.h"

namespace scann {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
int fail(int status, const std::string &msg) {
    g_last_error = msg;
    return status;
}

}  // namespace scann

using namespace scann;

struct scann_hip_ctx {
    int device = 0;
    int num_cus = 256;
};

enum IndexKind { KIND_BF = 1, KIND_TXH = 2 };

// A buffer of a search workspace: a range of the workspace's ARENA (one device allocation, sub-allocated at 256-byte
// granularity in declaration order).
struct WsBuf {
    void *p = nullptr;
    size_t bytes = 0;   // capacity assigned in the arena
    size_t need = 0;    // largest size any call has asked for
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

// Per-call scratch of the Tree-X-Hybrid pipeline.  Every buffer lives in ONE device allocation: a call first states
// what it needs (want), then commit() rebuilds the arena if any buffer has to grow -- one hipFree (which waits for the
// device) + one hipMalloc, every buffer re-placed in declaration order at the largest size seen so far.  Placement
// is therefore a function of the sizes alone, never of the order in which a handle met its batch sizes and
// pre_reorder_k values: round 2 allocated every buffer separately, and a workspace that had grown buffer by buffer
// (free one, allocate it larger, next) left rerank_short_kernel 3x slower than a fresh one of the same sizes.
struct TxhWorkspace {
    WsBuf queries, cdist, tokens, token_dists, vbase, leaf_cnt, leaf_cursor, pair_off, tile_off,
        counters, pair_q, pair_leaf, pair_vbase, pair_thr, slot_of, lutq, thr, cand_cnt, cand, cand_key,
        cand_idx, cand_dist, cand_exact, cand_row, cand_count, out_idx, out_dist, out_count, allow,
        sbase, pair_sbase, stile_off, samp, lut8, lut8_meta, cand32, cand32_codes, cand32_cnt, mfma_thr1, rr_lb, rr_ub, small_tickets,
          wide_min, wide_ckey, wide_ceb, wide_cidx, wide_cnt;
    void *arena = nullptr;
    size_t arena_bytes = 0;
    bool dirty = false;
    uint32_t rebuilds = 0;
    TxhWorkspace() = default;
    TxhWorkspace(const TxhWorkspace &) = delete;
    TxhWorkspace &operator=(const TxhWorkspace &) = delete;
    ~TxhWorkspace() { release_all(); }
    template <typename F>
    void for_each(F f) {
        WsBuf *all[] = {&queries, &cdist, &tokens, &token_dists, &vbase, &leaf_cnt, &leaf_cursor, &pair_off, &tile_off,
                        &counters, &pair_q, &pair_leaf, &pair_vbase, &pair_thr, &slot_of, &lutq, &thr, &cand_cnt, &cand,
                        &cand_key, &cand_idx, &cand_dist, &cand_exact, &cand_row, &cand_count, &out_idx, &out_dist,
                        &out_count, &allow, &sbase, &pair_sbase, &stile_off, &samp, &lut8, &lut8_meta, &cand32,
                        &cand32_codes, &cand32_cnt, &mfma_thr1, &rr_lb, &rr_ub, &small_tickets,
                        &wide_min, &wide_ckey, &wide_ceb, &wide_cidx, &wide_cnt};
        for (WsBuf *b : all) f(*b);
    }
    void want(WsBuf &b, size_t bytes) {
        if (bytes == 0) bytes = 16;
        if (bytes > b.need) b.need = bytes;
        if (b.need > b.bytes) dirty = true;
    }
    int commit() {
        if (!dirty) return SCANN_HIP_OK;
        // Every buffer starts at its own skew inside a 128 KB window.  Back to back, the six per-candidate arrays of a
        // batch ([nq][m] words each) lie exactly nq * m * 4 bytes apart -- 32 MB at nq = 1024, m = 8192 -- so that
        // element (q, i) of ALL of them maps to the same HBM channel and bank: rerank_short_kernel, which walks
        // several of them in step, ran 3x slower on such a layout (0.41 vs 0.14 ms; same instructions, same bytes:
        // round 2's "workspace growth" slowdown was this aliasing, present or absent by the accident of which other
        // buffers the handle had allocated in between).
        auto skew = [](size_t i) { return (((i + 1) * 37) % 509) * 256; };
        size_t total = 0, idx = 0;
        for_each([&](WsBuf &b) { total += ((b.need + 255) & ~(size_t)255) + skew(idx++); });
        if (arena) (void)hipFree(arena);   // (waits for the device: kernels of earlier calls may still read it)
        arena = nullptr;
        arena_bytes = 0;
        for_each([&](WsBuf &b) { b.p = nullptr; b.bytes = 0; });
        SCANN_HIP_CHECK(hipMalloc(&arena, total ? total : 256));
        arena_bytes = total;
        size_t off = 0;
        idx = 0;
        for_each([&](WsBuf &b) {
            off += skew(idx++);
            if (!b.need) return;
            b.p = static_cast<char *>(arena) + off;
            b.bytes = (b.need + 255) & ~(size_t)255;
            off += b.bytes;
        });
        // ticket counters of small_fused_kernel: zero between launches (the kernel leaves them zero)
        if (small_tickets.p) SCANN_HIP_CHECK(hipMemset(small_tickets.p, 0, small_tickets.bytes));
        dirty = false;
        ++rebuilds;
        return SCANN_HIP_OK;
    }
    void release_all() {
        if (arena) (void)hipFree(arena);
        arena = nullptr;
        arena_bytes = 0;
        for_each([&](WsBuf &b) { b = WsBuf(); });
        dirty = false;
    }
};

// Pinned host memory the GPU reads and writes in place (grow-only).  Small host-side searches keep
// their queries and result rows here: the kernels load the queries over the host link and store the
// rows straight into host memory, so a call is launches + ONE stream sync, with no copy commands (each
// hipMemcpyAsync to or from pageable memory costs 5-15 us of host time: five of them were most of a
// single-query call).
struct PinBuf {
    void *host = nullptr, *dev = nullptr;
    size_t bytes = 0;
    uint32_t seq = 0;   // completion-flag value of the last small call staged here
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { release(); }
    void release() {
        if (host) (void)hipHostFree(host);
        host = dev = nullptr;
        bytes = 0;
    }
    int ensure(size_t need) {
        if (need <= bytes && host) return SCANN_HIP_OK;
        release();
        need = (need + 4095) & ~(size_t)4095;
        SCANN_HIP_CHECK(hipHostMalloc(&host, need, hipHostMallocMapped));
        SCANN_HIP_CHECK(hipHostGetDevicePointer(&dev, host, 0));
        std::memset(host, 0, need);   // (completion flags: no stale word may equal a live sequence number)
        bytes = need;
        return SCANN_HIP_OK;
    }
};

// An extra stream + workspaces: host-side searches of concurrent caller threads (Searcher: Send +
// Sync, tests/stress_tests.rs:256-297) run side by side instead of queueing on one mutex.
struct SearchSlot {
    std::mutex mu;
    hipStream_t stream = nullptr;
    TxhWorkspace ws;
    BfWorkspace bfw;
    PinBuf pin;
};

struct scann_hip_index {
    scann_hip_ctx *ctx = nullptr;
    int kind = 0;
    std::mutex mu;  // the primary slot (stream, ws, bfw below): device entry points and timing use it
    std::mutex slots_mu;
    std::vector<std::unique_ptr<SearchSlot>> slots;   // created on demand, at most max_slots() - 1
    hipStream_t stream = nullptr;
    // ring of HIP event pairs bracketing the dominant kernel of each search launch
    static constexpr int kEvRing = 64;
    hipEvent_t evs[kEvRing][2] = {};
    uint32_t ev_n = 0;      // launches recorded since timing was enabled
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // pair handed to the current launch
    bool timing = false;
    bool timing_valid = false;
    void next_events() {
        if (!timing) {
            ev0 = ev1 = nullptr;
            return;
        }
        ev0 = evs[ev_n % kEvRing][0];
        ev1 = evs[ev_n % kEvRing][1];
        ++ev_n;
    }
    const char *timed_kernel = "";
    DevBuf status_word;

    // ---- brute force ----
    BfIndexDev bf{};
    DevBuf bf_rows, bf_rows_b, bf_rows_bl, bf_norm2, bf_leaf;
    BfWorkspace bfw;
    TxhIndexDev bfx{};        // the same rows as a one-leaf exact-scan index: the small-batch pipeline's view

    // ---- tree-x-hybrid / AH ----
    TxhIndexDev tx{};
    DevBuf d_centers, d_centers_t, d_leaf_off, d_leaf_gsize, d_leaf_ids, d_codes, d_codes_sp, d_rows, d_codebook, d_rows8, d_rows8_meta;
    std::vector<uint32_t> local_sizes_desc;  // local leaf sizes, descending, prefix-summed
    uint32_t default_P = 0;
    float multiplier = 3.0f;
    TxhWorkspace ws;
    PinBuf pin;               // primary slot's pinned staging
    TxhWork last_work{};
    // Workspaces of the `_device` entry points, one per CALLER STREAM (scann_hip.h "Device entry points"): slot 0 is
    // the primary workspace (ws / bfw above), further ones are created on demand.  A slot that changes streams is
    // ordered behind its previous stream's last call with an event.
    struct DeviceSlot {
        hipStream_t key = nullptr;
        bool used = false, done_valid = false;
        hipEvent_t done = nullptr;
        uint64_t tick = 0;
        TxhWorkspace *ws = nullptr;
        BfWorkspace *bfw = nullptr;
        std::unique_ptr<TxhWorkspace> ws_own;
        std::unique_ptr<BfWorkspace> bfw_own;
    };
    static constexpr int kMaxDeviceSlots = 4;
    DeviceSlot dslots[kMaxDeviceSlots];
    uint64_t dslot_tick = 0;
    bool sharded = false;     // created with leaf_sizes_global: local leaves are a subset of the global stream
};

static int set_device(const scann_hip_ctx *ctx) {
    SCANN_HIP_CHECK(hipSetDevice(ctx->device));
    return SCANN_HIP_OK;
}

int scann::ctx_device(const scann_hip_ctx *ctx) { return ctx ? ctx->device : 0; }

extern "C" {

const char *scann_hip_last_error(void) { return g_last_error.c_str(); }
const char *scann_hip_version(void) { return "scann_hip 0.1.0 (gfx950)"; }

uint32_t scann_hip_abi_layout(uint32_t *out, uint32_t n) {
    const uint32_t v[6] = {(uint32_t)sizeof(scann_hip_txh_desc), (uint32_t)offsetof(scann_hip_txh_desc, distance_measure),
                           (uint32_t)sizeof(scann_hip_search_opts), (uint32_t)offsetof(scann_hip_search_opts, bf_exact),
                           (uint32_t)sizeof(scann_hip_file_info), (uint32_t)offsetof(scann_hip_file_info, has_data)};
    for (uint32_t i = 0; i < 6 && i < n && out; ++i) out[i] = v[i];
    return 6;
}

uint32_t scann_hip_compute_stride(uint32_t dim) {
    const uint32_t per_line = 64 / sizeof(float);  // data_format/dataset.rs:90-96
    return (dim + per_line - 1) / per_line * per_line;
}

int scann_hip_init(int device_id, scann_hip_ctx **out_ctx) {
    if (!out_ctx) return fail(SCANN_HIP_INVALID_ARGUMENT, "out_ctx is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(SCANN_HIP_UNAVAILABLE,
                    "no HIP device available (libscann_hip has no CPU fallback)");
    if (device_id < 0 || device_id >= n)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "device_id out of range");
    SCANN_HIP_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    SCANN_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(SCANN_HIP_FAILED_PRECONDITION,
                    std::string("libscann_hip is built for gfx950 only; device is ") +
                        prop.gcnArchName);
    auto *c = new scann_hip_ctx();
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount;
    *out_ctx = c;
    return SCANN_HIP_OK;
}

void scann_hip_shutdown(scann_hip_ctx *ctx) { delete ctx; }

void scann_hip_search_opts_default(scann_hip_search_opts *o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->exact_reorder = 1;
}

static int index_common_init(scann_hip_index *ix) {
    SCANN_HIP_CHECK(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    for (int i = 0; i < scann_hip_index::kEvRing; ++i) {
        SCANN_HIP_CHECK(hipEventCreate(&ix->evs[i][0]));
        SCANN_HIP_CHECK(hipEventCreate(&ix->evs[i][1]));
    }
    return SCANN_HIP_OK;
}

void scann_hip_index_destroy(scann_hip_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->ctx->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    for (int i = 0; i < scann_hip_index::kEvRing; ++i) {
        if (ix->evs[i][0]) (void)hipEventDestroy(ix->evs[i][0]);
        if (ix->evs[i][1]) (void)hipEventDestroy(ix->evs[i][1]);
    }
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    (void)hipDeviceSynchronize();   // device entry points enqueue on caller streams
    for (auto &ds : ix->dslots)
        if (ds.done) (void)hipEventDestroy(ds.done);
    for (auto &sl : ix->slots)
        if (sl->stream) {
            (void)hipStreamSynchronize(sl->stream);
            (void)hipStreamDestroy(sl->stream);
        }
    delete ix;
}

uint64_t scann_hip_index_size(const scann_hip_index *ix) {
    if (!ix) return 0;
    return ix->kind == KIND_BF ? ix->bf.n : ix->tx.n_local;
}
uint32_t scann_hip_index_dimensionality(const scann_hip_index *ix) {
    if (!ix) return 0;
    return ix->kind == KIND_BF ? ix->bf.dim : ix->tx.dim;
}

void scann_hip_index_enable_timing(scann_hip_index *ix, int enable) {
    if (!ix) return;
    ix->timing = enable != 0;
    ix->ev_n = 0;
    ix->timing_valid = false;
}

float scann_hip_index_last_kernel_ms(scann_hip_index *ix, const char **name) {
    if (name) *name = ix ? ix->timed_kernel : "";
    if (!ix || !ix->timing_valid || ix->ev_n == 0) return 0.0f;
    const uint32_t cnt = ix->ev_n < (uint32_t)scann_hip_index::kEvRing
                             ? ix->ev_n : (uint32_t)scann_hip_index::kEvRing;
    double sum = 0.0;
    uint32_t ok = 0;
    for (uint32_t i = 0; i < cnt; ++i) {   // mean over the recorded launches
        float ms = 0.0f;
        if (hipEventSynchronize(ix->evs[i][1]) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms, ix->evs[i][0], ix->evs[i][1]) != hipSuccess) continue;
        sum += ms;
        ++ok;
    }
    return ok ? (float)(sum / ok) : 0.0f;
}

// =====================================================================================
// Brute force
// =====================================================================================
int scann_hip_bf_create(scann_hip_ctx *ctx, const float *data, uint64_t n, uint32_t dim,
                        uint32_t stride, int measure, scann_hip_index **out) {
    if (!ctx || !out) return fail(SCANN_HIP_INVALID_ARGUMENT, "null ctx/out_index");
    if (n > 0 && !data) return fail(SCANN_HIP_INVALID_ARGUMENT, "data is null");
    if (n > 0 && (dim == 0 || stride < dim))
        return fail(SCANN_HIP_INVALID_ARGUMENT, "bad dim/stride");
    if (measure < SCANN_HIP_SQUARED_L2 || measure > SCANN_HIP_COSINE)
        return fail(SCANN_HIP_UNIMPLEMENTED,
                    "brute force supports SquaredL2, L2, DotProduct, L1 and Cosine on the GPU path");
    if (n >= 0xFFFFFFFFull) return fail(SCANN_HIP_OUT_OF_RANGE, "DatapointIndex is u32");
    SCANN_TRY(set_device(ctx));
    auto *ix = new scann_hip_index();
    ix->ctx = ctx;
    ix->kind = KIND_BF;
    int s = index_common_init(ix);
    if (s == SCANN_HIP_OK) s = upload(ix->bf_rows, data, (size_t)n * stride * sizeof(float));
    if (s != SCANN_HIP_OK) {
        scann_hip_index_destroy(ix);
        return s;
    }
    ix->bf.rows = ix->bf_rows.as<float>();
    ix->bf.n = n;
    ix->bf.dim = dim;
    ix->bf.stride = stride;
    ix->bf.measure = measure;
    ix->bf.rows_b = nullptr;
    ix->bf.rows_bl = nullptr;
    ix->bf.norm2 = nullptr;
    ix->bf.max_norm = 0.0f;
    if (n > 0 && n <= kSmallMaxStream) {   // view for the three-launch small-batch pipeline (txh.hip)
        const uint32_t leaf[3] = {0u, (uint32_t)n, (uint32_t)n};   // leaf_off[0..1], leaf_gsize[0]
        s = upload(ix->bf_leaf, leaf, sizeof(leaf));
        if (s != SCANN_HIP_OK) {
            scann_hip_index_destroy(ix);
            return s;
        }
        TxhIndexDev &t = ix->bfx;
        t.dim = dim; t.stride = stride; t.L = 1; t.S = 0; t.K = 16; t.dsub = 0; t.nw = 0; t.code_bits = 4; t.kp = 16;
        t.n_local = n; t.centers = nullptr; t.leaf_off = ix->bf_leaf.as<uint32_t>();
        t.leaf_gsize = ix->bf_leaf.as<uint32_t>() + 2; t.leaf_ids = nullptr; t.codes = nullptr;
        t.rows = ix->bf_rows.as<float>(); t.rows8 = nullptr; t.rows8_meta = nullptr; t.rows_csr = 1;
        t.codebook = nullptr; t.use_residuals = 0; t.ah_mode = 1; t.measure = measure; t.exact_scan = 1;
    }
    if ((stride & 3u) == 0) {   // bf16 copy + norms for the shortlist path (big indexes only)
        s = bf_build_shortlist_data(ix->bf, ix->bf_rows_b, ix->bf_rows_bl, ix->bf_norm2, &ix->bf.max_norm,
                                    ix->stream);
        if (s != SCANN_HIP_OK) {
            scann_hip_index_destroy(ix);
            return s;
        }
        if (ix->bf_rows_b.p) {
            ix->bf.rows_b = ix->bf_rows_b.as<uint16_t>();
            ix->bf.rows_bl = ix->bf_rows_bl.as<uint16_t>();
            ix->bf.norm2 = ix->bf_norm2.as<float>();
        }
    }
    *out = ix;
    return SCANN_HIP_OK;
}

// =====================================================================================
// Tree-X-Hybrid / AsymmetricHasher index
// =====================================================================================
}  // extern "C"

// content_status: what inconsistent array CONTENTS are reported as (InvalidArgument for caller-built
// descriptors, DataLoss for index files, whose section sizes were already checked)
int scann::txh_create_checked(scann_hip_ctx *ctx, const scann_hip_txh_desc *d, scann_hip_index **out,
                              int content_status) {
    if (!ctx || !d || !out) return fail(SCANN_HIP_INVALID_ARGUMENT, "null ctx/desc/out_index");
    const bool ah = d->num_partitions == 0;
    if (d->n_local == 0)  // tree_x_hybrid/mod.rs:132-134, hashes/hasher.rs:110-112
        return fail(SCANN_HIP_INVALID_ARGUMENT, "Cannot build from empty dataset");
    if (d->n_local >= 0xFFFFFFFFull) return fail(SCANN_HIP_OUT_OF_RANGE, "DatapointIndex is u32");
    // SearchMode::Partitioned (scann.rs:213-252): no codebook, the selected leaves are scored exactly
    const bool exact = !d->codebook && !d->codes && d->num_subspaces == 0;
    if (!exact && (!d->codebook || !d->codes)) return fail(SCANN_HIP_INVALID_ARGUMENT, "codebook/codes null");
    if (d->distance_measure < SCANN_HIP_SQUARED_L2 || d->distance_measure > SCANN_HIP_COSINE)
        return fail(SCANN_HIP_UNIMPLEMENTED, "distance_measure must be SquaredL2, L2, DotProduct, L1 or Cosine");
    const uint32_t S = d->num_subspaces, K = exact ? 16u : d->num_codes, dsub = d->dims_per_subspace;
    if (d->dim == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "dim is 0");
    if (exact) {
        if (ah || !d->data || d->leaf_sizes_global || d->n_rows != d->n_local)
            return fail(SCANN_HIP_INVALID_ARGUMENT,
                        "the exact leaf scan needs centers, the dataset rows and an unsharded index");
    } else if (S == 0 || d->dim % S != 0 || dsub != d->dim / S) {  // codebook.rs:154-159
        return fail(SCANN_HIP_INVALID_ARGUMENT,
                    "Dimensionality " + std::to_string(d->dim) +
                        " must be divisible by num_subspaces " + std::to_string(S));
    }
    if (K == 0 || K > 256)
        return fail(SCANN_HIP_UNIMPLEMENTED, "num_codes must be 1..256");
    // K <= 16: 4-bit packed codes + 16-slot tables (LUT16); else bytes + 256-slot tables
    const uint32_t bits = K <= 16 ? 4u : 8u;
    if (!exact && bits == 4 && (S % 8 != 0 || S > 64 || S == 40 || S == 56))
        return fail(SCANN_HIP_UNIMPLEMENTED, "num_subspaces must be 8,16,24,32,48 or 64 for num_codes <= 16");
    if (bits == 8 && S != 4 && S != 8 && S != 16)
        return fail(SCANN_HIP_UNIMPLEMENTED, "num_subspaces must be 4, 8 or 16 for 16 < num_codes <= 256");
    if (bits == 8 && d->codes_packed4)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "codes_packed4 needs num_codes <= 16");
    if (!ah) {
        if (!d->centers || !d->leaf_offsets || !d->leaf_ids)
            return fail(SCANN_HIP_INVALID_ARGUMENT, "centers/leaf_offsets/leaf_ids null");
        if (d->num_partitions > kMaxLeavesSelect)
            return fail(SCANN_HIP_UNIMPLEMENTED, "num_partitions > 16384");
        if (d->leaf_offsets[0] != 0 || d->leaf_offsets[d->num_partitions] != d->n_local)
            return fail(content_status, "leaf_offsets must span [0, n_local]");
        for (uint32_t l = 0; l < d->num_partitions; ++l) {
            if (d->leaf_offsets[l + 1] < d->leaf_offsets[l])
                return fail(content_status, "leaf_offsets not monotone");
            // merge keys and their decoding assume a leaf's local rows are a prefix of the global leaf
            if (d->leaf_sizes_global && d->leaf_sizes_global[l] < d->leaf_offsets[l + 1] - d->leaf_offsets[l])
                return fail(content_status, "leaf_sizes_global smaller than the local leaf");
        }
        // the re-rank and the exact leaf scan read data + leaf_ids[row] * stride
        if (d->data && !d->data_is_csr_order)
            for (uint64_t i = 0; i < d->n_local; ++i)
                if (d->leaf_ids[i] >= d->n_rows)
                    return fail(content_status, "leaf_ids entry " + std::to_string(i) + " >= n_rows");
    }
    if (d->data && d->stride < d->dim) return fail(SCANN_HIP_INVALID_ARGUMENT, "stride < dim");
    if (d->data && d->data_is_csr_order && d->n_rows != d->n_local)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "CSR-ordered data must have n_local rows");
    // AsymmetricHasher mode reads row i of data for CSR row i
    if (d->data && ah && d->n_rows < d->n_local)
        return fail(content_status, "data has fewer rows than the index has points");
    SCANN_TRY(set_device(ctx));

    auto *ix = new scann_hip_index();
    ix->ctx = ctx;
    ix->kind = KIND_TXH;
    auto bail = [&](int s) {
        scann_hip_index_destroy(ix);
        return s;
    };
    int s = index_common_init(ix);
    if (s != SCANN_HIP_OK) return bail(s);

    const uint32_t L = ah ? 1u : d->num_partitions;
    const uint32_t nw = bits == 4 ? S / 8 : S / 4;
    const uint64_t n = d->n_local;

    // leaf tables
    std::vector<uint32_t> off(L + 1), gsz(L);
    if (ah) {
        off[0] = 0;
        off[1] = (uint32_t)n;
        gsz[0] = (uint32_t)n;
    } else {
        std::memcpy(off.data(), d->leaf_offsets, (size_t)(L + 1) * 4);
        for (uint32_t l = 0; l < L; ++l)
            gsz[l] = d->leaf_sizes_global ? d->leaf_sizes_global[l] : off[l + 1] - off[l];
    }
    ix->local_sizes_desc.resize(L);
    for (uint32_t l = 0; l < L; ++l) ix->local_sizes_desc[l] = off[l + 1] - off[l];
    std::sort(ix->local_sizes_desc.begin(), ix->local_sizes_desc.end(), std::greater<uint32_t>());

    // packed codes: [n][nw] words, nibble nb of word wi = subspace 8*wi+nb
    // (PackedCodes4Bit layout, hashes/lut16.rs:43-61, read little-endian)
    std::vector<uint32_t> words;
    const uint32_t *code_words = nullptr;
    const size_t bpp = S / 2;
    if (exact) {
        // no codes
    } else if (d->codes_packed4) {
        if (K < 16) {   // every nibble must address a trained centre (K == 16: all values are valid)
            const size_t nbytes = (size_t)n * bpp;
            for (size_t i = 0; i < nbytes; ++i)
                if ((d->codes[i] & 15u) >= K || (d->codes[i] >> 4) >= K)
                    return bail(fail(content_status, "code value >= num_codes"));
        }
        if ((reinterpret_cast<uintptr_t>(d->codes) & 3u) == 0) {
            code_words = reinterpret_cast<const uint32_t *>(d->codes);
        } else {
            words.resize((size_t)n * nw);
            std::memcpy(words.data(), d->codes, (size_t)n * bpp);
            code_words = words.data();
        }
    } else {
        words.assign((size_t)n * nw, 0u);
        for (uint64_t i = 0; i < n; ++i) {
            const uint8_t *c = d->codes + i * S;
            uint32_t *w = words.data() + i * nw;
            for (uint32_t sidx = 0; sidx < S; ++sidx) {
                if (c[sidx] >= K) {
                    return bail(fail(content_status, "code value >= num_codes"));
                }
                if (bits == 4) w[sidx >> 3] |= (uint32_t)(c[sidx] & 0x0F) << (4 * (sidx & 7u));
                else w[sidx >> 2] |= (uint32_t)c[sidx] << (8 * (sidx & 3u));
            }
        }
        code_words = words.data();
    }

    if ((s = upload(ix->d_leaf_off, off.data(), (size_t)(L + 1) * 4)) != SCANN_HIP_OK) return bail(s);
    if ((s = upload(ix->d_leaf_gsize, gsz.data(), (size_t)L * 4)) != SCANN_HIP_OK) return bail(s);
    if (!exact) {
        if ((s = upload(ix->d_codes, code_words, (size_t)n * nw * 4)) != SCANN_HIP_OK) return bail(s);
        if ((s = upload(ix->d_codebook, d->codebook, (size_t)S * K * dsub * 4)) != SCANN_HIP_OK) return bail(s);
    }
    if (!ah) {
        if ((s = upload(ix->d_centers, d->centers, (size_t)L * d->dim * 4)) != SCANN_HIP_OK) return bail(s);
        if ((s = upload(ix->d_leaf_ids, d->leaf_ids, (size_t)n * 4)) != SCANN_HIP_OK) return bail(s);
    }
    if (d->data) {
        if ((s = upload(ix->d_rows, d->data, (size_t)d->n_rows * d->stride * 4)) != SCANN_HIP_OK)
            return bail(s);
    }

    TxhIndexDev &t = ix->tx;
    t.dim = d->dim;
    t.stride = d->stride;
    t.L = L;
    t.S = S;
    t.K = K;
    t.dsub = dsub;
    t.nw = nw;
    t.code_bits = bits;
    t.kp = bits == 4 ? 16u : 256u;
    t.n_local = n;
    t.centers = ah ? nullptr : ix->d_centers.as<float>();
    t.centers_t = nullptr;
    t.centers_pitch = 0;
    if (!ah && L <= 4096) {   // transposed centroids: the small-batch leaf selection reads them coalesced (txh.hip)
        const uint32_t pitch = (L + 63u) & ~63u;
        if ((s = ix->d_centers_t.ensure((size_t)pitch * d->dim * 4)) != SCANN_HIP_OK) return bail(s);
        if ((s = launch_transpose_centers(ix->d_centers.as<float>(), L, d->dim, pitch, ix->d_centers_t.as<float>(),
                                          ix->stream)) != SCANN_HIP_OK)
            return bail(s);
        if (hipStreamSynchronize(ix->stream) != hipSuccess) return bail(fail(SCANN_HIP_INTERNAL, "centroid transpose failed"));
        t.centers_t = ix->d_centers_t.as<float>();
        t.centers_pitch = pitch;
    }
    t.leaf_off = ix->d_leaf_off.as<uint32_t>();
    t.leaf_gsize = ix->d_leaf_gsize.as<uint32_t>();
    t.leaf_ids = ah ? nullptr : ix->d_leaf_ids.as<uint32_t>();
    t.codes = exact ? nullptr : ix->d_codes.as<uint32_t>();
    t.codes_sp = nullptr;
    // operand planes of the sparse-MFMA prefilter (txh.hip K6e): a second copy of the 4-bit codes, S/2 .. 2 S bytes
    // per point.  SCANN_HIP_SMFMAC=0: not built, the dense integer-MFMA prefilter is used instead.
    {
        const char *e = std::getenv("SCANN_HIP_SMFMAC");
        if (!exact && bits == 4 && !(e && std::atoi(e) == 0)) {
            if ((s = ix->d_codes_sp.ensure((size_t)n * sp_words(S) * 4)) != SCANN_HIP_OK) return bail(s);
            if ((s = launch_codes_sp_build(ix->d_codes.as<uint32_t>(), n, S, ix->d_codes_sp.as<uint32_t>(), ix->stream)) != SCANN_HIP_OK)
                return bail(s);
            if (hipStreamSynchronize(ix->stream) != hipSuccess) return bail(fail(SCANN_HIP_INTERNAL, "code plane build failed"));
            t.codes_sp = ix->d_codes_sp.as<uint32_t>();
        }
    }
    t.measure = d->distance_measure;
    t.exact_scan = exact ? 1 : 0;
    t.rows = d->data ? ix->d_rows.as<float>() : nullptr;
    t.rows8 = nullptr;
    t.rows8_meta = nullptr;
    // 8-bit copy of the rows for the re-rank filter (txh.hip K8b): +25 % of the row bytes.  Large indexes
    // only (the filter pays for long candidate lists); SCANN_HIP_RERANK_I8 = 0 never, 2 always.
    // SCANN_HIP_RERANK_STORE = int8 (default: per-row scaled integers, the tighter filter) or fp8 (the
    // reference's E4M3 codec with a per-row calibrate_scale, quantization/fp8.rs).
    t.rows8_fmt = 0;
    t.rows8_uniform = 0;
    t.rows8_scale = t.rows8_emax = 0.0f;
    {
        int mode = 1;
        if (const char *e = std::getenv("SCANN_HIP_RERANK_I8")) mode = std::atoi(e);
        const char *store = std::getenv("SCANN_HIP_RERANK_STORE");
        const bool fp8 = store && std::strcmp(store, "fp8") == 0;
        const bool want = d->data && !exact && d->distance_measure == SCANN_HIP_SQUARED_L2 && (d->dim & 15u) == 0 &&
                          (mode == 2 || (mode == 1 && d->n_rows >= 65536));
        if (want) {
            if ((s = ix->d_rows8.ensure((size_t)d->n_rows * d->dim)) != SCANN_HIP_OK) return bail(s);
            if ((s = ix->d_rows8_meta.ensure((size_t)d->n_rows * 8)) != SCANN_HIP_OK) return bail(s);
            if (fp8) {
                DevBuf mism;
                if ((s = mism.ensure(4)) != SCANN_HIP_OK) return bail(s);
                if (hipMemsetAsync(mism.p, 0, 4, ix->stream) != hipSuccess) return bail(fail(SCANN_HIP_INTERNAL, "memset failed"));
                if ((s = launch_rows_fp8_build(ix->d_rows.as<float>(), d->n_rows, d->dim, d->stride,
                                               ix->d_rows8.as<uint8_t>(), ix->d_rows8_meta.p, mism.as<uint32_t>(),
                                               ix->stream)) != SCANN_HIP_OK)
                    return bail(s);
                uint32_t bad = 0;
                if (hipMemcpyAsync(&bad, mism.p, 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
                    hipStreamSynchronize(ix->stream) != hipSuccess)
                    return bail(fail(SCANN_HIP_INTERNAL, "fp8 row build failed"));
                if (bad) return bail(fail(SCANN_HIP_INTERNAL, "v_cvt_f32_fp8 decodes the reference's E4M3 codes differently"));
                t.rows8_fmt = 1;
            } else {
                if ((s = launch_rows_i8_build(ix->d_rows.as<float>(), d->n_rows, d->dim, d->stride, ix->d_rows8.as<int8_t>(),
                                              ix->d_rows8_meta.p, ix->stream)) != SCANN_HIP_OK)
                    return bail(s);
                if (hipStreamSynchronize(ix->stream) != hipSuccess) return bail(fail(SCANN_HIP_INTERNAL, "int8 row build failed"));
                // Rows of EQUAL magnitude (largest per-row scale within 2 % of the mean scale, every row finite): ONE
                // scale and ONE error bound (the largest ||x - x~||) for all rows, so that the filter pass gathers a
                // candidate's 128-byte row and nothing else -- the per-row {scale, error} pair is a second random
                // cache line per candidate, half of the pass's memory requests (C3, uniform data: 163 -> 128 us).  The
                // bound is strict because a coarser common scale widens every bracket: on the clustered 10M x 128 set
                // (per-row maxima 25 % apart, candidates nearly equidistant) the shortlists outgrew the fast path and
                // the step went from 1.43 to 1.86 ms.  SCANN_HIP_RERANK_UNIFORM=0: always per-row, 2: always one scale.
                const char *ue = std::getenv("SCANN_HIP_RERANK_UNIFORM");
                const double spread = (ue && std::atoi(ue) == 2) ? 1e30 : 1.02;
                if (!(ue && std::atoi(ue) == 0)) {
                    std::vector<float> meta((size_t)d->n_rows * 2);
                    if (hipMemcpy(meta.data(), ix->d_rows8_meta.p, meta.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
                        return bail(fail(SCANN_HIP_INTERNAL, "int8 row meta copy failed"));
                    double sum = 0.0;
                    float smax = 0.0f;
                    bool finite = true;
                    for (uint64_t r = 0; r < d->n_rows; ++r) {
                        const float sc = meta[2 * r], E = meta[2 * r + 1];
                        finite = finite && E < INFINITY && sc < INFINITY;
                        smax = std::max(smax, sc);
                        sum += sc;
                    }
                    if (finite && d->n_rows > 0 && (double)smax <= spread * (sum / (double)d->n_rows)) {
                        if ((s = launch_rows_i8_build(ix->d_rows.as<float>(), d->n_rows, d->dim, d->stride, ix->d_rows8.as<int8_t>(),
                                                      ix->d_rows8_meta.p, ix->stream, smax)) != SCANN_HIP_OK)
                            return bail(s);
                        if (hipMemcpy(meta.data(), ix->d_rows8_meta.p, meta.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
                            return bail(fail(SCANN_HIP_INTERNAL, "int8 row meta copy failed"));
                        float emax = 0.0f;
                        for (uint64_t r = 0; r < d->n_rows; ++r) emax = std::max(emax, meta[2 * r + 1]);
                        if (emax < INFINITY) {
                            t.rows8_uniform = 1;
                            t.rows8_scale = smax;
                            t.rows8_emax = emax;
                        }
                    }
                }
            }
            t.rows8 = ix->d_rows8.as<int8_t>();
            t.rows8_meta = ix->d_rows8_meta.p;
        }
    }
    // AH mode: CSR row == datapoint index.  Sharded: rows arrive in CSR order.
    t.rows_csr = (ah || d->data_is_csr_order) ? 1 : 0;
    t.codebook = exact ? nullptr : ix->d_codebook.as<float>();
    t.use_residuals = (!ah && d->use_residuals) ? 1 : 0;
    t.ah_mode = ah ? 1 : 0;
    ix->sharded = d->leaf_sizes_global != nullptr;
    ix->default_P = ah ? 1u : std::max(1u, d->partitions_to_search);
    ix->multiplier = d->pre_reorder_multiplier;
    *out = ix;
    return SCANN_HIP_OK;
}

extern "C" {

int scann_hip_txh_create(scann_hip_ctx *ctx, const scann_hip_txh_desc *d, scann_hip_index **out) {
    return scann::txh_create_checked(ctx, d, out, SCANN_HIP_INVALID_ARGUMENT);
}

// SCANN_HIP_SMALL=0 turns the small-batch pipeline off (read per call: tests flip it)
static bool small_batch_enabled() {
    const char *e = std::getenv("SCANN_HIP_SMALL");
    return !(e && std::atoi(e) == 0);
}

// ---- per-call parameter resolution --------------------------------------------------
struct TxhCallParams {
    uint32_t P, m, k, cap, st, scap;
    int exact_reorder;
    int no_threshold;
    int small;   // small-batch pipeline (txh.hip "Small batches"); 2 = the wide one ("Few queries, long streams")
    uint32_t wide_cap2;
};

static uint64_t max_stream(const scann_hip_index *ix, uint32_t P) {
    uint64_t s = 0;
    for (uint32_t i = 0; i < P && i < ix->local_sizes_desc.size(); ++i) s += ix->local_sizes_desc[i];
    return s;
}

static int resolve_params(const scann_hip_index *ix, uint32_t k, const scann_hip_search_opts *o,
                          bool full_cap, TxhCallParams *out, uint32_t nq = 0xFFFFFFFFu, bool allow_wide = true) {
    scann_hip_search_opts def;
    scann_hip_search_opts_default(&def);
    if (!o) o = &def;
    uint32_t P = o->partitions_to_search ? o->partitions_to_search : ix->default_P;
    P = std::min(P, ix->tx.L);  // tree_partitioner.rs:214
    if (ix->tx.ah_mode) P = 1;
    if (P > kMaxPartitionsToSearch)
        return fail(SCANN_HIP_UNIMPLEMENTED, "partitions_to_search > 4096");
    const bool exact_reorder = o->exact_reorder && !ix->tx.exact_scan;   // the scan's distances ARE exact
    if (ix->tx.exact_scan) {
        full_cap = true;   // dense key lists: one slot per scanned row, no threshold
        if (o->allow_bitmap) return fail(SCANN_HIP_UNIMPLEMENTED, "the exact leaf scan takes no filter");
    }
    uint32_t m;
    if (!exact_reorder) {
        m = k;
    } else if (o->pre_reorder_k) {
        m = o->pre_reorder_k;
    } else {
        const float mf = (float)k * ix->multiplier;  // mod.rs:263 (saturating truncation)
        m = mf > 0.0f ? (mf >= 4294967295.0f ? 0xFFFFFFFFu : (uint32_t)mf) : 0u;
    }
    if (m > kMaxPreReorderK)
        return fail(SCANN_HIP_UNIMPLEMENTED,
                    "pre-reorder candidate count " + std::to_string(m) + " exceeds " +
                        std::to_string(kMaxPreReorderK));
    if (exact_reorder && !ix->tx.rows)  // hasher.rs:194-197
        return fail(SCANN_HIP_FAILED_PRECONDITION, "Dataset not stored");
    const uint64_t ms = std::min<uint64_t>(std::max<uint64_t>(1, max_stream(ix, P)), 0xFFFFFFFFull);
    uint32_t st, scap;
    sample_plan(ms, P, &st, &scap);
    // A handful of queries over a short stream: three launches with dense candidate lists instead of
    // the batched pipeline (SCANN_HIP_SMALL=0 disables).  Unsharded indexes only: stream positions are
    // list slots, and a shard's local leaves leave holes in them.
    const bool small_on = small_batch_enabled();
    out->small = (small_on && nq <= kSmallBatch && !ix->sharded && m <= kSmallMaxCandidates &&
                  k <= 64 && ms <= kSmallMaxStream && ix->tx.L <= 4096) ? 1 : 0;
    // Fewer queries still, over a long stream: the wide pipeline (three launches, every stage spread over the chip).
    // Its scan rebuilds the tables per leaf inside a workgroup of 1024 stream positions: leaves of >= 512 points
    // (or one leaf); ADC scans only (Partitioned mode, 4 queries over 20 leaves of 1M x 128: 0.097 ms on the small-batch
    // pipeline, 0.147 through this one).  Streams of >= 8 m points: below that most of the stream is candidates anyway.
    // SCANN_HIP_WIDE: 0 = never, 2 = whenever the limits allow (tests), else from kWideMinStream points.
    out->wide_cap2 = 0;
    {
        int mode = 1;
        if (const char *e = std::getenv("SCANN_HIP_WIDE")) mode = std::atoi(e);
        const bool limits = small_on && allow_wide && mode != 0 && nq <= kWideBatch && !ix->sharded && m >= 1 &&
                            m <= kWideMaxCandidates && k <= 64 && ms <= kWideMaxStream && ix->tx.L <= 4096 && P <= 512 &&
                            !ix->tx.exact_scan;
        const bool long_leaves = P == 1 || ix->tx.n_local / std::max(1u, ix->tx.L) >= 512;
        const bool pays = ms >= kWideMinStream && ms >= 8ull * m && long_leaves;
        if (limits && (mode == 2 || pays)) {
            out->small = 2;
            out->wide_cap2 = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(ms, 3ull * m + 1024), 16384);
        }
    }
    if (out->small) full_cap = true;
    uint64_t cap = ms;
    if (!full_cap) {
        // upper bound of the survivors of the sampled threshold (rank j of a stride-st sample)
        const uint32_t j = sample_rank(m, st);
        cap = (uint64_t)((double)j + 8.0 * std::sqrt((double)j) + 16.0) * st + 256;
        cap = std::min(std::max<uint64_t>(cap, m), ms);
    }
    out->st = st;
    out->scap = scap;
    out->P = P;
    out->m = m;
    out->k = k;
    out->cap = (uint32_t)std::min<uint64_t>(cap, 0xFFFFFFFFull);
    out->exact_reorder = exact_reorder ? 1 : 0;
    out->no_threshold = full_cap ? 1 : 0;
    return SCANN_HIP_OK;
}

}  // extern "C"

int scann::txh_resolve_m(scann_hip_index *ix, uint32_t k, const scann_hip_search_opts *opts, uint32_t *out_m) {
    if (!ix || ix->kind != KIND_TXH) return fail(SCANN_HIP_INVALID_ARGUMENT, "not a tree index");
    scann_hip_search_opts o;
    if (opts) o = *opts; else scann_hip_search_opts_default(&o);
    o.exact_reorder = 1;
    TxhCallParams p;
    SCANN_TRY(resolve_params(ix, k, &o, false, &p));
    *out_m = p.m;
    return SCANN_HIP_OK;
}

extern "C" {

// allow_bytes: size of an allow-bitmap the caller will copy into s.allow (0 = none)
static int ensure_txh_workspace(scann_hip_index *ix, TxhWorkspace &s, uint32_t nq, const TxhCallParams &p,
                                bool own_queries, uint32_t q_stride, bool own_outputs,
                                TxhWork *w, size_t allow_bytes = 0) {
    const TxhIndexDev &t = ix->tx;
    const uint32_t L = t.L, P = p.P, m = std::max(1u, p.m), k = std::max(1u, p.k);
    const uint64_t max_slots64 = (uint64_t)nq * P + 3ull * L + 4;
    if (max_slots64 > 0xFFFFFFF0ull)
        return fail(SCANN_HIP_RESOURCE_EXHAUSTED, "batch x partitions_to_search exceeds the pair table (2^32 slots)");
    const uint32_t max_slots = (uint32_t)max_slots64;
    const uint32_t max_quads = max_slots / 4 + 1;
    // (per-candidate arrays sized for THIS call's pre_reorder_k; the arena re-places everything when one grows)
    const uint32_t m_al = m;
    const uint64_t cap_al = p.cap;
    if (allow_bytes) s.want(s.allow, allow_bytes);
    if (own_queries) s.want(s.queries, (size_t)nq * q_stride * 4);
    if (!t.ah_mode) s.want(s.cdist, (size_t)nq * L * 4);
    s.want(s.tokens, (size_t)nq * P * 4);
    s.want(s.token_dists, (size_t)nq * P * 4);
    s.want(s.vbase, (size_t)nq * (P + 1) * 4);
    s.want(s.sbase, (size_t)nq * (P + 2) * 4);
    s.want(s.pair_sbase, (size_t)max_slots * 4);
    s.want(s.stile_off, (size_t)(L + 1) * 4);
    s.want(s.samp, (size_t)nq * p.scap * 4);
    s.want(s.leaf_cnt, (size_t)L * 4);
    s.want(s.leaf_cursor, (size_t)L * 4);
    s.want(s.pair_off, (size_t)(L + 1) * 4);
    s.want(s.tile_off, (size_t)(L + 1) * 4);
    s.want(s.counters, CNT_WORDS * 4);
    s.want(s.pair_q, (size_t)max_slots * 4);
    s.want(s.pair_leaf, (size_t)max_slots * 4);
    s.want(s.pair_vbase, (size_t)max_slots * 4);
    s.want(s.pair_thr, (size_t)max_slots * 8);
    s.want(s.slot_of, (size_t)nq * P * 4);
    s.want(s.lutq, (size_t)max_quads * t.S * t.kp * 4 * 4);
    s.want(s.thr, (size_t)nq * 8);
    s.want(s.cand_cnt, (size_t)nq * 4);
    s.want(s.cand, (size_t)nq * cap_al * 8);
    s.want(s.cand_key, (size_t)nq * m_al * 8);
    s.want(s.cand_idx, (size_t)nq * m_al * 4);
    s.want(s.cand_dist, (size_t)nq * m_al * 4);
    s.want(s.cand_exact, (size_t)nq * m_al * 4);
    s.want(s.cand_row, (size_t)nq * m_al * 4);
    s.want(s.cand_count, (size_t)nq * 4);
    if (own_outputs) {
        s.want(s.out_idx, (size_t)nq * k * 4);
        s.want(s.out_dist, (size_t)nq * k * 4);
        s.want(s.out_count, (size_t)nq * 4);
    }
    w->nq = nq;
    w->q_stride = q_stride;
    w->P = P;
    w->m = p.m;
    w->k = p.k;
    w->cap = p.cap;
    w->exact_reorder = p.exact_reorder;
    w->no_threshold = p.no_threshold;
    w->small = (uint32_t)p.small;
    w->small_done = nullptr;
    w->small_seq = 0;
    w->small_tickets = nullptr;
    w->small_max_leaf = ix->local_sizes_desc.empty() ? 0u : ix->local_sizes_desc[0];
    w->need_sorted_cands = 0;
    w->allow = nullptr;
    w->allow_bits = 0;
    w->st = p.st;
    w->scap = p.scap;
    {   // sample tiles: enough of them to fill the chip (the sample pass is 1/st of the scan)
        const uint64_t quads = ((uint64_t)nq * P + 3) / 4;
        const uint32_t tp = scan_tile_points(t);
        const uint64_t chunks = std::max<uint64_t>(1, ((uint64_t)p.scap + tp - 1) / tp);
        uint32_t qpt = kScanQuadsPerTile;
        while (qpt > 2 && chunks * ((quads + qpt - 1) / qpt) < 4096) qpt >>= 1;
        if (const char *e = std::getenv("SCANN_HIP_SQPT")) qpt = (uint32_t)std::max(1, std::atoi(e));
        w->sqpt = qpt;
        // scan tiles: the largest quad group that still yields ~4 tiles per resident workgroup
        // (small shards and small batches would otherwise leave most of the chip idle)
        const uint64_t all_chunks = t.n_local / tp + L;
        const uint64_t quads_per_leaf = std::max<uint64_t>(1, quads / std::max(1u, L));
        uint32_t sq = kScanQuadsPerTile;
        while (sq > 8 && all_chunks * ((quads_per_leaf + sq - 1) / sq) < 6144) sq >>= 1;
        w->qpt = sq;
        if (const char *e = std::getenv("SCANN_HIP_QPT")) w->qpt = (uint32_t)std::max(1, std::atoi(e));
        // long leaves scanned by many queries (AsymmetricHasher mode, few big partitions): tables
        // resident in LDS over a range of chunks (adc_scan_res_kernel); needs 4-bit codes with
        // S <= 32.  With few quads per leaf (typical Tree-X-Hybrid batches: measured at 10M / 1000
        // leaves) the per-chunk kernel is as fast or faster.
        const uint64_t rtp = (uint64_t)kResThreads * kScanPPT;
        const uint64_t res_chunks_per_leaf = std::max<uint64_t>(1, (t.n_local / std::max(1u, L)) / rtp);
        const uint64_t qgroups = std::max<uint64_t>(1, (quads_per_leaf + kResQuads - 1) / kResQuads);
        const uint64_t units = (t.n_local / rtp + L) * qgroups;       // (chunk, quad group) pairs
        // a tile must cover >= 8 chunks to amortise its table load, and there must be >= 2 tiles
        // per resident workgroup (small shards: measured at 125k-500k rows)
        w->res_cl = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(8, units / 4096));
        w->resident = (t.code_bits == 4 && t.S <= 32 && res_chunks_per_leaf >= 8 && quads_per_leaf >= 32 &&
                       units / w->res_cl >= 1536) ? 1u : 0u;
        // SCANN_HIP_RESIDENT: 0 = never, 2 = whenever the code layout allows (tests), else the heuristic
        if (const char *e = std::getenv("SCANN_HIP_RESIDENT")) {
            const int v = std::atoi(e);
            if (v == 0) w->resident = 0u;
            if (v == 2) w->resident = (t.code_bits == 4 && t.S <= 32) ? 1u : 0u;
        }
        if (const char *e = std::getenv("SCANN_HIP_RES_CL")) w->res_cl = (uint32_t)std::max(1, std::atoi(e));
        // Integer-MFMA prefilter + exact refine (txh.hip K6d): 4-bit codes, a filter bound to prove
        // against, and enough pairs per leaf to fill 32-column MFMA tiles (a leaf scanned by few
        // queries would leave most columns empty; the LDS-gather kernels take those).
        // Leaves scanned by 8-24 queries (2-5 quads: typical Tree-X-Hybrid batches) take the 16-column form
        // (adc_mfma16_kernel); fewer than that and the LDS-gather kernel wins.
        const bool mfma_ok = t.code_bits == 4 && !p.no_threshold && !t.exact_scan;
        // The prefilter pays while the wanted candidates are a small share of the scanned stream: every
        // survivor (~1.5-2.4 m) is staged, flushed and recomputed by the refine.  Measured: 1M x 128 flat,
        // m = 5000 (0.5 %): 2.7x faster than the gather scan; 10M x 128 / 1000 leaves, m / stream 0.1-0.6 %:
        // steps 1.1-1.4x faster; 1-3 % (P = 10 or 25 with m = 8192): 1.1-1.2x slower.
        const uint64_t stream = std::max<uint64_t>(1, max_stream(ix, P));
        const bool sparse = (uint64_t)p.m * 128 <= stream;
        w->mfma = !(mfma_ok && sparse) ? 0u : quads_per_leaf >= 6 ? 1u : quads_per_leaf >= 2 ? 2u : 0u;
        // SCANN_HIP_MFMA: 0 = never, 2 / 3 = the 32- / 16-column form whenever the code layout allows (tests),
        // else the heuristic
        if (const char *e = std::getenv("SCANN_HIP_MFMA")) {
            const int v = std::atoi(e);
            if (v == 0) w->mfma = 0u;
            if (v == 2) w->mfma = mfma_ok ? 1u : 0u;
            if (v == 3) w->mfma = mfma_ok ? 2u : 0u;
        }
        // With the operand planes in the index the 32-pair form runs on the 2:4-sparse MFMA (SCANN_HIP_SMFMAC=0 at this
        // call keeps the dense instruction: tests compare the two) -- and takes over from 8 pairs per leaf, whatever
        // m / stream is: at 10M x 128, 1000 leaves, 1024 queries, P = 10 (10 pairs per leaf: tiles a third full)
        // m = 1000 / 4000 / 8192 run 0.68 / 0.95 / 1.29 ms per step on it against 0.88 / 1.09 / 1.61 ms on the 16-pair
        // dense form / the f32 gather scan the rules above pick (P = 25 / 50 / 100 at m = 1000: 0.94 / 1.11 / 1.44 ms
        // against round 2's 1.12 / 1.41 / 1.91).
        // (long leaves only -- an item is up to 2048 points of a leaf, and its fixed costs, the table fragments and the
        // flush, want most of that: at 1M x 128 / 1000 leaves of 1000 points the gather scan's 0.08 ms beats 0.13 ms)
        if (mfma_ok && t.codes_sp && quads_per_leaf >= 2 && t.n_local / std::max(1u, L) >= 4096) {
            const char *e = std::getenv("SCANN_HIP_SMFMAC");
            const char *f = std::getenv("SCANN_HIP_MFMA");
            if (!(e && std::atoi(e) == 0) && !(f && (std::atoi(f) == 0 || std::atoi(f) == 3))) w->mfma = 3u;
        }
        if (w->mfma == 1 && t.codes_sp) {   // (forced 32-pair form, fewer pairs per leaf)
            const char *e = std::getenv("SCANN_HIP_SMFMAC");
            if (!(e && std::atoi(e) == 0)) w->mfma = 3u;
        }
        if (w->mfma) w->resident = 0u;
    }
    // int8 re-rank filter: lists of a few hundred candidates and more (SCANN_HIP_RERANK_I8_MIN overrides)
    {
        uint32_t min_m = 512;
        if (const char *e = std::getenv("SCANN_HIP_RERANK_I8_MIN")) min_m = (uint32_t)std::max(1, std::atoi(e));
        w->use_i8 = (t.rows8 && p.exact_reorder && p.m >= min_m && p.m > 4 * p.k) ? 1u : 0u;
        w->rr_lb = w->rr_ub = nullptr;
        if (w->use_i8) {
            s.want(s.rr_lb, (size_t)nq * m_al * 4);
            s.want(s.rr_ub, (size_t)nq * m_al * 4);
        }
    }
    w->lut8 = nullptr;
    w->lut8_meta = nullptr;
    w->mfma_thr1 = nullptr;
    w->cand32 = nullptr;
    w->cand32_codes = nullptr;
    w->cand32_cnt = nullptr;
    w->cap32 = 0;
    if (w->mfma) {
        // survivors of the integer bound: the f32 filter's (<= cap) plus the quantisation margin
        const uint64_t ms2 = std::min<uint64_t>(std::max<uint64_t>(1, max_stream(ix, P)), 0xFFFFFFFFull);
        // (the margin's share is independent of m -- the points within 17 table steps above the bound -- and in
        // the dense nearest leaves of a tree index it can be several thousand: a generous floor)
        const uint64_t cap32 = std::min<uint64_t>(ms2, (uint64_t)p.cap * 4 + 16384);
        s.want(s.lut8, (size_t)max_slots * t.S * 16 + 64);
        s.want(s.lut8_meta, (size_t)(max_slots + 4) * 16);
        s.want(s.mfma_thr1, (size_t)(max_slots + 4) * 4);
        const uint64_t cap32_al = std::min<uint64_t>(ms2, cap_al * 4 + 16384);   // (allocation; the stride stays cap32)
        s.want(s.cand32, (size_t)nq * cap32_al * 4);
        // flat hashers: the survivors' packed codes (dense prefilter) or plane rows (sparse prefilter) travel with them
        if (t.ah_mode) s.want(s.cand32_codes, (size_t)nq * cap32_al * std::max(t.S / 8, w->mfma == 3 ? sp_words(t.S) : 0u) * 4);
        s.want(s.cand32_cnt, (size_t)nq * 4);
        w->cap32 = (uint32_t)cap32;
    }
    if (p.small) s.want(s.small_tickets, 64 * 4);
    if (p.small == 2) {
        s.want(s.wide_min, (size_t)nq * p.cap * 4);
        s.want(s.wide_ckey, (size_t)nq * p.wide_cap2 * 8);
        s.want(s.wide_ceb, (size_t)nq * p.wide_cap2 * 4);
        s.want(s.wide_cidx, (size_t)nq * p.wide_cap2 * 4);
        s.want(s.wide_cnt, 64 * 4);
    }
    SCANN_TRY(s.commit());
    w->queries = s.queries.as<float>();
    w->cdist = s.cdist.as<float>();
    w->tokens = s.tokens.as<uint32_t>();
    w->token_dists = s.token_dists.as<float>();
    w->vbase = s.vbase.as<uint32_t>();
    if (w->use_i8) {
        w->rr_lb = s.rr_lb.as<uint32_t>();
        w->rr_ub = s.rr_ub.as<uint32_t>();
    }
    if (w->mfma) {
        w->lut8 = s.lut8.as<int8_t>();
        w->lut8_meta = s.lut8_meta.p;
        w->mfma_thr1 = s.mfma_thr1.as<int>();
        w->cand32 = s.cand32.as<uint32_t>();
        w->cand32_codes = t.ah_mode ? s.cand32_codes.as<uint32_t>() : nullptr;
        w->cand32_cnt = s.cand32_cnt.as<uint32_t>();
    }
    if (p.small) w->small_tickets = s.small_tickets.as<uint32_t>();
    w->wide_cap2 = p.wide_cap2;
    w->wide_min = w->wide_ceb = w->wide_cidx = w->wide_cnt = nullptr;
    w->wide_ckey = nullptr;
    if (p.small == 2) {
        w->wide_min = s.wide_min.as<uint32_t>();
        w->wide_ckey = s.wide_ckey.as<uint64_t>();
        w->wide_ceb = s.wide_ceb.as<uint32_t>();
        w->wide_cidx = s.wide_cidx.as<uint32_t>();
        w->wide_cnt = s.wide_cnt.as<uint32_t>();
    }
    w->sbase = s.sbase.as<uint32_t>();
    w->pair_sbase = s.pair_sbase.as<uint32_t>();
    w->stile_off = s.stile_off.as<uint32_t>();
    w->samp = s.samp.as<uint32_t>();
    w->leaf_cnt = s.leaf_cnt.as<uint32_t>();
    w->leaf_cursor = s.leaf_cursor.as<uint32_t>();
    w->pair_off = s.pair_off.as<uint32_t>();
    w->tile_off = s.tile_off.as<uint32_t>();
    w->counters = s.counters.as<uint32_t>();
    w->pair_q = s.pair_q.as<uint32_t>();
    w->pair_leaf = s.pair_leaf.as<uint32_t>();
    w->pair_vbase = s.pair_vbase.as<uint32_t>();
    w->pair_thr = s.pair_thr.as<uint64_t>();
    w->slot_of = s.slot_of.as<uint32_t>();
    w->max_slots = max_slots;
    w->max_quads = max_quads;
    w->lutq = s.lutq.as<float>();
    w->thr = s.thr.as<uint64_t>();
    w->cand_cnt = s.cand_cnt.as<uint32_t>();
    w->cand = s.cand.as<uint64_t>();
    w->cand_key = s.cand_key.as<uint64_t>();
    w->cand_idx = s.cand_idx.as<uint32_t>();
    w->cand_dist = s.cand_dist.as<float>();
    w->cand_exact = s.cand_exact.as<float>();
    w->cand_row = s.cand_row.as<uint32_t>();
    w->cand_count = s.cand_count.as<uint32_t>();
    w->out_idx = s.out_idx.as<uint32_t>();
    w->out_dist = s.out_dist.as<float>();
    w->out_count = s.out_count.as<uint32_t>();
    return SCANN_HIP_OK;
}

static void fill_empty(uint32_t nq, uint32_t k, uint32_t *out_idx, float *out_dist,
                       uint32_t *out_count) {
    for (size_t i = 0; i < (size_t)nq * k; ++i) {
        if (out_idx) out_idx[i] = 0xFFFFFFFFu;
        if (out_dist) out_dist[i] = INFINITY;
    }
    for (uint32_t i = 0; i < nq; ++i)
        if (out_count) out_count[i] = 0;
}

// ---- workspaces of the device entry points: one per caller stream (ix->mu held) ---------------------------------
static int device_slot_count() {
    static const int v = [] {
        const char *e = std::getenv("SCANN_HIP_DEVICE_SLOTS");
        return std::max(1, std::min((int)scann_hip_index::kMaxDeviceSlots, e ? std::atoi(e) : 2));
    }();
    return v;
}

// The workspace of a call that will be enqueued on `st`: the one bound to that stream, else an unused one, else the
// least recently used one -- ordered behind its previous stream's last call.
static int device_slot(scann_hip_index *ix, hipStream_t st, scann_hip_index::DeviceSlot **out) {
    auto &ds = ix->dslots;
    if (!ds[0].ws) {
        ds[0].ws = &ix->ws;
        ds[0].bfw = &ix->bfw;
    }
    const int n = device_slot_count();
    scann_hip_index::DeviceSlot *pick = nullptr;
    for (int i = 0; i < n && !pick; ++i)
        if (ds[i].used && ds[i].key == st) pick = &ds[i];
    for (int i = 0; i < n && !pick; ++i)
        if (!ds[i].used) pick = &ds[i];
    if (!pick) {
        pick = &ds[0];
        for (int i = 1; i < n; ++i)
            if (ds[i].tick < pick->tick) pick = &ds[i];
    }
    if (!pick->ws) {
        pick->ws_own.reset(new TxhWorkspace());
        pick->bfw_own.reset(new BfWorkspace());
        pick->ws = pick->ws_own.get();
        pick->bfw = pick->bfw_own.get();
    }
    if (!pick->done) SCANN_HIP_CHECK(hipEventCreateWithFlags(&pick->done, hipEventDisableTiming));
    if (pick->used && pick->key != st && pick->done_valid) SCANN_HIP_CHECK(hipStreamWaitEvent(st, pick->done, 0));
    pick->used = true;
    pick->key = st;
    pick->tick = ++ix->dslot_tick;
    *out = pick;
    return SCANN_HIP_OK;
}

// after the call's last enqueue on `st`
static int device_slot_done(scann_hip_index::DeviceSlot *sl, hipStream_t st) {
    SCANN_HIP_CHECK(hipEventRecord(sl->done, st));
    sl->done_valid = true;
    return SCANN_HIP_OK;
}

// Host-side users of the primary workspace (they run on ix->stream and synchronise before they return): wait for a
// device-entry call that may still be using it on another stream.
static int claim_primary_workspace(scann_hip_index *ix) {
    auto &d0 = ix->dslots[0];
    if (d0.used && d0.key != ix->stream) {
        if (d0.done_valid) SCANN_HIP_CHECK(hipStreamWaitEvent(ix->stream, d0.done, 0));
        d0.used = false;
        d0.done_valid = false;
        d0.key = nullptr;
    }
    return SCANN_HIP_OK;
}

// A search slot for a host-side call: the primary one if it is free, else a free extra slot (created
// up to SCANN_HIP_SEARCH_SLOTS, default 4, streams + workspaces in total), else wait for the primary.
struct SlotLock {
    std::unique_lock<std::mutex> lock;
    hipStream_t stream = nullptr;
    TxhWorkspace *ws = nullptr;
    BfWorkspace *bfw = nullptr;
    PinBuf *pin = nullptr;
    bool primary = false;
};

static uint32_t max_slots() {
    static const uint32_t v = [] {
        const char *e = std::getenv("SCANN_HIP_SEARCH_SLOTS");
        return (uint32_t)std::max(1, std::min(64, e ? std::atoi(e) : 4));
    }();
    return v;
}

static int acquire_slot(scann_hip_index *ix, SlotLock *out) {
    std::unique_lock<std::mutex> lk(ix->mu, std::try_to_lock);
    if (!lk.owns_lock()) {
        SearchSlot *slot = nullptr;
        {
            std::lock_guard<std::mutex> g(ix->slots_mu);
            for (auto &sl : ix->slots) {
                std::unique_lock<std::mutex> l2(sl->mu, std::try_to_lock);
                if (l2.owns_lock()) {
                    slot = sl.get();
                    out->lock = std::move(l2);
                    break;
                }
            }
            if (!slot && ix->slots.size() + 1 < max_slots()) {
                std::unique_ptr<SearchSlot> ns(new SearchSlot());
                SCANN_TRY(set_device(ix->ctx));
                SCANN_HIP_CHECK(hipStreamCreateWithFlags(&ns->stream, hipStreamNonBlocking));
                out->lock = std::unique_lock<std::mutex>(ns->mu);
                slot = ns.get();
                ix->slots.push_back(std::move(ns));
            }
        }
        if (slot) {
            out->stream = slot->stream;
            out->ws = &slot->ws;
            out->bfw = &slot->bfw;
            out->pin = &slot->pin;
            out->primary = false;
            return SCANN_HIP_OK;
        }
        lk.lock();   // every slot busy: queue on the primary
    }
    out->lock = std::move(lk);
    SCANN_TRY(claim_primary_workspace(ix));
    out->stream = ix->stream;
    out->ws = &ix->ws;
    out->bfw = &ix->bfw;
    out->pin = &ix->pin;
    out->primary = true;
    return SCANN_HIP_OK;
}

// Completion of a small-batch call whose last kernel stores `seq` into the nq pinned flag words after
// its result rows (system-scope release): the host polls the flags instead of paying the wake-up
// latency of hipStreamSynchronize for a 30-microsecond job.  Falls back to the stream synchronise after
// ~2 ms without completion (long jobs, or a launch that failed: the synchronise reports it).
static int wait_small_done(hipStream_t stream, const volatile uint32_t *flags, uint32_t nq, uint32_t seq) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
        bool all = true;
        for (uint32_t i = 0; i < nq; ++i)
            if (flags[i] != seq) {
                all = false;
                break;
            }
        if (all) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return SCANN_HIP_OK;
        }
        if ((spin & 255u) == 255u &&
            std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000))
            break;
        __builtin_ia32_pause();
    }
    SCANN_HIP_CHECK(hipStreamSynchronize(stream));
    return SCANN_HIP_OK;
}

static int txh_search_host(scann_hip_index *ix, const float *queries, uint32_t nq,
                           uint32_t q_stride, uint32_t k, const scann_hip_search_opts *opts,
                           uint32_t *out_idx, float *out_dist, uint32_t *out_count) {
    SlotLock sl;
    SCANN_TRY(acquire_slot(ix, &sl));
    TxhWorkspace &ws = *sl.ws;
    const hipStream_t stream = sl.stream;
    SCANN_TRY(set_device(ix->ctx));
    for (int attempt = 0; attempt < 2; ++attempt) {
        TxhCallParams p;
        SCANN_TRY(resolve_params(ix, k, opts, /*full_cap=*/attempt == 1, &p, nq, /*allow_wide=*/attempt == 0));
        if (p.m == 0) {  // nothing can be kept (reference panics on FastTopNeighbors::new(0))
            fill_empty(nq, k, out_idx, out_dist, out_count);
            if (opts && opts->cand_count) std::memset(opts->cand_count, 0, (size_t)nq * 4);
            return SCANN_HIP_OK;
        }
        TxhWork w;
        const size_t allow_words = (opts && opts->allow_bitmap) ? (size_t)((opts->allow_bitmap_bits + 63) / 64) : 0;
        SCANN_TRY(ensure_txh_workspace(ix, ws, nq, p, true, q_stride, true, &w, allow_words * 8));
        // (cand_count alone also takes the staged pipeline: the small-batch one keeps no candidate counts)
        w.need_sorted_cands = (opts && (opts->cand_idx || opts->cand_dist || opts->cand_count)) ? 1 : 0;
        const bool stage_outputs = opts && (opts->tokens || opts->token_dists || opts->cand_idx || opts->cand_dist ||
                                            opts->cand_count);
        if (w.small && !stage_outputs && !(opts && opts->allow_bitmap)) {
            // small batch: queries and result rows live in pinned host memory the kernels access in place
            const size_t qb = (size_t)nq * q_stride * 4, ob = (size_t)nq * k * 4;
            const size_t off_idx = (qb + 255) & ~(size_t)255, off_dist = off_idx + ((ob + 255) & ~(size_t)255),
                         off_cnt = off_dist + ((ob + 255) & ~(size_t)255);
            const size_t off_flag = off_cnt + (((size_t)nq * 4 + 255) & ~(size_t)255);
            SCANN_TRY(sl.pin->ensure(off_flag + (size_t)nq * 4 + 256));
            char *hp = static_cast<char *>(sl.pin->host), *dp = static_cast<char *>(sl.pin->dev);
            std::memcpy(hp, queries, qb);
            const uint32_t seq = ++sl.pin->seq ? sl.pin->seq : ++sl.pin->seq;   // (never 0)
            // The flag words sit wherever THIS call's layout puts them: an earlier call (other nq / k / stride) may
            // have left result words there, and any of those can equal seq.  Zero them before the launch.
            std::memset(hp + off_flag, 0, (size_t)nq * 4);
            w.queries = reinterpret_cast<const float *>(dp);
            w.out_idx = reinterpret_cast<uint32_t *>(dp + off_idx);
            w.out_dist = reinterpret_cast<float *>(dp + off_dist);
            w.out_count = reinterpret_cast<uint32_t *>(dp + off_cnt);
            w.small_done = reinterpret_cast<uint32_t *>(dp + off_flag);
            w.small_seq = seq;
            if (sl.primary) ix->next_events();
            SCANN_TRY(txh_launch_search(ix->tx, w, false, stream, sl.primary ? ix->ev0 : nullptr,
                                        sl.primary ? ix->ev1 : nullptr));
            if (sl.primary) {
                ix->timing_valid = ix->timing;
                ix->timed_kernel = w.small == 2 ? "wide_scan_kernel" : "small_scan_kernel";
            }
            SCANN_TRY(wait_small_done(stream, reinterpret_cast<const volatile uint32_t *>(hp + off_flag), nq, seq));
            if (w.small == 2) {   // the wide pipeline's compact arrays can overflow: count row 0xFFFFFFFF -> the batched pipeline
                bool overflow = false;
                for (uint32_t i = 0; i < nq; ++i) overflow = overflow || reinterpret_cast<const uint32_t *>(hp + off_cnt)[i] == 0xFFFFFFFFu;
                if (overflow) continue;
            }
            std::memcpy(out_idx, hp + off_idx, ob);
            std::memcpy(out_dist, hp + off_dist, ob);
            std::memcpy(out_count, hp + off_cnt, (size_t)nq * 4);
            return SCANN_HIP_OK;   // (dense candidate lists: the three-launch / one-launch forms have no overflow / retry case)
        }
        if (opts && opts->allow_bitmap) {   // search_with_filter(Some(allow-list))
            SCANN_HIP_CHECK(hipMemcpyAsync(ws.allow.p, opts->allow_bitmap, allow_words * 8,
                                           hipMemcpyHostToDevice, stream));
            w.allow = ws.allow.as<uint64_t>();
            w.allow_bits = opts->allow_bitmap_bits;
        }
        SCANN_HIP_CHECK(hipMemcpyAsync(ws.queries.p, queries, (size_t)nq * q_stride * 4,
                                       hipMemcpyHostToDevice, stream));
        if (sl.primary) ix->next_events();   // kernel timing follows the primary slot only
        SCANN_TRY(txh_launch_search(ix->tx, w, false, stream, sl.primary ? ix->ev0 : nullptr,
                                    sl.primary ? ix->ev1 : nullptr));
        if (sl.primary) ix->timing_valid = ix->timing;
        if (sl.primary) ix->timed_kernel = ix->tx.exact_scan ? "leaf_exact_scan_kernel" : w.mfma == 2 ? "adc_mfma16_kernel" : w.mfma == 3 ? "adc_smfmac_kernel" : w.mfma ? "adc_mfma_kernel" : w.resident ? "adc_scan_res_kernel" : "adc_scan_kernel";
        uint32_t counters[CNT_N];
        SCANN_HIP_CHECK(hipMemcpyAsync(counters, w.counters, sizeof(counters), hipMemcpyDeviceToHost,
                                       stream));
        SCANN_HIP_CHECK(hipMemcpyAsync(out_idx, w.out_idx, (size_t)nq * k * 4, hipMemcpyDeviceToHost,
                                       stream));
        SCANN_HIP_CHECK(hipMemcpyAsync(out_dist, w.out_dist, (size_t)nq * k * 4,
                                       hipMemcpyDeviceToHost, stream));
        SCANN_HIP_CHECK(hipMemcpyAsync(out_count, w.out_count, (size_t)nq * 4, hipMemcpyDeviceToHost,
                                       stream));
        if (opts) {
            if (opts->tokens)
                SCANN_HIP_CHECK(hipMemcpyAsync(opts->tokens, w.tokens, (size_t)nq * p.P * 4,
                                               hipMemcpyDeviceToHost, stream));
            if (opts->token_dists)
                SCANN_HIP_CHECK(hipMemcpyAsync(opts->token_dists, w.token_dists, (size_t)nq * p.P * 4,
                                               hipMemcpyDeviceToHost, stream));
            if (opts->cand_idx)
                SCANN_HIP_CHECK(hipMemcpyAsync(opts->cand_idx, w.cand_idx, (size_t)nq * p.m * 4,
                                               hipMemcpyDeviceToHost, stream));
            if (opts->cand_dist)
                SCANN_HIP_CHECK(hipMemcpyAsync(opts->cand_dist, w.cand_dist, (size_t)nq * p.m * 4,
                                               hipMemcpyDeviceToHost, stream));
            if (opts->cand_count)
                SCANN_HIP_CHECK(hipMemcpyAsync(opts->cand_count, w.cand_count, (size_t)nq * 4,
                                               hipMemcpyDeviceToHost, stream));
        }
        SCANN_HIP_CHECK(hipStreamSynchronize(stream));
        if (counters[CNT_STATUS] == SCANN_HIP_OK) return SCANN_HIP_OK;
        if ((counters[CNT_STATUS] != SCANN_HIP_RESOURCE_EXHAUSTED &&
             counters[CNT_STATUS] != SCANN_HIP_ABORTED) || attempt == 1)
            return fail((int)counters[CNT_STATUS], "device reported a search failure");
        // candidate buffer overflow, or a statistical threshold that kept fewer than m
        // points: retry once without a threshold and with a buffer for the whole stream
    }
    return fail(SCANN_HIP_INTERNAL, "unreachable");
}

}  // extern "C"

// BruteForceSearcher::search for a handful of queries over a small dataset: the three-launch pipeline of
// txh.hip ("Small batches") on the one-leaf exact-scan view of the rows -- every distance with the
// one-to-many kernels' arithmetic, the k smallest (distance, index) keys = TopK -- with queries and result
// rows in pinned host memory (no copy commands).
static int bf_small_search_host(scann_hip_index *ix, SlotLock &sl, const float *queries, uint32_t nq,
                                uint32_t q_stride, uint32_t k, uint32_t *out_idx, float *out_dist,
                                uint32_t *out_count) {
    TxhWorkspace &ws = *sl.ws;
    const uint32_t n = (uint32_t)ix->bf.n, kk = std::min(k, n);
    ws.want(ws.tokens, (size_t)nq * 4);
    ws.want(ws.token_dists, (size_t)nq * 4);
    ws.want(ws.vbase, (size_t)nq * 2 * 4);
    ws.want(ws.sbase, (size_t)nq * 3 * 4);
    ws.want(ws.counters, CNT_WORDS * 4);
    ws.want(ws.cand, (size_t)nq * n * 8);
    ws.want(ws.small_tickets, 64 * 4);
    SCANN_TRY(ws.commit());
    const size_t qb = (size_t)nq * q_stride * 4, ob = (size_t)nq * k * 4;
    const size_t off_idx = (qb + 255) & ~(size_t)255, off_dist = off_idx + ((ob + 255) & ~(size_t)255),
                 off_cnt = off_dist + ((ob + 255) & ~(size_t)255);
    const size_t off_flag = off_cnt + (((size_t)nq * 4 + 255) & ~(size_t)255);
    SCANN_TRY(sl.pin->ensure(off_flag + (size_t)nq * 4 + 256));
    char *hp = static_cast<char *>(sl.pin->host), *dp = static_cast<char *>(sl.pin->dev);
    std::memcpy(hp, queries, qb);
    const uint32_t seq = ++sl.pin->seq ? sl.pin->seq : ++sl.pin->seq;
    std::memset(hp + off_flag, 0, (size_t)nq * 4);   // (stale words of an earlier call's layout: see txh_search_host)
    TxhWork w{};
    w.nq = nq; w.q_stride = q_stride; w.P = 1; w.m = kk; w.k = k; w.cap = n; w.exact_reorder = 0;
    w.no_threshold = 1; w.need_sorted_cands = 0; w.allow = nullptr; w.allow_bits = 0;
    w.queries = reinterpret_cast<const float *>(dp);
    w.tokens = ws.tokens.as<uint32_t>(); w.token_dists = ws.token_dists.as<float>();
    w.vbase = ws.vbase.as<uint32_t>(); w.sbase = ws.sbase.as<uint32_t>(); w.st = 1;
    w.counters = ws.counters.as<uint32_t>(); w.cand = ws.cand.as<uint64_t>();
    w.out_idx = reinterpret_cast<uint32_t *>(dp + off_idx);
    w.out_dist = reinterpret_cast<float *>(dp + off_dist);
    w.out_count = reinterpret_cast<uint32_t *>(dp + off_cnt);
    w.small = 1; w.small_max_leaf = n;
    w.small_tickets = ws.small_tickets.as<uint32_t>();
    w.small_done = reinterpret_cast<uint32_t *>(dp + off_flag);
    w.small_seq = seq;
    SCANN_TRY(txh_launch_search(ix->bfx, w, false, sl.stream, nullptr, nullptr));
    SCANN_TRY(wait_small_done(sl.stream, reinterpret_cast<const volatile uint32_t *>(hp + off_flag), nq, seq));
    std::memcpy(out_idx, hp + off_idx, ob);
    std::memcpy(out_dist, hp + off_dist, ob);
    std::memcpy(out_count, hp + off_cnt, (size_t)nq * 4);
    return SCANN_HIP_OK;
}

extern "C" {

int scann_hip_search_batched(scann_hip_index *ix, const float *queries, uint32_t nq,
                             uint32_t q_stride, uint32_t q_dim, uint32_t k,
                             const scann_hip_search_opts *opts, uint32_t *out_idx, float *out_dist,
                             uint32_t *out_count) {
    if (!ix) return fail(SCANN_HIP_INVALID_ARGUMENT, "index is null");
    if (nq == 0) return SCANN_HIP_OK;
    if (!queries || !out_count || (k > 0 && (!out_idx || !out_dist)))
        return fail(SCANN_HIP_INVALID_ARGUMENT, "null query/output pointer");
    if (ix->kind == KIND_BF) {
        if (ix->bf.n == 0) {  // brute_force/searcher.rs:78-80: Ok(empty) before the dim check
            fill_empty(nq, k, out_idx, out_dist, out_count);
            return SCANN_HIP_OK;
        }
        if (q_dim != ix->bf.dim)  // searcher.rs:83-89
            return fail(SCANN_HIP_INVALID_ARGUMENT,
                        "Query dimensionality " + std::to_string(q_dim) +
                            " does not match dataset dimensionality " + std::to_string(ix->bf.dim));
        if (q_stride < q_dim) return fail(SCANN_HIP_INVALID_ARGUMENT, "q_stride < q_dim");
        if (k == 0) {
            fill_empty(nq, 0, nullptr, nullptr, out_count);
            return SCANN_HIP_OK;
        }
        SlotLock sl;
        SCANN_TRY(acquire_slot(ix, &sl));
        SCANN_TRY(set_device(ix->ctx));
        const bool small_on = small_batch_enabled();
        if (small_on && ix->bfx.rows && nq <= kSmallBatch && k <= 64)
            return bf_small_search_host(ix, sl, queries, nq, q_stride, k, out_idx, out_dist, out_count);
        if (sl.primary) ix->next_events();
        int s = bf_search_host(ix->bf, *sl.bfw, queries, nq, q_stride, k, opts && opts->bf_exact, out_idx,
                               out_dist, out_count, sl.stream, sl.primary ? ix->ev0 : nullptr,
                               sl.primary ? ix->ev1 : nullptr);
        if (sl.primary) {
            ix->timing_valid = ix->timing && s == SCANN_HIP_OK;
            ix->timed_kernel = (!(opts && opts->bf_exact) && bf_shortlist_eligible(ix->bf, nq, k))
                                   ? "bf_bf16_kernel" : bf_pass_kernel_name(ix->bf, nq);
        }
        return s;
    }
    if (q_dim != ix->tx.dim)  // tree_x_hybrid/mod.rs:251-253, hashes/hasher.rs:167-171
        return fail(SCANN_HIP_INVALID_ARGUMENT, "Query dimensionality mismatch");
    if (q_stride < q_dim) return fail(SCANN_HIP_INVALID_ARGUMENT, "q_stride < q_dim");
    if (k == 0) {
        fill_empty(nq, 0, nullptr, nullptr, out_count);
        return SCANN_HIP_OK;
    }
    return txh_search_host(ix, queries, nq, q_stride, k, opts, out_idx, out_dist, out_count);
}

int scann_hip_search_batched_params(scann_hip_index *ix, const float *queries, uint32_t nq, uint32_t q_stride,
                                    uint32_t q_dim, const uint32_t *k_per_query, const scann_hip_search_opts *opts,
                                    uint32_t out_pitch, uint32_t *out_idx, float *out_dist, uint32_t *out_count) {
    if (!ix) return fail(SCANN_HIP_INVALID_ARGUMENT, "index is null");
    if (nq == 0) return SCANN_HIP_OK;
    if (!queries || !k_per_query || !out_count) return fail(SCANN_HIP_INVALID_ARGUMENT, "null query/k/output pointer");
    if (q_stride < q_dim) return fail(SCANN_HIP_INVALID_ARGUMENT, "q_stride < q_dim");
    uint32_t kmax = 0;
    for (uint32_t i = 0; i < nq; ++i) kmax = std::max(kmax, k_per_query[i]);
    if (kmax > out_pitch) return fail(SCANN_HIP_INVALID_ARGUMENT, "out_pitch is smaller than the largest k");
    if (kmax > 0 && (!out_idx || !out_dist)) return fail(SCANN_HIP_INVALID_ARGUMENT, "null output pointer");
    for (size_t i = 0; i < (size_t)nq * out_pitch; ++i) {
        out_idx[i] = 0xFFFFFFFFu;
        out_dist[i] = INFINITY;
    }
    // one batch per distinct k, in order of first appearance
    std::vector<uint32_t> order(nq);
    for (uint32_t i = 0; i < nq; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return k_per_query[a] < k_per_query[b]; });
    std::vector<float> qbuf;
    std::vector<uint32_t> gi, gc;
    std::vector<float> gd;
    for (uint32_t a0 = 0; a0 < nq;) {
        const uint32_t k = k_per_query[order[a0]];
        uint32_t a1 = a0;
        while (a1 < nq && k_per_query[order[a1]] == k) ++a1;
        const uint32_t g = a1 - a0;
        qbuf.resize((size_t)g * q_stride);
        for (uint32_t j = 0; j < g; ++j)
            std::memcpy(&qbuf[(size_t)j * q_stride], queries + (size_t)order[a0 + j] * q_stride, (size_t)q_stride * 4);
        gi.assign((size_t)g * std::max(1u, k), 0xFFFFFFFFu);
        gd.assign((size_t)g * std::max(1u, k), INFINITY);
        gc.assign(g, 0u);
        SCANN_TRY(scann_hip_search_batched(ix, qbuf.data(), g, q_stride, q_dim, k, opts, gi.data(), gd.data(), gc.data()));
        for (uint32_t j = 0; j < g; ++j) {
            const uint32_t q = order[a0 + j];
            out_count[q] = gc[j];
            for (uint32_t r = 0; r < gc[j]; ++r) {
                out_idx[(size_t)q * out_pitch + r] = gi[(size_t)j * k + r];
                out_dist[(size_t)q * out_pitch + r] = gd[(size_t)j * k + r];
            }
        }
        a0 = a1;
    }
    return SCANN_HIP_OK;
}

int scann_hip_index_reserve(scann_hip_index *ix, uint32_t max_nq, uint32_t max_k,
                            const scann_hip_search_opts *opts) {
    if (!ix) return fail(SCANN_HIP_INVALID_ARGUMENT, "index is null");
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    if (ix->kind == KIND_BF) return bf_reserve(ix->bf, ix->bfw, max_nq, max_k);
    TxhCallParams p;
    SCANN_TRY(resolve_params(ix, max_k, opts, false, &p));
    TxhWork w;
    return ensure_txh_workspace(ix, ix->ws, max_nq, p, false, 0, false, &w);
}

int scann_hip_search_batched_device(scann_hip_index *ix, const float *d_queries, uint32_t nq,
                                    uint32_t q_stride, uint32_t k, const scann_hip_search_opts *opts,
                                    uint32_t *d_out_idx, float *d_out_dist, uint32_t *d_out_count,
                                    void *hip_stream) {
    if (!ix) return fail(SCANN_HIP_INVALID_ARGUMENT, "index is null");
    if (nq == 0) return SCANN_HIP_OK;
    if (k == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "k must be > 0 on the device path");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    scann_hip_index::DeviceSlot *dsl = nullptr;
    SCANN_TRY(device_slot(ix, st, &dsl));
    if (ix->kind == KIND_BF) {
        ix->next_events();
        const bool exact_only = opts && opts->bf_exact;
        int s = bf_search_device(ix->bf, *dsl->bfw, d_queries, nq, q_stride, k, exact_only, d_out_idx,
                                 d_out_dist, d_out_count, st, ix->ev0, ix->ev1);
        if (s == SCANN_HIP_OK) s = device_slot_done(dsl, st);
        ix->timing_valid = ix->timing && s == SCANN_HIP_OK;
        ix->timed_kernel = (!exact_only && bf_shortlist_eligible(ix->bf, nq, k)) ? "bf_bf16_kernel"
                                                                                : bf_pass_kernel_name(ix->bf, nq);
        return s;
    }
    TxhCallParams p;
    SCANN_TRY(resolve_params(ix, k, opts, false, &p, nq));
    if (p.m == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "pre-reorder candidate count is 0");
    TxhWork w;
    SCANN_TRY(ensure_txh_workspace(ix, *dsl->ws, nq, p, false, q_stride, false, &w));
    w.queries = d_queries;
    w.out_idx = d_out_idx;
    w.out_dist = d_out_dist;
    w.out_count = d_out_count;
    if (opts && opts->allow_bitmap) {   // device pointer on this path
        w.allow = opts->allow_bitmap;
        w.allow_bits = opts->allow_bitmap_bits;
    }
    ix->last_work = w;
    ix->next_events();
    SCANN_TRY(txh_launch_search(ix->tx, w, false, st, ix->ev0,
                                ix->ev1));
    SCANN_TRY(device_slot_done(dsl, st));
    ix->timing_valid = ix->timing;
    ix->timed_kernel = ix->tx.exact_scan ? "leaf_exact_scan_kernel" : w.mfma == 2 ? "adc_mfma16_kernel" : w.mfma == 3 ? "adc_smfmac_kernel" : w.mfma ? "adc_mfma_kernel" : w.resident ? "adc_scan_res_kernel" : "adc_scan_kernel";
    return SCANN_HIP_OK;
}

int scann_hip_index_last_device_status(scann_hip_index *ix, void *hip_stream) {
    if (!ix) return fail(SCANN_HIP_INVALID_ARGUMENT, "index is null");
    SCANN_TRY(set_device(ix->ctx));
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    std::lock_guard<std::mutex> lock(ix->mu);
    // the workspace bound to this stream (else the primary one)
    const TxhWorkspace *ws = &ix->ws;
    const BfWorkspace *bfw = &ix->bfw;
    for (auto &d : ix->dslots)
        if (d.used && d.key == st && d.ws) {
            ws = d.ws;
            bfw = d.bfw;
        }
    if (ix->kind == KIND_BF) return bf_last_status(*bfw, st);
    if (ix->kind != KIND_TXH || !ws->counters.p) return SCANN_HIP_OK;
    uint32_t counters[CNT_N];
    SCANN_HIP_CHECK(hipMemcpyAsync(counters, ws->counters.p, sizeof(counters),
                                   hipMemcpyDeviceToHost, st));
    SCANN_HIP_CHECK(hipStreamSynchronize(st));
    if (counters[CNT_STATUS] != SCANN_HIP_OK)
        return fail((int)counters[CNT_STATUS],
                    "candidate threshold/buffer miss on the device path (use the host entry "
                    "point, which retries without a threshold and with a full-size buffer)");
    return SCANN_HIP_OK;
}

// ---- multi-GPU ------------------------------------------------------------------------
int scann_hip_txh_search_local_device(scann_hip_index *ix, const float *d_queries, uint32_t nq,
                                      uint32_t q_stride, uint32_t k,
                                      const scann_hip_search_opts *opts, uint64_t *d_keys,
                                      uint32_t *d_idx, float *d_exact, uint32_t *d_count,
                                      void *hip_stream) {
    if (!ix || ix->kind != KIND_TXH) return fail(SCANN_HIP_INVALID_ARGUMENT, "not a tree index");
    if (nq == 0) return SCANN_HIP_OK;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    TxhCallParams p;
    SCANN_TRY(resolve_params(ix, k, opts, false, &p));
    if (!p.exact_reorder) return fail(SCANN_HIP_INVALID_ARGUMENT, "local stage needs exact_reorder");
    if (p.m == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "pre-reorder candidate count is 0");
    scann_hip_index::DeviceSlot *dsl = nullptr;
    SCANN_TRY(device_slot(ix, st, &dsl));
    TxhWork w;
    SCANN_TRY(ensure_txh_workspace(ix, *dsl->ws, nq, p, false, q_stride, false, &w));
    w.queries = d_queries;
    w.cand_key = d_keys;
    w.cand_idx = d_idx;
    w.cand_exact = d_exact;
    w.cand_count = d_count;
    if (opts && opts->allow_bitmap) {   // device pointer on this path
        w.allow = opts->allow_bitmap;
        w.allow_bits = opts->allow_bitmap_bits;
    }
    ix->last_work = w;
    ix->next_events();
    SCANN_TRY(txh_launch_search(ix->tx, w, true, st, ix->ev0,
                                ix->ev1));
    SCANN_TRY(device_slot_done(dsl, st));
    ix->timing_valid = ix->timing;
    ix->timed_kernel = ix->tx.exact_scan ? "leaf_exact_scan_kernel" : w.mfma == 2 ? "adc_mfma16_kernel" : w.mfma == 3 ? "adc_smfmac_kernel" : w.mfma ? "adc_mfma_kernel" : w.resident ? "adc_scan_res_kernel" : "adc_scan_kernel";
    return SCANN_HIP_OK;
}

int scann_hip_txh_merge_device(scann_hip_ctx *ctx, uint32_t world, uint32_t nq, uint32_t m_local,
                               uint32_t m, uint32_t k, uint64_t rank_stride_bytes,
                               const uint64_t *d_keys, const uint32_t *d_idx,
                               const float *d_exact, const uint32_t *d_count, uint32_t *d_out_idx,
                               float *d_out_dist, uint32_t *d_out_count, uint32_t *d_status,
                               void *hip_stream) {
    if (!ctx) return fail(SCANN_HIP_INVALID_ARGUMENT, "ctx is null");
    SCANN_TRY(set_device(ctx));
    return txh_launch_merge(world, nq, m_local, m, k, (size_t)rank_stride_bytes, d_keys, d_idx, d_exact,
                            d_count, d_out_idx,
                            d_out_dist, d_out_count, d_status, static_cast<hipStream_t>(hip_stream));
}

int scann_hip_txh_pack_blocks_device(scann_hip_ctx *ctx, uint32_t world, uint32_t nq, uint32_t m_local,
                                     const uint64_t *d_keys, const uint32_t *d_idx, const float *d_exact,
                                     const uint32_t *d_count, void *d_out, uint64_t block_bytes,
                                     void *hip_stream) {
    if (!ctx) return fail(SCANN_HIP_INVALID_ARGUMENT, "ctx is null");
    if (!d_keys || !d_idx || !d_exact || !d_count || !d_out)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "null buffer");
    SCANN_TRY(set_device(ctx));
    return txh_launch_pack_blocks(world, nq, m_local, d_keys, d_idx, d_exact, d_count, d_out,
                                  (size_t)block_bytes, static_cast<hipStream_t>(hip_stream));
}

int scann_hip_assign_leaves(const uint32_t *sizes, uint32_t L, uint32_t world, uint32_t *owner) {
    if (!sizes || !owner || world == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "bad arguments");
    std::vector<uint32_t> order(L);
    for (uint32_t i = 0; i < L; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return sizes[a] > sizes[b]; });
    std::vector<uint64_t> load(world, 0);
    for (uint32_t i : order) {
        uint32_t best = 0;
        for (uint32_t g = 1; g < world; ++g)
            if (load[g] < load[best]) best = g;
        owner[i] = best;
        load[best] += sizes[i];
    }
    return SCANN_HIP_OK;
}

// ---- building blocks ---------------------------------------------------------------------
int scann_hip_txh_partition(scann_hip_index *ix, const float *queries, uint32_t nq,
                            uint32_t q_stride, uint32_t q_dim, uint32_t num_partitions,
                            uint32_t *out_tokens, float *out_dists, uint32_t *out_count) {
    if (!ix || ix->kind != KIND_TXH) return fail(SCANN_HIP_INVALID_ARGUMENT, "not a tree index");
    if (ix->tx.ah_mode)  // tree_partitioner.rs:197-198
        return fail(SCANN_HIP_FAILED_PRECONDITION, "Partitioner not built");
    if (q_dim != ix->tx.dim) return fail(SCANN_HIP_INVALID_ARGUMENT, "Query dimensionality mismatch");
    if (nq == 0) return SCANN_HIP_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    scann_hip_search_opts o;
    scann_hip_search_opts_default(&o);
    o.partitions_to_search = std::max(1u, num_partitions);
    o.exact_reorder = 0;
    TxhCallParams p;
    SCANN_TRY(resolve_params(ix, 1, &o, false, &p));
    TxhWork w;
    SCANN_TRY(claim_primary_workspace(ix));
    SCANN_TRY(ensure_txh_workspace(ix, ix->ws, nq, p, true, q_stride, true, &w));
    SCANN_HIP_CHECK(hipMemcpyAsync(ix->ws.queries.p, queries, (size_t)nq * q_stride * 4,
                                   hipMemcpyHostToDevice, ix->stream));
    SCANN_TRY(txh_launch_partition_only(ix->tx, w, ix->stream));
    const uint32_t P = num_partitions == 0 ? 0 : p.P;
    if (P) {
        SCANN_HIP_CHECK(hipMemcpyAsync(out_tokens, w.tokens, (size_t)nq * P * 4, hipMemcpyDeviceToHost,
                                       ix->stream));
        SCANN_HIP_CHECK(hipMemcpyAsync(out_dists, w.token_dists, (size_t)nq * P * 4,
                                       hipMemcpyDeviceToHost, ix->stream));
    }
    SCANN_HIP_CHECK(hipStreamSynchronize(ix->stream));
    if (out_count)
        for (uint32_t i = 0; i < nq; ++i) out_count[i] = P;
    return SCANN_HIP_OK;
}

int scann_hip_lut_from_query(scann_hip_index *ix, const float *queries, uint32_t nq,
                             uint32_t q_stride, const uint32_t *leaf_for_query, float *out_lut) {
    if (!ix || ix->kind != KIND_TXH) return fail(SCANN_HIP_INVALID_ARGUMENT, "not a tree index");
    if (nq == 0) return SCANN_HIP_OK;
    if (leaf_for_query && ix->tx.ah_mode)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "no centroids in AsymmetricHasher mode");
    if (leaf_for_query)
        for (uint32_t i = 0; i < nq; ++i)
            if (leaf_for_query[i] >= ix->tx.L)  // mod.rs:304-306
                return fail(SCANN_HIP_OUT_OF_RANGE, "Invalid partition ID");
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    DevBuf dq, dl, dout;
    SCANN_TRY(upload(dq, queries, (size_t)nq * q_stride * 4));
    if (leaf_for_query) SCANN_TRY(upload(dl, leaf_for_query, (size_t)nq * 4));
    const size_t ob = (size_t)nq * ix->tx.S * ix->tx.K * 4;
    SCANN_TRY(dout.ensure(ob));
    SCANN_TRY(txh_launch_lut_from_query(ix->tx, dq.as<float>(), nq, q_stride,
                                        leaf_for_query ? dl.as<uint32_t>() : nullptr,
                                        dout.as<float>(), ix->stream));
    SCANN_HIP_CHECK(hipMemcpyAsync(out_lut, dout.p, ob, hipMemcpyDeviceToHost, ix->stream));
    SCANN_HIP_CHECK(hipStreamSynchronize(ix->stream));
    return SCANN_HIP_OK;
}

int scann_hip_adc_distances(scann_hip_index *ix, const float *luts, uint32_t nq, float *out_dist) {
    if (!ix || ix->kind != KIND_TXH) return fail(SCANN_HIP_INVALID_ARGUMENT, "not a tree index");
    if (nq == 0) return SCANN_HIP_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    DevBuf dl, dout;
    SCANN_TRY(upload(dl, luts, (size_t)nq * ix->tx.S * ix->tx.K * 4));
    const size_t ob = (size_t)nq * ix->tx.n_local * 4;
    SCANN_TRY(dout.ensure(ob));
    SCANN_TRY(txh_launch_adc_distances(ix->tx, dl.as<float>(), nq, dout.as<float>(), ix->stream));
    SCANN_HIP_CHECK(hipMemcpyAsync(out_dist, dout.p, ob, hipMemcpyDeviceToHost, ix->stream));
    SCANN_HIP_CHECK(hipStreamSynchronize(ix->stream));
    return SCANN_HIP_OK;
}

int scann_hip_lut16_distances_batch(scann_hip_ctx *ctx, const uint8_t *packed, const uint8_t *lut8,
                                    uint32_t S, uint64_t n, float bias, float multiplier,
                                    float *out) {
    if (!ctx) return fail(SCANN_HIP_INVALID_ARGUMENT, "ctx is null");
    if (n == 0) return SCANN_HIP_OK;
    if (!packed || !lut8 || !out || S == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "null argument");
    SCANN_TRY(set_device(ctx));
    DevBuf dp, dl, dout;
    SCANN_TRY(upload(dp, packed, (size_t)n * ((S + 1) / 2)));
    SCANN_TRY(upload(dl, lut8, (size_t)S * 16));
    SCANN_TRY(dout.ensure((size_t)n * 4));
    SCANN_TRY(launch_lut16_u8_batch(dp.as<uint8_t>(), dl.as<uint8_t>(), S, n, bias, multiplier,
                                    dout.as<float>(), nullptr));
    SCANN_HIP_CHECK(hipMemcpy(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SCANN_HIP_OK;
}

int scann_hip_lut16_quantize(scann_hip_ctx *ctx, const float *tables, uint32_t S, uint8_t *out_lut8,
                             float *out_bias, float *out_multiplier) {
    if (!ctx) return fail(SCANN_HIP_INVALID_ARGUMENT, "ctx is null");
    if (!out_bias || !out_multiplier) return fail(SCANN_HIP_INVALID_ARGUMENT, "null bias/multiplier");
    if (S == 0) {   // lut16_simd.rs:42-49
        *out_bias = 0.0f;
        *out_multiplier = 1.0f;
        return SCANN_HIP_OK;
    }
    if (!tables || !out_lut8) return fail(SCANN_HIP_INVALID_ARGUMENT, "null tables/output");
    SCANN_TRY(set_device(ctx));
    DevBuf dt, dq, dp;
    SCANN_TRY(upload(dt, tables, (size_t)S * 16 * 4));
    SCANN_TRY(dq.ensure((size_t)S * 16));
    SCANN_TRY(dp.ensure(8));
    SCANN_TRY(launch_lut16_quantize(dt.as<float>(), S, dq.as<uint8_t>(), dp.as<float>(), nullptr));
    float bm[2];
    SCANN_HIP_CHECK(hipMemcpy(out_lut8, dq.p, (size_t)S * 16, hipMemcpyDeviceToHost));
    SCANN_HIP_CHECK(hipMemcpy(bm, dp.p, 8, hipMemcpyDeviceToHost));
    *out_bias = bm[0];
    *out_multiplier = bm[1];
    return SCANN_HIP_OK;
}

int scann_hip_fp8_quantize(scann_hip_ctx *ctx, const float *values, uint64_t n, float scale, int format,
                           uint8_t *out_bits) {
    if (!ctx) return fail(SCANN_HIP_INVALID_ARGUMENT, "ctx is null");
    if (format != SCANN_HIP_FP8_E4M3 && format != SCANN_HIP_FP8_E5M2) return fail(SCANN_HIP_INVALID_ARGUMENT, "unknown FP8 format");
    if (n == 0) return SCANN_HIP_OK;
    if (!values || !out_bits) return fail(SCANN_HIP_INVALID_ARGUMENT, "null values/output");
    SCANN_TRY(set_device(ctx));
    DevBuf dv, dout;
    SCANN_TRY(upload(dv, values, (size_t)n * 4));
    SCANN_TRY(dout.ensure((size_t)n));
    SCANN_TRY(launch_fp8_quantize(dv.as<float>(), n, scale, format, dout.as<uint8_t>(), nullptr));
    SCANN_HIP_CHECK(hipMemcpy(out_bits, dout.p, (size_t)n, hipMemcpyDeviceToHost));
    return SCANN_HIP_OK;
}

int scann_hip_fp8_dequantize(scann_hip_ctx *ctx, const uint8_t *bits, uint64_t n, float scale, int format,
                             float *out_values) {
    if (!ctx) return fail(SCANN_HIP_INVALID_ARGUMENT, "ctx is null");
    if (format != SCANN_HIP_FP8_E4M3 && format != SCANN_HIP_FP8_E5M2) return fail(SCANN_HIP_INVALID_ARGUMENT, "unknown FP8 format");
    if (n == 0) return SCANN_HIP_OK;
    if (!bits || !out_values) return fail(SCANN_HIP_INVALID_ARGUMENT, "null bits/output");
    SCANN_TRY(set_device(ctx));
    DevBuf db, dout;
    SCANN_TRY(upload(db, bits, (size_t)n));
    SCANN_TRY(dout.ensure((size_t)n * 4));
    SCANN_TRY(launch_fp8_dequantize(db.as<uint8_t>(), n, scale, format, dout.as<float>(), nullptr));
    SCANN_HIP_CHECK(hipMemcpy(out_values, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SCANN_HIP_OK;
}

int scann_hip_fp8_distances(scann_hip_ctx *ctx, const float *query, uint32_t dim, const uint8_t *database,
                            uint64_t stride, uint64_t num_points, int measure, float *out_distances) {
    if (!ctx) return fail(SCANN_HIP_INVALID_ARGUMENT, "ctx is null");
    if (measure != SCANN_HIP_SQUARED_L2 && measure != SCANN_HIP_DOT_PRODUCT)
        return fail(SCANN_HIP_UNIMPLEMENTED, "the reference's FP8 kernels are squared L2 and dot product");
    if (num_points == 0) return SCANN_HIP_OK;
    if (!query || !database || !out_distances || dim == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "bad argument");
    if (stride < dim) return fail(SCANN_HIP_INVALID_ARGUMENT, "stride < dim");
    if (dim > 32768) return fail(SCANN_HIP_RESOURCE_EXHAUSTED, "dim > 32768");
    SCANN_TRY(set_device(ctx));
    DevBuf dq, ddb, dout;
    SCANN_TRY(upload(dq, query, (size_t)dim * 4));
    SCANN_TRY(upload(ddb, database, (size_t)num_points * stride));
    SCANN_TRY(dout.ensure((size_t)num_points * 4));
    SCANN_TRY(launch_fp8_one_to_many(dq.as<float>(), dim, ddb.as<uint8_t>(), stride, num_points,
                                     measure == SCANN_HIP_DOT_PRODUCT ? 1 : 0, dout.as<float>(), nullptr));
    SCANN_HIP_CHECK(hipMemcpy(out_distances, dout.p, (size_t)num_points * 4, hipMemcpyDeviceToHost));
    return SCANN_HIP_OK;
}

int scann_hip_encode(scann_hip_ctx *ctx, const float *codebook, uint32_t S, uint32_t K, uint32_t dsub,
                     const float *rows, uint64_t n, uint32_t stride, const float *centers,
                     const uint32_t *leaf_of_row, uint8_t *out_codes) {
    if (!ctx) return fail(SCANN_HIP_INVALID_ARGUMENT, "ctx is null");
    if (n == 0) return SCANN_HIP_OK;
    if (!codebook || !rows || !out_codes || S == 0 || K == 0 || K > 256 || dsub == 0)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "bad argument");
    if ((centers == nullptr) != (leaf_of_row == nullptr))
        return fail(SCANN_HIP_INVALID_ARGUMENT, "centers and leaf_of_row go together");
    if (stride < S * dsub) return fail(SCANN_HIP_INVALID_ARGUMENT, "stride < dim");
    SCANN_TRY(set_device(ctx));
    DevBuf dcb, drows, dcen, dleaf, dout;
    SCANN_TRY(upload(dcb, codebook, (size_t)S * K * dsub * 4));
    SCANN_TRY(upload(drows, rows, (size_t)n * stride * 4));
    if (centers) {
        uint32_t maxleaf = 0;
        for (uint64_t i = 0; i < n; ++i) maxleaf = std::max(maxleaf, leaf_of_row[i]);
        SCANN_TRY(upload(dcen, centers, (size_t)(maxleaf + 1) * S * dsub * 4));
        SCANN_TRY(upload(dleaf, leaf_of_row, (size_t)n * 4));
    }
    SCANN_TRY(dout.ensure((size_t)n * S));
    SCANN_TRY(launch_encode(dcb.as<float>(), S, K, dsub, drows.as<float>(), n, stride,
                            centers ? dcen.as<float>() : nullptr,
                            centers ? dleaf.as<uint32_t>() : nullptr, dout.as<uint8_t>(), nullptr));
    SCANN_HIP_CHECK(hipMemcpy(out_codes, dout.p, (size_t)n * S, hipMemcpyDeviceToHost));
    return SCANN_HIP_OK;
}

int scann_hip_bf_distances(scann_hip_index *ix, const float *queries, uint32_t nq, uint32_t q_stride,
                           float *out) {
    if (!ix || ix->kind != KIND_BF) return fail(SCANN_HIP_INVALID_ARGUMENT, "not a brute-force index");
    if (nq == 0 || ix->bf.n == 0) return SCANN_HIP_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    SCANN_TRY(claim_primary_workspace(ix));
    return bf_distances_host(ix->bf, ix->bfw, queries, nq, q_stride, out, ix->stream);
}

int scann_hip_bf_search_radius(scann_hip_index *ix, const float *query, uint32_t q_dim, float radius,
                               uint32_t *out_idx, float *out_dist, uint64_t capacity,
                               uint64_t *out_count) {
    if (!ix || ix->kind != KIND_BF) return fail(SCANN_HIP_INVALID_ARGUMENT, "not a brute-force index");
    if (!out_count) return fail(SCANN_HIP_INVALID_ARGUMENT, "null out_count");
    *out_count = 0;
    if (ix->bf.n == 0) return SCANN_HIP_OK;   // searcher.rs:143-145
    if (!query || q_dim != ix->bf.dim)        // searcher.rs:148-152
        return fail(SCANN_HIP_INVALID_ARGUMENT, "Query dimensionality does not match dataset");
    if (capacity && (!out_idx || !out_dist)) return fail(SCANN_HIP_INVALID_ARGUMENT, "null outputs");
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    SCANN_TRY(claim_primary_workspace(ix));
    return bf_search_radius_host(ix->bf, ix->bfw, query, q_dim, radius, out_idx, out_dist, capacity,
                                 out_count, ix->stream);
}

int scann_hip_bf_assign_nearest(scann_hip_index *ix, const float *centers, uint32_t num_centers,
                                uint32_t *out_assign, float *out_dist) {
    if (!ix || ix->kind != KIND_BF) return fail(SCANN_HIP_INVALID_ARGUMENT, "not a brute-force index");
    if (!centers || num_centers == 0 || !out_assign)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "null/empty centres or output");
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    return bf_assign_nearest_host(ix->bf, centers, num_centers, out_assign, out_dist, ix->stream);
}

static int kmeans_args(scann_hip_index *ix, uint32_t col_offset, uint32_t sub_dim, uint32_t k, const void *c) {
    if (!ix || ix->kind != KIND_BF) return fail(SCANN_HIP_INVALID_ARGUMENT, "not a brute-force index");
    if (ix->bf.n == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "Cannot cluster empty dataset");   // kmeans.rs:167-169
    if (!c || k == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "null centres / zero clusters");
    if (sub_dim == 0 || (uint64_t)col_offset + sub_dim > ix->bf.dim)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "column window outside the rows");
    return SCANN_HIP_OK;
}

int scann_hip_kmeans_init_pp(scann_hip_index *ix, uint32_t col_offset, uint32_t sub_dim, uint32_t k,
                             uint64_t seed, uint32_t simd_threshold, float *centers_out) {
    SCANN_TRY(kmeans_args(ix, col_offset, sub_dim, k, centers_out));
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    return bf_kmeans_init_pp_host(ix->bf, col_offset, sub_dim, k, seed, simd_threshold, centers_out, ix->stream);
}

int scann_hip_kmeans_lloyd(scann_hip_index *ix, uint32_t col_offset, uint32_t sub_dim, float *centers,
                           uint32_t k, uint32_t max_iterations, double convergence_threshold,
                           uint32_t simd_threshold, uint32_t *out_assign, uint32_t *out_sizes,
                           double *out_inertia, uint32_t *out_iterations, int *out_converged) {
    SCANN_TRY(kmeans_args(ix, col_offset, sub_dim, k, centers));
    std::lock_guard<std::mutex> lock(ix->mu);
    SCANN_TRY(set_device(ix->ctx));
    return bf_kmeans_lloyd_host(ix->bf, col_offset, sub_dim, centers, k, max_iterations,
                                convergence_threshold, simd_threshold, out_assign, out_sizes, out_inertia,
                                out_iterations, out_converged, ix->stream);
}

}  // extern "C"
