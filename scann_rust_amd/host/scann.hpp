// scann.hpp -- C++ host-side mirror of the reference crate's API for the hot path, on top
// of the C ABI (include/scann_hip.h).  The reference is Rust; the build image has no Rust
// toolchain, so this header plays the role of the Rust wrappers shown in INTEGRATION.md:
// same type / method names, argument meaning and error behaviour.
//
//   scann::DenseDataset                 data_format/dataset.rs:46-280
//   scann::BruteForceSearcher           brute_force/searcher.rs:18-208
//   scann::AsymmetricHasher(+Config)    hashes/hasher.rs:19-258
//   scann::TreeXHybridSearcher(+Config) tree_x_hybrid/mod.rs:23-418
//   scann::ScannBuilder / scann::Scann  scann.rs:35-56, 364-426
//   scann::ScannError / ErrorCode       error.rs:10-147
//
// All distance computation and top-k selection runs in libscann_hip.so on the GPU.  Index
// TRAINING (k-means, PQ codebooks) is host-side C++ here with its own RNG: the reference's
// StdRng streams are not reproducible and the trained index is an input to
// the search path.  Encoding uses the GPU (scann_hip_encode, bit-exact with Codebook::encode).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/scann_hip.h"

namespace scann {

enum class ErrorCode : int {   // error.rs:10-45, declaration order
    Ok = 0, Cancelled, Unknown, InvalidArgument, DeadlineExceeded, NotFound, AlreadyExists,
    PermissionDenied, ResourceExhausted, FailedPrecondition, Aborted, OutOfRange, Unimplemented,
    Internal, Unavailable, DataLoss, Unauthenticated
};

struct ScannError : std::runtime_error {   // error.rs:73-147
    ErrorCode code;
    ScannError(ErrorCode c, const std::string &m) : std::runtime_error(m), code(c) {}
    static ScannError invalid_argument(const std::string &m) { return {ErrorCode::InvalidArgument, m}; }
    static ScannError failed_precondition(const std::string &m) { return {ErrorCode::FailedPrecondition, m}; }
};

inline void check(int status) {
    if (status != SCANN_HIP_OK) throw ScannError(static_cast<ErrorCode>(status), scann_hip_last_error());
}

enum class DistanceMeasure : int { SquaredL2 = 0, L2 = 1, DotProduct = 2, L1 = 3, Cosine = 4 };  // distance_measures/mod.rs:32-66 (the dense measures of the hot path)

using DatapointIndex = uint32_t;                                   // types.rs:10
using NNResultsVector = std::vector<std::pair<DatapointIndex, float>>;  // types.rs:17-20

// One context per device, shared by every searcher of the process.
inline scann_hip_ctx *context(int device = 0) {
    static scann_hip_ctx *ctxs[64] = {};
    if (device < 0 || device >= 64) throw ScannError::invalid_argument("device id");
    if (!ctxs[device]) check(scann_hip_init(device, &ctxs[device]));
    return ctxs[device];
}

// quantization/fp8.rs: the reference's FP8 codec and quantizer on the device (its own bit-level conversion,
// restated bit for bit by scann_hip_fp8_quantize / _dequantize; see include/scann_hip.h).
enum class Fp8Format : int { E4M3 = 0, E5M2 = 1 };               // fp8.rs:8-19

struct Fp8Config {                                               // fp8.rs:22-62
    Fp8Format format = Fp8Format::E4M3;
    float scale = 1.0f;
    static Fp8Config e4m3() { return Fp8Config{}; }
    static Fp8Config e5m2() { return Fp8Config{Fp8Format::E5M2, 1.0f}; }
    Fp8Config with_scale(float s) const { return Fp8Config{format, s}; }
};

class Fp8Quantizer {                                             // fp8.rs:206-273
public:
    explicit Fp8Quantizer(Fp8Config config = Fp8Config{}, int device = 0) : config_(config), device_(device) {}
    static Fp8Quantizer e4m3() { return Fp8Quantizer(Fp8Config::e4m3()); }
    static Fp8Quantizer e5m2() { return Fp8Quantizer(Fp8Config::e5m2()); }
    Fp8Format format() const { return config_.format; }
    float scale() const { return config_.scale; }
    size_t bits() const { return 8; }
    void calibrate_scale(float max_abs_value) {                  // :238-244
        const float fp8_max = config_.format == Fp8Format::E4M3 ? 448.0f : 57344.0f;
        config_.scale = fp8_max / std::max(max_abs_value, 1e-10f);
    }
    std::vector<uint8_t> quantize(const std::vector<float> &values) const {      // Quantizer::quantize
        std::vector<uint8_t> out(values.size());
        check(scann_hip_fp8_quantize(context(device_), values.data(), values.size(), config_.scale, (int)config_.format,
                                     out.data()));
        return out;
    }
    std::vector<float> dequantize(const std::vector<uint8_t> &bits) const {      // Quantizer::dequantize
        std::vector<float> out(bits.size());
        check(scann_hip_fp8_dequantize(context(device_), bits.data(), bits.size(), config_.scale, (int)config_.format,
                                       out.data()));
        return out;
    }

private:
    Fp8Config config_;
    int device_;
};

// one_to_many_fp8_float_{squared_l2,dot_product} (distance_measures/one_to_many_asymmetric.rs:327-377)
inline std::vector<float> one_to_many_fp8(const std::vector<float> &query, const std::vector<uint8_t> &database,
                                          size_t stride, size_t num_points, DistanceMeasure measure, int device = 0) {
    std::vector<float> out(num_points);
    check(scann_hip_fp8_distances(context(device), query.data(), (uint32_t)query.size(), database.data(), stride,
                                  num_points, (int)measure, out.data()));
    return out;
}

// data_format/dataset.rs: one row-major buffer, stride = align_up(dim, 16 floats).
class DenseDataset {
public:
    DenseDataset() = default;
    static DenseDataset from_vecs(const std::vector<std::vector<float>> &vecs) {  // :99-123
        DenseDataset d;
        if (vecs.empty()) return d;
        d.dim_ = static_cast<uint32_t>(vecs[0].size());
        d.n_ = vecs.size();
        d.stride_ = scann_hip_compute_stride(d.dim_);
        d.data_.assign(d.n_ * d.stride_, 0.0f);
        for (size_t i = 0; i < d.n_; ++i)
            std::memcpy(&d.data_[i * d.stride_], vecs[i].data(), std::min<size_t>(vecs[i].size(), d.dim_) * 4);
        return d;
    }
    static DenseDataset from_flat(const std::vector<float> &flat, uint32_t dim) {  // :126-174
        if (dim == 0) throw ScannError::invalid_argument("Dimensionality cannot be 0");
        if (flat.size() % dim) throw ScannError::invalid_argument("Data length is not a multiple of dimensionality");
        DenseDataset d;
        d.dim_ = dim;
        d.n_ = flat.size() / dim;
        d.stride_ = scann_hip_compute_stride(dim);
        d.data_.assign(d.n_ * d.stride_, 0.0f);
        for (size_t i = 0; i < d.n_; ++i) std::memcpy(&d.data_[i * d.stride_], &flat[i * dim], dim * 4);
        return d;
    }
    bool is_empty() const { return n_ == 0; }
    size_t size() const { return n_; }
    uint32_t dimensionality() const { return dim_; }
    uint32_t stride() const { return stride_; }
    const float *raw_data() const { return data_.data(); }
    const float *get(size_t i) const { return &data_[i * stride_]; }

private:
    std::vector<float> data_;
    size_t n_ = 0;
    uint32_t dim_ = 0, stride_ = 0;
};

// ---- restricts (restricts/mod.rs:16-45, restricts/allowlist.rs:8-187) ----------------------------
struct RestrictFilter {
    virtual ~RestrictFilter() = default;
    virtual bool is_allowed(DatapointIndex index) const = 0;
    // Words of an allow-bitmap covering datapoints [0, n): bit i = is_allowed(i).
    virtual std::vector<uint64_t> to_bitmap(size_t n) const {
        std::vector<uint64_t> w((n + 63) / 64, 0);
        for (size_t i = 0; i < n; ++i)
            if (is_allowed((DatapointIndex)i)) w[i >> 6] |= 1ull << (i & 63);
        return w;
    }
};

class RestrictAllowlist : public RestrictFilter {   // allowlist.rs:8-105
public:
    explicit RestrictAllowlist(size_t capacity) : bits_((capacity + 63) / 64, 0), capacity_(capacity) {}
    static RestrictAllowlist from_indices(const std::vector<DatapointIndex> &indices, size_t capacity) {
        RestrictAllowlist a(capacity);
        for (auto i : indices) a.add(i);
        return a;
    }
    void add(DatapointIndex i) {
        if (i < capacity_ && !get(i)) { bits_[i >> 6] |= 1ull << (i & 63); ++count_; }
    }
    void remove(DatapointIndex i) {
        if (i < capacity_ && get(i)) { bits_[i >> 6] &= ~(1ull << (i & 63)); --count_; }
    }
    void clear() { std::fill(bits_.begin(), bits_.end(), 0); count_ = 0; }
    size_t capacity() const { return capacity_; }
    size_t num_allowed() const { return count_; }
    bool is_allowed(DatapointIndex i) const override { return i < capacity_ && get(i); }
    std::vector<uint64_t> to_bitmap(size_t n) const override {
        std::vector<uint64_t> w = bits_;
        w.resize((n + 63) / 64, 0);
        if (n & 63) w.back() &= (1ull << (n & 63)) - 1;
        return w;
    }

private:
    bool get(DatapointIndex i) const { return (bits_[i >> 6] >> (i & 63)) & 1ull; }
    std::vector<uint64_t> bits_;
    size_t capacity_, count_ = 0;
};

class RestrictDenylist : public RestrictFilter {    // allowlist.rs:107-187
public:
    explicit RestrictDenylist(size_t capacity) : deny_(capacity) {}
    static RestrictDenylist from_indices(const std::vector<DatapointIndex> &indices, size_t capacity) {
        RestrictDenylist d(capacity);
        for (auto i : indices) d.deny(i);
        return d;
    }
    void deny(DatapointIndex i) { deny_.add(i); }
    void allow(DatapointIndex i) { deny_.remove(i); }
    void clear() { deny_.clear(); }
    bool is_allowed(DatapointIndex i) const override { return !deny_.is_allowed(i); }

private:
    RestrictAllowlist deny_;
};

namespace detail {

struct IndexHandle {
    scann_hip_index *h = nullptr;
    IndexHandle() = default;
    IndexHandle(const IndexHandle &) = delete;
    IndexHandle &operator=(const IndexHandle &) = delete;
    IndexHandle(IndexHandle &&o) noexcept : h(o.h) { o.h = nullptr; }
    ~IndexHandle() { if (h) scann_hip_index_destroy(h); }
};

inline std::vector<NNResultsVector> run_search(scann_hip_index *h, const float *q, uint32_t nq,
                                               uint32_t q_stride, uint32_t q_dim, uint32_t k,
                                               const scann_hip_search_opts *opts) {
    std::vector<uint32_t> idx((size_t)nq * std::max(1u, k)), cnt(nq);
    std::vector<float> dist((size_t)nq * std::max(1u, k));
    check(scann_hip_search_batched(h, q, nq, q_stride, q_dim, k, opts, idx.data(), dist.data(), cnt.data()));
    std::vector<NNResultsVector> out(nq);
    for (uint32_t i = 0; i < nq; ++i)
        for (uint32_t j = 0; j < cnt[i]; ++j) out[i].emplace_back(idx[(size_t)i * k + j], dist[(size_t)i * k + j]);
    return out;
}

// Searcher::search_batched_with_params (searcher.rs:148-186): one num_neighbors per query
inline std::vector<NNResultsVector> run_search_params(scann_hip_index *h, const float *q, uint32_t nq,
                                                      uint32_t q_stride, uint32_t q_dim, const std::vector<uint32_t> &ks,
                                                      const scann_hip_search_opts *opts) {
    uint32_t pitch = 1;
    for (uint32_t k : ks) pitch = std::max(pitch, k);
    std::vector<uint32_t> idx((size_t)nq * pitch), cnt(nq);
    std::vector<float> dist((size_t)nq * pitch);
    check(scann_hip_search_batched_params(h, q, nq, q_stride, q_dim, ks.data(), opts, pitch, idx.data(), dist.data(), cnt.data()));
    std::vector<NNResultsVector> out(nq);
    for (uint32_t i = 0; i < nq; ++i)
        for (uint32_t j = 0; j < cnt[i]; ++j) out[i].emplace_back(idx[(size_t)i * pitch + j], dist[(size_t)i * pitch + j]);
    return out;
}

inline std::vector<float> flatten(const std::vector<std::vector<float>> &qs, uint32_t *dim_out) {
    uint32_t d = qs.empty() ? 0 : (uint32_t)qs[0].size();
    for (auto &q : qs)
        if (q.size() != d) throw ScannError::invalid_argument("Query dimensionality mismatch");
    std::vector<float> flat((size_t)qs.size() * std::max(1u, d));
    for (size_t i = 0; i < qs.size(); ++i) std::memcpy(&flat[i * d], qs[i].data(), d * 4);
    *dim_out = d;
    return flat;
}

// splitmix64: the documented counter-based generator of this build (not rand::StdRng).
inline uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// KMeans::fit (trees/kmeans.rs:166-263) on the GPU over the rows of the brute-force index `bf`,
// restricted to columns [c0, c0 + d): k-means++ seeding (scann_hip_kmeans_init_pp, splitmix64 stream)
// then the Lloyd loop of fit_single (scann_hip_kmeans_lloyd: assignment with the lowest index on ties
// -- sequential-scalar distances below KMeansConfig.simd_threshold dims, the AVX2 order from there
// on -- f64 centre update in datapoint order, the reference's convergence test).
constexpr uint32_t kKMeansSimdThreshold = 128;   // KMeansConfig::default (trees/kmeans.rs:59)
inline void kmeans_gpu(scann_hip_index *bf, size_t n, size_t c0, size_t d, size_t k, size_t max_iterations,
                       double convergence_threshold, uint64_t seed, std::vector<float> &centers,
                       std::vector<uint32_t> &assign) {
    k = std::min(k, n);   // kmeans.rs:171-175: at most n clusters
    centers.assign(k * d, 0.0f);
    assign.assign(n, 0);
    check(scann_hip_kmeans_init_pp(bf, (uint32_t)c0, (uint32_t)d, (uint32_t)k, seed, kKMeansSimdThreshold,
                                   centers.data()));
    check(scann_hip_kmeans_lloyd(bf, (uint32_t)c0, (uint32_t)d, centers.data(), (uint32_t)k,
                                 (uint32_t)max_iterations, convergence_threshold, kKMeansSimdThreshold,
                                 assign.data(), nullptr, nullptr, nullptr, nullptr));
}

// Codebook::train (hashes/codebook.rs:146-202): per-subspace k-means with seed + s, on the GPU over
// the column windows of `bf` (an index holding the training rows).
inline std::vector<float> train_codebook(scann_hip_index *bf, size_t n, uint32_t dim, uint32_t S, uint32_t K,
                                         uint64_t seed, size_t max_iterations, double convergence_threshold) {
    if (n == 0) throw ScannError::invalid_argument("Cannot train on empty dataset");
    if (dim % S != 0)   // codebook.rs:154-159
        throw ScannError::invalid_argument("Dimensionality " + std::to_string(dim) +
                                           " must be divisible by num_subspaces " + std::to_string(S));
    const uint32_t dsub = dim / S;
    std::vector<float> cb((size_t)S * K * dsub, 0.0f), centers;
    std::vector<uint32_t> assign;
    for (uint32_t s = 0; s < S; ++s) {
        kmeans_gpu(bf, n, (size_t)s * dsub, dsub, K, max_iterations, convergence_threshold, seed + s, centers,
                   assign);
        const size_t got = centers.size() / dsub;
        for (uint32_t c = 0; c < K; ++c)
            std::memcpy(&cb[((size_t)s * K + c) * dsub], &centers[std::min<size_t>(c, got - 1) * dsub], dsub * 4);
    }
    return cb;
}

// TreePartitioner::build (tree_partitioner.rs:48-98): flat k-means with seed 42 on the GPU, then the
// CSR leaf lists (ascending datapoint index inside each leaf, :84-89).
inline void build_partition(const DenseDataset &ds, size_t num_partitions, size_t iterations, int device,
                            std::vector<float> &centers, std::vector<uint32_t> &leaf_off,
                            std::vector<uint32_t> &leaf_ids);

}  // namespace detail

// ---- BruteForceSearcher (brute_force/searcher.rs:18-208) --------------------------------
class BruteForceSearcher {
public:
    BruteForceSearcher(std::shared_ptr<DenseDataset> dataset, DistanceMeasure measure, int device = 0)
        : dataset_(std::move(dataset)), measure_(measure) {
        check(scann_hip_bf_create(context(device), dataset_->raw_data(), dataset_->size(),
                                  dataset_->dimensionality(), dataset_->stride(), (int)measure, &ix_.h));
    }
    BruteForceSearcher(DenseDataset dataset, DistanceMeasure measure, int device = 0)
        : BruteForceSearcher(std::make_shared<DenseDataset>(std::move(dataset)), measure, device) {}

    NNResultsVector search(const std::vector<float> &query, size_t k) const {   // :77-93
        return detail::run_search(ix_.h, query.data(), 1, (uint32_t)query.size(), (uint32_t)query.size(),
                                  (uint32_t)k, nullptr)[0];
    }
    std::vector<NNResultsVector> search_batched(const std::vector<std::vector<float>> &queries, size_t k) const {
        if (queries.empty()) return {};                                          // :175-177
        uint32_t d;
        auto flat = detail::flatten(queries, &d);
        return detail::run_search(ix_.h, flat.data(), (uint32_t)queries.size(), d, d, (uint32_t)k, nullptr);
    }
    NNResultsVector search_radius(const std::vector<float> &query, float radius) const {   // :142-167
        if (dataset_->size() == 0) return {};
        std::vector<uint32_t> idx(256);
        std::vector<float> dist(256);
        uint64_t found = 0;
        for (;;) {
            check(scann_hip_bf_search_radius(ix_.h, query.data(), (uint32_t)query.size(), radius, idx.data(),
                                             dist.data(), idx.size(), &found));
            if (found <= idx.size()) break;
            idx.resize(found);
            dist.resize(found);
        }
        NNResultsVector r(found);
        for (size_t i = 0; i < found; ++i) r[i] = {idx[i], dist[i]};
        return r;
    }
    size_t dataset_size() const { return dataset_->size(); }
    uint64_t dimensionality() const { return dataset_->dimensionality(); }
    DistanceMeasure distance_measure() const { return measure_; }
    const DenseDataset &dataset() const { return *dataset_; }

private:
    std::shared_ptr<DenseDataset> dataset_;
    DistanceMeasure measure_;
    detail::IndexHandle ix_;
};

// ---- AsymmetricHasher (hashes/hasher.rs) ---------------------------------------------------
struct AsymmetricHasherConfig {      // hasher.rs:19-69 (default 256 x 8: byte codes; K <= 16 takes the LUT16 path)
    size_t num_codes = 256, num_subspaces = 8;
    uint64_t seed = 42;
    bool has_seed = false;
    size_t training_iterations = 25;       // CodebookConfig.max_iterations
    double convergence_threshold = 1e-5;   // CodebookConfig.convergence_threshold
    AsymmetricHasherConfig() = default;
    AsymmetricHasherConfig(size_t codes, size_t subspaces) : num_codes(codes), num_subspaces(subspaces) {}
    AsymmetricHasherConfig with_seed(uint64_t s) const { auto c = *this; c.seed = s; c.has_seed = true; return c; }
};

class AsymmetricHasher {
public:
    explicit AsymmetricHasher(AsymmetricHasherConfig config, int device = 0) : config_(config), device_(device) {}

    void build(DenseDataset dataset) {            // hasher.rs:109-134
        build_impl(dataset, true);
        dataset_ = std::make_shared<DenseDataset>(std::move(dataset));
    }
    void build_no_store(const DenseDataset &dataset) { build_impl(dataset, false); }   // :137-159

    NNResultsVector search(const std::vector<float> &query, size_t k) const {          // :162-185
        if (!ix_.h) return {};
        scann_hip_search_opts o;
        scann_hip_search_opts_default(&o);
        o.exact_reorder = 0;
        return detail::run_search(ix_.h, query.data(), 1, (uint32_t)query.size(), (uint32_t)query.size(),
                                  (uint32_t)k, &o)[0];
    }
    NNResultsVector search_with_reordering(const std::vector<float> &query, size_t k, size_t pre_reorder_k) const {
        if (!stored_) throw ScannError::failed_precondition("Dataset not stored");    // :194-197
        scann_hip_search_opts o;
        scann_hip_search_opts_default(&o);
        o.pre_reorder_k = (uint32_t)pre_reorder_k;
        return detail::run_search(ix_.h, query.data(), 1, (uint32_t)query.size(), (uint32_t)query.size(),
                                  (uint32_t)k, &o)[0];
    }
    std::vector<NNResultsVector> search_batched(const std::vector<std::vector<float>> &queries, size_t k) const {
        if (queries.empty() || !ix_.h) return std::vector<NNResultsVector>(queries.size());
        uint32_t d;
        auto flat = detail::flatten(queries, &d);
        scann_hip_search_opts o;
        scann_hip_search_opts_default(&o);
        o.exact_reorder = 0;
        return detail::run_search(ix_.h, flat.data(), (uint32_t)queries.size(), d, d, (uint32_t)k, &o);
    }
    size_t num_datapoints() const { return n_; }
    size_t dimensionality() const { return dim_; }
    const std::vector<float> &codebook() const { return codebook_; }
    const std::vector<uint8_t> &encoded_database() const { return codes_; }

private:
    void build_impl(const DenseDataset &ds, bool store) {
        if (ds.is_empty()) throw ScannError::invalid_argument("Cannot build from empty dataset");  // :110-112
        dim_ = ds.dimensionality();
        n_ = ds.size();
        const uint32_t S = (uint32_t)config_.num_subspaces, K = (uint32_t)config_.num_codes;
        {
            detail::IndexHandle tmp;   // training rows resident on the GPU for the k-means passes
            check(scann_hip_bf_create(context(device_), ds.raw_data(), n_, (uint32_t)dim_, ds.stride(),
                                      SCANN_HIP_SQUARED_L2, &tmp.h));
            codebook_ = detail::train_codebook(tmp.h, n_, (uint32_t)dim_, S, K, config_.seed,
                                               config_.training_iterations, config_.convergence_threshold);
        }
        codes_.assign(n_ * S, 0);
        check(scann_hip_encode(context(device_), codebook_.data(), S, K, (uint32_t)dim_ / S, ds.raw_data(), n_,
                               ds.stride(), nullptr, nullptr, codes_.data()));
        scann_hip_txh_desc d{};
        d.data = store ? ds.raw_data() : nullptr;
        d.n_rows = n_;
        d.dim = (uint32_t)dim_;
        d.stride = ds.stride();
        d.n_local = n_;
        d.codebook = codebook_.data();
        d.num_subspaces = S;
        d.num_codes = K;
        d.dims_per_subspace = (uint32_t)dim_ / S;
        d.codes = codes_.data();
        d.partitions_to_search = 1;
        d.pre_reorder_multiplier = 1.0f;
        if (ix_.h) { scann_hip_index_destroy(ix_.h); ix_.h = nullptr; }
        check(scann_hip_txh_create(context(device_), &d, &ix_.h));
        stored_ = store;
    }
    AsymmetricHasherConfig config_;
    int device_;
    std::shared_ptr<DenseDataset> dataset_;
    std::vector<float> codebook_;
    std::vector<uint8_t> codes_;
    size_t n_ = 0, dim_ = 0;
    bool stored_ = false;
    detail::IndexHandle ix_;
};

// ---- TreeXHybridSearcher (tree_x_hybrid/mod.rs) ------------------------------------------------
struct TreeXHybridConfig {           // mod.rs:23-78
    size_t num_partitions = 100, partitions_to_search = 10;
    AsymmetricHasherConfig hash_config;
    bool use_residuals = true;
    float pre_reorder_multiplier = 3.0f;
    bool parallel_partition_search = true;   // no effect: every leaf scan is parallel on the GPU
    size_t kmeans_iterations = 100;   // tree_partitioner.rs:72 with_max_iterations(100)
    TreeXHybridConfig() = default;
    TreeXHybridConfig(size_t parts, size_t to_search) : num_partitions(parts), partitions_to_search(to_search) {}
    TreeXHybridConfig with_hash(AsymmetricHasherConfig c) const { auto t = *this; t.hash_config = c; return t; }
    TreeXHybridConfig with_residuals(bool r) const { auto t = *this; t.use_residuals = r; return t; }
    TreeXHybridConfig with_pre_reorder(float m) const { auto t = *this; t.pre_reorder_multiplier = m; return t; }
};

class TreeXHybridSearcher {
public:
    explicit TreeXHybridSearcher(TreeXHybridConfig config, int device = 0) : config_(config), device_(device) {}

    void build(DenseDataset dataset) {     // mod.rs:131-209
        if (dataset.is_empty()) throw ScannError::invalid_argument("Cannot build from empty dataset");
        dataset_ = std::make_shared<DenseDataset>(std::move(dataset));
        const DenseDataset &ds = *dataset_;
        const size_t n = ds.size(), dim = ds.dimensionality();
        const uint32_t S = (uint32_t)config_.hash_config.num_subspaces, K = (uint32_t)config_.hash_config.num_codes;
        if (dim % S != 0)
            throw ScannError::invalid_argument("Dimensionality " + std::to_string(dim) +
                                               " must be divisible by num_subspaces " + std::to_string(S));
        detail::build_partition(ds, config_.num_partitions, config_.kmeans_iterations, device_, centers_, leaf_off_,
                                leaf_ids_);
        const uint32_t L = (uint32_t)(centers_.size() / dim);
        // residual rows in CSR order (mod.rs:177-189), codebook on residuals (:151-158)
        std::vector<float> rows(n * dim);
        std::vector<uint32_t> leaf_of_row(n);
        for (uint32_t l = 0; l < L; ++l)
            for (uint32_t r = leaf_off_[l]; r < leaf_off_[l + 1]; ++r) {
                const float *x = ds.get(leaf_ids_[r]);
                leaf_of_row[r] = l;
                for (size_t j = 0; j < dim; ++j)
                    rows[r * dim + j] = config_.use_residuals ? x[j] - centers_[l * dim + j] : x[j];
            }
        {
            detail::IndexHandle tmp;   // residual rows resident on the GPU for the codebook k-means
            check(scann_hip_bf_create(context(device_), rows.data(), n, (uint32_t)dim, (uint32_t)dim,
                                      SCANN_HIP_SQUARED_L2, &tmp.h));
            codebook_ = detail::train_codebook(tmp.h, n, (uint32_t)dim, S, K, config_.hash_config.seed,
                                               config_.hash_config.training_iterations,
                                               config_.hash_config.convergence_threshold);
        }
        codes_.assign(n * S, 0);
        check(scann_hip_encode(context(device_), codebook_.data(), S, K, (uint32_t)dim / S, rows.data(), n,
                               (uint32_t)dim, nullptr, nullptr, codes_.data()));
        n_ = n;
        dim_ = dim;
        const scann_hip_txh_desc d = desc();
        if (ix_.h) { scann_hip_index_destroy(ix_.h); ix_.h = nullptr; }
        check(scann_hip_txh_create(context(device_), &d, &ix_.h));
    }

    NNResultsVector search(const std::vector<float> &query, size_t k) const {   // mod.rs:240-294
        if (!ix_.h) throw ScannError::failed_precondition("Partitioner not built");
        return detail::run_search(ix_.h, query.data(), 1, (uint32_t)query.size(), (uint32_t)query.size(),
                                  (uint32_t)k, nullptr)[0];
    }
    // mod.rs:245-294: disallowed datapoints are skipped before scoring.  The filter is
    // materialised once into an allow-bitmap that the scan kernels test.
    NNResultsVector search_with_filter(const std::vector<float> &query, size_t k,
                                       const RestrictFilter *filter) const {
        if (!filter) return search(query, k);
        if (!ix_.h) throw ScannError::failed_precondition("Partitioner not built");
        const auto bits = filter->to_bitmap(num_datapoints());
        scann_hip_search_opts o;
        scann_hip_search_opts_default(&o);
        o.allow_bitmap = bits.data();
        o.allow_bitmap_bits = bits.size() * 64;
        return detail::run_search(ix_.h, query.data(), 1, (uint32_t)query.size(), (uint32_t)query.size(),
                                  (uint32_t)k, &o)[0];
    }
    std::vector<NNResultsVector> search_batched(const std::vector<std::vector<float>> &queries, size_t k) const {
        if (!ix_.h) throw ScannError::failed_precondition("Partitioner not built");
        if (queries.empty()) return {};
        uint32_t d;
        auto flat = detail::flatten(queries, &d);
        return detail::run_search(ix_.h, flat.data(), (uint32_t)queries.size(), d, d, (uint32_t)k, nullptr);
    }
    // Searcher::search_batched_with_params (tree_x_hybrid/mod.rs:399-409): params[i].num_neighbors per query
    std::vector<NNResultsVector> search_batched_with_params(const std::vector<std::vector<float>> &queries,
                                                            const std::vector<uint32_t> &num_neighbors) const {
        if (!ix_.h) throw ScannError::failed_precondition("Partitioner not built");
        if (queries.size() != num_neighbors.size()) throw ScannError::invalid_argument("one SearchParameters per query");
        if (queries.empty()) return {};
        uint32_t d;
        auto flat = detail::flatten(queries, &d);
        return detail::run_search_params(ix_.h, flat.data(), (uint32_t)queries.size(), d, d, num_neighbors, nullptr);
    }
    size_t num_partitions() const { return L_ ? L_ : (leaf_off_.empty() ? 0 : leaf_off_.size() - 1); }
    size_t num_datapoints() const { return n_; }
    size_t dataset_size() const { return num_datapoints(); }
    uint64_t dimensionality() const { return dim_; }
    const TreeXHybridConfig &config() const { return config_; }

    // Index files (include/scann_hip.h "index files"; the reference has no save/load): the built
    // index as one SCANNIDX container, and a searcher over a file that is mmap'ed and uploaded
    // without a second host copy.
    void save(const std::string &path) const {
        if (!ix_.h || !dataset_) throw ScannError::failed_precondition("Partitioner not built");
        const scann_hip_txh_desc d = desc();
        check(scann_hip_txh_write_file(path.c_str(), &d));
    }
    static TreeXHybridSearcher load(const std::string &path, int device = 0) {
        scann_hip_file_info info;
        check(scann_hip_index_file_info(path.c_str(), &info));
        if (info.kind != 1 || info.num_partitions == 0 || info.num_subspaces == 0)
            throw ScannError::invalid_argument(path + " does not hold a Tree-X-Hybrid index");
        TreeXHybridConfig c(info.num_partitions, info.partitions_to_search);
        c.hash_config = AsymmetricHasherConfig(info.num_codes, info.num_subspaces);
        c.use_residuals = info.use_residuals != 0;
        c.pre_reorder_multiplier = info.pre_reorder_multiplier;
        TreeXHybridSearcher s(c, device);
        check(scann_hip_index_load_file(context(device), path.c_str(), &s.ix_.h));
        s.n_ = info.n_rows;
        s.dim_ = info.dim;
        s.L_ = info.num_partitions;
        return s;
    }

private:
    scann_hip_txh_desc desc() const {
        const DenseDataset &ds = *dataset_;
        const uint32_t S = (uint32_t)config_.hash_config.num_subspaces;
        scann_hip_txh_desc d{};
        d.data = ds.raw_data();
        d.n_rows = ds.size();
        d.dim = (uint32_t)ds.dimensionality();
        d.stride = ds.stride();
        d.centers = centers_.data();
        d.num_partitions = (uint32_t)(leaf_off_.size() - 1);
        d.leaf_offsets = leaf_off_.data();
        d.leaf_ids = leaf_ids_.data();
        d.n_local = ds.size();
        d.codebook = codebook_.data();
        d.num_subspaces = S;
        d.num_codes = (uint32_t)config_.hash_config.num_codes;
        d.dims_per_subspace = d.dim / S;
        d.codes = codes_.data();
        d.use_residuals = config_.use_residuals ? 1 : 0;
        d.partitions_to_search = (uint32_t)config_.partitions_to_search;
        d.pre_reorder_multiplier = config_.pre_reorder_multiplier;
        return d;
    }
    TreeXHybridConfig config_;
    int device_;
    std::shared_ptr<DenseDataset> dataset_;
    std::vector<float> centers_, codebook_;
    std::vector<uint32_t> leaf_off_, leaf_ids_;
    std::vector<uint8_t> codes_;
    size_t n_ = 0, dim_ = 0, L_ = 0;
    detail::IndexHandle ix_;
};

namespace detail {
inline void build_partition(const DenseDataset &ds, size_t num_partitions, size_t iterations, int device,
                            std::vector<float> &centers, std::vector<uint32_t> &leaf_off,
                            std::vector<uint32_t> &leaf_ids) {
    const size_t n = ds.size(), dim = ds.dimensionality();
    std::vector<uint32_t> assign;
    {
        IndexHandle tmp;
        check(scann_hip_bf_create(context(device), ds.raw_data(), n, (uint32_t)dim, ds.stride(), SCANN_HIP_SQUARED_L2,
                                  &tmp.h));
        kmeans_gpu(tmp.h, n, 0, dim, num_partitions, iterations, 1e-5, 42, centers, assign);
    }
    const uint32_t L = (uint32_t)(centers.size() / dim);
    leaf_off.assign(L + 1, 0);
    for (uint32_t a : assign) ++leaf_off[a + 1];
    for (uint32_t l = 0; l < L; ++l) leaf_off[l + 1] += leaf_off[l];
    leaf_ids.assign(n, 0);
    std::vector<uint32_t> cur(leaf_off.begin(), leaf_off.end() - 1);
    for (size_t i = 0; i < n; ++i) leaf_ids[cur[assign[i]]++] = (uint32_t)i;   // ascending idx per leaf
}
}  // namespace detail

// ---- ScannConfig (config.rs:10-320: the fields the hot path reads) ------------------------------
struct PartitioningConfig {          // config.rs:134-199
    uint32_t num_partitions = 100, num_partitions_to_search = 10;
};
struct HashConfig {                  // config.rs:202-261
    uint32_t num_buckets = 256, num_blocks = 16;
};
struct ExactReorderingConfig {       // config.rs:283-320
    uint32_t num_candidates = 100;
};
struct ScannConfig {                 // config.rs:10-118
    uint32_t num_neighbors = 10;
    DistanceMeasure distance_measure = DistanceMeasure::SquaredL2;
    bool brute_force = false;
    bool has_partitioning = false, has_hash = false, has_exact_reordering = false;
    PartitioningConfig partitioning;
    HashConfig hash;
    ExactReorderingConfig exact_reordering;
    ScannConfig with_num_neighbors(uint32_t k) const { auto c = *this; c.num_neighbors = k; return c; }
    ScannConfig with_distance_measure(DistanceMeasure m) const { auto c = *this; c.distance_measure = m; return c; }
    ScannConfig with_brute_force() const {   // :63-68 clears the other two
        auto c = *this; c.brute_force = true; c.has_partitioning = false; c.has_hash = false; return c;
    }
    ScannConfig with_partitioning(PartitioningConfig p) const { auto c = *this; c.partitioning = p; c.has_partitioning = true; return c; }
    ScannConfig with_hash(HashConfig h) const { auto c = *this; c.hash = h; c.has_hash = true; return c; }
    ScannConfig with_exact_reordering(ExactReorderingConfig r) const {
        auto c = *this; c.exact_reordering = r; c.has_exact_reordering = true; return c;
    }
};

enum class SearchMode { BruteForce, Partitioned, Hashed, TreeAH };   // scann.rs:17-28

// ---- Scann facade (scann.rs:30-362) ------------------------------------------------------------
// One GPU index per mode, all configurations of the same kernels (include/scann_hip.h):
//   Partitioned  exact scan of the selected leaves' rows     search_partitioned  :213-252
//   Hashed       AsymmetricHasher::search                    search_impl         :189-192
//   TreeAH       one non-residual table per query, k best    search_tree_ah      :255-294
// and the exact reordering of the k-truncated list with the configured measure (:199-209) runs
// inside the same search call (pre_reorder_k = k, exact_reorder = 1).
class Scann {
public:
    explicit Scann(DenseDataset dataset, int device = 0) : Scann(std::move(dataset), ScannConfig(), device) {}
    Scann(DenseDataset dataset, ScannConfig config, int device = 0) : config_(config), device_(device) {   // :63-103
        if (dataset.is_empty()) throw ScannError::invalid_argument("Dataset cannot be empty");  // :64-66
        dataset_ = std::make_shared<DenseDataset>(std::move(dataset));
        bf_.reset(new BruteForceSearcher(dataset_, config_.distance_measure, device));
        if (config_.brute_force) {
            mode_ = SearchMode::BruteForce;
        } else if (config_.has_partitioning && config_.has_hash) {
            init_partitioning();
            init_hashing();
            mode_ = SearchMode::TreeAH;
            create_index(true, true);
        } else if (config_.has_partitioning) {
            init_partitioning();
            mode_ = SearchMode::Partitioned;
            create_index(true, false);
        } else if (config_.has_hash) {
            init_hashing();
            mode_ = SearchMode::Hashed;
            create_index(false, true);
        }
    }
    static Scann brute_force(DenseDataset ds) { return Scann(std::move(ds), ScannConfig().with_brute_force()); }   // :106-109
    static Scann partitioned(DenseDataset ds, uint32_t num_partitions, uint32_t partitions_to_search) {          // :112-124
        PartitioningConfig p;
        p.num_partitions = num_partitions;
        p.num_partitions_to_search = partitions_to_search;
        return Scann(std::move(ds), ScannConfig().with_partitioning(p));
    }
    static Scann hashed(DenseDataset ds, uint32_t num_blocks) {                                                  // :127-137
        HashConfig h;
        h.num_blocks = num_blocks;
        return Scann(std::move(ds), ScannConfig().with_hash(h));
    }

    NNResultsVector search(const std::vector<float> &q, size_t k) const {      // :175-212
        return search_batched({q}, k)[0];
    }
    // :297-303 maps search over the queries; here the whole batch is one GPU call
    std::vector<NNResultsVector> search_batched(const std::vector<std::vector<float>> &qs, size_t k) const {
        if (qs.empty()) return {};
        // the reordering of a brute-force result re-scores it with the same kernel: a no-op
        if (mode_ == SearchMode::BruteForce) return bf_->search_batched(qs, k);
        uint32_t d;
        auto flat = detail::flatten(qs, &d);
        scann_hip_search_opts o;
        scann_hip_search_opts_default(&o);
        o.pre_reorder_k = (uint32_t)k;
        const bool reorder = config_.has_exact_reordering && config_.exact_reordering.num_candidates > k;   // :199-201
        o.exact_reorder = (reorder && mode_ != SearchMode::Partitioned) ? 1 : 0;
        if (mode_ != SearchMode::Hashed) o.partitions_to_search = config_.partitioning.num_partitions_to_search;
        return detail::run_search(ix_.h, flat.data(), (uint32_t)qs.size(), d, d, (uint32_t)k, &o);
    }
    SearchMode search_mode() const { return mode_; }
    size_t size() const { return dataset_->size(); }
    uint64_t dimensionality() const { return dataset_->dimensionality(); }
    DistanceMeasure distance_measure() const { return config_.distance_measure; }
    const ScannConfig &config() const { return config_; }
    size_t dataset_size() const { return size(); }

private:
    void init_partitioning() {   // :140-150 -> TreePartitioner::build (tree_partitioner.rs:48-98)
        detail::build_partition(*dataset_, config_.partitioning.num_partitions, 100, device_, centers_, leaf_off_,
                                leaf_ids_);
    }
    void init_hashing() {        // :153-165 -> AsymmetricHasher::build (hasher.rs:109-134)
        const DenseDataset &ds = *dataset_;
        const uint32_t S = config_.hash.num_blocks, K = config_.hash.num_buckets, dim = (uint32_t)ds.dimensionality();
        AsymmetricHasherConfig hc(K, S);
        detail::IndexHandle tmp;
        check(scann_hip_bf_create(context(device_), ds.raw_data(), ds.size(), dim, ds.stride(), SCANN_HIP_SQUARED_L2,
                                  &tmp.h));
        codebook_ = detail::train_codebook(tmp.h, ds.size(), dim, S, K, hc.seed, hc.training_iterations,
                                           hc.convergence_threshold);
        codes_.assign(ds.size() * S, 0);
        check(scann_hip_encode(context(device_), codebook_.data(), S, K, dim / S, ds.raw_data(), ds.size(),
                               ds.stride(), nullptr, nullptr, codes_.data()));
    }
    void create_index(bool tree, bool hash) {
        const DenseDataset &ds = *dataset_;
        const size_t n = ds.size();
        scann_hip_txh_desc d{};
        d.data = ds.raw_data();
        d.n_rows = n;
        d.dim = (uint32_t)ds.dimensionality();
        d.stride = ds.stride();
        d.n_local = n;
        d.distance_measure = (int)config_.distance_measure;
        d.partitions_to_search = tree ? config_.partitioning.num_partitions_to_search : 1;
        d.pre_reorder_multiplier = 1.0f;
        std::vector<uint8_t> csr_codes;
        if (tree) {
            d.centers = centers_.data();
            d.num_partitions = (uint32_t)(leaf_off_.size() - 1);
            d.leaf_offsets = leaf_off_.data();
            d.leaf_ids = leaf_ids_.data();
        }
        if (hash) {
            const uint32_t S = config_.hash.num_blocks;
            d.codebook = codebook_.data();
            d.num_subspaces = S;
            d.num_codes = config_.hash.num_buckets;
            d.dims_per_subspace = d.dim / S;
            d.codes = codes_.data();
            if (tree) {   // encoded_database()[idx] (scann.rs:283) laid out in CSR row order
                csr_codes.resize(n * S);
                for (size_t r = 0; r < n; ++r)
                    std::memcpy(&csr_codes[r * S], &codes_[(size_t)leaf_ids_[r] * S], S);
                d.codes = csr_codes.data();
            }
        }
        check(scann_hip_txh_create(context(device_), &d, &ix_.h));
    }

    ScannConfig config_;
    int device_;
    SearchMode mode_ = SearchMode::BruteForce;
    std::shared_ptr<DenseDataset> dataset_;
    std::unique_ptr<BruteForceSearcher> bf_;
    std::vector<float> centers_, codebook_;
    std::vector<uint32_t> leaf_off_, leaf_ids_;
    std::vector<uint8_t> codes_;
    detail::IndexHandle ix_;
};

class ScannBuilder {   // scann.rs:364-426
public:
    ScannBuilder &num_neighbors(uint32_t k) { config_.num_neighbors = k; return *this; }
    ScannBuilder &distance_measure(DistanceMeasure m) { config_.distance_measure = m; return *this; }
    ScannBuilder &brute_force() { config_ = config_.with_brute_force(); return *this; }
    ScannBuilder &tree(uint32_t num_partitions, uint32_t partitions_to_search) {
        config_.partitioning.num_partitions = num_partitions;
        config_.partitioning.num_partitions_to_search = partitions_to_search;
        config_.has_partitioning = true;
        return *this;
    }
    ScannBuilder &hash(uint32_t num_blocks) {
        config_.hash = HashConfig();
        config_.hash.num_blocks = num_blocks;
        config_.has_hash = true;
        return *this;
    }
    ScannBuilder &reorder(uint32_t num_candidates) {
        config_.exact_reordering.num_candidates = num_candidates;
        config_.has_exact_reordering = true;
        return *this;
    }
    Scann build(DenseDataset dataset) const { return Scann(std::move(dataset), config_); }

private:
    ScannConfig config_;
};

}  // namespace scann
