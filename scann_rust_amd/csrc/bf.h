// bf.h -- brute-force path (BruteForceSearcher, brute_force/searcher.rs:77-208).
#pragma once
#include "common.h"

namespace scann {

struct BfIndexDev {
    const float *rows;   // [n][stride]
    uint64_t n;
    uint32_t dim, stride;
    int measure;         // scann_hip_measure
    // bf16 shortlist (bf_shortlist_*): a bf16 copy of the rows [n][dim], f32 squared norms and the
    // largest row norm; nullptr when the index does not qualify (dim % 16, size)
    const uint16_t *rows_b, *rows_bl;   // hi and lo halves of the split bf16 copy
    const float *norm2;
    float max_norm;
};

struct BfWorkspace {
    DevBuf queries, sample, thr, cand_cnt, cand, counters, out_idx, out_dist, out_count;
    DevBuf q_b, q_bl, q_n2, sl_idx, sl_approx, sl_cnt, sl_exact, sl_fail;   // bf16 shortlist path
};

// bf16 copy + squared norms of the rows (index creation); *max_norm = largest row norm.
int bf_build_shortlist_data(const BfIndexDev &ix, DevBuf &rows_b, DevBuf &rows_bl, DevBuf &norm2,
                            float *max_norm, hipStream_t stream);
// true if searches on this index may take the bf16-shortlist path for (nq, k)
bool bf_shortlist_eligible(const BfIndexDev &ix, uint32_t nq, uint32_t k);

constexpr uint32_t kBfSampleRows = 8192;   // rows of the threshold sample (== LDS sort size)

int bf_reserve(const BfIndexDev &ix, BfWorkspace &w, uint32_t max_nq, uint32_t max_k);

const char *bf_pass_kernel_name(const BfIndexDev &ix, uint32_t nq);

// Host-pointer entry (copies in/out, synchronises).
int bf_search_host(const BfIndexDev &ix, BfWorkspace &w, const float *queries, uint32_t nq,
                   uint32_t q_stride, uint32_t k, bool exact_only, uint32_t *out_idx, float *out_dist,
                   uint32_t *out_count, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);

// Device-pointer entry (enqueue only).
// exact_only = false: searches that qualify take the bf16-shortlist path (status Aborted in the
// counters if a query's result could not be verified: repeat with exact_only = true).
int bf_search_device(const BfIndexDev &ix, BfWorkspace &w, const float *d_queries, uint32_t nq,
                     uint32_t q_stride, uint32_t k, bool exact_only, uint32_t *d_out_idx,
                     float *d_out_dist, uint32_t *d_out_count, hipStream_t stream, hipEvent_t ev0,
                     hipEvent_t ev1);
// OK, or the status the last enqueued search left in the workspace counters (synchronises).
int bf_last_status(const BfWorkspace &w, hipStream_t stream);

// Dense [nq][n] distance matrix to host memory.
int bf_distances_host(const BfIndexDev &ix, BfWorkspace &w, const float *queries, uint32_t nq,
                      uint32_t q_stride, float *out, hipStream_t stream);

// BruteForceSearcher::search_radius for one query: all rows with distance <= radius, sorted by
// (distance, index).  At most `capacity` rows are written; *out_count = number found.
int bf_search_radius_host(const BfIndexDev &ix, BfWorkspace &w, const float *query, uint32_t q_stride,
                          float radius, uint32_t *out_idx, float *out_dist, uint64_t capacity,
                          uint64_t *out_count, hipStream_t stream);

// Nearest of k centres for every row of the index (sequential-scalar SquaredL2, lowest index on
// ties): TreePartitioner::partition(x, 1) / KMeans::assign_clusters.
int bf_assign_nearest_host(const BfIndexDev &ix, const float *centers, uint32_t k, uint32_t *out_idx,
                           float *out_dist, hipStream_t stream);

// K-means over the rows of the index (optionally the column window [col_offset, col_offset +
// sub_dim)): k-means++ seeding (trees/kmeans.rs:295-349, splitmix64 stream) and the Lloyd loop of
// KMeans::fit_single (:210-263) from caller-supplied centres [k][sub_dim] (updated in place).
int bf_kmeans_init_pp_host(const BfIndexDev &ix, uint32_t col_offset, uint32_t sub_dim, uint32_t k,
                           uint64_t seed, uint32_t simd_threshold, float *centers_out, hipStream_t stream);
int bf_kmeans_lloyd_host(const BfIndexDev &ix, uint32_t col_offset, uint32_t sub_dim, float *centers,
                         uint32_t k, uint32_t max_iterations, double convergence_threshold,
                         uint32_t simd_threshold, uint32_t *out_assign, uint32_t *out_sizes,
                         double *out_inertia, uint32_t *out_iterations, int *out_converged,
                         hipStream_t stream);

}  // namespace scann
