/*
 * scann_oracle.h -- CPU restatement of the sunbains/scann-rust Tree-X-Hybrid /
 * brute-force hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the shipped product (libscann_hip.so) never links,
 * imports or calls anything in oracle/.
 *
 * PARITY PIN: the reference is Rust and cannot be compiled in the build
 * container (no cargo/rustc, no network), and it ships no golden files.  The
 * oracle is therefore pinned by transcribing every known-answer unit test
 * the reference holds for this path into
 * tests/test_oracle_known_answers.py.  The Tree-X-Hybrid and
 * AsymmetricHasher *searches* carry only structural asserts in the
 * reference (tree_x_hybrid/mod.rs:436-468, hashes/hasher.rs:322-380), so for
 * those two entry points the end-to-end numbers are "parity unpinned by the
 * reference"; every arithmetic building block they are composed of is pinned.
 *
 * Every function cites the reference file:line it restates (paths relative
 * to /root/reference/src).
 */
#ifndef SCANN_ORACLE_H
#define SCANN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DistanceMeasure discriminants used on this path (distance_measures/mod.rs:32-66). */
enum { OR_SQUARED_L2 = 0, OR_L2 = 1, OR_DOT_PRODUCT = 2, OR_L1 = 3, OR_COSINE = 4 };

/* l1_distance_avx2 simd/x86.rs:103-132: 8 lane chains of |a - b| (no FMA), the fixed horizontal-sum
 * tree (x86.rs:31-44), unfused scalar tail. */
float or_l1_avx2(const float *a, const float *b, size_t n);
/* cosine_distance one_to_one.rs:559-612: 1 - dot/(|a| |b|), 0 similarity when a norm is 0.  The three
 * sums run as 8-lane PortableF32x8 chains (mul then add, not fused) and are reduced by the THIRD-PARTY
 * wide::f32x8::reduce_add (crate `wide` = "0.7", Cargo.toml:49; no Cargo.lock pins the patch version).
 * The reference is built without -C target-feature=+avx (no .cargo/config), so wide takes its
 * non-AVX path: f32x8 is two f32x4 halves, reduce_add = a.reduce_add() + b.reduce_add(), each half
 * the sequential sum ((v0 + v1) + v2) + v3.  That order is restated here from the published crate
 * source; nothing in the reference pins it (its cosine tests use axis vectors): PARITY UNPINNED. */
float or_cosine_distance(const float *a, const float *b, size_t n);

/* ---- L1 kernels -------------------------------------------------------- */

/* simd/x86.rs:139-165 (squared_l2_avx2) + hsum :31-44.  8 FMA lane chains,
 * fixed horizontal-sum tree, scalar non-fused tail. */
float or_squared_l2_avx2(const float *a, const float *b, size_t n);
/* simd/x86.rs:72-96 (dot_product_avx2). */
float or_dot_product_avx2(const float *a, const float *b, size_t n);
/* simd/dispatch.rs:159-183 portable fallback (mul then add, no FMA). */
float or_squared_l2_portable(const float *a, const float *b, size_t n);
float or_dot_product_portable(const float *a, const float *b, size_t n);
/* partitioning/tree_partitioner.rs:184-192, hashes/codebook.rs:107-115:
 * strictly sequential scalar sum of (x-y)^2, no contraction. */
float or_squared_l2_sequential(const float *a, const float *b, size_t n);

/* simd/x86.rs:267-346 / :195-258 via distance_measures/one_to_many.rs:228-276,
 * 341-373.  Dot results are negated. */
void or_one_to_many_squared_l2(const float *q, size_t dim, const float *db,
                               size_t stride, size_t n, float *out);
void or_one_to_many_dot_product(const float *q, size_t dim, const float *db,
                                size_t stride, size_t n, float *out);

/* data_format/dataset.rs:90-96: stride = align_up(dim, 64 / sizeof(f32)). */
size_t or_compute_stride(size_t dim);

/* ---- top-k structures (brute_force/top_k.rs) --------------------------- */

/* TopK :20-113 driven with (idx[i], dist[i]) in order; drain_sorted(). */
size_t or_topk_run(size_t k, const uint32_t *idx, const float *dist, size_t n,
                   uint32_t *out_idx, float *out_dist);
/* FastTopNeighbors :264-393: push() each, then results(). */
size_t or_fast_top_neighbors_run(size_t cap, const uint32_t *idx,
                                 const float *dist, size_t n,
                                 uint32_t *out_idx, float *out_dist);
/* FastTopNeighbors::push_batch :358-365 then results(). */
size_t or_fast_top_neighbors_push_batch(size_t cap, const uint32_t *idx,
                                        const float *dist, size_t n,
                                        uint32_t *out_idx, float *out_dist);

/* ---- brute force (brute_force/searcher.rs:77-208) ---------------------- */

/* Returns result count (min(k,n)), or -3 (InvalidArgument) on dim mismatch. */
int or_bf_search(const float *data, size_t n, size_t dim, size_t stride,
                 int measure, const float *q, size_t qdim, size_t k,
                 uint32_t *out_idx, float *out_dist);
/* search_batched: one task per query (rayon par_iter analogue = OpenMP). */
int or_bf_search_batched(const float *data, size_t n, size_t dim, size_t stride,
                         int measure, const float *queries, size_t nq,
                         size_t q_stride, size_t k, uint32_t *out_idx,
                         float *out_dist, uint32_t *out_count, int nthreads);
/* search_radius :142-167. */
size_t or_bf_search_radius(const float *data, size_t n, size_t dim, size_t stride,
                           int measure, const float *q, float radius,
                           uint32_t *out_idx, float *out_dist);

/* ---- partitioner (partitioning/tree_partitioner.rs:175-229) ------------ */
size_t or_partition(const float *centers, size_t L, size_t dim, const float *q,
                    size_t num_partitions, uint32_t *out_tokens, float *out_dists);

/* ---- PQ codebook / LUT -------------------------------------------------- */
/* codebook layout: [S][K][dsub] f32. */
/* hashes/codebook.rs:82-95, 205-215 */
void or_encode(const float *codebook, size_t S, size_t K, size_t dsub,
               const float *x, uint8_t *codes);
/* hashes/lut.rs:47-70 -> codebook.rs:98-103; lut layout [S][K]. */
void or_lut_from_query(const float *codebook, size_t S, size_t K, size_t dsub,
                       const float *q, float *lut);
/* hashes/lut.rs:74-82 */
float or_lut_distance(const float *lut, size_t S, size_t K, const uint8_t *codes);

/* ---- LUT16 (hashes/lut16.rs, hashes/lut16_simd.rs, simd/dispatch.rs) ---- */
size_t or_pack4_bytes_per_point(size_t S);
/* PackedCodes4Bit::from_codes lut16.rs:43-61 */
void or_pack4(const uint8_t *codes, size_t n, size_t S, uint8_t *packed);
/* PackedCodes4Bit::get_codes lut16.rs:64-77 */
void or_unpack4(const uint8_t *packed, size_t n, size_t S, uint8_t *codes);
/* Lut16LookupTables::compute_distance_packed lut16.rs:186-203 (f32 tables [S][16]) */
float or_lut16_distance_packed_f32(const float *tables, size_t S, const uint8_t *packed);
/* Lut16SimdTables::from_float_tables lut16_simd.rs:39-90 */
void or_lut16_quantize(const float *tables, size_t S, uint8_t *lut8,
                       float *bias, float *multiplier);
/* simd/dispatch.rs:259-295 (raw u32 sums as f32) */
void or_lut16_distances_batch_raw(const uint8_t *packed, const uint8_t *lut8,
                                  size_t S, size_t n, float *out);
/* Lut16SimdTables::compute_distances_batch lut16_simd.rs:119-141 */
void or_lut16_distances_batch(const uint8_t *packed, const uint8_t *lut8, size_t S,
                              size_t n, float bias, float multiplier, float *out);
/* Lut16SimdTables::compute_distance_single lut16_simd.rs:144-154 */
float or_lut16_distance_single(const uint8_t *lut8, size_t S, float bias,
                               float multiplier, const uint8_t *codes);

/* ---- AsymmetricHasher (hashes/hasher.rs:162-229) ------------------------ */
/* codes: unpacked [n][S]. Returns count or -3 on dim mismatch. */
int or_ah_search(const float *codebook, size_t S, size_t K, size_t dsub,
                 const uint8_t *codes, size_t n, const float *q, size_t qdim,
                 size_t k, uint32_t *out_idx, float *out_dist);
int or_ah_search_with_reordering(const float *codebook, size_t S, size_t K,
                                 size_t dsub, const uint8_t *codes, size_t n,
                                 const float *data, size_t stride,
                                 const float *q, size_t qdim, size_t k,
                                 size_t pre_reorder_k, uint32_t *out_idx,
                                 float *out_dist);

/* hasher.rs:232-238 (sequential map in the reference; nthreads = tasks per query) */
int or_ah_search_batched(const float *codebook, size_t S, size_t K, size_t dsub,
                         const uint8_t *codes, size_t n, const float *data, size_t stride,
                         const float *queries, size_t nq, size_t q_stride, size_t k,
                         size_t pre_reorder_k, int reorder, uint32_t *out_idx,
                         float *out_dist, uint32_t *out_count, int nthreads);

/* ---- Tree-X-Hybrid (tree_x_hybrid/mod.rs:245-364) ----------------------- */
typedef struct {
    uint32_t n, dim, stride;
    const float *data;           /* [n*stride] original rows (re-rank) */
    uint32_t L;                  /* partitions */
    const float *centers;        /* [L*dim] */
    const uint32_t *leaf_off;    /* [L+1] CSR offsets */
    const uint32_t *leaf_ids;    /* [n] datapoint index of CSR row i */
    uint32_t S, K, dsub;
    const float *codebook;       /* [S*K*dsub] */
    const uint8_t *codes;        /* [n*S] unpacked, CSR row order */
    int32_t use_residuals;
    uint32_t partitions_to_search;
    float pre_reorder_multiplier;
    /* search_with_filter(.., Some(f)) with an allow-list f (restricts/mod.rs:17-30): bit i =
     * datapoint i allowed.  NULL = filter None. */
    const uint64_t *allow;
} or_txh_index;

/* Optional stage outputs (may be NULL):
 *   tokens/token_dists [P]; cand_idx/cand_dist [pre_reorder_k] (merged, approx);
 * *n_tokens, *n_cand receive the lengths.  Returns final count or -3. */
int or_txh_search(const or_txh_index *ix, const float *q, size_t qdim, size_t k,
                  uint32_t *out_idx, float *out_dist,
                  uint32_t *tokens, float *token_dists, size_t *n_tokens,
                  uint32_t *cand_idx, float *cand_dist, size_t *n_cand);
/* search_batched_with_params :399-409 (par_iter over queries). */
int or_txh_search_batched(const or_txh_index *ix, const float *queries, size_t nq,
                          size_t q_stride, size_t k, uint32_t *out_idx,
                          float *out_dist, uint32_t *out_count, int nthreads);

/* utils/reordering.rs:23-54 == tree_x_hybrid/mod.rs:342-364 */
size_t or_reorder(const float *data, size_t stride, size_t dim, const float *q,
                  const uint32_t *cand_idx, size_t n_cand, size_t k,
                  uint32_t *out_idx, float *out_dist);

/* ---- Scann facade modes (scann.rs:181-294) -------------------------------------------- */
/* DistanceMeasure::distance for dense f32 (distance_measures/mod.rs:70-81 ->
 * one_to_one.rs:156-171, 320-345, 464-469 -> simd/dispatch.rs single-pair kernels). */
float or_measure_distance(int measure, const float *a, const float *b, size_t dim);
/* ReorderingHelper::reorder (utils/reordering.rs:23-54) with the configured measure. */
size_t or_reorder_measure(const float *data, size_t stride, size_t dim, int measure, const float *q,
                          const uint32_t *cand_idx, size_t n_cand, size_t k, uint32_t *out_idx,
                          float *out_dist);
/* Scann::search_partitioned (scann.rs:213-252): rows of the P nearest leaves (token order, then
 * leaf order), exact distance by `measure`, stable sort, first k.  leaf_ids[leaf_off[l]..] are the
 * datapoint indices of leaf l.  Returns the count. */
int or_scann_search_partitioned(const float *centers, size_t L, size_t dim, const uint32_t *leaf_off,
                                const uint32_t *leaf_ids, const float *data, size_t stride, int measure,
                                const float *q, size_t P, size_t k, uint32_t *out_idx, float *out_dist);
/* Scann::search_tree_ah (scann.rs:255-294): one non-residual table for the query, every row of the
 * P nearest leaves scored by it, stable sort, first k; codes [n][S] by datapoint index.
 * reorder != 0: search_impl's exact reordering of the k-truncated list (scann.rs:199-209). */
int or_scann_search_tree_ah(const float *centers, size_t L, size_t dim, const uint32_t *leaf_off,
                            const uint32_t *leaf_ids, const float *codebook, size_t S, size_t K,
                            size_t dsub, const uint8_t *codes, const float *data, size_t stride,
                            int measure, int reorder, const float *q, size_t P, size_t k,
                            uint32_t *out_idx, float *out_dist);

/* harness helpers: bin/ann_benchmark.rs:427-471 */
void or_exact_ground_truth(const float *train, size_t n, size_t dim, size_t stride,
                           const float *queries, size_t nq, size_t q_stride,
                           size_t k, uint32_t *gt, int nthreads);

int or_max_threads(void);

/* trees/kmeans.rs:210-263: the Lloyd loop of KMeans::fit_single from given centres (in/out
 * [k][dim]) over the column window [col_offset, col_offset + dim) of data[n][stride].
 * simd_threshold: dimensions from which the reference uses its AVX2 squared_l2 (default 128;
 * pass SIZE_MAX for always-scalar). */
int or_kmeans_lloyd(const float *data, size_t n, size_t stride, size_t col_offset, size_t dim,
                    float *centers, size_t k, size_t max_iterations, double convergence_threshold,
                    size_t simd_threshold, uint32_t *assign, uint32_t *sizes, double *inertia_out,
                    uint32_t *iterations_out, int *converged_out);

/* quantization/fp8.rs: the reference's own FP8 codec (NOT a hardware format table: its rounding wraps
 * a mantissa carry without bumping the exponent, exponent field 15 only ever encodes the maximum, and
 * subnormals flush to zero on encode -- restated bit for bit).  format: 0 = E4M3 (:80-143), 1 = E5M2
 * (:146-203). */
#define OR_FP8_E4M3 0
#define OR_FP8_E5M2 1
uint8_t or_fp8_from_f32(float value, int format);
float or_fp8_to_f32(uint8_t bits, int format);
/* Fp8Quantizer::calibrate_scale (:238-244): fp8_max / max(max_abs, 1e-10). */
float or_fp8_calibrate_scale(float max_abs_value, int format);
/* Quantizer::quantize / dequantize over Fp8Quantizer (:247-268): from_f32(value * scale); to_f32 / scale. */
void or_fp8_quantize(const float *values, size_t n, float scale, int format, uint8_t *out);
void or_fp8_dequantize(const uint8_t *bits, size_t n, float scale, int format, float *out);
/* one_to_many_fp8_float_{dot_product, squared_l2} (distance_measures/one_to_many_asymmetric.rs:327-377):
 * E4M3 rows, sequential f32 sum; measure OR_DOT_PRODUCT (result negated) or OR_SQUARED_L2. */
void or_one_to_many_fp8(const float *query, size_t dim, const uint8_t *database, size_t stride,
                        size_t num_points, int measure, float *results);

#ifdef __cplusplus
}
#endif
#endif
