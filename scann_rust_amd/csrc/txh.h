// txh.h -- device-side views + launchers of the Tree-X-Hybrid / AsymmetricHasher path.
#pragma once
#include "common.h"

namespace scann {

// Tunables of the scan decomposition (see DESIGN.md "Leaf scan").
constexpr uint32_t kScanThreads = 256;
#ifndef SCANN_SCAN_PPT
#define SCANN_SCAN_PPT 2
#endif
constexpr uint32_t kScanPPT = SCANN_SCAN_PPT;              // points per thread per chunk
constexpr uint32_t kScanTP = kScanThreads * kScanPPT;     // points per tile chunk
constexpr uint32_t kScanQuadsPerTile = 32;                // query quads per tile
#ifndef SCANN_SCAN_DEPTH
#define SCANN_SCAN_DEPTH 1
#endif
#ifndef SCANN_SCAN_WAVES
#define SCANN_SCAN_WAVES 6
#endif
constexpr uint32_t kScanDepth = SCANN_SCAN_DEPTH;         // gather software-pipeline depth (subspaces)
constexpr uint32_t kScanWaves = SCANN_SCAN_WAVES;         // workgroups per CU = waves per SIMD (S <= 32)
constexpr uint32_t kResThreads = 512;                     // resident-table scan: threads per workgroup
constexpr uint32_t kResQuads = 4;                         // ... quads whose tables stay in LDS
constexpr uint32_t kSortCap = 16384;                      // u64 keys sorted in LDS (select)
#ifndef SCANN_SAMPLE_TARGET
#define SCANN_SAMPLE_TARGET 32768
#endif
#ifndef SCANN_SAMPLE_MIN
#define SCANN_SAMPLE_MIN 4096
#endif
constexpr uint32_t kSampleTarget = SCANN_SAMPLE_TARGET;   // max sample points per query (LDS of the select)
constexpr uint32_t kSampleMin = SCANN_SAMPLE_MIN;         // min sample points per query
constexpr uint32_t kMaxPreReorderK = 8192;                // m limit of the LDS select
constexpr uint32_t kMaxPartitionsToSearch = 4096;
constexpr uint32_t kMaxLeavesSelect = 16384;              // L limit of the LDS leaf sort
constexpr uint32_t kSelectThreads = 1024;
constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr uint32_t kSmallBatch = 16;                      // queries per call the small-batch pipeline takes
constexpr uint32_t kSmallMaxStream = 262144;              // ... longest candidate stream (points) it takes
constexpr uint32_t kSmallMaxCandidates = 1024;            // ... largest pre_reorder_k
// the wide few-query pipeline (txh.hip "Few queries, long streams"): three launches, every stage spread over the chip
constexpr uint32_t kWideBatch = 4;                        // queries per call it takes
constexpr uint32_t kWideMaxStream = 4u << 20;             // ... longest candidate stream (points)
constexpr uint32_t kWideMaxCandidates = 8192;             // ... largest pre_reorder_k
constexpr uint32_t kWideMinStream = 16384;                // ... and the stream from which it replaces the pipeline above

// ---- threshold sampling plan (shared by host buffer sizing and the device kernels) ----
// Every st-th point of each selected leaf is scored ahead of the scan (adc_sample_kernel,
// the same tiled LUT16 gather as the scan); the threshold is the j-th smallest sample
// distance (threshold_select_kernel).  One stride per batch, so that all queries of a
// tile share the sampled points' codes.  j == m gives a deterministic bound (any
// subset's m-th smallest bounds the stream's m-th smallest); j < m is a 6-sigma
// statistical bound that select_rerank VERIFIES (>= m survivors, else the host entry
// retries without a threshold), so results stay exact either way.
__host__ __device__ static inline uint32_t sample_stride(uint32_t total) {
    uint32_t ns = total / 16u;
    if (ns < kSampleMin) ns = kSampleMin;
    if (ns > kSampleTarget) ns = kSampleTarget;
    uint32_t st = (total + ns - 1u) / ns;
    return st ? st : 1u;
}
// Stride and per-query sample capacity for a batch whose longest stream (sum of the P
// largest local leaves) is max_stream: capacity = max_stream / st + P <= kSampleTarget.
static inline void sample_plan(uint64_t max_stream, uint32_t P, uint32_t *st_out, uint32_t *scap_out) {
    const uint32_t ms = (uint32_t)(max_stream > 0xFFFFFFFFull ? 0xFFFFFFFFull : max_stream);
    uint32_t st = sample_stride(ms);
    for (;;) {
        const uint64_t scap = (((uint64_t)(ms + st - 1u) / st + P) + 3u) & ~3ull;   // rows 16-B aligned
        if (scap <= kSampleTarget || P >= kSampleTarget - 4) {
            *st_out = st;
            *scap_out = (uint32_t)(scap < 4 ? 4 : scap);
            return;
        }
        st += 1 + st / 8;
    }
}
// 0 = use no threshold.
__host__ __device__ static inline uint32_t sample_rank(uint32_t m, uint32_t st) {
    if (m == 0) return 0;
    const float r = (float)m / (float)st;
    const float s = 0.5f * (6.0f + sqrtf(36.0f + 4.0f * r));
    const float jp = s * s + 2.0f;
    uint32_t j = jp >= (float)m ? m : (uint32_t)jp;
    if (j > m) j = m;
    if (j == 0) j = 1;
    return j;
}

struct TxhIndexDev {
    uint32_t dim, stride, L, S, K, dsub, nw;
    uint32_t code_bits, kp;       // 4-bit codes / 16 table slots (K <= 16) or 8-bit / 256 slots
    uint64_t n_local;
    const float *centers;         // [L][dim]; nullptr in AsymmetricHasher mode
    const float *centers_t;       // [dim][centers_pitch] transposed copy (small-batch leaf selection), or nullptr
    uint32_t centers_pitch;
    const uint32_t *leaf_off;     // [L+1] local CSR offsets
    const uint32_t *leaf_gsize;   // [L] global leaf sizes (== local when unsharded)
    const uint32_t *leaf_ids;     // [n_local] datapoint index of CSR row; nullptr = identity
    const uint32_t *codes;        // [n_local][nw] packed 4-bit codes, 8 subspaces per word
    const uint32_t *codes_sp;     // [n_local][sp_words(S)] the same codes as operand planes of the 2:4-sparse MFMA prefilter
                                  // (txh.hip K6e; 4-bit codes only), or nullptr
    const float *rows;            // re-rank rows; CSR order if rows_csr else by datapoint idx
    const int8_t *rows8;          // the same rows as int8 (per-row scale) for the re-rank filter, or nullptr
    int rows8_fmt;                // 0 = int8, 1 = the reference's FP8 E4M3 codes (quantization/fp8.rs)
    const void *rows8_meta;       // [n_rows] float2 {scale, ||x - s q||}
    int rows8_uniform;            // int8 store with ONE scale and ONE error bound for all rows (rows8_scale, rows8_emax)
    float rows8_scale, rows8_emax;
    int rows_csr;
    const float *codebook;        // [S][K][dsub]
    int use_residuals;
    int ah_mode;                  // single implicit leaf, no centroid stage
    int measure;                  // measure of the exact re-rank and of the exact leaf scan
    int exact_scan;               // SearchMode::Partitioned: no codes, rows of the leaves scored exactly
};

// points of one scan tile chunk for this index's code layout (Codec<S, bits>::TP in txh.hip)
static inline uint32_t scan_tile_points(const TxhIndexDev &ix) {
    return kScanThreads * (ix.code_bits == 4 ? kScanPPT : 8u);
}

// counters[] slots
enum {
    CNT_TOTAL_QUADS = 0, CNT_TOTAL_TILES = 1, CNT_QUEUE_HEAD = 2, CNT_STATUS = 3,
    CNT_TOTAL_STILES = 4, CNT_SQUEUE_HEAD = 5, CNT_N = 8,
    // eight scan-tile queues, one 128-byte line each: WGs of XCD x (blockIdx % 8) pull tiles
    // t = x (mod 8) from queue x first and steal from the others when it runs dry.  One
    // shared counter made every tile grab a ~100 ns serialized cross-XCD atomic.
    CNT_XQ = 32, CNT_XQ_STRIDE = 32, CNT_XS = CNT_XQ + 8 * CNT_XQ_STRIDE /* sample pass */,
    CNT_WORDS = CNT_XS + 8 * CNT_XQ_STRIDE
};

struct TxhWork {
    uint32_t nq, q_stride, P, m, k, cap;
    int exact_reorder;
    int no_threshold;          // retry mode: keep every scanned point as a candidate
    int need_sorted_cands;     // the caller reads cand_* (parity outputs): keep them sorted
    const uint64_t *allow;     // device allow-bitmap (bit = datapoint index) or nullptr
    uint64_t allow_bits;       // bitmap capacity in bits; indices >= capacity are not allowed
    const float *queries;      // device
    float *cdist;              // [nq][L]
    uint32_t *tokens;          // [nq][P]
    float *token_dists;        // [nq][P]
    uint32_t *vbase;           // [nq][P+1] prefix of global leaf sizes in token order
    uint32_t st, scap, sqpt;   // sample stride, per-query sample capacity, quads per sample tile
    uint32_t qpt;              // quads per scan tile
    uint32_t resident, res_cl; // resident-table scan kernel (long leaves) and its chunks per tile
    uint32_t small;            // small-batch pipeline (three launches; dense candidate lists: cap = the stream)
    uint32_t small_max_leaf;   // longest local leaf (grid of the small scan)
    uint32_t use_i8;           // int8 row filter in front of the exact re-rank (needs ix.rows8)
    uint32_t *rr_lb, *rr_ub;   // [nq][m] ordered lower / upper bounds of the candidates' exact distances
    uint32_t mfma;             // integer-MFMA prefilter + exact refine instead of the f32 LDS-gather scan:
                               // 1 = 32-pair dense tiles, 2 = 16-pair dense tiles, 3 = 32-pair tiles on the sparse MFMA
    int8_t *lut8;              // [max_slots][S][16] quantised tables (value - 128)
    void *lut8_meta;           // [max_slots] {f64 bias_sum, f64 scale}
    int *mfma_thr1;            // [max_slots] integer pass bound + 1 of every pair slot
    uint32_t *cand32_cnt;      // [nq]
    uint32_t *cand32;          // [nq][cap32] stream positions of the prefilter's survivors
    uint32_t *cand32_codes;    // [nq][cap32][S/8] their packed codes (written next to the positions)
    uint32_t *small_done;      // small-batch host calls: [nq] pinned completion flags (or nullptr) ...
    uint32_t small_seq;        // ... and the value the finish kernel stores there after the result rows
    uint32_t *small_tickets;   // [kSmallBatch] ticket counters of the one-launch pipeline (zero between launches)
    // small == 2: the wide few-query pipeline
    uint32_t wide_cap2;        // entries per query of the compact candidate arrays
    uint32_t *wide_min;        // [nq][cap] ordered approximate distance: minimum of each group of stream positions
    uint64_t *wide_ckey;       // [nq][wide_cap2] merge keys of the candidates under the pivot ...
    uint32_t *wide_ceb;        // ... their ordered exact distances (approximate ones without re-ordering) ...
    uint32_t *wide_cidx;       // ... and datapoint indices
    uint32_t *wide_cnt;        // [nq] entries appended
    uint32_t cap32;
    uint32_t *sbase;           // [nq][P+2] prefix of per-leaf sample counts; [P]=samples, [P+1]=local points
    uint32_t *pair_sbase;      // [max_slots]
    uint32_t *stile_off;       // [L+1] tile table of the sample pass
    uint32_t *samp;            // [nq][scap] ordered(approx distance) of the sampled points
    uint32_t *leaf_cnt;        // [L]
    uint32_t *leaf_cursor;     // [L]
    uint32_t *pair_off;        // [L+1] (slots, padded to quads)
    uint32_t *tile_off;        // [L+1]
    uint32_t *counters;        // [CNT_N]
    uint32_t *pair_q;          // [max_slots]
    uint32_t *pair_leaf;       // [max_slots]
    uint32_t *pair_vbase;      // [max_slots]
    uint32_t *slot_of;         // [nq][P]
    uint32_t max_slots, max_quads;
    float *lutq;               // [max_quads][S][16][4]
    uint64_t *thr;             // [nq]
    uint64_t *pair_thr;        // [max_slots] the same bound per (query, leaf) pair slot
    uint32_t *cand_cnt;        // [nq]
    uint64_t *cand;            // [nq][cap]
    uint64_t *cand_key;        // [nq][m] selected merge keys
    uint32_t *cand_idx;        // [nq][m]
    float *cand_dist;          // [nq][m] approximate
    float *cand_exact;         // [nq][m]
    uint32_t *cand_row;        // [nq][m] re-rank row of each candidate
    uint32_t *cand_count;      // [nq]
    uint32_t *out_idx;         // [nq][k]
    float *out_dist;           // [nq][k]
    uint32_t *out_count;       // [nq]
};

// Enqueue the whole search pipeline on `stream`.  local_only: stop after the local
// top-m + exact distances (multi-GPU local stage).
int txh_launch_search(const TxhIndexDev &ix, const TxhWork &w, bool local_only,
                      hipStream_t stream, hipEvent_t ev_scan_begin, hipEvent_t ev_scan_end);

int txh_launch_partition_only(const TxhIndexDev &ix, const TxhWork &w, hipStream_t stream);

int txh_launch_merge(uint32_t world, uint32_t nq, uint32_t m_local, uint32_t m, uint32_t k,
                     size_t rank_stride_bytes, const uint64_t *d_keys, const uint32_t *d_idx, const float *d_exact,
                     const uint32_t *d_count, uint32_t *d_out_idx, float *d_out_dist,
                     uint32_t *d_out_count, uint32_t *d_status, hipStream_t stream,
                     const uint32_t *d_qoff = nullptr /* compact lists: [world][nq] element offsets */);

int txh_launch_pack_blocks(uint32_t world, uint32_t nq, uint32_t m_local, const uint64_t *d_keys,
                           const uint32_t *d_idx, const float *d_exact, const uint32_t *d_count,
                           void *d_out, size_t block_bytes, hipStream_t stream);

int txh_launch_lut_from_query(const TxhIndexDev &ix, const float *d_queries, uint32_t nq,
                              uint32_t q_stride, const uint32_t *d_leaf_for_query,
                              float *d_out_lut, hipStream_t stream);

int txh_launch_adc_distances(const TxhIndexDev &ix, const float *d_luts, uint32_t nq,
                             float *d_out, hipStream_t stream);

int launch_lut16_u8_batch(const uint8_t *d_packed, const uint8_t *d_lut8, uint32_t S,
                          uint64_t n, float bias, float mult, float *d_out,
                          hipStream_t stream);

// Lut16SimdTables::from_float_tables (hashes/lut16_simd.rs:39-90); d_bias_mult = {bias, multiplier}
// int8 copy of n rows (per-row scale = max|x| / 127) + {scale, error norm} per row
int launch_transpose_centers(const float *d_centers, uint32_t L, uint32_t dim, uint32_t pitch, float *d_out,
                             hipStream_t st);
int launch_rows_fp8_build(const float *d_rows, uint64_t n, uint32_t dim, uint32_t stride, uint8_t *d_rows8,
                          void *d_meta, uint32_t *d_mismatch, hipStream_t st);
int launch_fp8_quantize(const float *d_values, uint64_t n, float scale, int format, uint8_t *d_out, hipStream_t st);
int launch_fp8_dequantize(const uint8_t *d_bits, uint64_t n, float scale, int format, float *d_out, hipStream_t st);
int launch_fp8_one_to_many(const float *d_query, uint32_t dim, const uint8_t *d_db, uint64_t stride, uint64_t n,
                           int dot, float *d_out, hipStream_t st);
int launch_rows_i8_build(const float *d_rows, uint64_t n, uint32_t dim, uint32_t stride, int8_t *d_rows8,
                         void *d_meta, hipStream_t stream, float uni_scale = 0.0f);

int launch_lut16_quantize(const float *d_tables, uint32_t S, uint8_t *d_lut8, float *d_bias_mult,
                          hipStream_t stream);

// Operand planes of the sparse-MFMA prefilter (txh.hip K6e) from the packed 4-bit codes: words per point and the
// build kernel.
static inline uint32_t sp_words(uint32_t S) { return 4u * ((((S - 4u) / 4u + 1u) + 7u) / 8u); }
int launch_codes_sp_build(const uint32_t *d_codes, uint64_t n, uint32_t S, uint32_t *d_codes_sp, hipStream_t stream);

int launch_encode(const float *d_codebook, uint32_t S, uint32_t K, uint32_t dsub,
                  const float *d_rows, uint64_t n, uint32_t stride, const float *d_centers,
                  const uint32_t *d_leaf_of_row, uint8_t *d_out, hipStream_t stream);

}  // namespace scann
