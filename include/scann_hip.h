/*
 * scann_hip.h -- C ABI of libscann_hip.so: the MI355X (gfx950) implementation
 * of the sunbains/scann-rust Tree-X-Hybrid / brute-force hot path.
 *
 * The reference crate has no FFI of its own (pure Rust); the
 * drop-in boundary is its Rust API.  Each entry point below names the Rust
 * item(s) whose body it replaces (paths relative to /root/reference/src); the
 * Rust-side `extern "C"` block a maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types.
 *  - every function returns a scann_hip_status (== ErrorCode discriminant in
 *    declaration order, error.rs:10-45); scann_hip_last_error() returns the
 *    thread-local message of the last failure on the calling thread.
 *  - *_create copies host arrays to the device; the caller keeps ownership of
 *    its host memory and may free it on return.  Handles own device memory until
 *    scann_hip_index_destroy.
 *  - host entry points are synchronous (outputs filled on return).  The *_device
 *    variants take device pointers + a hipStream_t (passed as void*), enqueue
 *    only, and never synchronise: inputs already resident in HBM.
 *  - result rows are ascending by distance; rows shorter than k are reported via
 *    out_count (the reference returns shorter Vecs: brute_force/searcher.rs:91,
 *    tree_x_hybrid/mod.rs:360-363).  Unused slots: idx 0xFFFFFFFF, dist +inf.
 *  - any thread may call search functions concurrently on one handle (Searcher:
 *    Send + Sync, searcher.rs:148): host-side searches draw a stream + workspace from a small
 *    per-handle pool (SCANN_HIP_SEARCH_SLOTS, default 4) and run side by side.  create/destroy need
 *    external synchronisation.
 *  - device entry points and streams: every *_device search call works in a per-handle workspace that is
 *    bound to the CALLER'S STREAM (SCANN_HIP_DEVICE_SLOTS workspaces per handle, default 2, at most 4; the
 *    first one is the handle's primary workspace).  Calls enqueued on the same stream reuse its workspace in
 *    stream order; calls on different streams use different workspaces and may overlap on the device -- a
 *    caller alternating two streams runs batch i+1's scan under batch i's re-rank.  With more streams than
 *    workspaces the least recently used workspace changes hands, and the library orders the new call behind
 *    the old stream's last call with an event (correct, no overlap for that pair).
 *    scann_hip_index_last_device_status(index, stream) reports the calls enqueued on that stream.
 */
#ifndef SCANN_HIP_H
#define SCANN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error.rs:10-45 (ErrorCode, declaration order) */
typedef enum {
    SCANN_HIP_OK = 0,
    SCANN_HIP_CANCELLED = 1,
    SCANN_HIP_UNKNOWN = 2,
    SCANN_HIP_INVALID_ARGUMENT = 3,
    SCANN_HIP_DEADLINE_EXCEEDED = 4,
    SCANN_HIP_NOT_FOUND = 5,
    SCANN_HIP_ALREADY_EXISTS = 6,
    SCANN_HIP_PERMISSION_DENIED = 7,
    SCANN_HIP_RESOURCE_EXHAUSTED = 8,
    SCANN_HIP_FAILED_PRECONDITION = 9,
    SCANN_HIP_ABORTED = 10,
    SCANN_HIP_OUT_OF_RANGE = 11,
    SCANN_HIP_UNIMPLEMENTED = 12,
    SCANN_HIP_INTERNAL = 13,
    SCANN_HIP_UNAVAILABLE = 14,
    SCANN_HIP_DATA_LOSS = 15,
    SCANN_HIP_UNAUTHENTICATED = 16
} scann_hip_status;

/* distance_measures/mod.rs:32-66: the measures the hot path dispatches on
 * (brute_force/searcher.rs:119-138). */
typedef enum {
    SCANN_HIP_SQUARED_L2 = 0,
    SCANN_HIP_L2 = 1,
    SCANN_HIP_DOT_PRODUCT = 2, /* distance = -dot (simd/x86.rs:247-250) */
    /* The two measures below go through DistanceMeasure::distance one pair at a time in the reference
     * (brute_force/searcher.rs:131-137, scann.rs:238-246, utils/reordering.rs:35-44).  They are valid for
     * brute-force handles and as the distance_measure of Scann-facade indexes (SearchMode::Partitioned,
     * exact reordering); TreeXHybridSearcher / AsymmetricHasher re-rank by squared L2 regardless. */
    SCANN_HIP_L1 = 3,          /* l1_distance_avx2, simd/x86.rs:103-132 */
    SCANN_HIP_COSINE = 4       /* 1 - cosine similarity, one_to_one.rs:559-612; its horizontal sums are the
                                * third-party wide::f32x8::reduce_add (wide 0.7, non-AVX build): restated,
                                * parity unpinned by the reference */
} scann_hip_measure;

typedef struct scann_hip_ctx scann_hip_ctx;     /* one device + stream/workspace pool */
typedef struct scann_hip_index scann_hip_index; /* one searcher */

/* ---- context ------------------------------------------------------------ */
int scann_hip_init(int device_id, scann_hip_ctx **out_ctx);
void scann_hip_shutdown(scann_hip_ctx *ctx);
const char *scann_hip_last_error(void);
const char *scann_hip_version(void);
/* ABI guard for bindings written by hand (Rust #[repr(C)], ctypes): fills out[0..n) with
 * { sizeof(scann_hip_txh_desc), offsetof(.., distance_measure), sizeof(scann_hip_search_opts),
 *   offsetof(.., bf_exact), sizeof(scann_hip_file_info), offsetof(.., has_data) } and returns the
 * number of values defined (6).  A binding asserts these against its own layout once at start-up. */
uint32_t scann_hip_abi_layout(uint32_t *out, uint32_t n);
/* data_format/dataset.rs:90-96 (DenseDataset::compute_stride for f32) */
uint32_t scann_hip_compute_stride(uint32_t dim);

/* ---- brute force --------------------------------------------------------- */
/* Replaces BruteForceSearcher::new / with_shared_dataset
 * (brute_force/searcher.rs:34-54).  data: n rows of `stride` floats, the
 * DenseDataset::raw_data() buffer (data_format/dataset.rs:176-179, 228-230).
 * n == 0 is legal (searches return empty rows, searcher.rs:78-80). */
int scann_hip_bf_create(scann_hip_ctx *ctx, const float *data, uint64_t n, uint32_t dim,
                        uint32_t stride, int measure, scann_hip_index **out_index);

/* ---- Tree-X-Hybrid / AsymmetricHasher ------------------------------------ */
/* The trained index TreeXHybridSearcher::build (tree_x_hybrid/mod.rs:131-209)
 * or AsymmetricHasher::build (hashes/hasher.rs:109-134) produced, flattened:
 *   centers       [num_partitions][dim]  TreePartitioner.centers (tree_partitioner.rs:29)
 *   leaf_offsets  [num_partitions+1]     CSR over PartitionData.indices (mod.rs:81-90)
 *   leaf_ids      [n_local]              datapoint index of CSR row i
 *   codebook      [S][K][dsub]           Codebook.subspaces[s].centroids (codebook.rs:59-67)
 *   codes         CSR row order; unpacked [n_local][S] u8 (PartitionData.encoded) or
 *                 packed 4-bit [n_local][ceil(S/2)] (PackedCodes4Bit, lut16.rs:43-61)
 *   data          [n_rows][stride] original rows indexed by DATAPOINT index (re-rank,
 *                 mod.rs:342-364); may be NULL (AsymmetricHasher::build_no_store,
 *                 hasher.rs:137-159): exact re-ordering then fails FailedPrecondition.
 * num_partitions == 0 selects AsymmetricHasher mode: one implicit leaf holding all
 * n points in datapoint order (leaf_offsets / leaf_ids / centers ignored).
 *
 * Multi-GPU leaf sharding (one process per GPU): each rank passes only the leaves
 * it owns (unowned leaves have zero local length) plus leaf_sizes_global, so every
 * rank derives identical merge keys; data then holds only the local rows in CSR row
 * order (data_is_csr_order = 1).  leaf_sizes_global == NULL means unsharded.
 *
 * Scann facade modes (scann.rs:181-294) are configurations of the same index:
 *   SearchMode::TreeAH       use_residuals = 0, codebook trained on the raw rows, searched with
 *                            pre_reorder_k = k (and exact_reorder = 0, or 1 for the exact
 *                            reordering of the k-truncated list, scann.rs:199-209);
 *   SearchMode::Hashed       num_partitions = 0, same options;
 *   SearchMode::Partitioned  codebook == NULL && codes == NULL && num_subspaces == 0: every row
 *                            of the selected leaves is scored exactly with distance_measure
 *                            (search_partitioned, scann.rs:213-252); needs data, unsharded.
 * distance_measure (SCANN_HIP_SQUARED_L2 = 0 / L2 / DOT_PRODUCT / L1 / COSINE) is the measure of the exact
 * re-ordering (ReorderingHelper, utils/reordering.rs:23-54) and of the Partitioned scan;
 * TreeXHybridSearcher and AsymmetricHasher always re-rank by squared L2 (mod.rs:350-358). */
typedef struct {
    const float *data;
    uint64_t n_rows;
    uint32_t dim;
    uint32_t stride;
    int32_t data_is_csr_order;
    const float *centers;
    uint32_t num_partitions;
    const uint32_t *leaf_offsets;
    const uint32_t *leaf_ids;
    const uint32_t *leaf_sizes_global;
    uint64_t n_local;
    const float *codebook;
    uint32_t num_subspaces;      /* S */
    uint32_t num_codes;          /* K <= 16: LUT16 (4-bit codes, S in {8,16,24,32,48,64});
                                  * 16 < K <= 256: byte codes (S in {4,8,16}) */
    uint32_t dims_per_subspace;  /* dsub; S * dsub must equal dim (codebook.rs:154-159) */
    const uint8_t *codes;
    int32_t codes_packed4;
    int32_t use_residuals;            /* TreeXHybridConfig.use_residuals (mod.rs:31) */
    uint32_t partitions_to_search;    /* TreeXHybridConfig.partitions_to_search (mod.rs:27) */
    float pre_reorder_multiplier;     /* TreeXHybridConfig.pre_reorder_multiplier (mod.rs:33) */
    int32_t distance_measure;         /* ScannConfig.distance_measure (config.rs:32-36); 0 = SquaredL2 */
} scann_hip_txh_desc;

/* Replaces the search side of TreeXHybridSearcher::build / AsymmetricHasher::build.
 * Errors: n_local == 0 -> InvalidArgument (mod.rs:132-134, hasher.rs:110-112);
 * dim % S != 0 -> InvalidArgument (codebook.rs:154-159).
 * The arrays' CONTENTS are validated too, so that no search can read out of bounds: leaf_offsets
 * monotone and spanning [0, n_local]; every leaf_ids[i] < n_rows when data is indexed by datapoint
 * (data != NULL, data_is_csr_order == 0); n_rows >= n_local when rows are read in CSR order
 * (AsymmetricHasher mode with data); leaf_sizes_global[l] >= the local length of leaf l; every code
 * < num_codes (packed or not).  Violations -> InvalidArgument here, DataLoss from
 * scann_hip_index_load_file (a file whose sizes are consistent but whose contents are not). */
int scann_hip_txh_create(scann_hip_ctx *ctx, const scann_hip_txh_desc *desc,
                         scann_hip_index **out_index);

typedef struct {
    /* 0 = the index default.  SearchParameters.num_leaves_to_search is ignored by
     * the reference searcher; this knob exists for sweeps. */
    uint32_t partitions_to_search;
    /* candidates kept by approximate distance before re-ranking.
     * 0 = (k as f32 * pre_reorder_multiplier) as usize (mod.rs:263). */
    uint32_t pre_reorder_k;
    /* 1 (default for Tree-X-Hybrid; AsymmetricHasher::search_with_reordering,
     * hasher.rs:188-229): exact SquaredL2 re-rank of the pre_reorder_k candidates.
     * 0: return the k best by APPROXIMATE distance (AsymmetricHasher::search,
     * hasher.rs:162-185); pre_reorder_k is then ignored. */
    int32_t exact_reorder;
    /* optional per-stage outputs for parity checks (host pointers or NULL):
     *   tokens/token_dists [nq][P]      TreePartitioner::partition result (tree_partitioner.rs:196-229)
     *   cand_idx/cand_dist [nq][m], cand_count [nq]   merged candidates before re-rank (mod.rs:283-290) */
    uint32_t *tokens;
    float *token_dists;
    uint32_t *cand_idx;
    float *cand_dist;
    uint32_t *cand_count;
    /* TreeXHybridSearcher::search_with_filter(query, k, Some(filter)) for allow-list filters
     * (tree_x_hybrid/mod.rs:245-250, 327-332; restricts/allowlist.rs): bit i set = datapoint i
     * may be returned; disallowed points are skipped before scoring.  NULL = no filter.  Host
     * pointer in the host entry points, DEVICE pointer in the *_device entry points.
     * allow_bitmap_bits = the bitmap's capacity (a multiple-of-64-bit allocation is read);
     * datapoint indices >= capacity are not allowed (allowlist.rs:97-100).  Any other
     * `dyn RestrictFilter` is served by materialising is_allowed(0..n) into such a bitmap
     * (what scann.hpp's search_with_filter does). */
    const uint64_t *allow_bitmap;
    uint64_t allow_bitmap_bits;
    /* Brute-force handles.  0 (default): large batches on large indexes take the bf16-shortlist
     * path (bf16 MFMA scores shortlist 4k rows per query, the reference's f32 arithmetic re-scores
     * them, and an error bound proves that no other row can enter the top k; results are
     * bit-identical to the exact kernels).  A query whose result cannot be proven sets status
     * Aborted: the host entry point repeats the batch on the exact kernels by itself, callers of the
     * *_device entry point repeat it with bf_exact = 1.  1: exact kernels only. */
    int32_t bf_exact;
} scann_hip_search_opts;

void scann_hip_search_opts_default(scann_hip_search_opts *opts);

/* ---- search ---------------------------------------------------------------- */
/* Replaces, per index kind:
 *   BruteForceSearcher::{search, search_batched}       brute_force/searcher.rs:77-208
 *   AsymmetricHasher::{search, search_with_reordering, search_batched}  hashes/hasher.rs:162-238
 *   TreeXHybridSearcher::{search, search_with_filter(None)} + Searcher::search_batched_with_params
 *                                                       tree_x_hybrid/mod.rs:240-294, 382-409
 * queries: nq rows, row i at queries + i*q_stride, q_dim valid floats each.
 * q_dim != index dim -> InvalidArgument (searcher.rs:83-89, hasher.rs:167-171,
 * mod.rs:251-253).  out_idx/out_dist: [nq][k]; out_count: [nq]. */
int scann_hip_search_batched(scann_hip_index *index, const float *queries, uint32_t nq,
                             uint32_t q_stride, uint32_t q_dim, uint32_t k,
                             const scann_hip_search_opts *opts, uint32_t *out_idx,
                             float *out_dist, uint32_t *out_count);

/* Searcher::search_batched_with_params (searcher.rs:148-186; tree_x_hybrid/mod.rs:383-409, hashes/hasher.rs,
 * brute_force/searcher.rs: one SearchParameters per query, of which the searchers on this path read
 * num_neighbors): query i is searched with k_per_query[i] -- and therefore with ITS OWN pre-reorder candidate
 * count k_i * pre_reorder_multiplier when opts->pre_reorder_k is 0 -- exactly as a single search(query, k_i)
 * would.  Queries that share a k travel as one batch.  Row i of out_idx / out_dist starts at i * out_pitch
 * (out_pitch >= the largest k); slots past out_count[i] hold idx 0xFFFFFFFF, dist +inf. */
int scann_hip_search_batched_params(scann_hip_index *index, const float *queries, uint32_t nq,
                                    uint32_t q_stride, uint32_t q_dim, const uint32_t *k_per_query,
                                    const scann_hip_search_opts *opts, uint32_t out_pitch,
                                    uint32_t *out_idx, float *out_dist, uint32_t *out_count);

/* Same, all pointers device-resident, enqueued on `hip_stream`, no sync.  The caller
 * must first reserve workspace for the largest batch it will submit. */
int scann_hip_index_reserve(scann_hip_index *index, uint32_t max_nq, uint32_t max_k,
                            const scann_hip_search_opts *opts);
int scann_hip_search_batched_device(scann_hip_index *index, const float *d_queries,
                                    uint32_t nq, uint32_t q_stride, uint32_t k,
                                    const scann_hip_search_opts *opts, uint32_t *d_out_idx,
                                    float *d_out_dist, uint32_t *d_out_count,
                                    void *hip_stream);
/* Device status word of the last *_device call on this index (0 = ok, else a
 * scann_hip_status: candidate-buffer overflow -> ResourceExhausted).  Synchronises
 * the stream. */
int scann_hip_index_last_device_status(scann_hip_index *index, void *hip_stream);

/* ---- multi-GPU: leaf-sharded Tree-X-Hybrid ------------------- */
/* Local stage: this rank's best-m candidates per query by approximate distance, with
 * their exact distances, as (merge key u64, datapoint idx u32, exact f32) triples
 * [nq][m] (+ count [nq]).  The merge key orders candidates exactly as the reference's
 * flatten + stable sort does (mod.rs:283-290) and is identical on every rank.
 * A candidate that provably cannot be among the k best exact distances of ANY prefix of
 * this rank's list (k candidates with smaller merge keys are nearer: int8 row brackets,
 * lists of > 512 candidates, SquaredL2) carries exact = +inf instead of its distance; it
 * still counts as one of the m candidates.  Merging (scann_hip_txh_merge_device, with a
 * num_neighbors <= the k of this call) returns the same rows as with every distance filled in. */
int scann_hip_txh_search_local_device(scann_hip_index *index, const float *d_queries,
                                      uint32_t nq, uint32_t q_stride, uint32_t k,
                                      const scann_hip_search_opts *opts, uint64_t *d_keys,
                                      uint32_t *d_idx, float *d_exact, uint32_t *d_count,
                                      void *hip_stream);
/* Merge stage on the gathered triples of `world` ranks ([world][nq][m_local] each):
 * stable sort by key -> truncate m -> stable sort by exact -> truncate k
 * (mod.rs:289-290, 360-361).  m_local is the per-rank pre_reorder_k of the local stage.
 * m_local == m is exact by construction.  m_local < m (a random shard holds ~m/world of the
 * global best m) is verified: if a truncated rank could have held more members,
 * *d_status (caller-zeroed device word, may be NULL) is raised to Aborted and the caller
 * re-runs with m_local = m.
 * rank_stride_bytes: 0 for dense arrays; otherwise the four pointers address rank 0's
 * sections of a packed per-rank buffer and rank g's sections lie g * rank_stride_bytes
 * further (one all_gather of [keys | idx | exact | count] instead of four). */
int scann_hip_txh_merge_device(scann_hip_ctx *ctx, uint32_t world, uint32_t nq, uint32_t m_local,
                               uint32_t m, uint32_t k, uint64_t rank_stride_bytes,
                               const uint64_t *d_keys,
                               const uint32_t *d_idx, const float *d_exact,
                               const uint32_t *d_count, uint32_t *d_out_idx, float *d_out_dist,
                               uint32_t *d_out_count, uint32_t *d_status, void *hip_stream);
/* Exchange by all_to_all instead of all_gather (xGMI is point-to-point: every rank sends each peer
 * only the candidates of the queries that peer merges).  Repacks the local stage's arrays
 * [nq][m_local] into `world` destination blocks, block d = the queries [d*nq/world, (d+1)*nq/world):
 * [keys u64 | idx u32 | exact f32 | count u32], block_bytes apart (>= (nq/world)*(16*m_local + 4),
 * multiple of 8).  After all_to_all_single (equal splits) a rank holds one block per source rank
 * and calls scann_hip_txh_merge_device with nq/world queries and rank_stride_bytes = block_bytes.
 * nq must be a multiple of world. */
int scann_hip_txh_pack_blocks_device(scann_hip_ctx *ctx, uint32_t world, uint32_t nq, uint32_t m_local,
                                     const uint64_t *d_keys, const uint32_t *d_idx, const float *d_exact,
                                     const uint32_t *d_count, void *d_out, uint64_t block_bytes,
                                     void *hip_stream);

/* Greedy size-balanced leaf->rank assignment used by the harness (not in the reference). */
int scann_hip_assign_leaves(const uint32_t *leaf_sizes, uint32_t num_partitions,
                            uint32_t world, uint32_t *out_owner);

/* ---- multi-GPU exchange inside the library: RCCL over xGMI --------------------
 * One process per GPU; every process holds a communicator.  A host that is not Python needs
 * nothing else: rank 0 calls scann_hip_comm_unique_id and hands the 128 bytes to the other ranks
 * over any side channel (a file, a socket, MPI, torch.distributed ...), every rank then calls
 * scann_hip_comm_create (collective).  librccl.so.1 is loaded on first use; without it these entry
 * points return Unavailable and everything else in the library keeps working.
 *
 * scann_hip_txh_search_sharded_device = the whole north-star step for a leaf-sharded index
 * (every rank created its shard with leaf_sizes_global set and data_is_csr_order = 1):
 *   local stage (this rank's best m_local candidates per query by merge key + their exact
 *   distances) -> one block per destination rank -> ONE all-to-all as grouped ncclSend/ncclRecv
 *   (xGMI is point-to-point: each peer receives only the candidates of the nq/world queries it
 *   merges) -> merge of this rank's queries (stable sort by key, truncate m, stable sort by exact,
 *   truncate k: tree_x_hybrid/mod.rs:283-293, 360-361) -> ncclAllGather of the k result rows.
 * All ranks pass the same queries, nq, k and options; every rank receives all nq result rows
 * ([nq][k] / [nq], device pointers).  m_local = 0 means m with destination blocks sized for the worst
 * case (exact by construction); m_local > 0 sends compact blocks (scann_hip_comm_layout) and is verified
 * by the merge -- a too-short list or an overflowing block gives scann_hip_comm_last_status -> Aborted on
 * every rank: repeat the batch with 0.
 * The exchange runs on the communicator's own stream; `hip_stream` carries the local stage and,
 * at the end of the call, waits for the results.  Internal buffers are double-buffered and ordered
 * with events, so a caller that alternates between two streams (and two sets of output buffers)
 * overlaps one step's exchange with the next step's local stage. */
typedef struct scann_hip_comm scann_hip_comm;
#define SCANN_HIP_UNIQUE_ID_BYTES 128
int scann_hip_comm_unique_id(void *out_id /* SCANN_HIP_UNIQUE_ID_BYTES */);
int scann_hip_comm_create(scann_hip_ctx *ctx, const void *unique_id, int rank, int world,
                          scann_hip_comm **out_comm);
void scann_hip_comm_destroy(scann_hip_comm *comm);
int scann_hip_txh_search_sharded_device(scann_hip_index *index, scann_hip_comm *comm,
                                        const float *d_queries, uint32_t nq, uint32_t q_stride,
                                        uint32_t k, const scann_hip_search_opts *opts,
                                        uint32_t m_local, uint32_t *d_out_idx, float *d_out_dist,
                                        uint32_t *d_out_count, void *hip_stream);
/* Byte layout of one sharded step for nq queries (no GPU needed; for hosts that size their own
 * buffers and for the protocol tests): out[0..16) = { qr = queries merged per rank (the batch is
 * padded to qr * world), nq_pad, block_bytes (one destination block = the bytes a rank sends each peer
 * over its xGMI link per step), offset of idx, of exact, of count inside a block, bytes of
 * the local-stage arrays [nq][m_local] and the offsets of idx, exact, count inside them, bytes of
 * the result rows ([nq_pad][k] idx | dist | [nq_pad] count | [world] status) and the offset of dist,
 * offset of keys inside a block, cap = entries a block has room for, offset of the block's overflow flag,
 * offset of the status words inside the result rows }.
 * A destination block is COMPACT: [count u32[qr] | overflow flag u32 | pad to 16 | keys u64[cap] | idx u32[cap] |
 * exact f32[cap]], the entries of the block's queries one behind the other.  cap = min(qr * m_local, max(m_local,
 * ceil(fill * qr * m_local / world))) with fill = SCANN_HIP_COMM_FILL (default 2.5): the candidates of one query total
 * about m over ALL ranks, so a peer's share averages m_local / world per query.  A block that overflows is flagged,
 * scann_hip_comm_last_status then returns Aborted on EVERY rank (the status words are gathered with the rows), and
 * the caller repeats the batch with m_local = 0, which sizes the blocks for the worst case (cap = qr * m).
 * This function reports the layout of calls with m_local > 0. */
int scann_hip_comm_layout(uint32_t nq, uint32_t world, uint32_t m_local, uint32_t k, uint64_t *out16);
/* Status of the merges since the last call of this function (0 = ok, Aborted = an m_local < m list
 * was too short or a compact block overflowed -- the same answer on every rank); synchronises the
 * communicator's stream and clears the word. */
int scann_hip_comm_last_status(scann_hip_comm *comm);

/* ---- index files ---------------------------------------------------------
 * The reference keeps indexes in memory only (no save/load: SURVEY section 5); its arrays are the
 * DenseDataset buffer (data_format/dataset.rs:46-61), TreePartitioner.centers and the partition
 * lists (partitioning/partitioner.rs:132-142, tree_x_hybrid/mod.rs:81-90), the Codebook and the
 * per-point codes.  One little-endian container holds exactly the fields of scann_hip_txh_desc /
 * the arguments of scann_hip_bf_create, so the GPU library, the CPU oracle (numpy reader:
 * scann_rust_amd/index_file.py) and the golden fixtures share it:
 *
 *   [0, 256)   header   "SCANNIDX", version 1, kind (0 brute force, 1 tree / hasher index), the
 *                        scalar descriptor fields, section count, file size
 *   [256, ...) table    64 bytes per section: name[24], dtype (0 f32, 1 u32, 2 u8), offset, bytes
 *   sections   "data" "centers" "leaf_offsets" "leaf_ids" "leaf_sizes_global" "codebook" "codes",
 *              each starting on a 4096-byte boundary (absent arrays have no section)
 *
 * write: host arrays -> file.  load: the file is mmap'ed (never read into a second host copy), the
 * mapping pinned for DMA when the driver allows it (SCANN_HIP_LOAD_PIN=0 skips that), and the
 * arrays uploaded by the same code as scann_hip_*_create.  Errors: missing file -> NotFound; bad
 * magic / version -> InvalidArgument; truncated or inconsistent file -> DataLoss. */
typedef struct {
    uint32_t version;
    uint32_t kind;                 /* 0 = brute force, 1 = tree / hasher index */
    uint64_t n_rows, n_local, file_bytes;
    uint32_t dim, stride, num_partitions, num_subspaces, num_codes, dims_per_subspace;
    int32_t distance_measure, data_is_csr_order, codes_packed4, use_residuals;
    uint32_t partitions_to_search;
    float pre_reorder_multiplier;
    int32_t has_data;              /* exact re-ordering possible */
} scann_hip_file_info;
int scann_hip_txh_write_file(const char *path, const scann_hip_txh_desc *desc);
int scann_hip_bf_write_file(const char *path, const float *data, uint64_t n, uint32_t dim,
                            uint32_t stride, int measure);
int scann_hip_index_file_info(const char *path, scann_hip_file_info *out_info);
int scann_hip_index_load_file(scann_hip_ctx *ctx, const char *path, scann_hip_index **out_index);

/* ---- building blocks exposed for parity tests / callers ---------------------- */
/* TreePartitioner::partition for a batch (tree_partitioner.rs:196-229). */
int scann_hip_txh_partition(scann_hip_index *index, const float *queries, uint32_t nq,
                            uint32_t q_stride, uint32_t q_dim, uint32_t num_partitions,
                            uint32_t *out_tokens, float *out_dists, uint32_t *out_count);
/* LookupTable::from_query (hashes/lut.rs:47-70) for nq queries; leaf_for_query NULL =
 * no residual, else residual against centers[leaf_for_query[i]] (mod.rs:309-316).
 * out_lut: [nq][S][K]. */
int scann_hip_lut_from_query(scann_hip_index *index, const float *queries, uint32_t nq,
                             uint32_t q_stride, const uint32_t *leaf_for_query,
                             float *out_lut);
/* LookupTable::compute_distance over every local point (hashes/lut.rs:74-82,
 * hasher.rs:179-182) for nq explicit f32 LUTs [nq][S][K]: out [nq][n_local] in CSR
 * row order. */
int scann_hip_adc_distances(scann_hip_index *index, const float *luts, uint32_t nq,
                            float *out_dist);
/* Lut16SimdTables::compute_distances_batch (hashes/lut16_simd.rs:119-141 ->
 * simd/dispatch.rs:259-295): u8 tables [S][16], packed codes [n][ceil(S/2)];
 * out[i] = sum_u32 * multiplier + bias * S. */
int scann_hip_lut16_distances_batch(scann_hip_ctx *ctx, const uint8_t *packed_codes,
                                    const uint8_t *lut8, uint32_t num_subspaces, uint64_t n,
                                    float bias, float multiplier, float *out);
/* Lut16SimdTables::from_float_tables (hashes/lut16_simd.rs:39-90): global min / max over the
 * S x 16 f32 tables, scale = 255 / range, lut8 = round((v - min) * scale) as u8 (round half away
 * from zero, saturating), bias = min, multiplier = 1 / scale; range < 1e-10 -> scale = multiplier
 * = 1.  S == 0 -> bias 0, multiplier 1, nothing written.  Runs on the device (one workgroup). */
int scann_hip_lut16_quantize(scann_hip_ctx *ctx, const float *tables, uint32_t num_subspaces,
                             uint8_t *out_lut8, float *out_bias, float *out_multiplier);
/* The reference's FP8 codec and quantizer (quantization/fp8.rs:80-268) -- its own bit-level conversion,
 * not a hardware format table: exponent bias 7 (E4M3) / 15 (E5M2); a mantissa carry wraps without bumping
 * the exponent; the top exponent field only encodes the maximum (0x7E / 0x7C: overflow, infinity, NaN);
 * values under the smallest normal flush to signed zero.
 *   quantize:   out[i] = from_f32(values[i] * scale)      (Quantizer::quantize over Fp8Quantizer, :247-255)
 *   dequantize: out[i] = to_f32(bits[i]) / scale          (:257-264)
 * (calibrate_scale, :238-244, is fp8_max / max(max_abs, 1e-10) with fp8_max 448 / 57344: a host one-liner.) */
#define SCANN_HIP_FP8_E4M3 0
#define SCANN_HIP_FP8_E5M2 1
int scann_hip_fp8_quantize(scann_hip_ctx *ctx, const float *values, uint64_t n, float scale, int format,
                           uint8_t *out_bits);
int scann_hip_fp8_dequantize(scann_hip_ctx *ctx, const uint8_t *bits, uint64_t n, float scale, int format,
                             float *out_values);
/* one_to_many_fp8_float_squared_l2 / one_to_many_fp8_float_dot_product
 * (distance_measures/one_to_many_asymmetric.rs:327-377): f32 query against E4M3 rows [num_points][stride],
 * one sequential f32 sum per row; measure SCANN_HIP_SQUARED_L2 or SCANN_HIP_DOT_PRODUCT (negated), others
 * Unimplemented.  Bit-identical to the reference's loops.  The same codec, with a per-row calibrate_scale,
 * is the optional FP8 row store of the re-rank filter (SCANN_HIP_RERANK_STORE=fp8 at index creation). */
int scann_hip_fp8_distances(scann_hip_ctx *ctx, const float *query, uint32_t dim, const uint8_t *database,
                            uint64_t stride, uint64_t num_points, int measure, float *out_distances);
/* Codebook::encode over rows (hashes/codebook.rs:82-95, 205-215); optional residual
 * against centers[leaf_of_row[i]] (tree_x_hybrid/mod.rs:177-189).  out_codes [n][S]. */
int scann_hip_encode(scann_hip_ctx *ctx, const float *codebook, uint32_t num_subspaces,
                     uint32_t num_codes, uint32_t dims_per_subspace, const float *rows,
                     uint64_t n, uint32_t stride, const float *centers,
                     const uint32_t *leaf_of_row, uint8_t *out_codes);
/* one_to_many_{squared_l2,dot_product}_strided for a batch = the dense Q x N matrix of
 * batch_squared_l2_simd / batch_dot_product_simd (distance_measures/one_to_many.rs:228-373,
 * many_to_many.rs:301-373).  out [nq][n]. */
int scann_hip_bf_distances(scann_hip_index *index, const float *queries, uint32_t nq,
                           uint32_t q_stride, float *out);

/* ---- index build on the GPU ------------------------------------------
 * K-means over the rows of a brute-force index, or over the column window [col_offset,
 * col_offset + sub_dim) of them (per-subspace codebook training, src/hashes/codebook.rs:177-199).
 *
 * simd_threshold = KMeansConfig.simd_threshold (src/trees/kmeans.rs:43-46, default 128): distances
 *   over sub_dim >= simd_threshold values use the AVX2 summation order of squared_l2_f32 (8 FMA lane
 *   chains + fixed horizontal-sum tree + unfused scalar tail, simd/x86.rs:139-165), shorter ones the
 *   sequential scalar sum (:419-431).  0 = always the AVX2 order, UINT32_MAX = always sequential.
 * scann_hip_kmeans_init_pp: KMeans::kmeans_plusplus_init (src/trees/kmeans.rs:295-349); centers_out
 *   [k][sub_dim].  DEVIATION (seeding parity is unpinned by construction): the reference draws from
 *   rand::StdRng, which is not reproducible here -- a documented splitmix64 stream is
 *   used instead -- and its D^2 sampling sums min_d sequentially in f32 (:318-331), an n-long
 *   dependent chain per seed; here the total and the cumulative search run in f64 with a fixed
 *   reduction tree.  The minimum distances themselves follow the reference's arithmetic.
 * scann_hip_kmeans_lloyd: the Lloyd loop of KMeans::fit_single (:210-263) from the caller's initial
 *   centres (updated in place): assign_clusters (strict '<', lowest index on ties), inertia = the f64
 *   sum of the minimum distances in datapoint order (:376; computed by a reduction tree when every
 *   partial sum is exactly representable -- then all orders agree -- and by the sequential chain
 *   otherwise), stop when |prev - inertia| / (prev + 1e-10) < convergence_threshold, update_centers
 *   (f64 sums in ascending datapoint order, mean cast to f32, empty cluster c takes row c % n), then
 *   the final assignment.  Bit-identical to the reference's loop from the same initial centres.
 *   Outputs may be NULL. */
int scann_hip_kmeans_init_pp(scann_hip_index *bf_index, uint32_t col_offset, uint32_t sub_dim, uint32_t k,
                             uint64_t seed, uint32_t simd_threshold, float *centers_out);
int scann_hip_kmeans_lloyd(scann_hip_index *bf_index, uint32_t col_offset, uint32_t sub_dim, float *centers,
                           uint32_t k, uint32_t max_iterations, double convergence_threshold,
                           uint32_t simd_threshold, uint32_t *out_assign, uint32_t *out_sizes,
                           double *out_inertia, uint32_t *out_iterations, int *out_converged);

/* BruteForceSearcher::search_radius (src/brute_force/searcher.rs:142-167) for one query: every
 * datapoint with distance <= radius, stable-sorted by distance.  Writes at most `capacity` rows;
 * *out_count receives the number found (call again with a larger capacity if it exceeds it).
 * Empty dataset -> OK with 0 rows; wrong q_dim -> InvalidArgument. */
int scann_hip_bf_search_radius(scann_hip_index *index, const float *query, uint32_t q_dim, float radius,
                               uint32_t *out_idx, float *out_dist, uint64_t capacity,
                               uint64_t *out_count);

/* Index build helper: for every row of a brute-force index, the nearest of
 * `num_centers` centres [num_centers][dim] under the partitioner's arithmetic --
 * TreePartitioner::partition(x, 1) as used by TreeXHybridSearcher::compute_residuals
 * (tree_x_hybrid/mod.rs:212-237) and KMeans::assign_clusters (trees/kmeans.rs:352-379):
 * sequential scalar SquaredL2, lowest centre index on ties.  out_dist may be NULL. */
int scann_hip_bf_assign_nearest(scann_hip_index *index, const float *centers, uint32_t num_centers,
                                uint32_t *out_assign, float *out_dist);

/* ---- introspection ------------------------------------------------------------- */
uint64_t scann_hip_index_size(const scann_hip_index *index);          /* Searcher::dataset_size */
uint32_t scann_hip_index_dimensionality(const scann_hip_index *index);/* Searcher::dimensionality */
void scann_hip_index_destroy(scann_hip_index *index);

/* Kernel timing hook for bench.py: mean HIP-event time (ms) of the dominant kernel over
 * the search launches issued on `index` since timing was enabled (ring of 64 event pairs
 * recorded on the stream each kernel was launched on; 0 if timing is off).  Enabling
 * resets the ring; reading synchronises on the recorded events. */
void scann_hip_index_enable_timing(scann_hip_index *index, int enable);
float scann_hip_index_last_kernel_ms(scann_hip_index *index, const char **out_kernel_name);

#ifdef __cplusplus
}
#endif
#endif
