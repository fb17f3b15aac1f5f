// txh.hip -- HIP kernels of the Tree-X-Hybrid / AsymmetricHasher search path (gfx950).
//
// Pipeline per query batch (reference: tree_x_hybrid/mod.rs:245-364):
//   centroid_scores -> select_leaves            TreePartitioner::partition
//   count/scan/fill worklist                    (groups (query, leaf) pairs by leaf)
//   lut_build                                   residual + LookupTable::from_query
//   sample_threshold                            valid upper bound of the m-th best key
//   adc_scan (dominant, LDS-staged LUT16)       LookupTable::compute_distance + FastTopNeighbors
//   select_rerank                               merge/sort/truncate + reorder_results
//
// All arithmetic that the reference performs in a fixed order is performed in the same
// order here; this file is compiled with -ffp-contract=off and uses fmaf() only where
// the reference uses _mm256_fmadd_ps.
#include <type_traits>

#include "txh.h"

namespace scann {

// =====================================================================================
// K1: centroid scores.  partitioning/tree_partitioner.rs:175-192: strictly sequential
// scalar sum of (q_j - c_j)^2, no FMA.  One thread per centroid, QT queries per block
// broadcast from LDS.
// =====================================================================================
template <int kCsQT>
__global__ __launch_bounds__(64) void centroid_scores_kernel(
    const float *__restrict__ centers, uint32_t L, uint32_t dim,
    const float *__restrict__ queries, uint32_t nq, uint32_t q_stride,
    float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [kCsQT][dim]
    const uint32_t q0 = blockIdx.y * kCsQT;
    for (uint32_t i = threadIdx.x; i < kCsQT * dim; i += blockDim.x) {
        uint32_t qi = i / dim, j = i - qi * dim;
        qs[i] = (q0 + qi < nq) ? queries[(size_t)(q0 + qi) * q_stride + j] : 0.0f;
    }
    __syncthreads();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= L) return;
    float acc[kCsQT];
#pragma unroll
    for (int qi = 0; qi < kCsQT; ++qi) acc[qi] = 0.0f;
    const float *crow = centers + (size_t)c * dim;
    if ((dim & 3u) == 0) {
        for (uint32_t j = 0; j < dim; j += 4) {
            const float4 cv = *reinterpret_cast<const float4 *>(crow + j);
#pragma unroll
            for (int qi = 0; qi < kCsQT; ++qi) {
                const float4 qv = *reinterpret_cast<const float4 *>(qs + qi * dim + j);
                float d0 = qv.x - cv.x, d1 = qv.y - cv.y, d2 = qv.z - cv.z, d3 = qv.w - cv.w;
                float a = acc[qi];
                a = a + d0 * d0;
                a = a + d1 * d1;
                a = a + d2 * d2;
                a = a + d3 * d3;
                acc[qi] = a;
            }
        }
    } else {
        for (uint32_t j = 0; j < dim; ++j) {
            const float cv = crow[j];
#pragma unroll
            for (int qi = 0; qi < kCsQT; ++qi) {
                float d = qs[qi * dim + j] - cv;
                acc[qi] = acc[qi] + d * d;
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < kCsQT; ++qi)
        if (q0 + qi < nq) out[(size_t)(q0 + qi) * L + c] = acc[qi];
}

// =====================================================================================
// K2: select leaves.  tree_partitioner.rs:206-228: stable sort of ALL L (dist, id) by
// OrderedFloat(dist), take the first P.  Key = (ordered(dist) << 32 | id) makes the
// stable order explicit; NaN sorts last as OrderedFloat does.
// =====================================================================================
// centers_inline != nullptr (small batches): the block scores the centroids itself first, with
// centroid_scores_kernel's arithmetic (sequential scalar sum of (q_j - c_j)^2, no FMA), instead of reading
// a matrix another launch produced.
// Bin of 1-based rank `rank` in a complete histogram of `bins` bins (1024 or 4096) in LDS (counts synchronised by the caller; the
// histogram holds >= rank entries); s_w[49] = the rank inside that bin.  Every thread of a kSelectThreads block calls; two
// barriers; s_w: LDS u32[>= 50].
__device__ __forceinline__ uint32_t block_hist_rank_bin(const uint32_t *s_hist, uint32_t *s_w, uint32_t rank,
                                                        uint32_t bins = kSelBinsMax) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t per = bins / kSelectThreads;   // bins per thread (1 or 4)
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per; ++j) mine += s_hist[tid * per + j];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if ((int)lane >= o) incl += up;
    }
    if (lane == 63) s_w[32 + wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w2 = 0; w2 < wave; ++w2) wbase += s_w[32 + w2];
    incl += wbase;
    const uint32_t excl = incl - mine;
    if (excl < rank && rank <= incl) {   // exactly one thread
        uint32_t c = excl;
        for (uint32_t j = 0; j < per; ++j) {
            const uint32_t h = s_hist[tid * per + j];
            if (c + h >= rank) {
                s_w[48] = tid * per + j;
                s_w[49] = rank - c;   // 1-based rank inside the bin
                break;
            }
            c += h;
        }
    }
    __syncthreads();
    return s_w[48];
}

// The kernel's body: also the first stage of small_fused_kernel, where EVERY workgroup of a query runs it
// (write_out: only one of them stores the global outputs; s_tok_out / s_vb_out: LDS copies of the tokens
// and the P + 1 key bases for the stages that follow in the same workgroup).
__device__ __forceinline__ void select_leaves_body(
    uint64_t *skeys, uint32_t q, bool write_out, uint32_t *s_tok_out, uint32_t *s_vb_out,
    float *__restrict__ cdist, uint32_t L, uint32_t n_pow2, uint32_t P, uint32_t p_pow2,
    const uint32_t *__restrict__ leaf_gsize, const uint32_t *__restrict__ leaf_off, uint32_t st,
    uint32_t *__restrict__ tokens, float *__restrict__ token_dists, uint32_t *__restrict__ vbase,
    uint32_t *__restrict__ sbase, const float *__restrict__ centers_inline, const float *__restrict__ queries,
    uint32_t q_stride, uint32_t dim, uint32_t centers_pitch) {
    uint64_t *s_top = skeys + n_pow2;                                   // [p_pow2] (select path)
    const SelCfg cfg = sel_cfg(L);
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_top + p_pow2);    // [cfg.bins]
    uint64_t *s_list = reinterpret_cast<uint64_t *>(s_hist + cfg.bins); // [cfg.list]
    uint64_t *s_red = s_list + cfg.list;                                // [48]
    uint32_t *s_scan = reinterpret_cast<uint32_t *>(s_red + 48);        // [3][kSelectThreads / 64] + cursor
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6, nwaves = nt >> 6;
    const bool select_path = p_pow2 != 0;
    const uint32_t nfill = select_path ? L : n_pow2;
    // inline mode (small batches): the leaf size tables of this index live in LDS behind the scan words,
    // so nothing after the scoring waits on global memory
#ifdef SCANN_WIDE_TIMING
    uint64_t tl[6];
    tl[0] = wall_clock64();
#endif
    uint32_t *s_lsz = s_scan + 64, *s_lgs = s_lsz + (centers_inline ? L : 0u);
    float *s_qv = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(s_lgs + (centers_inline ? L : 0u)) + 15u) & ~(uintptr_t)15u);
    if (centers_inline) {
        // centers_inline is the TRANSPOSED centroid matrix [dim][centers_pitch]: thread c reads element
        // (j, c), so a wave's load covers two cache lines instead of 64 (row-major rows made the kernel
        // wait on the L1's line-request rate: 15 us for 1000 x 128 centroids).  16 loads in flight; the sum is
        // the reference's sequential one.
        for (uint32_t j = tid; j < dim; j += nt) s_qv[j] = queries[(size_t)q * q_stride + j];
        for (uint32_t l = tid; l < L; l += nt) {
            s_lsz[l] = leaf_off[l + 1] - leaf_off[l];
            s_lgs[l] = leaf_gsize[l];
        }
        __syncthreads();
        for (uint32_t c = tid; c < L; c += nt) {
            const float *col = centers_inline + c;
            float acc = 0.0f;
            for (uint32_t j0 = 0; j0 < dim; j0 += 64) {   // 64 loads in flight: one memory round trip per 64 dims
                float cv[64];
#pragma unroll
                for (int u = 0; u < 64; ++u)
                    cv[u] = j0 + (uint32_t)u < dim ? col[(size_t)(j0 + u) * centers_pitch] : 0.0f;
#pragma unroll
                for (int u = 0; u < 64; ++u)
                    if (j0 + (uint32_t)u < dim) {
                        const float d = s_qv[j0 + u] - cv[u];
                        acc = acc + d * d;
                    }
            }
            if (write_out) cdist[(size_t)q * L + c] = acc;   // (kept for NaN payloads: see token_dists below)
            skeys[c] = ((uint64_t)((acc != acc) ? 0xFFFFFFFFu : f32_to_ordered(acc)) << 32) | c;
        }
        for (uint32_t i = L + tid; i < nfill; i += nt) skeys[i] = SCANN_KEY_MAX;
    } else {
        for (uint32_t i = tid; i < nfill; i += nt) {
            uint64_t key = SCANN_KEY_MAX;
            if (i < L) {
                float d = cdist[(size_t)q * L + i];
                uint32_t o = (d != d) ? 0xFFFFFFFFu : f32_to_ordered(d);
                key = ((uint64_t)o << 32) | i;
            }
            skeys[i] = key;
        }
    }
#ifdef SCANN_WIDE_TIMING
    tl[1] = wall_clock64();
#endif
    __syncthreads();
    const uint64_t *sorted = skeys;
#ifdef SCANN_WIDE_TIMING
    tl[2] = tl[3] = wall_clock64();
#endif
    if (select_path) {
        // P << L: the P-th smallest key by histogram select, then sort only the P survivors
        // (keys are unique: exactly P of them are <= the P-th smallest)
        // (Tried for P <= 16: per-wave tournaments of 64-bit wave minima, P rounds per wave and P over the finalists,
        // which also yields the sorted order -- 12 cross-lane shuffles per round, no faster than this.)
        const uint64_t T = block_select<uint64_t>(skeys, L, P, cfg, s_hist, s_list, s_red);
        for (uint32_t i = tid; i < p_pow2; i += nt) s_top[i] = SCANN_KEY_MAX;
        if (tid == 0) s_scan[48] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < L; i += nt) {
            const uint64_t key = skeys[i];
            if (key <= T) s_top[atomicAdd(&s_scan[48], 1u)] = key;
        }
        __syncthreads();
#ifdef SCANN_WIDE_TIMING
        tl[3] = wall_clock64();
#endif
        bitonic_sort_lds(s_top, p_pow2);
        sorted = s_top;
    } else {
        bitonic_sort_lds(skeys, n_pow2);
    }
#ifdef SCANN_WIDE_TIMING
    tl[4] = wall_clock64();
#endif
    for (uint32_t r = tid; r < P; r += nt) {
        uint64_t key = sorted[r];
        uint32_t id = (uint32_t)key;
        if (s_tok_out) s_tok_out[r] = id;
        if (write_out) {
            tokens[(size_t)q * P + r] = id;
            // inline mode: the distance is the key's high word (a NaN keeps its payload through cdist)
            const uint32_t ob = (uint32_t)(key >> 32);
            token_dists[(size_t)q * P + r] = (centers_inline && ob != 0xFFFFFFFFu) ? ordered_to_f32(ob)
                                                                                  : cdist[(size_t)q * L + id];
        }
    }
    // exclusive prefixes over the P tokens: global leaf sizes (merge-key base), sample counts
    // (every st-th local point of each selected leaf) and local points
    const uint32_t per = (P + nt - 1) / nt;
    const uint32_t r0 = min(P, tid * per), r1 = min(P, r0 + per);
    uint32_t a_g = 0, a_s = 0, a_t = 0;
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t leaf = (uint32_t)sorted[r];
        const uint32_t sz = centers_inline ? s_lsz[leaf] : leaf_off[leaf + 1] - leaf_off[leaf];
        a_g += centers_inline ? s_lgs[leaf] : leaf_gsize[leaf];
        a_s += (sz + st - 1) / st;
        a_t += sz;
    }
    uint32_t i_g = a_g, i_s = a_s, i_t = a_t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u_g = (uint32_t)__shfl_up((int)i_g, o), u_s = (uint32_t)__shfl_up((int)i_s, o),
                       u_t = (uint32_t)__shfl_up((int)i_t, o);
        if ((int)lane >= o) {
            i_g += u_g;
            i_s += u_s;
            i_t += u_t;
        }
    }
    if (lane == 63) {
        s_scan[wave] = i_g;
        s_scan[16 + wave] = i_s;
        s_scan[32 + wave] = i_t;
    }
    __syncthreads();
    uint32_t b_g = 0, b_s = 0, t_g = 0, t_s = 0, t_t = 0;
    for (uint32_t w2 = 0; w2 < nwaves; ++w2) {
        if (w2 < wave) {
            b_g += s_scan[w2];
            b_s += s_scan[16 + w2];
        }
        t_g += s_scan[w2];
        t_s += s_scan[16 + w2];
        t_t += s_scan[32 + w2];
    }
    uint32_t vb = b_g + i_g - a_g, sb = b_s + i_s - a_s;
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t leaf = (uint32_t)sorted[r];
        const uint32_t sz = centers_inline ? s_lsz[leaf] : leaf_off[leaf + 1] - leaf_off[leaf];
        if (s_vb_out) s_vb_out[r] = vb;
        if (write_out) {
            vbase[(size_t)q * (P + 1) + r] = vb;
            sbase[(size_t)q * (P + 2) + r] = sb;
        }
        vb += centers_inline ? s_lgs[leaf] : leaf_gsize[leaf];
        sb += (sz + st - 1) / st;
    }
    if (tid == 0) {
        if (s_vb_out) s_vb_out[P] = t_g;
        if (write_out) {
            vbase[(size_t)q * (P + 1) + P] = t_g;
            sbase[(size_t)q * (P + 2) + P] = t_s;
            sbase[(size_t)q * (P + 2) + P + 1] = t_t;
        }
    }
#ifdef SCANN_WIDE_TIMING
    tl[5] = wall_clock64();
    if (tid == 0 && write_out && q == 0) printf("leaves: score %.2f sync %.2f select %.2f sort %.2f prefix %.2f us\n", (tl[1] - tl[0]) * 0.01, (tl[2] - tl[1]) * 0.01, (tl[3] - tl[2]) * 0.01, (tl[4] - tl[3]) * 0.01, (tl[5] - tl[4]) * 0.01);
#endif
}

__global__ __launch_bounds__(kSelectThreads) void select_leaves_kernel(
    float *__restrict__ cdist, uint32_t L, uint32_t n_pow2, uint32_t P, uint32_t p_pow2,
    const uint32_t *__restrict__ leaf_gsize, const uint32_t *__restrict__ leaf_off, uint32_t st,
    uint32_t *__restrict__ tokens, float *__restrict__ token_dists, uint32_t *__restrict__ vbase,
    uint32_t *__restrict__ sbase, const float *__restrict__ centers_inline, const float *__restrict__ queries,
    uint32_t q_stride, uint32_t dim, uint32_t centers_pitch) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];    // [n_pow2] ...
    select_leaves_body(skeys, blockIdx.x, true, nullptr, nullptr, cdist, L, n_pow2, P, p_pow2, leaf_gsize, leaf_off, st,
                       tokens, token_dists, vbase, sbase, centers_inline, queries, q_stride, dim, centers_pitch);
}

// AsymmetricHasher mode: one implicit leaf (id 0) for every query.
__global__ void ah_tokens_kernel(uint32_t nq, const uint32_t *__restrict__ leaf_gsize,
                                 const uint32_t *__restrict__ leaf_off, uint32_t st,
                                 uint32_t *__restrict__ tokens, float *__restrict__ token_dists,
                                 uint32_t *__restrict__ vbase, uint32_t *__restrict__ sbase) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    tokens[q] = 0;
    token_dists[q] = 0.0f;
    vbase[2 * q] = 0;
    vbase[2 * q + 1] = leaf_gsize[0];
    const uint32_t sz = leaf_off[1] - leaf_off[0];
    sbase[3 * q] = 0;
    sbase[3 * q + 1] = (sz + st - 1) / st;
    sbase[3 * q + 2] = sz;
}

// =====================================================================================
// K3: worklist -- group (query, rank) pairs by leaf so that every leaf's codes are read
// once per batch and shared by all queries that selected it.
// =====================================================================================
// One launch instead of five memsets: zero the per-batch counters, mark all pair slots free.
__global__ void txh_init_kernel(uint32_t L, uint32_t nq, uint32_t max_slots,
                                uint32_t *__restrict__ leaf_cnt, uint32_t *__restrict__ leaf_cursor,
                                uint32_t *__restrict__ counters, uint32_t *__restrict__ cand_cnt,
                                uint32_t *__restrict__ cand32_cnt, uint32_t *__restrict__ pair_q) {
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    for (uint32_t i = i0; i < L; i += step) {
        leaf_cnt[i] = 0;
        leaf_cursor[i] = 0;
    }
    for (uint32_t i = i0; i < nq; i += step) {
        cand_cnt[i] = 0;
        if (cand32_cnt) cand32_cnt[i] = 0;
    }
    for (uint32_t i = i0; i < max_slots; i += step) pair_q[i] = kInvalid;
    for (uint32_t i = i0; i < CNT_WORDS; i += step) counters[i] = 0;
}

// ah != 0: one implicit leaf selected by every query -- no atomics (1024 same-address global
// atomics cost more than the whole LUT build).
__global__ void worklist_count_kernel(uint32_t npairs, int ah, const uint32_t *__restrict__ tokens,
                                      const uint32_t *__restrict__ leaf_off,
                                      uint32_t *__restrict__ leaf_cnt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (ah) {
        if (i == 0) leaf_cnt[0] = leaf_off[1] > leaf_off[0] ? npairs : 0u;
        return;
    }
    if (i >= npairs) return;
    uint32_t leaf = tokens[i];
    if (leaf_off[leaf + 1] > leaf_off[leaf]) atomicAdd(&leaf_cnt[leaf], 1u);
}

__global__ __launch_bounds__(1024) void worklist_scan_kernel(
    uint32_t L, const uint32_t *__restrict__ leaf_cnt, const uint32_t *__restrict__ leaf_off,
    uint32_t tp, uint32_t quads_per_tile, uint32_t chunks_per_tile, uint32_t stp, uint32_t st,
    uint32_t squads_per_tile,
    uint32_t *__restrict__ pair_off, uint32_t *__restrict__ tile_off,
    uint32_t *__restrict__ stile_off, uint32_t *__restrict__ counters) {
    __shared__ uint32_t s_pairs[1024], s_tiles[1024], s_stiles[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (L + 1023) / 1024;
    const uint32_t b = t * per, e = min(L, b + per);
    // tiles of the scan (all points) and of the sample pass (every st-th point)
    auto tiles_of = [&](uint32_t c, uint32_t pad, uint32_t sz, uint32_t *smp) {
        if (!c) { *smp = 0; return 0u; }
        const uint32_t ssz = (sz + st - 1) / st;
        *smp = ((ssz + stp - 1) / stp) * ((pad / 4 + squads_per_tile - 1) / squads_per_tile);
        const uint32_t nch = (sz + tp - 1) / tp;
        return ((nch + chunks_per_tile - 1) / chunks_per_tile) * ((pad / 4 + quads_per_tile - 1) / quads_per_tile);
    };
    uint32_t sp = 0, stl = 0, sst = 0;
    for (uint32_t l = b; l < e; ++l) {
        uint32_t c = leaf_cnt[l];
        uint32_t pad = (c + 3u) & ~3u;
        uint32_t sz = leaf_off[l + 1] - leaf_off[l];
        uint32_t smp;
        sp += pad;
        stl += tiles_of(c, pad, sz, &smp);
        sst += smp;
    }
    s_pairs[t] = sp;
    s_tiles[t] = stl;
    s_stiles[t] = sst;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        uint32_t a = 0, c2 = 0, c3 = 0;
        if (t >= off) {
            a = s_pairs[t - off];
            c2 = s_tiles[t - off];
            c3 = s_stiles[t - off];
        }
        __syncthreads();
        s_pairs[t] += a;
        s_tiles[t] += c2;
        s_stiles[t] += c3;
        __syncthreads();
    }
    uint32_t bp = s_pairs[t] - sp, bt = s_tiles[t] - stl, bs = s_stiles[t] - sst;
    for (uint32_t l = b; l < e; ++l) {
        uint32_t c = leaf_cnt[l];
        uint32_t pad = (c + 3u) & ~3u;
        uint32_t sz = leaf_off[l + 1] - leaf_off[l];
        uint32_t smp;
        pair_off[l] = bp;
        tile_off[l] = bt;
        stile_off[l] = bs;
        bp += pad;
        bt += tiles_of(c, pad, sz, &smp);
        bs += smp;
    }
    if (t == 1023) {
        pair_off[L] = s_pairs[1023];
        tile_off[L] = s_tiles[1023];
        stile_off[L] = s_stiles[1023];
        counters[CNT_TOTAL_QUADS] = s_pairs[1023] / 4;
        counters[CNT_TOTAL_TILES] = s_tiles[1023];
        counters[CNT_TOTAL_STILES] = s_stiles[1023];
    }
}

// AsymmetricHasher mode (one leaf, every query's only token): everything txh_init_kernel, ah_tokens_kernel and
// the three worklist kernels write is known from nq alone -- one single-workgroup kernel instead of five
// launches (~4.5 us of dispatch each).  Same arrays, same values.
struct AhSetupArgs {
    uint32_t nq, max_slots, st, tp, quads_per_tile, chunks_per_tile, stp, squads_per_tile;
    const uint32_t *leaf_gsize, *leaf_off;
    uint32_t *leaf_cnt, *leaf_cursor, *counters, *cand_cnt, *cand32_cnt, *pair_q, *pair_leaf, *pair_vbase, *pair_sbase,
        *slot_of, *tokens, *vbase, *sbase, *pair_off, *tile_off, *stile_off;
    float *token_dists;
};

__global__ __launch_bounds__(1024) void ah_setup_kernel(AhSetupArgs a) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t sz = a.leaf_off[1] - a.leaf_off[0], gs = a.leaf_gsize[0];
    const uint32_t c = sz ? a.nq : 0u, pad = (c + 3u) & ~3u;
    const uint32_t ssz = (sz + a.st - 1) / a.st;
    for (uint32_t i = tid; i < a.max_slots; i += nt) a.pair_q[i] = (sz && i < a.nq) ? i : kInvalid;   // slot of query i = i
    for (uint32_t i = tid; i < CNT_WORDS; i += nt) {
        uint32_t v = 0;
        if (c) {
            const uint32_t nch = (sz + a.tp - 1) / a.tp;
            if (i == CNT_TOTAL_QUADS) v = pad / 4;
            if (i == CNT_TOTAL_TILES)
                v = ((nch + a.chunks_per_tile - 1) / a.chunks_per_tile) * ((pad / 4 + a.quads_per_tile - 1) / a.quads_per_tile);
            if (i == CNT_TOTAL_STILES)
                v = ((ssz + a.stp - 1) / a.stp) * ((pad / 4 + a.squads_per_tile - 1) / a.squads_per_tile);
        }
        a.counters[i] = v;
    }
    for (uint32_t q = tid; q < a.nq; q += nt) {
        a.cand_cnt[q] = 0;
        if (a.cand32_cnt) a.cand32_cnt[q] = 0;
        a.tokens[q] = 0;
        a.token_dists[q] = 0.0f;
        a.vbase[2 * q] = 0;
        a.vbase[2 * q + 1] = gs;
        a.sbase[3 * q] = 0;
        a.sbase[3 * q + 1] = ssz;
        a.sbase[3 * q + 2] = sz;
        if (sz) {
            a.pair_leaf[q] = 0;
            a.pair_vbase[q] = 0;
            a.pair_sbase[q] = 0;
        }
        a.slot_of[q] = sz ? q : kInvalid;
    }
    if (tid == 0) {
        a.leaf_cnt[0] = c;
        a.leaf_cursor[0] = 0;
        uint32_t tiles = 0, stiles = 0;
        if (c) {
            const uint32_t nch = (sz + a.tp - 1) / a.tp;
            tiles = ((nch + a.chunks_per_tile - 1) / a.chunks_per_tile) * ((pad / 4 + a.quads_per_tile - 1) / a.quads_per_tile);
            stiles = ((ssz + a.stp - 1) / a.stp) * ((pad / 4 + a.squads_per_tile - 1) / a.squads_per_tile);
        }
        a.pair_off[0] = 0;
        a.pair_off[1] = pad;
        a.tile_off[0] = 0;
        a.tile_off[1] = tiles;
        a.stile_off[0] = 0;
        a.stile_off[1] = stiles;
    }
}

__global__ void worklist_fill_kernel(uint32_t nq, uint32_t P, int ah, const uint32_t *__restrict__ tokens,
                                     const uint32_t *__restrict__ vbase,
                                     const uint32_t *__restrict__ sbase,
                                     const uint32_t *__restrict__ leaf_off,
                                     const uint32_t *__restrict__ pair_off,
                                     uint32_t *__restrict__ leaf_cursor,
                                     uint32_t *__restrict__ pair_q, uint32_t *__restrict__ pair_leaf,
                                     uint32_t *__restrict__ pair_vbase,
                                     uint32_t *__restrict__ pair_sbase,
                                     uint32_t *__restrict__ slot_of) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * P) return;
    uint32_t q = i / P, r = i - q * P;
    uint32_t leaf = tokens[i];
    uint32_t slot = kInvalid;
    if (leaf_off[leaf + 1] > leaf_off[leaf]) {
        slot = pair_off[leaf] + (ah ? i : atomicAdd(&leaf_cursor[leaf], 1u));
        pair_q[slot] = q;
        pair_leaf[slot] = leaf;
        pair_vbase[slot] = vbase[(size_t)q * (P + 1) + r];
        pair_sbase[slot] = sbase[(size_t)q * (P + 2) + r];
    }
    slot_of[i] = slot;
}

// =====================================================================================
// K4: LUT build.  tree_x_hybrid/mod.rs:309-319 + hashes/lut.rs:47-70 +
// hashes/codebook.rs:98-115: q' = q - centroid (if residual); LUT[s][c] = sequential
// scalar sum over dsub of (q'_j - cb_j)^2.  Output layout is quad-interleaved
// [quad][s][kp][4] (kp = 16 or 256 slots) so that the scan reads four queries' entries with
// one ds_read_b128.
// =====================================================================================
__global__ __launch_bounds__(256) void lut_build_kernel(
    TxhIndexDev ix, const float *__restrict__ queries, uint32_t q_stride,
    const uint32_t *__restrict__ pair_q, const uint32_t *__restrict__ pair_leaf,
    const uint32_t *__restrict__ counters, float *__restrict__ lutq) {
    extern __shared__ float qres[];  // [4][dim]
    const uint32_t quad = blockIdx.x;
    if (quad >= counters[CNT_TOTAL_QUADS]) return;
    const uint32_t dim = ix.dim;
    for (uint32_t i = threadIdx.x; i < 4 * dim; i += blockDim.x) {
        uint32_t p = i / dim, j = i - p * dim;
        uint32_t q = pair_q[quad * 4 + p];
        float v = 0.0f;
        if (q != kInvalid) {
            v = queries[(size_t)q * q_stride + j];
            if (ix.use_residuals) v = v - ix.centers[(size_t)pair_leaf[quad * 4 + p] * dim + j];
        }
        qres[i] = v;
    }
    __syncthreads();
    const uint32_t S = ix.S, K = ix.K, dsub = ix.dsub, kp = ix.kp;   // kp = 16 or 256 table slots
    float4 *out = reinterpret_cast<float4 *>(lutq) + (size_t)quad * S * kp;
    for (uint32_t e = threadIdx.x; e < S * kp; e += blockDim.x) {
        uint32_t s = e / kp, c = e - s * kp;
        float r[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (c < K) {
            const float *cb = ix.codebook + ((size_t)s * K + c) * dsub;
            for (uint32_t j = 0; j < dsub; ++j) {
                float cv = cb[j];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float d = qres[p * dim + s * dsub + j] - cv;
                    r[p] = r[p] + d * d;
                }
            }
        }
        out[e] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// Plain LookupTable::from_query for callers/tests: out [nq][S][K].
__global__ __launch_bounds__(256) void lut_from_query_kernel(
    TxhIndexDev ix, const float *__restrict__ queries, uint32_t q_stride,
    const uint32_t *__restrict__ leaf_for_query, float *__restrict__ out) {
    extern __shared__ float qres[];  // [dim]
    const uint32_t q = blockIdx.x, dim = ix.dim;
    for (uint32_t j = threadIdx.x; j < dim; j += blockDim.x) {
        float v = queries[(size_t)q * q_stride + j];
        if (leaf_for_query) v = v - ix.centers[(size_t)leaf_for_query[q] * dim + j];
        qres[j] = v;
    }
    __syncthreads();
    const uint32_t S = ix.S, K = ix.K, dsub = ix.dsub;
    for (uint32_t e = threadIdx.x; e < S * K; e += blockDim.x) {
        uint32_t s = e / K;
        const float *cb = ix.codebook + (size_t)e * dsub;
        float r = 0.0f;
        for (uint32_t j = 0; j < dsub; ++j) {
            float d = qres[s * dsub + j] - cb[j];
            r = r + d * d;
        }
        out[(size_t)q * S * K + e] = r;
    }
}

// RestrictFilter::is_allowed (restricts/mod.rs:17-30) for the allow-bitmap form of
// search_with_filter (tree_x_hybrid/mod.rs:327-332): bit i of the bitmap = datapoint i.
// Indices at or beyond the bitmap's capacity are not allowed (allowlist.rs:97-100).
__device__ __forceinline__ bool row_allowed(const TxhIndexDev &ix, const uint64_t *allow,
                                            uint64_t allow_bits, uint32_t csr) {
    if (!allow) return true;
    const uint32_t idx = ix.leaf_ids ? ix.leaf_ids[csr] : csr;
    return idx < allow_bits && ((allow[idx >> 6] >> (idx & 63u)) & 1ull);
}

// =====================================================================================
// K6: ADC scan -- the dominant kernel.  hashes/lut.rs:74-82 driven by the loop at
// tree_x_hybrid/mod.rs:324-336 / hashes/hasher.rs:179-182.
//
// A tile = (leaf, chunk of kScanTP points, group of <= kScanQuadsPerTile query quads).
// Each lane keeps the packed codes of PPT points in VGPRs (coalesced 16-B loads), and
// for every quad reads one ds_read_b128 per (point, subspace): the 16-entry table of a
// subspace is 16 x 16 B = all 64 LDS banks, lanes with equal codes broadcast, so the
// gather is conflict-free.  Distances are accumulated in subspace order (bit-exact with
// the reference) and compared with the per-query threshold; survivors are appended to
// the query's candidate list.  Tiles are pulled from an atomic queue so ragged leaves
// balance across the 256 CUs.
// =====================================================================================
// Code layouts the scan understands.  BITS = 4: K <= 16, 8 subspaces per u32 word
// (PackedCodes4Bit, hashes/lut16.rs:43-61), 16-entry tables.  BITS = 8: 16 < K <= 256 (the
// reference's default 256 x 8 codebooks, hashes/hasher.rs:36-46), one byte per subspace, 4
// subspaces per word, 256-entry tables.  A subspace's quad-interleaved table is KP x 16 B.
template <int S_, int BITS_>
struct Codec {
    static constexpr int S = S_, BITS = BITS_;
    static constexpr int NWORDS = BITS == 4 ? S / 8 : S / 4;      // packed u32 words per point
    static constexpr int REGS = BITS == 4 ? 2 * NWORDS : NWORDS;  // registers per point in the scan
    static constexpr int KP = BITS == 4 ? 16 : 256;               // table entries per subspace
    static constexpr int SUB_BYTES = KP * 16;
    static constexpr int LUT4 = S * KP;                           // float4 per quad
    // points per thread per tile chunk: byte-code tables are 16x larger per subspace and there
    // are 4x fewer subspaces, so a tile takes 4x more points per staged table
    static constexpr int PPT = BITS == 4 ? (int)kScanPPT : 8;
    static constexpr int TP = (int)kScanThreads * PPT;            // points per tile chunk
    static_assert((S - 1) * SUB_BYTES < 65536, "ds_read immediate offset");
    // workgroups per CU the kernel is built for (LDS: two LUT buffers + survivor stage)
    static constexpr int WGS = BITS == 4 ? (S <= 32 ? (int)kScanWaves : 3)
                                         : (2 * LUT4 * 16 + 12288 <= 40 * 1024 ? 4
                                            : 2 * LUT4 * 16 + 12288 <= 80 * 1024 ? 2 : 1);
    // packed words -> register form: 4-bit codes pre-shifted to "code * 16" bytes
    __device__ static __forceinline__ void unpack(const uint32_t (&w)[NWORDS], uint32_t (&r)[REGS]) {
        if constexpr (BITS == 4) {
#pragma unroll
            for (int wi = 0; wi < NWORDS; ++wi) {
                r[2 * wi] = (w[wi] & 0x0F0F0F0Fu) << 4;
                r[2 * wi + 1] = w[wi] & 0xF0F0F0F0u;
            }
        } else {
#pragma unroll
            for (int wi = 0; wi < NWORDS; ++wi) r[wi] = w[wi];
        }
    }
    // byte offset of subspace s's code inside that subspace's table (code * 16)
    __device__ static __forceinline__ uint32_t offset(const uint32_t (&r)[REGS], int s) {
        if constexpr (BITS == 4) {
            const int wi = s >> 3, b = (s >> 1) & 3, h = s & 1;
            return (r[2 * wi + h] >> (8 * b)) & 0xFFu;
        } else {
            return ((r[s >> 2] >> (8 * (s & 3))) & 0xFFu) << 4;
        }
    }
    __device__ static __forceinline__ void load_words(const uint32_t *src, uint32_t (&w)[NWORDS]) {
        if constexpr (NWORDS == 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else if constexpr (NWORDS == 2) {
            const uint2 v = *reinterpret_cast<const uint2 *>(src);
            w[0] = v.x; w[1] = v.y;
        } else {
#pragma unroll
            for (int wi = 0; wi < NWORDS; ++wi) w[wi] = src[wi];
        }
    }
    __device__ static __forceinline__ void store_words(uint32_t *dst, const uint32_t (&w)[NWORDS]) {
        if constexpr (NWORDS == 4) {
            *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
        } else if constexpr (NWORDS == 2) {
            *reinterpret_cast<uint2 *>(dst) = make_uint2(w[0], w[1]);
        } else {
#pragma unroll
            for (int wi = 0; wi < NWORDS; ++wi) dst[wi] = w[wi];
        }
    }
};

// Points G .. G+NP-1 of the lane against the quad's tables.
template <typename C, int NP, int BUF, int G>
__device__ __forceinline__ void scan_quad_compute(const float4 *lut_base,
                                                  uint32_t (&regs)[C::PPT][C::REGS],
                                                  float (&acc)[4][C::PPT]) {
    constexpr int S = C::S;
    const char *lb = reinterpret_cast<const char *>(lut_base + BUF * C::LUT4);
    // The byte extractions below are invariant across the quad loop; without this
    // (instruction-free) barrier LICM hoists all S*NP of them into registers and spills.
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int ri = 0; ri < C::REGS; ++ri) asm volatile("" : "+v"(regs[G + i][ri]));
    // Software pipeline, kScanDepth subspaces deep: the NP ds_read_b128 of subspaces
    // s+1 .. s+D are in flight while subspace s is accumulated.  sched_barrier(0) pins the
    // stage order so the scheduler cannot hoist every gather to the top (256 VGPRs,
    // occupancy 1).
    constexpr int D = (int)kScanDepth;
    float4 v[D + 1][NP];
    auto issue = [&](int s, int slot) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            v[slot][i] = *reinterpret_cast<const float4 *>(lb + s * C::SUB_BYTES + C::offset(regs[G + i], s));
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < S) issue(d, d);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s + D < S) issue(s + D, (s + D) % (D + 1));
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float4 t = v[s % (D + 1)][i];
            if (s == 0) {
                acc[0][G + i] = t.x; acc[1][G + i] = t.y; acc[2][G + i] = t.z; acc[3][G + i] = t.w;
            } else {
                acc[0][G + i] = acc[0][G + i] + t.x;
                acc[1][G + i] = acc[1][G + i] + t.y;
                acc[2][G + i] = acc[2][G + i] + t.z;
                acc[3][G + i] = acc[3][G + i] + t.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename C, int BUF>
__device__ __forceinline__ void scan_quad_dispatch(const float4 *lut_s, uint32_t nsub,
                                                   uint32_t (&regs)[C::PPT][C::REGS],
                                                   float (&acc)[4][C::PPT]) {
    if constexpr (C::PPT <= 2) {
        switch (nsub) {
            case 1: scan_quad_compute<C, 1, BUF, 0>(lut_s, regs, acc); break;
            default: scan_quad_compute<C, (C::PPT < 2 ? 1 : 2), BUF, 0>(lut_s, regs, acc); break;
        }
    } else {   // pairs of points; a pair past nsub is skipped (wave-uniform)
        static_assert(C::PPT == 8, "point-pair groups are spelled out for 8 points per thread");
        scan_quad_compute<C, 2, BUF, 0>(lut_s, regs, acc);
        if (nsub > 2) scan_quad_compute<C, 2, BUF, 2>(lut_s, regs, acc);
        if (nsub > 4) scan_quad_compute<C, 2, BUF, 4>(lut_s, regs, acc);
        if (nsub > 6) scan_quad_compute<C, 2, BUF, 6>(lut_s, regs, acc);
    }
}

// Next tile of this workgroup: XCD x (blockIdx % 8) owns tiles t = x (mod 8) in queue x and
// steals from the other queues when its own is dry.  kInvalid = no tiles left.
__device__ __forceinline__ uint32_t grab_tile(uint32_t *queues, uint32_t total_tiles) {
    const uint32_t xcd = blockIdx.x & 7u;
    for (uint32_t a2 = 0; a2 < 8u; ++a2) {
        const uint32_t x = (xcd + a2) & 7u;
        const uint32_t nx = (total_tiles + 7u - x) >> 3;   // tiles = x (mod 8)
        uint32_t *ctr = queues + x * CNT_XQ_STRIDE;
        if (a2 && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nx) continue;
        const uint32_t l = atomicAdd(ctr, 1u);
        if (l < nx) return l * 8u + x;
    }
    return kInvalid;
}

#ifndef SCANN_SCAN_STAGE
#define SCANN_SCAN_STAGE 128
#endif
constexpr uint32_t kScanStage = SCANN_SCAN_STAGE;   // LDS-staged survivors per (quad, query); <= kScanThreads
static_assert(kScanStage <= kScanThreads, "the flush copies one survivor per thread");

struct ScanArgs {
    const uint32_t *pair_off, *tile_off, *pair_q, *pair_vbase;
    uint32_t *counters;
    const float *lutq;
    const uint64_t *pair_thr;
    uint32_t *cand_cnt;
    uint64_t *cand;
    uint32_t cap;
    uint32_t qpt;            // query quads per tile
    uint32_t res_cl;         // resident-table kernel: chunks per tile
    const uint64_t *allow;   // optional allow-bitmap (device), bit = datapoint index
    uint64_t allow_bits;
};

// LDS of the scan kernels (dynamic: two LUT buffers first, 16-byte aligned)
template <typename C>
__host__ __device__ constexpr size_t scan_lds_bytes() {
    return (size_t)2 * C::LUT4 * 16 + (size_t)2 * 4 * kScanStage * 8 + 8 * 4 + 3 * 4 * 4 + 16;
}

template <typename C>
__global__ __launch_bounds__(kScanThreads, C::WGS) void adc_scan_kernel(TxhIndexDev ix, ScanArgs a) {
    constexpr int LUT4 = C::LUT4;                                   // float4 per quad
    constexpr int STG = (LUT4 + kScanThreads - 1) / kScanThreads;   // staged float4 / thread
    extern __shared__ __attribute__((aligned(16))) float4 lut_s[];  // [2 * LUT4]
    // Survivors are staged per (quad, query) in LDS and flushed one quad later with ONE
    // returning global atomic per query, issued before the next quad's gather so its
    // latency hides under the compute (a per-lane returning atomic stalls the wave ~1 us).
    uint64_t (*ckey_s)[4][kScanStage] = reinterpret_cast<uint64_t (*)[4][kScanStage]>(lut_s + 2 * LUT4);
    uint32_t (*ccnt_s)[4] = reinterpret_cast<uint32_t (*)[4]>(ckey_s + 2);   // live counters (LDS atomics)
    uint32_t *cfrozen_s = reinterpret_cast<uint32_t *>(ccnt_s + 2);            // counts of the buffer being flushed
    uint32_t *cbase_s = cfrozen_s + 4;                                         // its global base slots
    uint32_t *cq_s = cbase_s + 4;                                              // its query ids
    uint32_t &tile_sh = cq_s[4];
    const uint32_t tid = threadIdx.x;
    const uint32_t total_tiles = a.counters[CNT_TOTAL_TILES];
    if (tid < 8) ccnt_s[tid >> 2][tid & 3u] = 0;

    for (;;) {
        if (tid == 0) tile_sh = grab_tile(a.counters + CNT_XQ, total_tiles);
        __syncthreads();
        const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_sh);
        if (tile == kInvalid) break;

        // leaf = largest l with tile_off[l] <= tile
        uint32_t lo = 0, hi = ix.L;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (uniform_load(a.tile_off + mid) <= tile) lo = mid; else hi = mid;
        }
        const uint32_t leaf = lo;
        const uint32_t lb = uniform_load(ix.leaf_off + leaf);
        const uint32_t size = uniform_load(ix.leaf_off + leaf + 1) - lb;
        const uint32_t nchunks = (size + (uint32_t)C::TP - 1) / (uint32_t)C::TP;
        const uint32_t local = tile - uniform_load(a.tile_off + leaf);
        const uint32_t chunk = local % nchunks, qg = local / nchunks;
        const uint32_t slot0 = uniform_load(a.pair_off + leaf);
        const uint32_t nquads = (uniform_load(a.pair_off + leaf + 1) - slot0) >> 2;
        const uint32_t q0 = qg * a.qpt;
        const uint32_t q1 = min(q0 + a.qpt, nquads);
        const uint32_t c0 = chunk * (uint32_t)C::TP;
        const uint32_t npts = min((uint32_t)C::TP, size - c0);
        const uint32_t nsub = (npts + kScanThreads - 1) / kScanThreads;

        // packed codes of this lane's points in the codec's register form
        uint32_t regs[C::PPT][C::REGS];
#pragma unroll
        for (int i = 0; i < C::PPT; ++i) {
            const uint32_t j = c0 + tid + kScanThreads * i;
            uint32_t w[C::NWORDS];
            C::load_words(ix.codes + (size_t)(lb + (j < size ? j : 0)) * C::NWORDS, w);
            C::unpack(w, regs[i]);
        }

        const float4 *gl = reinterpret_cast<const float4 *>(a.lutq) +
                           (size_t)((slot0 >> 2) + q0) * LUT4;
        // stage the first quad's LUT
        {
            float4 r[STG];
#pragma unroll
            for (int t = 0; t < STG; ++t) {
                uint32_t e = tid + t * kScanThreads;
                if (e < (uint32_t)LUT4) r[t] = gl[e];
            }
#pragma unroll
            for (int t = 0; t < STG; ++t) {
                uint32_t e = tid + t * kScanThreads;
                if (e < (uint32_t)LUT4) lut_s[e] = r[t];
            }
        }
        // wave-uniform filter parameters of a quad (SGPRs via s_load): query, key base, bound
        uint32_t f_pq[4], f_vb[4], f_thi[4], f_tlo[4];
        auto fetch_params = [&](uint32_t qd) {
            const uint32_t slot = slot0 + qd * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f_pq[p] = uniform_load(a.pair_q + slot + p);
                f_vb[p] = uniform_load(a.pair_vbase + slot + p);
                const uint64_t T = uniform_load(a.pair_thr + slot + p);
                f_thi[p] = (uint32_t)(T >> 32);
                f_tlo[p] = (uint32_t)T;
            }
        };
        if (q0 < q1) fetch_params(q0);
        __syncthreads();

        for (uint32_t qd = q0; qd <= q1; ++qd) {   // one extra trip flushes the last quad
            const uint32_t buf = (qd - q0) & 1u;
            const bool live = qd < q1;
            const bool more = qd + 1 < q1;
            const bool flush = qd > q0;             // buffer buf^1 holds quad qd-1's survivors
            // A. reserve global slots for the previous quad's survivors (not waited for yet)
            uint32_t gbase = 0, gq = kInvalid, gn = 0;
            if (flush && tid < 4) {
                gn = min(ccnt_s[buf ^ 1u][tid], kScanStage);
                gq = a.pair_q[slot0 + (qd - 1) * 4 + tid];
                if (gn) gbase = atomicAdd(&a.cand_cnt[gq], gn);
            }
            // A2. prefetch the next quad's LUT straight into the idle LDS buffer (LDS-DMA:
            // no VGPR staging; destination = wave-uniform base + lane * 16).
            if (more) {
                const float4 *g2 = gl + (size_t)(qd + 1 - q0) * LUT4;
#pragma unroll
                for (int t = 0; t < STG; ++t) {
                    const uint32_t e0 = (tid & ~63u) + t * kScanThreads;   // wave-uniform
                    if (e0 < (uint32_t)LUT4)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void *)(g2 + e0 + (tid & 63u)),
                            (__attribute__((address_space(3))) void *)(&lut_s[(buf ^ 1u) * LUT4 + e0]),
                            16, 0, 0);
                }
            }

            if (live) {
                // B. gather + accumulate
                float acc[4][C::PPT];
                if (buf == 0) scan_quad_dispatch<C, 0>(lut_s, nsub, regs, acc);
                else scan_quad_dispatch<C, 1>(lut_s, nsub, regs, acc);
                // threshold filter: survivors go to the LDS stage of this quad
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint32_t pq = f_pq[p];
                    if (pq == kInvalid) continue;   // wave-uniform
                    const uint64_t T = ((uint64_t)f_thi[p] << 32) | f_tlo[p];
                    const uint32_t Thi = f_thi[p];
                    const float Tf = (Thi == 0xFFFFFFFFu) ? __builtin_inff() : ordered_to_f32(Thi);
                    const uint32_t vb = f_vb[p];
#pragma unroll
                    for (int i = 0; i < C::PPT; ++i) {
                        if (i < (int)nsub && acc[p][i] <= Tf) {
                            const uint32_t j = c0 + tid + kScanThreads * i;
                            if (j < size) {
                                const uint64_t key = make_key(acc[p][i], vb + j);
                                if (key <= T && row_allowed(ix, a.allow, a.allow_bits, lb + j)) {
                                    const uint32_t sl = atomicAdd(&ccnt_s[buf][p], 1u);
                                    if (sl < kScanStage) {
                                        ckey_s[buf][p][sl] = key;
                                    } else {   // stage full: direct (slow) append
                                        const uint32_t pos = atomicAdd(&a.cand_cnt[pq], 1u);
                                        if (pos < a.cap) a.cand[(size_t)pq * a.cap + pos] = key;
                                    }
                                }
                            }
                        }
                    }
                }
            }
            // A3. the next quad's filter parameters (in flight across the barrier below)
            if (more) fetch_params(qd + 1);

            // C. publish the flush parameters, reset the flushed buffer's live counters
            if (tid < 4) {
                cfrozen_s[tid] = gn;
                cbase_s[tid] = gbase;
                cq_s[tid] = gq;
                if (flush) ccnt_s[buf ^ 1u][tid] = 0;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA prefetch has landed
            __syncthreads();
            // D. copy the previous quad's survivors to the per-query candidate lists
            if (flush) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint32_t n_p = cfrozen_s[p];
                    if (tid < n_p) {
                        const uint32_t pos = cbase_s[p] + tid;
                        if (pos < a.cap) a.cand[(size_t)cq_s[p] * a.cap + pos] = ckey_s[buf ^ 1u][p][tid];
                    }
                }
            }
            __syncthreads();
        }
    }
}

// =====================================================================================
// K6b: ADC scan with RESIDENT tables, for long leaves (AsymmetricHasher mode, big partitions).
//
// adc_scan_kernel above re-stages a quad's table for every 512-point chunk and synchronises
// the workgroup twice per quad.  Here a tile = (leaf, kResQuads quads, a RANGE of chunks): the
// quads' tables are loaded into LDS once per tile and stay read-only while the workgroup (8
// waves) walks the chunk range, so the steady state has NO workgroup barrier and no table
// traffic: waves drift freely and the LDS gather and the VALU adds of different waves overlap.
// Survivors are staged per WAVE (kResStage slots per (wave, query), double-buffered by chunk):
// the wave reserves global slots for chunk c-1's survivors with one returning atomic per query
// issued BEFORE chunk c's gather and writes them out after it, so the atomic's latency hides
// under the compute.  Arithmetic, keys and the filter are those of adc_scan_kernel.
// =====================================================================================
constexpr uint32_t kResStage = 8;      // staged survivors per (wave, query, chunk)

template <typename C>
__host__ __device__ constexpr size_t res_lds_bytes() {
    return (size_t)kResQuads * C::LUT4 * 16 + (size_t)(kResThreads / 64) * 2 * kResQuads * 4 * kResStage * 8 +
           (size_t)(kResThreads / 64) * 2 * kResQuads * 4 * 4 + kResQuads * 4 * 4 + 16;
}

template <typename C>
__global__ __launch_bounds__(kResThreads, 6) void adc_scan_res_kernel(TxhIndexDev ix, ScanArgs a) {
    static_assert(C::PPT == 2, "resident-table scan: 2 points per thread");
    constexpr int LUT4 = C::LUT4;
    constexpr int NQS = kResQuads * 4;                                   // resident (query, leaf) pairs
    constexpr int NWV = kResThreads / 64;
    constexpr uint32_t TPR = kResThreads * C::PPT;                       // points per chunk
    extern __shared__ __attribute__((aligned(16))) float4 lut_s[];       // [kResQuads * LUT4]
    uint64_t (*skey)[2][NQS][kResStage] =
        reinterpret_cast<uint64_t (*)[2][NQS][kResStage]>(lut_s + kResQuads * LUT4);   // [NWV]
    uint32_t (*scnt)[2][NQS] = reinterpret_cast<uint32_t (*)[2][NQS]>(skey + NWV);      // [NWV]
    uint32_t *sq = reinterpret_cast<uint32_t *>(scnt + NWV);                             // [NQS] query ids
    uint32_t &tile_sh = sq[NQS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t total_tiles = a.counters[CNT_TOTAL_TILES];

    for (;;) {
        __syncthreads();                       // the previous tile's table reads are done
        if (tid == 0) tile_sh = grab_tile(a.counters + CNT_XQ, total_tiles);
        __syncthreads();
        const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_sh);
        if (tile == kInvalid) break;

        uint32_t lo = 0, hi = ix.L;            // leaf = largest l with tile_off[l] <= tile
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (uniform_load(a.tile_off + mid) <= tile) lo = mid; else hi = mid;
        }
        const uint32_t leaf = lo;
        const uint32_t lb = uniform_load(ix.leaf_off + leaf);
        const uint32_t size = uniform_load(ix.leaf_off + leaf + 1) - lb;
        const uint32_t nchunks = (size + TPR - 1) / TPR;
        const uint32_t nranges = (nchunks + a.res_cl - 1) / a.res_cl;
        const uint32_t local = tile - uniform_load(a.tile_off + leaf);
        const uint32_t range = local % nranges, qg = local / nranges;
        const uint32_t slot0 = uniform_load(a.pair_off + leaf);
        const uint32_t nquads = (uniform_load(a.pair_off + leaf + 1) - slot0) >> 2;
        const uint32_t q0 = qg * kResQuads;
        const uint32_t nq_t = min(kResQuads, nquads - q0);       // resident quads of this tile
        const uint32_t c_begin = range * a.res_cl, c_end = min(nchunks, c_begin + a.res_cl);

        // tables of the tile's quads -> LDS (LDS-DMA, destination = wave-uniform base + lane * 16)
        {
            const float4 *gl = reinterpret_cast<const float4 *>(a.lutq) + (size_t)((slot0 >> 2) + q0) * LUT4;
            const uint32_t n4 = nq_t * LUT4;
            for (uint32_t e0 = (tid & ~63u); e0 < n4; e0 += kResThreads)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(gl + e0 + lane),
                    (__attribute__((address_space(3))) void *)(&lut_s[e0]), 16, 0, 0);
        }
        if (tid < (uint32_t)NQS) sq[tid] = tid < nq_t * 4 ? a.pair_q[slot0 + q0 * 4 + tid] : kInvalid;
        if (lane < (uint32_t)NQS) {
            scnt[wave][0][lane] = 0;
            scnt[wave][1][lane] = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // this lane's slot of the wave-level flush: pair j = lane (lanes 0 .. NQS-1)
        const uint32_t my_q = lane < (uint32_t)NQS ? sq[lane] : kInvalid;
        uint32_t raw[C::PPT][C::NWORDS];
        auto fetch_codes = [&](uint32_t c) {
#pragma unroll
            for (int i = 0; i < C::PPT; ++i) {
                const uint32_t j = c * TPR + tid + kResThreads * i;
                C::load_words(ix.codes + (size_t)(lb + (j < size ? j : 0)) * C::NWORDS, raw[i]);
            }
        };
        fetch_codes(c_begin);
        for (uint32_t c = c_begin; c <= c_end; ++c) {      // one extra trip flushes the last chunk
            const uint32_t buf = (c - c_begin) & 1u;
            const bool live = c < c_end;
            const bool flush = c > c_begin;                // stage buf^1 holds chunk c-1's survivors
            uint32_t regs[C::PPT][C::REGS];
            if (live) {
#pragma unroll
                for (int i = 0; i < C::PPT; ++i) C::unpack(raw[i], regs[i]);
                if (c + 1 < c_end) fetch_codes(c + 1);     // in flight during this chunk's gathers
            }
            // A. reserve global slots for the previous chunk's survivors (result used in C)
            uint32_t fcnt = 0, fbase = 0;
            if (flush && my_q != kInvalid) {
                fcnt = min(scnt[wave][buf ^ 1u][lane], kResStage);
                if (fcnt) fbase = atomicAdd(&a.cand_cnt[my_q], fcnt);
            }
            // B. gather + accumulate + filter, quad after quad against the resident tables
            if (live) {
                const uint32_t c0 = c * TPR;
                const uint32_t npts = min(TPR, size - c0);
                const uint32_t nsub = (npts + kResThreads - 1) / kResThreads;
                for (uint32_t qd = 0; qd < nq_t; ++qd) {
                    const uint32_t slot = slot0 + (q0 + qd) * 4;
                    uint32_t f_pq[4], f_vb[4], f_thi[4], f_tlo[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        f_pq[p] = uniform_load(a.pair_q + slot + p);
                        f_vb[p] = uniform_load(a.pair_vbase + slot + p);
                        const uint64_t T = uniform_load(a.pair_thr + slot + p);
                        f_thi[p] = (uint32_t)(T >> 32);
                        f_tlo[p] = (uint32_t)T;
                    }
                    float acc[4][C::PPT];
                    const float4 *lq = lut_s + qd * LUT4;
                    if (nsub == 1) scan_quad_compute<C, 1, 0, 0>(lq, regs, acc);
                    else scan_quad_compute<C, 2, 0, 0>(lq, regs, acc);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const uint32_t pq = f_pq[p];
                        if (pq == kInvalid) continue;   // wave-uniform
                        const uint64_t T = ((uint64_t)f_thi[p] << 32) | f_tlo[p];
                        const uint32_t Thi = f_thi[p];
                        const float Tf = (Thi == 0xFFFFFFFFu) ? __builtin_inff() : ordered_to_f32(Thi);
                        const uint32_t vb = f_vb[p];
#pragma unroll
                        for (int i = 0; i < C::PPT; ++i) {
                            if (i < (int)nsub && acc[p][i] <= Tf) {
                                const uint32_t j = c0 + tid + kResThreads * i;
                                if (j < size) {
                                    const uint64_t key = make_key(acc[p][i], vb + j);
                                    if (key <= T && row_allowed(ix, a.allow, a.allow_bits, lb + j)) {
                                        const uint32_t sl = atomicAdd(&scnt[wave][buf][qd * 4 + p], 1u);
                                        if (sl < kResStage) {
                                            skey[wave][buf][qd * 4 + p][sl] = key;
                                        } else {   // stage full: direct (slow) append
                                            const uint32_t pos = atomicAdd(&a.cand_cnt[pq], 1u);
                                            if (pos < a.cap) a.cand[(size_t)pq * a.cap + pos] = key;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            // C. write the previous chunk's survivors: lane = (pair j, entry e)
            if (flush) {
                constexpr int PER = 64 / NQS;                          // entries per pair per pass
#pragma unroll
                for (int e0 = 0; e0 < (int)kResStage; e0 += PER) {
                    const uint32_t j = lane / PER, e = e0 + lane % PER;
                    const uint32_t cj = (uint32_t)__shfl((int)fcnt, (int)j);
                    const uint32_t bj = (uint32_t)__shfl((int)fbase, (int)j);
                    const uint32_t qj = (uint32_t)__shfl((int)my_q, (int)j);
                    if (e < cj) {
                        const uint32_t pos = bj + e;
                        if (pos < a.cap) a.cand[(size_t)qj * a.cap + pos] = skey[wave][buf ^ 1u][j][e];
                    }
                }
                if (lane < (uint32_t)NQS) scnt[wave][buf ^ 1u][lane] = 0;
            }
        }
    }
}

// =====================================================================================
// K6d: ADC scan as an integer-MFMA prefilter + exact refine (4-bit codes, threshold known).
//
// The f32 scan above is bound by the LDS table gather (one ds_read_b128 per point, subspace and
// quad of queries).  The same sums over QUANTISED tables are a matrix product:
//     one-hot(codes) [points x (S*16)]  x  lut8 [(S*16) x pairs]   (u8 tables as i8 minus 128)
// which v_mfma_i32_32x32x32_i8 computes exactly (integer) at 1024 MAC/clk/SIMD: a 32-point x
// 32-pair tile costs S/2 MFMAs.  The integer sum BOUNDS the reference's f32 sum: with per-subspace
// offsets mn_s and one scale sc per pair, every table entry v satisfies |v - (mn_s + sc*q)| <=
// sc*(0.5 + 1e-9), so a point whose f32 sum passes the filter bound T has
//     sum_q <= (T*(1 + S*2^-23) - sum mn_s)/sc + S/2 + 1
// (the factor covers the rounding of the sequential f32 adds of non-negative terms).  Points under
// that integer bound -- the true survivors plus ~10 % -- are listed per query as stream positions,
// and adc_refine_kernel recomputes THEIR distances with the reference's arithmetic (f32 tables,
// subspace order: hashes/lut.rs:74-82), forms the merge keys and applies the exact filter.  The
// candidate lists handed to select_rerank_kernel are therefore identical to adc_scan_kernel's:
// the same shortlist-plus-proof pattern as the bf16 brute-force pass (bf.hip).
//
// Work decomposition: every WAVE pulls its own items (leaf, tile of 32 pair slots, range of
// kMfmaRange points) from the tile queues; the pair tile's tables are the wave's B fragments for
// the whole item (S/2 x 4 VGPRs), the A fragment of a (point, subspace pair) is one row of a
// 16 x 16-byte identity table in LDS (one conflict-free ds_read_b128 at offset code * 16), the
// 16 results of a lane belong to ONE pair (column) and are compared with that pair's bound.
// Survivors are staged per (wave, pair) in LDS and written at the end of the item as one
// contiguous segment per pair behind ONE returning atomic per pair.
// =====================================================================================
constexpr uint32_t kDecodeStage = 512;    // selected leaves whose decode tables are staged in LDS
#ifndef SCANN_MFMA_RANGE
#define SCANN_MFMA_RANGE 2048
#endif
constexpr uint32_t kMfmaRange = SCANN_MFMA_RANGE;     // points per item
constexpr uint32_t kMfmaStage = 56;       // staged survivors per (wave, pair)
constexpr uint32_t kMfmaWaves = 4;        // waves per workgroup
#ifndef SCANN_MFMA_MINW
#define SCANN_MFMA_MINW 3
#endif
#ifndef SCANN_MFMA_DEPTH
#define SCANN_MFMA_DEPTH 3
#endif
constexpr int kMfmaDepth = SCANN_MFMA_DEPTH;   // one-hot LDS reads in flight per wave
constexpr uint32_t kRefineTablesMax = 40; // pair tables (2 KB each at S = 32) staged in LDS by the refine

struct Lut8Meta {
    double bias_sum;   // sum over subspaces of the per-subspace minimum
    double scale;      // table step; 0 = this pair is not prefiltered (every point passes)
};

// lutq [quad][s][16][4] f32 -> lut8 [slot][s][16] i8 (quantised value - 128) + meta[slot]
// Pass bound of a pair slot on the integer sums (see the derivation above), as thr + 1: a point passes iff
// acc - thr1 < 0.  Sums lie in [-128 S, 127 S]; the bound is clamped just outside that range (everything
// passes: no filter bound, or a table that is not quantised; nothing passes: padding slots).
__device__ __forceinline__ int mfma_pass_bound(uint32_t S, uint32_t pq, uint64_t T, double bias_sum, double scale) {
    const int lim = 128 * (int)S + 8;
    int thr = -lim;
    if (pq != kInvalid) {
        thr = lim;
        if (T != SCANN_KEY_MAX && scale > 0.0) {
            const double Tf = (double)ordered_to_f32((uint32_t)(T >> 32));
            const double qmax = floor((Tf * (1.0 + (double)S * 1.1920928955078125e-07) - bias_sum) / scale +
                                      0.5 * (double)S + 1.0) - 128.0 * (double)S;
            thr = qmax >= (double)lim ? lim : (qmax <= -(double)lim ? -lim : (int)qmax);
        }
    }
    return thr + 1;
}

// fold != 0 (adc_smfmac_kernel): the pass bound is folded INTO the tables, so that a point passes iff its integer sum
// is negative (the sparse MFMA accumulates in place: there is no free zero / bound operand, and the sign test is one
// vector instruction per result instead of two).  With qmax = the largest quantised sum a passing point can have (as
// in mfma_pass_bound), D = 128 S - 1 - qmax >= 0 is spread over the subspaces, d_s = D / S (+ 1 for the first D % S),
// and the entries are e = min(127, q - 128 + d_s): sum(q - 128 + d_s) = sum q - qmax - 1 < 0 <=> sum q <= qmax; the
// clamp at 127 only lowers sums (more points pass, never fewer) and d_s >= 0 means no entry is clamped from below.
// A bound with qmax > 128 S - 1 (more than half of the table range: a very loose filter) gets a coarser scale first,
// sc' = (T' - bias) / (127.5 S - 3): every entry still satisfies |v - (mn_s + sc' q)| <= sc' (0.5 + 1e-9), q <= 255.
// All-pass pairs (no bound, unquantisable table) store -128 everywhere, padding slots of a quad and bounds no point
// can meet store 127 (sums >= 0).
__global__ __launch_bounds__(256) void lut8_build_kernel(uint32_t S, const float *__restrict__ lutq,
                                                        const uint32_t *__restrict__ counters,
                                                        int8_t *__restrict__ lut8, Lut8Meta *__restrict__ meta,
                                                        const uint32_t *__restrict__ pair_q,
                                                        const uint64_t *__restrict__ pair_thr, int *__restrict__ thr1,
                                                        int fold) {
    __shared__ float s_min[4][64], s_rng[4][64];
    __shared__ double s_scale[4];
    __shared__ int s_bad[4], s_mode[4], s_dbase[4], s_drem[4];   // fold: 0 = quantise, 1 = all pass, 2 = none pass
    const uint32_t quad = blockIdx.x, tid = threadIdx.x;
    if (quad >= counters[CNT_TOTAL_QUADS]) return;
    const float4 *src = reinterpret_cast<const float4 *>(lutq) + (size_t)quad * S * 16;
    if (tid < 4) s_bad[tid] = 0;
    __syncthreads();
    const uint32_t p = tid & 3u, sub = tid >> 2;    // thread = (pair of the quad, subspace)
    float v[16];
    if (sub < S) {
        float mn = __builtin_inff(), mx = -__builtin_inff();
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 e = src[sub * 16 + c];
            const float x = p == 0 ? e.x : p == 1 ? e.y : p == 2 ? e.z : e.w;
            v[c] = x;
            bad = bad || !(x >= 0.0f) || !(x < __builtin_inff());   // NaN, negative, infinite
            mn = fminf(mn, x);
            mx = fmaxf(mx, x);
        }
        s_min[p][sub] = mn;
        s_rng[p][sub] = mx - mn;
        if (bad) atomicOr(&s_bad[p], 1);
    }
    __syncthreads();
    if (tid < 4) {
        double bias = 0.0;
        float r = 0.0f;
        for (uint32_t j = 0; j < S; ++j) {
            bias += (double)s_min[tid][j];
            r = fmaxf(r, s_rng[tid][j]);
        }
        double sc = (s_bad[tid] || !(r > 0.0f)) ? 0.0 : (double)r / 255.0;
        const size_t slot = (size_t)quad * 4 + tid;
        if (fold) {
            int mode = 1, dbase = 0, drem = 0;
            const uint64_t T = pair_thr[slot];
            if (pair_q[slot] == kInvalid) {
                mode = 2;
            } else if (T != SCANN_KEY_MAX && sc > 0.0) {
                const double Tf = (double)ordered_to_f32((uint32_t)(T >> 32));
                const double tq = Tf * (1.0 + (double)S * 1.1920928955078125e-07) - bias;
                const double lim = 128.0 * (double)S - 1.0;
                if (tq < 3.0e38) {   // (false for a NaN bound: everything passes)
                    double qmax = floor(tq / sc + 0.5 * (double)S + 1.0);
                    if (qmax > lim) {
                        const double sc2 = tq / (lim - 0.5 * (double)S - 2.0) * (1.0 + 1e-12);
                        sc = sc2 > sc ? sc2 : sc;
                        qmax = floor(tq / sc + 0.5 * (double)S + 1.0);
                    }
                    if (qmax < 0.0) {
                        mode = 2;
                    } else if (qmax <= lim) {
                        const int delta = (int)(lim - qmax);
                        mode = 0;
                        dbase = delta / (int)S;
                        drem = delta % (int)S;
                    }
                }
            }
            s_mode[tid] = mode;
            s_dbase[tid] = dbase;
            s_drem[tid] = drem;
            thr1[slot] = 0;
        } else {
            // (the pair's pass bound right away: the filter bounds are known by now -- one launch less)
            thr1[slot] = mfma_pass_bound(S, pair_q[slot], pair_thr[slot], bias, sc);
        }
        s_scale[tid] = sc;
        Lut8Meta m;
        m.bias_sum = bias;
        m.scale = sc;
        meta[slot] = m;
    }
    __syncthreads();
    if (sub < S) {
        const double sc = s_scale[p], mn = (double)s_min[p][sub];
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            int q = 0;
            if (sc > 0.0) {
                const double t = ((double)v[c] - mn) / sc;
                q = (int)floor(t + 0.5);
                q = q < 0 ? 0 : (q > 255 ? 255 : q);
            }
            int e = q - 128;
            if (fold) {
                const int mode = s_mode[p];
                e += s_dbase[p] + ((int)sub < s_drem[p] ? 1 : 0);
                e = mode == 1 ? -128 : mode == 2 ? 127 : (e > 127 ? 127 : e);
            }
            w[c >> 2] |= (uint32_t)(e & 0xFF) << (8 * (c & 3));
        }
        *reinterpret_cast<uint4 *>(lut8 + (((size_t)quad * 4 + p) * S + sub) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

struct MfmaArgs {
    const int *thr1;          // [slots] pass bound + 1 (lut8_build_kernel)
    const uint32_t *pair_off, *tile_off, *pair_q, *pair_vbase;
    uint32_t *counters;
    const int8_t *lut8;
    const Lut8Meta *meta;
    const uint64_t *pair_thr;
    uint32_t *cand32_cnt;     // [nq]
    uint32_t *cand32;         // [nq][cap32] stream positions of the prefilter's survivors
    uint32_t *cand32_codes;   // [nq][cap32][S/8] their packed codes: the refine reads them in list order
    uint32_t cap32;
};

template <int S_>
__global__ __launch_bounds__(kMfmaWaves * 64, (S_ <= 32 ? SCANN_MFMA_MINW : 2)) void adc_mfma_kernel(TxhIndexDev ix, MfmaArgs a) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v16i __attribute__((ext_vector_type(16)));
    constexpr int S = S_, KS = S / 2, NW = S / 8;
    __shared__ __attribute__((aligned(16))) uint32_t s_ident[64];                 // 16 one-hot rows of 16 bytes
    __shared__ uint32_t s_stage[kMfmaWaves][32][kMfmaStage];
    __shared__ uint32_t s_cnt[kMfmaWaves][32];
    __shared__ uint32_t s_fpre[kMfmaWaves][32], s_fq[kMfmaWaves][32], s_fgb[kMfmaWaves][32], s_fvb[kMfmaWaves][32];   // flush: per pair
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, h = lane >> 5;
    // row c (16 bytes = words 4c .. 4c+3): byte c set to 1  ->  word 4c + (c >> 2) holds 1 << 8*(c & 3)
    if (tid < 64) {
        const uint32_t c = tid >> 2, wsel = tid & 3u;
        s_ident[tid] = (wsel == (c >> 2)) ? (1u << (8 * (c & 3u))) : 0u;
    }
    __syncthreads();
    const uint32_t total_tiles = a.counters[CNT_TOTAL_TILES];
    const char *ident = reinterpret_cast<const char *>(s_ident);

    uint32_t tile = 0;
    if (lane == 0) tile = grab_tile(a.counters + CNT_XQ, total_tiles);
    tile = __builtin_amdgcn_readfirstlane(tile);
    while (tile != kInvalid) {
        // the next item's queue atomic travels while this item is computed
        uint32_t next_tile = 0;
        if (lane == 0) next_tile = grab_tile(a.counters + CNT_XQ, total_tiles);
        uint32_t lo = 0, hi = ix.L;            // leaf = largest l with tile_off[l] <= tile
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (uniform_load(a.tile_off + mid) <= tile) lo = mid; else hi = mid;
        }
        const uint32_t leaf = lo;
        const uint32_t lb = uniform_load(ix.leaf_off + leaf);
        const uint32_t size = uniform_load(ix.leaf_off + leaf + 1) - lb;
        const uint32_t nranges = (size + kMfmaRange - 1) / kMfmaRange;
        const uint32_t local = tile - uniform_load(a.tile_off + leaf);
        const uint32_t range = local % nranges, pt = local / nranges;
        const uint32_t slot0 = uniform_load(a.pair_off + leaf);
        const uint32_t slot_end = uniform_load(a.pair_off + leaf + 1);
        const uint32_t c0 = range * kMfmaRange;
        const uint32_t npts = min(kMfmaRange, size - c0);

        // this lane's pair (column): tables, bound, key base
        const uint32_t slot = slot0 + pt * 32u + col;
        const bool pair_ok = slot < slot_end;
        const uint32_t pq = pair_ok ? a.pair_q[slot] : kInvalid;
        const uint32_t vb = pair_ok ? a.pair_vbase[slot] : 0u;
        v4i b[KS];
        {
            const int8_t *bsrc = a.lut8 + ((size_t)(pair_ok ? slot : slot0) * S + h) * 16;   // padding columns: any table
#pragma unroll
            for (int t = 0; t < KS; ++t) b[t] = *reinterpret_cast<const v4i *>(bsrc + (size_t)t * 32);
        }
        const int thr1 = pair_ok ? a.thr1[slot] : -(128 * S + 7);   // a point passes iff acc - thr1 < 0
        if (lane < 32) s_cnt[wave][lane] = 0;
        // (s_cnt / s_stage are private to the wave: no workgroup barrier anywhere in this loop)

        const uint32_t ntile = (npts + 31u) >> 5;
        uint32_t wn[NW];
        {
            const uint32_t j = c0 + col;
            Codec<S, 4>::load_words(ix.codes + (size_t)(lb + (j < size ? j : 0u)) * NW, wn);
        }
        // Software pipeline over the item's tiles, two accumulators: step(t) issues the MFMA chain of tile
        // t and, between its MFMAs (their shadow hides ~6 vector instructions each), builds the survivor
        // mask of tile t - 1 from the other accumulator; then the (rare, branchy) survivor staging of
        // tile t - 1.  One extra step drains the last tile (its MFMAs run on stale codes and are dropped).
        // Staged survivors -> the queries' lists: ONE returning atomic per flushed pair (all of them in one
        // wave instruction), then one contiguous segment per pair.  all = false flushes only the pairs
        // whose stage could overflow in the next tile (a tile adds at most 32 per pair): dense pairs --
        // the nearest leaves of a query, where a large share of the points pass -- flush often, sparse
        // ones once per item.
        auto flush = [&](bool all) {
            uint32_t n = 0, gbase = 0;
            if (lane < 32) {
                n = min(s_cnt[wave][lane], kMfmaStage);
                if (!all && n + 32u <= kMfmaStage) n = 0;
                if (n) {
                    gbase = atomicAdd(&a.cand32_cnt[pq], n);
                    s_cnt[wave][lane] = 0;
                }
            }
            // all flushed pairs as ONE list spread over the 64 lanes: entry e belongs to the pair c with
            // pre[c] <= e < pre[c] + n[c]; its position and its packed codes (the tile's lines are still in
            // L2) go to slot gbase[c] + (e - pre[c]) of the query's list
            uint32_t incl = n;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                if ((int)lane >= o) incl += up;
            }
            const uint32_t total = (uint32_t)__shfl((int)incl, 31);
            if (total == 0) return;
            if (lane < 32) {
                s_fpre[wave][lane] = incl - n;
                s_fq[wave][lane] = pq == kInvalid ? 0u : pq;
                s_fgb[wave][lane] = gbase;
                s_fvb[wave][lane] = vb;
            }
            // (per-pair values through LDS, not shuffles: the loop's last pass runs with lanes switched off)
            for (uint32_t e = lane; e < total; e += 64u) {
                uint32_t c = 0;
#pragma unroll
                for (uint32_t stp = 16; stp; stp >>= 1)
                    if (s_fpre[wave][c + stp] <= e) c += stp;
                const uint32_t idx = e - s_fpre[wave][c];
                const uint32_t j = s_stage[wave][c][idx];
                const uint32_t dst = s_fgb[wave][c] + idx;
                if (dst < a.cap32) {
                    const size_t o = (size_t)s_fq[wave][c] * a.cap32 + dst;
                    a.cand32[o] = s_fvb[wave][c] + j;
                    if (a.cand32_codes) {   // (wave-uniform)
                        uint32_t cw[NW];
                        Codec<S, 4>::load_words(ix.codes + (size_t)(lb + j) * NW, cw);
                        Codec<S, 4>::store_words(a.cand32_codes + o * NW, cw);
                    }
                }
            }
        };
        auto step = [&](v16i &accN, const v16i &accO, uint32_t t) {
            // nibbles of this lane's subspace parity h, pre-shifted to byte offsets code * 16
            uint32_t rg[NW];
#pragma unroll
            for (int wi = 0; wi < NW; ++wi) rg[wi] = h ? (wn[wi] & 0xF0F0F0F0u) : ((wn[wi] & 0x0F0F0F0Fu) << 4);
            if (t + 1 < ntile) {
                const uint32_t j = c0 + (t + 1) * 32u + col;
                Codec<S, 4>::load_words(ix.codes + (size_t)(lb + (j < size ? j : 0u)) * NW, wn);
            }
            accN = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            // A fragments: kMfmaDepth one-hot rows in flight ahead of the MFMA that consumes them
            constexpr int D = kMfmaDepth < KS ? kMfmaDepth : KS;
            v4i av[D + 1];
            auto onehot = [&](int kt) {
                const uint32_t off = (rg[kt >> 2] >> (8 * (kt & 3))) & 0xFFu;   // code * 16 of subspace 2 kt + h
                return *reinterpret_cast<const v4i *>(ident + off);
            };
#pragma unroll
            for (int kt = 0; kt < D; ++kt) av[kt] = onehot(kt);
            // lane (col, h), register r: point row (r & 3) + 8 * (r >> 2) + 4 * h of the tile.  Survivor
            // bits of the lane's 16 results without a branch per result (a tile holds ~10 survivors among
            // 1024 results): v_sub + v_alignbit shift the sign of acc - thr1 into the mask, so result r
            // ends up at bit 15 - r.
            uint32_t m16 = 0;
            // the MFMA chain issues ahead of the other waves' staging / flush streams (s_setprio: -2 % kernel time)
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int kt = 0; kt < KS; ++kt) {
                if (kt + D < KS) av[(kt + D) % (D + 1)] = onehot(kt + D);
                accN = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[kt % (D + 1)], b[kt], accN, 0, 0, 0);
#pragma unroll
                for (int r = kt * 16 / KS; r < (kt + 1) * 16 / KS; ++r)
                    m16 = __builtin_amdgcn_alignbit(m16, (uint32_t)(accO[r] - thr1), 31);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            if (t == 0) return;                // nothing before the first tile (wave-uniform)
            const uint32_t base = c0 + (t - 1) * 32u + 4u * h;
            if (t == ntile && (npts & 31u)) {  // partial last tile: rows past the leaf's end are padding
                uint32_t okm = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    okm |= (base + (uint32_t)((r & 3) + 8 * (r >> 2)) < size ? 1u : 0u) << (15 - r);
                m16 &= okm;
            }
            bool risk = false;                 // this lane's pair could overflow its stage in the next tile
            if (m16) {
                uint32_t sl = atomicAdd(&s_cnt[wave][col], (uint32_t)__popc(m16));   // one LDS atomic per lane
                do {
                    const uint32_t r = 15u - ((uint32_t)__ffs((int)m16) - 1u);
                    m16 &= m16 - 1u;
                    const uint32_t j = base + (r & 3u) + ((r >> 2) << 3);
                    if (sl < kMfmaStage) {
                        s_stage[wave][col][sl] = j;
                    } else {   // stage full: direct (slow) append
                        const uint32_t pos = atomicAdd(&a.cand32_cnt[pq], 1u);
                        if (pos < a.cap32) {
                            const size_t o = (size_t)pq * a.cap32 + pos;
                            a.cand32[o] = vb + j;
                            if (a.cand32_codes) {
                                uint32_t cw[NW];
                                Codec<S, 4>::load_words(ix.codes + (size_t)(lb + j) * NW, cw);
                                Codec<S, 4>::store_words(a.cand32_codes + o * NW, cw);
                            }
                        }
                    }
                    ++sl;
                } while (m16);
                risk = sl + 32u > kMfmaStage;   // (the lane that appended last to a pair saw its full count)
            }
            if (__any(risk)) flush(false);
        };
        // (tile 0 is peeled: inside the loop t >= 1 is known, so the compiler keeps the mask build between
        // the MFMAs in BOTH instances instead of sinking it below a `t == 0` branch)
        v16i accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA;
        step(accA, accB, 0u);
        for (uint32_t tl = 1; tl <= ntile; tl += 2) {
            step(accB, accA, tl);
            if (tl + 1 <= ntile) step(accA, accB, tl + 1);
        }
        flush(true);
        tile = __builtin_amdgcn_readfirstlane(next_tile);
    }
}

// =====================================================================================
// K6e: the 32-pair prefilter on the 2:4 STRUCTURED-SPARSE MFMA (v_smfmac_i32_32x32x64_i8).
//
// A one-hot row has one non-zero per 16 K-elements, so it satisfies 2:4 sparsity by construction: the
// sparse instruction multiplies a COMPRESSED A (two values per group of four K-elements + a 2-bit position
// each) with a dense 64-deep B in the time the dense instruction takes for K = 32 -- four subspaces per MFMA
// slot instead of two (tools/micro/smfmac_probe.hip: 21.5-26 ns against 19.4-25 ns per instruction per SIMD).
// Operand layout (probed on the hardware, same tool): A lane (m, ha) holds row m; its compressed byte b (value slot
// b & 1 of group (b >> 1) & 3 of half b >> 3) with position i multiplies B lane (n, hb = b >> 3), byte 16 ha + 4
// ((b >> 1) & 3) + i; the selection is a plain mux (equal or descending positions of a group's two values work).
// Sparse MFMA kt therefore covers subspaces s0 .. s0 + 3 (s0 = 4 + 4 kt): B lane (n, hb) = the 32 table bytes of
// subspaces s0 + 2 hb, s0 + 2 hb + 1 of pair n (contiguous in lut8), A lane (m, ha) = the codes ca = code[s0 + ha]
// (bytes 0..7) and cb = code[s0 + 2 + ha] (bytes 8..15) of point m: value 1 at byte 2 (ca >> 2) / 8 + 2 (cb >> 2),
// positions (ca & 3) / (cb & 3) replicated over the half's four groups (the other groups hold zeros).
//
// The sparse instruction accumulates in place (no C operand), so a tile starts with two DENSE MFMAs (subspaces 0..3,
// C = the inline constant 0: no accumulator clearing on the vector pipe) followed by (S - 4) / 4 sparse ones: 9 MFMA
// slots per 32 x 32 tile at S = 32 instead of 16.  Both A operands come from 16-row LDS tables (conflict-free
// ds_read_b128 / ds_read_b32: lanes with equal rows broadcast); their row numbers are precomputed per point at index
// creation as two nibble PLANES (codes_sp: V = (ca >> 2) | (cb >> 2) << 2 picks the value row, N = (ca & 3) |
// (cb & 3) << 2 the position word; the last nibble of each plane is the raw code of dense MFMA 0 / 1), so a tile
// costs 7 unpack instructions + 2 byte extractions per sparse MFMA.  The pass bound lives in the tables
// (lut8_build_kernel, fold): a point passes iff its sum is negative -- one v_alignbit per result.
// Items, survivor staging, flush and lists as in adc_mfma_kernel; candidate lists identical (the refine is exact).
// =====================================================================================
#ifndef SCANN_SP_STAGE
#define SCANN_SP_STAGE 512
#endif
constexpr uint32_t kSpStage = SCANN_SP_STAGE;   // adc_smfmac_kernel: list entries staged per flush round (per wave)
#ifndef SCANN_SP_U
#define SCANN_SP_U 8
#endif
constexpr uint32_t kSpU = SCANN_SP_U;         // ... and code rows in flight per lane in the copy phase

template <int S_>
struct SpLayout {
    static constexpr int NS = (S_ - 4) / 4;      // sparse MFMAs per tile
    static constexpr int NIB = NS + 1;           // nibbles per plane (the last one: a dense MFMA's raw code)
    static constexpr int NWP = (NIB + 7) / 8;    // words per plane
    static constexpr int SPW = 4 * NWP;          // words per point: [ha][plane V, N][word]
};

__global__ __launch_bounds__(256) void codes_sp_build_kernel(const uint32_t *__restrict__ codes, uint64_t n, uint32_t S,
                                                            uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t nw = S / 8, ns = (S - 4) / 4, nwp = (ns + 1 + 7) / 8;
    const uint32_t *w = codes + i * nw;
    auto code = [&](uint32_t sub) { return (w[sub >> 3] >> (4 * (sub & 7u))) & 15u; };
    uint32_t *o = out + i * 4 * nwp;
    for (uint32_t ha = 0; ha < 2; ++ha)
        for (uint32_t wi = 0; wi < nwp; ++wi) {
            uint32_t v = 0, nn = 0;
            for (uint32_t j = 8 * wi; j < 8 * wi + 8 && j <= ns; ++j) {
                uint32_t vn, nb;
                if (j < ns) {
                    const uint32_t ca = code(4 + 4 * j + ha), cb = code(4 + 4 * j + 2 + ha);
                    vn = (ca >> 2) | ((cb >> 2) << 2);
                    nb = (ca & 3u) | ((cb & 3u) << 2);
                } else {   // dense MFMA 0 scores subspace ha, dense MFMA 1 subspace 2 + ha
                    vn = code(ha);
                    nb = code(2 + ha);
                }
                v |= vn << (4 * (j & 7u));
                nn |= nb << (4 * (j & 7u));
            }
            o[(ha * 2 + 0) * nwp + wi] = v;
            o[(ha * 2 + 1) * nwp + wi] = nn;
        }
}

// inclusive prefix sum over the 64 lanes of a wave: DPP row shifts inside the 16-lane rows, then the row broadcasts
// (six v_add with a DPP operand; no LDS round trips)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return v;
}

#ifndef SCANN_SP_FLUSH_INLINE
#define SCANN_SP_FLUSH_INLINE __forceinline__
#endif
// The flush for FLAT hashers (one leaf, every pair sparse: ~0.4 survivors per bitmap word at C3): lane (col, h) keeps
// its own 32 words in registers and walks their bits itself, in rounds of kSpStage staged entries; the copy-out is the
// same as in sp_flush_item_words.  No per-pair round trip through LDS and the wave scan: 0.35 ms at C3 against
// 0.44 ms for the word-parallel form -- which wins wherever pairs are dense (tree indexes: 10M x 128, P = 25, m = 1000:
// scan 0.27 ms against 0.71 ms), because a lane walking its own survivors takes them one by one.
template <int S>
__device__ SCANN_SP_FLUSH_INLINE void sp_flush_item_lanes(const uint32_t *__restrict__ codes_sp, uint32_t *__restrict__ cand32_cnt,
                                                        uint32_t *__restrict__ cand32, uint32_t *__restrict__ cand32_codes,
                                                        uint32_t cap32, const uint32_t *bits, uint2 *stage, uint32_t *s_fq,
                                                        uint32_t *s_fvb, uint32_t *s_fgb, uint32_t ntile, uint32_t c0,
                                                        uint32_t lb, uint32_t pq, uint32_t vb) {
    constexpr int SPW = SpLayout<S>::SPW;
    constexpr int TTM = (int)(kMfmaRange / 64);
    const uint32_t lane = threadIdx.x & 63u, col = lane & 31u, h = lane >> 5;
    const uint32_t ntt = (ntile + 1u) >> 1;
    uint32_t w[TTM];
    uint32_t cnt = 0;
#pragma unroll
    for (int tt = 0; tt < TTM; ++tt) {
        w[tt] = (uint32_t)tt < ntt ? bits[tt * 64] : 0u;
        cnt += (uint32_t)__popc(w[tt]);
    }
    const uint32_t other = (uint32_t)__shfl_xor((int)cnt, 32);
    const uint32_t n_pair = cnt + other;
    uint32_t incl = n_pair;   // prefix over the pairs, computed alike in both halves of the wave
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 32);
        if ((int)col >= o) incl += up;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 31, 32);
    if (total == 0) return;
    uint32_t gbase = 0;
    if (h == 0 && n_pair) gbase = atomicAdd(&cand32_cnt[pq], n_pair);   // (padding pairs have no bits)
    if (lane < 32) {
        s_fq[lane] = pq == kInvalid ? 0u : pq;
        s_fvb[lane] = vb;
    }
    uint32_t sq = incl - n_pair + (h ? other : 0u);   // this lane's first entry in the wave's staging order
    uint32_t rel = h ? other : 0u;                    // ... and its position in the pair's segment
    for (uint32_t base = 0; base < total; base += kSpStage) {
        const uint32_t lim = base + kSpStage;
#pragma unroll
        for (int tt = 0; tt < TTM; ++tt) {
            while (w[tt] && sq < lim) {
                const uint32_t bpos = (uint32_t)__ffs((int)w[tt]) - 1u;
                w[tt] &= w[tt] - 1u;
                const uint32_t r = 15u - (bpos & 15u);
                const uint32_t jrel = (2u * (uint32_t)tt + (bpos >> 4)) * 32u + 4u * h + (r & 3u) + ((r >> 2) << 3);
                stage[sq - base] = make_uint2(rel, (col << 16) | jrel);
                ++sq;
                ++rel;
            }
        }
        if (base == 0 && lane < 32) s_fgb[lane] = gbase;   // (the atomics have travelled under the walk)
        __builtin_amdgcn_wave_barrier();
        const uint32_t n = min(kSpStage, total - base);
        for (uint32_t e0 = 0; e0 < n; e0 += 64u * kSpU) {
            uint2 ent[kSpU];
            uint4 cw[kSpU][SPW / 4];
#pragma unroll
            for (int u = 0; u < (int)kSpU; ++u) {
                const uint32_t e = e0 + lane + 64u * (uint32_t)u;
                ent[u] = e < n ? stage[e] : make_uint2(0xFFFFFFFFu, 0u);
            }
            if (cand32_codes) {   // (wave-uniform)
#pragma unroll
                for (int u = 0; u < (int)kSpU; ++u)
#pragma unroll
                    for (int x = 0; x < SPW / 4; ++x)
                        cw[u][x] = reinterpret_cast<const uint4 *>(codes_sp + (size_t)(lb + c0 + (ent[u].y & 0xFFFFu)) * SPW)[x];
            } else {   // (defined on every path: a conditionally initialised array stays in scratch memory -- 16 scratch
                       // round trips per copy-out, 0.41 instead of 0.35 ms at C3)
#pragma unroll
                for (int u = 0; u < (int)kSpU; ++u)
#pragma unroll
                    for (int x = 0; x < SPW / 4; ++x) cw[u][x] = make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < (int)kSpU; ++u) {
                if (ent[u].x != 0xFFFFFFFFu) {
                    const uint32_t c = ent[u].y >> 16;
                    const uint32_t dst = s_fgb[c] + ent[u].x;
                    if (dst < cap32) {
                        const size_t o = (size_t)s_fq[c] * cap32 + dst;
                        cand32[o] = s_fvb[c] + c0 + (ent[u].y & 0xFFFFu);
                        if (cand32_codes) {
#pragma unroll
                            for (int x = 0; x < SPW / 4; ++x) reinterpret_cast<uint4 *>(cand32_codes + o * SPW)[x] = cw[u][x];
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The flush of adc_smfmac_kernel: the item's survivor bitmap -> the queries' lists.  The tile loop left word
// [tt][h][col] = the masks of tiles 2 tt, 2 tt + 1 of lane (col, h).
//   1. every lane counts the bits of its own words; the two lanes of a pair share ONE returning atomic for the pair's
//      segment of the query's list (issued now, consumed in step 3);
//   2. pair by pair, the 64 lanes take the pair's 64 words ONE WORD EACH: a DPP prefix sum gives every word its offset
//      in the segment, and each lane stages (offset, pair, point) of its word's bits in LDS.  A pair whose every point
//      passes (a query's nearest leaf in a tree index) costs 32 rounds here, not the 2048 a lane walking its own
//      survivors one by one would need; a sparse pair (0.2 bits per word on a flat 1M index) costs two;
//   3. whenever the stage is full (kSpStage entries) or the pairs are done, the 64 lanes copy the staged entries to the
//      lists side by side, each with up to kSpU row loads in flight.  What travels with a position (flat hashers) is
//      the point's PLANE row (codes_sp: the lines the tile loop has just read, still in L2 -- the packed codes were
//      last touched at index creation); adc_refine_kernel decodes it (RefineArgs::planes).
template <int S>
__device__ SCANN_SP_FLUSH_INLINE void sp_flush_item_words(const uint32_t *__restrict__ codes_sp, uint32_t *__restrict__ cand32_cnt,
                                                        uint32_t *__restrict__ cand32, uint32_t *__restrict__ cand32_codes,
                                                        uint32_t cap32, const uint32_t *bits_w, uint2 *stage, uint32_t *s_fq,
                                                        uint32_t *s_fvb, uint32_t *s_fgb, uint32_t ntile, uint32_t c0,
                                                        uint32_t lb, uint32_t pq, uint32_t vb) {
    constexpr int SPW = SpLayout<S>::SPW;
    constexpr int TTM = (int)(kMfmaRange / 64);
    static_assert(TTM == 32, "the flush maps the 64 words of a pair onto the 64 lanes");
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    const uint32_t ntt = (ntile + 1u) >> 1;
    uint32_t cnt = 0;
#pragma unroll
    for (int tt = 0; tt < TTM; ++tt) cnt += (uint32_t)tt < ntt ? (uint32_t)__popc(bits_w[tt * 64 + lane]) : 0u;
    const uint32_t n_pair = cnt + (uint32_t)__shfl_xor((int)cnt, 32);   // (the same in both lanes of a pair)
    if (!__any(n_pair != 0)) return;
    uint32_t gbase = 0;
    if (h == 0 && n_pair) gbase = atomicAdd(&cand32_cnt[pq], n_pair);   // (padding pairs have no bits)
    if (lane < 32) {
        s_fq[lane] = pq == kInvalid ? 0u : pq;
        s_fvb[lane] = vb;
    }
    bool fgb_done = false;
    // step 3: stage[0 .. n) -> the lists
    auto copy_out = [&](uint32_t n) {
        if (!fgb_done) {   // (wave-uniform; the atomics have travelled under the first pairs' staging)
            if (lane < 32) s_fgb[lane] = gbase;
            fgb_done = true;
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t e0 = 0; e0 < n; e0 += 64u * kSpU) {
            uint2 ent[kSpU];
            uint4 cw[kSpU][SPW / 4];
#pragma unroll
            for (int u = 0; u < (int)kSpU; ++u) {
                const uint32_t e = e0 + lane + 64u * (uint32_t)u;
                ent[u] = e < n ? stage[e] : make_uint2(0xFFFFFFFFu, 0u);
            }
            if (cand32_codes) {   // (wave-uniform)
#pragma unroll
                for (int u = 0; u < (int)kSpU; ++u)
#pragma unroll
                    for (int x = 0; x < SPW / 4; ++x)
                        cw[u][x] = reinterpret_cast<const uint4 *>(codes_sp + (size_t)(lb + c0 + (ent[u].y & 0xFFFFu)) * SPW)[x];
            } else {   // (defined on every path: a conditionally initialised array stays in scratch memory -- 16 scratch
                       // round trips per copy-out, 0.41 instead of 0.35 ms at C3)
#pragma unroll
                for (int u = 0; u < (int)kSpU; ++u)
#pragma unroll
                    for (int x = 0; x < SPW / 4; ++x) cw[u][x] = make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < (int)kSpU; ++u) {
                if (ent[u].x != 0xFFFFFFFFu) {
                    const uint32_t c = ent[u].y >> 16;
                    const uint32_t dst = s_fgb[c] + ent[u].x;
                    if (dst < cap32) {
                        const size_t o = (size_t)s_fq[c] * cap32 + dst;
                        cand32[o] = s_fvb[c] + c0 + (ent[u].y & 0xFFFFu);
                        if (cand32_codes) {
#pragma unroll
                            for (int x = 0; x < SPW / 4; ++x) reinterpret_cast<uint4 *>(cand32_codes + o * SPW)[x] = cw[u][x];
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // step 2: lane = word (tt, hh) of the current pair
    const uint32_t tt = lane & 31u, hh = lane >> 5;
    const uint32_t woff = tt * 64u + hh * 32u;
    const bool wok = tt < ntt;
    auto jrel_of = [&](uint32_t bpos) {
        const uint32_t r = 15u - (bpos & 15u);
        return (2u * tt + (bpos >> 4)) * 32u + 4u * hh + (r & 3u) + ((r >> 2) << 3);
    };
    // (one copy_out site: the stage is filled with as many pairs -- or as much of a dense pair -- as fit, then copied)
    uint32_t c = 0, r0 = 0;   // current pair; entries of it already copied out (a dense pair spans several rounds)
    for (;;) {
        uint32_t fill = 0;
        while (c < 32u) {
            const uint32_t n_c = (uint32_t)__builtin_amdgcn_readlane((int)n_pair, (int)c);
            if (n_c == 0) {   // (wave-uniform)
                ++c;
                continue;
            }
            const uint32_t rem = n_c - r0;
            if (fill && fill + min(rem, kSpStage) > kSpStage) break;   // no room: copy out first
            const uint32_t take = min(rem, kSpStage - fill);           // entries [r0, r0 + take) of the pair's segment
            const uint32_t w = wok ? bits_w[woff + c] : 0u;
            const uint32_t p = (uint32_t)__popc(w);
            uint32_t x = w, i = wave_incl_scan(p) - p;
            while (x) {
                const uint32_t bpos = (uint32_t)__ffs((int)x) - 1u;
                x &= x - 1u;
                if (i - r0 < take) stage[fill + i - r0] = make_uint2(i, (c << 16) | jrel_of(bpos));
                ++i;
            }
            fill += take;
            r0 += take;
            if (r0 < n_c) break;   // a dense pair: the rest after this copy-out
            ++c;
            r0 = 0;
        }
        if (!fill) break;
        copy_out(fill);
    }
}

template <int S_, bool WORDS>   // WORDS: the word-parallel flush (tree indexes); else lanes walk their own words (flat)
__device__ __forceinline__ void adc_smfmac_body(const TxhIndexDev &ix, const MfmaArgs &a) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v8i __attribute__((ext_vector_type(8)));
    typedef int v16i __attribute__((ext_vector_type(16)));
    typedef SpLayout<S_> SP;
    constexpr int S = S_, NS = SP::NS, KT = NS + 2, NWP = SP::NWP, SPW = SP::SPW;
    constexpr int D = kMfmaDepth < KT ? kMfmaDepth : KT;      // operands in flight ahead of the MFMA that consumes them
    constexpr uint32_t kTT = kMfmaRange / 64;                   // tile pairs per item
    __shared__ __attribute__((aligned(16))) uint32_t s_ident[64];   // dense A: 16 one-hot rows of 16 bytes
    __shared__ __attribute__((aligned(16))) uint32_t s_vtab[64];    // sparse A values: row V = (ga | gb << 2)
    __shared__ uint32_t s_ntab[16];                                 // sparse A positions: word N = (ia | ib << 2)
    // survivor bitmap of the wave's item: word [tt][h][col] = the 16-bit masks of tiles 2 tt (low half) and 2 tt + 1 of
    // lane (col, h).  Written once per two tiles with one conflict-free ds_write_b32; no atomics, no branches and no
    // waits in the tile loop -- the item's flush turns it into list entries.
    __shared__ uint32_t s_bits[kMfmaWaves][kTT][64];
    __shared__ uint2 s_stage[kMfmaWaves][kSpStage];                                  // flush: staged list entries
    __shared__ uint32_t s_fq[kMfmaWaves][32], s_fvb[kMfmaWaves][32], s_fgb[kMfmaWaves][32];   // flush: per pair
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, h = lane >> 5;
    if (tid < 64) {
        const uint32_t c = tid >> 2, wsel = tid & 3u;
        s_ident[tid] = (wsel == (c >> 2)) ? (1u << (8 * (c & 3u))) : 0u;
        // row V, word wsel: words 0, 1 = bytes 0..7 (group ga = V & 3: value 1 at byte 2 ga), words 2, 3 = bytes 8..15 (gb = V >> 2)
        const uint32_t g = wsel < 2 ? (c & 3u) : (c >> 2);
        s_vtab[tid] = ((g >> 1) == (wsel & 1u)) ? (1u << (16 * (g & 1u))) : 0u;
        if (tid < 16) s_ntab[tid] = (tid & 3u) * 0x1111u | (tid >> 2) * 0x11110000u;
    }
    __syncthreads();
    const uint32_t total_tiles = a.counters[CNT_TOTAL_TILES];
    const char *ident = reinterpret_cast<const char *>(s_ident);
    const char *vtab = reinterpret_cast<const char *>(s_vtab);
    const char *ntab = reinterpret_cast<const char *>(s_ntab);
    uint32_t *bits = &s_bits[wave][0][lane];

    struct Planes {   // one tile's operand planes as LDS byte offsets (see step)
        uint32_t ve[NWP], vo[NWP], ne[NWP], no[NWP], d1;
    };
    struct Ops {      // the first D operands of a tile, fetched during the previous tile's MFMA chain
        v4i av[D];
        int iv[D];
    };

    uint32_t tile = 0;
    if (lane == 0) tile = grab_tile(a.counters + CNT_XQ, total_tiles);
    tile = __builtin_amdgcn_readfirstlane(tile);
    while (tile != kInvalid) {
        uint32_t next_tile = 0;
        if (lane == 0) next_tile = grab_tile(a.counters + CNT_XQ, total_tiles);
        uint32_t lo = 0, hi = ix.L;            // leaf = largest l with tile_off[l] <= tile
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (uniform_load(a.tile_off + mid) <= tile) lo = mid; else hi = mid;
        }
        const uint32_t leaf = lo;
        const uint32_t lb = uniform_load(ix.leaf_off + leaf);
        const uint32_t size = uniform_load(ix.leaf_off + leaf + 1) - lb;
        const uint32_t nranges = (size + kMfmaRange - 1) / kMfmaRange;
        const uint32_t local = tile - uniform_load(a.tile_off + leaf);
        const uint32_t range = local % nranges, pt = local / nranges;
        const uint32_t slot0 = uniform_load(a.pair_off + leaf);
        const uint32_t slot_end = uniform_load(a.pair_off + leaf + 1);
        const uint32_t c0 = range * kMfmaRange;
        const uint32_t npts = min(kMfmaRange, size - c0);

        // this lane's pair (column): tables (the pass bound is folded into them), key base
        const uint32_t slot = slot0 + pt * 32u + col;
        const bool pair_ok = slot < slot_end;
        const uint32_t pq = pair_ok ? a.pair_q[slot] : kInvalid;
        const uint32_t vb = pair_ok ? a.pair_vbase[slot] : 0u;
        v4i bd[2];
        v8i bs[NS];
        {
            const int8_t *bsrc = a.lut8 + (size_t)(pair_ok ? slot : slot0) * S * 16;
#pragma unroll
            for (int d = 0; d < 2; ++d) bd[d] = *reinterpret_cast<const v4i *>(bsrc + (size_t)(2 * d + h) * 16);
#pragma unroll
            for (int kt = 0; kt < NS; ++kt) {
                const v4i x0 = *reinterpret_cast<const v4i *>(bsrc + (size_t)(4 + 4 * kt + 2 * h) * 16);
                const v4i x1 = *reinterpret_cast<const v4i *>(bsrc + (size_t)(4 + 4 * kt + 2 * h) * 16 + 16);
                bs[kt] = v8i{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            }
            if (!pair_ok) {   // padding columns: every entry 127, sums stay positive (nothing passes)
                const int k7 = 0x7F7F7F7F;
#pragma unroll
                for (int d = 0; d < 2; ++d) bd[d] = v4i{k7, k7, k7, k7};
#pragma unroll
                for (int kt = 0; kt < NS; ++kt) bs[kt] = v8i{k7, k7, k7, k7, k7, k7, k7, k7};
            }
        }

        const uint32_t ntile = (npts + 31u) >> 5;
        static_assert(NWP == 1 || NWP == 2, "plane words");
        struct Raw {
            uint32_t wv[NWP], wn[NWP];
        };
        auto load_planes = [&](uint32_t t) {   // raw planes of tile t (rows past the leaf's end: any row, masked later)
            const uint32_t j = c0 + t * 32u + col;
            const uint32_t *src = ix.codes_sp + (size_t)(lb + (j < size ? j : 0u)) * SPW + h * 2u * NWP;
            Raw r;
            if constexpr (NWP == 1) {
                const uint2 v = *reinterpret_cast<const uint2 *>(src);
                r.wv[0] = v.x; r.wn[0] = v.y;
            } else {
                const uint4 v = *reinterpret_cast<const uint4 *>(src);
                r.wv[0] = v.x; r.wv[1] = v.y; r.wn[0] = v.z; r.wn[1] = v.w;
            }
            return r;
        };
        // planes -> LDS byte offsets: nibble j of the V plane times 16 (a 16-byte row), of the N plane times 4
        auto unpack = [&](const Raw &r) {
            Planes p;
#pragma unroll
            for (int wi = 0; wi < NWP; ++wi) {
                p.ve[wi] = r.wv[wi] & 0xF0F0F0F0u;
                p.vo[wi] = (r.wv[wi] << 4) & 0xF0F0F0F0u;
                p.ne[wi] = (r.wn[wi] >> 2) & 0x3C3C3C3Cu;
                p.no[wi] = (r.wn[wi] << 2) & 0x3C3C3C3Cu;
                // (opaque to the optimiser: it would otherwise re-derive every offset from the plane word with a
                // shift and a mask of its own -- two vector instructions per offset instead of one byte extraction)
                asm volatile("" : "+v"(p.ve[wi]), "+v"(p.vo[wi]), "+v"(p.ne[wi]), "+v"(p.no[wi]));
            }
            p.d1 = ((r.wn[NS >> 3] >> (4 * (NS & 7))) & 15u) << 4;   // dense MFMA 1: raw code, last nibble of the N plane
            return p;
        };
        auto voff = [&](const Planes &p, int j) { return (((j & 1) ? p.ve[j >> 3] : p.vo[j >> 3]) >> (8 * ((j & 7) >> 1))) & 0xFFu; };
        auto noff = [&](const Planes &p, int j) { return (((j & 1) ? p.ne[j >> 3] : p.no[j >> 3]) >> (8 * ((j & 7) >> 1))) & 0xFFu; };
        // operands of MFMA oi: 0, 1 dense (identity rows), 2 .. sparse (value row + position word)
        auto fetch = [&](const Planes &p, int oi, v4i &av, int &iv) {
            if (oi == 0) {
                av = *reinterpret_cast<const v4i *>(ident + voff(p, NS));
            } else if (oi == 1) {
                av = *reinterpret_cast<const v4i *>(ident + p.d1);
            } else {
                av = *reinterpret_cast<const v4i *>(vtab + voff(p, oi - 2));
                iv = *reinterpret_cast<const int *>(ntab + noff(p, oi - 2));
            }
        };
        // Software pipeline over the item's tiles.  step(t): the MFMA chain of tile t into accN; between its MFMAs
        // the survivor mask of tile t - 1 from accO (the sign of each result), the operand reads of the chain's
        // later MFMAs and -- in its last D slots -- of the FIRST D MFMAs of tile t + 1, so that no chain starts with
        // an exposed LDS round trip; the global load of tile t + 2's planes is issued at the top.  One extra step
        // drains the last tile (its MFMAs run on stale operands and are dropped).
        Raw rawn = load_planes(ntile > 1 ? 1u : 0u);
        Planes pl = unpack(load_planes(0u));
        Ops ops;
#pragma unroll
        for (int oi = 0; oi < D; ++oi) fetch(pl, oi, ops.av[oi], ops.iv[oi]);
        uint32_t mlo = 0;
        auto step = [&](v16i &accN, const v16i &accO, uint32_t t, auto hi_half) {
            const Planes pn = unpack(rawn);                        // tile t + 1 (loaded during step t - 1)
            if (t + 2 < ntile) rawn = load_planes(t + 2);
            v4i av[KT];
            int iv[KT];
            Ops nops;
#pragma unroll
            for (int oi = 0; oi < D; ++oi) {
                av[oi] = ops.av[oi];
                iv[oi] = ops.iv[oi];
            }
            // lane (col, h), register r: point row (r & 3) + 8 * (r >> 2) + 4 * h of the tile; result r's sign
            // (negative = passes) ends up at bit 15 - r of the mask
            uint32_t m16 = 0;
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int oi = 0; oi < KT; ++oi) {
                if (oi + D < KT) fetch(pl, oi + D, av[oi + D], iv[oi + D]);
                else fetch(pn, oi + D - KT, nops.av[oi + D - KT], nops.iv[oi + D - KT]);
                if (oi == 0)
                    accN = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[0], bd[0], v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
                else if (oi == 1)
                    accN = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[1], bd[1], accN, 0, 0, 0);
                else
                    accN = __builtin_amdgcn_smfmac_i32_32x32x64_i8(av[oi], bs[oi - 2], accN, iv[oi], 0, 0);
#pragma unroll
                for (int r = oi * 16 / KT; r < (oi + 1) * 16 / KT; ++r)
                    m16 = __builtin_amdgcn_alignbit(m16, (uint32_t)accO[r], 31);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            ops = nops;
            pl = pn;
            if (t == 0) return;                // nothing before the first tile (wave-uniform)
            if (t == ntile && (npts & 31u)) {  // partial last tile: rows past the leaf's end are padding
                const uint32_t base = c0 + (t - 1) * 32u + 4u * h;
                uint32_t okm = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    okm |= (base + (uint32_t)((r & 3) + 8 * (r >> 2)) < size ? 1u : 0u) << (15 - r);
                m16 &= okm;
            }
            if constexpr (decltype(hi_half)::value) {
                bits[((t - 1) >> 1) * 64u] = mlo | (m16 << 16);
            } else {
                mlo = m16;
            }
        };
        // (tile 0 is peeled: inside the loop t >= 1 is known, so the compiler keeps the mask build between
        // the MFMAs in BOTH instances instead of sinking it below a `t == 0` branch)
        v16i accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA;
        step(accA, accB, 0u, std::false_type());
        for (uint32_t tl = 1; tl <= ntile; tl += 2) {
            step(accB, accA, tl, std::false_type());                          // mask of tile tl - 1 (even): low half
            if (tl + 1 <= ntile) step(accA, accB, tl + 1, std::true_type());  // mask of tile tl (odd): high half, write
        }
        if (ntile & 1u) bits[(ntile >> 1) * 64u] = mlo;   // the last tile had an even number: its word has no high half

        // ---- flush: the item's bitmap -> the queries' lists (sp_flush_item: its own function, so that its registers
        // are allocated apart from the tile loop's -- inlined, the loop spilled its table fragments)
        if constexpr (WORDS)
            sp_flush_item_words<S>(ix.codes_sp, a.cand32_cnt, a.cand32, a.cand32_codes, a.cap32, &s_bits[wave][0][0], s_stage[wave],
                                   s_fq[wave], s_fvb[wave], s_fgb[wave], ntile, c0, lb, pq, vb);
        else
            sp_flush_item_lanes<S>(ix.codes_sp, a.cand32_cnt, a.cand32, a.cand32_codes, a.cap32, bits, s_stage[wave], s_fq[wave],
                                   s_fvb[wave], s_fgb[wave], ntile, c0, lb, pq, vb);
        tile = __builtin_amdgcn_readfirstlane(next_tile);
    }
}

// S <= 32: three waves per SIMD (the pair tile's table fragments alone are 64 registers); S = 48, 64: two.  (Capping the
// registers at 144 to leave room for a wave of another stream's kernel was tried: amdgpu_num_vgpr is ignored by this
// compiler, and two waves per SIMD -- SCANN_HIP_MFMA_WGS=2 -- cost the scan 10 % and gained the two-stream step nothing.)
#ifndef SCANN_SP_VGPRS
#define SCANN_SP_VGPRS 144
#endif
template <int S_, bool WORDS>
__global__ __launch_bounds__(kMfmaWaves * 64, SCANN_MFMA_MINW) __attribute__((amdgpu_num_vgpr(SCANN_SP_VGPRS)))
void adc_smfmac_kernel(TxhIndexDev ix, MfmaArgs a) {
    adc_smfmac_body<S_, WORDS>(ix, a);
}
template <int S_, bool WORDS>
__global__ __launch_bounds__(kMfmaWaves * 64, 2) void adc_smfmac_wide_kernel(TxhIndexDev ix, MfmaArgs a) {   // S = 48, 64
    adc_smfmac_body<S_, WORDS>(ix, a);
}

// The prefilter with 16-pair tiles on v_mfma_i32_16x16x64_i8, for leaves scanned by 8-24 queries of the batch
// (typical Tree-X-Hybrid batches: 1024 queries x 10 leaves over 1000 leaves): a 32-pair tile would be a
// third full there.  A tile = 32 points (two groups of 16) x 16 pairs = 2 x S/4 MFMAs of 4 subspaces each, two
// independent accumulator chains of 4 registers.  Lane (c16 = lane & 15, kb = lane >> 4): A row = point c16 of
// the group, K block kb = subspace 4 kt + kb (one-hot row from the LDS identity table); B column = pair c16;
// results D[row 4 kb + r][column c16], r = 0..3.  Items, bounds, staging, flush and lists as in adc_mfma_kernel
// (the worklist is built with 4 quads per tile).  Measured as a 32-pair kernel (two halves) this shape lost to
// adc_mfma_kernel (more vector work per tile); here it replaces the f32 LDS-gather scan.
template <int S_>
__global__ __launch_bounds__(kMfmaWaves * 64, (S_ <= 32 ? 4 : 2)) void adc_mfma16_kernel(TxhIndexDev ix, MfmaArgs a) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    constexpr int S = S_, KT = S / 4, NW = S / 8, NP = (NW + 1) / 2;
    __shared__ __attribute__((aligned(16))) uint32_t s_ident[64];                 // 16 one-hot rows of 16 bytes
    __shared__ uint32_t s_stage[kMfmaWaves][16][kMfmaStage];
    __shared__ uint32_t s_cnt[kMfmaWaves][16];
    __shared__ uint32_t s_fpre[kMfmaWaves][16], s_fq[kMfmaWaves][16], s_fgb[kMfmaWaves][16], s_fvb[kMfmaWaves][16];   // per pair
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t c16 = lane & 15u, kb = lane >> 4;
    if (tid < 64) {
        const uint32_t c = tid >> 2, wsel = tid & 3u;
        s_ident[tid] = (wsel == (c >> 2)) ? (1u << (8 * (c & 3u))) : 0u;
    }
    __syncthreads();
    const uint32_t total_tiles = a.counters[CNT_TOTAL_TILES];
    const char *ident = reinterpret_cast<const char *>(s_ident);

    uint32_t tile = 0;
    if (lane == 0) tile = grab_tile(a.counters + CNT_XQ, total_tiles);
    tile = __builtin_amdgcn_readfirstlane(tile);
    while (tile != kInvalid) {
        uint32_t next_tile = 0;
        if (lane == 0) next_tile = grab_tile(a.counters + CNT_XQ, total_tiles);
        uint32_t lo = 0, hi = ix.L;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (uniform_load(a.tile_off + mid) <= tile) lo = mid; else hi = mid;
        }
        const uint32_t leaf = lo;
        const uint32_t lb = uniform_load(ix.leaf_off + leaf);
        const uint32_t size = uniform_load(ix.leaf_off + leaf + 1) - lb;
        const uint32_t nranges = (size + kMfmaRange - 1) / kMfmaRange;
        const uint32_t local = tile - uniform_load(a.tile_off + leaf);
        const uint32_t range = local % nranges, pt = local / nranges;
        const uint32_t slot0 = uniform_load(a.pair_off + leaf);
        const uint32_t slot_end = uniform_load(a.pair_off + leaf + 1);
        const uint32_t c0 = range * kMfmaRange;
        const uint32_t npts = min(kMfmaRange, size - c0);

        // this lane's pair (column c16 of the tile): tables, bound; query and key base go to LDS for the flush
        const uint32_t slot = slot0 + pt * 16u + c16;
        const bool pair_ok = slot < slot_end;
        if (lane < 16) {
            s_cnt[wave][lane] = 0;
            s_fq[wave][lane] = pair_ok ? a.pair_q[slot] : kInvalid;
            s_fvb[wave][lane] = pair_ok ? a.pair_vbase[slot] : 0u;
        }
        v4i b[KT];
        {
            const int8_t *bsrc = a.lut8 + ((size_t)(pair_ok ? slot : slot0) * S + kb) * 16;   // padding columns: any table
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) b[kt] = *reinterpret_cast<const v4i *>(bsrc + (size_t)kt * 64);
        }
        const int thr1 = pair_ok ? a.thr1[slot] : -(128 * S + 7);   // a point passes iff acc - thr1 < 0

        const uint32_t ntile = (npts + 31u) >> 5;
        uint32_t wn[2][NW];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t j = c0 + 16u * g + c16;
            Codec<S, 4>::load_words(ix.codes + (size_t)(lb + (j < size ? j : 0u)) * NW, wn[g]);
        }
        auto flush = [&](bool all) {
            uint32_t n = 0, gbase = 0;
            if (lane < 16) {
                n = min(s_cnt[wave][lane], kMfmaStage);
                if (!all && n + 32u <= kMfmaStage) n = 0;
                if (n) {
                    gbase = atomicAdd(&a.cand32_cnt[s_fq[wave][lane]], n);
                    s_cnt[wave][lane] = 0;
                }
            }
            uint32_t incl = n;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                if ((int)lane >= o) incl += up;
            }
            const uint32_t total = (uint32_t)__shfl((int)incl, 15);
            if (total == 0) return;
            if (lane < 16) {
                s_fpre[wave][lane] = incl - n;
                s_fgb[wave][lane] = gbase;
            }
            for (uint32_t e = lane; e < total; e += 64u) {
                uint32_t c = 0;
#pragma unroll
                for (uint32_t stp = 8; stp; stp >>= 1)
                    if (s_fpre[wave][c + stp] <= e) c += stp;
                const uint32_t idx = e - s_fpre[wave][c];
                const uint32_t j = s_stage[wave][c][idx];
                const uint32_t dst = s_fgb[wave][c] + idx;
                if (dst < a.cap32) {
                    const size_t o = (size_t)s_fq[wave][c] * a.cap32 + dst;
                    a.cand32[o] = s_fvb[wave][c] + j;
                    if (a.cand32_codes) {
                        uint32_t cw[NW];
                        Codec<S, 4>::load_words(ix.codes + (size_t)(lb + j) * NW, cw);
                        Codec<S, 4>::store_words(a.cand32_codes + o * NW, cw);
                    }
                }
            }
        };
        // step(t): the MFMAs of tile t into accN, the survivor mask of tile t - 1 from accO between them, then
        // the staging of tile t - 1's survivors
        auto step = [&](v4i (&accN)[2], const v4i (&accO)[2], uint32_t t) {
            // byte kt' of pk[g][i] = code * 16 of subspace 4 kt + kb, kt = 4 i + {0, 2, 1, 3}[kt']
            uint32_t pk[2][NP];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const uint32_t y0 = (wn[g][2 * i] >> (4u * kb)) & 0x000F000Fu;
                    const uint32_t y1 = (2 * i + 1 < NW) ? ((wn[g][(2 * i + 1 < NW) ? 2 * i + 1 : 0] >> (4u * kb)) & 0x000F000Fu) : 0u;
                    pk[g][i] = (y0 | (y1 << 8)) << 4;
                }
            if (t + 1 < ntile) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const uint32_t j = c0 + (t + 1) * 32u + 16u * g + c16;
                    Codec<S, 4>::load_words(ix.codes + (size_t)(lb + (j < size ? j : 0u)) * NW, wn[g]);
                }
            }
            accN[0] = v4i{0, 0, 0, 0};
            accN[1] = v4i{0, 0, 0, 0};
            constexpr int NA = 2 * KT;                        // MFMAs per tile, in order (kt, g)
            constexpr int D = kMfmaDepth < NA ? kMfmaDepth : NA;
            v4i av[D + 1];
            auto onehot = [&](int ai) {
                const int kt = ai >> 1, g = ai & 1;
                const int byte = ((kt & 1) << 1) | ((kt >> 1) & 1);   // kt & 3 -> {0, 2, 1, 3}
                const uint32_t off = (pk[g][kt >> 2] >> (8 * byte)) & 0xFFu;
                return *reinterpret_cast<const v4i *>(ident + off);
            };
#pragma unroll
            for (int ai = 0; ai < D; ++ai) av[ai] = onehot(ai);
            uint32_t m8 = 0;   // survivor bits: result (g, r) at bit 7 - (4 g + r)
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int ai = 0; ai < NA; ++ai) {
                const int kt = ai >> 1, g = ai & 1;
                if (ai + D < NA) av[(ai + D) % (D + 1)] = onehot(ai + D);
                accN[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[ai % (D + 1)], b[kt], accN[g], 0, 0, 0);
#pragma unroll
                for (int ri = ai * 8 / NA; ri < (ai + 1) * 8 / NA; ++ri)
                    m8 = __builtin_amdgcn_alignbit(m8, (uint32_t)(accO[ri >> 2][ri & 3] - thr1), 31);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            if (t == 0) return;
            const uint32_t base = c0 + (t - 1) * 32u + 4u * kb;
            if (t == ntile && (npts & 31u)) {  // partial last tile: rows past the leaf's end are padding
                uint32_t okm = 0;
#pragma unroll
                for (int ri = 0; ri < 8; ++ri)
                    okm |= (base + 16u * (uint32_t)(ri >> 2) + (uint32_t)(ri & 3) < size ? 1u : 0u) << (7 - ri);
                m8 &= okm;
            }
            bool risk = false;
            m8 &= 0xFFu;
            if (m8) {
                uint32_t sl = atomicAdd(&s_cnt[wave][c16], (uint32_t)__popc(m8));
                do {
                    const uint32_t qi = 7u - ((uint32_t)__ffs((int)m8) - 1u);
                    m8 &= m8 - 1u;
                    const uint32_t j = base + 16u * (qi >> 2) + (qi & 3u);
                    if (sl < kMfmaStage) {
                        s_stage[wave][c16][sl] = j;
                    } else {   // stage full: direct (slow) append
                        const uint32_t pqd = s_fq[wave][c16];
                        const uint32_t pos = atomicAdd(&a.cand32_cnt[pqd], 1u);
                        if (pos < a.cap32) {
                            const size_t o = (size_t)pqd * a.cap32 + pos;
                            a.cand32[o] = s_fvb[wave][c16] + j;
                            if (a.cand32_codes) {
                                uint32_t cw[NW];
                                Codec<S, 4>::load_words(ix.codes + (size_t)(lb + j) * NW, cw);
                                Codec<S, 4>::store_words(a.cand32_codes + o * NW, cw);
                            }
                        }
                    }
                    ++sl;
                } while (m8);
                risk = sl + 32u > kMfmaStage;
            }
            if (__any(risk)) flush(false);
        };
        v4i accA[2], accB[2];
        accA[0] = accA[1] = accB[0] = accB[1] = v4i{0, 0, 0, 0};
        step(accA, accB, 0u);
        for (uint32_t tl = 1; tl <= ntile; tl += 2) {
            step(accB, accA, tl);
            if (tl + 1 <= ntile) step(accA, accB, tl + 1);
        }
        flush(true);
        tile = __builtin_amdgcn_readfirstlane(next_tile);
    }
}

// Exact refine of the prefilter's survivors: block per query.  Recomputes the reference's f32 sums
// (LookupTable::compute_distance, hashes/lut.rs:74-82: acc = lut[0][c0]; acc += lut[s][cs], s
// ascending), forms the merge keys and keeps key <= T -- exactly adc_scan_kernel's survivors.
struct RefineArgs {
    uint32_t P, cap, cap32;
    const uint32_t *tokens, *vbase, *slot_of;
    const float *lutq;
    const uint64_t *thr;
    const uint32_t *cand32_cnt, *cand32, *cand32_codes;
    uint32_t *cand_cnt;
    uint64_t *cand;
    uint32_t *counters;
    const uint64_t *allow;
    uint64_t allow_bits;
    int planes;   // cand32_codes holds codes_sp plane rows (adc_smfmac_kernel), not packed codes
};

// code of subspace s from a point's plane row (SpLayout: [ha][plane V, N][word]); ca / cb = the code nibbles of the
// first / second subspace each sparse MFMA takes from parity ha, rebuilt word-parallel by sp_row_decode
template <int S>
struct SpRow {
    static constexpr int NWP = SpLayout<S>::NWP, NS = SpLayout<S>::NS;
    uint32_t ca[2][NWP], cb[2][NWP], dv[2], dn[2];   // dv / dn: the dense MFMAs' raw codes (subspaces ha, 2 + ha)
    __device__ __forceinline__ void decode(const uint32_t *row) {
#pragma unroll
        for (int ha = 0; ha < 2; ++ha) {
#pragma unroll
            for (int wi = 0; wi < NWP; ++wi) {
                const uint32_t v = row[(ha * 2 + 0) * NWP + wi], n = row[(ha * 2 + 1) * NWP + wi];
                ca[ha][wi] = ((v & 0x33333333u) << 2) | (n & 0x33333333u);
                cb[ha][wi] = (v & 0xCCCCCCCCu) | ((n >> 2) & 0x33333333u);
            }
            dv[ha] = (row[(ha * 2 + 0) * NWP + (NS >> 3)] >> (4 * (NS & 7))) & 15u;
            dn[ha] = (row[(ha * 2 + 1) * NWP + (NS >> 3)] >> (4 * (NS & 7))) & 15u;
        }
    }
    __device__ __forceinline__ uint32_t code(int s) const {   // s: compile-time after unrolling
        if (s < 4) return (s >> 1) ? dn[s & 1] : dv[s & 1];
        const int kt = (s - 4) >> 2, q = (s - 4) & 3, ha = q & 1;
        const uint32_t src = (q >> 1) ? cb[ha][kt >> 3] : ca[ha][kt >> 3];
        return (src >> (4 * (kt & 7))) & 15u;
    }
};

#ifndef SCANN_REFINE_THREADS
#define SCANN_REFINE_THREADS 256
#endif
#ifndef SCANN_REFINE_U
#define SCANN_REFINE_U 4
#endif
constexpr uint32_t kRefineThreads = SCANN_REFINE_THREADS;

template <typename C>
__global__ __launch_bounds__(kRefineThreads) void adc_refine_kernel(TxhIndexDev ix, RefineArgs a) {
    constexpr int S = C::S, NW = C::NWORDS;
    // words per list entry: packed codes, or (4-bit codes behind the sparse-MFMA prefilter) the point's plane row
    constexpr int SPW = C::BITS == 4 ? (int)(4 * ((((S - 4) / 4 + 1) + 7) / 8)) : NW;
    constexpr int RW = SPW > NW ? SPW : NW;
    const bool planes = C::BITS == 4 && a.planes;   // (block-uniform)
    const uint32_t ew = planes ? (uint32_t)SPW : (uint32_t)NW;
    extern __shared__ __attribute__((aligned(16))) float s_tab[];       // [min(P, kRefineTablesMax)][S][16]
    __shared__ uint32_t s_dvb[kDecodeStage], s_drow[kDecodeStage], s_slot[kDecodeStage], s_out;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t P = a.P;
    const uint32_t cnt = a.cand32_cnt[q];
    if (cnt > a.cap32) {   // list overflow: report, never a wrong row
        if (tid == 0) {
            atomicMax(&a.counters[CNT_STATUS], (uint32_t)SCANN_HIP_RESOURCE_EXHAUSTED);
            a.cand_cnt[q] = a.cap + 1u;
        }
        return;
    }
    const uint64_t T = a.thr[q];
    const bool staged = P <= kDecodeStage;
    const bool tabs = P <= kRefineTablesMax;
    const uint32_t *vbq = a.vbase + (size_t)q * (P + 1);
    if (staged)
        for (uint32_t r = tid; r < P; r += kRefineThreads) {
            s_dvb[r] = vbq[r];
            s_drow[r] = ix.leaf_off[a.tokens[(size_t)q * P + r]];
            s_slot[r] = a.slot_of[(size_t)q * P + r];
        }
    if (tid == 0) s_out = 0;
    __syncthreads();
    if (tabs) {   // this query's pair tables, de-interleaved: [r][s][16]
        for (uint32_t e = tid; e < P * S * 16; e += kRefineThreads) {
            const uint32_t r = e / (S * 16), sc = e - r * (S * 16);
            const uint32_t slot = staged ? s_slot[r] : a.slot_of[(size_t)q * P + r];
            s_tab[e] = slot == kInvalid ? 0.0f : a.lutq[((size_t)(slot >> 2) * S * 16 + sc) * 4 + (slot & 3u)];
        }
        __syncthreads();
    }
    uint64_t *out = a.cand + (size_t)q * a.cap;
    const uint32_t *list = a.cand32 + (size_t)q * a.cap32;
    const uint32_t *list_codes = a.cand32_codes ? a.cand32_codes + (size_t)q * a.cap32 * ew : nullptr;
    constexpr int U = SCANN_REFINE_U;   // entries per thread per pass: their dependent loads (position -> codes) overlap
    for (uint32_t b0 = 0; b0 < cnt; b0 += kRefineThreads * U) {
        uint32_t vpos[U], csr[U], lo_[U];
        uint32_t w[U][RW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = b0 + tid + kRefineThreads * u;
            vpos[u] = e < cnt ? list[e] : 0xFFFFFFFFu;
            if (a.cand32_codes) {   // (written with the position)
                const uint32_t *src = list_codes + (size_t)(e < cnt ? e : 0u) * ew;
                if (planes) {
#pragma unroll
                    for (int x = 0; x < SPW / 4; ++x) {
                        const uint4 v = reinterpret_cast<const uint4 *>(src)[x];
                        w[u][4 * x] = v.x; w[u][4 * x + 1] = v.y; w[u][4 * x + 2] = v.z; w[u][4 * x + 3] = v.w;
                    }
                } else {
                    uint32_t t[NW];
                    C::load_words(src, t);
#pragma unroll
                    for (int x = 0; x < NW; ++x) w[u][x] = t[x];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t lo = 0, hi = P;
            const uint32_t vp = vpos[u] == 0xFFFFFFFFu ? 0u : vpos[u];
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((staged ? s_dvb[mid] : vbq[mid]) <= vp) lo = mid; else hi = mid;
            }
            lo_[u] = lo;
            csr[u] = (staged ? s_drow[lo] : ix.leaf_off[a.tokens[(size_t)q * P + lo]]) + (vp - (staged ? s_dvb[lo] : vbq[lo]));
            if (!a.cand32_codes) {
                uint32_t t[NW];
                C::load_words(ix.codes + (size_t)(vpos[u] == 0xFFFFFFFFu ? 0u : csr[u]) * NW, t);
#pragma unroll
                for (int x = 0; x < NW; ++x) w[u][x] = t[x];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bool keep = false;
            uint64_t key = 0;
            if (vpos[u] != 0xFFFFFFFFu) {
                float acc = 0.0f;
                bool done = false;
                if constexpr (C::BITS == 4) {
                    if (planes && tabs) {   // (the sparse prefilter runs with staged tables: P <= kRefineTablesMax or not, both forms)
                        SpRow<S> row;
                        row.decode(w[u]);
                        const float *tb = s_tab + lo_[u] * (S * 16);
#pragma unroll
                        for (int s2 = 0; s2 < S; ++s2) {
                            const float tv = tb[s2 * 16 + row.code(s2)];
                            acc = s2 == 0 ? tv : acc + tv;
                        }
                        done = true;
                    } else if (planes) {
                        SpRow<S> row;
                        row.decode(w[u]);
                        const uint32_t slot = staged ? s_slot[lo_[u]] : a.slot_of[(size_t)q * P + lo_[u]];
                        const float *tb = a.lutq + (size_t)(slot >> 2) * S * 64 + (slot & 3u);
#pragma unroll
                        for (int s2 = 0; s2 < S; ++s2) {
                            const float tv = tb[(s2 * 16 + row.code(s2)) * 4];
                            acc = s2 == 0 ? tv : acc + tv;
                        }
                        done = true;
                    }
                }
                if (done) {
                } else if (tabs) {
                    const float *tb = s_tab + lo_[u] * (S * 16);
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2) {
                        const uint32_t code = (w[u][s2 >> 3] >> (4 * (s2 & 7))) & 15u;
                        const float tv = tb[s2 * 16 + code];
                        acc = s2 == 0 ? tv : acc + tv;
                    }
                } else {
                    const uint32_t slot = staged ? s_slot[lo_[u]] : a.slot_of[(size_t)q * P + lo_[u]];
                    const float *tb = a.lutq + (size_t)(slot >> 2) * S * 64 + (slot & 3u);
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2) {   // (fully unrolled: a dynamic index would push w[] to scratch)
                        const uint32_t code = (w[u][s2 >> 3] >> (4 * (s2 & 7))) & 15u;
                        const float tv = tb[(s2 * 16 + code) * 4];
                        acc = s2 == 0 ? tv : acc + tv;
                    }
                }
                key = make_key(acc, vpos[u]);
                keep = key <= T && row_allowed(ix, a.allow, a.allow_bits, csr[u]);
            }
            uint32_t wtot;
            const uint32_t wpre = wave_prefix_count(keep, &wtot);
            uint32_t base = 0;
            if (lane == 0 && wtot) base = atomicAdd(&s_out, wtot);
            base = (uint32_t)__shfl((int)base, 0);
            if (keep && base + wpre < a.cap) out[base + wpre] = key;
        }
    }
    __syncthreads();
    if (tid == 0) a.cand_cnt[q] = s_out;   // > cap: select_rerank reports the overflow
}

// =====================================================================================
// K5: threshold from a strided sample (plan: sample_stride / sample_plan / sample_rank).
//
// K5a adc_sample_kernel: the scan's tiled LUT16 gather over every st-th point of each
// selected leaf.  Tiles = (leaf, chunk of (uint32_t)C::TP SAMPLED points, group of a.qpt query
// quads); the ordered approximate distance of sample i of (query, leaf) goes to
// samp[query][sbase(query, leaf) + i].  Points the allow-bitmap rejects are written as
// 0xFFFFFFFF (absent).
// K5b threshold_select_kernel: block per query; the j-th smallest sample by one LDS
// histogram over [min, max] of the sample + an exact rank inside the j-th's bin.
// =====================================================================================
struct SampleArgs {
    const uint32_t *pair_off, *stile_off, *pair_q, *pair_sbase;
    uint32_t *counters;
    const float *lutq;
    uint32_t *samp;
    uint32_t scap, st, qpt;
    const uint64_t *allow;
    uint64_t allow_bits;
};

template <typename C>
__global__ __launch_bounds__(kScanThreads, C::WGS) void adc_sample_kernel(TxhIndexDev ix, SampleArgs a) {
    constexpr int LUT4 = C::LUT4;
    constexpr int STG = (LUT4 + kScanThreads - 1) / kScanThreads;
    extern __shared__ __attribute__((aligned(16))) float4 lut_s[];   // [2 * LUT4] + tile slot
    uint32_t &tile_sh = *reinterpret_cast<uint32_t *>(lut_s + 2 * LUT4);
    const uint32_t tid = threadIdx.x;
    const uint32_t total_tiles = a.counters[CNT_TOTAL_STILES];
    const uint32_t st = a.st;

    for (;;) {
        if (tid == 0) tile_sh = grab_tile(a.counters + CNT_XS, total_tiles);
        __syncthreads();
        const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_sh);
        if (tile == kInvalid) break;

        uint32_t lo = 0, hi = ix.L;   // leaf = largest l with stile_off[l] <= tile
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (a.stile_off[mid] <= tile) lo = mid; else hi = mid;
        }
        const uint32_t leaf = lo;
        const uint32_t lb = ix.leaf_off[leaf];
        const uint32_t size = ix.leaf_off[leaf + 1] - lb;
        const uint32_t ssize = (size + st - 1) / st;            // sampled points of the leaf
        const uint32_t nchunks = (ssize + (uint32_t)C::TP - 1) / (uint32_t)C::TP;
        const uint32_t local = tile - a.stile_off[leaf];
        const uint32_t chunk = local % nchunks, qg = local / nchunks;
        const uint32_t slot0 = a.pair_off[leaf];
        const uint32_t nquads = (a.pair_off[leaf + 1] - slot0) >> 2;
        const uint32_t q0 = qg * a.qpt;
        const uint32_t q1 = min(q0 + a.qpt, nquads);
        const uint32_t c0 = chunk * (uint32_t)C::TP;
        const uint32_t npts = min((uint32_t)C::TP, ssize - c0);
        const uint32_t nsub = (npts + kScanThreads - 1) / kScanThreads;

        uint32_t regs[C::PPT][C::REGS];
        bool ok[C::PPT];
#pragma unroll
        for (int i = 0; i < C::PPT; ++i) {
            const uint32_t j = c0 + tid + kScanThreads * i;
            const uint32_t row = lb + (j < ssize ? j * st : 0u);
            ok[i] = j < ssize && row_allowed(ix, a.allow, a.allow_bits, row);
            uint32_t w[C::NWORDS];
            C::load_words(ix.codes + (size_t)row * C::NWORDS, w);
            C::unpack(w, regs[i]);
        }

        const float4 *gl = reinterpret_cast<const float4 *>(a.lutq) +
                           (size_t)((slot0 >> 2) + q0) * LUT4;
        {
            float4 r[STG];
#pragma unroll
            for (int t = 0; t < STG; ++t) {
                uint32_t e = tid + t * kScanThreads;
                if (e < (uint32_t)LUT4) r[t] = gl[e];
            }
#pragma unroll
            for (int t = 0; t < STG; ++t) {
                uint32_t e = tid + t * kScanThreads;
                if (e < (uint32_t)LUT4) lut_s[e] = r[t];
            }
        }
        __syncthreads();

        for (uint32_t qd = q0; qd < q1; ++qd) {
            const uint32_t buf = (qd - q0) & 1u;
            if (qd + 1 < q1) {   // LDS-DMA prefetch of the next quad's LUT into the idle buffer
                const float4 *g2 = gl + (size_t)(qd + 1 - q0) * LUT4;
#pragma unroll
                for (int t = 0; t < STG; ++t) {
                    const uint32_t e0 = (tid & ~63u) + t * kScanThreads;   // wave-uniform
                    if (e0 < (uint32_t)LUT4)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void *)(g2 + e0 + (tid & 63u)),
                            (__attribute__((address_space(3))) void *)(&lut_s[(buf ^ 1u) * LUT4 + e0]),
                            16, 0, 0);
                }
            }
            const uint32_t slot = slot0 + qd * 4;
            uint32_t f_pq[4], f_sb[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f_pq[p] = __builtin_amdgcn_readfirstlane(a.pair_q[slot + p]);
                f_sb[p] = __builtin_amdgcn_readfirstlane(a.pair_sbase[slot + p]);
            }
            float acc[4][C::PPT];
            if (buf == 0) scan_quad_dispatch<C, 0>(lut_s, nsub, regs, acc);
            else scan_quad_dispatch<C, 1>(lut_s, nsub, regs, acc);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (f_pq[p] == kInvalid) continue;   // wave-uniform
                uint32_t *dst = a.samp + (size_t)f_pq[p] * a.scap + f_sb[p];
#pragma unroll
                for (int i = 0; i < C::PPT; ++i) {
                    const uint32_t j = c0 + tid + kScanThreads * i;
                    if (i < (int)nsub && j < ssize) dst[j] = ok[i] ? f32_to_ordered(acc[p][i]) : 0xFFFFFFFFu;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA prefetch has landed
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kSelectThreads) void threshold_select_kernel(
    uint32_t P, uint32_t m, uint32_t st, int no_threshold, const uint32_t *__restrict__ sbase,
    const uint32_t *__restrict__ samp, uint32_t scap, const uint32_t *__restrict__ slot_of,
    uint64_t *__restrict__ thr, uint64_t *__restrict__ pair_thr, const uint32_t *__restrict__ vbase) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_vals[];   // [scap rounded to 4]
    const SelCfg cfg = sel_cfg(scap);
    uint32_t *s_hist = s_vals + ((scap + 3u) & ~3u);                    // [cfg.bins]
    uint32_t *s_list = s_hist + cfg.bins;                               // [cfg.list]
    uint64_t *s_red = reinterpret_cast<uint64_t *>(s_list + cfg.list);  // [48]
    const uint32_t q = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t ns = min(sbase[(size_t)q * (P + 2) + P], scap);
    const uint32_t total = sbase[(size_t)q * (P + 2) + P + 1];
    const uint32_t J = (no_threshold || total <= m) ? 0u : sample_rank(m, st);
    // the scan reads the bound per (query, leaf) pair slot: no dependent pair_q -> thr load
    auto publish = [&](uint64_t T) {
        if (tid == 0) thr[q] = T;
        for (uint32_t r = tid; r < P; r += nt) {
            const uint32_t sl = slot_of[(size_t)q * P + r];
            if (sl != kInvalid) pair_thr[sl] = T;
        }
    };
    if (J == 0 || ns < J) {   // block-uniform
        publish(SCANN_KEY_MAX);
        return;
    }
    // sample -> LDS (16-byte loads; rows of samp are 16-byte aligned: scap % 4 == 0)
    const uint4 *src = reinterpret_cast<const uint4 *>(samp + (size_t)q * scap);
    const uint32_t n4 = (ns + 3u) >> 2;
    for (uint32_t i0 = 0; i0 < n4; i0 += 4 * nt) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * nt + tid;
            v[u] = i < n4 ? src[i] : make_uint4(~0u, ~0u, ~0u, ~0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * nt + tid;
            if (i < n4) {
                // entries past ns (row padding) count as absent
                if (4 * i + 1 >= ns) v[u].y = ~0u;
                if (4 * i + 2 >= ns) v[u].z = ~0u;
                if (4 * i + 3 >= ns) v[u].w = ~0u;
                reinterpret_cast<uint4 *>(s_vals)[i] = v[u];
            }
        }
    }
    __syncthreads();
    // absent samples (0xFFFFFFFF: rejected by the allow-bitmap, padding) sort last; the bound
    // is MAX if the J-th smallest is one of them
    const uint32_t v = block_select<uint32_t>(s_vals, 4 * n4, J, cfg, s_hist, s_list, s_red);
    if (v == 0xFFFFFFFFu) {
        publish(SCANN_KEY_MAX);
        return;
    }
    // Ties at the bound.  Coarse codes over clustered data give thousands of points the SAME approximate
    // distance; a bound on the distance alone lets the whole tie group through (measured: 2.5x the expected
    // survivors, candidate buffers overflowing).  The merge keys (distance, stream position) are unique, so the
    // bound is the J-th smallest sample KEY: among the samples that tie on the distance, the one with the
    // (J - #smaller)-th smallest stream position (sample i of token r is point i * st of that leaf).
    __syncthreads();
    uint32_t *s_cnt2 = reinterpret_cast<uint32_t *>(s_red);   // [0] smaller, [1] tied, [2] cursor, [3] result
    if (tid < 4) s_cnt2[tid] = tid == 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    uint32_t less = 0, tied = 0;
    for (uint32_t i = tid; i < 4 * n4; i += nt) {
        const uint32_t x = s_vals[i];
        less += x < v ? 1u : 0u;
        tied += x == v ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        less += (uint32_t)__shfl_xor((int)less, o);
        tied += (uint32_t)__shfl_xor((int)tied, o);
    }
    if ((tid & 63u) == 0) {
        atomicAdd(&s_cnt2[0], less);
        atomicAdd(&s_cnt2[1], tied);
    }
    __syncthreads();
    const uint32_t a_less = s_cnt2[0], b_tied = s_cnt2[1];
    uint32_t low = 0xFFFFFFFFu;   // (a single sample at the bound, or a tie group too large to rank: all of it passes)
    if (vbase && b_tied > 1 && b_tied <= cfg.list && J > a_less) {   // block-uniform
        uint32_t *s_vp = s_hist;                                   // [b_tied] stream positions of the tied samples
        const uint32_t *sb = sbase + (size_t)q * (P + 2), *vb = vbase + (size_t)q * (P + 1);
        for (uint32_t i = tid; i < 4 * n4; i += nt) {
            if (s_vals[i] == v) {
                uint32_t lo = 0, hi = P;   // token of sample slot i: largest r with sb[r] <= i
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sb[mid] <= i) lo = mid; else hi = mid;
                }
                s_vp[atomicAdd(&s_cnt2[2], 1u)] = vb[lo] + (i - sb[lo]) * st;
            }
        }
        __syncthreads();
        const uint32_t need = J - a_less;                          // 1-based rank inside the tie group
        for (uint32_t i = tid; i < b_tied; i += nt) {
            const uint32_t x = s_vp[i];
            uint32_t r = 0;
            for (uint32_t j2 = 0; j2 < b_tied; ++j2) r += s_vp[j2] < x ? 1u : 0u;
            if (r + 1 == need) s_cnt2[3] = x;                      // (positions are unique)
        }
        __syncthreads();
        low = s_cnt2[3];
    }
    publish(((uint64_t)v << 32) | low);
}

// K5c: the same bound from the LOW TAIL of the sample only.  The bound is the J-th smallest of ~32 k sample keys with
// J a few hundred: the full histogram select above stages all of them in LDS (128 KB: one workgroup per compute unit,
// 1024 threads, ~10 workgroup barriers per pass) and makes five passes over them.  Here a 256-thread workgroup (eight
// per compute unit: every query of a 1024-query batch is resident at once) reads the sample from global memory (L2-hot:
// the sample pass has just written it) with 16-byte loads; every thread keeps the TWO smallest of its samples, and
// the J-th smallest of those 512 kept values is a pivot at or above the J-th smallest sample (any subset's J-th
// smallest is) and within a few percent of it; one more pass collects the (value, slot) keys at or under the pivot --
// a little over J of them -- and the J-th smallest of THOSE is the bound.  Sample slots grow with the stream position
// (sbase and vbase are prefixes in the same token order, samples of a leaf are st points apart), so ordering by
// (value, slot) IS ordering by the merge key (value, position): no separate ranking of the tie group.  If ties flood
// the list (more than kThrTailList samples at or under the pivot) the bound is (pivot, MAX): valid (pivot >= the J-th
// smallest), merely looser.
constexpr uint32_t kThrTailThreads = 256;
constexpr uint32_t kThrTailList = 2048;
constexpr uint32_t kThrTailMaxRank = 384;   // J above this takes threshold_select_kernel

__global__ __launch_bounds__(kThrTailThreads) void threshold_tail_kernel(
    uint32_t P, uint32_t m, uint32_t st, const uint32_t *__restrict__ sbase, const uint32_t *__restrict__ samp,
    uint32_t scap, const uint32_t *__restrict__ slot_of, uint64_t *__restrict__ thr, uint64_t *__restrict__ pair_thr,
    const uint32_t *__restrict__ vbase) {
    __shared__ uint32_t s_kept[2 * kThrTailThreads];
    __shared__ uint32_t s_hist[1024];
    __shared__ uint64_t s_slist[256];
    __shared__ uint64_t s_red[48];
    __shared__ uint64_t s_keys[kThrTailList];
    __shared__ uint32_t s_cnt[2];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, nt = kThrTailThreads;
    const uint32_t ns = min(sbase[(size_t)q * (P + 2) + P], scap);
    const uint32_t total = sbase[(size_t)q * (P + 2) + P + 1];
    const uint32_t J = total <= m ? 0u : sample_rank(m, st);
    auto publish = [&](uint64_t T) {
        if (tid == 0) thr[q] = T;
        for (uint32_t r = tid; r < P; r += nt) {
            const uint32_t sl = slot_of[(size_t)q * P + r];
            if (sl != kInvalid) pair_thr[sl] = T;
        }
    };
    if (J == 0 || ns < J) {   // block-uniform
        publish(SCANN_KEY_MAX);
        return;
    }
    // rows of samp are 16-byte aligned (scap % 4 == 0); entries past ns count as absent
    const uint4 *src = reinterpret_cast<const uint4 *>(samp + (size_t)q * scap);
    const uint32_t n4 = (ns + 3u) >> 2;
    auto load4 = [&](uint32_t i) {
        uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (i < n4) {
            v = src[i];
            if (4 * i + 1 >= ns) v.y = ~0u;
            if (4 * i + 2 >= ns) v.z = ~0u;
            if (4 * i + 3 >= ns) v.w = ~0u;
        }
        return v;
    };
    uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;   // the two smallest of this thread's samples, m0 <= m1
    auto keep = [&](uint32_t x) {
        const uint32_t hi = x > m0 ? x : m0;
        m0 = x < m0 ? x : m0;
        m1 = hi < m1 ? hi : m1;
    };
    constexpr int LU = 8;   // 16-byte loads in flight per thread
    for (uint32_t i0 = 0; i0 < n4; i0 += LU * nt) {
        uint4 v[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) v[u] = load4(i0 + u * nt + tid);
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            keep(v[u].x); keep(v[u].y); keep(v[u].z); keep(v[u].w);
        }
    }
    s_kept[2 * tid] = m0;
    s_kept[2 * tid + 1] = m1;
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    const SelCfg cfg = sel_cfg(kThrTailList);   // bins 1024, list 256
    uint32_t pivot = block_select<uint32_t>(s_kept, 2 * nt, J, cfg, s_hist, reinterpret_cast<uint32_t *>(s_slist), s_red);
    __syncthreads();
    // (absent samples -- rejected by the allow-bitmap -- sort last: with fewer than J kept values present the pivot is
    // "every present value")
    if (pivot == 0xFFFFFFFFu) pivot = 0xFFFFFFFEu;
    for (uint32_t i0 = 0; i0 < n4; i0 += LU * nt) {
        uint4 v[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) v[u] = load4(i0 + u * nt + tid);
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const uint32_t i = 4 * (i0 + u * nt + tid);
            const uint32_t x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (x[c] <= pivot) {
                    const uint32_t pos = atomicAdd(&s_cnt[0], 1u);
                    if (pos < kThrTailList) s_keys[pos] = ((uint64_t)x[c] << 32) | (i + c);
                }
        }
    }
    __syncthreads();
    const uint32_t n_le = s_cnt[0];
    if (n_le < J) {   // fewer than J present samples in all: no bound
        publish(SCANN_KEY_MAX);
        return;
    }
    if (n_le > kThrTailList) {   // ties flood the list: the pivot is >= the J-th smallest value; all of that distance pass
        publish(((uint64_t)pivot << 32) | 0xFFFFFFFFu);
        return;
    }
    const uint64_t key = block_select<uint64_t>(s_keys, n_le, J, cfg, s_hist, s_slist, s_red);
    const uint32_t v = (uint32_t)(key >> 32), slot = (uint32_t)key;
    if (!vbase) {   // diagnostic mode: the bound on the distance alone, whole tie group
        publish(((uint64_t)v << 32) | 0xFFFFFFFFu);
        return;
    }
    // stream position of that sample: token r = the largest with sb[r] <= slot
    const uint32_t *sb = sbase + (size_t)q * (P + 2), *vb = vbase + (size_t)q * (P + 1);
    uint32_t lo = 0, hi = P;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sb[mid] <= slot) lo = mid; else hi = mid;
    }
    publish(((uint64_t)v << 32) | (vb[lo] + (slot - sb[lo]) * st));
}

// =====================================================================================
// K6c: exact leaf scan -- Scann::search_partitioned (scann.rs:213-252).  Every row of the P
// selected leaves is scored with the configured measure's single-pair kernel
// (DistanceMeasure::distance, distance_measures/mod.rs:70-81 -> simd/x86.rs:72-96 / :139-165: 8
// FMA lane chains, fixed hsum tree, scalar tail); candidates in token order then leaf order,
// stable sort, first k.  Tile = (leaf, 256-row chunk, group of quads): a thread owns one row and
// walks the tile's queries eight at a time from LDS, the lane chains as packed-f32 pairs
// (v_pk_fma_f32).  Every (query, stream position) is written exactly once into the query's
// dense key list [vbase[P]], from which select_rerank_kernel takes the k smallest keys.
// =====================================================================================
constexpr uint32_t kExactRows = 256;   // rows per tile chunk (one per thread)
constexpr uint32_t kExactQuadsMax = 16; // quads per tile: all of them staged in LDS once per tile
constexpr int kExactQT = 8;            // queries per pass over the row
// quads per tile for this dimensionality: the tile's queries must fit 48 KB of LDS
static inline uint32_t exact_quads_per_tile(uint32_t dim) {
    const uint32_t dimp = (dim + 3u) & ~3u;
    uint32_t q = (48u * 1024u) / (dimp * 4u) / 8u * 2u;   // whole passes of 8 queries, in quads
    return q < 2u ? 2u : (q > kExactQuadsMax ? kExactQuadsMax : q);
}

struct ExactScanArgs {
    const uint32_t *pair_off, *tile_off, *pair_q, *pair_vbase;
    uint32_t *counters;
    const float *queries;
    uint32_t q_stride;
    uint64_t *cand;
    uint32_t cap;
    uint32_t qpt;   // quads per tile (exact_quads_per_tile)
};

__global__ void stream_counts_kernel(uint32_t nq, uint32_t P, const uint32_t *__restrict__ vbase,
                                     uint32_t *__restrict__ cand_cnt) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) cand_cnt[q] = vbase[(size_t)q * (P + 1) + P];
}

template <int MEASURE>
__global__ __launch_bounds__(256) void leaf_exact_scan_kernel(TxhIndexDev ix, ExactScanArgs a) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float qs_all[];   // [4 * a.qpt][dimp]: the tile's queries
    __shared__ uint32_t s_pq_all[kExactQuadsMax * 4], s_vb_all[kExactQuadsMax * 4], tile_sh;
    const uint32_t tid = threadIdx.x;
    const uint32_t dim = ix.dim, dimp = (dim + 3u) & ~3u, chunks = dim >> 3;
    const uint32_t total_tiles = a.counters[CNT_TOTAL_TILES];
    const bool vec = ((ix.stride & 3u) == 0) && ((reinterpret_cast<uintptr_t>(ix.rows) & 15u) == 0);
    const bool qvec = ((a.q_stride & 3u) == 0) && ((reinterpret_cast<uintptr_t>(a.queries) & 15u) == 0);
    for (;;) {
        __syncthreads();   // the previous tile's LDS reads are done
        if (tid == 0) tile_sh = grab_tile(a.counters + CNT_XQ, total_tiles);
        __syncthreads();
        const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_sh);
        if (tile == kInvalid) break;
        uint32_t lo = 0, hi = ix.L;   // leaf = largest l with tile_off[l] <= tile
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (uniform_load(a.tile_off + mid) <= tile) lo = mid; else hi = mid;
        }
        const uint32_t leaf = lo;
        const uint32_t lb = uniform_load(ix.leaf_off + leaf);
        const uint32_t size = uniform_load(ix.leaf_off + leaf + 1) - lb;
        const uint32_t nchunks = (size + kExactRows - 1) / kExactRows;
        const uint32_t local = tile - uniform_load(a.tile_off + leaf);
        const uint32_t chunk = local % nchunks, qg = local / nchunks;
        const uint32_t slot0 = uniform_load(a.pair_off + leaf);
        const uint32_t slot_end = uniform_load(a.pair_off + leaf + 1);
        const uint32_t s_begin = slot0 + qg * a.qpt * 4u;
        const uint32_t s_stop = min(s_begin + a.qpt * 4u, slot_end);
        const uint32_t j = chunk * kExactRows + tid;
        const bool valid = j < size;
        const uint32_t csr = lb + (valid ? j : 0u);
        const float *row = ix.rows + (size_t)(ix.rows_csr ? csr : ix.leaf_ids[csr]) * ix.stride;

        // stage every query of the tile once (the passes below then run without a barrier)
        const uint32_t nslots = s_stop - s_begin, nslots8 = (nslots + 7u) & ~7u;
        if (tid < nslots8) {
            const uint32_t sl = s_begin + tid;
            s_pq_all[tid] = sl < s_stop ? a.pair_q[sl] : kInvalid;
            s_vb_all[tid] = sl < s_stop ? a.pair_vbase[sl] : 0u;
        }
        __syncthreads();   // s_pq_all visible
        if (qvec) {
            // 16-byte copies, four independent loads in flight per thread (an element-at-a-time loop
            // serialised two dependent global loads per element: ~50 us per tile)
            const uint32_t dim4 = dimp >> 2, total4 = nslots8 * dim4;
            for (uint32_t i0 = 0; i0 < total4; i0 += 1024u) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = i0 + 256u * u + tid;
                    v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (i < total4) {
                        const uint32_t qi = i / dim4, j4 = (i - qi * dim4) * 4u;
                        const uint32_t pq = s_pq_all[qi];
                        if (pq != kInvalid) {
                            v[u] = *reinterpret_cast<const float4 *>(a.queries + (size_t)pq * a.q_stride + j4);
                            if (j4 + 1 >= dim) v[u].y = 0.0f;   // padding of the last piece
                            if (j4 + 2 >= dim) v[u].z = 0.0f;
                            if (j4 + 3 >= dim) v[u].w = 0.0f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = i0 + 256u * u + tid;
                    if (i < total4) reinterpret_cast<float4 *>(qs_all)[i] = v[u];
                }
            }
        } else {
            for (uint32_t i = tid; i < nslots8 * dimp; i += 256) {
                const uint32_t qi = i / dimp, jj = i - qi * dimp;
                const uint32_t pq = s_pq_all[qi];
                qs_all[i] = (pq != kInvalid && jj < dim) ? a.queries[(size_t)pq * a.q_stride + jj] : 0.0f;
            }
        }
        __syncthreads();

        for (uint32_t sg = 0; sg < nslots; sg += kExactQT) {
            const float *qs = qs_all + sg * dimp;
            const uint32_t *s_pq = s_pq_all + sg, *s_vb = s_vb_all + sg;
            f32x2 accv[kExactQT][4];
            // Cosine (one_to_one.rs:559-604) also sums a.a and b.b lane by lane; the query's chains are
            // recomputed per row like the reference does (same operations, same value every time)
            f32x2 aav[MEASURE == SCANN_HIP_COSINE ? kExactQT : 1][4];
            f32x2 bbv[4] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
#pragma unroll
            for (int qi = 0; qi < kExactQT; ++qi)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    accv[qi][u] = f32x2{0.0f, 0.0f};
                    if (MEASURE == SCANN_HIP_COSINE) aav[MEASURE == SCANN_HIP_COSINE ? qi : 0][u] = f32x2{0.0f, 0.0f};
                }
            // the row streams through registers four 8-dim chunks at a time, the next group in
            // flight while this one computes (a chunk-at-a-time loop exposed one global-load latency
            // per 8 dims)
            constexpr int G = 4;
            float4 xa[G], xb[G], na[G], nb[G];
            auto load_group = [&](uint32_t c0, float4 *pa, float4 *pb) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t c = c0 + g;
                    if (c < chunks) {
                        if (vec) {
                            pa[g] = *reinterpret_cast<const float4 *>(row + 8 * c);
                            pb[g] = *reinterpret_cast<const float4 *>(row + 8 * c + 4);
                        } else {
                            pa[g] = make_float4(row[8 * c], row[8 * c + 1], row[8 * c + 2], row[8 * c + 3]);
                            pb[g] = make_float4(row[8 * c + 4], row[8 * c + 5], row[8 * c + 6], row[8 * c + 7]);
                        }
                    }
                }
            };
            load_group(0, xa, xb);
            for (uint32_t c0 = 0; c0 < chunks; c0 += G) {
                if (c0 + G < chunks) load_group(c0 + G, na, nb);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t c = c0 + g;
                    if (c >= chunks) break;
                    const f32x2 x[4] = {f32x2{xa[g].x, xa[g].y}, f32x2{xa[g].z, xa[g].w}, f32x2{xb[g].x, xb[g].y},
                                        f32x2{xb[g].z, xb[g].w}};
#pragma unroll
                    for (int qi = 0; qi < kExactQT; ++qi) {
                        const float4 qa = *reinterpret_cast<const float4 *>(qs + qi * dimp + 8 * c);
                        const float4 qb = *reinterpret_cast<const float4 *>(qs + qi * dimp + 8 * c + 4);
                        const f32x2 qv[4] = {f32x2{qa.x, qa.y}, f32x2{qa.z, qa.w}, f32x2{qb.x, qb.y},
                                             f32x2{qb.z, qb.w}};
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (MEASURE == SCANN_HIP_DOT_PRODUCT) {
                                accv[qi][u] = __builtin_elementwise_fma(qv[u], x[u], accv[qi][u]);
                            } else if (MEASURE == SCANN_HIP_L1) {       // l1_distance_avx2: add |a - b|, no FMA
                                accv[qi][u] = accv[qi][u] + __builtin_elementwise_abs(qv[u] - x[u]);
                            } else if (MEASURE == SCANN_HIP_COSINE) {   // add(mul): two roundings
                                accv[qi][u] = accv[qi][u] + qv[u] * x[u];
                                if (qi == 0) bbv[u] = bbv[u] + x[u] * x[u];
                                aav[MEASURE == SCANN_HIP_COSINE ? qi : 0][u] = aav[MEASURE == SCANN_HIP_COSINE ? qi : 0][u] + qv[u] * qv[u];
                            } else {
                                const f32x2 d = qv[u] - x[u];
                                accv[qi][u] = __builtin_elementwise_fma(d, d, accv[qi][u]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    xa[g] = na[g];
                    xb[g] = nb[g];
                }
            }
#pragma unroll
            for (int qi = 0; qi < kExactQT; ++qi) {
                const uint32_t pq = s_pq[qi];
                if (pq == kInvalid) continue;   // block-uniform
                // hsum tree (x86.rs:31-44): lanes (0+4, 1+5), (2+6, 3+7) -> (s0+s1) + (s2+s3)
                const f32x2 s01 = accv[qi][0] + accv[qi][2], s23 = accv[qi][1] + accv[qi][3];
                float r = (s01.x + s01.y) + (s23.x + s23.y);
                float saa = 0.0f, sbb = 0.0f;
                if (MEASURE == SCANN_HIP_COSINE) {   // wide 0.7 reduce_add (non-AVX build): ((v0+v1)+v2)+v3 per half
                    const f32x2 *av = aav[MEASURE == SCANN_HIP_COSINE ? qi : 0];
                    r = (((accv[qi][0].x + accv[qi][0].y) + accv[qi][1].x) + accv[qi][1].y) +
                        (((accv[qi][2].x + accv[qi][2].y) + accv[qi][3].x) + accv[qi][3].y);
                    saa = (((av[0].x + av[0].y) + av[1].x) + av[1].y) + (((av[2].x + av[2].y) + av[3].x) + av[3].y);
                    sbb = (((bbv[0].x + bbv[0].y) + bbv[1].x) + bbv[1].y) + (((bbv[2].x + bbv[2].y) + bbv[3].x) + bbv[3].y);
                }
                for (uint32_t jj = chunks * 8; jj < dim; ++jj) {   // scalar tail, not fused
                    const float qv = qs[qi * dimp + jj];
                    if (MEASURE == SCANN_HIP_DOT_PRODUCT) {
                        r = r + qv * row[jj];
                    } else if (MEASURE == SCANN_HIP_L1) {
                        r = r + fabsf(qv - row[jj]);
                    } else if (MEASURE == SCANN_HIP_COSINE) {
                        r = r + qv * row[jj];
                        saa = saa + qv * qv;
                        sbb = sbb + row[jj] * row[jj];
                    } else {
                        const float d = qv - row[jj];
                        r = r + d * d;
                    }
                }
                float dist = r;
                if (MEASURE == SCANN_HIP_DOT_PRODUCT) dist = -r;
                if (MEASURE == SCANN_HIP_L2) dist = sqrtf(r);
                if (MEASURE == SCANN_HIP_COSINE) {   // one_to_one.rs:596-612
                    const float na = sqrtf(saa), nb = sqrtf(sbb);
                    dist = 1.0f - ((na == 0.0f || nb == 0.0f) ? 0.0f : r / (na * nb));
                }
                if (valid) {
                    const uint32_t vpos = s_vb[qi] + j;
                    if (vpos < a.cap) a.cand[(size_t)pq * a.cap + vpos] = make_key(dist, vpos);
                }
            }
        }
    }
}

// =====================================================================================
// K7: select + re-rank.  mod.rs:283-293 and :342-364 for one query per block:
//   candidates -> exact m best keys (sorted) -> decode -> exact SquaredL2 with the
//   reference's AVX2 arithmetic (simd/x86.rs:139-165: 8 FMA lane chains, fixed hsum
//   tree, scalar tail) -> stable sort by exact -> first k.
// =====================================================================================

struct SelectArgs {
    uint32_t P, m, k, cap;
    uint32_t lds_keys;   // capacity of the LDS key array (<= kSortCap)
    uint32_t direct;     // unsorted selection straight from the global list (no LDS staging); sel_n sizes its scratch
    uint32_t sel_n;
    int exact_reorder, local_only;
    int unsorted;   // candidates may leave in any order (final stage orders by 96-bit keys)
    const float *queries;
    uint32_t q_stride;
    const uint32_t *tokens, *vbase;
    const uint64_t *thr;
    uint32_t *cand_cnt;
    uint64_t *cand;
    uint32_t *counters;
    uint64_t *cand_key;
    uint32_t *cand_idx;
    float *cand_dist, *cand_exact;
    uint32_t *cand_row;
    uint32_t *cand_count;
    uint32_t *out_idx;
    float *out_dist;
    uint32_t *out_count;
};

// Stable compaction of list[0..cnt) keeping keys <= T into dst (dst == list: in place), by one block.
__device__ static uint32_t block_compact_le(const uint64_t *list, uint64_t *dst, uint32_t cnt, uint64_t T,
                                            uint32_t *s_wave, uint32_t *s_base) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, nw = nt >> 6;
    if (tid == 0) *s_base = 0;
    __syncthreads();
    for (uint32_t b = 0; b < cnt; b += nt) {
        const uint32_t i = b + tid;
        uint64_t key = 0;
        bool keep = false;
        if (i < cnt) {
            key = list[i];
            keep = key <= T;
        }
        uint32_t wtot;
        const uint32_t wpre = wave_prefix_count(keep, &wtot);
        if ((tid & 63u) == 0) s_wave[wave] = wtot;
        __syncthreads();
        uint32_t off = *s_base;
        for (uint32_t w = 0; w < wave; ++w) off += s_wave[w];
        if (keep) dst[off + wpre] = key;   // off + wpre <= i: never overtakes unread data
        __syncthreads();
        if (tid == 0) {
            uint32_t t = 0;
            for (uint32_t w = 0; w < nw; ++w) t += s_wave[w];
            *s_base += t;
        }
        __syncthreads();
    }
    return *s_base;
}

__device__ static void select_fail(const SelectArgs &a, uint32_t q, uint32_t status) {
    if (threadIdx.x == 0) {
        atomicMax(&a.counters[CNT_STATUS], status);
        a.cand_count[q] = 0;
        if (!a.local_only) a.out_count[q] = 0;
    }
}

__global__ __launch_bounds__(kSelectThreads) void select_rerank_kernel(TxhIndexDev ix,
                                                                       SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];      // [a.lds_keys]
    const uint32_t sort_cap = a.lds_keys;
    const SelCfg cfg = sel_cfg(a.direct ? a.sel_n : sort_cap);
    uint32_t *s_dvb = reinterpret_cast<uint32_t *>(skeys + sort_cap) + (kSelectThreads / 64 + 4) + cfg.bins +
                      2 * cfg.list + 96;                                  // [kDecodeStage]
    uint32_t *s_drow = s_dvb + kDecodeStage;                              // [kDecodeStage]
    uint32_t *s_wave = reinterpret_cast<uint32_t *>(skeys + sort_cap);    // [kSelectThreads/64]
    uint32_t *s_basep = s_wave + kSelectThreads / 64;                     // [4]
    const uint32_t q = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t m = a.m, k = a.k;

    uint32_t cnt = a.cand_cnt[q];
    if (cnt > a.cap) {  // candidate buffer overflow: report, never return a wrong row
        select_fail(a, q, (uint32_t)SCANN_HIP_RESOURCE_EXHAUSTED);
        return;
    }
    // A statistical threshold (sample rank j < m) must be verified: with >= m survivors
    // the exact top-m is among them; with fewer the threshold was too tight.
    if (a.thr[q] != SCANN_KEY_MAX && cnt < m) {
        select_fail(a, q, (uint32_t)SCANN_HIP_ABORTED);
        return;
    }
    uint64_t *list = a.cand + (size_t)q * a.cap;

    uint32_t *hist = reinterpret_cast<uint32_t *>(s_basep + 4);
    uint64_t *slist = reinterpret_cast<uint64_t *>(hist + cfg.bins);
    uint64_t *sred = slist + cfg.list;
    bool in_lds = false;   // skeys[0..cnt) already holds the candidates
    if (cnt > sort_cap && !a.direct) {
        // More candidates than LDS holds (no-threshold retry, dense lists of the exact leaf scan):
        // rank-select the m-th smallest key straight from the global list, then keep the keys <= it
        // (keys are unique, so exactly m remain; m <= kMaxPreReorderK <= sort_cap).
        const uint64_t T = block_select<uint64_t>(list, cnt, m, cfg, hist, slist, sred);
        __syncthreads();
        cnt = block_compact_le(list, skeys, cnt, T, s_wave, s_basep);
        __syncthreads();
        in_lds = true;
    }
    if (cnt > sort_cap && !a.direct) {   // cannot happen: m <= sort_cap
        select_fail(a, q, (uint32_t)SCANN_HIP_RESOURCE_EXHAUSTED);
        return;
    }

    const uint32_t nsel = min(m, cnt);   // truncate(pre_reorder_k)  mod.rs:290
    // decode tables of this query (key base and first CSR row of each selected leaf) in LDS:
    // the per-candidate chain vbase -> token -> leaf_off -> leaf_ids is otherwise four
    // dependent global loads
    const uint32_t *vb = a.vbase + (size_t)q * (a.P + 1);
    const bool staged = a.P <= kDecodeStage;
    if (staged) {
        for (uint32_t r = tid; r < a.P; r += nt) {
            s_dvb[r] = vb[r];
            s_drow[r] = ix.leaf_off[a.tokens[(size_t)q * a.P + r]];
        }
        __syncthreads();
    }
    auto decode = [&](uint64_t key, uint32_t i) {
        // merge key -> (rank, position in leaf) -> CSR row -> datapoint index
        const uint32_t vpos = (uint32_t)key;
        uint32_t lo = 0, hi = a.P;
        uint32_t csr;
        if (staged) {
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (s_dvb[mid] <= vpos) lo = mid; else hi = mid;
            }
            csr = s_drow[lo] + (vpos - s_dvb[lo]);
        } else {
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (vb[mid] <= vpos) lo = mid; else hi = mid;
            }
            csr = ix.leaf_off[a.tokens[(size_t)q * a.P + lo]] + (vpos - vb[lo]);
        }
        {
        const uint32_t idx = ix.leaf_ids ? ix.leaf_ids[csr] : csr;
        a.cand_row[(size_t)q * m + i] = ix.rows_csr ? csr : idx;
        a.cand_key[(size_t)q * m + i] = key;
        a.cand_idx[(size_t)q * m + i] = idx;
        a.cand_dist[(size_t)q * m + i] = ordered_to_f32((uint32_t)(key >> 32));
        }
    };

    if (a.unsorted) {
        // Selection without a sort: the m-th smallest key by histogram select, keep keys <= it.
        // The final stage orders by (exact, merge key), which equals (exact, approx rank).
        // direct: the keys stay in the (L2-hot) global list -- staging ~10 k keys takes 80-128 KB of LDS, ONE
        // workgroup per compute unit; without it every query's workgroup is resident at once
        const uint64_t *src = a.direct ? list : skeys;
        if (!in_lds && !a.direct)
            for (uint32_t i = tid; i < cnt; i += nt) skeys[i] = list[i];
        __syncthreads();
        const uint64_t T = cnt > m ? block_select<uint64_t>(src, cnt, m, cfg, hist, slist, sred) : SCANN_KEY_MAX;
        __syncthreads();
        uint32_t *s_slot = reinterpret_cast<uint32_t *>(sred);   // output cursor
        if (tid == 0) *s_slot = 0;
        __syncthreads();
        for (uint32_t b = 0; b < cnt; b += nt) {   // wave-aggregated slot allocation, any order
            const uint32_t i = b + tid;
            uint64_t key = 0;
            bool keep = false;
            if (i < cnt) {
                key = src[i];
                keep = key <= T;
            }
            uint32_t wtot;
            const uint32_t wpre = wave_prefix_count(keep, &wtot);
            uint32_t base = 0;
            if ((tid & 63u) == 0 && wtot) base = atomicAdd(s_slot, wtot);
            base = (uint32_t)__shfl((int)base, 0);
            if (keep) decode(key, base + wpre);
        }
        if (tid == 0) a.cand_count[q] = nsel;
        return;
    }

    if (!in_lds)
        for (uint32_t i = tid; i < cnt; i += nt) skeys[i] = list[i];
    __syncthreads();
    if (cnt > 2 * m && cnt > 256) {   // sort only the m best: select the m-th key, compact in place
        const uint64_t T = block_select<uint64_t>(skeys, cnt, m, cfg, hist, slist, sred);
        __syncthreads();
        cnt = block_compact_le(skeys, skeys, cnt, T, s_wave, s_basep);
        __syncthreads();
    }
    uint32_t n2 = 1;
    while (n2 < cnt) n2 <<= 1;
    for (uint32_t i = cnt + tid; i < n2; i += nt) skeys[i] = SCANN_KEY_MAX;
    __syncthreads();
    bitonic_sort_lds(skeys, n2);

    for (uint32_t i = tid; i < nsel; i += nt) decode(skeys[i], i);
    if (tid == 0) a.cand_count[q] = nsel;

    if (!a.exact_reorder) {  // AsymmetricHasher::search: k best by approximate distance
        if (!a.local_only) {
            __syncthreads();
            const uint32_t nout = min(k, nsel);
            for (uint32_t i = tid; i < k; i += nt) {
                a.out_idx[(size_t)q * k + i] = (i < nout) ? a.cand_idx[(size_t)q * m + i] : kInvalid;
                a.out_dist[(size_t)q * k + i] =
                    (i < nout) ? ordered_to_f32((uint32_t)(skeys[i] >> 32)) : __builtin_inff();
            }
            if (tid == 0) a.out_count[q] = nout;
        }
        return;
    }

    // exact re-rank and the final sort run in rerank_kernel / final_sort_kernel
}

// =====================================================================================
// K8: exact re-rank.  tree_x_hybrid/mod.rs:350-358 -> squared_l2_avx2
// (simd/x86.rs:139-165): 8 lanes per candidate are the 8 AVX2 FMA lane chains (chunk
// order), combined by the fixed hsum tree (x86.rs:31-44) with DPP shuffles, plus the
// non-fused scalar tail.  32 candidates per 256-thread block, grid over (candidates,
// queries): the random row gathers are spread over the whole chip.
// =====================================================================================
// Exact distance of one (query, row) pair by a group of 8 lanes (lane8 = the AVX2 lane chain), with the
// measure of the re-ordering (utils/reordering.rs:35-44): squared_l2_avx2 / dot_product_avx2 /
// l1_distance_avx2 (simd/x86.rs) or the cosine of one_to_one.rs:559-612.  The result is valid in the
// group's lane 0 (act must be uniform over the group).
template <int U = 8>   // independent loads in flight per lane (U * 8 dims per memory round trip)
__device__ __forceinline__ float exact_pair_8lanes(const TxhIndexDev &ix, const float *s_q, const float *row, bool act,
                                                  uint32_t lane8) {
    const uint32_t dim = ix.dim, chunks = dim >> 3;
    float accv = 0.0f, aav = 0.0f, bbv = 0.0f;   // (aa / bb: Cosine's two extra lane chains)
    for (uint32_t i0 = 0; i0 < chunks; i0 += U) {
        float xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = (act && i0 + u < chunks) ? row[8 * (i0 + u) + lane8] : 0.0f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u < chunks) {
                const float qv = s_q[8 * (i0 + u) + lane8];
                if (ix.measure == SCANN_HIP_DOT_PRODUCT) {   // dot_product_avx2, x86.rs:72-96
                    accv = fmaf(qv, xv[u], accv);
                } else if (ix.measure == SCANN_HIP_L1) {     // l1_distance_avx2, x86.rs:103-132
                    accv = accv + fabsf(qv - xv[u]);
                } else if (ix.measure == SCANN_HIP_COSINE) { // cosine_similarity_f32_simd: add(mul), not fused
                    accv = accv + qv * xv[u];
                    aav = aav + qv * qv;
                    bbv = bbv + xv[u] * xv[u];
                } else {
                    const float diff = qv - xv[u];
                    accv = fmaf(diff, diff, accv);        // _mm256_fmadd_ps(diff, diff, sum)
                }
            }
        }
    }
    float r;
    float saa = 0.0f, sbb = 0.0f;
    if (ix.measure == SCANN_HIP_COSINE) {
        // wide 0.7 f32x8::reduce_add (non-AVX build): ((v0 + v1) + v2) + v3 per half, then lo + hi;
        // lane 0 of each half runs the chain with values shuffled in
        auto half_sum = [&](float v) {
            float h = v + __shfl_down(v, 1, 8);     // lanes 0, 4: v0 + v1
            h = h + __shfl_down(v, 2, 8);           // + v2
            h = h + __shfl_down(v, 3, 8);           // + v3
            return h + __shfl_down(h, 4, 8);        // lane 0: lo + hi
        };
        r = half_sum(accv);
        saa = half_sum(aav);
        sbb = half_sum(bbv);
    } else {
        // horizontal_sum_f32_avx2: (lo+hi) -> +movehdup -> +movehl
        const float s = accv + __shfl_down(accv, 4, 8);     // lanes 0..3: v[j] + v[j+4]
        const float t = s + __shfl_down(s, 1, 8);           // lane 0: s0+s1, lane 2: s2+s3
        r = t + __shfl_down(t, 2, 8);                       // lane 0: (s0+s1) + (s2+s3)
    }
    if (act && lane8 == 0) {
        for (uint32_t j = chunks * 8; j < dim; ++j) {   // scalar tail, not fused
            if (ix.measure == SCANN_HIP_DOT_PRODUCT) {
                r = r + s_q[j] * row[j];
            } else if (ix.measure == SCANN_HIP_L1) {
                r = r + fabsf(s_q[j] - row[j]);
            } else if (ix.measure == SCANN_HIP_COSINE) {
                r = r + s_q[j] * row[j];
                saa = saa + s_q[j] * s_q[j];
                sbb = sbb + row[j] * row[j];
            } else {
                const float diff = s_q[j] - row[j];
                r = r + diff * diff;
            }
        }
        // ReorderingHelper with the configured measure (utils/reordering.rs:35-44)
        if (ix.measure == SCANN_HIP_DOT_PRODUCT) r = -r;
        if (ix.measure == SCANN_HIP_L2) r = sqrtf(r);
        if (ix.measure == SCANN_HIP_COSINE) {
            const float na = sqrtf(saa), nb = sqrtf(sbb);
            r = 1.0f - ((na == 0.0f || nb == 0.0f) ? 0.0f : r / (na * nb));
        }
    }
    return r;
}

__global__ __launch_bounds__(256) void rerank_kernel(TxhIndexDev ix, const float *__restrict__ queries,
                                                     uint32_t q_stride, uint32_t m,
                                                     const uint32_t *__restrict__ cand_row,
                                                     const uint32_t *__restrict__ cand_count,
                                                     float *__restrict__ cand_exact) {
    extern __shared__ __attribute__((aligned(16))) float s_q[];   // [dim]
    const uint32_t q = blockIdx.y, tid = threadIdx.x;
    const uint32_t nsel = cand_count[q];
    const uint32_t c0 = blockIdx.x * 32u;
    if (c0 >= nsel) return;   // uniform
    const uint32_t dim = ix.dim;
    for (uint32_t j = tid; j < dim; j += blockDim.x) s_q[j] = queries[(size_t)q * q_stride + j];
    __syncthreads();
    const uint32_t lane8 = tid & 7u;
    const uint32_t c = c0 + (tid >> 3);
    const bool act = c < nsel;
    const float *row = ix.rows;
    if (act) row = ix.rows + (size_t)cand_row[(size_t)q * m + c] * ix.stride;
    const float r = exact_pair_8lanes(ix, s_q, row, act, lane8);
    if (act && lane8 == 0) {
        cand_exact[(size_t)q * m + c] = r;
    }
}

// K9: stable sort by exact distance (key = ordered(exact) << 32 | approx rank), first k.
// tree_x_hybrid/mod.rs:360-361.
__global__ __launch_bounds__(kSelectThreads) void final_sort_kernel(
    uint32_t m, uint32_t k, const uint32_t *__restrict__ cand_count,
    const uint32_t *__restrict__ cand_idx, const float *__restrict__ cand_exact,
    uint32_t *__restrict__ out_idx, float *__restrict__ out_dist, uint32_t *__restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t nsel = cand_count[q];
    uint32_t m2 = 1;
    while (m2 < nsel) m2 <<= 1;
    for (uint32_t i = tid; i < m2; i += nt)
        skeys[i] = (i < nsel) ? make_key(cand_exact[(size_t)q * m + i], i) : SCANN_KEY_MAX;
    __syncthreads();
    bitonic_sort_lds(skeys, m2);
    const uint32_t nout = min(k, nsel);   // truncate(k)
    for (uint32_t i = tid; i < k; i += nt) {
        uint32_t oi = kInvalid;
        float od = __builtin_inff();
        if (i < nout) {
            const uint64_t key = skeys[i];
            oi = cand_idx[(size_t)q * m + (uint32_t)key];
            od = ordered_to_f32((uint32_t)(key >> 32));
        }
        out_idx[(size_t)q * k + i] = oi;
        out_dist[(size_t)q * k + i] = od;
    }
    if (tid == 0) out_count[q] = nout;
}

// K9b: the same ordering without a sort, for small k and unsorted candidates: arg-min rounds
// over (ordered(exact), merge key) -- the merge key is monotone in the approximate rank, so
// this is exactly the stable sort's order.  Two levels without block-wide rounds: every wave
// extracts the k best of its share with wave shuffles only, then wave 0 extracts the k best
// of the (waves x k) finalists.
constexpr uint32_t kTopkMaxK = 64;

// One arg-min round over the per-lane best (b_eb, b_kk, b_sl) of a wave; returns the winner
// in every lane.  Exact-distance ties are rare: reduce the 32-bit distance first and fall
// back to the 96-bit reduction only when several lanes tie.
__device__ __forceinline__ void wave_argmin96(uint32_t &b_eb, uint64_t &b_kk, uint32_t &b_sl) {
    uint32_t mn = b_eb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, d));
    const unsigned long long tie = __ballot(b_eb == mn);
    if (__popcll(tie) == 1) {
        const int src = __ffsll((long long)tie) - 1;
        b_eb = mn;
        b_sl = (uint32_t)__shfl((int)b_sl, src);
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)b_kk, src);
        const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(b_kk >> 32), src);
        b_kk = ((uint64_t)hi << 32) | lo;
        return;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o_eb = (uint32_t)__shfl_xor((int)b_eb, d), o_sl = (uint32_t)__shfl_xor((int)b_sl, d);
        const uint64_t o_kk = shfl_xor_t<uint64_t>(b_kk, d);
        if (o_eb < b_eb || (o_eb == b_eb && o_kk < b_kk)) {
            b_eb = o_eb;
            b_kk = o_kk;
            b_sl = o_sl;
        }
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void final_topk_kernel(
    uint32_t m, uint32_t k, const uint32_t *__restrict__ cand_count,
    const uint32_t *__restrict__ cand_idx, const uint64_t *__restrict__ cand_key,
    const float *__restrict__ cand_exact, uint32_t *__restrict__ out_idx,
    float *__restrict__ out_dist, uint32_t *__restrict__ out_count) {
    constexpr int E = 8;                                  // candidates per thread: m <= 8 * NT
    constexpr int NWV = NT / 64;
    constexpr int E2 = NWV * kTopkMaxK / 64;              // finalists per lane of wave 0
    __shared__ uint32_t s_eb[NWV * kTopkMaxK];
    __shared__ uint64_t s_kk[NWV * kTopkMaxK];
    __shared__ uint32_t s_sl[NWV * kTopkMaxK];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nsel = cand_count[q];
    if (nsel & 0x80000000u) return;   // rerank_short_kernel already wrote this query's rows (block-uniform)
    const uint32_t nout = min(k, nsel);
    uint32_t eb[E];
    uint64_t kk[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t slot = e * (uint32_t)NT + tid;
        const bool alive = slot < nsel;
        eb[e] = alive ? f32_to_ordered(cand_exact[(size_t)q * m + slot]) : 0xFFFFFFFFu;
        kk[e] = alive ? cand_key[(size_t)q * m + slot] : SCANN_KEY_MAX;
    }
    // level 1: the wave's nout best
    for (uint32_t r = 0; r < nout; ++r) {
        uint32_t b_eb = 0xFFFFFFFFu, b_sl = 0xFFFFFFFFu;
        uint64_t b_kk = SCANN_KEY_MAX;
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((uint32_t)e * (uint32_t)NT < nsel && (eb[e] < b_eb || (eb[e] == b_eb && kk[e] < b_kk))) {
                b_eb = eb[e];
                b_kk = kk[e];
                b_sl = e * (uint32_t)NT + tid;
            }
        wave_argmin96(b_eb, b_kk, b_sl);
        if (b_sl != 0xFFFFFFFFu && (b_sl & ((uint32_t)NT - 1)) == tid) {
            const uint32_t we = b_sl / (uint32_t)NT;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((uint32_t)e == we) {
                    eb[e] = 0xFFFFFFFFu;
                    kk[e] = SCANN_KEY_MAX;
                }
        }
        if (lane == 0) {
            s_eb[wave * nout + r] = b_eb;
            s_kk[wave * nout + r] = b_kk;
            s_sl[wave * nout + r] = b_sl;
        }
    }
    __syncthreads();
    // level 2: wave 0 takes the nout best of the NWV * nout finalists
    if (wave == 0) {
        const uint32_t nfin = NWV * nout;
        uint32_t feb[E2], fsl[E2];
        uint64_t fkk[E2];
#pragma unroll
        for (int e = 0; e < E2; ++e) {
            const uint32_t i = e * 64 + lane;
            const bool alive = i < nfin;
            feb[e] = alive ? s_eb[i] : 0xFFFFFFFFu;
            fkk[e] = alive ? s_kk[i] : SCANN_KEY_MAX;
            fsl[e] = alive ? s_sl[i] : 0xFFFFFFFFu;
        }
        for (uint32_t r = 0; r < nout; ++r) {
            uint32_t b_eb = 0xFFFFFFFFu, b_sl = 0xFFFFFFFFu;
            uint64_t b_kk = SCANN_KEY_MAX;
#pragma unroll
            for (int e = 0; e < E2; ++e)
                if ((uint32_t)e * 64 < nfin && (feb[e] < b_eb || (feb[e] == b_eb && fkk[e] < b_kk))) {
                    b_eb = feb[e];
                    b_kk = fkk[e];
                    b_sl = fsl[e];
                }
            wave_argmin96(b_eb, b_kk, b_sl);
#pragma unroll
            for (int e = 0; e < E2; ++e)
                if (fsl[e] == b_sl) {   // slots are unique: exactly one lane/element matches
                    feb[e] = 0xFFFFFFFFu;
                    fkk[e] = SCANN_KEY_MAX;
                    fsl[e] = 0xFFFFFFFEu;
                }
            if (lane == 0) {
                out_idx[(size_t)q * k + r] = cand_idx[(size_t)q * m + b_sl];
                out_dist[(size_t)q * k + r] = ordered_to_f32(b_eb);
            }
        }
    }
    for (uint32_t i = nout + tid; i < k; i += blockDim.x) {
        out_idx[(size_t)q * k + i] = kInvalid;
        out_dist[(size_t)q * k + i] = __builtin_inff();
    }
    if (tid == 0) out_count[q] = nout;
}

// =====================================================================================
// K8b: exact re-rank behind an int8 row filter (low-precision row stores as
// shortlist filters; the reference's int8 rows: distance_measures/one_to_many_asymmetric.rs:25-377,
// brute_force/scalar_quantized.rs:168-260).
//
// reorder_results (tree_x_hybrid/mod.rs:342-364) scores ALL m candidates exactly and keeps k of
// them; at m = 5000 that is 2.6 GB of random 512-byte row gathers per 1024 queries, as long as the
// scan.  Here every row also exists as int8 (per-row scale s = max|x| / 127, q = round(x / s),
// x~ = s q) with its quantisation error E = ||x - x~|| stored beside it.  For a query q:
//     d~ = ||q - x~||^2,   | ||q - x||^2 - d~ |  <=  2 sqrt(d~) E + E^2
// so [L, U] = d~ -+ (2 sqrt(d~) E + E^2 + slack) brackets the reference's f32 distance (slack covers
// the f32 rounding of both sums).  With tau = the k-th smallest U, at least k candidates lie at or
// under tau, so a candidate with L > tau can neither enter the top k nor tie with it: only the
// shortlist {L <= tau} (tens of rows) is scored with the reference's arithmetic, and the result --
// indices, distances, tie order -- is the one the full re-rank gives.
//   rerank_i8_kernel     L, U of every candidate (128-byte int8 rows instead of 512-byte f32 rows)
//   rerank_short_kernel  block per query: tau, shortlist, exact distances (rerank_kernel's
//                        arithmetic), the k best by (exact, merge key), output rows
// =====================================================================================
// uni_scale > 0: every row with that one scale (the store of rerank_i8_kernel's UNIFORM form, see the launcher)
__global__ __launch_bounds__(256) void rows_i8_build_kernel(const float *__restrict__ rows, uint64_t n, uint32_t dim,
                                                            uint32_t stride, int8_t *__restrict__ rows8,
                                                            float2 *__restrict__ meta, float uni_scale) {
    // 8 lanes per row
    const uint64_t r = (uint64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const uint32_t l8 = threadIdx.x & 7u;
    const bool act = r < n;
    const float *row = rows + (act ? r : 0) * stride;
    float mx = 0.0f;
    for (uint32_t j = l8; j < dim; j += 8) mx = fmaxf(mx, fabsf(row[j]));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 8));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 8));
    mx = fmaxf(mx, __shfl_xor(mx, 4, 8));
    const bool finite = mx < __builtin_inff();      // (NaN rows: mx stays finite-or-NaN; the error below turns NaN)
    const float sc = uni_scale > 0.0f ? uni_scale : (mx > 0.0f && finite) ? mx / 127.0f : 1.0f;
    float err = 0.0f;
    for (uint32_t j = l8; j < dim; j += 8) {
        const float x = row[j];
        float t = rintf(x / sc);
        t = fminf(fmaxf(t, -127.0f), 127.0f);
        if (act) rows8[r * dim + j] = (int8_t)(t == t ? (int)t : 0);
        const float e = x - sc * (t == t ? t : 0.0f);
        err = err + e * e;
    }
    err += __shfl_xor(err, 1, 8);
    err += __shfl_xor(err, 2, 8);
    err += __shfl_xor(err, 4, 8);
    if (act && l8 == 0) {
        // E rounded up a little (the sum above is f32); a NaN / infinite row gets E = +inf: never filtered out
        float E = sqrtf(err) * 1.0001f + 1e-30f;
        if (!(E == E) || !finite) E = __builtin_inff();
        meta[r] = make_float2(sc, E);
    }
}

// ---- the reference's FP8 codec (quantization/fp8.rs:80-203), bit for bit -----------------------------
// format 0 = E4M3 (bias 7, 3 mantissa bits, max code 0x7E), 1 = E5M2 (bias 15, 2 bits, max code 0x7C).
// NOT the hardware conversion: the mantissa carry wraps without bumping the exponent, the top exponent
// field only ever encodes the maximum, values under the smallest normal flush to (signed) zero.
__device__ __forceinline__ uint32_t fp8_from_f32(float value, int format) {
    const int mbits = format ? 2 : 3, bias = format ? 15 : 7, emax = format ? 31 : 15;
    const uint32_t maxcode = format ? 0x7Cu : 0x7Eu;
    if (value == 0.0f) return 0u;
    const uint32_t bits = __float_as_uint(value);
    const uint32_t sign = bits >> 31;
    const int exp = (int)((bits >> 23) & 0xFFu);
    const uint32_t mantissa = bits & 0x7FFFFFu;
    if (exp == 0xFF) return (sign << 7) | maxcode;
    const int e8 = exp - 127 + bias;
    if (e8 <= 0) return sign << 7;
    if (e8 >= emax) return (sign << 7) | maxcode;
    const uint32_t m = ((mantissa >> (23 - mbits)) + ((mantissa >> (22 - mbits)) & 1u)) & ((1u << mbits) - 1u);
    return (sign << 7) | ((uint32_t)e8 << mbits) | m;
}

__device__ __forceinline__ float fp8_to_f32(uint32_t b, int format) {
    const int mbits = format ? 2 : 3, bias = format ? 15 : 7;
    const uint32_t sign = (b >> 7) & 1u;
    const int exp = (int)((b >> mbits) & (format ? 0x1Fu : 0xFu));
    const uint32_t mantissa = b & ((1u << mbits) - 1u);
    if (exp == 0 && mantissa == 0) return sign ? -0.0f : 0.0f;
    const int e32 = exp == 0 ? 126 - bias : exp - bias + 127;
    return __uint_as_float((sign << 31) | ((uint32_t)e32 << 23) | (mantissa << (23 - mbits)));
}

// Quantizer::quantize / dequantize over Fp8Quantizer (fp8.rs:247-268)
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const float *__restrict__ values, uint64_t n, float scale,
                                                           int format, uint8_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        out[i] = (uint8_t)fp8_from_f32(values[i] * scale, format);
}

__global__ __launch_bounds__(256) void fp8_dequantize_kernel(const uint8_t *__restrict__ bits, uint64_t n, float scale,
                                                             int format, float *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        out[i] = fp8_to_f32(bits[i], format) / scale;
}

// one_to_many_fp8_float_{squared_l2, dot_product} (distance_measures/one_to_many_asymmetric.rs:327-377):
// E4M3 rows, one sequential f32 sum per row (no FMA), the dot product negated.
__global__ __launch_bounds__(256) void fp8_one_to_many_kernel(const float *__restrict__ query, uint32_t dim,
                                                              const uint8_t *__restrict__ db, uint64_t stride,
                                                              uint64_t n, int dot, float *__restrict__ out) {
    extern __shared__ float s_q8[];   // [dim]
    for (uint32_t j = threadIdx.x; j < dim; j += 256) s_q8[j] = query[j];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint8_t *row = db + i * stride;
        float sum = 0.0f;
        if (dot) {
            for (uint32_t j = 0; j < dim; ++j) sum = sum + s_q8[j] * fp8_to_f32(row[j], 0);
            out[i] = -sum;
        } else {
            for (uint32_t j = 0; j < dim; ++j) {
                const float diff = s_q8[j] - fp8_to_f32(row[j], 0);
                sum = sum + diff * diff;
            }
            out[i] = sum;
        }
    }
}

// The FP8 row store of the re-rank filter: the reference's E4M3 codec with Fp8Quantizer::calibrate_scale
// per row (scale = 448 / max|x|, fp8.rs:238-244); the filter decodes with v_cvt_pk_f32_fp8 (gfx950: OCP
// E4M3 -- equal to the reference's decode on every code its encoder emits) as x~ = dec * (1 / scale), and E
// is computed from that same expression.  `mismatch` counts codes the hardware decodes differently
// (never, unless the conversion instruction means another format: the create call then fails).
__global__ __launch_bounds__(256) void rows_fp8_build_kernel(const float *__restrict__ rows, uint64_t n, uint32_t dim,
                                                             uint32_t stride, uint8_t *__restrict__ rows8,
                                                             float2 *__restrict__ meta, uint32_t *__restrict__ mismatch) {
    const uint64_t r = (uint64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const uint32_t l8 = threadIdx.x & 7u;
    const bool act = r < n;
    const float *row = rows + (act ? r : 0) * stride;
    float mx = 0.0f;
    for (uint32_t j = l8; j < dim; j += 8) mx = fmaxf(mx, fabsf(row[j]));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 8));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 8));
    mx = fmaxf(mx, __shfl_xor(mx, 4, 8));
    const bool finite = mx < __builtin_inff();
    const float scale = 448.0f / fmaxf(finite ? mx : 1.0f, 1e-10f);
    const float inv = 1.0f / scale;
    float err = 0.0f;
    uint32_t bad = 0;
    for (uint32_t j = l8; j < dim; j += 8) {
        const float x = row[j];
        const uint32_t b = fp8_from_f32(x * scale, 0);
        const float hw = __builtin_amdgcn_cvt_f32_fp8((int)b, 0);
        bad += (__float_as_uint(hw) != __float_as_uint(fp8_to_f32(b, 0))) ? 1u : 0u;
        if (act) rows8[r * dim + j] = (uint8_t)b;
        const float e = x - hw * inv;
        err = err + e * e;
    }
    err += __shfl_xor(err, 1, 8);
    err += __shfl_xor(err, 2, 8);
    err += __shfl_xor(err, 4, 8);
    if (act && bad) atomicAdd(mismatch, bad);
    if (act && l8 == 0) {
        float E = sqrtf(err) * 1.0001f + 1e-30f;
        if (!(E == E) || !finite) E = __builtin_inff();
        meta[r] = make_float2(inv, E);
    }
}

struct I8RerankArgs {
    const int8_t *rows8;      // [n_rows][dim] int8, or the reference's E4M3 codes (FMT = 1)
    const float2 *meta;       // [n_rows] {dequantisation factor, error norm}
    float uni_scale, uni_E;   // UNI form: one dequantisation factor and one error bound for every row (no meta gather)
    const float *queries;
    uint32_t q_stride, m;
    const uint32_t *cand_row, *cand_count;
    uint32_t *lb, *ub;        // [nq][m] ordered(L), ordered(U)
};

constexpr uint32_t kI8PerBlock = 256;   // candidates per block (8 lanes each, 8 rounds): amortises the query staging

template <int FMT, bool UNI = false>   // 0 = int8 rows, 1 = FP8 (E4M3) rows
__global__ __launch_bounds__(256) void rerank_i8_kernel(uint32_t dim, I8RerankArgs a) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float s_q[];   // [dim]
    const uint32_t q = blockIdx.y, tid = threadIdx.x;
    const uint32_t nsel = a.cand_count[q];
    const uint32_t c0 = blockIdx.x * kI8PerBlock;
    if (c0 >= nsel) return;   // uniform
    for (uint32_t j = tid; j < dim; j += 256) s_q[j] = a.queries[(size_t)q * a.q_stride + j];
    __syncthreads();
    const uint32_t l8 = tid & 7u;
    constexpr int R = kI8PerBlock / 32;
    uint32_t row[R];
#pragma unroll
    for (int it = 0; it < R; ++it) {   // the rounds' row ids first: their gathers then overlap
        const uint32_t c = c0 + (uint32_t)it * 32u + (tid >> 3);
        row[it] = c < nsel ? a.cand_row[(size_t)q * a.m + c] : 0u;
    }
#pragma unroll
    for (int it = 0; it < R; ++it) {
        const uint32_t c = c0 + (uint32_t)it * 32u + (tid >> 3);
        const bool act = c < nsel;
        const float2 me = UNI ? make_float2(a.uni_scale, a.uni_E) : a.meta[row[it]];
        const int8_t *r8 = a.rows8 + (size_t)row[it] * dim;
        float acc = 0.0f;
        for (uint32_t j0 = l8 * 16u; j0 < dim; j0 += 128u) {   // 16 dims per lane per pass (dim % 16 == 0)
            const uint4 v = *reinterpret_cast<const uint4 *>(r8 + j0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            if constexpr (FMT == 1) {
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    const v2f lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[wi], false);
                    const v2f hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[wi], true);
                    const float xs[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float d = s_q[j0 + wi * 4 + i] - xs[i] * me.x;
                        acc = fmaf(d, d, acc);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float x = me.x * (float)(int)(int8_t)(w[i >> 2] >> (8 * (i & 3)));
                    const float d = s_q[j0 + i] - x;
                    acc = fmaf(d, d, acc);
                }
            }
        }
        acc += __shfl_xor(acc, 1, 8);
        acc += __shfl_xor(acc, 2, 8);
        acc += __shfl_xor(acc, 4, 8);
        if (act && l8 == 0) {
            const float E = me.y;
            // | d_f32 - d~ | <= 2 sqrt(d~) E + E^2 (triangle inequality on the real values) + the f32 rounding
            // of the two dim-term sums (each <= (dim + 4) 2^-24 relative) + slack on the bound itself
            const float slack = (2.0f * sqrtf(acc) * E + E * E) * 1.0001f + acc * ((float)(dim + 8) * 1.2e-7f) + 1e-30f;
            float L = acc - slack, U = acc + slack;
            if (!(slack == slack) || !(acc == acc)) {   // NaN anywhere: never filtered out, never a bound for others
                L = -__builtin_inff();
                U = __builtin_inff();
            }
            a.lb[(size_t)q * a.m + c] = f32_to_ordered(L);
            a.ub[(size_t)q * a.m + c] = f32_to_ordered(U);
        }
    }
}

constexpr uint32_t kShortMaxFast = 1024;   // shortlists up to this size finish inside rerank_short_kernel

struct ShortArgs {
    uint32_t m, k;
    const float *queries;
    uint32_t q_stride;
    const uint32_t *lb, *ub;
    const uint32_t *cand_row, *cand_idx;
    const uint64_t *cand_key;
    uint32_t *cand_count;     // fallback: left as is; fast path: top bit set so that final_topk skips the query
    float *cand_exact;        // fallback: exact of the shortlist, +inf elsewhere
    uint32_t *out_idx;
    float *out_dist;
    uint32_t *out_count;
    // > 0: the LOCAL stage of a leaf-sharded search (lists sorted by merge key).  This rank's share of the global
    // candidates (the m smallest keys over all ranks) is a PREFIX of its list, and the final rows are the k best exact
    // distances within the union of those prefixes: a candidate past the first local_head entries whose lower bound
    // lies above tau = the k-th smallest upper bound among the first local_head entries has k candidates with smaller
    // keys AND smaller exact distances in every prefix that contains it, so it can never reach the final k.  Its exact
    // distance is not computed: it travels as +inf (it still counts as one of the m candidates in the merge).  The
    // first local_head entries always get their exact distance.  No final rows are written in this mode.
    uint32_t local_head;
};

constexpr uint32_t kShortVPT = kMaxPreReorderK / 256;   // upper bounds per thread (registers)
constexpr uint32_t kLocalHead = 256;                    // entries of a sharded rank's list that are always re-ranked exactly

__global__ __launch_bounds__(256) void rerank_short_kernel(TxhIndexDev ix, ShortArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_q[];   // [dim]
    __shared__ uint32_t s_mins[256], s_tlist[1024], s_hist[1024], s_slist[256], s_tcnt;
    __shared__ uint64_t s_red[48];
    __shared__ uint32_t s_pos[kShortMaxFast], s_eb[kShortMaxFast];
    __shared__ uint64_t s_kk[kShortMaxFast];
    __shared__ uint32_t s_ns;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, nt = 256;
    const uint32_t m = a.m, k = a.k;
    const uint32_t nsel = a.cand_count[q];
    const uint32_t nout = min(k, nsel);
    for (uint32_t j = tid; j < ix.dim; j += nt) s_q[j] = a.queries[(size_t)q * a.q_stride + j];
    if (tid == 0) s_ns = 0;
    // tau = k-th smallest upper bound (everything is shortlisted when there are at most k candidates): a
    // tail rank, so the bounds stay in registers and only those under a pivot are ranked
    // (block_tail_select; an inexact result is the pivot, >= tau: a larger shortlist, still complete)
    uint32_t tau = 0xFFFFFFFFu;
    const uint32_t head = a.local_head;                        // (0: single-GPU final stage)
    const uint32_t ntau = head ? min(nsel, head) : nsel;       // entries whose upper bounds define tau
    if (ntau > k) {   // block-uniform
        uint32_t ubv[kShortVPT];   // (registers: 256 threads leave room, and the bounds are read once)
#pragma unroll
        for (int u = 0; u < (int)kShortVPT; ++u) {
            const uint32_t i = (uint32_t)u * nt + tid;
            ubv[u] = i < ntau ? a.ub[(size_t)q * m + i] : 0xFFFFFFFFu;
        }
        auto v = [&](int u) -> uint32_t { return ubv[u]; };
        bool exact;
        tau = block_tail_select<(int)kShortVPT>(v, k, s_mins, s_tlist, 1024, s_hist, s_slist, s_red, &s_tcnt, &exact);
    }
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nsel; b0 += nt) {
        const uint32_t i = b0 + tid;
        const bool keep = i < nsel && (i < head || a.lb[(size_t)q * m + i] <= tau);
        uint32_t wtot;
        const uint32_t wpre = wave_prefix_count(keep, &wtot);
        uint32_t base = 0;
        if ((tid & 63u) == 0 && wtot) base = atomicAdd(&s_ns, wtot);
        base = (uint32_t)__shfl((int)base, 0);
        if (keep && base + wpre < kShortMaxFast) s_pos[base + wpre] = i;
        if (head && i < nsel && !keep) a.cand_exact[(size_t)q * m + i] = __builtin_inff();   // (local stage: see ShortArgs)
    }
    __syncthreads();
    const uint32_t ns = s_ns;
    const bool fast = ns <= kShortMaxFast;
    // exact distances of the shortlist: rerank_kernel's arithmetic (8 FMA lane chains, fixed hsum tree,
    // unfused tail: simd/x86.rs:139-165, 31-44)
    const uint32_t chunks = ix.dim >> 3, lane8 = tid & 7u;
    const uint32_t total = fast ? ns : nsel;
    // (kShortR rows per 8-lane group and pass, their row gathers in flight together: one row per pass left every pass
    // waiting for a 512-byte gather from HBM -- 284 us at 10M x 128, m = 8192, where shortlists hold hundreds of rows)
    constexpr int kShortR = 4;
    for (uint32_t b0 = 0; b0 < total; b0 += (nt / 8) * kShortR) {
        uint32_t jx[kShortR], ci[kShortR];
        bool act[kShortR], listed[kShortR];
        const float *rowp[kShortR];
#pragma unroll
        for (int u = 0; u < kShortR; ++u) {
            jx[u] = b0 + (uint32_t)u * (nt / 8) + (tid >> 3);
            act[u] = jx[u] < total;
            ci[u] = 0;
            listed[u] = false;
            if (act[u]) {
                ci[u] = fast ? s_pos[jx[u]] : jx[u];
                listed[u] = fast || ci[u] < head || a.lb[(size_t)q * m + ci[u]] <= tau;
            }
            rowp[u] = ix.rows + (size_t)(listed[u] ? a.cand_row[(size_t)q * m + ci[u]] : 0u) * ix.stride;
        }
        float accv[kShortR];
#pragma unroll
        for (int u = 0; u < kShortR; ++u) accv[u] = 0.0f;
        for (uint32_t c = 0; c < chunks; ++c) {
#pragma unroll
            for (int u = 0; u < kShortR; ++u) {
                const float diff = s_q[8 * c + lane8] - rowp[u][8 * c + lane8];
                accv[u] = fmaf(diff, diff, accv[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < kShortR; ++u) {
            const float s1 = accv[u] + __shfl_down(accv[u], 4, 8);
            const float t1 = s1 + __shfl_down(s1, 1, 8);
            float r = t1 + __shfl_down(t1, 2, 8);
            if (lane8 == 0)
                for (uint32_t j = chunks * 8; j < ix.dim; ++j) {
                    const float diff = s_q[j] - rowp[u][j];
                    r = r + diff * diff;
                }
            if (!listed[u]) r = __builtin_inff();
            if (act[u] && lane8 == 0) {
                if (fast && !head) {
                    s_eb[jx[u]] = f32_to_ordered(r);
                    s_kk[jx[u]] = a.cand_key[(size_t)q * m + ci[u]];
                } else {
                    a.cand_exact[(size_t)q * m + ci[u]] = r;   // +inf outside the shortlist: final_topk orders the rest
                }
            }
        }
    }
    if (!fast || head) return;   // block-uniform: final_topk_kernel / the multi-GPU merge finish this query from cand_exact
    __syncthreads();
    // the k best of the shortlist by (exact, merge key) = the stable sort's first k: wave 0 runs k arg-min
    // rounds over its lanes' strided entries (ns <= 1024: <= 16 per lane)
    if (tid < 64) {
        constexpr int E = kShortMaxFast / 64;
        uint32_t eb[E];
        uint64_t kk[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t j = (uint32_t)e * 64u + tid;
            eb[e] = j < ns ? s_eb[j] : 0xFFFFFFFFu;
            kk[e] = j < ns ? s_kk[j] : SCANN_KEY_MAX;
        }
        for (uint32_t r0 = 0; r0 < nout; ++r0) {
            uint32_t b_eb = 0xFFFFFFFFu, b_sl = 0xFFFFFFFFu;
            uint64_t b_kk = SCANN_KEY_MAX;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((uint32_t)e * 64u < ns && (eb[e] < b_eb || (eb[e] == b_eb && kk[e] < b_kk))) {
                    b_eb = eb[e];
                    b_kk = kk[e];
                    b_sl = (uint32_t)e * 64u + tid;
                }
            wave_argmin96(b_eb, b_kk, b_sl);
            if (b_sl != 0xFFFFFFFFu && (b_sl & 63u) == tid) {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((uint32_t)e == (b_sl >> 6)) {
                        eb[e] = 0xFFFFFFFFu;
                        kk[e] = SCANN_KEY_MAX;
                    }
            }
            if (tid == 0) {
                a.out_idx[(size_t)q * k + r0] = a.cand_idx[(size_t)q * m + s_pos[b_sl]];
                a.out_dist[(size_t)q * k + r0] = ordered_to_f32(b_eb);
            }
        }
    }
    for (uint32_t i = nout + tid; i < k; i += nt) {
        a.out_idx[(size_t)q * k + i] = kInvalid;
        a.out_dist[(size_t)q * k + i] = __builtin_inff();
    }
    if (tid == 0) {
        a.out_count[q] = nout;
        a.cand_count[q] = 0x80000000u;   // final_topk_kernel has nothing left to do for this query
    }
}

// =====================================================================================
// Small batches (nq <= 16, short candidate streams): the search as THREE launches -- or ONE.
//
// The batched pipeline above groups (query, leaf) pairs by leaf, samples a filter bound and scans
// through tile queues: ~12 dependent launches, each ~4 us of dispatch on its own -- ~115-150 us of
// device time for ONE query, of which the scan is 15-40.  For a handful of queries none of that
// machinery pays:
//   1. select_leaves_kernel with inline centroid scoring            TreePartitioner::partition
//   2. small_scan_kernel: one workgroup per (query, leaf, chunk of points) builds the pair's table
//      in LDS (lut_build_kernel's arithmetic) and writes EVERY point's merge key at its stream
//      position (dense list, no bound, no atomics)                  mod.rs:297-339
//   3. small_finish_kernel: block per query: the m smallest keys (rank select), decode, exact
//      distances (exact_pair_8lanes), the k best by (exact, merge key)   mod.rs:283-293, 342-364
// small_fused_kernel (below) runs the three stages in one launch for exact-scan indexes and flat
// hashers.  Same keys, same arithmetic, same order: rows identical to the batched pipeline's.
// (Tried for the finish stage's rank select of a handful of keys: a tournament of 64-bit wave minima,
// m rounds per wave and m over the finalists -- slower than block_select's histogram passes.)
// =====================================================================================
constexpr uint32_t kSmallChunk = 1024;      // points per workgroup of the scan
constexpr uint32_t kSmallMaxM = 1024;       // candidates the finish kernel keeps in LDS
constexpr uint32_t kFinGroups = 4096;       // sub-stream minima of the finish kernel's tail select

// Exact distance of one (query, row) pair by ONE thread: the 8 AVX2 lane chains in registers, combined as
// the reference combines them (exact_pair_8lanes spreads the same chains over 8 lanes).  DistanceMeasure::
// distance (distance_measures/mod.rs:70-81) = the one-to-many kernels' per-row arithmetic (simd/x86.rs).
__device__ __forceinline__ float exact_pair_thread(int measure, uint32_t dim, const float *sq, const float *row) {
    const uint32_t chunks = dim >> 3;
    float ac[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, aa[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f},
          bb[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    // rows are 16-byte aligned when their stride is a multiple of 4 floats (compute_stride: always): 16-byte
    // loads, four 8-dim chunks (eight loads) in flight -- a 4-byte load per element costs the L1 as many line
    // requests as a 16-byte one
    const bool vec = (reinterpret_cast<uintptr_t>(row) & 15u) == 0;
    for (uint32_t c0 = 0; c0 < chunks; c0 += 4) {
        float xs[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + u < chunks) {
                if (vec) {
                    const float4 lo4 = *reinterpret_cast<const float4 *>(row + 8 * (c0 + u));
                    const float4 hi4 = *reinterpret_cast<const float4 *>(row + 8 * (c0 + u) + 4);
                    xs[u][0] = lo4.x; xs[u][1] = lo4.y; xs[u][2] = lo4.z; xs[u][3] = lo4.w;
                    xs[u][4] = hi4.x; xs[u][5] = hi4.y; xs[u][6] = hi4.z; xs[u][7] = hi4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xs[u][j] = row[8 * (c0 + u) + j];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
        if (c0 + u >= chunks) break;
        const uint32_t c = c0 + u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float qv = sq[8 * c + j], x = xs[u][j];
            if (measure == SCANN_HIP_DOT_PRODUCT) {
                ac[j] = fmaf(qv, x, ac[j]);
            } else if (measure == SCANN_HIP_L1) {
                ac[j] = ac[j] + fabsf(qv - x);
            } else if (measure == SCANN_HIP_COSINE) {
                ac[j] = ac[j] + qv * x;
                aa[j] = aa[j] + qv * qv;
                bb[j] = bb[j] + x * x;
            } else {
                const float d = qv - x;
                ac[j] = fmaf(d, d, ac[j]);
            }
        }
        }
    }
    float r, saa = 0.0f, sbb = 0.0f;
    if (measure == SCANN_HIP_COSINE) {   // wide 0.7 reduce_add, non-AVX build
        r = (((ac[0] + ac[1]) + ac[2]) + ac[3]) + (((ac[4] + ac[5]) + ac[6]) + ac[7]);
        saa = (((aa[0] + aa[1]) + aa[2]) + aa[3]) + (((aa[4] + aa[5]) + aa[6]) + aa[7]);
        sbb = (((bb[0] + bb[1]) + bb[2]) + bb[3]) + (((bb[4] + bb[5]) + bb[6]) + bb[7]);
    } else {                             // horizontal_sum_f32_avx2
        r = ((ac[0] + ac[4]) + (ac[1] + ac[5])) + ((ac[2] + ac[6]) + (ac[3] + ac[7]));
    }
    for (uint32_t j = chunks * 8; j < dim; ++j) {   // scalar tail, not fused
        const float qv = sq[j], x = row[j];
        if (measure == SCANN_HIP_DOT_PRODUCT) {
            r = r + qv * x;
        } else if (measure == SCANN_HIP_L1) {
            r = r + fabsf(qv - x);
        } else if (measure == SCANN_HIP_COSINE) {
            r = r + qv * x;
            saa = saa + qv * qv;
            sbb = sbb + x * x;
        } else {
            const float d = qv - x;
            r = r + d * d;
        }
    }
    if (measure == SCANN_HIP_DOT_PRODUCT) r = -r;
    if (measure == SCANN_HIP_L2) r = sqrtf(r);
    if (measure == SCANN_HIP_COSINE) {
        const float na = sqrtf(saa), nb = sqrtf(sbb);
        r = 1.0f - ((na == 0.0f || nb == 0.0f) ? 0.0f : r / (na * nb));
    }
    return r;
}

struct SmallArgs {
    uint32_t nq, P, m, k, cap, q_stride;
    int exact_reorder;
    const float *queries;
    const uint32_t *tokens, *vbase;
    uint64_t *cand;           // [nq][cap] dense: key of stream position v at cand[q][v]
    uint32_t *counters;
    const uint64_t *allow;
    uint64_t allow_bits;
    uint32_t *out_idx;
    float *out_dist;
    uint32_t *out_count;
    uint32_t *done;           // pinned completion flags [nq] of a host call (or nullptr) and their value
    uint32_t seq;
    uint32_t chunk;           // points per workgroup of the scan (a multiple of 256)
};

__global__ __launch_bounds__(256) void small_scan_kernel(TxhIndexDev ix, SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_lut[];   // [S][kp] then [dim] residual query
    const uint32_t pair = blockIdx.y, q = pair / a.P, r = pair - q * a.P, chunk = blockIdx.x, tid = threadIdx.x;
    if (pair == 0 && chunk == 0 && tid == 0) a.counters[CNT_STATUS] = 0;
    const uint32_t leaf = a.tokens[(size_t)q * a.P + r];
    const uint32_t lb = ix.leaf_off[leaf], size = ix.leaf_off[leaf + 1] - lb;
    const uint32_t c0 = chunk * a.chunk;
    if (c0 >= size) return;   // block-uniform
    const uint32_t S = ix.S, K = ix.K, dsub = ix.dsub, kp = ix.kp, dim = ix.dim;
    if (ix.exact_scan) {
        // SearchMode::Partitioned (scann.rs:213-252) and brute force: every row scored exactly; the dense
        // key list is then the whole result stream and the finish kernel takes its k smallest keys
        float *s_qe = s_lut;
        for (uint32_t j = tid; j < dim; j += 256) s_qe[j] = a.queries[(size_t)q * a.q_stride + j];
        __syncthreads();
        const uint32_t vbe = a.vbase[(size_t)q * (a.P + 1) + r];
        uint64_t *oute = a.cand + (size_t)q * a.cap;
        for (uint32_t j = c0 + tid; j < min(size, c0 + a.chunk); j += 256) {
            const uint32_t csr = lb + j;
            const float *row = ix.rows + (size_t)(ix.rows_csr ? csr : ix.leaf_ids[csr]) * ix.stride;
            const float dist = exact_pair_thread(ix.measure, dim, s_qe, row);
            const uint32_t vpos = vbe + j;
            if (vpos < a.cap) oute[vpos] = make_key(dist, vpos);
        }
        return;
    }
    float *s_qr = s_lut + S * kp;
    for (uint32_t j = tid; j < dim; j += 256) {   // residual q - centroid (mod.rs:309-316)
        float v = a.queries[(size_t)q * a.q_stride + j];
        if (ix.use_residuals) v = v - ix.centers[(size_t)leaf * dim + j];
        s_qr[j] = v;
    }
    __syncthreads();
    for (uint32_t e = tid; e < S * kp; e += 256) {   // LookupTable::from_query (lut.rs:47-70, codebook.rs:98-115)
        const uint32_t sub = e / kp, c = e - sub * kp;
        float acc = 0.0f;
        if (c < K) {
            const float *cb = ix.codebook + ((size_t)sub * K + c) * dsub;
            for (uint32_t j = 0; j < dsub; ++j) {
                const float d = s_qr[sub * dsub + j] - cb[j];
                acc = acc + d * d;
            }
        }
        s_lut[e] = acc;
    }
    __syncthreads();
    const uint32_t vb = a.vbase[(size_t)q * (a.P + 1) + r];
    const uint32_t bits = ix.code_bits, per = 32u / bits, mask = (1u << bits) - 1u, nw = ix.nw;
    uint64_t *out = a.cand + (size_t)q * a.cap;
    for (uint32_t j = c0 + tid; j < min(size, c0 + a.chunk); j += 256) {
        const uint32_t *w = ix.codes + (size_t)(lb + j) * nw;
        float acc = 0.0f;   // LookupTable::compute_distance (lut.rs:74-82): 0.0 + lut[0][c0] + lut[1][c1] ...
        for (uint32_t sub = 0; sub < S; ++sub) {
            const uint32_t code = (w[sub / per] >> (bits * (sub % per))) & mask;
            const float tv = s_lut[sub * kp + code];
            acc = sub == 0 ? tv : acc + tv;
        }
        const uint32_t vpos = vb + j;
        if (vpos < a.cap)
            out[vpos] = row_allowed(ix, a.allow, a.allow_bits, lb + j) ? make_key(acc, vpos) : SCANN_KEY_MAX;
    }
}

// The finish stage: body of small_finish_kernel and last stage of small_fused_kernel (there vbq / tokq are
// the workgroup's LDS copies of the query's key bases and tokens).
// AGENT: the key list was written by other workgroups of the SAME launch (agent-scope write-through stores):
// read it with agent-scope loads (the per-XCD L2s are not coherent with each other inside a kernel).
template <bool AGENT>
__device__ __forceinline__ uint64_t list_key(const uint64_t *p) {
    if constexpr (AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

template <bool AGENT>
__device__ __forceinline__ void small_finish_body(const TxhIndexDev &ix, const SmallArgs &a, uint32_t q, float *s_q,
                                                  const uint32_t *vbq, const uint32_t *tokq) {
    __shared__ uint64_t s_keys[kSmallMaxM], s_slist[kSelListMax], s_red[48], s_fin[kFinGroups];
    __shared__ uint32_t s_hist[kSelBinsMax], s_eb[kSmallMaxM], s_idx[kSmallMaxM], s_n;
    const uint32_t tid = threadIdx.x, nt = kSelectThreads;
    const uint32_t P = a.P, m = a.m, k = a.k;
    // result rows: plain stores, or -- when the host polls a completion flag in the same pinned buffer --
    // system-scope write-through stores (visible to the host once they have completed)
    const bool polled = a.done != nullptr;
    auto out_store = [&](auto *p, auto v) {
        if (polled) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else *p = v;
    };
    const uint32_t cnt = min(vbq[P], a.cap);
    const uint64_t *list = a.cand + (size_t)q * a.cap;
    for (uint32_t j = tid; j < ix.dim; j += nt) s_q[j] = a.queries[(size_t)q * a.q_stride + j];
    if (tid == 0) s_n = 0;
    __syncthreads();
    // the m smallest keys (keys are unique; SCANN_KEY_MAX = a point the restrict filter rejected).  m is
    // far down the stream's tail (1000 of 100 k), so: minima of kFinGroups interleaved sub-streams -> the
    // m-th smallest minimum is a pivot just above the m-th smallest key (the m smallest minima are m
    // distinct keys) -> only the keys under the pivot are ranked.  Two passes over the list instead of
    // the histogram select's five; the histogram select remains for m close to cnt.
    uint64_t T = SCANN_KEY_MAX - 1;
    uint32_t from_fin = 0;   // > 0: the candidates are among s_fin[0 .. from_fin)
    if (cnt > m) {
        bool done = false;
        if (cnt <= kFinGroups) {   // a short stream: one load of the list into LDS, everything else there
            for (uint32_t i = tid; i < cnt; i += nt) s_fin[i] = list_key<AGENT>(list + i);
            __syncthreads();
            const uint64_t t = block_select<uint64_t>(s_fin, cnt, m, sel_cfg(cnt), s_hist, s_slist, s_red);
            if (t != SCANN_KEY_MAX) T = t;
            done = true;
            from_fin = cnt;
            __syncthreads();
        } else if ((uint64_t)m * 4 <= kFinGroups) {
            // Both passes read the list as 16-byte pairs, four loads in flight per thread (one workgroup
            // streams 800 KB at 100 k points: the passes are bound by its load round trips).  `head` = one
            // leading key when the list starts on an odd 8-byte slot; a last odd key is `tail`.
            constexpr int G = kFinGroups / kSelectThreads;   // minima per thread (any partition of the list works)
            const uint32_t head = (reinterpret_cast<uintptr_t>(list) & 8u) ? 1u : 0u;
            const uint32_t npairs = (cnt - head) >> 1;
            const bool tail = ((cnt - head) & 1u) != 0;
            const ulonglong2 *pairs = reinterpret_cast<const ulonglong2 *>(list + head);
            uint64_t mn[G];
#pragma unroll
            for (int g = 0; g < G; ++g) mn[g] = SCANN_KEY_MAX;
            for (uint32_t p0 = 0; p0 < npairs; p0 += nt * G) {
                ulonglong2 v[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t pi = p0 + (uint32_t)g * nt + tid;
                    v[g] = make_ulonglong2(SCANN_KEY_MAX, SCANN_KEY_MAX);
                    if (pi < npairs) {
                        if constexpr (AGENT) {
                            v[g].x = list_key<true>(list + head + 2 * (size_t)pi);
                            v[g].y = list_key<true>(list + head + 2 * (size_t)pi + 1);
                        } else {
                            v[g] = pairs[pi];
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint64_t lo2 = v[g].x < v[g].y ? v[g].x : v[g].y;
                    mn[g] = lo2 < mn[g] ? lo2 : mn[g];
                }
            }
            if (tid == 0 && head) { const uint64_t k0 = list_key<AGENT>(list); mn[0] = k0 < mn[0] ? k0 : mn[0]; }
            if (tid == 1 && tail) { const uint64_t k1 = list_key<AGENT>(list + cnt - 1); mn[0] = k1 < mn[0] ? k1 : mn[0]; }
#pragma unroll
            for (int g = 0; g < G; ++g) s_fin[(uint32_t)g * nt + tid] = mn[g];
            __syncthreads();
            const SelCfg gcfg = sel_cfg(kFinGroups);
            const uint64_t pivot = block_select<uint64_t>(s_fin, kFinGroups, m, gcfg, s_hist, s_slist, s_red);
            __syncthreads();
            if (pivot != SCANN_KEY_MAX) {
                if (tid == 0) s_n = 0;
                __syncthreads();
                // keys under the pivot -> s_fin (reused): ~1 % of the list, one LDS atomic each
                auto take = [&](uint64_t key) {
                    if (key <= pivot) {
                        const uint32_t pos = atomicAdd(&s_n, 1u);
                        if (pos < kFinGroups) s_fin[pos] = key;
                    }
                };
                for (uint32_t p0 = 0; p0 < npairs; p0 += nt * G) {
                    ulonglong2 v[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const uint32_t pi = p0 + (uint32_t)g * nt + tid;
                        v[g] = make_ulonglong2(SCANN_KEY_MAX, SCANN_KEY_MAX);
                    if (pi < npairs) {
                        if constexpr (AGENT) {
                            v[g].x = list_key<true>(list + head + 2 * (size_t)pi);
                            v[g].y = list_key<true>(list + head + 2 * (size_t)pi + 1);
                        } else {
                            v[g] = pairs[pi];
                        }
                    }
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        take(v[g].x);
                        take(v[g].y);
                    }
                }
                if (tid == 0 && head) take(list_key<AGENT>(list));
                if (tid == 1 && tail) take(list_key<AGENT>(list + cnt - 1));
                __syncthreads();
                const uint32_t c2 = s_n;
                __syncthreads();
                if (c2 <= kFinGroups) {   // (>= m by construction)
                    const uint64_t t = block_select<uint64_t>(s_fin, c2, m, sel_cfg(c2), s_hist, s_slist, s_red);
                    if (t != SCANN_KEY_MAX) T = t;
                    done = true;
                    from_fin = c2;       // the m smallest keys are all among these: no third pass over the list
                    __syncthreads();
                }
                if (tid == 0) s_n = 0;
                __syncthreads();
            }
        }
        if (!done) {
            const SelCfg cfg = sel_cfg(cnt);
            const uint64_t t = block_select<uint64_t, AGENT>(list, cnt, m, cfg, s_hist, s_slist, s_red);
            if (t != SCANN_KEY_MAX) T = t;   // fewer than m allowed points: keep them all
            __syncthreads();
        }
    }

    const uint64_t *src = from_fin ? s_fin : list;
    const uint32_t nsrc = from_fin ? from_fin : cnt;
    for (uint32_t b0 = 0; b0 < nsrc; b0 += nt) {
        const uint32_t i = b0 + tid;
        uint64_t key = SCANN_KEY_MAX;
        if (i < nsrc) key = from_fin ? src[i] : list_key<AGENT>(src + i);
        const bool keep = key <= T;
        uint32_t wtot;
        const uint32_t wpre = wave_prefix_count(keep, &wtot);
        uint32_t base = 0;
        if ((tid & 63u) == 0 && wtot) base = atomicAdd(&s_n, wtot);
        base = (uint32_t)__shfl((int)base, 0);
        if (keep && base + wpre < kSmallMaxM) s_keys[base + wpre] = key;
    }
    __syncthreads();
    const uint32_t nsel = min(s_n, min(m, kSmallMaxM));

    // decode tables of this query (key base and first CSR row of each selected leaf) in LDS: the
    // per-candidate chain vbase -> token -> leaf_off is otherwise three dependent global loads per pass
    uint32_t *s_dvb = reinterpret_cast<uint32_t *>(s_fin), *s_drow = s_dvb + kDecodeStage;   // (s_fin is free now)
    const bool staged = P <= kDecodeStage;
    if (staged)
        for (uint32_t r = tid; r < P; r += nt) {
            s_dvb[r] = vbq[r];
            s_drow[r] = ix.leaf_off[tokq[r]];
        }
    __syncthreads();
    // decode + exact distance: 8 lanes per candidate.  The passes' row ids first (their leaf_ids loads
    // travel together), then one memory round trip per candidate row of up to 128 dims.
    const uint32_t lane8 = tid & 7u;
    constexpr uint32_t kPasses = kSmallMaxM / (kSelectThreads / 8);
    uint32_t idxs[kPasses], rowis[kPasses];
#pragma unroll
    for (uint32_t ps = 0; ps < kPasses; ++ps) {
        const uint32_t c = ps * (nt / 8) + (tid >> 3);
        uint32_t idx = 0, rowi = 0;
        if (c < nsel) {
            const uint32_t vpos = (uint32_t)s_keys[c];
            uint32_t lo = 0, hi = P;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((staged ? s_dvb[mid] : vbq[mid]) <= vpos) lo = mid; else hi = mid;
            }
            const uint32_t csr = staged ? s_drow[lo] + (vpos - s_dvb[lo])
                                        : ix.leaf_off[tokq[lo]] + (vpos - vbq[lo]);
            idx = ix.leaf_ids ? ix.leaf_ids[csr] : csr;
            rowi = ix.rows_csr ? csr : idx;
        }
        idxs[ps] = idx;
        rowis[ps] = rowi;
    }
#pragma unroll
    for (uint32_t ps = 0; ps < kPasses; ++ps) {
        const uint32_t c = ps * (nt / 8) + (tid >> 3);
        if (ps * (nt / 8) >= nsel) break;   // block-uniform
        const bool act = c < nsel;
        float r = 0.0f;
        if (a.exact_reorder) r = exact_pair_8lanes<16>(ix, s_q, ix.rows + (size_t)rowis[ps] * ix.stride, act, lane8);
        if (act && lane8 == 0) {
            s_idx[c] = idxs[ps];
            // without re-ordering the k best by approximate distance are wanted: order by the key itself
            s_eb[c] = a.exact_reorder ? f32_to_ordered(r) : (uint32_t)(s_keys[c] >> 32);
        }
    }
    __syncthreads();

    const uint32_t nout = min(k, nsel);
    if (nsel <= 64u) {
        // few candidates (m = k: exact scans, hashers without re-ordering): one per lane of wave 0, each lane
        // counts the candidates ahead of it in the (exact, merge key) order and writes its row at that rank
        if (tid < 64) {
            const uint32_t eb = tid < nsel ? s_eb[tid] : 0xFFFFFFFFu;
            const uint64_t kk = tid < nsel ? s_keys[tid] : SCANN_KEY_MAX;
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nsel; ++j) {
                const uint32_t oe = (uint32_t)__shfl((int)eb, (int)j);
                const uint64_t ok = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(kk >> 32), (int)j) << 32) |
                                    (uint32_t)__shfl((int)(uint32_t)kk, (int)j);
                rank += (oe < eb || (oe == eb && ok < kk)) ? 1u : 0u;
            }
            if (tid < nsel && rank < nout) {
                out_store(&a.out_idx[(size_t)q * k + rank], s_idx[tid]);
                out_store(&a.out_dist[(size_t)q * k + rank], ordered_to_f32(eb));
            }
        }
    } else {
        // the k best by (exact, merge key), two levels: every wave extracts the k best of ITS 64 candidates (one
        // per lane: a round is one wave arg-min), then wave 0 the k best of the waves' finalists
        // (s_hist / s_slist are free by now: [0, 1024) ordered exact, [1024, 2048) candidate slot; merge keys)
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        const uint32_t nwv = (nsel + 63u) >> 6;           // waves that hold candidates
        uint32_t *f_eb = s_hist, *f_sl = s_hist + kSmallMaxM;
        uint64_t *f_kk = s_slist;
        {
            uint32_t eb = tid < nsel ? s_eb[tid] : 0xFFFFFFFFu;
            uint64_t kk = tid < nsel ? s_keys[tid] : SCANN_KEY_MAX;
            if (wave < nwv) {
                for (uint32_t r0 = 0; r0 < nout; ++r0) {
                    uint32_t b_eb = eb, b_sl = kk == SCANN_KEY_MAX ? 0xFFFFFFFFu : tid;
                    uint64_t b_kk = kk;
                    wave_argmin96(b_eb, b_kk, b_sl);
                    if (b_sl == tid) {   // the winner leaves the pool
                        eb = 0xFFFFFFFFu;
                        kk = SCANN_KEY_MAX;
                    }
                    if (lane == 0) {
                        f_eb[wave * nout + r0] = b_eb;
                        f_kk[wave * nout + r0] = b_kk;
                        f_sl[wave * nout + r0] = b_sl;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 64) {
            constexpr int E = kSmallMaxM / 64;
            const uint32_t nfin = nwv * nout;             // <= 16 * 64
            uint32_t eb[E], sl[E];
            uint64_t kk[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t j = (uint32_t)e * 64u + tid;
                const bool ok = j < nfin && f_sl[j] != 0xFFFFFFFFu;
                eb[e] = ok ? f_eb[j] : 0xFFFFFFFFu;
                kk[e] = ok ? f_kk[j] : SCANN_KEY_MAX;
                sl[e] = ok ? f_sl[j] : 0xFFFFFFFFu;
            }
            for (uint32_t r0 = 0; r0 < nout; ++r0) {
                uint32_t b_eb = 0xFFFFFFFFu, b_sl = 0xFFFFFFFFu;
                uint64_t b_kk = SCANN_KEY_MAX;
                int b_e = -1;
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((uint32_t)e * 64u < nfin && sl[e] != 0xFFFFFFFFu &&
                        (eb[e] < b_eb || (eb[e] == b_eb && kk[e] < b_kk))) {
                        b_eb = eb[e];
                        b_kk = kk[e];
                        b_sl = sl[e];
                        b_e = e;
                    }
                const uint32_t mine = b_sl;
                wave_argmin96(b_eb, b_kk, b_sl);
                if (b_sl != 0xFFFFFFFFu && mine == b_sl) {   // (candidate slots are unique: exactly one lane)
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (e == b_e) {
                            eb[e] = 0xFFFFFFFFu;
                            kk[e] = SCANN_KEY_MAX;
                            sl[e] = 0xFFFFFFFFu;
                        }
                }
                if (tid == 0) {
                    out_store(&a.out_idx[(size_t)q * k + r0], s_idx[b_sl]);
                    out_store(&a.out_dist[(size_t)q * k + r0], ordered_to_f32(b_eb));
                }
            }
        }
    }
    for (uint32_t i = nout + tid; i < k; i += nt) {
        out_store(&a.out_idx[(size_t)q * k + i], kInvalid);
        out_store(&a.out_dist[(size_t)q * k + i], __builtin_inff());
    }
    if (tid == 0) out_store(&a.out_count[q], nout);

    if (a.done) {   // host call polling for completion: the flag after this block's result rows
        // (rows and flag live in pinned host memory and are written with system-scope write-through stores,
        // so waiting for their completion orders them; a system-scope release FENCE would write back the
        // whole L2 -- tens of microseconds)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&a.done[q], a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(kSelectThreads) void small_finish_kernel(TxhIndexDev ix, SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_qf[];   // [dim]
    const uint32_t q = blockIdx.x;
    small_finish_body<false>(ix, a, q, s_qf, a.vbase + (size_t)q * (a.P + 1), a.tokens + (size_t)q * a.P);
}

// =====================================================================================
// Multi-GPU merge of gathered (key, idx, exact) triples [world][nq][m].  Every rank's
// list is sorted by key and keys are unique per query, so an element's position in the
// merged order is the number of smaller keys over all lists (binary searches); no sort
// of world*m keys is needed.  Then the same stable exact sort as above.
// =====================================================================================
__global__ __launch_bounds__(kSelectThreads) void merge_kernel(
    uint32_t world, uint32_t nq, uint32_t m_local, uint32_t m, uint32_t k, uint32_t m2max,
    size_t rank_stride, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ idx,
    const float *__restrict__ exact, const uint32_t *__restrict__ count,
    uint32_t *__restrict__ out_idx, float *__restrict__ out_dist,
    uint32_t *__restrict__ out_count, uint32_t *__restrict__ status, const uint32_t *__restrict__ qoff) {
    // rank g's arrays start rank_stride BYTES after rank g-1's (a packed all_gather buffer);
    // rank_stride == 0 means dense [world][nq][m_local] arrays.  qoff != nullptr: COMPACT lists (comm.hip): the
    // entries of (rank g, query q) start at element qoff[g * nq + q] of rank g's arrays instead of q * m_local.
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];   // [m2max] (sort path)
    uint32_t *s_src = reinterpret_cast<uint32_t *>(skeys + m2max);     // [m2max] packed (g, slot)
    uint64_t *s_mth = reinterpret_cast<uint64_t *>(s_src + m2max);     // [1] key at rank nsel-1
    uint64_t *s_red = s_mth + 1;                                       // [kSelectThreads/64]
    const uint32_t q = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    auto at = [&](const void *base, uint32_t g, size_t dense_elems, size_t esz) -> const char * {
        return reinterpret_cast<const char *>(base) +
               (rank_stride ? (size_t)g * rank_stride : (size_t)g * dense_elems * esz);
    };
    auto first = [&](uint32_t g) { return qoff ? (size_t)qoff[(size_t)g * nq + q] : (size_t)q * m_local; };
    auto keys_of = [&](uint32_t g) {
        return reinterpret_cast<const uint64_t *>(at(keys, g, (size_t)nq * m_local, 8)) + first(g);
    };
    auto idx_of = [&](uint32_t g) {
        return reinterpret_cast<const uint32_t *>(at(idx, g, (size_t)nq * m_local, 4)) + first(g);
    };
    auto exact_of = [&](uint32_t g) {
        return reinterpret_cast<const float *>(at(exact, g, (size_t)nq * m_local, 4)) + first(g);
    };
    auto count_of = [&](uint32_t g) {
        return reinterpret_cast<const uint32_t *>(at(count, g, (size_t)nq, 4))[q];
    };
    uint32_t tot = 0;
    for (uint32_t g = 0; g < world; ++g) tot += count_of(g);
    const uint32_t nsel = min(m, tot);
    if (tid == 0) *s_mth = SCANN_KEY_MAX;
    __syncthreads();
    for (uint32_t g = 0; g < world; ++g) {
        const uint32_t cg = count_of(g);
        const uint64_t *kl = keys_of(g);
        for (uint32_t i = tid; i < cg; i += nt) {
            const uint64_t key = kl[i];
            uint32_t rank = i;                      // smaller keys in its own list
            for (uint32_t g2 = 0; g2 < world; ++g2) {
                if (g2 == g) continue;
                const uint64_t *k2 = keys_of(g2);
                uint32_t lo = 0, hi = count_of(g2);
                while (lo < hi) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (k2[mid] < key) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            if (rank < nsel) s_src[rank] = (g << 24) | i;     // world <= 64, m_local <= 8192
            if (rank + 1 == nsel) *s_mth = key;
        }
    }
    __syncthreads();
    // Ranks may send fewer than m candidates (m_local < m, sized for a random shard's share
    // of the global top-m).  A truncated list whose last key is still below the global m-th
    // key could have held more members: report it, the caller re-runs with m_local = m.
    if (status && m_local < m && tid < world) {
        const uint32_t cg = count_of(tid);
        if (cg == m_local && cg > 0) {
            const uint64_t last = keys_of(tid)[cg - 1];
            if (tot < m || last < *s_mth) atomicMax(status, (uint32_t)SCANN_HIP_ABORTED);
        }
    }
    const uint32_t nout = min(k, nsel);
    if (k <= kTopkMaxK) {
        // stable sort by exact == order by (exact, merged approx rank): k block-wide arg-mins
        constexpr int E = kMaxPreReorderK / kSelectThreads;
        uint64_t key2[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t r = e * kSelectThreads + tid;
            key2[e] = SCANN_KEY_MAX;
            if (r < nsel) {
                const uint32_t src = s_src[r];
                key2[e] = make_key(exact_of(src >> 24)[src & 0xFFFFFFu], r);
            }
        }
        for (uint32_t r = 0; r < nout; ++r) {
            uint64_t b = key2[0];
#pragma unroll
            for (int e = 1; e < E; ++e) b = min(b, key2[e]);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) b = min(b, (uint64_t)__shfl_xor((unsigned long long)b, d, 64));
            if ((tid & 63u) == 0) s_red[tid >> 6] = b;
            __syncthreads();
            uint64_t gmin = s_red[0];
#pragma unroll
            for (int w2 = 1; w2 < (int)(kSelectThreads / 64); ++w2) gmin = min(gmin, s_red[w2]);
            const uint32_t rk = (uint32_t)gmin;
            if ((rk & (kSelectThreads - 1)) == tid) {
                const uint32_t we = rk / kSelectThreads;
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((uint32_t)e == we) key2[e] = SCANN_KEY_MAX;
                const uint32_t src = s_src[rk];
                out_idx[(size_t)q * k + r] = idx_of(src >> 24)[src & 0xFFFFFFu];
                out_dist[(size_t)q * k + r] = ordered_to_f32((uint32_t)(gmin >> 32));
            }
            __syncthreads();
        }
        for (uint32_t i = nout + tid; i < k; i += nt) {
            out_idx[(size_t)q * k + i] = kInvalid;
            out_dist[(size_t)q * k + i] = __builtin_inff();
        }
        if (tid == 0) out_count[q] = nout;
        return;
    }
    uint32_t m2 = 1;
    while (m2 < nsel) m2 <<= 1;
    for (uint32_t i = tid; i < m2; i += nt) {
        uint64_t kv = SCANN_KEY_MAX;
        if (i < nsel) {
            const uint32_t src = s_src[i];
            kv = make_key(exact_of(src >> 24)[src & 0xFFFFFFu], i);
        }
        skeys[i] = kv;
    }
    __syncthreads();
    bitonic_sort_lds(skeys, m2);
    for (uint32_t i = tid; i < k; i += nt) {
        uint32_t oi = kInvalid;
        float od = __builtin_inff();
        if (i < nout) {
            const uint64_t key = skeys[i];
            const uint32_t src = s_src[(uint32_t)key];
            oi = idx_of(src >> 24)[src & 0xFFFFFFu];
            od = ordered_to_f32((uint32_t)(key >> 32));
        }
        out_idx[(size_t)q * k + i] = oi;
        out_dist[(size_t)q * k + i] = od;
    }
    if (tid == 0) out_count[q] = nout;
}

// =====================================================================================
// Building blocks exposed through the C ABI
// =====================================================================================
// All-pairs ADC distances for explicit f32 LUTs [nq][S][K]: out [nq][n_local].
// hashes/lut.rs:74-82: sum = 0.0; for s ascending: sum += lut[s][code[s]].
__global__ __launch_bounds__(256) void adc_distances_kernel(TxhIndexDev ix,
                                                            const float *__restrict__ luts,
                                                            float *__restrict__ out) {
    extern __shared__ float slut[];   // [S][K]
    const uint32_t q = blockIdx.y, S = ix.S, K = ix.K, nw = ix.nw;
    const uint32_t bits = ix.code_bits, per = 32u / bits, mask = (1u << bits) - 1u;
    for (uint32_t e = threadIdx.x; e < S * K; e += blockDim.x) slut[e] = luts[(size_t)q * S * K + e];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ix.n_local;
         i += (uint64_t)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (uint32_t sub = 0; sub < S; ++sub) {
            const uint32_t w = ix.codes[i * nw + sub / per];
            const uint32_t code = (w >> (bits * (sub % per))) & mask;
            acc = acc + slut[sub * K + (code < K ? code : 0u)];
        }
        out[(size_t)q * ix.n_local + i] = acc;
    }
}

// Lut16SimdTables::compute_distances_batch (hashes/lut16_simd.rs:119-141 over
// simd/dispatch.rs:259-295): u32 sum of u8 table entries, then sum * mult + bias * S.
__global__ __launch_bounds__(256) void lut16_u8_batch_kernel(
    const uint8_t *__restrict__ packed, const uint8_t *__restrict__ lut8, uint32_t S,
    uint64_t n, float bias, float mult, float *__restrict__ out) {
    extern __shared__ uint8_t s_lut8[];  // [S*16]
    for (uint32_t e = threadIdx.x; e < S * 16; e += blockDim.x) s_lut8[e] = lut8[e];
    __syncthreads();
    const uint32_t bpp = (S + 1) / 2;
    const float bias_total = bias * (float)S;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *row = packed + i * bpp;
        uint32_t sum = 0, sub = 0;
        for (uint32_t b = 0; b < bpp; ++b) {
            const uint32_t byte = row[b];
            if (sub < S) { sum += s_lut8[sub * 16 + (byte & 15u)]; ++sub; }
            if (sub < S) { sum += s_lut8[sub * 16 + (byte >> 4)]; ++sub; }
        }
        const float r = (float)sum * mult;
        out[i] = r + bias_total;
    }
}

// Lut16SimdTables::from_float_tables (hashes/lut16_simd.rs:39-90): global min / max of the S x 16
// entries (f32::min / f32::max: a NaN operand is ignored), range = max - min, scale = 255 / range
// (1 when range < 1e-10), lut8 = round((v - min) * scale) as u8 (round half away from zero; `as u8`
// saturates and maps NaN to 0), bias = min, multiplier = 1 / scale (1 in the degenerate case).
__global__ __launch_bounds__(256) void lut16_quantize_kernel(const float *__restrict__ tables, uint32_t S,
                                                             uint8_t *__restrict__ lut8,
                                                             float *__restrict__ bias_mult) {
    __shared__ float s_min[4], s_max[4];
    const uint32_t tid = threadIdx.x, n = S * 16;
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;   // f32::MAX / f32::MIN
    for (uint32_t i = tid; i < n; i += 256) {
        const float v = tables[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if ((tid & 63u) == 0) {
        s_min[tid >> 6] = mn;
        s_max[tid >> 6] = mx;
    }
    __syncthreads();
    mn = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
    mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    const float range = mx - mn;
    const bool degenerate = range < 1e-10f;
    const float scale = degenerate ? 1.0f : 255.0f / range;
    for (uint32_t i = tid; i < n; i += 256) {
        const float r = roundf((tables[i] - mn) * scale);
        lut8[i] = !(r > 0.0f) ? (uint8_t)0 : (r >= 255.0f ? (uint8_t)255 : (uint8_t)r);
    }
    if (tid == 0) {
        bias_mult[0] = mn;
        bias_mult[1] = degenerate ? 1.0f : 1.0f / scale;
    }
}

// Codebook::encode (hashes/codebook.rs:82-95): per subspace argmin over K with strict '<'.
__global__ __launch_bounds__(256) void encode_kernel(
    const float *__restrict__ codebook, uint32_t S, uint32_t K, uint32_t dsub,
    const float *__restrict__ rows, uint64_t n, uint32_t stride,
    const float *__restrict__ centers, const uint32_t *__restrict__ leaf_of_row,
    uint8_t *__restrict__ out) {
    const uint64_t total = n * S;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = e / S;
        const uint32_t s = (uint32_t)(e - i * S);
        const float *x = rows + i * stride + s * dsub;
        const float *cen = centers ? centers + (size_t)leaf_of_row[i] * (S * dsub) + s * dsub
                                   : nullptr;
        float best = __builtin_inff();
        uint32_t bi = 0;
        for (uint32_t c = 0; c < K; ++c) {
            const float *cb = codebook + ((size_t)s * K + c) * dsub;
            float d = 0.0f;
            for (uint32_t j = 0; j < dsub; ++j) {
                float xv = x[j];
                if (cen) xv = xv - cen[j];
                const float t = xv - cb[j];
                d = d + t * t;
            }
            if (d < best) {
                best = d;
                bi = c;
            }
        }
        out[e] = (uint8_t)bi;
    }
}

// =====================================================================================
// launchers
// =====================================================================================
#define LAUNCH_CHECK()                                                                \
    do {                                                                              \
        hipError_t _e = hipGetLastError();                                            \
        if (_e != hipSuccess)                                                         \
            return fail(SCANN_HIP_INTERNAL, std::string("kernel launch: ") + hipGetErrorString(_e)); \
    } while (0)

template <typename F>
static int set_dyn_lds(F kernel, size_t bytes) {
    // Always the same value (the CU's 160 KB), never the launch's own size: threads searching
    // different indexes set this attribute concurrently, and a smaller value written by one of
    // them must not undercut another's launch.
    constexpr size_t kMaxLds = 160 * 1024;
    if (bytes > kMaxLds) return fail(SCANN_HIP_RESOURCE_EXHAUSTED, "kernel needs more than 160 KB of LDS");
    if (bytes > 64 * 1024)
        SCANN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
    return SCANN_HIP_OK;
}

// The same for a kernel that also holds `static_bytes` of static LDS arrays: the attribute bounds the DYNAMIC part, and
// static + dynamic may not exceed the CU's 160 KB (again one constant per kernel).
template <typename F>
static int set_dyn_lds_with_static(F kernel, size_t bytes, size_t static_bytes) {
    constexpr size_t kMaxLds = 160 * 1024;
    if (bytes + static_bytes > kMaxLds) return fail(SCANN_HIP_RESOURCE_EXHAUSTED, "kernel needs more than 160 KB of LDS");
    if (bytes > 64 * 1024)   // (the attribute bounds static + dynamic: 160 KB for a kernel with static arrays is refused)
        SCANN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)(kMaxLds - static_bytes)));
    return SCANN_HIP_OK;
}

static int launch_partition_stage(const TxhIndexDev &ix, const TxhWork &w, hipStream_t st) {
    if (ix.ah_mode) {
        hipLaunchKernelGGL(ah_tokens_kernel, dim3(ceil_div_u32(w.nq, 256)), dim3(256), 0, st, w.nq,
                           ix.leaf_gsize, ix.leaf_off, w.st, w.tokens, w.token_dists, w.vbase, w.sbase);
        LAUNCH_CHECK();
        return SCANN_HIP_OK;
    }
    // queries per single-wave block: fewer when the grid would not fill the chip (one thread per
    // centroid, so a block's work is 64 centroids x QT queries)
    const uint64_t waves16 = (uint64_t)ceil_div_u32(ix.L, 64) * ceil_div_u32(w.nq, 16);
    if (waves16 >= 8192) {
        const size_t lds1 = (size_t)16 * ix.dim * sizeof(float);
        SCANN_TRY(set_dyn_lds(centroid_scores_kernel<16>, lds1));
        hipLaunchKernelGGL(centroid_scores_kernel<16>, dim3(ceil_div_u32(ix.L, 64), ceil_div_u32(w.nq, 16)),
                           dim3(64), lds1, st, ix.centers, ix.L, ix.dim, w.queries, w.nq, w.q_stride,
                           w.cdist);
    } else {
        const size_t lds1 = (size_t)4 * ix.dim * sizeof(float);
        SCANN_TRY(set_dyn_lds(centroid_scores_kernel<4>, lds1));
        hipLaunchKernelGGL(centroid_scores_kernel<4>, dim3(ceil_div_u32(ix.L, 64), ceil_div_u32(w.nq, 4)),
                           dim3(64), lds1, st, ix.centers, ix.L, ix.dim, w.queries, w.nq, w.q_stride,
                           w.cdist);
    }
    LAUNCH_CHECK();
    const uint32_t n2 = next_pow2_u32(ix.L);
    // select path when P is small against L: rank-select the P-th key, sort P keys instead of L
    const uint32_t p2 = (w.P * 4u <= n2) ? next_pow2_u32(std::max(1u, w.P)) : 0u;
    const SelCfg lcfg = sel_cfg(ix.L);
    const size_t lds2 = (size_t)(n2 + p2) * sizeof(uint64_t) + (size_t)lcfg.bins * 4 + (size_t)lcfg.list * 8 +
                        48 * 8 + 64 * 4;
    SCANN_TRY(set_dyn_lds(select_leaves_kernel, lds2));
    hipLaunchKernelGGL(select_leaves_kernel, dim3(w.nq), dim3(p2 && ix.L <= 4096 ? 256u : kSelectThreads), lds2, st,
                       w.cdist, ix.L, n2, w.P, p2, ix.leaf_gsize, ix.leaf_off, w.st, w.tokens, w.token_dists,
                       w.vbase, w.sbase, (const float *)nullptr, (const float *)nullptr, 0u, 0u, 0u);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int txh_launch_partition_only(const TxhIndexDev &ix, const TxhWork &w, hipStream_t st) {
    return launch_partition_stage(ix, w, st);
}

template <typename C>
static int launch_scan_stages(const TxhIndexDev &ix, const TxhWork &w, hipStream_t st,
                              hipEvent_t ev0, hipEvent_t ev1) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (!w.no_threshold) {
        SampleArgs sa;
        sa.pair_off = w.pair_off; sa.stile_off = w.stile_off; sa.pair_q = w.pair_q;
        sa.pair_sbase = w.pair_sbase; sa.counters = w.counters; sa.lutq = w.lutq; sa.samp = w.samp;
        sa.scap = w.scap; sa.st = w.st; sa.qpt = w.sqpt; sa.allow = w.allow; sa.allow_bits = w.allow_bits;
        const size_t lds_smp = (size_t)2 * C::LUT4 * 16 + 16;
        SCANN_TRY(set_dyn_lds(adc_sample_kernel<C>, lds_smp));
        hipLaunchKernelGGL(adc_sample_kernel<C>, dim3((uint32_t)cus * 8u), dim3(kScanThreads), lds_smp, st, ix, sa);
        LAUNCH_CHECK();
    }
    {
        const SelCfg tcfg = sel_cfg(w.scap);
        const size_t lds_thr = ((size_t)((w.scap + 3u) & ~3u) + tcfg.bins + tcfg.list) * 4 + 48 * 8;
        const uint32_t nt = w.scap > 8192 ? kSelectThreads : 256u;
        // SCANN_HIP_THR_TIES=0 (diagnostics / tests): bound on the distance alone, whole tie groups pass
        bool thr_ties = true;
        if (const char *e = std::getenv("SCANN_HIP_THR_TIES")) thr_ties = std::atoi(e) != 0;
        // a bound in the low tail of a long sample: threshold_tail_kernel (SCANN_HIP_THR_TAIL=0: the full select)
        static const bool tail_ok = [] {
            const char *e = std::getenv("SCANN_HIP_THR_TAIL");
            return !e || std::atoi(e) != 0;
        }();
        const uint32_t J = sample_rank(w.m, w.st);
        if (tail_ok && !w.no_threshold && J <= kThrTailMaxRank && w.scap > 4096) {
            hipLaunchKernelGGL(threshold_tail_kernel, dim3(w.nq), dim3(kThrTailThreads), 0, st, w.P, w.m, w.st, w.sbase,
                               w.samp, w.scap, w.slot_of, w.thr, w.pair_thr, thr_ties ? w.vbase : nullptr);
        } else {
            SCANN_TRY(set_dyn_lds(threshold_select_kernel, lds_thr));
            hipLaunchKernelGGL(threshold_select_kernel, dim3(w.nq), dim3(nt), lds_thr, st, w.P, w.m, w.st,
                               w.no_threshold, w.sbase, w.samp, w.scap, w.slot_of, w.thr, w.pair_thr,
                               thr_ties ? w.vbase : nullptr);
        }
        LAUNCH_CHECK();
    }
    if constexpr (C::BITS == 4) {
        if (w.mfma) {
            hipLaunchKernelGGL(lut8_build_kernel, dim3(w.max_quads), dim3(256), 0, st, (uint32_t)C::S, w.lutq,
                               w.counters, w.lut8, reinterpret_cast<Lut8Meta *>(w.lut8_meta), w.pair_q, w.pair_thr,
                               w.mfma_thr1, w.mfma == 3 ? 1 : 0);
            LAUNCH_CHECK();
            // The survivors' codes travel with their positions for flat hashers: their ~12 k survivors per query
            // are spread over the whole code array (random 16-byte gathers from 16 MB: refine 145 -> 65 us at
            // C3).  In a tree index the survivors sit densely in the query's nearest leaves, the gathers hit
            // L2, and writing the codes only costs the scan (10M x 128, P = 25 / 50: step +3.5 %).
            // SCANN_HIP_MFMA_CODES: 0 never, 1 always.
            bool codes_in_list = ix.ah_mode != 0;
            if (const char *e = std::getenv("SCANN_HIP_MFMA_CODES")) codes_in_list = std::atoi(e) != 0;
            MfmaArgs ma;
            ma.thr1 = w.mfma_thr1;
            ma.pair_off = w.pair_off; ma.tile_off = w.tile_off; ma.pair_q = w.pair_q; ma.pair_vbase = w.pair_vbase;
            ma.counters = w.counters; ma.lut8 = w.lut8; ma.meta = reinterpret_cast<const Lut8Meta *>(w.lut8_meta);
            ma.pair_thr = w.pair_thr; ma.cand32_cnt = w.cand32_cnt; ma.cand32 = w.cand32; ma.cand32_codes = codes_in_list ? w.cand32_codes : nullptr; ma.cap32 = w.cap32;
            if (ev0) SCANN_HIP_CHECK(hipEventRecord(ev0, st));
            uint32_t mwgs = 4;   // workgroups per CU (4 waves each)
            if (const char *e = std::getenv("SCANN_HIP_MFMA_WGS")) mwgs = (uint32_t)std::max(1, std::atoi(e));
            // the sparse kernel's flush: lanes walk their own words on flat hashers, word-parallel on tree indexes
            // (SCANN_HIP_SP_WORDS = 0 / 1 forces a form)
            bool words = !ix.ah_mode;
            if (const char *e = std::getenv("SCANN_HIP_SP_WORDS")) words = std::atoi(e) != 0;
            const dim3 mgrid((uint32_t)cus * mwgs), mblock(kMfmaWaves * 64);
            if (w.mfma == 2)
                hipLaunchKernelGGL(adc_mfma16_kernel<C::S>, mgrid, mblock, 0, st, ix, ma);
            else if (w.mfma != 3)
                hipLaunchKernelGGL(adc_mfma_kernel<C::S>, mgrid, mblock, 0, st, ix, ma);
            else if (C::S <= 32 && !words)
                hipLaunchKernelGGL((adc_smfmac_kernel<C::S, false>), mgrid, mblock, 0, st, ix, ma);
            else if (C::S <= 32)
                hipLaunchKernelGGL((adc_smfmac_kernel<C::S, true>), mgrid, mblock, 0, st, ix, ma);
            else if (!words)
                hipLaunchKernelGGL((adc_smfmac_wide_kernel<C::S, false>), mgrid, mblock, 0, st, ix, ma);
            else
                hipLaunchKernelGGL((adc_smfmac_wide_kernel<C::S, true>), mgrid, mblock, 0, st, ix, ma);
            LAUNCH_CHECK();
            if (ev1) SCANN_HIP_CHECK(hipEventRecord(ev1, st));
            RefineArgs ra;
            ra.P = w.P; ra.cap = w.cap; ra.cap32 = w.cap32; ra.tokens = w.tokens; ra.vbase = w.vbase;
            ra.slot_of = w.slot_of; ra.lutq = w.lutq; ra.thr = w.thr; ra.cand32_cnt = w.cand32_cnt;
            ra.cand32 = w.cand32; ra.cand32_codes = codes_in_list ? w.cand32_codes : nullptr; ra.cand_cnt = w.cand_cnt; ra.cand = w.cand; ra.counters = w.counters;
            ra.allow = w.allow; ra.allow_bits = w.allow_bits;
            ra.planes = (w.mfma == 3 && ra.cand32_codes) ? 1 : 0;
            const size_t lds_rf = w.P <= kRefineTablesMax ? (size_t)w.P * C::S * 16 * sizeof(float) : 16;
            SCANN_TRY(set_dyn_lds(adc_refine_kernel<C>, lds_rf));
            hipLaunchKernelGGL(adc_refine_kernel<C>, dim3(w.nq), dim3(kRefineThreads), lds_rf, st, ix, ra);
            LAUNCH_CHECK();
            return SCANN_HIP_OK;
        }
    }
    ScanArgs a;
    a.pair_off = w.pair_off; a.tile_off = w.tile_off; a.pair_q = w.pair_q;
    a.pair_vbase = w.pair_vbase; a.counters = w.counters; a.lutq = w.lutq; a.pair_thr = w.pair_thr;
    a.cand_cnt = w.cand_cnt; a.cand = w.cand; a.cap = w.cap; a.qpt = w.qpt; a.allow = w.allow; a.allow_bits = w.allow_bits;
    if (ev0) SCANN_HIP_CHECK(hipEventRecord(ev0, st));
    uint32_t wgs = (uint32_t)cus * 8u;
    if (const char *e = std::getenv("SCANN_HIP_WGS")) wgs = (uint32_t)cus * (uint32_t)std::max(1, std::atoi(e));
    a.res_cl = w.res_cl;
    if constexpr (C::BITS == 4 && C::S <= 32) {
        if (w.resident) {
            SCANN_TRY(set_dyn_lds(adc_scan_res_kernel<C>, res_lds_bytes<C>()));
            hipLaunchKernelGGL(adc_scan_res_kernel<C>, dim3((uint32_t)cus * 4u), dim3(kResThreads),
                               res_lds_bytes<C>(), st, ix, a);
            LAUNCH_CHECK();
            if (ev1) SCANN_HIP_CHECK(hipEventRecord(ev1, st));
            return SCANN_HIP_OK;
        }
    }
    SCANN_TRY(set_dyn_lds(adc_scan_kernel<C>, scan_lds_bytes<C>()));
    hipLaunchKernelGGL(adc_scan_kernel<C>, dim3(wgs), dim3(kScanThreads), scan_lds_bytes<C>(), st, ix, a);
    LAUNCH_CHECK();
    if (ev1) SCANN_HIP_CHECK(hipEventRecord(ev1, st));
    return SCANN_HIP_OK;
}

// SearchMode::Partitioned: dense key lists (no threshold), one exact-distance tile kernel.
static int launch_exact_scan(const TxhIndexDev &ix, const TxhWork &w, hipStream_t st, hipEvent_t ev0,
                             hipEvent_t ev1) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    {   // thr = MAX for every query: select_rerank takes the k smallest of the whole stream
        const SelCfg tcfg = sel_cfg(w.scap);
        const size_t lds_thr = ((size_t)((w.scap + 3u) & ~3u) + tcfg.bins + tcfg.list) * 4 + 48 * 8;
        SCANN_TRY(set_dyn_lds(threshold_select_kernel, lds_thr));
        hipLaunchKernelGGL(threshold_select_kernel, dim3(w.nq), dim3(256), lds_thr, st, w.P, w.m, w.st, 1, w.sbase,
                           w.samp, w.scap, w.slot_of, w.thr, w.pair_thr, w.vbase);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(stream_counts_kernel, dim3(ceil_div_u32(w.nq, 256)), dim3(256), 0, st, w.nq, w.P, w.vbase,
                       w.cand_cnt);
    LAUNCH_CHECK();
    ExactScanArgs a;
    a.pair_off = w.pair_off; a.tile_off = w.tile_off; a.pair_q = w.pair_q; a.pair_vbase = w.pair_vbase;
    a.counters = w.counters; a.queries = w.queries; a.q_stride = w.q_stride; a.cand = w.cand; a.cap = w.cap;
    a.qpt = exact_quads_per_tile(ix.dim);
    const size_t lds = (size_t)a.qpt * 4 * ((ix.dim + 3u) & ~3u) * sizeof(float);
    if (ev0) SCANN_HIP_CHECK(hipEventRecord(ev0, st));
    const dim3 grid((uint32_t)cus * 8u), block(256);
    switch (ix.measure) {
        case SCANN_HIP_SQUARED_L2:
            SCANN_TRY(set_dyn_lds(leaf_exact_scan_kernel<SCANN_HIP_SQUARED_L2>, lds));
            hipLaunchKernelGGL(leaf_exact_scan_kernel<SCANN_HIP_SQUARED_L2>, grid, block, lds, st, ix, a);
            break;
        case SCANN_HIP_L2:
            SCANN_TRY(set_dyn_lds(leaf_exact_scan_kernel<SCANN_HIP_L2>, lds));
            hipLaunchKernelGGL(leaf_exact_scan_kernel<SCANN_HIP_L2>, grid, block, lds, st, ix, a);
            break;
        case SCANN_HIP_L1:
            SCANN_TRY(set_dyn_lds(leaf_exact_scan_kernel<SCANN_HIP_L1>, lds));
            hipLaunchKernelGGL(leaf_exact_scan_kernel<SCANN_HIP_L1>, grid, block, lds, st, ix, a);
            break;
        case SCANN_HIP_COSINE:
            SCANN_TRY(set_dyn_lds(leaf_exact_scan_kernel<SCANN_HIP_COSINE>, lds));
            hipLaunchKernelGGL(leaf_exact_scan_kernel<SCANN_HIP_COSINE>, grid, block, lds, st, ix, a);
            break;
        default:
            SCANN_TRY(set_dyn_lds(leaf_exact_scan_kernel<SCANN_HIP_DOT_PRODUCT>, lds));
            hipLaunchKernelGGL(leaf_exact_scan_kernel<SCANN_HIP_DOT_PRODUCT>, grid, block, lds, st, ix, a);
            break;
    }
    LAUNCH_CHECK();
    if (ev1) SCANN_HIP_CHECK(hipEventRecord(ev1, st));
    return SCANN_HIP_OK;
}

// ---- the small-batch pipeline as ONE launch ----------------------------------------------------------
// A kernel that follows another in a stream starts ~8-10 us after it (dispatch latency), which for a handful
// of queries is most of the job.  Here every workgroup (query q, 1024 stream positions) first repeats the
// query's leaf selection (select_leaves_body: the tables come from L2, the repetition costs no time), then
// scores its positions into the dense key list, and the LAST workgroup of the query to finish (a ticket
// counter behind a __threadfence) runs the finish stage.  No workgroup ever waits for another.
constexpr uint32_t kFusedChunk = kSelectThreads;   // stream positions per workgroup
constexpr uint32_t kFusedMaxWgs = 512;              // nq x workgroups per query above which three launches are used

// Wide pipeline: stream positions per group minimum for a stream of cnt keys and m wanted candidates: the smallest
// power of two that leaves at most kWideGroups groups (wide_filter_kernel holds them in registers, 16 per thread) --
// the pivot, the m-th smallest group minimum, lets about -G ln(1 - m / G) keys pass when the candidates are spread
// evenly over G groups (1.2 m at G = 3 m), more when they sit in a few leaves -- but never fewer than 1.5 m groups.
// Chosen on the device from the query's OWN stream: leaves differ in size by an order of magnitude and the host only
// knows the longest stream.
constexpr uint32_t kWideGroups = 16384;
__host__ __device__ static inline uint32_t wide_group(uint32_t cnt, uint32_t m) {
    uint32_t g = 1;
    while (g < 64 && (cnt + g - 1) / g > kWideGroups) g <<= 1;
    while (g > 1 && (uint64_t)cnt / g < ((uint64_t)m * 3) / 2) g >>= 1;
    return g;
}

struct FusedArgs {
    uint32_t L, n_pow2, p_pow2, st;
    float *cdist, *token_dists;
    uint32_t *tokens, *vbase, *sbase;   // (written by workgroup 0 of each query)
    uint32_t *tickets;                  // [nq], zero between launches
};

__global__ __launch_bounds__(kSelectThreads) void small_fused_kernel(TxhIndexDev ix, SmallArgs a, FusedArgs f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // selection | scan tables | query
    __shared__ uint32_t s_tok[kDecodeStage], s_vb[kDecodeStage + 1], s_lrow[kDecodeStage], s_last;
    const uint32_t q = blockIdx.y, b = blockIdx.x, tid = threadIdx.x, nt = kSelectThreads;
    const uint32_t P = a.P, dim = ix.dim;
    if (q == 0 && b == 0 && tid == 0) a.counters[CNT_STATUS] = 0;
    // ---- stage 1: the P nearest leaves and their key bases (every workgroup of the query)
    if (ix.ah_mode) {
        if (tid == 0) {
            s_tok[0] = 0;
            s_vb[0] = 0;
            s_vb[1] = ix.leaf_gsize[0];
            if (b == 0) {   // (ah_tokens_kernel)
                const uint32_t sz = ix.leaf_off[1] - ix.leaf_off[0];
                f.tokens[q] = 0;
                f.token_dists[q] = 0.0f;
                f.vbase[2 * q] = 0;
                f.vbase[2 * q + 1] = ix.leaf_gsize[0];
                f.sbase[3 * q] = 0;
                f.sbase[3 * q + 1] = (sz + f.st - 1) / f.st;
                f.sbase[3 * q + 2] = sz;
            }
        }
    } else {
        select_leaves_body(reinterpret_cast<uint64_t *>(s_dyn), q, b == 0, s_tok, s_vb, f.cdist, f.L, f.n_pow2, P,
                           f.p_pow2, ix.leaf_gsize, ix.leaf_off, f.st, f.tokens, f.token_dists, f.vbase, f.sbase,
                           ix.centers_t, a.queries, a.q_stride, dim, ix.centers_pitch);
    }
    __syncthreads();
    for (uint32_t r = tid; r < P; r += nt) s_lrow[r] = ix.leaf_off[s_tok[r]];
    const uint32_t cnt = min(s_vb[P], a.cap);
    const uint32_t chunk = a.chunk;   // stream positions per workgroup (<= the workgroup's threads)
    const uint32_t v0 = b * chunk;
    if (v0 >= cnt && b != 0) return;   // an idle workgroup (block-uniform; workgroup 0 always takes part)
    __syncthreads();
    // ---- stage 2: keys of the stream positions [v0, v0 + kFusedChunk)
    const uint32_t v = v0 + tid;
    const bool have = tid < chunk && v < cnt;
    uint32_t r = 0;
    if (have) {
        uint32_t lo = 0, hi = P;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_vb[mid] <= v) lo = mid; else hi = mid;
        }
        r = lo;
    }
    const uint32_t j = have ? v - s_vb[r] : 0u;
    const uint32_t csr = have ? s_lrow[r] + j : 0u;
    uint64_t *out = a.cand + (size_t)q * a.cap;
    if (ix.exact_scan) {
        float *s_qe = reinterpret_cast<float *>(s_dyn);
        for (uint32_t d = tid; d < dim; d += nt) s_qe[d] = a.queries[(size_t)q * a.q_stride + d];
        __syncthreads();
        if (have) {
            const float *row = ix.rows + (size_t)(ix.rows_csr ? csr : ix.leaf_ids[csr]) * ix.stride;
            __hip_atomic_store(&out[v], make_key(exact_pair_thread(ix.measure, dim, s_qe, row), v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (cnt) {
        const uint32_t S = ix.S, K = ix.K, dsub = ix.dsub, kp = ix.kp;
        float *s_lut = reinterpret_cast<float *>(s_dyn), *s_qr = s_lut + S * kp;
        // the leaves this workgroup's positions fall into (uniform: from the first and last position)
        uint32_t r_first = 0, r_last = 0;
        {
            const uint32_t va = v0, vz = min(v0 + chunk, cnt) - 1u;
            uint32_t lo = 0, hi = P;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_vb[mid] <= va) lo = mid; else hi = mid;
            }
            r_first = lo;
            lo = 0; hi = P;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_vb[mid] <= vz) lo = mid; else hi = mid;
            }
            r_last = lo;
        }
        const uint32_t bits = ix.code_bits, per = 32u / bits, mask = (1u << bits) - 1u, nw = ix.nw;
        for (uint32_t rr = r_first; rr <= r_last; ++rr) {
            if (s_vb[rr + 1] == s_vb[rr]) continue;   // an empty leaf (uniform)
            const uint32_t leaf = s_tok[rr];
            for (uint32_t d = tid; d < dim; d += nt) {   // residual q - centroid (mod.rs:309-316)
                float x = a.queries[(size_t)q * a.q_stride + d];
                if (ix.use_residuals) x = x - ix.centers[(size_t)leaf * dim + d];
                s_qr[d] = x;
            }
            __syncthreads();
            for (uint32_t e = tid; e < S * kp; e += nt) {   // LookupTable::from_query (lut.rs:47-70, codebook.rs:98-115)
                const uint32_t sub = e / kp, c = e - sub * kp;
                float acc = 0.0f;
                if (c < K) {
                    const float *cb = ix.codebook + ((size_t)sub * K + c) * dsub;
                    for (uint32_t d = 0; d < dsub; ++d) {
                        const float t = s_qr[sub * dsub + d] - cb[d];
                        acc = acc + t * t;
                    }
                }
                s_lut[e] = acc;
            }
            __syncthreads();
            if (have && r == rr) {
                const uint32_t *w = ix.codes + (size_t)csr * nw;
                float acc = 0.0f;   // LookupTable::compute_distance (lut.rs:74-82)
                for (uint32_t sub = 0; sub < S; ++sub) {
                    const uint32_t code = (w[sub / per] >> (bits * (sub % per))) & mask;
                    const float tv = s_lut[sub * kp + code];
                    acc = sub == 0 ? tv : acc + tv;
                }
                __hip_atomic_store(&out[v], row_allowed(ix, a.allow, a.allow_bits, csr) ? make_key(acc, v) : SCANN_KEY_MAX,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();   // (the tables are rebuilt for the next leaf)
        }
    }
    // ---- stage 3: the last workgroup of the query to get here finishes it.  The keys were stored with
    // agent scope (write-through: the XCDs' L2s are not coherent with each other inside a kernel, and an
    // agent-scope FENCE would write back the whole L2 -- tens of microseconds); every thread waits for its own
    // store, the barrier orders the workgroup, the ticket is an agent-scope atomic, and the finish stage reads
    // the list with agent-scope loads ONLY (block_select<.., AGENT> included).  This is not a release / acquire pair
    // of the memory model but the hand-off MI355X_MICROARCH.md lists as valid on gfx950 in its place ("Valid forms":
    // every byte stored sc1, every storing wave drained with s_waitcnt vmcnt(0) ahead of the barrier, ONE lane of
    // each storing workgroup adding to ONE unsharded agent-scope counter, the workgroup whose add returned last reading
    // with sc1 loads behind a workgroup barrier; 8-byte stores and loads): a hardware property, measured, not an
    // architectural guarantee -- test_small_batch_pipeline_matches_staged_pipeline runs it for every searcher kind,
    // with a restrictive allow-bitmap (the fallback select) and under uneven load.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (inline asm: the compiler may drop a builtin wait it thinks is covered)
    __syncthreads();
    if (tid == 0) {
        const uint32_t target = max(1u, (cnt + chunk - 1u) / chunk);
        const uint32_t t = __hip_atomic_fetch_add(&f.tickets[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t + 1u == target) ? 1u : 0u;
        if (s_last) __hip_atomic_store(&f.tickets[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    small_finish_body<true>(ix, a, q, reinterpret_cast<float *>(s_dyn), s_vb, s_tok);
}

// =====================================================================================
// Few queries, long streams: the wide pipeline (TxhWork::small == 2).
//
// The small-batch pipeline above finishes a query in ONE workgroup: two passes over the dense key list, then m
// row gathers -- 0.1 ms for 100 k keys and m = 1000, and pre_reorder_k is capped by the workgroup's LDS.  One
// query over the flat 1M hasher at m = 5000 (the reference's ann_benchmark operating point,
// bin/ann_benchmark.rs:172-178) fell back to the batched pipeline's twelve launches.  Here every stage is
// spread over the chip, in three launches:
//   1. wide_scan_kernel: leaf selection + scan as in small_fused_kernel; every group of g stream positions
//      (wide_group: from the query's own stream length) also leaves its minimum approximate distance.
//   2. wide_filter_kernel: up to 256 workgroups per query.  Each derives the same pivot -- an upper bound of the
//      m-th smallest group minimum (the upper edge of its bin in a 4096-bin histogram of the minima), which is
//      >= the m-th smallest key's distance (the m smallest minima belong to m distinct points) and lets
//      1.2-1.6 m keys pass -- filters its share of the key list (64-key chunks interleaved between the workgroups),
//      decodes the passing keys, scores their rows exactly (exact_pair_8lanes) and appends (key, exact, index) to
//      the query's compact arrays (one global atomic per batch of candidates).
//   3. wide_final_kernel: a workgroup per query, the entries in registers: the m-th smallest key among them (the
//      candidates of mod.rs:283-293: everything above it drops out), then the k best of those by (exact, merge
//      key) (mod.rs:342-364): a histogram shortlist ranked by one wave, or the two-level tournament of
//      small_finish_body when ties crowd the shortlist -- exact under any number of ties.
// Measured (one query, MI355X): flat 1M x 128 hasher at m = 5000: 16 + 16 + 16 us of kernels, 0.061 ms per call
// (the batched pipeline's twelve launches: 0.175); Tree-X-Hybrid 1M x 128, 1000 leaves, P = 10, m = 1000: 35 (of
// which the leaf selection 27) + 12 + 16 us, 0.074 ms per call (three-launch small pipeline: 0.124).
// Same keys, same arithmetic, same tie order as the other pipelines: rows are identical.  If the compact arrays
// overflow (thousands of points tied at the pivot's distance: a dataset of few distinct code rows) the status word
// says RESOURCE_EXHAUSTED and a host call's count row 0xFFFFFFFF; the host entry repeats the call on the batched
// pipeline.
// =====================================================================================
constexpr uint32_t kWideList = 2048;        // passing keys a workgroup of the filter collects per batch

struct WideArgs {
    uint32_t ng_stride, cap2, wgs;
    const uint32_t *mins;
    uint64_t *ckey;
    uint32_t *ceb, *cidx, *ccnt;
};

// Launch 1: leaf selection (every workgroup repeats it, as in small_fused_kernel) and the scan of
// [v0, v0 + chunk) stream positions per workgroup -- for ADC scans up to kWideRep positions per thread, so that a
// workgroup's table build (and its four dependent round trips) is shared by 4096 points and one wave of workgroups
// covers a 1M stream -- plus the group minima.
constexpr int kWideRep = 4;

__global__ __launch_bounds__(kSelectThreads) void wide_scan_kernel(TxhIndexDev ix, SmallArgs a, FusedArgs f, WideArgs wa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // selection | scan tables | query
    __shared__ uint32_t s_tok[kDecodeStage], s_vb[kDecodeStage + 1], s_lrow[kDecodeStage];
    const uint32_t q = blockIdx.y, b = blockIdx.x, tid = threadIdx.x, nt = kSelectThreads;
    const uint32_t P = a.P, dim = ix.dim;
#ifdef SCANN_WIDE_TIMING
    uint64_t ts[6];
    ts[0] = wall_clock64();
#endif
    if (q == 0 && b == 0 && tid == 0) a.counters[CNT_STATUS] = 0;
    if (b == 0 && tid == 0) wa.ccnt[q] = 0;
    if (ix.ah_mode) {
        if (tid == 0) {
            s_tok[0] = 0;
            s_vb[0] = 0;
            s_vb[1] = ix.leaf_gsize[0];
            if (b == 0) {   // (ah_tokens_kernel)
                const uint32_t sz = ix.leaf_off[1] - ix.leaf_off[0];
                f.tokens[q] = 0;
                f.token_dists[q] = 0.0f;
                f.vbase[2 * q] = 0;
                f.vbase[2 * q + 1] = ix.leaf_gsize[0];
                f.sbase[3 * q] = 0;
                f.sbase[3 * q + 1] = (sz + f.st - 1) / f.st;
                f.sbase[3 * q + 2] = sz;
            }
        }
    } else {
        select_leaves_body(reinterpret_cast<uint64_t *>(s_dyn), q, b == 0, s_tok, s_vb, f.cdist, f.L, f.n_pow2, P,
                           f.p_pow2, ix.leaf_gsize, ix.leaf_off, f.st, f.tokens, f.token_dists, f.vbase, f.sbase,
                           ix.centers_t, a.queries, a.q_stride, dim, ix.centers_pitch);
    }
    __syncthreads();
    for (uint32_t r = tid; r < P; r += nt) s_lrow[r] = ix.leaf_off[s_tok[r]];
    const uint32_t cnt = min(s_vb[P], a.cap);
    const uint32_t chunk = a.chunk;   // stream positions per workgroup: nt * rep
    const uint32_t v0 = b * chunk;
    if (v0 >= cnt) return;            // an idle workgroup (block-uniform)
    __syncthreads();
#ifdef SCANN_WIDE_TIMING
    ts[1] = wall_clock64();
#endif
    const uint32_t g = wide_group(cnt, a.m);
    uint32_t *mins = const_cast<uint32_t *>(wa.mins) + (size_t)q * wa.ng_stride;
    uint64_t *out = a.cand + (size_t)q * a.cap;
    auto leaf_of = [&](uint32_t v) {
        uint32_t lo = 0, hi = P;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_vb[mid] <= v) lo = mid; else hi = mid;
        }
        return lo;
    };
    // minimum of every group of g positions (a group lies inside one wave) -> mins[v / g]
    auto group_min = [&](uint32_t d32, uint32_t v, bool slot) {
        for (uint32_t o = 1; o < g; o <<= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)d32, (int)o);
            d32 = other < d32 ? other : d32;
        }
        if (slot && (tid & (g - 1u)) == 0 && v < cnt) mins[v / g] = d32;
    };
    const uint32_t S = ix.S, K = ix.K, dsub = ix.dsub, kp = ix.kp;
    float *s_lut = reinterpret_cast<float *>(s_dyn), *s_qr = s_lut + S * kp;
    const uint32_t vz = min(v0 + chunk, cnt);                      // end of this workgroup's positions
    const uint32_t r_first = leaf_of(v0), r_last = leaf_of(vz - 1u);
    const uint32_t bits = ix.code_bits, per = 32u / bits, mask = (1u << bits) - 1u, nw = ix.nw;
    const bool fast4 = bits == 4 && nw <= 4 && S == nw * 8u;       // whole words of eight 4-bit codes
    uint32_t d32[kWideRep];
#pragma unroll
    for (int it = 0; it < kWideRep; ++it) d32[it] = 0xFFFFFFFFu;   // (no position / rejected by the restrict filter: absent)
    for (uint32_t rr = r_first; rr <= r_last; ++rr) {
        if (s_vb[rr + 1] == s_vb[rr]) continue;   // an empty leaf (uniform)
        const uint32_t leaf = s_tok[rr];
        for (uint32_t d = tid; d < dim; d += nt) {   // residual q - centroid (mod.rs:309-316)
            float x = a.queries[(size_t)q * a.q_stride + d];
            if (ix.use_residuals) x = x - ix.centers[(size_t)leaf * dim + d];
            s_qr[d] = x;
        }
        // this leaf's share of the positions: their code words travel while the tables are built
        const uint32_t la = max(v0, s_vb[rr]), lz = min(vz, s_vb[rr + 1]);
        uint32_t cw[kWideRep][4];
        bool in[kWideRep];
#pragma unroll
        for (int it = 0; it < kWideRep; ++it) {
            const uint32_t v = v0 + (uint32_t)it * nt + tid;
            in[it] = v >= la && v < lz;
#pragma unroll
            for (int wi = 0; wi < 4; ++wi) cw[it][wi] = 0;
            if (fast4 && in[it]) {
                const uint32_t *w = ix.codes + (size_t)(s_lrow[rr] + (v - s_vb[rr])) * nw;
#pragma unroll
                for (int wi = 0; wi < 4; ++wi)
                    if ((uint32_t)wi < nw) cw[it][wi] = w[wi];
            }
        }
        __syncthreads();
        for (uint32_t e = tid; e < S * kp; e += nt) {   // LookupTable::from_query (lut.rs:47-70, codebook.rs:98-115)
            const uint32_t sub = e / kp, c = e - sub * kp;
            float acc = 0.0f;
            if (c < K) {
                const float *cb = ix.codebook + ((size_t)sub * K + c) * dsub;
                for (uint32_t d = 0; d < dsub; ++d) {
                    const float t = s_qr[sub * dsub + d] - cb[d];
                    acc = acc + t * t;
                }
            }
            s_lut[e] = acc;
        }
        __syncthreads();
#ifdef SCANN_WIDE_TIMING
        ts[2] = wall_clock64();
#endif
#pragma unroll
        for (int it = 0; it < kWideRep; ++it) {
            if (!in[it]) continue;
            const uint32_t v = v0 + (uint32_t)it * nt + tid;
            const uint32_t csr = s_lrow[rr] + (v - s_vb[rr]);
            float acc = 0.0f;   // LookupTable::compute_distance (lut.rs:74-82): 0.0 + lut[0][c0] + lut[1][c1] ...
            if (fast4) {
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    if ((uint32_t)wi >= nw) break;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float tv = s_lut[((uint32_t)wi * 8u + (uint32_t)j) * kp + ((cw[it][wi] >> (4 * j)) & 15u)];
                        acc = (wi == 0 && j == 0) ? tv : acc + tv;
                    }
                }
            } else {
                const uint32_t *w = ix.codes + (size_t)csr * nw;
                for (uint32_t sub = 0; sub < S; ++sub) {
                    const uint32_t code = (w[sub / per] >> (bits * (sub % per))) & mask;
                    const float tv = s_lut[sub * kp + code];
                    acc = sub == 0 ? tv : acc + tv;
                }
            }
            const uint64_t key = row_allowed(ix, a.allow, a.allow_bits, csr) ? make_key(acc, v) : SCANN_KEY_MAX;
            out[v] = key;
            d32[it] = (uint32_t)(key >> 32);
        }
        __syncthreads();   // (the tables are rebuilt for the next leaf)
    }
#ifdef SCANN_WIDE_TIMING
    ts[3] = wall_clock64();
#endif
#pragma unroll
    for (int it = 0; it < kWideRep; ++it) {
        if ((uint32_t)it * nt >= chunk) break;   // (uniform)
        group_min(d32[it], v0 + (uint32_t)it * nt + tid, true);
    }
#ifdef SCANN_WIDE_TIMING
    ts[4] = wall_clock64();
    if (tid == 0 && b == 0) printf("scan: leaves %.2f tables %.2f points %.2f minima %.2f us\n", (ts[1] - ts[0]) * 0.01, (ts[2] - ts[1]) * 0.01, (ts[3] - ts[2]) * 0.01, (ts[4] - ts[3]) * 0.01);
#endif
}

__global__ __launch_bounds__(kSelectThreads) void wide_filter_kernel(TxhIndexDev ix, SmallArgs a, WideArgs wa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_wdyn[];   // [dim] query
    __shared__ uint64_t s_pass[kWideList], s_slist[kSelListMax], s_red[48];
    __shared__ uint32_t s_hist[kSelBinsMax], s_eb[kWideList], s_ix[kWideList], s_dvb[kDecodeStage + 1], s_drow[kDecodeStage];
    __shared__ uint32_t s_n, s_base;
    const uint32_t q = blockIdx.y, b = blockIdx.x, tid = threadIdx.x, nt = kSelectThreads;
    const uint32_t P = a.P, m = a.m;
    const uint32_t *vbq = a.vbase + (size_t)q * (P + 1), *tokq = a.tokens + (size_t)q * P;
#ifdef SCANN_WIDE_TIMING
    uint64_t tf[6];
    tf[0] = wall_clock64();
#endif
    const uint32_t cnt = min(vbq[P], a.cap);
    if ((uint64_t)b * 64u >= cnt) return;   // no key chunk for this workgroup (block-uniform)
    float *s_q = reinterpret_cast<float *>(s_wdyn);
    for (uint32_t j = tid; j < ix.dim; j += nt) s_q[j] = a.queries[(size_t)q * a.q_stride + j];
    for (uint32_t r = tid; r <= P; r += nt) {
        s_dvb[r] = vbq[r];
        if (r < P) s_drow[r] = ix.leaf_off[tokq[r]];
    }
    if (tid == 0) s_n = 0;
    // Key positions of this workgroup: 64-key chunks c = (j * 16 + wave) * wgs + b, j = 0, 1, ... -- interleaved at
    // wave granularity, so that the near leaves' dense runs of passing keys (most of a tree query's candidates lie in
    // its first one or two leaves) are shared by all workgroups.  The first KP rounds travel while the pivot is computed.
    const uint64_t *list = a.cand + (size_t)q * a.cap;
    const uint32_t kwave = tid >> 6, klane = tid & 63u;
    auto key_pos = [&](uint32_t j) { return ((uint64_t)(j * (nt >> 6) + kwave) * wa.wgs + b) * 64u + klane; };
    constexpr int KP = 4;
    uint64_t kpre[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const uint64_t i = key_pos((uint32_t)j);
        kpre[j] = i < cnt ? list[i] : SCANN_KEY_MAX;
    }
#ifdef SCANN_WIDE_TIMING
    tf[1] = wall_clock64();
#endif
    // ---- the pivot (every workgroup of the query computes the same one): an upper bound of the m-th smallest group
    // minimum -- the upper edge of its bin in a 4096-bin histogram over [min, max] of the minima.  One read of the
    // minima (16 per thread, in registers), four barriers; the exact m-th minimum (block_select: a dozen barriers
    // of 1024 threads, 9 us) would let a few dozen keys fewer pass.
    uint32_t pivot = 0xFFFFFFFEu;
    const uint32_t g = wide_group(cnt, m), ng = (cnt + g - 1u) / g;
    if (cnt > m && ng >= m) {
        const uint32_t *mins = wa.mins + (size_t)q * wa.ng_stride;
        constexpr int RV = 16;
        if (ng <= (uint32_t)RV * nt) {
            uint32_t *s_w = reinterpret_cast<uint32_t *>(s_red);   // [0..15] min, [16..31] max, [32..47] present / scan, [48] bin
            uint32_t mv[RV];
            uint32_t vmin = 0xFFFFFFFFu, vmax = 0, present = 0;
#pragma unroll
            for (int e = 0; e < RV; ++e) {
                const uint32_t i = (uint32_t)e * nt + tid;
                mv[e] = ((uint32_t)e * nt < ng && i < ng) ? mins[i] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int e = 0; e < RV; ++e)
                if (mv[e] != 0xFFFFFFFFu) {
                    vmin = min(vmin, mv[e]);
                    vmax = max(vmax, mv[e]);
                    ++present;
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                vmin = min(vmin, (uint32_t)__shfl_xor((int)vmin, o));
                vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, o));
                present += (uint32_t)__shfl_xor((int)present, o);
            }
            const uint32_t wave = tid >> 6, lane = tid & 63u;
            if (lane == 0) {
                s_w[wave] = vmin;
                s_w[16 + wave] = vmax;
                s_w[32 + wave] = present;
            }
            for (uint32_t i = tid; i < kSelBinsMax; i += nt) s_hist[i] = 0;
            __syncthreads();
            present = 0;
            for (uint32_t w2 = 0; w2 < (nt >> 6); ++w2) {
                vmin = min(vmin, s_w[w2]);
                vmax = max(vmax, s_w[16 + w2]);
                present += s_w[32 + w2];
            }
            __syncthreads();   // (s_w[32..] is reused by the scan)
            if (present >= m) {   // block-uniform (else: fewer than m groups hold an allowed point, everything passes)
                const uint32_t range = vmax - vmin;
                uint32_t sh = 0;
                while ((range >> sh) >= kSelBinsMax) ++sh;
#pragma unroll
                for (int e = 0; e < RV; ++e)
                    if (mv[e] != 0xFFFFFFFFu) atomicAdd(&s_hist[(mv[e] - vmin) >> sh], 1u);
                __syncthreads();
                const uint32_t bin = block_hist_rank_bin(s_hist, s_w, m);
                const uint64_t hi = (uint64_t)vmin + (((uint64_t)bin + 1u) << sh) - 1u;
                pivot = hi > vmax ? vmax : (uint32_t)hi;
                if (pivot == 0xFFFFFFFFu) pivot = 0xFFFFFFFEu;
            }
        } else {   // very long streams: the exact select, from L2
            __syncthreads();
            const uint32_t pv = block_select<uint32_t>(mins, ng, m, sel_cfg(ng), s_hist, reinterpret_cast<uint32_t *>(s_slist), s_red);
            if (pv != 0xFFFFFFFFu) pivot = pv;   // (fewer than m groups hold an allowed point: everything passes)
        }
    }
    __syncthreads();
#ifdef SCANN_WIDE_TIMING
    tf[2] = wall_clock64();
#endif
    // a batch of collected keys: decode, exact distance (8 lanes per candidate), append to the compact arrays
    auto flush = [&]() {
        const uint32_t n = min(s_n, kWideList);
        for (uint32_t c0 = 0; c0 < n; c0 += nt / 8) {
            const uint32_t c = c0 + (tid >> 3), lane8 = tid & 7u;
            const bool act = c < n;
            uint32_t idx = 0, rowi = 0;
            uint64_t key = 0;
            if (act) {
                key = s_pass[c];
                const uint32_t vpos = (uint32_t)key;
                uint32_t lo = 0, hi = P;
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_dvb[mid] <= vpos) lo = mid; else hi = mid;
                }
                const uint32_t csr = s_drow[lo] + (vpos - s_dvb[lo]);
                idx = ix.leaf_ids ? ix.leaf_ids[csr] : csr;
                rowi = ix.rows_csr ? csr : idx;
            }
            float r = 0.0f;
            if (a.exact_reorder) r = exact_pair_8lanes<16>(ix, s_q, ix.rows + (size_t)rowi * ix.stride, act, lane8);
            if (act && lane8 == 0) {
                s_ix[c] = idx;
                s_eb[c] = a.exact_reorder ? f32_to_ordered(r) : (uint32_t)(key >> 32);
            }
        }
        if (tid == 0) s_base = atomicAdd(&wa.ccnt[q], n);
        __syncthreads();
        const uint32_t base = s_base;
        if (tid == 0 && base + n > wa.cap2) atomicExch(&a.counters[CNT_STATUS], (uint32_t)SCANN_HIP_RESOURCE_EXHAUSTED);
        for (uint32_t i = tid; i < n; i += nt)
            if (base + i < wa.cap2) {
                const size_t e = (size_t)q * wa.cap2 + base + i;
                wa.ckey[e] = s_pass[i];
                wa.ceb[e] = s_eb[i];
                wa.cidx[e] = s_ix[i];
            }
        __syncthreads();
        if (tid == 0) s_n = 0;
        __syncthreads();
    };
    // ---- rounds of 16 chunks (one per wave)
    for (uint32_t jb = 0; (uint64_t)jb * (nt >> 6) * wa.wgs * 64u < cnt; ++jb) {
        const uint64_t i = key_pos(jb);
        uint64_t key = SCANN_KEY_MAX;
        if (jb < (uint32_t)KP) {
#pragma unroll
            for (int j = 0; j < KP; ++j)
                if (jb == (uint32_t)j) key = kpre[j];
        } else if (i < cnt) {
            key = list[i];
        }
        const bool keep = key != SCANN_KEY_MAX && (uint32_t)(key >> 32) <= pivot;
        uint32_t wtot;
        const uint32_t wpre = wave_prefix_count(keep, &wtot);
        uint32_t base = 0;
        if ((tid & 63u) == 0 && wtot) base = atomicAdd(&s_n, wtot);
        base = (uint32_t)__shfl((int)base, 0);
        if (keep) s_pass[base + wpre] = key;   // (s_n <= kWideList - nt before the block: never past the end)
        __syncthreads();
        const uint32_t collected = s_n;
        __syncthreads();                       // (everyone has read it before the next block adds to it)
        if (collected > kWideList - nt) flush();
    }
    __syncthreads();
#ifdef SCANN_WIDE_TIMING
    tf[3] = wall_clock64();
    const uint32_t npass = s_n;
#endif
    if (s_n) flush();
#ifdef SCANN_WIDE_TIMING
    tf[4] = wall_clock64();
    if (tid == 0 && b == 0) printf("filter: cnt %u pass %u setup %.2f pivot %.2f filter %.2f flush %.2f us\n", cnt, npass, (tf[1] - tf[0]) * 0.01, (tf[2] - tf[1]) * 0.01, (tf[3] - tf[2]) * 0.01, (tf[4] - tf[3]) * 0.01);
#endif
}

__global__ __launch_bounds__(kSelectThreads) void wide_final_kernel(SmallArgs a, WideArgs wa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_fdyn[];   // u64[cap2]
    __shared__ uint64_t s_slist[kSelListMax], s_red[48];
    __shared__ uint32_t s_hist[kSelBinsMax], s_ce[64], s_cs[64], s_cn;
    __shared__ uint64_t s_ck[64];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, nt = kSelectThreads, wave = tid >> 6, lane = tid & 63u;
    const uint32_t m = a.m, k = a.k;
    uint64_t *s_v = reinterpret_cast<uint64_t *>(s_fdyn);
    const bool polled = a.done != nullptr;
    auto out_store = [&](auto *p, auto v) {
        if (polled) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else *p = v;
    };
#ifdef SCANN_WIDE_TIMING
    uint64_t tq[8];
    int tqi = 0;
#define WIDE_T() do { __syncthreads(); tq[tqi++] = wall_clock64(); } while (0)
#else
#define WIDE_T() do { } while (0)
#endif
    WIDE_T();
    const uint32_t appended = wa.ccnt[q];
    const bool overflow = appended > wa.cap2;
    const uint32_t c2 = min(appended, wa.cap2);
    const uint64_t *ckey = wa.ckey + (size_t)q * wa.cap2;
    const uint32_t *ceb = wa.ceb + (size_t)q * wa.cap2, *cidx = wa.cidx + (size_t)q * wa.cap2;
    const uint32_t nsel = min(c2, m), nout = overflow ? 0u : min(k, nsel);
    if (nout) {
        // entry e * nt + tid of the compact arrays lives in this lane's registers (cap2 <= 16 * nt)
        constexpr int E0 = 16;
        uint64_t kk[E0];
        uint32_t eb[E0];
#pragma unroll
        for (int e = 0; e < E0; ++e) {
            const uint32_t i = (uint32_t)e * nt + tid;
            kk[e] = SCANN_KEY_MAX;
            eb[e] = 0xFFFFFFFFu;
            if ((uint32_t)e * nt < c2 && i < c2) {
                kk[e] = ckey[i];
                eb[e] = ceb[i];
            }
        }
        WIDE_T();
        // the candidates: the m smallest keys (mod.rs:283-293); everything above T drops out
        uint64_t T = SCANN_KEY_MAX - 1;
        if (c2 > m) {
            // block_select's scheme on the register copy (six barriers instead of a dozen): histogram of the keys over
            // [min, max], the bin of rank m, its members ranked exactly by counting
            uint32_t *s_w = reinterpret_cast<uint32_t *>(s_red);
            uint64_t kmin = SCANN_KEY_MAX, kmax = 0;
#pragma unroll
            for (int e = 0; e < E0; ++e)
                if ((uint32_t)e * nt < c2 && kk[e] != SCANN_KEY_MAX) {
                    kmin = kk[e] < kmin ? kk[e] : kmin;
                    kmax = kk[e] > kmax ? kk[e] : kmax;
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint64_t x = shfl_xor_t<uint64_t>(kmin, o), y = shfl_xor_t<uint64_t>(kmax, o);
                kmin = x < kmin ? x : kmin;
                kmax = y > kmax ? y : kmax;
            }
            if (lane == 0) {
                s_red[wave] = kmin;
                s_red[16 + wave] = kmax;
            }
            for (uint32_t i = tid; i < kSelBinsMax; i += nt) s_hist[i] = 0;
            if (tid == 0) s_cn = 0;
            __syncthreads();
            for (uint32_t w2 = 0; w2 < (nt >> 6); ++w2) {
                kmin = s_red[w2] < kmin ? s_red[w2] : kmin;
                kmax = s_red[16 + w2] > kmax ? s_red[16 + w2] : kmax;
            }
            uint32_t ksh = 0;
            while (((kmax - kmin) >> ksh) >= (uint64_t)kSelBinsMax) ++ksh;
#pragma unroll
            for (int e = 0; e < E0; ++e)
                if ((uint32_t)e * nt < c2 && kk[e] != SCANN_KEY_MAX) atomicAdd(&s_hist[(uint32_t)((kk[e] - kmin) >> ksh)], 1u);
            __syncthreads();   // (everyone has read the wave minima / maxima: the scan reuses their words)
            const uint32_t kbin = block_hist_rank_bin(s_hist, s_w, m);
            const uint32_t krk = s_w[49], kpop = s_hist[kbin];
            const uint64_t klo = kmin + ((uint64_t)kbin << ksh);
            uint64_t khi = klo + (((uint64_t)1 << ksh) - 1);
            if (khi > kmax || khi < klo) khi = kmax;
            if (ksh == 0) {
                T = klo;
            } else if (kpop <= kSelListMax) {
#pragma unroll
                for (int e = 0; e < E0; ++e)
                    if ((uint32_t)e * nt < c2 && kk[e] >= klo && kk[e] <= khi) s_slist[atomicAdd(&s_cn, 1u)] = kk[e];
                __syncthreads();
                for (uint32_t i = tid; i < kpop; i += nt) {
                    const uint64_t v = s_slist[i];
                    uint32_t r = 0;
                    for (uint32_t j2 = 0; j2 < kpop; ++j2) r += s_slist[j2] < v ? 1u : 0u;   // (keys are unique)
                    if (r + 1 == krk) s_red[40] = v;
                }
                __syncthreads();
                T = s_red[40];
            } else {   // a crowded bin: the general select on an LDS copy
#pragma unroll
                for (int e = 0; e < E0; ++e)
                    if ((uint32_t)e * nt < c2 && (uint32_t)e * nt + tid < c2) s_v[(uint32_t)e * nt + tid] = kk[e];
                __syncthreads();
                T = block_select<uint64_t>(s_v, c2, m, sel_cfg(c2), s_hist, s_slist, s_red);
            }
            __syncthreads();
        }
        WIDE_T();
#pragma unroll
        for (int e = 0; e < E0; ++e)
            if (kk[e] > T) {
                kk[e] = SCANN_KEY_MAX;
                eb[e] = 0xFFFFFFFFu;
            }
        // the nout best by (exact, merge key) (mod.rs:342-364).  The entries are in registers: a 4096-bin histogram of
        // the candidates' exact distances over [min, max] gives the bin of rank nout; the entries up to that bin's
        // upper edge (nout and a few more: the low tail is sparse) go to a short list, which one wave ranks by
        // counting.  More than 64 of them (heavy ties): the two-level tournament of small_finish_body instead.
        uint32_t *s_w = reinterpret_cast<uint32_t *>(s_red);
        uint32_t emin = 0xFFFFFFFFu, emax = 0;
#pragma unroll
        for (int e = 0; e < E0; ++e)
            if ((uint32_t)e * nt < c2 && kk[e] != SCANN_KEY_MAX) {
                emin = min(emin, eb[e]);
                emax = max(emax, eb[e]);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            emin = min(emin, (uint32_t)__shfl_xor((int)emin, o));
            emax = max(emax, (uint32_t)__shfl_xor((int)emax, o));
        }
        if (lane == 0) {
            s_w[wave] = emin;
            s_w[16 + wave] = emax;
        }
        for (uint32_t i = tid; i < kSelBinsMax; i += nt) s_hist[i] = 0;
        if (tid == 0) s_cn = 0;
        __syncthreads();
        for (uint32_t w2 = 0; w2 < (nt >> 6); ++w2) {
            emin = min(emin, s_w[w2]);
            emax = max(emax, s_w[16 + w2]);
        }
        uint32_t esh = 0;
        while (((emax - emin) >> esh) >= kSelBinsMax) ++esh;
#pragma unroll
        for (int e = 0; e < E0; ++e)
            if ((uint32_t)e * nt < c2 && kk[e] != SCANN_KEY_MAX) atomicAdd(&s_hist[(eb[e] - emin) >> esh], 1u);
        __syncthreads();
        const uint32_t ebin = block_hist_rank_bin(s_hist, s_w, nout);
        const uint64_t ehi64 = (uint64_t)emin + (((uint64_t)ebin + 1u) << esh) - 1u;
        const uint32_t ehi = ehi64 > emax ? emax : (uint32_t)ehi64;
#pragma unroll
        for (int e = 0; e < E0; ++e)
            if ((uint32_t)e * nt < c2 && kk[e] != SCANN_KEY_MAX && eb[e] <= ehi) {
                const uint32_t pos = atomicAdd(&s_cn, 1u);
                if (pos < 64u) {
                    s_ce[pos] = eb[e];
                    s_ck[pos] = kk[e];
                    s_cs[pos] = (uint32_t)e * nt + tid;
                }
            }
        __syncthreads();
        const uint32_t cn = s_cn;   // >= nout
        WIDE_T();
        if (cn <= 64u) {
            if (tid < 64) {
                const bool have = tid < cn;
                const uint32_t ce = have ? s_ce[tid] : 0xFFFFFFFFu;
                const uint64_t ck = have ? s_ck[tid] : SCANN_KEY_MAX;
                uint32_t rank = 0;
                for (uint32_t j = 0; j < cn; ++j) {
                    const uint32_t oe = (uint32_t)__shfl((int)ce, (int)j);
                    const uint64_t ok = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(ck >> 32), (int)j) << 32) |
                                        (uint32_t)__shfl((int)(uint32_t)ck, (int)j);
                    rank += (oe < ce || (oe == ce && ok < ck)) ? 1u : 0u;
                }
                if (have && rank < nout) {
                    out_store(&a.out_idx[(size_t)q * k + rank], cidx[s_cs[tid]]);
                    out_store(&a.out_dist[(size_t)q * k + rank], ordered_to_f32(ce));
                }
            }
        } else {
        __syncthreads();   // (the histogram and the select's list are reused below)
        uint32_t *f_eb = s_hist, *f_sl = s_hist + kSelectThreads;
        uint64_t *f_kk = s_slist;
        for (uint32_t r0 = 0; r0 < nout; ++r0) {
            uint32_t b_eb = 0xFFFFFFFFu, b_sl = 0xFFFFFFFFu;
            uint64_t b_kk = SCANN_KEY_MAX;
#pragma unroll
            for (int e = 0; e < E0; ++e)
                if ((uint32_t)e * nt < c2 && kk[e] != SCANN_KEY_MAX && (eb[e] < b_eb || (eb[e] == b_eb && kk[e] < b_kk))) {
                    b_eb = eb[e];
                    b_kk = kk[e];
                    b_sl = (uint32_t)e * nt + tid;
                }
            wave_argmin96(b_eb, b_kk, b_sl);
            if (b_kk == SCANN_KEY_MAX) b_sl = 0xFFFFFFFFu;   // (the wave's pool is empty)
#pragma unroll
            for (int e = 0; e < E0; ++e)
                if ((uint32_t)e * nt < c2 && (uint32_t)e * nt + tid == b_sl) {
                    kk[e] = SCANN_KEY_MAX;
                    eb[e] = 0xFFFFFFFFu;
                }
            if (lane == 0) {
                f_eb[wave * nout + r0] = b_eb;
                f_kk[wave * nout + r0] = b_kk;
                f_sl[wave * nout + r0] = b_sl;
            }
        }
        __syncthreads();
        if (tid < 64) {
            constexpr int E = kSelectThreads / 64;
            const uint32_t nfin = (nt >> 6) * nout;           // <= 16 * 64
            uint32_t feb[E], fsl[E];
            uint64_t fkk[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t j = (uint32_t)e * 64u + tid;
                const bool ok = j < nfin && f_sl[j] != 0xFFFFFFFFu;
                feb[e] = ok ? f_eb[j] : 0xFFFFFFFFu;
                fkk[e] = ok ? f_kk[j] : SCANN_KEY_MAX;
                fsl[e] = ok ? f_sl[j] : 0xFFFFFFFFu;
            }
            uint32_t my_eb = 0xFFFFFFFFu, my_sl = 0xFFFFFFFFu;   // lane r0 keeps round r0's winner
            for (uint32_t r0 = 0; r0 < nout; ++r0) {
                uint32_t b_eb = 0xFFFFFFFFu, b_sl = 0xFFFFFFFFu;
                uint64_t b_kk = SCANN_KEY_MAX;
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((uint32_t)e * 64u < nfin && fsl[e] != 0xFFFFFFFFu &&
                        (feb[e] < b_eb || (feb[e] == b_eb && fkk[e] < b_kk))) {
                        b_eb = feb[e];
                        b_kk = fkk[e];
                        b_sl = fsl[e];
                    }
                wave_argmin96(b_eb, b_kk, b_sl);
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((uint32_t)e * 64u < nfin && fsl[e] == b_sl) {   // (entry slots are unique: exactly one lane, one entry)
                        feb[e] = 0xFFFFFFFFu;
                        fkk[e] = SCANN_KEY_MAX;
                        fsl[e] = 0xFFFFFFFFu;
                    }
                if (tid == r0) {
                    my_eb = b_eb;
                    my_sl = b_sl;
                }
            }
            if (tid < nout) {   // the winners' datapoint indices: one parallel round trip
                out_store(&a.out_idx[(size_t)q * k + tid], cidx[my_sl]);
                out_store(&a.out_dist[(size_t)q * k + tid], ordered_to_f32(my_eb));
            }
        }
        }
    }
    WIDE_T();
#ifdef SCANN_WIDE_TIMING
    if (tid == 0 && nout) printf("final: c2 %u load %.2f select %.2f shortlist %.2f rank %.2f us\n", c2, (tq[1] - tq[0]) * 0.01, (tq[2] - tq[1]) * 0.01, (tq[3] - tq[2]) * 0.01, (tq[4] - tq[3]) * 0.01);
#endif
    for (uint32_t i = nout + tid; i < k; i += nt) {
        out_store(&a.out_idx[(size_t)q * k + i], kInvalid);
        out_store(&a.out_dist[(size_t)q * k + i], __builtin_inff());
    }
    if (tid == 0) out_store(&a.out_count[q], overflow ? (polled ? 0xFFFFFFFFu : 0u) : nout);   // (host calls: repeat on the batched pipeline)
    if (a.done) {   // (see small_finish_body)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&a.done[q], a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the three-launch pipeline for small batches (see "Small batches" above)
static int launch_search_small(const TxhIndexDev &ix, const TxhWork &w, hipStream_t st, hipEvent_t ev0,
                               hipEvent_t ev1) {
    if (w.small == 2) {   // the wide few-query pipeline ("Few queries, long streams")
        // 1024 x rep stream positions per workgroup, rep <= kWideRep chosen so that one wave of workgroups (256 CUs) covers
        // the stream.  (ADC scans only: an exact scan wants a row per thread and ~128 rows per workgroup -- hundreds of
        // workgroups per query that would each repeat the leaf selection; Partitioned mode keeps the pipelines above.)
        const uint32_t chunk = kFusedChunk * std::min<uint32_t>(kWideRep, std::max(1u, ceil_div_u32(w.cap, 256u * kFusedChunk)));
        const uint32_t G = std::max(1u, ceil_div_u32(w.cap, chunk));
        SmallArgs a;
        a.nq = w.nq; a.P = w.P; a.m = w.m; a.k = w.k; a.cap = w.cap; a.q_stride = w.q_stride;
        a.exact_reorder = w.exact_reorder; a.queries = w.queries; a.tokens = w.tokens; a.vbase = w.vbase;
        a.cand = w.cand; a.counters = w.counters; a.allow = w.allow; a.allow_bits = w.allow_bits;
        a.out_idx = w.out_idx; a.out_dist = w.out_dist; a.out_count = w.out_count;
        a.done = w.small_done; a.seq = w.small_seq; a.chunk = chunk;
        FusedArgs f;
        f.L = ix.L; f.n_pow2 = next_pow2_u32(ix.L);
        f.p_pow2 = (w.P * 4u <= f.n_pow2) ? next_pow2_u32(std::max(1u, w.P)) : 0u;
        f.st = w.st; f.cdist = w.cdist; f.token_dists = w.token_dists; f.tokens = w.tokens; f.vbase = w.vbase;
        f.sbase = w.sbase; f.tickets = nullptr;
        WideArgs wa;
        wa.ng_stride = w.cap; wa.cap2 = w.wide_cap2;
        wa.wgs = std::min(256u, std::max(8u, ceil_div_u32(w.cap, 1024u)));
        wa.mins = w.wide_min; wa.ckey = w.wide_ckey; wa.ceb = w.wide_ceb; wa.cidx = w.wide_cidx; wa.ccnt = w.wide_cnt;
        const SelCfg lcfg = sel_cfg(ix.L);
        const size_t lds_sel = ix.ah_mode ? 0 : (size_t)(f.n_pow2 + f.p_pow2) * sizeof(uint64_t) + (size_t)lcfg.bins * 4 +
                                                (size_t)lcfg.list * 8 + 48 * 8 + 64 * 4 + (size_t)ix.L * 8 +
                                                (size_t)((ix.dim + 3u) & ~3u) * 4 + 16;
        const size_t lds_scan = ((size_t)ix.S * ix.kp + ix.dim) * sizeof(float);
        SCANN_TRY(set_dyn_lds_with_static(wide_scan_kernel, std::max(lds_sel, lds_scan), 8 * 1024));   // (6160 B of static arrays)
        if (ev0) SCANN_HIP_CHECK(hipEventRecord(ev0, st));
        hipLaunchKernelGGL(wide_scan_kernel, dim3(G, w.nq), dim3(kSelectThreads), std::max(lds_sel, lds_scan), st, ix, a, f, wa);
        LAUNCH_CHECK();
        if (ev1) SCANN_HIP_CHECK(hipEventRecord(ev1, st));
        const size_t lds_f = (size_t)((ix.dim + 3u) & ~3u) * 4;
        SCANN_TRY(set_dyn_lds_with_static(wide_filter_kernel, lds_f, 62 * 1024));   // (61840 B of static arrays)
        hipLaunchKernelGGL(wide_filter_kernel, dim3(wa.wgs, w.nq), dim3(kSelectThreads), lds_f, st, ix, a, wa);
        LAUNCH_CHECK();
        const size_t lds_l = (size_t)w.wide_cap2 * 8;
        SCANN_TRY(set_dyn_lds_with_static(wide_final_kernel, lds_l, 27 * 1024));   // (26 KB of static arrays)
        hipLaunchKernelGGL(wide_final_kernel, dim3(w.nq), dim3(kSelectThreads), lds_l, st, a, wa);
        LAUNCH_CHECK();
        return SCANN_HIP_OK;
    }
    {   // one launch when the grid stays small (SCANN_HIP_FUSED=0: always three)
        // Exact scans read a whole row per thread, 64 cache lines per wave instruction: they are bound by the
        // L1's line-request rate of ONE compute unit, so their workgroups take only 128 positions each (more
        // compute units share the rows); ADC scans read 16 coalesced code bytes per point and take 1024 --
        // but rebuild the tables per leaf one after the other, so tree indexes with several leaves per
        // query keep the three-launch form, whose scan builds every leaf's table in its own workgroup.
        uint32_t chunk = kFusedChunk;
        if (ix.exact_scan) chunk = std::min(kFusedChunk, std::max(128u, (ceil_div_u32(w.cap, 256u) + 127u) & ~127u));
        const uint32_t G = std::max(1u, ceil_div_u32(w.cap, chunk));
        bool fused = w.small_tickets && w.P <= kDecodeStage && (uint64_t)w.nq * G <= kFusedMaxWgs &&
                     (ix.exact_scan || w.P == 1);
        if (const char *e = std::getenv("SCANN_HIP_FUSED")) fused = fused && std::atoi(e) != 0;
        if (fused) {
            SmallArgs a;
            a.nq = w.nq; a.P = w.P; a.m = w.m; a.k = w.k; a.cap = w.cap; a.q_stride = w.q_stride;
            a.exact_reorder = w.exact_reorder; a.queries = w.queries; a.tokens = w.tokens; a.vbase = w.vbase;
            a.cand = w.cand; a.counters = w.counters; a.allow = w.allow; a.allow_bits = w.allow_bits;
            a.out_idx = w.out_idx; a.out_dist = w.out_dist; a.out_count = w.out_count;
            a.done = w.small_done; a.seq = w.small_seq; a.chunk = chunk;
            FusedArgs f;
            f.L = ix.L; f.n_pow2 = next_pow2_u32(ix.L);
            f.p_pow2 = (w.P * 4u <= f.n_pow2) ? next_pow2_u32(std::max(1u, w.P)) : 0u;
            f.st = w.st; f.cdist = w.cdist; f.token_dists = w.token_dists; f.tokens = w.tokens; f.vbase = w.vbase;
            f.sbase = w.sbase; f.tickets = w.small_tickets;
            const SelCfg lcfg = sel_cfg(ix.L);
            const size_t lds_sel = ix.ah_mode ? 0 : (size_t)(f.n_pow2 + f.p_pow2) * sizeof(uint64_t) + (size_t)lcfg.bins * 4 +
                                                    (size_t)lcfg.list * 8 + 48 * 8 + 64 * 4 + (size_t)ix.L * 8 +
                                                    (size_t)((ix.dim + 3u) & ~3u) * 4 + 16;
            const size_t lds_scan = ((size_t)(ix.exact_scan ? 0u : ix.S * ix.kp) + ix.dim) * sizeof(float);
            const size_t lds = std::max(lds_sel, lds_scan);
            // (80272 B of static arrays -- the finish stage's: from ~2200 leaves the selection's dynamic part passes 64 KB, and
            // the attribute must leave room for both)
            SCANN_TRY(set_dyn_lds_with_static(small_fused_kernel, lds, 79 * 1024));
            if (ev0) SCANN_HIP_CHECK(hipEventRecord(ev0, st));
            hipLaunchKernelGGL(small_fused_kernel, dim3(G, w.nq), dim3(kSelectThreads), lds, st, ix, a, f);
            LAUNCH_CHECK();
            if (ev1) SCANN_HIP_CHECK(hipEventRecord(ev1, st));
            return SCANN_HIP_OK;
        }
    }
    if (ix.ah_mode) {
        hipLaunchKernelGGL(ah_tokens_kernel, dim3(ceil_div_u32(w.nq, 256)), dim3(256), 0, st, w.nq,
                           ix.leaf_gsize, ix.leaf_off, w.st, w.tokens, w.token_dists, w.vbase, w.sbase);
    } else {
        const uint32_t n2 = next_pow2_u32(ix.L);
        const uint32_t p2 = (w.P * 4u <= n2) ? next_pow2_u32(std::max(1u, w.P)) : 0u;
        const SelCfg lcfg = sel_cfg(ix.L);
        // (+ the leaf size tables and the query: the kernel's inline mode)
        const size_t lds2 = (size_t)(n2 + p2) * sizeof(uint64_t) + (size_t)lcfg.bins * 4 + (size_t)lcfg.list * 8 +
                            48 * 8 + 64 * 4 + (size_t)ix.L * 8 + (size_t)((ix.dim + 3u) & ~3u) * 4 + 16;
        SCANN_TRY(set_dyn_lds(select_leaves_kernel, lds2));
        hipLaunchKernelGGL(select_leaves_kernel, dim3(w.nq), dim3(kSelectThreads), lds2, st, w.cdist, ix.L, n2, w.P, p2,
                           ix.leaf_gsize, ix.leaf_off, w.st, w.tokens, w.token_dists, w.vbase, w.sbase, ix.centers_t,
                           w.queries, w.q_stride, ix.dim, ix.centers_pitch);
    }
    LAUNCH_CHECK();
    SmallArgs a;
    a.nq = w.nq; a.P = w.P; a.m = w.m; a.k = w.k; a.cap = w.cap; a.q_stride = w.q_stride;
    a.exact_reorder = w.exact_reorder; a.queries = w.queries; a.tokens = w.tokens; a.vbase = w.vbase;
    a.cand = w.cand; a.counters = w.counters; a.allow = w.allow; a.allow_bits = w.allow_bits;
    a.out_idx = w.out_idx; a.out_dist = w.out_dist; a.out_count = w.out_count;
    a.done = w.small_done; a.seq = w.small_seq;
    // exact scans read a whole row per point (one row per thread keeps every load in flight at once); the
    // ADC scan amortises the table build of its workgroup over four points per thread
    a.chunk = ix.exact_scan ? 256u : kSmallChunk;
    const size_t lds_scan = ((size_t)(ix.exact_scan ? 0u : ix.S * ix.kp) + ix.dim) * sizeof(float);
    SCANN_TRY(set_dyn_lds(small_scan_kernel, lds_scan));
    if (ev0) SCANN_HIP_CHECK(hipEventRecord(ev0, st));
    hipLaunchKernelGGL(small_scan_kernel, dim3(ceil_div_u32(w.small_max_leaf, a.chunk), w.nq * w.P), dim3(256),
                       lds_scan, st, ix, a);
    LAUNCH_CHECK();
    if (ev1) SCANN_HIP_CHECK(hipEventRecord(ev1, st));
    const size_t lds_fin = (size_t)ix.dim * sizeof(float);
    SCANN_TRY(set_dyn_lds(small_finish_kernel, lds_fin));
    hipLaunchKernelGGL(small_finish_kernel, dim3(w.nq), dim3(kSelectThreads), lds_fin, st, ix, a);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int txh_launch_search(const TxhIndexDev &ix, const TxhWork &w, bool local_only, hipStream_t st,
                      hipEvent_t ev0, hipEvent_t ev1) {
    if (w.nq == 0) return SCANN_HIP_OK;
    if (w.small && !local_only && !w.need_sorted_cands) return launch_search_small(ix, w, st, ev0, ev1);
    const uint32_t wl_tp = ix.exact_scan ? kExactRows : w.mfma ? kMfmaRange : w.resident ? kResThreads * kScanPPT : scan_tile_points(ix);
    const uint32_t wl_qpt = ix.exact_scan ? exact_quads_per_tile(ix.dim) : w.mfma == 2 ? 4u : w.mfma ? 8u : w.resident ? kResQuads : w.qpt;
    const uint32_t wl_cpt = (w.resident && !ix.exact_scan && !w.mfma) ? w.res_cl : 1u;
    if (ix.ah_mode && ix.L == 1 && w.P == 1) {
        AhSetupArgs h;
        h.nq = w.nq; h.max_slots = w.max_slots; h.st = w.st; h.tp = wl_tp; h.quads_per_tile = wl_qpt;
        h.chunks_per_tile = wl_cpt; h.stp = scan_tile_points(ix); h.squads_per_tile = w.sqpt;
        h.leaf_gsize = ix.leaf_gsize; h.leaf_off = ix.leaf_off; h.leaf_cnt = w.leaf_cnt; h.leaf_cursor = w.leaf_cursor;
        h.counters = w.counters; h.cand_cnt = w.cand_cnt; h.cand32_cnt = w.mfma ? w.cand32_cnt : nullptr;
        h.pair_q = w.pair_q; h.pair_leaf = w.pair_leaf; h.pair_vbase = w.pair_vbase; h.pair_sbase = w.pair_sbase;
        h.slot_of = w.slot_of; h.tokens = w.tokens; h.vbase = w.vbase; h.sbase = w.sbase; h.pair_off = w.pair_off;
        h.tile_off = w.tile_off; h.stile_off = w.stile_off; h.token_dists = w.token_dists;
        hipLaunchKernelGGL(ah_setup_kernel, dim3(1), dim3(1024), 0, st, h);
        LAUNCH_CHECK();
    } else {
    {
        const uint32_t work = std::max(std::max(ix.L, w.nq), w.max_slots);
        hipLaunchKernelGGL(txh_init_kernel, dim3(std::min(1024u, ceil_div_u32(work, 256))), dim3(256), 0, st,
                           ix.L, w.nq, w.max_slots, w.leaf_cnt, w.leaf_cursor, w.counters, w.cand_cnt,
                           w.mfma ? w.cand32_cnt : nullptr, w.pair_q);
        LAUNCH_CHECK();
    }
    SCANN_TRY(launch_partition_stage(ix, w, st));

    const uint32_t npairs = w.nq * w.P;
    hipLaunchKernelGGL(worklist_count_kernel, dim3(ceil_div_u32(npairs, 256)), dim3(256), 0, st,
                       npairs, ix.ah_mode, w.tokens, ix.leaf_off, w.leaf_cnt);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(worklist_scan_kernel, dim3(1), dim3(1024), 0, st, ix.L, w.leaf_cnt,
                       ix.leaf_off, wl_tp, wl_qpt, wl_cpt, scan_tile_points(ix), w.st,
                       w.sqpt, w.pair_off, w.tile_off,
                       w.stile_off, w.counters);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(worklist_fill_kernel, dim3(ceil_div_u32(npairs, 256)), dim3(256), 0, st, w.nq,
                       w.P, ix.ah_mode, w.tokens, w.vbase, w.sbase, ix.leaf_off, w.pair_off, w.leaf_cursor, w.pair_q,
                       w.pair_leaf, w.pair_vbase, w.pair_sbase, w.slot_of);
    LAUNCH_CHECK();
    }
    if (ix.exact_scan) {
        SCANN_TRY(launch_exact_scan(ix, w, st, ev0, ev1));
    } else {
    const size_t lds_lut = (size_t)4 * ix.dim * sizeof(float);
    SCANN_TRY(set_dyn_lds(lut_build_kernel, lds_lut));
    hipLaunchKernelGGL(lut_build_kernel, dim3(w.max_quads), dim3(256), lds_lut, st, ix, w.queries,
                       w.q_stride, w.pair_q, w.pair_leaf, w.counters, w.lutq);
    LAUNCH_CHECK();

    switch (ix.code_bits * 1000 + ix.S) {
        case 4008: SCANN_TRY((launch_scan_stages<Codec<8, 4>>(ix, w, st, ev0, ev1))); break;
        case 4016: SCANN_TRY((launch_scan_stages<Codec<16, 4>>(ix, w, st, ev0, ev1))); break;
        case 4024: SCANN_TRY((launch_scan_stages<Codec<24, 4>>(ix, w, st, ev0, ev1))); break;
        case 4032: SCANN_TRY((launch_scan_stages<Codec<32, 4>>(ix, w, st, ev0, ev1))); break;
        case 4048: SCANN_TRY((launch_scan_stages<Codec<48, 4>>(ix, w, st, ev0, ev1))); break;
        case 4064: SCANN_TRY((launch_scan_stages<Codec<64, 4>>(ix, w, st, ev0, ev1))); break;
        case 8004: SCANN_TRY((launch_scan_stages<Codec<4, 8>>(ix, w, st, ev0, ev1))); break;
        case 8008: SCANN_TRY((launch_scan_stages<Codec<8, 8>>(ix, w, st, ev0, ev1))); break;
        case 8016: SCANN_TRY((launch_scan_stages<Codec<16, 8>>(ix, w, st, ev0, ev1))); break;
        default:
            return fail(SCANN_HIP_UNIMPLEMENTED,
                        "num_subspaces must be 8,16,24,32,48,64 (num_codes <= 16) or 4,8,16 (num_codes <= 256)");
    }
    }

    SelectArgs s;
    s.P = w.P; s.m = w.m; s.k = w.k; s.cap = w.cap;
    s.exact_reorder = w.exact_reorder; s.local_only = local_only ? 1 : 0;
    const bool unsorted = w.exact_reorder && !local_only && !w.need_sorted_cands && w.k <= kTopkMaxK;
    s.unsorted = unsorted ? 1 : 0;
    s.queries = w.queries; s.q_stride = w.q_stride; s.tokens = w.tokens; s.vbase = w.vbase;
    s.thr = w.thr; s.cand_row = w.cand_row;
    s.cand_cnt = w.cand_cnt; s.cand = w.cand; s.counters = w.counters; s.cand_key = w.cand_key;
    s.cand_idx = w.cand_idx; s.cand_dist = w.cand_dist; s.cand_exact = w.cand_exact;
    s.cand_count = w.cand_count; s.out_idx = w.out_idx; s.out_dist = w.out_dist;
    s.out_count = w.out_count;
    // LDS key array: the candidate capacity rounded up to a power of two (bitonic path), at
    // most kSortCap; small lists run with 256-thread blocks so several fit a CU
    // (sorted path: room for the m keys that are kept -- a longer list is rank-selected straight from global
    // memory first, so the bitonic sort runs over pow2(m) keys, not pow2(capacity): half the stages' work at
    // capacity ~1.4 m)
    uint32_t lds_keys = std::min(kSortCap, next_pow2_u32(std::max(unsorted ? w.cap : std::min(w.cap, w.m), 64u)));
    // unsorted selection of a long list: straight from the global list, 512-thread workgroups with ~30 KB of
    // LDS (SCANN_HIP_SELECT_DIRECT=0: stage the keys as before)
    static const bool direct_ok = [] {
        const char *e = std::getenv("SCANN_HIP_SELECT_DIRECT");
        return !e || std::atoi(e) != 0;
    }();
    const bool direct = unsorted && direct_ok && lds_keys > 4096u;
    s.direct = direct ? 1u : 0u;
    s.sel_n = w.cap;
    if (direct) lds_keys = 64;
    s.lds_keys = lds_keys;
    const SelCfg scfg = sel_cfg(direct ? w.cap : lds_keys);
    const size_t lds_sel = (size_t)lds_keys * 8 + (size_t)(kSelectThreads / 64 + 4) * 4 +
                           (size_t)scfg.bins * 4 + (size_t)scfg.list * 8 + 48 * 8 + 2 * kDecodeStage * 4;
    const uint32_t sel_threads = direct ? 512u : (lds_keys <= 4096 ? 256u : kSelectThreads);
    SCANN_TRY(set_dyn_lds(select_rerank_kernel, lds_sel));
    hipLaunchKernelGGL(select_rerank_kernel, dim3(w.nq), dim3(sel_threads), lds_sel, st, ix, s);
    LAUNCH_CHECK();
    if (!w.exact_reorder) return SCANN_HIP_OK;
    const size_t lds_rr = (size_t)ix.dim * 4;
    // int8 row filter in front of the exact re-rank (K8b): single-GPU final stage, squared L2, lists long
    // enough for the two extra kernels to pay
    // (the local stage of a leaf-sharded search takes the same two kernels in their prefix form: ShortArgs::local_head;
    // SCANN_HIP_LOCAL_PRUNE=0: every local candidate re-ranked exactly, as before)
    bool local_prune_ok = true;   // (read per call: the tests flip it)
    if (local_only)
        if (const char *e = std::getenv("SCANN_HIP_LOCAL_PRUNE")) local_prune_ok = std::atoi(e) != 0;
    const bool i8_local = local_only && local_prune_ok && w.exact_reorder && !w.need_sorted_cands && w.k <= kTopkMaxK &&
                          w.m > 2 * kLocalHead;
    const bool i8 = (unsorted || i8_local) && w.use_i8 && ix.rows8 && ix.measure == SCANN_HIP_SQUARED_L2 && (ix.dim & 15u) == 0;
    if (i8) {
        I8RerankArgs ia;
        ia.rows8 = ix.rows8; ia.meta = reinterpret_cast<const float2 *>(ix.rows8_meta); ia.queries = w.queries;
        ia.q_stride = w.q_stride; ia.m = w.m; ia.cand_row = w.cand_row; ia.cand_count = w.cand_count;
        ia.lb = w.rr_lb; ia.ub = w.rr_ub;
        ia.uni_scale = ix.rows8_scale; ia.uni_E = ix.rows8_emax;
        if (ix.rows8_fmt == 1) {
            SCANN_TRY(set_dyn_lds(rerank_i8_kernel<1>, lds_rr));
            hipLaunchKernelGGL(rerank_i8_kernel<1>, dim3(ceil_div_u32(w.m, kI8PerBlock), w.nq), dim3(256), lds_rr, st, ix.dim, ia);
        } else if (ix.rows8_uniform) {
            SCANN_TRY(set_dyn_lds((rerank_i8_kernel<0, true>), lds_rr));
            hipLaunchKernelGGL((rerank_i8_kernel<0, true>), dim3(ceil_div_u32(w.m, kI8PerBlock), w.nq), dim3(256), lds_rr, st, ix.dim, ia);
        } else {
            SCANN_TRY(set_dyn_lds(rerank_i8_kernel<0>, lds_rr));
            hipLaunchKernelGGL(rerank_i8_kernel<0>, dim3(ceil_div_u32(w.m, kI8PerBlock), w.nq), dim3(256), lds_rr, st, ix.dim, ia);
        }
        LAUNCH_CHECK();
        ShortArgs sa;
        sa.m = w.m; sa.k = w.k; sa.queries = w.queries; sa.q_stride = w.q_stride; sa.lb = w.rr_lb; sa.ub = w.rr_ub;
        sa.cand_row = w.cand_row; sa.cand_idx = w.cand_idx; sa.cand_key = w.cand_key; sa.cand_count = w.cand_count;
        sa.cand_exact = w.cand_exact; sa.out_idx = w.out_idx; sa.out_dist = w.out_dist; sa.out_count = w.out_count;
        sa.local_head = local_only ? kLocalHead : 0u;
        const size_t lds_sh = (size_t)ix.dim * 4;
        SCANN_TRY(set_dyn_lds(rerank_short_kernel, lds_sh));
        hipLaunchKernelGGL(rerank_short_kernel, dim3(w.nq), dim3(256), lds_sh, st, ix, sa);
        LAUNCH_CHECK();
    } else {
        SCANN_TRY(set_dyn_lds(rerank_kernel, lds_rr));
        hipLaunchKernelGGL(rerank_kernel, dim3(ceil_div_u32(w.m, 32), w.nq), dim3(256), lds_rr, st, ix,
                           w.queries, w.q_stride, w.m, w.cand_row, w.cand_count, w.cand_exact);
        LAUNCH_CHECK();
    }
    if (local_only) return SCANN_HIP_OK;
    if (unsorted) {
        if (w.m <= 2048) {
            hipLaunchKernelGGL(final_topk_kernel<256>, dim3(w.nq), dim3(256), 0, st, w.m, w.k, w.cand_count,
                               w.cand_idx, w.cand_key, w.cand_exact, w.out_idx, w.out_dist, w.out_count);
        } else {
            hipLaunchKernelGGL(final_topk_kernel<1024>, dim3(w.nq), dim3(1024), 0, st, w.m, w.k, w.cand_count,
                               w.cand_idx, w.cand_key, w.cand_exact, w.out_idx, w.out_dist, w.out_count);
        }
        LAUNCH_CHECK();
        return SCANN_HIP_OK;
    }
    const size_t lds_fs = (size_t)next_pow2_u32(std::max(1u, w.m)) * 8;
    SCANN_TRY(set_dyn_lds(final_sort_kernel, lds_fs));
    hipLaunchKernelGGL(final_sort_kernel, dim3(w.nq), dim3(kSelectThreads), lds_fs, st, w.m, w.k,
                       w.cand_count, w.cand_idx, w.cand_exact, w.out_idx, w.out_dist, w.out_count);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

// Per-destination blocks for the all_to_all exchange: rank d merges the queries
// [d * nq/world, (d+1) * nq/world), so it needs exactly those rows of every rank's candidate
// arrays.  Block d = [keys u64 Qr*m | idx u32 Qr*m | exact f32 Qr*m | count u32 Qr], blocks
// block_bytes apart -- the layout txh_merge_device reads with rank_stride_bytes = block_bytes.
__global__ void pack_blocks_kernel(uint32_t world, uint32_t nq, uint32_t m,
                                   const uint64_t *__restrict__ keys, const uint32_t *__restrict__ idx,
                                   const float *__restrict__ exact, const uint32_t *__restrict__ count,
                                   unsigned char *__restrict__ out, size_t block_bytes) {
    const uint32_t qr = nq / world;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (uint64_t)nq * m) {
        const uint32_t q = (uint32_t)(e / m), i = (uint32_t)(e - (uint64_t)q * m);
        const uint32_t d = q / qr, ql = q - d * qr;
        unsigned char *blk = out + (size_t)d * block_bytes;
        const size_t slot = (size_t)ql * m + i, per = (size_t)qr * m;
        reinterpret_cast<uint64_t *>(blk)[slot] = keys[e];
        reinterpret_cast<uint32_t *>(blk + per * 8)[slot] = idx[e];
        reinterpret_cast<float *>(blk + per * 12)[slot] = exact[e];
    }
    if (e < nq) {
        const uint32_t q = (uint32_t)e, d = q / qr, ql = q - d * qr;
        reinterpret_cast<uint32_t *>(out + (size_t)d * block_bytes + (size_t)qr * m * 16)[ql] = count[q];
    }
}

int txh_launch_pack_blocks(uint32_t world, uint32_t nq, uint32_t m_local, const uint64_t *d_keys,
                           const uint32_t *d_idx, const float *d_exact, const uint32_t *d_count,
                           void *d_out, size_t block_bytes, hipStream_t st) {
    if (nq == 0) return SCANN_HIP_OK;
    if (world == 0 || nq % world != 0)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "the batch must divide evenly over the ranks");
    const size_t need = (size_t)(nq / world) * m_local * 16 + (size_t)(nq / world) * 4;
    if (block_bytes < need || (block_bytes & 7u))
        return fail(SCANN_HIP_INVALID_ARGUMENT, "block_bytes too small or not a multiple of 8");
    const uint64_t work = std::max<uint64_t>((uint64_t)nq * m_local, nq);
    hipLaunchKernelGGL(pack_blocks_kernel, dim3((uint32_t)ceil_div_u64(work, 256)), dim3(256), 0, st, world, nq,
                       m_local, d_keys, d_idx, d_exact, d_count, static_cast<unsigned char *>(d_out),
                       block_bytes);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int txh_launch_merge(uint32_t world, uint32_t nq, uint32_t m_local, uint32_t m, uint32_t k,
                     size_t rank_stride_bytes, const uint64_t *d_keys, const uint32_t *d_idx,
                     const float *d_exact, const uint32_t *d_count, uint32_t *d_out_idx,
                     float *d_out_dist, uint32_t *d_out_count, uint32_t *d_status, hipStream_t st,
                     const uint32_t *d_qoff) {
    if (nq == 0) return SCANN_HIP_OK;
    if (world == 0 || world > 64) return fail(SCANN_HIP_INVALID_ARGUMENT, "world must be 1..64");
    if (m_local == 0 || m_local > m) return fail(SCANN_HIP_INVALID_ARGUMENT, "need 0 < m_local <= m");
    if (m > kMaxPreReorderK)
        return fail(SCANN_HIP_UNIMPLEMENTED, "pre_reorder_k exceeds the LDS merge capacity");
    const uint32_t m2 = next_pow2_u32(std::max<uint32_t>(1u, m));
    const size_t lds = (size_t)m2 * 12 + 16 + (kSelectThreads / 64) * 8;
    SCANN_TRY(set_dyn_lds(merge_kernel, lds));
    hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(kSelectThreads), lds, st, world, nq, m_local, m, k, m2,
                       rank_stride_bytes, d_keys, d_idx, d_exact, d_count, d_out_idx, d_out_dist,
                       d_out_count, d_status, d_qoff);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int txh_launch_lut_from_query(const TxhIndexDev &ix, const float *d_queries, uint32_t nq,
                              uint32_t q_stride, const uint32_t *d_leaf_for_query,
                              float *d_out_lut, hipStream_t st) {
    if (nq == 0) return SCANN_HIP_OK;
    const size_t lds = (size_t)ix.dim * sizeof(float);
    SCANN_TRY(set_dyn_lds(lut_from_query_kernel, lds));
    hipLaunchKernelGGL(lut_from_query_kernel, dim3(nq), dim3(256), lds, st, ix, d_queries, q_stride,
                       d_leaf_for_query, d_out_lut);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int txh_launch_adc_distances(const TxhIndexDev &ix, const float *d_luts, uint32_t nq, float *d_out,
                             hipStream_t st) {
    if (nq == 0 || ix.n_local == 0) return SCANN_HIP_OK;
    const uint32_t gx = (uint32_t)std::min<uint64_t>(ceil_div_u64(ix.n_local, 256), 4096);
    dim3 grid(gx, nq);
    const size_t lds = (size_t)ix.S * ix.K * sizeof(float);
    SCANN_TRY(set_dyn_lds(adc_distances_kernel, lds));
    hipLaunchKernelGGL(adc_distances_kernel, grid, dim3(256), lds, st, ix, d_luts, d_out);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int launch_lut16_u8_batch(const uint8_t *d_packed, const uint8_t *d_lut8, uint32_t S, uint64_t n,
                          float bias, float mult, float *d_out, hipStream_t st) {
    if (n == 0) return SCANN_HIP_OK;
    const uint32_t gx = (uint32_t)std::min<uint64_t>(ceil_div_u64(n, 256), 8192);
    hipLaunchKernelGGL(lut16_u8_batch_kernel, dim3(gx), dim3(256), (size_t)S * 16, st, d_packed, d_lut8,
                       S, n, bias, mult, d_out);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

__global__ __launch_bounds__(256) void transpose_centers_kernel(const float *__restrict__ centers, uint32_t L,
                                                                uint32_t dim, uint32_t pitch, float *__restrict__ out) {
    const uint64_t total = (uint64_t)dim * pitch;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint32_t j = (uint32_t)(e / pitch), c = (uint32_t)(e - (uint64_t)j * pitch);
        out[e] = c < L ? centers[(size_t)c * dim + j] : 0.0f;
    }
}

int launch_transpose_centers(const float *d_centers, uint32_t L, uint32_t dim, uint32_t pitch, float *d_out,
                             hipStream_t st) {
    if (L == 0 || dim == 0) return SCANN_HIP_OK;
    hipLaunchKernelGGL(transpose_centers_kernel, dim3((uint32_t)std::min<uint64_t>(ceil_div_u64((uint64_t)dim * pitch, 256), 4096)),
                       dim3(256), 0, st, d_centers, L, dim, pitch, d_out);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int launch_rows_fp8_build(const float *d_rows, uint64_t n, uint32_t dim, uint32_t stride, uint8_t *d_rows8,
                          void *d_meta, uint32_t *d_mismatch, hipStream_t st) {
    if (n == 0) return SCANN_HIP_OK;
    hipLaunchKernelGGL(rows_fp8_build_kernel, dim3((uint32_t)ceil_div_u64(n, 32)), dim3(256), 0, st, d_rows, n, dim,
                       stride, d_rows8, reinterpret_cast<float2 *>(d_meta), d_mismatch);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int launch_fp8_quantize(const float *d_values, uint64_t n, float scale, int format, uint8_t *d_out, hipStream_t st) {
    if (n == 0) return SCANN_HIP_OK;
    hipLaunchKernelGGL(fp8_quantize_kernel, dim3((uint32_t)std::min<uint64_t>(ceil_div_u64(n, 256), 65535)), dim3(256), 0,
                       st, d_values, n, scale, format, d_out);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int launch_fp8_dequantize(const uint8_t *d_bits, uint64_t n, float scale, int format, float *d_out, hipStream_t st) {
    if (n == 0) return SCANN_HIP_OK;
    hipLaunchKernelGGL(fp8_dequantize_kernel, dim3((uint32_t)std::min<uint64_t>(ceil_div_u64(n, 256), 65535)), dim3(256),
                       0, st, d_bits, n, scale, format, d_out);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int launch_fp8_one_to_many(const float *d_query, uint32_t dim, const uint8_t *d_db, uint64_t stride, uint64_t n,
                           int dot, float *d_out, hipStream_t st) {
    if (n == 0) return SCANN_HIP_OK;
    const size_t lds = (size_t)dim * sizeof(float);
    SCANN_TRY(set_dyn_lds(fp8_one_to_many_kernel, lds));
    hipLaunchKernelGGL(fp8_one_to_many_kernel, dim3((uint32_t)std::min<uint64_t>(ceil_div_u64(n, 256), 65535)),
                       dim3(256), lds, st, d_query, dim, d_db, stride, n, dot, d_out);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int launch_rows_i8_build(const float *d_rows, uint64_t n, uint32_t dim, uint32_t stride, int8_t *d_rows8,
                         void *d_meta, hipStream_t st, float uni_scale) {
    if (n == 0) return SCANN_HIP_OK;
    hipLaunchKernelGGL(rows_i8_build_kernel, dim3((uint32_t)ceil_div_u64(n, 32)), dim3(256), 0, st, d_rows, n, dim,
                       stride, d_rows8, reinterpret_cast<float2 *>(d_meta), uni_scale);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int launch_lut16_quantize(const float *d_tables, uint32_t S, uint8_t *d_lut8, float *d_bias_mult,
                          hipStream_t st) {
    if (S == 0) return SCANN_HIP_OK;
    hipLaunchKernelGGL(lut16_quantize_kernel, dim3(1), dim3(256), 0, st, d_tables, S, d_lut8, d_bias_mult);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int launch_codes_sp_build(const uint32_t *d_codes, uint64_t n, uint32_t S, uint32_t *d_codes_sp, hipStream_t st) {
    if (n == 0) return SCANN_HIP_OK;
    hipLaunchKernelGGL(codes_sp_build_kernel, dim3((uint32_t)ceil_div_u64(n, 256)), dim3(256), 0, st, d_codes, n, S, d_codes_sp);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

int launch_encode(const float *d_codebook, uint32_t S, uint32_t K, uint32_t dsub, const float *d_rows,
                  uint64_t n, uint32_t stride, const float *d_centers, const uint32_t *d_leaf_of_row,
                  uint8_t *d_out, hipStream_t st) {
    if (n == 0) return SCANN_HIP_OK;
    const uint32_t gx = (uint32_t)std::min<uint64_t>(ceil_div_u64(n * S, 256), 16384);
    hipLaunchKernelGGL(encode_kernel, dim3(gx), dim3(256), 0, st, d_codebook, S, K, dsub, d_rows, n,
                       stride, d_centers, d_leaf_of_row, d_out);
    LAUNCH_CHECK();
    return SCANN_HIP_OK;
}

}  // namespace scann
