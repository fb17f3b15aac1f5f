// common.h -- shared host/device helpers for libscann_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <string>

#include "../../include/scann_hip.h"

namespace scann {

// ---- error plumbing ---------------------------------------------------------------
void set_last_error(const std::string &msg);
int fail(int status, const std::string &msg);
// scann_hip_txh_create with the status reported for inconsistent array CONTENTS (api.hip)
int txh_create_checked(scann_hip_ctx *ctx, const scann_hip_txh_desc *d, scann_hip_index **out,
                       int content_status);

#define SCANN_HIP_CHECK(expr)                                                         \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            return ::scann::fail(_e == hipErrorOutOfMemory ? SCANN_HIP_RESOURCE_EXHAUSTED \
                                                           : SCANN_HIP_INTERNAL,      \
                                 std::string(#expr) + ": " + hipGetErrorString(_e));  \
        }                                                                             \
    } while (0)

#define SCANN_TRY(expr)                  \
    do {                                 \
        int _s = (expr);                 \
        if (_s != SCANN_HIP_OK) return _s; \
    } while (0)

// RAII device buffer (grow-only)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int ensure(size_t need) {
        if (need <= bytes && p) return SCANN_HIP_OK;
        release();
        if (need == 0) need = 16;
        SCANN_HIP_CHECK(hipMalloc(&p, need));
        bytes = need;
        return SCANN_HIP_OK;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

static inline int upload(DevBuf &b, const void *src, size_t bytes) {
    SCANN_TRY(b.ensure(bytes));
    if (bytes) SCANN_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return SCANN_HIP_OK;
}

static inline uint32_t ceil_div_u32(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static inline uint64_t ceil_div_u64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
static inline uint32_t next_pow2_u32(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ---- order-preserving f32 <-> u32 ----------------------------------------------------
// Monotone map of IEEE-754 f32 onto u32 (total order: -NaN < -inf < ... < -0 < +0 < ...
// < +inf < +NaN).  For the non-negative values of every squared-L2 / LUT-sum path this
// is `bits | 0x80000000`.
__host__ __device__ static inline uint32_t f32_to_ordered(float f) {
    uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ static inline float ordered_to_f32(uint32_t o) {
    uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __builtin_bit_cast(float, b);
}
__host__ __device__ static inline uint64_t make_key(float dist, uint32_t pos) {
    return ((uint64_t)f32_to_ordered(dist) << 32) | pos;
}

#define SCANN_KEY_MAX 0xFFFFFFFFFFFFFFFFull

#if defined(__HIPCC__)
// ---- block-wide bitonic sort of u64 keys in LDS (ascending) --------------------------
// n must be a power of two; every thread of the block must call.
__device__ static inline void bitonic_sort_lds(uint64_t *keys, uint32_t n) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < (n >> 1); i += nt) {
                uint32_t lo = 2 * j * (i / j) + (i % j);
                uint32_t hi = lo + j;
                uint64_t a = keys[lo], b = keys[hi];
                bool up = ((lo & k) == 0);
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

// Wave64 exclusive prefix count of a predicate + wave total (ballot based).
__device__ static inline uint32_t wave_prefix_count(bool pred, uint32_t *total) {
    unsigned long long m = __ballot(pred);
    uint32_t lane = threadIdx.x & 63;
    *total = (uint32_t)__popcll(m);
    return (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// Load through the constant address space: with a wave-uniform address the compiler emits
// s_load (SMEM, SGPR result) instead of a vector load + v_readfirstlane.  Only for buffers
// that no thread of the running kernel writes.
template <typename T>
__device__ __forceinline__ T uniform_load(const T *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>((uint64_t)p);
}

constexpr uint32_t kSelBinsMax = 4096;   // histogram bins of block_select (large inputs)
constexpr uint32_t kSelListMax = 1024;   // members of the rank's bin ranked exactly

// =====================================================================================
// block_select: value of 1-based rank `rank` among vals[0..n) (LDS), ties allowed.  One
// pass over [min, max] of the values with kSelBins histogram bins ((v - min) >> sh, monotone)
// finds the rank's bin; a bin of <= kSelList members is ranked exactly by counting,
// otherwise the bin becomes the new range.  Every thread of the block must call; all get
// the same result.  Values equal to ~0 are "absent": they sort last and do not widen the
// range; if fewer than `rank` values are present the result is ~0.
// hist: LDS u32[cfg.bins]; list: LDS T[cfg.list]; red: LDS u64[48].
// =====================================================================================
struct SelCfg {
    uint32_t bins, list;
};
// Histogram bins / exact-rank list size for selecting among at most n values: small inputs
// take small tables so that several blocks share a CU's LDS.
__host__ __device__ static inline SelCfg sel_cfg(uint32_t n) {
    SelCfg c;
    c.bins = n <= 4096u ? 1024u : kSelBinsMax;
    c.list = n <= 4096u ? 256u : kSelListMax;
    return c;
}

template <typename T>
__device__ __forceinline__ T shfl_xor_t(T v, int d) {
    if constexpr (sizeof(T) == 8) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d);
        return ((uint64_t)hi << 32) | lo;
    } else {
        return (T)__shfl_xor((int)v, d);
    }
}

// AGENT: vals was written by OTHER workgroups of the same launch with agent-scope (write-through, sc1) stores: every
// read of it must then be an agent-scope load (a plain load may be served from this CU's L1 / a stale line).
template <typename T, bool AGENT = false>
__device__ static T block_select(const T *vals_in, uint32_t n, uint32_t rank, SelCfg cfg, uint32_t *hist,
                                 T *list, uint64_t *red) {
    struct Vals {
        const T *p;
        __device__ __forceinline__ T operator[](uint32_t i) const {
            if constexpr (AGENT) return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else return p[i];
        }
    };
    const Vals vals{vals_in};
    const uint32_t kSelBins = cfg.bins, kSelList = cfg.list;
    const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & 63u, wave = tid >> 6, nwaves = nt >> 6;
    constexpr T kAbsent = ~(T)0;
    T vmin = kAbsent, vmax = 0;
    uint32_t present = 0;
    for (uint32_t i = tid; i < n; i += nt) {
        const T v = vals[i];
        if (v != kAbsent) {
            vmin = v < vmin ? v : vmin;
            vmax = v > vmax ? v : vmax;
            ++present;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T a = shfl_xor_t(vmin, o), b = shfl_xor_t(vmax, o);
        vmin = a < vmin ? a : vmin;
        vmax = b > vmax ? b : vmax;
        present += (uint32_t)__shfl_xor((int)present, o);
    }
    uint32_t *red32 = reinterpret_cast<uint32_t *>(red + 32);   // [0..15] wave sums, 16.. results
    if (lane == 0) {
        red[wave] = vmin;
        red[16 + wave] = vmax;
        red32[wave] = present;
    }
    __syncthreads();
    present = 0;
    for (uint32_t w2 = 0; w2 < nwaves; ++w2) {
        const T a = (T)red[w2], b = (T)red[16 + w2];
        vmin = a < vmin ? a : vmin;
        vmax = b > vmax ? b : vmax;
        present += red32[w2];
    }
    if (present < rank) return kAbsent;   // block-uniform
    for (;;) {
        const T range = vmax - vmin;
        uint32_t sh = 0;
        while ((range >> sh) >= (T)kSelBins) ++sh;
        __syncthreads();          // previous readers of hist / red32 are done
        for (uint32_t i = tid; i < kSelBins; i += nt) hist[i] = 0;
        if (tid == 0) red32[20] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += nt) {
            const T v = vals[i];
            if (v >= vmin && v <= vmax) atomicAdd(&hist[(uint32_t)((v - vmin) >> sh)], 1u);
        }
        __syncthreads();
        // bin of the rank: block-wide inclusive scan over per-thread groups of bins
        const uint32_t per = (kSelBins + nt - 1) / nt;
        const uint32_t b0 = tid * per, b1 = min(b0 + per, kSelBins);
        uint32_t mine = 0;
        for (uint32_t b = b0; b < b1; ++b) mine += hist[b];
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if ((int)lane >= o) incl += up;
        }
        if (lane == 63) red32[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w2 = 0; w2 < wave; ++w2) wbase += red32[w2];
        incl += wbase;
        const uint32_t excl = incl - mine;
        if (excl < rank && rank <= incl) {   // exactly one thread
            uint32_t c = excl;
            for (uint32_t b = b0; b < b1; ++b) {
                const uint32_t h = hist[b];
                if (c + h >= rank) {
                    red32[16] = b;
                    red32[17] = rank - c;      // 1-based rank inside the bin
                    red32[18] = h;
                    break;
                }
                c += h;
            }
        }
        __syncthreads();
        const uint32_t bin = red32[16], rk = red32[17], pop = red32[18];
        const T lo = vmin + ((T)bin << sh);
        T hi = lo + (((T)1 << sh) - 1);
        if (hi > vmax || hi < lo) hi = vmax;
        if (sh == 0) return lo;
        if (pop <= kSelList) {
            for (uint32_t i = tid; i < n; i += nt) {
                const T v = vals[i];
                if (v >= lo && v <= hi) list[atomicAdd(&red32[20], 1u)] = v;
            }
            __syncthreads();
            for (uint32_t i = tid; i < pop; i += nt) {
                const T v = list[i];
                uint32_t r = 0;
                for (uint32_t j2 = 0; j2 < pop; ++j2) {
                    const T u = list[j2];
                    r += (u < v || (u == v && j2 < i)) ? 1u : 0u;
                }
                if (r + 1 == rk) red[24] = v;
            }
            __syncthreads();
            return (T)red[24];
        }
        vmin = lo;
        vmax = hi;
        rank = rk;
    }
}

// =====================================================================================
// block_tail_select: value of 1-based rank `rank` among the block's values (VPT per thread through
// load(i), ~0 = absent), for ranks in the low tail (rank <= blockDim.x / 2).
//   1. every thread takes the minimum of its values: nt minima, each an element of the set, so the
//      rank-th smallest minimum is >= the rank-th smallest value (the rank smallest minima are rank
//      distinct elements) -- a pivot a few percent above the answer;
//   2. the values <= pivot (a little over `rank` of them) are compacted into `list`;
//   3. the rank-th smallest of the list is the answer.
// If the list overflows `list_cap` (heavy ties at the pivot) the pivot itself is returned: still an
// upper bound of the answer, and *exact_out = false.  No histogram over the whole input, no staging of
// the input in LDS.  mins: LDS u32[blockDim.x]; list: LDS u32[list_cap]; hist/slist/red: block_select's
// scratch for sel_cfg(max(blockDim.x, list_cap)); cnt: LDS u32.  All threads must call.
// =====================================================================================
template <int VPT, typename Load>
__device__ static uint32_t block_tail_select(Load load, uint32_t rank, uint32_t *mins, uint32_t *list,
                                             uint32_t list_cap, uint32_t *hist, uint32_t *slist, uint64_t *red,
                                             uint32_t *cnt, bool *exact_out) {
    // load(i), i < VPT: this thread's i-th value (called twice per i: the values are re-read from
    // L2-hot memory rather than held in VPT registers, so that several blocks fit a CU)
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const uint32_t x = load(i);
        mn = x < mn ? x : mn;
    }
    mins[tid] = mn;
    if (tid == 0) *cnt = 0;
    __syncthreads();
    const SelCfg cfg = sel_cfg(nt > list_cap ? nt : list_cap);
    uint32_t pivot = block_select<uint32_t>(mins, nt, rank, cfg, hist, slist, red);
    __syncthreads();
    if (pivot == 0xFFFFFFFFu) pivot = 0xFFFFFFFEu;   // fewer than `rank` threads hold a value: list every value
    // one LDS atomic per wave: lanes count their own values under the pivot, a wave scan places them
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) mine += load(i) <= pivot ? 1u : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if ((int)(tid & 63u) >= o) incl += up;
    }
    uint32_t base = 0;
    if ((tid & 63u) == 63u && incl) base = atomicAdd(cnt, incl);
    uint32_t pos = (uint32_t)__shfl((int)base, 63) + incl - mine;
    if (mine) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const uint32_t x = load(i);
            if (x <= pivot) {
                if (pos < list_cap) list[pos] = x;
                ++pos;
            }
        }
    }
    __syncthreads();
    const uint32_t c = *cnt;
    if (c > list_cap) {
        *exact_out = false;
        return pivot;
    }
    *exact_out = true;
    const uint32_t r = block_select<uint32_t>(list, c, rank, cfg, hist, slist, red);
    __syncthreads();
    return r;
}

#endif

}  // namespace scann
