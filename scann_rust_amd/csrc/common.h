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
#endif

}  // namespace scann
