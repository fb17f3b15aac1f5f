// Microbenchmark of block_select (csrc/common.h): time per workgroup for the rank-select of `rank` among n
// values staged in LDS, as the select kernels call it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I scann_rust_amd/csrc tools/micro/select_rate.hip -o /tmp/select_rate
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace scann { void set_last_error(const std::string &) {} int fail(int s, const std::string &) { return s; } }
using namespace scann;

template <typename T>
__global__ __launch_bounds__(1024) void k(const T *in, uint32_t n, uint32_t rank, T *out, uint64_t *cyc, int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *vals = reinterpret_cast<T *>(smem);
    const SelCfg cfg = sel_cfg(n);
    uint32_t *hist = reinterpret_cast<uint32_t *>(vals + ((n + 3) & ~3u));
    T *list = reinterpret_cast<T *>(hist + cfg.bins);
    uint64_t *red = reinterpret_cast<uint64_t *>(list + cfg.list);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) vals[i] = in[(size_t)blockIdx.x * n + i];
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    T r = 0;
    for (int i = 0; i < reps; ++i) {
        r = block_select<T>(vals, n, rank, cfg, hist, list, red);
        __syncthreads();
    }
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = r;
        cyc[blockIdx.x] = t1 - t0;
    }
}

template <typename T>
void run(uint32_t n, uint32_t rank, int nt, int blocks, int dist) {
    std::vector<T> h((size_t)blocks * n);
    srand(n + rank);
    for (auto &v : h) {
        double u = 0;
        for (int j = 0; j < (dist ? 8 : 1); ++j) u += rand() / (double)RAND_MAX;   // dist 1: bell-shaped (sum of 8)
        v = (T)(u / (dist ? 8 : 1) * 1e9);
        if (sizeof(T) == 8) v = (T)(((uint64_t)v << 32) | (uint32_t)rand());
    }
    T *din, *dout; uint64_t *dc;
    hipMalloc(&din, h.size() * sizeof(T)); hipMalloc(&dout, blocks * sizeof(T)); hipMalloc(&dc, blocks * 8);
    hipMemcpy(din, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    const SelCfg cfg = sel_cfg(n);
    const size_t lds = (size_t)((n + 3) & ~3u) * sizeof(T) + cfg.bins * 4 + cfg.list * sizeof(T) + 48 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int reps = 20;
    k<T><<<blocks, nt, lds>>>(din, n, rank, dout, dc, reps);
    hipDeviceSynchronize();
    std::vector<uint64_t> c(blocks);
    std::vector<T> o(blocks);
    hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), dout, blocks * sizeof(T), hipMemcpyDeviceToHost);
    // check block 0 on the host
    std::vector<T> s(h.begin(), h.begin() + n);
    std::sort(s.begin(), s.end());
    double avg = 0;
    for (auto x : c) avg += x;
    avg /= blocks;
    printf("%s n %6u rank %6u nt %4d blocks %4d %s: %7.2f us per select  (%s)\n", sizeof(T) == 8 ? "u64" : "u32", n, rank, nt,
           blocks, dist ? "bell   " : "uniform", avg * 0.01 / reps, o[0] == s[rank - 1] ? "ok" : "WRONG");
    hipFree(din); hipFree(dout); hipFree(dc);
}

#include <algorithm>
int main() {
    for (int dist = 0; dist < 2; ++dist) {
        run<uint32_t>(32768, 237, 1024, 256, dist);    // threshold_select at C3
        run<uint64_t>(7400, 5000, 1024, 512, dist);    // select_rerank at C3
        run<uint64_t>(1000, 10, 1024, 256, dist);      // small finish, Partitioned 10k
        run<uint64_t>(4096, 10, 1024, 256, dist);      // small finish: minima
        run<uint32_t>(1024, 10, 1024, 256, dist);
        run<uint32_t>(1024, 10, 256, 256, dist);
    }
    return 0;
}
