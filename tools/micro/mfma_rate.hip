// Microbenchmark: sustained issue rate of the i8 MFMAs on gfx950 (one kernel per variant).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const v4i *in, int *out, int iters) {
    const v4i a = in[threadIdx.x], b = in[256 + threadIdx.x];
    v16i c0 = {0}, c1 = {0};
    v4i d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // one dependent chain, 32x32x32
#pragma unroll
            for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        } else if (MODE == 1) {     // two chains
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
            }
        } else if (MODE == 2) {     // 16x16x64, one chain (32 per iteration = same MACs as 16 of the 32x32x32)
#pragma unroll
            for (int j = 0; j < 32; ++j) d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
        } else {                    // 16x16x64, two chains
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, d1, 0, 0, 0);
            }
        }
    }
    int s = c0[0] + c1[5] + d0[1] + d1[2];
    if (s == 0x7fffffff) out[0] = s;
}

template <int MODE>
void run(const char *name, const v4i *din, int *dout, int wgs_per_cu, int zero) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * wgs_per_cu, 256>>>(din + (zero ? 512 : 0), dout, 10);
    hipEventRecord(e0);
    k<MODE><<<256 * wgs_per_cu, 256>>>(din + (zero ? 512 : 0), dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // MACs: per wave per iteration 16 x (32*32*32) [or 32 x 16*16*64]
    const double macs = (double)256 * wgs_per_cu * 4 * iters * 16.0 * 32768.0;
    const double per_simd_mfma = (double)wgs_per_cu * iters * (MODE < 2 ? 16.0 : 32.0);   // waves per SIMD = wgs_per_cu
    printf("%-28s waves/SIMD %d %s: %.3f ms  %.2f PMAC/s  -> %.1f ns per MFMA per SIMD\n", name, wgs_per_cu,
           zero ? "zeros " : "random", ms, macs / (ms * 1e-3) / 1e15, ms * 1e6 / per_simd_mfma);
}

int main() {
    v4i *din; int *dout;
    hipMalloc(&din, 1024 * sizeof(v4i)); hipMalloc(&dout, 64);
    v4i h[1024];
    srand(1);
    for (int i = 0; i < 512; ++i) h[i] = v4i{rand(), rand(), rand(), rand()};
    if (getenv("ONEHOT"))   // A operand as in the LUT16 prefilter: one byte of 16 set to 1
        for (int i = 0; i < 256; ++i) {
            int c = rand() & 15;
            int w[4] = {0, 0, 0, 0};
            w[c >> 2] = 1 << (8 * (c & 3));
            h[i] = v4i{w[0], w[1], w[2], w[3]};
        }
    for (int i = 512; i < 1024; ++i) h[i] = v4i{0, 0, 0, 0};
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    for (int z = 0; z < 2; ++z)
        for (int w = 1; w <= 4; w += (w == 1 ? 1 : 2)) {
            run<0>("i8 32x32x32, 1 chain", din, dout, w, z);
            run<1>("i8 32x32x32, 2 chains", din, dout, w, z);
            run<2>("i8 16x16x64, 1 chain", din, dout, w, z);
            run<3>("i8 16x16x64, 2 chains", din, dout, w, z);
        }
    return 0;
}
