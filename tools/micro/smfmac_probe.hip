// Probe of v_smfmac_i32_32x32x64_i8 on gfx950: operand layout (which lane/byte of A, B and which bits of the
// index register address which (m, k, n)) and sustained issue rate next to the dense v_mfma_i32_32x32x32_i8.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/smfmac_probe.hip -o tools/micro/smfmac_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));

// one wave per block: D = smfmac(A, B, 0, idx); operands straight from memory
__global__ __launch_bounds__(64) void one_kernel(const v4i *a, const v8i *b, const int *idx, v16i *d) {
    const int l = threadIdx.x, blk = blockIdx.x;
    v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a[blk * 64 + l], b[blk * 64 + l], c, idx[blk * 64 + l], 0, 0);
    d[blk * 64 + l] = c;
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(const int *in, int *out, int iters) {
    const v4i *in4 = reinterpret_cast<const v4i *>(in);
    const v4i a = in4[threadIdx.x], b = in4[256 + threadIdx.x];
    const v8i bb = {b[0], b[1], b[2], b[3], b[3], b[2], b[1], b[0]};
    const int idx = in[4096 + threadIdx.x];
    v16i c0 = {0}, c1 = {0};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, bb, c0, idx, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c0 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, bb, c0, idx, 0, 0);
                c1 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(b, bb, c1, idx, 0, 0);
            }
        }
    }
    int s = c0[0] + c1[5];
    if (s == 0x7fffffff) out[0] = s;
}

template <int MODE>
static void rate(const char *name, const int *din, int *dout, int wgs_per_cu) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<256 * wgs_per_cu, 256>>>(din, dout, 10);
    hipEventRecord(e0);
    rate_kernel<MODE><<<256 * wgs_per_cu, 256>>>(din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)wgs_per_cu * iters * 16.0;
    printf("%-34s waves/SIMD %d: %.3f ms -> %.1f ns per instruction per SIMD\n", name, wgs_per_cu, ms, ms * 1e6 / per_simd);
}

int main() {
    // ---- layout probe: A = one compressed value 1 at (lane la, byte ba) with index bits i; B position (lane, byte)
    // carries a 11-bit id in two passes (low 7 bits, high 4 bits)
    const int las[] = {0, 1, 5, 31, 32, 33, 63};
    const int nla = sizeof(las) / sizeof(las[0]);
    const int ncfg = nla * 16 * 4, nblk = ncfg * 2;
    std::vector<int> ha((size_t)nblk * 64 * 4, 0), hb((size_t)nblk * 64 * 8, 0), hi((size_t)nblk * 64, 0), hd((size_t)nblk * 64 * 16);
    for (int c = 0; c < ncfg; ++c) {
        const int la = las[c / 64], ba = (c / 4) % 16, ii = c % 4;
        for (int pass = 0; pass < 2; ++pass) {
            const int blk = c * 2 + pass;
            reinterpret_cast<signed char *>(&ha[((size_t)blk * 64 + la) * 4])[ba] = 1;
            hi[(size_t)blk * 64 + la] = ii << (2 * ba);
            for (int l = 0; l < 64; ++l)
                for (int by = 0; by < 32; ++by) {
                    const int id = l * 32 + by + 1;   // (0 = nothing selected)
                    reinterpret_cast<signed char *>(&hb[((size_t)blk * 64 + l) * 8])[by] = (signed char)(pass == 0 ? (id & 127) : (id >> 7));
                }
        }
    }
    int *da, *db, *di, *dd;
    hipMalloc(&da, ha.size() * 4); hipMalloc(&db, hb.size() * 4); hipMalloc(&di, hi.size() * 4); hipMalloc(&dd, hd.size() * 4);
    hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    one_kernel<<<nblk, 64>>>(reinterpret_cast<v4i *>(da), reinterpret_cast<v8i *>(db), di, reinterpret_cast<v16i *>(dd));
    hipMemcpy(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost);
    // D layout assumed (checked by the dense kernels of the library): lane l, reg r -> n = l & 31, m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    int bad = 0;
    for (int c = 0; c < ncfg; ++c) {
        const int la = las[c / 64], ba = (c / 4) % 16, ii = c % 4;
        int mrow = -1, rows = 0;
        int kl[32], kb[32];
        for (int m = 0; m < 32; ++m) {
            bool any = false;
            for (int n = 0; n < 32; ++n) {
                const int l = n + 32 * ((m >> 2) & 1), r = (m & 3) + 4 * (m >> 3);
                const int lo = hd[((size_t)(c * 2) * 64 + l) * 16 + r], hi2 = hd[((size_t)(c * 2 + 1) * 64 + l) * 16 + r];
                if (lo || hi2) {
                    any = true;
                    const int id = lo + (hi2 << 7) - 1;
                    kl[n] = id / 32; kb[n] = id % 32;
                }
            }
            if (any) { mrow = m; ++rows; }
        }
        // hypothesis: row = la & 31; compressed byte ba (value slot ba & 1 of group (ba >> 1) & 3) with index ii pairs with
        // B lane n + 32 (ba >> 3), byte 16 (la >> 5) + 4 ((ba >> 1) & 3) + ii
        bool ok = rows == 1 && mrow == (la & 31);
        for (int n = 0; n < 32 && ok; ++n) ok = kl[n] == n + 32 * (ba >> 3) && kb[n] == 16 * (la >> 5) + 4 * ((ba >> 1) & 3) + ii;
        if (!ok) {
            ++bad;
            if (bad <= 400) {
                printf("A(lane %2d, byte %2d, idx %d): rows=%d m=%d | n=0 -> B(lane %d, byte %d), n=1 -> B(lane %d, byte %d), n=31 -> B(lane %d, byte %d)\n", la, ba, ii,
                       rows, mrow, kl[0], kb[0], kl[1], kb[1], kl[31], kb[31]);
            }
        }
    }
    printf("layout hypothesis [A(lane l, byte b, index i) -> row l & 31, pairs with B(lane n + 32 (b >> 3), byte 16 (l >> 5) + 4 ((b >> 1) & 3) + i) of column n]: %s (%d of %d probes differ)\n",
           bad ? "WRONG" : "confirmed", bad, ncfg);

    // ---- both compressed values of a group on the SAME index, and index order reversed: is the selection a plain mux?
    {
        std::vector<int> a2(64 * 4 * 4, 0), b2(64 * 8 * 4, 0), i2(64 * 4, 0), d2(64 * 16 * 4);
        for (int t = 0; t < 4; ++t) {
            signed char *ap = reinterpret_cast<signed char *>(&a2[(size_t)t * 64 * 4]);
            ap[0] = 3; ap[1] = 5;   // lane 0, group 0: values (3, 5)
            const int sel[4][2] = {{0, 1}, {1, 0}, {2, 2}, {3, 0}};
            i2[t * 64] = sel[t][0] | (sel[t][1] << 2);
            for (int l = 0; l < 64; ++l)
                for (int by = 0; by < 32; ++by)
                    reinterpret_cast<signed char *>(&b2[((size_t)t * 64 + l) * 8])[by] = (signed char)(l < 32 && by < 4 ? (by == 0 ? 1 : by == 1 ? 10 : by == 2 ? 20 : 30) : 0);
        }
        int *pa, *pb, *pi, *pd;
        hipMalloc(&pa, a2.size() * 4); hipMalloc(&pb, b2.size() * 4); hipMalloc(&pi, i2.size() * 4); hipMalloc(&pd, d2.size() * 4);
        hipMemcpy(pa, a2.data(), a2.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(pb, b2.data(), b2.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(pi, i2.data(), i2.size() * 4, hipMemcpyHostToDevice);
        one_kernel<<<4, 64>>>(reinterpret_cast<v4i *>(pa), reinterpret_cast<v8i *>(pb), pi, reinterpret_cast<v16i *>(pd));
        hipMemcpy(d2.data(), pd, d2.size() * 4, hipMemcpyDeviceToHost);
        const int expect[4] = {3 * 1 + 5 * 10, 3 * 10 + 5 * 1, 3 * 20 + 5 * 20, 3 * 30 + 5 * 1};
        for (int t = 0; t < 4; ++t)
            printf("index pair test %d: D[0][0] = %d (plain mux expects %d)\n", t, d2[(size_t)t * 64 * 16], expect[t]);
    }

    // ---- rates
    std::vector<int> hr(4096 + 256);
    srand(1);
    for (auto &x : hr) x = rand() ^ (rand() << 16);
    if (getenv("ONEHOT"))
        for (int i = 0; i < 1024; ++i) hr[i] = (i & 3) == (rand() & 3) ? 1 << (8 * (rand() & 3)) : 0;
    int *dr, *dout;
    hipMalloc(&dr, hr.size() * 4); hipMalloc(&dout, 64);
    hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; w += (w == 1 ? 1 : 2)) {
        rate<0>("dense v_mfma_i32_32x32x32_i8", dr, dout, w);
        rate<1>("sparse v_smfmac_i32_32x32x64_i8", dr, dout, w);
        rate<2>("sparse, two chains", dr, dout, w);
    }
    return 0;
}
