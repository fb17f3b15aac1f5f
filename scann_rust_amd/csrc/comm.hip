// comm.hip -- the multi-GPU exchange of the leaf-sharded Tree-X-Hybrid search, inside the library
// (include/scann_hip.h "multi-GPU exchange").  One process per GPU, RCCL over xGMI.
//
// The reference's only parallelism on this path is rayon over partitions and queries
// (tree_x_hybrid/mod.rs:266-280, 399-409); the merge it performs after the per-partition scans
// (flatten in token order, stable sort by approximate distance, truncate to pre_reorder_k, exact
// re-rank, stable sort, truncate to k: mod.rs:283-293, 342-364) is what the exchange below
// distributes: every rank scans the leaves it owns, ONE all-to-all hands each rank the candidates
// of the queries it merges, and an all-gather returns the k result rows to everybody.
//
// librccl.so.1 is loaded with dlopen on first use (the library has no link-time dependency on it:
// single-GPU hosts never load it; a process that already holds an RCCL -- PyTorch -- shares it).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "comm.h"
#include "common.h"
#include "txh.h"

namespace scann {

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {std::getenv("SCANN_HIP_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1",
                               "librccl.so"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            r.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

int rccl_ready(Rccl **out) {
    Rccl *r = rccl();
    if (!r->handle || !r->error.empty()) return fail(SCANN_HIP_UNAVAILABLE, r->error);
    *out = r;
    return SCANN_HIP_OK;
}

#define RCCL_CHECK(r, expr)                                                                       \
    do {                                                                                          \
        ncclResult_t _e = (expr);                                                                 \
        if (_e != ncclSuccess)                                                                    \
            return fail(SCANN_HIP_INTERNAL, std::string(#expr) + ": " + (r)->GetErrorString(_e)); \
    } while (0)

}  // namespace

// Layout of the exchange for nq queries over `world` ranks with m_local candidates per (rank, query).
CommLayout comm_layout(uint32_t nq, uint32_t world, uint32_t m_local, uint32_t k) {
    CommLayout l;
    l.qr = (nq + world - 1) / world;
    l.nq_pad = l.qr * world;
    const uint64_t per = (uint64_t)l.qr * m_local;
    l.blk_idx = per * 8;
    l.blk_exact = per * 12;
    l.blk_count = per * 16;
    l.block_bytes = (per * 16 + (uint64_t)l.qr * 4 + 15) & ~15ull;
    l.soa_idx = (uint64_t)nq * m_local * 8;
    l.soa_exact = l.soa_idx + (uint64_t)nq * m_local * 4;
    l.soa_count = l.soa_exact + (uint64_t)nq * m_local * 4;
    l.soa_bytes = l.soa_count + (uint64_t)nq * 4;
    l.res_dist = (uint64_t)l.nq_pad * k * 4;
    l.res_count = 2 * l.res_dist;
    l.res_bytes = l.res_count + (uint64_t)l.nq_pad * 4;
    return l;
}

// Local-stage arrays [nq][m] -> one block per destination rank; block d = the queries
// [d * qr, (d + 1) * qr): [keys u64 | idx u32 | exact f32 | count u32].  Queries past nq (the batch
// padded to a multiple of the ranks) are sent with count 0.
__global__ void comm_pack_kernel(uint32_t nq, uint32_t qr, uint32_t world, uint32_t m,
                                 const uint64_t *__restrict__ keys, const uint32_t *__restrict__ idx,
                                 const float *__restrict__ exact, const uint32_t *__restrict__ count,
                                 unsigned char *__restrict__ out, uint64_t block_bytes) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per = (uint64_t)qr * m;
    if (e < (uint64_t)nq * m) {
        const uint32_t q = (uint32_t)(e / m), i = (uint32_t)(e - (uint64_t)q * m);
        const uint32_t d = q / qr, ql = q - d * qr;
        unsigned char *blk = out + (uint64_t)d * block_bytes;
        const uint64_t slot = (uint64_t)ql * m + i;
        reinterpret_cast<uint64_t *>(blk)[slot] = keys[e];
        reinterpret_cast<uint32_t *>(blk + per * 8)[slot] = idx[e];
        reinterpret_cast<float *>(blk + per * 12)[slot] = exact[e];
    }
    if (e < (uint64_t)qr * world) {
        const uint32_t q = (uint32_t)e, d = q / qr, ql = q - d * qr;
        reinterpret_cast<uint32_t *>(out + (uint64_t)d * block_bytes + per * 16)[ql] = q < nq ? count[q] : 0u;
    }
}

}  // namespace scann

using namespace scann;

struct scann_hip_comm {
    scann_hip_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t cstream = nullptr;            // the exchange's own stream
    DevBuf soa[2], send[2], recv[2], res[2], status;
    hipEvent_t ev_packed[2] = {}, ev_done[2] = {}, ev_local = nullptr;
    bool done_valid[2] = {false, false}, local_valid = false;
    uint64_t calls = 0;
    std::mutex mu;
};

extern "C" {

int scann_hip_comm_layout(uint32_t nq, uint32_t world, uint32_t m_local, uint32_t k, uint64_t *out) {
    if (!out || world == 0 || m_local == 0 || k == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "bad arguments");
    const CommLayout l = comm_layout(nq, world, m_local, k);
    const uint64_t v[12] = {l.qr, l.nq_pad, l.block_bytes, l.blk_idx, l.blk_exact, l.blk_count,
                            l.soa_bytes, l.soa_idx, l.soa_exact, l.soa_count, l.res_bytes, l.res_dist};
    std::memcpy(out, v, sizeof(v));
    return SCANN_HIP_OK;
}

int scann_hip_comm_unique_id(void *out_id) {
    if (!out_id) return fail(SCANN_HIP_INVALID_ARGUMENT, "out_id is null");
    Rccl *r = nullptr;
    SCANN_TRY(rccl_ready(&r));
    static_assert(sizeof(ncclUniqueId) == SCANN_HIP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    RCCL_CHECK(r, r->GetUniqueId(&id));
    std::memcpy(out_id, &id, sizeof(id));
    return SCANN_HIP_OK;
}

int scann_hip_comm_create(scann_hip_ctx *ctx, const void *unique_id, int rank, int world, scann_hip_comm **out) {
    if (!ctx || !unique_id || !out) return fail(SCANN_HIP_INVALID_ARGUMENT, "null ctx/unique_id/out_comm");
    if (world < 1 || world > 64 || rank < 0 || rank >= world)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "need 0 <= rank < world <= 64");
    Rccl *r = nullptr;
    SCANN_TRY(rccl_ready(&r));
    SCANN_HIP_CHECK(hipSetDevice(ctx_device(ctx)));
    auto *c = new scann_hip_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    auto bail = [&](int s) {
        scann_hip_comm_destroy(c);
        return s;
    };
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclResult_t e = r->CommInitRank(&c->comm, world, id, rank);
    if (e != ncclSuccess) return bail(fail(SCANN_HIP_INTERNAL, std::string("ncclCommInitRank: ") + r->GetErrorString(e)));
    if (hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(SCANN_HIP_INTERNAL, "hipStreamCreate failed"));
    for (int b = 0; b < 2; ++b)
        if (hipEventCreateWithFlags(&c->ev_packed[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_done[b], hipEventDisableTiming) != hipSuccess)
            return bail(fail(SCANN_HIP_INTERNAL, "hipEventCreate failed"));
    if (hipEventCreateWithFlags(&c->ev_local, hipEventDisableTiming) != hipSuccess)
        return bail(fail(SCANN_HIP_INTERNAL, "hipEventCreate failed"));
    int s = c->status.ensure(4);
    if (s != SCANN_HIP_OK) return bail(s);
    if (hipMemset(c->status.p, 0, 4) != hipSuccess) return bail(fail(SCANN_HIP_INTERNAL, "hipMemset failed"));
    *out = c;
    return SCANN_HIP_OK;
}

void scann_hip_comm_destroy(scann_hip_comm *c) {
    if (!c) return;
    (void)hipSetDevice(ctx_device(c->ctx));
    if (c->cstream) (void)hipStreamSynchronize(c->cstream);
    Rccl *r = rccl();
    if (c->comm && r->CommDestroy) (void)r->CommDestroy(c->comm);
    for (int b = 0; b < 2; ++b) {
        if (c->ev_packed[b]) (void)hipEventDestroy(c->ev_packed[b]);
        if (c->ev_done[b]) (void)hipEventDestroy(c->ev_done[b]);
    }
    if (c->ev_local) (void)hipEventDestroy(c->ev_local);
    if (c->cstream) (void)hipStreamDestroy(c->cstream);
    delete c;
}

int scann_hip_comm_last_status(scann_hip_comm *c) {
    if (!c) return fail(SCANN_HIP_INVALID_ARGUMENT, "comm is null");
    std::lock_guard<std::mutex> lock(c->mu);
    SCANN_HIP_CHECK(hipSetDevice(ctx_device(c->ctx)));
    uint32_t st = 0;
    SCANN_HIP_CHECK(hipMemcpyAsync(&st, c->status.p, 4, hipMemcpyDeviceToHost, c->cstream));
    SCANN_HIP_CHECK(hipMemsetAsync(c->status.p, 0, 4, c->cstream));
    SCANN_HIP_CHECK(hipStreamSynchronize(c->cstream));
    if (st != SCANN_HIP_OK)
        return fail((int)st, "a rank's m_local candidate list was too short for the global best pre_reorder_k; "
                             "repeat the batch with m_local = 0");
    return SCANN_HIP_OK;
}

int scann_hip_txh_search_sharded_device(scann_hip_index *index, scann_hip_comm *c, const float *d_queries,
                                        uint32_t nq, uint32_t q_stride, uint32_t k,
                                        const scann_hip_search_opts *opts, uint32_t m_local_in,
                                        uint32_t *d_out_idx, float *d_out_dist, uint32_t *d_out_count,
                                        void *hip_stream) {
    if (!index || !c) return fail(SCANN_HIP_INVALID_ARGUMENT, "index/comm is null");
    if (nq == 0) return SCANN_HIP_OK;
    if (k == 0 || !d_queries || !d_out_idx || !d_out_dist || !d_out_count)
        return fail(SCANN_HIP_INVALID_ARGUMENT, "k must be > 0 and the buffers non-null");
    Rccl *r = nullptr;
    SCANN_TRY(rccl_ready(&r));
    uint32_t m = 0;
    SCANN_TRY(txh_resolve_m(index, k, opts, &m));
    if (m == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "pre-reorder candidate count is 0");
    const uint32_t m_local = (m_local_in == 0 || m_local_in > m) ? m : m_local_in;
    const uint32_t world = (uint32_t)c->world, rank = (uint32_t)c->rank;
    hipStream_t S = static_cast<hipStream_t>(hip_stream), C = c->cstream;
    std::lock_guard<std::mutex> lock(c->mu);
    SCANN_HIP_CHECK(hipSetDevice(ctx_device(c->ctx)));
    const CommLayout L = comm_layout(nq, world, m_local, k);
    const int b = (int)(c->calls & 1u);
    SCANN_TRY(c->soa[b].ensure(L.soa_bytes));
    SCANN_TRY(c->send[b].ensure(L.block_bytes * world));
    SCANN_TRY(c->recv[b].ensure(L.block_bytes * world));
    SCANN_TRY(c->res[b].ensure(L.res_bytes));
    ++c->calls;

    // ---- on the caller's stream: local stage -> destination blocks -----------------------------
    // The index workspace belongs to one local stage at a time (the previous call may have run on
    // another stream), and this parity's buffers to the exchange issued two calls ago.
    if (c->local_valid) SCANN_HIP_CHECK(hipStreamWaitEvent(S, c->ev_local, 0));
    if (c->done_valid[b]) SCANN_HIP_CHECK(hipStreamWaitEvent(S, c->ev_done[b], 0));
    unsigned char *soa = c->soa[b].as<unsigned char>();
    scann_hip_search_opts lo;
    if (opts) lo = *opts; else scann_hip_search_opts_default(&lo);
    lo.pre_reorder_k = m_local;
    lo.exact_reorder = 1;
    SCANN_TRY(scann_hip_txh_search_local_device(index, d_queries, nq, q_stride, k, &lo,
                                                reinterpret_cast<uint64_t *>(soa),
                                                reinterpret_cast<uint32_t *>(soa + L.soa_idx),
                                                reinterpret_cast<float *>(soa + L.soa_exact),
                                                reinterpret_cast<uint32_t *>(soa + L.soa_count), S));
    {
        const uint64_t work = std::max<uint64_t>((uint64_t)nq * m_local, L.nq_pad);
        hipLaunchKernelGGL(comm_pack_kernel, dim3((uint32_t)ceil_div_u64(work, 256)), dim3(256), 0, S, nq, L.qr, world,
                           m_local, reinterpret_cast<const uint64_t *>(soa),
                           reinterpret_cast<const uint32_t *>(soa + L.soa_idx),
                           reinterpret_cast<const float *>(soa + L.soa_exact),
                           reinterpret_cast<const uint32_t *>(soa + L.soa_count), c->send[b].as<unsigned char>(),
                           L.block_bytes);
        if (hipGetLastError() != hipSuccess) return fail(SCANN_HIP_INTERNAL, "pack kernel launch failed");
    }
    SCANN_HIP_CHECK(hipEventRecord(c->ev_packed[b], S));
    SCANN_HIP_CHECK(hipEventRecord(c->ev_local, S));
    c->local_valid = true;

    // ---- on the communicator's stream: all-to-all -> merge -> all-gather -> caller's buffers -------
    SCANN_HIP_CHECK(hipStreamWaitEvent(C, c->ev_packed[b], 0));
    unsigned char *snd = c->send[b].as<unsigned char>(), *rcv = c->recv[b].as<unsigned char>();
    RCCL_CHECK(r, r->GroupStart());
    for (uint32_t p = 0; p < world; ++p) {
        RCCL_CHECK(r, r->Send(snd + (uint64_t)p * L.block_bytes, L.block_bytes, ncclUint8, (int)p, c->comm, C));
        RCCL_CHECK(r, r->Recv(rcv + (uint64_t)p * L.block_bytes, L.block_bytes, ncclUint8, (int)p, c->comm, C));
    }
    RCCL_CHECK(r, r->GroupEnd());
    unsigned char *res = c->res[b].as<unsigned char>();
    uint32_t *r_idx = reinterpret_cast<uint32_t *>(res);
    float *r_dist = reinterpret_cast<float *>(res + L.res_dist);
    uint32_t *r_cnt = reinterpret_cast<uint32_t *>(res + L.res_count);
    SCANN_TRY(txh_launch_merge(world, L.qr, m_local, m, k, (size_t)L.block_bytes,
                               reinterpret_cast<const uint64_t *>(rcv),
                               reinterpret_cast<const uint32_t *>(rcv + L.blk_idx),
                               reinterpret_cast<const float *>(rcv + L.blk_exact),
                               reinterpret_cast<const uint32_t *>(rcv + L.blk_count),
                               r_idx + (uint64_t)rank * L.qr * k, r_dist + (uint64_t)rank * L.qr * k,
                               r_cnt + (uint64_t)rank * L.qr, c->status.as<uint32_t>(), C));
    RCCL_CHECK(r, r->GroupStart());   // in-place all-gathers: rank g's rows land at g * qr
    RCCL_CHECK(r, r->AllGather(r_idx + (uint64_t)rank * L.qr * k, r_idx, (size_t)L.qr * k, ncclUint32, c->comm, C));
    RCCL_CHECK(r, r->AllGather(r_dist + (uint64_t)rank * L.qr * k, r_dist, (size_t)L.qr * k, ncclFloat32, c->comm, C));
    RCCL_CHECK(r, r->AllGather(r_cnt + (uint64_t)rank * L.qr, r_cnt, (size_t)L.qr, ncclUint32, c->comm, C));
    RCCL_CHECK(r, r->GroupEnd());
    SCANN_HIP_CHECK(hipMemcpyAsync(d_out_idx, r_idx, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, C));
    SCANN_HIP_CHECK(hipMemcpyAsync(d_out_dist, r_dist, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, C));
    SCANN_HIP_CHECK(hipMemcpyAsync(d_out_count, r_cnt, (size_t)nq * 4, hipMemcpyDeviceToDevice, C));
    SCANN_HIP_CHECK(hipEventRecord(c->ev_done[b], C));
    c->done_valid[b] = true;
    // the caller's stream sees the results; work it enqueues on ANOTHER stream meanwhile overlaps
    SCANN_HIP_CHECK(hipStreamWaitEvent(S, c->ev_done[b], 0));
    return SCANN_HIP_OK;
}

}  // extern "C"
