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

#include <algorithm>
#include <cmath>
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

// Fill factor of a destination block (SCANN_HIP_COMM_FILL, default 2.5; 0 = blocks sized for the worst case).
// A rank merges qr = nq / world queries and receives from every peer that peer's candidates of THOSE queries.  Sized
// for the worst case a block holds qr * m_local entries; but the candidates of one query total at most ~m over all
// ranks, so a peer's share averages m_local / world per query -- with shards that follow the data's clusters a query's
// leaves lie in ONE rank's shard: one block of the world is full for that query, the others are empty.  Blocks are
// therefore COMPACT (a count per query, the entries of the queries one behind the other) with room for fill / world of
// the worst case; a block that overflows is flagged, the merge reports Aborted on every rank, and the caller repeats the
// batch with m_local = 0 (worst-case blocks).
static double comm_fill() {   // (read per call: every rank must run with the same value)
    const char *e = std::getenv("SCANN_HIP_COMM_FILL");
    const double f = e ? std::atof(e) : 2.5;
    return f < 0.0 ? 0.0 : f;
}

// Layout of the exchange for nq queries over `world` ranks with m_local candidates per (rank, query).
CommLayout comm_layout(uint32_t nq, uint32_t world, uint32_t m_local, uint32_t k, bool worst_case) {
    CommLayout l;
    l.qr = (nq + world - 1) / world;
    l.nq_pad = l.qr * world;
    const uint64_t full = (uint64_t)l.qr * m_local;
    uint64_t cap = full;
    const double f = comm_fill();
    if (!worst_case && f > 0.0) {
        const uint64_t want = (uint64_t)std::ceil(f * (double)full / (double)world);
        cap = std::min<uint64_t>(full, std::max<uint64_t>(m_local, want));
    }
    l.cap = cap;
    l.blk_count = 0;                                            // [qr] entries of each query, then the overflow flag
    l.blk_flag = (uint64_t)l.qr * 4;
    l.blk_keys = (l.blk_flag + 4 + 15) & ~15ull;
    l.blk_idx = l.blk_keys + cap * 8;
    l.blk_exact = l.blk_idx + cap * 4;
    l.block_bytes = (l.blk_exact + cap * 4 + 15) & ~15ull;
    l.soa_idx = (uint64_t)nq * m_local * 8;
    l.soa_exact = l.soa_idx + (uint64_t)nq * m_local * 4;
    l.soa_count = l.soa_exact + (uint64_t)nq * m_local * 4;
    l.soa_bytes = l.soa_count + (uint64_t)nq * 4;
    l.res_dist = (uint64_t)l.nq_pad * k * 4;
    l.res_count = 2 * l.res_dist;
    l.res_status = l.res_count + (uint64_t)l.nq_pad * 4;        // [world] one status word per rank, gathered with the rows
    l.res_bytes = l.res_status + (uint64_t)world * 4;
    return l;
}

// Per destination d (one workgroup): entries of each of its queries in the block (count, cut by m_local and by the
// block's capacity), their first slot (exclusive prefix), the overflow flag.  soff[q] = first slot of query q in ITS
// destination's block (read by comm_pack_kernel).  Queries past nq (the batch padded to a multiple of the ranks)
// travel with count 0.
__global__ __launch_bounds__(256) void comm_offsets_kernel(uint32_t nq, uint32_t qr, uint32_t m_local, uint64_t cap,
                                                          const uint32_t *__restrict__ count, uint32_t *__restrict__ soff,
                                                          unsigned char *__restrict__ out, uint64_t block_bytes,
                                                          uint64_t blk_flag) {
    __shared__ uint32_t s_part[256];
    const uint32_t d = blockIdx.x, tid = threadIdx.x;
    uint32_t *hdr = reinterpret_cast<uint32_t *>(out + (uint64_t)d * block_bytes);
    const uint32_t per = (qr + 255u) / 256u;
    const uint32_t q0 = tid * per, q1 = min(q0 + per, qr);
    uint32_t mine = 0;
    for (uint32_t ql = q0; ql < q1; ++ql) {
        const uint32_t q = d * qr + ql;
        mine += q < nq ? min(count[q], m_local) : 0u;
    }
    s_part[tid] = mine;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t t = 0; t < tid; ++t) base += s_part[t];
    uint32_t overflow = 0;
    for (uint32_t ql = q0; ql < q1; ++ql) {
        const uint32_t q = d * qr + ql;
        const uint32_t c = q < nq ? min(count[q], m_local) : 0u;
        uint32_t sent = c;
        if ((uint64_t)base + c > cap) {
            sent = (uint64_t)base < cap ? (uint32_t)(cap - base) : 0u;
            overflow = 1;
        }
        hdr[ql] = sent;
        if (q < nq) soff[q] = base;
        base += c;
    }
    if (tid == 0) *reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(hdr) + blk_flag) = 0u;
    __syncthreads();
    if (overflow) *reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(hdr) + blk_flag) = 1u;
}

// Local-stage arrays [nq][m_local] -> the compact destination blocks (the slots comm_offsets_kernel assigned).
__global__ void comm_pack_kernel(uint32_t nq, uint32_t qr, uint32_t m, uint64_t cap,
                                 const uint64_t *__restrict__ keys, const uint32_t *__restrict__ idx,
                                 const float *__restrict__ exact, const uint32_t *__restrict__ count,
                                 const uint32_t *__restrict__ soff, unsigned char *__restrict__ out, uint64_t block_bytes,
                                 uint64_t blk_keys, uint64_t blk_idx, uint64_t blk_exact) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)nq * m) return;
    const uint32_t q = (uint32_t)(e / m), i = (uint32_t)(e - (uint64_t)q * m);
    if (i >= count[q]) return;
    const uint64_t slot = (uint64_t)soff[q] + i;
    if (slot >= cap) return;   // (overflow: flagged by comm_offsets_kernel)
    unsigned char *blk = out + (uint64_t)(q / qr) * block_bytes;
    reinterpret_cast<uint64_t *>(blk + blk_keys)[slot] = keys[e];
    reinterpret_cast<uint32_t *>(blk + blk_idx)[slot] = idx[e];
    reinterpret_cast<float *>(blk + blk_exact)[slot] = exact[e];
}

// Receiver: first slot of (source rank g, query ql) in g's block = prefix of the received counts; an overflow flag in
// any received block makes this rank's step status Aborted.
__global__ __launch_bounds__(256) void comm_recv_offsets_kernel(uint32_t qr, const unsigned char *__restrict__ recv,
                                                               uint64_t block_bytes, uint64_t blk_flag,
                                                               uint32_t *__restrict__ roff, uint32_t *__restrict__ status) {
    __shared__ uint32_t s_part[256];
    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(recv + (uint64_t)g * block_bytes);
    const uint32_t per = (qr + 255u) / 256u;
    const uint32_t q0 = tid * per, q1 = min(q0 + per, qr);
    uint32_t mine = 0;
    for (uint32_t ql = q0; ql < q1; ++ql) mine += hdr[ql];
    s_part[tid] = mine;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t t = 0; t < tid; ++t) base += s_part[t];
    for (uint32_t ql = q0; ql < q1; ++ql) {
        roff[(size_t)g * qr + ql] = base;
        base += hdr[ql];
    }
    if (tid == 0 && *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(hdr) + blk_flag))
        atomicMax(status, (uint32_t)SCANN_HIP_ABORTED);
}

// After the all-gather of the rows and the ranks' step status words: every rank ends with the same status.
__global__ void comm_status_kernel(uint32_t world, const uint32_t *__restrict__ step_status, uint32_t *__restrict__ status) {
    uint32_t v = 0;
    for (uint32_t g = 0; g < world; ++g) v = max(v, step_status[g]);
    if (v) atomicMax(status, v);
}

}  // namespace scann

using namespace scann;

struct scann_hip_comm {
    scann_hip_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t cstream = nullptr;            // the exchange's own stream
    DevBuf soa[2], send[2], recv[2], res[2], soff[2], roff[2], status;
    hipEvent_t ev_packed[2] = {}, ev_done[2] = {};
    bool done_valid[2] = {false, false};
    uint64_t calls = 0;
    std::mutex mu;
};

extern "C" {

int scann_hip_comm_layout(uint32_t nq, uint32_t world, uint32_t m_local, uint32_t k, uint64_t *out) {
    if (!out || world == 0 || m_local == 0 || k == 0) return fail(SCANN_HIP_INVALID_ARGUMENT, "bad arguments");
    const CommLayout l = comm_layout(nq, world, m_local, k, false);
    const uint64_t v[16] = {l.qr, l.nq_pad, l.block_bytes, l.blk_idx, l.blk_exact, l.blk_count,
                            l.soa_bytes, l.soa_idx, l.soa_exact, l.soa_count, l.res_bytes, l.res_dist,
                            l.blk_keys, l.cap, l.blk_flag, l.res_status};
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
    // m_local = 0: the safe form -- every rank's full best-m list, blocks sized for the worst case
    const bool worst_case = m_local_in == 0;
    const uint32_t m_local = (m_local_in == 0 || m_local_in > m) ? m : m_local_in;
    const uint32_t world = (uint32_t)c->world, rank = (uint32_t)c->rank;
    hipStream_t S = static_cast<hipStream_t>(hip_stream), C = c->cstream;
    std::lock_guard<std::mutex> lock(c->mu);
    SCANN_HIP_CHECK(hipSetDevice(ctx_device(c->ctx)));
    const CommLayout L = comm_layout(nq, world, m_local, k, worst_case);
    const int b = (int)(c->calls & 1u);
    SCANN_TRY(c->soa[b].ensure(L.soa_bytes));
    SCANN_TRY(c->send[b].ensure(L.block_bytes * world));
    SCANN_TRY(c->recv[b].ensure(L.block_bytes * world));
    SCANN_TRY(c->res[b].ensure(L.res_bytes));
    SCANN_TRY(c->soff[b].ensure((size_t)L.nq_pad * 4));
    SCANN_TRY(c->roff[b].ensure((size_t)L.nq_pad * 4));

    // ---- on the caller's stream: local stage -> destination blocks -----------------------------
    // This parity's buffers belong to the exchange issued two calls ago.  (The index workspace is bound to the
    // caller's stream by the library: two caller streams run two local stages side by side.)
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
    unsigned char *snd = c->send[b].as<unsigned char>(), *rcv = c->recv[b].as<unsigned char>();
    {
        const uint32_t *cnt = reinterpret_cast<const uint32_t *>(soa + L.soa_count);
        hipLaunchKernelGGL(comm_offsets_kernel, dim3(world), dim3(256), 0, S, nq, L.qr, m_local, L.cap, cnt,
                           c->soff[b].as<uint32_t>(), snd, L.block_bytes, L.blk_flag);
        if (hipGetLastError() != hipSuccess) return fail(SCANN_HIP_INTERNAL, "offsets kernel launch failed");
        const uint64_t work = (uint64_t)nq * m_local;
        hipLaunchKernelGGL(comm_pack_kernel, dim3((uint32_t)ceil_div_u64(work, 256)), dim3(256), 0, S, nq, L.qr,
                           m_local, L.cap, reinterpret_cast<const uint64_t *>(soa),
                           reinterpret_cast<const uint32_t *>(soa + L.soa_idx),
                           reinterpret_cast<const float *>(soa + L.soa_exact), cnt, c->soff[b].as<uint32_t>(), snd,
                           L.block_bytes, L.blk_keys, L.blk_idx, L.blk_exact);
        if (hipGetLastError() != hipSuccess) return fail(SCANN_HIP_INTERNAL, "pack kernel launch failed");
    }
    SCANN_HIP_CHECK(hipEventRecord(c->ev_packed[b], S));

    // ---- on the communicator's stream: all-to-all -> merge -> all-gather -> caller's buffers -------
    // From here on every rank has committed to the collective: an error must still close the RCCL group (an open
    // group would swallow the next call's operations) and is returned after the step's enqueue is complete or
    // abandoned as a whole -- calls / done_valid advance only on success.
    SCANN_HIP_CHECK(hipStreamWaitEvent(C, c->ev_packed[b], 0));
    unsigned char *res = c->res[b].as<unsigned char>();
    uint32_t *r_idx = reinterpret_cast<uint32_t *>(res);
    float *r_dist = reinterpret_cast<float *>(res + L.res_dist);
    uint32_t *r_cnt = reinterpret_cast<uint32_t *>(res + L.res_count);
    uint32_t *r_status = reinterpret_cast<uint32_t *>(res + L.res_status);
    SCANN_HIP_CHECK(hipMemsetAsync(r_status + rank, 0, 4, C));
    ncclResult_t first = ncclSuccess;
    const char *where = "";
    auto note = [&](ncclResult_t e, const char *w) {
        if (e != ncclSuccess && first == ncclSuccess) {
            first = e;
            where = w;
        }
    };
    note(r->GroupStart(), "ncclGroupStart");
    if (first == ncclSuccess) {
        for (uint32_t p = 0; p < world && first == ncclSuccess; ++p) {
            note(r->Send(snd + (uint64_t)p * L.block_bytes, L.block_bytes, ncclUint8, (int)p, c->comm, C), "ncclSend");
            note(r->Recv(rcv + (uint64_t)p * L.block_bytes, L.block_bytes, ncclUint8, (int)p, c->comm, C), "ncclRecv");
        }
        note(r->GroupEnd(), "ncclGroupEnd");   // (always: also after a failed Send / Recv)
    }
    if (first != ncclSuccess) return fail(SCANN_HIP_INTERNAL, std::string(where) + ": " + r->GetErrorString(first));
    hipLaunchKernelGGL(comm_recv_offsets_kernel, dim3(world), dim3(256), 0, C, L.qr, rcv, L.block_bytes, L.blk_flag,
                       c->roff[b].as<uint32_t>(), r_status + rank);
    if (hipGetLastError() != hipSuccess) return fail(SCANN_HIP_INTERNAL, "receive offsets kernel launch failed");
    int ms = txh_launch_merge(world, L.qr, m_local, m, k, (size_t)L.block_bytes,
                              reinterpret_cast<const uint64_t *>(rcv + L.blk_keys),
                              reinterpret_cast<const uint32_t *>(rcv + L.blk_idx),
                              reinterpret_cast<const float *>(rcv + L.blk_exact),
                              reinterpret_cast<const uint32_t *>(rcv + L.blk_count),
                              r_idx + (uint64_t)rank * L.qr * k, r_dist + (uint64_t)rank * L.qr * k,
                              r_cnt + (uint64_t)rank * L.qr, r_status + rank, C, c->roff[b].as<uint32_t>());
    // (a failed merge launch still takes part in the all-gathers below: the peers are already waiting in them)
    note(r->GroupStart(), "ncclGroupStart");   // in-place all-gathers: rank g's rows land at g * qr
    if (first == ncclSuccess) {
        note(r->AllGather(r_idx + (uint64_t)rank * L.qr * k, r_idx, (size_t)L.qr * k, ncclUint32, c->comm, C), "ncclAllGather");
        note(r->AllGather(r_dist + (uint64_t)rank * L.qr * k, r_dist, (size_t)L.qr * k, ncclFloat32, c->comm, C), "ncclAllGather");
        note(r->AllGather(r_cnt + (uint64_t)rank * L.qr, r_cnt, (size_t)L.qr, ncclUint32, c->comm, C), "ncclAllGather");
        note(r->AllGather(r_status + rank, r_status, 1, ncclUint32, c->comm, C), "ncclAllGather");
        note(r->GroupEnd(), "ncclGroupEnd");
    }
    if (first != ncclSuccess) return fail(SCANN_HIP_INTERNAL, std::string(where) + ": " + r->GetErrorString(first));
    if (ms != SCANN_HIP_OK) return ms;
    hipLaunchKernelGGL(comm_status_kernel, dim3(1), dim3(1), 0, C, world, r_status, c->status.as<uint32_t>());
    if (hipGetLastError() != hipSuccess) return fail(SCANN_HIP_INTERNAL, "status kernel launch failed");
    SCANN_HIP_CHECK(hipMemcpyAsync(d_out_idx, r_idx, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, C));
    SCANN_HIP_CHECK(hipMemcpyAsync(d_out_dist, r_dist, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, C));
    SCANN_HIP_CHECK(hipMemcpyAsync(d_out_count, r_cnt, (size_t)nq * 4, hipMemcpyDeviceToDevice, C));
    SCANN_HIP_CHECK(hipEventRecord(c->ev_done[b], C));
    c->done_valid[b] = true;
    ++c->calls;
    // the caller's stream sees the results; work it enqueues on ANOTHER stream meanwhile overlaps
    SCANN_HIP_CHECK(hipStreamWaitEvent(S, c->ev_done[b], 0));
    return SCANN_HIP_OK;
}

}  // extern "C"
