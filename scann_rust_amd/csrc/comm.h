// comm.h -- internals shared by comm.hip (the RCCL exchange) and api.hip.
#pragma once
#include <stdint.h>

#include "../../include/scann_hip.h"

namespace scann {

// Byte layout of one sharded step (see comm.hip): the batch is padded to qr * world queries.
struct CommLayout {
    uint32_t qr, nq_pad;                               // queries merged per rank, padded batch
    uint64_t cap;                                      // entries a destination block has room for
    // one destination block: [count u32[qr] | overflow flag u32 | pad | keys u64[cap] | idx u32[cap] | exact f32[cap]]
    uint64_t block_bytes, blk_count, blk_flag, blk_keys, blk_idx, blk_exact;
    uint64_t soa_bytes, soa_idx, soa_exact, soa_count;     // local-stage arrays [nq][m_local]
    uint64_t res_bytes, res_dist, res_count, res_status;   // result rows [nq_pad][k] x2 + counts + [world] status words
};
// worst_case: blocks with room for qr * m_local entries (m_local = 0 calls); else SCANN_HIP_COMM_FILL / world of that
CommLayout comm_layout(uint32_t nq, uint32_t world, uint32_t m_local, uint32_t k, bool worst_case);

// api.hip
int ctx_device(const scann_hip_ctx *ctx);
// pre-reorder candidate count m a search with (k, opts) on this index keeps (tree_x_hybrid/mod.rs:263)
int txh_resolve_m(scann_hip_index *index, uint32_t k, const scann_hip_search_opts *opts, uint32_t *out_m);

}  // namespace scann
