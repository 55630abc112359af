// Weight stream of the register-chain forward kernel (mlp_fwd_r.hip).  The whole network is ONE sequence of 32-KiB chunks
//   [layer][tile group of 4 output tiles][k-chunk of 4 k-steps][tile][k-step][plane hi | lo][64 lanes][8 halfs]
// in the order the kernel consumes it, so that a workgroup streams it through a ring of LDS slots with LDS-DMA
// (buffer_load ... lds: a fragment is 64 lanes x 16 bytes = one instruction).  Fragment contents as in the split-f16 forward
// blocks (mlp_pack.hip): lane l of fragment (tile t, k-step ks) holds W[out = 32 t + (l & 31)][k = 16 ks + 8 (l >> 5) + j] in
// the layer's packed input order, hi = rn16(w), lo = rn16((w - hi) * 2^11).
#pragma once
#include "mlp_split.h"

namespace mlp {

enum RLayer { RL_L0 = 0, RL_L1, RL_L2, RL_L3, RL_L4, RL_L5, RL_L6, RL_L7, RL_FEAT, RL_VIEWS, RL_COUNT };
__host__ __device__ constexpr int r_groups(int l) { return l == RL_VIEWS ? 1 : 2; }
// k-chunks of 4 k-steps: L0 K = 64; L5 K = 256 + 64; VIEWS K = 256 + 32 padded to 320 (two all-zero k-steps)
__host__ __device__ constexpr int r_kchunks(int l) { return l == RL_L0 ? 1 : (l == RL_L5 || l == RL_VIEWS) ? 5 : 4; }
__host__ __device__ constexpr int r_chunk_base(int l) {
    int c = 0;
    for (int i = 0; i < l; ++i) c += r_groups(i) * r_kchunks(i);
    return c;
}
constexpr int R_CHUNKS = r_chunk_base(RL_COUNT);          // 73
constexpr int R_CHUNK_BYTES = 4 * 4 * 2 * 1024;           // 4 tiles x 4 k-steps x 2 planes x 1 KiB
constexpr int64_t R_FLOATS = (int64_t)R_CHUNKS * R_CHUNK_BYTES / 4;
static_assert(R_CHUNKS == 73, "chunk count of the weight stream");
// forward pack id (mlp_common.h) whose pack_source() addresses this layer's weights
__host__ __device__ constexpr int r_pack_id(int l) { return l == RL_FEAT ? PF_FEAT : l == RL_VIEWS ? PF_VIEWS : PF_L0 + l; }

}  // namespace mlp
