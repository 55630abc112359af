// Shared layout definitions for the fused NeRF MLP kernels (K3).
//
// Network (model/nerf.py:41-64,67-116): PE(63) -> 8 x 256 ReLU (layer 5 consumes
// cat[PE, h4]) -> { alpha: 256->1 ; feature: 256->256 (linear) } ->
// views: cat[feature, PE_dir(27)] -> 128 ReLU -> rgb: 128 -> C.
//
// Arithmetic: exact-f32 MFMA (v_mfma_f32_32x32x2_f32), f32 accumulate.  A point tile is
// TM = 64 sample points; its activations live in ONE LDS tile Hs[64][LD] for the whole
// network:  columns [0,256) hidden state, [256,320) PE(pts) (63 + 1 zero pad; later reused
// for PE(dir), 27 + 5 zero pad).  Weights never touch LDS: every wave streams its own
// MFMA-shaped slices straight from L2 (pre-packed so that one global_load_dwordx4 per lane
// = the B operand of four consecutive MFMA k-steps).
#pragma once
#include "common.h"

namespace mlp {

constexpr int TM = 64;        // sample points per workgroup tile
constexpr int LD = 320;       // LDS row stride in floats: the tile is exactly 80 KiB -> 2 workgroups per CU
constexpr int COL_PE = 256;   // first PE column of the LDS tile
constexpr int COL_SCR = 288;  // 32 scratch columns (dead PE columns) for small per-point reductions
constexpr int NTHREADS = 256; // 4 wavefronts
constexpr size_t TILE_SMEM = (size_t)TM * LD * sizeof(float);   // 81 920 B

// Bank-conflict-free addressing without padding: element (row, col) of the tile lives at
// row*LD + (col ^ ((row & 15) << 2)).  The XOR permutes 16-byte slots inside each 64-column block,
// so float4 groups stay contiguous, the MFMA A-operand ds_read_b128 (16 distinct rows per lane
// group, one 4-column slot) and the epilogue ds_write_b32 (one row, 32 consecutive columns) are
// both conflict free.
__device__ __forceinline__ int swz(int row) { return (row & 15) << 2; }
__device__ __forceinline__ int tidx(int row, int col) { return row * LD + (col ^ swz(row)); }

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ---- saved activations (floats per point) ------------------------------------------------
constexpr int ACT_PE_W = 64, ACT_HV_W = 128, ACT_PED_W = 32;
constexpr int ACT_PER_POINT = ACT_PE_W + 8 * 256 + 256 + ACT_HV_W + ACT_PED_W;   // 2528
constexpr int DACT_PER_POINT = 8 * 256 + 256 + 128;                               // 2432
__host__ __device__ inline int64_t act_pe(int64_t M) { (void)M; return 0; }
__host__ __device__ inline int64_t act_h(int64_t M, int l) { return M * ACT_PE_W + (int64_t)l * M * 256; }
__host__ __device__ inline int64_t act_feat(int64_t M) { return M * ACT_PE_W + 8 * M * 256; }
__host__ __device__ inline int64_t act_hv(int64_t M) { return act_feat(M) + M * 256; }
__host__ __device__ inline int64_t act_ped(int64_t M) { return act_hv(M) + M * ACT_HV_W; }
// ReLU sign bits of h0..h7 in ACCUMULATOR layout: one uint64 per (layer, tile, thread); bit
// ((c*2 + r)*16 + e) <-> accumulator element e of row tile r, column tile wave*2 + c.  The
// backward kernel uses the same tiling, so each lane reads back exactly its own 64 bits.
__host__ __device__ inline int64_t n_tiles(int64_t M) { return (M + TM - 1) / TM; }
__host__ __device__ inline int64_t act_mask(int64_t M) { return act_ped(M) + M * ACT_PED_W; }   // float offset, 8-B aligned
__host__ __device__ inline int64_t act_total_floats(int64_t M) { return act_mask(M) + 8 * n_tiles(M) * NTHREADS * 2; }
__host__ __device__ inline int64_t dact_h(int64_t M, int l) { return (int64_t)l * M * 256; }
__host__ __device__ inline int64_t dact_feat(int64_t M) { return 8 * M * 256; }
__host__ __device__ inline int64_t dact_hv(int64_t M) { return 9 * M * 256; }

// ---- packed weights ----------------------------------------------------------------------
// A packed block is [n_tiles][k_blocks][64 lanes][4 floats]: lane l of tile t, block kb holds
// element (col = t*32 + (l&31), k = kb*8 + 4*(l>>5) + i), i = 0..3.  MFMA step i of block
// kb contracts k-pair (kb*8+i, kb*8+4+i); the LDS A operand uses the same pairing.
enum PackId {
    PF_L0 = 0, PF_L1, PF_L2, PF_L3, PF_L4, PF_L5, PF_L6, PF_L7, PF_FEAT, PF_VIEWS,   // forward: col = output feature
    PB_VIEWS, PB_FEAT, PB_L7, PB_L6, PB_L5, PB_L4, PB_L3, PB_L2, PB_L1, PB_L0,       // backward: col = input feature
    PB_VIEWSPE,   // PE(dir) slice of the views weights as [4 thread groups][128 n][8]: (g, n, q) = Wv[n][256 + g + 4q]
    // Round 5, split-f16 kernels: the feature layer (model/nerf.py:100-106: feature = W_f h7 + b_f, fed to the views layer WITHOUT a
    // ReLU) folded into the views layer - W_c = W_v[:, :256] W_f, b_c = W_v[:, :256] b_f + b_v, computed in float64 at every
    // re-pack (fuse_views_kernel) - so the split forward runs VIEWS straight on h7 and the split dX chain goes from dhv to dh7 in
    // one stage: a whole 256 x 256 GEMM stage less in each.  Same shapes as PF_VIEWS / PB_VIEWS, source [W_c | W_v[:, 256:283]].
    PF_VIEWSC, PB_VIEWSC,
    PACK_COUNT
};
__host__ __device__ constexpr bool pack_is_forward(int id) { return id <= PF_VIEWS || id == PF_VIEWSC; }
// fused matrix [128][283] + fused bias [128] (f32), behind the two packed sections of a network's buffer
constexpr int64_t FUSED_W_FLOATS = 128 * 283, FUSED_FLOATS = (FUSED_W_FLOATS + 128 + 63) / 64 * 64;
struct PackShape { int tiles, kblocks; };
__host__ __device__ constexpr PackShape pack_shape(int id) {
    switch (id) {
        case PF_L0: return {8, 8};        // K 63 -> 64
        case PF_L5: return {8, 40};       // K [h4 256 | PE 63 -> 64]
        case PF_VIEWS: case PF_VIEWSC: return {4, 36};    // N 128, K [feature (fused: h7) 256 | PE_dir 27 -> 32]
        case PB_VIEWS: case PB_VIEWSC: return {10, 16};   // out 256 feature (fused: h7) cols + tile 8 = the 27 PE(dir) cols (tile 9: zero pad), contraction n = 128
        case PB_L5: return {10, 32};      // out [h4 256 | PE 64]
        case PB_L0: return {2, 32};       // out PE 64
        case PB_VIEWSPE: return {1, 16};  // 4 * 128 * 8 floats, own layout
        default: return {8, 32};
    }
}
__host__ __device__ constexpr int64_t pack_floats(int id) {
    return (int64_t)pack_shape(id).tiles * pack_shape(id).kblocks * 256;
}
__host__ __device__ constexpr int64_t pack_offset(int id) {
    int64_t o = 0;
    for (int i = 0; i < id; ++i) o += pack_floats(i);
    return o;
}
constexpr int64_t PACKED_FLOATS = pack_offset(PACK_COUNT);

// nn.Linear source of every packed block (index into BenerfMlpParams.w)
__host__ __device__ constexpr int pack_layer(int id) {
    switch (id) {
        case PF_FEAT: case PB_FEAT: return BENERF_L_FEAT;
        case PF_VIEWS: case PB_VIEWS: case PB_VIEWSPE: case PF_VIEWSC: case PB_VIEWSC: return BENERF_L_VIEWS;
        case PB_L7: return 7; case PB_L6: return 6; case PB_L5: return 5; case PB_L4: return 4;
        case PB_L3: return 3; case PB_L2: return 2; case PB_L1: return 1; case PB_L0: return 0;
        default: return id;   // PF_L0..PF_L7
    }
}
__host__ __device__ constexpr int layer_in(int l) {
    return l == 0 ? 63 : l == 5 ? 319 : l == BENERF_L_VIEWS ? 283 : l == BENERF_L_RGB ? 128 : 256;
}
__host__ __device__ constexpr int layer_out(int l, int C) {
    return l == BENERF_L_VIEWS ? 128 : l == BENERF_L_ALPHA ? 1 : l == BENERF_L_RGB ? C : 256;
}

// ---- weight-gradient workspace -------------------------------------------------------------
// GEMM instances of the dW kernel (dW = dY^T X over all points), each split DW_SPLITS ways
// along the point dimension; partials are summed in fixed order by the reduce kernel.
enum DwInst { DW_L1 = 0, DW_L2, DW_L3, DW_L4, DW_L5H, DW_L6, DW_L7, DW_FEAT, DW_VIEWSF, DW_L0, DW_L5P, DW_VIEWSP, DW_RGB, DW_COUNT };
// Split counts (exact-f32 kernel, mlp_dw.hip): 512 workgroups that run as two rounds of one per CU, so every workgroup
// should take the same time.  Counts are proportional to MEASURED workgroup-time per instance (tools/experiments/trace_dw.py
// on a tracing build, 522 k points; round 3 - the FLOP-proportional table 8*52 + 28 + 14 + 14 + 16 + 24 left the second round
// waiting for the L0 workgroups while the VIEWSP / RGB ones idled for 2 of its 3 ms):  7*54 + 58 + 29 + 17 + 16 + 7 + 7 = 512.
__host__ __device__ constexpr int dw_splits(int inst) {
    switch (inst) {
        case DW_FEAT: return 58;        // + alpha head
        case DW_VIEWSF: return 29;
        case DW_L0: return 17;
        case DW_L5P: return 16;
        case DW_VIEWSP: return 7;
        case DW_RGB: return 7;
        default: return 54;
    }
}
__host__ __device__ constexpr int dw_block_base(int inst) {   // first workgroup id of an instance
    int b = 0;
    for (int i = 0; i < inst; ++i) b += dw_splits(i);
    return b;
}
constexpr int DW_TOTAL_BLOCKS = dw_block_base(DW_COUNT);
struct DwShape { int n, k; };   // output rows (layer outputs) x cols (layer inputs of this instance)
__host__ __device__ constexpr DwShape dw_shape(int inst) {
    switch (inst) {
        case DW_VIEWSF: return {128, 256};
        case DW_L0: return {256, 64};
        case DW_L5P: return {256, 64};
        case DW_VIEWSP: return {128, 32};
        case DW_RGB: return {4, 128};      // up to 3 channels + 1 spare row
        default: return {256, 256};
    }
}
// per split: weight partial [n*k] + bias partial [n] (+ alpha row [256+1] riding on DW_FEAT)
__host__ __device__ constexpr int64_t dw_inst_floats(int inst) {
    return (int64_t)dw_shape(inst).n * dw_shape(inst).k + dw_shape(inst).n + (inst == DW_FEAT ? 257 : 0);
}
__host__ __device__ constexpr int64_t dw_inst_offset(int inst) {
    int64_t o = 0;
    for (int i = 0; i < inst; ++i) o += dw_inst_floats(i) * dw_splits(i);
    return o;
}
constexpr int64_t DW_WS_FLOATS = dw_inst_offset(DW_COUNT);

// Split-f16 dW (mlp_dw_h.hip): HBM-bound, so workgroup counts follow the bytes an instance streams per point.
//   big kernel, one workgroup per CU: a whole 256x256 instance block per workgroup (one accumulator set); point-splits per
//     instance as tabulated below (7 x 28 + 38 + 22 = 256).
//   small kernel: the thin instances (PE / PE(dir) operands, rgb head) move few bytes per chunk and are latency-
//     bound per workgroup; 128 point-splits each = 512 light workgroups, two per CU (128 registers).
// Split counts of the big kernel, balanced by MEASURED time per point and instance (tools/experiments/trace_dw.py stamps every
// workgroup's start and finish).  Round 3 found the launch waiting for its slowest instance: with 30 splits per 256x256 instance
// and 16 for the views block, the views workgroups streamed 41 % more bytes than the others; after 29 / 24 the trace showed the
// FEAT instance (which also carries the alpha head and its d_sigma stream on the VALU) 37 % behind the seven plain instances.
// 28 / 38 / 22 lets all three kinds finish together.  The thin instances keep 128 splits each: they are latency-bound per
// workgroup, byte-proportional counts (176 / 176 / 96 / 64) measured 3 % slower (profiles/r03_dw_balance.log).
// Overridable at compile time for experiments.
#ifndef DWH_LS
#define DWH_LS 28
#endif
#ifndef DWH_FS
#define DWH_FS 38
#endif
#ifndef DWH_VS
#define DWH_VS 22
#endif
#ifndef DWH_T0
#define DWH_T0 128
#endif
#ifndef DWH_T1
#define DWH_T1 128
#endif
#ifndef DWH_T2
#define DWH_T2 128
#endif
__host__ __device__ constexpr int dwh_splits(int inst) {
    switch (inst) {
        case DW_FEAT: return DWH_FS;
        case DW_VIEWSF: return DWH_VS;
        case DW_L0: case DW_L5P: return DWH_T0;
        case DW_VIEWSP: return DWH_T1;
        case DW_RGB: return DWH_T2;
        default: return DWH_LS;
    }
}
constexpr int DWH_BIG_BLOCKS = 7 * DWH_LS + DWH_FS + DWH_VS;
// instance and split of workgroup b of the big kernel (instances DW_L1 .. DW_VIEWSF in DwInst order)
__host__ __device__ constexpr int dwh_big_inst(int b) { return b < 7 * DWH_LS ? b / DWH_LS : b < 7 * DWH_LS + DWH_FS ? DW_FEAT : DW_VIEWSF; }
__host__ __device__ constexpr int dwh_big_split(int b) { return b < 7 * DWH_LS ? b % DWH_LS : b < 7 * DWH_LS + DWH_FS ? b - 7 * DWH_LS : b - 7 * DWH_LS - DWH_FS; }
static_assert(DW_L7 == 6 && DW_FEAT == 7 && DW_VIEWSF == 8, "big-kernel instance order");
constexpr int DWH_SMALL_BLOCKS = 2 * DWH_T0 + DWH_T1 + DWH_T2;
// thin instance and split of workgroup b of the small kernel (instances in DwInst order: L0, L5P, VIEWSP, RGB)
__host__ __device__ constexpr int dwh_thin_inst(int b) { return b < DWH_T0 ? DW_L0 : b < 2 * DWH_T0 ? DW_L5P : b < 2 * DWH_T0 + DWH_T1 ? DW_VIEWSP : DW_RGB; }
__host__ __device__ constexpr int dwh_thin_split(int b) { return b < DWH_T0 ? b : b < 2 * DWH_T0 ? b - DWH_T0 : b < 2 * DWH_T0 + DWH_T1 ? b - 2 * DWH_T0 : b - 2 * DWH_T0 - DWH_T1; }
__host__ __device__ constexpr int64_t dwh_inst_offset(int inst) {
    int64_t o = 0;
    for (int i = 0; i < inst; ++i) o += dw_inst_floats(i) * dwh_splits(i);
    return o;
}
static_assert(dwh_inst_offset(DW_COUNT) <= DW_WS_FLOATS, "split-mode partials fit the f32-mode workspace");
static_assert(DWH_BIG_BLOCKS == 256, "one workgroup per CU");
#ifndef DWH_ALLOW_ANY   /* timing variants with other thin split counts (tools/experiments/build_variant.sh) */
static_assert(DWH_SMALL_BLOCKS == 512, "two light workgroups per CU");
#endif

// BENERF_MLP_SPLIT, round 5 (mlp_dw_s.hip).  The feature layer feeds the views layer WITHOUT a ReLU in between (model/nerf.py:
// 100-106: feature = W_f h7 + b_f; hv = relu(W_v [feature, PE(dir)] + b_v)), so with G = dhv^T h7 (128 x 256, one contraction over
// the points) both weight gradients that involve `feature` follow by two tiny GEMMs in the reduce stage:
//     d feature = dhv W_vf              =>  dW_f  = d feature^T h7 = W_vf^T G,        db_f = W_vf^T (sum dhv)
//     dW_v[:, :256] = dhv^T feature     =   G W_f^T + (sum dhv) b_f^T
// Neither `feature` nor its gradient is saved any more (768 bytes per point less out of the forward, 768 less out of dX, 1 536
// less into dW), the 256 x 256 FEAT instance of the big kernel is gone and the views block reads h7 instead of feature (and
// carries the alpha head, which wants h7 too): 7 plain instances x DWS_LS point-splits + DWS_GS for G = 256 workgroups, balanced
// by measured time (round 4: the views block cost 0.79 of a plain instance per point, the alpha head 0.36: G = 1.15 -> 31 / 39).  The alpha partials keep their place in the FEAT instance's part
// of the workspace (its first split's block doubles as the buffer of the reduced G + sum dhv).
#ifndef DWS_LS
#define DWS_LS 31
#endif
#ifndef DWS_GS
#define DWS_GS 39
#endif
__host__ __device__ constexpr int dws_splits(int inst) {
    switch (inst) {
        case DW_FEAT: case DW_VIEWSF: return DWS_GS;
        case DW_L0: case DW_L5P: return DWH_T0;
        case DW_VIEWSP: return DWH_T1;
        case DW_RGB: return DWH_T2;
        default: return DWS_LS;
    }
}
constexpr int DWS_BIG_BLOCKS = 7 * DWS_LS + DWS_GS;
__host__ __device__ constexpr int dws_big_inst(int b) { return b < 7 * DWS_LS ? b / DWS_LS : DW_VIEWSF; }
__host__ __device__ constexpr int dws_big_split(int b) { return b < 7 * DWS_LS ? b % DWS_LS : b - 7 * DWS_LS; }
__host__ __device__ constexpr int64_t dws_inst_offset(int inst) {
    int64_t o = 0;
    for (int i = 0; i < inst; ++i) o += dw_inst_floats(i) * dws_splits(i);
    return o;
}
static_assert(dws_inst_offset(DW_COUNT) <= DW_WS_FLOATS, "BENERF_MLP_SPLIT partials fit the f32-mode workspace");
static_assert(DWS_BIG_BLOCKS == 256, "one workgroup per CU");
static_assert(128 * 256 + 128 <= 256 * 256 + 256, "reduced G + sum dhv fit in front of the alpha partials of the FEAT block");

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and
// STORE (s_waitcnt vmcnt(0)) - with ~10 KB of activations stored per point that is one HBM write latency per stage.
// None of the MLP kernels passes data between threads through global memory, so LDS ordering is all they need.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// accumulator register r of a 32x32 tile holds row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace mlp
