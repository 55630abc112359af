// Shared pieces of the split-f16 MLP kernels (mlp_fwd_h.hip, mlp_bwd_h.hip, mlp_dw_h.hip): an f32 value x is
// carried as two f16 numbers x = hi + lo * 2^-11 and a product block is three f16 MFMAs (hi*hi, hi*lo, lo*hi)
// with f32 accumulation.  LDS holds two f16 planes Th/Tl[64][LD]; 16-byte slots (8 halfs) are XOR-swizzled:
// element (row, col) lives in slot (col>>3) ^ ((row>>1)&7) of its row.
#pragma once
#include "mlp_common.h"

namespace mlp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

__device__ __forceinline__ int hsw(int row) { return (row >> 1) & 7; }
// half index of element (row, col) inside a plane
__device__ __forceinline__ int hidx(int row, int col) { return row * LD + ((((col >> 3) ^ hsw(row)) << 3) | (col & 7)); }

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// acc1 += hi x hi ; acc2 += hi x lo + lo x hi   over T[:, kcol0 .. kcol0 + KS*16) and packed tile ct0+c.
// Weight fragments stream from L2 PF k-steps ahead (a k-step is only 12 MFMAs = 384 cycles, less than an L2 round
// trip under load); the loop is fully unrolled so the PF+1 register sets rotate at compile time.
template <int KS, int NCT, int PF = 2>
__device__ __forceinline__ void gemm_stage(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, int kcol0,
                                           const float* __restrict__ wp, int ct0, int lane, f32x16 (&acc1)[2][NCT],
                                           f32x16 (&acc2)[2][NCT]) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int row = lane & 31, lh = lane >> 5;
    const int sw = hsw(row);                        // rows row and row + 32 share the swizzle
    const int rbase = row * LD;
    // Weight fragments through a buffer descriptor: base and k-step offset stay in SGPRs, the only VGPR is lane*16
    // (plain pointers made the compiler materialise and spill one 64-bit VGPR address per unrolled k-step).
    const uint64_t wa = reinterpret_cast<uint64_t>(wp);
    const uint64_t wau = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wa >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wa);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(wau), 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    // packed layout (mlp_pack.hip): fragment (tile t, k-step ks, plane) at ((((t>>1)*KS + ks)*2 + (t&1))*2 + plane) KiB:
    // one SGPR offset per k-step, the four fragments of a tile pair at immediate offsets 0 / 1 / 2 / 3 KiB
    const int ct0u = __builtin_amdgcn_readfirstlane(ct0);
    const int tbase = (ct0u >> 1) * KS * 4096 + (ct0u & 1) * 2048;
    auto load_b = [&](int c, int ks, int plane) -> u32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + c * 2048 + plane * 1024, tbase + ks * 4096, 0);
    };
    u32x4 bq[PF + 1][NCT][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < KS) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                bq[p][c][0] = load_b(c, p, 0);
                bq[p][c][1] = load_b(c, p, 1);
            }
        }
    const int slot0 = kcol0 >> 3;                    // multiple of 8: the swizzle only permutes slots inside 8-slot groups
    // slot (slot0 + 2ks + lh) ^ sw = group base (compile-time) + ((2(ks&3) + lh) ^ sw): four lane-dependent bases
    int abase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) abase[j] = rbase + (((2 * j + lh) ^ sw) << 3);
    half8 an[2][2];                                  // A fragments (hi, lo) x (row tile) of the NEXT k-step
    auto load_a = [&](int ks) {
        const int off = abase[ks & 3] + ((slot0 + ((2 * ks) & ~7)) << 3);
        an[0][0] = *reinterpret_cast<const half8*>(Th + off);
        an[0][1] = *reinterpret_cast<const half8*>(Th + off + 32 * LD);
        an[1][0] = *reinterpret_cast<const half8*>(Tl + off);
        an[1][1] = *reinterpret_cast<const half8*>(Tl + off + 32 * LD);
    };
    load_a(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        half8 ah[2] = {an[0][0], an[0][1]}, al[2] = {an[1][0], an[1][1]};
        if (ks + PF < KS) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                bq[(ks + PF) % (PF + 1)][c][0] = load_b(c, ks + PF, 0);
                bq[(ks + PF) % (PF + 1)][c][1] = load_b(c, ks + PF, 1);
            }
        }
        if (ks + 1 < KS) load_a(ks + 1);
        half8 bh[NCT], bl[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            bh[c] = __builtin_bit_cast(half8, bq[ks % (PF + 1)][c][0]);
            bl[c] = __builtin_bit_cast(half8, bq[ks % (PF + 1)][c][1]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc1[r][c] = mfma16(ah[r], bh[c], acc1[r][c]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc2[r][c] = mfma16(ah[r], bl[c], acc2[r][c]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc2[r][c] = mfma16(al[r], bh[c], acc2[r][c]);
        __builtin_amdgcn_sched_barrier(0);          // one k-step per scheduling region: keeps the prefetch distances as written
    }
}

// Rolled variant (weights one k-step ahead, plain pointers): same result as gemm_stage; the dX kernel keeps it
// because the unrolled, descriptor-based version drives that kernel into heavy register spilling.
template <int KS, int NCT>
__device__ __forceinline__ void gemm_stage_rolled(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, int kcol0,
                                           const float* __restrict__ wp, int ct0, int lane, f32x16 (&acc1)[2][NCT],
                                           f32x16 (&acc2)[2][NCT]) {
    const int row = lane & 31, lh = lane >> 5;
    const int sw = hsw(row);                        // rows row and row + 32 share the swizzle
    const int rbase = row * LD;
    const uint4* bp[NCT];
    uint4 bhn[NCT], bln[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        bp[c] = reinterpret_cast<const uint4*>(wp) + ((int64_t)((ct0 + c) >> 1) * KS * 4 + ((ct0 + c) & 1) * 2) * 64 + lane;
        bhn[c] = bp[c][0];
        bln[c] = bp[c][64];
    }
    const int slot0 = kcol0 >> 3;
#pragma unroll 2
    for (int ks = 0; ks < KS; ++ks) {
        half8 bh[NCT], bl[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            bh[c] = __builtin_bit_cast(half8, bhn[c]);
            bl[c] = __builtin_bit_cast(half8, bln[c]);
        }
        if (ks + 1 < KS) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                bhn[c] = bp[c][(ks + 1) * 256];
                bln[c] = bp[c][(ks + 1) * 256 + 64];
            }
        }
        const int off = rbase + (((slot0 + ks * 2 + lh) ^ sw) << 3);
        half8 ah[2], al[2];
        ah[0] = *reinterpret_cast<const half8*>(Th + off);
        ah[1] = *reinterpret_cast<const half8*>(Th + off + 32 * LD);
        al[0] = *reinterpret_cast<const half8*>(Tl + off);
        al[1] = *reinterpret_cast<const half8*>(Tl + off + 32 * LD);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc1[r][c] = mfma16(ah[r], bh[c], acc1[r][c]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc2[r][c] = mfma16(ah[r], bl[c], acc2[r][c]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc2[r][c] = mfma16(al[r], bh[c], acc2[r][c]);
    }
}

template <int NCT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NCT]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
}

// 8 consecutive activations (one slot) of this thread's point row as f32
__device__ __forceinline__ void load8(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, int off, float (&v)[8]) {
    const half8 h = *reinterpret_cast<const half8*>(Th + off);
    const half8 l = *reinterpret_cast<const half8*>(Tl + off);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j] + (float)l[j] * LO_INV;
}


// ---- saved activations / activation gradients in split mode ("ST" arrays) -------------------------------------
// An ST array of width W over Mp = 64 * n_tiles points stores the split value of (m, w) as two halfs at
//   half index  ((m >> 3) * W + w) * 16 + ((m >> 2) & 1) * 8 + plane * 4 + (m & 3)        plane 0 = hi, 1 = lo * 2^11
// i.e. blocks of 8 points, feature-major; per (block, feature) 32 bytes = [points 0-3: hi x4, lo x4][points 4-7:
// hi x4, lo x4].  A lane of the forward / dX epilogue holds 4 consecutive points of one feature, so it writes its
// hi and lo quads as ONE 16-byte store and a wave covers 1 KiB contiguous.  The dW kernel copies chunks into LDS
// 16 bytes at a time and splits each unit into its two 8-byte quads there, which yields the MFMA fragment order
// (8 points of a feature contiguous per plane) without a transposition pass.  Same bytes per element (4) as the f32
// arrays.  Rows m >= M of the last tile are written too (duplicates of the last point for activations, exact
// zeros for gradients), so no store is masked and no dW chunk is ragged.
__host__ __device__ inline int64_t st_half_index(int64_t m, int W, int w, int plane) {
    return ((m >> 3) * W + w) * 16 + ((m >> 2) & 1) * 8 + plane * 4 + (m & 3);
}
__host__ __device__ inline int64_t m_pad(int64_t M) { return n_tiles(M) * TM; }
// float (4-byte) offsets inside the split-mode activation buffer
constexpr int SACT_MASK_LAYERS = 9;                                       // h0..h7 + hv
__host__ __device__ inline int64_t sact_pe32(int64_t Mp) { (void)Mp; return 0; }                 // f32 [Mp][64]  (dX: sin/cos)
__host__ __device__ inline int64_t sact_ped32(int64_t Mp) { return Mp * ACT_PE_W; }               // f32 [Mp][32]
__host__ __device__ inline int64_t sact_h(int64_t Mp, int l) { return sact_ped32(Mp) + Mp * ACT_PED_W + (int64_t)l * Mp * 256; }
__host__ __device__ inline int64_t sact_feat(int64_t Mp) { return sact_h(Mp, 8); }                // ST W = 256
__host__ __device__ inline int64_t sact_hv(int64_t Mp) { return sact_feat(Mp) + Mp * 256; }       // ST W = 128
__host__ __device__ inline int64_t sact_mask(int64_t Mp) { return sact_hv(Mp) + Mp * ACT_HV_W; }  // uint64 [9][tiles][256]
__host__ __device__ inline int64_t sact_total_floats(int64_t M) {
    return sact_mask(m_pad(M)) + (int64_t)SACT_MASK_LAYERS * n_tiles(M) * NTHREADS * 2 + 16 + n_tiles(M) * 4;   // + absmax slots + per-wave table
}
// activation gradients: ST arrays holding dY * s_g (s_g = power-of-two scale of this backward call from max|d_raw|),
// then 16 absmax slots: [0] max|d_raw|, [1..10] max|dY * s_g| of every array (see 'dW operand formats')
__host__ __device__ inline int64_t sdact_h(int64_t Mp, int l) { return (int64_t)l * Mp * 256; }
__host__ __device__ inline int64_t sdact_feat(int64_t Mp) { return 8 * Mp * 256; }
__host__ __device__ inline int64_t sdact_hv(int64_t Mp) { return 9 * Mp * 256; }
__host__ __device__ inline int64_t sdact_scale(int64_t Mp) { return 9 * Mp * 256 + Mp * ACT_HV_W; }
__host__ __device__ inline int64_t sdact_absmax_table(int64_t Mp) { return sdact_scale(Mp) + 16; }                      // [tiles][4 waves][16 stages]
__host__ __device__ inline int64_t sdact_total_floats(int64_t M) { return sdact_scale(m_pad(M)) + 16 + n_tiles(M) * 64; }

// power-of-two scale s = 2^(-4 - exponent(mx)) that brings values of magnitude <= mx to <= 2^-3, and its inverse
__device__ __forceinline__ void pow2_scale(float mx, float& s, float& inv_s) {
    int be = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    be = be < 4 ? 4 : (be > 246 ? 246 : be);
    s = __uint_as_float((uint32_t)(250 - be) << 23);
    inv_s = __uint_as_float((uint32_t)(be + 4) << 23);
}
// The dX chain works at 2^-4 (head room for growth through the layers); the dY arrays it STORES sit 2^8 higher
// (max|d_raw| in [2^4, 2^5)): the tile -> call factor is then a power of two <= 2^8 applied to the f16 halfs, which
// stays exact for every element down to 2^-18 of the call's largest gradient.
constexpr float DY_STORE_BOOST = 256.f;
// ---- dW operand formats ---------------------------------------------------------------------------------------
// The dW kernels add the three products of a block (hi*hi, hi*lo, lo*hi) into ONE accumulator set - that is what lets
// a workgroup hold a whole 256x256 output block and read every operand byte once.  It needs the lo parts UNSCALED
// (x = hi + lo), i.e. operands scaled so that lo stays inside f16's range for every element that matters.  The
// stored ST arrays keep the robust (hi, lo * 2^11) form; the forward / dX kernels also publish max|value| of every
// value they save (one running maximum per wave -> table -> absmax_reduce_kernel -> absmax slot), and the dW kernel
// rescales each unit while staging it into LDS:  hi' = hi * 2^k,  lo' = lo * 2^(k-11)  with k = 14 - exponent(absmax),
// one k for all activation arrays of the forward launch and one for all gradient arrays of the backward launch
// (they are all of one order of magnitude), so the largest stored element lands in [2^14, 2^15).  Both are exact powers of two (two packed-f16 multiplies each,
// every factor inside f16's range); elements down to 2^-17 of the maximum keep a normal lo', smaller ones an
// absolute floor of 2^-38 of the maximum.  The reduce kernel multiplies by 2^-(kY + kX) (and 1/s_g for dY).
// absmax slots: acts buffer [AX_ALL] = max |saved activation|; dacts buffer [AY_DRAW] = max|d_raw|, [AY_ALL] = max
// |stored gradient| (at the stored scale dY * s_g)
enum { AX_ALL = 0, AX_COUNT = 16 };
enum { AY_DRAW = 0, AY_ALL = 1, AY_COUNT = 16 };
__host__ __device__ inline int64_t sact_absmax(int64_t M) {
    return sact_mask(m_pad(M)) + (int64_t)SACT_MASK_LAYERS * n_tiles(M) * NTHREADS * 2;
}
__host__ __device__ inline int64_t sact_absmax_table(int64_t M) { return sact_absmax(M) + AX_COUNT; }   // [tiles][4 waves]
// k = 14 - exponent(mx), clamped so that 2^k and 2^(k-11) are products of two f16-representable powers of two
__device__ __forceinline__ int rescale_exp(float mx) {
    const int be = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    if (be == 0) return 0;
    int k = 14 - (be - 127);
    return k < -14 ? -14 : (k > 30 ? 30 : k);
}
__device__ __forceinline__ float exp2i(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }   // 2^k, |k| <= 126
// wave-wide max of non-negative values on the VALU (DPP row shifts + broadcasts, no LDS traffic); result in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_step(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return fmaxf(v, __builtin_bit_cast(float, o));
}
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = dpp_max_step<0x111, 0xf>(v);   // row_shr:1
    v = dpp_max_step<0x112, 0xf>(v);   // row_shr:2
    v = dpp_max_step<0x114, 0xf>(v);   // row_shr:4
    v = dpp_max_step<0x118, 0xf>(v);   // row_shr:8
    v = dpp_max_step<0x142, 0xa>(v);   // row_bcast:15 -> rows 1, 3
    v = dpp_max_step<0x143, 0xc>(v);   // row_bcast:31 -> rows 2, 3
    return v;
}
// one plain store per wave into a [tile][wave] table (no atomics: thousands of workgroups hammering one address
// serialise in L2; no read-compare: it would make the wave wait for all its outstanding stores).
// absmax_reduce_kernel folds the table into the slot after the kernel.
__device__ __forceinline__ void publish_absmax(float mx, float* wave_entry) {
    mx = wave_max_nonneg(mx);
    if ((threadIdx.x & 63) == 63) *wave_entry = mx;
}
__global__ void absmax_reduce_kernel(const float* __restrict__ table, int64_t n, float* __restrict__ slot);
int absmax_reduce_launch(const float* table, int64_t n, float* slot, hipStream_t stream);

// 4 consecutive points (one accumulator quad) of one feature = one 16-byte piece {hi x4, lo x4} of an ST array
struct Quad16 { _Float16 v[4]; };
struct Quad16x2 { Quad16 hi, lo; };

// f32 scratch inside the planes' PE columns [256,320): 64 floats per row, floats [0,32) in the hi plane, [32,64)
// in the lo plane (slot 32 + i/4 of the plane, swizzled like everything else).
__device__ __forceinline__ float* fscr(_Float16* Th, _Float16* Tl, int row, int i) {
    _Float16* pl = (i & 32) ? Tl : Th;
    return reinterpret_cast<float*>(pl + row * LD + (((32 + ((i & 31) >> 2)) ^ hsw(row)) << 3)) + (i & 3);
}

}  // namespace mlp
