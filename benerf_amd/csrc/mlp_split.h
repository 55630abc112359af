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

__device__ __forceinline__ void split_store(_Float16* __restrict__ Th, _Float16* __restrict__ Tl, int idx, float v) {
    const _Float16 hi = (_Float16)v;
    Th[idx] = hi;
    Tl[idx] = (_Float16)((v - (float)hi) * LO_SCALE);
}

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// acc1 += hi x hi ; acc2 += hi x lo + lo x hi   over T[:, kcol0 .. kcol0 + KS*16) and packed tile ct0+c
template <int KS, int NCT>
__device__ __forceinline__ void gemm_stage(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, int kcol0,
                                           const float* __restrict__ wp, int ct0, int lane, f32x16 (&acc1)[2][NCT],
                                           f32x16 (&acc2)[2][NCT]) {
    const int row = lane & 31, lh = lane >> 5;
    const int sw = hsw(row);                        // rows row and row + 32 share the swizzle
    const int rbase = row * LD;
    const uint4* bp[NCT];
    uint4 bhn[NCT], bln[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        bp[c] = reinterpret_cast<const uint4*>(wp) + (int64_t)(ct0 + c) * KS * 128 + lane;
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
                bhn[c] = bp[c][(ks + 1) * 128];
                bln[c] = bp[c][(ks + 1) * 128 + 64];
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


// f32 scratch inside the planes' PE columns [256,320): 64 floats per row, floats [0,32) in the hi plane, [32,64)
// in the lo plane (slot 32 + i/4 of the plane, swizzled like everything else).
__device__ __forceinline__ float* fscr(_Float16* Th, _Float16* Tl, int row, int i) {
    _Float16* pl = (i & 32) ? Tl : Th;
    return reinterpret_cast<float*>(pl + row * LD + (((32 + ((i & 31) >> 2)) ^ hsw(row)) << 3)) + (i & 3);
}

}  // namespace mlp
