// Shared pieces of the split-f16 MLP kernels (mlp_fwd_h.hip; mlp_bwd_s.hip / mlp_dw_s.hip: BENERF_MLP_SPLIT; mlp_bwd_h.hip /
// mlp_dw_h.hip: BENERF_MLP_SPLIT_F16BWD).  An f32 value x is carried as two f16 numbers x = hi + lo * 2^-11 and a product block
// is THREE f16 MFMAs (hi*hi, hi*lo, lo*hi) with f32 accumulation - in the forward pass and, in BENERF_MLP_SPLIT (the default),
// in both backward GEMMs too (dX chain: gradient and transposed weight as f16 pairs; dW: 19-bit saved operands = f16 hi +
// 8-bit residual code).  Only the opt-in BENERF_MLP_SPLIT_F16BWD backward takes single-f16 operands (dW 1 MFMA, dX 2).
// LDS holds two f16 planes Th/Tl[128][LD] (forward: one workgroup of 8 waves per CU).  16-byte LDS slots (8 halfs) are
// XOR-swizzled: element (row, col) lives in slot (col>>3) ^ ((row>>1)&7) of its row.
#pragma once
#include "mlp_common.h"

namespace mlp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

#if !defined(BENERF_HSW_V2) || defined(FWD_OLD_SWIZZLE)
// the translation units that do not ask for the round-5 swizzle (BENERF_MLP_SPLIT_F16BWD's dX kernel, mlp_bwd_h.hip, is written against this one)
__device__ __forceinline__ int hsw(int row) { return (row >> 1) & 7; }
#else
// Round 5.  (row >> 1) & 7 served the K-loop's 16-byte reads (16 rows per LDS cycle over 64 banks: a row's 640 bytes put odd rows
// half a bank sweep behind even ones, so the slot only has to separate rows of equal parity) and left the epilogue's 16-byte
// WRITES two-way conflicted (8 rows per cycle over 32 banks: eight consecutive rows shared four slots; 49-70 M
// SQ_LDS_BANK_CONFLICT cycles per 522 k-point launch).  (row & 7) ^ ((row >> 3) & 1) is a bijection on eight consecutive rows
// (writes) AND on the eight rows of equal parity among sixteen (reads): rows 0, 2, .., 14 -> 0, 2, 4, 6, 1, 3, 5, 7.  Rows r and
// r + 32 / r + 64 still share their swizzle (gemm_stage, the VIEWS half tiles).
__device__ __forceinline__ int hsw(int row) { return (row & 7) ^ ((row >> 3) & 1); }
#endif
// half index of element (row, col) inside a plane
__device__ __forceinline__ int hidx(int row, int col) { return row * LD + ((((col >> 3) ^ hsw(row)) << 3) | (col & 7)); }

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#ifndef FWD_PF
#define FWD_PF 2
#endif
// acc1 += hi x hi ; acc2 += hi x lo + lo x hi   over T[:, kcol0 .. kcol0 + KS*16) and packed tile ct0+c.
// Weight fragments stream from L2 PF k-steps ahead (a k-step is only 12 MFMAs = 384 cycles, less than an L2 round
// trip under load); the loop is fully unrolled so the PF+1 register sets rotate at compile time.
// SWAP: the transposed product (weights as the A operand): accumulator lane = point, elements = 16 features in quads of 4.
// NR: row (point) tiles of 32 per wave.
// after_head(): caller's work that issues vector-memory STORES (the previous stage's activation save), run right behind the
// requests for the first PF k-steps' fragments - vector-memory operations retire in order through one counter, so
// fragments requested behind those stores would not be usable before every store is acknowledged.
struct NoAfterHead { __device__ __forceinline__ void operator()() const {} };
// per_kstep(ks): caller's work spread over the k-steps (placed in the k-step's scheduling region, so its VALU / LDS / store
// instructions issue in the gaps between this wave's MFMAs: an MFMA occupies the issue port for one pass of its eight)
struct NoPerKstep { __device__ __forceinline__ void operator()(int) const {} };
template <int KS, int NCT, int PF = FWD_PF, bool SWAP = false, int NR = 2, class AH = NoAfterHead, class PK = NoPerKstep>
__device__ __forceinline__ void gemm_stage(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, int kcol0,
                                           const float* __restrict__ wp, int ct0, int lane, f32x16 (&acc1)[NR][NCT],
                                           f32x16 (&acc2)[NR][NCT], AH after_head = AH(), PK per_kstep = PK()) {
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
            __builtin_amdgcn_sched_barrier(0);      // in k-step order: the scheduler otherwise requests k-step 0 last
        }
    after_head();
    __builtin_amdgcn_sched_barrier(0);
    const int slot0 = kcol0 >> 3;                    // multiple of 8: the swizzle only permutes slots inside 8-slot groups
    // slot (slot0 + 2ks + lh) ^ sw = group base (compile-time) + ((2(ks&3) + lh) ^ sw): four lane-dependent bases
    int abase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) abase[j] = rbase + (((2 * j + lh) ^ sw) << 3);
    // ONE set of activation fragments (hi / lo x row tile): the MFMAs of a fragment are consecutive and its successor is
    // requested right behind them; the other fragments' MFMAs (>= 8 x 32 cycles) cover the LDS round trip.  The
    // registers this saves go to a deeper weight-fragment ring (PF).
    half8 ah[NR], al[NR];
    auto a_off = [&](int ks) { return abase[ks & 3] + ((slot0 + ((2 * ks) & ~7)) << 3); };
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        ah[r] = *reinterpret_cast<const half8*>(Th + a_off(0) + r * 32 * LD);
        al[r] = *reinterpret_cast<const half8*>(Tl + a_off(0) + r * 32 * LD);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + PF < KS) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                bq[(ks + PF) % (PF + 1)][c][0] = load_b(c, ks + PF, 0);
                bq[(ks + PF) % (PF + 1)][c][1] = load_b(c, ks + PF, 1);
            }
        }
        per_kstep(ks);
        half8 bh[NCT], bl[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            bh[c] = __builtin_bit_cast(half8, bq[ks % (PF + 1)][c][0]);
            bl[c] = __builtin_bit_cast(half8, bq[ks % (PF + 1)][c][1]);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc1[r][c] = SWAP ? mfma16(bh[c], ah[r], acc1[r][c]) : mfma16(ah[r], bh[c], acc1[r][c]);
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc2[r][c] = SWAP ? mfma16(bl[c], ah[r], acc2[r][c]) : mfma16(ah[r], bl[c], acc2[r][c]);
            if (ks + 1 < KS) ah[r] = *reinterpret_cast<const half8*>(Th + a_off(ks + 1) + r * 32 * LD);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc2[r][c] = SWAP ? mfma16(bh[c], al[r], acc2[r][c]) : mfma16(al[r], bh[c], acc2[r][c]);
            if (ks + 1 < KS) al[r] = *reinterpret_cast<const half8*>(Tl + a_off(ks + 1) + r * 32 * LD);
        }
        __builtin_amdgcn_sched_barrier(0);          // one k-step per scheduling region: keeps the prefetch distances as written
    }
}

template <int NR, int NCT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NR][NCT]) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
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


// ---- saved activations / activation gradients in split mode ("SH" arrays) -------------------------------------
// What the forward / dX kernels save for the dW kernels: the f16 hi half of every value in an "SH" array (described here) and,
// in BENERF_MLP_SPLIT, an 8-bit residual code in the array's lo8 twin (further down: 3 bytes, 19 bits per saved value; three
// MFMAs per dW block).  BENERF_MLP_SPLIT_F16BWD saves and multiplies the hi halves only (one MFMA per block).
// An SH array of width W over Mp points stores element (m, w) at
//   half index  ((m >> 3) * W + w) * 8 + (m & 7)
// i.e. blocks of 8 points, feature-major, one 16-byte unit per (block, feature) = the MFMA A/B fragment of a
// contraction over points (dW = dY^T X), so the dW kernel copies chunks straight into LDS.  A lane of the forward /
// dX epilogue holds 4 consecutive points of one feature = one 8-byte store; a wave covers 512 B contiguous.
// Mp = M rounded up to 128 points (rows m >= M: copies of the last point for activations, exact zeros for gradients),
// so no store is masked and no dW chunk is ragged.  2 bytes per element (+ 1 for the code in BENERF_MLP_SPLIT).
constexpr int SM_PAD = 128;
__host__ __device__ inline int64_t sh_half_index(int64_t m, int W, int w) { return ((m >> 3) * W + w) * 8 + (m & 7); }
__host__ __device__ inline int64_t m_pad(int64_t M) { return (M + SM_PAD - 1) / SM_PAD * SM_PAD; }
__host__ __device__ inline int64_t sn_tiles(int64_t M) { return m_pad(M) / TM; }                 // 64-point tiles, even
// float (4-byte) offsets inside the split-mode activation buffer
constexpr int SACT_MASK_LAYERS = 9;                                       // h0..h7 + hv
__host__ __device__ inline int64_t sact_pe32(int64_t Mp) { (void)Mp; return 0; }                 // f32 [Mp][64]  (dX: sin/cos)
__host__ __device__ inline int64_t sact_ped32(int64_t Mp) { return Mp * ACT_PE_W; }               // f32 [Mp][32]
__host__ __device__ inline int64_t sact_h(int64_t Mp, int l) { return sact_ped32(Mp) + Mp * ACT_PED_W + (int64_t)l * Mp * 128; }   // SH W = 256
__host__ __device__ inline int64_t sact_feat(int64_t Mp) { return sact_h(Mp, 8); }                // SH W = 256
__host__ __device__ inline int64_t sact_hv(int64_t Mp) { return sact_feat(Mp) + Mp * 128; }       // SH W = 128
__host__ __device__ inline int64_t sact_mask(int64_t Mp) { return sact_hv(Mp) + Mp * (ACT_HV_W / 2); }   // uint64 [9][tiles][256]
__host__ __device__ inline int64_t sact_info(int64_t Mp) { return sact_mask(Mp) + (int64_t)SACT_MASK_LAYERS * (Mp / TM) * NTHREADS * 2; }
__host__ __device__ inline int64_t sact_total_floats(int64_t M) { return sact_info(m_pad(M)) + 16; }
// info words (uint32) at sact_info: [SI_TAG] arithmetic mode that wrote the buffer (the backward launches check it)
enum { SI_TAG = 0, SI_COUNT = 16 };
constexpr uint32_t SACT_TAG_SPLIT = 0x53504c54u;   // 'SPLT': f16 halves only (BENERF_MLP_SPLIT_F16BWD)
constexpr uint32_t SACT_TAG_SPLIT22 = 0x53504c32u; // 'SPL2': hi + lo (BENERF_MLP_SPLIT, the 22-bit backward)
// BENERF_MLP_SPLIT (the fp32-equivalent backward, mlp_bwd_s.hip / mlp_dw_s.hip): every SH array above has a twin of 8-BIT RESIDUAL
// CODES ("lo8": one byte per value, same element order: an 8-byte unit per (block, feature)) in a second region behind the info
// words.  With hi = rn16(x), E = max(exponent(hi), -6):
//     code = clamp(rn((x - hi) * 2^(18 - E)) + 128, 0, 255)          x ~ hi + (code - 128) * 2^(E - 18)
// i.e. the residual in units of ulp(hi) / 256: x keeps 19 significant bits (absolute floor 2^-25, the f16 subnormal grid the
// decoded low half lives on).  3 bytes per value instead of the 4 of an f16 pair: the dW kernels are HBM-bound on exactly these
// bytes, the forward / dX kernels pay for every byte they store.  tools/experiments/backward_format_study.py: 19-bit stored
// operands ("h8") leave the weight gradients 2.5e-6 of the largest entry from float64 on identical masks (f16 pairs 1.2e-6,
// exact f32 0.8e-6, 15 bits 3.7e-5, one f16 6e-4) - inside float32's own error band; the dX CHAIN keeps f16 pairs (LDS only).
// Element i of a hi array (half index) <-> byte i of its lo8 twin.
__host__ __device__ inline int64_t sact_lo8_base(int64_t Mp) { return sact_info(Mp) + SI_COUNT; }      // float offset of the lo8 region
__host__ __device__ inline int64_t sact22_total_floats(int64_t M) {
    const int64_t Mp = m_pad(M);
    return sact_lo8_base(Mp) + (sact_mask(Mp) - sact_h(Mp, 0)) / 2;
}
// lo8 twin (byte pointer) of the hi array at float offset `hi_off` of the activation buffer
__host__ __device__ inline const uint8_t* sact_lo8(const float* acts, int64_t Mp, int64_t hi_off) {
    return reinterpret_cast<const uint8_t*>(acts + sact_lo8_base(Mp)) + 2 * (hi_off - sact_h(Mp, 0));
}

// BENERF_MLP_SPLIT also re-uses the two f32 regions at the head of the buffer (sact_pe32 / sact_ped32: f32 rows in the
// BENERF_MLP_SPLIT_F16BWD layout).  The thin dW instances (X = positional encoding) want X as MFMA fragments like every other
// operand, and the dX kernel recomputes sin / cos from the point (12 bytes) instead of reading them back (256 bytes), so:
//   pe region  (256 B / point): SH array W = 64 of PE(pts) [128 B] | its lo8 twin [64 B] | f32 [Mp][8]: pts xyz, 0, viewdir xyz, 0 [32 B]
//   ped region (128 B / point): SH array W = 32 of PE(dir) [64 B]  | its lo8 twin [32 B]
// The encodings are saved AS THE FORWARD PLANES HOLD THEM (BARF column weights applied: the dW reduce must not apply them again).
__host__ __device__ inline int64_t sact22_pe_hi(int64_t Mp) { return sact_pe32(Mp); }
__host__ __device__ inline int64_t sact22_pe_lo8(int64_t Mp) { return sact_pe32(Mp) + Mp * 32; }
__host__ __device__ inline int64_t sact22_pts(int64_t Mp) { return sact_pe32(Mp) + Mp * 48; }
__host__ __device__ inline int64_t sact22_ped_hi(int64_t Mp) { return sact_ped32(Mp); }
__host__ __device__ inline int64_t sact22_ped_lo8(int64_t Mp) { return sact_ped32(Mp) + Mp * 16; }
static_assert(48 + 8 <= ACT_PE_W && 16 + 8 <= ACT_PED_W, "the split22 encodings fit the f32 regions");

// ---- BENERF_MLP_SPLIT, round 5: the 256-wide activation arrays h0..h7 and feature in "SP" (slot-point) layout ------------------
// The forward's transposed product leaves a lane with ONE point and 8 consecutive features per 16-byte plane slot, the dW
// contraction wants 8 consecutive POINTS per feature.  Rounds 3-4 transposed in the forward (transpose reads of the finished
// LDS planes: ~65 VALU + 4 LDS reads per 16 saved values); now the forward stores what its epilogue registers hold and the dW
// kernel, which stages every chunk through LDS anyway, transposes with the reads it does there (ds_read_b64_tr_b16).  Per
// 16-point chunk c (= m >> 4: the dW kernel's k-step) and feature slot s (8 features): 16 consecutive 16-byte units, one per
// point - a chunk of the array is ONE contiguous 8-KiB run for the dW kernel's loads (with whole 128-point tiles per slot the
// chunk was 32 runs of 256 bytes and the in-step dW launch 3 % slower), a wave's store four 256-byte runs:
//   hi:     unit ((m >> 4) * W/8 + s) * 16 + (m & 15)  holds features 8 s .. 8 s + 7 of point m   (f16)
//   codes:  the 8-byte unit of the same index in the lo8 twin
// Same regions, same sizes as the SH arrays they replace (hv, PE, PE(dir) and every gradient array stay SH).
__host__ __device__ inline int64_t sp_half_index(int64_t m, int W, int w) {
    return (((m >> 4) * (W >> 3) + (w >> 3)) * 16 + (m & 15)) * 8 + (w & 7);
}
// ReLU sign bits of h0..h7 in BENERF_MLP_SPLIT, round 5: straight out of v_cmp_gt_f32 on the forward's accumulators (one VALU
// instruction per accumulator register -> a 64-lane mask in an SGPR pair -> s_store_dwordx2; tools/hwprobe/sstore_mask.hip).
// Accumulator element e of (row tile r, column tile ct) in the transposed product: lanes 0-31 = points 32 r + lane of feature
// 32 ct + 8 (e >> 2) + (e & 3), lanes 32-63 = the same points of feature + 4.  uint32 word index inside layer l (same region,
// same stride as before: sact_mask + l * (Mp / 64) * 512 words):   ((T * 4 + r) * 8 + ct) * 32 + 2 e + h,   bit p = point
// 32 r + p,   h = lane >> 5.  hv (mask layer 8) keeps the uint64 [tile64][256] format of mlp_common.h.
__host__ __device__ inline int sp_mask_word(int f32) { return 8 * (f32 >> 3) + 2 * (f32 & 3) + ((f32 >> 2) & 1); }   // feature 0..31 of its column tile

// ---- lo8 codec on packed f16 pairs (one 32-bit register = two values) ----------------------------------------------------
typedef unsigned short h8_ushort2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8_half2 __attribute__((ext_vector_type(2)));
// per half: max(biased exponent of hi, 9) << 10   (E = max(exponent, -6))
__device__ __forceinline__ uint32_t h8_exp(uint32_t hi_pair) {
    const h8_ushort2 lim = {9 << 10, 9 << 10};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h8_ushort2, hi_pair & 0x7C007C00u), lim));
}
// codes of one pair as the low bytes of the two halves of the result.  lo_pair holds (x - hi) * 2^S as f16 (the forward
// kernel's lo plane: S = 11; an unscaled residual pre-multiplied by 4096: S = 12).  One packed FMA does scale, + 128 and the
// rounding: 1024 + 128 + r lands where f16's spacing is 1 (round-to-nearest-even), the code is the low byte of the result.
template <int S>
__device__ __forceinline__ uint32_t h8_code_pair(uint32_t hi_pair, uint32_t lo_pair) {
    const uint32_t sc = (uint32_t)(((48 - S) << 10) * 0x10001u) - h8_exp(hi_pair);                     // 2^(18 - S - E) per half
    const h8_half2 c = __builtin_elementwise_min(
        __builtin_bit_cast(h8_half2, lo_pair) * __builtin_bit_cast(h8_half2, sc) + h8_half2{(_Float16)1152.f, (_Float16)1152.f},
        h8_half2{(_Float16)1279.f, (_Float16)1279.f});
    return __builtin_bit_cast(uint32_t, c);
}
// 8 codes of a unit (hi: 8 halfs as 4 pairs, lo likewise) -> 8 bytes
template <int S>
__device__ __forceinline__ uint2 h8_encode_unit(const uint32_t (&hi)[4], const uint32_t (&lo)[4]) {
    const uint32_t c0 = h8_code_pair<S>(hi[0], lo[0]), c1 = h8_code_pair<S>(hi[1], lo[1]);
    const uint32_t c2 = h8_code_pair<S>(hi[2], lo[2]), c3 = h8_code_pair<S>(hi[3], lo[3]);
    return uint2{__builtin_amdgcn_perm(c1, c0, 0x06040200u), __builtin_amdgcn_perm(c3, c2, 0x06040200u)};
}
// low halves (unscaled f16: lo = (code - 128) * 2^(E - 18)) of one unit from its hi halves and its 8 codes
__device__ __forceinline__ void h8_decode_unit(const uint32_t (&hi)[4], const uint2 code, uint32_t (&lo)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = j < 2 ? code.x : code.y;
        // [0x64, c(2j+1), 0x64, c(2j)]: the f16 numbers 1024 + code
        const uint32_t v = __builtin_amdgcn_perm(0x64646464u, w, (j & 1) ? 0x04030402u : 0x04010400u);
        const h8_half2 t = __builtin_bit_cast(h8_half2, v) * h8_half2{(_Float16)0x1p-10f, (_Float16)0x1p-10f} -
                           h8_half2{(_Float16)1.125f, (_Float16)1.125f};                                  // (code - 128) * 2^-10, exact
        const uint32_t sc = h8_exp(hi[j]) - (uint32_t)((8 << 10) * 0x10001u);                               // 2^(E - 8) per half
        lo[j] = __builtin_bit_cast(uint32_t, t * __builtin_bit_cast(h8_half2, sc));
    }
}
// one value (heads on the VALU, tests of the codec)
__device__ __forceinline__ float h8_decode_one(_Float16 hi, uint32_t code) {
    int e5 = (int)((__builtin_bit_cast(unsigned short, hi) >> 10) & 31u);
    e5 = e5 < 9 ? 9 : e5;
    return (float)((int)code - 128) * __uint_as_float((uint32_t)(127 + (e5 - 15) - 18) << 23);
}
// activation gradients: SH arrays holding dY * s_s (s_s = power-of-two scale of this backward call from max|d_raw|,
// pow2_scale6), then 16 info words: [SD_DRAW] max|d_raw| (float bits, grad_absmax_kernel)
__host__ __device__ inline int64_t sdact_h(int64_t Mp, int l) { return (int64_t)l * Mp * 128; }
__host__ __device__ inline int64_t sdact_feat(int64_t Mp) { return 8 * Mp * 128; }
__host__ __device__ inline int64_t sdact_hv(int64_t Mp) { return 9 * Mp * 128; }
__host__ __device__ inline int64_t sdact_info(int64_t Mp) { return 9 * Mp * 128 + Mp * (ACT_HV_W / 2); }
__host__ __device__ inline int64_t sdact_total_floats(int64_t M) { return sdact_info(m_pad(M)) + 16; }
enum { SD_DRAW = 0, SD_COUNT = 16 };
// fp32-equivalent backward: the lo8 twins of the gradient arrays (same scale s_s; mlp_split.h above) behind the info words
__host__ __device__ inline int64_t sdact_lo8_base(int64_t Mp) { return sdact_info(Mp) + SD_COUNT; }
__host__ __device__ inline int64_t sdact22_total_floats(int64_t M) { const int64_t Mp = m_pad(M); return sdact_lo8_base(Mp) + sdact_info(Mp) / 2; }
__host__ __device__ inline const uint8_t* sdact_lo8(const float* dacts, int64_t Mp, int64_t hi_off) {
    return reinterpret_cast<const uint8_t*>(dacts + sdact_lo8_base(Mp)) + 2 * hi_off;
}

// Gradients are far outside f16's range (d_raw ~ 1/n_rays) but the backward chain is LINEAR in d_raw: power-of-two
// scale s = 2^(6 - exponent(mx)) brings values of magnitude <= mx to < 2^7 (2^9 of head room below f16's maximum
// for growth through the layers, full 11-bit precision down to 2^-20 of mx, absolute floor 2^-31 of mx).  The dX
// kernel scales each 128-point tile by its own maximum; the dY arrays it stores for dW carry ONE scale per call.
__device__ __forceinline__ void pow2_scale6(float mx, float& s, float& inv_s) {
    int be = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    be = be < 8 ? 8 : (be > 250 ? 250 : be);
    s = __uint_as_float((uint32_t)(260 - be) << 23);
    inv_s = __uint_as_float((uint32_t)(be - 6) << 23);
}
// Range guard of the split forward pass: an activation of magnitude >= 65520 rounds to inf in its f16 hi half.  Every
// forward launch folds max|activation| into a caller-owned device word (f32 bit pattern, atomicMax; non-negative
// floats order like their bit patterns) - see benerf_mlp_status in include/benerf_hip.h.
constexpr float F16_RANGE_LIMIT = 65504.f;
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

// sin and cos of v = x * 2^f for the positional encoding (model/embedder.py:13-28): three-term Cody-Waite reduction
// by pi/2 with FMAs (exact product, so |v| up to ~1e5 reduces to < 1 ulp of the reduced argument) + the cephes
// minimax polynomials on [-pi/4, pi/4]; absolute error <= 1.2e-7 (ocml's sincosf takes ~4x the instructions).
// Arguments beyond 1e5 (never reached by NDC / scene-scale coordinates times 2^9) take the library path.
__device__ __forceinline__ void pe_sincos(float v, float& sn, float& cs) {
    if (__builtin_expect(fabsf(v) > 1.0e5f, 0)) {
        sincosf(v, &sn, &cs);
        return;
    }
    const float k = rintf(v * 0x1.45f306p-1f);
    float r = __builtin_fmaf(-k, 0x1.921fb6p+0f, v);
    r = __builtin_fmaf(-k, -0x1.777a5cp-25f, r);
    r = __builtin_fmaf(-k, -0x1.ee59dap-50f, r);
    const float z = r * r;
    const float ps = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float pc = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                                    __builtin_fmaf(-0.5f, z, 1.f));
    const int q = (int)k;
    const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

// 4 consecutive points (one accumulator quad) of one feature = one 8-byte piece of an SH array
struct Quad16 { _Float16 v[4]; };
// Two quads of one feature, from consecutive 8-point blocks A and B: this lane holds points r4..r4+3 of each (r4 = 0
// for lanes 0-31, 4 for lanes 32-63).  v_permlane32_swap hands lanes 0-31 the other half of block A and lanes 32-63
// the other half of block B, so every lane stores ONE whole 16-byte unit (block A + (lane >> 5)) instead of two
// 8-byte pieces: half the store instructions of an epilogue (the vector-memory queue is shared with the weight
// prefetch, and 8-byte stores are issue-bound).
__device__ __forceinline__ uint4 sh_pair_unit(const uint2 a, const uint2 b) {
    const auto w0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    const auto w1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    return uint4{w0[0], w1[0], w0[1], w1[1]};
}
__device__ __forceinline__ uint4 sh_pair_unit(const Quad16& qa, const Quad16& qb) {
    return sh_pair_unit(__builtin_bit_cast(uint2, qa), __builtin_bit_cast(uint2, qb));
}

// f32 scratch inside the planes' PE columns [256,320): 64 floats per row, floats [0,32) in the hi plane, [32,64)
// in the lo plane (slot 32 + i/4 of the plane, swizzled like everything else).
__device__ __forceinline__ float* fscr(_Float16* Th, _Float16* Tl, int row, int i) {
    _Float16* pl = (i & 32) ? Tl : Th;
    return reinterpret_cast<float*>(pl + row * LD + (((32 + ((i & 31) >> 2)) ^ hsw(row)) << 3)) + (i & 3);
}

}  // namespace mlp
