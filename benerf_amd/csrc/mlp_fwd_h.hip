// K3 forward, split-f16 variant: same network, tiling and raw outputs as mlp_fwd.hip, but every f32 operand x
// is carried as two f16 numbers  x = hi + lo * 2^-11  (hi = rn16(x), lo = rn16((x - hi) * 2^11)) and every
// product a*b is evaluated as  hi_a*hi_b + (hi_a*lo_b + lo_a*hi_b) * 2^-11  on v_mfma_f32_32x32x16_f16 with
// f32 accumulation.  The dropped lo*lo term is <= 2^-22 |a b|, i.e. the operands keep 22 significant bits
// (f32: 24) and the sums are accumulated in f32 exactly like the f32 MFMA does: measured against an f64
// evaluation the result error equals the exact-f32 kernel's (every K3 test in tests/test_kernels_gpu.py runs in
// both modes; test_mlp_modes_agree_at_full_size compares them at BASELINE size).
// Three f16 MFMAs (16-deep, 32 cycles each) replace eight f32 MFMAs (2-deep, 32 cycles each).
//
// LDS: two f16 planes Th/Tl[128][320] (hi / scaled lo) = 160 KiB: one workgroup of 8 waves per CU and 128-point tile.
// 16-byte slots (8 halfs) are XOR-swizzled: element (row, col) lives in slot (col>>3) ^ ((row>>1)&7).
// Range: |x| must stay below 65504 (f16 max) - NeRF activations are O(1..100); every launch folds max|activation|
// into the caller's status word once it passes 2^15 (mlp_split.h, 'Range guard'), so a violation is never silent.
// Training mode saves the activations for the backward kernels: BENERF_MLP_SPLIT (SAVE == 2) - f16 hi halves + 8-bit residual codes
// straight from the epilogue's registers in the SP layout, sign bits through the scalar path (DirectSave below; the narrow hv / PE
// arrays as SH arrays from the finished planes); BENERF_MLP_SPLIT_F16BWD (SAVE == 1) - the hi halves as SH arrays (mlp_split.h).
#define BENERF_HSW_V2      // this kernel's planes use the round-5 slot swizzle (mlp_split.h: hsw)
#include "mlp_split.h"
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "mlp_fwd_h.hip is written for gfx950: scalar stores (s_store_dwordx2 + s_dcache_wb), aux = 2 cache policy, v_permlane32_swap"
#endif
// Cache policy of the saved-operand stores (aux of raw_buffer_store on gfx950: 0 default, 1 = sc0, 2 = nt, 16 = sc1).  The activations are
// written once and read ~2 ms later by another kernel - 2.8 GB per 522 k-point launch streaming through the 4 MiB L2 of an XCD, where the
// weight fragments every K-loop re-reads (2.3 MB per network) live.  nt: C2 step 7.96-7.99 -> 7.66-7.72 ms with the dX kernel's stores
// and the dW kernels' loads (forward 1.28 -> 1.22 ms in the step; sc0 7.88, sc1 8.00, sc1 + nt = nt: profiles/r05_cache_policy_ab.log).
#ifndef FWD_ST_AUX
#define FWD_ST_AUX 2
#endif

// -DBENERF_TRACE_FWD: thread 0 of the first 2048 workgroups stamps the 100 MHz wall clock at the phase boundaries into the `raw`
// output (which is then not written) - tools/experiments/trace_phases.py prints the per-phase durations behind DESIGN.md 4.
#ifdef BENERF_TRACE_FWD
#define TRF(i) do { if (tid == 0 && blockIdx.x < 2048) reinterpret_cast<unsigned long long*>(a.raw)[blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#else
#define TRF(i) do { } while (0)
#endif
namespace {
using namespace mlp;

struct FwdArgs {
    const float* rays_o;
    const float* rays_d;
    const float* viewdirs;
    const float* z;
    const float* packed;     // split-f16 section of the packed buffer
    const float* bias[10];
    const float* w_alpha;
    const float* b_alpha;
    const float* w_rgb;
    const float* b_rgb;
    const float* bias_c;     // fused feature -> views bias b_c [128] (mlp_common.h: PF_VIEWSC)
    const float* pe_w;       // BARF c2f column weights (include/benerf_hip.h) or null
    float* raw;
    float* acts;
    uint32_t* status;        // [0] sticky / [3] per-call max |activation| bits, written only when >= 2^15 (may be null)
    int64_t M;
    int S;
};

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// buffer descriptor on a wave-uniform base address (mlp_bwd_h.hip)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p) {
    const uint64_t wa = reinterpret_cast<uint64_t>(p);
    const uint64_t wau = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wa >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wa);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(wau), 0, 0x7fffffff, 0x00020000);
}

// The stage GEMMs run as the TRANSPOSED product (gemm_stage<..., SWAP>: weights as the MFMA A operand): a lane of the
// accumulator holds ONE point (row r*32 + lane&31 of the tile) and, per column tile, 16 features in four quads of
// consecutive features 8q + 4*(lane>>5) + 0..3.  The lane pair (l, l+32) therefore holds the 8 consecutive features of a
// 16-byte plane slot: after one v_permlane32_swap per register pair every lane writes whole slots - 16 ds_write_b128 per
// wave and stage for both planes instead of 256 two-byte writes.
// The bias enters as the INITIAL value of the hi x hi accumulator (acc_init_bias; requested one stage ahead, at the top of the
// previous epilogue), which saves the epilogue an addition per element and a load behind its predecessor's stores.
template <int NCT>
__device__ __forceinline__ void load_bias(const float* __restrict__ bias, int ct0, int lane, float4 (&bq)[NCT][4]) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[c][q] = *reinterpret_cast<const float4*>(bias + (ct0 + c) * 32 + 8 * q + 4 * (lane >> 5));
}
template <int NR, int NCT>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc1)[NR][NCT], const float4 (&bq)[NCT][4]) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc1[r][c][q * 4 + 0] = bq[c][q].x;
                acc1[r][c][q * 4 + 1] = bq[c][q].y;
                acc1[r][c][q * 4 + 2] = bq[c][q].z;
                acc1[r][c][q * 4 + 3] = bq[c][q].w;
            }
}

// BENERF_MLP_SPLIT training launches (SAVE == 2, round 5): the stage's activations leave for HBM straight from the epilogue's
// registers - what a lane holds after the lane exchange IS a 16-byte unit of the SP layout (mlp_split.h: 8 features of one
// point), hi and scaled lo side by side for the residual codes - and the ReLU sign bits go out through the scalar path
// (v_cmp_gt_f32 -> SGPR pair -> s_store_dwordx2).  No transpose reads of the finished planes, no per-pair address arithmetic.
struct DirectSave {
    __amdgpu_buffer_rsrc_t rs, rs8;   // this tile's part of the layer's SP array (hi halves) / of its lo8 twin
    const uint32_t* mask;             // wave-uniform: sign-bit word ((T * 4 + 0) * 8 + ct) * 32 of the layer, or null (no ReLU)
};

// combine the two accumulators (+ReLU) -> both LDS planes (hi, scaled lo) [-> HBM: sv].
// (Tried in round 5 and not kept: requesting the next stage's first weight fragments from inside this epilogue, in front of its
// stores - vector-memory operations retire in order.  Training launch 1.993 -> 1.975 ms at 522 k points, inside the run-to-run
// spread, for five spilled registers: profiles/r05_fwd_direct_save_ab.log.  The other way round - the finished units wait in
// registers and are stored at the head of the next K-loop, behind its first fragment requests - measured 1.655-1.672 -> 1.677-1.681 ms
// with the last row tile's units deferred (all that fits without spills inside the loop): profiles/r05_fwd_deferred_stores_ab.log.)
template <int NCT, bool RELU, int NR, bool SV = false>
__device__ __forceinline__ void epilogue_t(f32x16 (&acc1)[NR][NCT], f32x16 (&acc2)[NR][NCT], _Float16* __restrict__ Th,
                                           _Float16* __restrict__ Tl, int ct0, int lane, float& amax, const DirectSave* sv = nullptr) {
    static_assert(!SV || NCT == 1, "direct save: one column tile per wave");
    const int pl = lane & 31, hf = lane >> 5;
    // byte offset of this lane's unit (16-point chunk pl >> 4 of the tile, slot 4 ct + hf, point pl & 15) in the tile's part of the SP
    // array; slot + 2 i is an immediate of the store (i * 512), row tile r two chunks (r * 16 384; the lo8 twin: half of everything)
    const int vo = SV ? (((pl >> 4) * 32 + ct0 * 4 + hf) * 16 + (pl & 15)) * 16 : 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int row = r * 32 + pl, sw = hsw(row);
        _Float16* rowh = Th + row * LD;
        _Float16* rowl = Tl + row * LD;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            uint2 qh[4], ql[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t wh[2], wl[2];
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int e = q * 4 + jp * 2;
                    // v = acc1 + acc2 * 2^-11 for the pair as ONE v_pk_fma_f32 (accumulator registers e, e + 1 are an aligned pair)
                    float2v v;
                    {
                        const float2v a1 = {acc1[r][c][e], acc1[r][c][e + 1]}, a2 = {acc2[r][c][e], acc2[r][c][e + 1]};
                        const float2v sc = {LO_INV, LO_INV};
                        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(a2), "v"(sc), "v"(a1));
                    }
                    if (RELU) {     // asm: fmaxf() on an asm result costs a canonicalising v_max_f32 v, v, v in front of the real one
                        asm("v_max_f32 %0, 0, %0" : "+v"(v[0]));
                        asm("v_max_f32 %0, 0, %0" : "+v"(v[1]));
                    }
                    // range guard: amax = max(amax, |v0|, |v1|) as ONE v_max3_f32 (the compiler's IEEE-mode fmaxf chain is two or three)
                    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v[0]), "v"(v[1]));
                    if (SV && RELU) {     // sign bits of accumulator elements e, e + 1: 64-lane masks, 8 bytes each, through the scalar cache
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            uint64_t m;
                            asm volatile("v_cmp_gt_f32_e64 %0, %1, 0" : "=s"(m) : "v"(v[t]));
#ifndef FWD_NO_MASK_STORE     /* timing variant (dX then reads stale masks): what do the scalar stores cost? */
                            asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m), "s"(sv->mask), "n"((r * 8 * 32 + 2 * (e + t)) * 4) : "memory");
#endif
                        }
                    }
                    const half2v hi = __builtin_convertvector(v, half2v);
                    // lo = rn16((v - hi) * 2^11) = rn16(fma(-hi, 2^11, v * 2^11)): every step exact in f32 except the final rounding,
                    // i.e. the same bits as "convert back, subtract, scale, convert" in three instructions per pair instead of five
                    // (v_pk_mul_f32, v_fma_mixlo_f16, v_fma_mixhi_f16: hi is read as f16 by the mixed-precision FMA)
                    const float2v vs = v * LO_SCALE;
                    uint32_t l;
                    asm("v_fma_mixlo_f16 %0, -%1, %4, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, -%1, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                        : "=&v"(l) : "v"(__builtin_bit_cast(uint32_t, hi)), "v"(vs[0]), "v"(vs[1]), "s"(LO_SCALE));
                    wh[jp] = __builtin_bit_cast(uint32_t, hi);
                    wl[jp] = l;
                }
                qh[q] = uint2{wh[0], wh[1]};
                ql[q] = uint2{wl[0], wl[1]};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {   // lanes 0-31 end up with slot 2i, lanes 32-63 with slot 2i + 1 of the column tile
                const int slot = (ct0 + c) * 4 + 2 * i + hf;
                const uint4 uh = sh_pair_unit(qh[2 * i], qh[2 * i + 1]), ul = sh_pair_unit(ql[2 * i], ql[2 * i + 1]);
                *reinterpret_cast<uint4*>(rowh + ((slot ^ sw) << 3)) = uh;
                *reinterpret_cast<uint4*>(rowl + ((slot ^ sw) << 3)) = ul;
                if (SV) {
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    const uint32_t uh4[4] = {uh.x, uh.y, uh.z, uh.w}, ul4[4] = {ul.x, ul.y, ul.z, ul.w};
                    const uint2 code = h8_encode_unit<11>(uh4, ul4);
                    // vector offset + immediate, zero scalar offset (mlp_bwd_h.hip: the scalar-offset form of a 16-byte store reads its data late)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{uh.x, uh.y, uh.z, uh.w}, sv->rs, vo + i * 512 + r * 16384, 0, FWD_ST_AUX);
#ifndef FWD_SKIP_LO_STORE    // timing variants only (tools/experiments/build_variant.sh)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{code.x, code.y}, sv->rs8, vo / 2 + i * 256 + r * 8192, 0, FWD_ST_AUX);
#endif
                }
            }
        }
    }
}

// Training mode, after the barrier behind epilogue_t: the finished hi plane -> the SH activation array of width W (tile part
// at `st_tile`) and the ReLU sign-bit word.  ds_read_b64_tr_b16 (tools/hwprobe/tr_read.hip: within a 16-lane group lane t
// supplies the address of row t >> 2, halfs 4 (t & 3) .. +3 of a 4 x 16 block and receives column t) hands lane l the 4
// points 4 (l >> 5) .. +3 of an 8-point block for feature l & 31 of the column tile: the lane pair (l, l + 32) again
// holds one 16-byte SH unit.  Sign bits: bit (c * NBLK + b) * 4 + j = point 8b + 4 (l >> 5) + j of column tile c; with
// NBLK = 8 that is the accumulator layout of the un-transposed 64-point product the dX kernel reads (mlp_common.h).
// bit j = (half j of the four f16 in q != 0): two v_pk_min_u16 against 1, then three bit operations
__device__ __forceinline__ uint32_t nonzero4(uint2 q) {
    // asm: written with __builtin_elementwise_min the compiler turns "min(x, 1)" into per-half compares and selects
    // (v_cmp_ne_u16 + v_cndmask, ~25 VALU instructions per block pair instead of 12)
    uint32_t t0, t1;
    const uint32_t one = 0x00010001u;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t0) : "v"(q.x), "v"(one));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t1) : "v"(q.y), "v"(one));
    const uint32_t u = t0 | (t1 << 2);            // bits 0, 16, 2, 18
    return (u | (u >> 15)) & 0xFu;
}

// One block pair (blocks 2bp, 2bp + 1) of column tile ct: the unit of work the layer loop spreads over its k-steps.
template <int W, bool MASK, int NBLK>
__device__ __forceinline__ void save_pair(const _Float16* __restrict__ Th, int ct, int bp, int lane, __amdgpu_buffer_rsrc_t rs, uint64_t& bits) {
    asm volatile("" : "+v"(lane));      // the addresses below are cheap: recomputed per call, not hoisted out of the layer loop and spilled
    const int t = lane & 15, g = lane >> 4, hf = lane >> 5, pl = lane & 31;
    const int n = ct * 32 + pl;
    const int col = ct * 32 + 16 * (g & 1) + 4 * (t & 3);
    const int rsub = 4 * (g >> 1) + (t >> 2);
    uint2 q[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int b = 2 * bp + k, row = b * 8 + rsub;
        const _Float16* src = Th + row * LD + ((((col >> 3) ^ hsw(row)) << 3) | (col & 7));
        const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(src)));
        q[k] = __builtin_bit_cast(uint2, v);
        if (MASK) bits |= (uint64_t)nonzero4(q[k]) << (b * 4);        // post-ReLU: > 0 <=> != 0
    }
    const uint4 u = sh_pair_unit(q[0], q[1]);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{u.x, u.y, u.z, u.w}, rs, (((2 * bp + hf) * W + n) * 8) * 2, 0, FWD_ST_AUX);
}

template <int NCT, int W, bool MASK, int NBLK>
__device__ __forceinline__ uint64_t save_tile(const _Float16* __restrict__ Th, int ct0, int lane, const _Float16* __restrict__ st_tile) {
    static_assert(NCT * NBLK * 4 <= 64, "one sign-bit word per thread");
    const int t = lane & 15, g = lane >> 4, hf = lane >> 5, pl = lane & 31;
    const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(st_tile);
    uint64_t bits = 0;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int n = (ct0 + c) * 32 + pl;
        const int col = (ct0 + c) * 32 + 16 * (g & 1) + 4 * (t & 3);      // first of the 4 halfs this lane addresses
        const int rsub = 4 * (g >> 1) + (t >> 2);                         // its row inside an 8-point block
#pragma unroll
        for (int bp = 0; bp < NBLK / 2; ++bp) {
            uint2 q[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int b = 2 * bp + k, row = b * 8 + rsub;
                const _Float16* src = Th + row * LD + ((((col >> 3) ^ hsw(row)) << 3) | (col & 7));
                const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(src)));
                q[k] = __builtin_bit_cast(uint2, v);
                if (MASK) bits |= (uint64_t)nonzero4(q[k]) << (c * NBLK * 4 + b * 4);     // post-ReLU: > 0 <=> != 0
            }
            const uint4 u = sh_pair_unit(q[0], q[1]);           // lanes 0-31: block 2bp, lanes 32-63: block 2bp + 1
            // vector offset, zero scalar offset (mlp_bwd_h.hip: the scalar-offset form reads its data late)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{u.x, u.y, u.z, u.w}, rs, (((2 * bp + hf) * W + n) * 8) * 2, 0, FWD_ST_AUX);
        }
    }
    return bits;
}

// BENERF_MLP_SPLIT (SAVE == 2, the fp32-equivalent backward): the low halves leave too - as 8-bit residual codes (lo8 twin of
// the SH array, mlp_split.h): the same transpose read on the lo plane (which holds (x - hi) * 2^11), the lane pair forms the
// unit like the hi halves, h8_encode_unit<11> turns the 8 + 8 halfs into 8 bytes: one 8-byte store per lane.
// COL0: first plane column of the array's feature 0 (the encodings sit in columns [COL_PE, COL_PE + 64) of the planes)
template <int W, bool MASK, int NBLK, int COL0 = 0>
__device__ __forceinline__ void save_pair22(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, int ct, int bp, int lane,
                                            __amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t rs8, uint64_t& bits) {
    asm volatile("" : "+v"(lane));      // addresses recomputed per call, not hoisted out of the layer loop and spilled
    const int t = lane & 15, g = lane >> 4, hf = lane >> 5, pl = lane & 31;
    const int n = ct * 32 + pl;
    const int col = COL0 + ct * 32 + 16 * (g & 1) + 4 * (t & 3);
    const int rsub = 4 * (g >> 1) + (t >> 2);
    uint2 q[2], ql[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int b = 2 * bp + k, row = b * 8 + rsub;
        const int off = row * LD + ((((col >> 3) ^ hsw(row)) << 3) | (col & 7));
        const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(Th + off)));
        const short4v vl = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(Tl + off)));
        q[k] = __builtin_bit_cast(uint2, v);
        ql[k] = __builtin_bit_cast(uint2, vl);
        if (MASK) bits |= (uint64_t)nonzero4(q[k]) << (b * 4);        // post-ReLU: > 0 <=> != 0
    }
    const uint4 u = sh_pair_unit(q[0], q[1]);
    const uint4 ul = sh_pair_unit(ql[0], ql[1]);
    const uint32_t uh4[4] = {u.x, u.y, u.z, u.w}, ul4[4] = {ul.x, ul.y, ul.z, ul.w};
    const uint2 code = h8_encode_unit<11>(uh4, ul4);
    const int unit = (2 * bp + hf) * W + n;
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{u.x, u.y, u.z, u.w}, rs, unit * 16, 0, FWD_ST_AUX);
#ifndef FWD_SKIP_LO_STORE    // timing variants only (tools/experiments/build_variant.sh)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{code.x, code.y}, rs8, unit * 8, 0, FWD_ST_AUX);
#endif
}
// a whole tile part (column tile ct, NBLK blocks): hi array at `st_tile` (halfs), codes at `st8_tile` (bytes)
template <int W, bool MASK, int NBLK>
__device__ __forceinline__ uint64_t save_tile22(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, int ct, int lane,
                                                const _Float16* __restrict__ st_tile, const uint8_t* __restrict__ st8_tile) {
    const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(st_tile), rs8 = uniform_rsrc(st8_tile);
    uint64_t bits = 0;
#pragma unroll
    for (int bp = 0; bp < NBLK / 2; ++bp) save_pair22<W, MASK, NBLK>(Th, Tl, ct, bp, lane, rs, rs8, bits);
    return bits;
}

// offset of the forward block of hidden layer l (1..7, l != 5 at the call site) without the generic pack_offset() summation,
// which becomes a scalar loop of branches for a run-time l
__device__ __forceinline__ int fwd_layer_offset(int l) {
    return (int)pack_offset(PF_L1) + (l - 1) * (int)pack_floats(PF_L1) + (l > 5 ? (int)(pack_floats(PF_L5) - pack_floats(PF_L1)) : 0);
}
static_assert(pack_offset(PF_L1) + 3 * pack_floats(PF_L1) == pack_offset(PF_L4) &&
              pack_offset(PF_L1) + 5 * pack_floats(PF_L1) + (pack_floats(PF_L5) - pack_floats(PF_L1)) == pack_offset(PF_L6) &&
              pack_offset(PF_L1) + 6 * pack_floats(PF_L1) + (pack_floats(PF_L5) - pack_floats(PF_L1)) == pack_offset(PF_L7), "fwd_layer_offset");

// One workgroup of 8 waves per 128 points (the whole LDS: two planes [128][320] f16); wave w owns column tile w (32 output
// features) x all four point tiles: every weight fragment pair (hi, lo: 2 KiB) feeds 12 MFMAs - half the fragment bytes
// per MFMA of the 64-point / 2 x 2 tiling, and the 8-register ring slot leaves room for a 4-deep prefetch.
constexpr int FTM = 128, FNT = 512;
constexpr size_t FWD_SMEM = (size_t)2 * FTM * LD * sizeof(_Float16);      // 163 840 B

// SAVE: 0 inference, 1 training with the f16 backward (hi halves saved), 2 training with the 22-bit backward (hi + lo)
// FUSE: the feature layer folded into the views layer (mlp_common.h: PF_VIEWSC) - no FEAT stage; BENERF_MLP_SPLIT_F16BWD, whose
// backward kernels want `feature` saved, runs the unfused sequence (training AND inference, so that they agree bit for bit)
template <int C, int SAVE, bool FUSE = (SAVE != 1)>
__global__ __launch_bounds__(FNT, 1) void mlp_fwd_split_kernel(FwdArgs a) {
    // weight-fragment prefetch depth (k-steps).  The SAVE == 2 launch runs its K-loops behind the previous epilogue's 24 activation
    // stores: three k-steps ahead measured 1.954 -> 1.921 ms at 522 k points (no spill since the direct save freed registers);
    // the inference launch is indifferent (1.494 / 1.498), the SAVE == 1 variant spills at 3 (profiles/r05_fwd_direct_save_ab.log)
#ifdef FWD_FPF
    constexpr int FPF = FWD_FPF;
#else
    constexpr int FPF = SAVE == 2 ? 3 : 2;
#endif
    extern __shared__ __attribute__((aligned(16))) _Float16 Tsm[];   // Th | Tl
    _Float16* Th = Tsm;
    _Float16* Tl = Tsm + FTM * LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7, wave-uniform: weight pointers stay scalar
    const int64_t m0 = (int64_t)blockIdx.x * FTM;
    // timing variant (tools/experiments/build_variant.sh -DFWD_STORE_WINDOW=256): every tile saves into the first WINDOW points of
    // the arrays - same instructions, the bytes stay in L2: separates the cost of ISSUING the stores from the HBM write path
#ifdef FWD_STORE_WINDOW
    const int64_t ms0 = m0 & (int64_t)(FWD_STORE_WINDOW - 1);
#else
    const int64_t ms0 = m0;
#endif
    const int64_t M = a.M;
    const int pt = tid & (FTM - 1);
    const int grp = tid >> 7;                                        // 4 thread groups per point
    const int64_t m = m0 + pt;
    const int64_t mc = m < M ? m : M - 1;
    const int64_t ray = (int64_t)((uint32_t)mc / (uint32_t)a.S);      // M < 2^31 (launcher): a 32-bit division
    float* acts = a.acts;
    const int64_t Mp = m_pad(M);
    _Float16* st_h = SAVE ? reinterpret_cast<_Float16*>(acts + sact_h(Mp, 0)) : nullptr;     // layer l: + l * Mp * 256 halfs
    // sign-bit words in the dX kernel's layout: uint64 [layer][64-point tile][256 threads = wave' (4) x lane]; this wave
    // produces the 32 bits of column tile wave & 1 of wave' = wave >> 1 for both 64-point tiles of the workgroup
    uint32_t* mask_out = SAVE ? reinterpret_cast<uint32_t*>(reinterpret_cast<uint64_t*>(acts + sact_mask(Mp)) +
                                                            (int64_t)blockIdx.x * 2 * NTHREADS + (wave >> 1) * 64 + lane) + (wave & 1)
                              : nullptr;
    const int64_t mask_stride = (Mp / TM) * NTHREADS * 2;            // uint32 units between layers
    auto store_bits = [&](int layer, uint64_t b) {
        mask_out[layer * mask_stride] = (uint32_t)b;                               // points 0..63
        mask_out[layer * mask_stride + 2 * NTHREADS] = (uint32_t)(b >> 32);        // points 64..127: the next 64-point tile
    };
    const bool live = m < M;
    if (SAVE && blockIdx.x == 0 && tid == 0) reinterpret_cast<uint32_t*>(acts + sact_info(Mp))[SI_TAG] = SAVE == 2 ? SACT_TAG_SPLIT22 : SACT_TAG_SPLIT;
    // SAVE == 2: byte i of the lo8 region <-> half i of the SH region (mlp_split.h)
    uint8_t* st8_h = SAVE == 2 ? reinterpret_cast<uint8_t*>(acts + sact_lo8_base(Mp)) : nullptr;          // layer l: + l * Mp * 256 bytes
    // SAVE == 2: the direct save of stage `layer` (0..7: h_layer with its sign bits) - mlp_split.h, SP layout
    auto direct = [&](int layer, bool relu) {
        DirectSave d;
        d.rs = uniform_rsrc(SAVE == 2 ? st_h + ((int64_t)layer * Mp + ms0) * 256 : nullptr);
        d.rs8 = uniform_rsrc(SAVE == 2 ? st8_h + ((int64_t)layer * Mp + ms0) * 256 : nullptr);
        d.mask = (SAVE == 2 && relu) ? reinterpret_cast<const uint32_t*>(acts + sact_mask(Mp)) + (int64_t)layer * mask_stride +
                                           ((int64_t)blockIdx.x * 32 + wave) * 32
                                     : nullptr;
        return d;
    };
    float amax = 0.f;        // running max |activation| of this thread (range guard)
    // f32 scratch in the dead PE columns [288,320) of the lo plane: logical slot 36 + j of this thread's row
    const int psw = hsw(pt);
    auto scratch = [&](int j) { return reinterpret_cast<float*>(Tl + pt * LD + (((36 + j) ^ psw) << 3)); };

    TRF(0);
#ifdef BENERF_TRACE_FWD
    if (tid == 0 && blockIdx.x < 2048) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); reinterpret_cast<unsigned long long*>(a.raw)[blockIdx.x * 32 + 31] = ((unsigned long long)xcc << 32) | hw; }
#endif
    // ---- prologue: pts = o + d*z (separately rounded like torch), PE(pts) -> planes (+ acts) ------------
    {
        const float zz = a.z[mc];
        float x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = __fadd_rn(a.rays_o[ray * 3 + c], __fmul_rn(a.rays_d[ray * 3 + c], zz));
        // training, SAVE == 1: PE as f32 rows (sin/cos derivatives in dX; dW operand of the thin instances), all Mp rows.  Staged in
        // the still unused columns [0,128) of this point's hi-plane row and written out below as whole 16-byte units
        // (a 4-byte store per thread and column touches 64 cache lines per instruction); the columns are rotated by
        // 4 * point so that the 64 lanes of a staging write do not all hit one bank.
        float* stage = reinterpret_cast<float*>(Th) + pt * (LD / 2);
        auto put = [&](int col, float v) {
            if (SAVE == 1) stage[(col + 4 * pt) & 63] = v;           // saved UNWEIGHTED (dX needs sin / cos themselves)
            if (a.pe_w) v *= a.pe_w[col];
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)((v - (float)hi) * LO_SCALE);
            const int idx = hidx(pt, COL_PE + col);
            Th[idx] = hi;
            Tl[idx] = lo;
        };
        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                put(c, x[c]);
                amax = fmaxf(amax, fabsf(x[c]));
            }
            put(63, 0.f);
        }
        for (int p = grp; p < 30; p += 4) {                    // model/embedder.py:13-28
            const int f = p / 3, d = p - 3 * f;
            const float v = x[d] * (float)(1 << f);
            float s, c;
            pe_sincos(v, s, c);
            put(3 + f * 6 + d, s);
            put(3 + f * 6 + 3 + d, c);
        }
    }
    lds_barrier();
    TRF(1);
    if (SAVE == 2) {
        // the 22-bit backward: the encoding leaves as an SH array + lo8 twin straight from the planes' PE columns (the thin dW
        // instances load MFMA fragments from it like from every other saved operand), and the point itself (+ its view
        // direction) as f32 for the dX kernel, which recomputes sin / cos (mlp_split.h: sact22_*).  Wave w: column tile w & 1,
        // block pairs 2 (w >> 1), + 1.  The PE columns stay untouched until the views stage: no barrier behind this.
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(reinterpret_cast<_Float16*>(acts + sact22_pe_hi(Mp)) + ms0 * ACT_PE_W);
        const __amdgpu_buffer_rsrc_t rs8 = uniform_rsrc(reinterpret_cast<uint8_t*>(acts + sact22_pe_lo8(Mp)) + ms0 * ACT_PE_W);
        uint64_t nobits = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) save_pair22<ACT_PE_W, false, 16, COL_PE>(Th, Tl, wave & 1, (wave >> 1) * 2 + i, lane, rs, rs8, nobits);
        if (grp == 0) {
            const float zz = a.z[mc];
            float4* dst = reinterpret_cast<float4*>(acts + sact22_pts(Mp)) + m * 2;
            dst[0] = make_float4(__fadd_rn(a.rays_o[ray * 3 + 0], __fmul_rn(a.rays_d[ray * 3 + 0], zz)),
                                 __fadd_rn(a.rays_o[ray * 3 + 1], __fmul_rn(a.rays_d[ray * 3 + 1], zz)),
                                 __fadd_rn(a.rays_o[ray * 3 + 2], __fmul_rn(a.rays_d[ray * 3 + 2], zz)), 0.f);
            dst[1] = make_float4(a.viewdirs[ray * 3 + 0], a.viewdirs[ray * 3 + 1], a.viewdirs[ray * 3 + 2], 0.f);
        }
    }
    if (SAVE == 1) {
        float4* pe_tile = reinterpret_cast<float4*>(acts + sact_pe32(Mp) + ms0 * ACT_PE_W);
#pragma unroll
        for (int k = 0; k < FTM * ACT_PE_W / 4 / FNT; ++k) {
            const int u = tid + k * FNT, row = u >> 4, c4 = u & 15;
            pe_tile[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Th) + row * (LD / 2) + ((4 * c4 + 4 * row) & 63));
        }
        lds_barrier();          // the staging columns are layer 0's output columns
    }

    f32x16 acc1[4][1], acc2[4][1];

    // ---- L0 ---------------------------------------------------------------------------------
    float4 bq[1][4];
    load_bias<1>(a.bias[0], wave, lane, bq);
    acc_init_bias(acc1, bq);
    zero_acc(acc2);
    gemm_stage<4, 1, FPF, true, 4>(Th, Tl, COL_PE, a.packed + pack_offset(PF_L0), wave, lane, acc1, acc2);
    load_bias<1>(a.bias[1], wave, lane, bq);
    {
        const DirectSave d0 = direct(0, true);
        epilogue_t<1, true, 4, SAVE == 2>(acc1, acc2, Th, Tl, wave, lane, amax, &d0);
    }
    lds_barrier();
    TRF(2);

    // ---- L1..L7 -------------------------------------------------------------------------------
#pragma unroll 1
    for (int l = 1; l < 8; ++l) {
        // SAVE == 1 (BENERF_MLP_SPLIT_F16BWD: hi halves only, SH layout): the previous layer's activations leave for HBM during this
        // layer's LAST two k-steps (four block pairs each: two transpose reads of the finished hi plane, the sign bits, one 16-byte
        // store per pair): behind the K-loop's last weight-fragment request, so that no fragment is queued behind a store (in-order
        // retirement), and in the gaps between the MFMAs.  SAVE == 2 (BENERF_MLP_SPLIT) saves from the epilogue's registers instead
        // (DirectSave; rounds 3-4 spread transpose reads of BOTH planes + the codec over the last eight k-steps here).
        uint64_t pbits = 0;
        const __amdgpu_buffer_rsrc_t prs = uniform_rsrc(SAVE == 1 ? st_h + ((int64_t)(l - 1) * Mp + ms0) * 256 : nullptr);
        auto save_at = [&](int ks, int last) {      // `last` = the loop's final k-step
            if (SAVE == 1 && ks >= last - 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) save_pair<256, true, 16>(Th, wave, (ks - (last - 1)) * 4 + i, lane, prs, pbits);
                if (ks == last) store_bits(l - 1, pbits);
            }
        };
        acc_init_bias(acc1, bq);
        zero_acc(acc2);
        if (l == 5) gemm_stage<20, 1, FPF, true, 4>(Th, Tl, 0, a.packed + pack_offset(PF_L5), wave, lane, acc1, acc2, NoAfterHead(),
                                                    [&](int ks) { save_at(ks, 19); });
        else gemm_stage<16, 1, FPF, true, 4>(Th, Tl, 0, a.packed + fwd_layer_offset(l), wave, lane, acc1, acc2, NoAfterHead(),
                                             [&](int ks) { save_at(ks, 15); });
        lds_barrier();
        TRF(3 + 2 * (l - 1));
        load_bias<1>(a.bias[l < 7 ? l + 1 : BENERF_L_FEAT], wave, lane, bq);
        {
            const DirectSave dl = direct(l, true);
            epilogue_t<1, true, 4, SAVE == 2>(acc1, acc2, Th, Tl, wave, lane, amax, &dl);
        }
        lds_barrier();
        TRF(4 + 2 * (l - 1));
    }
    if (SAVE == 1) store_bits(7, save_tile<1, 256, true, 16>(Th, wave, lane, st_h + ((int64_t)7 * Mp + ms0) * 256));

    // ---- alpha partials (reads h7) + PE(viewdir) into columns [256,288) ---------------------------
    {
        const float* wa = a.w_alpha + grp * 64;
        float s = 0.f;
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {
            float h[8];
            load8(Th, Tl, pt * LD + (((grp * 8 + q) ^ psw) << 3), h);
            const float4 w0 = *reinterpret_cast<const float4*>(wa + q * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(wa + q * 8 + 4);
            s += h[0] * w0.x + h[1] * w0.y + h[2] * w0.z + h[3] * w0.w;
            s += h[4] * w1.x + h[5] * w1.y + h[6] * w1.z + h[7] * w1.w;
        }
        scratch(0)[grp] = s;
        float vd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) vd[c] = a.viewdirs[ray * 3 + c];
        float* aped = SAVE == 1 ? acts + sact_ped32(Mp) + m * ACT_PED_W : nullptr;
        auto put = [&](int col, float v) {
            if (SAVE == 1) {
                aped[col] = v;
            }
            if (a.pe_w) v *= a.pe_w[64 + col];
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)((v - (float)hi) * LO_SCALE);
            const int idx = hidx(pt, COL_PE + col);
            Th[idx] = hi;
            Tl[idx] = lo;
        };
        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) put(c, vd[c]);
        }
        if (grp == 1) {
#pragma unroll
            for (int k = 27; k < 32; ++k) put(k, 0.f);
        }
        for (int p = grp; p < 12; p += 4) {
            const int f = p / 3, d = p - 3 * f;
            const float v = vd[d] * (float)(1 << f);
            float sn, cs;
            pe_sincos(v, sn, cs);
            put(3 + f * 6 + d, sn);
            put(3 + f * 6 + 3 + d, cs);
        }
    }

    if (!FUSE) {
        // ---- FEAT (linear) ------------------------------------------------------------------------
        acc_init_bias(acc1, bq);
        zero_acc(acc2);
        gemm_stage<16, 1, FPF, true, 4>(Th, Tl, 0, a.packed + pack_offset(PF_FEAT), wave, lane, acc1, acc2);
        lds_barrier();
        load_bias<1>(a.bias[BENERF_L_VIEWS], wave & 3, lane, bq);
        epilogue_t<1, false, 4>(acc1, acc2, Th, Tl, wave, lane, amax);
    } else {
        // no FEAT stage: VIEWS runs on [h7 | PE(dir)] with W_c = W_v[:, :256] W_f, bias b_c = W_v[:, :256] b_f + b_v
        lds_barrier();      // the alpha partials and the PE(dir) columns are visible
        TRF(17);
        load_bias<1>(a.bias_c, wave & 3, lane, bq);
    }
    if (tid < FTM && live) {
        const float4 p = *reinterpret_cast<const float4*>(scratch(0));
#ifndef BENERF_TRACE_FWD
        a.raw[m * (C + 1) + C] = ((p.x + p.y) + (p.z + p.w)) + a.b_alpha[0];
#endif
    }
    if (!FUSE) lds_barrier();       // feature is in the planes
    if (SAVE == 1) save_tile<1, 256, false, 16>(Th, wave, lane, reinterpret_cast<_Float16*>(acts + sact_feat(Mp)) + ms0 * 256);
    if (SAVE == 2) {
        // PE(dir) (planes' columns [256,288), visible since the barrier behind the FEAT GEMM) as an SH array of width 32 + lo8
        // twin: one block pair per wave
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(reinterpret_cast<_Float16*>(acts + sact22_ped_hi(Mp)) + ms0 * ACT_PED_W);
        const __amdgpu_buffer_rsrc_t rs8 = uniform_rsrc(reinterpret_cast<uint8_t*>(acts + sact22_ped_lo8(Mp)) + ms0 * ACT_PED_W);
        uint64_t nobits = 0;
        save_pair22<ACT_PED_W, false, 16, COL_PE>(Th, Tl, 0, wave, lane, rs, rs8, nobits);
    }

    // ---- VIEWS: [feature (FUSE: h7) | PE(dir)] (288) -> 128: wave w computes column tile w & 3 for the point half w >> 2 ----
    {
        const int vct = wave & 3, vrh = wave >> 2;
        _Float16* Thh = Th + vrh * 64 * LD;          // rows + 64: same swizzle
        _Float16* Tlh = Tl + vrh * 64 * LD;
        f32x16 av1[2][1], av2[2][1];
        acc_init_bias(av1, bq);
        zero_acc(av2);
        gemm_stage<18, 1, FPF, true, 2>(Thh, Tlh, 0, a.packed + pack_offset(FUSE ? PF_VIEWSC : PF_VIEWS), vct, lane, av1, av2);
        lds_barrier();
        TRF(18);
        epilogue_t<1, true, 2>(av1, av2, Thh, Tlh, vct, lane, amax);
        lds_barrier();
        TRF(19);
        if (SAVE) {  // sign bits of hv: bit b*4 + j = point 8b + 4 (lane >> 5) + j of this half, column tile = wave & 3
            _Float16* sthv = reinterpret_cast<_Float16*>(acts + sact_hv(Mp)) + (ms0 + vrh * 64) * ACT_HV_W;
            const uint64_t bits = SAVE == 2 ? save_tile22<ACT_HV_W, true, 8>(Thh, Tlh, vct, lane, sthv, st8_h + (int64_t)9 * Mp * 256 + (ms0 + vrh * 64) * ACT_HV_W)
                                            : save_tile<1, ACT_HV_W, true, 8>(Thh, vct, lane, sthv);
            reinterpret_cast<uint64_t*>(acts + sact_mask(Mp))[8 * (Mp / TM) * NTHREADS + ((int64_t)blockIdx.x * 2 + vrh) * NTHREADS + vct * 64 + lane] = bits;
        }
    }

    // ---- rgb: 128 -> C on the VALU; partials of channel c in scratch slot 1 + c ------------------------
    {
        float s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = 0.f;
#pragma unroll 2
        for (int q = 0; q < 4; ++q) {
            float h[8];
            load8(Th, Tl, pt * LD + (((grp * 4 + q) ^ psw) << 3), h);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float4 w0 = *reinterpret_cast<const float4*>(a.w_rgb + c * 128 + grp * 32 + q * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(a.w_rgb + c * 128 + grp * 32 + q * 8 + 4);
                s[c] += h[0] * w0.x + h[1] * w0.y + h[2] * w0.z + h[3] * w0.w;
                s[c] += h[4] * w1.x + h[5] * w1.y + h[6] * w1.z + h[7] * w1.w;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) scratch(1 + c)[grp] = s[c];
    }
    // range guard: one atomic per wave, only when something came within a factor of two of f16's maximum
    if (a.status) {
        const float wmax = wave_max_nonneg(amax);
        if (lane == 63 && !(wmax < 32768.f)) {
            const uint32_t bits = __float_as_uint(wmax == wmax ? wmax : __builtin_inff());
            atomicMax(a.status, bits);          // sticky maximum (benerf_mlp_status_check)
            atomicMax(a.status + 3, bits);      // maximum of this call (BENERF_MLP_AUTO gate)
        }
    }
    lds_barrier();
    if (tid < FTM && live) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 p = *reinterpret_cast<const float4*>(scratch(1 + c));
#ifndef BENERF_TRACE_FWD
            a.raw[m * (C + 1) + c] = ((p.x + p.y) + (p.z + p.w)) + a.b_rgb[c];
#endif
        }
    }
    TRF(20);
    // the sign-bit words went through the scalar data cache: scalar stores are invisible to the compiler's wait-count tracking, so
    // wait for them explicitly, then write the cache back
    if (SAVE == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}

}  // namespace

// save_lo: training launches (acts != NULL) also save the low halves (BENERF_MLP_SPLIT); fuse: the feature layer folded into the views
// layer (everything but BENERF_MLP_SPLIT_F16BWD, whose backward wants `feature`)
int benerf_mlp_fwd_split_launch(const BenerfMlpParams* params, const float* packed, int channels, int n_rays, int n_samples,
                                const float* rays_o, const float* rays_d, const float* viewdirs, const float* z, float* raw,
                                float* acts, int save_lo, int fuse, uint32_t* status, hipStream_t stream) {
    FwdArgs a;
    a.rays_o = rays_o;
    a.rays_d = rays_d;
    a.viewdirs = viewdirs;
    a.z = z;
    a.packed = packed + mlp::PACKED_FLOATS;
    a.bias_c = packed + 2 * mlp::PACKED_FLOATS + mlp::FUSED_W_FLOATS;
    for (int l = 0; l < 8; ++l) a.bias[l] = params->b[l];
    a.bias[BENERF_L_VIEWS] = params->b[BENERF_L_VIEWS];
    a.bias[BENERF_L_FEAT] = params->b[BENERF_L_FEAT];
    a.w_alpha = params->w[BENERF_L_ALPHA];
    a.b_alpha = params->b[BENERF_L_ALPHA];
    a.w_rgb = params->w[BENERF_L_RGB];
    a.b_rgb = params->b[BENERF_L_RGB];
    a.pe_w = params->pe_weights;
    a.raw = raw;
    a.acts = acts;
    a.status = status;
    a.M = (int64_t)n_rays * n_samples;
    a.S = n_samples;
    // training launches cover the padded point range (whole 128-point tiles, the dX kernel's too), inference the live tiles
    const int64_t tiles = acts ? mlp::m_pad(a.M) / FTM : (a.M + FTM - 1) / FTM;
    static_assert(mlp::SM_PAD == FTM, "the padded point count is a whole number of forward tiles");
    BENERF_REQUIRE(tiles < (1ll << 31) && a.M < (1ll << 31), "mlp_fwd(split): too many points");
    dim3 grid((unsigned)tiles), block(FNT);
    const int smem = (int)FWD_SMEM;
    BENERF_REQUIRE(!acts || (save_lo ? fuse : !fuse), "mlp_fwd(split): BENERF_MLP_SPLIT training launches are fused, BENERF_MLP_SPLIT_F16BWD ones are not");
#define BENERF_FWD_LAUNCH(CH, SV, FU)                                                                                    \
    do {                                                                                                                 \
        static BenerfLdsAttr attr_;                                                                                      \
        if (!benerf_lds_attr(attr_, (const void*)mlp_fwd_split_kernel<CH, SV, FU>, smem)) {                              \
            benerf_set_error("mlp_fwd(split): cannot reserve %d bytes of LDS", smem);                                    \
            return BENERF_EHIP;                                                                                          \
        }                                                                                                                \
        hipLaunchKernelGGL((mlp_fwd_split_kernel<CH, SV, FU>), grid, block, smem, stream, a);                            \
    } while (0)
    if (channels == 1) {
        if (acts && save_lo) BENERF_FWD_LAUNCH(1, 2, true);
        else if (acts) BENERF_FWD_LAUNCH(1, 1, false);
        else if (fuse) BENERF_FWD_LAUNCH(1, 0, true);
        else BENERF_FWD_LAUNCH(1, 0, false);
    } else {
        if (acts && save_lo) BENERF_FWD_LAUNCH(3, 2, true);
        else if (acts) BENERF_FWD_LAUNCH(3, 1, false);
        else if (fuse) BENERF_FWD_LAUNCH(3, 0, true);
        else BENERF_FWD_LAUNCH(3, 0, false);
    }
#undef BENERF_FWD_LAUNCH
    BENERF_LAUNCH_CHECK("mlp_fwd(split)");
    return BENERF_OK;
}
