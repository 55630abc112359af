// K3 backward part 1, f16 variant of mlp_bwd.hip: the same activation-gradient chain (phases P0..P6, same d_pts /
// d_viewdirs outputs; the dY arrays in SH layout, mlp_split.h) with every GEMM as TWO f16 MFMAs per product block:
// the gradient enters as f16 (11-bit operand, one rounding per layer - random, unbiased, independent from point to
// point, so it averages out in every sum over points), the transposed weight as hi + lo (two f16 numbers, ~20 bits: a
// weight's rounding error is the SAME for every point and would not average out - tools/experiments/
// lowprec_backward.py: f16 weights put 8e-4 on the pose gradients, split weights 1e-4), f32 accumulation.  With that
// the gradients of a training step move by far less than they already differ between an f32 and an f64 evaluation
// of the same step (ReLU-kink flips, ~1e-3 of the largest entry).
//
// Gradients are far outside the f16 range (d_raw ~ 1/n_rays), but the whole chain is LINEAR in d_raw: each 128-point
// tile multiplies its d_raw by a power of two s (mlp_split.h, pow2_scale6: tile maximum -> [2^6, 2^7)), runs the chain
// on the scaled values and multiplies every output by 1/s - both exact.  The dY arrays for the dW kernels are stored
// with ONE scale per call (s_s from max|d_raw| over the whole launch, computed by a small pre-kernel) so that dW can
// accumulate across tiles; the dW reduce kernel divides by s_s.
//
// Tiling: one workgroup (4 waves) per 128 points; wave w owns output features [64w, 64w + 64) x 128 points = 4 x 2
// MFMA tiles in ONE accumulator set (128 registers): every weight fragment fetched from L2 feeds four row tiles.
// LDS: one f16 plane T[128][320] = 80 KiB, two workgroups per CU.  The plane's PE columns [256,320) are never a GEMM
// operand here and serve as 32 floats of f32 scratch per point (fscr1): dPE(dir) at [0,27) during P2, the tile's
// maximum at [26] of rows 0 / 1 during P0, scaled d_raw at [28,32).  dPE (64 floats per point, layer-5 skip + layer 0)
// stays in accumulator registers from layer 5 to the end.
#include "mlp_split.h"

// -DBENERF_TRACE_DX: wave 0 of the first 2048 workgroups stamps the 100 MHz wall clock at every barrier into the d_viewdirs
// output (which is then not written) - tools/experiments/trace_dx.py decodes the per-phase durations behind DESIGN.md 4.
#ifdef BENERF_TRACE_DX
#define TR(i) do { if (tid == 0 && blockIdx.x < 2048) reinterpret_cast<unsigned long long*>(a.d_vdir)[blockIdx.x * 64 + (i)] = wall_clock64(); } while (0)
#else
#define TR(i) do { } while (0)
#endif
namespace {
using namespace mlp;

constexpr int TMB = 128;                                          // points per workgroup
constexpr size_t BWD_SMEM = (size_t)TMB * LD * sizeof(_Float16);  // 81 920 B
static_assert(TMB == SM_PAD, "the padded point count is a whole number of dX tiles");

struct BwdArgs {
    const float* d_raw;
    const float* acts;
    float* dacts;
    const float* packed;    // split-f16 section at + PACKED_FLOATS (backward blocks: hi + unscaled lo)
    const float* w_alpha;   // [256]
    const float* w_rgb;     // [C][128]
    const float* pe_w;      // BARF c2f column weights (include/benerf_hip.h) or null
    float* d_pts;           // [M][3]
    float* d_vdir;          // [M][3]
    uint32_t* status;       // [1]: max |tile-scaled gradient| bits once >= 2^15, [2]: acts buffer written by another mode (may be null)
    const float* absmax;    // max |d_raw| of the call from the compositing backward, or null: then it is in the dacts info word
    int64_t M;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// f32 scratch float i (0..31) of `row`: slot 32 + i/4 of the plane, swizzled like everything else
__device__ __forceinline__ float* fscr1(_Float16* T, int row, int i) {
    return reinterpret_cast<float*>(T + row * LD + (((32 + (i >> 2)) ^ hsw(row)) << 3)) + (i & 3);
}

// buffer descriptor on a wave-uniform base address: per-lane addresses become ONE 32-bit VGPR offset (+ a scalar offset),
// instead of 64-bit pointer pairs that the register allocator spills - and every spill reload costs an s_waitcnt vmcnt(0),
// i.e. a drain of the whole in-order vector-memory queue (the epilogue's stores, the prefetched weight fragments)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p) {
    const uint64_t wa = reinterpret_cast<uint64_t>(p);
    const uint64_t wau = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wa >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wa);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(wau), 0, 0x7fffffff, 0x00020000);
}

// weight fragments (hi plane) of a packed block through a buffer descriptor; layout: mlp_pack.hip / mlp_split.h
struct WFrag {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
    __device__ __forceinline__ WFrag(const float* wp, int lane) {
        const uint64_t wa = reinterpret_cast<uint64_t>(wp);
        const uint64_t wau = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wa >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wa);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(wau), 0, 0x7fffffff, 0x00020000);
        voff = lane * 16;
    }
    // fragment of column tile t (wave-uniform), k-step ks of a block with KS k-steps; plane 0 = hi, 1 = lo (unscaled)
    __device__ __forceinline__ u32x4 load(int t_uniform, int ks, int KS, int plane) const {
        int soff = (((t_uniform >> 1) * KS + ks) * 2 + (t_uniform & 1)) * 2048 + plane * 1024;
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    }
};

// acc[rt][c] += T[rt*32.., 0 .. KS*16) x W(tile ct0 + c), rt = 0..3: 4 x NCT tiles; per block and k-step two MFMAs,
// dY x W_hi and dY x W_lo, into the same accumulator.  Weight fragments PF k-steps ahead (a k-step is 8 * NCT MFMAs),
// activation fragments one k-step ahead.
// offset of the backward block of hidden layer l (7..1) without the generic pack_offset() summation, which the compiler
// turns into a scalar loop full of branches when l is a run-time value
__device__ __forceinline__ int bwd_layer_offset(int l) {
    return (int)pack_offset(PB_L7) + (7 - l) * (int)pack_floats(PB_L7) + (l < 5 ? (int)(pack_floats(PB_L5) - pack_floats(PB_L7)) : 0);
}
static_assert(pack_offset(PB_L7) + 1 * pack_floats(PB_L7) == pack_offset(PB_L6) && pack_offset(PB_L7) + 2 * pack_floats(PB_L7) == pack_offset(PB_L5) &&
              pack_offset(PB_L7) + 3 * pack_floats(PB_L7) + (pack_floats(PB_L5) - pack_floats(PB_L7)) == pack_offset(PB_L4) &&
              pack_offset(PB_L7) + 6 * pack_floats(PB_L7) + (pack_floats(PB_L5) - pack_floats(PB_L7)) == pack_offset(PB_L1), "bwd_layer_offset");

// The lane-derived LDS / buffer offsets of a stage are cheap to compute; hoisted out of the layer loop and kept live (the
// compiler's choice) they overflow the register file.  An opaque copy of the lane index pins them to the stage.
__device__ __forceinline__ int stage_local(int lane) {
    asm volatile("" : "+v"(lane));
    return lane;
}

template <int NCT, int PF>
struct WRing { u32x4 q[PF + 1][NCT][2]; };

// issues the first PF k-steps of weight fragments of a block.  Called BEFORE the previous stage's epilogue: vector-memory
// operations retire in order (one counter for loads and stores), so fragments requested after the epilogue's 16 stores
// would not be usable before every one of those stores is acknowledged.
template <int KS, int NCT, int PF>
__device__ __forceinline__ void gemm16_head(const float* __restrict__ wp, int ct0, int lane, WRing<NCT, PF>& r) {
    const WFrag wf(wp, lane);
    const int ct0u = __builtin_amdgcn_readfirstlane(ct0);
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < KS) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                r.q[p][c][0] = wf.load(ct0u + c, p, KS, 0);
                r.q[p][c][1] = wf.load(ct0u + c, p, KS, 1);
            }
        }
}

template <int KS, int NCT, int PF>
__device__ __forceinline__ void gemm16_body(const _Float16* __restrict__ T, const float* __restrict__ wp, int ct0, int lane,
                                            WRing<NCT, PF>& r, f32x16 (&acc)[4][NCT]) {
    lane = stage_local(lane);
    const int row = lane & 31, lh = lane >> 5;
    const int sw = hsw(row);                        // rows row + 32 * rt share the swizzle
    const int rbase = row * LD;
    const WFrag wf(wp, lane);
    const int ct0u = __builtin_amdgcn_readfirstlane(ct0);
    int abase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) abase[j] = rbase + (((2 * j + lh) ^ sw) << 3);
    // ONE set of activation fragments: row tile rt's MFMAs of a k-step are consecutive, and its fragment of the next
    // k-step is requested right behind them (the other 3 row tiles' MFMAs, >= 384 cycles, cover the LDS round trip)
    half8 a[4];
    auto a_off = [&](int ks) { return abase[ks & 3] + ((((2 * ks) & ~7)) << 3); };
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) a[rt] = *reinterpret_cast<const half8*>(T + a_off(0) + rt * 32 * LD);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + PF < KS) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                r.q[(ks + PF) % (PF + 1)][c][0] = wf.load(ct0u + c, ks + PF, KS, 0);
                r.q[(ks + PF) % (PF + 1)][c][1] = wf.load(ct0u + c, ks + PF, KS, 1);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    acc[rt][c] = mfma16(a[rt], __builtin_bit_cast(half8, r.q[ks % (PF + 1)][c][pl]), acc[rt][c]);
            if (ks + 1 < KS) a[rt] = *reinterpret_cast<const half8*>(T + a_off(ks + 1) + rt * 32 * LD);
        }
        __builtin_amdgcn_sched_barrier(0);          // one k-step per scheduling region: keeps the prefetch distances as written
    }
}

// acc[rt][c] += T[rt*32.., 0 .. KS*16) x W(tile ct0 + c), rt = 0..3: 4 x NCT tiles; per block and k-step two MFMAs,
// dY x W_hi and dY x W_lo, into the same accumulator.  Weight fragments PF k-steps ahead (a k-step is 8 * NCT MFMAs),
// activation fragments one k-step ahead.
template <int KS, int NCT, int PF = 2>
__device__ __forceinline__ void gemm16(const _Float16* __restrict__ T, const float* __restrict__ wp, int ct0, int lane,
                                       f32x16 (&acc)[4][NCT]) {
    WRing<NCT, PF> r;
    gemm16_head<KS, NCT, PF>(wp, ct0, lane, r);
    gemm16_body<KS, NCT, PF>(T, wp, ct0, lane, r, acc);
}

// One row tile x NT column tiles: out[t] += T[rt*32.., 0 .. KS*16) x W(tile0 + t).  Only 4 * NT MFMAs per k-step, so the
// weight fragments are fetched PF k-steps ahead (an L2 round trip is several hundred cycles); callers place it where the
// big accumulator set is dead, whose registers then hold the fragment ring.
template <int KS, int NT, int PF = 4>
__device__ __forceinline__ void gemm_row(const _Float16* __restrict__ T, const float* __restrict__ wp, int tile0, int rt, int lane,
                                         f32x16 (&out)[NT]) {
    lane = stage_local(lane);
    const int row = rt * 32 + (lane & 31), lh = lane >> 5;
    const int sw = hsw(row);
    const int rbase = row * LD;
    const WFrag wf(wp, lane);
    const int tu = __builtin_amdgcn_readfirstlane(tile0);
    u32x4 bq[PF + 1][NT][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < KS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bq[p][t][0] = wf.load(tu + t, p, KS, 0);
                bq[p][t][1] = wf.load(tu + t, p, KS, 1);
            }
        }
    half8 an = *reinterpret_cast<const half8*>(T + rbase + ((lh ^ sw) << 3));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const half8 a = an;
        if (ks + PF < KS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bq[(ks + PF) % (PF + 1)][t][0] = wf.load(tu + t, ks + PF, KS, 0);
                bq[(ks + PF) % (PF + 1)][t][1] = wf.load(tu + t, ks + PF, KS, 1);
            }
        }
        if (ks + 1 < KS) an = *reinterpret_cast<const half8*>(T + rbase + ((((ks + 1) * 2 + lh) ^ sw) << 3));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            out[t] = mfma16(a, __builtin_bit_cast(half8, bq[ks % (PF + 1)][t][0]), out[t]);
            out[t] = mfma16(a, __builtin_bit_cast(half8, bq[ks % (PF + 1)][t][1]), out[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NCT>
__device__ __forceinline__ void zero4(f32x16 (&acc)[4][NCT]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
}

// dY = acc masked by the forward pass' ReLU sign bits (bits[h]: the 64-point forward tile of row tiles 2h, 2h+1, in its
// accumulator-layout convention, mlp_common.h) -> the plane (tile scale) and, rescaled by gf = s_call / s_tile (a power
// of two <= 1) and rounded to f16, the SH gradient array `st` of width 256 (m0 = first point of the tile).
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

// bit k (compile-time) of a 64-bit mask word as an all-ones / all-zeros 32-bit mask (one v_bfe_i32)
__device__ __forceinline__ uint32_t bit_mask32(uint64_t w, int k) {
    const uint32_t h = k < 32 ? (uint32_t)w : (uint32_t)(w >> 32);
    uint32_t m;     // asm: the compiler would turn "x & sbfe(...)" back into v_and + v_cmp + v_cndmask
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(h), "n"(k & 31));
    return m;
}

template <bool MASK>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[4][2], const uint64_t (&bits)[2], _Float16* __restrict__ T, int ct0, int lane,
                                         const _Float16* __restrict__ st_tile, float gf, float& amax) {
    lane = stage_local(lane);
    const int lr = lane & 31, r4 = 4 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t st_rsrc = uniform_rsrc(st_tile);     // this tile's 16 blocks of the SH array (64 KiB)
    const _Float16 gh = (_Float16)gf;             // a power of two (or 0 below 2^-24: such a tile's gradients are below f16 anyway)
    const half2v g2 = {gh, gh};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = (ct0 + c) * 32 + lr;
        const int ns = (n >> 3) ^ ((lane >> 5) << 1);
        // plane addresses: 4 swizzle variants x {rows 0-63, rows 64-127}; everything else is an immediate offset (< 64 KiB:
        // with ONE base the offsets of rows >= 103 exceed the ds_write immediate, and the compiler's extra address
        // registers, hoisted out of the layer loop, were spilled)
        _Float16* tb[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tb[q][0] = T + r4 * LD + ((((ns ^ ((q & 1) | ((q >> 1) << 2)))) << 3) | (n & 7));
            tb[q][1] = tb[q][0] + 64 * LD;
        }
        const int st_lane = (((lane >> 5) * 256 + n) * 8) * 2;   // byte offset of unit (block, n); lanes 32-63: the odd block of a pair
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ep = 0; ep < 2; ++ep) {
                uint2 q[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint32_t w[2];
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        float v[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int e = (ep * 2 + h) * 4 + jp * 2 + t;
                            v[t] = acc[rt][c][e];
                            if (MASK) v[t] = __uint_as_float(__float_as_uint(v[t]) & bit_mask32(bits[rt >> 1], (c * 2 + (rt & 1)) * 16 + e));
                        }
                        amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])));   // v_max3_f32
                        const half2v hv = __builtin_convertvector(float2v{v[0], v[1]}, half2v);            // v_cvt_pk_f16_f32 (RNE)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int e = (ep * 2 + h) * 4 + jp * 2 + t;
                            tb[((e >> 1) & 1) | (((e >> 2) & 1) << 1)][rt >> 1][((rt & 1) * 32 + (e & 3) + 8 * (e >> 2)) * LD] = hv[t];
                        }
                        w[jp] = __builtin_bit_cast(uint32_t, hv);
                    }
                    q[h] = uint2{w[0], w[1]};
                }
                // lanes exchange halves, then the exact rescale (v_pk_mul_f16 by a power of two)
                const uint4 u = sh_pair_unit(q[0], q[1]);
                const uint32_t uw[4] = {u.x, u.y, u.z, u.w};
                u32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, uw[i]) * g2);
                // vector offset + zero scalar offset: with a SCALAR offset register the compiler (ROCm 7.2) assumes the
                // 16-byte store's data registers may be overwritten at once, but gfx950 still reads them late - the
                // last lanes of each row then stored whatever the next instruction wrote (caught by the dW parity tests)
                __builtin_amdgcn_raw_buffer_store_b128(o, st_rsrc, st_lane + (rt * 4 + ep * 2) * 256 * 8 * 2, 0, 0);
            }
    }
}

template <int C>
__global__ __launch_bounds__(NTHREADS, 2) void mlp_bwd_f16_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 T[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: weight pointers stay scalar
    const int64_t m0 = (int64_t)blockIdx.x * TMB;
    const int64_t M = a.M;
    const float* acts = a.acts;
    float* dacts = a.dacts;
    const float* packed_h = a.packed + PACKED_FLOATS;
    const int ct0 = wave * 2;
    const int64_t Mp = m_pad(M);
    // ReLU sign-bit words of the two 64-point forward tiles this workgroup covers
    const __amdgpu_buffer_rsrc_t mask_rsrc =
        uniform_rsrc(reinterpret_cast<const uint64_t*>(acts + sact_mask(Mp)) + (int64_t)blockIdx.x * 2 * NTHREADS);
    const int mask_stride_b = (int)((Mp / TM) * NTHREADS * 8);           // bytes between layers (< 2^31 up to 8M points)
    auto load_bits = [&](int layer, uint64_t (&b)[2]) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const int so = __builtin_amdgcn_readfirstlane(layer * mask_stride_b);
        const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(mask_rsrc, tid * 8, so, 0);
        const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(mask_rsrc, tid * 8 + NTHREADS * 8, so, 0);
        b[0] = ((uint64_t)lo[1] << 32) | lo[0];
        b[1] = ((uint64_t)hi[1] << 32) | hi[0];
    };
    _Float16* st_dyh = reinterpret_cast<_Float16*>(dacts + sdact_h(Mp, 0));      // layer l: + l * Mp * 256 halfs
    auto st_tile = [&](int l) { return st_dyh + ((int64_t)l * Mp + m0) * 256; };   // tile's part of layer l's SH array
    float s_g, inv_s_g;
    const float mx_call = a.absmax ? *a.absmax : dacts[sdact_info(Mp) + SD_DRAW];
    if (a.absmax && blockIdx.x == 0 && tid == 0) dacts[sdact_info(Mp) + SD_DRAW] = mx_call;   // the dW reduce reads it there
    pow2_scale6(mx_call, s_g, inv_s_g);                                          // scale of the dY arrays of this call
    if (a.status && blockIdx.x == 0 && tid == 0 && reinterpret_cast<const uint32_t*>(acts + sact_info(Mp))[SI_TAG] != SACT_TAG_SPLIT)
        a.status[2] = 1u;
    TR(0);
#ifdef BENERF_TRACE_DX
    if (tid == 0 && blockIdx.x < 2048) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); reinterpret_cast<unsigned long long*>(a.d_vdir)[blockIdx.x * 64 + 63] = ((unsigned long long)xcc << 32) | hw; }
#endif
    float amax = 0.f;          // max |tile-scaled gradient| of this thread before its f16 rounding (range guard)

    // ---- P0: d_raw tile, its power-of-two scale, scaled values -> scratch floats [28, 28+C] of each row ------
    float dr0[C + 1];
    if (tid < TMB) {
        const int64_t m = m0 + tid;
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c <= C; ++c) {
            dr0[c] = m < M ? a.d_raw[m * (C + 1) + c] : 0.f;
            mx = fmaxf(mx, fabsf(dr0[c]));
        }
        mx = wave_max_nonneg(mx);                                           // lane 63 of waves 0 and 1
        if (lane == 63) *fscr1(T, wave, 26) = mx;
    }
    lds_barrier(); TR(1);
    float s, inv_s;
    pow2_scale6(fmaxf(*fscr1(T, 0, 26), *fscr1(T, 1, 26)), s, inv_s);       // 2^(6 - exponent(max)), exact inverse
    if (tid < TMB) {
#pragma unroll
        for (int c = 0; c <= C; ++c) *fscr1(T, tid, 28 + c) = dr0[c] * s;
    }
    lds_barrier(); TR(2);
    const float gf = s_g * inv_s;   // tile scale -> scale of the stored dY (power of two <= 1)

    // ---- P1: rgb layer backward + ReLU mask of the views layer -> dYv in plane[:, 0:128), accumulator layout ----
    // thread <-> (column wave*32 + lane&31, rows rt*32 + acc_row(e)): the hv sign bits the forward pass saved for
    // its VIEWS accumulators line up with this thread's elements, so hv itself is not read.
    {
        uint64_t hvbits[2];
        load_bits(8, hvbits);
        const int col = wave * 32 + (lane & 31), r4 = 4 * (lane >> 5);
        float wr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) wr[c] = a.w_rgb[c * 128 + col];
        _Float16* st_lane = reinterpret_cast<_Float16*>(dacts + sdact_hv(Mp)) + (((m0 >> 3) + (lane >> 5)) * ACT_HV_W + col) * 8;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ep = 0; ep < 2; ++ep) {
                Quad16 q[2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = (ep * 2 + h) * 4 + j;
                        const int p = rt * 32 + (e & 3) + 8 * (e >> 2) + r4;
                        const float4 dr = *reinterpret_cast<const float4*>(fscr1(T, p, 28));
                        const float drv[4] = {dr.x, dr.y, dr.z, dr.w};
                        float g = 0.f;
#pragma unroll
                        for (int c = 0; c < C; ++c) g += drv[c] * wr[c];
                        const float v = ((hvbits[rt >> 1] >> ((rt & 1) * 16 + e)) & 1ull) ? g : 0.f;
                        T[hidx(p, col)] = (_Float16)v;
                        const float sv = v * gf;
                        amax = fmaxf(amax, fabsf(sv));
                        q[h].v[j] = (_Float16)sv;
                    }
                *reinterpret_cast<uint4*>(st_lane + (int64_t)(rt * 4 + ep * 2) * ACT_HV_W * 8) = sh_pair_unit(q[0], q[1]);
            }
    }
    lds_barrier(); TR(3);

    f32x16 acc[4][2];
    uint64_t bits[2] = {0ull, 0ull};

    // ---- P2: VIEWS^T: dFeat = dYv x Wv[:, :256]; dPE(dir) = dYv x Wv[:, 256:283] --------------------------------
    {   // dPE(dir): tile 8 of the block, row tile = wave -> scratch floats [0,27)
        f32x16 ap[1];
#pragma unroll
        for (int e = 0; e < 16; ++e) ap[0][e] = 0.f;
        gemm_row<8, 1>(T, packed_h + pack_offset(PB_VIEWS), 8, wave, lane, ap);
        const int ln = stage_local(lane);       // addresses of this block only (not shared with the dPE blocks further down)
        if ((ln & 31) < 27) {
#pragma unroll
            for (int e = 0; e < 16; ++e) *fscr1(T, wave * 32 + acc_row(e, ln), ln & 31) = ap[0][e];
        }
        __builtin_amdgcn_sched_barrier(0);      // the block's accumulators are stored before the next GEMM's prologue starts
    }
    zero4(acc);
    gemm16<8, 2>(T, packed_h + pack_offset(PB_VIEWS), ct0, lane, acc);
    lds_barrier(); TR(4);   // dYv fully consumed; dPE(dir) visible
    epilogue<false>(acc, bits, T, ct0, lane, reinterpret_cast<_Float16*>(dacts + sdact_feat(Mp)) + m0 * 256, gf, amax);
    if (tid < TMB && m0 + tid < M) {   // d viewdirs (per point) through PE(dir)
        const int64_t m = m0 + tid;
        const float* ped = acts + sact_ped32(Mp) + m * ACT_PED_W;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float sv = *fscr1(T, tid, d);
            if (a.pe_w) sv *= a.pe_w[64 + d];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int es = 3 + f * 6 + d, ec = es + 3;
                const float sn = ped[es], cs = ped[ec];
                const float ws = a.pe_w ? a.pe_w[64 + es] : 1.f, wc = a.pe_w ? a.pe_w[64 + ec] : 1.f;
                sv += (float)(1 << f) * (cs * (ws * *fscr1(T, tid, es)) - sn * (wc * *fscr1(T, tid, ec)));
            }
#ifndef BENERF_TRACE_DX
            a.d_vdir[m * 3 + d] = sv * inv_s;
#endif
        }
    }
    load_bits(7, bits);
    lds_barrier(); TR(5);

    // ---- P3: FEAT^T (+ alpha head), mask h7 -> dY7 ----------------------------------------------------
    zero4(acc);
    gemm16<16, 2>(T, packed_h + pack_offset(PB_FEAT), ct0, lane, acc);
    {
        const float wa0 = a.w_alpha[ct0 * 32 + (lane & 31)];
        const float wa1 = a.w_alpha[(ct0 + 1) * 32 + (lane & 31)];
        const int r4 = 4 * (lane >> 5);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float ds = *fscr1(T, rt * 32 + (e & 3) + 8 * (e >> 2) + r4, 28 + C);
                acc[rt][0][e] += ds * wa0;
                acc[rt][1][e] += ds * wa1;
            }
    }
    lds_barrier(); TR(6);
    // Loads the next stage needs are requested BEFORE this stage's epilogue stores (in-order retirement, gemm16_head): the
    // sign bits of the stage after (from HBM: a whole epilogue + K-loop of cover) and the first weight fragments.
    uint64_t bits_n[2];
    WRing<2, 2> ring;
    load_bits(6, bits_n);
    gemm16_head<16, 2, 2>(packed_h + pack_offset(PB_L7), ct0, lane, ring);
    epilogue<true>(acc, bits, T, ct0, lane, st_tile(7), gf, amax);
    lds_barrier(); TR(7);

    // ---- P4: L7 .. L1: dY_l x W_l, mask h_{l-1} -> dY_{l-1} ----------------------------------------------
    auto layer = [&](int l) __attribute__((always_inline)) {
        zero4(acc);
        gemm16_body<16, 2, 2>(T, packed_h + bwd_layer_offset(l), ct0, lane, ring, acc);
        lds_barrier(); TR(10 + (7 - l) * 2 + 0);
        bits[0] = bits_n[0];
        bits[1] = bits_n[1];
        if (l >= 2) {
            load_bits(l - 2, bits_n);
            gemm16_head<16, 2, 2>(packed_h + bwd_layer_offset(l - 1), ct0, lane, ring);
        }
        epilogue<true>(acc, bits, T, ct0, lane, st_tile(l - 1), gf, amax);
        lds_barrier(); TR(10 + (7 - l) * 2 + 1);
    };
#pragma unroll 1
    for (int l = 7; l >= 6; --l) layer(l);
    // The layer-5 skip's dPE block [row tile = wave][col tile 0, 1] = dY5 x W5[:, PE part] (tiles 8, 9 of the block)
    // waits for layer 0's part in the plane's scratch columns [256,320) as f16 (tile scale; free from P3 on): its 32
    // registers serve the K-loops in between.  Outside the layer loop, so that the loop carries no state around it.
    {
        f32x16 dpe[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) dpe[c][e] = 0.f;
        gemm_row<16, 2>(T, packed_h + pack_offset(PB_L5), 8, wave, lane, dpe);
        const int ln = stage_local(lane);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                amax = __builtin_fmaxf(amax, __builtin_fabsf(dpe[c][e]));
                T[hidx(wave * 32 + acc_row(e, ln), 256 + c * 32 + (ln & 31))] = (_Float16)dpe[c][e];
            }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int l = 5; l >= 1; --l) layer(l);

    if (a.status) {   // range guard of the stored f16 gradients: one atomic per wave, only near f16's maximum
        const float wmax = wave_max_nonneg(amax);
        if (lane == 63 && !(wmax < 32768.f)) atomicMax(a.status + 1, __float_as_uint(wmax == wmax ? wmax : __builtin_inff()));
    }

    // ---- P5: L0^T: dPE += dY0 x W0 -----------------------------------------------------------------------
    // The tile's saved PE rows (f32 [128][64] = 32 KiB, contiguous: the arrays cover all Mp rows) are requested here as
    // coalesced 16-byte loads, so that their HBM round trip hides under the L0 GEMM; P6 reads them from LDS.
    float4 per[8];
    {
        const float4* pe_tile = reinterpret_cast<const float4*>(acts + sact_pe32(Mp) + m0 * ACT_PE_W);
#pragma unroll
        for (int k = 0; k < 8; ++k) per[k] = pe_tile[tid + k * NTHREADS];
    }
    f32x16 dpe[2];
    {
        const int ln = stage_local(lane);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) dpe[c][e] = (float)T[hidx(wave * 32 + acc_row(e, ln), 256 + c * 32 + (ln & 31))];
    }
    gemm_row<16, 2>(T, packed_h + pack_offset(PB_L0), 0, wave, lane, dpe);
    lds_barrier(); TR(40);      // every wave is done reading dY0: the plane becomes f32 scratch, 133 floats per point:
    // [0,64) dPE, [64,68) the odd-frequency partial sums, [68,132) PE.  The odd row stride keeps P6's per-point walks
    // (lane = point, same column) free of bank conflicts.
    float* F = reinterpret_cast<float*>(T);
    constexpr int FLD = 133;
    static_assert((size_t)TMB * FLD * sizeof(float) <= BWD_SMEM, "P6 scratch fits the plane");
    {
        const int ln = stage_local(lane);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) F[(wave * 32 + acc_row(e, ln)) * FLD + c * 32 + (ln & 31)] = dpe[c][e];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int u = tid + k * NTHREADS;                         // float4 u of the tile: point u / 16, floats 4 (u % 16) ..
        float* dst = F + (u >> 4) * FLD + 68 + (u & 15) * 4;
        dst[0] = per[k].x;
        dst[1] = per[k].y;
        dst[2] = per[k].z;
        dst[3] = per[k].w;
    }
    lds_barrier(); TR(41);

    // ---- P6: dPE -> d_pts through the saved PE values; two threads per point (even / odd frequencies) ------------
    {
        const int pt = tid & (TMB - 1), g = tid >> 7;
        const int64_t m = m0 + pt;
        const float* dp = F + pt * FLD;
        const float* pe = dp + 68;
        float sp[3] = {0.f, 0.f, 0.f};
        if (g == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) sp[d] = a.pe_w ? a.pe_w[d] * dp[d] : dp[d];
        }
        for (int f = g; f < 10; f += 2) {
            const float sc = (float)(1 << f);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int es = 3 + f * 6 + d, ec = es + 3;
                const float sn = pe[es], cs = pe[ec];
                const float ws = a.pe_w ? a.pe_w[es] : 1.f, wc = a.pe_w ? a.pe_w[ec] : 1.f;
                sp[d] += sc * (cs * (ws * dp[es]) - sn * (wc * dp[ec]));
            }
        }
        float* part = F + pt * FLD + 64;                          // floats [64,68) of the row: past the dPE block
        if (g == 1) {
#pragma unroll
            for (int d = 0; d < 3; ++d) part[d] = sp[d];
        }
        lds_barrier(); TR(42);
        if (g == 0 && m < M) {
#pragma unroll
            for (int d = 0; d < 3; ++d) a.d_pts[m * 3 + d] = (sp[d] + part[d]) * inv_s;
        }
    }
}

// max |d_raw| -> dacts info word (zeroed by the launcher; non-negative floats order like their bit patterns)
__global__ void grad_absmax_kernel(const float* __restrict__ d_raw, int64_t n, float* __restrict__ out) {
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mx = fmaxf(mx, fabsf(d_raw[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(mx));
}

}  // namespace

int benerf_mlp_dx_split_launch(const BenerfMlpParams* params, const float* packed, int channels, int64_t M, const float* d_raw,
                               const float* acts, float* dacts, float* d_pts, float* d_vdir_pts, uint32_t* status,
                               const float* d_raw_absmax, hipStream_t stream) {
    BwdArgs a;
    a.d_raw = d_raw;
    a.acts = acts;
    a.dacts = dacts;
    a.packed = packed;
    a.w_alpha = params->w[BENERF_L_ALPHA];
    a.w_rgb = params->w[BENERF_L_RGB];
    a.pe_w = params->pe_weights;
    a.d_pts = d_pts;
    a.d_vdir = d_vdir_pts;
    a.status = status;
    a.absmax = d_raw_absmax;
    a.M = M;
    const int64_t tiles = mlp::m_pad(M) / TMB;
    BENERF_REQUIRE(tiles < (1ll << 31), "mlp_bwd: too many points");
    if (!d_raw_absmax) {    // nobody computed max |d_raw| for us: one pass over d_raw into the info word
        float* info = dacts + mlp::sdact_info(mlp::m_pad(M));
        if (hipMemsetAsync(info, 0, mlp::SD_COUNT * sizeof(float), stream) != hipSuccess) {
            benerf_set_error("mlp_bwd: memset failed");
            return BENERF_EHIP;
        }
        hipLaunchKernelGGL(grad_absmax_kernel, dim3(256), dim3(256), 0, stream, d_raw, M * (channels + 1), info + mlp::SD_DRAW);
    }
    dim3 grid((unsigned)tiles), block(mlp::NTHREADS);
    const int smem = (int)BWD_SMEM;
    static BenerfLdsAttr attr[2];       // once per device and variant
    if (!benerf_lds_attr(attr[channels == 1 ? 0 : 1], channels == 1 ? (const void*)mlp_bwd_f16_kernel<1> : (const void*)mlp_bwd_f16_kernel<3>, smem)) {
        benerf_set_error("mlp_bwd(dx, f16): cannot reserve %d bytes of LDS", smem);
        return BENERF_EHIP;
    }
    if (channels == 1) hipLaunchKernelGGL((mlp_bwd_f16_kernel<1>), grid, block, smem, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_f16_kernel<3>), grid, block, smem, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dx, f16)");
    return BENERF_OK;
}
