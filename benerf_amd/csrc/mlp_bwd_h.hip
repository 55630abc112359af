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
    uint32_t* status;       // [1]: max |stored gradient| bits once >= 2^15, [2]: acts buffer written by another mode (may be null)
    int64_t M;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// f32 scratch float i (0..31) of `row`: slot 32 + i/4 of the plane, swizzled like everything else
__device__ __forceinline__ float* fscr1(_Float16* T, int row, int i) {
    return reinterpret_cast<float*>(T + row * LD + (((32 + (i >> 2)) ^ hsw(row)) << 3)) + (i & 3);
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
        const int soff = (((t_uniform >> 1) * KS + ks) * 2 + (t_uniform & 1)) * 2048 + plane * 1024;
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    }
};

// acc[rt][c] += T[rt*32.., 0 .. KS*16) x W(tile ct0 + c), rt = 0..3: 4 x NCT tiles; per block and k-step two MFMAs,
// dY x W_hi and dY x W_lo, into the same accumulator.  Weight fragments PF k-steps ahead (a k-step is 8 * NCT MFMAs),
// activation fragments one k-step ahead.
template <int KS, int NCT, int PF = 2>
__device__ __forceinline__ void gemm16(const _Float16* __restrict__ T, const float* __restrict__ wp, int ct0, int lane,
                                       f32x16 (&acc)[4][NCT]) {
    const int row = lane & 31, lh = lane >> 5;
    const int sw = hsw(row);                        // rows row + 32 * rt share the swizzle
    const int rbase = row * LD;
    const WFrag wf(wp, lane);
    const int ct0u = __builtin_amdgcn_readfirstlane(ct0);
    u32x4 bq[PF + 1][NCT][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < KS) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                bq[p][c][0] = wf.load(ct0u + c, p, KS, 0);
                bq[p][c][1] = wf.load(ct0u + c, p, KS, 1);
            }
        }
    int abase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) abase[j] = rbase + (((2 * j + lh) ^ sw) << 3);
    half8 an[4];
    auto load_a = [&](int ks) {
        const int off = abase[ks & 3] + ((((2 * ks) & ~7)) << 3);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) an[rt] = *reinterpret_cast<const half8*>(T + off + rt * 32 * LD);
    };
    load_a(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        half8 a[4] = {an[0], an[1], an[2], an[3]};
        if (ks + PF < KS) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                bq[(ks + PF) % (PF + 1)][c][0] = wf.load(ct0u + c, ks + PF, KS, 0);
                bq[(ks + PF) % (PF + 1)][c][1] = wf.load(ct0u + c, ks + PF, KS, 1);
            }
        }
        if (ks + 1 < KS) load_a(ks + 1);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const half8 b = __builtin_bit_cast(half8, bq[ks % (PF + 1)][c][pl]);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt][c] = mfma16(a[rt], b, acc[rt][c]);
            }
        __builtin_amdgcn_sched_barrier(0);          // one k-step per scheduling region: keeps the prefetch distances as written
    }
}

// one 32x32 output tile: acc += T[rt*32.., 0 .. KS*16) x W(tile)
template <int KS>
__device__ __forceinline__ void gemm_one(const _Float16* __restrict__ T, const float* __restrict__ wp, int tile, int rt, int lane,
                                         f32x16& acc) {
    const int row = rt * 32 + (lane & 31), lh = lane >> 5;
    const int sw = hsw(row);
    const int rbase = row * LD;
    const WFrag wf(wp, lane);
    const int tu = __builtin_amdgcn_readfirstlane(tile);
    u32x4 bn = wf.load(tu, 0, KS, 0), ln = wf.load(tu, 0, KS, 1);
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
        const half8 b = __builtin_bit_cast(half8, bn), bl = __builtin_bit_cast(half8, ln);
        if (ks + 1 < KS) {
            bn = wf.load(tu, ks + 1, KS, 0);
            ln = wf.load(tu, ks + 1, KS, 1);
        }
        const half8 a = *reinterpret_cast<const half8*>(T + rbase + (((ks * 2 + lh) ^ sw) << 3));
        acc = mfma16(a, b, acc);
        acc = mfma16(a, bl, acc);
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
template <bool MASK>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[4][2], const uint64_t (&bits)[2], _Float16* __restrict__ T, int ct0, int lane,
                                         _Float16* __restrict__ st, int64_t m0, float gf, float& amax) {
    const int lr = lane & 31, r4 = 4 * (lane >> 5);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = (ct0 + c) * 32 + lr;
        const int ns = (n >> 3) ^ ((lane >> 5) << 1);
        int base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) base[q] = r4 * LD + ((((ns ^ ((q & 1) | ((q >> 1) << 2)))) << 3) | (n & 7));
        _Float16* st_lane = st + (((m0 >> 3) + (lane >> 5)) * 256 + n) * 8;   // unit (block, n); lanes 32-63: the odd block of a pair
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
                        float v = acc[rt][c][e];
                        if (MASK) v = ((bits[rt >> 1] >> ((c * 2 + (rt & 1)) * 16 + e)) & 1ull) ? v : 0.f;
                        const int idx = base[((e >> 1) & 1) | (((e >> 2) & 1) << 1)] + (rt * 32 + (e & 3) + 8 * (e >> 2)) * LD;
                        T[idx] = (_Float16)v;
                        const float sv = v * gf;
                        amax = fmaxf(amax, fabsf(sv));
                        q[h].v[j] = (_Float16)sv;
                    }
                *reinterpret_cast<uint4*>(st_lane + (int64_t)(rt * 4 + ep * 2) * 256 * 8) = sh_pair_unit(q[0], q[1]);
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
    const uint64_t* mask_in = reinterpret_cast<const uint64_t*>(acts + sact_mask(Mp)) + (int64_t)blockIdx.x * 2 * NTHREADS + tid;
    const int64_t mask_stride = (Mp / TM) * NTHREADS;
    _Float16* st_dyh = reinterpret_cast<_Float16*>(dacts + sdact_h(Mp, 0));      // layer l: + l * Mp * 256 halfs
    float s_g, inv_s_g;
    pow2_scale6(dacts[sdact_info(Mp) + SD_DRAW], s_g, inv_s_g);                   // scale of the dY arrays of this call
    if (a.status && blockIdx.x == 0 && tid == 0 && reinterpret_cast<const uint32_t*>(acts + sact_info(Mp))[SI_TAG] != SACT_TAG_SPLIT)
        a.status[2] = 1u;
    float amax = 0.f;          // max |stored gradient| of this thread (range guard)

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
    lds_barrier();
    float s, inv_s;
    pow2_scale6(fmaxf(*fscr1(T, 0, 26), *fscr1(T, 1, 26)), s, inv_s);       // 2^(6 - exponent(max)), exact inverse
    if (tid < TMB) {
#pragma unroll
        for (int c = 0; c <= C; ++c) *fscr1(T, tid, 28 + c) = dr0[c] * s;
    }
    lds_barrier();
    const float gf = s_g * inv_s;   // tile scale -> scale of the stored dY (power of two <= 1)

    // ---- P1: rgb layer backward + ReLU mask of the views layer -> dYv in plane[:, 0:128), accumulator layout ----
    // thread <-> (column wave*32 + lane&31, rows rt*32 + acc_row(e)): the hv sign bits the forward pass saved for
    // its VIEWS accumulators line up with this thread's elements, so hv itself is not read.
    {
        const uint64_t hvbits[2] = {mask_in[8 * mask_stride], mask_in[8 * mask_stride + NTHREADS]};
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
    lds_barrier();

    f32x16 acc[4][2];
    uint64_t bits[2] = {0ull, 0ull};

    // ---- P2: VIEWS^T: dFeat = dYv x Wv[:, :256]; dPE(dir) = dYv x Wv[:, 256:283] --------------------------------
    zero4(acc);
    gemm16<8, 2>(T, packed_h + pack_offset(PB_VIEWS), ct0, lane, acc);
    {   // dPE(dir): tile 8 of the block, row tile = wave -> scratch floats [0,27)
        f32x16 ap;
#pragma unroll
        for (int e = 0; e < 16; ++e) ap[e] = 0.f;
        gemm_one<8>(T, packed_h + pack_offset(PB_VIEWS), 8, wave, lane, ap);
        if ((lane & 31) < 27) {
#pragma unroll
            for (int e = 0; e < 16; ++e) *fscr1(T, wave * 32 + acc_row(e, lane), lane & 31) = ap[e];
        }
    }
    lds_barrier();   // dYv fully consumed; dPE(dir) visible
    epilogue<false>(acc, bits, T, ct0, lane, reinterpret_cast<_Float16*>(dacts + sdact_feat(Mp)), m0, gf, amax);
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
            a.d_vdir[m * 3 + d] = sv * inv_s;
        }
    }
    bits[0] = mask_in[7 * mask_stride];
    bits[1] = mask_in[7 * mask_stride + NTHREADS];
    lds_barrier();

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
    lds_barrier();
    epilogue<true>(acc, bits, T, ct0, lane, st_dyh + 7 * Mp * 256, m0, gf, amax);
    lds_barrier();

    // ---- P4: L7 .. L1: dY_l x W_l, mask h_{l-1} -> dY_{l-1} ----------------------------------------------
    f32x16 dpe[2];          // dPE block [row tile = wave][col tile 0, 1]: layer-5 skip + layer 0, kept in registers
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) dpe[c][e] = 0.f;
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        bits[0] = mask_in[(l - 1) * mask_stride];
        bits[1] = mask_in[(l - 1) * mask_stride + NTHREADS];
        zero4(acc);
        const int pid = PB_L7 + (7 - l);
        gemm16<16, 2>(T, packed_h + pack_offset(pid), ct0, lane, acc);
        if (l == 5) {   // skip connection: dPE = dY5 x W5[:, PE part] (tiles 8, 9 of the block)
            gemm_one<16>(T, packed_h + pack_offset(PB_L5), 8, wave, lane, dpe[0]);
            gemm_one<16>(T, packed_h + pack_offset(PB_L5), 9, wave, lane, dpe[1]);
        }
        lds_barrier();
        epilogue<true>(acc, bits, T, ct0, lane, st_dyh + (int64_t)(l - 1) * Mp * 256, m0, gf, amax);
        lds_barrier();
    }

    if (a.status) {   // range guard of the stored f16 gradients: one atomic per wave, only near f16's maximum
        const float wmax = wave_max_nonneg(amax);
        if (lane == 63 && !(wmax < 32768.f)) atomicMax(a.status + 1, __float_as_uint(wmax == wmax ? wmax : __builtin_inff()));
    }

    // ---- P5: L0^T: dPE += dY0 x W0 -----------------------------------------------------------------------
    gemm_one<16>(T, packed_h + pack_offset(PB_L0), 0, wave, lane, dpe[0]);
    gemm_one<16>(T, packed_h + pack_offset(PB_L0), 1, wave, lane, dpe[1]);
    lds_barrier();      // every wave is done reading dY0: the plane becomes f32 scratch [128][64] (row stride LD halfs)
    float* F = reinterpret_cast<float*>(T);
    constexpr int FLD = LD / 2;                                   // row stride in floats
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) F[(wave * 32 + acc_row(e, lane)) * FLD + c * 32 + (lane & 31)] = dpe[c][e];
    lds_barrier();

    // ---- P6: dPE -> d_pts through the saved PE values; two threads per point (even / odd frequencies) ------------
    {
        const int pt = tid & (TMB - 1), g = tid >> 7;
        const int64_t m = m0 + pt;
        const int64_t mc = m < M ? m : M - 1;
        const float* pe = acts + sact_pe32(Mp) + mc * ACT_PE_W;
        const float* dp = F + pt * FLD;
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
        lds_barrier();
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
                               const float* acts, float* dacts, float* d_pts, float* d_vdir_pts, uint32_t* status, hipStream_t stream) {
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
    a.M = M;
    const int64_t tiles = mlp::m_pad(M) / TMB;
    BENERF_REQUIRE(tiles < (1ll << 31), "mlp_bwd: too many points");
    float* info = dacts + mlp::sdact_info(mlp::m_pad(M));
    if (hipMemsetAsync(info, 0, mlp::SD_COUNT * sizeof(float), stream) != hipSuccess) {
        benerf_set_error("mlp_bwd: memset failed");
        return BENERF_EHIP;
    }
    hipLaunchKernelGGL(grad_absmax_kernel, dim3(256), dim3(256), 0, stream, d_raw, M * (channels + 1), info + mlp::SD_DRAW);
    dim3 grid((unsigned)tiles), block(mlp::NTHREADS);
    const int smem = (int)BWD_SMEM;
    const void* fn = channels == 1 ? (const void*)mlp_bwd_f16_kernel<1> : (const void*)mlp_bwd_f16_kernel<3>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        benerf_set_error("mlp_bwd(dx, f16): cannot reserve %d bytes of LDS", smem);
        return BENERF_EHIP;
    }
    if (channels == 1) hipLaunchKernelGGL((mlp_bwd_f16_kernel<1>), grid, block, smem, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_f16_kernel<3>), grid, block, smem, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dx, f16)");
    return BENERF_OK;
}
