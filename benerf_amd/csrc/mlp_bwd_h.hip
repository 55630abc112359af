// K3 backward part 1, split-f16 variant of mlp_bwd.hip: the same activation-gradient chain (phases P0..P6, same
// d_pts / d_viewdirs outputs; the dY arrays in ST layout, mlp_split.h) with every GEMM as three f16 MFMAs on
// hi/lo-split operands.
//
// Gradients are far outside the f16 range (d_raw ~ 1/n_rays), but the whole chain is LINEAR in d_raw: each
// tile multiplies its d_raw by a power of two s = 2^(-4 - exponent(max|d_raw| of the tile)), runs the chain on the
// scaled values (|dY| = O(2^-4 .. 2^6), inside f16's normal range with 2^20 of head room) and multiplies every
// output by 1/s - both exact.  Elements more than 2^-14 below the tile's largest lose relative precision down to
// an absolute floor of 2^-35 of that largest value, far below the f32 rounding of the sums they enter.  The dY arrays
// for the dW kernels are stored with ONE scale per call (s_g from max|d_raw| over the whole launch, computed by a
// small pre-kernel) so that dW can accumulate across tiles; the dW reduce kernel divides by s_g.
//
// LDS: two f16 planes (80 KiB, two workgroups per CU).  The planes' PE columns [256,320) are never a GEMM
// operand here and serve as 64 floats of f32 scratch per point (fscr): scaled d_raw at [60,64), dPE(dir) at
// [0,27) during P2, dPE at [0,64) from layer 5 on.
#include "mlp_split.h"

namespace {
using namespace mlp;

struct BwdArgs {
    const float* d_raw;
    const float* acts;
    float* dacts;
    const float* packed;    // f32 section (PB_VIEWSPE) ; split-f16 section at + PACKED_FLOATS
    const float* w_alpha;   // [256]
    const float* w_rgb;     // [C][128]
    float* d_pts;           // [M][3]
    float* d_vdir;          // [M][3]
    uint32_t* status;       // [1]: max |stored gradient| bits once >= 2^15, [2]: acts buffer written by another mode (may be null)
    int64_t M;
};

// one 32x32 output tile: rows rt*32.., column tile `tile` of the packed block; returns hi*hi + cross * 2^-11
template <int KS>
__device__ __forceinline__ f32x16 gemm_one(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl,
                                           const float* __restrict__ wp, int tile, int rt, int lane) {
    const int row = rt * 32 + (lane & 31), lh = lane >> 5;
    const int sw = hsw(row);
    const int rbase = row * LD;
    const uint4* bp = reinterpret_cast<const uint4*>(wp) + ((int64_t)(tile >> 1) * KS * 4 + (tile & 1) * 2) * 64 + lane;   // mlp_pack.hip layout
    f32x16 a1, a2;
#pragma unroll
    for (int e = 0; e < 16; ++e) a1[e] = a2[e] = 0.f;
    uint4 bhn = bp[0], bln = bp[64];
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
        const half8 bh = __builtin_bit_cast(half8, bhn), bl = __builtin_bit_cast(half8, bln);
        if (ks + 1 < KS) {
            bhn = bp[(ks + 1) * 256];
            bln = bp[(ks + 1) * 256 + 64];
        }
        const int off = rbase + (((ks * 2 + lh) ^ sw) << 3);
        const half8 ah = *reinterpret_cast<const half8*>(Th + off);
        const half8 al = *reinterpret_cast<const half8*>(Tl + off);
        a1 = mfma16(ah, bh, a1);
        a2 = mfma16(ah, bl, a2);
        a2 = mfma16(al, bh, a2);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) a1[e] += a2[e] * LO_INV;
    return a1;
}

// dY = (acc1 + acc2 * 2^-11) masked by the forward pass' ReLU sign bits -> both planes (tile scale) and, rescaled by
// gf = s_call / s_tile (a power of two <= 1) and rounded to f16, the SH gradient array `st` (W = 256; m0 = first point
// of the tile).
template <bool MASK>
__device__ __forceinline__ void epilogue(f32x16 (&acc1)[2][2], f32x16 (&acc2)[2][2], uint64_t bits, _Float16* __restrict__ Th,
                                         _Float16* __restrict__ Tl, int ct0, int lane, _Float16* __restrict__ st, int64_t m0,
                                         float gf, float& amax) {
    const int lr = lane & 31, r4 = 4 * (lane >> 5);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = (ct0 + c) * 32 + lr;
        const int ns = (n >> 3) ^ ((lane >> 5) << 1);
        int base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) base[q] = r4 * LD + ((((ns ^ ((q & 1) | ((q >> 1) << 2)))) << 3) | (n & 7));
        _Float16* st_lane = st + sh_half_index(m0 + r4, 256, n);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int eq = 0; eq < 4; ++eq) {
                Quad16 q;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = eq * 4 + j;
                    float v = acc1[r][c][e] + acc2[r][c][e] * LO_INV;
                    if (MASK) v = ((bits >> ((c * 2 + r) * 16 + e)) & 1ull) ? v : 0.f;
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)((v - (float)hi) * LO_SCALE);
                    const int idx = base[((e >> 1) & 1) | (((e >> 2) & 1) << 1)] + (r * 32 + (e & 3) + 8 * (e >> 2)) * LD;
                    Th[idx] = hi;
                    Tl[idx] = lo;
                    const float sv = v * gf;
                    amax = fmaxf(amax, fabsf(sv));
                    q.v[j] = (_Float16)sv;
                }
                *reinterpret_cast<uint2*>(st_lane + (int64_t)(r * 4 + eq) * 256 * 8) = __builtin_bit_cast(uint2, q);
            }
    }
}

template <int C>
__global__ __launch_bounds__(NTHREADS, 2) void mlp_bwd_split_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 Tsm[];   // Th | Tl
    _Float16* Th = Tsm;
    _Float16* Tl = Tsm + TM * LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: weight pointers stay scalar
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int64_t M = a.M;
    const int pt = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m = m0 + pt;
    const float* acts = a.acts;
    float* dacts = a.dacts;
    const float* packed_h = a.packed + PACKED_FLOATS;
    const int ct0 = wave * 2;
    const int64_t Mp = m_pad(M);
    const uint64_t* mask_in = reinterpret_cast<const uint64_t*>(acts + sact_mask(Mp)) + (int64_t)blockIdx.x * NTHREADS + tid;
    const int64_t mask_stride = (Mp / TM) * NTHREADS;
    _Float16* st_dyh = reinterpret_cast<_Float16*>(dacts + sdact_h(Mp, 0));      // layer l: + l * Mp * 256 halfs
    float s_g, inv_s_g;
    pow2_scale6(dacts[sdact_info(Mp) + SD_DRAW], s_g, inv_s_g);                   // scale of the dY arrays of this call
    if (a.status && blockIdx.x == 0 && tid == 0 && reinterpret_cast<const uint32_t*>(acts + sact_info(Mp))[SI_TAG] != SACT_TAG_SPLIT)
        a.status[2] = 1u;
    float amax = 0.f;          // max |stored gradient| of this thread (range guard)
    const int prow = pt * LD;

    // ---- P0: d_raw tile, its power-of-two scale, scaled values -> scratch floats [60, 60+C] of each row ------
    if (tid < 64) {
        float dr[C + 1];
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c <= C; ++c) {
            dr[c] = m < M ? a.d_raw[m * (C + 1) + c] : 0.f;
            mx = fmaxf(mx, fabsf(dr[c]));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float s, inv;
        pow2_scale6(mx, s, inv);                                            // 2^(6 - exponent(max)), exact inverse
#pragma unroll
        for (int c = 0; c <= C; ++c) *fscr(Th, Tl, pt, 60 + c) = dr[c] * s;
        if (tid == 0) *fscr(Th, Tl, 0, 56) = inv;
    }
    lds_barrier();
    const float inv_s = *fscr(Th, Tl, 0, 56);
    const float gf = s_g * inv_s;   // tile scale -> scale of the stored dY (power of two <= 1)

    // ---- P1: rgb layer backward + ReLU mask of the views layer -> dYv in planes[:,0:128), accumulator layout ----
    // thread <-> (column wave*32 + lane&31, rows r*32 + acc_row(e)): the hv sign bits the forward pass saved for
    // its VIEWS accumulators line up with this thread's elements, so hv itself is not read.
    {
        const uint64_t hvbits = mask_in[8 * mask_stride];
        const int col = wave * 32 + (lane & 31), r4 = 4 * (lane >> 5);
        float wr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) wr[c] = a.w_rgb[c * 128 + col];
        _Float16* st_lane = reinterpret_cast<_Float16*>(dacts + sdact_hv(Mp)) + sh_half_index(m0 + r4, ACT_HV_W, col);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int eq = 0; eq < 4; ++eq) {
                Quad16 q;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = eq * 4 + j;
                    const int p = r * 32 + (e & 3) + 8 * (e >> 2) + r4;
                    const float4 dr = *reinterpret_cast<const float4*>(fscr(Th, Tl, p, 60));
                    const float drv[4] = {dr.x, dr.y, dr.z, dr.w};
                    float g = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) g += drv[c] * wr[c];
                    const float v = ((hvbits >> (r * 16 + e)) & 1ull) ? g : 0.f;
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)((v - (float)hi) * LO_SCALE);
                    const int idx = hidx(p, col);
                    Th[idx] = hi;
                    Tl[idx] = lo;
                    const float sv = v * gf;
                    amax = fmaxf(amax, fabsf(sv));
                    q.v[j] = (_Float16)sv;
                }
                *reinterpret_cast<uint2*>(st_lane + (int64_t)(r * 4 + eq) * ACT_HV_W * 8) = __builtin_bit_cast(uint2, q);
            }
    }
    lds_barrier();

    f32x16 acc1[2][2], acc2[2][2];

    // ---- P2: VIEWS^T: dFeat = dYv x Wv[:, :256]; dPE(dir) = dYv x Wv[:, 256:283] --------------------------------
    zero_acc(acc1);
    zero_acc(acc2);
    gemm_stage_rolled<8, 2>(Th, Tl, 0, packed_h + pack_offset(PB_VIEWS), ct0, lane, acc1, acc2);
    if (wave < 2) {   // dPE(dir) = dYv x Wv[:, 256:283]: tile 8 of the block, one row tile per wave -> scratch floats [0,32)
        const f32x16 ap = gemm_one<8>(Th, Tl, packed_h + pack_offset(PB_VIEWS), 8, wave, lane);
#pragma unroll
        for (int e = 0; e < 16; ++e) *fscr(Th, Tl, wave * 32 + acc_row(e, lane), lane & 31) = ap[e];
    }
    lds_barrier();   // dYv fully consumed; dPE(dir) visible
    epilogue<false>(acc1, acc2, 0ull, Th, Tl, ct0, lane, reinterpret_cast<_Float16*>(dacts + sdact_feat(Mp)), m0, gf, amax);
    if (grp == 0 && m < M) {   // d viewdirs (per point) through PE(dir)
        const float* ped = acts + sact_ped32(Mp) + m * ACT_PED_W;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float s = *fscr(Th, Tl, pt, d);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const float sn = ped[3 + f * 6 + d], cs = ped[3 + f * 6 + 3 + d];
                s += (float)(1 << f) * (cs * *fscr(Th, Tl, pt, 3 + f * 6 + d) - sn * *fscr(Th, Tl, pt, 3 + f * 6 + 3 + d));
            }
            a.d_vdir[m * 3 + d] = s * inv_s;
        }
    }
    uint64_t bits = mask_in[7 * mask_stride];
    lds_barrier();

    // ---- P3: FEAT^T (+ alpha head), mask h7 -> dY7 ----------------------------------------------------
    zero_acc(acc1);
    zero_acc(acc2);
    gemm_stage_rolled<16, 2>(Th, Tl, 0, packed_h + pack_offset(PB_FEAT), ct0, lane, acc1, acc2);
    {
        const float wa0 = a.w_alpha[ct0 * 32 + (lane & 31)];
        const float wa1 = a.w_alpha[(ct0 + 1) * 32 + (lane & 31)];
        const int r4 = 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float ds = *fscr(Th, Tl, r * 32 + (e & 3) + 8 * (e >> 2) + r4, 60 + C);
                acc1[r][0][e] += ds * wa0;
                acc1[r][1][e] += ds * wa1;
            }
    }
    lds_barrier();
    epilogue<true>(acc1, acc2, bits, Th, Tl, ct0, lane, st_dyh + 7 * Mp * 256, m0, gf, amax);
    lds_barrier();

    // ---- P4: L7 .. L1: dY_l x W_l, mask h_{l-1} -> dY_{l-1} ----------------------------------------------
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        bits = mask_in[(l - 1) * mask_stride];
        zero_acc(acc1);
        zero_acc(acc2);
        const int pid = PB_L7 + (7 - l);
        gemm_stage<16, 2>(Th, Tl, 0, packed_h + pack_offset(pid), ct0, lane, acc1, acc2);
        if (l == 5) {   // skip connection: dPE = dY5 x W5[:, PE part] (tiles 8, 9 of the block) -> scratch [0,64)
            const f32x16 ap = gemm_one<16>(Th, Tl, packed_h + pack_offset(PB_L5), 8 + (wave & 1), wave >> 1, lane);
            const int col = (wave & 1) * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) *fscr(Th, Tl, (wave >> 1) * 32 + acc_row(e, lane), col) = ap[e];
        }
        lds_barrier();
        epilogue<true>(acc1, acc2, bits, Th, Tl, ct0, lane, st_dyh + (int64_t)(l - 1) * Mp * 256, m0, gf, amax);
        lds_barrier();
    }

    if (a.status) {   // range guard of the stored f16 gradients: one atomic per wave, only near f16's maximum
        const float wmax = wave_max_nonneg(amax);
        if (lane == 63 && !(wmax < 32768.f)) atomicMax(a.status + 1, __float_as_uint(wmax == wmax ? wmax : __builtin_inff()));
    }

    // ---- P5: L0^T: dPE += dY0 x W0 -----------------------------------------------------------------------
    {
        const f32x16 ap = gemm_one<16>(Th, Tl, packed_h + pack_offset(PB_L0), wave & 1, wave >> 1, lane);
        const int col = (wave & 1) * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) *fscr(Th, Tl, (wave >> 1) * 32 + acc_row(e, lane), col) += ap[e];
    }
    lds_barrier();

    // ---- P6: dPE -> d_pts through the saved PE values; group partials as f32 in the (dead) hi-plane columns ----
    float* part = reinterpret_cast<float*>(Th + prow);      // 4 groups x 4 floats = first 64 bytes of the row
    {
        float s[3] = {0.f, 0.f, 0.f};
        const int64_t mc = m < M ? m : M - 1;
        const float* pe = acts + sact_pe32(Mp) + mc * ACT_PE_W;
        if (grp == 0) {
            s[0] = *fscr(Th, Tl, pt, 0);
            s[1] = *fscr(Th, Tl, pt, 1);
            s[2] = *fscr(Th, Tl, pt, 2);
        }
        for (int f = grp; f < 10; f += 4) {
            const float sc = (float)(1 << f);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float sn = pe[3 + f * 6 + d], cs = pe[3 + f * 6 + 3 + d];
                s[d] += sc * (cs * *fscr(Th, Tl, pt, 3 + f * 6 + d) - sn * *fscr(Th, Tl, pt, 3 + f * 6 + 3 + d));
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) part[grp * 4 + d] = s[d];
    }
    lds_barrier();
    if (tid < 64 && m < M) {
#pragma unroll
        for (int d = 0; d < 3; ++d) a.d_pts[m * 3 + d] = ((part[d] + part[4 + d]) + (part[8 + d] + part[12 + d])) * inv_s;
    }
}

// max |d_raw| -> dacts[sdact_scale] (zeroed by the launcher; non-negative floats order like their bit patterns)
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
    a.d_pts = d_pts;
    a.d_vdir = d_vdir_pts;
    a.status = status;
    a.M = M;
    const int64_t tiles = mlp::sn_tiles(M);
    BENERF_REQUIRE(tiles < (1ll << 31), "mlp_bwd: too many points");
    float* info = dacts + mlp::sdact_info(mlp::m_pad(M));
    if (hipMemsetAsync(info, 0, mlp::SD_COUNT * sizeof(float), stream) != hipSuccess) {
        benerf_set_error("mlp_bwd: memset failed");
        return BENERF_EHIP;
    }
    hipLaunchKernelGGL(grad_absmax_kernel, dim3(256), dim3(256), 0, stream, d_raw, M * (channels + 1), info + mlp::SD_DRAW);
    dim3 grid((unsigned)tiles), block(mlp::NTHREADS);
    const int smem = (int)mlp::TILE_SMEM;
    const void* fn = channels == 1 ? (const void*)mlp_bwd_split_kernel<1> : (const void*)mlp_bwd_split_kernel<3>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        benerf_set_error("mlp_bwd(dx, split): cannot reserve %d bytes of LDS", smem);
        return BENERF_EHIP;
    }
    if (channels == 1) hipLaunchKernelGGL((mlp_bwd_split_kernel<1>), grid, block, smem, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_split_kernel<3>), grid, block, smem, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dx, split)");
    return BENERF_OK;
}
