// K3 backward, part 1: activation-gradient chain for one tile of 64 sample points per
// workgroup, exact-f32 MFMA.  Walks NeRF.forward (model/nerf.py:67-116) in reverse:
//
//   d_raw -> rgb layer (VALU) -> dYv = dHV * [hv>0]                       -> DYV
//   VIEWS^T : dYv (128)  x Wv[:, :256]   -> dFeat ; dYv x Wv[:,256:283] (VALU) -> dPE(dir)
//   FEAT^T  : dFeat      x Wf  + d_sigma * w_alpha, * [h7>0]              -> DYH[7]
//   L7..L1  : dY_l       x W_l (h part), * [h_{l-1}>0]                    -> DYH[l-1]
//             (L5 additionally emits dPE = dY_5 x W5[:, :63])
//   L0^T    : dY_0       x W_0                                            -> dPE +=
//   dPE -> d_pts, dPE(dir) -> d_viewdirs (per point) through the sin/cos derivatives, using
//   the PE values saved by the forward pass (no sincos recomputation).
//
// The dY tiles are streamed to HBM for part 2 (mlp_dw.hip: dW = dY^T X).  Same tiling, LDS tile
// and swizzle as the forward kernel (mlp_common.h): the gradient tile lives in T, each wave owns
// 64 columns, B operands (transposed-packed weights) stream from L2; ReLU masks are the sign-bit
// words the forward pass wrote in accumulator layout (one 8-byte load per lane per stage).  All
// small per-point scratch lives in the dead PE columns, so the tile is exactly 80 KiB and two
// workgroups share a CU (one's MFMAs cover the other's epilogue / store phase).
#include "mlp_common.h"

namespace {
using namespace mlp;

struct BwdArgs {
    const float* d_raw;
    const float* acts;
    float* dacts;
    const float* packed;
    const float* w_alpha;   // [256]
    const float* w_rgb;     // [C][128]
    const float* pe_w;      // BARF c2f column weights (include/benerf_hip.h) or null
    float* d_pts;           // [M][3]
    float* d_vdir;          // [M][3]
    int64_t M;
};

template <int KB, int NCT>
__device__ __forceinline__ void gemm_stage(const float* __restrict__ T, int kcol0, const float* __restrict__ wp,
                                           int ct0, int lane, f32x16 (&acc)[2][NCT]) {
    const int row = lane & 31;
    const int sw = swz(row);
    const float* a0p = T + row * LD + kcol0;
    const float* a1p = a0p + 32 * LD;
    const int cl = 4 * (lane >> 5);
    const float4* bp[NCT];
    float4 bn[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        bp[c] = reinterpret_cast<const float4*>(wp) + (int64_t)(ct0 + c) * KB * 64 + lane;
        bn[c] = bp[c][0];
    }
#pragma unroll 2
    for (int kb = 0; kb < KB; ++kb) {
        float4 b[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) b[c] = bn[c];
        if (kb + 1 < KB) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) bn[c] = bp[c][(kb + 1) * 64];
        }
        const int col = (kb * 8 + cl) ^ sw;
        const float4 a0 = *reinterpret_cast<const float4*>(a0p + col);
        const float4 a1 = *reinterpret_cast<const float4*>(a1p + col);
        const float a0v[4] = {a0.x, a0.y, a0.z, a0.w};
        const float a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float bv = i == 0 ? b[c].x : i == 1 ? b[c].y : i == 2 ? b[c].z : b[c].w;
                acc[0][c] = mfma32(a0v[i], bv, acc[0][c]);
                acc[1][c] = mfma32(a1v[i], bv, acc[1][c]);
            }
        }
    }
}

// one 32x32 output tile: rows rt*32.., column tile `tile` of the packed block
template <int KB>
__device__ __forceinline__ void gemm_one(const float* __restrict__ T, int kcol0, const float* __restrict__ wp, int tile,
                                         int rt, int lane, f32x16& acc) {
    const int row = rt * 32 + (lane & 31);
    const int sw = swz(row);
    const float* ap = T + row * LD + kcol0;
    const int cl = 4 * (lane >> 5);
    const float4* bp = reinterpret_cast<const float4*>(wp) + (int64_t)tile * KB * 64 + lane;
    float4 bn = bp[0];
#pragma unroll 4
    for (int kb = 0; kb < KB; ++kb) {
        const float4 b = bn;
        if (kb + 1 < KB) bn = bp[(kb + 1) * 64];
        const float4 a = *reinterpret_cast<const float4*>(ap + ((kb * 8 + cl) ^ sw));
        acc = mfma32(a.x, b.x, acc);
        acc = mfma32(a.y, b.y, acc);
        acc = mfma32(a.z, b.z, acc);
        acc = mfma32(a.w, b.w, acc);
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

// dY = acc masked by the forward pass' ReLU sign bits (bit ((c*2+r)*16+e), see mlp_common.h)
// -> LDS tile + DY array whose tile starts at `dy_tile` (row m0).  `rows_valid` is block-uniform
// and < 64 only for the ragged last tile.
template <bool MASK>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[2][2], uint64_t bits, float* __restrict__ T, int ct0, int lane,
                                         float* __restrict__ dy_tile, int rows_valid) {
    const int lr = lane & 31, r4 = 4 * (lane >> 5);
    float* dy_lane = dy_tile + (int64_t)r4 * 256 + lr;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[r][c][e];
                if (MASK) v = ((bits >> ((c * 2 + r) * 16 + e)) & 1ull) ? v : 0.f;
                acc[r][c][e] = v;
                T[tidx(r * 32 + (e & 3) + 8 * (e >> 2) + r4, (ct0 + c) * 32 + lr)] = v;
            }
    if (rows_valid >= TM) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    dy_lane[(r * 32 + (e & 3) + 8 * (e >> 2)) * 256 + (ct0 + c) * 32] = acc[r][c][e];
    } else {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rowoff = r * 32 + (e & 3) + 8 * (e >> 2);
                    if (rowoff + r4 < rows_valid) dy_lane[rowoff * 256 + (ct0 + c) * 32] = acc[r][c][e];
                }
    }
}

template <int C>
__global__ __launch_bounds__(NTHREADS, 2) void mlp_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float T[];   // [TM][LD], swizzled

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int64_t M = a.M;
    const int pt = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar) thread-group id
    const int64_t m = m0 + pt;
    const float* acts = a.acts;
    float* dacts = a.dacts;
    const int ct0 = wave * 2;
    const int rows_valid = (int)(M - m0 < TM ? M - m0 : TM);   // block-uniform
    const uint64_t* mask_in = reinterpret_cast<const uint64_t*>(acts + act_mask(M)) + (int64_t)blockIdx.x * NTHREADS + tid;
    const int64_t mask_stride = n_tiles(M) * NTHREADS;           // per layer
    float* dyh_tile = dacts + dact_h(M, 0) + m0 * 256;           // layer l: + l * M * 256
    float* trow = T + pt * LD;
    const int psw = swz(pt);

    // ---- P0: d_raw tile -> scratch columns [288, 288+C] of each point's row --------------------------
    if (tid < 64) {
#pragma unroll
        for (int c = 0; c <= C; ++c) trow[(COL_SCR + c) ^ psw] = m < M ? a.d_raw[m * (C + 1) + c] : 0.f;
    }
    lds_barrier();

    // ---- P1: rgb layer backward + ReLU mask of the views layer -> dYv in T[:,0:128) ------------------
    {
        const int j = tid & 127, half = tid >> 7;
        float wr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) wr[c] = a.w_rgb[c * 128 + j];
        const float* hv = acts + act_hv(M) + (m0 + half * 32) * ACT_HV_W + j;
        float* dyv = dacts + dact_hv(M) + (m0 + half * 32) * ACT_HV_W + j;
        // all 32 saved activations of this thread's column in flight at once (one HBM round trip, not 8)
        float hvv[32];
        if (rows_valid >= TM) {
#pragma unroll
            for (int q = 0; q < 32; ++q) hvv[q] = hv[q * ACT_HV_W];
        } else {
#pragma unroll
            for (int q = 0; q < 32; ++q) hvv[q] = half * 32 + q < rows_valid ? hv[q * ACT_HV_W] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int p = half * 32 + q;
            float g = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) g += T[tidx(p, COL_SCR + c)] * wr[c];
            const float v = hvv[q] > 0.f ? g : 0.f;      // rows beyond the tail have hv = 0 and d_raw = 0
            if (rows_valid >= TM || p < rows_valid) dyv[q * ACT_HV_W] = v;
            T[tidx(p, j)] = v;
        }
    }
    lds_barrier();

    f32x16 acc[2][2];

    // ---- P2: VIEWS^T: dFeat = dYv x Wv[:, :256]; dPE(dir) = dYv x Wv[:, 256:283] (VALU) ------------------
    zero_acc(acc);
    gemm_stage<16, 2>(T, 0, a.packed + pack_offset(PB_VIEWS), ct0, lane, acc);
    {
        // this wave's 7 PE(dir) columns j = grp + 4q; weights [grp][n][8] are wave-uniform (scalar loads)
        const float* wq = a.packed + pack_offset(PB_VIEWSPE) + (int64_t)grp * 128 * 8;
        float s[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) s[q] = 0.f;
#pragma unroll 2
        for (int n = 0; n < 128; n += 4) {
            const float4 h4 = *reinterpret_cast<const float4*>(trow + (n ^ psw));
            const float hv4[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 7; ++q) s[q] += hv4[i] * wq[(n + i) * 8 + q];
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int j = grp + 4 * q;
            if (j < 27) trow[(COL_PE + j) ^ psw] = s[q];
        }
    }
    lds_barrier();   // dYv fully consumed; dPE(dir) visible
    epilogue<false>(acc, 0ull, T, ct0, lane, dacts + dact_feat(M) + m0 * 256, rows_valid);
    if (grp == 0 && m < M) {   // d viewdirs (per point) through PE(dir)
        const float* ped = acts + act_ped(M) + m * ACT_PED_W;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float s = trow[(COL_PE + d) ^ psw];
            if (a.pe_w) s *= a.pe_w[64 + d];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int es = 3 + f * 6 + d, ec = es + 3;
                const float sn = ped[es], cs = ped[ec];
                const float ws = a.pe_w ? a.pe_w[64 + es] : 1.f, wc = a.pe_w ? a.pe_w[64 + ec] : 1.f;
                s += (float)(1 << f) * (cs * (ws * trow[(COL_PE + es) ^ psw]) - sn * (wc * trow[(COL_PE + ec) ^ psw]));
            }
            a.d_vdir[m * 3 + d] = s;
        }
    }
    uint64_t bits = mask_in[7 * mask_stride];
    lds_barrier();

    // ---- P3: FEAT^T (+ alpha head), mask h7 -> dY7 ----------------------------------------------------
    zero_acc(acc);
    gemm_stage<32, 2>(T, 0, a.packed + pack_offset(PB_FEAT), ct0, lane, acc);
    {
        const float wa0 = a.w_alpha[ct0 * 32 + (lane & 31)];
        const float wa1 = a.w_alpha[(ct0 + 1) * 32 + (lane & 31)];
        const int r4 = 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float ds = T[tidx(r * 32 + (e & 3) + 8 * (e >> 2) + r4, COL_SCR + C)];
                acc[r][0][e] += ds * wa0;
                acc[r][1][e] += ds * wa1;
            }
    }
    lds_barrier();
    epilogue<true>(acc, bits, T, ct0, lane, dyh_tile + 7 * M * 256, rows_valid);
    lds_barrier();

    // ---- P4: L7 .. L1: dY_l x W_l, mask h_{l-1} -> dY_{l-1} ----------------------------------------------
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        bits = mask_in[(l - 1) * mask_stride];
        zero_acc(acc);
        const int pid = PB_L7 + (7 - l);
        gemm_stage<32, 2>(T, 0, a.packed + pack_offset(pid), ct0, lane, acc);
        if (l == 5) {   // skip connection: dPE = dY5 x W5[:, PE part] (tiles 8, 9 of the block) -> T[:,256:320)
            f32x16 ap;
#pragma unroll
            for (int e = 0; e < 16; ++e) ap[e] = 0.f;
            gemm_one<32>(T, 0, a.packed + pack_offset(PB_L5), 8 + (wave & 1), wave >> 1, lane, ap);
            const int col = COL_PE + (wave & 1) * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) T[tidx((wave >> 1) * 32 + acc_row(e, lane), col)] = ap[e];
        }
        lds_barrier();
        epilogue<true>(acc, bits, T, ct0, lane, dyh_tile + (int64_t)(l - 1) * M * 256, rows_valid);
        lds_barrier();
    }

    // ---- P5: L0^T: dPE += dY0 x W0 -----------------------------------------------------------------------
    {
        f32x16 ap;
#pragma unroll
        for (int e = 0; e < 16; ++e) ap[e] = 0.f;
        gemm_one<32>(T, 0, a.packed + pack_offset(PB_L0), wave & 1, wave >> 1, lane, ap);
        const int col = COL_PE + (wave & 1) * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) T[tidx((wave >> 1) * 32 + acc_row(e, lane), col)] += ap[e];
    }
    lds_barrier();

    // ---- P6: dPE -> d_pts through the saved PE values; group partials in columns [0,16) of the row --------
    {
        float s[3] = {0.f, 0.f, 0.f};
        const int64_t mc = m < M ? m : M - 1;
        const float* pe = acts + act_pe(M) + mc * ACT_PE_W;
        if (grp == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) s[d] = (a.pe_w ? a.pe_w[d] : 1.f) * trow[(COL_PE + d) ^ psw];
        }
        for (int f = grp; f < 10; f += 4) {
            const float sc = (float)(1 << f);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int es = 3 + f * 6 + d, ec = es + 3;
                const float sn = pe[es], cs = pe[ec];
                const float ws = a.pe_w ? a.pe_w[es] : 1.f, wc = a.pe_w ? a.pe_w[ec] : 1.f;
                s[d] += sc * (cs * (ws * trow[(COL_PE + es) ^ psw]) - sn * (wc * trow[(COL_PE + ec) ^ psw]));
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) trow[(grp * 4 + d) ^ psw] = s[d];   // columns < 256 are dead (dY0 consumed)
    }
    lds_barrier();
    if (tid < 64 && m < M) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
            a.d_pts[m * 3 + d] = (trow[(0 + d) ^ psw] + trow[(4 + d) ^ psw]) + (trow[(8 + d) ^ psw] + trow[(12 + d) ^ psw]);
    }
}

}  // namespace

// part 2 (mlp_dw.hip)
int benerf_mlp_dw_launch(const BenerfMlpParams* params, int precision, int channels, int64_t M, const float* d_raw, const float* acts,
                         const float* dacts, float* dw_ws, const BenerfMlpGrads* grads, int accumulate, const float* pe_weights,
                         hipStream_t stream);

// f16 variant (mlp_bwd_h.hip)
int benerf_mlp_dx_split_launch(const BenerfMlpParams* params, const float* packed, int channels, int64_t M, const float* d_raw,
                               const float* acts, float* dacts, float* d_pts, float* d_vdir_pts, uint32_t* status,
                               const float* d_raw_absmax, hipStream_t stream);

// 22-bit variant (mlp_bwd_s.hip)
int benerf_mlp_dx_split22_launch(const BenerfMlpParams* params, const float* packed, int channels, int64_t M, const float* d_raw,
                                 const float* acts, float* dacts, float* d_pts, float* d_vdir_pts, uint32_t* status,
                                 const float* d_raw_absmax, hipStream_t stream);

static int launch_dx(const BenerfMlpParams* params, const float* packed, int channels, int64_t M, const float* d_raw,
                     const float* acts, float* dacts, float* d_pts, float* d_vdir_pts, hipStream_t stream) {
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
    a.M = M;
    const int64_t tiles = (M + mlp::TM - 1) / mlp::TM;
    BENERF_REQUIRE(tiles < (1ll << 31), "mlp_bwd: too many points");
    dim3 grid((unsigned)tiles), block(mlp::NTHREADS);
    const int smem = (int)mlp::TILE_SMEM;
    static BenerfLdsAttr attr[2];       // once per device and variant
    if (!benerf_lds_attr(attr[channels == 1 ? 0 : 1], channels == 1 ? (const void*)mlp_bwd_kernel<1> : (const void*)mlp_bwd_kernel<3>, smem)) {
        benerf_set_error("mlp_bwd(dx): cannot reserve %d bytes of LDS", smem);
        return BENERF_EHIP;
    }
    if (channels == 1) hipLaunchKernelGGL((mlp_bwd_kernel<1>), grid, block, smem, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_kernel<3>), grid, block, smem, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dx)");
    return BENERF_OK;
}

extern "C" int benerf_mlp_bwd_dx(const BenerfMlpParams* params, const float* packed, int channels, int n_rays,
                                 int n_samples, const float* d_raw, const float* acts, float* dacts, float* d_pts,
                                 float* d_vdir_pts, int precision, uint32_t* status, const float* d_raw_absmax,
                                 benerf_stream_t stream) {
    BENERF_REQUIRE(params && packed && d_raw && acts && dacts && d_pts && d_vdir_pts, "mlp_bwd_dx: null pointer");
    BENERF_REQUIRE(channels == 1 || channels == 3, "mlp_bwd_dx: channels must be 1 or 3");
    BENERF_REQUIRE(n_rays > 0 && n_samples > 0, "mlp_bwd_dx: bad sizes");
    BENERF_REQUIRE(precision == BENERF_MLP_F32 || precision == BENERF_MLP_SPLIT || precision == BENERF_MLP_SPLIT_F16BWD,
                   "mlp_bwd_dx: precision must be BENERF_MLP_F32, BENERF_MLP_SPLIT or BENERF_MLP_SPLIT_F16BWD");
    for (int l = 0; l < BENERF_NLAYERS; ++l) BENERF_REQUIRE(params->w[l], "mlp_bwd_dx: null parameter %d", l);
    const int64_t M = (int64_t)n_rays * n_samples;
    if (precision == BENERF_MLP_SPLIT)
        return benerf_mlp_dx_split22_launch(params, packed, channels, M, d_raw, acts, dacts, d_pts, d_vdir_pts, status, d_raw_absmax,
                                            as_stream(stream));
    if (precision == BENERF_MLP_SPLIT_F16BWD)
        return benerf_mlp_dx_split_launch(params, packed, channels, M, d_raw, acts, dacts, d_pts, d_vdir_pts, status, d_raw_absmax,
                                          as_stream(stream));
    return launch_dx(params, packed, channels, M, d_raw, acts, dacts, d_pts, d_vdir_pts, as_stream(stream));
}

extern "C" int benerf_mlp_bwd_dw(const BenerfMlpParams* params, int channels, int n_rays, int n_samples, const float* d_raw,
                                 const float* acts, const float* dacts, float* dw_ws, size_t dw_ws_floats, const BenerfMlpGrads* grads,
                                 int accumulate, int precision, const float* pe_weights, benerf_stream_t stream) {
    BENERF_REQUIRE(d_raw && acts && dacts && dw_ws && grads, "mlp_bwd_dw: null pointer");
    BENERF_REQUIRE(precision != BENERF_MLP_SPLIT || (params && params->w[BENERF_L_VIEWS] && params->w[BENERF_L_FEAT] && params->b[BENERF_L_FEAT]),
                   "mlp_bwd_dw: BENERF_MLP_SPLIT needs the network's parameters (views / feature weights, feature bias)");
    BENERF_REQUIRE(channels == 1 || channels == 3, "mlp_bwd_dw: channels must be 1 or 3");
    BENERF_REQUIRE(n_rays > 0 && n_samples > 0, "mlp_bwd_dw: bad sizes");
    BENERF_REQUIRE(precision == BENERF_MLP_F32 || precision == BENERF_MLP_SPLIT || precision == BENERF_MLP_SPLIT_F16BWD,
                   "mlp_bwd_dw: precision must be BENERF_MLP_F32, BENERF_MLP_SPLIT or BENERF_MLP_SPLIT_F16BWD");
    if (dw_ws_floats < (size_t)mlp::DW_WS_FLOATS) {
        benerf_set_error("mlp_bwd_dw: dw workspace too small (%zu < %lld floats)", dw_ws_floats, (long long)mlp::DW_WS_FLOATS);
        return BENERF_EWORKSPACE;
    }
    for (int l = 0; l < BENERF_NLAYERS; ++l) BENERF_REQUIRE(grads->w[l] && grads->b[l], "mlp_bwd_dw: null grad %d", l);
    return benerf_mlp_dw_launch(params, precision, channels, (int64_t)n_rays * n_samples, d_raw, acts, dacts, dw_ws, grads, accumulate,
                                pe_weights, as_stream(stream));
}
