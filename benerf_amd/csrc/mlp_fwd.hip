// K3 forward: fused positional encoding + NeRF MLP for one tile of 64 sample points per
// workgroup (4 wavefronts), exact-f32 MFMA.  Replaces Embedder.embed (model/embedder.py:9-34)
// + NeRF.forward (model/nerf.py:67-116) + pts = o + d*z (model/nerf.py:308,327).
//
// Data flow per tile (LDS tile layout and swizzle: mlp_common.h):
//   prologue : pts, PE(pts) -> T[:,256:320)
//   L0       : T[:,256:320) x W0^T            -> relu -> T[:,0:256)
//   L1..L4   : T[:,0:256)   x Wl^T            -> relu -> T[:,0:256)
//   L5       : T[:,0:320)   x [W5h|W5pe]^T    -> relu -> T[:,0:256)   (skip connection)
//   L6,L7    : as L1
//   alpha    : VALU dot(T[:,0:256), w_alpha)  ; PE(dir) -> T[:,256:288)
//   FEAT     : T[:,0:256)   x Wf^T   (linear) -> T[:,0:256)
//   VIEWS    : T[:,0:288)   x Wv^T            -> relu -> T[:,0:128)
//   rgb      : VALU dot(T[:,0:128), w_rgb[c])
// Each wave owns 64 output features (2 MFMA column tiles) x all 64 points (2 row tiles):
// per 8-deep k-block it issues 2 ds_read_b128 (A, points) + 2 global_load_dwordx4 (B,
// weights, L2-resident) for 16 MFMAs (1024 cycles) - operand traffic is negligible, the
// kernel is bound by the f32 matrix pipe.  The tile is exactly 80 KiB and the kernel fits 256
// registers, so TWO workgroups share a CU: while one is in an epilogue (bias/ReLU, LDS writes,
// activation stores for the backward pass, barriers) the other keeps the matrix pipe busy.
#include "mlp_split.h"

namespace {
using namespace mlp;

struct FwdArgs {
    const float* rays_o;
    const float* rays_d;
    const float* viewdirs;
    const float* z;
    const float* packed;
    const float* bias[10];   // L0..L7, views, feat  (index by BENERF layer id for 8, 9)
    const float* w_alpha;
    const float* b_alpha;
    const float* w_rgb;
    const float* b_rgb;
    const float* pe_w;       // BARF c2f column weights (include/benerf_hip.h) or null
    float* raw;
    float* acts;
    const uint32_t* gate;    // BENERF_MLP_AUTO re-run: workgroups exit unless *gate (max |activation| of the split launch, f32 bits) left f16's range
    int64_t M;
    int S;
};

// acc[r][c] += T[rows r*32.., kcol0 + 0..KB*8) x Wp(tile ct0+c);  kcol0 is a multiple of 64
template <int KB, int NCT>
__device__ __forceinline__ void gemm_stage(const float* __restrict__ T, int kcol0, const float* __restrict__ wp,
                                           int ct0, int lane, f32x16 (&acc)[2][NCT]) {
    const int row = lane & 31;
    const int sw = swz(row);                       // rows row and row+32 share the swizzle
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

template <int NCT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NCT]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
}

// bias (+ReLU) -> LDS tile columns [ct*32..) and, in training mode, to the [M][LDO] activation
// array whose tile starts at `save_tile` (row m0).  Every store instruction writes two full 128-B
// rows.  Returns the ReLU sign bits of this lane's accumulator elements (bit ((c*2+r)*16+e)).
// `rows_valid` (block-uniform) < 64 only for the ragged last tile: the common path has no
// per-element guards.
// Activation stores: STAGED = false writes the accumulators straight from registers (64 dword stores per lane
// and stage, two full 128-B rows per instruction); STAGED = true leaves them to copy_tile() after the barrier
// (16-byte stores, 4x fewer instructions).  Measured at M = 522 368 with two workgroups per CU: 5.29 ms vs
// 5.27 ms - the co-resident workgroup already hides the store issue time, so the simpler path is kept.
constexpr bool STAGED = false;

// LDS tile columns [0, W) of the first `rows_valid` rows -> [rows][W] global array: 16-byte stores, one full row
// (1 KiB for W = 256) per wave instruction, 4x fewer store instructions than the register path.
template <int W>
__device__ __forceinline__ void copy_tile(const float* __restrict__ T, float* __restrict__ dst_tile, int rows_valid) {
    constexpr int Q = W / 4;                      // float4 per row
#pragma unroll 4
    for (int j = 0; j < TM * Q / NTHREADS; ++j) {
        const int q = threadIdx.x + j * NTHREADS;
        const int row = q / Q, c4 = q % Q;
        const float4 v = *reinterpret_cast<const float4*>(T + tidx(row, c4 * 4));
        if (rows_valid >= TM || row < rows_valid) *reinterpret_cast<float4*>(dst_tile + (int64_t)row * W + c4 * 4) = v;
    }
}

template <int NCT, bool RELU, bool SAVE, int LDO>
__device__ __forceinline__ uint64_t epilogue(f32x16 (&acc)[2][NCT], float* __restrict__ T, int ct0, int lane,
                                             const float* __restrict__ bias, float* __restrict__ save_tile,
                                             int rows_valid) {
    const int lr = lane & 31, r4 = 4 * (lane >> 5);
    uint64_t bits = 0;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int n = (ct0 + c) * 32 + lr;
        const float bv = bias[n];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r * 32 + (e & 3) + 8 * (e >> 2) + r4;
                float v = acc[r][c][e] + bv;
                if (RELU) {
                    v = fmaxf(v, 0.f);
                    bits |= (uint64_t)(v > 0.f) << ((c * 2 + r) * 16 + e);
                }
                acc[r][c][e] = v;
                T[tidx(row, n)] = v;
            }
        }
    }
    if (SAVE && !STAGED) {
        float* sv_lane = save_tile + (int64_t)r4 * LDO + lr;
        if (rows_valid >= TM) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        sv_lane[(r * 32 + (e & 3) + 8 * (e >> 2)) * LDO + (ct0 + c) * 32] = acc[r][c][e];
        } else {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rowoff = r * 32 + (e & 3) + 8 * (e >> 2);
                        if (rowoff + r4 < rows_valid) sv_lane[rowoff * LDO + (ct0 + c) * 32] = acc[r][c][e];
                    }
        }
    }
    return bits;
}

template <int C, bool SAVE>
__global__ __launch_bounds__(NTHREADS, 2) void mlp_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float T[];   // [TM][LD], swizzled (mlp_common.h)
    if (a.gate && __builtin_amdgcn_readfirstlane((int)(*a.gate < __float_as_uint(F16_RANGE_LIMIT)))) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int64_t M = a.M;
    const int pt = tid & 63;           // this thread's point in prologue / VALU phases
    const int grp = tid >> 6;
    const int64_t m = m0 + pt;
    const int64_t mc = m < M ? m : M - 1;
    const int64_t ray = mc / a.S;
    float* acts = a.acts;
    const int rows_valid = (int)(M - m0 < TM ? M - m0 : TM);          // block-uniform; < 64 only in the last tile
    float* act_h_tile = SAVE ? acts + act_h(M, 0) + m0 * 256 : nullptr;  // layer l: + l * M * 256
    uint64_t* mask_out = SAVE ? reinterpret_cast<uint64_t*>(acts + act_mask(M)) + (int64_t)blockIdx.x * NTHREADS + tid
                              : nullptr;                                 // layer l: + l * n_tiles * 256
    const int64_t mask_stride = n_tiles(M) * NTHREADS;
    float* trow = T + pt * LD;          // this thread's point row; column c lives at trow[c ^ psw]
    const int psw = swz(pt);

    // ---- prologue: pts = o + d*z (separately rounded like torch), PE(pts) ----------------------
    {
        const float zz = a.z[mc];
        float x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = __fadd_rn(a.rays_o[ray * 3 + c], __fmul_rn(a.rays_d[ray * 3 + c], zz));
        if (grp == 0) {
            trow[(COL_PE + 0) ^ psw] = x[0];
            trow[(COL_PE + 1) ^ psw] = x[1];
            trow[(COL_PE + 2) ^ psw] = x[2];
            trow[(COL_PE + 63) ^ psw] = 0.f;
        }
        // 30 (freq, dim) pairs, strided over the 4 thread groups        model/embedder.py:13-28
        for (int p = grp; p < 30; p += 4) {
            const int f = p / 3, d = p - 3 * f;
            const float v = x[d] * (float)(1 << f);
            float s, c;
            sincosf(v, &s, &c);
            trow[(COL_PE + 3 + f * 6 + d) ^ psw] = s;
            trow[(COL_PE + 3 + f * 6 + 3 + d) ^ psw] = c;
        }
    }
    lds_barrier();
    if (SAVE) {   // PE tile -> acts (256 B per point, 4 threads per row)
        const int r = tid >> 2, q = tid & 3;
        if (r < rows_valid) {
            float4* dst = reinterpret_cast<float4*>(acts + act_pe(M) + (m0 + r) * ACT_PE_W + q * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const float4*>(T + tidx(r, COL_PE + q * 16 + 4 * j));
        }
    }
    if (a.pe_w) {   // BARF c2f: the saved tile stays unweighted (dX needs sin / cos themselves), the GEMM operand is scaled
        if (SAVE) lds_barrier();
        for (int col = grp; col < 64; col += 4) trow[(COL_PE + col) ^ psw] *= a.pe_w[col];
        lds_barrier();
    }

    f32x16 acc[2][2];
    const int ct0 = wave * 2;

    // ---- L0 ---------------------------------------------------------------------------------
    zero_acc(acc);
    gemm_stage<8, 2>(T, COL_PE, a.packed + pack_offset(PF_L0), ct0, lane, acc);
    // L0 reads columns >= 256 and writes columns < 256: no barrier needed before the epilogue
    {
        const uint64_t bits = epilogue<2, true, SAVE, 256>(acc, T, ct0, lane, a.bias[0], act_h_tile, rows_valid);
        if (SAVE) mask_out[0] = bits;
    }
    lds_barrier();
    if (SAVE && STAGED) copy_tile<256>(T, act_h_tile, rows_valid);

    // ---- L1..L7 -------------------------------------------------------------------------------
#pragma unroll 1
    for (int l = 1; l < 8; ++l) {
        zero_acc(acc);
        if (l == 5) gemm_stage<40, 2>(T, 0, a.packed + pack_offset(PF_L5), ct0, lane, acc);
        else gemm_stage<32, 2>(T, 0, a.packed + pack_offset(PF_L0 + l), ct0, lane, acc);
        lds_barrier();   // every wave finished reading the previous hidden state
        const uint64_t bits = epilogue<2, true, SAVE, 256>(acc, T, ct0, lane, a.bias[l],
                                                           SAVE ? act_h_tile + (int64_t)l * M * 256 : nullptr, rows_valid);
        if (SAVE) mask_out[l * mask_stride] = bits;
        lds_barrier();
        if (SAVE && STAGED) copy_tile<256>(T, act_h_tile + (int64_t)l * M * 256, rows_valid);
    }

    // ---- alpha partials (reads h7) + PE(viewdir) into columns [256,288) ---------------------------
    // scratch columns [288,320) (dead PE columns): alpha partial of group g at column 288 + g
    {
        const float* wa = a.w_alpha + grp * 64;
        float s = 0.f;
#pragma unroll 4
        for (int k = 0; k < 64; k += 4) {
            const float4 h = *reinterpret_cast<const float4*>(trow + ((grp * 64 + k) ^ psw));
            const float4 w = *reinterpret_cast<const float4*>(wa + k);
            s += h.x * w.x + h.y * w.y + h.z * w.z + h.w * w.w;
        }
        trow[(COL_SCR + grp) ^ psw] = s;
        float vd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) vd[c] = a.viewdirs[ray * 3 + c];
        if (grp == 0) {
            trow[(COL_PE + 0) ^ psw] = vd[0];
            trow[(COL_PE + 1) ^ psw] = vd[1];
            trow[(COL_PE + 2) ^ psw] = vd[2];
        }
        if (grp == 1) {
#pragma unroll
            for (int k = 27; k < 32; ++k) trow[(COL_PE + k) ^ psw] = 0.f;
        }
        for (int p = grp; p < 12; p += 4) {
            const int f = p / 3, d = p - 3 * f;
            const float v = vd[d] * (float)(1 << f);
            float sn, cs;
            sincosf(v, &sn, &cs);
            trow[(COL_PE + 3 + f * 6 + d) ^ psw] = sn;
            trow[(COL_PE + 3 + f * 6 + 3 + d) ^ psw] = cs;
        }
    }

    // ---- FEAT (linear) ----------------------------------------------------------------------------
    zero_acc(acc);
    gemm_stage<32, 2>(T, 0, a.packed + pack_offset(PF_FEAT), ct0, lane, acc);
    lds_barrier();   // h7 fully consumed (GEMM + alpha partials); PE(dir) + alpha partials visible
    epilogue<2, false, SAVE, 256>(acc, T, ct0, lane, a.bias[BENERF_L_FEAT], SAVE ? acts + act_feat(M) + m0 * 256 : nullptr,
                                  rows_valid);
    if (tid < 64 && m < M) {
        const float4 p = *reinterpret_cast<const float4*>(trow + (COL_SCR ^ psw));
        a.raw[m * (C + 1) + C] = ((p.x + p.y) + (p.z + p.w)) + a.b_alpha[0];
    }
    if (SAVE) {   // PE(dir) tile -> acts (128 B per point)
        const int r = tid >> 2, q = tid & 3;
        if (r < rows_valid) {
            float4* dst = reinterpret_cast<float4*>(acts + act_ped(M) + (m0 + r) * ACT_PED_W + q * 8);
            dst[0] = *reinterpret_cast<const float4*>(T + tidx(r, COL_PE + q * 8));
            dst[1] = *reinterpret_cast<const float4*>(T + tidx(r, COL_PE + q * 8 + 4));
        }
    }
    if (a.pe_w) {
        if (SAVE) lds_barrier();
        for (int col = grp; col < 32; col += 4) trow[(COL_PE + col) ^ psw] *= a.pe_w[64 + col];
    }
    lds_barrier();
    if (SAVE && STAGED) copy_tile<256>(T, acts + act_feat(M) + m0 * 256, rows_valid);

    // ---- VIEWS: [feature | PE(dir)] (288) -> 128, one column tile per wave ----------------------------
    {
        f32x16 av[2][1];
        zero_acc(av);
        gemm_stage<36, 1>(T, 0, a.packed + pack_offset(PF_VIEWS), wave, lane, av);
        lds_barrier();
        epilogue<1, true, SAVE, ACT_HV_W>(av, T, wave, lane, a.bias[BENERF_L_VIEWS],
                                          SAVE ? acts + act_hv(M) + m0 * ACT_HV_W : nullptr, rows_valid);
    }
    lds_barrier();
    if (SAVE && STAGED) copy_tile<ACT_HV_W>(T, acts + act_hv(M) + m0 * ACT_HV_W, rows_valid);

    // ---- rgb: 128 -> C on the VALU; partial of (channel c, group g) at scratch column 288 + 4 + 4c + g ----
    {
        float s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = 0.f;
#pragma unroll 2
        for (int k = 0; k < 32; k += 4) {
            const float4 h = *reinterpret_cast<const float4*>(trow + ((grp * 32 + k) ^ psw));
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float4 w = *reinterpret_cast<const float4*>(a.w_rgb + c * 128 + grp * 32 + k);
                s[c] += h.x * w.x + h.y * w.y + h.z * w.z + h.w * w.w;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) trow[(COL_SCR + 4 + 4 * c + grp) ^ psw] = s[c];
    }
    lds_barrier();
    if (tid < 64 && m < M) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 p = *reinterpret_cast<const float4*>(trow + ((COL_SCR + 4 + 4 * c) ^ psw));
            a.raw[m * (C + 1) + c] = ((p.x + p.y) + (p.z + p.w)) + a.b_rgb[c];
        }
    }
}

}  // namespace

// split-f16 variant (mlp_fwd_h.hip)
int benerf_mlp_fwd_split_launch(const BenerfMlpParams* params, const float* packed, int channels, int n_rays, int n_samples,
                                const float* rays_o, const float* rays_d, const float* viewdirs, const float* z, float* raw,
                                float* acts, int save_lo, int fuse, uint32_t* status, hipStream_t stream);

extern "C" int benerf_mlp_fwd(const BenerfMlpParams* params, const float* packed, int channels, int n_rays,
                              int n_samples, const float* rays_o, const float* rays_d, const float* viewdirs,
                              const float* z, float* raw, float* acts, int precision, uint32_t* status,
                              benerf_stream_t stream) {
    BENERF_REQUIRE(params && packed && rays_o && rays_d && viewdirs && z && raw, "mlp_fwd: null pointer");
    BENERF_REQUIRE(channels == 1 || channels == 3, "mlp_fwd: channels must be 1 or 3");
    BENERF_REQUIRE(n_rays > 0 && n_samples > 0, "mlp_fwd: bad sizes");
    BENERF_REQUIRE(precision == BENERF_MLP_F32 || precision == BENERF_MLP_SPLIT || precision == BENERF_MLP_AUTO ||
                       precision == BENERF_MLP_SPLIT_F16BWD, "mlp_fwd: unknown precision %d", precision);
    BENERF_REQUIRE(precision != BENERF_MLP_AUTO || (status && !acts),
                   "mlp_fwd: BENERF_MLP_AUTO is the inference mode (acts == NULL) and needs a status word");
    for (int l = 0; l < BENERF_NLAYERS; ++l) BENERF_REQUIRE(params->b[l] && params->w[l], "mlp_fwd: null parameter %d", l);
    const uint32_t* gate = nullptr;
    if (precision != BENERF_MLP_F32) {
        if (precision == BENERF_MLP_AUTO) {   // per-call maximum in status[3]; the f32 launch below runs only if it left the range
            if (hipMemsetAsync(status + 3, 0, sizeof(uint32_t), as_stream(stream)) != hipSuccess) {
                benerf_set_error("mlp_fwd: memset failed");
                return BENERF_EHIP;
            }
            gate = status + 3;
        }
        // BENERF_MLP_SPLIT_F16BWD (training and inference alike) runs the unfused layer sequence: its backward wants `feature`
        const int rc = benerf_mlp_fwd_split_launch(params, packed, channels, n_rays, n_samples, rays_o, rays_d, viewdirs, z, raw, acts,
                                                   precision == BENERF_MLP_SPLIT, precision != BENERF_MLP_SPLIT_F16BWD, status,
                                                   as_stream(stream));
        if (rc != BENERF_OK || precision != BENERF_MLP_AUTO) return rc;
    }
    FwdArgs a;
    a.rays_o = rays_o;
    a.rays_d = rays_d;
    a.viewdirs = viewdirs;
    a.z = z;
    a.packed = packed;
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
    a.gate = gate;
    a.M = (int64_t)n_rays * n_samples;
    a.S = n_samples;
    const int64_t tiles = (a.M + mlp::TM - 1) / mlp::TM;
    BENERF_REQUIRE(tiles < (1ll << 31), "mlp_fwd: too many points");
    dim3 grid((unsigned)tiles), block(mlp::NTHREADS);
    const int smem = (int)mlp::TILE_SMEM;
#define BENERF_FWD_LAUNCH(CH, SV)                                                                                  \
    do {                                                                                                           \
        static BenerfLdsAttr attr_;                                                                                \
        if (!benerf_lds_attr(attr_, (const void*)mlp_fwd_kernel<CH, SV>, smem)) {                                  \
            benerf_set_error("mlp_fwd: cannot reserve %d bytes of LDS", smem);                                     \
            return BENERF_EHIP;                                                                                    \
        }                                                                                                          \
        hipLaunchKernelGGL((mlp_fwd_kernel<CH, SV>), grid, block, smem, as_stream(stream), a);                     \
    } while (0)
    if (channels == 1) {
        if (acts) BENERF_FWD_LAUNCH(1, true);
        else BENERF_FWD_LAUNCH(1, false);
    } else {
        if (acts) BENERF_FWD_LAUNCH(3, true);
        else BENERF_FWD_LAUNCH(3, false);
    }
#undef BENERF_FWD_LAUNCH
    BENERF_LAUNCH_CHECK("mlp_fwd");
    return BENERF_OK;
}
