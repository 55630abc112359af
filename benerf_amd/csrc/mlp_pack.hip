// K3 helper: re-pack one network's nn.Linear weights ([out,in] row-major, model/nerf.py:53-64)
// into the MFMA-operand-shaped blocks described in mlp_common.h.  One launch per network
// per optimiser step (2.4 M floats read, 4.9 M written) - once as f32 blocks, once as split-f16 blocks.
#include "mlp_split.h"

namespace {
using namespace mlp;

struct PackArgs {
    const float* w[BENERF_NLAYERS];
    const float* b_feat;     // fuse_views_kernel
    const float* b_views;
    float* packed;
};

// W_c = W_v[:, :256] W_f and b_c = W_v[:, :256] b_f + b_v in float64 (mlp_common.h: PF_VIEWSC) -> f32 [128][283] (the PE(dir)
// columns copied) + [128] behind the packed sections.  32 x 32 output tiles through LDS; blocks [0, 32): W_c, block 32: the rest.
__device__ __forceinline__ void fuse_body(const PackArgs& a) {
    // One load phase for the whole contraction range (every global load of the block in flight at once; the operands are f32, so
    // the tiles [32][256] / [256][32] are 64 KiB of LDS as floats) instead of 8 chunks with two barriers each: the kernel was a chain
    // of global-load latencies on the step's critical path (between Adam and the re-pack).  Same float64 products, same summation
    // order (k ascending): bit-identical W_c.
    __shared__ float As[32 * 256], Bs[256 * 32];
    const float* Wv = a.w[BENERF_L_VIEWS];
    const float* Wf = a.w[BENERF_L_FEAT];
    float* out = a.packed + 2 * PACKED_FLOATS;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5, b = blockIdx.x;
    if (b == 32) {
        for (int e = t; e < 128 * 27; e += 256) out[(e / 27) * 283 + 256 + e % 27] = Wv[(e / 27) * 283 + 256 + e % 27];
        if (t < 128) {
            double acc = (double)a.b_views[t];
            for (int j = 0; j < 256; ++j) acc += (double)Wv[t * 283 + j] * (double)a.b_feat[j];
            out[FUSED_W_FLOATS + t] = (float)acc;
        }
        return;
    }
    const int m0 = (b >> 3) * 32, n0 = (b & 7) * 32;      // W_c[m][n] = sum_k W_v[m][k] W_f[k][n]
    for (int k0 = 0; k0 < 256; k0 += 32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = ty + 8 * r;
            As[q * 256 + k0 + tx] = Wv[(m0 + q) * 283 + k0 + tx];
            Bs[(k0 + q) * 32 + tx] = Wf[(k0 + q) * 256 + n0 + tx];
        }
    }
    __syncthreads();
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int k = 0; k < 256; ++k) {
        const double bv = (double)Bs[k * 32 + tx];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += (double)As[(ty + 8 * r) * 256 + k] * bv;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(m0 + ty + 8 * r) * 283 + n0 + tx] = (float)acc[r];
}

// source element for packed block `id`, column index `col` (tile*32 + lane&31) and
// contraction index `kk` (kblock*8 + 4*(lane>>5) + i); returns 0 for padding.
__device__ __forceinline__ float pack_source(const PackArgs& a, int id, int col, int kk) {
    const int layer = pack_layer(id);
    // the fused blocks read [W_c | W_v[:, 256:283]] as fuse_views_kernel left it behind the packed sections
    const float* W = (id == PF_VIEWSC || id == PB_VIEWSC) ? a.packed + 2 * PACKED_FLOATS : a.w[layer];
    const int in = layer_in(layer);
    if (pack_is_forward(id)) {
        // forward: col = output feature n, kk = packed input index
        const int n = col;
        int k;
        if (id == PF_L0) {
            if (kk >= 63) return 0.f;
            k = kk;
        } else if (id == PF_L5) {          // packed order [h4 (256) | PE (63)], nn.Linear order [PE | h4]
            if (kk < 256) k = 63 + kk;
            else if (kk < 319) k = kk - 256;
            else return 0.f;
        } else if (id == PF_VIEWS || id == PF_VIEWSC) {       // [feature (256) | PE_dir (27)] = nn.Linear order
            if (kk >= 283) return 0.f;
            k = kk;
        } else {
            k = kk;
        }
        return W[(int64_t)n * in + k];
    }
    // backward: col = input feature (packed order), kk = output feature n (contraction)
    const int n = kk;
    int k;
    if (id == PB_L5) {
        if (col < 256) k = 63 + col;
        else if (col < 319) k = col - 256;
        else return 0.f;
    } else if (id == PB_L0) {
        if (col >= 63) return 0.f;
        k = col;
    } else {
        if ((id == PB_VIEWS || id == PB_VIEWSC) && col >= 283) return 0.f;   // 256 feature columns, then the 27 PE(dir) columns, then padding
        k = col;
    }
    return W[(int64_t)n * in + k];
}

struct PackArgs2 { PackArgs net[2]; };

__device__ __forceinline__ void pack_body(const PackArgs& a);

__global__ void pack_kernel(PackArgs a) { pack_body(a); }
// both networks of a training step in one launch (blockIdx.z)
__global__ void pack_pair_kernel(PackArgs2 a) { pack_body(a.net[blockIdx.z]); }
__global__ __launch_bounds__(256) void fuse_views_kernel(PackArgs a) { fuse_body(a); }
__global__ __launch_bounds__(256) void fuse_views_pair_kernel(PackArgs2 a) { fuse_body(a.net[blockIdx.z]); }

__device__ __forceinline__ void pack_body(const PackArgs& a) {
    const int id = blockIdx.y;
    const PackShape sh = pack_shape(id);
    const int64_t n4 = (int64_t)sh.tiles * sh.kblocks * 64;   // float4 slots
    float4* dst = reinterpret_cast<float4*>(a.packed + pack_offset(id));
    if (id == PB_VIEWSPE) {   // [g][n][8] <- Wv[n][256 + g + 4q], zero beyond the 27 PE(dir) columns
        float* d = a.packed + pack_offset(id);
        const float* W = a.w[BENERF_L_VIEWS];
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 4 * 128 * 8; e += gridDim.x * blockDim.x) {
            const int q = e & 7, n = (e >> 3) & 127, g = e >> 10;
            const int j = g + 4 * q;
            d[e] = j < 27 ? W[n * 283 + 256 + j] : 0.f;
        }
        return;
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        int lane = (int)(e & 63);
        int64_t tb = e >> 6;
        int kb = (int)(tb % sh.kblocks);
        int tile = (int)(tb / sh.kblocks);
        int col = tile * 32 + (lane & 31);
        int k0 = kb * 8 + 4 * (lane >> 5);
        float4 v;
        v.x = pack_source(a, id, col, k0 + 0);
        v.y = pack_source(a, id, col, k0 + 1);
        v.z = pack_source(a, id, col, k0 + 2);
        v.w = pack_source(a, id, col, k0 + 3);
        dst[e] = v;
    }
    // split-f16 section (mlp_split.h): [tile pair][kstep of 16][tile in pair][plane hi|lo][64 lanes][8 halfs]; lane l
    // holds col = tile*32 + (l&31), k = kstep*16 + 8*(l>>5) + j.  The four fragments a wave needs per k-step (two
    // column tiles x hi/lo) are one contiguous 4 KiB.  Same float count as the f32 block.
    // Forward blocks: lo = (w - hi) * 2^11 (the forward kernel keeps a second accumulator set for the cross terms).
    // Backward blocks: lo = w - hi UNSCALED (the f16 dX kernel adds dY*hi and dY*lo into one accumulator; |w| ~ 0.1
    // puts lo into f16's subnormals, absolute floor 2^-25, i.e. the weight still carries ~20 bits).
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    half8* dsth = reinterpret_cast<half8*>(a.packed + PACKED_FLOATS + pack_offset(id));
    const int ksteps = sh.kblocks / 2;
    const int64_t n8 = (int64_t)sh.tiles * ksteps * 64;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        const int64_t tb = e >> 6;
        const int ks = (int)(tb % ksteps);
        const int tile = (int)(tb / ksteps);
        const int col = tile * 32 + (lane & 31);
        const int k0 = ks * 16 + 8 * (lane >> 5);
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = pack_source(a, id, col, k0 + j);
            hi[j] = (_Float16)w;
            lo[j] = (_Float16)((w - (float)hi[j]) * (pack_is_forward(id) ? 2048.f : 1.f));
        }
        const int64_t frag = (((int64_t)(tile >> 1) * ksteps + ks) * 2 + (tile & 1)) * 2;   // 1 KiB fragments
        dsth[frag * 64 + lane] = hi;
        dsth[(frag + 1) * 64 + lane] = lo;
    }
}

// known-answer access to the lo8 codec of the saved operands (mlp_split.h): one thread per unit of 8 values
template <int S>
__global__ void h8_roundtrip_kernel(const float* __restrict__ x, int64_t units, uint16_t* __restrict__ hi_out, uint8_t* __restrict__ code_out,
                                    float* __restrict__ dec_out) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    uint32_t hi[4], lo[4], dl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[u * 8 + 2 * j], b = x[u * 8 + 2 * j + 1];
        const _Float16 ha = (_Float16)a, hb = (_Float16)b;
        const h8_half2 h = {ha, hb};
        const h8_half2 l = {(_Float16)((a - (float)ha) * (float)(1 << S)), (_Float16)((b - (float)hb) * (float)(1 << S))};
        hi[j] = __builtin_bit_cast(uint32_t, h);
        lo[j] = __builtin_bit_cast(uint32_t, l);
    }
    const uint2 code = h8_encode_unit<S>(hi, lo);
    h8_decode_unit(hi, code, dl);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const h8_half2 h = __builtin_bit_cast(h8_half2, hi[j]), d = __builtin_bit_cast(h8_half2, dl[j]);
        hi_out[u * 8 + 2 * j] = (uint16_t)(hi[j] & 0xffffu);
        hi_out[u * 8 + 2 * j + 1] = (uint16_t)(hi[j] >> 16);
        dec_out[u * 8 + 2 * j] = (float)h[0] + (float)d[0];
        dec_out[u * 8 + 2 * j + 1] = (float)h[1] + (float)d[1];
    }
    reinterpret_cast<uint2*>(code_out)[u] = code;
}

}  // namespace

extern "C" int benerf_mlp_h8_roundtrip(const float* x, int64_t n, int residual_log2_scale, uint16_t* hi_bits, uint8_t* codes,
                                       float* decoded, benerf_stream_t stream) {
    BENERF_REQUIRE(x && hi_bits && codes && decoded && n > 0 && n % 8 == 0, "mlp_h8_roundtrip: bad args (n must be a multiple of 8)");
    BENERF_REQUIRE(residual_log2_scale == 11 || residual_log2_scale == 12, "mlp_h8_roundtrip: residual scale must be 2^11 (forward planes) or 2^12 (dX)");
    const int64_t units = n / 8;
    const dim3 grid((unsigned)((units + 255) / 256)), block(256);
    if (residual_log2_scale == 11) hipLaunchKernelGGL(h8_roundtrip_kernel<11>, grid, block, 0, as_stream(stream), x, units, hi_bits, codes, decoded);
    else hipLaunchKernelGGL(h8_roundtrip_kernel<12>, grid, block, 0, as_stream(stream), x, units, hi_bits, codes, decoded);
    BENERF_LAUNCH_CHECK("mlp_h8_roundtrip");
    return BENERF_OK;
}

// f32 blocks | split-f16 blocks | fused feature -> views matrix and bias (f32)
extern "C" size_t benerf_mlp_packed_floats(void) { return (size_t)(2 * mlp::PACKED_FLOATS + mlp::FUSED_FLOATS); }
// buffers are sized for either arithmetic mode (the split mode pads the point count to whole 128-point tiles)
extern "C" size_t benerf_mlp_act_floats(int64_t n_points) {
    const int64_t a = mlp::act_total_floats(n_points), b = mlp::sact22_total_floats(n_points);   // sact22 >= sact
    return (size_t)(a > b ? a : b);
}
extern "C" size_t benerf_mlp_dact_floats_per_point(void) { return (size_t)mlp::DACT_PER_POINT; }
extern "C" size_t benerf_mlp_dact_floats(int64_t n_points) {
    const int64_t a = n_points * mlp::DACT_PER_POINT, b = mlp::sdact22_total_floats(n_points);   // sdact22 >= sdact
    return (size_t)(a > b ? a : b);
}
// the same two buffers sized for ONE arithmetic mode (`precision`: the K3 launch argument): the lo8 twins of BENERF_MLP_SPLIT add a
// third to the activations and half to the gradients, which the other modes never touch.  0 for an unknown mode.
extern "C" size_t benerf_mlp_act_floats_for(int64_t n_points, int precision) {
    switch (precision) {
        case BENERF_MLP_F32: return (size_t)mlp::act_total_floats(n_points);
        case BENERF_MLP_SPLIT: return (size_t)mlp::sact22_total_floats(n_points);
        case BENERF_MLP_SPLIT_F16BWD: return (size_t)mlp::sact_total_floats(n_points);
        default: return 0;
    }
}
extern "C" size_t benerf_mlp_dact_floats_for(int64_t n_points, int precision) {
    switch (precision) {
        case BENERF_MLP_F32: return (size_t)(n_points * mlp::DACT_PER_POINT);
        case BENERF_MLP_SPLIT: return (size_t)mlp::sdact22_total_floats(n_points);
        case BENERF_MLP_SPLIT_F16BWD: return (size_t)mlp::sdact_total_floats(n_points);
        default: return 0;
    }
}
extern "C" size_t benerf_mlp_dw_workspace_floats(int64_t n_points) {
    (void)n_points;
#ifdef BENERF_TRACE_DW
    return (size_t)mlp::DW_WS_FLOATS + 4096;      // room for the tracing builds' time stamps (mlp_dw.hip)
#else
    return (size_t)mlp::DW_WS_FLOATS;
#endif
}

extern "C" int benerf_mlp_pack_weights(const BenerfMlpParams* params, int channels, float* packed,
                                       benerf_stream_t stream) {
    BENERF_REQUIRE(params && packed, "mlp_pack_weights: null pointer");
    BENERF_REQUIRE(channels >= 1 && channels <= 3, "mlp_pack_weights: channels must be 1..3");
    PackArgs a;
    for (int l = 0; l < BENERF_NLAYERS; ++l) {
        BENERF_REQUIRE(params->w[l], "mlp_pack_weights: null weight %d", l);
        a.w[l] = params->w[l];
    }
    BENERF_REQUIRE(params->b[BENERF_L_FEAT] && params->b[BENERF_L_VIEWS], "mlp_pack_weights: null feature / views bias");
    a.b_feat = params->b[BENERF_L_FEAT];
    a.b_views = params->b[BENERF_L_VIEWS];
    a.packed = packed;
    hipLaunchKernelGGL(fuse_views_kernel, dim3(33), dim3(256), 0, as_stream(stream), a);
    hipLaunchKernelGGL(pack_kernel, dim3(40, mlp::PACK_COUNT), dim3(256), 0, as_stream(stream), a);
    BENERF_LAUNCH_CHECK("mlp_pack_weights");
    return BENERF_OK;
}

extern "C" int benerf_mlp_pack_weights_pair(const BenerfMlpParams* params_a, float* packed_a, const BenerfMlpParams* params_b,
                                            float* packed_b, int channels, benerf_stream_t stream) {
    BENERF_REQUIRE(params_a && packed_a && params_b && packed_b, "mlp_pack_weights_pair: null pointer");
    BENERF_REQUIRE(channels >= 1 && channels <= 3, "mlp_pack_weights_pair: channels must be 1..3");
    PackArgs2 a;
    for (int l = 0; l < BENERF_NLAYERS; ++l) {
        BENERF_REQUIRE(params_a->w[l] && params_b->w[l], "mlp_pack_weights_pair: null weight %d", l);
        a.net[0].w[l] = params_a->w[l];
        a.net[1].w[l] = params_b->w[l];
    }
    BENERF_REQUIRE(params_a->b[BENERF_L_FEAT] && params_a->b[BENERF_L_VIEWS] && params_b->b[BENERF_L_FEAT] && params_b->b[BENERF_L_VIEWS],
                   "mlp_pack_weights_pair: null feature / views bias");
    a.net[0].b_feat = params_a->b[BENERF_L_FEAT];
    a.net[0].b_views = params_a->b[BENERF_L_VIEWS];
    a.net[1].b_feat = params_b->b[BENERF_L_FEAT];
    a.net[1].b_views = params_b->b[BENERF_L_VIEWS];
    a.net[0].packed = packed_a;
    a.net[1].packed = packed_b;
    hipLaunchKernelGGL(fuse_views_pair_kernel, dim3(33, 1, 2), dim3(256), 0, as_stream(stream), a);
    hipLaunchKernelGGL(pack_pair_kernel, dim3(40, mlp::PACK_COUNT, 2), dim3(256), 0, as_stream(stream), a);
    BENERF_LAUNCH_CHECK("mlp_pack_weights_pair");
    return BENERF_OK;
}
