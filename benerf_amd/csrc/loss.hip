// K6: event brightness-difference loss + blur-average photometric loss, value and gradient.
//
// Follows train.py:163-337:
//   event (synthetic, threshold > 0, train.py:207-236): target = acc*threshold;
//       diff = L(I_end) - L(I_start) (gray first when C == 3); loss = coeff * MSE(diff, target)
//   event (real, threshold <= 0, train.py:238-292): diff and target each divided by their
//       L2 norm over the batch (+1e-9) before the MSE
//   blur (train.py:299-331): blur = (sum_j rgb[j*R:(j+1)*R]) / n; loss = rgb_coeff * MSE
//   L = safelog log(x+1e-9) or linlog (x*255, linear below 20)   utils/math_utils.py:4-23
//   gray = .299 r + .587 g + .114 b                               utils/img_utils.py:7-16
// fine and coarse (rgb_map, rgb0) terms are summed.
//
// Two passes so a data-parallel job can all-reduce the 16 partial sums in between (the
// normalised loss needs the GLOBAL norms): loss_stats -> [all-reduce] -> loss_grads.
// Reductions are single-block, fixed order, in double => deterministic.
#include "common.h"

namespace {

constexpr int LT = 1024;

__device__ __forceinline__ float bright_log(float x, int linlog) {
    if (!linlog) return logf(x + 1e-9f);
    float c = x * 255.0f;
    const float slope = logf(20.0f) / 20.0f;
    return c < 20.0f ? slope * c : logf(c + 1e-9f);
}
__device__ __forceinline__ float bright_log_grad(float x, int linlog) {
    if (!linlog) return 1.0f / (x + 1e-9f);
    float c = x * 255.0f;
    const float slope = logf(20.0f) / 20.0f;
    return c < 20.0f ? slope * 255.0f : 255.0f / (c + 1e-9f);
}
__device__ __forceinline__ float to_gray(const float* p, int C) {
    if (C == 1) return p[0];
    return (p[0] * 0.299f + p[1] * 0.587f) + p[2] * 0.114f;
}

__device__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = LT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float event_diff(const float* img, int i, int R, int C, int linlog) {
    float a = to_gray(img + (int64_t)i * C, C);
    float b = to_gray(img + (int64_t)(R + i) * C, C);
    return bright_log(b, linlog) - bright_log(a, linlog);
}

__device__ __forceinline__ float blur_value(const float* img, int r, int c, int R, int C, int P) {
    float s = 0.f;
    for (int j = 0; j < P; ++j) s += img[((int64_t)j * R + r) * C + c];
    return s / (float)P;
}

__global__ void loss_stats_kernel(BenerfLossCfg cfg, const float* __restrict__ rgb_evt,
                                  const float* __restrict__ rgb0_evt, const float* __restrict__ target_acc,
                                  const float* __restrict__ rgb_rgb, const float* __restrict__ rgb0_rgb,
                                  const float* __restrict__ target_rgb, double* __restrict__ stats) {
    __shared__ double red[LT];
    const int C = cfg.channels, Re = cfg.n_evt_pix, Rr = cfg.n_rgb_pix, P = cfg.n_poses;
    double a[9];
    for (int q = 0; q < 9; ++q) a[q] = 0.0;
    const double thr = cfg.event_threshold > 0.f ? (double)cfg.event_threshold : 1.0;
    if (rgb_evt) {
        for (int i = threadIdx.x; i < Re; i += LT) {
            double t = (double)target_acc[i] * thr;
            double df = (double)event_diff(rgb_evt, i, Re, C, cfg.linlog);
            double dc = (double)event_diff(rgb0_evt, i, Re, C, cfg.linlog);
            a[0] += df * df;
            a[1] += df * t;
            a[2] += (df - t) * (df - t);
            a[3] += dc * dc;
            a[4] += dc * t;
            a[5] += (dc - t) * (dc - t);
            a[6] += t * t;
        }
    }
    if (rgb_rgb) {
        for (int e = threadIdx.x; e < Rr * C; e += LT) {
            int r = e / C, c = e % C;
            float t = target_rgb[e];
            float bf = blur_value(rgb_rgb, r, c, Rr, C, P) - t;
            float bc = blur_value(rgb0_rgb, r, c, Rr, C, P) - t;
            a[7] += (double)(bf * bf);
            a[8] += (double)(bc * bc);
        }
    }
    for (int q = 0; q < 9; ++q) {
        double s = block_sum(a[q], red);
        if (threadIdx.x == 0) stats[q] = s;
    }
    if (threadIdx.x >= 9 && threadIdx.x < BENERF_LOSS_NSTATS) stats[threadIdx.x] = 0.0;
}

__global__ void loss_grads_kernel(BenerfLossCfg cfg, const double* __restrict__ stats,
                                  const float* __restrict__ rgb_evt, const float* __restrict__ rgb0_evt,
                                  const float* __restrict__ target_acc, const float* __restrict__ rgb_rgb,
                                  const float* __restrict__ rgb0_rgb, const float* __restrict__ target_rgb,
                                  float* __restrict__ losses, float* __restrict__ d_rgb_evt,
                                  float* __restrict__ d_rgb0_evt, float* __restrict__ d_rgb_rgb,
                                  float* __restrict__ d_rgb0_rgb) {
    const int C = cfg.channels, Re = cfg.n_evt_pix, Rr = cfg.n_rgb_pix, P = cfg.n_poses;
    const double Rg = (double)cfg.n_evt_pix_global, Rrg = (double)cfg.n_rgb_pix_global * C;
    const bool syn = cfg.event_threshold > 0.f;
    const double thr = syn ? (double)cfg.event_threshold : 1.0;
    const double coeff = (double)cfg.event_coeff;
    const double s_tt = stats ? stats[6] : 0.0;      // stats == NULL: gradient-only call for mean-squared losses (launcher checks)
    const double n_t = sqrt(s_tt), sc_t = n_t + 1e-9;
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    int nth = gridDim.x * blockDim.x;

    if (tid == 0 && losses) {
        double ef = 0, ec = 0, rf = 0, rc = 0;
        if (rgb_evt) {
            if (syn) {
                ef = coeff * stats[2] / Rg;
                ec = coeff * stats[5] / Rg;
            } else {
                for (int x = 0; x < 2; ++x) {
                    double s_dd = stats[3 * x], s_dt = stats[3 * x + 1];
                    double sc_d = sqrt(s_dd) + 1e-9;
                    double v = coeff * (s_dd / (sc_d * sc_d) - 2.0 * s_dt / (sc_d * sc_t) + s_tt / (sc_t * sc_t)) / Rg;
                    if (x == 0) ef = v; else ec = v;
                }
            }
        }
        if (rgb_rgb) {
            rf = (double)cfg.rgb_coeff * stats[7] / Rrg;
            rc = (double)cfg.rgb_coeff * stats[8] / Rrg;
        }
        losses[0] = (float)((ec + ef) + (rf + rc));
        losses[1] = (float)(ec + ef);
        losses[2] = (float)ef;
        losses[3] = (float)ec;
        losses[4] = (float)(rf + rc);
        losses[5] = (float)rf;
        losses[6] = (float)rc;
        losses[7] = 0.f;
    }

    if (rgb_evt) {
        for (int w = tid; w < 2 * Re; w += nth) {
            int x = w / Re, i = w % Re;   // x: 0 fine, 1 coarse
            const float* img = x == 0 ? rgb_evt : rgb0_evt;
            float* dimg = x == 0 ? d_rgb_evt : d_rgb0_evt;
            if (!dimg) continue;
            double t = (double)target_acc[i] * thr;
            double d = (double)event_diff(img, i, Re, C, cfg.linlog);
            double g;
            if (syn) {
                g = coeff * 2.0 * (d - t) / Rg;
            } else {
                double s_dd = stats[3 * x], s_dt = stats[3 * x + 1];
                double n_d = sqrt(s_dd), sc_d = n_d + 1e-9;
                double gi = 2.0 * coeff * (d / sc_d - t / sc_t) / Rg;
                double sum_gd = 2.0 * coeff * (s_dd / sc_d - s_dt / sc_t) / Rg;
                g = gi / sc_d - (n_d > 0.0 ? d / (sc_d * sc_d * n_d) * sum_gd : 0.0);
            }
            const float* pa = img + (int64_t)i * C;
            const float* pb = img + (int64_t)(Re + i) * C;
            float ga = bright_log_grad(to_gray(pa, C), cfg.linlog);
            float gb = bright_log_grad(to_gray(pb, C), cfg.linlog);
            const float wts[3] = {0.299f, 0.587f, 0.114f};
            for (int c = 0; c < C; ++c) {
                float wc = C == 1 ? 1.0f : wts[c];
                dimg[(int64_t)i * C + c] = (float)(-g * (double)ga) * wc;
                dimg[(int64_t)(Re + i) * C + c] = (float)(g * (double)gb) * wc;
            }
        }
    }
    if (rgb_rgb) {
        for (int w = tid; w < 2 * Rr * C; w += nth) {
            int x = w / (Rr * C), e = w % (Rr * C);
            const float* img = x == 0 ? rgb_rgb : rgb0_rgb;
            float* dimg = x == 0 ? d_rgb_rgb : d_rgb0_rgb;
            if (!dimg) continue;
            int r = e / C, c = e % C;
            float bl = blur_value(img, r, c, Rr, C, P);
            float g = (float)((double)cfg.rgb_coeff * 2.0 * (double)(bl - target_rgb[e]) / Rrg / (double)P);
            for (int j = 0; j < P; ++j) dimg[((int64_t)j * Rr + r) * C + c] = g;
        }
    }
}

}  // namespace

static int check_cfg(const BenerfLossCfg* cfg, const char* who) {
    BENERF_REQUIRE(cfg, "%s: null cfg", who);
    BENERF_REQUIRE(cfg->channels == 1 || cfg->channels == 3, "%s: channels must be 1 or 3", who);
    BENERF_REQUIRE(cfg->n_evt_pix >= 0 && cfg->n_rgb_pix >= 0 && cfg->n_poses >= 1, "%s: bad sizes", who);
    BENERF_REQUIRE(cfg->n_evt_pix_global >= cfg->n_evt_pix && cfg->n_rgb_pix_global >= cfg->n_rgb_pix,
                   "%s: global batch smaller than local", who);
    return BENERF_OK;
}

extern "C" int benerf_loss_stats(const BenerfLossCfg* cfg, const float* rgb_evt, const float* rgb0_evt,
                                 const float* target_acc, const float* rgb_rgb, const float* rgb0_rgb,
                                 const float* target_rgb, double* stats, benerf_stream_t stream) {
    int rc = check_cfg(cfg, "loss_stats");
    if (rc) return rc;
    BENERF_REQUIRE(stats, "loss_stats: null stats");
    BENERF_REQUIRE(!rgb_evt || (rgb0_evt && target_acc), "loss_stats: event inputs incomplete");
    BENERF_REQUIRE(!rgb_rgb || (rgb0_rgb && target_rgb), "loss_stats: rgb inputs incomplete");
    hipLaunchKernelGGL(loss_stats_kernel, dim3(1), dim3(LT), 0, as_stream(stream), *cfg, rgb_evt, rgb0_evt, target_acc,
                       rgb_rgb, rgb0_rgb, target_rgb, stats);
    BENERF_LAUNCH_CHECK("loss_stats");
    return BENERF_OK;
}

extern "C" int benerf_loss_grads(const BenerfLossCfg* cfg, const double* stats, const float* rgb_evt,
                                 const float* rgb0_evt, const float* target_acc, const float* rgb_rgb,
                                 const float* rgb0_rgb, const float* target_rgb, float* losses, float* d_rgb_evt,
                                 float* d_rgb0_evt, float* d_rgb_rgb, float* d_rgb0_rgb, benerf_stream_t stream) {
    int rc = check_cfg(cfg, "loss_grads");
    if (rc) return rc;
    BENERF_REQUIRE(stats || (!losses && (!rgb_evt || cfg->event_threshold > 0.f)),
                   "loss_grads: stats may only be NULL for a gradient-only call (losses == NULL) without the L2-normalised event loss");
    BENERF_REQUIRE(!rgb_evt || (rgb0_evt && target_acc), "loss_grads: event inputs incomplete");
    BENERF_REQUIRE(!rgb_rgb || (rgb0_rgb && target_rgb), "loss_grads: rgb inputs incomplete");
    int work = 2 * cfg->n_evt_pix + 2 * cfg->n_rgb_pix * cfg->channels;
    int blocks = (work + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(loss_grads_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), *cfg, stats, rgb_evt, rgb0_evt,
                       target_acc, rgb_rgb, rgb0_rgb, target_rgb, losses, d_rgb_evt, d_rgb0_evt, d_rgb_rgb,
                       d_rgb0_rgb);
    BENERF_LAUNCH_CHECK("loss_grads");
    return BENERF_OK;
}
