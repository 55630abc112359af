// Stand-alone forms of operators that the fused kernels otherwise inline; they back the
// reference's public helper functions one to one (forward only):
//   benerf_pixel_rays  <- run_nerf_helpers.get_specific_rays / get_rays   (run_nerf_helpers.py:13-44)
//   benerf_ndc_rays    <- run_nerf_helpers.ndc_rays                       (run_nerf_helpers.py:46-71)
//   benerf_posenc      <- Embedder.embed                                  (model/embedder.py:9-34)
//   benerf_mse_fwd/bwd <- MSELoss                                         (loss/imgloss.py:3-5)
//   benerf_bright_log_fwd/bwd <- rgb2brightlog                            (utils/math_utils.py:4-23)
//   benerf_rgb2gray_fwd/bwd   <- RGB2Gray                                 (utils/img_utils.py:7-16)
// Built with -ffp-contract=off (separately rounded mul/add like the torch ops).
#include "common.h"

namespace {

__global__ void pixel_rays_kernel(const float* __restrict__ c2w, int pose_stride, const int64_t* __restrict__ pi,
                                  const int64_t* __restrict__ pj, int64_t n, float fx, float fy, float cx, float cy,
                                  float* __restrict__ rays_o, float* __restrict__ rays_d) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* p = c2w + e * pose_stride;
    float dx = ((float)pi[e] - cx) / fx;
    float dy = -((float)pj[e] - cy) / fy;
    float dz = -1.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rays_d[e * 3 + r] = (dx * p[r * 4 + 0] + dy * p[r * 4 + 1]) + dz * p[r * 4 + 2];
        rays_o[e * 3 + r] = p[r * 4 + 3];
    }
}

__global__ void ndc_rays_kernel(int H, int W, float focal, float near, const float* __restrict__ ro,
                                const float* __restrict__ rd, int64_t n, float* __restrict__ oo,
                                float* __restrict__ od) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float o0 = ro[e * 3], o1 = ro[e * 3 + 1], o2 = ro[e * 3 + 2];
    float d0 = rd[e * 3], d1 = rd[e * 3 + 1], d2 = rd[e * 3 + 2];
    float t = -(near + o2) / d2;
    float ox = o0 + t * d0, oy = o1 + t * d1, oz = o2 + t * d2;
    float sw = -1.0f / ((float)W / (2.0f * focal));
    float sh = -1.0f / ((float)H / (2.0f * focal));
    oo[e * 3 + 0] = sw * ox / oz;
    oo[e * 3 + 1] = sh * oy / oz;
    oo[e * 3 + 2] = 1.0f + (2.0f * near) / oz;
    od[e * 3 + 0] = sw * (d0 / d2 - ox / oz);
    od[e * 3 + 1] = sh * (d1 / d2 - oy / oz);
    od[e * 3 + 2] = (-2.0f * near) / oz;
}

__global__ void posenc_kernel(const float* __restrict__ x, int64_t n, int dims, int n_freqs, int include_input,
                              float* __restrict__ out) {
    const int width = (include_input ? dims : 0) + 2 * dims * n_freqs;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * width) return;
    int64_t row = e / width;
    int c = (int)(e % width);
    float v;
    if (include_input && c < dims) {
        v = x[row * dims + c];
    } else {
        int q = c - (include_input ? dims : 0);
        int f = q / (2 * dims), r = q % (2 * dims);
        float a = x[row * dims + (r % dims)] * (float)(1 << f);
        v = r < dims ? sinf(a) : cosf(a);
    }
    out[e] = v;
}

constexpr int MT = 1024;
__global__ void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                               float* __restrict__ out) {
    __shared__ double red[MT];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += MT) {
        float d = a[i] - b[i];
        s += (double)(d * d);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = MT / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)n);
}

__global__ void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                               const float* __restrict__ g, float* __restrict__ da, float* __restrict__ db) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = g[0] * (2.0f * (a[i] - b[i]) / (float)n);
    if (da) da[i] = v;
    if (db) db[i] = -v;
}

// rgb2brightlog / RGB2Gray of the reference's loss lines as single launches (the same curves as loss.hip's bright_log / to_gray)
__device__ __forceinline__ float h_bright_log(float x, int linlog) {
    if (!linlog) return logf(x + 1e-9f);
    const float c = x * 255.0f;
    const float slope = logf(20.0f) / 20.0f;
    return c < 20.0f ? slope * c : logf(c + 1e-9f);
}
__device__ __forceinline__ float h_bright_log_grad(float x, int linlog) {
    if (!linlog) return 1.0f / (x + 1e-9f);
    const float c = x * 255.0f;
    const float slope = logf(20.0f) / 20.0f;
    return c < 20.0f ? slope * 255.0f : 255.0f / (c + 1e-9f);
}
__global__ void bright_log_fwd_kernel(const float* __restrict__ x, int64_t n, int linlog, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = h_bright_log(x[i], linlog);
}
__global__ void bright_log_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, int64_t n, int linlog, float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = g[i] * h_bright_log_grad(x[i], linlog);
}
__global__ void rgb2gray_fwd_kernel(const float* __restrict__ rgb, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (rgb[3 * i] * 0.299f + rgb[3 * i + 1] * 0.587f) + rgb[3 * i + 2] * 0.114f;
}
__global__ void rgb2gray_bwd_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ d_rgb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = g[i];
        d_rgb[3 * i] = v * 0.299f;
        d_rgb[3 * i + 1] = v * 0.587f;
        d_rgb[3 * i + 2] = v * 0.114f;
    }
}

}  // namespace

extern "C" int benerf_bright_log_fwd(const float* x, int64_t n, int linlog, float* out, benerf_stream_t stream) {
    BENERF_REQUIRE(x && out && n >= 0, "bright_log_fwd: bad args");
    if (n == 0) return BENERF_OK;
    hipLaunchKernelGGL(bright_log_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, n, linlog, out);
    BENERF_LAUNCH_CHECK("bright_log_fwd");
    return BENERF_OK;
}

extern "C" int benerf_bright_log_bwd(const float* x, const float* grad, int64_t n, int linlog, float* d_x, benerf_stream_t stream) {
    BENERF_REQUIRE(x && grad && d_x && n >= 0, "bright_log_bwd: bad args");
    if (n == 0) return BENERF_OK;
    hipLaunchKernelGGL(bright_log_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, grad, n, linlog, d_x);
    BENERF_LAUNCH_CHECK("bright_log_bwd");
    return BENERF_OK;
}

extern "C" int benerf_rgb2gray_fwd(const float* rgb, int64_t n, float* out, benerf_stream_t stream) {
    BENERF_REQUIRE(rgb && out && n >= 0, "rgb2gray_fwd: bad args");
    if (n == 0) return BENERF_OK;
    hipLaunchKernelGGL(rgb2gray_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), rgb, n, out);
    BENERF_LAUNCH_CHECK("rgb2gray_fwd");
    return BENERF_OK;
}

extern "C" int benerf_rgb2gray_bwd(const float* grad, int64_t n, float* d_rgb, benerf_stream_t stream) {
    BENERF_REQUIRE(grad && d_rgb && n >= 0, "rgb2gray_bwd: bad args");
    if (n == 0) return BENERF_OK;
    hipLaunchKernelGGL(rgb2gray_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), grad, n, d_rgb);
    BENERF_LAUNCH_CHECK("rgb2gray_bwd");
    return BENERF_OK;
}

extern "C" int benerf_pixel_rays(const float* c2w, int per_ray_pose, const int64_t* i, const int64_t* j, int64_t n,
                                 float fx, float fy, float cx, float cy, float* rays_o, float* rays_d,
                                 benerf_stream_t stream) {
    BENERF_REQUIRE(c2w && i && j && rays_o && rays_d && n >= 0, "pixel_rays: bad args");
    if (n == 0) return BENERF_OK;
    int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(pixel_rays_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), c2w, per_ray_pose ? 12 : 0, i, j,
                       n, fx, fy, cx, cy, rays_o, rays_d);
    BENERF_LAUNCH_CHECK("pixel_rays");
    return BENERF_OK;
}

extern "C" int benerf_ndc_rays(int H, int W, float focal, float near, const float* rays_o, const float* rays_d,
                               int64_t n, float* out_o, float* out_d, benerf_stream_t stream) {
    BENERF_REQUIRE(rays_o && rays_d && out_o && out_d && n >= 0 && H > 0 && W > 0, "ndc_rays: bad args");
    if (n == 0) return BENERF_OK;
    int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(ndc_rays_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), H, W, focal, near, rays_o, rays_d,
                       n, out_o, out_d);
    BENERF_LAUNCH_CHECK("ndc_rays");
    return BENERF_OK;
}

extern "C" int benerf_posenc(const float* x, int64_t n, int dims, int n_freqs, int include_input, float* out,
                             benerf_stream_t stream) {
    BENERF_REQUIRE(x && out && n >= 0 && dims > 0 && n_freqs >= 0 && n_freqs <= 30, "posenc: bad args");
    if (n == 0) return BENERF_OK;
    int64_t total = n * ((include_input ? dims : 0) + 2 * dims * n_freqs);
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(posenc_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, n, dims, n_freqs, include_input,
                       out);
    BENERF_LAUNCH_CHECK("posenc");
    return BENERF_OK;
}

extern "C" int benerf_mse_fwd(const float* a, const float* b, int64_t n, float* out, benerf_stream_t stream) {
    BENERF_REQUIRE(a && b && out && n > 0, "mse_fwd: bad args");
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(1), dim3(MT), 0, as_stream(stream), a, b, n, out);
    BENERF_LAUNCH_CHECK("mse_fwd");
    return BENERF_OK;
}

extern "C" int benerf_mse_bwd(const float* a, const float* b, int64_t n, const float* grad_out, float* d_a, float* d_b,
                              benerf_stream_t stream) {
    BENERF_REQUIRE(a && b && grad_out && n > 0, "mse_bwd: bad args");
    int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a, b, n, grad_out, d_a, d_b);
    BENERF_LAUNCH_CHECK("mse_bwd");
    return BENERF_OK;
}
