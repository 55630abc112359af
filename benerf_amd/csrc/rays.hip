// K2: ray generation, view directions, LLFF NDC, stratified depths, ray-gradient reduce.
//
// Forward follows run_nerf_helpers.py:35-44 (pinhole dirs, rays_d = R dirs, rays_o = t),
// model/nerf.py:272-275 (viewdirs from PRE-NDC rays_d), run_nerf_helpers.py:46-71 (NDC,
// near = 1, focal = K[0][0]) and model/nerf.py:297-307 (stratified z).
// Backward: a pose enters a ray only through rd = R dirs and ro = t (6 numbers), so the same templated code runs on
// forward-mode duals with one tangent per component of (rd, ro) - 6 passes per ray - and the pose gradient is the outer
// product d_pose[r][0..2] = g_rd[r] * dirs, d_pose[r][3] = g_ro[r]; grid over (pixel chunk, pose), wave shuffles + one
// LDS hop per block, chunks summed in fixed order -> deterministic d_poses.
//
// Built with -ffp-contract=off so mul/add stay separately rounded like the torch ops.
#include "common.h"

namespace {

struct Dual {
    float v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    float q = a.v / b.v;
    return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator+(float a, Dual b) { return {a + b.v, b.d}; }
__device__ __forceinline__ Dual operator/(float a, Dual b) {
    float q = a / b.v;
    return {q, -q * b.d / b.v};
}
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ Dual t_sqrt(Dual x) {
    float s = sqrtf(x.v);
    return {s, s > 0.f ? 0.5f * x.d / s : 0.f};
}

struct Cam {
    int H, W;
    float fx, fy, cx, cy;
    int ndc;
};

// pose: 12 entries row-major [3][4]; pixel (i = column, j = row).
// pixel -> (column, row) coordinates of the ray: the integer pixel itself, or - TUM_VIE - the entry of the undistortion
// look-up table remap [H, W, 2] = (x, y) the reference gathers with `rect = remap[j, i]` (model/nerf.py:247-250,
// run_nerf_helpers.py:17-23)
__device__ __forceinline__ void pixel_coords(int64_t idx, const Cam& c, const float* __restrict__ remap, float& fi, float& fj) {
    if (remap) {
        fi = remap[idx * 2 + 0];
        fj = remap[idx * 2 + 1];
    } else {
        fj = (float)(int)(idx / c.W);               // model/nerf.py:244-245
        fi = (float)(int)(idx % c.W);
    }
}

// everything behind the camera-to-world product: rd = R dirs (un-normalised), ro = t  ->  o, d (NDC'd when c.ndc), viewdirs
template <class T>
__device__ __forceinline__ void ray_from_rd(const T rd[3], const T ro[3], const Cam& c, T o[3], T d[3], T vd[3]) {
    T nrm = t_sqrt((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);   // model/nerf.py:272-275
    vd[0] = rd[0] / nrm;
    vd[1] = rd[1] / nrm;
    vd[2] = rd[2] / nrm;
    if (c.ndc) {                                   // run_nerf_helpers.py:46-71, near = 1
        const float near = 1.0f;
        T t = -(near + ro[2]) / rd[2];
        T ox = ro[0] + t * rd[0], oy = ro[1] + t * rd[1], oz = ro[2] + t * rd[2];
        float sw = -1.0f / ((float)c.W / (2.0f * c.fx));
        float sh = -1.0f / ((float)c.H / (2.0f * c.fx));
        o[0] = sw * ox / oz;
        o[1] = sh * oy / oz;
        o[2] = 1.0f + (2.0f * near) / oz;
        d[0] = sw * (rd[0] / rd[2] - ox / oz);
        d[1] = sh * (rd[1] / rd[2] - oy / oz);
        d[2] = (-2.0f * near) / oz;
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o[r] = ro[r];
            d[r] = rd[r];
        }
    }
}

__device__ __forceinline__ void pixel_dirs(float fi, float fj, const Cam& c, float dirs[3]) {
    dirs[0] = (fi - c.cx) / c.fx;          // run_nerf_helpers.py:36-38
    dirs[1] = -(fj - c.cy) / c.fy;
    dirs[2] = -1.0f;
}

template <class T>
__device__ __forceinline__ void ray_from_pose(const T pose[12], float fi, float fj, const Cam& c, T o[3], T d[3], T vd[3]) {
    float dirs[3];
    pixel_dirs(fi, fj, c, dirs);
    T rd[3], ro[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rd[r] = (dirs[0] * pose[r * 4 + 0] + dirs[1] * pose[r * 4 + 1]) + dirs[2] * pose[r * 4 + 2];
        ro[r] = pose[r * 4 + 3];
    }
    ray_from_rd(rd, ro, c, o, d, vd);
}

__global__ void rays_fwd_kernel(const float* __restrict__ poses, const int64_t* __restrict__ ray_idx, int n_poses,
                                int n_pix, Cam cam, const float* __restrict__ remap, float* __restrict__ rays_o,
                                float* __restrict__ rays_d, float* __restrict__ viewdirs) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t N = (int64_t)n_poses * n_pix;
    if (n >= N) return;
    int p = (int)(n / n_pix);
    float i, j;
    pixel_coords(ray_idx[n % n_pix], cam, remap, i, j);
    float pose[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) pose[e] = poses[p * 12 + e];
    float o[3], d[3], vd[3];
    ray_from_pose(pose, i, j, cam, o, d, vd);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rays_o[n * 3 + r] = o[r];
        rays_d[n * 3 + r] = d[r];
        viewdirs[n * 3 + r] = vd[r];
    }
}

// grid (pixel chunks of 256, poses), one pixel per thread.  out: d_poses itself when there is one chunk, else the chunk
// partials [n_poses][n_chunks][12] that rays_bwd_reduce_kernel sums in chunk order.
constexpr int RB_T = 256;
__global__ __launch_bounds__(RB_T) void rays_bwd_kernel(const float* __restrict__ poses, const int64_t* __restrict__ ray_idx, int n_pix,
                                                         Cam cam, const float* __restrict__ remap, const float* __restrict__ g_o,
                                                         const float* __restrict__ g_d, const float* __restrict__ g_v,
                                                         float* __restrict__ out) {
    __shared__ float red[RB_T / 64][12];
    const int p = blockIdx.y, r = blockIdx.x * RB_T + threadIdx.x;
    float acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 0.f;
    if (r < n_pix) {
        const int64_t n = (int64_t)p * n_pix + r;
        float fi, fj, dirs[3], rdv[3], rov[3];
        pixel_coords(ray_idx[r], cam, remap, fi, fj);
        pixel_dirs(fi, fj, cam, dirs);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* row = poses + p * 12 + c * 4;
            rdv[c] = (dirs[0] * row[0] + dirs[1] * row[1]) + dirs[2] * row[2];
            rov[c] = row[3];
        }
        float go[3], gd[3], gv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            go[c] = g_o ? g_o[n * 3 + c] : 0.f;
            gd[c] = g_d ? g_d[n * 3 + c] : 0.f;
            gv[c] = g_v ? g_v[n * 3 + c] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 6; ++e) {            // tangent e: rd[e] (e < 3) or ro[e - 3]
            Dual rd[3], ro[3], o[3], d[3], vd[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rd[c] = Dual{rdv[c], e == c ? 1.f : 0.f};
                ro[c] = Dual{rov[c], e == c + 3 ? 1.f : 0.f};
            }
            ray_from_rd(rd, ro, cam, o, d, vd);
            float g = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) g += go[c] * o[c].d + gd[c] * d[c].d + gv[c] * vd[c].d;
            if (e < 3) {
                acc[e * 4 + 0] = dirs[0] * g;
                acc[e * 4 + 1] = dirs[1] * g;
                acc[e * 4 + 2] = dirs[2] * g;
            } else {
                acc[(e - 3) * 4 + 3] = g;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        float v = acc[e];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        acc[e] = v;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < 12; ++e) red[threadIdx.x >> 6][e] = acc[e];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        out[((int64_t)p * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = v;
    }
}

__global__ void rays_bwd_reduce_kernel(const float* __restrict__ part, int n_chunks, int n_out, float* __restrict__ d_poses) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (pose, entry)
    if (i >= n_out) return;
    const int p = i / 12, e = i - p * 12;
    float v = 0.f;
    for (int c = 0; c < n_chunks; ++c) v += part[((int64_t)p * n_chunks + c) * 12 + e];
    d_poses[i] = v;
}

// torch.linspace(0,1,S)[i]
__device__ __forceinline__ float lin01(int S, int i) {
    if (S <= 1) return 0.f;
    float step = 1.0f / (float)(S - 1);
    return (i < S / 2) ? 0.0f + step * (float)i : 1.0f - step * (float)(S - 1 - i);
}

__global__ void stratified_z_kernel(int64_t total, int S, float near, float far, const float* __restrict__ t_rand,
                                    uint64_t seed, uint64_t offset, float* __restrict__ z) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int s = (int)(e % S);
    // z_vals = near*(1-t) + far*t                          model/nerf.py:297-299
    float t = lin01(S, s);
    float zc = near * (1.0f - t) + far * t;
    float lower = zc, upper = zc;
    if (s > 0) {
        float tp = lin01(S, s - 1);
        lower = 0.5f * (zc + (near * (1.0f - tp) + far * tp));
    }
    if (s < S - 1) {
        float tn = lin01(S, s + 1);
        upper = 0.5f * ((near * (1.0f - tn) + far * tn) + zc);
    }
    float r = t_rand ? t_rand[e] : philox_uniform(seed, offset, (uint64_t)e);
    z[e] = lower + (upper - lower) * r;                    // model/nerf.py:301-307
}

// one wave per ray: sums over samples in a fixed (lane-strided then butterfly) order.
__global__ void ray_grad_reduce_kernel(int n_rays, int S, const float* __restrict__ z, const float* __restrict__ d_pts,
                                       const float* __restrict__ d_vd, int accumulate, float* __restrict__ d_o,
                                       float* __restrict__ d_d, float* __restrict__ d_v) {
    int ray = blockIdx.x * (blockDim.x / 64) + (threadIdx.x / 64);
    int lane = threadIdx.x & 63;
    if (ray >= n_rays) return;
    float a[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) a[q] = 0.f;
    for (int s = lane; s < S; s += 64) {
        int64_t m = (int64_t)ray * S + s;
        float zz = z[m];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float g = d_pts[m * 3 + c];
            a[c] += g;
            a[3 + c] += zz * g;
            a[6 + c] += d_vd ? d_vd[m * 3 + c] : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        float v = a[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        a[q] = v;
    }
    if (lane < 3) {
        int c = lane;
        float vo = a[c], vdd = a[3 + c], vv = a[6 + c];
        // accumulate: 0 overwrite, 1 add into all three, 2 add into d_rays_d only (it already holds the compositing
        // backward's part: the first of a step's two calls needs no zeroed d_rays_o / d_viewdirs then)
        const bool add_od = accumulate == 1, add_d = accumulate != 0;
        if (d_o) d_o[ray * 3 + c] = add_od ? d_o[ray * 3 + c] + vo : vo;
        if (d_d) d_d[ray * 3 + c] = add_d ? d_d[ray * 3 + c] + vdd : vdd;
        if (d_v) d_v[ray * 3 + c] = add_od ? d_v[ray * 3 + c] + vv : vv;
    }
}

}  // namespace

extern "C" int benerf_rays_fwd(const float* poses, const int64_t* ray_idx, int n_poses, int n_pix, int H, int W,
                               float fx, float fy, float cx, float cy, int ndc, const float* remap, float* rays_o,
                               float* rays_d, float* viewdirs, benerf_stream_t stream) {
    BENERF_REQUIRE(poses && ray_idx && rays_o && rays_d && viewdirs, "rays_fwd: null pointer");
    BENERF_REQUIRE(n_poses > 0 && n_pix > 0 && H > 0 && W > 0, "rays_fwd: bad sizes");
    Cam cam{H, W, fx, fy, cx, cy, ndc};
    int64_t N = (int64_t)n_poses * n_pix;
    int threads = 256;
    int blocks = (int)((N + threads - 1) / threads);
    hipLaunchKernelGGL(rays_fwd_kernel, dim3(blocks), dim3(threads), 0, as_stream(stream), poses, ray_idx, n_poses,
                       n_pix, cam, remap, rays_o, rays_d, viewdirs);
    BENERF_LAUNCH_CHECK("rays_fwd");
    return BENERF_OK;
}

extern "C" size_t benerf_rays_bwd_workspace_floats(int n_poses, int n_pix) {
    const int chunks = (n_pix + RB_T - 1) / RB_T;
    return chunks > 1 ? (size_t)n_poses * chunks * 12 : 0;
}

extern "C" int benerf_rays_bwd(const float* poses, const int64_t* ray_idx, int n_poses, int n_pix, int H, int W,
                               float fx, float fy, float cx, float cy, int ndc, const float* remap,
                               const float* d_rays_o, const float* d_rays_d, const float* d_viewdirs, float* d_poses,
                               float* workspace, size_t workspace_floats, benerf_stream_t stream) {
    BENERF_REQUIRE(poses && ray_idx && d_poses, "rays_bwd: null pointer");
    BENERF_REQUIRE(n_poses > 0 && n_pix > 0, "rays_bwd: bad sizes");
    Cam cam{H, W, fx, fy, cx, cy, ndc};
    const int chunks = (n_pix + RB_T - 1) / RB_T;
    const size_t need = benerf_rays_bwd_workspace_floats(n_poses, n_pix);
    if (need > 0 && (!workspace || workspace_floats < need)) {
        benerf_set_error("rays_bwd: workspace of %zu floats needed, %zu given", need, workspace ? workspace_floats : (size_t)0);
        return BENERF_EWORKSPACE;
    }
    hipLaunchKernelGGL(rays_bwd_kernel, dim3(chunks, n_poses), dim3(RB_T), 0, as_stream(stream), poses, ray_idx, n_pix, cam, remap,
                       d_rays_o, d_rays_d, d_viewdirs, chunks > 1 ? workspace : d_poses);
    BENERF_LAUNCH_CHECK("rays_bwd");
    if (chunks > 1) {
        hipLaunchKernelGGL(rays_bwd_reduce_kernel, dim3((n_poses * 12 + 63) / 64), dim3(64), 0, as_stream(stream), workspace, chunks,
                           n_poses * 12, d_poses);
        BENERF_LAUNCH_CHECK("rays_bwd(reduce)");
    }
    return BENERF_OK;
}

extern "C" int benerf_stratified_z(int n_rays, int n_samples, float near, float far, const float* t_rand,
                                   uint64_t seed, uint64_t offset, float* z, benerf_stream_t stream) {
    BENERF_REQUIRE(z && n_rays > 0 && n_samples > 0, "stratified_z: bad args");
    int64_t total = (int64_t)n_rays * n_samples;
    int threads = 256;
    int blocks = (int)((total + threads - 1) / threads);
    hipLaunchKernelGGL(stratified_z_kernel, dim3(blocks), dim3(threads), 0, as_stream(stream), total, n_samples, near,
                       far, t_rand, seed, offset, z);
    BENERF_LAUNCH_CHECK("stratified_z");
    return BENERF_OK;
}

extern "C" int benerf_ray_grad_reduce(int n_rays, int n_samples, const float* z, const float* d_pts,
                                      const float* d_vdir_pts, int accumulate, float* d_rays_o, float* d_rays_d,
                                      float* d_viewdirs, benerf_stream_t stream) {
    BENERF_REQUIRE(z && d_pts && n_rays > 0 && n_samples > 0, "ray_grad_reduce: bad args");
    int waves = 4;
    int blocks = (n_rays + waves - 1) / waves;
    hipLaunchKernelGGL(ray_grad_reduce_kernel, dim3(blocks), dim3(64 * waves), 0, as_stream(stream), n_rays, n_samples,
                       z, d_pts, d_vdir_pts, accumulate, d_rays_o, d_rays_d, d_viewdirs);
    BENERF_LAUNCH_CHECK("ray_grad_reduce");
    return BENERF_OK;
}
