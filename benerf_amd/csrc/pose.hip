// K1: SE(3) trajectory interpolation (cubic B-spline / linear), forward + backward.
//
// Follows spline.py:16-26 (se3 -> q,t with 11-term Taylor B,C), :79-100 (exp), :167-192
// (log, plain arctan), :130-148 (quaternion product matrix / conjugate), :111-118 (q -> R),
// :247-303 (cumulative cubic B-spline), :305-331 (linear) and model/optimize.py:58-111
// (knots + transform in se(3), torch.linspace of the query times).
//
// One thread per pose, everything in registers.  The backward runs the SAME templated
// code on forward-mode dual numbers (one tangent per knot coefficient): 24 tangents x P
// poses, then a fixed-order reduction, so results are run-to-run deterministic.
#include "common.h"

namespace {

struct Dual {
    float v, d;
};
__device__ __forceinline__ Dual mk(float v, float d = 0.f) { return Dual{v, d}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    float q = a.v / b.v;
    return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator+(Dual a, float b) { return {a.v + b, a.d}; }
__device__ __forceinline__ Dual operator+(float a, Dual b) { return {a + b.v, b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return {a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return {a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, float b) { return {a.v / b, a.d / b}; }
__device__ __forceinline__ Dual operator/(float a, Dual b) {
    float q = a / b.v;
    return {q, -q * b.d / b.v};
}

__device__ __forceinline__ float val(float x) { return x; }
__device__ __forceinline__ float val(Dual x) { return x.v; }
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
// torch's norm/sqrt backward at 0 yields 0 for the cases on this path (masked_fill in
// norm_backward); keep the tangent finite.
__device__ __forceinline__ Dual t_sqrt(Dual x) {
    float s = sqrtf(x.v);
    return {s, s > 0.f ? 0.5f * x.d / s : 0.f};
}
__device__ __forceinline__ float t_sin(float x) { return sinf(x); }
__device__ __forceinline__ Dual t_sin(Dual x) { return {sinf(x.v), cosf(x.v) * x.d}; }
__device__ __forceinline__ float t_cos(float x) { return cosf(x); }
__device__ __forceinline__ Dual t_cos(Dual x) { return {cosf(x.v), -sinf(x.v) * x.d}; }
__device__ __forceinline__ float t_atan(float x) { return atanf(x); }
__device__ __forceinline__ Dual t_atan(Dual x) { return {atanf(x.v), x.d / (1.f + x.v * x.v)}; }
// x ** n for integer n >= 0 as torch.pow does it (n==0 -> 1 with zero gradient)
__device__ __forceinline__ float t_powi(float x, int n) {
    if (n == 0) return 1.f;
    if (n == 2) return x * x;
    return powf(x, (float)n);
}
__device__ __forceinline__ Dual t_powi(Dual x, int n) {
    if (n == 0) return {1.f, 0.f};
    if (n == 2) return {x.v * x.v, 2.f * x.v * x.d};
    return {powf(x.v, (float)n), (float)n * powf(x.v, (float)(n - 1)) * x.d};
}
template <class T>
__device__ __forceinline__ T lift(float x);
template <>
__device__ __forceinline__ float lift<float>(float x) { return x; }
template <>
__device__ __forceinline__ Dual lift<Dual>(float x) { return {x, 0.f}; }

// sum_i (-1)^i th^(2i) / prod (2j+a)(2j+a+1)            spline.py:46-62
// th^(2i) by repeated multiplication with th^2 (the reference's th ** (2i) through pow() differs from it by an ulp of terms
// that are themselves <= th^2 / 6 of the sum; powf made this series, evaluated 8 times per pose, the whole cost of K1)
template <class T>
__device__ T taylor_series(T th, int a) {
    T acc = lift<T>(0.f);
    const T th2 = th * th;
    T pw = lift<T>(1.f);
    double denom = 1.0;
#pragma unroll
    for (int i = 0; i <= 10; ++i) {
        denom *= (double)((2 * i + a) * (2 * i + a + 1));
        T term = pw / (float)denom;
        acc = (i & 1) ? acc - term : acc + term;
        pw = pw * th2;
    }
    return acc;
}

// rotation vector -> quaternion xyzw; th = half angle        spline.py:79-100
template <class T>
__device__ void rotvec_to_quat(const T r[3], T q[4]) {
    T th = 0.5f * t_sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (val(th) < 1e-9f) {
        T th2 = th * th, th4 = th2 * th2;
        T s = 0.5f - (1.0f / 12.0f) * th2 - (1.0f / 240.0f) * th4;
        q[0] = s * r[0];
        q[1] = s * r[1];
        q[2] = s * r[2];
        q[3] = 1.0f - 0.5f * th2 + (1.0f / 24.0f) * th4;
    } else {
        T lam = t_sin(th) / (2.0f * th);
        q[0] = lam * r[0];
        q[1] = lam * r[1];
        q[2] = lam * r[2];
        q[3] = t_cos(th);
    }
}

// quaternion -> rotation vector, plain arctan              spline.py:167-192
template <class T>
__device__ void quat_to_rotvec(const T q[4], T r[3]) {
    const float PI = 3.14159265358979323846f;
    T th = t_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    T w = q[3];
    T lam;
    if (fabsf(val(w)) < 1e-10f) {
        lam = (val(w) < 0.f) ? (-PI) / th : PI / th;
    } else if (val(th) < 1e-20f) {
        lam = 2.0f / w - (2.0f / 3.0f) * (th * th) / (w * w * w);
    } else {
        lam = 2.0f * t_atan(th / w) / th;
    }
    r[0] = lam * q[0];
    r[1] = lam * q[1];
    r[2] = lam * q[2];
}

// a (x) b with the left-product matrix of spline.py:130-138
template <class T>
__device__ void quat_mul(const T a[4], const T b[4], T o[4]) {
    T x = a[0], y = a[1], z = a[2], w = a[3];
    o[0] = w * b[0] - z * b[1] + y * b[2] + x * b[3];
    o[1] = z * b[0] + w * b[1] - x * b[2] + y * b[3];
    o[2] = -y * b[0] + x * b[1] + w * b[2] + z * b[3];
    o[3] = -x * b[0] - y * b[1] - z * b[2] + w * b[3];
}
template <class T>
__device__ void quat_conj(const T q[4], T o[4]) {
    o[0] = -q[0];
    o[1] = -q[1];
    o[2] = -q[2];
    o[3] = q[3];
}

// se(3) [w,u] -> q, t = V u, V = I + B wx + (C wx) wx       spline.py:16-26
template <class T>
__device__ void se3_to_qt(const T wu[6], T q[4], T t[3]) {
    T w0 = wu[0], w1 = wu[1], w2 = wu[2];
    T th = t_sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    T B = taylor_series(th, 1);
    T C = taylor_series(th, 2);
    T z = lift<T>(0.f);
    T wx[3][3] = {{z, -w2, w1}, {w2, z, -w0}, {-w1, w0, z}};
    T V[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            T s = (C * wx[i][0]) * wx[0][j] + (C * wx[i][1]) * wx[1][j] + (C * wx[i][2]) * wx[2][j];
            V[i][j] = ((i == j ? 1.0f : 0.0f) + B * wx[i][j]) + s;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = V[i][0] * wu[3] + V[i][1] * wu[4] + V[i][2] * wu[5];
    T r[3] = {w0, w1, w2};
    rotvec_to_quat(r, q);
}

template <class T>
__device__ void quat_to_pose(const T q[4], const T t[3], T out[12]) {
    T b = q[0], c = q[1], d = q[2], a = q[3];   // spline.py:111-118
    out[0] = 1.0f - 2.0f * (c * c + d * d);
    out[1] = 2.0f * (b * c - a * d);
    out[2] = 2.0f * (a * c + b * d);
    out[3] = t[0];
    out[4] = 2.0f * (b * c + a * d);
    out[5] = 1.0f - 2.0f * (b * b + d * d);
    out[6] = 2.0f * (c * d - a * b);
    out[7] = t[1];
    out[8] = 2.0f * (b * d - a * c);
    out[9] = 2.0f * (a * b + c * d);
    out[10] = 1.0f - 2.0f * (b * b + c * c);
    out[11] = t[2];
}

// torch.linspace(t0, t1, n)[i] (float kernel: symmetric two-sided formula)
__device__ __forceinline__ float linspace_at(float t0, float t1, int n, int i) {
    if (n <= 1) return t0;
    float step = (t1 - t0) / (float)(n - 1);
    return (i < n / 2) ? t0 + step * (float)i : t1 - step * (float)(n - 1 - i);
}
__device__ __forceinline__ float nudge(float u) {   // spline.py:249-252
    if (u == 0.f) u = u + 0.000001f;
    if (u == 1.f) u = u - 0.000001f;
    return u;
}

// BEZIER = false: uniform cubic B-spline, spline.py:247-303.  BEZIER = true: cubic Bezier with the same cumulative
// construction - translation by the Bernstein basis C(3,k)(1-u)^(3-k)u^k (bezier.py:7-20), rotation
// q0 (x) exp(b1 log(q0^-1 q1)) (x) exp(b2 log(q1^-1 q2)) (x) exp(b3 log(q2^-1 q3)) with the cumulative Bernstein basis
// b_i = sum_{k>=i} B_k.  The reference's bezier.py:22-74 is an unfinished draft of this (it indexes a size-1 dimension and
// raises IndexError on every call, and would use B_1 for all three increments); this is its evident intent.
template <bool BEZIER, class T>
__device__ void cubic_from_qt(const T q[4][4], const T t[4][3], float u, T out[12]);
template <bool BEZIER, class T>
__device__ void cubic_pose(const T k[4][6], float u, T out[12]) {
    T q[4][4], t[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) se3_to_qt(k[i], q[i], t[i]);
    cubic_from_qt<BEZIER>(q, t, u, out);
}
// the part of the curve evaluation that depends on the query time (the knots' (q, t) do not: the backward evaluates
// them once per launch, not once per pose and tangent)
template <bool BEZIER, class T>
__device__ void cubic_from_qt(const T q[4][4], const T t[4][3], float u, T out[12]) {
    float uu = u * u, uuu = u * u * u;
    const float sixth = 1.0f / 6.0f, half = 0.5f;
    float c0, c1, c2, c3, r1, r2, r3;
    if (BEZIER) {
        const float v = 1.0f - u;
        c0 = v * v * v;
        c1 = 3.0f * v * v * u;
        c2 = 3.0f * v * uu;
        c3 = uuu;
        r1 = c1 + c2 + c3;
        r2 = c2 + c3;
        r3 = c3;
    } else {
        c0 = sixth - half * u + half * uu - sixth * uuu;
        c1 = 4 * sixth - uu + half * uuu;
        c2 = sixth + half * u + half * uu - half * uuu;
        c3 = sixth * uuu;
        r1 = 5 * sixth + half * u - half * uu + sixth * uuu;
        r2 = sixth + half * u + half * uu - 2 * sixth * uuu;
        r3 = sixth * uuu;
    }
    T tr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) tr[i] = c0 * t[0][i] + c1 * t[1][i] + c2 * t[2][i] + c3 * t[3][i];
    T cj[4], d01[4], d12[4], d23[4], rv[3], e0[4], e1[4], e2[4];
    quat_conj(q[0], cj);
    quat_mul(cj, q[1], d01);
    quat_conj(q[1], cj);
    quat_mul(cj, q[2], d12);
    quat_conj(q[2], cj);
    quat_mul(cj, q[3], d23);
    quat_to_rotvec(d01, rv);
    rv[0] = rv[0] * r1; rv[1] = rv[1] * r1; rv[2] = rv[2] * r1;
    rotvec_to_quat(rv, e0);
    quat_to_rotvec(d12, rv);
    rv[0] = rv[0] * r2; rv[1] = rv[1] * r2; rv[2] = rv[2] * r2;
    rotvec_to_quat(rv, e1);
    quat_to_rotvec(d23, rv);
    rv[0] = rv[0] * r3; rv[1] = rv[1] * r3; rv[2] = rv[2] * r3;
    rotvec_to_quat(rv, e2);
    T p1[4], p2[4], qt[4];
    quat_mul(e1, e2, p1);
    quat_mul(e0, p1, p2);
    quat_mul(q[0], p2, qt);
    quat_to_pose(qt, tr, out);
}

template <class T>
__device__ void linear_from_qt(const T qs[4], const T ts[3], const T qe[4], const T te[3], float u, T out[12]);
template <class T>
__device__ void linear_pose(const T k0[6], const T k3[6], float u, T out[12]) {   // spline.py:305-331
    T qs[4], ts[3], qe[4], te[3];
    se3_to_qt(k0, qs, ts);
    se3_to_qt(k3, qe, te);
    linear_from_qt(qs, ts, qe, te, u, out);
}
template <class T>
__device__ void linear_from_qt(const T qs[4], const T ts[3], const T qe[4], const T te[3], float u, T out[12]) {
    T tr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) tr[i] = (1.0f - u) * ts[i] + u * te[i];
    T cj[4], rel[4], rv[3], st[4], qt[4];
    quat_conj(qs, cj);
    quat_mul(cj, qe, rel);
    quat_to_rotvec(rel, rv);
    rv[0] = u * rv[0]; rv[1] = u * rv[1]; rv[2] = u * rv[2];
    rotvec_to_quat(rv, st);
    quat_mul(qs, st, qt);
    quat_to_pose(qt, tr, out);
}

__device__ __forceinline__ void spline_fwd_body(const float* __restrict__ knots, const float* __restrict__ transform,
                                                const float* __restrict__ ts2, int n_poses, int traj, int explicit_ts,
                                                float* __restrict__ poses, int p) {
    if (p >= n_poses) return;
    float k[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) k[i][j] = knots[i * 6 + j] + (transform ? transform[j] : 0.f);
    float u = nudge(explicit_ts ? ts2[p] : linspace_at(ts2[0], ts2[1], n_poses, p));
    float out[12];
    if (traj == 1)
        linear_pose(k[0], k[3], u, out);
    else if (traj == 2)
        cubic_pose<true>(k, u, out);
    else
        cubic_pose<false>(k, u, out);
#pragma unroll
    for (int i = 0; i < 12; ++i) poses[p * 12 + i] = out[i];
}

__global__ void spline_fwd_kernel(const float* __restrict__ knots, const float* __restrict__ transform,
                                  const float* __restrict__ ts2, int n_poses, int traj, int explicit_ts,
                                  float* __restrict__ poses) {
    spline_fwd_body(knots, transform, ts2, n_poses, traj, explicit_ts, poses, blockIdx.x * blockDim.x + threadIdx.x);
}

// the two trajectories of a training step in one launch (blockIdx.y: 0 = event poses on the knots themselves, 1 = exposure
// poses on knots + transform): a pose is one thread's long serial chain, two launches were two of them back to back
__global__ void spline_fwd_pair_kernel(const float* __restrict__ knots, const float* __restrict__ transform_b,
                                       const float* __restrict__ ts_a, int n_a, const float* __restrict__ ts_b, int n_b, int traj,
                                       float* __restrict__ poses_a, float* __restrict__ poses_b) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) spline_fwd_body(knots, nullptr, ts_a, n_a, traj, 0, poses_a, p);
    else spline_fwd_body(knots, transform_b, ts_b, n_b, traj, 0, poses_b, p);
}

// one block; thread (p, j): tangent of pose p w.r.t. effective-knot coefficient j (0..23);
// contrib[p][j] = <d_poses[p], tangent>; then j-threads sum over p in index order.
__device__ __forceinline__ void spline_bwd_body(const float* __restrict__ knots, const float* __restrict__ transform,
                                                const float* __restrict__ ts2, int n_poses, int traj, int explicit_ts,
                                                const float* __restrict__ d_poses, float* __restrict__ d_knots,
                                                float* __restrict__ d_transform, float* contrib) {
    // phase A: (q, t) of knot j/6 with the tangent of coefficient j - independent of the pose, 24 threads
    float* qt_v = contrib + n_poses * 24;      // [4][7] values
    float* qt_d = qt_v + 28;                   // [24][7] tangents
    if (threadIdx.x < 24) {
        const int j = threadIdx.x, i = j / 6;
        Dual wu[6], q[4], t[3];
#pragma unroll
        for (int c = 0; c < 6; ++c) wu[c] = Dual{knots[i * 6 + c] + (transform ? transform[c] : 0.f), (c == j % 6) ? 1.f : 0.f};
        se3_to_qt(wu, q, t);
#pragma unroll
        for (int c = 0; c < 4; ++c) qt_d[j * 7 + c] = q[c].d;
#pragma unroll
        for (int c = 0; c < 3; ++c) qt_d[j * 7 + 4 + c] = t[c].d;
        if (j % 6 == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) qt_v[i * 7 + c] = q[c].v;
#pragma unroll
            for (int c = 0; c < 3; ++c) qt_v[i * 7 + 4 + c] = t[c].v;
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < n_poses * 24; w += blockDim.x) {
        int p = w / 24, j = w % 24;
        Dual q[4][4], t[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool mine = (i == j / 6);
#pragma unroll
            for (int c = 0; c < 4; ++c) q[i][c] = Dual{qt_v[i * 7 + c], mine ? qt_d[j * 7 + c] : 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) t[i][c] = Dual{qt_v[i * 7 + 4 + c], mine ? qt_d[j * 7 + 4 + c] : 0.f};
        }
        float u = nudge(explicit_ts ? ts2[p] : linspace_at(ts2[0], ts2[1], n_poses, p));
        Dual out[12];
        if (traj == 1)
            linear_from_qt(q[0], t[0], q[3], t[3], u, out);
        else if (traj == 2)
            cubic_from_qt<true>(q, t, u, out);
        else
            cubic_from_qt<false>(q, t, u, out);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) s += d_poses[p * 12 + i] * out[i].d;
        contrib[w] = s;
    }
    __syncthreads();
    if (threadIdx.x < 24) {
        float s = 0.f;
        for (int p = 0; p < n_poses; ++p) s += contrib[p * 24 + threadIdx.x];
        contrib[threadIdx.x] = s;   // row 0 reused: safe, every reader of row 0 is this thread's column
        d_knots[threadIdx.x] = s;
    }
    __syncthreads();
    if (d_transform && threadIdx.x < 6) {
        d_transform[threadIdx.x] = contrib[threadIdx.x] + contrib[6 + threadIdx.x] + contrib[12 + threadIdx.x] +
                                   contrib[18 + threadIdx.x];
    }
}

__global__ void spline_bwd_kernel(const float* __restrict__ knots, const float* __restrict__ transform,
                                  const float* __restrict__ ts2, int n_poses, int traj, int explicit_ts,
                                  const float* __restrict__ d_poses, float* __restrict__ d_knots,
                                  float* __restrict__ d_transform) {
    extern __shared__ float contrib[];   // [n_poses][24]
    spline_bwd_body(knots, transform, ts2, n_poses, traj, explicit_ts, d_poses, d_knots, d_transform, contrib);
}

// the two trajectories of a training step (event camera: no transform; RGB camera: knots + transform) as the two
// blocks of ONE launch - each is a single block bound by the latency of the dual-number evaluation
struct SplinePair {
    const float* ts[2];
    const float* d_poses[2];
    float* d_knots[2];
    int n_poses[2];
};
__global__ void spline_bwd_pair_kernel(const float* __restrict__ knots, const float* __restrict__ transform_b, SplinePair p,
                                       int traj, float* __restrict__ d_transform_b) {
    extern __shared__ float contrib[];
    const int b = blockIdx.x;
    spline_bwd_body(knots, b ? transform_b : nullptr, p.ts[b], p.n_poses[b], traj, 0, p.d_poses[b], p.d_knots[b],
                    b ? d_transform_b : nullptr, contrib);
}

// ---- the reference's public spline helpers as single-op kernels (spline.py:16-192), same device functions ------------
enum SplineOp { OP_SE3_2_QT = 0, OP_EXP_R2Q, OP_LOG_Q2R, OP_Q_TO_R, OP_TAYLOR_B, OP_TAYLOR_C, OP_SKEW, OP_Q_TO_Q, OP_Q_CONJ, OP_COUNT };
__host__ __device__ constexpr int op_in(int op) {
    return op == OP_SE3_2_QT ? 6 : op == OP_EXP_R2Q || op == OP_SKEW ? 3 : op == OP_TAYLOR_B || op == OP_TAYLOR_C ? 1 : 4;
}
__host__ __device__ constexpr int op_out(int op) {
    return op == OP_SE3_2_QT ? 7 : op == OP_EXP_R2Q || op == OP_Q_CONJ ? 4 : op == OP_LOG_Q2R ? 3 : op == OP_Q_TO_R || op == OP_SKEW ? 9
           : op == OP_Q_TO_Q ? 16 : 1;
}
template <class T>
__device__ void apply_op(int op, const T* x, T* y) {
    switch (op) {
        case OP_SE3_2_QT: se3_to_qt(x, y, y + 4); break;          // [q xyzw | t]
        case OP_EXP_R2Q: rotvec_to_quat(x, y); break;
        case OP_LOG_Q2R: quat_to_rotvec(x, y); break;
        case OP_Q_TO_R: {
            T tz[3] = {lift<T>(0.f), lift<T>(0.f), lift<T>(0.f)}, o[12];
            quat_to_pose(x, tz, o);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) y[i * 3 + j] = o[i * 4 + j];
            break;
        }
        case OP_TAYLOR_B: y[0] = taylor_series(x[0], 1); break;
        case OP_TAYLOR_C: y[0] = taylor_series(x[0], 2); break;
        case OP_SKEW: {                                           // spline.py:28-34
            T z = lift<T>(0.f);
            y[0] = z; y[1] = -x[2]; y[2] = x[1];
            y[3] = x[2]; y[4] = z; y[5] = -x[0];
            y[6] = -x[1]; y[7] = x[0]; y[8] = z;
            break;
        }
        case OP_Q_TO_Q: {                                         // left-product matrix, spline.py:130-138
            T qx = x[0], qy = x[1], qz = x[2], qw = x[3];
            y[0] = qw; y[1] = -qz; y[2] = qy; y[3] = qx;
            y[4] = qz; y[5] = qw; y[6] = -qx; y[7] = qy;
            y[8] = -qy; y[9] = qx; y[10] = qw; y[11] = qz;
            y[12] = -qx; y[13] = -qy; y[14] = -qz; y[15] = qw;
            break;
        }
        default: quat_conj(x, y); break;                          // OP_Q_CONJ
    }
}
__global__ void spline_op_fwd_kernel(int op, const float* __restrict__ in, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int di = op_in(op), dout = op_out(op);
    float x[6], y[16];
    for (int c = 0; c < di; ++c) x[c] = in[i * di + c];
    apply_op<float>(op, x, y);
    for (int c = 0; c < dout; ++c) out[i * dout + c] = y[c];
}
// d_in[i][j] = <d_out[i], d y / d x_j>: one thread per (item, input component), forward-mode dual on that component
__global__ void spline_op_bwd_kernel(int op, const float* __restrict__ in, int64_t n, const float* __restrict__ d_out,
                                     float* __restrict__ d_in) {
    const int di = op_in(op), dout = op_out(op);
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n * di) return;
    const int64_t i = w / di;
    const int j = (int)(w % di);
    Dual x[6], y[16];
    for (int c = 0; c < di; ++c) x[c] = Dual{in[i * di + c], c == j ? 1.f : 0.f};
    apply_op<Dual>(op, x, y);
    float s = 0.f;
    for (int c = 0; c < dout; ++c) s += d_out[i * dout + c] * y[c].d;
    d_in[w] = s;
}

}  // namespace

extern "C" int benerf_spline_poses_fwd(const float* knots, const float* transform, const float* ts2, int n_poses,
                                       int traj, int explicit_ts, float* poses, benerf_stream_t stream) {
    BENERF_REQUIRE(knots && ts2 && poses, "spline_poses_fwd: null pointer");
    BENERF_REQUIRE(n_poses > 0 && (traj >= 0 && traj <= 2), "spline_poses_fwd: bad n_poses/traj");
    int threads = 64, blocks = (n_poses + threads - 1) / threads;
    hipLaunchKernelGGL(spline_fwd_kernel, dim3(blocks), dim3(threads), 0, as_stream(stream), knots, transform, ts2,
                       n_poses, traj, explicit_ts, poses);
    BENERF_LAUNCH_CHECK("spline_poses_fwd");
    return BENERF_OK;
}

extern "C" int benerf_spline_poses_fwd_pair(const float* knots, const float* transform_b, const float* ts_a, int n_a,
                                            const float* ts_b, int n_b, int traj, float* poses_a, float* poses_b,
                                            benerf_stream_t stream) {
    BENERF_REQUIRE(knots && transform_b && ts_a && ts_b && poses_a && poses_b, "spline_poses_fwd_pair: null pointer");
    BENERF_REQUIRE(n_a > 0 && n_b > 0 && (traj >= 0 && traj <= 2), "spline_poses_fwd_pair: bad sizes");
    const int n_max = n_a > n_b ? n_a : n_b, threads = 64;
    hipLaunchKernelGGL(spline_fwd_pair_kernel, dim3((n_max + threads - 1) / threads, 2), dim3(threads), 0, as_stream(stream), knots,
                       transform_b, ts_a, n_a, ts_b, n_b, traj, poses_a, poses_b);
    BENERF_LAUNCH_CHECK("spline_poses_fwd_pair");
    return BENERF_OK;
}

extern "C" int benerf_spline_poses_bwd(const float* knots, const float* transform, const float* ts2, int n_poses,
                                       int traj, int explicit_ts, const float* d_poses, float* d_knots,
                                       float* d_transform, benerf_stream_t stream) {
    BENERF_REQUIRE(knots && ts2 && d_poses && d_knots, "spline_poses_bwd: null pointer");
    BENERF_REQUIRE(n_poses > 0 && n_poses <= 512 && (traj >= 0 && traj <= 2), "spline_poses_bwd: n_poses must be in [1,512]");
    size_t smem = ((size_t)n_poses * 24 + 28 + 24 * 7) * sizeof(float);
    // one (pose, tangent) evaluation per thread where possible: the dual-number evaluation is a long serial
    // chain, so width beats depth (19 poses x 24 tangents = 456 threads in one block)
    int threads = ((n_poses * 24 + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    hipLaunchKernelGGL(spline_bwd_kernel, dim3(1), dim3(threads), smem, as_stream(stream), knots, transform, ts2, n_poses,
                       traj, explicit_ts, d_poses, d_knots, d_transform);
    BENERF_LAUNCH_CHECK("spline_poses_bwd");
    return BENERF_OK;
}

extern "C" int benerf_spline_poses_bwd_pair(const float* knots, const float* transform_b, const float* ts_a, int n_a,
                                            const float* ts_b, int n_b, int traj, const float* d_poses_a,
                                            const float* d_poses_b, float* d_knots_a, float* d_knots_b, float* d_transform_b,
                                            benerf_stream_t stream) {
    BENERF_REQUIRE(knots && transform_b && ts_a && ts_b && d_poses_a && d_poses_b && d_knots_a && d_knots_b && d_transform_b,
                   "spline_poses_bwd_pair: null pointer");
    BENERF_REQUIRE(n_a > 0 && n_a <= 512 && n_b > 0 && n_b <= 512 && (traj >= 0 && traj <= 2), "spline_poses_bwd_pair: bad sizes");
    const int n_max = n_a > n_b ? n_a : n_b;
    int threads = ((n_max * 24 + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    SplinePair p;
    p.ts[0] = ts_a; p.ts[1] = ts_b;
    p.d_poses[0] = d_poses_a; p.d_poses[1] = d_poses_b;
    p.d_knots[0] = d_knots_a; p.d_knots[1] = d_knots_b;
    p.n_poses[0] = n_a; p.n_poses[1] = n_b;
    hipLaunchKernelGGL(spline_bwd_pair_kernel, dim3(2), dim3(threads), ((size_t)n_max * 24 + 28 + 24 * 7) * sizeof(float), as_stream(stream), knots,
                       transform_b, p, traj, d_transform_b);
    BENERF_LAUNCH_CHECK("spline_poses_bwd_pair");
    return BENERF_OK;
}

extern "C" int benerf_spline_op_fwd(int op, const float* in, int64_t n, float* out, benerf_stream_t stream) {
    BENERF_REQUIRE(in && out && n >= 0 && op >= 0 && op < OP_COUNT, "spline_op_fwd: bad arguments (op %d)", op);
    if (n == 0) return BENERF_OK;
    hipLaunchKernelGGL(spline_op_fwd_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, as_stream(stream), op, in, n, out);
    BENERF_LAUNCH_CHECK("spline_op_fwd");
    return BENERF_OK;
}

extern "C" int benerf_spline_op_bwd(int op, const float* in, int64_t n, const float* d_out, float* d_in, benerf_stream_t stream) {
    BENERF_REQUIRE(in && d_out && d_in && n >= 0 && op >= 0 && op < OP_COUNT, "spline_op_bwd: bad arguments (op %d)", op);
    if (n == 0) return BENERF_OK;
    const int64_t w = n * op_in(op);
    hipLaunchKernelGGL(spline_op_bwd_kernel, dim3((unsigned)((w + 127) / 128)), dim3(128), 0, as_stream(stream), op, in, n, d_out, d_in);
    BENERF_LAUNCH_CHECK("spline_op_bwd");
    return BENERF_OK;
}
