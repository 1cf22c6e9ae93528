// Geodesic pose error right after retrieval: replaces
//   pred_R = template_poses[nearest_idx]                         src/model/model.py:352-354
//   GeodesicError / so3_relative_angle_with_symmetry             src/model/loss.py:14-115
// for the (B, k) retrieved poses of a batch, in float64 as the reference computes it (loss.py:87,103 cast to float64).
//
// One thread per (query, rank).  Per element:
//   symmetry 0  angle(P, G) = acos_le((tr(P G^T) - 1) / 2)                                   loss.py:20-22 -> pytorch3d so3_relative_angle(eps = 1e-2)
//   symmetry 1  min(angle(P, G), angle(f64(f32(RotY180) f32(P)), G))                         loss.py:29-48 (the flipped pose is an f32 product)
//   symmetry 2  acos(cos_sim(-inv(P)[2, :], -inv(G)[2, :])), unclamped                       loss.py:55-73 (camera Z axis in OpenGL convention)
// acos_le is pytorch3d's acos_linear_extrapolation with its default bound 1 - 1e-4 (transforms/math.py, published form):
//   |x| <  b : acos(x);   x >= b : (x - b) * (-1 / sqrt(1 - b^2)) + acos(b);   x <= -b : (x + b) * (-1 / sqrt(1 - b^2)) + acos(-b)
// and a trace outside [-1 - eps, 3 + eps] is the ValueError of so3_rotation_angle: reported through *status (bit 0), the
// host binding raises.  pytorch3d is not vendored by the reference and not installed: the formula is restated from its
// published source and pinned by hand-computable known answers (tests/test_host_logic.py, tests/test_gpu_configs.py).
#include <cmath>

#include "nope_common.h"

namespace nope {

namespace {

__device__ __forceinline__ double acos_le(double x) {
    const double b = 1.0 - 1e-4;
    const double slope = -1.0 / sqrt(1.0 - b * b);
    if (x >= b) return (x - b) * slope + acos(b);
    if (x <= -b) return (x + b) * slope + acos(-b);
    return acos(x);
}

// tr(P G^T) = sum_ij P_ij G_ij, summed row by row of the product P G^T as a batched matrix product does
__device__ __forceinline__ double rel_angle(const double* P, const double* G, double eps, int* status) {
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double d = 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) d = fma(P[3 * i + j], G[3 * i + j], d);
        tr += d;
    }
    if (tr < -1.0 - eps || tr > 3.0 + eps) atomicOr(status, 1);
    return acos_le((tr - 1.0) * 0.5);
}

// third row of inv(M) for a 3x3 matrix (adjugate / determinant), negated: the camera Z axis after convert_openCV_to_openGL
__device__ __forceinline__ void neg_inv_row2(const double* M, double* r) {
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    // inv = adj / det, adj[i][j] = cofactor[j][i]; row 2 of inv = cofactors of column 2 of M
    const double a20 = M[3] * M[7] - M[4] * M[6], a21 = M[1] * M[6] - M[0] * M[7], a22 = M[0] * M[4] - M[1] * M[3];
    r[0] = -(a20 / det); r[1] = -(a21 / det); r[2] = -(a22 / det);
}

__global__ __launch_bounds__(256) void geodesic_kernel(const double* __restrict__ poses, long long stride_b, int N,
                                                       const long long* __restrict__ idx, const double* __restrict__ gt,
                                                       const int* __restrict__ symmetry, double* __restrict__ err, int* __restrict__ status,
                                                       int B, int k) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * k) return;
    const int b = t / k, j = t - b * k;
    long long n = j;
    if (idx) {
        n = idx[t];
        if (n < 0 || n >= N) { atomicOr(status, 2); err[t] = nan(""); return; }
    }
    double P[9], G[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) { P[e] = poses[(size_t)b * stride_b + (size_t)n * 9 + e]; G[e] = gt[(size_t)b * 9 + e]; }
    const int sym = symmetry ? symmetry[b] : 0;
    const double eps = 1e-2;       // loss.py:21,32,46
    if (sym == 0) { err[t] = rel_angle(P, G, eps, status); return; }
    if (sym == 1) {
        const double e0 = rel_angle(P, G, eps, status);
        // RotY(180 deg) as load_rotation_transform("y", 180)[:3, :3].float() builds it (poses/utils.py:136-139): cos(pi), sin(pi) in
        // float64 rounded to f32; the product with the f32 prediction is taken in f32 (loss.py:36-43)
        const float c = (float)cos(M_PI), s = (float)sin(M_PI);
        float Pf[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) Pf[e] = (float)P[e];
        double R[9];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            R[x] = (double)fmaf(s, Pf[6 + x], fmaf(0.f, Pf[3 + x], c * Pf[x]));
            R[3 + x] = (double)fmaf(0.f, Pf[6 + x], fmaf(1.f, Pf[3 + x], 0.f * Pf[x]));
            R[6 + x] = (double)fmaf(c, Pf[6 + x], fmaf(0.f, Pf[3 + x], -s * Pf[x]));
        }
        const double e1 = rel_angle(R, G, eps, status);
        err[t] = e0 < e1 ? e0 : e1;
        return;
    }
    double p[3], g[3];
    neg_inv_row2(P, p);
    neg_inv_row2(G, g);
    // F.cosine_similarity(dim = 1, eps = 1e-8): x . y / (max(|x|, eps) max(|y|, eps))
    const double pn = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]), gn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    const double dot = p[0] * g[0] + p[1] * g[1] + p[2] * g[2];
    err[t] = acos(dot / ((pn > 1e-8 ? pn : 1e-8) * (gn > 1e-8 ? gn : 1e-8)));
}

}  // namespace

int launch_geodesic(const double* poses, long long stride_b, int N, const long long* idx, const double* gt, const int* symmetry,
                    double* err, int* status, int B, int k, hipStream_t s) {
    if (!poses || !gt || !err || !status || B <= 0 || k <= 0 || N <= 0 || stride_b < 0) return NOPE_ERR_ARG;
    if (!idx && k > N) return NOPE_ERR_ARG;
    if (hipMemsetAsync(status, 0, sizeof(int), s) != hipSuccess) return NOPE_ERR_LAUNCH;
    hipLaunchKernelGGL(geodesic_kernel, dim3((unsigned)cdiv(B * k, 256)), dim3(256), 0, s, poses, stride_b, N, idx, gt, symmetry, err, status, B, k);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

}  // namespace nope
