// Kernels that only the template encoder needs (src/model/encoder/resnet.py:118-152, template.py:24-53);
// its 1x1 / 3x3 / stride-2 convolutions run on the implicit-GEMM kernel of kernels_gemm.hip.
//   bn_fold_kernel    : eval-mode BatchNorm2d as a per-channel affine, scale = gamma / sqrt(var + eps),
//                       shift = beta - mean * scale (resnet.py:61-66,74-84 bn1/bn2/bn3, :124); the scale goes
//                       into the conv weights at pack time, the shift becomes the conv bias.
//   stem_pack_kernel  : conv1 weight [64][3][7][7] * scale[co] -> f32 [147][64] (tap-major, channel-minor).
//   stem_conv_kernel  : conv1 7x7 / stride 2 / pad 3 on the NCHW f32 image + shift + ReLU, written as the
//                       NHWC activation the GEMM convs consume.  3 input channels are no GEMM: one thread owns
//                       one output pixel x 16 output channels, image patch and weights come from LDS.
#include "nope_common.h"

namespace nope {

namespace {

constexpr int NT = 256;
constexpr int STEM_K = 3 * 7 * 7;      // 147
constexpr int STEM_C = 64;             // resnet.py:94-98 (features = 64)
constexpr int STEM_TILE = 8;           // 8 x 8 output pixels per workgroup
constexpr int STEM_PATCH = 2 * STEM_TILE + 5;   // 21 input rows / cols

__global__ __launch_bounds__(NT) void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                     float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - mean[c] * s;
}

__global__ __launch_bounds__(NT) void stem_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                       float* __restrict__ out) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= STEM_K * STEM_C) return;
    const int k = i / STEM_C, co = i - k * STEM_C;
    out[i] = w[co * STEM_K + k] * scale[co];
}

template <class T>
__global__ __launch_bounds__(NT) void stem_conv_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                       const float* __restrict__ shift, T* __restrict__ out, int H, int W, int Ho,
                                                       int Wo) {
    constexpr int VEC = Elt<T>::VEC;
    __shared__ __attribute__((aligned(16))) float s_w[STEM_K * STEM_C];
    __shared__ float s_x[3][STEM_PATCH][STEM_PATCH + 1];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * STEM_TILE, ox0 = blockIdx.x * STEM_TILE;
    for (int i = tid; i < STEM_K * STEM_C / 4; i += NT)
        reinterpret_cast<f32x4*>(s_w)[i] = reinterpret_cast<const f32x4*>(w)[i];
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    for (int i = tid; i < 3 * STEM_PATCH * STEM_PATCH; i += NT) {
        const int ci = i / (STEM_PATCH * STEM_PATCH);
        const int r = i - ci * STEM_PATCH * STEM_PATCH;
        const int py = r / STEM_PATCH, px = r - py * STEM_PATCH;
        const int iy = iy0 + py, ix = ix0 + px;
        s_x[ci][py][px] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? img[((size_t)(b * 3 + ci) * H + iy) * W + ix] : 0.f;
    }
    __syncthreads();
    const int pix = tid >> 2, cg = tid & 3;      // 64 pixels x 4 groups of 16 output channels
    const int ty = pix >> 3, tx = pix & 7;
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int ci = 0; ci < 3; ++ci)
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float x = s_x[ci][2 * ty + ky][2 * tx + kx];
                const float* wk = s_w + ((ci * 7 + ky) * 7 + kx) * STEM_C + cg * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wk + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q * 4 + e] = fmaf(x, wv[e], acc[q * 4 + e]);
                }
            }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= Ho || ox >= Wo) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = acc[e] + shift[cg * 16 + e];
        acc[e] = v > 0.f ? v : 0.f;
    }
    T* o = out + (((size_t)b * Ho + oy) * Wo + ox) * STEM_C + cg * 16;
#pragma unroll
    for (int q = 0; q < 16 / VEC; ++q) st16(o + q * VEC, Elt<T>::pack(acc + q * VEC));
}

}  // namespace

int launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale, float* shift,
                   int C, hipStream_t s) {
    if (!gamma || !beta || !mean || !var || !scale || !shift || C <= 0) return NOPE_ERR_ARG;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)cdiv(C, NT)), dim3(NT), 0, s, gamma, beta, mean, var, eps, scale, shift, C);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_stem_pack(const float* w, const float* scale, float* out, hipStream_t s) {
    if (!w || !scale || !out) return NOPE_ERR_ARG;
    hipLaunchKernelGGL(stem_pack_kernel, dim3((unsigned)cdiv(STEM_K * STEM_C, NT)), dim3(NT), 0, s, w, scale, out);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_stem_conv(int dt, const float* img, const float* w_packed, const float* shift, void* out, int n_img, int H, int W,
                     hipStream_t s) {
    if (!img || !w_packed || !shift || !out || n_img <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return NOPE_ERR_ARG;
    const int Ho = H / 2, Wo = W / 2;
    if (n_img > 65535) return NOPE_ERR_UNSUPPORTED;
    dim3 grid((unsigned)cdiv(Wo, STEM_TILE), (unsigned)cdiv(Ho, STEM_TILE), (unsigned)n_img);
    NOPE_DISPATCH_T(dt, T, hipLaunchKernelGGL((stem_conv_kernel<T>), grid, dim3(NT), 0, s, img, w_packed, shift, (T*)out, H, W, Ho, Wo));
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

}  // namespace nope
