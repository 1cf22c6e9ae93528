// Attention cores of the NOPE U-Net on a fused NHWC qkv tensor [hyp][n][3*heads*d]
// (channel order q | k | v, each heads x d -- the `chunk(3, dim=1)` + "b (h c) x y" split of
// model_utils.py:377-380,404-407).  One workgroup per (hypothesis, head); everything is
// accumulated in f32.
//
//  linattn_kernel  LinearAttention, model_utils.py:409-416:
//      q <- softmax_d(q) * d^-0.5 ; k <- softmax_n(k)
//      ctx[d,e] = sum_n k[d,n] v[e,n] ;  out[e,n] = sum_d ctx[d,e] q[d,n]
//    Three sweeps over the (L2-resident, <= 128 KB) head slice: max_n k, then
//    exp/normaliser + ctx through LDS tiles of 64 pixels, then the per-pixel q softmax and
//    the 32x32 ctx product.
//  attn_kernel     Attention, model_utils.py:381-389 (n <= 64 tokens, the 4x4 bottleneck).
#include "nope_common.h"

namespace nope {

namespace {

constexpr int NT = 256;
constexpr int D = 32;      // dim_head (fixed by the reference, model_utils.py:368,394)
constexpr int PT = 64;     // pixels per LDS tile

template <class T>
__global__ __launch_bounds__(NT) void linattn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n, int heads) {
    __shared__ float s_red[NT];
    __shared__ float s_kmax[D];
    __shared__ float s_ek[PT][D + 1];
    __shared__ __attribute__((aligned(16))) float s_v[PT][D];
    __shared__ float s_ctx[D][D + 1];
    __shared__ float s_z[D];
    const int hyp = blockIdx.x / heads, head = blockIdx.x % heads;
    const int tid = threadIdx.x;
    const int HD = heads * D;
    const int ldq = 3 * HD;
    const T* base = qkv + (size_t)hyp * n * ldq;
    const T* qp = base + head * D;
    const T* kp = base + HD + head * D;
    const T* vp = base + 2 * HD + head * D;
    const float scale = 0.17677669529663687f;   // 32^-0.5

    // ---- sweep 1: kmax[d] = max_n k[n][d] --------------------------------------------------
    {
        const int d = tid & (D - 1), pg = tid >> 5;   // 8 pixel groups
        float m = -3.0e38f;
        for (int p = pg; p < n; p += NT / D) m = fmaxf(m, Elt<T>::ld(kp + (size_t)p * ldq + d));
        s_red[tid] = m;
        __syncthreads();
        if (tid < D) {
            float mm = s_red[tid];
#pragma unroll
            for (int g = 1; g < NT / D; ++g) mm = fmaxf(mm, s_red[g * D + tid]);
            s_kmax[tid] = mm;
        }
        __syncthreads();
    }

    // ---- sweep 2: ctx[d][e] = sum_n exp(k[n][d]-kmax[d]) v[n][e],  z[d] = sum_n exp(...) -----
    const int cd = tid >> 3;          // 0..31
    const int ce = (tid & 7) * 4;     // 0,4,..,28
    float cacc[4] = {0.f, 0.f, 0.f, 0.f};
    float zacc = 0.f;
    for (int t0 = 0; t0 < n; t0 += PT) {
        for (int i = tid; i < PT * D; i += NT) {
            const int pp = i >> 5, d = i & (D - 1);
            const int p = t0 + pp;
            float ek = 0.f, vv = 0.f;
            if (p < n) {
                ek = expf(Elt<T>::ld(kp + (size_t)p * ldq + d) - s_kmax[d]);
                vv = Elt<T>::ld(vp + (size_t)p * ldq + d);
            }
            s_ek[pp][d] = ek;
            s_v[pp][d] = vv;
        }
        __syncthreads();
#pragma unroll 8
        for (int pp = 0; pp < PT; ++pp) {
            const float ek = s_ek[pp][cd];
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(&s_v[pp][ce]);
            cacc[0] += ek * v4[0]; cacc[1] += ek * v4[1]; cacc[2] += ek * v4[2]; cacc[3] += ek * v4[3];
            zacc += ek;
        }
        __syncthreads();
    }
    if ((tid & 7) == 0) s_z[cd] = zacc;
    __syncthreads();
    {
        const float inv = 1.0f / s_z[cd];
#pragma unroll
        for (int j = 0; j < 4; ++j) s_ctx[cd][ce + j] = cacc[j] * inv;
    }
    __syncthreads();

    // ---- sweep 3: out[n][e] = sum_d ctx[d][e] * softmax_d(q[n])[d] * scale ----------------------
    // 4 threads per pixel, 8 output channels each
    const int sub = tid & 3;
    for (int p = tid >> 2; p < n; p += NT / 4) {
        float qv[D];
        float m = -3.0e38f;
#pragma unroll
        for (int d = 0; d < D; ++d) { qv[d] = Elt<T>::ld(qp + (size_t)p * ldq + d); m = fmaxf(m, qv[d]); }
        float z = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { qv[d] = expf(qv[d] - m); z += qv[d]; }
        const float qs = scale / z;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float qd = qv[d] * qs;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += s_ctx[d][sub * 8 + j] * qd;
        }
        T* op = out + ((size_t)hyp * n + p) * HD + head * D + sub * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) Elt<T>::st(op + j, o[j]);
    }
}

constexpr int AT_MAXN = 64;

template <class T>
__global__ __launch_bounds__(64) void attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n, int heads) {
    __shared__ float s_q[AT_MAXN][D + 1];
    __shared__ float s_k[AT_MAXN][D + 1];
    __shared__ float s_v[AT_MAXN][D + 1];
    __shared__ float s_sim[AT_MAXN][AT_MAXN + 1];
    const int hyp = blockIdx.x / heads, head = blockIdx.x % heads;
    const int tid = threadIdx.x;
    const int HD = heads * D;
    const int ldq = 3 * HD;
    const T* base = qkv + (size_t)hyp * n * ldq + head * D;
    const float scale = 0.17677669529663687f;
    for (int i = tid; i < n * D; i += 64) {
        const int p = i >> 5, d = i & (D - 1);
        s_q[p][d] = Elt<T>::ld(base + (size_t)p * ldq + d) * scale;
        s_k[p][d] = Elt<T>::ld(base + (size_t)p * ldq + HD + d);
        s_v[p][d] = Elt<T>::ld(base + (size_t)p * ldq + 2 * HD + d);
    }
    __syncthreads();
    for (int ij = tid; ij < n * n; ij += 64) {
        const int i = ij / n, j = ij - i * n;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) a += s_q[i][d] * s_k[j][d];
        s_sim[i][j] = a;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 64) {
        float m = -3.0e38f;
        for (int j = 0; j < n; ++j) m = fmaxf(m, s_sim[i][j]);
        float z = 0.f;
        for (int j = 0; j < n; ++j) { const float e = expf(s_sim[i][j] - m); s_sim[i][j] = e; z += e; }
        const float inv = 1.0f / z;
        for (int j = 0; j < n; ++j) s_sim[i][j] *= inv;
    }
    __syncthreads();
    for (int id = tid; id < n * D; id += 64) {
        const int i = id >> 5, d = id & (D - 1);
        float a = 0.f;
        for (int j = 0; j < n; ++j) a += s_sim[i][j] * s_v[j][d];
        Elt<T>::st(out + ((size_t)hyp * n + i) * HD + head * D + d, a);
    }
}

}  // namespace

int launch_linattn(int dt, const void* qkv, void* out, int nhyp, int HW, int heads, int dim_head, hipStream_t s) {
    if (!qkv || !out || nhyp <= 0 || HW <= 0 || heads <= 0) return NOPE_ERR_ARG;
    if (dim_head != D) return NOPE_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(nhyp * heads)), block(NT);
    if (dt == NOPE_F32) hipLaunchKernelGGL((linattn_kernel<float>), grid, block, 0, s, (const float*)qkv, (float*)out, HW, heads);
    else if (dt == NOPE_BF16) hipLaunchKernelGGL((linattn_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)qkv, (bf16_t*)out, HW, heads);
    else return NOPE_ERR_UNSUPPORTED;
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_attn(int dt, const void* qkv, void* out, int nhyp, int HW, int heads, int dim_head, hipStream_t s) {
    if (!qkv || !out || nhyp <= 0 || HW <= 0 || heads <= 0) return NOPE_ERR_ARG;
    if (dim_head != D || HW > AT_MAXN) return NOPE_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(nhyp * heads)), block(64);
    if (dt == NOPE_F32) hipLaunchKernelGGL((attn_kernel<float>), grid, block, 0, s, (const float*)qkv, (float*)out, HW, heads);
    else if (dt == NOPE_BF16) hipLaunchKernelGGL((attn_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)qkv, (bf16_t*)out, HW, heads);
    else return NOPE_ERR_UNSUPPORTED;
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

}  // namespace nope
