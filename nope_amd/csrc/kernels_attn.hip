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

// 4 consecutive channels as one 8-byte (bf16) / 16-byte (f32) store
template <class T> __device__ __forceinline__ void store4(T* p, const f32x4& v);
template <> __device__ __forceinline__ void store4<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const f32x4& v) {
    const unsigned long long lo = cvt_pk_bf16(v[0], v[1]);
    const unsigned long long hi = cvt_pk_bf16(v[2], v[3]);
    *reinterpret_cast<unsigned long long*>(p) = lo | (hi << 32);
}

template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, const f32x4& v) {
    const unsigned long long lo = cvt_pk_f16(v[0], v[1]);
    const unsigned long long hi = cvt_pk_f16(v[2], v[3]);
    *reinterpret_cast<unsigned long long*>(p) = lo | (hi << 32);
}

template <bool FAST> __device__ __forceinline__ float fexp(float x) { return FAST ? __expf(x) : expf(x); }

// All global traffic is 16-byte vectors: a head row is 32 channels = CG = 32/VEC vectors.
template <class T>
__global__ __launch_bounds__(NT, 4) void linattn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n, int heads) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr bool FASTM = sizeof(T) == 2;   // bf16 storage: hardware exp2 is far inside the format's precision
    constexpr int CG = D / VEC;            // channel vectors per head row (4 bf16 / 8 f32)
    constexpr int PT = NT / CG;            // pixels staged per pass (64 bf16 / 32 f32)
    __shared__ __attribute__((aligned(16))) float s_a[PT][D];       // exp(k - kmax) tile; also the sweep-1 scratch
    __shared__ __attribute__((aligned(16))) float s_v[PT][D];
    __shared__ __attribute__((aligned(16))) float s_part[NT / 64][D][D];
    __shared__ __attribute__((aligned(16))) float s_ctx[D][D];
    __shared__ float s_kmax[D];
    __shared__ float s_zp[NT / 64][D];
    // A head's rows are 64 B (bf16): two heads share every 128-byte line of q, k and v.  Workgroup b runs on XCD b % 8, so
    // the plain (hyp, head) = (b / heads, b % heads) order puts the two sharers on different L2s and each line is fetched
    // from HBM twice.  With 4 heads and a grid that is a multiple of 16 the head PAIRS are dealt to the XCDs instead and
    // the two heads of a pair take adjacent launch slots of the same XCD.
    int hyp = blockIdx.x / heads, head = blockIdx.x % heads;
    if (heads == 4 && (gridDim.x & 15) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int pair = (slot >> 1) * 8 + xcd;          // pair = hyp * 2 + head / 2
        hyp = pair >> 1;
        head = (pair & 1) * 2 + (slot & 1);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HD = heads * D;
    const int ldq = 3 * HD;
    const T* base = qkv + (size_t)hyp * n * ldq;
    const T* qp = base + head * D;
    const T* kp = base + HD + head * D;
    const T* vp = base + 2 * HD + head * D;
    const float scale = 0.17677669529663687f;   // 32^-0.5
    const int cg = tid % CG, pp = tid / CG;     // staging role: (pixel-in-pass, channel vector)

    // ---- sweep 1: kmax[d] = max_n k[n][d] --------------------------------------------------
    {
        float m[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) m[e] = -3.0e38f;
        // (four independent loads in flight per thread: the sweep is latency-bound, 16 dependent iterations at n = 1024 before)
        for (int p = pp; p < n; p += 4 * PT) {
            u32x4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pu = p + u * PT < n ? p + u * PT : p;      // (a repeat of the first pixel past the end: max is idempotent)
                raw[u] = ld16(kp + (size_t)pu * ldq + cg * VEC);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float kv[VEC];
                Elt<T>::unpack(raw[u], kv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], kv[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) s_a[pp][cg * VEC + e] = m[e];
        __syncthreads();
        if (tid < D) {
            float mm = s_a[0][tid];
            for (int j = 1; j < PT; ++j) mm = fmaxf(mm, s_a[j][tid]);
            s_kmax[tid] = mm;
        }
        __syncthreads();
    }

    // ---- sweep 2: ctx[d][e] = sum_n exp(k[n][d]-kmax[d]) v[n][e],  z[d] = sum_n exp(...) -----
    // each wave takes a quarter of every staged tile; lane -> 4x4 block of ctx
    const int d0 = (lane >> 3) * 4, e0 = (lane & 7) * 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float zacc[4] = {0.f, 0.f, 0.f, 0.f};
    float kmx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) kmx[e] = s_kmax[cg * VEC + e];
    // the next tile's k / v vectors are requested before this tile's products: one HBM / L2 round trip per 64 pixels was exposed
    u32x4 kraw = {0u, 0u, 0u, 0u}, vraw = {0u, 0u, 0u, 0u};
    if (pp < n) { kraw = ld16(kp + (size_t)pp * ldq + cg * VEC); vraw = ld16(vp + (size_t)pp * ldq + cg * VEC); }
    for (int t0 = 0; t0 < n; t0 += PT) {
        const int p = t0 + pp;
        float ek[VEC], vv[VEC];
        if (p < n) {
            Elt<T>::unpack(kraw, ek);
            Elt<T>::unpack(vraw, vv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) ek[e] = fexp<FASTM>(ek[e] - kmx[e]);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { ek[e] = 0.f; vv[e] = 0.f; }
        }
        if (p + PT < n) { kraw = ld16(kp + (size_t)(p + PT) * ldq + cg * VEC); vraw = ld16(vp + (size_t)(p + PT) * ldq + cg * VEC); }
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s_a[pp][cg * VEC + e] = ek[e]; s_v[pp][cg * VEC + e] = vv[e]; }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < PT / 4; ++i) {
            const int r = wave * (PT / 4) + i;
            const f32x4 a = *reinterpret_cast<const f32x4*>(&s_a[r][d0]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&s_v[r][e0]);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                zacc[x] += a[x];
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] += a[x] * b[y];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int y = 0; y < 4; ++y) s_part[wave][d0 + x][e0 + y] = acc[x][y];
        if ((lane & 7) == 0) s_zp[wave][d0 + x] = zacc[x];
    }
    __syncthreads();
    for (int i = tid; i < D * D; i += NT) {
        const int d = i >> 5, e = i & (D - 1);
        float c = 0.f, z = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { c += s_part[w][d][e]; z += s_zp[w][d]; }
        s_ctx[d][e] = c / z;
    }
    __syncthreads();

    // ---- sweep 3: out[n][e] = sum_d ctx[d][e] * softmax_d(q[n])[d] * scale ----------------------
    // out^T tile = ctx^T (32 x 32, loop-invariant: 16 A-fragments in registers) x qs (32 x 16 pixels)
    // on the exact-f32 MFMA 16x16x4.  Lane (pixel = lane & 15, kg = lane >> 4) owns channels
    // d = kg*8 .. kg*8+7 of its pixel -- one 16-byte load for bf16 -- and MFMA step j contracts
    // d = kg*8 + j on both operands (any K permutation is valid when A and B agree).
    {
        const int pr = lane & 15, kg = lane >> 4;
        float cf[2][8];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 8; ++j) cf[mt][j] = s_ctx[kg * 8 + j][mt * 16 + pr];
        const int groups = (n + 15) >> 4;
        constexpr int QL = 8 / VEC;                      // 16-byte loads per lane and pixel (1 for the 16-bit types, 2 for f32)
        u32x4 qraw[QL];
        auto fetch_q = [&](int g) {
            const int p = g * 16 + pr;
#pragma unroll
            for (int c = 0; c < QL; ++c) qraw[c] = p < n ? ld16(qp + (size_t)p * ldq + kg * 8 + c * VEC) : u32x4{0u, 0u, 0u, 0u};
        };
        if (wave < groups) fetch_q(wave);
        for (int g = wave; g < groups; g += NT / 64) {
            const int p = g * 16 + pr;
            float qv[8];
#pragma unroll
            for (int c = 0; c < QL; ++c) Elt<T>::unpack(qraw[c], &qv[c * VEC]);      // (pixels past the end: zeros, never stored)
            if (g + NT / 64 < groups) fetch_q(g + NT / 64);                          // next group's q while this one is reduced
            float m = qv[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) m = fmaxf(m, qv[j]);
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float z = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { qv[j] = fexp<FASTM>(qv[j] - m); z += qv[j]; }
            z += __shfl_xor(z, 16, 64);
            z += __shfl_xor(z, 32, 64);
            const float qs = scale / z;
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float bq = qv[j] * qs;
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[0][j], bq, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[1][j], bq, o1, 0, 0, 0);
            }
            // D map: col = lane & 15 -> pixel, row = kg*4 + r -> e (4 consecutive channels per tile)
            if (p < n) {
                T* op = out + ((size_t)hyp * n + p) * HD + head * D + kg * 4;
                store4<T>(op, o0);
                store4<T>(op + 16, o1);
            }
        }
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
    NOPE_DISPATCH_T(dt, T, hipLaunchKernelGGL((linattn_kernel<T>), grid, block, 0, s, (const T*)qkv, (T*)out, HW, heads));
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_attn(int dt, const void* qkv, void* out, int nhyp, int HW, int heads, int dim_head, hipStream_t s) {
    if (!qkv || !out || nhyp <= 0 || HW <= 0 || heads <= 0) return NOPE_ERR_ARG;
    if (dim_head != D || HW > AT_MAXN) return NOPE_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(nhyp * heads)), block(64);
    NOPE_DISPATCH_T(dt, T, hipLaunchKernelGGL((attn_kernel<T>), grid, block, 0, s, (const T*)qkv, (T*)out, HW, heads));
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

}  // namespace nope
