// Small data-movement kernels around the U-Net: layout changes at the NCHW API boundary,
// one-time weight repacking, and the tiny f32 linears of the pose embedding path.
#include "nope_common.h"

namespace nope {

namespace {

constexpr int NT = 256;

// Store element i of a packed weight tensor whose rows hold Cin values along K.  For NOPE_BF16X3 a row is 4 Cin bytes of
// 32-byte groups [hi x 8 | lo x 8] bf16, hi = rn(v), lo = rn(v - hi): channel c of a row sits in group c / 8 (Tile<f32s_t>,
// conv_gemm_common.h).  Cin % 8 == 0 (checked by the launcher).
template <class T> __device__ __forceinline__ void st_w(T* out, size_t i, int Cin, float v) { (void)Cin; Elt<T>::st(out + i, v); }
template <> __device__ __forceinline__ void st_w<f32s_t>(f32s_t* out, size_t i, int Cin, float v) {
    const size_t row = i / (size_t)Cin;
    const int c = (int)(i - row * (size_t)Cin);
    bf16_t* o = reinterpret_cast<bf16_t*>(out) + row * 2 * (size_t)Cin + (size_t)(c >> 3) * 16 + (c & 7);
    const bf16_t hi = f32_to_bf16(v);
    o[0] = hi;
    o[8] = f32_to_bf16(v - bf16_to_f32(hi));
}

// (n, C, HW) f32 NCHW -> [n][HW][C] T.  C is small (8 latent channels) at this boundary.
// (Csrc < C: the source has fewer channels; channels Csrc .. C-1 of the output are zeros -- inputs padded to a whole 16-byte vector)
template <class T>
__global__ __launch_bounds__(NT) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int C, int HW, size_t total, int Csrc) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        const size_t t = i / C;
        const int p = (int)(t % HW);
        const size_t b = t / HW;
        Elt<T>::st(y + i, c < Csrc ? x[(b * Csrc + c) * HW + p] : 0.f);
    }
}

template <class T>
__global__ __launch_bounds__(NT) void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int C, int HW, size_t total) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const int p = (int)(i % HW);
        const size_t t = i / HW;
        const int c = (int)(t % C);
        const size_t b = t / C;
        y[i] = Elt<T>::ld(x + (b * HW + p) * C + c);
    }
}

// torch Conv2d weight [Cout][Cin_t][kh][kw] f32 -> packed [Cout][tap][Cin] T.
//   PLAIN / UP2: Cin_t = Cin, tap = kh*3 + kw (or the single 1x1 tap).
//   DOWN2: the conv is 1x1 over the space-to-depth tensor whose channel index is
//          c*4 + p1*2 + p2 (einops "b (c p1 p2) h w", model_utils.py:170); tap = p1*2 + p2.
template <class T>
__global__ __launch_bounds__(NT) void pack_conv_w_kernel(const float* __restrict__ w, T* __restrict__ out, int Cin, int ntaps, int mode,
                                                         size_t total, const float* __restrict__ cin_scale,
                                                         const float* __restrict__ cout_scale, int Csrc) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % Cin);
        const size_t t = i / Cin;
        const int tap = (int)(t % ntaps);
        const size_t co = t / ntaps;
        size_t src;
        if (mode == NOPE_CONV_DOWN2) src = co * ((size_t)Csrc * 4) + (size_t)c * 4 + tap;
        else src = (co * Csrc + c) * ntaps + tap;
        float v = c < Csrc ? w[src] : 0.f;         // (Csrc < Cin: zero weights for the channels the input was padded with)
        if (cin_scale) v *= cin_scale[c];      // per-input-channel scale (PreNorm gamma)
        if (cout_scale) v *= cout_scale[co];   // per-output-channel scale (folded eval-mode BatchNorm)
        st_w<T>(out, i, Cin, v);
    }
}

// ---- NOPE_F16X2 weight layout (Tile<f16x2_t>, conv_gemm_common.h): rows of Cin channels in chunks of 32 = 128 bytes = eight 16-byte slots:
// slots 0..3 the f16 hi parts of the chunk's 32 channels (8 per slot); slot 4 + h the 16 e4m3 bytes of w * 2^sw over "channel set" h =
// channels 8 h .. 8 h + 7 and 16 + 8 h .. 16 + 8 h + 7 (what the lanes of half h hold for the two f16 MFMAs), slot 6 + h those of
// (w - hi) * 2^(sw + 11).  sw is the layer's power-of-two pre-scale, the largest that keeps max |w| * 2^sw below 256 (the format reaches
// 448); |w - hi| <= 2^-11 max |w|, so the second operand cannot overflow either.  Three launches: the f32 rows in GEMM order into the
// destination (pack_conv_w_kernel<float> and its phase-conv siblings: every conv mode, same row order as every other pack), max |w| of
// those rows into the tail of the buffer (bits of a non-negative float order like unsigned integers), then the rows are encoded IN PLACE
// -- a chunk is 128 bytes before and after, one thread owns it; the first thread also writes the E8M0 scale byte the conv kernel hands to
// the MFMA: 127 - 9 - sw (nope_common.h).
__global__ __launch_bounds__(NT) void absmax_bits_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const float v = fabsf(w[i]);
        const unsigned b = __builtin_bit_cast(unsigned, v);
        if (v == v && b > m) m = b;                    // (a NaN weight does not decide the scale)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// ... of an activation tensor (the NOPE_F16X2 range tracking of tensors no gn_apply produced): 16-byte loads, four in flight per thread
__global__ __launch_bounds__(NT) void absmax_f32_kernel(const float* __restrict__ x, size_t nvec, size_t n, unsigned* __restrict__ out) {
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * NT;
    size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const f32x4*>(x)[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) m = amax4(m, v[u][0], v[u][1], v[u][2], v[u][3]);
    }
    for (; i < nvec; i += stride) { const f32x4 v = reinterpret_cast<const f32x4*>(x)[i]; m = amax4(m, v[0], v[1], v[2], v[3]); }
    if (blockIdx.x == 0) for (size_t k = nvec * 4 + threadIdx.x; k < n; k += NT) m = __builtin_fmaxf(m, __builtin_fabsf(x[k]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) amax_publish(out, m, blockIdx.x * (NT / 64) + (threadIdx.x >> 6));
}
// compact[i] = max over the kX2Spread lines of range slot i, which are zeroed for the next round (one 64-thread block per slot)
__global__ __launch_bounds__(64) void amax_reduce_kernel(unsigned* __restrict__ slots, unsigned* __restrict__ compact) {
    unsigned* p = slots + (size_t)blockIdx.x * kX2SlotWords + (threadIdx.x & (kX2Spread - 1)) * 32;
    unsigned m = 0;
    if (threadIdx.x < kX2Spread) { m = *p; *p = 0u; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if (threadIdx.x == 0) compact[blockIdx.x] = m;
}
// The verdict of one forward (NOPE_F16X2 range tracking; one block, enqueued behind the forward's last kernel): fold every range slot (and
// zero it for the next forward), take each layer's largest input maximum through `tab` -- [layer] = {t, slot, slot, slot, slot}, slot -1 =
// none; a layer that ran no two-pass launch in this forward has no slots -- and judge it against the layer's window (nope_common.h: kX2*):
// out = the e4m3(a') operand saturated (max |a| 2^-t > 1792), has no full-precision element (< 2^-4), or the maximum is not finite.
// status[0] = layers out in THIS forward (read by x2_poison_kernel), status[1] = forwards judged, status[2] / [3] = out / non-finite layers
// since create.  `host` (mapped host memory, polled without a synchronisation: nope_unet_x2_poll): [0] = status[1], [1] = status[0],
// [2] = status[2], [3] = status[3], [4 + l] = bits of the maximum that put layer l out of or near the end of its window (the host
// re-centres t from it and clears the word), [4 + n + l] = bits of layer l's maximum in this forward.
__global__ __launch_bounds__(NT) void x2_verdict_kernel(unsigned* __restrict__ slots, int nslots, const int* __restrict__ tab, int n_layers,
                                                        unsigned* __restrict__ status, volatile unsigned* __restrict__ host) {
    __shared__ unsigned compact[512];
    __shared__ unsigned n_out, n_inf;
    if (threadIdx.x == 0) { n_out = 0; n_inf = 0; }
    // fold: 32 lanes per slot, one line each (the loads of a pass are independent: a 100-slot forward folds in ~13 round trips, not 3 200)
    for (int i0 = 0; i0 < nslots && i0 < 512; i0 += NT / kX2Spread) {
        const int i = i0 + (int)(threadIdx.x / kX2Spread);
        unsigned m = 0;
        if (i < nslots && i < 512) {
            unsigned* p = slots + (size_t)i * kX2SlotWords + (threadIdx.x % kX2Spread) * 32;
            m = *p;
            if (m) *p = 0u;
        }
#pragma unroll
        for (int o = kX2Spread / 2; o >= 1; o >>= 1) { const unsigned v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
        if (i < nslots && i < 512 && threadIdx.x % kX2Spread == 0) compact[i] = m;
    }
    __syncthreads();
    for (int l = threadIdx.x; l < n_layers; l += NT) {
        const int t = tab[l * 5];
        unsigned b = 0;
#pragma unroll
        for (int k = 1; k < 5; ++k) { const int sl = tab[l * 5 + k]; if (sl >= 0 && sl < 512 && compact[sl] > b) b = compact[sl]; }
        host[4 + n_layers + l] = b;
        if (!b) continue;
        const float amax = __builtin_bit_cast(float, b);
        if (b >= 0x7f800000u) { atomicAdd(&n_out, 1u); atomicAdd(&n_inf, 1u); host[4 + l] = b; continue; }      // inf (a NaN never enters a slot)
        const float v = ldexpf(amax, -t);
        const bool out = v > kX2AMaxFull || v < 0.0625f, uneasy = v > 1024.f || v < 1.f;
        if (out) atomicAdd(&n_out, 1u);
        if (out || uneasy) host[4 + l] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        status[0] = n_out; status[1] += 1u; status[2] += n_out; status[3] += n_inf;
        host[1] = n_out; host[2] = status[2]; host[3] = status[3];
        __threadfence_system();
        host[0] = status[1];
    }
}
// ... and what an out-of-range forward leaves in its output: NaNs, so that no caller reads inaccurate values for accurate ones (status[0] == 0:
// every block returns at once)
__global__ __launch_bounds__(NT) void x2_poison_kernel(unsigned* __restrict__ out, size_t nwords, unsigned pattern, const unsigned* __restrict__ status) {
    if (status[0] == 0u) return;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < nwords; i += (size_t)gridDim.x * NT) out[i] = pattern;
}
__device__ __forceinline__ int x2_weight_shift(unsigned maxbits) {
    const int e = (int)(maxbits >> 23) - 127;          // floor(log2 max |w|) (normal numbers; zero / subnormal maxima give e = -127)
    const int sw = 7 - e;                              // max |w| * 2^sw in [128, 256)
    return sw < -64 ? -64 : (sw > 100 ? 100 : sw);
}
__global__ __launch_bounds__(NT) void encode_w_x2_kernel(unsigned char* __restrict__ buf, size_t chunks, int* __restrict__ tail) {
    const int sw = x2_weight_shift((unsigned)tail[2]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { tail[0] = 127 - kX2ALoShift - sw; tail[1] = sw; tail[3] = 0; }
    const float s8 = ldexpf(1.0f, sw), sl8 = ldexpf(1.0f, sw + kX2WLoExtra);
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < chunks; i += (size_t)gridDim.x * NT) {
        unsigned char* chunk = buf + i * 128;
        float v[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u32x4 t = reinterpret_cast<const u32x4*>(chunk)[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const unsigned u = t[e]; v[4 * q + e] = __builtin_bit_cast(float, u); }
        }
        union { unsigned char b[128]; u32x4 q[8]; } o;
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) {
            const f16_t hi = f32_to_f16_sat(v[cc]);
            const float lo = v[cc] - (float)hi;
            reinterpret_cast<f16_t*>(o.b)[cc] = hi;                                            // slots 0..3
            const int h = (cc >> 3) & 1, pos = (cc >> 4) * 8 + (cc & 7);
            o.b[(4 + h) * 16 + pos] = (unsigned char)(cvt_pk_e4m3(v[cc] * s8, 0.f) & 0xffu);
            o.b[(6 + h) * 16 + pos] = (unsigned char)(cvt_pk_e4m3(lo * sl8, 0.f) & 0xffu);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) reinterpret_cast<u32x4*>(chunk)[q] = o.q[q];
    }
}

// out[r] = sum_k packed[r][k] (f32 sum of the values as the GEMM will see them)
template <class T>
__global__ __launch_bounds__(NT) void rowsum_kernel(const T* __restrict__ a, float* __restrict__ out, int rows, int K) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * NT + threadIdx.x) >> 6;
    if (wave >= rows) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += Elt<T>::ld(a + (size_t)wave * K + k);
    s = wave_sum(s);
    if (lane == 0) out[wave] = s;
}

// HardUpsample 3x3 weight [Cout][Cin][3][3] -> four phase sets [py*2+px][Cout][ty*2+tx][Cin]:
// tap (ty, tx) of phase (py, px) reads source pixel (y + ty + py - 1, x + tx + px - 1) and carries the sum of
// the 3x3 taps that land on it: rows {0} | {1,2} for py = 0, {0,1} | {2} for py = 1 (same for columns).
template <class T>
__global__ __launch_bounds__(NT) void pack_up2p_w_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, size_t total) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % Cin);
        size_t t = i / Cin;
        const int tap = (int)(t % 4); t /= 4;
        const int co = (int)(t % Cout);
        const int ph = (int)(t / Cout);
        const int py = ph >> 1, px = ph & 1, ty = tap >> 1, tx = tap & 1;
        const int y0 = py == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), y1 = py == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
        const int x0 = px == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), x1 = px == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
        float a = 0.f;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) a += w[(((size_t)co * Cin + c) * 3 + yy) * 3 + xx];
        st_w<T>(out, i, Cin, a);
    }
}

// Upsample = ConvTranspose2d(4, stride 2, pad 1), weight [Cin][Cout][4][4] (model_utils.py:119-126), as the same four phase
// sets: output pixel (2y + py, 2x + px) receives source pixel (y + ty + py - 1, x + tx + px - 1) through kernel element
// (3 - py - 2 ty, 3 - px - 2 tx) -- two source rows per output row, exactly the footprint of the nearest-x2 + 3x3 phases.
template <class T>
__global__ __launch_bounds__(NT) void pack_convT4_w_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, size_t total) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % Cin);
        size_t t = i / Cin;
        const int tap = (int)(t % 4); t /= 4;
        const int co = (int)(t % Cout);
        const int ph = (int)(t / Cout);
        const int ky = 3 - (ph >> 1) - 2 * (tap >> 1), kx = 3 - (ph & 1) - 2 * (tap & 1);
        st_w<T>(out, i, Cin, w[(((size_t)c * Cout + co) * 4 + ky) * 4 + kx]);
    }
}

template <class T>
__global__ __launch_bounds__(NT) void cast_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) Elt<T>::st(out + i, in[i]);
}

__global__ __launch_bounds__(NT) void silu_f32_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) out[i] = silu_f<false>(in[i]);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// out[m][n] = sum_k act(in[m][k]) * w[n][k] + bias[n]; one wave per output element row-chunk:
// each wave owns (m, 64 consecutive n?) -- no: K is the contiguous axis of both operands, so a
// wave owns one (m, n) pair group: lane l accumulates k = l, l+64, ... and the wave reduces.
__global__ __launch_bounds__(NT) void linear_naive_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int M, int N, int K,
                                                          int act_in, int ldo) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * NT + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * NT) >> 6;
    for (size_t o = wave; o < (size_t)M * N; o += nwaves) {
        const int m = (int)(o / N), n = (int)(o % N);
        float a = 0.f;
        for (int k = lane; k < K; k += 64) {
            float x = in[(size_t)m * K + k];
            if (act_in == 1) x = silu_f<false>(x);
            else if (act_in == 2) x = gelu_erf(x);
            a += x * w[(size_t)n * K + k];
        }
        a = wave_sum(a);
        if (lane == 0) out[(size_t)m * ldo + n] = a + (bias ? bias[n] : 0.f);
    }
}

inline unsigned grid_for(size_t n) {
    size_t g = (n + NT - 1) / NT;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

int launch_nchw_to_nhwc(int dt, const float* x, void* y, int n, int C, int HW, hipStream_t s, int C_src) {
    if (!x || !y || n <= 0 || C <= 0 || HW <= 0 || C_src < 0 || C_src > C) return NOPE_ERR_ARG;
    const size_t total = (size_t)n * C * HW;
    const int cs = C_src ? C_src : C;
    NOPE_DISPATCH_T(dt, T, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(total)), dim3(NT), 0, s, x, (T*)y, C, HW, total, cs));
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_nhwc_to_nchw_f32(int dt, const void* x, float* y, int n, int C, int HW, hipStream_t s) {
    if (!x || !y || n <= 0 || C <= 0 || HW <= 0) return NOPE_ERR_ARG;
    const size_t total = (size_t)n * C * HW;
    NOPE_DISPATCH_T(dt, T, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for(total)), dim3(NT), 0, s, (const T*)x, y, C, HW, total));
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_pack_conv_w(int dt, const float* w, void* out, int Cout, int Cin, int ntaps, int mode, hipStream_t s, const float* cin_scale,
                       const float* cout_scale, int Cin_src) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || ntaps <= 0 || Cin_src < 0 || Cin_src > Cin) return NOPE_ERR_ARG;
    if (Cin_src && Cin_src != Cin && (mode == NOPE_CONV_UP2P || cin_scale)) return NOPE_ERR_ARG;
    const int csrc = Cin_src ? Cin_src : Cin;
    if (dt == NOPE_F16X2) {                                             // the f16 + MX-fp8 tile's layout (tap-resident and per-tap ping-pong kernels)
        if (cin_scale || cout_scale || csrc != Cin) return NOPE_ERR_UNSUPPORTED;
        return launch_pack_conv_w_x2(w, out, Cout, Cin, s, ntaps, mode);
    }
    if (mode == NOPE_CONV_DOWN2 && ntaps != 4) return NOPE_ERR_ARG;
    if (dt == NOPE_BF16X3 && Cin % 8) return NOPE_ERR_UNSUPPORTED;      // (hi, lo) groups of 8 channels
    if (mode == NOPE_CONV_UP2P && ntaps == 16) {     // source is a ConvTranspose2d(4, 2, 1) weight
        if (cin_scale || cout_scale) return NOPE_ERR_ARG;
        const size_t tot = (size_t)4 * Cout * 4 * Cin;
        NOPE_DISPATCH_W(dt, T, hipLaunchKernelGGL((pack_convT4_w_kernel<T>), dim3(grid_for(tot)), dim3(NT), 0, s, w, (T*)out, Cout, Cin, tot));
        NOPE_CHECK_LAUNCH();
        return NOPE_OK;
    }
    if (mode == NOPE_CONV_UP2P) {
        if (ntaps != 4 || cin_scale || cout_scale) return NOPE_ERR_ARG;
        const size_t tot = (size_t)4 * Cout * 4 * Cin;
        NOPE_DISPATCH_W(dt, T, hipLaunchKernelGGL((pack_up2p_w_kernel<T>), dim3(grid_for(tot)), dim3(NT), 0, s, w, (T*)out, Cout, Cin, tot));
        NOPE_CHECK_LAUNCH();
        return NOPE_OK;
    }
    const size_t total = (size_t)Cout * ntaps * Cin;
    NOPE_DISPATCH_W(dt, T, hipLaunchKernelGGL((pack_conv_w_kernel<T>), dim3(grid_for(total)), dim3(NT), 0, s, w, (T*)out, Cin, ntaps, mode, total, cin_scale, cout_scale, csrc));
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

static size_t x2_rows(int Cout, int ntaps, int mode) { return mode == NOPE_CONV_UP2P ? (size_t)16 * Cout : (size_t)Cout * ntaps; }      // (four phases x four taps)
size_t conv_w_x2_bytes(int Cout, int Cin, int ntaps, int mode) { return x2_rows(Cout, ntaps, mode) * Cin * 4 + kX2TailBytes; }

int launch_x2_verdict(unsigned* slots, int nslots, const int* tab, int n_layers, unsigned* status, unsigned* host_mapped, hipStream_t s) {
    if (!slots || !tab || !status || !host_mapped || nslots <= 0 || nslots > 512 || n_layers <= 0) return NOPE_ERR_ARG;
    hipLaunchKernelGGL(x2_verdict_kernel, dim3(1), dim3(NT), 0, s, slots, nslots, tab, n_layers, status, (volatile unsigned*)host_mapped);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}
int launch_x2_poison(void* out, size_t bytes, int out_dt, const unsigned* status, hipStream_t s) {
    if (!out || !status || bytes % 4) return NOPE_ERR_ARG;
    const unsigned pattern = out_dt == NOPE_F32 ? 0x7fc00000u : out_dt == NOPE_F16 ? 0x7e007e00u : 0x7fc07fc0u;      // quiet NaNs of the output type
    const size_t nwords = bytes / 4;
    size_t blocks = (nwords + (size_t)NT * 16 - 1) / ((size_t)NT * 16);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(x2_poison_kernel, dim3((unsigned)blocks), dim3(NT), 0, s, (unsigned*)out, nwords, pattern, status);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_amax_reduce(unsigned* slots, int nslots, unsigned* compact, hipStream_t s) {
    if (!slots || !compact || nslots <= 0) return NOPE_ERR_ARG;
    hipLaunchKernelGGL(amax_reduce_kernel, dim3((unsigned)nslots), dim3(64), 0, s, slots, compact);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_absmax_f32(const float* x, size_t n, unsigned* out, hipStream_t s) {
    if (!x || !out) return NOPE_ERR_ARG;
    if (n == 0) return NOPE_OK;
    const size_t nvec = n / 4;
    size_t blocks = (nvec + (size_t)NT * 8 - 1) / ((size_t)NT * 8);
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(absmax_f32_kernel, dim3((unsigned)blocks), dim3(NT), 0, s, x, nvec, n, out);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_pack_conv_w_x2(const float* w, void* out, int Cout, int Cin, hipStream_t s, int ntaps, int mode) {
    if (!w || !out || Cout <= 0 || Cin <= 0) return NOPE_ERR_ARG;
    if (Cin % 32) return NOPE_ERR_UNSUPPORTED;                          // whole 32-channel chunks
    if (mode != NOPE_CONV_PLAIN && mode != NOPE_CONV_DOWN2 && mode != NOPE_CONV_UP2P) return NOPE_ERR_UNSUPPORTED;      // the modes of the ping-pong kernels
    if (const int e = launch_pack_conv_w(NOPE_F32, w, out, Cout, Cin, ntaps, mode, s)) return e;
    const size_t total = x2_rows(Cout, ntaps, mode) * Cin;
    int* tail = reinterpret_cast<int*>((unsigned char*)out + total * 4);
    if (hipMemsetAsync(tail, 0, kX2TailBytes, s) != hipSuccess) return NOPE_ERR_LAUNCH;
    hipLaunchKernelGGL(absmax_bits_kernel, dim3(grid_for(total)), dim3(NT), 0, s, (const float*)out, total, reinterpret_cast<unsigned*>(tail) + 2);
    NOPE_CHECK_LAUNCH();
    hipLaunchKernelGGL(encode_w_x2_kernel, dim3(grid_for(total / 32)), dim3(NT), 0, s, (unsigned char*)out, total / 32, tail);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

// (NOPE_BF16X3: a packed row of K channels is 2 K bf16 values, hi and lo parts: their plain sum is the row sum the MFMAs see)
int launch_rowsum(int dt, const void* packed, float* out, int rows, int K, hipStream_t s) {
    if (!packed || !out || rows <= 0 || K <= 0) return NOPE_ERR_ARG;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (dt == NOPE_BF16X3) { dt = NOPE_BF16; K *= 2; }
    NOPE_DISPATCH_T(dt, T, hipLaunchKernelGGL((rowsum_kernel<T>), dim3(blocks), dim3(NT), 0, s, (const T*)packed, out, rows, K));
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_cast(int dt, const float* in, void* out, size_t n, hipStream_t s) {
    if (!in || !out) return NOPE_ERR_ARG;
    if (n == 0) return NOPE_OK;
    NOPE_DISPATCH_T(dt, T, hipLaunchKernelGGL((cast_kernel<T>), dim3(grid_for(n)), dim3(NT), 0, s, in, (T*)out, n));
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

// "posEncoding" pose embedding, src/model/utils.py:36-51 with dim = classes / pose_dim: out[b, i*half + j] = sin(pose[b,i] * f_j),
// out[b, classes/2 + i*half + j] = cos(pose[b,i] * f_j), f_j = exp(-j * ln(1e4) / (half - 1)) evaluated in f32 as torch does.
__global__ __launch_bounds__(NT) void pos_emb_kernel(const float* __restrict__ pose, float* __restrict__ out, int n, int pose_dim, int half) {
    const int per = pose_dim * half;
    const size_t total = (size_t)n * per;
    const float step = logf(10000.0f) / (float)(half - 1);
    for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
        const int b = (int)(idx / per), r = (int)(idx - (size_t)b * per);
        const int i = r / half, j = r - i * half;
        const float a = pose[(size_t)b * pose_dim + i] * expf((float)j * -step);
        out[(size_t)b * 2 * per + r] = sinf(a);
        out[(size_t)b * 2 * per + per + r] = cosf(a);
    }
}

int launch_pos_emb(const float* pose, float* out, int n, int pose_dim, int classes, hipStream_t s) {
    if (!pose || !out || n <= 0 || pose_dim <= 0 || classes % (2 * pose_dim) || classes / (2 * pose_dim) < 2) return NOPE_ERR_ARG;
    const int half = classes / (2 * pose_dim);
    hipLaunchKernelGGL(pos_emb_kernel, dim3(grid_for((size_t)n * pose_dim * half)), dim3(NT), 0, s, pose, out, n, pose_dim, half);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

// Perspective warp of the dataset-side crop (src/poses/utils.py:204-272: cv2.warpPerspective(img, M, (S, S)), bilinear, constant
// zero border) fused with the image transform of the loader (dataloader/shapeNet.py:64-69: /255, *2-1, HWC -> CHW):
//   dst[c, y, x] = scale * bilinear(src, Minv * (x, y, 1)) + shift,   src (Hs, Ws, C) uint8 or f32, dst (C, Hd, Wd) f32.
// Pixel centres sit on integer coordinates, as in OpenCV; samples outside the source contribute zeros tap by tap.
struct Mat3 { float m[9]; };
// ROUND_U8: the interpolated value is rounded to the nearest integer and clamped to [0, 255] before scale / shift, as a uint8
// destination image does (cv2.warpPerspective on uint8 frames, then ToTensor's / 255): outputs land on the reference loader's
// k / 255 grid.  OpenCV's own fixed-point interpolation weights (1/32 pixel) are not restated: parity unpinned (cv2 is absent).
template <class TIN, bool ROUND_U8>
__global__ __launch_bounds__(NT) void warp_perspective_kernel(const TIN* __restrict__ src, int Hs, int Ws, int C, Mat3 inv, float* __restrict__ dst,
                                                              int Hd, int Wd, float scale, float shift) {
    const int total = Hd * Wd;
    for (int i = blockIdx.x * NT + threadIdx.x; i < total; i += gridDim.x * NT) {
        const int y = i / Wd, x = i - y * Wd;
        const float w = inv.m[6] * x + inv.m[7] * y + inv.m[8];
        const float sx = (inv.m[0] * x + inv.m[1] * y + inv.m[2]) / w;
        const float sy = (inv.m[3] * x + inv.m[4] * y + inv.m[5]) / w;
        const float fx = floorf(sx), fy = floorf(sy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = sx - fx, ay = sy - fy;
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
                const float wt = ((t & 1) ? ax : 1.f - ax) * ((t >> 1) ? ay : 1.f - ay);
                if (xx >= 0 && xx < Ws && yy >= 0 && yy < Hs && w != 0.f) v += wt * (float)src[((size_t)yy * Ws + xx) * C + c];
            }
            if (ROUND_U8) v = fminf(fmaxf(rintf(v), 0.f), 255.f);
            dst[(size_t)c * total + i] = scale * v + shift;
        }
    }
}

int launch_warp_perspective(const void* src, int src_u8, int Hs, int Ws, int C, const float* minv9, float* dst, int Hd, int Wd, float scale,
                            float shift, hipStream_t s) {
    if (!src || !minv9 || !dst || Hs <= 0 || Ws <= 0 || C <= 0 || Hd <= 0 || Wd <= 0) return NOPE_ERR_ARG;
    Mat3 m;
    for (int i = 0; i < 9; ++i) m.m[i] = minv9[i];
    const dim3 grid(grid_for((size_t)Hd * Wd));
    if (src_u8 == 2) hipLaunchKernelGGL((warp_perspective_kernel<unsigned char, true>), grid, dim3(NT), 0, s, (const unsigned char*)src, Hs, Ws, C, m, dst, Hd, Wd, scale, shift);
    else if (src_u8) hipLaunchKernelGGL((warp_perspective_kernel<unsigned char, false>), grid, dim3(NT), 0, s, (const unsigned char*)src, Hs, Ws, C, m, dst, Hd, Wd, scale, shift);
    else hipLaunchKernelGGL((warp_perspective_kernel<float, false>), grid, dim3(NT), 0, s, (const float*)src, Hs, Ws, C, m, dst, Hd, Wd, scale, shift);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_silu_f32(const float* in, float* out, size_t n, hipStream_t s) {
    if (!in || !out) return NOPE_ERR_ARG;
    if (n == 0) return NOPE_OK;
    hipLaunchKernelGGL(silu_f32_kernel, dim3(grid_for(n)), dim3(NT), 0, s, in, out, n);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_linear_naive(const float* in, const float* w, const float* bias, float* out, int M, int N, int K, int act_in, int ldo,
                        hipStream_t s) {
    if (!in || !w || !out || M <= 0 || N <= 0 || K <= 0) return NOPE_ERR_ARG;
    const size_t outs = (size_t)M * N;
    size_t blocks = (outs + 3) / 4;   // 4 waves per block, one output per wave per iteration
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(linear_naive_kernel, dim3((unsigned)blocks), dim3(NT), 0, s, in, w, bias, out, M, N, K, act_in, ldo);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

}  // namespace nope
