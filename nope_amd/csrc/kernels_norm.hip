// GroupNorm over NHWC activations, in two HBM passes:
//   gn_stats_kernel : per (hypothesis, pixel-chunk, group) partial (sum, sum of squares), f32,
//                     written to a scratch table -- no atomics, so results are run-to-run
//                     deterministic;
//   gn_apply_kernel : folds the chunk partials into (mean, rstd), then
//                     y = act((x - mean) * rstd * gamma + beta) [+ emb[hyp, c]] [+ resid]
// which covers Block.norm + SiLU (model_utils.py:241-252), the pose-embedding add that
// follows block1 (:274-276), the residual add of ResnetBlock (:279), PreNorm's
// GroupNorm(1, C) (:226-234) and Residual (:198-204).  Both passes are pure streaming
// (16-byte loads, every byte touched once), i.e. HBM-bound.
#include "nope_common.h"

namespace nope {

namespace {

constexpr int NT = 256;

// Each thread owns one 16-byte channel vector (tid % tpr) and walks the chunk's pixels with
// stride `rows`.  FASTG: C/VEC <= 256 and channels-per-group is a multiple of VEC, so a
// vector belongs to exactly one group; partials are folded by two fixed-shape LDS trees
// (over pixel rows, then over the vectors of a group) -- deterministic, no atomics.
template <class T, bool FASTG>
__global__ __launch_bounds__(NT) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ partial, int HW, int C,
                                                      int G, int nchunk) {
    constexpr int VEC = Elt<T>::VEC;
    __shared__ float red_s[FASTG ? NT : 2048];
    __shared__ float red_q[FASTG ? NT : 2048];
    const int hyp = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const int tid = threadIdx.x;
    const int pchunk = (HW + nchunk - 1) / nchunk;
    const int p0 = chunk * pchunk;
    const int p1 = (p0 + pchunk < HW) ? p0 + pchunk : HW;
    const int cpg = C / G;
    const T* xb = x + (size_t)hyp * HW * C;
    float* outp = partial + ((size_t)hyp * nchunk + chunk) * G * 2;

    if (FASTG) {
        const int tpr = C / VEC;          // <= NT
        const int rows = NT / tpr;
        const int row = tid / tpr, lc = tid % tpr;
        const int vpg = cpg / VEC;        // vectors per group
        float s = 0.f, q = 0.f;
        if (row < rows) {
            for (int pix = p0 + row; pix < p1; pix += rows) {
                float v[VEC];
                Elt<T>::unpack(ld16(xb + (size_t)pix * C + lc * VEC), v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) { s += v[e]; q += v[e] * v[e]; }
            }
        }
        red_s[tid] = s; red_q[tid] = q;
        __syncthreads();
        int st = 1;
        while (st < rows) st <<= 1;
        for (st >>= 1; st >= 1; st >>= 1) {            // tree over pixel rows
            if (row < st && row + st < rows) {
                red_s[row * tpr + lc] += red_s[(row + st) * tpr + lc];
                red_q[row * tpr + lc] += red_q[(row + st) * tpr + lc];
            }
            __syncthreads();
        }
        const int j = lc % vpg;
        st = 1;
        while (st < vpg) st <<= 1;
        for (st >>= 1; st >= 1; st >>= 1) {            // tree over the vectors of a group
            if (row == 0 && j < st && j + st < vpg) {
                red_s[lc] += red_s[lc + st];
                red_q[lc] += red_q[lc + st];
            }
            __syncthreads();
        }
        for (int g = tid; g < G; g += NT) { outp[g * 2] = red_s[g * vpg]; outp[g * 2 + 1] = red_q[g * vpg]; }
    } else {
        // generic path (C/VEC > 256, tiny or odd channel counts): per-channel sums with
        // coalesced scalar loads, then a per-group fold.  C <= 2048.
        for (int c = tid; c < C; c += NT) {
            float s = 0.f, q = 0.f;
            for (int pix = p0; pix < p1; ++pix) {
                const float v = Elt<T>::ld(xb + (size_t)pix * C + c);
                s += v; q += v * v;
            }
            red_s[c] = s; red_q[c] = q;
        }
        __syncthreads();
        for (int g = tid; g < G; g += NT) {
            float S = 0.f, Q = 0.f;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { S += red_s[c]; Q += red_q[c]; }
            outp[g * 2] = S; outp[g * 2 + 1] = Q;
        }
    }
}

// Per-group sums of per-channel (sum, sum of squares) held in LDS: wave w takes groups w, w + 4, ..., a lane adds every 64th channel
// of the group in channel order and the 64 lane sums are folded by the fixed butterfly of wave_sum -- deterministic, and for the
// single 1536-channel group of a GroupNorm(1) not a 1536-step chain of dependent LDS reads by ONE thread (50 us at 4 x 4 x 64
// hypotheses, profiles/r04d).  Every thread of the workgroup must call it (wave-collective).
template <class Emit>
__device__ __forceinline__ void group_sums(const float* ch_s, const float* ch_q, int cpg, int G, int tid, Emit emit) {
    const int lane = tid & 63;
    for (int g = tid >> 6; g < G; g += NT / 64) {
        float S = 0.f, Q = 0.f;
        for (int c = lane; c < cpg; c += 64) { S += ch_s[g * cpg + c]; Q += ch_q[g * cpg + c]; }
        S = wave_sum(S); Q = wave_sum(Q);
        if (lane == 0) emit(g, S, Q);
    }
}

// Fold [HW/64 row blocks][C][2] column statistics (written by the conv epilogue) of one hypothesis into
// [G][2] group sums.  Fixed summation order -> deterministic.
__global__ __launch_bounds__(NT) void gn_fold_kernel(const float* __restrict__ colstats, float* __restrict__ partial, int nb, int C, int G) {
    __shared__ float ch_s[2048];
    __shared__ float ch_q[2048];
    const int hyp = blockIdx.x, tid = threadIdx.x;
    const float* base = colstats + (size_t)hyp * nb * C * 2;
    for (int c = tid; c < C; c += NT) {
        float s = 0.f, q = 0.f;
        int b = 0;
        for (; b + 8 <= nb; b += 8) {          // 8 independent loads in flight, then the adds in row-block order
            f32x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x2*>(base + ((size_t)(b + u) * C + c) * 2);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += v[u][0]; q += v[u][1]; }
        }
        for (; b < nb; ++b) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(base + ((size_t)b * C + c) * 2);
            s += v[0]; q += v[1];
        }
        ch_s[c] = s; ch_q[c] = q;
    }
    __syncthreads();
    group_sums(ch_s, ch_q, C / G, G, tid, [&](int g, float S, float Q) {
        partial[((size_t)hyp * G + g) * 2] = S;
        partial[((size_t)hyp * G + g) * 2 + 1] = Q;
    });
}

typedef f32x2_t f32x2;

// SiLU of two values in the 16-bit storage modes: v_pk_mul, 2 x v_exp_f32, v_pk_add, 2 x v_rcp_f32, v_pk_mul (the transcendentals
// are quarter rate: 8 of the ~14 issue slots an element costs).  The f32 mode keeps expf and a true division (parity bar 1e-6).
template <bool FAST> __device__ __forceinline__ f32x2 silu2(f32x2 t) {
    if (FAST) {
        const f32x2 a = t * -1.4426950408889634f;
        f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
        e = e + 1.0f;
        const f32x2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
        return t * r;
    }
    return f32x2{silu_f<false>(t.x), silu_f<false>(t.y)};
}

// ACT / RES are template parameters and the arithmetic is written on float pairs: the body is VALU-bound before it is HBM-bound
// (per 16-bit element: unpack, fma, SiLU = 2 quarter-rate transcendentals + 3, adds, clamp, pack), so runtime `if (act)` /
// `if (resid)` selects, the unpack of an absent residual and unpaired adds were 40 % of its issue slots (193 -> 117 VALU per
// two pixels).  U = pixels in flight per thread.
// FOLD: `partial` holds the producing conv's column statistics [x sample][nchunk row blocks][C][2] and every workgroup folds its
// sample's itself, in gn_fold_kernel's order (same bits) -- for small batches, where a separate fold launch costs more than the
// few KiB every workgroup re-reads.
template <class T, bool FAST, bool OS, bool FILM, bool ACT, bool RES, int U, bool FOLD>
__global__ __launch_bounds__(NT) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ partial,
                                                      int nchunk, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int HW, int C, int G, const float* __restrict__ emb, int emb_stride,
                                                      const T* __restrict__ resid, float eps, int blocks_per_hyp, int x_rep, int resid_rep,
                                                      float* __restrict__ out_stats, const float* __restrict__ film, int film_stride,
                                                      unsigned* __restrict__ amax_out) {
    constexpr int VEC = Elt<T>::VEC, V2 = VEC / 2;
    // AMAX (the f32-storage instantiations with the fast SiLU = the split-precision modes): max |y| over what this workgroup writes, for the
    // range shifts of the NOPE_F16X2 convs that consume y (unet_runtime.hip) -- one v_max3_f32 per two values in a kernel that waits for HBM
    constexpr bool AMAX = FAST && sizeof(T) == 4 && !FILM;
    float amax = 0.f;
    __shared__ float s_mean[64];
    __shared__ float s_rstd[64];
    __shared__ float s_os[NT / 64], s_oq[NT / 64];
    const int hyp = blockIdx.x / blocks_per_hyp, blk = blockIdx.x % blocks_per_hyp;
    const int tid = threadIdx.x;
    const int cpg = C / G;
    const int xs = hyp / x_rep;
    if constexpr (FOLD) {
        __shared__ float ch_s[2048];
        __shared__ float ch_q[2048];
        const float* base = partial + (size_t)xs * nchunk * C * 2;
        for (int c = tid; c < C; c += NT) {
            float s = 0.f, q = 0.f;
            int b = 0;
            for (; b + 8 <= nchunk; b += 8) {
                f32x2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x2*>(base + ((size_t)(b + u) * C + c) * 2);
#pragma unroll
                for (int u = 0; u < 8; ++u) { s += v[u][0]; q += v[u][1]; }
            }
            for (; b < nchunk; ++b) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(base + ((size_t)b * C + c) * 2);
                s += v[0]; q += v[1];
            }
            ch_s[c] = s; ch_q[c] = q;
        }
        __syncthreads();
        group_sums(ch_s, ch_q, cpg, G, tid, [&](int g, float S, float Q) {
            const float cnt = (float)cpg * (float)HW;
            const float mean = S / cnt;
            float var = Q / cnt - mean * mean;
            var = var > 0.f ? var : 0.f;
            s_mean[g] = mean;
            s_rstd[g] = 1.0f / sqrtf(var + eps);
        });
    } else
    for (int g = tid; g < G; g += NT) {
        float S = 0.f, Q = 0.f;
        for (int k = 0; k < nchunk; ++k) {
            const float* pp = partial + ((size_t)xs * nchunk + k) * G * 2 + g * 2;
            S += pp[0]; Q += pp[1];
        }
        const float cnt = (float)cpg * (float)HW;
        const float mean = S / cnt;
        float var = Q / cnt - mean * mean;
        var = var > 0.f ? var : 0.f;
        s_mean[g] = mean;
        s_rstd[g] = 1.0f / sqrtf(var + eps);
    }
    __syncthreads();
    // Pure streaming body: a thread owns ONE 16-byte channel vector for its whole pixel walk, so
    // the per-channel affine (rstd*gamma, beta - mean*rstd*gamma) and the embedding live in
    // registers and the inner loop is load -> fma -> SiLU -> adds -> store.
    const int pper = (HW + blocks_per_hyp - 1) / blocks_per_hyp;
    const int p0 = blk * pper;
    const int p1 = p0 + pper < HW ? p0 + pper : HW;
    const T* xb = x + (size_t)xs * HW * C;
    T* yb = y + (size_t)hyp * HW * C;
    const T* rb = RES ? resid + (size_t)(hyp / resid_rep) * HW * C : nullptr;
    const float* eb = emb ? emb + (size_t)hyp * emb_stride : nullptr;
    // FiLM: [scale (C) | shift (C)] of this hypothesis.  Its own instantiation: compiled into the common kernel it cost 16 VGPRs
    // = one resident wave per SIMD, +7 % on every GroupNorm of the default U-Net.
    const float* fb = FILM ? film + (size_t)hyp * film_stride : nullptr;
    const int cvecs = C / VEC;
    const int tpr = cvecs < NT ? cvecs : NT;
    const int rows = NT / tpr;
    const int row = tid / tpr, lc = tid - row * tpr;
    const bool one_group = cpg % VEC == 0;     // a 16-byte vector never straddles two groups (every GroupNorm of the shipped networks)
    f32x2 os2 = {0.f, 0.f}, oq2 = {0.f, 0.f};  // sum / sum of squares of this thread's OUTPUT values (optional out_stats)
    for (int cv = lc; cv < cvecs && row < rows; cv += tpr) {
        f32x2 sc[V2], sh[V2], ev[V2];
        auto coeff = [&](int e, int g) {
            const int c = cv * VEC + e;
            float a = s_rstd[g] * gamma[c];
            float b = beta[c] - s_mean[g] * a;
            if (FILM) {                                // norm(x) * (1 + scale) + shift: still one fma per element
                const float f = 1.0f + fb[c];
                a = a * f;
                b = b * f + fb[C + c];
            }
            sc[e >> 1][e & 1] = a;
            sh[e >> 1][e & 1] = b;
            ev[e >> 1][e & 1] = eb ? eb[c] : 0.f;
        };
        if (one_group) {                               // (a real branch: written as a select, the compiler kept the eight divisions)
            const int g0 = cv * VEC / cpg;
#pragma unroll
            for (int e = 0; e < VEC; ++e) coeff(e, g0);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) coeff(e, (cv * VEC + e) / cpg);
        }
        auto apply = [&](const u32x4 xa, const u32x4 xr) -> u32x4 {
            float v[VEC], r[VEC];
            Elt<T>::unpack(xa, v);
            if (RES) Elt<T>::unpack(xr, r);
#pragma unroll
            for (int q = 0; q < V2; ++q) {
                f32x2 t = f32x2{v[2 * q], v[2 * q + 1]} * sc[q] + sh[q];
                if (ACT) t = silu2<FAST>(t);
                t += ev[q];
                if (RES) t += f32x2{r[2 * q], r[2 * q + 1]};
                v[2 * q] = t.x; v[2 * q + 1] = t.y;
                if (OS) { os2 += t; oq2 += t * t; }
                if (AMAX) amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(t.x)), __builtin_fabsf(t.y));
            }
            return Elt<T>::pack(v);
        };
        const size_t coff = (size_t)cv * VEC;
        int pix = p0 + row;
        for (; pix + (U - 1) * rows < p1; pix += U * rows) {       // U pixels in flight per thread
            u32x4 xa[U], xr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t o = (size_t)(pix + u * rows) * C + coff;
                xa[u] = ld16(xb + o);
                xr[u] = RES ? ld16(rb + o) : xa[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) st16(yb + (size_t)(pix + u * rows) * C + coff, apply(xa[u], xr[u]));
        }
        for (; pix < p1; pix += rows) {
            const size_t o = (size_t)pix * C + coff;
            const u32x4 xa = ld16(xb + o);
            const u32x4 xr = RES ? ld16(rb + o) : xa;
            st16(yb + o, apply(xa, xr));
        }
    }
    if (AMAX && amax_out) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) amax = __builtin_fmaxf(amax, __shfl_xor(amax, o, 64));
        if ((tid & 63) == 0 && amax > 0.f) amax_publish(amax_out, amax, blockIdx.x * (NT / 64) + (tid >> 6));
    }
    if (OS) {
        // per-block (sum, sum of squares) of what was just written: the GroupNorm(1) statistics of the NEXT op
        // (PreNorm of the attention that follows a ResnetBlock) without another pass over the tensor.
        const float os = wave_sum(os2.x + os2.y), oq = wave_sum(oq2.x + oq2.y);
        if ((tid & 63) == 0) { s_os[tid >> 6] = os; s_oq[tid >> 6] = oq; }
        __syncthreads();
        if (tid == 0) {
            float S = 0.f, Q = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) { S += s_os[w]; Q += s_oq[w]; }
            out_stats[(size_t)blockIdx.x * 2] = S;
            out_stats[(size_t)blockIdx.x * 2 + 1] = Q;
        }
    }
}

// (mean, rstd) per hypothesis from `nchunk` (sum, sum of squares) partials of a whole-sample GroupNorm(1).
__global__ __launch_bounds__(NT) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ ms, int nhyp, int nchunk,
                                                         float count, float eps) {
    const int b = blockIdx.x * NT + threadIdx.x;
    if (b >= nhyp) return;
    float S = 0.f, Q = 0.f;
    for (int k = 0; k < nchunk; ++k) { S += partial[((size_t)b * nchunk + k) * 2]; Q += partial[((size_t)b * nchunk + k) * 2 + 1]; }
    const float mean = S / count;
    float var = Q / count - mean * mean;
    var = var > 0.f ? var : 0.f;
    ms[b * 2] = mean;
    ms[b * 2 + 1] = 1.0f / sqrtf(var + eps);
}

}  // namespace

int gn_stats_chunks(int HW, int C, int dt) {
    // aim for >= ~32 KB of streaming per workgroup while keeping thousands of workgroups
    const size_t bytes = (size_t)HW * C * dt_es(dt);
    int n = (int)(bytes / (64 * 1024));
    if (n < 1) n = 1;
    if (n > 16) n = 16;
    while (n > 1 && HW / n < 1) --n;
    return n;
}

int launch_gn_stats(int dt, const void* x, float* partial, int nhyp, int HW, int C, int G, int nchunk, hipStream_t s) {
    if (!x || !partial || nhyp <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G || nchunk < 1) return NOPE_ERR_ARG;
    const int vec = dt_vec(dt);
    if (C % vec) return NOPE_ERR_UNSUPPORTED;
    if (G > 64) return NOPE_ERR_UNSUPPORTED;
    const int cpg = C / G;
    const int cvecs = C / vec;
    const bool fast = (cpg % vec == 0) && (cvecs <= NT);
    if (!fast && C > 2048) return NOPE_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(nhyp * nchunk)), block(NT);
    if (fast) NOPE_DISPATCH_T(dt, T, hipLaunchKernelGGL((gn_stats_kernel<T, true>), grid, block, 0, s, (const T*)x, partial, HW, C, G, nchunk));
    else NOPE_DISPATCH_T(dt, T, hipLaunchKernelGGL((gn_stats_kernel<T, false>), grid, block, 0, s, (const T*)x, partial, HW, C, G, nchunk));
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_gn_fold(const float* colstats, float* partial, int nhyp, int blocks, int C, int G, hipStream_t s) {
    if (!colstats || !partial || nhyp <= 0 || blocks < 1 || C % G || C > 2048 || G > 64) return NOPE_ERR_ARG;
    hipLaunchKernelGGL(gn_fold_kernel, dim3((unsigned)nhyp), dim3(NT), 0, s, colstats, partial, blocks, C, G);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

// ---- GroupNorm + SiLU (+ residual) + a 1x1 projection to <= 8 channels, NCHW out: the tail of the U-Net ---------------------------------
// final_conv = ResnetBlock(dim, dim) -> Conv2d(dim, out_dim, 1) (u_net.py:154-157,197): block2's normalised output (0.4 GB at 512 hypotheses)
// has ONE reader, a 1x1 conv to 8 channels that ran at 0.014 of the matrix peak -- as its own launch it re-read what gn_apply had just written.
// Here the projection is applied where the value is produced.  A wave takes FOUR pixels at a time, 16 lanes each; lane j of a pixel owns the
// 16-byte channel vectors j, j + 16, .. (K of them: C <= 64 K; K = 3 at 192 channels, every lane busy) with their affine coefficients and
// 8 x 4 K projection weights in registers, so a load instruction reads 256 contiguous bytes per pixel.  The 8 partial dot products are folded
// across the 16 lanes with a halving butterfly (exchange 4 values over lane ^ 8, 2 over ^ 4, 1 over ^ 2, then one plain step: 8 cross-lane moves
// per four pixels), and lane 2 o of a pixel writes its output channel o.  (First form, one pixel per wave and one vector per lane: 250
// instructions per pixel, the launch VALU-bound at ~250 us where its 0.8 GB take 150.)  f32 storage + hardware SiLU only (the split-precision
// modes): GroupNorm arithmetic as gn_apply_kernel<float, true, ...> (same folds, same coefficients), the dot product in f32 FMAs where the conv
// kernel used three bf16 MFMA passes -- not bit-identical to the two-launch form, closer to the f32 result.
template <bool RES, int K>
__global__ __launch_bounds__(NT) void gn_apply_proj_kernel(const float* __restrict__ x, const float* __restrict__ colstats, int stat_blocks,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C, int G,
                                                           const float* __restrict__ resid, float eps, int blocks_per_hyp, int resid_rep,
                                                           const float* __restrict__ pw, const float* __restrict__ pb, int OC,
                                                           void* __restrict__ out, int out_dt) {
    __shared__ float s_mean[64];
    __shared__ float s_rstd[64];
    __shared__ float ch_s[256];
    __shared__ float ch_q[256];
    const int hyp = blockIdx.x / blocks_per_hyp, blk = blockIdx.x % blocks_per_hyp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpg = C / G;
    {
        const float* base = colstats + (size_t)hyp * stat_blocks * C * 2;
        for (int c = tid; c < C; c += NT) {          // (gn_apply_kernel's FOLD: same order, same bits)
            float s = 0.f, q = 0.f;
            int b = 0;
            for (; b + 8 <= stat_blocks; b += 8) {
                f32x2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x2*>(base + ((size_t)(b + u) * C + c) * 2);
#pragma unroll
                for (int u = 0; u < 8; ++u) { s += v[u][0]; q += v[u][1]; }
            }
            for (; b < stat_blocks; ++b) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(base + ((size_t)b * C + c) * 2);
                s += v[0]; q += v[1];
            }
            ch_s[c] = s; ch_q[c] = q;
        }
        __syncthreads();
        group_sums(ch_s, ch_q, cpg, G, tid, [&](int g, float S, float Q) {
            const float cnt = (float)cpg * (float)HW;
            const float mean = S / cnt;
            float var = Q / cnt - mean * mean;
            var = var > 0.f ? var : 0.f;
            s_mean[g] = mean;
            s_rstd[g] = 1.0f / sqrtf(var + eps);
        });
    }
    __syncthreads();
    const int pper = (HW + blocks_per_hyp - 1) / blocks_per_hyp;
    const int p0 = blk * pper;
    const int p1 = p0 + pper < HW ? p0 + pper : HW;
    const float* xb = x + (size_t)hyp * HW * C;
    const float* rb = RES ? resid + (size_t)(hyp / resid_rep) * HW * C : nullptr;
    const int j = lane & 15, pl = lane >> 4;             // lane j of pixel slot pl
    bool own[K];
    int c0[K];
    f32x2 sc[K][2], sh[K][2];
    float w[8][K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        own[k] = (j + 16 * k) * 4 < C;                   // this lane holds channels 4 (j + 16 k) .. + 3
        c0[k] = own[k] ? (j + 16 * k) * 4 : 0;
        float ga[4], be[4];
        Elt<float>::unpack(ld16(gamma + c0[k]), ga);
        Elt<float>::unpack(ld16(beta + c0[k]), be);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (c0[k] + e) / cpg;
            const float a = s_rstd[g] * ga[e];
            const float b = be[e] - s_mean[g] * a;
            sc[k][e >> 1][e & 1] = own[k] ? a : 0.f;
            sh[k][e >> 1][e & 1] = own[k] ? b : 0.f;
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) {                    // (one 16-byte load per output row: C % 4 == 0)
            float wv[4];
            Elt<float>::unpack(ld16(pw + (size_t)(o < OC ? o : 0) * C + c0[k]), wv);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[o][k][e] = (own[k] && o < OC) ? wv[e] : 0.f;
        }
    }
    // after the butterfly the even lane 2 o of a pixel holds output channel o
    const int oo = j >> 1;
    const float bias_o = (oo < OC && pb) ? pb[oo] : 0.f;
    const bool hi8 = j & 8, hi4 = j & 4, hi2 = j & 2;
    auto project = [&](const u32x4 (&xa)[K], const u32x4 (&xr)[K], int pix) {
        float d[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) d[o] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float v[4], r[4];
            Elt<float>::unpack(xa[k], v);
            if (RES) Elt<float>::unpack(xr[k], r);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x2 t = f32x2{v[2 * q], v[2 * q + 1]} * sc[k][q] + sh[k][q];
                t = silu2<true>(t);
                if (RES) t += f32x2{r[2 * q], r[2 * q + 1]};
                v[2 * q] = t.x; v[2 * q + 1] = t.y;
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) d[o] += ((v[0] * w[o][k][0] + v[1] * w[o][k][1]) + v[2] * w[o][k][2]) + v[3] * w[o][k][3];
        }
        float e4[4], e2[2], e1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // lanes 8.. of the pixel keep outputs 4..7, lanes ..7 outputs 0..3
            const float give = hi8 ? d[i] : d[4 + i], keep = hi8 ? d[4 + i] : d[i];
            e4[i] = keep + __shfl_xor(give, 8, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float give = hi4 ? e4[i] : e4[2 + i], keep = hi4 ? e4[2 + i] : e4[i];
            e2[i] = keep + __shfl_xor(give, 4, 64);
        }
        {
            const float give = hi2 ? e2[0] : e2[1], keep = hi2 ? e2[1] : e2[0];
            e1 = keep + __shfl_xor(give, 2, 64);
        }
        e1 += __shfl_xor(e1, 1, 64);
        if (!(j & 1) && oo < OC && pix < p1) {
            const float val = e1 + bias_o;
            const size_t o = ((size_t)hyp * OC + oo) * HW + pix;
            if (out_dt == NOPE_F32) reinterpret_cast<float*>(out)[o] = val;
            else if (out_dt == NOPE_F16) reinterpret_cast<f16_t*>(out)[o] = f32_to_f16_sat(val);
            else reinterpret_cast<bf16_t*>(out)[o] = f32_to_bf16(val);
        }
    };
    constexpr int U = 2, PPI = NT / 16;                   // pixel groups in flight per wave; pixels per workgroup iteration
    // (every lane of a wave walks the same number of iterations -- the butterfly is wave-collective; a slot behind the range loads its last pixel again)
    for (int base = p0; base < p1; base += U * PPI) {
        u32x4 xa[U][K], xr[U][K];
        int px[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            px[u] = base + u * PPI + wave * 4 + pl;
            const int pc = px[u] < p1 ? px[u] : p1 - 1;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const size_t o = (size_t)pc * C + c0[k];
                xa[u][k] = ld16(xb + o);
                xr[u][k] = RES ? ld16(rb + o) : xa[u][k];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) project(xa[u], xr[u], px[u]);
    }
}

int gn_apply_blocks(int HW, int C, int dt, int nhyp) {
    // Streaming bytes per workgroup.  Every workgroup first rebuilds (mean, rstd) and its per-channel coefficients (a barrier and
    // ~25 dependent loads): at 32 KiB that set-up was a third of a workgroup's instructions; 16 / 32 / 64 / 128 KiB measured
    // 198 / 143 / 124 / 121 us per statistics + apply pass over 512 x 32 x 32 x 192 f16 (profiles/r03k_gn_apply_ab.txt).
    const int block_kb = NOPE_ENV("NOPE_GN_BLOCK_KB", 64);
    const size_t bytes = (size_t)HW * C * dt_es(dt);
    int bph = (int)(bytes / ((size_t)(block_kb > 0 ? block_kb : 64) * 1024));
    if (bph < 1) bph = 1;
    // Small batches (the reference's 26 / 91-template banks, a 64-template shard): one 64 KiB workgroup per sample leaves 64 workgroups
    // on 256 CUs, each streaming its sample serially (11 us for a 3 MB tensor): spread a sample over more workgroups until the grid
    // has ~512 of them, at least 8 pixels each.  NOPE_GN_MIN_GRID=0 keeps the byte rule alone.
    const int min_grid = NOPE_ENV("NOPE_GN_MIN_GRID", 1024);
    if (nhyp > 0 && (long long)nhyp * bph < min_grid) {
        // (a thread walks its pixels serially, two loads in flight: on a 16-pixel map with 1536 channels one workgroup per sample is eight
        //  dependent memory round trips -- 17 us for 3 MB, profiles/r04d -- so: down to one round trip per thread)
        const int cvecs = C / dt_vec(dt), tpr = cvecs < NT ? cvecs : NT, rows = NT / tpr;
        int max_bph = HW / (2 * rows);
        if (max_bph < 1) max_bph = 1;
        int want = (min_grid + nhyp - 1) / nhyp;
        if (want > max_bph) want = max_bph;
        if (want > bph) bph = want;
    }
    if (bph > 64) bph = 64;
    return bph;
}

int launch_gn_finalize(const float* partial, float* ms, int nhyp, int nchunk, float count, float eps, hipStream_t s) {
    if (!partial || !ms || nhyp <= 0 || nchunk < 1) return NOPE_ERR_ARG;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)cdiv(nhyp, NT)), dim3(NT), 0, s, partial, ms, nhyp, nchunk, count, eps);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_gn_apply(int dt, const GnApplyArgs& a, hipStream_t s) {
    const bool fold = a.colstats != nullptr;
    if (!a.x || !a.y || (!a.partial && !fold) || !a.gamma || !a.beta || a.nhyp <= 0 || a.C % a.G) return NOPE_ERR_ARG;
    if (fold && (a.stat_blocks < 1 || a.C > 2048 || a.film)) return NOPE_ERR_ARG;
    const int vec = dt_vec(dt);
    if (a.C % vec || a.G > 64) return NOPE_ERR_UNSUPPORTED;
    if (a.x_rep < 1 || a.resid_rep < 1) return NOPE_ERR_ARG;
    const int bph = gn_apply_blocks(a.HW, a.C, dt, a.nhyp);
    dim3 grid((unsigned)(a.nhyp * bph)), block(NT);
    const bool act = a.act != 0, res = a.resid != nullptr;
#define NOPE_GN_APPLY_F(T, FAST, OS, FILM, ACT, RES, U, FOLD)                                                                    \
    hipLaunchKernelGGL((gn_apply_kernel<T, FAST, OS, FILM, ACT, RES, U, FOLD>), grid, block, 0, s, (const T*)a.x, (T*)a.y,       \
                       FOLD ? a.colstats : a.partial, FOLD ? a.stat_blocks : a.nchunk, a.gamma, a.beta, a.HW, a.C, a.G, a.emb,   \
                       a.emb_stride, (const T*)a.resid, a.eps, bph, a.x_rep, a.resid_rep, a.out_stats, a.film, a.film_stride, a.amax_out)
#define NOPE_GN_APPLY_U(T, FAST, OS, FILM, ACT, RES, U)                                                                          \
    do { if (!FILM && fold) NOPE_GN_APPLY_F(T, FAST, OS, false, ACT, RES, U, true); else NOPE_GN_APPLY_F(T, FAST, OS, FILM, ACT, RES, U, false); } while (0)
#define NOPE_GN_APPLY_AR(T, FAST, OS, FILM, ACT, RES) NOPE_GN_APPLY_U(T, FAST, OS, FILM, ACT, RES, 2)   /* (4 in flight: +-0) */
#define NOPE_GN_APPLY(T, FAST, OS, FILM)                                                                                         \
    do {                                                                                                                         \
        if (act) { if (res) NOPE_GN_APPLY_AR(T, FAST, OS, FILM, true, true); else NOPE_GN_APPLY_AR(T, FAST, OS, FILM, true, false); }   \
        else     { if (res) NOPE_GN_APPLY_AR(T, FAST, OS, FILM, false, true); else NOPE_GN_APPLY_AR(T, FAST, OS, FILM, false, false); } \
    } while (0)
    if (a.film && a.out_stats) return NOPE_ERR_UNSUPPORTED;
    if (dt == NOPE_F32 && a.fast_silu && !a.film) {
        if (a.out_stats) NOPE_GN_APPLY(float, true, true, false); else NOPE_GN_APPLY(float, true, false, false);
    } else if (dt == NOPE_F32) {
        if (a.film) NOPE_GN_APPLY(float, false, false, true);
        else if (a.out_stats) NOPE_GN_APPLY(float, false, true, false); else NOPE_GN_APPLY(float, false, false, false);
    } else if (dt == NOPE_BF16) {
        if (a.film) NOPE_GN_APPLY(bf16_t, true, false, true);
        else if (a.out_stats) NOPE_GN_APPLY(bf16_t, true, true, false); else NOPE_GN_APPLY(bf16_t, true, false, false);
    } else if (dt == NOPE_F16) {
        if (a.film) NOPE_GN_APPLY(f16_t, true, false, true);
        else if (a.out_stats) NOPE_GN_APPLY(f16_t, true, true, false); else NOPE_GN_APPLY(f16_t, true, false, false);
    } else return NOPE_ERR_UNSUPPORTED;
#undef NOPE_GN_APPLY
#undef NOPE_GN_APPLY_AR
#undef NOPE_GN_APPLY_U
#undef NOPE_GN_APPLY_F
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

// GnApplyArgs::proj_*: the fused tail above.  Eligible: f32 storage with the hardware SiLU (the split-precision modes), SiLU on, statistics folded
// inline (colstats), no embedding / FiLM / output statistics / shared x, C <= 256 in whole 16-byte vectors, <= 8 output channels.
bool gn_apply_proj_ok(int dt, const GnApplyArgs& a) {
    return dt == NOPE_F32 && a.fast_silu && a.act && a.colstats && a.stat_blocks >= 1 && !a.emb && !a.film && !a.out_stats && a.x_rep == 1 &&
           a.C % 4 == 0 && a.C <= 256 && a.G >= 1 && a.G <= 64 && a.C % a.G == 0 && a.proj_cout >= 1 && a.proj_cout <= 8 && a.proj_w && a.proj_out &&
           (a.proj_out_dt == NOPE_F32 || a.proj_out_dt == NOPE_F16 || a.proj_out_dt == NOPE_BF16) && NOPE_ENV("NOPE_FINAL_FUSED", 1) != 0;
}
int launch_gn_apply_proj(int dt, const GnApplyArgs& a, hipStream_t s) {
    if (!a.x || !a.gamma || !a.beta || a.nhyp <= 0 || a.resid_rep < 1 || !gn_apply_proj_ok(dt, a)) return NOPE_ERR_ARG;
    int bph = 1;
    {   // a workgroup iteration is 2 x 16 pixels, and a workgroup's set-up (statistics fold, two barriers, 8 x 4 K weights + coefficients per lane) costs
        // several iterations: NOPE_PROJ_PIXELS pixels per workgroup (whole iterations), fewer only to keep ~1024 workgroups in the grid
        int pper = NOPE_ENV("NOPE_PROJ_PIXELS", 256);
        while (pper > 32 && (long long)a.nhyp * ((a.HW + pper - 1) / pper) < 1024) pper >>= 1;
        if (pper > a.HW) pper = a.HW;
        pper = pper >= 32 ? pper / 32 * 32 : (pper > 16 ? 32 : 16);
        bph = (a.HW + pper - 1) / pper;
    }
    dim3 grid((unsigned)(a.nhyp * bph)), block(NT);
    const int K = (a.C + 63) / 64;                                    // 16-byte channel vectors per lane
#define NOPE_GN_PROJ(RES, KK)                                                                                                                    \
    hipLaunchKernelGGL((gn_apply_proj_kernel<RES, KK>), grid, block, 0, s, (const float*)a.x, a.colstats, a.stat_blocks, a.gamma, a.beta, a.HW, a.C, a.G, \
                       (const float*)a.resid, a.eps, bph, RES ? a.resid_rep : 1, a.proj_w, a.proj_b, a.proj_cout, a.proj_out, a.proj_out_dt)
#define NOPE_GN_PROJ_K(RES)                                                                                                                      \
    do { if (K == 1) NOPE_GN_PROJ(RES, 1); else if (K == 2) NOPE_GN_PROJ(RES, 2); else if (K == 3) NOPE_GN_PROJ(RES, 3); else NOPE_GN_PROJ(RES, 4); } while (0)
    if (a.resid) NOPE_GN_PROJ_K(true); else NOPE_GN_PROJ_K(false);
#undef NOPE_GN_PROJ_K
#undef NOPE_GN_PROJ
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

}  // namespace nope
