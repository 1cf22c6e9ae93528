// Template-bank scoring (the HBM-roofline kernel of the path) and top-k.
//
//   score[b,n] = - sum_{pixels} sqrt( sum_c (q[b,c,p] - t[b,n,c,p])^4 )      model.py:257-262
//
// The bank keeps the API layout (B,N,C,h,w): for a fixed channel the h*w plane is contiguous,
// so lane i of a workgroup owns pixel-vector i (16 bytes) and issues C independent, fully
// coalesced 16-byte loads (1 KiB per wave instruction) -- one per channel plane -- before it
// touches any of them.  The query tile lives in registers for the whole n-loop (it is the
// same for every template of a sample; the reference materialises it N times, model.py:258),
// so HBM traffic is the algorithmic minimum: C*h*w*sizeof(elt) per hypothesis + 4 bytes out.
// Arithmetic is ~1.2 flop/byte: far below the VALU ridge, purely HBM-bound.
// Accumulation is f32; reductions are fixed-shape (wave xor-tree + ordered LDS fold), so
// scores are run-to-run deterministic.
#include <cstdlib>

#include "nope_common.h"

namespace nope {

namespace {

constexpr int NT = 256;

// A lane's share of one channel plane: LV consecutive pixels = W 32-bit words (16 bytes, or 8 for the narrow 16-bit form).
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <int W> struct RawVec { unsigned w[W]; };

// streaming load of bank data: each byte is read exactly once per launch, so it is marked non-temporal (no point in keeping
// it in L2 / Infinity Cache ahead of the query tiles and the next kernel's operands).
template <bool NTL, int W> __device__ __forceinline__ RawVec<W> ld_stream(const void* p) {
    RawVec<W> r;
    if constexpr (W == 4) {
        const u32x4 v = NTL ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) r.w[i] = v[i];
    } else {
        const u32x2 v = NTL ? __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)) : *reinterpret_cast<const u32x2*>(p);
        r.w[0] = v[0]; r.w[1] = v[1];
    }
    return r;
}


template <class T, int W> __device__ __forceinline__ void unpack_words(const RawVec<W>& raw, float* t) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < W; ++i) t[i] = __builtin_bit_cast(float, raw.w[i]);
    } else if constexpr (Elt<T>::DT == NOPE_BF16) {
#pragma unroll
        for (int i = 0; i < W; ++i) {
            t[2 * i] = __builtin_bit_cast(float, raw.w[i] << 16);
            t[2 * i + 1] = __builtin_bit_cast(float, raw.w[i] & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i) {
            union { unsigned u; f16_t h[2]; } x; x.u = raw.w[i];
            t[2 * i] = (float)x.h[0];
            t[2 * i + 1] = (float)x.h[1];
        }
    }
}

template <class T, int LV, int W>
__device__ __forceinline__ void accum_quartic(const RawVec<W>& raw, const float* q, float* acc) {
    float t[LV];
    unpack_words<T, W>(raw, t);
#pragma unroll
    for (int e = 0; e < LV; ++e) {
        const float d = q[e] - t[e];
        const float d2 = d * d;
        acc[e] += d2 * d2;
    }
}

// A lane owns LV consecutive pixels of every channel plane; P = HW / LV lanes make a hypothesis; requires P <= 256 and
// 256 % P == 0.  hpi = 256 / P hypotheses are scored per iteration (thread -> (sub-hypothesis, pixel-vector)).
// LV = 16 bytes' worth.  (LV = 4 for the 16-bit banks -- 8-byte loads, half the registers, five resident workgroups per CU
// instead of three -- is kept as tuning variant 8: measured 4-7 % slower, 8-byte loads do not reach the 16-byte rate.)
// CEXACT: C == CMAX (the shipped descriptor size, 8): the per-channel guards fold away (40 branches of the general form).
// The per-pixel sqrt is the hardware v_sqrt_f32 (1 ulp; sqrt(0) = 0 exactly, so a planted match still scores -0.0): the
// correctly rounded sqrtf expands to ~12 instructions per pixel, a third of the 16-bit path's VALU work, for a 1e-9
// relative change of a 1024-term sum.
// QLDS (tuning variant 16; 16-bit banks, C * HW <= 8192): the query tile of the sample sits in LDS (32 KiB, stored as
// [channel][4-pixel quad parity][lane] so that both 16-byte reads of a lane's 8 pixels are stride-16 across lanes:
// conflict-free) instead of 64 registers per lane: 91-115 VGPRs instead of 124-150.  Measured slower than the register tile
// (the extra 16 LDS reads per hypothesis and lane cost more than the occupancy buys): not the default.
template <class T, int CMAX, bool NTL, int LV, bool CEXACT, bool QLDS = false>
__global__ __launch_bounds__(NT) void sim_reg_kernel(const float* __restrict__ q, const T* __restrict__ bank, float* __restrict__ scores,
                                                     int N, int C, int HW, long long bank_stride_b, int score_ld, int nsplit) {
    constexpr int W = LV * (int)sizeof(T) / 4;
    __shared__ float s_part[2][NT / 64];
    __shared__ __attribute__((aligned(16))) float s_q[QLDS ? 8192 : 4];
    const int P = HW / LV;
    const int hpi = NT / P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = tid / P, pv = tid - sub * P;
    // Workgroup (b, split) scores template groups split, split + nsplit, split + 2 nsplit, ... of sample b (a group = hpi
    // templates): at any moment the resident workgroups of a sample read adjacent groups, i.e. the chip walks B sequential
    // streams through the bank (contiguous per-workgroup ranges, as in round 1, made it thousands of streams).
    const int b = blockIdx.x / nsplit, split = blockIdx.x - b * nsplit;
    const int groups = (N + hpi - 1) / hpi;
    const int g0 = split, g1 = groups, gs = nsplit;

    if constexpr (QLDS) {
        // element (c, pixel p) -> s_q[c * HW + ((p >> 2) & 1) * (HW / 2) + (p >> 3) * 4 + (p & 3)]   (LV == 8)
        const float* qb = q + (size_t)b * C * HW;
        for (int i = tid * 4; i < C * HW; i += NT * 4) {
            const int c = i / HW, p = i - c * HW;
            *reinterpret_cast<f32x4*>(&s_q[c * HW + ((p >> 2) & 1) * (HW / 2) + (p >> 3) * 4]) = *reinterpret_cast<const f32x4*>(qb + i);
        }
        __syncthreads();
    }
    float qr[QLDS ? 1 : CMAX][QLDS ? 1 : LV];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        if (!QLDS && (CEXACT || c < C)) {
            const float* qp = q + ((size_t)b * C + c) * HW + (size_t)pv * LV;
#pragma unroll
            for (int v4 = 0; v4 < LV / 4; ++v4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(qp + 4 * v4);
#pragma unroll
                for (int e = 0; e < 4; ++e) qr[QLDS ? 0 : c][QLDS ? 0 : 4 * v4 + e] = x[e];
            }
        }
    }
    const T* bb = bank + (size_t)b * bank_stride_b;
    const size_t hyp_elems = (size_t)C * HW;
    int buf = 0;
    // Software pipeline, depth 1, in ONE register set: channel plane c of group g + 1 is loaded into raw[c] right after plane c
    // of group g has been consumed, so every lane keeps C loads in flight across the reduction + barrier (without the
    // prefetch the memory pipe of a workgroup drains once per hypothesis) and the kernel needs 32 registers less than with
    // two alternating sets: four resident waves per SIMD for the 16-bit banks instead of three.
    RawVec<W> raw[CMAX];
    auto plane_ptr = [&](int g) {
        const int n = g * hpi + sub;
        return bb + (size_t)(n < N ? n : 0) * hyp_elems + (size_t)pv * LV;
    };
    if (g0 < g1) {
        const T* tp = plane_ptr(g0);
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (CEXACT || c < C) raw[c] = ld_stream<NTL, W>(tp + (size_t)c * HW);
    }
    // (the last group is peeled: a per-plane `if (more)` around the prefetch makes the compiler keep a second register set)
    auto body = [&](int g, auto more_tag) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_tag)::value;
        const T* tn = plane_ptr(MORE ? g + gs : g);
        const int n = g * hpi + sub;
        float acc[LV];
#pragma unroll
        for (int e = 0; e < LV; ++e) acc[e] = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (CEXACT || c < C) {
                if constexpr (QLDS) {
                    float qv[LV];
#pragma unroll
                    for (int h = 0; h < LV / 4; ++h) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(&s_q[c * HW + h * (HW / 2) + pv * 4]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) qv[4 * h + e] = x[e];
                    }
                    accum_quartic<T, LV, W>(raw[c], qv, acc);
                } else {
                    accum_quartic<T, LV, W>(raw[c], qr[c], acc);
                }
                if constexpr (MORE) raw[c] = ld_stream<NTL, W>(tn + (size_t)c * HW);
                __builtin_amdgcn_sched_barrier(0);      // (one plane at a time: unpacking all C planes ahead costs 30-90 VGPRs)
            }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < LV; ++e) s += sizeof(T) == 4 ? sqrtf(acc[e]) : __builtin_amdgcn_sqrtf(acc[e]);   // f32 banks (the parity path): correctly rounded
        // reduce over the P threads of this sub-hypothesis
        if (P >= 64) {
            s = wave_sum(s);
            if (lane == 0) s_part[buf][wave] = s;
            __syncthreads();
            if (pv == 0 && n < N) {
                const int w0 = sub * (P / 64);
                float tot = 0.f;
                for (int w = 0; w < P / 64; ++w) tot += s_part[buf][w0 + w];
                scores[(size_t)b * score_ld + n] = -tot;
            }
            buf ^= 1;   // double-buffered partials: one barrier per iteration
        } else {
            for (int o = P >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
            if (pv == 0 && n < N) scores[(size_t)b * score_ld + n] = -s;
        }
    };
    int g = g0;
    for (; g + gs < g1; g += gs) body(g, BoolTag<true>{});
    if (g < g1) body(g, BoolTag<false>{});
}

// Generic shapes: one hypothesis per iteration, query tile in LDS (C*HW*4 <= 64 KiB).
template <class T>
__global__ __launch_bounds__(NT) void sim_lds_kernel(const float* __restrict__ q, const T* __restrict__ bank, float* __restrict__ scores,
                                                     int N, int C, int HW, long long bank_stride_b, int score_ld, int nsplit) {
    constexpr int VEC = Elt<T>::VEC;
    __shared__ __attribute__((aligned(16))) float s_q[16384];
    __shared__ float s_part[NT / 64];
    const int P = HW / VEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / nsplit, split = blockIdx.x - b * nsplit;
    const int nper = (N + nsplit - 1) / nsplit;
    const int n0 = split * nper;
    const int n1 = (n0 + nper < N) ? n0 + nper : N;
    for (int i = tid; i < C * HW; i += NT) s_q[i] = q[(size_t)b * C * HW + i];
    __syncthreads();
    const T* bb = bank + (size_t)b * bank_stride_b;
    for (int n = n0; n < n1; ++n) {
        const T* tp = bb + (size_t)n * C * HW;
        float s = 0.f;
        for (int pv = tid; pv < P; pv += NT) {
            float acc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
            for (int c = 0; c < C; ++c)
                accum_quartic<T, VEC, 4>(ld_stream<false, 4>(tp + (size_t)c * HW + (size_t)pv * VEC), &s_q[c * HW + pv * VEC], acc);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s += sqrtf(acc[e]);
        }
        s = wave_sum(s);
        if (lane == 0) s_part[wave] = s;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) tot += s_part[w];
            scores[(size_t)b * score_ld + n] = -tot;
        }
        __syncthreads();
    }
}

// ---- top-k: k rounds of (value desc, index asc) arg-max with exclusion ------------------------
constexpr int KMAX = 16;

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// `map` (optional, [B][ld]): the value written to idx is map[b][position] instead of the position -- candidates that carry their own
// (global) template index: the merge of per-shard top-k lists.
__device__ __forceinline__ void topk_row(const float* row, const long long* map_row, long long* __restrict__ idx, float* __restrict__ vals, int b, int N, int k) {
    __shared__ int s_chosen[KMAX];
    __shared__ float s_bv[NT / 64];
    __shared__ int s_bi[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float INF = __builtin_huge_valf();
    for (int r = 0; r < k; ++r) {
        float bv = -INF;
        int bi = 0x7fffffff;
        for (int n = tid; n < N; n += NT) {
            bool taken = false;
            for (int j = 0; j < r; ++j) taken = taken || (s_chosen[j] == n);
            if (taken) continue;
            float v = row[n];
            if (v != v) v = INF;             // NaN ranks highest (torch.topk convention)
            if (bi == 0x7fffffff || better(v, n, bv, bi)) { bv = v; bi = n; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || better(ov, oi, bv, bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_bv[wave] = bv; s_bi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float fv = s_bv[0];
            int fi = s_bi[0];
            for (int w = 1; w < NT / 64; ++w)
                if (s_bi[w] != 0x7fffffff && (fi == 0x7fffffff || better(s_bv[w], s_bi[w], fv, fi))) { fv = s_bv[w]; fi = s_bi[w]; }
            s_chosen[r] = fi;
            idx[(size_t)b * k + r] = map_row ? map_row[fi] : (long long)fi;
            if (vals) vals[(size_t)b * k + r] = row[fi];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT) void topk_kernel(const float* __restrict__ scores, const long long* __restrict__ map, long long* __restrict__ idx,
                                                  float* __restrict__ vals, int N, int k, int ld) {
    const int b = blockIdx.x;
    topk_row(scores + (size_t)b * ld, map ? map + (size_t)b * ld : nullptr, idx, vals, b, N, k);
}

// The tail of a template-sharded step in ONE launch: the all-gathered (G, B, nmax) padded score slices (rank r holds columns
// [lo_r, hi_r) of the contiguous, balanced split: the first N % G ranks one column more) -> the (B, N) similarity every caller of
// `retrieval` gets back (model.py:323,369-375 save it) AND its top-k (model.py:265).  One workgroup per query: a thread copies the columns
// it will rank (same stride in both passes: it only ever reads back its own writes; thread 0's read of the winner's value sits behind
// the round's barrier).
__global__ __launch_bounds__(NT) void gather_topk_kernel(const float* __restrict__ gathered, float* __restrict__ scores, long long* __restrict__ idx,
                                                         float* __restrict__ vals, int G, int B, int N, int nmax, int k) {
    const int b = blockIdx.x;
    const int base = N / G, extra = N - base * G, cut = extra * (base + 1);
    float* row = scores + (size_t)b * N;
    for (int n = threadIdx.x; n < N; n += NT) {
        int r, c;
        if (n < cut) { r = n / (base + 1); c = n - r * (base + 1); }
        else { const int m = n - cut; r = extra + m / base; c = m - (r - extra) * base; }
        row[n] = gathered[((size_t)r * B + b) * nmax + c];
    }
    __syncthreads();
    if (k > 0) topk_row(row, nullptr, idx, vals, b, N, k);
}

}  // namespace

int launch_similarity(const float* q, const void* bank, int bank_dt, float* scores, int B, int N, int C, int HW,
                      long long bank_stride_b, int score_ld, hipStream_t s) {
    if (!q || !bank || !scores || B <= 0 || N <= 0 || C <= 0 || HW <= 0 || score_ld < N || bank_stride_b < 0) return NOPE_ERR_ARG;
    if (bank_dt != NOPE_F32 && bank_dt != NOPE_BF16 && bank_dt != NOPE_F16) return NOPE_ERR_UNSUPPORTED;
    const int vec = bank_dt == NOPE_F32 ? 4 : 8;
    if (HW % vec) return NOPE_ERR_UNSUPPORTED;
    const int variant = NOPE_ENV("NOPE_SIM_VARIANT", 1);          // // tuning: 1 = non-temporal bank loads (0.69 -> 0.75-0.82 of HBM peak), 2 = one residency round of long workgroups, 8 = 4 pixels per lane for 16-bit banks
    // tuning: 16 = query tile in LDS for the 16-bit banks (6 instead of 4 waves per SIMD; measured SLOWER than the register tile with the
    // single-set pipeline: bf16 0.61 / 0.78 of peak at 32 x 512 / 32 x 2048 against 0.72 / 0.84, profiles/r03b_sim_bench.txt)
    const bool qlds = (variant & 16) && (long long)C * HW <= 8192 && HW % 8 == 0;
    // workgroups per sample: ~4096 in all, but at least NOPE_SIM_MINGROUPS template groups each (a workgroup's fixed cost is the 32 KiB
    // query tile: with two groups per workgroup -- 32 x 512 templates -- it is a third of the workgroup's traffic)
    const int min_groups = NOPE_ENV("NOPE_SIM_MINGROUPS", 1);
    const int lv = (bank_dt != NOPE_F32 && (variant & 8) && C <= 8 && HW % 4 == 0 && HW / 4 <= NT) ? 4 : vec;
    const int P = HW / lv;
    const bool reg_ok = (P <= NT) && (NT % P == 0) && (C <= 16);
    int nsplit;
    if (reg_ok) {
        const int hpi = NT / P;
        const int groups = cdiv(N, hpi);
        // nsplit workgroups per sample, ~4096 in all (several residency rounds: the workgroups drift out of phase, which the
        // memory system likes better than one round of long workgroups running load / reduce in lockstep -- NOPE_SIM_VARIANT & 2
        // sizes the grid to exactly one round, CUs x resident workgroups: bf16 +3 %, f32 -9 %, fp16 -6 %, profiles/r02i_sim_bench.txt)
        int cus = 256;                 // (only the tuning variant sizes its grid by the CU count: no device query on the default path)
        if (variant & 2) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        }
#define NOPE_SIM_LAUNCH(T, CM, NTLOAD, LV, EX) NOPE_SIM_LAUNCH_Q(T, CM, NTLOAD, LV, EX, false)
#define NOPE_SIM_LAUNCH_Q(T, CM, NTLOAD, LV, EX, QL)                                                                                       \
        do {                                                                                                                       \
            static int occ = 0;        /* (per instantiation; a property of the kernel's register count) */                       \
            if (occ < 1 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sim_reg_kernel<T, CM, NTLOAD, LV, EX, QL>, NT, 0) != hipSuccess || occ < 1)) occ = 2; \
            nsplit = ((variant & 2) ? cus * occ : 4096) / B;                                                                       \
            if (nsplit > groups / min_groups) nsplit = groups / min_groups;                                                        \
            if (nsplit > groups) nsplit = groups;                                                                                  \
            if (nsplit < 1) nsplit = 1;                                                                                            \
            hipLaunchKernelGGL((sim_reg_kernel<T, CM, NTLOAD, LV, EX, QL>), dim3((unsigned)((long long)B * nsplit)), dim3(NT), 0, s, q, (const T*)bank, scores, \
                               N, C, HW, bank_stride_b, score_ld, nsplit);                                                         \
        } while (0)
#define NOPE_SIM_T(T, LVD)                                                                                              \
        do {                                                                                                            \
            if (lv != LVD) { if (variant & 1) NOPE_SIM_LAUNCH(T, 8, true, 4, false); else NOPE_SIM_LAUNCH(T, 8, false, 4, false); }    \
            else if (variant & 1) { if (C == 8 && LVD == 8 && qlds) NOPE_SIM_LAUNCH_Q(T, 8, true, LVD, true, (LVD == 8)); else if (C == 8) NOPE_SIM_LAUNCH(T, 8, true, LVD, true); else if (C < 8) NOPE_SIM_LAUNCH(T, 8, true, LVD, false); else NOPE_SIM_LAUNCH(T, 16, true, LVD, false); } \
            else { if (C <= 8) NOPE_SIM_LAUNCH(T, 8, false, LVD, false); else NOPE_SIM_LAUNCH(T, 16, false, LVD, false); }             \
        } while (0)
        if (bank_dt == NOPE_F32) NOPE_SIM_T(float, 4);
        else if (bank_dt == NOPE_BF16) NOPE_SIM_T(bf16_t, 8);
        else NOPE_SIM_T(f16_t, 8);
#undef NOPE_SIM_T
#undef NOPE_SIM_LAUNCH
#undef NOPE_SIM_LAUNCH_Q
    } else {
        if ((size_t)C * HW > 16384) return NOPE_ERR_UNSUPPORTED;
        nsplit = cdiv(4096, B);
        if (nsplit > N) nsplit = N;
        dim3 grid((unsigned)((long long)B * nsplit)), block(NT);
        if (bank_dt == NOPE_F32) hipLaunchKernelGGL((sim_lds_kernel<float>), grid, block, 0, s, q, (const float*)bank, scores, N, C, HW, bank_stride_b, score_ld, nsplit);
        else if (bank_dt == NOPE_BF16) hipLaunchKernelGGL((sim_lds_kernel<bf16_t>), grid, block, 0, s, q, (const bf16_t*)bank, scores, N, C, HW, bank_stride_b, score_ld, nsplit);
        else hipLaunchKernelGGL((sim_lds_kernel<f16_t>), grid, block, 0, s, q, (const f16_t*)bank, scores, N, C, HW, bank_stride_b, score_ld, nsplit);
    }
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_topk(const float* scores, long long* idx, float* vals, int B, int N, int k, int ld, hipStream_t s, const long long* map) {
    if (!scores || !idx || B <= 0 || N <= 0 || k < 1 || k > KMAX || k > N || ld < N) return NOPE_ERR_ARG;
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)B), dim3(NT), 0, s, scores, map, idx, vals, N, k, ld);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

int launch_gather_topk(const float* gathered, int G, int B, int N, float* scores, long long* idx, float* vals, int k, hipStream_t s) {
    if (!gathered || !scores || G < 1 || B <= 0 || N <= 0 || k < 0 || k > KMAX || k > N || (k > 0 && !idx)) return NOPE_ERR_ARG;
    const int nmax = (N + G - 1) / G;
    hipLaunchKernelGGL(gather_topk_kernel, dim3((unsigned)B), dim3(NT), 0, s, gathered, scores, idx, vals, G, B, N, nmax, k);
    NOPE_CHECK_LAUNCH();
    return NOPE_OK;
}

}  // namespace nope
