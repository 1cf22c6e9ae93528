// Shared device/host helpers for the gfx950 kernels of the NOPE hot path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <atomic>
#include <cstdint>

#include "../../include/nope_hip.h"

namespace nope {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned short bf16_t;   // raw bfloat16 bits
typedef _Float16 f16_t;          // IEEE half (bank storage, NOPE_F16 compute mode): hardware v_cvt_f16_f32 / v_cvt_f32_f16, round-to-nearest-even
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int kWave = 64;

// ---- bfloat16 <-> f32 (round-to-nearest-even, NaN preserved) ----------------------------
__device__ __host__ __forceinline__ float bf16_to_f32(bf16_t h) {
    union { unsigned u; float f; } v;
    v.u = (unsigned)h << 16;
    return v.f;
}
__device__ __host__ __forceinline__ bf16_t f32_to_bf16(float f) {
    union { unsigned u; float f; } v;
    v.f = f;
    unsigned u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// Two f32 -> packed bf16x2 in one v_cvt_pk_bf16_f32 (gfx950 hardware RNE; same results as f32_to_bf16).
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
typedef float f32x2_hw_t __attribute__((ext_vector_type(2)));
typedef f32x2_hw_t f32x2_t;       // float pairs: arithmetic on them compiles to v_pk_add / v_pk_mul / v_pk_fma_f32
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    const f32x2_hw_t v = {lo, hi};
    union { bf16x2_hw_t h; unsigned u; } x;
    x.h = __builtin_convertvector(v, bf16x2_hw_t);
    return x.u;
}

// f32 -> f16 SATURATES: values beyond the format's range become +-65504 instead of +-inf (one v_med3_f32 per element).  The f16
// compute mode stores every activation through these two functions; an overflowing conv output (possible with real checkpoints:
// nothing bounds a pre-GroupNorm activation) then costs precision on that element instead of turning the sample's GroupNorm
// statistics, and with them the whole embedding map, into NaN.  v_med3_f32 returns MIN3 when an operand is NaN, so a NaN is stored as
// -65504: the f16 mode does not carry NaN inputs through (the f32 / bf16x3 / bf16 modes do).
constexpr float kF16Max = 65504.0f;
__device__ __forceinline__ float sat_f16(float v) { return __builtin_amdgcn_fmed3f(v, -kF16Max, kF16Max); }
__device__ __forceinline__ _Float16 f32_to_f16_sat(float v) { return (_Float16)sat_f16(v); }
// Two f32 -> packed f16x2 (saturating, round-to-nearest-even; v_cvt_pk_f16_f32 on gfx950)
typedef _Float16 f16x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_f16_raw(float lo, float hi) {      // no clamp: the caller knows |lo|, |hi| <= 65504
    const f32x2_hw_t v = {lo, hi};
    union { f16x2_hw_t h; unsigned u; } x;
    x.h = __builtin_convertvector(v, f16x2_hw_t);
    return x.u;
}
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) { return cvt_pk_f16_raw(sat_f16(lo), sat_f16(hi)); }

// compile-time flag passed by value to a generic lambda (decltype(tag)::value)
template <bool V> struct BoolTag { static constexpr bool value = V; };

// Element traits: VEC = elements per 16-byte vector.
template <class T> struct Elt;
template <> struct Elt<float> {
    static constexpr int VEC = 4;
    static constexpr int DT = NOPE_F32;
    static constexpr bool SATURATES = false;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* o) {
        union { u32x4 u; float f[4]; } x; x.u = v;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = x.f[i];
    }
    static __device__ __forceinline__ u32x4 pack(const float* o) {
        union { u32x4 u; float f[4]; } x;
#pragma unroll
        for (int i = 0; i < 4; ++i) x.f[i] = o[i];
        return x.u;
    }
};
template <> struct Elt<bf16_t> {
    static constexpr int VEC = 8;
    static constexpr int DT = NOPE_BF16;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* o) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            union { unsigned u; float f; } lo, hi;
            lo.u = v[i] << 16;
            hi.u = v[i] & 0xffff0000u;
            o[2 * i] = lo.f;
            o[2 * i + 1] = hi.f;
        }
    }
    static __device__ __forceinline__ u32x4 pack(const float* o) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = cvt_pk_bf16(o[2 * i], o[2 * i + 1]);
        return v;
    }
    static __device__ __forceinline__ unsigned cvt_pk(float lo, float hi) { return cvt_pk_bf16(lo, hi); }
    static constexpr bool SATURATES = false;     // (bf16 has f32's exponent range)
    static __device__ __forceinline__ unsigned cvt_pk_raw(float lo, float hi) { return cvt_pk_bf16(lo, hi); }
};

template <> struct Elt<f16_t> {
    static constexpr int VEC = 8;
    static constexpr int DT = NOPE_F16;
    static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void st(f16_t* p, float v) { *p = f32_to_f16_sat(v); }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* o) {
        union { u32x4 u; f16_t h[8]; } x; x.u = v;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)x.h[i];
    }
    static __device__ __forceinline__ u32x4 pack(const float* o) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = cvt_pk_f16(o[2 * i], o[2 * i + 1]);
        return v;
    }
    static __device__ __forceinline__ unsigned cvt_pk(float lo, float hi) { return cvt_pk_f16(lo, hi); }
    static constexpr bool SATURATES = true;      // cvt_pk clamps; cvt_pk_raw is for values known to be in range
    static __device__ __forceinline__ unsigned cvt_pk_raw(float lo, float hi) { return cvt_pk_f16_raw(lo, hi); }
};

// NOPE_BF16X3: activations are plain f32 in memory; the tag type only selects the conv kernels' split-precision MFMA tile
// (conv_gemm_common.h) and the (hi, lo) weight layout of the pack kernels.  Everything else treats the data as float.
struct f32s_t { float f; };
template <> struct Elt<f32s_t> {
    static constexpr int VEC = 4;
    static constexpr int DT = NOPE_BF16X3;
    static constexpr bool SATURATES = false;
    static __device__ __forceinline__ float ld(const f32s_t* p) { return p->f; }
    static __device__ __forceinline__ void st(f32s_t* p, float v) { p->f = v; }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* o) { Elt<float>::unpack(v, o); }
    static __device__ __forceinline__ u32x4 pack(const float* o) { return Elt<float>::pack(o); }
};

// NOPE_F16X2: f32 activations as NOPE_BF16X3; the tag selects the ping-pong kernels' f16 + MX-fp8 tile (Tile<f16x2_t>, conv_gemm_common.h:
// conv3x3_halo_kernel and conv_gemm_pp_kernel) and its weight layout (launch_pack_conv_w_x2, kernels_misc.hip).  No other kernel is instantiated for it.
struct f16x2_t { float f; };
template <> struct Elt<f16x2_t> {
    static constexpr int VEC = 4;
    static constexpr int DT = NOPE_F16X2;
    static constexpr bool SATURATES = false;
    static __device__ __forceinline__ float ld(const f16x2_t* p) { return p->f; }
    static __device__ __forceinline__ void st(f16x2_t* p, float v) { p->f = v; }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* o) { Elt<float>::unpack(v, o); }
    static __device__ __forceinline__ u32x4 pack(const float* o) { return Elt<float>::pack(o); }
};
// Two f32 -> two OCP e4m3 bytes (bits 0..15 of the result; v_cvt_pk_fp8_f32: round to nearest even, subnormals kept, 480 and beyond
// become NaN -- tools/probes/mx_probe.hip -- hence the clamp to the format's largest value first, as HIP's own fp8 header does)
constexpr float kE4M3Max = 448.0f;
__device__ __forceinline__ unsigned cvt_pk_e4m3(float lo, float hi) {
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(lo, -kE4M3Max, kE4M3Max), __builtin_amdgcn_fmed3f(hi, -kE4M3Max, kE4M3Max), 0, false) & 0xffffu;
}
// Four f32 -> four e4m3 bytes of (x * 2^shift), one dword: v_cvt_scalef32_pk_fp8_f32 divides by the power of two in its scale operand (only
// the exponent counts) and writes the half of the destination word_sel names, so pre-scale, conversion and packing are two instructions
// + the clamps (the instruction does not saturate either: probe fact 6).  shift is a compile-time constant.
typedef short s16x2_hw_t __attribute__((ext_vector_type(2)));
// OVFL_MODE: the caller runs with MODE.FP16_OVFL = 1 (fp16_ovfl_on below), under which the conversion saturates by itself (probe fact 7): no clamps.
template <int SHIFT, bool OVFL_MODE = false> __device__ __forceinline__ unsigned cvt4_e4m3_scaled(float x0, float x1, float x2, float x3) {
    constexpr float LIM = kE4M3Max / (SHIFT >= 0 ? (float)(1 << (SHIFT >= 0 ? SHIFT : 0)) : 1.0f / (float)(1 << (SHIFT < 0 ? -SHIFT : 0)));      // 448 * 2^-SHIFT
    constexpr float INV = SHIFT >= 0 ? 1.0f / (float)(1 << (SHIFT >= 0 ? SHIFT : 0)) : (float)(1 << (SHIFT < 0 ? -SHIFT : 0));                     // 2^-SHIFT: the divisor
    s16x2_hw_t r = {0, 0};
    if constexpr (OVFL_MODE) {
        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x0, x1, INV, false);
        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x2, x3, INV, true);
    } else {
        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, __builtin_amdgcn_fmed3f(x0, -LIM, LIM), __builtin_amdgcn_fmed3f(x1, -LIM, LIM), INV, false);
        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, __builtin_amdgcn_fmed3f(x2, -LIM, LIM), __builtin_amdgcn_fmed3f(x3, -LIM, LIM), INV, true);
    }
    return __builtin_bit_cast(unsigned, r);
}
// ... with the divisor in a register (only its exponent counts); MODE.FP16_OVFL form only
__device__ __forceinline__ unsigned cvt4_e4m3_div(float x0, float x1, float x2, float x3, float divisor) {
    s16x2_hw_t r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x0, x1, divisor, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x2, x3, divisor, true);
    return __builtin_bit_cast(unsigned, r);
}
// A range word under many writers.  One device word per tensor would take an atomic from every wave of every producer launch -- 24 576 same-
// address atomics per level-0 gn_apply, measured +0.3 ms PER LAUNCH (profiles/r06c_*) -- so a slot is kX2Spread words, one per 128-byte line,
// a wave picks one by its index and only issues the atomic when its value beats what the word already holds (a relaxed load: a stale
// value costs an unnecessary atomic, never a wrong maximum); amax_reduce_kernel (kernels_misc.hip) folds the lines before the host reads them.
constexpr int kX2Spread = 32, kX2SlotWords = kX2Spread * 32;
__device__ __forceinline__ void amax_publish(unsigned* slot, float m, unsigned who) {
    unsigned* p = slot + (who & (kX2Spread - 1)) * 32;
    const unsigned b = __builtin_bit_cast(unsigned, m);
    if (b > __atomic_load_n(p, __ATOMIC_RELAXED)) atomicMax(p, b);
}
// running max |x| of four values (two v_max3_f32 with |.| modifiers; a NaN is ignored: it stays a NaN in the result either way)
__device__ __forceinline__ float amax4(float m, float x0, float x1, float x2, float x3) {
    m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(x0)), __builtin_fabsf(x1));
    return __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(x2)), __builtin_fabsf(x3));
}
// MODE.FP16_OVFL (hwreg 1 = MODE, bit 23) for the rest of the wave's life: f32 -> f16 and f32 -> fp8 conversions SATURATE (+-65504 / +-448)
// instead of producing inf / NaN; a NaN stays a NaN (tools/probes/mx_probe.hip fact 7).  The f16x2 operand rewrite runs under it: 12 clamps
// per piece and lane gone.  NOPE_CVT_PK_F16_OVFL: the f16 pair conversion under that mode -- the plain one on the device; tests/hipemu
// supplies its own, which consults the interpreter's copy of the mode bit (a host compiler's conversion knows nothing of it).
__device__ __forceinline__ void fp16_ovfl_on() { __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }
#ifndef NOPE_CVT_PK_F16_OVFL
#define NOPE_CVT_PK_F16_OVFL(lo, hi) cvt_pk_f16_raw(lo, hi)
#endif
// Power-of-two pre-scales of the f16x2 cross-term operands (exact multiplications), undone by the MFMA's E8M0 block scale:
//   e4m3(a_lo * 2^9), e4m3(a * 2^-2)  [activations, fixed]      e4m3(w * 2^sw), e4m3(w_lo * 2^(sw + 11))  [weights, sw per layer]
// chosen so that BOTH products of the K-concatenated instruction, a_lo w and a w_lo, carry the same total 2^(9 + sw): one scale for all
// lanes (the instruction's scale blocks follow byte positions, not lane halves: tools/probes/mx_probe.hip fact 3h).  |a| up to 1792
// and |a_lo| of |a| up to 2048 stay below the clamp; beyond, the element degrades towards plain f16 accuracy.
// RANGE SHIFT.  A layer's activations enter the tile as a' = a * 2^-t (an exact multiplication; t per layer: tail word 3, 0 as packed): f16(a'),
// e4m3(a'_lo * 2^9), e4m3(a' * 2^-2) -- in instructions: one packed multiply per two elements for the f16 part, a'_lo = fma(a, 2^-t, -hi) in place
// of the subtraction, and t added to the divisor the e4m3(a') conversion takes anyway: +2 VALU per four elements -- so the window of full accuracy -- |a'| <= 1792 for the e4m3(a') operand, <= 65504 for the f16 part, and
// >= 2^-4 for four significant bits in e4m3(a') -- moves with t; the accumulators then hold 2^-t x the convolution and the epilogue multiplies by
// 2^t (exact) before bias / statistics / residual.  The kernels record max |a| of what they converted (ConvParams::x2_amax) and the runtime that
// owns the layer moves t when a launch left the window (unet_runtime.hip: nope_unet_x2_range_check); operator-level launches run at t = 0.
// (NOPE_X2_TRACK = 0 at compile time: t == 0 folded in -- the round-5 form of the rewrite, kept for same-box A/B timing:
//  python -m nope_amd.csrc.build --variant notrack -DNOPE_X2_TRACK=0.  NOPE_X2_KERNEL_AMAX = 1: the conv kernels themselves record max |a| of
//  what they convert into ConvParams::x2_amax -- measured +5 % on the 512-template step, profiles/r06c_*: the U-Net runtime takes the maxima from
//  the PRODUCERS of its tensors instead, gn_apply's spare VALU slots and absmax passes over the few conv-produced ones; off by default.)
#ifndef NOPE_X2_TRACK
#define NOPE_X2_TRACK 1
#endif
#ifndef NOPE_X2_KERNEL_AMAX
#define NOPE_X2_KERNEL_AMAX 0
#endif
constexpr int kX2ALoShift = 9, kX2AShift = -2, kX2WLoExtra = 11;
constexpr int kX2TailBytes = 16;      // behind the packed weights: int A-scale byte (127 - 9 - sw), int sw, float max |w|, int t (range shift of the activations)
constexpr float kX2AMaxFull = 1792.f;  // |a| * 2^-t above this saturates e4m3(a * 2^(-2 - t)) = 448
__device__ __forceinline__ float x2_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }      // 2^e, |e| <= 126

// dtype code -> bytes per stored element / elements per 16-byte vector / the dtype the non-conv kernels see
static inline int dt_es(int dt) { return (dt == NOPE_F32 || dt == NOPE_BF16X3 || dt == NOPE_F16X2) ? 4 : 2; }
static inline int dt_vec(int dt) { return 16 / dt_es(dt); }
static inline int dt_storage(int dt) { return (dt == NOPE_BF16X3 || dt == NOPE_F16X2) ? NOPE_F32 : dt; }
static inline bool dt_is_compute(int dt) { return dt == NOPE_F32 || dt == NOPE_BF16 || dt == NOPE_F16 || dt == NOPE_BF16X3 || dt == NOPE_F16X2; }
// the element type every launch EXCEPT the tap-resident 3x3 kernel sees under a compute mode (NOPE_F16X2 is NOPE_BF16X3 there)
static inline int dt_base(int dt) { return dt == NOPE_F16X2 ? NOPE_BF16X3 : dt; }

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// x * gelu(gate), exact (erf) GELU: one definition for geglu_kernel and the fused conv epilogue
__device__ __forceinline__ float geglu_f(float x, float g) { return x * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f))); }

template <bool FAST> __device__ __forceinline__ float silu_f(float x) {
    if (FAST) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));   // v_exp_f32 + v_rcp_f32 (1 ulp each)
    return x / (1.0f + expf(-x));
}

// Keeps a value (and the loads that produced it) alive in a tuning build that skips its consumer -- without this the
// compiler deletes the loads too and the ablation measures less than it claims (cdna_hip_programming.md rule 17).
#ifndef NOPE_KEEP_VGPR
#define NOPE_KEEP_VGPR(x) asm volatile("" ::"v"(x))
#endif
// Makes a per-lane value opaque at this point: nothing computed from it can be hoisted above (out of a loop) -- the handle on
// loop-invariant code motion when the hoisted values would not fit the register file (cdna_hip_programming.md section 5.7, item 3).
#ifndef NOPE_OPAQUE_VGPR
#define NOPE_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#endif

// s_waitcnt vmcnt(n) where the n youngest outstanding operations are PLAIN loads (issued after the LDS-DMA pieces being waited for).
// tests/hipemu executes plain loads synchronously and does not count them: there the wait is vmcnt(0).
#ifdef HIPEMU
#define NOPE_WAIT_VMCNT_KEEP_LOADS(n) __builtin_amdgcn_s_waitcnt(0x0F70)
#else
#define NOPE_WAIT_VMCNT_KEEP_LOADS(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
#endif

// Run a statement with T bound to the element type of a STORAGE dtype code (f32 / bf16 / f16); returns NOPE_ERR_UNSUPPORTED
// from the enclosing function for anything else.
#define NOPE_DISPATCH_T(dt, T, ...)                                              \
    do {                                                                         \
        if ((dt) == NOPE_F32) { typedef float T; __VA_ARGS__; }                  \
        else if ((dt) == NOPE_BF16) { typedef bf16_t T; __VA_ARGS__; }           \
        else if ((dt) == NOPE_F16) { typedef f16_t T; __VA_ARGS__; }             \
        else return NOPE_ERR_UNSUPPORTED;                                        \
    } while (0)
// ... and of a weight-pack dtype, which adds NOPE_BF16X3's (hi, lo) layout
#define NOPE_DISPATCH_W(dt, T, ...)                                              \
    do {                                                                         \
        if ((dt) == NOPE_BF16X3) { typedef f32s_t T; __VA_ARGS__; }              \
        else NOPE_DISPATCH_T(dt, T, __VA_ARGS__);                                \
    } while (0)

// ---- tuning / test switches (NOPE_* environment variables) -----------------------------------------------------------------------
// Read ONCE per call site and cached: a launch does not walk the environment (a conv launch consulted ~40 variables).  nope_tuning_reload()
// (C ABI) starts a new generation, after which every site reads its variable again -- nope_amd/hip.py calls it whenever it sees a NOPE_*
// variable change between two calls into the library, so in-process A/B sweeps and the tests that flip a switch keep working; a C caller
// that changes the environment after the first launch calls it itself.  NOPE_ENV(name, default) -> int (the value, or the default
// when unset); NOPE_ENV_LL -> long long (byte thresholds); NOPE_ENV_SET(name) -> bool (is it set at all).  No site wraps a lookup in a
// `static const`: every one of them follows a reload.
unsigned tuning_generation();                          // capi.hip
// One cache per call site.  Every field is an atomic: a site may be refreshed by one host thread (under capi.hip's mutex) while another
// reads it; `gen` is published with release order AFTER val / set, and read with acquire order BEFORE them, so a reader that sees the
// current generation sees that generation's (or a newer one's) value -- never a torn or stale one.
struct EnvCache { std::atomic<unsigned> gen{0xffffffffu}; std::atomic<bool> set{false}; std::atomic<long long> val{0}; };
struct EnvVal { bool set; long long val; };
EnvVal env_lookup(EnvCache& c, const char* name);      // refreshes c when the generation moved; {is it set, atoll of its value (0 when unset)}
#define NOPE_ENV(name, def) ({ static ::nope::EnvCache nope_env_c__; const ::nope::EnvVal nope_env_v__ = ::nope::env_lookup(nope_env_c__, name); nope_env_v__.set ? (int)nope_env_v__.val : (def); })
#define NOPE_ENV_LL(name, def) ({ static ::nope::EnvCache nope_env_c__; const ::nope::EnvVal nope_env_v__ = ::nope::env_lookup(nope_env_c__, name); nope_env_v__.set ? nope_env_v__.val : (long long)(def); })
#define NOPE_ENV_SET(name) ({ static ::nope::EnvCache nope_env_c__; ::nope::env_lookup(nope_env_c__, name).set; })

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define NOPE_CHECK_LAUNCH()                                      \
    do {                                                         \
        hipError_t e__ = hipGetLastError();                      \
        if (e__ != hipSuccess) return NOPE_ERR_LAUNCH;           \
    } while (0)

// ---- launchers implemented in the kernels_*.hip files (host side) ---------------------
struct ConvArgs {
    const void* src1 = nullptr; const void* src2 = nullptr;   // NHWC activations
    int C1 = 0, C2 = 0;          // channels per source; Cin = C1 + C2 (virtual concat, src1 first)
    int rep1 = 1, rep2 = 1;      // source sample = hypothesis / rep  (broadcast of shared tensors)
    int Hs = 1, Ws = 1;          // source spatial size
    int Ho = 1, Wo = 1;          // output spatial size
    int mode = NOPE_CONV_PLAIN;  // NOPE_CONV_PLAIN | UP2 | DOWN2 | UP2P | STRIDE2
    int ntaps = 1;               // 1 or 9 (PLAIN, STRIDE2), 9 (UP2), 4 (DOWN2, UP2P)
    const void* w = nullptr;     // packed [Cout][ntaps][Cin]
    const void* w_x2 = nullptr;  // NOPE_BF16X3 launches only: the same weights in the NOPE_F16X2 layout (launch_pack_conv_w_x2) -- taken, with
                                 // the f16 + MX-fp8 tile, when the launch goes to a ping-pong kernel (conv_takes_x2); `w` (may be null then) otherwise
    unsigned* x2_amax = nullptr; // NOPE_F16X2 launches: optional device word, atomicMax of the bits of max |a| over the A elements converted (NOPE_X2_KERNEL_AMAX builds)
    int x2_t_zero = 0;           // NOPE_F16X2 launches: 1 = the caller knows the pack's range shift (tail word 3) is 0 (the operator-level entry never sets it)
    unsigned* out_amax = nullptr;// f32 storage: optional range slot (kX2SlotWords words, amax_publish) for max |out| -- honoured by the launches that end in the wide NHWC
                                 // epilogue of the 128 x 192 / ping-pong / tap-resident kernels without split-K (conv_records_out_amax); ignored otherwise
    const float* bias = nullptr; // [Cout] or null
    const void* resid = nullptr; // optional NHWC [M][Cout] added in the epilogue
    void* out = nullptr;
    int Cout = 0;
    int nhyp = 0;                // M = nhyp * Ho * Wo
    int out_nchw = 0;            // 1: write (nhyp, Cout, Ho, Wo) with dtype out_dt
    int out_dt = NOPE_F32;       // only for out_nchw
    int act = 0;                 // 0 none, 1 ReLU applied after bias (+ residual)
    void* splitk_ws = nullptr;   // optional f32 scratch [splits][M][Cout]: allows a deterministic split-K launch for
    size_t splitk_bytes = 0;     //   small-M / long-K problems (conv_splitk_factor); ignored when too small
    int force_generic = 0;       // tests: take the register-staged kernel even when the LDS-DMA one applies
    float* colstats = nullptr;   // optional fused GroupNorm statistics: [M/stat_rows][Cout][2] (needs M % stat_rows == 0, NHWC out)
    int stat_rows = 64;          // rows per statistics block: 64, or 16 / 32 on the small-tile kernel (maps of 16 / 32 pixels: conv_stat_rows)
    // optional fused PreNorm (GroupNorm(1) in front of a 1x1 conv whose weights already carry gamma):
    //   out[m,n] = rstd[b] * (acc[m,n] - mean[b] * pn_c1[n]) + pn_c0[n],  b = sample of row m
    const float* pn_ms = nullptr;   // [nhyp][2] (mean, rstd)
    const float* pn_c0 = nullptr;   // [Cout]  sum_c W[n,c] * beta[c]
    const float* pn_c1 = nullptr;   // [Cout]  sum_c W[n,c] * gamma[c]
    // GEGLU in the epilogue (the LDM variant's feed-forward projection, ldm/attention.py:37-44): output columns are (x_j, gate_j) pairs --
    // the caller interleaved the weight rows -- and out is [M][Cout / 2] = x_j * gelu(gate_j).  16-bit types, 1x1, 128 x 192 kernel only
    // (conv_geglu_fusable); same operands (rounded to the storage type) and same formula as geglu_kernel: bit-identical to conv + geglu.
    int geglu = 0;
};
int launch_conv(int dt, const ConvArgs& a, hipStream_t s);
bool conv_geglu_fusable(int dt, const ConvArgs& a);   // would launch_conv take this conv with geglu = 1?
int conv_splitk_factor(int dt, const ConvArgs& a);
bool conv_is_posmajor(int dt, const ConvArgs& a);
int conv_kernel_kind(int dt, const ConvArgs& a);      // NOPE_CONV_KERNEL_* launch_conv would pick
bool conv_takes_x2(int dt, const ConvArgs& a);        // would launch_conv run this NOPE_BF16X3 launch on the f16 + MX-fp8 tile (a.w_x2 set)?
int conv_stat_rows(int dt, const ConvArgs& a);        // rows per block of the fused column statistics this conv can emit (0: none)
double conv_executed_flops(int dt, const ConvArgs& a);

int launch_gn_stats(int dt, const void* x, float* partial, int nhyp, int HW, int C, int G, int nchunk, hipStream_t s);
struct GnApplyArgs {
    const void* x = nullptr; void* y = nullptr;
    const float* partial = nullptr; int nchunk = 1;
    const float* colstats = nullptr; int stat_blocks = 0;   // instead of `partial`: the producing conv's column statistics [x sample][stat_blocks][C][2],
                                                            // folded by every workgroup itself (small batches: no gn_fold launch)
    const float* gamma = nullptr; const float* beta = nullptr;
    int nhyp = 0, HW = 0, C = 0, G = 1;
    int act = 0;                       // 1 = SiLU
    const float* emb = nullptr; int emb_stride = 0;   // optional per-(hyp, channel) add after the activation
    const float* film = nullptr; int film_stride = 0; // optional FiLM rows [scale (C) | shift (C)] per hypothesis (stride 0: one row for
                                                      // all): y = act(norm(x) * (1 + scale) + shift)
    const void* resid = nullptr;       // optional NHWC tensor added last
    int x_rep = 1;                     // x (and its statistics) shared by x_rep consecutive hypotheses
    int resid_rep = 1;                 // resid shared by resid_rep consecutive hypotheses
    float* out_stats = nullptr;        // optional [nhyp][gn_apply_blocks()][2]: (sum, sum sq) of the values written
    float eps = 1e-5f;
    unsigned* amax_out = nullptr;      // f32 storage + fast_silu only (the split-precision modes): a range slot (kX2SlotWords words, amax_publish) for max |y| of what this launch writes
    // gn_apply_proj only (kernels_norm.hip: GroupNorm + SiLU + residual + a 1x1 projection to <= 8 channels in one pass, NCHW out; y is not written):
    const float* proj_w = nullptr;     // [proj_cout][C] f32, the 1x1 conv's weight as stored in the state dict
    const float* proj_b = nullptr;     // [proj_cout] or null
    int proj_cout = 0;
    void* proj_out = nullptr;          // [nhyp][proj_cout][HW], element type proj_out_dt (NOPE_F32 / NOPE_F16 / NOPE_BF16)
    int proj_out_dt = NOPE_F32;
    int fast_silu = 0;                 // f32 storage only: SiLU on v_exp_f32 + v_rcp_f32 (1 ulp each, what the 16-bit types always use) instead of expf + an
                                       // IEEE division -- set by the runtimes in the split-precision modes (bf16x3, f16x2), whose bar is 1e-4, not bit parity
};
int launch_gn_apply(int dt, const GnApplyArgs& a, hipStream_t s);
bool gn_apply_proj_ok(int dt, const GnApplyArgs& a);          // would launch_gn_apply_proj take these arguments?
int launch_gn_apply_proj(int dt, const GnApplyArgs& a, hipStream_t s);
bool conv_records_out_amax(int dt, const ConvArgs& a);     // would launch_conv's kernel fill a.out_amax? (kernels_gemm.hip)
int launch_absmax_f32(const float* x, size_t n, unsigned* slot, hipStream_t s);     // amax_publish(slot, max |x[i]|) (kernels_misc.hip; NaNs ignored); slot = kX2SlotWords words
int launch_amax_reduce(unsigned* slots, int nslots, unsigned* compact, hipStream_t s);
// the verdict of a forward over its range slots and the NaN fill of an out-of-range forward's output (kernels_misc.hip; unet_runtime.hip)
int launch_x2_verdict(unsigned* slots, int nslots, const int* tab, int n_layers, unsigned* status, unsigned* host_mapped, hipStream_t s);
int launch_x2_poison(void* out, size_t bytes, int out_dt, const unsigned* status, hipStream_t s);      // compact[i] = max over slot i's lines; the lines are zeroed
int gn_apply_blocks(int HW, int C, int dt, int nhyp);      // workgroups per hypothesis of launch_gn_apply over nhyp samples (= chunks of out_stats)
int launch_gn_finalize(const float* partial, float* ms, int nhyp, int nchunk, float count, float eps, hipStream_t s);
int gn_stats_chunks(int HW, int C, int dt);
// fold the conv epilogue's per-row-block column statistics [nhyp][blocks][C][2] into per-(hypothesis, group) partials (nchunk = 1)
int launch_gn_fold(const float* colstats, float* partial, int nhyp, int blocks, int C, int G, hipStream_t s);

int launch_linattn(int dt, const void* qkv, void* out, int nhyp, int HW, int heads, int dim_head, hipStream_t s);
int launch_attn(int dt, const void* qkv, void* out, int nhyp, int HW, int heads, int dim_head, hipStream_t s);

int launch_nchw_to_nhwc(int dt, const float* x, void* y, int n, int C, int HW, hipStream_t s, int C_src = 0);   // C_src < C: zero-padded channels
int launch_nhwc_to_nchw_f32(int dt, const void* x, float* y, int n, int C, int HW, hipStream_t s);
int launch_pack_conv_w(int dt, const float* w, void* out, int Cout, int Cin, int ntaps, int mode, hipStream_t s,
                       const float* cin_scale = nullptr, const float* cout_scale = nullptr, int Cin_src = 0);   // Cin_src < Cin: zero weights for the padding
// NOPE_F16X2 layout of a conv weight (any mode the ping-pong kernels run: PLAIN 1x1 / 3x3, DOWN2, UP2P incl. the ConvTranspose2d source with
// ntaps = 16), Cin % 32 == 0: conv_w_x2_bytes() bytes (the rows, in launch_pack_conv_w's order, + a 16-byte tail with the layer's block scale,
// derived on the device from max |w| of the rows the GEMM sees: no host round trip)
size_t conv_w_x2_bytes(int Cout, int Cin, int ntaps = 9, int mode = NOPE_CONV_PLAIN);
int launch_pack_conv_w_x2(const float* w, void* out, int Cout, int Cin, hipStream_t s, int ntaps = 9, int mode = NOPE_CONV_PLAIN);
// template encoder (kernels_encoder.hip)
int launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale, float* shift,
                   int C, hipStream_t s);
int launch_stem_pack(const float* w, const float* scale, float* out, hipStream_t s);
int launch_stem_conv(int dt, const float* img, const float* w_packed, const float* shift, void* out, int n_img, int H, int W,
                     hipStream_t s);
int launch_rowsum(int dt, const void* packed, float* out, int rows, int K, hipStream_t s);
int launch_linear_naive(const float* in, const float* w, const float* bias, float* out, int M, int N, int K, int act_in,
                        int ldo, hipStream_t s);
int launch_silu_f32(const float* in, float* out, size_t n, hipStream_t s);
int launch_warp_perspective(const void* src, int src_u8, int Hs, int Ws, int C, const float* minv9, float* dst, int Hd, int Wd, float scale,
                            float shift, hipStream_t s);
int launch_pos_emb(const float* pose, float* out, int n, int pose_dim, int classes, hipStream_t s);
int launch_cast(int dt, const float* in, void* out, size_t n, hipStream_t s);

// LDM variant (kernels_ldm.hip)
int launch_layernorm(int dt, const void* x, void* y, const float* gamma, const float* beta, long long M, int C, float eps, hipStream_t s);
int launch_geglu(int dt, const void* in, void* out, long long M, int D, hipStream_t s, int interleaved = 0);   // interleaved: in = (x_j, gate_j) pairs
int launch_add_rowvec(int dt, const void* x, void* y, const float* u, long long M, int tokens, int C, hipStream_t s, int u_stride = 0);   // (u_stride 0 = C)
int launch_token_attention(int dt, const void* qkv, void* out, int nsmp, int N, int C, int dim_head, hipStream_t s);
int launch_copy_cols(int dt, const void* x, void* y, long long M, int C, int C2, int off, hipStream_t s);

int launch_similarity(const float* q, const void* bank, int bank_dt, float* scores, int B, int N, int C, int HW,
                      long long bank_stride_b, int score_ld, hipStream_t s);
int launch_topk(const float* scores, long long* idx, float* vals, int B, int N, int k, int ld, hipStream_t s, const long long* map = nullptr);
int launch_gather_topk(const float* gathered, int G, int B, int N, float* scores, long long* idx, float* vals, int k, hipStream_t s);
int launch_geodesic(const double* poses, long long stride_b, int N, const long long* idx, const double* gt, const int* symmetry,
                    double* err, int* status, int B, int k, hipStream_t s);

}  // namespace nope
