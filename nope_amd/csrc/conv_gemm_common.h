// Shared internals of the implicit-GEMM conv kernels (kernels_gemm.hip: the 128 x 192 kernels and the launcher;
// kernels_gemm_pp.hip: the 256 x 192 eight-wave ping-pong kernel): launch parameters, the MFMA tile traits, the LDS
// swizzle, the fragment stage and the epilogues.  Included inside each translation unit (internal linkage).
#pragma once
#include "nope_common.h"

#ifndef NOPE_EPILOGUE_LEAN_GROUP
#define NOPE_EPILOGUE_LEAN_GROUP 2      // 16-byte chunks a lane keeps in flight between the panel read and the store (1, 2, 4 or 8)
#endif

namespace nope {

namespace {

constexpr int BM = 128;
constexpr int BN = 192;
constexpr int NT = 256;
constexpr int ROWB = 128;  // bytes per LDS row
constexpr int A_ITERS = BM / 32;
constexpr int B_ITERS = BN / 32;

// Unsigned division by a launch-time constant, exact for n < 2^31: q = mulhi(n, M) >> sh with
// M = floor(2^(32+sh) / d) + 1, sh = ceil(log2 d) - 1 (d >= 2); M == 0 encodes d == 1.  Replaces the ~35-instruction
// integer-division sequences of the per-row index arithmetic in the conv prologue.
struct FastDiv {
    unsigned M, sh;
    __device__ __forceinline__ unsigned div(unsigned n) const {
        return M ? (unsigned)(((unsigned long long)n * M) >> 32) >> sh : n;
    }
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f{0u, 0u};
    if (d <= 1) return f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;                       // s = ceil(log2 d) >= 1
    f.sh = s - 1;
    f.M = (unsigned)(((1ull << (31 + s)) / d) + 1);    // < 2^32 because d > 2^(s-1)
    return f;
}

struct ConvParams {
    const unsigned char* src1; const unsigned char* src2;
    int C1, C2, rep1, rep2;
    int Hs, Ws, Ho, Wo;
    int mode, ntaps;
    const unsigned char* w;
    const float* bias;
    const unsigned char* resid;
    unsigned char* out;
    int Cout, M;
    int out_nchw, out_dt;
    int act;                           // 0 none, 1 ReLU (after bias and residual)
    int tiles_m, tiles_n, xcd_map, wide_out;
    int geglu;                         // host side only: selects the GEGLU-epilogue instantiation (ConvArgs::geglu)
    int nchw_staged;                   // out_nchw through the per-wave LDS panels (epilogue_nchw): whole 64-row blocks inside one sample
    int xcd_gn;                        // (xcd_map: 0 none, 1 one panel per XCD, 2 below, 3 small-tile kernel, 4 any tiles_n) xcd_map == 2: XCD columns the weight panels are split over (tile_coords)
    int variant;                       // tuning switches (NOPE_CONV_VARIANT), 0 in production
    FastDiv d_hw, d_w, d_rep1, d_rep2; // / (Hm*Wm), / Wm, / rep1, / rep2
    unsigned char pos_order[64];       // posmajor: pixel positions by descending number of valid taps
    int persist_iters;                 // > 1: a workgroup walks this many tiles
    unsigned persist_d1, persist_d2;   // byte advance of the A offsets per walked tile (src1 / src2)
    int persist_dm;                    // GEMM rows between the tiles a workgroup of the 128 x 192 kernel walks
    unsigned* timeline;                // tuning only (NOPE_PP_VARIANT & 256): cycle stamps of workgroup 0, see conv3x3_halo_kernel
    int posmajor;                      // 1: GEMM rows ordered (pixel position, sample) instead of (sample, pixel) -- see launch_conv
    FastDiv d_n;                       // / nhyp (posmajor)
    int nhyp;
    int splits;                        // > 1: blockIdx.z owns a K range and writes raw f32 partial sums
    float* split_out;                  // [splits][M][Cout]
    int Hm, Wm;                        // grid the GEMM rows enumerate: output grid, or the SOURCE grid for UP2P
    unsigned w_phase_bytes;            // UP2P: byte stride between the 4 phase weight sets
    float* colstats;                   // optional [M/stat_rows][Cout][2]: per row block column sum / sum of squares
    int stat_rows;                     // 64 (every kernel), 16 / 32 (small-tile kernel only)
    const float* pn_ms; const float* pn_c0; const float* pn_c1;   // optional fused PreNorm (see ConvArgs)
    unsigned bytes1, bytes2, bytesw;   // tensor sizes for the buffer descriptors of the DMA kernel
    const int* x2_scale;               // NOPE_F16X2 (ping-pong kernels): the tail of the packed weights, [0] = E8M0 scale of the A operand, [3] = range shift t
    unsigned* x2_amax;                 // NOPE_F16X2: optional device word, atomicMax of the bits of max |a| over every A element the launch converted (NOPE_X2_KERNEL_AMAX builds)
    int x2_t_zero;                     // NOPE_F16X2: the caller vouches that the layer's range shift (tail word 3) is 0: the tap-resident kernel skips the a * 2^-t multiplies
    int lean;                          // 1: f32 storage, every wave tile of the launch whole and in NHWC row order (see epilogue_wide, LEANM): the kernels' LEAN instantiations
    unsigned* out_amax;                // f32-storage launches with a wide NHWC epilogue: optional range slot (amax_publish) for max |out| of what the launch writes
};

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// MFMA tile of a wave's 64 x 96 output block, per element type.
//   f32 : v_mfma_f32_16x16x4_f32  -- 4 x 6 tiles, a lane's 16-byte fragment = 4 channels = 4 chained steps
//   bf16: v_mfma_f32_32x32x16_bf16 -- 2 x 3 tiles (the 32x32 form sustains ~15 % more than 16x16x32 on gfx950:
//         2382 vs 2075 TFLOP/s, cdna_hip_programming.md section 3), a lane's fragment = 8 channels = 1 step
//   f16 : v_mfma_f32_32x32x16_f16, as bf16
//   f32s (NOPE_BF16X3): f32 data, three v_mfma_f32_32x32x16_bf16 per 16 channels over (hi, lo) bf16 splits, below
// A K STEP is what one round of MFMAs over the wave tile consumes: STEP_SLOTS 16-byte slots of a staged row, of which a lane
// reads RAW (slot = step * STEP_SLOTS + frag_slot(lane) + r).  frag_row / frag_slot: which tile row and first slot a lane
// feeds; out_row / out_col: the C/D map.  prep_step turns the raw A reads into MFMA operands (a no-op except for f32s).
template <class T> struct Tile;
template <> struct Tile<float> {
    static constexpr int TM = 16, MT = 4, NTL = 6, R = 4, STEP_SLOTS = 4, RAW = 1, RAW_STRIDE = 1;
    typedef f32x4 acc_t;
    static __device__ __forceinline__ int frag_row(int lane) { return lane & 15; }
    static __device__ __forceinline__ int frag_slot(int lane) { return lane >> 4; }
    static __device__ __forceinline__ int out_row(int lane, int r) { return (lane >> 4) * 4 + r; }
    static __device__ __forceinline__ int out_col(int lane) { return lane & 15; }
    static __device__ __forceinline__ void prep_step(u32x4 (&)[RAW][MT]) {}
    static constexpr int TERMS = 1;
    static __device__ __forceinline__ void mma(int, const u32x4 (&a)[RAW][MT], const u32x4 (&b)[RAW][NTL], int i, int j, acc_t& c, int = 0) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a[0][i]), fb = __builtin_bit_cast(f32x4, b[0][j]);
#pragma unroll
        for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[q], fb[q], c, 0, 0, 0);
    }
};
// shared geometry of the 32x32x16 tiles
struct Tile32 {
    static constexpr int TM = 32, MT = 2, NTL = 3, R = 16;
    typedef f32x16 acc_t;
    static __device__ __forceinline__ int frag_row(int lane) { return lane & 31; }
    static __device__ __forceinline__ int out_row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
    static __device__ __forceinline__ int out_col(int lane) { return lane & 31; }
};
template <> struct Tile<bf16_t> : Tile32 {
    static constexpr int STEP_SLOTS = 2, RAW = 1, RAW_STRIDE = 1;
    static __device__ __forceinline__ int frag_slot(int lane) { return lane >> 5; }
    static __device__ __forceinline__ void prep_step(u32x4 (&)[RAW][MT]) {}
    static constexpr int TERMS = 1;
    static __device__ __forceinline__ void mma(int, const u32x4 (&a)[RAW][MT], const u32x4 (&b)[RAW][NTL], int i, int j, acc_t& c, int = 0) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0][i]), __builtin_bit_cast(bf16x8, b[0][j]), c, 0, 0, 0);
    }
};
template <> struct Tile<f16_t> : Tile32 {
    static constexpr int STEP_SLOTS = 2, RAW = 1, RAW_STRIDE = 1;
    static __device__ __forceinline__ int frag_slot(int lane) { return lane >> 5; }
    static __device__ __forceinline__ void prep_step(u32x4 (&)[RAW][MT]) {}
    static constexpr int TERMS = 1;
    static __device__ __forceinline__ void mma(int, const u32x4 (&a)[RAW][MT], const u32x4 (&b)[RAW][NTL], int i, int j, acc_t& c, int = 0) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0][i]), __builtin_bit_cast(f16x8, b[0][j]), c, 0, 0, 0);
    }
};
// NOPE_BF16X3.  A rows are f32 in LDS (straight from the f32 activations, LDS-DMA included), weight rows were split at pack
// time: 8 channels = 32 bytes = [hi x 8 | lo x 8] bf16 with hi = rn(w), lo = rn(w - hi) (kernels_misc.hip) -- so both operands
// have 32 channels per 128-byte row and every loader, swizzle and DMA path is the f32 one.  A K step is 16 channels: lanes
// 0..31 own channels 0..7 of it (slots 0, 1), lanes 32..63 channels 8..15 (slots 2, 3), two raw reads per row each; prep_step
// splits a lane's 8 f32 A values into (hi, lo) in place (~3 VALU per element, issued in the LOAD phase of the ping-pong
// kernels, under the other group's MFMAs) and the step issues  acc += a_lo w_hi;  acc += a_hi w_lo;  acc += a_hi w_hi  --
// the three products whose error is O(2^-17) per operand; a_lo w_lo (2^-18 relative) is dropped.
template <> struct Tile<f32s_t> : Tile32 {
    static constexpr int STEP_SLOTS = 4, RAW = 2, RAW_STRIDE = 1;
    static __device__ __forceinline__ int frag_slot(int lane) { return (lane >> 5) * 2; }
    static __device__ __forceinline__ void prep_step(u32x4 (&a)[RAW][MT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i) prep_one(a, i);
    }
    // ... of ONE row tile (the per-tap ping-pong kernel splits them inside its COMPUTE phase, one row tile ahead of the MFMAs that use it)
    static __device__ __forceinline__ void prep_one(u32x4 (&a)[RAW][MT], int i) {
        {
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {      // (through a scalar: __builtin_bit_cast straight from a vector-element lvalue reads element 0)
                const unsigned u0 = a[0][i][e], u1 = a[1][i][e];
                x[e] = __builtin_bit_cast(float, u0); x[4 + e] = __builtin_bit_cast(float, u1);
            }
            u32x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned h = cvt_pk_bf16(x[2 * e], x[2 * e + 1]);
                const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
                hi[e] = h;
                lo[e] = cvt_pk_bf16(x[2 * e] - h0, x[2 * e + 1] - h1);
            }
            a[0][i] = hi; a[1][i] = lo;
        }
    }
    static constexpr int TERMS = 3;           // (lo, hi), (hi, lo), (hi, hi): term outer in the callers' loops, so MFMAs on one accumulator sit 6 apart
    static __device__ __forceinline__ void mma(int t, const u32x4 (&a)[RAW][MT], const u32x4 (&b)[RAW][NTL], int i, int j, acc_t& c, int = 0) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t == 0 ? 1 : 0][i]), __builtin_bit_cast(bf16x8, b[t == 1 ? 1 : 0][j]), c, 0, 0, 0);
    }
};
// NOPE_F16X2 (the ping-pong kernels).  Both operands hold 32 channels per 128-byte row, as NOPE_BF16X3, in EIGHT 16-byte slots:
//     slot 2 ks + h   (ks, h in {0, 1})   f16 hi parts of channels 16 ks + 8 h .. + 7           -- the operands of two 32x32x16 f16 MFMAs
//     slot 4 + h      e4m3 bytes over "channel set" h = channels 8 h .. 8 h + 7, 16 + 8 h .. 16 + 8 h + 7:   A: a_lo * 2^9,  B: w * 2^sw
//     slot 6 + h      the same 16 channels:                                                               A: a * 2^-2,    B: w_lo * 2^(sw + 11)
// (hi = f16(x), lo = x - hi; the weights arrive like this from launch_pack_conv_w_x2.  The activations: the tap-resident kernel rewrites
// its staged rows in LDS -- convert_piece in conv3x3_halo_kernel, the rewrite amortised over nine taps -- and reads A exactly as B; the
// per-tap kernel stages raw f32 rows and splits them in REGISTERS: a lane of half h reads the 16 f32 channels of its channel set -- raw
// slots 2 h, 2 h + 1, 4 + 2 h, 5 + 2 h: frag_slot_raw / raw_slot_a -- and prep_hi / prep_lo turn them into the same four operands.)
// A lane of half h = lane >> 5 reads slot h + 2 r for r = 0..3 (RAW_STRIDE 2: the XOR of raw_slot into a fragment address stays disjoint
// from frag_slot's bit): reads 0, 1 feed the two f16 MFMAs, reads 2 + 3 are the 32 bytes of its half of ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 whose K axis is the concatenation [a_lo | a] x [w ; w_lo] over the channel sets of the step's 32
// channels -- both cross terms at once, at twice the f16 rate, into the same accumulator: the instruction's E8M0 block scale
// (one value for every lane, `sc` = 127 - 9 - sw; B's is 1.0) undoes the pre-scales.  What the instruction really does with operands and
// scales is pinned by tools/probes/mx_probe.hip (byte e of lane half h pairs with byte e of lane half h; its 64-term sum is truncated at
// ~2^-12 of the largest term: irrelevant for terms that are 2^-11 of the result).  Three "terms" per step = 2 pass equivalents of an f16
// MFMA step against NOPE_BF16X3's 3.
template <> struct Tile<f16x2_t> : Tile32 {
    static constexpr int STEP_SLOTS = 8, RAW = 4, RAW_STRIDE = 2;
    static __device__ __forceinline__ int frag_slot(int lane) { return lane >> 5; }
    static __device__ __forceinline__ void prep_step(u32x4 (&)[RAW][MT]) {}
    // ---- A rows staged as raw f32 (per-tap kernel): which slots a lane reads, and the register split.  r = the lane's four raw reads of row
    // tile i (channels 8 h .. + 3 | + 4 .. + 7 | 16 + 8 h .. + 3 | + 4 .. + 7), x = the four MFMA operands.  Same arithmetic as the LDS rewrite
    // (saturating conversions: the wave runs with MODE.FP16_OVFL = 1).
    static __device__ __forceinline__ int frag_slot_raw(int lane) { return (lane >> 5) * 2; }
    static __device__ __forceinline__ constexpr int raw_slot_a(int q) { return (q & 1) | ((q >> 1) << 2); }
    // `inv` = 2^-t, the layer's range shift (nope_common.h: kX2*): every operand is formed from a' = a * 2^-t
    static __device__ __forceinline__ void prep_hi(const u32x4 (&r)[RAW][MT], u32x4 (&x)[RAW][MT], int i, int ks, float inv) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned u0 = r[2 * ks + q][i][2 * e], u1 = r[2 * ks + q][i][2 * e + 1];
                const f32x2_t v = f32x2_t{__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1)} * inv;
                x[ks][i][2 * q + e] = NOPE_CVT_PK_F16_OVFL(v.x, v.y);
            }
    }
    static __device__ __forceinline__ void prep_lo(const u32x4 (&r)[RAW][MT], u32x4 (&x)[RAW][MT], int i, int q, float inv, float div_a, float& amax) {      // raw read q: 4 channels; div_a = 2^(2 + t)
        float v[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned u = r[q][i][e]; v[e] = __builtin_bit_cast(float, u); }
        if (NOPE_X2_KERNEL_AMAX) amax = amax4(amax, v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            union { unsigned u; f16_t f[2]; } hh; hh.u = x[q >> 1][i][2 * (q & 1) + e];
            l[2 * e] = __builtin_fmaf(v[2 * e], inv, -(float)hh.f[0]);              // a * 2^-t - hi: the product is exact, one rounding-free subtraction
            l[2 * e + 1] = __builtin_fmaf(v[2 * e + 1], inv, -(float)hh.f[1]);
        }
        x[2][i][q] = cvt4_e4m3_scaled<kX2ALoShift, true>(l[0], l[1], l[2], l[3]);
        x[3][i][q] = cvt4_e4m3_div(v[0], v[1], v[2], v[3], div_a);
    }
    static constexpr int TERMS = 3;
    static __device__ __forceinline__ void mma(int t, const u32x4 (&a)[RAW][MT], const u32x4 (&b)[RAW][NTL], int i, int j, acc_t& c, int sc) {
        if (t < 2) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t][i]), __builtin_bit_cast(f16x8, b[t][j]), c, 0, 0, 0);
        } else {
            const u32x4 a0 = a[2][i], a1 = a[3][i], b0 = b[2][j], b1 = b[3][j];
            const i32x8 va = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
            const i32x8 vb = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, 0, sc, 0, 127);      // (cbsz = blgp = 0: e4m3 x e4m3)
        }
    }
};
// NOPE_F16X2: a wave's max |a| over the A elements it converted -> the launch's range word (one atomic per wave and tile; the word is
// read by the runtime that owns the layer, unet_runtime.hip: x2_range_check).  Non-negative floats order like their bit patterns.
__device__ __forceinline__ void x2_publish_amax(const ConvParams& p, float m, int lane) {
    if (!p.x2_amax) return;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) atomicMax(p.x2_amax, __builtin_bit_cast(unsigned, m));
}

// slot offset of raw read q (= step * RAW + r) of a staged row, to be XORed into a fragment address (<< 4)
template <class T> __device__ __forceinline__ constexpr int raw_slot(int q) { return (q / Tile<T>::RAW) * Tile<T>::STEP_SLOTS + (q % Tile<T>::RAW) * Tile<T>::RAW_STRIDE; }

__device__ __forceinline__ void tile_coords(const ConvParams& p, int& tile_m, int& tile_n) {
    const int g = blockIdx.x;
    if (p.xcd_map == 2) {   // ping-pong kernels with several weight panels
        // The 8 XCDs form a (8 / xcd_gn) x xcd_gn grid: XCD (xm, xn) owns a contiguous run of tiles_m / (8 / xcd_gn) M tiles and
        // tiles_n / xcd_gn weight panels, and walks the run with its panels fastest.  Every activation row then crosses the
        // fabric xcd_gn times and every weight panel 8 / xcd_gn times (each XCD has its own L2); the launcher picks xcd_gn to
        // minimise that (xcd_gn = tiles_n is the one-panel-per-XCD map below).
        const int x = g & 7, j = g >> 3;
        const int span = p.tiles_n / p.xcd_gn;              // panels per XCD
        const int xm = x / p.xcd_gn, xn = x - xm * p.xcd_gn;
        tile_n = xn * span + (j & (span - 1));
        tile_m = xm * (p.tiles_m / (8 / p.xcd_gn)) + j / span;
    } else if (p.xcd_map == 4) {   // any tiles_n (the LDM variant's 256 / 512 / 1024-channel linears: 3, 6, 11, 22 ... panels), tiles_m % 8 == 0
        // XCD x owns a contiguous run of tiles_m / 8 M tiles and walks every weight panel of each before the next: an activation
        // tile crosses the fabric once instead of once per XCD its tiles_n consumers would otherwise land on
        const int x = g & 7, j = g >> 3;
        const int q = j / p.tiles_n;
        tile_n = j - q * p.tiles_n;
        tile_m = x * (p.tiles_m >> 3) + q;
    } else if (p.xcd_map) {   // tiles_n in {1,2,4,8}, tiles_m % (8 / tiles_n) == 0
        // XCD x = g & 7 keeps one weight panel (tile_n) and a CONTIGUOUS run of M tiles, so the
        // 3x3 halo rows shared by neighbouring tiles hit the same XCD's L2.
        const int x = g & 7, j = g >> 3;
        const int per = 8 / p.tiles_n;
        tile_n = x % p.tiles_n;
        tile_m = (x / p.tiles_n) * (p.tiles_m / per) + j;
    } else {
        tile_n = g % p.tiles_n;
        tile_m = g / p.tiles_n;
    }
}

// Row geometry of a staged tile: RB bytes of K per row, 16-byte slots XOR-swizzled so that the rows x one slot
// of a ds_read_b128 lane group hit different bank positions (for both the 16- and the 32-row fragment shapes).
template <int RB> __device__ __forceinline__ int swz_of(int row) { return RB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
template <int RB> __device__ __forceinline__ int lds_off_rb(int row, int slot) { return row * RB + ((slot ^ swz_of<RB>(row)) << 4); }

// One K stage (RB bytes of K per row) of the 64 x 96 wave tile from the swizzled LDS tiles.  The fragments of
// K step ks+1 are read while the MFMAs of step ks execute (two statically named register sets).
template <class T, int RB>
__device__ __forceinline__ void mma_stage(const unsigned char* ldsA, const unsigned char* ldsB, int wm, int wn, int lane,
                                          typename Tile<T>::acc_t (&acc)[Tile<T>::MT][Tile<T>::NTL]) {
    typedef Tile<T> TL;
    constexpr int KS = RB / 16 / TL::STEP_SLOTS;
    u32x4 af[2][TL::RAW][TL::MT], bfr[2][TL::RAW][TL::NTL];
    auto load = [&](int set, int ks) {
#pragma unroll
        for (int r = 0; r < TL::RAW; ++r) {
            const int s = ks * TL::STEP_SLOTS + TL::frag_slot(lane) + r;
#pragma unroll
            for (int i = 0; i < TL::MT; ++i) af[set][r][i] = ld16(ldsA + lds_off_rb<RB>(wm * 64 + i * TL::TM + TL::frag_row(lane), s));
#pragma unroll
            for (int j = 0; j < TL::NTL; ++j) bfr[set][r][j] = ld16(ldsB + lds_off_rb<RB>(wn * 96 + j * TL::TM + TL::frag_row(lane), s));
        }
        TL::prep_step(af[set]);
    };
    load(0, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) load((ks + 1) & 1, ks + 1);
#pragma unroll
        for (int t = 0; t < TL::TERMS; ++t)
#pragma unroll
            for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j) TL::mma(t, af[ks & 1], bfr[ks & 1], i, j, acc[i][j]);
    }
}

// Output pixel row of GEMM row m.  Identity except for UP2P, whose rows walk the source grid and land on
// output pixel (2y + py, 2x + px) of phase blockIdx.y.
__device__ __forceinline__ size_t out_row(const ConvParams& p, int m) {
    if (p.posmajor) {                  // row m = (position, sample) -> NHWC row (sample, position)
        const unsigned pos = p.d_n.div((unsigned)m);
        const unsigned b = (unsigned)m - pos * (unsigned)p.nhyp;
        return (size_t)b * (size_t)(p.Hm * p.Wm) + pos;
    }
    if (p.mode != NOPE_CONV_UP2P) return (size_t)m;
    const int hw = p.Hm * p.Wm;
    const int b = m / hw;
    const int r = m - b * hw;
    const int y = r / p.Wm, x = r - y * p.Wm;
    const int py = (int)blockIdx.y >> 1, px = (int)blockIdx.y & 1;
    return ((size_t)b * p.Ho + 2 * y + py) * p.Wo + 2 * x + px;
}

// Fused PreNorm on an accumulator: rstd * (acc - mean * c1) + c0 with the two roundings pinned in the source (two fmas) -- the generic loop, the
// lean row walk and the small-tile kernel's epilogue must agree bit for bit whatever the compiler would contract or hoist in each of them.
__device__ __forceinline__ float pn_apply(float acc, float mean, float rstd, float c1, float c0) {
    return __builtin_fmaf(rstd, __builtin_fmaf(-mean, c1, acc), c0);
}

// C/D map of the 16x16 MFMA tiles: col = lane & 15, row = (lane >> 4) * 4 + r
template <class T, bool PN>
__device__ __forceinline__ void epilogue(const ConvParams& p, const typename Tile<T>::acc_t (&acc)[Tile<T>::MT][Tile<T>::NTL], int m0,
                                         int n0, int wm, int wn, int lane) {
    // Every index into acc must stay a compile-time constant (full unroll): a runtime index would move the
    // accumulators to scratch for the WHOLE kernel.  Row bookkeeping (incl. the divisions) is hoisted to
    // once per (i, r).
    typedef Tile<T> TL;
    const int HWo = p.Ho * p.Wo;
    T* out = reinterpret_cast<T*>(p.out);
    const T* resid = reinterpret_cast<const T*>(p.resid);
    float bv[TL::NTL];
    int ncol[TL::NTL];
#pragma unroll
    for (int j = 0; j < TL::NTL; ++j) {
        ncol[j] = n0 + wn * 96 + j * TL::TM + TL::out_col(lane);
        bv[j] = (p.bias && ncol[j] < p.Cout) ? p.bias[ncol[j]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TL::MT; ++i) {
#pragma unroll
        for (int r = 0; r < TL::R; ++r) {
            const int m = m0 + wm * 64 + i * TL::TM + TL::out_row(lane, r);
            const bool row_ok = m < p.M;
            const size_t mo = out_row(p, row_ok ? m : 0);
            float pn_mean = 0.f, pn_rstd = 1.f;
            if (PN) { const int b = (row_ok ? m : 0) / (p.Hm * p.Wm); pn_mean = p.pn_ms[2 * b]; pn_rstd = p.pn_ms[2 * b + 1]; }
            size_t nchw_base = 0;
            if (p.out_nchw) {
                const int mm = row_ok ? m : 0;
                const int b = mm / HWo;
                nchw_base = (size_t)b * p.Cout * HWo + (mm - b * HWo);
            }
#pragma unroll
            for (int j = 0; j < TL::NTL; ++j) {
                const int n = ncol[j];
                if (!row_ok || n >= p.Cout) continue;
                float v = acc[i][j][r] + bv[j];
                if (PN) v = pn_apply(acc[i][j][r], pn_mean, pn_rstd, p.pn_c1[n], p.pn_c0[n] + bv[j]);
                if (resid) v += Elt<T>::ld(resid + mo * p.Cout + n);
                if (p.act) v = v > 0.f ? v : 0.f;
                if (p.out_nchw) {
                    const size_t o = nchw_base + (size_t)n * HWo;
                    if (p.out_dt == NOPE_F32) reinterpret_cast<float*>(p.out)[o] = v;
                    else if (p.out_dt == NOPE_F16) reinterpret_cast<f16_t*>(p.out)[o] = f32_to_f16_sat(v);
                    else reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(v);
                } else {
                    Elt<T>::st(out + mo * p.Cout + n, v);
                }
            }
        }
    }
}

// Wide-store epilogue (NHWC output, Cout % VEC == 0).  The MFMA C/D layout gives a lane one column x a few
// rows per tile, i.e. 2-byte scattered stores; instead each wave stages its 64 x 96 f32 accumulators (+bias)
// through its private 64 x 52-word LDS panel, PANW columns at a time (48 = three 16-wide tiles for f32, 32 =
// one 32-wide tile for bf16), and writes rows back as 16-byte vectors (residual added in f32 before the
// single rounding to T).  On the way it emits the per-column sums the following GroupNorm needs.
// (panel row stride: 32 + 4 words for the bf16 tiles -- four 9 KiB panels then fit into ONE 40 KiB DMA stage, which
// lets a persistent workgroup prefetch its next tile into the other stage during the epilogue -- 48 + 4 for f32)
template <class T> struct Ep {
    static constexpr int LD = Tile<T>::TM == 32 ? 36 : 52;
    static constexpr int WAVE_BYTES = 64 * LD * 4;
};

// DRAIN: wait for the vector-memory queue (a prefetch issued before the call) after the first panel is filled, before the
// first store.
struct NoStamp { __device__ __forceinline__ void operator()() const {} };
// GG: GEGLU on column pairs (ConvArgs::geglu; packed 16-bit path only -- the launcher guarantees it is the one taken).
// LEANM = 1 (f32 storage on the 32 x 32 tiles only; ConvParams::lean: the launcher vouches for EVERY wave tile of the launch): the lean row
// walk below instead of the generic loop -- a kernel instantiation of its own, because both forms in one kernel spill.
template <class T, bool PN, bool DRAIN = false, class Stamp = NoStamp, bool GG = false, int LEANM = 0>
__device__ __forceinline__ void epilogue_wide(const ConvParams& p, const typename Tile<T>::acc_t (&acc)[Tile<T>::MT][Tile<T>::NTL],
                                              int m0, int n0, int wm, int wn, int lane, unsigned char* lds_wave, Stamp stamp = Stamp(),
                                              float acc_scale = 1.0f) {
    typedef Tile<T> TL;
    constexpr int VEC = Elt<T>::VEC;
    // NOPE_F16X2: the accumulators hold 2^-t x the convolution (t = the layer's activation range shift, nope_common.h: kX2*); acc_scale = 2^t
    // enters where the bias does (an fma in place of the add).  Every other element type: 1, folded away at compile time.
    constexpr bool SCALED = Elt<T>::DT == NOPE_F16X2;
    const float asc = SCALED ? acc_scale : 1.0f;
    const bool track_out = sizeof(T) == 4 && p.out_amax != nullptr;      // (wave-uniform)
    float out_max = 0.f;
    constexpr int PANW = TL::TM == 32 ? 32 : 48;   // panel width in columns
    constexpr int TPP = PANW / TL::TM;             // MFMA tiles per panel pass
    constexpr int CH = PANW / VEC;                 // 16-byte output chunks per panel row
    float* pan = reinterpret_cast<float*>(lds_wave);
    T* out = reinterpret_cast<T*>(p.out);
    const T* resid = reinterpret_cast<const T*>(p.resid);
    // the bias of this lane's NTL columns: all loads in flight at once, ahead of everything that waits for them
    float bvj[TL::NTL];
#pragma unroll
    for (int j = 0; j < TL::NTL; ++j) {
        const int n = n0 + wn * 96 + j * TL::TM + TL::out_col(lane);
        bvj[j] = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;
    }
    bool inrange[TL::NTL];              // f16 storage: this lane's values of column tile j are known to lie inside the format's range
#pragma unroll
    for (int j = 0; j < TL::NTL; ++j) inrange[j] = false;
    if (p.colstats && p.stat_rows == 16) {
        // ... per 16-row block: maps of 16 pixels (the 4 x 4 level: a wave's 64 rows are four samples).  A lane's accumulator registers
        // r < 8 / r >= 8 of a 32 x 32 tile are rows 0-15 / 16-31 of it (out_row), a 16 x 16 tile is one block: the sums split in the lane,
        // only the fold over the lanes that share a column stays.  Same order and single-writer rule as the 64-row form below.
        constexpr int HALVES = TL::TM == 32 ? 2 : 1, RH = TL::R / HALVES;
#pragma unroll
        for (int j = 0; j < TL::NTL; ++j) {
            const int n = n0 + wn * 96 + j * TL::TM + TL::out_col(lane);
            const float bv = bvj[j];
            float qtot = 0.f;
#pragma unroll
            for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                for (int h = 0; h < HALVES; ++h) {
                    f32x2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
                    for (int r = h * RH; r < (h + 1) * RH; r += 2) {
                        const f32x2_t v = SCALED ? f32x2_t{acc[i][j][r], acc[i][j][r + 1]} * asc + bv : f32x2_t{acc[i][j][r], acc[i][j][r + 1]} + bv;
                        s2 += v; q2 += v * v;
                    }
                    float s = s2.x + s2.y, q = q2.x + q2.y;
                    qtot += q;
#pragma unroll
                    for (int o = TL::TM; o < 64; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
                    const int mrow = m0 + wm * 64 + i * TL::TM + h * 16;
                    if (lane < TL::TM && n < p.Cout && mrow < p.M) {
                        float* cs = p.colstats + ((size_t)(mrow >> 4) * p.Cout + n) * 2;
                        cs[0] = s; cs[1] = q;
                    }
                }
            if (Elt<T>::SATURATES) inrange[j] = qtot <= kF16Max * kF16Max;
        }
    } else if (p.colstats) {
        // GroupNorm statistics of the conv output, fused: per column (sum, sum of squares) over this wave's 64
        // rows, straight from the accumulators (f32, before the rounding to T): a lane adds up the rows it
        // holds, the 64/TM lanes sharing a column are folded with cross-lane adds.  Fixed order and exactly one
        // writer per [row block][column] entry -> the later fold is deterministic.  Requires M % 64 == 0
        // (checked by the launcher).
#pragma unroll
        for (int j = 0; j < TL::NTL; ++j) {
            const int n = n0 + wn * 96 + j * TL::TM + TL::out_col(lane);
            const float bv = bvj[j];
            // (on float pairs -- v_pk_add / v_pk_add / v_pk_fma per two values: the epilogue is VALU-issue-bound with two waves per SIMD)
            f32x2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                for (int r = 0; r < TL::R; r += 2) {
                    const f32x2_t v = SCALED ? f32x2_t{acc[i][j][r], acc[i][j][r + 1]} * asc + bv : f32x2_t{acc[i][j][r], acc[i][j][r + 1]} + bv;
                    s2 += v; q2 += v * v;
                }
            float s = s2.x + s2.y, q = q2.x + q2.y;
            // q bounds the lane's 32 values of this column: q <= 65504^2 means none of them needs the f16 clamp (free overflow test)
            if (Elt<T>::SATURATES) inrange[j] = q <= kF16Max * kF16Max;
#pragma unroll
            for (int o = TL::TM; o < 64; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            // (M % 128 == 64: the second wave row of the last tile lies beyond M -- it owns no row block)
            if (lane < TL::TM && n < p.Cout && m0 + wm * 64 < p.M) {
                float* cs = p.colstats + ((size_t)((m0 + wm * 64) >> 6) * p.Cout + n) * 2;
                cs[0] = s; cs[1] = q;
            }
        }
    }
    if constexpr (sizeof(T) == 2 && TL::TM == 32) {
        // (fused PreNorm: with HW % 64 == 0 the wave's 64 rows are one sample, so rstd * (acc - mean c1[n]) + c0[n] + bias[n] is
        //  acc * rstd + off[n] with one constant per lane and pass -- the same fill loop with an fma in place of the add)
        const bool pn_uniform = PN && ((p.Hm * p.Wm) % 64 == 0);
        if (!resid && (!PN || pn_uniform)) {
            float pn_scale = 1.f;
            if (PN) {
                const int b = (m0 + wm * 64 < p.M ? m0 + wm * 64 : 0) / (p.Hm * p.Wm);
                const float mean = p.pn_ms[2 * b];
                pn_scale = p.pn_ms[2 * b + 1];
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j) {
                    const int n = n0 + wn * 96 + j * TL::TM + TL::out_col(lane);
                    if (n < p.Cout) bvj[j] += p.pn_c0[n] - pn_scale * mean * p.pn_c1[n];
                }
            }
            // 16-bit without a residual: the values are final before they leave the registers, so the panel holds them ROUNDED,
            // two rows per dword (a lane's accumulator registers r, r + 1 are rows 2q, 2q + 1 of one column): half the LDS
            // stores of the f32 panel -- the LDS store path (64 B/clk) is what bounds the fill -- and half the reads; the
            // store side splits 8 dwords (8 columns x 2 rows) into the two rows' 16-byte chunks.
            constexpr int PLD = 36;                                   // dwords per row pair (32 columns + pad)
            unsigned* pk = reinterpret_cast<unsigned*>(lds_wave);
            // The four rows a lane stores (two row pairs) are the same in every pass: their offsets -- out_row() is a chain of
            // uniform branches around divisions for the position-major / 2x2-phase launches -- are taken once, not per store.
            const int ch = lane & 3;
            size_t ro[2][2];
            bool ok[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m = m0 + wm * 64 + 2 * ((lane >> 2) + 16 * t) + h;
                    ok[t][h] = m < p.M;
                    ro[t][h] = ok[t][h] ? out_row(p, m) * p.Cout : 0;
                }
            const bool relu = p.act != 0;                             // (encoder convs; a real branch around the whole fill)
#pragma unroll
            for (int pass = 0; pass < TL::NTL; ++pass) {
                const float bv = bvj[pass];
                auto fill = [&](auto relu_tag, auto sat_tag) {
#pragma unroll
                    for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                        for (int r = 0; r < TL::R; r += 2) {
                            f32x2_t v = f32x2_t{acc[i][pass][r], acc[i][pass][r + 1]};
                            if (PN) v = v * pn_scale + bv; else v = v + bv;
                            if (decltype(relu_tag)::value) v = f32x2_t{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)};
                            pk[((i * TL::TM + TL::out_row(lane, r)) >> 1) * PLD + TL::out_col(lane)] =
                                decltype(sat_tag)::value ? Elt<T>::cvt_pk(v.x, v.y) : Elt<T>::cvt_pk_raw(v.x, v.y);
                        }
                };
                // f16: the clamp (one v_med3_f32 per value, half of this loop's VALU) is skipped when the column statistics
                // above have shown every value of the wave's tile column in range (wave-uniform vote)
                const bool sat = Elt<T>::SATURATES && !__all(inrange[pass]);
                if (relu) fill(BoolTag<true>(), BoolTag<true>());
                else if (sat) fill(BoolTag<false>(), BoolTag<true>());
                else fill(BoolTag<false>(), BoolTag<false>());
                __builtin_amdgcn_wave_barrier();                      // (same-wave LDS write -> read, see below)
                stamp();
                if (DRAIN && pass == 0) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
                stamp();
                const int n = n0 + wn * 96 + pass * TL::TM + ch * 8;
                u32x4 d[2][2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int rp = (lane >> 2) + 16 * t;
                    d[t][0] = ld16(&pk[rp * PLD + ch * 8]);
                    d[t][1] = ld16(&pk[rp * PLD + ch * 8 + 4]);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    u32x4 lo, hi;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned e = d[t][k >> 1][(2 * k) & 3], o = d[t][k >> 1][(2 * k + 1) & 3];
                        lo[k] = (e & 0xffffu) | (o << 16);
                        hi[k] = (e >> 16) | (o & 0xffff0000u);
                    }
                    if constexpr (GG) {
                        // the lane's 8 columns are four (x, gate) pairs, already rounded to T exactly as a stored projection would be: x * gelu(gate),
                        // four values = 8 bytes per row into the [M][Cout / 2] output
                        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                        float vl[8], vh[8];
                        Elt<T>::unpack(lo, vl);
                        Elt<T>::unpack(hi, vh);
                        const u32x2_t yl = {Elt<T>::cvt_pk(geglu_f(vl[0], vl[1]), geglu_f(vl[2], vl[3])), Elt<T>::cvt_pk(geglu_f(vl[4], vl[5]), geglu_f(vl[6], vl[7]))};
                        const u32x2_t yh = {Elt<T>::cvt_pk(geglu_f(vh[0], vh[1]), geglu_f(vh[2], vh[3])), Elt<T>::cvt_pk(geglu_f(vh[4], vh[5]), geglu_f(vh[6], vh[7]))};
                        if (n < p.Cout) {
                            if (ok[t][0]) *reinterpret_cast<u32x2_t*>(out + (ro[t][0] >> 1) + (n >> 1)) = yl;
                            if (ok[t][1]) *reinterpret_cast<u32x2_t*>(out + (ro[t][1] >> 1) + (n >> 1)) = yh;
                        }
                    } else if (n < p.Cout) {
                        if (ok[t][0]) st16(out + ro[t][0] + n, lo);
                        if (ok[t][1]) st16(out + ro[t][1] + n, hi);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                stamp();
            }
            return;
        }
    }
    static_assert(LEANM == 0 || (sizeof(T) == 4 && TL::TM == 32 && !GG), "the lean epilogue: f32 storage, 32 x 32 tiles");
    if constexpr (LEANM == 1) {
        // f32 storage on the 32 x 32 tiles (NOPE_BF16X3 / NOPE_F16X2: the modes the benchmark runs in), whole wave tile inside the tensor, rows in
        // NHWC order, one sample per wave tile under a fused PreNorm: the same arithmetic as the loop below with everything that does not
        // depend on the row hoisted and the chunk loop unrolled.  With CH = 8 chunks per panel row a lane keeps ONE column chunk (lane & 7)
        // and walks rows (lane >> 3) + 8 c: one base pointer + a constant stride, no division, no bounds test, no per-element constant
        // load -- the loop below (a real loop around out_row()'s uniform branches, the PreNorm division and per-element bias loads) cost a
        // level-0 1x1 launch more than its MFMAs and its memory traffic together (profiles/r06i_dma_ablations_bf16x3.txt).
        const int mw = m0 + wm * 64, nw = n0 + wn * 96;
        const unsigned long long obytes = (unsigned long long)p.M * (unsigned)p.Cout * 4ull;      // (< 4 GiB: launch_conv)
        {
            const int ch = lane & 7, r0 = lane >> 3;
            float pmean = 0.f, prstd = 1.f;
            if (PN) { const int b = mw / (p.Hm * p.Wm); pmean = p.pn_ms[2 * b]; prstd = p.pn_ms[2 * b + 1]; }
            // Buffer addressing: 32-bit byte offsets (one per-lane base + a uniform pass / row-group term: one v_add per access) -- no 64-bit
            // address arithmetic in a kernel that has no register to spare.  The uniform term goes into the VECTOR offset, not the
            // instruction's scalar offset, on purpose: with an SGPR offset hipcc (ROCm 7.2) assumes a dwordx4 store has read its data before
            // the next instruction and lets a VALU overwrite the data registers right behind it -- measured on gfx950: the first dword of
            // lanes 12-15 (mod 16) of a pass's last store took the NEXT pass's panel-fill temporaries (tests/test_conv_pingpong.py caught it).
            // Without a scalar offset register the compiler inserts the wait state itself.
            const auto ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, (short)0, (int)(unsigned)obytes, 0x00020000);
            const auto rr = __builtin_amdgcn_make_buffer_rsrc((void*)(resid ? (const void*)resid : (const void*)p.out), (short)0, (int)(unsigned)obytes, 0x00020000);
            const unsigned vo = ((unsigned)(mw + r0) * (unsigned)p.Cout + (unsigned)(nw + ch * 4)) * 4u;
            const unsigned rstep = 8u * (unsigned)p.Cout * 4u;   // bytes between the rows of consecutive chunks
            constexpr int LG = NOPE_EPILOGUE_LEAN_GROUP;
            auto run = [&](auto resid_tag) {
                constexpr bool RESID = decltype(resid_tag)::value;
#pragma unroll
                for (int pass = 0; pass < 3; ++pass) {
                    if (nw + pass * 32 >= p.Cout) continue;       // (wave-uniform; Cout % 32 == 0: a 32-column pass is whole or absent -- channel counts that are not multiples of 192)
                    __builtin_amdgcn_sched_barrier(0);            // (keeps the next pass's constant loads and panel fill out of this pass: registers)
                    float pc0[4], pc1[4], pb[4];
                    if (PN) {
                        const int nn = nw + pass * 32 + ch * 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            pb[e] = p.bias ? p.bias[nn + e] : 0.f;      // (qkv has no bias: a uniform branch around four loads)
                            pc1[e] = p.pn_c1[nn + e];
                            pc0[e] = p.pn_c0[nn + e] + pb[e];            // v = rstd * (pan - bias - mean * c1) + c0 + bias: the panel holds acc + bias
                        }
                    }
                    const float bv = bvj[pass];
#pragma unroll
                    for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                        for (int r = 0; r < TL::R; ++r)
                            pan[(i * TL::TM + TL::out_row(lane, r)) * Ep<T>::LD + TL::out_col(lane)] = SCALED ? acc[i][pass][r] * asc + bv : acc[i][pass][r] + bv;
                    __builtin_amdgcn_wave_barrier();              // (same-wave LDS write -> read: in-order LDS queue; rendezvous point of tests/hipemu)
                    stamp();
                    if (DRAIN && pass == 0) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
                    stamp();
#pragma unroll
                    for (int grp = 0; grp < 8 / LG; ++grp) {           // LG chunks at a time: LG LDS reads (and residual loads) in flight
                        u32x4 rv[LG];
                        f32x4 t[LG];
                        if (RESID) {
#pragma unroll
                            for (int c = 0; c < LG; ++c) rv[c] = __builtin_amdgcn_raw_buffer_load_b128(rr, vo + (unsigned)(pass * 128) + (unsigned)(grp * LG + c) * rstep, 0, 0);
                        }
#pragma unroll
                        for (int c = 0; c < LG; ++c) t[c] = *reinterpret_cast<const f32x4*>(&pan[(r0 + 8 * (grp * LG + c)) * Ep<T>::LD + ch * 4]);
#pragma unroll
                        for (int c = 0; c < LG; ++c) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = t[c][e];
                            if (PN) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = pn_apply(v[e] - pb[e], pmean, prstd, pc1[e], pc0[e]);
                            }
                            if (RESID) {
                                float rf[4];
                                Elt<T>::unpack(rv[c], rf);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += rf[e];
                            }
                            if (p.act) {                          // (uniform branch: the encoder's convs)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                            }
                            if (track_out) out_max = amax4(out_max, v[0], v[1], v[2], v[3]);
                            __builtin_amdgcn_raw_buffer_store_b128(Elt<T>::pack(v), ro, vo + (unsigned)(pass * 128) + (unsigned)(grp * LG + c) * rstep, 0, 0);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    stamp();
                }
            };
            if (resid) run(BoolTag<true>()); else run(BoolTag<false>());
            if (track_out) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) out_max = __builtin_fmaxf(out_max, __shfl_xor(out_max, o, 64));
                if (lane == 0 && out_max > 0.f) amax_publish(p.out_amax, out_max, (unsigned)blockIdx.x * 8u + (unsigned)(wm * 2 + wn));
            }
            return;
        }
    }
#pragma unroll
    for (int pass = 0; pass < 96 / PANW; ++pass) {
#pragma unroll
        for (int jj = 0; jj < TPP; ++jj) {
            const int j = pass * TPP + jj;
            const float bv = bvj[j];
#pragma unroll
            for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                for (int r = 0; r < TL::R; ++r)
                    pan[(i * TL::TM + TL::out_row(lane, r)) * Ep<T>::LD + jj * TL::TM + TL::out_col(lane)] = SCALED ? acc[i][j][r] * asc + bv : acc[i][j][r] + bv;
        }
        // same-wave LDS write -> read: the LDS queue of a wave is in order, so no s_barrier; the wave
        // barrier only pins the compiler's ordering (and is the rendezvous point of tests/hipemu)
        __builtin_amdgcn_wave_barrier();
        stamp();
        if (DRAIN && pass == 0) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
        stamp();
        // Fused PreNorm operands, hoisted out of the chunk loop where they are loop-invariant: with 64 % CH == 0
        // a lane keeps the same column chunk for the whole pass, and with HW % 64 == 0 the wave's 64 rows are
        // one sample (one mean / rstd).
        // (16-bit types reach this loop with PN only on maps whose samples do not fill whole 64-row blocks -- the 4 x 4 level; the
        //  hoisted operands stay out of that instantiation: next to the packed path above they pushed the kernel into scratch)
        constexpr bool PN_COLS_FIXED = PN && (64 % CH == 0) && !(sizeof(T) == 2 && TL::TM == 32);
        float pc0[VEC], pc1[VEC];
        float pmean = 0.f, prstd = 1.f;
        const bool pn_row_uniform = PN && ((p.Hm * p.Wm) % 64 == 0);
        if (PN) {
            if (PN_COLS_FIXED) {
                const int nn = n0 + wn * 96 + pass * PANW + (lane % CH) * VEC;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const bool okc = nn + e < p.Cout;
                    const float bias_e = (p.bias && okc) ? p.bias[nn + e] : 0.f;
                    pc1[e] = okc ? p.pn_c1[nn + e] : 0.f;
                    pc0[e] = (okc ? p.pn_c0[nn + e] : 0.f) + bias_e;     // v = rstd*(pan - bias - mean*c1) + c0 + bias
                }
            }
            if (pn_row_uniform) {
                const int b = (m0 + wm * 64 < p.M ? m0 + wm * 64 : 0) / (p.Hm * p.Wm);
                pmean = p.pn_ms[2 * b]; prstd = p.pn_ms[2 * b + 1];
            }
        }
        for (int idx = lane; idx < 64 * CH; idx += 64) {
            const int row = idx / CH, ch = idx - row * CH;
            const int m = m0 + wm * 64 + row;
            const int n = n0 + wn * 96 + pass * PANW + ch * VEC;
            if (m >= p.M || n >= p.Cout) continue;
            float v[VEC];
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&pan[row * Ep<T>::LD + ch * VEC + q * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q * 4 + e] = t[e];
            }
            if (PN) {   // fused PreNorm: v = rstd_b * (acc - mean_b * c1[n]) + c0[n] (+ bias; the panel holds acc + bias)
                float mean = pmean, rstd = prstd;
                if (!pn_row_uniform) { const int b = m / (p.Hm * p.Wm); mean = p.pn_ms[2 * b]; rstd = p.pn_ms[2 * b + 1]; }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (PN_COLS_FIXED) {
                        const float bias_e = p.bias ? p.bias[n + e] : 0.f;   // (qkv has no bias: folds away)
                        v[e] = pn_apply(v[e] - bias_e, mean, rstd, pc1[e], pc0[e]);
                    } else {
                        const float bias_e = p.bias ? p.bias[n + e] : 0.f;
                        v[e] = pn_apply(v[e] - bias_e, mean, rstd, p.pn_c1[n + e], p.pn_c0[n + e] + bias_e);
                    }
                }
            }
            const size_t o = out_row(p, m) * p.Cout + n;
            if constexpr (GG && sizeof(T) == 4) {
                // f32 storage: the lane's four columns are two (x, gate) pairs -> two values = 8 bytes per row into the [M][Cout / 2] output (no residual /
                // activation / PreNorm: geglu_shape_ok); same operands and formula as geglu_kernel<float, true>: bit-identical to conv + geglu
                const f32x2_t y = {geglu_f(v[0], v[1]), geglu_f(v[2], v[3])};
                *reinterpret_cast<f32x2_t*>(out + (o >> 1)) = y;
                continue;
            }
            if (resid) {
                float rv[VEC];
                Elt<T>::unpack(ld16(resid + o), rv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] += rv[e];
            }
            if (p.act) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if constexpr (sizeof(T) == 4) { if (track_out) out_max = amax4(out_max, v[0], v[1], v[2], v[3]); }
            st16(out + o, Elt<T>::pack(v));
        }
        __builtin_amdgcn_wave_barrier();
        stamp();
    }
    if constexpr (sizeof(T) == 4) {
        if (track_out) {       // the range slot of the tensor this launch writes (NOPE_F16X2 consumers downstream: unet_runtime.hip)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) out_max = __builtin_fmaxf(out_max, __shfl_xor(out_max, o, 64));
            if (lane == 0 && out_max > 0.f) amax_publish(p.out_amax, out_max, (unsigned)blockIdx.x * 8u + (unsigned)(wm * 2 + wn));
        }
    }
}

// Split-K partials: this workgroup's raw f32 sums to split_out[blockIdx.z][m][n], through the per-wave panel as 16-byte rows.  Written
// straight from the MFMA C/D layout they are 96 four-byte store instructions per wave (measured on the tap-resident kernel at
// 4 x 4 x 64 hypotheses: ~25 of the 45 us of a 27-step workgroup); staged, 24 sixteen-byte ones.  Cout % 4 == 0 (checked by the launcher:
// Cout % VEC).  Bias, residual and activation are applied by splitk_reduce_kernel, which adds the partials in a fixed order.
template <class T>
__device__ __forceinline__ void epilogue_split_wide(const ConvParams& p, const typename Tile<T>::acc_t (&acc)[Tile<T>::MT][Tile<T>::NTL], int m0, int n0,
                                                    int wm, int wn, int lane, unsigned char* lds_wave, float acc_scale = 1.0f) {
    typedef Tile<T> TL;
    constexpr int PANW = TL::TM == 32 ? 32 : 48, TPP = PANW / TL::TM, LD = Ep<T>::LD, CH = PANW / 4;
    constexpr bool SCALED = Elt<T>::DT == NOPE_F16X2;      // (see epilogue_wide)
    float* pan = reinterpret_cast<float*>(lds_wave);
    float* so = p.split_out + (size_t)blockIdx.z * p.M * p.Cout;
#pragma unroll
    for (int pass = 0; pass < 96 / PANW; ++pass) {
#pragma unroll
        for (int jj = 0; jj < TPP; ++jj)
#pragma unroll
            for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                for (int r = 0; r < TL::R; ++r)
                    pan[(i * TL::TM + TL::out_row(lane, r)) * LD + jj * TL::TM + TL::out_col(lane)] = SCALED ? acc[i][pass * TPP + jj][r] * acc_scale : acc[i][pass * TPP + jj][r];
        __builtin_amdgcn_wave_barrier();       // same-wave LDS write -> read (in-order LDS queue; rendezvous point of tests/hipemu)
        for (int idx = lane; idx < 64 * CH; idx += 64) {
            const int row = idx / CH, ch = idx - row * CH;
            const int m = m0 + wm * 64 + row, n = n0 + wn * 96 + pass * PANW + ch * 4;
            if (m < p.M && n < p.Cout) *reinterpret_cast<f32x4*>(so + (size_t)m * p.Cout + n) = *reinterpret_cast<const f32x4*>(&pan[row * LD + ch * 4]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// NCHW output (the last conv writes the (hypothesis, C, h, w) template bank directly) through the per-wave panel: the MFMA C/D layout
// gives a lane one channel x a few pixels, i.e. 2-byte stores scattered over Cout planes (the 8-channel final conv ran at
// 1.2 TB/s of input); staged, lane r owns pixel r of the wave's 64 and every channel plane receives 64 consecutive
// pixels per store instruction.  Requires HWo % 64 == 0 and M % 64 == 0 (checked by the launcher), no residual.
template <class T>
__device__ __forceinline__ void epilogue_nchw(const ConvParams& p, const typename Tile<T>::acc_t (&acc)[Tile<T>::MT][Tile<T>::NTL], int m0, int n0,
                                              int wm, int wn, int lane, unsigned char* lds_wave) {
    typedef Tile<T> TL;
    constexpr int PANW = TL::TM == 32 ? 32 : 48;
    constexpr int TPP = PANW / TL::TM;
    float* pan = reinterpret_cast<float*>(lds_wave);
    const int HWo = p.Ho * p.Wo;
    const int mrow = m0 + wm * 64;
    if (mrow >= p.M || n0 + wn * 96 >= p.Cout) return;            // (wave-uniform)
    const int b = mrow / HWo, pix0 = mrow - b * HWo;
    const size_t base = (size_t)b * p.Cout * HWo + pix0 + lane;
#pragma unroll
    for (int pass = 0; pass < 96 / PANW; ++pass) {
        if (n0 + wn * 96 + pass * PANW >= p.Cout) break;            // (wave-uniform)
#pragma unroll
        for (int jj = 0; jj < TPP; ++jj) {
            const int j = pass * TPP + jj;
            const int n = n0 + wn * 96 + j * TL::TM + TL::out_col(lane);
            const float bv = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                for (int r = 0; r < TL::R; ++r) {
                    float v = acc[i][j][r] + bv;
                    if (p.act) v = v > 0.f ? v : 0.f;
                    pan[(i * TL::TM + TL::out_row(lane, r)) * Ep<T>::LD + jj * TL::TM + TL::out_col(lane)] = v;
                }
        }
        __builtin_amdgcn_wave_barrier();       // same-wave LDS write -> read (in-order LDS queue; rendezvous point of tests/hipemu)
        for (int cc = 0; cc < PANW; ++cc) {
            const int n = n0 + wn * 96 + pass * PANW + cc;
            if (n >= p.Cout) break;
            const float v = pan[lane * Ep<T>::LD + cc];
            const size_t o = base + (size_t)n * HWo;
            if (p.out_dt == NOPE_F32) reinterpret_cast<float*>(p.out)[o] = v;
            else if (p.out_dt == NOPE_F16) reinterpret_cast<f16_t*>(p.out)[o] = f32_to_f16_sat(v);
            else reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(v);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

typedef __attribute__((address_space(3))) void lds_void_t;
constexpr unsigned OOB = 0x80000000u;   // >= any num_records (tensors < 2 GiB); stays out of range after adding a K offset

}  // namespace

}  // namespace nope
