// Implicit-GEMM convolution for launches that CANNOT fill the chip with 128 x 192 tiles: the reference's own workloads.
// The reference evaluates on 26 / 91 / 341-template banks (src/dataloader/shapeNet.py:248-263) one query at a time
// (model.py:212-222), and template-parallel sharding leaves 64 templates per GPU of a 512-template bank: at 64 pose hypotheses
// the 16 x 16 / 8 x 8 / 4 x 4 levels of the U-Net have 128 / 64 / 32 tiles of 128 x 192 for 256 CUs, the one-image encoder pass
// (encoder/resnet.py:135-152) 8-256, and every launch of the big kernels costs >= 8-15 us whatever its size (a 96-accumulator
// epilogue through LDS, a 600-instruction prologue, one exposed memory round trip per K step).  Round 3 papered over that
// with split-K + a reduce launch (92 reduce launches per 64-hypothesis step).
//
// This kernel is built for LATENCY instead:
//   * small output tiles, 64 x 64 (four waves as 2 x 2, each ONE 32 x 32 MFMA tile: 16 accumulator registers) or 128 x 128
//     (2 x 2 tiles per wave) -- 4-9x the workgroups of the 128 x 192 tiling, no split-K, no reduce launch;
//   * an NS-stage LDS ring fed by LDS-DMA (buffer_load ... lds) with COUNTED waits: the pieces of K steps k+1 .. k+NS-2 are in
//     flight while step k multiplies (s_waitcnt vmcnt(n) retires only the oldest stage, raw s_barrier publishes it: hipcc's
//     __syncthreads() would drain the whole queue), so a tile pays one memory latency per tile, not one per K step;
//   * a short prologue (2-4 rows per lane) and a one-pass epilogue: the 64 x 64 f32 tile goes through one LDS panel and
//     leaves as 16-byte rows with bias / fused PreNorm / residual / ReLU applied, the GroupNorm column statistics of the
//     next layer are folded from the same panel (fixed order, one writer per entry), NCHW output is written plane by plane.
// Same implicit A operand as the other kernels (taps, virtual concat of two sources, space-to-depth, 2 x 2 phase convs,
// stride 2), same source-side XOR swizzle, same K order (channel chunk outer, tap inner): sums differ from the
// 128 x 192 kernels' only by the association of the f32 partial sums across K steps (none: the K order is identical and
// every output element is accumulated by ONE wave in ONE accumulator -- results are bit-identical to an unsplit launch
// of the other kernels, tests/test_conv_small.py).
#include <cstdio>
#include <cstdlib>

#include "conv_gemm_common.h"

namespace nope {

namespace {

// one fragment's worth of MFMA work, per element type (the Tile<T> traits fix the 64 x 96 wave tile of the big kernels)
// (a_frag_slot / a_raw_slot: which 16-byte slots of a staged A row a lane reads -- the B row's, except for NOPE_F16X2, whose A rows are staged
//  as raw f32 and split in registers; `sc`: the E8M0 block scale of NOPE_F16X2's cross-term MFMA, unused by the other types)
template <class T> struct FragSlots {
    static __device__ __forceinline__ int a_frag_slot(int lane) { return Tile<T>::frag_slot(lane); }
    static __device__ __forceinline__ constexpr int a_raw_slot(int q) { return raw_slot<T>(q); }
};
template <class T> struct Frag;
template <> struct Frag<float> : FragSlots<float> {
    static __device__ __forceinline__ void prep(u32x4 (&)[1]) {}
    static __device__ __forceinline__ void mma(int, const u32x4 (&a)[1], const u32x4 (&b)[1], f32x4& c, int = 0) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a[0]), fb = __builtin_bit_cast(f32x4, b[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[q], fb[q], c, 0, 0, 0);
    }
};
template <> struct Frag<bf16_t> : FragSlots<bf16_t> {
    static __device__ __forceinline__ void prep(u32x4 (&)[1]) {}
    static __device__ __forceinline__ void mma(int, const u32x4 (&a)[1], const u32x4 (&b)[1], f32x16& c, int = 0) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), c, 0, 0, 0);
    }
};
template <> struct Frag<f16_t> : FragSlots<f16_t> {
    static __device__ __forceinline__ void prep(u32x4 (&)[1]) {}
    static __device__ __forceinline__ void mma(int, const u32x4 (&a)[1], const u32x4 (&b)[1], f32x16& c, int = 0) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
    }
};
template <> struct Frag<f32s_t> : FragSlots<f32s_t> {      // NOPE_BF16X3: see Tile<f32s_t>
    static __device__ __forceinline__ void prep(u32x4 (&a)[2]) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned u0 = a[0][e], u1 = a[1][e];
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
        a[0] = hi; a[1] = lo;
    }
    static __device__ __forceinline__ void mma(int t, const u32x4 (&a)[2], const u32x4 (&b)[2], f32x16& c, int = 0) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t == 0 ? 1 : 0]), __builtin_bit_cast(bf16x8, b[t == 1 ? 1 : 0]), c, 0, 0, 0);
    }
};
// NOPE_F16X2 (round 6: the reference-sized banks, an 8-way shard of the 512-template bank): Tile<f16x2_t>'s operands -- B rows from the layer's
// second pack, A rows staged as raw f32 (the bf16x3 loaders unchanged) and split in registers exactly as the per-tap ping-pong kernel
// splits them: a lane of half h reads the 16 f32 channels of its channel set (raw slots 2 h, 2 h + 1, 4 + 2 h, 5 + 2 h) and forms
// f16(a') x 2, e4m3(a'_lo * 2^9), e4m3(a' * 2^-2) from a' = a * 2^-t (the layer's range shift; saturating conversions: MODE.FP16_OVFL).
template <> struct Frag<f16x2_t> {
    static __device__ __forceinline__ int a_frag_slot(int lane) { return Tile<f16x2_t>::frag_slot_raw(lane); }
    static __device__ __forceinline__ constexpr int a_raw_slot(int q) { return Tile<f16x2_t>::raw_slot_a(q); }
    static __device__ __forceinline__ void prep(u32x4 (&a)[4], float inv, float div_a, float& amax) {
        u32x4 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const unsigned u = a[q][e]; v[e] = __builtin_bit_cast(float, u); }
            if (NOPE_X2_KERNEL_AMAX) amax = amax4(amax, v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x2_t t2 = f32x2_t{v[2 * e], v[2 * e + 1]} * inv;
                const unsigned h = NOPE_CVT_PK_F16_OVFL(t2.x, t2.y);
                x[q >> 1][2 * (q & 1) + e] = h;
                union { unsigned u; f16_t f[2]; } hh; hh.u = h;
                l[2 * e] = __builtin_fmaf(v[2 * e], inv, -(float)hh.f[0]);
                l[2 * e + 1] = __builtin_fmaf(v[2 * e + 1], inv, -(float)hh.f[1]);
            }
            x[2][q] = cvt4_e4m3_scaled<kX2ALoShift, true>(l[0], l[1], l[2], l[3]);
            x[3][q] = cvt4_e4m3_div(v[0], v[1], v[2], v[3], div_a);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = x[q];
    }
    static __device__ __forceinline__ void mma(int t, const u32x4 (&a)[4], const u32x4 (&b)[4], f32x16& c, int sc) {
        if (t < 2) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t]), __builtin_bit_cast(f16x8, b[t]), c, 0, 0, 0);
        } else {
            const i32x8 va = {(int)a[2][0], (int)a[2][1], (int)a[2][2], (int)a[2][3], (int)a[3][0], (int)a[3][1], (int)a[3][2], (int)a[3][3]};
            const i32x8 vb = {(int)b[2][0], (int)b[2][1], (int)b[2][2], (int)b[2][3], (int)b[3][0], (int)b[3][1], (int)b[3][2], (int)b[3][3]};
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, 0, sc, 0, 127);
        }
    }
};

constexpr int S_WAIT_LGKMCNT0 = 0xC07F;
// s_waitcnt vmcnt(n), other counters at their maximum (gfx9 encoding: vmcnt = [3:0] | [15:14] << 4)
template <int N> __device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }

// WMT x WNT: 32 x 32 blocks per wave (waves 2 x 2): tile = 64 WMT x 64 WNT.  NS: ring stages of (64 WMT + 64 WNT) x 128 B.
// KG = 2: TWO wave groups (512 threads) share the tile and take alternate K steps, each through its own ring; their f32 sums are added
// (group 0 + group 1, a fixed order) in the epilogue panel.  For launches with at most one workgroup per CU and a long K -- the
// one-image encoder's 3x3 convs: 64-128 workgroups of 36-72 K steps, 0.45 us per step -- where a wave's K step is a serial chain
// (wait, barrier, issue four DMA pieces at ~100 cycles each, eight ds_reads, four MFMAs): a deeper ring does not shorten that chain
// (measured: the 6-stage ring is no faster), a second group halves the number of links.  Not bit-identical to the other kernels
// (the sum over K is associated as (even steps) + (odd steps)); deterministic.
template <class T, int MODE, int WMT, int WNT, int NS, bool PN, int KG = 1>
__global__ __launch_bounds__(256 * KG) void conv_gemm_small_kernel(ConvParams p) {
    typedef Tile<T> TL;
    constexpr int VEC = Elt<T>::VEC;
    constexpr unsigned ES = (unsigned)sizeof(T);
    constexpr int RB = 128, BK = RB / (int)ES;
    constexpr int BMS = 64 * WMT, BNS = 64 * WNT;
    constexpr int STAGE = (BMS + BNS) * RB;
    constexpr int AI = 2 * WMT, BI = 2 * WNT, L = AI + BI;          // 1 KiB DMA pieces per wave and stage
    constexpr int MT = 32 * WMT / TL::TM, NTL = 32 * WNT / TL::TM;  // MFMA tiles per wave
    constexpr int KS = RB / 16 / TL::STEP_SLOTS, RAW = TL::RAW;
    constexpr int LDP = BNS + 4;                                    // f32 panel row stride (words)
    constexpr int RING = NS * STAGE, PANEL = BMS * LDP * 4 + 4 * BNS * 2 * 4;
    constexpr int NTH = 256 * KG;
    static_assert(KG == 1 || (KG == 2 && RING >= BMS * LDP * 4), "the second group's panel sits in its own ring");
    constexpr int LDS_BYTES = KG * RING > PANEL ? KG * RING : PANEL;
    static_assert(NS >= 2 && NS <= 6 && (NS - 2) * L <= 63, "vmcnt field");
    static_assert(MODE != NOPE_CONV_UP2, "the nearest-x2 + 3x3 form runs as four phase convs");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = KG > 1 ? wave8 >> 2 : 0, wave = wave8 & 3;          // K group, wave inside the group
    constexpr bool X2 = Elt<T>::DT == NOPE_F16X2;
    if constexpr (X2) fp16_ovfl_on();
    const int x2_t = (X2 && NOPE_X2_TRACK) ? p.x2_scale[3] : 0;                            // the layer's activation range shift (nope_common.h: kX2*)
    const int x2_sc = X2 ? p.x2_scale[0] : 0;
    const float x2_inv = x2_pow2(-x2_t), x2_out = x2_pow2(x2_t), x2_da = x2_pow2(x2_t - kX2AShift);
    float x2_amax = 0.f;
    const int wm = wave >> 1, wn = wave & 1;
    unsigned char* const ring = lds + grp * RING;                       // this group's ring
    int tile_m, tile_n;
    {
        const int g = blockIdx.x;
        if (p.xcd_map == 3) {       // (8 / gn) x gn XCD grid: XCD (xm, xn) owns a run of M tiles and tiles_n / gn panels, panels fastest
            const int x = g & 7, j = g >> 3, gn = p.xcd_gn;
            const int span = p.tiles_n / gn, run = p.tiles_m / (8 / gn);
            const int xm = x / gn, xn = x - xm * gn;
            tile_n = xn * span + j % span;
            tile_m = xm * run + j / span;
        } else { tile_n = g % p.tiles_n; tile_m = g / p.tiles_n; }
    }
    const int m0 = tile_m * BMS, n0 = tile_n * BNS;
    const int HWo = p.Hm * p.Wm;
    const int Cin = p.C1 + p.C2;
    const int ph_y = MODE == NOPE_CONV_UP2P ? ((int)blockIdx.y >> 1) : 0, ph_x = MODE == NOPE_CONV_UP2P ? ((int)blockIdx.y & 1) : 0;

    const auto r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src1, (short)0, (int)p.bytes1, 0x00020000);
    const auto r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src2 ? p.src2 : p.src1), (short)0, (int)(p.src2 ? p.bytes2 : p.bytes1), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (size_t)blockIdx.y * p.w_phase_bytes), (short)0, (int)p.bytesw, 0x00020000);

    // ---- this lane's rows of the DMA pieces: wave w stages A rows 8 (AI w + i) .. + 7 and B rows 8 (BI w + j) .. + 7
    const int rsub = lane >> 3, lslot = lane & 7;
    // (fixed-size arrays: with template-sized arrays captured by the `issue` lambda hipcc (ROCm 7.2) drops the kernel's host stub)
    static_assert(AI <= 4 && BI <= 4, "row bookkeeping arrays");
    unsigned a_b1[4], a_b2[4], a_mask[4];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = 8 * (AI * wave + i) + rsub;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const unsigned mm = ok ? (unsigned)m : 0u;
        const unsigned b = p.d_hw.div(mm), r = mm - b * (unsigned)HWo;
        const int oy = (int)p.d_w.div(r), ox = (int)r - oy * p.Wm;
        const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(row)) * VEC);   // source channel chunk of this LDS slot
        const unsigned s1 = p.d_rep1.div(b), s2 = p.d_rep2.div(b);
        unsigned mask = 0;
        auto mask3x3 = [](int y, int x, int H, int W) {
            const unsigned vx = (x > 0 ? 1u : 0u) | (x >= 0 && x < W ? 2u : 0u) | (x + 1 < W ? 4u : 0u);
            return (y > 0 ? vx : 0u) | (y >= 0 && y < H ? vx << 3 : 0u) | (y + 1 < H ? vx << 6 : 0u);
        };
        if (MODE == NOPE_CONV_UP2P) {
            a_b1[i] = (((s1 * p.Hs + oy) * p.Ws + ox) * p.C1 + cs) * ES;
            a_b2[i] = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int iy = oy + (t >> 1) + ph_y - 1, ix = ox + (t & 1) + ph_x - 1;
                if (iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws) mask |= 1u << t;
            }
        } else if (MODE == NOPE_CONV_PLAIN) {
            a_b1[i] = (((s1 * p.Hs + oy) * p.Ws + ox) * p.C1 + cs) * ES;
            a_b2[i] = (((s2 * p.Hs + oy) * p.Ws + ox) * p.C2 + cs) * ES;
            mask = p.ntaps == 9 ? mask3x3(oy, ox, p.Hs, p.Ws) : 1u;
        } else if (MODE == NOPE_CONV_DOWN2) {
            a_b1[i] = (((s1 * p.Hs + 2 * oy) * p.Ws + 2 * ox) * p.C1 + cs) * ES;
            a_b2[i] = 0;
            mask = 0xfu;
        } else {   // STRIDE2: centre tap at source pixel (2 oy, 2 ox)
            a_b1[i] = (((s1 * p.Hs + 2 * oy) * p.Ws + 2 * ox) * p.C1 + cs) * ES;
            a_b2[i] = 0;
            mask = p.ntaps == 9 ? mask3x3(2 * oy, 2 * ox, p.Hs, p.Ws) : 1u;
        }
        a_mask[i] = ok ? mask : 0u;
    }
    unsigned b_off[4];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = 8 * (BI * wave + j) + rsub;
        const int n = n0 + row;
        const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(row)) * VEC);
        b_off[j] = n < p.Cout ? ((unsigned)n * p.ntaps * Cin + cs) * ES : OOB;
    }

    const int kc_per_tap = Cin / BK;
    int ks0 = 0, nk = p.ntaps * kc_per_tap;
    if (p.splits > 1) {                       // split-K: blockIdx.z owns K steps [ks0, ks0 + nk)
        const int tot = nk, z = (int)blockIdx.z;
        ks0 = (int)((long long)z * tot / p.splits);
        nk = (int)((long long)(z + 1) * tot / p.splits) - ks0;
    }
    // KG = 2: group g takes steps ks0 + g, ks0 + g + 2, ...
    const int nk_all = nk;
    if (KG > 1) { ks0 += grp; nk = (nk_all - grp + 1) >> 1; }
    int ld_kc = ks0 / p.ntaps, ld_tap = ks0 - (ks0 / p.ntaps) * p.ntaps;
    // K order: channel chunk outer, tap inner (as the other kernels: the sum order over K is identical)
    auto issue = [&](int stage) {
        unsigned char* dA = ring + stage * STAGE + (AI * wave) * 1024;
        unsigned char* dB = ring + stage * STAGE + BMS * RB + (BI * wave) * 1024;
        const int c0 = ld_kc * BK;
        const bool first = c0 < p.C1;                   // wave-uniform: a K step lies inside one source
        const int Cs = first ? p.C1 : p.C2;
        unsigned kadd = (unsigned)(first ? c0 : c0 - p.C1) * ES;
        if (MODE == NOPE_CONV_PLAIN || MODE == NOPE_CONV_STRIDE2) {
            if (p.ntaps == 9) {
                const int dyi = ld_tap / 3, dxi = ld_tap - dyi * 3;
                kadd += (unsigned)(((dyi - 1) * p.Ws + (dxi - 1)) * Cs) * ES;
            }
        } else if (MODE == NOPE_CONV_DOWN2) {
            kadd += (unsigned)(((ld_tap >> 1) * p.Ws + (ld_tap & 1)) * Cs) * ES;
        } else {
            kadd += (unsigned)((((ld_tap >> 1) + ph_y - 1) * p.Ws + ((ld_tap & 1) + ph_x - 1)) * Cs) * ES;
        }
        const unsigned kofs = (unsigned)(ld_tap * Cin + c0) * ES;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const unsigned base = first ? a_b1[i] : a_b2[i];
            const unsigned off = (((a_mask[i] >> ld_tap) & 1u) ? base : OOB) + kadd;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? r1 : r2, (lds_void_t*)(dA + i * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(dB + j * 1024), 16, b_off[j] + kofs, 0, 0, 0);
#pragma unroll
        for (int adv = 0; adv < KG; ++adv)
            if (++ld_tap == p.ntaps) { ld_tap = 0; ++ld_kc; }
    };

    typename TL::acc_t acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < TL::R; ++r) acc[i][j][r] = 0.f;

    int fa[MT], fb[NTL];        // fragment addresses of raw read 0 inside a stage; raw read q flips slot bits
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = lds_off_rb<RB>(wm * 32 * WMT + i * TL::TM + TL::frag_row(lane), Frag<T>::a_frag_slot(lane));
#pragma unroll
    for (int j = 0; j < NTL; ++j) fb[j] = BMS * RB + lds_off_rb<RB>(wn * 32 * WNT + j * TL::TM + TL::frag_row(lane), TL::frag_slot(lane));

    // ---- prologue: NS - 1 stages in flight
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s);
    int st = 0;                                     // ring slot of K step ks
    const int nloop = KG > 1 ? (nk_all + 1) >> 1 : nk;     // (both groups pass the same barriers; group 1 may have one step less)
    for (int ks = 0; ks < nloop; ++ks) {
        if (KG > 1 && ks >= nk) {                   // group 1's missing last step: only the rendezvous
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
        // Stage ks is this wave's OLDEST outstanding group; min(NS - 2, nk - 1 - ks) younger groups may stay in flight.
        const int younger = nk - 1 - ks;
        if (NS >= 6 && younger >= 4) wait_vmcnt<(NS >= 6 ? 4 * L : 0)>();
        else if (NS >= 5 && younger >= 3) wait_vmcnt<(NS >= 5 ? 3 * L : 0)>();
        else if (NS >= 4 && younger >= 2) wait_vmcnt<(NS >= 4 ? 2 * L : 0)>();
        else if (NS >= 3 && younger >= 1) wait_vmcnt<(NS >= 3 ? L : 0)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();              // every wave's pieces of stage ks have landed; everyone is done with stage ks - 1
        __builtin_amdgcn_sched_barrier(0);
        if (ks + NS - 1 < nk) issue(st == 0 ? NS - 1 : st - 1);     // into the slot of stage ks - 1
        const unsigned char* base = ring + st * STAGE;
#pragma unroll
        for (int kq = 0; kq < KS; ++kq) {
            u32x4 af[MT][RAW], bfr[NTL][RAW];
#pragma unroll
            for (int r = 0; r < RAW; ++r) {
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i][r] = ld16(base + (fa[i] ^ (Frag<T>::a_raw_slot(kq * RAW + r) << 4)));
#pragma unroll
                for (int j = 0; j < NTL; ++j) bfr[j][r] = ld16(base + (fb[j] ^ (raw_slot<T>(kq * RAW + r) << 4)));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (X2) Frag<T>::prep(af[i], x2_inv, x2_da, x2_amax);
                else Frag<T>::prep(af[i]);
            }
#pragma unroll
            for (int t = 0; t < TL::TERMS; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) Frag<T>::mma(t, af[i], bfr[j], acc[i][j], x2_sc);
        }
        __builtin_amdgcn_s_waitcnt(S_WAIT_LGKMCNT0);    // my reads of stage ks are done before I pass the next barrier
        st = st + 1 == NS ? 0 : st + 1;
    }
    __builtin_amdgcn_s_barrier();                  // the ring is free: every read is behind this barrier, every DMA has been waited for
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (X2 && NOPE_X2_KERNEL_AMAX) x2_publish_amax(p, x2_amax, lane);
    // ---- epilogue: the tile's f32 accumulators through ONE panel, then rows out (NOPE_F16X2: x 2^t, the accumulators hold 2^-t x the sums)
    float* pan = reinterpret_cast<float*>(lds);
    {
        float* mine = reinterpret_cast<float*>(ring);          // (group 0: the panel itself)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int r = 0; r < TL::R; ++r)
                    mine[(wm * 32 * WMT + i * TL::TM + TL::out_row(lane, r)) * LDP + wn * 32 * WNT + j * TL::TM + TL::out_col(lane)] = X2 ? acc[i][j][r] * x2_out : acc[i][j][r];
    }
    __syncthreads();
    if (KG > 1) {                                  // sum of the two K groups, group 0 first
        const float* other = reinterpret_cast<const float*>(lds + RING);
        for (int idx = tid; idx < BMS * (BNS / 4); idx += NTH) {
            const int row = idx / (BNS / 4), ch = idx - row * (BNS / 4);
            f32x4* d = reinterpret_cast<f32x4*>(&pan[row * LDP + ch * 4]);
            *d = *d + *reinterpret_cast<const f32x4*>(&other[row * LDP + ch * 4]);
        }
        __syncthreads();
    }

    if (p.splits > 1) {                            // raw f32 partial sums; splitk_reduce_kernel finishes (bias, residual, activation)
        float* so = p.split_out + (size_t)blockIdx.z * p.M * p.Cout;
        for (int idx = tid; idx < BMS * (BNS / 4); idx += NTH) {
            const int row = idx / (BNS / 4), ch = idx - row * (BNS / 4);
            const int m = m0 + row, n = n0 + ch * 4;
            if (m >= p.M || n >= p.Cout) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&pan[row * LDP + ch * 4]);
            if (n + 4 <= p.Cout) *reinterpret_cast<f32x4*>(&so[(size_t)m * p.Cout + n]) = v;
            else for (int e = 0; e < 4 && n + e < p.Cout; ++e) so[(size_t)m * p.Cout + n + e] = v[e];
        }
        return;
    }
    if (p.colstats) {
        // GroupNorm statistics of the conv output (f32, before the rounding to T, bias included): per block of stat_rows (16 / 32 /
        // 64) rows and column (sum, sum of squares).  Four threads per column add 16 rows each; the 16-row partials of a block
        // are then added in row order by one thread: fixed order, one writer per entry.
        float* part = pan + BMS * LDP;             // [4][BNS][2]
        const int qpb = p.stat_rows >> 4;          // 16-row quarters per statistics block: 1, 2 or 4
        for (int blk = 0; blk < WMT; ++blk) {
            for (int c = tid & 63; c < BNS && tid < 256; c += 64) {        // (the first four waves: one 16-row quarter each)
                const int q = tid >> 6;
                const int n = n0 + c;
                const float bv = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;
                float s = 0.f, sq = 0.f;
#pragma unroll 4
                for (int r = 0; r < 16; ++r) { const float v = pan[(blk * 64 + q * 16 + r) * LDP + c] + bv; s += v; sq += v * v; }
                part[(q * BNS + c) * 2] = s; part[(q * BNS + c) * 2 + 1] = sq;
            }
            __syncthreads();
            for (int idx = tid; idx < BNS * (4 / qpb); idx += NTH) {
                const int c = idx % BNS, ob = idx / BNS;
                const int n = n0 + c, mrow = m0 + blk * 64 + ob * p.stat_rows;
                if (n < p.Cout && mrow < p.M) {
                    float s = 0.f, sq = 0.f;
                    for (int q = ob * qpb; q < (ob + 1) * qpb; ++q) { s += part[(q * BNS + c) * 2]; sq += part[(q * BNS + c) * 2 + 1]; }
                    float* cs = p.colstats + ((size_t)(mrow / p.stat_rows) * p.Cout + n) * 2;
                    cs[0] = s; cs[1] = sq;
                }
            }
            if (WMT > 1) __syncthreads();
        }
    }
    if (p.out_nchw) {
        // (hypothesis, Cout, Ho, Wo) planes: consecutive threads take consecutive pixels of one channel
        const int ncols = p.Cout - n0 < BNS ? p.Cout - n0 : BNS;
        for (int idx = tid; idx < BMS * ncols; idx += NTH) {
            const int c = idx / BMS, row = idx - c * BMS;
            const int m = m0 + row, n = n0 + c;
            if (m >= p.M) continue;
            float v = pan[row * LDP + c] + (p.bias ? p.bias[n] : 0.f);
            if (p.resid) v += Elt<T>::ld(reinterpret_cast<const T*>(p.resid) + (size_t)m * p.Cout + n);      // (NHWC residual, as the generic epilogue; a fused PreNorm never reaches this kernel with an NCHW output: plan_small)
            if (p.act) v = v > 0.f ? v : 0.f;
            const int b = m / HWo;
            const size_t o = ((size_t)b * p.Cout + n) * HWo + (m - b * HWo);
            if (p.out_dt == NOPE_F32) reinterpret_cast<float*>(p.out)[o] = v;
            else if (p.out_dt == NOPE_F16) reinterpret_cast<f16_t*>(p.out)[o] = f32_to_f16_sat(v);
            else reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(v);
        }
        return;
    }
    T* out = reinterpret_cast<T*>(p.out);
    const T* resid = reinterpret_cast<const T*>(p.resid);
    constexpr int CH = BNS / VEC;
    if (p.wide_out) {
        // per-column constants of the tile, once: out = acc * sc_row + off[n] (bias; fused PreNorm: off = c0 + bias, and -mean * rstd * c1[n]
        // per row) -- read from LDS in the row loop instead of three global loads per element
        float* cbias = pan + BMS * LDP;            // [BNS] bias (+ c0), [BNS] c1   (the statistics scratch is free again)
        if (p.colstats) __syncthreads();
        for (int c = tid; c < BNS; c += NTH) {
            const int n = n0 + c;
            const bool okc = n < p.Cout;
            cbias[c] = (okc && p.bias ? p.bias[n] : 0.f) + (PN && okc ? p.pn_c0[n] : 0.f);
            if (PN) cbias[BNS + c] = okc ? p.pn_c1[n] : 0.f;
        }
        __syncthreads();
        for (int idx = tid; idx < BMS * CH; idx += NTH) {
            const int row = idx / CH, ch = idx - row * CH;
            const int m = m0 + row, n = n0 + ch * VEC;
            if (m >= p.M || n >= p.Cout) continue;          // (Cout % VEC == 0: whole chunks)
            float v[VEC];
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&pan[row * LDP + ch * VEC + q * 4]);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(&cbias[ch * VEC + q * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q * 4 + e] = t[e];
                if (!PN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q * 4 + e] += bq[e];
                }
            }
            if (PN) {       // fused PreNorm: rstd_b * (acc - mean_b * c1[n]) + c0[n] (+ bias)
                const int b = m / HWo;
                const float mean = p.pn_ms[2 * b], rstd = p.pn_ms[2 * b + 1];
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = pn_apply(v[e], mean, rstd, cbias[BNS + ch * VEC + e], cbias[ch * VEC + e]);
            }
            const size_t o = out_row(p, m) * p.Cout + n;
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
            st16(out + o, Elt<T>::pack(v));
        }
        return;
    }
    // any Cout (not a whole number of 16-byte vectors): one element per thread and step
    const int ncols = p.Cout - n0 < BNS ? p.Cout - n0 : BNS;
    for (int idx = tid; idx < BMS * ncols; idx += NTH) {
        const int row = idx / ncols, c = idx - row * ncols;
        const int m = m0 + row, n = n0 + c;
        if (m >= p.M) continue;
        float v = pan[row * LDP + c];
        if (PN) { const int b = m / HWo; v = pn_apply(v, p.pn_ms[2 * b], p.pn_ms[2 * b + 1], p.pn_c1[n], p.pn_c0[n]); }
        if (p.bias) v += p.bias[n];
        const size_t o = out_row(p, m) * p.Cout + n;
        if (resid) v += Elt<T>::ld(resid + o);
        if (p.act) v = v > 0.f ? v : 0.f;
        Elt<T>::st(out + o, v);
    }
}

template <class T, int WMT, int WNT, int NS, int KG = 1>
void launch_small_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    const dim3 block(256 * KG);
    if (p.pn_ms) hipLaunchKernelGGL((conv_gemm_small_kernel<T, NOPE_CONV_PLAIN, WMT, WNT, NS, true, KG>), grid, block, 0, s, p);
    else if (p.mode == NOPE_CONV_PLAIN) hipLaunchKernelGGL((conv_gemm_small_kernel<T, NOPE_CONV_PLAIN, WMT, WNT, NS, false, KG>), grid, block, 0, s, p);
    else if (p.mode == NOPE_CONV_UP2P) hipLaunchKernelGGL((conv_gemm_small_kernel<T, NOPE_CONV_UP2P, WMT, WNT, NS, false, KG>), grid, block, 0, s, p);
    else if (p.mode == NOPE_CONV_STRIDE2) hipLaunchKernelGGL((conv_gemm_small_kernel<T, NOPE_CONV_STRIDE2, WMT, WNT, NS, false, KG>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_gemm_small_kernel<T, NOPE_CONV_DOWN2, WMT, WNT, NS, false, KG>), grid, block, 0, s, p);
}

}  // namespace

// tile: 0 = 64 x 64 (3-stage ring, 48 KiB: three workgroups per CU), 1 = 128 x 128 (3 stages, 96 KiB), 3 = 64 x 64 by two K groups (512
// threads, two 3-stage rings), 2 = 64 x 64 with a 6-stage ring
// (96 KiB: four K steps in flight per workgroup -- launches with at most a workgroup or two per CU and a long K, where a tile's time
// is its chain of memory round trips: the one-image encoder's 3x3 convs ran 0.45 us per K step on the 3-stage ring)
void launch_conv_small(int dt, const void* params, int tile, dim3 grid, hipStream_t s) {
    const ConvParams& p = *reinterpret_cast<const ConvParams*>(params);
#define NOPE_SMALL_T(T)                                                        \
    do {                                                                       \
        if (tile == 1) launch_small_t<T, 2, 2, 3>(p, grid, s);                 \
        else if (tile == 3) launch_small_t<T, 1, 1, 3, 2>(p, grid, s);         \
        else if (tile == 2) launch_small_t<T, 1, 1, 6>(p, grid, s);            \
        else launch_small_t<T, 1, 1, 3>(p, grid, s);                           \
    } while (0)
    if (dt == NOPE_F32) NOPE_SMALL_T(float);
    else if (dt == NOPE_BF16X3 && p.x2_scale) NOPE_SMALL_T(f16x2_t);      // the layer's second pack: launch_conv took it (plan_takes_x2)
    else if (dt == NOPE_BF16X3) NOPE_SMALL_T(f32s_t);
    else if (dt == NOPE_F16) NOPE_SMALL_T(f16_t);
    else NOPE_SMALL_T(bf16_t);
#undef NOPE_SMALL_T
}

}  // namespace nope
