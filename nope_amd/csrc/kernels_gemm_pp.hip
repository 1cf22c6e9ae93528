// Implicit-GEMM convolution, 256 x 192 tile, eight waves in two ping-pong groups -- the kernel for the launches that
// fill the chip (levels 0-2 of the U-Net at hundreds of pose hypotheses; model_utils.py:240,269 / u_net.py:106,143).
//
// Why another tile.  The 128 x 192 kernel (kernels_gemm.hip) runs two independent 4-wave workgroups per CU; each stages
// its own copy of the 192-row weight panel, so a K step moves (128 + 192) x 128 B = 40 KiB through the L2 -> LDS path for
// 3.1 MFLOP.  Round-1 cycle counters showed that stream alone occupying 80 % of the kernel's cycles (41 B/clk/CU), with
// the MFMA pipe busy 62-69 %.  One 8-wave workgroup on a 256-row tile shares the weight panel: (256 + 192) x 128 B =
// 56 KiB for 6.3 MFLOP, 30 % fewer bytes per flop -- but eight waves marching in lock step (load, barrier, multiply)
// leave the matrix pipe idle during every load phase, which is why the plain 8-wave variant measured no better.
//
// Schedule.  The waves form two groups (waves 0-3 own tile rows 0-127, waves 4-7 rows 128-255; a SIMD holds one wave of
// each).  A group alternates between a LOAD phase -- read all A/B fragments of K step k from LDS into registers (80
// VGPRs), issue its share of the LDS-DMA pieces of the following steps -- and a COMPUTE phase -- 24 back-to-back MFMAs
// (32x32x16 bf16) from registers, nothing else.  The groups run half a K step apart, one s_barrier per phase, so on
// every SIMD one wave multiplies while the other loads:
//
//     slot      0        1        2        3        4
//     group 0   LOAD 0   COMP 0   LOAD 1   COMP 1   LOAD 2  ...
//     group 1   (idle)   LOAD 0   COMP 0   LOAD 1   COMP 1  ...
//
// LDS (136 KiB, one workgroup per CU): A ring of 2 stages (each group only ever touches its own 128-row half), B ring of
// 3 stages.  Who loads what, and when it may be read (`slot` as above; a piece issued in slot t is waited for by its
// issuer -- s_waitcnt vmcnt(0) at the end of the COMPUTE phase, slot t+1 -- and published by the barrier that ends t+1):
//     group 0, LOAD k (slot 2k):    A0(k+1) -> A stage (k+1)&1     first read: group 0, slot 2k+2
//                                   B(k+1) rows 96..191 -> B stage (k+1)%3    first read: group 0, slot 2k+2
//     group 1, LOAD k (slot 2k+1):  A1(k+1) -> A stage (k+1)&1     first read: group 1, slot 2k+3
//                                   B(k+2) rows 0..95 -> B stage (k+2)%3      first read: group 0, slot 2k+4
// and what each of them overwrites was last read at least one full slot earlier (A_g(k-1): slot 2k-2+g; B(k-2): slot
// 2k-3; B(k-1): slots 2k-2 / 2k-1), by reads that were drained (lgkmcnt(0)) before the barrier ending that slot.  The
// third B stage is what lets group 1 issue half of B(k+2) while B(k) is still being read.  No wait ever sits between an
// issue and the phase that needs the data: every piece has one whole COMPUTE phase (~770 cycles) to land.
//
// Everything else -- implicit A operand (taps, virtual concat, space-to-depth, phase convs), source-side XOR swizzle,
// out-of-range offsets for padding, position-major rows with skipped padding taps, wide-store epilogue with fused
// GroupNorm statistics / PreNorm -- is shared with the 128 x 192 kernel (conv_gemm_common.h).
#include "conv_gemm_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace nope {

namespace {

constexpr int PP_BM = 256;
constexpr int PP_WAVES = 8;

// s_waitcnt simm16 on gfx950: vmcnt = [3:0] | [15:14] << 4, expcnt = [6:4], lgkmcnt = [11:8]; unused counters at their maximum
constexpr int WAIT_VMCNT0 = 0x0F70;      // vmcnt(0): every LDS-DMA piece this wave issued has landed
constexpr int WAIT_LGKMCNT0 = 0xC07F;    // lgkmcnt(0): every ds_read of this wave has returned

struct KPos { int tap, kc; };

// TUNE: the instantiation that honours the NOPE_PP_VARIANT ablations (run-time tests inside the K loop); production launches (variant 0) take the
// one without them.
template <class T, int MODE, bool PN, bool TUNE = false, bool LEAN = false>
__global__ __launch_bounds__(PP_WAVES * 64, 2) void conv_gemm_pp_kernel(ConvParams p) {
    typedef Tile<T> TL;
    constexpr int VEC = Elt<T>::VEC;
    constexpr unsigned ES = (unsigned)sizeof(T);
    constexpr int RB = 128;
    constexpr int BK = RB / (int)ES;
    constexpr int A_STAGE = PP_BM * RB, B_STAGE = BN * RB;
    constexpr int B_BASE = 2 * A_STAGE;
    constexpr int RING = 2 * A_STAGE + 3 * B_STAGE;
    constexpr int PANELS = PP_WAVES * Ep<T>::WAVE_BYTES;
    constexpr int LDS_BYTES = RING > PANELS ? RING : PANELS;
    constexpr int KS = RB / 16 / TL::STEP_SLOTS, RAW = TL::RAW, KK = KS * RAW;     // K steps per stage, raw 16-byte reads per row and step
    static_assert(MODE == NOPE_CONV_PLAIN || MODE == NOPE_CONV_DOWN2 || MODE == NOPE_CONV_UP2P, "modes of the U-Net's large launches");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];     // the ONLY LDS object (cdna_hip_programming.md, section 5 trap (a))

    const int tid = threadIdx.x;
    const int variant = TUNE ? p.variant : 0;
    if (variant & 128) { if (tid == 9999) lds[0] = 1; return; }   // tuning only (NOPE_PP_VARIANT): launch cost of the grid
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // 4 (M) x 2 (N) waves of 64 x 96
    const int grp = wave >> 2, wl = wave & 3;
    // NOPE_F16X2: A rows are staged as raw f32 (every loader below is the bf16x3 one: 4 bytes per channel on both sides) and split into the
    // tile's four operands in registers, inside the COMPUTE phase (Tile<f16x2_t>::prep_hi / prep_lo); saturating conversions as in the
    // tap-resident kernel.  (The PreNorm instantiation is never launched for it: launch_conv.)
    constexpr bool X2 = Elt<T>::DT == NOPE_F16X2;
    if constexpr (X2) fp16_ovfl_on();
    const int x2_t = (X2 && NOPE_X2_TRACK) ? p.x2_scale[3] : 0;                 // the layer's activation range shift t (nope_common.h: kX2*): operands from a * 2^-t,
    const int x2_sc = X2 ? p.x2_scale[0] : 0;                // the accumulators hold 2^-t x the convolution, the epilogue multiplies by 2^t
    const float x2_inv = x2_pow2(-x2_t), x2_out = x2_pow2(x2_t), x2_da = x2_pow2(x2_t - kX2AShift);
    float x2_amax = 0.f;                                     // max |a| over the A elements this lane converts
    int tile_m, tile_n;
    tile_coords(p, tile_m, tile_n);
    const int m0 = tile_m * PP_BM, n0 = tile_n * BN;
    const int HWo = p.Hm * p.Wm;
    const int Cin = p.C1 + p.C2;
    const int ph_y = MODE == NOPE_CONV_UP2P ? ((int)blockIdx.y >> 1) : 0, ph_x = MODE == NOPE_CONV_UP2P ? ((int)blockIdx.y & 1) : 0;

    const auto r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src1, (short)0, (int)p.bytes1, 0x00020000);
    const auto r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src2 ? p.src2 : p.src1), (short)0, (int)(p.src2 ? p.bytes2 : p.bytes1), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (size_t)blockIdx.y * p.w_phase_bytes), (short)0, (int)p.bytesw, 0x00020000);

    // ---- this lane's rows of the DMA pieces: wave w stages tile rows 32 w .. 32 w + 31 of A (= rows of its own group's
    // half) and 24 rows of B: group 1 the panel's rows 0..95, group 0 rows 96..191.  A piece = 8 rows x 128 B.
    const int rsub = lane >> 3, lslot = lane & 7;
    unsigned a_b1[4], a_b2[4], a_mask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * wave + i) + rsub;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const unsigned mm = ok ? (unsigned)m : 0u;
        unsigned b, r;
        if (MODE == NOPE_CONV_PLAIN && p.posmajor) { r = p.d_n.div(mm); b = mm - r * (unsigned)p.nhyp; }
        else { b = p.d_hw.div(mm); r = mm - b * (unsigned)HWo; }
        const int oy = (int)p.d_w.div(r), ox = (int)r - oy * p.Wm;
        const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(row)) * VEC);   // source channel chunk of this LDS slot
        const unsigned s1 = p.d_rep1.div(b), s2 = p.d_rep2.div(b);
        unsigned mask = 0;
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
            if (p.ntaps == 9) {
                const unsigned vx = (ox > 0 ? 1u : 0u) | 2u | (ox + 1 < p.Ws ? 4u : 0u);
                mask = (oy > 0 ? vx : 0u) | (vx << 3) | (oy + 1 < p.Hs ? vx << 6 : 0u);
            } else mask = 1u;
        } else {   // DOWN2: the 4 taps of a 2x2 block are always inside the image
            a_b1[i] = (((s1 * p.Hs + 2 * oy) * p.Ws + 2 * ox) * p.C1 + cs) * ES;
            a_b2[i] = 0;
            mask = 0xfu;
        }
        a_mask[i] = ok ? mask : 0u;
    }
    const int brow0 = 96 * (1 - grp) + 24 * wl;
    unsigned b_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int row = brow0 + 8 * j + rsub;
        const int n = n0 + row;
        const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(row)) * VEC);
        b_off[j] = n < p.Cout ? ((unsigned)n * p.ntaps * Cin + cs) * ES : OOB;
    }

    const int kc_per_tap = Cin / BK;
    unsigned tile_taps = (1u << p.ntaps) - 1u;
    if (MODE == NOPE_CONV_PLAIN && p.posmajor) {       // one pixel position per tile: its padding taps vanish
        const int pos = (int)p.d_n.div((unsigned)m0);
        const int ty = (int)p.d_w.div((unsigned)pos), tx = pos - ty * p.Wm;
        const unsigned vx = (tx > 0 ? 1u : 0u) | 2u | (tx + 1 < p.Ws ? 4u : 0u);
        tile_taps = (ty > 0 ? vx : 0u) | (vx << 3) | (ty + 1 < p.Hs ? vx << 6 : 0u);
    }
    const int nk = __builtin_popcount(tile_taps) * kc_per_tap;
    const int tap0 = __builtin_ctz(tile_taps);

    // K order: channel chunk outer, valid taps inner (as the 128 x 192 kernel)
    auto advance = [&](KPos& s) {
        const unsigned rest = tile_taps >> (s.tap + 1);
        if (rest) s.tap += 1 + __builtin_ctz(rest);
        else { s.tap = tap0; ++s.kc; }
    };
    // One K step of A = 4 pieces per wave, of B = 3: `prep` folds the step into scalars, `piece` issues one 1 KiB DMA.
    bool pa_first = true; unsigned pa_kadd = 0; int pa_tap = 0; unsigned char* pa_dst = nullptr;
    auto prep_a = [&](const KPos& s, unsigned char* dst) {
        const int c0 = s.kc * BK;
        pa_first = c0 < p.C1;                         // wave-uniform: a K step lies inside one source
        const int Cs = pa_first ? p.C1 : p.C2;
        unsigned kadd = (unsigned)(pa_first ? c0 : c0 - p.C1) * ES;
        if (MODE == NOPE_CONV_PLAIN) {
            if (p.ntaps == 9) {
                const int dyi = s.tap / 3, dxi = s.tap - dyi * 3;
                kadd += (unsigned)(((dyi - 1) * p.Ws + (dxi - 1)) * Cs) * ES;
            }
        } else if (MODE == NOPE_CONV_DOWN2) {
            kadd += (unsigned)(((s.tap >> 1) * p.Ws + (s.tap & 1)) * Cs) * ES;
        } else {
            kadd += (unsigned)((((s.tap >> 1) + ph_y - 1) * p.Ws + ((s.tap & 1) + ph_x - 1)) * Cs) * ES;
        }
        pa_kadd = kadd; pa_tap = s.tap; pa_dst = dst;
    };
    auto piece_a = [&](int i) {
        const unsigned base = pa_first ? a_b1[i] : a_b2[i];
        const unsigned off = (((a_mask[i] >> pa_tap) & 1u) ? base : OOB) + pa_kadd;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pa_first ? r1 : r2, (lds_void_t*)(pa_dst + i * 1024), 16, off, 0, 0, 0);
    };
    unsigned pb_kofs = 0; unsigned char* pb_dst = nullptr;
    auto prep_b = [&](const KPos& s, unsigned char* dst) { pb_kofs = (unsigned)(s.tap * Cin + s.kc * BK) * ES; pb_dst = dst; };
    auto piece_b = [&](int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(pb_dst + j * 1024), 16, b_off[j] + pb_kofs, 0, 0, 0);
    };
    auto issue_a = [&](const KPos& s, unsigned char* dst) {
        prep_a(s, dst);
#pragma unroll
        for (int i = 0; i < 4; ++i) piece_a(i);
    };
    auto issue_b = [&](const KPos& s, unsigned char* dst) {
        prep_b(s, dst);
#pragma unroll
        for (int j = 0; j < 3; ++j) piece_b(j);
    };
    unsigned char* const a_dst = lds + (4 * wave) * 1024;                    // + stage * A_STAGE
    unsigned char* const b_dst = lds + B_BASE + (brow0 >> 3) * 1024;         // + stage * B_STAGE

    typename TL::acc_t acc[TL::MT][TL::NTL];
#pragma unroll
    for (int i = 0; i < TL::MT; ++i)
#pragma unroll
        for (int j = 0; j < TL::NTL; ++j)
#pragma unroll
            for (int r = 0; r < TL::R; ++r) acc[i][j][r] = 0.f;

    // fragment addresses of raw read 0 inside a stage; raw read q flips slot bits: off ^ (raw_slot(q) << 4)
    int fa[TL::MT], fb[TL::NTL];
#pragma unroll
    for (int i = 0; i < TL::MT; ++i) {
        if constexpr (X2) fa[i] = lds_off_rb<RB>(wm * 64 + i * TL::TM + TL::frag_row(lane), TL::frag_slot_raw(lane));
        else fa[i] = lds_off_rb<RB>(wm * 64 + i * TL::TM + TL::frag_row(lane), TL::frag_slot(lane));
    }
#pragma unroll
    for (int j = 0; j < TL::NTL; ++j) fb[j] = lds_off_rb<RB>(wn * 96 + j * TL::TM + TL::frag_row(lane), TL::frag_slot(lane));
    auto a_slot = [](int kk) constexpr -> int {      // slot bits of raw A read kk
        if constexpr (X2) return TL::raw_slot_a(kk);
        else return raw_slot<T>(kk);
    };

    // ---- prologue: A_g(0), this group's half of B(0), and (group 1, which issues B one step ahead) its half of B(1)
    KPos ka{tap0, 0}, kb{tap0, 0};
    if (nk > 0) {
        issue_a(ka, a_dst); advance(ka);
        issue_b(kb, b_dst); advance(kb);
        if (grp == 1 && nk > 1) { issue_b(kb, b_dst + B_STAGE); advance(kb); }
    }
    __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {                                    // slot 0: group 0 loads, group 1 has nothing to do yet
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    int sa = 0, sb = 0;                                // A stage (k & 1) and B stage (k % 3) of the current K step
    // One K step.  FULL: a step of the steady state (k + 2 < nk) -- both DMA streams run, no barrier is skipped: the loop below runs these
    // without a single run-time test on k; the last two steps take the general form.
    auto k_step = [&](auto full_tag, int k) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        // ---- LOAD k: fragments of step k into registers, then the DMA pieces of the steps after it
        const unsigned char* la = lds + sa * A_STAGE;
        const unsigned char* lb = lds + B_BASE + sb * B_STAGE;
        u32x4 af[KS][RAW][TL::MT], bfr[KS][RAW][TL::NTL];
        const int sa1 = sa ^ 1;
        const int sb1 = sb == 2 ? 0 : sb + 1;
        const int sb2 = sb1 == 2 ? 0 : sb1 + 1;
        // (tuning: 16 = no DMA stream, 512 = A pieces only on the first tap of a channel chunk, 1024 = no B pieces: wrong results,
        //  they size what a tap-resident A stage / a cheaper B stream would buy)
        const bool do_a = (FULL || k + 1 < nk) && !(variant & 16) && !((variant & 512) && ka.tap != tap0);
        const bool do_b = (FULL || (grp == 0 ? k + 1 < nk : k + 2 < nk)) && !(variant & (16 | 1024));
        if (variant & 4) __builtin_amdgcn_s_setprio(2);      // (tuning: 4 = the LOAD phase outranks the other group's MFMA issue)
        if (do_a) prep_a(ka, a_dst + sa1 * A_STAGE);
        if (do_b) prep_b(kb, b_dst + (grp == 0 ? sb1 : sb2) * B_STAGE);
        if (variant & 1) {                           // (tuning: 1 = all fragment reads first, then all DMA pieces)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) af[kk / RAW][kk % RAW][i] = ld16(la + (fa[i] ^ (a_slot(kk) << 4)));
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j) bfr[kk / RAW][kk % RAW][j] = ld16(lb + (fb[j] ^ (raw_slot<T>(kk) << 4)));
            }
            if (do_a) {
#pragma unroll
                for (int i = 0; i < 4; ++i) piece_a(i);
            }
            if (do_b) {
#pragma unroll
                for (int j = 0; j < 3; ++j) piece_b(j);
            }
        } else {
            // The CU has ONE vector-memory path (64 B/clk): the 28 pieces a group issues per phase keep it busy for
            // ~450 cycles, the 80 fragment reads keep the LDS busy for ~320.  Issued one after the other they add up;
            // dealt out alternately (one piece, then a few reads) both units work through the whole phase, and the
            // first pieces are in flight from the start of the phase.
            constexpr int NFR = KK * (TL::MT + TL::NTL);
            auto frag = [&](int f) {      // f-th fragment read of the step, K sub-step major (static index after unrolling)
                const int kk = f / (TL::MT + TL::NTL), r = f - kk * (TL::MT + TL::NTL);
                if (r < TL::MT) af[kk / RAW][kk % RAW][r] = ld16(la + (fa[r] ^ (a_slot(kk) << 4)));
                else bfr[kk / RAW][kk % RAW][r - TL::MT] = ld16(lb + (fb[r - TL::MT] ^ (raw_slot<T>(kk) << 4)));
            };
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                if (q < 4) { if (do_a) piece_a(q); }
                else { if (do_b) piece_b(q - 4); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = q * NFR / 7; f < (q + 1) * NFR / 7; ++f) frag(f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (FULL || k + 1 < nk) advance(ka);
        if (FULL || (grp == 0 ? k + 1 < nk : k + 2 < nk)) advance(kb);
#ifndef NOPE_PP_PREP_COMPUTE
#define NOPE_PP_PREP_COMPUTE 1
#endif
        constexpr bool PREP_IN_COMPUTE = RAW > 1 && NOPE_PP_PREP_COMPUTE && !X2;
        if constexpr (RAW > 1 && !PREP_IN_COMPUTE && !X2) {   // (bf16x3: split the f32 A values into hi / lo, still in the LOAD phase)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) TL::prep_step(af[ks]);
        }
        __builtin_amdgcn_s_waitcnt(WAIT_LGKMCNT0);     // my reads of stage k are done: after the barrier the other group may overwrite it
        if (variant & 4) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- COMPUTE k: registers only
        if (!(variant & 2)) __builtin_amdgcn_s_setprio(1);       // (tuning: 2 = no priority for the MFMA phase)
        if (!(variant & 32)) {                       // (tuning: 32 = no MFMA)
            if constexpr (X2) {
                // f16x2 (KS = 1: a stage is one 32-channel step): the split of the two row tiles' raw f32 fragments -- 8 packed f16 conversions
                // for the hi parts, then per row tile 16 unpack + 16 subtract + 16 packed e4m3 conversions -- rides behind the MFMAs that do
                // not need it yet: term 0 (hi x hi, channels 0..15 of the step) waits for 4 conversions only, the cross-term MFMAs come last.
                u32x4 ax[RAW][TL::MT];
                TL::prep_hi(af[0], ax, 0, 0, x2_inv);
                __builtin_amdgcn_sched_barrier(0);
                TL::prep_hi(af[0], ax, 1, 0, x2_inv);
                TL::prep_hi(af[0], ax, 0, 1, x2_inv);
                TL::prep_hi(af[0], ax, 1, 1, x2_inv);
#pragma unroll
                for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                    for (int q = 0; q < RAW; ++q) TL::prep_lo(af[0], ax, i, q, x2_inv, x2_da, x2_amax);
#pragma unroll
                for (int t = 0; t < TL::TERMS; ++t)
#pragma unroll
                    for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                        for (int j = 0; j < TL::NTL; ++j) TL::mma(t, ax, bfr[0], i, j, acc[i][j], x2_sc);
                // 12 f16 MFMAs (32 cycles each) carry the 12 remaining hi conversions and row tile 0's lo parts, the first cross-term MFMAs
                // (64 cycles each) row tile 1's
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (PREP_IN_COMPUTE) {
                // bf16x3: the (hi, lo) split of the f32 A fragments (24 VALU per row tile) rides in THIS phase, one row tile ahead of the MFMAs
                // that consume it and dealt out three behind each of the previous row tile's nine MFMAs -- the loading group's VALU takes issue
                // slots from the other group's MFMAs, the multiplying wave's own does not (same finding as the tap-resident kernel's rewrite).
                // Row tile outer, term inner: every accumulator still sees (lo, hi), (hi, lo), (hi, hi) per K step in the same order: same bits.
                TL::prep_one(af[0], 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < KS * TL::MT; ++g) {
                    const int ks = g / TL::MT, i = g % TL::MT;
                    if (g + 1 < KS * TL::MT) TL::prep_one(af[(g + 1) / TL::MT], (g + 1) % TL::MT);
#pragma unroll
                    for (int t = 0; t < TL::TERMS; ++t)
#pragma unroll
                        for (int j = 0; j < TL::NTL; ++j) TL::mma(t, af[ks], bfr[ks], i, j, acc[i][j]);
                    if (g + 1 < KS * TL::MT) {
#pragma unroll
                        for (int q = 0; q < TL::TERMS * TL::NTL; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < TL::TERMS; ++t)
#pragma unroll
                    for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                        for (int j = 0; j < TL::NTL; ++j) TL::mma(t, af[ks], bfr[ks], i, j, acc[i][j]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {          // keep the fragment reads alive
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) NOPE_KEEP_VGPR(af[kk / RAW][kk % RAW][i]);
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j) NOPE_KEEP_VGPR(bfr[kk / RAW][kk % RAW][j]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (!(variant & 8))                          // (tuning: 8 = never wait for the DMA -- wrong results, shows the issue-bound time)
        __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);       // the pieces I issued in LOAD k have landed (they had this whole phase)
        if (FULL || !(grp == 1 && k == nk - 1)) {      // (group 1 started one barrier late: it skips the last one)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        sa = sa1; sb = sb1;
    };
    {
        int k = 0;
        for (; k + 2 < nk; ++k) k_step(std::true_type{}, k);
        for (; k < nk; ++k) k_step(std::false_type{}, k);
    }
    // Both groups have passed 2 nk + 1 barriers.  Group 0 arrives here while group 1 still multiplies (registers only);
    // every LDS stage read and every DMA is complete, so the per-wave epilogue panels may reuse the ring.
    if (variant & 64) {                              // tuning only: no epilogue (keeps the accumulators live)
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.out)[0] = 1.f;
        return;
    }
    if constexpr (X2 && NOPE_X2_KERNEL_AMAX) x2_publish_amax(p, x2_amax, lane);
    epilogue_wide<T, PN, false, NoStamp, false, LEAN ? 1 : 0>(p, acc, m0, n0, wm, wn, lane, lds + wave * Ep<T>::WAVE_BYTES, NoStamp(), x2_out);
}

// ---- 3x3 convolutions: the A operand stays in LDS across the 9 taps ------------------------------------------------
// With GEMM rows in (sample, pixel) order, tap (dy, dx) of tile row i is the pixel dy*W + dx rows further along the
// same flat pixel axis.  So instead of staging 256 rows per tap (9 x 32 KiB per channel chunk), a stage holds the tile's
// pixel range extended by W + 1 rows on either side (<= 328 rows, 41 KiB, loaded ONCE per channel chunk) and every tap
// reads its fragments at a row offset.  A tap that leaves the image (zero padding) or wraps into a neighbouring image
// row / sample is redirected, per lane, to a 128-byte row of zeros kept behind each stage -- decided once, when the
// per-tap fragment addresses are precomputed; nothing touches the data.
// Per channel chunk the L2 -> LDS stream is 41 KiB of A + 9 x 24 KiB of B instead of 9 x 56 KiB: half the bytes, and
// 4 instead of 7 DMA pieces per wave per K step.  Same K order (channel chunk outer, tap inner, padding taps add exact
// zeros) and the same accumulation chain as the other conv kernels: results are bit-identical to theirs.
//
// Schedule, B ring, barriers and waits are those of conv_gemm_pp_kernel above; the A ring has two stages of whole
// chunks: the (<= 6) pieces a wave owns of chunk c+1 are issued one per K step during taps 0..5 of chunk c, into the
// stage chunk c-1 was read from (last read: K step 9c-1, one barrier-separated slot before the first such issue), and
// have landed -- issuer's vmcnt(0) + barrier -- at least three K steps before chunk c+1 begins.
//
// The LOAD phase is kept SHORT IN INSTRUCTIONS.  Cycle counters on the first version of this kernel (profiles/
// r02b_pmc_halo_L3.txt) showed the LDS ~30 % busy and the MFMA pipe 61 % busy: a LOAD phase of ~180 instructions
// (per-step tap arithmetic, swizzle and mask VALU, scalar bookkeeping) simply takes longer to ISSUE (~5 cycles each)
// than the 768 cycles the other group's 24 MFMAs cover.  So the nine taps are unrolled (two chunks = 18 steps of
// straight-line code, the stage parities become immediates), the 9 x MT fragment addresses live in registers, B
// fragment addresses are one register per K sub-step plus instruction immediates (the B ring sits at LDS offset 0 so
// they fit 16 bits), and the K offsets of the DMA pieces travel in the scalar offset operand: ~20 ds_read + 4 DMA +
// a few dozen scalar/vector instructions per step.
constexpr int HALO_ROWS = 328;       // 256 + 2 * (32 + 1), rounded up to whole 8-row pieces: maps up to 32 pixels wide
constexpr int HALO_MAX_W = 32;

constexpr int TIMELINE_STAMPS = 720;  // per group; 5 per K step (tuning instantiation only)

// SPLIT: the split-K instantiation (chunk range from blockIdx.z, raw f32 partials out) -- its own kernel so that the main one keeps
// its register allocation (256 VGPRs, no spill: one more live scalar pair spilled it).
// SHIFT (NOPE_F16X2): false = the instantiation for layers whose activation range shift t is 0 (the host knows: ConvParams::x2_t_zero) -- the rewrite
// then works on a itself and loses the two packed multiplies per piece that a * 2^-t costs (+1.7 % on the kernel, same-box A/B,
// profiles/r06c_*); every layer starts at t = 0 and stays there while its inputs peak inside [1, 1024].
// UP (NOPE_CONV_UP2P: nearest x 2 + 3 x 3 as four 2 x 2 phase convs on the low-resolution map, blockIdx.y = phase; model_utils.py:161-165): the
// tile's pixel range + halo is the 3 x 3 neighbourhood every phase draws its four taps from, so the stage is loaded (and, f32 storage, rewritten)
// ONCE per channel chunk and read by the phase's four taps at row offsets -- where the per-tap kernel stages 256 rows per tap and splits them in
// registers at every read (its launches are bound by that LOAD phase: 0.32-0.37 of the pipe against 0.54 here).  A chunk is FOUR K steps, each
// the 3 x 3 kernel's (fragment reads, 18-24 MFMAs from registers), and everything the next chunk needs rides in them:
//   * A stage of the next chunk (<= 5 pieces per wave: maps up to 30 pixels wide, checked by the launcher): pieces {0, 4} / {1} / {2, 3} are
//     issued at taps 0 / 1 / 2 and rewritten (f32 storage) behind the MFMAs of the NEXT tap -- two pieces in one COMPUTE phase at taps 1 and 3:
//     ~44 VALU + 8 LDS instructions behind 15 MFMAs.  Piece 4 is rewritten by every wave (not every wave loads one: its rows exist in every
//     stage and are read only where a wave loaded them), so the phase stays one basic block.
//   * Weight ring: four taps on three stages cannot rotate with immediates (4 % 3 != 0), so the stage of tap t is fixed -- 1, 2, 0, 2 -- and
//     the ring's discipline (group 0 issues its half one tap ahead, group 1 two taps ahead) is bent where two taps share a stage: taps 1 and 3
//     share stage 2, which group 1 is still reading when it would issue its half of the other, so group 0 issues BOTH halves of taps 1 and 3
//     (at taps 0 and 2); tap 0 of the next chunk has its own stage (group 1 at tap 2, group 0 at tap 3).  Every piece has a whole COMPUTE
//     phase to land, as in the 3 x 3 kernel.
// Earlier forms, both bit-identical and slower: six positions per chunk with pieces issued in two MFMA-less ones (two exposed L2 round trips
// per chunk: 0.63 of the K steps' rate, profiles/r06v_*), five positions with one light position (0.78, r06w_*).  Both groups pass two
// barriers per K step.  Same K order as the per-tap kernel (chunk outer, tap inner): bit-identical to it.
template <class T, bool TIMELINE = false, bool SPLIT = false, bool SHIFT = true, bool LEAN = false, bool UP = false>
__global__ __launch_bounds__(PP_WAVES * 64, 2) void conv3x3_halo_kernel(ConvParams p) {
    static_assert(!UP || (!TIMELINE && !SPLIT && !LEAN), "the phase-conv form: one tile per workgroup, generic wide epilogue");
    typedef Tile<T> TL;
    constexpr int VEC = Elt<T>::VEC;
    constexpr unsigned ES = (unsigned)sizeof(T);
    constexpr int RB = 128;
    constexpr int BK = RB / (int)ES;
    constexpr int B_STAGE = BN * RB;
    constexpr int A_BASE = 3 * B_STAGE;                            // B ring first: its addresses fit instruction immediates
    constexpr int ZROW = HALO_ROWS * RB;                           // 128 bytes of zeros behind the rows of a stage
    constexpr int A_STAGE = ZROW + RB;
    constexpr int RING = A_BASE + 2 * A_STAGE;
    constexpr int PANELS = PP_WAVES * Ep<T>::WAVE_BYTES;
    constexpr int LDS_USED = RING > PANELS ? RING : PANELS;
    constexpr int LDS_BYTES = LDS_USED + (TIMELINE ? 2 * TIMELINE_STAMPS * 4 : 0);
    constexpr int KS = RB / 16 / TL::STEP_SLOTS, RAW = TL::RAW, KK = KS * RAW;
    static_assert(LDS_BYTES <= 160 * 1024 && A_STAGE < 65536 && 2 * B_STAGE + BN * RB < 65536 + B_STAGE && 2 * B_STAGE + 2 * Tile<T>::TM * RB < 65536, "LDS budget / ds_read immediates (16 bits)");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    if (p.variant & 128) { if (tid == 9999) lds[0] = 1; return; }   // tuning only (NOPE_PP_VARIANT): launch cost of the grid
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = wave >> 2, wl = wave & 3;
    // (tuning instantiation: the first wave of each group of workgroup 0 stamps the shader clock at five points of every K
    //  step -- LOAD reads done / barrier passed / MFMAs issued / DMA landed / barrier passed -- into the LDS tail)
    int n_stamp = 0;
    auto stamp = [&]() {
        if constexpr (TIMELINE) {
            if (blockIdx.x == 0 && wl == 0 && n_stamp < TIMELINE_STAMPS - 4) {
                const unsigned t = (unsigned)__builtin_readcyclecounter();
                if (lane == 0) reinterpret_cast<unsigned*>(lds + LDS_USED)[grp * TIMELINE_STAMPS + n_stamp] = t;
                ++n_stamp;
            }
        }
    };
    // (... and the shader clock next to the constant 100 MHz clock at both ends of the kernel: the actual clock rate)
    auto stamp_clocks = [&](int slot) {
        if constexpr (TIMELINE) {
            if (blockIdx.x == 0 && wave == 0) {
                const unsigned t = (unsigned)__builtin_readcyclecounter(), rt = (unsigned)__builtin_amdgcn_s_memrealtime();
                if (lane == 0) {
                    reinterpret_cast<unsigned*>(lds + LDS_USED)[TIMELINE_STAMPS - 4 + 2 * slot] = t;
                    reinterpret_cast<unsigned*>(lds + LDS_USED)[TIMELINE_STAMPS - 3 + 2 * slot] = rt;
                }
            }
        }
    };
    stamp_clocks(0);
    int tile_m, tile_n;
    tile_coords(p, tile_m, tile_n);
    int m0 = tile_m * PP_BM;                                       // (advances when the workgroup walks several tiles, see below)
    const int n0 = tile_n * BN;
    const int W = p.Ws, HW = p.Hs * p.Ws;
    const int halo = W + 1;
    const int Cin = p.C1 + p.C2;
    const int npieces = (PP_BM + 2 * halo + 7) >> 3;               // 8-row pieces of a stage (33 .. 41)
    const int ph_y = UP ? ((int)blockIdx.y >> 1) : 0, ph_x = UP ? ((int)blockIdx.y & 1) : 0;      // UP: the output phase of this workgroup
    constexpr unsigned NTAPS = UP ? 4u : 9u;

    const auto r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src1, (short)0, (int)p.bytes1, 0x00020000);
    const auto r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src2 ? p.src2 : p.src1), (short)0, (int)(p.src2 ? p.bytes2 : p.bytes1), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (UP ? (size_t)blockIdx.y * p.w_phase_bytes : (size_t)0)), (short)0, (int)p.bytesw, 0x00020000);

    if (tid < 16) st16(lds + A_BASE + (tid >> 3) * A_STAGE + ZROW + (tid & 7) * 16, u32x4{0u, 0u, 0u, 0u});

    // ---- A pieces of this wave: stage rows 8 q .. 8 q + 7 for q = wave, wave + 8, ... (<= 6 of them); stage row r holds
    // flat pixel m0 - halo + r of the (hypothesis, y, x) axis.  The first source is never broadcast here (rep1 == 1, checked by
    // the launcher), so its byte offset is LINEAR in the pixel index: piece i is piece 0 plus i * 64 pixels (a scalar), the
    // swizzled channel chunk is the same for all of a lane's pieces (rows 64 apart), and pixels before / behind the tensor
    // (first and last tile) give offsets that wrap above / run past num_records: the buffer range check returns zeros.
    const int rsub = lane >> 3, lslot = lane & 7;
    const int r0 = 8 * wave + rsub;
    const unsigned a_cs = (unsigned)((lslot ^ swz_of<RB>(r0)) * VEC);
    unsigned a_base1 = (unsigned)((m0 - halo + r0) * p.C1 + (int)a_cs) * ES;
    const unsigned a_step1 = 64u * (unsigned)p.C1 * ES;
    const bool a_has4 = wave + 32 < npieces, a_has5 = wave + 40 < npieces;      // (pieces 0..3 of a wave always exist)
    unsigned char* const a_dst = lds + A_BASE + wave * 1024;                    // + stage * A_STAGE + i * 8192
    // piece i of this wave for the chunk described by (first, a_soff).  The channel offset rides in the scalar operand (it stays
    // inside the pixel); the piece stride must be part of the VECTOR offset -- the hardware range check covers only that.
    bool a_first = true; unsigned a_soff = 0;
    // A broadcast second source (rep2 > 1: the U-Net's final block concatenates the per-reference skip `r` behind the
    // per-hypothesis activations) is not linear in the pixel index: sample b reads sample b / rep2.  Its six piece offsets are
    // kept per lane instead (rows before the tensor or behind it: out of range -> zeros); without the broadcast they are the
    // linear offsets.
    const bool a_rep2 = p.C2 > 0 && p.rep2 > 1;
    unsigned a_off2[6];
#define NOPE_HALO_SET_OFF2()                                                                                                     \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                           \
            const int m = m0 - halo + r0 + 64 * i;                                                                               \
            unsigned o = (unsigned)(m * p.C2 + (int)a_cs) * ES;                                                                  \
            if (a_rep2) {                                                                                                        \
                const unsigned b = p.d_hw.div((unsigned)(m < 0 ? 0 : m)), r = (unsigned)m - b * (unsigned)HW;                    \
                o = (m >= 0 && m < p.M) ? ((p.d_rep2.div(b) * (unsigned)HW + r) * (unsigned)p.C2 + a_cs) * ES : OOB;             \
            }                                                                                                                    \
            a_off2[i] = o;                                                                                                       \
        }                                                                                                                        \
    } while (0)
    NOPE_HALO_SET_OFF2();
    auto piece_a = [&](int i, int stage) __attribute__((always_inline)) {
        if (a_first) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_void_t*)(a_dst + stage * A_STAGE + i * 8192), 16, a_base1 + i * a_step1, a_soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_void_t*)(a_dst + stage * A_STAGE + i * 8192), 16, a_off2[i], a_soff, 0, 0);
    };
    auto set_chunk = [&](int chunk) {
        const int c0 = chunk * BK;
        a_first = c0 < p.C1;                                       // wave-uniform: a chunk lies inside one source
        a_soff = (unsigned)(a_first ? c0 : c0 - p.C1) * ES;
    };
    // ---- NOPE_BF16X3: the A stage is split into (hi, lo) bf16 IN LDS, once per element.  A resident stage is read by nine taps and
    // two waves per row: splitting at the fragment reads (Tile<f32s_t>::prep_step, what the other conv kernels do) costs 96 VALU
    // per wave and K step, and every VALU instruction of the loading group takes an issue slot from the other group's MFMAs
    // (measured: 3.3 x the bf16 kernel's time for 3 x its MFMAs).  Instead the wave that issued a piece rewrites it after it
    // has landed (its own vmcnt(0) at the end of the COMPUTE phase): lane l holds 4 channels (its 16-byte slot); with the
    // neighbouring slot they form one group of 8 channels whose two slots become [hi x 8 | lo x 8] -- the layout of the packed
    // weights, so the fragment reads need no conversion at all.  A lane writes its 4 hi values into half of the group's first
    // logical slot and its 4 lo values into half of the second (two ds_write_b64); the wave's single ds_read_b128 has returned
    // for all lanes before either write is issued (same wave, in-order LDS queue + data dependence), other waves read the piece
    // only behind the barrier that ends this LOAD phase.  Zero rows and out-of-range rows are zeros either way.
    constexpr bool A_SPLIT_LDS = TL::RAW > 1;
    constexpr bool X2 = Elt<T>::DT == NOPE_F16X2;
    const int cv_half = (lslot ^ swz_of<RB>(r0)) & 1;                         // which half of its 8-channel group this lane's slot holds
    // ---- NOPE_F16X2: same in-place rewrite, into the layout of Tile<f16x2_t>.  A lane's 16-byte slot holds channels 4 ls .. 4 ls + 3 of the
    // chunk (ls = its LOGICAL slot); it writes their four f16 hi parts into half of logical slot ls / 2 (8 bytes), the four e4m3 bytes of
    // a_lo * 2^9 into slot 4 + h and the four of a * 2^-2 into slot 6 + h, h = (ls / 2) % 2 its channel set, each at byte 8 (ls / 4) + 4 (ls % 2)
    // (two ds_write_b32).
    // The f16 part saturates at +-65504, the fp8 parts at +-448: the wave sets MODE.FP16_OVFL, under which the three conversions saturate by
    // themselves (probe fact 7; a NaN stays a NaN, as in the f32 / bf16x3 modes) -- 16 VALU per piece and lane instead of 46 with explicit
    // pre-scale multiplies, clamps and byte packing.
    if constexpr (X2) fp16_ovfl_on();
    const int x2_t = (X2 && NOPE_X2_TRACK && SHIFT) ? p.x2_scale[3] : 0;                         // the layer's activation range shift t (nope_common.h: kX2*): the rewrite works on a * 2^-t,
    const int x2_sc = X2 ? p.x2_scale[0] : 0;                                 // E8M0 block scale of the cross-term MFMA (uniform; waited for with the prologue's DMA)
    const float x2_inv = x2_pow2(-x2_t), x2_out = x2_pow2(x2_t), x2_da = x2_pow2(x2_t - kX2AShift);      // the accumulators hold 2^-t x the convolution, the epilogue multiplies by 2^t
    float x2_amax = 0.f;                                                      // max |a| over the A elements this lane rewrites
    // (two halves.  bf16x3: the READ of a piece opens the LOAD phase, the DMA pieces of the phase are issued and the tap's fragment addresses
    //  formed underneath it, then the arithmetic + writes: -1.6 % on the kernel against read + rewrite back to back.  f16x2: back to back, the
    //  same order measured +1.3 % there -- same-box A/B, profiles/r05h_rewrite_order_ab.txt.  Carrying the value across the barrier from the
    //  previous COMPUTE phase spills: 256 VGPRs)
    constexpr bool CV_UNDER_ISSUE = A_SPLIT_LDS && !X2;
    // f16x2: the rewrite rides in the COMPUTE phase instead -- between the MFMAs of the wave that multiplies, whose VALU and LDS ports are
    // idle while the matrix pipe works (a wave issues in order: the rewrite's instructions are dealt out two to four behind each MFMA of the
    // second term, sched_group_barrier, so none of them waits behind an MFMA that waits for the pipe).  The piece is the same one, half a
    // step later (its stage is first read three or more steps from now).  NOPE_X2_CV_COMPUTE (compile time) = 0 restores the LOAD-phase form.
#ifndef NOPE_X2_CV_COMPUTE
#define NOPE_X2_CV_COMPUTE 1
#endif
    constexpr bool CV_IN_COMPUTE = A_SPLIT_LDS && NOPE_X2_CV_COMPUTE;
    // ... and so do the fragment addresses of the NEXT step (row + tap offset, swizzle, zero-row redirect: 12 VALU), carried across the barrier
    // in two registers that are free at that point (the step's own 80 fragment registers are dead once its MFMAs are issued).
    // (the 16-bit instantiations would take the same address trick -- 12 VALU out of their LOAD phase -- but the f16 one sits at 256 VGPRs and
    //  spills two registers with it: NOPE_FA_AHEAD_16, off)
#ifndef NOPE_FA_AHEAD_16
#define NOPE_FA_AHEAD_16 0
#endif
    constexpr bool FA_AHEAD = CV_IN_COMPUTE || (NOPE_FA_AHEAD_16 && TL::TM == 32 && sizeof(T) == 2);
    auto convert_load = [&](int i, int stage) __attribute__((always_inline)) -> u32x4 {
        return ld16(a_dst + stage * A_STAGE + i * 8192 + rsub * RB + lslot * 16);
    };
    // `track`: the piece holds fresh activations (the COMPUTE-phase rewrite of a tile's LAST chunk re-converts an already rewritten stage that
    // nothing reads again: its bit patterns are not activations and must not reach the range word)
    auto convert_store = [&](const u32x4& v, int i, int stage, bool track = true) __attribute__((always_inline)) {
        if constexpr (X2) {
            unsigned char* row = a_dst + stage * A_STAGE + i * 8192 + rsub * RB;
            const int sw = swz_of<RB>(r0), ls = lslot ^ sw;
            unsigned hi[2], lo8, a8;
            float x[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const unsigned u = v[e]; x[e] = __builtin_bit_cast(float, u); }
            if (NOPE_X2_KERNEL_AMAX) { const float m = amax4(x2_amax, x[0], x[1], x[2], x[3]); x2_amax = track ? m : x2_amax; }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x2_t t2 = f32x2_t{x[2 * e], x[2 * e + 1]} * x2_inv;         // a' = a * 2^-t (exact; the layer's range shift, nope_common.h: kX2*)
                const unsigned h = NOPE_CVT_PK_F16_OVFL(t2.x, t2.y);                   // (saturating: the wave runs with MODE.FP16_OVFL = 1)
                hi[e] = h;
                union { unsigned u; f16_t f[2]; } hh; hh.u = h;
                l[2 * e] = __builtin_fmaf(x[2 * e], x2_inv, -(float)hh.f[0]);          // a' - hi in one instruction (the product is exact)
                l[2 * e + 1] = __builtin_fmaf(x[2 * e + 1], x2_inv, -(float)hh.f[1]);
            }
            lo8 = cvt4_e4m3_scaled<kX2ALoShift, true>(l[0], l[1], l[2], l[3]);
            a8 = cvt4_e4m3_div(x[0], x[1], x[2], x[3], x2_da);                        // e4m3(a' * 2^-2) = e4m3(a / 2^(2 + t))
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            __builtin_amdgcn_wave_barrier();       // every lane's read precedes every lane's write (see below)
            *reinterpret_cast<u32x2*>(row + (((ls >> 1) ^ sw) << 4) + 8 * (ls & 1)) = u32x2{hi[0], hi[1]};
            // (channels 4 ls .. 4 ls + 3 belong to channel set (ls >> 1) & 1, at bytes 8 (ls >> 2) + 4 (ls & 1) of its two e4m3 slots)
            *reinterpret_cast<unsigned*>(row + (((4 + ((ls >> 1) & 1)) ^ sw) << 4) + 8 * (ls >> 2) + 4 * (ls & 1)) = lo8;
            *reinterpret_cast<unsigned*>(row + (((6 + ((ls >> 1) & 1)) ^ sw) << 4) + 8 * (ls >> 2) + 4 * (ls & 1)) = a8;
        } else if constexpr (A_SPLIT_LDS) {
            unsigned char* row = a_dst + stage * A_STAGE + i * 8192 + rsub * RB;
            unsigned hi[2], lo[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned u0 = v[2 * e], u1 = v[2 * e + 1];
                const float x0 = __builtin_bit_cast(float, u0), x1 = __builtin_bit_cast(float, u1);
                const unsigned h = cvt_pk_bf16(x0, x1);
                hi[e] = h;
                lo[e] = cvt_pk_bf16(x0 - __builtin_bit_cast(float, h << 16), x1 - __builtin_bit_cast(float, h & 0xffff0000u));
            }
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            // (every lane's read precedes every lane's write: one wave instruction each on the hardware; the wave barrier pins the
            //  compiler's order and is the rendezvous point of tests/hipemu, whose lanes run one after the other)
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<u32x2*>(row + ((lslot ^ cv_half) << 4) + 8 * cv_half) = u32x2{hi[0], hi[1]};
            *reinterpret_cast<u32x2*>(row + ((lslot ^ cv_half ^ 1) << 4) + 8 * cv_half) = u32x2{lo[0], lo[1]};
        }
    };
    auto convert_piece = [&](int i, int stage) __attribute__((always_inline)) { convert_store(convert_load(i, stage), i, stage); };
    // ---- B pieces: 24 rows per wave (group 1: panel rows 0..95, group 0: rows 96..191), as in conv_gemm_pp_kernel
    const int brow0 = 96 * (1 - grp) + 24 * wl;
    unsigned b_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int row = brow0 + 8 * j + rsub;
        const int n = n0 + row;
        const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(row)) * VEC);
        b_off[j] = n < p.Cout ? ((unsigned)n * NTAPS * Cin + cs) * ES : OOB;
    }
    unsigned char* const b_dst = lds + (brow0 >> 3) * 1024;                     // + stage * B_STAGE + j * 1024
    // (UP: group 0 also issues group 1's half of one weight step per chunk -- rows 96 grp + 24 wl ..)
    const int brow0_o = 96 * grp + 24 * wl;
    unsigned b_off_o[3] = {OOB, OOB, OOB};
    if constexpr (UP) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int row = brow0_o + 8 * j + rsub;
            const int n = n0 + row;
            const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(row)) * VEC);
            b_off_o[j] = n < p.Cout ? ((unsigned)n * NTAPS * Cin + cs) * ES : OOB;
        }
    }
    unsigned char* const b_dst_o = lds + (brow0_o >> 3) * 1024;
    const unsigned cin_es = (unsigned)Cin * ES;
    unsigned bkofs = 0;                                            // K offset (tap * Cin + chunk * BK) * ES of the next B step this wave issues
    auto issue_b = [&](int stage) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(b_dst + stage * B_STAGE + j * 1024), 16, b_off[j], bkofs, 0, 0);
    };

    // ---- fragment rows of this lane: stage row of the centre tap, and which of the 9 taps stay inside its image.  The
    // per-tap addresses are formed in the loop (row + scalar tap offset, swizzle, zero-row redirect: ~10 VALU per
    // fragment row and step) -- a table of all 9 x MT of them does not fit the register file next to 96 accumulators and
    // 80 fragment registers, and a spilled table is reloaded through vmcnt, which would drain the DMA queue.
    unsigned f_mask[TL::MT];
    const int fslot = TL::frag_slot(lane);
#pragma unroll
    for (int i = 0; i < TL::MT; ++i) {
        const int il = wm * 64 + i * TL::TM + TL::frag_row(lane);
        const unsigned m = (unsigned)(m0 + il);
        const unsigned b = p.d_hw.div(m), r = m - b * (unsigned)HW;
        const int oy = (int)p.d_w.div(r), ox = (int)r - oy * W;
        const unsigned vx = (ox > 0 ? 1u : 0u) | 2u | (ox + 1 < W ? 4u : 0u);
        f_mask[i] = (oy > 0 ? vx : 0u) | (vx << 3) | (oy + 1 < p.Hs ? vx << 6 : 0u);
    }
    const int f_row0 = wm * 64 + TL::frag_row(lane) + halo;       // fragment row i sits i * TM stage rows further
    // B: the swizzle of rows wn*96 + j*TM + l does not depend on j (48 wn and j TM / 2 are multiples of 8): one address per
    // K sub-step, tile j and ring stage are immediates
    int fbk[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) fbk[kk] = lds_off_rb<RB>(wn * 96 + TL::frag_row(lane), fslot) ^ (raw_slot<T>(kk) << 4);

    typename TL::acc_t acc[TL::MT][TL::NTL];
#pragma unroll
    for (int i = 0; i < TL::MT; ++i)
#pragma unroll
        for (int j = 0; j < TL::NTL; ++j)
#pragma unroll
            for (int r = 0; r < TL::R; ++r) acc[i][j][r] = 0.f;

    // Split-K (p.splits > 1; few tiles, long K -- the 4 x 4 / 8 x 8 levels at a few dozen hypotheses): blockIdx.z owns the channel
    // chunks [c_lo, nchunks) of the tile and writes raw f32 partial sums; splitk_reduce_kernel adds them in a fixed order.
    int c_lo = 0, nchunks = Cin / BK;
    if constexpr (SPLIT) {
        const int tot = nchunks, z = (int)blockIdx.z;
        c_lo = (int)((long long)z * tot / p.splits);
        nchunks = (int)((long long)(z + 1) * tot / p.splits);
    }
    const unsigned wrap_inc = (unsigned)BK * ES - (NTAPS - 1u) * cin_es;     // K offset step from the last tap of a chunk to tap 0 of the next

    // ---- prologue of a tile: the whole A stage of chunk 0, this group's half of B(0), and (group 1) its half of B(1)
    auto tile_prologue = [&]() __attribute__((always_inline)) {
        set_chunk(c_lo);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i < 4 || (i == 4 ? a_has4 : a_has5)) piece_a(i, 0);
        bkofs = (unsigned)(c_lo * BK) * ES;
        if constexpr (UP) {                                        // tap 0 -> ring stage 1 (both groups their half)
            issue_b(1);                                            // (tap 1: both halves by group 0 at tap 0)
        } else {
            issue_b(0); bkofs += cin_es;
            if (grp == 1) { issue_b(1); bkofs += cin_es; }         // (nk >= 9 > 2)
        }
    };
    // ---- the epilogue panels of a wave.  bf16: outside everything the NEXT tile's prologue writes (A stage 0, B stage 0
    // and the first half of B stage 1), so a workgroup that walks several tiles can have that prologue in flight while it
    // stores this tile: group 0 in the second half of B stage 1 + B stage 2, group 1 in A stage 1 (below its zero row).
    unsigned char* lds_panel = lds + wave * Ep<T>::WAVE_BYTES;
    if constexpr (TL::TM == 32) {
        static_assert(TL::TM != 32 || (4 * Ep<T>::WAVE_BYTES <= B_STAGE + B_STAGE / 2 && 4 * Ep<T>::WAVE_BYTES <= ZROW), "panel placement");
        lds_panel = lds + (grp == 0 ? B_STAGE + B_STAGE / 2 : A_BASE + A_STAGE) + wl * Ep<T>::WAVE_BYTES;
    }
    // A workgroup walks `iters` tiles of the same weight panel, gridDim.x / 8 M tiles apart (a multiple of the image size:
    // checked by the launcher, so the padding masks above hold for every tile of the walk).
    const int iters = !UP && TL::TM == 32 && p.persist_iters > 1 ? p.persist_iters : 1;
    const int walk_rows = (int)(gridDim.x >> 3) / (p.xcd_map == 2 ? p.tiles_n / p.xcd_gn : 1) * PP_BM;
    tile_prologue();
    // Tuning only (NOPE_PP_VARIANT & 2048): touch the workgroup's whole weight stream up front -- one 4-byte load per 128-byte weight row
    // segment, PFN per thread, after the prologue's DMA pieces (older: they land first) -- to pull it towards this XCD's L2 while the
    // first K steps run.  Measured and NOT adopted (profiles/r04e_conv_split_bench.txt): a split-K launch at 64 hypotheses takes 1.2-1.6 us
    // per K step against 0.6 in the launches that fill the chip because every B piece misses L2 (each weight byte is used by four
    // workgroups at once and never again) with only ~1.5 steps of look-ahead in the 3-stage B ring; but an XCD's share of the
    // weights (5.3 MB of 42.5) does not fit its 4 MB L2 next to the partials, so the prefetched lines are gone before their step:
    // 63.6 us with, 57.8 us without (warm or cold weights alike: the Infinity Cache holds them either way).
    constexpr int PFN = 16;
    unsigned pfv[SPLIT ? PFN : 1];
    bool prefetched = false;
    if constexpr (SPLIT) {
        prefetched = (p.variant & 2048) != 0;
        if (prefetched) {
            const int nlines = 9 * (nchunks - c_lo) * BN;
#pragma unroll
            for (int k = 0; k < PFN; ++k) {
                int li = tid + k * (PP_WAVES * 64);
                li = li < nlines ? li : nlines - 1;
                const int st = li / BN, row = li - st * BN;
                const int ch = c_lo + st / 9, tap = st - (st / 9) * 9;
                const int n = n0 + row < p.Cout ? n0 + row : p.Cout - 1;
                pfv[k] = *reinterpret_cast<const unsigned*>(p.w + ((size_t)n * 9u * Cin + (size_t)tap * Cin + (size_t)ch * BK) * ES);
            }
            NOPE_WAIT_VMCNT_KEEP_LOADS(PFN);           // vmcnt(PFN): the prologue's DMA pieces have landed
        } else __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
    } else
    __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);                       // DMA landed
    const bool dma_on = !(p.variant & 16);                         // (tuning: 16 = no DMA stream)

    int fa_next[TL::MT] = {};                                      // FA_AHEAD: fragment addresses of the step about to start (formed during the previous COMPUTE phase)
    // The nine K steps of channel chunk `chunk`, whose A stage is `par` (a literal at both call sites: after inlining and
    // unrolling every tap, stage and ring index below is an immediate).
    auto chunk_steps = [&](const int chunk, const int par) __attribute__((always_inline)) {
        const bool last = chunk + 1 == nchunks;
        if (!last) set_chunk(chunk + 1);
        // The per-tap fragment addresses below are loop invariants: hoisted out of the chunk loop they would need 9 x MT
        // registers the kernel does not have (and a spilled table is reloaded through vmcnt, draining the DMA queue).
        int frow = f_row0;
        NOPE_OPAQUE_VGPR(frow);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // (bf16x3: the piece this wave issued in the previous step has landed -- its vmcnt(0) closed that step -- split it in place,
            //  first thing in the phase, while no fragment register is live; the stage is first read three or more steps from
            //  now, behind this step's barrier.  The LOAD phase has the slack: it is the shorter one with 36 MFMAs per step.)
            const bool cv_now = A_SPLIT_LDS && dma_on && tap >= 1 && tap <= 6 && !last && (tap - 1 < 4 || (tap - 1 == 4 ? a_has4 : a_has5));
            u32x4 cv = {0u, 0u, 0u, 0u};
            if (cv_now && !(CV_IN_COMPUTE && tap <= 4)) {      // (f16x2: pieces 0..3 are rewritten inside the COMPUTE phase below; 4 and 5, which not every wave has, here)
                if constexpr (CV_UNDER_ISSUE) cv = convert_load(tap - 1, par ^ 1);
                else convert_piece(tap - 1, par ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- LOAD: DMA pieces first (they fly for the rest of this phase and the whole next one), then the fragments
            if (dma_on) {
                if (tap < 6 && !last && (tap < 4 || (tap == 4 ? a_has4 : a_has5))) piece_a(tap, par ^ 1);
                // group 0 issues the B half of step k+1 into ring stage (tap+1)%3, group 1 of step k+2 into (tap+2)%3
                const bool b_more = !last || (grp == 0 ? tap < 8 : tap < 7);
                if (b_more) {
                    if (grp == 0) issue_b((tap + 1) % 3); else issue_b((tap + 2) % 3);
                    bkofs += ((grp == 0 ? tap + 1 : tap + 2) % 9 == 8) ? wrap_inc : cin_es;
                }
            }
            u32x4 af[KS][RAW][TL::MT], bfr[KS][RAW][TL::NTL];
            int fa[TL::MT];
            auto tap_addresses = [&](int tp, int (&dst)[TL::MT]) __attribute__((always_inline)) {
                const int toff = (tp / 3 - 1) * W + (tp % 3 - 1);          // scalar: the tap's row offset along the flat pixel axis
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) {
                    const int rr = frow + i * TL::TM + toff;
                    const int off = A_BASE + (rr << 7) + (((fslot ^ (rr >> 1)) & 7) << 4);
                    dst[i] = ((f_mask[i] >> tp) & 1u) ? off : A_BASE + ZROW;   // (absolute, stage 0)
                }
            };
            if constexpr (FA_AHEAD) {
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) fa[i] = fa_next[i];
            } else tap_addresses(tap, fa);
            if (CV_UNDER_ISSUE && cv_now && !(CV_IN_COMPUTE && tap <= 4)) {
                __builtin_amdgcn_sched_barrier(0);
                convert_store(cv, tap - 1, par ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) af[kk / RAW][kk % RAW][i] = ld16(lds + (fa[i] ^ (raw_slot<T>(kk) << 4)) + par * A_STAGE);
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j) bfr[kk / RAW][kk % RAW][j] = ld16(lds + fbk[kk] + ((tap % 3) * B_STAGE + j * TL::TM * RB));
            }
            __builtin_amdgcn_s_waitcnt(WAIT_LGKMCNT0);     // my reads of this step are done: after the barrier the other group may overwrite them
            stamp();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp();
            // ---- COMPUTE: registers only
            __builtin_amdgcn_s_setprio(1);
            if (!(p.variant & 32)) {                       // (tuning: 32 = no MFMA)
                // (taps 1..4 rewrite pieces 0..3, which every wave has: no run-time condition, so the phase stays ONE basic block -- the
                //  scheduler deals instructions out inside a block only, and two copies of the phase behind a branch cost the register
                //  allocator its accumulators: 500 bytes of scratch.  In the last chunk there is nothing new to rewrite; the rewrite then
                //  re-converts whatever the other stage holds, which nothing reads before the next tile's prologue refills it.)
                constexpr bool CV = CV_IN_COMPUTE;
                const bool cv_c = CV && tap >= 1 && tap <= 4;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int t = 0; t < TL::TERMS; ++t) {
                        const int g = ks * TL::TERMS + t;              // MFMA group of the step (6 MFMAs each): the extra work sits behind groups 1 and 2
                        if (cv_c && g == 1) {
                            // (the first term's six MFMAs are issued, their operand registers are free: read the piece now -- its LDS latency
                            //  passes under the next three MFMAs -- and deal the rewrite out behind the remaining nine of the step)
                            __builtin_amdgcn_sched_barrier(0);
                            cv = convert_load(tap - 1, par ^ 1);
                        }
                        if (cv_c && g == 2) convert_store(cv, tap - 1, par ^ 1, !last);
                        if (FA_AHEAD && g == 1) {
                            if (!cv_c) __builtin_amdgcn_sched_barrier(0);
                            tap_addresses((tap + 1) % 9, fa_next);
                        }
#pragma unroll
                        for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                            for (int j = 0; j < TL::NTL; ++j) TL::mma(t, af[ks], bfr[ks], i, j, acc[i][j], x2_sc);
                        if (cv_c && g == 2) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // the read
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);         // three MFMAs cover its latency
#pragma unroll
                            for (int q = 0; q < 9; ++q) {                              // then one MFMA, up to four of the rewrite's / the address arithmetic's VALU / LDS instructions
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002 | 0x080, 4, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (FA_AHEAD && !cv_c && g == 2) {
#pragma unroll
                            for (int q = 0; q < 12; ++q) {                             // the next step's address arithmetic alone: one instruction behind each MFMA
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            } else {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                    for (int i = 0; i < TL::MT; ++i) NOPE_KEEP_VGPR(af[kk / RAW][kk % RAW][i]);
#pragma unroll
                    for (int j = 0; j < TL::NTL; ++j) NOPE_KEEP_VGPR(bfr[kk / RAW][kk % RAW][j]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            stamp();
            __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);       // the pieces issued in this step's LOAD have landed (they had this whole phase)
            stamp();
            if (!(grp == 1 && last && tap == 8)) {         // (group 1 started one barrier late: it skips the last one)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp();
        }
    };
    // ---- UP: the four K steps of channel chunk `chunk` (A stage `par`), see the kernel's header.  Weight ring stage of tap t: 1, 2, 0, 2.
    // K offsets are formed per issue (chunk base + tap * Cin): the two groups and the chunk-crossing issues do not walk one common sequence here.
    unsigned up_kc = 0;                                            // (chunk * BK) * ES of the chunk being multiplied
    auto up_issue_b = [&](int tap, bool next_chunk, bool other_half) __attribute__((always_inline)) {
        const int stage = tap == 2 ? 0 : (tap == 0 ? 1 : 2);
        const unsigned kofs = up_kc + (next_chunk ? (unsigned)BK * ES : 0u) + (unsigned)tap * cin_es;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)((other_half ? b_dst_o : b_dst) + stage * B_STAGE + j * 1024), 16, other_half ? b_off_o[j] : b_off[j], kofs, 0, 0);
    };
    auto up_tap_addresses = [&](int tp, int frow, int (&dst)[TL::MT]) __attribute__((always_inline)) {
        // tap tp = (ty, tx) of this workgroup's phase sits at (ty + ph_y, tx + ph_x) of the 3 x 3 neighbourhood (uniform at run time)
        const int dy = (tp >> 1) + ph_y, dx = (tp & 1) + ph_x;
        const int toff = (dy - 1) * W + (dx - 1);
        const unsigned bit = (unsigned)(dy * 3 + dx);
#pragma unroll
        for (int i = 0; i < TL::MT; ++i) {
            const int rr = frow + i * TL::TM + toff;
            const int off = A_BASE + (rr << 7) + (((fslot ^ (rr >> 1)) & 7) << 4);
            dst[i] = ((f_mask[i] >> bit) & 1u) ? off : A_BASE + ZROW;   // (absolute, stage 0)
        }
    };
    auto chunk_steps_up = [&](const int chunk, const int par) __attribute__((always_inline)) {
        const bool last = chunk + 1 == nchunks;
        if (!last) set_chunk(chunk + 1);
        up_kc = (unsigned)(chunk * BK) * ES;
        int frow = f_row0;
        NOPE_OPAQUE_VGPR(frow);
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            // ---- LOAD: DMA pieces first, then the fragments
            if (dma_on) {
                if (!last) {                                       // A pieces of the next chunk: {0, 4} / {1} / {2, 3} at taps 0 / 1 / 2 (piece 5 never exists: launcher, Ws <= 30)
                    if (tap == 0) { piece_a(0, par ^ 1); if (a_has4) piece_a(4, par ^ 1); }
                    if (tap == 1) piece_a(1, par ^ 1);
                    if (tap == 2) { piece_a(2, par ^ 1); piece_a(3, par ^ 1); }
                }
                if (grp == 0) {                                    // one tap ahead; BOTH halves of the two taps whose stage group 1 is reading when it would issue them
                    if (tap == 0) { up_issue_b(1, false, false); up_issue_b(1, false, true); }
                    else if (tap == 1) up_issue_b(2, false, false);
                    else if (tap == 2) { up_issue_b(3, false, false); up_issue_b(3, false, true); }
                    else if (!last) up_issue_b(0, true, false);
                } else {                                           // two taps ahead
                    if (tap == 0) up_issue_b(2, false, false);
                    else if (tap == 2 && !last) up_issue_b(0, true, false);
                }
            }
            u32x4 af[KS][RAW][TL::MT], bfr[KS][RAW][TL::NTL];
            int fa[TL::MT];
            if constexpr (FA_AHEAD) {
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) fa[i] = fa_next[i];
            } else up_tap_addresses(tap, frow, fa);
            constexpr int BSTG[4] = {1, 2, 0, 2};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) af[kk / RAW][kk % RAW][i] = ld16(lds + (fa[i] ^ (raw_slot<T>(kk) << 4)) + par * A_STAGE);
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j) bfr[kk / RAW][kk % RAW][j] = ld16(lds + fbk[kk] + (BSTG[tap] * B_STAGE + j * TL::TM * RB));
            }
            __builtin_amdgcn_s_waitcnt(WAIT_LGKMCNT0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- COMPUTE: the 3 x 3 kernel's, with the next chunk's stage rewritten behind the MFMAs of taps 1..3: pieces {0, 4} / {1} / {2, 3}
            // (each landed one tap earlier; piece 4 is rewritten by every wave -- its rows exist in every stage, and nothing reads them where no wave loaded them)
            __builtin_amdgcn_s_setprio(1);
            constexpr bool CV = CV_IN_COMPUTE;
            const bool cv_c = CV && tap >= 1;
            const int pa = tap == 1 ? 0 : (tap == 2 ? 1 : 2), pb = tap == 1 ? 4 : 3;      // pieces rewritten in this phase (pb: taps 1 and 3 only)
            const bool two = tap != 2;
            u32x4 cva = {0u, 0u, 0u, 0u}, cvb = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < TL::TERMS; ++t) {
                    const int g = ks * TL::TERMS + t;
                    if (cv_c && g == 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        cva = convert_load(pa, par ^ 1);
                        if (two) cvb = convert_load(pb, par ^ 1);
                    }
                    if (cv_c && g == 2) {
                        convert_store(cva, pa, par ^ 1, !last);
                        if (two) convert_store(cvb, pb, par ^ 1, !last);
                    }
                    if (FA_AHEAD && g == 1) {
                        if (!cv_c) __builtin_amdgcn_sched_barrier(0);
                        up_tap_addresses((tap + 1) & 3, frow, fa_next);
                    }
#pragma unroll
                    for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                        for (int j = 0; j < TL::NTL; ++j) TL::mma(t, af[ks], bfr[ks], i, j, acc[i][j], x2_sc);
                    if (cv_c && g == 2) {
                        // the read(s); three MFMAs cover their latency; then one MFMA, up to four (one piece) / six (two) of the rewrite's / the address arithmetic's instructions
                        if (two) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#pragma unroll
                            for (int q = 0; q < 9; ++q) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002 | 0x080, 6, 0);
                            }
                        } else {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#pragma unroll
                            for (int q = 0; q < 9; ++q) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002 | 0x080, 4, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (FA_AHEAD && !cv_c && g == 2) {
#pragma unroll
                        for (int q = 0; q < 12; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
            if (!(grp == 1 && last && tap == 3)) {                 // (group 1 started one barrier late: it skips the last one)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if constexpr (FA_AHEAD && UP) up_tap_addresses(0, f_row0, fa_next);
    if constexpr (FA_AHEAD && !UP) {                               // tap 0 of the first step (the row geometry is the same for every tile of a walk)
#pragma unroll
        for (int i = 0; i < TL::MT; ++i) {
            const int rr = f_row0 + i * TL::TM - W - 1;
            fa_next[i] = (f_mask[i] & 1u) ? A_BASE + (rr << 7) + (((fslot ^ (rr >> 1)) & 7) << 4) : A_BASE + ZROW;
        }
    }
    for (int it = 0; it < iters; ++it) {
        if (A_SPLIT_LDS && dma_on) {                               // the prologue's A pieces have landed (vmcnt(0) above / in the epilogue's drain)
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (i < 4 || (i == 4 ? a_has4 : a_has5)) convert_piece(i, 0);
        }
        __builtin_amdgcn_s_waitcnt(WAIT_LGKMCNT0);                 // zero rows written / my panel reads of the previous tile done
        __builtin_amdgcn_s_barrier();                              // every wave's prologue pieces have landed (waited by their issuers)
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) {                                            // group 1 runs one slot behind
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp();
        for (int chunk = c_lo; chunk < nchunks; chunk += 2) {
            if constexpr (UP) {
                chunk_steps_up(chunk, 0);
                if (chunk + 1 < nchunks) chunk_steps_up(chunk + 1, 1);
            } else {
                chunk_steps(chunk, 0);
                if (chunk + 1 < nchunks) chunk_steps(chunk + 1, 1);
            }
        }
        if (p.variant & 64) {                          // tuning only: no epilogue (keeps the accumulators live)
            if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.out)[0] = 1.f;
            return;
        }
        // All fragment reads of this tile are behind a barrier every wave has passed (group 0: the one after its last
        // COMPUTE, which group 1 reached after its last LOAD), nothing is in flight: the ring is free.  Start the next tile's
        // prologue now -- it lands while the panels are filled -- and wait for it before the first store of the epilogue
        // (so the K loop's vmcnt never has to wait for a prologue behind a queue of stores).
        if constexpr (X2 && NOPE_X2_KERNEL_AMAX) x2_publish_amax(p, x2_amax, lane);
        const int m_this = m0;
        const bool more = it + 1 < iters;
        if (more) {
            m0 += walk_rows;
            a_base1 += (unsigned)walk_rows * (unsigned)p.C1 * ES;
            NOPE_HALO_SET_OFF2();
            if (dma_on) tile_prologue();
        }
        if constexpr (SPLIT) {                         // (never with a tile walk: iters == 1)
            epilogue_split_wide<T>(p, acc, m_this, n0, wm, wn, lane, lds_panel, x2_out);
            if (prefetched) {
#pragma unroll
                for (int k = 0; k < PFN; ++k) NOPE_KEEP_VGPR(pfv[k]);
            }
            return;
        } else if constexpr (TIMELINE) epilogue_wide<T, false, true>(p, acc, m_this, n0, wm, wn, lane, lds_panel, stamp, x2_out);
        else epilogue_wide<T, false, true, NoStamp, false, LEAN ? 1 : 0>(p, acc, m_this, n0, wm, wn, lane, lds_panel, NoStamp(), x2_out);
        if (more) {
#pragma unroll
            for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j)
#pragma unroll
                    for (int r = 0; r < TL::R; ++r) acc[i][j][r] = 0.f;
        }
        stamp();
    }
    stamp_clocks(1);
    if constexpr (TIMELINE) {
        __syncthreads();
        if (blockIdx.x == 0 && p.timeline)
            for (int i = tid; i < 2 * TIMELINE_STAMPS; i += PP_WAVES * 64) p.timeline[i] = reinterpret_cast<unsigned*>(lds + LDS_USED)[i];
    }
}

template <class T, bool TUNE>
void launch_pp_tt(const ConvParams& p, dim3 grid, hipStream_t s) {
    const dim3 block(PP_WAVES * 64);
    if constexpr (sizeof(T) == 4 && Tile<T>::TM == 32 && !TUNE) {
        if (p.lean && !p.pn_ms && p.mode == NOPE_CONV_PLAIN) { hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_PLAIN, false, false, true>), grid, block, 0, s, p); return; }
        if (p.lean && p.mode == NOPE_CONV_DOWN2) { hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_DOWN2, false, false, true>), grid, block, 0, s, p); return; }
    }
    if (p.pn_ms) hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_PLAIN, true, TUNE>), grid, block, 0, s, p);
    else if (p.mode == NOPE_CONV_PLAIN) hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_PLAIN, false, TUNE>), grid, block, 0, s, p);
    else if (p.mode == NOPE_CONV_UP2P) hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_UP2P, false, TUNE>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_DOWN2, false, TUNE>), grid, block, 0, s, p);
}
template <class T>
void launch_pp_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    if (p.variant) launch_pp_tt<T, true>(p, grid, s);      // (NOPE_PP_VARIANT ablations: tools/pp_stream_probe.py)
    else launch_pp_tt<T, false>(p, grid, s);
}

}  // namespace

// Launch the ping-pong kernel for already validated parameters (called by launch_conv, kernels_gemm.hip).  Requires
// the LDS-DMA preconditions, wide_out, no split-K, mode PLAIN / DOWN2 / UP2P; tiles_m was computed for 256-row tiles.
void launch_conv_pp(int dt, const void* params, dim3 grid, hipStream_t s) {
    const ConvParams& p = *static_cast<const ConvParams*>(params);
    if (dt == NOPE_F32) launch_pp_t<float>(p, grid, s);
    else if (dt == NOPE_BF16X3) launch_pp_t<f32s_t>(p, grid, s);
    else if (dt == NOPE_F16X2) launch_pp_t<f16x2_t>(p, grid, s);
    else if (dt == NOPE_F16) launch_pp_t<f16_t>(p, grid, s);
    else launch_pp_t<bf16_t>(p, grid, s);
}

// The tap-resident 3x3 kernel: PLAIN mode, 9 taps, (sample, pixel) row order, maps at most conv_halo_max_width() wide.
int conv_halo_max_width() { return HALO_MAX_W; }
void launch_conv_halo(int dt, const void* params, dim3 grid, hipStream_t s) {
    const ConvParams& p = *static_cast<const ConvParams*>(params);
    const dim3 block(PP_WAVES * 64);
    if (dt == NOPE_BF16 && (p.variant & 256)) {        // tuning only: cycle stamps of workgroup 0 appended to $NOPE_PP_TIMELINE
        static unsigned* dev = nullptr;
        if (!dev && hipMalloc((void**)&dev, 2 * TIMELINE_STAMPS * 4) != hipSuccess) return;
        (void)hipMemsetAsync(dev, 0, 2 * TIMELINE_STAMPS * 4, s);
        ConvParams q = p;
        q.timeline = dev;
        hipLaunchKernelGGL((conv3x3_halo_kernel<bf16_t, true>), grid, block, 0, s, q);
        unsigned host[2 * TIMELINE_STAMPS];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* f = fopen(getenv("NOPE_PP_TIMELINE") ? getenv("NOPE_PP_TIMELINE") : "/tmp/nope_pp_timeline.txt", "a")) {
                fprintf(f, "halo Cin %d Cout %d M %d W %d iters %d\n", p.C1 + p.C2, p.Cout, p.M, p.Ws, p.persist_iters);
                for (int g = 0; g < 2; ++g) {
                    for (int i = 0; i < TIMELINE_STAMPS; ++i) fprintf(f, "%u ", host[g * TIMELINE_STAMPS + i]);
                    fprintf(f, "\n");
                }
                fclose(f);
            }
        return;
    }
    if (p.mode == NOPE_CONV_UP2P) {                    // the four phase convs of an up-sampling (plan_conv: f32 storage only)
        if (dt == NOPE_F16X2 && p.x2_t_zero) hipLaunchKernelGGL((conv3x3_halo_kernel<f16x2_t, false, false, false, false, true>), grid, block, 0, s, p);
        else if (dt == NOPE_F16X2) hipLaunchKernelGGL((conv3x3_halo_kernel<f16x2_t, false, false, true, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<f32s_t, false, false, true, false, true>), grid, block, 0, s, p);
        return;
    }
    if (p.splits > 1) {
        if (dt == NOPE_F32) hipLaunchKernelGGL((conv3x3_halo_kernel<float, false, true>), grid, block, 0, s, p);
        else if (dt == NOPE_BF16X3) hipLaunchKernelGGL((conv3x3_halo_kernel<f32s_t, false, true>), grid, block, 0, s, p);
        else if (dt == NOPE_F16) hipLaunchKernelGGL((conv3x3_halo_kernel<f16_t, false, true>), grid, block, 0, s, p);
        else if (dt == NOPE_F16X2 && p.x2_t_zero) hipLaunchKernelGGL((conv3x3_halo_kernel<f16x2_t, false, true, false>), grid, block, 0, s, p);
        else if (dt == NOPE_F16X2) hipLaunchKernelGGL((conv3x3_halo_kernel<f16x2_t, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<bf16_t, false, true>), grid, block, 0, s, p);
        return;
    }
    if (p.lean && dt == NOPE_BF16X3) { hipLaunchKernelGGL((conv3x3_halo_kernel<f32s_t, false, false, true, true>), grid, block, 0, s, p); return; }
    if (p.lean && dt == NOPE_F16X2) {
        if (p.x2_t_zero) hipLaunchKernelGGL((conv3x3_halo_kernel<f16x2_t, false, false, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<f16x2_t, false, false, true, true>), grid, block, 0, s, p);
        return;
    }
    if (dt == NOPE_F32) hipLaunchKernelGGL((conv3x3_halo_kernel<float>), grid, block, 0, s, p);
    else if (dt == NOPE_BF16X3) hipLaunchKernelGGL((conv3x3_halo_kernel<f32s_t>), grid, block, 0, s, p);
    else if (dt == NOPE_F16) hipLaunchKernelGGL((conv3x3_halo_kernel<f16_t>), grid, block, 0, s, p);
    else if (dt == NOPE_F16X2 && p.x2_t_zero) hipLaunchKernelGGL((conv3x3_halo_kernel<f16x2_t, false, false, false>), grid, block, 0, s, p);
    else if (dt == NOPE_F16X2) hipLaunchKernelGGL((conv3x3_halo_kernel<f16x2_t>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<bf16_t>), grid, block, 0, s, p);
}

}  // namespace nope
