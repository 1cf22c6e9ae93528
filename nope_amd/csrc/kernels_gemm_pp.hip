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

namespace nope {

namespace {

constexpr int PP_BM = 256;
constexpr int PP_WAVES = 8;

// s_waitcnt simm16 on gfx950: vmcnt = [3:0] | [15:14] << 4, expcnt = [6:4], lgkmcnt = [11:8]; unused counters at their maximum
constexpr int WAIT_VMCNT0 = 0x0F70;      // vmcnt(0): every LDS-DMA piece this wave issued has landed
constexpr int WAIT_LGKMCNT0 = 0xC07F;    // lgkmcnt(0): every ds_read of this wave has returned

struct KPos { int tap, kc; };

template <class T, int MODE, bool PN>
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
    constexpr int KK = RB / 16 / TL::KSLOTS;
    static_assert(MODE == NOPE_CONV_PLAIN || MODE == NOPE_CONV_DOWN2 || MODE == NOPE_CONV_UP2P, "modes of the U-Net's large launches");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];     // the ONLY LDS object (cdna_hip_programming.md, section 5 trap (a))

    const int tid = threadIdx.x;
    if (p.variant & 128) { if (tid == 9999) lds[0] = 1; return; }   // tuning only (NOPE_PP_VARIANT): launch cost of the grid
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // 4 (M) x 2 (N) waves of 64 x 96
    const int grp = wave >> 2, wl = wave & 3;
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

    // fragment addresses of K sub-step 0 inside a stage; sub-step kk flips slot bits: off ^ (kk * KSLOTS << 4)
    int fa[TL::MT], fb[TL::NTL];
#pragma unroll
    for (int i = 0; i < TL::MT; ++i) fa[i] = lds_off_rb<RB>(wm * 64 + i * TL::TM + TL::frag_row(lane), TL::frag_slot(lane));
#pragma unroll
    for (int j = 0; j < TL::NTL; ++j) fb[j] = lds_off_rb<RB>(wn * 96 + j * TL::TM + TL::frag_row(lane), TL::frag_slot(lane));

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
    for (int k = 0; k < nk; ++k) {
        // ---- LOAD k: fragments of step k into registers, then the DMA pieces of the steps after it
        const unsigned char* la = lds + sa * A_STAGE;
        const unsigned char* lb = lds + B_BASE + sb * B_STAGE;
        u32x4 af[KK][TL::MT], bfr[KK][TL::NTL];
        const int sa1 = sa ^ 1;
        const int sb1 = sb == 2 ? 0 : sb + 1;
        const int sb2 = sb1 == 2 ? 0 : sb1 + 1;
        // (tuning: 16 = no DMA stream, 512 = A pieces only on the first tap of a channel chunk, 1024 = no B pieces: wrong results,
        //  they size what a tap-resident A stage / a cheaper B stream would buy)
        const bool do_a = k + 1 < nk && !(p.variant & 16) && !((p.variant & 512) && ka.tap != tap0);
        const bool do_b = (grp == 0 ? k + 1 < nk : k + 2 < nk) && !(p.variant & (16 | 1024));
        if (p.variant & 4) __builtin_amdgcn_s_setprio(2);      // (tuning: 4 = the LOAD phase outranks the other group's MFMA issue)
        if (do_a) prep_a(ka, a_dst + sa1 * A_STAGE);
        if (do_b) prep_b(kb, b_dst + (grp == 0 ? sb1 : sb2) * B_STAGE);
        if (p.variant & 1) {                           // (tuning: 1 = all fragment reads first, then all DMA pieces)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) af[kk][i] = ld16(la + (fa[i] ^ ((kk * TL::KSLOTS) << 4)));
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j) bfr[kk][j] = ld16(lb + (fb[j] ^ ((kk * TL::KSLOTS) << 4)));
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
                if (r < TL::MT) af[kk][r] = ld16(la + (fa[r] ^ ((kk * TL::KSLOTS) << 4)));
                else bfr[kk][r - TL::MT] = ld16(lb + (fb[r - TL::MT] ^ ((kk * TL::KSLOTS) << 4)));
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
        if (k + 1 < nk) advance(ka);
        if (grp == 0 ? k + 1 < nk : k + 2 < nk) advance(kb);
        __builtin_amdgcn_s_waitcnt(WAIT_LGKMCNT0);     // my reads of stage k are done: after the barrier the other group may overwrite it
        if (p.variant & 4) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- COMPUTE k: registers only
        if (!(p.variant & 2)) __builtin_amdgcn_s_setprio(1);       // (tuning: 2 = no priority for the MFMA phase)
        if (!(p.variant & 32)) {                       // (tuning: 32 = no MFMA)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                    for (int j = 0; j < TL::NTL; ++j) TL::mma(af[kk][i], bfr[kk][j], acc[i][j]);
        } else {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {          // keep the fragment reads alive
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) NOPE_KEEP_VGPR(af[kk][i]);
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j) NOPE_KEEP_VGPR(bfr[kk][j]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (!(p.variant & 8))                          // (tuning: 8 = never wait for the DMA -- wrong results, shows the issue-bound time)
        __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);       // the pieces I issued in LOAD k have landed (they had this whole phase)
        if (!(grp == 1 && k == nk - 1)) {              // (group 1 started one barrier late: it skips the last one)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        sa = sa1; sb = sb1;
    }
    // Both groups have passed 2 nk + 1 barriers.  Group 0 arrives here while group 1 still multiplies (registers only);
    // every LDS stage read and every DMA is complete, so the per-wave epilogue panels may reuse the ring.
    if (p.variant & 64) {                              // tuning only: no epilogue (keeps the accumulators live)
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.out)[0] = 1.f;
        return;
    }
    epilogue_wide<T, PN>(p, acc, m0, n0, wm, wn, lane, lds + wave * Ep<T>::WAVE_BYTES);
}

// ---- 3x3 convolutions: the A operand stays in LDS across the 9 taps ------------------------------------------------
// With GEMM rows in (sample, pixel) order, tap (dy, dx) of tile row i is the pixel dy*W + dx rows further along the
// same flat pixel axis.  So instead of staging 256 rows per tap (9 x 32 KiB per channel chunk), a stage holds the tile's
// pixel range extended by W + 1 rows on either side (<= 328 rows, 41 KiB, loaded ONCE per channel chunk) and every tap
// reads its fragments at a row offset.  A tap that leaves the image (zero padding) or wraps into a neighbouring image
// row / sample is redirected, per lane, to a 128-byte row of zeros -- one v_cndmask on the address, none on the data.
// Per channel chunk the L2 -> LDS stream is 41 KiB of A + 9 x 24 KiB of B instead of 9 x 56 KiB: half the bytes, and
// 4 instead of 7 DMA pieces per wave per K step (the vector-memory path of a CU, ~40 B/clk, was the longest pole of the
// LOAD phase: profiles/r02_pp_ablation.txt).  Same K order (channel chunk outer, tap inner, padding taps add exact
// zeros) and the same accumulation chain as the other conv kernels: results are bit-identical to theirs.
// Schedule, B ring, barriers and waits are those of conv_gemm_pp_kernel above; the A ring has two stages of whole
// chunks: the (<= 6) pieces a wave owns of chunk c+1 are issued one per K step during taps 0..5 of chunk c, into the
// stage chunk c-1 was read from (last read: K step 9c-1, one barrier-separated slot before the first such issue), and
// have landed -- issuer's vmcnt(0) + barrier -- at least three K steps before chunk c+1 begins.
constexpr int HALO_ROWS = 328;       // 256 + 2 * (32 + 1), rounded up to whole 8-row pieces: maps up to 32 pixels wide
constexpr int HALO_MAX_W = 32;

template <class T>
__global__ __launch_bounds__(PP_WAVES * 64, 2) void conv3x3_halo_kernel(ConvParams p) {
    typedef Tile<T> TL;
    constexpr int VEC = Elt<T>::VEC;
    constexpr unsigned ES = (unsigned)sizeof(T);
    constexpr int RB = 128;
    constexpr int BK = RB / (int)ES;
    constexpr int A_STAGE = HALO_ROWS * RB, B_STAGE = BN * RB;
    constexpr int B_BASE = 2 * A_STAGE;
    constexpr int ZERO_OFF = B_BASE + 3 * B_STAGE;                 // 128 bytes of zeros (a multiple of 128: the kk slot flips stay inside)
    constexpr int RING = ZERO_OFF + RB;
    constexpr int PANELS = PP_WAVES * Ep<T>::WAVE_BYTES;
    constexpr int LDS_BYTES = RING > PANELS ? RING : PANELS;
    constexpr int KK = RB / 16 / TL::KSLOTS;
    static_assert(PANELS <= ZERO_OFF, "epilogue panels must not need more than the ring");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = wave >> 2, wl = wave & 3;
    int tile_m, tile_n;
    tile_coords(p, tile_m, tile_n);
    const int m0 = tile_m * PP_BM, n0 = tile_n * BN;
    const int W = p.Ws, HW = p.Hs * p.Ws;
    const int halo = W + 1;
    const int Cin = p.C1 + p.C2;
    const int npieces = (PP_BM + 2 * halo + 7) >> 3;               // 8-row pieces of a stage (<= 41)

    const auto r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src1, (short)0, (int)p.bytes1, 0x00020000);
    const auto r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src2 ? p.src2 : p.src1), (short)0, (int)(p.src2 ? p.bytes2 : p.bytes1), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)p.bytesw, 0x00020000);

    if (p.variant & 128) { if (tid == 9999) lds[0] = 1; return; }   // tuning only (NOPE_PP_VARIANT): launch cost of the grid
    if (tid < 8) st16(lds + ZERO_OFF + tid * 16, u32x4{0u, 0u, 0u, 0u});

    // ---- A pieces of this wave: stage rows 8 q .. 8 q + 7 for q = wave, wave + 8, ... ; stage row r holds flat pixel
    // m0 - halo + r of the (hypothesis, y, x) axis (zeros outside the tensor: out-of-range buffer offset)
    const int rsub = lane >> 3, lslot = lane & 7;
    unsigned a_o1[6], a_o2[6];
    const long long m_total = (long long)p.nhyp * HW;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = wave + 8 * i;
        const int r = 8 * q + rsub;
        const long long gm = (long long)m0 - halo + r;
        const bool ok = q < npieces && gm >= 0 && gm < m_total;
        const unsigned g = ok ? (unsigned)gm : 0u;
        const unsigned b = p.d_hw.div(g), pix = g - b * (unsigned)HW;
        const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(r)) * VEC);
        a_o1[i] = ok ? ((p.d_rep1.div(b) * (unsigned)HW + pix) * p.C1 + cs) * ES : OOB;
        a_o2[i] = ok ? ((p.d_rep2.div(b) * (unsigned)HW + pix) * p.C2 + cs) * ES : OOB;
    }
    auto piece_a = [&](int i, int chunk, int stage) {              // piece i of this wave, channel chunk `chunk`
        const int c0 = chunk * BK;
        const bool first = c0 < p.C1;                              // wave-uniform: a chunk lies inside one source
        const unsigned off = (first ? a_o1[i] : a_o2[i]) + (unsigned)(first ? c0 : c0 - p.C1) * ES;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? r1 : r2, (lds_void_t*)(lds + stage * A_STAGE + (wave + 8 * i) * 1024), 16, off, 0, 0, 0);
    };
    // ---- B pieces: 24 rows per wave (group 1: panel rows 0..95, group 0: rows 96..191), as in conv_gemm_pp_kernel
    const int brow0 = 96 * (1 - grp) + 24 * wl;
    unsigned b_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int row = brow0 + 8 * j + rsub;
        const int n = n0 + row;
        const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(row)) * VEC);
        b_off[j] = n < p.Cout ? ((unsigned)n * 9u * Cin + cs) * ES : OOB;
    }
    unsigned char* const b_dst = lds + B_BASE + (brow0 >> 3) * 1024;
    auto issue_b = [&](int tap, int chunk, int stage) {
        const unsigned kofs = (unsigned)(tap * Cin + chunk * BK) * ES;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(b_dst + stage * B_STAGE + j * 1024), 16, b_off[j] + kofs, 0, 0, 0);
    };

    // ---- fragment rows of this lane: tile row, and which of the 9 taps stay inside its image
    int f_row[TL::MT];
    unsigned f_mask[TL::MT];
#pragma unroll
    for (int i = 0; i < TL::MT; ++i) {
        const int il = wm * 64 + i * TL::TM + TL::frag_row(lane);
        const unsigned m = (unsigned)(m0 + il);
        const unsigned b = p.d_hw.div(m), r = m - b * (unsigned)HW;
        const int oy = (int)p.d_w.div(r), ox = (int)r - oy * W;
        const unsigned vx = (ox > 0 ? 1u : 0u) | 2u | (ox + 1 < W ? 4u : 0u);
        f_mask[i] = (oy > 0 ? vx : 0u) | (vx << 3) | (oy + 1 < p.Hs ? vx << 6 : 0u);
        f_row[i] = il + halo;
    }
    const int fslot = TL::frag_slot(lane);
    int fb[TL::NTL];
#pragma unroll
    for (int j = 0; j < TL::NTL; ++j) fb[j] = lds_off_rb<RB>(wn * 96 + j * TL::TM + TL::frag_row(lane), fslot);

    typename TL::acc_t acc[TL::MT][TL::NTL];
#pragma unroll
    for (int i = 0; i < TL::MT; ++i)
#pragma unroll
        for (int j = 0; j < TL::NTL; ++j)
#pragma unroll
            for (int r = 0; r < TL::R; ++r) acc[i][j][r] = 0.f;

    const int nchunks = Cin / BK;
    const int nk = 9 * nchunks;

    // ---- prologue: the whole A stage of chunk 0, this group's half of B(0), and (group 1) its half of B(1)
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (wave + 8 * i < npieces) piece_a(i, 0, 0);
    int btap = 0, bchunk = 0;                          // next K step whose B pieces this wave issues
    auto b_advance = [&]() { if (++btap == 9) { btap = 0; ++bchunk; } };
    issue_b(btap, bchunk, 0); b_advance();
    if (grp == 1 && nk > 1) { issue_b(btap, bchunk, 1); b_advance(); }
    __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0 & WAIT_LGKMCNT0);       // DMA landed, zero row written
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    int sb = 0, tap = 0, chunk = 0;
    for (int k = 0; k < nk; ++k) {
        // ---- LOAD k
        const unsigned char* la = lds + (chunk & 1) * A_STAGE;
        const unsigned char* lb = lds + B_BASE + sb * B_STAGE;
        const int dy = tap / 3, dx = tap - dy * 3;
        const int toff = (p.variant & 4) ? 0 : (dy - 1) * W + (dx - 1);          // (tuning: 4 = every tap reads the centre rows)
        int fa[TL::MT];
#pragma unroll
        for (int i = 0; i < TL::MT; ++i) {
            const int r = f_row[i] + toff;
            const int off = (r << 7) + (((fslot ^ (r >> 1)) & 7) << 4);
            fa[i] = (((f_mask[i] >> tap) & 1u) || (p.variant & 2)) ? off : (ZERO_OFF - (chunk & 1) * A_STAGE);     // (relative to `la`; tuning: 2 = no padding redirect)
        }
        u32x4 af[KK][TL::MT], bfr[KK][TL::NTL];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int i = 0; i < TL::MT; ++i) af[kk][i] = ld16(la + (fa[i] ^ ((kk * TL::KSLOTS) << 4)));
#pragma unroll
            for (int j = 0; j < TL::NTL; ++j) bfr[kk][j] = ld16(lb + (fb[j] ^ ((kk * TL::KSLOTS) << 4)));
        }
        const int sb1 = sb == 2 ? 0 : sb + 1;
        const int sb2 = sb1 == 2 ? 0 : sb1 + 1;
        const bool dma_on = !(p.variant & 16);                                  // (tuning: 16 = no DMA stream)
        if (dma_on && tap < 6 && chunk + 1 < nchunks && wave + 8 * tap < npieces) {      // one piece of the NEXT chunk's A stage
            // (static piece index for the register arrays: the switch unrolls)
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (i == tap) piece_a(i, chunk + 1, (chunk + 1) & 1);
        }
        if (grp == 0) { if (k + 1 < nk) { if (dma_on) issue_b(btap, bchunk, sb1); b_advance(); } }
        else          { if (k + 2 < nk) { if (dma_on) issue_b(btap, bchunk, sb2); b_advance(); } }
        __builtin_amdgcn_s_waitcnt(WAIT_LGKMCNT0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- COMPUTE k
        __builtin_amdgcn_s_setprio(1);
        if (!(p.variant & 32)) {                       // (tuning: 32 = no MFMA)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                    for (int j = 0; j < TL::NTL; ++j) TL::mma(af[kk][i], bfr[kk][j], acc[i][j]);
        } else {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < TL::MT; ++i) NOPE_KEEP_VGPR(af[kk][i]);
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j) NOPE_KEEP_VGPR(bfr[kk][j]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
        if (!(grp == 1 && k == nk - 1)) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        sb = sb1;
        if (++tap == 9) { tap = 0; ++chunk; }
    }
    if (p.variant & 64) {                              // tuning only: no epilogue (keeps the accumulators live)
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.out)[0] = 1.f;
        return;
    }
    epilogue_wide<T, false>(p, acc, m0, n0, wm, wn, lane, lds + wave * Ep<T>::WAVE_BYTES);
}

template <class T>
void launch_pp_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    const dim3 block(PP_WAVES * 64);
    if (p.pn_ms) hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_PLAIN, true>), grid, block, 0, s, p);
    else if (p.mode == NOPE_CONV_PLAIN) hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_PLAIN, false>), grid, block, 0, s, p);
    else if (p.mode == NOPE_CONV_UP2P) hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_UP2P, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_gemm_pp_kernel<T, NOPE_CONV_DOWN2, false>), grid, block, 0, s, p);
}

}  // namespace

// Launch the ping-pong kernel for already validated parameters (called by launch_conv, kernels_gemm.hip).  Requires
// the LDS-DMA preconditions, wide_out, no split-K, mode PLAIN / DOWN2 / UP2P; tiles_m was computed for 256-row tiles.
void launch_conv_pp(int dt, const void* params, dim3 grid, hipStream_t s) {
    const ConvParams& p = *static_cast<const ConvParams*>(params);
    if (dt == NOPE_F32) launch_pp_t<float>(p, grid, s);
    else launch_pp_t<bf16_t>(p, grid, s);
}

// The tap-resident 3x3 kernel: PLAIN mode, 9 taps, (sample, pixel) row order, maps at most conv_halo_max_width() wide.
int conv_halo_max_width() { return HALO_MAX_W; }
void launch_conv_halo(int dt, const void* params, dim3 grid, hipStream_t s) {
    const ConvParams& p = *static_cast<const ConvParams*>(params);
    const dim3 block(PP_WAVES * 64);
    if (dt == NOPE_F32) hipLaunchKernelGGL((conv3x3_halo_kernel<float>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<bf16_t>), grid, block, 0, s, p);
}

}  // namespace nope
