// Implicit-GEMM convolution for gfx950 (MI355X): every conv / linear of the NOPE U-Net.
//
//   out[m, n] = sum_{tap, c} A[m, tap, c] * W[n, tap, c] + bias[n] (+ resid[m, n])
//
// m = (hypothesis, oy, ox) over NHWC activations, n = output channel, K = taps x Cin.
// The A operand is never materialised: the loader walks taps and channel chunks directly
// over one or two NHWC sources (virtual torch.cat, u_net.py:186,189,194), optionally through
// a nearest-x2 upsample (HardUpsample, model_utils.py:161-165) or a 2x2 space-to-depth
// (HardDownsample, :168-172; the weight's K axis is re-ordered at pack time so each of the 4
// taps reads a contiguous channel run), and sources may be shared by `rep` consecutive
// hypotheses (the reference embedding feeding all N pose hypotheses, model.py:219).
//
// Tiling (64-wide wavefronts): 128 x 192 output tile per 256-thread workgroup, 4 waves as
// 2(M) x 2(N), each wave 64 x 96 = 4 x 6 MFMA 16x16 tiles (96 accumulator VGPRs).  192
// divides every Cout of the network (192/384/768/1536 and the 384-wide qkv).  K step = 128
// bytes per row (64 bf16 / 32 f32).  LDS rows are 128 B = 8 x 16-B slots, XOR-swizzled by
// (row >> 1) & 7 so the ds_read_b128 fragment reads (16 rows x one slot per lane group) are
// bank-conflict free (MI355X_MICROARCH.md, LDS table).
//
// Two kernels share the MFMA stage / epilogues:
//   conv_gemm_dma_kernel  the fast path: LDS-DMA staging (buffer_load ... lds), two 40 KiB stages,
//                         one barrier per K step; needs Cin (and C1 of a concat) to be a multiple of
//                         the K step.  Used by every conv of the real network.
//   conv_gemm_kernel      any channel count (global -> VGPR -> LDS staging): init conv (Cin = 8) and
//                         the tiny nets of the parity tests.
// Epilogue: accumulators (+bias) are staged per wave through LDS in f32 and written as 16-byte rows
// (epilogue_wide); on the way the per-64-row-block column sums needed by the following GroupNorm are
// emitted (colstats), so no separate statistics pass reads the tensor again.
//
// MFMA: bf16 -> v_mfma_f32_16x16x32_bf16; f32 -> v_mfma_f32_16x16x4_f32 (exact f32 fma
// chain; gfx950 has no xf32).  For f32 each lane's 16-byte fragment holds 4 consecutive
// channels that feed 4 successive 16x16x4 steps - the reduction order inside a 32-channel
// chunk is permuted identically for A and W, which only reorders the fp32 sum.
//
// Workgroup -> tile map is XCD-aware: blocks that land on one XCD (blockIdx % 8) share the
// weight panel (tile_n) and a contiguous run of M tiles (shared 3x3 halo rows stay in that L2).
#include <cstdio>
#include <cstdlib>

#include "conv_gemm_dma.h"      // (conv_gemm_common.h + epilogue_split; the LDS-DMA kernel itself is instantiated in kernels_gemm_dma_*.hip)

namespace nope {

namespace {

// out = act(sum_z part[z] + bias + resid): NHWC T, or NCHW f32 (out_nchw).
template <class T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int Cout,
                                                            const float* __restrict__ bias, const T* __restrict__ resid, int act,
                                                            T* __restrict__ out, int out_nchw, int HWo) {
    const size_t MN = (size_t)M * Cout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < MN; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i % Cout);
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += part[(size_t)z * MN + i];
        if (bias) v += bias[n];
        if (resid) v += Elt<T>::ld(resid + i);
        if (act) v = v > 0.f ? v : 0.f;
        if (out_nchw) {
            const size_t m = i / Cout;
            const size_t b = m / HWo;
            reinterpret_cast<float*>(out)[(b * Cout + n) * HWo + (m - b * HWo)] = v;
        } else Elt<T>::st(out + i, v);
    }
}

// The same reduction for NHWC outputs that feed a GroupNorm: one thread per column walks the `rows` rows of a statistics block
// (rows = min(H W, 64): whole samples or 64-row blocks of one sample), so the (sum, sum of squares) of what it writes -- f32, before the
// rounding to T, like the conv epilogues' -- come for free: colstats[M / rows][Cout][2], one writer per entry.  Grid (M / rows, Cout / 256).
template <class T>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ part, int splits, int M, int Cout,
                                                                  const float* __restrict__ bias, int act, T* __restrict__ out,
                                                                  float* __restrict__ colstats, int rows) {
    // 64 columns x 4 row quarters per workgroup; a thread adds the partials of its rows / 4 rows (all split loads of a row in flight
    // at once), the four quarters of a column are then added in row order by its first thread.  Grid (M / rows, Cout / 64).
    __shared__ float s_s[4][64], s_q[4][64];
    const int c = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + c;
    const size_t MN = (size_t)M * Cout;
    const int rper = rows >> 2;
    const int m0 = blockIdx.x * rows + rq * rper;
    float s = 0.f, q = 0.f;
    if (n < Cout) {
        const float bv = bias ? bias[n] : 0.f;
        for (int r = 0; r < rper; ++r) {
            const size_t i = (size_t)(m0 + r) * Cout + n;
            float pv[16];
#pragma unroll
            for (int z = 0; z < 16; ++z) pv[z] = z < splits ? part[(size_t)z * MN + i] : 0.f;
            float v = 0.f;
#pragma unroll
            for (int z = 0; z < 16; ++z) if (z < splits) v += pv[z];
            v += bv;
            if (act) v = v > 0.f ? v : 0.f;
            Elt<T>::st(out + i, v);
            s += v; q += v * v;
        }
    }
    s_s[rq][c] = s; s_q[rq][c] = q;
    __syncthreads();
    if (rq == 0 && n < Cout) {
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { S += s_s[k][c]; Q += s_q[k][c]; }
        float* cs = colstats + ((size_t)blockIdx.x * Cout + n) * 2;
        cs[0] = S; cs[1] = Q;
    }
}

// Source pixel of output pixel (oy, ox) under tap (dy, dx); false = zero padding / beyond M.
__device__ __forceinline__ bool tap_pixel(const ConvParams& p, int oy, int ox, int dy, int dx, int& iy, int& ix) {
    if (p.mode == NOPE_CONV_DOWN2) { iy = 2 * oy + dy; ix = 2 * ox + dx; return oy >= 0; }
    if (p.mode == NOPE_CONV_STRIDE2) {   // stride 2, pad 1 (3x3) / pad 0 (1x1)
        iy = 2 * oy + dy; ix = 2 * ox + dx;
        return oy >= 0 && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
    }
    if (p.mode == NOPE_CONV_UP2P) {   // rows are source pixels; (dy, dx) already include the phase shift
        iy = oy + dy; ix = ox + dx;
        return iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
    }
    if (p.mode == NOPE_CONV_UP2) {
        const int uy = oy + dy, ux = ox + dx;
        iy = uy >> 1; ix = ux >> 1;
        return uy >= 0 && uy < p.Ho && ux >= 0 && ux < p.Wo;
    }
    iy = oy + dy; ix = ox + dx;
    return iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
}
__device__ __forceinline__ void tap_delta(const ConvParams& p, int tap, int& dy, int& dx) {
    dy = 0; dx = 0;
    if (p.mode == NOPE_CONV_DOWN2) { dy = tap >> 1; dx = tap & 1; }
    else if (p.mode == NOPE_CONV_UP2P) { dy = (tap >> 1) + ((int)blockIdx.y >> 1) - 1; dx = (tap & 1) + ((int)blockIdx.y & 1) - 1; }
    else if (p.ntaps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
    else if (p.ntaps == 16) { dy = (tap >> 2) - 1; dx = (tap & 3) - 1; }      // 4x4, pad 1 (STRIDE2 only)
}

// ---- generic kernel: global -> VGPR -> LDS staging, any channel counts ----------------------------
template <class T, bool PN>
__global__ __launch_bounds__(NT, 2) void conv_gemm_kernel(ConvParams p) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int BK = 8 * VEC;
    constexpr int ES = (int)sizeof(T);
    constexpr int LDS_BYTES = (BM + BN) * ROWB > 4 * Ep<T>::WAVE_BYTES ? (BM + BN) * ROWB : 4 * Ep<T>::WAVE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];   // K-step tile, then the epilogue panels
    unsigned char* ldsA = lds;
    unsigned char* ldsB = lds + BM * ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    tile_coords(p, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int slot = tid & 7;
    const int rbase = tid >> 3;
    const int HWo = p.Hm * p.Wm;
    const int Cin = p.C1 + p.C2;
    const unsigned char* wbase = p.w + (size_t)blockIdx.y * p.w_phase_bytes;

    int a_s1[A_ITERS], a_s2[A_ITERS], a_oy[A_ITERS], a_ox[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int m = m0 + rbase + 32 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / HWo;
        const int r = mm - b * HWo;
        const int oy = r / p.Wm;
        a_oy[i] = ok ? oy : -100000;   // poisons the bounds test for rows beyond M
        a_ox[i] = r - oy * p.Wm;
        a_s1[i] = b / p.rep1;
        a_s2[i] = b / p.rep2;
    }

    const int kc_per_tap = (Cin + BK - 1) / BK;
    const int nk = p.ntaps * kc_per_tap;

    u32x4 ra[A_ITERS], rb[B_ITERS];
    int ld_tap = 0, ld_kc = 0;

    auto load_step = [&]() {
        const int c = ld_kc * BK + slot * VEC;
        int dy, dx;
        tap_delta(p, ld_tap, dy, dx);
        const bool c_ok = c < Cin;
        const bool first = c < p.C1;
        const unsigned char* sbase = first ? p.src1 : p.src2;
        const int cs = first ? c : c - p.C1;
        const int Cs = first ? p.C1 : p.C2;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            int iy, ix;
            const bool ok = tap_pixel(p, a_oy[i], a_ox[i], dy, dx, iy, ix) && c_ok;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) {
                const int sb = first ? a_s1[i] : a_s2[i];
                const size_t pix = ((size_t)sb * p.Hs + iy) * p.Ws + ix;
                v = ld16(sbase + (pix * Cs + cs) * ES);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int j = 0; j < B_ITERS; ++j) {
            const int n = n0 + rbase + 32 * j;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n < p.Cout && c_ok) v = ld16(wbase + (((size_t)n * p.ntaps + ld_tap) * Cin + c) * ES);
            rb[j] = v;
        }
        if (++ld_kc == kc_per_tap) { ld_kc = 0; ++ld_tap; }
    };

    typename Tile<T>::acc_t acc[Tile<T>::MT][Tile<T>::NTL];
#pragma unroll
    for (int i = 0; i < Tile<T>::MT; ++i)
#pragma unroll
        for (int j = 0; j < Tile<T>::NTL; ++j)
#pragma unroll
            for (int r = 0; r < Tile<T>::R; ++r) acc[i][j][r] = 0.f;

    load_step();
    for (int ks = 0; ks < nk; ++ks) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) st16(ldsA + lds_off_rb<ROWB>(rbase + 32 * i, slot), ra[i]);
#pragma unroll
        for (int j = 0; j < B_ITERS; ++j) st16(ldsB + lds_off_rb<ROWB>(rbase + 32 * j, slot), rb[j]);
        __syncthreads();
        if (ks + 1 < nk) load_step();   // global loads for step ks+1 fly under the MFMAs below
        mma_stage<T, ROWB>(ldsA, ldsB, wm, wn, lane, acc);
        __syncthreads();
    }
    if (PN) {
        if (p.wide_out) epilogue_wide<T, true>(p, acc, m0, n0, wm, wn, lane, lds + wave * Ep<T>::WAVE_BYTES);
        else epilogue<T, true>(p, acc, m0, n0, wm, wn, lane);
    } else {
        if (p.wide_out) epilogue_wide<T, false>(p, acc, m0, n0, wm, wn, lane, lds + wave * Ep<T>::WAVE_BYTES);
        else if (p.nchw_staged) epilogue_nchw<T>(p, acc, m0, n0, wm, wn, lane, lds + wave * Ep<T>::WAVE_BYTES);
        else epilogue<T, false>(p, acc, m0, n0, wm, wn, lane);
    }
}

}  // namespace

// Split-K factor for a conv that would otherwise leave most of the 256 CUs idle (few output tiles, long K):
// enough K slices to reach ~2 workgroups per CU, at least 4 K steps each.  1 = do not split.
constexpr int POSMAJOR_MAX_HW = 64;

void launch_conv_dma_f32(const void* params, int bm, dim3 grid, hipStream_t s);            // kernels_gemm_dma_*.hip
void launch_conv_dma_bf16x3(const void* params, dim3 grid, hipStream_t s);
void launch_conv_dma_f16(const void* params, dim3 grid, hipStream_t s);
void launch_conv_dma_bf16(const void* params, dim3 grid, hipStream_t s);
void launch_conv_dma_bf16_variant(const void* params, int bm, dim3 grid, hipStream_t s);
void launch_conv_pp(int dt, const void* params, dim3 grid, hipStream_t s);   // kernels_gemm_pp.hip
void launch_conv_small(int dt, const void* params, int tile, dim3 grid, hipStream_t s);   // kernels_gemm_small.hip
void launch_conv_halo(int dt, const void* params, dim3 grid, hipStream_t s);
void launch_conv_stream(int dt, const void* params, dim3 grid, hipStream_t s);   // kernels_gemm_stream.hip
int conv_halo_max_width();

// Which kernel a conv takes.  NOPE_CONV_PP (tuning; default 3): bit 0 = the 256 x 192 ping-pong kernels for launches with
// at least one 256-row tile per CU, bit 1 = also for the small-map 3x3 convs that would otherwise run position-major on
// the 128 x 192 kernel (they then run in standard row order, all 9 taps, on the tap-resident kernel: 239 vs 257 us for
// 1536 -> 1536 at 4 x 4 x 512 although it executes the 31 % of MACs the position-major order skips), bit 2 = those run position-major on the
// ping-pong kernel (needs nhyp % 256 == 0), bit 3 = no minimum tile count (tests: small shapes on the ping-pong kernel),
// bit 4 = 3x3 convs per tap on the ping-pong kernel instead of the tap-resident (halo) kernel.
struct ConvPlan { bool dma, pp, posmajor, halo; int small; int hsplit; bool stream; };      // hsplit > 1: tap-resident kernel with that many K splits      // small: -1, or the tile of conv_gemm_small_kernel (0 = 64 x 64, 1 = 128 x 128, 2 = 64 x 64 / 6-stage ring, 3 = 64 x 64 by two K groups)

// Launches that cannot give every CU a 128 x 192 tile take the small-tile kernel (kernels_gemm_small.hip): fewer than
// NOPE_SMALL_MAX_TILES (default 320) tiles of 128 x 192.  NOPE_CONV_SMALL: 0 = never, 1 = that policy (default), 2 = whenever the
// kernel applies (tests); NOPE_SMALL_TILE forces the tile.
static int plan_small(int dt, const ConvArgs& a, bool dma) {
    dt = dt_base(dt);      // (NOPE_F16X2 plans as NOPE_BF16X3: same storage, same tiles)
    const int mode_env = NOPE_ENV("NOPE_CONV_SMALL", 1);
    if (!dma || mode_env == 0 || a.mode == NOPE_CONV_UP2 || a.ntaps == 16 || a.force_generic) return -1;
    if (a.out_nchw && a.pn_ms) return -1;      // (the small-tile kernel's NCHW epilogue has no fused PreNorm: the 128 x 192 kernel's generic epilogue does)
    if (mode_env == 1 && (NOPE_ENV("NOPE_CONV_PP", 0) & 8)) return -1;      // bit 3 = "ping-pong kernels at ANY tile count" (their tests)
    const bool phased = a.mode == NOPE_CONV_UP2P;
    const long long M = (long long)a.nhyp * (phased ? a.Hs * a.Ws : a.Ho * a.Wo);
    const long long tiles128 = (long long)cdiv((int)M, BM) * cdiv(a.Cout, BN) * (phased ? 4 : 1);
    const int max_tiles = NOPE_ENV("NOPE_SMALL_MAX_TILES", 320);      // (tuning sweeps toggle it: nope_tuning_reload)
    // Short-K 1x1 convs of ANY size (NOPE_SMALL_1X1_MAXK = K steps, default 0 = off): their 128 x 192 launches are bound by the epilogue
    // of a three-K-step tile, not by HBM (192 -> 384 at 32 x 32 x 512: 604 MB in 221 us = 2.7 TB/s); the 128 x 128 small tile has a one-pass
    // epilogue.
    const int maxk_1x1 = NOPE_ENV("NOPE_SMALL_1X1_MAXK", 0);
    if (mode_env == 1 && a.mode == NOPE_CONV_PLAIN && a.ntaps == 1 && !a.out_nchw && (a.C1 + a.C2) / (8 * dt_vec(dt)) <= maxk_1x1 && tiles128 >= max_tiles)
        return NOPE_ENV("NOPE_SMALL_TILE", 1);
    if (mode_env == 1 && tiles128 >= max_tiles) return -1;
    if (mode_env == 1 && dt != NOPE_F32 && a.mode == NOPE_CONV_PLAIN && a.ntaps == 9 && a.Ws <= conv_halo_max_width() && a.rep1 == 1 && !a.out_nchw &&
        !a.pn_ms && a.Cout % dt_vec(dt) == 0 && 9 * ((a.C1 + a.C2) / (8 * dt_vec(dt))) >= 54 && (long long)cdiv((int)M, 256) * cdiv(a.Cout, BN) >= 128)
        return -1;       // the tap-resident kernel has its 128 tiles of 256 rows (measured at 16 x 16 x 64: 60.6 / 83 us against 83 / 116 us on 64 x 64 tiles)
    if (NOPE_ENV_SET("NOPE_SMALL_TILE")) { const int t = NOPE_ENV("NOPE_SMALL_TILE", 0); return t < 0 || t > 3 ? 0 : t; }
    const long long tiles64 = (long long)cdiv((int)M, 64) * cdiv(a.Cout, 64) * (phased ? 4 : 1);
    if (tiles64 > 1536) return 1;
    // (tile 2, the 6-stage ring, for launches of at most NOPE_SMALL_DEEP_MAX tiles with a K loop of >= 8 steps: with 512 -- launches that
    //  the 48 KiB ring runs two workgroups per CU -- measured SLOWER than the 3-stage ring, 26 / 64 templates 4.16 / 4.95 ms against
    //  4.04 / 4.81, profiles/r04e_small_bank_sweep.txt; off by default)
    const int nk = a.ntaps * ((a.C1 + a.C2) / (8 * dt_vec(dt)));
    const int deep_max = NOPE_ENV("NOPE_SMALL_DEEP_MAX", 0);
    if (tiles64 <= deep_max && nk >= 8) return 2;
    // at most one workgroup per CU and a long K: two wave groups per tile on alternate K steps (tile 3)
    const int kg2_max = NOPE_ENV("NOPE_SMALL_KG2_MAX", 256);
    return (tiles64 <= kg2_max && nk >= 12) ? 3 : 0;
}

// 3x3 convs with FEW 256 x 192 tiles and a LONG K (the 4 x 4 / 8 x 8 levels of the U-Net at a few dozen pose hypotheses: 16-64 tiles
// of 100-200 K steps) run on the tap-resident kernel with the channel chunks split over blockIdx.z (f32 partials + the fixed-order
// reduce): its 256-row tiles move a third of the L2 -> LDS bytes per flop of the 128 x 192 kernel, which is what bounds these
// launches (profiles/r04b: 51 us for 1536 -> 1536 at 4 x 4 x 64 on the 128 x 192 kernel split 8 ways = 1.9 us per K step, 111 us on 64 x 64
// tiles).  Returns the number of splits (1: does not apply).  NOPE_HALO_SPLIT=0 turns it off.
static int halo_split_factor(int dt, const ConvArgs& a) {
    dt = dt_base(dt);      // (NOPE_F16X2 plans as NOPE_BF16X3: same storage, same tiles)
    if (NOPE_ENV("NOPE_HALO_SPLIT", -1) == 0) return 1;
    const int pp_mode = NOPE_ENV("NOPE_CONV_PP", (dt != NOPE_F32 ? 3 : 0));
    if (!(pp_mode & 1) || (pp_mode & 16)) return 1;
    const int vec = dt_vec(dt), bk = 8 * vec, Cin = a.C1 + a.C2;
    if (a.mode != NOPE_CONV_PLAIN || a.ntaps != 9 || a.Ws > conv_halo_max_width() || a.rep1 != 1 || a.out_nchw || a.pn_ms ||
        a.force_generic || a.Cout % vec || Cin % bk || (a.C2 && a.C1 % bk)) return 1;
    const long long M = (long long)a.nhyp * a.Ho * a.Wo;
    const int nchunks = Cin / bk;
    const long long tiles = (long long)cdiv((int)M, 256) * cdiv(a.Cout, BN);
    const int min_chunks = NOPE_ENV("NOPE_HALO_SPLIT_MIN_CHUNKS", 12);
    // 128 tiles split in two fill the 256 CUs in one round (256 hypotheses at the 4 x 4 level: 11.35 -> 10.76 ms per step, run t); above that a
    // split needs a second round of workgroups and loses (176 tiles, 341 hypotheses: 14.8 -> 15.3 ms)
    const int max_tiles = NOPE_ENV("NOPE_HALO_SPLIT_MAX_TILES", 128);
    if (tiles > max_tiles || nchunks < min_chunks) return 1;
    // as many splits as fit ONE round of 256 workgroups (one per CU: 158 KiB of LDS each): 88 tiles x 3 = 264 would run a second
    // round for 8 of them
    int S = (int)(256 / tiles);
    if (S > nchunks) S = nchunks;
    if (S > 16) S = 16;
    return S < 2 ? 1 : S;
}

// workgroups of a streaming launch: one per CU (NOPE_STREAM_GRID: the tests walk small grids; a multiple of 8)
static int stream_grid() { const int g = NOPE_ENV("NOPE_STREAM_GRID", 256); return g >= 8 && g % 8 == 0 ? g : 256; }

static ConvPlan plan_conv(int dt, const ConvArgs& a) {
    dt = dt_base(dt);      // (NOPE_F16X2 plans as NOPE_BF16X3: same storage, same tiles)
    // (read per launch: the tests toggle it.  f32 -- the parity mode -- stays on the 128 x 192 kernel unless asked: its MFMA phase is
    //  16x longer per K step, loads were never its bound, and two workgroups per CU beat one: 126 vs 135 ms per 512-template step)
    const int pp_mode = NOPE_ENV("NOPE_CONV_PP", (dt != NOPE_F32 ? 3 : 0));
    const int variant = NOPE_ENV("NOPE_CONV_VARIANT", 0);
    ConvPlan pl{false, false, false, false, -1, 1, false};
    const int vec = dt_vec(dt), es = dt_es(dt), bk = 8 * vec;
    const int Cin = a.C1 + a.C2;
    const unsigned long long lim = 0x7fffffffULL;
    const bool phased = a.mode == NOPE_CONV_UP2P;
    const unsigned long long b1 = (unsigned long long)cdiv(a.nhyp, a.rep1) * a.Hs * a.Ws * a.C1 * es;
    const unsigned long long b2 = a.C2 ? (unsigned long long)cdiv(a.nhyp, a.rep2) * a.Hs * a.Ws * a.C2 * es : 0;
    const unsigned long long bw = (unsigned long long)a.Cout * a.ntaps * Cin * es;
    // LDS-DMA kernels: a K step (128 B of channels) never straddles sources and 32-bit offsets suffice
    // (the 4x4 stride-2 conv of the non-default soft downsampling runs on the generic kernel: its tap geometry is not a 3x3 mask)
    pl.dma = !a.force_generic && a.ntaps != 16 && Cin % bk == 0 && (a.C2 == 0 || (a.C1 % bk == 0 && a.mode == NOPE_CONV_PLAIN)) && b1 < lim && b2 < lim && bw < lim;
    if (!pl.dma || a.geglu) return pl;          // (GEGLU epilogue: an instantiation of the 128 x 192 kernel only)
    if (a.splitk_ws) {
        const int hs = halo_split_factor(dt, a);
        const long long Mr = (long long)a.nhyp * a.Ho * a.Wo;
        if (hs > 1 && (size_t)hs * (size_t)Mr * a.Cout * 4 <= a.splitk_bytes) { pl.pp = pl.halo = true; pl.hsplit = hs; return pl; }
    }
    pl.small = plan_small(dt, a, pl.dma);
    if (pl.small >= 0) return pl;
    const bool small3x3 = a.mode == NOPE_CONV_PLAIN && a.ntaps == 9 && !a.colstats && !a.pn_ms && !a.out_nchw && !a.splitk_ws &&
                          a.Hs * a.Ws <= POSMAJOR_MAX_HW;
    const bool posmajor128 = small3x3 && a.nhyp % BM == 0 && !(variant & 8) && variant != 4;
    const long long M = (long long)a.nhyp * (phased ? a.Hs * a.Ws : a.Ho * a.Wo);
    // (short K loops -- the 1x1 convs around the attention blocks, 2-3 K steps per tile -- stay on the 128 x 192 kernel, whose two
    //  workgroups per CU cover each other's prologue and epilogue: measured 193 vs 234 us for 192 -> 384 at 32 x 32)
    // (long 3x3 launches take the tap-resident kernel from 128 tiles on -- NOPE_HALO_MIN_TILES: the 768 -> 768 convs of the 4 x 4
    //  level at 512 hypotheses have 128 tiles of 108 K steps; on half the CUs they still beat the position-major 128 x 192 launch,
    //  one workgroup per CU: 20.26 -> 20.19 ms per step, profiles/r03d_defaults_ab.txt)
    const int halo_min_tiles = NOPE_ENV("NOPE_HALO_MIN_TILES", 128);
    const long long min_tiles = (a.mode == NOPE_CONV_PLAIN && a.ntaps == 9 && a.ntaps * (Cin / bk) >= 54) ? halo_min_tiles : 256;
    const bool pp_shape = (a.mode == NOPE_CONV_PLAIN || a.mode == NOPE_CONV_DOWN2 || phased) && !a.out_nchw && a.Cout % vec == 0 &&
                          ((pp_mode & 8) || a.ntaps * (Cin / bk) >= 12) &&
                          ((pp_mode & 8) || (long long)cdiv((int)M, 256) * cdiv(a.Cout, BN) * (phased ? 4 : 1) >= min_tiles) && variant == 0;
    if (pp_shape && (pp_mode & 1)) {
        if (!posmajor128) pl.pp = true;
        else if ((pp_mode & 4) && a.nhyp % 256 == 0) { pl.pp = true; pl.posmajor = true; }
        else if (pp_mode & 2) pl.pp = true;
    }
    if (!pl.pp) pl.posmajor = posmajor128;
    // 3x3 convs in standard row order on the ping-pong schedule keep their A operand in LDS across the 9 taps (bit 4 = off)
    // (the first source must not be broadcast: its A offsets are linear in the flat pixel index; a broadcast second source is fine)
    pl.halo = pl.pp && !pl.posmajor && a.mode == NOPE_CONV_PLAIN && a.ntaps == 9 && a.Ws <= conv_halo_max_width() && a.rep1 == 1 &&
              !(pp_mode & 16);
    // ... and so do the four phase convs of an up-sampling (conv3x3_halo_kernel<..., UP>: the stage is the 3 x 3 neighbourhood the phases' 2 x 2 taps
    // come from) under the two-pass tile, whose per-tap launches are bound by staging 256 f32 rows per tap and splitting them in registers at every
    // read: 512-hypothesis launches 0.77 / 0.72 / 0.67 -> ~0.59 ms, step -1.2 % (profiles/r06w_up2p_tap_resident_ab.txt).  As bf16x3 (three MFMA
    // passes per step: the per-tap kernel's LOAD phase is hidden) it is +-2 % per launch: per tap stays.  NOPE_UP2P_HALO: 0 = per tap, 2 = bf16x3 too.
    {
        const int uh = NOPE_ENV("NOPE_UP2P_HALO", 1);
        const bool x2_layer = a.w_x2 && !a.pn_ms && !a.geglu && Cin % 32 == 0;
        if (pl.pp && phased && !pl.posmajor && a.ntaps == 4 && dt == NOPE_BF16X3 && a.C2 == 0 && a.Ws <= 30 && a.rep1 == 1 && !(pp_mode & 16) &&      // (Ws <= 30: at most five A pieces per wave)
            (uh == 2 || (uh == 1 && x2_layer)))
            pl.halo = true;
    }
    // The streaming 1x1 kernel (kernels_gemm_stream.hip: one persistent workgroup per CU, the activations through a five-stage ring that runs
    // across tile boundaries) for 1x1 convs with thousands of 128 x 192 tiles.  OFF by default: built on the premise that these HBM-bound
    // launches wait for memory round trips, measured 20-25 % SLOWER than the 128 x 192 kernel they would leave (profiles/r06h_stream_bench_*.txt,
    // r06l_*): what bounds them is the epilogue -- its instruction stream (the lean form of epilogue_wide took 12 % off) and, on gfx9, a wave's
    // stores draining through the same in-order vmcnt as its loads (r06i ablations: stores alone 166 us + MFMA 100 us + loads 96 us ~ the
    // launch's 375 us) -- and two workgroups per CU overlap that better than one.  NOPE_CONV_STREAM: bit 0 = the launches the 128 x 192 kernel
    // would take, bit 1 = also the long 1x1 convs of the per-tap ping-pong kernel; NOPE_STREAM_MIN_ITERS = tiles per workgroup from which on.
    {
        const int sm = NOPE_ENV("NOPE_CONV_STREAM", 0);
        const long long tm = M / BM, tn = cdiv(a.Cout, BN);
        if (sm && dt != NOPE_F32 && a.mode == NOPE_CONV_PLAIN && a.ntaps == 1 && !pl.posmajor && !a.out_nchw && a.Cout % vec == 0 && a.w &&
            a.rep1 == 1 && a.rep2 == 1 && variant == 0 && (!pl.pp || (sm & 2)) && M % BM == 0 && (tn == 1 || tn == 2 || tn == 4 || tn == 8) &&
            tm % 8 == 0 && (tm * tn) % stream_grid() == 0 && (tm * tn) / stream_grid() >= NOPE_ENV("NOPE_STREAM_MIN_ITERS", 2)) {
            pl.stream = true; pl.pp = false; pl.halo = false;
        }
    }
    return pl;
}

// ConvArgs::geglu: a plain 1x1 conv on the 128 x 192 LDS-DMA kernel's wide epilogue (no residual / statistics / PreNorm / activation / split):
// 16-bit storage on the packed path, column pairs whole inside a lane's 8-column chunk and 8-byte output rows; f32 storage (round 6; NOPE_GEGLU_FUSED_F32=0:
// A/B switch) in the generic row loop, two pairs per 4-column chunk
static bool geglu_shape_ok(int dt, const ConvArgs& a, const ConvPlan& pl) {
    const int variant = NOPE_ENV("NOPE_CONV_VARIANT", 0);
    return (dt_es(dt) == 2 || NOPE_ENV("NOPE_GEGLU_FUSED_F32", 1)) && a.mode == NOPE_CONV_PLAIN && a.ntaps == 1 && !a.resid && !a.colstats && !a.pn_ms && !a.out_nchw && !a.act && !a.splitk_ws &&
           a.Cout % 16 == 0 && pl.dma && !pl.pp && pl.small < 0 && !pl.posmajor && variant == 0;
}
bool conv_geglu_fusable(int dt, const ConvArgs& a0) {
    const int mode = NOPE_ENV("NOPE_GEGLU_FUSED", 1);      // (A/B switch; 2: only launches the 128 x 192 kernel would get anyway)
    if (mode == 0) return false;
    if (mode == 2) { const ConvPlan q = plan_conv(dt, a0); if (q.pp || q.small >= 0) return false; }
    ConvArgs a = a0;
    a.geglu = 1;
    return geglu_shape_ok(dt, a, plan_conv(dt, a));
}

// Would launch_conv run this conv in position-major row order?
bool conv_is_posmajor(int dt, const ConvArgs& a) { return plan_conv(dt, a).posmajor; }

// Fused GroupNorm column statistics, rows per block (0: this conv cannot emit them): 64 wherever samples are whole 64-row blocks; 16 on
// 16-pixel maps (the 4 x 4 level: per-sample blocks, every LDS-DMA kernel's wide epilogue, the small-tile kernel, the split-K reduce) -- which
// removes that level's gn_stats passes; 32 on 32-pixel maps from the small-tile kernel and the split-K reduce.
int conv_stat_rows(int dt, const ConvArgs& a) {
    dt = dt_base(dt);      // (NOPE_F16X2 plans as NOPE_BF16X3: same storage, same tiles)
    const int vec = dt_vec(dt);
    if (a.mode == NOPE_CONV_UP2P || a.resid || a.out_nchw || a.Cout % vec || a.Cout > 2048) return 0;
    const long long HW = (long long)a.Ho * a.Wo, M = (long long)a.nhyp * HW;
    if (HW % 64 == 0) return 64;
    // 16-pixel maps (the 4 x 4 level): per-sample blocks of 16 rows -- every kernel with the wide epilogue, the small-tile kernel and the
    // split-K reduce emit them; 32-pixel maps only the latter two
    ConvArgs b = a;
    b.colstats = nullptr;
    const bool small_or_split = halo_split_factor(dt, b) > 1 || plan_conv(dt, b).small >= 0;
    const bool wide16 = (NOPE_ENV("NOPE_STATS16", -1) != 0);      // (A/B switch: 0 = the 128 x 192 / ping-pong kernels leave 16-pixel maps to gn_stats)
    if (HW == 16 && M % 16 == 0 && (small_or_split || (wide16 && plan_conv(dt, b).dma))) return 16;
    if (HW == 32 && M % 32 == 0 && small_or_split) return 32;
    return 0;
}

// the f16 + MX-fp8 tile: launches of a layer that carries the second pack on a ping-pong kernel (tap-resident or per-tap) or -- round 6 -- on the
// small-tile kernel (reference-sized banks, an 8-way shard).  NOPE_X2_PP=0: tap-resident only.  NOPE_X2_SMALL=1: also on the small-tile kernel
// -- built, bit-identical to the ping-pong kernels, measured SLOWER than its bf16x3 form there (26 / 64 / 91 templates 7.29 / 9.13 / 11.47 ms
// against 6.52 / 8.59 / 11.05, same box, profiles/r06c_small_tile_x2_ab.txt: a 64 x 64 tile's wave holds ONE accumulator, so its three MFMAs per
// K step are a dependent chain either way and the register split costs more VALU than the third pass costs matrix time): off by default
static bool plan_takes_x2(const ConvArgs& a, const ConvPlan& plan) {
    if (!a.w_x2 || a.pn_ms || a.geglu || (a.C1 + a.C2) % 32) return false;
    if (plan.small >= 0) return NOPE_ENV("NOPE_X2_SMALL", 0) != 0;
    return plan.pp && (plan.halo || NOPE_ENV("NOPE_X2_PP", 1) != 0);
}
// the wide NHWC epilogue (epilogue_wide) of a 4-byte element type, whole launch in one pass: the 128 x 192 LDS-DMA kernel and the ping-pong kernels
static bool plan_records_out_amax(int dt, const ConvArgs& a, const ConvPlan& plan) {
    return dt_es(dt_base(dt)) == 4 && plan.dma && plan.small < 0 && plan.hsplit <= 1 && !a.out_nchw && !a.geglu && a.Cout % 4 == 0 &&
           (plan.pp || !a.splitk_ws || conv_splitk_factor(dt, a) <= 1);
}
bool conv_records_out_amax(int dt, const ConvArgs& a) { return a.out_amax && plan_records_out_amax(dt, a, plan_conv(dt, a)); }
bool conv_takes_x2(int dt, const ConvArgs& a) { return dt_base(dt) == NOPE_BF16X3 && plan_takes_x2(a, plan_conv(dt, a)); }

int conv_kernel_kind(int dt, const ConvArgs& a) {
    const ConvPlan pl = plan_conv(dt, a);
    return pl.small >= 0 ? NOPE_CONV_KERNEL_SMALL : pl.stream ? NOPE_CONV_KERNEL_STREAM : pl.halo ? NOPE_CONV_KERNEL_HALO256 : pl.pp ? NOPE_CONV_KERNEL_PP256 : pl.dma ? NOPE_CONV_KERNEL_DMA128 : NOPE_CONV_KERNEL_GENERIC;
}

// Multiply-adds x2 the launch actually executes (position-major launches skip the taps that lie in the padding:
// (3H-2)(3W-2) of the 9 H W tap instances remain).
double conv_executed_flops(int dt, const ConvArgs& a) {
    dt = dt_base(dt);      // (NOPE_F16X2 plans as NOPE_BF16X3: same storage, same tiles)
    double taps = (double)a.ntaps;
    if (conv_is_posmajor(dt, a)) taps = (double)(3 * a.Hs - 2) * (3 * a.Ws - 2) / ((double)a.Hs * a.Ws);
    return 2.0 * (double)a.nhyp * a.Ho * a.Wo * a.Cout * taps * (a.C1 + a.C2);
}

int conv_splitk_factor(int dt, const ConvArgs& a) {
    dt = dt_base(dt);      // (NOPE_F16X2 plans as NOPE_BF16X3: same storage, same tiles)
    {
        const int hs = halo_split_factor(dt, a);      // long 3x3 convs with few tiles: split-K on the tap-resident kernel (its reduce kernel
        if (hs > 1) return hs;                        // also serves a colstats request)
    }
    if (a.colstats || a.pn_ms || a.mode == NOPE_CONV_UP2P || a.mode == NOPE_CONV_UP2) return 1;
    const int vec = dt_vec(dt);
    const int Cin = a.C1 + a.C2;
    if (Cin % (8 * vec)) return 1;
    if (plan_conv(dt, a).small >= 0) return 1;        // the small-tile kernel has enough workgroups without splitting K
    const long long M = (long long)a.nhyp * a.Ho * a.Wo;
    const long long tiles = (long long)cdiv((int)M, BM) * cdiv(a.Cout, BN);
    const int nk = a.ntaps * (Cin / (8 * vec));
    if (tiles > 128 || nk < 8) return 1;
    int S = (int)((512 + tiles - 1) / tiles);
    if (S > nk / 4) S = nk / 4;
    if (S > 16) S = 16;
    return S < 2 ? 1 : S;
}

int launch_conv(int dt, const ConvArgs& a, hipStream_t s) {
    if (dt == NOPE_F16X2) {      // as an element type (nope_op_conv): `w` is in the NOPE_F16X2 layout, which only the ping-pong kernels read
        if (a.w_x2) return NOPE_ERR_ARG;
        ConvArgs b = a;
        b.w_x2 = a.w; b.w = nullptr;
        return launch_conv(NOPE_BF16X3, b, s);
    }
    if (a.w_x2 && dt != NOPE_BF16X3) return NOPE_ERR_ARG;
    if (!a.src1 || (!a.w && !a.w_x2) || !a.out || a.C1 <= 0 || a.Cout <= 0 || a.nhyp <= 0) return NOPE_ERR_ARG;
    if (a.C2 > 0 && !a.src2) return NOPE_ERR_ARG;
    const int vec = dt_vec(dt);
    if (!dt_is_compute(dt)) return NOPE_ERR_UNSUPPORTED;
    if (a.C1 % vec || a.C2 % vec) return NOPE_ERR_UNSUPPORTED;
    if (dt == NOPE_BF16X3 && (a.C1 % 8 || a.C2 % 8)) return NOPE_ERR_UNSUPPORTED;     // (hi, lo) weight groups hold 8 channels
    if (a.rep1 < 1 || a.rep2 < 1) return NOPE_ERR_ARG;
    if (a.mode == NOPE_CONV_PLAIN) {
        if ((a.ntaps != 1 && a.ntaps != 9) || a.Hs != a.Ho || a.Ws != a.Wo) return NOPE_ERR_ARG;
    } else if (a.mode == NOPE_CONV_UP2) {
        if (a.ntaps != 9 || a.Ho != 2 * a.Hs || a.Wo != 2 * a.Ws) return NOPE_ERR_ARG;
    } else if (a.mode == NOPE_CONV_DOWN2) {
        if (a.ntaps != 4 || a.Hs != 2 * a.Ho || a.Ws != 2 * a.Wo) return NOPE_ERR_ARG;
    } else if (a.mode == NOPE_CONV_UP2P) {
        if (a.ntaps != 4 || a.Ho != 2 * a.Hs || a.Wo != 2 * a.Ws || a.C2 != 0 || a.out_nchw) return NOPE_ERR_ARG;
    } else if (a.mode == NOPE_CONV_STRIDE2) {
        if ((a.ntaps != 1 && a.ntaps != 9 && a.ntaps != 16) || a.Hs != 2 * a.Ho || a.Ws != 2 * a.Wo || a.C2 != 0) return NOPE_ERR_ARG;
    } else return NOPE_ERR_ARG;
    const bool phased = a.mode == NOPE_CONV_UP2P;
    const long long M = phased ? (long long)a.nhyp * a.Hs * a.Ws : (long long)a.nhyp * a.Ho * a.Wo;
    if (M > 0x7fffffffLL) return NOPE_ERR_UNSUPPORTED;

    ConvParams p;
    p.src1 = (const unsigned char*)a.src1; p.src2 = (const unsigned char*)a.src2;
    p.C1 = a.C1; p.C2 = a.C2; p.rep1 = a.rep1; p.rep2 = a.rep2;
    p.Hs = a.Hs; p.Ws = a.Ws; p.Ho = a.Ho; p.Wo = a.Wo;
    p.Hm = phased ? a.Hs : a.Ho; p.Wm = phased ? a.Ws : a.Wo;
    p.mode = a.mode; p.ntaps = a.ntaps;
    p.w = (const unsigned char*)a.w; p.bias = a.bias; p.resid = (const unsigned char*)a.resid;
    p.out = (unsigned char*)a.out; p.Cout = a.Cout; p.M = (int)M;
    p.out_nchw = a.out_nchw; p.out_dt = a.out_dt;
    p.act = a.act;
    p.wide_out = (!a.out_nchw && a.Cout % vec == 0) ? 1 : 0;
    p.nchw_staged = (a.out_nchw && !a.resid && !a.pn_ms && M % 64 == 0 && ((long long)a.Ho * a.Wo) % 64 == 0 &&
                     (NOPE_ENV("NOPE_NCHW_STAGED", -1) != 0)) ? 1 : 0;
    p.colstats = a.colstats;
    p.stat_rows = a.stat_rows;
    p.pn_ms = a.pn_ms; p.pn_c0 = a.pn_c0; p.pn_c1 = a.pn_c1;
    if (a.pn_ms && (!a.pn_c0 || !a.pn_c1 || a.mode != NOPE_CONV_PLAIN || a.ntaps != 1 || a.colstats)) return NOPE_ERR_ARG;
    if (a.colstats && (!p.wide_out || phased || a.resid || a.act || a.stat_rows != conv_stat_rows(dt, a))) return NOPE_ERR_ARG;      // (act: the epilogues take the
                                                                                                      // statistics before an activation, the split-K reduce after it -- nobody needs the pair)
    const int es = dt_es(dt);
    const int Cin = a.C1 + a.C2;
    const unsigned long long b1 = (unsigned long long)cdiv(a.nhyp, a.rep1) * a.Hs * a.Ws * a.C1 * es;
    const unsigned long long b2 = a.C2 ? (unsigned long long)cdiv(a.nhyp, a.rep2) * a.Hs * a.Ws * a.C2 * es : 0;
    const unsigned long long bw = (unsigned long long)a.Cout * a.ntaps * Cin * es;     // one phase's weights
    p.w_phase_bytes = phased ? (unsigned)bw : 0u;
    const ConvPlan plan = plan_conv(dt, a);
    const bool dma = plan.dma;
    const bool x2 = plan_takes_x2(a, plan);
    if (!x2 && !a.w) return NOPE_ERR_UNSUPPORTED;               // (NOPE_F16X2 as an element type on a shape the ping-pong kernels do not take)
    p.x2_scale = nullptr; p.x2_amax = x2 ? a.x2_amax : nullptr; p.x2_t_zero = x2 ? a.x2_t_zero : 0;
    p.out_amax = (a.out_amax && plan_records_out_amax(dt, a, plan)) ? a.out_amax : nullptr;
    if (x2) { p.w = (const unsigned char*)a.w_x2; p.x2_scale = reinterpret_cast<const int*>(p.w + bw * (phased ? 4 : 1)); }
    if (a.geglu && !geglu_shape_ok(dt, a, plan)) return NOPE_ERR_UNSUPPORTED;
    p.geglu = a.geglu;
    if (a.colstats && a.stat_rows == 32 && plan.small < 0 && plan.hsplit <= 1) return NOPE_ERR_ARG;   // 32-row blocks: small-tile kernel or split-K reduce only
    if (a.colstats && a.stat_rows == 16 && plan.small < 0 && plan.hsplit <= 1 && (!dma || plan.posmajor)) return NOPE_ERR_ARG;
    p.bytes1 = (unsigned)(dma ? b1 : 0); p.bytes2 = (unsigned)(dma ? b2 : 0); p.bytesw = (unsigned)(dma ? bw : 0);
    // NOPE_CONV_VARIANT=4 selects the 256x192 / 8-wave tile (measured on par with the default 128x192 / 4-wave
    // tile in round 1; kept for tuning, see DESIGN.md section 4 for the other variants that were tried).
    const int variant = NOPE_ENV("NOPE_CONV_VARIANT", 0);
    p.variant = variant;
    p.d_hw = make_fastdiv((unsigned)(p.Hm * p.Wm)); p.d_w = make_fastdiv((unsigned)p.Wm);
    p.d_rep1 = make_fastdiv((unsigned)p.rep1); p.d_rep2 = make_fastdiv((unsigned)p.rep2);
    const int bm = plan.small >= 0 ? (plan.small == 1 ? 128 : 64)
                   : (plan.pp || (dma && variant == 4 && M >= 256 * 256 && (dt == NOPE_F32 || dt == NOPE_BF16))) ? 256 : BM;
    const int bn = plan.small >= 0 ? (plan.small == 1 ? 128 : 64) : BN;
    p.tiles_m = cdiv((int)M, bm); p.tiles_n = cdiv(a.Cout, bn);
    const int tn = p.tiles_n;
    p.xcd_map = plan.small < 0 && (tn == 1 || tn == 2 || tn == 4 || tn == 8) && (p.tiles_m % (8 / tn) == 0) ? 1 : 0;
    const long long nblocks = (long long)p.tiles_m * p.tiles_n;
    if (nblocks > 0x7fffffffLL) return NOPE_ERR_UNSUPPORTED;
    p.splits = 1; p.split_out = nullptr;
    // Position-major row order for small images: with 4x4 / 8x8 maps 31 % / 16 % of the 3x3 taps fall into the padding;
    // grouping the rows of a tile by pixel position makes those taps invalid for whole tiles, whose K steps then vanish.
    p.nhyp = a.nhyp; p.d_n = make_fastdiv((unsigned)a.nhyp);
    p.posmajor = plan.posmajor ? 1 : 0;
    if (p.posmajor) {       // positions sorted by descending valid-tap count (stable)
        int cnt[64], n = a.Hs * a.Ws;
        for (int i = 0; i < n; ++i) {
            const int y = i / a.Ws, x = i % a.Ws;
            cnt[i] = ((y > 0) + 1 + (y + 1 < a.Hs)) * ((x > 0) + 1 + (x + 1 < a.Ws));
            p.pos_order[i] = (unsigned char)i;
        }
        for (int i = 1; i < n; ++i)      // insertion sort, n <= 64
            for (int j = i; j > 0 && cnt[p.pos_order[j]] > cnt[p.pos_order[j - 1]]; --j) {
                const unsigned char tmp = p.pos_order[j]; p.pos_order[j] = p.pos_order[j - 1]; p.pos_order[j - 1] = tmp;
            }
    }
    if (plan.hsplit > 1) {
        p.splits = plan.hsplit;
        p.split_out = (float*)a.splitk_ws;
    } else if (dma && !plan.pp && a.splitk_ws) {
        p.splits = conv_splitk_factor(dt, a);
        if ((size_t)p.splits * (size_t)M * a.Cout * 4 > a.splitk_bytes) p.splits = 1;
        if (p.splits > 1) p.split_out = (float*)a.splitk_ws;
    }
    // LDS-DMA launches with several weight panels: split the panels over gn XCD columns and the M tiles over 8 / gn XCD rows
    // so that the bytes crossing the fabric, gn x activations + (8 / gn) x weights, are fewest (tile_coords, map 2).
    // NOPE_XCD_MAP=1 keeps one panel per XCD (gn = tiles_n).
    p.xcd_gn = tn;
    if (plan.small >= 0) {
        // small tiles: an (8 / gn) x gn XCD grid over (runs of M tiles) x (groups of weight panels), gn chosen like below; at small
        // M the weights are most of the bytes, so they usually cross the fabric once (gn = 8) and the activations 8 times
        const double abytes = (double)b1 + (double)b2, wbytes = (double)bw * (phased ? 4 : 1);
        double best = -1.0;
        for (int gn = 1; gn <= 8; gn *= 2)
            if (p.tiles_m % (8 / gn) == 0 && p.tiles_n % gn == 0 && (best < 0 || gn * abytes + (8 / gn) * wbytes < best)) {
                best = gn * abytes + (8 / gn) * wbytes; p.xcd_gn = gn; p.xcd_map = 3;
            }
    }
    if (dma && p.xcd_map && p.xcd_map != 3 && tn > 1 && (NOPE_ENV("NOPE_XCD_MAP", -1) != 1)) {
        const double abytes = (double)b1 + (double)b2, wbytes = (double)bw * (phased ? 4 : 1);
        double best = tn * abytes + (8 / tn) * wbytes;
        for (int gn = 1; gn < tn; gn *= 2)
            if (p.tiles_m % (8 / gn) == 0 && gn * abytes + (8 / gn) * wbytes < best) { best = gn * abytes + (8 / gn) * wbytes; p.xcd_gn = gn; }
        if (NOPE_ENV_SET("NOPE_XCD_GN")) {            // tests: force the split
            const int gn = NOPE_ENV("NOPE_XCD_GN", 0);
            if (gn >= 1 && gn <= tn && (gn & (gn - 1)) == 0 && p.tiles_m % (8 / gn) == 0) p.xcd_gn = gn;
        }
        if (p.xcd_gn != tn) p.xcd_map = 2;
    }
    // Panel counts outside {1, 2, 4, 8} (no launch of the default U-Net; the LDM variant's linears): map 4 of tile_coords.  NOPE_XCD_ANY=0: off.
    if (!p.xcd_map && plan.small < 0 && tn > 1 && p.tiles_m % 8 == 0 && (NOPE_ENV("NOPE_XCD_ANY", -1) != 0)) p.xcd_map = 4;
    // Persistent walk: 512 workgroups (2 per CU), each `iters` tiles 64 / span tile_m apart (same XCD, same weight panel; span =
    // panels an XCD interleaves under map 2).
    p.persist_iters = 1; p.persist_d1 = p.persist_d2 = 0; p.persist_dm = 0; p.timeline = nullptr;
    unsigned gx = (unsigned)nblocks;
    {
        const long long hw = (long long)a.Hs * a.Ws;
        const int persist_on = NOPE_ENV("NOPE_CONV_PERSIST", 1);
        const int span = p.xcd_map == 2 ? tn / p.xcd_gn : 1;
        const int pg = NOPE_ENV("NOPE_PERSIST_GRID", 512) == 256 ? 256 : 512;      // (tuning: 256 = one workgroup per CU)
        if (persist_on && dma && plan.small < 0 && bm == BM && dt != NOPE_F32 && a.mode == NOPE_CONV_PLAIN && !p.posmajor && p.splits == 1 && p.xcd_map && p.xcd_map != 4 &&
            p.wide_out && a.rep1 == 1 && a.rep2 == 1 && M % BM == 0 && nblocks > pg && nblocks % pg == 0 && (pg / 8) % span == 0 &&
            ((long long)(pg / 8 / span) * BM) % hw == 0 && !(variant & 2)) {
            p.persist_iters = (int)(nblocks / pg);
            p.persist_dm = (pg / 8 / span) * BM;
            p.persist_d1 = (unsigned)((long long)p.persist_dm * a.C1 * es);
            p.persist_d2 = (unsigned)((long long)p.persist_dm * a.C2 * es);
            gx = (unsigned)pg;
        }
    }
    // The tap-resident kernel walks tiles too (bf16): one workgroup per CU, tiles gx / 8 apart inside the XCD's run of M tiles,
    // the next tile's prologue in flight under the epilogue.  NOPE_HALO_PERSIST = workgroups (default 256, 0 = one tile per
    // workgroup; the tests use small grids).
    if (plan.halo && !phased && dt != NOPE_F32 && p.xcd_map && p.xcd_map != 4 && p.splits == 1) {      // (the split-K instantiation returns after its first tile; the phase-conv form walks no tiles)
        const int want = NOPE_ENV("NOPE_HALO_PERSIST", 256);
        const long long hw = (long long)a.Hs * a.Ws;
        const int span = p.xcd_map == 2 ? tn / p.xcd_gn : 1;     // workgroups of one XCD that share an M tile
        if (want >= 8 && want % (8 * span) == 0 && nblocks > want && nblocks % want == 0 && ((long long)(want / 8 / span) * 256) % hw == 0) {
            gx = (unsigned)want;
            p.persist_iters = (int)(nblocks / want);
        }
    }
    // The streaming 1x1 kernel: 256 workgroups (one per CU), each `iters` tiles 32 / span tile_m apart (same XCD, same weight panel).
    const int sgrid = stream_grid(), sspan = p.xcd_map == 2 ? tn / p.xcd_gn : 1;
    const bool stream = plan.stream && p.splits == 1 && (p.xcd_map == 1 || p.xcd_map == 2) && nblocks % sgrid == 0 && (sgrid / 8) % sspan == 0;
    if (stream) {
        p.persist_iters = (int)(nblocks / sgrid);
        p.persist_dm = (sgrid / 8 / sspan) * BM;
        p.persist_d1 = (unsigned)((long long)p.persist_dm * a.C1 * es);
        p.persist_d2 = (unsigned)((long long)p.persist_dm * a.C2 * es);
        gx = (unsigned)sgrid;
    }
    // The lean wide epilogue (epilogue_wide, LEANM = 1): f32 storage on the 32 x 32 tiles, EVERY wave tile of the launch whole along M and in whole 32-column passes along N, rows in NHWC order,
    // one sample per wave tile under a fused PreNorm, 32-bit byte offsets.  NOPE_EPILOGUE_LEAN=0: the generic row loop everywhere (A/B).
    p.lean = (dt == NOPE_BF16X3 && dma && plan.small < 0 && p.wide_out && !p.posmajor && !phased && p.splits == 1 && !a.geglu && M % bm == 0 && a.Cout % 32 == 0 &&
              (!a.pn_ms || ((long long)p.Hm * p.Wm) % 64 == 0) && (unsigned long long)M * a.Cout * 4ull < 0xffffffffull && NOPE_ENV("NOPE_EPILOGUE_LEAN", 1) != 0) ? 1 : 0;
    if (p.splits > 1) p.out_amax = nullptr;        // (raw partials: the reduce kernel writes the tensor)
    const dim3 grid(gx, phased ? 4u : 1u, (unsigned)p.splits), block(NT);
    const bool trace = NOPE_ENV_SET("NOPE_CONV_TRACE");     // tuning aid: one line per launch
    if (trace && plan.small >= 0) fprintf(stderr, "conv small%d mode %d taps %d Cin %d Cout %d M %lld tiles %dx%d grid %u,%u,%u xcd %d/%d\n", plan.small, a.mode, a.ntaps, Cin, a.Cout, M,
                                        p.tiles_m, p.tiles_n, grid.x, grid.y, grid.z, p.xcd_map, p.xcd_gn);
    else if (trace) fprintf(stderr, "conv %s mode %d taps %d Cin %d Cout %d M %lld tiles %dx%d grid %u,%u,%u posmajor %d persist %d xcd %d/%d%s%s\n",
                       stream ? "stream128" : plan.halo ? "halo256" : plan.pp ? "pp256" : dma ? "dma128" : "generic", a.mode, a.ntaps, Cin, a.Cout, M, p.tiles_m, p.tiles_n, grid.x, grid.y, grid.z,
                       p.posmajor, p.persist_iters, p.xcd_map, p.xcd_gn, a.geglu ? " geglu" : x2 ? " x2" : "", p.lean ? " lean" : "");
    if (plan.small >= 0) {
        launch_conv_small(dt, &p, plan.small, grid, s);
    } else if (stream) {
        launch_conv_stream(dt, &p, grid, s);
    } else if (plan.pp) {
        if (NOPE_ENV_SET("NOPE_PP_VARIANT")) p.variant = NOPE_ENV("NOPE_PP_VARIANT", 0);      // tuning ablations of the ping-pong kernel
        if (plan.halo) launch_conv_halo(x2 ? NOPE_F16X2 : dt, &p, grid, s);
        else launch_conv_pp(x2 ? NOPE_F16X2 : dt, &p, grid, s);
    } else if (dt == NOPE_F32) {
        if (dma) launch_conv_dma_f32(&p, bm, grid, s);
        else if (p.pn_ms) hipLaunchKernelGGL((conv_gemm_kernel<float, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_gemm_kernel<float, false>), grid, block, 0, s, p);
    } else if (dt == NOPE_BF16X3) {
        if (dma) launch_conv_dma_bf16x3(&p, grid, s);
        else if (p.pn_ms) hipLaunchKernelGGL((conv_gemm_kernel<f32s_t, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_gemm_kernel<f32s_t, false>), grid, block, 0, s, p);
    } else if (dt == NOPE_F16) {
        if (dma) launch_conv_dma_f16(&p, grid, s);
        else if (p.pn_ms) hipLaunchKernelGGL((conv_gemm_kernel<f16_t, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_gemm_kernel<f16_t, false>), grid, block, 0, s, p);
    } else {
        if (dma && (bm == 256 || (variant & 2))) launch_conv_dma_bf16_variant(&p, bm, grid, s);
        else if (dma) launch_conv_dma_bf16(&p, grid, s);
        else if (p.pn_ms) hipLaunchKernelGGL((conv_gemm_kernel<bf16_t, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_gemm_kernel<bf16_t, false>), grid, block, 0, s, p);
    }
    NOPE_CHECK_LAUNCH();
    if (p.splits > 1 && a.colstats) {          // (tap-resident split-K only: plan.hsplit)
        const dim3 rg((unsigned)(M / a.stat_rows), (unsigned)cdiv(a.Cout, 64));      // (splits <= 16, stat_rows in {16, 32, 64})
        if (dt_es(dt) == 4) hipLaunchKernelGGL((splitk_reduce_stats_kernel<float>), rg, dim3(256), 0, s, p.split_out, p.splits, (int)M, a.Cout, a.bias, a.act, (float*)a.out, a.colstats, a.stat_rows);
        else if (dt == NOPE_F16) hipLaunchKernelGGL((splitk_reduce_stats_kernel<f16_t>), rg, dim3(256), 0, s, p.split_out, p.splits, (int)M, a.Cout, a.bias, a.act, (f16_t*)a.out, a.colstats, a.stat_rows);
        else hipLaunchKernelGGL((splitk_reduce_stats_kernel<bf16_t>), rg, dim3(256), 0, s, p.split_out, p.splits, (int)M, a.Cout, a.bias, a.act, (bf16_t*)a.out, a.colstats, a.stat_rows);
        NOPE_CHECK_LAUNCH();
    } else if (p.splits > 1) {
        const size_t MN = (size_t)M * a.Cout;
        const unsigned rb = (unsigned)((MN + 255) / 256 < 4096 ? (MN + 255) / 256 : 4096);
        if (dt_es(dt) == 4) hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(rb), dim3(256), 0, s, p.split_out, p.splits, (int)M, a.Cout,
                                               a.bias, (const float*)a.resid, a.act, (float*)a.out, a.out_nchw, a.Ho * a.Wo);
        else if (dt == NOPE_F16) hipLaunchKernelGGL((splitk_reduce_kernel<f16_t>), dim3(rb), dim3(256), 0, s, p.split_out, p.splits, (int)M, a.Cout,
                                a.bias, (const f16_t*)a.resid, a.act, (f16_t*)a.out, a.out_nchw, a.Ho * a.Wo);
        else hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(rb), dim3(256), 0, s, p.split_out, p.splits, (int)M, a.Cout,
                                a.bias, (const bf16_t*)a.resid, a.act, (bf16_t*)a.out, a.out_nchw, a.Ho * a.Wo);
        NOPE_CHECK_LAUNCH();
    }
    return NOPE_OK;
}

}  // namespace nope
