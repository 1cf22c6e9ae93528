// Streaming 1x1 convolution (a plain GEMM over NHWC rows) for the launches that are bound by HBM, not by the matrix pipe: the short-K
// 1x1 convs around the attention blocks and the output head at the 32 x 32 / 16 x 16 levels of a 512-hypothesis step
// (reference: to_qkv / to_out of LinearAttention and Attention, model_utils.py:373-374,399-401; res_conv, :269; final_conv.1, u_net.py:156).
//
// What bounded them on the 128 x 192 LDS-DMA kernel (conv_gemm_dma.h): a tile of 3-6 K steps pays one exposed memory round trip PER STEP
// (two-stage ring: the loads of step k + 1 are issued at step k) and nothing is in flight while its epilogue runs -- 192 -> 384 at
// 32 x 32 x 512 moved 1.2 GB in 460 us = 2.6 TB/s with 32 KB of activations in flight per CU.  Little's law wants ~50 KB per CU.
//
// This kernel keeps the activation stream CONTINUOUS:
//   * one persistent workgroup per CU (256 threads, a 128 x 192 tile per step of its walk: same tile, same MFMA stage, same epilogues
//     and therefore the same bits as the 128 x 192 kernel), walking `persist_iters` tiles of one weight panel inside its XCD's run;
//   * the A operand flows through a FIVE-stage ring as one flat sequence of K steps over all the tiles of the walk: the loads of four
//     steps are in flight at any time, across tile boundaries and under the epilogue (64 KB per CU);
//   * the weights (L2 hits) flow through a three-stage ring per tile; the epilogue's per-wave panels alias it;
//   * vmcnt retires a wave's loads IN ORDER, so a wave that waited for a young weight piece would drain every older activation piece
//     with it and the deep ring would be worth nothing.  Hence the ROLES: waves 0, 1 issue every activation piece (and count only
//     those), waves 2, 3 every weight piece; the workgroup barrier joins them.  All four multiply.
// Preconditions (plan_stream, kernels_gemm.hip): PLAIN 1 x 1, one or two sources without broadcast, M % 128 == 0, Cin % K-step == 0,
// wide NHWC output, no split, tiles a multiple of the grid.
#include <cstdio>
#include <cstdlib>

#include "conv_gemm_common.h"

namespace nope {

namespace {

// s_waitcnt vmcnt(n), other counters at their maximum (gfx9 encoding: vmcnt = [3:0] | [15:14] << 4)
template <int N> __device__ __forceinline__ void stream_wait_vmcnt() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }

template <class T, bool PN, bool LEAN = false>
__global__ __launch_bounds__(256, 1) void conv1x1_stream_kernel(ConvParams p) {
    typedef Tile<T> TL;
    constexpr int VEC = Elt<T>::VEC;
    constexpr unsigned ES = (unsigned)sizeof(T);
    constexpr int RB = 128, BK = RB / (int)ES;
    constexpr int NSA = 5, NSB = 3, DA = NSA - 1, DB = NSB - 1;      // ring stages; prefetch distances in K steps
    constexpr int ASTAGE = BM * RB, BSTAGE = BN * RB;                // 16 KiB, 24 KiB
    constexpr int AI = BM / 8 / 2, BI = BN / 8 / 2;                  // 1 KiB DMA pieces per ISSUING wave and stage: 8 (waves 0, 1), 12 (waves 2, 3)
    constexpr int RING_A = NSA * ASTAGE, RING_B = NSB * BSTAGE;
    static_assert(4 * Ep<T>::WAVE_BYTES <= RING_B, "the epilogue panels alias the weight ring");
    static_assert((DA - 1) * AI <= 63 && (DB - 1) * BI <= 63, "vmcnt field");
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING_A + RING_B];
    unsigned char* const ringA = lds;
    unsigned char* const ringB = lds + RING_A;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool a_role = wave < 2;                                    // (wave-uniform)
    const int rw_ = wave & 1;                                        // index inside the role
    int tile_m, tile_n;
    tile_coords(p, tile_m, tile_n);
    int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int Cin = p.C1 + p.C2;
    const int nk = Cin / BK;
    const int iters = p.persist_iters;
    const int G = iters * nk;                                        // A steps of the whole walk

    const auto r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src1, (short)0, (int)p.bytes1, 0x00020000);
    const auto r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src2 ? p.src2 : p.src1), (short)0, (int)(p.src2 ? p.bytes2 : p.bytes1), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)p.bytesw, 0x00020000);

    // ---- this lane's rows of the pieces its wave issues.  A piece = 8 consecutive 128-byte rows (lane -> row l >> 3, slot l & 7); the XOR
    // swizzle is applied to the SOURCE channel chunk so the LDS image is the swizzled tile the fragment reads expect (conv_gemm_dma.h).
    // (fixed-size arrays: with template-sized arrays captured by a lambda hipcc (ROCm 7.2) drops the kernel's host stub)
    const int rsub = lane >> 3, lslot = lane & 7;
    unsigned a_b1[8], a_b2[8], b_off[12];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = 8 * (AI * rw_ + i) + rsub;
        const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(row)) * VEC);
        const unsigned m = (unsigned)(m0 + row);                     // (M % 128 == 0: every row exists; 1 x 1, no broadcast: pixel index = m)
        a_b1[i] = (m * (unsigned)p.C1 + cs) * ES;
        a_b2[i] = (m * (unsigned)p.C2 + cs) * ES;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = 8 * (BI * rw_ + j) + rsub;
        const int n = n0 + row;
        const unsigned cs = (unsigned)((lslot ^ swz_of<RB>(row)) * VEC);
        b_off[j] = n < p.Cout ? ((unsigned)n * (unsigned)Cin + cs) * ES : OOB;
    }

    // ---- the activation stream (waves 0, 1): step ld_k of tile ld_it into ring slot slot_ld
    int ld_k = 0, ld_g = 0, slot_ld = 0;
    auto issue_a = [&]() {
        const int c0 = ld_k * BK;
        const bool first = c0 < p.C1;                                // wave-uniform: a K step lies inside one source
        const unsigned kadd = (unsigned)(first ? c0 : c0 - p.C1) * ES;
        unsigned char* d = ringA + slot_ld * ASTAGE + (AI * rw_) * 1024;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? r1 : r2, (lds_void_t*)(d + i * 1024), 16, (first ? a_b1[i] : a_b2[i]) + kadd, 0, 0, 0);
        if (++ld_k == nk) {                                          // the next tile of the walk: a constant added to every row
            ld_k = 0;
#pragma unroll
            for (int i = 0; i < AI; ++i) { a_b1[i] += p.persist_d1; a_b2[i] += p.persist_d2; }
        }
        ++ld_g;
        slot_ld = slot_ld + 1 == NSA ? 0 : slot_ld + 1;
    };
    // ---- the weights (waves 2, 3): step k of the current tile into ring slot k % NSB
    auto issue_b = [&](int k, int slot) {
        const unsigned kofs = (unsigned)(k * BK) * ES;
        unsigned char* d = ringB + slot * BSTAGE + (BI * rw_) * 1024;
#pragma unroll
        for (int j = 0; j < BI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(d + j * 1024), 16, b_off[j] + kofs, 0, 0, 0);
    };

    typename TL::acc_t acc[TL::MT][TL::NTL];
#pragma unroll
    for (int i = 0; i < TL::MT; ++i)
#pragma unroll
        for (int j = 0; j < TL::NTL; ++j)
#pragma unroll
            for (int r = 0; r < TL::R; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: DA activation steps and DB weight steps in flight
    if (a_role) {
#pragma unroll
        for (int s = 0; s < DA; ++s)
            if (s < G) issue_a();
    } else {
#pragma unroll
        for (int s = 0; s < DB; ++s)
            if (s < nk) issue_b(s, s);
    }
    int g = 0, slot_a = 0;                                           // the A step being consumed and its ring slot
    for (int it = 0; it < iters; ++it) {
        int slot_b = 0;
        for (int k = 0; k < nk; ++k, ++g) {
            // This wave's OLDEST outstanding stage is the one consumed now; the younger ones stay in flight.
            if (a_role) {
                const int younger = G - 1 - g;
                if (younger >= 3) stream_wait_vmcnt<3 * AI>();
                else if (younger == 2) stream_wait_vmcnt<2 * AI>();
                else if (younger == 1) stream_wait_vmcnt<AI>();
                else stream_wait_vmcnt<0>();
            } else {
                if (nk - 1 - k >= 1) stream_wait_vmcnt<BI>();
                else stream_wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();          // every piece of this step has landed; everyone is done with the previous step's slots
            __builtin_amdgcn_sched_barrier(0);
            if (a_role) { if (g + DA < G) issue_a(); }                                   // into the slot of step g - 1
            else if (k + DB < nk) issue_b(k + DB, slot_b == 0 ? NSB - 1 : slot_b - 1);   // into the slot of step k - 1
            __builtin_amdgcn_s_setprio(2);
            mma_stage<T, RB>(ringA + slot_a * ASTAGE, ringB + slot_b * BSTAGE, wm, wn, lane, acc);
            __builtin_amdgcn_s_setprio(0);
            slot_a = slot_a + 1 == NSA ? 0 : slot_a + 1;
            slot_b = slot_b + 1 == NSB ? 0 : slot_b + 1;
        }
        __builtin_amdgcn_s_barrier();              // every wave is done reading the weight ring: the panels alias it
        __builtin_amdgcn_sched_barrier(0);
        epilogue_wide<T, PN, false, NoStamp, false, LEAN ? 1 : 0>(p, acc, m0, n0, wm, wn, lane, ringB + wave * Ep<T>::WAVE_BYTES);
        m0 += p.persist_dm;
        if (it + 1 < iters) {
#pragma unroll
            for (int i = 0; i < TL::MT; ++i)
#pragma unroll
                for (int j = 0; j < TL::NTL; ++j)
#pragma unroll
                    for (int r = 0; r < TL::R; ++r) acc[i][j][r] = 0.f;
            __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0): my panel reads are done
            __builtin_amdgcn_s_barrier();          // the panels are free: the next tile's first weight steps may land on them
            __builtin_amdgcn_sched_barrier(0);
            if (!a_role) {
#pragma unroll
                for (int s = 0; s < DB; ++s)
                    if (s < nk) issue_b(s, s);
            }
        }
    }
}

template <class T>
void launch_stream_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
        if (p.lean) {
            if (p.pn_ms) hipLaunchKernelGGL((conv1x1_stream_kernel<T, true, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((conv1x1_stream_kernel<T, false, true>), grid, dim3(256), 0, s, p);
            return;
        }
    }
    if (p.pn_ms) hipLaunchKernelGGL((conv1x1_stream_kernel<T, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv1x1_stream_kernel<T, false>), grid, dim3(256), 0, s, p);
}

}  // namespace

void launch_conv_stream(int dt, const void* params, dim3 grid, hipStream_t s) {
    const ConvParams& p = *reinterpret_cast<const ConvParams*>(params);
    if (dt == NOPE_BF16X3) launch_stream_t<f32s_t>(p, grid, s);
    else if (dt == NOPE_F16) launch_stream_t<f16_t>(p, grid, s);
    else launch_stream_t<bf16_t>(p, grid, s);
}

}  // namespace nope
