// hipemu -- a tiny CPU interpreter for the subset of HIP used by nope_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY.  It lives under tests/, is never built by
// `__graft_entry__.build()` into the product library and nothing under `nope_amd/`
// can load it.  Its single purpose: the GPU-less build container can execute the
// *unmodified* kernel sources (index math, tap geometry, LDS swizzles, masks,
// epilogues, the C++ U-Net schedule) against the oracle before a GPU-minute is spent.
// It is NOT a fallback: `nope_amd.hip` loads only the gfx950 library and fails loudly
// without it.
//
// Model: a workgroup is a set of fibers (ucontext) on one OS thread; `__syncthreads`
// and every wave-collective (shuffles, MFMA) are rendezvous points.  Workgroups are
// distributed over OS threads; `__shared__` is `static thread_local`, so concurrently
// running workgroups never share LDS.  MFMA fragment layouts follow
// /opt/skills/guides/cdna_hip_programming.md §3 (what the real kernels assume, too --
// the first GPU run has a dedicated fragment-layout test because the emulator cannot
// validate that assumption).
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define HIPEMU 1
#define NOPE_KEEP_VGPR(x) ((void)(x))   // (register-liveness pin of the tuning builds: nothing to pin on the host)
#define NOPE_OPAQUE_VGPR(x) ((void)(x)) // (code-motion fence of the device compiler)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipemu_uint3 { unsigned x, y, z; };

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace hipemu {

constexpr int kWave = 64;
constexpr size_t kStack = 96 * 1024;

struct WaveSync {
    int arrived = 0;
    unsigned gen = 0;
    int nlanes = 0;
    alignas(16) unsigned char slot[2][kWave][64];
};

// An LDS-DMA transfer issued but not yet landed (HIPEMU_DMA=late: it lands at the issuing lane's next vmcnt wait that
// retires it -- the LATEST moment the hardware allows, so a ds_read placed before the covering wait sees stale LDS).
struct PendingDma { unsigned char* dst; const unsigned char* src; unsigned size; };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    hipemu_uint3 tid;
    int flat = 0;
    bool at_barrier = false;      // parked in a workgroup barrier
    std::vector<PendingDma> dma;
};

struct BlockCtx {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    std::vector<WaveSync> waves;
    int nthreads = 0;
    int barrier_arrived = 0;
    unsigned barrier_gen = 0;
    int cur = -1;
    std::function<void()> body;
    hipemu_uint3 bid, bdim, gdim;
};

inline thread_local BlockCtx* g_blk = nullptr;
inline thread_local bool g_fp16_ovfl = false;     // the interpreter's copy of MODE.FP16_OVFL (see hipemu_s_setreg)
inline thread_local hipemu_uint3 g_tid, g_bid, g_bdim, g_gdim;

inline void yield_to_sched() {
    BlockCtx* b = g_blk;
    int me = b->cur;
    swapcontext(&b->fibers[me].ctx, &b->sched);
    // resumed
    g_tid = b->fibers[me].tid;
}

inline void fiber_entry() {
    BlockCtx* b = g_blk;
    int me = b->cur;
    g_tid = b->fibers[me].tid;
    b->body();
    b->fibers[me].done = true;
    swapcontext(&b->fibers[me].ctx, &b->sched);
}

inline bool dma_late() { static const bool v = getenv("HIPEMU_DMA") && !strcmp(getenv("HIPEMU_DMA"), "late"); return v; }
inline unsigned shuffle_seed() { static const unsigned v = getenv("HIPEMU_SHUFFLE") ? (unsigned)atoi(getenv("HIPEMU_SHUFFLE")) : 0u; return v; }

// retire this lane's outstanding LDS-DMA transfers, oldest first, until at most `keep` remain (s_waitcnt vmcnt(keep))
inline void dma_wait(size_t keep) {
    BlockCtx* b = g_blk;
    auto& q = b->fibers[b->cur].dma;
    if (q.size() <= keep) return;
    const size_t n = q.size() - keep;
    for (size_t i = 0; i < n; ++i) {
        if (q[i].src) memcpy(q[i].dst, q[i].src, q[i].size);
        else memset(q[i].dst, 0, q[i].size);
    }
    q.erase(q.begin(), q.begin() + n);
}

inline void run_block(BlockCtx& b) {
    g_blk = &b;
    g_fp16_ovfl = false;
    g_bid = b.bid; g_bdim = b.bdim; g_gdim = b.gdim;
    int n = b.nthreads;
    for (int i = 0; i < n; ++i) {
        Fiber& f = b.fibers[i];
        f.done = false;
        f.at_barrier = false;
        f.dma.clear();
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &b.sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    for (auto& w : b.waves) { w.arrived = 0; w.gen = 0; }
    int nw = (n + kWave - 1) / kWave;
    for (int w = 0; w < nw; ++w) b.waves[w].nlanes = std::min(kWave, n - w * kWave);
    b.barrier_arrived = 0; b.barrier_gen = 0;
    int remaining = n;
    long spins = 0;
    unsigned rng = shuffle_seed() * 2654435761u + b.bid.x * 40503u + 12345u;
    while (remaining > 0 && shuffle_seed()) {
        // HIPEMU_SHUFFLE=seed -- adversarial wave scheduling: waves are taken in a pseudo-random order and each one runs
        // on its own, through all its wave-level rendezvous points, until every lane of it is parked in a WORKGROUP
        // barrier (or has finished).  A wave therefore gets as far ahead of the others as the barriers allow: an LDS
        // hand-off that relies on waves happening to move in step (a missing barrier, a ring one stage too short)
        // gives wrong results under some order.
        rng = rng * 1664525u + 1013904223u;
        const int start = (int)((rng >> 8) % (unsigned)nw), dir = (rng >> 20) & 1 ? 1 : -1;
        for (int wi = 0; wi < nw; ++wi) {
            const int w = ((start + dir * wi) % nw + nw) % nw;
            const int lo = w * kWave, hi = std::min(n, lo + kWave);
            for (;;) {
                const unsigned gen0 = b.barrier_gen;
                bool parked = true;
                for (int i = lo; i < hi; ++i) {
                    if (b.fibers[i].done) continue;
                    b.cur = i;
                    swapcontext(&b.sched, &b.fibers[i].ctx);
                    if (b.fibers[i].done) { --remaining; continue; }
                    if (!b.fibers[i].at_barrier) parked = false;
                }
                if (parked && b.barrier_gen == gen0) break;
                if (++spins > 100000000L) { fprintf(stderr, "hipemu: deadlock?\n"); abort(); }
            }
        }
    }
    while (remaining > 0) {
        int progressed = 0;
        for (int i = 0; i < n; ++i) {
            if (b.fibers[i].done) continue;
            b.cur = i;
            swapcontext(&b.sched, &b.fibers[i].ctx);
            if (b.fibers[i].done) { --remaining; }
            ++progressed;
        }
        if (++spins > 100000000L) { fprintf(stderr, "hipemu: deadlock?\n"); abort(); }
        (void)progressed;
    }
    g_blk = nullptr;
}

// s_barrier alone: a rendezvous, nothing is waited for
inline void raw_barrier() {
    BlockCtx* b = g_blk;
    unsigned gen = b->barrier_gen;
    if (++b->barrier_arrived == b->nthreads) { b->barrier_arrived = 0; ++b->barrier_gen; return; }
    Fiber& me = b->fibers[b->cur];
    me.at_barrier = true;
    while (b->barrier_gen == gen) yield_to_sched();
    me.at_barrier = false;
}

// __syncthreads(): hipcc puts s_waitcnt vmcnt(0) in front of the s_barrier while an LDS-DMA is outstanding
inline void syncthreads() {
    dma_wait(0);
    raw_barrier();
}

// wave rendezvous: publish `bytes` of payload, wait for the whole wave, return slot table
inline const unsigned char (*wave_exchange(const void* payload, size_t bytes))[64] {
    BlockCtx* b = g_blk;
    int flat = b->fibers[b->cur].flat;
    WaveSync& w = b->waves[flat / kWave];
    unsigned gen = w.gen;
    int lane = flat % kWave;
    memcpy(w.slot[gen & 1][lane], payload, bytes);
    if (++w.arrived == w.nlanes) { w.arrived = 0; ++w.gen; }
    else while (w.gen == gen) yield_to_sched();
    return w.slot[gen & 1];
}
inline int lane_id() { BlockCtx* b = g_blk; return b->fibers[b->cur].flat % kWave; }

template <class F>
void launch(dim3 grid, dim3 block, F&& body) {
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    int nthreads = block.x * block.y * block.z;
    unsigned hw = std::thread::hardware_concurrency();
    const char* env = getenv("HIPEMU_THREADS");
    if (env) hw = (unsigned)atoi(env);
    size_t nos = std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, nblocks));
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        BlockCtx b;
        b.nthreads = nthreads;
        b.fibers.resize(nthreads);
        b.waves.resize((nthreads + kWave - 1) / kWave);
        for (int i = 0; i < nthreads; ++i) {
            b.fibers[i].stack = (char*)malloc(kStack);
            b.fibers[i].flat = i;
            b.fibers[i].tid = {unsigned(i % block.x), unsigned((i / block.x) % block.y), unsigned(i / (block.x * block.y))};
        }
        b.body = body;
        b.bdim = {block.x, block.y, block.z};
        b.gdim = {grid.x, grid.y, grid.z};
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            b.bid = {unsigned(i % grid.x), unsigned((i / grid.x) % grid.y), unsigned(i / ((size_t)grid.x * grid.y))};
            run_block(b);
        }
        for (auto& f : b.fibers) free(f.stack);
    };
    if (nos == 1) { worker(); return; }
    std::vector<std::thread> ts;
    for (size_t t = 0; t < nos; ++t) ts.emplace_back(worker);
    for (auto& t : ts) t.join();
}

}  // namespace hipemu

#define threadIdx (hipemu::g_tid)
#define blockIdx (hipemu::g_bid)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), [=]() { kern(__VA_ARGS__); })

inline void __syncthreads() { hipemu::syncthreads(); }

template <class T>
inline T __shfl(T v, int src, int width = 64) {
    auto s = hipemu::wave_exchange(&v, sizeof(T));
    int lane = hipemu::lane_id();
    int base = lane & ~(width - 1);
    T r; memcpy(&r, s[base + (src & (width - 1))], sizeof(T)); return r;
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    auto s = hipemu::wave_exchange(&v, sizeof(T));
    int lane = hipemu::lane_id();
    int src = lane ^ mask;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    T r; memcpy(&r, s[src], sizeof(T)); return r;
}
template <class T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
    auto s = hipemu::wave_exchange(&v, sizeof(T));
    int lane = hipemu::lane_id();
    int src = lane + (int)delta;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    T r; memcpy(&r, s[src], sizeof(T)); return r;
}
inline unsigned long long __ballot(int pred) {
    auto s = hipemu::wave_exchange(&pred, sizeof(int));
    unsigned long long m = 0;
    hipemu::BlockCtx* b = hipemu::g_blk;
    int n = b->waves[b->fibers[b->cur].flat / 64].nlanes;
    for (int i = 0; i < n; ++i) { int p; memcpy(&p, s[i], 4); if (p) m |= 1ull << i; }
    return m;
}
inline int __all(int pred) {
    hipemu::BlockCtx* b = hipemu::g_blk;
    int n = b->waves[b->fibers[b->cur].flat / 64].nlanes;
    unsigned long long full = n == 64 ? ~0ull : ((1ull << n) - 1);
    return __ballot(pred) == full;
}
inline int __any(int pred) { return __ballot(pred) != 0; }

// ---- MFMA (fragment maps: cdna_hip_programming.md §3) ---------------------------------
typedef __attribute__((ext_vector_type(8))) short hipemu_bf16x8;
typedef __attribute__((ext_vector_type(4))) float hipemu_f32x4;
typedef __attribute__((ext_vector_type(16))) float hipemu_f32x16;

inline float hipemu_bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// v_mfma_f32_16x16x32_bf16: A[i=l&15][k=(l>>4)*8+e], B[k=(l>>4)*8+e][j=l&15]; D: col=l&15,row=(l>>4)*4+r
inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c, int, int, int) {
    struct P { hipemu_bf16x8 a, b; } p{a, b};
    auto s = hipemu::wave_exchange(&p, sizeof(P));
    int lane = hipemu::lane_id();
    int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            P pa, pb;
            memcpy(&pa, s[row + 16 * (k >> 3)], sizeof(P));
            memcpy(&pb, s[col + 16 * (k >> 3)], sizeof(P));
            acc = fmaf(hipemu_bf16_to_f32((unsigned short)pa.a[k & 7]), hipemu_bf16_to_f32((unsigned short)pb.b[k & 7]), acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x4_f32: A[l&15][k=l>>4], B[k=l>>4][l&15]; exact fmaf chain over k
inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    struct P { float a, b; } p{a, b};
    auto s = hipemu::wave_exchange(&p, sizeof(P));
    int lane = hipemu::lane_id();
    int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            P pa, pb;
            memcpy(&pa, s[row + 16 * k], sizeof(P));
            memcpy(&pb, s[col + 16 * k], sizeof(P));
            acc = fmaf(pa.a, pb.b, acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=(l>>5)*8+e], B[k][j=l&31]; D: col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5)
inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c, int, int, int) {
    struct P { hipemu_bf16x8 a, b; } p{a, b};
    auto s = hipemu::wave_exchange(&p, sizeof(P));
    int lane = hipemu::lane_id();
    int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            P pa, pb;
            memcpy(&pa, s[row + 32 * (k >> 3)], sizeof(P));
            memcpy(&pb, s[col + 32 * (k >> 3)], sizeof(P));
            acc = fmaf(hipemu_bf16_to_f32((unsigned short)pa.a[k & 7]), hipemu_bf16_to_f32((unsigned short)pb.b[k & 7]), acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_32x32x16_f16: same fragment maps as the bf16 form
typedef __attribute__((ext_vector_type(8))) _Float16 hipemu_f16x8;
inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x16 c, int, int, int) {
    struct P { hipemu_f16x8 a, b; } p{a, b};
    auto s = hipemu::wave_exchange(&p, sizeof(P));
    int lane = hipemu::lane_id();
    int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            P pa, pb;
            memcpy(&pa, s[row + 32 * (k >> 3)], sizeof(P));
            memcpy(&pb, s[col + 32 * (k >> 3)], sizeof(P));
            acc = fmaf((float)pa.a[k & 7], (float)pb.b[k & 7], acc);
        }
        c[r] = acc;
    }
    return c;
}
// ---- MODE.FP16_OVFL (s_setreg hwreg(MODE, 23, 1)): saturating f16 / fp8 conversions for the rest of the wave's life.  One flag per OS thread
// = per workgroup being interpreted, cleared when a block starts (hipemu::run_block).
inline void hipemu_s_setreg(int hwreg, int value) {
    if (hwreg == (1 | (23 << 6))) hipemu::g_fp16_ovfl = value != 0;
    else { fprintf(stderr, "hipemu: s_setreg of hwreg encoding %d is not emulated\n", hwreg); abort(); }
}
#define __builtin_amdgcn_s_setreg hipemu_s_setreg
#define NOPE_CVT_PK_F16_OVFL(lo, hi) (hipemu::g_fp16_ovfl ? cvt_pk_f16(lo, hi) : cvt_pk_f16_raw(lo, hi))
// ---- OCP e4m3 (fn) and the MX-scaled fp8 MFMA, as measured on gfx950 by tools/probes/mx_probe.hip
inline float hipemu_e4m3_to_f32(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7) r = NAN;
    else if (e == 0) r = ldexpf((float)m, -9);
    else r = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -r : r;
}
inline unsigned char hipemu_f32_to_e4m3(float x) {       // round to nearest even; beyond the largest value (448 + half a step): NaN
    const unsigned char sign = std::signbit(x) ? 0x80 : 0;
    const float ax = fabsf(x);
    if (!(ax == ax)) return (unsigned char)(sign | 0x7f);
    if (ax >= 480.0f) return (unsigned char)(sign | (hipemu::g_fp16_ovfl ? 0x7e : 0x7f));      // FP16_OVFL: saturate at 448 (infinities too, close enough)
    unsigned char best = 0; float bd = INFINITY;
    for (int v = 0; v < 127; ++v) {
        const float dd = fabsf(hipemu_e4m3_to_f32((unsigned char)v) - ax);
        if (dd < bd || (dd == bd && !(v & 1) && (best & 1))) { bd = dd; best = (unsigned char)v; }
    }
    return (unsigned char)(best | sign);
}
// v_cvt_pk_fp8_f32 dst, a, b: byte 0 (or 2 with word_sel) = e4m3(a), byte 1 (3) = e4m3(b); the other half of `old` is kept
inline int hipemu_cvt_pk_fp8_f32(float a, float b, int old, bool word_sel) {
    const unsigned pk = (unsigned)hipemu_f32_to_e4m3(a) | ((unsigned)hipemu_f32_to_e4m3(b) << 8);
    return word_sel ? (int)(((unsigned)old & 0x0000ffffu) | (pk << 16)) : (int)(((unsigned)old & 0xffff0000u) | pk);
}
#define __builtin_amdgcn_cvt_pk_fp8_f32 hipemu_cvt_pk_fp8_f32
// v_cvt_scalef32_pk_fp8_f32 (probe fact 6): e4m3(value / 2^floor(log2 scale)) into the half of `old` that word_sel names
typedef short hipemu_s16x2 __attribute__((ext_vector_type(2)));
inline hipemu_s16x2 hipemu_cvt_scalef32_pk_fp8_f32(hipemu_s16x2 old, float a, float b, float scale, bool word_sel) {
    const float pw = ldexpf(1.0f, ilogbf(scale));
    const short pk = (short)((unsigned)hipemu_f32_to_e4m3(a / pw) | ((unsigned)hipemu_f32_to_e4m3(b / pw) << 8));
    old[word_sel ? 1 : 0] = pk;
    return old;
}
#define __builtin_amdgcn_cvt_scalef32_pk_fp8_f32 hipemu_cvt_scalef32_pk_fp8_f32
// v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 x e4m3 (cbsz = blgp = 0), op_sel 0: byte e of lane (i, half h) of A pairs with byte e of lane
// (j, half h) of B; an element's scale block is its byte-position half: bytes 0..15 take byte 0 of the scale register of lane i (j),
// bytes 16..31 that of lane i + 32 (j + 32); value x 2^(scale - 127).  The 64-term sum is exact here (the hardware truncates it at ~2^-12
// of the largest term).
typedef __attribute__((ext_vector_type(8))) int hipemu_i32x8;
inline hipemu_f32x16 hipemu_mfma_scale_f32_32x32x64_f8f6f4(hipemu_i32x8 a, hipemu_i32x8 b, hipemu_f32x16 c, int cbsz, int blgp, int opsel_a, int scale_a,
                                                          int opsel_b, int scale_b) {
    if (cbsz || blgp || opsel_a || opsel_b) { fprintf(stderr, "hipemu: only the e4m3 x e4m3, op_sel 0 form of the MX MFMA is emulated\n"); abort(); }
    struct P { unsigned char a[32], b[32]; int sa, sb; } p;
    memcpy(p.a, &a, 32); memcpy(p.b, &b, 32); p.sa = scale_a & 0xff; p.sb = scale_b & 0xff;
    // (the wave_exchange slot is 64 bytes: send the two operands in two rounds)
    struct Q { unsigned char v[32]; int s; };
    Q qa, qb;
    memcpy(qa.v, p.a, 32); qa.s = p.sa; memcpy(qb.v, p.b, 32); qb.s = p.sb;
    Q A[64], B[64];
    { auto s = hipemu::wave_exchange(&qa, sizeof(Q)); for (int l = 0; l < 64; ++l) memcpy(&A[l], s[l], sizeof(Q)); }
    { auto s = hipemu::wave_exchange(&qb, sizeof(Q)); for (int l = 0; l < 64; ++l) memcpy(&B[l], s[l], sizeof(Q)); }
    int lane = hipemu::lane_id();
    int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        double acc = 0.0;
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 32; ++e) {
                const double sa = ldexp(1.0, A[row + 32 * (e >> 4)].s - 127), sb = ldexp(1.0, B[col + 32 * (e >> 4)].s - 127);
                acc += (double)hipemu_e4m3_to_f32(A[row + 32 * h].v[e]) * sa * (double)hipemu_e4m3_to_f32(B[col + 32 * h].v[e]) * sb;
            }
        c[r] = (float)((double)c[r] + acc);
    }
    return c;
}
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4 hipemu_mfma_scale_f32_32x32x64_f8f6f4
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 hipemu_mfma_f32_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 hipemu_mfma_f32_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu_mfma_f32_32x32x16_bf16
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_s_memrealtime() 0ull   // (clock stamps of the tuning instantiations: no clock on the host)
// s_waitcnt simm16 (gfx9 encoding): vmcnt = bits [3:0] | [15:14] << 4; only the vmcnt field matters to the interpreter
inline void hipemu_s_waitcnt(unsigned imm) { hipemu::dma_wait((imm & 0xF) | (((imm >> 14) & 3) << 4)); }
#define __builtin_amdgcn_s_waitcnt(x) hipemu_s_waitcnt(x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
#define __builtin_amdgcn_s_barrier() hipemu::raw_barrier()
inline void hipemu_wave_barrier() { int z = 0; hipemu::wave_exchange(&z, sizeof(z)); }
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()
inline int hipemu_readfirstlane(int v) { return __shfl(v, 0); }
#define __builtin_amdgcn_readfirstlane hipemu_readfirstlane

// ---- buffer resources + LDS-DMA (buffer_load ... lds): destination = wave-uniform base + lane*size,
// out-of-range lanes write zeros (the raw-buffer range check).  The hardware lands the data asynchronously, any time
// between the issue and the vmcnt wait that retires it.  The interpreter offers the two extremes: the default copies
// immediately (EARLIEST landing: catches a DMA that overwrites LDS another wave is still reading), HIPEMU_DMA=late
// copies at the covering wait (LATEST landing: catches a ds_read placed before that wait + barrier).
struct hipemu_rsrc { const unsigned char* base; unsigned num; };
inline hipemu_rsrc hipemu_make_rsrc(void* p, short, int num, int) { return hipemu_rsrc{(const unsigned char*)p, (unsigned)num}; }
template <class P>
inline void hipemu_buffer_load_lds(hipemu_rsrc r, P ldsptr, unsigned size, unsigned voffset, unsigned soffset, unsigned imm, int) {
    uintptr_t base = (uintptr_t)ldsptr;
    auto s = hipemu::wave_exchange(&base, sizeof(base));
    uintptr_t b0; memcpy(&b0, s[0], sizeof(b0));
    if (b0 != base) { fprintf(stderr, "hipemu: LDS-DMA base is not wave-uniform\n"); abort(); }
    unsigned char* dst = (unsigned char*)base + imm + (size_t)hipemu::lane_id() * size;
    // the raw-buffer range check covers the vector offset + instruction offset only; the scalar offset is added afterwards
    const unsigned long long chk = (unsigned long long)voffset + imm;
    const unsigned long long off = chk + soffset;
    const unsigned char* src = chk + size > r.num ? nullptr : r.base + off;
    if (src && off + size > r.num) { fprintf(stderr, "hipemu: buffer load passes the range check but reads beyond the buffer (scalar offset)\n"); abort(); }
    if (hipemu::dma_late()) {
        hipemu::BlockCtx* b = hipemu::g_blk;
        b->fibers[b->cur].dma.push_back(hipemu::PendingDma{dst, src, size});
        return;
    }
    if (!src) memset(dst, 0, size);
    else memcpy(dst, src, size);
}
#define __builtin_amdgcn_make_buffer_rsrc hipemu_make_rsrc
#define __builtin_amdgcn_raw_ptr_buffer_load_lds hipemu_buffer_load_lds
// buffer_load / buffer_store_dwordx4 (address = base + vector offset + scalar offset; the range check covers the vector offset: out-of-range
// loads return zeros, out-of-range stores are dropped)
typedef __attribute__((ext_vector_type(4))) unsigned int hipemu_u32x4;
inline hipemu_u32x4 hipemu_buffer_load_b128(hipemu_rsrc r, unsigned voffset, unsigned soffset, int) {
    hipemu_u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned long long)voffset + 16 > r.num) return v;
    if ((unsigned long long)voffset + soffset + 16 > r.num) { fprintf(stderr, "hipemu: buffer load passes the range check but reads beyond the buffer (scalar offset)\n"); abort(); }
    memcpy(&v, r.base + voffset + soffset, 16);
    return v;
}
inline void hipemu_buffer_store_b128(hipemu_u32x4 v, hipemu_rsrc r, unsigned voffset, unsigned soffset, int) {
    if ((unsigned long long)voffset + 16 > r.num) return;
    if ((unsigned long long)voffset + soffset + 16 > r.num) { fprintf(stderr, "hipemu: buffer store passes the range check but writes beyond the buffer (scalar offset)\n"); abort(); }
    memcpy(const_cast<unsigned char*>(r.base) + voffset + soffset, &v, 16);
}
#define __builtin_amdgcn_raw_buffer_load_b128 hipemu_buffer_load_b128
#define __builtin_amdgcn_raw_buffer_store_b128 hipemu_buffer_store_b128

// v_med3_f32: with a NaN among the operands the hardware returns MIN3 of them (IEEE minNum: the NaN is ignored), else the median
inline float hipemu_fmed3f(float a, float b, float c) {
    if (a != a || b != b || c != c) return fminf(fminf(a, b), c);
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}
#define __builtin_amdgcn_fmed3f hipemu_fmed3f
inline float __expf(float x) { return expf(x); }
inline float hipemu_rcpf(float x) { return 1.0f / x; }
#define __builtin_amdgcn_rcpf hipemu_rcpf
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline float __fsqrt_rn(float a) { return sqrtf(a); }

inline float atomicAdd(float* p, float v) {
    float old = *p, neu;
    do { neu = old + v; } while (!__atomic_compare_exchange(p, &old, &neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}

inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- runtime API shims --------------------------------------------------------------------
inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
#define hipHostMallocMapped 0x2
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// device queries: a small pretend device (2 CUs x 2 resident workgroups), so launches sized by residency stay cheap to interpret
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }
template <class F> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 2; return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
// stream capture / graphs: not emulated -- BeginCapture fails, callers take their direct-launch path
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorInvalidValue; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
