// GPU probe behind the f16x2 compute mode: what v_mfma_scale_f32_32x32x64_f8f6f4 and v_cvt_pk_fp8_f32 really do on
// gfx950 -- the facts the tap-resident kernel's cross-term MFMA relies on and the CPU interpreter (tests/hipemu) can only assume.
//     hipcc --offload-arch=gfx950 -O2 tools/probes/mx_probe.hip -o tools/probes/mx_probe && tools/probes/mx_probe
// Prints one line per fact with PASS / FAIL:
//   1. v_cvt_pk_fp8_f32 produces OCP e4m3fn (1.0 -> 0x38), rounds to nearest even, and what it does beyond 448 (no clamp in the instruction)
//   2. operand map: byte e of lane l = (i, half h) of A pairs with byte e of lane (j, half h) of B -- never across halves or byte positions
//      (the test arrays call that K index 32 h + e; the hardware's own numbering is K = 32 (e / 16) + 16 h + e % 16, see 3h)
//   3. block scales: a scale byte (op_sel 0 = byte 0 of the register) multiplies by 2^(scale - 127), separately for A and B, per row / column.
//      3h: the 32-value scale BLOCK of an element is its BYTE-POSITION half, not its lane half -- bytes 0..15 of both lane halves take the
//      scale offered by lanes 0..31, bytes 16..31 the one offered by lanes 32..63 (measured; all-ones operands cannot tell the two apart).
//      The f16x2 kernel offers ONE scale in every lane (3d), so only the pairing rule of fact 2 matters to it.
//   4. the C/D map is the 32x32 one: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5); fp8 subnormals are kept (4a)
//   6. v_cvt_scalef32_pk_fp8_f32 = e4m3(value / 2^floor(log2 scale)), round to nearest even, word_sel picks the half written -- the pre-scale
//      multiply folded into the conversion; like the plain conversion it does NOT saturate
//   7. under MODE.FP16_OVFL = 1 (s_setreg hwreg(MODE, 23, 1)) v_cvt_pk_f16_f32, v_cvt_pk_fp8_f32 and v_cvt_scalef32_pk_fp8_f32 SATURATE (+-65504 /
//      +-448) instead of producing inf / NaN; NaN inputs stay NaN -- the f16x2 operand rewrite runs under it and needs no clamps
//   5. accumulation: products are exact, but the 64-term sum is NOT f32-exact: measured error up to 2^-11.7 of the LARGEST term (the terms are
//      aligned to the largest and truncated) -- harmless for cross terms that are 2^-11 of the result, fatal for a main term
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void mx_kernel(const unsigned char* a, const unsigned char* b, const int* sa, const int* sb, const float* cin, float* d) {
    const int l = threadIdx.x;
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) {
        va[i] = reinterpret_cast<const int*>(a)[l * 8 + i];
        vb[i] = reinterpret_cast<const int*>(b)[l * 8 + i];
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = cin[l * 16 + i];
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 16; ++i) d[l * 16 + i] = c[i];
}

__global__ void cvt_kernel(const float* x, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (2 * i + 1 < n + 1) out[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false) & 0xffffu;
}

typedef __attribute__((ext_vector_type(2))) short s16x2;
__global__ void cvt_scale_kernel(const float* x, unsigned* out, int n, float scale) {
    const int i = threadIdx.x;
    if (2 * i + 1 < n + 1) {
        s16x2 old = {0, 0};
        s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[2 * i], x[2 * i + 1], scale, false);
        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[2 * i + 1], x[2 * i], scale, true);      // word_sel: the upper half, swapped pair
        out[i] = __builtin_bit_cast(unsigned, r);
    }
}

// MODE.FP16_OVFL (hwreg 1, bit 23): do the fp8 / f16 conversions saturate under it?
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__global__ void ovfl_kernel(const float* x, unsigned* out, int n) {
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    const int i = threadIdx.x;
    if (2 * i + 1 < n + 1) {
        s16x2 z = {0, 0};
        const s16x2 a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, x[2 * i], x[2 * i + 1], 1.0f, false);
        const unsigned b = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false) & 0xffffu;
        const f32x2_t v = {x[2 * i], x[2 * i + 1]};
        const f16x2_t h = __builtin_convertvector(v, f16x2_t);
        out[3 * i] = __builtin_bit_cast(unsigned, a) & 0xffffu;
        out[3 * i + 1] = b;
        out[3 * i + 2] = __builtin_bit_cast(unsigned, h);
    }
}

static float e4m3_to_f32(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7) r = NAN;
    else if (e == 0) r = ldexpf((float)m, -9);
    else r = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -r : r;
}
static unsigned char f32_to_e4m3(float x) {      // round to nearest even, saturating (host model)
    unsigned char best = 0; float bd = INFINITY;
    const float ax = fabsf(x);
    for (int v = 0; v < 127; ++v) {               // magnitudes; 0x7f is NaN
        const float dd = fabsf(e4m3_to_f32((unsigned char)v) - ax);
        if (dd < bd || (dd == bd && !(v & 1) && (best & 1))) { bd = dd; best = (unsigned char)v; }
    }
    return (unsigned char)(best | (std::signbit(x) ? 0x80 : 0));
}

struct Case { std::vector<unsigned char> A, B; std::vector<int> sa, sb; std::vector<float> C; };   // A[i][k], B[k][j] as fp8 bytes, scales per (row, half)

static int fails = 0;
static void report(const char* what, bool ok, const char* detail = "") {
    printf("%s  %s %s\n", ok ? "PASS" : "FAIL", what, detail);
    if (!ok) ++fails;
}

// run one MFMA with logical A (32 x 64), B (64 x 32), per-(row, k block) scale bytes, C; returns D (32 x 32) under the ASSUMED maps
static std::vector<float> run(const std::vector<unsigned char>& A, const std::vector<unsigned char>& B, const std::vector<int>& SA, const std::vector<int>& SB,
                              const std::vector<float>& C) {
    std::vector<unsigned char> la(64 * 32), lb(64 * 32);
    std::vector<int> lsa(64), lsb(64);
    std::vector<float> lc(64 * 16), ld(64 * 16);
    for (int l = 0; l < 64; ++l) {
        const int ij = l & 31, h = l >> 5;
        for (int e = 0; e < 32; ++e) { la[l * 32 + e] = A[ij * 64 + 32 * h + e]; lb[l * 32 + e] = B[(32 * h + e) * 32 + ij]; }
        lsa[l] = SA[ij * 2 + h]; lsb[l] = SB[ij * 2 + h];
        for (int r = 0; r < 16; ++r) lc[l * 16 + r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + ij];
    }
    unsigned char *da, *db; int *dsa, *dsb; float *dc, *dd;
    hipMalloc(&da, la.size()); hipMalloc(&db, lb.size()); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 4096); hipMalloc(&dd, 4096);
    hipMemcpy(da, la.data(), la.size(), hipMemcpyHostToDevice); hipMemcpy(db, lb.data(), lb.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsa, lsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, lsb.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dc, lc.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, dd);
    hipMemcpy(ld.data(), dd, 4096, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dc); hipFree(dd);
    std::vector<float> D(32 * 32);
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = ld[l * 16 + r];
    return D;
}

int main() {
    // ---- 1. conversion
    {
        const float xs[] = {1.0f, 2.0f, 0.3f, -3.3f, 448.0f, 460.0f, 480.0f, 1e6f, 0.001953125f, 0.0009765625f, 1.0625f, 1.1875f, 0.0029296875f, -0.0f, 17.0f, 19.0f};
        const int n = 16;
        float* dx; unsigned* dout;
        hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
        hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(n / 2), 0, 0, dx, dout, n);
        unsigned out[8];
        hipMemcpy(out, dout, 32, hipMemcpyDeviceToHost);
        bool ok = true;
        for (int i = 0; i < n; ++i) {
            const unsigned char got = (out[i / 2] >> (8 * (i & 1))) & 0xff, want = f32_to_e4m3(xs[i]);
            const bool in_range = fabsf(xs[i]) <= 448.0f;
            printf("     cvt_pk_fp8_f32(%g) = 0x%02x (%g)   host RNE-saturating model 0x%02x (%g)%s\n", xs[i], got, e4m3_to_f32(got), want, e4m3_to_f32(want),
                   in_range ? "" : "   [beyond 448: the kernel clamps with v_med3_f32 first]");
            if (in_range && got != want) ok = false;
        }
        report("1. v_cvt_pk_fp8_f32 = OCP e4m3fn, round to nearest even, inside +-448", ok);
    }
    // ---- 6. v_cvt_scalef32_pk_fp8_f32 (the conversion with the pre-scale folded in): value / scale?  value * scale?  saturation?
    {
        const float xs[] = {1.0f, 2.0f, 0.3f, -3.3f, 448.0f, 500.0f, 1e6f, -1e6f, 0.001953125f, 0.0009765625f, 100.0f, 200.0f, 7.0f, -0.0f, 17.0f, 19.0f};
        const int n = 16;
        float* dx; unsigned* dout;
        hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
        hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
        for (float scale : {1.0f, 4.0f, 0.25f, 3.0f}) {
            hipLaunchKernelGGL(cvt_scale_kernel, dim3(1), dim3(n / 2), 0, 0, dx, dout, n, scale);
            unsigned out[8];
            hipMemcpy(out, dout, 32, hipMemcpyDeviceToHost);
            printf("     cvt_scalef32_pk_fp8_f32, scale %g:", scale);
            bool div_ok = true, hi_ok = true;
            for (int i = 0; i < n; ++i) {
                const unsigned char got = (out[i / 2] >> (8 * (i & 1))) & 0xff, hi = (out[i / 2] >> (16 + 8 * ((i & 1) ^ 1))) & 0xff;
                printf(" %g->%g", xs[i], e4m3_to_f32(got));
                const float pw = ldexpf(1.0f, ilogbf(scale));                 // only the exponent of `scale` counts (an E8M0 scale in f32 clothing)
                const float want = xs[i] / pw;
                if (fabsf(want) <= 448.0f ? got != f32_to_e4m3(want) : (fabsf(want) >= 480.0f && (got & 0x7f) != 0x7f)) div_ok = false;
                if (hi != got) hi_ok = false;
            }
            printf("\n");
            char name[128];
            snprintf(name, sizeof(name), "6. scale %g: result = RNE e4m3(value / 2^floor(log2 scale)); NO saturation (NaN from 480 on): clamp first", scale);
            report(name, div_ok);
            if (!hi_ok) report("6. word_sel = true writes the same bytes into the upper half (old lower half kept)", false);
        }
    }
    // ---- 7. MODE.FP16_OVFL = 1: saturating conversions?
    {
        const float xs[] = {1.0f, 447.0f, 460.0f, 500.0f, 1e6f, -1e6f, 70000.0f, -70000.0f, INFINITY, NAN};
        const int n = 10;
        float* dx; unsigned* dout;
        hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 6);
        hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(ovfl_kernel, dim3(1), dim3(n / 2), 0, 0, dx, dout, n);
        unsigned out[15];
        hipMemcpy(out, dout, 60, hipMemcpyDeviceToHost);
        bool sat8 = true, sat8p = true, sat16 = true;
        printf("     under MODE.FP16_OVFL = 1:\n");
        for (int i = 0; i < n; ++i) {
            const unsigned char a = (out[3 * (i / 2)] >> (8 * (i & 1))) & 0xff, b = (out[3 * (i / 2) + 1] >> (8 * (i & 1))) & 0xff;
            const unsigned short h = (out[3 * (i / 2) + 2] >> (16 * (i & 1))) & 0xffff;
            _Float16 hf; memcpy(&hf, &h, 2);
            printf("       %g -> cvt_scalef32_pk_fp8 %g, cvt_pk_fp8 %g, f16 %g\n", xs[i], e4m3_to_f32(a), e4m3_to_f32(b), (float)hf);
            if (xs[i] == xs[i] && fabsf(xs[i]) > 448.0f && fabsf(xs[i]) < INFINITY) {
                if (fabsf(e4m3_to_f32(a)) != 448.0f) sat8 = false;
                if (fabsf(e4m3_to_f32(b)) != 448.0f) sat8p = false;
                if (fabsf(xs[i]) > 65504.0f && fabsf((float)hf) != 65504.0f) sat16 = false;
            }
        }
        printf("     -> v_cvt_scalef32_pk_fp8_f32 saturates: %s, v_cvt_pk_fp8_f32 saturates: %s, v_cvt_pk_f16_f32 saturates: %s (informational)\n", sat8 ? "yes" : "no", sat8p ? "yes" : "no", sat16 ? "yes" : "no");
    }
    const unsigned char ONE = f32_to_e4m3(1.0f);
    std::vector<int> S0(64, 127);
    std::vector<float> C0(1024, 0.f);
    // ---- 2. operand map
    {
        // A[3][k] = 1 for k < 32 (lane half 0 only), B[k][5] = 1 for all k  ->  D[3][5] = 32
        std::vector<unsigned char> A(32 * 64, 0), B(64 * 32, 0);
        for (int k = 0; k < 32; ++k) A[3 * 64 + k] = ONE;
        for (int k = 0; k < 64; ++k) B[k * 32 + 5] = ONE;
        auto D = run(A, B, S0, S0, C0);
        bool ok = D[3 * 32 + 5] == 32.f;
        for (int i = 0; i < 1024; ++i) if (i != 3 * 32 + 5 && D[i] != 0.f) ok = false;
        report("2a. A row 3 (k < 32) x B column 5 (all k) -> D[3][5] = 32, everything else 0 (also pins the C/D map, fact 4)", ok);
        // ... B only k >= 32 -> 0
        std::fill(B.begin(), B.end(), 0);
        for (int k = 32; k < 64; ++k) B[k * 32 + 5] = ONE;
        D = run(A, B, S0, S0, C0);
        ok = true;
        for (int i = 0; i < 1024; ++i) if (D[i] != 0.f) ok = false;
        report("2b. A k < 32 against B k >= 32 -> 0: lane halves never pair across", ok);
        // single K index: A[7][k0] x B[k1][9] = (k0 == k1)
        ok = true;
        for (int k0 : {0, 5, 15, 16, 31, 32, 47, 63})
            for (int k1 : {0, 5, 6, 16, 31, 32, 48, 63}) {
                std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
                A[7 * 64 + k0] = f32_to_e4m3(2.0f); B[k1 * 32 + 9] = f32_to_e4m3(3.0f);
                D = run(A, B, S0, S0, C0);
                if (D[7 * 32 + 9] != (k0 == k1 ? 6.f : 0.f)) { ok = false; printf("     k0 %d k1 %d -> %g\n", k0, k1, D[7 * 32 + 9]); }
            }
        report("2c. byte e of lane half h pairs with byte e of lane half h only (K index = 32 h + e on both operands)", ok);
    }
    // ---- 3. scales
    {
        std::vector<unsigned char> A(32 * 64, ONE), B(64 * 32, ONE);
        std::vector<int> SA(64, 127), SB(64, 127);
        for (int i = 0; i < 32; ++i) { SA[i * 2 + 0] = 127 - 3; SB[i * 2 + 1] = 127 - 5; }     // A: k < 32 scaled 2^-3; B: k >= 32 scaled 2^-5
        auto D = run(A, B, SA, SB, C0);
        bool ok = true;
        for (int i = 0; i < 1024; ++i) if (D[i] != 32.f / 8 + 32.f / 32) ok = false;
        printf("     D[0][0] = %g (expected 32 * 2^-3 + 32 * 2^-5 = 5)\n", D[0]);
        report("3a. a lane's scale byte applies to its own 32 K values (per operand, per lane half)", ok);
        // row-dependent scale
        for (int i = 0; i < 32; ++i) { SA[i * 2 + 0] = 127 - (i & 7); SA[i * 2 + 1] = 127; SB[i * 2 + 0] = 127; SB[i * 2 + 1] = 127; }
        D = run(A, B, SA, SB, C0);
        ok = true;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (D[i * 32 + j] != 32.f * ldexpf(1.f, -(i & 7)) + 32.f) ok = false;
        report("3b. A scales are per ROW (lane & 31), B scales per COLUMN", ok);
        // upper bytes of the scale register are ignored with op_sel 0
        for (int i = 0; i < 64; ++i) { SA[i] = 127 | 0x11223300; SB[i] = 127 | 0x7f000000; }
        D = run(A, B, SA, SB, C0);
        ok = true;
        for (int i = 0; i < 1024; ++i) if (D[i] != 64.f) ok = false;
        report("3c. op_sel 0 reads byte 0 of the scale register only", ok);
        // what the f16x2 kernel does: ONE scale for every lane of A, 1.0 for B
        for (int i = 0; i < 64; ++i) { SA[i] = 127 - 23; SB[i] = 127; }
        D = run(A, B, SA, SB, C0);
        ok = true;
        for (int i = 0; i < 1024; ++i) if (D[i] != 64.f * ldexpf(1.f, -23)) ok = false;
        report("3d. uniform A scale 2^-23, B scale 1 -> 64 * 2^-23 everywhere (the kernel's use)", ok);
        for (int i = 0; i < 32; ++i) { SA[i * 2 + 0] = 127; SA[i * 2 + 1] = 127 - (i & 7); SB[i * 2 + 0] = 127; SB[i * 2 + 1] = 127; }
        D = run(A, B, SA, SB, C0);
        ok = true;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (D[i * 32 + j] != 32.f * ldexpf(1.f, -(i & 7)) + 32.f) ok = false;
        report("3e. A scales per row in lane half 1 too", ok);
        for (int i = 0; i < 32; ++i) { SA[i * 2 + 0] = 127; SA[i * 2 + 1] = 127; SB[i * 2 + 0] = 127 - (i & 3); SB[i * 2 + 1] = 127 - 2 * (i & 3); }
        D = run(A, B, SA, SB, C0);
        ok = true;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (D[i * 32 + j] != 32.f * ldexpf(1.f, -(j & 3)) + 32.f * ldexpf(1.f, -2 * (j & 3))) ok = false;
        printf("     D[0][0..3] = %g %g %g %g (expected 64 24 10 4.5)\n", D[0], D[1], D[2], D[3]);
        report("3f. B scales per column, both lane halves", ok);
        for (int i = 0; i < 32; ++i) { SA[i * 2 + 0] = 127 - (i & 1); SA[i * 2 + 1] = 127 - (i & 1); SB[i * 2 + 0] = 127 - (i & 3); SB[i * 2 + 1] = 127 - (i & 3); }
        D = run(A, B, SA, SB, C0);
        ok = true;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (D[i * 32 + j] != 64.f * ldexpf(1.f, -(i & 1) - (j & 3))) ok = false;
        printf("     D[1][0..3] = %g %g %g %g (expected 32 16 8 4)\n", D[32], D[33], D[34], D[35]);
        report("3g. row scales of A and column scales of B together", ok);
    }
    // ---- 3h. which scale does K index (lane half h, byte e) take?  (all-ones operands cannot tell)
    {
        std::vector<unsigned char> A(32 * 64, 0), B(64 * 32, 0);
        std::vector<int> SA(64), SB(64, 127);
        for (int i = 0; i < 32; ++i) { SA[i * 2 + 0] = 127 - 1; SA[i * 2 + 1] = 127 - 2; }      // lanes 0..31 say 2^-1, lanes 32..63 say 2^-2
        bool ok = true;
        printf("     scale taken by A element (lane half h, byte e), lanes 0-31 offering 2^-1 and lanes 32-63 offering 2^-2:\n     ");
        for (int k0 = 0; k0 < 64; ++k0) {
            std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
            A[7 * 64 + k0] = f32_to_e4m3(2.0f); B[k0 * 32 + 9] = f32_to_e4m3(4.0f);
            auto D = run(A, B, SA, SB, C0);
            const float got = D[7 * 32 + 9];
            const int which = got == 4.f ? 0 : got == 2.f ? 1 : -1;
            printf("%d", which);
            if (which != (k0 % 32) / 16) ok = false;
        }
        printf("   (0 = the scale of lanes 0-31, 1 = of lanes 32-63; index = 32 h + e)\n");
        report("3h. element (h, e) takes the scale of lanes 0-31 for e < 16 and of lanes 32-63 for e >= 16 (block = byte-position half)", ok);
    }
    // ---- 4. subnormal elements
    {
        std::vector<unsigned char> A(32 * 64, 0x01), B(64 * 32, 0x7e);      // 2^-9 x 448
        auto D = run(A, B, S0, S0, C0);
        printf("     A = 2^-9 (subnormal e4m3) x B = 448, K = 64: D[0][0] = %g (56 if subnormals are kept, 0 if flushed)\n", D[0]);
        report("4a. fp8 subnormal elements are not flushed", D[0] == 56.f);
        std::vector<unsigned char> A1(32 * 64, ONE), B1(64 * 32, ONE);
        std::vector<int> SA(64), SB(64);
        unsigned rng = 777u;
        auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        for (int i = 0; i < 64; ++i) { SA[i] = 127 - (int)(next() % 12); SB[i] = 127 - (int)(next() % 12); }
        D = run(A1, B1, SA, SB, C0);
        bool ok = true; int shown = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            const float want = 32.f * ldexpf(1.f, SA[i * 2] + SB[j * 2] - 254) + 32.f * ldexpf(1.f, SA[i * 2 + 1] + SB[j * 2 + 1] - 254);
            if (D[i * 32 + j] != want) { ok = false; if (shown++ < 4) printf("     D[%d][%d] = %g, expected %g (sa %d %d sb %d %d)\n", i, j, D[i * 32 + j], want, SA[i * 2], SA[i * 2 + 1], SB[j * 2], SB[j * 2 + 1]); }
        }
        report("4b. random per-(row, half) / per-(column, half) scales on all-ones operands", ok);
    }
    // ---- 5. accumulation
    for (int variant = 0; variant < 8; ++variant) {      // (4..7: the same without subnormal elements)      // 0: unit scales, C = 0; 1: + random C; 2: random scales, C = 0; 3: everything
        std::vector<unsigned char> A(32 * 64), B(64 * 32);
        std::vector<float> C(1024, 0.f);
        unsigned rng = 12345u;
        auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        for (auto& v : A) { v = (unsigned char)(next() & 0xff); if ((v & 0x7f) == 0x7f) v = 0; if ((variant & 4) && !(v & 0x78)) v |= 0x08; }
        for (auto& v : B) { v = (unsigned char)(next() & 0xff); if ((v & 0x7f) == 0x7f) v = 0; if ((variant & 4) && !(v & 0x78)) v |= 0x08; }
        if (variant & 1) for (auto& v : C) v = (float)((int)(next() % 2001) - 1000) / 16.f;
        std::vector<int> SA(64, 127), SB(64, 127);
        if (variant & 2) for (int i = 0; i < 64; ++i) { SA[i] = 127 - (int)(next() % 12); SB[i] = 127 - (int)(next() % 12); }
        auto D = run(A, B, SA, SB, C);
        double worst = 0; int shown = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = C[i * 32 + j], mag = fabs(ref), big = fabs(ref);
                for (int k = 0; k < 64; ++k) {
                    const double t = (double)e4m3_to_f32(A[i * 64 + k]) * ldexp(1.0, SA[i * 2 + (k % 32) / 16] - 127) * (double)e4m3_to_f32(B[k * 32 + j]) * ldexp(1.0, SB[j * 2 + (k % 32) / 16] - 127);
                    ref += t; mag += fabs(t); big = fmax(big, fabs(t));
                }
                const double e = fabs(D[i * 32 + j] - ref) / big;
                if (e > 1e-4 && shown < 4) { printf("     variant %d D[%d][%d] = %.9g, f64 reference %.9g (sum |terms| %.4g)\n", variant, i, j, D[i * 32 + j], ref, mag); ++shown; }
                worst = fmax(worst, e);
            }
        printf("     variant %d (%s scales, %s C): worst |D - f64 reference| / max |term| = %.3g = 2^%.1f\n", variant, variant & 2 ? "random" : "unit", variant & 1 ? "random" : "zero", worst, log2(worst));
        char name[96];
        snprintf(name, sizeof(name), "5.%d accumulation error below 2^-11 of the largest term", variant);
        report(name, worst < 4.9e-4);
    }
    printf("%s (%d failed)\n", fails ? "MX PROBE FAILED" : "MX PROBE OK", fails);
    return fails ? 1 : 0;
}
