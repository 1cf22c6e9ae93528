// Store-pattern probe: what write bandwidth does the conv epilogues' access pattern get from this board, against longer contiguous runs?
// A (M x C) f32 matrix is written tile by tile exactly as the 128 x 192 kernels walk it (512 workgroups of 4 waves, each wave a 64 x 96 block
// of a 128 x 192 tile), nothing is read:
//   pattern 0  the wide epilogue's: per instruction 8 rows x 128 B (a lane keeps one 16-byte column chunk, rows 8 apart), three 32-column passes
//   pattern 1  per instruction 2 2/3 rows x 384 B: the wave's 96 columns of a row contiguous
//   pattern 2  per instruction 1 1/3 rows x 768 B: the two waves of a row block interleaved as if one wrote all 192 columns of the tile
//   pattern 3  linear: the matrix as one stream (what a fill reaches)
// optionally with non-temporal stores (nt).   hipcc --offload-arch=gfx950 -O3 -o tools/probes/store_pattern_probe tools/probes/store_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool NT>
__global__ __launch_bounds__(256, 2) void store_kernel(float* __restrict__ out, int M, int C, int tiles_n, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    auto st = [&](size_t off) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + off)); else *reinterpret_cast<f32x4*>(out + off) = v; };
    if (PAT == 3) {
        const size_t nvec = (size_t)M * C / 4;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) st(i * 4);
        return;
    }
    const int nblocks = (M / 128) * tiles_n;
    for (int it = 0; it < iters; ++it) {
        const int blk = blockIdx.x + it * gridDim.x;
        if (blk >= nblocks) break;
        const int tile_n = blk % tiles_n, tile_m = blk / tiles_n;
        const int m0 = tile_m * 128 + wm * 64, n0 = tile_n * 192 + wn * 96;
        if (PAT == 0) {
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                for (int c = 0; c < 8; ++c) st((size_t)(m0 + (lane >> 3) + 8 * c) * C + n0 + pass * 32 + (lane & 7) * 4);
        } else if (PAT == 1) {
#pragma unroll
            for (int k = 0; k < 24; ++k) { const int idx = k * 64 + lane; st((size_t)(m0 + idx / 24) * C + n0 + (idx % 24) * 4); }
        } else {
            const int mt = tile_m * 128, nt0 = tile_n * 192;        // the workgroup's 128 x 192 tile row-major, 4 waves x 24 instructions x 1 KiB
#pragma unroll
            for (int k = 0; k < 24; ++k) { const int idx = (k * 4 + wave) * 64 + lane; st((size_t)(mt + idx / 48) * C + nt0 + (idx % 48) * 4); }
        }
    }
}

template <int PAT, bool NT>
static float run(float* out, int M, int C, int reps) {
    const int tiles_n = C / 192, nblocks = (M / 128) * tiles_n;
    const int grid = 512, iters = (nblocks + grid - 1) / grid;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((store_kernel<PAT, NT>), dim3(PAT == 3 ? 2048 : grid), dim3(256), 0, 0, out, M, C, tiles_n, iters);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((store_kernel<PAT, NT>), dim3(PAT == 3 ? 2048 : grid), dim3(256), 0, 0, out, M, C, tiles_n, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

int main() {
    const int M = 524288;
    for (int C : {192, 384}) {
        float* out;
        const size_t bytes = (size_t)M * C * 4;
        if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(out, 0, bytes);
        printf("-- %d x %d f32 = %zu MB written\n", M, C, bytes / 1000000);
#define ROW(P, NT, what) do { const float ms = run<P, NT>(out, M, C, 20); printf("pattern %d nt=%d  %-40s %7.1f us  %5.2f TB/s\n", P, (int)NT, what, ms * 1e3, bytes / ms / 1e9); } while (0)
        ROW(0, false, "8 rows x 128 B per instruction (epilogue)"); ROW(0, true, "8 rows x 128 B per instruction (epilogue)");
        ROW(1, false, "2.67 rows x 384 B"); ROW(1, true, "2.67 rows x 384 B");
        ROW(2, false, "1.33 rows x 768 B (tile row-major)"); ROW(2, true, "1.33 rows x 768 B (tile row-major)");
        ROW(3, false, "linear"); ROW(3, true, "linear");
        hipFree(out);
    }
    return 0;
}
