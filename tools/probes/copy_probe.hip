// Streaming ceiling probe: read one buffer, write another, 16 bytes per lane, in the launch shapes a GroupNorm-apply could take.
// Variants: U vectors in flight per lane; plain / non-temporal loads and stores; one pass per workgroup or a grid-stride walk with G workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/copy_probe tools/probes/copy_probe.hip && tools/probes/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < nvec; base += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256;
            if (i < nvec) v[u] = NTL ? __builtin_nontemporal_load(x + i) : x[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256;
            if (i < nvec) { if (NTS) __builtin_nontemporal_store(v[u], y + i); else y[i] = v[u]; }
        }
    }
}

template <int U, bool NTL, bool NTS>
static float run(const void* x, void* y, size_t bytes, int wg_per_cu, int reps) {
    const size_t nvec = bytes / 16;
    size_t blocks = (nvec + 256 * U - 1) / (256 * U);
    if (wg_per_cu > 0 && blocks > (size_t)256 * wg_per_cu) blocks = (size_t)256 * wg_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_kernel<U, NTL, NTS>), dim3((unsigned)blocks), dim3(256), 0, 0, (const u32x4*)x, (u32x4*)y, nvec);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_kernel<U, NTL, NTS>), dim3((unsigned)blocks), dim3(256), 0, 0, (const u32x4*)x, (u32x4*)y, nvec);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

int main() {
    const size_t sizes[] = {201u << 20, 403u << 20};
    for (size_t bytes : sizes) {
        void *x, *y;
        if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
        printf("-- %zu MB in, %zu MB out\n", bytes >> 20, bytes >> 20);
        const int reps = 20;
#define ROW(U, NTL, NTS, G)                                                                                                   \
    do {                                                                                                                      \
        const float ms = run<U, NTL, NTS>(x, y, bytes, G, reps);                                                              \
        printf("U=%d ntl=%d nts=%d wg/cu=%2d : %7.1f us  %5.2f TB/s\n", U, (int)NTL, (int)NTS, G, ms * 1e3, 2.0 * bytes / ms / 1e9); \
    } while (0)
        ROW(1, false, false, 0); ROW(2, false, false, 0); ROW(4, false, false, 0); ROW(8, false, false, 0);
        ROW(2, true, false, 0); ROW(4, true, false, 0);
        ROW(2, false, true, 0); ROW(4, false, true, 0);
        ROW(2, true, true, 0); ROW(4, true, true, 0); ROW(8, true, true, 0);
        ROW(2, true, true, 4); ROW(2, true, true, 8); ROW(4, true, true, 4); ROW(4, true, true, 8); ROW(4, true, true, 16);
        ROW(2, false, false, 8); ROW(4, false, false, 8);
        { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
          for (int i = 0; i < 3; ++i) hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0);
          hipEventRecord(a, 0);
          for (int i = 0; i < reps; ++i) hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0);
          hipEventRecord(b, 0); hipEventSynchronize(b);
          float ms = 0.f; hipEventElapsedTime(&ms, a, b); ms /= reps;
          printf("hipMemcpyAsync D2D            : %7.1f us  %5.2f TB/s\n", ms * 1e3, 2.0 * bytes / ms / 1e9); }
        hipFree(x); hipFree(y);
    }
    return 0;
}
