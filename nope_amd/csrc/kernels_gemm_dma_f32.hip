// conv_gemm_dma_kernel<float, ...>: the exact-f32 mode (128- and 256-row tiles).
// (one translation unit per element type: see conv_gemm_dma.h)
#include "conv_gemm_dma.h"

namespace nope {

void launch_conv_dma_f32(const void* params, int bm, dim3 grid, hipStream_t s) {
    const ConvParams& p = *reinterpret_cast<const ConvParams*>(params);
    if (bm == 256) launch_dma<float, 128, 2, 256>(p, grid, s);
    else launch_dma<float, 128, 2, 128>(p, grid, s);
}

}  // namespace nope
