// conv_gemm_dma_kernel<bf16_t, ...>, tuning variants (NOPE_CONV_VARIANT): the 256-row / 8-wave tile (4) and 64-byte K stages (bit 1).
// (one translation unit per element type: see conv_gemm_dma.h)
#include "conv_gemm_dma.h"

namespace nope {

void launch_conv_dma_bf16_variant(const void* params, int bm, dim3 grid, hipStream_t s) {
    const ConvParams& p = *reinterpret_cast<const ConvParams*>(params);
    if (bm == 256) launch_dma<bf16_t, 128, 2, 256>(p, grid, s);
    else launch_dma<bf16_t, 64, 2, 128>(p, grid, s);
}

}  // namespace nope
