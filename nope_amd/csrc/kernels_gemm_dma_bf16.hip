// conv_gemm_dma_kernel<bf16_t, ...>, the 128 x 192 tile every default launch uses (incl. the GEGLU-epilogue instantiation).
// (one translation unit per element type: see conv_gemm_dma.h)
#include "conv_gemm_dma.h"

namespace nope {

void launch_conv_dma_bf16(const void* params, dim3 grid, hipStream_t s) {
    launch_dma<bf16_t, 128, 2, 128>(*reinterpret_cast<const ConvParams*>(params), grid, s);
}

}  // namespace nope
