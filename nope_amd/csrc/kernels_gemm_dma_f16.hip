// conv_gemm_dma_kernel<f16_t, ...> (incl. the GEGLU-epilogue instantiation).
// (one translation unit per element type: see conv_gemm_dma.h)
#include "conv_gemm_dma.h"

namespace nope {

void launch_conv_dma_f16(const void* params, dim3 grid, hipStream_t s) {
    launch_dma<f16_t, 128, 2, 128>(*reinterpret_cast<const ConvParams*>(params), grid, s);
}

}  // namespace nope
