// conv_gemm_dma_kernel<f32s_t, ...>: f32 storage, three bf16 MFMA passes per product.
// (one translation unit per element type: see conv_gemm_dma.h)
#include "conv_gemm_dma.h"

namespace nope {

void launch_conv_dma_bf16x3(const void* params, dim3 grid, hipStream_t s) {
    launch_dma<f32s_t, 128, 2, 128>(*reinterpret_cast<const ConvParams*>(params), grid, s);
}

}  // namespace nope
