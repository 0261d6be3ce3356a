// Instantiates the implicit-GEMM convolution kernels (conv_igemm_impl.h) for one element type: GPP_BF16X3
// (float32 storage, three bf16 matrix products per float32 product).
#include "conv_igemm_impl.h"
#include "conv_igemm_types.h"

int gpp_conv_dispatch_bf16x3(gpp_conv_desc& d, hipStream_t st) { return dispatch<GPP_BF16X3>(d, st); }
