// Instantiates the implicit-GEMM convolution kernels (conv_igemm_impl.h) for one element type: GPP_F32.
#include "conv_igemm_impl.h"
#include "conv_igemm_types.h"

int gpp_conv_dispatch_f32(gpp_conv_desc& d, hipStream_t st) { return dispatch<GPP_F32>(d, st); }
