// Instantiates the implicit-GEMM convolution kernels (conv_igemm_impl.h) for one element type: GPP_F16.
#include "conv_igemm_impl.h"
#include "conv_igemm_types.h"

int gpp_conv_dispatch_f16(gpp_conv_desc& d, hipStream_t st)
{
    return dispatch<GPP_F16>(d, st);
}

int gpp_tail_dispatch_f16(gpp_conv_desc& d1, gpp_conv_desc& d2, int tile_rows, hipStream_t st)
{
    return dispatch_tail<GPP_F16>(d1, d2, tile_rows, st);
}
