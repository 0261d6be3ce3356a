// Instantiates the implicit-GEMM convolution kernels (conv_igemm_impl.h) for one element type: GPP_F16.
#include "conv_igemm_impl.h"
#include "conv_ring_impl.h"
#include "conv_igemm_types.h"

int gpp_conv_dispatch_f16(gpp_conv_desc& d, hipStream_t st)
{
    if (d.tile_hint >= 3000000) return dispatch_ring<GPP_F16>(d, st);      // loader-wavefront form (conv_ring_impl.h)
    return dispatch<GPP_F16>(d, st);
}

int gpp_tail_dispatch_f16(gpp_conv_desc& d1, gpp_conv_desc& d2, gpp_conv_desc* d3, int tile_rows, hipStream_t st)
{
    return dispatch_tail<GPP_F16>(d1, d2, d3, tile_rows, st);
}
