// Instantiates the whole-bottleneck kernel (conv_block_impl.h) for the two float32-storage x3 types; called by the C ABI in conv_igemm.hip.
#include "conv_block_impl.h"
#include "conv_igemm_types.h"

int gpp_block_dispatch_f16x3(gpp_conv_desc& d1, gpp_conv_desc& d2, gpp_conv_desc& d3, int tile, hipStream_t st)
{
    return dispatch_block_x3<GPP_F16X3>(d1, d2, d3, tile, st);
}

int gpp_block_dispatch_bf16x3(gpp_conv_desc& d1, gpp_conv_desc& d2, gpp_conv_desc& d3, int tile, hipStream_t st)
{
    return dispatch_block_x3<GPP_BF16X3>(d1, d2, d3, tile, st);
}
