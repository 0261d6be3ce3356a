// Per-element-type entry points of the convolution kernels (one translation unit each, conv_igemm_{bf16,f16,f32}.hip);
// called by the C ABI in conv_igemm.hip.  Internal to the library.
#ifndef GPP_CONV_IGEMM_TYPES_H_
#define GPP_CONV_IGEMM_TYPES_H_

#include <hip/hip_runtime.h>

#include "gpp.h"

int gpp_conv_dispatch_bf16(gpp_conv_desc& d, hipStream_t st);
int gpp_conv_dispatch_f16(gpp_conv_desc& d, hipStream_t st);
int gpp_conv_dispatch_f32(gpp_conv_desc& d, hipStream_t st);
int gpp_conv_dispatch_bf16x3(gpp_conv_desc& d, hipStream_t st);
int gpp_conv_dispatch_f16x3(gpp_conv_desc& d, hipStream_t st);
int gpp_tail_dispatch_bf16(gpp_conv_desc& d1, gpp_conv_desc& d2, int tile_rows, hipStream_t st);
int gpp_tail_dispatch_f16(gpp_conv_desc& d1, gpp_conv_desc& d2, int tile_rows, hipStream_t st);
int gpp_tail_dispatch_bf16x3(gpp_conv_desc& d1, gpp_conv_desc& d2, int tile_rows, hipStream_t st);
int gpp_tail_dispatch_f16x3(gpp_conv_desc& d1, gpp_conv_desc& d2, int tile_rows, hipStream_t st);
int gpp_block_dispatch_f16x3(gpp_conv_desc& d1, gpp_conv_desc& d2, gpp_conv_desc& d3, int tile, hipStream_t st);
int gpp_block_dispatch_bf16x3(gpp_conv_desc& d1, gpp_conv_desc& d2, gpp_conv_desc& d3, int tile, hipStream_t st);
int gpp_x3_range_events_f16x3(unsigned long long* host_count, int reset);
int gpp_x3_range_snapshot_f16x3(const unsigned long long* counter, unsigned long long* device_count, hipStream_t st);
unsigned long long* gpp_x3_range_counter_f16x3();

#endif
