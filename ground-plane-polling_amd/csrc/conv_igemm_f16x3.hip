// Instantiates the implicit-GEMM convolution kernels (conv_igemm_impl.h) for one element type: GPP_F16X3
// (float32 storage, three IEEE-half matrix products per float32 product: 11 + 11 significant bits per operand).
#include "conv_igemm_impl.h"
#include "conv_igemm_types.h"

// the library's per-device range-event counter: what a launch adds to when its descriptor names no counter of its own
__device__ unsigned long long g_x3_range_events = 0;

int gpp_conv_dispatch_f16x3(gpp_conv_desc& d, hipStream_t st) { return dispatch<GPP_F16X3>(d, st); }

int gpp_tail_dispatch_f16x3(gpp_conv_desc& d1, gpp_conv_desc& d2, int tile_rows, hipStream_t st)
{
    return dispatch_tail_x3<GPP_F16X3>(d1, d2, tile_rows, st);
}

// the range ledger of this type (conv_igemm_impl.h: g_x3_range_events): read, optionally reset; synchronises the device
int gpp_x3_range_events_f16x3(unsigned long long* host_count, int reset)
{
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    unsigned long long v = 0;
    e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_x3_range_events), sizeof v);
    if (e != hipSuccess) return (int)e;
    if (host_count) *host_count = v;
    if (reset) {
        const unsigned long long zero = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_x3_range_events), &zero, sizeof zero);
        if (e != hipSuccess) return (int)e;
    }
    return GPP_OK;
}

// the counter's value at this point of the stream, written to device (or device-visible host) memory: no synchronisation
// (counter: the caller's slot, or NULL = the library's per-device counter)
__global__ void x3_range_snapshot_kernel(const unsigned long long* src, unsigned long long* dst) { *dst = src ? *src : g_x3_range_events; }

int gpp_x3_range_snapshot_f16x3(const unsigned long long* counter, unsigned long long* device_count, hipStream_t st)
{
    hipLaunchKernelGGL(x3_range_snapshot_kernel, dim3(1), dim3(1), 0, st, counter, device_count);
    return (int)hipGetLastError();
}

// device address of the counter on the current device (the x3 stem of stem.hip counts into it too: its float32 output map is split --
// and clamped -- by the first bottleneck's loop, which has no epilogue of its own to count in)
// (looked up once per device ordinal: every f16x3 launch without a counter of its own asks)
unsigned long long* gpp_x3_range_counter_f16x3()
{
    static std::atomic<unsigned long long*> counter_of_device[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    unsigned long long* c = counter_of_device[dev].load(std::memory_order_acquire);
    if (c) return c;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_x3_range_events)) != hipSuccess) return nullptr;
    counter_of_device[dev].store((unsigned long long*)p, std::memory_order_release);
    return (unsigned long long*)p;
}
