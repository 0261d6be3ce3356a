// NHWC convolution as an implicit GEMM on the gfx950 matrix cores, no im2col buffer, fused bias / residual
// (+ nearest resize) / ReLU epilogue.  Element types: bf16 / f16 operands on v_mfma_f32_16x16x32_{bf16,f16}, and
// float32 operands on v_mfma_f32_16x16x4_f32 (exact float32: every product rounded once, float32 accumulation -- the
// arithmetic type of the reference, keras.backend.floatx() = float32, /root/reference/keras_retinanet_3D/utils/image.py:47).
//
// This header holds the kernels and their launchers as templates over the element type; conv_igemm_{bf16,f16,f32}.hip
// instantiate one type each (three translation units compile in parallel), conv_igemm.hip holds the C ABI.
//
// Replaces the Conv2D / BatchNormalization(frozen) / Activation / Add / UpsampleLike nodes of
//   /root/reference/keras_retinanet_3D/models/retinanet.py:24-205  (heads, FPN)
//   keras_resnet bottleneck stack used at models/resnet.py:88-93     (third party)
// for every layer with C_in % 64 == 0 (everything except the 3-channel stem, csrc/stem.hip).
//
// GEMM view (one launch = up to 5 feature maps sharing the weights):
//   C[M = batch*Ho*Wo pixels][N = C_out] = A[M][K] * B[K][N],  K = (c_in chunk of CK, kh, kw, CK channels),
//   CK = 128 bytes of channels = 64 (16-bit) or 32 (float32)
//   A is never materialised: for K-step (channel chunk, tap) row m is the 128 contiguous bytes
//   in[b, oy*s - pt + kh, ox*s - pl + kw, c0:c0+CK]; outside the image the buffer descriptor's range check
//   makes the LDS-DMA deliver zeros.
//
// Work decomposition
//   block tile BM x BN with WM x WN wavefronts and a STAGES-deep LDS ring, three configurations:
//     256 x 256, 2 x 4 wavefronts (wave tile 128 x 64 = 8 x 4 MFMA 16x16 accumulators), 2 buffers, software-
//               pipelined + explicitly interleaved main loop (PIPE): the big 3x3 layers, 1 workgroup / CU
//     128 x 128 and 128 x 64, 2 x 2 wavefronts, 2 buffers: everything else, 2+ workgroups / CU, optional split-K
//   K-step = CK channels of one tap; A and B tiles (128-byte rows) go L2 -> LDS with buffer_load ... lds
//   (LDS-DMA, no VGPR staging): per-lane offset fixed per tap, per-step offset scalar.
//   LDS rows are XOR-swizzled in 16-byte chunks (chunk ^= row & 7) by permuting the *source* chunk each
//   lane fetches (the LDS-DMA destination is lane-linear): conflict-free ds_read_b128 fragment reads.
//   A 16-byte fragment is 8 consecutive K values of one 16x16x32 MFMA (16-bit), or 4 consecutive K values that feed
//   4 16x16x4 MFMAs (float32: MFMA e of chunk q sums k = {16j + 4q' + e}, both operands use the same map, so the
//   four together cover the 16 K values of the two chunk columns once).
//   Workgroup ids are remapped so that each XCD (private 4 MiB L2) owns a contiguous range of tiles;
//   the N-tiles of one M-tile are adjacent, so the activation rows are shared in that L2.
//   Epilogue: straight from the accumulators (operands swapped + host-interleaved weight rows give
//   every lane 8 consecutive output channels): bias, residual (+ nearest resize), ReLU, 16-byte stores.
//   Split-K (gridDim.y) writes float32 partial tiles; splitk_reduce_kernel sums them in split order and
//   runs the same epilogue.
#ifndef GPP_CONV_IGEMM_IMPL_H_
#define GPP_CONV_IGEMM_IMPL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "gpp.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// 8 float32 values with 16-byte alignment (two dwordx4 accesses): the float32 twin of the 16-byte bf16x8 / f16x8
struct f32x8 {
    f32x4 lo, hi;
    __device__ __forceinline__ float operator[](int e) const { return e < 4 ? lo[e] : hi[e - 4]; }
};

template <int DT> struct Elem;
template <> struct Elem<GPP_BF16> {
    using scalar = __bf16;
    using vec8 = bf16x8;          // 8 stored elements (epilogue granule)
    using frag = bf16x8;          // one 16-byte LDS fragment
    static constexpr int ESZ = 2;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ vec8 pack(const float (&v)[8])
    {
        vec8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (scalar)v[e];
        return o;
    }
};
template <> struct Elem<GPP_F16> {
    using scalar = _Float16;
    using vec8 = f16x8;
    using frag = f16x8;
    static constexpr int ESZ = 2;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ vec8 pack(const float (&v)[8])
    {
        vec8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (scalar)v[e];
        return o;
    }
};
template <> struct Elem<GPP_F32> {
    using scalar = float;
    using vec8 = f32x8;
    using frag = f32x4;           // 4 consecutive K values of one row
    static constexpr int ESZ = 4;
    // four exact float32 MFMAs (each D = fma chain over its 4 K values, one rounding per product)
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c)
    {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ vec8 pack(const float (&v)[8])
    {
        vec8 o;
        o.lo = (f32x4){v[0], v[1], v[2], v[3]};
        o.hi = (f32x4){v[4], v[5], v[6], v[7]};
        return o;
    }
};

constexpr int kRowBytes = 128;     // one K-step of one tile row: 64 two-byte or 32 four-byte elements

// LDS-DMA through a buffer descriptor: 16 bytes per lane from base + voffset + soffset to
// lds_dst_wave_base + lane*16.  voffset is per lane, soffset wave-uniform (SGPR), so the per-K-step
// address arithmetic is scalar; a lane whose voffset is out of range (kOutOfRange) gets zeros,
// which is how convolution padding is produced without a zero page or per-step predication.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, void* lds_dst_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16,
                                             voffset, soffset, 0, 0);
}

constexpr int kOutOfRange = (int)0x80000000;

#ifndef GPP_DMA_SPREAD
#define GPP_DMA_SPREAD 1
#endif
#ifndef GPP_X3_PRIO
#define GPP_X3_PRIO 1
#endif
#ifndef GPP_X3_WB
#define GPP_X3_WB 0
#endif
#ifndef GPP_X3_BEARLY
#define GPP_X3_BEARLY 0
#endif
#ifndef GPP_X3_TAP_EARLY
#define GPP_X3_TAP_EARLY 0
#endif
constexpr bool kSpreadDma = GPP_DMA_SPREAD != 0;

// Diagnostic build only (-DGPP_STAMPS, tools/bench_conv.py stamps): wave 0 of every workgroup writes the 100 MHz
// real-time counter at five points into a buffer of its own (handed in through the otherwise unused zero_page
// field when reserved bit 4 is set).  No output depends on it; the production build contains none of this.
// Order of the matrix instructions inside a fragment row: serpentine (even rows left to right, odd rows right to left), so that two consecutive
// MFMAs always share one operand.  The matrix pipe's clock under load depends on what toggles between instructions: on post-ReLU data a
// register-only x3 loop sustains 2075 TFLOP/s in this order against 1969 with both row ends changing (tools/micro/mfma_order.hip,
// profiles/r4/mfma_order.txt).  Per accumulator the three terms keep their order: the bytes do not change.
#ifndef GPP_X3_SERPENTINE
#define GPP_X3_SERPENTINE 2
#endif
#define GPP_SERP(g, j, n) ((GPP_X3_SERPENTINE && ((g) & 1)) ? (n) - 1 - (j) : (j))
#define GPP_SERP2(g, j, n) ((GPP_X3_SERPENTINE > 1 && ((g) & 1)) ? (n) - 1 - (j) : (j))      // the loops the compiler schedules itself

#ifdef GPP_STAMPS
#define GPP_STAMP(k)                                                                                          \
    do {                                                                                                      \
        if ((d.reserved & 16) && wave == 0 && lane == 0)                                                      \
            ((unsigned long long*)d.zero_page)[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define GPP_STAMP_END()                                                                                       \
    do {                                                                                                      \
        GPP_STAMP(3);                                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        GPP_STAMP(4);                                                                                         \
    } while (0)
#define GPP_ABL(flag) if (!(flag))
#else
#define GPP_ABL(flag)
#define GPP_STAMP(k) do { } while (0)
#define GPP_STAMP_END() do { } while (0)
#endif

// Bijective remap: blocks b and b+8 share an XCD; give each XCD a contiguous tile range.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// The two "three 16-bit matrix products per float32 product" types: x = hi + lo with hi = h(x), lo = h(x - hi) (x - hi is exact in
// float32), x * w ~ hi*whi + hi*wlo + lo*whi.  GPP_BF16X3: h = bfloat16, 8 + 8 significant bits, ~2^-16 per product, float32's
// range.  GPP_F16X3: h = IEEE half, 11 + 11 bits, ~2^-22 per product (float32: 2^-24); range: a finite activation beyond +-65504 is
// clamped where it is split -- and COUNTED (g_x3_range_events, gpp_x3_range_events): a clamped value is a wrong value, and the caller
// can ask whether one occurred --, a non-finite one stays non-finite (hi = x, lo = x - x), as it would in the float32 path; weights are
// scaled per output channel by a power of two so that both halves are normal halfs (gpp_conv_desc.out_scale undoes it in the epilogue).
template <int DT> constexpr bool kX3 = (DT == GPP_BF16X3 || DT == GPP_F16X3);
// GPP_F16X3: an epilogue that stores an 8-channel group with at least one value outside the half range (finite beyond +-65504, inf or NaN)
// adds one event to the counter the launch was given (gpp_conv_desc.range_counter: a plan's own 8-byte slot, or -- filled in by the
// entry points when the caller left it NULL -- the library's per-device counter, conv_igemm_f16x3.hip).
// every other type: placeholders so that discarded `if constexpr` branches still parse
template <int DT> struct X3Half {      // primary template
    using half = __bf16;
    using vec = bf16x8;
    static __device__ __forceinline__ f32x4 mfma(vec, vec, f32x4 c) { return c; }
    static __device__ __forceinline__ float clamp(float x) { return x; }
};

template <> struct X3Half<GPP_BF16X3> {
    using half = __bf16;
    using vec = bf16x8;
    static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float clamp(float x) { return x; }
};
template <> struct X3Half<GPP_F16X3> {
    using half = _Float16;
    using vec = f16x8;
    static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float clamp(float x) { return fminf(fmaxf(x, -65504.0f), 65504.0f); }
};
// The 8 values an epilogue is about to split: clamped into the half range (GPP_F16X3; the other types pass through).  The rare group
// that holds a value the clamp changed -- or a NaN, which compares unequal to everything -- takes the branch: the event is counted and
// non-finite values are put back, so that a NaN / inf is still one after the split instead of reading as +-65504.
template <int DT>
__device__ __forceinline__ void x3_range(float (&v)[8], unsigned long long* counter)
{
    if constexpr (DT == GPP_F16X3) {
        float c[8];
        bool changed = false;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            c[e] = X3Half<DT>::clamp(v[e]);
            changed |= (c[e] != v[e]);
        }
        if (__builtin_expect(changed, 0)) {
            atomicAdd(counter, 1ull);
#pragma unroll
            for (int e = 0; e < 8; ++e) c[e] = (fabsf(v[e]) <= 3.402823466e38f) ? c[e] : v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c[e];
    }
}
// Pre-split maps (gpp_conv_desc.x3_split): channels n .. n + 7 (n a multiple of 8) of the pixel whose float32-sized
// element offset is `base` live at byte (base + (n & ~31)) * 4 + (n & 31) * 2 (8 halves hi) and 64 bytes further (8 halves lo).
__device__ __forceinline__ const char* x3_addr(const void* buf, int64_t base, int n)
{
    return (const char*)buf + ((base + (n & ~31)) << 2) + ((n & 31) << 1);
}
template <int DT>
__device__ __forceinline__ void x3_unpack(const f32x4 hi_bits, const f32x4 lo_bits, float (&r)[8])
{
    union { f32x4 f; typename X3Half<DT>::vec b; } h, l;
    h.f = hi_bits;
    l.f = lo_bits;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (float)h.b[e] + (float)l.b[e];
}
template <int DT>
__device__ __forceinline__ void x3_store(void* buf, int64_t base, int n, float (&v)[8], unsigned long long* counter)
{
    using half = typename X3Half<DT>::half;
    typename X3Half<DT>::vec h, l;
    x3_range<DT>(v, counter);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e];
        h[e] = (half)x;
        l[e] = (half)(x - (float)h[e]);
    }
    char* p = (char*)x3_addr(buf, base, n);
    *(typename X3Half<DT>::vec*)p = h;
    *(typename X3Half<DT>::vec*)(p + 64) = l;
}

// Finish 8 consecutive output channels of one output pixel: (+ residual through the optional TF
// nearest resize) (+ ReLU), convert, store.  v already holds accumulator + bias.
template <int DT>
__device__ __forceinline__ void finish8(const gpp_conv_desc& d, float (&v)[8], int n, int64_t obase,
                                        const typename Elem<DT>::scalar* rrow)
{
    using vec8 = typename Elem<DT>::vec8;
    using scalar = typename Elem<DT>::scalar;
    const bool full = (n + 8 <= d.C_out);
    if constexpr (kX3<DT>) {
        if (rrow && (d.x3_split & GPP_X3_RES)) {            // pre-split shortcut map (whole 8-channel groups by construction)
            const char* p = x3_addr(rrow, 0, n);
            float r[8];
            x3_unpack<DT>(*(const f32x4*)p, *(const f32x4*)(p + 64), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
            rrow = nullptr;
        }
    }
    if (rrow) {
        if (full) {
            const vec8 rv = *(const vec8*)(rrow + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
        } else {
            for (int e = 0; e < 8 && n + e < d.C_out; ++e) v[e] += (float)rrow[n + e];
        }
    }
    if (d.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
    }
    if constexpr (kX3<DT>) {
        if (d.x3_split & GPP_X3_OUT) {                      // (validated: not out_f32, C_out a multiple of 32)
            x3_store<DT>(d.out, obase, n, v, (unsigned long long*)d.range_counter);
            return;
        }
    }
    if (d.out_f32) {
        float* dst = (float*)d.out + obase + n;
        if (full) {
            *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
            *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
            for (int e = 0; e < 8 && n + e < d.C_out; ++e) dst[e] = v[e];
        }
    } else {
        if constexpr (DT == GPP_F16X3) x3_range<DT>(v, (unsigned long long*)d.range_counter);   // an activation map kept as float32: checked here, split by its consumers
        scalar* dst = (scalar*)d.out + obase + n;
        if (full) {
            *(vec8*)dst = Elem<DT>::pack(v);
        } else {
            for (int e = 0; e < 8 && n + e < d.C_out; ++e) dst[e] = (scalar)v[e];
        }
    }
}

// finish8 with the residual already in registers (prefetched): same arithmetic, so the same bits.  Full groups only.
template <int DT>
__device__ __forceinline__ void finish8_pre(const gpp_conv_desc& d, float (&v)[8], int n, int64_t obase, bool has_res,
                                            const typename Elem<DT>::vec8 rv)
{
    using vec8 = typename Elem<DT>::vec8;
    using scalar = typename Elem<DT>::scalar;
    if (has_res) {
        bool done = false;
        if constexpr (kX3<DT>) {
            if (d.x3_split & GPP_X3_RES) {                  // the prefetched registers hold the raw [8 hi][8 lo] bits
                float r[8];
                x3_unpack<DT>(rv.lo, rv.hi, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
        }
    }
    if (d.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
    }
    if constexpr (kX3<DT>) {
        if (d.x3_split & GPP_X3_OUT) {
            x3_store<DT>(d.out, obase, n, v, (unsigned long long*)d.range_counter);
            return;
        }
    }
    if (d.out_f32) {
        float* dst = (float*)d.out + obase + n;
        *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
        *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
        if constexpr (DT == GPP_F16X3) x3_range<DT>(v, (unsigned long long*)d.range_counter);
        scalar* dst = (scalar*)d.out + obase + n;
        *(vec8*)dst = Elem<DT>::pack(v);
    }
}

// Where output row m of a group lives (and its residual row).
struct RowAddr { int64_t obase; int64_t rbase; };
__device__ __forceinline__ RowAddr row_addr(const gpp_conv_desc& d, int m, int HoWo, int W_out, int H_out, int H_res, int W_res,
                                            int64_t out_off, int64_t out_bs, int64_t res_off, int64_t res_bs)
{
    const int b = m / HoWo, p = m - b * HoWo;
    int64_t rp = p;
    if (d.residual && (H_res != H_out || W_res != W_out)) {
        const float sy = (float)H_res / (float)H_out, sx = (float)W_res / (float)W_out;
        const int oy = p / W_out, ox = p - oy * W_out;
        const int ry = min((int)floorf((float)oy * sy), H_res - 1);
        const int rx = min((int)floorf((float)ox * sx), W_res - 1);
        rp = (int64_t)ry * W_res + rx;
    }
    RowAddr a;
    a.obase = out_off + (int64_t)b * out_bs + (int64_t)p * d.out_pitch;
    a.rbase = res_off + (int64_t)b * res_bs + rp * d.res_pitch;
    return a;
}

// Output pixel (b, oy, ox) of GEMM row m, and the walk to row m + n without dividing again: a workgroup's setup used to
// spend two integer divisions (~35 VALU instructions each) on every staged / prefetched / stored row -- 1.3-2.4 us of the
// 7-17 us a small-tile workgroup lives (profiles/r2/small_layer_stamps.txt).
struct PixWalk {
    int b, oy, ox;
    __device__ __forceinline__ void init(int m, int HoWo, int W_out)
    {
        b = m / HoWo;
        const int p = m - b * HoWo;
        oy = p / W_out;
        ox = p - oy * W_out;
    }
    __device__ __forceinline__ void advance(int n, int H_out, int W_out)
    {
        ox += n;
        while (ox >= W_out) { ox -= W_out; ++oy; }
        while (oy >= H_out) { oy -= H_out; ++b; }
    }
};

__device__ __forceinline__ RowAddr row_addr_at(const gpp_conv_desc& d, const PixWalk& w, int W_out, int H_out, int H_res, int W_res,
                                               int64_t out_off, int64_t out_bs, int64_t res_off, int64_t res_bs)
{
    const int p = w.oy * W_out + w.ox;
    int64_t rp = p;
    if (d.residual && (H_res != H_out || W_res != W_out)) {
        const float sy = (float)H_res / (float)H_out, sx = (float)W_res / (float)W_out;
        const int ry = min((int)floorf((float)w.oy * sy), H_res - 1);
        const int rx = min((int)floorf((float)w.ox * sx), W_res - 1);
        rp = (int64_t)ry * W_res + rx;
    }
    RowAddr a;
    a.obase = out_off + (int64_t)w.b * out_bs + (int64_t)p * d.out_pitch;
    a.rbase = res_off + (int64_t)w.b * res_bs + rp * d.res_pitch;
    return a;
}

// float32 storage, three bf16 MFMAs per product (GPP_BF16X3): x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (x - hi is
// exact in float32), x * w ~ hi*whi + hi*wlo + lo*whi; the dropped lo*wlo term and the two roundings of lo leave a relative
// error of about 2^-16 per product -- 2^8 closer to float32 than plain bf16 operands, at a third of the bf16 matrix rate.
// Activations are split in registers on their way from LDS to the matrix pipe; weights are stored pre-split.
template <int DT> struct ElemX3 {
    using scalar = float;
    using vec8 = f32x8;
    using frag = f32x4;
    static constexpr int ESZ = 4;
    static __device__ __forceinline__ vec8 pack(const float (&v)[8]) { return Elem<GPP_F32>::pack(v); }
    static __device__ __forceinline__ f32x4 mfma(frag, frag, f32x4 c) { return c; }      // unused: the K-step has its own form
    // 8 consecutive float32 K values -> their hi and lo halves (round to nearest even)
    static __device__ __forceinline__ void split(const f32x4 x0, const f32x4 x1, typename X3Half<DT>::vec& hi, typename X3Half<DT>::vec& lo)
    {
        using half = typename X3Half<DT>::half;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = X3Half<DT>::clamp(e < 4 ? x0[e] : x1[e - 4]);
            const half h = (half)x;
            hi[e] = h;
            lo[e] = (half)(x - (float)h);
        }
    }
};
template <> struct Elem<GPP_BF16X3> : ElemX3<GPP_BF16X3> {};
template <> struct Elem<GPP_F16X3> : ElemX3<GPP_F16X3> {};

template <int DT> constexpr bool kF32Storage = (DT == GPP_F32 || DT == GPP_BF16X3 || DT == GPP_F16X3);

// activation rows a workgroup of NW wavefronts stages per K-step for a BM-row tile: whole 8-row LDS-DMA pieces per wavefront
constexpr int stage_rows(int BM, int NW) { return (BM + 8 * NW - 1) / (8 * NW) * (8 * NW); }

// XIN (GPP_BF16X3 only): the input map is pre-split (gpp_conv_desc.x3_split & GPP_X3_IN) -- a compile-time property of the kernel, so
// that the loop of either form carries no trace of the other (the 256 x 256 tile has no registers to spare for both)
template <int DT, int BM, int BN, int WM, int WN, int STAGES, bool PIPE, bool XIN = false>
__device__ __forceinline__ void conv_igemm_body(const gpp_conv_desc& d, const int block_x, const int grid_x)
{
    static_assert(!XIN || kX3<DT>, "pre-split input maps: GPP_BF16X3 / GPP_F16X3");
    using E = Elem<DT>;
    using vec8 = typename E::vec8;
    using frag = typename E::frag;
    using scalar = typename E::scalar;
    using xh8 = typename X3Half<DT>::vec;                // GPP_BF16X3 / GPP_F16X3: 8 halves of a fragment
    constexpr int ESZ = E::ESZ, CK = kRowBytes / ESZ;     // bytes per element, channels per K-step
    constexpr int NW = WM * WN;                          // wavefronts per workgroup
    constexpr int MF = BM / WM / 16, NF = BN / WN / 16;  // 16x16 accumulators per wave: MF x NF
    // BMS: activation rows STAGED per K-step = BM rounded up to a whole number of 8-row LDS-DMA pieces per wavefront.  BM = 224 / 160 on
    // 8 wavefronts (the short tiles of the mixed grid below) stage 256 / 192 rows and compute on the first BM of them: the extra rows
    // belong to the next tile (or lie past the end: zeros from the range check) and are never read from LDS.
    constexpr int BMS = stage_rows(BM, NW);
    constexpr int A_BYTES = BMS * kRowBytes, B_BYTES = BN * kRowBytes, STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = BMS / 8 / NW, B_IT = BN / 8 / NW;   // LDS-DMA instructions per wave per stage
    constexpr int PER_STAGE = A_IT + B_IT;
    constexpr int PF = STAGES - 1;                       // K-steps in flight ahead of the one computed
    static_assert(MF >= 1 && NF >= 1 && A_IT >= 1 && B_IT >= 1 && BMS % (8 * NW) == 0 && BN % (8 * NW) == 0 && BM % (16 * WM) == 0, "tile / wave shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    GPP_STAMP(0);

    // ---- which tile
    const int bid = xcd_remap(block_x, grid_x);
    const int n_tiles = (d.C_out + BN - 1) / BN;
    const int nt = bid % n_tiles, mt = bid / n_tiles;
    int tile_start = 0, row_begin = 0, H_in = 0, W_in = 0, H_out = 0, W_out = 0, H_res = 0, W_res = 0;
    int64_t in_off = 0, in_bs = 0, out_off = 0, out_bs = 0, res_off = 0, res_bs = 0;
#pragma unroll
    for (int q = 0; q < GPP_MAX_GROUPS; ++q) {
        if (q < d.n_groups && mt >= d.groups[q].tile_start) {
            tile_start = d.groups[q].tile_start;
            row_begin = d.groups[q].row_begin;
            H_in = d.groups[q].H_in; W_in = d.groups[q].W_in;
            H_out = d.groups[q].H_out; W_out = d.groups[q].W_out;
            H_res = d.groups[q].H_res; W_res = d.groups[q].W_res;
            in_off = d.groups[q].in_off; in_bs = d.groups[q].in_bstride;
            out_off = d.groups[q].out_off; out_bs = d.groups[q].out_bstride;
            res_off = d.groups[q].res_off; res_bs = d.groups[q].res_bstride;
        }
    }
    const int HoWo = H_out * W_out;
    const int Mg = d.batch * HoWo;
    const int m0 = row_begin + (mt - tile_start) * BM, n0 = nt * BN;       // (row_begin: 0 except in the second part of a mixed grid)
    const int Ktot = d.KH * d.KW * d.C_in;
    const int cpt = d.C_in / CK;                        // channel chunks per tap
    const int nk_total = d.KH * d.KW * cpt;
    // split-K: blockIdx.y owns K-steps [ks0, ks0 + nk); partial sums go to d.partial, the epilogue runs
    // in splitk_reduce_kernel
    // (32-bit: nk_total * nsplit is a few thousand at most; the unsplit case -- almost every launch -- divides nothing.  The
    // 64-bit divisions that stood here cost every workgroup of every layer a few hundred instructions of setup.)
    const int nsplit = gridDim.y, split = blockIdx.y;
    int ks0 = 0, nk = nk_total;
    if (nsplit > 1) {
        ks0 = (int)((unsigned)(nk_total * split) / (unsigned)nsplit);
        nk = (int)((unsigned)(nk_total * (split + 1)) / (unsigned)nsplit) - ks0;
    }

    // ---- staging bookkeeping: this lane owns LDS chunk (row srow of each 8-row piece, slot lane&7)
    // and fetches source chunk gchunk = slot ^ srow (inverse of the read swizzle).  Byte offsets are
    // relative to d.in / d.weight and go through buffer descriptors (32-bit, range checked).
    const int srow = lane >> 3;
    const int gchunk = (lane & 7) ^ srow;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.in, 0, d.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.weight, 0, d.weight_bytes, 0x00020000);
    // Per staged row: byte offset of its tap (0,0) source (may be negative at the border -- it is
    // only used when the tap is valid) and a validity mask: bit kh = input row iy0+kh inside the
    // image, bit 8+kw = input column ix0+kw inside.  Per K-step the source is base + (kh*W + kw)*pitch
    // (a scalar) when both bits are set, else kOutOfRange: ~4 VALU per row and step.
    // VOFF_INLINE (the 512 x 128 side of the dual-shape grid: 8 activation pieces per wavefront): the tap's offsets are not kept in
    // a register per piece but formed from (a_base, a_mask) and the tap's two scalars when the piece is issued -- the same
    // arithmetic, 8 registers less (that form spilled 8 registers of the dual kernel to scratch)
    constexpr bool VOFF_INLINE = PIPE && A_IT >= 8;
    int a_base[A_IT], a_mask[A_IT], a_voff[VOFF_INLINE ? 1 : A_IT];
    int tap_delta = 0, tap_need = 0;
    const int pitch2 = d.in_pitch * ESZ;                // bytes per input pixel
    {
        PixWalk pw;                                     // rows m, m + 8, m + 16, ...: one pair of divisions, then a walk
        pw.init(m0 + wave * A_IT * 8 + srow, HoWo, W_out);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + (wave * A_IT + i) * 8 + srow;
            a_mask[i] = 0;
            a_base[i] = 0;
            if (m < Mg) {
                const int iy0 = pw.oy * d.stride - d.pad_top, ix0 = pw.ox * d.stride - d.pad_left;
                // bit k: input row iy0 + k inside the image = k in [max(0, -iy0), min(KH, H_in - iy0)); columns likewise (no loops)
                const int rlo = max(0, -iy0), rhi = min(d.KH, max(0, H_in - iy0));
                const int clo = max(0, -ix0), chi = min(d.KW, max(0, W_in - ix0));
                const int rmask = rhi > rlo ? ((1 << rhi) - 1) & ~((1 << rlo) - 1) : 0;
                const int cmask = chi > clo ? ((1 << chi) - 1) & ~((1 << clo) - 1) : 0;
                a_mask[i] = rmask | (cmask << 8);
                a_base[i] = (int)((in_off + (int64_t)pw.b * in_bs) * ESZ) + gchunk * 16 + (iy0 * W_in + ix0) * pitch2;
            }
            if (i + 1 < A_IT) pw.advance(8, H_out, W_out);
        }
    }
    int w_voff[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) w_voff[i] = (n0 + (wave * B_IT + i) * 8 + srow) * Ktot * ESZ + gchunk * 16;

    auto set_tap = [&](int kh, int kw) {
        const int delta = (kh * W_in + kw) * pitch2;
        const int need = (1 << kh) | (1 << (8 + kw));
        if constexpr (VOFF_INLINE) {
            tap_delta = delta; tap_need = need;
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) a_voff[i] = ((a_mask[i] & need) == need) ? a_base[i] + delta : kOutOfRange;
        }
    };
    auto voff_of = [&](int i) {
        if constexpr (VOFF_INLINE) return ((a_mask[i] & tap_need) == tap_need) ? a_base[i] + tap_delta : kOutOfRange;
        else return a_voff[i];
    };
    // per K-step: only scalar offsets change (cc*128 bytes into the pixel, ks*128 bytes into the weight row)
    auto stage = [&](int buf, int cc, int ks) {
        unsigned char* sa = smem + buf * STAGE + wave * A_IT * 8 * kRowBytes;
        unsigned char* sb = smem + buf * STAGE + A_BYTES + wave * B_IT * 8 * kRowBytes;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16(in_rsrc, voff_of(i), cc * kRowBytes, sa + i * 8 * kRowBytes);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16(w_rsrc, w_voff[i], ks * kRowBytes, sb + i * 8 * kRowBytes);
    };

    // ---- fragment read offsets (bytes inside a stage)
    const int frow = lane & 15, fq = lane >> 4;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fq) ^ (frow & 7)) << 4;
        a_rd[kk] = (wm * (BM / WM) + frow) * kRowBytes + sw;
        b_rd[kk] = A_BYTES + (wn * (BN / WN) + frow) * kRowBytes + sw;
    }

    // GPP_BF16X3: a lane's 8 consecutive float32 K values (8 fq .. 8 fq + 7 of the 32-channel K-step) are chunks 2 fq and
    // 2 fq + 1 of the activation row; the weight row holds [32 bf16 hi | 32 bf16 lo], i.e. chunk fq and chunk 4 + fq = b_rd[0], b_rd[1]
    int a_rdx[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) a_rdx[h] = (wm * (BM / WM) + frow) * kRowBytes + (((2 * fq + h) ^ (frow & 7)) << 4);

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- shortcut prefetch (small non-pipelined tiles only: the layers that carry a residual are the 1x1 "branch2c" /
    // FPN lateral convs with 4-16 K-steps, where load -> wait -> add -> store in the epilogue is a large part of a
    // workgroup's life): the residual rows of this tile are requested now and are in registers when the loop ends
    // (a float32-storage row piece is 8 registers, a 16-bit one 4: the 160 x 128 / 128 x 160 float32 tiles spilled 60 - 150 registers
    // to scratch with the prefetch, tools/isa_audit.py)
    constexpr bool RESPRE = !PIPE && (MF * NF / 2 <= (ESZ == 4 ? 8 : 10));
    vec8 rpre[RESPRE ? MF : 1][RESPRE ? NF / 2 : 1];
    RowAddr ra_pre[RESPRE ? MF : 1];
    const bool use_pre = RESPRE && d.residual != nullptr && gridDim.y == 1 && (d.C_out & 7) == 0;
    if constexpr (RESPRE) {
        if (use_pre) {
            PixWalk pw;
            pw.init(m0 + wm * (BM / WM) + (lane & 15), HoWo, W_out);
            const PixWalk first = {0, 0, 0};
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int m = m0 + wm * (BM / WM) + i * 16 + (lane & 15);
                ra_pre[i] = row_addr_at(d, m < Mg ? pw : first, W_out, H_out, H_res, W_res, out_off, out_bs, res_off, res_bs);
                if (i + 1 < MF) pw.advance(16, H_out, W_out);
#pragma unroll
                for (int jj = 0; jj < NF / 2; ++jj) {
                    const int n = n0 + wn * (BN / WN) + jj * 32 + (lane >> 4) * 8;
                    const int nc = n < d.C_out ? n : 0;
                    bool raw = false;
                    if constexpr (kX3<DT>) {
                        if (d.x3_split & GPP_X3_RES) {          // pre-split shortcut map: the two 16-byte halves as they are
                            const char* p = x3_addr(d.residual, ra_pre[i].rbase, nc);
                            rpre[i][jj].lo = *(const f32x4*)p;
                            rpre[i][jj].hi = *(const f32x4*)(p + 64);
                            raw = true;
                        }
                    }
                    if (!raw) rpre[i][jj] = *(const vec8*)((const scalar*)d.residual + ra_pre[i].rbase + nc);
                }
            }
        }
    }

    // bias of this lane's output channels: the plain tiles fetch it here, under the main loop (their workgroups live 7-17 us
    // and used to pay this load's latency in the epilogue); the pipelined tiles have no registers to spare and fetch it there
    constexpr int COLS = BN / WN;                        // output channels owned by this wave
    static_assert(NF % 2 == 0, "N tiles come in interleaved pairs");
    float bias_v[NF / 2][8];
    // GPP_F16X3: the packed weights of output channel n are 2^k(n) times the real ones (both halves normal halfs); out_scale[n] =
    // 2^-k(n) brings the accumulator back before the bias is added (an exact multiplication)
    constexpr bool OSCALE = (DT == GPP_F16X3);
    float scale_v[OSCALE ? NF / 2 : 1][8];
    auto load_bias = [&]() {
#pragma unroll
        for (int jj = 0; jj < NF / 2; ++jj) {
            const int n = n0 + wn * COLS + jj * 32 + (lane >> 4) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) bias_v[jj][e] = (d.bias && n + e < d.C_out) ? d.bias[n + e] : 0.0f;
            if constexpr (OSCALE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) scale_v[jj][e] = (d.out_scale && n + e < d.C_out) ? d.out_scale[n + e] : 1.0f;
            }
        }
    };
    auto comb = [&](float a, int jj, int e) {
        if constexpr (OSCALE) return a * scale_v[jj][e] + bias_v[jj][e];
        else return a + bias_v[jj][e];
    };
    // (... and only where it costs at most 16 registers beside at most 96 accumulator registers: the 160-column tiles (40 registers
    // of bias) and the plain 256 x 256 tile spilled with it)
    // (GPP_F16X3 fetches bias AND scale, 32 registers: under the loop only on pre-split input maps -- no split temporaries -- and small tiles)
    constexpr bool BIASPRE = !PIPE && NF <= 4 && MF * NF <= (DT == GPP_F16X3 ? (XIN ? 16 : 0) : 24);
    if constexpr (BIASPRE) load_bias();

#ifdef GPP_VALU_PAD
    // experiment (make variant EXTRA=-DGPP_VALU_PAD=n): n extra vector-ALU instructions per wavefront ahead of the main loop -- is a layer
    // bound by instruction issue?  (profiles/r4/valu_pad_experiment.txt)
    {
        int pad = lane;
#pragma unroll
        for (int i = 0; i < GPP_VALU_PAD; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pad));
        if (pad == -12345) acc[0][0][0] = 1.0f;
    }
#endif
    GPP_STAMP(1);
    // ---- main loop.  Ring of STAGES buffers, PF = STAGES-1 K-steps of LDS-DMA in flight; one raw
    // s_barrier per K-step.  At the top of step ks a counted vmcnt retires this wave's loads of
    // stage ks only (later stages stay in flight across the barrier); after the barrier every
    // wave's loads of stage ks have landed and every wave has finished reading the buffer of step
    // ks-1, which is exactly the buffer the next prefetch (stage ks+PF) overwrites.
    // K order: 64-channel chunk OUTER, tap INNER (weights are packed to match): the nine taps of one
    // chunk re-read the same few input rows back to back, so they hit in the XCD's L2 instead of being
    // re-fetched across the fabric once per tap
    const int taps = d.KH * d.KW;
    int cc = 0, kw = 0, kh = 0, issued = 0, ibuf = 0;
    if (ks0 != 0) { cc = ks0 / taps; kw = (ks0 % taps) % d.KW; kh = (ks0 % taps) / d.KW; }
    set_tap(kh, kw);
    auto issue_next = [&]() {
        stage(ibuf, cc, ks0 + issued);
        if (++kw == d.KW) {
            kw = 0;
            if (++kh == d.KH) { kh = 0; ++cc; }
        }
        set_tap(kh, kw);
        ++issued;
        if (++ibuf == STAGES) ibuf = 0;
    };
    auto load_frags = [&](frag (&af)[MF], frag (&bfr)[NF], int buf, int kk) {
        const unsigned char* sbase = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < MF; ++i) af[i] = *(const frag*)(sbase + a_rd[kk] + i * 16 * kRowBytes);
#pragma unroll
        for (int j = 0; j < NF; ++j) bfr[j] = *(const frag*)(sbase + b_rd[kk] + j * 16 * kRowBytes);
    };
    auto mfma_all = [&](const frag (&af)[MF], const frag (&bfr)[NF]) {
        if constexpr (DT == GPP_F32) {
            // K value outermost: consecutive v_mfma_f32_16x16x4_f32 go to different accumulators (40-cycle dependent
            // latency against a 32-cycle issue interval)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j][e], af[i][e], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = E::mfma(bfr[j], af[i], acc[i][j]);
        }
    };

    if constexpr (PIPE) {
        // Software-pipelined, explicitly interleaved form (two LDS buffers).  Per K-step k:
        //   phase 0:  MFMA(kk=0 of k)  ||  LDS reads of kk=1 of k
        //   wait stage k+1 landed, lgkmcnt(0), s_barrier      (everyone is done reading buffer k&1)
        //   phase 1:  MFMA(kk=1 of k)  ||  LDS-DMA of stage k+2 into buffer k&1  ||  LDS reads of kk=0 of k+1
        // Each phase is cut into MF groups {PER_STAGE/MF LDS-DMA, 1-2 ds_read_b128, NF MFMA} pinned with
        // sched_barrier, so the ~100-cycle issue cost of every LDS-DMA and the LDS read latency sit
        // under matrix-pipe work instead of in front of it.  In the tail the DMA goes through a
        // zero-length descriptor (dropped by the range check) and the look-ahead reads hit a buffer
        // nobody uses: no branches inside the interleaved region.
        static_assert(STAGES == 2, "pipelined loop: two buffers");
        // kSpreadDma: every read of stage k lies between the barriers of steps k-1 and k, so the buffer of stage k+2 (= that of stage k) may
        // be written anywhere between the barriers of steps k and k+1.  Issuing all of a stage's LDS-DMA in the phase right after the
        // barrier made that phase as long as the CU's tile-fill rate allows (~38 cycles per 1 KB piece: profiles/r2/fill_rate_microbench.txt)
        // while the other phases ran at matrix speed with the vector-memory path idle (in-kernel phase stamps, profiles/r3/x3_phase_stamps.txt).
        // The weight rows (L2-hot: every workgroup reads the same ones) therefore go out one phase LATER -- phase A / phase 0 of the next step,
        // with that step's saved descriptor and offset -- and still land before the barrier that publishes them; the activation rows (which
        // may come from HBM) keep the early slot.  Same bytes into the same LDS addresses before the same barrier: results unchanged.
        // Only the three-phase x3 loop does this: there the late pieces have phase B to land.  In the two-phase 16-bit loop the barrier
        // follows phase 0 directly and the late pieces were waited for (measured: bf16 regression tower 0.57 -> 0.52 of its peak).
        // (and not the 4 x 1-wavefront 128 x 160 tile, MF = 2: its two groups per phase leave the late pieces no room; 327 -> 338 us with them)
        constexpr bool SPREAD = kSpreadDma && kX3<DT> && MF >= 4;
        constexpr int C_PIECES = SPREAD ? A_IT : PER_STAGE;
        // the zero-length descriptor of the tail is built from a SCALAR select of its record count: selecting between two
        // whole descriptors made the compiler carry them in VGPRs and wrap every LDS-DMA of the loop in a waterfall loop
        // (v_readfirstlane / v_cmp / s_and_saveexec / s_cbranch_execnz, four per K-step)
        auto tail_rsrc = [](const void* base, int bytes, bool live) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, live ? bytes : 0, 0x00020000);
        };
        auto issue_one = [&](int idx, int buf, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, int so_a, int so_w) {
            unsigned char* sa = smem + buf * STAGE + wave * A_IT * 8 * kRowBytes;
            unsigned char* sb = smem + buf * STAGE + A_BYTES + wave * B_IT * 8 * kRowBytes;
            if (idx < A_IT) glds16(ra, voff_of(idx < A_IT ? idx : 0), so_a, sa + idx * 8 * kRowBytes);
            else glds16(rw, w_voff[idx >= A_IT ? idx - A_IT : 0], so_w, sb + (idx - A_IT) * 8 * kRowBytes);
        };
        auto advance_tap = [&]() {
            if (++kw == d.KW) {
                kw = 0;
                if (++kh == d.KH) { kh = 0; ++cc; }
            }
            set_tap(kh, kw);
            ++issued;
        };
        if constexpr (kX3<DT>) {
            // ---- pre-split bf16x3 (XIN): a K-step is 32 channels = one k-slice, three matrix products per accumulator, three phases:
            //   phase A:  MFMA(hi * wlo)  ||  LDS reads of whi (this stage)
            //   phase B:  MFMA(hi * whi)  ||  LDS reads of lo  (this stage)
            //   wait stage k+1 landed, lgkmcnt(0), s_barrier                 (everyone is done reading buffer k&1)
            //   phase C:  MFMA(lo * whi)  ||  LDS-DMA of stage k+2 into buffer k&1  ||  LDS reads of hi, wlo of stage k+1
            // hi is live in A-B, lo in C, whi in B-C, wlo in A: every fragment set is reloaded while the phase that runs does not use
            // it, so the four sets take 96 registers (as in the 16-bit loop) and the body needs no second copy.  One barrier per
            // 3 MF NF MFMAs; the DMA of stage k+2 has two phases to land.
            static_assert(XIN, "the pipelined bf16x3 loop reads pre-split activation rows");
            // which weight fragments group g of a phase fetches: spread over all MF groups, or (GPP_X3_BEARLY) one per group from the first
            // group on, so that the last of them has more than a group's time to arrive before the next phase's first MFMA needs all NF
#if GPP_X3_BEARLY
            constexpr int BPG = (NF + MF - 1) / MF;
            auto BJ0 = [](int g) { return g * BPG < NF ? g * BPG : NF; };
#else
            auto BJ0 = [](int g) { return g * NF / MF; };
#endif
            xh8 ah[MF], al[MF], bh[NF], bl[NF];
#pragma unroll
            for (int idx = 0; idx < PER_STAGE; ++idx) issue_one(idx, 0, in_rsrc, w_rsrc, cc * kRowBytes, ks0 * kRowBytes);
            advance_tap();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // (kSpreadDma) the weight rows of a stage go out one phase after its activation rows, those of stage 1 in phase A of step 0
            __amdgpu_buffer_rsrc_t rw_late;
            int so_w_late;
            {
                const bool live = issued < nk;
                const __amdgpu_buffer_rsrc_t ra = tail_rsrc(d.in, d.in_bytes, live), rw = tail_rsrc(d.weight, d.weight_bytes, live);
#pragma unroll
                for (int idx = 0; idx < C_PIECES; ++idx) issue_one(idx, 1, ra, rw, cc * kRowBytes, (ks0 + issued) * kRowBytes);
                rw_late = rw;
                so_w_late = __builtin_amdgcn_readfirstlane((ks0 + issued) * kRowBytes);
                if (live) advance_tap();
            }
#pragma unroll
            for (int i = 0; i < MF; ++i) ah[i] = *(const xh8*)(smem + a_rd[0] + i * 16 * kRowBytes);
#pragma unroll
            for (int j = 0; j < NF; ++j) bl[j] = *(const xh8*)(smem + b_rd[1] + j * 16 * kRowBytes);
            if (GPP_X3_PRIO && NW == 8 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#ifdef GPP_STAMPS
            // diagnostic build, reserved bit 5: wavefront (reserved >> 8) & 7 of the first 64 workgroups sums, in scalar registers, the
            // real-time counter differences of the three phases and of the wait + barrier over its K-steps (the counter is read at
            // the phase boundaries and consumed where lgkmcnt is 0 anyway); five words per workgroup at the end
            // ablations of the same build (timing only, the results are wrong): bit 6 no LDS-DMA in the loop, bit 7 no LDS reads in phase C,
            // bit 11 none in phases A / B, bit 12 no static priority
            const bool abl_dma = d.reserved & 64, abl_rdc = d.reserved & 128, abl_rdab = d.reserved & 2048;
            if (NW == 8 && (d.reserved & 4096)) __builtin_amdgcn_s_setprio(0);
            const bool stamping = (d.reserved & 32) && blockIdx.x < 64 && wave == ((d.reserved >> 8) & 7);
            unsigned long long t_top = 0, t_a = 0, t_b = 0, t_bar = 0, t_prev = 0, sum_a = 0, sum_b = 0, sum_w = 0, sum_c = 0;
#endif
            for (int ks = 0; ks < nk; ++ks) {
                const int cur = ks & 1;
                const unsigned char* scur = smem + cur * STAGE;
                const unsigned char* snxt = smem + (cur ^ 1) * STAGE;
                const bool live = issued < nk;
                const __amdgpu_buffer_rsrc_t ra = tail_rsrc(d.in, d.in_bytes, live), rw = tail_rsrc(d.weight, d.weight_bytes, live);
                const int so_a = __builtin_amdgcn_readfirstlane(cc * kRowBytes), so_w = __builtin_amdgcn_readfirstlane((ks0 + issued) * kRowBytes);
#ifdef GPP_STAMPS
                if (stamping) t_top = __builtin_amdgcn_s_memrealtime();
#endif
                __builtin_amdgcn_sched_barrier(0);
                // ---- phase A: hi * wlo, fetch whi
#pragma unroll
                for (int g = 0; g < MF; ++g) {
                    if constexpr (SPREAD) {
#pragma unroll
                        for (int idx = A_IT + g * (B_IT - GPP_X3_WB) / MF; idx < A_IT + (g + 1) * (B_IT - GPP_X3_WB) / MF; ++idx) GPP_ABL(abl_dma) issue_one(idx, cur ^ 1, rw_late, rw_late, 0, so_w_late);
                    }
#pragma unroll
                    for (int j = BJ0(g); j < BJ0(g + 1); ++j) GPP_ABL(abl_rdab) bh[j] = *(const xh8*)(scur + b_rd[0] + j * 16 * kRowBytes);
#pragma unroll
                    for (int j = 0; j < NF; ++j) { const int js = GPP_SERP(g, j, NF); acc[g][js] = X3Half<DT>::mfma(bl[js], ah[g], acc[g][js]); }
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifdef GPP_STAMPS
                if (stamping) t_a = __builtin_amdgcn_s_memrealtime();
                __builtin_amdgcn_sched_barrier(0);
#endif
                // ---- phase B: hi * whi, fetch lo
#pragma unroll
                for (int g = 0; g < MF; ++g) {
                    if constexpr (SPREAD && GPP_X3_WB > 0) {
                        if (g < GPP_X3_WB) GPP_ABL(abl_dma) issue_one(A_IT + B_IT - GPP_X3_WB + g, cur ^ 1, rw_late, rw_late, 0, so_w_late);
                    }
                    GPP_ABL(abl_rdab) al[g] = *(const xh8*)(scur + a_rd[1] + g * 16 * kRowBytes);
#pragma unroll
                    for (int j = 0; j < NF; ++j) { const int js = GPP_SERP(g, j, NF); acc[g][js] = X3Half<DT>::mfma(bh[js], ah[g], acc[g][js]); }
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifdef GPP_STAMPS
                if (stamping) t_b = __builtin_amdgcn_s_memrealtime();
                __builtin_amdgcn_sched_barrier(0);
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef GPP_STAMPS
                if (stamping) {                 // every counter read so far has returned (lgkmcnt 0): the previous step's wait and phase C, this step's A and B
                    if (ks > 0) { sum_w += t_bar - t_prev; sum_c += t_top - t_bar; }
                    sum_a += t_a - t_top;
                    sum_b += t_b - t_a;
                    t_prev = t_b;
                }
                __builtin_amdgcn_sched_barrier(0);
#endif
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#ifdef GPP_STAMPS
                if (stamping) t_bar = __builtin_amdgcn_s_memrealtime();
#endif
                __builtin_amdgcn_sched_barrier(0);
                // ---- phase C: lo * whi, stage k+2 goes out, fetch hi and wlo of stage k+1
#pragma unroll
                for (int g = 0; g < MF; ++g) {
#pragma unroll
                    for (int idx = g * C_PIECES / MF; idx < (g + 1) * C_PIECES / MF; ++idx) GPP_ABL(abl_dma) issue_one(idx, cur, ra, rw, so_a, so_w);
                    GPP_ABL(abl_rdc) ah[g] = *(const xh8*)(snxt + a_rd[0] + g * 16 * kRowBytes);
#pragma unroll
                    for (int j = BJ0(g); j < BJ0(g + 1); ++j) GPP_ABL(abl_rdc) bl[j] = *(const xh8*)(snxt + b_rd[1] + j * 16 * kRowBytes);
#if GPP_X3_TAP_EARLY
                    if (g == MF - 1) {              // the next tap's offsets, computed beside the last group's MFMAs (branch-free: past the last stage the
                        rw_late = rw;               // descriptors are zero-length, whatever the offsets)
                        so_w_late = so_w;
                        ++kw;
                        const bool ww = kw == d.KW;
                        kw = ww ? 0 : kw;
                        kh += ww ? 1 : 0;
                        const bool wh = kh == d.KH;
                        kh = wh ? 0 : kh;
                        cc += wh ? 1 : 0;
                        set_tap(kh, kw);
                        ++issued;
                    }
#endif
#pragma unroll
                    for (int j = 0; j < NF; ++j) { const int js = GPP_SERP(g, j, NF); acc[g][js] = X3Half<DT>::mfma(bh[js], al[g], acc[g][js]); }
                    __builtin_amdgcn_sched_barrier(0);
                }
#if !GPP_X3_TAP_EARLY
                rw_late = rw;
                so_w_late = so_w;
                if (live) advance_tap();
#endif
            }
            if (NW == 8) __builtin_amdgcn_s_setprio(0);
#ifdef GPP_STAMPS
            if (stamping && lane == 0) {
                unsigned long long* o = (unsigned long long*)d.zero_page + (1 << 19) + blockIdx.x * 8;
                o[0] = sum_a; o[1] = sum_b; o[2] = sum_w; o[3] = sum_c; o[4] = (unsigned long long)nk;
            }
#endif
        } else {
        frag a0[MF], b0[NF], a1[MF], b1[NF];
        // prologue: stage 0 -> buffer 0, wait, stage 1 -> buffer 1, fragments kk=0 of step 0
#pragma unroll
        for (int idx = 0; idx < PER_STAGE; ++idx) issue_one(idx, 0, in_rsrc, w_rsrc, cc * kRowBytes, ks0 * kRowBytes);
        advance_tap();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __amdgpu_buffer_rsrc_t rw_late;
        int so_w_late;
        {
            const bool live = issued < nk;
            const __amdgpu_buffer_rsrc_t ra = tail_rsrc(d.in, d.in_bytes, live), rw = tail_rsrc(d.weight, d.weight_bytes, live);
#pragma unroll
            for (int idx = 0; idx < C_PIECES; ++idx) issue_one(idx, 1, ra, rw, cc * kRowBytes, (ks0 + issued) * kRowBytes);
            rw_late = rw;
            so_w_late = __builtin_amdgcn_readfirstlane((ks0 + issued) * kRowBytes);
            if (live) advance_tap();
        }
        load_frags(a0, b0, 0, 0);
        // static priority for the later-dispatched half of an 8-wavefront workgroup: it loses every issue arbitration to
        // its SIMD partner otherwise (measured +0.5 ... 2 % on the 256 x 256 tile; no effect on results)
        if (NW == 8 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
        for (int ks = 0; ks < nk; ++ks) {
#ifdef GPP_STAMPS
            if ((d.reserved & 32) && blockIdx.x < 64 && ks < 120 && wave == 0 && lane == 0)     // per-K-step timeline of a few workgroups
                ((unsigned long long*)d.zero_page)[(1 << 19) + blockIdx.x * 128 + ks] = __builtin_amdgcn_s_memrealtime();
#endif
            const int cur = ks & 1;
            const unsigned char* scur = smem + cur * STAGE;
            const unsigned char* snxt = smem + (cur ^ 1) * STAGE;
            // everything phase 1's LDS-DMA needs that is known already (descriptors, scalar offsets): computed HERE, in the shadow
            // of phase 0's MFMAs, not behind the barrier where all eight wavefronts would do scalar arithmetic while the matrix
            // pipe waits.  Wave-uniform by construction and said so explicitly (v_readfirstlane): left to its own analysis the
            // compiler kept the weight offset in a VGPR and wrapped four of the eight LDS-DMA of a K-step in waterfall loops.
            const bool live = issued < nk;
            const __amdgpu_buffer_rsrc_t ra = tail_rsrc(d.in, d.in_bytes, live), rw = tail_rsrc(d.weight, d.weight_bytes, live);
            const int so_a = __builtin_amdgcn_readfirstlane(cc * kRowBytes), so_w = __builtin_amdgcn_readfirstlane((ks0 + issued) * kRowBytes);
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 0
#pragma unroll
            for (int g = 0; g < MF; ++g) {
                if constexpr (SPREAD) {
#pragma unroll
                    for (int idx = A_IT + g * B_IT / MF; idx < A_IT + (g + 1) * B_IT / MF; ++idx) issue_one(idx, cur ^ 1, rw_late, rw_late, 0, so_w_late);
                }
                a1[g] = *(const frag*)(scur + a_rd[1] + g * 16 * kRowBytes);
#pragma unroll
                for (int j = g * NF / MF; j < (g + 1) * NF / MF; ++j) b1[j] = *(const frag*)(scur + b_rd[1] + j * 16 * kRowBytes);
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[g][j] = E::mfma(b0[j], a0[g], acc[g][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 1
#pragma unroll
            for (int g = 0; g < MF; ++g) {
#pragma unroll
                for (int idx = g * C_PIECES / MF; idx < (g + 1) * C_PIECES / MF; ++idx) issue_one(idx, cur, ra, rw, so_a, so_w);
                a0[g] = *(const frag*)(snxt + a_rd[0] + g * 16 * kRowBytes);
#pragma unroll
                for (int j = g * NF / MF; j < (g + 1) * NF / MF; ++j) b0[j] = *(const frag*)(snxt + b_rd[0] + j * 16 * kRowBytes);
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[g][j] = E::mfma(b1[j], a1[g], acc[g][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            rw_late = rw;
            so_w_late = so_w;
            if (live) advance_tap();         // a_voff for the next issue changes only after this step's DMA is out
        }
        if (NW == 8) __builtin_amdgcn_s_setprio(0);
        }
    } else {
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (issued < nk) issue_next();
    int cbuf = 0;
    // (diagnostic build: four stamps per K-step of the first 64 workgroups -- top, after the wait, after the barrier,
    // after the LDS-DMA issue; the MFMA part runs up to the next top)
#ifdef GPP_STAMPS
#define GPP_KSTAMP(j)                                                                                                  \
    do {                                                                                                               \
        if ((d.reserved & 32) && blockIdx.x < 64 && ks < 30 && wave == 0 && lane == 0)                                 \
            ((unsigned long long*)d.zero_page)[(1 << 19) + blockIdx.x * 128 + ks * 4 + (j)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define GPP_KSTAMP(j) do { } while (0)
#endif
    for (int ks = 0; ks < nk; ++ks) {
        GPP_KSTAMP(0);
        if (issued - ks - 1 >= PF - 1 && PF > 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * PER_STAGE) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GPP_KSTAMP(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        GPP_KSTAMP(2);
#ifdef GPP_STAMPS
        if (issued < nk) {
            if (d.reserved & 1) { ++issued; } else issue_next();      // bit 0 (diagnostic build only): skip the LDS-DMA
        }
        GPP_KSTAMP(3);
        if (!(d.reserved & 2))                                         // bit 1 (diagnostic build only): skip LDS reads + MFMA
#else
        if (issued < nk) issue_next();
#endif
        {
            if constexpr (kX3<DT>) {
                // one 32-channel K-step = one k-slice of v_mfma_f32_16x16x32_bf16, three matrix products per accumulator
                const unsigned char* sbase = smem + cbuf * STAGE;
                xh8 ah[MF], al[MF], bh[NF], bl[NF];
                if constexpr (XIN) {
                    // pre-split activations: the row holds [32 bf16 hi | 32 bf16 lo], exactly as the weight rows do
#pragma unroll
                    for (int i = 0; i < MF; ++i) {
                        ah[i] = *(const xh8*)(sbase + a_rd[0] + i * 16 * kRowBytes);
                        al[i] = *(const xh8*)(sbase + a_rd[1] + i * 16 * kRowBytes);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < MF; ++i)
                        Elem<DT>::split(*(const f32x4*)(sbase + a_rdx[0] + i * 16 * kRowBytes),
                                                *(const f32x4*)(sbase + a_rdx[1] + i * 16 * kRowBytes), ah[i], al[i]);
                }
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    bh[j] = *(const xh8*)(sbase + b_rd[0] + j * 16 * kRowBytes);
                    bl[j] = *(const xh8*)(sbase + b_rd[1] + j * 16 * kRowBytes);
                }
                // per accumulator and K-step: hi * wlo, hi * whi, lo * whi -- the order of the software-pipelined form below (its
                // three phases), so that every bf16x3 tile sums an output element in the same order
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        const int js = GPP_SERP2(i, j, NF);
                        acc[i][js] = X3Half<DT>::mfma(bl[js], ah[i], acc[i][js]);
                        acc[i][js] = X3Half<DT>::mfma(bh[js], ah[i], acc[i][js]);
                    }
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) { const int js = GPP_SERP2(i, j, NF); acc[i][js] = X3Half<DT>::mfma(bh[js], al[i], acc[i][js]); }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    frag af[MF], bfr[NF];
                    load_frags(af, bfr, cbuf, kk);
                    mfma_all(af, bfr);
                }
            }
        }
        if (++cbuf == STAGES) cbuf = 0;
    }
    }
    GPP_STAMP(2);
    // ---- epilogue, straight from registers.  The MFMA was issued as D = W_tile * X_tile^T, so
    // lane (fq = lane>>4, c = lane&15) of accumulator (i, j) holds output pixel i*16 + c and the four
    // weight-tile rows fq*4 + 0..3 of N-tile j.  The packed weight rows are interleaved on the host
    // (row 16h + 4q + r of every 32-row group = output channel 8q + 4h + r), hence tiles 2jj and
    // 2jj+1 together give this lane EIGHT CONSECUTIVE output channels n = n0 + 32jj + 8fq + 0..7:
    // one 16-byte store (two for float32 output), no LDS round trip.
    if (nsplit > 1) {
        // raw float32 partial tile -> d.partial[split][mt*BM + row][nt*BN + col]
        const int64_t rows_pad = (int64_t)d.partial_rows, npad = (int64_t)n_tiles * BN;
        float* part = (float*)d.partial + ((int64_t)split * rows_pad + (int64_t)mt * BM) * npad + nt * BN;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int lr = wm * (BM / WM) + i * 16 + frow;
#pragma unroll
            for (int jj = 0; jj < NF / 2; ++jj) {
                float* dst = part + (int64_t)lr * npad + wn * COLS + jj * 32 + fq * 8;
                *(f32x4*)dst = acc[i][2 * jj];
                *(f32x4*)(dst + 4) = acc[i][2 * jj + 1];
            }
        }
        return;
    }
    const scalar* res = (const scalar*)d.residual;
    // An INTERIOR tile -- every row a pixel, every column a channel: all but the last tile of a group / of the channel range --
    // runs the epilogue without per-lane conditions (the general form costs ~280 exec-mask branches per wavefront: 6 us of a
    // 256 x 256 tile's life, a quarter of a 18-K-step tile's) and walks the output rows instead of dividing per row.
    const bool interior = (m0 + BM <= Mg) && (n0 + BN <= d.C_out) && ((d.C_out & 7) == 0);
    if constexpr (!BIASPRE) load_bias();
    if constexpr (RESPRE) {
        if (use_pre && interior) {
#pragma unroll
            for (int i = 0; i < MF; ++i) {
#pragma unroll
                for (int jj = 0; jj < NF / 2; ++jj) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = comb(acc[i][2 * jj][e], jj, e);
                        v[4 + e] = comb(acc[i][2 * jj + 1][e], jj, 4 + e);
                    }
                    finish8_pre<DT>(d, v, n0 + wn * COLS + jj * 32 + fq * 8, ra_pre[i].obase, true, rpre[i][jj]);
                }
            }
            GPP_STAMP_END();
            return;
        }
        if (use_pre) {
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int m = m0 + wm * (BM / WM) + i * 16 + frow;
                if (m >= Mg) continue;
#pragma unroll
                for (int jj = 0; jj < NF / 2; ++jj) {
                    const int n = n0 + wn * COLS + jj * 32 + fq * 8;
                    if (n >= d.C_out) continue;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = comb(acc[i][2 * jj][e], jj, e);
                        v[4 + e] = comb(acc[i][2 * jj + 1][e], jj, 4 + e);
                    }
                    finish8_pre<DT>(d, v, n, ra_pre[i].obase, true, rpre[i][jj]);
                }
            }
            GPP_STAMP_END();
            return;
        }
    }
    if (interior && !res) {
        // output row of tile row i: pixel (b, p); the next tile row is 16 pixels further (walk, do not divide)
        const int mfirst = m0 + wm * (BM / WM) + frow;
        int b = mfirst / HoWo, p = mfirst - b * HoWo;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int64_t obase = out_off + (int64_t)b * out_bs + (int64_t)p * d.out_pitch;
#pragma unroll
            for (int jj = 0; jj < NF / 2; ++jj) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = comb(acc[i][2 * jj][e], jj, e);
                    v[4 + e] = comb(acc[i][2 * jj + 1][e], jj, 4 + e);
                }
                finish8_pre<DT>(d, v, n0 + wn * COLS + jj * 32 + fq * 8, obase, false, typename E::vec8());
            }
            p += 16;
            while (p >= HoWo) { p -= HoWo; ++b; }
        }
        GPP_STAMP_END();
        return;
    }
    PixWalk pwe;
    pwe.init(m0 + wm * (BM / WM) + frow, HoWo, W_out);
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + frow;
        if (m >= Mg) break;                                            // rows ascend with i
        const RowAddr ra = row_addr_at(d, pwe, W_out, H_out, H_res, W_res, out_off, out_bs, res_off, res_bs);
        pwe.advance(16, H_out, W_out);
        const scalar* rrow = res ? res + ra.rbase : nullptr;
#pragma unroll
        for (int jj = 0; jj < NF / 2; ++jj) {
            const int n = n0 + wn * COLS + jj * 32 + fq * 8;
            if (n >= d.C_out) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = comb(acc[i][2 * jj][e], jj, e);
                v[4 + e] = comb(acc[i][2 * jj + 1][e], jj, 4 + e);
            }
            finish8<DT>(d, v, n, ra.obase, rrow);
        }
    }
    GPP_STAMP_END();
}

template <int DT, int BM, int BN, int WM, int WN, int STAGES, bool PIPE, bool XIN = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_igemm_kernel(const gpp_conv_desc d)
{
    conv_igemm_body<DT, BM, BN, WM, WN, STAGES, PIPE, XIN>(d, blockIdx.x, gridDim.x);
}

// A layer whose C_out is an odd multiple of 128 (the fused tower inputs: 896 = 3 x 256 + 128) in ONE grid of two tile
// shapes: workgroups [0, n0) cover the first C_out - 128 columns with 256 x 256 tiles (descriptor d0), workgroups
// [split0, split0 + n1) the last 128 columns with the same kernel turned on its side, 512 x 128 tiles (descriptor d1 = the
// same layer with its weight / bias / output pointers moved to that column block).  Every workgroup does the same amount
// of matrix work, none of it on padding, and the hardware hands the second range out behind the first, into the CUs its
// last partial round leaves idle: 1253 workgroups = 5 rounds instead of 1432 = 6 for the 896-column layer.
template <int DT, bool XIN = false>
__global__ __launch_bounds__(512, 2) void conv_igemm_dual_kernel(const gpp_conv_desc d0, const gpp_conv_desc d1, const int n0,
                                                                 const int split0, const int n1)
{
    if ((int)blockIdx.x < split0) {
        if ((int)blockIdx.x < n0) conv_igemm_body<DT, 256, 256, 2, 4, 2, true, XIN>(d0, blockIdx.x, n0);
    } else {
        conv_igemm_body<DT, 512, 128, 4, 2, 2, true, XIN>(d1, (int)blockIdx.x - split0, n1);
    }
}

// A grid of TWO tile heights against the round quantisation of one-workgroup-per-CU tiles.  722 tiles of 256 x 256 (the regression
// tower at B = 8) are 2.82 rounds over 256 CUs and cost 3: the last round's 210 workgroups take as long as a full one.  Here
// workgroups [0, na) run BMA-row tiles over the first rounds_a * 256 / n_tiles M tiles of group 0 (descriptor da: whole rounds, every
// tile interior), workgroups [splita, splita + nb) run BMB-row tiles over everything that is left (descriptor db: group 0 from row
// row_begin on, then the other groups), with BMB the tallest tile whose grid still fits the rounds it needs: 512 + 236 workgroups of
// 256 / 224 rows = 2 + 0.875 rounds' worth of time instead of 3.  Same K order per output element in every tile: not a bit changes.
template <int DT, int BMA, int BMB, int BN, bool XIN>
__global__ __launch_bounds__(512, 2) void conv_igemm_mix_kernel(const gpp_conv_desc da, const gpp_conv_desc db, const int na,
                                                                const int splita, const int nb)
{
    if ((int)blockIdx.x < splita) {
        if ((int)blockIdx.x < na) conv_igemm_body<DT, BMA, BN, 2, 4, 2, true, XIN>(da, blockIdx.x, na);
    } else {
        conv_igemm_body<DT, BMB, BN, 2, 4, 2, true, XIN>(db, (int)blockIdx.x - splita, nb);
    }
}

// Second pass of a split-K launch: sum the partial slabs in split order (deterministic), then the
// same epilogue as the fused path.  One thread per (output row, 8 output channels).
template <int DT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const gpp_conv_desc d, int BM, int npad, int nsplit)
{
    using scalar = typename Elem<DT>::scalar;
    const int n8 = (d.C_out + 7) / 8;
    const int64_t total = (int64_t)d.partial_rows * n8;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int r = (int)(e / n8), n = (int)(e - (int64_t)r * n8) * 8;
    const int mt = r / BM;
    int tile_start = 0, H_out = 0, W_out = 0, H_res = 0, W_res = 0;
    int64_t out_off = 0, out_bs = 0, res_off = 0, res_bs = 0;
#pragma unroll
    for (int q = 0; q < GPP_MAX_GROUPS; ++q) {
        if (q < d.n_groups && mt >= d.groups[q].tile_start) {
            tile_start = d.groups[q].tile_start;
            H_out = d.groups[q].H_out; W_out = d.groups[q].W_out;
            H_res = d.groups[q].H_res; W_res = d.groups[q].W_res;
            out_off = d.groups[q].out_off; out_bs = d.groups[q].out_bstride;
            res_off = d.groups[q].res_off; res_bs = d.groups[q].res_bstride;
        }
    }
    const int HoWo = H_out * W_out;
    const int m = r - tile_start * BM;
    if (m >= d.batch * HoWo) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.0f;
    const float* src = (const float*)d.partial + (int64_t)r * npad + n;
    for (int s = 0; s < nsplit; ++s) {
        const f32x4 a = *(const f32x4*)(src + (int64_t)s * d.partial_rows * npad);
        const f32x4 b = *(const f32x4*)(src + (int64_t)s * d.partial_rows * npad + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] += a[k]; v[4 + k] += b[k]; }
    }
    if constexpr (DT == GPP_F16X3) {
        if (d.out_scale) {
#pragma unroll
            for (int k = 0; k < 8; ++k) if (n + k < d.C_out) v[k] *= d.out_scale[n + k];
        }
    }
    if (d.bias) {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (n + k < d.C_out) v[k] += d.bias[n + k];
    }
    const RowAddr ra = row_addr(d, m, HoWo, W_out, H_out, H_res, W_res, out_off, out_bs, res_off, res_bs);
    finish8<DT>(d, v, n, ra.obase, d.residual ? (const scalar*)d.residual + ra.rbase : nullptr);
}

// ---------------------------------------------------------------------------------------------
// Fused tail of a ResNet bottleneck: y = relu(W2 * relu(W1 (*) a + b1) + b2 + shortcut), i.e. the
// 3x3 conv "branch2b" (CMID -> CMID, stride 1, pad 1) and the 1x1 conv "branch2c" (CMID -> 4*CMID,
// + residual + ReLU) in one launch.  The BM x CMID tile of the intermediate never leaves the CU: it
// goes accumulators -> (bias, ReLU, round to 16 bit exactly as the unfused layer would store it) ->
// LDS in the A-operand layout, and is multiplied there by W2 in 128-wide output tiles.  Saves one
// write + one read of the intermediate map (2 x 34.5 MB per res2 block at B = 8) on layers that are
// HBM-bound; results are bit-identical to the two separate launches (same K order, same rounding).
// d1 = descriptor of the 3x3 layer (its `out` is not written), d2 = descriptor of the 1x1 layer.
//
// (Round 2 also carried a form that computed the FOLLOWING block's first 1x1 layer in the same launch: bit-identical, measured
// slower than the separate launch -- C = 64: 137 us against 92 + 36 -- and removed in round 3.)
template <int DT, int BM, int CMID>
__global__ __launch_bounds__(256, 2) void bottleneck_tail_kernel(const gpp_conv_desc d1, const gpp_conv_desc d2)
{
    using E = Elem<DT>;
    using vec8 = typename E::vec8;
    using scalar = typename E::scalar;
    constexpr int WM = 2, WN = 2, NW = 4;
    constexpr int P2M = 2, P2N = 2;                                        // wavefront layout of phase 2
    constexpr int MF = BM / WM / 16, NF1 = CMID / WN / 16;
    constexpr int MF2 = BM / P2M / 16, NF2 = 128 / P2N / 16, COLS2 = 128 / P2N;   // phase 2: BM x 128 output tiles
    static_assert(BM % (16 * P2M) == 0, "phase-2 wave tile");
    constexpr int KC = CMID / 64;                                          // 64-channel chunks of the intermediate
    constexpr int A_BYTES = BM * kRowBytes, B_BYTES = CMID * kRowBytes, STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = BM / 8 / NW, B_IT = CMID / 8 / NW, PER_STAGE = A_IT + B_IT;
    constexpr int T_BYTES = KC * A_BYTES;                                  // intermediate tile, A-operand layout
    constexpr int W2_IT = 128 / 8 / NW;                                    // LDS-DMA per wave per chunk of a W2 tile
    static_assert(BM % 32 == 0 && (CMID == 64 || CMID == 128), "tile shape");
    static_assert(T_BYTES + KC * 128 * kRowBytes <= 2 * STAGE, "phase-2 buffers alias the phase-1 ring");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int mt = xcd_remap(blockIdx.x, gridDim.x);
    const gpp_conv_desc& d = d1;                                           // (GPP_STAMP reads d.reserved / d.zero_page)
    (void)d;
    GPP_STAMP(0);
    const gpp_conv_group& G1 = d1.groups[0];
    const gpp_conv_group& G2 = d2.groups[0];
    const int H = G1.H_out, W = G1.W_out, HW = H * W;
    const int Mg = d1.batch * HW;
    const int m0 = mt * BM;

    // ---- phase 1: 3x3 conv, the main loop of conv_igemm_kernel with BN = CMID
    const int srow = lane >> 3;
    const int gchunk = (lane & 7) ^ srow;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d1.in, 0, d1.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d1.weight, 0, d1.weight_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d2.weight, 0, d2.weight_bytes, 0x00020000);
    int a_base[A_IT], a_mask[A_IT], a_voff[A_IT];
    const int pitch2 = d1.in_pitch * 2;
    {
        PixWalk pw;
        pw.init(m0 + wave * A_IT * 8 + srow, HW, W);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + (wave * A_IT + i) * 8 + srow;
            a_mask[i] = 0;
            a_base[i] = 0;
            if (m < Mg) {
                const int iy0 = pw.oy - 1, ix0 = pw.ox - 1;
                int mask = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) mask |= (((unsigned)(iy0 + k) < (unsigned)H) << k) | (((unsigned)(ix0 + k) < (unsigned)W) << (8 + k));
                a_mask[i] = mask;
                a_base[i] = (int)((G1.in_off + (int64_t)pw.b * G1.in_bstride) * 2) + gchunk * 16 + (iy0 * W + ix0) * pitch2;
            }
            if (i + 1 < A_IT) pw.advance(8, H, W);
        }
    }
    constexpr int Ktot1 = 9 * CMID;
    int w_voff[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) w_voff[i] = ((wave * B_IT + i) * 8 + srow) * Ktot1 * 2 + gchunk * 16;
    auto set_tap = [&](int kh, int kw) {
        const int delta = (kh * W + kw) * pitch2;
        const int need = (1 << kh) | (1 << (8 + kw));
#pragma unroll
        for (int i = 0; i < A_IT; ++i) a_voff[i] = ((a_mask[i] & need) == need) ? a_base[i] + delta : kOutOfRange;
    };
    const int frow = lane & 15, fq = lane >> 4;
    const int wm2 = wave / P2N, wn2 = wave % P2N;
    int a_rd[2], b1_rd[2], a_rd2[2], b2_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fq) ^ (frow & 7)) << 4;
        a_rd[kk] = (wm * (BM / WM) + frow) * kRowBytes + sw;
        b1_rd[kk] = A_BYTES + (wn * (CMID / WN) + frow) * kRowBytes + sw;
        a_rd2[kk] = (wm2 * (BM / P2M) + frow) * kRowBytes + sw;
        b2_rd[kk] = T_BYTES + (wn2 * COLS2 + frow) * kRowBytes + sw;
    }
    f32x4 acc1[MF][NF1];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF1; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int nk = 9 * KC;
    int cc = 0, kw = 0, kh = 0, issued = 0, ibuf = 0;
    set_tap(0, 0);
    auto issue_next = [&]() {
        unsigned char* sa = smem + ibuf * STAGE + wave * A_IT * 8 * kRowBytes;
        unsigned char* sb = smem + ibuf * STAGE + A_BYTES + wave * B_IT * 8 * kRowBytes;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16(in_rsrc, a_voff[i], cc * kRowBytes, sa + i * 8 * kRowBytes);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16(w1_rsrc, w_voff[i], issued * kRowBytes, sb + i * 8 * kRowBytes);
        if (++kw == 3) {
            kw = 0;
            if (++kh == 3) { kh = 0; ++cc; }
        }
        set_tap(kh, kw);
        ++issued;
        ibuf ^= 1;
    };
    // The shortcut rows of output tile t (phase 2) are fetched into registers long before they are used: tile 0's right
    // here, in flight underneath the whole 3x3 phase; tile t+1's under tile t's epilogue stores.  When the epilogue
    // issued them itself (load -> wait -> add -> store, twice per workgroup) it was the longest phase of this HBM-bound
    // kernel: 20.7 of 29.8 us per workgroup at C = 64.
    RowAddr ra[MF2];
    {
        PixWalk pw;
        pw.init(m0 + wm2 * (BM / P2M) + frow, HW, W);
        const PixWalk first = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < MF2; ++i) {
            const int m = m0 + wm2 * (BM / P2M) + i * 16 + frow;
            ra[i] = row_addr_at(d2, m < Mg ? pw : first, W, H, G2.H_res, G2.W_res, G2.out_off, G2.out_bstride, G2.res_off, G2.res_bstride);
            if (i + 1 < MF2) pw.advance(16, H, W);
        }
    }
    const scalar* res = (const scalar*)d2.residual;
    vec8 rpre[MF2][NF2 / 2];
    auto prefetch_res = [&](int t) {
#pragma unroll
        for (int i = 0; i < MF2; ++i)
#pragma unroll
            for (int jj = 0; jj < NF2 / 2; ++jj) {
                const int n = t * 128 + wn2 * COLS2 + jj * 32 + fq * 8;
                rpre[i][jj] = *(const vec8*)(res + ra[i].rbase + n);       // rows past the end were clamped to row 0: a valid address
            }
    };
    if (res) prefetch_res(0);
    issue_next();
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issued < nk) issue_next();
        const unsigned char* sbase = smem + (ks & 1) * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            vec8 af[MF], bfr[NF1];
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const vec8*)(sbase + a_rd[kk] + i * 16 * kRowBytes);
#pragma unroll
            for (int j = 0; j < NF1; ++j) bfr[j] = *(const vec8*)(sbase + b1_rd[kk] + j * 16 * kRowBytes);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF1; ++j) acc1[i][j] = E::mfma(bfr[j], af[i], acc1[i][j]);
        }
    }

    GPP_STAMP(1);
    // ---- hand-over: everyone is done with the ring; W2 tile 0 starts streaming in while the
    // intermediate tile is written (bias, ReLU, rounded to the storage type) in A-operand layout
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int w2_voff[W2_IT];
#pragma unroll
    for (int i = 0; i < W2_IT; ++i) w2_voff[i] = ((wave * W2_IT + i) * 8 + srow) * CMID * 2 + gchunk * 16;
    auto stage_w2 = [&](int t) {
        // rows t*128 .. t*128+127 of the packed 1x1 weights, KC chunks of 128 bytes each
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int i = 0; i < W2_IT; ++i)
                glds16(w2_rsrc, w2_voff[i], t * 128 * CMID * 2 + kc * kRowBytes,
                       smem + T_BYTES + kc * 128 * kRowBytes + (wave * W2_IT + i) * 8 * kRowBytes);
    };
    stage_w2(0);
    {
        constexpr int COLS1 = CMID / WN;
#pragma unroll
        for (int jj = 0; jj < NF1 / 2; ++jj) {
            const int n = wn * COLS1 + jj * 32 + fq * 8;              // 8 consecutive intermediate channels
            float bias_v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bias_v[e] = d1.bias ? d1.bias[n + e] : 0.0f;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int r = wm * (BM / WM) + i * 16 + frow;
                vec8 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float lo = acc1[i][2 * jj][e] + bias_v[e], hi = acc1[i][2 * jj + 1][e] + bias_v[4 + e];
                    if (d1.relu) { lo = fmaxf(lo, 0.0f); hi = fmaxf(hi, 0.0f); }
                    ov[e] = (scalar)lo;
                    ov[4 + e] = (scalar)hi;
                }
                const int chunk = (n & 63) >> 3;
                *(vec8*)(smem + (n >> 6) * A_BYTES + r * kRowBytes + ((chunk ^ (r & 7)) << 4)) = ov;
            }
        }
    }

    GPP_STAMP(2);
    // ---- phase 2: y tile = T (BM x CMID) * W2^T, 128 output channels at a time
    const int n2_tiles = d2.C_out / 128;
    for (int t = 0; t < n2_tiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // W2 tile t (and, for t = 0, T) is in LDS
        asm volatile("" ::: "memory");
        f32x4 acc2[MF2][NF2];
#pragma unroll
        for (int i = 0; i < MF2; ++i)
#pragma unroll
            for (int j = 0; j < NF2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                vec8 af[MF2], bfr[NF2];
#pragma unroll
                for (int i = 0; i < MF2; ++i) af[i] = *(const vec8*)(smem + kc * A_BYTES + a_rd2[kk] + i * 16 * kRowBytes);
#pragma unroll
                for (int j = 0; j < NF2; ++j) bfr[j] = *(const vec8*)(smem + kc * 128 * kRowBytes + b2_rd[kk] + j * 16 * kRowBytes);
#pragma unroll
                for (int i = 0; i < MF2; ++i)
#pragma unroll
                    for (int j = 0; j < NF2; ++j) acc2[i][j] = E::mfma(bfr[j], af[i], acc2[i][j]);
            }
        if (t + 1 < n2_tiles) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // everyone has read W2 tile t
            asm volatile("" ::: "memory");
            stage_w2(t + 1);                                           // streams in under the epilogue below
        }
        // bias + shortcut + ReLU into the accumulators (this consumes rpre), then refill rpre for the next tile, then store
        float outv[MF2][NF2 / 2][8];
#pragma unroll
        for (int jj = 0; jj < NF2 / 2; ++jj) {
            const int n = t * 128 + wn2 * COLS2 + jj * 32 + fq * 8;
            float bias_v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bias_v[e] = d2.bias ? d2.bias[n + e] : 0.0f;
#pragma unroll
            for (int i = 0; i < MF2; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    outv[i][jj][e] = acc2[i][2 * jj][e] + bias_v[e];
                    outv[i][jj][4 + e] = acc2[i][2 * jj + 1][e] + bias_v[4 + e];
                }
                if (res) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) outv[i][jj][e] += (float)rpre[i][jj][e];
                }
            }
        }
        if (res && t + 1 < n2_tiles) prefetch_res(t + 1);
#pragma unroll
        for (int jj = 0; jj < NF2 / 2; ++jj) {
            const int n = t * 128 + wn2 * COLS2 + jj * 32 + fq * 8;
#pragma unroll
            for (int i = 0; i < MF2; ++i) {
                const int m = m0 + wm2 * (BM / P2M) + i * 16 + frow;
                if (m >= Mg) continue;
                finish8_pre<DT>(d2, outv[i][jj], n, ra[i].obase, false, rpre[i][jj]);
            }
        }
    }
    GPP_STAMP(3);
#ifdef GPP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    GPP_STAMP(4);
}

// Fill in what the kernel needs beyond the caller's fields (buffer extents, first tile of every group); returns the
// number of M tiles or a negative error.
template <int BM, int BN>
int prepare(gpp_conv_desc& d)
{
    const int esz = (d.dtype == GPP_F32 || d.dtype == GPP_BF16X3 || d.dtype == GPP_F16X3) ? 4 : 2;
    if (d.weight_rows < ((d.C_out + BN - 1) / BN) * BN) return GPP_ERR_BAD_ARG;
    int64_t in_elems = 0;
    for (int g = 0; g < d.n_groups; ++g) {
        const gpp_conv_group& G = d.groups[g];
        const int64_t end = G.in_off + (int64_t)(d.batch - 1) * G.in_bstride + ((int64_t)G.H_in * G.W_in - 1) * d.in_pitch + d.C_in;
        if (G.in_off < 0 || G.in_bstride < 0) return GPP_ERR_BAD_ARG;
        in_elems = end > in_elems ? end : in_elems;
    }
    const int64_t w_bytes = (int64_t)d.weight_rows * d.KH * d.KW * d.C_in * esz;
    if (in_elems * esz >= (1LL << 31) || w_bytes >= (1LL << 31)) return GPP_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    d.in_bytes = (int32_t)(in_elems * esz);
    d.weight_bytes = (int32_t)w_bytes;
    int tiles = 0;
    for (int g = 0; g < d.n_groups; ++g) {
        d.groups[g].tile_start = tiles;
        tiles += (d.batch * d.groups[g].H_out * d.groups[g].W_out - d.groups[g].row_begin + BM - 1) / BM;
    }
    d.partial_rows = tiles * BM;
    return tiles;
}

// ---------------------------------------------------------------------------------------------
// The fused bottleneck tail for the float32-storage x3 types (GPP_BF16X3 / GPP_F16X3) on PRE-SPLIT maps: the same two phases as
// bottleneck_tail_kernel with every matrix product in its three-product form (hi*wlo, hi*whi, lo*whi per 32-channel K-step, the order
// of every x3 tile) and the intermediate tile written to LDS as the [32 hi | 32 lo] rows the unfused 3x3 layer would have stored
// (acc * out_scale + bias, ReLU, clamp, split): bit-identical to the two separate launches.  Input map, shortcut map and output map
// are pre-split (x3_split: d1 GPP_X3_IN; d2 GPP_X3_OUT | GPP_X3_RES); the 69 MB (res2, B = 8) intermediate map is never written.
// LDS: max(two phase-1 stages, intermediate tile + one 128-row tile of W2) -- 64 KB for C = 64 at 128 rows: two workgroups per CU.
template <int DT, int BM, int CMID>
__global__ __launch_bounds__(256, (BM <= 64 ? 3 : 2)) void bottleneck_tail_x3_kernel(const gpp_conv_desc d1, const gpp_conv_desc d2)
{
    static_assert(kX3<DT>, "x3 types");
    using xh8 = typename X3Half<DT>::vec;
    constexpr bool OSCALE = (DT == GPP_F16X3);
    constexpr int WM = 2, WN = 2, NW = 4, P2M = 2, P2N = 2;
    constexpr int MF = BM / WM / 16, NF1 = CMID / WN / 16;
    constexpr int MF2 = BM / P2M / 16, NF2 = 128 / P2N / 16, COLS2 = 128 / P2N;
    constexpr int KC = CMID / 32;                                          // 32-channel K-steps of the intermediate
    constexpr int A_BYTES = BM * kRowBytes, B_BYTES = CMID * kRowBytes, STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = BM / 8 / NW, B_IT = CMID / 8 / NW;
    constexpr int T_BYTES = KC * A_BYTES;
    constexpr int W2_IT = 128 / 8 / NW;
    static_assert(BM % 32 == 0 && (CMID == 64 || CMID == 128) && NF1 % 2 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int mt = xcd_remap(blockIdx.x, gridDim.x);
    const gpp_conv_group& G1 = d1.groups[0];
    const gpp_conv_group& G2 = d2.groups[0];
    const int H = G1.H_out, W = G1.W_out, HW = H * W;
    const int Mg = d1.batch * HW;
    const int m0 = mt * BM;

    // ---- phase 1: 3x3 conv over the pre-split input map (float32-sized elements: 4 bytes)
    const int srow = lane >> 3;
    const int gchunk = (lane & 7) ^ srow;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d1.in, 0, d1.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d1.weight, 0, d1.weight_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d2.weight, 0, d2.weight_bytes, 0x00020000);
    int a_base[A_IT], a_mask[A_IT], a_voff[A_IT];
    const int pitch4 = d1.in_pitch * 4;
    {
        PixWalk pw;
        pw.init(m0 + wave * A_IT * 8 + srow, HW, W);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + (wave * A_IT + i) * 8 + srow;
            a_mask[i] = 0;
            a_base[i] = 0;
            if (m < Mg) {
                const int iy0 = pw.oy - 1, ix0 = pw.ox - 1;
                int mask = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) mask |= (((unsigned)(iy0 + k) < (unsigned)H) << k) | (((unsigned)(ix0 + k) < (unsigned)W) << (8 + k));
                a_mask[i] = mask;
                a_base[i] = (int)((G1.in_off + (int64_t)pw.b * G1.in_bstride) * 4) + gchunk * 16 + (iy0 * W + ix0) * pitch4;
            }
            if (i + 1 < A_IT) pw.advance(8, H, W);
        }
    }
    constexpr int Ktot1 = 9 * CMID;
    int w_voff[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) w_voff[i] = ((wave * B_IT + i) * 8 + srow) * Ktot1 * 4 + gchunk * 16;
    auto set_tap = [&](int kh, int kw) {
        const int delta = (kh * W + kw) * pitch4;
        const int need = (1 << kh) | (1 << (8 + kw));
#pragma unroll
        for (int i = 0; i < A_IT; ++i) a_voff[i] = ((a_mask[i] & need) == need) ? a_base[i] + delta : kOutOfRange;
    };
    const int frow = lane & 15, fq = lane >> 4;
    const int wm2 = wave / P2N, wn2 = wave % P2N;
    int a_rd[2], b1_rd[2], a_rd2[2], b2_rd[2];            // [0]: the hi halves of a K-step's row (16-byte pieces 0..3), [1]: the lo halves (4..7)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fq) ^ (frow & 7)) << 4;
        a_rd[kk] = (wm * (BM / WM) + frow) * kRowBytes + sw;
        b1_rd[kk] = A_BYTES + (wn * (CMID / WN) + frow) * kRowBytes + sw;
        a_rd2[kk] = (wm2 * (BM / P2M) + frow) * kRowBytes + sw;
        b2_rd[kk] = T_BYTES + (wn2 * COLS2 + frow) * kRowBytes + sw;
    }
    f32x4 acc1[MF][NF1];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF1; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int nk = 9 * KC;
    int cc = 0, kw = 0, kh = 0, issued = 0, ibuf = 0;
    set_tap(0, 0);
    auto issue_next = [&]() {
        unsigned char* sa = smem + ibuf * STAGE + wave * A_IT * 8 * kRowBytes;
        unsigned char* sb = smem + ibuf * STAGE + A_BYTES + wave * B_IT * 8 * kRowBytes;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16(in_rsrc, a_voff[i], cc * kRowBytes, sa + i * 8 * kRowBytes);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16(w1_rsrc, w_voff[i], issued * kRowBytes, sb + i * 8 * kRowBytes);
        if (++kw == 3) {
            kw = 0;
            if (++kh == 3) { kh = 0; ++cc; }
        }
        set_tap(kh, kw);
        ++issued;
        ibuf ^= 1;
    };
    // shortcut rows of output tile t: requested early (tile 0 underneath the whole 3x3 phase), as raw [8 hi][8 lo] bits
    RowAddr ra[MF2];
    {
        PixWalk pw;
        pw.init(m0 + wm2 * (BM / P2M) + frow, HW, W);
        const PixWalk first = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < MF2; ++i) {
            const int m = m0 + wm2 * (BM / P2M) + i * 16 + frow;
            ra[i] = row_addr_at(d2, m < Mg ? pw : first, W, H, G2.H_res, G2.W_res, G2.out_off, G2.out_bstride, G2.res_off, G2.res_bstride);
            if (i + 1 < MF2) pw.advance(16, H, W);
        }
    }
    const bool has_res = d2.residual != nullptr;
    f32x8 rpre[MF2][NF2 / 2];
    auto prefetch_res = [&](int t, int jj) {
#pragma unroll
        for (int i = 0; i < MF2; ++i) {
            const int n = t * 128 + wn2 * COLS2 + jj * 32 + fq * 8;
            const char* p = x3_addr(d2.residual, ra[i].rbase, n);          // rows past the end were clamped to row 0: a valid address
            rpre[i][jj].lo = *(const f32x4*)p;
            rpre[i][jj].hi = *(const f32x4*)(p + 64);
        }
    };
    if (has_res) {
#pragma unroll
        for (int jj = 0; jj < NF2 / 2; ++jj) prefetch_res(0, jj);
    }
    issue_next();
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issued < nk) issue_next();
        const unsigned char* sbase = smem + (ks & 1) * STAGE;
        xh8 ah[MF], al[MF], bh[NF1], bl[NF1];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            ah[i] = *(const xh8*)(sbase + a_rd[0] + i * 16 * kRowBytes);
            al[i] = *(const xh8*)(sbase + a_rd[1] + i * 16 * kRowBytes);
        }
#pragma unroll
        for (int j = 0; j < NF1; ++j) {
            bh[j] = *(const xh8*)(sbase + b1_rd[0] + j * 16 * kRowBytes);
            bl[j] = *(const xh8*)(sbase + b1_rd[1] + j * 16 * kRowBytes);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF1; ++j) {
                const int js = GPP_SERP2(i, j, NF1);
                acc1[i][js] = X3Half<DT>::mfma(bl[js], ah[i], acc1[i][js]);
                acc1[i][js] = X3Half<DT>::mfma(bh[js], ah[i], acc1[i][js]);
            }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF1; ++j) { const int js = GPP_SERP2(i, j, NF1); acc1[i][js] = X3Half<DT>::mfma(bh[js], al[i], acc1[i][js]); }
    }

    // ---- hand-over: everyone is done with the ring; W2 tile 0 streams in while the intermediate tile is written as pre-split rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int w2_voff[W2_IT];
#pragma unroll
    for (int i = 0; i < W2_IT; ++i) w2_voff[i] = ((wave * W2_IT + i) * 8 + srow) * CMID * 4 + gchunk * 16;
    auto stage_w2 = [&](int t) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int i = 0; i < W2_IT; ++i)
                glds16(w2_rsrc, w2_voff[i], t * 128 * CMID * 4 + kc * kRowBytes,
                       smem + T_BYTES + kc * 128 * kRowBytes + (wave * W2_IT + i) * 8 * kRowBytes);
    };
    stage_w2(0);
    {
        constexpr int COLS1 = CMID / WN;
#pragma unroll
        for (int jj = 0; jj < NF1 / 2; ++jj) {
            const int n = wn * COLS1 + jj * 32 + fq * 8;              // 8 consecutive intermediate channels
            float bias_v[8], scale_v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bias_v[e] = d1.bias ? d1.bias[n + e] : 0.0f;
                scale_v[e] = (OSCALE && d1.out_scale) ? d1.out_scale[n + e] : 1.0f;
            }
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int r = wm * (BM / WM) + i * 16 + frow;
                xh8 h, l;
                float v8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = e < 4 ? acc1[i][2 * jj][e] : acc1[i][2 * jj + 1][e - 4];
                    if constexpr (OSCALE) v = v * scale_v[e] + bias_v[e];
                    else v = v + bias_v[e];
                    if (d1.relu) v = fmaxf(v, 0.0f);
                    v8[e] = v;
                }
                x3_range<DT>(v8, (unsigned long long*)d1.range_counter);      // exactly what x3_store does to the map the unfused layer writes
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    h[e] = (typename X3Half<DT>::half)v8[e];
                    l[e] = (typename X3Half<DT>::half)(v8[e] - (float)h[e]);
                }
                const int piece = (n & 31) >> 3;                      // 16-byte piece of the hi half of the 128-byte row; lo: + 4
                unsigned char* row = smem + (n >> 5) * A_BYTES + r * kRowBytes;
                *(xh8*)(row + ((piece ^ (r & 7)) << 4)) = h;
                *(xh8*)(row + (((4 + piece) ^ (r & 7)) << 4)) = l;
            }
        }
    }

    // ---- phase 2: y tile = T (BM x CMID) * W2^T, 128 output channels at a time
    const int n2_tiles = d2.C_out / 128;
    for (int t = 0; t < n2_tiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // W2 tile t (and, for t = 0, T) is in LDS
        asm volatile("" ::: "memory");
        f32x4 acc2[MF2][NF2];
#pragma unroll
        for (int i = 0; i < MF2; ++i)
#pragma unroll
            for (int j = 0; j < NF2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            xh8 ah[MF2], al[MF2], bh[NF2], bl[NF2];
#pragma unroll
            for (int i = 0; i < MF2; ++i) {
                ah[i] = *(const xh8*)(smem + kc * A_BYTES + a_rd2[0] + i * 16 * kRowBytes);
                al[i] = *(const xh8*)(smem + kc * A_BYTES + a_rd2[1] + i * 16 * kRowBytes);
            }
#pragma unroll
            for (int j = 0; j < NF2; ++j) {
                bh[j] = *(const xh8*)(smem + kc * 128 * kRowBytes + b2_rd[0] + j * 16 * kRowBytes);
                bl[j] = *(const xh8*)(smem + kc * 128 * kRowBytes + b2_rd[1] + j * 16 * kRowBytes);
            }
#pragma unroll
            for (int i = 0; i < MF2; ++i)
#pragma unroll
                for (int j = 0; j < NF2; ++j) {
                    const int js = GPP_SERP2(i, j, NF2);
                    acc2[i][js] = X3Half<DT>::mfma(bl[js], ah[i], acc2[i][js]);
                    acc2[i][js] = X3Half<DT>::mfma(bh[js], ah[i], acc2[i][js]);
                }
#pragma unroll
            for (int i = 0; i < MF2; ++i)
#pragma unroll
                for (int j = 0; j < NF2; ++j) { const int js = GPP_SERP2(i, j, NF2); acc2[i][js] = X3Half<DT>::mfma(bh[js], al[i], acc2[i][js]); }
        }
        if (t + 1 < n2_tiles) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // everyone has read W2 tile t
            asm volatile("" ::: "memory");
            stage_w2(t + 1);                                           // streams in under the epilogue below
        }
        // per 8-channel group: scale + bias + shortcut, refill that group's shortcut registers for the next tile, ReLU + split + store
#pragma unroll
        for (int jj = 0; jj < NF2 / 2; ++jj) {
            const int n = t * 128 + wn2 * COLS2 + jj * 32 + fq * 8;
            float bias_v[8], scale_v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bias_v[e] = d2.bias ? d2.bias[n + e] : 0.0f;
                scale_v[e] = (OSCALE && d2.out_scale) ? d2.out_scale[n + e] : 1.0f;
            }
            float outv[MF2][8];
#pragma unroll
            for (int i = 0; i < MF2; ++i) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = e < 4 ? acc2[i][2 * jj][e] : acc2[i][2 * jj + 1][e - 4];
                    if constexpr (OSCALE) outv[i][e] = a * scale_v[e] + bias_v[e];
                    else outv[i][e] = a + bias_v[e];
                }
                if (has_res) {
                    float r[8];
                    x3_unpack<DT>(rpre[i][jj].lo, rpre[i][jj].hi, r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) outv[i][e] += r[e];
                }
            }
            if (has_res && t + 1 < n2_tiles) prefetch_res(t + 1, jj);
#pragma unroll
            for (int i = 0; i < MF2; ++i) {
                const int m = m0 + wm2 * (BM / P2M) + i * 16 + frow;
                if (m >= Mg) continue;
                finish8_pre<DT>(d2, outv[i], n, ra[i].obase, false, f32x8());
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Weight-stationary, persistent form of the shallow 1 x 1 layers of the x3 types (round 4; tile codes 4000000 + BM * 1000 + BN: 4128064, 4064064,
// 4128128, 4064128).
// The 1 x 1 layers of res2 - res4 with 2 - 16 K-steps take their matrix time PLUS their memory time in the tile kernels above: a workgroup's
// K loop is a chain of dependent tile loads one stage ahead, its epilogue and stores follow, and only two or three such workgroups share a CU
// (profiles/r4/layer_pmc_f16x3.txt: wavefronts waiting 0.3 - 0.6 of their cycles, traffic = algorithmic).  Here a workgroup of 8 wavefronts
//   * keeps ONE n-tile of the weights (BN = 64 or 128 output channels x all of K, <= 128 KB) resident in LDS for its whole life,
//   * walks a sequence of M tiles and streams their activation rows through a ring of R = 4 slabs (one slab = BM rows x one 32-channel
//     K-step = what a K-step of the tile kernels stages), D = 3 slabs in flight AHEAD of the one computed -- across tile boundaries: the
//     loads of the next tile's first slabs and of its shortcut rows are in flight while this tile's epilogue runs,
//   * one s_barrier per K-step as before; waits on the LDS-DMA are COUNTED (vmcnt((D - 1) A_IT): every younger operation is a load, and
//     loads return in order) except at the first K-step of a tile, where a vmcnt(0) also retires the previous tile's stores (a store
//     younger than the awaited load would make a counted wait unsafe: loads and stores return out of order with respect to each other).
// Work split: 256 workgroups = 8 XCDs x 32; the n_tiles = C_out / BN workgroups that share a sequence of M tiles sit on ONE XCD and walk it
// together, so the activation slabs they all read come from that XCD's L2 once they have been fetched.
// Every output element is summed in the order of the tile kernels (K-steps ascending; hi * wlo, hi * whi, lo * whi per step; scale, bias,
// shortcut, ReLU, range, split in the same arithmetic): bit-identical to them (tests/test_conv_f16x3_gpu.py).
template <int DT, int BM, int BN, bool HASRES>
__global__ __launch_bounds__(512, 2) void conv1x1_ws_kernel(const gpp_conv_desc d, const int n_mtiles)
{
    static_assert(kX3<DT>, "x3 types on pre-split maps");
    using xh8 = typename X3Half<DT>::vec;
    constexpr int NW = 8, WM = 4, WN = 2;
    constexpr int MF = BM / WM / 16, NF = BN / WN / 16;
    constexpr int A_IT = BM / 8 / NW, B_IT = BN / 8 / NW;
    constexpr int NG = NF / 2;                                   // groups of 8 consecutive output channels per lane and row
    constexpr int R = 4, D = R - 1;
    constexpr int SLAB = BM * kRowBytes;
    constexpr bool OSCALE = (DT == GPP_F16X3);
    static_assert(MF >= 1 && NF % 2 == 0 && A_IT >= 1 && B_IT >= 1 && BN % (8 * NW) == 0, "tile shape: 64 or 128 output channels");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const gpp_conv_group& G = d.groups[0];
    const int HoWo = G.H_out * G.W_out, Mg = d.batch * HoWo;
    const int nk = d.C_in / 32;                                  // K-steps (KH = KW = 1)
    const int n_tiles = d.C_out / BN;                            // divides 32 (checked by the launcher)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int nt = local % n_tiles, per_xcd = 32 / n_tiles;
    const int part = xcd * per_xcd + local / n_tiles, P = 8 * per_xcd;
    const int n0 = nt * BN;
    const int my_tiles = __builtin_amdgcn_readfirstlane(part < n_mtiles ? (n_mtiles - part + P - 1) / P : 0);
    if (my_tiles == 0) return;
    unsigned char* const wbase = smem;                           // nk slabs of BN weight rows, resident
    unsigned char* const ring = smem + nk * BN * kRowBytes;      // R activation slabs

    const int srow = lane >> 3, gchunk = (lane & 7) ^ srow;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.weight, 0, d.weight_bytes, 0x00020000);
    {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int w_voff = (n0 + (wave * B_IT + i) * 8 + srow) * d.C_in * 4 + gchunk * 16;
            for (int ks = 0; ks < nk; ++ks) glds16(w_rsrc, w_voff, ks * kRowBytes, wbase + ks * BN * kRowBytes + (wave * B_IT + i) * 8 * kRowBytes);
        }
    }
    // byte offsets of the activation rows this lane stages for M tile number c of this workgroup's sequence
    int a_off[A_IT];
    auto a_offsets = [&](int c) {
        const int m0 = (part + c * P) * BM;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + (wave * A_IT + i) * 8 + srow;
            a_off[i] = kOutOfRange;
            if (m < Mg) {
                const int b = m / HoWo, p = m - b * HoWo;
                a_off[i] = (int)((G.in_off + (int64_t)b * G.in_bstride + (int64_t)p * d.in_pitch) * 4) + gchunk * 16;
            }
        }
    };
    a_offsets(0);
    int ic = 0, iks = 0, islot = 0;                              // issue cursor: tile, K-step, ring slot
    auto issue = [&]() {
        const bool live = ic < my_tiles;                         // past the last slab: a zero-length descriptor (the loads are dropped, the count stays)
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)d.in, 0, live ? d.in_bytes : 0, 0x00020000);
        const int so = __builtin_amdgcn_readfirstlane(iks * kRowBytes);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16(ra, a_off[i], so, ring + islot * SLAB + (wave * A_IT + i) * 8 * kRowBytes);
        islot = (islot + 1) & (R - 1);
        if (++iks == nk) {
            iks = 0;
            ++ic;
            if (ic < my_tiles) a_offsets(ic);
        }
    };
#pragma unroll
    for (int p = 0; p < D; ++p) issue();

    const int frow = lane & 15, fq = lane >> 4;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fq) ^ (frow & 7)) << 4;
        a_rd[kk] = (wm * (BM / WM) + frow) * kRowBytes + sw;
        b_rd[kk] = (wn * (BN / WN) + frow) * kRowBytes + sw;
    }
    // scale and bias of this lane's 8 output channels: the same for every tile of the workgroup
    const int n = n0 + wn * (BN / WN) + fq * 8;                   // group jj: channels n + 32 jj ... + 7
    float bias_v[NG][8], scale_v[NG][8];
#pragma unroll
    for (int jj = 0; jj < NG; ++jj)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bias_v[jj][e] = d.bias ? d.bias[n + 32 * jj + e] : 0.0f;
            scale_v[jj][e] = (OSCALE && d.out_scale) ? d.out_scale[n + 32 * jj + e] : 1.0f;
        }
    // output / shortcut rows of a tile (element offsets), and the shortcut rows' raw [8 hi][8 lo] bits
    int64_t obase[MF], rbase[MF];
    bool valid[MF];
    auto rows_of = [&](int c, int64_t (&ob)[MF], int64_t (&rb)[MF], bool (&ok)[MF]) {
        const int m0 = (part + c * P) * BM;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = m0 + wm * (BM / WM) + i * 16 + frow;
            ok[i] = m < Mg;
            const int mm = ok[i] ? m : 0;                        // (rows past the end: row 0, a valid address that is never stored to)
            const int b = mm / HoWo, p = mm - b * HoWo;
            ob[i] = G.out_off + (int64_t)b * G.out_bstride + (int64_t)p * d.out_pitch;
            rb[i] = G.res_off + (int64_t)b * G.res_bstride + (int64_t)p * d.res_pitch;
        }
    };
    f32x8 rpre[HASRES ? MF : 1][NG];
    auto fetch_res = [&](const int64_t (&rb)[MF]) {
        if constexpr (HASRES) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int jj = 0; jj < NG; ++jj) {
                    const char* q = x3_addr(d.residual, rb[i], n + 32 * jj);
                    rpre[i][jj].lo = *(const f32x4*)q;
                    rpre[i][jj].hi = *(const f32x4*)(q + 64);
                }
        }
    };
    rows_of(0, obase, rbase, valid);
    fetch_res(rbase);

    f32x4 acc[MF][NF];
    for (int c = 0; c < my_tiles; ++c) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int slot = (c * nk) & (R - 1);
        for (int ks = 0; ks < nk; ++ks) {
            // slab (c, ks) has landed: at the first K-step of a tile everything older is retired too (the previous tile's stores: see above)
            if (ks == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * A_IT) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue();                                             // slab (c, ks) + D goes into the slot everyone has just finished reading
            const unsigned char* sa = ring + slot * SLAB;
            const unsigned char* sw = wbase + ks * BN * kRowBytes;
            xh8 ah[MF], al[MF], bh[NF], bl[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                ah[i] = *(const xh8*)(sa + a_rd[0] + i * 16 * kRowBytes);
                al[i] = *(const xh8*)(sa + a_rd[1] + i * 16 * kRowBytes);
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                bh[j] = *(const xh8*)(sw + b_rd[0] + j * 16 * kRowBytes);
                bl[j] = *(const xh8*)(sw + b_rd[1] + j * 16 * kRowBytes);
            }
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int js = GPP_SERP2(i, j, NF);
                    acc[i][js] = X3Half<DT>::mfma(bl[js], ah[i], acc[i][js]);
                    acc[i][js] = X3Half<DT>::mfma(bh[js], ah[i], acc[i][js]);
                }
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) { const int js = GPP_SERP2(i, j, NF); acc[i][js] = X3Half<DT>::mfma(bh[js], al[i], acc[i][js]); }
            slot = (slot + 1) & (R - 1);
        }
        // ---- epilogue of tile c; the slabs of the next tiles are in flight underneath it
        float v[MF][NG][8];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int jj = 0; jj < NG; ++jj) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (OSCALE) {
                        v[i][jj][e] = __builtin_fmaf(acc[i][2 * jj][e], scale_v[jj][e], bias_v[jj][e]);
                        v[i][jj][4 + e] = __builtin_fmaf(acc[i][2 * jj + 1][e], scale_v[jj][4 + e], bias_v[jj][4 + e]);
                    } else {
                        v[i][jj][e] = acc[i][2 * jj][e] + bias_v[jj][e];
                        v[i][jj][4 + e] = acc[i][2 * jj + 1][e] + bias_v[jj][4 + e];
                    }
                }
                if constexpr (HASRES) {
                    float r[8];
                    x3_unpack<DT>(rpre[i][jj].lo, rpre[i][jj].hi, r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[i][jj][e] += r[e];
                }
            }
        int64_t ob_next[MF], rb_next[MF];
        bool ok_next[MF];
        if (c + 1 < my_tiles) {                                  // the next tile's shortcut rows: requested BEFORE this tile's stores go out
            rows_of(c + 1, ob_next, rb_next, ok_next);
            fetch_res(rb_next);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
            if (valid[i]) {
#pragma unroll
                for (int jj = 0; jj < NG; ++jj) finish8_pre<DT>(d, v[i][jj], n + 32 * jj, obase[i], false, f32x8());
            }
        if (c + 1 < my_tiles) {
#pragma unroll
            for (int i = 0; i < MF; ++i) { obase[i] = ob_next[i]; rbase[i] = rb_next[i]; valid[i] = ok_next[i]; }
        }
    }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember which devices a kernel has been configured
// on (one bit per device ordinal; racing first calls both set the same value)
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    template <typename K>
    int configure(K kernel, int lds)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        const uint64_t bit = 1ull << (dev & 63);
        if (mask.load(std::memory_order_acquire) & bit) return GPP_OK;
        e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        mask.fetch_or(bit, std::memory_order_release);
        return GPP_OK;
    }
};

// One tile configuration: block tile BM x BN, WM x WN wavefronts, STAGES-deep LDS ring.
template <int DT, int BM, int BN, int WM, int WN, int STAGES, bool PIPE, bool XIN = false>
int launch(gpp_conv_desc& d, hipStream_t st)
{
    if constexpr (kX3<DT> && !XIN) {
        if (d.x3_split & GPP_X3_IN) return launch<DT, BM, BN, WM, WN, STAGES, PIPE, true>(d, st);      // the pre-split-input form of this tile
    }
    constexpr int BMS = stage_rows(BM, WM * WN);
    constexpr int lds = STAGES * (BMS + BN) * kRowBytes;
    constexpr int CK = kRowBytes / Elem<DT>::ESZ;
    static DeviceOnce once;
    auto kernel = conv_igemm_kernel<DT, BM, BN, WM, WN, STAGES, PIPE, XIN>;
    int rc = once.configure(kernel, lds);
    if (rc != GPP_OK) return rc;
    const int tiles = prepare<BM, BN>(d);
    if (tiles < 0) return tiles;
    const int n_tiles = (d.C_out + BN - 1) / BN;
    // split-K: d.split_k partitions of the K-steps (the caller's explicit choice, or the library's batch-independent rule,
    // gpp_conv2d_split_rule, applied by the entry point before it gets here); every split keeps >= 1 K-step and the
    // partial slabs [split][tiles*BM][n_tiles*BN] float32 must fit the workspace -- otherwise GPP_ERR_WORKSPACE, never
    // a silently different summation order
    const int nk = d.KH * d.KW * (d.C_in / CK);
    int nsplit = 1;
    if (d.split_k > 1) {
        nsplit = d.split_k;
        if (nk / nsplit < 1) return GPP_ERR_BAD_ARG;
        const int64_t slab = (int64_t)tiles * BM * n_tiles * BN * 4;
        if (!d.partial || slab * nsplit > (int64_t)d.partial_bytes) return GPP_ERR_WORKSPACE;
    }
    // a layer with fewer K-steps than ring slots (1x1 convs with C_in = 64) only touches the first slots:
    // declaring just those lets more workgroups share a CU, which is what the HBM-bound layers need
    const int steps = (nk + nsplit - 1) / nsplit;
    // (the pipelined loops always touch both buffers -- with a single K-step the look-ahead reads fragments of buffer 1 that nobody
    // uses: they get the whole ring whatever the step count, never less LDS than they address)
    const int lds_used = PIPE ? lds : (steps < STAGES ? steps : STAGES) * (BMS + BN) * kRowBytes;
    kernel<<<dim3((unsigned)(tiles * n_tiles), (unsigned)nsplit), dim3(64 * WM * WN), lds_used, st>>>(d);
    if (nsplit > 1) {
        const int64_t total = (int64_t)d.partial_rows * ((d.C_out + 7) / 8);
        splitk_reduce_kernel<DT><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(d, BM, n_tiles * BN, nsplit);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

// conv_igemm_dual_kernel: C_out = 256 k + 128.  The caller's descriptor is split into the two column blocks here.
template <int DT>
int launch_dual(const gpp_conv_desc& d, hipStream_t st)
{
    // 16-bit types, and GPP_BF16X3 on a pre-split input map (its pipelined three-phase loop)
    constexpr bool X3 = kX3<DT>;
    constexpr int ESZ = Elem<DT>::ESZ, CK = kRowBytes / ESZ;
    if (X3 && !(d.x3_split & GPP_X3_IN)) return GPP_ERR_UNSUPPORTED;
    if (d.C_out < 384 || d.C_out % 256 != 128 || d.KH * d.KW * (d.C_in / CK) < 2 || d.split_k > 1) return GPP_ERR_UNSUPPORTED;
    constexpr int lds = 2 * (512 + 128) * kRowBytes;          // 160 KB: the larger of the two bodies
    static DeviceOnce once;
    auto kernel = conv_igemm_dual_kernel<DT, X3>;
    int rc = once.configure(kernel, lds);
    if (rc != GPP_OK) return rc;
    const int head = d.C_out - 128;                           // columns of the 256-wide part: a multiple of 256
    gpp_conv_desc d0 = d, d1 = d;
    d0.C_out = head;
    d1.C_out = 128;
    d1.weight = (const char*)d.weight + (int64_t)head * d.KH * d.KW * d.C_in * ESZ;
    d1.weight_rows = d.weight_rows - head;
    if (d.bias) d1.bias = d.bias + head;
    if (d.out_scale) d1.out_scale = d.out_scale + head;
    for (int g = 0; g < d.n_groups; ++g) {
        d1.groups[g].out_off += head;
        d1.groups[g].res_off += head;
    }
    const int t0 = prepare<256, 256>(d0), t1 = prepare<512, 128>(d1);
    if (t0 < 0) return t0;
    if (t1 < 0) return t1;
    const int n0 = t0 * (head / 256), n1 = t1;
    const int split0 = (n0 + 7) / 8 * 8;                      // keeps workgroup index % 8 = XCD for the second range's remap
    kernel<<<dim3((unsigned)(split0 + n1)), dim3(512), lds, st>>>(d0, d1, n0, split0, n1);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

// conv_igemm_mix_kernel: 256-column tiles of two heights in one grid (x3 types on pre-split input maps: the three-phase loop).
// The split is arithmetic on the layer and the 256 CUs of the device: part A = as many WHOLE rounds of BMA-row tiles as group 0 holds
// and as make the total cheapest, part B = the rest in BMB-row tiles; cost model = rounds x tile rows (a round of one-workgroup-per-CU
// tiles takes as long as its tile is tall).  Nothing to gain, or no whole round fits: GPP_ERR_UNSUPPORTED (the tuner moves on).
template <int DT, int BMA, int BMB>
int launch_mix(const gpp_conv_desc& d, hipStream_t st)
{
    constexpr int BN = 256, CUS = 256;
    constexpr int ESZ = Elem<DT>::ESZ, CK = kRowBytes / ESZ;
    static_assert(kX3<DT>, "mixed grids exist for the x3 types");
    if (!(d.x3_split & GPP_X3_IN)) return GPP_ERR_UNSUPPORTED;
    if (d.C_out % BN != 0 || d.KH * d.KW * (d.C_in / CK) < 2 || d.split_k > 1) return GPP_ERR_UNSUPPORTED;
    const int n_tiles = d.C_out / BN;
    const int64_t rows0 = (int64_t)d.batch * d.groups[0].H_out * d.groups[0].W_out;
    auto tiles_b = [&](int64_t begin, int bm) {
        int64_t t = (rows0 - begin + bm - 1) / bm;
        for (int g = 1; g < d.n_groups; ++g) t += ((int64_t)d.batch * d.groups[g].H_out * d.groups[g].W_out + bm - 1) / bm;
        return t * n_tiles;
    };
    int64_t best_cost = (tiles_b(0, BMA) + CUS - 1) / CUS * BMA;          // the uniform BMA grid: what the mix has to beat
    int best_ra = 0;
    for (int ra = 1; ra <= 64; ++ra) {
        if ((ra * CUS) % n_tiles != 0) continue;
        const int64_t ma = (int64_t)ra * CUS / n_tiles;                     // M tiles of part A
        if (ma * BMA > rows0) break;
        const int64_t cost = (int64_t)ra * BMA + (tiles_b(ma * BMA, BMB) + CUS - 1) / CUS * BMB;
        if (cost < best_cost) { best_cost = cost; best_ra = ra; }
    }
    if (best_ra == 0) return GPP_ERR_UNSUPPORTED;
    constexpr int lds = 2 * (stage_rows(BMA > BMB ? BMA : BMB, 8) + BN) * kRowBytes;
    static DeviceOnce once;
    auto kernel = conv_igemm_mix_kernel<DT, BMA, BMB, BN, true>;
    int rc = once.configure(kernel, lds);
    if (rc != GPP_OK) return rc;
    const int ma = best_ra * CUS / n_tiles;
    gpp_conv_desc da = d, db = d;
    da.n_groups = 1;                                           // part A: whole tiles of group 0 only
    const int ta = prepare<BMA, BN>(da);
    if (ta < 0) return ta;
    db.groups[0].row_begin = ma * BMA;
    const int tb = prepare<BMB, BN>(db);
    if (tb < 0) return tb;
    da.in_bytes = db.in_bytes;                                 // (extents of the whole buffer, as computed over every group)
    const int na = ma * n_tiles, nb = tb * n_tiles;
    const int splita = (na + 7) / 8 * 8;                       // keeps workgroup index % 8 = XCD for the second range's remap
    kernel<<<dim3((unsigned)(splita + nb)), dim3(512), lds, st>>>(da, db, na, splita, nb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

// conv1x1_ws_kernel: which layers it takes, and how it is launched (256 workgroups, W n-tile + ring in LDS).
template <int DT, int BM, int BN>
int launch_ws(gpp_conv_desc& d, hipStream_t st)
{
    constexpr int R = 4;
    static_assert(kX3<DT>, "weight-stationary 1 x 1: x3 types");
    const gpp_conv_group& G = d.groups[0];
    if (!(d.x3_split & GPP_X3_IN) || d.KH != 1 || d.KW != 1 || d.stride != 1 || d.pad_top != 0 || d.pad_left != 0 || d.n_groups != 1 || d.split_k > 1)
        return GPP_ERR_UNSUPPORTED;
    if (d.C_out % BN != 0 || 32 % (d.C_out / BN) != 0 || d.C_in % 32 != 0) return GPP_ERR_UNSUPPORTED;
    if (G.H_in != G.H_out || G.W_in != G.W_out) return GPP_ERR_UNSUPPORTED;
    if (d.residual && (!(d.x3_split & GPP_X3_RES) || G.H_res != G.H_out || G.W_res != G.W_out)) return GPP_ERR_UNSUPPORTED;   // shortcut: pre-split, same size
    const int nk = d.C_in / 32;
    const int lds = nk * BN * kRowBytes + R * BM * kRowBytes;
    if (lds > 160 * 1024) return GPP_ERR_UNSUPPORTED;
    const int tiles = prepare<BM, BN>(d);                       // buffer extents (+ the weight-row check)
    if (tiles < 0) return tiles;
    const int n_mtiles = (int)(((int64_t)d.batch * G.H_out * G.W_out + BM - 1) / BM);
    int rc;
    if (d.residual) {
        static DeviceOnce once;
        auto kernel = conv1x1_ws_kernel<DT, BM, BN, true>;
        rc = once.configure(kernel, 160 * 1024);
        if (rc != GPP_OK) return rc;
        kernel<<<dim3(256), dim3(512), lds, st>>>(d, n_mtiles);
    } else {
        static DeviceOnce once;
        auto kernel = conv1x1_ws_kernel<DT, BM, BN, false>;
        rc = once.configure(kernel, 160 * 1024);
        if (rc != GPP_OK) return rc;
        kernel<<<dim3(256), dim3(512), lds, st>>>(d, n_mtiles);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

template <int DT>
int dispatch(gpp_conv_desc& d, hipStream_t st)
{
    switch (d.tile_hint) {               // explicit choices (BM*1000 + BN, or the legacy codes): what the host-side autotuner hands in
        case 64:
        case 128064: return launch<DT, 128, 64, 2, 2, 2, false>(d, st);
        case 64064: return launch<DT, 64, 64, 2, 2, 2, false>(d, st);
        case 96064: return launch<DT, 96, 64, 2, 2, 2, false>(d, st);
        case 160064: return launch<DT, 160, 64, 2, 2, 2, false>(d, st);
        case 192064: return launch<DT, 192, 64, 2, 2, 2, false>(d, st);
        case 64128: return launch<DT, 64, 128, 2, 2, 2, false>(d, st);
        case 96128: return launch<DT, 96, 128, 2, 2, 2, false>(d, st);
        case 128:
        case 128128:
#ifdef GPP_STAMPS
            if constexpr (!kF32Storage<DT>) if (d.reserved & 4) return launch<DT, 128, 128, 2, 2, 2, true>(d, st);
#endif
            return launch<DT, 128, 128, 2, 2, 2, false>(d, st);
        case 160128: return launch<DT, 160, 128, 2, 2, 2, false>(d, st);
        case 192128: return launch<DT, 192, 128, 2, 2, 2, false>(d, st);
        case 224128: return launch<DT, 224, 128, 2, 2, 2, false>(d, st);
        // N-remainder tiles (4 x 1 wavefronts, wave tile BM/4 x 160): layers whose C_out is far from a multiple of 128
        // (regression outputs: 144 -> 160 instead of 256 columns; measured 200 -> 162 us).  96-wide tiles and 3/4-deep
        // LDS rings on the small tiles were measured too and lost everywhere (fewer workgroups per CU).
        case 128160: return launch<DT, 128, 160, 4, 1, 2, false>(d, st);
        case 192160:
            if constexpr (kX3<DT>) {
                // float32-input form of this tile: 3 registers over the budget (scratch); only the pre-split form exists
                if (!(d.x3_split & GPP_X3_IN)) return GPP_ERR_UNSUPPORTED;
                return launch<DT, 192, 160, 4, 1, 2, false, true>(d, st);
            } else {
                return launch<DT, 192, 160, 4, 1, 2, false>(d, st);
            }
        case 0: break;
        default:
            // the software-pipelined / 8-wavefront forms exist for the 16-bit types only: the float32 path is bound by the
            // matrix pipe (8x the MFMA time per staged byte) and gains nothing from them
            if constexpr (!kF32Storage<DT>) {
                switch (d.tile_hint) {
                    case 256: return launch<DT, 256, 128, 4, 2, 3, false>(d, st);          // 3-deep ring, experiments only
                    case 1128128: return launch<DT, 128, 128, 2, 2, 2, true>(d, st);        // 1000000 + ...: the pipelined main loop
                    case 1192128: return launch<DT, 192, 128, 2, 2, 2, true>(d, st);
                    case 1128256: return launch<DT, 128, 256, 2, 4, 2, true>(d, st);
                    case 1192256: return launch<DT, 192, 256, 2, 4, 2, true>(d, st);
                    case 1192160: return launch<DT, 192, 160, 4, 1, 2, true>(d, st);
                    case 1128160: return launch<DT, 128, 160, 4, 1, 2, true>(d, st);
                    case 1192096: return launch<DT, 192, 96, 4, 1, 2, true>(d, st);          // 96-channel outputs (the classification logits)
                    case 2256256: return launch_dual<DT>(d, st);            // 256 x 256 tiles + 512 x 128 tiles for the last 128 columns, one grid
                    case 512:
                    case 256256:
#ifdef GPP_STAMPS
                        if (d.reserved & 8) return launch<DT, 256, 256, 2, 4, 2, false>(d, st);
#endif
                        return launch<DT, 256, 256, 2, 4, 2, true>(d, st);
                    default: return GPP_ERR_BAD_ARG;
                }
            } else {
                // GPP_BF16X3 spends 3 MFMAs per fragment pair: with 4-wavefront tiles its LDS traffic equals its matrix time;
                // the 8-wavefront 256-column tiles (plain two-buffer loop) halve the LDS bytes per MFMA
                if constexpr (kX3<DT>) {
                    if (d.x3_split & GPP_X3_IN) {
                        // the software-pipelined three-phase loop: pre-split input maps only (1000000 + tile, as for the 16-bit types)
                        switch (d.tile_hint) {
                            case 1256256: return launch<DT, 256, 256, 2, 4, 2, true, true>(d, st);
                            case 1224256: return launch<DT, 224, 256, 2, 4, 2, true, true>(d, st);      // (224 / 160 rows: staged as 256 / 192, see BMS)
                            case 1192256: return launch<DT, 192, 256, 2, 4, 2, true, true>(d, st);
                            case 1160256: return launch<DT, 160, 256, 2, 4, 2, true, true>(d, st);
                            case 1128256: return launch<DT, 128, 256, 2, 4, 2, true, true>(d, st);
                            case 1192128: return launch<DT, 192, 128, 2, 2, 2, true, true>(d, st);
                            case 1128128: return launch<DT, 128, 128, 2, 2, 2, true, true>(d, st);
                            case 1128160: return launch<DT, 128, 160, 4, 1, 2, true, true>(d, st);      // 144-channel outputs: three-phase loop on a 4 x 1 layout
                            case 1192096: return launch<DT, 192, 96, 4, 1, 2, true, true>(d, st);       // 96-channel outputs, same layout
                            case 2256256: return launch_dual<DT>(d, st);      // C_out = 256 k + 128: the dual-shape grid
                            case 3256224: return launch_mix<DT, 256, 224>(d, st);       // 3000000 + BMA * 1000 + BMB: two tile heights, one grid
                            case 3192160: return launch_mix<DT, 192, 160>(d, st);
                            case 4128064: return launch_ws<DT, 128, 64>(d, st);        // 4000000 + BM * 1000 + BN: weight-stationary persistent 1 x 1
                            case 4064064: return launch_ws<DT, 64, 64>(d, st);
                            case 4128128: return launch_ws<DT, 128, 128>(d, st);
                            case 4064128: return launch_ws<DT, 64, 128>(d, st);
                            // 5000000 + BM * 1000 + BN: the plain loop on a FOUR-deep LDS ring (three K-steps of LDS-DMA in flight ahead of the one
                            // computed).  For the launches that field at most one workgroup per CU anyway -- deep-K, small-M layers, every layer
                            // at batch 1 -- whose K-steps are bound by the latency of their own tile loads; where more workgroups would share a CU
                            // the larger footprint loses (round 2: 1.6 x slower at B = 8) and the tuner keeps the two-deep tiles.
                            case 5064064: return launch<DT, 64, 64, 2, 2, 4, false, true>(d, st);
                            case 5096064: return launch<DT, 96, 64, 2, 2, 4, false, true>(d, st);
                            case 5064128: return launch<DT, 64, 128, 2, 2, 4, false, true>(d, st);
                            case 5096128: return launch<DT, 96, 128, 2, 2, 4, false, true>(d, st);
                            case 5128128: return launch<DT, 128, 128, 2, 2, 4, false, true>(d, st);
                            default: break;
                        }
                    }
                    if (d.tile_hint == 256256) return launch<DT, 256, 256, 2, 4, 2, false>(d, st);
                    if (d.tile_hint == 192256) return launch<DT, 192, 256, 2, 4, 2, false>(d, st);
                    if (d.tile_hint == 128256) return launch<DT, 128, 256, 2, 4, 2, false>(d, st);
                }
                switch (d.tile_hint) {
                    case 256: case 1128128: case 1192128: case 1128256: case 1192256: case 1192160: case 1128160: case 2256256: case 512: case 256256: case 1256256:
                    case 192256: case 128256: case 1192096: case 3256224: case 3192160: case 1224256: case 1160256: case 4128064: case 4064064: case 4128128: case 4064128:
                    case 5064064: case 5096064: case 5064128: case 5096128: case 5128128:
                        return GPP_ERR_UNSUPPORTED;
                    default: return GPP_ERR_BAD_ARG;
                }
            }
    }
    // ---- default heuristic (tile_hint == 0)
    if (d.C_out <= 64 || (d.C_out % 128 != 0 && d.C_out % 128 <= 64)) return launch<DT, 128, 64, 2, 2, 2, false>(d, st);
    if constexpr (!kF32Storage<DT>) {
        // 256x256 tile (8 wavefronts, 1 workgroup / CU) halves the L2 -> LDS traffic per FLOP; it pays
        // when there are enough tiles for two rounds over the 256 CUs and enough K-steps to amortise
        // its longer prologue/epilogue
        int64_t rows = 0;
        for (int g = 0; g < d.n_groups; ++g) rows += (int64_t)d.batch * d.groups[g].H_out * d.groups[g].W_out;
        const int nk = d.KH * d.KW * (d.C_in / 64);
        const int n256 = (d.C_out + 255) / 256;
        const int64_t big_blocks = ((rows + 255) / 256) * n256;
        const bool n_fits = d.C_out >= 256 && n256 * 256 * 7 <= d.C_out * 8;      // at most 1/8 of the N tiles is padding
        if (n_fits && big_blocks >= 512 && nk >= 8 && d.split_k <= 1) return launch<DT, 256, 256, 2, 4, 2, true>(d, st);
    }
    return launch<DT, 128, 128, 2, 2, 2, false>(d, st);
}

template <int DT, int BM, int CMID>
int launch_tail(gpp_conv_desc& d1, gpp_conv_desc& d2, hipStream_t st)
{
    constexpr int lds = 2 * (BM + CMID) * kRowBytes;
    static DeviceOnce once;
    auto kernel = bottleneck_tail_kernel<DT, BM, CMID>;
    int rc = once.configure(kernel, lds);
    if (rc != GPP_OK) return rc;
    const gpp_conv_group& G = d1.groups[0];
    const int64_t in_elems = G.in_off + (int64_t)(d1.batch - 1) * G.in_bstride + ((int64_t)G.H_in * G.W_in - 1) * d1.in_pitch + d1.C_in;
    const int64_t w1_bytes = (int64_t)d1.weight_rows * 9 * CMID * 2, w2_bytes = (int64_t)d2.weight_rows * CMID * 2;
    if (G.in_off < 0 || G.in_bstride < 0 || in_elems * 2 >= (1LL << 31) || w1_bytes >= (1LL << 31) || w2_bytes >= (1LL << 31))
        return GPP_ERR_UNSUPPORTED;
    d1.in_bytes = (int32_t)(in_elems * 2);
    d1.weight_bytes = (int32_t)w1_bytes;
    d2.weight_bytes = (int32_t)w2_bytes;
    const int64_t rows = (int64_t)d1.batch * G.H_out * G.W_out;
    kernel<<<dim3((unsigned)((rows + BM - 1) / BM)), dim3(256), lds, st>>>(d1, d2);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

template <int DT, int BM, int CMID>
int launch_tail_x3(gpp_conv_desc& d1, gpp_conv_desc& d2, hipStream_t st)
{
    constexpr int KC = CMID / 32;
    constexpr int ring = 2 * (BM + CMID) * kRowBytes, phase2 = KC * BM * kRowBytes + KC * 128 * kRowBytes;
    constexpr int lds = ring > phase2 ? ring : phase2;
    static DeviceOnce once;
    auto kernel = bottleneck_tail_x3_kernel<DT, BM, CMID>;
    int rc = once.configure(kernel, lds);
    if (rc != GPP_OK) return rc;
    const gpp_conv_group& G = d1.groups[0];
    const int64_t in_elems = G.in_off + (int64_t)(d1.batch - 1) * G.in_bstride + ((int64_t)G.H_in * G.W_in - 1) * d1.in_pitch + d1.C_in;
    const int64_t w1_bytes = (int64_t)d1.weight_rows * 9 * CMID * 4, w2_bytes = (int64_t)d2.weight_rows * CMID * 4;
    if (G.in_off < 0 || G.in_bstride < 0 || in_elems * 4 >= (1LL << 31) || w1_bytes >= (1LL << 31) || w2_bytes >= (1LL << 31))
        return GPP_ERR_UNSUPPORTED;
    d1.in_bytes = (int32_t)(in_elems * 4);
    d1.weight_bytes = (int32_t)w1_bytes;
    d2.weight_bytes = (int32_t)w2_bytes;
    const int64_t rows = (int64_t)d1.batch * G.H_out * G.W_out;
    kernel<<<dim3((unsigned)((rows + BM - 1) / BM)), dim3(256), lds, st>>>(d1, d2);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

// x3 types: C = 64 only (at C = 128 the intermediate tile and a W2 tile take 128 KB of LDS: one workgroup per CU)
template <int DT>
int dispatch_tail_x3(gpp_conv_desc& d1, gpp_conv_desc& d2, int tile_rows, hipStream_t st)
{
    if (d1.C_in != 64) return GPP_ERR_UNSUPPORTED;
    switch (tile_rows) {
        case 64: return launch_tail_x3<DT, 64, 64>(d1, d2, st);        // 48 KB of LDS, <= 168 registers: three workgroups per CU
        case 96: return launch_tail_x3<DT, 96, 64>(d1, d2, st);
        case 0:
        case 128: return launch_tail_x3<DT, 128, 64>(d1, d2, st);
        case 160: return launch_tail_x3<DT, 160, 64>(d1, d2, st);
        default: return GPP_ERR_BAD_ARG;
    }
}

template <int DT>
int dispatch_tail(gpp_conv_desc& d1, gpp_conv_desc& d2, int tile_rows, hipStream_t st)
{
    const bool c64 = d1.C_in == 64;
    switch (tile_rows) {
        case 96: return c64 ? launch_tail<DT, 96, 64>(d1, d2, st) : launch_tail<DT, 96, 128>(d1, d2, st);
        case 0:
        case 128: return c64 ? launch_tail<DT, 128, 64>(d1, d2, st) : launch_tail<DT, 128, 128>(d1, d2, st);
        case 160: return c64 ? launch_tail<DT, 160, 64>(d1, d2, st) : launch_tail<DT, 160, 128>(d1, d2, st);
        default: return GPP_ERR_BAD_ARG;
    }
}

}  // namespace

#endif  // GPP_CONV_IGEMM_IMPL_H_
