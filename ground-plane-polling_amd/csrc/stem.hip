// ResNet stem on gfx950: 7x7 stride-2 convolution of the 3-channel float32 input image with the
// frozen BatchNormalization folded in, ReLU, 16-bit NHWC output; and the 3x3 stride-2 'same'
// max-pool that follows it.
//
// Replaces (third-party keras_resnet, instantiated at
// /root/reference/keras_retinanet_3D/models/resnet.py:88-93):
//   ZeroPadding2D(3) -> Conv2D(64, 7x7, stride 2, valid, no bias) 'conv1' -> BatchNormalization
//   (eps 1e-5, frozen) 'bn_conv1' -> ReLU -> MaxPooling2D(3x3, stride 2, 'same') 'pool1'
//
// K = 7*7*3 = 147 is too thin for the 64-channel implicit-GEMM path; this version keeps the
// contraction on the vector ALUs in float32: a workgroup owns a 4-row x 64-column tile of
// output pixels, stages the (13 x 133 x 3) input patch in LDS once, and every lane computes
// all 64 output channels of one pixel with the weights broadcast from scalar registers.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpp.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int TW = 64, TH = 4;                 // output tile
constexpr int PW = TW * 2 + 5, PH = TH * 2 + 5;  // input patch
constexpr int PPITCH = PW * 3 + 1;             // floats per patch row (odd: spreads LDS banks)

template <typename scalar, typename vec8>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, scalar* __restrict__ out,
                                                   int H, int W, int Ho, int Wo)
{
    __shared__ float patch[PH * PPITCH];
    const int tiles_x = (Wo + TW - 1) / TW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
    const int ox0 = tx * TW, oy0 = ty * TH;
    const int ix0 = ox0 * 2 - 3, iy0 = oy0 * 2 - 3;
    const float* img = in + (size_t)b * H * W * 3;
    for (int e = threadIdx.x; e < PH * PW * 3; e += 256) {
        const int r = e / (PW * 3), c = e - r * (PW * 3);
        const int iy = iy0 + r, ix = ix0 + c / 3;
        float v = 0.0f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = img[((size_t)iy * W + ix) * 3 + (c % 3)];
        patch[r * PPITCH + c] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int ox = ox0 + lx, oy = oy0 + ly;
    const float* p0 = patch + (ly * 2) * PPITCH + lx * 6;
    scalar* dst = out + (((size_t)b * Ho + oy) * Wo + ox) * 64;
#pragma unroll 1
    for (int cg = 0; cg < 4; ++cg) {
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
        const float* wg = w + cg * 16;          // w laid out [147][64]
        for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
            for (int kc = 0; kc < 21; ++kc) {
                const float x = p0[kh * PPITCH + kc];
                const float* wk = wg + (kh * 21 + kc) * 64;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(x, wk[c], acc[c]);
            }
        }
        if (ox < Wo && oy < Ho) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                vec8 v;
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = (scalar)fmaxf(acc[h * 8 + c] + bias[cg * 16 + h * 8 + c], 0.0f);
                *(vec8*)(dst + cg * 16 + h * 8) = v;
            }
        }
    }
}

// 3x3 stride-2 max-pool, TF 'same' (pad_before = pad_total / 2, padding never wins)
template <typename scalar, typename vec8>
__global__ __launch_bounds__(256) void maxpool_kernel(const scalar* __restrict__ in, scalar* __restrict__ out,
                                                      int B, int H, int W, int C, int Ho, int Wo, int pt, int pl)
{
    const int cv = C / 8;
    const int64_t total = (int64_t)B * Ho * Wo * cv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(e % cv);
        int64_t p = e / cv;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float m[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) m[c] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - pt + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - pl + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const vec8 v = *(const vec8*)(in + (((size_t)b * H + iy) * W + ix) * C + c8 * 8);
#pragma unroll
                for (int c = 0; c < 8; ++c) m[c] = fmaxf(m[c], (float)v[c]);
            }
        }
        vec8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (scalar)m[c];
        *(vec8*)(out + (((size_t)b * Ho + oy) * Wo + ox) * C + c8 * 8) = o;
    }
}

template <typename scalar, typename vec8>
__global__ __launch_bounds__(256) void relu_kernel(const scalar* __restrict__ in, int64_t in_bs, scalar* __restrict__ out,
                                                   int64_t out_bs, int64_t n8)
{
    const scalar* src = in + (int64_t)blockIdx.y * in_bs;
    scalar* dst = out + (int64_t)blockIdx.y * out_bs;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        vec8 v = *(const vec8*)(src + e * 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (scalar)fmaxf((float)v[c], 0.0f);
        *(vec8*)(dst + e * 8) = v;
    }
}

inline int result()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

}  // namespace

extern "C" int gpp_stem_conv7x7_bn_relu(const float* in, const float* weight, const float* bias, void* out, int dtype,
                                        int B, int H, int W, void* stream)
{
    if (!in || !weight || !bias || !out || B <= 0 || H <= 0 || W <= 0) return GPP_ERR_BAD_ARG;
    if (((uintptr_t)out) & 15) return GPP_ERR_ALIGN;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    dim3 grid((unsigned)(((Wo + TW - 1) / TW) * ((Ho + TH - 1) / TH)), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GPP_BF16)
        stem_kernel<__bf16, bf16x8><<<grid, 256, 0, st>>>(in, weight, bias, (__bf16*)out, H, W, Ho, Wo);
    else if (dtype == GPP_F16)
        stem_kernel<_Float16, f16x8><<<grid, 256, 0, st>>>(in, weight, bias, (_Float16*)out, H, W, Ho, Wo);
    else
        return GPP_ERR_UNSUPPORTED;
    return result();
}

extern "C" int gpp_maxpool3x3s2_same(const void* in, void* out, int dtype, int B, int H, int W, int C, void* stream)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0) return GPP_ERR_BAD_ARG;
    if (((uintptr_t)in | (uintptr_t)out) & 15) return GPP_ERR_ALIGN;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pt = ((Ho - 1) * 2 + 3 - H > 0 ? (Ho - 1) * 2 + 3 - H : 0) / 2;
    const int pl = ((Wo - 1) * 2 + 3 - W > 0 ? (Wo - 1) * 2 + 3 - W : 0) / 2;
    const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GPP_BF16)
        maxpool_kernel<__bf16, bf16x8><<<blocks, 256, 0, st>>>((const __bf16*)in, (__bf16*)out, B, H, W, C, Ho, Wo, pt, pl);
    else if (dtype == GPP_F16)
        maxpool_kernel<_Float16, f16x8><<<blocks, 256, 0, st>>>((const _Float16*)in, (_Float16*)out, B, H, W, C, Ho, Wo, pt, pl);
    else
        return GPP_ERR_UNSUPPORTED;
    return result();
}

extern "C" int gpp_relu_strided(const void* in, int64_t in_bstride, void* out, int64_t out_bstride, int dtype, int B,
                                int64_t count, void* stream)
{
    if (!in || !out || B <= 0 || count <= 0 || count % 8 != 0 || in_bstride % 8 != 0 || out_bstride % 8 != 0)
        return GPP_ERR_BAD_ARG;
    if (((uintptr_t)in | (uintptr_t)out) & 15) return GPP_ERR_ALIGN;
    const int64_t n8 = count / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GPP_BF16)
        relu_kernel<__bf16, bf16x8><<<grid, 256, 0, st>>>((const __bf16*)in, in_bstride, (__bf16*)out, out_bstride, n8);
    else if (dtype == GPP_F16)
        relu_kernel<_Float16, f16x8><<<grid, 256, 0, st>>>((const _Float16*)in, in_bstride, (_Float16*)out, out_bstride, n8);
    else
        return GPP_ERR_UNSUPPORTED;
    return result();
}

extern "C" int gpp_relu(const void* in, void* out, int dtype, int64_t count, void* stream)
{
    return gpp_relu_strided(in, 0, out, 0, dtype, 1, count, stream);
}
