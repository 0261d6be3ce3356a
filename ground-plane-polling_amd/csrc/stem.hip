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
#include <stdlib.h>
#include <stdint.h>

#include <atomic>

#include "gpp.h"
#include "conv_igemm_types.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) float f32x8_natural;
typedef f32x8_natural f32x8 __attribute__((aligned(16)));     // 8 floats, accessed as two 16-byte halves (GPP_F32 maps)

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

// ---- MFMA version ---------------------------------------------------------------------------------
// GEMM view per output row segment: D[n][px] = sum_k W[n][k] * X[k][px], k = (kh, kw*3 + c) with the 21
// taps of one kernel row padded to 32 (weights zero there), i.e. K = 7 * 32 = 224.  The pixel operand
// of output pixel ox for kernel row kh is the 32 CONSECUTIVE f16 values that start at element 6*ox of
// input row 2*oy + kh - 3 in the staged patch (stride-2 conv over 3 interleaved channels): no im2col.
// Input and weights are rounded to f16 (11-bit significand; |x| <= 152 so the step is <= 2^-4) and
// accumulated in float32 by v_mfma_f32_16x16x32_f16, whatever the activation type of the network.
// Weights [64 rows interleaved as for conv_igemm][7][32] f16 stay in LDS for the life of the
// (persistent) workgroup; each wavefront owns one output row of 64 pixels x 64 channels.
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// diagnostic build only (-DGPP_STAMPS): per-tile phase stamps of the first 64 workgroups (tools/stem_time.py)
#ifdef GPP_STAMPS
__device__ unsigned long long* g_stem_stamps = nullptr;
#define STEM_STAMP(j)                                                                                     \
    do {                                                                                                  \
        if (g_stem_stamps && blockIdx.x < 64 && iter < 16 && tid == 0)                                    \
            g_stem_stamps[(blockIdx.x * 16 + iter) * 8 + (j)] = __builtin_amdgcn_s_memrealtime();        \
    } while (0)
#else
#define STEM_STAMP(j) do { } while (0)
#endif

constexpr int MW_PITCH = 232;                   // halfs per weight row in LDS (224 + 8: conflict-free b128 reads)
constexpr int MP_PX = TW * 2 + 9;               // patch pixels per row: 2*63 + 32/3 rounded up
constexpr int MP_PITCH = 416;                   // halfs per patch row (>= 3 * MP_PX = 411)
constexpr int MP_ROWS = TH * 2 + 5;

template <typename scalar, typename vec8>
// (256, 2): without the second bound the compiler parks 96 values in AGPRs, 264 registers per lane in all, and only ONE
// workgroup fits a CU -- the kernel then ran its 512 persistent workgroups as two rounds (80 us instead of 62)
__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(const float* __restrict__ in, const _Float16* __restrict__ w,
                                                        const float* __restrict__ bias, scalar* __restrict__ out,
                                                        int B, int H, int W, int Ho, int Wo)
{
    __shared__ __attribute__((aligned(16))) _Float16 s_w[64 * MW_PITCH];
    __shared__ __attribute__((aligned(16))) _Float16 s_p[MP_ROWS * MP_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 64 * MW_PITCH / 8; e += 256) ((uint4*)s_w)[e] = ((const uint4*)w)[e];
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y * B;
    const int frow = lane & 15, fq = lane >> 4;
    float bias_v[2][8];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_v[jj][e] = bias[jj * 32 + fq * 8 + e];

    // The input patch of a tile (MP_ROWS x MP_PITCH halfs) is fetched by all 256 threads, PATCH_IT pairs of
    // floats each.  All loads of a tile are issued back to back into registers (one latency, not PATCH_IT of
    // them), and they are issued for the NEXT tile before the MFMA work of the current one, so the fetch
    // runs under the matrix work and the output stores.
    constexpr int PATCH_PAIRS = MP_ROWS * (MP_PITCH / 2);
    constexpr int PATCH_IT = (PATCH_PAIRS + 255) / 256;
    float patch[PATCH_IT][2];          // raw floats: converting at load time would make the loads blocking
    auto load_patch = [&](int t) {
        const int b = t / (tiles_x * tiles_y), r = t - b * (tiles_x * tiles_y);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int ix0 = tx * TW * 2 - 3, iy0 = ty * TH * 2 - 3;
        const float* img = in + (size_t)b * H * W * 3;
#pragma unroll
        for (int it = 0; it < PATCH_IT; ++it) {
            const int e = tid + it * 256;
            const int pr = e / (MP_PITCH / 2), c2 = (e - pr * (MP_PITCH / 2)) * 2;
            const int iy = iy0 + pr;
            const int x0 = ix0 * 3 + c2;                     // element index inside the image row
            float v0 = 0.0f, v1 = 0.0f;
            if (e < PATCH_PAIRS && (unsigned)iy < (unsigned)H) {
                const float* rowp = img + (size_t)iy * W * 3;
                if (x0 >= 0 && x0 < W * 3) v0 = rowp[x0];
                if (x0 + 1 >= 0 && x0 + 1 < W * 3) v1 = rowp[x0 + 1];
            }
            patch[it][0] = v0;
            patch[it][1] = v1;
        }
    };
    if ((int)blockIdx.x < tiles) load_patch(blockIdx.x);
    int iter = 0;
    (void)iter;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x, ++iter) {
        const int b = t / (tiles_x * tiles_y), r = t - b * (tiles_x * tiles_y);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int ox0 = tx * TW, oy0 = ty * TH;
        STEM_STAMP(0);
        __syncthreads();                                     // previous tile's readers are done with s_p
        STEM_STAMP(1);
#pragma unroll
        for (int it = 0; it < PATCH_IT; ++it) {
            const int e = tid + it * 256;
            const int pr = e / (MP_PITCH / 2), c2 = (e - pr * (MP_PITCH / 2)) * 2;
            if (e < PATCH_PAIRS) *(f16x2*)(s_p + pr * MP_PITCH + c2) = (f16x2){(_Float16)patch[it][0], (_Float16)patch[it][1]};
        }
        __syncthreads();
        STEM_STAMP(2);
        if (t + (int)gridDim.x < tiles) load_patch(t + gridDim.x);
        STEM_STAMP(3);
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            f16x8 wf[4], xf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *(const f16x8*)(s_w + (j * 16 + frow) * MW_PITCH + kh * 32 + fq * 8);
            const _Float16* prow = s_p + (wave * 2 + kh) * MP_PITCH + fq * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x2* src = (const f16x2*)(prow + (i * 16 + frow) * 6);
                const f16x2 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
                xf[i] = (f16x8){p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        STEM_STAMP(4);
        const int oy = oy0 + wave;
        if (oy < Ho) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ox = ox0 + i * 16 + frow;
                if (ox >= Wo) continue;
                scalar* dst = out + (((size_t)b * Ho + oy) * Wo + ox) * 64;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    vec8 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (scalar)fmaxf(acc[i][2 * jj][e] + bias_v[jj][e], 0.0f);
                        v[4 + e] = (scalar)fmaxf(acc[i][2 * jj + 1][e] + bias_v[jj][4 + e], 0.0f);
                    }
                    *(vec8*)(dst + jj * 32 + fq * 8) = v;
                }
            }
        }
        STEM_STAMP(5);
    }
}

// ---- MFMA stem for the float32-storage "x3" types (GPP_F16X3 / GPP_BF16X3 models) ------------------------------------
// The same GEMM view as stem_mfma_kernel, at (almost) float32 precision: input pixels and weights are each split into two IEEE
// halves (hi = f16(v), lo = f16(v - hi): 22 significant bits; |x| <= 152 and the weights carry a per-channel power of two, so both
// halves are normal halfs) and every product is three matrix products, hi*whi + hi*wlo + lo*whi, accumulated in float32; the
// output is float32.  Replaces the float32 fmaf stem of rounds 1-2 for these types: that kernel ran at a fifth of the vector peak
// and lost another third of its speed when the packed-FP32 instructions went (372 us at B = 8; this one: see HISTORY.md 4.9).
// ROWS wavefronts per workgroup, each owning one output row of 64 pixels x 64 channels of a ROWS x 64 tile; persistent workgroups.
// Packed weights: [whi 64 x 232 halfs][wlo 64 x 232 halfs][64 float32 out_scale] (gpp_stem_pack_weights_f16x3).
template <int ROWS>
__global__ __launch_bounds__(64 * ROWS) void stem_mfma_x3_kernel(const float* __restrict__ in, const _Float16* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int B, int H, int W, int Ho, int Wo, unsigned long long* range_events)
{
    constexpr int NT = 64 * ROWS, PROWS = ROWS * 2 + 5;
    constexpr int W_HALFS = 64 * MW_PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    _Float16* s_wh = (_Float16*)xsm;
    _Float16* s_wl = s_wh + W_HALFS;
    _Float16* s_ph = s_wl + W_HALFS;
    _Float16* s_pl = s_ph + PROWS * MP_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 2 * W_HALFS / 8; e += NT) ((uint4*)s_wh)[e] = ((const uint4*)w)[e];
    const float* scale = (const float*)(w + 2 * W_HALFS);
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + ROWS - 1) / ROWS;
    const int tiles = tiles_x * tiles_y * B;
    const int frow = lane & 15, fq = lane >> 4;
    float bias_v[2][8], scale_v[2][8];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 8; ++e) { bias_v[jj][e] = bias[jj * 32 + fq * 8 + e]; scale_v[jj][e] = scale[jj * 32 + fq * 8 + e]; }

    constexpr int PATCH_PAIRS = PROWS * (MP_PITCH / 2);
    constexpr int PATCH_IT = (PATCH_PAIRS + NT - 1) / NT;
    float patch[PATCH_IT][2];
    auto load_patch = [&](int t) {
        const int b = t / (tiles_x * tiles_y), r = t - b * (tiles_x * tiles_y);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int ix0 = tx * TW * 2 - 3, iy0 = ty * ROWS * 2 - 3;
        const float* img = in + (size_t)b * H * W * 3;
#pragma unroll
        for (int it = 0; it < PATCH_IT; ++it) {
            const int e = tid + it * NT;
            const int pr = e / (MP_PITCH / 2), c2 = (e - pr * (MP_PITCH / 2)) * 2;
            const int iy = iy0 + pr;
            const int x0 = ix0 * 3 + c2;
            float v0 = 0.0f, v1 = 0.0f;
            if (e < PATCH_PAIRS && (unsigned)iy < (unsigned)H) {
                const float* rowp = img + (size_t)iy * W * 3;
                if (x0 >= 0 && x0 < W * 3) v0 = rowp[x0];
                if (x0 + 1 >= 0 && x0 + 1 < W * 3) v1 = rowp[x0 + 1];
            }
            patch[it][0] = v0;
            patch[it][1] = v1;
        }
    };
    if ((int)blockIdx.x < tiles) load_patch(blockIdx.x);
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int b = t / (tiles_x * tiles_y), r = t - b * (tiles_x * tiles_y);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int ox0 = tx * TW, oy0 = ty * ROWS;
        __syncthreads();                                     // previous tile's readers are done with the patch
#pragma unroll
        for (int it = 0; it < PATCH_IT; ++it) {
            const int e = tid + it * NT;
            const int pr = e / (MP_PITCH / 2), c2 = (e - pr * (MP_PITCH / 2)) * 2;
            if (e < PATCH_PAIRS) {
                const _Float16 h0 = (_Float16)patch[it][0], h1 = (_Float16)patch[it][1];
                *(f16x2*)(s_ph + pr * MP_PITCH + c2) = (f16x2){h0, h1};
                *(f16x2*)(s_pl + pr * MP_PITCH + c2) = (f16x2){(_Float16)(patch[it][0] - (float)h0), (_Float16)(patch[it][1] - (float)h1)};
            }
        }
        __syncthreads();
        if (t + (int)gridDim.x < tiles) load_patch(t + gridDim.x);
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            f16x8 wh[4], wl[4], xh[4], xl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wh[j] = *(const f16x8*)(s_wh + (j * 16 + frow) * MW_PITCH + kh * 32 + fq * 8);
                wl[j] = *(const f16x8*)(s_wl + (j * 16 + frow) * MW_PITCH + kh * 32 + fq * 8);
            }
            const int poff = (wave * 2 + kh) * MP_PITCH + fq * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x2* sh = (const f16x2*)(s_ph + poff + (i * 16 + frow) * 6);
                const f16x2* sl = (const f16x2*)(s_pl + poff + (i * 16 + frow) * 6);
                const f16x2 a0 = sh[0], a1 = sh[1], a2 = sh[2], a3 = sh[3];
                const f16x2 c0 = sl[0], c1 = sl[1], c2 = sl[2], c3 = sl[3];
                xh[i] = (f16x8){a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
                xl[i] = (f16x8){c0[0], c0[1], c1[0], c1[1], c2[0], c2[1], c3[0], c3[1]};
            }
            // three products per accumulator; consecutive matrix instructions go to different accumulators
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
        }
        const int oy = oy0 + wave;
        if (oy < Ho) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ox = ox0 + i * 16 + frow;
                if (ox >= Wo) continue;
                float* dst = out + (((size_t)b * Ho + oy) * Wo + ox) * 64;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    f32x4 v0, v1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = fmaxf(acc[i][2 * jj][e] * scale_v[jj][e] + bias_v[jj][e], 0.0f);
                        v1[e] = fmaxf(acc[i][2 * jj + 1][e] * scale_v[jj][4 + e] + bias_v[jj][4 + e], 0.0f);
                    }
                    *(f32x4*)(dst + jj * 32 + fq * 8) = v0;
                    *(f32x4*)(dst + jj * 32 + fq * 8 + 4) = v1;
                    // GPP_F16X3 range ledger (conv_igemm_impl.h x3_range): this map is stored as float32 and split -- clamped to the half
                    // range -- by the loop of the layers that read it; a value they would alter (or a NaN) is counted here
                    bool outside = false;
#pragma unroll
                    for (int e = 0; e < 4; ++e) outside |= !(v0[e] <= 65504.0f) | !(v1[e] <= 65504.0f);
                    if (__builtin_expect(outside, 0)) atomicAdd(range_events, 1ull);
                }
            }
        }
    }
}

// ---- MFMA stem fused with pool1 -------------------------------------------------------------------
// conv1 + bn_conv1 + ReLU + the 3x3 stride-2 'same' max-pool in one launch: the (B, Ho, Wo, 64) conv map -- 137 MB at
// B = 8, 402 x 1333, written by the stem and read back by the pool -- never exists.  A persistent workgroup of 8
// wavefronts marches DOWN a strip of 64 conv columns (31 pooled columns), 8 conv rows (one per wavefront, the matrix work
// of stem_mfma_kernel) = 4 pooled rows per step.  The rounded conv rows go to a ring of 9 rows in LDS: pooled row py needs
// conv rows 2 py - pt .. 2 py - pt + 2, so the last row of a step is the first of the next and is CARRIED in the ring,
// not recomputed; a workgroup that starts in the middle of a strip computes just that one row first (a "pre-step" of one
// wavefront).  Steps are numbered (image, strip, row block) with the row block fastest and cut into gridDim.x
// contiguous, equally long ranges.  Pooling compares the stored (rounded) values, as maxpool_kernel does on the map the
// unfused stem stores: the result is bit-identical to the two launches.
// FP_ROWS = conv rows per step = wavefronts per workgroup: 8 (512 threads, 118 KB of LDS, one workgroup per CU) or 4 (256 threads,
// 79.6 KB, two per CU).
constexpr int FP_PCOLS = 31;                      // pooled columns per strip: conv columns 2j .. 2j + 2 <= 62 of the 64
constexpr int FP_ROW_BYTES = 64 * 64 * 2;         // one conv row of a strip: 64 pixels x 64 channels, 16-bit
constexpr int FP_W_BYTES = 64 * MW_PITCH * 2;
constexpr int fp_lds(int rows) { return FP_W_BYTES + (rows * 2 + 5) * MP_PITCH * 2 + (rows + 1) * FP_ROW_BYTES; }

template <typename scalar, typename vec8, int FP_ROWS>
__global__ __launch_bounds__(64 * FP_ROWS, 8 / FP_ROWS) void stem_pool_mfma_kernel(const float* __restrict__ in, const _Float16* __restrict__ w,
                                                                const float* __restrict__ bias, scalar* __restrict__ out,
                                                                int B, int H, int W, int Ho, int Wo, int Hp, int Wp, int pt, int pl)
{
    constexpr int FP_PATCH_ROWS = FP_ROWS * 2 + 5;    // input rows under the conv rows of a step
    constexpr int FP_RING = FP_ROWS + 1;              // conv rows resident in LDS
    constexpr int FP_P_BYTES = FP_PATCH_ROWS * MP_PITCH * 2;
    constexpr int NT = 64 * FP_ROWS, PR = FP_ROWS / 2;   // threads; pooled rows per step
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    _Float16* s_w = (_Float16*)fsm;
    _Float16* s_p = (_Float16*)(fsm + FP_W_BYTES);
    unsigned char* s_c = fsm + FP_W_BYTES + FP_P_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 64 * MW_PITCH / 8; e += NT) ((uint4*)s_w)[e] = ((const uint4*)w)[e];
    const int n_strips = (Wp + FP_PCOLS - 1) / FP_PCOLS, n_blocks = (Hp + PR - 1) / PR;
    const int64_t total = (int64_t)B * n_strips * n_blocks;
    const int lo = (int)(total * blockIdx.x / gridDim.x), hi = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int frow = lane & 15, fq = lane >> 4;
    float bias_v[2][8];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_v[jj][e] = bias[jj * 32 + fq * 8 + e];

    constexpr int PATCH_PAIRS = FP_PATCH_ROWS * (MP_PITCH / 2);
    constexpr int PATCH_IT = (PATCH_PAIRS + NT - 1) / NT;
    float patch[PATCH_IT][2];
    // step idx = ((b * n_strips) + s) * n_blocks + t; a pre-step of step (b, s, t) is row block t - 1, last wavefront only
    auto load_patch = [&](int idx, bool pre) {
        const int t = idx % n_blocks, bs = idx / n_blocks;
        const int sidx = bs % n_strips, b = bs / n_strips;
        const int tt = pre ? t - 1 : t;
        const int iy0 = (FP_ROWS * tt - pt + 1) * 2 - 3, ix0 = (2 * FP_PCOLS * sidx - pl) * 2 - 3;
        const float* img = in + (size_t)b * H * W * 3;
#pragma unroll
        for (int it = 0; it < PATCH_IT; ++it) {
            const int e = tid + it * NT;
            const int pr = e / (MP_PITCH / 2), c2 = (e - pr * (MP_PITCH / 2)) * 2;
            const int iy = iy0 + pr;
            const int x0 = ix0 * 3 + c2;
            float v0 = 0.0f, v1 = 0.0f;
            if (e < PATCH_PAIRS && (unsigned)iy < (unsigned)H) {
                const float* rowp = img + (size_t)iy * W * 3;
                if (x0 >= 0 && x0 < W * 3) v0 = rowp[x0];
                if (x0 + 1 >= 0 && x0 + 1 < W * 3) v1 = rowp[x0 + 1];
            }
            patch[it][0] = v0;
            patch[it][1] = v1;
        }
    };
    // the carried row of the first step of a range comes from nobody: compute it, unless it is the padding row above the map
    auto needs_pre = [&](int idx, bool first) { return (idx % n_blocks) == 0 ? (pt == 0) : first; };
    int idx = lo;
    bool pre = lo < hi && needs_pre(lo, true);
    if (lo < hi) load_patch(idx, pre);
    while (idx < hi) {
        const int t = idx % n_blocks, bs = idx / n_blocks;
        const int sidx = bs % n_strips, b = bs / n_strips;
        const int tt = pre ? t - 1 : t;
        __syncthreads();                                     // the previous step's readers are done with s_p and the ring
#pragma unroll
        for (int it = 0; it < PATCH_IT; ++it) {
            const int e = tid + it * NT;
            const int pr = e / (MP_PITCH / 2), c2 = (e - pr * (MP_PITCH / 2)) * 2;
            if (e < PATCH_PAIRS) *(f16x2*)(s_p + pr * MP_PITCH + c2) = (f16x2){(_Float16)patch[it][0], (_Float16)patch[it][1]};
        }
        __syncthreads();
        const int nidx = pre ? idx : idx + 1;
        const bool npre = pre ? false : (nidx < hi && needs_pre(nidx, false));
        if (nidx < hi) load_patch(nidx, npre);
        if (!pre || wave == FP_ROWS - 1) {
            f32x4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                f16x8 wf[4], xf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[j] = *(const f16x8*)(s_w + (j * 16 + frow) * MW_PITCH + kh * 32 + fq * 8);
                const _Float16* prow = s_p + (wave * 2 + kh) * MP_PITCH + fq * 8;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f16x2* src = (const f16x2*)(prow + (i * 16 + frow) * 6);
                    const f16x2 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
                    xf[i] = (f16x8){p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
            // conv row FP_ROWS tt - pt + 1 + wave -> ring slot (row + pt) mod FP_RING; 16-byte chunks XOR-swizzled by the pixel
            unsigned char* crow = s_c + ((FP_ROWS * tt + 1 + wave + FP_RING) % FP_RING) * FP_ROW_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = i * 16 + frow;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    vec8 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (scalar)fmaxf(acc[i][2 * jj][e] + bias_v[jj][e], 0.0f);
                        v[4 + e] = (scalar)fmaxf(acc[i][2 * jj + 1][e] + bias_v[jj][4 + e], 0.0f);
                    }
                    *(vec8*)(crow + px * 128 + (((jj * 4 + fq) ^ (px & 7)) << 4)) = v;
                }
            }
        }
        if (!pre) {
            __syncthreads();
            const int c0 = 2 * FP_PCOLS * sidx - pl;
            for (int item = tid; item < PR * FP_PCOLS * 8; item += NT) {
                const int c8 = item & 7, q = item >> 3;
                const int k = q / FP_PCOLS, j = q - k * FP_PCOLS;
                const int py = PR * t + k, px = FP_PCOLS * sidx + j;
                if (py >= Hp || px >= Wp) continue;
                float m[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) m[c] = -INFINITY;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = FP_ROWS * t - pt + 2 * k + dy;
                    if ((unsigned)r >= (unsigned)Ho) continue;
                    const unsigned char* crow = s_c + ((FP_ROWS * t + 2 * k + dy) % FP_RING) * FP_ROW_BYTES;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int cr = 2 * j + dx;
                        if ((unsigned)(c0 + cr) >= (unsigned)Wo) continue;
                        const vec8 v = *(const vec8*)(crow + cr * 128 + ((c8 ^ (cr & 7)) << 4));
#pragma unroll
                        for (int c = 0; c < 8; ++c) m[c] = fmaxf(m[c], (float)v[c]);
                    }
                }
                vec8 o;
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = (scalar)m[c];
                *(vec8*)(out + (((size_t)b * Hp + py) * Wp + px) * 64 + c8 * 8) = o;
            }
        }
        idx = nidx;
        pre = npre;
    }
}

// ---- x3 stem fused with pool1 (round 6) ------------------------------------------------------------
// conv1 + bn_conv1 + ReLU + pool1 of the float32-storage x3 types in one launch: stem_mfma_x3_kernel's matrix work on the steps
// of stem_pool_mfma_kernel (a persistent workgroup marches down a strip of 64 conv columns, one conv row per wavefront, the last row of a
// step carried to the next).  A ring of float32 conv rows does not fit beside 58 KB of hi / lo weights (9 x 16 KB); what the ring holds
// here is each conv row AFTER the horizontal half of the pool -- max over conv columns 2j .. 2j + 2, taken in registers with row shifts
// of the accumulator lanes (a lane holds one pixel of a 16-pixel fragment; pixel 14's third column comes from the next fragment) --
// i.e. 31 pooled columns x 64 channels x 4 bytes = 7.75 KB per row.  The vertical half reads three ring rows per pooled pixel.
// max is exact and has no order, so the result is bit-identical to gpp_stem_conv7x7_bn_relu_x3 + gpp_maxpool3x3s2_same(GPP_F32), and a
// conv value beyond the half range is counted exactly as there: once per (pixel, 32-channel group) of the conv map, by the workgroup
// that OWNS the pixel (strips overlap by two conv columns, a range's first carried row is computed twice).
// The (B, Ho, Wo, 64) float32 conv map -- 274 MB at B = 8, 402 x 1333, written and read back -- never exists.
// FP_ROWS = conv rows per step = wavefronts: 6 (142 KB of LDS) or 4 (119 KB); one workgroup per CU either way.
constexpr int XP_ROW_BYTES = 32 * 64 * 4;         // one half-pooled conv row of a strip: 32 (31 used) columns x 64 channels float32
constexpr int xp_lds(int rows) { return 2 * FP_W_BYTES + 2 * (rows * 2 + 5) * MP_PITCH * 2 + (rows + 1) * XP_ROW_BYTES + 128 * 4; }

// lane l of a row of 16 receives the value of lane l + n (row_shl) / l - n (row_shr); lanes whose source is outside the row keep `old`
template <int CTRL>
__device__ __forceinline__ float dpp_row(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

template <int FP_ROWS>
__global__ __launch_bounds__(64 * FP_ROWS) void stem_pool_mfma_x3_kernel(const float* __restrict__ in, const _Float16* __restrict__ w,
                                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                                      int B, int H, int W, int Ho, int Wo, int Hp, int Wp, int pt, int pl,
                                                                      unsigned long long* range_events)
{
    constexpr int PROWS = FP_ROWS * 2 + 5, RING = FP_ROWS + 1, NT = 64 * FP_ROWS, PR = FP_ROWS / 2;
    constexpr int W_HALFS = 64 * MW_PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char qsm[];
    _Float16* s_wh = (_Float16*)qsm;
    _Float16* s_wl = s_wh + W_HALFS;
    _Float16* s_ph = s_wl + W_HALFS;
    _Float16* s_pl = s_ph + PROWS * MP_PITCH;
    unsigned char* s_c = (unsigned char*)(s_pl + PROWS * MP_PITCH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 2 * W_HALFS / 8; e += NT) ((uint4*)s_wh)[e] = ((const uint4*)w)[e];
    // out_scale and bias of the 64 channels: read from LDS in the epilogue (held in registers across the matrix loop they cost 32 VGPRs -- spills at 6 wavefronts)
    float* s_sb = (float*)(s_c + RING * XP_ROW_BYTES);
    if (tid < 64) { s_sb[tid] = ((const float*)(w + 2 * W_HALFS))[tid]; s_sb[64 + tid] = bias[tid]; }
    const int n_strips = (Wp + FP_PCOLS - 1) / FP_PCOLS, n_blocks = (Hp + PR - 1) / PR;
    const int64_t total = (int64_t)B * n_strips * n_blocks;
    const int lo = (int)(total * blockIdx.x / gridDim.x), hi = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int frow = lane & 15, fq = lane >> 4;
    constexpr int PATCH_PAIRS = PROWS * (MP_PITCH / 2);
    constexpr int PATCH_IT = (PATCH_PAIRS + NT - 1) / NT;
    float patch[PATCH_IT][2];
    // step idx = ((b * n_strips) + s) * n_blocks + t; a pre-step of step (b, s, t) is row block t - 1, last wavefront only
    auto load_patch = [&](int idx, bool pre) {
        const int t = idx % n_blocks, bs = idx / n_blocks;
        const int sidx = bs % n_strips, b = bs / n_strips;
        const int tt = pre ? t - 1 : t;
        const int iy0 = (FP_ROWS * tt - pt + 1) * 2 - 3, ix0 = (2 * FP_PCOLS * sidx - pl) * 2 - 3;
        const float* img = in + (size_t)b * H * W * 3;
#pragma unroll
        for (int it = 0; it < PATCH_IT; ++it) {
            const int e = tid + it * NT;
            const int pr = e / (MP_PITCH / 2), c2 = (e - pr * (MP_PITCH / 2)) * 2;
            const int iy = iy0 + pr;
            const int x0 = ix0 * 3 + c2;
            float v0 = 0.0f, v1 = 0.0f;
            if (e < PATCH_PAIRS && (unsigned)iy < (unsigned)H) {
                const float* rowp = img + (size_t)iy * W * 3;
                if (x0 >= 0 && x0 < W * 3) v0 = rowp[x0];
                if (x0 + 1 >= 0 && x0 + 1 < W * 3) v1 = rowp[x0 + 1];
            }
            patch[it][0] = v0;
            patch[it][1] = v1;
        }
    };
    // the carried row of the first step of a range comes from nobody: compute it, unless it is the padding row above the map
    auto needs_pre = [&](int idx, bool first) { return (idx % n_blocks) == 0 ? (pt == 0) : first; };
    int idx = lo;
    bool pre = lo < hi && needs_pre(lo, true);
    if (lo < hi) load_patch(idx, pre);
    while (idx < hi) {
        const int t = idx % n_blocks, bs = idx / n_blocks;
        const int sidx = bs % n_strips, b = bs / n_strips;
        const int tt = pre ? t - 1 : t;
        __syncthreads();                                     // the previous step's readers are done with the patch and the ring
#pragma unroll
        for (int it = 0; it < PATCH_IT; ++it) {
            const int e = tid + it * NT;
            const int pr = e / (MP_PITCH / 2), c2 = (e - pr * (MP_PITCH / 2)) * 2;
            if (e < PATCH_PAIRS) {
                const _Float16 h0 = (_Float16)patch[it][0], h1 = (_Float16)patch[it][1];
                *(f16x2*)(s_ph + pr * MP_PITCH + c2) = (f16x2){h0, h1};
                *(f16x2*)(s_pl + pr * MP_PITCH + c2) = (f16x2){(_Float16)(patch[it][0] - (float)h0), (_Float16)(patch[it][1] - (float)h1)};
            }
        }
        __syncthreads();
        const int nidx = pre ? idx : idx + 1;
        const bool npre = pre ? false : (nidx < hi && needs_pre(nidx, false));
        if (nidx < hi) load_patch(nidx, npre);
        if (!pre || wave == FP_ROWS - 1) {
            f32x4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                f16x8 wh[4], wl[4], xh[4], xl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wh[j] = *(const f16x8*)(s_wh + (j * 16 + frow) * MW_PITCH + kh * 32 + fq * 8);
                    wl[j] = *(const f16x8*)(s_wl + (j * 16 + frow) * MW_PITCH + kh * 32 + fq * 8);
                }
                const int poff = (wave * 2 + kh) * MP_PITCH + fq * 8;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f16x2* sh = (const f16x2*)(s_ph + poff + (i * 16 + frow) * 6);
                    const f16x2* sl = (const f16x2*)(s_pl + poff + (i * 16 + frow) * 6);
                    const f16x2 a0 = sh[0], a1 = sh[1], a2 = sh[2], a3 = sh[3];
                    const f16x2 c0 = sl[0], c1 = sl[1], c2 = sl[2], c3 = sl[3];
                    xh[i] = (f16x8){a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
                    xl[i] = (f16x8){c0[0], c0[1], c1[0], c1[1], c2[0], c2[1], c3[0], c3[1]};
                }
                // the product order of stem_mfma_x3_kernel (the accumulation order is part of the result)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
            }
            // conv row FP_ROWS tt - pt + 1 + wave: the stored values of the unfused stem, in place of the accumulators
            const int crow_idx = FP_ROWS * tt - pt + 1 + wave;
            const int c0 = 2 * FP_PCOLS * sidx - pl;
            const bool count_row = (unsigned)crow_idx < (unsigned)Ho && (!pre || t == 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cr = i * 16 + frow;
                const bool col_ok = (unsigned)(c0 + cr) < (unsigned)Wo;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    bool outside = false;
                    const f32x4 sc0 = *(const f32x4*)(s_sb + jj * 32 + fq * 8), sc1 = *(const f32x4*)(s_sb + jj * 32 + fq * 8 + 4);
                    const f32x4 bi0 = *(const f32x4*)(s_sb + 64 + jj * 32 + fq * 8), bi1 = *(const f32x4*)(s_sb + 64 + jj * 32 + fq * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v0 = fmaxf(acc[i][2 * jj][e] * sc0[e] + bi0[e], 0.0f);
                        const float v1 = fmaxf(acc[i][2 * jj + 1][e] * sc1[e] + bi1[e], 0.0f);
                        outside |= !(v0 <= 65504.0f) | !(v1 <= 65504.0f);
                        acc[i][2 * jj][e] = col_ok ? v0 : -INFINITY;           // a column outside the conv map never wins
                        acc[i][2 * jj + 1][e] = col_ok ? v1 : -INFINITY;
                    }
                    if (__builtin_expect(outside && count_row && col_ok && cr < 2 * FP_PCOLS, 0)) atomicAdd(range_events, 1ull);
                }
            }
            // horizontal half of the pool: pooled column q = conv columns 2q, 2q + 1, 2q + 2 of the strip -> the even lanes of a fragment
            float* hrow = (float*)(s_c + ((FP_ROWS * tt + 1 + wave + RING) % RING) * XP_ROW_BYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = i * 8 + (frow >> 1);
                const bool producer = !(frow & 1) && q < FP_PCOLS;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = acc[i][j][e];
                        const float s1 = dpp_row<0x101>(a, a);                                             // lane + 1 (lane 15: unused)
                        const float nx = i < 3 ? dpp_row<0x11e>(a, acc[i < 3 ? i + 1 : 3][j][e]) : a;      // lanes 14, 15 <- lanes 0, 1 of the next fragment
                        const float s2 = dpp_row<0x102>(nx, a);                                            // lane + 2; lanes 14, 15 keep nx
                        m[e] = fmaxf(a, fmaxf(s1, s2));
                    }
                    // channels (j >> 1) * 32 + fq * 8 + (j & 1) * 4 .. + 3 = 16-byte chunk (j >> 1) * 8 + fq * 2 + (j & 1) of the pixel's 256 bytes
                    if (producer) *(f32x4*)((unsigned char*)hrow + q * 256 + (((((j >> 1) * 8 + fq * 2 + (j & 1))) ^ (q & 15)) << 4)) = m;
                }
            }
        }
        if (!pre) {
            __syncthreads();
            for (int item = tid; item < PR * FP_PCOLS * 16; item += NT) {
                const int c16 = item & 15, qq = item >> 4;
                const int k = qq / FP_PCOLS, j = qq - k * FP_PCOLS;
                const int py = PR * t + k, px = FP_PCOLS * sidx + j;
                if (py >= Hp || px >= Wp) continue;
                f32x4 m = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = FP_ROWS * t - pt + 2 * k + dy;
                    if ((unsigned)r >= (unsigned)Ho) continue;
                    const unsigned char* crow = s_c + ((FP_ROWS * t + 2 * k + dy) % RING) * XP_ROW_BYTES;
                    const f32x4 v = *(const f32x4*)(crow + j * 256 + ((c16 ^ (j & 15)) << 4));
#pragma unroll
                    for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], v[c]);
                }
                *(f32x4*)(out + (((size_t)b * Hp + py) * Wp + px) * 64 + c16 * 4) = m;
            }
        }
        idx = nidx;
        pre = npre;
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

// ReLU on a pre-split GPP_BF16X3 map: every 128 bytes are [32 bf16 hi | 32 bf16 lo] of 32 channels, value = hi + lo with
// |lo| <= ulp(hi) / 2, so the sign of the value is the sign of hi: a negative hi clears the pair, everything else stays.
__global__ __launch_bounds__(256) void relu_x3_kernel(const char* __restrict__ in, int64_t in_bs_bytes, char* __restrict__ out,
                                                      int64_t out_bs_bytes, int64_t groups)
{
    const char* src = in + (int64_t)blockIdx.y * in_bs_bytes;
    char* dst = out + (int64_t)blockIdx.y * out_bs_bytes;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        const int64_t off = (g >> 2) * 128 + (g & 3) * 16;       // 8 channels: 16 bytes of hi, their lo 64 bytes further
        bf16x8 h = *(const bf16x8*)(src + off), l = *(const bf16x8*)(src + off + 64);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bool neg = (float)h[c] < 0.0f;
            h[c] = neg ? (__bf16)0.0f : h[c];
            l[c] = neg ? (__bf16)0.0f : l[c];
        }
        *(bf16x8*)(dst + off) = h;
        *(bf16x8*)(dst + off + 64) = l;
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
    else if (dtype == GPP_F32)
        stem_kernel<float, f32x8><<<grid, 256, 0, st>>>(in, weight, bias, (float*)out, H, W, Ho, Wo);
    else
        return GPP_ERR_UNSUPPORTED;
    return result();
}

#ifdef GPP_STAMPS
extern "C" int gpp_debug_set_stem_stamps(void* buffer)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stem_stamps), &buffer, sizeof(buffer));
}
#endif

extern "C" int gpp_stem_pack_weights_f16(const float* host_weight_147x64, void* host_packed, size_t packed_bytes)
{
    // host-side helper: [147][64] float32 (HWIO flattened, BN scale folded) -> [64][232] f16, rows interleaved as for
    // gpp_conv2d_igemm (row 16h + 4q + r of a 32-group = channel 8q + 4h + r), k = kh*32 + kw*3 + c, zero padded
    if (!host_weight_147x64 || !host_packed || packed_bytes < (size_t)64 * MW_PITCH * 2) return GPP_ERR_BAD_ARG;
    _Float16* dst = (_Float16*)host_packed;
    for (int pos = 0; pos < 64; ++pos) {
        const int g = pos / 32, within = pos % 32, h = within / 16, q = (within % 16) / 4, r = within % 4;
        const int n = g * 32 + 8 * q + 4 * h + r;
        for (int k = 0; k < MW_PITCH; ++k) {
            const int kh = k / 32, kc = k % 32;
            float v = 0.0f;
            if (k < 224 && kc < 21) v = host_weight_147x64[(kh * 21 + kc) * 64 + n];
            dst[pos * MW_PITCH + k] = (_Float16)v;
        }
    }
    return GPP_OK;
}

extern "C" int gpp_stem_pack_weights_f16x3(const float* host_weight_147x64, void* host_packed, size_t packed_bytes)
{
    // host-side helper: [147][64] float32 -> [whi 64 x 232 halfs][wlo 64 x 232 halfs][64 float32 out_scale]; channel n's weights are
    // multiplied by 2^k(n) (largest weight of the channel in [2^13, 2^14)) before they are split into two halves, out_scale[n] =
    // 2^-k(n); rows interleaved and k ordered as gpp_stem_pack_weights_f16
    const size_t need = (size_t)2 * 64 * MW_PITCH * 2 + 64 * sizeof(float);
    if (!host_weight_147x64 || !host_packed || packed_bytes < need) return GPP_ERR_BAD_ARG;
    _Float16* hi = (_Float16*)host_packed;
    _Float16* lo = hi + 64 * MW_PITCH;
    float* out_scale = (float*)(lo + 64 * MW_PITCH);
    for (int pos = 0; pos < 64; ++pos) {
        const int g = pos / 32, within = pos % 32, h = within / 16, q = (within % 16) / 4, r = within % 4;
        const int n = g * 32 + 8 * q + 4 * h + r;
        float amax = 0.0f;
        for (int k = 0; k < 147; ++k) amax = fmaxf(amax, fabsf(host_weight_147x64[k * 64 + n]));
        int e = 0;
        if (amax > 0.0f) { (void)frexpf(amax, &e); e = 14 - e; }          // amax = m * 2^(14 - e_new), m in [0.5, 1) -> amax * 2^e in [2^13, 2^14)
        const float sc = ldexpf(1.0f, e);
        out_scale[n] = ldexpf(1.0f, -e);
        for (int k = 0; k < MW_PITCH; ++k) {
            const int kh = k / 32, kc = k % 32;
            float v = 0.0f;
            if (k < 224 && kc < 21) v = host_weight_147x64[(kh * 21 + kc) * 64 + n] * sc;
            const _Float16 vh = (_Float16)v;
            hi[pos * MW_PITCH + k] = vh;
            lo[pos * MW_PITCH + k] = (_Float16)(v - (float)vh);
        }
    }
    return GPP_OK;
}

extern "C" int gpp_stem_conv7x7_bn_relu_x3(const float* in, const void* packed_weight_x3, const float* bias, float* out,
                                           int B, int H, int W, void* stream)
{
    return gpp_stem_conv7x7_bn_relu_x3_rc(in, packed_weight_x3, bias, out, B, H, W, nullptr, stream);
}

extern "C" int gpp_stem_conv7x7_bn_relu_x3_rc(const float* in, const void* packed_weight_x3, const float* bias, float* out,
                                              int B, int H, int W, uint64_t* range_counter, void* stream)
{
    if (!in || !packed_weight_x3 || !bias || !out || B <= 0 || H <= 0 || W <= 0) return GPP_ERR_BAD_ARG;
    if ((uintptr_t)range_counter & 7) return GPP_ERR_ALIGN;
    if (((uintptr_t)out | (uintptr_t)packed_weight_x3) & 15) return GPP_ERR_ALIGN;
    constexpr int ROWS = 8;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const int tiles = ((Wo + TW - 1) / TW) * ((Ho + ROWS - 1) / ROWS) * B;
    const int lds = 2 * 64 * MW_PITCH * 2 + 2 * (ROWS * 2 + 5) * MP_PITCH * 2;
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GPP_ERR_UNSUPPORTED;
    if (!(configured.load(std::memory_order_acquire) >> dev & 1ull)) {
        hipError_t e = hipFuncSetAttribute((const void*)stem_mfma_x3_kernel<ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        configured.fetch_or(1ull << dev, std::memory_order_release);
    }
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);                           // persistent workgroups, one per CU
    unsigned long long* counter = range_counter ? (unsigned long long*)range_counter : gpp_x3_range_counter_f16x3();   // (the library's: cached per device there)
    if (!counter) return GPP_ERR_UNSUPPORTED;
    stem_mfma_x3_kernel<ROWS><<<grid, 64 * ROWS, lds, (hipStream_t)stream>>>(in, (const _Float16*)packed_weight_x3, bias, out, B, H, W, Ho, Wo, counter);
    return result();
}

extern "C" int gpp_stem_conv7x7_bn_relu_mfma(const float* in, const void* packed_weight_f16, const float* bias, void* out,
                                             int dtype, int B, int H, int W, void* stream)
{
    if (!in || !packed_weight_f16 || !bias || !out || B <= 0 || H <= 0 || W <= 0) return GPP_ERR_BAD_ARG;
    if (((uintptr_t)out | (uintptr_t)packed_weight_f16) & 15) return GPP_ERR_ALIGN;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const int tiles = ((Wo + TW - 1) / TW) * ((Ho + TH - 1) / TH) * B;
    static const int per_cu = [] { const char* e = getenv("GPP_STEM_WGS_PER_CU"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
    const unsigned grid = (unsigned)(tiles < 256 * per_cu ? tiles : 256 * per_cu);      // persistent workgroups
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GPP_BF16)
        stem_mfma_kernel<__bf16, bf16x8><<<grid, 256, 0, st>>>(in, (const _Float16*)packed_weight_f16, bias, (__bf16*)out, B, H, W, Ho, Wo);
    else if (dtype == GPP_F16)
        stem_mfma_kernel<_Float16, f16x8><<<grid, 256, 0, st>>>(in, (const _Float16*)packed_weight_f16, bias, (_Float16*)out, B, H, W, Ho, Wo);
    else
        return GPP_ERR_UNSUPPORTED;
    return result();
}

extern "C" int gpp_stem_pool_fused_mfma(const float* in, const void* packed_weight_f16, const float* bias, void* out,
                                        int dtype, int B, int H, int W, void* stream)
{
    if (!in || !packed_weight_f16 || !bias || !out || B <= 0 || H <= 0 || W <= 0) return GPP_ERR_BAD_ARG;
    if (((uintptr_t)out | (uintptr_t)packed_weight_f16) & 15) return GPP_ERR_ALIGN;
    if (dtype != GPP_BF16 && dtype != GPP_F16) return GPP_ERR_UNSUPPORTED;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const int Hp = (Ho + 1) / 2, Wp = (Wo + 1) / 2;
    const int pt = ((Hp - 1) * 2 + 3 - Ho > 0 ? (Hp - 1) * 2 + 3 - Ho : 0) / 2;
    const int pl = ((Wp - 1) * 2 + 3 - Wo > 0 ? (Wp - 1) * 2 + 3 - Wo : 0) / 2;
    // 8 conv rows per step, one workgroup per CU; GPP_STEM_POOL_ROWS=4: 4 rows, two workgroups per CU (measured equal:
    // 81 against 82 us at B = 8, 402 x 1333, tools/stem_time.py -- the kernel is bound by its LDS reads and patch loads, not by
    // the order of its phases)
    static const int rows = [] { const char* e = getenv("GPP_STEM_POOL_ROWS"); return (e && atoi(e) == 4) ? 4 : 8; }();
    const int pr = rows / 2;
    const int64_t total = (int64_t)B * ((Wp + FP_PCOLS - 1) / FP_PCOLS) * ((Hp + pr - 1) / pr);
    if (total >= (1LL << 30)) return GPP_ERR_UNSUPPORTED;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return GPP_ERR_UNSUPPORTED;
    const int64_t slots = (int64_t)cus * (8 / rows);                             // persistent workgroups
    const unsigned grid = (unsigned)(total < slots ? total : slots);
    hipStream_t st = (hipStream_t)stream;
    // hipFuncSetAttribute is per device: remember the devices each instantiation has been configured on
    static std::atomic<uint64_t> done[4];
    const int which = (dtype == GPP_BF16 ? 0 : 1) + (rows == 8 ? 0 : 2);
    const void* fns[4] = {(const void*)stem_pool_mfma_kernel<__bf16, bf16x8, 8>, (const void*)stem_pool_mfma_kernel<_Float16, f16x8, 8>,
                          (const void*)stem_pool_mfma_kernel<__bf16, bf16x8, 4>, (const void*)stem_pool_mfma_kernel<_Float16, f16x8, 4>};
    const int lds = fp_lds(rows);
    if (!(done[which].load(std::memory_order_acquire) & (1ull << (dev & 63)))) {
        const hipError_t e = hipFuncSetAttribute(fns[which], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        done[which].fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const _Float16* wp = (const _Float16*)packed_weight_f16;
    if (which == 0) stem_pool_mfma_kernel<__bf16, bf16x8, 8><<<grid, 512, lds, st>>>(in, wp, bias, (__bf16*)out, B, H, W, Ho, Wo, Hp, Wp, pt, pl);
    else if (which == 1) stem_pool_mfma_kernel<_Float16, f16x8, 8><<<grid, 512, lds, st>>>(in, wp, bias, (_Float16*)out, B, H, W, Ho, Wo, Hp, Wp, pt, pl);
    else if (which == 2) stem_pool_mfma_kernel<__bf16, bf16x8, 4><<<grid, 256, lds, st>>>(in, wp, bias, (__bf16*)out, B, H, W, Ho, Wo, Hp, Wp, pt, pl);
    else stem_pool_mfma_kernel<_Float16, f16x8, 4><<<grid, 256, lds, st>>>(in, wp, bias, (_Float16*)out, B, H, W, Ho, Wo, Hp, Wp, pt, pl);
    return result();
}

extern "C" int gpp_stem_pool_fused_x3(const float* in, const void* packed_weight_x3, const float* bias, float* out,
                                      int B, int H, int W, uint64_t* range_counter, void* stream)
{
    if (!in || !packed_weight_x3 || !bias || !out || B <= 0 || H <= 0 || W <= 0) return GPP_ERR_BAD_ARG;
    if ((uintptr_t)range_counter & 7) return GPP_ERR_ALIGN;
    if (((uintptr_t)out | (uintptr_t)packed_weight_x3) & 15) return GPP_ERR_ALIGN;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const int Hp = (Ho + 1) / 2, Wp = (Wo + 1) / 2;
    const int pt = ((Hp - 1) * 2 + 3 - Ho > 0 ? (Hp - 1) * 2 + 3 - Ho : 0) / 2;
    const int pl = ((Wp - 1) * 2 + 3 - Wo > 0 ? (Wp - 1) * 2 + 3 - Wo : 0) / 2;
    // 6 conv rows per step (6 wavefronts); GPP_STEM_POOL_X3_ROWS=4: 4.  One workgroup per CU either way (the hi / lo weights alone are 58 KB).
    static const int rows = [] { const char* e = getenv("GPP_STEM_POOL_X3_ROWS"); return (e && atoi(e) == 4) ? 4 : 6; }();
    const int pr = rows / 2;
    const int64_t total = (int64_t)B * ((Wp + FP_PCOLS - 1) / FP_PCOLS) * ((Hp + pr - 1) / pr);
    if (total >= (1LL << 30)) return GPP_ERR_UNSUPPORTED;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return GPP_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)(total < cus ? total : cus);                  // persistent workgroups
    hipStream_t st = (hipStream_t)stream;
    static std::atomic<uint64_t> done[2];
    const int which = rows == 6 ? 0 : 1;
    const void* fns[2] = {(const void*)stem_pool_mfma_x3_kernel<6>, (const void*)stem_pool_mfma_x3_kernel<4>};
    const int lds = xp_lds(rows);
    if (!(done[which].load(std::memory_order_acquire) & (1ull << (dev & 63)))) {
        const hipError_t e = hipFuncSetAttribute(fns[which], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        done[which].fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    unsigned long long* counter = range_counter ? (unsigned long long*)range_counter : gpp_x3_range_counter_f16x3();
    if (!counter) return GPP_ERR_UNSUPPORTED;
    const _Float16* wp = (const _Float16*)packed_weight_x3;
    if (which == 0) stem_pool_mfma_x3_kernel<6><<<grid, 384, lds, st>>>(in, wp, bias, out, B, H, W, Ho, Wo, Hp, Wp, pt, pl, counter);
    else stem_pool_mfma_x3_kernel<4><<<grid, 256, lds, st>>>(in, wp, bias, out, B, H, W, Ho, Wo, Hp, Wp, pt, pl, counter);
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
    else if (dtype == GPP_F32)
        maxpool_kernel<float, f32x8><<<blocks, 256, 0, st>>>((const float*)in, (float*)out, B, H, W, C, Ho, Wo, pt, pl);
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
    if (dtype == GPP_BF16X3 || dtype == GPP_F16X3) {      // a pre-split map (the sign bit of a half sits in the same place for both types)
        if (count % 32 != 0 || in_bstride % 32 != 0 || out_bstride % 32 != 0) return GPP_ERR_BAD_ARG;
        relu_x3_kernel<<<grid, 256, 0, st>>>((const char*)in, in_bstride * 4, (char*)out, out_bstride * 4, count / 8);
        return result();
    }
    if (dtype == GPP_BF16)
        relu_kernel<__bf16, bf16x8><<<grid, 256, 0, st>>>((const __bf16*)in, in_bstride, (__bf16*)out, out_bstride, n8);
    else if (dtype == GPP_F16)
        relu_kernel<_Float16, f16x8><<<grid, 256, 0, st>>>((const _Float16*)in, in_bstride, (_Float16*)out, out_bstride, n8);
    else if (dtype == GPP_F32)
        relu_kernel<float, f32x8><<<grid, 256, 0, st>>>((const float*)in, in_bstride, (float*)out, out_bstride, n8);
    else
        return GPP_ERR_UNSUPPORTED;
    return result();
}

extern "C" int gpp_relu(const void* in, void* out, int dtype, int64_t count, void* stream)
{
    return gpp_relu_strided(in, 0, out, 0, dtype, 1, count, stream);
}
