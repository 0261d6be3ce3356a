// What does the matrix pipe sustain on REAL operand data?  Register-only loops of v_mfma_f32_16x16x32_{f16,bf16} and
// v_mfma_f32_32x32x16_{f16,bf16} (no LDS, no memory in the loop), 2 wavefronts per SIMD, every CU busy, operands = random normal values
// or zeros.  The conv kernels' rate depends on the operand data (profiles/r3/mfma_rate_depends_on_data.txt): this separates the clock
// under matrix load from everything a schedule could change.
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o gpurun_out/mfma_power      run: gpurun_out/mfma_power
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// SHAPE 0: 16x16x32 (16 cycles, 16 accumulators of 4 registers in flight); SHAPE 1: 32x32x16 (32 cycles, 4 accumulators of 16)
template <int SHAPE, bool BF>
__global__ __launch_bounds__(512, 2) void mfma_loop(const uint4* ops, int iters, float* sink)
{
    const int lane = threadIdx.x & 63;
    // 8 A fragments and 4 B fragments per wavefront, as the 128 x 64 wavefront tile of the conv kernel holds them
    uint4 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = ops[(i * 64 + lane) % 4096];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = ops[((8 + i) * 64 + lane + 17) % 4096];
    if constexpr (SHAPE == 0) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (BF) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&b[j], *(bf16x8*)&a[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(f16x8*)&b[j], *(f16x8*)&a[i], acc[i][j], 0, 0, 0);
                }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
        if (s == 12345.678f) sink[0] = s;
    } else {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)                       // the same 128 x 64 x 32 of products per iteration: 4 x 2 blocks x 2 k-halves
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (BF) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8*)&b[j * 2 + kk], *(bf16x8*)&a[i * 2 + kk], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(f16x8*)&b[j * 2 + kk], *(f16x8*)&a[i * 2 + kk], acc[i][j], 0, 0, 0);
                    }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
        if (s == 12345.678f) sink[0] = s;
    }
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

template <int SHAPE, bool BF>
static void run(const char* name, const uint4* d_ops, float* d_sink, const char* data)
{
    const int iters = 4000, grid = 512;                   // 2 workgroups of 8 wavefronts per CU: 4 wavefronts per SIMD ... (launch bounds: 2)
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    mfma_loop<SHAPE, BF><<<grid, 512>>>(d_ops, 200, d_sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        mfma_loop<SHAPE, BF><<<grid, 512>>>(d_ops, iters, d_sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    const double flop = (double)grid * 8 * iters * 32 * 2.0 * 16 * 16 * 32;      // per wavefront and iteration: 32 x (16x16x32) = 8 x (32x32x16) products
    printf("%-22s %-8s  mean %.3f ms  ->  %7.1f TFLOP/s  (best %.1f)\n", name, data, sum / 5, flop / (sum / 5) / 1e9, flop / best / 1e9);
}

int main()
{
    std::vector<uint16_t> h(4096 * 8);
    uint4* d_ops; float* d_sink;
    CHECK(hipMalloc(&d_ops, 4096 * 16)); CHECK(hipMalloc(&d_sink, 64));
    srand(7);
    for (int pass = 0; pass < 3; ++pass) {               // random data, zeros, then max(random, 0) in the activation fragments (post-ReLU maps)
        for (int type = 0; type < 2; ++type) {
            for (size_t i = 0; i < h.size(); ++i) {
                float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
                float v = pass == 1 ? 0.0f : sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * 0.5f;
                if (pass == 2 && i < 8 * 64 * 8 && v < 0.0f) v = 0.0f;            // the first 8 x 64 x 8 values are the A (activation) fragments
                h[i] = type ? f2bf(v) : f2h(v);
            }
            CHECK(hipMemcpy(d_ops, h.data(), 4096 * 16, hipMemcpyHostToDevice));
            const char* data = pass == 0 ? "random" : pass == 1 ? "zeros" : "relu(A)";
            if (type == 0) { run<0, false>("mfma 16x16x32 f16", d_ops, d_sink, data); run<1, false>("mfma 32x32x16 f16", d_ops, d_sink, data); }
            else { run<0, true>("mfma 16x16x32 bf16", d_ops, d_sink, data); run<1, true>("mfma 32x32x16 bf16", d_ops, d_sink, data); }
        }
    }
    return 0;
}
