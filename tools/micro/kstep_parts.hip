// What each non-matrix part of the x3 K-step costs the matrix pipe, on IDENTICAL operand data (the pipe's clock under load depends on
// the data, so the library's ablation switches -- which leave stale registers behind when a read is removed -- cannot tell).
// The loop below is the pipelined K-step of conv_igemm_impl.h for the 256 x 256 tile (8 wavefronts, 128 x 64 each, 96 MFMAs per
// wavefront and K-step in three phases, serpentine order, one s_barrier), on a 2 x 64 KB LDS ring whose CONTENT NEVER CHANGES: the
// fragment reads return the same values every step and the LDS-DMA rewrites the same bytes from an L2-resident image, so every variant
// multiplies the same numbers:
//   READS  0: operands stay in registers              1: the 24 ds_read_b128 per wavefront and K-step, placed as in the kernel
//   DMA    0: none     1: the 8 buffer_load_dwordx4 ... lds per wavefront and K-step (4 in phase C, 4 in phase A), vmcnt(0) before the barrier
//   BAR    0 / 1: s_barrier per K-step
//   NW     8: 2 x 4 wavefronts of 128 x 64 (the kernel)      4: 2 x 2 wavefronts of 128 x 128, one per SIMD -- 2/3 of the LDS read bytes
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/kstep_parts.hip -o gpurun_out/kstep_parts      run: gpurun_out/kstep_parts
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int kRow = 128, kStage = 65536, kABytes = 32768;

template <int AUX = 0>
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, int voffset, void* lds_dst_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, voffset, 0, 0, AUX);
}

template <int READS, int DMA, int BAR, int NW = 8, int AUX = 0, int ASHARE = 1>
__global__ __launch_bounds__(NW * 64, 1) void kstep(const unsigned char* image, int iters, float* sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NF = NW == 8 ? 4 : 8, PC = 32 / NW;        // weight fragments per wavefront (64 or 128 columns); 1 KB pieces per wavefront and tile
    const int wm = NW == 8 ? wave >> 2 : wave >> 1, wn = NW == 8 ? wave & 3 : wave & 1;
    const unsigned char* mine = image + (size_t)blockIdx.x * kStage;               // this workgroup's 64 KB stage image
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, kStage, 0x00020000);
    // piece p (0..3) of the activation tile / of the weight tile that this wavefront moves: 1 KB = 8 rows each
    auto a_off = [&](int p) { return (wave * PC + p) * 1024; };
    auto b_off = [&](int p) { return kABytes + (wave * PC + p) * 1024; };
    for (int s = 0; s < 2; ++s)
        for (int p = 0; p < PC; ++p) {
            glds16(rsrc, a_off(p) + lane * 16, smem + s * kStage + a_off(p));
            glds16(rsrc, b_off(p) + lane * 16, smem + s * kStage + b_off(p));
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 15, fq = lane >> 4;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fq) ^ (frow & 7)) << 4;
        a_rd[kk] = (wm * 128 + frow) * kRow + sw;
        b_rd[kk] = kABytes + (wn * (NF * 16) + frow) * kRow + sw;
    }
    f16x8 ah[8], al[8], bh[NF], bl[NF];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ah[i] = *(const f16x8*)(smem + a_rd[0] + i * 16 * kRow); al[i] = *(const f16x8*)(smem + a_rd[1] + i * 16 * kRow); }
#pragma unroll
    for (int j = 0; j < NF; ++j) { bh[j] = *(const f16x8*)(smem + b_rd[0] + j * 16 * kRow); bl[j] = *(const f16x8*)(smem + b_rd[1] + j * 16 * kRow); }
    f32x4 acc[8][NF];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);                                 // as the kernel: the second wavefront of every SIMD goes first
    // (ASHARE 3: the activation rows of three K-steps -- the horizontal taps of a kernel row -- in one transfer: three copies of the body, the first one moves them)
    auto step = [&](const int ks, auto share_tag) {
        constexpr bool share_now = decltype(share_tag)::value;
        const int cur = ks & 1;
        const unsigned char* scur = smem + cur * kStage;
        const unsigned char* snxt = smem + (cur ^ 1) * kStage;
        FENCE();
        // ---- phase A: hi * wlo; reads whi; the weight pieces of the next stage go out
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int q = g * PC / 8; q < (g + 1) * PC / 8; ++q) if (DMA) glds16<AUX>(rsrc, b_off(q) + lane * 16, smem + (cur ^ 1) * kStage + b_off(q));
#pragma unroll
            for (int q = g * NF / 8; q < (g + 1) * NF / 8; ++q) if (READS) bh[q] = *(const f16x8*)(scur + b_rd[0] + q * 16 * kRow);
#pragma unroll
            for (int j = 0; j < NF; ++j) { const int js = (g & 1) ? NF - 1 - j : j; acc[g][js] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[js], ah[g], acc[g][js], 0, 0, 0); }
            FENCE();
        }
        // ---- phase B: hi * whi; reads lo
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (READS) al[g] = *(const f16x8*)(scur + a_rd[1] + g * 16 * kRow);
#pragma unroll
            for (int j = 0; j < NF; ++j) { const int js = (g & 1) ? NF - 1 - j : j; acc[g][js] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[js], ah[g], acc[g][js], 0, 0, 0); }
            FENCE();
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BAR) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        FENCE();
        // ---- phase C: lo * whi; the activation pieces of the stage after next go out; reads hi and wlo of the next stage
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int q = g * PC / 8; q < (g + 1) * PC / 8; ++q) if (DMA && (ASHARE == 1 || share_now)) glds16<AUX>(rsrc, a_off(q) + lane * 16, smem + cur * kStage + a_off(q));
            if (READS) {
                ah[g] = *(const f16x8*)(snxt + a_rd[0] + g * 16 * kRow);
#pragma unroll
                for (int q = g * NF / 8; q < (g + 1) * NF / 8; ++q) bl[q] = *(const f16x8*)(snxt + b_rd[1] + q * 16 * kRow);
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) { const int js = (g & 1) ? NF - 1 - j : j; acc[g][js] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[js], al[g], acc[g][js], 0, 0, 0); }
            FENCE();
        }
    };
    if constexpr (ASHARE == 1) {
        for (int ks = 0; ks < iters; ++ks) step(ks, std::true_type{});
    } else {
        for (int ks = 0; ks + 2 < iters; ks += 3) { step(ks, std::true_type{}); step(ks + 1, std::false_type{}); step(ks + 2, std::false_type{}); }
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

template <int READS, int DMA, int BAR, int NW = 8, int AUX = 0, int ASHARE = 1>
static double run(const unsigned char* d_img, float* d_sink, int iters)
{
    static bool once = false;
    if (!once) { once = true; }
    CHECK(hipFuncSetAttribute((const void*)kstep<READS, DMA, BAR, NW, AUX, ASHARE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStage));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    kstep<READS, DMA, BAR, NW, AUX, ASHARE><<<256, NW * 64, 2 * kStage>>>(d_img, iters, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    const size_t bytes = (size_t)256 * kStage;
    std::vector<uint16_t> h(bytes / 2);
    unsigned char* d_img; float* d_sink;
    CHECK(hipMalloc(&d_img, bytes)); CHECK(hipMalloc(&d_sink, 64));
    srand(7);
    for (int pass = 0; pass < 2; ++pass) {                 // post-ReLU activations x random weights; zeros
        for (size_t wg = 0; wg < 256; ++wg)
            for (int row = 0; row < 512; ++row) {          // rows 0..255: activations, 256..511: weights; physical 16-byte chunk p holds logical chunk p ^ (row & 7)
                float v[32];
                for (int c = 0; c < 32; ++c) {
                    float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
                    float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
                    v[c] = pass ? 0.f : (row < 256 ? fmaxf(g, 0.f) : g * 32.f);
                }
                for (int p = 0; p < 8; ++p) {
                    const int c = p ^ (row & 7);
                    for (int e = 0; e < 8; ++e) {
                        const float x = v[(c & 3) * 8 + e];
                        const uint16_t hi = f2h(x);
                        h[(wg * kStage + (size_t)row * kRow + p * 16) / 2 + e] = c < 4 ? hi : f2h(x - h2f(hi));
                    }
                }
            }
        CHECK(hipMemcpy(d_img, h.data(), bytes, hipMemcpyHostToDevice));
        const int iters = 6000;
        const double flop = 256.0 * 8 * iters * 96 * 2.0 * 16 * 16 * 32;
        double sum[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        run<0, 0, 0>(d_img, d_sink, 300);
        for (int rep = 0; rep < 5; ++rep) {
            sum[0] += run<0, 0, 0>(d_img, d_sink, iters); sum[1] += run<0, 0, 1>(d_img, d_sink, iters); sum[2] += run<1, 0, 1>(d_img, d_sink, iters);
            sum[3] += run<0, 1, 1>(d_img, d_sink, iters); sum[4] += run<1, 1, 1>(d_img, d_sink, iters); sum[5] += run<1, 1, 0>(d_img, d_sink, iters);
            sum[6] += run<0, 0, 1, 4>(d_img, d_sink, iters); sum[7] += run<1, 0, 1, 4>(d_img, d_sink, iters); sum[8] += run<1, 1, 1, 4>(d_img, d_sink, iters);
            sum[9] += run<1, 1, 1, 8, 1>(d_img, d_sink, iters); sum[10] += run<1, 1, 1, 8, 2>(d_img, d_sink, iters); sum[11] += run<1, 1, 1, 8, 3>(d_img, d_sink, iters);
            sum[12] += run<1, 1, 1, 8, 0, 3>(d_img, d_sink, iters);
        }
        const char* names[13] = {"MFMAs only", "+ s_barrier", "+ s_barrier + 24 LDS reads", "+ s_barrier + 8 LDS-DMA", "+ s_barrier + reads + DMA (the kernel's K-step)", "reads + DMA, no barrier (unsafe, timing only)",
                                "4 wavefronts of 128 x 128: MFMAs + s_barrier", "4 wavefronts: + 32 LDS reads each (2/3 of the bytes)", "4 wavefronts: + reads + 16 LDS-DMA each",
                                "the kernel's K-step, LDS-DMA with cache policy sc0", "the kernel's K-step, LDS-DMA with cache policy nt", "the kernel's K-step, LDS-DMA with sc0 + nt",
                                "the kernel's K-step, activation rows moved every 3rd step only"};
        for (int o = 0; o < 13; ++o)
            printf("%-8s %-52s mean %.3f ms  K-step %.3f us -> %7.1f TFLOP/s of MFMA = %5.1f of float32 products\n", pass ? "zeros" : "relu(A)", names[o], sum[o] / 5,
                   sum[o] / 5 / iters * 1e3, flop / (sum[o] / 5) / 1e9, flop / (sum[o] / 5) / 3e9);
    }
    return 0;
}
