// The K-step a position-sequential Winograd F(2, 3) form of the 3 x 3 layers would run -- the ONE fast-convolution form the arithmetic of
// profiles/r5/fastconv_feasibility.md does not exclude -- beside the shipped K-step, on identical operand data and the same box.
//
// F(2, 3) along W: 4 position GEMMs per 2 outputs (direct: 6), each over K = 3 kernel rows x C_in.  All 4 position accumulators of a
// 256 x 256 tile do not fit the register file, so the positions are run one after the other over the same tile (M_p accumulated, then added
// into Y0 / Y1): THREE accumulator sets live (M, Y0, Y1) instead of one.  What fits 2 wavefronts per SIMD is a per-position tile of
// 192 (output pairs) x 128 (channels): 8 wavefronts 4 x 2, wavefront tile 48 x 64 = 12 accumulators x 3 sets = 144 registers.
// Per K-step of 32 channels (x3 arithmetic, three phases as the shipped loop):
//     36 MFMAs, 14 ds_read_b128, 5 LDS-DMA pieces per wavefront;  stage = (192 + 128) rows x 128 B = 40 KB, THREE stages in flight (120 KB)
//     shipped 256 x 256 tile: 96 MFMAs, 24 reads, 8 pieces per wavefront; 64 KB stages, two in flight
// One Winograd MFMA is worth 1.5 direct ones (4 instead of 6 per output pair): the last column prints the DIRECT-EQUIVALENT rate
// (MFMA rate x 1.5) next to the shipped K-step's rate measured by the same binary.
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/kstep_wino.hip -o gpurun_out/kstep_wino      run: gpurun_out/kstep_wino
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int kRow = 128;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, int voffset, void* lds_dst_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, voffset, 0, 0, 0);
}

// AROWS x BROWS per-position tile, WM x WN wavefronts, STAGES stages in flight; the shipped K-step is <256, 256, 2, 4, 2>
template <int AROWS, int BROWS, int WM, int WN, int STAGES>
__global__ __launch_bounds__(512, 1) void kstep(const unsigned char* image, int iters, float* sink)
{
    constexpr int kStage = (AROWS + BROWS) * kRow, kABytes = AROWS * kRow;
    constexpr int MF = AROWS / WM / 16, NF = BROWS / WN / 16;
    constexpr int PIECES = (AROWS + BROWS) / 8 / 8;             // 1 KB (8-row) pieces per wavefront and stage
    static_assert((AROWS + BROWS) % 64 == 0, "whole pieces per wavefront");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const unsigned char* mine = image + (size_t)blockIdx.x * kStage;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, kStage, 0x00020000);
    auto piece = [&](int p) { return (wave * PIECES + p) * 1024; };
    for (int s = 0; s < STAGES; ++s)
        for (int p = 0; p < PIECES; ++p) glds16(rsrc, piece(p) + lane * 16, smem + s * kStage + piece(p));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 15, fq = lane >> 4;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fq) ^ (frow & 7)) << 4;
        a_rd[kk] = (wm * (MF * 16) + frow) * kRow + sw;
        b_rd[kk] = kABytes + (wn * (NF * 16) + frow) * kRow + sw;
    }
    f16x8 ah[MF], al[MF], bh[NF], bl[NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) { ah[i] = *(const f16x8*)(smem + a_rd[0] + i * 16 * kRow); al[i] = *(const f16x8*)(smem + a_rd[1] + i * 16 * kRow); }
#pragma unroll
    for (int j = 0; j < NF; ++j) { bh[j] = *(const f16x8*)(smem + b_rd[0] + j * 16 * kRow); bl[j] = *(const f16x8*)(smem + b_rd[1] + j * 16 * kRow); }
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    int cur = 0;
    for (int ks = 0; ks < iters; ++ks) {
        const int nxt = cur + 1 == STAGES ? 0 : cur + 1;
        // the DMA of this step (issued behind the barrier) rewrites the stage this step has just finished reading, with the data of step
        // ks + STAGES: first read STAGES - 1 steps later (as `snxt`): one K-step of latency tolerance with two stages, two with three
        const unsigned char* scur = smem + cur * kStage;
        const unsigned char* snxt = smem + nxt * kStage;
        FENCE();
        // ---- phase A: hi * wlo; reads whi
#pragma unroll
        for (int g = 0; g < MF; ++g) {
#pragma unroll
            for (int q = g * NF / MF; q < (g + 1) * NF / MF; ++q) bh[q] = *(const f16x8*)(scur + b_rd[0] + q * 16 * kRow);
#pragma unroll
            for (int j = 0; j < NF; ++j) { const int js = (g & 1) ? NF - 1 - j : j; acc[g][js] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[js], ah[g], acc[g][js], 0, 0, 0); }
            FENCE();
        }
        // ---- phase B: hi * whi; reads lo
#pragma unroll
        for (int g = 0; g < MF; ++g) {
            al[g] = *(const f16x8*)(scur + a_rd[1] + g * 16 * kRow);
#pragma unroll
            for (int j = 0; j < NF; ++j) { const int js = (g & 1) ? NF - 1 - j : j; acc[g][js] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[js], ah[g], acc[g][js], 0, 0, 0); }
            FENCE();
        }
        // the next stage must have landed: with two stages everything issued so far; with three the newest step's pieces may stay in flight
        if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        FENCE();
        // ---- phase C: lo * whi; the DMA of the stage after next; reads hi and wlo of the next stage
#pragma unroll
        for (int g = 0; g < MF; ++g) {
#pragma unroll
            for (int q = g * PIECES / MF; q < (g + 1) * PIECES / MF; ++q) glds16(rsrc, piece(q) + lane * 16, smem + cur * kStage + piece(q));
            ah[g] = *(const f16x8*)(snxt + a_rd[0] + g * 16 * kRow);
#pragma unroll
            for (int q = g * NF / MF; q < (g + 1) * NF / MF; ++q) bl[q] = *(const f16x8*)(snxt + b_rd[1] + q * 16 * kRow);
#pragma unroll
            for (int j = 0; j < NF; ++j) { const int js = (g & 1) ? NF - 1 - j : j; acc[g][js] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[js], al[g], acc[g][js], 0, 0, 0); }
            FENCE();
        }
        cur = nxt;
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

template <int AROWS, int BROWS, int WM, int WN, int STAGES>
static double run(const unsigned char* d_img, float* d_sink, int iters)
{
    constexpr int lds = STAGES * (AROWS + BROWS) * kRow;
    CHECK(hipFuncSetAttribute((const void*)kstep<AROWS, BROWS, WM, WN, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    kstep<AROWS, BROWS, WM, WN, STAGES><<<256, 512, lds>>>(d_img, iters, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ms;
}

// stage images: `arows` activation rows (post-ReLU) then `brows` weight rows, [hi 64 B | lo 64 B] per row, 16-byte chunks swizzled by the row
static void fill_image(std::vector<uint16_t>& h, int arows, int brows, bool zeros)
{
    const size_t stage = (size_t)(arows + brows) * kRow;
    h.assign(256 * stage / 2, 0);
    for (size_t wg = 0; wg < 256; ++wg)
        for (int row = 0; row < arows + brows; ++row) {
            float v[32];
            for (int c = 0; c < 32; ++c) {
                float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
                float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
                v[c] = zeros ? 0.f : (row < arows ? fmaxf(g, 0.f) : g * 32.f);
            }
            for (int p = 0; p < 8; ++p) {
                const int c = p ^ (row & 7);
                for (int e = 0; e < 8; ++e) {
                    const float x = v[(c & 3) * 8 + e];
                    const uint16_t hi = f2h(x);
                    h[(wg * stage + (size_t)row * kRow + p * 16) / 2 + e] = c < 4 ? hi : f2h(x - h2f(hi));
                }
            }
        }
}

int main()
{
    unsigned char* d_img; float* d_sink;
    CHECK(hipMalloc(&d_img, (size_t)256 * 512 * kRow)); CHECK(hipMalloc(&d_sink, 64));
    srand(7);
    std::vector<uint16_t> h;
    const int iters = 6000;
    for (int pass = 0; pass < 2; ++pass) {
        double ms[4] = {0, 0, 0, 0};
        const char* names[4] = {"shipped K-step: 256 x 256 tile, 2 stages of 64 KB (direct 3 x 3)",
                                "F(2,3) position pass: 192 x 128 tile, 3 stages of 40 KB", "F(2,3) position pass: 192 x 128 tile, 2 stages of 40 KB",
                                "F(2,3) position pass: 128 x 128 tile, 3 stages of 32 KB"};
        const double mfma[4] = {96, 36, 36, 24};
        for (int rep = 0; rep < 5; ++rep) {
            fill_image(h, 256, 256, pass == 1); CHECK(hipMemcpy(d_img, h.data(), h.size() * 2, hipMemcpyHostToDevice));
            if (rep == 0) run<256, 256, 2, 4, 2>(d_img, d_sink, 300);
            ms[0] += run<256, 256, 2, 4, 2>(d_img, d_sink, iters);
            fill_image(h, 192, 128, pass == 1); CHECK(hipMemcpy(d_img, h.data(), h.size() * 2, hipMemcpyHostToDevice));
            ms[1] += run<192, 128, 4, 2, 3>(d_img, d_sink, iters);
            ms[2] += run<192, 128, 4, 2, 2>(d_img, d_sink, iters);
            fill_image(h, 128, 128, pass == 1); CHECK(hipMemcpy(d_img, h.data(), h.size() * 2, hipMemcpyHostToDevice));
            ms[3] += run<128, 128, 4, 2, 3>(d_img, d_sink, iters);
        }
        for (int o = 0; o < 4; ++o) {
            const double flop = 256.0 * 8 * iters * mfma[o] * 2.0 * 16 * 16 * 32, t = ms[o] / 5;
            const double rate = flop / t / 1e9, equiv = o == 0 ? rate : rate * 1.5;
            printf("%-8s %-66s K-step %.3f us  %7.1f TFLOP/s of MFMA  -> direct-equivalent %7.1f TFLOP/s = %5.1f of float32 products\n",
                   pass ? "zeros" : "relu(A)", names[o], t / iters * 1e3, rate, equiv, equiv / 3);
        }
    }
    return 0;
}
