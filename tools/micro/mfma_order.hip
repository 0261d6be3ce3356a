// Does the ORDER of the matrix instructions of an x3 K-step change what the (power-managed) matrix pipe sustains?
// Register-only loop with the operand structure of the x3 conv kernels: per wavefront 8 activation fragments (hi, lo) and 4 weight fragments
// (hi, lo) of a 128 x 64 wavefront tile, 96 v_mfma_f32_16x16x32_f16 per K-step (hi*wlo, hi*whi, lo*whi for each of 8 x 4 accumulators),
// operands = the two-half split of post-ReLU random activations / random weights, 2 wavefronts per SIMD, every CU busy.
//   ORDER 0  term-major, activation fragment outer, weight fragment inner (the kernels' order: 4 consecutive MFMAs share the activation operand)
//   ORDER 1  term-major, weight fragment outer, activation fragment inner (8 consecutive MFMAs share the weight operand)
//   ORDER 2  term-major, serpentine (one operand is always shared between consecutive MFMAs)
//   ORDER 3  term-major, diagonal (both operands change with every MFMA)
//   ORDER 4  activation fragment outer, term middle, weight fragment inner (the three terms of a fragment row together)
//   ORDER 5  term-major, weight fragment outer, activation fragments serpentine
//   ORDER 6  term-major, 2 x 2 blocks of accumulators (each operand shared by two of four consecutive MFMAs)
//   ORDER 7  as 4 with the middle term's weight fragments reversed (an operand shared across every boundary inside a row)
//   ORDER 8  weight fragment outer, the three terms of a weight fragment together
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_order.hip -o gpurun_out/mfma_order      run: gpurun_out/mfma_order
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
#define MFMA(w, a, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], a[i], acc[i][j], 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int ORDER>
__global__ __launch_bounds__(512, 1) void x3_loop(const f16x8* ops, int iters, float* sink)
{
    const int lane = threadIdx.x & 63;
    f16x8 ah[8], al[8], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ah[i] = ops[(i * 64 + lane) % 2048]; al[i] = ops[2048 + (i * 64 + lane) % 2048]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { bh[j] = ops[4096 + (j * 64 + lane + 17) % 2048]; bl[j] = ops[6144 + (j * 64 + lane + 17) % 2048]; }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        FENCE();
        if constexpr (ORDER == 4 || ORDER == 7) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) MFMA(bl, ah, i, j);
                FENCE();
#pragma unroll
                for (int j = 0; j < 4; ++j) MFMA(bh, ah, i, (ORDER == 7 ? 3 - j : j));
                FENCE();
#pragma unroll
                for (int j = 0; j < 4; ++j) MFMA(bh, al, i, j);
                FENCE();
            }
        } else if constexpr (ORDER == 8) {
            // weight fragment outer, the three terms of a weight fragment together: bl[j] x ah[0..7], bh[j] x ah[7..0], bh[j] x al[0..7]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) MFMA(bl, ah, i, j);
                FENCE();
#pragma unroll
                for (int i = 0; i < 8; ++i) MFMA(bh, ah, 7 - i, j);
                FENCE();
#pragma unroll
                for (int i = 0; i < 8; ++i) MFMA(bh, al, i, j);
                FENCE();
            }
        } else {
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int n = 0; n < 32; ++n) {
                    int i, j;
                    if constexpr (ORDER == 0) { i = n / 4; j = n % 4; }
                    else if constexpr (ORDER == 1) { j = n / 8; i = n % 8; }
                    else if constexpr (ORDER == 2) { i = n / 4; j = (i & 1) ? 3 - n % 4 : n % 4; }
                    else if constexpr (ORDER == 3) { i = n % 8; j = (n + n / 8) % 4; }
                    else if constexpr (ORDER == 5) { j = n / 8; i = (j & 1) ? 7 - n % 8 : n % 8; }
                    else { const int q = n / 4, r = n % 4; i = 2 * (q % 4) + (r == 1 || r == 2); j = 2 * (q / 4) + (r >= 2); }   // 6: 2 x 2 blocks
                    if (term == 0) MFMA(bl, ah, i, j);
                    else if (term == 1) MFMA(bh, ah, i, j);
                    else MFMA(bh, al, i, j);
                    FENCE();
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

template <int ORDER>
static double run(const f16x8* d_ops, float* d_sink, int iters)
{
    const int grid = 256;                                  // one workgroup of 8 wavefronts per CU: 2 wavefronts per SIMD, as the 256 x 256 tile
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    x3_loop<ORDER><<<grid, 512>>>(d_ops, iters, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

// What a workgroup barrier per K-step costs the matrix pipe: the serpentine loop above with (BAR) one s_barrier per 96 MFMAs, as the conv
// kernels have it, and (FILL) the ~40 scalar / vector instructions of the tap arithmetic after it.
template <int BAR, int FILL>
__global__ __launch_bounds__(512, 1) void x3_loop_bar(const f16x8* ops, int iters, float* sink, int one)
{
    const int lane = threadIdx.x & 63;
    f16x8 ah[8], al[8], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ah[i] = ops[(i * 64 + lane) % 2048]; al[i] = ops[2048 + (i * 64 + lane) % 2048]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { bh[j] = ops[4096 + (j * 64 + lane + 17) % 2048]; bl[j] = ops[6144 + (j * 64 + lane + 17) % 2048]; }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int t0 = lane, t1 = one, t2 = 3;
    for (int it = 0; it < iters; ++it) {
        FENCE();
#pragma unroll
        for (int term = 0; term < 3; ++term) {
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                const int i = n / 4, j = (i & 1) ? 3 - n % 4 : n % 4;
                if (term == 0) MFMA(bl, ah, i, j);
                else if (term == 1) MFMA(bh, ah, i, j);
                else MFMA(bh, al, i, j);
                FENCE();
            }
            if (BAR && term == 1) { __builtin_amdgcn_s_barrier(); FENCE(); }
        }
        if constexpr (FILL) {
#pragma unroll
            for (int f = 0; f < 10; ++f) {               // 4 dependent integer instructions each
                t0 = (t0 + t1) ^ t2;
                t2 = t2 * one + (t0 == 77 ? 1 : 0);
            }
            FENCE();
        }
    }
    float s = (float)(t0 + t2);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

template <int BAR, int FILL>
static double run_bar(const f16x8* d_ops, float* d_sink, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    x3_loop_bar<BAR, FILL><<<256, 512>>>(d_ops, iters, d_sink, 1);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

// Table-driven orders of the 32 accumulators of a phase (term-major, as the kernels' three phases): entry = 4 * activation fragment + weight fragment.
struct Tab { int v[32]; };
constexpr Tab make_tab(int kind)
{
    Tab t{};
    int n = 0;
    auto put = [&](int i, int j) { t.v[n++] = 4 * i + j; };
    if (kind == 0) { for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) put(i, j); }                                    // rows, left to right
    else if (kind == 1) { for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) put(i, (i & 1) ? 3 - j : j); }             // serpentine
    else if (kind == 2) { for (int jb = 0; jb < 2; ++jb) for (int ib = 0; ib < 4; ++ib) { put(2 * ib, 2 * jb); put(2 * ib + 1, 2 * jb); put(2 * ib + 1, 2 * jb + 1); put(2 * ib, 2 * jb + 1); } }   // 2 x 2 blocks, column pair outer
    else if (kind == 3) { for (int ib = 0; ib < 4; ++ib) for (int jb = 0; jb < 2; ++jb) { put(2 * ib, 2 * jb); put(2 * ib + 1, 2 * jb); put(2 * ib + 1, 2 * jb + 1); put(2 * ib, 2 * jb + 1); } }   // 2 x 2 blocks, row pair outer
    else if (kind == 4) {                                                                                                       // 2-row bands, snake through the columns, bands alternate direction
        for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) { const int j = (b & 1) ? 3 - c : c; const bool up = (c & 1) != 0; put(2 * b + (up ? 1 : 0), j); put(2 * b + (up ? 0 : 1), j); }
    }
    else if (kind == 5) { for (int jb = 0; jb < 2; ++jb) for (int r = 0; r < 8; ++r) { const int i = (jb & 1) ? 7 - r : r; const bool rev = (r & 1) != 0; put(i, 2 * jb + (rev ? 1 : 0)); put(i, 2 * jb + (rev ? 0 : 1)); } }   // column pair outer, snake down the rows
    else if (kind == 6) { for (int j = 0; j < 4; ++j) for (int r = 0; r < 8; ++r) put((j & 1) ? 7 - r : r, j); }                // columns, serpentine
    else if (kind == 7) { for (int ib = 0; ib < 2; ++ib) for (int jb = 0; jb < 2; ++jb) for (int r = 0; r < 4; ++r) { const bool rev = (r & 1) != 0; put(4 * ib + r, 2 * jb + (rev ? 1 : 0)); put(4 * ib + r, 2 * jb + (rev ? 0 : 1)); } }   // 4 x 2 blocks
    else if (kind == 8) { for (int ib = 0; ib < 4; ++ib) { for (int j = 0; j < 4; ++j) put(2 * ib, (ib & 1) ? 3 - j : j); for (int j = 0; j < 4; ++j) put(2 * ib + 1, (ib & 1) ? j : 3 - j); } }        // = serpentine (check)
    else { for (int d = 0; d < 32; ++d) put(d % 8, (d + d / 8) % 4); }                                                          // diagonal: both change
    return t;
}

constexpr bool is_permutation(Tab t)
{
    unsigned seen = 0;
    for (int n = 0; n < 32; ++n) { if (t.v[n] < 0 || t.v[n] > 31) return false; seen |= 1u << t.v[n]; }
    return seen == 0xffffffffu;
}

template <int KIND>
__global__ __launch_bounds__(512, 1) void x3_loop_tab(const f16x8* ops, int iters, float* sink)
{
    constexpr Tab T = make_tab(KIND);
    static_assert(is_permutation(T), "every accumulator once per phase");
    const int lane = threadIdx.x & 63;
    f16x8 ah[8], al[8], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ah[i] = ops[(i * 64 + lane) % 2048]; al[i] = ops[2048 + (i * 64 + lane) % 2048]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { bh[j] = ops[4096 + (j * 64 + lane + 17) % 2048]; bl[j] = ops[6144 + (j * 64 + lane + 17) % 2048]; }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        FENCE();
#pragma unroll
        for (int term = 0; term < 3; ++term) {
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                constexpr int dummy = 0; (void)dummy;
                const int i = T.v[n] / 4, j = T.v[n] % 4;
                if (term == 0) MFMA(bl, ah, i, j);
                else if (term == 1) MFMA(bh, ah, i, j);
                else MFMA(bh, al, i, j);
                FENCE();
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

template <int KIND>
static double run_tab(const f16x8* d_ops, float* d_sink, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    x3_loop_tab<KIND><<<256, 512>>>(d_ops, iters, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    std::vector<uint16_t> h(8192 * 8);
    f16x8* d_ops; float* d_sink;
    CHECK(hipMalloc(&d_ops, 8192 * 16)); CHECK(hipMalloc(&d_sink, 64));
    srand(7);
    for (int pass = 0; pass < 2; ++pass) {                 // post-ReLU activations x random weights; zeros
        for (size_t i = 0; i < 2048 * 8; ++i) {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
            float v = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
            float a = pass ? 0.f : fmaxf(v, 0.f), w = pass ? 0.f : v * 0.5f * 64.f;      // weights carry a power-of-two scale like the packed ones
            h[i] = f2h(a);                 h[2048 * 8 + i] = f2h(a - h2f(h[i]));
            h[4096 * 8 + i] = f2h(w);      h[6144 * 8 + i] = f2h(w - h2f(h[4096 * 8 + i]));
        }
        CHECK(hipMemcpy(d_ops, h.data(), 8192 * 16, hipMemcpyHostToDevice));
        const int iters = 6000;
        const double flop = 256.0 * 8 * iters * 96 * 2.0 * 16 * 16 * 32;
        double sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        run<0>(d_ops, d_sink, 300);
        for (int rep = 0; rep < 6; ++rep) {                // alternating, so that a drifting clock hits every order alike
            sum[0] += run<0>(d_ops, d_sink, iters); sum[1] += run<1>(d_ops, d_sink, iters); sum[2] += run<2>(d_ops, d_sink, iters);
            sum[3] += run<3>(d_ops, d_sink, iters); sum[4] += run<4>(d_ops, d_sink, iters); sum[5] += run<5>(d_ops, d_sink, iters);
            sum[6] += run<6>(d_ops, d_sink, iters); sum[7] += run<7>(d_ops, d_sink, iters); sum[8] += run<8>(d_ops, d_sink, iters);
        }
        const char* names[9] = {"term-major, activation outer (kernel)", "term-major, weight outer", "term-major, serpentine", "term-major, diagonal",
                                "row-major: 3 terms per activation fragment", "term-major, weight outer, serpentine", "term-major, 2 x 2 blocks",
                                "row-major, middle term reversed", "column-major: 3 terms per weight fragment"};
        for (int o = 0; o < 9; ++o)
            printf("%-8s %-46s mean %.3f ms -> %7.1f TFLOP/s of MFMA = %6.1f of float32 products\n", pass ? "zeros" : "relu(A)", names[o], sum[o] / 6, flop / (sum[o] / 6) / 1e9, flop / (sum[o] / 6) / 3e9);
        double st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int rep = 0; rep < 6; ++rep) {
            st[0] += run_tab<0>(d_ops, d_sink, iters); st[1] += run_tab<1>(d_ops, d_sink, iters); st[2] += run_tab<2>(d_ops, d_sink, iters);
            st[3] += run_tab<3>(d_ops, d_sink, iters); st[4] += run_tab<4>(d_ops, d_sink, iters); st[5] += run_tab<5>(d_ops, d_sink, iters);
            st[6] += run_tab<6>(d_ops, d_sink, iters); st[7] += run_tab<7>(d_ops, d_sink, iters); st[8] += run_tab<8>(d_ops, d_sink, iters);
            st[9] += run_tab<9>(d_ops, d_sink, iters);
        }
        const char* nt[10] = {"table: rows, left to right", "table: serpentine", "table: 2 x 2 blocks, column pair outer", "table: 2 x 2 blocks, row pair outer",
                              "table: 2-row bands, snake", "table: column pairs, snake down the rows", "table: columns, serpentine", "table: 4 x 2 blocks",
                              "table: serpentine by row pairs (= serpentine)", "table: diagonal"};
        for (int o = 0; o < 10; ++o)
            printf("%-8s %-50s mean %.3f ms -> %7.1f TFLOP/s of MFMA\n", pass ? "zeros" : "relu(A)", nt[o], st[o] / 6, flop / (st[o] / 6) / 1e9);
        double sb[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 6; ++rep) {
            sb[0] += run_bar<0, 0>(d_ops, d_sink, iters); sb[1] += run_bar<1, 0>(d_ops, d_sink, iters);
            sb[2] += run_bar<0, 1>(d_ops, d_sink, iters); sb[3] += run_bar<1, 1>(d_ops, d_sink, iters);
        }
        const char* nb[4] = {"serpentine, no barrier", "serpentine + s_barrier per K-step", "serpentine + 40 integer instructions per K-step", "serpentine + s_barrier + 40 integer instructions"};
        for (int o = 0; o < 4; ++o)
            printf("%-8s %-50s mean %.3f ms -> %7.1f TFLOP/s of MFMA\n", pass ? "zeros" : "relu(A)", nb[o], sb[o] / 6, flop / (sb[o] / 6) / 1e9);
    }
    return 0;
}
