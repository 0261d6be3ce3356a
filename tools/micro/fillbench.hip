// Micro-benchmark: how fast can one CU pull 128-byte tile rows from L2 into LDS with LDS-DMA, as a function of the
// ring depth (K-steps in flight), the workgroups sharing the CU and the synchronisation per K-step?  The access pattern is the
// small-M 1x1 convolution's (conv_igemm_impl.h): per K-step ROWS_A activation rows private to the workgroup and ROWS_B weight
// rows shared by all workgroups, 128 contiguous bytes each at a pitch of ksteps*128 bytes, 8 rows per wave-instruction.
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/fillbench.hip -o gpurun_out/fillbench      run: gpurun_out/fillbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int AUX = 0>
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, void* lds)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voffset, soffset, 0, AUX);
}

// MODE 0: vmcnt wait + barrier per K-step (the conv kernel's loop); MODE 1: no barrier (each wave waits for its own pieces only)
// WORK: MFMAs per wave and K-step (with 24/64 * WORK ds_read_b128 in front of them), 0 = fills only
// SHARED_A (round 6): every workgroup reads the SAME activation rows (an L2-resident operand) instead of rows of its own (first touches from
// HBM / the Infinity Cache); AUX: cache policy of the activation loads (2 = nt)
template <int NW, int ROWS_A, int ROWS_B, int STAGES, int MODE, int WORK, bool SHARED_A = false, int AUX = 0>
__global__ __launch_bounds__(64 * NW) void fill_kernel(const char* A, const char* B, int a_bytes, int b_bytes, int ksteps, float* sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_IT = ROWS_A / 8 / NW, B_IT = ROWS_B / 8 / NW, PER = A_IT + B_IT, STAGE = (ROWS_A + ROWS_B) * 128, PF = STAGES - 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int srow = lane >> 3, chunk = (lane & 7) ^ srow;
    const int pitch = ksteps * 128;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    int a_voff[A_IT], b_voff[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_voff[i] = (((SHARED_A ? 0 : blockIdx.x) * ROWS_A + (wave * A_IT + i) * 8 + srow)) * pitch + chunk * 16;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) b_voff[i] = ((wave * B_IT + i) * 8 + srow) * pitch + chunk * 16;
    auto stage = [&](int buf, int ks) {
        unsigned char* sa = smem + buf * STAGE + wave * A_IT * 1024;
        unsigned char* sb = smem + buf * STAGE + ROWS_A * 128 + wave * B_IT * 1024;
        const int so = __builtin_amdgcn_readfirstlane(ks * 128);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16<AUX>(ra, a_voff[i], so, sa + i * 1024);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16(rb, b_voff[i], so, sb + i * 1024);
    };
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int issued = 0, ibuf = 0, cbuf = 0;
    for (int p = 0; p < PF; ++p)
        if (issued < ksteps) { stage(ibuf, issued); ++issued; if (++ibuf == STAGES) ibuf = 0; }
    for (int ks = 0; ks < ksteps; ++ks) {
        if (PF > 1 && issued - ks - 1 >= PF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (issued < ksteps) { stage(ibuf, issued); ++issued; if (++ibuf == STAGES) ibuf = 0; }
        if (WORK > 0) {
            const unsigned char* sb = smem + cbuf * STAGE + (lane & 15) * 128 + ((lane >> 4) << 4);
            constexpr int NR = (WORK * 24 + 63) / 64;
            bf16x8 f[NR < 2 ? 2 : NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) f[i] = *(const bf16x8*)(sb + ((i * 2048) % (STAGE - 2048)));
#pragma unroll
            for (int i = 0; i < WORK; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i % NR], f[(i + 1) % NR], acc[i & 7], 0, 0, 0);
        }
        if (++cbuf == STAGES) cbuf = 0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (WORK == 0) s = ((const float*)smem)[threadIdx.x];
    if (s == 123.456f) sink[0] = s;
}

struct Bufs { char *A, *B; float* sink; size_t a_bytes, b_bytes; };

template <int NW, int ROWS_A, int ROWS_B, int STAGES, int MODE, int WORK, bool SHARED_A = false, int AUX = 0>
void run(const Bufs& bf, int wgs, int ksteps, int lds_pad, const char* note)
{
    auto k = fill_kernel<NW, ROWS_A, ROWS_B, STAGES, MODE, WORK, SHARED_A, AUX>;
    const int lds = STAGES * (ROWS_A + ROWS_B) * 128 + lds_pad;         // lds_pad forces fewer workgroups per CU
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const size_t need_a = (size_t)wgs * ROWS_A * ksteps * 128, need_b = (size_t)ROWS_B * ksteps * 128;
    if (need_a > bf.a_bytes || need_b > bf.b_bytes || need_a >= (1u << 31)) { printf("skip (buffers)\n"); return; }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k<<<wgs, 64 * NW, lds>>>(bf.A, bf.B, (int)need_a, (int)need_b, ksteps, bf.sink);
    CHECK(hipDeviceSynchronize());
    const int iters = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) k<<<wgs, 64 * NW, lds>>>(bf.A, bf.B, (int)need_a, (int)need_b, ksteps, bf.sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters;
    const double bytes = (double)wgs * ksteps * (ROWS_A + ROWS_B) * 128;
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64 * NW, lds));
    printf("%-34s NW %d rows %3d+%3d stages %d mode %d work %2d | wgs %4d x %2d ksteps, %d wg/CU fit | %7.1f us  %6.1f GB/s per CU  %5.1f TB/s chip | K-step %5.2f us\n",
           note, NW, ROWS_A, ROWS_B, STAGES, MODE, WORK, wgs, ksteps, per_cu, us, bytes / us / 1e3 / 256.0, bytes / us / 1e6,
           us / ksteps / ((wgs + 256 * per_cu - 1) / (256 * per_cu)));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main()
{
    Bufs bf;
    bf.a_bytes = 1ull << 30; bf.b_bytes = 64ull << 20;
    CHECK(hipMalloc(&bf.A, bf.a_bytes)); CHECK(hipMalloc(&bf.B, bf.b_bytes)); CHECK(hipMalloc(&bf.sink, 1024));
    CHECK(hipMemset(bf.A, 1, bf.a_bytes)); CHECK(hipMemset(bf.B, 1, bf.b_bytes));
    // ---- res4 2a shape: 16 K-steps, 96 + 128 rows, 364 workgroups (what the network runs today: 2 stages, barrier)
    printf("== res4 2a (1x1 1024->256, M 17472): 16 K-steps\n");
    run<4, 96, 128, 2, 0, 0>(bf, 364, 16, 0, "today, fills only");
    run<4, 96, 128, 2, 0, 24>(bf, 364, 16, 0, "today, + reads + MFMA");
    run<4, 96, 128, 2, 1, 0>(bf, 364, 16, 0, "no barrier, fills only");
    run<4, 96, 128, 3, 0, 24>(bf, 364, 16, 0, "3 stages");
    run<4, 96, 128, 4, 0, 24>(bf, 364, 16, 0, "4 stages");
    run<4, 96, 128, 5, 0, 24>(bf, 364, 16, 0, "5 stages");
    run<4, 160, 128, 2, 0, 40>(bf, 220, 16, 0, "160x128 one round");
    run<4, 160, 128, 3, 0, 40>(bf, 220, 16, 0, "160x128 3 stages");
    run<4, 160, 128, 4, 0, 40>(bf, 220, 16, 0, "160x128 4 stages");
    run<8, 128, 256, 2, 0, 32>(bf, 137, 16, 0, "128x256 8 waves");
    run<8, 128, 256, 3, 0, 32>(bf, 137, 16, 0, "128x256 8 waves 3 stages");
    // ---- steady state: many K-steps, exactly N workgroups per CU: rate against bytes in flight
    printf("== steady state, 64 K-steps, workgroups = 256 x (fit per CU)\n");
    run<4, 96, 128, 2, 0, 0>(bf, 512, 64, 0, "2 wg/CU 2 stages");
    run<4, 96, 128, 2, 0, 0>(bf, 256, 64, 60000, "1 wg/CU 2 stages");
    run<4, 96, 128, 3, 0, 0>(bf, 256, 64, 0, "1 wg/CU 3 stages");
    run<4, 96, 128, 4, 0, 0>(bf, 256, 64, 0, "1 wg/CU 4 stages");
    run<4, 96, 128, 5, 0, 0>(bf, 256, 64, 0, "1 wg/CU 5 stages");
    run<4, 96, 128, 5, 1, 0>(bf, 256, 64, 0, "1 wg/CU 5 stages no barrier");
    run<4, 64, 64, 2, 0, 0>(bf, 1024, 64, 0, "4 wg/CU 64+64 2 stages");
    run<4, 64, 64, 4, 0, 0>(bf, 512, 64, 0, "2 wg/CU 64+64 4 stages");
    run<4, 64, 64, 8, 0, 0>(bf, 256, 64, 0, "1 wg/CU 64+64 8 stages");
    run<4, 64, 64, 8, 1, 0>(bf, 256, 64, 0, "1 wg/CU 64+64 8 stages no barrier");
    run<8, 128, 128, 4, 0, 0>(bf, 256, 64, 0, "1 wg/CU 8 waves 128+128 4 stages");
    run<8, 128, 128, 4, 1, 0>(bf, 256, 64, 0, "same, no barrier");
    run<16, 128, 128, 4, 1, 0>(bf, 256, 64, 0, "16 waves, no barrier");
    run<4, 96, 128, 2, 0, 24>(bf, 512, 64, 0, "2 wg/CU 2 stages + work");
    run<4, 96, 128, 4, 0, 24>(bf, 256, 64, 0, "1 wg/CU 4 stages + work");
    run<4, 96, 128, 5, 0, 24>(bf, 256, 64, 0, "1 wg/CU 5 stages + work");
    run<8, 96 * 2, 128, 3, 0, 24>(bf, 256, 64, 0, "1 wg/CU 8 waves 192+128 3 stages + work");
    // ---- round 6: where the bytes come from.  The "56 GB/s per CU" of the lines above is a MIX: the activation rows are private to a workgroup (201 MB per
    // launch: first touches from HBM / the Infinity Cache), the weight rows are L2-hot.  Same loop with every operand L2-resident, with 2 / 4 / 8 / 16 loader
    // wavefronts, and with the nt policy on the activation loads (MI355X_MICROARCH.md "ldsdma-fill" / "nt-weights" rows)
    printf("== round 6: operand residency and cache policy, 64 K-steps, 1 workgroup per CU, 96 + 128 rows, fills only\n");
    run<4, 96, 128, 4, 0, 0, false, 0>(bf, 256, 64, 0, "A private (HBM side), default policy");
    run<4, 96, 128, 4, 0, 0, false, 2>(bf, 256, 64, 0, "A private (HBM side), nt");
    run<4, 96, 128, 4, 0, 0, true, 0>(bf, 256, 64, 0, "A shared (L2-resident), default");
    run<4, 96, 128, 4, 1, 0, true, 0>(bf, 256, 64, 0, "A shared, no barrier");
    run<2, 96, 128, 4, 1, 0, true, 0>(bf, 256, 64, 0, "A shared, 2 loader wavefronts");
    run<8, 128, 128, 4, 1, 0, true, 0>(bf, 256, 64, 0, "A shared, 8 loader wavefronts");
    run<16, 128, 128, 4, 1, 0, true, 0>(bf, 256, 64, 0, "A shared, 16 loader wavefronts");
    run<4, 96, 128, 4, 0, 0, true, 0>(bf, 32, 64, 0, "A shared, only 32 CUs busy");
    run<4, 96, 128, 4, 0, 0, false, 0>(bf, 32, 64, 0, "A private, only 32 CUs busy");
    run<4, 96, 128, 4, 0, 0, false, 2>(bf, 32, 64, 0, "A private nt, only 32 CUs busy");
    // ---- does the K-step time follow the BYTES of a K-step?  (a 3x3 layer re-reads its activation rows once per tap: sharing a staged
    // patch between taps would cut the activation bytes by 2/3 -- worth building only if the loop is bandwidth- and not latency-bound)
    printf("== res4 2b shape (36 K-steps, 364 workgroups, 2 stages, barrier, + work): activation rows per K-step 96 / 64 / 32, weights 128\n");
    run<4, 96, 128, 2, 0, 24>(bf, 364, 36, 0, "A 96 rows (today)");
    run<4, 64, 128, 2, 0, 24>(bf, 364, 36, 0, "A 64 rows");
    run<4, 32, 128, 2, 0, 24>(bf, 364, 36, 0, "A 32 rows (= patch shared by the 9 taps)");
    run<4, 96, 64, 2, 0, 24>(bf, 364, 36, 0, "A 96 rows, weights 64");
    run<4, 32, 32, 2, 0, 24>(bf, 364, 36, 0, "A 32 rows, weights 32");
    printf("== res3 tail phase 1 shape (18 K-steps, 418 workgroups, 160 + 128 rows)\n");
    run<4, 160, 128, 2, 0, 40>(bf, 418, 18, 0, "A 160 rows (today)");
    run<4, 64, 128, 2, 0, 40>(bf, 418, 18, 0, "A 64 rows");
    // ---- one more K-step in flight at an UNCHANGED number of workgroups per CU (smaller stages, three of them)
    printf("== depth at equal occupancy: 36 K-steps, 364 workgroups, 2 workgroups per CU in every line\n");
    run<4, 64, 128, 2, 0, 24>(bf, 364, 36, 24576, "64+128, 2 stages (padded to 2 wg/CU)");
    run<4, 64, 128, 3, 0, 24>(bf, 364, 36, 0, "64+128, 3 stages");
    run<4, 32, 128, 2, 0, 24>(bf, 364, 36, 40960, "32+128, 2 stages (padded to 2 wg/CU)");
    run<4, 32, 128, 3, 0, 24>(bf, 364, 36, 16384, "32+128, 3 stages (padded to 2 wg/CU)");
    run<4, 32, 128, 4, 0, 24>(bf, 364, 36, 0, "32+128, 4 stages");
    return 0;
}
