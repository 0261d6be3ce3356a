// A whole ResNet bottleneck in ONE launch for the float32-storage x3 types (GPP_F16X3 / GPP_BF16X3) on pre-split maps:
//     y = relu( W3 * relu( W2 (*) relu( W1 * x + b1 ) + b2 ) + b3 + shortcut )
// i.e. the 1x1 conv "branch2a" (C_in -> C, stride 1 or 2), the 3x3 conv "branch2b" (C -> C, stride 1, pad 1) and the 1x1 conv "branch2c"
// (C -> 4C, + shortcut + ReLU) of keras_resnet's bottleneck_2d (the graph instantiated at
// /root/reference/keras_retinanet_3D/models/resnet.py:88-93, blocks consumed at :102), C = 64 or 128.
//
// Why: at float32-sized storage the three layers of a res2 / res3 block move 966 / 590 MB through HBM at B = 8 (x in, a out, a in,
// b out, b in, x in again as the shortcut, y out) for 552 / 280 MB of x-in / y-out; the two intermediate maps are what this kernel
// never writes.  bottleneck_tail_x3_kernel (conv_igemm_impl.h) already keeps `b` in LDS; this one keeps `a` there too.
//
// One workgroup -- C = 128: 8 wavefronts, 144 KB of LDS, one per CU; C = 64: 4 wavefronts, 72 KB, TWO per CU, so that the HBM-bound phases of one
// run beside the matrix-bound phase of the other -- computes a TH x TW tile of output pixels of one image:
//   phase 1  a-tile = relu(W1 * x + b1) on the tile PLUS its one-pixel halo, (TH+2) x (TW+2) = R1 rows of a GEMM with K = C_in:
//            x rows and W1 rows stream through an LDS ring (C = 128: four deep, three K-steps of LDS-DMA in flight -- the x rows are first
//            touches from HBM; C = 64: two deep); the result goes accumulators -> (scale, bias, ReLU, range check, split) -> LDS as the pre-split rows the
//            unfused layer would have stored; halo pixels outside the image are the 3x3 layer's zero padding: zero rows.
//            The halo is recomputed by the neighbouring tiles (R1 / (TH TW) = 1.43 x the 2a work, which is 1/4.5 of the block's).
//   phase 2  b-tile = relu(W2 (*) a-tile + b2): implicit GEMM with the A operand read STRAIGHT from the a-tile in LDS (tap (kh, kw) of
//            output pixel (ty, tx) is halo row (ty + kh)(TW + 2) + tx + kw), only W2 streams (four-deep ring, L2-hot);
//            the result goes to LDS the same way (over the a-tile, which is dead by then).
//   phase 3  y-tile = relu(W3 * b-tile + b3 + shortcut) in 128-channel tiles of W3, shortcut rows prefetched into registers one tile
//            ahead, 16-byte stores of pre-split rows.
// Every output element is summed exactly as the three separate launches sum it -- K-steps ascending (32-channel chunk outer, tap inner),
// hi*wlo, hi*whi, lo*whi per K-step, fma(acc, out_scale, bias), shortcut added as float32(hi) + float32(lo), ReLU, clamp, split -- so the
// result is BIT-IDENTICAL to gpp_conv2d_igemm x 3 (tests/test_block_gpu.py) and the kernel is a pure scheduling choice of the plan.
// Range events (GPP_F16X3) are counted once per stored group, for the pixels the tile owns (not for the recomputed halo).
//
// Measured (profiles/r6, B = 8, same box): res2 identity block 245 us (branch2a + fused tail) -> 205 us, fabric bytes 966 -> ~600 MB; res3
// identity block 185 us (three launches) -> 180 us, read bytes 307 (general form) -> 195 MB; the step +1.7 %.  What holds the C = 128 form at
// parity: one workgroup per CU runs its three phases one after the other -- 17 us of x in at the chip's HBM rate, 20 us of matrix work
// that touches no HBM, 20 us of y out -- and all workgroups of a round do so in lockstep (block_stamps_*.txt); starting half of them late
// (tile + 10000 k), the W2 fragments straight from L2 into registers (no barriers in phase 2: 35 us instead of 20) and one barrier per
// two K-steps were measured and changed nothing or lost.  C = 256 (res4) does not fit: its a-tile alone is 160 KB.
#ifndef GPP_CONV_BLOCK_IMPL_H_
#define GPP_CONV_BLOCK_IMPL_H_

#include "conv_igemm_impl.h"

namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `stages` x PER of this wavefront's vector-memory operations are outstanding (stages = 0, 1, 2: wave-uniform)
template <int PER>
__device__ __forceinline__ void wait_stages(int stages)
{
    if (stages >= 2) wait_vmcnt<2 * PER>();
    else if (stages == 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
}

template <int V> struct IntC { static constexpr int value = V; };

// Diagnostic build only (make variant NAME=blockstamps EXTRA=-DGPP_BLOCK_STAMPS, tools/bench_block.py): wavefront 0 of every workgroup writes the
// 100 MHz real-time counter at the phase boundaries into a buffer handed in through conv1x1_a->zero_page (8 words per workgroup).  No output
// depends on it; the production build contains none of it.
#ifdef GPP_BLOCK_STAMPS
#define GPP_BSTAMP(k)                                                                                          \
    do {                                                                                                       \
        if (wave == 0 && lane == 0) ((unsigned long long*)d1.zero_page)[(int64_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define GPP_BSTAMP(k) do { } while (0)
#endif

// clamp + (optionally) count: what x3_range does, with the counting under the caller's control
template <int DT>
__device__ __forceinline__ void x3_range_if(float (&v)[8], unsigned long long* counter, bool count)
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
            if (count) atomicAdd(counter, 1ull);
#pragma unroll
            for (int e = 0; e < 8; ++e) c[e] = (fabsf(v[e]) <= 3.402823466e38f) ? c[e] : v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c[e];
    }
}

// NWAVES wavefronts per workgroup, RING1 stages in the phase-1 ring: C = 128 runs 8 wavefronts on a four-deep ring (144 KB of LDS: one workgroup per
// CU); C = 64 runs 4 wavefronts on a two-deep ring (72 KB: TWO workgroups per CU, one wavefront of each per SIMD, so that the HBM-bound phases
// of one workgroup run beside the matrix-bound phase of the other)
template <int CMID, int TH, int TW, int NWAVES, int RING1>
struct BlockShape {
    static constexpr int NW = NWAVES;
    static constexpr int HWD = TW + 2;                       // halo width
    static constexpr int R1 = (TH + 2) * HWD;                // halo pixels: rows of the 2a GEMM
    static constexpr int M2 = TH * TW;                       // output pixels of the tile
    static constexpr int KC = CMID / 32;                     // 32-channel slabs of the intermediate tiles
    static constexpr int WN1 = CMID / 32, WM1 = NW / WN1;    // phase 1 / 2: one 32-column pair per wavefront column
    static constexpr int MF1 = R1 / WM1 / 16;
    static constexpr int MB2 = M2 / 16;                      // 16-row blocks of the output tile
    static constexpr int WN2 = WN1, WM2 = WM1;
    static constexpr int MF2 = (MB2 + WM2 - 1) / WM2;
    static constexpr int WN3 = 4, WM3 = NW / WN3;            // phase 3: 128-column tiles of W3
    static constexpr int MF3 = (MB2 + WM3 - 1) / WM3;
    static constexpr int PA = R1 / 8;                        // 8-row LDS-DMA pieces of an activation slab
    static constexpr int A_FULL = PA / NW, A_EXTRA = PA - A_FULL * NW;     // pieces every wavefront issues; wavefronts 0 .. A_EXTRA-1 issue one more
    static constexpr int B_IT1 = CMID / 8 / NW;
    static constexpr int A1_BYTES = R1 * kRowBytes, STAGE1 = A1_BYTES + CMID * kRowBytes;
    static constexpr int S1 = RING1;                         // phase-1 ring depth, a power of two (C = 128: the a-tile's space is free until the phase ends)
    static constexpr int T1_SLAB = R1 * kRowBytes, T1_BYTES = KC * T1_SLAB;
    static constexpr int S2 = 4, STAGE2 = CMID * kRowBytes, B_IT2 = CMID / 8 / NW;
    static constexpr int T2_SLAB = MB2 * 16 * kRowBytes;
    static constexpr int W3_SLAB = 128 * kRowBytes, W3_IT = 128 / 8 / NW;
    static constexpr int REGION_B = (S2 * STAGE2 > KC * W3_SLAB) ? S2 * STAGE2 : KC * W3_SLAB;
    static constexpr int LDS = (S1 * STAGE1 > T1_BYTES + REGION_B) ? S1 * STAGE1 : T1_BYTES + REGION_B;
    static_assert(CMID == 64 || CMID == 128, "bottleneck width");
    static_assert((NW == 4 || NW == 8) && NW % WN1 == 0 && (S1 == 2 || S1 == 4) && CMID % (8 * NW) == 0, "wavefront layout / ring depth");
    static_assert(R1 % (16 * WM1) == 0 && R1 % 8 == 0, "halo rows: whole 16-row blocks per wavefront row, whole 8-row pieces");
    static_assert(M2 % 16 == 0 && HWD % 8 == 0, "tile: whole 16-row blocks; halo width a multiple of 8 (the LDS swizzle key of a tap-shifted row)");
    static_assert(MB2 >= WM2 * MF2 - 1 && MB2 >= WM3 * MF3 - 1, "at most the last wavefront row is one block short");
    static_assert(KC * T2_SLAB <= T1_BYTES && LDS <= 160 * 1024, "LDS budget");
};

// IDENT: an identity block -- the shortcut IS the block's input (same map, stride 1, C_in = 4 C).  Its rows pass through the LDS ring in
// phase 1 anyway: K-step k of branch2a stages channels 32 k .. 32 k + 31 of every halo pixel, which are the shortcut values of output
// channels 32 k .. 32 k + 31.  The two wavefronts that will finish those channels in phase 3 copy their 16-byte pieces from the ring into
// registers there and then, and those channels of the shortcut map are never read a second time (C = 64: all 256 channels, 112 registers per
// lane; C = 128: 384 of 512 -- the fourth output tile's rows are fetched like the general form's: with all of them the compiler spilled).
template <int DT, int CMID, int TH, int TW, bool IDENT, int NWAVES, int RING1>
__global__ __launch_bounds__(64 * NWAVES, 2) void bottleneck_block_x3_kernel(const gpp_conv_desc d1, const gpp_conv_desc d2, const gpp_conv_desc d3,
                                                                      const int tiles_x, const int tiles_y, const int stagger_ticks)
{
    static_assert(kX3<DT>, "x3 types on pre-split maps");
    using S = BlockShape<CMID, TH, TW, NWAVES, RING1>;
    using xh8 = typename X3Half<DT>::vec;
    using half_t = typename X3Half<DT>::half;
    constexpr bool OSCALE = (DT == GPP_F16X3);
    constexpr int NW = S::NW, HWD = S::HWD, R1 = S::R1, M2 = S::M2, KC = S::KC;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4, srow = lane >> 3, gchunk = (lane & 7) ^ srow;
    const gpp_conv_group& G1 = d1.groups[0];
    const gpp_conv_group& G3 = d3.groups[0];
    const int H = G1.H_out, W = G1.W_out;                          // the block's output size (2a's output = 2b's = 2c's)
    unsigned long long* const counter = (unsigned long long*)d3.range_counter;

    // ---- which tile (each XCD owns a contiguous range of tiles: neighbours share their halo columns / rows in that XCD's L2)
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tpi = tiles_x * tiles_y;
    const int b = tile / tpi, trem = tile - b * tpi;
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int y0 = tyi * TH, x0 = txi * TW;                        // first output pixel of the tile

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d1.in, 0, d1.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d1.weight, 0, d1.weight_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d2.weight, 0, d2.weight_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d3.weight, 0, d3.weight_bytes, 0x00020000);

    // Stagger: every workgroup runs an HBM-bound phase (x in), a matrix-bound phase that touches no HBM at all, and another HBM-bound phase
    // (shortcut in, y out).  Workgroups that start together stay in lockstep -- the whole chip waits on HBM, then the whole chip leaves it
    // idle.  Odd tile rows therefore start `stagger_ticks` (100 MHz) late, once: from then on one half of the CUs computes while the other
    // half moves bytes.  Timing only: no result depends on it.
    if (stagger_ticks > 0 && (tyi & 1)) {
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + (unsigned long long)stagger_ticks;
        while (__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(32);
    }
    // rows of the output tile this lane finishes in phase 3 (and their shortcut rows): wavefront layout WM3 x WN3
    const int wm3 = wave / S::WN3, wn3 = wave % S::WN3;
    // a tile that hangs over the bottom edge of the image owns fewer than TH rows of pixels: the 16-row blocks that hold none of them (and the halo
    // rows below them) are skipped in every phase -- wave-uniform counts of the row blocks a wavefront really works on
    const int rows_here = min(TH, H - y0);                         // >= 1
    const int blocks_out = (rows_here * TW + 15) / 16, blocks_halo = ((rows_here + 2) * HWD + 15) / 16;
    const int nrb3 = __builtin_amdgcn_readfirstlane(max(0, min(min(S::MF3, S::MB2 - wm3 * S::MF3), blocks_out - wm3 * S::MF3)));
    int pix3[S::MF3];                                              // pixel index inside the image, -1 = nothing to store
    // (addresses: uniform 64-bit bases of this image + 32-bit byte offsets per lane -- the launcher checks that a map of the block stays below
    // 2 GiB -- instead of a 64-bit pointer per row block: those cost the 4-wavefront form, 7 row blocks per lane, its last registers)
    char* const out_img = (char*)d3.out + ((G3.out_off + (int64_t)b * G3.out_bstride) << 2);
    const char* const res_img = (const char*)d3.residual + ((G3.res_off + (int64_t)b * G3.res_bstride) << 2);
    auto map_off = [&](int pixel, int pitch, int n) { return (int)((((unsigned)pixel * (unsigned)pitch + (unsigned)(n & ~31)) << 2) + ((unsigned)(n & 31) << 1)); };
    int ctr_off[IDENT ? S::MF3 : 1];                               // IDENT: where the pixel's own row sits in a ring stage (hi piece; lo: ^ 64)
#pragma unroll
    for (int i = 0; i < S::MF3; ++i) {
        const int m = (wm3 * S::MF3 + i) * 16 + frow;
        const int ty = m / TW, tx = m - ty * TW;
        const int oy = y0 + ty, ox = x0 + tx;
        const bool ok = i < nrb3 && oy < H && ox < W;
        pix3[i] = ok ? oy * W + ox : -1;
        if constexpr (IDENT) {
            const int r = (ty + 1) * HWD + tx + 1;                 // its halo row
            ctr_off[i] = (i < nrb3 ? r : 0) * kRowBytes + ((fq ^ (r & 7)) << 4);
        }
    }
    // IDENT: the shortcut rows of the first RT 128-channel output tiles live in registers from phase 1 on; the last tile's are fetched from the
    // map like the general form's (with all four in registers -- 128 per lane -- the compiler spilled 30 of them: the budget is 256 beside
    // the accumulators and fragments of three matrix phases; three quarters of the re-read is what there is room for)
    constexpr int N3T = CMID / 32;                                 // output tiles of an identity block (C_out = 4 C)
    constexpr int RT = IDENT ? (CMID == 64 ? N3T : N3T - 1) : 0;
    f32x8 rres[RT > 0 ? RT : 1][S::MF3];                           // [output tile][row block] raw [8 hi][8 lo] bits
    f32x8 rpre[S::MF3];                                            // the prefetched tile of the general form / the last tile of an identity block
    auto prefetch_res = [&](int t) {
        const int n = t * 128 + wn3 * 32 + fq * 8;
#pragma unroll
        for (int i = 0; i < S::MF3; ++i) {
            // (rows past the end: pixel 0 of the image, a valid address whose values are never used)
            const char* p = res_img + map_off(pix3[i] >= 0 ? pix3[i] : 0, d3.res_pitch, n);
            rpre[i].lo = *(const f32x4*)p;
            rpre[i].hi = *(const f32x4*)(p + 64);
        }
    };
    GPP_BSTAMP(0);
    // =============================================================== phase 1: the a-tile (tile + halo) = relu(W1 x + b1)
    constexpr int A_IT = S::A_FULL + (S::A_EXTRA ? 1 : 0);
    const bool extra = S::A_EXTRA && wave < S::A_EXTRA;            // this wavefront issues A_IT activation pieces, not A_FULL
    int a_voff[A_IT];
    {
        const int pitch4 = d1.in_pitch * 4, st = d1.stride;
        const int img = (int)((G1.in_off + (int64_t)b * G1.in_bstride) * 4) + gchunk * 16;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = (i * NW + wave) * 8 + srow;              // halo row this lane stages (piece i * NW + wave)
            a_voff[i] = kOutOfRange;
            if (r < R1) {
                const int hy = r / HWD, hx = r - hy * HWD;
                const int oy = y0 - 1 + hy, ox = x0 - 1 + hx;
                if ((unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W) a_voff[i] = img + (oy * st * G1.W_in + ox * st) * pitch4;
            }
        }
    }
    int w1_voff[S::B_IT1];
#pragma unroll
    for (int i = 0; i < S::B_IT1; ++i) w1_voff[i] = ((wave * S::B_IT1 + i) * 8 + srow) * d1.C_in * 4 + gchunk * 16;
    auto issue1 = [&](int buf, int ks) {
        unsigned char* sa = smem + buf * S::STAGE1;
        const int so = __builtin_amdgcn_readfirstlane(ks * kRowBytes);
#pragma unroll
        for (int i = 0; i < S::A_FULL; ++i) glds16(in_rsrc, a_voff[i], so, sa + (i * NW + wave) * 8 * kRowBytes);
        if constexpr (S::A_EXTRA > 0) {
            if (extra) glds16(in_rsrc, a_voff[A_IT - 1], so, sa + (S::A_FULL * NW + wave) * 8 * kRowBytes);
        }
#pragma unroll
        for (int i = 0; i < S::B_IT1; ++i) glds16(w1_rsrc, w1_voff[i], so, sa + S::A1_BYTES + (wave * S::B_IT1 + i) * 8 * kRowBytes);
    };
    const int wm1 = wave / S::WN1, wn1 = wave % S::WN1;
    const int nrb1 = __builtin_amdgcn_readfirstlane(max(0, min(S::MF1, blocks_halo - wm1 * S::MF1)));
    int a_rd1[2], b_rd1[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int sw = ((h * 4 + fq) ^ (frow & 7)) << 4;
        a_rd1[h] = (wm1 * (R1 / S::WM1) + frow) * kRowBytes + sw;
        b_rd1[h] = S::A1_BYTES + (wn1 * 32 + frow) * kRowBytes + sw;
    }
    f32x4 acc1[S::MF1][2];
#pragma unroll
    for (int i = 0; i < S::MF1; ++i) { acc1[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    auto step1 = [&](int ks) {
        const unsigned char* sb = smem + (ks & (S::S1 - 1)) * S::STAGE1;
        xh8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bh[j] = *(const xh8*)(sb + b_rd1[0] + j * 16 * kRowBytes);
            bl[j] = *(const xh8*)(sb + b_rd1[1] + j * 16 * kRowBytes);
        }
#pragma unroll
        for (int i = 0; i < S::MF1; ++i) {
            if (i >= nrb1) break;
            const xh8 ah = *(const xh8*)(sb + a_rd1[0] + i * 16 * kRowBytes);
            const xh8 al = *(const xh8*)(sb + a_rd1[1] + i * 16 * kRowBytes);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int js = GPP_SERP2(i, j, 2);
                acc1[i][js] = X3Half<DT>::mfma(bl[js], ah, acc1[i][js]);
                acc1[i][js] = X3Half<DT>::mfma(bh[js], ah, acc1[i][js]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) { const int js = GPP_SERP2(i, j, 2); acc1[i][js] = X3Half<DT>::mfma(bh[js], al, acc1[i][js]); }
        }
    };
    if constexpr (IDENT) {
        constexpr int NK1 = 4 * CMID / 32;                         // C_in = 4 C
        int issued = 0;
#pragma unroll
        for (int p = 0; p < S::S1 - 1; ++p) { issue1(issued, issued); ++issued; }
#pragma unroll
        for (int ks = 0; ks < NK1; ++ks) {
            const int ahead = (NK1 - 1 - ks) < (S::S1 - 2) ? (NK1 - 1 - ks) : (S::S1 - 2);
            if (extra) wait_stages<A_IT + S::B_IT1>(ahead);
            else wait_stages<S::A_FULL + S::B_IT1>(ahead);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (issued < NK1) { issue1(issued & (S::S1 - 1), issued); ++issued; }
            if ((ks >> 2) < RT && wn3 == (ks & 3)) {               // this stage holds the shortcut values of this wavefront's channels of output tile ks / 4
                const unsigned char* sb = smem + (ks & (S::S1 - 1)) * S::STAGE1;
#pragma unroll
                for (int i = 0; i < S::MF3; ++i) {
                    if (i >= nrb3) break;
                    rres[ks >> 2][i].lo = *(const f32x4*)(sb + ctr_off[i]);
                    rres[ks >> 2][i].hi = *(const f32x4*)(sb + (ctr_off[i] ^ 64));
                }
            }
            step1(ks);
        }
    } else {
        const int nk1 = d1.C_in / 32;
        int issued = 0;
#pragma unroll
        for (int p = 0; p < S::S1 - 1; ++p)
            if (issued < nk1) { issue1(issued, issued); ++issued; }
        for (int ks = 0; ks < nk1; ++ks) {
            const int ahead = issued - ks - 1;                     // stages younger than ks in flight
            if (extra) wait_stages<A_IT + S::B_IT1>(ahead);
            else wait_stages<S::A_FULL + S::B_IT1>(ahead);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (issued < nk1) { issue1(issued & (S::S1 - 1), issued); ++issued; }
            step1(ks);
        }
    }
    GPP_BSTAMP(1);
    // ---- hand-over 1: everyone is done with the ring; the shortcut rows of output tile 0 and the first W2 stages start moving while the
    // a-tile is written (scale, bias, ReLU, range check, split) as pre-split rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    constexpr int nk2 = 9 * KC;
    int w2_voff[S::B_IT2];
#pragma unroll
    for (int i = 0; i < S::B_IT2; ++i) w2_voff[i] = ((wave * S::B_IT2 + i) * 8 + srow) * (9 * CMID * 4) + gchunk * 16;
    auto issue2 = [&](int buf, int ks) {
        const int so = __builtin_amdgcn_readfirstlane(ks * kRowBytes);
#pragma unroll
        for (int i = 0; i < S::B_IT2; ++i)
            glds16(w2_rsrc, w2_voff[i], so, smem + S::T1_BYTES + buf * S::STAGE2 + (wave * S::B_IT2 + i) * 8 * kRowBytes);
    };
    int issued2 = 0;
#pragma unroll
    for (int p = 0; p < S::S2 - 1; ++p) { issue2(p, p); ++issued2; }

    {
        const int n = wn1 * 32 + fq * 8;                           // 8 consecutive channels of the a-tile
        float bias_v[8], scale_v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bias_v[e] = d1.bias ? d1.bias[n + e] : 0.0f;
            scale_v[e] = (OSCALE && d1.out_scale) ? d1.out_scale[n + e] : 1.0f;
        }
        const int piece = (n & 31) >> 3;
        unsigned char* slab = smem + (n >> 5) * S::T1_SLAB;
#pragma unroll
        for (int i = 0; i < S::MF1; ++i) {
            if (i >= nrb1) break;
            const int r = wm1 * (R1 / S::WM1) + i * 16 + frow;
            const int hy = r / HWD, hx = r - hy * HWD;
            const int oy = y0 - 1 + hy, ox = x0 - 1 + hx;
            const bool inside = (unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W;
            const bool owned = inside && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = e < 4 ? acc1[i][0][e] : acc1[i][1][e - 4];
                float v;
                if constexpr (OSCALE) v = __builtin_fmaf(a, scale_v[e], bias_v[e]);
                else v = a + bias_v[e];
                if (d1.relu) v = fmaxf(v, 0.0f);
                v8[e] = inside ? v : 0.0f;                         // outside the image: the 3x3 layer's zero padding
            }
            x3_range_if<DT>(v8, counter, owned);
            xh8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[e] = (half_t)v8[e];
                l[e] = (half_t)(v8[e] - (float)h[e]);
            }
            unsigned char* row = slab + r * kRowBytes;
            *(xh8*)(row + ((piece ^ (r & 7)) << 4)) = h;
            *(xh8*)(row + (((4 + piece) ^ (r & 7)) << 4)) = l;
        }
    }

    GPP_BSTAMP(2);
    // =============================================================== phase 2: the b-tile = relu(W2 (*) a-tile + b2)
    const int wm2 = wm1, wn2 = wn1;
    const int nrb2 = __builtin_amdgcn_readfirstlane(max(0, min(min(S::MF2, S::MB2 - wm2 * S::MF2), blocks_out - wm2 * S::MF2)));
    int t1off[S::MF2][3];                                          // [row block][kw]: byte offset inside a slab of the hi piece of the tap's row, tap row 0 (lo: ^ 64)
#pragma unroll
    for (int i = 0; i < S::MF2; ++i) {
        const int m = (wm2 * S::MF2 + i) * 16 + frow;
        const int ty = m / TW, tx = m - ty * TW;
        const int rr = ty * HWD + tx;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) t1off[i][kw] = (rr + kw) * kRowBytes + ((fq ^ ((rr + kw) & 7)) << 4);
    }
    int b_rd2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) b_rd2[h] = S::T1_BYTES + (wn2 * 32 + frow) * kRowBytes + (((h * 4 + fq) ^ (frow & 7)) << 4);
    f32x4 acc2[S::MF2][2];
#pragma unroll
    for (int i = 0; i < S::MF2; ++i) { acc2[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    auto phase2 = [&]() {
        constexpr int NRB = S::MF2;
        for (int cc = 0; cc < KC; ++cc) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                const int ks = cc * 9 + tap;
                wait_stages<S::B_IT2>(issued2 - ks - 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (issued2 < nk2) { issue2(issued2 & (S::S2 - 1), issued2); ++issued2; }
                const unsigned char* sa = smem + cc * S::T1_SLAB + kh * HWD * kRowBytes;
                const unsigned char* sb = smem + (ks & (S::S2 - 1)) * S::STAGE2;
                // fragments one row block at a time (the shortcut rows of an identity block sit in 100+ registers by now): per accumulator the
                // three products keep their order
                xh8 bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = *(const xh8*)(sb + b_rd2[0] + j * 16 * kRowBytes);
                    bl[j] = *(const xh8*)(sb + b_rd2[1] + j * 16 * kRowBytes);
                }
#pragma unroll
                for (int i = 0; i < NRB; ++i) {
                    if (i >= nrb2) break;
                    const xh8 ah = *(const xh8*)(sa + t1off[i][kw]);
                    const xh8 al = *(const xh8*)(sa + (t1off[i][kw] ^ 64));
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int js = GPP_SERP2(i, j, 2);
                        acc2[i][js] = X3Half<DT>::mfma(bl[js], ah, acc2[i][js]);
                        acc2[i][js] = X3Half<DT>::mfma(bh[js], ah, acc2[i][js]);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) { const int js = GPP_SERP2(i, j, 2); acc2[i][js] = X3Half<DT>::mfma(bh[js], al, acc2[i][js]); }
                }
            }
        }
    };
    // (the a-tile is visible to everyone behind the first barrier of the loop)
    phase2();

    GPP_BSTAMP(3);
    // ---- hand-over 2: everyone is done with the a-tile and the W2 ring; W3 tile 0 streams in while the b-tile is written over the a-tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int w3_voff[S::W3_IT];
#pragma unroll
    for (int i = 0; i < S::W3_IT; ++i) w3_voff[i] = ((wave * S::W3_IT + i) * 8 + srow) * CMID * 4 + gchunk * 16;
    auto stage_w3 = [&](int t) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int i = 0; i < S::W3_IT; ++i)
                glds16(w3_rsrc, w3_voff[i], t * 128 * CMID * 4 + kc * kRowBytes,
                       smem + S::T1_BYTES + kc * S::W3_SLAB + (wave * S::W3_IT + i) * 8 * kRowBytes);
    };
    stage_w3(0);
    if constexpr (!IDENT) prefetch_res(0);                         // the shortcut rows of output tile 0: in flight under the b-tile's epilogue
    {
        const int n = wn2 * 32 + fq * 8;
        float bias_v[8], scale_v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bias_v[e] = d2.bias ? d2.bias[n + e] : 0.0f;
            scale_v[e] = (OSCALE && d2.out_scale) ? d2.out_scale[n + e] : 1.0f;
        }
        const int piece = (n & 31) >> 3;
        unsigned char* slab = smem + (n >> 5) * S::T2_SLAB;
#pragma unroll
        for (int i = 0; i < S::MF2; ++i) {
            if (i < nrb2) {
                const int m = (wm2 * S::MF2 + i) * 16 + frow;
                const int ty = m / TW, tx = m - ty * TW;
                const bool owned = y0 + ty < H && x0 + tx < W;     // (rows of an edge tile that lie outside the image: computed, never stored, not counted)
                float v8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = e < 4 ? acc2[i][0][e] : acc2[i][1][e - 4];
                    float v;
                    if constexpr (OSCALE) v = __builtin_fmaf(a, scale_v[e], bias_v[e]);
                    else v = a + bias_v[e];
                    if (d2.relu) v = fmaxf(v, 0.0f);
                    v8[e] = v;
                }
                x3_range_if<DT>(v8, counter, owned);
                xh8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    h[e] = (half_t)v8[e];
                    l[e] = (half_t)(v8[e] - (float)h[e]);
                }
                unsigned char* row = slab + m * kRowBytes;
                *(xh8*)(row + ((piece ^ (m & 7)) << 4)) = h;
                *(xh8*)(row + (((4 + piece) ^ (m & 7)) << 4)) = l;
            }
        }
    }

    GPP_BSTAMP(4);
    // =============================================================== phase 3: y-tile = relu(W3 b-tile + b3 + shortcut), 128 channels at a time
    int a_rd3[2], b_rd3[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int sw = ((h * 4 + fq) ^ (frow & 7)) << 4;
        a_rd3[h] = (wm3 * S::MF3 * 16 + frow) * kRowBytes + sw;
        b_rd3[h] = S::T1_BYTES + (wn3 * 32 + frow) * kRowBytes + sw;
    }
    const int n3_tiles = IDENT ? N3T : d3.C_out / 128;
    auto phase3 = [&]() {
        constexpr int NRB = S::MF3;
#pragma unroll
        for (int t = 0; t < (IDENT ? N3T : n3_tiles); ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // W3 tile t (and, for t = 0, the b-tile) is in LDS
            asm volatile("" ::: "memory");
            f32x4 acc3[NRB][2];
#pragma unroll
            for (int i = 0; i < NRB; ++i) { acc3[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc3[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                xh8 bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = *(const xh8*)(smem + kc * S::W3_SLAB + b_rd3[0] + j * 16 * kRowBytes);
                    bl[j] = *(const xh8*)(smem + kc * S::W3_SLAB + b_rd3[1] + j * 16 * kRowBytes);
                }
#pragma unroll
                for (int i = 0; i < NRB; ++i) {
                    if (i >= nrb3) break;
                    const xh8 ah = *(const xh8*)(smem + kc * S::T2_SLAB + a_rd3[0] + i * 16 * kRowBytes);
                    const xh8 al = *(const xh8*)(smem + kc * S::T2_SLAB + a_rd3[1] + i * 16 * kRowBytes);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int js = GPP_SERP2(i, j, 2);
                        acc3[i][js] = X3Half<DT>::mfma(bl[js], ah, acc3[i][js]);
                        acc3[i][js] = X3Half<DT>::mfma(bh[js], ah, acc3[i][js]);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) { const int js = GPP_SERP2(i, j, 2); acc3[i][js] = X3Half<DT>::mfma(bh[js], al, acc3[i][js]); }
                }
            }
            if (t + 1 < n3_tiles) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // everyone has read W3 tile t
                asm volatile("" ::: "memory");
                stage_w3(t + 1);                                   // streams in under the epilogue below
            }
            const int n = t * 128 + wn3 * 32 + fq * 8;
            float bias_v[8], scale_v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bias_v[e] = d3.bias ? d3.bias[n + e] : 0.0f;
                scale_v[e] = (OSCALE && d3.out_scale) ? d3.out_scale[n + e] : 1.0f;
            }
            // one row block at a time (registers): accumulator -> scale, bias -> + shortcut -> the next tile's shortcut rows of this row block are
            // requested (general form; before this block's stores go out) -> ReLU, range check, split, store
            const bool fetch_next = t + 1 < n3_tiles && (!IDENT || t + 1 >= RT);
            const int n_next = (t + 1) * 128 + wn3 * 32 + fq * 8;
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
                if (i >= nrb3) break;
                float outv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = e < 4 ? acc3[i][0][e] : acc3[i][1][e - 4];
                    if constexpr (OSCALE) outv[e] = __builtin_fmaf(a, scale_v[e], bias_v[e]);
                    else outv[e] = a + bias_v[e];
                }
                float r[8];
                if (IDENT && t < RT) x3_unpack<DT>(rres[t < RT ? t : 0][i].lo, rres[t < RT ? t : 0][i].hi, r);
                else x3_unpack<DT>(rpre[i].lo, rpre[i].hi, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) outv[e] += r[e];
                const int pixel = pix3[i] >= 0 ? pix3[i] : 0;      // (rows past the end: pixel 0 of the image, a valid address whose values are never used)
                if (fetch_next) {
                    const char* p = res_img + map_off(pixel, d3.res_pitch, n_next);
                    rpre[i].lo = *(const f32x4*)p;
                    rpre[i].hi = *(const f32x4*)(p + 64);
                }
                if (pix3[i] >= 0) {                                // what finish8_pre does for a pre-split output: ReLU, range check, split, two 16-byte stores
                    if (d3.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) outv[e] = fmaxf(outv[e], 0.0f);
                    }
                    x3_range<DT>(outv, counter);
                    xh8 h, l;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        h[e] = (half_t)outv[e];
                        l[e] = (half_t)(outv[e] - (float)h[e]);
                    }
                    char* q = out_img + map_off(pixel, d3.out_pitch, n);
                    *(xh8*)q = h;
                    *(xh8*)(q + 64) = l;
                }
            }
        }
    };
    phase3();
    GPP_BSTAMP(5);
#ifdef GPP_BLOCK_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GPP_BSTAMP(6);
#endif
}

template <int DT, int CMID, int TH, int TW, bool IDENT, int NWAVES, int RING1>
int launch_block_x3(gpp_conv_desc& d1, gpp_conv_desc& d2, gpp_conv_desc& d3, int stagger_us, hipStream_t st)
{
    using S = BlockShape<CMID, TH, TW, NWAVES, RING1>;
    static DeviceOnce once;
    auto kernel = bottleneck_block_x3_kernel<DT, CMID, TH, TW, IDENT, NWAVES, RING1>;
    int rc = once.configure(kernel, S::LDS);
    if (rc != GPP_OK) return rc;
    const gpp_conv_group& G = d1.groups[0];
    const int64_t in_elems = G.in_off + (int64_t)(d1.batch - 1) * G.in_bstride + ((int64_t)G.H_in * G.W_in - 1) * d1.in_pitch + d1.C_in;
    const int64_t w1_bytes = (int64_t)d1.weight_rows * d1.C_in * 4, w2_bytes = (int64_t)d2.weight_rows * 9 * CMID * 4,
                  w3_bytes = (int64_t)d3.weight_rows * CMID * 4;
    if (G.in_off < 0 || G.in_bstride < 0 || in_elems * 4 >= (1LL << 31) || w1_bytes >= (1LL << 31) || w2_bytes >= (1LL << 31) || w3_bytes >= (1LL << 31))
        return GPP_ERR_UNSUPPORTED;
    const gpp_conv_group& G3 = d3.groups[0];
    const int64_t px = (int64_t)G3.H_out * G3.W_out;
    if (px * d3.out_pitch * 4 >= (1LL << 31) || px * d3.res_pitch * 4 >= (1LL << 31)) return GPP_ERR_UNSUPPORTED;     // 32-bit byte offsets inside one image
    d1.in_bytes = (int32_t)(in_elems * 4);
    d1.weight_bytes = (int32_t)w1_bytes;
    d2.weight_bytes = (int32_t)w2_bytes;
    d3.weight_bytes = (int32_t)w3_bytes;
    const int tiles_x = (G.W_out + TW - 1) / TW, tiles_y = (G.H_out + TH - 1) / TH;
    const int64_t grid = (int64_t)d1.batch * tiles_x * tiles_y;
    if (grid >= (1LL << 31)) return GPP_ERR_UNSUPPORTED;
    kernel<<<dim3((unsigned)grid), dim3(64 * NWAVES), S::LDS, st>>>(d1, d2, d3, tiles_x, tiles_y, stagger_us * 100);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

// tile = TH * 100 + TW (0 = the default of the width)
template <int DT>
int dispatch_block_x3(gpp_conv_desc& d1, gpp_conv_desc& d2, gpp_conv_desc& d3, int tile, hipStream_t st)
{
    const int stagger_us = tile / 10000;                      // (experiments: tile + 10000 * microseconds the odd tile rows start late)
    tile %= 10000;
    if (stagger_us < 0 || stagger_us > 200) return GPP_ERR_BAD_ARG;
    // an identity block: the shortcut is the very map branch2a reads (same pointer, offsets and pitches), stride 1, C_in = 4 C -- its rows are
    // taken from the LDS ring (IDENT); tile + 1000 forces the general form (the shortcut read from its map) on such a block (A/B, tests)
    const gpp_conv_group &G1 = d1.groups[0], &G3 = d3.groups[0];
    const bool general = tile >= 1000;
    tile %= 1000;
    const bool ident = !general && d3.residual == d1.in && G3.res_off == G1.in_off && G3.res_bstride == G1.in_bstride && d3.res_pitch == d1.in_pitch &&
                       d1.stride == 1 && d1.C_in == 4 * d2.C_in && d3.C_out == d1.C_in;
    if (d2.C_in == 128) {
        switch (tile) {
            case 0:
            case 814: return ident ? launch_block_x3<DT, 128, 8, 14, true, 8, 4>(d1, d2, d3, stagger_us, st)
                                   : launch_block_x3<DT, 128, 8, 14, false, 8, 4>(d1, d2, d3, stagger_us, st);
            default: return GPP_ERR_BAD_ARG;
        }
    }
    if (d2.C_in == 64) {
        switch (tile) {
            case 0:
            case 814: return ident ? launch_block_x3<DT, 64, 8, 14, true, 4, 2>(d1, d2, d3, stagger_us, st)
                                   : launch_block_x3<DT, 64, 8, 14, false, 4, 2>(d1, d2, d3, stagger_us, st);
            default: return GPP_ERR_BAD_ARG;
        }
    }
    return GPP_ERR_UNSUPPORTED;
}

}  // namespace

#endif  // GPP_CONV_BLOCK_IMPL_H_
