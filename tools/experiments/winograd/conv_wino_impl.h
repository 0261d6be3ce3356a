// Winograd F(2, 3) along W for the 3 x 3 / stride-1 / pad-1 layers of the head towers, in the GPP_F16X3 arithmetic (float32-sized pre-split maps,
// three IEEE-half matrix products per float32 product, float32 accumulation): 4 position GEMMs per output PAIR instead of the 6 products of the
// direct form, the three kernel rows staying a direct K loop.  Numerics: oracle/fastconv_numerics.py (inside utils/ledger.REFERENCE_BARS with
// margin); what the K-step gains: tools/micro/kstep_wino.hip, profiles/r5/fastconv_feasibility.md.
//
// Replaces, for the layers the plan builder gives it (models/retinanet.py, GPP_WINO), the same Conv2D + bias + ReLU nodes of
//   /root/reference/keras_retinanet_3D/models/retinanet.py:100-107 (regression tower)
// that conv_igemm_impl.h computes directly.  Included by conv_igemm_f16x3.hip only (it shares that unit's range counter and split helpers).
//
// Two launches per layer:
//   wino_transform_kernel   d (B, pixels, pitch) pre-split  ->  V (B, pairs, 4, C) pre-split:  for output pair tx of image row y, with
//                           d_j = in[y][2 tx - 1 + j] (zero outside the row):  V0 = d0 - d2,  V1 = d1 + d2,  V2 = d2 - d1,  V3 = d1 - d3
//                           (float32, each split into halves again -- and range-checked -- as any stored activation)
//   wino_conv_kernel        per tile of 192 pairs x 128 output channels: for position p = 0..3 the K loop over (32-channel chunk, kernel row) of
//                           V_p x U_p (U_p = sum_kw G[p][kw] g[kh][kw], packed position-major with its own power-of-two scale per output
//                           channel) into M; between positions M x 2^-k(p, n) is folded into Y0 = (M0 + M1) + M2, Y1 = (M1 - M2) - M3; then bias,
//                           ReLU, range check, split, store Y0 -> pixel 2 tx, Y1 -> pixel 2 tx + 1 of a pre-split map.
//   8 wavefronts 4 x 2, wavefront tile 48 pairs x 64 channels: 12 accumulators x 3 sets = 144 registers; 40 KB stages, two in flight;
//   the three-phase x3 K-step of conv_igemm_impl.h (hi.wlo | hi.whi | barrier | lo.whi + the LDS-DMA of the stage after next), MFMAs in
//   serpentine order, accumulation in place; padding rows through the buffer descriptor's range check.
#ifndef GPP_CONV_WINO_IMPL_H_
#define GPP_CONV_WINO_IMPL_H_

#include "conv_igemm_impl.h"

namespace {

constexpr int kWinoPairs = 192, kWinoCols = 128;                  // per-position tile
constexpr int kWinoStage = (kWinoPairs + kWinoCols) * kRowBytes;  // 40 KB
constexpr int kWinoPieces = (kWinoPairs + kWinoCols) / 8 / 8;     // 1 KB (8-row) LDS-DMA pieces per wavefront and stage: 5
constexpr int kWinoAPieces = kWinoPairs / 8;                      // pieces 0..23 are activation rows, 24..39 weight rows
// stages of the LDS ring.  The DMA of a stage goes out right behind the barrier of the step that has finished with its buffer and is first read
// kWinoStages - 1 steps later: with two stages one K-step (~0.9 us) of latency tolerance, with three two.  On an LDS / L2-resident image the
// third stage buys nothing (tools/micro/kstep_wino.hip); on the real layer the activation rows of V are first touches from HBM.
#ifndef GPP_WINO_STAGES
#define GPP_WINO_STAGES 3
#endif
constexpr int kWinoStages = GPP_WINO_STAGES;
static_assert(kWinoStages == 2 || kWinoStages == 3, "two or three stages");

struct WinoTiles { int tile_start[GPP_MAX_GROUPS + 1]; };         // first M tile of every group (prefix sums), filled in by the launcher

// Accumulating MFMA that writes the register it reads (a tied operand).  Left to itself the register allocator rotates the 12 accumulators of
// this loop through its three phases (22 of 36 MFMAs with vdst != srcC), which costs a power-bound loop ~20 % at the same instruction count
// (HISTORY.md 4.10).  An `asm` statement is invisible to the compiler's MFMA hazard recogniser, so the loop must not need one: every operand of
// these MFMAs comes from an LDS read (waited for by s_waitcnt, which the compiler does place for asm inputs) or from an MFMA 12 instructions
// earlier; the vector-ALU code that reads the accumulators (the fold between positions) sits behind explicit s_nops.
#ifndef GPP_WINO_TIED_MFMA
#define GPP_WINO_TIED_MFMA 1
#endif
__device__ __forceinline__ void wino_mfma(f32x4& acc, const f16x8 a, const f16x8 b)
{
#if GPP_WINO_TIED_MFMA
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
#endif
}

// ---------------------------------------------------------------------------------------------------------------- input transform
__global__ __launch_bounds__(256) void wino_transform_kernel(const gpp_wino_desc d)
{
    const int groups8 = d.C_in >> 3;
    const int64_t total = (int64_t)d.batch * d.pairs_per_image * groups8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(t % groups8) << 3;
        const int64_t bp = t / groups8;
        const int pair = (int)(bp % d.pairs_per_image), b = (int)(bp / d.pairs_per_image);
        int g = 0;
        while (g + 1 < d.n_groups && pair >= d.groups[g + 1].pair_off) ++g;
        const gpp_wino_group& G = d.groups[g];
        const int te = (G.W + 1) >> 1;
        const int local = pair - (int)G.pair_off, y = local / te, tx = local - y * te;
        float v[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = 2 * tx - 1 + j;
            if (x >= 0 && x < G.W) {
                const int64_t base = (int64_t)b * d.in_bstride + G.map_off + ((int64_t)y * G.W + x) * d.in_pitch;
                const char* p = x3_addr(d.in, base, n);
                x3_unpack<GPP_F16X3>(*(const f32x4*)p, *(const f32x4*)(p + 64), v[j]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[j][e] = 0.0f;
            }
        }
        float o[8];
        const int64_t vbase = ((int64_t)b * d.pairs_per_image + pair) * 4 * d.C_in;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[0][e] - v[2][e];
        x3_store<GPP_F16X3>(d.out, vbase, n, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[1][e] + v[2][e];
        x3_store<GPP_F16X3>(d.out, vbase + d.C_in, n, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[2][e] - v[1][e];
        x3_store<GPP_F16X3>(d.out, vbase + 2 * d.C_in, n, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[1][e] - v[3][e];
        x3_store<GPP_F16X3>(d.out, vbase + 3 * d.C_in, n, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------- the position GEMMs
__global__ __launch_bounds__(512, 1) void wino_conv_kernel(const gpp_wino_desc d, const WinoTiles tiles, const int n_tiles_n, const int total_wgs)
{
    constexpr int MF = 3, NF = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, total_wgs);
    const int mt = bid / n_tiles_n, nt = bid - mt * n_tiles_n;
    int g = 0;
    while (g + 1 < d.n_groups && mt >= tiles.tile_start[g + 1]) ++g;
    const gpp_wino_group G = d.groups[g];
    const int te = (G.W + 1) >> 1;
    const int rows_per_image = G.H * te;
    const int m_total = d.batch * rows_per_image;
    const int m0 = (mt - tiles.tile_start[g]) * kWinoPairs;
    const int n0 = nt * kWinoCols;
    const int chunks = d.C_in >> 5, ksteps_pos = chunks * 3, ksteps = ksteps_pos * 4;
    const int row_stride = te * 4 * d.C_in * 4;                                   // bytes between image rows y and y + 1 of V (same tx, same position)

    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.in, 0, d.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.weight, 0, d.weight_bytes, 0x00020000);

    // ---- what this lane moves per stage: kWinoPieces pieces of 8 rows; piece P = wave * 5 + q covers tile rows P * 8 .. + 7 (activation rows
    // for P < 24, weight rows behind them); the lane fetches 16 bytes: row P * 8 + lane / 8, source chunk (lane % 8) ^ (row & 7)
    int base[kWinoPieces];          // byte offset of the row's K-step 0 (kernel row 1 = the pair's own image row), incl. the swizzled chunk
    unsigned valid = 0;             // bit q * 3 + kh: the row of piece q exists for kernel row kh
    const int prow = lane >> 3, pchunk = lane & 7;
#pragma unroll
    for (int q = 0; q < kWinoPieces; ++q) {
        const int P = wave * kWinoPieces + q;
        const int r = P * 8 + prow;
        const int swz = (pchunk ^ (r & 7)) << 4;
        if (P < kWinoAPieces) {
            const int m = m0 + r;
            if (m < m_total) {
                const int b = m / rows_per_image, rem = m - b * rows_per_image, y = rem / te;
                const int64_t pair = (int64_t)b * d.pairs_per_image + G.pair_off + rem;
                base[q] = (int)(pair * 4 * d.C_in * 4) + swz;
                valid |= ((y > 0 ? 1u : 0u) | 2u | (y + 1 < G.H ? 4u : 0u)) << (q * 3);
            } else {
                base[q] = 0;
            }
        } else {
            base[q] = (n0 + (r - kWinoPairs)) * (ksteps * kRowBytes) + swz;
            valid |= 7u << (q * 3);
        }
    }
    // K-step ks = (position p, chunk c, kernel row kh), K order per position: chunk, kernel row.  The stage issued is always two steps ahead of
    // the one computed: its (p, c, kh) are kept as scalar counters instead of being divided out of ks every step.
    int ip = 0, ic = 0, ikh = 0, iks = 0;
    auto issue = [&](unsigned char* stage) {
        // (the scalar offset of a buffer load is unsigned and outside the range check: the kernel row's signed step goes into the lane offset)
        const int soff_a = __builtin_amdgcn_readfirstlane(ip * d.C_in * 4 + ic * kRowBytes);
        const int soff_b = __builtin_amdgcn_readfirstlane(iks * kRowBytes);
        const int row_step = __builtin_amdgcn_readfirstlane((ikh - 1) * row_stride);
#pragma unroll
        for (int q = 0; q < kWinoPieces; ++q) {
            const int P = wave * kWinoPieces + q;
            const bool is_a = P < kWinoAPieces;                                      // wave-uniform
            if (is_a) {
                const int voff = ((valid >> (q * 3 + ikh)) & 1u) ? base[q] + row_step : kOutOfRange;
                glds16(v_rsrc, voff, soff_a, stage + P * 1024);
            } else {
                glds16(w_rsrc, base[q], soff_b, stage + P * 1024);
            }
        }
        ++iks;
        if (++ikh == 3) { ikh = 0; if (++ic == chunks) { ic = 0; ++ip; } }
    };
    for (int st = 0; st < kWinoStages; ++st) issue(wsm + st * kWinoStage);

    const int frow = lane & 15, fq = lane >> 4;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int sw = ((h * 4 + fq) ^ (frow & 7)) << 4;
        a_rd[h] = (wm * (MF * 16) + frow) * kRowBytes + sw;
        b_rd[h] = kWinoPairs * kRowBytes + (wn * (NF * 16) + frow) * kRowBytes + sw;
    }
    f32x4 M[MF][NF], Y0[MF][NF], Y1[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) M[i][j] = Y0[i][j] = Y1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f16x8 ah[MF], al[MF], bh[NF], bl[NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) ah[i] = *(const f16x8*)(wsm + a_rd[0] + i * 16 * kRowBytes);
#pragma unroll
    for (int j = 0; j < NF; ++j) bl[j] = *(const f16x8*)(wsm + b_rd[1] + j * 16 * kRowBytes);
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);

    // one K-step: the software pipeline runs across position boundaries (the stage of step ks + 2 may belong to the next position)
    int cur = 0;                                 // ring slot of the stage being computed
    auto kstep = [&](const int ks) {
        unsigned char* scur = wsm + cur * kWinoStage;
        const int nxt = cur + 1 == kWinoStages ? 0 : cur + 1;
        const unsigned char* snxt = wsm + nxt * kWinoStage;
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase A: hi * wlo; reads whi
#pragma unroll
        for (int gi = 0; gi < MF; ++gi) {
#pragma unroll
            for (int q = gi * NF / MF; q < (gi + 1) * NF / MF; ++q) bh[q] = *(const f16x8*)(scur + b_rd[0] + q * 16 * kRowBytes);
#pragma unroll
            for (int j = 0; j < NF; ++j) { const int js = (gi & 1) ? NF - 1 - j : j; wino_mfma(M[gi][js], bl[js], ah[gi]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- phase B: hi * whi; reads lo
#pragma unroll
        for (int gi = 0; gi < MF; ++gi) {
            al[gi] = *(const f16x8*)(scur + a_rd[1] + gi * 16 * kRowBytes);
#pragma unroll
            for (int j = 0; j < NF; ++j) { const int js = (gi & 1) ? NF - 1 - j : j; wino_mfma(M[gi][js], bh[js], ah[gi]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next stage has landed: everything issued so far (two stages), or everything but the newest stage's pieces (three: loads return in order)
        if (kWinoStages == 2 || ks + kWinoStages > ksteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWinoPieces) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wavefront has read what it needs of the current one
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase C: lo * whi; the stage after next goes out into the buffer this step has finished with; reads hi, wlo of the next step
        if (ks + kWinoStages < ksteps) issue(scur);
#pragma unroll
        for (int gi = 0; gi < MF; ++gi) {
            ah[gi] = *(const f16x8*)(snxt + a_rd[0] + gi * 16 * kRowBytes);
#pragma unroll
            for (int q = gi * NF / MF; q < (gi + 1) * NF / MF; ++q) bl[q] = *(const f16x8*)(snxt + b_rd[1] + q * 16 * kRowBytes);
#pragma unroll
            for (int j = 0; j < NF; ++j) { const int js = (gi & 1) ? NF - 1 - j : j; wino_mfma(M[gi][js], bh[js], al[gi]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = nxt;
    };
    // ONE copy of the K-step in the code (several copies make the register allocator rotate the accumulators through them: out-of-place MFMAs,
    // spills -- HISTORY.md 4.10): a loop over the positions around the loop over a position's K-steps, the fold between them
    int ks = 0;
#pragma nounroll
    for (int pos = 0; pos < 4; ++pos) {
#pragma nounroll
        for (int k = 0; k < ksteps_pos; ++k) kstep(ks++);
        // end of a position: M x the inverse of the position's weight scale (a power of two: exact) goes into the two outputs of the pair
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");              // (matrix-pipe results read by the vector ALU: see wino_mfma)
        const float* sc = d.out_scale + pos * d.C_out + n0 + wn * (NF * 16);
        const float to_y0 = pos < 3 ? 1.0f : 0.0f, to_y1 = pos == 0 ? 0.0f : (pos == 1 ? 1.0f : -1.0f);      // Y0 = M0 + M1 + M2, Y1 = M1 - M2 - M3
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const f32x4 s4 = *(const f32x4*)(sc + (j >> 1) * 32 + fq * 8 + (j & 1) * 4);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m = M[i][j][e] * s4[e];
                    Y0[i][j][e] = fmaf(to_y0, m, Y0[i][j][e]);           // (x 1, x 0 or x -1 and an addition of an exact product: no rounding beyond the sum's)
                    Y1[i][j][e] = fmaf(to_y1, m, Y1[i][j][e]);
                    M[i][j][e] = 0.f;
                }
        }
    }
    __builtin_amdgcn_s_setprio(0);

    // ---- epilogue: bias, ReLU, range check + split + store (x3_store), 8 consecutive channels per lane and fragment pair
#pragma unroll
    for (int jj = 0; jj < NF / 2; ++jj) {
        const int n = n0 + wn * (NF * 16) + jj * 32 + fq * 8;
        const f32x4 b0 = d.bias ? *(const f32x4*)(d.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 b1 = d.bias ? *(const f32x4*)(d.bias + n + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = m0 + wm * (MF * 16) + i * 16 + frow;
            if (m >= m_total) continue;
            const int b = m / rows_per_image, rem = m - b * rows_per_image, y = rem / te, tx = rem - y * te;
            const int64_t obase = (int64_t)b * d.out_bstride + G.map_off + ((int64_t)y * G.W + 2 * tx) * d.out_pitch;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = Y0[i][2 * jj][e] + b0[e]; v[4 + e] = Y0[i][2 * jj + 1][e] + b1[e]; }
            if (d.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
            }
            x3_store<GPP_F16X3>(d.out, obase, n, v);
            if (2 * tx + 1 < G.W) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = Y1[i][2 * jj][e] + b0[e]; v[4 + e] = Y1[i][2 * jj + 1][e] + b1[e]; }
                if (d.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
                }
                x3_store<GPP_F16X3>(d.out, obase + d.out_pitch, n, v);
            }
        }
    }
}

int wino_validate(const gpp_wino_desc& d, bool conv)
{
    if (!d.in || !d.out || d.batch <= 0 || d.n_groups < 1 || d.n_groups > GPP_MAX_GROUPS || d.pairs_per_image <= 0) return GPP_ERR_BAD_ARG;
    if (d.C_in <= 0 || d.C_in % 32 != 0) return GPP_ERR_UNSUPPORTED;
    if (((uintptr_t)d.in | (uintptr_t)d.out | (uintptr_t)d.weight | (uintptr_t)d.bias | (uintptr_t)d.out_scale) & 15) return GPP_ERR_ALIGN;
    int64_t pairs = 0;
    for (int g = 0; g < d.n_groups; ++g) {
        const gpp_wino_group& G = d.groups[g];
        if (G.H <= 0 || G.W <= 0 || G.pair_off != pairs || G.map_off % 32 != 0) return GPP_ERR_BAD_ARG;
        pairs += (int64_t)G.H * ((G.W + 1) / 2);
    }
    if (pairs != d.pairs_per_image) return GPP_ERR_BAD_ARG;
    const int64_t v_bytes = (int64_t)d.batch * d.pairs_per_image * 4 * d.C_in * 4;
    if (v_bytes >= (1LL << 31)) return GPP_ERR_UNSUPPORTED;                        // 32-bit offsets into the transformed map
    if (conv) {
        if (!d.weight || !d.out_scale || d.C_out <= 0 || d.C_out % kWinoCols != 0) return GPP_ERR_UNSUPPORTED;
        if (d.out_pitch < d.C_out || d.out_pitch % 32 != 0 || d.out_bstride % 32 != 0) return GPP_ERR_ALIGN;
        if ((int64_t)d.C_out * 12 * (d.C_in / 32) * kRowBytes >= (1LL << 31)) return GPP_ERR_UNSUPPORTED;
    } else {
        if (d.in_pitch < d.C_in || d.in_pitch % 32 != 0 || d.in_bstride % 32 != 0) return GPP_ERR_ALIGN;
    }
    return GPP_OK;
}

}  // namespace

int gpp_wino_transform_dispatch_f16x3(const gpp_wino_desc& d, hipStream_t st)
{
    const int rc = wino_validate(d, false);
    if (rc != GPP_OK) return rc;
    const int64_t total = (int64_t)d.batch * d.pairs_per_image * (d.C_in >> 3);
    const int64_t blocks = (total + 255) / 256;
    wino_transform_kernel<<<(unsigned)(blocks < 65536 * 16 ? blocks : 65536 * 16), 256, 0, st>>>(d);
    return (int)hipGetLastError();
}

int gpp_wino_conv_dispatch_f16x3(const gpp_wino_desc& host, hipStream_t st)
{
    gpp_wino_desc d = host;
    const int rc = wino_validate(d, true);
    if (rc != GPP_OK) return rc;
    WinoTiles tiles;
    int t = 0;
    for (int g = 0; g < d.n_groups; ++g) {
        tiles.tile_start[g] = t;
        const int64_t rows = (int64_t)d.batch * d.groups[g].H * ((d.groups[g].W + 1) / 2);
        t += (int)((rows + kWinoPairs - 1) / kWinoPairs);
    }
    for (int g = d.n_groups; g <= GPP_MAX_GROUPS; ++g) tiles.tile_start[g] = t;
    const int n_tiles_n = d.C_out / kWinoCols;
    const int total_wgs = t * n_tiles_n;
    d.in_bytes = (int32_t)((int64_t)d.batch * d.pairs_per_image * 4 * d.C_in * 4);
    d.weight_bytes = (int32_t)((int64_t)d.C_out * 12 * (d.C_in / 32) * kRowBytes);
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GPP_ERR_UNSUPPORTED;
    if (!(configured.load(std::memory_order_acquire) >> dev & 1ull)) {
        hipError_t e = hipFuncSetAttribute((const void*)wino_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWinoStages * kWinoStage);
        if (e != hipSuccess) return (int)e;
        configured.fetch_or(1ull << dev, std::memory_order_release);
    }
    wino_conv_kernel<<<total_wgs, 512, kWinoStages * kWinoStage, st>>>(d, tiles, n_tiles_n, total_wgs);
    return (int)hipGetLastError();
}

#endif
