// Loader-wavefront form of the implicit-GEMM convolution for the LATENCY-BOUND layers: small M, deep K (res4 / res5 of the
// backbone, the small FPN convolutions; reference graph: keras_resnet bottlenecks instantiated at
// /root/reference/keras_retinanet_3D/models/resnet.py:88-93, FPN models/retinanet.py:170-205).
//
// What bounds those layers in conv_igemm_kernel (csrc/conv_igemm_impl.h): their activation rows come from the Infinity
// Cache / HBM (the producing layer ran on other XCDs) with ~2.5 us of latency under load, a workgroup of the two-buffer
// loop has exactly ONE K-step of them in flight, and every wavefront's loads retire in order (one vmcnt per wavefront), so
// the low-latency weight stream cannot run ahead of the activation stream either: K-step time = activation latency /
// workgroups per CU (measured 0.6 - 1.2 us per K-step against 0.1 - 0.3 us of matrix work).
//
// This kernel separates the two streams by WAVEFRONT, which is what separates their vmcnt queues:
//   * wavefront 4 (the loader) owns the activation tile: it keeps SA - 1 K-steps of LDS-DMA (buffer_load ... lds, 128-byte
//     rows, padding through the descriptor's range check, exactly as in conv_igemm_kernel) in flight into an SA-deep LDS
//     ring, waits with a counted vmcnt for the oldest one only, and joins the workgroup barrier of that K-step;
//   * wavefronts 0..3 (1 x 4 over the output channels) never touch the activation loads: each fetches the weight fragments
//     of ITS 1/4 of the output channels straight from L2 into registers (global_load_dwordx4, PB K-steps ahead, no LDS, no
//     sharing needed: the four wavefronts own disjoint columns), reads the activation fragments of the whole BM-row tile
//     from the ring and runs the MFMAs;
//   * one s_barrier per K-step is both the FULL signal (the loader arrives after its vmcnt wait) and the FREE signal
//     (the compute wavefronts arrive after their LDS reads): the loader refills the slot of K-step ks - 1 right after the
//     barrier of K-step ks.
// LDS holds only activations: 64-row tile x 6 slots = 48 KB -> 3 workgroups per CU with 5 activation K-steps in flight
// each (120 KB per CU against 24 KB in the two-buffer loop).
//
// MEASURED AND REJECTED (MI355X, B = 8, tools/bench_conv.py ring, profiles/r2/ring_kernel.txt): 1.5 - 2x SLOWER than the plain
// two-buffer tiles on every layer it was built for (res4 2a: 36 - 40 us against 20 - 21 us; res5 2b: 59 - 83 against 45; P4: 169 -
// 211 against 84), hot and cold alike.  The premise was wrong: those layers are not waiting for latency, they are bound by the
// CU's vector-memory instruction throughput -- a 1 KB wave-instruction (LDS-DMA piece or global_load_dwordx4 alike) retires
// every 20 - 30 ns per CU, i.e. 35 - 50 GB/s per CU whatever the destination, and this form issues the same number of them (the
// register-direct weight fragments touch twice as many cache lines on top).  Kept, tested and reachable by explicit tile code
// as the record of that experiment; the autotuner does not consider it.
//
// The K order of every output element is the one of conv_igemm_kernel (channel chunk, kh, kw; kk = 0, 1 inside a K-step), and
// the epilogue is the same arithmetic: results are bit-identical to every other block tile (tests/test_conv_gpu.py).
#ifndef GPP_CONV_RING_IMPL_H_
#define GPP_CONV_RING_IMPL_H_

#include "conv_igemm_impl.h"

namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt_const()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most `young` whole stages (STEP instructions each) are outstanding; young is wave-uniform, 0..MAXY
template <int STEP, int MAXY>
__device__ __forceinline__ void wait_vmcnt_stages(int young)
{
    static_assert(STEP * MAXY <= 63, "vmcnt is a 6-bit counter");
    if constexpr (MAXY >= 7) { if (young >= 7) { wait_vmcnt_const<STEP * 7>(); return; } }
    if constexpr (MAXY >= 6) { if (young == 6) { wait_vmcnt_const<STEP * 6>(); return; } }
    if constexpr (MAXY >= 5) { if (young == 5) { wait_vmcnt_const<STEP * 5>(); return; } }
    if constexpr (MAXY >= 4) { if (young == 4) { wait_vmcnt_const<STEP * 4>(); return; } }
    if constexpr (MAXY >= 3) { if (young == 3) { wait_vmcnt_const<STEP * 3>(); return; } }
    if constexpr (MAXY >= 2) { if (young == 2) { wait_vmcnt_const<STEP * 2>(); return; } }
    if constexpr (MAXY >= 1) { if (young == 1) { wait_vmcnt_const<STEP * 1>(); return; } }
    wait_vmcnt_const<0>();
}

template <int S> struct Slot { static constexpr int value = S; };

template <int DT, int BM, int BN, int SA, int PB>
__global__ __launch_bounds__(320) void conv_ring_kernel(const gpp_conv_desc d)
{
    using E = Elem<DT>;
    using vec8 = typename E::vec8;
    using frag = typename E::frag;
    using scalar = typename E::scalar;
    static_assert(E::ESZ == 2, "16-bit storage types");
    constexpr int NCW = 4;                               // compute wavefronts, 1 x 4 over the output channels
    constexpr int MF = BM / 16, COLS = BN / NCW, NF = COLS / 16;
    constexpr int A_BYTES = BM * kRowBytes;
    constexpr int A_INSTR = BM / 8;                      // LDS-DMA instructions of one activation stage (8 rows each)
    constexpr int NB = PB + 1;                           // register sets of the weight fragments
    static_assert(BM % 16 == 0 && NF >= 2 && NF % 2 == 0 && SA >= 3 && PB >= 1 && PB <= 3, "tile shape");
    static_assert((SA - 2) * A_INSTR <= 63, "the loader's counted wait must fit vmcnt");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- which tile (as conv_igemm_body)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tiles = (d.C_out + BN - 1) / BN;
    const int nt = bid % n_tiles, mt = bid / n_tiles;
    int tile_start = 0, H_in = 0, W_in = 0, H_out = 0, W_out = 0, H_res = 0, W_res = 0;
    int64_t in_off = 0, in_bs = 0, out_off = 0, out_bs = 0, res_off = 0, res_bs = 0;
#pragma unroll
    for (int q = 0; q < GPP_MAX_GROUPS; ++q) {
        if (q < d.n_groups && mt >= d.groups[q].tile_start) {
            tile_start = d.groups[q].tile_start;
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
    const int m0 = (mt - tile_start) * BM, n0 = nt * BN;
    const int Ktot = d.KH * d.KW * d.C_in;
    const int cpt = d.C_in >> 6;                         // 64-channel chunks per tap
    const int nk_total = d.KH * d.KW * cpt;
    const int nsplit = gridDim.y, split = blockIdx.y;    // split-K exactly as conv_igemm_body
    const int ks0 = (int)((int64_t)nk_total * split / nsplit);
    const int nk = (int)((int64_t)nk_total * (split + 1) / nsplit) - ks0;

    if (wave == NCW) {
        // =========================================================================== the loader wavefront
        const int srow = lane >> 3;
        const int gchunk = (lane & 7) ^ srow;            // source chunk of the XOR-swizzled destination (as conv_igemm_body)
        const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.in, 0, d.in_bytes, 0x00020000);
        int a_base[A_INSTR], a_mask[A_INSTR];
        const int pitch2 = d.in_pitch * 2;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int m = m0 + i * 8 + srow;
            a_mask[i] = 0;
            a_base[i] = 0;
            if (m < Mg) {
                const int b = m / HoWo, p = m - b * HoWo;
                const int oy = p / W_out, ox = p - oy * W_out;
                const int iy0 = oy * d.stride - d.pad_top, ix0 = ox * d.stride - d.pad_left;
                int mask = 0;
                for (int k = 0; k < d.KH; ++k) mask |= ((unsigned)(iy0 + k) < (unsigned)H_in) << k;
                for (int k = 0; k < d.KW; ++k) mask |= ((unsigned)(ix0 + k) < (unsigned)W_in) << (8 + k);
                a_mask[i] = mask;
                a_base[i] = (int)((in_off + (int64_t)b * in_bs) * 2) + gchunk * 16 + (iy0 * W_in + ix0) * pitch2;
            }
        }
        const int taps = d.KH * d.KW;
        int cc = ks0 / taps, kw = (ks0 % taps) % d.KW, kh = (ks0 % taps) / d.KW, issued = 0, slot = 0;
        auto issue_stage = [&]() {
            const int delta = (kh * W_in + kw) * pitch2;
            const int need = (1 << kh) | (1 << (8 + kw));
            unsigned char* sa = smem + slot * A_BYTES;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) {
                const int voff = ((a_mask[i] & need) == need) ? a_base[i] + delta : kOutOfRange;
                glds16(in_rsrc, voff, cc * kRowBytes, sa + i * 8 * kRowBytes);
            }
            if (++kw == d.KW) {
                kw = 0;
                if (++kh == d.KH) { kh = 0; ++cc; }
            }
            ++issued;
            if (++slot == SA) slot = 0;
        };
        for (int p = 0; p < SA - 1; ++p)
            if (issued < nk) issue_stage();
        for (int ks = 0; ks < nk; ++ks) {
            wait_vmcnt_stages<A_INSTR, SA - 2>(issued - ks - 1);      // stage ks has landed; the younger ones stay in flight
            __builtin_amdgcn_s_barrier();                             // FULL for K-step ks, FREE for the slot of K-step ks - 1
            asm volatile("" ::: "memory");
            if (issued < nk) issue_stage();
        }
        return;
    }

    // =============================================================================== the four compute wavefronts
    const int wn = wave;
    const int frow = lane & 15, fq = lane >> 4;
    int a_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_rd[kk] = frow * kRowBytes + (((kk * 4 + fq) ^ (frow & 7)) << 4);
    // weight fragments: stored row n0 + wn*COLS + 16 j + frow, bytes [ks*128 + (kk*4 + fq)*16, +16) of that row
    const unsigned char* wrow = (const unsigned char*)d.weight + ((int64_t)(n0 + wn * COLS + frow) * Ktot) * 2 + fq * 16 + (int64_t)ks0 * kRowBytes;
    const int64_t jstride = (int64_t)16 * Ktot * 2;
    frag bq[NB][NF][2];
    auto load_b = [&](auto s, int ks) {
        constexpr int S = decltype(s)::value;
        const unsigned char* p = wrow + (int64_t)(ks < nk ? ks : nk - 1) * kRowBytes;      // past the end: a valid, unused K-step
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            bq[S][j][0] = *(const frag*)(p + j * jstride);
            bq[S][j][1] = *(const frag*)(p + j * jstride + 64);
        }
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // shortcut rows of this tile, requested before anything else (in flight under the whole main loop)
    vec8 rpre[MF][NF / 2];
    RowAddr ra_pre[MF];
    const bool use_pre = d.residual != nullptr && nsplit == 1 && (d.C_out & 7) == 0;
    if (use_pre) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = m0 + i * 16 + frow;
            ra_pre[i] = row_addr(d, m < Mg ? m : 0, HoWo, W_out, H_out, H_res, W_res, out_off, out_bs, res_off, res_bs);
#pragma unroll
            for (int jj = 0; jj < NF / 2; ++jj) {
                const int n = n0 + wn * COLS + jj * 32 + fq * 8;
                rpre[i][jj] = *(const vec8*)((const scalar*)d.residual + ra_pre[i].rbase + (n < d.C_out ? n : 0));
            }
        }
    }

    if constexpr (PB >= 1) load_b(Slot<0>{}, 0);
    if constexpr (PB >= 2) load_b(Slot<1>{}, 1);
    if constexpr (PB >= 3) load_b(Slot<2>{}, 2);
    int aslot = 0;
    auto step = [&](auto s, int ks) {
        constexpr int S = decltype(s)::value;
        __builtin_amdgcn_s_barrier();                    // activation stage ks is in LDS; everyone is done with K-step ks - 1
        asm volatile("" ::: "memory");
        load_b(Slot<(S + PB) % NB>{}, ks + PB);          // into the register set of K-step ks - 1
        const unsigned char* sa = smem + aslot * A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag af[MF];
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const frag*)(sa + a_rd[kk] + i * 16 * kRowBytes);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = E::mfma(bq[S][j][kk], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wavefront's reads of the slot are done before the next barrier
        if (++aslot == SA) aslot = 0;
    };
    for (int kb = 0; kb < nk; kb += NB) {
        step(Slot<0>{}, kb);
        if constexpr (NB >= 2) { if (kb + 1 < nk) step(Slot<1>{}, kb + 1); }
        if constexpr (NB >= 3) { if (kb + 2 < nk) step(Slot<2>{}, kb + 2); }
        if constexpr (NB >= 4) { if (kb + 3 < nk) step(Slot<3>{}, kb + 3); }
    }

    // ---- epilogue, straight from registers (conv_igemm_body's, with one row of four wavefronts)
    if (nsplit > 1) {
        const int64_t rows_pad = (int64_t)d.partial_rows, npad = (int64_t)n_tiles * BN;
        float* part = (float*)d.partial + ((int64_t)split * rows_pad + (int64_t)mt * BM) * npad + nt * BN;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int lr = i * 16 + frow;
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
    float bias_v[NF / 2][8];
#pragma unroll
    for (int jj = 0; jj < NF / 2; ++jj) {
        const int n = n0 + wn * COLS + jj * 32 + fq * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_v[jj][e] = (d.bias && n + e < d.C_out) ? d.bias[n + e] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + i * 16 + frow;
        if (m >= Mg) continue;
        RowAddr ra;
        if (use_pre) ra = ra_pre[i];
        else ra = row_addr(d, m, HoWo, W_out, H_out, H_res, W_res, out_off, out_bs, res_off, res_bs);
#pragma unroll
        for (int jj = 0; jj < NF / 2; ++jj) {
            const int n = n0 + wn * COLS + jj * 32 + fq * 8;
            if (n >= d.C_out) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[i][2 * jj][e] + bias_v[jj][e];
                v[4 + e] = acc[i][2 * jj + 1][e] + bias_v[jj][4 + e];
            }
            if (use_pre) finish8_pre<DT>(d, v, n, ra.obase, true, rpre[i][jj]);
            else finish8<DT>(d, v, n, ra.obase, res ? res + ra.rbase : nullptr);
        }
    }
}

template <int DT, int BM, int BN, int SA, int PB>
int launch_ring(gpp_conv_desc& d, hipStream_t st)
{
    constexpr int lds = SA * BM * kRowBytes;
    static DeviceOnce once;
    auto kernel = conv_ring_kernel<DT, BM, BN, SA, PB>;
    int rc = once.configure(kernel, lds);
    if (rc != GPP_OK) return rc;
    const int tiles = prepare<BM, BN>(d);
    if (tiles < 0) return tiles;
    const int n_tiles = (d.C_out + BN - 1) / BN;
    const int nk = d.KH * d.KW * (d.C_in / 64);
    int nsplit = 1;
    if (d.split_k > 1) {
        nsplit = d.split_k;
        if (nk / nsplit < 1) return GPP_ERR_BAD_ARG;
        const int64_t slab = (int64_t)tiles * BM * n_tiles * BN * 4;
        if (!d.partial || slab * nsplit > (int64_t)d.partial_bytes) return GPP_ERR_WORKSPACE;
    }
    kernel<<<dim3((unsigned)(tiles * n_tiles), (unsigned)nsplit), dim3(320), lds, st>>>(d);
    if (nsplit > 1) {
        const int64_t total = (int64_t)d.partial_rows * ((d.C_out + 7) / 8);
        splitk_reduce_kernel<DT><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(d, BM, n_tiles * BN, nsplit);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

// tile codes 3000000 + BM * 1000 + BN
template <int DT>
int dispatch_ring(gpp_conv_desc& d, hipStream_t st)
{
    switch (d.tile_hint) {
        case 3064128: return launch_ring<DT, 64, 128, 6, 2>(d, st);
        case 3096128: return launch_ring<DT, 96, 128, 5, 2>(d, st);
        case 3128128: return launch_ring<DT, 128, 128, 4, 2>(d, st);
        case 3064256: return launch_ring<DT, 64, 256, 6, 1>(d, st);
        default: return GPP_ERR_BAD_ARG;
    }
}

}  // namespace

#endif  // GPP_CONV_RING_IMPL_H_
