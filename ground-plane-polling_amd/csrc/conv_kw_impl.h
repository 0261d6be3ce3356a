// 3x3 / stride 1 / pad 1 convolution with the activation rows staged ONCE per (channel chunk, kernel row) and shared by the
// three kernel columns: the form for the 3x3 layers that are bound by the CU's vector-memory instruction throughput
// (res4 / res5 branch2b, P4, P5 at B = 8: a 1 KB LDS-DMA piece retires every 20 - 30 ns per CU whatever the tile, so their
// time is proportional to the number of pieces; profiles/r2/ring_kernel.txt).
// Reference graph: keras_resnet bottleneck "branch2b" (models/resnet.py:88-93), FPN P4 / P5 (models/retinanet.py:185-196).
//
// conv_igemm_kernel stages, per K-step (chunk, kh, kw), the BM activation rows of that tap: consecutive output pixels of an
// image row read consecutive input pixels, so the tiles of kw = 0, 1, 2 are the same rows shifted by one.  Here a PATCH of
// BM + 2 rows (linear output pixels m0 - 1 .. m0 + BM at the centre column, + 6 rows of padding to a multiple of 8) is
// staged once per (chunk, kh) and tap kw of output row i reads patch row i + kw: 13 instead of 36 activation pieces per
// three K-steps at BM = 96 (weights unchanged: 16 per K-step) -> 61 instead of 84 pieces.
// Where the linear neighbour is not the horizontal neighbour -- ox = 0 for kw = 0, ox = W - 1 for kw = 2: the reference
// pads with zeros there (Keras 'same') -- the fragment is replaced by zeros in registers (v_cndmask on a per-lane flag).
//
// Same K order (chunk, kh, kw; kk = 0, 1), same operand values, same epilogue arithmetic as conv_igemm_kernel: results are
// bit-identical to every other block tile (tests/test_conv_gpu.py).  16-bit storage types; no residual input (none of the
// 3x3 layers of the graph has one); split-K as everywhere (float32 partial slabs + splitk_reduce_kernel).
#ifndef GPP_CONV_KW_IMPL_H_
#define GPP_CONV_KW_IMPL_H_

#include "conv_igemm_impl.h"

namespace {

template <int DT, int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv3x3_kw_kernel(const gpp_conv_desc d)
{
    using E = Elem<DT>;
    using frag = typename E::frag;
    using scalar = typename E::scalar;
    static_assert(E::ESZ == 2, "16-bit storage types");
    constexpr int WM = 2, WN = 2, NW = 4;
    constexpr int MF = BM / WM / 16, NF = BN / WN / 16;
    constexpr int PROWS = BM + 8;                        // patch rows: row r <-> linear output pixel m0 + r - 1 (rows BM + 2 .. unused)
    constexpr int A_PIECES = PROWS / 8, NP = (A_PIECES + NW - 1) / NW;     // LDS-DMA pieces of a patch; at most NP per wavefront
    constexpr int P_BYTES = PROWS * kRowBytes, B_BYTES = BN * kRowBytes;
    constexpr int B_IT = BN / 8 / NW;
    static_assert(BM % 32 == 0 && BN % 64 == 0 && NF % 2 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2 patches][2 weight tiles]
    unsigned char* const s_patch = smem;
    unsigned char* const s_w = smem + 2 * P_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- which tile (as conv_igemm_body)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tiles = (d.C_out + BN - 1) / BN;
    const int nt = bid % n_tiles, mt = bid / n_tiles;
    int tile_start = 0, H_in = 0, W_in = 0, H_out = 0, W_out = 0;
    int64_t in_off = 0, in_bs = 0, out_off = 0, out_bs = 0;
#pragma unroll
    for (int q = 0; q < GPP_MAX_GROUPS; ++q) {
        if (q < d.n_groups && mt >= d.groups[q].tile_start) {
            tile_start = d.groups[q].tile_start;
            H_in = d.groups[q].H_in; W_in = d.groups[q].W_in;
            H_out = d.groups[q].H_out; W_out = d.groups[q].W_out;
            in_off = d.groups[q].in_off; in_bs = d.groups[q].in_bstride;
            out_off = d.groups[q].out_off; out_bs = d.groups[q].out_bstride;
        }
    }
    const int HoWo = H_out * W_out;
    const int Mg = d.batch * HoWo;
    const int m0 = (mt - tile_start) * BM, n0 = nt * BN;
    const int Ktot = 9 * d.C_in;
    const int cpt = d.C_in >> 6;
    const int nk_total = 9 * cpt;
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int ks0 = (int)((int64_t)nk_total * split / nsplit);
    const int nk = (int)((int64_t)nk_total * (split + 1) / nsplit) - ks0;
    if (nk <= 0) return;

    // ---- staging bookkeeping
    const int srow = lane >> 3;
    const int gchunk = (lane & 7) ^ srow;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.in, 0, d.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.weight, 0, d.weight_bytes, 0x00020000);
    const int pitch2 = d.in_pitch * 2;
    // patch piece t of this wavefront = piece wave + 4 t: patch rows 8 (wave + 4 t) + srow <-> linear pixel m0 + row - 1, centre column
    int p_base[NP], p_mask[NP];
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        const int m = m0 + (wave + NW * t) * 8 + srow - 1;
        p_base[t] = 0;
        p_mask[t] = 0;
        if (m >= 0 && m < Mg && wave + NW * t < A_PIECES) {
            const int b = m / HoWo, p = m - b * HoWo;
            const int oy = p / W_out, ox = p - oy * W_out;
            int mask = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) mask |= ((unsigned)(oy - 1 + k) < (unsigned)H_in) << k;
            p_mask[t] = mask;
            p_base[t] = (int)((in_off + (int64_t)b * in_bs) * 2) + gchunk * 16 + ((oy - 1) * W_in + ox) * pitch2;
        }
    }
    int w_voff[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) w_voff[i] = (n0 + (wave * B_IT + i) * 8 + srow) * Ktot * 2 + gchunk * 16;

    auto stage_patch = [&](int pid_abs, int slot) {      // patch pid_abs = (chunk, kh) = (pid_abs / 3, pid_abs % 3)
        const int cc = pid_abs / 3, kh = pid_abs - cc * 3;
        unsigned char* sp = s_patch + slot * P_BYTES;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            if (wave + NW * t < A_PIECES) {
                const int voff = ((p_mask[t] >> kh) & 1) ? p_base[t] + kh * W_in * pitch2 : kOutOfRange;
                glds16(in_rsrc, voff, cc * kRowBytes, sp + (wave + NW * t) * 8 * kRowBytes);
            }
        }
    };
    auto stage_w = [&](int ks_abs, int slot) {
        unsigned char* sb = s_w + slot * B_BYTES + wave * B_IT * 8 * kRowBytes;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16(w_rsrc, w_voff[i], ks_abs * kRowBytes, sb + i * 8 * kRowBytes);
    };

    // ---- fragment read offsets: activation row (tile row + kw) of the patch, weight rows as conv_igemm_body
    const int frow = lane & 15, fq = lane >> 4;
    int a_rd[3][2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            a_rd[kw][kk] = (wm * (BM / WM) + frow + kw) * kRowBytes + (((kk * 4 + fq) ^ ((frow + kw) & 7)) << 4);
        b_rd[kk] = (wn * (BN / WN) + frow) * kRowBytes + (((kk * 4 + fq) ^ (frow & 7)) << 4);
    }
    // per M tile of this lane: is its output pixel at the left / right edge of the image row?  (bit i: tile i)
    int edge_l = 0, edge_r = 0;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + frow;
        const int ox = ((m < Mg ? m : 0) % HoWo) % W_out;
        edge_l |= (ox == 0) << i;
        edge_r |= (ox == W_out - 1) << i;
    }

    frag zero_frag;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero_frag[e] = (scalar)0.0f;

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- main loop.  Absolute K-step a = ks0 + ks = 3 * patch + kw.  Two patch slots, two weight slots; one barrier per K-step.
    const int pid0 = ks0 / 3, pid_last = (ks0 + nk - 1) / 3;
    stage_patch(pid0, 0);
    stage_w(ks0, 0);
    for (int ks = 0; ks < nk; ++ks) {
        const int a = ks0 + ks, pid = a / 3, kw = a - pid * 3;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // patch pid and weight tile ks have landed; everyone is done with K-step ks - 1
        asm volatile("" ::: "memory");
        if (ks + 1 < nk) stage_w(a + 1, (ks + 1) & 1);
        if ((kw == 0 || ks == 0) && pid < pid_last) stage_patch(pid + 1, (pid + 1 - pid0) & 1);     // its slot was last read in K-step ks - 1
        const unsigned char* sp = s_patch + ((pid - pid0) & 1) * P_BYTES;
        const unsigned char* sb = s_w + (ks & 1) * B_BYTES;
        const int zero_bits = kw == 0 ? edge_l : (kw == 2 ? edge_r : 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag af[MF], bfr[NF];
            const int aoff = kw == 0 ? a_rd[0][kk] : (kw == 1 ? a_rd[1][kk] : a_rd[2][kk]);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                af[i] = *(const frag*)(sp + aoff + i * 16 * kRowBytes);
                if ((zero_bits >> i) & 1) af[i] = zero_frag;                             // 'same' padding at the image-row edge
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) bfr[j] = *(const frag*)(sb + b_rd[kk] + j * 16 * kRowBytes);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = E::mfma(bfr[j], af[i], acc[i][j]);
        }
    }

    // ---- epilogue (conv_igemm_body's, without a residual input)
    constexpr int COLS = BN / WN;
    if (nsplit > 1) {
        const int64_t rows_pad = (int64_t)d.partial_rows, npad = (int64_t)n_tiles * BN;
        float* part = (float*)d.partial + ((int64_t)split * rows_pad + (int64_t)mt * BM) * npad + nt * BN;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int lr = wm * (BM / WM) + i * 16 + frow;
#pragma unroll
            for (int jj = 0; jj < NF / 2; ++jj) {
                float* dst = part + (int64_t)lr * npad + wn * COLS + jj * 32 + fq * 8;
                *(f32x4*)dst = acc[i][2 * jj];
                *(f32x4*)(dst + 4) = acc[i][2 * jj + 1];
            }
        }
        return;
    }
    float bias_v[NF / 2][8];
#pragma unroll
    for (int jj = 0; jj < NF / 2; ++jj) {
        const int n = n0 + wn * COLS + jj * 32 + fq * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_v[jj][e] = (d.bias && n + e < d.C_out) ? d.bias[n + e] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + frow;
        if (m >= Mg) continue;
        const int b = m / HoWo, p = m - b * HoWo;
        const int64_t obase = out_off + (int64_t)b * out_bs + (int64_t)p * d.out_pitch;
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
            finish8<DT>(d, v, n, obase, (const scalar*)nullptr);
        }
    }
}

template <int DT, int BM, int BN>
int launch_kw(gpp_conv_desc& d, hipStream_t st)
{
    if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad_top != 1 || d.pad_left != 1 || d.residual) return GPP_ERR_UNSUPPORTED;
    for (int g = 0; g < d.n_groups; ++g)
        if (d.groups[g].H_in != d.groups[g].H_out || d.groups[g].W_in != d.groups[g].W_out) return GPP_ERR_UNSUPPORTED;
    constexpr int lds = 2 * (BM + 8) * kRowBytes + 2 * BN * kRowBytes;
    static DeviceOnce once;
    auto kernel = conv3x3_kw_kernel<DT, BM, BN>;
    int rc = once.configure(kernel, lds);
    if (rc != GPP_OK) return rc;
    const int tiles = prepare<BM, BN>(d);
    if (tiles < 0) return tiles;
    const int n_tiles = (d.C_out + BN - 1) / BN;
    const int nk = 9 * (d.C_in / 64);
    int nsplit = 1;
    if (d.split_k > 1) {
        nsplit = d.split_k;
        if (nk / nsplit < 1) return GPP_ERR_BAD_ARG;
        const int64_t slab = (int64_t)tiles * BM * n_tiles * BN * 4;
        if (!d.partial || slab * nsplit > (int64_t)d.partial_bytes) return GPP_ERR_WORKSPACE;
    }
    kernel<<<dim3((unsigned)(tiles * n_tiles), (unsigned)nsplit), dim3(256), lds, st>>>(d);
    if (nsplit > 1) {
        const int64_t total = (int64_t)d.partial_rows * ((d.C_out + 7) / 8);
        splitk_reduce_kernel<DT><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(d, BM, n_tiles * BN, nsplit);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

// tile codes 4000000 + BM * 1000 + BN
template <int DT>
int dispatch_kw(gpp_conv_desc& d, hipStream_t st)
{
    switch (d.tile_hint) {
        case 4064128: return launch_kw<DT, 64, 128>(d, st);
        case 4096128: return launch_kw<DT, 96, 128>(d, st);
        case 4128128: return launch_kw<DT, 128, 128>(d, st);
        case 4160128: return launch_kw<DT, 160, 128>(d, st);
        case 4192128: return launch_kw<DT, 192, 128>(d, st);
        default: return GPP_ERR_BAD_ARG;
    }
}

}  // namespace

#endif  // GPP_CONV_KW_IMPL_H_
