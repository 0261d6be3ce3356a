// NHWC convolution as an implicit GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x32_{bf16,f16}),
// no im2col buffer, fused bias / residual (+ nearest resize) / ReLU epilogue.
//
// Replaces the Conv2D / BatchNormalization(frozen) / Activation / Add / UpsampleLike nodes of
//   /root/reference/keras_retinanet_3D/models/retinanet.py:24-205  (heads, FPN)
//   keras_resnet bottleneck stack used at models/resnet.py:88-93     (third party)
// for every layer with C_in % 64 == 0 (everything except the 3-channel stem, csrc/stem.hip).
//
// GEMM view (one launch = up to 5 feature maps sharing the weights):
//   C[M = batch*Ho*Wo pixels][N = C_out] = A[M][K] * B[K][N],  K = (kh, kw, c_in)
//   A is never materialised: for K-step (tap, 64-channel chunk) row m is the 128 contiguous
//   bytes in[b, oy*s - pt + kh, ox*s - pl + kw, c0:c0+64]  (or the zero page outside the image).
//
// Work decomposition
//   workgroup = 256 threads = 4 wavefronts (2 x 2), block tile BM x BN (128 x 128 or 128 x 64),
//   wavefront tile (BM/2) x (BN/2) as (BM/32) x (BN/32) MFMA 16x16 accumulators.
//   K-step = 64 channels of one tap; A and B tiles (128-byte rows) go HBM/L2 -> LDS with
//   global_load_lds_dwordx4 (LDS-DMA, no VGPR staging), double buffered, one barrier per K-step.
//   LDS rows are XOR-swizzled in 16-byte chunks (chunk ^= row & 7) by permuting the *source*
//   chunk each lane fetches (the LDS-DMA destination is lane-linear), which makes the
//   ds_read_b128 fragment reads conflict-free (bank = (addr/4) % 64, 16-lane groups).
//   Workgroup ids are remapped so that each XCD (private 4 MiB L2) owns a contiguous range of
//   tiles; the N-tiles of one M-tile are adjacent, so the activation rows are shared in L2.
//   Epilogue: accumulators -> LDS (per-wave private slab) -> 16-byte coalesced row stores.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpp.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int DT> struct Elem;
template <> struct Elem<GPP_BF16> {
    using scalar = __bf16;
    using vec8 = bf16x8;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Elem<GPP_F16> {
    using scalar = _Float16;
    using vec8 = f16x8;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

constexpr int kThreads = 256;
constexpr int kRowBytes = 128;     // one K-step of one tile row: 64 two-byte elements
constexpr int kEpiPitch = 68;      // floats per row of the epilogue slab (64 + 4 pad)

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// Bijective remap: blocks b and b+8 share an XCD; give each XCD a contiguous tile range.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

template <int DT, int BM, int BN>
__global__ __launch_bounds__(kThreads, 2) void conv_igemm_kernel(const gpp_conv_desc d)
{
    using E = Elem<DT>;
    using vec8 = typename E::vec8;
    using scalar = typename E::scalar;
    constexpr int MF = BM / 32, NF = BN / 32;          // 16x16 accumulators per wave: MF x NF
    constexpr int A_BYTES = BM * kRowBytes, B_BYTES = BN * kRowBytes, STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;      // LDS-DMA instructions per wave per stage
    static_assert(2 * STAGE >= 4 * 32 * kEpiPitch * 4, "epilogue slab must fit in the staging LDS");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- which tile
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
    const int cpt = d.C_in >> 6;                        // 64-channel chunks per tap
    const int nk = d.KH * d.KW * cpt;

    // ---- staging bookkeeping: this lane owns LDS chunk (row srow of each 8-row piece, slot lane&7)
    const int srow = lane >> 3;
    const int gchunk = (lane & 7) ^ srow;               // source chunk: inverse of the read swizzle
    const scalar* in = (const scalar*)d.in;
    const scalar* zero = (const scalar*)d.zero_page + gchunk * 8;
    int64_t a_base[A_IT];
    int a_iy0[A_IT], a_ix0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + wave * (BM / 4) + i * 8 + srow;
        if (m < Mg) {
            const int b = m / HoWo, p = m - b * HoWo;
            const int oy = p / W_out, ox = p - oy * W_out;
            a_iy0[i] = oy * d.stride - d.pad_top;
            a_ix0[i] = ox * d.stride - d.pad_left;
            a_base[i] = in_off + (int64_t)b * in_bs + gchunk * 8;
        } else {
            a_iy0[i] = -(1 << 28); a_ix0[i] = 0; a_base[i] = 0;
        }
    }
    const scalar* w_src[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
        w_src[i] = (const scalar*)d.weight + (int64_t)(n0 + wave * (BN / 4) + i * 8 + srow) * Ktot + gchunk * 8;

    auto stage = [&](int buf, int kh, int kw, int cc, int ks) {
        unsigned char* sa = smem + buf * STAGE + wave * (BM / 4) * kRowBytes;
        unsigned char* sb = smem + buf * STAGE + A_BYTES + wave * (BN / 4) * kRowBytes;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            const bool ok = (unsigned)iy < (unsigned)H_in && (unsigned)ix < (unsigned)W_in;
            const scalar* src = ok ? in + a_base[i] + ((int64_t)iy * W_in + ix) * d.in_pitch + cc * 64 : zero;
            glds16(src, sa + i * 8 * kRowBytes);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16(w_src[i] + (int64_t)ks * 64, sb + i * 8 * kRowBytes);
    };

    // ---- fragment read offsets (bytes inside a stage)
    const int frow = lane & 15, fq = lane >> 4;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fq) ^ (frow & 7)) << 4;
        a_rd[kk] = (wm * (BM / 2) + frow) * kRowBytes + sw;
        b_rd[kk] = A_BYTES + (wn * (BN / 2) + frow) * kRowBytes + sw;
    }

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- main loop: stage k+1 while computing k; one barrier per K-step
    int kh = 0, kw = 0, cc = 0;
    stage(0, kh, kw, cc, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) {
            if (++cc == cpt) { cc = 0; if (++kw == d.KW) { kw = 0; ++kh; } }
            stage(cur ^ 1, kh, kw, cc, ks + 1);
        }
        const unsigned char* sbase = smem + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            vec8 af[MF], bfr[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const vec8*)(sbase + a_rd[kk] + i * 16 * kRowBytes);
#pragma unroll
            for (int j = 0; j < NF; ++j) bfr[j] = *(const vec8*)(sbase + b_rd[kk] + j * 16 * kRowBytes);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = E::mfma(af[i], bfr[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: per-wave slab of 32 rows x 64 cols float32, two (MF > 2: MF/2) passes
    float* slab = (float*)smem + wave * (32 * kEpiPitch);
    const int erow = lane >> 3, ecol = (lane & 7) * 8;
    constexpr int COLS = BN / 2;                         // columns owned by this wave (64 or 32)
    const scalar* res = (const scalar*)d.residual;
    const bool resize = (H_res != H_out) || (W_res != W_out);
    const float sy = resize ? (float)H_res / (float)H_out : 1.0f;
    const float sx = resize ? (float)W_res / (float)W_out : 1.0f;
#pragma unroll
    for (int half = 0; half < MF / 2; ++half) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    slab[(i2 * 16 + fq * 4 + r) * kEpiPitch + j * 16 + frow] = acc[half * 2 + i2][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ecol < COLS) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int lr = pass * 8 + erow;
                const int m = m0 + wm * (BM / 2) + half * 32 + lr;
                const int n = n0 + wn * COLS + ecol;
                if (m < Mg && n < d.C_out) {
                    const int b = m / HoWo, p = m - b * HoWo;
                    float v[8];
                    const f32x4 v0 = *(const f32x4*)(slab + lr * kEpiPitch + ecol);
                    const f32x4 v1 = *(const f32x4*)(slab + lr * kEpiPitch + ecol + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
                    const bool full = (n + 8 <= d.C_out);
                    if (d.bias) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (full || n + e < d.C_out) v[e] += d.bias[n + e];
                    }
                    if (res) {
                        int64_t rp = p;
                        if (resize) {
                            const int oy = p / W_out, ox = p - oy * W_out;
                            const int ry = min((int)floorf((float)oy * sy), H_res - 1);
                            const int rx = min((int)floorf((float)ox * sx), W_res - 1);
                            rp = (int64_t)ry * W_res + rx;
                        }
                        const scalar* rsrc = res + res_off + (int64_t)b * res_bs + rp * d.res_pitch + n;
                        if (full) {
                            const vec8 rv = *(const vec8*)rsrc;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                        } else {
                            for (int e = 0; e < 8 && n + e < d.C_out; ++e) v[e] += (float)rsrc[e];
                        }
                    }
                    if (d.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    const int64_t o = out_off + (int64_t)b * out_bs + (int64_t)p * d.out_pitch + n;
                    if (d.out_f32) {
                        float* dst = (float*)d.out + o;
                        if (full) {
                            *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
                            *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                        } else {
                            for (int e = 0; e < 8 && n + e < d.C_out; ++e) dst[e] = v[e];
                        }
                    } else {
                        scalar* dst = (scalar*)d.out + o;
                        if (full) {
                            vec8 ov;
#pragma unroll
                            for (int e = 0; e < 8; ++e) ov[e] = (scalar)v[e];
                            *(vec8*)dst = ov;
                        } else {
                            for (int e = 0; e < 8 && n + e < d.C_out; ++e) dst[e] = (scalar)v[e];
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int DT, int BM, int BN>
int launch(const gpp_conv_desc& d, int total_tiles, hipStream_t st)
{
    constexpr int lds = 2 * (BM + BN) * kRowBytes;
    static bool configured = false;
    auto kernel = conv_igemm_kernel<DT, BM, BN>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int n_tiles = (d.C_out + BN - 1) / BN;
    kernel<<<dim3((unsigned)(total_tiles * n_tiles)), dim3(kThreads), lds, st>>>(d);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

int validate(const gpp_conv_desc& d)
{
    if (!d.in || !d.weight || !d.out || !d.zero_page) return GPP_ERR_BAD_ARG;
    if (d.dtype != GPP_BF16 && d.dtype != GPP_F16) return GPP_ERR_UNSUPPORTED;
    if (d.batch <= 0 || d.C_in <= 0 || d.C_out <= 0 || d.KH <= 0 || d.KW <= 0) return GPP_ERR_BAD_ARG;
    if (d.C_in % 64 != 0 || d.C_out % 4 != 0) return GPP_ERR_UNSUPPORTED;
    if (d.stride != 1 && d.stride != 2) return GPP_ERR_UNSUPPORTED;
    if (d.n_groups < 1 || d.n_groups > GPP_MAX_GROUPS) return GPP_ERR_BAD_ARG;
    if (d.in_pitch < d.C_in || d.in_pitch % 8 != 0) return GPP_ERR_ALIGN;
    if (d.out_pitch < d.C_out || d.out_pitch % (d.out_f32 ? 4 : 8) != 0) return GPP_ERR_ALIGN;
    if (d.residual && (d.res_pitch < d.C_out || d.res_pitch % 8 != 0)) return GPP_ERR_ALIGN;
    if (((uintptr_t)d.in | (uintptr_t)d.weight | (uintptr_t)d.out | (uintptr_t)d.zero_page | (uintptr_t)d.residual |
         (uintptr_t)d.bias) & 15)
        return GPP_ERR_ALIGN;
    for (int g = 0; g < d.n_groups; ++g) {
        const gpp_conv_group& G = d.groups[g];
        if (G.H_in <= 0 || G.W_in <= 0 || G.H_out <= 0 || G.W_out <= 0) return GPP_ERR_BAD_ARG;
        if ((G.in_off | G.in_bstride | G.out_off | G.out_bstride) % (d.out_f32 ? 4 : 8) != 0) return GPP_ERR_ALIGN;
        if ((G.in_off | G.in_bstride) % 8 != 0) return GPP_ERR_ALIGN;
        if (d.residual && ((G.res_off | G.res_bstride) % 8 != 0 || G.H_res <= 0 || G.W_res <= 0)) return GPP_ERR_ALIGN;
        if ((int64_t)d.batch * G.H_out * G.W_out >= (1LL << 31)) return GPP_ERR_UNSUPPORTED;
    }
    return GPP_OK;
}

}  // namespace

extern "C" int gpp_conv2d_flops(const gpp_conv_desc* host_desc, double* flops)
{
    if (!host_desc || !flops) return GPP_ERR_BAD_ARG;
    double f = 0.0;
    for (int g = 0; g < host_desc->n_groups; ++g)
        f += 2.0 * host_desc->batch * (double)host_desc->groups[g].H_out * host_desc->groups[g].W_out *
             host_desc->KH * host_desc->KW * (double)host_desc->C_in * host_desc->C_out;
    *flops = f;
    return GPP_OK;
}

extern "C" int gpp_conv2d_igemm(const gpp_conv_desc* host_desc, void* stream)
{
    if (!host_desc) return GPP_ERR_BAD_ARG;
    gpp_conv_desc d = *host_desc;
    int rc = validate(d);
    if (rc != GPP_OK) return rc;
    constexpr int BM = 128;
    const bool narrow = d.C_out <= 64 || (d.C_out % 128 != 0 && d.C_out % 128 <= 64);
    const int BN = narrow ? 64 : 128;
    if (d.weight_rows < ((d.C_out + BN - 1) / BN) * BN) return GPP_ERR_BAD_ARG;
    int tiles = 0;
    for (int g = 0; g < d.n_groups; ++g) {
        d.groups[g].tile_start = tiles;
        tiles += (d.batch * d.groups[g].H_out * d.groups[g].W_out + BM - 1) / BM;
    }
    hipStream_t st = (hipStream_t)stream;
    if (d.dtype == GPP_BF16) return narrow ? launch<GPP_BF16, 128, 64>(d, tiles, st) : launch<GPP_BF16, 128, 128>(d, tiles, st);
    return narrow ? launch<GPP_F16, 128, 64>(d, tiles, st) : launch<GPP_F16, 128, 128>(d, tiles, st);
}
