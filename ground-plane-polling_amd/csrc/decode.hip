// Detection decode on gfx950: sigmoid + orientation fold + score threshold compaction, then per
// image sort, greedy NMS, top-k, full box / dimension decode of the survivors and -1 padding.
// The head tensors never leave the device between the convolutions and the polling kernel.
//
// Replaces (citations relative to /root/reference/keras_retinanet_3D):
//   Activation('sigmoid')                         models/retinanet.py:72-73
//   RegressBoxes.call + bbox_transform_inv        layers/_misc.py:133-141, backend/common.py:43-81
//   RegressDims + dim_transform_inv               layers/_misc.py:186-187, backend/common.py:23-40
//   filter_detections (default path: nms=True, class_specific_filter=True,
//   orientation_specific_filter=False, num_classes == 1)   layers/filter_detections.py:18-189
//   incl. tf.image.non_max_suppression, tf.nn.top_k, tf.pad
//
// Exactness: this file is compiled with -ffp-contract=off and evaluates the same float32
// operation sequence as oracle/decode_np.py, including a Cephes-style expf (the algorithm of
// Eigen's packet exp that TF's CPU sigmoid uses), so scores, boxes and the NMS decisions agree
// with the oracle bit for bit.  Ordering rules: candidates by (score desc, anchor index asc);
// NMS suppresses IoU > threshold (strict), zero-area boxes never overlap.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

#include <atomic>

#include "gpp.h"

namespace {

constexpr int kCounterStride = 4096;              // bytes between the candidate counters of two images
constexpr int kHeaderBytes = 64 * kCounterStride;  // up to 64 images per call
// inside an image's header slot: word 0 = candidate counter, word 1 = number of NMS survivors, and from byte 1024 the
// survivors' keys (<= 128 x 8 bytes), written by nms_kernel for emit_kernel
constexpr int kKeptCountWord = 1;
constexpr int kKeptKeysOffset = 1024;

// ---- Cephes expf, one float32 operation at a time (mirrors oracle/decode_np.py:cephes_expf)
__device__ __forceinline__ float cephes_expf(float x)
{
    x = fminf(fmaxf(x, -88.3762626647949f), 88.3762626647949f);
    float fx = floorf(x * 1.44269504088896341f + 0.5f);
    x = (x - fx * 0.693359375f) - fx * -2.12194440e-4f;
    const float z = x * x;
    float y = 1.9875691500e-4f;
    y = y * x + 1.3981999507e-3f;
    y = y * x + 8.3334519073e-3f;
    y = y * x + 4.1665795894e-2f;
    y = y * x + 1.6666665459e-1f;
    y = y * x + 5.0000001201e-1f;
    y = (y * z + x) + 1.0f;
    return ldexpf(y, (int)fx);
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + cephes_expf(-x)); }

struct Folded { float score; int orient; float sign; };

// filter_detections.py:78-82,123-125 and _misc.py:134-136 on the 8 sigmoid scores of one anchor
__device__ __forceinline__ Folded fold8(const float* l)
{
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = sigmoidf(l[k]);
    int am = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) if (s[k] > s[am]) am = k;          // first maximum
    Folded f;
    f.sign = (am < 4) ? -1.0f : 1.0f;
    float best = fmaxf(s[0], s[4]);
    int bo = 0;
#pragma unroll
    for (int o = 1; o < 4; ++o) {
        const float v = fmaxf(s[o], s[o + 4]);
        if (v > best) { best = v; bo = o; }
    }
    f.score = best;
    f.orient = bo;
    return f;
}

__global__ __launch_bounds__(256) void candidates_kernel(const float* __restrict__ cls, int64_t n_anchors, int64_t key_stride,
                                                         float thr, unsigned long long* __restrict__ keys,
                                                         int32_t* __restrict__ counts)
{
    const int b = blockIdx.y;
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= n_anchors) return;
    const float4* src = (const float4*)(cls + ((int64_t)b * n_anchors + a) * 8);
    const float4 v0 = src[0], v1 = src[1];
    const float l[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    // cheap exact rejects (the folded score is the largest of the 8 sigmoids, and the sigmoid used here is
    // accurate to ~1e-7 and monotone to within that): (1) sigmoid(-3.0) = 0.0474 decides every threshold
    // >= 0.048 without any exp; (2) one sigmoid of the largest logit with a 1e-5 guard band decides the rest.
    // Only anchors inside the guard band or above the threshold pay for the exact 8-sigmoid fold.
    const float lmax = fmaxf(fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3])), fmaxf(fmaxf(l[4], l[5]), fmaxf(l[6], l[7])));
    if (thr >= 0.048f && lmax < -3.0f) return;
    if (sigmoidf(lmax) < thr - 1e-5f) return;
    const Folded f = fold8(l);
    if (f.score > thr) {
        const int slot = atomicAdd(&counts[b * (kCounterStride / 4)], 1);
        keys[(int64_t)b * key_stride + slot] =
            ((unsigned long long)__float_as_uint(f.score) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)a);
    }
}

// orientation_specific_filter=True (filter_detections.py:84-98): one candidate list per (image, orientation), keyed by that
// orientation's folded score max(s[o], s[o + 4]); an anchor can enter up to four lists.  Counter / key list / header slot
// of list (b, o) = 4 * b + o.
__global__ __launch_bounds__(256) void candidates_osf_kernel(const float* __restrict__ cls, int64_t n_anchors, int64_t key_stride,
                                                             float thr, unsigned long long* __restrict__ keys,
                                                             int32_t* __restrict__ counts)
{
    const int b = blockIdx.y;
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= n_anchors) return;
    const float4* src = (const float4*)(cls + ((int64_t)b * n_anchors + a) * 8);
    const float4 v0 = src[0], v1 = src[1];
    const float l[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    const float lmax = fmaxf(fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3])), fmaxf(fmaxf(l[4], l[5]), fmaxf(l[6], l[7])));
    if (thr >= 0.048f && lmax < -3.0f) return;               // the same exact rejects as candidates_kernel: no orientation can pass
    if (sigmoidf(lmax) < thr - 1e-5f) return;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = sigmoidf(l[k]);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float so = fmaxf(s[o], s[o + 4]);
        if (so > thr) {
            const int list = 4 * b + o;
            const int slot = atomicAdd(&counts[list * (kCounterStride / 4)], 1);
            keys[(int64_t)list * key_stride + slot] =
                ((unsigned long long)__float_as_uint(so) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)a);
        }
    }
}

__constant__ float kBoxMean[12] = {-0.0373f, -0.0165f, 0.0373f, 0.0171f, -0.0286f, -0.0478f,
                                   0.2929f, 0.0114f, 0.0288f, -0.0589f, 0.2932f, -0.0007f};   // _misc.py:115
__constant__ float kBoxStd[12] = {0.1957f, 0.1896f, 0.1957f, 0.1897f, 0.1967f, 0.2034f,
                                  0.2046f, 0.1898f, 0.1964f, 0.2052f, 0.2048f, 0.1903f};      // _misc.py:117
__constant__ float kDimMean[3] = {1.6570f, 1.7999f, 4.2907f};                                   // _misc.py:168
__constant__ float kDimStd[3] = {0.2681f, 0.2243f, 0.6281f};                                    // _misc.py:170

struct Layout { int fused; int nba; };   // fused: regression stored (P, [4A | 2A | 2A | 2A | 2A]) per pixel

// regression value j (0..11) of anchor a of image b
__device__ __forceinline__ float reg_at(const float* reg, Layout L, int64_t n_anchors, int b, int64_t a, int j)
{
    if (!L.fused) return reg[((int64_t)b * n_anchors + a) * 12 + j];
    const int64_t p = a / L.nba;
    const int k = (int)(a - p * L.nba);
    const int A = L.nba;
    const int c = (j < 4) ? (k * 4 + j) : (4 * A + ((j - 4) >> 1) * 2 * A + k * 2 + ((j - 4) & 1));
    return reg[((int64_t)b * (n_anchors / A) + p) * (12 * A) + c];
}

// backend/common.py:62-77
__device__ __forceinline__ float box_coord(int j, const float4 an, float delta, float sign)
{
    const float w = an.z - an.x, h = an.w - an.y;
    const float t = delta * kBoxStd[j] + kBoxMean[j];
    switch (j) {
    case 0: return an.x + t * w;
    case 1: return an.y + t * h;
    case 2: return an.z + t * w;
    case 3: return an.w + t * h;
    case 4: return an.x + t * w;
    case 5: return an.w + t * h;
    case 6: return (an.x + an.z) / 2.0f + (t * w) * sign;
    case 7: return an.w + t * h;
    case 8: return an.z + t * w;
    case 9: return an.w + t * h;
    case 10: return (an.x + an.z) / 2.0f + (t * w) * sign;
    default: return an.y + t * h;
    }
}

// tf.image.non_max_suppression's overlap test (corners min/max-normalised, zero area -> no overlap)
__device__ __forceinline__ bool iou_above(const float4 a, const float4 b, float thr)
{
    const float ay0 = fminf(a.x, a.z), ax0 = fminf(a.y, a.w), ay1 = fmaxf(a.x, a.z), ax1 = fmaxf(a.y, a.w);
    const float by0 = fminf(b.x, b.z), bx0 = fminf(b.y, b.w), by1 = fmaxf(b.x, b.z), bx1 = fmaxf(b.y, b.w);
    const float area_a = (ay1 - ay0) * (ax1 - ax0);
    const float area_b = (by1 - by0) * (bx1 - bx0);
    if (area_a <= 0.0f || area_b <= 0.0f) return false;
    const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.0f);
    const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.0f);
    const float inter = ih * iw;
    return inter / ((area_a + area_b) - inter) > thr;
}

constexpr int kNmsThreads = 1024;
constexpr int kLdsKeys = 8192;

// Bitonic sort, descending, n a power of two, all kNmsThreads threads.  `wave_local`: passes whose
// stride is <= 64 keep every wavefront inside its own 128-element block (pair t lives in block t/64), so
// between two such passes a wavefront-level ordering is enough; block barriers are only needed around the
// wider passes (LDS arrays only: global memory keeps a barrier per pass).
__device__ __forceinline__ void bitonic_desc(unsigned long long* k, int n, int tid, bool wave_local)
{
    int prev = 1 << 30;
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (!wave_local || stride >= 128 || prev >= 128) __syncthreads();
            else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            prev = stride;
            for (int t = tid; t < (n >> 1); t += kNmsThreads) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = k[lo], b = k[hi];
                if ((a < b) == desc) { k[lo] = b; k[hi] = a; }
            }
        }
    }
    __syncthreads();
}

constexpr int kHistBins = 4096;

// bin of a key for the top-T selection: monotone in the score (high 32 bits = float bits of a positive score)
__device__ __forceinline__ int score_bin(unsigned long long key)
{
    const int v = ((int)(key >> 32) - 0x3D000000) >> 14;          // scores in [2^-5, 1] -> 0 .. 2560
    return v < 0 ? 0 : (v > kHistBins - 1 ? kHistBins - 1 : v);
}

struct NmsShared {
    int nxt[3];
    int kept[128];
    int cut, nsel, wsum[kNmsThreads / 64];
    int wmin[2][kNmsThreads / 64];
};

// Greedy NMS over candidates 0..K-1 (already in score order, corners in boxes4, alive = 1), one block
// barrier per kept box: while suppressing against box i every thread also tracks the lowest surviving
// index it sees; the minimum over the workgroup (wave shuffle + one LDS atomicMin per wavefront, into a
// rotating word) is the next box.  Returns the number of boxes kept (indices in sh.kept).
__device__ __forceinline__ int greedy_nms(const float4* boxes4, unsigned char* alive, int K, float iou_thr, int max_det,
                                          NmsShared& sh, int tid)
{
    if (tid == 0) { sh.nxt[0] = K > 0 ? 0 : INT_MAX; sh.nxt[1] = INT_MAX; sh.nxt[2] = INT_MAX; }
    __syncthreads();
    int kept = 0;
    for (int it = 0; kept < max_det; ++it) {
        const int i = sh.nxt[it % 3];
        if (i >= K) break;
        if (tid == 0) { sh.kept[kept] = i; sh.nxt[(it + 2) % 3] = INT_MAX; }
        ++kept;
        const float4 bi = boxes4[i];
        int lmin = INT_MAX;
        for (int j = i + 1 + tid; j < K; j += kNmsThreads) {
            if (alive[j]) {
                if (iou_above(bi, boxes4[j], iou_thr)) alive[j] = 0;
                else lmin = min(lmin, j);
            }
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) lmin = min(lmin, __shfl_xor(lmin, s, 64));
        if ((tid & 63) == 0 && lmin != INT_MAX) atomicMin(&sh.nxt[(it + 1) % 3], lmin);
        __syncthreads();
    }
    return kept;
}

// The same decision as iou_above for boxes whose corners are already min/max-normalised (y0 x0 y1 x1 in
// .x .y .z .w) and whose areas are known.  The division is only evaluated inside a 2^-19 guard band around
// the threshold: outside it the sign of inter - thr*union decides (a float32 quotient is within 2^-24 of
// the exact ratio and thr*union within 2^-24 of the exact product), so the result is bit-for-bit the
// reference decision at about half the instructions.
__device__ __forceinline__ bool iou_above_norm(const float4 a, float area_a, const float4 b, float area_b, float thr)
{
    if (area_a <= 0.0f || area_b <= 0.0f) return false;
    const float ih = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f);
    const float iw = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
    const float inter = ih * iw;
    const float uni = (area_a + area_b) - inter;
    const float tu = thr * uni;
    if (uni > 0.0f) {
        if (inter > tu * 1.000002f) return true;
        if (inter < tu * 0.999998f) return false;
    }
    return inter / uni > thr;
}

__device__ __forceinline__ float4 normalise(const float4 v)
{
    return make_float4(fminf(v.x, v.z), fminf(v.y, v.w), fmaxf(v.x, v.z), fmaxf(v.y, v.w));
}

constexpr int kOwn = kLdsKeys / kNmsThreads;          // candidates owned by one thread on the LDS path (8)

// Greedy NMS with everything on chip, run by T threads (the whole workgroup):
// thread t owns candidates t, t + T, ... (corners in registers, one alive bit each); all corners also sit in
// LDS so that the box being kept can be broadcast.  Per kept box: one LDS broadcast read, the IoU tests, a
// ballot per owned slot (the lowest surviving index of a wavefront is its lowest set lane), one LDS word per
// wavefront, ONE barrier.  No global memory traffic inside the loop.
__device__ __forceinline__ int greedy_nms_lds(const float4* lbox, const float4 (&ob)[kOwn], int K, int T, float iou_thr,
                                              int max_det, NmsShared& sh, int tid)
{
    const int wave = tid >> 6, lane = tid & 63, nw = T >> 6;
    float oarea[kOwn];                                     // ob / lbox hold normalised corners
#pragma unroll
    for (int q = 0; q < kOwn; ++q) oarea[q] = (ob[q].z - ob[q].x) * (ob[q].w - ob[q].y);
    unsigned alive = 0xFFu;
    int kept = 0, i = K > 0 ? 0 : INT_MAX;
    for (int it = 0; i < K; ++it) {
        if (tid == 0) sh.kept[kept] = i;
        if (++kept >= max_det) break;
        const float4 bi = lbox[i];
        const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
        int wmin = INT_MAX;
#pragma unroll
        for (int q = 0; q < kOwn; ++q) {
            if (q * T >= K) break;                         // (uniform) no candidate lives in this slot or beyond
            const int j = tid + q * T;
            bool cand = j > i && j < K && (alive & (1u << q));
            if (cand && iou_above_norm(bi, area_i, ob[q], oarea[q], iou_thr)) { alive &= ~(1u << q); cand = false; }
            const unsigned long long m = __ballot(cand);
            if (m != 0ull && wmin == INT_MAX) wmin = wave * 64 + (__ffsll((long long)m) - 1) + q * T;
        }
        if (lane == 0) sh.wmin[it & 1][wave] = wmin;
        __syncthreads();
        i = INT_MAX;
        for (int w = 0; w < nw; ++w) i = min(i, sh.wmin[it & 1][w]);
    }
    return kept;
}

__global__ __launch_bounds__(kNmsThreads) void nms_kernel(
    unsigned long long* __restrict__ keys, const int32_t* __restrict__ counts, int64_t key_stride,
    const float* __restrict__ cls, const float* __restrict__ reg, const float* __restrict__ regdim,
    const float4* __restrict__ anchors, int64_t n_anchors, Layout L, float iou_thr, int max_det,
    int32_t* __restrict__ o_counts, float4* __restrict__ ws_boxes, unsigned char* __restrict__ ws_alive,
    unsigned char* __restrict__ header, int lists_per_image)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
    unsigned long long* lkeys = (unsigned long long*)nms_smem;                 // kLdsKeys keys while sorting ...
    float4* lbox = (float4*)nms_smem;                                          // ... then kLdsKeys corner boxes
    int* hist = (int*)(nms_smem + (size_t)kLdsKeys * 16);                      // kHistBins counters
    __shared__ NmsShared sh;

    // one workgroup per candidate list: list = image (default) or 4 * image + orientation (orientation_specific_filter)
    const int list = blockIdx.x, tid = threadIdx.x;
    const int b = list / lists_per_image;
    const int K = min(counts[list * (kCounterStride / 4)], (int)n_anchors);
    if (o_counts && tid == 0 && lists_per_image == 1) o_counts[b] = K;
    unsigned long long* gkeys = keys + (int64_t)list * key_stride;
    float4* boxes4 = ws_boxes + (int64_t)list * n_anchors;
    unsigned char* alive = ws_alive + (int64_t)list * n_anchors;

    // corners x1 y1 x2 y2 of the first `cnt` sorted candidates (filter_detections.py:58,61 uses boxes[:, :4])
    auto prepare = [&](const unsigned long long* sk, int cnt) {
        for (int t = tid; t < cnt; t += kNmsThreads) {
            const unsigned long long key = sk[t];
            const int64_t a = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
            const float4 an = anchors[a];
            float4 bx;
            bx.x = box_coord(0, an, reg_at(reg, L, n_anchors, b, a, 0), 0.0f);
            bx.y = box_coord(1, an, reg_at(reg, L, n_anchors, b, a, 1), 0.0f);
            bx.z = box_coord(2, an, reg_at(reg, L, n_anchors, b, a, 2), 0.0f);
            bx.w = box_coord(3, an, reg_at(reg, L, n_anchors, b, a, 3), 0.0f);
            boxes4[t] = bx;
            alive[t] = 1;
        }
        __syncthreads();
    };

    // On-chip path for <= kLdsKeys sorted keys sitting in lkeys: every thread takes its own candidates'
    // keys into registers, the sorted keys go to global memory (boxes4's space, unused on this path) for the
    // output phase, the LDS is reused for the corner boxes, then the all-on-chip greedy loop runs.
    unsigned long long* skeys = (unsigned long long*)boxes4;
    auto lds_nms = [&](int cnt, int T) -> int {
        unsigned long long mykey[kOwn];
#pragma unroll
        for (int q = 0; q < kOwn; ++q) {
            const int j = tid + q * T;
            mykey[q] = j < cnt ? lkeys[j] : 0ull;
        }
        __syncthreads();                                       // lkeys is dead from here: its LDS becomes lbox
        float4 ob[kOwn];
#pragma unroll
        for (int q = 0; q < kOwn; ++q) {
            const int j = tid + q * T;
            ob[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < cnt) {
                const int64_t a = (int64_t)(0xFFFFFFFFu - (uint32_t)(mykey[q] & 0xFFFFFFFFull));
                const float4 an = anchors[a];
                ob[q].x = box_coord(0, an, reg_at(reg, L, n_anchors, b, a, 0), 0.0f);
                ob[q].y = box_coord(1, an, reg_at(reg, L, n_anchors, b, a, 1), 0.0f);
                ob[q].z = box_coord(2, an, reg_at(reg, L, n_anchors, b, a, 2), 0.0f);
                ob[q].w = box_coord(3, an, reg_at(reg, L, n_anchors, b, a, 3), 0.0f);
                ob[q] = normalise(ob[q]);
                lbox[j] = ob[q];
                skeys[j] = mykey[q];
            }
        }
        __syncthreads();
        return greedy_nms_lds(lbox, ob, cnt, T, iou_thr, max_det, sh, tid);
    };

    // ---- candidates in (score desc, anchor asc) order
    const unsigned long long* sk = skeys;
    int kept = 0;
    bool done = false;
    if (K <= kLdsKeys) {
        int n = 1;
        while (n < K) n <<= 1;
        for (int t = tid; t < n; t += kNmsThreads) lkeys[t] = (t < K) ? gkeys[t] : 0ull;
        bitonic_desc(lkeys, n, tid, true);
        // (measured: letting wavefronts exit and giving each survivor more candidates is slower -- the loop
        // is bound by single-wavefront instruction issue, so all 16 wavefronts stay)
        const int T = kNmsThreads;
        kept = lds_nms(K, T);
        done = true;
    } else {
        // More candidates than the LDS holds.  Only the first max_det survivors matter and NMS walks the
        // candidates in score order, so take the best-scoring ones first: histogram the scores, cut at the
        // lowest bin for which everything above still fits, sort + NMS that prefix; if it already yields
        // max_det boxes the rest of the list cannot change the result.
        for (int t = tid; t < kHistBins; t += kNmsThreads) hist[t] = 0;
        __syncthreads();
        for (int t = tid; t < K; t += kNmsThreads) atomicAdd(&hist[score_bin(gkeys[t])], 1);
        __syncthreads();
        // suffix counts over bins: thread t owns bins 4t .. 4t+3
        int own[4], mine = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { own[q] = hist[4 * tid + q]; mine += own[q]; }
        int suf = mine;                                        // inclusive suffix sum inside the wavefront
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int o = __shfl_down(suf, s, 64);
            if ((tid & 63) + s < 64) suf += o;
        }
        if ((tid & 63) == 0) sh.wsum[tid >> 6] = suf;
        if (tid == 0) { sh.cut = kHistBins; sh.nsel = 0; }
        __syncthreads();
        int above = suf - mine;                                // keys in higher bins of this wavefront ...
        for (int w = (tid >> 6) + 1; w < kNmsThreads / 64; ++w) above += sh.wsum[w];   // ... and of later wavefronts
        if (above <= kLdsKeys && above + mine > kLdsKeys) {    // the cut falls inside this thread's bins
            int acc = above, cut = 4 * tid + 4;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                if (acc + own[q] <= kLdsKeys) { acc += own[q]; cut = 4 * tid + q; } else break;
            }
            sh.cut = cut;
            sh.nsel = acc;
        }
        __syncthreads();
        const int cut = sh.cut, nsel = sh.nsel;
        if (nsel > 0) {
            if (tid == 0) sh.wsum[0] = 0;
            __syncthreads();
            for (int t = tid; t < K; t += kNmsThreads) {
                const unsigned long long key = gkeys[t];
                if (score_bin(key) >= cut) lkeys[atomicAdd(&sh.wsum[0], 1)] = key;
            }
            __syncthreads();
            int n = 1;
            while (n < nsel) n <<= 1;
            for (int t = nsel + tid; t < n; t += kNmsThreads) lkeys[t] = 0ull;
            bitonic_desc(lkeys, n, tid, true);
            kept = lds_nms(nsel, kNmsThreads);
            done = (kept >= max_det);
            __syncthreads();
        }
        if (!done) {
            // rare: the prefix did not yield max_det boxes (or one score bin alone overflows the LDS):
            // sort everything in global memory and start over
            int n = 1;
            while (n < K) n <<= 1;
            for (int t = K + tid; t < n; t += kNmsThreads) gkeys[t] = 0ull;
            bitonic_desc(gkeys, n, tid, false);
            sk = gkeys;
            prepare(gkeys, K);
            kept = greedy_nms(boxes4, alive, K, iou_thr, max_det, sh, tid);
        }
    }

    // ---- hand-over to emit_kernel: the keys of the survivors, in score order, next to this image's counter
    __syncthreads();
    unsigned char* slot = header + (int64_t)list * kCounterStride;
    if (tid == 0) ((int32_t*)slot)[kKeptCountWord] = kept;
    for (int t = tid; t < kept; t += kNmsThreads) ((unsigned long long*)(slot + kKeptKeysOffset))[t] = sk[sh.kept[t]];
}

// Outputs of one image from the survivors' keys: full box / dimension decode, scores re-derived from the logits,
// in score order (tf.nn.top_k of an already sorted list is the identity), then -1 padding
// (filter_detections.py:155-177).  A separate launch so that the selection above only depends on the
// classification logits and the four corner regressions: in a plan it runs on a side stream underneath the
// dimension tower, and this kernel joins once every head tensor is there.
__global__ __launch_bounds__(128) void emit_kernel(
    const unsigned char* __restrict__ header, const float* __restrict__ cls, const float* __restrict__ reg,
    const float* __restrict__ regdim, const float4* __restrict__ anchors, int64_t n_anchors, Layout L, int max_det,
    float* __restrict__ o_boxes, float* __restrict__ o_dims, float* __restrict__ o_scores,
    int32_t* __restrict__ o_labels, int32_t* __restrict__ o_orient, int32_t* __restrict__ o_anchor)
{
    const int b = blockIdx.x;
    const unsigned char* slot = header + (int64_t)b * kCounterStride;
    const int kept = ((const int32_t*)slot)[kKeptCountWord];
    const unsigned long long* kk = (const unsigned long long*)(slot + kKeptKeysOffset);
    for (int t = threadIdx.x; t < max_det; t += 128) {
        float* ob = o_boxes + ((int64_t)b * max_det + t) * 12;
        float* od = o_dims + ((int64_t)b * max_det + t) * 3;
        const int64_t row = (int64_t)b * max_det + t;
        if (t < kept) {
            const unsigned long long key = kk[t];
            const int64_t a = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
            const float4* src = (const float4*)(cls + ((int64_t)b * n_anchors + a) * 8);
            const float4 v0 = src[0], v1 = src[1];
            const float l[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const Folded f = fold8(l);
            const float4 an = anchors[a];
#pragma unroll
            for (int j = 0; j < 12; ++j) ob[j] = box_coord(j, an, reg_at(reg, L, n_anchors, b, a, j), f.sign);
#pragma unroll
            for (int j = 0; j < 3; ++j) od[j] = regdim[((int64_t)b * n_anchors + a) * 3 + j] * kDimStd[j] + kDimMean[j];
            o_scores[row] = f.score;
            o_labels[row] = 0;
            o_orient[row] = f.orient;
            if (o_anchor) o_anchor[row] = (int32_t)a;
        } else {
#pragma unroll
            for (int j = 0; j < 12; ++j) ob[j] = -1.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) od[j] = -1.0f;
            o_scores[row] = -1.0f;
            o_labels[row] = -1;
            o_orient[row] = -1;
            if (o_anchor) o_anchor[row] = -1;
        }
    }
}

// orientation_specific_filter: the four survivor lists of an image (each in score order) are concatenated in orientation
// order and tf.nn.top_k picks max_det of them: descending score, earlier position first on ties (filter_detections.py:152-167).
// Rank by counting; the orientation of an entry is the list it came from, its score is the key's.
__global__ __launch_bounds__(512) void emit_osf_kernel(
    const unsigned char* __restrict__ header, const float* __restrict__ cls, const float* __restrict__ reg,
    const float* __restrict__ regdim, const float4* __restrict__ anchors, int64_t n_anchors, Layout L, int max_det,
    float* __restrict__ o_boxes, float* __restrict__ o_dims, float* __restrict__ o_scores,
    int32_t* __restrict__ o_labels, int32_t* __restrict__ o_orient, int32_t* __restrict__ o_anchor, int32_t* __restrict__ o_counts)
{
    __shared__ unsigned long long s_key[512];
    __shared__ int s_total;
    const int b = blockIdx.x, tid = threadIdx.x;
    int cnt[4], off[5];
    off[0] = 0;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        cnt[o] = ((const int32_t*)(header + (int64_t)(4 * b + o) * kCounterStride))[kKeptCountWord];
        off[o + 1] = off[o] + cnt[o];
    }
    const int total = off[4];                                   // <= 4 * 128
    int my_o = -1;
    unsigned long long my_key = 0ull;
    if (tid < total) {
        my_o = (tid >= off[3]) ? 3 : (tid >= off[2]) ? 2 : (tid >= off[1]) ? 1 : 0;
        my_key = ((const unsigned long long*)(header + (int64_t)(4 * b + my_o) * kCounterStride + kKeptKeysOffset))[tid - off[my_o]];
    }
    s_key[tid] = my_key;
    if (tid == 0) {
        s_total = total;
        if (o_counts) {                                         // candidates of the image = sum over its four lists
            int c = 0;
            for (int o = 0; o < 4; ++o) c += ((const int32_t*)(header + (int64_t)(4 * b + o) * kCounterStride))[0];
            o_counts[b] = c;
        }
    }
    __syncthreads();
    if (tid < total) {
        const uint32_t mine = (uint32_t)(my_key >> 32);
        int rank = 0;
        for (int q = 0; q < total; ++q) {
            const uint32_t other = (uint32_t)(s_key[q] >> 32);  // positive floats: the bit patterns order like the values
            rank += (other > mine || (other == mine && q < tid)) ? 1 : 0;
        }
        if (rank < max_det) {
            const int64_t row = (int64_t)b * max_det + rank;
            const int64_t a = (int64_t)(0xFFFFFFFFu - (uint32_t)(my_key & 0xFFFFFFFFull));
            const float4* src = (const float4*)(cls + ((int64_t)b * n_anchors + a) * 8);
            const float4 v0 = src[0], v1 = src[1];
            const float l[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const Folded f = fold8(l);                          // only its sign is used: RegressBoxes does not know the list
            const float4 an = anchors[a];
            float* ob = o_boxes + row * 12;
            float* od = o_dims + row * 3;
#pragma unroll
            for (int j = 0; j < 12; ++j) ob[j] = box_coord(j, an, reg_at(reg, L, n_anchors, b, a, j), f.sign);
#pragma unroll
            for (int j = 0; j < 3; ++j) od[j] = regdim[((int64_t)b * n_anchors + a) * 3 + j] * kDimStd[j] + kDimMean[j];
            o_scores[row] = __uint_as_float(mine);
            o_labels[row] = 0;
            o_orient[row] = my_o;
            if (o_anchor) o_anchor[row] = (int32_t)a;
        }
    }
    for (int t = min(total, max_det) + tid; t < max_det; t += 512) {
        const int64_t row = (int64_t)b * max_det + t;
#pragma unroll
        for (int j = 0; j < 12; ++j) o_boxes[row * 12 + j] = -1.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) o_dims[row * 3 + j] = -1.0f;
        o_scores[row] = -1.0f;
        o_labels[row] = -1;
        o_orient[row] = -1;
        if (o_anchor) o_anchor[row] = -1;
    }
}

inline int64_t pow2_ceil(int64_t v)
{
    int64_t n = 1;
    while (n < v) n <<= 1;
    return n;
}

}  // namespace

namespace {

// (beyond these the workspace arithmetic would overflow long before any buffer could exist: 2^28 anchors are 2 000 images' worth)
constexpr int kMaxBatch = 1 << 16;
constexpr int64_t kMaxAnchors = (int64_t)1 << 28;

size_t detect_bytes(int B, int64_t n_anchors, int lists_per_image)
{
    const size_t keys = (size_t)pow2_ceil(n_anchors) * 8;
    const size_t boxes = (size_t)n_anchors * 16;
    const size_t alive = ((size_t)n_anchors + 15) / 16 * 16;
    return kHeaderBytes + (size_t)B * lists_per_image * (keys + boxes + alive);
}

// Zero the per-list candidate counters (one int per kCounterStride bytes).  A launch of our own instead of hipMemsetAsync:
// the runtime's fill kernel covered the whole 32 KiB header region and showed up at ~116 us per call in the profile.
__global__ void clear_counters_kernel(int32_t* cnt, int lists)
{
    if ((int)threadIdx.x < lists) cnt[threadIdx.x * (kCounterStride / 4)] = 0;
}

int detect_impl(int stages, int lists_per_image, const float* cls_logits, const float* regression, const float* regression_dim,
                const float* anchors, int B, int64_t n_anchors, int num_base_anchors, int fused_layout, float score_thr,
                float iou_thr, int max_det, float* boxes, float* dims, float* scores, int32_t* labels, int32_t* orientations,
                int32_t* anchor_index, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream)
{
    if (stages <= 0 || stages > 7) return GPP_ERR_BAD_ARG;
    if (B < 0 || B > kMaxBatch || n_anchors <= 0 || n_anchors > kMaxAnchors || max_det <= 0 || max_det > 128 || num_base_anchors <= 0) return GPP_ERR_BAD_ARG;
    if (n_anchors >= (1LL << 31) || n_anchors % num_base_anchors != 0) return GPP_ERR_UNSUPPORTED;
    if (B == 0) return GPP_OK;
    if (!cls_logits || !regression || !regression_dim || !anchors || !boxes || !dims || !scores || !labels ||
        !orientations || !workspace)
        return GPP_ERR_BAD_ARG;
    if (((uintptr_t)cls_logits | (uintptr_t)anchors | (uintptr_t)workspace) & 15) return GPP_ERR_ALIGN;
    if (workspace_bytes < detect_bytes(B, n_anchors, lists_per_image)) return GPP_ERR_WORKSPACE;
    const int lists = B * lists_per_image;
    if (lists > 64) return GPP_ERR_UNSUPPORTED;            // header slots
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    // per-list candidate counters, one per kCounterStride bytes: adjacent counters share an L2 channel and
    // their returning atomics serialise (measured: 66 us for 8 x 800 candidates on one cache line)
    int32_t* cnt = (int32_t*)ws;
    const int64_t kstride = pow2_ceil(n_anchors);
    unsigned long long* keys = (unsigned long long*)(ws + kHeaderBytes);
    float4* wboxes = (float4*)(ws + kHeaderBytes + (size_t)lists * kstride * 8);
    unsigned char* alive = (unsigned char*)(wboxes + (size_t)lists * n_anchors);
    Layout L = {fused_layout, num_base_anchors};
    const bool osf = lists_per_image == 4;
    hipError_t e;
    if (stages & GPP_DETECT_CANDIDATES) {                  // needs cls_logits only
        clear_counters_kernel<<<1, 64, 0, st>>>(cnt, lists);                 // the header slots of this call's lists
        const dim3 grid((unsigned)((n_anchors + 255) / 256), (unsigned)B);
        if (osf) candidates_osf_kernel<<<grid, 256, 0, st>>>(cls_logits, n_anchors, kstride, score_thr, keys, cnt);
        else candidates_kernel<<<grid, 256, 0, st>>>(cls_logits, n_anchors, kstride, score_thr, keys, cnt);
    }
    if (stages & GPP_DETECT_SELECT) {                      // needs the candidates and the corner regressions
        // the dynamic-LDS attribute is per device: one bit per device ordinal (racing first calls set the same value)
        static std::atomic<uint64_t> configured{0};
        int dev = 0;
        e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (!(configured.load(std::memory_order_acquire) & (1ull << (dev & 63)))) {
            e = hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsKeys * 16 + kHistBins * 4);
            if (e != hipSuccess) return (int)e;
            configured.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        nms_kernel<<<dim3((unsigned)lists), kNmsThreads, kLdsKeys * 16 + kHistBins * 4, st>>>(
            keys, cnt, kstride, cls_logits, regression, regression_dim, (const float4*)anchors, n_anchors, L, iou_thr, max_det,
            counts, wboxes, alive, ws, lists_per_image);
    }
    if (stages & GPP_DETECT_EMIT) {                        // needs every head tensor
        if (osf)
            emit_osf_kernel<<<dim3((unsigned)B), 512, 0, st>>>(ws, cls_logits, regression, regression_dim, (const float4*)anchors,
                                                               n_anchors, L, max_det, boxes, dims, scores, labels, orientations,
                                                               anchor_index, counts);
        else
            emit_kernel<<<dim3((unsigned)B), 128, 0, st>>>(ws, cls_logits, regression, regression_dim, (const float4*)anchors,
                                                           n_anchors, L, max_det, boxes, dims, scores, labels, orientations, anchor_index);
    }
    e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

}  // namespace

namespace {

// The eight result arrays of one image batch -> one (B, D, 35) float32 tensor (what the ranks exchange);
// labels / orientations are small integers, exact in float32.
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ boxes, const float* __restrict__ dims,
                                                   const float* __restrict__ scores, const int32_t* __restrict__ labels,
                                                   const int32_t* __restrict__ orient, const float* __restrict__ keypoints,
                                                   const float* __restrict__ keyplanes, const float* __restrict__ residuals,
                                                   float* __restrict__ out, int64_t rows)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * 35) return;
    const int64_t r = e / 35;
    const int c = (int)(e - r * 35);
    float v;
    if (c < 12) v = boxes[r * 12 + c];
    else if (c < 15) v = dims[r * 3 + (c - 12)];
    else if (c == 15) v = scores[r];
    else if (c == 16) v = (float)labels[r];
    else if (c == 17) v = (float)orient[r];
    else if (c < 30) v = keypoints[r * 12 + (c - 18)];
    else if (c < 34) v = keyplanes[r * 4 + (c - 30)];
    else v = residuals[r];
    out[e] = v;
}

}  // namespace

extern "C" int gpp_pack_detections(const float* boxes, const float* dims, const float* scores, const int32_t* labels,
                                   const int32_t* orientations, const float* keypoints, const float* keyplanes,
                                   const float* residuals, int B, int D, float* packed, void* stream)
{
    if (B < 0 || D < 0) return GPP_ERR_BAD_ARG;
    if (B == 0 || D == 0) return GPP_OK;
    if (!boxes || !dims || !scores || !labels || !orientations || !keypoints || !keyplanes || !residuals || !packed) return GPP_ERR_BAD_ARG;
    const int64_t rows = (int64_t)B * D;
    pack_kernel<<<dim3((unsigned)((rows * 35 + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
        boxes, dims, scores, labels, orientations, keypoints, keyplanes, residuals, packed, rows);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}

extern "C" int gpp_detect_workspace_bytes(int B, int64_t n_anchors, size_t* bytes)
{
    if (!bytes || B < 0 || B > kMaxBatch || n_anchors <= 0 || n_anchors > kMaxAnchors) return GPP_ERR_BAD_ARG;
    *bytes = detect_bytes(B, n_anchors, 1);
    return GPP_OK;
}

extern "C" int gpp_detect_osf_workspace_bytes(int B, int64_t n_anchors, size_t* bytes)
{
    if (!bytes || B < 0 || B > kMaxBatch || n_anchors <= 0 || n_anchors > kMaxAnchors) return GPP_ERR_BAD_ARG;
    *bytes = detect_bytes(B, n_anchors, 4);
    return GPP_OK;
}

extern "C" int gpp_detect_stages_f32(int stages, const float* cls_logits, const float* regression,
                                     const float* regression_dim, const float* anchors, int B, int64_t n_anchors,
                                     int num_base_anchors, int fused_layout, float score_thr, float iou_thr, int max_det,
                                     float* boxes, float* dims, float* scores, int32_t* labels, int32_t* orientations,
                                     int32_t* anchor_index, int32_t* counts,
                                     void* workspace, size_t workspace_bytes, void* stream)
{
    return detect_impl(stages, 1, cls_logits, regression, regression_dim, anchors, B, n_anchors, num_base_anchors, fused_layout, score_thr,
                       iou_thr, max_det, boxes, dims, scores, labels, orientations, anchor_index, counts, workspace, workspace_bytes, stream);
}

extern "C" int gpp_detect_osf_f32(const float* cls_logits, const float* regression, const float* regression_dim,
                                  const float* anchors, int B, int64_t n_anchors, int num_base_anchors, int fused_layout,
                                  float score_thr, float iou_thr, int max_det,
                                  float* boxes, float* dims, float* scores, int32_t* labels, int32_t* orientations,
                                  int32_t* anchor_index, int32_t* counts,
                                  void* workspace, size_t workspace_bytes, void* stream)
{
    return detect_impl(GPP_DETECT_CANDIDATES | GPP_DETECT_SELECT | GPP_DETECT_EMIT, 4, cls_logits, regression, regression_dim, anchors, B,
                       n_anchors, num_base_anchors, fused_layout, score_thr, iou_thr, max_det, boxes, dims, scores, labels, orientations,
                       anchor_index, counts, workspace, workspace_bytes, stream);
}

extern "C" int gpp_detect_f32(const float* cls_logits, const float* regression, const float* regression_dim,
                              const float* anchors, int B, int64_t n_anchors, int num_base_anchors, int fused_layout,
                              float score_thr, float iou_thr, int max_det,
                              float* boxes, float* dims, float* scores, int32_t* labels, int32_t* orientations,
                              int32_t* anchor_index, int32_t* counts,
                              void* workspace, size_t workspace_bytes, void* stream)
{
    return gpp_detect_stages_f32(GPP_DETECT_CANDIDATES | GPP_DETECT_SELECT | GPP_DETECT_EMIT, cls_logits, regression, regression_dim,
                                 anchors, B, n_anchors, num_base_anchors, fused_layout, score_thr, iou_thr, max_det, boxes, dims,
                                 scores, labels, orientations, anchor_index, counts, workspace, workspace_bytes, stream);
}
