// Ground-plane polling on gfx950 (MI355X): one workgroup per detection, planes strided over
// the 256 lanes of its four wavefronts, single pass over the plane database, exact
// reference selection semantics.
//
// Replaces the TF graph built by /root/reference/keras_retinanet_3D/layers/fit_road_planes.py
//   fit_road_planes :49-139  (poll :18-32, calc_X_t :34-47)
// which materialises ~40 (B, D, N, ...) temporaries; here nothing per-pair is stored.
//
// Arithmetic: float32, every operation separate (this file is compiled with
// -ffp-contract=off, IEEE division and square root), evaluation order identical to
// oracle/polling_np.py / oracle/polling.c, so results agree with the oracle bit for bit.
//
// Selection (fit_road_planes.py:112-119) in one pass:
//   R'_j = 100 if votes_j < max_j votes, then 100 if zc_j < 0, else R_j;  j* = first argmin R'.
// Each lane keeps, for the highest vote level L it has met so far, the lexicographic minimum
// (R, j) over planes at level L with zc >= 0, and i100 = the lowest index of any plane that
// would carry the sentinel if L were the global maximum (lower level, or level L with zc < 0).
// When a lane meets a higher level everything it saw before becomes sentinel; the lowest
// index it has seen is its first plane (= its global lane id).  The merge across lanes
// (DPP/shuffle inside a wavefront, LDS across the four wavefronts) first agrees on the
// global maximum level, demotes lanes below it, then resolves
//   min( (rmin, imin), (100, i100) )  lexicographically  -> j*.
// NaN residuals never win (tf.argmin / Eigen compare with '<' from FLT_MAX); if nothing
// finite and nothing masked exists the answer is index 0, as in the reference.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <stdlib.h>

#include "gpp.h"

// This file is compiled WITHOUT packed-FP32 instructions (Makefile: NOPK).  Round 3 found that on this platform a wavefront that is
// context-saved and resumed (compute wave save / restore: the driver rebuilds the runlist whenever ANY process on the GPU creates or
// destroys a queue) can lose the lanes 48-63 of a v_pk_{mul,add,fma}_f32 result: with the packed build 87 of 32 000 plan runs under
// queue churn returned a wrong plane (the one-lane-state dump showed a uniform poll target reading 0 in exactly those 16 lanes of one
// wavefront), with the unpacked build 0 of 32 000 on the same box (tools/poll_race_stress.py, DESIGN.md section 4.4, profiles/r3/).
//
// Diagnostic build only (make polldbg -> libgpp_hip_polldbg.so, tools/poll_race_stress.py): the kernel additionally stores every
// lane's state after its scan (8 words per lane).  The production library contains none of it.
#ifdef GPP_POLL_DEBUG
static float* g_poll_dbg = nullptr;
extern "C" int gpp_poll_debug_buffer(void* buf) { g_poll_dbg = (float*)buf; return GPP_OK; }
#define GPP_POLL_DBG_PARAM , float* __restrict__ dbg
#define GPP_POLL_DBG_ARG , g_poll_dbg
#else
#define GPP_POLL_DBG_PARAM
#define GPP_POLL_DBG_ARG
#endif

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = 64 * kWaves;

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 scale3(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float norm3(V3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.0f) - (v < 0.0f)); }

struct Hyp { V3 X[4]; float zc, votes, res; };

// fit_road_planes.py:84-109 for one (detection, plane) pair
__device__ __forceinline__ Hyp evaluate(const V3 (&ray)[4], float4 pl, const float (&target)[6], float thr)
{
    Hyp h;
    V3 n = {pl.x, pl.y, pl.z};
    float nd = -pl.w;
#pragma unroll
    for (int k = 0; k < 3; ++k) h.X[k] = scale3(ray[k], fabsf(nd / dot3(n, ray[k])));
    h.zc = cross3(sub3(h.X[0], h.X[1]), sub3(h.X[2], h.X[1])).y;
    V3 perp = cross3(ray[3], cross3(n, ray[3]));
    float num = dot3(perp, h.X[1]);
    float den = dot3(perp, n);
    h.X[3] = sub3(h.X[1], scale3(n, num / den));
    constexpr int seg[6][2] = {{1, 3}, {0, 1}, {1, 2}, {0, 2}, {0, 3}, {2, 3}};
    h.votes = 0.0f;
    h.res = 0.0f;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        float r = fabsf(norm3(sub3(h.X[seg[p][0]], h.X[seg[p][1]])) - target[p]);
        float v = (r > thr) ? 0.0f : 1.0f;
        h.votes = (p == 0) ? v : h.votes + v;
        h.res = (p == 0) ? r : h.res + r;
    }
    return h;
}

// fit_road_planes.py:75-77
__global__ void canonical_planes_kernel(const float4* __restrict__ planes, float4* __restrict__ canon, int64_t total)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    float4 p = planes[j];
    float dir = -sgn(p.y);
    float a = p.x * dir, b = p.y * dir, c = p.z * dir, d = p.w * dir;
    float nn = sqrtf((a * a + b * b) + c * c);
    canon[j] = make_float4(a / nn, b / nn, c / nn, d / nn);
}

// UNROLL planes of a lane are evaluated per loop iteration (independent instruction streams: the exact divide / square-root
// sequences are long dependent chains, and a detection's workgroup brings only four wavefronts to its CU); the selection
// state is updated in ascending plane order afterwards, so the result is the same for every UNROLL.
template <int UNROLL>
__global__ __launch_bounds__(kThreads) void poll_kernel(
    const float* __restrict__ boxes, const float* __restrict__ dims, const int32_t* __restrict__ orient,
    const float* __restrict__ P_inv, const float4* __restrict__ canon, int D, int N, int planes_batched, float thr,
    float* __restrict__ keypoints, float* __restrict__ keyplanes, float* __restrict__ residuals,
    int32_t* __restrict__ best_idx GPP_POLL_DBG_PARAM)
{
    const int det = blockIdx.x;           // one workgroup per (image, detection)
    const int b = det / D;
    const int tid = threadIdx.x;
    const float* bx = boxes + (size_t)det * 12;
    const float* dm = dims + (size_t)det * 3;
    const float* Pi = P_inv + (size_t)b * 12;
    const int o = orient[det];

    // Padding rows.  FilterDetections pads its outputs with -1 (filter_detections.py:170-177) and the reference polls those rows like any
    // other (fit_road_planes.py has no mask): boxes = dims = orientation = -1 give the same finite garbage for EVERY padding row of an
    // image (same P_inv, same planes).  A run of consecutive padding rows is therefore polled ONCE -- by the workgroup of its first row,
    // which writes its result to every row of the run -- and the workgroups of the other rows leave at once: the same bytes as polling
    // each row, at the cost of one.  A frame with 7 detections pays for 8 scans of the database instead of 100.
    // (Only the exact padding pattern qualifies; a row with orientation -1 and anything else in its boxes is polled as the reference would.)
    auto is_padding = [&](int r) {
        if (orient[r] != -1) return false;
        bool pad = true;
        for (int k = 0; k < 12; ++k) pad = pad && boxes[(size_t)r * 12 + k] == -1.0f;
        for (int k = 0; k < 3; ++k) pad = pad && dims[(size_t)r * 3 + k] == -1.0f;
        return pad;
    };
    int run = 1;                          // rows this workgroup writes: its own, plus the padding rows that follow a first padding row
    __shared__ int s_run;
    if (o == -1 && is_padding(det)) {     // (uniform over the workgroup: every lane reads the same addresses)
        if (det % D != 0 && is_padding(det - 1)) return;                 // not the first of its run: that row's workgroup writes this one
        // length of the run: the lanes look at the following rows of the image in parallel (a serial walk over 93 rows of 15 dependent
        // loads each cost more than the scan it saves)
        const int left = D - 1 - det % D;                                // rows of this image behind this one
        if (tid == 0) s_run = left + 1;
        __syncthreads();
        for (int r = 1 + tid; r <= left; r += kThreads)
            if (!is_padding(det + r)) atomicMin(&s_run, r);
        __syncthreads();
        run = s_run;
    }

    // :80-83 back-projection (uniform over the workgroup; every lane keeps its own copy)
    V3 ray[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = bx[4 + 2 * k], y = bx[5 + 2 * k];
        float r0 = (Pi[0] * x + Pi[1] * y) + Pi[2] * 1.0f;
        float r1 = (Pi[3] * x + Pi[4] * y) + Pi[5] * 1.0f;
        float r2 = (Pi[6] * x + Pi[7] * y) + Pi[8] * 1.0f;
        float s = sgn(r2);
        ray[k] = {r0 * s, r1 * s, r2 * s};
    }
    // :61-73, 95-109 poll targets (one_hot(-1) = 0 -> orientation dependent targets are 0)
    float h = dm[0], w = dm[1], l = dm[2];
    float hw = sqrtf(h * h + w * w), wl = sqrtf(w * w + l * l), hl = sqrtf(h * h + l * l);
    float oh0 = (o == 0) ? 1.0f : 0.0f, oh1 = (o == 1) ? 1.0f : 0.0f;
    float oh2 = (o == 2) ? 1.0f : 0.0f, oh3 = (o == 3) ? 1.0f : 0.0f;
#define GPP_MIX(a, b, c, d) (((oh0 * (a) + oh1 * (b)) + oh2 * (c)) + oh3 * (d))
    const float target[6] = {h, GPP_MIX(l, w, w, l), GPP_MIX(w, l, l, w), wl, GPP_MIX(hl, hw, hw, hl), GPP_MIX(hw, hl, hl, hw)};
#undef GPP_MIX

    const float4* pl = canon + (planes_batched ? (size_t)b * N : 0);

    // ---- single pass over this lane's planes
    int level = -1;            // highest vote count met so far (0..6)
    float rmin = INFINITY;     // best residual among level-`level` planes with zc >= 0
    int imin = INT_MAX;
    int i100 = INT_MAX;        // lowest index that carries the sentinel if `level` is the maximum
    auto update = [&](const Hyp& hy, int j) {
        int v = (int)hy.votes;
        if (v > level) {
            if (j != tid) i100 = tid;      // all earlier planes of this lane drop to the sentinel
            rmin = INFINITY; imin = INT_MAX; level = v;
        }
        if (v < level || hy.zc < 0.0f) {
            i100 = min(i100, j);
        } else {
            float key = (hy.res < FLT_MAX) ? hy.res : INFINITY;   // NaN / inf never win
            if (key < rmin) { rmin = key; imin = j; }
        }
    };
#ifdef GPP_POLL_DEBUG
    int dbg_iters = 0;
#endif
    for (int j = tid; j < N; j += UNROLL * kThreads) {
#ifdef GPP_POLL_DEBUG
        ++dbg_iters;
#endif
        Hyp hy[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int ju = j + u * kThreads;
            hy[u] = evaluate(ray, pl[ju < N ? ju : j], target, thr);     // past the end: plane j again, not used
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (j + u * kThreads < N) update(hy[u], j + u * kThreads);
    }

#ifdef GPP_POLL_DEBUG
    if (dbg) {      // every lane's state after its scan (8 words per lane): rmin, imin, i100, level, active-lane mask (lo, hi), iterations, tid
        float* q = dbg + ((size_t)det * kThreads + tid) * 8;
        const unsigned long long act = __ballot(1);
        q[0] = rmin; q[1] = __int_as_float(imin); q[2] = __int_as_float(i100); q[3] = __int_as_float(level);
        q[4] = __int_as_float((int)(act & 0xffffffffu)); q[5] = __int_as_float((int)(act >> 32)); q[6] = __int_as_float(dbg_iters);
        q[7] = __int_as_float(tid);
    }
#endif
    // ---- merge: agree on the maximum level, demote lanes below it
    int vmax = level;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) vmax = max(vmax, __shfl_xor(vmax, s, 64));
    __shared__ int s_level[kWaves];
    __shared__ float s_rmin[kWaves];
    __shared__ int s_imin[kWaves];
    __shared__ int s_i100[kWaves];
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) s_level[wave] = vmax;
    __syncthreads();
    vmax = max(max(s_level[0], s_level[1]), max(s_level[2], s_level[3]));
    if (level < vmax) {
        rmin = INFINITY; imin = INT_MAX;
        i100 = (tid < N) ? tid : INT_MAX;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        float orr = __shfl_xor(rmin, s, 64);
        int oi = __shfl_xor(imin, s, 64);
        int o100 = __shfl_xor(i100, s, 64);
        if (orr < rmin || (orr == rmin && oi < imin)) { rmin = orr; imin = oi; }
        i100 = min(i100, o100);
    }
    if (lane == 0) { s_rmin[wave] = rmin; s_imin[wave] = imin; s_i100[wave] = i100; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kWaves; ++q) {
        float orr = s_rmin[q];
        int oi = s_imin[q];
        if (orr < rmin || (orr == rmin && oi < imin)) { rmin = orr; imin = oi; }
        i100 = min(i100, s_i100[q]);
    }

    int best;
    if (rmin < 100.0f) best = imin;
    else if (rmin == 100.0f) best = min(imin, i100);
    else if (i100 != INT_MAX) best = i100;
    else best = (rmin < INFINITY) ? imin : 0;

    // ---- outputs (:122-137): re-evaluate the winner, one lane writes
    if (tid == 0) {
        float4 p = pl[best];
        Hyp hy = evaluate(ray, p, target, thr);
        float r = hy.res;
        if ((int)hy.votes < vmax) r = 100.0f;
        if (hy.zc < 0.0f) r = 100.0f;
        for (int row = det; row < det + run; ++row) {
            float* kp = keypoints + (size_t)row * 12;
#pragma unroll
            for (int k = 0; k < 4; ++k) { kp[3 * k] = hy.X[k].x; kp[3 * k + 1] = hy.X[k].y; kp[3 * k + 2] = hy.X[k].z; }
            float* kq = keyplanes + (size_t)row * 4;
            kq[0] = p.x; kq[1] = p.y; kq[2] = p.z; kq[3] = p.w;
            residuals[row] = r / 6.0f;
            if (best_idx) best_idx[row] = best;
        }
    }
}

}  // namespace

extern "C" int gpp_poll_workspace_bytes(int B, int N, int planes_batched, size_t* bytes)
{
    if (!bytes || B < 0 || N < 0) return GPP_ERR_BAD_ARG;
    *bytes = sizeof(float) * 4 * (size_t)N * (planes_batched ? (size_t)(B > 0 ? B : 1) : 1);
    return GPP_OK;
}

extern "C" int gpp_poll_f32(const float* boxes, const float* dims, const int32_t* orient, const float* P_inv,
                            const float* planes, int B, int D, int N, int planes_batched, float thr,
                            float* keypoints, float* keyplanes, float* residuals, int32_t* best_idx,
                            void* workspace, size_t workspace_bytes, void* stream)
{
    if (B < 0 || D < 0 || N <= 0) return GPP_ERR_BAD_ARG;
    if (B == 0 || D == 0) return GPP_OK;
    if (!boxes || !dims || !orient || !P_inv || !planes || !keypoints || !keyplanes || !residuals || !workspace)
        return GPP_ERR_BAD_ARG;
    size_t need = 0;
    gpp_poll_workspace_bytes(B, N, planes_batched, &need);
    if (workspace_bytes < need) return GPP_ERR_WORKSPACE;
    if (((uintptr_t)planes & 15) || ((uintptr_t)workspace & 15)) return GPP_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    int64_t total = (int64_t)N * (planes_batched ? B : 1);
    canonical_planes_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(
        (const float4*)planes, (float4*)workspace, total);
    static const int unroll = [] { const char* e = getenv("GPP_POLL_UNROLL"); return e ? atoi(e) : 0; }();      // 0 = by N (tools/bench_poll.py)
    // measured (MI355X, 8 x 100 detections): 10k planes 94.5 / 91.0 / 90.4 us with 1 / 2 / 4 planes per iteration, 22k planes
    // (4 x 100) 113 / 109 / 104 us, 1k planes no difference: the kernel is bound by VALU issue (about 300 instructions per pair
    // for the exact IEEE divides and square roots), not by latency
    const int u = unroll ? unroll : (N >= 16 * kThreads ? 4 : (N >= 4 * kThreads ? 2 : 1));
    if (u >= 4)
        poll_kernel<4><<<dim3((unsigned)(B * D)), dim3(kThreads), 0, st>>>(boxes, dims, orient, P_inv, (const float4*)workspace, D, N,
                                                                             planes_batched, thr, keypoints, keyplanes, residuals, best_idx GPP_POLL_DBG_ARG);
    else if (u >= 2)
        poll_kernel<2><<<dim3((unsigned)(B * D)), dim3(kThreads), 0, st>>>(boxes, dims, orient, P_inv, (const float4*)workspace, D, N,
                                                                             planes_batched, thr, keypoints, keyplanes, residuals, best_idx GPP_POLL_DBG_ARG);
    else
        poll_kernel<1><<<dim3((unsigned)(B * D)), dim3(kThreads), 0, st>>>(boxes, dims, orient, P_inv, (const float4*)workspace, D, N,
                                                                             planes_batched, thr, keypoints, keyplanes, residuals, best_idx GPP_POLL_DBG_ARG);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPP_OK : (int)e;
}
