/*
 * gpp.h -- C ABI of libgpp_hip.so: the MI355X (gfx950) implementation of the data-parallel
 * hot path of arangesh/Ground-Plane-Polling, i.e. everything that one
 *     model.predict_on_batch([images, P_inv, planes])
 * (reference keras_retinanet_3D/bin/run_network.py:110) executes on the device:
 * RetinaNet-3D forward (ResNet + FPN + three heads), anchor decode, NMS / top-k, and the
 * per-detection ground-plane polling.
 *
 * The reference has no native code and no FFI; its "operator API" for this path is the
 * alias table keras_retinanet_3D/backend/tensorflow_backend.py:20-156 plus the Keras layers
 * in keras_retinanet_3D/layers/.  Each entry point below names the reference code it
 * replaces.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int: 0 = GPP_OK, < 0 = argument error (below), > 0 = hipError_t
 *   - never throws, never aborts, allocates nothing: the caller owns every buffer, including
 *     workspaces whose sizes are reported by the *_workspace_bytes functions
 *   - all pointers are DEVICE pointers unless named host_*; kernels are enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream) and run asynchronously
 *   - no global state; safe to call from several host threads on distinct streams
 *   - tensors are dense, row-major, NHWC for images / feature maps
 */
#ifndef GPP_H_
#define GPP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPP_OK 0
#define GPP_ERR_BAD_ARG (-1)    /* null pointer, negative size, unsupported shape */
#define GPP_ERR_WORKSPACE (-2)  /* workspace too small */
#define GPP_ERR_ALIGN (-3)      /* pointer not aligned as documented */
#define GPP_ERR_UNSUPPORTED (-4)

/* Library / build identification: "gpp-hip <version> gfx950". Host pointer, static storage. */
const char* gpp_version(void);

/* ------------------------------------------------------------------------------------------
 * Ground-plane polling.
 * Replaces layers/fit_road_planes.py:49-139 `fit_road_planes` (+ `poll` :18-32, `calc_X_t`
 * :34-47) and the `FitRoadPlanes` layer :142-186.
 *
 *   boxes      (B, D, 12) f32   x1 y1 x2 y2 xl yl xm ym xr yr xt yt   (-1 rows = padding)
 *   dims       (B, D, 3)  f32   h w l
 *   orient     (B, D)     i32   orientation class 0..3, -1 = padding
 *   P_inv      (B, 4, 3)  f32   pseudo-inverse of the scaled camera matrix
 *   planes     (N, 4) f32 if planes_batched == 0 (one database shared by the batch), else
 *              (B, N, 4) as the reference feeds it (preprocessing/kitti.py:220); 16-byte aligned
 *   thr        poll threshold in metres (reference constant 0.7, fit_road_planes.py:94)
 *   keypoints  (B, D, 4, 3) f32  X_l X_m X_r X_t on the selected plane
 *   keyplanes  (B, D, 1, 4) f32  the selected plane, canonicalised (normal up, unit norm)
 *   residuals  (B, D)     f32   masked residual of the selected plane / 6
 *   best_idx   (B, D)     i32   index of the selected plane (may be NULL; the reference
 *                               computes it at :119 but never returns it)
 *   workspace  gpp_poll_workspace_bytes() bytes, 16-byte aligned (canonical planes)
 * ---------------------------------------------------------------------------------------- */
int gpp_poll_workspace_bytes(int B, int N, int planes_batched, size_t* bytes);

int gpp_poll_f32(const float* boxes, const float* dims, const int32_t* orient, const float* P_inv,
                 const float* planes, int B, int D, int N, int planes_batched, float thr,
                 float* keypoints, float* keyplanes, float* residuals, int32_t* best_idx,
                 void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif

#endif /* GPP_H_ */
