"""
ORACLE (test infrastructure, not product code): NumPy restatement of the reference's
ground-plane polling, float32 arithmetic evaluated operation by operation.

Follows /root/reference/keras_retinanet_3D/layers/fit_road_planes.py:
    poll            :18-32
    calc_X_t        :34-47
    fit_road_planes :49-139

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (keras_retinanet_3D.utils.gpp_utils / layers.FitRoadPlanes)
runs the HIP kernel and never falls back to it.

Pinned by tests/golden/polling_*.npz, which hold inputs and outputs of the reference's
own fit_road_planes.py executed unmodified on a NumPy stand-in for keras.backend /
tensorflow (oracle/gen_polling_goldens.py).

Deliberate properties
* every product, sum, division and square root is a separate float32 operation (no
  BLAS, no fused multiply-add) so that the C restatement (oracle/polling.c) and the HIP
  kernel (built with -ffp-contract=off) can match it bit for bit;
* 3-term dot products are evaluated ((a0*b0 + a1*b1) + a2*b2), the order a row-times-
  column matmul visits them (fit_road_planes.py:43-44,86);
* argmin returns the first minimum (tf.argmin); NaN residuals are never selected.
"""

import numpy as np

F = np.float32
POLL_THRESHOLD = F(0.7)        # fit_road_planes.py:94
MASK_SENTINEL = F(100.0)       # fit_road_planes.py:117-118


def _dot3(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cross(a, b):
    return np.stack([
        a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0],
    ], axis=-1)


def _norm3(a):
    return np.sqrt((a[..., 0] * a[..., 0] + a[..., 1] * a[..., 1]) + a[..., 2] * a[..., 2])


def _norm2(a, b):
    return np.sqrt(a * a + b * b)


def canonical_planes(planes):
    """ fit_road_planes.py:75-77.  planes (..., N, 4) -> canonical float32 planes. """
    planes = np.asarray(planes, dtype=F)
    direction = -np.sign(planes[..., 1:2])
    planes = planes * direction
    return planes / _norm3(planes[..., 0:3])[..., None]


def back_project(boxes, P_inv):
    """ fit_road_planes.py:80-83.  boxes (B, D, 12), P_inv (B, 4, 3) -> rays (B, D, 4, 3)
    in keypoint order l, m, r, t.  The homogeneous 4th component is discarded. """
    boxes = np.asarray(boxes, dtype=F)
    P_inv = np.asarray(P_inv, dtype=F)
    kp = boxes[:, :, 4:].reshape(boxes.shape[0], boxes.shape[1], 4, 2)
    x = kp[..., 0]                                    # (B, D, 4)
    y = kp[..., 1]
    P = P_inv[:, None, None, :, :]                    # (B, 1, 1, 4, 3)
    # row r of P_inv times (x, y, 1): ((p0*x + p1*y) + p2*1)
    d = (P[..., 0:3, 0] * x[..., None] + P[..., 0:3, 1] * y[..., None]) + P[..., 0:3, 2] * F(1.0)
    return d * np.sign(d[..., 2:3])                   # (B, D, 4, 3)


def poll(P0, P1, target):
    """ fit_road_planes.py:18-32. """
    dist = _norm3(P0 - P1)
    res = np.abs(dist - target)
    votes = np.where(res > POLL_THRESHOLD, F(0.0), F(1.0)).astype(F)
    return votes, res


def hypotheses(rays, planes_c):
    """ fit_road_planes.py:84-91.  rays (B, D, 4, 3), planes_c (B, N, 4) ->
    X (B, D, N, 4, 3) keypoints l, m, r, t on every plane and zc (B, D, N). """
    n = planes_c[:, None, :, None, 0:3]               # (B, 1, N, 1, 3)
    dd = planes_c[:, None, :, None, 3]                # (B, 1, N, 1)
    r = rays[:, :, None, 0:3, :]                      # (B, D, 1, 3, 3)
    den = _dot3(n, r)                                 # (B, D, N, 3)
    scale = np.abs((-dd) / den)
    X = r * scale[..., None]                          # (B, D, N, 3, 3)
    X_l, X_m, X_r = X[..., 0, :], X[..., 1, :], X[..., 2, :]
    zc = _cross(X_l - X_m, X_r - X_m)[..., 1]
    # calc_X_t(d_1 = normal, d_2 = t ray, X_m), fit_road_planes.py:34-47
    d1 = np.broadcast_to(planes_c[:, None, :, 0:3], X_m.shape)
    d2 = np.broadcast_to(rays[:, :, None, 3, :], X_m.shape)
    perp = _cross(d2, _cross(d1, d2))
    num = _dot3(perp, X_m)
    dend = _dot3(perp, d1)
    X_t = X_m - (num / dend)[..., None] * d1
    return np.concatenate([X, X_t[..., None, :]], axis=-2), zc


def poll_targets(dimensions, orientations):
    """ The six target lengths per detection, fit_road_planes.py:65-73,95-109.
    dimensions (B, D, 3) = (h, w, l); orientations (B, D) int, -1 on padding rows
    (one_hot(-1) is all zero, so the orientation dependent targets become 0). """
    dimensions = np.asarray(dimensions, dtype=F)
    h, w, l = dimensions[..., 0], dimensions[..., 1], dimensions[..., 2]
    hw, wl, hl = _norm2(h, w), _norm2(w, l), _norm2(h, l)
    oh = np.stack([(np.asarray(orientations) == k) for k in range(4)], axis=-1).astype(F)

    def mix(a, b, c, d):
        # keras.backend.sum(one_hot * concat, axis=2): sum over 4 terms in order
        return ((oh[..., 0] * a + oh[..., 1] * b) + oh[..., 2] * c) + oh[..., 3] * d

    return [h, mix(l, w, w, l), mix(w, l, l, w), wl, mix(hl, hw, hw, hl), mix(hw, hl, hl, hw)]


POLL_SEGMENTS = [(1, 3), (0, 1), (1, 2), (0, 2), (0, 3), (2, 3)]   # fit_road_planes.py:95-109


def fit_road_planes(boxes, dimensions, orientations, P_inv, planes, return_index=False):
    """ Same arguments and outputs as the reference fit_road_planes (fit_road_planes.py:49-139):
    keypoints (B, D, 4, 3), keyplanes (B, D, 1, 4), residuals (B, D), all float32.
    With return_index=True also the selected plane index (B, D) int64, which the
    reference computes (:119) but never returns. """
    planes_c = canonical_planes(planes)
    rays = back_project(boxes, P_inv)
    X, zc = hypotheses(rays, planes_c)
    targets = poll_targets(dimensions, orientations)
    votes = None
    residuals = None
    for (a, b), t in zip(POLL_SEGMENTS, targets):
        v, r = poll(X[..., a, :], X[..., b, :], t[..., None])
        votes = v if votes is None else votes + v
        residuals = r if residuals is None else residuals + r
    vmax = votes.max(axis=2, keepdims=True)
    residuals = np.where(votes - vmax < 0, MASK_SENTINEL, residuals)
    residuals = np.where(zc < 0, MASK_SENTINEL, residuals).astype(F)
    # tf.argmin: first minimum, NaN never wins (Eigen compares with '<' starting from
    # (index 0, highest finite value)).
    key = np.where(residuals < np.finfo(F).max, residuals, np.inf)
    best = np.argmin(key, axis=2)
    best = np.where(np.isinf(np.take_along_axis(key, best[..., None], axis=2)[..., 0]), 0, best)
    B, D = best.shape
    bi = np.arange(B)[:, None]
    di = np.arange(D)[None, :]
    keyplanes = planes_c[bi, best][:, :, None, :]
    keypoints = X[bi, di, best]
    res = residuals[bi, di, best] / F(6.0)
    out = [keypoints.astype(F), keyplanes.astype(F), res.astype(F)]
    if return_index:
        out.append(best.astype(np.int64))
    return out
