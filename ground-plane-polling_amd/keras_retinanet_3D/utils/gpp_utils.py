"""
Ground-plane-polling utilities: the polling entry point (HIP), calibration handling, and the
host-side pose recovery / cuboid assembly that run_network.py performs on the model outputs.

Reference code restated here (citations relative to /root/reference/keras_retinanet_3D):
    fit_road_planes        layers/fit_road_planes.py:49-139  -> HIP kernel, csrc/poll.hip
    load_calibration       bin/run_network.py:48-59
    select_detections      bin/run_network.py:113-135
    recover_pose           bin/run_network.py:137-287 (live branches only: :147-150 make the
                           `else` branch :248-287 unreachable)
    cuboid_corners         bin/run_network.py:298-310
    kitti_lines            bin/run_network.py:295-330
"""

import numpy as np

from ..backend import hip

POLL_THRESHOLD = 0.7      # metres, fit_road_planes.py:94


def _as_device(x, dtype, device):
    import torch
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x)).to(device=device, dtype=dtype).contiguous()


def fit_road_planes(boxes, dimensions, orientations, P_inv, planes, return_index=False, threshold=POLL_THRESHOLD):
    """ Identify 3D keypoints and keyplane for each detection (HIP kernel, gfx950).

    Same arguments, in the same order, as the reference layer (fit_road_planes.py:49-62,157-161):
        boxes        (B, D, 12)  x1 y1 x2 y2 xl yl xm ym xr yr xt yt
        dimensions   (B, D, 3)   h w l
        orientations (B, D)      orientation class, -1 on padding rows
        P_inv        (B, 4, 3)
        planes       (B, N, 4), or (N, 4) for one database shared by the whole batch
    Returns [keypoints (B, D, 4, 3), keyplanes (B, D, 1, 4), residuals (B, D)] (+ the selected
    plane index (B, D) int32 when return_index is set).  NumPy in -> NumPy out; torch tensors
    in -> torch tensors (on the device) out.
    """
    import torch
    device = hip.require_device()
    numpy_out = not isinstance(boxes, torch.Tensor)
    boxes_d = _as_device(boxes, torch.float32, device)
    dims_d = _as_device(dimensions, torch.float32, device)
    orient_d = _as_device(orientations, torch.int32, device)
    pinv_d = _as_device(P_inv, torch.float32, device)
    planes_d = _as_device(planes, torch.float32, device)
    if boxes_d.dim() != 3 or boxes_d.shape[2] != 12:
        raise ValueError('boxes must be (B, D, 12), got {}'.format(tuple(boxes_d.shape)))
    B, D = int(boxes_d.shape[0]), int(boxes_d.shape[1])
    if tuple(dims_d.shape) != (B, D, 3) or tuple(orient_d.shape) != (B, D) or tuple(pinv_d.shape) != (B, 4, 3):
        raise ValueError('inconsistent shapes: dimensions {}, orientations {}, P_inv {}'.format(
            tuple(dims_d.shape), tuple(orient_d.shape), tuple(pinv_d.shape)))
    if planes_d.dim() == 3:
        if planes_d.shape[0] != B or planes_d.shape[2] != 4:
            raise ValueError('planes must be (B, N, 4) or (N, 4), got {}'.format(tuple(planes_d.shape)))
        batched, N = 1, int(planes_d.shape[1])
    elif planes_d.dim() == 2 and planes_d.shape[1] == 4:
        batched, N = 0, int(planes_d.shape[0])
    else:
        raise ValueError('planes must be (B, N, 4) or (N, 4), got {}'.format(tuple(planes_d.shape)))
    if N < 1:
        raise ValueError('the plane database is empty')

    keypoints = torch.empty((B, D, 4, 3), dtype=torch.float32, device=device)
    keyplanes = torch.empty((B, D, 1, 4), dtype=torch.float32, device=device)
    residuals = torch.empty((B, D), dtype=torch.float32, device=device)
    index = torch.empty((B, D), dtype=torch.int32, device=device)
    lib = hip.lib()
    need = hip.c_size_t(0)
    hip.check(lib.gpp_poll_workspace_bytes(B, N, batched, need), 'gpp_poll_workspace_bytes')
    workspace = torch.empty((max(int(need.value), 16),), dtype=torch.uint8, device=device)
    if B * D > 0:
        hip.check(lib.gpp_poll_f32(hip.ptr(boxes_d), hip.ptr(dims_d), hip.ptr(orient_d), hip.ptr(pinv_d),
                                   hip.ptr(planes_d), B, D, N, batched, float(threshold),
                                   hip.ptr(keypoints), hip.ptr(keyplanes), hip.ptr(residuals), hip.ptr(index),
                                   hip.ptr(workspace), workspace.numel(), hip.stream_ptr()), 'gpp_poll_f32')
    out = [keypoints, keyplanes, residuals] + ([index] if return_index else [])
    if numpy_out:
        out = [o.cpu().numpy() for o in out]
    return out


def load_calibration(calib_path, image_scale):
    """ (P, P_inv) from the P2 line of a KITTI calibration file, run_network.py:48-59. """
    with open(calib_path, 'r') as f:
        line = f.readlines()[2]
    P = np.array([float(v) for v in line.split(':', 1)[1].split()]).reshape((3, 4))
    P = np.diag([image_scale, image_scale, 1.0]).dot(P)
    return P, np.linalg.pinv(P)


# ------------------------------------------------------------------------------------------------
# Host-side post-processing of the 8 model outputs (vectorised; the reference loops per detection)
# ------------------------------------------------------------------------------------------------
def select_detections(outputs, scale, image_index=0, score_threshold=0.05, max_detections=100):
    """ run_network.py:113-135 for one image of the batch: undo the image scale on the boxes,
    keep scores > threshold, sort by descending score (stable; the input already is), flatten
    keypoints to (n, 12) and keyplanes to (n, 4).  Returns a dict of NumPy arrays. """
    boxes, dimensions, scores, labels, orientations, keypoints, keyplanes, residuals = [np.asarray(o) for o in outputs[:8]]
    b = boxes[image_index] / scale
    s = scores[image_index]
    idx = np.where(s > score_threshold)[0]
    order = idx[np.argsort(-s[idx], kind='stable')[:max_detections]]
    return {
        'boxes': b[order].astype(np.float32), 'dimensions': dimensions[image_index][order].astype(np.float32),
        'scores': s[order].astype(np.float32), 'labels': labels[image_index][order], 'orientations': orientations[image_index][order],
        'keypoints': keypoints[image_index][order].reshape(-1, 12).astype(np.float32),
        'keyplanes': keyplanes[image_index][order].reshape(-1, 4).astype(np.float32),
        'residuals': residuals[image_index][order].astype(np.float32),
    }


def rotation_vector_from_matrix(R):
    """ cv2.Rodrigues(matrix)[0][:, 0] for a batch (n, 3, 3) -> (n, 3): orthonormalise by SVD
    (R <- U V^T), then axis * angle.  (OpenCV is absent here; restated from its documentation.) """
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    finite = np.isfinite(R).all(axis=(1, 2))
    if not finite.all():                 # a degenerate detection (zero-length edge): NaN pose for that row, the others unaffected
        out = np.full((R.shape[0], 3), np.nan)
        if finite.any():
            out[finite] = rotation_vector_from_matrix(R[finite])
        return out
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
    s = np.sqrt((r * r).sum(axis=1) * 0.25)
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    out = np.zeros_like(r)
    regular = s >= 1e-5
    out[regular] = r[regular] * (theta[regular] / (2.0 * s[regular]))[:, None]
    flip = (~regular) & (c <= 0)                       # rotation by pi: axis from the diagonal
    if flip.any():
        Rf = R[flip]
        t = np.sqrt(np.maximum((np.stack([Rf[:, 0, 0], Rf[:, 1, 1], Rf[:, 2, 2]], axis=1) + 1.0) * 0.5, 0.0))
        t[:, 1] = t[:, 1] * np.where(Rf[:, 0, 1] < 0, -1.0, 1.0)
        t[:, 2] = t[:, 2] * np.where(Rf[:, 0, 2] < 0, -1.0, 1.0)
        sign_fix = (np.abs(t[:, 0]) < np.abs(t[:, 1])) & (np.abs(t[:, 0]) < np.abs(t[:, 2])) & ((Rf[:, 1, 2] > 0) != (t[:, 1] * t[:, 2] > 0))
        t[sign_fix, 2] = -t[sign_fix, 2]
        t *= (theta[flip] / np.maximum(np.linalg.norm(t, axis=1), 1e-300))[:, None]
        out[flip] = t
    return out


def rotation_matrix_from_vector(r):
    """ cv2.Rodrigues(vector)[0] for a batch (n, 3) -> (n, 3, 3) """
    r = np.asarray(r, dtype=np.float64).reshape(-1, 3)
    theta = np.linalg.norm(r, axis=1)
    k = r / np.maximum(theta, 1e-300)[:, None]
    K = np.zeros((r.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -k[:, 2], k[:, 1], k[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 0], -k[:, 1], k[:, 0]
    c, s = np.cos(theta)[:, None, None], np.sin(theta)[:, None, None]
    kk = k[:, :, None] * k[:, None, :]
    R = c * np.eye(3)[None] + (1.0 - c) * kk + s * K
    R[theta < 1e-300] = np.eye(3)
    return R


def recover_pose(det):
    """ 6-DoF pose from the 3D keypoints, run_network.py:137-247 (live branches; the reference
    overwrites h and l with keypoint distances and keeps the network's w).
    det: dict from select_detections.  Adds 'locations' (n, 3), 'angles' (n, 3) and returns the
    updated 'dimensions' (n, 3) = (h, w, l), all float32 like np.empty_like(dimensions) there. """
    kp = det['keypoints'].astype(np.float32).reshape(-1, 4, 3)
    o = np.asarray(det['orientations'])
    dims = det['dimensions'].astype(np.float32).copy()
    n = kp.shape[0]
    X_l, X_m, X_r, X_t = kp[:, 0], kp[:, 1], kp[:, 2], kp[:, 3]
    use_l = (o == 0) | (o == 3)                      # 'outlier == 2' branch: X_l, X_m, X_t
    X_s = np.where(use_l[:, None], X_l, X_r)         # the second bottom keypoint that is used
    h = np.linalg.norm(X_t - X_m, axis=1).astype(np.float32)
    e = np.linalg.norm(X_s - X_m, axis=1).astype(np.float32)
    dims[:, 0] = h
    # which dimension the bottom edge measures: length for o in {0,3} with X_l and {1,2} with X_r
    dims[:, 2] = e
    y_dir = (X_m - X_t) / h[:, None]
    # x (length) axis direction and its sign, run_network.py:167-247
    sign_x = np.select([o == 0, o == 1, o == 2, o == 3], [1.0, 1.0, -1.0, -1.0], default=1.0).astype(np.float32)
    x_dir = sign_x[:, None] * (X_m - X_s) / e[:, None]
    z_dir = np.cross(x_dir, y_dir)
    sign_z = np.select([o == 0, o == 1, o == 2, o == 3], [1.0, -1.0, 1.0, -1.0], default=1.0).astype(np.float32)
    locations = (X_m + X_s) / 2 + sign_z[:, None] * z_dir * dims[:, 1:2] / 2
    Rm = np.stack([x_dir, y_dir, z_dir], axis=-1)
    angles = rotation_vector_from_matrix(Rm).astype(np.float32) if n else np.zeros((0, 3), np.float32)
    det = dict(det)
    det['dimensions'] = dims
    det['locations'] = locations.astype(np.float32)
    det['angles'] = angles
    return det


def cuboid_corners(det):
    """ 8 corners (n, 3, 8) in camera coordinates, run_network.py:298-310 (corner order of
    label_prep/computeBox3D.m:22-24) """
    h, w, l = (det['dimensions'][:, k].astype(np.float64) for k in range(3))
    zero = np.zeros_like(h)
    x = np.stack([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2], axis=1)
    y = np.stack([zero, zero, zero, zero, -h, -h, -h, -h], axis=1)
    z = np.stack([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2], axis=1)
    R = rotation_matrix_from_vector(det['angles'])
    return R @ np.stack([x, y, z], axis=1) + det['locations'].astype(np.float64)[:, :, None]


def _wrap(a):
    a = a % (2 * np.pi)
    return np.where(a >= np.pi, a - 2 * np.pi, a)


def kitti_lines(det, image_shape):
    """ KITTI result lines, run_network.py:295-330 """
    X = cuboid_corners(det)
    r_y = _wrap(det['angles'][:, 1].astype(np.float64))
    Y = X[:, 1, :].max(axis=1)
    h = Y - X[:, 1, :].min(axis=1)
    loc = det['locations'].astype(np.float64)
    alpha = _wrap(r_y + np.arctan2(loc[:, 2], loc[:, 0]) + 1.5 * np.pi)
    b = det['boxes']
    lines = []
    for i in range(len(det['scores'])):
        lines.append("Car -1 -1 %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n" % (
            alpha[i], max(b[i, 0], 0.0), max(b[i, 1], 0.0), min(b[i, 2], image_shape[1]), min(b[i, 3], image_shape[0]),
            h[i], det['dimensions'][i, 1], det['dimensions'][i, 2], loc[i, 0], Y[i], loc[i, 2], r_y[i], det['scores'][i]))
    return lines
