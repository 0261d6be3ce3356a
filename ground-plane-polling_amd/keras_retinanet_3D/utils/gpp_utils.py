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
