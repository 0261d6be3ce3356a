"""
Detection decode + filtering on the device (HIP kernels csrc/decode.hip, C ABI gpp_detect_f32).

Takes the place of three reference layers at once, so that the (B, 137256, .) head tensors are
read exactly once and only the 100 padded detections per image are written:
    RegressBoxes       /root/reference/keras_retinanet_3D/layers/_misc.py:103-153
    RegressDims        /root/reference/keras_retinanet_3D/layers/_misc.py:156-199
    FilterDetections   /root/reference/keras_retinanet_3D/layers/filter_detections.py:192-304
(plus the sigmoid of models/retinanet.py:72-73).  Implemented: nms=True|False, class_specific_filter=True|False (identical for one class),
orientation_specific_filter=False|True, one object class (the reference's only trained case, preprocessing/kitti.py:28-35).
"""

import numpy as np

from ..backend import hip
from ..utils.anchors import NUM_BASE_ANCHORS

SCORE_THRESHOLD = 0.05     # filter_detections.py:26
MAX_DETECTIONS = 100       # filter_detections.py:27
NMS_THRESHOLD = 0.5        # filter_detections.py:28


class FilterDetections(object):
    """ Callable with preallocated outputs/workspace for a fixed (batch, n_anchors). """

    def __init__(self, batch, n_anchors, device, nms=True, class_specific_filter=True, orientation_specific_filter=False,
                 nms_threshold=NMS_THRESHOLD, score_threshold=SCORE_THRESHOLD, max_detections=MAX_DETECTIONS,
                 fused_regression=False):
        import torch
        self.osf = bool(orientation_specific_filter)     # per-orientation threshold + NMS (filter_detections.py:84-98)
        if self.osf and int(batch) > 16:
            raise ValueError('orientation_specific_filter=True handles at most 16 images per call')
        if not nms:
            nms_threshold = 2.0      # IoU never exceeds 1: nothing is suppressed, the kernel reduces to threshold + top-k
        self.batch, self.n_anchors, self.device = int(batch), int(n_anchors), device
        self.nms_threshold, self.score_threshold, self.max_detections = nms_threshold, score_threshold, max_detections
        self.fused = int(bool(fused_regression))
        B, D = self.batch, int(max_detections)
        f32, i32 = torch.float32, torch.int32
        self.boxes = torch.empty((B, D, 12), dtype=f32, device=device)
        self.dimensions = torch.empty((B, D, 3), dtype=f32, device=device)
        self.scores = torch.empty((B, D), dtype=f32, device=device)
        self.labels = torch.empty((B, D), dtype=i32, device=device)
        self.orientations = torch.empty((B, D), dtype=i32, device=device)
        self.anchor_index = torch.empty((B, D), dtype=i32, device=device)
        self.counts = torch.zeros((max(B, 1),), dtype=i32, device=device)
        need = hip.c_size_t(0)
        size_fn = hip.lib().gpp_detect_osf_workspace_bytes if self.osf else hip.lib().gpp_detect_workspace_bytes
        hip.check(size_fn(B, self.n_anchors, need), 'gpp_detect_workspace_bytes')
        self.workspace = torch.empty((int(need.value),), dtype=torch.uint8, device=device)

    def args(self, cls_logits, regression, regression_dim, anchors):
        return (hip.ptr(cls_logits), hip.ptr(regression), hip.ptr(regression_dim), hip.ptr(anchors),
                self.batch, self.n_anchors, NUM_BASE_ANCHORS, self.fused,
                float(self.score_threshold), float(self.nms_threshold), int(self.max_detections),
                hip.ptr(self.boxes), hip.ptr(self.dimensions), hip.ptr(self.scores), hip.ptr(self.labels),
                hip.ptr(self.orientations), hip.ptr(self.anchor_index), hip.ptr(self.counts),
                hip.ptr(self.workspace), self.workspace.numel())

    def __call__(self, cls_logits, regression, regression_dim, anchors):
        """ cls_logits (B, A, 8), regression (B, A, 12) [or fused (B, A/12, 144)], regression_dim (B, A, 3),
        anchors (A, 4): float32 device tensors.  Returns [boxes, dimensions, scores, labels, orientations]
        in the reference's output order (filter_detections.py:189). """
        if self.batch > 0:
            fn = hip.lib().gpp_detect_osf_f32 if self.osf else hip.lib().gpp_detect_f32
            hip.check(fn(*(self.args(cls_logits, regression, regression_dim, anchors) + (hip.stream_ptr(),))), 'gpp_detect_f32')
        return [self.boxes, self.dimensions, self.scores, self.labels, self.orientations]


def filter_detections(cls_logits, regression, regression_dim, anchors, fused_regression=False, **kwargs):
    """ One-shot convenience wrapper: NumPy or torch in, same kind out. """
    import torch
    device = hip.require_device()
    numpy_out = not isinstance(cls_logits, torch.Tensor)

    def dev(x):
        return (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(x))).to(device=device, dtype=torch.float32).contiguous()

    cls_logits, regression, regression_dim, anchors = dev(cls_logits), dev(regression), dev(regression_dim), dev(anchors)
    op = FilterDetections(cls_logits.shape[0], anchors.shape[0], device, fused_regression=fused_regression, **kwargs)
    out = op(cls_logits, regression, regression_dim, anchors) + [op.anchor_index, op.counts[:op.batch]]
    return [o.cpu().numpy() for o in out] if numpy_out else out
