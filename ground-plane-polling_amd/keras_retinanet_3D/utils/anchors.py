"""
Anchor boxes for the five pyramid levels, computed once per input shape on the host and kept on
the device (the reference rebuilds them inside the graph on every call).

Restates, in float32 as the graph-side code does:
    AnchorParameters.default          models/retinanet.py:230-235
    generate_anchors                  utils/anchors.py:234-265   (ratio-major: index = r*4 + s)
    Anchors layer / backend.shift     layers/_misc.py:24-87, backend/common.py:84-114
    concatenation P3..P7              models/retinanet.py:284-311
Anchor id = level offset + (y*W + x)*12 + a, matching the Reshape((-1, k)) of the head outputs
(models/retinanet.py:71,113-121,163).
"""

import numpy as np

PYRAMID_LEVELS = (3, 4, 5, 6, 7)
SIZES = (32, 64, 128, 256, 512)
STRIDES = (8, 16, 32, 64, 128)
RATIOS = (0.5, 1.0, 2.0)
SCALES = (2 ** (-2.0 / 3.0), 2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0))
NUM_BASE_ANCHORS = len(RATIOS) * len(SCALES)


def pyramid_shapes(image_hw):
    """ (H, W) of P3..P7 for an input of image_hw: ceil(size / 2**level) at every level. """
    return [(-(-int(image_hw[0]) // 2 ** l), -(-int(image_hw[1]) // 2 ** l)) for l in PYRAMID_LEVELS]


def base_anchors(size):
    """ the 12 reference windows of one level, centred on the origin, float32 """
    scales = np.asarray(np.asarray(SCALES, dtype=np.float32), dtype=np.float64)   # floatx constants
    ratios = np.asarray(np.asarray(RATIOS, dtype=np.float32), dtype=np.float64)
    out = np.empty((NUM_BASE_ANCHORS, 4), dtype=np.float64)
    for r, ratio in enumerate(ratios):
        for s, scale in enumerate(scales):
            side = size * scale
            w = np.sqrt(side * side / ratio)
            h = w * ratio
            out[r * len(scales) + s] = (0.0 - 0.5 * w, 0.0 - 0.5 * h, w - 0.5 * w, h - 0.5 * h)
    return out.astype(np.float32)


def anchors_for_image(image_hw):
    """ (A, 4) float32 anchors x1 y1 x2 y2 for an input image of image_hw, levels P3..P7. """
    chunks = []
    for (fh, fw), size, stride in zip(pyramid_shapes(image_hw), SIZES, STRIDES):
        base = base_anchors(size)
        cx = (np.arange(fw, dtype=np.float32) + np.float32(0.5)) * np.float32(stride)
        cy = (np.arange(fh, dtype=np.float32) + np.float32(0.5)) * np.float32(stride)
        centre = np.empty((fh, fw, 1, 4), dtype=np.float32)
        centre[..., 0, 0] = cx[None, :]
        centre[..., 0, 1] = cy[:, None]
        centre[..., 0, 2] = cx[None, :]
        centre[..., 0, 3] = cy[:, None]
        chunks.append((centre + base[None, None]).reshape(-1, 4))
    return np.ascontiguousarray(np.concatenate(chunks, axis=0), dtype=np.float32)


def compute_overlap(a, b):
    """ IoU of every box of a (N, 4) with every box of b (K, 4) -> (N, K), float64; the union is
    clamped to machine epsilon so degenerate boxes give 0, not NaN (utils/anchors.py:339-363). """
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    iw = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    union = np.maximum(area_a[:, None] + area_b[None, :] - inter, np.finfo(float).eps)
    return inter / union
