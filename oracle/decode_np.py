"""
ORACLE (test infrastructure, not product code): NumPy restatement of the reference's detection
decode: anchors, sigmoid, RegressBoxes, RegressDims, filter_detections (NMS, top-k, padding).

Follows (relative to /root/reference/keras_retinanet_3D):
    generate_anchors          utils/anchors.py:234-265
    shift (TF twin, float32)  backend/common.py:84-114
    Anchors layer             layers/_misc.py:24-87, parameters models/retinanet.py:230-235
    RegressBoxes.call         layers/_misc.py:133-141  + bbox_transform_inv backend/common.py:43-81
    RegressDims               layers/_misc.py:186-187  + dim_transform_inv  backend/common.py:23-40
    filter_detections         layers/filter_detections.py:18-189 (default path: nms=True,
                              class_specific_filter=True, orientation_specific_filter=False, 1 class)
Third-party pieces restated from the TF 1.x documentation (absent from /root/reference):
    tf.image.non_max_suppression (greedy, descending score, suppress IoU > threshold, corner
    order normalised, zero-area boxes never overlap), tf.nn.top_k (descending, lower index first
    on ties), tf.sigmoid (1 / (1 + exp(-x)); exp = Cephes expf as in Eigen's packet exp).

Pinned by tests/golden/decode_*.npz = outputs of the reference's own _misc.py / common.py /
filter_detections.py / utils/anchors.py executed on the NumPy stand-in
(oracle/gen_decode_goldens.py).  Everything is float32, one operation at a time, so that the
HIP kernels (csrc/decode.hip, -ffp-contract=off) can match bit for bit.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

import numpy as np

F = np.float32

BOX_MEAN = np.array([-0.0373, -0.0165, 0.0373, 0.0171, -0.0286, -0.0478, 0.2929, 0.0114, 0.0288, -0.0589, 0.2932, -0.0007])  # _misc.py:115
BOX_STD = np.array([0.1957, 0.1896, 0.1957, 0.1897, 0.1967, 0.2034, 0.2046, 0.1898, 0.1964, 0.2052, 0.2048, 0.1903])        # _misc.py:117
DIM_MEAN = np.array([1.6570, 1.7999, 4.2907])   # _misc.py:168
DIM_STD = np.array([0.2681, 0.2243, 0.6281])    # _misc.py:170

SIZES = [32, 64, 128, 256, 512]                 # retinanet.py:230-235
STRIDES = [8, 16, 32, 64, 128]
RATIOS = np.array([0.5, 1, 2], dtype=F)
SCALES = np.array([2 ** (-2.0 / 3.0), 2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)], dtype=F)


# ----------------------------------------------------------------------------- sigmoid
def cephes_expf(x):
    """ Cephes expf (the algorithm of Eigen's pexp), every step a separate float32 operation. """
    x = np.asarray(x, dtype=F)
    x = np.minimum(np.maximum(x, F(-88.3762626647949)), F(88.3762626647949))
    fx = np.floor(x * F(1.44269504088896341) + F(0.5))
    x = (x - fx * F(0.693359375)) - fx * F(-2.12194440e-4)
    z = x * x
    y = F(1.9875691500e-4)
    y = y * x + F(1.3981999507e-3)
    y = y * x + F(8.3334519073e-3)
    y = y * x + F(4.1665795894e-2)
    y = y * x + F(1.6666665459e-1)
    y = y * x + F(5.0000001201e-1)
    y = (y * z + x) + F(1.0)
    return np.ldexp(y, fx.astype(np.int32)).astype(F)


def sigmoid(x):
    return (F(1.0) / (F(1.0) + cephes_expf(-np.asarray(x, dtype=F)))).astype(F)


# ----------------------------------------------------------------------------- anchors
def generate_anchors(base_size, ratios=RATIOS, scales=SCALES):
    """ utils/anchors.py:234-265 (float64 arithmetic, as NumPy promotes it) """
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    n = len(ratios) * len(scales)
    anchors = np.zeros((n, 4))
    anchors[:, 2:] = base_size * np.tile(scales, (2, len(ratios))).T
    areas = anchors[:, 2] * anchors[:, 3]
    anchors[:, 2] = np.sqrt(areas / np.repeat(ratios, len(scales)))
    anchors[:, 3] = anchors[:, 2] * np.repeat(ratios, len(scales))
    anchors[:, 0::2] -= np.tile(anchors[:, 2] * 0.5, (2, 1)).T
    anchors[:, 1::2] -= np.tile(anchors[:, 3] * 0.5, (2, 1)).T
    return anchors


def shift_f32(shape, stride, anchors_f32):
    """ backend/common.py:84-114: the graph-side twin, float32 throughout """
    sx = (np.arange(0, shape[1]).astype(F) + F(0.5)) * F(stride)
    sy = (np.arange(0, shape[0]).astype(F) + F(0.5)) * F(stride)
    gx, gy = np.meshgrid(sx, sy)
    shifts = np.stack([gx.reshape(-1), gy.reshape(-1), gx.reshape(-1), gy.reshape(-1)], axis=0).T
    out = anchors_f32.reshape(1, -1, 4) + shifts.reshape(-1, 1, 4).astype(F)
    return out.reshape(-1, 4).astype(F)


def pyramid_shapes(image_hw):
    """ feature map sizes P3..P7 for an input of image_hw (conv arithmetic, = utils/anchors.py:140-152) """
    h, w = image_hw
    return [((h + 2 ** l - 1) // 2 ** l, (w + 2 ** l - 1) // 2 ** l) for l in (3, 4, 5, 6, 7)]


def anchors_for_image(image_hw):
    """ Anchors layers over P3..P7 concatenated (retinanet.py:284-311): (A, 4) float32 """
    out = []
    for (fh, fw), size, stride in zip(pyramid_shapes(image_hw), SIZES, STRIDES):
        base = generate_anchors(size).astype(F)       # keras.backend.variable(...) -> floatx
        out.append(shift_f32((fh, fw), stride, base))
    return np.concatenate(out, axis=0)


# ----------------------------------------------------------------------------- box / dim decode
def regress_boxes(anchors, regression, classification):
    """ _misc.py:133-141 + common.py:43-81.  anchors (B, A, 4), regression (B, A, 12),
    classification (B, A, 8) sigmoid scores -> boxes (B, A, 12) float32 """
    anchors = np.asarray(anchors, dtype=F)
    d = np.asarray(regression, dtype=F)
    am = np.argmax(classification, axis=2)                      # first maximum
    sign = np.where(am < 4, F(-1.0), F(1.0)).astype(F)
    mean, std = BOX_MEAN.astype(F), BOX_STD.astype(F)
    x1a, y1a, x2a, y2a = (anchors[..., k] for k in range(4))
    w = x2a - x1a
    h = y2a - y1a
    t = [d[..., j] * std[j] + mean[j] for j in range(12)]
    cx = (x1a + x2a) / F(2.0)
    out = [
        x1a + t[0] * w, y1a + t[1] * h, x2a + t[2] * w, y2a + t[3] * h,
        x1a + t[4] * w, y2a + t[5] * h,
        cx + (t[6] * w) * sign, y2a + t[7] * h,
        x2a + t[8] * w, y2a + t[9] * h,
        cx + (t[10] * w) * sign, y1a + t[11] * h,
    ]
    return np.stack(out, axis=2).astype(F)


def regress_dims(regression_dim):
    """ _misc.py:186-187 + common.py:23-40 """
    return (np.asarray(regression_dim, dtype=F) * DIM_STD.astype(F) + DIM_MEAN.astype(F)).astype(F)


# ----------------------------------------------------------------------------- NMS
def _iou_above(a, b, thr):
    ay0, ax0, ay1, ax1 = min(a[0], a[2]), min(a[1], a[3]), max(a[0], a[2]), max(a[1], a[3])
    by0, bx0, by1, bx1 = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])
    area_a = F(F(ay1 - ay0) * F(ax1 - ax0))
    area_b = F(F(by1 - by0) * F(bx1 - bx0))
    if area_a <= 0 or area_b <= 0:
        return False
    ih = max(F(min(ay1, by1) - max(ay0, by0)), F(0.0))
    iw = max(F(min(ax1, bx1) - max(ax0, bx0)), F(0.0))
    inter = F(ih * iw)
    return bool(F(inter / F(F(area_a + area_b) - inter)) > F(thr))


def non_max_suppression(boxes4, scores, max_output_size, iou_threshold):
    """ greedy NMS over candidates; returns indices into boxes4 in selection order """
    boxes4 = np.asarray(boxes4, dtype=F)
    order = np.argsort(-np.asarray(scores, dtype=F), kind='stable')
    keep = []
    for i in order:
        if len(keep) >= max_output_size:
            break
        ok = True
        for j in keep:
            if _iou_above(boxes4[i], boxes4[j], iou_threshold):
                ok = False
                break
        if ok:
            keep.append(int(i))
    return np.asarray(keep, dtype=np.int64)


def fold_classification(classification):
    """ filter_detections.py:78-82,123-125: (A, 8) -> score (A,), orientation (A,) """
    c = np.asarray(classification, dtype=F)
    folded = np.maximum(c[:, :4], c[:, 4:])
    return folded.max(axis=1), np.argmax(folded, axis=1)


def filter_detections(boxes, dimensions, classification, score_threshold=0.05, max_detections=100, nms_threshold=0.5, nms=True,
                      orientation_specific_filter=False):
    """ One image.  boxes (A, 12), dimensions (A, 3), classification (A, 8) sigmoid scores ->
    [boxes (100, 12), dimensions (100, 3), scores (100,), labels (100,) i32, orientations (100,) i32],
    padded with -1, plus the selected anchor indices (for tests).
    orientation_specific_filter (filter_detections.py:84-98): threshold + NMS once per orientation on that orientation's
    folded score, the four survivor lists concatenated in orientation order, then the common top-k; an anchor can appear
    once per orientation. """
    boxes = np.asarray(boxes, dtype=F)
    dimensions = np.asarray(dimensions, dtype=F)
    scores_all, orient_all = fold_classification(classification)
    if orientation_specific_filter:
        c = np.asarray(classification, dtype=F)
        folded = np.maximum(c[:, :4], c[:, 4:])                  # folded[:, o] = classification_all[o::4, 0]
        idx_parts, or_parts = [], []
        for o in range(4):
            so = folded[:, o]
            io = np.nonzero(so > F(score_threshold))[0]
            if nms:
                io = io[non_max_suppression(boxes[io, :4], so[io], max_detections, nms_threshold)]
            idx_parts.append(io)
            or_parts.append(np.full((len(io),), o, dtype=np.int64))
        idx = np.concatenate(idx_parts)
        orient_sel = np.concatenate(or_parts)
        sc = folded[idx, orient_sel]
        order = np.argsort(-sc, kind='stable')[:max_detections]  # tf.nn.top_k over the concatenation
        idx, orient_sel, sc = idx[order], orient_sel[order], sc[order]
        n = len(idx)
        out_boxes = -np.ones((max_detections, 12), dtype=F)
        out_dims = -np.ones((max_detections, 3), dtype=F)
        out_scores = -np.ones((max_detections,), dtype=F)
        out_labels = -np.ones((max_detections,), dtype=np.int32)
        out_orient = -np.ones((max_detections,), dtype=np.int32)
        out_boxes[:n] = boxes[idx]
        out_dims[:n] = dimensions[idx]
        out_scores[:n] = sc
        out_labels[:n] = 0
        out_orient[:n] = orient_sel
        anchor_idx = -np.ones((max_detections,), dtype=np.int64)
        anchor_idx[:n] = idx
        return [out_boxes, out_dims, out_scores, out_labels, out_orient], anchor_idx
    idx = np.nonzero(scores_all > F(score_threshold))[0]
    if nms:                                                      # filter_detections.py:56-64
        keep = non_max_suppression(boxes[idx, :4], scores_all[idx], max_detections, nms_threshold)
        idx = idx[keep]
    sc = scores_all[idx]
    order = np.argsort(-sc, kind='stable')[:max_detections]      # tf.nn.top_k
    idx = idx[order]
    n = len(idx)
    out_boxes = -np.ones((max_detections, 12), dtype=F)
    out_dims = -np.ones((max_detections, 3), dtype=F)
    out_scores = -np.ones((max_detections,), dtype=F)
    out_labels = -np.ones((max_detections,), dtype=np.int32)
    out_orient = -np.ones((max_detections,), dtype=np.int32)
    out_boxes[:n] = boxes[idx]
    out_dims[:n] = dimensions[idx]
    out_scores[:n] = scores_all[idx]
    out_labels[:n] = 0
    out_orient[:n] = orient_all[idx]
    anchor_idx = -np.ones((max_detections,), dtype=np.int64)
    anchor_idx[:n] = idx
    return [out_boxes, out_dims, out_scores, out_labels, out_orient], anchor_idx


def detect(cls_logits, regression, regression_dim, anchors, **filter_kwargs):
    """ Whole decode for a batch: logits (B, A, 8), regression (B, A, 12), regression_dim (B, A, 3),
    anchors (A, 4) -> the five padded tensors of filter_detections, batched, + anchor indices. """
    cls = sigmoid(cls_logits)
    B = cls.shape[0]
    boxes = regress_boxes(np.broadcast_to(anchors[None], (B,) + anchors.shape), regression, cls)
    dims = regress_dims(regression_dim)
    outs, aidx = [], []
    for b in range(B):
        o, a = filter_detections(boxes[b], dims[b], cls[b], **filter_kwargs)
        outs.append(o)
        aidx.append(a)
    return [np.stack([o[k] for o in outs]) for k in range(5)], np.stack(aidx)
