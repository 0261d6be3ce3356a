"""
GPU parity of the HIP decode / NMS kernels (C ABI gpp_detect_f32) against the golden vectors from
the reference's layers and against the CPU oracle: bit-exact (integer + float32 op-by-op work).
"""
import glob
import os

import numpy as np
import pytest

import helpers
from oracle import decode_np
from keras_retinanet_3D.layers.filter_detections import filter_detections
from keras_retinanet_3D.utils import anchors as A

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[len('decode_'):-4] for p in glob.glob(os.path.join(helpers.GOLDEN, 'decode_*.npz')))


def to_fused(regression, n_base=12):
    """ reference layout (B, A, 12) -> fused conv layout (B, P, [4A | 2A | 2A | 2A | 2A]) """
    B, n, _ = regression.shape
    r = regression.reshape(B, n // n_base, n_base, 12)
    parts = [r[..., 0:4].reshape(B, -1, 4 * n_base)] + [r[..., 4 + 2 * k:6 + 2 * k].reshape(B, -1, 2 * n_base) for k in range(4)]
    return np.ascontiguousarray(np.concatenate(parts, axis=2))


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('name', CASES)
def test_hip_decode_matches_reference_goldens(name, fused):
    g = dict(np.load(os.path.join(helpers.GOLDEN, 'decode_{}.npz'.format(name))))
    reg = to_fused(g['regression']) if fused else g['regression']
    out = filter_detections(g['logits'], reg, g['regression_dim'], g['anchors'], fused_regression=fused)
    for got, key in zip(out[:5], ('boxes', 'dimensions', 'scores', 'labels', 'orientations')):
        assert got.dtype == g[key].dtype and helpers.bits_equal(got, g[key]), key
    cand = (np.maximum(g['classification'][..., :4], g['classification'][..., 4:]).max(-1) > np.float32(0.05)).sum(axis=1)
    assert np.array_equal(out[6], cand.astype(np.int32))
    out = filter_detections(g['logits'], reg, g['regression_dim'], g['anchors'], fused_regression=fused, nms=False)
    for got, key in zip(out[:5], ('boxes', 'dimensions', 'scores', 'labels', 'orientations')):
        assert helpers.bits_equal(got, g['nonms_' + key]), 'nms=False ' + key


@pytest.mark.parametrize('hw,batch,mean,std', [((96, 160), 3, -3.5, 1.0), ((128, 416), 2, -4.6, 0.8), ((402, 1333), 1, -4.6, 0.75)])
def test_hip_decode_random_against_oracle(hw, batch, mean, std):
    rng = np.random.default_rng(hw[0] + batch)
    anchors = A.anchors_for_image(hw)
    n = anchors.shape[0]
    logits = rng.normal(mean, std, size=(batch, n, 8)).astype(np.float32)
    reg = rng.normal(0, 1, size=(batch, n, 12)).astype(np.float32)
    dim = rng.normal(0, 1, size=(batch, n, 3)).astype(np.float32)
    ref, ref_idx = decode_np.detect(logits, reg, dim, anchors)
    out = filter_detections(logits, to_fused(reg), dim, anchors, fused_regression=True)
    for got, want in zip(out[:5], ref):
        assert helpers.bits_equal(got, want)
    assert np.array_equal(out[5].astype(np.int64), ref_idx)


def test_hip_decode_more_candidates_than_the_lds_sort_holds():
    """ > 8192 candidates in one image takes the global-memory sort path """
    rng = np.random.default_rng(9)
    anchors = A.anchors_for_image((160, 256))
    n = anchors.shape[0]
    logits = rng.normal(-1.0, 1.0, size=(2, n, 8)).astype(np.float32)
    logits[1] -= 8.0                                          # second image: nothing
    reg = rng.normal(0, 1, size=(2, n, 12)).astype(np.float32)
    dim = rng.normal(0, 1, size=(2, n, 3)).astype(np.float32)
    ref, ref_idx = decode_np.detect(logits, reg, dim, anchors)
    out = filter_detections(logits, reg, dim, anchors)
    assert out[6][0] > 8192 and out[6][1] == 0
    for got, want in zip(out[:5], ref):
        assert helpers.bits_equal(got, want)


def test_hip_decode_score_ties_break_by_anchor_index():
    anchors = A.anchors_for_image((64, 64))
    n = anchors.shape[0]
    logits = np.full((1, n, 8), -9.0, np.float32)
    logits[0, 5::97, 2] = 1.25                                 # many identical scores
    reg = np.zeros((1, n, 12), np.float32)
    dim = np.zeros((1, n, 3), np.float32)
    ref, ref_idx = decode_np.detect(logits, reg, dim, anchors)
    out = filter_detections(logits, reg, dim, anchors)
    assert np.array_equal(out[5].astype(np.int64), ref_idx)
    for got, want in zip(out[:5], ref):
        assert helpers.bits_equal(got, want)


def test_hip_decode_large_k_prefix_and_fallback_paths():
    """ > 8192 candidates: (a) the best-scoring prefix already yields 100 boxes; (b) with a tiny IoU threshold
    nearly everything is suppressed, the prefix yields < 100 boxes and the full global sort takes over """
    rng = np.random.default_rng(21)
    anchors = A.anchors_for_image((160, 256))
    n = anchors.shape[0]
    logits = rng.normal(-1.0, 1.0, size=(1, n, 8)).astype(np.float32)
    reg = rng.normal(0, 1, size=(1, n, 12)).astype(np.float32)
    dim = rng.normal(0, 1, size=(1, n, 3)).astype(np.float32)
    cls = decode_np.sigmoid(logits)
    boxes = decode_np.regress_boxes(anchors[None], reg, cls)
    dims = decode_np.regress_dims(dim)
    for iou in (0.5, 1e-6):
        ref, ref_idx = decode_np.filter_detections(boxes[0], dims[0], cls[0], nms_threshold=iou)
        out = filter_detections(logits, reg, dim, anchors, nms_threshold=iou)
        assert out[6][0] > 8192
        kept = int((ref[2] > 0).sum())
        assert kept == 100 if iou == 0.5 else kept < 100
        for got, want in zip(out[:5], ref):
            assert helpers.bits_equal(got[0], want)
        assert np.array_equal(out[5][0].astype(np.int64), ref_idx)


def test_hip_decode_without_nms():
    """ nms=False (load_model(..., nms=False)): score threshold + top-k only (filter_detections.py:56,159) """
    rng = np.random.default_rng(31)
    anchors = A.anchors_for_image((96, 160))
    n = anchors.shape[0]
    logits = rng.normal(-3.0, 1.0, size=(2, n, 8)).astype(np.float32)
    reg = rng.normal(0, 1, size=(2, n, 12)).astype(np.float32)
    dim = rng.normal(0, 1, size=(2, n, 3)).astype(np.float32)
    cls = decode_np.sigmoid(logits)
    boxes = decode_np.regress_boxes(np.broadcast_to(anchors[None], (2,) + anchors.shape), reg, cls)
    dims = decode_np.regress_dims(dim)
    out = filter_detections(logits, reg, dim, anchors, nms=False)
    for b in range(2):
        ref, ref_idx = decode_np.filter_detections(boxes[b], dims[b], cls[b], nms=False)
        for got, want in zip(out[:5], ref):
            assert helpers.bits_equal(got[b], want)
        assert np.array_equal(out[5][b].astype(np.int64), ref_idx)


def test_stages_enqueued_separately_equal_the_single_call():
    """ gpp_detect_stages_f32(CANDIDATES), (SELECT), (EMIT) one after the other == gpp_detect_f32 (the plan runs the
    first two on a side stream underneath the dimension head); a bad stage mask is refused. """
    import torch
    from keras_retinanet_3D.backend import hip
    from keras_retinanet_3D.layers.filter_detections import FilterDetections
    rng = np.random.default_rng(5)
    anchors = A.anchors_for_image((128, 416))
    n, B = anchors.shape[0], 3
    dev = hip.require_device()
    t = lambda a: torch.as_tensor(a).to(dev)  # noqa: E731
    logits = t(rng.normal(-4.2, 1.0, size=(B, n, 8)).astype(np.float32))
    reg = t(to_fused(rng.normal(0, 1, size=(B, n, 12)).astype(np.float32)))
    dim = t(rng.normal(0, 1, size=(B, n, 3)).astype(np.float32))
    anc = t(anchors)
    whole = FilterDetections(B, n, dev, fused_regression=True)
    want = [o.cpu().numpy().copy() for o in whole(logits, reg, dim, anc) + [whole.anchor_index]]
    assert (want[2] > 0.05).sum() > B * 20
    split = FilterDetections(B, n, dev, fused_regression=True)
    for o in (split.boxes, split.dimensions, split.scores):
        o.fill_(float('nan'))
    args = split.args(logits, reg, dim, anc) + (hip.stream_ptr(),)
    for stage in (1, 2, 4):
        hip.check(hip.lib().gpp_detect_stages_f32(stage, *args), 'gpp_detect_stages_f32')
    got = [o.cpu().numpy() for o in (split.boxes, split.dimensions, split.scores, split.labels, split.orientations, split.anchor_index)]
    for g_, w_ in zip(got, want):
        assert helpers.bits_equal(g_, w_)
    assert hip.lib().gpp_detect_stages_f32(0, *args) == -1 and hip.lib().gpp_detect_stages_f32(8, *args) == -1


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('name', CASES)
def test_orientation_specific_filter_matches_reference_goldens(name, fused):
    """ orientation_specific_filter=True (filter_detections.py:84-98): per-orientation threshold + NMS, concatenation in
    orientation order, common top-k -- against goldens from the reference's own code, with and without NMS """
    g = dict(np.load(os.path.join(helpers.GOLDEN, 'decode_{}.npz'.format(name))))
    reg = to_fused(g['regression']) if fused else g['regression']
    out = filter_detections(g['logits'], reg, g['regression_dim'], g['anchors'], fused_regression=fused, orientation_specific_filter=True)
    for got, key in zip(out[:5], ('boxes', 'dimensions', 'scores', 'labels', 'orientations')):
        assert got.dtype == g['osf_' + key].dtype and helpers.bits_equal(got, g['osf_' + key]), key
    folded = np.maximum(g['classification'][..., :4], g['classification'][..., 4:])
    assert np.array_equal(out[6], (folded > np.float32(0.05)).sum(axis=(1, 2)).astype(np.int32))      # candidates over the four lists
    out = filter_detections(g['logits'], reg, g['regression_dim'], g['anchors'], fused_regression=fused, orientation_specific_filter=True,
                            nms=False)
    for got, key in zip(out[:5], ('boxes', 'dimensions', 'scores', 'labels', 'orientations')):
        assert helpers.bits_equal(got, g['osfnonms_' + key]), 'nms=False ' + key


@pytest.mark.parametrize('hw,batch,mean,std', [((96, 160), 3, -2.5, 1.0), ((402, 1333), 2, -4.2, 0.8)])
def test_orientation_specific_filter_random_against_oracle(hw, batch, mean, std):
    rng = np.random.default_rng(hw[0] * 7 + batch)
    anchors = A.anchors_for_image(hw)
    n = anchors.shape[0]
    logits = rng.normal(mean, std, size=(batch, n, 8)).astype(np.float32)
    logits[:, ::5, 1] = logits[:, ::5, 0]                     # exact score ties between orientations of one anchor
    reg = rng.normal(0, 1, size=(batch, n, 12)).astype(np.float32)
    dim = rng.normal(0, 1, size=(batch, n, 3)).astype(np.float32)
    ref, ref_idx = decode_np.detect(logits, reg, dim, anchors, orientation_specific_filter=True)
    out = filter_detections(logits, to_fused(reg), dim, anchors, fused_regression=True, orientation_specific_filter=True)
    assert (ref[2] > 0).sum() > batch * 50
    for got, want in zip(out[:5], ref):
        assert helpers.bits_equal(got, want)
    assert np.array_equal(out[5].astype(np.int64), ref_idx)
