"""
CPU tests: the decode oracle (oracle/decode_np.py) and the product's host-side anchor table
against golden vectors produced by the reference's own layers (oracle/gen_decode_goldens.py).
"""
import glob
import hashlib
import os

import numpy as np
import pytest

import helpers
from oracle import decode_np
from keras_retinanet_3D.utils import anchors as product_anchors

CASES = sorted(os.path.basename(p)[len('decode_'):-4] for p in glob.glob(os.path.join(helpers.GOLDEN, 'decode_*.npz')))


def load(name):
    return dict(np.load(os.path.join(helpers.GOLDEN, 'decode_{}.npz'.format(name))))


@pytest.mark.parametrize('name', CASES)
def test_oracle_decode_matches_reference_goldens(name):
    g = load(name)
    hw = tuple(int(v) for v in g['image_hw'])
    anchors = decode_np.anchors_for_image(hw)
    assert np.array_equal(anchors, g['anchors'])                        # anchor generation + ordering
    B = g['logits'].shape[0]
    boxes = decode_np.regress_boxes(np.broadcast_to(anchors[None], (B,) + anchors.shape), g['regression'], g['classification'])
    assert np.array_equal(boxes, g['all_boxes'])                        # sign rule + 12-value decode, bit for bit
    assert np.array_equal(decode_np.regress_dims(g['regression_dim']), g['all_dims'])
    det, _ = decode_np.detect(g['logits'], g['regression'], g['regression_dim'], anchors)
    for got, key in zip(det, ('boxes', 'dimensions', 'scores', 'labels', 'orientations')):
        assert got.dtype == g[key].dtype and np.array_equal(got, g[key]), key
    for b in range(B):                                                  # nms=False variant (filter_detections.py:56)
        out, _ = decode_np.filter_detections(boxes[b], decode_np.regress_dims(g['regression_dim'])[b], g['classification'][b], nms=False)
        for got, key in zip(out, ('boxes', 'dimensions', 'scores', 'labels', 'orientations')):
            assert np.array_equal(got, g['nonms_' + key][b]), key
        for prefix, kw in (('osf_', {}), ('osfnonms_', {'nms': False})):      # orientation_specific_filter=True (:84-98)
            out, _ = decode_np.filter_detections(boxes[b], decode_np.regress_dims(g['regression_dim'])[b], g['classification'][b],
                                                 orientation_specific_filter=True, **kw)
            for got, key in zip(out, ('boxes', 'dimensions', 'scores', 'labels', 'orientations')):
                assert got.dtype == g[prefix + key].dtype and np.array_equal(got, g[prefix + key][b]), prefix + key


def test_goldens_cover_padding_empty_and_saturated_images():
    assert (load('none')['scores'] == -1).all() and (load('none')['labels'] == -1).all()
    s = load('sparse')
    kept = (s['scores'] > 0).sum(axis=1)
    assert (kept > 0).all() and (kept < 100).all()
    assert (s['boxes'][0, kept[0]:] == -1).all() and (s['orientations'][0, kept[0]:] == -1).all()
    assert ((load('dense')['scores'] > 0).sum(axis=1) == 100).all()
    d = load('small')
    assert (np.diff(d['scores'], axis=1) <= 0).all()                    # descending scores
    assert set(np.unique(d['orientations'])) <= {-1, 0, 1, 2, 3}


def test_full_resolution_anchor_table():
    g = dict(np.load(os.path.join(helpers.GOLDEN, 'anchors_402x1333.npz')))
    for table in (decode_np.anchors_for_image((402, 1333)), product_anchors.anchors_for_image((402, 1333))):
        assert table.shape == (137256, 4) and table.dtype == np.float32 and int(g['count']) == 137256
        # the reference's NumPy generator (float64) agrees to float32 rounding ...
        assert np.abs(table[:24] - g['first']).max() < 1e-4
        assert np.abs(table[-24:] - g['last']).max() < 1e-4
        assert np.abs(table[::1009] - g['rows_every_1009']).max() < 1e-4
        # ... and the graph-side float32 twin (layers.Anchors on the stand-in) agrees bit for bit
        assert hashlib.sha256(table.tobytes()).hexdigest() == str(g['sha256_f32_twin'])
    assert product_anchors.pyramid_shapes((402, 1333)) == [(51, 167), (26, 84), (13, 42), (7, 21), (4, 11)]


@pytest.mark.parametrize('hw', [(64, 96), (40, 72), (33, 40), (375, 1242)])
def test_product_anchor_table_equals_oracle(hw):
    assert np.array_equal(product_anchors.anchors_for_image(hw), decode_np.anchors_for_image(hw))


def test_sigmoid_accuracy_and_monotonicity():
    x = np.linspace(-30, 30, 20001).astype(np.float32)
    s = decode_np.sigmoid(x)
    ref = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    assert np.abs(s - ref).max() < 2e-7
    assert (np.diff(s) >= 0).all()
    assert decode_np.sigmoid(np.float32(-np.log(99.0))) == pytest.approx(0.01, rel=1e-6)
