""" CPU tests of the network oracle and of the product's weight handling (no GPU). """
import numpy as np

from oracle import decode_np, net_torch
from keras_retinanet_3D.models import weights as W

MEAN = np.array([103.939, 116.779, 123.68], np.float32)


def test_layer_inventory_matches_survey_parameter_counts():
    for name, millions in (('resnet50', 58.6), ('resnet101', 77.6), ('resnet152', 93.2)):
        n = 0
        for conv, bn, kh, kw, cin, cout, _ in W.backbone_layers(name):
            n += kh * kw * cin * cout + 2 * cout                  # BN folded to scale + shift
        for _, k, cin, cout, _ in W.fpn_layers():
            n += k * k * cin * cout + cout
        for _, cin, cout, _ in W.head_layers():
            n += 9 * cin * cout + cout
        assert abs(n / 1e6 - millions) < 0.15, (name, n)
    names = [l[0] for l in W.backbone_layers('resnet101')]
    assert 'res4b22_branch2c' in names and 'res3b3_branch2a' in names and 'res2c_branch2a' in names and 'res5c_branch2b' in names
    assert 'res4f_branch2a' in [l[0] for l in W.backbone_layers('resnet50')]


def test_bn_folding_and_shapes_small_image():
    w = W.synthetic_weights('resnet50', 3)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(1, 67, 99, 3)).astype(np.float32) - MEAN
    lit = net_torch.forward(w, img, 'resnet50', storage=None, keep_features=True)
    # literal BN == folded BN (same float32 mode, weights folded by the product code)
    folded = dict(w)
    for conv, bn, *_ in W.backbone_layers('resnet50'):
        k, b = W.folded_conv(w, conv, bn)
        folded[conv + '/kernel'] = k
        folded[bn + '/gamma'] = np.ones_like(b)
        folded[bn + '/beta'] = b
        folded[bn + '/moving_mean'] = np.zeros_like(b)
        folded[bn + '/moving_variance'] = np.ones_like(b) - np.float32(1e-5)
    fol = net_torch.forward(folded, img, 'resnet50', storage=None, keep_features=True)
    for key in ('C3', 'C5', 'P3', 'P7', 'regression', 'classification_logits'):
        assert np.abs(lit[key] - fol[key]).max() < 2e-3 * max(1.0, np.abs(lit[key]).max()), key
    # conv arithmetic of keras_resnet / TF 'same': ceil(size / 2**level)
    assert lit['C2'].shape[1:3] == (17, 25) and lit['C3'].shape[1:3] == (9, 13) and lit['C5'].shape[1:3] == (3, 4)
    assert lit['P6'].shape[1:3] == (2, 2) and lit['P7'].shape[1:3] == (1, 1)
    assert lit['regression'].shape == (1, decode_np.anchors_for_image((67, 99)).shape[0], 12)
    # fused regression output kernel keeps the reference's op1..op5 channel order
    k, b = W.fused_regression_outputs(w)
    assert k.shape == (3, 3, 512, 144) and np.array_equal(k[..., 48:72], w['pyramid_regression_op2/kernel'])


def test_synthetic_statistics_reduced_image():
    """ activations stay O(1), logits sit around the prior bias, deltas ~ N(0, 1) """
    w = W.synthetic_weights('resnet50', 1234)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(1, 128, 416, 3)).astype(np.float32) - MEAN
    out = net_torch.forward(w, img, 'resnet50', keep_features=True)
    for key in ('C3', 'C4', 'C5', 'P3', 'P5'):
        rms = np.sqrt((out[key] ** 2).mean())
        assert 0.5 < rms < 8.0, (key, rms)
    assert abs(out['classification_logits'].mean() + 4.6) < 0.3 and 0.3 < out['classification_logits'].std() < 0.9
    assert 0.5 < out['regression'].std() < 2.0 and 0.5 < out['regression_dim'].std() < 2.0
    q = net_torch.forward(w, img, 'resnet50', storage='bf16')
    assert np.abs(q['classification_logits'] - out['classification_logits']).max() < 0.15


def test_weight_file_round_trip(tmp_path):
    w = W.synthetic_weights('resnet50', 5)
    small = {k: v for k, v in w.items() if k.startswith('P7') or k.startswith('bn_conv1')}
    W.save_weights(str(tmp_path / 'w.npz'), small)
    back = W.load_weights(str(tmp_path / 'w.npz'))
    assert set(back) == set(small) and all(np.array_equal(back[k], small[k]) for k in small)
    assert np.array_equal(W.synthetic_weights('resnet50', 5)['P7/kernel'], w['P7/kernel'])     # seeded, reproducible


def test_weight_validation_names_missing_and_misshaped_arrays():
    """ models.weights.validate_weights: one clear error instead of a KeyError deep in the plan builder (a converted
    checkpoint with a missing / transposed layer); synthetic weights of every backbone validate """
    import pytest
    from keras_retinanet_3D.models import weights as W
    for backbone in ('resnet50', 'resnet101', 'resnet152'):
        exp = W.expected_arrays(backbone)
        assert 'res5c_branch2c/kernel' in exp and exp['pyramid_regression_op1/kernel'] == (3, 3, 512, 48)
    w = W.synthetic_weights('resnet50', 3)
    W.validate_weights(w, 'resnet50')
    broken = dict(w)
    del broken['P4/kernel']
    broken['res3b_branch2b/kernel'] = np.transpose(broken['res3b_branch2b/kernel'], (3, 2, 0, 1))      # OIHW instead of HWIO
    with pytest.raises(ValueError) as e:
        W.validate_weights(broken, 'resnet50')
    assert 'P4/kernel' in str(e.value) and 'res3b_branch2b/kernel' in str(e.value) and '1 arrays missing' in str(e.value)
    with pytest.raises(ValueError):
        W.validate_weights(w, 'resnet101')              # resnet50 weights are not a resnet101
