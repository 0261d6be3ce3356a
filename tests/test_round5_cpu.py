"""
Host-side pieces of round 5 that need no GPU: the emulation the fast-convolution numerics study
rests on (oracle/fastconv_numerics.py) against float64, and the 'trained' weight family (models/weights.trained_like).
"""
import numpy as np
import torch

from keras_retinanet_3D.layers import conv as C
from keras_retinanet_3D.models import weights as W
from oracle import fastconv_numerics as FN
from oracle import net_torch


def test_the_emulated_arithmetic_of_the_numerics_study_matches_float64():
    torch.manual_seed(0)
    for (H, Wd, cin, cout) in [(13, 21, 64, 48), (7, 11, 128, 32), (5, 6, 32, 16), (4, 1, 32, 16)]:
        x = FN.stored(torch.relu(torch.randn(1, cin, H, Wd)) * 3)
        k = torch.randn(3, 3, cin, cout, dtype=torch.float64) * 0.05
        ref = torch.nn.functional.conv2d(x.double(), k.permute(3, 2, 0, 1), padding=1)
        for acc in ('mfma', 'blas'):
            FN.ACCUMULATION = acc
            for fn in (lambda: FN.conv_direct(x, k, 1, (1, 1, 1, 1)), lambda: FN.conv_w2(x, k), lambda: FN.conv_w4(x, k)):
                y = fn()
                assert y.shape == ref.shape
                rel = float((y.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
                assert rel < 1e-6, (H, Wd, cin, cout, acc, rel)
    FN.ACCUMULATION = 'mfma'


def test_the_trained_weight_family_computes_the_base_function_with_other_statistics():
    assert W.parse_synthetic('synthetic') == (1234, 'he') and W.parse_synthetic('synthetic:7') == (7, 'he')
    assert W.parse_synthetic('synthetic:1234:trained') == (1234, 'trained') and W.parse_synthetic('synthetic:7.h5') == (7, 'he')
    base = W.synthetic_weights('resnet50', 1234)
    tr = W.synthetic_weights('resnet50', 1234, 'trained')
    assert set(base) == set(tr)
    W.validate_weights(tr, 'resnet50')
    g = tr['bn3b_branch2a/gamma']
    live = g[g > 0]
    assert (g == 0).sum() >= 1 and live.max() / live.min() > 100.0                              # dead channels; scales decades apart
    img = (np.random.default_rng(1).integers(0, 256, size=(1, 96, 160, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32))
    a = net_torch.forward(base, img, 'resnet50', keep_features=True)
    b = net_torch.forward(tr, img, 'resnet50', keep_features=True)
    for level, gain in (('C3', W.TRAINED_STAGE_GAIN[1]), ('C4', W.TRAINED_STAGE_GAIN[2]), ('C5', W.TRAINED_STAGE_GAIN[3])):
        ratio = float(np.abs(b[level]).mean() / np.abs(a[level]).mean())
        assert 0.5 * gain < ratio < 2.0 * gain, (level, ratio)                                  # the residual stream grows stage by stage
    # ... and the heads see (nearly) the base draw's features: only the dead channels change the function
    assert abs(float(b['P3'].std() / a['P3'].std()) - 1.0) < 0.25
    assert float(np.corrcoef(a['classification_logits'].ravel(), b['classification_logits'].ravel())[0, 1]) > 0.8


def test_packed_results_with_the_range_counter_behind_them_unpack_on_the_host():
    """ models/retinanet.py pack_with_range / unpack_with_range: B x 100 x 35 packed detections + the 8 bytes of the f16x3 range counter """
    from keras_retinanet_3D.models.retinanet import RetinaNet3D
    from keras_retinanet_3D.utils import distributed as D
    rng = np.random.default_rng(0)
    B = 3
    packed = rng.standard_normal((B, 100, D.PACK_WIDTH)).astype(np.float32)
    packed[:, :, 16:18] = rng.integers(-1, 4, size=(B, 100, 2))                  # labels, orientations: small integers
    flat = np.empty((B * 100 * D.PACK_WIDTH + 2,), np.float32)
    flat[:-2] = packed.ravel()
    flat[-2:].view(np.uint64)[0] = 123456789012
    outs, count = RetinaNet3D.unpack_with_range(flat, B)
    assert count == 123456789012 and len(outs) == 8
    want = D.unpack_outputs(packed)
    for a, b in zip(outs, want):
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b) and a.flags.writeable
    assert outs[3].dtype == np.int32 and outs[5].shape == (B, 100, 4, 3) and outs[6].shape == (B, 100, 1, 4)
