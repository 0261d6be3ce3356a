"""
Winograd F(2, 3) along W for the 3 x 3 tower layers in the f16x3 arithmetic (csrc/conv_wino_impl.h: gpp_wino_transform_f16x3 +
gpp_wino_conv3x3_f16x3), against float64 and against the direct f16x3 kernel on the same pre-split maps.

  * the input transform is exact arithmetic on stored values (differences / sums of two float32, then the split every stored activation
    gets): compared bit for bit with the same formula in torch
  * the position GEMMs + output transform: against the float64 convolution of the values the input map holds, at the bars the direct
    f16x3 kernels are held to (tests/test_conv_f16x3_gpu.py: float32-grade), on pyramids with odd widths, ragged tiles and several images
  * against gpp_conv2d_igemm (direct) on the same maps: two float32-grade evaluations of the same layer
"""
import ctypes
import zlib

import numpy as np
import pytest
import torch

from keras_retinanet_3D.backend import hip
from keras_retinanet_3D.layers import conv as C

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def pyramid_maps(batch, shapes, channels, pitch=None):
    pitch = pitch or channels
    total = sum(h * w for h, w in shapes)
    buf = torch.zeros((batch, total, pitch), dtype=torch.float32, device=DEV)
    maps, off = [], 0
    for h, w in shapes:
        maps.append(C.FMap(buf, batch, h, w, channels, off=off * pitch, bstride=total * pitch, pitch=pitch, split=True, half='f16x3'))
        off += h * w
    return buf, maps


def run_wino(x_levels, kernel, bias, relu, pitch=None):
    """ x_levels: list of (B, H, W, C) float32 tensors (one per level) -> list of (B, H, W, N) outputs of the Winograd path """
    B, Cin = x_levels[0].shape[0], x_levels[0].shape[3]
    Cout = kernel.shape[3]
    shapes = [(t.shape[1], t.shape[2]) for t in x_levels]
    _, src = pyramid_maps(B, shapes, Cin, pitch)
    _, dst = pyramid_maps(B, shapes, Cout)
    for f, t in zip(src, x_levels):
        f.write(t.to(DEV))
    pairs = C.wino_pairs(src)
    v = torch.zeros((B * pairs * 4 * Cin,), dtype=torch.float32, device=DEV)
    w, scale = C.pack_weight_wino(kernel, DEV)
    bias_d = torch.as_tensor(bias).to(DEV)
    dt = C.wino_desc(src, None, B, Cin, Cout, v=v, transform=True)
    dc = C.wino_desc(None, dst, B, Cin, Cout, weight=w, bias=bias_d, out_scale=scale, relu=relu, v=v)
    hip.check(hip.lib().gpp_wino_transform_f16x3(ctypes.byref(dt), hip.stream_ptr()), 'gpp_wino_transform_f16x3')
    hip.check(hip.lib().gpp_wino_conv3x3_f16x3(ctypes.byref(dc), hip.stream_ptr()), 'gpp_wino_conv3x3_f16x3')
    torch.cuda.synchronize()
    return [f.read().cpu() for f in dst], [f.read().cpu() for f in src], v, pairs


def run_direct(x_levels, kernel, bias, relu):
    B, Cin = x_levels[0].shape[0], x_levels[0].shape[3]
    Cout = kernel.shape[3]
    shapes = [(t.shape[1], t.shape[2]) for t in x_levels]
    _, src = pyramid_maps(B, shapes, Cin)
    _, dst = pyramid_maps(B, shapes, Cout)
    for f, t in zip(src, x_levels):
        f.write(t.to(DEV))
    w, b, s = C.pack_weight(kernel, 'f16x3', DEV), torch.as_tensor(bias).to(DEV), C.out_scale_of(kernel, DEV)     # (alive until the launch is through)
    d = C.conv_desc(src, dst, w, b, 3, 3, Cin, Cout, pad=(1, 1), relu=relu, dtype='f16x3', out_scale=s)
    C.run_conv(d)
    torch.cuda.synchronize()
    return [f.read().cpu() for f in dst]


def reference(x, kernel, bias, relu):
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), torch.as_tensor(kernel).double().permute(3, 2, 0, 1), padding=1)
    y = y + torch.as_tensor(bias).double()[None, :, None, None]
    return (torch.relu(y) if relu else y).permute(0, 2, 3, 1)


CASES = {
    'one_level_even': (1, [(6, 8)], 64, 128),
    'one_level_odd_width': (2, [(5, 7)], 64, 128),
    'single_column': (1, [(4, 1)], 32, 128),
    'pyramid_ragged': (2, [(13, 21), (7, 11), (4, 6), (2, 3), (1, 1)], 128, 256),
    'more_than_one_tile': (2, [(20, 33), (10, 17)], 64, 128),          # 2 x 20 x 17 = 680 pairs: four 192-pair tiles on the first level
    'tower_shape_small': (1, [(9, 14)], 512, 512),
}


@pytest.mark.parametrize('case', list(CASES))
@pytest.mark.parametrize('relu', [True, False])
def test_winograd_layer_matches_float64_and_the_direct_kernel(case, relu):
    B, shapes, Cin, Cout = CASES[case]
    rng = np.random.default_rng(zlib.crc32(case.encode()) % 1000)
    xs = [torch.as_tensor(np.maximum(rng.standard_normal((B, h, w, Cin)), 0.0).astype(np.float32) * 3.0) for h, w in shapes]
    kernel = (rng.standard_normal((3, 3, Cin, Cout)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    bias = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    got, stored, _, _ = run_wino(xs, kernel, bias, relu)
    direct = run_direct(xs, kernel, bias, relu)
    for y, x_stored, yd in zip(got, stored, direct):
        ref = reference(x_stored, kernel, bias, relu)
        scale = float(ref.abs().max()) + 1e-30
        err = (y.double() - ref).abs()
        rms = float(torch.sqrt((err ** 2).mean()) / torch.sqrt((ref ** 2).mean() + 1e-30))
        errd = (yd.double() - ref).abs()
        rmsd = float(torch.sqrt((errd ** 2).mean()) / torch.sqrt((ref ** 2).mean() + 1e-30))
        assert rms <= 1e-6 and float(err.max()) <= 4e-6 * scale, (case, rms, float(err.max()) / scale)
        assert rms <= 3.0 * rmsd + 1e-7, (case, rms, rmsd)                      # no worse than the direct kernel by more than rounding noise
        assert float((y.double() - yd.double()).abs().max()) <= 6e-6 * scale


def test_input_transform_is_exact():
    B, shapes, Cin = 2, [(5, 7), (3, 4), (2, 1)], 64
    rng = np.random.default_rng(5)
    xs = [torch.as_tensor(rng.standard_normal((B, h, w, Cin)).astype(np.float32) * 10.0) for h, w in shapes]
    kernel = np.zeros((3, 3, Cin, 128), np.float32)
    _, stored, v, pairs = run_wino(xs, kernel, np.zeros(128, np.float32), False, pitch=96)       # the input map may be a channel slice of a wider tensor
    halves = v.view(torch.float16).reshape(B, pairs, 4, Cin // 32, 2, 32).float().cpu()
    got = (halves[..., 0, :] + halves[..., 1, :]).reshape(B, pairs, 4, Cin)
    pair0 = 0
    for x in stored:
        H, W = x.shape[1], x.shape[2]
        te = (W + 1) // 2
        xp = torch.zeros((B, H, 2 * te + 2, Cin))
        xp[:, :, 1:W + 1] = x
        d = [xp[:, :, j:j + 2 * te:2] for j in range(4)]                        # d_j[tx] = in[2 tx - 1 + j]
        want = torch.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]], dim=3)      # (B, H, te, 4, C) float32
        hi = want.half().float()
        want = hi + (want - hi).half().float()                                  # stored as hi + lo
        blk = got[:, pair0:pair0 + H * te].reshape(B, H, te, 4, Cin)
        assert torch.equal(blk, want)
        pair0 += H * te
    assert pair0 == pairs


def test_arguments_are_validated():
    B, shapes, Cin = 1, [(4, 6)], 64
    _, src = pyramid_maps(B, shapes, Cin)
    _, dst = pyramid_maps(B, shapes, 128)
    v = torch.zeros((B * C.wino_pairs(src) * 4 * Cin,), dtype=torch.float32, device=DEV)
    w, scale = C.pack_weight_wino(np.zeros((3, 3, Cin, 128), np.float32), DEV)
    d = C.wino_desc(None, dst, B, Cin, 128, weight=w, bias=None, out_scale=scale, v=v)
    assert hip.lib().gpp_wino_conv3x3_f16x3(ctypes.byref(d), hip.stream_ptr()) == 0
    d.C_out = 96
    assert hip.lib().gpp_wino_conv3x3_f16x3(ctypes.byref(d), hip.stream_ptr()) == -4            # GPP_ERR_UNSUPPORTED
    d.C_out, d.pairs_per_image = 128, 5
    assert hip.lib().gpp_wino_conv3x3_f16x3(ctypes.byref(d), hip.stream_ptr()) == -1            # GPP_ERR_BAD_ARG
    assert hip.lib().gpp_wino_conv3x3_f16x3(None, hip.stream_ptr()) == -1
    torch.cuda.synchronize()


def test_model_with_winograd_regression_tower(monkeypatch):
    """ GPP_WINO=1: layers 1 - 3 of the regression tower in the Winograd form inside the real plan, against the default plan on the same frames """
    from keras_retinanet_3D import models
    from keras_retinanet_3D.models.retinanet import OP_WINO_CONV, OP_WINO_TRANSFORM
    from keras_retinanet_3D.utils import ledger, synthetic
    B, H, W = 2, 128, 224
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(B, H, W, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)
    planes = synthetic.load_plane_database('100').astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    inputs = [img, np.tile(P_inv[None].astype(np.float32), (B, 1, 1)), np.tile(planes[None], (B, 1, 1))]

    def run(wino):
        monkeypatch.setenv('GPP_WINO', wino)
        m = models.load_model('synthetic:1234', backbone_name='resnet50', dtype='f16x3')
        out = m.predict_on_batch(inputs)
        plan = m.plan_for(B, H, W, 100, True)
        return out, plan, plan.regression.cpu().numpy(), plan.anchor_index.cpu().numpy(), plan.best_index.cpu().numpy(), m

    out0, plan0, reg0, a0, p0, _ = run('0')
    out1, plan1, reg1, a1, p1, m1 = run('1')
    kinds = [op[0] for op in plan1.ops]
    assert kinds.count(OP_WINO_CONV) == 3 and kinds.count(OP_WINO_TRANSFORM) == 3 and OP_WINO_CONV not in [op[0] for op in plan0.ops]
    assert plan1.check_stream_ordering() == []
    assert m1.range_fallbacks == 0
    scale = float(np.abs(reg0).max())
    assert float(np.abs(reg1 - reg0).max()) <= 2e-5 * scale                       # two float32-grade evaluations of three layers + the output layer
    assert not np.array_equal(reg1, reg0)                                          # (it IS another arithmetic)
    led = ledger.parity_ledger(out0, a0, p0, out1, a1, p1)
    assert led['set_differences_unexplained'] == 0 and led['same_orientation'] == led['common'] > 0 and led['same_plane'] == led['common'], led


def test_the_reference_bars_hold_with_the_winograd_tower_at_402x1333(monkeypatch):
    """ the first 8 frames of the resnet50 / 1k-plane fixture through the plan with GPP_WINO=1, against the float64 oracle: utils/ledger.REFERENCE_BARS unchanged """
    import test_fullsize_golden_gpu as G
    from keras_retinanet_3D.utils import ledger
    monkeypatch.setenv('GPP_WINO', '1')
    g64 = G.CD.load_golden('resnet50_1k', 'f64', 8)
    got, events = G.run_hip('resnet50_1k', 'f16x3', frames=8)
    exact = G.CD.compare(g64, got, ledger)
    print('GPP_WINO=1 vs f64: {}/{} common, corners max {:.2e}, scaled beyond 100 m {:.2e}'.format(
        exact['common'], exact['union'], exact['max_corner_dev_m_within_100m'], exact['max_corner_dev_scaled_beyond_100m']))
    assert events == 0 and exact['common'] == exact['union'] == 800 and ledger.meets_reference_bars(exact), exact
