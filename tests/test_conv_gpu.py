"""
GPU numerics of the implicit-GEMM convolution kernel (C ABI gpp_conv2d_igemm) against a plain
PyTorch float32 reference of the same op on the CPU.  Inputs/weights are rounded to the
16-bit compute type first, so the only differences are float32 accumulation order and the
final rounding of the output to 16 bits:  |err| <= 2^-8 * |ref| + 1e-3 (bf16), 2^-10 (f16),
1e-4 relative for float32 outputs.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from keras_retinanet_3D.backend import hip
from keras_retinanet_3D.layers import conv as C

pytestmark = pytest.mark.gpu


def tf_nearest(x, oh, ow):
    ih, iw = x.shape[1:3]
    ys = torch.clamp(torch.floor(torch.arange(oh, dtype=torch.float32) * (np.float32(ih) / np.float32(oh))).long(), max=ih - 1)
    xs = torch.clamp(torch.floor(torch.arange(ow, dtype=torch.float32) * (np.float32(iw) / np.float32(ow))).long(), max=iw - 1)
    return x[:, ys][:, :, xs]


def reference(x, k, bias, stride, pad_t, pad_l, oh, ow, relu, res):
    """ x (B,H,W,Cin) f32, k HWIO f32 -> (B,oh,ow,Cout) f32 """
    B, H, W, _ = x.shape
    KH, KW = k.shape[:2]
    pad_b = max((oh - 1) * stride + KH - H - pad_t, 0)
    pad_r = max((ow - 1) * stride + KW - W - pad_l, 0)
    xp = F.pad(x.permute(0, 3, 1, 2), (pad_l, pad_r, pad_t, pad_b))
    y = F.conv2d(xp, k.permute(3, 2, 0, 1), bias, stride=stride)[:, :, :oh, :ow].permute(0, 2, 3, 1)
    if res is not None:
        y = y + (tf_nearest(res, oh, ow) if tuple(res.shape[1:3]) != (oh, ow) else res)
    return torch.relu(y) if relu else y


CASES = [
    # name, B, H, W, Cin, Cout, K, stride, pad(t,l), out(h,w) or None (same), relu, residual (None|'same'|(h,w)), out_f32
    ('1x1', 2, 13, 17, 64, 128, 1, 1, (0, 0), None, True, None, False),
    ('1x1_s2', 2, 13, 18, 128, 256, 1, 2, (0, 0), (7, 9), False, None, False),
    ('3x3', 2, 13, 17, 64, 64, 3, 1, (1, 1), None, True, None, False),
    ('3x3_wide', 1, 26, 31, 128, 512, 3, 1, (1, 1), None, False, 'same', False),
    ('3x3_s2_tfsame', 2, 13, 42, 64, 128, 3, 2, None, None, False, None, False),
    ('1x1_res_up', 2, 26, 31, 64, 128, 1, 1, (0, 0), None, False, (13, 16), False),
    ('1x1_res_up_nonint', 1, 51, 67, 64, 128, 1, 1, (0, 0), None, False, (26, 34), False),
    ('head_out36_f32', 2, 9, 11, 128, 36, 3, 1, (1, 1), None, False, None, True),
    ('head_out96_f32', 1, 9, 11, 256, 96, 3, 1, (1, 1), None, False, None, True),
    ('head_out144_f32', 1, 9, 11, 64, 144, 3, 1, (1, 1), None, False, None, True),
    ('bottleneck_2c', 1, 26, 21, 64, 256, 1, 1, (0, 0), None, True, 'same', False),
    ('deepK', 1, 7, 9, 1024, 128, 3, 1, (1, 1), None, True, None, False),
]


@pytest.mark.parametrize('tile', [128, 256, 512])
@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_matches_torch_fp32(case, dtype, tile):
    name, B, H, W, Cin, Cout, K, stride, pad, out_hw, relu, resmode, out_f32 = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    tdt = C.torch_dtype(dtype)
    dev = torch.device('cuda')
    x = torch.randn((B, H, W, Cin), generator=g).to(tdt)
    k = (torch.randn((K, K, Cin, Cout), generator=g) * (2.0 / (K * K * Cin)) ** 0.5).to(tdt)
    bias = torch.randn((Cout,), generator=g) * 0.1
    if pad is None:
        oh, pt = C.same_pad(H, K, stride)
        ow, pl = C.same_pad(W, K, stride)
    else:
        pt, pl = pad
        oh, ow = out_hw if out_hw else (H, W)
    res = None
    if resmode == 'same':
        res = torch.randn((B, oh, ow, Cout), generator=g).to(tdt)
    elif resmode is not None:
        res = torch.randn((B, resmode[0], resmode[1], Cout), generator=g).to(tdt)
    ref = reference(x.float(), k.float(), bias, stride, pt, pl, oh, ow, relu, None if res is None else res.float())

    xin = C.FMap(x.to(dev).contiguous(), B, H, W, Cin)
    out = C.FMap.empty(B, oh, ow, Cout, torch.float32 if out_f32 else tdt, dev)
    out.buf.fill_(float('nan'))
    w = C.pack_weight(k.float().numpy(), dtype, dev)
    rmap = None if res is None else [C.FMap(res.to(dev).contiguous(), B, res.shape[1], res.shape[2], Cout)]
    d = C.conv_desc([xin], [out], w, bias.to(dev), K, K, Cin, Cout, stride=stride, pad=(pt, pl), relu=relu,
                    residuals=rmap, dtype=dtype, out_f32=out_f32, tile_hint=tile)
    C.run_conv(d)
    got = out.buf.float().cpu()
    assert torch.isfinite(got).all()
    eps = 1e-4 if out_f32 else (2.0 ** -8 if dtype == 'bf16' else 2.0 ** -10)
    err = (got - ref).abs()
    tol = eps * ref.abs() + 1e-3
    assert bool((err <= tol).all()), 'max err {} at ref {}'.format(err.max().item(), ref.flatten()[err.argmax()].item())
    assert abs(C.conv_flops(d) - 2.0 * B * oh * ow * K * K * Cin * Cout) < 1.0


@pytest.mark.parametrize('tile', [128, 256, 1128160, 1192128])
def test_grouped_pyramid_launch_and_channel_slices(tile):
    """ five feature maps of different sizes in one launch, inputs read as a channel slice of a
    wider tensor, outputs written at level offsets of one (B, sum(HW), C) pyramid tensor """
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(5)
    B, Cin, Cwide, Cout = 2, 64, 192, 128
    shapes = [(13, 21), (7, 11), (4, 6), (2, 3), (1, 2)]
    total = sum(h * w for h, w in shapes)
    xin = torch.randn((B, total, Cwide), generator=g).to(torch.bfloat16)
    k = (torch.randn((3, 3, Cin, Cout), generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn((Cout,), generator=g) * 0.1
    xd = xin.to(dev).contiguous()
    od = torch.full((B, total, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
    ins, outs, off = [], [], 0
    for h, w in shapes:
        ins.append(C.FMap(xd, B, h, w, Cin, off=off * Cwide + 64, bstride=total * Cwide, pitch=Cwide))
        outs.append(C.FMap(od, B, h, w, Cout, off=off * Cout, bstride=total * Cout))
        off += h * w
    d = C.conv_desc(ins, outs, C.pack_weight(k.float().numpy(), 'bf16', dev), bias.to(dev), 3, 3, Cin, Cout,
                    pad=(1, 1), relu=True, tile_hint=tile)
    C.run_conv(d)
    got = od.float().cpu()
    off = 0
    for h, w in shapes:
        x = xin[:, off:off + h * w, 64:128].float().reshape(B, h, w, Cin)
        ref = reference(x, k.float(), bias, 1, 1, 1, h, w, True, None).reshape(B, h * w, Cout)
        err = (got[:, off:off + h * w] - ref).abs()
        assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-3).all())
        off += h * w


def test_conv_rejects_bad_descriptors():
    from keras_retinanet_3D.backend import hip
    dev = torch.device('cuda')
    x = C.FMap.empty(1, 4, 4, 48, torch.bfloat16, dev)       # C_in not a multiple of 64
    o = C.FMap.empty(1, 4, 4, 64, torch.bfloat16, dev)
    w = torch.zeros((128, 48), dtype=torch.bfloat16, device=dev)
    d = C.conv_desc([x], [o], w, None, 1, 1, 48, 64)
    with pytest.raises(hip.GppError):
        C.run_conv(d)


@pytest.mark.parametrize('split', [0, 2, 3, 8])
@pytest.mark.parametrize('case', ['P6_like', 'res5_like', 'f32_out_resid'])
def test_split_k_matches_torch_fp32(case, split):
    """ deep-K, small-M layers: K range split over blockIdx.y, float32 partial tiles, second-pass reduce """
    B, H, W, Cin, Cout, K, stride, relu, out_f32, with_res = {
        'P6_like': (2, 13, 42, 512, 512, 3, 2, False, False, False),
        'res5_like': (1, 13, 21, 512, 256, 3, 1, True, False, True),
        'f32_out_resid': (1, 9, 11, 1024, 36, 1, 1, False, True, False),
    }[case]
    g = torch.Generator().manual_seed(7)
    dev = torch.device('cuda')
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16)
    k = (torch.randn((K, K, Cin, Cout), generator=g) * (2.0 / (K * K * Cin)) ** 0.5).to(torch.bfloat16)
    bias = torch.randn((Cout,), generator=g) * 0.1
    oh, pt = C.same_pad(H, K, stride)
    ow, pl = C.same_pad(W, K, stride)
    res = torch.randn((B, oh, ow, Cout), generator=g).to(torch.bfloat16) if with_res else None
    ref = reference(x.float(), k.float(), bias, stride, pt, pl, oh, ow, relu, None if res is None else res.float())
    xin = C.FMap(x.to(dev).contiguous(), B, H, W, Cin)
    out = C.FMap.empty(B, oh, ow, Cout, torch.float32 if out_f32 else torch.bfloat16, dev)
    out.buf.fill_(float('nan'))
    ws = torch.empty((32 << 20,), dtype=torch.uint8, device=dev)
    rmap = None if res is None else [C.FMap(res.to(dev).contiguous(), B, oh, ow, Cout)]
    d = C.conv_desc([xin], [out], C.pack_weight(k.float().numpy(), 'bf16', dev), bias.to(dev), K, K, Cin, Cout, stride=stride,
                    pad=(pt, pl), relu=relu, residuals=rmap, out_f32=out_f32, workspace=ws, split_k=split)
    C.run_conv(d)
    got = out.buf.float().cpu()
    eps = 1e-4 if out_f32 else 2.0 ** -8
    err = (got - ref).abs()
    assert bool((err <= eps * ref.abs() + 1e-3).all()), err.max().item()
    C.run_conv(d)                                              # deterministic: fixed summation order over the splits
    assert torch.equal(out.buf.float().cpu(), got)


ALL_TILES = [64064, 96064, 128064, 160064, 192064, 64128, 96128, 128128, 160128, 192128, 224128,
             1128128, 1192128, 1128256, 1192256, 256256,
             128160, 192160, 1192160, 1128160,        # N-remainder tiles (4 x 1 wavefronts, 160 columns)
             1192096]                                 # ... and 96 columns


def _layer(name, dtype='bf16', workspace=False):
    """ one CASES layer on the device: (descriptor factory, output map, float32 reference) """
    _, B, H, W, Cin, Cout, K, stride, pad, out_hw, relu, resmode, out_f32 = [c for c in CASES if c[0] == name][0]
    g = torch.Generator().manual_seed(len(name))
    tdt = C.torch_dtype(dtype)
    dev = torch.device('cuda')
    x = torch.randn((B, H, W, Cin), generator=g).to(tdt)
    k = (torch.randn((K, K, Cin, Cout), generator=g) * (2.0 / (K * K * Cin)) ** 0.5).to(tdt)
    bias = torch.randn((Cout,), generator=g) * 0.1
    if pad is None:
        oh, pt = C.same_pad(H, K, stride)
        ow, pl = C.same_pad(W, K, stride)
    else:
        pt, pl = pad
        oh, ow = out_hw if out_hw else (H, W)
    res = None
    if resmode == 'same':
        res = torch.randn((B, oh, ow, Cout), generator=g).to(tdt)
    elif resmode is not None:
        res = torch.randn((B, resmode[0], resmode[1], Cout), generator=g).to(tdt)
    ref = reference(x.float(), k.float(), bias, stride, pt, pl, oh, ow, relu, None if res is None else res.float())
    xin = C.FMap(x.to(dev).contiguous(), B, H, W, Cin)
    out = C.FMap.empty(B, oh, ow, Cout, torch.float32 if out_f32 else tdt, dev)
    w = C.pack_weight(k.float().numpy(), dtype, dev)
    rmap = None if res is None else [C.FMap(res.to(dev).contiguous(), B, res.shape[1], res.shape[2], Cout)]
    ws = torch.empty((32 << 20,), dtype=torch.uint8, device=dev) if workspace else None
    keep = (xin, w, rmap, ws, bias.to(dev))

    def make(tile, split_k=1):
        return C.conv_desc([xin], [out], w, keep[4], K, K, Cin, Cout, stride=stride, pad=(pt, pl), relu=relu,
                           residuals=rmap, dtype=dtype, out_f32=out_f32, tile_hint=tile, workspace=ws, split_k=split_k)
    return make, out, ref, (2.0 ** -8 if dtype == 'bf16' else 2.0 ** -10) if not out_f32 else 1e-4


@pytest.mark.parametrize('case', ['3x3_wide', '1x1_res_up_nonint', 'head_out144_f32', '3x3_s2_tfsame', 'deepK'])
def test_every_tile_gives_identical_results(case):
    """ The block tile only changes which workgroup computes an output, never the order of its
    K summation: every explicit tile code must reproduce the default bit for bit (and match torch). """
    make, out, ref, eps = _layer(case)
    out.buf.fill_(float('nan'))
    C.run_conv(make(128128))
    base = out.buf.float().cpu()
    assert bool(((base - ref).abs() <= eps * ref.abs() + 1e-3).all())
    for tile in ALL_TILES:
        out.buf.fill_(float('nan'))
        d = make(tile)
        bn = tile % 1000
        if -(-d.C_out // bn) * bn > d.weight_rows:              # the tile grid would read past the packed weight rows
            assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -1
            continue
        C.run_conv(d)
        assert torch.equal(out.buf.float().cpu(), base), tile


def test_unknown_tile_code_is_rejected():
    make, _, _, _ = _layer('3x3')
    rc = hip.lib().gpp_conv2d_igemm(ctypes.byref(make(12345)), hip.stream_ptr())
    assert rc == -1
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(make(3064128)), hip.stream_ptr()) == -1      # the loader-wavefront form of round 2 is gone
    assert rc == -1


@pytest.mark.parametrize('case', ['3x3_wide', 'deepK', 'bottleneck_2c'])
def test_autotune_picks_a_valid_configuration(case):
    make, out, ref, eps = _layer(case, workspace=True)
    d = make(0, split_k=0)
    best = ctypes.c_float(-1.0)
    hip.check(hip.lib().gpp_conv2d_autotune(ctypes.byref(d), 3, hip.stream_ptr(), ctypes.byref(best)), 'gpp_conv2d_autotune')
    assert (d.tile_hint == 0 or d.tile_hint in ALL_TILES) and 0 <= d.split_k <= 16 and 0.0 < best.value < 1e5
    out.buf.fill_(float('nan'))
    C.run_conv(d)
    got = out.buf.float().cpu()
    assert bool(((got - ref).abs() <= eps * ref.abs() + 1e-3).all())
    assert hip.lib().gpp_conv2d_autotune(None, 3, hip.stream_ptr(), None) == -1


@pytest.mark.parametrize('tile_rows', [0, 96, 128, 160])
@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('cmid,B,H,W,shortcut', [(64, 2, 25, 31, 'same'), (128, 1, 26, 21, 'same'), (64, 1, 7, 5, None), (128, 3, 9, 40, 'same')])
def test_bottleneck_tail_equals_the_two_layers(cmid, B, H, W, shortcut, dtype, tile_rows):
    """ gpp_bottleneck_tail (3x3 + 1x1 + residual in one launch, intermediate in LDS) must reproduce
    the two separate launches bit for bit, and the unfused pair is checked against torch here too. """
    g = torch.Generator().manual_seed(cmid + H)
    tdt = C.torch_dtype(dtype)
    dev = torch.device('cuda')
    cout = 4 * cmid
    a = torch.randn((B, H, W, cmid), generator=g).to(tdt)
    k1 = (torch.randn((3, 3, cmid, cmid), generator=g) * (2.0 / (9 * cmid)) ** 0.5).to(tdt)
    k2 = (torch.randn((1, 1, cmid, cout), generator=g) * (2.0 / cmid) ** 0.5).to(tdt)
    b1, b2 = torch.randn((cmid,), generator=g) * 0.1, torch.randn((cout,), generator=g) * 0.1
    sc = torch.randn((B, H, W, cout), generator=g).to(tdt) if shortcut else None
    amap = C.FMap(a.to(dev).contiguous(), B, H, W, cmid)
    mid = C.FMap.empty(B, H, W, cmid, tdt, dev)
    y_sep, y_fused = C.FMap.empty(B, H, W, cout, tdt, dev), C.FMap.empty(B, H, W, cout, tdt, dev)
    w1, w2 = C.pack_weight(k1.float().numpy(), dtype, dev), C.pack_weight(k2.float().numpy(), dtype, dev)
    rmap = [C.FMap(sc.to(dev).contiguous(), B, H, W, cout)] if shortcut else None
    b1d, b2d = b1.to(dev), b2.to(dev)                         # descriptors hold raw pointers: keep the tensors alive
    d1 = C.conv_desc([amap], [mid], w1, b1d, 3, 3, cmid, cmid, pad=(1, 1), relu=True, dtype=dtype)
    d2 = C.conv_desc([mid], [y_sep], w2, b2d, 1, 1, cmid, cout, relu=True, residuals=rmap, dtype=dtype)
    C.run_conv(d1)
    C.run_conv(d2)
    want = y_sep.buf.float().cpu()
    mid_ref = reference(a.float(), k1.float(), b1, 1, 1, 1, H, W, True, None).to(tdt).float()
    ref = reference(mid_ref, k2.float(), b2, 1, 0, 0, H, W, True, None if sc is None else sc.float())
    eps = 2.0 ** -7 if dtype == 'bf16' else 2.0 ** -9       # two roundings: the intermediate may differ by one step
    assert ((want - ref).abs() <= eps * ref.abs() + 2e-2).float().mean() > 0.999
    d2f = C.conv_desc([mid], [y_fused], w2, b2d, 1, 1, cmid, cout, relu=True, residuals=rmap, dtype=dtype)
    mid.buf.fill_(float('nan'))                               # the fused launch must not depend on (or write) it
    y_fused.buf.fill_(float('nan'))
    hip.check(hip.lib().gpp_bottleneck_tail(ctypes.byref(d1), ctypes.byref(d2f), tile_rows, hip.stream_ptr()), 'gpp_bottleneck_tail')
    got = y_fused.buf.float().cpu()
    assert torch.equal(got, want)
    assert torch.isnan(mid.buf.float()).all()


def test_bottleneck_tail_rejects_other_shapes():
    make, _, _, _ = _layer('3x3')                             # 64 -> 64 3x3, fine as first half
    make2, _, _, _ = _layer('1x1_s2')                         # strided 1x1 with 128 input channels
    rc = hip.lib().gpp_bottleneck_tail(ctypes.byref(make(0)), ctypes.byref(make2(0)), 0, hip.stream_ptr())
    assert rc == -4
    assert hip.lib().gpp_bottleneck_tail(None, None, 0, hip.stream_ptr()) == -1
    rc = hip.lib().gpp_bottleneck_tail(ctypes.byref(make(0)), ctypes.byref(make(0)), 0, hip.stream_ptr())   # 3x3 as second half
    assert rc == -4


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('cin,cout,relu,f32', [(128, 384, True, False), (64, 896, True, False), (64, 640, False, True)])
def test_dual_shape_grid_equals_the_plain_tile(cin, cout, relu, f32, dtype):
    """ tile code 2256256 (C_out = 256 k + 128): 256 x 256 tiles for the first C_out - 128 columns and 512 x 128 tiles for the
    last 128 in one grid -- the bits of the ordinary launch, over several feature maps with ragged row counts """
    g = torch.Generator().manual_seed(cout)
    tdt = C.torch_dtype(dtype)
    dev = torch.device('cuda')
    B, shapes = 2, [(21, 29), (11, 15), (6, 8), (3, 4)]
    total = sum(h * w for h, w in shapes)
    x = torch.randn((B, total, cin), generator=g).to(tdt).to(dev)
    w = C.pack_weight((torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5).numpy(), dtype, dev)
    b = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    results = []
    for tile in (128128, 2256256):
        o = torch.full((B, total, cout), float('nan'), dtype=torch.float32 if f32 else tdt, device=dev)
        ins, outs, off = [], [], 0
        for h, wd in shapes:
            ins.append(C.FMap(x, B, h, wd, cin, off=off * cin, bstride=total * cin))
            outs.append(C.FMap(o, B, h, wd, cout, off=off * cout, bstride=total * cout))
            off += h * wd
        C.run_conv(C.conv_desc(ins, outs, w, b, 3, 3, cin, cout, pad=(1, 1), relu=relu, dtype=dtype, out_f32=f32, tile_hint=tile))
        results.append(o.float().cpu())
    assert not torch.isnan(results[1]).any() and torch.equal(results[0], results[1])
    # other widths are refused
    o = torch.empty((B, total, 256), dtype=tdt, device=dev)
    w2 = C.pack_weight((torch.randn((3, 3, cin, 256), generator=g) * 0.05).numpy(), dtype, dev)
    d = C.conv_desc([C.FMap(x, B, total, 1, cin)], [C.FMap(o, B, total, 1, 256)], w2, b[:256].contiguous(), 3, 3, cin, 256, pad=(1, 1), dtype=dtype, tile_hint=2256256)
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -4
