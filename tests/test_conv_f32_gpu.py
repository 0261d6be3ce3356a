"""
GPU numerics of the reference-precision (GPP_F32) convolution path: float32 activations and weights on
v_mfma_f32_16x16x4_f32 -- every product rounded once, float32 accumulation, i.e. the arithmetic type of the reference
(keras.backend.floatx() = float32, /root/reference/keras_retinanet_3D/utils/image.py:47).

Checked against a float64 torch reference of the same op on UNROUNDED random float32 operands.  The only difference
is the float32 summation (order and rounding of K = KH*KW*C_in <= 9216 terms): |err| <= 2e-6 * sum_k |a_k b_k| would be
the textbook bound; the tests use the simpler |err| <= 1e-5 * |ref| + 2e-6 * rms(ref) * sqrt(K) stated next to each assert.
Also: stem / max-pool / ReLU in float32, split-K, tile invariance, grouped pyramid launches, channel slices.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from keras_retinanet_3D.backend import hip
from keras_retinanet_3D.layers import conv as C
from test_conv_gpu import CASES, tf_nearest

pytestmark = pytest.mark.gpu

F32_TILES = [64064, 96064, 128064, 160064, 192064, 64128, 96128, 128128, 160128, 192128, 224128, 128160, 192160]
X3_TILES = F32_TILES + [128256, 192256, 256256]          # GPP_BF16X3 also has the 8-wavefront 256-column tiles (plain loop)
# ... and not the 192 x 160 tile on float32 input maps (3 registers over the budget -> scratch: removed in round 3; the pre-split
# form exists): GPP_ERR_UNSUPPORTED there
X3_F32IN_TILES = [t for t in X3_TILES if t != 192160]


def reference64(x, k, bias, stride, pad_t, pad_l, oh, ow, relu, res):
    """ float64 reference: x (B,H,W,Cin), k HWIO -> (B,oh,ow,Cout) """
    B, H, W, _ = x.shape
    KH, KW = k.shape[:2]
    pad_b = max((oh - 1) * stride + KH - H - pad_t, 0)
    pad_r = max((ow - 1) * stride + KW - W - pad_l, 0)
    xp = F.pad(x.double().permute(0, 3, 1, 2), (pad_l, pad_r, pad_t, pad_b))
    y = F.conv2d(xp, k.double().permute(3, 2, 0, 1), bias.double(), stride=stride)[:, :, :oh, :ow].permute(0, 2, 3, 1)
    if res is not None:
        y = y + (tf_nearest(res, oh, ow) if tuple(res.shape[1:3]) != (oh, ow) else res).double()
    return torch.relu(y) if relu else y


def _layer(case, seed_shift=0, dtype='f32'):
    name, B, H, W, Cin, Cout, K, stride, pad, out_hw, relu, resmode, _ = case
    g = torch.Generator().manual_seed(sum(map(ord, name)) + seed_shift)
    dev = torch.device('cuda')
    x = torch.randn((B, H, W, Cin), generator=g)
    k = torch.randn((K, K, Cin, Cout), generator=g) * (2.0 / (K * K * Cin)) ** 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    if pad is None:
        oh, pt = C.same_pad(H, K, stride)
        ow, pl = C.same_pad(W, K, stride)
    else:
        pt, pl = pad
        oh, ow = out_hw if out_hw else (H, W)
    res = None
    if resmode == 'same':
        res = torch.randn((B, oh, ow, Cout), generator=g)
    elif resmode is not None:
        res = torch.randn((B, resmode[0], resmode[1], Cout), generator=g)
    ref = reference64(x, k, bias, stride, pt, pl, oh, ow, relu, res)
    xin = C.FMap(x.to(dev).contiguous(), B, H, W, Cin)
    out = C.FMap.empty(B, oh, ow, Cout, torch.float32, dev)
    w = C.pack_weight(k.numpy(), dtype, dev)
    rmap = None if res is None else [C.FMap(res.to(dev).contiguous(), B, res.shape[1], res.shape[2], Cout)]
    ws = torch.empty((32 << 20,), dtype=torch.uint8, device=dev)
    keep = (xin, w, rmap, ws, bias.to(dev))

    def make(tile, split_k=1, workspace=False):
        return C.conv_desc([xin], [out], w, keep[4], K, K, Cin, Cout, stride=stride, pad=(pt, pl), relu=relu, residuals=rmap,
                           dtype=dtype, tile_hint=tile, workspace=ws if workspace else None, split_k=split_k)
    return make, out, ref, K * K * Cin, keep


def _check(out, ref, kdepth):
    got = out.buf.double().cpu()
    assert torch.isfinite(got).all()
    rms = float(ref.pow(2).mean().sqrt())
    err = (got - ref).abs()
    tol = 1e-5 * ref.abs() + 2e-6 * rms * kdepth ** 0.5          # float32 summation of kdepth terms
    assert bool((err <= tol).all()), 'max err {} (rms {})'.format(err.max().item(), rms)


@pytest.mark.parametrize('tile', [0, 64064, 128128, 192128, 128160])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_f32_matches_float64_reference(case, tile):
    make, out, ref, kdepth, _ = _layer(case)
    out.buf.fill_(float('nan'))
    d = make(tile)
    bn = tile % 1000 if tile else 64
    if -(-d.C_out // bn) * bn > d.weight_rows:
        pytest.skip('tile grid would read past the packed weight rows')
    C.run_conv(d)
    _check(out, ref, kdepth)
    assert abs(C.conv_flops(d) - 2.0 * ref.numel() * kdepth) < 1.0


@pytest.mark.parametrize('case', ['3x3_wide', '1x1_res_up_nonint', 'head_out144_f32', '3x3_s2_tfsame', 'deepK'])
def test_every_f32_tile_gives_identical_results(case):
    """ the block tile never changes the K order of an output element: all float32 tiles agree bit for bit """
    make, out, ref, kdepth, _ = _layer([c for c in CASES if c[0] == case][0])
    C.run_conv(make(128128))
    base = out.buf.clone()
    _check(out, ref, kdepth)
    for tile in F32_TILES:
        out.buf.fill_(float('nan'))
        d = make(tile)
        bn = tile % 1000
        if -(-d.C_out // bn) * bn > d.weight_rows:
            assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -1
            continue
        C.run_conv(d)
        assert torch.equal(out.buf, base), tile


@pytest.mark.parametrize('tile', [0, 64064, 128128, 192128, 128160])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_bf16x3_matches_float64_reference(case, tile):
    """ GPP_BF16X3: float32 storage, x*w computed as hi*whi + hi*wlo + lo*whi on the bf16 matrix pipe.  Per product the
    dropped lo*wlo term and the roundings of the lo parts leave ~2^-16 |x w|; over K random-sign terms that is about
    1e-5 * rms(ref) -- bar: |err| <= 1e-4 * |ref| + 1e-4 * rms(ref)  (plain bf16 operands: 4e-3; float32: 1e-6). """
    make, out, ref, kdepth, _ = _layer(case, dtype='bf16x3')
    out.buf.fill_(float('nan'))
    d = make(tile)
    if tile == 192160:
        assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -4      # float32 input map: no such tile
        return
    bn = tile % 1000 if tile else 64
    if -(-d.C_out // bn) * bn > d.weight_rows:
        pytest.skip('tile grid would read past the packed weight rows')
    C.run_conv(d)
    got = out.buf.double().cpu()
    assert torch.isfinite(got).all()
    rms = float(ref.pow(2).mean().sqrt())
    err = (got - ref).abs()
    assert bool((err <= 1e-4 * ref.abs() + 1e-4 * rms).all()), 'max err {} (rms {})'.format(err.max().item(), rms)
    assert float(err.pow(2).mean().sqrt()) < 2e-5 * rms + 1e-7          # and ~100x closer than bf16 operands on average


@pytest.mark.parametrize('case', ['3x3_wide', '1x1_res_up_nonint', 'head_out144_f32', '3x3_s2_tfsame', 'deepK'])
def test_every_bf16x3_tile_gives_identical_results(case):
    make, out, _, _, _ = _layer([c for c in CASES if c[0] == case][0], dtype='bf16x3')
    C.run_conv(make(128128))
    base = out.buf.clone()
    for tile in X3_F32IN_TILES:
        out.buf.fill_(float('nan'))
        d = make(tile)
        if -(-d.C_out // (tile % 1000)) * (tile % 1000) > d.weight_rows:
            continue
        C.run_conv(d)
        assert torch.equal(out.buf, base), tile
    if case in ('3x3_wide', 'deepK'):                                   # enough K-steps to split
        out.buf.fill_(float('nan'))
        C.run_conv(make(96128, split_k=3, workspace=True))
        first = out.buf.clone()
        C.run_conv(make(192128, split_k=3, workspace=True))           # split-K: same summation order on every tile
        assert torch.equal(out.buf, first)


def test_pipelined_and_wide_tiles_are_16_bit_only():
    make, _, _, _, _ = _layer(CASES[3])
    for tile in (256256, 1128128, 1192256, 2256256):
        assert hip.lib().gpp_conv2d_igemm(ctypes.byref(make(tile)), hip.stream_ptr()) == -4      # GPP_ERR_UNSUPPORTED
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(make(12345)), hip.stream_ptr()) == -1


@pytest.mark.parametrize('split', [2, 3, 8])
@pytest.mark.parametrize('case', ['deepK', '3x3_wide'])
def test_f32_split_k(case, split):
    make, out, ref, kdepth, _ = _layer([c for c in CASES if c[0] == case][0])
    out.buf.fill_(float('nan'))
    d = make(128128, split_k=split, workspace=True)
    C.run_conv(d)
    _check(out, ref, kdepth)
    first = out.buf.clone()
    out.buf.fill_(float('nan'))
    C.run_conv(make(64128, split_k=split, workspace=True))           # another tile, same split: same summation order
    assert torch.equal(out.buf, first)


def test_split_rule_depends_on_the_layer_only():
    """ gpp_conv2d_split_rule: same answer for every batch size and tile; res5-like and P6-like layers are split """
    dev = torch.device('cuda')
    ws = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
    for dtype in ('bf16', 'f32'):
        tdt = C.torch_dtype(dtype)
        seen = {}
        for B in (1, 2, 8):
            for name, (H, W, Cin, Cout, K, stride) in {'res5_2b': (13, 42, 512, 512, 3, 1), 'P6': (13, 42, 2048, 512, 3, 2),
                                                       'res4_2b': (26, 84, 256, 256, 3, 1), 'C5_reduced': (13, 42, 2048, 512, 1, 1),
                                                       'P3': (51, 167, 512, 512, 3, 1)}.items():
                oh, pt = C.same_pad(H, K, stride)
                ow, pl = C.same_pad(W, K, stride)
                x = C.FMap.empty(B, H, W, Cin, tdt, dev)
                o = C.FMap.empty(B, oh, ow, Cout, tdt, dev)
                w = torch.zeros((512, K * K * Cin), dtype=tdt, device=dev)
                for tile in (0, 64128, 128128):
                    d = C.conv_desc([x], [o], w, None, K, K, Cin, Cout, stride=stride, pad=(pt, pl), dtype=dtype, tile_hint=tile, workspace=ws)
                    s = C.split_rule(d)
                    assert seen.setdefault(name, s) == s, (name, B, tile)
                    assert C.workspace_bytes(d) >= (s > 1) * s * B * oh * ow * Cout * 4
        assert seen['res5_2b'] == 3 and seen['P6'] == 8 and seen['res4_2b'] == 1 and seen['C5_reduced'] == 1 and seen['P3'] == 1, seen


def test_f32_grouped_pyramid_and_channel_slices():
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(5)
    B, Cin, Cwide, Cout = 2, 64, 160, 128
    shapes = [(13, 21), (7, 11), (4, 6), (2, 3), (1, 2)]
    total = sum(h * w for h, w in shapes)
    xin = torch.randn((B, total, Cwide), generator=g)
    k = torch.randn((3, 3, Cin, Cout), generator=g) * 0.05
    bias = torch.randn((Cout,), generator=g) * 0.1
    xd = xin.to(dev).contiguous()
    od = torch.full((B, total, Cout), float('nan'), dtype=torch.float32, device=dev)
    ins, outs, off = [], [], 0
    for h, w in shapes:
        ins.append(C.FMap(xd, B, h, w, Cin, off=off * Cwide + 32, bstride=total * Cwide, pitch=Cwide))
        outs.append(C.FMap(od, B, h, w, Cout, off=off * Cout, bstride=total * Cout))
        off += h * w
    d = C.conv_desc(ins, outs, C.pack_weight(k.numpy(), 'f32', dev), bias.to(dev), 3, 3, Cin, Cout, pad=(1, 1), relu=True, dtype='f32')
    C.run_conv(d)
    got = od.double().cpu()
    off = 0
    for h, w in shapes:
        x = xin[:, off:off + h * w, 32:96].reshape(B, h, w, Cin)
        ref = reference64(x, k, bias, 1, 1, 1, h, w, True, None).reshape(B, h * w, Cout)
        err = (got[:, off:off + h * w] - ref).abs()
        assert bool((err <= 1e-5 * ref.abs() + 1e-5).all()), err.max().item()
        off += h * w


def test_f32_rejects_bad_descriptors():
    dev = torch.device('cuda')
    x = C.FMap.empty(1, 4, 4, 48, torch.float32, dev)        # C_in not a multiple of 32
    o = C.FMap.empty(1, 4, 4, 64, torch.float32, dev)
    w = torch.zeros((256, 48), dtype=torch.float32, device=dev)
    with pytest.raises(hip.GppError):
        C.run_conv(C.conv_desc([x], [o], w, None, 1, 1, 48, 64, dtype='f32'))
    x = C.FMap.empty(1, 4, 4, 64, torch.float32, dev)
    w = torch.zeros((256, 64), dtype=torch.float32, device=dev)
    d = C.conv_desc([x], [o], w, None, 1, 1, 64, 64, dtype='f32', diag=1)     # diagnostic bits: production library refuses
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -1
    d16 = C.conv_desc([x], [o], w, None, 1, 1, 64, 64, dtype='f32')
    assert hip.lib().gpp_bottleneck_tail(ctypes.byref(d16), ctypes.byref(d16), 0, hip.stream_ptr()) == -4


def test_f32_stem_pool_relu():
    """ conv1 7x7/2 + folded bn + ReLU, 3x3/2 'same' max-pool and ReLU on float32 maps """
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 37, 53
    img = (torch.rand((B, H, W, 3), generator=g) * 255.0 - 115.0)
    k = torch.randn((7, 7, 3, 64), generator=g) * 0.01
    bias = torch.randn((64,), generator=g) * 0.1
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    ref = torch.relu(F.conv2d(F.pad(img.double().permute(0, 3, 1, 2), (3, 3, 3, 3)), k.double().permute(3, 2, 0, 1), bias.double(), stride=2))
    out = torch.full((B, Ho, Wo, 64), float('nan'), dtype=torch.float32, device=dev)
    wd, img_d, bias_d = k.reshape(147, 64).contiguous().to(dev), img.to(dev).contiguous(), bias.to(dev)      # kept alive over the launch
    hip.check(hip.lib().gpp_stem_conv7x7_bn_relu(hip.ptr(img_d), hip.ptr(wd), hip.ptr(bias_d), hip.ptr(out),
                                                 hip.GPP_F32, B, H, W, hip.stream_ptr()), 'stem')
    got = out.double().cpu().permute(0, 3, 1, 2)
    assert bool(((got - ref).abs() <= 1e-5 * ref.abs() + 2e-5).all()), (got - ref).abs().max().item()
    Hp, Wp = (Ho + 1) // 2, (Wo + 1) // 2
    pooled = torch.full((B, Hp, Wp, 64), float('nan'), dtype=torch.float32, device=dev)
    hip.check(hip.lib().gpp_maxpool3x3s2_same(hip.ptr(out), hip.ptr(pooled), hip.GPP_F32, B, Ho, Wo, 64, hip.stream_ptr()), 'pool')
    pt = max((Hp - 1) * 2 + 3 - Ho, 0) // 2
    pl = max((Wp - 1) * 2 + 3 - Wo, 0) // 2
    xp = F.pad(out.cpu().permute(0, 3, 1, 2), (pl, 2, pt, 2), value=float('-inf'))
    want = F.max_pool2d(xp, 3, 2)[:, :, :Hp, :Wp].permute(0, 2, 3, 1)
    assert torch.equal(pooled.cpu(), want)
    x = torch.randn((3, 40, 64), generator=g).to(dev)
    y = torch.empty_like(x)
    hip.check(hip.lib().gpp_relu(hip.ptr(x), hip.ptr(y), hip.GPP_F32, x.numel(), hip.stream_ptr()), 'relu')
    assert torch.equal(y, torch.relu(x))


def _x3_round(t):
    """ what a pre-split map can hold: hi + lo with hi = bf16(x), lo = bf16(x - hi) (about 16 significant bits) """
    hi = t.to(torch.bfloat16).float()
    return hi + (t - hi).to(torch.bfloat16).float()


@pytest.mark.parametrize('flags', [1, 2, 4, 3, 7])
@pytest.mark.parametrize('tile', [0, 96128, 192128, 128256, 1256256, 1192128])
@pytest.mark.parametrize('case', ['3x3_wide', '1x1_res_up_nonint', 'bottleneck_2c', '3x3_s2_tfsame', 'deepK'])
def test_bf16x3_pre_split_maps(case, tile, flags):
    """ gpp_conv_desc.x3_split: input / output / shortcut maps stored as [32 bf16 hi | 32 bf16 lo] per 32 channels.  The matrix
    loop then reads its operands as they are; the epilogue splits what it stores and re-joins the shortcut it reads.  Against the
    float64 reference on the values the maps actually hold, with the bar of the float32-storage form, plus the 2^-17 of a split
    output; split-K (deepK), every combination of the three flags, plain and 8-wavefront tiles. """
    name, B, H, W, Cin, Cout, K, stride, pad, out_hw, relu, resmode, _ = [c for c in CASES if c[0] == case][0]
    if (flags & 4) and resmode is None:
        pytest.skip('no shortcut in this layer')
    if Cin % 32 or ((flags & 6) and Cout % 32):
        pytest.skip('pre-split maps hold whole 32-channel blocks')
    g = torch.Generator().manual_seed(sum(map(ord, name)) + flags)
    dev = torch.device('cuda')
    x = torch.randn((B, H, W, Cin), generator=g)
    k = torch.randn((K, K, Cin, Cout), generator=g) * (2.0 / (K * K * Cin)) ** 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    if pad is None:
        oh, pt = C.same_pad(H, K, stride)
        ow, pl = C.same_pad(W, K, stride)
    else:
        pt, pl = pad
        oh, ow = out_hw if out_hw else (H, W)
    res = None
    if resmode == 'same':
        res = torch.randn((B, oh, ow, Cout), generator=g)
    elif resmode is not None:
        res = torch.randn((B, resmode[0], resmode[1], Cout), generator=g)
    x_held = _x3_round(x) if flags & 1 else x
    res_held = None if res is None else (_x3_round(res) if flags & 4 else res)
    ref = reference64(x_held, k, bias, stride, pt, pl, oh, ow, relu, res_held)
    xin = C.FMap.empty(B, H, W, Cin, torch.float32, dev)
    xin.split = bool(flags & 1)
    xin.write(x)
    out = C.FMap.empty(B, oh, ow, Cout, torch.float32, dev)
    out.split = bool(flags & 2)
    out.buf.fill_(float('nan'))
    rmap = None
    if res is not None:
        rm = C.FMap.empty(B, res.shape[1], res.shape[2], Cout, torch.float32, dev)
        rm.split = bool(flags & 4)
        rm.write(res)
        rmap = [rm]
    w = C.pack_weight(k.numpy(), 'bf16x3', dev)
    ws = torch.empty((32 << 20,), dtype=torch.uint8, device=dev)
    split_k = 3 if case == 'deepK' else 1
    d = C.conv_desc([xin], [out], w, bias.to(dev), K, K, Cin, Cout, stride=stride, pad=(pt, pl), relu=relu, residuals=rmap,
                    dtype='bf16x3', tile_hint=tile, workspace=ws, split_k=split_k)
    assert d.x3_split == (flags if res is not None else flags & 3)
    bn = tile % 1000 if tile else 64
    if -(-d.C_out // bn) * bn > d.weight_rows:
        pytest.skip('tile grid would read past the packed weight rows')
    if tile > 1000000 and not (flags & 1):
        assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -4      # the pipelined loop reads pre-split rows
        return
    C.run_conv(d)
    got = out.read().double().cpu()
    assert torch.isfinite(got).all()
    rms = float(ref.pow(2).mean().sqrt())
    err = (got - ref).abs()
    assert bool((err <= 1.2e-4 * ref.abs() + 1e-4 * rms).all()), 'max err {} (rms {})'.format(err.max().item(), rms)
    assert float(err.pow(2).mean().sqrt()) < 2.5e-5 * rms + 1e-7


def test_bf16x3_pre_split_flags_are_validated():
    make, _, _, _, _ = _layer([c for c in CASES if c[0] == '3x3_wide'][0], dtype='f32')
    d = make(0)
    d.x3_split = 1                                             # only GPP_BF16X3 knows the layout
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -1
    make, _, _, _, _ = _layer([c for c in CASES if c[0] == 'head_out144_f32'][0], dtype='bf16x3')
    d = make(0)
    d.x3_split = 2                                             # 144 output channels: not whole 32-channel blocks
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -4
    d.x3_split = 8
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -1


def test_relu_on_a_pre_split_map():
    dev = torch.device('cuda')
    x = torch.randn((2, 5, 7, 64))
    src = C.FMap.empty(2, 5, 7, 64, torch.float32, dev)
    dst = C.FMap.empty(2, 5, 7, 64, torch.float32, dev)
    src.split = dst.split = True
    src.write(x)
    hip.check(hip.lib().gpp_relu(hip.ptr(src.buf), hip.ptr(dst.buf), hip.GPP_BF16X3, x.numel(), hip.stream_ptr()))
    assert torch.equal(dst.read().cpu(), torch.relu(_x3_round(x)))


X3_PIPE_TILES = [1128128, 1192128, 1128256, 1192256, 1256256, 1128160, 1192096]


@pytest.mark.parametrize('case', ['3x3_wide', 'bottleneck_2c', '3x3_s2_tfsame', 'deepK', '1x1'])
def test_every_bf16x3_tile_on_pre_split_maps_gives_identical_results(case):
    """ plain and software-pipelined (three-phase) tiles sum an output element in the same order: hi * wlo, hi * whi, lo * whi per
    K-step -- identical bits, with and without split-K """
    name, B, H, W, Cin, Cout, K, stride, pad, out_hw, relu, resmode, _ = [c for c in CASES if c[0] == case][0]
    g = torch.Generator().manual_seed(len(name))
    dev = torch.device('cuda')
    x = torch.randn((B, H, W, Cin), generator=g)
    k = torch.randn((K, K, Cin, Cout), generator=g) * (2.0 / (K * K * Cin)) ** 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    if pad is None:
        oh, pt = C.same_pad(H, K, stride)
        ow, pl = C.same_pad(W, K, stride)
    else:
        pt, pl = pad
        oh, ow = out_hw if out_hw else (H, W)
    xin = C.FMap.empty(B, H, W, Cin, torch.float32, dev)
    xin.split = True
    xin.write(x)
    out = C.FMap.empty(B, oh, ow, Cout, torch.float32, dev)
    out.split = Cout % 32 == 0
    rmap = None
    if resmode == 'same':
        rm = C.FMap.empty(B, oh, ow, Cout, torch.float32, dev)
        rm.split = True
        rm.write(torch.randn((B, oh, ow, Cout), generator=g))
        rmap = [rm]
    w = C.pack_weight(k.numpy(), 'bf16x3', dev)
    ws = torch.empty((32 << 20,), dtype=torch.uint8, device=dev)
    nk = K * K * (Cin // 32)
    for split_k in (1, 3):
        if nk < 4 * split_k:
            continue
        base = None
        for tile in X3_TILES + X3_PIPE_TILES:
            d = C.conv_desc([xin], [out], w, bias.to(dev), K, K, Cin, Cout, stride=stride, pad=(pt, pl), relu=relu, residuals=rmap,
                            dtype='bf16x3', tile_hint=tile, workspace=ws, split_k=split_k)
            if -(-d.C_out // (tile % 1000)) * (tile % 1000) > d.weight_rows:
                continue
            out.buf.fill_(float('nan'))
            C.run_conv(d)
            got = out.buf.clone()
            assert torch.isfinite(out.read()).all()
            if base is None:
                base = got
            assert torch.equal(got.view(torch.int32), base.view(torch.int32)), (tile, split_k)


@pytest.mark.parametrize('cin,cout,relu', [(128, 384, True), (64, 896, True)])
def test_bf16x3_dual_shape_grid_on_pre_split_maps(cin, cout, relu):
    """ tile code 2256256 for GPP_BF16X3 (C_out = 256 k + 128, pre-split input): 256 x 256 tiles + 512 x 128 tiles in one grid, both
    running the three-phase pipelined loop -- the bits of the plain tile, over several feature maps with ragged row counts """
    g = torch.Generator().manual_seed(cout)
    dev = torch.device('cuda')
    B, shapes = 2, [(21, 29), (11, 15), (6, 8), (3, 4)]
    total = sum(h * w for h, w in shapes)
    xbuf = torch.empty((B, total, cin), dtype=torch.float32, device=dev)
    w = C.pack_weight((torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5).numpy(), 'bf16x3', dev)
    b = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    ins, off = [], 0
    for h, wd in shapes:
        fm = C.FMap(xbuf, B, h, wd, cin, off=off * cin, bstride=total * cin, split=True)
        fm.write(torch.randn((B, h, wd, cin), generator=g))
        ins.append(fm)
        off += h * wd
    results = []
    for tile in (128128, 2256256):
        o = torch.full((B, total, cout), float('nan'), dtype=torch.float32, device=dev)
        outs, off = [], 0
        for h, wd in shapes:
            outs.append(C.FMap(o, B, h, wd, cout, off=off * cout, bstride=total * cout, split=True))
            off += h * wd
        C.run_conv(C.conv_desc(ins, outs, w, b, 3, 3, cin, cout, pad=(1, 1), relu=relu, dtype='bf16x3', tile_hint=tile))
        results.append(o.view(torch.int32).cpu())
        assert torch.isfinite(torch.cat([fm.read().reshape(-1) for fm in outs])).all()
    assert torch.equal(results[0], results[1])
    # float32 input maps: the dual grid (a pipelined form) is refused
    plain = [C.FMap(xbuf, B, h, wd, cin, off=fm.off, bstride=total * cin) for fm, (h, wd) in zip(ins, shapes)]
    d = C.conv_desc(plain, outs, w, b, 3, 3, cin, cout, pad=(1, 1), relu=relu, dtype='bf16x3', tile_hint=2256256)
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -4
