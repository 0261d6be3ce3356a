"""
GPU numerics of GPP_F16X3: float32 storage, every float32 product as three IEEE-half matrix products (x = hi + lo, hi = f16(x),
lo = f16(x - hi): 11 + 11 significant bits per operand, the lo * lo term dropped: ~2^-22 per product; float32: 2^-24).  This is
the throughput mode that has to stay inside the reference-precision tolerance of BASELINE.json (plane index exact, 3-D corners
within 1e-3 of the float32 path), so its bars are float32's:

    |err| <= 1e-5 |ref| + 2e-6 rms(ref) sqrt(K)  per element  (the bar of tests/test_conv_f32_gpu.py for GPP_F32), and
    rms(err) <= (3e-8 sqrt(K) + 1e-7) rms(ref)                (measured 1.0e-7 at K = 64 ... 1.0e-6 at K = 9216,
                                                               tools/x3_error_probe.py; GPP_BF16X3 sits at 4e-6 throughout)

against a float64 reference of the same op on the values the operands hold.  Also: the range handling (activations clamped at
+-65504, per-output-channel power-of-two weight scale + gpp_conv_desc.out_scale, subnormal lo halves -- the matrix pipe keeps
half subnormals, tools/micro/mfma_denorm.hip), pre-split maps in every flag combination, tile / split-K invariance, the
dual-shape grid, ReLU on a pre-split map, argument validation.
"""
import ctypes

import pytest
import torch

from keras_retinanet_3D.backend import hip
from keras_retinanet_3D.layers import conv as C
from test_conv_gpu import CASES
from test_conv_f32_gpu import reference64

pytestmark = pytest.mark.gpu

PLAIN_TILES = [64064, 96064, 128064, 160064, 192064, 64128, 96128, 128128, 160128, 192128, 224128, 128160, 128256, 192256, 256256]
PIPE_TILES = [1128128, 1192128, 1128256, 1160256, 1192256, 1224256, 1256256, 1128160, 1192096]
DEEP_TILES = [5064064, 5096064, 5064128, 5096128, 5128128]          # the plain loop on a four-deep LDS ring (pre-split inputs; round 5)


def held(t):
    """ what a pre-split GPP_F16X3 map holds of a value: hi + lo (about 22 significant bits, clamped to the half range) """
    t = t.clamp(-65504.0, 65504.0)
    hi = t.to(torch.float16).float()
    return hi + (t - hi).to(torch.float16).float()


def build(case, flags=0, seed_shift=0, x_scale=1.0, channel_spread=False):
    """ one layer of tests/test_conv_gpu.CASES in GPP_F16X3; flags = which maps are pre-split (GPP_X3_IN | OUT | RES) """
    name, B, H, W, Cin, Cout, K, stride, pad, out_hw, relu, resmode, _ = case
    g = torch.Generator().manual_seed(sum(map(ord, name)) + seed_shift)
    dev = torch.device('cuda')
    x = torch.randn((B, H, W, Cin), generator=g) * x_scale
    k = torch.randn((K, K, Cin, Cout), generator=g) * (2.0 / (K * K * Cin)) ** 0.5
    if channel_spread:                                    # output channels nine decades apart: the per-channel scale has to cope
        k = k * torch.pow(10.0, torch.linspace(-6.0, 3.0, Cout))[None, None, None, :]
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
    if res is None:
        flags &= 3
    x_held = held(x) if flags & 1 else x.clamp(-65504.0, 65504.0)
    res_held = None if res is None else (held(res) if flags & 4 else res)
    ref = reference64(x_held, k, bias, stride, pt, pl, oh, ow, relu, res_held)
    xin = C.FMap.empty(B, H, W, Cin, torch.float32, dev, split=bool(flags & 1), half='f16x3')
    xin.write(x)
    out = C.FMap.empty(B, oh, ow, Cout, torch.float32, dev, split=bool(flags & 2), half='f16x3')
    rmap = None
    if res is not None:
        rm = C.FMap.empty(B, res.shape[1], res.shape[2], Cout, torch.float32, dev, split=bool(flags & 4), half='f16x3')
        rm.write(res)
        rmap = [rm]
    w = C.pack_weight(k.numpy(), 'f16x3', dev)
    scale = C.out_scale_of(k.numpy(), dev)
    ws = torch.empty((32 << 20,), dtype=torch.uint8, device=dev)
    bias_d = bias.to(dev)

    def make(tile, split_k=1):
        return C.conv_desc([xin], [out], w, bias_d, K, K, Cin, Cout, stride=stride, pad=(pt, pl), relu=relu, residuals=rmap,
                           dtype='f16x3', tile_hint=tile, workspace=ws, split_k=split_k, out_scale=scale)
    return make, out, ref, K * K * Cin, (xin, w, scale, rmap, ws, bias_d), flags


def check(out, ref, kdepth, rms_factor=1.0):
    got = out.read().double().cpu()
    assert torch.isfinite(got).all()
    rms = float(ref.pow(2).mean().sqrt())
    err = (got - ref).abs()
    tol = 1e-5 * ref.abs() + 2e-6 * rms * kdepth ** 0.5                      # float32's own bar
    assert bool((err <= tol).all()), 'max err {} (rms {})'.format(err.max().item(), rms)
    assert float(err.pow(2).mean().sqrt()) <= rms_factor * (3e-8 * kdepth ** 0.5 + 1e-7) * rms


def fits(d, tile):
    bn = tile % 1000 if tile else 64
    return -(-d.C_out // bn) * bn <= d.weight_rows


@pytest.mark.parametrize('tile', [0, 64064, 128128, 192128, 128160, 256256])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_f16x3_matches_float64_like_float32_does(case, tile):
    make, out, ref, kdepth, _, _ = build(case)
    out.buf.fill_(float('nan'))
    d = make(tile)
    if not fits(d, tile):
        pytest.skip('tile grid would read past the packed weight rows')
    C.run_conv(d)
    check(out, ref, kdepth)


@pytest.mark.parametrize('flags', [1, 2, 4, 3, 7])
@pytest.mark.parametrize('tile', [0, 96128, 192128, 192160, 128256, 1256256, 1192128])
@pytest.mark.parametrize('case', ['3x3_wide', '1x1_res_up_nonint', 'bottleneck_2c', '3x3_s2_tfsame', 'deepK'])
def test_f16x3_pre_split_maps(case, tile, flags):
    c = [c for c in CASES if c[0] == case][0]
    if c[4] % 32 or ((flags & 6) and c[5] % 32):
        pytest.skip('pre-split maps hold whole 32-channel blocks')
    make, out, ref, kdepth, _, eff = build(c, flags=flags, seed_shift=flags)
    out.buf.fill_(float('nan'))
    d = make(tile, split_k=3 if case == 'deepK' else 1)
    assert d.x3_split == eff
    if not fits(d, tile):
        pytest.skip('tile grid would read past the packed weight rows')
    if (tile > 1000000 or tile == 192160) and not (eff & 1):
        assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -4      # these forms read pre-split rows only
        return
    C.run_conv(d)
    check(out, ref, kdepth, rms_factor=1.5 if eff & 2 else 1.0)                        # a pre-split OUTPUT is rounded to 22 bits once more


@pytest.mark.parametrize('case', ['3x3_wide', 'bottleneck_2c', '3x3_s2_tfsame', 'deepK', '1x1'])
def test_every_f16x3_tile_gives_identical_results(case):
    """ plain, 8-wavefront, software-pipelined and four-deep-ring tiles, float32 and pre-split input maps, with and without split-K: an output
    element is summed in the same order (hi * wlo, hi * whi, lo * whi per K-step) -> identical bits """
    c = [c for c in CASES if c[0] == case][0]
    for flags in (0, 7):
        make, out, _, kdepth, _, eff = build(c, flags=flags)
        nk = kdepth // 32
        for split_k in (1, 3):
            if nk < 4 * split_k:
                continue
            base = None
            for tile in PLAIN_TILES + ([192160] + PIPE_TILES + DEEP_TILES if eff & 1 else []):
                d = make(tile, split_k=split_k)
                if not fits(d, tile):
                    continue
                out.buf.fill_(float('nan'))
                C.run_conv(d)
                got = out.buf.clone()
                assert torch.isfinite(out.read()).all()
                if base is None:
                    base = got
                assert torch.equal(got.view(torch.int32), base.view(torch.int32)), (tile, split_k, flags)


@pytest.mark.parametrize('cin,cout,relu', [(128, 384, True), (64, 896, True)])
def test_f16x3_dual_shape_grid(cin, cout, relu):
    """ tile code 2256256 (C_out = 256 k + 128, pre-split input): the bits of the plain tile over several ragged feature maps;
    the out_scale pointer of the 128-column block is moved with its weights """
    g = torch.Generator().manual_seed(cout)
    dev = torch.device('cuda')
    B, shapes = 2, [(21, 29), (11, 15), (6, 8), (3, 4)]
    total = sum(h * w for h, w in shapes)
    xbuf = torch.empty((B, total, cin), dtype=torch.float32, device=dev)
    k = torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5
    k = k * torch.pow(2.0, torch.randint(-6, 7, (cout,), generator=g).float())[None, None, None, :]      # a different scale per channel
    w = C.pack_weight(k.numpy(), 'f16x3', dev)
    scale = C.out_scale_of(k.numpy(), dev)
    b = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    ins, off, xs = [], 0, []
    for h, wd in shapes:
        fm = C.FMap(xbuf, B, h, wd, cin, off=off * cin, bstride=total * cin, split=True, half='f16x3')
        xs.append(torch.randn((B, h, wd, cin), generator=g))
        fm.write(xs[-1])
        ins.append(fm)
        off += h * wd
    results = []
    for tile in (128128, 2256256):
        o = torch.full((B, total, cout), float('nan'), dtype=torch.float32, device=dev)
        outs, off = [], 0
        for h, wd in shapes:
            outs.append(C.FMap(o, B, h, wd, cout, off=off * cout, bstride=total * cout, split=True, half='f16x3'))
            off += h * wd
        C.run_conv(C.conv_desc(ins, outs, w, b, 3, 3, cin, cout, pad=(1, 1), relu=relu, dtype='f16x3', tile_hint=tile, out_scale=scale))
        results.append(o.view(torch.int32).cpu())
        for fm, x in zip(outs, xs):
            ref = reference64(held(x), k, b.cpu(), 1, 1, 1, fm.H, fm.W, relu, None)
            got = fm.read().double().cpu()
            rms = float(ref.pow(2).mean().sqrt())
            assert float((got - ref).pow(2).mean().sqrt()) <= 1.5 * (3e-8 * (9 * cin) ** 0.5 + 1e-7) * rms
    assert torch.equal(results[0], results[1])


@pytest.mark.parametrize('dtype', ['f16x3', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('cout', [96, 180])
def test_the_96_column_tile_on_the_logits_shape(dtype, cout):
    """ tile code 1192096 (192 x 96, 4 x 1 wavefronts, the pipelined loop): float32 outputs of 96 / 180 channels over ragged pyramid
    levels -- the bits of the plain 128 x 128 tile (and of the pipelined 192 x 128 one) """
    g = torch.Generator().manual_seed(cout)
    dev = torch.device('cuda')
    B, cin, shapes = 2, 128, [(25, 31), (13, 16), (7, 8), (4, 4), (2, 2)]
    total = sum(h * w for h, w in shapes)
    x3 = dtype in C.X3_TYPES
    tdt = torch.float32 if x3 else C.torch_dtype(dtype)
    xbuf = torch.empty((B, total, cin), dtype=tdt, device=dev)
    k = torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5
    w = C.pack_weight(k.numpy(), dtype, dev)
    scale = C.out_scale_of(k.numpy(), dev) if dtype == 'f16x3' else None
    b = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    ins, off = [], 0
    for h, wd in shapes:
        fm = C.FMap(xbuf, B, h, wd, cin, off=off * cin, bstride=total * cin, split=x3, half=dtype if x3 else 'bf16x3')
        fm.write(torch.randn((B, h, wd, cin), generator=g))
        ins.append(fm)
        off += h * wd
    tiles, count = (ctypes.c_int * 32)(), ctypes.c_int(0)
    o_probe = torch.empty((B, total, cout), dtype=torch.float32, device=dev)
    outs_probe, off = [], 0
    for h, wd in shapes:
        outs_probe.append(C.FMap(o_probe, B, h, wd, cout, off=off * cout, bstride=total * cout))
        off += h * wd
    probe = C.conv_desc(ins, outs_probe, w, b, 3, 3, cin, cout, pad=(1, 1), dtype=dtype, out_f32=True, out_scale=scale)      # (the library validates what it lists tiles for)
    hip.check(hip.lib().gpp_conv2d_tile_candidates(ctypes.byref(probe), tiles, 32, ctypes.byref(count)), 'gpp_conv2d_tile_candidates')
    assert 1192096 in list(tiles[:count.value])                      # offered to the autotuner where it cuts the N padding
    results = []
    for tile in (128128, 1192128, 1192096):
        o = torch.full((B, total, cout), float('nan'), dtype=torch.float32, device=dev)
        outs, off = [], 0
        for h, wd in shapes:
            outs.append(C.FMap(o, B, h, wd, cout, off=off * cout, bstride=total * cout))
            off += h * wd
        C.run_conv(C.conv_desc(ins, outs, w, b, 3, 3, cin, cout, pad=(1, 1), relu=False, dtype=dtype, tile_hint=tile, out_f32=True, out_scale=scale))
        assert torch.isfinite(o).all()
        results.append(o.view(torch.int32).cpu())
    assert torch.equal(results[0], results[1]) and torch.equal(results[0], results[2])


def test_f16x3_range_activations_beyond_the_half_range_are_clamped_not_inf():
    c = [c for c in CASES if c[0] == '3x3'][0]
    for flags in (0, 1):
        make, out, ref, kdepth, keep, _ = build(c, flags=flags, x_scale=1.0)
        xin = keep[0]
        x = xin.read().clone()
        x[0, 3, 4, 5] = 1.0e6                              # -> 65504
        x[1, 2, 2, 7] = -3.0e38                            # -> -65504
        xin.write(x)
        k_ref = reference64(x.cpu().clamp(-65504.0, 65504.0), *_case_tensors(c)[1:])
        out.buf.fill_(float('nan'))
        C.run_conv(make(128128))
        got = out.read().double().cpu()
        assert torch.isfinite(got).all()
        rms = float(k_ref.pow(2).mean().sqrt())
        assert float((got - k_ref).abs().max()) <= 1e-5 * float(k_ref.abs().max()) + 1e-5 * rms


def _case_tensors(case):
    """ the tensors build() draws for a case (same generator order), for references on modified inputs """
    name, B, H, W, Cin, Cout, K, stride, pad, out_hw, relu, resmode, _ = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    x = torch.randn((B, H, W, Cin), generator=g)
    k = torch.randn((K, K, Cin, Cout), generator=g) * (2.0 / (K * K * Cin)) ** 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    pt, pl = pad
    oh, ow = out_hw if out_hw else (H, W)
    return x, k, bias, stride, pt, pl, oh, ow, relu, None


@pytest.mark.parametrize('x_scale', [1e-2, 1e-4, 3e-6])
def test_f16x3_tiny_activations_keep_their_low_half(x_scale):
    """ |x| < 2^-3: lo = f16(x - hi) is a SUBNORMAL half (the matrix pipe keeps them); |x| < 6e-5: hi is subnormal too.  The absolute
    error stays at the subnormal spacing 2^-24 per operand, so a map of tiny values loses RELATIVE precision only below ~1e-3:
    bar rms(err) <= (2^-24 / x_scale + 3e-7) * sqrt(K) * rms(w) * x_scale-ish, stated as a fraction of rms(ref) """
    c = [c for c in CASES if c[0] == '3x3_wide'][0]
    for flags in (0, 1):
        make, out, ref, kdepth, _, _ = build(c, flags=flags, x_scale=x_scale)
        out.buf.fill_(float('nan'))
        C.run_conv(make(128128))
        got = out.read().double().cpu()
        rms = float(ref.pow(2).mean().sqrt())
        rel = float((got - ref).pow(2).mean().sqrt()) / rms
        assert rel <= 3e-6 + 2.0 ** -24 / x_scale, (x_scale, flags, rel)


def test_f16x3_output_channels_nine_decades_apart():
    """ weights 1e-6 ... 1e3 across the output channels of one layer: every channel keeps float32-like RELATIVE accuracy because its
    packed weights carry their own power of two (without it the small channels' lo halves would be subnormal or zero) """
    c = [c for c in CASES if c[0] == '3x3_wide'][0]
    c = c[:10] + (False, None, c[12])                    # no ReLU, no shortcut: the channel's own scale sets the output's
    make, out, ref, kdepth, _, _ = build(c, channel_spread=True)
    out.buf.fill_(float('nan'))
    C.run_conv(make(128128))
    got = out.read().double().cpu()
    bias_free = ref.abs().reshape(-1, ref.shape[-1])
    per_channel_rms = bias_free.pow(2).mean(dim=0).sqrt()
    err = (got - ref).reshape(-1, ref.shape[-1]).pow(2).mean(dim=0).sqrt()
    # (the 0.1-sized bias dominates the small channels' outputs; its float32 addition bounds their error at ~6e-9 absolute)
    assert bool((err <= 2e-6 * per_channel_rms + 1e-8).all()), (err / per_channel_rms).max().item()


def test_relu_on_a_pre_split_f16x3_map():
    dev = torch.device('cuda')
    x = torch.randn((2, 5, 7, 64))
    src = C.FMap.empty(2, 5, 7, 64, torch.float32, dev, split=True, half='f16x3')
    dst = C.FMap.empty(2, 5, 7, 64, torch.float32, dev, split=True, half='f16x3')
    src.write(x)
    hip.check(hip.lib().gpp_relu(hip.ptr(src.buf), hip.ptr(dst.buf), hip.GPP_F16X3, x.numel(), hip.stream_ptr()))
    assert torch.equal(dst.read().cpu(), torch.relu(held(x)))


def test_f16x3_arguments_are_validated():
    c = [c for c in CASES if c[0] == '3x3_wide'][0]
    make, _, _, _, keep, _ = build(c)
    d = make(0)
    d.dtype = hip.GPP_BF16X3                                   # out_scale belongs to GPP_F16X3 alone
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -1
    d = make(0)
    d.x3_split = 8
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -1
    d = make(1128128)                                          # pipelined loop on a float32 input map
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == -4
    d = make(0)
    d.out_scale = None                                         # allowed: all ones (weights packed without a scale)
    assert hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) == 0


@pytest.mark.parametrize('tile_rows', [0, 64, 96, 128, 160])
@pytest.mark.parametrize('dtype', ['f16x3', 'bf16x3'])
@pytest.mark.parametrize('B,H,W', [(2, 25, 31), (1, 7, 5), (3, 9, 40), (1, 101, 67)])
def test_x3_bottleneck_tail_equals_the_two_layers(B, H, W, dtype, tile_rows):
    """ gpp_bottleneck_tail for the x3 types on pre-split maps (bottleneck_tail_x3_kernel: 3x3 64 -> 64 + 1x1 64 -> 256 + shortcut +
    ReLU in one launch, the intermediate tile kept in LDS as [32 hi | 32 lo] rows) reproduces the two separate launches bit for
    bit, and never touches the intermediate map; the unfused pair is checked against float64 here too. """
    cmid, cout = 64, 256
    g = torch.Generator().manual_seed(H * W + B)
    dev = torch.device('cuda')
    a = torch.randn((B, H, W, cmid), generator=g)
    k1 = torch.randn((3, 3, cmid, cmid), generator=g) * (2.0 / (9 * cmid)) ** 0.5
    k1 = k1 * torch.pow(2.0, torch.randint(-3, 4, (cmid,), generator=g).float())[None, None, None, :]      # a different weight scale per channel
    k2 = torch.randn((1, 1, cmid, cout), generator=g) * (2.0 / cmid) ** 0.5
    b1, b2 = torch.randn((cmid,), generator=g) * 0.1, torch.randn((cout,), generator=g) * 0.1
    sc = torch.randn((B, H, W, cout), generator=g)

    def split_map(c, values=None):
        m = C.FMap.empty(B, H, W, c, torch.float32, dev, split=True, half=dtype)
        if values is not None:
            m.write(values)
        return m
    amap, mid, rmap = split_map(cmid, a), split_map(cmid), split_map(cout, sc)
    y_sep, y_fused = split_map(cout), split_map(cout)
    w1, w2 = C.pack_weight(k1.numpy(), dtype, dev), C.pack_weight(k2.numpy(), dtype, dev)
    s1 = C.out_scale_of(k1.numpy(), dev) if dtype == 'f16x3' else None
    s2 = C.out_scale_of(k2.numpy(), dev) if dtype == 'f16x3' else None
    b1d, b2d = b1.to(dev), b2.to(dev)
    d1 = C.conv_desc([amap], [mid], w1, b1d, 3, 3, cmid, cmid, pad=(1, 1), relu=True, dtype=dtype, out_scale=s1)
    d2 = C.conv_desc([mid], [y_sep], w2, b2d, 1, 1, cmid, cout, relu=True, residuals=[rmap], dtype=dtype, out_scale=s2)
    C.run_conv(d1)
    C.run_conv(d2)
    want = y_sep.buf.clone()
    mid_ref = reference64(amap.read().cpu(), k1, b1, 1, 1, 1, H, W, True, None)
    ref = reference64(mid_ref.float(), k2, b2, 1, 0, 0, H, W, True, rmap.read().cpu())
    got_sep = y_sep.read().double().cpu()
    rms = float(ref.pow(2).mean().sqrt())
    assert float((got_sep - ref).pow(2).mean().sqrt()) <= (1e-6 if dtype == 'f16x3' else 2e-5) * rms
    d2f = C.conv_desc([mid], [y_fused], w2, b2d, 1, 1, cmid, cout, relu=True, residuals=[rmap], dtype=dtype, out_scale=s2)
    mid.buf.fill_(float('nan'))                               # the fused launch must not depend on (or write) it
    y_fused.buf.fill_(float('nan'))
    hip.check(hip.lib().gpp_bottleneck_tail(ctypes.byref(d1), ctypes.byref(d2f), tile_rows, hip.stream_ptr()), 'gpp_bottleneck_tail')
    assert torch.equal(y_fused.buf.view(torch.int32), want.view(torch.int32))
    assert torch.isnan(mid.buf).all()


def test_x3_bottleneck_tail_needs_pre_split_maps_and_64_channels():
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(3)
    B, H, W = 1, 9, 11

    def layer_pair(cmid, split):
        cout = 4 * cmid
        mk = lambda c: C.FMap.empty(B, H, W, c, torch.float32, dev, split=split, half='f16x3')  # noqa: E731
        k1 = torch.randn((3, 3, cmid, cmid), generator=g) * 0.05
        k2 = torch.randn((1, 1, cmid, cout), generator=g) * 0.1
        keep = [C.pack_weight(k1.numpy(), 'f16x3', dev), C.pack_weight(k2.numpy(), 'f16x3', dev), torch.zeros((cout,), device=dev),
                C.out_scale_of(k1.numpy(), dev), C.out_scale_of(k2.numpy(), dev), mk(cmid), mk(cmid), mk(cout), mk(cout)]
        d1 = C.conv_desc([keep[5]], [keep[6]], keep[0], keep[2], 3, 3, cmid, cmid, pad=(1, 1), relu=True, dtype='f16x3', out_scale=keep[3])
        d2 = C.conv_desc([keep[6]], [keep[7]], keep[1], keep[2], 1, 1, cmid, cout, relu=True, residuals=[keep[8]], dtype='f16x3', out_scale=keep[4])
        return d1, d2, keep
    d1, d2, keep = layer_pair(64, False)                       # float32 maps: the fused form reads and writes pre-split rows only
    assert hip.lib().gpp_bottleneck_tail(ctypes.byref(d1), ctypes.byref(d2), 0, hip.stream_ptr()) == -4
    d1, d2, keep2 = layer_pair(128, True)                      # C = 128: one workgroup per CU, not built
    assert hip.lib().gpp_bottleneck_tail(ctypes.byref(d1), ctypes.byref(d2), 0, hip.stream_ptr()) == -4
    d1, d2, keep3 = layer_pair(64, True)
    assert hip.lib().gpp_bottleneck_tail(ctypes.byref(d1), ctypes.byref(d2), 72, hip.stream_ptr()) == -1
    assert hip.lib().gpp_bottleneck_tail(ctypes.byref(d1), ctypes.byref(d2), 0, hip.stream_ptr()) == 0


def _pyramid_layer(cin, cout, shapes, B, seed):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device('cuda')
    total = sum(h * w for h, w in shapes)
    xbuf = torch.empty((B, total, cin), dtype=torch.float32, device=dev)
    k = torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5
    w = C.pack_weight(k.numpy(), 'f16x3', dev)
    scale = C.out_scale_of(k.numpy(), dev)
    b = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    ins, off = [], 0
    for h, wd in shapes:
        fm = C.FMap(xbuf, B, h, wd, cin, off=off * cin, bstride=total * cin, split=True, half='f16x3')
        fm.write(torch.randn((B, h, wd, cin), generator=g))
        ins.append(fm)
        off += h * wd

    def run(tile):
        o = torch.full((B, total, cout), float('nan'), dtype=torch.float32, device=dev)
        outs, off = [], 0
        for h, wd in shapes:
            outs.append(C.FMap(o, B, h, wd, cout, off=off * cout, bstride=total * cout, split=True, half='f16x3'))
            off += h * wd
        d = C.conv_desc(ins, outs, w, b, 3, 3, cin, cout, pad=(1, 1), relu=True, dtype='f16x3', tile_hint=tile, out_scale=scale)
        rc = hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr())
        return rc, o, d
    return run


@pytest.mark.parametrize('cout,tile', [(256, 3256224), (256, 3192160), (512, 3256224), (512, 3192160)])
def test_mixed_height_grid_gives_the_bits_of_the_plain_tile(cout, tile):
    """ tile codes 3256224 / 3192160: whole rounds of 256- (192-) row tiles over the head of the first map, 224- (160-) row tiles over
    the rest of it and over the other maps, ONE grid (the short tiles stage 256 / 192 rows and compute on 224 / 160).  The first map is
    large enough for a whole round of 256 workgroups; the others are ragged.  Bits of the plain 128 x 128 tile and of 1256256. """
    shapes = [(187, 181), (31, 43), (13, 17), (5, 3)] if cout == 256 else [(131, 129), (31, 43), (13, 17), (5, 3)]
    run = _pyramid_layer(64, cout, shapes, 2, seed=cout + tile)
    rc0, base, d = run(128128)
    assert rc0 == 0 and torch.isfinite(base).all()
    tiles, count = (ctypes.c_int * 64)(), ctypes.c_int(0)
    hip.check(hip.lib().gpp_conv2d_tile_candidates(ctypes.byref(d), tiles, 64, ctypes.byref(count)), 'gpp_conv2d_tile_candidates')
    assert tile in list(tiles[:count.value])
    rc, got, _ = run(tile)
    assert rc == 0, rc
    assert torch.equal(got.view(torch.int32), base.view(torch.int32))
    rc, big, _ = run(1256256)
    assert rc == 0 and torch.equal(big.view(torch.int32), base.view(torch.int32))


def test_mixed_height_grid_declines_layers_it_cannot_help():
    """ no whole round of 256 workgroups fits the first map, or the mix costs as many rounds as the uniform grid: GPP_ERR_UNSUPPORTED,
    which the autotuner skips -- never a silently different grid """
    run = _pyramid_layer(64, 256, [(40, 50), (13, 17)], 2, seed=5)
    rc, _, _ = run(3256224)
    assert rc == -4


def test_f16x3_range_ledger_counts_what_the_half_range_alters_and_keeps_non_finite_values():
    """ gpp_x3_range_events: zero on ordinary data; an activation an epilogue has to clamp (finite beyond +-65504), an inf or a NaN is
    counted, and a non-finite value is still non-finite after the split (it is not laundered into +-65504) """
    lib = hip.lib()
    n = ctypes.c_uint64(0)
    c = [c for c in CASES if c[0] == '3x3'][0]
    c = c[:10] + (False, None, c[12])                                         # no ReLU (it would hide a NaN: max(NaN, 0) = 0), no shortcut
    for flags in (2, 3):                                                       # pre-split output; pre-split input too
        make, out, ref, kdepth, keep, _ = build(c, flags=flags)
        hip.check(lib.gpp_x3_range_events(ctypes.byref(n), 1))
        C.run_conv(make(128128))
        hip.check(lib.gpp_x3_range_events(ctypes.byref(n), 0))
        assert n.value == 0                                                    # nothing out of range: nothing counted
        bias = keep[5]
        saved = bias.clone()
        bias[3] = 1.0e9                                                        # every pixel of channel 3 overflows the half range
        bias[9] = float('nan')
        bias[17] = float('inf')
        C.run_conv(make(128128))
        hip.check(lib.gpp_x3_range_events(ctypes.byref(n), 1))
        got = out.read().cpu()
        pixels = got.shape[0] * got.shape[1] * got.shape[2]
        assert pixels <= n.value <= 3 * pixels                                 # events are per 8-channel group: channels 3 | 9 | 17 are three groups
        assert bool((got[..., 3] == 65504.0).all())                            # clamped, and counted
        assert bool(torch.isnan(got[..., 9]).all()) and bool((~torch.isfinite(got[..., 17])).all())
        ok = [ch for ch in range(got.shape[-1]) if ch not in (3, 9, 17)]
        assert torch.isfinite(got[..., ok]).all()
        bias.copy_(saved)
        hip.check(lib.gpp_x3_range_events(ctypes.byref(n), 0))
        assert n.value == 0                                                    # reset


WS_CASES = [  # B, H, W, C_in, C_out, shortcut, relu, float32 output
    (2, 37, 41, 256, 64, False, True, False),        # one n-tile: every workgroup its own M tiles
    (2, 29, 23, 128, 512, True, True, False),        # 8 n-tiles share an M sequence on one XCD; shortcut
    (3, 13, 17, 256, 1024, True, True, False),       # 16 n-tiles
    (2, 19, 21, 512, 128, False, True, False),       # K = 512: 128 KB of weights, only the 64-row form fits
    (1, 9, 7, 64, 256, False, False, False),         # two K-steps: fewer than the ring runs ahead
    (2, 13, 42, 512, 2048, True, True, False),       # 32 n-tiles
    (1, 5, 3, 128, 64, True, False, False),          # 15 rows: one ragged tile, 255 workgroups with nothing to do
    (2, 23, 31, 128, 192, False, False, True),       # float32 output (a head-like layer); 192 / 64 = 3 n-tiles do not divide 32: refused
    (2, 23, 31, 128, 128, False, False, True),       # float32 output, accepted
    (4, 101, 167, 256, 64, False, True, False),      # res2-like: 67 468 rows, 528 M tiles over 256 workgroups
]


@pytest.mark.parametrize('dtype', ['f16x3', 'bf16x3'])
@pytest.mark.parametrize('case', WS_CASES, ids=['{}x{}x{}_{}to{}{}{}{}'.format(c[0], c[1], c[2], c[3], c[4], '_sc' if c[5] else '', '_relu' if c[6] else '', '_f32' if c[7] else '') for c in WS_CASES])
def test_weight_stationary_1x1_gives_the_bits_of_the_tile_kernels(case, dtype):
    """ tile codes 4128064 / 4064064 / 4128128 / 4064128 (conv1x1_ws_kernel): the W n-tile resident in LDS, activation slabs streamed through a ring across
    tile boundaries, counted waits -- against the plain 128 x 128 tile on the same pre-split maps, bit for bit; layers it does not
    take (weights + ring beyond 160 KB, n-tile counts that do not divide 32) are refused, not approximated """
    B, H, W, cin, cout, res, relu, f32out = case
    g = torch.Generator().manual_seed(B * H * W + cin + cout)
    dev = torch.device('cuda')
    k = torch.randn((1, 1, cin, cout), generator=g) * (2.0 / cin) ** 0.5
    k = k * torch.pow(2.0, torch.randint(-3, 4, (cout,), generator=g).float())[None, None, None, :]
    w = C.pack_weight(k.numpy(), dtype, dev)
    sc = C.out_scale_of(k.numpy(), dev) if dtype == 'f16x3' else None
    bias = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    xin = C.FMap.empty(B, H, W, cin, torch.float32, dev, split=True, half=dtype)
    xin.write(torch.randn((B, H, W, cin), generator=g))
    rmap = None
    if res:
        rm = C.FMap.empty(B, H, W, cout, torch.float32, dev, split=True, half=dtype)
        rm.write(torch.randn((B, H, W, cout), generator=g))
        rmap = [rm]

    def run(tile):
        out = C.FMap.empty(B, H, W, cout, torch.float32, dev, split=not f32out, half=dtype)
        out.buf.fill_(float('nan'))
        d = C.conv_desc([xin], [out], w, bias, 1, 1, cin, cout, relu=relu, residuals=rmap, dtype=dtype, tile_hint=tile, out_scale=sc, out_f32=f32out)
        rc = hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr())
        torch.cuda.synchronize()
        return rc, out.buf.view(torch.int32).cpu()
    rc, base = run(128128)
    assert rc == 0
    for tile in (4128064, 4064064, 4128128, 4064128):
        bm, bn = (tile // 1000) % 1000, tile % 1000
        fits = (cin // 32) * bn * 128 + 4 * bm * 128 <= 160 * 1024
        divides = cout % bn == 0 and 32 % (cout // bn) == 0
        rc, got = run(tile)
        if fits and divides:
            assert rc == 0, (tile, rc)
            assert torch.equal(got, base), (tile, int((got != base).sum()))
        else:
            assert rc == -4, (tile, rc)                        # GPP_ERR_UNSUPPORTED


def test_weight_stationary_1x1_is_offered_where_it_applies():
    """ gpp_conv2d_tile_candidates lists the weight-stationary forms for a shallow 1 x 1 layer on pre-split maps, and for nothing else """
    dev = torch.device('cuda')

    def cands(cin, cout, ksz, split, stride=1):
        k = torch.randn((ksz, ksz, cin, cout)) * 0.05
        w = C.pack_weight(k.numpy(), 'f16x3', dev)
        x = C.FMap.empty(8, 51, 167, cin, torch.float32, dev, split=split, half='f16x3')
        o = C.FMap.empty(8, 51 if stride == 1 else 26, 167 if stride == 1 else 84, cout, torch.float32, dev, split=split, half='f16x3')
        d = C.conv_desc([x], [o], w, torch.zeros((cout,), device=dev), ksz, ksz, cin, cout, pad=(ksz // 2, ksz // 2), stride=stride, dtype='f16x3',
                        out_scale=C.out_scale_of(k.numpy(), dev))
        tiles, count = (ctypes.c_int * 64)(), ctypes.c_int(0)
        hip.check(hip.lib().gpp_conv2d_tile_candidates(ctypes.byref(d), tiles, 64, ctypes.byref(count)), 'candidates')
        return set(tiles[:count.value])
    assert {4128064, 4064064, 4128128, 4064128} <= cands(128, 512, 1, True)
    assert 4064064 in cands(512, 128, 1, True) and 4128064 not in cands(512, 128, 1, True)      # 128 KB of weights: the 64-row ring only
    ws = {4128064, 4064064, 4128128, 4064128}
    assert not (ws & cands(128, 512, 1, False))                                 # float32 maps
    assert not (ws & cands(128, 128, 3, True))                                  # 3 x 3
    assert not (ws & cands(1024, 256, 1, True))                                 # 256 KB of weights
    assert not (ws & cands(256, 128, 1, True, stride=2))                        # strided


def test_the_four_deep_ring_tiles_are_offered_to_small_grids_only():
    """ tile codes 5BBBNNN: candidates for a deep-K layer whose grid is about one workgroup per CU (every layer at batch 1), not for a big grid """
    dev = torch.device('cuda')

    def candidates(B, H, W, cin, cout, split=True):
        x = C.FMap.empty(B, H, W, cin, torch.float32, dev, split=split, half='f16x3')
        o = C.FMap.empty(B, H, W, cout, torch.float32, dev, split=True, half='f16x3')
        k = torch.zeros((3, 3, cin, cout)).numpy()
        w, s = C.pack_weight(k, 'f16x3', dev), C.out_scale_of(k, dev)
        d = C.conv_desc([x], [o], w, torch.zeros((cout,), device=dev), 3, 3, cin, cout, pad=(1, 1), dtype='f16x3', out_scale=s)
        tiles, count = (ctypes.c_int * 64)(), ctypes.c_int(0)
        hip.check(hip.lib().gpp_conv2d_tile_candidates(ctypes.byref(d), tiles, 64, ctypes.byref(count)), 'candidates')
        return set(tiles[:count.value])

    small = candidates(1, 26, 84, 256, 256)                    # res4 branch2b at batch 1: 35 x 2 tiles of 64 x 128
    assert {5064064, 5064128, 5096128, 5128128} <= small
    assert not (candidates(8, 51, 167, 128, 128) & set(DEEP_TILES))          # res3 branch2b at batch 8: hundreds of workgroups
    assert not (candidates(1, 26, 84, 256, 256, split=False) & set(DEEP_TILES))      # float32 input maps: no
