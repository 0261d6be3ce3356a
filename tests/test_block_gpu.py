"""
gpp_bottleneck_block (csrc/conv_block_impl.h): a whole bottleneck -- branch2a 1x1, branch2b 3x3, branch2c 1x1 + shortcut + ReLU of
keras_resnet's bottleneck_2d (the graph /root/reference/keras_retinanet_3D/models/resnet.py:88-93 instantiates) -- in ONE launch on pre-split
x3 maps, both intermediate maps kept in LDS.  The bar is the strongest one there is: the bytes of the three separate launches
(gpp_conv2d_igemm x 3, themselves held to float64 here), on identity blocks (shortcut = the block's input), projection blocks (stride 2,
shortcut = another map), tiles that hang over every image edge, images smaller than a tile, and with the intermediate maps poisoned.
"""
import ctypes

import pytest
import torch

from keras_retinanet_3D.backend import hip
from keras_retinanet_3D.layers import conv as C
from test_conv_f32_gpu import reference64

pytestmark = pytest.mark.gpu


def make_block(B, H, W, cmid, dtype, stride=1, cin=None, seed=0, x_scale=1.0):
    """ the three layers of one bottleneck over fresh pre-split maps; cin = None: identity block (C_in = 4 C, shortcut = input) """
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(1000 * H + W + B + seed)
    cout = 4 * cmid
    identity = cin is None
    cin = cout if identity else cin
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = torch.randn((B, H, W, cin), generator=g) * x_scale
    ks = [torch.randn((1, 1, cin, cmid), generator=g) * (2.0 / cin) ** 0.5,
          torch.randn((3, 3, cmid, cmid), generator=g) * (2.0 / (9 * cmid)) ** 0.5,
          torch.randn((1, 1, cmid, cout), generator=g) * (1.0 / cmid) ** 0.5]
    ks = [k * torch.pow(2.0, torch.randint(-3, 4, (k.shape[3],), generator=g).float())[None, None, None, :] for k in ks]      # a weight scale per channel
    bs = [torch.randn((k.shape[3],), generator=g) * 0.1 for k in ks]

    def split_map(h, w, c, values=None):
        m = C.FMap.empty(B, h, w, c, torch.float32, dev, split=True, half=dtype)
        if values is not None:
            m.write(values)
        return m
    xin = split_map(H, W, cin, x)
    sc = xin if identity else split_map(Ho, Wo, cout, torch.randn((B, Ho, Wo, cout), generator=g))
    a, bmap = split_map(Ho, Wo, cmid), split_map(Ho, Wo, cmid)
    ws = [C.pack_weight(k.numpy(), dtype, dev) for k in ks]
    ss = [C.out_scale_of(k.numpy(), dev) if dtype == 'f16x3' else None for k in ks]
    bd = [b.to(dev) for b in bs]

    def descs(y):
        d1 = C.conv_desc([xin], [a], ws[0], bd[0], 1, 1, cin, cmid, stride=stride, relu=True, dtype=dtype, out_scale=ss[0])
        d2 = C.conv_desc([a], [bmap], ws[1], bd[1], 3, 3, cmid, cmid, pad=(1, 1), relu=True, dtype=dtype, out_scale=ss[1])
        d3 = C.conv_desc([bmap], [y], ws[2], bd[2], 1, 1, cmid, cout, relu=True, residuals=[sc], dtype=dtype, out_scale=ss[2])
        return d1, d2, d3
    keep = (ws, ss, bd)
    return dict(x=xin, sc=sc, a=a, b=bmap, descs=descs, split_map=split_map, ks=ks, bs=bs, Ho=Ho, Wo=Wo, cout=cout, stride=stride, keep=keep)


def run_block(d1, d2, d3, tile=0):
    return hip.lib().gpp_bottleneck_block(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(d3), tile, hip.stream_ptr())


SHAPES = [(2, 25, 31), (1, 7, 5), (3, 9, 40), (1, 51, 167), (2, 8, 14), (1, 17, 29), (1, 1, 1), (2, 16, 28)]


@pytest.mark.parametrize('cmid', [128, 64])
@pytest.mark.parametrize('dtype', ['f16x3', 'bf16x3'])
@pytest.mark.parametrize('B,H,W', SHAPES)
def test_identity_block_in_one_launch_gives_the_bytes_of_its_three_layers(B, H, W, dtype, cmid):
    """ C = 128 (res3: 8 wavefronts, one workgroup per CU) and C = 64 (res2: 4 wavefronts, two workgroups per CU) """
    blk = make_block(B, H, W, cmid, dtype)
    y_sep, y_fused = blk['split_map'](H, W, 4 * cmid), blk['split_map'](H, W, 4 * cmid)
    d1, d2, d3 = blk['descs'](y_sep)
    for d in (d1, d2, d3):
        C.run_conv(d)
    # the separate launches against float64 (so that "the same bytes" means the right bytes)
    a_ref = reference64(blk['x'].read().cpu(), blk['ks'][0], blk['bs'][0], 1, 0, 0, H, W, True, None)
    b_ref = reference64(blk['a'].read().cpu(), blk['ks'][1], blk['bs'][1], 1, 1, 1, H, W, True, None)
    y_ref = reference64(blk['b'].read().cpu(), blk['ks'][2], blk['bs'][2], 1, 0, 0, H, W, True, blk['x'].read().cpu())
    bar = 1e-6 if dtype == 'f16x3' else 2e-5
    for got, ref in ((blk['a'], a_ref), (blk['b'], b_ref), (y_sep, y_ref)):
        rms = float(ref.pow(2).mean().sqrt())
        assert float((got.read().double().cpu() - ref).pow(2).mean().sqrt()) <= bar * max(rms, 1e-3)
    want = y_sep.buf.clone()
    f1, f2, f3 = blk['descs'](y_fused)
    blk['a'].buf.fill_(float('nan'))                           # the fused launch neither reads nor writes the intermediate maps
    blk['b'].buf.fill_(float('nan'))
    y_fused.buf.fill_(float('nan'))
    hip.check(run_block(f1, f2, f3), 'gpp_bottleneck_block')
    assert torch.equal(y_fused.buf.view(torch.int32), want.view(torch.int32))
    assert torch.isnan(blk['a'].buf).all() and torch.isnan(blk['b'].buf).all()
    # ... and the general form (the shortcut read from its map instead of taken from the LDS ring) on the same block: tile code + 1000
    y_fused.buf.fill_(float('nan'))
    hip.check(run_block(f1, f2, f3, 1814), 'gpp_bottleneck_block (general form)')
    assert torch.equal(y_fused.buf.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize('dtype', ['f16x3', 'bf16x3'])
@pytest.mark.parametrize('B,H,W,stride,cin,cmid', [(2, 25, 31, 2, 256, 128), (1, 101, 67, 2, 256, 128), (2, 13, 20, 1, 64, 128), (1, 9, 9, 2, 512, 128),
                                                  (1, 30, 44, 1, 256, 128), (2, 25, 31, 1, 64, 64), (1, 101, 67, 1, 64, 64), (1, 17, 30, 2, 128, 64)])
def test_projection_block_with_a_strided_first_layer(B, H, W, stride, cin, cmid, dtype):
    """ block 0 of a stage: branch2a carries the stride, the shortcut is the map the projection launch wrote (res2a: stride 1, 64 -> 64 -> 256) """
    blk = make_block(B, H, W, cmid, dtype, stride=stride, cin=cin, seed=7)
    Ho, Wo = blk['Ho'], blk['Wo']
    y_sep, y_fused = blk['split_map'](Ho, Wo, 4 * cmid), blk['split_map'](Ho, Wo, 4 * cmid)
    for d in blk['descs'](y_sep):
        C.run_conv(d)
    want = y_sep.buf.clone()
    f1, f2, f3 = blk['descs'](y_fused)
    blk['a'].buf.fill_(float('nan'))
    blk['b'].buf.fill_(float('nan'))
    y_fused.buf.fill_(float('nan'))
    hip.check(run_block(f1, f2, f3), 'gpp_bottleneck_block')
    assert torch.equal(y_fused.buf.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize('cmid', [128, 64])
def test_block_counts_the_range_events_its_three_layers_count(cmid):
    """ GPP_F16X3: activations beyond the half range are clamped AND counted -- once per stored group, as by the separate launches (the
    recomputed halo of a tile is clamped too, and not counted twice) """
    B, H, W = 1, 20, 30
    blk = make_block(B, H, W, cmid, 'f16x3', x_scale=3.0e4)
    n = ctypes.c_uint64(0)
    y_sep, y_fused = blk['split_map'](H, W, 4 * cmid), blk['split_map'](H, W, 4 * cmid)
    hip.check(hip.lib().gpp_x3_range_events(ctypes.byref(n), 1), 'reset')
    for d in blk['descs'](y_sep):
        C.run_conv(d)
    hip.check(hip.lib().gpp_x3_range_events(ctypes.byref(n), 1), 'read')
    separate = int(n.value)
    assert separate > 0
    hip.check(run_block(*blk['descs'](y_fused)), 'gpp_bottleneck_block')
    hip.check(hip.lib().gpp_x3_range_events(ctypes.byref(n), 1), 'read')
    assert int(n.value) == separate
    assert torch.equal(y_fused.buf.view(torch.int32), y_sep.buf.view(torch.int32))


def test_block_arguments_are_validated():
    blk = make_block(1, 9, 11, 128, 'f16x3')
    y = blk['split_map'](9, 11, 512)
    d1, d2, d3 = blk['descs'](y)
    assert run_block(d1, d2, d3, 0) == 0
    assert run_block(d1, d2, d3, 999) == -1                    # not a tile code
    assert run_block(d2, d2, d3, 0) == -4                      # a 3x3 layer where the first 1x1 belongs
    assert run_block(d1, d2, d2, 0) == -4
    plain = C.FMap.empty(1, 9, 11, 512, torch.float32, torch.device('cuda'))
    d3p = C.conv_desc([blk['b']], [plain], blk['keep'][0][2], blk['keep'][2][2], 1, 1, 128, 512, relu=True, residuals=[blk['sc']], dtype='f16x3',
                      out_scale=blk['keep'][1][2])
    assert run_block(d1, d2, d3p, 0) == -4                     # a float32 output map: the fused form writes pre-split rows only
    assert hip.lib().gpp_bottleneck_block(None, ctypes.byref(d2), ctypes.byref(d3), 0, hip.stream_ptr()) == -1
