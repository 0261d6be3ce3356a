""" One-off randomized sweep (not part of the suite) of the two fused kernels of round 6 against the launches they replace, byte for byte:
gpp_bottleneck_block (identity and projection blocks, C = 64 / 128, both x3 types, both tile forms) and gpp_stem_pool_fused_x3 (incl. the range count).
    python tools/fuzz_fused_kernels.py [cases = 120] [seed = 0] """
import ctypes
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
from keras_retinanet_3D.backend import hip  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402
import test_block_gpu as TB  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda')
bad = 0
for case in range(cases):
    cmid = rnd.choice((64, 128))
    dtype = rnd.choice(('f16x3', 'bf16x3'))
    B, H, W = rnd.randint(1, 3), rnd.randint(1, 70), rnd.randint(1, 90)
    ident = rnd.random() < 0.6
    if ident:
        blk = TB.make_block(B, H, W, cmid, dtype, seed=case)
    else:
        stride = rnd.choice((1, 2))
        cin = rnd.choice((64, 128, 256, 512))
        blk = TB.make_block(B, H, W, cmid, dtype, stride=stride, cin=cin, seed=case)
    Ho, Wo = blk['Ho'], blk['Wo']
    y_sep, y_fused = blk['split_map'](Ho, Wo, 4 * cmid), blk['split_map'](Ho, Wo, 4 * cmid)
    for d in blk['descs'](y_sep):
        C.run_conv(d)
    want = y_sep.buf.clone()
    f = blk['descs'](y_fused)
    for tile in ((0, 1814) if ident else (0,)):
        y_fused.buf.fill_(float('nan'))
        rc = TB.run_block(*f, tile)
        torch.cuda.synchronize()
        same = rc == 0 and torch.equal(y_fused.buf.view(torch.int32), want.view(torch.int32))
        if not same:
            bad += 1
            print('BLOCK MISMATCH', dict(cmid=cmid, dtype=dtype, B=B, H=H, W=W, ident=ident, tile=tile, rc=rc))
print('gpp_bottleneck_block: %d cases, %d mismatches' % (cases, bad))

lib = hip.lib()
bad_s = 0
for case in range(cases):
    B, H, W = rnd.randint(1, 3), rnd.randint(1, 300), rnd.randint(1, 520)
    big = rnd.random() < 0.4
    g = torch.Generator().manual_seed(case)
    x = torch.rand((B, H, W, 3), generator=g) * 255.0 - 120.0
    k = torch.randn((7, 7, 3, 64), generator=g) * (40.0 if big else 0.05)
    bias = torch.randn((64,), generator=g)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Ho + 1) // 2, (Wo + 1) // 2
    xd, bd = x.to(dev).contiguous(), bias.to(dev)
    packed = hip.pack_stem_weights_x3(k.reshape(147, 64).numpy(), dev)
    conv = torch.empty((B, Ho, Wo, 64), device=dev)
    want = torch.full((B, Hp, Wp, 64), float('nan'), device=dev)
    got = torch.full((B, Hp, Wp, 64), float('nan'), device=dev)
    slots = torch.zeros((2,), dtype=torch.int64, device=dev)
    hip.check(lib.gpp_stem_conv7x7_bn_relu_x3_rc(hip.ptr(xd), hip.ptr(packed), hip.ptr(bd), hip.ptr(conv), B, H, W, slots.data_ptr(), hip.stream_ptr()))
    hip.check(lib.gpp_maxpool3x3s2_same(hip.ptr(conv), hip.ptr(want), hip.GPP_F32, B, Ho, Wo, 64, hip.stream_ptr()))
    hip.check(lib.gpp_stem_pool_fused_x3(hip.ptr(xd), hip.ptr(packed), hip.ptr(bd), hip.ptr(got), B, H, W, slots.data_ptr() + 8, hip.stream_ptr()))
    torch.cuda.synchronize()
    c = slots.cpu().tolist()
    if not torch.equal(got.view(torch.int32), want.view(torch.int32)) or c[0] != c[1]:
        bad_s += 1
        print('STEM MISMATCH', dict(B=B, H=H, W=W, big=big, counts=c))
print('gpp_stem_pool_fused_x3: %d cases, %d mismatches' % (cases, bad_s))
sys.exit(1 if bad or bad_s else 0)
