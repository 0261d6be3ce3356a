""" The x3 stem: gpp_stem_conv7x7_bn_relu_x3_rc + gpp_maxpool3x3s2_same(GPP_F32) against gpp_stem_pool_fused_x3 (one launch), hot, HIP events.
    python tools/bench_stem.py [B = 8] [H = 402] [W = 1333] """
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ground-plane-polling_amd'))
import torch
from keras_retinanet_3D.backend import hip

B, H, W = (int(v) for v in (sys.argv[1:4] + ['8', '402', '1333'][len(sys.argv) - 1:]))
dev = torch.device('cuda')
g = torch.Generator().manual_seed(1)
x = (torch.rand((B, H, W, 3), generator=g) * 255.0 - 120.0).to(dev)
k = torch.randn((7, 7, 3, 64), generator=g) * 0.05
bias = torch.randn((64,), generator=g).to(dev)
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
Hp, Wp = (Ho + 1) // 2, (Wo + 1) // 2
packed = hip.pack_stem_weights_x3(k.reshape(147, 64).numpy(), dev)
conv = torch.empty((B, Ho, Wo, 64), device=dev)
want = torch.empty((B, Hp, Wp, 64), device=dev)
got = torch.empty((B, Hp, Wp, 64), device=dev)
slot = torch.zeros((1,), dtype=torch.int64, device=dev)
lib = hip.lib()


def two():
    hip.check(lib.gpp_stem_conv7x7_bn_relu_x3_rc(hip.ptr(x), hip.ptr(packed), hip.ptr(bias), hip.ptr(conv), B, H, W, slot.data_ptr(), hip.stream_ptr()))
    hip.check(lib.gpp_maxpool3x3s2_same(hip.ptr(conv), hip.ptr(want), hip.GPP_F32, B, Ho, Wo, 64, hip.stream_ptr()))


def one():
    hip.check(lib.gpp_stem_pool_fused_x3(hip.ptr(x), hip.ptr(packed), hip.ptr(bias), hip.ptr(got), B, H, W, slot.data_ptr(), hip.stream_ptr()))


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(5):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / n)
    return min(best), sorted(best)[len(best) // 2]


two(); one(); torch.cuda.synchronize()
print('B = %d, %d x %d: bytes equal: %s' % (B, H, W, torch.equal(got.view(torch.int32), want.view(torch.int32))))
print('stem + pool, two launches: min %.1f us, median %.1f us' % timed(two))
print('fused (GPP_STEM_POOL_X3_ROWS=%s): min %.1f us, median %.1f us' % ((os.environ.get('GPP_STEM_POOL_X3_ROWS', '6'),) + timed(one)))
