""" The HBM-bound 1x1 layers of the f16x3 plan in isolation (B = 8, 402 x 1333): every tile the library offers, hot (back to back) and
cold (600 MB rewritten in between), the algorithmic bytes (input map + shortcut + output map, weights once) and the bandwidth they
amount to -- beside what a plain elementwise kernel (torch: out = a + b, the same read / read / write mix) reaches on the same box.
    python tools/hbm_layers.py [dtype]          (on the GPU box) """
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
import torch  # noqa: E402
from keras_retinanet_3D.backend import hip  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
B = 8
dev = torch.device('cuda')
flush = torch.empty((600 << 20,), dtype=torch.uint8, device=dev)


def timeit(fn, cold):
    ts = []
    for _ in range(5):
        if cold:
            flush.fill_(1)
        n = 1 if cold else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


# the elementwise yardstick: read a, read b, write out (float32), sizes of the layers below
for mb in (70, 140, 276):
    n = mb * (1 << 20) // 4
    a, b = torch.randn((n,), device=dev), torch.randn((n,), device=dev)
    o = torch.empty_like(a)
    for cold in (False, True):
        us = timeit(lambda: torch.add(a, b, out=o), cold)
        print('elementwise add, 3 x {:3d} MB: {:6.1f} us {:5s} = {:.2f} TB/s'.format(mb, us, 'cold' if cold else 'hot', 3 * n * 4 / us / 1e6))
    o2 = torch.empty_like(a)
    us = timeit(lambda: o2.copy_(a), True)
    print('copy,            2 x {:3d} MB: {:6.1f} us cold  = {:.2f} TB/s'.format(mb, us, 2 * n * 4 / us / 1e6))
    del a, b, o, o2

LAYERS = [('res2x_branch2a 256->64', 101, 334, 256, 64, False), ('res2a_branch1 64->256', 101, 334, 64, 256, False),
          ('res3x_branch2a 512->128', 51, 167, 512, 128, False), ('res3x_branch2c 128->512 +sc', 51, 167, 128, 512, True),
          ('res4x_branch2c 256->1024 +sc', 26, 84, 256, 1024, True), ('C3_reduced 512->512 +sc', 51, 167, 512, 512, True)]
for name, h, w, cin, cout, res in LAYERS:
    g = torch.Generator().manual_seed(cin + cout)
    k = (torch.randn((1, 1, cin, cout), generator=g) * (2.0 / cin) ** 0.5)
    wt = C.pack_weight(k.numpy(), dtype, dev)
    sc = C.out_scale_of(k.numpy(), dev) if dtype == 'f16x3' else None
    bias = torch.zeros((cout,), device=dev)
    x3 = dtype in C.X3_TYPES
    tdt = C.torch_dtype(dtype)
    xin = C.FMap.empty(B, h, w, cin, tdt, dev, split=x3, half=dtype if x3 else 'bf16x3')
    xin.write(torch.randn((B, h, w, cin), generator=g).abs())
    out = C.FMap.empty(B, h, w, cout, tdt, dev, split=x3, half=dtype if x3 else 'bf16x3')
    rmap = None
    if res:
        rm = C.FMap.empty(B, h, w, cout, tdt, dev, split=x3, half=dtype if x3 else 'bf16x3')
        rm.write(torch.randn((B, h, w, cout), generator=g).abs())
        rmap = [rm]
    esz = C.elem_size(dtype)
    nbytes = B * h * w * (cin + cout * (2 if res else 1)) * esz + cin * cout * esz
    d = C.conv_desc([xin], [out], wt, bias, 1, 1, cin, cout, relu=True, residuals=rmap, dtype=dtype, out_scale=sc)
    tiles, count = (ctypes.c_int * 64)(), ctypes.c_int(0)
    hip.check(hip.lib().gpp_conv2d_tile_candidates(ctypes.byref(d), tiles, 64, ctypes.byref(count)), 'candidates')
    rows = []
    extra = [t for t in (64064, 96064, 128064, 160064, 192064) if t not in list(tiles[:count.value])]      # narrow tiles the tuner does not offer wide layers
    for tile in list(tiles[:count.value]) + extra:
        d.tile_hint = tile
        if hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) != 0:
            continue
        rows.append((timeit(lambda: C.run_conv(d), True), timeit(lambda: C.run_conv(d), False), tile))
    rows.sort()
    print('{:30s} M {:6d}  {:6.1f} MB algorithmic, {:5.1f} us at 6.3 TB/s'.format(name, B * h * w, nbytes / 1e6, nbytes / 6.3e6))
    for cold, hot, tile in rows[:4] + [r for r in rows[4:] if r[2] in extra]:
        print('      tile {:8d}: cold {:6.1f} us = {:.2f} TB/s   hot {:6.1f} us = {:.2f} TB/s'.format(tile, cold, nbytes / cold / 1e6, hot, nbytes / hot / 1e6))
