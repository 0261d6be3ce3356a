""" The narrow head-output layers (pyramid_regression_ops 512 -> 144, pyramid_classification 256 -> 96, pyramid_regression_dim 128 -> 36; 3 x 3 over
the five pyramid levels, float32 outputs) in GPP_F16X3 on pre-split input maps: every tile the library offers, us and bytes against tile 0.
    python tools/bench_head_tiles.py [B=8] [iters=30]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
from keras_retinanet_3D.backend import hip  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402

PYR = [(51, 167), (26, 84), (13, 42), (7, 21), (4, 11)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device('cuda')
    total = sum(h * w for h, w in PYR)
    g = torch.Generator().manual_seed(0)
    for name, cin, cout in (('pyramid_regression_ops', 512, 144), ('pyramid_classification', 256, 96), ('pyramid_regression_dim', 128, 36)):
        x = torch.empty((B, total, cin), dtype=torch.float32, device=dev)
        o = torch.empty((B, total, cout), dtype=torch.float32, device=dev)
        k = torch.randn((3, 3, cin, cout), generator=g) * 0.02
        w, scale, bias = C.pack_weight(k.numpy(), 'f16x3', dev), C.out_scale_of(k.numpy(), dev), torch.zeros((cout,), device=dev)
        ins, outs, off = [], [], 0
        for h, wd in PYR:
            fm = C.FMap(x, B, h, wd, cin, off=off * cin, bstride=total * cin, split=True, half='f16x3')
            fm.write(torch.relu(torch.randn((B, h, wd, cin), generator=g)))
            ins.append(fm)
            outs.append(C.FMap(o, B, h, wd, cout, off=off * cout, bstride=total * cout))
            off += h * wd
        d = C.conv_desc(ins, outs, w, bias, 3, 3, cin, cout, pad=(1, 1), dtype='f16x3', out_f32=True, out_scale=scale)
        tiles, count = (ctypes.c_int * 64)(), ctypes.c_int(0)
        hip.check(hip.lib().gpp_conv2d_tile_candidates(ctypes.byref(d), tiles, 64, ctypes.byref(count)), 'candidates')
        flops = 2.0 * B * total * 9 * cin * cout
        ref = None
        rows = []
        for tile in list(tiles[:count.value]):
            d.tile_hint = tile
            if hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) != 0:
                continue
            torch.cuda.synchronize()
            same = True if ref is None else bool(torch.equal(o.view(torch.int32), ref.view(torch.int32)))
            if ref is None:
                ref = o.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                C.run_conv(d)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1000.0 / iters
            rows.append((us, tile, same))
        rows.sort()
        print('{} ({} -> {}), B = {}: {:.1f} GFLOP'.format(name, cin, cout, B, flops / 1e9))
        for us, tile, same in rows:
            print('   tile {:8d}  {:7.1f} us  {:6.1f} TFLOP/s  bytes equal to tile 0: {}'.format(tile, us, flops / us / 1e6, same))


if __name__ == '__main__':
    main()
