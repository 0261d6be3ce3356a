#!/usr/bin/env python
"""
Shared horizontal taps (tile codes 6... / 7..., include/gpp.h GPP_X3_PADCOL) on the regression-tower layer: bits and time.

  * the layer on maps WITH a padding column (every image row one zero pixel wider), plain tiles against the shared-tap tiles: same bytes
  * the same layer on compact maps: the values at the real pixels are the same bytes too (the padding column IS the zero padding)
  * back-to-back launch times of every form, same run

    python tools/kws_check.py [batch] [f16x3|bf16x3] [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))

import torch  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402

PYR = [(51, 167), (26, 84), (13, 42), (7, 21), (4, 11)]


def build(B, shapes, cin, cout, dtype, wk, values, padded, tile, dev):
    """ one layer over the pyramid: maps of width W (+ 1 when padded), input = values (post-ReLU), output zero-initialised """
    wd = [(h, w + (1 if padded else 0)) for h, w in shapes]
    total = sum(h * w for h, w in wd)
    ib = torch.zeros((B, total, cin), device=dev)
    ob = torch.zeros((B, total, cout), device=dev)
    ins, outs, off = [], [], 0
    for (h, w), (_, w0), v in zip(wd, shapes, values):
        ins.append(C.FMap(ib, B, h, w, cin, off=off * cin, bstride=total * cin, split=True, half=dtype))
        outs.append(C.FMap(ob, B, h, w, cout, off=off * cout, bstride=total * cout, split=True, half=dtype))
        x = torch.zeros((B, h, w, cin), device=dev)
        x[:, :, :w0] = v
        ins[-1].write(x)
        off += h * w
    w = C.pack_weight(wk, dtype, dev)
    sc = C.out_scale_of(wk, dev) if dtype == 'f16x3' else None
    bias = torch.randn((cout,), device=dev) * 0.1
    torch.manual_seed(5)
    d = C.conv_desc(ins, outs, w, bias, 3, 3, cin, cout, pad=(1, 1), relu=True, dtype=dtype, tile_hint=tile, out_scale=sc, padcol=padded)
    return d, outs, (ib, ob, w, sc, bias)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dtype = sys.argv[2] if len(sys.argv) > 2 else 'f16x3'
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device('cuda')
    cin = cout = 512
    torch.manual_seed(1)
    wk = (torch.randn((3, 3, cin, cout)) * 0.02).numpy()
    values = [torch.relu(torch.randn((B, h, w, cin), device=dev) * 0.5) for h, w in PYR]
    forms = [('compact maps, 256 x 256', False, 1256256), ('compact maps, 256 + 224 mixed', False, 3256224),
             ('padding column, 256 x 256', True, 1256256), ('padding column, 256 + 224 mixed', True, 3256224),
             ('padding column, 192 x 256', True, 1192256),
             ('padding column, shared taps 256 x 256', True, 6256256), ('padding column, shared taps 224 x 256', True, 6224256),
             ('padding column, shared taps 192 x 256', True, 6192256), ('padding column, shared taps 256 + 224 mixed', True, 7256224)]
    layers = []
    for name, padded, tile in forms:
        torch.manual_seed(3)
        d, outs, keep = build(B, PYR, cin, cout, dtype, wk, values, padded, tile, dev)
        C.run_conv(d)
        torch.cuda.synchronize()
        layers.append((name, padded, tile, d, outs, keep))
    ref_c = [o.dense().clone() for o in layers[0][4]]
    ref_p = [o.dense().clone() for o in layers[2][4]]
    ok = True
    for name, padded, tile, d, outs, keep in layers:
        same = all(torch.equal(o.dense().view(torch.int32), r.view(torch.int32)) for o, r in zip(outs, ref_p if padded else ref_c))
        ok &= same
        print('%-48s tile %7d: %s' % (name, tile, 'same bytes as the plain 256 x 256 tile on the same maps' if same else 'DIFFERENT BYTES'))
    # the padding column is the zero padding: real pixels of the padded run == the compact run, padding pixels untouched (zero)
    for (h, w), p, c in zip(PYR, ref_p, ref_c):
        same = torch.equal(p[:, :, :w].contiguous().view(torch.int32), c.view(torch.int32)) and bool((p[:, :, w] == 0).all())
        ok &= same
        print('level %3d x %3d: real pixels of the padded maps == the compact maps, padding column still zero: %s' % (h, w, same))
    flops = 2.0 * B * sum(h * w for h, w in PYR) * 9 * cin * cout
    best = {}
    for rep in range(reps + 1):
        for name, padded, tile, d, outs, keep in layers:
            for _ in range(2):
                C.run_conv(d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                C.run_conv(d)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                best.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    for name, padded, tile, d, outs, keep in layers:
        v = sorted(best[name])
        med = v[len(v) // 2]
        print('%-48s tile %7d: median %7.1f us  min %7.1f  (%.0f TFLOP/s of the layer\'s float32 products)' % (name, tile, med, v[0], flops / med / 1e6))
    print('ALL SAME' if ok else 'MISMATCH')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
