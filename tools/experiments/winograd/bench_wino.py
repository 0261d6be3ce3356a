#!/usr/bin/env python
"""
The regression-tower layer (3 x 3, 512 -> 512 over the five pyramid levels of a 402 x 1333 frame, B = 8: 431.8 GFLOP direct) three ways, back
to back on one box, on post-ReLU data:  the direct f16x3 kernel at its tuned tile, and the Winograd F(2, 3) form (csrc/conv_wino_impl.h) as
its two launches -- the input transform (HBM-bound) and the four position GEMMs.   python tools/bench_wino.py [batch = 8] [reps = 20]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from keras_retinanet_3D.backend import hip  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SHAPES = [(51, 167), (26, 84), (13, 42), (7, 21), (4, 11)]
CIN = COUT = 512
DEV = 'cuda'


def pyramid(channels):
    total = sum(h * w for h, w in SHAPES)
    buf = torch.zeros((B, total, channels), dtype=torch.float32, device=DEV)
    maps, off = [], 0
    for h, w in SHAPES:
        maps.append(C.FMap(buf, B, h, w, channels, off=off * channels, bstride=total * channels, split=True, half='f16x3'))
        off += h * w
    return buf, maps


def timed(fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(reps))
    return t[len(t) // 2], t[0]


rng = np.random.default_rng(0)
_, src = pyramid(CIN)
_, dst = pyramid(COUT)
_, dst2 = pyramid(COUT)
for f in src:
    f.write(torch.relu(torch.randn((B, f.H, f.W, CIN), device=DEV)) * 2.0)
kernel = (rng.standard_normal((3, 3, CIN, COUT)) * np.sqrt(2.0 / (9 * CIN))).astype(np.float32)
bias = torch.zeros((COUT,), dtype=torch.float32, device=DEV)
w_direct, s_direct = C.pack_weight(kernel, 'f16x3', DEV), C.out_scale_of(kernel, DEV)        # (kept alive: the descriptor holds raw pointers)
d = C.conv_desc(src, dst, w_direct, bias, 3, 3, CIN, COUT, pad=(1, 1), relu=True, dtype='f16x3', out_scale=s_direct)
best = ctypes.c_float(0.0)
hip.check(hip.lib().gpp_conv2d_autotune(ctypes.byref(d), 6, hip.stream_ptr(), ctypes.byref(best)), 'autotune')
pairs = C.wino_pairs(src)
v = torch.zeros((B * pairs * 4 * CIN,), dtype=torch.float32, device=DEV)
w, scale = C.pack_weight_wino(kernel, DEV)
dt = C.wino_desc(src, None, B, CIN, COUT, v=v, transform=True)
dc = C.wino_desc(None, dst2, B, CIN, COUT, weight=w, bias=bias, out_scale=scale, relu=True, v=v)
lib = hip.lib()
flop = 2.0 * 9 * CIN * COUT * B * sum(h * w for h, w in SHAPES)
rows = []
for rnd in range(3):
    t_direct = timed(lambda: C.run_conv(d))
    t_tr = timed(lambda: hip.check(lib.gpp_wino_transform_f16x3(ctypes.byref(dt), hip.stream_ptr())))
    t_conv = timed(lambda: hip.check(lib.gpp_wino_conv3x3_f16x3(ctypes.byref(dc), hip.stream_ptr())))
    rows.append((t_direct, t_tr, t_conv))
errs = []
for lvl in (0, 2, 4):                                  # both outputs against the float64 convolution of the stored input (P3, P5, P7 levels)
    ref = torch.relu(torch.nn.functional.conv2d(src[lvl].read().double().permute(0, 3, 1, 2), torch.as_tensor(kernel).double().to(DEV).permute(3, 2, 0, 1), padding=1)).permute(0, 2, 3, 1)
    errs.append((float((dst[lvl].read().double() - ref).abs().max() / ref.abs().max()), float((dst2[lvl].read().double() - ref).abs().max() / ref.abs().max())))
err = errs
print('B = {}: {:.1f} GFLOP direct per launch, M = {} pixels / {} pairs; library {}'.format(B, flop / 1e9, B * sum(h * w for h, w in SHAPES), B * pairs, lib.gpp_version().decode()))
print('direct tile {}; max |error| / max against float64 on levels P3, P5, P7 (direct, Winograd): {}'.format(int(d.tile_hint), ', '.join('({:.1e}, {:.1e})'.format(a, b) for a, b in err)))
for t_direct, t_tr, t_conv in rows:
    print('direct {:7.1f} us (min {:7.1f}) = {:5.1f} TFLOP/s | transform {:6.1f} us = {:4.2f} TB/s | position GEMMs {:7.1f} us (min {:7.1f}) = {:5.1f} TFLOP/s direct-equivalent '
          '({:5.1f} executed) | transform + GEMMs {:7.1f} us = {:+.1f} % vs direct'.format(
              t_direct[0], t_direct[1], flop / t_direct[0] / 1e6, t_tr[0], (3.0 * B * sum(h * w for h, w in SHAPES) * CIN * 4) / t_tr[0] / 1e6,
              t_conv[0], t_conv[1], flop / t_conv[0] / 1e6, flop / 1.5 / t_conv[0] / 1e6, t_tr[0] + t_conv[0],
              100.0 * (t_tr[0] + t_conv[0] - t_direct[0]) / t_direct[0]))
