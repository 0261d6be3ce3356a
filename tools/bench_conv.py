""" Micro-benchmark of gpp_conv2d_igemm on the shapes that dominate the network. """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))

import torch  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402

PYR = [(51, 167), (26, 84), (13, 42), (7, 21), (4, 11)]


def bench(name, B, shapes, Cin, Cout, K, dtype='bf16', iters=20, out_f32=False, tile=0, diag=0, residual=False):
    dev = torch.device('cuda')
    tdt = C.torch_dtype(dtype)
    total = sum(h * w for h, w in shapes)
    x = (torch.randn((B, total, Cin), device=dev) * 0.5).to(tdt)
    o = torch.empty((B, total, Cout), device=dev, dtype=torch.float32 if out_f32 else tdt)
    w = C.pack_weight((torch.randn((K, K, Cin, Cout)) * 0.02).numpy(), dtype, dev)
    bias = torch.zeros((Cout,), device=dev)
    ins, outs, off = [], [], 0
    for h, wd in shapes:
        ins.append(C.FMap(x, B, h, wd, Cin, off=off * Cin, bstride=total * Cin))
        outs.append(C.FMap(o, B, h, wd, Cout, off=off * Cout, bstride=total * Cout))
        off += h * wd
    res = None
    if residual:
        r = (torch.randn((B, total, Cout), device=dev) * 0.5).to(tdt)
        res, off = [], 0
        for h, wd in shapes:
            res.append(C.FMap(r, B, h, wd, Cout, off=off * Cout, bstride=total * Cout))
            off += h * wd
    d = C.conv_desc(ins, outs, w, bias, K, K, Cin, Cout, pad=(K // 2, K // 2), relu=True, dtype=dtype, out_f32=out_f32, tile_hint=tile, diag=diag,
                    residuals=res)
    for _ in range(3):
        C.run_conv(d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        C.run_conv(d)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = C.conv_flops(d)
    nbytes = B * total * (Cin + Cout * (2 if residual else 1)) * 2 + K * K * Cin * Cout * 2
    print('{:28s} {:8.3f} ms  {:8.1f} TFLOP/s  ({:.1f} GFLOP)  {:6.2f} TB/s algorithmic'.format(name, ms, fl / ms / 1e9, fl / 1e9, nbytes / ms / 1e9))
    return ms


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    if len(sys.argv) > 2 and sys.argv[2] == 'stages':
        for name, shp, cin, cout, k in (('res4 2a 1x1 1024->256', [(26, 84)], 1024, 256, 1), ('res4 2b 3x3 256->256', [(26, 84)], 256, 256, 3),
                                         ('res4 2c 1x1 256->1024', [(26, 84)], 256, 1024, 1), ('res3 2b 3x3 128->128', [(51, 167)], 128, 128, 3),
                                         ('res5 2b 3x3 512->512', [(13, 42)], 512, 512, 3), ('res5 2c 1x1 512->2048', [(13, 42)], 512, 2048, 1),
                                         ('P4 3x3 512->512', [(26, 84)], 512, 512, 3), ('cls 3x3 256->256', PYR, 256, 256, 3)):
            for tile in (128, 64, 96, 160, 192):
                bench('%s t%d' % (name, tile), B, shp, cin, cout, k, tile=tile)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'newtiles':
        rem = (128160, 128160, 128160, 128160, 192160, 1192160)
        deep = ()
        for name, shp, cin, cout, k, f32, res, tiles in (
                ('reg_ops 3x3 512->144 f32', PYR, 512, 144, 3, True, False, (192128, 1192128, 128128) + rem[3:]),
                ('cls_out 3x3 256->96 f32', PYR, 256, 96, 3, True, False, (192128, 128128, 96128) + rem[:3]),
                ('res4 2a 1x1 1024->256', [(26, 84)], 1024, 256, 1, False, False, (96128, 64128, 128128) + deep),
                ('res4 2b 3x3 256->256', [(26, 84)], 256, 256, 3, False, False, (96128, 64128, 128128) + deep),
                ('res4 2c 1x1 256->1024 +res', [(26, 84)], 256, 1024, 1, False, True, (64128, 96128, 128128) + deep),
                ('res5 2a 1x1 2048->512', [(13, 42)], 2048, 512, 1, False, False, (64128, 96128) + deep),
                ('res5 2b 3x3 512->512', [(13, 42)], 512, 512, 3, False, False, (128128, 64128) + deep),
                ('res5 2c 1x1 512->2048 +res', [(13, 42)], 512, 2048, 1, False, True, (160128, 64128) + deep),
                ('res3 2a 1x1 512->128', [(51, 167)], 512, 128, 1, False, False, (160128, 64128) + deep)):
            for tile in tiles:
                bench('%s t%d' % (name, tile), B, shp, cin, cout, k, tile=tile, out_f32=f32, residual=res, iters=30)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'stamps':
        # needs a library built with -DGPP_STAMPS (GPP_LIB=...): where does the per-tile time go?
        import numpy as np
        dev = torch.device('cuda')
        shapes = (('reg 3x3 512->512', PYR, 512, 512, 3, 256256, False),
                  ('C3_reduced 1x1 512->512', PYR[:1], 512, 512, 1, 1192256, False),
                  ('cls 3x3 256->256', PYR, 256, 256, 3, 1192256, False),
                  ('res4 2b 3x3 256->256', [(26, 84)], 256, 256, 3, 96128, False))
        if len(sys.argv) > 3 and sys.argv[3] == 'small':      # the 1x1 layers of res3 / res4 / res5
            shapes = (('res4 2a 1x1 1024->256', [(26, 84)], 1024, 256, 1, 96128, False),
                      ('res4 2c 1x1 256->1024', [(26, 84)], 256, 1024, 1, 64128, False),
                      ('res5 2a 1x1 2048->512', [(13, 42)], 2048, 512, 1, 64128, False),
                      ('res5 2c 1x1 512->2048', [(13, 42)], 512, 2048, 1, 160128, False),
                      ('res3 2a 1x1 512->128', [(51, 167)], 512, 128, 1, 160128, False))
        for name, shp, cin, cout, k, tile, f32 in shapes:
            tdt = C.torch_dtype('bf16')
            total = sum(h * w for h, w in shp)
            x = (torch.randn((B, total, cin), device=dev) * 0.5).to(tdt)
            o = torch.empty((B, total, cout), device=dev, dtype=torch.float32 if f32 else tdt)
            w = C.pack_weight((torch.randn((k, k, cin, cout)) * 0.02).numpy(), 'bf16', dev)
            bias = torch.zeros((cout,), device=dev)
            ins, outs, off = [], [], 0
            for h, wd in shp:
                ins.append(C.FMap(x, B, h, wd, cin, off=off * cin, bstride=total * cin))
                outs.append(C.FMap(o, B, h, wd, cout, off=off * cout, bstride=total * cout))
                off += h * wd
            d = C.conv_desc(ins, outs, w, bias, k, k, cin, cout, pad=(k // 2, k // 2), relu=True, dtype='bf16', out_f32=f32, tile_hint=tile, diag=16 + 32)
            stamps = torch.zeros(((1 << 16) + 1024, 8), dtype=torch.int64, device=dev)
            d.zero_page = stamps.data_ptr()
            for _ in range(5):
                C.run_conv(d)
            torch.cuda.synchronize()
            stamps.zero_()
            C.run_conv(d)
            torch.cuda.synchronize()
            st = stamps.cpu().numpy()
            steps = st[(1 << 16):].reshape(-1)[:64 * 128].reshape(64, 128).astype(np.float64) * 0.01
            st = st[:(1 << 16)]
            if (steps[:, 1] > 0).any():
                live = steps[steps[:, 1] > 0]
                n = int((live[0] > 0).sum())
                dd = np.diff(live[:, :n], axis=1)
                print('   per-K-step time (us), mean over %d workgroups, steps 0..%d:' % (len(live), n - 2))
                print('   ' + ' '.join('%.2f' % v for v in dd.mean(axis=0)))
            st = st[st[:, 0] != 0][:, :5].astype(np.float64) * 0.01          # us
            t0 = st[:, 0].min()
            dur = np.diff(st, axis=1)
            print('%s tile %d: %d workgroups, kernel span %.1f us' % (name, tile, len(st), st[:, 4].max() - t0))
            print('   mean us  setup %.2f  main loop %.2f  epilogue issue %.2f  store drain %.2f   (whole workgroup %.2f)' %
                  (dur[:, 0].mean(), dur[:, 1].mean(), dur[:, 2].mean(), dur[:, 3].mean(), (st[:, 4] - st[:, 0]).mean()))
            order = np.argsort(st[:, 0])
            starts = st[order, 0] - t0
            ends = st[order, 4] - t0
            print('   workgroup starts (us), percentiles 0/25/50/75/100: ' + ' '.join('%.1f' % v for v in np.percentile(starts, [0, 25, 50, 75, 100])))
            print('   workgroup ends   (us), percentiles 0/25/50/75/100: ' + ' '.join('%.1f' % v for v in np.percentile(ends, [0, 25, 50, 75, 100])))
            first = starts < 1.0
            print('   first wave of workgroups: %d started within 1 us; their mean setup/main/epi/drain: %s' %
                  (first.sum(), ' '.join('%.2f' % v for v in dur[order][first].mean(axis=0))))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'towers0':
        for rep in range(2):
            for tile in (256256, 2256256):
                bench('towers_0 3x3 512->896 t%d' % tile, B, PYR, 512, 896, 3, tile=tile, iters=20)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'prio':
        for rep in range(3):
            for diag, what in ((0, 'base'),):
                bench('reg 3x3 512 t256256 %s' % what, B, PYR, 512, 512, 3, tile=256256, diag=diag, iters=30)
            for diag, what in ((0, 'base'),):
                bench('cls 3x3 256 t1192256 %s' % what, B, PYR, 256, 256, 3, tile=1192256, diag=diag, iters=30)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'stamps_tail':
        # phase timeline of the fused bottleneck tail (library built with -DGPP_STAMPS)
        import ctypes
        import numpy as np
        from keras_retinanet_3D.backend import hip
        dev = torch.device('cuda')
        tdt = C.torch_dtype('bf16')
        for cmid, H, W in ((64, 101, 334), (128, 51, 167)):
            cout = 4 * cmid
            a = C.FMap((torch.randn((B, H, W, cmid), device=dev) * 0.5).to(tdt), B, H, W, cmid)
            mid = C.FMap.empty(B, H, W, cmid, tdt, dev)
            y = C.FMap.empty(B, H, W, cout, tdt, dev)
            sc = C.FMap((torch.randn((B, H, W, cout), device=dev) * 0.5).to(tdt), B, H, W, cout)
            w1 = C.pack_weight((torch.randn((3, 3, cmid, cmid)) * 0.05).numpy(), 'bf16', dev)
            w2 = C.pack_weight((torch.randn((1, 1, cmid, cout)) * 0.1).numpy(), 'bf16', dev)
            b1, b2 = torch.zeros((cmid,), device=dev), torch.zeros((cout,), device=dev)
            for rows in (96, 128, 160):
                d1 = C.conv_desc([a], [mid], w1, b1, 3, 3, cmid, cmid, pad=(1, 1), relu=True, dtype='bf16', diag=16)
                d2 = C.conv_desc([mid], [y], w2, b2, 1, 1, cmid, cout, relu=True, residuals=[sc], dtype='bf16')
                stamps = torch.zeros((1 << 16, 8), dtype=torch.int64, device=dev)
                d1.zero_page = stamps.data_ptr()
                run = lambda: hip.check(hip.lib().gpp_bottleneck_tail(ctypes.byref(d1), ctypes.byref(d2), rows, hip.stream_ptr()), 'tail')  # noqa: E731
                for _ in range(5):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run()
                e1.record()
                torch.cuda.synchronize()
                stamps.zero_()
                run()
                torch.cuda.synchronize()
                st = stamps.cpu().numpy()
                st = st[st[:, 0] != 0][:, :5].astype(np.float64) * 0.01
                dur = np.diff(st, axis=1)
                t0 = st[:, 0].min()
                print('tail C=%d %dx%d rows %d: %.1f us per launch (20 back to back); %d workgroups, span %.1f us' %
                      (cmid, H, W, rows, e0.elapsed_time(e1) * 50.0, len(st), st[:, 4].max() - t0))
                print('   mean us  phase 1 (3x3) %.2f  hand-over %.2f  phase 2 (1x1 + residual + stores) %.2f  drain %.2f  whole %.2f' %
                      (dur[:, 0].mean(), dur[:, 1].mean(), dur[:, 2].mean(), dur[:, 3].mean(), (st[:, 4] - st[:, 0]).mean()))
                print('   starts pct 0/25/50/75/100: ' + ' '.join('%.1f' % v for v in np.percentile(st[:, 0] - t0, [0, 25, 50, 75, 100])))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'mid':
        # the latency-bound middle of the backbone, one launch shape each (for rocprofv3 --pmc)
        bench('res4 2a 1x1 1024->256 t96128', B, [(26, 84)], 1024, 256, 1, tile=96128, iters=20)
        bench('res4 2b 3x3 256->256 t96128', B, [(26, 84)], 256, 256, 3, tile=96128, iters=20)
        bench('res5 2b 3x3 512->512 t64128', B, [(13, 42)], 512, 512, 3, tile=64128, iters=20)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'x3time':
        # the big head / FPN layers of the x3 types on pre-split maps, one tile each (A/B of library builds: GPP_LIB=...)
        dev = torch.device('cuda')
        dtype = sys.argv[3] if len(sys.argv) > 3 else 'f16x3'
        reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
        layers = []
        for name, shp, cin, cout, k, tile, f32 in (('reg 3x3 512->512', PYR, 512, 512, 3, 1256256, False), ('towers_0 3x3 512->896', PYR, 512, 896, 3, 2256256, False),
                                                   ('cls 3x3 256->256', PYR, 256, 256, 3, 1192256, False), ('P3 3x3 512->512', PYR[:1], 512, 512, 3, 1192256, False),
                                                   ('reg_ops 3x3 512->144', PYR, 512, 144, 3, 1128160, True), ('dim 3x3 128->128', PYR, 128, 128, 3, 1192128, False),
                                                   ('cls_out 3x3 256->96 t192x128', PYR, 256, 96, 3, 1192128, True), ('cls_out 3x3 256->96 t192x96', PYR, 256, 96, 3, 1192096, True),
                                                   # the mixed-height grids (round 4) beside the uniform ones above, same run, same box
                                                   ('reg mix 256+224', PYR, 512, 512, 3, 3256224, False), ('reg 192x256', PYR, 512, 512, 3, 1192256, False),
                                                   ('P3 mix 192+160', PYR[:1], 512, 512, 3, 3192160, False), ('P3 256x256', PYR[:1], 512, 512, 3, 1256256, False)):
            total = sum(h * w for h, w in shp)
            wk = (torch.randn((k, k, cin, cout)) * 0.02).numpy()
            w = C.pack_weight(wk, dtype, dev)
            sc = C.out_scale_of(wk, dev) if dtype == 'f16x3' else None
            bias = torch.zeros((cout,), device=dev)
            ib = torch.empty((B, total, cin), device=dev); ob = torch.empty((B, total, cout), device=dev)
            ins, outs, off = [], [], 0
            for h, wd in shp:
                ins.append(C.FMap(ib, B, h, wd, cin, off=off * cin, bstride=total * cin, split=True, half=dtype))
                outs.append(C.FMap(ob, B, h, wd, cout, off=off * cout, bstride=total * cout, split=not f32, half=dtype))
                ins[-1].write(torch.randn((B, h, wd, cin), device=dev) * 0.5 * float(os.environ.get('X3TIME_ASCALE', '1')))      # (0: what the data costs)
                off += h * wd
            d = C.conv_desc(ins, outs, w, bias, k, k, cin, cout, pad=(k // 2, k // 2), relu=not f32, dtype=dtype, tile_hint=tile, out_scale=sc, out_f32=f32)
            layers.append((name, tile, d, C.conv_flops(d), (ib, ob, w, bias, sc)))
        best = {}
        for rep in range(reps + 1):
            for name, tile, d, fl, _ in layers:
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
        tot = 0.0
        for name, tile, d, fl, _ in layers:
            v = sorted(best[name])
            med = v[len(v) // 2]
            tot += med
            print('%-24s %s tile %7d: median %7.1f us  min %7.1f  (%.0f TFLOP/s of float32 products)' % (name, dtype, tile, med, v[0], fl / med / 1e6))
        print('sum of medians %.1f us' % tot)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'x3stamps':
        # phases of the pipelined x3 loop (library built with -DGPP_STAMPS): per K-step, for one wavefront of the first 64 workgroups,
        # phase A (hi * wlo), phase B (hi * whi), wait + barrier, phase C (lo * whi + the LDS-DMA of the stage after next)
        import numpy as np
        dev = torch.device('cuda')
        dtype = sys.argv[3] if len(sys.argv) > 3 else 'f16x3'
        # optional 4th argument: comma-separated ablation masks of the diagnostic build (64 no LDS-DMA, 128 no LDS reads in phase C,
        # 2048 none in phases A / B, 4096 no static priority; sums combine them) -- timing only
        masks = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else [0]
        for name, shp, cin, cout, k, tile in (('reg 3x3 512->512', PYR, 512, 512, 3, 1192256), ('cls 3x3 256->256', PYR, 256, 256, 3, 1192256)):
            total = sum(h * w for h, w in shp)
            x = torch.randn((B, total, cin), device=dev) * 0.5
            wk = (torch.randn((k, k, cin, cout)) * 0.02).numpy()
            w = C.pack_weight(wk, dtype, dev)
            sc = C.out_scale_of(wk, dev) if dtype == 'f16x3' else None
            bias = torch.zeros((cout,), device=dev)
            ib = torch.empty((B, total, cin), device=dev); ob = torch.empty((B, total, cout), device=dev)
            ins, outs, off = [], [], 0
            for h, wd in shp:
                ins.append(C.FMap(ib, B, h, wd, cin, off=off * cin, bstride=total * cin, split=True, half=dtype))
                outs.append(C.FMap(ob, B, h, wd, cout, off=off * cout, bstride=total * cout, split=True, half=dtype))
                ins[-1].write(x[:, off:off + h * wd].reshape(B, h, wd, cin))
                off += h * wd
            for mask, wv in [(m, v) for m in masks for v in ((0, 1, 4, 5) if len(masks) > 1 else range(8))]:
                d = C.conv_desc(ins, outs, w, bias, k, k, cin, cout, pad=(k // 2, k // 2), relu=True, dtype=dtype, tile_hint=tile, diag=32 + (wv << 8) + mask, out_scale=sc)
                stamps = torch.zeros(((1 << 16) + 1024, 8), dtype=torch.int64, device=dev)
                d.zero_page = stamps.data_ptr()
                for _ in range(3):
                    C.run_conv(d)
                torch.cuda.synchronize()
                stamps.zero_()
                C.run_conv(d)
                torch.cuda.synchronize()
                st = stamps.cpu().numpy()[(1 << 16):].reshape(-1)[:64 * 8].reshape(64, 8).astype(np.float64)
                st = st[st[:, 4] > 0]
                per = st[:, :4] * 0.01 / st[:, 4:5]
                m = per.mean(axis=0)
                print('%-18s %s tile %d ablation %4d wavefront %d: per K-step %.3f us = A %.3f + B %.3f + wait, barrier %.3f + C %.3f   (%d workgroups, %d K-steps)' %
                      (name, dtype, tile, mask, wv, m.sum(), m[0], m[1], m[2], m[3], len(st), int(st[0, 4])))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'kstamps':
        # inside a K-step of the plain (non-pipelined) loop, library built with -DGPP_STAMPS
        import numpy as np
        dev = torch.device('cuda')
        for name, shp, cin, cout, k, tile in (('res4 2a 1x1 1024->256', [(26, 84)], 1024, 256, 1, 96128), ('res4 2b 3x3 256->256', [(26, 84)], 256, 256, 3, 96128),
                                              ('res5 2b 3x3 512->512', [(13, 42)], 512, 512, 3, 64128), ('res5 2a 1x1 2048->512', [(13, 42)], 2048, 512, 1, 64128)):
            tdt = C.torch_dtype('bf16')
            total = sum(h * w for h, w in shp)
            x = (torch.randn((B, total, cin), device=dev) * 0.5).to(tdt)
            o = torch.empty((B, total, cout), device=dev, dtype=tdt)
            w = C.pack_weight((torch.randn((k, k, cin, cout)) * 0.02).numpy(), 'bf16', dev)
            bias = torch.zeros((cout,), device=dev)
            ins = [C.FMap(x, B, shp[0][0], shp[0][1], cin)]
            outs = [C.FMap(o, B, shp[0][0], shp[0][1], cout)]
            d = C.conv_desc(ins, outs, w, bias, k, k, cin, cout, pad=(k // 2, k // 2), relu=True, dtype='bf16', tile_hint=tile, diag=16 + 32)
            stamps = torch.zeros(((1 << 16) + 1024, 8), dtype=torch.int64, device=dev)
            d.zero_page = stamps.data_ptr()
            for _ in range(20):
                C.run_conv(d)
            torch.cuda.synchronize()
            stamps.zero_()
            for _ in range(3):
                C.run_conv(d)                       # the stamps of the last of three back-to-back launches remain
            torch.cuda.synchronize()
            st = stamps.cpu().numpy()[(1 << 16):].reshape(-1)[:64 * 128].reshape(64, 32, 4).astype(np.float64) * 0.01
            live = st[st[:, 1, 0] > 0]
            nsteps = min(30, (live[0, :, 0] > 0).sum())
            seg = live[:, 1:nsteps - 1, :]                       # skip the first step
            nxt = live[:, 2:nsteps, 0]
            wait = (seg[:, :, 1] - seg[:, :, 0]).mean()
            bar = (seg[:, :, 2] - seg[:, :, 1]).mean()
            dma = (seg[:, :, 3] - seg[:, :, 2]).mean()
            mma = (nxt - seg[:, :, 3]).mean()
            print('%-24s tile %d: per K-step %.2f us = wait for the tile %.2f + barrier %.2f + LDS-DMA issue %.2f + LDS reads and MFMA %.2f  (%d workgroups, %d steps)' %
                  (name, tile, wait + bar + dma + mma, wait, bar, dma, mma, len(live), nsteps - 2))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'autotune':
        # what gpp_conv2d_autotune picks (tile, split-K, us) for the under-filled deep-K layers
        import ctypes
        from keras_retinanet_3D.backend import hip
        dev = torch.device('cuda')
        ws = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
        for name, shp, cin, cout, k in (('P4 3x3 512->512', [(26, 84)], 512, 512, 3), ('P5 3x3 512->512', [(13, 42)], 512, 512, 3),
                                        ('res5 2b 3x3 512->512', [(13, 42)], 512, 512, 3), ('C5_reduced 1x1 2048->512', [(13, 42)], 2048, 512, 1),
                                        ('res4 2b 3x3 256->256', [(26, 84)], 256, 256, 3), ('C4_reduced 1x1 1024->512', [(26, 84)], 1024, 512, 1)):
            tdt = C.torch_dtype('bf16')
            h, wd = shp[0]
            x = (torch.randn((B, h * wd, cin), device=dev) * 0.5).to(tdt)
            o = torch.empty((B, h * wd, cout), device=dev, dtype=tdt)
            w = C.pack_weight((torch.randn((k, k, cin, cout)) * 0.02).numpy(), 'bf16', dev)
            bias = torch.zeros((cout,), device=dev)
            d = C.conv_desc([C.FMap(x, B, h, wd, cin)], [C.FMap(o, B, h, wd, cout)], w, bias, k, k, cin, cout, pad=(k // 2, k // 2), relu=True,
                            dtype='bf16', workspace=ws, split_k=0)
            best = ctypes.c_float(0.0)
            hip.check(hip.lib().gpp_conv2d_autotune(ctypes.byref(d), 16, hip.stream_ptr(), ctypes.byref(best)), 'autotune')
            print('%-28s -> tile %8d split %2d  %.1f us' % (name, d.tile_hint, d.split_k, best.value))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'cold':
        # how much of a small layer's in-network time is cold caches?  hot = back to back; cold = after a 1 GiB fill
        # (L2 and Infinity Cache flushed); weights-warm = after the fill, the weights are read once (-> Infinity Cache);
        # all-warm = after the fill, weights and input are read once
        dev = torch.device('cuda')
        flush = torch.empty((256 << 20,), dtype=torch.float32, device=dev)
        for name, shp, cin, cout, k, tile, res in (('res5 2a 1x1 2048->512', [(13, 42)], 2048, 512, 1, 64128, False), ('res5 2b 3x3 512->512', [(13, 42)], 512, 512, 3, 64128, False),
                                                   ('res5 2c 1x1 512->2048 +res', [(13, 42)], 512, 2048, 1, 160128, True), ('res4 2a 1x1 1024->256', [(26, 84)], 1024, 256, 1, 96128, False),
                                                   ('res4 2b 3x3 256->256', [(26, 84)], 256, 256, 3, 96128, False), ('res4 2c 1x1 256->1024 +res', [(26, 84)], 256, 1024, 1, 64128, True),
                                                   ('C5_reduced 1x1 2048->512', [(13, 42)], 2048, 512, 1, 64128, False), ('res3 2a 1x1 512->128', [(51, 167)], 512, 128, 1, 160128, False)):
            tdt = C.torch_dtype('bf16')
            h, wd = shp[0]
            x = (torch.randn((B, h * wd, cin), device=dev) * 0.5).to(tdt)
            o = torch.empty((B, h * wd, cout), device=dev, dtype=tdt)
            r = (torch.randn((B, h * wd, cout), device=dev) * 0.5).to(tdt) if res else None
            w = C.pack_weight((torch.randn((k, k, cin, cout)) * 0.02).numpy(), 'bf16', dev)
            bias = torch.zeros((cout,), device=dev)
            d = C.conv_desc([C.FMap(x, B, h, wd, cin)], [C.FMap(o, B, h, wd, cout)], w, bias, k, k, cin, cout, pad=(k // 2, k // 2), relu=True,
                            dtype='bf16', tile_hint=tile, residuals=[C.FMap(r, B, h, wd, cout)] if res else None)
            out = []
            for mode in ('hot', 'cold', 'weights-warm', 'all-warm'):
                ts = []
                for it in range(8):
                    if mode != 'hot':
                        flush.fill_(float(it))
                    if mode in ('weights-warm', 'all-warm'):
                        w.view(torch.int16).sum()
                    if mode == 'all-warm':
                        x.view(torch.int16).sum()
                        if res:
                            r.view(torch.int16).sum()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    C.run_conv(d)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                out.append('%s %.1f' % (mode, sorted(ts)[len(ts) // 2]))
            print('%-30s tile %7d   us (median of 8, single launches): %s' % (name, tile, '   '.join(out)))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'big':
        for rep in range(2):
            for tile in (512, 128):
                bench('reg 3x3 512->512 t%d' % tile, B, PYR, 512, 512, 3, tile=tile)
                bench('towers_0 3x3 512->896 t%d' % tile, B, PYR, 512, 896, 3, tile=tile)
                bench('cls 3x3 256->256 t%d' % tile, B, PYR, 256, 256, 3, tile=tile)
                bench('C3_reduced 1x1 512->512 t%d' % tile, B, [(51, 167)], 512, 512, 1, tile=tile)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'hbm':
        for name, shp, cin, cout, k, res in (('res2 2a 1x1 256->64', [(101, 334)], 256, 64, 1, False), ('res2 2b 3x3 64->64', [(101, 334)], 64, 64, 3, False),
                                              ('res2 2c 1x1 64->256 +res', [(101, 334)], 64, 256, 1, True), ('res2 br1 1x1 64->256', [(101, 334)], 64, 256, 1, False),
                                              ('res3 2a 1x1 512->128', [(51, 167)], 512, 128, 1, False), ('res3 2b 3x3 128->128', [(51, 167)], 128, 128, 3, False),
                                              ('res3 2c 1x1 128->512 +res', [(51, 167)], 128, 512, 1, True)):
            for tile in (64, 128, 512):
                bench('%s t%d' % (name, tile), B, shp, cin, cout, k, tile=tile, residual=res)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'pipe':
        for rep in range(2):
            for tile in (128, 512):
                for diag, what in ((0, 'plain'), (4, 'pipelined')):
                    bench('reg 3x3 512 t%d %s' % (tile, what), B, PYR, 512, 512, 3, tile=tile, diag=diag)
            bench('cls 3x3 256 t128 plain', B, PYR, 256, 256, 3, tile=128, diag=0)
            bench('cls 3x3 256 t128 pipelined', B, PYR, 256, 256, 3, tile=128, diag=4)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == 'ablate':
        for tile in (128, 256, 512):
            for diag, what in ((0, 'full'), (1, 'no LDS-DMA (compute only)'), (2, 'no LDS reads/MFMA (loads only)'), (3, 'neither')):
                bench('reg 3x3 512 t%d %s' % (tile, what), B, PYR, 512, 512, 3, tile=tile, diag=diag)
        sys.exit(0)
    bench('reg tower 3x3 512->512 t128', B, PYR, 512, 512, 3, tile=128)
    bench('reg tower 3x3 512->512 t256', B, PYR, 512, 512, 3, tile=256)
    bench('reg tower 3x3 512->512 t512', B, PYR, 512, 512, 3, tile=512)
    bench('P3 3x3 512->512 t512', B, PYR[:1], 512, 512, 3, tile=512)
    bench('cls tower 3x3 256->256 t512', B, PYR, 256, 256, 3, tile=512)
    bench('cls tower 3x3 256->256 t256', B, PYR, 256, 256, 3, tile=256)
    bench('P3 3x3 512->512 t256', B, PYR[:1], 512, 512, 3, tile=256)
    bench('cls0 3x3 512->256 t256', B, PYR, 512, 256, 3, tile=256)
    bench('cls0 3x3 512->256 t128', B, PYR, 512, 256, 3, tile=128)
    bench('cls tower 3x3 256->256', B, PYR, 256, 256, 3)
    bench('dim tower 3x3 128->128', B, PYR, 128, 128, 3)
    bench('reg out 3x3 512->144 f32', B, PYR, 512, 144, 3, out_f32=True)
    bench('P3 3x3 512->512', B, PYR[:1], 512, 512, 3)
    bench('res2 1x1 256->64', B, [(101, 334)], 256, 64, 1)
    bench('res2 3x3 64->64', B, [(101, 334)], 64, 64, 3)
    bench('res2 1x1 64->256', B, [(101, 334)], 64, 256, 1)
    bench('res3 3x3 128->128', B, [(51, 167)], 128, 128, 3)
    bench('res4 3x3 256->256', B, [(26, 84)], 256, 256, 3)
    bench('res4 1x1 1024->256', B, [(26, 84)], 1024, 256, 1)
    bench('res5 3x3 512->512', B, [(13, 42)], 512, 512, 3)
    bench('res5 1x1 512->2048', B, [(13, 42)], 512, 2048, 1)
    bench('f16 reg tower 3x3', B, PYR, 512, 512, 3, dtype='f16')
    if len(sys.argv) > 2 and sys.argv[2] == 'cold_tiles':
        # the plain tiles on the latency-bound layers (round 2 compared the since-removed loader-wavefront form here).  Back to back on the
        # same buffers (inputs Infinity-Cache resident) AND with a 600 MB buffer rewritten between launches ("cold": inputs
        # come from HBM, closer to the network where another layer's output is what was written last)
        flush = torch.empty((600 << 20,), dtype=torch.uint8, device='cuda')

        def cold(name, shapes, cin, cout, k, tile, residual=False):
            dev = torch.device('cuda')
            total = sum(h * w for h, w in shapes)
            x = (torch.randn((B, total, cin), device=dev) * 0.5).to(torch.bfloat16)
            o = torch.empty((B, total, cout), device=dev, dtype=torch.bfloat16)
            w = C.pack_weight((torch.randn((k, k, cin, cout)) * 0.02).numpy(), 'bf16', dev)
            bias = torch.zeros((cout,), device=dev)
            ins = [C.FMap(x, B, shapes[0][0], shapes[0][1], cin)]
            outs = [C.FMap(o, B, shapes[0][0], shapes[0][1], cout)]
            res = [C.FMap((torch.randn((B, total, cout), device=dev) * 0.5).to(torch.bfloat16), B, shapes[0][0], shapes[0][1], cout)] if residual else None
            d = C.conv_desc(ins, outs, w, bias, k, k, cin, cout, pad=(k // 2, k // 2), relu=True, tile_hint=tile, residuals=res)
            C.run_conv(d)
            ts = []
            for _ in range(6):
                flush.fill_(1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                C.run_conv(d)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            return sorted(ts)[len(ts) // 2]

        for name, shp, cin, cout, k, res in (('res4 2a 1x1 1024->256', [(26, 84)], 1024, 256, 1, False), ('res4 2b 3x3 256->256', [(26, 84)], 256, 256, 3, False),
                                              ('res4 2c 1x1 256->1024 +res', [(26, 84)], 256, 1024, 1, True), ('res5 2a 1x1 2048->512', [(13, 42)], 2048, 512, 1, False),
                                              ('res5 2b 3x3 512->512', [(13, 42)], 512, 512, 3, False), ('res5 2c 1x1 512->2048 +res', [(13, 42)], 512, 2048, 1, True),
                                              ('P4 3x3 512->512', [(26, 84)], 512, 512, 3, False), ('C3_reduced 1x1 512->512', [(51, 167)], 512, 512, 1, False),
                                              ('res3 2a 1x1 512->128', [(51, 167)], 512, 128, 1, False)):
            for tile in (64128, 96128, 128128, 160128):
                hot = bench('%s tile %d' % (name, tile), B, shp, cin, cout, k, tile=tile, residual=res, iters=20)
                print('    cold (inputs from HBM): %.1f us   hot %.1f us' % (cold(name, shp, cin, cout, k, tile, res), hot * 1e3))
