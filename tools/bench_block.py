"""
One bottleneck as three tuned launches against gpp_bottleneck_block (csrc/conv_block_impl.h), same maps, same box, alternating.

    python tools/bench_block.py [stage=3] [B=8] [dtype=f16x3] [iters=30] [tiles=0,...]

stage 2 / 3: res2 / res3 identity block at 402 x 1333 (101 x 334 x 256 / 51 x 167 x 512).  Prints us per block for the three launches
(each layer on its own fastest tile, gpp_conv2d_autotune), for the fused tail where it exists (res2) and for every block tile asked for,
hot (back to back) and cold (a 512 MB buffer written between launches: what a layer sees inside the step).
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'ground-plane-polling_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)

from keras_retinanet_3D.backend import hip  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402


def main():
    stage = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dtype = sys.argv[3] if len(sys.argv) > 3 else 'f16x3'
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    tiles = [int(t) for t in sys.argv[5].split(',')] if len(sys.argv) > 5 else [0]
    from test_block_gpu import make_block, run_block
    H, W, cmid = {2: (101, 334, 64), 3: (51, 167, 128), 4: (26, 84, 256)}[stage]
    torch.cuda.set_device(0)
    blk = make_block(B, H, W, cmid, dtype)
    y = blk['split_map'](H, W, 4 * cmid)
    y2 = blk['split_map'](H, W, 4 * cmid)
    d = blk['descs'](y)
    f = blk['descs'](y2)
    best = ctypes.c_float(0)
    per_layer = []
    for dd in d:
        hip.check(hip.lib().gpp_conv2d_autotune(ctypes.byref(dd), 16, hip.stream_ptr(), ctypes.byref(best)), 'autotune')
        per_layer.append((int(dd.tile_hint), round(float(best.value), 1)))
    flops = 2.0 * B * H * W * (4 * cmid * cmid + 9 * cmid * cmid + cmid * 4 * cmid)
    mb = B * H * W * 4 * cmid * 4 * 2 / 1e6
    print('stage res{} B={} {}x{} C={} {}: {:.1f} GFLOP, x in + y out {:.0f} MB; per-layer tiles / us (hot, alone): {}'.format(
        stage, B, H, W, cmid, dtype, flops / 1e9, mb, per_layer))
    trash = torch.empty((512 << 20,), dtype=torch.uint8, device='cuda')

    def separate():
        for dd in d:
            C.run_conv(dd)

    def timed(fn, cold):
        fn()
        torch.cuda.synchronize()
        tot = 0.0
        for _ in range(iters):
            if cold:
                trash.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            tot += e0.elapsed_time(e1)
        return tot * 1000.0 / iters

    variants = [('three launches', separate)]
    for t in tiles:
        rc = run_block(f[0], f[1], f[2], t)
        if rc != 0:
            print('block tile {}: rc {}'.format(t, rc))
            continue
        variants.append(('block tile {}'.format(t), (lambda t=t: hip.check(run_block(f[0], f[1], f[2], t), 'block'))))
    separate()
    torch.cuda.synchronize()
    for name, fn in variants[1:]:
        y2.buf.fill_(float('nan'))
        fn()
        torch.cuda.synchronize()
        same = torch.equal(y2.buf.view(torch.int32), y.buf.view(torch.int32))
        print('{}: bytes equal to the three launches: {}'.format(name, same))
    if os.environ.get('GPP_BLOCK_STAMPS'):
        # diagnostic library (make variant NAME=blockstamps EXTRA=-DGPP_BLOCK_STAMPS CONV_UNITS=conv_block_x3; GPP_LIB=.../libgpp_hip_blockstamps.so)
        import numpy as np
        nwg = 1 << 16
        stamps = torch.zeros((nwg * 8,), dtype=torch.int64, device='cuda')
        f[0].zero_page = stamps.data_ptr()
        for t in tiles:
            for cold in (False, True):
                stamps.zero_()
                if cold:
                    trash.fill_(1)
                hip.check(run_block(f[0], f[1], f[2], t), 'block')
                torch.cuda.synchronize()
                s = stamps.cpu().numpy().reshape(nwg, 8)
                s = s[s[:, 0] > 0].astype(np.float64) / 100.0            # us
                t0 = s[:, 0].min()
                names = ['phase 1 (2a loop)', 'hand-over 1 (a-tile)', 'phase 2 (2b loop)', 'hand-over 2 (b-tile)', 'phase 3 (2c + stores)', 'store drain']
                print('tile {} {}: {} workgroups, launch {:.1f} us first start -> last end'.format(t, 'cold' if cold else 'hot', len(s), s[:, 6].max() - t0))
                for k, nm in enumerate(names):
                    dt = s[:, k + 1] - s[:, k]
                    print('   {:24s} median {:6.2f} us   p10 {:6.2f}   p90 {:6.2f}'.format(nm, np.median(dt), np.percentile(dt, 10), np.percentile(dt, 90)))
                life = s[:, 6] - s[:, 0]
                print('   {:24s} median {:6.2f} us   p10 {:6.2f}   p90 {:6.2f}'.format('workgroup life', np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
                starts = np.sort(s[:, 0] - t0)
                print('   starts: 256th {:.1f} us, 512th {:.1f} us, last {:.1f} us'.format(starts[min(255, len(starts) - 1)], starts[min(511, len(starts) - 1)], starts[-1]))
        f[0].zero_page = d[0].zero_page
        return
    for rep in range(2):
        for cold in (False, True):
            row = []
            for name, fn in variants:
                us = timed(fn, cold)
                row.append('{} {:.1f} us ({:.0f} TFLOP/s, {:.2f} TB/s of x+y)'.format(name, us, flops / us / 1e6, mb / us))
            print(('cold: ' if cold else 'hot:  ') + ' | '.join(row))


if __name__ == '__main__':
    main()
