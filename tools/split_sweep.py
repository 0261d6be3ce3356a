""" Sweep (block tile) x (split-K factor) over the small-M / deep-K layers of the backbone and the FPN head of the pyramid, cold and
hot, to choose gpp_conv2d_split_rule: the rule has to be a function of the LAYER ALONE (kernel, channels, output pixels per image),
so it is read off this table once, not tuned at run time.
    python tools/split_sweep.py [B] [dtype]
Prints, per layer: the best unsplit (tile, us), and the best (tile, split, us) for every split factor. """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
import torch  # noqa: E402
from keras_retinanet_3D.backend import hip  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402
import ctypes  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
LAYERS = [('res4x 2a 1x1 1024->256', (26, 84), 1024, 256, 1, False), ('res4 2b 3x3 256->256', (26, 84), 256, 256, 3, False),
          ('res4 2c 1x1 256->1024 +res', (26, 84), 256, 1024, 1, True), ('res5a 2a 1x1 s1 1024->512', (13, 42), 1024, 512, 1, False),
          ('res5x 2a 1x1 2048->512', (13, 42), 2048, 512, 1, False), ('res5 2b 3x3 512->512', (13, 42), 512, 512, 3, False),
          ('res5 2c 1x1 512->2048 +res', (13, 42), 512, 2048, 1, True), ('C4_reduced 1x1 1024->512 +res', (26, 84), 1024, 512, 1, True),
          ('P4 3x3 512->512', (26, 84), 512, 512, 3, False), ('P5 3x3 512->512', (13, 42), 512, 512, 3, False)]
TILES = [64064, 96064, 128064, 64128, 96128, 128128, 160128, 192128, 1128128, 1128256, 1192128]
dev = torch.device('cuda')
flush = torch.empty((600 << 20,), dtype=torch.uint8, device=dev)
ws = torch.empty((256 << 20,), dtype=torch.uint8, device=dev)


def time_one(d, cold):
    if hip.lib().gpp_conv2d_igemm(ctypes.byref(d), hip.stream_ptr()) != 0:
        return None
    ts = []
    for _ in range(5 if cold else 1):
        if cold:
            flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 1 if cold else 20
        e0.record()
        for _ in range(n):
            C.run_conv(d)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


for name, (h, w_), cin, cout, k, residual in LAYERS:
    tdt = C.torch_dtype(dtype)
    x = (torch.randn((B, h, w_, cin), device=dev) * 0.5).to(tdt)
    o = torch.empty((B, h, w_, cout), device=dev, dtype=tdt)
    wt = C.pack_weight((torch.randn((k, k, cin, cout)) * 0.02).numpy(), dtype, dev)
    bias = torch.zeros((cout,), device=dev)
    res = [C.FMap((torch.randn((B, h, w_, cout), device=dev) * 0.5).to(tdt), B, h, w_, cout)] if residual else None
    nk = k * k * cin // C.k_chunk(dtype)
    for cold in (False, True):
        best = {}
        for split in (1, 2, 3, 4, 6, 8):
            if nk // split < 4:
                continue
            for tile in TILES:
                d = C.conv_desc([C.FMap(x, B, h, w_, cin)], [C.FMap(o, B, h, w_, cout)], wt, bias, k, k, cin, cout, pad=(k // 2, k // 2), relu=True,
                                dtype=dtype, tile_hint=tile, residuals=res, workspace=ws, split_k=split)
                us = time_one(d, cold)
                if us is not None and (split not in best or us < best[split][1]):
                    best[split] = (tile, us)
        print('{:32s} B={} {:4s} '.format(name, B, 'cold' if cold else 'hot') +
              '  '.join('k{}: {:7d} {:6.1f}us'.format(s, t, us) for s, (t, us) in sorted(best.items())), flush=True)
