""" Per conv layer of the B = 8 plan: the tile the autotuner chose, its grid, the bytes every workgroup has to pull through its CU's
vector-memory path for the FIRST time (weight rows BN x K once + activation rows: BM x C_in for a 1x1 layer, 3 x BM x C_in for a
3x3 layer -- the other two kernel columns re-hit the CU's L1, profiles/r2/shared_patch_v2_experiment.txt), the time those bytes
take at the measured per-CU fill ceiling (56 GB/s, tools/micro/fillbench.hip, profiles/r2/fill_rate_microbench.txt) for the CU
that gets the most workgroups, the MFMA floor, the HBM floor (round 4: the layer's compulsory bytes -- input map once, shortcut map,
output map, weights -- at 6.3 TB/s; the float32-sized maps of the x3 types make the 1x1 layers of res2 / res3 and the expand layers
HBM-bound, not fill-bound; a plain elementwise kernel reaches 5.9 TB/s hot and 4.3 - 4.9 TB/s cold on the same boxes,
tools/hbm_layers.py) and the measured launch time hot (back to back) and cold (600 MB rewritten in between, as inside the network).
    python tools/fill_floor_table.py [dtype] [backbone] [batch = 8]              (on the GPU box) """
import ctypes
import os
os.environ.setdefault('GPP_HALF_LANES', '')          # one launch per layer (whole batch)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from keras_retinanet_3D import models  # noqa: E402
from keras_retinanet_3D.backend import hip  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402
from keras_retinanet_3D.models.retinanet import OP_CONV  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
backbone = sys.argv[2] if len(sys.argv) > 2 else 'resnet50'
HBM_TBPS = 6.3
FILL_GBPS, PEAK = 56.0, {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3, 'bf16x3': 2500.0 / 3, 'f16x3': 2500.0 / 3}[dtype]
B, H, W = (int(sys.argv[3]) if len(sys.argv) > 3 else 8), 402, 1333
model = models.load_model('synthetic:1234', backbone_name=backbone, dtype=dtype)
plan = model.plan_for(B, H, W, 1000, True)
model.run_plan(plan)
torch.cuda.synchronize()
flush = torch.empty((600 << 20,), dtype=torch.uint8, device='cuda')
esz = C.elem_size(dtype)


def timed(index, cold):
    ts = []
    for _ in range(5 if cold else 1):
        if cold:
            flush.fill_(1)
        n = 1 if cold else 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            model.run_op(plan, index)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


print('{} {} B = {}: fill ceiling {} GB/s per CU, MFMA peak {:.0f} TFLOP/s'.format(backbone, dtype, B, FILL_GBPS, PEAK))
print('{:26s} {:>7s} {:>5s} {:>5s} {:>12s} {:>6s} {:>7s} {:>9s} {:>9s} {:>8s} {:>8s} {:>8s} {:>8s} {:>6s} {:>6s}'.format(
    'layer', 'M', 'K', 'N', 'tile', 'WGs', 'max/CU', 'KB/WG', 'fill us', 'mfma us', 'hbm us', 'hot us', 'cold us', 'hot/fl', 'cold/fl'))
tot = {'fill': 0.0, 'mfma': 0.0, 'hbm': 0.0, 'floor': 0.0, 'hot': 0.0, 'cold': 0.0}
for index, (kind, _, desc, name, flops) in enumerate(plan.ops):
    if kind != OP_CONV:
        continue
    M = sum(desc.batch * desc.groups[g].H_out * desc.groups[g].W_out for g in range(desc.n_groups))
    K = desc.KH * desc.KW * desc.C_in
    tile = plan.tuning.get(name, (0, 0))[0]
    if tile >= 3000000:
        bm, bn, label = (tile // 1000) % 1000, 256, 'mix {}+{}'.format((tile // 1000) % 1000, tile % 1000)
    elif tile >= 2000000:
        bm, bn, label = 256, 256, 'dual grid'
    else:
        bm, bn = (tile // 1000) % 1000 or 128, tile % 1000 or 128
        label = '{}x{}{}'.format(bm, bn, 'p' if tile // 1000000 == 1 else '')
    if M * desc.C_out > 3.0e7:
        continue                                         # the big matrix-pipe-bound layers are not the subject here
    wgs = sum(-(-desc.batch * desc.groups[g].H_out * desc.groups[g].W_out // bm) for g in range(desc.n_groups)) * -(-desc.C_out // bn)
    split = ctypes.c_int(0)
    hip.lib().gpp_conv2d_split_rule(ctypes.byref(desc), ctypes.byref(split))
    k_wg = K // max(1, split.value)
    rows_first = bm * desc.C_in * (3 if desc.KH == 3 else 1) / max(1, split.value)
    bytes_wg = (bn * k_wg + rows_first) * esz
    wgs_total = wgs * max(1, split.value)
    per_cu = -(-wgs_total // 256)
    fill_us = per_cu * bytes_wg / (FILL_GBPS * 1e3)
    mfma_us = flops / (PEAK * 1e6)
    # compulsory HBM bytes: every input pixel once, the shortcut map, the output map (float32 head outputs: 4 bytes), the weights
    m_in = sum(desc.batch * desc.groups[g].H_in * desc.groups[g].W_in for g in range(desc.n_groups))
    if desc.KH == 1 and desc.stride == 2:
        m_in = M                                         # a strided 1x1 layer reads every fourth pixel only
    m_res = sum(desc.batch * desc.groups[g].H_res * desc.groups[g].W_res for g in range(desc.n_groups)) if desc.residual else 0
    hbm_bytes = (m_in * desc.C_in + m_res * desc.C_out) * esz + M * desc.C_out * (4 if desc.out_f32 else esz) + K * desc.C_out * esz
    hbm_us = hbm_bytes / (HBM_TBPS * 1e6)
    hot, cold = timed(index, False), timed(index, True)
    floor = max(fill_us, mfma_us, hbm_us)
    for k, v in (('fill', fill_us), ('mfma', mfma_us), ('hbm', hbm_us), ('floor', floor), ('hot', hot), ('cold', cold)):
        tot[k] += v
    print('{:26s} {:7d} {:5d} {:5d} {:>12s} {:6d} {:7d} {:9.0f} {:9.1f} {:8.1f} {:8.1f} {:8.1f} {:8.1f} {:6.2f} {:6.2f}'.format(
        name[:26], M, K, desc.C_out, label + ('/k%d' % split.value if split.value > 1 else ''), wgs_total, per_cu, bytes_wg / 1024, fill_us, mfma_us,
        hbm_us, hot, cold, hot / floor, cold / floor))
print('sum over these layers: fill floor {:.0f} us, MFMA floor {:.0f} us, HBM floor {:.0f} us, max of the three per layer {:.0f} us; '
      'measured hot {:.0f} us, cold {:.0f} us'.format(tot['fill'], tot['mfma'], tot['hbm'], tot['floor'], tot['hot'], tot['cold']))
