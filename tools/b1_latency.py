#!/usr/bin/env python
"""
The reference's OWN timer on the HIP path: the latency of ONE synchronous batch-1 predict_on_batch.

/root/reference/keras_retinanet_3D/bin/run_network.py:108-111 brackets exactly this -- `start = time.time();
model.predict_on_batch([image (1,H,W,3) float32, P_inv (1,4,3), planes (1,N,4)]); time.time() - start` -- feed + run + fetch of one
preprocessed frame (BASELINE.json configs[0] is that call on the CPU).  measure() times it the same way on the drop-in model:

    sync_ms        wall time of model.predict_on_batch([...NumPy arrays...]) -> 8 NumPy arrays, median / p90 / min over `n` calls on
                   rotating frames (H2D of the 6.4 MB float32 image and D2H of the results INSIDE the bracket)
    plan_only_ms   the same plan with its inputs already in HBM, HIP events around gpp_plan_run (what the GPU needs)
    stages_ms      stem + backbone / FPN / heads (+ candidates, selection) / emit + polling, each sub-range of the plan run and timed
                   on its own (a sub-range ends with its side lanes joined, so the four do not overlap as they do in the whole plan)
    floor_ms       sum over the conv layers of max(MFMA time, compulsory HBM bytes at 6.3 TB/s) + launches x LAUNCH_GAP_US: what a
                   plan of this many dependent launches cannot beat (tools/fill_floor_table.py has the per-layer columns)

    python tools/b1_latency.py [--dtype f16x3] [--backbone resnet50] [--planes 1k] [--n 60]        (on the GPU box)
bench.py imports measure() for `config.b1` of its line (1-GPU runs, outside `value`).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))

LAUNCH_GAP_US = 4.0        # back-to-back dependent launches on one HIP stream of this platform (profiles/r4/default_plan_timeline.txt: 88 us over ~125 launches + the kernels' own ramp)
HBM_TBPS = 6.3
PEAK = {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3, 'bf16x3': 2500.0 / 3, 'f16x3': 2500.0 / 3}


def _stage_ranges(plan):
    names = [name for _, _, _, name, _ in plan.ops]
    i_fpn = names.index('C5_reduced')
    i_heads = names.index('pyramid_towers_0')
    i_emit = max(i for i, n in enumerate(names) if n == 'filtered_detections')
    return [('stem+backbone', 0, i_fpn), ('fpn', i_fpn, i_heads), ('heads+selection', i_heads, i_emit), ('emit+polling', i_emit, len(names))]


def _floor_ms(model, plan):
    from keras_retinanet_3D.layers import conv as C
    from keras_retinanet_3D.models.retinanet import OP_BLOCK, OP_CONV, OP_TAIL
    esz = C.elem_size(model.dtype)
    total, launches = 0.0, 0
    for kind, _, desc, name, flops in plan.ops:
        launches += 1
        descs = [desc] if kind == OP_CONV else []
        if kind == OP_TAIL:
            from keras_retinanet_3D.backend import hip
            descs = [ctypes.cast(desc.conv3x3, ctypes.POINTER(hip.ConvDesc)).contents, ctypes.cast(desc.conv1x1, ctypes.POINTER(hip.ConvDesc)).contents]
        if kind == OP_BLOCK:                 # a whole bottleneck in one launch: neither intermediate map reaches HBM
            from keras_retinanet_3D.backend import hip
            descs = [ctypes.cast(p, ctypes.POINTER(hip.ConvDesc)).contents for p in (desc.conv1x1_a, desc.conv3x3_b, desc.conv1x1_c)]
        for j, d in enumerate(descs):
            m_out = sum(d.batch * d.groups[g].H_out * d.groups[g].W_out for g in range(d.n_groups))
            m_in = sum(d.batch * d.groups[g].H_in * d.groups[g].W_in for g in range(d.n_groups))
            k = d.KH * d.KW * d.C_in
            mfma_us = 2.0 * m_out * k * d.C_out / (PEAK[model.dtype] * 1e6)
            byts = d.C_out * k * esz
            if not (kind == OP_TAIL and j == 1) and not (kind == OP_BLOCK and j >= 1):
                byts += m_in * d.C_in * esz                     # (the fused tail's / block's intermediate maps never reach HBM)
            if not (kind == OP_TAIL and j == 0) and not (kind == OP_BLOCK and j <= 1):
                byts += m_out * d.C_out * (4 if d.out_f32 else esz) + (m_out * d.C_out * esz if d.residual else 0)
            total += max(mfma_us, byts / (HBM_TBPS * 1e6))
    return (total + launches * LAUNCH_GAP_US) / 1e3, launches


def measure(model, planes, n=60, warm=5, hw=(402, 1333)):
    import torch
    from keras_retinanet_3D.backend import hip
    from keras_retinanet_3D.models.retinanet import PlanOp
    from keras_retinanet_3D.utils import synthetic
    H, W = hw
    frames = [synthetic.synthetic_network_input([5000 + i])[:, :H, :W] for i in range(4)]
    _, P_inv = synthetic.synthetic_calibration()
    P_inv = P_inv[None].astype(np.float32)
    planes1 = np.ascontiguousarray(planes[None], np.float32)
    for i in range(warm):
        model.predict_on_batch([frames[i % 4], P_inv, planes1])
    torch.cuda.synchronize()
    sync = []
    for i in range(n):
        t0 = time.perf_counter()
        out = model.predict_on_batch([frames[i % 4], P_inv, planes1])
        sync.append(1e3 * (time.perf_counter() - t0))
    dets = int((out[2] > 0.05).sum())
    plan = model.plan_for(1, H, W, planes.shape[0], True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        model.run_plan(plan)
        ev[i + 1].record()
    torch.cuda.synchronize()
    plan_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    stages = {}
    lib = hip.lib()
    for name, a, b in _stage_ranges(plan):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        e[0].record()
        for i in range(10):
            hip.check(lib.gpp_plan_run(ctypes.byref(plan.array, a * ctypes.sizeof(PlanOp)), b - a, hip.stream_ptr(), None, 0), 'gpp_plan_run')
            e[i + 1].record()
        torch.cuda.synchronize()
        stages[name] = round(sorted(e[i].elapsed_time(e[i + 1]) for i in range(10))[5], 4)
    # what the bracket adds to the plan: the upload of the float32 frame and the fetch of the eight arrays, each alone
    img_d = plan.images
    t0 = time.perf_counter()
    for i in range(10):
        model.stage_inputs([frames[i % 4], P_inv, planes1])
        torch.cuda.synchronize()
    h2d = 1e2 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for i in range(10):
        model.fetch(plan)
    d2h = 1e2 * (time.perf_counter() - t0)
    del img_d
    floor, launches = _floor_ms(model, plan)
    s = sorted(sync)
    return {'what': 'ONE synchronous predict_on_batch at batch 1: float32 frame {}x{} + P_inv + {} planes in host memory -> 8 NumPy arrays, as '
                    'bin/run_network.py:108-111 brackets it (upload and fetch inside the bracket)'.format(H, W, planes.shape[0]),
            'calls': n, 'sync_ms_median': round(s[n // 2], 4), 'sync_ms_p90': round(s[int(0.9 * (n - 1))], 4), 'sync_ms_min': round(s[0], 4),
            'images_per_s_at_the_median': round(1e3 / s[n // 2], 1),
            'plan_only_ms_median': round(plan_ms[n // 2], 4), 'plan_only_ms_p90': round(plan_ms[int(0.9 * (n - 1))], 4),
            'upload_ms': round(h2d, 4), 'fetch_ms': round(d2h, 4), 'stages_ms': stages,
            'floor_ms': round(floor, 4), 'launches': launches, 'plan_over_floor': round(plan_ms[n // 2] / floor, 3),
            'floor_note': 'sum over conv layers of max(MFMA time at the nominal peak, compulsory HBM bytes at 6.3 TB/s) + {} launches x {} us'.format(launches, LAUNCH_GAP_US),
            'detections_last_call': dets}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f16x3')
    ap.add_argument('--backbone', default='resnet50')
    ap.add_argument('--planes', default='1k')
    ap.add_argument('--n', type=int, default=60)
    ap.add_argument('--host-variants', action='store_true', help='also time packed / separate fetch x pinned / pageable upload')
    ap.add_argument('--variants', default=None, help='plan-builder environment settings to compare with the default: "A=1,B=2;C=3"')
    args = ap.parse_args()
    import torch
    torch.cuda.set_device(0)
    from keras_retinanet_3D import models
    from keras_retinanet_3D.utils import synthetic
    model = models.load_model('synthetic:1234', backbone_name=args.backbone, dtype=args.dtype)
    planes = synthetic.load_plane_database(args.planes).astype(np.float32)
    rec = measure(model, planes, n=args.n)
    if args.host_variants:
        # the two host-side choices of the bracket, each against the other form (same process, same plan)
        rec['variants_sync_ms_median'] = {}
        for fetch in ('packed', 'separate'):
            for upload in ('pageable', 'pinned'):
                os.environ['GPP_FETCH'], os.environ['GPP_UPLOAD'] = fetch, upload
                r = measure(model, planes, n=max(20, args.n // 2))
                rec['variants_sync_ms_median']['fetch={} upload={}'.format(fetch, upload)] = [r['sync_ms_median'], r['upload_ms'], r['fetch_ms']]
        del os.environ['GPP_FETCH'], os.environ['GPP_UPLOAD']
    # plan-builder settings against the default, each on a fresh model in this process (same box, same minute): "A=1,B=2;C=3" = two variants
    rec['plan_variants'] = {}
    for variant in [v for v in (args.variants or '').split(';') if v.strip()]:
        saved = {}
        for kv in variant.split(','):
            k, v = kv.split('=', 1)
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        m = models.load_model('synthetic:1234', backbone_name=args.backbone, dtype=args.dtype)
        if os.environ.get('B1_GRAPH') == '1':              # (pseudo-setting of this tool: the plan replayed as ONE HIP-graph launch)
            m.capture(m.plan_for(1, 402, 1333, planes.shape[0], True))
        r = measure(m, planes, n=max(20, args.n // 2))
        rec['plan_variants'][variant] = {k: r[k] for k in ('sync_ms_median', 'plan_only_ms_median', 'stages_ms', 'launches')}
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
        del m
        torch.cuda.empty_cache()
    rec['dtype'], rec['backbone'] = args.dtype, args.backbone
    from keras_retinanet_3D.backend import hip
    rec['library'] = hip.lib().gpp_version().decode()
    rec['tiles'] = {k: v[0] for k, v in model.plan_for(1, 402, 1333, planes.shape[0], True).tuning.items()}
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
