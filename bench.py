#!/usr/bin/env python
"""
bench.py -- images/sec end-to-end of the predict_on_batch path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver under torch.distributed.run, one rank per GPU)

A step = one predict_on_batch-equivalent pass (ResNet-50 + FPN + heads + decode/NMS + ground-plane
polling, 1k-plane database) over a batch of 8 synthetic 1242x375 frames per GPU, already resized to
the network input 402x1333 and resident in HBM (the reference's own timer, bin/run_network.py:108-111,
also starts after preprocessing).  N > 1: every rank runs its own 8 images (weak scaling, BASELINE
config 3 = 64 images over 8 GPUs) and the step ends with ONE all-gather of the packed detections.

The JSON line also carries
  roofline      the dominant kernel = conv_igemm_kernel on the 3x3 512->512 regression-tower layers
                (45 % of all FLOPs): algorithmic FLOPs per launch / mean launch duration measured with
                HIP events inside the timed region, against the 2.5 PFLOP/s dense bf16 MFMA peak
  cpu_baseline  the CPU oracle (torch float32 restatement + NumPy decode + C polling) timed on this
                host on a bounded sample -- "CPU restatement, not TF1" (TF1 cannot be installed)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

MEAN = np.array([103.939, 116.779, 123.68], np.float32)
PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0}     # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=30)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    p.add_argument('--backbone', default='resnet50')
    p.add_argument('--planes', default='1k')
    p.add_argument('--dtype', default='bf16')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-images', type=int, default=2)
    return p.parse_args()


def synthetic_batch(batch, rank):
    """ uint8 noise frames 375x1242, 'resized' to the network input 402x1333 (nearest, host side,
    outside the timed region), BGR mean subtracted -- the tensor predict_on_batch receives """
    from keras_retinanet_3D.utils import synthetic
    out = np.empty((batch, 402, 1333, 3), np.float32)
    ys = np.minimum((np.arange(402) * (375.0 / 402.0)).astype(np.int64), 374)
    xs = np.minimum((np.arange(1333) * (1242.0 / 1333.0)).astype(np.int64), 1241)
    for i in range(batch):
        frame = synthetic.synthetic_image(seed=1000 * rank + i)
        out[i] = frame[ys][:, xs].astype(np.float32) - MEAN
    return out


def cpu_baseline(n_images, backbone, planes):
    """ whole path on the host cores with the oracle (bounded sample) """
    import torch
    from oracle import decode_np, net_torch
    from keras_retinanet_3D.models import weights as W
    from keras_retinanet_3D.utils import anchors as A
    from keras_retinanet_3D.utils import synthetic
    import ctypes
    import subprocess
    lib_path = os.path.join(ROOT, 'oracle', 'liboracle_polling.so')
    if not os.path.isfile(lib_path):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(lib_path)
    weights = W.synthetic_weights(backbone, 1234)
    net = net_torch.Net(weights, backbone)
    anchors = A.anchors_for_image((402, 1333))
    _, P_inv = synthetic.synthetic_calibration()
    P_inv = P_inv[None].astype(np.float32)
    img = synthetic_batch(1, 0)
    net.forward(img[:, :64, :96])                    # warm the thread pool
    t0 = time.perf_counter()
    for i in range(n_images):
        f = net.forward(synthetic_batch(1, 77 + i))
        det, _ = decode_np.detect(f['classification_logits'], f['regression'], f['regression_dim'], anchors)
        boxes, dims, orient = det[0], det[1], det[4]
        kp = np.empty((1, 100, 4, 3), np.float32)
        kpl = np.empty((1, 100, 1, 4), np.float32)
        res = np.empty((1, 100), np.float32)
        idx = np.empty((1, 100), np.int32)
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        lib.gpp_oracle_poll_f32(ptr(boxes), ptr(dims), ptr(orient), ptr(P_inv), ptr(planes), 1, 100, planes.shape[0], 0,
                                ctypes.c_float(0.7), ptr(kp), ptr(kpl), ptr(res), ptr(idx))
    dt = time.perf_counter() - t0
    return {'value': round(n_images / dt, 4), 'unit': 'images/s', 'cores': int(torch.get_num_threads()), 'kind': 'port',
            'sample': '{} synthetic 402x1333 frames, batch 1, whole path (torch-CPU float32 conv stack + NumPy decode/NMS + '
                      'C polling, {} planes); CPU restatement, not TF1'.format(n_images, planes.shape[0])}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    force_dist = os.environ.get('GPP_BENCH_FORCE_DIST') == '1'      # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)

    from keras_retinanet_3D import models
    from keras_retinanet_3D.backend import hip
    from keras_retinanet_3D.utils import synthetic
    from keras_retinanet_3D.utils import distributed as D
    import ctypes

    model = models.load_model('synthetic:1234', backbone_name=args.backbone, dtype=args.dtype)
    planes = synthetic.load_plane_database(args.planes).astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    B = args.batch
    images = torch.as_tensor(synthetic_batch(B, rank)).cuda()
    P_inv_d = torch.as_tensor(np.tile(P_inv[None].astype(np.float32), (B, 1, 1))).cuda()
    planes_d = torch.as_tensor(np.tile(planes[None], (B, 1, 1))).cuda()      # tiled per image, as kitti.py:220
    plan = model.stage_inputs([images, P_inv_d, planes_d])                    # inputs resident in HBM from here on
    torch.cuda.synchronize()

    lib = hip.lib()
    n_tagged = len(plan.tagged)
    events = []
    for _ in range(2 * n_tagged * args.steps):
        e = ctypes.c_void_p()
        hip.check(lib.gpp_event_create(ctypes.byref(e)))
        events.append(e)

    pending = []          # the previous step's gather: on the wire while this step computes, waited for before the next one is issued

    def step(k=None):
        ev = None if k is None else [e.value for e in events[2 * n_tagged * k: 2 * n_tagged * (k + 1)]]
        model.run_plan(plan, ev)
        if not (world > 1 or force_dist):
            return None                                          # the eight result arrays are the plan's output buffers
        packed = D.pack_outputs(model.outputs(plan))             # one launch (gpp_pack_detections): what the ranks exchange
        while pending:
            pending.pop().wait()
        out, work = D.gather_detections(packed, async_op=True)
        pending.append(work)
        return out

    for _ in range(args.warmup):
        out = step()
    while pending:
        pending.pop().wait()
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step(k)
    while pending:
        pending.pop().wait()                                   # the last gather completes inside the timed region
    torch.cuda.synchronize()
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1 or force_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: mean launch duration from the HIP events recorded inside the timed region
    durations = []
    for i in range(0, len(events), 2):
        ms = ctypes.c_float(0.0)
        hip.check(lib.gpp_event_elapsed_ms(events[i], events[i + 1], ctypes.byref(ms)))
        durations.append(ms.value)
    tags = [tag for _, tag, _, _, _ in plan.ops if tag]                   # per step: the tagged ops in launch order
    by_tag = {}
    for i, ms_ in enumerate(durations):
        by_tag.setdefault(tags[i % len(tags)], []).append(ms_)
    tagged_flops = [fl for _, tag, _, _, fl in plan.ops if tag == 1]       # tag 1 = the regression-tower launches
    flops_per_launch = float(np.mean(tagged_flops)) if tagged_flops else 0.0
    durations = by_tag.get(1, [])
    mean_ms = float(np.mean(durations)) if durations else float('nan')
    achieved = flops_per_launch / (mean_ms * 1e-3) / 1e12 if durations else float('nan')
    # polling (tag 2 = canonical planes + poll kernel): the three figures SURVEY 8(d) asks for, from the same live events
    polling = None
    if by_tag.get(2):
        poll_ms = float(np.mean(by_tag[2]))
        n_pl, n_det = int(planes.shape[0]), 100
        alg_bytes = B * (16.0 * n_pl + 13600.0)                            # plane DB tiled per image, as the reference feeds it
        l2_bytes = B * n_det * n_pl * 16.0                                 # every detection streams the (L2-resident) database
        poll_flops = 162.0 * B * n_det * n_pl
        polling = {'launch_us': round(poll_ms * 1e3, 1), 'planes': n_pl,
                   'algorithmic_GBps': round(alg_bytes / (poll_ms * 1e-3) / 1e9, 1),
                   'frac_of_hbm_peak': round(alg_bytes / (poll_ms * 1e-3) / 8e12, 5),
                   'l2_level_GBps': round(l2_bytes / (poll_ms * 1e-3) / 1e9, 1),
                   'valu_tflops': round(poll_flops / (poll_ms * 1e-3) / 1e12, 2),
                   'frac_of_fp32_vector_peak': round(poll_flops / (poll_ms * 1e-3) / 157.3e12, 4),
                   'note': 'latency / VALU bound (exact IEEE divide + sqrt per pair), not HBM bound: the whole input is 16*N + 13600 bytes per image'}
    # HBM/fabric traffic of that kernel cannot be read live (PMC needs rocprofv3): report the committed
    # measurement of the same kernel on the same workload (profiles/r1/dominant_kernel_pmc.*, tools/pmc_bench.sh)
    traffic = None
    pmc_path = os.path.join(ROOT, 'profiles', 'r1', 'dominant_kernel_pmc.json')
    if os.path.isfile(pmc_path) and args.backbone == 'resnet50' and B == 8 and args.dtype == 'bf16':
        with open(pmc_path) as f:
            traffic = round(json.load(f)['traffic_bytes_per_launch'] / 1e6, 1)
    counts = plan.counts.cpu().numpy()
    if out is None:
        out = D.pack_outputs(model.outputs(plan))
    dets = int((out[:, :, 15] > 0.05).sum().item())

    # informational, outside the timed region and never `value`: the same step fed from HOST memory --
    # raw uint8 frames uploaded over PCIe, preprocessed on the GPU, 8 result arrays copied back
    pcie_rate = None
    pcie_pipelined = None
    host_fed_detections = None
    if rank == 0 and world == 1:
        # binary noise keeps its contrast through the bilinear resize, so decode / NMS / polling see candidates here
        # too (uniform noise is smoothed to nothing by the resize and would make this leg's decode free)
        frames = (np.random.default_rng(5).integers(0, 2, size=(B, 375, 1242, 3)) * 255).astype(np.uint8)
        _, P_inv_s = synthetic.synthetic_calibration(1333.0 / 1242.0)
        P_host = np.tile(P_inv_s[None].astype(np.float32), (B, 1, 1))
        planes_host = np.tile(planes[None], (B, 1, 1))
        for _ in range(3):
            host_out = model.predict_on_frames(frames, P_host, planes_host)
        host_fed_detections = int((np.asarray(host_out[0][2]) > 0.05).sum())          # scores of the 8 reference outputs
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_it = 10
        for _ in range(n_it):
            model.predict_on_frames(frames, P_host, planes_host)
        torch.cuda.synchronize()
        pcie_rate = round(B * n_it / (time.perf_counter() - t1), 1)
        from keras_retinanet_3D.utils.pipeline import FramePipeline
        pipe = FramePipeline(model, depth=2)
        list(pipe.run(iter([(frames, P_host, planes_host)] * 3)))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_it = 20
        for _ in pipe.run(iter([(frames, P_host, planes_host)] * n_it)):
            pass
        pcie_pipelined = round(B * n_it / (time.perf_counter() - t1), 1)

    if rank == 0:
        total_images = world * B * args.steps
        rec = {
            'metric': 'images/sec end-to-end ({}, {} planes, 1242x375)'.format(args.backbone, args.planes),
            'value': round(total_images / elapsed, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'batch={} synthetic 1242x375 frames per GPU (network input 402x1333, resident in HBM), '
                                   '{} + FPN + heads + decode/NMS + polling, {}-plane database ({} planes), seeded random weights'.format(
                                       B, args.backbone, args.planes, planes.shape[0]),
                       'global_batch': world * B, 'parallelism': 'dp{} image shards, one all_gather of (B,100,35) f32'.format(world),
                       'candidates_per_image': [int(c) for c in counts], 'detections_rank0': dets,
                       'algorithmic_gflop_per_image': round(plan.flops / B / 1e9, 1),
                       'achieved_tflops_whole_path': round(plan.flops * args.steps / elapsed / 1e12, 1),
                       'host_fed_images_per_s_incl_pcie_and_gpu_preprocessing': pcie_rate,
                       'host_fed_images_per_s_pipelined_uploads': pcie_pipelined, 'host_fed_detections': host_fed_detections,
                       'polling_kernel': polling},
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 1), 'peak': PEAK_TFLOPS[args.dtype], 'unit': 'TFLOP/s',
                         'frac': round(achieved / PEAK_TFLOPS[args.dtype], 4), 'traffic': traffic,
                         'traffic_unit': 'MB per launch at the L2<->fabric interface (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes; '
                                         'algorithmic 192.1 MB)',
                         'kernel': 'conv_igemm_kernel<{},256,256,2,4,2> on pyramid_regression_1..3 (3x3, 512->512, 5 levels, M={})'.format(
                             args.dtype, B * (plan.n_anchors // 12)),
                         'gflop_per_launch': round(flops_per_launch / 1e9, 1), 'mean_launch_ms': round(mean_ms, 4),
                         'launches_timed': len(durations)},
        }
        if world == 1 and not args.no_cpu_baseline:
            rec['cpu_baseline'] = cpu_baseline(args.cpu_images, args.backbone, planes)
        print(json.dumps(rec))
    for e in events:
        lib.gpp_event_destroy(e)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
