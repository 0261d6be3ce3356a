#!/usr/bin/env python
"""
bench.py -- images/sec end-to-end of the predict_on_batch path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--dtype f16x3|f32|bf16x3|f16|bf16] [--all-dtypes]
    (N > 1: launched by the driver under torch.distributed.run, one rank per GPU)

A step = one predict_on_batch-equivalent pass (ResNet-50 + FPN + heads + decode/NMS + ground-plane
polling, 1k-plane database) over a batch of 8 synthetic 1242x375 frames per GPU, already resized to
the network input 402x1333 and resident in HBM (the reference's own timer, bin/run_network.py:108-111,
also starts after preprocessing).  N > 1: every rank runs its own 8 images (weak scaling, BASELINE
config 3 = 64 images over 8 GPUs) and the step ends with ONE all-gather of the packed detections.

`--dtype` is the arithmetic of the conv stack.  The DEFAULT, f16x3, is the fastest type whose results stay inside BASELINE.json's
tolerance against the reference-precision path (same detections, same plane index for every one, 3-D corners within 1e-3 m:
utils/ledger.REFERENCE_BARS): float32 storage, every float32 product as three IEEE-half matrix products on
v_mfma_f32_16x16x32_f16 (11 + 11 significant bits per operand, ~2^-22 per product; float32: 2^-24), float32 accumulation.
f32 = the reference's own arithmetic type (float32 operands on v_mfma_f32_16x16x4_f32, 1/16 of the 16-bit matrix rate);
bf16x3 (three bf16 products, ~2^-16), f16, bf16 = faster types that do NOT meet the tolerance with these weights and are therefore
never the headline: `--all-dtypes` measures them on the same frames with their ledgers (config.other_types_same_frames).
Decode and polling are float32 / int32 in every mode.  A single-GPU run ALSO measures the float32 path on the same frames
(config.f32_images_per_s) and checks its own headline against it (config.parity_ledger, config.parity_bars_met); a headline
that misses a bar makes the run exit non-zero after printing the line.

The JSON line also carries
  roofline      the dominant kernel = conv_igemm_kernel on the 3x3 512->512 regression-tower layers
                (45 % of all FLOPs): algorithmic FLOPs per launch / mean launch duration measured with
                HIP events inside the timed region, against the dense MFMA peak of the operand type
  cpu_baseline  the CPU oracle (torch float32 restatement + NumPy decode + C polling) timed on this
                host on a bounded sample -- "CPU restatement, not TF1" (TF1 cannot be installed); the same leg replays
                decode + polling of the GPU's own head tensors of the last timed step and reports whether they agree
                bit for bit (config.gpu_decode_polling_replay_bit_exact)
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
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md (f32: v_mfma_f32_16x16x4_f32 / 32x32x2, = the vector rate)
# bf16x3: three bf16 matrix products per float32 product -> a third of the bf16 peak in float32-product FLOPs
PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3, 'bf16x3': 2500.0 / 3.0, 'f16x3': 2500.0 / 3.0}
# what the matrix pipe ALONE sustains on random operands (register-only MFMA loops, tools/micro/mfma_power.hip, profiles/r3/mfma_power.txt:
# measured once, not by this script): the clock under matrix load depends on the data, zeros run at the nominal figure above
PIPE_ON_RANDOM_DATA_TFLOPS = {'bf16': 2033.9, 'f16': 1753.4, 'bf16x3': 2033.9 / 3.0, 'f16x3': 1753.4 / 3.0}
PIPE_ON_POST_RELU_DATA_TFLOPS = {'bf16': 2134.7, 'f16': 1961.8, 'bf16x3': 2134.7 / 3.0, 'f16x3': 1961.8 / 3.0}      # half of the activation values zero
PROFILE_ROUNDS = ('r6', 'r5', 'r4', 'r3')      # the PMC file of the newest round whose library hash matches the running build is quoted
# algorithmic bytes of one regression-tower launch at B = 8: M x 512 channels in + out and the packed weights,
# 2 bytes per element for the 16-bit storage types, 4 for the float32-sized maps of f32 / bf16x3 / f16x3
ALGORITHMIC_MB = {'bf16': 192.1, 'f16': 192.1, 'f32': 384.2, 'bf16x3': 384.2, 'f16x3': 384.2}
RESIDENT_BATCHES = 6      # the timed steps rotate over this many distinct resident batches (6 x 51 MB > the 256 MB Infinity Cache)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=30)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    p.add_argument('--backbone', default='resnet50')
    p.add_argument('--planes', default='1k')
    p.add_argument('--dtype', default='f16x3', choices=['bf16', 'f16', 'f32', 'bf16x3', 'f16x3'])
    p.add_argument('--all-dtypes', action='store_true', help='also measure the other types on the same frames, with their ledgers')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-f32-leg', action='store_true', help='skip the float32 leg + parity ledger of a 16-bit run')
    p.add_argument('--no-host-fed', action='store_true', help='skip the host-fed (PCIe-inclusive) legs')
    p.add_argument('--cpu-images', type=int, default=4, help='frames of the bounded CPU-baseline sample (about 3.5 s each on 128 host threads)')
    p.add_argument('--no-b1', action='store_true', help='skip the batch-1 synchronous latency leg (config.b1)')
    p.add_argument('--repeats', type=int, default=3, help='further timed repeats of the K steps after the headline loop (config.repeat_images_per_s)')
    p.add_argument('--dry-launch', action='store_true',
                   help='rehearse the rank launcher without a GPU: every rank joins a gloo group, gathers a packed (B,100,35) tensor and exits')
    return p.parse_args()


def launch_ranks(args):
    """ `python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks here, one child process per
    GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what `python -m torch.distributed.run --nproc-per-node N` would
    give them.  Nothing in this (parent) process touches the GPU.  Rank 0's stdout is this process's stdout (the one JSON line);
    the other ranks' stdout goes to stderr.  The first non-zero exit code ends the job and is returned. """
    import socket
    import subprocess
    n = args.gpus
    if not args.dry_launch:
        import torch
        have = torch.cuda.device_count()                        # (counting devices does not initialise the GPU)
        if have < n:
            raise SystemExit('bench.py --gpus {} needs {} devices, this host has {}: refusing to report a {}-GPU number '
                             'from fewer ranks'.format(n, n, have, n))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:                                  # a rank died: the others would wait in a collective for ever
                    q.terminate()
    return rc


class c_stdout_to_stderr(object):
    """ RCCL (and gloo) print a banner on the C-level STDOUT when their first communicator comes up; this program's stdout carries
    exactly one JSON line, so file descriptor 1 points at stderr while a communicator is built """

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


DIAG_KEYS = ('ms_per_step', 'gather_wait_ms_per_step', 'sclk_mhz_median', 'power_w_median', 'power_cap_w', 'model_load_s', 'plan_build_and_tune_s')


def diagnose_over_ranks(local, device):
    """ every rank's own figures (DIAG_KEYS; a missing one is NaN) -> per key the list over ranks + min / median / max: what a reader of
    a non-linear scaling curve looks at first -- a slow rank, a card at a lower clock under the same power cap, an exposed gather,
    a rank that was still tuning.  One all_gather of len(DIAG_KEYS) doubles, outside the timed region. """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.tensor([float(local.get(k, float('nan'))) if local.get(k) is not None else float('nan') for k in DIAG_KEYS],
                        dtype=torch.float64, device=device)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    table = torch.stack(allv).cpu().numpy()                       # (world, keys)
    out = {'ranks': world}
    for j, k in enumerate(DIAG_KEYS):
        col = table[:, j]
        ok = col[np.isfinite(col)]
        out[k] = {'per_rank': [None if not np.isfinite(v) else round(float(v), 4) for v in col],
                  'min': None if not len(ok) else round(float(ok.min()), 4), 'median': None if not len(ok) else round(float(np.median(ok)), 4),
                  'max': None if not len(ok) else round(float(ok.max()), 4)}
    return out


def dry_launch(args, rank, world):
    """ what a rank does under --dry-launch: the rendezvous + the one collective of the path on gloo, no GPU """
    import torch
    import torch.distributed as dist
    from keras_retinanet_3D.utils import distributed as D
    if args.batch < 1:
        raise SystemExit('--batch must be at least 1')
    with c_stdout_to_stderr():
        dist.init_process_group('gloo', rank=rank, world_size=world)
        packed = torch.full((args.batch, 100, D.PACK_WIDTH), float(rank), dtype=torch.float32)
        out = D.gather_detections(packed, [args.batch] * world)
    ok = all(bool((out[r * args.batch:(r + 1) * args.batch] == float(r)).all()) for r in range(world))
    # the diagnosis block of a real multi-GPU line, rehearsed with stand-in figures (rank r reports r in every field it has)
    with c_stdout_to_stderr():
        diag = diagnose_over_ranks({k: float(rank) for k in DIAG_KEYS if k != 'power_cap_w'}, torch.device('cpu'))
    dist.barrier()
    if rank == 0:
        print(json.dumps({'dry_launch': True, 'n_gpus': world, 'world_size': dist.get_world_size(), 'gpus_requested': args.gpus,
                          'gathered_images_per_step': int(out.shape[0]), 'gather_correct': ok, 'multi_gpu_diagnosis': diag}))
    dist.destroy_process_group()
    return 0 if ok else 1


def synthetic_batch(batch, rank):
    """ uint8 noise frames 375x1242, 'resized' to the network input 402x1333 (nearest, host side,
    outside the timed region), BGR mean subtracted -- the tensor predict_on_batch receives """
    from keras_retinanet_3D.utils import synthetic
    return synthetic.synthetic_network_input(range(1000 * rank, 1000 * rank + batch))


def cpu_baseline(n_images, backbone, planes, replay=None):
    """ whole path on the host cores with the oracle (bounded sample); `replay` = the GPU's head tensors + outputs of
    the last timed step (first images): decode + polling recomputed by the oracle must give the same bits """
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
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731

    def poll(boxes, dims, orient, n):
        kp = np.empty((n, 100, 4, 3), np.float32)
        kpl = np.empty((n, 100, 1, 4), np.float32)
        res = np.empty((n, 100), np.float32)
        idx = np.empty((n, 100), np.int32)
        pinv = np.ascontiguousarray(np.tile(P_inv, (n, 1, 1)))
        lib.gpp_oracle_poll_f32(ptr(np.ascontiguousarray(boxes)), ptr(np.ascontiguousarray(dims)), ptr(np.ascontiguousarray(orient)),
                                ptr(pinv), ptr(planes), n, 100, planes.shape[0], 0, ctypes.c_float(0.7), ptr(kp), ptr(kpl), ptr(res), ptr(idx))
        return kp, kpl, res, idx

    t0 = time.perf_counter()
    for i in range(n_images):
        f = net.forward(synthetic_batch(1, 77 + i))
        det, _ = decode_np.detect(f['classification_logits'], f['regression'], f['regression_dim'], anchors)
        poll(det[0], det[1], det[4], 1)
    dt = time.perf_counter() - t0
    rec = {'value': round(n_images / dt, 4), 'unit': 'images/s', 'cores': int(torch.get_num_threads()), 'kind': 'port',
           'sample': '{} synthetic 402x1333 frames, batch 1, whole path (torch-CPU float32 conv stack + NumPy decode/NMS + '
                     'C polling, {} planes); CPU restatement, not TF1'.format(n_images, planes.shape[0])}
    exact = None
    if replay is not None:
        n = replay['cls'].shape[0]
        det, aidx = decode_np.detect(replay['cls'], replay['reg'], replay['dim'], anchors)
        exact = all(np.array_equal(a, b) for a, b in zip(det, replay['out'][:5])) and np.array_equal(aidx, replay['anchor_index'])
        kp, kpl, res, idx = poll(replay['out'][0], replay['out'][1], replay['out'][4], n)
        same = lambda a, b: bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))  # noqa: E731
        exact = bool(exact and np.array_equal(idx, replay['plane_index']) and same(kp, replay['out'][5]) and
                     same(kpl, replay['out'][6]) and same(res, replay['out'][7]))
    return rec, exact


def unfuse(reg, n_base=12):
    """ fused conv layout (B, P, 144) -> reference layout (B, A, 12) """
    B, P, _ = reg.shape
    op1 = reg[:, :, :4 * n_base].reshape(B, P, n_base, 4)
    rest = [reg[:, :, 4 * n_base + 2 * n_base * k: 4 * n_base + 2 * n_base * (k + 1)].reshape(B, P, n_base, 2) for k in range(4)]
    return np.concatenate([op1] + rest, axis=3).reshape(B, P * n_base, 12)


def tile_name(code):
    if not code:
        return 'library heuristic'
    if code in (64, 128, 256, 512):                     # legacy codes (include/gpp.h)
        return {64: '128x64', 128: '128x128', 256: '256x128 (3-deep ring)', 512: '256x256 pipelined'}[code]
    if code // 1000000 == 2:
        return '256x256 + 512x128 dual grid'
    if code // 1000000 == 3:
        return '{0}x256 + {1}x256 mixed-height grid pipelined'.format((code // 1000) % 1000, code % 1000)
    bm, bn = (code // 1000) % 1000, code % 1000
    return '{}x{}{}'.format(bm, bn, ' pipelined' if code // 1000000 == 1 else '')


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:        # no launcher around us: be the launcher (before any GPU call)
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus {} was started with WORLD_SIZE={}: the line would claim a GPU count it did not run on'.format(
            args.gpus, world))
    if args.dry_launch:
        raise SystemExit(dry_launch(args, rank, world))
    import torch
    import torch.distributed as dist
    force_dist = os.environ.get('GPP_BENCH_FORCE_DIST') == '1'      # exercise the RCCL path on a single GPU
    distributed = world > 1 or force_dist
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        with c_stdout_to_stderr():
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
            dist.barrier()
            torch.cuda.synchronize()
    else:
        torch.cuda.set_device(0)

    from keras_retinanet_3D import models
    from keras_retinanet_3D.backend import hip
    from keras_retinanet_3D.utils import synthetic
    from keras_retinanet_3D.utils import distributed as D
    from keras_retinanet_3D.utils import ledger
    import ctypes

    # host-side preparation first (seconds of NumPy work): once the plan is built and tuned nothing but a few uploads stands between the GPU's
    # last tuning launch and the first warm-up step (a GPU that sat idle while the host generated frames ran its first ~25 steps 1.5 % slow)
    host_batches = [synthetic_batch(args.batch, 1000 * rank + 100 * j) for j in range(1, RESIDENT_BATCHES)]
    t_load = time.perf_counter()
    model = models.load_model('synthetic:1234', backbone_name=args.backbone, dtype=args.dtype)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t_load
    planes = synthetic.load_plane_database(args.planes).astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    B = args.batch
    images = torch.as_tensor(synthetic_batch(B, rank)).cuda()
    P_inv_d = torch.as_tensor(np.tile(P_inv[None].astype(np.float32), (B, 1, 1))).cuda()
    planes_d = torch.as_tensor(np.tile(planes[None], (B, 1, 1))).cuda()      # tiled per image, as kitti.py:220
    t_plan = time.perf_counter()
    plan = model.stage_inputs([images, P_inv_d, planes_d])                    # inputs resident in HBM from here on (first call: plan build + tile timing)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t_plan
    # RESIDENT_BATCHES distinct batches of frames live in HBM; step k reads batch k mod RESIDENT_BATCHES (the stem's input pointer is
    # switched on the host, nothing is copied): the 51 MB of input are not served from the Infinity Cache step after step.
    # Batch 0 (the plan's own buffer) is the one the ledger legs and the oracle replay use; the last timed step lands on it.
    batches = [plan.images] + [torch.as_tensor(hb).cuda() for hb in host_batches]
    del host_batches
    stem_desc = plan.ops[0][2]
    assert hasattr(stem_desc, 'inp') and stem_desc.inp == plan.images.data_ptr()
    torch.cuda.synchronize()

    lib = hip.lib()
    n_tagged = len(plan.tagged)
    # HIP events bracket the tagged launches (regression tower, polling) on every EVENT_EVERY-th timed step only: an event
    # pair costs ~12 us of stream time around the launch it brackets, which the other steps do not pay
    EVENT_EVERY = 3
    sampled = [k for k in range(args.steps) if k % EVENT_EVERY == 0]
    events = []
    for _ in range(2 * n_tagged * len(sampled)):
        e = ctypes.c_void_p()
        hip.check(lib.gpp_event_create(ctypes.byref(e)))
        events.append(e)

    pending = []          # the previous step's gather: on the wire while this step computes, waited for before the next one is issued
    gather_wait_s = [0.0]

    def wait_pending():
        t_w = time.perf_counter()
        while pending:
            pending.pop().wait()
        gather_wait_s[0] += time.perf_counter() - t_w

    def step(k=None, warm=0):
        # (the last timed step reads batch 0 again: its outputs are what the ledger compares; the W warm-up steps walk the resident batches
        # too, so that no timed step is the first launch that ever touches its 51 MB of input)
        stem_desc.inp = batches[warm % RESIDENT_BATCHES if k is None else (args.steps - 1 - k) % RESIDENT_BATCHES].data_ptr()
        ev = None
        if k is not None and k % EVENT_EVERY == 0:
            i = k // EVENT_EVERY
            ev = [e.value for e in events[2 * n_tagged * i: 2 * n_tagged * (i + 1)]]
        model.run_plan(plan, ev)
        if not distributed:
            return None                                          # the eight result arrays are the plan's output buffers
        packed = D.pack_outputs(model.outputs(plan))             # one launch (gpp_pack_detections): what the ranks exchange
        wait_pending()
        out, work = D.gather_detections(packed, async_op=True)
        pending.append(work)
        return out

    # (the sampler is BUILT here, before the warm-up: finding the card's hwmon files walks sysfs for tens of milliseconds, and a device left idle that
    # long between the warm-up and the timed region starts the region from its idle clocks -- see DESIGN 6)
    from keras_retinanet_3D.utils import devmon
    monitor = devmon.Sampler(local_rank if distributed else 0, period=float(os.environ.get('GPP_DEVMON_PERIOD_S', '0.02')))     # (a thread reading sysfs files: no GPU call)
    if distributed:
        dist.barrier()          # the ranks leave plan build + tuning seconds apart: without this the early ones would sit idle at the barrier BEHIND the warm-up
    #                             and start the timed region from idle clocks (the same ~3.5 ms, on every rank but the last)
    for i in range(args.warmup):
        out = step(warm=i + 1)
    wait_pending()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    gather_wait_s[0] = 0.0
    monitor.__enter__()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step(k)
    wait_pending()                                             # the last gather completes inside the timed region
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0                     # this rank alone, before it waits for the others
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # (the sampler is stopped AFTER the clock is read: joining its thread waits for a sysfs read in flight -- 2 - 4 ms, which rounds 3 - 6 had inside
    # the bracket: 0.4 % of a 100-step region, 1 % of the 30-step default; the repeats below never had it, hence their higher rates in those rounds' lines)
    monitor.__exit__(None, None, None)
    device_state = monitor.summary()
    per_rank = None
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # diagnosis of a scaling run, rank by rank: its own time for its K steps, the host time it spent blocked on the previous step's
        # gather (exposed collective time; the gather itself overlaps the next step's kernels), the clock and power the driver granted
        # ITS card during the timed region (eight boards that each want the 1400 W cap are the first suspect of a non-linear curve),
        # and how long it took to load the model and to build + tune its plan
        ds = device_state or {}
        per_rank = diagnose_over_ranks({'ms_per_step': 1e3 * own_elapsed / args.steps, 'gather_wait_ms_per_step': 1e3 * gather_wait_s[0] / args.steps,
                                        'sclk_mhz_median': ds.get('sclk_mhz_median'), 'power_w_median': ds.get('power_w_median'),
                                        'power_cap_w': ds.get('power_cap_w'), 'model_load_s': t_load, 'plan_build_and_tune_s': t_plan},
                                       torch.device('cuda', local_rank))
        per_rank['ms_per_step_min_over_ranks'] = per_rank['ms_per_step']['min']
        per_rank['ms_per_step_max_over_ranks'] = per_rank['ms_per_step']['max']
        per_rank['gather_wait_ms_per_step_min_over_ranks'] = per_rank['gather_wait_ms_per_step']['min']
        per_rank['gather_wait_ms_per_step_max_over_ranks'] = per_rank['gather_wait_ms_per_step']['max']
    gathered_images = int(out.shape[0]) if out is not None else B
    rccl_world = dist.get_world_size() if distributed else 1
    if rccl_world != args.gpus:
        raise SystemExit('RCCL reports {} ranks, --gpus {} was asked for'.format(rccl_world, args.gpus))

    # dominant kernel: mean launch duration from the HIP events recorded inside the timed region
    tags = [tag for _, tag, _, _, _ in plan.ops if tag]                   # per step: the tagged ops in launch order

    def read_events():
        by = {}
        for i in range(0, len(events), 2):
            ms = ctypes.c_float(0.0)
            hip.check(lib.gpp_event_elapsed_ms(events[i], events[i + 1], ctypes.byref(ms)))
            by.setdefault(tags[(i // 2) % len(tags)], []).append(ms.value)
        return by

    by_tag = read_events()
    tagged_flops = [fl for _, tag, _, _, fl in plan.ops if tag == 1]       # tag 1 = the regression-tower launches
    flops_per_launch = float(np.mean(tagged_flops)) if tagged_flops else 0.0
    durations = by_tag.get(1, [])
    mean_ms = float(np.mean(durations)) if durations else float('nan')
    achieved = flops_per_launch / (mean_ms * 1e-3) / 1e12 if durations else float('nan')

    # the line's own error bar (reported, never `value`): the same K steps timed again --repeats times, same bracket (barrier +
    # synchronize on both sides, max over ranks), same rotation over the resident batches, the dominant kernel's events read per repeat
    repeat_rates, repeat_fracs = [], []
    for _ in range(max(0, args.repeats)):
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            out = step(k)
        wait_pending()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        repeat_rates.append(round(world * B * args.steps / dt, 2))
        d1 = read_events().get(1, [])
        repeat_fracs.append(round(flops_per_launch / (float(np.mean(d1)) * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4) if d1 else None)
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
    # HBM/fabric traffic of that kernel cannot be read live (PMC needs rocprofv3): report the committed measurement of the
    # same kernel on the same workload (profiles/<round>/dominant_kernel_pmc.json, tools/pmc_bench.sh) -- but ONLY when it
    # was collected with this very build of the library (gpp_version() carries a hash of the kernel sources)
    traffic, traffic_note = None, None
    version = lib.gpp_version().decode()
    for rnd in PROFILE_ROUNDS:
        pmc_path = os.path.join(ROOT, 'profiles', rnd, 'dominant_kernel_pmc_{}.json'.format(args.dtype))
        if traffic is not None or not (os.path.isfile(pmc_path) and args.backbone == 'resnet50' and B == 8):
            continue
        with open(pmc_path) as f:
            pmc = json.load(f)
        if pmc.get('library_version') == version:
            traffic = round(pmc['traffic_bytes_per_launch'] / 1e6, 1)
            traffic_note = None
        elif traffic_note is None:
            traffic_note = 'omitted: {} was collected with "{}", this build is "{}"'.format(
                os.path.relpath(pmc_path, ROOT), pmc.get('library_version'), version)
    counts = plan.counts.cpu().numpy()
    if out is None:
        out = D.pack_outputs(model.outputs(plan))
    dets = int((out[:, :, 15] > 0.05).sum().item())
    extras = rank == 0 and world == 1

    # outputs + head tensors of the last timed step (host copies): the ledger's run under test, the oracle's replay input
    main_outs = [t.cpu().numpy() for t in model.outputs(plan)]
    main_anchor, main_plane = plan.anchor_index.cpu().numpy(), plan.best_index.cpu().numpy()
    replay = None
    if extras and not args.no_cpu_baseline:
        nrep = min(2, B)
        replay = {'cls': plan.cls_logits[:nrep].cpu().numpy().reshape(nrep, -1, 8), 'reg': unfuse(plan.regression[:nrep].cpu().numpy()),
                  'dim': plan.regression_dim[:nrep].cpu().numpy().reshape(nrep, -1, 3), 'out': [o[:nrep] for o in main_outs],
                  'anchor_index': main_anchor[:nrep], 'plane_index': main_plane[:nrep]}

    # ---- the float32 (reference-precision) leg of a run at another type + the parity ledger against it; a default (bf16)
    # run also measures the two other fast types on the same frames: f16 and bf16x3
    def leg(dtype, n_timed):
        m = models.load_model('synthetic:1234', backbone_name=args.backbone, dtype=dtype)
        p = m.stage_inputs([images, P_inv_d, planes_d])
        for _ in range(2):
            m.run_plan(p)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_timed):
            m.run_plan(p)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        res = {'images_per_s': round(B * n_timed / dt, 2), 'ms_per_step': round(1e3 * dt / n_timed, 3),
               'achieved_tflops_whole_path': round(p.flops * n_timed / dt / 1e12, 1),
               'frac_of_mfma_peak_whole_path': round(p.flops * n_timed / dt / 1e12 / PEAK_TFLOPS[dtype], 4)}
        state = ([t.cpu().numpy() for t in m.outputs(p)], p.anchor_index.cpu().numpy(), p.best_index.cpu().numpy())
        del m, p
        torch.cuda.empty_cache()
        return res, state

    # ---- the run under test against the CPU ORACLE, from the committed full-size fixtures (tests/golden/fullsize_*.npz, made by
    # oracle/gen_fullsize_goldens.py: the oracle's detections for frames 0 .. of every BASELINE backbone / plane database; batch 0 of
    # rank 0 is frames 0 .. B-1).  f64 = the conv stack in float64: the exact value of what the reference's float32 graph computes --
    # THE bars (utils/ledger.REFERENCE_BARS) are measured against it; f32 = one float32 CPU evaluation, informational
    oracle_ledgers = {}
    fixture = os.path.join(ROOT, 'tests', 'golden', 'fullsize_{}_{}_{{}}.npz'.format(args.backbone, args.planes))
    if rank == 0 and world == 1 and os.path.isfile(fixture.format('f64')):
        for prec in ('f64', 'f32'):
            g = np.load(fixture.format(prec))
            if len(g['frames']) >= B:
                ref = ([g[k][:B] for k in ('boxes', 'dimensions', 'scores', 'labels', 'orientations', 'keypoints', 'keyplanes', 'residuals')],
                       g['anchor_index'][:B], g['plane_index'][:B])
                led = ledger.parity_ledger(*(ref + (main_outs, main_anchor, main_plane)))
                led['what'] = '{} HIP path vs the {} CPU oracle (committed fixture), frames 0..{}'.format(args.dtype, prec, B - 1)
                led['meets_reference_bars'] = ledger.meets_reference_bars(led, pair=(prec == 'f32'))
                oracle_ledgers[prec] = led

    f32_leg = None
    parity = None
    other_legs = {}
    bars_met = oracle_ledgers['f64']['meets_reference_bars'] if 'f64' in oracle_ledgers else None
    if extras and args.dtype != 'f32' and not args.no_f32_leg:
        f32_leg, ref_state = leg('f32', 10)
        parity = ledger.parity_ledger(*(ref_state + (main_outs, main_anchor, main_plane)))
        parity['what'] = '{} HIP path vs float32 HIP path, same {} frames, same weights (two float32-grade runs: pair bars, 2e-3 m)'.format(args.dtype, B)
        parity['meets_reference_bars_as_a_pair'] = ledger.meets_reference_bars(parity, pair=True)
        bars_met = parity['meets_reference_bars_as_a_pair'] and (bars_met is None or bars_met)
        if args.all_dtypes:
            for other in ('bf16x3', 'f16', 'bf16', 'f16x3'):
                if other == args.dtype:
                    continue
                res, state = leg(other, 8)
                res['parity_ledger_vs_f32'] = ledger.parity_ledger(*(ref_state + state))
                res['meets_reference_bars'] = ledger.meets_reference_bars(res['parity_ledger_vs_f32'], pair=True)
                other_legs[other] = res

    # informational, outside the timed region and never `value`: the same step fed from HOST memory --
    # raw uint8 frames uploaded over PCIe, preprocessed on the GPU, results copied back
    pcie_rate = None
    pcie_pipelined = None
    host_frames_resident = None
    host_fed_detections = None
    if extras and not args.no_host_fed:
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
        n_it = 10 if args.dtype != 'f32' else 3
        for _ in range(n_it):
            model.predict_on_frames(frames, P_host, planes_host)
        torch.cuda.synchronize()
        pcie_rate = round(B * n_it / (time.perf_counter() - t1), 1)
        from keras_retinanet_3D.utils.pipeline import FramePipeline
        pipe = FramePipeline(model)
        list(pipe.run(iter([(frames, P_host, planes_host)] * 4)))
        torch.cuda.synchronize()
        n_it = 40 if args.dtype != 'f32' else 16               # (a short run of a deep pipeline ends in a burst of results: not a rate)
        stamps = []
        for _ in pipe.run(iter([(frames, P_host, planes_host)] * n_it)):
            stamps.append(time.perf_counter())
        skip = 3                                     # steady state: results per second between the 4th and the last batch
        pcie_pipelined = round(B * (n_it - 1 - skip) / (stamps[-1] - stamps[skip]), 1)
        # the plan alone on the SAME frames, uploaded and preprocessed once (these frames carry more candidates than the timed
        # steps' frames, so `value` is not the like-for-like denominator of the streaming rate)
        hplan, _ = model.stage_frames(torch.as_tensor(frames).cuda(), torch.as_tensor(P_host).cuda(), torch.as_tensor(planes_host).cuda())
        for _ in range(3):
            model.run_plan(hplan)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_it):
            model.run_plan(hplan)
        torch.cuda.synchronize()
        host_frames_resident = round(B * n_it / (time.perf_counter() - t1), 1)

    # the reference's own timer (bin/run_network.py:108-111): ONE synchronous batch-1 predict_on_batch, upload and fetch inside the bracket
    b1 = None
    if extras and not args.no_b1:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import b1_latency
        b1 = b1_latency.measure(model, planes, n=40)
        b1['plan'] = model.plan_mode
        if model.plan_mode == 'throughput':
            # ... and with the plan a caller who times one image per call would load (models.load_model(..., plan='latency'): more layers split
            # their K loop -- a rule of (layer, plan mode); same bytes at every batch size within the mode, parity bars met: tests/test_latency_plan_gpu.py)
            from keras_retinanet_3D import models as _models
            lat_model = _models.load_model('synthetic:1234', backbone_name=args.backbone, dtype=args.dtype, plan='latency')
            b1_lat = b1_latency.measure(lat_model, planes, n=40)
            b1_lat['plan'] = 'latency'
            b1 = dict(b1_lat, default_plan={k: b1[k] for k in ('sync_ms_median', 'sync_ms_p90', 'plan_only_ms_median', 'stages_ms', 'launches', 'floor_ms', 'plan_over_floor', 'plan')})
            del lat_model

    if rank == 0:
        total_images = world * B * args.steps
        reg_tile = getattr(plan, 'tuning', {}).get('pyramid_regression_1', (0, 0.0))[0]
        rec = {
            'metric': 'images/sec end-to-end ({}, {} planes, 1242x375)'.format(args.backbone, args.planes),
            'value': round(total_images / elapsed, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'batch={} synthetic 1242x375 frames per GPU (network input 402x1333, resident in HBM), '
                                   '{} + FPN + heads + decode/NMS + polling, {}-plane database ({} planes), seeded random weights'.format(
                                       B, args.backbone, args.planes, planes.shape[0]),
                       'arithmetic': {'bf16': 'bf16 storage + operands, float32 accumulation', 'f16': 'f16 storage + operands, float32 accumulation',
                                      'f32': 'float32 storage + operands + accumulation (the reference\'s floatx)',
                                      'bf16x3': 'float32 storage, every product as three bf16 matrix products (~2^-16 relative), float32 accumulation',
                                      'f16x3': 'float32 storage, every product as three IEEE-half matrix products (hi + lo halves, 11 + 11 bits: ~2^-22 '
                                               'relative; float32: 2^-24), float32 accumulation'}[args.dtype] +
                                     '; decode / NMS / polling float32 + int32',
                       'global_batch': world * B, 'parallelism': 'dp{} image shards, one all_gather of (B,100,35) f32'.format(world),
                       'rccl_world_size': rccl_world, 'gathered_images_per_step': gathered_images,
                       'candidates_per_image': [int(c) for c in counts], 'detections_rank0': dets,
                       'algorithmic_gflop_per_image': round(plan.flops / B / 1e9, 1),
                       'achieved_tflops_whole_path': round(plan.flops * args.steps / elapsed / 1e12, 1),
                       'frac_of_mfma_peak_whole_path': round(plan.flops * args.steps / elapsed / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                       'f32_images_per_s': None if f32_leg is None else f32_leg['images_per_s'],
                       'f32_ms_per_step': None if f32_leg is None else f32_leg['ms_per_step'],
                       'f32_achieved_tflops_whole_path': None if f32_leg is None else f32_leg['achieved_tflops_whole_path'],
                       'f32_frac_of_f32_mfma_peak': None if f32_leg is None else f32_leg['frac_of_mfma_peak_whole_path'],
                       'parity_ledger': parity,
                       'parity_ledger_vs_f64_oracle': oracle_ledgers.get('f64'),
                       'parity_ledger_vs_f32_cpu_oracle': oracle_ledgers.get('f32'),
                       'parity_bars': dict(ledger.REFERENCE_BARS, what='the type under test against the float64 CPU oracle of the same frames (the exact '
                                           'value of what the reference graph computes; committed fixture): identical detection sets up to ties at the '
                                           'top-k cut (integer counts), identical orientation and plane index for every detection, 3-D corners within '
                                           '1e-3 m (BASELINE.json north_star) for the detections whose keypoints lie within 100 m of the camera (at least '
                                           'one must), within 1e-3 m x (distance / 100 m)^2 beyond (the condition number of a ray-plane intersection '
                                           'grows with the square of the distance); AND against the float32 HIP path of the same run at twice the metre '
                                           'bars (two float32-grade evaluations: the float32 CPU oracle itself is 1.15e-3 m from the float64 one, '
                                           'utils/ledger.py)'),
                       'parity_bars_met': bars_met,
                       # dtype='f16x3': range events every plan of this model counted over the WHOLE run (timed loop included: run_plan callers do not fetch
                       # through model.fetch, which would have reacted) -- an activation beyond +-65504 would have been clamped; must be 0
                       'f16x3_range_events_in_run': model.x3_range_events() if args.dtype == 'f16x3' else None,
                       'resident_batches_rotated': RESIDENT_BATCHES,
                       'side_stream_launches': dict(getattr(plan, 'side_lanes', {}), decode=bool(getattr(plan, 'decode_overlap', False))),
                       'multi_gpu_diagnosis': per_rank,
                       'step_contents_note': 'a 1-GPU step is the plan alone (its eight result arrays are the plan\'s output buffers); an N > 1 step adds '
                                             'one pack launch (gpp_pack_detections, ~5 us) and the asynchronous all_gather of (B,100,35) per rank',
                       'repeat_images_per_s': repeat_rates or None,
                       'repeat_spread_pct': None if not repeat_rates else round(100.0 * (max(repeat_rates + [round(total_images / elapsed, 2)]) -
                                                                                          min(repeat_rates + [round(total_images / elapsed, 2)])) /
                                                                                (total_images / elapsed), 2),
                       'b1': b1,
                       'other_types_same_frames': other_legs or None,
                       'host_fed_synchronous_images_per_s_incl_pcie_and_gpu_preprocessing': pcie_rate,
                       'host_fed_streaming_images_per_s': pcie_pipelined,
                       'host_fed_streaming_same_frames_resident_images_per_s': host_frames_resident,
                       'host_fed_streaming_fraction_of_resident': None if not (pcie_pipelined and host_frames_resident) else round(pcie_pipelined / host_frames_resident, 4),
                       'host_fed_note': 'synchronous = one predict_on_frames call after the other at batch {} (upload of uint8 frames, GPU preprocessing, '
                                        'plan, fetch): the closest analogue of the reference\'s timer (bin/run_network.py:108-111 brackets ONE synchronous '
                                        'batch-1 predict_on_batch on a preprocessed float32 image); streaming = depth-4 FramePipeline (uploads, compute and '
                                        'one packed (B,100,35) fetch overlapped), not what the reference times'.format(B),
                       'host_fed_detections': host_fed_detections,
                       'polling_kernel': polling},
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 1), 'peak': PEAK_TFLOPS[args.dtype], 'unit': 'TFLOP/s',
                         'frac': round(achieved / PEAK_TFLOPS[args.dtype], 4), 'traffic': traffic,
                         'traffic_unit': 'MB per launch at the L2<->fabric interface (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes; '
                                         'algorithmic {} MB for {})'.format(ALGORITHMIC_MB[args.dtype], args.dtype),
                         'algorithmic_mb_per_launch': ALGORITHMIC_MB[args.dtype],
                         'kernel': 'conv_igemm_kernel<{}> tile {} on pyramid_regression_1..3 (3x3, 512->512, 5 levels, M={})'.format(
                             args.dtype, ' / '.join(sorted(set(tile_name(getattr(plan, 'tuning', {}).get('pyramid_regression_{}'.format(i), (0, 0.0))[0])
                                                               for i in (1, 2, 3)))), B * (plan.n_anchors // 12)),
                         'gflop_per_launch': round(flops_per_launch / 1e9, 1), 'mean_launch_ms': round(mean_ms, 4),
                         'launches_timed': len(durations), 'timed_on_steps': 'every {}rd of the {} timed steps'.format(EVENT_EVERY, args.steps),
                         'frac_per_repeat': repeat_fracs or None,
                         'library': version},
        }
        if args.dtype in PIPE_ON_RANDOM_DATA_TFLOPS:
            rec['roofline']['pipe_on_random_data'] = round(PIPE_ON_RANDOM_DATA_TFLOPS[args.dtype], 1)
            rec['roofline']['frac_of_pipe_on_random_data'] = round(achieved / PIPE_ON_RANDOM_DATA_TFLOPS[args.dtype], 4)
            rec['roofline']['pipe_on_post_relu_data'] = round(PIPE_ON_POST_RELU_DATA_TFLOPS[args.dtype], 1)       # (what this layer reads)
            rec['roofline']['frac_of_pipe_on_post_relu_data'] = round(achieved / PIPE_ON_POST_RELU_DATA_TFLOPS[args.dtype], 4)
            rec['roofline']['pipe_on_random_data_note'] = ('register-only MFMA loops on random operands, every CU busy (tools/micro/mfma_power.hip, '
                                                          'profiles/r3/mfma_power.txt; measured once, not by this run): `peak` is the nominal dense figure, '
                                                          'which the pipe reaches on all-zero operands only')
        if traffic_note:
            rec['roofline']['traffic_note'] = traffic_note
        if device_state:
            # what the driver reported for rank 0's card during the timed region: `peak` is the figure at the nominal 2400 MHz; the matrix pipe's peak
            # at the clock the power management actually granted is peak x sclk / 2400.  The median is over the WHOLE step; under the dominant kernel alone the
            # clock is lower still (tools/clock_under_load.py: 2006 MHz at 1395 W of 1400 on real data, 2398 MHz on zeros -- profiles/r4/clock_under_load.txt)
            rec['roofline']['device_during_timed_region'] = device_state
            if device_state.get('sclk_mhz_median'):
                rec['roofline']['frac_at_the_steps_median_clock'] = round(achieved / (PEAK_TFLOPS[args.dtype] * device_state['sclk_mhz_median'] / 2400.0), 4)
        if world == 1 and not args.no_cpu_baseline:
            rec['cpu_baseline'], exact = cpu_baseline(args.cpu_images, args.backbone, planes, replay)
            rec['config']['gpu_decode_polling_replay_bit_exact'] = exact
            if exact is False:
                print(json.dumps(rec))
                raise SystemExit('decode / polling of the GPU head tensors differ from the oracle replay')
        print(json.dumps(rec))
        headline = args.dtype == 'f16x3' and args.backbone == 'resnet50' and args.planes == '1k' and B == 8
        if rec['config'].get('f16x3_range_events_in_run'):
            raise SystemExit('dtype f16x3: {} range events during the run: the timed results were clamped'.format(rec['config']['f16x3_range_events_in_run']))
        if bars_met is False and headline:           # (other configurations report parity_bars_met and carry on)
            raise SystemExit('the headline type {} misses a reference-precision bar: {}'.format(args.dtype, parity))
    for e in events:
        lib.gpp_event_destroy(e)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
