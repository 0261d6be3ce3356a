#!/usr/bin/env python
"""
ORACLE fixtures at the BASELINE sizes (test infrastructure; the script that made tests/golden/fullsize_*.npz).

What the CPU oracle returns for whole 402x1333 frames -- conv stack (oracle/net_torch.py), decode / NMS / top-k
(oracle/decode_np.py), polling (oracle/polling.c) -- for every backbone / plane database BASELINE.json's configs name:

    resnet50  + 1k  planes   frames 0..63   (frames 0..7 = rank 0's batch 0 of bench.py)     configs[1], configs[2]
    resnet101 + 10k planes   frames 0..31   (config 4's batch is the first 8)                configs[3]
    resnet152 + 22k planes   frames 0..31   (config 5's per-GPU share is the first 4)        configs[4]

each in TWO precisions of the conv stack:
    f32   float32 throughout, literal BatchNormalization: the reference's floatx graph (models/retinanet.py:395-422)
    f64   the same graph in float64, head tensors rounded to float32 once: the exact value the float32 graph approximates.
          Decode and polling are float32 op by op in both (they are bit-exact stages of the path, tests/golden/decode_*, polling_*).
The f64 set is the yardstick: an arithmetic mode of the HIP path "is as good as float32" when its detections deviate from the f64
set no more than the f32 set does (tests/test_fullsize_golden_gpu.py, tools/corner_deviation.py).

The whole conv stack costs 0.5-0.65 TFLOP per frame: too slow to recompute on the GPU box's host in every test run, cheap to keep as
data (14 KB per frame: the 8 output arrays of predict_on_batch + anchor ids + plane indices, + the scores of the first 120 survivors
of the NMS, which show how close the 100th / 101st are).  Frames and weights are seeded (utils/synthetic.synthetic_network_input,
models/weights.synthetic_weights(backbone, 1234)): nothing of /root/reference is read.

Round 5: further weight draws (`--weights synthetic:2024`, `--weights synthetic:1234:trained`, 8 frames each: files
fullsize_<backbone>_<db>_<tag>_{f32,f64}.npz with tag = s<seed>[t]): another seed, and the 'trained' family of models/weights.trained_like
(per-channel scales three decades apart, dead channels, a residual stream growing stage by stage).  The bars the GPU tests hold the HIP
path to (utils/ledger.py) were fitted on seed 1234 alone; these fixtures are what tells a property of the arithmetic from a property
of one draw.

    python oracle/gen_fullsize_goldens.py [--only resnet50] [--frames N] [--threads T] [--weights synthetic:<seed>[:trained]]
"""
import argparse
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
sys.path.insert(0, ROOT)

CONFIGS = [('resnet50', '1k', 64), ('resnet101', '10k', 32), ('resnet152', '22k', 32)]
H, W = 402, 1333


def poll(lib, det, P_inv, planes):
    n = det[0].shape[0]
    kp = np.empty((n, 100, 4, 3), np.float32)
    kpl = np.empty((n, 100, 1, 4), np.float32)
    res = np.empty((n, 100), np.float32)
    idx = np.empty((n, 100), np.int32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    pinv = np.ascontiguousarray(np.tile(P_inv[None], (n, 1, 1)), np.float32)
    rc = lib.gpp_oracle_poll_f32(ptr(np.ascontiguousarray(det[0])), ptr(np.ascontiguousarray(det[1])), ptr(np.ascontiguousarray(det[4])),
                                 ptr(pinv), ptr(planes), n, 100, planes.shape[0], 0, ctypes.c_float(0.7), ptr(kp), ptr(kpl), ptr(res), ptr(idx))
    assert rc == 0
    return kp, kpl, res, idx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--frames', type=int, default=None)
    ap.add_argument('--threads', type=int, default=None)
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    ap.add_argument('--weights', default='synthetic:1234')
    args = ap.parse_args()
    import torch
    if args.threads:
        torch.set_num_threads(args.threads)
    from oracle import decode_np, net_torch
    from keras_retinanet_3D.models import weights as Wt
    from keras_retinanet_3D.utils import synthetic
    lib_path = os.path.join(ROOT, 'oracle', 'liboracle_polling.so')
    if not os.path.isfile(lib_path):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(lib_path)
    anchors = decode_np.anchors_for_image((H, W))
    _, P_inv = synthetic.synthetic_calibration()
    P_inv = P_inv.astype(np.float32)
    for backbone, db, n_frames in CONFIGS:
        if args.only and backbone != args.only:
            continue
        n_frames = args.frames or n_frames
        planes = np.ascontiguousarray(synthetic.load_plane_database(db), np.float32)
        wseed, family = Wt.parse_synthetic(args.weights)
        tag = '' if (wseed, family) == (1234, 'he') else '_s{}{}'.format(wseed, 't' if family == 'trained' else '')
        weights = Wt.synthetic_weights(backbone, wseed, family)
        for precision in ('f32', 'f64'):
            net = net_torch.Net(weights, backbone, precision=precision)
            rows = {k: [] for k in ('boxes', 'dimensions', 'scores', 'labels', 'orientations', 'keypoints', 'keyplanes', 'residuals',
                                    'anchor_index', 'plane_index', 'nms_scores_120', 'candidates')}
            t0 = time.time()
            for seed in range(n_frames):
                f = net.forward(synthetic.synthetic_network_input([seed]))
                det, aidx = decode_np.detect(f['classification_logits'], f['regression'], f['regression_dim'], anchors)
                kp, kpl, res, idx = poll(lib, det, P_inv, planes)
                det120, _ = decode_np.detect(f['classification_logits'], f['regression'], f['regression_dim'], anchors, max_detections=120)
                score, _ = decode_np.fold_classification(decode_np.sigmoid(f['classification_logits'][0]))
                for k, v in zip(('boxes', 'dimensions', 'scores', 'labels', 'orientations'), det):
                    rows[k].append(v[0])
                for k, v in (('keypoints', kp), ('keyplanes', kpl), ('residuals', res), ('plane_index', idx), ('anchor_index', aidx.astype(np.int32)),
                             ('nms_scores_120', det120[2])):
                    rows[k].append(v[0])
                rows['candidates'].append(np.int32((score > np.float32(0.05)).sum()))
                print('{} {} {} frame {:2d}: {} detections, {} candidates, {:.0f} s'.format(
                    backbone, db, precision, seed, int((det[2][0] > 0.05).sum()), int(rows['candidates'][-1]), time.time() - t0), flush=True)
            path = os.path.join(args.out, 'fullsize_{}_{}{}_{}.npz'.format(backbone, db, tag, precision))
            np.savez_compressed(path, frames=np.arange(n_frames, dtype=np.int32), weights_seed=np.int32(wseed),
                                **{k: np.stack(v) for k, v in rows.items()})
            print('wrote', path, flush=True)


if __name__ == '__main__':
    main()
