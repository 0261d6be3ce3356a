#!/usr/bin/env python
"""
Distribution of the end-to-end deviation of the HIP path from the CPU oracle at the BASELINE sizes, over the frames of the committed
full-size fixtures (tests/golden/fullsize_*.npz, made by oracle/gen_fullsize_goldens.py): not a maximum over 8 frames, the whole
distribution over 64 (resnet50, 1k planes) / 32 (resnet101, 10k) / 32 (resnet152, 22k) frames.

For every arithmetic mode asked for (default f32 and f16x3) and each of the two oracle precisions

    f64   the conv stack in float64 = the exact value the reference's float32 graph approximates
    f32   the conv stack in float32 on the CPU = one float32 evaluation of the reference graph (another summation order than ours)

it reports the ledger of utils/ledger.py plus p50 / p90 / p99 / max of the 3-D corner deviation of the detections within 100 m, by
distance bin (the condition number of the ray-plane intersection grows with the square of the distance), the distance-scaled
deviation beyond 100 m, and -- first row -- the same numbers for the f32 ORACLE against the f64 oracle: how far float32 itself is from
the exact result.  A mode is "as good as float32" when its row against f64 is no worse than that one.

    python tools/corner_deviation.py [--config resnet50_1k] [--dtypes f32,f16x3] [--frames N] [--json out.json]
Needs the GPU (the HIP path) but not the CPU oracle: the oracle's results are the fixtures.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
sys.path.insert(0, ROOT)

KEYS = ('boxes', 'dimensions', 'scores', 'labels', 'orientations', 'keypoints', 'keyplanes', 'residuals')


def load_golden(config, precision, frames=None):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'fullsize_{}_{}.npz'.format(config, precision)))
    n = len(g['frames']) if frames is None else min(frames, len(g['frames']))
    return [g[k][:n] for k in KEYS], g['anchor_index'][:n], g['plane_index'][:n], {k: g[k][:n] for k in ('nms_scores_120', 'candidates')}


def distribution(detail):
    d = np.array([(r, c, k, s) for _, _, r, c, k, s in detail], dtype=np.float64).reshape(-1, 4)
    same = d[:, 3] > 0
    near = d[same & (d[:, 0] <= 100.0)]
    far = d[same & ~(d[:, 0] <= 100.0)]
    out = {'n_within_100m': int(len(near)), 'n_beyond_100m': int(len(far))}
    if len(near):
        p = np.percentile(near[:, 1], [50, 90, 99, 100])
        out.update({'corner_p50': p[0], 'corner_p90': p[1], 'corner_p99': p[2], 'corner_max': p[3],
                    'corner_above_1e-3': int((near[:, 1] > 1e-3).sum())})
        bins = {}
        for lo, hi in ((0, 10), (10, 25), (25, 50), (50, 100)):
            m = (near[:, 0] > lo) & (near[:, 0] <= hi)
            if m.any():
                bins['{}-{}m'.format(lo, hi)] = {'n': int(m.sum()), 'p50': float(np.median(near[m, 1])), 'max': float(near[m, 1].max())}
        out['by_distance'] = bins
    if len(far):
        with np.errstate(all='ignore'):
            sc = np.where(np.isfinite(far[:, 0]), far[:, 1] / (far[:, 0] / 100.0) ** 2, np.where(far[:, 1] == 0, 0.0, np.inf))
        p = np.percentile(sc, [50, 99, 100])
        out.update({'scaled_beyond_p50': p[0], 'scaled_beyond_p99': p[1], 'scaled_beyond_max': p[2]})
    return out


def compare(ref, got, ledger):
    detail = []
    led = ledger.parity_ledger(ref[0], ref[1], ref[2], got[0], got[1], got[2], detail=detail)
    led['distribution'] = distribution(detail)
    led['plane_difference_list'] = [{'frame': int(i), 'anchor': int(a), 'planes': [int(ref[2][i][list(ref[1][i]).index(a)]), int(got[2][i][list(got[1][i]).index(a)])]}
                                    for i, a, _, _, _, s in detail if not s]
    led['meets_reference_bars'] = ledger.meets_reference_bars(led)
    return led


def set_differences(ref, got, extra):
    """ which detections one run has and the other has not, with what the f64 fixture says about how close the call was """
    out = []
    for b in range(ref[1].shape[0]):
        a, g = set(int(x) for x in ref[1][b] if x >= 0), set(int(x) for x in got[1][b] if x >= 0)
        if a != g:
            s = extra['nms_scores_120'][b]
            out.append({'frame': b, 'only_oracle': sorted(a - g), 'only_hip': sorted(g - a),
                        'oracle_score_rank100': float(s[99]), 'oracle_score_rank101': float(s[100]), 'gap_at_the_cut': float(s[99] - s[100]),
                        'oracle_scores_of_only_oracle': [float(ref[0][2][b][list(ref[1][b]).index(x)]) for x in sorted(a - g)]})
    return out


def fmt(x):
    return '{:.2e}'.format(x) if isinstance(x, float) else str(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='resnet50_1k', choices=['resnet50_1k', 'resnet101_10k', 'resnet152_22k'])
    ap.add_argument('--dtypes', default='f32,f16x3')
    ap.add_argument('--frames', type=int, default=None)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    import torch
    from keras_retinanet_3D import models
    from keras_retinanet_3D.utils import ledger, synthetic
    backbone, db = args.config.split('_')
    g64, g32 = load_golden(args.config, 'f64', args.frames), load_golden(args.config, 'f32', args.frames)
    n = g64[1].shape[0]
    planes = synthetic.load_plane_database(db).astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    report = {'config': args.config, 'frames': int(n), 'rows': {}}
    report['rows']['f32 CPU oracle vs f64 oracle'] = compare(g64, g32, ledger)
    for dtype in args.dtypes.split(','):
        model = models.load_model('synthetic:1234', backbone_name=backbone, dtype=dtype)
        if dtype == 'f16x3':
            model.x3_range_events(reset=True)
        outs, aidx, pidx = [], [], []
        for f0 in range(0, n, args.batch):
            seeds = list(range(f0, min(n, f0 + args.batch)))
            img = synthetic.synthetic_network_input(seeds)
            B = len(seeds)
            o = model.predict_on_batch([img, np.tile(P_inv[None].astype(np.float32), (B, 1, 1)), np.tile(planes[None], (B, 1, 1))])
            plan = model.plan_for(B, 402, 1333, planes.shape[0], True)
            outs.append(o)
            aidx.append(plan.anchor_index.cpu().numpy())
            pidx.append(plan.best_index.cpu().numpy())
        got = ([np.concatenate([o[k] for o in outs]) for k in range(8)], np.concatenate(aidx), np.concatenate(pidx))
        for name, ref in (('f64 oracle', g64), ('f32 CPU oracle', g32)):
            row = compare(ref, got, ledger)
            row['set_difference_list'] = set_differences(ref, got, ref[3])
            report['rows']['{} HIP vs {}'.format(dtype, name)] = row
        if dtype == 'f16x3':
            report['f16x3_range_events'] = model.x3_range_events()
        report['library'] = __import__('keras_retinanet_3D.backend.hip', fromlist=['lib']).lib().gpp_version().decode()
        del model
        torch.cuda.empty_cache()
    print('{}: {} frames, {} planes; library {}'.format(args.config, n, planes.shape[0], report.get('library')))
    head = ('run', 'dets', 'set', 'plane', 'n<=100m', 'p50', 'p90', 'p99', 'max', '>1e-3', 'n>100m', 'scaled max', 'bars')
    print(' | '.join(head))
    for name, r in report['rows'].items():
        d = r['distribution']
        print(' | '.join(fmt(v) for v in (name, '{}/{}'.format(r['common'], r['union']), r['detection_set_agreement'], '{}/{}'.format(r['same_plane'], r['common']),
                                          d.get('n_within_100m'), d.get('corner_p50', 0.0), d.get('corner_p90', 0.0), d.get('corner_p99', 0.0),
                                          d.get('corner_max', 0.0), d.get('corner_above_1e-3', 0), d.get('n_beyond_100m'), d.get('scaled_beyond_max', 0.0),
                                          r['meets_reference_bars'])))
        for k, v in d.get('by_distance', {}).items():
            print('      {:>8s}: n {:5d}  p50 {:.2e}  max {:.2e}'.format(k, v['n'], v['p50'], v['max']))
        for s in r.get('set_difference_list', []):
            print('      set difference:', s)
        for s in r.get('plane_difference_list', []):
            print('      plane difference ({} of {} with equal polling inputs):'.format(r['plane_differences_with_equal_inputs'], r['plane_differences']), s)
    if 'f16x3_range_events' in report:
        print('f16x3 range events (values outside the half range stored by an epilogue):', report['f16x3_range_events'])
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(report, f, indent=1, default=float)


if __name__ == '__main__':
    main()
