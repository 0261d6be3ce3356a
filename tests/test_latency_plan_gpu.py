"""
models.load_model(..., plan='latency'): the plan for callers that time ONE image per call -- the reference's own timer,
/root/reference/keras_retinanet_3D/bin/run_network.py:108-111.  More layers split their K loop (layers/conv.latency_split: a rule of the
layer and the plan mode, never of the batch), so

  * within the mode a model returns the same BYTES for an image whatever batch it arrives in (tests/test_zz_sharded_gpu.py runs the
    two-process form of this in both modes),
  * against the float64 oracle fixtures the mode meets the same parity bars as the default plan (utils/ledger.REFERENCE_BARS), and
  * against the default plan it differs by float32 summation order only: same detections, same planes, corners within the pair bars.
"""
import numpy as np
import pytest

from keras_retinanet_3D import models
from keras_retinanet_3D.layers import conv as C
from keras_retinanet_3D.utils import ledger, synthetic

pytestmark = pytest.mark.gpu


def inputs(frames, db='1k'):
    planes = synthetic.load_plane_database(db).astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    n = len(frames)
    return [synthetic.synthetic_network_input(frames), np.tile(P_inv[None].astype(np.float32), (n, 1, 1)), np.tile(planes[None], (n, 1, 1))]


def test_latency_plan_splits_more_layers_and_keeps_an_images_bytes_at_every_batch_size():
    model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype='f16x3', plan='latency')
    default = models.load_model('synthetic:1234', backbone_name='resnet50', dtype='f16x3')
    assert model.plan_mode == 'latency' and default.plan_mode == 'throughput'
    three = model.predict_on_batch(inputs([0, 1, 2]))
    plan = model.plan_for(3, 402, 1333, 1000, True)
    splits = {name: int(desc.split_k) for kind, _, desc, name, _ in plan.ops if hasattr(desc, 'split_k')}
    assert splits['res4b_branch2b'] > 1 and splits['P4'] > 1 and splits['res5b_branch2a'] > 1 and splits['pyramid_regression_1'] == 1, splits
    dplan = default.plan_for(1, 402, 1333, 1000, True)
    dsplits = {name: int(desc.split_k) for kind, _, desc, name, _ in dplan.ops if hasattr(desc, 'split_k')}
    assert dsplits['res4b_branch2b'] == 0 and dsplits['P4'] == 0                       # (0 = the library's own rule)
    for k in range(3):
        alone = model.predict_on_batch(inputs([k]))
        for a, b in zip(alone, three):
            assert a.tobytes() == b[k:k + 1].tobytes()
    # against the default plan: the same detections and planes, corners at float32 summation-order noise
    base = default.predict_on_batch(inputs([0, 1, 2]))
    bplan = default.plan_for(3, 402, 1333, 1000, True)
    led = ledger.parity_ledger(base, bplan.anchor_index.cpu().numpy(), bplan.best_index.cpu().numpy(), three, plan.anchor_index.cpu().numpy(),
                               plan.best_index.cpu().numpy())
    print('latency plan vs default plan: {}'.format({k: led[k] for k in ('common', 'union', 'same_plane', 'max_corner_dev_m_within_100m')}))
    assert ledger.meets_reference_bars(led, pair=True), led
    assert model.x3_range_events() == 0


def test_latency_plan_meets_the_reference_bars_against_the_float64_fixture():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import corner_deviation as CD
    frames = 8
    g64 = CD.load_golden('resnet50_1k', 'f64', frames)
    model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype='f16x3', plan='latency')
    outs, aidx, pidx = [], [], []
    for f in range(frames):                                              # ONE image per call: what the mode is for
        outs.append(model.predict_on_batch(inputs([f])))
        plan = model.plan_for(1, 402, 1333, 1000, True)
        aidx.append(plan.anchor_index.cpu().numpy())
        pidx.append(plan.best_index.cpu().numpy())
    got = ([np.concatenate([o[k] for o in outs]) for k in range(8)], np.concatenate(aidx), np.concatenate(pidx))
    exact = CD.compare(g64, got, ledger)
    print('LEDGER latency plan resnet50_1k f16x3 {} frames (batch 1) vs f64: {}/{} common, corners max {:.2e}, beyond 100 m scaled {:.2e} (bars 1e-3)'.format(
        frames, exact['common'], exact['union'], exact['max_corner_dev_m_within_100m'], exact['max_corner_dev_scaled_beyond_100m']))
    assert exact['detections'] == exact['detections_ref'] == 100 * frames
    assert ledger.meets_reference_bars(exact), exact
    assert model.x3_range_events() == 0 and model.range_fallbacks == 0
