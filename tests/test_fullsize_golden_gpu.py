"""
The HIP path against the CPU ORACLE at the BASELINE sizes, for every backbone / plane database BASELINE.json names, in the
reference-precision type (f32) and in the headline type (f16x3):

    resnet50  + 1k  planes, 64 frames (8 batches of 8)     configs[1] / [2]
    resnet101 + 10k planes, 32 frames (4 batches of 8)      configs[3]
    resnet152 + 22k planes, 32 frames (8 batches of 4)      configs[4]'s per-GPU share

The oracle's side is DATA: tests/golden/fullsize_<backbone>_<db>_{f32,f64}.npz hold what oracle/net_torch.py + decode_np.py + polling.c
return for those frames (oracle/gen_fullsize_goldens.py; 0.5 TFLOP per frame and precision: minutes of host time, too slow to repeat in
every GPU run).  f64 = the conv stack in float64 = the exact value of what the reference's float32 graph computes; f32 = one float32
CPU evaluation of it.  What is asserted (utils/ledger.py has the reasoning and the measured numbers behind every bar):

  * against the f64 oracle: the same detections up to ties at the top-k cut, the same orientation for EVERY common detection, the same
    plane index for every one but at most 1 per 1000 whose polling inputs agree to float32 noise (a vote at its threshold), 3-D corners within 1e-3 m inside 100 m and within 1e-3 m x (r / 100 m)^2 beyond -- ledger.REFERENCE_BARS, for f16x3;
    the float32 HIP path is held to 1.5e-3: float32 itself is that far from the exact value (the float32 CPU oracle: 1.15e-3)
  * "as good as float32": the p50 / p90 / p99 of f16x3's corner deviation from f64 are no larger than 1.25 x those of the float32
    CPU oracle from f64 (computed from the two fixtures)
  * against the f32 CPU oracle: the pair bars (two float32-grade runs: 2e-3 m)
  * f16x3: no activation left the half range (gpp_x3_range_events == 0)
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from keras_retinanet_3D import models  # noqa: E402
from keras_retinanet_3D.utils import ledger, synthetic  # noqa: E402

import corner_deviation as CD  # noqa: E402  (tools/corner_deviation.py: fixture loading, per-detection distribution)

pytestmark = pytest.mark.gpu

CONFIGS = {'resnet50_1k': (64, 8), 'resnet101_10k': (32, 8), 'resnet152_22k': (32, 4)}      # frames, batch


# round 5: further weight draws per backbone (8 frames each; oracle/gen_fullsize_goldens.py --weights ...): another seed, and the 'trained'
# family of models/weights.trained_like -- per-channel scales three decades apart in the bottleneck maps, dead channels, a residual stream
# that grows 64-fold from res2 to res5.  Held to the SAME constants of utils/ledger.py as the draw the bars were fitted on.
DRAWS = {'s2024': 'synthetic:2024', 's1234t': 'synthetic:1234:trained'}
DRAW_FRAMES = 8
LEDGER_LINES = []         # what the ledger tests measured: printed once more at the end of the session (conftest.py), so that a log of dots shows margins


def run_hip(config, dtype, weights='synthetic:1234', frames=None):
    n_frames, batch = CONFIGS[config]
    frames = frames or n_frames
    batch = min(batch, frames)
    backbone, db = config.split('_')
    planes = synthetic.load_plane_database(db).astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    model = models.load_model(weights, backbone_name=backbone, dtype=dtype)
    if dtype == 'f16x3':
        model.x3_range_events(reset=True)
    outs, aidx, pidx = [], [], []
    for f0 in range(0, frames, batch):
        img = synthetic.synthetic_network_input(range(f0, f0 + batch))
        outs.append(model.predict_on_batch([img, np.tile(P_inv[None].astype(np.float32), (batch, 1, 1)), np.tile(planes[None], (batch, 1, 1))]))
        plan = model.plan_for(batch, 402, 1333, planes.shape[0], True)
        aidx.append(plan.anchor_index.cpu().numpy())
        pidx.append(plan.best_index.cpu().numpy())
    events = model.x3_range_events() if dtype == 'f16x3' else 0
    return ([np.concatenate([o[k] for o in outs]) for k in range(8)], np.concatenate(aidx), np.concatenate(pidx)), events


@pytest.mark.parametrize('frames', ['default', pytest.param(None, marks=pytest.mark.slow)], ids=['default', 'all'])
@pytest.mark.parametrize('dtype', ['f32', 'f16x3'])
@pytest.mark.parametrize('config', list(CONFIGS))
def test_hip_path_against_the_cpu_oracle_fixtures(config, dtype, frames):
    """ the default GPU run compares the first 32 frames of every fixture in the headline type (f16x3) and the first 16 in float32 (three times
    the time per frame); all 64 / 32 / 32 in both types run under --run-slow (tools/collect_r6.sh).  The ledger maxima are printed -- `pytest -rP`
    / the captured output of the driver's log shows the margins, not only dots. """
    if frames == 'default':
        frames = 32 if dtype == 'f16x3' else 16
    g64, g32 = CD.load_golden(config, 'f64', frames), CD.load_golden(config, 'f32', frames)
    n_frames = frames or CONFIGS[config][0]
    assert g64[1].shape[0] == n_frames
    got, events = run_hip(config, dtype, frames=frames)
    exact = CD.compare(g64, got, ledger)
    pair = CD.compare(g32, got, ledger)
    floor = CD.compare(g64, g32, ledger)                              # float32 itself (the CPU oracle) against the exact value
    de, df = exact['distribution'], floor['distribution']
    line = ('LEDGER {} {} {} frames vs f64: {}/{} common, ties {}, corners p50 {:.2e} p99 {:.2e} max {:.2e} (bar 1e-3 m), beyond 100 m scaled {:.2e}; '
            'float32 CPU oracle vs f64: p50 {:.2e} p99 {:.2e} max {:.2e}'.format(
                config, dtype, n_frames, exact['common'], exact['union'], exact['set_differences_at_a_tie'], de['corner_p50'], de['corner_p99'],
                de['corner_max'], exact['max_corner_dev_scaled_beyond_100m'], df['corner_p50'], df['corner_p99'], df['corner_max']))
    print(line)
    LEDGER_LINES.append(line)
    n = 100 * n_frames
    assert exact['detections_ref'] == exact['detections'] == n
    # the same detections (a difference must be a tie at the cut: the float64 fixture shows how close the 100th and 101st are)
    assert exact['set_differences_unexplained'] == 0 and exact['set_differences'] <= 2 * n_frames // 4, exact
    for s in CD.set_differences(g64, got, g64[3]):
        assert s['gap_at_the_cut'] <= ledger.TIE_EPS, s
    # the same orientation for every detection both report, and the same plane -- up to ledger.PLANE_FLIPS_PER_1000 detections whose polling
    # INPUTS agree to float32 noise (<= 1e-3 px) and whose plane still differs: a vote at its 0.7 m threshold (ledger.py; the float32 CPU
    # oracle has one such detection in 3199 against the f64 one on resnet152 / 22k planes, asserted below so that the allowance stays honest)
    flips = exact['plane_differences_with_equal_inputs']
    print('    plane differences: {} (with equal inputs {}); float32 CPU oracle vs f64: {}'.format(
        exact['plane_differences'], flips, floor['plane_differences']))
    assert exact['same_orientation'] == exact['common'] and exact['same_plane'] + flips == exact['common'], exact
    assert flips <= ledger.PLANE_FLIPS_PER_1000 * -(-exact['common'] // 1000), exact
    assert floor['plane_differences'] == floor['plane_differences_with_equal_inputs'] == (1 if config == 'resnet152_22k' else 0)      # (frame 9)
    # 3-D corners
    bar = 1.0e-3 if dtype == 'f16x3' else 1.5e-3
    assert exact['same_plane_within_100m'] >= 0.8 * exact['common']
    assert exact['max_corner_dev_m_within_100m'] <= bar and exact['max_corner_dev_scaled_beyond_100m'] <= bar, exact
    if dtype == 'f16x3':
        assert ledger.meets_reference_bars(exact), exact
        assert events == 0                                            # no activation left the half range
        for key in ('corner_p50', 'corner_p90', 'corner_p99'):        # as close to the exact value as float32 is
            assert de[key] <= 1.25 * df[key], (key, de[key], df[key])
    assert ledger.meets_reference_bars(pair, pair=True), pair
    assert pair['max_box_diff_px'] <= 1e-2 and pair['max_score_diff'] <= ledger.TIE_EPS


@pytest.mark.parametrize('draw', list(DRAWS))
@pytest.mark.parametrize('config', list(CONFIGS))
def test_the_bars_hold_on_weight_draws_they_were_not_fitted_on(config, draw):
    """ f16x3 HIP against the float64 oracle of the same frames on two further weight families per backbone, utils/ledger.py unchanged """
    g64, g32 = CD.load_golden('{}_{}'.format(config, draw), 'f64'), CD.load_golden('{}_{}'.format(config, draw), 'f32')
    assert g64[1].shape[0] == DRAW_FRAMES
    got, events = run_hip(config, 'f16x3', DRAWS[draw], DRAW_FRAMES)
    exact = CD.compare(g64, got, ledger)
    floor = CD.compare(g64, g32, ledger)
    pair = CD.compare(g32, got, ledger)
    de, df = exact['distribution'], floor['distribution']
    print('{} {} f16x3 vs f64: {}/{} common, ties {}, plane flips {}, corners p50 {:.2e} p99 {:.2e} max {:.2e}, beyond 100 m scaled {:.2e}; '
          'float32 CPU oracle vs f64: {}/{} common, p50 {:.2e} p99 {:.2e} max {:.2e}, scaled {:.2e}; range events {}'.format(
              config, draw, exact['common'], exact['union'], exact['set_differences_at_a_tie'], exact['plane_differences_with_equal_inputs'],
              de.get('corner_p50', 0), de.get('corner_p99', 0), de.get('corner_max', 0), exact['max_corner_dev_scaled_beyond_100m'],
              floor['common'], floor['union'], df.get('corner_p50', 0), df.get('corner_p99', 0), df.get('corner_max', 0),
              floor['max_corner_dev_scaled_beyond_100m'], events))
    LEDGER_LINES.append('LEDGER {} {} f16x3 {} frames vs f64: corners max {:.2e}, beyond 100 m scaled {:.2e} (bars 1e-3); float32 CPU oracle vs f64: max {:.2e}, scaled {:.2e}'.format(
        config, draw, DRAW_FRAMES, de.get('corner_max', 0), exact['max_corner_dev_scaled_beyond_100m'], df.get('corner_max', 0), floor['max_corner_dev_scaled_beyond_100m']))
    assert exact['detections_ref'] == exact['detections'] > 0
    assert events == 0
    assert ledger.meets_reference_bars(exact), exact
    assert ledger.meets_reference_bars(pair, pair=True), pair
    for key in ('corner_p50', 'corner_p90', 'corner_p99'):            # as close to the exact value as float32 itself is
        assert de[key] <= 1.25 * df[key], (key, de[key], df[key])


def test_config_1_shape_one_frame_ten_planes_against_the_oracle(oracle_lib):
    """ BASELINE.json configs[0]: ONE 1242x375 frame (402x1333 network input), resnet50, road_planes_database_10 -- the call
    bin/run_network.py:108-111 times -- through the HIP path at batch 1 in the headline type, against the float64 oracle's detections of
    the same frame (the committed fixture) polled by oracle/polling.c over the 10-plane database. """
    import helpers
    g64 = CD.load_golden('resnet50_1k', 'f64', 2)
    planes = synthetic.load_plane_database('10').astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    P_inv = P_inv[None].astype(np.float32)
    model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype='f16x3')
    model.x3_range_events(reset=True)
    for frame in (0, 1):
        det = [g64[0][k][frame:frame + 1] for k in range(5)]
        kp, kpl, res, idx = helpers.c_oracle_poll(oracle_lib, det[0], det[1], det[4], P_inv, planes[None])
        out = model.predict_on_batch([synthetic.synthetic_network_input([frame]), P_inv, planes[None]])
        plan = model.plan_for(1, 402, 1333, 10, True)
        assert [o.shape for o in out] == [(1, 100, 12), (1, 100, 3), (1, 100), (1, 100), (1, 100), (1, 100, 4, 3), (1, 100, 1, 4), (1, 100)]
        led = ledger.parity_ledger(det + [kp, kpl, res], g64[1][frame:frame + 1], idx, out, plan.anchor_index.cpu().numpy(), plan.best_index.cpu().numpy())
        assert led['common'] == led['union'] == 100 and ledger.meets_reference_bars(led), led
        assert 0 <= int(plan.best_index.max()) < 10
    assert model.x3_range_events() == 0 and model.range_fallbacks == 0
