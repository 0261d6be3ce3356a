"""
CPU tests of the host-side harness (utils.gpp_utils post-processing, utils.image, bin/run_network)
against golden outputs of the reference's own bin/run_network.py main() (executed on stubs by
oracle/gen_harness_goldens.py).  Tolerances: float32 fields 1e-4 absolute (angles go through an
SVD), KITTI text fields equal after %.2f formatting up to one unit in the last digit.
"""
import glob
import os

import numpy as np
import pytest
import scipy.io

import helpers
from oracle import pose_np
from keras_retinanet_3D.bin import run_network
from keras_retinanet_3D.utils import gpp_utils, image, synthetic

CASES = sorted(glob.glob(os.path.join(helpers.GOLDEN, 'harness_*.npz')))


def model_outputs(g):
    return [g['in_' + k] for k in ('boxes', 'dimensions', 'scores', 'labels', 'orientations', 'keypoints', 'keyplanes', 'residuals')]


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p) for p in CASES])
def test_postprocessing_matches_reference_harness(path):
    g = dict(np.load(path))
    det = gpp_utils.recover_pose(gpp_utils.select_detections(model_outputs(g), float(g['scale'])))
    n = g['mat_scores'].shape[1]
    assert len(det['scores']) == n and n < 23 + 1                 # scores <= 0.05 dropped
    assert np.allclose(det['boxes'][:, :4], g['mat_boxes'], atol=1e-4)
    assert np.allclose(det['boxes'][:, 4:], g['mat_keypoints'], atol=1e-4)
    assert np.array_equal(det['labels'], g['mat_labels'][0]) and np.allclose(det['scores'], g['mat_scores'][0])
    assert np.allclose(det['dimensions'], g['mat_dimensions'], atol=1e-4)      # h, l overwritten from keypoints
    assert np.allclose(det['locations'], g['mat_locations'], atol=1e-4)
    assert np.allclose(det['angles'], g['mat_angles'], atol=1e-4)
    assert np.allclose(det['residuals'], g['mat_residuals'][0], atol=1e-6)
    lines = gpp_utils.kitti_lines(det, tuple(g['image_shape']))
    want = str(g['kitti_text']).splitlines()
    assert len(lines) == len(want)
    for a, b in zip(lines, want):
        fa, fb = a.split(), b.split()
        assert fa[:3] == fb[:3] == ['Car', '-1', '-1']
        assert np.allclose([float(v) for v in fa[3:]], [float(v) for v in fb[3:]], atol=0.011)


def test_rodrigues_batch_equals_scalar_oracle_and_round_trips():
    rng = np.random.default_rng(0)
    vecs = rng.normal(size=(64, 3))
    vecs *= rng.uniform(0.01, 3.0, size=(64, 1)) / np.linalg.norm(vecs, axis=1, keepdims=True)   # |r| < pi: unique
    R = gpp_utils.rotation_matrix_from_vector(vecs)
    for k in range(64):
        assert np.allclose(R[k], pose_np.rodrigues(vecs[k])[0], atol=1e-12)
        assert np.allclose(pose_np.rodrigues(R[k])[0][:, 0], vecs[k], atol=1e-8)
    back = gpp_utils.rotation_vector_from_matrix(R)
    assert np.allclose(back, vecs, atol=1e-8)
    noisy = R + rng.normal(size=R.shape) * 1e-3               # not exactly orthonormal: SVD projection
    for k in range(8):
        assert np.allclose(gpp_utils.rotation_vector_from_matrix(noisy[k:k + 1])[0], pose_np.rodrigues(noisy[k])[0][:, 0], atol=1e-10)
    half_turn = np.diag([1.0, -1.0, -1.0])                    # angle pi: the degenerate branch
    assert np.allclose(gpp_utils.rotation_vector_from_matrix(half_turn[None])[0], pose_np.rodrigues(half_turn)[0][:, 0])
    assert np.allclose(gpp_utils.rotation_vector_from_matrix(np.eye(3)[None]), 0.0)


def test_pose_recovery_reconstructs_the_synthetic_cuboids():
    planes = synthetic.load_plane_database('100')
    P, P_inv = synthetic.synthetic_calibration(1.0)
    rng = np.random.default_rng(4)
    for _ in range(20):
        h, w, l = rng.uniform(1.4, 1.8), rng.uniform(1.5, 1.8), rng.uniform(3.5, 4.8)
        yaw, x, z = rng.uniform(-np.pi, np.pi), rng.uniform(-8, 8), rng.uniform(8, 40)
        corners, t = synthetic.cuboid_corners_on_plane(planes[3], x, z, yaw, h, w, l)
        o = synthetic.orientation_class(yaw, t)
        kp = corners[[c - 1 for c in synthetic.KEYPOINT_CORNERS[o]]].astype(np.float32)
        det = {'keypoints': kp.reshape(1, 12), 'orientations': np.array([o]), 'dimensions': np.array([[h, w, l]], np.float32),
               'boxes': np.zeros((1, 12), np.float32), 'scores': np.ones(1, np.float32)}
        det = gpp_utils.recover_pose(det)
        assert np.allclose(det['dimensions'][0], [h, w, l], atol=1e-3)
        assert np.allclose(det['locations'][0], t, atol=2e-3)                       # centre of the bottom face
        got = np.sort(gpp_utils.cuboid_corners(det)[0].T, axis=0)
        assert np.allclose(got, np.sort(corners, axis=0), atol=5e-3)


def test_image_preprocessing():
    raw = synthetic.synthetic_image(seed=3)
    x = image.preprocess_image(raw)
    assert x.dtype == np.float32 and np.allclose(x[0, 0], raw[0, 0].astype(np.float32) - np.array(image.IMAGENET_MEAN_BGR, np.float32))
    y, scale = image.resize_image(x)
    assert y.shape == (402, 1333, 3) and abs(scale - 1333.0 / 1242.0) < 1e-12
    assert image.compute_resize_scale((480, 640, 3)) == 800 / 480
    const = np.full((20, 30, 3), 7.5, np.float32)
    assert np.allclose(image.resize_bilinear(const, 1.7), 7.5)                       # interpolation preserves constants
    ramp = np.tile(np.arange(40, dtype=np.float32)[None, :, None], (10, 1, 3))
    up = image.resize_bilinear(ramp, 2.0)
    assert up.shape == (20, 80, 3) and np.allclose(np.diff(up[0, 1:-1, 0]), 0.5, atol=1e-5)


def test_run_network_cli_end_to_end_with_a_fake_model(tmp_path, monkeypatch):
    from PIL import Image
    g = dict(np.load(CASES[0]))
    (tmp_path / 'img').mkdir(); (tmp_path / 'calib').mkdir(); (tmp_path / 'out').mkdir()
    Image.fromarray(synthetic.synthetic_image(seed=0)[:, :, ::-1]).save(str(tmp_path / 'img' / '000001.png'))
    (tmp_path / 'calib' / '000001.txt').write_text(str(g['calib_text']))
    seen = {}

    class Fake(object):
        def predict_on_batch(self, inputs):
            seen['shapes'] = [np.asarray(i).shape for i in inputs]
            return [a.copy() for a in model_outputs(g)]

    monkeypatch.setattr(run_network.models, 'load_model', lambda *a, **k: Fake())
    (tmp_path / 'out' / 'mymodel').mkdir()                          # an existing output dir is replaced (:79-80)
    (tmp_path / 'out' / 'mymodel' / 'stale').write_text('x')
    run_network.main(['mymodel.h5', str(tmp_path / 'img'), str(tmp_path / 'calib'), synthetic.plane_database_path('100'),
                      str(tmp_path / 'out'), '--kitti'])
    assert seen['shapes'] == [(1, 402, 1333, 3), (1, 4, 3), (1, 100, 4)]
    assert not (tmp_path / 'out' / 'mymodel' / 'stale').exists()
    mat = scipy.io.loadmat(str(tmp_path / 'out' / 'mymodel' / 'outputs' / 'full' / '000001.mat'))
    for key in ('boxes', 'keypoints', 'labels', 'scores', 'locations', 'angles', 'dimensions', 'residuals'):
        assert np.allclose(mat[key], g['mat_' + key], atol=1e-4), key
    txt = (tmp_path / 'out' / 'mymodel' / 'outputs' / 'kitti' / '000001.txt').read_text()
    assert txt.count('\n') == str(g['kitti_text']).count('\n')


def test_pose_recovery_survives_a_degenerate_detection():
    """ a detection whose keypoints coincide (zero-length edge -> 0/0 axes) gives a NaN pose for THAT row only; the SVD of
    the other rows must not be poisoned (run_network's per-detection loop in the reference isolates rows the same way) """
    from keras_retinanet_3D.utils import gpp_utils
    rng = np.random.default_rng(0)
    n = 5
    det = {'keypoints': rng.normal(size=(n, 12)).astype(np.float32) * 5, 'orientations': np.array([0, 1, 2, 3, 0], np.int32),
           'dimensions': rng.uniform(1, 4, (n, 3)).astype(np.float32), 'boxes': rng.uniform(0, 100, (n, 12)).astype(np.float32),
           'scores': np.linspace(0.9, 0.5, n).astype(np.float32)}
    good = gpp_utils.recover_pose(det)
    assert np.isfinite(good['angles']).all() and np.isfinite(good['locations']).all()
    bad = {k: v.copy() for k, v in det.items()}
    bad['keypoints'][2] = 1.0                     # X_l = X_m = X_r = X_t
    with np.errstate(all='ignore'):
        out = gpp_utils.recover_pose(bad)
    assert not np.isfinite(out['angles'][2]).all()
    keep = [0, 1, 3, 4]
    assert np.array_equal(out['angles'][keep], good['angles'][keep]) and np.array_equal(out['locations'][keep], good['locations'][keep])
