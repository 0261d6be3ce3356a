"""
CPU tests of the polling oracle (no GPU): the NumPy and C restatements against the golden
vectors produced by the reference's own fit_road_planes.py (oracle/gen_polling_goldens.py),
against each other (bit for bit), and property checks of the selection rule.
"""
import numpy as np
import pytest

import helpers
from oracle import polling_np
from keras_retinanet_3D.utils import synthetic


@pytest.mark.parametrize('name', helpers.polling_golden_names())
def test_numpy_oracle_matches_reference_goldens(name):
    g = helpers.load_polling_golden(name)
    B = g['boxes'].shape[0]
    planes_b = np.tile(g['planes'][None], (B, 1, 1))
    kp, kpl, res, idx = polling_np.fit_road_planes(g['boxes'], g['dimensions'], g['orientations'], g['P_inv'],
                                                   planes_b, return_index=True)
    # plane index: exact on every row, padding rows included
    assert np.array_equal(idx, g['best_index'])
    # the canonical plane is a pure gather: exact
    assert np.array_equal(kpl, g['keyplanes'])
    valid = g['orientations'] >= 0
    # keypoints / residuals: the goldens went through BLAS matmul (fused multiply-add), the
    # restatement is operation by operation: agree to float32 rounding of 10-40 m coordinates
    assert np.abs(kp - g['keypoints'])[valid].max() <= 5e-4
    assert np.abs(res - g['residuals'])[valid].max() <= 1e-4
    assert kp.dtype == np.float32 and kpl.dtype == np.float32 and res.dtype == np.float32
    assert kp.shape == (B, 100, 4, 3) and kpl.shape == (B, 100, 1, 4) and res.shape == (B, 100)


@pytest.mark.parametrize('name', helpers.polling_golden_names())
def test_c_oracle_is_bit_identical_to_numpy_oracle(name, oracle_lib):
    g = helpers.load_polling_golden(name)
    B = g['boxes'].shape[0]
    ref = polling_np.fit_road_planes(g['boxes'], g['dimensions'], g['orientations'], g['P_inv'],
                                     np.tile(g['planes'][None], (B, 1, 1)), return_index=True)
    for planes in (g['planes'], np.tile(g['planes'][None], (B, 1, 1))):   # shared and batched databases
        got = helpers.c_oracle_poll(oracle_lib, g['boxes'], g['dimensions'], g['orientations'], g['P_inv'], planes)
        assert helpers.bits_equal(got[0], ref[0])
        assert helpers.bits_equal(got[1], ref[1])
        assert helpers.bits_equal(got[2], ref[2])
        assert np.array_equal(got[3], ref[3])


def test_goldens_cover_the_documented_edge_cases():
    g = helpers.load_polling_golden('allmasked')
    valid = g['orientations'] >= 0
    assert np.all(g['best_index'][valid] == 0)                      # every plane masked -> index 0
    assert np.allclose(g['residuals'][valid], 100.0 / 6.0)
    g = helpers.load_polling_golden('over100')
    valid = g['orientations'] >= 0
    assert np.allclose(g['residuals'][valid], 100.0 / 6.0)          # a masked plane beats residual > 100
    g = helpers.load_polling_golden('dup10k')
    valid = g['orientations'][0] >= 0
    planes = g['planes']
    chosen = planes[g['best_index'][0][valid]]
    # first-index rule: no earlier row of the database equals the chosen row
    for k, row in zip(g['best_index'][0][valid], chosen):
        assert not np.any(np.all(planes[:k] == row, axis=1))
    # and at least one scene sits on a duplicated plane whose later copy would tie
    assert any(np.sum(np.all(planes == row, axis=1)) > 1 for row in chosen)
    for name in ('db10', 'db1k'):
        g = helpers.load_polling_golden(name)
        assert np.any(g['orientations'] < 0)                         # -1 padding rows present
        assert set(np.unique(g['orientations'])) >= {0, 1, 2, 3}     # all four orientation classes


def test_selection_rule_against_literal_masked_argmin():
    """ brute-force: literal (votes, residual, zc) masking vs the oracle's index on random scenes """
    planes = synthetic.load_plane_database('100')
    d = synthetic.synthetic_polling_batch(planes, batch=2, num_dets=16, seed=99)
    kp, kpl, res, idx = polling_np.fit_road_planes(d['boxes'], d['dimensions'], d['orientations'], d['P_inv'],
                                                   d['planes'], return_index=True)
    planes_c = polling_np.canonical_planes(d['planes'])
    rays = polling_np.back_project(d['boxes'], d['P_inv'])
    X, zc = polling_np.hypotheses(rays, planes_c)
    # keypoints l, m, r lie on the plane and on their rays; t lies on the normal through m
    for b in range(2):
        for i in range(16):
            j = idx[b, i]
            n, dd = planes_c[b, j, :3], planes_c[b, j, 3]
            for k in range(3):
                assert abs(np.dot(n, kp[b, i, k]) + dd) < 1e-3
                assert np.linalg.norm(np.cross(kp[b, i, k], rays[b, i, k])) / np.linalg.norm(kp[b, i, k]) < 1e-4
            v = kp[b, i, 3] - kp[b, i, 1]
            assert np.linalg.norm(np.cross(v, n)) < 1e-3 * max(1.0, np.linalg.norm(v))
    assert np.all(planes_c[..., 1] <= 0)
    assert np.allclose(np.linalg.norm(planes_c[..., :3], axis=-1), 1.0, atol=1e-6)


def test_orientation_minus_one_targets_are_zero():
    t = polling_np.poll_targets(np.array([[[1.5, 1.6, 4.0]]], np.float32), np.array([[-1]]))
    assert t[1][0, 0] == 0 and t[2][0, 0] == 0 and t[4][0, 0] == 0 and t[5][0, 0] == 0
    assert t[0][0, 0] == np.float32(1.5)
    t = polling_np.poll_targets(np.array([[[1.5, 1.6, 4.0]]], np.float32), np.array([[1]]))
    assert t[1][0, 0] == np.float32(1.6) and t[2][0, 0] == np.float32(4.0)
