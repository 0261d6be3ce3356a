"""
GPU parity of the HIP polling kernel (through the C ABI) against the golden vectors from the
reference and against the CPU oracle.  Bar: plane index exact, everything else bit for bit
against the oracle (the kernel is built with -ffp-contract=off), <= 1e-3 against the goldens.
"""
import numpy as np
import pytest

import helpers
from keras_retinanet_3D.utils import gpp_utils, synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', helpers.polling_golden_names())
def test_hip_polling_matches_reference_goldens(name):
    g = helpers.load_polling_golden(name)
    B = g['boxes'].shape[0]
    kp, kpl, res, idx = gpp_utils.fit_road_planes(g['boxes'], g['dimensions'], g['orientations'], g['P_inv'],
                                                  np.tile(g['planes'][None], (B, 1, 1)), return_index=True)
    assert np.array_equal(idx.astype(np.int64), g['best_index'])       # bit-exact plane selection
    assert np.array_equal(kpl, g['keyplanes'])
    valid = g['orientations'] >= 0
    assert np.abs(kp - g['keypoints'])[valid].max() <= 1e-3            # 3D keypoints within 1e-3 m
    assert np.abs(res - g['residuals'])[valid].max() <= 1e-4


@pytest.mark.parametrize('name', helpers.polling_golden_names())
@pytest.mark.parametrize('batched', [0, 1])
def test_hip_polling_is_bit_identical_to_oracle(name, batched, oracle_lib):
    g = helpers.load_polling_golden(name)
    B = g['boxes'].shape[0]
    planes = np.tile(g['planes'][None], (B, 1, 1)) if batched else g['planes']
    ref = helpers.c_oracle_poll(oracle_lib, g['boxes'], g['dimensions'], g['orientations'], g['P_inv'], planes)
    got = gpp_utils.fit_road_planes(g['boxes'], g['dimensions'], g['orientations'], g['P_inv'], planes, return_index=True)
    assert np.array_equal(got[3], ref[3])
    assert helpers.bits_equal(got[0], ref[0])
    assert helpers.bits_equal(got[1], ref[1])
    assert helpers.bits_equal(got[2], ref[2])


@pytest.mark.parametrize('db,batch,dets', [('10', 1, 100), ('1k', 8, 100), ('10k', 8, 100), ('22k', 4, 100), ('100', 3, 7), ('1k', 1, 1)])
def test_hip_polling_full_size_configs_against_oracle(db, batch, dets, oracle_lib):
    """ BASELINE.json configurations (1k planes x batch 8, 10k x 8, 22k x 4/GPU) and ragged shapes """
    planes = synthetic.load_plane_database(db).astype(np.float32)
    d = synthetic.synthetic_polling_batch(planes, batch=batch, num_dets=dets, num_valid=max(1, dets * 3 // 4), seed=7)
    ref = helpers.c_oracle_poll(oracle_lib, d['boxes'], d['dimensions'], d['orientations'], d['P_inv'], planes)
    got = gpp_utils.fit_road_planes(d['boxes'], d['dimensions'], d['orientations'], d['P_inv'], planes, return_index=True)
    assert np.array_equal(got[3], ref[3])
    for a, b in zip(got[:3], ref[:3]):
        assert helpers.bits_equal(a, b)


def test_hip_polling_per_image_databases_differ(oracle_lib):
    """ (B, N, 4) planes with a different database per image """
    rng = np.random.default_rng(3)
    base = synthetic.load_plane_database('1k').astype(np.float32)
    planes = np.stack([base[rng.permutation(1000)[:640]] for _ in range(3)])
    d = synthetic.synthetic_polling_batch(base, batch=3, num_dets=40, seed=21)
    ref = helpers.c_oracle_poll(oracle_lib, d['boxes'], d['dimensions'], d['orientations'], d['P_inv'], planes)
    got = gpp_utils.fit_road_planes(d['boxes'], d['dimensions'], d['orientations'], d['P_inv'], planes, return_index=True)
    assert np.array_equal(got[3], ref[3])
    for a, b in zip(got[:3], ref[:3]):
        assert helpers.bits_equal(a, b)


def test_hip_polling_adversarial_ties_and_sentinels(oracle_lib):
    """ duplicated planes (first index must win), planes reversed, tiny databases (N < 64 lanes) """
    base = synthetic.load_plane_database('100').astype(np.float32)
    for planes in (np.concatenate([base, base, base]), base[::-1].copy(), base[:1], base[:3], base[:65],
                   np.concatenate([base[:50]] * 7)):
        d = synthetic.synthetic_polling_batch(base, batch=2, num_dets=33, seed=5)
        # make some rows degenerate: swapped l/r (all masked) and absurd height (> 100 residuals)
        d['boxes'][0, :5, 4:6], d['boxes'][0, :5, 8:10] = d['boxes'][0, :5, 8:10].copy(), d['boxes'][0, :5, 4:6].copy()
        d['dimensions'][1, :5, 0] = 300.0
        ref = helpers.c_oracle_poll(oracle_lib, d['boxes'], d['dimensions'], d['orientations'], d['P_inv'], planes)
        got = gpp_utils.fit_road_planes(d['boxes'], d['dimensions'], d['orientations'], d['P_inv'], planes, return_index=True)
        assert np.array_equal(got[3], ref[3])
        for a, b in zip(got[:3], ref[:3]):
            assert helpers.bits_equal(a, b)


def test_hip_polling_is_deterministic_and_torch_in_torch_out():
    import torch
    planes = synthetic.load_plane_database('1k').astype(np.float32)
    d = synthetic.synthetic_polling_batch(planes, batch=4, num_dets=100, seed=1)
    args = [torch.as_tensor(d[k]).cuda() for k in ('boxes', 'dimensions', 'orientations', 'P_inv')] + [torch.as_tensor(planes).cuda()]
    a = gpp_utils.fit_road_planes(*args, return_index=True)
    b = gpp_utils.fit_road_planes(*args, return_index=True)
    assert all(isinstance(t, torch.Tensor) and t.is_cuda for t in a)
    for x, y in zip(a, b):
        assert torch.equal(x, y) or bool(torch.all((x == y) | (x != x)))


def test_hip_polling_empty_and_bad_arguments():
    planes = synthetic.load_plane_database('10').astype(np.float32)
    out = gpp_utils.fit_road_planes(np.zeros((0, 100, 12), np.float32), np.zeros((0, 100, 3), np.float32),
                                    np.zeros((0, 100), np.int32), np.zeros((0, 4, 3), np.float32), planes)
    assert out[0].shape == (0, 100, 4, 3) and out[1].shape == (0, 100, 1, 4) and out[2].shape == (0, 100)
    with pytest.raises(ValueError):
        gpp_utils.fit_road_planes(np.zeros((1, 4, 11), np.float32), np.zeros((1, 4, 3), np.float32),
                                  np.zeros((1, 4), np.int32), np.zeros((1, 4, 3), np.float32), planes)
    with pytest.raises(ValueError):
        gpp_utils.fit_road_planes(np.zeros((1, 4, 12), np.float32), np.zeros((1, 4, 3), np.float32),
                                  np.zeros((1, 4), np.int32), np.zeros((1, 4, 3), np.float32), planes[:0])


def test_padding_rows_are_polled_once_per_run_and_keep_the_reference_bytes(oracle_lib):
    """ rows with the exact -1 padding of FilterDetections give the same result for every padding row of an image: the kernel polls
    the first row of a run and copies.  Runs at the end (the usual case), in the middle, at the start, whole images of padding, a
    single valid row, and rows that LOOK like padding (orientation -1) but are not -- all against the oracle, which polls every row """
    planes = synthetic.load_plane_database('1k').astype(np.float32)
    d = synthetic.synthetic_polling_batch(planes, batch=5, num_dets=40, num_valid=40, seed=11)

    def pad(b, rows):
        d['boxes'][b, rows] = -1.0
        d['dimensions'][b, rows] = -1.0
        d['orientations'][b, rows] = -1

    pad(0, slice(7, 40))                                  # 7 detections + 33 padding rows
    pad(1, slice(0, 5)); pad(1, slice(20, 23)); pad(1, slice(39, 40))      # runs at the start, in the middle, a single row at the end
    pad(2, slice(0, 40))                                  # nothing detected
    pad(3, slice(1, 40))                                  # one detection
    d['orientations'][4, 10:14] = -1                      # orientation -1 with real boxes: not padding, polled as the reference would
    d['boxes'][4, 20] = -1.0; d['dimensions'][4, 20] = -1.0          # padding boxes with a valid orientation: not padding either
    d['boxes'][4, 30] = -1.0; d['dimensions'][4, 30] = -1.0; d['orientations'][4, 30] = -1; d['boxes'][4, 30, 11] = -1.5   # almost
    pad(4, slice(31, 36))
    for planes_in in (planes, np.tile(planes[None], (5, 1, 1))):
        ref = helpers.c_oracle_poll(oracle_lib, d['boxes'], d['dimensions'], d['orientations'], d['P_inv'], planes_in)
        got = gpp_utils.fit_road_planes(d['boxes'], d['dimensions'], d['orientations'], d['P_inv'], planes_in, return_index=True)
        assert np.array_equal(got[3], ref[3])
        for a, b in zip(got[:3], ref[:3]):
            assert helpers.bits_equal(a, b)
    # and it is the point of the exercise: a frame with 7 detections of 100 costs a fraction of a full one
    import time
    import torch
    full = synthetic.synthetic_polling_batch(synthetic.load_plane_database('10k').astype(np.float32), batch=8, num_dets=100, seed=3)
    few = {k: v.copy() for k, v in full.items()}
    few['boxes'][:, 7:] = -1.0; few['dimensions'][:, 7:] = -1.0; few['orientations'][:, 7:] = -1
    p10k = torch.as_tensor(synthetic.load_plane_database('10k').astype(np.float32)).cuda()
    times = []
    for batch in (full, few):
        args = [torch.as_tensor(batch[k]).cuda() for k in ('boxes', 'dimensions', 'orientations', 'P_inv')] + [p10k]
        for _ in range(3):
            gpp_utils.fit_road_planes(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            gpp_utils.fit_road_planes(*args)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 20)
    print('polling 8 x 100 rows x 10k planes: all valid {:.1f} us, 7 valid {:.1f} us'.format(times[0] * 1e6, times[1] * 1e6))
    # (not 8 / 100 of it: a scan of 10k planes by one workgroup takes ~40 us whatever the other CUs do -- with few detections the
    # stage is bound by the latency of one scan, not by the number of scans)
    assert times[1] < 0.7 * times[0]
