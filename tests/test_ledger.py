"""
CPU tests of utils/ledger.py (the end-to-end parity ledger bench.py and the GPU tests report): identical runs, a dropped
detection, a changed plane index, a keypoint deviation, and the treatment of far-away / non-finite geometry.
"""
import numpy as np

from keras_retinanet_3D.utils import ledger


def fake_run(seed=0, batch=2, valid=80):
    rng = np.random.default_rng(seed)
    boxes = rng.uniform(0, 1000, (batch, 100, 12)).astype(np.float32)
    dims = rng.uniform(1, 4, (batch, 100, 3)).astype(np.float32)
    scores = np.sort(rng.uniform(0.1, 1, (batch, 100)).astype(np.float32), axis=1)[:, ::-1].copy()
    scores[:, valid:] = -1
    labels = np.zeros((batch, 100), np.int32)
    orient = rng.integers(0, 4, (batch, 100)).astype(np.int32)
    kp = (rng.normal(size=(batch, 100, 4, 3)) * 20).astype(np.float32)
    kpl = rng.normal(size=(batch, 100, 1, 4)).astype(np.float32)
    res = rng.uniform(size=(batch, 100)).astype(np.float32)
    anchors = np.stack([rng.permutation(5000)[:100] for _ in range(batch)]).astype(np.int32)
    anchors[:, valid:] = -1
    planes = rng.integers(0, 1000, (batch, 100)).astype(np.int32)
    return [boxes, dims, scores, labels, orient, kp, kpl, res], anchors, planes


def test_identical_runs():
    o, a, p = fake_run()
    led = ledger.parity_ledger(o, a, p, o, a, p)
    assert led['detections_ref'] == led['detections'] == led['common'] == 160
    assert led['detection_set_agreement'] == 1.0 and led['images_with_identical_detection_lists'] == 2
    assert led['orientation_agreement'] == 1.0 and led['plane_index_agreement'] == 1.0
    assert led['max_keypoint_dev_m_within_100m'] == 0.0 and led['max_corner_rel_dev'] == 0.0 and led['max_box_diff_px'] == 0.0


def test_each_kind_of_difference_is_counted_where_it_belongs():
    o, a, p = fake_run()
    o2 = [x.copy() for x in o]
    a2, p2 = a.copy(), p.copy()
    a2[1, 5] = 99999                       # another anchor survived NMS in image 1
    p2[0, 7] += 1                          # another plane selected for detection 7 of image 0
    o2[5][0, 3] += np.float32(0.01)        # keypoints of detection 3 of image 0 moved by 1 cm
    o2[4][1, 9] = (o2[4][1, 9] + 1) % 4    # orientation class of detection 9 of image 1 changed
    led = ledger.parity_ledger(o, a, p, o2, a2, p2)
    assert led['common'] == 159 and abs(led['detection_set_agreement'] - 159 / 161) < 1e-6
    assert led['images_with_identical_detection_lists'] == 1
    assert abs(led['orientation_agreement'] - 158 / 159) < 1e-6
    assert abs(led['plane_index_agreement'] - 157 / 159) < 1e-6          # the orientation change also leaves the "same plane" set
    assert 0.0099 < led['max_keypoint_dev_m_within_100m'] < 0.0101


def test_far_away_and_non_finite_geometry():
    o, a, p = fake_run()
    o[5][0, 0] *= 1e5                      # a grazing-ray detection: keypoints 10^6 m away
    o2 = [x.copy() for x in o]
    o2[5][0, 0] *= np.float32(1.0 + 1e-5)  # 10 m absolute, 1e-5 relative
    o[5][1, 1, 0, 0] = np.nan              # the same non-finite value in both runs counts as equal
    o2[5][1, 1, 0, 0] = np.nan
    led = ledger.parity_ledger(o, a, p, o2, a, p)
    assert led['max_keypoint_dev_m_within_100m'] == 0.0                  # the far detection is outside the 100 m range
    assert 0.5e-5 < led['max_keypoint_rel_dev'] < 2e-5
    o2[5][1, 2, 1, 1] = np.inf             # non-finite in one run only: an infinite deviation
    assert ledger.parity_ledger(o, a, p, o2, a, p)['max_keypoint_rel_dev'] == np.inf


def test_reference_bars_compare_counts_and_need_detections_in_range():
    o, a, p = fake_run()
    led = ledger.parity_ledger(o, a, p, o, a, p)
    assert led['union'] == led['common'] == led['same_plane'] == led['same_orientation'] == 160
    assert led['same_plane_within_100m'] + led['same_plane_beyond_100m'] == 160 and ledger.meets_reference_bars(led)
    # one detection of 160 000 would vanish in a ratio rounded to six digits; the counts see it
    near_miss = dict(led, common=159999, union=160000, same_plane=159999, same_orientation=159999, detection_set_agreement=1.0,
                     set_differences=1, set_differences_unexplained=1)
    assert not ledger.meets_reference_bars(near_miss)
    # no detection inside the working range: the corner bar has not been met, it has not been measured
    o_far = [x.copy() for x in o]
    o_far[5] *= 1e4
    led_far = ledger.parity_ledger(o_far, a, p, o_far, a, p)
    assert led_far['same_plane_within_100m'] == 0 and led_far['max_corner_dev_m_within_100m'] == 0.0
    assert not ledger.meets_reference_bars(led_far)
    # nothing detected at all meets nothing either
    empty = [x.copy() for x in o]
    empty[2][:] = -1
    assert not ledger.meets_reference_bars(ledger.parity_ledger(empty, a, p, empty, a, p))


def test_the_bar_beyond_100m_scales_with_the_square_of_the_distance():
    o, a, p = fake_run()
    o[5][0, 0] = np.float32(1000.0) * np.sign(o[5][0, 0] + np.float32(1e-9))        # every coordinate of detection 0 at 1 km: (r / 100)^2 = 100
    o2 = [x.copy() for x in o]
    o2[5][0, 0] += np.float32(0.05)                                                 # 5 cm at 1 km = 5e-4 m scaled to 100 m
    detail = []
    led = ledger.parity_ledger(o, a, p, o2, a, p, detail=detail)
    assert led['same_plane_beyond_100m'] >= 1 and led['max_corner_dev_m_within_100m'] == 0.0
    assert len(detail) == led['common'] and max(d[2] for d in detail) >= 1000.0
    far = [d for d in detail if d[2] >= 1000.0]
    assert all(d[5] for d in far) and abs(max(d[4] for d in far) - 0.05) < 1e-3      # (image, anchor, reach, corner dev, keypoint dev, same plane)
    o3 = [x.copy() for x in o]
    o3[5][0, 0] += np.float32(5.0)                                                  # 5 m at 1 km = 5e-2 scaled: far outside
    assert ledger.parity_ledger(o, a, p, o3, a, p)['max_corner_dev_scaled_beyond_100m'] > 1e-3
    assert not ledger.meets_reference_bars(ledger.parity_ledger(o, a, p, o3, a, p))


def test_a_set_difference_at_a_tie_of_the_top_k_cut_is_told_from_a_real_one():
    o, a, p = fake_run(valid=100)                                          # full lists: the cut is the 100th score
    o2 = [x.copy() for x in o]
    a2 = a.copy()
    a2[0, 99] = 77777                                                      # another anchor in the last place, its score the same to 1e-7
    o2[2][0, 99] = o[2][0, 99] + np.float32(1e-7)
    led = ledger.parity_ledger(o, a, p, o2, a2, p)
    assert led['set_differences'] == 2 and led['set_differences_at_a_tie'] == 2 and led['set_differences_unexplained'] == 0
    assert ledger.meets_reference_bars(led)                                # decided by rounding noise: in the reference as much as here
    a3 = a.copy()
    a3[0, 50] = 88888                                                      # a detection from the middle of the list replaced: not a tie
    led = ledger.parity_ledger(o, a, p, o, a3, p)
    assert led['set_differences'] == 2 and led['set_differences_at_a_tie'] == 0 and not ledger.meets_reference_bars(led)
    o4, a4, p4 = fake_run(valid=80)                                        # lists that are not full have no cut: nothing is a tie
    a5 = a4.copy()
    a5[0, 79] = 99999
    assert ledger.parity_ledger(o4, a4, p4, o4, a5, p4)['set_differences_at_a_tie'] == 0


def test_pair_bars_are_twice_the_metre_bars():
    o, a, p = fake_run()
    o2 = [x.copy() for x in o]
    near = np.abs(o[5]).reshape(2, 100, -1).max(axis=2) <= 60.0
    b, dts = np.nonzero(near[:, :80])
    o2[5][b[0], dts[0]] += np.float32(1.5e-3)                              # 1.5 mm on a detection inside the working range
    led = ledger.parity_ledger(o, a, p, o2, a, p)
    assert 1.2e-3 < led['max_keypoint_dev_m_within_100m'] < 1.8e-3
    if led['max_corner_dev_m_within_100m'] <= 2e-3:
        assert not ledger.meets_reference_bars(led) and ledger.meets_reference_bars(led, pair=True)


def test_a_plane_difference_with_equal_polling_inputs_is_told_from_one_with_different_inputs():
    """ the polling selection is discontinuous in its inputs (a vote at its 0.7 m threshold): with boxes / dimensions that agree to float32
    noise one flip per thousand detections is within the bars; a flip whose boxes differ visibly, or more flips than that, is not """
    o, a, p = fake_run(batch=10, valid=100)
    o2 = [x.copy() for x in o]
    p2 = p.copy()
    p2[0, 7] += 1
    o2[0][0, 7] += np.float32(6e-5)                                      # the 2-D box moved by float32 noise
    led = ledger.parity_ledger(o, a, p, o2, a, p2)
    assert led['plane_differences'] == led['plane_differences_with_equal_inputs'] == 1 and led['same_plane'] == 999
    assert ledger.meets_reference_bars(led)
    p2[3, 1] += 1                                                        # a second one in 1000: over the allowance
    led = ledger.parity_ledger(o, a, p, o2, a, p2)
    assert led['plane_differences_with_equal_inputs'] == 2 and not ledger.meets_reference_bars(led)
    p2[3, 1] -= 1
    o2[0][0, 7] += np.float32(0.01)                                      # the box moved by 0.01 px: a difference of the inputs, not a vote at its threshold
    led = ledger.parity_ledger(o, a, p, o2, a, p2)
    assert led['plane_differences'] == 1 and led['plane_differences_with_equal_inputs'] == 0 and not ledger.meets_reference_bars(led)
