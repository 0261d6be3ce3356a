"""
End-to-end parity ledger between two runs of the predict_on_batch path on the same inputs -- a run at the reference's
arithmetic precision (float32) and a run at a narrower storage type (bf16 / f16), or the float32 HIP path against the
CPU oracle.

What the reference returns from bin/run_network.py:110 is decided AFTER the conv stack by
layers/filter_detections.py:54-64,155-177 (threshold, NMS, top-k) and layers/fit_road_planes.py:112-137 (masked
arg-min over the plane database), so agreement is measured there, in the terms BASELINE.json's north_star uses:

    detection-set agreement     which anchors survive threshold + NMS + top-k (identified by anchor id)
    orientation / plane index   on the detections both runs report: same orientation class, same selected plane
    corner deviation            max distance between the 8 cuboid corners (run_network.py:137-310) of such a detection,
                                and between the four 3-D keypoints the polling layer returns, in metres -- reported
                                (a) over the detections whose reference keypoints all lie within RANGE_M = 100 m of the
                                camera (the working range of the KITTI scenes north_star speaks of), (b) relative to the
                                distance of the point, over all of them: with random weights many "detections" are
                                geometric nonsense whose rays graze the plane, their keypoints lie 10^3..10^6 m away
                                and an absolute deviation says nothing there

Nothing here is reference code; it is the measurement both bench.py and tests/ report.
"""

import numpy as np

from . import gpp_utils

RANGE_M = 100.0

# BASELINE.json north_star: "bit-exact plane-index/argmax selection, 3D box corners within 1e-3" against the reference-precision
# path on identical inputs.  A throughput mode may be quoted as the BASELINE metric only when its ledger against the float32 path
# meets all three (bench.py enforces it for its headline type, tests/test_fullsize_gpu.py for 'f16x3').
#
# Beyond 100 m (round 4): a keypoint is the intersection of a pixel ray with a road plane, z = d f / (v - c_y), so a pixel
# perturbation dv moves it by dz = z^2 / (d f) dv -- the condition number grows with the SQUARE of the distance (camera height
# d ~ 1.65 m, focal length f ~ 774 px at the network's scale: 7.8 m per pixel at 100 m, i.e. 1e-3 m there is 1.3e-4 px, four
# float32 ulps of a pixel coordinate).  The bar for a detection whose farthest coordinate is r > 100 m is therefore
# 1e-3 m x (r / 100 m)^2: continuous at 100 m, the same bound on the pixel-space perturbation at every distance
# ('max_corner_dev_scaled_beyond_100m' = max over those detections of deviation / (r / 100)^2).
REFERENCE_BARS = {'detection_set_agreement': 1.0, 'plane_index_agreement': 1.0, 'max_corner_dev_m_within_100m': 1e-3,
                  'max_corner_dev_scaled_beyond_100m': 1e-3}


def meets_reference_bars(led):
    """ the same detections, the same orientation and plane for every one of them (integer counts, not rounded ratios), 3-D corners
    within 1e-3 m wherever the geometry lies within 100 m -- and at least one detection must lie there, an empty set meets nothing --,
    within 1e-3 m x (r / 100 m)^2 beyond """
    return bool(led['common'] == led['union'] and led['same_orientation'] == led['common'] and led['same_plane'] == led['common'] and
                led['common'] > 0 and led['same_plane_within_100m'] > 0 and
                led['max_corner_dev_m_within_100m'] <= REFERENCE_BARS['max_corner_dev_m_within_100m'] and
                led['max_corner_dev_scaled_beyond_100m'] <= REFERENCE_BARS['max_corner_dev_scaled_beyond_100m'])


def _dev(a, b):
    """ |a - b| element-wise in float64; a non-finite value (a degenerate pose of a random-weight detection) counts as
    equal when both runs produce it at the same place, as an infinite deviation otherwise """
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fa, fb = np.isfinite(a), np.isfinite(b)
    with np.errstate(invalid='ignore'):
        d = np.abs(a - b)
    return np.where(fa & fb, d, np.where(~fa & ~fb, 0.0, np.inf))


def _per_image(outs, anchor_index, plane_index, b):
    det = gpp_utils.select_detections(outs, 1.0, image_index=b)
    scores = np.asarray(outs[2][b])
    idx = np.where(scores > 0.05)[0]
    order = idx[np.argsort(-scores[idx], kind='stable')[:100]]
    det['anchor'] = np.asarray(anchor_index[b])[order]
    det['plane'] = np.asarray(plane_index[b])[order]
    return det


def parity_ledger(ref_outs, ref_anchor, ref_plane, outs, anchor, plane, detail=None):
    """ ref_* : the 8 output arrays + (B, 100) anchor ids + (B, 100) selected plane indices of the reference-precision
    run; outs / anchor / plane: the same of the run under test.  Returns a dict of plain numbers.
    detail: a list that receives one (image, anchor id, reach_m, corner_dev_m, keypoint_dev_m, same_plane) tuple per common detection
    (tools/corner_deviation.py draws the distribution from it). """
    B = int(np.asarray(ref_outs[0]).shape[0])
    n_ref = n_got = n_common = 0
    same_orient = same_plane = 0
    max_kp = max_corner = max_box = max_score = 0.0
    max_kp_rel = max_corner_rel = 0.0
    max_corner_far = 0.0
    in_range = beyond = 0
    identical_images = 0
    for b in range(B):
        A = _per_image(ref_outs, ref_anchor, ref_plane, b)
        G = _per_image(outs, anchor, plane, b)
        pos_a = {int(a): i for i, a in enumerate(A['anchor'])}
        pos_g = {int(a): i for i, a in enumerate(G['anchor'])}
        common = sorted(set(pos_a) & set(pos_g))
        n_ref += len(pos_a)
        n_got += len(pos_g)
        n_common += len(common)
        identical_images += int(list(A['anchor']) == list(G['anchor']))
        if not common:
            continue
        ia = np.array([pos_a[a] for a in common])
        ig = np.array([pos_g[a] for a in common])
        so = A['orientations'][ia] == G['orientations'][ig]
        sp = so & (A['plane'][ia] == G['plane'][ig])
        same_orient += int(so.sum())
        same_plane += int(sp.sum())
        max_box = max(max_box, float(np.abs(A['boxes'][ia] - G['boxes'][ig]).max()))
        max_score = max(max_score, float(np.abs(A['scores'][ia] - G['scores'][ig]).max()))
        kp = _dev(A['keypoints'][ia], G['keypoints'][ig]).reshape(len(common), -1).max(axis=1)
        # the 8 cuboid corners are only comparable when both runs chose the same orientation (the pose branches differ)
        with np.errstate(all='ignore'):
            ca = gpp_utils.cuboid_corners(gpp_utils.recover_pose({k: v[ia] for k, v in A.items()}))
            cg = gpp_utils.cuboid_corners(gpp_utils.recover_pose({k: v[ig] for k, v in G.items()}))
        cd = _dev(ca, cg).reshape(len(common), -1).max(axis=1)
        with np.errstate(all='ignore'):
            reach = np.abs(A['keypoints'][ia].astype(np.float64)).reshape(len(common), -1).max(axis=1)     # farthest coordinate, m
            reach = np.where(np.isfinite(reach), reach, np.inf)
        near = sp & (reach <= RANGE_M)                                # same plane selected, geometry in the working range
        in_range += int(near.sum())
        if near.any():
            max_kp = max(max_kp, float(kp[near].max()))
            max_corner = max(max_corner, float(cd[near].max()))
        far = sp & ~(reach <= RANGE_M)
        beyond += int(far.sum())
        if far.any():
            with np.errstate(all='ignore'):
                scaled = np.where(np.isfinite(reach[far]), cd[far] / (reach[far] / RANGE_M) ** 2, np.where(cd[far] == 0.0, 0.0, np.inf))
            max_corner_far = max(max_corner_far, float(scaled.max()))
        if detail is not None:
            detail.extend((b, int(a), float(r_), float(c_), float(k_), bool(s_)) for a, r_, c_, k_, s_ in zip(common, reach, cd, kp, sp))
        if sp.any():
            scale = np.maximum(np.where(np.isfinite(reach), reach, 1.0), 1.0)
            max_kp_rel = max(max_kp_rel, float((kp[sp] / scale[sp]).max()))
            max_corner_rel = max(max_corner_rel, float((cd[sp] / scale[sp]).max()))
    union = n_ref + n_got - n_common
    return {
        'images': B, 'detections_ref': n_ref, 'detections': n_got, 'common': n_common, 'union': union,
        'same_orientation': same_orient, 'same_plane': same_plane,
        'detection_set_agreement': round(n_common / union, 6) if union else 1.0,          # Jaccard index over anchor ids
        'detection_recall_of_ref': round(n_common / n_ref, 6) if n_ref else 1.0,
        'images_with_identical_detection_lists': identical_images,
        'orientation_agreement': round(same_orient / n_common, 6) if n_common else 1.0,
        'plane_index_agreement': round(same_plane / n_common, 6) if n_common else 1.0,
        'max_score_diff': max_score, 'max_box_diff_px': max_box,
        'same_plane_within_100m': in_range, 'same_plane_beyond_100m': beyond, 'max_corner_dev_scaled_beyond_100m': max_corner_far,
        'max_keypoint_dev_m_within_100m': max_kp, 'max_corner_dev_m_within_100m': max_corner,
        'max_keypoint_rel_dev': max_kp_rel, 'max_corner_rel_dev': max_corner_rel,
    }
