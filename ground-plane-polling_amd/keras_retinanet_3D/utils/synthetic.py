"""
Seeded synthetic inputs for tests and benchmarks (no dataset, no checkpoint, no network).

Nothing here exists in the reference: the reference ships no sample image, calibration
or weights.  The conventions that the generators follow are the
reference's own:

* cuboid corner numbering and the keypoint <-> corner <-> orientation-class table:
  label_prep/computeBox3D.m:22-24 and label_prep/create_mod_labels.m:57-100
* batched model inputs [images, P_inv (B,4,3), planes (B,N,4)]:
  keras_retinanet_3D/preprocessing/kitti.py:204-223
* calibration handling (P = diag(s,s,1) P2, P_inv = pinv(P)):
  keras_retinanet_3D/bin/run_network.py:48-59
"""

import os

import numpy as np

# A KITTI-like camera (build-defined constant; the reference ships no calibration file).
KITTI_LIKE_P2 = np.array([
    [721.5377, 0.0, 609.5593, 44.85728],
    [0.0, 721.5377, 172.854, 0.2163791],
    [0.0, 0.0, 1.0, 0.002745884],
], dtype=np.float64)

KITTI_IMAGE_SHAPE = (375, 1242, 3)
# scale chosen by utils.image.resize_image for a 375x1242 frame (max side 1333)
KITTI_SCALE = 1333.0 / 1242.0
NETWORK_INPUT_SHAPE = (402, 1333, 3)

# keypoint (l, m, r, t) -> corner number (1-based) per orientation class,
# create_mod_labels.m:57-100
KEYPOINT_CORNERS = {
    0: (3, 2, 1, 6),
    1: (2, 1, 4, 5),
    2: (4, 3, 2, 7),
    3: (1, 4, 3, 8),
}


def plane_database_path(name):
    """ Path of one of the shipped road plane databases ('10', '100', '1k', '10k', '22k'). """
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.normpath(os.path.join(here, '..', '..', '..'))
    return os.path.join(root, 'road_planes_database', 'road_planes_database_{}.mat'.format(name))


def load_plane_database(name):
    """ (N, 4) float64 rows [a b c d] exactly as run_network.py:75 reads them. """
    import scipy.io
    return np.ascontiguousarray(scipy.io.loadmat(plane_database_path(name))['road_planes_database'])


def synthetic_calibration(scale=KITTI_SCALE, P2=KITTI_LIKE_P2):
    """ (P, P_inv) in float64 as load_calibration (run_network.py:48-59) would return them. """
    P = np.dot(np.array([[scale, 0.0, 0.0], [0.0, scale, 0.0], [0.0, 0.0, 1.0]]), P2)
    return P, np.linalg.pinv(P)


def canonical_plane(plane):
    """ Flip so that the normal points up (b <= 0) and normalise, fit_road_planes.py:75-77. """
    plane = np.asarray(plane, dtype=np.float64)
    plane = plane * -np.sign(plane[1])
    return plane / np.linalg.norm(plane[:3])


def cuboid_corners_on_plane(plane, x, z, yaw, h, w, l):
    """ The 8 corners (8,3) of a cuboid whose bottom face lies on `plane`.

    Object frame as computeBox3D.m:22-24: x = length axis, y = down, z = width axis;
    'down' is the negated (upward) plane normal, so the vertical edges are parallel to
    the normal, which is the model fit_road_planes.py:34-47 (calc_X_t) assumes.
    Returns (corners, centre_of_bottom_face).
    """
    p = canonical_plane(plane)
    n, d = p[:3], p[3]
    y = -(n[0] * x + n[2] * z + d) / n[1]
    t = np.array([x, y, z])
    ex = np.array([1.0, 0.0, 0.0])
    u = ex - np.dot(ex, n) * n
    u /= np.linalg.norm(u)
    v = np.cross(n, u)                    # (u, -n, v) is right handed
    axis_l = np.cos(yaw) * u - np.sin(yaw) * v
    axis_w = np.cross(n, axis_l)
    xs = np.array([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2])
    ys = np.array([0, 0, 0, 0, -h, -h, -h, -h], dtype=np.float64)
    zs = np.array([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2])
    corners = t[None, :] + xs[:, None] * axis_l[None, :] + ys[:, None] * (-n)[None, :] + zs[:, None] * axis_w[None, :]
    return corners, t


def orientation_class(yaw, t):
    """ Orientation class from the observation angle, create_mod_labels.m:57-100. """
    alpha = yaw - np.arctan2(t[0], t[2])
    alpha = (alpha + np.pi) % (2 * np.pi) - np.pi
    deg = np.degrees(alpha)
    if 0 <= deg < 90:
        return 0
    if 90 <= deg < 180:
        return 1
    if -90 <= deg < 0:
        return 2
    return 3


def synthetic_detections(planes, num_dets=100, num_valid=None, seed=2, P=None, pixel_noise=0.5,
                         dim_noise=0.02, plane_indices=None):
    """ One image worth of detections whose cuboids rest on rows of the plane database.

    Returns a dict with
        boxes (num_dets, 12) f32, dimensions (num_dets, 3) f32 (h, w, l),
        orientations (num_dets,) i32, true_plane (num_dets,) i64 (-1 on padding rows).
    Rows >= num_valid are the -1 padding FilterDetections emits (filter_detections.py:170-177).
    """
    rng = np.random.default_rng(seed)
    if P is None:
        P, _ = synthetic_calibration()
    planes = np.asarray(planes, dtype=np.float64)
    if num_valid is None:
        num_valid = num_dets
    boxes = -np.ones((num_dets, 12), dtype=np.float32)
    dims = -np.ones((num_dets, 3), dtype=np.float32)
    orient = -np.ones((num_dets,), dtype=np.int32)
    true_plane = -np.ones((num_dets,), dtype=np.int64)
    for i in range(num_valid):
        k = int(rng.integers(0, planes.shape[0])) if plane_indices is None else int(plane_indices[i])
        h = rng.uniform(1.4, 1.8)
        w = rng.uniform(1.5, 1.8)
        l = rng.uniform(3.5, 4.8)
        z = rng.uniform(8.0, 40.0)
        x = rng.uniform(-8.0, 8.0)
        yaw = rng.uniform(-np.pi, np.pi)
        corners, t = cuboid_corners_on_plane(planes[k], x, z, yaw, h, w, l)
        o = orientation_class(yaw, t)
        homo = np.concatenate([corners, np.ones((8, 1))], axis=1) @ P.T
        px = homo[:, :2] / homo[:, 2:3]
        kp = px[[c - 1 for c in KEYPOINT_CORNERS[o]]] + rng.normal(0.0, pixel_noise, size=(4, 2))
        boxes[i, 0] = px[:, 0].min()
        boxes[i, 1] = px[:, 1].min()
        boxes[i, 2] = px[:, 0].max()
        boxes[i, 3] = px[:, 1].max()
        boxes[i, 4:] = kp.reshape(-1)
        dims[i] = np.array([h, w, l]) * (1.0 + rng.normal(0.0, dim_noise, size=3))
        orient[i] = o
        true_plane[i] = k
    return {'boxes': boxes, 'dimensions': dims, 'orientations': orient, 'true_plane': true_plane}


def synthetic_polling_batch(planes, batch=1, num_dets=100, num_valid=None, seed=2):
    """ Batched polling inputs in the layer's input order (fit_road_planes.py:157-161). """
    P, P_inv = synthetic_calibration()
    dets = [synthetic_detections(planes, num_dets, num_valid, seed + 1000 * b, P) for b in range(batch)]
    return {
        'boxes': np.stack([d['boxes'] for d in dets]),
        'dimensions': np.stack([d['dimensions'] for d in dets]),
        'orientations': np.stack([d['orientations'] for d in dets]),
        'P_inv': np.tile(P_inv[None].astype(np.float32), (batch, 1, 1)),
        'planes': np.tile(np.asarray(planes, dtype=np.float32)[None], (batch, 1, 1)),
        'true_plane': np.stack([d['true_plane'] for d in dets]),
    }


def synthetic_image(seed=0, shape=KITTI_IMAGE_SHAPE):
    """ uint8 BGR frame, uniform noise. """
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=shape, dtype=np.uint8)


BGR_MEAN = np.array([103.939, 116.779, 123.68], np.float32)        # utils/image.py:26-62 ('caffe' mode)


def synthetic_network_input(seeds):
    """ (len(seeds), 402, 1333, 3) float32: the uint8 noise frames synthetic_image(seed) at 375x1242, brought to the network input
    size by nearest sampling on the host, BGR mean subtracted -- the tensor predict_on_batch receives.  bench.py's resident
    batches and the full-size oracle fixtures (oracle/gen_fullsize_goldens.py) are built from these. """
    seeds = list(seeds)
    out = np.empty((len(seeds), 402, 1333, 3), np.float32)
    ys = np.minimum((np.arange(402) * (375.0 / 402.0)).astype(np.int64), 374)
    xs = np.minimum((np.arange(1333) * (1242.0 / 1333.0)).astype(np.int64), 1241)
    for i, seed in enumerate(seeds):
        out[i] = synthetic_image(seed=int(seed))[ys][:, xs].astype(np.float32) - BGR_MEAN
    return out
