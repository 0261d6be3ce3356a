"""
CPU tests of the evaluation harness (utils/eval.py, preprocessing/kitti.py)
against goldens produced by the reference's own utils/eval.py `evaluate()` and
preprocessing/kitti.py `KittiGenerator` (oracle/gen_eval_goldens.py).  AP and error values must
agree to 1e-12 (same float64 arithmetic, same summation order), detection rows exactly.
"""
import glob
import os

import numpy as np
import pytest
import scipy.io

import helpers
from keras_retinanet_3D.preprocessing import kitti
from keras_retinanet_3D.utils import anchors, eval as gpp_eval

CASES = [p for p in sorted(glob.glob(os.path.join(helpers.GOLDEN, 'eval_*.npz'))) if 'kitti_parsing' not in p]


class GoldenGenerator(object):
    def __init__(self, g):
        self.g = g
        self.plane_params = g['planes']
        cuts = np.cumsum(g['ann_counts'])[:-1]
        self.annotations = np.split(g['annotations'], cuts, axis=0)
        self._current = None

    def size(self):
        return len(self.annotations)

    def num_classes(self):
        return int(self.g['num_classes'])

    def label_to_name(self, label):
        return 'Car'

    def load_image(self, i):
        self._current = i
        return np.zeros((4, 6, 3), np.uint8)

    def preprocess_image(self, image):
        return image.astype(np.float32)

    def resize_image(self, image):
        return image, float(self.g['scales'][self._current])

    def load_calibration(self, i):
        return np.array([[700.0 + i, 0, 600, 40], [0, 700.0 + i, 170, 0.2], [0, 0, 1, 0.003]])

    def load_annotations(self, i):
        return self.annotations[i], np.zeros((0, 4))


class GoldenModel(object):
    """ replays the stored model outputs; serves any batch size and checks the inputs it is fed """

    def __init__(self, g):
        self.g, self.cursor, self.batches = g, 0, []

    def predict_on_batch(self, inputs):
        images, P_inv, planes = inputs
        B = images.shape[0]
        assert P_inv.shape == (B, 4, 3) and planes.shape == (B,) + self.g['planes'].shape
        for k in range(B):
            i = self.cursor + k
            s = float(self.g['scales'][i])
            P = np.diag([s, s, 1.0]).dot(np.array([[700.0 + i, 0, 600, 40], [0, 700.0 + i, 170, 0.2], [0, 0, 1, 0.003]]))
            assert np.allclose(P_inv[k], np.linalg.pinv(P), rtol=0, atol=1e-15)
        out = [self.g['outputs_{}'.format(j)][self.cursor:self.cursor + B].copy() for j in range(8)]
        self.cursor += B
        self.batches.append(B)
        return out


SETTINGS = {'default': {}, 'strict': {'iou_threshold': 0.7, 'score_threshold': 0.3, 'max_detections': 5}}


@pytest.mark.parametrize('batch_size', [1, 4])
@pytest.mark.parametrize('tag', sorted(SETTINGS))
@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p) for p in CASES])
def test_evaluate_matches_the_reference(path, tag, batch_size):
    g = dict(np.load(path))
    gen, model = GoldenGenerator(g), GoldenModel(g)
    aps, ke, he, we, le = gpp_eval.evaluate(gen, model, batch_size=batch_size, **SETTINGS[tag])
    assert max(model.batches) == min(batch_size, gen.size())
    want = g[tag + '_ap']
    assert sorted(aps) == list(range(4 * gen.num_classes())) == list(range(len(want)))
    for label in aps:
        assert abs(float(aps[label][0]) - want[label, 0]) < 1e-12 and float(aps[label][1]) == want[label, 1]
    assert np.allclose([ke, he, we, le], g[tag + '_errors'], rtol=0, atol=1e-12)


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p) for p in CASES])
def test_detection_rows_match_the_reference(path):
    g = dict(np.load(path))
    for tag, args in SETTINGS.items():
        dets = gpp_eval._get_detections(GoldenGenerator(g), GoldenModel(g), score_threshold=args.get('score_threshold', 0.05),
                                        max_detections=args.get('max_detections', 100), batch_size=3)
        counts = np.array([[d.shape[0] for d in per_image] for per_image in dets])
        assert np.array_equal(counts, g[tag + '_det_counts'])
        rows = np.concatenate([d for per_image in dets for d in per_image], axis=0)
        assert rows.shape[1] == 32 and np.array_equal(rows, g[tag + '_det_concat'])


def test_compute_ap_known_answers():
    assert gpp_eval._compute_ap(np.array([]), np.array([])) == 0.0
    assert gpp_eval._compute_ap(np.array([0.5, 1.0]), np.array([1.0, 1.0])) == 1.0
    # recall 0.5 @ precision 1, then 1.0 @ 2/3 after one miss: 0.5 * 1 + 0.5 * 2/3
    ap = gpp_eval._compute_ap(np.array([0.5, 0.5, 1.0]), np.array([1.0, 0.5, 2.0 / 3.0]))
    assert abs(ap - (0.5 + 0.5 * 2.0 / 3.0)) < 1e-15


def test_compute_overlap_properties():
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 100, (7, 4))
    a[:, 2:] += a[:, :2]
    iou = anchors.compute_overlap(a, a)
    assert np.allclose(np.diag(iou), 1.0) and np.allclose(iou, iou.T) and iou.min() >= 0 and iou.max() <= 1 + 1e-12
    assert anchors.compute_overlap(np.array([[0., 0, 1, 1]]), np.array([[2., 2, 3, 3]]))[0, 0] == 0.0
    assert anchors.compute_overlap(np.array([[0., 0, 2, 2]]), np.array([[1., 0, 3, 2]]))[0, 0] == pytest.approx(1.0 / 3.0)
    assert anchors.compute_overlap(np.zeros((1, 4)), np.zeros((1, 4)))[0, 0] == 0.0       # degenerate: eps-clamped union


def test_summarize_is_the_callback_mean():
    aps = {0: (0.5, 3.0), 1: (0, 0), 2: (0.25, 1.0), 3: (0.0, 2.0)}
    logs = gpp_eval.summarize((aps, 1.5, 0.1, 0.2, 0.3), verbose=0)
    assert logs['mAP'] == pytest.approx(0.25) and logs['keypoints (mean L1 error)'] == 1.5 and logs['length (mean L1 error)'] == 0.3


def test_kitti_generator_parses_like_the_reference(tmp_path):
    g = dict(np.load(os.path.join(helpers.GOLDEN, 'eval_kitti_parsing.npz')))
    from PIL import Image
    base = tmp_path / 'kitti'
    for d in ('images', 'labels', 'calibs'):
        os.makedirs(str(base / 'val' / d))
    scipy.io.savemat(str(base / 'road_planes_database.mat'), {'road_planes_database': g['plane_params']})
    stems = sorted(k[len('ann_'):] for k in g if k.startswith('ann_'))
    for n, stem in enumerate(stems):
        ext = '.jpg' if n == 1 else '.png'
        Image.fromarray(np.full((6, 8, 3), 10 * n, np.uint8)).save(str(base / 'val' / 'images' / (stem + ext)))
        (base / 'val' / 'labels' / (stem + '.txt')).write_text(str(g['label_text_' + stem]))
        (base / 'val' / 'calibs' / (stem + '.txt')).write_text(str(g['calib_text_' + stem]))
    gen = kitti.KittiGenerator(str(base), subset='val')
    assert gen.size() == len(stems) and gen.num_classes() == int(g['num_classes']) and gen.label_to_name(0) == str(g['name0'])
    assert np.array_equal(gen.plane_params, g['plane_params'])
    for i in range(gen.size()):
        stem = os.path.basename(gen.images[i])[:6]
        boxes, ignore = gen.load_annotations(i)
        assert boxes.shape == g['ann_' + stem].shape and np.array_equal(boxes, g['ann_' + stem])
        assert ignore.shape == g['ignore_' + stem].shape and np.array_equal(ignore, g['ignore_' + stem])
        assert np.array_equal(gen.load_calibration(i), g['P_' + stem])
        assert gen.load_image(i).shape == (6, 8, 3)
        img, scale = gen.resize_image(gen.preprocess_image(gen.load_image(i)))
        assert scale == pytest.approx(800.0 / 6.0) and img.dtype == np.float32


def test_drawing_is_refused_loudly():
    with pytest.raises(NotImplementedError):
        gpp_eval._get_detections(None, None, save_path='/tmp/x')
