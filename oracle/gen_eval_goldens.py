"""
ORACLE SUPPORT (test infrastructure): generate tests/golden/eval_*.npz by executing the
reference's own evaluation code, UNMODIFIED:

  * /root/reference/keras_retinanet_3D/utils/eval.py `evaluate()` on a fake generator and a fake
    model (prepared predict_on_batch outputs): pins detection selection, the per-bin split, the
    greedy matching, tie handling of the unstable argsort, AP integration and the L1 errors;
  * /root/reference/keras_retinanet_3D/preprocessing/kitti.py `KittiGenerator` label / calibration
    parsing on a small KITTI-style directory written here (base-class constructor skipped: it
    builds a TF graph for training-time colour augmentation, unrelated to parsing).

keras / tensorflow -> oracle/np_tf_shim.py stand-ins (import-time only), cv2 -> an empty stub
(nothing here draws).  Run here only (needs /root/reference):   python oracle/gen_eval_goldens.py
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import scipy.io

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

import np_tf_shim  # noqa: E402

keras, tf = np_tf_shim.install()
cv2 = types.ModuleType('cv2')
for name in ('FONT_HERSHEY_PLAIN', 'LINE_AA', 'INTER_LINEAR', 'BORDER_CONSTANT', 'INTER_NEAREST', 'INTER_CUBIC',
             'INTER_AREA', 'INTER_LANCZOS4', 'BORDER_REPLICATE', 'BORDER_REFLECT_101', 'BORDER_WRAP'):
    setattr(cv2, name, 0)
sys.modules['cv2'] = cv2
sys.modules.setdefault('matplotlib', types.ModuleType('matplotlib'))

sys.path.insert(0, '/root/reference')
from keras_retinanet_3D.utils import eval as ref_eval  # noqa: E402


class FakeGenerator(object):
    """ the surface utils/eval.py uses; images are constant arrays, `scale` is fixed per image """

    def __init__(self, annotations, scales, planes, num_classes=1):
        self.annotations, self.scales, self.plane_params, self._nc = annotations, scales, planes, num_classes
        self._current = None

    def size(self):
        return len(self.annotations)

    def num_classes(self):
        return self._nc

    def label_to_name(self, label):
        return 'Car'

    def load_image(self, i):
        self._current = i
        return np.zeros((4, 6, 3), np.uint8)

    def preprocess_image(self, image):
        return image.astype(np.float32)

    def resize_image(self, image):
        return image, self.scales[self._current]

    def load_calibration(self, i):
        return np.array([[700.0 + i, 0, 600, 40], [0, 700.0 + i, 170, 0.2], [0, 0, 1, 0.003]])

    def load_annotations(self, i):
        return self.annotations[i], np.zeros((0, 4))


class FakeModel(object):
    def __init__(self, outputs):
        self.outputs, self.calls = outputs, 0

    def predict_on_batch(self, inputs):
        assert inputs[0].shape[0] == 1
        out = [o.copy() for o in self.outputs[self.calls]]
        self.calls += 1
        return out


def make_case(seed, num_images, num_classes=1, tie_scores=False):
    """ annotations + model outputs: detections are jittered copies of most annotations (some with
    the wrong orientation, some duplicated), plus random false positives and sub-threshold rows """
    rng = np.random.default_rng(seed)
    annotations, outputs, scales = [], [], []
    for i in range(num_images):
        n_ann = int(rng.integers(0, 7)) if i != 1 else 0                  # image 1 has no annotations
        x1 = rng.uniform(0, 1000, n_ann)
        y1 = rng.uniform(100, 300, n_ann)
        w = rng.uniform(30, 200, n_ann)
        h = rng.uniform(20, 120, n_ann)
        ann = np.zeros((n_ann, 17))
        ann[:, 0], ann[:, 1], ann[:, 2], ann[:, 3] = x1, y1, x1 + w, y1 + h
        ann[:, 4:12] = rng.uniform(0, 1242, (n_ann, 8))
        ann[:, 12:15] = rng.uniform(1.3, 4.5, (n_ann, 3))
        ann[:, 15] = rng.integers(0, num_classes, n_ann)
        ann[:, 16] = rng.integers(0, 4, n_ann)
        annotations.append(ann)
        scale = float(rng.uniform(0.9, 1.2))
        scales.append(scale)

        rows = []
        for a in ann:
            if rng.random() < 0.8:
                copies = 2 if rng.random() < 0.3 else 1                   # duplicates: second one is a false positive
                for _ in range(copies):
                    jitter = rng.normal(0, 6.0 if rng.random() < 0.8 else 60.0, 4)      # some miss IoU 0.5
                    o = a[16] if rng.random() < 0.85 else (a[16] + 1) % 4               # wrong orientation bin
                    rows.append((a[:4] + jitter, a[4:12] + rng.normal(0, 3, 8), a[12:15] + rng.normal(0, 0.1, 3), a[15], o))
        for _ in range(int(rng.integers(0, 5))):                            # clutter
            bx = rng.uniform(0, 1000)
            by = rng.uniform(100, 300)
            rows.append((np.array([bx, by, bx + rng.uniform(30, 200), by + rng.uniform(20, 120)]), rng.uniform(0, 1242, 8),
                         rng.uniform(1.3, 4.5, 3), rng.integers(0, num_classes), rng.integers(0, 4)))
        n = len(rows)
        assert n <= 100
        scores = rng.uniform(0.02, 0.99, n)
        if tie_scores:
            scores = np.round(scores * 5) / 5 + 0.01                      # many exact ties
        order = np.argsort(-scores)
        boxes = -np.ones((1, 100, 12), np.float32)
        dims = -np.ones((1, 100, 3), np.float32)
        sc = -np.ones((1, 100), np.float32)
        labels = -np.ones((1, 100), np.int32)
        orient = -np.ones((1, 100), np.int32)
        for k, j in enumerate(order):
            b, kp, d, lab, o = rows[j]
            boxes[0, k, :4] = b * scale
            boxes[0, k, 4:] = kp * scale
            dims[0, k] = d
            sc[0, k] = scores[j]
            labels[0, k] = lab
            orient[0, k] = o
        plane_pts = rng.normal(0, 10, (1, 100, 4, 3)).astype(np.float32)
        planes = rng.normal(0, 1, (1, 100, 1, 4)).astype(np.float32)
        residuals = rng.uniform(0, 2, (1, 100)).astype(np.float32)
        outputs.append([boxes, dims, sc, labels, orient, plane_pts, planes, residuals])
    return annotations, outputs, scales


def run_case(name, seed, num_images, out_dir, **kw):
    annotations, outputs, scales = make_case(seed, num_images, **kw)
    planes = np.random.default_rng(seed).normal(size=(10, 4))
    gen = FakeGenerator(annotations, scales, planes, num_classes=kw.get('num_classes', 1))
    result = {}
    for tag, args in (('default', {}), ('strict', {'iou_threshold': 0.7, 'score_threshold': 0.3, 'max_detections': 5})):
        model = FakeModel(outputs)
        aps, ke, he, we, le = ref_eval.evaluate(gen, model, **args)
        labels = sorted(aps.keys())
        result[tag + '_ap'] = np.array([[float(aps[l][0]), float(aps[l][1])] for l in labels])
        result[tag + '_errors'] = np.array([ke, he, we, le], dtype=np.float64)
        dets = ref_eval._get_detections(gen, FakeModel(outputs), score_threshold=args.get('score_threshold', 0.05),
                                        max_detections=args.get('max_detections', 100))
        result[tag + '_det_counts'] = np.array([[d.shape[0] for d in per_image] for per_image in dets])
        result[tag + '_det_concat'] = np.concatenate([d for per_image in dets for d in per_image], axis=0)
    np.savez_compressed(
        os.path.join(out_dir, 'eval_{}.npz'.format(name)), num_classes=np.array(kw.get('num_classes', 1)),
        scales=np.array(scales), planes=planes, ann_counts=np.array([a.shape[0] for a in annotations]),
        annotations=np.concatenate(annotations, axis=0),
        **{'outputs_{}'.format(k): np.concatenate([o[k] for o in outputs], axis=0) for k in range(8)},
        **result)
    print(name, 'AP', result['default_ap'][:, 0].round(4), 'n', result['default_ap'][:, 1], 'errors', result['default_errors'].round(4))


def kitti_parsing_golden(out_dir):
    from keras_retinanet_3D.preprocessing import generator as ref_generator
    from keras_retinanet_3D.preprocessing import kitti as ref_kitti
    ref_generator.Generator.__init__ = lambda self, **kw: None          # skip the TF augmentation graph
    rng = np.random.default_rng(7)
    work = tempfile.mkdtemp()
    try:
        base = os.path.join(work, 'kitti')
        for d in ('images', 'labels', 'calibs'):
            os.makedirs(os.path.join(base, 'val', d))
        scipy.io.savemat(os.path.join(base, 'road_planes_database.mat'), {'road_planes_database': rng.normal(size=(5, 4))})
        from PIL import Image
        label_texts, calib_texts = {}, {}
        for stem, ext in (('000003', '.png'), ('000011', '.jpg'), ('000042', '.png')):
            Image.fromarray(rng.integers(0, 255, (6, 8, 3), dtype=np.uint8)).save(os.path.join(base, 'val', 'images', stem + ext))
            lines = []
            for kind in rng.choice(['Car', 'Van', 'Truck', 'DontCare', 'Misc', 'Pedestrian', 'Car'], size=int(rng.integers(0, 6))):
                vals = ['%.2f' % v for v in rng.uniform(0, 1, 3)] + ['%.2f' % v for v in rng.uniform(0, 1200, 12)] + \
                       ['%.2f' % v for v in rng.uniform(1, 5, 3)] + [str(int(rng.integers(0, 4)))]
                lines.append(' '.join([str(kind)] + vals))
            label_texts[stem] = '\n'.join(lines) + ('\n' if lines else '')
            P = rng.uniform(-1, 800, (4, 12))
            calib_texts[stem] = ''.join('P{}: {}\n'.format(k, ' '.join('%.12e' % v for v in P[k])) for k in range(4))
            open(os.path.join(base, 'val', 'labels', stem + '.txt'), 'w').write(label_texts[stem])
            open(os.path.join(base, 'val', 'calibs', stem + '.txt'), 'w').write(calib_texts[stem])
        gen = ref_kitti.KittiGenerator(base, subset='val')
        saved = {'num_classes': np.array(gen.num_classes()), 'name0': np.array(gen.label_to_name(0)), 'plane_params': gen.plane_params}
        for i in range(gen.size()):
            stem = os.path.basename(gen.images[i])[:6]
            boxes, ignore = gen.load_annotations(i)
            saved['ann_' + stem] = boxes
            saved['ignore_' + stem] = ignore
            saved['P_' + stem] = gen.load_calibration(i)
            saved['label_text_' + stem] = np.array(label_texts[stem])
            saved['calib_text_' + stem] = np.array(calib_texts[stem])
        np.savez_compressed(os.path.join(out_dir, 'eval_kitti_parsing.npz'), **saved)
        print('kitti parsing:', {k: v.shape for k, v in saved.items() if k.startswith('ann_')})
    finally:
        shutil.rmtree(work)


def main():
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    run_case('small', 11, 6, out_dir)
    run_case('ties', 12, 12, out_dir, tie_scores=True)
    run_case('two_classes', 13, 10, out_dir, num_classes=2)
    kitti_parsing_golden(out_dir)


if __name__ == '__main__':
    main()
