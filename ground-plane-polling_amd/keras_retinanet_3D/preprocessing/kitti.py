"""
KITTI dataset reader for inference and evaluation: the part of the reference's
preprocessing/kitti.py KittiGenerator (:37-196) and preprocessing/generator.py (:200-207) that
utils/eval.py touches -- size, num_classes, label_to_name, load_image, load_annotations,
load_calibration, plane_params, preprocess_image, resize_image.  The training side of the
generator (batching, augmentation, anchor targets) is out of scope.

Directory layout (kitti.py:60-63):
    <base_dir>/<subset>/images/<id>.png|.jpg
    <base_dir>/<subset>/labels/<id>.txt     20 space-separated fields per object (below)
    <base_dir>/<subset>/calibs/<id>.txt     KITTI calibration file; the third line (P2) is used
    <base_dir>/road_planes_database.mat     key 'road_planes_database', (N, 4)

Label fields (kitti.py:98-99): type truncated occluded alpha left top right bottom xl yl xm ym xr
yr xt yt height width length orientation.  'Car' and 'Van' are class 0 (kitti.py:28-35),
'DontCare' / 'Misc' rows become ignore regions, every other type is dropped.
"""

import os

import numpy as np
import scipy.io

from ..utils.image import preprocess_image, read_image_bgr, resize_image

kitti_classes = {'Car': 0, 'Van': 0}
IGNORED_TYPES = ('DontCare', 'Misc')


def parse_label_file(path):
    """ -> (annotations (n, 17) float64: x1 y1 x2 y2 xl yl xm ym xr yr xt yt h w l class
    orientation, ignore boxes (m, 4) float64), kitti.py:100-119,154-186 """
    objects, ignore = [], []
    with open(path, 'r') as f:
        for line in f:
            fields = line.split(' ')          # single-space delimiter, like the reference's csv reader
            fields[-1] = fields[-1].rstrip('\r\n')
            if not fields or fields == ['']:
                continue
            kind = fields[0]
            if kind in IGNORED_TYPES:
                ignore.append([float(v) for v in fields[4:8]])
            elif kind in kitti_classes:
                objects.append([float(v) for v in fields[4:19]] + [kitti_classes[kind], int(fields[19])])
    return (np.asarray(objects, dtype=np.float64).reshape(-1, 17),
            np.asarray(ignore, dtype=np.float64).reshape(-1, 4))


class KittiGenerator(object):
    """ Read-only view of a KITTI-style directory for `utils.eval.evaluate`. """

    def __init__(self, base_dir, subset='train', image_min_side=800, image_max_side=1333, plane_params_path=None):
        self.base_dir = base_dir
        self.subset = subset
        self.image_min_side = image_min_side
        self.image_max_side = image_max_side
        image_dir = os.path.join(base_dir, subset, 'images')
        label_dir = os.path.join(base_dir, subset, 'labels')
        calib_dir = os.path.join(base_dir, subset, 'calibs')
        if plane_params_path is None:
            plane_params_path = os.path.join(base_dir, 'road_planes_database.mat')
        self.plane_params = scipy.io.loadmat(plane_params_path)['road_planes_database']
        self.id_to_labels = {v: k for k, v in kitti_classes.items()}      # the last name wins, as upstream
        self.images, self.calibs, self.image_data, self.ignore_regions = [], [], {}, {}
        for i, fn in enumerate(os.listdir(image_dir)):
            stem = fn.replace('.png', '.txt').replace('.jpg', '.txt')
            self.images.append(os.path.join(image_dir, fn))
            self.calibs.append(os.path.join(calib_dir, stem))
            self.image_data[i], self.ignore_regions[i] = parse_label_file(os.path.join(label_dir, stem))

    def size(self):
        return len(self.images)

    def num_classes(self):
        return max(kitti_classes.values()) + 1

    def name_to_label(self, name):
        raise NotImplementedError()

    def label_to_name(self, label):
        return self.id_to_labels[label]

    def image_aspect_ratio(self, image_index):
        from PIL import Image
        image = Image.open(self.images[image_index])
        return float(image.width) / float(image.height)

    def load_image(self, image_index):
        return read_image_bgr(self.images[image_index])

    def load_annotations(self, image_index):
        return self.image_data[image_index].copy(), self.ignore_regions[image_index].copy()

    def load_calibration(self, image_index):
        """ the 3x4 projection matrix of camera 2 (kitti.py:188-196) """
        with open(self.calibs[image_index], 'r') as f:
            line = f.readlines()[2]
        return np.array([float(x) for x in line.split(':', 1)[1].split()]).reshape((3, 4))

    def preprocess_image(self, image):
        return preprocess_image(image)

    def resize_image(self, image):
        return resize_image(image, min_side=self.image_min_side, max_side=self.image_max_side)
