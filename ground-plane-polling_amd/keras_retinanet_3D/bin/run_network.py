#!/usr/bin/env python
"""
Run the network on a directory of images -- the MI355X counterpart of the reference harness
/root/reference/keras_retinanet_3D/bin/run_network.py (same positional arguments, flags, output
tree and file formats):

    run_network.py model_path image_dir calib_dir plane_params_path output_dir
                   [--kitti] [--save-images] [--backbone resnet50] [--batch-size N]

    <output_dir>/<model name>/outputs/full/<image>.mat     boxes keypoints labels scores locations
                                                           angles dimensions residuals  (:291-292)
    <output_dir>/<model name>/outputs/kitti/<image>.txt    KITTI result lines           (:295-330)
    <output_dir>/<model name>/images/composite/<image>.png only with --save-images and cv2 present

Differences, all on the host side: images are processed in batches (--batch-size, default 1 =
the reference's behaviour), the per-detection Python loop of :137-287 is vectorised
(utils.gpp_utils.recover_pose), and `model_path` may be 'synthetic:<seed>' because no trained
weights ship with the reference.
"""

import argparse
import os
import shutil
import sys
import time

# Allow relative imports when being executed as script.
if __name__ == "__main__" and __package__ is None:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    import keras_retinanet_3D.bin  # noqa: F401
    __package__ = "keras_retinanet_3D.bin"

import numpy as np
import scipy.io

from .. import models
from ..utils import gpp_utils
from ..utils.image import compute_resize_scale, preprocess_image, read_image_bgr, resize_image


def parse_args(args):
    parser = argparse.ArgumentParser(description='Simple script for running the network on a directory of images.')
    parser.add_argument('model_path', help='Path to inference model (or synthetic:<seed>).', type=str)
    parser.add_argument('image_dir', help='Path to directory of input images.', type=str)
    parser.add_argument('calib_dir', help='Path to directory of calibration files.', type=str)
    parser.add_argument('plane_params_path', help='Path to .MAT file containing road planes.', type=str)
    parser.add_argument('output_dir', help='Path to output directory', type=str)
    parser.add_argument('--kitti', help='Include to save results in KITTI format.', action='store_true')
    parser.add_argument('--save-images', help='Include to save result images.', action='store_true')
    parser.add_argument('--backbone', help='The backbone of the model to load.', default='resnet50')
    parser.add_argument('--batch-size', help='Images per predict_on_batch call.', type=int, default=1)
    parser.add_argument('--dtype', default='f16x3', choices=['f16x3', 'f32', 'bf16x3', 'f16', 'bf16'],
                        help='Arithmetic of the conv stack (not in the reference CLI).  Default f16x3: the fastest type whose detections, plane '
                             'indices and 3-D corners stay within 1e-3 of the float32 (reference floatx) path; f32 = floatx itself; '
                             'bf16x3 / f16 / bf16 are faster and leave that tolerance.')
    return parser.parse_args(args)


def make_output_tree(args):
    name = os.path.basename(args.model_path)[:-3]
    output_dir = os.path.join(args.output_dir, name)
    if os.path.isdir(output_dir):
        shutil.rmtree(output_dir)
    os.makedirs(os.path.join(output_dir, 'outputs', 'full'))
    if args.kitti:
        os.mkdir(os.path.join(output_dir, 'outputs', 'kitti'))
    if args.save_images:
        os.makedirs(os.path.join(output_dir, 'images', 'composite'))
    return output_dir


def load_item(args, fn, on_device):
    """ everything the reference does per image before the timer starts (:91-105); with a model that
    preprocesses on the GPU (predict_on_frames) only the raw frame and the scale are prepared here """
    image_fp = os.path.join(args.image_dir, fn.replace('.txt', '.png'))
    raw_image = read_image_bgr(image_fp)
    if on_device:
        image, scale = None, compute_resize_scale(raw_image.shape)
    else:
        image, scale = resize_image(preprocess_image(raw_image))
    P, P_inv = gpp_utils.load_calibration(os.path.join(args.calib_dir, fn), scale)
    return {'image_fp': image_fp, 'raw_image': raw_image, 'image': image, 'scale': scale, 'P': P, 'P_inv': P_inv}


def write_results(args, output_dir, item, det):
    stem = os.path.basename(item['image_fp'])[:-3]
    outputs = {'boxes': det['boxes'][:, :4], 'keypoints': det['boxes'][:, 4:], 'labels': det['labels'], 'scores': det['scores'],
               'locations': det['locations'], 'angles': det['angles'], 'dimensions': det['dimensions'], 'residuals': det['residuals']}
    scipy.io.savemat(os.path.join(output_dir, 'outputs', 'full', stem + 'mat'), outputs)
    if args.kitti:
        with open(os.path.join(output_dir, 'outputs', 'kitti', stem + 'txt'), 'w') as f:
            f.writelines(gpp_utils.kitti_lines(det, item['raw_image'].shape))
    if args.save_images:
        try:
            import cv2  # noqa: F401
        except ImportError:
            print('--save-images needs OpenCV (cv2), which is not installed: skipping the composite image')


def main(args=None):
    if args is None:
        args = sys.argv[1:]
    args = parse_args(args)

    model = models.load_model(args.model_path, backbone_name=args.backbone, dtype=args.dtype)
    plane_params = scipy.io.loadmat(args.plane_params_path)['road_planes_database']
    output_dir = make_output_tree(args)

    files = os.listdir(args.calib_dir)
    j = 0
    for start in range(0, len(files), max(args.batch_size, 1)):
        on_device = hasattr(model, 'predict_on_frames')
        items = [load_item(args, fn, on_device) for fn in files[start:start + max(args.batch_size, 1)]]
        # images of one batch must share a shape (KITTI frames of one drive do); split otherwise
        groups = {}
        for it in items:
            groups.setdefault(it['raw_image'].shape, []).append(it)
        for group in groups.values():
            P_inv = np.stack([it['P_inv'] for it in group])
            planes = np.tile(plane_params[None], (len(group), 1, 1))
            t0 = time.time()
            if on_device:
                outputs = model.predict_on_frames(np.stack([it['raw_image'] for it in group]), P_inv, planes)[0][:8]
            else:
                outputs = model.predict_on_batch([np.stack([it['image'] for it in group]), P_inv, planes])[:8]
            dt = time.time() - t0
            for k, it in enumerate(group):
                print("Image {}: frame rate: {:.2f}".format(j, len(group) / dt))
                j += 1
                det = gpp_utils.recover_pose(gpp_utils.select_detections(outputs, it['scale'], image_index=k))
                write_results(args, output_dir, it, det)


if __name__ == '__main__':
    main()
