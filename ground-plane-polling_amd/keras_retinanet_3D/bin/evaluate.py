#!/usr/bin/env python
"""
Evaluate a model on a KITTI-style directory: mAP over the (class, orientation) bins and the mean
L1 errors of keypoints / height / width / length.

    evaluate.py model_path kitti_dir [--subset val] [--backbone resnet50] [--batch-size N]
                [--iou-threshold 0.5] [--score-threshold 0.05] [--max-detections 100]

The reference has no evaluation script: it evaluates only from the training callback
(callbacks/eval.py:50-114 -> utils/eval.py:168-262, created in bin/train.py).  This CLI runs the
same `evaluate` + the callback's summary on the inference model.
"""

import argparse
import os
import sys

if __name__ == "__main__" and __package__ is None:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    import keras_retinanet_3D.bin  # noqa: F401
    __package__ = "keras_retinanet_3D.bin"

from .. import models
from ..preprocessing.kitti import KittiGenerator
from ..utils.eval import evaluate, summarize


def parse_args(args):
    parser = argparse.ArgumentParser(description='Evaluate a model on a KITTI-style dataset directory.')
    parser.add_argument('model_path', help='Path to inference model (or synthetic:<seed>).', type=str)
    parser.add_argument('kitti_dir', help='Dataset directory (<subset>/images|labels|calibs, road_planes_database.mat).', type=str)
    parser.add_argument('--subset', help='Subset to evaluate.', default='val')
    parser.add_argument('--backbone', help='The backbone of the model to load.', default='resnet50')
    parser.add_argument('--batch-size', help='Images per predict_on_batch call.', type=int, default=1)
    parser.add_argument('--dtype', default='f16x3', choices=['f16x3', 'f32', 'bf16x3', 'f16', 'bf16'],
                        help='Arithmetic of the conv stack (not in the reference CLI).  Default f16x3: the fastest type whose detections, plane '
                             'indices and 3-D corners stay within 1e-3 of the float32 (reference floatx) path; f32 = floatx itself; '
                             'bf16x3 / f16 / bf16 are faster and leave that tolerance.')
    parser.add_argument('--iou-threshold', type=float, default=0.5)
    parser.add_argument('--score-threshold', type=float, default=0.05)
    parser.add_argument('--max-detections', type=int, default=100)
    parser.add_argument('--plane-params-path', help='Plane database (.mat); default <kitti_dir>/road_planes_database.mat.', default=None)
    return parser.parse_args(args)


def main(args=None):
    args = parse_args(sys.argv[1:] if args is None else args)
    model = models.load_model(args.model_path, backbone_name=args.backbone, dtype=args.dtype)
    generator = KittiGenerator(args.kitti_dir, subset=args.subset, plane_params_path=args.plane_params_path)
    results = evaluate(generator, model, iou_threshold=args.iou_threshold, score_threshold=args.score_threshold,
                       max_detections=args.max_detections, batch_size=args.batch_size)
    return summarize(results, generator)


if __name__ == '__main__':
    main()
