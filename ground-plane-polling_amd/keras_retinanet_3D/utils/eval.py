"""
Evaluation harness: mean average precision over the 4 * num_classes
(class, orientation) bins plus the mean L1 errors of the matched keypoints and dimensions --
the host-side counterpart of /root/reference/keras_retinanet_3D/utils/eval.py:

    _compute_ap        utils/eval.py:29-55     (py-faster-rcnn area under the precision envelope)
    _get_detections    utils/eval.py:58-138    (model outputs -> per image, per bin detection rows)
    _get_annotations   utils/eval.py:141-165
    evaluate           utils/eval.py:168-262
    summarize          callbacks/eval.py:60-77 (the numbers the training callback logs)

Same signatures and return values.  What differs is how the work is arranged: images go through
`predict_on_batch` `batch_size` at a time (the reference feeds one image per call), and the
greedy matching of one (image, bin) works on one precomputed overlap matrix instead of one
`compute_overlap` call per detection.  The order in which (score, hit/miss) pairs are collected
is the reference's (bins outermost, then images, then detections in descending score), so the
unstable `np.argsort(-scores)` that ranks them sees the same array and ties fall the same way.

Drawing (`save_path`, utils/eval.py:120-129) needs OpenCV and is out of scope.
"""

from __future__ import print_function

import numpy as np

from .anchors import compute_overlap


def _compute_ap(recall, precision):
    """ Area under the monotone envelope of the precision/recall curve (utils/eval.py:29-55). """
    mrec = np.concatenate(([0.], recall, [1.]))
    mpre = np.concatenate(([0.], precision, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]            # precision envelope, right to left
    step = np.flatnonzero(mrec[1:] != mrec[:-1])
    return np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])


def _image_rows(outputs, k, scale, score_threshold, max_detections):
    """ The (n, 34) detection rows of image k of a batch (utils/eval.py:93-118):
    12 box/keypoint pixels (already divided by scale) | h w l | score | 12 plane points | 4 plane
    coefficients | orientation | label, best score first, at most max_detections. """
    boxes, dimensions, scores, labels, orientations, plane_pts, planes, _ = outputs
    keep = np.where(scores[k, :] > score_threshold)[0]
    s = scores[k][keep]
    order = np.argsort(-s)[:max_detections]
    sel = keep[order]
    return np.concatenate([boxes[k, sel, :] / scale, dimensions[k, sel, :], s[order][:, None],
                           plane_pts[k, sel].reshape(len(sel), 12), planes[k, sel].reshape(len(sel), 4),
                           orientations[k, sel][:, None], labels[k, sel][:, None]], axis=1)


def _get_detections(generator, model, score_threshold=0.05, max_detections=300, save_path=None, batch_size=1):
    """ all_detections[image][4 * label + orientation] = (n, 32) rows: 12 box + 3 dims + score +
    12 plane points + 4 plane coefficients (utils/eval.py:58-138). """
    if save_path is not None:
        raise NotImplementedError('drawing detections needs OpenCV, which this build does not use (DESIGN.md §8)')
    num_bins = 4 * generator.num_classes()
    all_detections = [[None] * num_bins for _ in range(generator.size())]

    def flush(batch):
        if not batch:
            return
        inputs = [np.stack([b[1] for b in batch]), np.stack([b[3] for b in batch]),
                  np.tile(np.asarray(generator.plane_params)[None], (len(batch), 1, 1))]
        outputs = [np.asarray(o) for o in model.predict_on_batch(inputs)[:8]]
        for k, (i, _, scale, _) in enumerate(batch):
            rows = _image_rows(outputs, k, scale, score_threshold, max_detections)
            for label in range(generator.num_classes()):
                for orientation in range(4):
                    pick = np.logical_and(rows[:, -1] == label, rows[:, -2] == orientation)
                    all_detections[i][4 * label + orientation] = rows[pick, :-2]
            print('{}/{}'.format(i + 1, generator.size()), end='\r')

    batch = []
    for i in range(generator.size()):
        image = generator.preprocess_image(generator.load_image(i).copy())
        image, scale = generator.resize_image(image)
        P = np.dot(np.diag([scale, scale, 1.0]), generator.load_calibration(i))
        item = (i, image, scale, np.linalg.pinv(P))
        if batch and (len(batch) >= max(batch_size, 1) or batch[0][1].shape != image.shape):
            flush(batch)
            batch = []
        batch.append(item)
    flush(batch)
    return all_detections


def _get_annotations(generator):
    """ all_annotations[image][4 * label + orientation] = (n, 15) rows: 2D box, 8 keypoint pixels,
    h w l (utils/eval.py:141-165; annotation columns 15/16 are class and orientation). """
    num_bins = 4 * generator.num_classes()
    all_annotations = [[None] * num_bins for _ in range(generator.size())]
    for i in range(generator.size()):
        annotations = generator.load_annotations(i)[0]
        for label in range(generator.num_classes()):
            for orientation in range(4):
                pick = np.logical_and(annotations[:, -2] == label, annotations[:, -1] == orientation)
                all_annotations[i][4 * label + orientation] = annotations[pick, :15].copy()
    return all_annotations


def _match_bin(detections, annotations, iou_threshold):
    """ Greedy assignment inside one (image, bin): every detection, best score first, claims the
    annotation it overlaps most -- a hit if IoU >= threshold and that annotation is still free
    (utils/eval.py:207-226).  Returns (hit flags, per-hit |detection - annotation| over the 8
    keypoint pixels + 3 dimensions). """
    n = detections.shape[0]
    hits = np.zeros((n,), dtype=bool)
    errors = []
    if n == 0 or annotations.shape[0] == 0:
        return hits, errors
    overlaps = compute_overlap(detections[:, :4], annotations[:, :4])
    best = np.argmax(overlaps, axis=1)
    taken = np.zeros((annotations.shape[0],), dtype=bool)
    for d in range(n):
        a = best[d]
        if overlaps[d, a] >= iou_threshold and not taken[a]:
            taken[a] = True
            hits[d] = True
            errors.append(np.absolute(detections[d, 4:15] - annotations[a, 4:15]))
    return hits, errors


def evaluate(generator, model, iou_threshold=0.5, score_threshold=0.05, max_detections=100, save_path=None, batch_size=1):
    """ Evaluate a dataset (utils/eval.py:168-262).  Returns
    (average_precisions {bin: (AP, number of annotations)}, keypoint_error, height_error,
    width_error, length_error); a bin without annotations reports (0, 0). """
    all_detections = _get_detections(generator, model, score_threshold=score_threshold, max_detections=max_detections,
                                     save_path=save_path, batch_size=batch_size)
    all_annotations = _get_annotations(generator)
    average_precisions = {}
    regression_errors = []

    for label in range(4 * generator.num_classes()):
        flags, scores = [], []
        num_annotations = 0.0
        for i in range(generator.size()):
            detections, annotations = all_detections[i][label], all_annotations[i][label]
            num_annotations += annotations.shape[0]
            hits, errors = _match_bin(detections, annotations, iou_threshold)
            flags.append(hits)
            scores.append(detections[:, 15])
            regression_errors.extend(errors)
        if num_annotations == 0:
            average_precisions[label] = 0, 0
            continue
        scores = np.concatenate(scores) if scores else np.zeros((0,))
        hits = np.concatenate(flags).astype(np.float64) if flags else np.zeros((0,))
        rank = np.argsort(-scores)
        true_positives = np.cumsum(hits[rank])
        false_positives = np.cumsum(1.0 - hits[rank])
        recall = true_positives / num_annotations
        precision = true_positives / np.maximum(true_positives + false_positives, np.finfo(np.float64).eps)
        average_precisions[label] = _compute_ap(recall, precision), num_annotations

    if len(regression_errors) == 0:
        return average_precisions, 0, 0, 0, 0
    regression_errors = np.vstack(regression_errors)
    return (average_precisions, np.average(regression_errors[:, :8]), np.average(regression_errors[:, 8]),
            np.average(regression_errors[:, 9]), np.average(regression_errors[:, 10]))


def summarize(results, generator=None, verbose=1):
    """ The numbers the reference's Evaluate callback logs from `evaluate`'s return value
    (callbacks/eval.py:60-77,105-114): mAP = mean AP over the bins that have annotations. """
    average_precisions, keypoint_error, height_error, width_error, length_error = results
    present, total = 0, 0.0
    for label, (average_precision, num_annotations) in average_precisions.items():
        if verbose == 1:
            name = generator.label_to_name(int(label / 4)) if generator is not None else 'class {}'.format(int(label / 4))
            print('{:.0f} instances of class'.format(num_annotations), name,
                  'with average precision: {:.4f}'.format(average_precision))
        if num_annotations > 0:
            present += 1
            total += average_precision
    logs = {'mAP': total / present,                                   # ZeroDivisionError with no annotations, as upstream
            'keypoints (mean L1 error)': keypoint_error, 'height (mean L1 error)': height_error,
            'width (mean L1 error)': width_error, 'length (mean L1 error)': length_error}
    if verbose == 1:
        print('mAP: {:.4f}'.format(logs['mAP']))
        for key in ('keypoints', 'height', 'width', 'length'):
            print('{} (mean L1 error): {:.2f}'.format(key, logs[key + ' (mean L1 error)']))
    return logs
