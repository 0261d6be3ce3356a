"""
ORACLE SUPPORT (test infrastructure): generate tests/golden/decode_*.npz and anchors_402x1333.npz by
executing the reference's own layers/_misc.py (Anchors, RegressBoxes, RegressDims),
backend/common.py (shift, bbox_transform_inv, dim_transform_inv), layers/filter_detections.py and
utils/anchors.py, UNMODIFIED, on the NumPy stand-in of oracle/np_tf_shim.py.

Run here only (needs /root/reference):   python oracle/gen_decode_goldens.py

What the stand-in supplies (and therefore what these goldens do NOT pin): the TF primitives
tf.image.non_max_suppression, tf.nn.top_k, tf.pad, tf.where/gather/gather_nd and the sigmoid.
What they do pin: the reference's own glue -- anchor generation and ordering, the sign rule,
the 12-value box decode, the 8 -> (orientation, score) fold, the index plumbing around NMS,
the re-gather of scores, the padding and the output order.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

import np_tf_shim  # noqa: E402

np_tf_shim.install()
sys.path.insert(0, '/root/reference')
from keras_retinanet_3D import layers as ref_layers  # noqa: E402
from keras_retinanet_3D.layers.filter_detections import filter_detections as ref_filter  # noqa: E402
from keras_retinanet_3D.utils import anchors as ref_anchors  # noqa: E402

import decode_np  # noqa: E402

F = np.float32
RATIOS = np.array([0.5, 1, 2], F)
SCALES = np.array([2 ** (-2.0 / 3.0), 2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)], F)


def ref_anchor_layers(image_hw, batch):
    """ the five Anchors layers of retinanet.py:284-311 applied to dummy P3..P7 features """
    out = []
    for (fh, fw), size, stride in zip(decode_np.pyramid_shapes(image_hw), [32, 64, 128, 256, 512], [8, 16, 32, 64, 128]):
        layer = ref_layers.Anchors(size=size, stride=stride, ratios=RATIOS, scales=SCALES)
        out.append(layer.call(np.zeros((batch, fh, fw, 1), F)))
    return np.concatenate(out, axis=1)


def make_case(image_hw, batch, seed, logit_mean, logit_std):
    rng = np.random.default_rng(seed)
    anchors = ref_anchor_layers(image_hw, batch).astype(F)
    A = anchors.shape[1]
    logits = rng.normal(logit_mean, logit_std, size=(batch, A, 8)).astype(F)
    regression = rng.normal(0.0, 1.0, size=(batch, A, 12)).astype(F)
    regression_dim = rng.normal(0.0, 1.0, size=(batch, A, 3)).astype(F)
    cls = decode_np.sigmoid(logits)
    boxes = ref_layers.RegressBoxes(mean=decode_np.BOX_MEAN.astype(F), std=decode_np.BOX_STD.astype(F)).call([anchors, regression, cls])
    dims = ref_layers.RegressDims(mean=decode_np.DIM_MEAN.astype(F), std=decode_np.DIM_STD.astype(F)).call(regression_dim)
    outs = [ref_filter(boxes[b], dims[b], cls[b]) for b in range(batch)]
    det = [np.stack([np.asarray(o[k]) for o in outs]) for k in range(5)]
    outs2 = [ref_filter(boxes[b], dims[b], cls[b], nms=False) for b in range(batch)]          # load_model(..., nms=False)
    det2 = [np.stack([np.asarray(o[k]) for o in outs2]) for k in range(5)]
    outs3 = [ref_filter(boxes[b], dims[b], cls[b], orientation_specific_filter=True) for b in range(batch)]
    det3 = [np.stack([np.asarray(o[k]) for o in outs3]) for k in range(5)]
    outs4 = [ref_filter(boxes[b], dims[b], cls[b], orientation_specific_filter=True, nms=False) for b in range(batch)]
    det4 = [np.stack([np.asarray(o[k]) for o in outs4]) for k in range(5)]
    return dict(image_hw=np.array(image_hw), anchors=anchors[0], logits=logits, classification=cls, regression=regression,
                regression_dim=regression_dim, all_boxes=boxes.astype(F), all_dims=dims.astype(F),
                boxes=det[0].astype(F), dimensions=det[1].astype(F), scores=det[2].astype(F),
                labels=det[3].astype(np.int32), orientations=det[4].astype(np.int32),
                nonms_boxes=det2[0].astype(F), nonms_dimensions=det2[1].astype(F), nonms_scores=det2[2].astype(F),
                nonms_labels=det2[3].astype(np.int32), nonms_orientations=det2[4].astype(np.int32),
                osf_boxes=det3[0].astype(F), osf_dimensions=det3[1].astype(F), osf_scores=det3[2].astype(F),
                osf_labels=det3[3].astype(np.int32), osf_orientations=det3[4].astype(np.int32),
                osfnonms_boxes=det4[0].astype(F), osfnonms_dimensions=det4[1].astype(F), osfnonms_scores=det4[2].astype(F),
                osfnonms_labels=det4[3].astype(np.int32), osfnonms_orientations=det4[4].astype(np.int32))


def main():
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    cases = {
        'small': make_case((64, 96), 2, 1, -3.2, 1.2),       # some hundred candidates per image, heavy overlap
        'sparse': make_case((40, 72), 3, 2, -6.0, 1.0),      # fewer than 100 survivors -> padding rows
        'none': make_case((33, 40), 1, 3, -12.0, 0.5),       # nothing above the threshold -> all -1
        'dense': make_case((96, 160), 1, 4, -1.0, 2.0),      # thousands of candidates, > 100 survivors
    }
    for name, c in cases.items():
        np.savez_compressed(os.path.join(out_dir, 'decode_{}.npz'.format(name)), **c)
        kept = (c['scores'] > 0).sum(axis=1)
        cand = (np.maximum(c['classification'][..., :4], c['classification'][..., 4:]).max(-1) > 0.05).sum(axis=1)
        print(name, 'anchors', c['anchors'].shape[0], 'candidates', cand, 'kept', kept)

    # full-resolution anchors: the NumPy generator of the reference (float64) and the graph-side twin
    a64 = ref_anchors.anchors_for_shape((402, 1333, 3))
    a32 = ref_anchor_layers((402, 1333), 1)[0].astype(F)
    print('anchors', a64.shape, 'max |f32 twin - f64|', np.abs(a32 - a64).max())
    np.savez_compressed(os.path.join(out_dir, 'anchors_402x1333.npz'),
                        count=np.array(a64.shape[0]), first=a64[:24], last=a64[-24:],
                        rows_every_1009=a64[::1009],
                        sha256_f32_twin=np.array(hashlib.sha256(np.ascontiguousarray(a32).tobytes()).hexdigest()),
                        sha256_f64_cast_f32=np.array(hashlib.sha256(np.ascontiguousarray(a64.astype(F)).tobytes()).hexdigest()))


if __name__ == '__main__':
    main()
