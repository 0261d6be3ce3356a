"""
ORACLE SUPPORT (test infrastructure): a NumPy stand-in for the few dozen
`keras.backend` / `tensorflow` primitives that the reference's layer code calls, so that
the reference's own Python files can be *executed unmodified* in this container (which has
neither TensorFlow nor Keras) to produce golden vectors.

It is used only by oracle/gen_*_goldens.py, here, where /root/reference exists.  Nothing in
the product or in the GPU tests imports it.  The primitive list is the alias table in
/root/reference/keras_retinanet_3D/backend/tensorflow_backend.py:20-156 plus the
keras.backend calls made by layers/fit_road_planes.py, layers/filter_detections.py,
layers/_misc.py and backend/common.py.

Semantics restated from the TF 1.x documentation (unverifiable here):
  * argmax/argmin return the first extremum; argmin never selects NaN
  * tf.where(cond) -> int64 indices (K, rank); tf.where(c, a, b) -> select
  * tf.one_hot of an out-of-range index (-1) is an all-zero row
  * tf.nn.top_k: descending, ties keep the lower index first
  * tf.image.non_max_suppression: greedy by descending score, suppress IoU > threshold
  * tf.image.resize_images NEAREST, align_corners=False: src = min(floor(dst*in/out), in-1)
"""

import sys
import types

import numpy as np

FLOATX = 'float32'


class _T(np.ndarray):
    """ ndarray that tolerates Tensor.set_shape (filter_detections.py:180-187). """

    def set_shape(self, shape):
        pass


def _t(x):
    return np.asarray(x).view(_T)


RECORD = {}     # side channel: last argmin result (the reference never returns the plane index)


# ----------------------------------------------------------------------------- keras.backend
def _kb():
    kb = types.ModuleType('keras.backend')
    kb.floatx = lambda: FLOATX
    kb.image_data_format = lambda: 'channels_last'
    kb.backend = lambda: 'tensorflow'
    kb.abs = np.abs
    kb.greater = np.greater
    kb.less = np.less
    kb.zeros_like = np.zeros_like
    kb.ones_like = np.ones_like
    kb.permute_dimensions = lambda x, p: np.transpose(x, p)
    kb.transpose = np.transpose
    kb.shape = lambda x: np.array(np.shape(x), dtype=np.int64)
    kb.int_shape = lambda x: tuple(np.shape(x))
    kb.sign = np.sign
    kb.reshape = lambda x, s: np.reshape(x, [int(v) for v in s])
    kb.concatenate = lambda xs, axis=-1: np.concatenate(xs, axis=axis)
    kb.ones = lambda shape, dtype=None: np.ones([int(v) for v in shape], dtype=dtype or FLOATX)
    kb.zeros = lambda shape, dtype=None: np.zeros([int(v) for v in shape], dtype=dtype or FLOATX)
    kb.tile = lambda x, n: np.tile(x, [int(v) for v in n])
    kb.expand_dims = lambda x, axis=-1: np.expand_dims(x, axis)
    kb.sum = lambda x, axis=None, keepdims=False: np.sum(x, axis=axis, keepdims=keepdims)
    kb.max = lambda x, axis=None, keepdims=False: np.max(x, axis=axis, keepdims=keepdims)
    kb.minimum = np.minimum
    kb.maximum = np.maximum
    kb.stack = lambda xs, axis=0: np.stack(xs, axis=axis)
    kb.cast = lambda x, dtype: _t(np.asarray(x).astype(dtype))
    kb.constant = lambda v, dtype=None: np.asarray(v, dtype=dtype or FLOATX)
    kb.variable = lambda v, dtype=None: np.asarray(v, dtype=dtype or FLOATX)
    kb.arange = lambda a, b=None, dtype='int32': np.arange(a, b).astype(dtype)

    def argmin(x, axis=-1):
        key = np.where(x < np.finfo(x.dtype).max, x, np.inf) if x.dtype.kind == 'f' else x
        r = np.argmin(key, axis=axis).astype(np.int64)
        RECORD['argmin'] = r
        return r

    kb.argmin = argmin
    kb.argmax = lambda x, axis=-1: np.argmax(x, axis=axis).astype(np.int64)
    return kb


# ----------------------------------------------------------------------------- tensorflow
def _iou(a, b):
    # TF 1.x non_max_suppression_op.cc: corners are min/max-normalised first, zero-area -> 0
    y0a, x0a, y1a, x1a = min(a[0], a[2]), min(a[1], a[3]), max(a[0], a[2]), max(a[1], a[3])
    y0b, x0b, y1b, x1b = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])
    f = np.float32
    area_a = f(f(y1a - y0a) * f(x1a - x0a))
    area_b = f(f(y1b - y0b) * f(x1b - x0b))
    if area_a <= 0 or area_b <= 0:
        return f(0.0)
    ih = max(f(min(y1a, y1b) - max(y0a, y0b)), f(0.0))
    iw = max(f(min(x1a, x1b) - max(x0a, x0b)), f(0.0))
    inter = f(ih * iw)
    return f(inter / f(f(area_a + area_b) - inter))


def _nms(boxes, scores, max_output_size, iou_threshold=0.5):
    boxes = np.asarray(boxes, dtype=np.float32)
    order = np.argsort(-np.asarray(scores), kind='stable')
    keep = []
    for i in order:
        if len(keep) >= int(max_output_size):
            break
        if all(not (_iou(boxes[i], boxes[j]) > np.float32(iou_threshold)) for j in keep):
            keep.append(int(i))
    return np.asarray(keep, dtype=np.int32)


def _tf():
    tf = types.ModuleType('tensorflow')
    tf.norm = lambda x, axis=None, keep_dims=False: np.sqrt(np.sum(x * x, axis=axis, keepdims=keep_dims))

    def where(cond, a=None, b=None):
        if a is None:
            return np.argwhere(cond).astype(np.int64)
        return np.where(cond, a, b)

    tf.where = where
    tf.cross = lambda a, b: np.cross(a, b).astype(np.result_type(a, b))
    tf.matmul = np.matmul
    tf.multiply = np.multiply
    tf.divide = np.divide
    tf.gather = lambda p, i, axis=0: np.take(p, np.asarray(i), axis=axis)
    tf.gather_nd = lambda p, i: np.asarray(p)[tuple(np.asarray(i)[..., k] for k in range(np.asarray(i).shape[-1]))]

    def one_hot(i, depth, dtype=FLOATX):
        i = np.asarray(i)
        return (i[..., None] == np.arange(depth)).astype(dtype)

    tf.one_hot = one_hot

    def map_fn(fn, elems, dtype=None, parallel_iterations=None):
        single = not isinstance(elems, (list, tuple))
        n = len(elems) if single else len(elems[0])
        outs = [fn(elems[k] if single else [e[k] for e in elems]) for k in range(n)]
        if isinstance(outs[0], (list, tuple)):
            return [np.stack([o[j] for o in outs]) for j in range(len(outs[0]))]
        return np.stack(outs)

    tf.map_fn = map_fn
    tf.range = lambda *a, **k: np.arange(*[int(v) for v in a]).astype(np.asarray(a[-1]).dtype if hasattr(a[-1], 'dtype') else np.int32)
    tf.meshgrid = np.meshgrid
    tf.clip_by_value = np.clip

    def pad(x, paddings, constant_values=0):
        paddings = [[int(a), int(b)] for a, b in paddings]
        return _t(np.pad(x, paddings, mode='constant', constant_values=constant_values))

    tf.pad = pad
    tf.nn = types.SimpleNamespace()

    def top_k(x, k):
        order = np.argsort(-np.asarray(x), kind='stable')[:int(k)]
        return _t(np.asarray(x)[order]), order.astype(np.int32)

    tf.nn.top_k = top_k
    tf.image = types.SimpleNamespace()
    tf.image.non_max_suppression = _nms
    tf.image.ResizeMethod = types.SimpleNamespace(BILINEAR=0, NEAREST_NEIGHBOR=1, BICUBIC=2, AREA=3)

    def resize_images(images, size, method=0, align_corners=False):
        assert method == 1 and not align_corners
        images = np.asarray(images)
        ih, iw = images.shape[1:3]
        oh, ow = int(size[0]), int(size[1])
        ys = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * (np.float32(ih) / np.float32(oh))).astype(np.int64), ih - 1)
        xs = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * (np.float32(iw) / np.float32(ow))).astype(np.int64), iw - 1)
        return images[:, ys][:, :, xs]

    tf.image.resize_images = resize_images
    tf.Graph = tf.Session = lambda *a, **k: None
    return tf


def install():
    """ Put the stand-ins into sys.modules as `keras`, `keras.backend`, `keras.layers`,
    `tensorflow`.  Returns (keras, tensorflow). """
    keras = types.ModuleType('keras')
    kb = _kb()
    kl = types.ModuleType('keras.layers')

    class Layer(object):
        def __init__(self, *args, **kwargs):
            self.name = kwargs.get('name')

        def get_config(self):
            return {}

        def __call__(self, inputs, **kwargs):
            return self.call(inputs, **kwargs)

    kl.Layer = Layer
    keras.backend = kb
    keras.layers = kl
    tf = _tf()
    sys.modules['keras'] = keras
    sys.modules['keras.backend'] = kb
    sys.modules['keras.layers'] = kl
    sys.modules['tensorflow'] = tf
    return keras, tf
