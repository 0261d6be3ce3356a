"""
Host-side image handling of the inference harness (reference utils/image.py:26-62,174-200):
read as BGR, subtract the ImageNet mean, resize so that the short side is 800 unless the long
side would exceed 1333.  The augmentation helpers of the reference file are training-only and
out of scope.

cv2 is not available in this image; `resize_image` restates cv2.resize(img, None, fx=s, fy=s)
(INTER_LINEAR: dsize = round(size * s), source coordinate (d + 0.5) / s - 0.5, border replicated)
in NumPy.  It cannot be checked against OpenCV here.
"""

import numpy as np

IMAGENET_MEAN_BGR = (103.939, 116.779, 123.68)      # utils/image.py:58-60


def read_image_bgr(path):
    """ Read an image in BGR format (utils/image.py:26-33). """
    from PIL import Image
    image = np.asarray(Image.open(path).convert('RGB'))
    return image[:, :, ::-1].copy()


def preprocess_image(x):
    """ float32, ImageNet mean subtracted per BGR channel (utils/image.py:36-62, channels_last). """
    x = x.astype(np.float32)
    x[..., 0] -= 103.939
    x[..., 1] -= 116.779
    x[..., 2] -= 123.68
    return x


def compute_resize_scale(image_shape, min_side=800, max_side=1333):
    """ utils/image.py:184-195 """
    rows, cols = image_shape[0], image_shape[1]
    scale = min_side / min(rows, cols)
    if max(rows, cols) * scale > max_side:
        scale = max_side / max(rows, cols)
    return scale


def _axis_taps(dst_size, src_size, scale):
    """ bilinear taps along one axis: indices i0, i1 and weight of i1 (float32) """
    s = (np.arange(dst_size, dtype=np.float64) + 0.5) / scale - 0.5
    i0 = np.floor(s).astype(np.int64)
    w1 = (s - i0).astype(np.float32)
    lo = i0 < 0
    i0 = np.where(lo, 0, i0)
    w1 = np.where(lo, np.float32(0.0), w1)
    hi = i0 >= src_size - 1
    i1 = np.where(hi, src_size - 1, i0 + 1)
    i0 = np.where(hi, src_size - 1, i0)
    w1 = np.where(hi, np.float32(0.0), w1)
    return i0, i1, w1.astype(np.float32)


def resize_bilinear(img, scale):
    """ cv2.resize(img, None, fx=scale, fy=scale) for float32 HWC input (INTER_LINEAR) """
    rows, cols = img.shape[:2]
    out_r, out_c = int(np.rint(rows * scale)), int(np.rint(cols * scale))
    y0, y1, wy = _axis_taps(out_r, rows, scale)
    x0, x1, wx = _axis_taps(out_c, cols, scale)
    img = np.asarray(img, dtype=np.float32)
    top = img[y0][:, x0] * (1 - wx)[None, :, None] + img[y0][:, x1] * wx[None, :, None]
    bot = img[y1][:, x0] * (1 - wx)[None, :, None] + img[y1][:, x1] * wx[None, :, None]
    return (top * (1 - wy)[:, None, None] + bot * wy[:, None, None]).astype(np.float32)


def resize_image(img, min_side=800, max_side=1333):
    """ (resized image, scale), utils/image.py:174-200 """
    scale = compute_resize_scale(img.shape, min_side, max_side)
    return resize_bilinear(img, scale), scale
