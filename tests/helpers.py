""" Shared helpers for the test-suite (oracle access, golden loading). """
import ctypes
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def polling_golden_names():
    return sorted(os.path.basename(p)[len('polling_'):-len('.npz')] for p in glob.glob(os.path.join(GOLDEN, 'polling_*.npz')))


def load_polling_golden(name):
    from keras_retinanet_3D.utils import synthetic
    g = dict(np.load(os.path.join(GOLDEN, 'polling_{}.npz'.format(name))))
    g['planes'] = synthetic.load_plane_database(str(g['db'])).astype(np.float32)
    return g


def c_oracle_poll(lib, boxes, dims, orient, P_inv, planes, thr=0.7):
    """ oracle/polling.c through ctypes; planes (N,4) shared or (B,N,4). """
    boxes = np.ascontiguousarray(boxes, np.float32)
    dims = np.ascontiguousarray(dims, np.float32)
    orient = np.ascontiguousarray(orient, np.int32)
    P_inv = np.ascontiguousarray(P_inv, np.float32)
    planes = np.ascontiguousarray(planes, np.float32)
    B, D = boxes.shape[:2]
    batched = int(planes.ndim == 3)
    N = planes.shape[-2]
    kp = np.empty((B, D, 4, 3), np.float32)
    kpl = np.empty((B, D, 1, 4), np.float32)
    res = np.empty((B, D), np.float32)
    idx = np.empty((B, D), np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    lib.gpp_oracle_poll_f32.restype = ctypes.c_int
    rc = lib.gpp_oracle_poll_f32(p(boxes), p(dims), p(orient), p(P_inv), p(planes), B, D, N, batched,
                                 ctypes.c_float(thr), p(kp), p(kpl), p(res), p(idx))
    assert rc == 0
    return kp, kpl, res, idx


def bits_equal(a, b):
    """ bitwise equality of float arrays (NaN == NaN, +0 != -0 is tolerated as equal). """
    a = np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))
