""" CPU: the C-ABI library loads and exports every symbol that include/gpp.h declares. """
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'gpp.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gpp_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from keras_retinanet_3D.backend import hip
    if not os.path.isfile(hip.LIB_PATH):
        hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    names = declared_symbols()
    assert 'gpp_poll_f32' in names
    for name in names:
        assert hasattr(lib, name), 'libgpp_hip.so does not export {}'.format(name)
    lib.gpp_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.gpp_version()


def test_product_path_fails_loudly_without_a_gpu():
    import numpy as np
    import pytest
    import torch
    from keras_retinanet_3D.backend import hip
    from keras_retinanet_3D.utils import gpp_utils
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    with pytest.raises(hip.GppError):
        gpp_utils.fit_road_planes(np.zeros((1, 1, 12), np.float32), np.zeros((1, 1, 3), np.float32),
                                  np.zeros((1, 1), np.int32), np.zeros((1, 4, 3), np.float32), np.ones((4, 4), np.float32))
