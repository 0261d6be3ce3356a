"""
CPU test of the shelved Winograd weight packing (was tests/test_round5_cpu.py): run from this directory,
  PYTHONPATH=../../../ground-plane-polling_amd:../../.. python -m pytest test_wino_packing_cpu.py
"""
import numpy as np
import torch

import conv_wino_host as C
from keras_retinanet_3D.layers.conv import weight_row_order
C.weight_row_order = weight_row_order


def test_winograd_weight_packing_is_the_transformed_kernel_in_the_kernels_layout():
    rng = np.random.default_rng(0)
    cin, cout = 64, 128
    k = (rng.standard_normal((3, 3, cin, cout)) * np.logspace(-3, 1, cout)[None, None, None, :]).astype(np.float32)     # channels four decades apart
    w, inv_scale = C.pack_weight_wino(k, 'cpu')
    assert tuple(w.shape) == (cout, 4 * (cin // 32) * 3 * 32) and tuple(inv_scale.shape) == (4, cout)
    halves = w.view(torch.float16).reshape(cout, 4, cin // 32, 3, 2, 32).float()               # [row][position][chunk][kernel row][hi | lo][32]
    val = (halves[..., 0, :] + halves[..., 1, :]).numpy()
    U = np.einsum('pw,hwcn->phcn', C.WINO_G, k.astype(np.float64))                              # (position, kernel row, c, n)
    rows = C.weight_row_order(cout).numpy()
    for row in (0, 17, 37, 127):
        n = int(rows[row])
        got = val[row] * inv_scale[:, n].numpy()[:, None, None, None]                           # undo the power-of-two scale
        want = U[:, :, :, n].reshape(4, 3, cin // 32, 32).transpose(0, 2, 1, 3)
        assert np.allclose(got, want, rtol=3e-7, atol=0.0)                                      # hi + lo carries 22 bits
    # both halves of every stored weight are normal halfs or zero: the largest |U| of a (position, channel) sits in [2^13, 2^14)
    amax = np.abs(val).reshape(cout, 4, -1).max(axis=2)
    assert np.all((amax >= 2.0 ** 13) & (amax < 2.0 ** 14))
    log2 = np.log2(inv_scale.numpy())
    assert np.all(log2 == np.round(log2))                                                       # powers of two: the rescale is exact


