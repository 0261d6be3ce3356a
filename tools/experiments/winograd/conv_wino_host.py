"""
Host side of the shelved Winograd F(2, 3) path (was in keras_retinanet_3D/layers/conv.py and backend/hip.py up to commit 32fff05):
weight packing, descriptors and the ctypes mirror of gpp_wino_desc.  Not imported by the package.
"""
import ctypes

import numpy as np

from keras_retinanet_3D.backend import hip
from keras_retinanet_3D.layers.conv import weight_row_order

c_void_p, c_int64 = ctypes.c_void_p, ctypes.c_int64


class WinoGroup(ctypes.Structure):
    _fields_ = [('H', ctypes.c_int32), ('W', ctypes.c_int32), ('map_off', c_int64), ('pair_off', c_int64)]


class WinoDesc(ctypes.Structure):
    _fields_ = [('inp', c_void_p), ('out', c_void_p), ('weight', c_void_p), ('bias', c_void_p), ('out_scale', c_void_p),
                ('in_bstride', c_int64), ('out_bstride', c_int64),
                ('batch', ctypes.c_int32), ('C_in', ctypes.c_int32), ('C_out', ctypes.c_int32), ('in_pitch', ctypes.c_int32),
                ('out_pitch', ctypes.c_int32), ('relu', ctypes.c_int32), ('n_groups', ctypes.c_int32), ('pairs_per_image', ctypes.c_int32),
                ('in_bytes', ctypes.c_int32), ('weight_bytes', ctypes.c_int32),
                ('groups', WinoGroup * hip.GPP_MAX_GROUPS)]


# ---- Winograd F(2, 3) along W (csrc/conv_wino_impl.h): G of the standard interpolation points 0, 1, -1, inf
WINO_G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], np.float64)


def pack_weight_wino(kernel_hwio, device):
    """ Keras HWIO float32 3 x 3 kernel -> (weight tensor for gpp_wino_conv3x3_f16x3, out_scale (4, C_out) float32), both on `device`.
    U_p[kh, c, n] = sum_kw G[p, kw] g[kh, kw, c, n] in float64; per (position, output channel) the power of two that puts the largest
    |U_p[.., n]| in [2^13, 2^14); rounded to float32, split into two IEEE halves.  Layout: [C_out rows in the kernels' 32-row interleave]
    [4 positions][C_in / 32 chunks][3 kernel rows][32 hi | 32 lo]. """
    import torch
    k = np.asarray(kernel_hwio, np.float64)
    KH, KW, Cin, Cout = k.shape
    assert (KH, KW) == (3, 3) and Cin % 32 == 0 and Cout % 128 == 0
    U = np.einsum('pw,hwcn->phcn', WINO_G, k)                                   # (4, kh, c, n)
    amax = np.abs(U).reshape(4, -1, Cout).max(axis=1)
    e = np.where(amax > 0, 13.0 - np.floor(np.log2(np.maximum(amax, 1e-300))), 0.0).clip(-100, 100)
    scale = np.power(2.0, e)                                                     # (4, n)
    Us = (U * scale[:, None, None, :]).astype(np.float32)
    # rows: output channel; K per position: (chunk, kh, 32 channels)
    w = torch.as_tensor(Us).permute(3, 0, 2, 1).reshape(Cout, 4, Cin // 32, 32, 3).permute(0, 1, 2, 4, 3).reshape(Cout, 4 * (Cin // 32) * 3, 32)
    w = w[weight_row_order(Cout)].contiguous()
    hi = w.to(torch.float16)
    lo = (w - hi.to(torch.float32)).to(torch.float16)
    both = torch.stack([hi, lo], dim=2).reshape(Cout, -1)                       # per K-step [32 hi | 32 lo]
    return both.contiguous().view(torch.float32).to(device).contiguous(), torch.as_tensor((1.0 / scale).astype(np.float32)).to(device).contiguous()


def wino_desc(inputs, outputs, batch, C_in, C_out, weight=None, bias=None, out_scale=None, relu=False, v=None, transform=False):
    """ gpp_wino_desc over FMaps: transform=True: inputs (pre-split pixel-major maps, one per level) -> v (the transformed map, a flat
    float32 tensor); else v -> outputs (pre-split pixel-major maps) """
    d = WinoDesc()
    maps = inputs if transform else outputs
    assert all(f.split and f.half == 'f16x3' for f in maps) and 1 <= len(maps) <= hip.GPP_MAX_GROUPS
    assert all(f.buf.data_ptr() == maps[0].buf.data_ptr() and f.pitch == maps[0].pitch and f.bstride == maps[0].bstride and f.B == batch for f in maps)
    pair = 0
    for g, f in enumerate(maps):
        d.groups[g].H, d.groups[g].W, d.groups[g].map_off, d.groups[g].pair_off = f.H, f.W, f.off, pair
        pair += f.H * ((f.W + 1) // 2)
    d.n_groups, d.pairs_per_image, d.batch, d.C_in, d.C_out, d.relu = len(maps), pair, batch, C_in, C_out, int(relu)
    assert v.numel() >= batch * pair * 4 * C_in
    if transform:
        d.inp, d.out = maps[0].buf.data_ptr(), v.data_ptr()
        d.in_pitch, d.in_bstride = maps[0].pitch, maps[0].bstride
    else:
        d.inp, d.out = v.data_ptr(), maps[0].buf.data_ptr()
        d.out_pitch, d.out_bstride = maps[0].pitch, maps[0].bstride
        d.weight, d.bias, d.out_scale = weight.data_ptr(), (bias.data_ptr() if bias is not None else None), out_scale.data_ptr()
    return d


def wino_pairs(maps):
    return sum(f.H * ((f.W + 1) // 2) for f in maps)


