"""
Host side of the HIP convolution kernels (csrc/conv_igemm.hip, C ABI gpp_conv2d_igemm).

Plays the role of keras.layers.Conv2D (+ folded frozen BatchNormalization, Activation('relu'),
Add and layers.UpsampleLike) for the graph of
/root/reference/keras_retinanet_3D/models/retinanet.py:24-205 and the keras_resnet backbone
(models/resnet.py:88-93): it only builds gpp_conv_desc records over torch device buffers
and enqueues the kernel; no arithmetic happens in Python or in torch.
"""

import ctypes

import numpy as np

from ..backend import hip


def torch_dtype(dtype):
    import torch
    return {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32, 'bf16x3': torch.float32, 'f16x3': torch.float32}[dtype]


def gpp_dtype(dtype):
    """ element / arithmetic type of the convolution kernels (include/gpp.h) """
    return {'bf16': hip.GPP_BF16, 'f16': hip.GPP_F16, 'f32': hip.GPP_F32, 'bf16x3': hip.GPP_BF16X3, 'f16x3': hip.GPP_F16X3}[dtype]


def gpp_storage_dtype(dtype):
    """ what the stem / pool / ReLU kernels see: 'bf16x3' / 'f16x3' store float32 and only multiply differently """
    return hip.GPP_F32 if dtype in ('f32', 'bf16x3', 'f16x3') else gpp_dtype(dtype)


X3_TYPES = ('bf16x3', 'f16x3')       # float32 storage, three 16-bit matrix products per float32 product (include/gpp.h)


def x3_half(dtype):
    """ torch type of the two halves a value is split into """
    import torch
    return {'bf16x3': torch.bfloat16, 'f16x3': torch.float16}[dtype]


def elem_size(dtype):
    return 4 if dtype in ('f32', 'bf16x3', 'f16x3') else 2


def k_chunk(dtype):
    """ channels per K-step = 128 bytes of one pixel (include/gpp.h: CK) """
    return 128 // elem_size(dtype)


class FMap(object):
    """ A feature map living inside a torch buffer: pixel (b, y, x) starts at element
    off + b*bstride + (y*W + x)*pitch and has C contiguous channels. """

    def __init__(self, buf, B, H, W, C, off=0, bstride=None, pitch=None, split=False, half='bf16x3'):
        self.buf, self.B, self.H, self.W, self.C = buf, int(B), int(H), int(W), int(C)
        self.off = int(off)
        self.pitch = int(C if pitch is None else pitch)
        self.bstride = int(self.H * self.W * self.pitch if bstride is None else bstride)
        # dtype='bf16x3' only: a PRE-SPLIT map -- every 32 channels of a pixel (128 bytes of the float32-typed buffer) hold
        # [32 bf16 hi | 32 bf16 lo], hi = bf16(x), lo = bf16(x - hi), instead of 32 float32 (gpp_conv_desc.x3_split)
        self.split = bool(split)
        self.half = half               # which of the two split types the halves are ('bf16x3' | 'f16x3'); only read when split
        if self.split:
            assert self.C % 32 == 0 and self.pitch % 32 == 0 and self.off % 32 == 0 and self.bstride % 32 == 0

    def mark_split(self):
        """ declare the map pre-split (same checks as the constructor's) """
        assert self.C % 32 == 0 and self.pitch % 32 == 0 and self.off % 32 == 0 and self.bstride % 32 == 0
        self.split = True
        return self

    @classmethod
    def empty(cls, B, H, W, C, dtype, device, split=False, half='bf16x3'):
        import torch
        return cls(torch.empty((B, H, W, C), dtype=dtype, device=device), B, H, W, C, split=split, half=half)

    def dense(self):
        """ (B, H, W, C) torch view of the map (needs pitch*W*H <= bstride); raw storage -- for a pre-split map these are the
        [hi | lo] bits, not values: use read / write there. """
        import torch
        return torch.as_strided(self.buf.view(-1), (self.B, self.H, self.W, self.C),
                                (self.bstride, self.W * self.pitch, self.pitch, 1), self.off)

    def read(self):
        """ (B, H, W, C) tensor of the VALUES of the map (a copy for pre-split maps: hi + lo in float32) """
        import torch
        if not self.split:
            return self.dense()
        raw = self.dense().contiguous().view(x3_half(self.half)).reshape(self.B, self.H, self.W, self.C // 32, 2, 32).float()
        return (raw[..., 0, :] + raw[..., 1, :]).reshape(self.B, self.H, self.W, self.C)

    def write(self, values):
        """ store a (B, H, W, C) tensor of values into the map (pre-split maps: split here, as the kernels' epilogue does) """
        import torch
        if not self.split:
            self.dense().copy_(values.to(self.buf.dtype))
            return
        v = values.to(device=self.buf.device, dtype=torch.float32).reshape(self.B, self.H, self.W, self.C // 32, 32)
        if self.half == 'f16x3':          # as the epilogue does: finite values clamped to the half range, non-finite ones left as they are
            v = torch.where(torch.isfinite(v), v.clamp(-65504.0, 65504.0), v)
        hi = v.to(x3_half(self.half))
        lo = (v - hi.float()).to(x3_half(self.half))
        both = torch.stack([hi, lo], dim=4).reshape(self.B, self.H, self.W, 2 * self.C)            # [.., chunk, (hi | lo), 32]
        self.dense().copy_(both.contiguous().view(torch.float32).reshape(self.B, self.H, self.W, self.C))


def pack_weight(kernel_hwio, dtype, device):
    """ Keras HWIO float32 kernel (KH, KW, C_in, C_out) -> device tensor
    [C_out rounded up to 256][KH*KW*C_in] in the compute type, K ordered (channel chunk, kh, kw, channel in chunk);
    a chunk is 128 bytes of channels: 64 for the 16-bit types, 32 for float32. """
    import torch
    k = torch.as_tensor(np.ascontiguousarray(kernel_hwio, dtype=np.float32))
    KH, KW, Cin, Cout = k.shape
    rows = ((Cout + 255) // 256) * 256
    ck = k_chunk(dtype)
    w = torch.zeros((rows, KH * KW * Cin), dtype=torch.float32)
    # K order (chunk of CK input channels, kh, kw, CK channels): see include/gpp.h
    w[:Cout] = k.permute(3, 0, 1, 2).reshape(Cout, KH * KW, Cin // ck, ck).permute(0, 2, 1, 3).reshape(Cout, KH * KW * Cin)
    w = w[weight_row_order(rows)]
    if dtype in X3_TYPES:
        # every K-step of 32 channels becomes [32 halves hi | 32 halves lo], hi = h(w), lo = h(w - hi), both round-to-nearest:
        # the same 128 bytes per row and K-step as float32, handed out as a float32-typed tensor of the usual shape
        if dtype == 'f16x3':
            w = w * weight_scale(k, rows)[weight_row_order(rows)][:, None]          # exact: powers of two
        hi = w.to(x3_half(dtype))
        lo = (w - hi.to(torch.float32)).to(x3_half(dtype))
        both = torch.stack([hi.reshape(rows, -1, 32), lo.reshape(rows, -1, 32)], dim=2).reshape(rows, 2 * KH * KW * Cin)
        return both.contiguous().view(torch.float32).to(device).contiguous()
    return w.to(torch_dtype(dtype)).to(device).contiguous()


def weight_scale(kernel_hwio, rows=None):
    """ 'f16x3': the power of two every output channel's weights are multiplied by before they are split into two IEEE halves,
    chosen so that the channel's largest weight lands in [2^13, 2^14) -- far from the half overflow (65504) and high enough that
    the lo half of any weight down to 2^-13 of the largest is still a NORMAL half (a subnormal lo would carry fewer than 11 bits).
    (C_out,) float32, or (rows,) padded with ones.  Its inverse is gpp_conv_desc.out_scale (out_scale_of). """
    import torch
    k = torch.as_tensor(np.ascontiguousarray(kernel_hwio, dtype=np.float32))
    Cout = k.shape[3]
    amax = k.abs().reshape(-1, Cout).max(dim=0).values.double()
    e = torch.where(amax > 0, 13.0 - torch.floor(torch.log2(amax.clamp_min(1e-300))), torch.zeros_like(amax)).clamp(-100, 100)
    scale = torch.pow(torch.tensor(2.0, dtype=torch.float64), e).float()
    if rows is None:
        return scale
    out = torch.ones((rows,), dtype=torch.float32)
    out[:Cout] = scale
    return out


def out_scale_of(kernel_hwio, device):
    """ gpp_conv_desc.out_scale of a layer packed with pack_weight(..., 'f16x3'): 1 / weight_scale, exact """
    return (1.0 / weight_scale(kernel_hwio)).to(device).contiguous()


def weight_row_order(rows):
    """ Row interleave the kernel expects (include/gpp.h): within every group of 32 output channels,
    stored row 16*h + 4*q + r holds output channel 8*q + 4*h + r (h in 0..1, q in 0..3, r in 0..3), so
    that the two MFMA tiles of a pair hand each lane 8 consecutive output channels. """
    import torch
    pos = torch.arange(rows)
    g, within = pos // 32, pos % 32
    h, q, r = within // 16, (within % 16) // 4, within % 4
    return g * 32 + 8 * q + 4 * h + r


_ZERO_PAGES = {}


def zero_page(device):
    import torch
    key = str(device)
    if key not in _ZERO_PAGES:
        _ZERO_PAGES[key] = torch.zeros((256,), dtype=torch.uint8, device=device)
    return _ZERO_PAGES[key]


def same_pad(in_size, k, stride):
    """ TF 'same' padding: (out, pad_before). """
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return out, total // 2


def conv_desc(inputs, outputs, weight, bias, KH, KW, C_in, C_out, stride=1, pad=(0, 0), relu=False,
              residuals=None, dtype='bf16', out_f32=False, tile_hint=0, diag=0,
              workspace=None, split_k=0, out_scale=None):
    """ Build a gpp_conv_desc.  inputs / outputs / residuals are lists of FMap (one per group,
    all groups share weights; every list member must live in the same torch buffer). """
    d = hip.ConvDesc()
    d.inp = inputs[0].buf.data_ptr()
    d.weight = weight.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.out = outputs[0].buf.data_ptr()
    d.zero_page = zero_page(inputs[0].buf.device).data_ptr()
    d.residual = residuals[0].buf.data_ptr() if residuals else None
    d.dtype = gpp_dtype(dtype)
    d.out_f32 = int(out_f32)
    d.batch = inputs[0].B
    d.C_in, d.C_out, d.KH, d.KW, d.stride = C_in, C_out, KH, KW, stride
    d.pad_top, d.pad_left = pad
    d.in_pitch, d.out_pitch = inputs[0].pitch, outputs[0].pitch
    d.res_pitch = residuals[0].pitch if residuals else 0
    d.weight_rows = int(weight.shape[0])
    d.relu = int(relu)
    d.n_groups = len(inputs)
    d.tile_hint = int(tile_hint)
    d.reserved = int(diag)          # diagnostic ablation bits (-DGPP_STAMPS build only; the production library rejects non-zero)
    if workspace is not None:       # split-K partial tiles (float32); the library decides whether to split
        d.partial = workspace.data_ptr()
        d.partial_bytes = workspace.numel() * workspace.element_size()
    d.split_k = int(split_k)
    d.x3_split = (1 if inputs[0].split else 0) | (2 if outputs[0].split else 0) | (4 if (residuals and residuals[0].split) else 0)
    assert d.x3_split == 0 or dtype in X3_TYPES
    assert all(f.half == dtype for f in list(inputs) + list(outputs) + list(residuals or []) if f.split)
    assert (out_scale is not None) == (dtype == 'f16x3'), 'f16x3 layers carry the inverse of their weight scale (out_scale_of)'
    d.out_scale = out_scale.data_ptr() if out_scale is not None else None
    assert all(f.split == inputs[0].split for f in inputs) and all(f.split == outputs[0].split for f in outputs)
    assert 1 <= len(inputs) <= hip.GPP_MAX_GROUPS and len(outputs) == len(inputs)
    assert int(weight.shape[1]) == KH * KW * C_in
    for g, (fi, fo) in enumerate(zip(inputs, outputs)):
        assert fi.buf.data_ptr() == inputs[0].buf.data_ptr() and fo.buf.data_ptr() == outputs[0].buf.data_ptr()
        assert fi.pitch == d.in_pitch and fo.pitch == d.out_pitch and fi.B == d.batch and fo.B == d.batch
        G = d.groups[g]
        G.in_off, G.in_bstride = fi.off, fi.bstride
        G.out_off, G.out_bstride = fo.off, fo.bstride
        G.H_in, G.W_in, G.H_out, G.W_out = fi.H, fi.W, fo.H, fo.W
        if residuals:
            fr = residuals[g]
            assert fr.buf.data_ptr() == residuals[0].buf.data_ptr() and fr.pitch == d.res_pitch
            G.res_off, G.res_bstride, G.H_res, G.W_res = fr.off, fr.bstride, fr.H, fr.W
        else:
            G.H_res, G.W_res = fo.H, fo.W
    return d


def default_split(KH, KW, C_in, C_out, pixels_per_image):
    """ the library's own split-K rule (csrc/conv_igemm.hip split_rule), restated: deep-K layers whose per-image grid is tiny at any batch """
    tiles = -(-pixels_per_image // 128) * -(-C_out // 128)
    kdepth = KH * KW * C_in
    if tiles > 24 or kdepth < 3072:
        return 1
    return max(1, min(8, (kdepth + 768) // 1536))


def latency_split_config():
    import os
    return ';lat={},{},{},{}'.format(os.environ.get('GPP_LAT_MIN_K', '2048'), os.environ.get('GPP_LAT_TARGET', '512'),
                                     os.environ.get('GPP_LAT_MAX', '8'), os.environ.get('GPP_LAT_STEP_K', '512'))


def latency_split(KH, KW, C_in, C_out, pixels_per_image):
    """ plan='latency' (models.load_model): the split-K factor of a layer as a function of the LAYER ALONE -- kernel size, channels, output
    pixels per image -- for callers that run one image per call (the reference's own timer, bin/run_network.py:108-111).  At batch 1 a
    layer of res4 / res5 / the small pyramid levels fields 36 - 140 workgroups for 256 CUs, each pulling its whole K depth of both
    operands through one CU; splitting K over `split` workgroups per tile fills the chip and shortens every workgroup's chain of
    dependent tile loads (float32 partial slabs + the deterministic reduce pass of gpp_conv2d_igemm).  Never below the library's own rule,
    never a function of the batch, the tile or a timing: results are byte-identical at every batch size within the mode. """
    import os
    min_k, target = int(os.environ.get('GPP_LAT_MIN_K', '2048')), int(os.environ.get('GPP_LAT_TARGET', '512'))
    max_split, step_k = int(os.environ.get('GPP_LAT_MAX', '8')), int(os.environ.get('GPP_LAT_STEP_K', '512'))
    base = default_split(KH, KW, C_in, C_out, pixels_per_image)
    kdepth = KH * KW * C_in
    wgs = -(-pixels_per_image // 64) * -(-C_out // 128)          # workgroups of the small tiles a batch-1 launch of this layer runs
    if kdepth < min_k or wgs >= 256:
        return base
    return max(base, min(max_split, -(-target // wgs), max(1, kdepth // step_k)))


def run_conv(desc):
    hip.check(hip.lib().gpp_conv2d_igemm(ctypes.byref(desc), hip.stream_ptr()), 'gpp_conv2d_igemm')


def split_rule(desc):
    """ split-K factor the library will use for this layer (a function of the layer alone, include/gpp.h) """
    k = ctypes.c_int(0)
    hip.check(hip.lib().gpp_conv2d_split_rule(ctypes.byref(desc), ctypes.byref(k)), 'gpp_conv2d_split_rule')
    return k.value


def workspace_bytes(desc):
    n = hip.c_size_t(0)
    hip.check(hip.lib().gpp_conv2d_workspace_bytes(ctypes.byref(desc), ctypes.byref(n)), 'gpp_conv2d_workspace_bytes')
    return int(n.value)


def conv_flops(desc):
    f = ctypes.c_double(0.0)
    hip.check(hip.lib().gpp_conv2d_flops(ctypes.byref(desc), ctypes.byref(f)), 'gpp_conv2d_flops')
    return f.value
