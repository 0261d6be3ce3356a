"""
ctypes binding of libgpp_hip.so (C ABI declared in include/gpp.h).

This module takes the place of the reference's operator-alias layer
(/root/reference/keras_retinanet_3D/backend/tensorflow_backend.py:20-156): the layers of this
package call hand-written HIP kernels through it instead of `tensorflow.*`.

There is no CPU fallback.  If the shared library is missing, or no HIP device is present, the
functions below raise; nothing silently routes around the kernels.
"""

import ctypes
import os

_LIB = None
_HERE = os.path.dirname(os.path.abspath(__file__))
# GPP_LIB: alternative build of the library (A/B timing of kernel changes); default = the in-tree build
LIB_PATH = os.environ.get('GPP_LIB') or os.path.normpath(os.path.join(_HERE, '..', '..', 'lib', 'libgpp_hip.so'))
CSRC_DIR = os.path.normpath(os.path.join(_HERE, '..', '..', 'csrc'))

GPP_OK = 0
_ERRORS = {
    -1: 'GPP_ERR_BAD_ARG',
    -2: 'GPP_ERR_WORKSPACE',
    -3: 'GPP_ERR_ALIGN',
    -4: 'GPP_ERR_UNSUPPORTED',
}

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t
c_int64 = ctypes.c_int64


GPP_BF16 = 1
GPP_F16 = 2
GPP_F32 = 3
GPP_BF16X3 = 4
GPP_F16X3 = 5
GPP_MAX_GROUPS = 5


class GppError(RuntimeError):
    pass


class ConvGroup(ctypes.Structure):
    """ gpp_conv_group (include/gpp.h) """
    _fields_ = [('in_off', c_int64), ('in_bstride', c_int64), ('out_off', c_int64), ('out_bstride', c_int64),
                ('res_off', c_int64), ('res_bstride', c_int64),
                ('H_in', ctypes.c_int32), ('W_in', ctypes.c_int32), ('H_out', ctypes.c_int32), ('W_out', ctypes.c_int32),
                ('H_res', ctypes.c_int32), ('W_res', ctypes.c_int32), ('tile_start', ctypes.c_int32),
                ('row_begin', ctypes.c_int32)]


class ConvDesc(ctypes.Structure):
    """ gpp_conv_desc (include/gpp.h) """
    _fields_ = [('inp', c_void_p), ('weight', c_void_p), ('bias', c_void_p), ('residual', c_void_p),
                ('out', c_void_p), ('zero_page', c_void_p),
                ('dtype', ctypes.c_int32), ('out_f32', ctypes.c_int32),
                ('batch', ctypes.c_int32), ('C_in', ctypes.c_int32), ('C_out', ctypes.c_int32),
                ('KH', ctypes.c_int32), ('KW', ctypes.c_int32), ('stride', ctypes.c_int32),
                ('pad_top', ctypes.c_int32), ('pad_left', ctypes.c_int32),
                ('in_pitch', ctypes.c_int32), ('out_pitch', ctypes.c_int32), ('res_pitch', ctypes.c_int32),
                ('weight_rows', ctypes.c_int32), ('relu', ctypes.c_int32), ('n_groups', ctypes.c_int32),
                ('tile_hint', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('in_bytes', ctypes.c_int32), ('weight_bytes', ctypes.c_int32),
                ('partial', c_void_p), ('partial_bytes', c_int64), ('split_k', ctypes.c_int32), ('partial_rows', ctypes.c_int32),
                ('groups', ConvGroup * GPP_MAX_GROUPS),
                ('x3_split', ctypes.c_int32), ('reserved2', ctypes.c_int32), ('out_scale', c_void_p), ('range_counter', c_void_p)]


def _declare(lib):
    lib.gpp_version.restype = ctypes.c_char_p
    lib.gpp_version.argtypes = []
    lib.gpp_poll_workspace_bytes.restype = c_int
    lib.gpp_poll_workspace_bytes.argtypes = [c_int, c_int, c_int, ctypes.POINTER(c_size_t)]
    lib.gpp_poll_f32.restype = c_int
    lib.gpp_poll_f32.argtypes = [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_float] + [c_void_p] * 4 + \
        [c_void_p, c_size_t, c_void_p]
    lib.gpp_conv2d_igemm.restype = c_int
    lib.gpp_conv2d_igemm.argtypes = [ctypes.POINTER(ConvDesc), c_void_p]
    lib.gpp_stem_conv7x7_bn_relu.restype = c_int
    lib.gpp_stem_conv7x7_bn_relu.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.gpp_stem_conv7x7_bn_relu_mfma.restype = c_int
    lib.gpp_stem_conv7x7_bn_relu_mfma.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.gpp_stem_pack_weights_f16.restype = c_int
    lib.gpp_stem_pack_weights_f16.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.gpp_stem_pack_weights_f16x3.restype = c_int
    lib.gpp_stem_pack_weights_f16x3.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.gpp_stem_conv7x7_bn_relu_x3.restype = c_int
    lib.gpp_stem_conv7x7_bn_relu_x3.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.gpp_stem_pool_fused_mfma.restype = c_int
    lib.gpp_stem_pool_fused_mfma.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.gpp_maxpool3x3s2_same.restype = c_int
    lib.gpp_maxpool3x3s2_same.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.gpp_relu.restype = c_int
    lib.gpp_relu.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_void_p]
    lib.gpp_preprocess_u8_bgr.restype = c_int
    lib.gpp_preprocess_u8_bgr.argtypes = [c_void_p] * 8 + [c_int] * 5 + [c_float] * 3 + [c_void_p]
    lib.gpp_detect_workspace_bytes.restype = c_int
    lib.gpp_detect_workspace_bytes.argtypes = [c_int, c_int64, ctypes.POINTER(c_size_t)]
    lib.gpp_detect_f32.restype = c_int
    lib.gpp_detect_f32.argtypes = [c_void_p] * 4 + [c_int, c_int64, c_int, c_int, c_float, c_float, c_int] + \
        [c_void_p] * 7 + [c_void_p, c_size_t, c_void_p]
    lib.gpp_detect_osf_workspace_bytes.restype = c_int
    lib.gpp_detect_osf_workspace_bytes.argtypes = lib.gpp_detect_workspace_bytes.argtypes
    lib.gpp_detect_osf_f32.restype = c_int
    lib.gpp_detect_osf_f32.argtypes = lib.gpp_detect_f32.argtypes
    lib.gpp_pack_detections.restype = c_int
    lib.gpp_pack_detections.argtypes = [c_void_p] * 8 + [c_int, c_int, c_void_p, c_void_p]
    lib.gpp_detect_stages_f32.restype = c_int
    lib.gpp_detect_stages_f32.argtypes = [c_int] + lib.gpp_detect_f32.argtypes
    lib.gpp_conv2d_flops.restype = c_int
    lib.gpp_conv2d_flops.argtypes = [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_double)]
    lib.gpp_conv2d_split_rule.restype = c_int
    lib.gpp_conv2d_split_rule.argtypes = [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int)]
    lib.gpp_conv2d_workspace_bytes.restype = c_int
    lib.gpp_conv2d_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_size_t)]
    lib.gpp_plan_run.restype = c_int
    lib.gpp_plan_run.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int]
    lib.gpp_event_create.restype = c_int
    lib.gpp_event_create.argtypes = [ctypes.POINTER(c_void_p)]
    lib.gpp_event_destroy.restype = c_int
    lib.gpp_event_destroy.argtypes = [c_void_p]
    lib.gpp_event_elapsed_ms.restype = c_int
    lib.gpp_event_elapsed_ms.argtypes = [c_void_p, c_void_p, ctypes.POINTER(c_float)]
    lib.gpp_conv2d_tile_candidates.restype = c_int
    lib.gpp_conv2d_tile_candidates.argtypes = [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int), c_int, ctypes.POINTER(c_int)]
    lib.gpp_x3_range_events.restype = c_int
    lib.gpp_x3_range_events.argtypes = [ctypes.POINTER(ctypes.c_uint64), c_int]
    lib.gpp_x3_range_snapshot.restype = c_int
    lib.gpp_x3_range_snapshot.argtypes = [c_void_p, c_void_p]
    lib.gpp_x3_range_snapshot_of.restype = c_int
    lib.gpp_x3_range_snapshot_of.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.gpp_stem_conv7x7_bn_relu_x3_rc.restype = c_int
    lib.gpp_stem_conv7x7_bn_relu_x3_rc.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.gpp_stem_pool_fused_x3.restype = c_int
    lib.gpp_stem_pool_fused_x3.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.gpp_bottleneck_block.restype = c_int
    lib.gpp_bottleneck_block.argtypes = [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), c_int, c_void_p]
    lib.gpp_conv2d_autotune.restype = c_int
    lib.gpp_conv2d_autotune.argtypes = [ctypes.POINTER(ConvDesc), c_int, c_void_p, ctypes.POINTER(c_float)]


def build(verbose=False):
    """ Compile libgpp_hip.so in-tree with hipcc for gfx950 (works without a GPU). """
    import subprocess
    cmd = ['make', '-C', CSRC_DIR, '-j4']
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise GppError('building libgpp_hip.so failed (see output above)')


def lib():
    """ The loaded library.  Raises GppError when it has not been built. """
    global _LIB
    if _LIB is None:
        if not os.path.isfile(LIB_PATH):
            raise GppError('{} not found: build it with `make -C {}` (or __graft_entry__.build()); '
                           'there is no CPU fallback for the HIP path'.format(LIB_PATH, CSRC_DIR))
        # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the HIP runtime already in the
        # process when libgpp_hip.so (linked against the same soname) is loaded, otherwise two
        # runtimes coexist and the kernels see no device (hipErrorNoDevice).
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        _declare(handle)
        _LIB = handle
    return _LIB


def check(rc, what=''):
    if rc == GPP_OK:
        return
    if rc < 0:
        raise GppError('{} failed: {}'.format(what or 'gpp call', _ERRORS.get(rc, rc)))
    raise GppError('{} failed: hipError_t {}'.format(what or 'gpp call', rc))


def require_device():
    """ torch.device('cuda', current) or raise: the product path needs an MI355X. """
    import torch
    if not torch.cuda.is_available():
        raise GppError('no HIP device visible: the ground-plane-polling hot path runs only on the GPU '
                       '(there is no CPU fallback; the CPU oracle under oracle/ is test infrastructure)')
    return torch.device('cuda', torch.cuda.current_device())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """ Raw device pointer of a contiguous torch tensor (or None). """
    if t is None:
        return None
    assert t.is_contiguous(), 'gpp kernels need dense tensors'
    return ctypes.c_void_p(t.data_ptr())


def pack_stem_weights_x3(kernel_147x64, device):
    """ [147][64] float32 folded stem kernel -> device blob of the x3 MFMA stem: [whi 64 x 232 f16][wlo 64 x 232 f16][64 float32 out_scale] """
    import numpy as np
    import torch
    src = np.ascontiguousarray(kernel_147x64, dtype=np.float32)
    dst = np.zeros((2 * 64 * 232 * 2 + 64 * 4,), dtype=np.uint8)
    check(lib().gpp_stem_pack_weights_f16x3(src.ctypes.data_as(c_void_p), dst.ctypes.data_as(c_void_p), dst.nbytes), 'gpp_stem_pack_weights_f16x3')
    return torch.as_tensor(dst).to(device).contiguous()


def pack_stem_weights(kernel_147x64, device):
    """ [147][64] float32 folded stem kernel -> device tensor holding the [64][232] f16 image of the MFMA stem """
    import numpy as np
    import torch
    src = np.ascontiguousarray(kernel_147x64, dtype=np.float32)
    dst = np.zeros((64, 232), dtype=np.float16)
    check(lib().gpp_stem_pack_weights_f16(src.ctypes.data_as(c_void_p), dst.ctypes.data_as(c_void_p), dst.nbytes), 'gpp_stem_pack_weights_f16')
    return torch.as_tensor(dst).to(device).contiguous()
