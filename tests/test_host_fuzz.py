"""
The HOST side of libgpp_hip under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers -- on the CPU build only, never
on the GPU): `make asan` compiles every translation unit with the sanitizers on its host pass and links csrc/host_fuzz.cpp, a driver that
feeds descriptors to the device-free entry points (gpp_conv2d_flops, _split_rule, _workspace_bytes, _tile_candidates, the stem weight
packers, the workspace-size functions) and to the argument checks of the launching ones (gpp_conv2d_igemm, gpp_bottleneck_tail,
gpp_bottleneck_block, gpp_plan_run, gpp_poll_f32: without a device they end in an error code before any launch).

Hypothesis draws the seeds; each example is a file of a few hundred descriptors: well-formed layers of the real network, the same with a
handful of fields replaced by boundary values (0, -1, INT_MAX, INT_MIN, misaligned and null pointers, absurd pitches and group offsets),
and pure noise.  The bar: every call returns GPP_OK, a GPP_ERR_* or a hipError_t -- and the sanitizers stay silent.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ground-plane-polling_amd', 'csrc')
DRIVER = os.path.join(ROOT, 'ground-plane-polling_amd', 'lib', 'asan', 'gpp_host_fuzz')

from keras_retinanet_3D.backend import hip  # noqa: E402


@pytest.fixture(scope='module')
def driver():
    if not os.path.isfile('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc: the sanitizer build of the host code cannot be made here')
    rc = subprocess.run(['make', '-C', CSRC, '-j8', 'asan'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert rc.returncode == 0, rc.stdout[-3000:]
    sizes = subprocess.run([DRIVER, '--sizes'], stdout=subprocess.PIPE, universal_newlines=True, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0'))
    rec, desc = (int(v) for v in sizes.stdout.split())
    assert desc == ctypes.sizeof(hip.ConvDesc) and rec == 16 + 3 * desc        # the ctypes mirror and the C struct agree
    return DRIVER


INT_FIELDS = [n for n, t in hip.ConvDesc._fields_ if t is ctypes.c_int32]
PTR_FIELDS = [n for n, t in hip.ConvDesc._fields_ if t is ctypes.c_void_p]
GROUP_FIELDS = [n for n, _ in hip.ConvGroup._fields_]
BOUNDARY = [0, 1, -1, 2, 3, 7, 8, 31, 32, 33, 64, 127, 128, 129, 255, 256, 512, 1024, 4608, 65535, 65536, 2 ** 31 - 1, -2 ** 31, 2 ** 30, 12345, 1000003]
# (kh, kw, cin, cout, stride, pad, H, W, relu, residual): layers of the real graph at small sizes
LAYERS = [(1, 1, 64, 64, 1, 0, 25, 40, 1, 0), (3, 3, 64, 64, 1, 1, 25, 40, 1, 0), (1, 1, 64, 256, 1, 0, 25, 40, 1, 1), (1, 1, 256, 128, 2, 0, 25, 40, 1, 0),
          (3, 3, 512, 512, 1, 1, 13, 21, 1, 0), (3, 3, 2048, 512, 2, 1, 13, 42, 0, 0), (3, 3, 512, 144, 1, 1, 51, 167, 0, 0), (1, 1, 1024, 512, 1, 0, 26, 84, 0, 1)]


def base_desc(rng, dtype):
    kh, kw, cin, cout, stride, pad, H, W, relu, res = LAYERS[int(rng.integers(len(LAYERS)))]
    d = hip.ConvDesc()
    ptr = lambda: int(rng.integers(1, 2 ** 40)) * 16  # noqa: E731
    d.inp, d.weight, d.bias, d.out, d.zero_page = ptr(), ptr(), ptr(), ptr(), ptr()
    d.residual = ptr() if res else None
    d.dtype = dtype
    d.batch, d.C_in, d.C_out, d.KH, d.KW, d.stride, d.pad_top, d.pad_left = int(rng.integers(1, 9)), cin, cout, kh, kw, stride, pad, pad
    d.in_pitch, d.out_pitch, d.res_pitch = cin, cout, cout if res else 0
    d.weight_rows = (cout + 255) // 256 * 256
    d.relu, d.n_groups = relu, 1
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    G = d.groups[0]
    G.in_bstride, G.out_bstride, G.res_bstride = H * W * cin, Ho * Wo * cout, Ho * Wo * cout
    G.H_in, G.W_in, G.H_out, G.W_out, G.H_res, G.W_res = H, W, Ho, Wo, Ho, Wo
    if dtype in (hip.GPP_F16X3, hip.GPP_BF16X3) and rng.integers(2):
        d.x3_split = 1 | 2 | (4 if res else 0)
    if dtype == hip.GPP_F16X3:
        d.out_scale = ptr()
    if rng.integers(3) == 0:
        d.partial, d.partial_bytes = ptr(), int(rng.integers(0, 2 ** 34))
    return d


def mutate(d, rng):
    for _ in range(int(rng.integers(1, 6))):
        which = int(rng.integers(4))
        if which == 0:
            setattr(d, INT_FIELDS[int(rng.integers(len(INT_FIELDS)))], int(np.int32(BOUNDARY[int(rng.integers(len(BOUNDARY)))])))
        elif which == 1:
            v = [0, 1, 8, 12, int(rng.integers(1, 2 ** 47)), int(rng.integers(1, 2 ** 40)) * 16 + int(rng.integers(1, 16))][int(rng.integers(6))]
            setattr(d, PTR_FIELDS[int(rng.integers(len(PTR_FIELDS)))], v or None)
        elif which == 2:
            g = d.groups[int(rng.integers(hip.GPP_MAX_GROUPS))]
            name = GROUP_FIELDS[int(rng.integers(len(GROUP_FIELDS)))]
            big = name in ('in_off', 'in_bstride', 'out_off', 'out_bstride', 'res_off', 'res_bstride')
            v = BOUNDARY[int(rng.integers(len(BOUNDARY)))] * ([1, 2 ** 20, 2 ** 31, -2 ** 31][int(rng.integers(4))] if big else 1)
            setattr(g, name, int(np.int64(v)) if big else int(np.int32(v)))
        else:
            d.partial_bytes = int(np.int64(BOUNDARY[int(rng.integers(len(BOUNDARY)))]) * [1, 2 ** 32, -1][int(rng.integers(3))])
    return d


def records(seed, n=300):
    rng = np.random.default_rng(seed)
    out = bytearray()
    for _ in range(n):
        dtype = [hip.GPP_BF16, hip.GPP_F16, hip.GPP_F32, hip.GPP_BF16X3, hip.GPP_F16X3, 0, 7][int(rng.integers(7))]
        style = int(rng.integers(10))
        descs = []
        for _k in range(3):
            if style == 0:                                          # pure noise
                raw = rng.integers(0, 256, size=ctypes.sizeof(hip.ConvDesc), dtype=np.uint8).tobytes()
                descs.append(raw)
            else:
                d = base_desc(rng, dtype)
                if style >= 4:
                    mutate(d, rng)
                descs.append(bytes(d))
        head = np.array([rng.integers(0, 2 ** 32) for _ in range(4)], dtype=np.uint32).tobytes()
        out += head + b''.join(descs)
    return bytes(out)


def run_driver(driver, blob, tmp_path, name):
    path = os.path.join(str(tmp_path), name)
    with open(path, 'wb') as f:
        f.write(blob)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:halt_on_error=1:abort_on_error=0', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    return subprocess.run([driver, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=300)


def test_well_formed_layers_pass_validation_and_reach_the_missing_device(driver, tmp_path):
    """ the generator's unmutated descriptors are accepted by the device-free entry points (so the fuzzing below starts from the inside of
    the valid region, not from noise that the first check rejects) """
    rng = np.random.default_rng(1)
    blob = bytearray()
    for _ in range(64):
        head = np.array([0, 40, 0, 0], dtype=np.uint32).tobytes()           # kind 0: gpp_conv2d_igemm
        blob += head + b''.join(bytes(base_desc(rng, hip.GPP_F16X3)) for _ in range(3))
    r = run_driver(driver, bytes(blob), tmp_path, 'valid.bin')
    assert r.returncode == 0, r.stderr[-3000:]
    n, ok, err, hiperr = (int(v) for v in __import__('re').findall(r'\d+', r.stdout)[:4])
    assert n == 64 and ok >= 4 * 64, r.stdout        # flops, split rule, workspace, tile candidates: GPP_OK on every record


@settings(max_examples=int(os.environ.get('GPP_FUZZ_EXAMPLES', '12')), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(min_value=0, max_value=2 ** 32 - 1))
def test_no_descriptor_trips_a_sanitizer(driver, tmp_path, seed):
    r = run_driver(driver, records(seed), tmp_path, 'fuzz_{}.bin'.format(seed))
    assert r.returncode == 0, 'seed {}: rc {}\n{}\n{}'.format(seed, r.returncode, r.stdout[-500:], r.stderr[-4000:])
    assert 'records:' in r.stdout
