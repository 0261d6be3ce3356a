""" Error of one conv layer against float64, per arithmetic type, on the unit-test cases of tests/test_conv_gpu.py (random N(0,1)
activations, He weights): rms and max of |got - ref| / rms(ref).  Calibrates the bars of tests/test_conv_f16x3_gpu.py.
    python tools/x3_error_probe.py            (needs the GPU) """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'ground-plane-polling_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402
import test_conv_f32_gpu as T  # noqa: E402
from test_conv_gpu import CASES  # noqa: E402

for case in CASES:
    line = '{:22s} K={:5d}'.format(case[0], case[6] * case[6] * case[4])
    for dtype in ('f32', 'f16x3', 'bf16x3'):
        name, B, H, W, Cin, Cout, K, stride, pad, out_hw, relu, resmode, _ = case
        make, out, ref, kdepth, keep = T._layer(case, dtype=dtype) if dtype != 'f16x3' else (None,) * 5
        if dtype == 'f16x3':
            make0, out, ref, kdepth, keep = T._layer(case, dtype='f32')
            xin, w, rmap, ws, bias = keep
            g = torch.Generator().manual_seed(sum(map(ord, name)))
            _ = torch.randn((B, H, W, Cin), generator=g)
            k = torch.randn((K, K, Cin, Cout), generator=g) * (2.0 / (K * K * Cin)) ** 0.5
            w16 = C.pack_weight(k.numpy(), 'f16x3', xin.buf.device)
            sc = C.out_scale_of(k.numpy(), xin.buf.device)
            d0 = make0(128128)
            d = C.conv_desc([xin], [out], w16, bias, K, K, Cin, Cout, stride=stride, pad=(d0.pad_top, d0.pad_left), relu=relu, residuals=rmap,
                            dtype='f16x3', tile_hint=128128, out_scale=sc)
        else:
            d = make(128128)
        out.buf.fill_(float('nan'))
        C.run_conv(d)
        got = out.buf.double().cpu()
        rms = float(ref.pow(2).mean().sqrt())
        err = (got - ref).abs()
        line += '  {}: rms {:.2e} max {:.2e}'.format(dtype, float(err.pow(2).mean().sqrt()) / rms, float(err.max()) / rms)
    print(line, flush=True)
