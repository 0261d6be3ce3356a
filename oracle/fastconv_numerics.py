#!/usr/bin/env python
"""
ORACLE-side numerics study (test infrastructure, not product code; nothing under ground-plane-polling_amd/ imports it):

    would a FAST-CONVOLUTION (Winograd) form of the 3x3 stride-1 layers, in the headline arithmetic of the HIP path -- float32-sized
    maps, every float32 product as three IEEE-half matrix products (hi.wlo + hi.whi + lo.whi), float32 accumulation -- keep the
    detections of the RetinaNet-3D graph inside utils/ledger.REFERENCE_BARS against the float64 oracle fixtures
    (tests/golden/fullsize_*_f64.npz), with the bars UNCHANGED?

The graph is oracle/net_torch.py's (reference: models/retinanet.py:24-205,257-281 + keras_resnet, models/resnet.py:88-93); only the
ARITHMETIC of every convolution is replaced by an emulation of what a kernel would execute, layer by layer selectable:

    x3     direct convolution as the shipped f16x3 kernels compute it: BatchNormalization folded in float64 and rounded once, weights
           x 2^k(n) per output channel (largest weight of the channel in [2^13, 2^14)), both operands split into two IEEE halves
           (hi = h(x), lo = h(x - hi)), the three half products summed into ONE float32 accumulation (one GEMM over the concatenated
           K = [hi | hi | lo] x [wlo ; whi ; whi]), x 2^-k(n), + bias, (+ shortcut), ReLU; the stored map keeps hi + lo (22 bits)
    w2     Winograd F(2, 3) along W (4 products per 2 outputs instead of 6; the 3 kernel rows stay a direct K loop):
           V_p = sum_b BT[p, b] d[.., b] in float32 from the stored map, split into halves; U_p = sum_kw G[p, kw] g[kh, kw] in float64,
           x 2^k(p, n), split into halves; M_p = the same three-product GEMM per position; Y = AT M in float32; scale, bias, ReLU
    w4     Winograd F(2x2, 3x3) (16 products per 4 outputs instead of 36): V = BT d B, U = G g GT, Y = AT M A, same recipe
    (the float32 summation order inside a GEMM is the BLAS's, not the kernel's K order: statistically the same error, other bits)

Every run is reported as the parity ledger (utils/ledger.py) of its detections -- decode / NMS / top-k by oracle/decode_np.py, polling by
oracle/polling.c, both bit-exact stages -- against the float64 fixture of the same frames, next to the control (every layer 'x3').

    python oracle/fastconv_numerics.py [--config resnet50_1k] [--frames 64] [--device cuda|cpu] [--only name,name] [--json out.json]

--device cuda uses torch on the GPU as the calculator of the emulation (float32 GEMMs; no kernel of the product is involved).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
sys.path.insert(0, ROOT)

from oracle import net_torch  # noqa: E402

# Winograd F(2, 3) (Lavin & Gray 2016, the standard interpolation points 0, 1, -1, inf)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def split_half(t):
    """ hi = h(x), lo = h(x - hi): what the f16x3 epilogue stores and what the loop multiplies (csrc/conv_igemm_impl.h X3Half) """
    t = torch.where(torch.isfinite(t), t.clamp(-65504.0, 65504.0), t)
    hi = t.half().float()
    return hi, (t - hi).half().float()


def stored(t):
    hi, lo = split_half(t)
    return hi + lo


def pow2_scale(amax):
    """ layers/conv.py weight_scale: the power of two that puts the largest weight in [2^13, 2^14) """
    e = torch.where(amax > 0, 13.0 - torch.floor(torch.log2(amax.clamp_min(1e-300))), torch.zeros_like(amax)).clamp(-100, 100)
    return torch.pow(torch.tensor(2.0, dtype=torch.float64, device=amax.device), e)


ACCUMULATION = 'mfma'      # 'mfma' | 'blas' (main() sets it)


def mfma_accumulate(ah, al, wh, wl, rows=4096):
    """ what a chain of v_mfma_f32_16x16x32_f16 does to one accumulator: per K-step of 32 channels three instructions (hi.wlo, hi.whi,
    lo.whi, the kernels' order), each adding the sum of its 32 products -- products of halves are exact in float32, and their sum is formed
    before the instruction rounds (profiles/r3/x3_error_probe.txt: the reason f16x3 is closer to float64 than a float32 MFMA chain, which
    rounds after every 4 products) -- to the float32 accumulator with ONE rounding.  Emulated literally: the 32-product sums in float64,
    the running accumulator rounded to float32 after every instruction.  (.., M, K) x (.., K, N); K a multiple of 32 (zero-padded). """
    lead = ah.shape[:-2]
    M, K = ah.shape[-2:]
    N = wh.shape[-1]
    pad = (-K) % 32
    if pad:
        ah, al = F.pad(ah, (0, pad)), F.pad(al, (0, pad))
        wh, wl = F.pad(wh, (0, 0, 0, pad)), F.pad(wl, (0, 0, 0, pad))
        K += pad
    nb = K // 32
    wh64 = wh.double().reshape(lead + (nb, 32, N))
    wl64 = wl.double().reshape(lead + (nb, 32, N))
    out = torch.empty(lead + (M, N), dtype=torch.float32, device=ah.device)
    step = max(256, min(rows, (1 << 27) // max(1, nb * N * max(1, int(np.prod(lead))))))       # (.., nb, rows, N) float64 per term: <= 1 GB
    for m0 in range(0, M, step):
        a_h = ah[..., m0:m0 + step, :].double().reshape(lead + (-1, nb, 32)).transpose(-3, -2)      # (.., nb, m, 32)
        a_l = al[..., m0:m0 + step, :].double().reshape(lead + (-1, nb, 32)).transpose(-3, -2)
        terms = (torch.matmul(a_h, wl64), torch.matmul(a_h, wh64), torch.matmul(a_l, wh64))         # each (.., nb, m, N): exact 32-product sums
        acc = torch.zeros(lead + (a_h.shape[-2], N), dtype=torch.float32, device=ah.device)
        for b in range(nb):
            for t in terms:
                acc = (acc.double() + t[..., b, :, :]).float()
        out[..., m0:m0 + step, :] = acc
    return out


def x3_gemm(A, Wm):
    """ A (.., M, K) float32 x Wm (.., K, N) float64 (unscaled) -> float32 (.., M, N): three half products per float32 product,
    accumulated as the matrix pipe does (ACCUMULATION 'mfma') or by one float32 BLAS GEMM over the concatenated K ('blas': another
    summation order, rounding after every product pair -- noisier than the hardware) """
    s = pow2_scale(Wm.abs().amax(dim=-2))                       # per output column (and per leading index: the Winograd position)
    ws = (Wm * s.unsqueeze(-2)).float()
    ah, al = split_half(A)
    wh, wl = split_half(ws)
    if ACCUMULATION == 'mfma':
        acc = mfma_accumulate(ah, al, wh, wl)
    else:
        acc = torch.matmul(torch.cat([ah, ah, al], dim=-1), torch.cat([wl, wh, wh], dim=-2))
    return acc * (1.0 / s).float().unsqueeze(-2)                # exact (power of two)


def conv_direct(x, k64, stride, pads):
    """ x (1, C, H, W) float32, k64 HWIO float64, pads (top, bottom, left, right) """
    kh, kw, cin, cout = k64.shape
    xp = F.pad(x, (pads[2], pads[3], pads[0], pads[1]))
    ho, wo = (xp.shape[2] - kh) // stride + 1, (xp.shape[3] - kw) // stride + 1
    if kh == 1 and kw == 1:
        a = xp[:, :, ::stride, ::stride].reshape(cin, -1).t()
    else:
        a = F.unfold(xp, (kh, kw), stride=stride)[0].t()        # (L, C * kh * kw), column order (c, kh, kw)
    y = x3_gemm(a, k64.permute(2, 0, 1, 3).reshape(cin * kh * kw, cout))
    return y.t().reshape(1, cout, ho, wo)


def conv_w2(x, k64):
    """ 3x3 stride 1 pad 1, F(2, 3) along W """
    _, cin, H, W = x.shape
    cout = k64.shape[3]
    te = (W + 1) // 2
    xp = F.pad(x, (1, 2 * te + 2 - W - 1, 1, 1))
    d = F.unfold(xp, (3, 4), stride=(1, 2))[0].t().reshape(H * te, cin * 3, 4)            # (L, (c, kh), b)
    V = torch.matmul(d, BT.t().to(d.device)).permute(2, 0, 1).contiguous()                 # (p, L, (c, kh)) float32: one rounding per element
    U = torch.einsum('pw,hwcn->pchn', G.to(k64.device), k64).reshape(4, cin * 3, cout)     # float64
    M = x3_gemm(V, U)                                                                       # (4, L, N)
    y0 = (M[0] + M[1]) + M[2]
    y1 = (M[1] - M[2]) - M[3]
    y = torch.stack([y0, y1], dim=1).reshape(H, te, 2, cout).reshape(H, 2 * te, cout)[:, :W]
    return y.permute(2, 0, 1).unsqueeze(0)


def conv_w4(x, k64):
    """ 3x3 stride 1 pad 1, F(2x2, 3x3) """
    _, cin, H, W = x.shape
    cout = k64.shape[3]
    th, tw = (H + 1) // 2, (W + 1) // 2
    xp = F.pad(x, (1, 2 * tw + 2 - W - 1, 1, 2 * th + 2 - H - 1))
    d = F.unfold(xp, (4, 4), stride=2)[0].t().reshape(th * tw, cin, 4, 4)                  # (L, c, a, b)
    bt = BT.to(d.device)
    V = torch.matmul(bt, torch.matmul(d, bt.t()))                                           # BT d B, float32, two stages
    V = V.permute(2, 3, 0, 1).reshape(16, th * tw, cin).contiguous()
    g = G.to(k64.device)
    U = torch.einsum('ph,hwcn,qw->pqcn', g, k64, g).reshape(16, cin, cout)
    M = x3_gemm(V, U).reshape(4, 4, th * tw, cout)
    at = AT.to(d.device)
    Y = torch.einsum('ip,pqln->iqln', at, M)                                                # AT M   (float32)
    Y = torch.einsum('iqln,jq->ijln', Y, at)                                                # .. A
    y = Y.reshape(2, 2, th, tw, cout).permute(2, 0, 3, 1, 4).reshape(2 * th, 2 * tw, cout)[:H, :W]
    return y.permute(2, 0, 1).unsqueeze(0)


class EmuNet(net_torch.Net):
    """ the oracle's graph with every convolution's arithmetic emulated; `modes`: layer name -> 'x3' | 'w2' | 'w4' (default 'x3') """

    def __init__(self, weights, backbone, modes=None, device='cpu'):
        net_torch.Net.__init__(self, weights, backbone)
        self.modes = modes or {}
        self.device = torch.device(device)
        self.used = {}

    def _run(self, x, name, k64, bias, stride, pads, relu, add, store):
        mode = self.modes.get(name, 'x3')
        if mode != 'x3' and not (k64.shape[0] == 3 and k64.shape[1] == 3 and stride == 1 and pads == (1, 1, 1, 1)):
            mode = 'x3'
        self.used[name] = mode
        k64 = k64.to(self.device)
        y = conv_direct(x, k64, stride, pads) if mode == 'x3' else (conv_w2(x, k64) if mode == 'w2' else conv_w4(x, k64))
        y = y + bias.to(self.device)[None, :, None, None]
        if add is not None:
            y = y + add
        if relu:
            y = torch.relu(y)
        return stored(y) if store else y

    def conv_bn(self, x, conv, bn, stride=1, pad=None, relu=True, add=None, quant_weights=True, f16_operands=False):
        w = self.w
        s = torch.as_tensor(w[bn + '/gamma']).double() / torch.sqrt(torch.as_tensor(w[bn + '/moving_variance']).double() + net_torch.BN_EPS)
        kf = (torch.as_tensor(w[conv + '/kernel']).double() * s[None, None, None, :]).float().double()     # folded in float64, rounded once
        bf = (torch.as_tensor(w[bn + '/beta']).double() - torch.as_tensor(w[bn + '/moving_mean']).double() * s).float()
        p = pad or 0
        return self._run(x, conv, kf, bf, stride, (p, p, p, p), relu, add, conv != 'conv1')       # (the stem's map and pool1 stay float32)

    def conv_bias(self, x, name, stride=1, relu=False, add=None, store=True):
        k = torch.as_tensor(self.w[name + '/kernel']).double()
        ph = net_torch._same_pad(x.shape[2], k.shape[0], stride)
        pw = net_torch._same_pad(x.shape[3], k.shape[1], stride)
        return self._run(x, name, k, torch.as_tensor(self.w[name + '/bias']), stride, (ph[0], ph[1], pw[0], pw[1]), relu, add, store)

    def forward(self, images_nhwc):
        with torch.no_grad():
            x = torch.as_tensor(np.ascontiguousarray(images_nhwc, dtype=np.float32)).permute(0, 3, 1, 2).to(self.device)
            _, C3, C4, C5 = self.resnet(x)
            reg, dim, cls = self.heads(self.fpn(C3, C4, C5))
        return {'regression': reg.float().cpu().numpy(), 'regression_dim': dim.float().cpu().numpy(),
                'classification_logits': cls.float().cpu().numpy()}


# ---- which layers a fast form would cover (the 3x3 stride-1 layers with C_in >= 128 are 87 % of the FLOPs)
def layer_sets(backbone):
    from keras_retinanet_3D.models import weights as Wt
    reg = ['pyramid_regression_{}'.format(i) for i in range(1, 4)]
    towers0 = ['pyramid_regression_0', 'pyramid_classification_0', 'pyramid_regression_dim_0']
    small_towers = ['pyramid_classification_{}'.format(i) for i in range(1, 4)] + ['pyramid_regression_dim_{}'.format(i) for i in range(1, 4)]
    outs = ['pyramid_regression_op{}'.format(i) for i in range(1, 6)] + ['pyramid_classification', 'pyramid_regression_dim']
    fpn = ['P3', 'P4', 'P5']
    res = [c for c, _, kh, _, cin, _, _ in Wt.backbone_layers(backbone) if kh == 3 and cin >= 128]
    return {'regression tower 1-3 (3 x 45.3/4 % of the FLOPs)': reg,
            'regression tower + fused tower inputs': reg + towers0,
            'every tower layer': reg + towers0 + small_towers,
            'towers + head outputs': reg + towers0 + small_towers + outs,
            'towers + outputs + P3-P5': reg + towers0 + small_towers + outs + fpn,
            'every 3x3 s1 layer with C_in >= 128': reg + towers0 + small_towers + outs + fpn + res}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='resnet50_1k')
    ap.add_argument('--frames', type=int, default=None)
    ap.add_argument('--device', default='cuda' if torch.cuda.is_available() else 'cpu')
    ap.add_argument('--only', default=None, help='substrings of run names, separated by ;')
    ap.add_argument('--json', default=None)
    ap.add_argument('--accumulation', default='mfma', choices=['mfma', 'blas'])
    ap.add_argument('--height', type=int, default=402, help='(smaller frames: a smoke run of the script, no fixture to compare with)')
    ap.add_argument('--width', type=int, default=1333)
    args = ap.parse_args()
    from oracle import decode_np
    from oracle.gen_fullsize_goldens import poll
    from keras_retinanet_3D.models import weights as Wt
    from keras_retinanet_3D.utils import ledger, synthetic
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import corner_deviation as CD
    torch.backends.cuda.matmul.allow_tf32 = False
    global ACCUMULATION
    ACCUMULATION = args.accumulation
    backbone, db = args.config.split('_')
    full = (args.height, args.width) == (402, 1333)
    lib_path = os.path.join(ROOT, 'oracle', 'liboracle_polling.so')
    if not os.path.isfile(lib_path):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(lib_path)
    g64 = CD.load_golden(args.config, 'f64', args.frames) if full else None
    g32 = CD.load_golden(args.config, 'f32', args.frames) if full else None
    n = g64[1].shape[0] if full else (args.frames or 1)
    planes = np.ascontiguousarray(synthetic.load_plane_database(db), np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    P_inv = P_inv.astype(np.float32)
    anchors = decode_np.anchors_for_image((args.height, args.width))
    weights = Wt.synthetic_weights(backbone, 1234)
    runs = [('control: every layer direct x3', {})]
    for mode, label in (('w2', 'F(2,3) along W'), ('w4', 'F(2x2,3x3)')):
        for what, names in layer_sets(backbone).items():
            runs.append(('{} on {}'.format(label, what), {nm: mode for nm in names}))
    if args.only:
        runs = [r for r in runs if any(s in r[0] for s in args.only.split(';'))]
    report = {'config': args.config, 'frames': int(n), 'device': args.device, 'accumulation': args.accumulation, 'rows': {}}

    def print_row(name, r):
        d = r['distribution']
        print(' | '.join(CD.fmt(v) for v in (name, '{}/{}'.format(r['common'], r['union']), r['set_differences_unexplained'], r['set_differences_at_a_tie'],
                                            '{}/{}'.format(r['same_plane'], r['common']), r['plane_differences_with_equal_inputs'],
                                            d.get('n_within_100m'), d.get('corner_p50', 0.0), d.get('corner_p99', 0.0), d.get('corner_max', 0.0),
                                            d.get('corner_above_1e-3', 0), d.get('scaled_beyond_max', 0.0), r['meets_reference_bars'])), flush=True)

    if full:
        print('{}: {} frames against tests/golden/fullsize_{}_f64.npz; bars: utils/ledger.REFERENCE_BARS, unchanged; accumulation model: {}'.format(args.config, n, args.config, args.accumulation))
        print(' | '.join(('run', 'dets', 'unexplained set diff', 'ties', 'plane', 'flips', 'n<=100m', 'p50', 'p99', 'max', '>1e-3', 'scaled max >100m', 'bars met')))
        report['rows']['float32 CPU oracle vs f64 oracle (float32 itself)'] = CD.compare(g64, g32, ledger)
        print_row('float32 CPU oracle vs f64 oracle (float32 itself)', report['rows']['float32 CPU oracle vs f64 oracle (float32 itself)'])
    for title, modes in runs:
        net = EmuNet(weights, backbone, modes, args.device)
        outs, aidx, pidx = {k: [] for k in range(8)}, [], []
        t0 = time.time()
        for seed in range(n):
            img = synthetic.synthetic_network_input([seed])[:, :args.height, :args.width]
            f = net.forward(img)
            det, ai = decode_np.detect(f['classification_logits'], f['regression'], f['regression_dim'], anchors)
            kp, kpl, res, idx = poll(lib, det, P_inv, planes)
            for k, v in enumerate(list(det) + [kp, kpl, res]):
                outs[k].append(v)
            aidx.append(ai.astype(np.int32))
            pidx.append(idx)
            if seed == 0 or seed % 8 == 7:
                print('  [{}] frame {}: {:.1f} s, {} layers in a Winograd form'.format(title, seed, time.time() - t0, sum(1 for m in net.used.values() if m != 'x3')),
                      file=sys.stderr, flush=True)
        got = ([np.concatenate(outs[k]) for k in range(8)], np.concatenate(aidx), np.concatenate(pidx))
        if full:
            row = CD.compare(g64, got, ledger)
            row['set_difference_list'] = CD.set_differences(g64, got, g64[3])
            row['winograd_layers'] = sorted(nm for nm, m in net.used.items() if m != 'x3')
            row['seconds'] = round(time.time() - t0, 1)
            report['rows'][title] = row
            print_row(title, row)
            if args.json:
                with open(args.json, 'w') as f:
                    json.dump(report, f, indent=1, default=float)
        else:
            print(title, 'detections', int((got[0][2] > 0.05).sum()), 'planes', got[2][0][:5])


if __name__ == '__main__':
    main()
