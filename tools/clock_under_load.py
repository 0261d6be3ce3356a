#!/usr/bin/env python
"""
The clock and the power the device reports while the dominant kernel runs -- on real operand data and on zeros.

HISTORY.md 4.9 / 4.10 (DESIGN.md 4.1 in short) say the regression-tower kernel is bound by the clock the power management grants under matrix load, and that the
clock depends on the operand data.  The evidence so far was indirect (the same launch is 25 % faster on all-zero activations; the
register-only loops of tools/micro).  This reads what the driver itself reports -- sysfs (hwmon freq1_input / power1_average, pp_dpm_sclk)
or, failing that, `rocm-smi --showclocks --showpower --json` -- every 50 ms while one thread launches the layer back to back for a few seconds.

    python tools/clock_under_load.py [seconds per case, default 4]
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))

import torch  # noqa: E402
from keras_retinanet_3D.layers import conv as C  # noqa: E402

PYR = [(51, 167), (26, 84), (13, 42), (7, 21), (4, 11)]


def my_pci_address():
    """ PCI address of HIP device 0 of this process (a box has eight cards; ours is one of them) """
    p = torch.cuda.get_device_properties(0)
    try:
        return '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except AttributeError:
        return None


def sysfs_sources():
    out = {}
    mine = my_pci_address()
    for card in sorted(glob.glob('/sys/class/drm/card*/device')):
        if not os.path.isfile(os.path.join(card, 'gpu_busy_percent')):
            continue
        if mine and os.path.basename(os.path.realpath(card)) != mine:
            continue
        for hw in glob.glob(os.path.join(card, 'hwmon', 'hwmon*')):
            for name in ('freq1_input', 'power1_average', 'power1_input'):
                p = os.path.join(hw, name)
                if os.path.isfile(p):
                    out.setdefault(card, {})[name] = p
        p = os.path.join(card, 'pp_dpm_sclk')
        if os.path.isfile(p):
            out.setdefault(card, {})['pp_dpm_sclk'] = p
    return out


def read_sysfs(src):
    rec = {}
    for name, path in src.items():
        try:
            text = open(path).read()
        except OSError:
            continue
        if name == 'pp_dpm_sclk':
            for line in text.splitlines():
                if '*' in line:
                    rec['sclk_mhz_dpm'] = float(line.split(':')[1].replace('Mhz', '').replace('*', '').strip())
        elif name == 'freq1_input':
            rec['sclk_mhz'] = float(text) / 1e6
        else:
            rec['power_w'] = float(text) / 1e6
    return rec


def read_smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, timeout=5).stdout
        d = json.loads(out)
    except (OSError, ValueError, subprocess.SubprocessError):
        return {}
    rec = {}
    for card, fields in d.items():
        for k, v in fields.items():
            kl = k.lower()
            if 'sclk' in kl and 'mhz' in str(v).lower():
                rec['sclk_mhz'] = float(str(v).lower().replace('(', '').replace(')', '').replace('mhz', '').strip())
            elif 'power' in kl and 'w' in kl:
                try:
                    rec['power_w'] = float(v)
                except (TypeError, ValueError):
                    pass
        break
    return rec


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    dev = torch.device('cuda')
    B, cin, cout, dtype = 8, 512, 512, 'f16x3'
    total = sum(h * w for h, w in PYR)
    torch.manual_seed(1)
    wk = (torch.randn((3, 3, cin, cout)) * 0.02).numpy()
    w = C.pack_weight(wk, dtype, dev)
    sc = C.out_scale_of(wk, dev)
    bias = torch.zeros((cout,), device=dev)
    src = sysfs_sources()
    card = sorted(src)[0] if src else None
    print('HIP device 0 is PCI', my_pci_address())
    print('reading', ('sysfs ' + card + ' ' + ','.join(sorted(src[card]))) if card else 'rocm-smi --json (no sysfs counters visible)')
    flops = 2.0 * B * total * 9 * cin * cout
    for case, scale in (('post-ReLU random activations', 1.0), ('all-zero activations', 0.0), ('post-ReLU random activations (again)', 1.0)):
        ib = torch.empty((B, total, cin), device=dev)
        ob = torch.empty((B, total, cout), device=dev)
        ins, outs, off = [], [], 0
        for h, wd in PYR:
            ins.append(C.FMap(ib, B, h, wd, cin, off=off * cin, bstride=total * cin, split=True, half=dtype))
            outs.append(C.FMap(ob, B, h, wd, cout, off=off * cout, bstride=total * cout, split=True, half=dtype))
            ins[-1].write(torch.relu(torch.randn((B, h, wd, cin), device=dev)) * scale)
            off += h * wd
        d = C.conv_desc(ins, outs, w, bias, 3, 3, cin, cout, pad=(1, 1), relu=True, dtype=dtype, tile_hint=3256224, out_scale=sc)
        for _ in range(5):
            C.run_conv(d)
        torch.cuda.synchronize()
        stop = threading.Event()
        launches = [0]

        def load():
            torch.cuda.set_device(0)
            while not stop.is_set():
                for _ in range(20):
                    C.run_conv(d)
                torch.cuda.synchronize()
                launches[0] += 20

        t = threading.Thread(target=load)
        t0 = time.perf_counter()
        t.start()
        samples = []
        time.sleep(0.5)                                   # let the clock settle
        n0, ts0 = launches[0], time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            rec = read_sysfs(src[card]) if card else read_smi()
            if rec:
                samples.append(rec)
            time.sleep(0.05)
        n1, ts1 = launches[0], time.perf_counter()
        stop.set()
        t.join()
        us = (ts1 - ts0) / max(1, n1 - n0) * 1e6
        line = '%-40s %7.1f us per launch = %5.1f TFLOP/s of float32 products' % (case, us, flops / us / 1e6)
        for key, unit in (('sclk_mhz', 'MHz'), ('sclk_mhz_dpm', 'MHz (dpm level)'), ('power_w', 'W')):
            v = sorted(s[key] for s in samples if key in s)
            if v:
                line += '   %s: median %.0f, min %.0f, max %.0f %s (%d samples)' % (key.split('_')[0], v[len(v) // 2], v[0], v[-1], unit, len(v))
        print(line)
    # an HBM-bound layer of the backbone (res2x branch2a, 1 x 1, 256 -> 64 on the 101 x 334 map: 345 MB per launch) and the whole step, for contrast
    def under_load(name, launch, work_note):
        stop = threading.Event()
        count = [0]

        def load():
            torch.cuda.set_device(0)
            while not stop.is_set():
                for _ in range(10):
                    launch()
                torch.cuda.synchronize()
                count[0] += 10

        t = threading.Thread(target=load)
        t0 = time.perf_counter()
        t.start()
        samples = []
        time.sleep(0.5)
        n0, ts0 = count[0], time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            rec = read_sysfs(src[card]) if card else read_smi()
            if rec:
                samples.append(rec)
            time.sleep(0.05)
        n1, ts1 = count[0], time.perf_counter()
        stop.set()
        t.join()
        us = (ts1 - ts0) / max(1, n1 - n0) * 1e6
        line = '%-40s %8.1f us per launch %s' % (name, us, work_note(us))
        for key, unit in (('sclk_mhz', 'MHz'), ('power_w', 'W')):
            v = sorted(s[key] for s in samples if key in s)
            if v:
                line += '   %s: median %.0f, min %.0f, max %.0f %s' % (key.split('_')[0], v[len(v) // 2], v[0], v[-1], unit)
        print(line)

    H2, W2 = 101, 334
    xb = torch.empty((B, H2 * W2, 256), device=dev)
    yb = torch.empty((B, H2 * W2, 64), device=dev)
    fi = C.FMap(xb, B, H2, W2, 256, split=True, half=dtype)
    fo = C.FMap(yb, B, H2, W2, 64, split=True, half=dtype)
    fi.write(torch.relu(torch.randn((B, H2, W2, 256), device=dev)))
    wk1 = (torch.randn((1, 1, 256, 64)) * 0.05).numpy()
    d1 = C.conv_desc([fi], [fo], C.pack_weight(wk1, dtype, dev), torch.zeros((64,), device=dev), 1, 1, 256, 64, relu=True, dtype=dtype, tile_hint=4064064,
                     out_scale=C.out_scale_of(wk1, dev))
    mb = (xb.numel() + yb.numel()) * 4 / 1e6
    under_load('res2x branch2a 1x1 256->64 (HBM-bound)', lambda: C.run_conv(d1), lambda us: '= %.2f TB/s of compulsory bytes' % (mb / us))

    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from keras_retinanet_3D import models
    from keras_retinanet_3D.utils import synthetic
    model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype=dtype)
    planes = synthetic.load_plane_database('1k').astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    images = torch.as_tensor(bench.synthetic_batch(B, 0)).cuda()
    P = torch.as_tensor(np.tile(P_inv[None].astype(np.float32), (B, 1, 1))).cuda()
    pl = torch.as_tensor(np.tile(planes[None], (B, 1, 1))).cuda()
    plan = model.stage_inputs([images, P, pl])
    for _ in range(3):
        model.run_plan(plan)
    torch.cuda.synchronize()
    under_load('the whole step (default plan, B = 8)', lambda: model.run_plan(plan), lambda us: '= %.1f images/s' % (B / us * 1e6))
    time.sleep(1.0)
    idle = read_sysfs(src[card]) if card else read_smi()
    print('idle, one second later:', idle)


if __name__ == '__main__':
    main()
