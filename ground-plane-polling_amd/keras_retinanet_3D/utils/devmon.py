"""
What the amdgpu driver reports about THIS process's card while it works: shader clock and board power, read from sysfs (hwmon freq1_input,
power1_input / power1_average, power1_cap) by a sampling thread.  No GPU call, no child process: file reads every few tens of milliseconds.

The dominant kernel of this path runs at the clock the power management grants under matrix load, and that clock depends on the operand
data (HISTORY.md 4.10: 2006 MHz at 1395 W of a 1400 W cap on real data, 2398 MHz on zeros, launch time in the same ratio).  `bench.py`
carries these medians next to `roofline` so that a reader can tell a schedule from a clock.  Everything here is optional: on a host that
hides sysfs the sampler returns nothing.
"""
import glob
import os
import threading
import time


def card_of_device(index=0):
    """ sysfs directory (/sys/class/drm/cardN/device) of HIP device `index` of this process, or None """
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        mine = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:                                    # no device, or a torch without the PCI fields
        return None
    for card in sorted(glob.glob('/sys/class/drm/card*/device')):
        if os.path.basename(os.path.realpath(card)) == mine:
            return card
    return None


def sources(card):
    out = {}
    if not card:
        return out
    for hw in glob.glob(os.path.join(card, 'hwmon', 'hwmon*')):
        for key, names in (('sclk_mhz', ('freq1_input',)), ('power_w', ('power1_average', 'power1_input')), ('power_cap_w', ('power1_cap',))):
            for name in names:
                p = os.path.join(hw, name)
                if key not in out and os.path.isfile(p):
                    out[key] = p
    return out


def read(src):
    rec = {}
    for key, path in src.items():
        try:
            rec[key] = float(open(path).read()) / 1e6     # Hz -> MHz, uW -> W
        except (OSError, ValueError):
            pass
    return rec


class Sampler(object):
    """ with Sampler(device_index) as s: ...work...; s.summary() -> {'sclk_mhz_median': ..., 'power_w_median': ..., 'power_cap_w': ..., 'samples': n} """

    def __init__(self, index=0, period=0.02):
        self.src = sources(card_of_device(index))
        self.period = period
        self.samples = []
        self._stop = threading.Event()
        self._thread = None

    def __enter__(self):
        if self.src:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            rec = read(self.src)
            if rec:
                self.samples.append(rec)
            self._stop.wait(self.period)                 # (returns at once when the region ends, whatever the period)

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
        return False

    def summary(self):
        if not self.samples:
            return None
        out = {'samples': len(self.samples), 'source': 'amdgpu sysfs hwmon of this rank\'s card, every {:.0f} ms over the timed region'.format(self.period * 1e3)}
        for key in ('sclk_mhz', 'power_w'):
            v = sorted(s[key] for s in self.samples if key in s)
            if v:
                out[key + '_median'] = round(v[len(v) // 2], 1)
                out[key + '_min'] = round(v[0], 1)
                out[key + '_max'] = round(v[-1], 1)
        caps = [s['power_cap_w'] for s in self.samples if 'power_cap_w' in s]
        if caps:
            out['power_cap_w'] = round(caps[0], 1)
        return out
