"""
CPU tests of utils/devmon.py (the sysfs clock / power sampler bench.py runs during its timed region): nothing to read -> no summary;
a fake hwmon directory -> medians in MHz and W.
"""
import os
import time

from keras_retinanet_3D.utils import devmon


def test_no_card_no_summary():
    with devmon.Sampler(0) as s:
        time.sleep(0.05)
    assert devmon.card_of_device(0) is None or isinstance(devmon.card_of_device(0), str)
    assert s.summary() is None or s.summary()['samples'] > 0


def test_a_fake_hwmon_directory_is_read_in_mhz_and_watts(tmp_path):
    hw = tmp_path / 'hwmon' / 'hwmon3'
    hw.mkdir(parents=True)
    (hw / 'freq1_input').write_text('2006000000\n')
    (hw / 'power1_input').write_text('1395000000\n')
    (hw / 'power1_cap').write_text('1400000000\n')
    src = devmon.sources(str(tmp_path))
    assert set(src) == {'sclk_mhz', 'power_w', 'power_cap_w'}
    assert devmon.read(src) == {'sclk_mhz': 2006.0, 'power_w': 1395.0, 'power_cap_w': 1400.0}
    s = devmon.Sampler.__new__(devmon.Sampler)
    s.src, s.period, s.samples = src, 0.005, []
    import threading
    s._stop, s._thread = threading.Event(), None
    with s:
        time.sleep(0.06)
    out = s.summary()
    assert out['samples'] >= 3 and out['sclk_mhz_median'] == 2006.0 and out['power_w_median'] == 1395.0 and out['power_cap_w'] == 1400.0
    assert devmon.sources(None) == {} and devmon.sources(os.path.join(str(tmp_path), 'missing')) == {}


def test_the_sampler_stops_at_once_whatever_its_period(tmp_path):
    """ round 5: a sampler with a long period used to sleep it out in __exit__ (a bench run with GPP_DEVMON_PERIOD_S=1000 hung) """
    import threading
    hw = tmp_path / 'hwmon' / 'hwmon0'
    hw.mkdir(parents=True)
    (hw / 'freq1_input').write_text('2100000000\n')
    s = devmon.Sampler.__new__(devmon.Sampler)
    s.src, s.period, s.samples = devmon.sources(str(tmp_path)), 1000.0, []
    s._stop, s._thread = threading.Event(), None
    t0 = time.time()
    with s:
        time.sleep(0.05)
    assert time.time() - t0 < 2.0 and s.summary()['samples'] == 1
