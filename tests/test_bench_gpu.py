""" The driver's contract for bench.py: one JSON line with the agreed keys (run as a child process, like the driver does). """
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--cpu-images', '1',
                          '--all-dtypes'],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in rec, key
    assert rec['n_gpus'] == 1 and rec['steps'] == 3 and rec['warmup'] == 1 and rec['unit'] == 'images/s' and rec['vs_baseline'] is None
    assert rec['higher_is_better'] is True and rec['scaling'] == 'weak' and rec['dtype'] == 'f16x3' and rec['data'] == 'synthetic'
    assert 'workload' in rec['config'] and 'model' not in rec['config']
    assert abs(rec['value'] - 8 * 3 / (rec['ms_per_step'] * 3e-3)) < 0.01 * rec['value']
    roof = rec['roofline']
    assert roof['bound'] == 'mfma' and roof['unit'] == 'TFLOP/s' and abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    assert 0.2 < roof['frac'] < 1.0 and roof['launches_timed'] == 3                 # 3 regression-tower launches on 1 of the 3 steps
    cpu = rec['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['cores'] >= 1 and cpu['value'] > 0 and 'sample' in cpu
    poll = rec['config']['polling_kernel']
    assert poll['planes'] == 1000 and poll['launch_us'] > 0 and 0 < poll['frac_of_hbm_peak'] < 0.01
    cfg = rec['config']
    # the reference-precision leg and the ledger of the headline type against it, measured by the same run: the headline has to
    # meet BASELINE.json's tolerance (same detections, same planes, corners within 1e-3 m) and its throughput target (>= 500)
    assert cfg['f32_images_per_s'] > 20 and 0.2 < cfg['f32_frac_of_f32_mfma_peak'] < 1.0
    led = cfg['parity_ledger']                              # f16x3 HIP vs f32 HIP: two float32-grade runs, twice the metre bars
    assert led['images'] == 8 and led['detections_ref'] > 400
    assert led['set_differences_unexplained'] == 0 and led['plane_index_agreement'] == 1.0 and led['max_corner_dev_m_within_100m'] <= 2e-3
    assert led['meets_reference_bars_as_a_pair'] is True
    exact = cfg['parity_ledger_vs_f64_oracle']              # against the float64 CPU oracle (committed fixture): THE bars
    assert exact['images'] == 8 and exact['common'] == exact['union'] == 800 and exact['same_plane'] == 800
    assert exact['max_corner_dev_m_within_100m'] <= 1e-3 and exact['max_corner_dev_scaled_beyond_100m'] <= 1e-3 and exact['meets_reference_bars'] is True
    assert cfg['parity_ledger_vs_f32_cpu_oracle']['meets_reference_bars'] is True
    assert cfg['parity_bars_met'] is True and rec['value'] >= 500.0
    others = cfg['other_types_same_frames']               # --all-dtypes: the faster types, none of which meets the bars with these weights
    assert others['bf16']['images_per_s'] > 1000 and others['bf16']['meets_reference_bars'] is False
    assert others['f16']['images_per_s'] > 500 and others['bf16x3']['images_per_s'] > 200
    assert others['bf16x3']['parity_ledger_vs_f32']['detection_set_agreement'] >= others['bf16']['parity_ledger_vs_f32']['detection_set_agreement']
    assert cfg['gpu_decode_polling_replay_bit_exact'] is True and cfg['resident_batches_rotated'] >= 4
    assert cfg['host_fed_streaming_images_per_s'] > 0.5 * rec['value'] and cfg['rccl_world_size'] == 1
    assert roof['library'].startswith('gpp-hip') and 'src:' in roof['library']
    # round 5: the line's own error bar, and the reference's own timer (one synchronous batch-1 call, bin/run_network.py:108-111)
    assert len(cfg['repeat_images_per_s']) == 3 and all(v > 400.0 for v in cfg['repeat_images_per_s']) and len(roof['frac_per_repeat']) == 3
    b1 = cfg['b1']
    assert 0.5 < b1['plan_only_ms_median'] < b1['sync_ms_median'] < 20.0 and b1['sync_ms_p90'] >= b1['sync_ms_median'] and b1['detections_last_call'] > 0
    assert set(b1['stages_ms']) == {'stem+backbone', 'fpn', 'heads+selection', 'emit+polling'} and b1['floor_ms'] < b1['plan_only_ms_median']


@pytest.mark.gpu
def test_bench_at_reference_precision():
    """ --dtype f32: the same JSON line for the float32 conv path, roofline against the float32 MFMA peak """
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dtype', 'f32', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline', '--no-host-fed'],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert rec['dtype'] == 'f32' and rec['roofline']['peak'] == 157.3 and 0.3 < rec['roofline']['frac'] < 1.0
    assert rec['value'] > 20 and rec['config']['parity_ledger'] is None and rec['roofline']['launches_timed'] == 3


@pytest.mark.gpu
def test_bench_stdout_is_one_json_line_on_the_rccl_path():
    """ the multi-GPU path (RCCL communicator, asynchronous all_gather of the packed detections), forced onto the one GPU of the
    test box: RCCL prints a version banner on stdout when its first communicator comes up -- bench.py keeps its stdout to the
    one JSON line the driver parses """
    env = dict(os.environ, GPP_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29641')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '1', '--no-cpu-baseline',
                          '--no-f32-leg', '--no-host-fed'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                         timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[:500]
    rec = json.loads(lines[0])
    assert rec['config']['rccl_world_size'] == 1 and rec['config']['gathered_images_per_step'] == 8 and rec['n_gpus'] == 1
    diag = rec['config']['multi_gpu_diagnosis']               # what a scaling run is read with: per-rank step time, exposed gather time
    assert 0 < diag['ms_per_step_min_over_ranks'] <= diag['ms_per_step_max_over_ranks'] <= rec['ms_per_step'] * 1.05
    assert 0 <= diag['gather_wait_ms_per_step_max_over_ranks'] < rec['ms_per_step']


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_the_box_has():
    """ `bench.py --gpus 2` on a one-GPU box: the launcher counts the devices before it starts a rank and fails loudly -- it never
    reports a 2-GPU line from one rank (tests/test_bench_launcher.py rehearses the launcher itself on gloo) """
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('this box has the devices')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and 'needs 2 devices' in out.stderr and out.stdout.strip() == ''
