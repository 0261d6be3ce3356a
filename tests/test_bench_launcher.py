""" bench.py --gpus N starts its own N ranks when no launcher did (CPU rehearsal on gloo: --dry-launch), refuses to report an
N-GPU line from fewer devices, and refuses a WORLD_SIZE that differs from --gpus. """
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(kw)
    return env


def test_gpus_2_launches_two_ranks_by_itself():
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-launch', '--batch', '3'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=300, cwd=ROOT, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout                          # rank 0 alone writes to stdout
    rec = json.loads(lines[0])
    assert rec['world_size'] == 2 and rec['n_gpus'] == 2 and rec['gathered_images_per_step'] == 6 and rec['gather_correct'] is True


def test_gpus_2_without_two_devices_fails_loudly():
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=300, cwd=ROOT, env=_env())
    assert out.returncode != 0 and 'needs 2 devices' in out.stderr and out.stdout.strip() == ''


def test_a_world_size_that_differs_from_gpus_is_refused():
    out = subprocess.run([sys.executable, BENCH, '--gpus', '4', '--dry-launch'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=300, cwd=ROOT,
                         env=_env(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29655'))
    assert out.returncode != 0 and 'WORLD_SIZE=2' in out.stderr


def test_a_failing_rank_ends_the_job_with_its_exit_code():
    # rank 1 of 2 cannot reach a GPU here: under the launcher the whole job must come back non-zero, not hang
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-launch', '--batch', '0'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=300, cwd=ROOT, env=_env())
    assert out.returncode != 0


def test_gpus_8_launches_eight_ranks_and_prints_the_diagnosis_block():
    # the shape of the driver's scaling run (N = 8), rehearsed on gloo: eight children, one stdout line, the per-rank diagnosis keys of a real line
    out = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--dry-launch', '--batch', '8'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=600, cwd=ROOT, env=_env(OMP_NUM_THREADS='1'))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['world_size'] == 8 and rec['n_gpus'] == 8 and rec['gathered_images_per_step'] == 64 and rec['gather_correct'] is True
    diag = rec['multi_gpu_diagnosis']
    assert diag['ranks'] == 8
    for key in ('ms_per_step', 'gather_wait_ms_per_step', 'sclk_mhz_median', 'power_w_median', 'power_cap_w', 'model_load_s', 'plan_build_and_tune_s'):
        assert set(diag[key]) == {'per_rank', 'min', 'median', 'max'} and len(diag[key]['per_rank']) == 8
    assert diag['sclk_mhz_median']['per_rank'] == [float(r) for r in range(8)] and diag['sclk_mhz_median']['median'] == 3.5
    assert diag['power_cap_w']['per_rank'] == [None] * 8 and diag['power_cap_w']['max'] is None       # a figure no rank could read stays empty


def test_a_failing_rank_of_eight_ends_the_job():
    out = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--dry-launch', '--batch', '0'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=600, cwd=ROOT, env=_env(OMP_NUM_THREADS='1'))
    assert out.returncode != 0 and out.stdout.strip() == ''
