"""The benchmark script itself: it must run end to end and print ONE JSON line with the contract's fields."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '2', '--warmup', '1', '--frames', '2',
                          '--points', '4000', '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['unit'] == 'frames/s' and d['n_gpus'] == 1 and d['steps'] == 2 and d['value'] > 0
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'mfma' and 0 < r['frac'] < 1 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert d['alt_modes'] and all(a['value'] > 0 for a in d['alt_modes'])
    assert {a['workload'] for a in d['alt_modes']} == {'S1', 'S2'}
    assert d['config']['frame_sets'] and 'crop_project' in d['hbm_stages']


@pytest.mark.gpu
def test_bench_vfe_mode_checks_voxel_indices_against_the_oracle():
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--mode', 'vfe', '--steps', '2', '--warmup', '1', '--frames',
                          '3', '--points', '4000'], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith('{')][0])
    assert d['roofline']['bound'] == 'hbm' and d['value'] > 0
    assert d['cpu_baseline']['voxel_indices_vs_oracle'].startswith('ok') and d['cpu_baseline']['value'] > 0
