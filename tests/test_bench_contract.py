"""bench.py's output contract (one JSON line with the driver's keys, the roofline and cpu_baseline objects) on a
tiny run; the multi-rank path is rehearsed in tests/test_parallel_gloo.py and by hand."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '24', '--warmup', '8'],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['metric'] == 'images/sec' and d['unit'] == 'img/s' and d['n_gpus'] == 1 and d['steps'] == 24
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert r['kernel_ms'] > 0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and r['kernel_ms_samples'] >= 1
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'img/s' and c['value'] > 0 and c['cores'] >= 1 and 'sample' in c
    assert abs(c['map_delta']['delta']) <= 0.002
    assert d['value'] > 0 and abs(d['ms_per_step'] * d['value'] - 1000.0) < 1.0
