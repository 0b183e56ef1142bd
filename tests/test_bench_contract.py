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
    # (the driver's own command line plus --no-e2e: the end-to-end detector record takes a minute of MIOpen searches)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5',
                        '--no-e2e'], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['metric'] == 'images/sec' and d['unit'] == 'img/s' and d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert r['kernel_ms'] > 0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and r['kernel_ms_samples'] >= 10
    assert r['B_min'] <= r['algorithmic_bytes'] <= r['B_taps'] and r['bytes_output'] < r['B_min']
    assert abs(r['achieved'] - r['algorithmic_bytes'] / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-6 * r['achieved']
    if r['traffic'] is not None:
        assert abs(r['hbm_frac_measured'] - r['traffic'] / (r['kernel_ms'] * 1e-3) / 8e12) < 1e-9
    assert 'k_roi_pool' in r['rocprof_kernel_name']
    cal = r['calibration']            # the same bytes moved by a kernel that does nothing else, same protocol
    assert cal['samples'] >= 5 and cal['ms'] > 0 and cal['bytes_written'] <= r['bytes_output']
    assert abs(cal['roi_kernel_vs_calibration'] - cal['ms'] / r['kernel_ms']) < 1e-9
    assert 0.3 < cal['roi_kernel_vs_calibration'] < 1.5
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'img/s' and c['value'] > 0 and c['cores'] >= 1 and 'sample' in c
    assert abs(c['map_delta']['delta']) <= 0.002
    # both thread modes of BASELINE.md section 3, a bounded sample of each
    assert c['value_1thread'] > 0 and c['samples'] >= 16 and c['samples_1thread'] >= 4
    # the other RPN score distribution (trained-like clusters) in the same line, its own timed region; the bench widens
    # its sync-free NMS plan by itself instead of aborting
    assert d['value_clustered'] > 0 and d['ms_per_step_clustered'] > 0
    sd = d['config']['second_distribution']
    assert sd['rpn_scores'] == 'clustered' and sd['proposals_kept'] == 1000
    assert abs(sd['ms_per_step'] * sd['value'] / (1000.0 * d['config']['images_per_step_per_gpu']) - 1.0) < 1e-6
    # a step = images_per_step_per_gpu images on every GPU; the driver's 20 steps time at least half a second
    ips = d['config']['images_per_step_per_gpu']
    assert ips == 48 * d['config']['streams_per_gpu'] * d['config']['images_per_launch']
    assert d['value'] > 0 and abs(d['ms_per_step'] * d['value'] / (1000.0 * ips * d['n_gpus']) - 1.0) < 1e-6
    assert d['config']['timed_region_s'] >= 0.3 and d['config']['timed_images'] == 20 * ips
