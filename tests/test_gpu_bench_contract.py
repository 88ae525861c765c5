"""The measurement contract of bench.py (one JSON line on stdout with `roofline` and `cpu_baseline`), exercised at a small size."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line
    return json.loads(lines[0])


def test_headline_line_has_the_contract_fields():
    d = _run('--rows', '131072', '--queries', '1024', '--steps', '3', '--warmup', '1', '--cpu-sample-queries', '16')
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['metric'] == 'queries/sec' and d['unit'] == 'queries/s' and d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1
    assert d['higher_is_better'] is True and d['vs_baseline'] is None and d['dtype'] == 'bf16' and 'synthetic' in d['data']
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 1024 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6      # whole-job throughput of the timed steps
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0
    assert 0.0 < r['frac'] < 1.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert r['kernel_ms_per_step'] <= d['ms_per_step']                                   # the dominant kernels ran inside the timed region
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == 'queries/s' and c['sample']
    assert d['recall@1'] == 1.0 and d['overflowed_queries'] == 0
    # the secondary workloads ride on the same line (measured after the timed region; none of them enters `value`)
    sec = d['secondary']
    assert 'error' not in sec, sec
    for name in ('flickr_1k', 'coco_5k'):
        assert sec[name]['ms_per_evaluation'] > 0 and sec[name]['recall_t2i@1'] == 1.0 and sec[name]['recall_i2t@1'] == 1.0
        lat = sec[name]['latency_ms']                                                   # tails, not only medians
        assert lat['calls'] >= 200 and 0 < lat['p50'] <= lat['p99'] <= lat['worst'] and lat['p50'] == sec[name]['ms_per_evaluation']
    for key in ('1q_x_headline_index', '64q_x_headline_index', '1q_x_123k', '64q_x_123k'):
        e = sec['serving_latency'][key]
        assert e['ms'] > 0 and 0.0 < e['hbm_frac_whole_search'] < 1.0 and e['rank1_ok']
        assert e['latency_ms']['calls'] >= 200 and e['latency_ms']['p50'] <= e['latency_ms']['p99'] <= e['latency_ms']['worst']
    # the mining searches of dvl/hn.py:45-66 at the Flickr30k train set's size, top-50 and top-1000, exact scores and ids-only
    mn = sec['mining_flickr_train']
    assert mn['images'] == 29000 and mn['captions'] == 145000 and 0 < mn['sampled_hard_negatives_wall_s'] < 2.0
    for k in (50, 1000):
        for name in (f't2i_145k_x_29k_top{k}', f'i2t_29k_x_145k_top{k}'):
            row = mn['searches'][name]
            for mode in ('exact_scores', 'ids_only'):
                m = row[mode]
                assert 0 < m['score_kernel_ms'] <= m['device_ms'] and 0 < m['score_kernel_frac'] < 1.0
            assert 0 < row['ids_only']['candidates_rescored_share'] < 0.5
            assert row['ids_only']['rows_gathered_GB'] < row['exact_scores']['rows_gathered_GB']
        assert mn['searches'][f't2i_145k_x_29k_top{k}']['exact_scores']['rank1_ok'] and mn['searches'][f't2i_145k_x_29k_top{k}']['ids_only']['rank1_ok']
        assert 'exact_scores' in mn['searches'][f'i2t_undeduplicated_145k_x_145k_top{k}']
    for key in ('512x512', '512x1536'):
        e = sec['loss_step'][key]
        assert 0 < e['device_us'] <= e['end_to_end_us'] * 1.5 and e['torch_end_to_end_us'] > 0 and e['torch_device_us'] > 0
    iv = sec['ivf_123k']
    assert iv['ms_1_query'] > 0 and iv['recall@10_vs_exact'] > 0.8 and iv['nprobe'] == 32


def test_serving_line_reports_an_hbm_roofline():
    d = _run('--workload', 'serving', '--rows', '200000', '--steps', '10', '--warmup', '2', '--no-cpu-baseline')
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and 0.0 < r['frac'] < 1.0
    assert d['config']['workload'] and d['ms_per_step'] > 0


def test_sharded_path_on_rccl_with_one_rank():
    """`--force-sharded` at world size 1: ShardedFlatIndexer on the nccl (= RCCL) backend — all-gather of the queries, all-reduce(MAX) of
    the shard statistics, the blocked all-to-all of the partial lists, the merge into pinned host buffers — with the one rank a GPU box
    offers.  (More ranks: gloo tests on one GPU, tests/test_gpu_sharded.py; the driver's multi-GPU runs.)"""
    d = _run('--rows', '131072', '--queries', '1024', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-secondary',
             '--force-sharded', '--backend', 'nccl')
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['recall@1'] == 1.0 and d['results_sorted'] is True
    assert d['overflowed_queries'] == 0 and d['roofline']['kernel_ms_per_step'] <= d['ms_per_step']
    # per-phase device times of the exchange: every phase of the default path is there, they add up to the total, the total fits the step
    ph = d['phases_ms_per_step']
    for name in ('all_gather_queries', 'candidate_pass', 'all_reduce_statistics', 'floor', 'rescore', 'exchange_lists', 'merge', 'total'):
        assert name in ph and ph[name] >= 0.0, name
    parts = sum(v for k, v in ph.items() if k != 'total')
    assert abs(parts - ph['total']) <= 0.05 * ph['total'] and ph['total'] <= d['ms_per_step'] * 1.02
    assert ph['candidate_pass'] >= d['roofline']['kernel_ms_per_step'] * 0.98


def test_sharded_forced_repeat_reports_the_verdict_and_repeat_phases():
    """`--force-repeat`: every step runs the pooled scheme's verdict (all-reduce SUM of the counts + one host read) and then the whole search
    again — what a failed pooled search costs, on RCCL with the one rank a GPU box offers."""
    d = _run('--rows', '131072', '--queries', '1024', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-secondary',
             '--force-sharded', '--force-repeat', '--backend', 'nccl')
    assert d['ranks_seen'] == 1 and d['recall@1'] == 1.0 and d['results_sorted'] is True and 'forced_repeat' in d
    ph, phm = d['phases_ms_per_step'], d['phases_ms_per_step_slowest_rank']
    assert ph['verdict'] >= 0.0 and ph['repeat:candidate_pass'] > 0.0 and ph['repeat:merge'] > 0.0 and set(ph) == set(phm)
    assert all(abs(ph[k_] - phm[k_]) < 1e-9 for k_ in ph)                               # one rank: the slowest rank is rank 0
