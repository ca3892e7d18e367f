"""The N > 1 control flow of bench.py on the one-GPU test box: two ranks share the GPU and exchange gradients over gloo
(WDNO_DIST_BACKEND / WDNO_DIST_SHARE_GPU are test switches of wdno_amd.trainer.init_distributed). What is checked is what a real
8-GPU run over RCCL needs from the script itself: every collective is entered by every rank (rank 0's profiled extra step once hung
the other ranks' all-reduce), the max-over-ranks timing and the single JSON line of rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ, WDNO_DIST_BACKEND='gloo', WDNO_DIST_SHARE_GPU='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--sample-steps', '3']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 only
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['config']['parallelism'] == 'dp2' and out['config']['global_batch'] == 16
    assert len(out['per_rank']['ms_per_step']) == 2 and out['value'] > 0
    assert out['roofline'] is not None and out['cpu_baseline'] is None
    # `value` is rank-steps/s; the rate of optimiser steps on the global batch is printed next to it
    assert abs(out['global_steps_per_sec'] * 2 - out['value']) < 1e-3 * out['value'] and out['global_batch'] == 16 and 'rank-steps' in out['value_definition']
    assert 'skipped_legs' not in out and out['wall_seconds']['total'] < out['wall_seconds']['max_seconds']
    # the sampling half of the metric and configs[4] at N > 1 -- every rank samples its own batch / shard, global rates + per-rank lists
    assert 'sampling_error' not in out, out.get('sampling_error')
    assert out['ddpm_sample_steps_per_sec'] == out['sampling']['graph_steps_per_sec_global'] > 0
    assert out['sampling']['ranks'] == 2 and len(out['sampling']['per_rank']['graph_steps_per_sec']) == 2
    sr = out['sr_sampling']
    assert sr['ranks'] == 2 and sr['global_batch'] == 4 and len(sr['per_rank']['graph_ddim_steps_per_sec']) == 2 and sr['graph_ddim_steps_per_sec_global'] > 0
    assert sr['fields'] == [2, 5, 64, 128, 128]


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the form the driver uses at N = 1) must start two ranks itself and say so."""
    env = dict(os.environ, WDNO_DIST_BACKEND='gloo', WDNO_DIST_SHARE_GPU='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-extras']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['config']['global_batch'] == 16
    assert out['config']['process_group']['ranks'] == 2 and out['config']['process_group']['backend'] == 'gloo'
    assert len(out['per_rank']['ms_per_step']) == 2


def test_bench_refuses_a_rank_count_mismatch():
    """A process group whose size differs from --gpus must not print a line (CPU: the check sits before any GPU use... after the CUDA assert,
    so here only the launcher-less N > 1 path is exercised: without GPUs every rank fails loudly and the exit code is non-zero)."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    import torch
    if torch.cuda.is_available():
        env.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')          # one-rank environment, --gpus 2 requested
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and 'refusing' in (r.stderr + r.stdout)
    else:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and '{"metric"' not in r.stdout


@pytest.mark.gpu
def test_bench_max_seconds_guard_drops_side_legs_never_the_main_line():
    """with no time left (--max-seconds 1) every side leg -- at N > 1 the sampling / SR legs behind one collective decision -- is
    skipped and named, and the main line is still printed."""
    env = dict(os.environ, WDNO_DIST_BACKEND='gloo', WDNO_DIST_SHARE_GPU='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--sample-steps', '3', '--max-seconds', '1']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['roofline'] is not None
    assert 'sampling' not in out and 'sr_sampling' not in out and out['skipped_legs'][0]['leg'].startswith('sampling')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--max-seconds', '1'], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert out['value'] > 0 and out['roofline'] is not None and out['cpu_baseline'] == {'skipped': '--max-seconds'}
    assert {d['leg'] for d in out['skipped_legs']} >= {'sampling', 'dwt', 'sr_sampling', 'smoke_bf16', 'burgers.fp32_equivalent_batch16', 'cpu_baseline'}
