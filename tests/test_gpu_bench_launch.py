"""bench.py starts its own ranks: `python bench.py --gpus 2` (no external launcher) must exit 0 and its 2-rank ELBO must be the sum
of the two 1-rank shard ELBOs (the data term is a plain sum over points, onoffgpf/OnOffSVGP.py:119-122).  gloo backend: both
ranks share the box's one GPU; the RCCL leg itself needs >= 2 GPUs and is the driver's SCALE run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--rows', '65536', '--M', '256', '--steps', '2',
           '--warmup', '1', '--no-cpu-baseline', '--no-other-configs', '--profile-steps', '0'] + extra
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_weak(engine):
    sys.path.insert(0, ROOT)
    import bench
    res = _run([])
    assert res['n_gpus'] == 2 and res['n_ranks_seen'] == 2 and res['scaling'] == 'weak' and res['backend'] == 'gloo'
    assert res['config']['rows_total'] == 2 * 65536
    # the same two shards through the engine directly, one rank each
    tot, kl = 0.0, None
    engine.set_chunk(32768)
    for r in range(2):
        X, Y, p = bench.synth(65536, 256, 3, rank=r)
        engine.set_data(X, Y)
        ed, k, _ = engine.elbo(p, jitter=1e-6, include_kl=(r == 0))
        tot += ed
        kl = k if r == 0 else kl
    assert abs(res['elbo_data'] - tot) <= 1e-11 * abs(tot), (res['elbo_data'], tot)
    assert abs(res['kl'] - kl) <= 1e-12 * abs(kl)
    assert abs(res['elbo'] - (tot - kl)) <= 1e-11 * abs(tot - kl)


def test_bench_self_launches_two_ranks_strong(engine):
    sys.path.insert(0, ROOT)
    import bench
    res = _run(['--scaling', 'strong'])
    assert res['n_ranks_seen'] == 2 and res['scaling'] == 'strong' and res['config']['rows_total'] == 65536
    X, Y, p = bench.synth(65536, 256, 3, rank=0)
    engine.set_data(X, Y)
    ed, kl, _ = engine.elbo(p, jitter=1e-6)
    assert abs(res['elbo_data'] - ed) <= 1e-11 * abs(ed) and abs(res['kl'] - kl) <= 1e-12 * abs(kl)


def test_rccl_single_rank_exchange():
    """The RCCL leg of the N > 1 path on the one-GPU box: process group 'nccl' with one rank, the packed vector through ShardedELBO's
    pinned staging -> all_reduce on the GPU -> pinned -> unpack, barrier and MAX reduce (tools/nccl_selftest.py, own process)."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    env['MASTER_ADDR'] = '127.0.0.1'
    env['MASTER_PORT'] = '29671'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'nccl_selftest.py')], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'single-rank exchange ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
