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
    engine.set_chunk(32768)          # as bench.py's library rule at M = 256 would not: pin it for both sides
    for r in range(2):
        X, Y, p = bench.weak_shard(65536, 256, 3, r, 2)
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


def test_rccl_single_rank_exchange_through_the_library():
    """The RCCL leg of the N > 1 path on the one-GPU box (tools/nccl_selftest.py, own process): zigp_comm_unique_id / zigp_comm_init with
    one rank, then every kind of step (dense, Kronecker fused / larger grid / panels, a head) through ncclAllReduce on the packed DEVICE
    vector -- bit-identical to the same call without a communicator; a Cholesky failure surfaces after the exchange."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    env['MASTER_ADDR'] = '127.0.0.1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'nccl_selftest.py')], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'single-rank exchange ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_cfg4_per_rank_workload_two_ranks_through_the_launcher(engine):
    """BASELINE cfg4 is 8 ranks x (1e6 rows, M = 1024); this is its PER-RANK workload through the same entry point with 2 ranks (gloo,
    both on the box's one GPU): `bench.py --gpus 2 --rows 1000000 --M 1024`, one step.  The 2-rank ELBO must be the sum of the two
    1-rank shard ELBOs (KL once), computed here with the engine directly on the same seeded shards."""
    sys.path.insert(0, ROOT)
    import bench
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--rows', '1000000', '--M', '1024', '--steps', '1',
           '--warmup', '0', '--no-cpu-baseline', '--no-other-configs', '--profile-steps', '0']
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert res['n_gpus'] == 2 and res['n_ranks_seen'] == 2 and res['config']['rows_total'] == 2000000 and res['config']['M'] == 1024
    assert res['config']['chunk_rows'] == 32768 and res['scaling'] == 'weak'
    tot, kl = 0.0, None
    engine.set_chunk(32768)          # the library's own rule at M = 1024 (other tests leave the shared engine on small chunks)
    for rk in range(2):
        X, Y, p = bench.weak_shard(1000000, 1024, 3, rk, 2)
        engine.set_data(X, Y)
        ed, k, _ = engine.elbo(p, jitter=1e-6, include_kl=(rk == 0), need_grad=False)
        tot += ed
        kl = k if rk == 0 else kl
    print('cfg4 per-rank workload x 2 ranks: elbo_data %.12e (sum of shards %.12e), kl %.10e, %.1f ms/step with two ranks on one GPU'
          % (res['elbo_data'], tot, res['kl'], res['ms_per_step']))
    assert abs(res['elbo_data'] - tot) <= 1e-11 * abs(tot), (res['elbo_data'], tot)
    assert abs(res['kl'] - kl) <= 1e-12 * abs(kl)


def test_comm_init_gives_up_when_a_peer_never_joins():
    """zigp_comm_init is collective; a rank whose peer never arrives must get ZIGP_ECOMM after the timeout (zigp_comm_set_timeout)
    instead of waiting forever.  Own process (a helper thread stays behind inside RCCL's bootstrap; the contract after a timeout is
    'fall back or exit'): rank 0 of a 2-rank communicator nobody else joins, timeout 4 s."""
    code = (
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import zigp\n"
        "eng = zigp.DenseEngine(0)\n"
        "assert eng.comm_available()\n"
        "eng.comm_set_timeout(4.0)\n"
        "uid = eng.comm_unique_id()\n"
        "t0 = time.time()\n"
        "try:\n"
        "    eng.comm_init(0, 2, uid)\n"
        "    print('NO ERROR'); sys.stdout.flush(); os._exit(3)\n"
        "except zigp.ZigpError as e:\n"
        "    dt = time.time() - t0\n"
        "    assert 3.0 < dt < 30.0, dt\n"
        "    assert 'did not return within' in str(e), str(e)\n"
        "    assert eng.comm_info()['nranks'] == 0\n"
        "    print('gave up after %%.1f s: %%s' %% (dt, e)); sys.stdout.flush()\n"
        "os._exit(0)\n" % os.path.join(ROOT, 'zero-inflated-gp_amd'))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0 and 'gave up after' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_force_dist_runs_the_multi_rank_code_path_with_one_rank():
    """`bench.py --gpus 1 --force-dist` = the code path of the 8-GPU run (torch.distributed process group on RCCL, the per-step all-reduce
    of the packed vector, barriers, MAX of the times) with ONE rank, and after the timed region the library-exchange check: communicator
    through zigp_comm_init, one step through ncclAllReduce inside libzigp.so, compared with the torch.distributed sums."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--rows', '65536', '--M', '256', '--steps', '2',
           '--warmup', '1', '--no-cpu-baseline', '--no-other-configs', '--profile-steps', '0']
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert res['n_gpus'] == 1 and res['backend'] == 'rccl(nccl)' and 'torch.distributed' in res['exchange']
    chk = res['library_exchange_check']
    print('library exchange check at one rank:', chk)
    assert chk['status'] == 'ok' and chk['packed_vector_max_rel_diff'] == 0.0 and chk['allreduce_calls_in_library'] >= 2
    clk = res['sustained_clock_mhz']['timed_region']
    print('sustained clock of the timed region: %s MHz' % clk)
    assert clk is None or 500 < clk < 3000


def test_two_rank_rccl_exchange_matches_the_sum_of_the_shards(engine):
    """Needs TWO GPUs (skipped on the one-GPU boxes this suite normally runs on): `bench.py --gpus 2` on RCCL -- the timed region
    on the torch.distributed exchange, then the library exchange (ncclAllReduce inside libzigp.so with two REAL ranks) checked against it --
    and the 2-rank ELBO against the sum of the two shard ELBOs computed by one engine."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs: RCCL refuses two ranks on one device')
    sys.path.insert(0, ROOT)
    import bench
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'nccl', '--rows', '65536', '--M', '256', '--steps', '2',
           '--warmup', '1', '--no-cpu-baseline', '--no-other-configs', '--profile-steps', '0']
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    print('2-rank RCCL run:', {k: res[k] for k in ('value', 'ms_per_step', 'exchange', 'library_exchange_check')})
    assert res['n_ranks_seen'] == 2 and res['library_exchange_check']['status'] == 'ok'
    tot, kl = 0.0, None
    engine.set_chunk(32768)
    for rk in range(2):
        X, Y, p = bench.weak_shard(65536, 256, 3, rk, 2)
        engine.set_data(X, Y)
        ed, k, _ = engine.elbo(p, jitter=1e-6, include_kl=(rk == 0))
        tot += ed
        kl = k if rk == 0 else kl
    assert abs(res['elbo_data'] - tot) <= 1e-11 * abs(tot) and abs(res['kl'] - kl) <= 1e-12 * abs(kl)
