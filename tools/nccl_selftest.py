"""Single-rank RCCL rehearsal of the N > 1 exchange on a one-GPU box, through the library's own communicator:
zigp_comm_unique_id -> (torch.distributed 'nccl' broadcast of the id, world_size 1) -> zigp_comm_init -> every step's packed device
vector goes through ncclAllReduce(sum, f64) on the engine's stream.  The sum over ONE rank is the identity, so every result must be
bit-identical to the same call without a communicator: dense step, value-only step, Kronecker step (fused 32 x 32 and 10 x 100, and
the panel path), a single-latent head.  Plus the barrier / MAX reduce bench.py brackets its timed region with."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.distributed as dist
import bench, zigp
from zigp.parallel import ShardedELBO, ShardedKronELBO, ShardedKronFit
from onofftf.model import KronDeviceFit
from test_gpu_kron import make_kron_problem
from test_cpu_host import _kron_pset

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', str(bench.free_port()))
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))


def same(a, b):
    if isinstance(a, dict):
        return set(a) == set(b) and all(same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    return np.array_equal(np.asarray(a), np.asarray(b))


X, Y, p = bench.synth(50000, 256, 3)
Xk, Yk, pk = make_kron_problem(3000, 32, 32, seed=5)
Xl, Yl, pl = make_kron_problem(1500, 10, 100, seed=6)
Xp, Yp, pp = make_kron_problem(900, 40, 9, seed=7)            # beyond the fused kernels: panel path (host-side exchange)
ph = {k: pk[k] for k in ('Zf', 'ell_f', 'var_f', 'u_fm', 'u_fs_sqrt', 'noise')}
eng = zigp.DenseEngine(0); eng.set_data(X, Y)
ref = dict(dense=eng.elbo(p), dense_value=eng.elbo(p, need_grad=False), dense_nokl=eng.elbo(p, include_kl=False),
           kron=eng.kron_elbo(pk, Xk, Yk, scale=3.0, f_mu=0.25), kron_value=eng.kron_elbo(pk, Xk, Yk, need_grad=False),
           kron_nokl=eng.kron_elbo(pk, Xk, Yk, include_kl=False), kron_large=eng.kron_elbo(pl, Xl, Yl),
           kron_panel=eng.kron_elbo(pp, Xp, Yp), head=eng.kron_head_elbo(ph, Xk, (Yk > 0).astype(float), 'bernoulli', f_mu=0.1))
# the device fit loop (zigp_kron_fit_steps: the all-reduce of each step's result block sits in front of k_fit_update): first WITHOUT a communicator
FIT_ROWS = [0, 700, 1400, 300, 2000]
fit_ref = {}
for tag, (Xq, Yq, pq) in (('fit_32x32', (Xk, Yk, pk)), ('fit_10x100', (Xl, Yl, pl))):
    eng.set_data(Xq, Yq)
    ps = _kron_pset(pq)
    f = KronDeviceFit(eng, ps)
    rows = [r % (Xq.shape[0] - 500) for r in FIT_ROWS]
    ed_, kl_ = f.steps(rows, 500, 1e-5, Xq.shape[0] / 500.0)
    fit_ref[tag] = (ed_, kl_, f.x.copy(), f.m.copy(), f.v.copy())
eng.set_data(X, Y)
assert eng.comm_info()['nranks'] == 0
sh = ShardedELBO(eng, dist, device='cuda:0', library_comm=True)      # the opt-in library communicator
shk = ShardedKronELBO(eng, dist, device='cuda:0', library_comm=True)        # same engine: shares the communicator
assert sh.library_comm and shk.library_comm and eng.comm_info() == dict(rank=0, nranks=1, allreduce_calls=1)   # 1: the wrapper's self-check sum
got = dict(dense=sh.elbo(p), dense_value=eng.elbo(p, need_grad=False), dense_nokl=eng.elbo(p, include_kl=False),
           kron=eng.kron_elbo(pk, Xk, Yk, scale=3.0, f_mu=0.25), kron_value=eng.kron_elbo(pk, Xk, Yk, need_grad=False),
           kron_nokl=eng.kron_elbo(pk, Xk, Yk, include_kl=False), kron_large=shk.kron_elbo(pl, Xl, Yl),
           kron_panel=eng.kron_elbo(pp, Xp, Yp), head=eng.kron_head_elbo(ph, Xk, (Yk > 0).astype(float), 'bernoulli', f_mu=0.1))
for k in ref:
    assert same(ref[k], got[k]), k
n = eng.comm_info()['allreduce_calls']
assert n == len(ref) + 1, n                                # one all-reduce per step (+ the self-check), no more
# ... and WITH the communicator, through the data-parallel wrapper (one rank: the sum is the identity, every bit must be the same)
for tag, (Xq, Yq, pq) in (('fit_32x32', (Xk, Yk, pk)), ('fit_10x100', (Xl, Yl, pl))):
    eng.set_data(Xq, Yq)
    ps = _kron_pset(pq)
    f = ShardedKronFit(eng, ps, dist, device='cuda:0', library_comm=True)
    assert f.on_device and f.library_comm
    rows = [r % (Xq.shape[0] - 500) for r in FIT_ROWS]
    before = eng.comm_info()['allreduce_calls']
    ed_, kl_ = f.steps(rows, 500, 1e-5, Xq.shape[0] / 500.0)
    assert eng.comm_info()['allreduce_calls'] - before == len(rows), (tag, eng.comm_info()['allreduce_calls'] - before)   # one all-reduce per iteration
    r_ = fit_ref[tag]
    assert np.array_equal(ed_, r_[0]) and np.array_equal(kl_, r_[1]) and np.array_equal(f.fit.x, r_[2]) and np.array_equal(f.fit.m, r_[3]) and np.array_equal(f.fit.v, r_[4]), tag
    assert f.t == len(rows)
eng.set_data(X, Y)
n = eng.comm_info()['allreduce_calls']
# a non-PD Kuu is reported after the exchange, and the communicator keeps working
bad = dict(p, Zf=p['Zf'].copy()); bad['Zf'][1] = bad['Zf'][0]
try:
    eng.elbo(bad, jitter=0.0); raise SystemExit('expected NotPositiveDefiniteError')
except zigp.NotPositiveDefiniteError:
    pass
assert same(eng.elbo(p), ref['dense'])
t = torch.tensor([1.25], dtype=torch.float64, device='cuda:0'); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
assert float(t.item()) == 1.25
assert np.array_equal(eng.comm_allreduce([3.0, -1.5]), [3.0, -1.5])      # zigp_comm_allreduce_host: a host vector through the same communicator
sh.close()
assert eng.comm_info()['nranks'] == 0 and same(eng.elbo(p), ref['dense'])
dist.destroy_process_group()
print('RCCL single-rank exchange ok: %d all-reduces through zigp_comm_init' % n)
