"""CPU suite, part 2: host logic and the C-ABI boundary (no compute calls without a GPU)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import make_problem, ROOT

GOLD = os.path.join(ROOT, 'tests', 'golden')


def test_library_loads_and_exports_every_declared_symbol():
    """Every function declared in include/zigp.h (the boundary) and include/zigp_diag.h (measurement hooks, test diagnostics) resolves in
    libzigp.so and is bound in zigp._lib."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    from zigp import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'zigp.h')).read()
    api = set(re.findall(r'\b(zigp_[A-Za-z0-9_]+)\s*\(', hdr))
    assert len(api) >= 20 and {'zigp_comm_init', 'zigp_comm_unique_id', 'zigp_kron_elbo_rows', 'zigp_elbo'} <= api
    assert not any(n.startswith(('zigp_test_', 'zigp_profile_')) for n in api)        # diagnostics live in zigp_diag.h
    names = sorted(api | set(re.findall(r'\b(zigp_[A-Za-z0-9_]+)\s*\(', open(os.path.join(ROOT, 'include', 'zigp_diag.h')).read())))
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, 'declared in zigp.h but not bound: ' + n
    assert set(_lib.SIGNATURES) == set(names)
    # struct layouts agree with the header (LP64): 4 int32 + 8 ptr + 3 double ; 8 ptr + 3 double
    assert ctypes.sizeof(_lib.zigp_params) == 16 + 8 * 8 + 3 * 8
    assert ctypes.sizeof(_lib.zigp_grads) == 8 * 8 + 3 * 8
    assert ctypes.sizeof(_lib.zigp_kron_params) == 32 + 8 * 8 + 4 * 8 + 4 * 8 + 8
    assert ctypes.sizeof(_lib.zigp_kron_grads) == 8 * 8 + 4 * 8 + 4 * 8 + 8


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import zigp
    with pytest.raises(zigp.ZigpError):
        zigp.DenseEngine(0)


def test_null_context_is_rejected():
    from zigp import _lib
    lib = _lib.load()
    assert lib.zigp_destroy(None) == _lib.ZIGP_EARG
    assert lib.zigp_set_chunk(None, 1024) == _lib.ZIGP_EARG
    assert lib.zigp_last_error(None) == b'null context'
    assert lib.zigp_get_chunk(None, 1024) == _lib.ZIGP_EARG
    assert lib.zigp_select_rows(None, None, 0) == _lib.ZIGP_EARG
    assert lib.zigp_set_pivot_rtol(None, 0.0) == _lib.ZIGP_EARG
    assert lib.zigp_comm_init(None, 0, 1, None) == _lib.ZIGP_EARG and lib.zigp_comm_destroy(None) == _lib.ZIGP_EARG
    assert lib.zigp_comm_unique_id(None) == _lib.ZIGP_EARG and lib.zigp_comm_info(None, None, None, None) == _lib.ZIGP_EARG


def test_log1pe_transform_roundtrip_and_gradient():
    from zigp.transforms import Log1pe
    t = Log1pe()
    y = np.array([1e-5, 0.01, 1.0, 20.0, 800.0])
    x = t.backward(y)
    assert np.allclose(t.forward(x), y, rtol=1e-12, atol=0)
    h = 1e-6
    fd = (t.forward(x + h) - t.forward(x - h)) / (2 * h)
    assert np.allclose(t.grad_free(x, np.ones_like(x)), fd, rtol=1e-6)
    assert np.isfinite(t.forward(np.array([-800.0, 800.0]))).all()


def test_paramset_flatten_roundtrip():
    from collections import OrderedDict
    from zigp.optim import P, ParamSet
    from zigp.transforms import positive
    ps = ParamSet(OrderedDict(a=P(np.arange(6.0).reshape(2, 3)), b=P([0.5, 2.0], positive), c=P(3.0, positive, fixed=True)))
    x = ps.get_free()
    assert x.size == 8
    ps.set_free(x + 0.0)
    assert np.allclose(ps.params['b'].value, [0.5, 2.0]) and ps.params['c'].value == 3.0
    g = ps.free_grad(dict(a=np.ones((2, 3)), b=np.ones(2), c=1.0))
    assert g.size == 8 and np.all(g[:6] == 1.0) and np.all(g[6:] < 1.0)


def test_adam_matches_tf_update_rule():
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v moments; x -= lr_t*m/(sqrt(v)+eps)."""
    from collections import OrderedDict
    from zigp.optim import P, ParamSet, AdamGroups
    ps = ParamSet(OrderedDict(w=P(np.array([1.0, -2.0]), learning_rate=0.1)))
    opt = AdamGroups(ps)
    x = np.array([1.0, -2.0]); m = np.zeros(2); v = np.zeros(2)
    for t in range(1, 4):
        g_elbo = -2 * ps.params['w'].value            # ELBO = -|w|^2  ->  cost gradient = 2w
        opt.step(dict(w=g_elbo))
        g = 2 * x
        m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
        x = x - 0.1 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-8)
        assert np.allclose(ps.params['w'].value, x, rtol=1e-13)


def test_shard_bounds_cover_ragged_rows():
    from zigp.parallel import shard_bounds
    for n, w in ((10, 3), (8, 8), (5, 8), (1000003, 8), (0, 2)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_pack_unpack_roundtrip():
    from zigp.parallel import pack, unpack, GRAD_KEYS
    rs = np.random.RandomState(0)
    g = dict(Zf=rs.randn(5, 3), Zg=rs.randn(4, 3), u_fm=rs.randn(5), u_gm=rs.randn(4), u_fs_sqrt=rs.randn(5), u_gs_sqrt=rs.randn(4),
             ell_f=rs.randn(3), ell_g=rs.randn(3), var_f=1.5, var_g=-2.5, noise=0.25)
    vec, shapes = pack(3.0, 4.0, g)
    ed, kl, g2 = unpack(vec, shapes)
    assert (ed, kl) == (3.0, 4.0)
    for k in GRAD_KEYS:
        assert np.array_equal(np.asarray(g[k]), np.asarray(g2[k]))
    # optional mean-function gradients ride along when present (zigp_set_mean_function)
    gm = dict(g, mean_a=rs.randn(3), mean_b=0.75)
    vec, shapes = pack(1.0, 2.0, gm)
    assert vec.size == 2 + sum(np.asarray(v).size for v in gm.values())
    _, _, g3 = unpack(vec, shapes)
    assert np.array_equal(g3['mean_a'], gm['mean_a']) and g3['mean_b'] == 0.75 and set(g3) == set(gm)


class _OracleShardEngine:
    """Test double with the DenseEngine.elbo signature, computing on its own shard with the CPU oracle."""

    def __init__(self, X, Y):
        self.X, self.Y = X, Y

    def elbo(self, p, jitter=1e-6, scale=1.0, g_offset=0.0, rows=None, include_kl=True, need_grad=True):
        import zigp_oracle_torch as ot
        lo, hi = (0, self.X.shape[0]) if rows is None else rows
        if hi == lo:       # an empty shard: as the library evaluates it -- one row at the origin with scale 0 (zero data term, KL per include_kl)
            e, d, kl, g = ot.elbo_and_grad(np.zeros((1, np.asarray(p['Zf']).shape[1])), np.zeros((1, 1)), p, jitter, scale=0.0, g_offset=g_offset,
                                           include_kl=include_kl, need_grad=True)
            scale = 0.0
        else:
            e, d, kl, g = ot.elbo_and_grad(self.X[lo:hi], self.Y[lo:hi], p, jitter, scale=scale, g_offset=g_offset,
                                           include_kl=include_kl, need_grad=True)
        g = {k: (np.asarray(v).reshape(-1) if k.startswith('u_') else v) for k, v in g.items()}
        return d * scale, kl, g


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from zigp.parallel import ShardedELBO, shard_bounds
    from conftest import make_problem as mp_
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    X, Y, p = mp_(301, 10, 2, seed=21, ell=0.5)          # ragged: 301 rows over 2 ranks
    lo, hi = shard_bounds(X.shape[0], world, rank)
    sh = ShardedELBO(_OracleShardEngine(X[lo:hi], Y[lo:hi]), dist)
    ed, kl, g = sh.elbo(p, jitter=1e-6, scale=1.0)
    if rank == 0:
        q.put((ed, kl, {k: np.asarray(v) for k, v in g.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _OracleKronShardEngine:
    """Test double with the DenseEngine.kron_elbo signature on its own shard (CPU oracle, literal dense order)."""

    def __init__(self, X, Y):
        self.X, self.Y = X, Y

    def kron_elbo(self, p, X=None, Y=None, jitter=1e-5, scale=1.0, g_offset=0.0, include_kl=True, need_grad=True, rows=None, f_mu=None):
        import zigp_oracle_torch as ot
        lo, hi = rows
        if hi == lo:       # empty shard: one row at the origin with scale 0 (csrc/zigp_kron.hip kron_no_rows)
            e, d, kl, g = ot.kron_elbo_and_grad(np.zeros((1, self.X.shape[1])), np.zeros((1, 1)), p, jitter, scale=0.0, g_offset=g_offset, include_kl=include_kl)
            return 0.0, kl, g
        e, d, kl, g = ot.kron_elbo_and_grad(self.X[lo:hi], self.Y[lo:hi], p, jitter, scale=scale, g_offset=g_offset, include_kl=include_kl)
        return d * scale, kl, g


def _gloo_kron_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from zigp.parallel import ShardedKronELBO, shard_bounds
    from test_gpu_kron import make_kron_problem
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    X, Y, p = make_kron_problem(203, 5, 4, seed=8, M0g=4, M1g=6)
    lo, hi = shard_bounds(X.shape[0], world, rank)
    sh = ShardedKronELBO(_OracleKronShardEngine(X[lo:hi], Y[lo:hi]), dist)
    assert not sh.library_comm                     # gloo: the packed host vector goes through torch.distributed
    ed, kl, g = sh.kron_elbo(p, rows=(0, hi - lo), jitter=1e-5, scale=2.0)
    if rank == 1:                                  # every rank holds the sums
        q.put((ed, kl, g))
    dist.barrier()
    dist.destroy_process_group()


class _FlakyCommEngine(_OracleShardEngine):
    """engine double whose library communicator cannot be formed on ONE rank (what a broken RCCL set-up on a node would look like)"""

    def __init__(self, X, Y, rank, mode):
        super().__init__(X, Y)
        self.rank, self.mode, self.destroyed, self.inited = rank, mode, 0, 0

    def comm_info(self):
        return dict(rank=0, nranks=0, allreduce_calls=0)

    def comm_unique_id(self):
        return b'\1' * 128

    def comm_init(self, rank, nranks, uid):
        self.inited += 1
        if self.mode == 'init' and rank == 1:
            raise RuntimeError('RCCL error in ncclCommInitRank: unhandled system error')

    def comm_allreduce(self, vec):
        if self.mode == 'sum' and self.rank == 0:
            return np.array([1.0, 0.0])          # a sum that does not add up on one rank
        return np.array([2.0, 1.0])

    def comm_destroy(self):
        self.destroyed += 1


def _gloo_fallback_worker(rank, world, port, mode, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from zigp.parallel import ShardedELBO, shard_bounds
    from conftest import make_problem as mp_
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    X, Y, p = mp_(120, 6, 2, seed=3, ell=0.5)
    lo, hi = shard_bounds(X.shape[0], world, rank)
    eng = _FlakyCommEngine(X[lo:hi], Y[lo:hi], rank, mode)
    sh = ShardedELBO(eng, dist, library_comm=True)          # ask for the library exchange; it cannot be set up on every rank
    ed, kl, g = sh.elbo(p, jitter=1e-6)
    q.put((rank, sh.library_comm, eng.inited, eng.destroyed, ed, kl))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['init', 'sum'])
def test_library_exchange_failure_on_one_rank_makes_every_rank_fall_back(mode):
    """If the library communicator cannot be formed (or its self-check sum is wrong) on ANY rank, ALL ranks agree to use the
    torch.distributed exchange of the packed host vector instead, tear down what they had set up, and still return the right sums."""
    import torch.multiprocessing as mp
    import zigp_oracle_torch as ot
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_fallback_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    X, Y, p = make_problem(120, 6, 2, seed=3, ell=0.5)
    e1, d1, kl1, g1 = ot.elbo_and_grad(X, Y, p, 1e-6)
    for rank, lib_comm, inited, destroyed, ed, kl in res:
        assert lib_comm is False and inited == 1
        assert destroyed == (0 if (mode == 'init' and rank == 1) else 1)      # whoever got a communicator gave it back
        assert abs(ed - d1) < 1e-10 * abs(d1) and abs(kl - kl1) < 1e-12 * abs(kl1)


def test_data_parallel_kronecker_gloo_world2_equals_single_process():
    """The Kronecker step (cfg5) sharded over rows: 2 gloo ranks, ragged shards, f and g on different grids; KL counted once."""
    import torch.multiprocessing as mp
    import zigp_oracle_torch as ot
    from test_gpu_kron import make_kron_problem
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_kron_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    ed, kl, g = q.get(timeout=180)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    X, Y, p = make_kron_problem(203, 5, 4, seed=8, M0g=4, M1g=6)
    e1, d1, kl1, g1 = ot.kron_elbo_and_grad(X, Y, p, 1e-5, scale=2.0)
    assert abs(ed - 2.0 * d1) < 1e-10 * abs(2.0 * d1) and abs(kl - kl1) < 1e-12 * abs(kl1)
    for k in g1:
        a, b = g[k], g1[k]
        for x, y in (zip(a, b) if isinstance(b, (list, tuple)) else ((a, b),)):
            x, y = np.asarray(x, dtype=float).reshape(-1), np.asarray(y, dtype=float).reshape(-1)
            assert np.max(np.abs(x - y)) <= 1e-9 * max(np.max(np.abs(y)), 1e-300), k


def _kron_pset(p, lr=1e-2):
    """a ParamSet in the layout of onofftf.model.init_params from a make_kron_problem parameter dict"""
    from collections import OrderedDict
    from onofftf.main import Param
    from zigp.optim import ParamSet
    from zigp.transforms import Log1pe, positive
    q = OrderedDict()
    for tag in ('f', 'g'):
        for i in range(2):
            q['%s_kern/lengthscale_%d' % (tag, i)] = Param(p['ell_' + tag][i], Log1pe(), name='lengthscale', learning_rate=lr)
            q['%s_kern/variance_%d' % (tag, i)] = Param(p['var_' + tag][i], Log1pe(), name='variance', learning_rate=lr)
            q['%s_ind/z_%d' % (tag, i)] = Param(p['Z' + tag][i].copy(), name='z', learning_rate=2 * lr)
        q['%s_ind/value' % tag] = Param(p['u_%sm' % tag].copy(), name='value', learning_rate=2 * lr)
        q['%s_ind/variance' % tag] = Param(p['u_%ss_sqrt' % tag].copy(), positive, name='variance', learning_rate=2 * lr)
    q['likelihood/variance'] = Param(p['noise'], Log1pe(), name='variance', learning_rate=lr)
    return ParamSet(q)


_FIT_ROWS = [0, 30, 60, 15]      # first row of each rank's minibatch inside its OWN shard, per iteration
_FIT_BATCH = 20


def _gloo_kron_fit_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from zigp.parallel import ShardedKronFit, shard_bounds
    from test_gpu_kron import make_kron_problem
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    X, Y, p = make_kron_problem(203, 5, 4, seed=8, M0g=4, M1g=6)
    lo, hi = shard_bounds(X.shape[0], world, rank)
    pset = _kron_pset(p)
    fit = ShardedKronFit(_OracleKronShardEngine(X[lo:hi], Y[lo:hi]), pset, dist)
    assert not fit.on_device and not fit.library_comm       # gloo: host loop on the all-reduced gradient
    ed, kl = fit.steps(_FIT_ROWS, _FIT_BATCH, 1e-5, 203.0 / (world * _FIT_BATCH))
    q.put((rank, fit.t, ed, kl, {k: v.value.copy() for k, v in pset.params.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_kron_fit_gloo_world2_equals_single_process():
    """ShardedKronFit (the data-parallel form of scripts/onoff.py:375-381) without a library communicator: 2 gloo ranks, each stepping on a
    minibatch of its own shard, the gradient all-reduced, Adam on every rank -- both ranks must end with the parameters a single process
    gets from the union of the two minibatches, and with each other's bit for bit."""
    import torch.multiprocessing as mp
    import zigp_oracle_torch as ot
    from zigp.optim import AdamGroups
    from zigp.parallel import shard_bounds
    from onofftf.model import engine_params, named_grads
    from test_gpu_kron import make_kron_problem
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_kron_fit_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    X, Y, p = make_kron_problem(203, 5, 4, seed=8, M0g=4, M1g=6)
    pset = _kron_pset(p)
    adam = AdamGroups(pset)
    scale = 203.0 / (2 * _FIT_BATCH)
    hist = []
    for rb in _FIT_ROWS:
        idx = np.concatenate([np.arange(shard_bounds(203, 2, r)[0] + rb, shard_bounds(203, 2, r)[0] + rb + _FIT_BATCH) for r in range(2)])
        e, d, kl, g = ot.kron_elbo_and_grad(X[idx], Y[idx], engine_params(pset), 1e-5, scale=scale)
        hist.append((d * scale, kl))
        adam.step(named_grads(g))
    for rank, t, ed, kl, vals in got:
        assert t == len(_FIT_ROWS)
        assert np.max(np.abs(ed - np.array([h[0] for h in hist])) / np.abs(ed)) < 1e-9 and np.max(np.abs(kl - np.array([h[1] for h in hist])) / np.abs(kl)) < 1e-9
        for k, v in pset.params.items():
            assert np.max(np.abs(vals[k] - v.value)) <= 1e-8 * max(np.max(np.abs(v.value)), 1e-300), (rank, k)
    for k in got[0][4]:
        assert np.array_equal(got[0][4][k], got[1][4][k]), k          # the ranks agree bit for bit: they applied the same all-reduced gradient


def test_kron_device_fit_counts_the_steps_applied_before_a_failure_and_notices_outside_changes():
    """(1) a Cholesky failure in step k > 0 of a zigp_kron_fit_steps call has applied k updates: KronDeviceFit.t advances by k and
    the exception carries the history of those steps; (2) a change made to the ParamSet behind the optimiser's back (load_checkpoint, an
    assignment) is taken up by the next call instead of being overwritten by the cached free state -- KronDeviceFit and AdamGroups."""
    import zigp
    from zigp.optim import AdamGroups
    from onofftf.model import KronDeviceFit, FIT_BLOCK_NAMES
    from test_gpu_kron import make_kron_problem
    X, Y, p = make_kron_problem(50, 3, 4, seed=2)

    class FakeEngine:
        fail_at = None
        seen_x = None

        def kron_fit_steps(self, shape, x, m, v, lr, positive, t0, row_begin, batch, **kw):
            self.seen_x = x.copy()
            n = len(row_begin)
            k = n if self.fail_at is None else self.fail_at
            x += 0.125 * k                       # k updates applied in place
            if self.fail_at is not None:
                e = zigp.NotPositiveDefiniteError('Cholesky failed in step %d' % k)
                e.steps_applied, e.elbo_data, e.kl = k, np.arange(k, dtype=float), np.zeros(k)
                raise e
            return np.zeros(n), np.zeros(n)

    eng = FakeEngine()
    pset = _kron_pset(p)
    fit = KronDeviceFit(eng, pset)
    fit.steps([0, 1, 2], 10, 1e-5, 1.0)
    assert fit.t == 3
    eng.fail_at = 2
    x_before = fit.x.copy()
    with pytest.raises(zigp.NotPositiveDefiniteError) as ei:
        fit.steps([0, 1, 2, 3, 4], 10, 1e-5, 1.0)
    assert fit.t == 5 and ei.value.steps_applied == 2 and ei.value.elbo_data.tolist() == [0.0, 1.0]
    assert np.allclose(fit.x, x_before + 0.25)
    assert np.allclose(pset.params['f_ind/value'].value.reshape(-1), fit.x[sum(fit.sizes[:2]):sum(fit.sizes[:3])])   # the ParamSet has the state after those 2 updates
    # (2) outside change
    eng.fail_at = None
    pset.params['f_ind/value'].value = np.full_like(pset.params['f_ind/value'].value, 7.0)
    fit.steps([0], 10, 1e-5, 1.0)
    o = sum(fit.sizes[:2])
    assert np.all(eng.seen_x[o:o + fit.sizes[2]] == 7.0)            # the call started from the new values
    adam = AdamGroups(pset)
    g0 = {k: np.zeros_like(q.value) for k, q in pset.params.items()}
    adam.step(g0)
    pset.params['likelihood/variance'].value = np.array([0.5])
    adam.step(g0)                                                    # zero gradient: the value must stay where it was PUT, not snap back
    assert abs(float(pset.params['likelihood/variance'].value[0]) - 0.5) < 1e-12
    adam.resync()
    assert adam.t == 2
    # (3) a parameter that has gone NaN is still the value these objects wrote: no resync from free(NaN) on every call, and an
    # explicit reset (load_checkpoint(..., fitter=...)) restarts the moments and the iteration count
    fit.x[:] = np.nan
    fit.sync_params()
    assert not fit._stale()
    adam.m['likelihood/variance'][:] = 3.0
    adam.resync(reset=True)
    assert adam.t == 0 and float(adam.m['likelihood/variance'][0]) == 0.0
    pset2 = _kron_pset(p)
    fit2 = KronDeviceFit(eng, pset2)
    fit2.steps([0, 1], 10, 1e-5, 1.0)
    fit2.m[:] = 1.0
    import tempfile
    from onofftf.model import save_checkpoint, load_checkpoint
    with tempfile.TemporaryDirectory() as td:
        save_checkpoint(_kron_pset(p), os.path.join(td, 'model'))
        load_checkpoint(pset2, os.path.join(td, 'model'), fitter=fit2)
    assert fit2.t == 0 and not fit2.m.any() and not fit2._stale()
    assert np.array_equal(fit2.x, KronDeviceFit(eng, _kron_pset(p)).x)


def test_wrapper_never_host_reduces_on_an_engine_that_owns_a_communicator():
    """libzigp all-reduces every result block whenever the context has a communicator (zigp_comm_init), whatever the Python
    wrapper was asked for.  A wrapper built with library_comm=False on such an engine must hand the (already summed) results through --
    reducing them again on the host would return world x the sums -- and refuse an engine whose communicator is for another rank."""
    from zigp.parallel import ShardedELBO, ShardedKronELBO

    class Dist:
        def __init__(self, rank, world): self.r, self.w = rank, world
        def get_rank(self): return self.r
        def get_world_size(self): return self.w
        def all_reduce(self, *a, **k): raise AssertionError('host all-reduce on an engine that sums in the library')
        def get_backend(self): return 'gloo'

    class Eng:
        def __init__(self, rank, n): self.info = dict(rank=rank, nranks=n, allreduce_calls=0)
        def comm_info(self): return self.info
        def elbo(self, p, **kw): return 1.5, 0.25, {k: np.ones(2) for k in ('Zf', 'Zg', 'u_fm', 'u_gm', 'u_fs_sqrt', 'u_gs_sqrt', 'ell_f', 'ell_g', 'var_f', 'var_g', 'noise')}
        def kron_elbo(self, p, X, Y, **kw): return 2.5, 0.5, {'noise': 1.0}

    sh = ShardedELBO(Eng(1, 2), Dist(1, 2), library_comm=False)
    assert sh.library_comm and not sh._owns_comm
    ed, kl, g = sh.elbo({})
    assert (ed, kl) == (1.5, 0.25) and np.array_equal(g['Zf'], np.ones(2))
    assert ShardedKronELBO(Eng(0, 2), Dist(0, 2), library_comm=False).kron_elbo({})[0] == 2.5
    with pytest.raises(ValueError):
        ShardedELBO(Eng(0, 2), Dist(1, 2), library_comm=False)
    with pytest.raises(ValueError):
        ShardedELBO(Eng(0, 4), Dist(0, 2), library_comm=False)
    sh0 = ShardedELBO(Eng(0, 0), Dist(0, 2), library_comm=False)      # no communicator: the torch.distributed exchange, as before
    assert not sh0.library_comm


def test_data_parallel_allreduce_gloo_world2_equals_single_process():
    """N>1 path on CPU: 2 ranks (gloo), row shards, one all-reduce; KL counted once (rank 0)."""
    import torch.multiprocessing as mp
    import zigp_oracle_torch as ot
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    ed, kl, g = q.get(timeout=180)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    X, Y, p = make_problem(301, 10, 2, seed=21, ell=0.5)
    e1, d1, kl1, g1 = ot.elbo_and_grad(X, Y, p, 1e-6)
    assert abs(ed - d1) < 1e-10 * abs(d1) and abs(kl - kl1) < 1e-12 * abs(kl1)
    for k in ot.PARAM_KEYS:
        a, b = np.asarray(g[k]).reshape(-1), np.asarray(g1[k]).reshape(-1)
        assert np.max(np.abs(a - b)) <= 1e-9 * max(np.max(np.abs(b)), 1e-300), k


W8_BOUNDS = [0, 40, 40, 95, 130, 131, 200, 200, 260]      # 8 ragged shards: ranks 1 and 6 hold NO rows, rank 4 a single one


def _gloo_w8_worker(rank, world, port, kind, q):
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from zigp.parallel import ShardedELBO, ShardedKronELBO
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    lo, hi = W8_BOUNDS[rank], W8_BOUNDS[rank + 1]
    if kind == 'dense':
        from conftest import make_problem as mp_
        X, Y, p = mp_(260, 7, 2, seed=5, ell=0.5)
        sh = ShardedELBO(_OracleShardEngine(X[lo:hi], Y[lo:hi]), dist)
        ed, kl, g = sh.elbo(p, jitter=1e-6, scale=1.0, rows=(0, hi - lo))
        g = {k: np.asarray(v) for k, v in g.items()}
    else:
        from test_gpu_kron import make_kron_problem
        X, Y, p = make_kron_problem(260, 4, 3, seed=9, M0g=3, M1g=5)
        sh = ShardedKronELBO(_OracleKronShardEngine(X[lo:hi], Y[lo:hi]), dist)
        ed, kl, g = sh.kron_elbo(p, rows=(0, hi - lo), jitter=1e-5, scale=1.5)
    assert not sh.library_comm                     # default: the torch.distributed exchange
    if rank in (0, 6):                             # every rank holds the sums: a rank WITH rows and one WITHOUT report
        q.put((rank, ed, kl, g))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['dense', 'kron'])
def test_data_parallel_gloo_world8_ragged_and_empty_shards_equal_single_process(kind):
    """cfg4's rank count on the CPU: 8 gloo ranks over ragged row shards, two of them EMPTY (they still make the same calls and take
    part in the exchange with a zero contribution), one with a single row; the KL is counted once; every rank -- also an empty one --
    ends with the single-process sums."""
    import torch.multiprocessing as mp
    import zigp_oracle_torch as ot
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_w8_worker, args=(r, 8, port, kind, q)) for r in range(8)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    if kind == 'dense':
        X, Y, p = make_problem(260, 7, 2, seed=5, ell=0.5)
        e1, d1, kl1, g1 = ot.elbo_and_grad(X, Y, p, 1e-6)
        keys, sc = ot.PARAM_KEYS, 1.0
    else:
        from test_gpu_kron import make_kron_problem
        X, Y, p = make_kron_problem(260, 4, 3, seed=9, M0g=3, M1g=5)
        e1, d1, kl1, g1 = ot.kron_elbo_and_grad(X, Y, p, 1e-5, scale=1.5)
        keys, sc = list(g1), 1.5
    for rank, ed, kl, g in res:
        assert abs(ed - sc * d1) < 1e-10 * abs(sc * d1) and abs(kl - kl1) < 1e-12 * abs(kl1), (rank, ed, sc * d1, kl, kl1)
        for k in keys:
            a, b = g[k], g1[k]
            for x, y in (zip(a, b) if isinstance(b, (list, tuple)) else ((a, b),)):
                x, y = np.asarray(x, dtype=float).reshape(-1), np.asarray(y, dtype=float).reshape(-1)
                assert np.max(np.abs(x - y)) <= 1e-9 * max(np.max(np.abs(y)), 1e-300), (rank, k)
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]          # the ranks agree bit for bit


class _LiteralDataSet:
    """checker: the branch structure of onofftf/main.py:98-133 written out (arrays re-gathered at every shuffle)"""

    def __init__(self, x, y, seed=121):
        np.random.seed(seed)
        self.n, self.x, self.y, self.ep, self.i = x.shape[0], x, y, 0, 0

    def next_batch(self, b, shuffle=True):
        start = self.i
        if self.ep == 0 and start == 0 and shuffle:
            perm = np.arange(self.n); np.random.shuffle(perm); self.x, self.y = self.x[perm], self.y[perm]
        if start + b > self.n:
            self.ep += 1
            xr, yr = self.x[start:], self.y[start:]
            if shuffle:
                perm = np.arange(self.n); np.random.shuffle(perm); self.x, self.y = self.x[perm], self.y[perm]
            self.i = b - (self.n - start)
            return np.concatenate((xr, self.x[:self.i])), np.concatenate((yr, self.y[:self.i]))
        self.i += b
        return self.x[start:self.i], self.y[start:self.i]


@pytest.mark.parametrize('n,b,shuffle', [(37, 5, True), (40, 8, True), (10, 10, True), (23, 7, False), (105, 100, True)])
def test_dataset_order_iterator_yields_the_reference_batch_sequence(n, b, shuffle):
    """DataSet keeps an index ORDER instead of re-gathered copies; its batches -- through next_batch, next_indices and the resident-epoch
    form next_span -- are those of the reference's iterator, wrap-around batches and epoch counter included."""
    from onofftf.main import DataSet
    X = np.arange(n, dtype=float)[:, None] * np.array([[1.0, -1.0]])
    Y = np.arange(n, dtype=float)[:, None] + 0.5
    ref = _LiteralDataSet(X, Y)
    seq = [ref.next_batch(b, shuffle) for _ in range(4 * n // b + 3)]
    ds = DataSet(X, Y)
    for xb, yb in seq:
        x2, y2 = ds.next_batch(b, shuffle)
        assert np.array_equal(x2, xb) and np.array_equal(y2, yb)
    assert ds.epochs_completed == ref.ep
    ds = DataSet(X, Y)
    for xb, yb in seq:
        idx = ds.next_indices(b, shuffle)
        assert np.array_equal(X[idx], xb)
    ds, uploads, last = DataSet(X, Y), 0, None
    for xb, yb in seq:
        gen, lo, hi, wrap = ds.next_span(b, shuffle)
        if wrap is None:
            if gen != last:
                resident, last, uploads = (ds.xtrain.copy(), ds.ytrain.copy()), gen, uploads + 1     # what onoff() sends to the GPU
            assert np.array_equal(resident[0][lo:hi], xb) and np.array_equal(resident[1][lo:hi], yb)
        else:
            assert np.array_equal(wrap[0], xb) and np.array_equal(wrap[1], yb)
    assert uploads <= ds.epochs_completed + 1        # one upload per epoch, not one per batch


def test_dataset_iterator_semantics():
    from onofftf.main import DataSet
    X, Y = np.arange(10)[:, None].astype(float), np.arange(10)[:, None].astype(float)
    ds = DataSet(X, Y)
    seen = []
    for _ in range(5):
        xb, yb = ds.next_batch(4)
        assert xb.shape == (4, 1) and np.array_equal(xb, yb)
        seen.append(xb.reshape(-1))
    first_epoch = np.concatenate(seen)[:10]
    assert sorted(first_epoch.tolist()) == list(range(10))       # one full permutation before the wrap-around
    assert ds.epochs_completed == 1
    ds2 = DataSet(X, Y)
    assert np.array_equal(ds2.next_batch(4)[0].reshape(-1), seen[0])    # seed 121 -> reproducible


# ---- data preparation of the precipitation experiments ------------------------------------
def test_preprocessing_matches_reference_golden():
    """onofftf.utils_pptr.preprocessing against outputs of the reference's own class (tests/golden/g3_utils_pptr.npz, made by
    oracle/make_golden.py::g3 importing /root/reference/onofftf/utils_pptr.py in place)."""
    import warnings
    from make_golden import pptr_like_table
    from onofftf.utils_pptr import preprocessing
    g = np.load(os.path.join(GOLD, 'g3_utils_pptr.npz'))
    for tag, filt, loc, tim in (('raw', False, False, False), ('filt', True, False, False), ('loc', False, True, False), ('all', True, True, True)):
        pp = preprocessing(pptr_like_table())
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            if filt:
                pp.filter_time(50, 320)
            if loc or tim:
                pp.scale(scale_loc=loc, scale_time=tim)
        md = pp.model_data
        for k in ('Xtrain', 'Xtest', 'Ytrain', 'Ytest'):
            assert np.array_equal(md[k], g['%s_%s' % (tag, k)]), (tag, k)
        assert np.array_equal(md['traindf'][['lat', 'lon', 'ndatehour']].values, g[tag + '_traindf'])
        assert np.array_equal(np.array(pp.shape), g[tag + '_shape'])
        var, ell = pp.kernel_params
        assert var == float(g[tag + '_kvar']) and np.array_equal(np.array(ell), g[tag + '_kell'])
        if loc or tim:
            for c, v in pp.scale_param.items():
                assert np.array_equal(np.array([v['min'], v['range']]), g['%s_sp_%s' % (tag, c)])


def test_create_cvsplits_protocol(tmp_path):
    """scripts/create_cvsplits.py: 5 folds of the pooled data, time / 1000, KFold(shuffle, random_state=1234), data.pickle per fold."""
    import pickle
    from scripts.create_cvsplits import create_cvsplits
    d = np.load(os.path.join(GOLD, 'pptr.npz'))
    small = {'Xtrain': d['Xtrain'][:400].copy(), 'Ytrain': d['Ytrain'][:400], 'Xtest': d['Xtest'][:100].copy(), 'Ytest': d['Ytest'][:100]}
    src = tmp_path / 'pptr.pickle'
    pickle.dump(small, open(src, 'wb'))
    dirs = create_cvsplits(str(src), str(tmp_path / 'cv'))
    assert [os.path.basename(x) for x in dirs] == ['1', '2', '3', '4', '5']
    pooledX = np.concatenate([small['Xtrain'], small['Xtest']])
    pooledX[:, 2] /= 1000
    seen = []
    for x in dirs:
        f = pickle.load(open(os.path.join(x, 'data.pickle'), 'rb'))
        assert f['Xtrain'].shape == (400, 3) and f['Xtest'].shape == (100, 3) and f['Ytest'].shape == (100, 1)
        seen.append(f['Xtest'])
    allx = np.concatenate(seen)
    assert sorted(map(tuple, allx)) == sorted(map(tuple, pooledX))      # every pooled row is a test row exactly once
    from sklearn.model_selection import KFold
    tr, te = next(iter(KFold(n_splits=5, random_state=1234, shuffle=True).split(pooledX)))
    assert np.array_equal(pickle.load(open(os.path.join(dirs[0], 'data.pickle'), 'rb'))['Xtest'], pooledX[te])


def test_zero_inflated_combination_host_only(tmp_path):
    """scripts/zero_inflated.py is pure post-processing: probability- and indicator-weighted regression mean, clipped metrics."""
    from scripts.zero_inflated import zero_inflated, rmse, mad
    rs = np.random.RandomState(0)
    Ytr, Yte = np.abs(rs.randn(50, 1)), np.abs(rs.randn(20, 1))
    clf = {'pred_train': {'pfmean': rs.rand(50, 1)}, 'pred_test': {'pfmean': rs.rand(20, 1)}}
    reg = {'pred_train': {'fmean': rs.randn(50, 1)}, 'pred_test': {'fmean': rs.randn(20, 1)}}
    r = zero_inflated(Ytr, Yte, clf, reg, str(tmp_path))
    assert np.array_equal(r['pred_test_zi_prob'], clf['pred_test']['pfmean'] * reg['pred_test']['fmean'])
    assert r['test_zi_indc_reg_rmse'] == rmse((clf['pred_test']['pfmean'] > 0.5) * reg['pred_test']['fmean'], Yte)
    assert r['train_zi_prob_reg_mae'] == mad(r['pred_train_zi_prob'], Ytr)
    assert rmse(np.array([-1.0]), np.array([0.0])) == 0.0                # predictions are clipped at 0 before scoring
    assert os.path.exists(os.path.join(str(tmp_path), 'results_zi.pickle'))


def test_classifier_scores_match_sklearn():
    from scripts.classifier import _scores
    from sklearn.metrics import accuracy_score, precision_score, recall_score, roc_auc_score
    rs = np.random.RandomState(1)
    y = rs.rand(500) > 0.7
    p = np.clip(0.3 * y + 0.5 * rs.rand(500), 0, 1)
    p[:20] = 0.5                                                           # ties
    acc, prec, rec, auc = _scores(p.reshape(-1, 1), y.reshape(-1, 1))
    assert abs(acc - accuracy_score(y, p > 0.5)) < 1e-15 and abs(prec - precision_score(y, p > 0.5)) < 1e-15
    assert abs(rec - recall_score(y, p > 0.5)) < 1e-15 and abs(auc - roc_auc_score(y, p)) < 1e-12


def test_triangular_tile_lists_are_partitions_with_and_without_the_lpt_tail():
    """The chunk loop's triangular products run host-built tile lists (paired units, both latents in one launch; since round 6 the last,
    partly filled wave re-dealt longest-first -- zigp_host.h tiles_trmm / trmm_tail_plan).  zigp_test_trmm_list rebuilds them on the host,
    as chunk_forward does, and checks that every (row block, column panel) tile occurs exactly once with the whole k range of its row
    block -- for the bench configurations and a sweep of ragged shapes; no GPU involved."""
    import ctypes as C
    from zigp import _lib
    lib = _lib.load()
    out = (C.c_int64 * 8)()
    shapes = [(512, 512, 100352), (1024, 1024, 125952), (1024, 1024, 32768), (1024, 1024, 17408), (300, 520, 38912), (2048, 1100, 27648),
              (9, 9, 1024), (520, 100, 2048), (1024, 1024, 131072), (640, 128, 65536)]
    rs = np.random.RandomState(3)
    shapes += [(int(rs.randint(1, 2300)), int(rs.randint(1, 2300)), 1024 * int(rs.randint(1, 129))) for _ in range(60)]
    tails = 0
    for Mf, Mg, Nc in shapes:
        for lower in (1, 0):
            res = {}
            for tail in (0, 1):
                assert lib.zigp_test_trmm_list(lower, Mf, Mg, Nc, tail, out) == 0, (Mf, Mg, Nc, lower, tail)
                res[tail] = list(out)
            plain, lpt = res[0], res[1]
            assert plain[4] == plain[5] == 0
            if lpt[4] or lpt[5]:                                   # a tail was planned: one wave of at most 512 workgroups replaces r + 512 units
                tails += 1
                assert lpt[7] == 1 and (lpt[0] + lpt[1]) % 512 == 0 and lpt[0] + lpt[1] < plain[0] + plain[1]
                nbm = max((Mf + 127) // 128, (Mg + 127) // 128)
                assert 0 < lpt[6] < 2 * (nbm + 1)                  # shorter than the two lockstep waves it replaces
            else:
                assert lpt[:4] == plain[:4]
    assert tails >= 8                                              # the sweep does exercise the tail (cfg2 and the 125 952-row shard among them)
    assert lib.zigp_test_trmm_list(1, 512, 512, 1000, 1, out) == _lib.ZIGP_EARG      # Nc must be a multiple of 128
