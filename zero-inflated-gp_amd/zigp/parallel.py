"""Data-parallel ELBO over row shards: one process per GPU, one all-reduce per step.

The ELBO data term is a plain sum over points (tf.reduce_sum(var_exp), onoffgpf/OnOffSVGP.py:122;
scripts/onoff.py:307), so rank r evaluates rows [lo_r, hi_r) of its resident shard with the replicated
O(M^2) state and the packed vector [elbo_data, kl, d/d(params)...] (~82 KB at M=1024, D=3) is summed
with ONE all-reduce: by default through torch.distributed on the packed vector (backend "nccl" = RCCL over xGMI on a device
tensor; "gloo" in the CPU rehearsals), optionally inside libzigp.so (ncclAllReduce on the packed DEVICE vector, zigp_comm_init;
torch.distributed then only carries the 128-byte communicator id) -- see _Sharded.  KL and its gradient are added on rank 0 only.  The reference
has no distributed code; this is new.
"""
import numpy as np

GRAD_KEYS = ('Zf', 'Zg', 'u_fm', 'u_gm', 'u_fs_sqrt', 'u_gs_sqrt', 'ell_f', 'ell_g', 'var_f', 'var_g', 'noise')


def shard_bounds(n_rows, world_size, rank):
    """Contiguous row blocks, sizes differing by at most one (ragged N is fine)."""
    base, rem = divmod(int(n_rows), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


OPTIONAL_KEYS = ('mean_a', 'mean_b')   # present when a mean function is set (zigp_set_mean_function)


def pack(elbo_data, kl, grads):
    parts = [np.array([elbo_data, kl], dtype=np.float64)]
    shapes = []
    for k in GRAD_KEYS + tuple(k for k in OPTIONAL_KEYS if k in grads):
        a = np.asarray(grads[k], dtype=np.float64)
        shapes.append((k, a.shape))
        parts.append(a.reshape(-1))
    return np.concatenate(parts), shapes


def unpack(vec, shapes):
    elbo_data, kl = float(vec[0]), float(vec[1])
    g, o = {}, 2
    for k, sh in shapes:
        n = int(np.prod(sh)) if len(sh) else 1
        g[k] = vec[o:o + n].reshape(sh) if len(sh) else float(vec[o])
        o += n
    return elbo_data, kl, g


def _flatten_kron(g):
    """Kronecker gradient dict (lists per factor, scalars) -> (vector, spec) and back; used by the host-side exchange only"""
    parts, spec = [], []
    for k in sorted(g):
        v = g[k]
        items = v if isinstance(v, (list, tuple)) else [v]
        shapes = []
        for it in items:
            a = np.asarray(it, dtype=np.float64)
            shapes.append(a.shape)
            parts.append(a.reshape(-1))
        spec.append((k, isinstance(v, (list, tuple)), shapes))
    return (np.concatenate(parts) if parts else np.zeros(0)), spec


def _unflatten_kron(vec, spec):
    g, o = {}, 0
    for k, is_list, shapes in spec:
        items = []
        for sh in shapes:
            n = int(np.prod(sh)) if len(sh) else 1
            items.append(vec[o:o + n].reshape(sh) if len(sh) else float(vec[o]))
            o += n
        g[k] = items if is_list else items[0]
    return g


class _Sharded:
    """Common part of the data-parallel wrappers: who am I, and where does the exchange run.

    library_comm=False (the DEFAULT): the packed host vector goes through torch.distributed (backend 'nccl' = RCCL on a device tensor,
    'gloo' in the CPU rehearsals) -- the path that has run with more than one rank.
    library_comm=True (opt-in; also env ZIGP_LIBRARY_COMM=1): the exchange is ONE ncclAllReduce inside libzigp.so, on the device, on
    the engine's stream (zigp_comm_init; the 128-byte id is broadcast over `dist`) -- engine calls then return the sums.  It saves the
    host round trip per step, but until a run with two or more GPUs has been recorded it has only ever summed over ONE rank,
    so it is not the default; bench.py --gpus N checks it against the torch.distributed sums after its timed region and reports the
    result.  Setting it up is itself a sequence of agreements, so that no rank is left waiting inside a collective its peers never
    enter: (1) all ranks agree (MIN) that RCCL is loadable BEFORE anyone calls comm_init; (2) comm_init gives up after the engine's
    timeout (zigp_comm_set_timeout) if a peer does not arrive; (3) all ranks agree (MIN) on the outcome of comm_init and of a
    self-check sum.  Any 'no' leaves every rank on the torch.distributed exchange."""

    def __init__(self, engine, dist=None, device=None, library_comm=None):
        import os
        self.engine, self.dist, self.device = engine, dist, device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self._buf = None
        self._host = None
        if library_comm is None:
            library_comm = os.environ.get('ZIGP_LIBRARY_COMM', '0') not in ('', '0') and dist is not None and hasattr(engine, 'comm_init')
        self.library_comm = bool(library_comm) and dist is not None
        self._owns_comm = False
        if self.library_comm:
            info = engine.comm_info()
            if info['nranks'] == 0:
                if not self._agree(1.0 if (not hasattr(engine, 'comm_available') or engine.comm_available()) else 0.0):
                    self.library_comm = False          # some rank cannot bind RCCL: nobody enters the collective comm_init
                    return
                ok = 1.0
                try:
                    obj = [engine.comm_unique_id() if self.rank == 0 else None]
                except Exception:
                    obj, ok = [None], 0.0
                dist.broadcast_object_list(obj, src=0)
                if obj[0] is None:
                    ok = 0.0
                else:
                    try:
                        engine.comm_init(self.rank, self.world, obj[0])     # gives up after the engine's timeout if a peer never joins
                        self._owns_comm = True
                        chk = engine.comm_allreduce([1.0, float(self.rank)])
                        if chk[0] != self.world or chk[1] != self.world * (self.world - 1) / 2.0:
                            ok = 0.0
                    except Exception:
                        ok = 0.0
                if not self._agree(ok):
                    if self._owns_comm:
                        try:
                            engine.comm_destroy()
                        except Exception:
                            pass
                    self._owns_comm = False
                    self.library_comm = False
            elif (info['rank'], info['nranks']) != (self.rank, self.world):       # another wrapper of this engine set it up
                raise ValueError('engine already has a communicator for rank %d of %d' % (info['rank'], info['nranks']))
        elif dist is not None and hasattr(engine, 'comm_info'):
            # An engine that OWNS a communicator (another wrapper formed it) sums every result block over the ranks inside the library,
            # whatever this wrapper was asked for: reducing its results once more on the host would return world x the sums, silently.
            info = engine.comm_info()
            if info['nranks'] > 0:
                if (info['rank'], info['nranks']) != (self.rank, self.world):
                    raise ValueError('engine already has a communicator for rank %d of %d (this wrapper: rank %d of %d)'
                                     % (info['rank'], info['nranks'], self.rank, self.world))
                self.library_comm = True           # results arrive summed; never host-reduce them again

    def _agree(self, ok):
        """True when every rank says ok (MIN over ranks through torch.distributed)"""
        import torch
        dist = self.dist
        flag = torch.tensor([ok], dtype=torch.float64, device=self.device or ('cuda' if dist.get_backend() == 'nccl' else 'cpu'))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return float(flag.item()) == 1.0

    def close(self):
        if self._owns_comm:
            self.engine.comm_destroy()
        self.library_comm = self._owns_comm = False

    def _allreduce_host(self, vec):
        """sum a host float64 vector over ranks through torch.distributed (pinned staging when the exchange runs on the GPU)"""
        import torch
        if self._buf is None or self._buf.numel() != vec.size:
            self._buf = torch.empty(vec.size, dtype=torch.float64, device=self.device or 'cpu')
            self._host = torch.empty(vec.size, dtype=torch.float64, pin_memory=self._buf.is_cuda)
        self._host.numpy()[:] = vec
        if self._buf.is_cuda:
            self._buf.copy_(self._host, non_blocking=True)
            self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM)
            self._host.copy_(self._buf)           # synchronises with the all-reduce on the current stream
            return self._host.numpy().copy()
        self._buf.copy_(self._host)
        self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM)
        return self._buf.numpy().copy()


class ShardedELBO(_Sharded):
    """engine: anything with .elbo(p, jitter=, scale=, g_offset=, rows=, include_kl=, need_grad=True)
    evaluating rows of ITS OWN resident shard; dist: torch.distributed (initialised) or None for 1 process."""

    def elbo(self, p, jitter=1e-6, scale=1.0, g_offset=0.0, rows=None):
        # a rank whose shard is empty passes rows=(k, k): the engine then contributes zeros (and the KL on rank 0) but still takes part in
        # the exchange -- every rank makes the same calls
        ed, kl, g = self.engine.elbo(p, jitter=jitter, scale=scale, g_offset=g_offset, rows=rows,
                                     include_kl=(self.rank == 0), need_grad=True)
        if self.dist is None or self.library_comm:    # library_comm: already summed over ranks on the device (zigp_comm_init)
            return ed, kl, g
        vec, shapes = pack(ed, kl, g)
        return unpack(self._allreduce_host(vec), shapes)


class ShardedKronELBO(_Sharded):
    """The same for the Kronecker step (cfg5): rank r evaluates rows (lo, hi) of its own resident shard, or its own host minibatch,
    with the minibatch scale of the WHOLE job (scripts/onoff.py:311: num_data / num_minibatch over all ranks' rows)."""

    def kron_elbo(self, p, X=None, Y=None, jitter=1e-5, scale=1.0, g_offset=0.0, rows=None, f_mu=None):
        ed, kl, g = self.engine.kron_elbo(p, X, Y, jitter=jitter, scale=scale, g_offset=g_offset, rows=rows, f_mu=f_mu,
                                          include_kl=(self.rank == 0), need_grad=True)
        if self.dist is None or self.library_comm:
            return ed, kl, g
        gv, spec = _flatten_kron(g)
        out = self._allreduce_host(np.concatenate([np.array([ed, kl]), gv]))
        return float(out[0]), float(out[1]), _unflatten_kron(out[2:], spec)


class ShardedKronFit(_Sharded):
    """Data-parallel form of the reference's training loop (scripts/onoff.py:375-381: minibatch gradient of -ELBO, Log1pe chain, one Adam
    per learning rate): every rank holds the same parameters and Adam state, steps on a minibatch of ITS OWN resident shard, and the
    step's result block is summed over the ranks before the update -- so the parameters stay identical without ever being exchanged.

    With the library communicator (library_comm=True and it formed: zigp_comm_init) the whole loop runs on the device
    (onofftf.model.KronDeviceFit -> zigp_kron_fit_steps; the all-reduce of each step's result block is an ncclAllReduce on the engine's
    stream in front of k_fit_update, one host synchronisation per call).  Otherwise (gloo rehearsals, two ranks on one GPU, RCCL not
    loadable) the same iterations run on the host: ShardedKronELBO's all-reduced gradient + zigp.optim.AdamGroups, one engine call per
    iteration.  Both give every rank the single-process result of the same batches (sum order aside).
    scale is the minibatch scale of the WHOLE job, num_data / (rows of all ranks' batches) (scripts/onoff.py:311)."""

    def __init__(self, engine, pset, dist=None, device=None, library_comm=None, beta1=0.9, beta2=0.999, eps=1e-8):
        super().__init__(engine, dist, device, library_comm)
        from onofftf.model import KronDeviceFit
        self.pset = pset
        self.on_device = dist is None or self.library_comm
        if self.on_device:
            self.fit = KronDeviceFit(engine, pset, beta1=beta1, beta2=beta2, eps=eps)
        else:
            from .optim import AdamGroups
            self.adam = AdamGroups(pset, beta1=beta1, beta2=beta2, eps=eps)
            self._elbo = ShardedKronELBO(engine, dist, device=device, library_comm=False)

    @property
    def t(self):
        return self.fit.t if self.on_device else self.adam.t

    def steps(self, row_begin, batch, jitter, scale):
        """len(row_begin) iterations; row_begin[i] = first row (of THIS rank's resident shard) of its batch i.  Returns the history
        (elbo_data[n], kl[n]) summed over ranks, each at the parameters before that step's update; the ParamSet holds the new values."""
        if self.on_device:
            return self.fit.steps(row_begin, batch, jitter, scale, include_kl=(self.rank == 0))
        from onofftf.model import engine_params, named_grads
        ed_h, kl_h = np.zeros(len(row_begin)), np.zeros(len(row_begin))
        for i, rb in enumerate(row_begin):
            ed, kl, g = self._elbo.kron_elbo(engine_params(self.pset), jitter=jitter, scale=scale, rows=(int(rb), int(rb) + int(batch)))
            ed_h[i], kl_h[i] = ed, kl
            self.adam.step(named_grads(g))
        return ed_h, kl_h
