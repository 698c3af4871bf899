"""Data-parallel ELBO over row shards: one process per GPU, one all-reduce per step.

The ELBO data term is a plain sum over points (tf.reduce_sum(var_exp), onoffgpf/OnOffSVGP.py:122;
scripts/onoff.py:307), so rank r evaluates rows [lo_r, hi_r) of its resident shard with the replicated
O(M^2) state and the packed vector [elbo_data, kl, d/d(params)...] (~82 KB at M=1024, D=3) is summed
with ONE torch.distributed all_reduce (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  KL and its gradient are added on rank 0 only.  The reference has no distributed code; this is new.
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


class ShardedELBO:
    """engine: anything with .elbo(p, jitter=, scale=, g_offset=, rows=, include_kl=, need_grad=True)
    evaluating rows of ITS OWN resident shard; dist: torch.distributed (initialised) or None for 1 process."""

    def __init__(self, engine, dist=None, device=None):
        self.engine, self.dist, self.device = engine, dist, device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self._buf = None
        self._host = None

    def elbo(self, p, jitter=1e-6, scale=1.0, g_offset=0.0, rows=None):
        ed, kl, g = self.engine.elbo(p, jitter=jitter, scale=scale, g_offset=g_offset, rows=rows,
                                     include_kl=(self.rank == 0), need_grad=True)
        if self.world == 1:
            return ed, kl, g
        import torch
        vec, shapes = pack(ed, kl, g)
        if self._buf is None or self._buf.numel() != vec.size:
            self._buf = torch.empty(vec.size, dtype=torch.float64, device=self.device or 'cpu')
            # page-locked staging for the packed vector when the exchange runs on the GPU (RCCL): no pageable copies per step
            on_gpu = self._buf.is_cuda
            self._host = torch.empty(vec.size, dtype=torch.float64, pin_memory=on_gpu)
        self._host.numpy()[:] = vec
        if self._buf.is_cuda:
            self._buf.copy_(self._host, non_blocking=True)
            self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM)
            self._host.copy_(self._buf)           # synchronises with the all-reduce on the current stream
            out = self._host.numpy().copy()
        else:
            self._buf.copy_(self._host)
            self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM)
            out = self._buf.numpy().copy()
        return unpack(out, shapes)
