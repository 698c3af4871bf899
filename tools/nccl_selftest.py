"""Single-rank RCCL rehearsal of the N > 1 exchange on a one-GPU box: init_process_group('nccl', world_size=1), then the
packed [elbo_data, kl, grads] vector goes through the same copy -> all_reduce(SUM) -> copy path ShardedELBO uses for N > 1,
plus the barrier / MAX-reduce bench.py brackets its timed region with."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import torch, torch.distributed as dist
import bench, zigp
from zigp.parallel import ShardedELBO, pack, unpack
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29655')
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
X, Y, p = bench.synth(50000, 256, 3)
eng = zigp.DenseEngine(0); eng.set_data(X, Y)
sh = ShardedELBO(eng, dist, device='cuda:0')
ed, kl, g = sh.elbo(p)                      # world == 1: returns the local result
vec, shapes = pack(ed, kl, g)
buf = torch.empty(vec.size, dtype=torch.float64, device='cuda:0'); buf.copy_(torch.from_numpy(vec))
dist.all_reduce(buf, op=dist.ReduceOp.SUM)
ed2, kl2, g2 = unpack(buf.cpu().numpy(), shapes)
assert ed2 == ed and kl2 == kl and all(np.array_equal(np.asarray(g[k]), np.asarray(g2[k])) for k in g)
sh.world = 2                                 # force the N > 1 branch (pinned staging -> all_reduce on the GPU -> pinned -> unpack); the sum over ONE rank is the identity
ed3, kl3, g3 = sh.elbo(p)
assert ed3 == ed and kl3 == kl and all(np.array_equal(np.asarray(g[k]), np.asarray(g3[k])) for k in g)
assert sh._buf.is_cuda and sh._host.is_pinned()
t = torch.tensor([1.25], dtype=torch.float64, device='cuda:0'); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
assert float(t.item()) == 1.25
dist.destroy_process_group()
print('nccl (RCCL) single-rank exchange ok: %d doubles' % vec.size)
