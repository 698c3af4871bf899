"""Where a cfg2 step (N=1e5, M=512, D=3) spends its time: wall clock vs the per-class HIP-event totals."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/zero-inflated-gp_amd')
import bench, zigp, torch
N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 512)
X, Y, p = bench.synth(N, M, 3)
e = zigp.DenseEngine(0)
if len(sys.argv) > 3: e.set_chunk(int(sys.argv[3]))
e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
for _ in range(3): e.elbo(p)
t0 = time.time()
for _ in range(20): e.elbo(p)
print('wall: %.3f ms/step' % ((time.time() - t0) / 20 * 1e3))
t0 = time.time()
for _ in range(20): e.elbo(p, rows=(0, 0))
print('MxM only (no rows): %.3f ms' % ((time.time() - t0) / 20 * 1e3))
e.profile_enable(True); e.profile_reset(); e.elbo(p)
pr = e.profile_get()
print({k: round(v['est_total_ms'], 3) for k, v in pr.items()}, 'sum %.3f' % sum(v['est_total_ms'] for v in pr.values()))
