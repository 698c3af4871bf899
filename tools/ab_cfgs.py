"""Wall-clock step times of one build of libzigp.so (ZIGP_LIB) at cfg3 (N=1e6, M=1024), the 125 000-row shard of it and cfg2 (N=1e5, M=512):
one line, for same-box comparisons of builds -- `for L in a.so b.so; do ZIGP_LIB=$L python tools/ab_cfgs.py; done`, several rounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import bench, zigp, torch
e = zigp.DenseEngine(0)
out = []
for name, N, M, rows, reps in (('cfg3', 1000000, 1024, None, 3), ('shard125k', 1000000, 1024, (0, 125000), 10), ('cfg2', 100000, 512, None, 20)):
    X, Y, p = bench.synth(N, M, 3)
    e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    kw = {} if rows is None else {'rows': rows}
    for _ in range(2): e.elbo(p, **kw)
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        for _ in range(reps): e.elbo(p, **kw)
        best = min(best, (time.time() - t0) / reps * 1e3)
    out.append('%s %.3f' % (name, best))
    if name == 'cfg3':          # value-only ELBO of the same rows (the forward pass alone: A1, A2 column sums, point-wise)
        e.elbo(p, need_grad=False)
        t0 = time.time()
        for _ in range(3): e.elbo(p, need_grad=False)
        out.append('cfg3fwd %.3f' % ((time.time() - t0) / 3 * 1e3))
print(os.path.basename(os.environ.get('ZIGP_LIB', 'libzigp.so')), ' '.join(out), 'ms')
