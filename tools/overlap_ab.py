"""Same-process A/B of the side-stream overlap (zigp_set_overlap) on the cfg3 step, profiling off."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/zero-inflated-gp_amd')
import bench, zigp, torch
X, Y, p = bench.synth(1000000, 1024, 3)
e = zigp.DenseEngine(0); e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
e.elbo(p)
for rnd in range(2):
    for on in (False, True):
        e.set_overlap(on)
        e.elbo(p)
        t0 = time.time()
        for _ in range(4): e.elbo(p)
        print('overlap %-5s %.2f ms/step' % (on, (time.time() - t0) / 4 * 1e3))
