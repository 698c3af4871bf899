"""Same-process A/B of the stream-overlap modes (zigp_set_overlap 0 / 1): python tools/overlap_ab.py [N] [M]; results must be bit-identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import numpy as np, torch, bench, zigp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
X, Y, p = bench.synth(N, M, 3)
e = zigp.DenseEngine(0)
e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
ref = None
reps = 3 if N * M >= 5e8 else 20
for rnd in range(3):
    for mode in (0, 1):
        e.set_overlap(mode)
        out = e.elbo(p)
        t0 = time.time()
        for _ in range(reps): out = e.elbo(p)
        dt = (time.time() - t0) / reps * 1e3
        if ref is None: ref = out
        same = out[0] == ref[0] and all(np.array_equal(np.asarray(out[2][k]), np.asarray(ref[2][k])) for k in ref[2])
        print('round %d overlap mode %d: %.3f ms/step  bit-identical %s' % (rnd, mode, dt, same), flush=True)
