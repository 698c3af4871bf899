"""Repeat one dense step many times and compare every result bit for bit with the first (stream / event ordering of the three-stream step,
reuse of buffers between calls): `python tools/repeat_check.py [N M reps]` (default 100000 512 1500; also 3 x 20 steps of cfg3)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import bench, zigp, torch
N, M, reps = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 512, 1500)))
e = zigp.DenseEngine(0)


def same(a, b):
    return all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in a)


for n, m, r in ((N, M, reps), (1000000, 1024, 20)):
    X, Y, p = bench.synth(n, m, 3)
    e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    ref = e.elbo(p)
    t0 = time.time()
    bad = 0
    for i in range(r):
        if i % 3 == 1:
            out = e.elbo(p, rows=(0, n // 2))        # a different row range in between: buffers are re-sized / re-zeroed
            continue
        out = e.elbo(p)
        if not (out[0] == ref[0] and out[1] == ref[1] and same(out[2], ref[2])):
            bad += 1
    print('N=%d M=%d: %d calls in %.1f s, %d differ from the first' % (n, m, r, time.time() - t0, bad), flush=True)
    assert bad == 0
print('repeat check ok')
