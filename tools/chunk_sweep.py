"""Wall-clock step at several chunk sizes (zigp_set_chunk): cfg3 (N = 1e6, M = 1024), its 125 000-row shard, cfg2 (N = 1e5, M = 512).
`python tools/chunk_sweep.py 16384 24576 ...`: cfg3 only, at the chunk sizes given."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import bench, zigp, torch
e = zigp.DenseEngine(0)
CASES = (('cfg3', 1000000, 1024, None, 3, (32768, 65536, 131072, 40960)), ('shard125k', 1000000, 1024, (0, 125000), 10, (32768, 65536, 131072, 43008)),
         ('cfg2', 100000, 512, None, 20, (65536, 131072, 51200)))
if len(sys.argv) > 1:
    CASES = (('cfg3', 1000000, 1024, None, 3, tuple(int(a) for a in sys.argv[1:])),)
for name, N, M, rows, reps, chunks in CASES:
    X, Y, p = bench.synth(N, M, 3)
    e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    kw = {} if rows is None else {'rows': rows}
    for ch in chunks:
        e.set_chunk(ch)
        for _ in range(2): e.elbo(p, **kw)
        best = 1e9
        for _ in range(3):
            t0 = time.time()
            for _ in range(reps): e.elbo(p, **kw)
            best = min(best, (time.time() - t0) / reps * 1e3)
        print('%s chunk %6d: %.3f ms' % (name, ch, best), flush=True)
