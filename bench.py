#!/usr/bin/env python
"""ELBO steps/sec (fp64) of the zero-inflated GP hot path on MI355X -- BASELINE.json's metric.

Workload (config.workload): synthetic N=1e6 rows, D=3, M=1024 inducing points per latent (BASELINE.json configs[2];
generator of SURVEY.md section 8d).  One step = one full-data ELBO value plus its gradient w.r.t. every trainable
parameter (= one L-BFGS-B function evaluation / one sess.run(train_op), scripts/onoff.py:379).  Data is resident in HBM
before the timed region.

`python bench.py --gpus N` starts its own ranks: with N > 1 and no WORLD_SIZE in the environment it launches
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` as a CHILD process (never an
exec) and exits with its return code; under an external torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.
  --scaling weak   (default) every rank holds its own --rows shard (N=8: cfg4, 8e6 rows); `value` counts 1e6-row ELBO steps
                   per second summed over ranks
  --scaling strong the --rows rows are split over the ranks (zigp.parallel.shard_bounds); `value` = steps/s of that one job
Each step ends with ONE all-reduce of the packed [ELBO, KL, gradient] vector (~82 KB): --exchange torch (default) through torch.distributed
(backend nccl = RCCL over xGMI on a device tensor), --exchange library inside libzigp.so (ncclAllReduce on the packed device vector,
zigp_comm_init).  With more than one rank the library exchange is additionally CHECKED after the timed region (one step through it, compared
with the torch.distributed sums, under a watchdog) and the outcome is part of the JSON line: `library_exchange_check`.
`--gpus 1 --force-dist` runs this exact multi-rank code path (process group + communicator in one process) with one rank.

The timed region runs with kernel event timing OFF and the side-stream overlap ON (the fast configuration); the per-kernel
numbers behind `roofline` come from a separate short profiled pass after it (every launch timed, single stream).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))

PEAK_FP64_MFMA = 78.6e12   # vendor fp64 matrix peak, MI355X (256 CU x 2.4 GHz x 128 flop/clk/CU); see DESIGN.md
PEAK_HBM = 8.0e12          # HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a streaming copy reaches


CFG4_SHARDS = 8            # cfg4 = the generator stream over 8e6 rows, rank r takes rows [r 1e6, (r + 1) 1e6)


def synth(N, M, D, rank=0, shards=1):
    """SURVEY.md section 8d generator: RandomState(0) over shards * N rows (X, then the gate noise, then the observation noise), of which
    rows [rank N, (rank + 1) N) are returned; Z from RandomState(1), u from RandomState(2).  shards = 1: cfg2 / cfg3 as they stand;
    a weak-scaling run draws every rank's shard from the cfg4 stream (weak_shard), so the one-GPU line IS the first shard of cfg4."""
    Nt = N * shards
    rs = np.random.RandomState(0)
    X = rs.rand(Nt, D)
    lo, hi = rank * N, (rank + 1) * N
    Xs = np.ascontiguousarray(X[lo:hi])
    del X
    f = np.sin(2 * np.pi * Xs[:, 0]) * np.cos(2 * np.pi * Xs[:, 1]) + Xs[:, 2]
    g = 2 * np.sin(2 * np.pi * (Xs[:, 0] + Xs[:, 2]))
    e1 = rs.randn(Nt)[lo:hi].copy()
    e2 = rs.randn(Nt)[lo:hi].copy()
    Y = np.where(g + e1 > 0, f + 0.1 * e2, 0.0)
    Z = np.random.RandomState(1).rand(M, D)
    ru = np.random.RandomState(2)
    p = dict(Zf=Z.copy(), Zg=Z.copy(), u_fm=0.01 * ru.randn(M, 1), u_gm=0.01 * ru.randn(M, 1),
             u_fs_sqrt=np.ones((M, 1)), u_gs_sqrt=np.ones((M, 1)), ell_f=np.full(D, 0.1), ell_g=np.full(D, 0.1),
             var_f=1.0, var_g=5.0, noise=0.01)
    return Xs, Y, p


def weak_shard(rows, M, D, rank, world):
    """rank r's rows of a weak-scaling run: rows [r rows, (r + 1) rows) of the cfg4 stream (8 shards; more when there are more ranks)"""
    return synth(rows, M, D, rank=rank, shards=max(CFG4_SHARDS, world))


def csrc_hash():
    """hash of the kernel sources: counter summaries under profiles/ are only quoted when they were collected on this code"""
    from zigp import build as zb
    return zb.source_hash(zb.DENSE_FILES)


def cpu_baseline(X, Y, p, jitter, sample_rows, threads, repeats=3):
    """The CPU oracle (reference op order + autograd) timed on a bounded sample of the same workload; median of `repeats`."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import torch
    import zigp_oracle_torch as ot
    torch.set_num_threads(threads)
    Xs, Ys = X[:sample_rows], Y[:sample_rows]
    ot.elbo_and_grad(Xs[:2000], Ys[:2000], p, jitter, chunk=2000)          # warm-up
    ts, data = [], None
    for _ in range(repeats):
        t0 = time.time()
        elbo, data, kl, g = ot.elbo_and_grad(Xs, Ys, p, jitter, chunk=20000)
        ts.append(time.time() - t0)
    return float(np.median(ts)), ts, data


def kernel_key(name):
    """template arguments of a gemm_f64_kernel instantiation as a tuple of strings: ('1', '1', '2', 'false', '1', '8', 'EpiStoreColsum')"""
    if 'gemm_f64_kernel<' not in name:
        return None
    a = name.split('gemm_f64_kernel<')[-1].split('>')[0].replace(' ', '').split(',')
    return tuple(a[:6] + [a[6].split('::')[-1]]) if len(a) >= 7 else None


def live_pmc_traffic(dom_kernel, M, D, chunk, timeout_s=240):
    """HBM bytes per launch of the dominant kernel measured IN THIS RUN: two child `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE
    separately, with --kernel-trace only -- MI355X_MICROARCH.md section HBM) of a short run of the same step (4 full chunks, overlap off),
    corrected as the guide prescribes (FETCH_SIZE x 2 on gfx950, units of KB).  Returns (bytes per launch, launches) or None."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof) or 'rocprof' in os.environ.get('LD_PRELOAD', '') + os.environ.get('ROCP_TOOL_LIBRARIES', ''):
        return None          # no profiler here, or this process already runs under one
    want = kernel_key(dom_kernel)
    vals = {}
    tmp = tempfile.mkdtemp(prefix='zigp_pmc_', dir='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, counter)
            cmd = [prof, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.abspath(__file__),
                   '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-other-configs', '--profile-steps', '0', '--no-overlap', '--no-pmc',
                   '--rows', str(4 * chunk), '--M', str(M), '--D', str(D), '--chunk', str(chunk)]
            env = dict(os.environ, TMPDIR='/tmp')
            for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
                env.pop(k, None)
            # own process group: a pass that overruns is ended as a whole (profiler and the program under it), by its group id only
            pr = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGTERM)
                try:
                    pr.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGKILL)
                    pr.wait()
                return None
            if pr.returncode != 0:
                return None
            files = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
            if not files:
                return None
            tot, n = 0.0, 0
            for row in csv.DictReader(open(files[0])):
                if row['Counter_Name'] == counter and kernel_key(row['Kernel_Name']) == want:
                    tot += float(row['Counter_Value']); n += 1
            if n == 0:
                return None
            vals[counter] = (tot / n, n)
        return (2.0 * vals['FETCH_SIZE'][0] + vals['WRITE_SIZE'][0]) * 1024.0, vals['FETCH_SIZE'][1]
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def headline(args, world, total_rows, dt):
    return {'metric': 'elbo_steps_per_sec', 'value': args.steps / dt * (total_rows / 1e6), 'unit': 'ELBO steps/s (value+gradient, 1e6-row steps, fp64)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic'}


def library_exchange_check(eng, dist, red_dev, p, jitter, scale, ref_out, rank, dt, args, world, total_rows, limit_s=150.0):
    """One step through ncclAllReduce inside libzigp.so (zigp_comm_init) compared with the sums torch.distributed produced for the same
    step, plus a short timing of that path.  Returns a dict for the JSON line.  A hang anywhere inside ends the process after limit_s:
    rank 0 first prints the headline it already has (with the check marked as timed out), then every rank exits with code 3."""
    import threading
    import torch
    from zigp.parallel import ShardedELBO, pack
    done = threading.Event()

    def overrun():
        if done.is_set():
            return
        if rank == 0:
            res = headline(args, world, total_rows, dt)
            res['config'] = {'workload': 'dense zero-inflated GP ELBO step (value+gradient), see bench.py', 'rows_total': total_rows, 'M': args.M, 'D': args.D}
            res['library_exchange_check'] = {'status': 'TIMED OUT after %.0f s inside the library-exchange check (after the timed region); '
                                                       'the headline above is the torch.distributed exchange' % limit_s}
            print(json.dumps(res))
            sys.stdout.flush()
        os._exit(3)      # the launcher sees that a collective hung (rc 3) -- after the headline is out

    timer = threading.Timer(limit_s, overrun)
    timer.daemon = True
    timer.start()
    info = {}
    try:
        eng.comm_set_timeout(45.0)
        shl = ShardedELBO(eng, dist, device=red_dev, library_comm=True)
        if not shl.library_comm:
            info = {'status': 'not formed: the ranks agreed to stay on torch.distributed (RCCL not loadable, zigp_comm_init failed / timed out, or the self-check sum was wrong)'}
        else:
            got = shl.elbo(p, jitter=jitter, scale=scale)
            va, _ = pack(got[0], got[1], got[2])
            vb, _ = pack(ref_out[0], ref_out[1], ref_out[2])
            rel = float(np.max(np.abs(va - vb)) / max(float(np.max(np.abs(vb))), 1e-300))
            rel_e = abs(got[0] - ref_out[0]) / abs(ref_out[0])
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.time()
            nrep = 3
            for _ in range(nrep):
                shl.elbo(p, jitter=jitter, scale=scale)
            torch.cuda.synchronize(); dist.barrier()
            tl = (time.time() - t0) / nrep
            info = {'status': 'ok' if rel <= 1e-12 and rel_e <= 1e-12 else 'MISMATCH', 'ranks': dist.get_world_size(),
                    'elbo_data_rel_diff': rel_e, 'packed_vector_max_rel_diff': rel,
                    'allreduce_calls_in_library': eng.comm_info()['allreduce_calls'], 'ms_per_step_library_exchange': tl * 1e3,
                    'what': 'one step through ncclAllReduce inside libzigp.so vs the same step through torch.distributed all_reduce (sum order '
                            'may differ between the two collectives: 1e-12)'}
            shl.close()
    except Exception as e:       # never lose the headline over the check
        info = {'status': 'error: %r' % (e,)}
    done.set()
    timer.cancel()
    return info


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """--gpus N > 1 without a launcher: start the ranks ourselves as a child torch.distributed.run and pass its rc on."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def timeit(f, n, warm):
    for _ in range(warm):
        f()
    t0 = time.time()
    for _ in range(n):
        f()
    return (time.time() - t0) / n


def kron_fused_products(nb0, nb1, large):
    """16x16x4 MFMA products the fused Kronecker kernels EXECUTE per 16-point tile and latent (forward kernel + backward kernel, which
    recomputes the forward tile), for a grid of nb0 x nb1 16-row blocks (csrc/zigp_kronf.hip, zigp_kronl.h; HISTORY.md section 5b):
    forward A0, A1, B0, C0; backward + B1, C1, P0 dA0, P1 dA1, the sums over points dAlpha, dS2, dP0, dP1 and the moment products."""
    fwd = 4 * (nb0 * nb0 + nb1 * nb1 + 2 * nb0 * nb1)
    bwd = fwd + 2 * 4 * nb0 * nb1 + 4 * (nb0 * nb0 + nb1 * nb1) + 4 * (2 * nb0 * nb1 + nb0 * nb0 + nb1 * nb1) + 4 * (nb0 + nb1)
    return fwd + bwd


def kron_roofline(n_rows, t, nb0, nb1, large=False):
    """cfg5-style entries: the path is MFMA / VALU-issue bound, not HBM bound (HISTORY.md section 5b) -- executed matrix flops over the step"""
    tiles = (n_rows + 15) // 16
    fl = 2.0 * tiles * kron_fused_products(nb0, nb1, large) * 2048.0        # two latents; a 16x16x4 product = 2048 flop
    return dict(executed_mfma_flops=fl, tflops=fl / t / 1e12, frac_mfma=fl / t / PEAK_FP64_MFMA,
                bound='mfma/valu issue (fp64 MFMA and VALU serialise on a SIMD); launch latency at minibatch size')


def oracle_timing(f, repeats=3):
    """median wall time of `repeats` runs of the CPU oracle call f (after one warm-up call)"""
    f()
    ts = []
    for _ in range(repeats):
        t0 = time.time()
        f()
        ts.append(time.time() - t0)
    return float(np.median(ts)), [round(x, 4) for x in ts]


def other_configs(eng, X3, Y3, p3, jitter, cpu_threads=None):
    """The other BASELINE.json configurations and SURVEY section 8d's 'reported separately' numbers, each well under a second
    (N=1 only; `value` above is cfg3 value+gradient).  cpu_threads: time the CPU oracle beside cfg2 and beside the reference's own loop body
    (one 1000-row Kronecker minibatch step, scripts/onoff.py:375-386) with that many torch threads; None: skip."""
    out = {}
    N3, M3 = X3.shape[0], p3['Zf'].shape[0]
    if cpu_threads:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import torch as _torch
        import zigp_oracle_torch as ot
        _torch.set_num_threads(cpu_threads)
    # cfg3 forward-only ELBO and predict (OnOffSVGP.py:160-162; the reference's slowest code is the per-row loop onoffpred.py:176-195)
    t = timeit(lambda: eng.elbo(p3, jitter=jitter, need_grad=False), 2, 1)
    out['cfg3_forward_only'] = dict(ms_per_eval=t * 1e3, evals_per_s=1 / t, frac_4M2N=4.0 * M3 * M3 * N3 / t / PEAK_FP64_MFMA)
    npred = min(262144, N3)
    t = timeit(lambda: eng.predict(p3, X3[:npred], jitter=jitter), 2, 1)
    out['cfg3_predict'] = dict(rows=npred, ms=t * 1e3, rows_per_s=npred / t, note='host X in, (9,N) host out: PCIe-inclusive')
    try:      # the same prediction with the rows and the (9, N) result resident in HBM (zigp_predict_device): only the parameters cross PCIe
        import torch
        X3d = torch.from_numpy(X3).to('cuda:%d' % eng.device)
        o9 = torch.empty((9, N3), dtype=torch.float64, device=X3d.device)
        t = timeit(lambda: eng.predict_device(p3, X3d, jitter=jitter, out=o9), 3, 1)
        out['cfg3_predict_device'] = dict(rows=N3, ms=t * 1e3, rows_per_s=N3 / t, frac_4M2N=4.0 * M3 * M3 * N3 / t / PEAK_FP64_MFMA,
                                          note='device X in, device (9,N) out (zigp_predict_device): 4 M^2 N flops, the A2 panel is never written')
        del X3d, o9
    except Exception as e:
        out['cfg3_predict_device_error'] = repr(e)
    # the per-rank step of cfg3 under 8-GPU STRONG scaling (1e6 / 8 rows, the replicated M x M stage in full), timed on this one GPU
    if N3 >= 8 and M3 >= 128:
        n8 = N3 // 8
        eng.set_data(X3[:n8], Y3[:n8])
        t8 = timeit(lambda: eng.elbo(p3, jitter=jitter), 5, 2)
        out['strong_1of8'] = dict(workload='rows [0, %d) of the headline workload (its 1/8 shard), M=%d, value+gradient: what each rank runs under '
                                           '8-GPU strong scaling' % (n8, M3), ms_per_step=t8 * 1e3, rows_per_pass=eng.get_chunk_rows(M3, n8),
                                  allreduce_us_assumed=50.0,
                                  note='projected_8gpu_speedup (top level) = ms_per_step / (this + the assumed all-reduce): arithmetic on two '
                                       'one-GPU measurements, NOT a measured 8-GPU run; the all-reduce of the 82 KB vector is assumed, never measured here')
    # cfg4 (N = 8e6 = 8 ranks x 1e6 rows) as ONE resident call on this one GPU: the whole of the 8-GPU job's arithmetic, no exchange
    # (tests/test_gpu_fullsize.py::test_cfg4_all_eight_shards... checks the eight per-rank results against exactly this call)
    if N3 == 1000000 and M3 >= 128:
        try:
            import torch
            X8, Y8, _ = synth(CFG4_SHARDS * N3, M3, X3.shape[1])
            X8d, Y8d = torch.from_numpy(X8).to('cuda:%d' % eng.device), torch.from_numpy(Y8).to('cuda:%d' % eng.device)
            eng.set_data_device(X8d, Y8d)
            t4 = timeit(lambda: eng.elbo(p3, jitter=jitter), 2, 1)
            ed4, kl4, _ = eng.elbo(p3, jitter=jitter, need_grad=False)
            out['cfg4_on_one_gpu'] = dict(workload='cfg4 stream, N=%d rows (8 x 1e6), M=%d, value+gradient, ONE resident call on one GPU (no exchange)' % (X8.shape[0], M3),
                                          ms_per_step=t4 * 1e3, steps_per_s_1e6_row_units=X8.shape[0] / 1e6 / t4,
                                          frac_10M2N=10.0 * M3 * M3 * X8.shape[0] / t4 / PEAK_FP64_MFMA, elbo=ed4 - kl4)
            eng.set_data(X3[:1024], Y3[:1024])
            del X8d, Y8d, X8, Y8
            torch.cuda.empty_cache()
        except Exception as e:
            out['cfg4_on_one_gpu_error'] = repr(e)
    # cfg2: N=1e5, M=512
    X2, Y2, p2 = synth(100000, 512, 3)
    eng.set_data(X2, Y2)
    t = timeit(lambda: eng.elbo(p2, jitter=jitter), 8, 2)
    ed, kl, _ = eng.elbo(p2, jitter=jitter)
    out['cfg2'] = dict(workload='N=1e5, D=3, M=512, value+gradient', ms_per_step=t * 1e3, steps_per_s=1 / t,
                       frac_10M2N=10.0 * 512 * 512 * 1e5 / t / PEAK_FP64_MFMA, elbo=ed - kl, rows_per_pass=eng.get_chunk_rows(512, 100000))
    if cpu_threads:
        srows = 20000
        med, ts = oracle_timing(lambda: ot.elbo_and_grad(X2[:srows], Y2[:srows], p2, jitter, chunk=srows))
        out['cfg2']['cpu_baseline'] = dict(value=1.0 / (med * 100000 / srows), unit='ELBO steps/s (extrapolated to 100000 rows)', cores=cpu_threads, kind='port',
                                           sample='oracle (torch CPU fp64, reference op order + autograd) on the first %d rows of cfg2, median of 3 runs: %.2f s '
                                                  'at %d threads; EXTRAPOLATED x %d' % (srows, med, cpu_threads, 100000 // srows), runs_s=ts)
    # cfg5: Kronecker pptr, 32 x 32 (tests/golden/pptr.npz is the reference's data file)
    try:
        from onofftf.model import init_params, engine_params
        d = np.load(os.path.join(ROOT, 'tests', 'golden', 'pptr.npz'))
        Xtr, Ytr = d['Xtrain'].copy(), d['Ytrain']
        Xtr[:, 2] /= 1000.0                                   # create_cvsplits.py:17
        np.random.seed(0)
        pk = engine_params(init_params(Xtr, (32, 32), (32, 32), kmeans_seed=1))
        n5 = Xtr.shape[0]
        eng.set_data(Xtr, Ytr)
        st = eng.kron_stepper(pk)                              # the fit loop's prepared step (fixed model shape): same entry points
        t = timeit(lambda: st(pk, rows=(0, n5), jitter=1e-5), 20, 3)
        # algorithmic bytes: X 24 B + Y 8 B per point read per pass; two passes (value, gradient)
        out['cfg5_full'] = dict(workload='pptr N=%d, 32x32, value+gradient, data resident in HBM' % n5, ms_per_step=t * 1e3,
                                rows_per_s=n5 / t, algorithmic_GBps=2 * 32.0 * n5 / t / 1e9, frac_hbm=2 * 32.0 * n5 / t / PEAK_HBM,
                                **kron_roofline(n5, t, 2, 2))
        t = timeit(lambda: st(pk, Xtr, Ytr, jitter=1e-5), 10, 2)
        out['cfg5_full']['ms_per_step_host_minibatch_in'] = t * 1e3      # PCIe-inclusive: X, Y (3.4 MB) staged and copied every step
        xb, yb = Xtr[:1000], Ytr[:1000]
        t = timeit(lambda: st(pk, xb, yb, jitter=1e-5, scale=n5 / 1000.0), 50, 5)
        t_generic = timeit(lambda: eng.kron_elbo(pk, xb, yb, jitter=1e-5, scale=n5 / 1000.0), 50, 5)
        out['cfg5_mb1000'] = dict(workload='pptr minibatch 1000 (scripts/onoff.py:55), 32x32', ms_per_step=t * 1e3, steps_per_s=1 / t,
                                  ms_per_step_unprepared_call=t_generic * 1e3, **kron_roofline(1000, t, 2, 2))
        np.random.seed(0)
        pk2 = engine_params(init_params(Xtr, (10, 100), (10, 100), kmeans_seed=1))
        st2 = eng.kron_stepper(pk2)
        t = timeit(lambda: st2(pk2, xb, yb, jitter=1e-5, scale=n5 / 1000.0), 50, 5)
        out['ref_grid_10x100_mb1000'] = dict(workload='pptr minibatch 1000, the reference\'s [10,100] grid (scripts/onoff.py:52-53)',
                                              ms_per_step=t * 1e3, steps_per_s=1 / t, **kron_roofline(1000, t, 1, 7, True))
        t = timeit(lambda: st2(pk2, rows=(0, n5), jitter=1e-5), 10, 2)
        out['ref_grid_10x100_full'] = dict(workload='pptr N=%d full batch, the reference\'s [10,100] grid, value+gradient, data resident' % n5,
                                            ms_per_step=t * 1e3, rows_per_s=n5 / t, **kron_roofline(n5, t, 1, 7, True))
        # the fit loop as onoff() runs it since round 4: 200 iterations per call on the device (zigp_kron_fit_steps), one synchronisation per call
        from onofftf.model import KronDeviceFit
        rbs = [int(r) for r in np.random.RandomState(3).randint(0, n5 - 1000, size=200)]
        for key, grid in (('cfg5_mb1000', (32, 32)), ('ref_grid_10x100_mb1000', (10, 100))):
            np.random.seed(0)
            fitter = KronDeviceFit(eng, init_params(Xtr, grid, grid, kmeans_seed=1))
            fitter.steps(rbs[:20], 1000, 1e-5, n5 / 1000.0)                      # warm-up
            t0 = time.time()
            ed_, kl_ = fitter.steps(rbs, 1000, 1e-5, n5 / 1000.0)
            td = (time.time() - t0) / len(rbs)
            out[key]['device_loop'] = dict(ms_per_step_amortised=td * 1e3, steps_per_s=1 / td, steps_per_call=len(rbs),
                                           what='zigp_kron_fit_steps: gradient + Log1pe chain + per-learning-rate Adam update on the device, '
                                                'ONE host synchronisation per call (ms_per_step above = one host call per iteration, no update)',
                                           cost_first_last=[float(-(ed_[0] - kl_[0])), float(-(ed_[-1] - kl_[-1]))])
        if cpu_threads:      # the reference's own timing hook: wall time per iteration of the Kronecker loop (scripts/onoff.py:376,385-386)
            for key, pq in (('cfg5_mb1000', pk), ('ref_grid_10x100_mb1000', pk2)):
                med, ts = oracle_timing(lambda: ot.kron_elbo_and_grad(xb, yb, pq, 1e-5, scale=n5 / 1000.0))
                out[key]['cpu_baseline'] = dict(value=1.0 / med, unit='minibatch steps/s (value + gradient, no update)', cores=cpu_threads, kind='port',
                                                sample='oracle: LITERAL kron_inf + GaussKLkron (dense Kronecker products, scripts/onoff.py:186-241, onofftf/main.py:350-387) '
                                                       '+ torch autograd on ONE 1000-row minibatch (the whole unit of work, not a sample), median of 3 runs: %.3f s at %d threads'
                                                       % (med, cpu_threads), runs_s=ts)
        t = timeit(lambda: eng.kron_predict(pk, Xtr, jitter=1e-6, g_offset=-1.0), 3, 1)
        out['cfg5_predict'] = dict(rows=n5, ms=t * 1e3, rows_per_s=n5 / t)
    except Exception as e:   # the Kronecker numbers are extras: never lose the headline line over them
        out['cfg5_error'] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--rows', type=int, default=1000000, help='rows per GPU (weak) or in total (strong)')
    ap.add_argument('--M', type=int, default=1024)
    ap.add_argument('--D', type=int, default=3)
    ap.add_argument('--chunk', type=int, default=None, help='rows per pass; default: the library rule 32768 * 1024 / M, clamped to [32768, 131072]')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true')
    ap.add_argument('--no-overlap', action='store_true', help='timed region without stream overlap')
    ap.add_argument('--no-pmc', action='store_true', help='skip the two child rocprofv3 --pmc passes that measure the dominant kernel\'s HBM traffic')
    ap.add_argument('--profile-steps', type=int, default=1, help='steps of the separate profiled pass (0: none)')
    ap.add_argument('--backend', default='nccl', help="'nccl' (= RCCL, the default) or 'gloo' to rehearse the multi-rank path on fewer GPUs than ranks")
    ap.add_argument('--cpu-sample-rows', type=int, default=60000)
    ap.add_argument('--exchange', choices=('torch', 'library'), default='torch',
                    help="where the per-step all-reduce runs: torch.distributed on the packed vector (default) or ncclAllReduce inside libzigp.so")
    ap.add_argument('--force-dist', action='store_true', help='with --gpus 1: still initialise the process group and run the multi-rank code path with one rank')
    ap.add_argument('--no-library-check', action='store_true', help='skip the post-timing check of the library exchange against the torch.distributed sums')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)              # before anything touches the GPU; never returns

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: WORLD_SIZE=%d but --gpus %d: launch with --nproc-per-node == --gpus\n' % (world, args.gpus))
        sys.exit(2)
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    dev = local_rank % ndev          # a gloo rehearsal may put several ranks on one GPU
    if world == 1 and args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod
        torch.cuda.set_device(dev)
        if args.backend == 'nccl':
            dist_mod.init_process_group(backend='nccl', device_id=torch.device('cuda', dev))
        else:
            dist_mod.init_process_group(backend=args.backend)
        dist = dist_mod

    from zigp import build as zigp_build
    if local_rank == 0:
        zigp_build.ensure()                   # (re)builds libzigp.so when it is missing or older than its sources
    if dist is not None:
        dist.barrier()
    import zigp
    from zigp.parallel import ShardedELBO, shard_bounds
    M, D, jitter = args.M, args.D, 1e-6
    if args.scaling == 'weak':
        N = args.rows
        X, Y, p = weak_shard(N, M, D, rank, world)
        total_rows = N * world
    else:
        Xa, Ya, p = synth(args.rows, M, D, 0)
        lo, hi = shard_bounds(args.rows, world, rank)
        X, Y = np.ascontiguousarray(Xa[lo:hi]), np.ascontiguousarray(Ya[lo:hi])
        N = hi - lo
        total_rows = args.rows
    eng = zigp.DenseEngine(dev)               # raises if libzigp.so is missing: no CPU fallback
    if args.chunk is not None:
        eng.set_chunk(args.chunk)
    args.chunk = eng.get_chunk(M)             # what the library uses (its own rule unless --chunk fixed it): reported, and matched against PMC summaries
    Xd = torch.from_numpy(X).to('cuda:%d' % dev)
    Yd = torch.from_numpy(Y).to('cuda:%d' % dev)
    eng.set_data_device(Xd, Yd)               # inputs resident in HBM before timing
    red_dev = ('cuda:%d' % dev) if args.backend == 'nccl' else 'cpu'
    sh = ShardedELBO(eng, dist, device=red_dev, library_comm=(args.exchange == 'library' and args.backend == 'nccl'))
    scale = 1.0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng.profile_enable(False)
    eng.set_overlap(0 if args.no_overlap else 1)
    for _ in range(args.warmup):
        out = sh.elbo(p, jitter=jitter, scale=scale)
    barrier()
    ck0 = eng.clock_stamp()
    t0 = time.time()
    for _ in range(args.steps):
        out = sh.elbo(p, jitter=jitter, scale=scale)
    barrier()
    dt = time.time() - t0
    clock_timed = eng.clock_mhz(ck0, eng.clock_stamp())
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    elbo_data, kl = out[0], out[1]

    # ---- the library exchange against the torch.distributed sums (all ranks; after the timed region, so that nothing it does can cost
    # the headline number).  Everything that can block has a time limit: zigp_comm_init gives up by itself, and a watchdog ends the
    # process (rank 0 prints the line it has, marked) if the check as a whole overruns.
    lib_check = None
    if dist is not None and args.backend == 'nccl' and not sh.library_comm and not args.no_library_check:
        lib_check = library_exchange_check(eng, dist, red_dev, p, jitter, scale, out, rank, dt, args, world, total_rows)
    elif sh.library_comm:
        lib_check = {'status': 'the timed region itself ran on the library exchange'}

    # separate profiled pass (rank 0 only, its own shard, no all-reduce): every launch timed with HIP events on the stream
    # it runs on, single stream, so the per-kernel durations are those of kernels running alone
    prof, prof_wall_ms, clock_prof = None, None, None
    if rank == 0 and args.profile_steps > 0:
        eng.set_overlap(False)
        eng.profile_sampling(1)
        eng.profile_enable(True)
        eng.elbo(p, jitter=jitter, scale=scale, include_kl=True)
        eng.profile_reset()
        ckp = eng.clock_stamp()
        tp = time.time()
        for _ in range(args.profile_steps):
            eng.elbo(p, jitter=jitter, scale=scale, include_kl=True)
        prof_wall_ms = (time.time() - tp) / args.profile_steps * 1e3
        clock_prof = eng.clock_mhz(ckp, eng.clock_stamp())
        prof = eng.profile_get()
        eng.profile_enable(False)
        eng.profile_sampling(8)
        eng.set_overlap(0 if args.no_overlap else 1)

    extras_failed = []
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        res = headline(args, world, total_rows, dt)
        res.update({
            'config': {'workload': 'dense zero-inflated GP ELBO step (value+gradient), N=%d rows %s, D=%d, M=%d per latent, full batch'
                                   % (args.rows, 'per GPU' if args.scaling == 'weak' else 'in total', D, M),
                       'rows_this_rank': N, 'rows_total': total_rows, 'M': M, 'D': D, 'chunk_rows': args.chunk, 'jitter': jitter,
                       'parallelism': 'row-shard x%d, 1 all-reduce/step' % world, 'timed_region': 'event timing off, stream overlap %s' % ('off' if args.no_overlap else 'on')},
            'n_ranks_seen': dist.get_world_size() if dist is not None else 1,
            'backend': (('rccl(nccl)' if args.backend == 'nccl' else args.backend) if dist is not None else 'none'),
            'exchange': ('ncclAllReduce inside libzigp.so (zigp_comm_init), %d calls' % eng.comm_info()['allreduce_calls']) if sh.library_comm
                        else ('torch.distributed all_reduce of the packed vector (%s)' % ('device tensor, RCCL' if args.backend == 'nccl' else 'host tensor, gloo')
                              if dist is not None else 'none'),
            'library_exchange_check': lib_check,
            'elbo': elbo_data - kl, 'elbo_data': elbo_data, 'kl': kl,
            'sustained_clock_mhz': {'timed_region': clock_timed, 'profiled_pass': clock_prof,
                                    'how': 's_memtime / s_memrealtime stamps on every XCD before and after the region (zigp_clock_stamp), median over XCDs; nominal 2400'},
        })
        if prof is not None:
            from zigp._lib import PROF_KERNELS
            gemm_classes = tuple(k for k in ('gemm_A1', 'gemm_A2', 'gemm_H', 'gemm_J', 'syrk') if prof[k]['launches'] > 0)   # (H: folded into J' since round 4)
            dom = max(gemm_classes, key=lambda k: prof[k]['ms'])          # dominant kernel = largest share of the step
            gk = prof[dom]
            avg_launch_s = gk['ms'] * 1e-3 / max(gk['launches'], 1)
            flops_per_launch = gk['flops'] / max(gk['launches'], 1)     # algorithmic, triangle-aware: M^2 * chunk_rows, averaged over all launches
            achieved = flops_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0
            gemm_ms = sum(prof[k]['ms'] for k in gemm_classes)
            gemm_fl = sum(prof[k]['flops'] for k in gemm_classes)
            # HBM traffic of the dominant kernel: measured in this run by two child rocprofv3 --pmc passes (live_pmc_traffic); if that is
            # not possible (no profiler, already under one, --no-pmc) it is quoted from a summary under profiles/ collected on these
            # kernel sources at this (M, chunk, D), with the file named; null otherwise.  The MFMA busy fraction (three more passes)
            # is always quoted from such a summary (tools/pmc_mfma.sh).
            h = csrc_hash()
            traffic, traffic_src, mfma_util = None, None, None
            if not args.no_pmc and world == 1:      # measured in THIS run (child rocprofv3 --pmc passes); the committed summary is the fallback
                live = live_pmc_traffic(PROF_KERNELS[dom], M, D, args.chunk)
                if live is not None:
                    traffic = live[0]
                    traffic_src = ('this run: child rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --rows %d --no-overlap`, '
                                   '%d launches of the kernel, FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, KB -> bytes' % (4 * args.chunk, live[1]))
            for fn in sorted(os.listdir(os.path.join(ROOT, 'profiles')), reverse=True):
                try:
                    if fn.endswith('_pmc_hbm_traffic.json') and traffic is None:
                        tr = json.load(open(os.path.join(ROOT, 'profiles', fn)))
                        meta = tr.get('_meta', {})
                        if meta.get('csrc_hash') == h and (meta.get('M'), meta.get('chunk'), meta.get('D')) == (M, args.chunk, D):
                            sym = PROF_KERNELS[dom].split('>')[0].replace('gemm_f64_kernel<', '').split(',')
                            for name, v in tr.items():
                                a_ = name.split('gemm_f64_kernel<')[-1].split('>')[0].replace(' ', '').split(',') if 'gemm_f64_kernel<' in name else []
                                if len(a_) >= 7 and a_[:6] == sym[:6] and a_[6].endswith(sym[6]) and v['launches'] >= 8:
                                    traffic, traffic_src = v['hbm_bytes_per_launch_corrected'], 'profiles/' + fn
                    if fn.endswith('_pmc_mfma_util.json') and mfma_util is None:
                        mu = json.load(open(os.path.join(ROOT, 'profiles', fn)))
                        meta = mu.get('_meta', {})
                        if meta.get('csrc_hash') == h and (meta.get('M'), meta.get('chunk'), meta.get('D')) == (M, args.chunk, D):
                            mfma_util = {'source': 'profiles/' + fn, 'collected_at_rows': meta.get('rows'),
                                         'busy_frac': {k.split('gemm_f64_kernel')[-1].split('(')[0]: round(v['mfma_busy_frac_of_simd_cycles'], 3)
                                                       for k, v in mu.items() if k != '_meta' and v['launches'] >= 8 and v['avg_us_under_pmc'] > 300}}
                except Exception:
                    pass
            Mp = (M + 127) // 128 * 128
            hbm = {}
            # kgrad reads the J' and the K panel (16 Mp bytes per column): since late round 4 it reads K again instead of recomputing it from
            # x, z (ZIGP_KGRAD_RECOMPUTE = 0: beside the MFMA-bound products the saved fp64 VALU work is worth more than the bytes)
            for k, bytes_per_col, bound in (('kgrad', 16.0 * Mp + 8.0 * (D + 2), 'HBM read (J\' and K panels; 12.5 fp64 instructions per element)'),
                                            ('kuf_build', 8.0 * Mp + 8.0 * D, 'HBM write')):
                if prof[k]['launches'] > 0 and prof[k]['ms'] > 0:
                    # bytes over ALL launches of the profiled pass: every column of the shard is swept once per latent and step
                    nbytes = bytes_per_col * N * 2 * args.profile_steps
                    hbm[k] = {'GBps': nbytes / (prof[k]['ms'] * 1e-3) / 1e9, 'frac_of_8TBps': nbytes / (prof[k]['ms'] * 1e-3) / PEAK_HBM, 'bound': bound}
            res['roofline'] = {
                'bound': 'mfma', 'achieved': achieved / 1e12, 'peak': PEAK_FP64_MFMA / 1e12, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_FP64_MFMA, 'traffic': traffic, 'traffic_source': traffic_src,
                'kernel': PROF_KERNELS[dom], 'measured_in': 'separate profiled pass of this run (HIP events on the launch stream, every launch, overlap off)',
                'per_kernel_tflops': {k: (prof[k]['flops'] / (prof[k]['ms'] * 1e-3) / 1e12 if prof[k]['ms'] > 0 else 0.0) for k in gemm_classes},
                'flops_per_launch': flops_per_launch, 'avg_launch_ms': avg_launch_s * 1e3, 'launches_timed': gk['launches'],
                'all_gemm_tflops': gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
                # whole step (timed region) against the same peak, counting the 10 M^2 N flops this engine's algorithm executes
                # (two triangular products + one full product of twice their flops + one symmetric rank-N update per latent)
                'step_frac_10M2N': (10.0 * M * M * N / (dt / args.steps)) / PEAK_FP64_MFMA,
            }
            res['side_kernels'] = hbm
            res['mfma_busy_pmc'] = mfma_util
            # device time per class in the profiled pass (exact sums, no extrapolation).  mxm_stage covers BOTH latents' chains, which
            # run concurrently on two streams: wall time of the forward stage + wall time of the reverse stage (events on the main
            # stream around fork ... join; until round 3 the reverse part was the SUM of the two concurrent chains, which overstated
            # the stage by ~0.8 ms).  It is listed apart and not part of the single-stream sum.
            kms = {k: v['ms'] / args.profile_steps for k, v in prof.items()}
            res['profiled_pass'] = {'ms_per_step_wall': prof_wall_ms,
                                    'kernel_ms_per_step': {k: v for k, v in kms.items() if k != 'mxm_stage'},
                                    'mxm_stage_ms_both_streams': kms.get('mxm_stage', 0.0),
                                    'chunk_loop_ms_sum': sum(v for k, v in kms.items() if k != 'mxm_stage')}
        if not args.no_other_configs and world == 1:
            try:
                ncpu_oc = len(os.sched_getaffinity(0))
            except Exception:
                ncpu_oc = os.cpu_count() or 1
            res['other_configs'] = other_configs(eng, X, Y, p, jitter, cpu_threads=None if args.no_cpu_baseline else min(16, ncpu_oc))
            eng.set_data_device(Xd, Yd)
            s8 = res['other_configs'].get('strong_1of8')
            if s8:
                res['projected_8gpu_speedup'] = ms_per_step / (s8['ms_per_step'] + s8['allreduce_us_assumed'] * 1e-3)
        if not args.no_cpu_baseline and world == 1:
            srows = min(args.cpu_sample_rows, N)
            try:
                ncpu = len(os.sched_getaffinity(0))
            except Exception:
                ncpu = os.cpu_count() or 1
            # a one-GPU box owns a 16-core share of its host (more threads than that oversubscribe the share: 256 threads ran
            # 14x SLOWER than 16 on the 256-thread host, profiles/r02a_bench.json)
            threads = min(16, ncpu)
            med, ts, cdata = cpu_baseline(X, Y, p, jitter, srows, threads)
            gdata = eng.elbo(p, jitter=jitter, rows=(0, srows), include_kl=False, need_grad=False)[0]
            res['cpu_baseline'] = {'value': 1.0 / (med * N / srows), 'unit': 'ELBO steps/s (extrapolated to %d rows)' % N,
                                   'cores': threads, 'kind': 'port', 'host_cpus_visible': ncpu,
                                   'sample': 'oracle (torch CPU fp64, reference op order + autograd) on the first %d rows in 20000-row chunks, '
                                             'median of 3 runs: %.2f s at %d threads (the box\'s CPU share)' % (srows, med, threads),
                                   'runs_s': [round(x, 3) for x in ts],
                                   'elbo_data_rel_diff_on_sample': abs(gdata - cdata) / abs(cdata)}
        print(json.dumps(res))
        sys.stdout.flush()
        extras_failed = [k for k in res.get('other_configs', {}) if k.endswith('_error')]
    if dist is not None:
        dist.barrier()
        sh.close()
        dist.destroy_process_group()
    eng.close()
    if rank == 0 and extras_failed:      # the headline is out; a configuration that silently stopped being measured must not pass unnoticed
        sys.stderr.write('bench.py: other_configs raised: %s\n' % ', '.join(extras_failed))
        sys.exit(4)


if __name__ == '__main__':
    main()
