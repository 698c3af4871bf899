#!/usr/bin/env python
"""ELBO steps/sec (fp64) of the zero-inflated GP hot path on MI355X -- BASELINE.json's metric.

Workload (config.workload): synthetic N=1e6 rows per GPU, D=3, M=1024 inducing points per latent
(BASELINE.json configs[2]; generator of SURVEY.md section 8d).  One step = one full-data ELBO value plus
its gradient w.r.t. every trainable parameter (= one L-BFGS-B function evaluation / one sess.run(train_op),
scripts/onoff.py:379).  Data is resident in HBM before the timed region.  With --gpus N (launched by
torch.distributed.run, one rank per GPU) every rank holds its own 1e6-row shard (weak scaling, = cfg4 at N=8)
and the packed [ELBO, gradient] vector is all-reduced over RCCL each step; `value` counts 1e6-row ELBO steps
per second summed over ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))

PEAK_FP64_MFMA = 78.6e12   # vendor fp64 matrix peak, MI355X (256 CU x 2.4 GHz x 128 flop/clk/CU); see DESIGN.md


def synth(N, M, D, rank=0):
    """SURVEY.md section 8d generator; rank r of a multi-GPU run draws its own shard (seed r)."""
    rs = np.random.RandomState(rank)
    X = rs.rand(N, D)
    f = np.sin(2 * np.pi * X[:, 0]) * np.cos(2 * np.pi * X[:, 1]) + X[:, 2]
    g = 2 * np.sin(2 * np.pi * (X[:, 0] + X[:, 2]))
    Y = np.where(g + rs.randn(N) > 0, f + 0.1 * rs.randn(N), 0.0)
    Z = np.random.RandomState(1001).rand(M, D)
    ru = np.random.RandomState(1002)
    p = dict(Zf=Z.copy(), Zg=Z.copy(), u_fm=0.01 * ru.randn(M, 1), u_gm=0.01 * ru.randn(M, 1),
             u_fs_sqrt=np.ones((M, 1)), u_gs_sqrt=np.ones((M, 1)), ell_f=np.full(D, 0.1), ell_g=np.full(D, 0.1),
             var_f=1.0, var_g=5.0, noise=0.01)
    return X, Y, p


def cpu_baseline(X, Y, p, jitter, sample_rows, threads):
    """The CPU oracle (reference op order + autograd) timed on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import torch
    import zigp_oracle_torch as ot
    torch.set_num_threads(threads)
    Xs, Ys = X[:sample_rows], Y[:sample_rows]
    ot.elbo_and_grad(Xs[:2000], Ys[:2000], p, jitter, chunk=2000)          # warm-up
    t0 = time.time()
    elbo, data, kl, g = ot.elbo_and_grad(Xs, Ys, p, jitter, chunk=20000)
    dt = time.time() - t0
    return dt, data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--rows', type=int, default=1000000, help='rows per GPU')
    ap.add_argument('--M', type=int, default=1024)
    ap.add_argument('--D', type=int, default=3)
    ap.add_argument('--chunk', type=int, default=32768)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help="'nccl' (= RCCL, the default) or 'gloo' to rehearse the multi-rank path on fewer GPUs than ranks")
    ap.add_argument('--cpu-sample-rows', type=int, default=60000)
    args = ap.parse_args()

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    dev = local_rank % ndev          # gloo rehearsal may put several ranks on one GPU
    if world > 1:
        import torch.distributed as dist_mod
        torch.cuda.set_device(dev)
        if args.backend == 'nccl':
            dist_mod.init_process_group(backend='nccl', device_id=torch.device('cuda', dev))
        else:
            dist_mod.init_process_group(backend=args.backend)
        dist = dist_mod
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node == --gpus'

    from zigp import build as zigp_build
    if local_rank == 0:
        zigp_build.ensure()                   # builds libzigp.so only if the snapshot does not carry it (git checkout)
    if dist is not None:
        dist.barrier()
    import zigp
    from zigp.parallel import ShardedELBO
    N, M, D, jitter = args.rows, args.M, args.D, 1e-6
    X, Y, p = synth(N, M, D, rank)
    eng = zigp.DenseEngine(dev)               # raises if libzigp.so is missing: no CPU fallback
    eng.set_chunk(args.chunk)
    Xd = torch.from_numpy(X).to('cuda:%d' % dev)
    Yd = torch.from_numpy(Y).to('cuda:%d' % dev)
    eng.set_data_device(Xd, Yd)               # inputs resident in HBM before timing
    red_dev = ('cuda:%d' % dev) if args.backend == 'nccl' else 'cpu'
    sh = ShardedELBO(eng, dist, device=red_dev)
    scale = 1.0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = sh.elbo(p, jitter=jitter, scale=scale)
    eng.profile_enable(True)
    eng.profile_reset()
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        out = sh.elbo(p, jitter=jitter, scale=scale)
    barrier()
    dt = time.time() - t0
    prof = eng.profile_get()
    eng.profile_enable(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    elbo_gpu = out[0] - out[1]

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * args.steps / dt * (N / 1e6)
        from zigp._lib import PROF_KERNELS
        gemm_classes = ('gemm_A1', 'gemm_A2', 'gemm_H', 'gemm_J', 'syrk')
        dom = max(gemm_classes, key=lambda k: prof[k]['est_total_ms'])          # dominant kernel = largest share of the timed region
        gk = prof[dom]
        avg_launch_s = gk['ms'] * 1e-3 / max(gk['launches'], 1)
        flops_per_launch = gk['flops'] / max(gk['launches'], 1)     # algorithmic, triangle-aware: M^2 * chunk_rows
        achieved = flops_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0
        gemm_ms = sum(prof[k]['ms'] for k in gemm_classes)
        gemm_fl = sum(prof[k]['flops'] for k in gemm_classes)
        # HBM traffic per launch of the dominant kernel: from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        # (tools/pmc_traffic.py; FETCH_SIZE doubled per the gfx950 correction).  Per-launch traffic depends on (M, chunk)
        # only, so the summary collected at this chunk size applies; null when it is missing or the shape differs.
        traffic = None
        try:
            if (M, args.chunk, D) == (1024, 32768, 3):
                tr = json.load(open(os.path.join(ROOT, 'profiles', 'r01h_pmc_hbm_traffic.json')))
                sym = PROF_KERNELS[dom].split('>')[0].replace('gemm_f64_kernel<', '').split(',')
                for name, v in tr.items():
                    args_ = name.split('gemm_f64_kernel<')[-1].split('>')[0].replace(' ', '').split(',') if 'gemm_f64_kernel<' in name else []
                    if len(args_) >= 7 and args_[:6] == sym[:6] and args_[6].endswith(sym[6]) and v['launches'] >= 8:
                        traffic = v['hbm_bytes_per_launch_corrected']
        except Exception:
            traffic = None
        # HBM-bound side kernels: algorithmic bytes / measured average launch time
        Mp = (M + 127) // 128 * 128
        panel = 8.0 * Mp * args.chunk
        hbm = {}
        for k, nbytes in (('kgrad', 2 * panel), ('kuf_build', panel)):
            if prof[k]['launches'] > 0 and prof[k]['ms'] > 0:
                hbm[k] = {'GBps': nbytes / (prof[k]['ms'] * 1e-3 / prof[k]['launches']) / 1e9, 'bytes_per_launch': nbytes}
        mfma_util = None
        try:
            mu = json.load(open(os.path.join(ROOT, 'profiles', 'r01h_pmc_mfma_util.json')))
            mfma_util = {k.split('gemm_f64_kernel')[-1].split('(')[0]: round(v['mfma_busy_frac_of_simd_cycles'], 3)
                         for k, v in mu.items() if v['launches'] >= 8 and v['avg_us_under_pmc'] > 300}
        except Exception:
            pass
        res = {
            'metric': 'elbo_steps_per_sec', 'value': value, 'unit': 'ELBO steps/s (value+gradient, 1e6-row steps, fp64)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'dense zero-inflated GP ELBO step, N=%d rows/GPU, D=%d, M=%d per latent, full batch' % (N, D, M),
                       'rows_per_gpu': N, 'M': M, 'D': D, 'chunk_rows': args.chunk, 'jitter': jitter,
                       'parallelism': 'row-shard x%d, 1 all-reduce/step' % world},
            'elbo': elbo_gpu,
            'roofline': {'bound': 'mfma', 'achieved': achieved / 1e12, 'peak': PEAK_FP64_MFMA / 1e12, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP64_MFMA, 'traffic': traffic,
                         'kernel': PROF_KERNELS[dom],
                         'per_kernel_tflops': {k: (prof[k]['flops'] / (prof[k]['ms'] * 1e-3) / 1e12 if prof[k]['ms'] > 0 else 0.0) for k in gemm_classes},
                         'flops_per_launch': flops_per_launch, 'avg_launch_ms': avg_launch_s * 1e3,
                         'all_gemm_tflops': gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
                         # whole step against the same peak: flops this engine's algorithm needs (10 M^2 N: four
                         # triangular products + one symmetric rank-N update per latent) and, for reference, the
                         # 12 M^2 N of the literal reverse pass it replaces
                         'step_frac_10M2N': (10.0 * M * M * N / (dt / args.steps)) / PEAK_FP64_MFMA,
                         'step_frac_12M2N_literal': (12.0 * M * M * N / (dt / args.steps)) / PEAK_FP64_MFMA},
            'hbm_bound_kernels': hbm, 'mfma_busy_pmc': mfma_util,
            'kernel_ms_per_step': {k: v['est_total_ms'] / args.steps for k, v in prof.items()},   # avg of the timed launches x all launches
        }
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            srows = min(args.cpu_sample_rows, N)
            cdt, cdata = cpu_baseline(X, Y, p, jitter, srows, threads)
            gdata = eng.elbo(p, jitter=jitter, rows=(0, srows), include_kl=False, need_grad=False)[0]
            res['cpu_baseline'] = {'value': 1.0 / (cdt * N / srows), 'unit': 'ELBO steps/s (extrapolated to %d rows)' % N,
                                   'cores': threads, 'kind': 'port',
                                   'sample': 'oracle (torch CPU fp64, reference op order + autograd) on the first %d rows in 20000-row chunks: %.2f s' % (srows, cdt),
                                   'elbo_data_rel_diff_on_sample': abs(gdata - cdata) / abs(cdata)}
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
