"""`onoff(Xtrain, Ytrain, Xtest, Ytest, dir)` -- the Kronecker zero-inflated GP fit of scripts/onoff.py:22-500 on the
MI355X engine.  Same defaults (50 000 Adam iterations, minibatch 1000, inducing grid [10,100], jitter 1e-5), same
return dict; the TensorBoard summaries / plots of the reference are not reproduced (out of scope).
The training loop runs on the device (`_device_loop`, zigp_kron_fit_steps: one host synchronisation per 200 iterations);
`device_loop=False` steps it from the host with zigp.optim.AdamGroups, one call per iteration (the checker of the device loop)."""
import logging
import os
import time

import numpy as np

import zigp
from zigp.optim import AdamGroups
from .main import DataSet
from .model import init_params, engine_params, named_grads, save_checkpoint

jitter_level = 1e-5   # scripts/onoff.py:18


def _device_loop(eng, pset, train_data, num_iter, num_minibatch, scale, log_every, save_every, dir, logger, history):
    """scripts/onoff.py:375-431 with the loop body on the device (zigp_kron_fit_steps): the iterations between two log lines (:418, every
    200) are ONE call -- gradient, Log1pe chain and the per-learning-rate Adam update of every step run back to back on the GPU, the host
    synchronises once per call.  The batch sequence is DataSet's (onofftf/main.py:98-133): the permuted epoch is resident, a call ends at
    the one wrap-around batch of an epoch (its rows go down with the call), at a checkpoint iteration, or after log_every steps."""
    from .model import KronDeviceFit
    fitter = KronDeviceFit(eng, pset)
    resident, i = None, 0
    while i < num_iter:
        n_target = min(num_iter - i, log_every - (i % log_every))
        if save_every:
            n_target = min(n_target, save_every - (i % save_every) if i % save_every else 1)       # a checkpoint iteration ends its call
        t0 = time.time()
        rbs, wrap = [], None
        while len(rbs) < n_target:
            gen, lo, hi, wrap = train_data.next_span(num_minibatch)
            if wrap is not None:
                rbs.append(-1)
                break
            if gen != resident:              # only ever at the start of a call: a generation changes right after a wrap-around batch
                eng.set_data(train_data.xtrain, train_data.ytrain)
                resident = gen
            rbs.append(lo)
        ed, kl = fitter.steps(rbs, num_minibatch, jitter_level, scale, *(wrap if wrap is not None else (None, None)))
        if history is not None:
            history.extend((-(ed - kl)).tolist())
        per_it = (time.time() - t0) / len(rbs)
        for j in range(i, i + len(rbs)):
            if j % log_every == 0:
                logger.info('{:>16d}'.format(j) + '{:>6.3f}'.format(per_it / 60))
            if save_every and j % save_every == 0 and dir:
                save_checkpoint(pset, os.path.join(dir, 'model'))
        i += len(rbs)


def onoff(Xtrain, Ytrain, Xtest, Ytest, dir, num_iter=50000, num_inducing_f=(10, 100), num_inducing_g=(10, 100),
          num_minibatch=1000, log_every=200, save_every=10000, device=0, engine=None, kmeans_seed=None, history=None, device_loop=True):
    os.makedirs(dir, exist_ok=True) if dir else None
    logger = logging.getLogger('log')                                                # :35-40
    logger.setLevel(logging.DEBUG)
    handler = logging.FileHandler(os.path.join(dir, 'modelsumm.log')) if dir else logging.NullHandler()
    logger.addHandler(handler)
    logger.info('traning size   = ' + str(Xtrain.shape[0]))
    logger.info('test size   = ' + str(Xtest.shape[0]))
    train_data = DataSet(Xtrain, Ytrain)                                             # :43
    num_data = Xtrain.shape[0]                                                       # :54
    pset = init_params(Xtrain, num_inducing_f, num_inducing_g, init_noisevar=0.01, kmeans_seed=kmeans_seed)   # :51-137
    eng = engine or zigp.reference_engine(device)      # tf.cholesky's acceptance rule (pivot > 0)
    scale = float(num_data) / float(num_minibatch)                                   # :311
    logger.info('*******  started optimization at ' + time.strftime('%Y%m%d-%H%M') + ' *******')
    # the device loop covers the grids of the fused kernels (<= 32 x <= 32, <= 16 x <= 112: the reference's [10, 100] and BASELINE's
    # 32 x 32); anything larger is stepped from the host, one call per iteration, as in rounds 1-3
    mf, mg = tuple(int(q) for q in num_inducing_f), tuple(int(q) for q in num_inducing_g)
    m0, m1 = max(mf[0], mg[0]), max(mf[1], mg[1])
    if device_loop and ((m0 <= 32 and m1 <= 32) or (m0 <= 16 and m1 <= 112)) and num_minibatch <= num_data:
        try:
            _device_loop(eng, pset, train_data, num_iter, num_minibatch, scale, log_every, save_every, dir, logger, history)
        except KeyboardInterrupt:
            print('Stopping training')
        num_iter = 0
    opt = AdamGroups(pset) if num_iter else None                                     # :325-350 (one Adam per learning rate)
    resident = None     # generation of the epoch order whose arrays are resident in HBM
    step = eng.kron_stepper(engine_params(pset)) if num_iter else None   # one model shape for the whole fit: buffers and structs prepared once
    for i in range(num_iter):                                                        # :375-431
        t0 = time.time()
        # DataSet.next_batch shuffles once per epoch and then slices (onofftf/main.py:98-133): the permuted epoch goes to the GPU once,
        # every batch but the one wrap-around per epoch is a row range of it (zigp_kron_elbo_rows: nothing is staged per step)
        gen, lo, hi, wrap = train_data.next_span(num_minibatch)
        try:
            if wrap is None:
                if gen != resident:
                    eng.set_data(train_data.xtrain, train_data.ytrain)
                    resident = gen
                ed, kl, g = step(engine_params(pset), rows=(lo, hi), jitter=jitter_level, scale=scale)
            else:
                ed, kl, g = step(engine_params(pset), wrap[0], wrap[1], jitter=jitter_level, scale=scale)
            opt.step(named_grads(g))                                                 # minimises cost = -(var_exp*scale - kl), :318
            if history is not None:
                history.append(-(ed - kl))
            if i % log_every == 0:
                logger.info('{:>16d}'.format(i) + '{:>6.3f}'.format((time.time() - t0) / 60))
            if save_every and i % save_every == 0 and dir:
                save_checkpoint(pset, os.path.join(dir, 'model'))
        except KeyboardInterrupt:
            print('Stopping training')
            break
    if dir:
        save_checkpoint(pset, os.path.join(dir, 'model'))
    v = {k: q.value for k, q in pset.params.items()}
    logger.info('Noise variance          = ' + str(v['likelihood/variance']))
    for tag, nm in (('f', 'Kf'), ('g', 'Kg')):
        logger.info('%s spatial lengthscale  = %s' % (nm, v['%s_kern/lengthscale_0' % tag]))
        logger.info('%s spatial variance     = %s' % (nm, v['%s_kern/variance_0' % tag]))
        logger.info('%s temporal lengthscale = %s' % (nm, v['%s_kern/lengthscale_1' % tag]))
        logger.info('%s temporal variance    = %s' % (nm, v['%s_kern/variance_1' % tag]))
    # test predictions with the TRAINING graph (jitter 1e-5, no gmean shift), clipped at 0  (:466-481)
    pred_test = np.maximum(eng.kron_predict(engine_params(pset), Xtest, jitter=jitter_level, g_offset=0.0)[0].reshape(-1, 1), 0)
    test_rmse = np.sqrt(np.mean((pred_test - Ytest) ** 2))
    test_mae = np.mean(np.abs(pred_test - Ytest))
    logger.info('test rmse:' + str(test_rmse))
    logger.info('test mae:' + str(test_mae))
    logger.removeHandler(handler)
    return {'Xtrain': Xtrain, 'Ytrain': Ytrain, 'Xtest': Xtest, 'Ytest': Ytest, 'test_rmse': test_rmse, 'test_mae': test_mae}   # :487-500
