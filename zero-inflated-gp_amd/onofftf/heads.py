"""Single-latent Kronecker SVGP models shared by the reference's baselines: the parameter set that scripts/svgp.py:51-112,
scripts/hurdle.py:64-124 and scripts/classifier.py:56-112 declare (TF scopes f_kern/, likelihood/, f_ind/), the Adam fit loop
of svgp.py:240-330 and the restore-and-predict step of onofftf/svgppred.py / onofftf/svcppred.py -- on libzigp's
zigp_kron_head_elbo / zigp_kron_head_predict (include/zigp.h)."""
import logging
import os
import time
from collections import OrderedDict

import numpy as np

import zigp
from zigp.optim import AdamGroups, ParamSet
from zigp.transforms import Log1pe, positive
from .main import DataSet, Param
from .model import load_checkpoint, save_checkpoint

TRAIN_JITTER = 1e-5     # scripts/svgp.py:18, classifier.py:19, hurdle.py:18
PREDICT_JITTER = 1e-6   # onofftf/svgppred.py:13, onofftf/svcppred.py:13


def init_head_params(Xtrain, num_inducing_f, lik, init_ell=(5., 5.), u_scale=0.01, init_noisevar=0.01, include_f_mu=False,
                     kern_lr=1e-3, indp_lr=1e-3, rng=None, kmeans_seed=None):
    """svgp.py:51-112 (ell [5,5],[5/1000]; var 20; noise 0.01; u 0.01*randn; s 1); the predictors rebuild the same set with
    ell [8,8], u 0.1*randn, noise 0.001 before restoring (svgppred.py:21-37) -- those values are overwritten by the restore."""
    from scipy.cluster.vq import kmeans
    rng = rng or np.random
    M0, M1 = int(num_inducing_f[0]), int(num_inducing_f[1])
    Zs = kmeans(Xtrain[:, 0:2], M0, seed=kmeans_seed)[0]                          # svgp.py:64
    if Zs.shape[0] < M0:                                                          # kmeans may return fewer centroids
        Zs = np.vstack([Zs, Xtrain[rng.choice(Xtrain.shape[0], M0 - Zs.shape[0], replace=False), 0:2] + 1e-3])
    Zt = np.linspace(Xtrain[:, 2].min(), Xtrain[:, 2].max(), M1)[:, None]        # :65
    ells = [np.array(init_ell, dtype=np.float64), np.array([5. / 1000])]          # :58
    p = OrderedDict()
    for i in range(2):
        p['f_kern/lengthscale_%d' % i] = Param(ells[i], Log1pe(), name='lengthscale', learning_rate=kern_lr)
        p['f_kern/variance_%d' % i] = Param([20.], Log1pe(), name='variance', learning_rate=kern_lr)      # :59
    if lik == 'gaussian':
        p['likelihood/variance'] = Param(init_noisevar, Log1pe(), name='variance', learning_rate=kern_lr)  # :93-95
    if include_f_mu:
        p['f_mu'] = Param(0., name='fmu', learning_rate=indp_lr)                                           # classifier.py:70-72
    p['f_ind/z_0'] = Param(Zs.copy(), name='z', learning_rate=indp_lr)
    p['f_ind/z_1'] = Param(Zt.copy(), name='z', learning_rate=indp_lr)
    p['f_ind/value'] = Param(rng.randn(M0 * M1, 1) * u_scale, name='value', learning_rate=indp_lr)         # :68
    p['f_ind/variance'] = Param(np.ones((M0 * M1, 1)), positive, name='variance', learning_rate=indp_lr)   # :69,104-106
    return ParamSet(p)


def head_engine_params(pset):
    v = {k: q.value for k, q in pset.params.items()}
    out = dict(Zf=[v['f_ind/z_0'], v['f_ind/z_1']], ell_f=[v['f_kern/lengthscale_0'], v['f_kern/lengthscale_1']],
               var_f=[v['f_kern/variance_0'], v['f_kern/variance_1']], u_fm=v['f_ind/value'], u_fs_sqrt=v['f_ind/variance'])
    if 'likelihood/variance' in v:
        out['noise'] = v['likelihood/variance']
    return out


def head_f_mu(pset):
    return float(pset.params['f_mu'].value.reshape(-1)[0]) if 'f_mu' in pset.params else 0.0


def named_head_grads(g):
    out = {'likelihood/variance': np.array([g['noise']]), 'f_mu': np.array([g['f_mu']]),
           'f_ind/value': np.asarray(g['u_fm']), 'f_ind/variance': np.asarray(g['u_fs_sqrt'])}
    for i in range(2):
        out['f_kern/lengthscale_%d' % i] = np.asarray(g['ell_f'][i])
        out['f_kern/variance_%d' % i] = np.array([g['var_f'][i]])
        out['f_ind/z_%d' % i] = np.asarray(g['Zf'][i])
    return out


def fit_head(pset, lik, Xtrain, Ytrain, num_iter, num_minibatch, logger, ckpt=None, eng=None, save_every=10000, history=None):
    """The optimisation loop of svgp.py:289-330 / classifier.py:276-316: Adam per learning-rate group on
    cost = -(sum(var_exp) * num_data / num_minibatch - kl)."""
    train_data = DataSet(Xtrain, Ytrain)                                          # svgp.py:43
    scale = float(Xtrain.shape[0]) / float(num_minibatch)                         # :212
    opt = AdamGroups(pset)                                                        # :225-252
    logger.info('*******  started optimization at ' + time.strftime('%Y%m%d-%H%M') + ' *******')
    logger.info('{:>16s}'.format('iteration') + '{:>6s}'.format('time'))
    for i in range(num_iter):
        t0 = time.time()
        xb, yb = train_data.next_batch(num_minibatch)
        try:
            ed, kl, g = eng.kron_head_elbo(head_engine_params(pset), xb, yb, lik, jitter=TRAIN_JITTER, scale=scale, f_mu=head_f_mu(pset))
            opt.step(named_head_grads(g))
            if history is not None:
                history.append(-(ed - kl))
            if i % 100 == 0:
                logger.info('{:>16d}'.format(i) + '{:>6.3f}'.format((time.time() - t0) / 60))
            if ckpt and save_every and i % save_every == 0:
                save_checkpoint(pset, ckpt)
        except KeyboardInterrupt:
            print('Stopping training')
            break
    if ckpt:
        save_checkpoint(pset, ckpt)
    return pset


def log_kernel_summary(logger, pset):
    v = {k: q.value for k, q in pset.params.items()}
    if 'likelihood/variance' in v:
        logger.info('Noise variance          = ' + str(v['likelihood/variance']))
    logger.info('Kf spatial lengthscale  = ' + str(v['f_kern/lengthscale_0']))
    logger.info('Kf spatial variance     = ' + str(v['f_kern/variance_0']))
    logger.info('Kf temporal lengthscale = ' + str(v['f_kern/lengthscale_1']))
    logger.info('Kf temporal variance    = ' + str(v['f_kern/variance_1']))


def open_logger(path):
    logger = logging.getLogger('log')
    logger.setLevel(logging.DEBUG)
    handler = logging.FileHandler(path) if path else logging.NullHandler()
    logger.addHandler(handler)
    return logger, handler


def close_logger(logger, handler):
    handler.close()
    logger.removeHandler(handler)


def restore_and_predict(lik, Xtrain, Xtest, checkpointPath, num_inducing_f, include_f_mu, rows, device=0, engine=None):
    """svgppred.py:15-203 / svcppred.py:15-224: rebuild the parameter set, restore, evaluate with jitter 1e-6."""
    pset = init_head_params(Xtrain, num_inducing_f, lik, init_ell=(8., 8.), u_scale=0.1, init_noisevar=0.001,
                            include_f_mu=include_f_mu, kern_lr=1e-4, indp_lr=1e-4)
    ck = os.path.join(checkpointPath, 'model') if os.path.isdir(checkpointPath) else checkpointPath
    load_checkpoint(pset, ck)
    eng = engine or zigp.reference_engine(device)      # tf.cholesky's acceptance rule (pivot > 0)
    p, f_mu = head_engine_params(pset), head_f_mu(pset)

    def run(X):
        o = eng.kron_head_predict(p, X, lik, jitter=PREDICT_JITTER, f_mu=f_mu)
        return {name: o[r].reshape(-1, 1) for name, r in rows}

    pred_train = run(Xtrain)
    if Xtest is not None:
        return pred_train, run(Xtest)
    return pred_train
