"""`svgp(Xtrain, Ytrain, Xtest, Ytest, dir)` -- scripts/svgp.py:22-404: Gaussian-likelihood Kronecker SVGP baseline
(50 000 Adam iterations, minibatch 1000, inducing grid [10,100], jitter 1e-5); same return dict.  TensorBoard summaries and
the inducing-point monitoring plots (:254-286,303-326) are not reproduced."""
import os

import numpy as np

import zigp
from onofftf.heads import (TRAIN_JITTER, close_logger, fit_head, head_engine_params, init_head_params, log_kernel_summary,
                           open_logger)

jitter_level = TRAIN_JITTER   # scripts/svgp.py:18


def svgp(Xtrain, Ytrain, Xtest, Ytest, dir, num_iter=50000, num_inducing_f=(10, 100), num_minibatch=1000, device=0, engine=None,
         kmeans_seed=None, history=None):
    if dir:
        os.makedirs(dir, exist_ok=True)
    logger, handler = open_logger(os.path.join(dir, 'modelsumm.log') if dir else None)          # :30-39
    logger.info('traning size   = ' + str(Xtrain.shape[0]))
    logger.info('test size   = ' + str(Xtest.shape[0]))
    logger.info('number of training examples:' + str(Xtrain.shape))
    pset = init_head_params(Xtrain, num_inducing_f, 'gaussian', kmeans_seed=kmeans_seed)       # :51-112
    eng = engine or zigp.reference_engine(device)      # tf.cholesky's acceptance rule (pivot > 0)
    fit_head(pset, 'gaussian', Xtrain, Ytrain, num_iter, num_minibatch, logger, ckpt=os.path.join(dir, 'model') if dir else None,
             eng=eng, history=history)                                                          # :289-334
    log_kernel_summary(logger, pset)                                                            # :337-345
    # test predictions from the TRAINING graph (jitter 1e-5), clipped at 0  (:377-386)
    fmean = eng.kron_head_predict(head_engine_params(pset), Xtest, 'gaussian', jitter=jitter_level)[0].reshape(-1, 1)
    pred_test = np.maximum(fmean, 0)
    test_rmse = np.sqrt(np.mean((pred_test - Ytest) ** 2))
    test_mae = np.mean(np.abs(pred_test - Ytest))
    logger.info('test rmse:' + str(test_rmse))
    logger.info('test mae:' + str(test_mae))
    close_logger(logger, handler)
    return {'Xtrain': Xtrain, 'Ytrain': Ytrain, 'Xtest': Xtest, 'Ytest': Ytest, 'test_rmse': test_rmse, 'test_mae': test_mae}   # :391-404
