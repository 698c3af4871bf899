"""scripts/hurdle.py:21-407: hurdle baseline = the classifier's hard on/off decision + a Gaussian Kronecker SVGP regression
trained ONLY on the points the classifier marks "on".  Reads <dir>/data.pickle and <dir>/results_scgp.pickle (written by
classifier.py), writes <dir>/model_hurdle.ckpt(.npz), <dir>/modelsumm_hurdle.log and <dir>/results_hurdle.pickle."""
import os
import pickle
import sys

import numpy as np

import zigp
from onofftf.heads import close_logger, fit_head, init_head_params, log_kernel_summary, open_logger
from onofftf.svgppred import predict_svgp


def rmse(predict, actual):
    return np.sqrt(np.mean((actual - np.maximum(predict, 0)) ** 2))      # :338-340


def mad(predict, actual):
    return np.mean(np.abs(actual - np.maximum(predict, 0)))             # :342-344


def hurdle(Xtrain, Ytrain, Xtest, Ytest, cresults, dir, num_iter=50000, num_inducing_f=(10, 100), num_minibatch=1000, device=0,
           engine=None, kmeans_seed=None, history=None):
    os.makedirs(dir, exist_ok=True)
    logger, handler = open_logger(os.path.join(dir, 'modelsumm_hurdle.log'))
    train_on = np.where(cresults['pred_train']['pfmean'] > 0.5)[0]                               # :49-50
    test_on = np.where(cresults['pred_test']['pfmean'] > 0.5)[0]
    Xtr, Ytr, Xte, Yte = Xtrain[train_on, :], Ytrain[train_on], Xtest[test_on, :], Ytest[test_on]   # :51-54
    logger.info('traning size   = ' + str(Xtrain.shape[0]))
    logger.info('test size   = ' + str(Xtest.shape[0]))
    # inducing inputs are initialised from ALL training inputs (:79-80), the data iterator holds the "on" subset (:57)
    pset = init_head_params(Xtrain, num_inducing_f, 'gaussian', kmeans_seed=kmeans_seed)
    eng = engine or zigp.reference_engine(device)      # tf.cholesky's acceptance rule (pivot > 0)
    ckpt = os.path.join(dir, 'model_hurdle.ckpt')
    fit_head(pset, 'gaussian', Xtr, Ytr, num_iter, min(num_minibatch, Xtr.shape[0]), logger, ckpt=ckpt, eng=eng, history=history)
    log_kernel_summary(logger, pset)
    ptr, pte = predict_svgp(Xtrain=Xtr, Xtest=Xte, checkpointPath=ckpt, num_inducing_f=np.array(num_inducing_f), engine=eng)   # :346-348
    res = {'pred_train_hurdle_svgp': ptr, 'pred_test_hurdle_svgp': pte,
           'train_hurdle_reg_rmse': rmse(ptr['fmean'], Ytr), 'train_hurdle_reg_mae': mad(ptr['fmean'], Ytr),
           'test_hurdle_reg_rmse': rmse(pte['fmean'], Yte), 'test_hurdle_reg_mae': mad(pte['fmean'], Yte)}
    # classifier decision everywhere, regression mean where it says "on"  (:361-366)
    comb_tr = (cresults['pred_train']['pfmean'] > 0.5) * 1.0
    comb_te = (cresults['pred_test']['pfmean'] > 0.5) * 1.0
    comb_tr[train_on] = ptr['fmean']
    comb_te[test_on] = pte['fmean']
    res.update({'train_pred_hurdle_comb': comb_tr, 'test_pred_hurdle_comb': comb_te,
                'train_hurdle_comb_rmse': rmse(comb_tr, Ytrain), 'train_hurdle_comb_mae': mad(comb_tr, Ytrain),
                'test_hurdle_comb_rmse': rmse(comb_te, Ytest), 'test_hurdle_comb_mae': mad(comb_te, Ytest),
                'train_pred_on_idx': train_on, 'test_pred_on_idx': test_on})
    for k in ('train_hurdle_reg_rmse', 'train_hurdle_reg_mae', 'test_hurdle_reg_rmse', 'test_hurdle_reg_mae',
              'train_hurdle_comb_rmse', 'train_hurdle_comb_mae', 'test_hurdle_comb_rmse', 'test_hurdle_comb_mae'):
        logger.info('%s for hurdle svgp : %s' % (k, res[k]))
    close_logger(logger, handler)
    with open(os.path.join(dir, 'results_hurdle.pickle'), 'wb') as f:
        pickle.dump(res, f)                                                                      # :385-402
    return res


def main(scriptPath, **kw):
    dir = os.path.dirname(os.path.realpath(scriptPath))
    with open(os.path.join(dir, 'data.pickle'), 'rb') as f:
        data = pickle.load(f)
    with open(os.path.join(dir, 'results_scgp.pickle'), 'rb') as f:
        cresults = pickle.load(f)
    return hurdle(data['Xtrain'], data['Ytrain'], data['Xtest'], data['Ytest'], cresults, dir, **kw)


if __name__ == '__main__':
    main(sys.argv[0])
