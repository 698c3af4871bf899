"""scripts/classifier.py:22-397: the on/off (rain / no rain) Kronecker GP classifier of the hurdle and zero-inflated
baselines.  `main(scriptPath)` keeps the reference's file protocol: reads <dir>/data.pickle, trains on (Y > 0), writes
<dir>/model_scgp.ckpt(.npz), <dir>/modelsumm_scgp.log and <dir>/results_scgp.pickle, where <dir> is the folder holding the
script copy (classifier.py:24-25).  `classifier(...)` is the same computation on arrays."""
import os
import pickle
import sys

import numpy as np

import zigp
from onofftf.heads import close_logger, fit_head, init_head_params, log_kernel_summary, open_logger
from onofftf.svcppred import predict_scgp


def _scores(prob, actual):
    """accuracy / precision / recall at the 0.5 cut and AUC  (classifier.py:337-350; sklearn.metrics definitions)"""
    actual = np.asarray(actual).reshape(-1).astype(bool)
    prob = np.asarray(prob).reshape(-1)
    hard = prob > 0.5
    tp, fp, fn = np.sum(hard & actual), np.sum(hard & ~actual), np.sum(~hard & actual)
    acc = float(np.mean(hard == actual))
    prec = float(tp) / float(tp + fp) if tp + fp else 0.0
    rec = float(tp) / float(tp + fn) if tp + fn else 0.0
    # AUC = Mann-Whitney U / (n_pos * n_neg) with average ranks for ties
    from scipy.stats import rankdata
    r = rankdata(prob)
    npos, nneg = int(actual.sum()), int((~actual).sum())
    auc = (r[actual].sum() - npos * (npos + 1) / 2.0) / (npos * nneg) if npos and nneg else float('nan')
    return acc, prec, rec, float(auc)


def classifier(Xtrain, Ytrain, Xtest, Ytest, dir, num_iter=500, num_inducing_f=(10, 100), num_minibatch=1000, include_f_mu=False,
               device=0, engine=None, kmeans_seed=None, history=None):
    os.makedirs(dir, exist_ok=True)
    Ytrain_c, Ytest_c = (Ytrain > 0) * 1.0, (Ytest > 0) * 1.0                                    # :43-47
    logger, handler = open_logger(os.path.join(dir, 'modelsumm_scgp.log'))
    logger.info('traning size   = ' + str(Xtrain.shape[0]))
    logger.info('test size   = ' + str(Xtest.shape[0]))
    pset = init_head_params(Xtrain, num_inducing_f, 'bernoulli', include_f_mu=include_f_mu, kmeans_seed=kmeans_seed)   # :56-112
    eng = engine or zigp.reference_engine(device)      # tf.cholesky's acceptance rule (pivot > 0)
    ckpt = os.path.join(dir, 'model_scgp.ckpt')
    fit_head(pset, 'bernoulli', Xtrain, Ytrain_c, num_iter, num_minibatch, logger, ckpt=ckpt, eng=eng, history=history)   # :276-321
    log_kernel_summary(logger, pset)
    pred_train, pred_test = predict_scgp(Xtrain=Xtrain, Xtest=Xtest, checkpointPath=ckpt, num_inducing_f=np.array(num_inducing_f),
                                         include_f_mu=include_f_mu, engine=eng)                  # :352-354
    results = {'pred_train': pred_train, 'pred_test': pred_test}
    for split, pred, truth in (('train', pred_train, Ytrain_c), ('test', pred_test, Ytest_c)):
        acc, prec, rec, auc = _scores(pred['pfmean'], truth)
        for name, val in (('accuracy', acc), ('precision', prec), ('recall', rec), ('auc', auc)):
            logger.info('%s on %s set for scgp : %s' % (name, split, val))                       # :356-372
            results['%s_%s' % (split, name)] = val
    close_logger(logger, handler)
    with open(os.path.join(dir, 'results_scgp.pickle'), 'wb') as f:
        pickle.dump(results, f)                                                                  # :380-392
    return results


def main(scriptPath, **kw):
    dir = os.path.dirname(os.path.realpath(scriptPath))
    with open(os.path.join(dir, 'data.pickle'), 'rb') as f:
        data = pickle.load(f)
    return classifier(data['Xtrain'], data['Ytrain'], data['Xtest'], data['Ytest'], dir, **kw)


if __name__ == '__main__':
    main(sys.argv[0])
