"""scripts/zero_inflated.py:26-119: zero-inflated baseline = classifier output x regression mean, both as probabilities
("prob") and as the hard 0.5 cut ("indc").  Reads <dir>/data.pickle, <dir>/results_scgp.pickle (classifier.py) and
<dir>/results_svgp.pickle ({'pred_train': {'fmean'}, 'pred_test': {'fmean'}} of a regression fit), writes
<dir>/modelsumm_zi.log and <dir>/results_zi.pickle.  Host-side post-processing only."""
import os
import pickle
import sys

import numpy as np

from onofftf.heads import close_logger, open_logger


def rmse(predict, actual):
    return np.sqrt(np.mean((actual - np.maximum(predict, 0)) ** 2))      # :64-66


def mad(predict, actual):
    return np.mean(np.abs(actual - np.maximum(predict, 0)))             # :68-70


def zero_inflated(Ytrain, Ytest, clf_results, reg_results, dir=None):
    logger, handler = open_logger(os.path.join(dir, 'modelsumm_zi.log') if dir else None)
    p_tr, p_te = clf_results['pred_train']['pfmean'], clf_results['pred_test']['pfmean']        # :55-56
    f_tr, f_te = reg_results['pred_train']['fmean'], reg_results['pred_test']['fmean']
    res = {'pred_train_zi_prob': p_tr * f_tr, 'pred_test_zi_prob': p_te * f_te,                  # :60-61
           'pred_train_zi_indc': (p_tr > 0.5) * 1.0 * f_tr, 'pred_test_zi_indc': (p_te > 0.5) * 1.0 * f_te}   # :57-58,62-63
    for kind in ('prob', 'indc'):
        for split, truth in (('train', Ytrain), ('test', Ytest)):
            pred = res['pred_%s_zi_%s' % (split, kind)]
            res['%s_zi_%s_reg_rmse' % (split, kind)] = rmse(pred, truth)
            res['%s_zi_%s_reg_mae' % (split, kind)] = mad(pred, truth)
            logger.info('rmse on %s set for zi %s : %s' % (split, kind, res['%s_zi_%s_reg_rmse' % (split, kind)]))
            logger.info('mae on %s set for zi %s : %s' % (split, kind, res['%s_zi_%s_reg_mae' % (split, kind)]))
    close_logger(logger, handler)
    if dir:
        with open(os.path.join(dir, 'results_zi.pickle'), 'wb') as f:
            pickle.dump(res, f)                                                                  # :99-114
    return res


def main(scriptPath):
    dir = os.path.dirname(os.path.realpath(scriptPath))
    load = lambda name: pickle.load(open(os.path.join(dir, name), 'rb'))
    data = load('data.pickle')
    return zero_inflated(data['Ytrain'], data['Ytest'], load('results_scgp.pickle'), load('results_svgp.pickle'), dir)


if __name__ == '__main__':
    main(sys.argv[0])
