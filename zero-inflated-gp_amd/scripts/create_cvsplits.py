"""scripts/create_cvsplits.py:1-34: 5-fold cross-validation splits of the precipitation data (train+test pooled, time
column / 1000, KFold(n_splits=5, shuffle=True, random_state=1234)), one data/cv/<fold>/data.pickle per fold."""
import os
import pickle

import numpy as np


def create_cvsplits(pickle_path='data/pptr.pickle', out_root='data/cv', n_splits=5, random_state=1234):
    from sklearn.model_selection import KFold
    with open(pickle_path, 'rb') as f:
        data = pickle.load(f)
    Xraw = np.concatenate([data['Xtrain'], data['Xtest']])                    # :15-16
    Yraw = np.concatenate([data['Ytrain'], data['Ytest']])
    Xraw[:, 2] = Xraw[:, 2] / 1000                                            # :17
    dirs = []
    for nfold, (tr, te) in enumerate(KFold(n_splits=n_splits, random_state=random_state, shuffle=True).split(Xraw), start=1):   # :19-22
        fold = {'Xtrain': Xraw[tr], 'Ytrain': Yraw[tr], 'Xtest': Xraw[te], 'Ytest': Yraw[te]}
        print(fold['Ytrain'].shape, fold['Ytest'].shape)
        d = os.path.join(out_root, str(nfold))
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'data.pickle'), 'wb') as f:
            pickle.dump(fold, f)                                              # :33-34
        dirs.append(d)
    return dirs


if __name__ == '__main__':
    create_cvsplits()
