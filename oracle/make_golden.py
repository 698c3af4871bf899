"""Generates tests/golden/*.npz.  Run in the build container (needs /root/reference for G1 only).

G1  g1_kernse_np.npz : outputs of the REFERENCE's own NumPy kernel `kernse_np` (onofftf/utils.py:26-58,
    imported from /root/reference -- NumPy only, no TensorFlow needed) on seeded inputs.  Pins the RBF kernel.
G2  g2_dense_oracle.npz : outputs of this repo's CPU restatement (oracle/zigp_oracle*.py) on seeded
    inputs: the 9-tuple, KL_f, KL_g, ELBO and the full gradient.  NOT produced by the reference
    (TensorFlow/GPflow are not installable): these fixtures freeze the oracle so that a later edit
    cannot silently change it; they do not pin it to the reference ("parity unpinned").
G4  g4_kron_oracle.npz : this repo's literal Kronecker restatements (on/off, Gaussian and Bernoulli heads) frozen on a seeded
    minibatch; like G2 it guards the oracle against drift, it does not pin it to the reference.
G3  g3_utils_pptr.npz : outputs of the REFERENCE's own `preprocessing` class (onofftf/utils_pptr.py, NumPy/pandas only,
    imported in place) on a seeded synthetic station table: time filter, location/time scaling, kernel_params.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, HERE)


def g1():
    sys.path.insert(0, '/root/reference')
    from onofftf.utils import kernse_np   # the reference's own code, imported in place
    rs = np.random.RandomState(1234)
    out = {}
    for tag, D, ell in (('d1', 1, np.array([2.0])), ('d3s', 3, np.array([0.3])), ('d3ard', 3, np.array([0.2, 0.5, 1.1]))):
        Z, X = rs.rand(23, D) * 4, rs.rand(61, D) * 4
        var = np.array(1.0 + rs.rand() * 4)
        k = kernse_np(ell, var)
        out.update({tag + '_Z': Z, tag + '_X': X, tag + '_ell': ell, tag + '_var': var,
                    tag + '_Kzx': k.K(Z, X), tag + '_Kzz': k.K(Z), tag + '_Kdiag': k.Kdiag(X)})
    np.savez_compressed(os.path.join(OUT, 'g1_kernse_np.npz'), **out)


def g2():
    import zigp_oracle as o
    import zigp_oracle_torch as ot
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
    from conftest import make_problem
    out = {}
    for tag, (N, M, D, ell) in (('toy', (120, 9, 1, 2.0)), ('d3', (256, 40, 3, 0.3))):
        X, Y, p = make_problem(N, M, D, seed=42, ell=ell)
        pred = o.build_predict(X, p, 1e-6, 0.0)
        elbo, data, kl, g = ot.elbo_and_grad(X, Y, p, 1e-6, scale=1.3)
        klf, klg = o.build_prior_KL(p, 1e-6)
        out.update({tag + '_X': X, tag + '_Y': Y, tag + '_pred': np.stack([q.reshape(-1) for q in pred]),
                    tag + '_elbo': elbo, tag + '_data': data, tag + '_klf': klf, tag + '_klg': klg})
        for k, v in p.items():
            out[tag + '_p_' + k] = np.asarray(v)
        for k, v in g.items():
            out[tag + '_g_' + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, 'g2_dense_oracle.npz'), **out)


def pptr_like_table(seed=7, ntr=300, nte=90):
    """seeded stand-in for the precipitation table: lat, lon, ndatehour columns + X arrays in that column order"""
    import pandas as pd
    rs = np.random.RandomState(seed)

    def mk(n):
        df = pd.DataFrame({'lat': 60 + 8 * rs.rand(n), 'lon': 20 + 10 * rs.rand(n), 'ndatehour': rs.randint(0, 400, n).astype(float)})
        X = df[['lat', 'lon', 'ndatehour']].values.copy()
        Y = np.where(rs.rand(n) > 0.6, 3 * rs.rand(n), 0.0)[:, None]
        return df, X, Y
    tr, te = mk(ntr), mk(nte)
    return {'traindf': tr[0], 'testdf': te[0], 'Xtrain': tr[1], 'Ytrain': tr[2], 'Xtest': te[1], 'Ytest': te[2]}


def g3():
    sys.path.insert(0, '/root/reference')
    from onofftf.utils_pptr import preprocessing   # the reference's own code, imported in place
    out = {}
    for tag, filt, loc, tim in (('raw', False, False, False), ('filt', True, False, False), ('loc', False, True, False), ('all', True, True, True)):
        pp = preprocessing(pptr_like_table())
        if filt:
            pp.filter_time(50, 320)
        if loc or tim:
            pp.scale(scale_loc=loc, scale_time=tim)
        md = pp.model_data
        var, ell = pp.kernel_params
        out.update({tag + '_Xtrain': md['Xtrain'], tag + '_Xtest': md['Xtest'], tag + '_Ytrain': md['Ytrain'], tag + '_Ytest': md['Ytest'],
                    tag + '_traindf': md['traindf'][['lat', 'lon', 'ndatehour']].values, tag + '_shape': np.array(pp.shape),
                    tag + '_kvar': np.array(var), tag + '_kell': np.array(ell)})
        if loc or tim:
            for c, v in pp.scale_param.items():
                out['%s_sp_%s' % (tag, c)] = np.array([v['min'], v['range']])
    np.savez_compressed(os.path.join(OUT, 'g3_utils_pptr.npz'), **out)


def kron_problem(seed=3, N=180, M0=5, M1=6):
    """seeded pptr-like minibatch + Kronecker parameter set (2 spatial columns, 1 temporal)"""
    rs = np.random.RandomState(seed)
    X = np.hstack([rs.rand(N, 2) * 10.0, rs.rand(N, 1)])
    Y = np.where(rs.rand(N) > 0.6, np.abs(np.sin(X[:, 0]) + 0.3 * rs.randn(N)), 0.0)[:, None]
    p = dict(Zf=[rs.rand(M0, 2) * 10.0, np.linspace(0, 1, M1)[:, None]], Zg=[rs.rand(M0, 2) * 10.0, np.linspace(0, 1, M1)[:, None]],
             ell_f=[np.array([3.0, 3.6]), np.array([0.3])], ell_g=[np.array([2.4, 3.0]), np.array([0.45])],
             var_f=[np.array([2.0]), np.array([1.5])], var_g=[np.array([1.2]), np.array([0.9])],
             u_fm=0.1 * rs.randn(M0 * M1, 1), u_gm=0.1 * rs.randn(M0 * M1, 1),
             u_fs_sqrt=0.5 + rs.rand(M0 * M1, 1), u_gs_sqrt=0.5 + rs.rand(M0 * M1, 1), noise=0.05)
    return X, Y, p


def g4():
    """G4 g4_kron_oracle.npz : this repo's LITERAL Kronecker restatements (scripts/onoff.py:143-319, svgp.py, classifier.py) frozen on a
    seeded minibatch: on/off ELBO + 9-tuple, Gaussian and Bernoulli head ELBOs + predictions.  Not produced by the reference."""
    import zigp_oracle as o
    X, Y, p = kron_problem()
    out = {}
    e, d, klf, klg = o.kron_elbo(X, Y, p, 1e-5, scale=4.0, g_offset=0.0)
    out.update(onoff_elbo=e, onoff_data=d, onoff_klf=klf, onoff_klg=klg,
               onoff_pred=np.stack([q.reshape(-1) for q in o.kron_build_predict(X, p, 1e-6, -1.0)]))
    ph = {k: p[k] for k in ('Zf', 'ell_f', 'var_f', 'u_fm', 'u_fs_sqrt', 'noise')}
    for lik, Yl in (('gaussian', Y), ('bernoulli', (Y > 0) * 1.0)):
        e, d, kl = o.kron_head_elbo(X, Yl, ph, lik, 1e-5, scale=4.0, f_mu=0.2)
        out.update({lik + '_elbo': e, lik + '_data': d, lik + '_kl': kl,
                    lik + '_pred': np.stack([np.asarray(q).reshape(-1) for q in o.kron_head_predict(X, ph, lik, 1e-6, 0.2)])})
    np.savez_compressed(os.path.join(OUT, 'g4_kron_oracle.npz'), **out)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    g1()
    g2()
    g3()
    g4()
    print(os.listdir(OUT))
