"""Generates tests/golden/*.npz.  Run in the build container (needs /root/reference for G1 only).

G1  g1_kernse_np.npz : outputs of the REFERENCE's own NumPy kernel `kernse_np` (onofftf/utils.py:26-58,
    imported from /root/reference -- NumPy only, no TensorFlow needed) on seeded inputs.  Pins the RBF kernel.
G2  g2_dense_oracle.npz : outputs of this repo's CPU restatement (oracle/zigp_oracle*.py) on seeded
    inputs: the 9-tuple, KL_f, KL_g, ELBO and the full gradient.  NOT produced by the reference
    (TensorFlow/GPflow are not installable): these fixtures freeze the oracle so that a later edit
    cannot silently change it; they do not pin it to the reference ("parity unpinned").
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


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    g1()
    g2()
    print(os.listdir(OUT))
