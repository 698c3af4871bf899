"""Torch (CPU, float64) twin of oracle/zigp_oracle.py with reverse-mode gradients -- TEST INFRASTRUCTURE ONLY.

Same op order as the NumPy oracle (and therefore as the reference); gradients come from
torch.autograd exactly as the reference gets them from tf.gradients (scripts/onoff.py:334) /
GPflow's optimiser.  Used (i) by tests as the gradient checker, (ii) by bench.py's
`cpu_baseline` leg (kind "port": the reference's TensorFlow/GPflow stack is not installable).
PARITY STATUS: parity unpinned (see zigp_oracle.py header).  Never imported by the product path.
"""
import math
import numpy as np
import torch

DT = torch.float64
PARAM_KEYS = ('Zf', 'Zg', 'u_fm', 'u_gm', 'u_fs_sqrt', 'u_gs_sqrt', 'ell_f', 'ell_g', 'var_f', 'var_g', 'noise')


def _t(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=DT)


def rbf_K(X, X2, ell, var):
    """onofftf/main.py:41-57."""
    X = X / ell
    Xs = torch.sum(torch.square(X), 1)
    if X2 is None:
        r2 = -2 * torch.matmul(X, X.t()) + Xs.reshape(-1, 1) + Xs.reshape(1, -1)
    else:
        X2 = X2 / ell
        X2s = torch.sum(torch.square(X2), 1)
        r2 = -2 * torch.matmul(X, X2.t()) + Xs.reshape(-1, 1) + X2s.reshape(1, -1)
    return var * torch.exp(-r2 / 2)


def conditional(Xnew, Z, ell, var, q_mu, q_sqrt, jitter):
    """onofftf/main.py:257-305 (diag q, unwhitened)."""
    M = Z.shape[0]
    Kmn = rbf_K(Z, Xnew, ell, var)
    Kmm = rbf_K(Z, None, ell, var) + torch.eye(M, dtype=DT) * jitter
    Lm = torch.linalg.cholesky(Kmm)
    A = torch.linalg.solve_triangular(Lm, Kmn, upper=False)
    fvar = var - torch.sum(torch.square(A), 0)
    A = torch.linalg.solve_triangular(Lm.t(), A, upper=True)
    fmean = torch.matmul(A.t(), q_mu.reshape(M, 1))
    fvar = fvar + torch.sum(torch.square(A * q_sqrt.reshape(M, 1)), 0)
    return fmean.reshape(-1, 1), fvar.reshape(-1, 1)


def gauss_kl_diag(q_mu, q_sqrt, K):
    """onofftf/main.py:187-252 diag branch, minus the extra jitter at :199."""
    M = K.shape[0]
    q_mu = q_mu.reshape(M, 1)
    q_sqrt = q_sqrt.reshape(M, 1)
    Lp = torch.linalg.cholesky(K)
    alpha = torch.linalg.solve_triangular(Lp, q_mu, upper=False)
    Lp_inv = torch.linalg.solve_triangular(Lp, torch.eye(M, dtype=DT), upper=False)
    K_inv = torch.linalg.solve_triangular(Lp.t(), Lp_inv, upper=True)
    twoKL = torch.sum(torch.square(alpha)) - float(M) - torch.sum(torch.log(torch.square(q_sqrt))) \
        + torch.sum(torch.diagonal(K_inv).reshape(M, 1) * torch.square(q_sqrt)) \
        + torch.sum(torch.log(torch.square(torch.diagonal(Lp))))
    return 0.5 * twoKL


def probit_expectations(gmean, gvar):
    """onoffgpf/OnOffSVGP.py:168-204."""
    z = gmean / torch.sqrt(1. + gvar)
    a = 1 / torch.sqrt(1. + (2 * gvar))
    cdfz = 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0))) * (1. - 2.e-3) + 1.e-3
    tz = torch.atan(a) / (2 * math.pi) * torch.exp((-1 / 2) * (torch.square(torch.abs(z)) * (torch.square(a) + 1)))
    pgmean = cdfz
    pgmeansq = cdfz - 2. * tz
    pgvar = cdfz - 2. * tz - torch.square(cdfz)
    pgmeansq = (pgmeansq + torch.abs(pgmeansq)) / 2.
    pgvar = (pgvar + torch.abs(pgvar)) / 2.
    return pgmean, pgmeansq, pgvar


def variational_expectations(Fmu, Fvar, Fmuvar, Y, noise):
    """onoffgpf/OnOffLikelihood.py:30-32."""
    return -0.5 * math.log(2 * math.pi) - 0.5 * torch.log(noise) - 0.5 * (torch.square(Y - Fmu) + Fvar + Fmuvar) / noise


def data_term(X, Y, p, jitter, g_offset=0.0):
    """sum_n var_exp_n over the rows of X (onoffgpf/OnOffSVGP.py:113-116,124-152)."""
    fmean, fvar = conditional(X, p['Zf'], p['ell_f'], p['var_f'], p['u_fm'], p['u_fs_sqrt'], jitter)
    if 'mean_a' in p:                                                   # fmean + self.mean_function(Xnew), OnOffSVGP.py:134
        fmean = fmean + torch.matmul(X, p['mean_a'].reshape(-1, 1))
    if 'mean_b' in p:
        fmean = fmean + p['mean_b']
    gmean, gvar = conditional(X, p['Zg'], p['ell_g'], p['var_g'], p['u_gm'], p['u_gs_sqrt'], jitter)
    gmean = gmean + g_offset
    e1, e2, ev = probit_expectations(gmean, gvar)
    return torch.sum(variational_expectations(e1 * fmean, e2 * fvar, ev * torch.square(fmean), Y.reshape(-1, 1), p['noise']))


def prior_kl(p, jitter):
    """onoffgpf/OnOffSVGP.py:96-101."""
    Kf = rbf_K(p['Zf'], None, p['ell_f'], p['var_f']) + torch.eye(p['Zf'].shape[0], dtype=DT) * jitter
    Kg = rbf_K(p['Zg'], None, p['ell_g'], p['var_g']) + torch.eye(p['Zg'].shape[0], dtype=DT) * jitter
    return gauss_kl_diag(p['u_fm'], p['u_fs_sqrt'], Kf) + gauss_kl_diag(p['u_gm'], p['u_gs_sqrt'], Kg)


MEAN_KEYS = ('mean_a', 'mean_b')   # optional mean function of f: m(x) = mean_b + mean_a . x


def make_leaves(p_np):
    keys = PARAM_KEYS + tuple(k for k in MEAN_KEYS if p_np.get(k) is not None)
    return {k: _t(p_np[k]).clone().requires_grad_(True) for k in keys}


def elbo_and_grad(X, Y, p_np, jitter, scale=1.0, g_offset=0.0, chunk=20000, include_kl=True, need_grad=True):
    """One ELBO 'step' on the CPU: value + gradient w.r.t. every (constrained) parameter.

    Rows are processed in chunks of `chunk` (exact: the data term is a sum over points;
    the un-chunked graph needs > 64 GB at N=1e6, M=1024).  Returns (elbo, data, kl, grads dict of numpy).
    """
    Xt, Yt = _t(X), _t(Y).reshape(-1, 1)
    p = make_leaves(p_np)
    data = 0.0
    for s in range(0, Xt.shape[0], chunk):
        if need_grad:
            d = data_term(Xt[s:s + chunk], Yt[s:s + chunk], p, jitter, g_offset)
            (d * scale).backward()
        else:
            with torch.no_grad():
                d = data_term(Xt[s:s + chunk], Yt[s:s + chunk], p, jitter, g_offset)
        data += float(d.detach())
    kl = 0.0
    if include_kl:
        if need_grad:
            k = prior_kl(p, jitter)
            (-k).backward()
        else:
            with torch.no_grad():
                k = prior_kl(p, jitter)
        kl = float(k.detach())
    grads = {k: (p[k].grad.numpy().copy() if p[k].grad is not None else np.zeros(tuple(p[k].shape))) for k in p} \
        if need_grad else None
    return data * scale - kl, data, kl, grads


# ---------------------------------------------------------------- Kronecker (literal order) -----------
def np_kron(*args):
    out = torch.ones((1, 1), dtype=DT)
    for A in args:
        out = (out.reshape(out.shape[0], 1, out.shape[1], 1) * A.reshape(1, A.shape[0], 1, A.shape[1])
               ).reshape(out.shape[0] * A.shape[0], out.shape[1] * A.shape[1])
    return out


def kron_mv(As, x):
    """scripts/onoff.py:215-225."""
    num = [A.shape[0] for A in As]
    N = int(np.prod(num))
    b = x.reshape(N, 1)
    for p_, Ap in enumerate(As):
        Xm = b.reshape(num[p_], N // num[p_])
        b = torch.matmul(Xm.t(), Ap.t()).reshape(N, 1)
    return b


def kron_inf(Xnew, Z_list, ell_list, var_list, q_mu, q_sqrt, jitter):
    """scripts/onoff.py:186-213, literal (dense) op order."""
    Kmm = [rbf_K(Z, None, l, v) + torch.eye(Z.shape[0], dtype=DT) * jitter for Z, l, v in zip(Z_list, ell_list, var_list)]
    Kinv = [torch.linalg.inv(K) for K in Kmm]
    alpha = kron_mv(Kinv, q_mu)
    Nb = Xnew.shape[0]
    Knn = torch.ones((Nb, 1), dtype=DT)
    Kmn_kron, c0 = [], 0
    for Z, l, v in zip(Z_list, ell_list, var_list):
        xnew = Xnew[:, c0:c0 + Z.shape[1]]
        c0 += Z.shape[1]
        Knn = Knn * v
        Kmn_kron.append(rbf_K(Z, xnew, l, v))
    S = torch.diag(torch.square(q_sqrt).reshape(-1))
    Kmn = (Kmn_kron[0][:, None, :] * Kmn_kron[1][None, :, :]).reshape(-1, Nb)
    A = torch.matmul(np_kron(*Kinv), Kmn)
    mu = torch.matmul(Kmn.t(), alpha)
    var = Knn - torch.diagonal(torch.matmul(Kmn.t(), A) - torch.matmul(torch.matmul(A.t(), S), A)).reshape(-1, 1)
    return mu, var


def gauss_kl_kron(q_mu, q_sqrt, K_kron):
    """onofftf/main.py:350-387 literal."""
    Lp = np_kron(*[torch.linalg.cholesky(K) for K in K_kron])
    M = Lp.shape[0]
    q_mu, q_sqrt = q_mu.reshape(M, 1), q_sqrt.reshape(M, 1)
    alpha = torch.linalg.solve_triangular(Lp, q_mu, upper=False)
    Lp_inv = torch.linalg.solve_triangular(Lp, torch.eye(M, dtype=DT), upper=False)
    K_inv = torch.linalg.solve_triangular(Lp.t(), Lp_inv, upper=True)
    twoKL = torch.sum(torch.square(alpha)) - float(M) - torch.sum(torch.log(torch.square(q_sqrt))) \
        + torch.sum(torch.diagonal(K_inv).reshape(M, 1) * torch.square(q_sqrt)) \
        + torch.sum(torch.log(torch.square(torch.diagonal(Lp))))
    return 0.5 * twoKL


KRON_KEYS = ('Zf', 'Zg', 'ell_f', 'ell_g', 'var_f', 'var_g')  # lists (one entry per factor)
KRON_VEC_KEYS = ('u_fm', 'u_gm', 'u_fs_sqrt', 'u_gs_sqrt', 'noise')


def kron_elbo_and_grad(X, Y, p_np, jitter, scale=1.0, g_offset=0.0, include_kl=True, need_grad=True, f_mu=None):
    """scripts/onoff.py:286-319 value (+ autograd gradient); literal dense order -> small batches only.  f_mu: the optional constant
    of build_predict (:161,168-169); when given, grads['f_mu'] is its gradient."""
    Xt, Yt = _t(X), _t(Y).reshape(-1, 1)
    p = {}
    for k in KRON_KEYS:
        p[k] = [_t(v).clone().requires_grad_(need_grad) for v in p_np[k]]
    for k in KRON_VEC_KEYS:
        p[k] = _t(p_np[k]).clone().requires_grad_(need_grad)
    fmu = None if f_mu is None else _t(f_mu).clone().requires_grad_(need_grad)
    with torch.set_grad_enabled(need_grad):
        fmean, fvar = kron_inf(Xt, p['Zf'], p['ell_f'], p['var_f'], p['u_fm'], p['u_fs_sqrt'], jitter)
        if fmu is not None:
            fmean = fmean + fmu
        gmean, gvar = kron_inf(Xt, p['Zg'], p['ell_g'], p['var_g'], p['u_gm'], p['u_gs_sqrt'], jitter)
        gmean = gmean + g_offset
        e1, e2, ev = probit_expectations(gmean, gvar)
        data = torch.sum(variational_expectations(e1 * fmean, e2 * fvar, ev * torch.square(fmean), Yt, p['noise']))
        kl = torch.zeros((), dtype=DT)
        if include_kl:
            Kf = [rbf_K(Z, None, l, v) + torch.eye(Z.shape[0], dtype=DT) * jitter for Z, l, v in zip(p['Zf'], p['ell_f'], p['var_f'])]
            Kg = [rbf_K(Z, None, l, v) + torch.eye(Z.shape[0], dtype=DT) * jitter for Z, l, v in zip(p['Zg'], p['ell_g'], p['var_g'])]
            kl = gauss_kl_kron(p['u_fm'], p['u_fs_sqrt'], Kf) + gauss_kl_kron(p['u_gm'], p['u_gs_sqrt'], Kg)
        elbo = data * scale - kl
    grads = None
    if need_grad:
        elbo.backward()
        grads = {}
        for k in KRON_KEYS:
            grads[k] = [v.grad.numpy().copy() for v in p[k]]
        for k in KRON_VEC_KEYS:
            grads[k] = p[k].grad.numpy().copy()
        if fmu is not None:
            grads['f_mu'] = float(fmu.grad)
    return float(elbo.detach()), float(data.detach()), float(kl.detach()), grads


# ---------------------------------------------------------------- single-latent heads (baselines) -----
HEAD_KEYS = ('Zf', 'ell_f', 'var_f')


def kron_head_elbo_and_grad(X, Y, p_np, lik, jitter, scale=1.0, f_mu=0.0, include_kl=True, need_grad=True):
    """scripts/svgp.py:116-233 ('gaussian') / scripts/classifier.py:116-240 ('bernoulli'), literal dense order + autograd.
    Returns (elbo, data, kl, grads) with grads keys Zf, ell_f, var_f (lists), u_fm, u_fs_sqrt, noise, f_mu."""
    Xt, Yt = _t(X), _t(Y).reshape(-1, 1)
    p = {k: [_t(v).clone().requires_grad_(need_grad) for v in p_np[k]] for k in HEAD_KEYS}
    for k in ('u_fm', 'u_fs_sqrt'):
        p[k] = _t(p_np[k]).clone().requires_grad_(need_grad)
    p['noise'] = _t(p_np.get('noise', 1.0)).clone().requires_grad_(need_grad)
    p['f_mu'] = _t(f_mu).clone().requires_grad_(need_grad)
    with torch.set_grad_enabled(need_grad):
        fmean, fvar = kron_inf(Xt, p['Zf'], p['ell_f'], p['var_f'], p['u_fm'], p['u_fs_sqrt'], jitter)
        fmean = fmean + p['f_mu']
        if lik == 'gaussian':
            ve = -0.5 * np.log(2 * np.pi) - 0.5 * torch.log(p['noise']) - 0.5 * (torch.square(Yt - fmean) + fvar) / p['noise']
        else:
            pr = 0.5 * (1.0 + torch.erf(fmean / torch.sqrt(1 + fvar) / np.sqrt(2.0))) * (1 - 2e-3) + 1e-3
            ve = torch.log(torch.where(Yt == 1, pr, 1 - pr))
        data = torch.sum(ve)
        kl = torch.zeros((), dtype=DT)
        if include_kl:
            Kf = [rbf_K(Z, None, l, v) + torch.eye(Z.shape[0], dtype=DT) * jitter for Z, l, v in zip(p['Zf'], p['ell_f'], p['var_f'])]
            kl = gauss_kl_kron(p['u_fm'], p['u_fs_sqrt'], Kf)
        elbo = data * scale - kl
    grads = None
    if need_grad:
        elbo.backward()
        grads = {k: [v.grad.numpy().copy() for v in p[k]] for k in HEAD_KEYS}
        for k in ('u_fm', 'u_fs_sqrt', 'noise', 'f_mu'):
            grads[k] = p[k].grad.numpy().copy() if p[k].grad is not None else np.zeros(tuple(p[k].shape))
    return float(elbo.detach()), float(data.detach()), float(kl.detach()), grads
