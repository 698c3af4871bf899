"""CPU oracle for the zero-inflated ("OnOff") GP ELBO hot path -- TEST INFRASTRUCTURE ONLY.

This is a NumPy/SciPy float64 restatement, in the reference's op order, of the
arithmetic on the hot path of hegdepashupati/zero-inflated-gp.  It is NOT part
of the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product path (zero-inflated-gp_amd/) never
imports this module and fails loudly when the HIP library is missing.

PARITY STATUS: *parity unpinned* except for the RBF kernel.  The reference has
no tests, golden vectors or fixtures for this path, and its own implementation
needs TensorFlow 1.x + GPflow 0.4.0 (un-vendored; README.md:7), neither of
which is installed or installable here.  Only onofftf/utils.py (NumPy-only) is
importable: its `kernse_np` pins `rbf_K`/`rbf_Kdiag` below via
tests/golden/g1_kernse_np.npz (made by oracle/make_golden.py).  Everything else
is pinned only against (i) an mpmath 50-digit evaluation of the same formulas,
(ii) central finite differences and (iii) torch autograd (oracle/zigp_oracle_torch.py).

Citations are file:line into /root/reference.
"""
import numpy as np
from scipy.linalg import cholesky, solve_triangular
from scipy.special import erf

LOG2PI = np.log(2.0 * np.pi)


# --------------------------------------------------------------------------
# a1  RBF kernel -- GPflow kernels.RBF; in-tree twins onofftf/main.py:41-63 (KernSE)
#     and onofftf/utils.py:34-58 (kernse_np)
# --------------------------------------------------------------------------
def square_dist(X, X2, lengthscales):
    """onofftf/main.py:41-51: r2 = |x/l|^2 + |z/l|^2 - 2 (x/l)(z/l)^T (broadcast divide :42)."""
    X = X / lengthscales
    Xs = np.sum(np.square(X), 1)
    if X2 is None:
        return -2 * np.matmul(X, X.T) + Xs.reshape(-1, 1) + Xs.reshape(1, -1)
    X2 = X2 / lengthscales
    X2s = np.sum(np.square(X2), 1)
    return -2 * np.matmul(X, X2.T) + Xs.reshape(-1, 1) + X2s.reshape(1, -1)


def rbf_K(X, X2, lengthscales, variance):
    """onofftf/main.py:53-57: K = variance * exp(-r2/2)."""
    return variance * np.exp(-square_dist(X, X2, lengthscales) / 2)


def rbf_Kdiag(X, variance):
    """onofftf/main.py:62-63: fill(variance)."""
    return np.full(X.shape[0], np.squeeze(variance), dtype=np.float64)


# --------------------------------------------------------------------------
# a2  dense conditional -- gpflow.conditionals.conditional, call sites
#     onoffgpf/OnOffSVGP.py:132,136; in-tree twin GPConditional onofftf/main.py:257-305
#     (q_diag=True, whiten=False, full_cov=False, num_latent=1)
# --------------------------------------------------------------------------
def conditional(Xnew, Z, lengthscales, variance, q_mu, q_sqrt, jitter):
    M = Z.shape[0]
    Kmn = rbf_K(Z, Xnew, lengthscales, variance)                      # main.py:266
    Kmm = rbf_K(Z, None, lengthscales, variance) + np.eye(M) * jitter  # main.py:267
    Lm = cholesky(Kmm, lower=True)                                     # main.py:268
    A = solve_triangular(Lm, Kmn, lower=True)                          # main.py:271
    fvar = rbf_Kdiag(Xnew, variance) - np.sum(np.square(A), 0)         # main.py:278
    A = solve_triangular(Lm.T, A, lower=False)                         # main.py:284
    fmean = np.matmul(A.T, q_mu.reshape(M, 1))                         # main.py:287
    LTA = A * q_sqrt.reshape(M, 1)                                     # main.py:291
    fvar = fvar + np.sum(np.square(LTA), 0)                            # main.py:302
    return fmean.reshape(-1, 1), fvar.reshape(-1, 1)                   # main.py:303


# --------------------------------------------------------------------------
# a3  gauss_kl_diag -- gpflow.kullback_leiblers.gauss_kl_diag, call sites
#     onoffgpf/OnOffSVGP.py:100-101 (K already jittered at :96-97);
#     in-tree twin GaussKL diag branch onofftf/main.py:187-252, WITHOUT its
#     second jitter at :199 (GPflow does not add one).
# --------------------------------------------------------------------------
def gauss_kl_diag(q_mu, q_sqrt, K):
    M = K.shape[0]
    q_mu = q_mu.reshape(M, 1)
    q_sqrt = q_sqrt.reshape(M, 1)
    Lp = cholesky(K, lower=True)                                       # main.py:200
    alpha = solve_triangular(Lp, q_mu, lower=True)                     # main.py:201
    mahalanobis = np.sum(np.square(alpha))                             # main.py:218
    constant = -float(q_sqrt.size)                                     # main.py:221
    logdet_qcov = np.sum(np.log(np.square(q_sqrt)))                    # main.py:224
    Lp_inv = solve_triangular(Lp, np.eye(M), lower=True)               # main.py:231-232
    K_inv = solve_triangular(Lp.T, Lp_inv, lower=False)                # main.py:233-234
    trace = np.sum(np.diag(K_inv).reshape(M, 1) * np.square(q_sqrt))   # main.py:235-236
    twoKL = mahalanobis + constant - logdet_qcov + trace               # main.py:242
    twoKL += np.sum(np.log(np.square(np.diag(Lp))))                    # main.py:246-250
    return 0.5 * twoKL


# --------------------------------------------------------------------------
# a4  probit moments -- onoffgpf/OnOffSVGP.py:168-204
# --------------------------------------------------------------------------
def probit_expectations(gmean, gvar):
    def normcdf(x):                                                    # OnOffSVGP.py:177-178
        return 0.5 * (1.0 + erf(x / np.sqrt(2.0))) * (1. - 2.e-3) + 1.e-3

    def owent(h, a):                                                   # OnOffSVGP.py:180-188
        h = np.abs(h)
        term1 = np.arctan(a) / (2 * np.pi)
        term2 = np.exp((-1 / 2) * (np.square(h) * (np.square(a) + 1)))
        return term1 * term2

    z = gmean / np.sqrt(1. + gvar)                                     # :190
    a = 1 / np.sqrt(1. + (2 * gvar))                                   # :191
    cdfz = normcdf(z)
    tz = owent(z, a)
    pgmean = cdfz                                                      # :196
    pgmeansq = cdfz - 2. * tz                                          # :197
    pgvar = cdfz - 2. * tz - np.square(cdfz)                           # :198
    pgmeansq = (pgmeansq + np.abs(pgmeansq)) / 2.                      # :201
    pgvar = (pgvar + np.abs(pgvar)) / 2.                               # :202
    return pgmean, pgmeansq, pgvar


# --------------------------------------------------------------------------
# a6  OnOffLikelihood.variational_expectations -- onoffgpf/OnOffLikelihood.py:30-32
# --------------------------------------------------------------------------
def variational_expectations(Fmu, Fvar, Fmuvar, Y, noise_variance):
    return -0.5 * np.log(2 * np.pi) - 0.5 * np.log(noise_variance) \
        - 0.5 * (np.square(Y - Fmu) + Fvar + Fmuvar) / noise_variance


# --------------------------------------------------------------------------
# a5  build_predict -- onoffgpf/OnOffSVGP.py:124-152 (zero mean function :134);
#     g_offset reproduces the `gmean - 1` prediction quirk of onofftf/onoffpred.py:141
# --------------------------------------------------------------------------
def mean_function(Xnew, p):
    """self.mean_function(Xnew), OnOffSVGP.py:29,134.  GPflow 0.4 mean functions with one output: Zero (default) -> 0,
    Constant(c) -> c, Linear(A, b) -> X A + b; here p['mean_a'] (D,) and p['mean_b'] (scalar), both optional."""
    m = np.zeros((Xnew.shape[0], 1))
    if p.get('mean_a') is not None:
        m = m + np.matmul(Xnew, np.asarray(p['mean_a'], dtype=np.float64).reshape(-1, 1))
    if p.get('mean_b') is not None:
        m = m + float(np.squeeze(p['mean_b']))
    return m


def build_predict(Xnew, p, jitter, g_offset=0.0):
    fmean, fvar = conditional(Xnew, p['Zf'], p['ell_f'], p['var_f'], p['u_fm'], p['u_fs_sqrt'], jitter)
    fmean = fmean + mean_function(Xnew, p)                              # :134
    gmean, gvar = conditional(Xnew, p['Zg'], p['ell_g'], p['var_g'], p['u_gm'], p['u_gs_sqrt'], jitter)
    gmean = gmean + g_offset
    ephi_g, ephi2_g, evar_phi_g = probit_expectations(gmean, gvar)
    gfmean = ephi_g * fmean                                            # :146
    gfvar = ephi2_g * fvar                                             # :147
    gfmeanu = evar_phi_g * np.square(fmean)                            # :148
    return gfmean, gfvar, gfmeanu, fmean, fvar, gmean, gvar, ephi_g, evar_phi_g  # :152


# a8  build_prior_KL -- onoffgpf/OnOffSVGP.py:96-101
def build_prior_KL(p, jitter):
    Kfmm = rbf_K(p['Zf'], None, p['ell_f'], p['var_f']) + np.eye(p['Zf'].shape[0]) * jitter
    Kgmm = rbf_K(p['Zg'], None, p['ell_g'], p['var_g']) + np.eye(p['Zg'].shape[0]) * jitter
    return gauss_kl_diag(p['u_fm'], p['u_fs_sqrt'], Kfmm), gauss_kl_diag(p['u_gm'], p['u_gs_sqrt'], Kgmm)


# a7  build_likelihood -- onoffgpf/OnOffSVGP.py:107-122 ; scale = num_data/|batch| (:119-120)
def elbo(X, Y, p, jitter, scale=1.0, g_offset=0.0):
    """Returns (ELBO, data term sum(var_exp) (unscaled), KL_f, KL_g)."""
    klf, klg = build_prior_KL(p, jitter)
    gfmean, gfvar, gfmeanu = build_predict(X, p, jitter, g_offset)[:3]
    var_exp = variational_expectations(gfmean, gfvar, gfmeanu, Y.reshape(-1, 1), p['noise'])
    data = np.sum(var_exp)
    return data * scale - (klf + klg), data, klf, klg


def elbo_chunked(X, Y, p, jitter, chunk=20000, scale=1.0, g_offset=0.0):
    """Same value as `elbo`, rows processed in chunks (the data term is a plain sum over points)."""
    klf, klg = build_prior_KL(p, jitter)
    data = 0.0
    for s in range(0, X.shape[0], chunk):
        gfmean, gfvar, gfmeanu = build_predict(X[s:s + chunk], p, jitter, g_offset)[:3]
        data += np.sum(variational_expectations(gfmean, gfvar, gfmeanu, Y[s:s + chunk].reshape(-1, 1), p['noise']))
    return data * scale - (klf + klg), data, klf, klg


# --------------------------------------------------------------------------
# a10 Kronecker conditional, LITERAL reference order -- scripts/onoff.py:186-241
#     p lists: Z_list[p] (M_p, d_p), ell_list[p], var_list[p]
# --------------------------------------------------------------------------
def _gen_inp_mask(Z_list):                                             # scripts/onoff.py:243-250
    mask, tmp = [], 0
    for Z in Z_list:
        mask.append(np.arange(tmp, tmp + Z.shape[1], dtype=np.int32))
        tmp += Z.shape[1]
    return mask


def np_kron(*args):                                                    # scripts/onoff.py:227-241
    out = np.ones((1, 1))
    for A in args:
        out = (out.reshape(out.shape[0], 1, out.shape[1], 1) * A.reshape(1, A.shape[0], 1, A.shape[1])
               ).reshape(out.shape[0] * A.shape[0], out.shape[1] * A.shape[1])
    return out


def kron_mv(As, x):                                                    # scripts/onoff.py:215-225
    num = [A.shape[0] for A in As]
    N = int(np.prod(num))
    b = x.reshape(N, 1)
    for p, Ap in enumerate(As):
        Xm = b.reshape(num[p], N // num[p])
        b = np.matmul(Xm.T, Ap.T).reshape(N, 1)
    return b


def kron_inf(Xnew, Z_list, ell_list, var_list, q_mu, q_sqrt, jitter):
    P = len(Z_list)
    mask = _gen_inp_mask(Z_list)
    Kmm = [rbf_K(Z_list[p], None, ell_list[p], var_list[p]) + np.eye(Z_list[p].shape[0]) * jitter
           for p in range(P)]                                          # :188-190
    Kmm_inv = [np.linalg.inv(Kmm[p]) for p in range(P)]                # :192 (tf.matrix_inverse, LU)
    alpha = kron_mv(Kmm_inv, q_mu)                                     # :193
    Nb = Xnew.shape[0]
    Knn = np.ones((Nb, 1))
    Kmn_kron = []
    for p in range(P):
        xnew = Xnew[:, mask[p]]                                        # :199
        Knn = Knn * rbf_Kdiag(xnew, var_list[p]).reshape(Nb, 1)        # :200
        Kmn_kron.append(rbf_K(Z_list[p], xnew, ell_list[p], var_list[p]))  # :201
    S = np.diag(np.square(q_sqrt).reshape(-1))                         # :204
    assert P == 2, "reference forms the Khatri-Rao product for exactly two factors (:206)"
    Kmn = (Kmn_kron[0][:, None, :] * Kmn_kron[1][None, :, :]).reshape(-1, Nb)  # :206
    A = np.matmul(np_kron(*Kmm_inv), Kmn)                              # :207
    mu = np.matmul(Kmn.T, alpha)                                       # :209
    var = Knn - np.diag(np.matmul(Kmn.T, A) - np.matmul(np.matmul(A.T, S), A)).reshape(-1, 1)  # :210-211
    return mu, var


# a11 GaussKLkron, LITERAL -- onofftf/main.py:350-387
def gauss_kl_kron(q_mu, q_sqrt, K_kron):
    Lp = np_kron(*[cholesky(K, lower=True) for K in K_kron])           # :355-356
    M = Lp.shape[0]
    q_mu = q_mu.reshape(M, 1)
    q_sqrt = q_sqrt.reshape(M, 1)
    alpha = solve_triangular(Lp, q_mu, lower=True)                     # :358
    mahalanobis = np.sum(np.square(alpha))
    constant = -float(q_sqrt.size)
    logdet_qcov = np.sum(np.log(np.square(q_sqrt)))
    Lp_inv = solve_triangular(Lp, np.eye(M), lower=True)               # :372-373
    K_inv = solve_triangular(Lp.T, Lp_inv, lower=False)                # :374-375
    trace = np.sum(np.diag(K_inv).reshape(M, 1) * np.square(q_sqrt))   # :376-377
    twoKL = mahalanobis + constant - logdet_qcov + trace
    twoKL += np.sum(np.log(np.square(np.diag(Lp))))                    # :381-385
    return 0.5 * twoKL


def kron_build_predict(Xnew, p, jitter, g_offset=0.0, f_mu=None):
    """scripts/onoff.py:161-184 (fit: g_offset=0) / onofftf/onoffpred.py:127-154 (predict: g_offset=-1, :141); f_mu: the optional
    constant of :161,168-169 (None in every call the reference makes)."""
    fmean, fvar = kron_inf(Xnew, p['Zf'], p['ell_f'], p['var_f'], p['u_fm'], p['u_fs_sqrt'], jitter)
    if f_mu is not None:
        fmean = fmean + f_mu                                           # :168-169
    gmean, gvar = kron_inf(Xnew, p['Zg'], p['ell_g'], p['var_g'], p['u_gm'], p['u_gs_sqrt'], jitter)
    gmean = gmean + g_offset
    ephi_g, ephi2_g, evar_phi_g = probit_expectations(gmean, gvar)
    return (ephi_g * fmean, ephi2_g * fvar, evar_phi_g * np.square(fmean),
            fmean, fvar, gmean, gvar, ephi_g, evar_phi_g)


def kron_prior_KL(p, jitter):                                          # scripts/onoff.py:143-159
    Kf = [rbf_K(Z, None, l, v) + np.eye(Z.shape[0]) * jitter for Z, l, v in zip(p['Zf'], p['ell_f'], p['var_f'])]
    Kg = [rbf_K(Z, None, l, v) + np.eye(Z.shape[0]) * jitter for Z, l, v in zip(p['Zg'], p['ell_g'], p['var_g'])]
    return gauss_kl_kron(p['u_fm'], p['u_fs_sqrt'], Kf), gauss_kl_kron(p['u_gm'], p['u_gs_sqrt'], Kg)


def kron_elbo(X, Y, p, jitter, scale=1.0, g_offset=0.0, f_mu=None):
    """scripts/onoff.py:286-319: cost = -(sum(var_exp)*scale - kl); returns (ELBO, data, KL_f, KL_g)."""
    klf, klg = kron_prior_KL(p, jitter)
    gfmean, gfvar, gfmeanu = kron_build_predict(X, p, jitter, g_offset, f_mu)[:3]
    data = np.sum(variational_expectations(gfmean, gfvar, gfmeanu, Y.reshape(-1, 1), p['noise']))
    return data * scale - (klf + klg), data, klf, klg


# --------------------------------------------------------------------------
# (f) rank 4: single-latent heads on the same kron_inf -- the reference's baselines
#     Gaussian regression  scripts/svgp.py:116-233 (= scripts/hurdle.py:127-252 on the "on" subset)
#     Bernoulli classifier scripts/classifier.py:116-240
# --------------------------------------------------------------------------
def head_probit(x):                                                    # classifier.py:216-217
    return 0.5 * (1.0 + erf(x / np.sqrt(2.0))) * (1 - 2e-3) + 1e-3


def kron_head_predict(Xnew, p, lik, jitter, f_mu=0.0):
    """svgp.py:127-138 -> (fmean, fvar); classifier.py:128-142 -> (pfmean, pfvar, fmean, fvar)."""
    fmean, fvar = kron_inf(Xnew, p['Zf'], p['ell_f'], p['var_f'], p['u_fm'], p['u_fs_sqrt'], jitter)
    fmean = fmean + f_mu                                               # classifier.py:136-137
    if lik == 'gaussian':
        return fmean, fvar
    pr = head_probit(fmean / np.sqrt(1 + fvar))                        # classifier.py:139
    return pr, pr - np.square(pr), fmean, fvar                         # :140


def kron_head_elbo(X, Y, p, lik, jitter, scale=1.0, f_mu=0.0):
    """svgp.py:207-233 / classifier.py:219-240: cost = -(sum(var_exp)*scale - kl); returns (ELBO, data, KL)."""
    Kf = [rbf_K(Z, None, l, v) + np.eye(Z.shape[0]) * jitter for Z, l, v in zip(p['Zf'], p['ell_f'], p['var_f'])]
    kl = gauss_kl_kron(p['u_fm'], p['u_fs_sqrt'], Kf)                  # svgp.py:116-125
    Y = Y.reshape(-1, 1)
    if lik == 'gaussian':
        fmean, fvar = kron_head_predict(X, p, lik, jitter, f_mu)
        ve = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(p['noise']) - 0.5 * (np.square(Y - fmean) + fvar) / p['noise']   # svgp.py:198-200
    else:
        pr = kron_head_predict(X, p, lik, jitter, f_mu)[0]
        ve = np.log(np.where(Y == 1, pr, 1 - pr))                      # classifier.py:213-214
    data = np.sum(ve)
    return data * scale - kl, data, kl
