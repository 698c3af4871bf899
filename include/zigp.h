/* libzigp -- C-ABI of the MI355X-native zero-inflated ("OnOff") sparse variational GP engine.
 *
 * The reference (hegdepashupati/zero-inflated-gp) is pure Python on TensorFlow 1.x / GPflow 0.4.0 and
 * has NO FFI / plugin / operator interface of its own.  The narrowest seams on
 * its ELBO hot path are Python methods; each entry point below names the reference code it replaces
 * (file:line relative to the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer
 * would add on the reference side.
 *
 * Conventions: every function returns 0 on success and < 0 on error (never throws across the ABI):
 *   ZIGP_EARG  bad argument        ZIGP_EHIP   HIP runtime error        ZIGP_ECOMM  RCCL error (zigp_comm_*)
 *   ZIGP_ENOTPD  Cholesky hit a pivot <= rtol * eps * (variance + jitter), rtol = 8 by default: non-positive, or zero to rounding
 *                (tf.cholesky raises InvalidArgumentError on a non-positive pivot; an exactly singular Kuu rounds either way there;
 *                zigp_set_pivot_rtol(ctx, 0) gives that bare `pivot > 0` test); zigp_last_info() returns the 1-based pivot index
 *                within the failing matrix, zigp_last_error() the message (which names the matrix).
 * All arrays are float64, C-contiguous (row-major), owned by the caller.  Pointers are HOST pointers
 * unless the name says "device".  One ctx per GPU; a ctx is not thread-safe; calls are synchronous
 * (all internal streams are joined before returning).  Reductions are fixed-order: results are
 * bit-stable run to run on the same device, chunk size and shard layout.
 */
#ifndef ZIGP_H
#define ZIGP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZIGP_OK 0
#define ZIGP_EARG (-1)
#define ZIGP_EHIP (-2)
#define ZIGP_ENOTPD (-3)
#define ZIGP_ECOMM (-4)

typedef struct zigp_ctx zigp_ctx;

/* Constrained parameter values of the dense model.  Replaces the Param members created in
 * OnOffSVGP.__init__ (onoffgpf/OnOffSVGP.py:50-71), the two gpflow.kernels.RBF objects
 * (zero-inflated-gpflow.ipynb:98-104; twin KernSE onofftf/main.py:33-63) and
 * OnOffLikelihood.variance (onoffgpf/OnOffLikelihood.py:26).
 * ell_*: D lengthscales (ARD; broadcast a scalar on the host). u_*s_sqrt: diagonal q_sqrt (q_diag=True,
 * OnOffSVGP.py:34). */
typedef struct {
  int32_t Mf, Mg, D, reserved;
  const double *Zf, *Zg;             /* (Mf,D), (Mg,D) inducing inputs */
  const double *u_fm, *u_gm;         /* (Mf), (Mg) variational means */
  const double *u_fs_sqrt, *u_gs_sqrt; /* (Mf), (Mg) variational std-devs (> 0) */
  const double *ell_f, *ell_g;       /* (D), (D) */
  double var_f, var_g;               /* kernel variances */
  double noise;                      /* likelihood variance */
} zigp_params;

/* Gradient of the returned ELBO w.r.t. the constrained values above (same shapes). The host applies
 * the Log1pe ("positive" transform) chain rule, as GPflow's Param / onofftf Param.get_tfv do
 * (onofftf/main.py:170-174). */
typedef struct {
  double *Zf, *Zg, *u_fm, *u_gm, *u_fs_sqrt, *u_gs_sqrt, *ell_f, *ell_g;
  double var_f, var_g, noise;
} zigp_grads;

int zigp_create(zigp_ctx** out, int device_id);
int zigp_destroy(zigp_ctx* ctx);
const char* zigp_last_error(zigp_ctx* ctx);
int zigp_last_info(zigp_ctx* ctx);

/* Tunables: chunk = number of data rows processed per pass through the fused pipeline (multiple of
 * 1024 and <= 1048576).  Default, until this is called: 32768 * 1024 / M rows, clamped to [32768, 131072]; a row range of up to 131072
 * rows goes through in one pass while its panels stay within 9 GB.  chunk_rows = 0 returns to that default rule. */
int zigp_set_chunk(zigp_ctx* ctx, int64_t chunk_rows);
/* The chunk (rows per pass) the dense path uses for M inducing points per latent on a long row range: the zigp_set_chunk value, else
 * the default rule. */
int64_t zigp_get_chunk(zigp_ctx* ctx, int32_t M);
/* Rows per pass the dense path actually uses for a row range of `span` rows: the range is cut into equal passes of at most the chunk above
 * (a multiple of 1024 rows each), and a range of up to 131072 rows goes through in ONE pass while its panels stay within 9 GB (M <= 1024). */
int64_t zigp_get_chunk_rows(zigp_ctx* ctx, int32_t M, int64_t span);
/* Smallest Cholesky pivot accepted, as a multiple of eps * (kernel variance + jitter).  Default 8 (see ZIGP_ENOTPD above);
 * 0 reproduces tf.cholesky / LAPACK potrf, which fail on a non-positive pivot only (onofftf/main.py:200,268,355). */
int zigp_set_pivot_rtol(zigp_ctx* ctx, double rtol);

/* Training data.  Replaces the MinibatchData / DataHolder objects of OnOffSVGP.__init__
 * (onoffgpf/OnOffSVGP.py:38-47) and the X/Y feed_dict of scripts/onoff.py:379-381.
 * zigp_set_data copies host arrays into HBM once; zigp_set_data_device adopts device pointers
 * (e.g. torch tensors) without copying -- the caller keeps them alive. X is (N,D), Y is (N). */
int zigp_set_data(zigp_ctx* ctx, const double* X, const double* Y, int64_t N, int32_t D);
int zigp_set_data_device(zigp_ctx* ctx, const double* dX, const double* dY, int64_t N, int32_t D);
/* Minibatch by row indices: the n rows `rows` of the resident set (repeats allowed) are gathered on the device and become the active
 * data, rows [0, n), of the calls that follow -- replaces the per-step sample of GPflow's MinibatchData (onoffgpf/OnOffSVGP.py:46-47)
 * without re-uploading X and Y (8 bytes per row cross the bus instead of 8 (D + 1)).  n = 0: back to the whole resident set;
 * zigp_set_data / zigp_set_data_device reset it. */
int zigp_select_rows(zigp_ctx* ctx, const int64_t* rows, int64_t n);

/* One ELBO evaluation ("step" when grads != NULL) over rows [row_begin, row_end) of the resident data.
 * Replaces OnOffSVGP.build_likelihood (onoffgpf/OnOffSVGP.py:107-122) + its gradient (GPflow
 * Model.optimize -> tf.gradients) i.e. one L-BFGS-B function evaluation / one sess.run(train_op)
 * (scripts/onoff.py:379).
 *   *elbo_data = scale * sum_n var_exp_n  (the data term incl. minibatch scale, OnOffSVGP.py:119-122)
 *   *kl        = KL_f + KL_g              (build_prior_KL, OnOffSVGP.py:96-101); 0 if include_kl == 0
 *   ELBO = *elbo_data - *kl ; grads (nullable) = d ELBO / d params, with the KL part only if include_kl.
 * jitter is added to diag(Kuu) (settings.numerics.jitter_level, OnOffSVGP.py:96-97);
 * g_offset is added to gmean before the probit moments (0 for fit; -1 reproduces onofftf/onoffpred.py:141).
 * Data-parallel use: every rank calls this on its own row shard with include_kl = (rank == 0) and the
 * host sums {elbo_data, kl, grads} over ranks (one all-reduce). */
int zigp_elbo(zigp_ctx* ctx, const zigp_params* p, double jitter, double scale, double g_offset,
              int64_t row_begin, int64_t row_end, int32_t include_kl,
              double* elbo_data, double* kl, zigp_grads* grads);

/* Prediction.  Replaces OnOffSVGP.predict_onoffgp -> build_predict (onoffgpf/OnOffSVGP.py:124-152,160-162).
 * out9 is (9,N): gfmean, gfvar, gfmeanu, fmean, fvar, gmean, gvar, ephi_g, evar_phi_g (order of :152). */
int zigp_predict(zigp_ctx* ctx, const zigp_params* p, const double* Xnew, int64_t N, double jitter,
                 double g_offset, double* out9);
/* The same with Xnew (N,D) and out9 (9,N) in DEVICE memory of the context's GPU (the companion of zigp_set_data_device): nothing crosses
 * PCIe but the parameters; complete on return.  The caller orders its own streams around the call (it is synchronous on the host). */
int zigp_predict_device(zigp_ctx* ctx, const zigp_params* p, const double* dXnew, int64_t N, double jitter,
                        double g_offset, double* d_out9);

/* Prior KL only (OnOffSVGP.compute_prior_KL, onoffgpf/OnOffSVGP.py:164-166): kl2 = {KL_f, KL_g}. */
int zigp_prior_kl(zigp_ctx* ctx, const zigp_params* p, double jitter, double* kl2);

/* RBF kernel matrices (gpflow.kernels.RBF.K / KernSE.K onofftf/main.py:53-57; kernse_np.K
 * onofftf/utils.py:48-52): K (n1,n2) = var * exp(-0.5 * sum_d ((x1-x2)/ell_d)^2).  X2 == NULL -> X1. */
int zigp_rbf_K(zigp_ctx* ctx, const double* X1, int64_t n1, const double* X2, int64_t n2, int32_t D,
               const double* ell, double var, double* K);

/* ---- Kronecker (space x time) variant: scripts/onoff.py:143-319, onofftf/main.py:350-387 ---- */
/* Two factors (the reference's Khatri-Rao line scripts/onoff.py:206 is written for exactly two):
 * factor 0 acts on the first D0 input columns, factor 1 on the next D1 (scripts/onoff.py:243-250).
 * u_* and u_*s_sqrt have M0*M1 entries, factor-0-major (row = i*M1 + j, scripts/onoff.py:206). */
typedef struct {
  int32_t M0f, M1f, M0g, M1g, D0, D1, reserved0, reserved1;
  const double *Z0f, *Z1f, *Z0g, *Z1g;       /* (M0,D0), (M1,D1) */
  const double *ell0f, *ell1f, *ell0g, *ell1g; /* (D0), (D1) */
  double var0f, var1f, var0g, var1g;
  const double *u_fm, *u_gm, *u_fs_sqrt, *u_gs_sqrt;
  double noise;
} zigp_kron_params;

typedef struct {
  double *Z0f, *Z1f, *Z0g, *Z1g, *ell0f, *ell1f, *ell0g, *ell1g;
  double var0f, var1f, var0g, var1g;
  double *u_fm, *u_gm, *u_fs_sqrt, *u_gs_sqrt;
  double noise;
} zigp_kron_grads;

/* ELBO / step on an explicit minibatch (host arrays), as scripts/onoff.py:377-381 feeds it.
 * Replaces build_prior_kl + build_predict/kron_inf + cost (scripts/onoff.py:143-213,286-319). */
/* f_mu: the optional constant added to fmean (`if not f_mu is None: fmean = fmean + f_mu.get_tfv()`, scripts/onoff.py:161,168-169;
 * onofftf/onoffpred.py:127,135-136; pass 0 for the reference's default f_mu=None); d_f_mu (nullable) receives its gradient. */
int zigp_kron_elbo(zigp_ctx* ctx, const zigp_kron_params* p, const double* X, const double* Y, int64_t N,
                   double jitter, double scale, double g_offset, double f_mu, int32_t include_kl,
                   double* elbo_data, double* kl, zigp_kron_grads* grads, double* d_f_mu);
/* The same step on rows [row_begin, row_end) of the RESIDENT data set (zigp_set_data / zigp_set_data_device, D = D0 + D1): no
 * host->device copy of the minibatch.  The reference's iterator (DataSet.next_batch, onofftf/main.py:98-133) shuffles once per
 * epoch and then hands out contiguous slices, so a host that makes the permuted epoch resident feeds every step by row range;
 * the full-batch configuration (BASELINE cfg5) is rows [0, N). */
int zigp_kron_elbo_rows(zigp_ctx* ctx, const zigp_kron_params* p, int64_t row_begin, int64_t row_end, double jitter, double scale,
                        double g_offset, double f_mu, int32_t include_kl, double* elbo_data, double* kl, zigp_kron_grads* grads,
                        double* d_f_mu);
/* Replaces predict_onoff's graph (onofftf/onoffpred.py:127-200); out9 as zigp_predict. */
int zigp_kron_predict(zigp_ctx* ctx, const zigp_kron_params* p, const double* Xnew, int64_t N, double jitter,
                      double g_offset, double f_mu, double* out9);

/* ---- the fit loop on the device --------------------------------------------------------------------------------------------------
 * Replaces the reference's training loop body, scripts/onoff.py:375-381 (`sess.run(train_op)` on `train_data.next_batch(num_minibatch)`),
 * for n_steps consecutive iterations: gradient of cost = -(scale * sum var_exp - KL) (:318,334), chained through the Log1pe transform of
 * the positive parameters (GPflow transforms.positive, :88-123), one tf.train.AdamOptimizer per learning rate (:325-350; lr_t =
 * lr sqrt(1 - beta2^t) / (1 - beta1^t), m / (sqrt(v) + eps)).  The parameters live on the device for the whole call: every step's
 * kernels read the parameter image the previous step's update wrote, the steps are enqueued back to back on one stream and the host
 * synchronises ONCE, at the end of the call (the reference logs every 200 iterations, :418: call it with n_steps = 200).
 *   shape        the model's sizes (M0f .. D1); its pointer and value fields are ignored
 *   free_state   in/out [n_free]: the UNCONSTRAINED parameters in this block order -- for latent f, then g: Z0 (M0 x D0), Z1 (M1 x D1),
 *                u_m (M0 M1), u_s_sqrt (M0 M1), ell0 (D0), ell1 (D1), var0, var1; then the likelihood variance (ZIGP_FIT_BLOCKS = 17 blocks)
 *   adam_m, adam_v   in/out [n_free]: Adam moments (zeros at iteration 0)
 *   t0           iterations done before this call (the first step of the call is Adam's t = t0 + 1)
 *   row_begin    [n_steps]: step i uses rows [row_begin[i], row_begin[i] + batch) of the RESIDENT data set (zigp_set_data; the permuted
 *                epoch, onofftf/main.py:98-133); a negative value -(1 + k) selects rows [k batch, (k + 1) batch) of the host arrays Xw, Yw
 *                instead (the one wrap-around batch per epoch, which is a concatenation of the old and the new permutation, :125-129)
 *   elbo_data, kl    out [n_steps] (nullable): scale * sum var_exp and KL of every step, evaluated at the parameters BEFORE its update
 * Grids beyond the fused kernels (a factor of more than 32 points next to one of more than 16, or more than 112) return ZIGP_EARG: step
 * them with zigp_kron_elbo_rows and a host optimiser.  A Cholesky failure in step k returns ZIGP_ENOTPD; free_state / adam_* then hold the state
 * before the failing step -- the k updates before it HAVE been applied: the caller's iteration count advances by k -- and the history
 * entries from step k on are NaN (the finite prefix is the history of the applied steps).  With a communicator (zigp_comm_init) every step's result
 * block is summed over the ranks before its update, so all ranks hold the same parameters (each passes its own rows, include_kl as usual
 * is rank 0's). */
#define ZIGP_FIT_BLOCKS 17
typedef struct {
  double lr[ZIGP_FIT_BLOCKS];          /* Adam learning rate of each block */
  int32_t positive[ZIGP_FIT_BLOCKS];   /* 1: value = log(1 + exp(x)) + 1e-6 (Log1pe), 0: value = x */
  int32_t reserved;
  double beta1, beta2, eps;            /* TensorFlow's defaults: 0.9, 0.999, 1e-8 */
} zigp_kron_fit_opts;
int zigp_kron_fit_steps(zigp_ctx* ctx, const zigp_kron_params* shape, const zigp_kron_fit_opts* opts,
                        double* free_state, double* adam_m, double* adam_v, int64_t n_free,
                        int64_t t0, int32_t n_steps, const int64_t* row_begin, int64_t batch,
                        const double* Xw, const double* Yw, double jitter, double scale, int32_t include_kl,
                        double* elbo_data, double* kl);
/* Updates applied by the LAST zigp_kron_fit_steps call of this context: n_steps after a call that returned 0, the k steps before the failing
 * one after ZIGP_ENOTPD, 0 when the call ended before its first step (bad argument, HIP error).  This -- not a scan of the history for
 * NaN: an applied step may itself have a non-finite ELBO -- is what the caller's iteration count and Adam's bias correction advance by. */
int64_t zigp_kron_fit_steps_applied(zigp_ctx* ctx);

/* Mean function of the latent f: m(x) = b + a . x, added to fmean before the likelihood and in zigp_predict
 * (`fmean = fmean + self.mean_function(Xnew)`, onoffgpf/OnOffSVGP.py:29,134).  Covers GPflow's Zero (the reference default:
 * D = -1 -- the state after zigp_create; a and b are ignored), Constant (D = 0, b = c; enabled also when c == 0, so that the
 * parameter still receives its gradient) and Linear with one output (a[D], b).  The setting
 * persists in the context.  zigp_get_mean_function_grad returns d(scale * sum var_exp)/d(a, b) of the LAST zigp_elbo
 * called with grads != NULL (zeros when the mean function is off); like the other gradients it is a per-shard partial
 * sum under data-parallel use. */
int zigp_set_mean_function(zigp_ctx* ctx, const double* a, int32_t D, double b);
int zigp_get_mean_function_grad(zigp_ctx* ctx, double* da, int32_t D, double* db);

/* Single-latent heads on the same Kronecker conditional -- the reference's baselines, which re-use kron_inf and
 * GaussKLkron with one latent f:
 *   ZIGP_LIK_GAUSSIAN   scripts/svgp.py:127-200,207-233 and scripts/hurdle.py:127-252 (regression; noise variance)
 *   ZIGP_LIK_BERNOULLI  scripts/classifier.py:116-240 (log probit(fmean / sqrt(1 + fvar)) of the 0/1 label, y == 1 is "on")
 * Only the f fields of zigp_kron_params / zigp_kron_grads are read / written (g pointers may be NULL; noise is ignored
 * by the Bernoulli head).  f_mu is the constant mean offset of classifier.py:70-72,136-137 (0 when include_f_mu is False);
 * d_f_mu (nullable) receives the gradient of the scaled data term with respect to it. */
#define ZIGP_LIK_ONOFF 0
#define ZIGP_LIK_GAUSSIAN 1
#define ZIGP_LIK_BERNOULLI 2
int zigp_kron_head_elbo(zigp_ctx* ctx, const zigp_kron_params* p, int32_t lik, const double* X, const double* Y,
                        int64_t N, double jitter, double scale, double f_mu, int32_t include_kl,
                        double* elbo_data, double* kl, zigp_kron_grads* grads, double* d_f_mu);
/* Replaces predict_svgp / predict_scgp (onofftf/svgppred.py:15-203, onofftf/svcppred.py:15-224).  out4 = 4 x N rows:
 * fmean, fvar, pfmean, pfvar.  Bernoulli: pfmean = probit(fmean / sqrt(1 + fvar)), pfvar = pfmean - pfmean^2
 * (svcppred.py "pfmean"/"pfvar"); Gaussian: pfmean = fmean, pfvar = fvar + noise (svgppred.py returns rows 0-1). */
int zigp_kron_head_predict(zigp_ctx* ctx, const zigp_kron_params* p, int32_t lik, const double* Xnew, int64_t N,
                           double jitter, double f_mu, double* out4);

/* ---- data-parallel exchange (new: the reference is single-process) ----
 * The ELBO data term is a sum over points (tf.reduce_sum(var_exp), onoffgpf/OnOffSVGP.py:122; scripts/onoff.py:307): one process per GPU
 * holds a row shard, and ONE ncclAllReduce(sum, f64) per step adds the packed [elbo_data, kl, gradient] vector of every rank, on the
 * device, on the library's stream (RCCL over xGMI; ~82 KB at M = 1024, D = 3).  RCCL is bound at run time (librccl.so.1).
 *   rank 0:      zigp_comm_unique_id(id)  -> send the 128 bytes to the other ranks by any means (torch.distributed broadcast, MPI, a file)
 *   every rank:  zigp_comm_init(ctx, rank, nranks, id)       (collective: returns when all ranks have joined)
 * From then on zigp_elbo, zigp_kron_elbo, zigp_kron_elbo_rows and zigp_kron_head_elbo return the SUM over ranks of elbo_data, kl, grads
 * (and d_f_mu, the mean-function gradient) on every rank: each rank passes its own rows and include_kl = (rank == 0), so that the KL
 * and its gradient are counted once.  Every rank must make the same calls in the same order with the same model sizes.  Prediction
 * entry points are unaffected.  A Cholesky failure is reported by every rank (their Kuu are identical), after the exchange. */
#define ZIGP_COMM_ID_BYTES 128
int zigp_comm_unique_id(void* id /* [ZIGP_COMM_ID_BYTES] */);
int zigp_comm_init(zigp_ctx* ctx, int32_t rank, int32_t nranks, const void* id);
/* ZIGP_OK when RCCL can be bound in this process (librccl.so.1 found, every symbol resolved, an NCCL 2.x version of the same major as
 * the headers the library was built against); *version (nullable) = its ncclGetVersion code.  Ranks should agree on this BEFORE any of
 * them enters the collective zigp_comm_init: a rank that cannot load RCCL would otherwise leave its peers waiting. */
int zigp_comm_available(int32_t* version);
/* zigp_comm_init gives up with ZIGP_ECOMM when its peers have not joined after `seconds` (default 120, or env ZIGP_COMM_TIMEOUT_S at
 * zigp_create); the communicator is then unusable on every rank: fall back to a host-side exchange or exit, do not retry on the same id.
 * The helper thread that is still inside ncclCommInitRank stays behind (it destroys the communicator itself should the call return late):
 * a process that goes on after a timeout should end through os._exit / a fresh child process rather than a normal interpreter shutdown. */
int zigp_comm_set_timeout(zigp_ctx* ctx, double seconds);
int zigp_comm_destroy(zigp_ctx* ctx);
/* Sum n host doubles over the ranks of the context's communicator, in place (staged through the device): for the few scalars a host loop
 * wants agreed on (a convergence flag, a timing), and the self-check the Python wrapper runs right after zigp_comm_init. */
int zigp_comm_allreduce_host(zigp_ctx* ctx, double* inout, int64_t n);
/* rank / nranks of the context's communicator (nranks = 0: none) and the number of all-reduces issued through it so far */
int zigp_comm_info(zigp_ctx* ctx, int32_t* rank, int32_t* nranks, int64_t* allreduce_calls);

/* Stream overlap inside zigp_elbo (default ON since round 3): the HBM-bound kernels of a row chunk (Kuf-cotangent reductions,
 * the next chunk's Kuf panels) run on a second HIP stream underneath the chunk's two MFMA-bound rank-N updates (-3 % per
 * cfg3 step), and (round 5) the point-wise stage of a gradient step rides as the leading workgroups of the launch of the J' product,
 * which does not depend on it (one launch boundary less per chunk).  Results are bit-identical either way.  Turn it off (0) to profile:
 * every kernel then runs alone, under its own name, on one stream and per-kernel durations from an external tracer (rocprofv3 --stats) mean
 * what they say; the library's own event timing (zigp_profile_*, include/zigp_diag.h) already keeps the chunks it times that way. */
int zigp_set_overlap(zigp_ctx* ctx, int32_t on);

#ifdef __cplusplus
}
#endif
#endif
