/* libzigp -- C-ABI of the MI355X-native zero-inflated ("OnOff") sparse variational GP engine.
 *
 * The reference (hegdepashupati/zero-inflated-gp) is pure Python on TensorFlow 1.x / GPflow 0.4.0 and
 * has NO FFI / plugin / operator interface of its own.  The narrowest seams on
 * its ELBO hot path are Python methods; each entry point below names the reference code it replaces
 * (file:line relative to the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer
 * would add on the reference side.
 *
 * Conventions: every function returns 0 on success and < 0 on error (never throws across the ABI):
 *   ZIGP_EARG  bad argument        ZIGP_EHIP   HIP runtime error
 *   ZIGP_ENOTPD  Cholesky hit a pivot <= 8 eps (variance + jitter), i.e. non-positive or zero to rounding (tf.cholesky raises
 *                InvalidArgumentError on a non-positive pivot; an exactly singular Kuu rounds either way there);
 *                zigp_last_info() returns 1-based pivot index, zigp_last_error() the message.
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
 * 1024 and <= 1048576).  Default, until this is called: 32768 * 1024 / M rows, clamped to [32768, 131072]. */
int zigp_set_chunk(zigp_ctx* ctx, int64_t chunk_rows);

/* Training data.  Replaces the MinibatchData / DataHolder objects of OnOffSVGP.__init__
 * (onoffgpf/OnOffSVGP.py:38-47) and the X/Y feed_dict of scripts/onoff.py:379-381.
 * zigp_set_data copies host arrays into HBM once; zigp_set_data_device adopts device pointers
 * (e.g. torch tensors) without copying -- the caller keeps them alive. X is (N,D), Y is (N). */
int zigp_set_data(zigp_ctx* ctx, const double* X, const double* Y, int64_t N, int32_t D);
int zigp_set_data_device(zigp_ctx* ctx, const double* dX, const double* dY, int64_t N, int32_t D);

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
int zigp_kron_elbo(zigp_ctx* ctx, const zigp_kron_params* p, const double* X, const double* Y, int64_t N,
                   double jitter, double scale, double g_offset, int32_t include_kl,
                   double* elbo_data, double* kl, zigp_kron_grads* grads);
/* The same step on rows [row_begin, row_end) of the RESIDENT data set (zigp_set_data / zigp_set_data_device, D = D0 + D1): no
 * host->device copy of the minibatch.  The reference's iterator (DataSet.next_batch, onofftf/main.py:98-133) shuffles once per
 * epoch and then hands out contiguous slices, so a host that makes the permuted epoch resident feeds every step by row range;
 * the full-batch configuration (BASELINE cfg5) is rows [0, N). */
int zigp_kron_elbo_rows(zigp_ctx* ctx, const zigp_kron_params* p, int64_t row_begin, int64_t row_end, double jitter, double scale,
                        double g_offset, int32_t include_kl, double* elbo_data, double* kl, zigp_kron_grads* grads);
/* Replaces predict_onoff's graph (onofftf/onoffpred.py:127-200); out9 as zigp_predict. */
int zigp_kron_predict(zigp_ctx* ctx, const zigp_kron_params* p, const double* Xnew, int64_t N, double jitter,
                      double g_offset, double* out9);

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

/* ---- measurement hooks (bench.py) ---- */
/* Diagnostic: eager vs hipGraph-replayed Kronecker minibatch step; out_ms = {eager ms/step, replay ms/step} (tools/kron_graph.py). */
int zigp_test_kron_graph(zigp_ctx* ctx, const zigp_kron_params* p, const double* X, const double* Y, int64_t N,
                         double jitter, double scale, int32_t iters, double* out_ms);
/* Stream overlap inside zigp_elbo (default off): when on, the HBM-bound kernels of a row chunk (Kuf-cotangent reductions,
 * the next chunk's Kuf panels) run on a second HIP stream underneath the chunk's two MFMA-bound rank-N updates.  Results
 * are bit-identical either way and the step is ~0.8 % shorter (tools/overlap_ab.py); it is off by default so that every
 * kernel runs alone on one stream and per-kernel durations (HIP events, rocprofv3 --stats) mean what they say. */
int zigp_set_overlap(zigp_ctx* ctx, int32_t on);   /* 0 off; 1 as above; 2: the per-chunk kernel chains of the latents f and g on two
 * streams (they are independent up to the point-wise stage): measured 1.5 % SLOWER than mode 0 on cfg3 (two full-chip GEMMs in flight
 * share the LDS / L2 rather than fill each other's tails: tools/overlap_ab.py), kept as an option; bit-identical results; ignored while kernel timing (zigp_profile_enable) is on */
/* Accumulated HIP-event time (ms), launch count and algorithmic flops per kernel class since the last reset,
 * measured with HIP events on the stream the kernels run on.  Classes (gemm_f64_kernel template arguments are
 * <A layout, B layout, ring stages, k-scale, triangular mode, waves, epilogue>):
 * 0 gemm_A1  A1 = W K            gemm_f64_kernel<0,1,2,false,1,4,EpiStoreColsum>
 * 1 gemm_A2  A2 = W^T A1         gemm_f64_kernel<1,1,2,false,2,8,EpiStoreColsum>
 * 2 gemm_H   H = W diag(s^2) A2  gemm_f64_kernel<0,1,2,false,1,4,EpiStore>
 * 3 gemm_J   J' = W^T H - A2     gemm_f64_kernel<1,1,2,false,2,8,EpiSubLoad>
 * 4 syrk     C1 += A1 G A1^T     gemm_f64_kernel<0,0,2,true,3,4,EpiAccum>
 * 5 kuf_build   6 pointwise   7 kgrad   8 MxM stage (all kernels)   9 everything else. */
#define ZIGP_NCLASS 10
int zigp_profile_enable(zigp_ctx* ctx, int32_t on);
int zigp_profile_get(zigp_ctx* ctx, double* ms /*[ZIGP_NCLASS]*/, int64_t* launches /*[ZIGP_NCLASS]*/,
                     double* flops /*[ZIGP_NCLASS] algorithmic*/);
int zigp_profile_reset(zigp_ctx* ctx);
/* Event pairs cost ~10 us each, so launches of the chunk loop are TIMED on every 8th full-size chunk only (ms / launches /
 * flops above describe those sampled launches); zigp_profile_totals returns the number of launches per class, sampled or not. */
int zigp_profile_totals(zigp_ctx* ctx, int64_t* total_launches /*[ZIGP_NCLASS]*/);
/* every = 1: time EVERY launch of the chunk loop, the partial last chunk included (sums are then exact, the step is ~1 % slower:
 * bench.py's separate profiled pass); every = n > 1: full-size chunks only, every n-th (default 8). */
int zigp_profile_sampling(zigp_ctx* ctx, int32_t every);

/* ---- diagnostics used by the parity tests (building blocks through the same kernels) ---- */
/* Kronecker entry points: on != 0 forces the GEMM-panel path (zigp_kron.hip) also for grids the fused register-resident kernels
 * (zigp_kronf.hip) cover -- two independent implementations of the same factored algebra that the tests check against each other. */
int zigp_set_kron_panels(zigp_ctx* ctx, int32_t on);
/* C (m,n) = op(A) * op(B) with the fp64 MFMA GEMM core; transA/transB as BLAS; all dims padded internally. */
int zigp_test_gemm(zigp_ctx* ctx, int32_t transA, int32_t transB, int64_t m, int64_t n, int64_t k,
                   const double* A, const double* B, double* C);
/* L = chol(A) (lower), W = L^-1, A is (n,n) SPD; either output may be NULL. */
int zigp_test_potrf_trtri(zigp_ctx* ctx, int64_t n, const double* A, double* L, double* W);

#ifdef __cplusplus
}
#endif
#endif
