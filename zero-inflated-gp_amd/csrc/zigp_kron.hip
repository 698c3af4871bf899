// Kronecker (space x time) variant of the zero-inflated GP ELBO path: scripts/onoff.py:143-319 (build_prior_kl,
// build_predict, kron_inf, cost), onofftf/main.py:350-387 (GaussKLkron), onofftf/onoffpred.py:127-200 (predict).
//
// The reference forms the dense (M0*M1) x Nb matrix Kmn, the dense Kronecker inverse and two Nb x Nb products of which only
// the diagonal is used (scripts/onoff.py:206-211).  Here everything stays factored (identities checked on the literal oracle in
// tests/test_cpu_oracle.py::test_kronecker_literal_equals_factored_identities).  Per latent, with P_p = K_p^-1, U = reshape(u),
// S2 = reshape(s^2), Alpha = P0 U P1, panels K_p = k_p(Z_p, X_p) [M_p][N], A_p = P_p K_p:
//     mean_n = sum_i K0[i,n] (Alpha K1)[i,n]                      (= Kmn^T alpha, :209)
//     var_n  = v0 v1 - (K0.A0)_n (K1.A1)_n + sum_i A0[i,n]^2 (S2 A1^2)[i,n]      (= Knn - diag(Kmn^T A - A^T S A), :210-211)
//     KL     = 1/2 [ sum U.Alpha - M0 M1 - sum log s^2 + sum_ij diag(P0)_i diag(P1)_j s_ij^2 + M1 logdet K0 + M0 logdet K1 ]
// K_p^-1 comes from the blocked Cholesky + triangular inverse of the dense path (the reference uses an LU inverse,
// tf.matrix_inverse :192; the two agree to O(cond * eps)).  All O(M_p^2 N) / O(M0 M1 N) products and every reduction over N run
// on the fp64 MFMA GEMM core (operands padded to 128); the reverse pass is hand-derived.
#include "zigp_host.h"
#include "zigp_comm.h"

using namespace zigp;

namespace zigp {

struct KronFactor {
  int M = 0, Mq = 0, D = 0, col0 = 0;
  double var = 1.0;
  std::vector<double> ell;
  DevBuf Z, K, L, W, P, T, dP, G, krow;
  DevBuf Kp, Ap, Asq, dA, E, PdA, Bx, Cx;   // panels [Mq][Nc]: K_p, A_p, A_p^2, dA_p, E_p, P_p dA_p, (Alpha K_other), (S2 A_other^2)
};

struct KronLatent {
  KronFactor f[2];
  DevBuf U, S, S2, Al, T0, T1, dAl, dS2, dU;   // [Mq0][Mq1]
  DevBuf part;                                 // column sums [4][Nc]: q0, q1, mean, st
  DevBuf gm, gv, dq0, dq1;                     // [Nc]
  DevBuf planes;                               // split-K partial planes
  DevBuf vec;                                  // diag(P0) [Mq0], diag(P1) [Mq1], scalars[8]
};

struct KronState {
  KronLatent lat[2];
  DevBuf X, Y, acc, out9;
};

static void kron_free(KronState* k) {
  for (int h = 0; h < 2; ++h) {
    KronLatent& l = k->lat[h];
    for (int p = 0; p < 2; ++p) {
      KronFactor& f = l.f[p];
      DevBuf* bs[] = {&f.Z, &f.K, &f.L, &f.W, &f.P, &f.T, &f.dP, &f.G, &f.krow, &f.Kp, &f.Ap, &f.Asq, &f.dA, &f.E, &f.PdA, &f.Bx, &f.Cx};
      for (DevBuf* b : bs) b->release();
    }
    DevBuf* bs[] = {&l.U, &l.S, &l.S2, &l.Al, &l.T0, &l.T1, &l.dAl, &l.dS2, &l.dU, &l.part, &l.gm, &l.gv, &l.dq0, &l.dq1, &l.planes, &l.vec};
    for (DevBuf* b : bs) b->release();
  }
  DevBuf* bs[] = {&k->X, &k->Y, &k->acc, &k->out9};
  for (DevBuf* b : bs) b->release();
  delete k;
}

// ------------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------------
// Factor panel: K[m][n] = var * exp(-0.5 |(z_m - x_n[col0:col0+D]) / ell|^2), rows m >= M are 0 (kern.K(Z_p, xnew), :199-201)
__global__ void __launch_bounds__(256)
k_kron_kbuild(const double* __restrict__ X, int64_t N, int ldx, int col0, const double* __restrict__ Z, int M, KernHyp h,
              double* __restrict__ K, int64_t Nc) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int m0 = blockIdx.y * 16;
  double xs[MAXD];
  const bool valid = n < N;
#pragma unroll
  for (int d = 0; d < MAXD; ++d) xs[d] = (d < h.D && valid) ? X[n * ldx + col0 + d] * h.inv_ell[d] : 0.0;
#pragma unroll 4
  for (int mm = 0; mm < 16; ++mm) {
    const int m = m0 + mm;
    double v = 0.0;
    if (m < M) {
      double r2 = 0.0;
#pragma unroll
      for (int d = 0; d < MAXD; ++d)
        if (d < h.D) { double t = Z[m * h.D + d] * h.inv_ell[d] - xs[d]; r2 = fma(t, t, r2); }
      v = h.var * exp(-0.5 * r2);
    }
    K[(int64_t)m * Nc + n] = v;
  }
}

__global__ void k_square_panel(const double* __restrict__ A, double* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double a = A[i]; out[i] = a * a; }
}

// column sums: part[0]=sum_i K0.A0, [1]=sum_j K1.A1, [2]=sum_i K0.B1 (mean), [3]=sum_i A0^2.C1
__global__ void __launch_bounds__(256)
k_kron_colsum(const double* __restrict__ K0, const double* __restrict__ A0, const double* __restrict__ K1, const double* __restrict__ A1,
              const double* __restrict__ B1, const double* __restrict__ A0sq, const double* __restrict__ C1, int M0, int M1, int64_t Nc,
              double* __restrict__ part) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double q0 = 0, q1 = 0, mu = 0, st = 0;
  for (int i = 0; i < M0; ++i) {
    const int64_t o = (int64_t)i * Nc + n;
    q0 = fma(K0[o], A0[o], q0);
    mu = fma(K0[o], B1[o], mu);
    st = fma(A0sq[o], C1[o], st);
  }
  for (int j = 0; j < M1; ++j) {
    const int64_t o = (int64_t)j * Nc + n;
    q1 = fma(K1[o], A1[o], q1);
  }
  part[0 * Nc + n] = q0; part[1 * Nc + n] = q1; part[2 * Nc + n] = mu; part[3 * Nc + n] = st;
}

constexpr int KPW_ACC = 5;
struct KronPwArgs {
  const double* part_f; const double* part_g; const double* Y; int64_t N, Nc;
  double knn_f, knn_g, noise, g_offset, f_offset, scale;   // f_offset: the constant f_mu added to fmean (scripts/onoff.py:168-169, classifier.py:136-137)
  double *gm_f, *gv_f, *gm_g, *gv_g, *dq0_f, *dq1_f, *dq0_g, *dq1_g;
  double* acc;   // [blocks][KPW_ACC]: var_exp, d noise, sum gv_f, sum gv_g, sum gm_f (= d / d f_mu)
  double* out9; int64_t ld9;
  const double* hyp;   // nullable: device hyperparameter block (KH_*): knn_f, knn_g, noise are read from it instead of the fields above
};

template <bool PREDICT>
__global__ void __launch_bounds__(PW_THREADS)
k_kron_pointwise(KronPwArgs p) {
  __shared__ double sh[4];
  const int64_t n = (int64_t)blockIdx.x * PW_THREADS + threadIdx.x;
  // Knn = var_0 var_1 (scripts/onoff.py:196-200): the product of the two factor variances of the latent's records
  const double knn_f = p.hyp ? KF_CONST(p.hyp)[KH_VAR] * KF_CONST(p.hyp)[KH_FAC + KH_VAR] : p.knn_f;
  const double knn_g = p.hyp ? KF_CONST(p.hyp)[2 * KH_FAC + KH_VAR] * KF_CONST(p.hyp)[3 * KH_FAC + KH_VAR] : p.knn_g;
  const double noise = p.hyp ? KF_CONST(p.hyp)[KH_NOISE] : p.noise;
  const double q0f = p.part_f[n], q1f = p.part_f[p.Nc + n], q0g = p.part_g[n], q1g = p.part_g[p.Nc + n];
  const double fm = p.part_f[2 * p.Nc + n] + p.f_offset, fv = knn_f - q0f * q1f + p.part_f[3 * p.Nc + n];
  const double gmn = p.part_g[2 * p.Nc + n] + p.g_offset, gvr = knn_g - q0g * q1g + p.part_g[3 * p.Nc + n];
  const bool valid = n < p.N;
  const double y = (valid && p.Y) ? p.Y[n] : 0.0;
  PwOut o = pointwise_eval(fm, fv, gmn, gvr, y, noise);
  if (PREDICT) {
    if (valid) {
      double* q = p.out9 + n;
      q[0 * p.ld9] = o.gfmean; q[1 * p.ld9] = o.gfvar; q[2 * p.ld9] = o.gfmeanu; q[3 * p.ld9] = fm; q[4 * p.ld9] = fv;
      q[5 * p.ld9] = gmn; q[6 * p.ld9] = gvr; q[7 * p.ld9] = o.e1; q[8 * p.ld9] = o.ev;
    }
    return;
  }
  const double sc = valid ? p.scale : 0.0;
  if (p.gm_f) {
    const double gvf = sc * o.dfv, gvg = sc * o.dgv;
    p.gm_f[n] = sc * o.dfm; p.gv_f[n] = gvf; p.gm_g[n] = sc * o.dgm; p.gv_g[n] = gvg;
    p.dq0_f[n] = -gvf * q1f; p.dq1_f[n] = -gvf * q0f; p.dq0_g[n] = -gvg * q1g; p.dq1_g[n] = -gvg * q0g;
  }
  double s0 = block_sum<4>(valid ? p.scale * o.ve : 0.0, sh);
  double s1 = block_sum<4>(sc * o.dnoise, sh);
  double s2 = block_sum<4>(sc * o.dfv, sh);
  double s3 = block_sum<4>(sc * o.dgv, sh);
  double s4 = block_sum<4>(sc * o.dfm, sh);
  if (threadIdx.x == 0) { double* a = p.acc + (int64_t)blockIdx.x * KPW_ACC; a[0] = s0; a[1] = s1; a[2] = s2; a[3] = s3; a[4] = s4; }
}

// Single-latent heads on the same kron_inf (the reference's baselines):
//   lik 1, Gaussian  (scripts/svgp.py:198-200, hurdle.py:217-219):  ve = -1/2 log 2pi - 1/2 log s2 - 1/2 ((y - fm)^2 + fv) / s2
//   lik 2, Bernoulli (scripts/classifier.py:139-140,210-217):       p = probit(fm / sqrt(1 + fv)),  ve = log(y == 1 ? p : 1 - p)
// with fm = kron mean + f_mu (classifier.py:136-137).  acc[b] = {ve, d noise, sum gv, 0, sum gm (= d f_mu)}.
// Predict rows (ld = N): fmean, fvar, pfmean, pfvar -- Gaussian: pfmean = fm, pfvar = fv + s2 (density of y);
// Bernoulli: pfmean = p, pfvar = p - p^2 (classifier.py:140).
template <bool PREDICT>
__global__ void __launch_bounds__(PW_THREADS)
k_kron_head_pointwise(KronPwArgs p, int lik) {
  __shared__ double sh[4];
  const int64_t n = (int64_t)blockIdx.x * PW_THREADS + threadIdx.x;
  const double q0 = p.part_f[n], q1 = p.part_f[p.Nc + n];
  const double fm = p.part_f[2 * p.Nc + n] + p.f_offset, fv = p.knn_f - q0 * q1 + p.part_f[3 * p.Nc + n];
  const bool valid = n < p.N;
  const double y = (valid && p.Y) ? p.Y[n] : 0.0;
  double ve, dfm, dfv, dnoise = 0.0, pm, pv;
  if (lik == ZIGP_LIK_GAUSSIAN) {
    const double inv = 1.0 / p.noise, res = y - fm, q = res * res + fv;
    ve = -0.5 * 1.8378770664093454836 - 0.5 * log(p.noise) - 0.5 * q * inv;
    dfm = res * inv; dfv = -0.5 * inv; dnoise = -0.5 * inv + 0.5 * q * inv * inv;
    pm = fm; pv = fv + p.noise;
  } else {
    const double c1 = 1.0 - 2.e-3, c0 = 1.e-3;
    const double r = 1.0 / sqrt(1.0 + fv), z = fm * r;
    const double pr = 0.5 * (1.0 + erf(z * 0.70710678118654752440)) * c1 + c0;   // classifier.py:216-217
    const bool on = (y == 1.0);                                                  // tf.equal(y, 1) :214
    ve = log(on ? pr : 1.0 - pr);
    const double dp = on ? 1.0 / pr : -1.0 / (1.0 - pr);
    const double dz = dp * c1 * 0.39894228040143267794 * exp(-0.5 * z * z);
    dfm = dz * r; dfv = dz * (-0.5 * z / (1.0 + fv));
    pm = pr; pv = pr - pr * pr;
  }
  if (PREDICT) {
    if (valid) { double* q = p.out9 + n; q[0] = fm; q[p.ld9] = fv; q[2 * p.ld9] = pm; q[3 * p.ld9] = pv; }
    return;
  }
  const double sc = valid ? p.scale : 0.0;
  if (p.gm_f) {
    const double gvf = sc * dfv;
    p.gm_f[n] = sc * dfm; p.gv_f[n] = gvf; p.dq0_f[n] = -gvf * q1; p.dq1_f[n] = -gvf * q0;
  }
  double s0 = block_sum<4>(sc * ve, sh);
  double s1 = block_sum<4>(sc * dnoise, sh);
  double s2 = block_sum<4>(sc * dfv, sh);
  double s4 = block_sum<4>(sc * dfm, sh);
  if (threadIdx.x == 0) { double* a = p.acc + (int64_t)blockIdx.x * KPW_ACC; a[0] = s0; a[1] = s1; a[2] = s2; a[3] = 0.0; a[4] = s4; }
}

// dA[i][n] = 2 A[i][n] gv[n] C[i][n] ; E[i][n] = dq[n] K[i][n] + dA[i][n]
__global__ void k_kron_da(const double* __restrict__ A, const double* __restrict__ C, const double* __restrict__ K,
                          const double* __restrict__ gv, const double* __restrict__ dq, int64_t Nc, int64_t total,
                          double* __restrict__ dA, double* __restrict__ E) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t n = i % Nc;
  const double d = 2.0 * A[i] * gv[n] * C[i];
  dA[i] = d;
  E[i] = fma(dq[n], K[i], d);
}

// factor-kernel cotangent reductions, block per row m:  dK[m,n] = gm[n] B[m,n] + 2 dq[n] A[m,n] + PdA[m,n]
//   krow[m][0] += sum dK K ; krow[m][1+d] += sum dK K (x_d - z_md) ; krow[m][1+D+d] += sum dK K (x_d - z_md)^2
__global__ void __launch_bounds__(256)
k_kron_kgrad(const double* __restrict__ B, const double* __restrict__ A, const double* __restrict__ PdA, const double* __restrict__ K,
             const double* __restrict__ gm, const double* __restrict__ dq, const double* __restrict__ X, int64_t N, int ldx, int col0,
             const double* __restrict__ Z, int M, int D, int64_t Nc, double* __restrict__ krow) {
  __shared__ double sh[4];
  const int m = blockIdx.x;
  if (m >= M) return;
  double zz[MAXD];
#pragma unroll
  for (int d = 0; d < MAXD; ++d) zz[d] = (d < D) ? Z[m * D + d] : 0.0;
  double s0 = 0.0, s1[MAXD], s2[MAXD];
#pragma unroll
  for (int d = 0; d < MAXD; ++d) { s1[d] = 0.0; s2[d] = 0.0; }
  const int64_t r = (int64_t)m * Nc;
  for (int64_t n = threadIdx.x; n < N; n += 256) {
    const double dk = fma(gm[n], B[r + n], fma(2.0 * dq[n], A[r + n], PdA[r + n]));
    const double t = dk * K[r + n];
    s0 += t;
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
      if (d < D) {
        const double df = X[n * ldx + col0 + d] - zz[d];
        const double td = t * df;
        s1[d] += td;
        s2[d] = fma(td, df, s2[d]);
      }
  }
  const int W = 2 + 2 * D;
  s0 = block_sum<4>(s0, sh);
  if (threadIdx.x == 0) krow[(int64_t)m * W] += s0;
  for (int d = 0; d < D; ++d) {
    double a = block_sum<4>(s1[d], sh);
    double b = block_sum<4>(s2[d], sh);
    if (threadIdx.x == 0) { krow[(int64_t)m * W + 1 + d] += a; krow[(int64_t)m * W + 1 + D + d] += b; }
  }
}

// out[idx] = sum_s planes[s][idx]
__global__ void k_sum_planes(const double* __restrict__ planes, int S, int64_t n, double* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = 0.0;
  for (int s = 0; s < S; ++s) a += planes[(int64_t)s * n + i];
  out[i] = a;
}
// d[i] = A[i][i]
__global__ void k_diag(const double* __restrict__ A, int64_t ld, int n, double* __restrict__ d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = A[(int64_t)i * ld + i];
}
// KL scalars (one block): out[0] = sum U.Al ; out[1] = sum log s^2 ; out[2] = sum d0_i d1_j s2_ij ; out[3] = ld0 ; out[4] = ld1
__global__ void __launch_bounds__(256)
k_kron_kl(const double* __restrict__ U, const double* __restrict__ Al, const double* __restrict__ S, const double* __restrict__ d0,
          const double* __restrict__ d1, const double* __restrict__ L0, const double* __restrict__ L1, int M0, int M1, int Mq0, int Mq1,
          double* __restrict__ out) {
  __shared__ double sh[4];
  double a = 0, b = 0, c = 0, e = 0, f = 0;
  for (int idx = threadIdx.x; idx < M0 * M1; idx += 256) {
    const int i = idx / M1, j = idx - i * M1;
    const int64_t o = (int64_t)i * Mq1 + j;
    const double s = S[o];
    a = fma(U[o], Al[o], a);
    b += log(s * s);
    c = fma(d0[i] * d1[j], s * s, c);
  }
  for (int i = threadIdx.x; i < M0; i += 256) { const double l = L0[(int64_t)i * Mq0 + i]; e += log(l * l); }
  for (int j = threadIdx.x; j < M1; j += 256) { const double l = L1[(int64_t)j * Mq1 + j]; f += log(l * l); }
  a = block_sum<4>(a, sh); b = block_sum<4>(b, sh); c = block_sum<4>(c, sh); e = block_sum<4>(e, sh); f = block_sum<4>(f, sh);
  if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = c; out[3] = e; out[4] = f; }
}
// dP (in place) = sym(dP_data) - kl * ( 0.5 * Q + diag(0.5 * w) ),  w_i = sum_j dother_j s2[i,j] (p = 0) or sum_i dother_i s2[i,j] (p = 1)
__global__ void k_kron_dp_combine(double* __restrict__ dP, const double* __restrict__ Q, const double* __restrict__ s2,
                                  const double* __restrict__ dother, int p, int Mq, int Mqo, int Mo, int64_t lds2, int with_kl) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Mq * Mq) return;
  const int i = (int)(idx / Mq), j = (int)(idx - (int64_t)i * Mq);
  if (j > i) return;   // handle the pair (i,j),(j,i) once
  double v = 0.5 * (dP[(int64_t)i * Mq + j] + dP[(int64_t)j * Mq + i]);
  if (with_kl) {
    v -= 0.25 * (Q[(int64_t)i * Mq + j] + Q[(int64_t)j * Mq + i]);
    if (i == j) {
      double w = 0.0;
      for (int o = 0; o < Mo; ++o) w = fma(dother[o], (p == 0) ? s2[(int64_t)i * lds2 + o] : s2[(int64_t)o * lds2 + i], w);
      v -= 0.5 * w;
    }
  }
  dP[(int64_t)i * Mq + j] = v; dP[(int64_t)j * Mq + i] = v;
}
// G = -(P dP P) - kl * 0.5 * Mother * P      (T holds P dP P)
__global__ void k_kron_dk(const double* __restrict__ T, const double* __restrict__ P, double coef, int64_t n, double* __restrict__ G) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) G[i] = -T[i] - coef * P[i];
}
// u / s gradients:  gu = dU_data - kl*Al ;  gs = 2 s dS2 - kl*(-1/s + d0_i d1_j s)
__global__ void k_kron_us(const double* __restrict__ dUd, const double* __restrict__ Al, const double* __restrict__ dS2,
                          const double* __restrict__ S, const double* __restrict__ d0, const double* __restrict__ d1, int M0, int M1,
                          int Mq1, int with_kl, double* __restrict__ gu, double* __restrict__ gs) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M0 * M1) return;
  const int i = idx / M1, j = idx - i * M1;
  const int64_t o = (int64_t)i * Mq1 + j;
  const double s = S[o];
  double a = dUd[o], b = 2.0 * s * dS2[o];
  if (with_kl) { a -= Al[o]; b -= (-1.0 / s + d0[i] * d1[j] * s); }
  gu[idx] = a; gs[idx] = b;
}

}  // namespace zigp

namespace {

// C[ra x ca blocks] = op(A) op(B) over nkb k-blocks, full tiles
template <int AL, int BL>
int mm(zigp_ctx* c, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int nbi, int nbj, int nkb) {
  TileList tl;
  ZIGP_TRY(tiles_full(c, nbi, nbj, nkb * (BM / BK), tl));
  return run_gemm<AL, BL, false>(c, tl, mk_args(A, lda, B, ldb, C, ldc), EpiStore());
}

// out[Mqa x Mqb] = PA[Mqa][Nc] diag(scale) PB[Mqb][Nc]^T, split-K over S slices + plane sum
int nred(zigp_ctx* c, KronLatent& lt, const double* PA, const double* PB, const double* scale, int nba, int nbb, int64_t Nc, double* out) {
  const int nk = (int)(Nc / BK);
  int S = std::max(1, std::min(nk, 512 / std::max(1, nba * nbb)));
  TileList tl;
  ZIGP_TRY(get_tiles(c, "kr_n:" + std::to_string(nba) + ":" + std::to_string(nbb) + ":" + std::to_string(nk) + ":" + std::to_string(S),
                     [&](std::vector<GemmTile>& v) {
                       for (int s = 0; s < S; ++s)
                         for (int bi = 0; bi < nba; ++bi)
                           for (int bj = 0; bj < nbb; ++bj)
                             v.push_back(mk_tile(bi, bj, (int)((int64_t)nk * s / S), (int)((int64_t)nk * (s + 1) / S), s));
                     }, tl));
  const int64_t plane = (int64_t)nba * BM * nbb * BN;
  ZIGP_ENSURE(c, lt.planes, (size_t)S * plane);
  GemmArgs g = mk_args(PA, Nc, PB, Nc, lt.planes.p, (int64_t)nbb * BN);
  g.slice_stride = plane; g.kscale = scale;
  if (scale) ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_KCONTIG, true>(c, tl, g, EpiStore())));
  else ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_KCONTIG, false>(c, tl, g, EpiStore())));
  hipLaunchKernelGGL(k_sum_planes, dim3(ceil_div(plane, 256)), dim3(256), 0, c->stream, lt.planes.p, S, plane, out);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

struct HostKronLatent {
  int M[2]; const double* Z[2]; const double* ell[2]; double var[2]; const double* u; const double* s;
};

// lik != ZIGP_LIK_ONOFF: single latent, only the f fields (and, for the Gaussian head, noise) are read
int validate_kron(zigp_ctx* c, const zigp_kron_params* p, int lik = ZIGP_LIK_ONOFF) {
  if (!p) return fail_arg(c, "kron params is NULL");
  if (lik != ZIGP_LIK_ONOFF && lik != ZIGP_LIK_GAUSSIAN && lik != ZIGP_LIK_BERNOULLI) return fail_arg(c, "unknown likelihood head");
  const bool two = lik == ZIGP_LIK_ONOFF;
  if (p->M0f <= 0 || p->M1f <= 0 || (two && (p->M0g <= 0 || p->M1g <= 0))) return fail_arg(c, "inducing counts must be positive");
  if (p->D0 <= 0 || p->D1 <= 0 || p->D0 > MAXD || p->D1 > MAXD) return fail_arg(c, "factor dimensions must be in [1, 8]");
  if (!p->Z0f || !p->Z1f || !p->ell0f || !p->ell1f || !p->u_fm || !p->u_fs_sqrt) return fail_arg(c, "NULL pointer in kron params");
  if (two && (!p->Z0g || !p->Z1g || !p->ell0g || !p->ell1g || !p->u_gm || !p->u_gs_sqrt)) return fail_arg(c, "NULL pointer in kron params");
  if (!(p->var0f > 0) || !(p->var1f > 0) || (two && (!(p->var0g > 0) || !(p->var1g > 0)))) return fail_arg(c, "variances must be positive");
  if (lik != ZIGP_LIK_BERNOULLI && !(p->noise > 0)) return fail_arg(c, "variances must be positive");
  return 0;
}

// factor MxM forward: K_p (+jitter), L_p, W_p, P_p = W^T W
int factor_forward(zigp_ctx* c, KronFactor& f, int M, int D, int col0, const double* Z, const double* ell, double var, double jitter) {
  f.M = M; f.Mq = (int)round_up(M, BM); f.D = D; f.col0 = col0; f.var = var;
  f.ell.assign(ell, ell + D);
  const int Mq = f.Mq, nb = Mq / BM, kb = BM / BK;
  const size_t mm_ = (size_t)Mq * Mq;
  ZIGP_TRY(upload_padded(c, f.Z, Z, (size_t)M * D, (size_t)Mq * D));
  ZIGP_ENSURE(c, f.K, mm_); ZIGP_ENSURE(c, f.L, mm_); ZIGP_ENSURE(c, f.W, mm_); ZIGP_ENSURE(c, f.P, mm_); ZIGP_ENSURE(c, f.T, mm_);
  KernHyp hyp = make_hyp(ell, var, D);
  hipLaunchKernelGGL(k_rbf_matrix, dim3(ceil_div((int64_t)mm_, 256)), dim3(256), 0, c->stream, f.Z.p, (int64_t)M, f.Z.p, (int64_t)M, hyp, jitter,
                     f.K.p, (int64_t)Mq, (int64_t)Mq, (int64_t)Mq);
  ZIGP_HIP(c, hipGetLastError());
  ZIGP_HIP(c, hipMemcpyAsync(f.L.p, f.K.p, sizeof(double) * mm_, hipMemcpyDeviceToDevice, c->stream));
  ZIGP_TRY(potrf_trtri(c, f.L.p, f.W.p, f.T.p, Mq, true, M, pivot_tol(var, jitter, c->pivot_rtol)));
  TileList t;
  ZIGP_TRY(get_tiles(c, "bw_s:" + std::to_string(nb), [&](std::vector<GemmTile>& v) {
    for (int bi = 0; bi < nb; ++bi)
      for (int bj = 0; bj < nb; ++bj) v.push_back(mk_tile(bi, bj, std::max(bi, bj) * kb, nb * kb));
  }, t));
  ZIGP_TRY((run_gemm<LAY_MNCONTIG, LAY_MNCONTIG, false>(c, t, mk_args(f.W.p, Mq, f.W.p, Mq, f.P.p, Mq), EpiStore())));
  return 0;
}

int upload_grid(zigp_ctx* c, DevBuf& b, const double* src, int M0, int M1, int Mq0, int Mq1, bool square) {
  const size_t n = (size_t)Mq0 * Mq1;
  ZIGP_PINNED(c, h, n);   // padded image staged in page-locked memory: the copy is asynchronous
  memset(h, 0, sizeof(double) * n);
  for (int i = 0; i < M0; ++i)
    for (int j = 0; j < M1; ++j) { const double v = src[(size_t)i * M1 + j]; h[(size_t)i * Mq1 + j] = square ? v * v : v; }
  ZIGP_ENSURE(c, b, n);
  ZIGP_HIP(c, hipMemcpyAsync(b.p, h, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  return 0;
}

int latent_setup(zigp_ctx* c, KronLatent& lt, const HostKronLatent& h, int D0, int D1, double jitter) {
  ZIGP_TRY(factor_forward(c, lt.f[0], h.M[0], D0, 0, h.Z[0], h.ell[0], h.var[0], jitter));
  ZIGP_TRY(factor_forward(c, lt.f[1], h.M[1], D1, D0, h.Z[1], h.ell[1], h.var[1], jitter));
  const int M0 = h.M[0], M1 = h.M[1], Mq0 = lt.f[0].Mq, Mq1 = lt.f[1].Mq, nb0 = Mq0 / BM, nb1 = Mq1 / BM;
  ZIGP_TRY(upload_grid(c, lt.U, h.u, M0, M1, Mq0, Mq1, false));
  ZIGP_TRY(upload_grid(c, lt.S, h.s, M0, M1, Mq0, Mq1, false));
  ZIGP_TRY(upload_grid(c, lt.S2, h.s, M0, M1, Mq0, Mq1, true));
  const size_t g = (size_t)Mq0 * Mq1;
  ZIGP_ENSURE(c, lt.Al, g); ZIGP_ENSURE(c, lt.T0, g); ZIGP_ENSURE(c, lt.T1, g);
  // Alpha = P0 (U P1) : T0 = U P1 ; Al = P0 T0
  ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.U.p, Mq1, lt.f[1].P.p, Mq1, lt.T0.p, Mq1, nb0, nb1, nb1)));
  ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.f[0].P.p, Mq0, lt.T0.p, Mq1, lt.Al.p, Mq1, nb0, nb1, nb0)));
  ZIGP_ENSURE(c, lt.vec, (size_t)Mq0 + Mq1 + 8);
  hipLaunchKernelGGL(k_diag, dim3(ceil_div(Mq0, 256)), dim3(256), 0, c->stream, lt.f[0].P.p, (int64_t)Mq0, Mq0, lt.vec.p);
  hipLaunchKernelGGL(k_diag, dim3(ceil_div(Mq1, 256)), dim3(256), 0, c->stream, lt.f[1].P.p, (int64_t)Mq1, Mq1, lt.vec.p + Mq0);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

// forward panels + column sums for one latent
int latent_forward_panels(zigp_ctx* c, KronLatent& lt, const double* dX, int64_t N, int64_t Nc, int ldx) {
  const int nbn = (int)(Nc / BN);
  for (int p = 0; p < 2; ++p) {
    KronFactor& f = lt.f[p];
    const size_t pn = (size_t)f.Mq * Nc;
    ZIGP_ENSURE(c, f.Kp, pn); ZIGP_ENSURE(c, f.Ap, pn); ZIGP_ENSURE(c, f.Asq, pn);
    KernHyp hyp = make_hyp(f.ell.data(), f.var, f.D);
    hipLaunchKernelGGL(k_kron_kbuild, dim3((unsigned)(Nc / 256), f.Mq / 16), dim3(256), 0, c->stream, dX, N, ldx, f.col0, f.Z.p, f.M, hyp,
                       f.Kp.p, Nc);
    ZIGP_HIP(c, hipGetLastError());
    // A_p = P_p K_p
    ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, f.P.p, f.Mq, f.Kp.p, Nc, f.Ap.p, Nc, f.Mq / BM, nbn, f.Mq / BM)));
    hipLaunchKernelGGL(k_square_panel, dim3(ceil_div((int64_t)pn, 256)), dim3(256), 0, c->stream, f.Ap.p, f.Asq.p, (int64_t)pn);
  }
  KronFactor &f0 = lt.f[0], &f1 = lt.f[1];
  ZIGP_ENSURE(c, f0.Bx, (size_t)f0.Mq * Nc); ZIGP_ENSURE(c, f0.Cx, (size_t)f0.Mq * Nc);
  // B1 = Alpha K1 ; C1 = S2 A1^2   (both [Mq0][Nc])
  ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.Al.p, f1.Mq, f1.Kp.p, Nc, f0.Bx.p, Nc, f0.Mq / BM, nbn, f1.Mq / BM)));
  ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.S2.p, f1.Mq, f1.Asq.p, Nc, f0.Cx.p, Nc, f0.Mq / BM, nbn, f1.Mq / BM)));
  ZIGP_ENSURE(c, lt.part, (size_t)4 * Nc);
  hipLaunchKernelGGL(k_kron_colsum, dim3((unsigned)(Nc / 256)), dim3(256), 0, c->stream, f0.Kp.p, f0.Ap.p, f1.Kp.p, f1.Ap.p, f0.Bx.p, f0.Asq.p,
                     f0.Cx.p, f0.M, f1.M, Nc, lt.part.p);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

int latent_backward(zigp_ctx* c, KronLatent& lt, const double* dX, int64_t N, int64_t Nc, int ldx, bool with_kl) {
  KronFactor &f0 = lt.f[0], &f1 = lt.f[1];
  const int nbn = (int)(Nc / BN), nb0 = f0.Mq / BM, nb1 = f1.Mq / BM;
  const int Mq0 = f0.Mq, Mq1 = f1.Mq;
  // B0 = Alpha^T K0 ; C0 = S2^T A0^2   ([Mq1][Nc])
  ZIGP_ENSURE(c, f1.Bx, (size_t)Mq1 * Nc); ZIGP_ENSURE(c, f1.Cx, (size_t)Mq1 * Nc);
  ZIGP_TRY((mm<LAY_MNCONTIG, LAY_MNCONTIG>(c, lt.Al.p, Mq1, f0.Kp.p, Nc, f1.Bx.p, Nc, nb1, nbn, nb0)));
  ZIGP_TRY((mm<LAY_MNCONTIG, LAY_MNCONTIG>(c, lt.S2.p, Mq1, f0.Asq.p, Nc, f1.Cx.p, Nc, nb1, nbn, nb0)));
  double* dq[2] = {lt.dq0.p, lt.dq1.p};
  for (int p = 0; p < 2; ++p) {
    KronFactor& f = lt.f[p];
    const size_t pn = (size_t)f.Mq * Nc;
    ZIGP_ENSURE(c, f.dA, pn); ZIGP_ENSURE(c, f.E, pn); ZIGP_ENSURE(c, f.PdA, pn);
    ZIGP_ENSURE(c, f.krow, (size_t)f.Mq * (2 + 2 * f.D));
    ZIGP_HIP(c, hipMemsetAsync(f.krow.p, 0, sizeof(double) * f.Mq * (2 + 2 * f.D), c->stream));
    hipLaunchKernelGGL(k_kron_da, dim3(ceil_div((int64_t)pn, 256)), dim3(256), 0, c->stream, f.Ap.p, f.Cx.p, f.Kp.p, lt.gv.p, dq[p], Nc,
                       (int64_t)pn, f.dA.p, f.E.p);
    ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, f.P.p, f.Mq, f.dA.p, Nc, f.PdA.p, Nc, f.Mq / BM, nbn, f.Mq / BM)));
    hipLaunchKernelGGL(k_kron_kgrad, dim3(f.Mq), dim3(256), 0, c->stream, f.Bx.p, f.Ap.p, f.PdA.p, f.Kp.p, lt.gm.p, dq[p], dX, N, ldx, f.col0,
                       f.Z.p, f.M, f.D, Nc, f.krow.p);
    ZIGP_HIP(c, hipGetLastError());
    // dP_p (data) = E_p K_p^T
    ZIGP_ENSURE(c, f.dP, (size_t)f.Mq * f.Mq);
    ZIGP_TRY(nred(c, lt, f.E.p, f.Kp.p, nullptr, f.Mq / BM, f.Mq / BM, Nc, f.dP.p));
  }
  const size_t g = (size_t)Mq0 * Mq1;
  ZIGP_ENSURE(c, lt.dAl, g); ZIGP_ENSURE(c, lt.dS2, g); ZIGP_ENSURE(c, lt.dU, g);
  // dAlpha = K0 diag(gm) K1^T ; dS2 = A0^2 diag(gv) (A1^2)^T
  ZIGP_TRY(nred(c, lt, f0.Kp.p, f1.Kp.p, lt.gm.p, nb0, nb1, Nc, lt.dAl.p));
  ZIGP_TRY(nred(c, lt, f0.Asq.p, f1.Asq.p, lt.gv.p, nb0, nb1, Nc, lt.dS2.p));
  // ---- MxM: Alpha = P0 U P1 ----
  // dU = P0 dAl P1 : T1 = dAl P1 ; dU = P0 T1
  ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.dAl.p, Mq1, f1.P.p, Mq1, lt.T1.p, Mq1, nb0, nb1, nb1)));
  ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, f0.P.p, Mq0, lt.T1.p, Mq1, lt.dU.p, Mq1, nb0, nb1, nb0)));
  // dP0 += dAl (U P1)^T = dAl T0^T  (T0 = U P1 from setup) ; dP1 += (P0 U)^T dAl
  {
    TileList t;
    ZIGP_TRY(tiles_full(c, nb0, nb0, nb1 * (BM / BK), t));
    ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_KCONTIG, false>(c, t, mk_args(lt.dAl.p, Mq1, lt.T0.p, Mq1, f0.dP.p, Mq0), EpiAccum())));
    // T1 = P0 U  ([Mq0][Mq1]) ; dP1 += T1^T dAl
    ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, f0.P.p, Mq0, lt.U.p, Mq1, lt.T1.p, Mq1, nb0, nb1, nb0)));
    TileList t2;
    ZIGP_TRY(tiles_full(c, nb1, nb1, nb0 * (BM / BK), t2));
    ZIGP_TRY((run_gemm<LAY_MNCONTIG, LAY_MNCONTIG, false>(c, t2, mk_args(lt.T1.p, Mq1, lt.dAl.p, Mq1, f1.dP.p, Mq1), EpiAccum())));
  }
  // KL pieces on P: Q0 = U P1 U^T = T0 U^T ; Q1 = U^T P0 U = U^T T1(P0 U)
  for (int p = 0; p < 2; ++p) {
    KronFactor& f = lt.f[p];
    const int Mq = f.Mq, nb = Mq / BM;
    ZIGP_ENSURE(c, f.G, (size_t)Mq * Mq);
    if (with_kl) {
      if (p == 0) {
        TileList t; ZIGP_TRY(tiles_full(c, nb0, nb0, nb1 * (BM / BK), t));
        ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_KCONTIG, false>(c, t, mk_args(lt.T0.p, Mq1, lt.U.p, Mq1, f.G.p, Mq0), EpiStore())));
      } else {
        TileList t; ZIGP_TRY(tiles_full(c, nb1, nb1, nb0 * (BM / BK), t));
        ZIGP_TRY((run_gemm<LAY_MNCONTIG, LAY_MNCONTIG, false>(c, t, mk_args(lt.U.p, Mq1, lt.T1.p, Mq1, f.G.p, Mq1), EpiStore())));
      }
    }
    const KronFactor& fo = lt.f[1 - p];
    hipLaunchKernelGGL(k_kron_dp_combine, dim3(ceil_div((int64_t)Mq * Mq, 256)), dim3(256), 0, c->stream, f.dP.p, f.G.p, lt.S2.p,
                       lt.vec.p + (p == 0 ? Mq0 : 0), p, Mq, fo.Mq, fo.M, (int64_t)Mq1, with_kl ? 1 : 0);
    // G = -(P dP P) - kl * 0.5 * M_other * P :  T = dP P ; Kt = P T
    ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, f.dP.p, Mq, f.P.p, Mq, f.T.p, Mq, nb, nb, nb)));
    ZIGP_TRY((mm<LAY_KCONTIG, LAY_MNCONTIG>(c, f.P.p, Mq, f.T.p, Mq, f.L.p, Mq, nb, nb, nb)));   // L is free after setup? no: keep logdet source
    ZIGP_HIP(c, hipGetLastError());
  }
  return 0;
}

}  // namespace

#include "zigp_kronf.hip"

namespace {

int kron_run_panels(zigp_ctx* c, const zigp_kron_params* p, const double* X, const double* Y, int64_t N, double jitter, double scale,
                    double g_offset, double f_mu, int include_kl, bool predict, double* out9, double* elbo_data, double* kl, zigp_kron_grads* grads,
                    int lik, double* d_offset, bool dev_xy);

// Data-parallel exchange of the PANEL path (grids beyond the fused kernels' capacity): its results reach the host piecewise, so with a
// communicator (zigp_comm_init) the assembled outputs take one more round trip -- packed, summed over ranks on the device, unpacked.
// (The fused path reduces its device result block in place before its single download.)
int kron_exchange_host(zigp_ctx* c, const zigp_kron_params* p, int nlat, double* elbo_data, double* kl, zigp_kron_grads* g, double* d_offset) {
  if (!c->comm) return 0;
  struct Piece { double* p; size_t n; };
  std::vector<Piece> pc;
  double scal[11] = {elbo_data ? *elbo_data : 0.0, kl ? *kl : 0.0, d_offset ? *d_offset : 0.0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (g) { scal[3] = g->var0f; scal[4] = g->var1f; scal[5] = g->var0g; scal[6] = g->var1g; scal[7] = g->noise; }
  pc.push_back({scal, 11});
  if (g) {
    const int M0[2] = {p->M0f, p->M0g}, M1[2] = {p->M1f, p->M1g};
    double* Z0[2] = {g->Z0f, g->Z0g}; double* Z1[2] = {g->Z1f, g->Z1g}; double* l0[2] = {g->ell0f, g->ell0g}; double* l1[2] = {g->ell1f, g->ell1g};
    double* um[2] = {g->u_fm, g->u_gm}; double* us[2] = {g->u_fs_sqrt, g->u_gs_sqrt};
    for (int h = 0; h < nlat; ++h) {
      if (Z0[h]) pc.push_back({Z0[h], (size_t)M0[h] * p->D0});
      if (Z1[h]) pc.push_back({Z1[h], (size_t)M1[h] * p->D1});
      if (l0[h]) pc.push_back({l0[h], (size_t)p->D0});
      if (l1[h]) pc.push_back({l1[h], (size_t)p->D1});
      if (um[h]) pc.push_back({um[h], (size_t)M0[h] * M1[h]});
      if (us[h]) pc.push_back({us[h], (size_t)M0[h] * M1[h]});
    }
  }
  size_t n = 0;
  for (auto& q : pc) n += q.n;
  ZIGP_TRY(begin_staged_call(c));
  ZIGP_ENSURE(c, c->packed, n);
  ZIGP_PINNED(c, hin, n);
  size_t o = 0;
  for (auto& q : pc) { memcpy(hin + o, q.p, sizeof(double) * q.n); o += q.n; }
  ZIGP_HIP(c, hipMemcpyAsync(c->packed.p, hin, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  ZIGP_TRY(comm_allreduce(c, c->packed.p, n));
  double* hout = nullptr;
  ZIGP_TRY(download(c, c->packed.p, n, &hout));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  o = 0;
  for (auto& q : pc) { memcpy(q.p, hout + o, sizeof(double) * q.n); o += q.n; }
  if (elbo_data) *elbo_data = scal[0];
  if (kl) *kl = scal[1];
  if (d_offset) *d_offset = scal[2];
  if (g) { g->var0f = scal[3]; g->var1f = scal[4]; g->var0g = scal[5]; g->var1g = scal[6]; g->noise = scal[7]; }
  return 0;
}

// small grids go through the fused register-resident kernels (zigp_kronf.hip); larger factors through the panel path below
// g_offset: added to gmean before the probit moments (onofftf/onoffpred.py:141); f_mu: constant added to fmean (scripts/onoff.py:168-169,
// classifier.py:136-137), its gradient (sum of the fmean cotangents) goes to *d_offset
int kron_run(zigp_ctx* c, const zigp_kron_params* p, const double* X, const double* Y, int64_t N, double jitter, double scale,
             double g_offset, double f_mu, int include_kl, bool predict, double* out9, double* elbo_data, double* kl, zigp_kron_grads* grads,
             int lik = ZIGP_LIK_ONOFF, double* d_offset = nullptr, bool dev_xy = false) {
  const int nlat = (lik == ZIGP_LIK_ONOFF) ? 2 : 1;
  if (!c->kron_panels && kf_eligible(p, nlat))
    return kronf_run(c, p, X, Y, N, jitter, scale, g_offset, f_mu, include_kl, predict, out9, elbo_data, kl, grads, lik, d_offset, dev_xy);
  return kron_run_panels(c, p, X, Y, N, jitter, scale, g_offset, f_mu, include_kl, predict, out9, elbo_data, kl, grads, lik, d_offset, dev_xy);
}

int kron_run_panels(zigp_ctx* c, const zigp_kron_params* p, const double* X, const double* Y, int64_t N, double jitter, double scale,
                    double g_offset, double f_mu, int include_kl, bool predict, double* out9, double* elbo_data, double* kl, zigp_kron_grads* grads,
                    int lik, double* d_offset, bool dev_xy) {
  const int nlat = (lik == ZIGP_LIK_ONOFF) ? 2 : 1;   // single-latent heads use the f latent only
  if (!c->kron) { c->kron = new (std::nothrow) KronState(); c->kron_free = kron_free; if (!c->kron) { c->err = "out of memory"; return ZIGP_EHIP; } }
  KronState& ks = *c->kron;
  ZIGP_TRY(begin_staged_call(c));
  const bool need_grad = grads != nullptr && !predict;
  const int D0 = p->D0, D1 = p->D1, ldx = D0 + D1;
  const int64_t Nc = std::max<int64_t>(1024, round_up(N, 1024));
  if (dev_xy) {   // rows of the resident data set: device-to-device
    ZIGP_ENSURE(c, ks.X, (size_t)N * ldx); ZIGP_ENSURE(c, ks.Y, (size_t)N);
    ZIGP_HIP(c, hipMemcpyAsync(ks.X.p, X, sizeof(double) * N * ldx, hipMemcpyDeviceToDevice, c->stream));
    if (Y) ZIGP_HIP(c, hipMemcpyAsync(ks.Y.p, Y, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
  } else {
    ZIGP_TRY(upload_padded(c, ks.X, X, (size_t)N * ldx, (size_t)N * ldx));   // the minibatch, staged like the parameters
    if (Y) ZIGP_TRY(upload_padded(c, ks.Y, Y, (size_t)N, (size_t)N));
  }
  HostKronLatent hl[2] = {{{p->M0f, p->M1f}, {p->Z0f, p->Z1f}, {p->ell0f, p->ell1f}, {p->var0f, p->var1f}, p->u_fm, p->u_fs_sqrt},
                          {{p->M0g, p->M1g}, {p->Z0g, p->Z1g}, {p->ell0g, p->ell1g}, {p->var0g, p->var1g}, p->u_gm, p->u_gs_sqrt}};
  if (nlat == 1) hl[1] = hl[0];
  ZIGP_HIP(c, hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream));
  {
    TwoStream ts(c);   // the two latents are independent launch chains of small kernels: f on the main stream, g on stream2
    ZIGP_TRY(ts.fork());
    for (int h = 0; h < nlat; ++h) {
      if (h == 1) ts.second();
      ZIGP_TRY(latent_setup(c, ks.lat[h], hl[h], D0, D1, jitter));
    }
    ZIGP_TRY(ts.join());
  }
  int* hinfo = nullptr;
  ZIGP_TRY(request_info(c, &hinfo));   // read after the final synchronisation
  // KL scalars (value) -- before the backward pass overwrites nothing it needs
  double* hkl[2] = {nullptr, nullptr};
  if (include_kl && !predict) {
    for (int h = 0; h < nlat; ++h) {
      KronLatent& lt = ks.lat[h];
      const int Mq0 = lt.f[0].Mq, Mq1 = lt.f[1].Mq;
      hipLaunchKernelGGL(k_kron_kl, dim3(1), dim3(256), 0, c->stream, lt.U.p, lt.Al.p, lt.S.p, lt.vec.p, lt.vec.p + Mq0, lt.f[0].L.p, lt.f[1].L.p,
                         lt.f[0].M, lt.f[1].M, Mq0, Mq1, lt.vec.p + Mq0 + Mq1);
      ZIGP_TRY(download(c, lt.vec.p + Mq0 + Mq1, 8, &hkl[h]));
    }
  }
  {
    TwoStream ts(c);
    ZIGP_TRY(ts.fork());
    for (int h = 0; h < nlat; ++h) {
      if (h == 1) ts.second();
      KronLatent& lt = ks.lat[h];
      ZIGP_TRY(latent_forward_panels(c, lt, ks.X.p, N, Nc, ldx));
      ZIGP_ENSURE(c, lt.gm, Nc); ZIGP_ENSURE(c, lt.gv, Nc); ZIGP_ENSURE(c, lt.dq0, Nc); ZIGP_ENSURE(c, lt.dq1, Nc);
    }
    ZIGP_TRY(ts.join());
  }
  const int blocks = (int)(Nc / PW_THREADS);
  ZIGP_ENSURE(c, ks.acc, (size_t)blocks * KPW_ACC);
  KronPwArgs a;
  const int gl_ = nlat - 1;   // latent whose buffers stand in for g (unused by the single-latent kernels)
  a.part_f = ks.lat[0].part.p; a.part_g = ks.lat[gl_].part.p; a.Y = Y ? ks.Y.p : nullptr; a.N = N; a.Nc = Nc;
  a.knn_f = p->var0f * p->var1f; a.knn_g = p->var0g * p->var1g; a.noise = p->noise; a.g_offset = g_offset; a.f_offset = f_mu; a.scale = scale;
  a.gm_f = need_grad ? ks.lat[0].gm.p : nullptr; a.gv_f = ks.lat[0].gv.p; a.gm_g = ks.lat[gl_].gm.p; a.gv_g = ks.lat[gl_].gv.p;
  a.dq0_f = ks.lat[0].dq0.p; a.dq1_f = ks.lat[0].dq1.p; a.dq0_g = ks.lat[gl_].dq0.p; a.dq1_g = ks.lat[gl_].dq1.p;
  a.acc = ks.acc.p; a.out9 = nullptr; a.ld9 = N; a.hyp = nullptr;
  if (predict) {
    const int rows = nlat == 2 ? 9 : 4;
    ZIGP_ENSURE(c, ks.out9, (size_t)rows * N);
    a.out9 = ks.out9.p;
    if (nlat == 2) hipLaunchKernelGGL(k_kron_pointwise<true>, dim3(blocks), dim3(PW_THREADS), 0, c->stream, a);
    else hipLaunchKernelGGL(k_kron_head_pointwise<true>, dim3(blocks), dim3(PW_THREADS), 0, c->stream, a, lik);
    ZIGP_HIP(c, hipGetLastError());
    ZIGP_HIP(c, hipMemcpyAsync(out9, ks.out9.p, sizeof(double) * rows * N, hipMemcpyDeviceToHost, c->stream));
    ZIGP_HIP(c, hipStreamSynchronize(c->stream));
    return info_result(c, hinfo, "a Kronecker factor of Kuu");
  }
  if (nlat == 2) hipLaunchKernelGGL(k_kron_pointwise<false>, dim3(blocks), dim3(PW_THREADS), 0, c->stream, a);
  else hipLaunchKernelGGL(k_kron_head_pointwise<false>, dim3(blocks), dim3(PW_THREADS), 0, c->stream, a, lik);
  ZIGP_HIP(c, hipGetLastError());
  double* hacc = nullptr;
  ZIGP_TRY(download(c, ks.acc.p, (size_t)blocks * KPW_ACC, &hacc));

  double *hkrow[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *hgu[2] = {nullptr, nullptr}, *hgs[2] = {nullptr, nullptr};
  if (need_grad) {
    TwoStream ts(c);
    ZIGP_TRY(ts.fork());
    for (int h = 0; h < nlat; ++h) {
      if (h == 1) ts.second();
      KronLatent& lt = ks.lat[h];
      ZIGP_TRY(latent_backward(c, lt, ks.X.p, N, Nc, ldx, include_kl != 0));
      const int M0 = lt.f[0].M, M1 = lt.f[1].M, Mq0 = lt.f[0].Mq, Mq1 = lt.f[1].Mq;
      for (int q = 0; q < 2; ++q) {
        KronFactor& f = lt.f[q];
        const int Mq = f.Mq;
        // G = -(P dP P) - kl*0.5*M_other*P  -> f.G ; then Kuu-style reductions into krow
        hipLaunchKernelGGL(k_kron_dk, dim3(ceil_div((int64_t)Mq * Mq, 256)), dim3(256), 0, c->stream, f.L.p, f.P.p,
                           include_kl ? 0.5 * (double)lt.f[1 - q].M : 0.0, (int64_t)Mq * Mq, f.G.p);
        hipLaunchKernelGGL(k_kuu_grad, dim3(Mq), dim3(256), 0, c->stream, f.G.p, f.K.p, jitter, f.Z.p, f.M, f.D, (int64_t)Mq, f.krow.p);
        ZIGP_HIP(c, hipGetLastError());
        ZIGP_TRY(download(c, f.krow.p, (size_t)Mq * (2 + 2 * f.D), &hkrow[h][q]));
      }
      // u, s gradients (reuse T0 / T1 as outputs)
      hipLaunchKernelGGL(k_kron_us, dim3(ceil_div((int64_t)M0 * M1, 256)), dim3(256), 0, c->stream, lt.dU.p, lt.Al.p, lt.dS2.p, lt.S.p, lt.vec.p,
                         lt.vec.p + Mq0, M0, M1, Mq1, include_kl ? 1 : 0, lt.T0.p, lt.T1.p);
      ZIGP_HIP(c, hipGetLastError());
      ZIGP_TRY(download(c, lt.T0.p, (size_t)M0 * M1, &hgu[h]));
      ZIGP_TRY(download(c, lt.T1.p, (size_t)M0 * M1, &hgs[h]));
    }
    ZIGP_TRY(ts.join());
  }
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  ZIGP_TRY(info_result(c, hinfo, "a Kronecker factor of Kuu"));
  double s_ve = 0, s_dn = 0, s_gv[2] = {0, 0}, s_gm = 0;
  for (int b = 0; b < blocks; ++b) {
    const double* r = hacc + (size_t)KPW_ACC * b;
    s_ve += r[0]; s_dn += r[1]; s_gv[0] += r[2]; s_gv[1] += r[3]; s_gm += r[4];
  }
  if (elbo_data) *elbo_data = s_ve;
  if (d_offset) *d_offset = s_gm;
  double klsum = 0.0;
  if (include_kl) {
    for (int h = 0; h < nlat; ++h) {
      const int M0 = ks.lat[h].f[0].M, M1 = ks.lat[h].f[1].M;
      klsum += 0.5 * (hkl[h][0] - (double)M0 * M1 - hkl[h][1] + hkl[h][2] + (double)M1 * hkl[h][3] + (double)M0 * hkl[h][4]);
    }
  }
  if (kl) *kl = klsum;
  if (need_grad) {
    double* gZ[2][2] = {{grads->Z0f, grads->Z1f}, {grads->Z0g, grads->Z1g}};
    double* gl[2][2] = {{grads->ell0f, grads->ell1f}, {grads->ell0g, grads->ell1g}};
    double gvar[2][2];
    double* gu[2] = {grads->u_fm, grads->u_gm};
    double* gs[2] = {grads->u_fs_sqrt, grads->u_gs_sqrt};
    for (int h = 0; h < nlat; ++h) {
      for (int q = 0; q < 2; ++q) {
        const KronFactor& f = ks.lat[h].f[q];
        const int D = f.D, W = 2 + 2 * D;
        double dv = 0.0;
        std::vector<double> dl(D, 0.0);
        for (int m = 0; m < f.M; ++m) {
          const double* r = &hkrow[h][q][(size_t)m * W];
          dv += r[0];
          for (int d = 0; d < D; ++d) {
            if (gZ[h][q]) gZ[h][q][m * D + d] = r[1 + d] / (f.ell[d] * f.ell[d]);
            dl[d] += r[1 + D + d];
          }
        }
        for (int d = 0; d < D; ++d)
          if (gl[h][q]) gl[h][q][d] = dl[d] / (f.ell[d] * f.ell[d] * f.ell[d]);
        // Knn = var0 * var1 enters var_n directly (scripts/onoff.py:196-200)
        gvar[h][q] = dv / f.var + s_gv[h] * ks.lat[h].f[1 - q].var;
      }
      const size_t ng = (size_t)ks.lat[h].f[0].M * ks.lat[h].f[1].M;
      if (gu[h]) memcpy(gu[h], hgu[h], sizeof(double) * ng);
      if (gs[h]) memcpy(gs[h], hgs[h], sizeof(double) * ng);
    }
    grads->var0f = gvar[0][0]; grads->var1f = gvar[0][1];
    grads->var0g = nlat == 2 ? gvar[1][0] : 0.0; grads->var1g = nlat == 2 ? gvar[1][1] : 0.0;
    grads->noise = s_dn;
  }
  return kron_exchange_host(c, p, nlat, elbo_data, kl, need_grad ? grads : nullptr, d_offset);
}

}  // namespace

namespace {
// Prediction sets can be much larger than a training minibatch (predict_onoff runs over the full Xtrain / Xtest): rows go through
// the path in chunks, so device panels and the pinned staging arena stay bounded; the factor stage is recomputed per chunk (microseconds).
constexpr int64_t KRON_PREDICT_CHUNK = 131072;
int kron_predict_chunked(zigp_ctx* c, const zigp_kron_params* p, const double* Xnew, int64_t N, double jitter, double g_offset, double f_mu,
                         double* out, int lik, int rows) {
  if (N <= KRON_PREDICT_CHUNK) return kron_run(c, p, Xnew, nullptr, N, jitter, 1.0, g_offset, f_mu, 0, true, out, nullptr, nullptr, nullptr, lik, nullptr);
  const int ldx = p->D0 + p->D1;
  std::vector<double> tmp((size_t)rows * KRON_PREDICT_CHUNK);
  for (int64_t n0 = 0; n0 < N; n0 += KRON_PREDICT_CHUNK) {
    const int64_t nc = std::min(KRON_PREDICT_CHUNK, N - n0);
    ZIGP_TRY(kron_run(c, p, Xnew + n0 * ldx, nullptr, nc, jitter, 1.0, g_offset, f_mu, 0, true, tmp.data(), nullptr, nullptr, nullptr, lik, nullptr));
    for (int r = 0; r < rows; ++r) memcpy(out + (size_t)r * N + n0, tmp.data() + (size_t)r * nc, sizeof(double) * nc);
  }
  return ZIGP_OK;
}
// An EMPTY row set (a rank of a data-parallel run whose shard is empty) still has to make the same calls as its peers -- the exchange
// inside the step is collective.  It is evaluated as ONE row at the origin with scale 0: the data term and every data-term gradient
// are exactly zero, the KL part follows include_kl as usual.
static const double kron_no_rows[2 * MAXD] = {0};
}  // namespace

extern "C" {

int zigp_kron_elbo(zigp_ctx* c, const zigp_kron_params* p, const double* X, const double* Y, int64_t N, double jitter, double scale,
                   double g_offset, double f_mu, int32_t include_kl, double* elbo_data, double* kl, zigp_kron_grads* grads, double* d_f_mu) {
  if (!c) return ZIGP_EARG;
  ZIGP_TRY(validate_kron(c, p));
  if (N < 0 || (N > 0 && (!X || !Y))) return fail_arg(c, "zigp_kron_elbo: need X, Y for N > 0 rows");
  if (!(jitter >= 0)) return fail_arg(c, "zigp_kron_elbo: jitter must be >= 0");
  ZIGP_HIP(c, hipSetDevice(c->device));
  if (N == 0) return kron_run(c, p, kron_no_rows, kron_no_rows, 1, jitter, 0.0, g_offset, f_mu, include_kl, false, nullptr, elbo_data, kl, grads, ZIGP_LIK_ONOFF, d_f_mu);
  return kron_run(c, p, X, Y, N, jitter, scale, g_offset, f_mu, include_kl, false, nullptr, elbo_data, kl, grads, ZIGP_LIK_ONOFF, d_f_mu);
}

int zigp_kron_elbo_rows(zigp_ctx* c, const zigp_kron_params* p, int64_t row_begin, int64_t row_end, double jitter, double scale, double g_offset,
                        double f_mu, int32_t include_kl, double* elbo_data, double* kl, zigp_kron_grads* grads, double* d_f_mu) {
  if (!c) return ZIGP_EARG;
  ZIGP_TRY(validate_kron(c, p));
  if (!(jitter >= 0)) return fail_arg(c, "zigp_kron_elbo_rows: jitter must be >= 0");
  if (row_begin == row_end && row_begin >= 0 && row_begin <= c->N) {     // an empty range (an empty shard of a data-parallel run): zero data term, same calls as its peers
    ZIGP_HIP(c, hipSetDevice(c->device));
    return kron_run(c, p, kron_no_rows, kron_no_rows, 1, jitter, 0.0, g_offset, f_mu, include_kl, false, nullptr, elbo_data, kl, grads, ZIGP_LIK_ONOFF, d_f_mu);
  }
  if (!c->dX) return fail_arg(c, "zigp_kron_elbo_rows: no data set (call zigp_set_data first)");
  if (p->D0 + p->D1 != c->D) return fail_arg(c, "zigp_kron_elbo_rows: D0 + D1 differs from the data's D");
  if (row_begin < 0 || row_end > c->N || row_begin > row_end) return fail_arg(c, "zigp_kron_elbo_rows: bad row range");
  ZIGP_HIP(c, hipSetDevice(c->device));
  return kron_run(c, p, c->dX + row_begin * c->D, c->dY + row_begin, row_end - row_begin, jitter, scale, g_offset, f_mu, include_kl, false, nullptr,
                  elbo_data, kl, grads, ZIGP_LIK_ONOFF, d_f_mu, true);
}

int zigp_kron_fit_steps(zigp_ctx* c, const zigp_kron_params* p, const zigp_kron_fit_opts* opts, double* free_state, double* adam_m, double* adam_v,
                        int64_t n_free, int64_t t0, int32_t n_steps, const int64_t* row_begin, int64_t batch, const double* Xw, const double* Yw,
                        double jitter, double scale, int32_t include_kl, double* elbo_data, double* kl) {
  if (!c) return ZIGP_EARG;
  c->fit_steps_applied = 0;      // whatever ends this call early, no update of it has been applied
  if (!p || !opts || !free_state || !adam_m || !adam_v || !row_begin) return fail_arg(c, "zigp_kron_fit_steps: NULL argument");
  if (p->M0f <= 0 || p->M1f <= 0 || p->M0g <= 0 || p->M1g <= 0) return fail_arg(c, "inducing counts must be positive");
  if (p->D0 <= 0 || p->D1 <= 0 || p->D0 > MAXD || p->D1 > MAXD) return fail_arg(c, "factor dimensions must be in [1, 8]");
  if (n_steps <= 0 || batch <= 0 || t0 < 0) return fail_arg(c, "zigp_kron_fit_steps: need n_steps > 0, batch > 0, t0 >= 0");
  if (!(jitter >= 0)) return fail_arg(c, "zigp_kron_fit_steps: jitter must be >= 0");
  if (!(opts->beta1 >= 0 && opts->beta1 < 1 && opts->beta2 >= 0 && opts->beta2 < 1 && opts->eps > 0)) return fail_arg(c, "zigp_kron_fit_steps: bad Adam constants");
  if (!c->dX) return fail_arg(c, "zigp_kron_fit_steps: no data set (call zigp_set_data first)");
  if (p->D0 + p->D1 != c->D) return fail_arg(c, "zigp_kron_fit_steps: D0 + D1 differs from the data's D");
  if (c->kron_panels || !kf_eligible(p, 2))
    return fail_arg(c, "zigp_kron_fit_steps: this inducing grid is beyond the fused Kronecker kernels (<= 32 x <= 32 or <= 16 x <= 112 points): "
                       "step it with zigp_kron_elbo_rows and a host optimiser");
  for (int i = 0; i < n_steps; ++i) {
    if (row_begin[i] >= 0 ? row_begin[i] + batch > c->N : (!Xw || !Yw)) return fail_arg(c, "zigp_kron_fit_steps: a row range leaves the resident data set (or a host batch without Xw / Yw)");
  }
  ZIGP_HIP(c, hipSetDevice(c->device));
  KfFitCall fit = {opts, free_state, adam_m, adam_v, n_free, t0, (int)n_steps, row_begin, batch, Xw, Yw, elbo_data, kl};
  return kronf_run(c, p, c->dX, c->dY, batch, jitter, scale, 0.0, 0.0, include_kl, false, nullptr, nullptr, nullptr, nullptr, ZIGP_LIK_ONOFF, nullptr, true, &fit);
}

int64_t zigp_kron_fit_steps_applied(zigp_ctx* c) { return c ? c->fit_steps_applied : (int64_t)ZIGP_EARG; }

int zigp_kron_predict(zigp_ctx* c, const zigp_kron_params* p, const double* Xnew, int64_t N, double jitter, double g_offset, double f_mu, double* out9) {
  if (!c) return ZIGP_EARG;
  ZIGP_TRY(validate_kron(c, p));
  if (N < 0 || (N > 0 && (!Xnew || !out9))) return fail_arg(c, "zigp_kron_predict: bad arguments");
  if (N == 0) return ZIGP_OK;
  ZIGP_HIP(c, hipSetDevice(c->device));
  return kron_predict_chunked(c, p, Xnew, N, jitter, g_offset, f_mu, out9, ZIGP_LIK_ONOFF, 9);
}

int zigp_kron_head_elbo(zigp_ctx* c, const zigp_kron_params* p, int32_t lik, const double* X, const double* Y, int64_t N, double jitter,
                        double scale, double f_mu, int32_t include_kl, double* elbo_data, double* kl, zigp_kron_grads* grads, double* d_f_mu) {
  if (!c) return ZIGP_EARG;
  if (lik == ZIGP_LIK_ONOFF) return fail_arg(c, "zigp_kron_head_elbo: use zigp_kron_elbo for the OnOff likelihood");
  ZIGP_TRY(validate_kron(c, p, lik));
  if (N < 0 || (N > 0 && (!X || !Y))) return fail_arg(c, "zigp_kron_head_elbo: need X, Y for N > 0 rows");
  if (!(jitter >= 0)) return fail_arg(c, "zigp_kron_head_elbo: jitter must be >= 0");
  ZIGP_HIP(c, hipSetDevice(c->device));
  if (N == 0) return kron_run(c, p, kron_no_rows, kron_no_rows, 1, jitter, 0.0, 0.0, f_mu, include_kl, false, nullptr, elbo_data, kl, grads, lik, d_f_mu);
  return kron_run(c, p, X, Y, N, jitter, scale, 0.0, f_mu, include_kl, false, nullptr, elbo_data, kl, grads, lik, d_f_mu);
}

int zigp_kron_head_predict(zigp_ctx* c, const zigp_kron_params* p, int32_t lik, const double* Xnew, int64_t N, double jitter, double f_mu,
                           double* out4) {
  if (!c) return ZIGP_EARG;
  if (lik == ZIGP_LIK_ONOFF) return fail_arg(c, "zigp_kron_head_predict: use zigp_kron_predict for the OnOff likelihood");
  ZIGP_TRY(validate_kron(c, p, lik));
  if (N < 0 || (N > 0 && (!Xnew || !out4))) return fail_arg(c, "zigp_kron_head_predict: bad arguments");
  if (N == 0) return ZIGP_OK;
  ZIGP_HIP(c, hipSetDevice(c->device));
  return kron_predict_chunked(c, p, Xnew, N, jitter, 0.0, f_mu, out4, lik, 4);
}


int zigp_set_kron_range_tiles(zigp_ctx* c, int32_t tiles) {
  if (!c) return ZIGP_EARG;
  if (tiles < 1) return fail_arg(c, "zigp_set_kron_range_tiles: need >= 1 tile");
  c->kron_range_tiles = tiles;
  return ZIGP_OK;
}

int zigp_set_kron_panels(zigp_ctx* c, int32_t on) {
  if (!c) return ZIGP_EARG;
  c->kron_panels = on != 0;
  return ZIGP_OK;
}

}  // extern "C"
