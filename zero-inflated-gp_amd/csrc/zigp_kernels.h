// Device kernels of the dense zero-inflated-GP ELBO path (everything that is not the GEMM core).
// Reference call sites each kernel replaces are cited per kernel (file:line in the reference tree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace zigp {

constexpr int MAXD = 8;       // max input dimension handled by the fused kernels

// Hyperparameter block of the fused Kronecker path (zigp_kronf.hip).  Its kernels take the kernel hyperparameters from a small DEVICE
// block instead of by value: a step then needs nothing from the host but pointers and sizes, so a fit loop can update the parameters
// on the device and enqueue the next step without a round trip (zigp_kron_fit_steps).  Layout in doubles: a record of KH_FAC per
// (latent h, factor q) at (2 h + q) KH_FAC --
//   [KH_INV, +MAXD) 1 / ell_d    [KH_ZC, +MAXD) centre of the moment sums (mid-range of Z_p)    [KH_ELL, +MAXD) ell_d    KH_VAR: var_p
// (KH_ZC is written by k_kf_factor, the first kernel of a step) -- then KH_NOISE: the likelihood variance.
// Read through the CONSTANT address space (scalar loads, exactly as kernel arguments are fetched): the block is written by an earlier
// launch or copy, never by a kernel that reads it.
constexpr int KH_INV = 0, KH_ZC = MAXD, KH_ELL = 2 * MAXD, KH_VAR = 3 * MAXD, KH_FAC = 3 * MAXD + 2;
constexpr int KH_NOISE = 4 * KH_FAC, KH_SIZE = KH_NOISE + 4;
#define KF_CONST(p) ((const __attribute__((address_space(4))) double*)(p))
constexpr int PW_THREADS = 256;

struct KernHyp { double inv_ell[MAXD]; double var; int D; };

// ---------------------------------------------------------------------------------------------
// block-wide deterministic sum (fixed order: lanes by xor-shuffle tree, then waves 0..nw-1)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* sh /*[NW]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) r += sh[w];
  return r;
}

// ---------------------------------------------------------------------------------------------
// RBF kernel matrix, general (KernSE.K onofftf/main.py:53-57 / kernse_np.K onofftf/utils.py:48-52).
// out is (r_pad, c_pad) with ld; entries outside (n1,n2) are identity-padded (1 on the diagonal).
// ---------------------------------------------------------------------------------------------
__global__ void k_rbf_matrix(const double* __restrict__ X1, int64_t n1, const double* __restrict__ X2, int64_t n2,
                             KernHyp h, double jitter, double* __restrict__ out, int64_t r_pad, int64_t c_pad, int64_t ld) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= r_pad * c_pad) return;
  int64_t i = idx / c_pad, j = idx - i * c_pad;
  double v;
  if (i < n1 && j < n2) {
    double r2 = 0.0;
    for (int d = 0; d < h.D; ++d) {
      double t = (X1[i * h.D + d] - X2[j * h.D + d]) * h.inv_ell[d];
      r2 = fma(t, t, r2);
    }
    v = h.var * exp(-0.5 * r2) + ((i == j) ? jitter : 0.0);
  } else {
    v = (i == j) ? 1.0 : 0.0;
  }
  out[i * ld + j] = v;
}

// Kuu of the M x M forward (k_rbf_matrix with X1 = X2 = Z, same expression) together with everything the factorisation chain wants next to
// it, in ONE launch: the working copy L = Kuu with its strictly-upper 128-blocks zero (the chain factors in place and never reads them;
// potrf_trtri_jobs otherwise clears them block row by block row afterwards: 7 memsets at M = 1024), W = 0, and s2 = s^2.
__global__ void k_kuu_setup(const double* __restrict__ Z, int64_t M, KernHyp h, double jitter, double* __restrict__ Kuu, double* __restrict__ L,
                            double* __restrict__ W, const double* __restrict__ s, double* __restrict__ s2, int64_t Mp) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mp * Mp) return;
  if (idx < Mp) s2[idx] = s[idx] * s[idx];
  int64_t i = idx / Mp, j = idx - i * Mp;
  double v;
  if (i < M && j < M) {
    double r2 = 0.0;
    for (int d = 0; d < h.D; ++d) {
      double t = (Z[i * h.D + d] - Z[j * h.D + d]) * h.inv_ell[d];
      r2 = fma(t, t, r2);
    }
    v = h.var * exp(-0.5 * r2) + ((i == j) ? jitter : 0.0);
  } else {
    v = (i == j) ? 1.0 : 0.0;
  }
  Kuu[idx] = v;
  L[idx] = (j / 128 > i / 128) ? 0.0 : v;
  W[idx] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// Kuf panel for one chunk: K[m][n] = var*exp(-0.5*|(z_m - x_n)/ell|^2), m < M ; 0 for padded rows.
// (kern.K(X, Xnew), onofftf/main.py:266.)  grid (Nc/512, Mp/16); thread = two adjacent columns, 16 rows.
// This kernel runs beside the MFMA-bound products and every fp64 VALU instruction of it is paid in MFMA time (the matrix pipe and the
// vector ALU share issue slots: a knock-out with a 2-instruction stand-in for exp() takes 1.35 ms off the cfg3 step,
// profiles/r05o_kuf_exp.log) -- so the exponential is written out with as few instructions as full precision allows instead of calling
// the library's exp() (~25 instructions with its special cases; cfg3 -0.15 % same-box, r05s_ab_milestones.log):
//   inputs pre-scaled by c / ell_d, c = sqrt(16 / ln 2) (Zs on the host, x here), so that  w = -sum_d (zs_d - xs_d)^2 = -y 32 / ln 2,
//   y = 0.5 r^2;  exp(-y) = 2^(w / 32) = 2^E * T[j] * 2^(g / 32)  with  n = rint(w) = 32 E + j (j = n & 31, E = n >> 5: floor semantics,
//   w <= 0), g = w - n in [-1/2, 1/2] (exact), T[j] = var 2^(j / 32) from a 32-entry LDS table (a wave's reads hit at most two entries per
//   bank), 2^(g / 32) = exp(g ln2 / 32) by its degree-6 Taylor polynomial (next term < 3.5e-18), v_ldexp_f64 for 2^E (gradual underflow,
//   0 below 2^-1075; a saturated conversion of a huge |w| gives E = -2^26: 0 as well).  16 + 2 D fp64 / integer VALU instructions per
//   element (22 at D = 3, was 35); the result is within 2 ulp of the correctly rounded value of exp at the computed argument (table
//   entry, product, final fma): tests/test_gpu_blocks.py::test_kuf_panel_kernel_golden_kernse_np_and_ulp.
// ---------------------------------------------------------------------------------------------
constexpr double KUF_C = 0x1.337cc2183b050p+2;   // sqrt(16 / ln 2)
struct KufHyp { double scale[MAXD]; double var; };   // scale[d] = KUF_C * (1 / ell_d): the same doubles the host multiplied into Zs
__device__ const double KUF_T[32] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0, 0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0,
    0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0, 0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0, 0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0,
    0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0, 0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0, 0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0,
    0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0};
__device__ __forceinline__ double kuf_exp2_32(double w, const double* T) {   // T[j] * 2^(w / 32), w <= 0
  const double kn = __builtin_rint(w);
  const double g = w - kn;
  double p = fma(g, 0x1.430912f86c787p-43, 0x1.5d87fe78a6731p-35);
  p = fma(g, p, 0x1.3b2ab6fba4e77p-27);
  p = fma(g, p, 0x1.c6b08d704a0c0p-20);
  p = fma(g, p, 0x1.ebfbdff82c58fp-13);
  p = fma(g, p, 0x1.62e42fefa39efp-6);
  p = fma(g, p, 1.0);
  const int n = (int)kn;
  return __builtin_amdgcn_ldexp(T[n & 31] * p, n >> 5);
}
template <int D>
__global__ void __launch_bounds__(256)
k_kuf_build(const double* __restrict__ X, int64_t N, int64_t n0, const double* __restrict__ Zs, int M, KufHyp h,
            double* __restrict__ K, int64_t Nc) {
  __shared__ double T[32];
  if (threadIdx.x < 32) T[threadIdx.x] = h.var * KUF_T[threadIdx.x];
  const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;   // two adjacent columns: 16-byte stores
  const int m0 = blockIdx.y * 16;
  double xs[2][D];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const bool valid = (n0 + n + e) < N;
#pragma unroll
    for (int d = 0; d < D; ++d) xs[e][d] = valid ? X[(n0 + n + e) * D + d] * h.scale[d] : 0.0;
  }
  __syncthreads();
#pragma unroll 4
  for (int mm = 0; mm < 16; ++mm) {
    const int m = m0 + mm;
    double2* out = reinterpret_cast<double2*>(K + (int64_t)m * Nc + n);
    if (m >= M) { *out = make_double2(0.0, 0.0); continue; }     // uniform over the block
    double w0, w1;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double zs = Zs[m * D + d];      // uniform: a scalar load
      const double t0 = zs - xs[0][d], t1 = zs - xs[1][d];
      w0 = d == 0 ? -t0 * t0 : fma(-t0, t0, w0); w1 = d == 0 ? -t1 * t1 : fma(-t1, t1, w1);
    }
    *out = make_double2(kuf_exp2_32(w0, T), kuf_exp2_32(w1, T));
  }
}

// ---------------------------------------------------------------------------------------------
// Point-wise stage: probit moments (OnOffSVGP.ProbitExpectations onoffgpf/OnOffSVGP.py:168-204),
// augmentation (:146-148), expected log-likelihood (OnOffLikelihood.py:30-32) and the hand-derived
// reverse pass to the cotangents of (fmean, fvar, gmean, gvar) and the noise variance.
// ---------------------------------------------------------------------------------------------
struct PwArgs {
  const double* part_f; const double* part_g;    // per latent [3][NP][Nc]: sum v A1, sum A1^2, sum s2 A2^2 partial rows (EpiStoreColsum)
  int np_f, np_g;                                 // allocated partial rows per quantity (plane stride np * Nc)
  int np1_f, np2_f, np1_g, np2_g;                 // rows actually written: np1 by the A1 kernel (quantities 0, 1), np2 by the A2 kernel
  const double* Y; int64_t n0, row_end, Nc;
  double var_f, var_g, noise, g_offset, scale;
  double* gm_f; double* gv_f; double* gm_g; double* gv_g;
  double* acc;      // [gridDim.x][PW_ACC] running sums: var_exp, dnoise, sum gv_f, sum gv_g, then (mean function on) sum gm_f, sum gm_f x_d
  // mean function of f, m(x) = mean_b + mean_a . x  (self.mean_function(Xnew), OnOffSVGP.py:134; Zero / Constant / Linear)
  const double* X; int D; int mean_on; double mean_a[MAXD]; double mean_b;
  double* out9;     // predict: (9, ld9)
  int64_t ld9;
};

constexpr int PW_ACC = 5 + MAXD;

struct PwOut { double gfmean, gfvar, gfmeanu, e1, e2, ev; double dfm, dfv, dgm, dgv, ve, dnoise; };

__device__ __forceinline__ PwOut pointwise_eval(double fm, double fv, double gmn, double gvr, double y, double noise) {
  PwOut o;
  const double c1 = 1.0 - 2.e-3, c0 = 1.e-3;
  const double inv_sqrt_1pv = 1.0 / sqrt(1.0 + gvr);
  const double z = gmn * inv_sqrt_1pv;                          // OnOffSVGP.py:190
  const double a = 1.0 / sqrt(1.0 + 2.0 * gvr);                 // :191
  const double cdf = 0.5 * (1.0 + erf(z * 0.70710678118654752440)) * c1 + c0;   // :178
  const double ex = exp(-0.5 * (z * z) * (a * a + 1.0));        // :187
  const double at = atan(a) * 0.15915494309189533577;           // :186  atan(a)/(2 pi)
  const double T = at * ex;
  const double e1 = cdf;                                        // :196
  const double e2r = cdf - 2.0 * T;                             // :197
  const double evr = cdf - 2.0 * T - cdf * cdf;                 // :198
  const double e2 = (e2r + fabs(e2r)) * 0.5;                    // :201
  const double ev = (evr + fabs(evr)) * 0.5;                    // :202
  o.e1 = e1; o.e2 = e2; o.ev = ev;
  o.gfmean = e1 * fm; o.gfvar = e2 * fv; o.gfmeanu = ev * fm * fm;          // :146-148
  const double inv_noise = 1.0 / noise;
  const double res = y - o.gfmean;
  const double q = res * res + o.gfvar + o.gfmeanu;
  o.ve = -0.5 * 1.8378770664093454836 - 0.5 * log(noise) - 0.5 * q * inv_noise;   // OnOffLikelihood.py:31-32
  // ---- reverse pass ----
  const double dFmu = res * inv_noise, dFv = -0.5 * inv_noise;
  o.dnoise = -0.5 * inv_noise + 0.5 * q * inv_noise * inv_noise;
  o.dfm = dFmu * e1 + dFv * ev * 2.0 * fm;
  o.dfv = dFv * e2;
  const double de1 = dFmu * fm, de2 = dFv * fv, dev = dFv * fm * fm;
  const double m2 = (e2r > 0.0) ? 1.0 : ((e2r < 0.0) ? 0.0 : 0.5);
  const double mv = (evr > 0.0) ? 1.0 : ((evr < 0.0) ? 0.0 : 0.5);
  const double dcdf = de1 + de2 * m2 + dev * mv * (1.0 - 2.0 * cdf);
  const double dT = -2.0 * (de2 * m2 + dev * mv);
  const double phi = 0.39894228040143267794 * exp(-0.5 * z * z);
  const double dTdz = -z * (a * a + 1.0) * T;
  const double dTda = 0.15915494309189533577 / (1.0 + a * a) * ex - z * z * a * T;
  const double dz = dcdf * c1 * phi + dT * dTdz;
  const double da = dT * dTda;
  o.dgm = dz * inv_sqrt_1pv;
  o.dgv = dz * (-0.5 * z / (1.0 + gvr)) + da * (-(a * a * a));
  return o;
}

// A block owns PW_PTS points; wave g of its PW_GROUPS waves adds the partial rows q = g, g + PW_GROUPS, ... (so that
// Nc / PW_PTS = 512 blocks x 4 waves keep enough loads in flight to stream the partial-row planes), wave 0 adds the group sums
// in group order (fixed order: bit-stable) and evaluates the point.
constexpr int PW_PTS = 64, PW_GROUPS = PW_THREADS / PW_PTS;
// (a device function: the gradient step runs it as the leading workgroups of the J' launch, two blocks per 512-thread workgroup)
template <bool PREDICT>
__device__ __forceinline__ void pw_block(const PwArgs& p, int blk, int tid, double (*grp)[PW_GROUPS][PW_PTS]) {
  const int lane = tid & (PW_PTS - 1), g = tid / PW_PTS;
  const int64_t n = (int64_t)blk * PW_PTS + lane;
  {
    double fm = 0.0, fsq = 0.0, fs2 = 0.0, gmn = 0.0, gsq = 0.0, gs2 = 0.0;
    for (int q = g; q < p.np1_f; q += PW_GROUPS) {
      fm += p.part_f[(int64_t)(0 * p.np_f + q) * p.Nc + n];
      fsq += p.part_f[(int64_t)(1 * p.np_f + q) * p.Nc + n];
    }
    for (int q = g; q < p.np2_f; q += PW_GROUPS) fs2 += p.part_f[(int64_t)(2 * p.np_f + q) * p.Nc + n];
    for (int q = g; q < p.np1_g; q += PW_GROUPS) {
      gmn += p.part_g[(int64_t)(0 * p.np_g + q) * p.Nc + n];
      gsq += p.part_g[(int64_t)(1 * p.np_g + q) * p.Nc + n];
    }
    for (int q = g; q < p.np2_g; q += PW_GROUPS) gs2 += p.part_g[(int64_t)(2 * p.np_g + q) * p.Nc + n];
    grp[0][g][lane] = fm; grp[1][g][lane] = fsq; grp[2][g][lane] = fs2;
    grp[3][g][lane] = gmn; grp[4][g][lane] = gsq; grp[5][g][lane] = gs2;
  }
  __syncthreads();
  if (g != 0) return;
  double tot[6];
#pragma unroll
  for (int v = 0; v < 6; ++v) {
    double a = grp[v][0][lane];
#pragma unroll
    for (int w = 1; w < PW_GROUPS; ++w) a += grp[v][w][lane];
    tot[v] = a;
  }
  double fm = tot[0], gmn = tot[3];
  const double fv = p.var_f - tot[1] + tot[2], gvr = p.var_g - tot[4] + tot[5];   // main.py:278,302
  gmn += p.g_offset;
  const bool valid = (p.n0 + n) < p.row_end;
  double xs[MAXD];
  if (p.mean_on) {
    double m = p.mean_b;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
      xs[d] = (d < p.D && valid) ? p.X[(p.n0 + n) * p.D + d] : 0.0;
      m = fma(p.mean_a[d], xs[d], m);
    }
    fm += m;
  }
  const double y = (valid && p.Y) ? p.Y[p.n0 + n] : 0.0;
  PwOut o = pointwise_eval(fm, fv, gmn, gvr, y, p.noise);
  if (PREDICT) {
    if (valid) {
      double* q = p.out9 + (p.n0 + n);
      q[0 * p.ld9] = o.gfmean; q[1 * p.ld9] = o.gfvar; q[2 * p.ld9] = o.gfmeanu;
      q[3 * p.ld9] = fm; q[4 * p.ld9] = fv; q[5 * p.ld9] = gmn; q[6 * p.ld9] = gvr;
      q[7 * p.ld9] = o.e1; q[8 * p.ld9] = o.ev;
    }
    return;
  }
  const double sc = valid ? p.scale : 0.0;
  if (p.gm_f) {
    p.gm_f[n] = sc * o.dfm; p.gv_f[n] = sc * o.dfv; p.gm_g[n] = sc * o.dgm; p.gv_g[n] = sc * o.dgv;
  }
  // block sums = sums over this one wave; acc[block] keeps accumulating chunk after chunk
  const double s0 = wave_sum(valid ? p.scale * o.ve : 0.0), s1 = wave_sum(sc * o.dnoise), s2 = wave_sum(sc * o.dfv), s3 = wave_sum(sc * o.dgv);
  double* a = p.acc + (int64_t)blk * PW_ACC;
  if (lane == 0) { a[0] += s0; a[1] += s1; a[2] += s2; a[3] += s3; }
  if (p.mean_on) {   // d/d mean_b = sum gm_f, d/d mean_a[d] = sum gm_f x_d
    const double gmf = sc * o.dfm;
    const double sb = wave_sum(gmf);
    if (lane == 0) a[4] += sb;
    for (int d = 0; d < p.D; ++d) {
      const double sa = wave_sum(gmf * xs[d]);
      if (lane == 0) a[5 + d] += sa;
    }
  }
}
template <bool PREDICT>
__global__ void __launch_bounds__(PW_THREADS)
k_pointwise(PwArgs p) {
  __shared__ double grp[6][PW_GROUPS][PW_PTS];
  pw_block<PREDICT>(p, blockIdx.x, threadIdx.x, grp);
}

// ---------------------------------------------------------------------------------------------
// Kuf -> (Z, ell, var) cotangent reductions.  A block owns KG_ROWS rows of K / J' and sweeps the chunk's columns, so
// gm, gv and the D coordinates of x_n are loaded once per column and reused for every row.  The Kuf cotangent is formed
// on the fly, F[m,n] = dK = alpha[m] gm[n] + 2 gv[n] J'[m,n]:
//   krow[m][0]     += sum_n F K
//   krow[m][1+d]   += sum_n F K (x_nd - z_md)
//   krow[m][1+D+d] += sum_n F K (x_nd - z_md)^2
//   krow[m][1+2D]  += sum_n K[m,n] gm[n]          (K gm: seeds A1 gm = W (K gm) and A2 gm = W^T W (K gm))
// (reverse of KernSE.K, onofftf/main.py:41-57)
// ---------------------------------------------------------------------------------------------
constexpr int KG_ROWS = 4;
constexpr int KG_SPLIT = 4;   // column splits (blockIdx.y); each split accumulates into its own krow slab [KG_SPLIT][Mp][W]
// K[m,n] is READ from the chunk's Kuf panel.  Round 3 recomputed it from x_n and z_m (half the HBM bytes, 28 more fp64 VALU instructions per
// element: 0.3 % faster per step on that round's core); on the 16x16x4 core reading is faster in every round (profiles/r04ai_ab_kgread.log:
// cfg3 -0.3 ... -1.1 ms per step, cfg2 -1.3 %): beside MFMA-bound products that leave 7 of the 8 TB/s of HBM idle, fp64 VALU work costs
// more than bytes.  K stays valid until this kernel is done: the next chunk's panels are built behind it on the same stream
// (dense_chunk_loop).  (The recomputing form, the eight-way column split and a slim 1-row x 2-column version are in
// tools/r4_experiment_arms.patch and HISTORY.md section 5.)
// The moment sums are taken about ONE centre c (the mean of the latent's inducing inputs, from the host) and moved to z_m after the block
// reduction:  sum F K (x - z) = S1 - dz S0,  sum F K (x - z)^2 = S2 - 2 dz S1 + dz^2 S0,  dz = z_m - c  -- two fused multiply-adds per
// element and dimension instead of four (x - c and its square are formed once per column, for all KG_ROWS rows): 12.5 instead of 17 fp64
// VALU instructions per element at D = 3, and this kernel runs beside the MFMA-bound rank-N updates, where every VALU instruction is
// paid in matrix-pipe time (k_kuf_build above; cfg3 -0.3 % same-box, profiles/r05s_ab_milestones.log).  The shift costs (|dz| / ell)^2 ulp of cancellation -- the spread of the inducing
// inputs in lengthscales, not of the data's offset from the origin.
// EXACT (round 6): the differences x - z_m are formed per row and nothing is shifted afterwards -- the 17-instruction form of
// round 4.  The host selects it for a latent whose inducing inputs lie more than KG_EXACT_SPREAD lengthscales from their mean in some
// dimension (a long 1-D / time-series input: (1e3)^2 ulp = 2e-10 is where the centred form starts to show next to the 1e-6 parity bar).
constexpr double KG_EXACT_SPREAD = 1.0e3;
struct KgCentre { double c[MAXD]; };
template <int D, bool EXACT>
__global__ void __launch_bounds__(256)
k_kgrad(const double* __restrict__ Jp, const double* __restrict__ K, const double* __restrict__ alpha,
        const double* __restrict__ gm, const double* __restrict__ gv, const double* __restrict__ X, int64_t N, int64_t n0,
        const double* __restrict__ Z, int M, int64_t Nc, int64_t slab, KgCentre ctr, double* __restrict__ krow) {
  constexpr int W = 2 + 2 * D;
  __shared__ double sh[4][KG_ROWS * W];
  __shared__ double tot[KG_ROWS * W];
  krow += (int64_t)blockIdx.y * slab;
  const int m0 = blockIdx.x * KG_ROWS;
  if (m0 >= M) return;
  double am[KG_ROWS], acc[KG_ROWS][W], zr[KG_ROWS][EXACT ? D : 1];
#pragma unroll
  for (int r = 0; r < KG_ROWS; ++r) {
    am[r] = alpha[min(m0 + r, M - 1)];
    if (EXACT) {
#pragma unroll
      for (int d = 0; d < D; ++d) zr[r][d] = Z[min(m0 + r, M - 1) * D + d];
    }
#pragma unroll
    for (int q = 0; q < W; ++q) acc[r][q] = 0.0;
  }
  const int64_t nspan = Nc / KG_SPLIT, nbeg = (int64_t)blockIdx.y * nspan;
  const int64_t nmax = min(nbeg + nspan, N - n0);
  for (int64_t n = nbeg + threadIdx.x; n < nmax; n += 256) {
    const double gmn = gm[n], gv2 = 2.0 * gv[n];
    double xc[D], xx[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (EXACT) { xc[d] = X[(n0 + n) * D + d]; xx[d] = 0.0; }
      else { xc[d] = X[(n0 + n) * D + d] - ctr.c[d]; xx[d] = xc[d] * xc[d]; }
    }
#pragma unroll
    for (int r = 0; r < KG_ROWS; ++r) {
      const int64_t o = (int64_t)(m0 + r) * Nc + n;     // rows beyond M are zero-padded panels (inside the allocation)
      const double kk = K[o];
      const double t = fma(gv2, Jp[o], am[r] * gmn) * kk;
      acc[r][0] += t;
      acc[r][1 + 2 * D] = fma(kk, gmn, acc[r][1 + 2 * D]);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (EXACT) {
          const double df = xc[d] - zr[r][d], td = t * df;
          acc[r][1 + d] += td;
          acc[r][1 + D + d] = fma(td, df, acc[r][1 + D + d]);
        } else {
          acc[r][1 + d] = fma(t, xc[d], acc[r][1 + d]);
          acc[r][1 + D + d] = fma(t, xx[d], acc[r][1 + D + d]);
        }
      }
    }
  }
  // fixed-order block reduction of the KG_ROWS*W partials: lanes by xor tree, then waves 0..3
#pragma unroll
  for (int r = 0; r < KG_ROWS; ++r)
#pragma unroll
    for (int q = 0; q < W; ++q) {
      const double v = wave_sum(acc[r][q]);
      if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][r * W + q] = v;
    }
  __syncthreads();
  if (threadIdx.x < KG_ROWS * W) tot[threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
  __syncthreads();
  if (threadIdx.x < KG_ROWS * W) {
    const int r = threadIdx.x / W, q = threadIdx.x - r * W;
    if (m0 + r < M) {
      const double* S = tot + r * W;
      double v = S[q];
      if (!EXACT && q >= 1 && q <= 2 * D) {      // moments about c -> about z_m
        const int d = (q - 1) % D;
        const double dz = Z[(m0 + r) * D + d] - ctr.c[d];
        v = (q <= D) ? fma(-dz, S[0], S[1 + d]) : fma(dz, fma(dz, S[0], -2.0 * S[1 + d]), S[1 + D + d]);
      }
      krow[(int64_t)(m0 + r) * W + q] += v;
    }
  }
}

// Kuu -> (Z, ell, var) cotangent reductions with a symmetric G = dELBO/dKuu, block per row i:
//   krow[i][0] += sum_j G Kz ; krow[i][1+d] += sum_j 2 G Kz (z_jd - z_id) ; krow[i][1+D+d] += sum_j G Kz (z_id-z_jd)^2
// where Kz = Kuu - jitter*I.
__global__ void __launch_bounds__(256)
k_kuu_grad(const double* __restrict__ G, const double* __restrict__ Kuu, double jitter, const double* __restrict__ Z,
           int M, int D, int64_t ld, double* __restrict__ krow) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  if (i >= M) return;
  double zz[MAXD];
#pragma unroll
  for (int d = 0; d < MAXD; ++d) zz[d] = (d < D) ? Z[i * D + d] : 0.0;
  double s0 = 0.0, s1[MAXD], s2[MAXD];
#pragma unroll
  for (int d = 0; d < MAXD; ++d) { s1[d] = 0.0; s2[d] = 0.0; }
  for (int j = threadIdx.x; j < M; j += 256) {
    double kz = Kuu[(int64_t)i * ld + j] - ((i == j) ? jitter : 0.0);
    const double t = G[(int64_t)i * ld + j] * kz;
    s0 += t;
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
      if (d < D) {
        const double df = Z[j * D + d] - zz[d];
        const double td = t * df;
        s1[d] = fma(2.0, td, s1[d]);
        s2[d] = fma(td, df, s2[d]);
      }
  }
  const int W = 2 + 2 * D;
  s0 = block_sum<4>(s0, sh);
  if (threadIdx.x == 0) krow[(int64_t)i * W] += s0;
  for (int d = 0; d < D; ++d) {
    double a = block_sum<4>(s1[d], sh);
    double b = block_sum<4>(s2[d], sh);
    if (threadIdx.x == 0) { krow[(int64_t)i * W + 1 + d] += a; krow[(int64_t)i * W + 1 + D + d] += b; }
  }
}

// C1[i][j] = sum_s part[s][max(i,j)][min(i,j)]   (planes hold the lower triangle of A1 G A1^T; fixed summation order: s ascending)
// One workgroup per 32 x 32 tile on or below the diagonal: the planes are read ONCE, along rows (the first version read the upper
// half through transposed addresses, S strided loads per element: 118-150 us at M = 512 with 64 planes, a quarter of the M x M reverse
// pass), the mirror image goes out through LDS.  Tiles inside a diagonal 128-block live in the first Sd planes only (the others hold
// the zeros of the memset, and x + 0 = x: the sums keep their bits).
__global__ void __launch_bounds__(256)
k_sym_from_planes(const double* __restrict__ part, int So, int Sd, int64_t Mp, double* __restrict__ out) {
  __shared__ double sh[32][33];
  const int t = blockIdx.x;
  int tr = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (tr * (tr + 1) / 2 > t) --tr;
  while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
  const int tc = t - tr * (tr + 1) / 2;
  const int S = (tr * 32 / BM == tc * 32 / BM) ? Sd : So;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t plane = Mp * Mp;
  const double* __restrict__ src = part + (int64_t)(32 * tr + ty) * Mp + 32 * tc + tx;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += src[(int64_t)s * plane + (int64_t)(8 * k) * Mp];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sh[ty + 8 * k][tx] = a[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = ty + 8 * k;
    if (tr != tc) {
      out[(int64_t)(32 * tr + r) * Mp + 32 * tc + tx] = sh[r][tx];
      out[(int64_t)(32 * tc + r) * Mp + 32 * tr + tx] = sh[tx][r];
    } else {
      out[(int64_t)(32 * tr + r) * Mp + 32 * tc + tx] = (tx <= r) ? sh[r][tx] : sh[tx][r];
    }
  }
}
// zero up to twelve device ranges in one launch (the per-call accumulators of the dense path: a memset launch each costs ~5 us of a
// launch-bound M x M stage)
constexpr int ZERO_RANGES_MAX = 12;
struct ZeroRanges { double* p[ZERO_RANGES_MAX]; int64_t n[ZERO_RANGES_MAX]; int count; };
__global__ void __launch_bounds__(256)
k_zero_ranges(ZeroRanges z) {
  for (int q = 0; q < z.count; ++q)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < z.n[q]; i += (int64_t)gridDim.x * 256) z.p[q][i] = 0.0;
}
// V = U + U^T - C
__global__ void k_uut_minus(const double* __restrict__ U, const double* __restrict__ C, int64_t Mp, double* __restrict__ V) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mp * Mp) return;
  int64_t i = idx / Mp, j = idx - i * Mp;
  V[idx] = U[idx] + U[j * Mp + i] - C[idx];
}
// dL[i][j] = -(alpha[i] a1gm[j] + a2gm[i] v[j] + 2 R[i][j]) on/below the diagonal, 0 above
// (L-bar of the two triangular solves; the rank-1 terms are the gm-parts of F A1^T + A2 E^T)
__global__ void k_dl_assemble(const double* __restrict__ R, int64_t Mp, const double* __restrict__ alpha,
                              const double* __restrict__ a1gm, const double* __restrict__ a2gm, const double* __restrict__ v,
                              double* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mp * Mp) return;
  int64_t i = idx / Mp, j = idx - i * Mp;
  out[idx] = (j <= i) ? -(2.0 * R[idx] + alpha[i] * a1gm[j] + a2gm[i] * v[j]) : 0.0;
}

// G = 0.5*(S + S^T) - 0.5*(P - alpha alpha^T - PSP)   (KL part only if with_kl)
__global__ void k_sym_combine(const double* __restrict__ S, const double* __restrict__ P, const double* __restrict__ PSP,
                              const double* __restrict__ alpha, int with_data, int with_kl, int64_t Mp, double* __restrict__ G) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mp * Mp) return;
  int64_t i = idx / Mp, j = idx - i * Mp;
  double v = 0.0;
  if (with_data) v = 0.5 * (S[i * Mp + j] + S[j * Mp + i]);
  if (with_kl) v -= 0.5 * (0.5 * (P[i * Mp + j] + P[j * Mp + i]) - alpha[i] * alpha[j] - 0.5 * (PSP[i * Mp + j] + PSP[j * Mp + i]));
  G[idx] = v;
}

// v = W u  (row-contiguous W), block per row
__global__ void __launch_bounds__(256)
k_gemv_rows(const double* __restrict__ W, const double* __restrict__ u, int64_t Mp, double* __restrict__ v) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  double a = 0.0;
  for (int k = threadIdx.x; k <= i; k += 256) a = fma(W[(int64_t)i * Mp + k], u[k], a);
  a = block_sum<4>(a, sh);
  if (threadIdx.x == 0) v[i] = a;
}
// strided gather over slabs: x[i] = sum_sp src[sp*slab + i*stride + off]   (fixed order)
__global__ void k_gather(const double* __restrict__ src, int stride, int off, int n, int nslab, int64_t slab, double* __restrict__ x) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = 0.0;
  for (int sp = 0; sp < nslab; ++sp) a += src[(int64_t)sp * slab + (int64_t)i * stride + off];
  x[i] = a;
}
// Column reductions over the lower triangle of W (row-major, Mp x Mp).  A block owns 64 columns; its 16 row lanes (threadIdx.y)
// stride over the rows k >= first column of the strip, then the 16 partial sums are added in lane order (fixed order:
// bit-stable).  Launch with dim3(Mp / 64), dim3(64, COL_LANES).
constexpr int COL_LANES = 16;
template <int NV>
__device__ __forceinline__ void col_lanes_reduce(double (&a)[NV], double (*sh)[COL_LANES][64]) {
#pragma unroll
  for (int q = 0; q < NV; ++q) sh[q][threadIdx.y][threadIdx.x] = a[q];
  __syncthreads();
  if (threadIdx.y == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      double t = 0.0;
      for (int l = 0; l < COL_LANES; ++l) t += sh[q][l][threadIdx.x];
      a[q] = t;
    }
  }
}
// y[i] = sum_{k>=i} W[k][i] x[k]   (W^T x, W lower triangular)
__global__ void __launch_bounds__(64 * COL_LANES)
k_gemv_cols(const double* __restrict__ W, const double* __restrict__ x, int64_t Mp, double* __restrict__ y) {
  __shared__ double sh[1][COL_LANES][64];
  const int i = blockIdx.x * 64 + threadIdx.x;
  double a[1] = {0.0};
  for (int k = blockIdx.x * 64 + threadIdx.y; k < Mp; k += COL_LANES)
    if (k >= i) a[0] = fma(W[(int64_t)k * Mp + i], x[k], a[0]);
  col_lanes_reduce<1>(a, sh);
  if (threadIdx.y == 0) y[i] = a[0];
}
// d[i] = sum_{k>=i} W[k][i] Y[k][i]   (diag(W^T Y))
__global__ void __launch_bounds__(64 * COL_LANES)
k_coldot(const double* __restrict__ W, const double* __restrict__ Y, int64_t Mp, double* __restrict__ d) {
  __shared__ double sh[1][COL_LANES][64];
  const int i = blockIdx.x * 64 + threadIdx.x;
  double a[1] = {0.0};
  for (int k = blockIdx.x * 64 + threadIdx.y; k < Mp; k += COL_LANES)
    if (k >= i) a[0] = fma(W[(int64_t)k * Mp + i], Y[(int64_t)k * Mp + i], a[0]);
  col_lanes_reduce<1>(a, sh);
  if (threadIdx.y == 0) d[i] = a[0];
}
// alpha[i] = sum_k W[k][i] v[k] ; dkinv[i] = sum_k W[k][i]^2
__global__ void __launch_bounds__(64 * COL_LANES)
k_kl_cols(const double* __restrict__ W, const double* __restrict__ v, int64_t Mp,
          double* __restrict__ alpha, double* __restrict__ dkinv) {
  __shared__ double sh[2][COL_LANES][64];
  const int i = blockIdx.x * 64 + threadIdx.x;
  double a[2] = {0.0, 0.0};
  for (int k = blockIdx.x * 64 + threadIdx.y; k < Mp; k += COL_LANES)
    if (k >= i) {
      const double w = W[(int64_t)k * Mp + i];
      a[0] = fma(w, v[k], a[0]);
      a[1] = fma(w, w, a[1]);
    }
  col_lanes_reduce<2>(a, sh);
  if (threadIdx.y == 0) { alpha[i] = a[0]; dkinv[i] = a[1]; }
}
// KL value (gauss_kl_diag, onofftf/main.py:218-250): one block.
__global__ void __launch_bounds__(256)
k_kl_value(const double* __restrict__ v, const double* __restrict__ L, const double* __restrict__ s, const double* __restrict__ dkinv,
           int M, int64_t Mp, double* __restrict__ out) {
  __shared__ double sh[4];
  double mah = 0.0, lq = 0.0, tr = 0.0, lp = 0.0;
  for (int i = threadIdx.x; i < M; i += 256) {
    mah = fma(v[i], v[i], mah);
    const double si = s[i], li = L[(int64_t)i * Mp + i];
    lq += log(si * si);
    tr = fma(dkinv[i], si * si, tr);
    lp += log(li * li);
  }
  mah = block_sum<4>(mah, sh); lq = block_sum<4>(lq, sh); tr = block_sum<4>(tr, sh); lp = block_sum<4>(lp, sh);
  if (threadIdx.x == 0) out[0] = 0.5 * (mah - (double)M - lq + tr + lp);
}

// ---------------------------------------------------------------------------------------------
// Result vector of one dense ELBO call, assembled on the device so that a data-parallel run can sum it over ranks where it
// lies (ncclAllReduce on the library's stream, zigp_comm_init) and one download brings everything back:
//   out[0] elbo_data  [1] kl  [2] d var_f  [3] d var_g  [4] d noise  [5] d mean_b  [6..13] d mean_a   (header: DP_HDR doubles)
//   then per latent h = f, g (only when a gradient was asked for):  dZ (M_h x D)  du (M_h)  ds (M_h)  dell (D)
// Block h assembles latent h; block 0 also writes the header scalars.  Every sum has a fixed order (strided partial per thread,
// xor tree over lanes, waves in index order): bit-stable run to run.
// ---------------------------------------------------------------------------------------------
constexpr int DP_HDR = 16;
struct DensePackLat {
  const double* krow; const double* du; const double* dsq; const double* vec; const double* s; const double* ell;
  int M, Mp; double var; int64_t out_off;
};
struct DensePackArgs {
  DensePackLat lat[2];
  const double* pw; int pw_blocks; int D; int need_grad, include_kl, mean_on;
  double* out;
};
__global__ void __launch_bounds__(256)
k_dense_pack(DensePackArgs a) {
  __shared__ double sh[4];
  const int h = blockIdx.x, t = threadIdx.x;
  const DensePackLat& L = a.lat[h];
  const int D = a.D, W = 2 + 2 * D;
  // sums over the point-wise blocks: var_exp, d noise, sum gv_f, sum gv_g, (mean function) sum gm_f, sum gm_f x_d
  double pws[PW_ACC];
#pragma unroll
  for (int q = 0; q < PW_ACC; ++q) {
    double v = 0.0;
    if (q < 4 || (a.mean_on && q < 5 + D))
      for (int b = t; b < a.pw_blocks; b += 256) v += a.pw[(int64_t)PW_ACC * b + q];
    pws[q] = block_sum<4>(v, sh);
  }
  if (h == 0 && t == 0) {
    a.out[0] = pws[0];
    a.out[1] = a.include_kl ? a.lat[0].vec[3 * a.lat[0].Mp] + a.lat[1].vec[3 * a.lat[1].Mp] : 0.0;
    a.out[4] = pws[1];
    a.out[5] = a.mean_on ? pws[4] : 0.0;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) a.out[6 + d] = (a.mean_on && d < D) ? pws[5 + d] : 0.0;
    a.out[14] = 0.0; a.out[15] = 0.0;
    if (!a.need_grad) { a.out[2] = 0.0; a.out[3] = 0.0; }
  }
  if (!a.need_grad) return;
  double* o = a.out + L.out_off;
  const int M = L.M, Mp = L.Mp;
  const int64_t slab = (int64_t)Mp * W;
  auto ksum = [&](int m, int q) {   // column splits of the Kuf-cotangent reductions, in split order
    double r = 0.0;
#pragma unroll
    for (int sp = 0; sp < KG_SPLIT; ++sp) r += L.krow[sp * slab + (int64_t)m * W + q];
    return r;
  };
  for (int idx = t; idx < M * D; idx += 256) {
    const int m = idx / D, d = idx - m * D;
    const double e = L.ell[d];
    o[idx] = ksum(m, 1 + d) / (e * e);
  }
  double* ou = o + (int64_t)M * D; double* os = ou + M; double* ol = os + M;
  for (int m = t; m < M; m += 256) {
    const double sm = L.s[m];
    double dum = L.du[m], dsm = 2.0 * sm * L.dsq[m];
    if (a.include_kl) {   // - dKL/du = -alpha ; - dKL/ds = -(-1/s + diag(K^-1) s)
      dum -= L.vec[Mp + m];
      dsm -= (-1.0 / sm + L.vec[2 * Mp + m] * sm);
    }
    ou[m] = dum; os[m] = dsm;
  }
  double dv = 0.0;
  for (int m = t; m < M; m += 256) dv += ksum(m, 0);
  dv = block_sum<4>(dv, sh);
  for (int d = 0; d < D; ++d) {
    double v = 0.0;
    for (int m = t; m < M; m += 256) v += ksum(m, 1 + D + d);
    v = block_sum<4>(v, sh);
    if (t == 0) { const double e = L.ell[d]; ol[d] = v / (e * e * e); }
  }
  if (t == 0) a.out[2 + h] = dv / L.var + pws[2 + h];
}

// rows `idx` of the resident (X, Y) gathered into a contiguous batch (MinibatchData's per-step sample, onoffgpf/OnOffSVGP.py:46-47)
__global__ void k_gather_rows(const double* __restrict__ X, const double* __restrict__ Y, const int64_t* __restrict__ idx, int64_t n, int D,
                              double* __restrict__ Xb, double* __restrict__ Yb) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (D + 1)) return;
  const int64_t i = t / (D + 1); const int d = (int)(t - i * (D + 1));
  const int64_t r = idx[i];
  if (d < D) Xb[i * D + d] = X[r * D + d]; else Yb[i] = Y[r];
}

// s2 = s*s
__global__ void k_square(const double* s, double* s2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) s2[i] = s[i] * s[i];
}
// out[i][j] = W[i][j] * s2[j]
__global__ void k_colscale(const double* __restrict__ W, const double* __restrict__ s2, int64_t Mp, double* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mp * Mp) return;
  out[idx] = s2[idx % Mp] * W[idx];
}
// Wt = W^T and (optionally) Wpt = (W diag(s2))^T = diag(s2) W^T, through a 32 x 33 LDS tile (coalesced reads and writes).
// The lower-triangular products A1 = W K and H = W diag(s^2) A2 then read their triangular factor m-contiguous, like the W^T products:
// all four run the 8-wave kernel shape (zigp_host.h, WavesFor).  Launch with dim3(Mp / 32, Mp / 32), dim3(32, 8).
__global__ void __launch_bounds__(256)
k_transpose_scale(const double* __restrict__ W, const double* __restrict__ s2, int64_t Mp, double* __restrict__ Wt, double* __restrict__ Wpt) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) tile[j][threadIdx.x] = W[(int64_t)(by + j) * Mp + bx + threadIdx.x];   // W[i = by + j][k = bx + x]
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const double v = tile[threadIdx.x][j];                       // W[i = by + x][k = bx + j]
    const int64_t o = (int64_t)(bx + j) * Mp + by + threadIdx.x;   // Wt[k][i]
    Wt[o] = v;
    if (Wpt) Wpt[o] = v * s2[bx + j];
  }
}
// Qt[k][i] = s2[k] * P[k][i] - (k == i):  Q^T for Q = P diag(s^2) - I (P symmetric)
__global__ void k_rowscale_minus_eye(const double* __restrict__ P, const double* __restrict__ s2, int64_t Mp, double* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mp * Mp) return;
  const int64_t k = idx / Mp;
  out[idx] = fma(s2[k], P[idx], (idx - k * Mp == k) ? -1.0 : 0.0);
}
// Bs[k][j] = s2[k] * P[k][j]
__global__ void k_rowscale(const double* __restrict__ P, const double* __restrict__ s2, int64_t Mp, double* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mp * Mp) return;
  out[idx] = s2[idx / Mp] * P[idx];
}

// ---------------------------------------------------------------------------------------------
// Cholesky of one 128x128 diagonal block + its triangular inverse, one workgroup, all in LDS.
// (tf.cholesky, onofftf/main.py:200,268; the explicit inverse replaces the two
// tf.matrix_triangular_solve calls :271,284 by GEMMs -- "W-form")
// A/L/W point at the (j0,j0) block; ld = leading dimension.  info: first failing 1-based pivot index.
// ---------------------------------------------------------------------------------------------
constexpr int PB = 128, PBLD = 129, PNB = 32;   // block size, LDS row stride, panel width
constexpr int PHB = 16;                          // half a panel: the register-level pieces of the second version work on 16 x 16 blocks
// broadcast of one lane's double through the scalar unit (lane index is a compile-time constant after unrolling)
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// Blocked in 32-column panels so that the serial part runs in the registers of ONE wave:
//   (1) wave 0 holds the 32x32 diagonal block row-per-lane, factors it (rank-1 updates fed by v_readlane broadcasts) and
//       inverts it in place (dtrti2 order); L11 goes to the lower, inv(L11) transposed to the upper triangle of its LDS block
//   (2) all threads: L21 = A21 inv(L11)^T, (3) all threads: A22 -= L21 L21^T
// then the off-diagonal 32x32 blocks of W = L^-1 by block forward substitution, all blocks of one block-diagonal at a time:
//   T = sum_k L_ik W_kj,  W_ij = -W_ii T.
// LDS image: S[i][j], j <= i: L;  W[i][c], i > c, lives at S[c][i];  diag(W) in dinv.  ~25 workgroup barriers in total.
// Body shared by k_potrf_diag and the fused Kronecker factor kernel (zigp_kronf.hip): on entry S[i][j], j <= i, holds the lower
// triangle of the SPD block (identity beyond the real rows); on exit L in the lower triangle, W = L^-1 as described above
// (only if want_W).  Returns false when a pivot was not above `tol` (info set, S unfinished).  All 1024 threads must call it.
// tol: the callers pass 8 eps (variance + jitter) -- the diagonal of an RBF Kuu is constant -- so that an exactly singular matrix
// (duplicate inducing points, jitter 0: tf.cholesky raises, onofftf/main.py:355) is reported whichever way its +-1e-16 pivot rounds;
// tol = 0 is the bare LAPACK / Eigen test.
struct PotrfShared { double dinv[PB]; double T[3][PNB][PNB + 1]; int fail; };
__device__ __forceinline__ double potrf_wget(const double* S, const double* dinv, int x, int y) {
  return x > y ? S[y * PBLD + x] : (x == y ? dinv[x] : 0.0);
}
// ---- second version: the serial work is the 32 x 32 factorisation only; everything else runs beside it or on the MFMA pipe.
//   per panel j:  [A] wave 0 factors the diagonal block (registers, v_readlane broadcasts, v_rsq_f64 + two Newton steps per pivot)
//                     wave 1 inverts the PREVIOUS diagonal block (needed for W only, so it is off the critical path): two 16 x 16
//                     halves in registers, the coupling block as two chained MFMA products
//                     waves 2..15 apply the PREVIOUS panel to the columns beyond panel j (16 x 16 MFMA tiles, operands from LDS)
//                 [B] rows below: L21 = A21 L11^-T by forward substitution, four threads per row (no inverse on the critical path)
//                 [C] panel j -> the next panel's 32 columns only (MFMA tiles), so that [A] can start on them
//   then W = L^-1 block diagonal by block diagonal, both products (T = sum L_ik W_kj, W_ij = -W_ii T) as MFMA tiles.
// In-kernel stamps of the first version at M = 100: 296 k cycles, 4 x 30 k of them in the serial factor + inverse of wave 0 and most of
// the rest in LDS-bandwidth-bound scalar dot products of the trailing updates and of W.
// one 16 x 16 x 4 product: c[r] of lane l is C[4 r + l / 16][l % 16], a = A[l % 16][l / 16], b = B[l / 16][l % 16]  (r4: was four 4x4x4 MFMAs
// with the A block read at the same address in all four 16-lane groups; one instruction and one LDS read instead of four each)
typedef double potrf_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void potrf_mfma16(double (&c)[4], double a, double b) {
  potrf_d4 v = {c[0], c[1], c[2], c[3]};
  v = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, v, 0, 0, 0);
  c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
}
__device__ __forceinline__ double potrf_rsqrt(double d) {      // d > 0: hardware estimate + two Newton steps
  double y = __builtin_amdgcn_rsq(d);
  double e = fma(-d * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-d * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  return y;
}
// S[i0 + ., k0 + .] (16 x 16) -= sum_{c = c0 .. c0 + 31} S[i][c] S[k][c]     (one wave)
__device__ __forceinline__ void potrf_tile_update(double* S, int i0, int k0, int c0) {
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < PNB / 4; ++ks) {
    const int c = c0 + 4 * ks + g;
    potrf_mfma16(acc, S[(i0 + n) * PBLD + c], S[(k0 + n) * PBLD + c]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) S[(i0 + 4 * r + g) * PBLD + k0 + n] -= acc[r];
}
// lower tiles (ti >= tk) of the tile range [t0, nt): the idx-th in row-major order
__device__ __forceinline__ void potrf_tri_decode(int idx, int& ti, int& tk) {
  int r = 0;
  while (idx >= r + 1) { idx -= r + 1; ++r; }
  ti = r; tk = idx;
}
// one wave: Cholesky of the 32 x 32 diagonal block at jb -> T[0] (L11, rows), T[1][0][.] and dinv = 1 / diag.  Returns the failing
// 1-based column or 0.  Row per lane, rank-1 updates fed by v_readlane broadcasts -- in two 16-column halves: the 256 updates of the
// lower-right 16 x 16 block by the first 16 columns go through LDS and the MFMA pipe instead (in-kernel stamps: 17 k cycles per
// block with all 496 readlane-fed updates).
template <int J0, int J1>
__device__ __forceinline__ void potrf_factor_cols(double (&a)[PNB], int r, double tol, int& bad, double& rdv) {
#pragma unroll
  for (int j = J0; j < J1; ++j) {
    int rr = r;
    asm volatile("" : "+v"(rr));   // lane masks of (rr == j) are recomputed per column instead of living in 2 SGPRs each
    const double d = readlane_f64(a[j], j);
    if (!(d > tol)) { if (!bad) bad = j + 1; }   // pivot <= tol or NaN (uniform); keep going on garbage, report below
    const double rd = potrf_rsqrt(d);
    double sq = d * rd;
    sq = fma(0.5 * rd, fma(-sq, sq, d), sq);
    const double l = (rr == j) ? sq : a[j] * rd;
    rdv = (rr == j) ? rd : rdv;
    asm volatile("" : "+v"(rdv));      // select now (otherwise every column's rd stays live to the end)
    a[j] = l;
#pragma unroll
    for (int k = j + 1; k < J1; ++k) {   // row r, column k (k > r: unused)
      a[k] = fma(-l, readlane_f64(l, k), a[k]);
      asm volatile("" : "+v"(a[k]));     // materialise now: otherwise the update is sunk to column k and every broadcast stays live
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the broadcasts (SGPR pairs) of one column from piling up across columns
  }
}
__device__ __forceinline__ void potrf_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int potrf_factor32(const double* S, PotrfShared& psh, int jb, double tol) {
  const int lane = threadIdx.x & 63, r = lane & 31;     // lanes 32..63 shadow lanes 0..31 (they never store)
  const int g = lane >> 4, n = lane & 15;
  double a[PNB];
#pragma unroll
  for (int k = 0; k < PNB; ++k) a[k] = S[(jb + r) * PBLD + jb + k];
  int bad = 0;
  double rdv = 1.0;
  potrf_factor_cols<0, PHB>(a, r, tol, bad, rdv);
  // rows 16..31, columns 16..31 -= L21 L21^T: L21 (the first 16 columns of those rows) through T[0], the product through T[1][16..31]
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k < PHB; ++k) psh.T[0][r][k] = a[k];
  }
  potrf_wave_sync();
  {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < PHB / 4; ++ks) {
      const double bv = psh.T[0][PHB + n][4 * ks + g];      // L21 L21^T: the A and the B operand of a lane are the same element
      potrf_mfma16(acc, bv, bv);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) psh.T[1][PHB + 4 * q + g][n] = acc[q];
  }
  potrf_wave_sync();
#pragma unroll
  for (int c = 0; c < PHB; ++c) a[PHB + c] -= psh.T[1][PHB + (r & 15)][c];      // lanes of rows 0..15: leftovers, never used
  potrf_factor_cols<PHB, PNB>(a, r, tol, bad, rdv);
  if (!bad && lane < 32) {
#pragma unroll
    for (int k = PHB; k < PNB; ++k) psh.T[0][r][k] = a[k];
    psh.T[1][0][r] = rdv;
    psh.dinv[jb + r] = rdv;
  }
  return bad;
}
// Inverse of the lower-triangular 32 x 32 diagonal block at pb (already in S, diagonal reciprocals in dinv), transposed into the
// upper triangle of the same block: the two 16 x 16 diagonal halves by one wave each (potrf_invert16), then the coupling block
// W21 = -W22 L21 W11 as two chained MFMA products (potrf_invert_couple).
// potrf_invert16, columns right to left:  w_rj = -w_jj sum_{k=j+1..r} w_rk l_kj   (w_rk lane-local, l_kj: lane k)
__device__ __forceinline__ void potrf_invert16(double* S, PotrfShared& psh, int pb, int h) {
  const int lane = threadIdx.x & 63, r = lane & 15, p0 = pb + PHB * h;     // lanes 16..63 shadow lanes 0..15
  double a[PHB];
#pragma unroll
  for (int k = 0; k < PHB; ++k) a[k] = S[(p0 + r) * PBLD + p0 + k];
  {
    int rr = r;
    asm volatile("" : "+v"(rr));
#pragma unroll
    for (int k = 1; k < PHB; ++k) a[k] = (k <= rr) ? a[k] : 0.0;
  }
  const double dv = psh.dinv[p0 + r];
#pragma unroll
  for (int j = PHB - 1; j >= 0; --j) {
    int rr = r;
    asm volatile("" : "+v"(rr));
    const double wjj = readlane_f64(dv, j);
    double sum = 0.0;
#pragma unroll
    for (int k = j + 1; k < PHB; ++k) sum = fma(a[k], readlane_f64(a[j], k), sum);   // a[k] = 0 for k > r
    a[j] = (rr == j) ? wjj : ((rr > j) ? -wjj * sum : 0.0);
    asm volatile("" : "+v"(a[j]));
    __builtin_amdgcn_sched_barrier(0);
  }
  // rows through the staging tile T[2] with unconditional stores (predicated stores would keep one lane mask per column in SGPRs)
  if (lane < PHB) {
#pragma unroll
    for (int k = 0; k < PHB; ++k) psh.T[2][PHB * h + r][k] = a[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int idx = lane; idx < PHB * PHB; idx += 64) {
    const int rr = idx >> 4, k = idx & 15;
    if (k < rr) S[(p0 + k) * PBLD + p0 + rr] = psh.T[2][PHB * h + rr][k];
  }
}
// one wave, after both halves: X = L21 W11 stays in the accumulators (the D layout of one product is the B layout of the next),
// W21 = -W22 X goes to S[pb + col][pb + 16 + row]
__device__ __forceinline__ void potrf_invert_couple(double* S, PotrfShared& psh, int pb) {
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
  double x[4] = {0.0, 0.0, 0.0, 0.0}, w[4] = {0.0, 0.0, 0.0, 0.0};
  const double dn = psh.dinv[pb + n];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int k = 4 * ks + g;                                  // B = W11[k][n]: S[pb + n][pb + k] for k > n
    const double sv = S[(pb + n) * PBLD + pb + k];
    const double bv = k > n ? sv : (k == n ? dn : 0.0);
    potrf_mfma16(x, S[(pb + PHB + n) * PBLD + pb + k], bv);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int k = 4 * ks + g;                                  // A = W22[row][k]: S[pb + 16 + k][pb + 16 + row] for row > k
    const double sv = S[(pb + PHB + k) * PBLD + pb + PHB + n], dv = psh.dinv[pb + PHB + n];
    potrf_mfma16(w, n > k ? sv : (n == k ? dv : 0.0), x[ks]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) S[(pb + n) * PBLD + pb + PHB + 4 * r + g] = -w[r];
}
// forward substitution of the rows below a factored diagonal block: L21 = A21 L11^-T.  Four threads per row, thread q owns the columns
// k = q (mod 4); the owner of column c scales it and hands x to its quad through a DPP broadcast, everyone updates the columns it owns
template <int SEL> __device__ __forceinline__ double potrf_quad_bcast(double v) {
  constexpr int ctrl = SEL * 0x55;     // quad_perm [SEL, SEL, SEL, SEL]
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void potrf_solve_rows(double* S, PotrfShared& psh, int jb, int nbelow) {
  const int t = threadIdx.x;
  if (t >= 4 * nbelow) return;
  const int i = jb + PNB + (t >> 2), q = t & 3;
  double a[PNB / 4];
#pragma unroll
  for (int s = 0; s < PNB / 4; ++s) a[s] = S[i * PBLD + jb + 4 * s + q];
#pragma unroll
  for (int c = 0; c < PNB; ++c) {
    const int sc = c >> 2, qc = c & 3;
    double x = a[sc] * psh.T[1][0][c];
    switch (qc) { case 0: x = potrf_quad_bcast<0>(x); break; case 1: x = potrf_quad_bcast<1>(x); break;
                  case 2: x = potrf_quad_bcast<2>(x); break; default: x = potrf_quad_bcast<3>(x); break; }
    const double upd = fma(-x, psh.T[0][4 * sc + q][c], a[sc]);
    a[sc] = q > qc ? upd : (q == qc ? x : a[sc]);
#pragma unroll
    for (int s = sc + 1; s < PNB / 4; ++s) a[s] = fma(-x, psh.T[0][4 * s + q][c], a[s]);
  }
#pragma unroll
  for (int s = 0; s < PNB / 4; ++s) S[i * PBLD + jb + 4 * s + q] = a[s];
}
__device__ __forceinline__ bool potrf_diag_lds(double* S, PotrfShared& psh, int j0, int* info, int npan, bool want_W, double tol) {
  double* dinv = psh.dinv;
  double (*T)[PNB][PNB + 1] = psh.T;
  const int t = threadIdx.x, wave = t >> 6;
  if (t == 0) psh.fail = 0;
  if (t < PB) dinv[t] = 1.0;
  __syncthreads();
  const int nreal = npan * PNB, nt = nreal / 16;
  for (int jb = 0; jb < nreal; jb += PNB) {
    // [A]
    if (wave == 0) {
      const int bad = potrf_factor32(S, psh, jb, tol);
      if (bad && t == 0) { atomicCAS(info, 0, j0 + jb + bad); psh.fail = 1; }
    } else if (jb > 0) {
      if (wave == 1) {
        if (want_W) {
          potrf_invert16(S, psh, jb - PNB, 0);
          potrf_invert16(S, psh, jb - PNB, 1);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          potrf_invert_couple(S, psh, jb - PNB);
        }
      } else {
        const int t0 = (jb + PNB) / 16, n = nt - t0;
        for (int idx = wave - 2; idx < n * (n + 1) / 2; idx += 14) {
          int ti, tk;
          potrf_tri_decode(idx, ti, tk);
          potrf_tile_update(S, 16 * (t0 + ti), 16 * (t0 + tk), jb - PNB);
        }
      }
    }
    __syncthreads();
    if (psh.fail) return false;
    // [B] scatter L11 into the lower triangle; the rows below solve against it
    {
      const int r = t >> 5, k = t & 31;
      if (k <= r) S[(jb + r) * PBLD + jb + k] = T[0][r][k];
    }
    const int nbelow = nreal - jb - PNB;
    potrf_solve_rows(S, psh, jb, nbelow);
    __syncthreads();
    // [C] panel jb -> the next panel's columns
    if (nbelow > 0) {
      const int t1 = (jb + PNB) / 16, nrow = nt - t1;     // tiles (t1 + ti, t1 + tk), tk in {0, 1}, ti >= tk
      for (int idx = wave; idx < 2 * nrow - 1; idx += 16) {
        const int tk = idx >= nrow ? 1 : 0, ti = tk ? idx - nrow + 1 : idx;
        potrf_tile_update(S, 16 * (t1 + ti), 16 * (t1 + tk), jb);
      }
    }
    __syncthreads();
  }
  if (!want_W) return true;
  if (wave < 2) potrf_invert16(S, psh, nreal - PNB, wave);
  __syncthreads();
  if (wave == 0) potrf_invert_couple(S, psh, nreal - PNB);
  __syncthreads();
  // W[x][y]: x > y at S[y][x], x == y in dinv, x < y zero
  const int lane = t & 63, g = lane >> 4, n = lane & 15;
  for (int dist = 1; dist < npan; ++dist) {
    const int nblk = npan - dist;                 // blocks (b + dist, b), b = 0 .. nblk-1, four 16 x 16 tiles each
    for (int tile = wave; tile < 4 * nblk; tile += 16) {   // T = sum_x L[ib + r][x] W[x][jb + c],  jb <= x < ib
      const int b = tile >> 2, rt = (tile >> 1) & 1, ct = tile & 1, jb = b * PNB, ib = (b + dist) * PNB;
      const int y = jb + 16 * ct + n;
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int ks4 = 4 * ct; ks4 < 8 * dist; ks4 += 4) {     // four k-steps of loads at a time
        double bv[4], av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int x = jb + 4 * (ks4 + u) + g;
          const double sv = S[y * PBLD + x], dv = dinv[y];
          bv[u] = x > y ? sv : (x == y ? dv : 0.0);
          av[u] = S[(ib + 16 * rt + n) * PBLD + x];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) potrf_mfma16(acc, av[u], bv[u]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) T[b][16 * rt + 4 * r + g][16 * ct + n] = acc[r];
    }
    __syncthreads();
    for (int tile = wave; tile < 4 * nblk; tile += 16) {   // W_ij = -W_ii T
      const int b = tile >> 2, rt = (tile >> 1) & 1, ct = tile & 1, jb = b * PNB, ib = (b + dist) * PNB;
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int ks4 = 0; ks4 < 4 * (rt + 1); ks4 += 4) {
        double bv[4], av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = 4 * (ks4 + u) + g, row = 16 * rt + n;
          bv[u] = T[b][m][16 * ct + n];
          const double sv = S[(ib + m) * PBLD + ib + row], dv = dinv[ib + row];
          av[u] = row > m ? sv : (row == m ? dv : 0.0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) potrf_mfma16(acc, av[u], bv[u]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(jb + 16 * ct + n) * PBLD + ib + 16 * rt + 4 * r + g] = -acc[r];
    }
    __syncthreads();
  }
  return true;
}

__global__ void __launch_bounds__(1024)
k_potrf_diag(const double* __restrict__ A, double* __restrict__ L, double* __restrict__ W, int64_t ld, int j0, int* info, int npan, double tol) {
  // npan = number of 32-column panels that hold real rows; the rest of the block is identity padding (L = W = I there)
  extern __shared__ double S[];   // [128][129]
  __shared__ PotrfShared psh;
  const int t = threadIdx.x;
  for (int idx = t; idx < PB * PB; idx += 1024) {
    const int i = idx >> 7, j = idx & 127;
    S[i * PBLD + j] = (j <= i) ? A[(int64_t)i * ld + j] : 0.0;
  }
  __syncthreads();
  if (!potrf_diag_lds(S, psh, j0, info, npan, W != nullptr, tol)) return;
  if (L) {
    for (int idx = t; idx < PB * PB; idx += 1024) {
      const int i = idx >> 7, j = idx & 127;
      L[(int64_t)i * ld + j] = (j <= i) ? S[i * PBLD + j] : 0.0;
    }
  }
  if (W) {
    for (int idx = t; idx < PB * PB; idx += 1024) {
      const int i = idx >> 7, j = idx & 127;
      W[(int64_t)i * ld + j] = potrf_wget(S, psh.dinv, i, j);
    }
  }
}

}  // namespace zigp
