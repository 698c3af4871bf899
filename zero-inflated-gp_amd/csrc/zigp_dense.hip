// Dense zero-inflated-GP ELBO path: host orchestration + C-ABI (include/zigp.h).
//
// Per ELBO step (value + gradient), for each latent h in {f, g}:
//   MxM stage   Kuu = k(Z,Z)+jitter I ; L = chol(Kuu) ; W = L^-1                    (OnOffSVGP.py:96-97, main.py:267-268)
//   per chunk   K = k(Z, Xc) ; A1 = W K ; A2 = W^T A1 ; column sums -> mean, var      (main.py:266-303; A2 lives in accumulators only)
//               H = (W diag(s^2)) A2 ; J' = W^T H - A2 = Q A2 = (Q W^T) A1            (gradient panels, independent of the cotangents)
//               point-wise probit / likelihood / reverse pass -> gm, gv               (OnOffSVGP.py:168-204, OnOffLikelihood.py:30-32)
//               reverse of the two triangular solves, with G = diag(gv), v = W u, alpha = W^T v:
//                 E = W dA2 = v gm^T + 2 H G ;  F = dK = W^T(E - 2 A1 G) = alpha gm^T + 2 J' G
//                 dL = -tril(F A1^T + A2 E^T) = -tril(alpha (A1 gm)^T + (A2 gm) v^T + 2 [J' G A1^T + A2 G H^T])
//               and, with T = W diag(s^2) W^T (so H = T A1, J' = W^T (T - I) A1) and C1 = A1 G A1^T:
//                 J' G A1^T + A2 G H^T = W^T (T C1 + C1 T - C1)
//               so all four O(M^2 N) triangular products run back to back before the point-wise stage, and ONE
//               gv-weighted symmetric rank-N update C1 (gv applied as a k-scale inside the GEMM core, split-K,
//               fixed order) replaces the two rank-N updates of the literal reverse pass; the rest is O(M^3).
//   MxM stage   Kuu-bar = sym(W^T Phi(L^T dL) W) - dKL/dKuu ; -> dZ, dell, dvar        (Cholesky reverse, Murray 2016 / TF CholeskyGrad)
#include "zigp_ctx.h"
#include "zigp_kernels.h"
#include "zigp_host.h"
#include "zigp_comm.h"
#include <algorithm>
#include <cmath>
#include <functional>

using namespace zigp;

namespace {

struct HostLatent {
  int M; const double *Z, *u, *s, *ell; double var;
};

// Parameters of both latents to the device (zero-padded to Mp): ONE staged image [Z | ell | u | s | Zs] x 2 and one copy -- the eight
// separate copies of the first version were eight launches (~7 us apart) in front of a launch-bound M x M stage.  lt.Z / ell / u / s
// are views into the context's parameter arena.
int latents_upload(zigp_ctx* c, const HostLatent (&hl)[2], int D) {
  size_t off[2][6], total = 0;
  for (int h = 0; h < 2; ++h) {
    Latent& lt = c->lat[h];
    lt.M = hl[h].M;
    lt.Mp = (int)round_up(hl[h].M, BM);
    lt.var = hl[h].var;
    const size_t Mp = lt.Mp;
    off[h][0] = total; total += Mp * D;      // Z
    off[h][1] = total; total += MAXD;        // ell
    off[h][2] = total; total += Mp;          // u
    off[h][3] = total; total += Mp;          // s
    off[h][4] = total; total += Mp * D;      // Zs = Z scaled to k_kuf_build's units (the same doubles the kernel multiplies into x)
    off[h][5] = total;
  }
  ZIGP_ENSURE(c, c->parm, total);
  ZIGP_PINNED(c, img, total);
  memset(img, 0, sizeof(double) * total);
  for (int h = 0; h < 2; ++h) {
    const HostLatent& q = hl[h];
    memcpy(img + off[h][0], q.Z, sizeof(double) * q.M * D);
    memcpy(img + off[h][1], q.ell, sizeof(double) * D);
    memcpy(img + off[h][2], q.u, sizeof(double) * q.M);
    memcpy(img + off[h][3], q.s, sizeof(double) * q.M);
    const KufHyp kh = make_kuf_hyp(q.ell, q.var, D);
    double spread = 0.0;
    for (int d = 0; d < MAXD; ++d) {      // centre of k_kgrad's moment sums: the mean inducing input
      double sum = 0.0;
      if (d < D) for (int m = 0; m < q.M; ++m) sum += q.Z[(size_t)m * D + d];
      c->lat[h].zc[d] = q.M > 0 ? sum / q.M : 0.0;
      if (d < D) for (int m = 0; m < q.M; ++m) spread = std::max(spread, std::fabs(q.Z[(size_t)m * D + d] - c->lat[h].zc[d]) / q.ell[d]);
    }
    // inducing inputs further than KG_EXACT_SPREAD lengthscales from their mean (or not finite): per-row differences instead of the shift
    c->lat[h].kg_exact = !(spread <= KG_EXACT_SPREAD);
    for (int m = 0; m < q.M; ++m)
      for (int d = 0; d < D; ++d) img[off[h][4] + (size_t)m * D + d] = q.Z[(size_t)m * D + d] * kh.scale[d];
  }
  ZIGP_HIP(c, hipMemcpyAsync(c->parm.p, img, sizeof(double) * total, hipMemcpyHostToDevice, c->stream));
  for (int h = 0; h < 2; ++h) {
    Latent& lt = c->lat[h];
    const size_t Mp = lt.Mp;
    lt.Z.alias(c->parm.p + off[h][0], Mp * D); lt.ell.alias(c->parm.p + off[h][1], MAXD);
    lt.u.alias(c->parm.p + off[h][2], Mp); lt.s.alias(c->parm.p + off[h][3], Mp); lt.Zs.alias(c->parm.p + off[h][4], Mp * D);
    ZIGP_ENSURE(c, lt.s2, Mp);
    ZIGP_ENSURE(c, lt.Kuu, Mp * Mp);
    ZIGP_ENSURE(c, lt.L, Mp * Mp);
    ZIGP_ENSURE(c, lt.W, Mp * Mp);
    ZIGP_ENSURE(c, lt.T1, Mp * Mp);
    ZIGP_ENSURE(c, lt.vec, 4 * Mp + 8);
    ZIGP_ENSURE(c, lt.Wp, Mp * Mp);
    ZIGP_ENSURE(c, lt.Wt, Mp * Mp);
    ZIGP_ENSURE(c, lt.Wpt, Mp * Mp);
    ZIGP_ENSURE(c, lt.P, Mp * Mp); ZIGP_ENSURE(c, lt.Qt, Mp * Mp); ZIGP_ENSURE(c, lt.Rt, Mp * Mp);
  }
  return 0;
}

// MxM forward of BOTH latents (kernels only): Kuu, L = chol, W = L^-1 (+ W^T, (W diag(s^2))^T), the KL pieces v = W u,
// alpha = W^T v, dkinv = diag(K^-1), kl -> vec[3*Mp], and W' = W diag(s^2) for gradient steps.  Latent f runs on the main stream and g
// on stream2 (the caller forks / joins); the launches ALTERNATE between the two chains step by step, so that both streams are fed
// from the start (see potrf_trtri_jobs).
int latents_forward(zigp_ctx* c, const HostLatent (&hl)[2], int D, double jitter, bool with_kl, bool need_grad) {
  const hipStream_t st[2] = {c->stream_main, c->stream2};
  struct Restore { zigp_ctx* c; ~Restore() { c->stream = c->stream_main; } } restore{c};
  for (int h = 0; h < 2; ++h) {
    Latent& lt = c->lat[h];
    const int Mp = lt.Mp;
    c->stream = st[h];
    KernHyp hyp = make_hyp(hl[h].ell, hl[h].var, D);
    hipLaunchKernelGGL(k_kuu_setup, dim3(ceil_div((int64_t)Mp * Mp, 256)), dim3(256), 0, c->stream, lt.Z.p, (int64_t)hl[h].M, hyp, jitter, lt.Kuu.p,
                       lt.L.p, lt.W.p, lt.s.p, lt.s2.p, (int64_t)Mp);
    ZIGP_HIP(c, hipGetLastError());
  }
  {
    PotrfJob jobs[2];
    for (int h = 0; h < 2; ++h) {
      Latent& lt = c->lat[h];
      jobs[h] = PotrfJob{lt.L.p, lt.W.p, lt.T1.p, lt.Mp, true, lt.M, pivot_tol(hl[h].var, jitter, c->pivot_rtol), true, &lt.sk};
    }
    ZIGP_TRY(potrf_trtri_jobs(c, 2, jobs, st));
  }
  for (int step = 0; step < 8; ++step)
    for (int h = 0; h < 2; ++h) {
      Latent& lt = c->lat[h];
      const int Mp = lt.Mp;
      c->stream = st[h];
      if (step == 0) lt.P_ready = false;
      double* v = lt.vec.p; double* alpha = v + Mp; double* dkinv = v + 2 * Mp; double* klv = v + 3 * Mp;
      switch (step) {
        case 0:   // W^T: the m-contiguous image of the factor that the lower-triangular product A1 = W K reads
          hipLaunchKernelGGL(k_transpose_scale, dim3(Mp / 32, Mp / 32), dim3(32, 8), 0, c->stream, lt.W.p, lt.s2.p, (int64_t)Mp, lt.Wt.p, lt.Wpt.p);
          break;
        // v = W u and alpha = W^T v are needed by the KL value, by the fused mean (v^T A1) and by the rank-1 parts of the data-term gradient
        case 1: if (with_kl) hipLaunchKernelGGL(k_gemv_rows, dim3(Mp), dim3(256), 0, c->stream, lt.W.p, lt.u.p, (int64_t)Mp, v); break;
        case 2: if (with_kl) hipLaunchKernelGGL(k_kl_cols, dim3(Mp / 64), dim3(64, COL_LANES), 0, c->stream, lt.W.p, v, (int64_t)Mp, alpha, dkinv); break;
        case 3: if (with_kl) hipLaunchKernelGGL(k_kl_value, dim3(1), dim3(256), 0, c->stream, v, lt.L.p, lt.s.p, dkinv, lt.M, (int64_t)Mp, klv); break;
        case 4:   // W' = W diag(s^2) (operand of the reverse M x M stage)
          if (need_grad) hipLaunchKernelGGL(k_colscale, dim3(ceil_div((int64_t)Mp * Mp, 256)), dim3(256), 0, c->stream, lt.W.p, lt.s2.p, (int64_t)Mp, lt.Wp.p);
          break;
        case 5:   // P = W^T W = Kuu^-1 (the reverse M x M stage needs it anyway and takes it from here)
          if (need_grad) {
            const int nb = Mp / BM, kb = BM / BK;
            ZIGP_TRY((run_gemm_sk<LAY_MNCONTIG, LAY_MNCONTIG>(c, lt.sk, "s", nb, [&](int bi, int bj, int& k0, int& k1) { k0 = std::max(bi, bj) * kb; k1 = nb * kb; },
                                                              lt.W.p, lt.W.p, lt.P.p, Mp, SK_STORE, 1.0, false)));
            lt.P_ready = true;
          }
          break;
        case 6:   // Q^T = diag(s^2) P - I: J' = W^T (W diag(s^2) A2) - A2 = (P diag(s^2) - I) A2 = Q A2 is ONE full product per chunk
          if (need_grad) hipLaunchKernelGGL(k_rowscale_minus_eye, dim3(ceil_div((int64_t)Mp * Mp, 256)), dim3(256), 0, c->stream, lt.P.p, lt.s2.p, (int64_t)Mp, lt.Qt.p);
          break;
        case 7:   // R^T = W Q^T (W lower triangular: k blocks 0 .. bi): J' = Q (W^T A1) = (Q W^T) A1 reads the A1 panel, so that the A2 panel
                  // has no reader left and is never written (r6; 8 Mp Nc bytes per latent and chunk, the A2 product 66 -> 69 TFLOP/s)
          if (need_grad) {
            const int nb = Mp / BM, kb = BM / BK;
            ZIGP_TRY((run_gemm_sk<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.sk, "rt", nb, [&](int bi, int, int& k0, int& k1) { k0 = 0; k1 = (bi + 1) * kb; },
                                                             lt.W.p, lt.Qt.p, lt.Rt.p, Mp, SK_STORE, 1.0, false)));
          }
          break;
        default: break;
      }
      ZIGP_HIP(c, hipGetLastError());
    }
  return 0;
}

// Kuf panel of one latent for the chunk starting at row n0 (HBM-write bound; runs on the side stream under the previous chunk's SYRKs)
int latent_chunk_kuf(zigp_ctx* c, Latent& lt, const double* dX, int64_t Nrows, int64_t n0, int64_t Nc, int D, const double* ell_host) {
  const int Mp = lt.Mp;
  const KufHyp kh = make_kuf_hyp(ell_host, lt.var, D);
  ProfScope ps(c, PC_KUF);
  const dim3 grid((unsigned)(Nc / 512), Mp / 16), block(256);
#define ZIGP_KUF(DD) \
  case DD: hipLaunchKernelGGL(k_kuf_build<DD>, grid, block, 0, c->stream, dX, Nrows, n0, lt.Zs.p, lt.M, kh, lt.K.p, Nc); break;
  switch (D) {
    ZIGP_KUF(1) ZIGP_KUF(2) ZIGP_KUF(3) ZIGP_KUF(4) ZIGP_KUF(5) ZIGP_KUF(6) ZIGP_KUF(7) ZIGP_KUF(8)
    default: return fail_arg(c, "D out of range");
  }
#undef ZIGP_KUF
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

// Tile order of the chunk's two triangular products: paired units + merged launch where both latents' units fill waves of the 512 slots
// (trmm_paired_pays), and then the plan for a last wave that is not full (trmm_tail_plan; merged launch = [latent f's units | latent g's
// units], unit counts multiples of 8).  Returns `paired`.
bool chunk_trmm_plan(int Mp0, int Mp1, int nbn, bool tail_on, TrmmTail& tail) {
  const int u0 = ((nbn + 7) / 8) * 8 * ((Mp0 / BM + 1) / 2), u1 = ((nbn + 7) / 8) * 8 * ((Mp1 / BM + 1) / 2);
  const bool paired = trmm_paired_pays(nbn * ((Mp0 / BM + 1) / 2 + (Mp1 / BM + 1) / 2));
  tail = TrmmTail{{0, 0}, {64, 64}};
  if (paired && tail_on) tail = trmm_tail_plan(u0, Mp0 / BM, u1, Mp1 / BM);
  return paired;
}

// Forward panels of both latents for one chunk: A1, A2 (+ J' when a gradient is wanted), column partials.
// Where the triangular products run the paired order (trmm_paired_pays: cfg3, cfg2), each product class is ONE launch for both latents
// (run_gemm2: latent g's workgroups fill the tail of latent f's, three launch boundaries fewer per chunk; cfg3 -0.4 ... -0.8 % same-box,
// profiles/r05l_ab_merge_fg.log, r05s_ab_milestones.log).  In the LPT regime the products stay per latent, in the order A1 A2 J' (f), A1 A2 J' (g) (merged there:
// cfg2 +1.2 %), and so does the rank-N update everywhere (its 512-workgroup split-K plan fills the chip exactly; merged +0.2 %).
int chunk_forward(zigp_ctx* c, int64_t Nc, bool need_grad, const PwArgs* fuse_pw = nullptr, bool* fused = nullptr, const std::function<int()>& after_a1 = nullptr) {
  const int nbn = (int)(Nc / BN);
  struct Set { TileList tl, tu, tf; double fl; GemmArgs a1, a2, j; EpiStoreColsum e1; EpiColsum e2; } q[2];
  TrmmTail tail;
  const bool paired = chunk_trmm_plan(c->lat[0].Mp, c->lat[1].Mp, nbn, c->trmm_tail, tail), merge = paired;
  for (int h = 0; h < 2; ++h) {
    Latent& lt = c->lat[h];
    const int Mp = lt.Mp, nbm = Mp / BM;
    const int np = Mp / 32;   // allocated partial rows per fused column sum (a kernel writes one per wave tile: 64 or 32 rows)
    ZIGP_TRY(tiles_trmm_lower(c, nbm, nbn, q[h].tl, paired, tail.units[h], tail.bins[h]));
    ZIGP_TRY(tiles_trmm_upper(c, nbm, nbn, q[h].tu, paired, tail.units[h], tail.bins[h]));
    if (need_grad) ZIGP_TRY(tiles_full_xcd(c, nbm, nbn, nbm * (BM / BK), q[h].tf));
    q[h].fl = (double)lt.M * lt.M * (double)Nc;
    // A1 = W K ; partial column sums  v^T A1 (= mean, since A2^T u = A1^T W u)  and  sum A1^2
    q[h].a1 = mk_args(lt.Wt.p, Mp, lt.K.p, Nc, lt.A1.p, Nc);
    q[h].e1 = EpiStoreColsum{lt.vec.p, nullptr, lt.part.p, lt.part.p + (size_t)np * Nc};
    // A2 = W^T A1 ; partial column sums  sum s^2 A2^2 -- the sums only: the panel itself has no reader (EpiColsum; ldc = stride of the partial rows)
    q[h].a2 = mk_args(lt.W.p, Mp, lt.A1.p, Nc, nullptr, Nc);
    q[h].e2 = EpiColsum{nullptr, lt.s2.p, nullptr, lt.part.p + (size_t)2 * np * Nc};
    // J' = Q A2 = (Q W^T) A1, Q = Kuu^-1 diag(s^2) - I (M x M, dense): the two triangular products H = W diag(s^2) A2, J' = W^T H - A2 of the
    // reverse pass as ONE full product of the same flop count -- every tile the full k range (no triangular padding, half as many prologues and
    // epilogues per flop), no H panel written and read back, no operand tile in the epilogue (r4: J' 61.9 -> 70.2 TFLOP/s, step -3.8 %,
    // profiles/r04ak_ab_qform.log; the two-product form is in tools/r4_experiment_arms.patch).  r6: with R = Q W^T formed once per step in the
    // M x M stage (latents_forward) the product reads the A1 panel, not A2.
    q[h].j = mk_args(lt.Rt.p, Mp, lt.A1.p, Nc, lt.Jp.p, Nc);
  }
  if (merge) {
    {
      ProfScope ps(c, PC_GEMM_A1, q[0].fl + q[1].fl);
      ZIGP_TRY((run_gemm2<LAY_MNCONTIG, LAY_MNCONTIG, false, TRI_A_LOWER>(c, q[0].tl, q[0].a1, q[0].e1, q[1].tl, q[1].a1, q[1].e1)));
    }
    if (after_a1) ZIGP_TRY(after_a1());     // the Kuf panels have had their only reader of a value-only / predict pass
    {
      ProfScope ps(c, PC_GEMM_A2, q[0].fl + q[1].fl);
      ZIGP_TRY((run_gemm2<LAY_MNCONTIG, LAY_MNCONTIG, false, TRI_A_UPPER>(c, q[0].tu, q[0].a2, q[0].e2, q[1].tu, q[1].a2, q[1].e2)));
    }
    if (need_grad) {
      ProfScope ps(c, PC_GEMM_J, 2.0 * (q[0].fl + q[1].fl));
      if (fuse_pw && (Nc / PW_PTS) % 16 == 0) {
        ZIGP_TRY(run_gemm_j_pw(c, q[0].tf, q[0].j, q[1].tf, q[1].j, *fuse_pw, (int)(Nc / PW_PTS)));
        if (fused) *fused = true;
      } else
        ZIGP_TRY((run_gemm2<LAY_MNCONTIG, LAY_MNCONTIG, false>(c, q[0].tf, q[0].j, EpiStorePanel(), q[1].tf, q[1].j, EpiStorePanel())));
    }
    return 0;
  }
  for (int h = 0; h < 2; ++h) {
    {
      ProfScope ps(c, PC_GEMM_A1, q[h].fl);
      ZIGP_TRY((run_gemm<LAY_MNCONTIG, LAY_MNCONTIG, false, TRI_A_LOWER>(c, q[h].tl, q[h].a1, q[h].e1)));
    }
    if (h == 1 && after_a1) ZIGP_TRY(after_a1());
    {
      ProfScope ps(c, PC_GEMM_A2, q[h].fl);
      ZIGP_TRY((run_gemm<LAY_MNCONTIG, LAY_MNCONTIG, false, TRI_A_UPPER>(c, q[h].tu, q[h].a2, q[h].e2)));
    }
    if (need_grad) {
      ProfScope ps(c, PC_GEMM_J, 2.0 * q[h].fl);
      ZIGP_TRY((run_gemm<LAY_MNCONTIG, LAY_MNCONTIG, false>(c, q[h].tf, q[h].j, EpiStorePanel())));
    }
  }
  return 0;
}

// Kuf-cotangent reductions of one latent and chunk (HBM-read bound; runs on the side stream under the chunk's SYRKs)
int latent_chunk_kgrad(zigp_ctx* c, Latent& lt, const double* dX, int64_t Nrows, int64_t n0, int64_t Nc, int D, const double* ell_host) {
  const int Mp = lt.Mp;
  KgCentre hyp;
  for (int d = 0; d < MAXD; ++d) hyp.c[d] = lt.zc[d];
  double* alpha = lt.vec.p + Mp;
  {
    ProfScope ps(c, PC_RED);
    const dim3 gk((unsigned)ceil_div(lt.M, KG_ROWS), KG_SPLIT), bk(256);
    const int64_t slab = (int64_t)Mp * (2 + 2 * D);
#define ZIGP_KGRAD(DD)                                                                                                            \
  case DD:                                                                                                                        \
    if (lt.kg_exact)                                                                                                              \
      hipLaunchKernelGGL((k_kgrad<DD, true>), gk, bk, 0, c->stream, lt.Jp.p, lt.K.p, alpha, lt.gm.p, lt.gv.p, dX, Nrows, n0, lt.Z.p, lt.M, Nc, \
                         slab, hyp, lt.krow.p);                                                                                   \
    else                                                                                                                          \
      hipLaunchKernelGGL((k_kgrad<DD, false>), gk, bk, 0, c->stream, lt.Jp.p, lt.K.p, alpha, lt.gm.p, lt.gv.p, dX, Nrows, n0, lt.Z.p, lt.M, Nc, \
                         slab, hyp, lt.krow.p);                                                                                   \
    break;
    switch (D) {
      ZIGP_KGRAD(1) ZIGP_KGRAD(2) ZIGP_KGRAD(3) ZIGP_KGRAD(4) ZIGP_KGRAD(5) ZIGP_KGRAD(6) ZIGP_KGRAD(7) ZIGP_KGRAD(8)
      default: return fail_arg(c, "D out of range");
    }
#undef ZIGP_KGRAD
    ZIGP_HIP(c, hipGetLastError());
  }
  return 0;
}

// Rank-N update of the lower-triangular cotangent of one latent and chunk
int latent_chunk_syrk(zigp_ctx* c, Latent& lt, int64_t Nc) {
  const int Mp = lt.Mp, nbm = Mp / BM;
  TileList ts;
  ZIGP_TRY(tiles_syr2k(c, nbm, (int)(Nc / BK), syr_plan(nbm), ts));
  const double fl = (double)lt.M * lt.M * (double)Nc;
  {
    ProfScope ps(c, PC_SYR2K, fl);   // planes += tril(A1 G A1^T)   (G = diag(gv) applied as k-scale on the B operand)
    GemmArgs g = mk_args(lt.A1.p, Nc, lt.A1.p, Nc, lt.dLpart.p, Mp);
    g.slice_stride = (int64_t)Mp * Mp; g.kscale = lt.gv.p;
    ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_KCONTIG, true, TRI_C_LOWER>(c, ts, g, EpiAccum())));
  }
  return 0;
}

// MxM backward: G = dELBO/dKuu (symmetric) -> krow accumulators.
int latent_mxm_backward(zigp_ctx* c, Latent& lt, int D, double jitter, bool with_data, bool with_kl) {
  const int Mp = lt.Mp, nb = Mp / BM, kb = BM / BK;
  const size_t mm = (size_t)Mp * Mp;
  ZIGP_ENSURE(c, lt.T1, mm); ZIGP_ENSURE(c, lt.T2, mm); ZIGP_ENSURE(c, lt.T3, mm); ZIGP_ENSURE(c, lt.G, mm);
  const int gridmm = ceil_div((int64_t)mm, 256);
  double* S = lt.T1.p;
  if (with_data) {
    TileList ta, tb, tc, td;
    // rank-1 seeds from K gm (accumulated by k_kgrad): A1 gm = W (K gm), A2 gm = du = W^T (A1 gm)
    {
      double* kgm = lt.vec.p + 3 * Mp + 8;
      hipLaunchKernelGGL(k_gather, dim3(ceil_div(Mp, 256)), dim3(256), 0, c->stream, lt.krow.p, 2 + 2 * D, 1 + 2 * D, Mp, KG_SPLIT,
                         (int64_t)Mp * (2 + 2 * D), kgm);
      hipLaunchKernelGGL(k_gemv_rows, dim3(Mp), dim3(256), 0, c->stream, lt.W.p, kgm, (int64_t)Mp, lt.a1gm.p);
      hipLaunchKernelGGL(k_gemv_cols, dim3(Mp / 64), dim3(64, COL_LANES), 0, c->stream, lt.W.p, lt.a1gm.p, (int64_t)Mp, lt.du.p);
    }
    // C1 = sym(sum_s planes) -> T1
    {
      const SyrPlan sp = syr_plan(nb);
      const int nt = Mp / 32;
      hipLaunchKernelGGL(k_sym_from_planes, dim3(nt * (nt + 1) / 2), dim3(256), 0, c->stream, lt.dLpart.p, sp.So, sp.Sd, (int64_t)Mp, lt.T1.p);
    }
    // dsq = diag(A2 G A2^T) = diag(W^T C1 W): Y = C1 W -> T3 ; dsq[m] = sum_k W[k][m] Y[k][m]
    ZIGP_TRY((run_gemm_sk<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.sk, "y", nb, [&](int bi, int bj, int& k0, int& k1) { k0 = bj * kb; k1 = nb * kb; },
                                                     lt.T1.p, lt.W.p, lt.T3.p, Mp, SK_STORE, 1.0, false)));
    hipLaunchKernelGGL(k_coldot, dim3(Mp / 64), dim3(64, COL_LANES), 0, c->stream, lt.W.p, lt.T3.p, (int64_t)Mp, lt.dsq.p);
    // T = (W diag(s^2)) W^T -> T2   (both factors lower triangular: k <= min(i,j))
    ZIGP_TRY((run_gemm_sk<LAY_KCONTIG, LAY_KCONTIG>(c, lt.sk, "tt", nb, [&](int bi, int bj, int& k0, int& k1) { k0 = 0; k1 = (std::min(bi, bj) + 1) * kb; },
                                                    lt.Wp.p, lt.W.p, lt.T2.p, Mp, SK_STORE, 1.0, false)));
    // U = T C1 -> T3 ; V = U + U^T - C1 -> G
    ZIGP_TRY((run_gemm_sk<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.sk, "full", nb, [&](int, int, int& k0, int& k1) { k0 = 0; k1 = nb * kb; },
                                                     lt.T2.p, lt.T1.p, lt.T3.p, Mp, SK_STORE, 1.0, false)));
    hipLaunchKernelGGL(k_uut_minus, dim3(gridmm), dim3(256), 0, c->stream, lt.T3.p, lt.T1.p, (int64_t)Mp, lt.G.p);
    // R = W^T V (lower part) -> T2
    auto lower_up = [&](int bi, int bj, int& k0, int& k1) { if (bj <= bi) { k0 = bi * kb; k1 = nb * kb; } else { k0 = 0; k1 = 0; } };
    ZIGP_TRY((run_gemm_sk<LAY_MNCONTIG, LAY_MNCONTIG>(c, lt.sk, "r", nb, lower_up, lt.W.p, lt.G.p, lt.T2.p, Mp, SK_STORE, 1.0, true)));
    // dL = -tril(alpha (A1 gm)^T + (A2 gm) v^T + 2 R) -> T1
    hipLaunchKernelGGL(k_dl_assemble, dim3(gridmm), dim3(256), 0, c->stream, lt.T2.p, (int64_t)Mp, lt.vec.p + Mp, lt.a1gm.p, lt.du.p,
                       lt.vec.p, lt.T1.p);
    // Q = Phi(L^T dL) -> T2  (upper tiles are not computed)
    ZIGP_TRY((run_gemm_sk<LAY_MNCONTIG, LAY_MNCONTIG>(c, lt.sk, "r", nb, lower_up, lt.L.p, lt.T1.p, lt.T2.p, Mp, SK_PHI, 1.0, true)));
    // T = Q W -> T3 (lower)
    ZIGP_TRY((run_gemm_sk<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.sk, "t", nb, [&](int bi, int bj, int& k0, int& k1) {
      if (bj <= bi) { k0 = bj * kb; k1 = (bi + 1) * kb; } else { k0 = 0; k1 = 0; } }, lt.T2.p, lt.W.p, lt.T3.p, Mp, SK_STORE, 1.0, true)));
    // S = W^T T -> T1
    ZIGP_TRY((run_gemm_sk<LAY_MNCONTIG, LAY_MNCONTIG>(c, lt.sk, "s", nb, [&](int bi, int bj, int& k0, int& k1) { k0 = std::max(bi, bj) * kb; k1 = nb * kb; },
                                                      lt.W.p, lt.T3.p, lt.T1.p, Mp, SK_STORE, 1.0, false)));
  }
  double* P = lt.P_ready ? lt.P.p : lt.T2.p; double* PSP = lt.G.p;
  if (with_kl) {
    // P = W^T W -> T2   (a gradient step has it from the forward stage: latents_forward)
    if (!lt.P_ready)
      ZIGP_TRY((run_gemm_sk<LAY_MNCONTIG, LAY_MNCONTIG>(c, lt.sk, "s", nb, [&](int bi, int bj, int& k0, int& k1) { k0 = std::max(bi, bj) * kb; k1 = nb * kb; },
                                                        lt.W.p, lt.W.p, lt.T2.p, Mp, SK_STORE, 1.0, false)));
    // Ps = diag(s2) P -> T3 ; PSP = P Ps -> G
    hipLaunchKernelGGL(k_rowscale, dim3(gridmm), dim3(256), 0, c->stream, P, lt.s2.p, (int64_t)Mp, lt.T3.p);
    ZIGP_TRY((run_gemm_sk<LAY_KCONTIG, LAY_MNCONTIG>(c, lt.sk, "full", nb, [&](int, int, int& k0, int& k1) { k0 = 0; k1 = nb * kb; },
                                                     P, lt.T3.p, lt.G.p, Mp, SK_STORE, 1.0, false)));
  }
  // G = sym(S) - dKL/dKuu -> T3 (T3 free again)
  hipLaunchKernelGGL(k_sym_combine, dim3(gridmm), dim3(256), 0, c->stream, S, P, PSP, lt.vec.p + Mp, with_data ? 1 : 0, with_kl ? 1 : 0,
                     (int64_t)Mp, lt.T3.p);
  hipLaunchKernelGGL(k_kuu_grad, dim3(Mp), dim3(256), 0, c->stream, lt.T3.p, lt.Kuu.p, jitter, lt.Z.p, lt.M, D, (int64_t)Mp, lt.krow.p);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

int validate_params(zigp_ctx* c, const zigp_params* p) {
  if (!p) return fail_arg(c, "params is NULL");
  if (p->Mf <= 0 || p->Mg <= 0) return fail_arg(c, "Mf and Mg must be positive");
  if (p->D <= 0 || p->D > MAXD) return fail_arg(c, "D must be in [1, 8]");
  if (!p->Zf || !p->Zg || !p->u_fm || !p->u_gm || !p->u_fs_sqrt || !p->u_gs_sqrt || !p->ell_f || !p->ell_g)
    return fail_arg(c, "NULL pointer in params");
  if (!(p->var_f > 0) || !(p->var_g > 0) || !(p->noise > 0)) return fail_arg(c, "variances must be positive");
  for (int d = 0; d < p->D; ++d)
    if (!(p->ell_f[d] > 0) || !(p->ell_g[d] > 0)) return fail_arg(c, "lengthscales must be positive");
  for (int m = 0; m < p->Mf; ++m)
    if (!(p->u_fs_sqrt[m] > 0)) return fail_arg(c, "u_fs_sqrt must be positive (diagonal q_sqrt, transforms.positive)");
  for (int m = 0; m < p->Mg; ++m)
    if (!(p->u_gs_sqrt[m] > 0)) return fail_arg(c, "u_gs_sqrt must be positive (diagonal q_sqrt, transforms.positive)");
  return 0;
}

// ---- one call of the dense path (zigp_elbo / zigp_predict), in four stages: MxM forward, chunk loop, MxM backward, gather ----
struct DenseCall {
  const zigp_params* p; const double* dX; const double* dY; int64_t Nrows; int D;
  double jitter, scale, g_offset; int64_t row_begin, row_end; int include_kl; bool predict; double* d_out9;
  bool need_grad, has_rows;
  HostLatent hl[2]; const double* ell_h[2];
  int64_t Nc = 0;         // rows per full chunk
  int pw_blocks = 0;
  int* hinfo = nullptr;   // Cholesky status, staged with the other results
  bool prep_side = false; // buffers / zeroed accumulators / first Kuf panels were issued on the third stream (dense_mxm_forward)
};

// Parameters to the device, then the MxM forward of f on the main stream and of g on stream2 (dozens of small dependent launches each)
int dense_prepare_buffers(zigp_ctx* c, DenseCall& k);
int64_t dense_first_chunk_rows(const DenseCall& k) { return std::min<int64_t>(k.Nc, round_up(k.row_end - k.row_begin, 1024)); }
int dense_mxm_forward(zigp_ctx* c, DenseCall& k) {
  ZIGP_TRY(begin_staged_call(c));
  ZIGP_HIP(c, hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream));
  ZIGP_TRY(latents_upload(c, k.hl, k.D));
  // The call's buffers, its zeroed accumulators and the first chunk's Kuf panels need the uploaded parameters only: third stream, under
  // the two factorisation chains (which are dependent launches of <= 36 workgroups).  Not while kernels are being timed (they run alone).
  k.prep_side = c->overlap == 1 && !c->prof_on;
  if (k.prep_side) {
    struct Guard { zigp_ctx* c; ~Guard() { c->stream = c->stream_main; } } guard{c};
    ZIGP_HIP(c, hipEventRecord(c->ev_prep_fork, c->stream_main));
    ZIGP_HIP(c, hipStreamWaitEvent(c->stream3, c->ev_prep_fork, 0));
    c->stream = c->stream3;
    ZIGP_TRY(dense_prepare_buffers(c, k));
    if (k.has_rows)
      for (int h = 0; h < 2; ++h)
        ZIGP_TRY(latent_chunk_kuf(c, c->lat[h], k.dX, k.Nrows, k.row_begin, dense_first_chunk_rows(k), k.D, k.ell_h[h]));
    ZIGP_HIP(c, hipEventRecord(c->ev_prep, c->stream3));
  }
  {
    ProfScope ps(c, PC_MXM);     // wall time of the two concurrent chains: both events on the main stream, the second after the join
    TwoStream ts(c);
    ZIGP_TRY(ts.fork());
    ZIGP_TRY(latents_forward(c, k.hl, k.D, k.jitter, true, k.need_grad));
    ZIGP_TRY(ts.join());
  }
  return request_info(c, &k.hinfo);   // read after the final synchronisation
}

// Chunk size and per-call buffers.  The row range is cut into ceil(span / chunk) chunks of (nearly) equal size, a multiple of 1024, so
// that the last chunk is not a sliver whose GEMMs leave most of the 512 workgroup slots empty (N = 1e5, chunk 32768: 4 x 25600).
// Unless the caller fixed it (zigp_set_chunk), the chunk scales with 1 / M so that a launch keeps its ~2000 tiles (4 waves of the 512
// workgroup slots) and the panels their size: 32768 rows at M = 1024, 65536 at M = 512 (cfg2: 2 chunks instead of 4, 8.05 -> 7.6 ms)
int64_t auto_chunk_for(bool chunk_auto, int64_t chunk_set, int64_t Mp) {
  if (!chunk_auto) return chunk_set;
  return std::min<int64_t>(131072, std::max<int64_t>(32768, round_up(32768 * 1024 / std::max<int64_t>(Mp, 128), 1024)));
}
// Rows per pass for a row range of `span` rows.  A range of up to 131072 rows goes through in ONE pass unless the caller fixed the chunk: no
// chunk boundary (where the side stream's kgrads outlast the rank-N updates), one prologue / tail per product instead of two to four -- cfg2
// (1e5 rows, M = 512) 5.98 -> 5.81 ms, the 125 000-row shard of cfg3 23.4 -> 23.0 ms (tools/chunk_sweep.py, profiles/r04ao_chunk_sweep.log).
// The rule is bounded by the panels' bytes (4 panels of 8 Mp span bytes per latent: <= 9 GB, i.e. M <= 1024 at 131072 rows -- what was
// measured); beyond that, and on long ranges, the M-scaled chunk applies (cfg3: 32768 rows 167.8 ms, 65536: 170.2, 131072: 169.2).
int64_t chunk_rows_for(bool chunk_auto, int64_t chunk_set, int64_t Mp, int64_t span) {
  int64_t chunk = auto_chunk_for(chunk_auto, chunk_set, Mp);
  if (chunk_auto && span > 0 && span <= 131072 && 4 * 2 * 8 * Mp * round_up(span, 1024) <= ((int64_t)9 << 30)) chunk = 131072;
  if (span <= 0) return 1024;
  const int64_t nchunks = (span + chunk - 1) / chunk;
  return std::max<int64_t>(1024, round_up((span + nchunks - 1) / nchunks, 1024));
}
int dense_prepare_buffers(zigp_ctx* c, DenseCall& k) {
  const int64_t span = k.has_rows ? (k.row_end - k.row_begin) : 0;
  k.Nc = chunk_rows_for(c->chunk_auto, c->chunk, std::max(c->lat[0].Mp, c->lat[1].Mp), span);
  const int64_t Nc = k.Nc;
  const int D = k.D;
  k.pw_blocks = (int)(Nc / PW_PTS);
  ZIGP_ENSURE(c, c->pw_part, (size_t)k.pw_blocks * PW_ACC);
  ZeroRanges zr;               // the small accumulators of the call: one launch instead of a memset each
  zr.count = 0;
  static_assert(1 + 2 * 4 <= ZERO_RANGES_MAX, "one launch zeroes every small accumulator of a call");
  auto zero = [&](double* ptr, int64_t n) { zr.p[zr.count] = ptr; zr.n[zr.count] = n; ++zr.count; };
  zero(c->pw_part.p, (int64_t)k.pw_blocks * PW_ACC);
  for (int h = 0; h < 2; ++h) {
    Latent& lt = c->lat[h];
    const int Mp = lt.Mp;
    ZIGP_ENSURE(c, lt.gm, Nc); ZIGP_ENSURE(c, lt.gv, Nc);
    if (k.has_rows) {
      ZIGP_ENSURE(c, lt.K, (size_t)Mp * Nc); ZIGP_ENSURE(c, lt.A1, (size_t)Mp * Nc);
      ZIGP_ENSURE(c, lt.part, (size_t)3 * (Mp / 32) * Nc);
      if (k.need_grad) ZIGP_ENSURE(c, lt.Jp, (size_t)Mp * Nc);
    }
    if (k.need_grad) {
      ZIGP_ENSURE(c, lt.du, Mp); ZIGP_ENSURE(c, lt.dsq, Mp); ZIGP_ENSURE(c, lt.krow, (size_t)KG_SPLIT * Mp * (2 + 2 * D));
      const int S = syr_plan(Mp / BM).planes();
      ZIGP_ENSURE(c, lt.dLpart, (size_t)S * Mp * Mp);
      ZIGP_ENSURE(c, lt.a1gm, Mp);
      zero(lt.a1gm.p, Mp); zero(lt.du.p, Mp); zero(lt.dsq.p, Mp); zero(lt.krow.p, (int64_t)KG_SPLIT * Mp * (2 + 2 * D));
      if (k.has_rows) ZIGP_HIP(c, hipMemsetAsync(lt.dLpart.p, 0, sizeof(double) * S * Mp * Mp, c->stream));
    }
  }
  hipLaunchKernelGGL(k_zero_ranges, dim3(64), dim3(256), 0, c->stream, zr);      // at most 1 + 2 x 4 ranges
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

// point-wise stage of the chunk starting at row n0 (probit moments, expected log-likelihood, reverse pass to gm / gv)
PwArgs dense_pointwise_args(zigp_ctx* c, const DenseCall& k, int64_t n0, int64_t Nc) {
  PwArgs a;
  a.part_f = c->lat[0].part.p; a.part_g = c->lat[1].part.p; a.np_f = c->lat[0].Mp / 32; a.np_g = c->lat[1].Mp / 32;
  {
    constexpr int RW2 = Shape<WavesFor<LAY_MNCONTIG, LAY_MNCONTIG, false>::value>::RW, RW1 = RW2;
    static_assert(RW1 >= 32 && RW2 >= 32, "partial-row planes are allocated for 32-row wave tiles");
    a.np1_f = c->lat[0].Mp / RW1; a.np2_f = c->lat[0].Mp / RW2; a.np1_g = c->lat[1].Mp / RW1; a.np2_g = c->lat[1].Mp / RW2;
  }
  a.Y = k.dY; a.n0 = n0; a.row_end = k.row_end; a.Nc = Nc;
  a.var_f = k.p->var_f; a.var_g = k.p->var_g; a.noise = k.p->noise; a.g_offset = k.g_offset; a.scale = k.scale;
  a.gm_f = k.need_grad ? c->lat[0].gm.p : nullptr; a.gv_f = c->lat[0].gv.p; a.gm_g = c->lat[1].gm.p; a.gv_g = c->lat[1].gv.p;
  a.X = k.dX; a.D = k.D; a.mean_on = c->mean_on ? 1 : 0; a.mean_b = c->mean_b;
  for (int d = 0; d < MAXD; ++d) a.mean_a[d] = (d < k.D) ? c->mean_a[d] : 0.0;
  a.acc = c->pw_part.p; a.out9 = k.d_out9 ? k.d_out9 - k.row_begin : nullptr; a.ld9 = k.row_end - k.row_begin;
  return a;
}
int dense_pointwise(zigp_ctx* c, const DenseCall& k, int64_t n0, int64_t Nc) {
  ProfScope ps(c, PC_POINT);
  const PwArgs a = dense_pointwise_args(c, k, n0, Nc);
  const int nblk = (int)(Nc / PW_PTS);
  if (k.predict) hipLaunchKernelGGL(k_pointwise<true>, dim3(nblk), dim3(PW_THREADS), 0, c->stream, a);
  else hipLaunchKernelGGL(k_pointwise<false>, dim3(nblk), dim3(PW_THREADS), 0, c->stream, a);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

// ---- chunk loop.  The MFMA-bound GEMMs stay on the main stream; with zigp_set_overlap(1) the HBM-bound kernels of a chunk -- the two
// Kuf-cotangent reductions and the two Kuf panels of the NEXT chunk -- run on the side stream underneath the chunk's two SYRKs:
//   main:  [wait side]  A1 (f|g)  A2 (f|g)  [point-wise +] J' (f|g)  (record)  SYRK f  SYRK g
//   side:                                              (wait)    kgrad f  kgrad g  Kuf f'  Kuf g'  (record)
// K is only read by A1 (and its rows' x, z by kgrad), J' / gm only by kgrad: the next chunk's GEMMs wait for the side stream, nothing
// else is shared.  A chunk whose kernels are being timed (profiling samples every prof_every-th chunk) runs everything on the main
// stream, so the per-kernel durations bench.py reports are those of kernels running alone.
int dense_chunk_loop(zigp_ctx* c, const DenseCall& k) {
  if (!k.has_rows) return 0;
  const int64_t Nc_full = k.Nc, row_begin = k.row_begin, row_end = k.row_end;
  const int D = k.D;
  auto chunk_rows = [&](int64_t n0) { return std::min<int64_t>(Nc_full, round_up(row_end - n0, 1024)); };
  auto sampled = [&](int64_t n0) {   // kernel timing (HIP events) covers FULL chunks only, so the averages describe full-size launches
    if (!c->prof_on) return false;
    if (c->prof_every <= 1 || row_end - row_begin <= Nc_full) return true;   // every launch, the partial last chunk included
    return chunk_rows(n0) == Nc_full && (((n0 - row_begin) / Nc_full) % c->prof_every) == 0;
  };
  struct SideGuard { zigp_ctx* c; ~SideGuard() { c->stream = c->stream_main; c->prof_skip = false; } } side_guard{c};
  bool side_busy = false;
  c->prof_skip = c->prof_on && !sampled(row_begin);
  if (!k.prep_side)     // (otherwise built on the third stream under the M x M forward: dense_mxm_forward)
    for (int h = 0; h < 2; ++h) ZIGP_TRY(latent_chunk_kuf(c, c->lat[h], k.dX, k.Nrows, row_begin, chunk_rows(row_begin), D, k.ell_h[h]));
  for (int64_t n0 = row_begin; n0 < row_end; n0 += Nc_full) {
    const int64_t Nc = chunk_rows(n0);   // the last (partial) chunk shrinks to the next multiple of 1024 rows
    const int64_t n1 = n0 + Nc_full;
    const bool has_next = n1 < row_end;
    const bool timed = sampled(n0), timed_next = has_next && sampled(n1);
    c->prof_skip = c->prof_on && !timed;
    if (side_busy) {
      // A1 of this chunk needs the Kuf panels the side stream built behind the previous chunk's kgrads (which read J' and gm)
      ZIGP_HIP(c, hipStreamWaitEvent(c->stream_main, c->ev_join, 0));
      side_busy = false;
    }
    // gradient steps with zigp_set_overlap(1): the point-wise stage rides in the J' launch (run_gemm_j_pw: cfg3 -0.4 % same-box,
    // profiles/r05t_ab_fuse_pointwise.log); timed chunks and overlap 0 keep every kernel on its own, as for the side stream
    bool pw_fused = false;
    const PwArgs pwa = dense_pointwise_args(c, k, n0, Nc);
    // value-only ELBO and predict (r6): there are no rank-N updates to hide the next chunk's Kuf panels under, but K has ONE reader there
    // -- A1 -- so the side stream builds the next panels right behind this chunk's A1, beside its A2 product and point-wise stage
    // (cfg3 value-only: the 4 ms of panel building per pass were serial on the main stream)
    const bool kuf_fwd_side = c->overlap == 1 && c->fwd_kuf_side && !k.need_grad && has_next && !timed && !timed_next;
    std::function<int()> after_a1;
    if (kuf_fwd_side)
      after_a1 = [&]() -> int {
        ZIGP_HIP(c, hipEventRecord(c->ev_fork, c->stream_main));
        ZIGP_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        c->stream = c->stream2;
        int rc = 0;
        for (int h = 0; h < 2 && !rc; ++h) rc = latent_chunk_kuf(c, c->lat[h], k.dX, k.Nrows, n1, chunk_rows(n1), D, k.ell_h[h]);
        c->stream = c->stream_main;
        if (rc) return rc;
        ZIGP_HIP(c, hipEventRecord(c->ev_join, c->stream2));
        side_busy = true;
        return 0;
      };
    ZIGP_TRY(chunk_forward(c, Nc, k.need_grad, (c->overlap == 1 && k.need_grad && !k.predict && !timed) ? &pwa : nullptr, &pw_fused, after_a1));
    if (!pw_fused) ZIGP_TRY(dense_pointwise(c, k, n0, Nc));
    // side work of this chunk: its kgrads and the next chunk's Kuf panels (gradient mode only: without the SYRKs there is
    // nothing on the main stream to hide them under)
    const bool kgrad_side = c->overlap == 1 && k.need_grad && !timed;
    const bool kuf_side = kgrad_side && has_next && !timed_next;
    if (kgrad_side) {
      ZIGP_HIP(c, hipEventRecord(c->ev_fork, c->stream_main));
      ZIGP_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      c->stream = c->stream2;
      // (the next chunk's panels BEFORE this chunk's kgrads was measured and dropped: cfg2 +0.1 ms, HISTORY.md section 5 r4)
      for (int h = 0; h < 2; ++h) ZIGP_TRY(latent_chunk_kgrad(c, c->lat[h], k.dX, k.Nrows, n0, Nc, D, k.ell_h[h]));
      if (kuf_side)
        for (int h = 0; h < 2; ++h) ZIGP_TRY(latent_chunk_kuf(c, c->lat[h], k.dX, k.Nrows, n1, chunk_rows(n1), D, k.ell_h[h]));
      ZIGP_HIP(c, hipEventRecord(c->ev_join, c->stream2));
      side_busy = true;
      c->stream = c->stream_main;
    }
    if (k.need_grad) {
      if (!kgrad_side)
        for (int h = 0; h < 2; ++h) ZIGP_TRY(latent_chunk_kgrad(c, c->lat[h], k.dX, k.Nrows, n0, Nc, D, k.ell_h[h]));
      for (int h = 0; h < 2; ++h) ZIGP_TRY(latent_chunk_syrk(c, c->lat[h], Nc));
    }
    if (has_next && !kuf_side && !kuf_fwd_side) {   // a timed next chunk gets its panels from the main stream, with the side stream drained
      if (side_busy) { ZIGP_HIP(c, hipStreamWaitEvent(c->stream_main, c->ev_join, 0)); side_busy = false; }
      c->prof_skip = c->prof_on && !timed_next;
      for (int h = 0; h < 2; ++h) ZIGP_TRY(latent_chunk_kuf(c, c->lat[h], k.dX, k.Nrows, n1, chunk_rows(n1), D, k.ell_h[h]));
    }
  }
  if (side_busy) { ZIGP_HIP(c, hipStreamWaitEvent(c->stream_main, c->ev_join, 0)); side_busy = false; }
  c->prof_skip = false;
  return 0;
}

// The call's result vector (layout: k_dense_pack) is assembled on the device, summed over the ranks of a data-parallel run where it
// lies (zigp_comm_init; no-op otherwise), downloaded once and unpacked: the call's single synchronisation.
int dense_gather(zigp_ctx* c, DenseCall& k, double* elbo_data, double* kl, zigp_grads* grads) {
  const int D = k.D;
  size_t n = DP_HDR;
  DensePackArgs a;
  memset(&a, 0, sizeof(a));
  for (int h = 0; h < 2; ++h) {
    Latent& lt = c->lat[h];
    DensePackLat& L = a.lat[h];
    L.krow = lt.krow.p; L.du = lt.du.p; L.dsq = lt.dsq.p; L.vec = lt.vec.p; L.s = lt.s.p; L.ell = lt.ell.p;
    L.M = lt.M; L.Mp = lt.Mp; L.var = lt.var; L.out_off = (int64_t)n;
    if (k.need_grad) n += (size_t)lt.M * D + 2 * (size_t)lt.M + D;
  }
  a.pw = c->pw_part.p; a.pw_blocks = k.pw_blocks; a.D = D; a.need_grad = k.need_grad ? 1 : 0; a.include_kl = k.include_kl ? 1 : 0;
  a.mean_on = c->mean_on ? 1 : 0;
  ZIGP_ENSURE(c, c->packed, n);
  a.out = c->packed.p;
  hipLaunchKernelGGL(k_dense_pack, dim3(2), dim3(256), 0, c->stream, a);
  ZIGP_HIP(c, hipGetLastError());
  ZIGP_TRY(comm_allreduce(c, c->packed.p, n));
  double* hv = nullptr;
  ZIGP_TRY(download(c, c->packed.p, n, &hv));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  ZIGP_TRY(info_result(c, k.hinfo, "Kuu"));
  if (elbo_data) *elbo_data = hv[0];
  if (kl) *kl = hv[1];
  c->mean_db = hv[5];
  for (int d = 0; d < MAXD; ++d) c->mean_da[d] = hv[6 + d];
  if (k.need_grad) {
    double* gZ[2] = {grads->Zf, grads->Zg};
    double* gu[2] = {grads->u_fm, grads->u_gm};
    double* gs[2] = {grads->u_fs_sqrt, grads->u_gs_sqrt};
    double* gl[2] = {grads->ell_f, grads->ell_g};
    for (int h = 0; h < 2; ++h) {
      const size_t M = (size_t)c->lat[h].M;
      const double* o = hv + a.lat[h].out_off;
      if (gZ[h]) memcpy(gZ[h], o, sizeof(double) * M * D);
      if (gu[h]) memcpy(gu[h], o + M * D, sizeof(double) * M);
      if (gs[h]) memcpy(gs[h], o + M * D + M, sizeof(double) * M);
      if (gl[h]) memcpy(gl[h], o + M * D + 2 * M, sizeof(double) * D);
    }
    grads->var_f = hv[2]; grads->var_g = hv[3]; grads->noise = hv[4];
  }
  return 0;
}

// shared driver for zigp_elbo / zigp_predict
int run_dense(zigp_ctx* c, const zigp_params* p, const double* dX, const double* dY, int64_t Nrows, int D, double jitter,
              double scale, double g_offset, int64_t row_begin, int64_t row_end, int include_kl, bool predict, double* d_out9,
              double* elbo_data, double* kl, zigp_grads* grads) {
  DenseCall k;
  k.p = p; k.dX = dX; k.dY = dY; k.Nrows = Nrows; k.D = D; k.jitter = jitter; k.scale = scale; k.g_offset = g_offset;
  k.row_begin = row_begin; k.row_end = row_end; k.include_kl = include_kl; k.predict = predict; k.d_out9 = d_out9;
  k.need_grad = (grads != nullptr) && !predict;
  k.has_rows = row_end > row_begin;
  k.hl[0] = HostLatent{p->Mf, p->Zf, p->u_fm, p->u_fs_sqrt, p->ell_f, p->var_f};
  k.hl[1] = HostLatent{p->Mg, p->Zg, p->u_gm, p->u_gs_sqrt, p->ell_g, p->var_g};
  k.ell_h[0] = p->ell_f; k.ell_h[1] = p->ell_g;
  ZIGP_TRY(dense_mxm_forward(c, k));
  if (k.prep_side) ZIGP_HIP(c, hipStreamWaitEvent(c->stream_main, c->ev_prep, 0));
  else ZIGP_TRY(dense_prepare_buffers(c, k));
  ZIGP_TRY(dense_chunk_loop(c, k));
  if (predict) { ZIGP_HIP(c, hipStreamSynchronize(c->stream)); prof_collect(c); return info_result(c, k.hinfo, "Kuu"); }
  if (k.need_grad) {
    ProfScope ps(c, PC_MXM);     // wall time of the two concurrent chains, as in the forward: both events on the main stream, the second after the join
    TwoStream ts(c);
    ZIGP_TRY(ts.fork());
    for (int h = 0; h < 2; ++h) {
      if (h == 1) ts.second();
      ZIGP_TRY(latent_mxm_backward(c, c->lat[h], D, jitter, k.has_rows, include_kl != 0));
    }
    ZIGP_TRY(ts.join());
  }
  return dense_gather(c, k, elbo_data, kl, grads);
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

int zigp_create(zigp_ctx** out, int device_id) {
  if (!out) return ZIGP_EARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZIGP_EHIP;
  if (device_id < 0 || device_id >= ndev) return ZIGP_EARG;
  if (hipSetDevice(device_id) != hipSuccess) return ZIGP_EHIP;
  zigp_ctx* c = new (std::nothrow) zigp_ctx();
  if (!c) return ZIGP_EHIP;
  c->device = device_id;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ZIGP_EHIP; }
  c->stream_main = c->stream;
  if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { delete c; return ZIGP_EHIP; }
  if (hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess) { delete c; return ZIGP_EHIP; }
  if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_prep_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_prep, hipEventDisableTiming) != hipSuccess) { delete c; return ZIGP_EHIP; }
  if (hipMalloc((void**)&c->d_info, sizeof(int)) != hipSuccess) { delete c; return ZIGP_EHIP; }
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_potrf_diag), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(sizeof(double) * PB * PBLD)) != hipSuccess) { delete c; return ZIGP_EHIP; }
  if (const char* e = getenv("ZIGP_FWD_KUF_SIDE")) c->fwd_kuf_side = atoi(e) != 0;   // A/B switch (tools/ab_envs.sh); default on
  if (const char* e = getenv("ZIGP_TRMM_TAIL")) c->trmm_tail = atoi(e) != 0;      // A/B switch of the LPT tail (tools/ab_envs.sh); default on
  if (const char* e = getenv("ZIGP_COMM_TIMEOUT_S")) { const double v = atof(e); if (v > 0) c->comm_timeout_s = v; }
  *out = c;
  return ZIGP_OK;
}

int zigp_destroy(zigp_ctx* c) {
  if (!c) return ZIGP_EARG;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream_main);
  (void)hipStreamSynchronize(c->stream2);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->comm) { RcclApi* api = rccl_api(nullptr); if (api) (void)api->CommDestroy(static_cast<ncclComm_t>(c->comm)); c->comm = nullptr; }
  for (int h = 0; h < 2; ++h) {
    Latent& l = c->lat[h];
    DevBuf* bs[] = {&l.Z, &l.ell, &l.u, &l.s, &l.s2, &l.Kuu, &l.L, &l.W, &l.K, &l.A1, &l.Jp, &l.Wp, &l.Wt, &l.Wpt, &l.P, &l.Qt, &l.Rt, &l.a1gm, &l.part, &l.gm, &l.gv, &l.du, &l.dsq, &l.krow,
                    &l.dLpart, &l.T1, &l.T2, &l.T3, &l.G, &l.vec, &l.sk};
    for (DevBuf* b : bs) b->release();
  }
  DevBuf* bs[] = {&c->ownX, &c->ownY, &c->pw_part, &c->out9, &c->scratch, &c->scratch2, &c->packed, &c->parm, &c->selX, &c->selY, &c->selIdx};
  for (DevBuf* b : bs) b->release();
  if (c->kron && c->kron_free) c->kron_free(c->kron);
  if (c->kronf && c->kronf_free) c->kronf_free(c->kronf);
  for (auto& kv : c->tiles) if (kv.second.d) (void)hipFree(kv.second.d);
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->d_info) (void)hipFree(c->d_info);
  c->pinned.release();
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_prep_fork) (void)hipEventDestroy(c->ev_prep_fork);
  if (c->ev_prep) (void)hipEventDestroy(c->ev_prep);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream_main) (void)hipStreamDestroy(c->stream_main);
  delete c;
  return ZIGP_OK;
}

const char* zigp_last_error(zigp_ctx* c) { return c ? c->err.c_str() : "null context"; }
int zigp_last_info(zigp_ctx* c) { return c ? c->info : 0; }

int zigp_set_overlap(zigp_ctx* c, int32_t on) {
  if (!c) return ZIGP_EARG;
  if (on < 0 || on > 1) return fail_arg(c, "zigp_set_overlap: mode must be 0 or 1");
  c->overlap = on;
  return ZIGP_OK;
}

int zigp_set_chunk(zigp_ctx* c, int64_t chunk_rows) {
  if (!c) return ZIGP_EARG;
  if (chunk_rows == 0) { c->chunk_auto = true; c->chunk = 32768; return ZIGP_OK; }   // back to the automatic rule
  if (chunk_rows < 1024 || chunk_rows % 1024 != 0) return fail_arg(c, "chunk must be a positive multiple of 1024 (or 0: automatic)");
  if (chunk_rows > (1 << 20)) return fail_arg(c, "chunk must be <= 1048576 rows (32-bit staging offsets; 5 panels of 8*M*chunk bytes per latent)");
  c->chunk = chunk_rows;
  c->chunk_auto = false;
  return ZIGP_OK;
}

int zigp_set_pivot_rtol(zigp_ctx* c, double rtol) {
  if (!c) return ZIGP_EARG;
  if (!(rtol >= 0) || !std::isfinite(rtol)) return fail_arg(c, "zigp_set_pivot_rtol: rtol must be finite and >= 0");
  c->pivot_rtol = rtol;
  return ZIGP_OK;
}

int64_t zigp_get_chunk(zigp_ctx* c, int32_t M) {
  if (!c || M <= 0) return ZIGP_EARG;
  return auto_chunk_for(c->chunk_auto, c->chunk, round_up(M, BM));
}

int64_t zigp_get_chunk_rows(zigp_ctx* c, int32_t M, int64_t span) {
  if (!c || M <= 0 || span < 0) return ZIGP_EARG;
  return chunk_rows_for(c->chunk_auto, c->chunk, round_up(M, BM), span);
}

static_assert(MAXD == 8, "zigp_ctx::mean_a / mean_da hold MAXD entries");

int zigp_set_mean_function(zigp_ctx* c, const double* a, int32_t D, double b) {
  if (!c) return ZIGP_EARG;
  if (D < -1 || D > MAXD || (D > 0 && !a)) return fail_arg(c, "zigp_set_mean_function: need -1 <= D <= 8 and a[D]");
  if (D < 0) {   // Zero: no mean function, nothing to differentiate
    for (int d = 0; d < MAXD; ++d) c->mean_a[d] = 0.0;
    c->mean_b = 0.0; c->mean_on = false;
    return ZIGP_OK;
  }
  if (!std::isfinite(b)) return fail_arg(c, "zigp_set_mean_function: b must be finite");
  for (int d = 0; d < D; ++d)
    if (!std::isfinite(a[d])) return fail_arg(c, "zigp_set_mean_function: a must be finite");
  for (int d = 0; d < MAXD; ++d) c->mean_a[d] = (d < D) ? a[d] : 0.0;
  c->mean_b = b;
  c->mean_on = true;   // "enabled" is independent of the values: a Constant at exactly 0 still gets its gradient
  return ZIGP_OK;
}

int zigp_get_mean_function_grad(zigp_ctx* c, double* da, int32_t D, double* db) {
  if (!c) return ZIGP_EARG;
  if (D < 0 || D > MAXD || (D > 0 && !da)) return fail_arg(c, "zigp_get_mean_function_grad: need 0 <= D <= 8 and da[D]");
  for (int d = 0; d < D; ++d) da[d] = c->mean_da[d];
  if (db) *db = c->mean_db;
  return ZIGP_OK;
}

int zigp_set_data(zigp_ctx* c, const double* X, const double* Y, int64_t N, int32_t D) {
  if (!c) return ZIGP_EARG;
  if (!X || !Y || N <= 0 || D <= 0 || D > MAXD) return fail_arg(c, "zigp_set_data: bad arguments (need X, Y, N>0, 1<=D<=8)");
  ZIGP_HIP(c, hipSetDevice(c->device));
  ZIGP_ENSURE(c, c->ownX, (size_t)N * D);
  ZIGP_ENSURE(c, c->ownY, (size_t)N);
  ZIGP_HIP(c, hipMemcpyAsync(c->ownX.p, X, sizeof(double) * N * D, hipMemcpyHostToDevice, c->stream));
  ZIGP_HIP(c, hipMemcpyAsync(c->ownY.p, Y, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  c->dX = c->ownX.p; c->dY = c->ownY.p; c->N = N; c->D = D;
  c->fullX = c->dX; c->fullY = c->dY; c->fullN = N;
  return ZIGP_OK;
}

int zigp_set_data_device(zigp_ctx* c, const double* dX, const double* dY, int64_t N, int32_t D) {
  if (!c) return ZIGP_EARG;
  if (!dX || !dY || N <= 0 || D <= 0 || D > MAXD) return fail_arg(c, "zigp_set_data_device: bad arguments");
  c->dX = dX; c->dY = dY; c->N = N; c->D = D;
  c->fullX = dX; c->fullY = dY; c->fullN = N;
  return ZIGP_OK;
}

int zigp_select_rows(zigp_ctx* c, const int64_t* rows, int64_t n) {
  if (!c) return ZIGP_EARG;
  if (!c->fullX) return fail_arg(c, "zigp_select_rows: no data set (call zigp_set_data first)");
  if (n < 0 || (n > 0 && !rows)) return fail_arg(c, "zigp_select_rows: bad arguments");
  if (n == 0) { c->dX = c->fullX; c->dY = c->fullY; c->N = c->fullN; return ZIGP_OK; }   // back to the whole resident set
  for (int64_t i = 0; i < n; ++i)
    if (rows[i] < 0 || rows[i] >= c->fullN) return fail_arg(c, "zigp_select_rows: row index out of range");
  ZIGP_HIP(c, hipSetDevice(c->device));
  ZIGP_TRY(begin_staged_call(c));
  const int D = c->D;
  ZIGP_ENSURE(c, c->selX, (size_t)n * D); ZIGP_ENSURE(c, c->selY, (size_t)n); ZIGP_ENSURE(c, c->selIdx, (size_t)n);
  int64_t* h = (int64_t*)c->pinned.alloc(sizeof(int64_t) * n);
  if (!h) { c->err = "hipHostMalloc failed for the staging arena"; return ZIGP_EHIP; }
  memcpy(h, rows, sizeof(int64_t) * n);
  ZIGP_HIP(c, hipMemcpyAsync(c->selIdx.p, h, sizeof(int64_t) * n, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_gather_rows, dim3(ceil_div(n * (D + 1), 256)), dim3(256), 0, c->stream, c->fullX, c->fullY,
                     reinterpret_cast<const int64_t*>(c->selIdx.p), n, D, c->selX.p, c->selY.p);
  ZIGP_HIP(c, hipGetLastError());
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));   // the staged index block may be reused by the next call
  c->dX = c->selX.p; c->dY = c->selY.p; c->N = n;
  return ZIGP_OK;
}

int zigp_elbo(zigp_ctx* c, const zigp_params* p, double jitter, double scale, double g_offset, int64_t row_begin, int64_t row_end,
              int32_t include_kl, double* elbo_data, double* kl, zigp_grads* grads) {
  if (!c) return ZIGP_EARG;
  ZIGP_TRY(validate_params(c, p));
  if (!c->dX) return fail_arg(c, "zigp_elbo: no data set (call zigp_set_data first)");
  if (p->D != c->D) return fail_arg(c, "zigp_elbo: params.D differs from the data's D");
  if (row_begin < 0 || row_end > c->N || row_begin > row_end) return fail_arg(c, "zigp_elbo: bad row range");
  if (!(jitter >= 0)) return fail_arg(c, "zigp_elbo: jitter must be >= 0");
  ZIGP_HIP(c, hipSetDevice(c->device));
  return run_dense(c, p, c->dX, c->dY, c->N, c->D, jitter, scale, g_offset, row_begin, row_end, include_kl, false, nullptr, elbo_data, kl,
                   grads);
}

int zigp_predict(zigp_ctx* c, const zigp_params* p, const double* Xnew, int64_t N, double jitter, double g_offset, double* out9) {
  if (!c) return ZIGP_EARG;
  ZIGP_TRY(validate_params(c, p));
  if (N < 0 || (N > 0 && (!Xnew || !out9))) return fail_arg(c, "zigp_predict: bad arguments");
  if (N == 0) return ZIGP_OK;
  ZIGP_HIP(c, hipSetDevice(c->device));
  ZIGP_ENSURE(c, c->scratch, (size_t)N * p->D);
  ZIGP_ENSURE(c, c->out9, (size_t)9 * N);
  ZIGP_HIP(c, hipMemcpyAsync(c->scratch.p, Xnew, sizeof(double) * N * p->D, hipMemcpyHostToDevice, c->stream));
  ZIGP_TRY(run_dense(c, p, c->scratch.p, nullptr, N, p->D, jitter, 1.0, g_offset, 0, N, 0, true, c->out9.p, nullptr, nullptr, nullptr));
  ZIGP_HIP(c, hipMemcpyAsync(out9, c->out9.p, sizeof(double) * 9 * N, hipMemcpyDeviceToHost, c->stream));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  return ZIGP_OK;
}

int zigp_predict_device(zigp_ctx* c, const zigp_params* p, const double* dXnew, int64_t N, double jitter, double g_offset, double* d_out9) {
  if (!c) return ZIGP_EARG;
  ZIGP_TRY(validate_params(c, p));
  if (N < 0 || (N > 0 && (!dXnew || !d_out9))) return fail_arg(c, "zigp_predict_device: bad arguments");
  if (N == 0) return ZIGP_OK;
  ZIGP_HIP(c, hipSetDevice(c->device));
  hipPointerAttribute_t ax, ao;     // both must be device memory of this context's GPU (a host pointer here would fault inside a kernel)
  if (hipPointerGetAttributes(&ax, dXnew) != hipSuccess || hipPointerGetAttributes(&ao, d_out9) != hipSuccess ||
      ax.type != hipMemoryTypeDevice || ao.type != hipMemoryTypeDevice || ax.device != c->device || ao.device != c->device) {
    (void)hipGetLastError();
    return fail_arg(c, "zigp_predict_device: Xnew and out9 must be device memory of the context's GPU");
  }
  // rows in place: no copy in, no copy out; the call is complete on return (run_dense ends with the stream's synchronisation)
  return run_dense(c, p, dXnew, nullptr, N, p->D, jitter, 1.0, g_offset, 0, N, 0, true, d_out9, nullptr, nullptr, nullptr);
}

int zigp_prior_kl(zigp_ctx* c, const zigp_params* p, double jitter, double* kl2) {
  if (!c) return ZIGP_EARG;
  ZIGP_TRY(validate_params(c, p));
  if (!kl2) return fail_arg(c, "zigp_prior_kl: kl2 is NULL");
  ZIGP_HIP(c, hipSetDevice(c->device));
  HostLatent hl[2] = {{p->Mf, p->Zf, p->u_fm, p->u_fs_sqrt, p->ell_f, p->var_f}, {p->Mg, p->Zg, p->u_gm, p->u_gs_sqrt, p->ell_g, p->var_g}};
  ZIGP_TRY(begin_staged_call(c));
  ZIGP_HIP(c, hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream));
  ZIGP_TRY(latents_upload(c, hl, p->D));
  {
    TwoStream ts(c);
    ZIGP_TRY(ts.fork());
    ZIGP_TRY(latents_forward(c, hl, p->D, jitter, true, false));
    ZIGP_TRY(ts.join());
  }
  double klh[2] = {0.0, 0.0};
  for (int h = 0; h < 2; ++h)
    ZIGP_HIP(c, hipMemcpyAsync(&klh[h], c->lat[h].vec.p + 3 * c->lat[h].Mp, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  ZIGP_TRY(check_info(c, "Kuu"));          // synchronises; on failure kl2 is left untouched
  kl2[0] = klh[0]; kl2[1] = klh[1];
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  return ZIGP_OK;
}

int zigp_rbf_K(zigp_ctx* c, const double* X1, int64_t n1, const double* X2, int64_t n2, int32_t D, const double* ell, double var, double* K) {
  if (!c) return ZIGP_EARG;
  if (!X1 || n1 <= 0 || D <= 0 || D > MAXD || !ell || !K) return fail_arg(c, "zigp_rbf_K: bad arguments");
  if (!X2) n2 = n1;
  if (n2 <= 0) return fail_arg(c, "zigp_rbf_K: bad n2");
  ZIGP_HIP(c, hipSetDevice(c->device));
  ZIGP_ENSURE(c, c->scratch, (size_t)(n1 + n2) * D);
  ZIGP_ENSURE(c, c->scratch2, (size_t)n1 * n2);
  double* d1 = c->scratch.p; double* d2 = d1 + n1 * D;
  ZIGP_HIP(c, hipMemcpyAsync(d1, X1, sizeof(double) * n1 * D, hipMemcpyHostToDevice, c->stream));
  ZIGP_HIP(c, hipMemcpyAsync(d2, X2 ? X2 : X1, sizeof(double) * n2 * D, hipMemcpyHostToDevice, c->stream));
  KernHyp h = make_hyp(ell, var, D);
  hipLaunchKernelGGL(k_rbf_matrix, dim3(ceil_div(n1 * n2, 256)), dim3(256), 0, c->stream, d1, n1, d2, n2, h, 0.0, c->scratch2.p, n1, n2, n2);
  ZIGP_HIP(c, hipGetLastError());
  ZIGP_HIP(c, hipMemcpyAsync(K, c->scratch2.p, sizeof(double) * n1 * n2, hipMemcpyDeviceToHost, c->stream));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  return ZIGP_OK;
}

int zigp_profile_enable(zigp_ctx* c, int32_t on) { if (!c) return ZIGP_EARG; c->prof_on = on != 0; return ZIGP_OK; }
int zigp_profile_reset(zigp_ctx* c) {
  if (!c) return ZIGP_EARG;
  prof_collect(c);
  for (int i = 0; i < ZIGP_NCLASS; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; c->prof_flops[i] = 0; c->prof_total[i] = 0; }
  return ZIGP_OK;
}
int zigp_profile_get(zigp_ctx* c, double* ms, int64_t* launches, double* flops) {
  if (!c) return ZIGP_EARG;
  prof_collect(c);
  for (int i = 0; i < ZIGP_NCLASS; ++i) { if (ms) ms[i] = c->prof_ms[i]; if (launches) launches[i] = c->prof_n[i]; if (flops) flops[i] = c->prof_flops[i]; }
  return ZIGP_OK;
}
int zigp_profile_sampling(zigp_ctx* c, int32_t every) {
  if (!c) return ZIGP_EARG;
  if (every < 1) return fail_arg(c, "zigp_profile_sampling: every must be >= 1");
  c->prof_every = every;
  return ZIGP_OK;
}
int zigp_profile_totals(zigp_ctx* c, int64_t* total_launches) {
  if (!c || !total_launches) return ZIGP_EARG;
  for (int i = 0; i < ZIGP_NCLASS; ++i) total_launches[i] = c->prof_total[i];
  return ZIGP_OK;
}

// ---- sustained shader clock (bench.py) ------------------------------------------------------------
// One workgroup per XCD (workgroups of a dispatch go round-robin over the 8 XCDs): lane 0 records its XCC id, the shader-clock counter
// (s_memtime) and the constant 100 MHz counter (s_memrealtime).  Two calls bracket a measured region; the host pairs the stamps by XCC id.
__global__ void k_clock_stamp(long long* __restrict__ out) {
  if (threadIdx.x != 0) return;
  const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xf;     // HW_REG_XCC_ID, bits [3:0]
  const long long cyc = (long long)__builtin_readcyclecounter();
  const long long rt = (long long)__builtin_amdgcn_s_memrealtime();
  out[3 * blockIdx.x + 0] = (long long)xcc;
  out[3 * blockIdx.x + 1] = cyc;
  out[3 * blockIdx.x + 2] = rt;
}
int zigp_clock_stamp(zigp_ctx* c, int64_t* out) {
  if (!c || !out) return ZIGP_EARG;
  static_assert(sizeof(long long) == sizeof(int64_t), "stamp layout");
  ZIGP_HIP(c, hipSetDevice(c->device));
  ZIGP_ENSURE(c, c->scratch2, 64);
  hipLaunchKernelGGL(k_clock_stamp, dim3(8), dim3(64), 0, c->stream_main, reinterpret_cast<long long*>(c->scratch2.p));
  ZIGP_HIP(c, hipGetLastError());
  ZIGP_HIP(c, hipMemcpyAsync(out, c->scratch2.p, sizeof(int64_t) * 24, hipMemcpyDeviceToHost, c->stream_main));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream_main));
  return ZIGP_OK;
}

// ---- diagnostics -------------------------------------------------------------------------------
int zigp_test_kuf(zigp_ctx* c, int64_t N, int32_t M, int32_t D, const double* X, const double* Z, const double* ell, double var, double* K) {
  if (!c) return ZIGP_EARG;
  if (N <= 0 || M <= 0 || D < 1 || D > MAXD || !X || !Z || !ell || !K) return fail_arg(c, "zigp_test_kuf: bad arguments");
  ZIGP_HIP(c, hipSetDevice(c->device));
  const int64_t Nc = round_up(N, 1024);
  const int Mp = (int)round_up(M, 16);
  const KufHyp kh = make_kuf_hyp(ell, var, D);
  std::vector<double> zs((size_t)Mp * D, 0.0), hk((size_t)Mp * Nc);
  for (int m = 0; m < M; ++m)
    for (int d = 0; d < D; ++d) zs[(size_t)m * D + d] = Z[(size_t)m * D + d] * kh.scale[d];
  DevBuf dx, dz, dk;
  auto body = [&]() -> int {
    ZIGP_ENSURE(c, dx, (size_t)N * D); ZIGP_ENSURE(c, dz, zs.size()); ZIGP_ENSURE(c, dk, hk.size());
    ZIGP_HIP(c, hipMemcpyAsync(dx.p, X, sizeof(double) * N * D, hipMemcpyHostToDevice, c->stream));
    ZIGP_HIP(c, hipMemcpyAsync(dz.p, zs.data(), sizeof(double) * zs.size(), hipMemcpyHostToDevice, c->stream));
    const dim3 grid((unsigned)(Nc / 512), Mp / 16), block(256);
#define ZIGP_KUF(DD) \
  case DD: hipLaunchKernelGGL(k_kuf_build<DD>, grid, block, 0, c->stream, dx.p, N, (int64_t)0, dz.p, M, kh, dk.p, Nc); break;
    switch (D) { ZIGP_KUF(1) ZIGP_KUF(2) ZIGP_KUF(3) ZIGP_KUF(4) ZIGP_KUF(5) ZIGP_KUF(6) ZIGP_KUF(7) ZIGP_KUF(8) }
#undef ZIGP_KUF
    ZIGP_HIP(c, hipGetLastError());
    ZIGP_HIP(c, hipMemcpyAsync(hk.data(), dk.p, sizeof(double) * hk.size(), hipMemcpyDeviceToHost, c->stream));
    ZIGP_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
  };
  const int rc = body();
  dx.release(); dz.release(); dk.release();
  if (rc) return rc;
  for (int m = 0; m < M; ++m) memcpy(K + (size_t)m * N, &hk[(size_t)m * Nc], sizeof(double) * N);
  return ZIGP_OK;
}

int zigp_test_trmm_list(int32_t lower, int32_t Mf, int32_t Mg, int64_t Nc, int32_t tail_on, int64_t* out) {
  // host only (no context, no GPU): the lists chunk_forward would launch for a chunk of Nc rows, checked tile by tile
  if (Mf <= 0 || Mg <= 0 || Nc <= 0 || Nc % BN != 0 || !out) return ZIGP_EARG;
  const int Mp[2] = {(int)round_up(Mf, BM), (int)round_up(Mg, BM)}, nbn = (int)(Nc / BN), kb = BM / BK;
  TrmmTail tail;
  const bool paired = chunk_trmm_plan(Mp[0], Mp[1], nbn, tail_on != 0, tail);
  int64_t wgs[2], per[2], worst_tail = 0;
  for (int h = 0; h < 2; ++h) {
    const int nbm = Mp[h] / BM;
    std::vector<GemmTile> v;
    per[h] = build_trmm_list(lower != 0, nbm, nbn, paired, tail.units[h], tail.bins[h], v);
    if (v.size() % (size_t)per[h]) return -10;
    wgs[h] = (int64_t)(v.size() / per[h]);
    std::vector<int> seen((size_t)nbm * nbn, 0);
    for (const GemmTile& t : v) {
      if (t.kend <= t.kbeg) continue;                                   // padding
      if (t.bi < 0 || t.bi >= nbm || t.bj < 0 || t.bj >= nbn) return -11;
      const int k0 = lower ? 0 : t.bi * kb, k1 = lower ? (t.bi + 1) * kb : nbm * kb;
      if (t.kbeg != k0 || t.kend != k1 || (t.kdir != 1 && t.kdir != -1) || t.slice != 0) return -12;   // the whole k range of its row block, nothing else
      seen[(size_t)t.bi * nbn + t.bj] += 1;
    }
    for (int q : seen) if (q != 1) return -13;                          // every tile exactly once
    if (paired && tail.units[h] > 0) {                                  // workgroups behind the regular units: the LPT tail
      const int64_t regular = wgs[h] - 8 * (int64_t)std::min(64, tail.bins[h]);
      for (int64_t w = std::max<int64_t>(regular, 0); w < wgs[h]; ++w) {
        int64_t load = 0;
        for (int e = 0; e < per[h]; ++e) { const GemmTile& t = v[(size_t)w * per[h] + e]; load += std::max(0, t.kend - t.kbeg) / kb; }
        worst_tail = std::max(worst_tail, load);
      }
    }
  }
  out[0] = wgs[0]; out[1] = wgs[1]; out[2] = per[0]; out[3] = per[1]; out[4] = tail.units[0]; out[5] = tail.units[1]; out[6] = worst_tail; out[7] = paired ? 1 : 0;
  return ZIGP_OK;
}

int zigp_test_gemm(zigp_ctx* c, int32_t transA, int32_t transB, int64_t m, int64_t n, int64_t k, const double* A, const double* B, double* C) {
  if (!c) return ZIGP_EARG;
  if (m <= 0 || n <= 0 || k <= 0 || !A || !B || !C) return fail_arg(c, "zigp_test_gemm: bad arguments");
  ZIGP_HIP(c, hipSetDevice(c->device));
  const int64_t mp = round_up(m, BM), np = round_up(n, BN), kp = round_up(k, BM);
  // stored shapes: A is (m,k) or (k,m) if transA; B is (k,n) or (n,k) if transB
  const int64_t ar = transA ? kp : mp, ac = transA ? mp : kp, br = transB ? np : kp, bc = transB ? kp : np;
  std::vector<double> ha((size_t)ar * ac, 0.0), hb((size_t)br * bc, 0.0), hc((size_t)mp * np);
  const int64_t ar0 = transA ? k : m, ac0 = transA ? m : k, br0 = transB ? n : k, bc0 = transB ? k : n;
  for (int64_t i = 0; i < ar0; ++i) memcpy(&ha[i * ac], &A[i * ac0], sizeof(double) * ac0);
  for (int64_t i = 0; i < br0; ++i) memcpy(&hb[i * bc], &B[i * bc0], sizeof(double) * bc0);
  DevBuf da, db, dc;
  int rc = 0;
  auto body = [&]() -> int {
    ZIGP_ENSURE(c, da, ha.size()); ZIGP_ENSURE(c, db, hb.size()); ZIGP_ENSURE(c, dc, hc.size());
    ZIGP_HIP(c, hipMemcpyAsync(da.p, ha.data(), sizeof(double) * ha.size(), hipMemcpyHostToDevice, c->stream));
    ZIGP_HIP(c, hipMemcpyAsync(db.p, hb.data(), sizeof(double) * hb.size(), hipMemcpyHostToDevice, c->stream));
    TileList tl;
    ZIGP_TRY(tiles_full(c, (int)(mp / BM), (int)(np / BN), (int)(kp / BK), tl));
    GemmArgs g = mk_args(da.p, ac, db.p, bc, dc.p, np);
    if (!transA && !transB) ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_MNCONTIG, false>(c, tl, g, EpiStore())));
    if (transA && !transB) ZIGP_TRY((run_gemm<LAY_MNCONTIG, LAY_MNCONTIG, false>(c, tl, g, EpiStore())));
    if (!transA && transB) ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_KCONTIG, false>(c, tl, g, EpiStore())));
    if (transA && transB) ZIGP_TRY((run_gemm<LAY_MNCONTIG, LAY_KCONTIG, false>(c, tl, g, EpiStore())));
    ZIGP_HIP(c, hipMemcpyAsync(hc.data(), dc.p, sizeof(double) * hc.size(), hipMemcpyDeviceToHost, c->stream));
    ZIGP_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
  };
  rc = body();
  da.release(); db.release(); dc.release();
  if (rc) return rc;
  for (int64_t i = 0; i < m; ++i) memcpy(&C[i * n], &hc[i * np], sizeof(double) * n);
  return ZIGP_OK;
}

int zigp_test_potrf_trtri(zigp_ctx* c, int64_t n, const double* A, double* L, double* W, int32_t split_k) {
  if (!c) return ZIGP_EARG;
  if (n <= 0 || !A) return fail_arg(c, "zigp_test_potrf_trtri: bad arguments");
  ZIGP_HIP(c, hipSetDevice(c->device));
  const int Mp = (int)round_up(n, BM);
  std::vector<double> ha((size_t)Mp * Mp, 0.0);
  for (int64_t i = 0; i < Mp; ++i) {
    if (i < n) memcpy(&ha[i * Mp], &A[i * n], sizeof(double) * n);
    else ha[i * Mp + i] = 1.0;
  }
  DevBuf dl, dw, dt, dplanes;
  auto body = [&]() -> int {
    ZIGP_ENSURE(c, dl, ha.size()); ZIGP_ENSURE(c, dw, ha.size()); ZIGP_ENSURE(c, dt, ha.size());
    ZIGP_HIP(c, hipMemcpyAsync(dl.p, ha.data(), sizeof(double) * ha.size(), hipMemcpyHostToDevice, c->stream));
    ZIGP_HIP(c, hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream));
    if (split_k) {     // the chain as the dense M x M forward runs it: every block product cut into k slices (run_gemm_sk_tiles)
      const PotrfJob job = {dl.p, dw.p, dt.p, Mp, true, (int)n, 0.0, false, &dplanes};
      const hipStream_t st = c->stream;
      ZIGP_TRY(potrf_trtri_jobs(c, 1, &job, &st));
    } else ZIGP_TRY(potrf_trtri(c, dl.p, dw.p, dt.p, Mp, true, (int)n));
    ZIGP_TRY(check_info(c, "A"));
    std::vector<double> ho(ha.size());
    if (L) {
      ZIGP_HIP(c, hipMemcpyAsync(ho.data(), dl.p, sizeof(double) * ho.size(), hipMemcpyDeviceToHost, c->stream));
      ZIGP_HIP(c, hipStreamSynchronize(c->stream));
      for (int64_t i = 0; i < n; ++i) memcpy(&L[i * n], &ho[i * Mp], sizeof(double) * n);
    }
    if (W) {
      ZIGP_HIP(c, hipMemcpyAsync(ho.data(), dw.p, sizeof(double) * ho.size(), hipMemcpyDeviceToHost, c->stream));
      ZIGP_HIP(c, hipStreamSynchronize(c->stream));
      for (int64_t i = 0; i < n; ++i) memcpy(&W[i * n], &ho[i * Mp], sizeof(double) * n);
    }
    return 0;
  };
  int rc = body();
  dl.release(); dw.release(); dt.release(); dplanes.release();
  return rc;
}

}  // extern "C"
