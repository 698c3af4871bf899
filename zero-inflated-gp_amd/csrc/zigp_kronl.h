// Larger Kronecker grids on the fused path: capacity <NB0, NB1> 16-row blocks per factor with NB0 + NB1 <= 8 (the reference's
// [10, 100] grid, scripts/onoff.py:52-53, is <1, 7>).  Same wave-per-16-points design as the small-grid kernels of zigp_kronf.hip,
// but a wave cannot keep the 72 accumulator blocks of that grid in registers, so the backward kernel SPILLS its eight per-tile
// operands (K_p, E_p, A_p, t_p) to a point-major image in global memory (4 KB per point; L2-resident for a minibatch) and
// k_kfl_accum forms the sums over points, one wave per 16 x 16 output block and point split.
// Included inside namespace zigp by zigp_kronf.hip.

// LDS layout of the fragment images: P0 | P1 | Al | S2 | AlT | S2T, packed by the ACTUAL block counts
// STAGE = false: a wave that owns only a tile or two reads the fragments straight from global memory (L2): staging would move the
// same bytes once per workgroup and put a barrier in front of the first MFMA
// The staged copy is ASYNCHRONOUS (kf_stage_frag_async): the caller computes the K tiles of its first tile, which need the inducing
// inputs only, and then calls kf_stage_wait() -- in a minibatch step a wave owns one tile, and the 131 / 160 KB copy (8-10 k cycles)
// used to sit in front of its 10 k cycles of K-tile arithmetic instead of under it.
template <bool STAGE>
__device__ __forceinline__ KfFrags kfl_stage_frags(double* lds, const KfLat& L, bool with_transposes) {
  if (!STAGE) { KfFrags G = {L.f[0].PF, L.f[1].PF, L.AlF, L.S2F, L.AlTF, L.S2TF, L.f[0].Zs, L.f[1].Zs}; return G; }
  const int nb0 = L.f[0].nb, nb1 = L.f[1].nb;
  const int n0 = nb0 * nb0 * 256, n1 = nb1 * nb1 * 256, n01 = nb0 * nb1 * 256;
  double* p = lds;
  KfFrags F;
  F.P0 = p; kf_stage_frag_async(p, L.f[0].PF, n0); p += n0;
  F.P1 = p; kf_stage_frag_async(p, L.f[1].PF, n1); p += n1;
  F.Al = p; kf_stage_frag_async(p, L.AlF, n01); p += n01;
  F.S2 = p; kf_stage_frag_async(p, L.S2F, n01); p += n01;
  F.AlT = nullptr; F.S2T = nullptr;
  F.Z0 = L.f[0].Zs; F.Z1 = L.f[1].Zs;      // global (L1-resident): the LDS of these kernels is full of fragment images
  if (with_transposes) {
    F.AlT = p; kf_stage_frag_async(p, L.AlTF, n01); p += n01;
    F.S2T = p; kf_stage_frag_async(p, L.S2TF, n01);
  }
  return F;
}

template <int NB0, int NB1, bool STAGE, bool EXACT>
__global__ void __launch_bounds__(64 * KF_WAVES, 1)
k_kfl_forward(KfArgs a) {
  extern __shared__ double lds[];
  const KfLat& L = a.lat[a.lat0 + blockIdx.y];
  const KfFrags F = kfl_stage_frags<STAGE>(lds, L, false);
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15, slot = lane;      // slot: this lane's position in a fragment block
  const int w = blockIdx.x * KF_WAVES + (threadIdx.x >> 6);
  const int t1 = min((w + 1) * a.tpw, a.ntiles);
  KfTile<NB0, NB1> t;
  bool first = true;
  if (w * a.tpw < t1) {          // K tiles of the first tile under the asynchronous staging copy
    const int64_t pn = (int64_t)(w * a.tpw) * 16 + n;
    kf_forward_ktiles<NB0, NB1, EXACT>(t, L, F.Z0, F.Z1, a.X + (pn < a.N ? pn : 0) * a.ldx, pn < a.N, g);
  }
  if (STAGE) kf_stage_wait();
  for (int tile = w * a.tpw; tile < t1; ++tile) {
    const int64_t pn = (int64_t)tile * 16 + n;
    const bool valid = pn < a.N;
    if (!first) kf_forward_ktiles<NB0, NB1, EXACT>(t, L, F.Z0, F.Z1, a.X + (valid ? pn : 0) * a.ldx, valid, g);
    first = false;
    kf_forward_products<NB0, NB1, EXACT>(t, L, F, slot);
    double q0 = 0.0, q1 = 0.0, mu = 0.0, st = 0.0;
#pragma unroll
    for (int q = 0; q < 4 * NB0; ++q) {
      q0 = fma(t.K0[q], t.A0[q], q0);
      mu = fma(t.K0[q], t.B0[q], mu);
      st = fma(t.A0[q] * t.A0[q], t.C0[q], st);
    }
#pragma unroll
    for (int q = 0; q < 4 * NB1; ++q) q1 = fma(t.K1[q], t.A1[q], q1);
    q0 = kf_colsum(q0); q1 = kf_colsum(q1); mu = kf_colsum(mu); st = kf_colsum(st);
    if (g == 0) { L.part[pn] = q0; L.part[a.Npad + pn] = q1; L.part[2 * a.Npad + pn] = mu; L.part[3 * a.Npad + pn] = st; }
  }
}

// Spill image of one tile quantity V (rows x 16 points), point-major with the rows of every 16-block permuted so that both operand
// roles of k_kfl_accum read it coalesced:  element (row = 16 rb + 4 r + a, point n)  ->  base[n * Mq + 16 rb + 4 a + r]
//   A role (lane (a, kk), fragments r = 0..3): ONE 32-byte load at [(4 ks + kk) * Mq + 16 rb + 4 a]
//   B role (lane (kk, j)):                     [(4 ks + kk) * Mq + 16 cb + 4 (j % 4) + j / 4]
template <int NB>
__device__ __forceinline__ void kfl_spill(double* __restrict__ base, const double (&V)[4 * NB], int nb, int Mq, int g, int n) {
#pragma unroll
  for (int rb = 0; rb < NB; ++rb)
    if (rb < nb) *reinterpret_cast<double4*>(base + n * Mq + 16 * rb + 4 * g) = make_double4(V[4 * rb], V[4 * rb + 1], V[4 * rb + 2], V[4 * rb + 3]);
}

// backward for the larger grids: the per-point reverse pass of k_kf_backward, operands of the sums over points spilled per tile:
// record = K0 | E0 | A0 | t0 (16 Mq0 doubles each) | K1 | E1 | A1 | t1 (16 Mq1 each)
template <int NB0, int NB1, bool STAGE, bool EXACT>
__global__ void __launch_bounds__(64 * KF_WAVES, 1)
k_kfl_backward(KfArgs a) {
  extern __shared__ double lds[];
  const KfLat& L = a.lat[a.lat0 + blockIdx.y];
  const KfFac &f0 = L.f[0], &f1 = L.f[1];
  const KfFrags F = kfl_stage_frags<STAGE>(lds, L, true);
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15, slot = lane;      // slot: this lane's position in a fragment block
  const int w = blockIdx.x * KF_WAVES + (threadIdx.x >> 6);
  const int nb0 = EXACT ? NB0 : f0.nb, nb1 = EXACT ? NB1 : f1.nb;
  const int Mq0 = 16 * nb0, Mq1 = 16 * nb1;
  const int64_t rec = 64 * (int64_t)(Mq0 + Mq1);
  const int tb = a.tile0 + w * a.tpw;                    // this launch covers tiles [tile0, tile1); its records are numbered from tile0
  const int t1 = min(tb + a.tpw, a.tile1);
  KfTile<NB0, NB1> t;
  bool first = true;
  if (tb < t1) {          // K tiles of the first tile under the asynchronous staging copy
    const int64_t pn = (int64_t)tb * 16 + n;
    kf_forward_ktiles<NB0, NB1, EXACT>(t, L, F.Z0, F.Z1, a.X + (pn < a.N ? pn : 0) * a.ldx, pn < a.N, g);
  }
  if (STAGE) kf_stage_wait();
  for (int tile = tb; tile < t1; ++tile) {
    const int64_t pn = (int64_t)tile * 16 + n;
    const bool valid = pn < a.N;
    const double* xrow = a.X + (valid ? pn : 0) * a.ldx;
    if (!first) kf_forward_ktiles<NB0, NB1, EXACT>(t, L, F.Z0, F.Z1, xrow, valid, g);
    first = false;
    kf_forward_products<NB0, NB1, EXACT>(t, L, F, slot);
    const double gmn = L.gm[pn], gvn = L.gv[pn], dq0n = L.dq0[pn], dq1n = L.dq1[pn];
    double* R = L.spill + (int64_t)(tile - a.tile0) * rec;
    double* R1 = R + 64 * Mq0;
    kfl_spill<NB0>(R, t.K0, nb0, Mq0, g, n);
    kfl_spill<NB1>(R1, t.K1, nb1, Mq1, g, n);
    kfl_spill<NB0>(R + 32 * Mq0, t.A0, nb0, Mq0, g, n);
    kfl_spill<NB1>(R1 + 32 * Mq1, t.A1, nb1, Mq1, g, n);
    {   // factor 0: B0, C0 are part of the forward tile
      double dA[4 * NB0], PdA[4 * NB0], E[4 * NB0];
#pragma unroll
      for (int q = 0; q < 4 * NB0; ++q) { dA[q] = 2.0 * gvn * t.A0[q] * t.C0[q]; PdA[q] = 0.0; E[q] = fma(dq0n, t.K0[q], dA[q]); }
      kfl_spill<NB0>(R + 16 * Mq0, E, nb0, Mq0, g, n);
      kf_frag_mm<NB0, 4 * NB0>(PdA, F.P0, nb0, 4 * nb0, dA, slot);
#pragma unroll
      for (int q = 0; q < 4 * NB0; ++q) E[q] = fma(gmn, t.B0[q], fma(2.0 * dq0n, t.A0[q], PdA[q])) * t.K0[q];
      kfl_spill<NB0>(R + 48 * Mq0, E, nb0, Mq0, g, n);
    }
    {   // factor 1: B1 = Alpha^T K0, C1 = S2^T A0^2
      double B1[4 * NB1], C1[4 * NB1], sq[4 * NB0];
#pragma unroll
      for (int q = 0; q < 4 * NB1; ++q) { B1[q] = 0.0; C1[q] = 0.0; }
#pragma unroll
      for (int q = 0; q < 4 * NB0; ++q) sq[q] = t.A0[q] * t.A0[q];
      kf_frag_mm<NB1, 4 * NB0>(B1, F.AlT, nb1, 4 * nb0, t.K0, slot);
      kf_frag_mm<NB1, 4 * NB0>(C1, F.S2T, nb1, 4 * nb0, sq, slot);
      double PdA[4 * NB1];
#pragma unroll
      for (int q = 0; q < 4 * NB1; ++q) { C1[q] = 2.0 * gvn * t.A1[q] * C1[q]; PdA[q] = 0.0; }       // C1 <- dA1
      kf_frag_mm<NB1, 4 * NB1>(PdA, F.P1, nb1, 4 * nb1, C1, slot);
#pragma unroll
      for (int q = 0; q < 4 * NB1; ++q) {
        PdA[q] = fma(gmn, B1[q], fma(2.0 * dq1n, t.A1[q], PdA[q])) * t.K1[q];                          // PdA <- t1
        C1[q] = fma(dq1n, t.K1[q], C1[q]);                                                             // C1 <- E1
      }
      kfl_spill<NB1>(R1 + 16 * Mq1, C1, nb1, Mq1, g, n);
      kfl_spill<NB1>(R1 + 48 * Mq1, PdA, nb1, Mq1, g, n);
    }
  }
}

struct KflAccLat { const double* spill; const double *gm, *gv; double* acc; int nb0, nb1, D0, D1; const double *hyp0, *hyp1; };   // hyp_p: factor records of the hyperparameter block (zc)
struct KflAccArgs { KflAccLat lat[2]; const double* X; int64_t N; int ldx, ntiles, tps, nb0c, nb1c; int tile0, tile1, part0; };   // tile range of this launch, its first part

// sums over points for the larger grids: wave = one 16 x 16 output block over the tiles of one split
//   dAlpha += K0 diag(gm) K1^T ; dS2 += A0^2 diag(gv) (A1^2)^T ; dP_p += E_p K_p^T ; moments_p += t_p Psi_p
__global__ void __launch_bounds__(256)
k_kfl_accum(KflAccArgs a) {
  const KflAccLat& L = a.lat[blockIdx.z];
  const int lane = threadIdx.x & 63, kk = lane >> 4, bj = lane & 15;
  const int nblk = kf_nblocks(a.nb0c, a.nb1c);
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= nblk) return;
  const KfBlock b = kf_block_decode(blk, a.nb0c, a.nb1c);
  const int Mq0 = 16 * L.nb0, Mq1 = 16 * L.nb1;
  const int64_t rec = 64 * (int64_t)(Mq0 + Mq1);
  // operand images inside a tile record, and the block counts that bound (rb, cb)
  int offA, MqA, offB, MqB, nbr, nbc;
  switch (b.kind) {
    case 0: offA = 0; MqA = Mq0; offB = 64 * Mq0; MqB = Mq1; nbr = L.nb0; nbc = L.nb1; break;                         // K0 , K1 gm
    case 1: offA = 32 * Mq0; MqA = Mq0; offB = 64 * Mq0 + 32 * Mq1; MqB = Mq1; nbr = L.nb0; nbc = L.nb1; break;       // A0^2 , A1^2 gv
    case 2: offA = 16 * Mq0; MqA = Mq0; offB = 0; MqB = Mq0; nbr = L.nb0; nbc = L.nb0; break;                         // E0 , K0
    case 3: offA = 64 * Mq0 + 16 * Mq1; MqA = Mq1; offB = 64 * Mq0; MqB = Mq1; nbr = L.nb1; nbc = L.nb1; break;       // E1 , K1
    case 4: offA = 48 * Mq0; MqA = Mq0; offB = 0; MqB = 0; nbr = L.nb0; nbc = 1; break;                               // t0 , Psi_0
    default: offA = 64 * Mq0 + 48 * Mq1; MqA = Mq1; offB = 0; MqB = 0; nbr = L.nb1; nbc = 1; break;                   // t1 , Psi_1
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (b.rb < nbr && b.cb < nbc) {
    const int D = b.kind == 4 ? L.D0 : L.D1, col0 = b.kind == 4 ? 0 : L.D0;
    const auto zc = KF_CONST(b.kind == 4 ? L.hyp0 : L.hyp1) + KH_ZC;
    const int psi_d = (bj == 0) ? -1 : ((bj <= D) ? bj - 1 : ((bj <= 2 * D) ? bj - 1 - D : -2));   // -1: constant 1, -2: 0
    const int t0 = a.tile0 + blockIdx.y * a.tps, t1 = min(t0 + a.tps, a.tile1);
    const int oa = 16 * b.rb + 4 * (bj & 3) + (bj >> 2), ob = 16 * b.cb + 4 * (bj & 3) + (bj >> 2);     // both roles: row / column l % 16 of the block (permuted image, kfl_spill)
#pragma unroll 2
    for (int tile = t0; tile < t1; ++tile) {
      const double* R = L.spill + (int64_t)(tile - a.tile0) * rec;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int pt = 4 * ks + kk;
        const int64_t pn = (int64_t)tile * 16 + pt;
        double av = R[offA + pt * MqA + oa];          // A operand of v_mfma_f64_16x16x4: (row l % 16, point 4 ks + l / 16)
        double bv;
        if (b.kind <= 3) {
          bv = R[offB + pt * MqB + ob];
          if (b.kind == 0) bv *= L.gm[pn];
          if (b.kind == 1) { bv = bv * bv * L.gv[pn]; av *= av; }
        } else {
          bv = 0.0;
          if (pn < a.N) {
            if (psi_d == -1) bv = 1.0;
            else if (psi_d >= 0) { const double xv = a.X[pn * a.ldx + col0 + psi_d] - zc[psi_d]; bv = (bj <= D) ? xv : xv * xv; }
          }
        }
        kf_mfma16(acc, av, bv);
      }
    }
  }
  double* out = L.acc + ((int64_t)(a.part0 + blockIdx.y) * nblk + blk) * 256;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[r * 64 + lane] = acc[r];
}

// ---- M x M stages with operands in global memory (L2 resident): the LDS-resident kernels of zigp_kronf.hip hold 32 x 32 matrices ----
// C (m x n) [+]= op(A) (m x k) op(B) (k x n); m, k multiples of 4.  A thread owns a 4 x 1 micro-tile (rows i0..i0+3, column j) and
// unrolls k by 4: 20 independent loads per step keep the memory pipe busy (the naive loop is one L2 round trip per FMA).
template <bool TA, bool TB, bool ACC>
__device__ __forceinline__ void kf_gmm(double* C, int ldc, const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, int m, int n,
                                       int k) {
  const int mt = m >> 2;
  for (int idx = threadIdx.x; idx < mt * n; idx += blockDim.x) {
    const int it = idx / n, j = idx - it * n, i0 = 4 * it;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q0 = 0; q0 < k; q0 += 4) {
      double av[4][4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u;
        bv[u] = TB ? B[j * ldb + q] : B[q * ldb + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r][u] = TA ? A[q * lda + i0 + r] : A[(i0 + r) * lda + q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fma(av[r][u], bv[u], v[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { double* c = C + (i0 + r) * ldc + j; *c = ACC ? *c + v[r] : v[r]; }
  }
}

// C (16 nbr x 16 ncb) = A . B on the MFMA pipe: A in fragment order (k = 4 ksn; P_p ships that way for the point kernels), B in
// global memory -- row-major B[k][j], or (TB) the transpose of a row-major matrix, B[k][j] = Bsrc[j][k].  The waves of the workgroup
// stride over the 16 x 16 output blocks; TC stores the result transposed (C^T row-major).  One 32-byte and one 8-byte load per
// 16 x 16 x 4 product, independent across k-steps: a 112^3 product takes microseconds where the scalar loop took a hundred.
template <bool TB, bool TC, int CB>
__device__ __forceinline__ void kf_frag_gmm_cb(double* __restrict__ C, int ldc, const double* __restrict__ AF, int nbr, int ksn,
                                               const double* __restrict__ B, int ldb, int ncb) {
  // a wave owns a block ROW and up to CB column blocks of it: the A fragment of a k-step is loaded once and feeds CB independent
  // accumulator chains, and 1 + CB independent loads per k-step are in flight (a one-block loop is a chain of L2 round trips)
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int ncg = (ncb + CB - 1) / CB;
  for (int unit = wave; unit < nbr * ncg; unit += nw) {
    const int rb = unit / ncg, cb0 = (unit - rb * ncg) * CB;
    double acc[CB][4];
#pragma unroll
    for (int c = 0; c < CB; ++c) { acc[c][0] = 0.0; acc[c][1] = 0.0; acc[c][2] = 0.0; acc[c][3] = 0.0; }
    // k-steps in batches of 4 (ksn is a multiple of 4): the 4 fragment loads and 4 CB operand loads of a batch are issued together,
    // then its 16 CB MFMAs -- one memory round trip per batch instead of one per k-step
    for (int k0 = 0; k0 < ksn; k0 += 4) {
      double av[4];
      double bv[4][CB];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ks = k0 + u;
        av[u] = AF[(rb * ksn + ks) * 64 + lane];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          const int cb = min(cb0 + c, ncb - 1);      // clamped: the surplus chains of the last group are computed and dropped
          bv[u][c] = TB ? B[(16 * cb + n) * ldb + 4 * ks + g] : B[(4 * ks + g) * ldb + 16 * cb + n];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < CB; ++c) kf_mfma16(acc[c], av[u], bv[u][c]);
    }
#pragma unroll
    for (int c = 0; c < CB; ++c)
      if (cb0 + c < ncb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * rb + 4 * r + g, col = 16 * (cb0 + c) + n;
          if (TC) C[col * ldc + row] = acc[c][r]; else C[row * ldc + col] = acc[c][r];
        }
      }
  }
}

template <bool TB, bool TC>
__device__ __forceinline__ void kf_frag_gmm(double* __restrict__ C, int ldc, const double* __restrict__ AF, int nbr, int ksn,
                                            const double* __restrict__ B, int ldb, int ncb) {
  if (ncb >= 3) kf_frag_gmm_cb<TB, TC, 4>(C, ldc, AF, nbr, ksn, B, ldb, ncb);
  else kf_frag_gmm_cb<TB, TC, 1>(C, ldc, AF, nbr, ksn, B, ldb, ncb);
}

// The latent stage of the larger grids (same job structure as k_kf_latent).  U, S2, T0, T1 and Alpha (16 x <= 112 each) live in LDS
// while the kernel works on them and go out to global memory on the side (the reverse pass reads them): the first version kept them
// in global memory only, and every phase began with an L2 round trip on what the phase before had just written -- 22.6 us for a few
// hundred MFMAs.  P_p comes as fragment images from global memory as before (P1 is 100 KB at 100 points, read once).
constexpr int KFL_LAT_ROWS = 16;      // Mq0 of the larger-grid plan (kf_plan: m0 <= 16)
__host__ __device__ inline size_t kfl_latent_lds(int Mq1) { return sizeof(double) * 5 * KFL_LAT_ROWS * (size_t)(Mq1 + 1); }
__global__ void __launch_bounds__(1024)
k_kfl_latent(KfLatentArgs a) {
  extern __shared__ double sm[];
  __shared__ double sh[16];
  const KfLatentJob& jb = a.job[blockIdx.x];
  const int t = threadIdx.x, M0 = jb.M0, M1 = jb.M1, Mq0 = jb.Mq0, Mq1 = jb.Mq1;
  const int LD = Mq1 + 1, SZ = KFL_LAT_ROWS * LD;
  double *sU = sm, *sS2 = sm + SZ, *sT0 = sm + 2 * SZ, *sT1 = sm + 3 * SZ, *sAl = sm + 4 * SZ;
  for (int idx = t; idx < Mq0 * Mq1; idx += 1024) {
    const int i = idx / Mq1, j = idx - i * Mq1;
    const bool in = i < M0 && j < M1;
    const double sv = in ? jb.s[i * M1 + j] : 0.0, uv = in ? jb.u[i * M1 + j] : 0.0;
    sU[i * LD + j] = uv; sS2[i * LD + j] = sv * sv;
    jb.U[idx] = uv; jb.S2[idx] = sv * sv;
  }
  __syncthreads();
  kf_frag_gmm<true, true>(sT0, LD, jb.PF1, Mq1 / 16, Mq1 / 4, sU, LD, Mq0 / 16);     // T0^T = P1 U^T  (P1 symmetric)
  kf_frag_gmm<false, false>(sT1, LD, jb.PF0, Mq0 / 16, Mq0 / 4, sU, LD, Mq1 / 16);   // T1 = P0 U
  __syncthreads();
  kf_frag_gmm<false, false>(sAl, LD, jb.PF0, Mq0 / 16, Mq0 / 4, sT0, LD, Mq1 / 16);  // Alpha = P0 (U P1)
  for (int idx = t; idx < Mq0 * Mq1; idx += 1024) {                                  // (T0, T1 are complete: out they go)
    const int i = idx / Mq1, j = idx - i * Mq1;
    jb.T0[idx] = sT0[i * LD + j]; jb.T1[idx] = sT1[i * LD + j];
  }
  __syncthreads();
  for (int idx = t; idx < Mq0 * Mq1; idx += 1024) { const int i = idx / Mq1, j = idx - i * Mq1; jb.Al[idx] = sAl[i * LD + j]; }
  kf_write_frag(jb.AlF, Mq0 / 16, Mq1 / 4, t, 1024, sAl, LD, false);
  kf_write_frag(jb.S2F, Mq0 / 16, Mq1 / 4, t, 1024, sS2, LD, false);
  kf_write_frag(jb.AlTF, Mq1 / 16, Mq0 / 4, t, 1024, sAl, LD, true);
  kf_write_frag(jb.S2TF, Mq1 / 16, Mq0 / 4, t, 1024, sS2, LD, true);
  double av = 0.0, bv = 0.0, cv = 0.0;
  for (int idx = t; idx < M0 * M1; idx += 1024) {
    const int i = idx / M1, j = idx - i * M1;
    const double s2 = sS2[i * LD + j];
    av = fma(sU[i * LD + j], sAl[i * LD + j], av);
    bv += log(s2);
    cv = fma(jb.dvec0[i] * jb.dvec1[j], s2, cv);
  }
  // the three scalars in one pass: wave sums, then 16 partials each in fixed order
  __shared__ double sh3[3][16];
  const double vals[3] = {av, bv, cv};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const double v = wave_sum(vals[q]);
    if ((t & 63) == 0) sh3[q][t >> 6] = v;
  }
  __syncthreads();
  if (t < 3) { double r = 0.0; for (int w = 0; w < 16; ++w) r += sh3[t][w]; jb.klv[t] = r; }
  if (t == 0) { jb.klv[3] = jb.dvec0[Mq0]; jb.klv[4] = jb.dvec1[Mq1]; }
  (void)sh;
}

// C (m x n) [+]= op(A) op(B) for SMALL outputs with a long k: one output element per wave, the lanes split k (coalesced reads of the
// contiguous operand rows), fixed-order wave sum
template <bool TA, bool TB, bool ACC>
__device__ __forceinline__ void kf_gmm_wave(double* C, int ldc, const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, int m,
                                            int n, int k) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int idx = wave; idx < m * n; idx += nw) {
    const int i = idx / n, j = idx - i * n;
    double v = 0.0;
    for (int q = lane; q < k; q += 64) v = fma(TA ? A[q * lda + i] : A[i * lda + q], TB ? B[j * ldb + q] : B[q * ldb + j], v);
    v = wave_sum(v);
    if (lane == 0) { double* c = C + i * ldc + j; *c = ACC ? *c + v : v; }
  }
}

// global-operand twin of k_kf_finish: grid (2, latents); work matrices have leading dimension ldw, scratch behind them
struct KflFinishArgs { KfFinishJob job[2]; double jitter; int with_kl; int ldw, wS2, wP0, wP1, wK0, wK1; int64_t scratch_off, scratch_set; };

// Crows (16 x n) = A (16 x k) . B (k x n), B row-major in global memory (coalesced over the columns), A through a lambda (LDS or a
// broadcast global read); a thread owns 4 rows x 1 column, k unrolled by 4 (k is a multiple of 16)
template <class AF, class ST>
__device__ __forceinline__ void kf_rowblock_mm(int n, int k, AF af, const double* __restrict__ B, int ldb, ST st) {
  // the B loads are the latency of this loop (two waves per SIMD, nothing else to switch to): 16 of them are in flight while the
  // previous 16 are consumed
  for (int idx = threadIdx.x; idx < 4 * n; idx += blockDim.x) {
    const int rq = idx / n, j = idx - rq * n;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    double bv[16], bn[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) bv[u] = B[u * ldb + j];
    for (int q0 = 0; q0 < k; q0 += 16) {
      const bool more = q0 + 16 < k;
      if (more) {
#pragma unroll
        for (int u = 0; u < 16; ++u) bn[u] = B[(q0 + 16 + u) * ldb + j];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fma(af(4 * rq + r, q0 + u), bv[u], v[r]);
      if (more) {
#pragma unroll
        for (int u = 0; u < 16; ++u) bv[u] = bn[u];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) st(4 * rq + r, j, v[r]);
  }
}

// C (16 x 16) = A (16 x k, LDS, row stride xs) . B (k x 16 through a lambda); 512 threads: (output, k parity), pair sum by one shuffle
template <class BF, class ST>
__device__ __forceinline__ void kf_small_mm(int k, const double* A, int xs, BF bf, ST st) {
  const int t = threadIdx.x, half = t & 1, out = t >> 1, i = out >> 4, j = out & 15;
  double v = 0.0;
#pragma unroll 8
  for (int q = half; q < k; q += 2) v = fma(A[i * xs + q], bf(q, j), v);
  v += __shfl_xor(v, 1);
  if (!half) st(i, j, v);
}

// The M x M reverse pass of the larger grids, spread over the chip: grid (row blocks, 2 factors, latents), three launches (STAGE 1..3)
// with the global dependencies between them.  A single workgroup doing the 112^3 products of factor 1 is bound by ONE compute unit's
// L1 / L2 pipe (in-kernel stamps: 328 k cycles = 137 us, none of it arithmetic); a workgroup per 16-row block of the outputs needs its
// own rows of the left operand only, so the three stages are embarrassingly parallel over the row blocks:
//   1  dP_p += (data-term products), Q_p (KL), w_p ;  factor 0: X0 = dAl P1 by column blocks
//   2  factor 0: dU = P0 X0 -> gu, gs by column blocks ;  X rows = sym(dP) - kl (1/4 (Q + Q^T) + 1/2 diag w) -> Q2 rows = X rows . P
//   3  G rows = P rows . Q2 ;  Kuu-style reductions of the rows against K_p -> krow (+ the re-centred data moments)
constexpr int KFL_FIN_THREADS = 512;     // kf_small_mm: 256 outputs x 2 k parities
template <int STAGE>
__global__ void __launch_bounds__(KFL_FIN_THREADS)
k_kfl_finish(KflFinishArgs a) {
  __shared__ double Xs[(STAGE == 1 ? 3 : 1) * 16 * (16 * 8 + 1)];      // 16 rows of X or P (row stride Mq + 1); stage 1: dAl, T0, U
  const KfFinishJob& jb = a.job[blockIdx.z];
  const int p = blockIdx.y, rbk = blockIdx.x;
  const int t = threadIdx.x, M0 = jb.M0, M1 = jb.M1, Mq0 = jb.Mq0, Mq1 = jb.Mq1, ldw = a.ldw;
  const bool kl = a.with_kl != 0;
  const int M = p == 0 ? M0 : M1, Mq = p == 0 ? Mq0 : Mq1, Mo = p == 0 ? M1 : M0, D = p == 0 ? jb.D0 : jb.D1;
  const int nb = Mq / 16, nb1 = Mq1 / 16;
  const double* Z = p == 0 ? jb.Z0 : jb.Z1;
  const auto zc = KF_CONST(p == 0 ? jb.hyp0 : jb.hyp1) + KH_ZC;
  const double* P = p == 0 ? jb.P0 : jb.P1;
  const double* Kuu = p == 0 ? jb.K0 : jb.K1;
  double* work = const_cast<double*>(jb.work);
  const double* dAl = work;
  double* dP = work + (p == 0 ? a.wP0 : a.wP1);
  const double* Kr = work + (p == 0 ? a.wK0 : a.wK1);
  double* krow = p == 0 ? jb.krow0 : jb.krow1;
  double* Xb = work + a.scratch_off + (int64_t)p * a.scratch_set;     // per factor: Xb, G, Q (ldw x ldw each), dU, w
  double* G = Xb + (int64_t)ldw * ldw; double* Q = G + (int64_t)ldw * ldw; double* dU = Q + (int64_t)ldw * ldw;
  double* wv = Xb + a.scratch_set - ldw;           // w_p: the tail of the factor's scratch set
  const int R = 16 * rbk;
  const int xs = Mq + 1, xs1 = Mq1 + 1;
  if (STAGE == 1) {
    if (p == 1) {
      if (rbk >= nb) return;
      kf_rowblock_mm(Mq1, Mq0, [&](int r, int q) { return jb.T1[q * Mq1 + R + r]; }, dAl, ldw,
                     [&](int r, int j, double v) { dP[(R + r) * ldw + j] += v; });                                    // dP1 += T1^T dAl
      if (kl) {
        kf_rowblock_mm(Mq1, Mq0, [&](int r, int q) { return jb.U[q * Mq1 + R + r]; }, jb.T1, Mq1,
                       [&](int r, int j, double v) { Q[(R + r) * ldw + j] = v; });                                    // Q1 = U^T T1
        if (rbk == 0 && t < Mq1) { double w = 0.0; if (t < M1) for (int o = 0; o < M0; ++o) w = fma(jb.dvec0[o], jb.S2[o * Mq1 + t], w); wv[t] = w; }
      }
    } else {
      if (rbk >= nb1) return;
      // factor 0 has few rows (M0 <= 16): its 16 x 16 products over k = M1 run from LDS copies of the 16-row operands, a thread per
      // (output, k parity); the M0 x M1 product X0 = dAl P1 is spread over the column blocks of the grid
      double* As = Xs; double* Ts = Xs + 16 * xs1; double* Us = Ts + 16 * xs1;
      for (int idx = t; idx < 16 * Mq1; idx += KFL_FIN_THREADS) {
        const int r = idx / Mq1, c = idx - r * Mq1;
        const bool in = r < Mq0;
        As[r * xs1 + c] = in ? dAl[r * ldw + c] : 0.0;
        if (rbk == 0) { Ts[r * xs1 + c] = in ? jb.T0[r * Mq1 + c] : 0.0; if (kl) Us[r * xs1 + c] = in ? jb.U[r * Mq1 + c] : 0.0; }
      }
      __syncthreads();
      kf_small_mm(Mq1, As, xs1, [&](int q, int j) { return jb.P1[q * Mq1 + R + j]; },
                  [&](int i, int j, double v) { if (i < Mq0) Xb[i * ldw + R + j] = v; });                              // X0 = dAl P1
      if (rbk == 0) {
        kf_small_mm(Mq1, As, xs1, [&](int q, int j) { return Ts[j * xs1 + q]; },
                    [&](int i, int j, double v) { if (i < Mq0 && j < Mq0) dP[i * ldw + j] += v; });                    // dP0 += dAl T0^T
        if (kl) {
          kf_small_mm(Mq1, Ts, xs1, [&](int q, int j) { return Us[j * xs1 + q]; },
                      [&](int i, int j, double v) { if (i < Mq0 && j < Mq0) Q[i * ldw + j] = v; });                    // Q0 = T0 U^T
          {   // w0 = S2 dvec1: 32 lanes per row
            const int r = t >> 5, l = t & 31;
            double w = 0.0;
            if (r < M0) for (int o = l; o < M1; o += 32) w = fma(jb.dvec1[o], jb.S2[r * Mq1 + o], w);
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) w += __shfl_xor(w, sft);
            if (l == 0 && r < Mq0) wv[r] = w;
          }
        }
      }
    }
    return;
  }
  if (STAGE == 2) {
    if (p == 0 && rbk < nb1) {
      // dU = P0 X0, column block rbk -> gu, gs of those columns
      kf_rowblock_mm(16, Mq0, [&](int r, int q) { return r < Mq0 ? jb.P0[r * Mq0 + q] : 0.0; }, Xb + R, ldw,
                     [&](int r, int j, double v) { if (r < Mq0) dU[r * ldw + R + j] = v; });
      __syncthreads();
      for (int idx = t; idx < M0 * 16; idx += KFL_FIN_THREADS) {
        const int i = idx / 16, j = R + (idx & 15);
        if (j < M1) {
          const double sv = jb.s[i * M1 + j];
          double gu = dU[i * ldw + j], gs = 2.0 * sv * work[a.wS2 + i * ldw + j];
          if (kl) { gu -= jb.Al[i * Mq1 + j]; gs -= (-1.0 / sv + jb.dvec0[i] * jb.dvec1[j] * sv); }
          jb.gu[i * M1 + j] = gu; jb.gs[i * M1 + j] = gs;
        }
      }
    }
    if (rbk >= nb) return;
    // X rows of this block in LDS (the transposed reads dP[c][R + r] are 128-byte row segments), then Q2 rows = X rows . P
    for (int idx = t; idx < 16 * Mq; idx += KFL_FIN_THREADS) {
      const int c = idx / 16, r = idx & 15;
      double v = 0.5 * (dP[(R + r) * ldw + c] + dP[c * ldw + R + r]);
      if (kl) { v -= 0.25 * (Q[(R + r) * ldw + c] + Q[c * ldw + R + r]); if (c == R + r) v -= 0.5 * wv[c]; }
      Xs[r * xs + c] = v;
    }
    __syncthreads();
    if (p == 0) {
      // the factor of <= 16 points (the spatial one: the factor the reference's initialisation makes ill-conditioned, scripts/onoff.py:57-66)
      // takes the compensated sandwich of the small grids (kf_dd_mac, zigp_kronf.hip): Q2 = X P as (hi, lo), lo in the unused rows 16..31
      // of this factor's Q region.  The <= 112-point factor stays in float64 (cond(K_t) = 1.5 at the reference's lengthscale).
      for (int idx = t; idx < Mq * Mq; idx += KFL_FIN_THREADS) {
        const int i = idx / Mq, j = idx - i * Mq;
        double sh = 0.0, sl = 0.0;
        for (int q = 0; q < Mq; ++q) kf_dd_mac(sh, sl, Xs[i * xs + q], P[q * Mq + j]);
        const double h = sh + sl;
        G[i * ldw + j] = h;
        Q[(16 + i) * ldw + j] = sl - (h - sh);
      }
      return;
    }
    kf_rowblock_mm(Mq, Mq, [&](int r, int q) { return Xs[r * xs + q]; }, P, Mq, [&](int r, int j, double v) { G[(R + r) * ldw + j] = v; });   // Q2 -> G region
    return;
  }
  // ---- STAGE 3
  if (rbk >= nb) return;
  for (int idx = t; idx < 16 * Mq; idx += KFL_FIN_THREADS) { const int r = idx / Mq, c = idx - r * Mq; Xs[r * xs + c] = P[(R + r) * Mq + c]; }
  __syncthreads();
  double* Gm = Q;     // the KL product is dead: G rows go there
  const double coef = kl ? 0.5 * (double)Mo : 0.0;
  if (p == 0) {       // compensated: Gm = -(P (Q2_hi + Q2_lo) + coef P), rounded once
    const double* Ql = Q + 16 * ldw;
    for (int idx = t; idx < Mq * Mq; idx += KFL_FIN_THREADS) {
      const int i = idx / Mq, j = idx - i * Mq;
      double sh = 0.0, sl = 0.0;
      for (int q = 0; q < Mq; ++q) kf_dd_mac2(sh, sl, Xs[i * xs + q], G[q * ldw + j], Ql[q * ldw + j]);
      kf_dd_mac(sh, sl, coef, Xs[i * xs + j]);
      Gm[i * ldw + j] = -(sh + sl);
    }
  } else {
    kf_rowblock_mm(Mq, Mq, [&](int r, int q) { return Xs[r * xs + q]; }, G, ldw, [&](int r, int j, double v) { Gm[(R + r) * ldw + j] = v; });     // G = P Q2
  }
  __syncthreads();
  // krow[m][c]: Kuu part (as k_kuu_grad, Kz = K_p - jitter I) + data moments rebuilt around z_m.  A wave owns row m, its lanes sweep the
  // columns j (coalesced rows of G, P, K_p), fixed-order wave sums
  const int W = 2 + 2 * D;
  const int lane = t & 63;
  for (int m = R + (t >> 6); m < min(R + 16, M); m += KFL_FIN_THREADS / 64) {
    double s0 = 0.0, s1[MAXD], s2[MAXD], zm[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) { s1[d] = 0.0; s2[d] = 0.0; zm[d] = d < D ? Z[m * D + d] : 0.0; }
    for (int j = lane; j < M; j += 64) {
      const double kz = Kuu[m * PB + j] - ((m == j) ? a.jitter : 0.0);
      const double tt = (p == 0 ? Gm[m * ldw + j] : -Gm[m * ldw + j] - coef * P[m * Mq + j]) * kz;
      s0 += tt;
#pragma unroll
      for (int d = 0; d < MAXD; ++d)
        if (d < D) { const double df = Z[j * D + d] - zm[d]; s1[d] = fma(2.0 * tt, df, s1[d]); s2[d] = fma(tt * df, df, s2[d]); }
    }
    s0 = wave_sum(s0);
    const double k0 = Kr[m * 16];
    if (lane == 0) krow[m * W] = s0 + k0;
    for (int d = 0; d < D; ++d) {
      const double a1 = wave_sum(s1[d]), a2 = wave_sum(s2[d]);
      if (lane == 0) {
        const double dz = zm[d] - zc[d], k1 = Kr[m * 16 + 1 + d];
        krow[m * W + 1 + d] = a1 + k1 - dz * k0;
        krow[m * W + 1 + D + d] = a2 + Kr[m * 16 + 1 + D + d] - 2.0 * dz * k1 + dz * dz * k0;
      }
    }
    if (lane == 0) krow[m * W + 1 + 2 * D] = 0.0;
  }
}
