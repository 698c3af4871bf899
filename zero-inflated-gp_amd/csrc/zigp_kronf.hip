// Fused Kronecker (space x time) path for small inducing grids (every factor <= 16 * KF_NBMAX points): scripts/onoff.py:143-319,
// onofftf/main.py:350-387, onofftf/onoffpred.py:127-200.  Same factored algebra as zigp_kron.hip (see its header), but nothing
// of size O(M N) ever reaches HBM: a WAVE owns a tile of 16 points and keeps every per-point vector in registers in the
// operand layout of v_mfma_f64_16x16x4 (round 4; rounds 2-3: four v_mfma_f64_4x4x4 per product), whose B operand and D result share one lane layout
//     V[q] of lane l  =  V(row 4 q + l / 16, point l % 16)
// so the result of one product feeds the next one with no shuffle or LDS round trip.  The M_p x M_p / M_0 x M_1 matrices
// (P_p = K_p^-1, Alpha, S2 and the transposes) are read from global memory (L1 / L2 resident: 8 KB each at 32 x 32) in
// A-FRAGMENT ORDER, one coalesced 32-byte load per lane and 16 x 16 x 4 product.  Launches per step:
//   k_kf_factor   one workgroup per factor: K_p + jitter, Cholesky + triangular inverse in LDS, P_p = W^T W, logdet, diag(P_p)
//   k_kf_latent   one workgroup per latent: Alpha = P0 U P1, T0 = U P1, T1 = P0 U, KL scalars, fragment images
//   k_kf_forward  q0, q1, mean, S-term per point (part[4][N])                      (kron_inf value, scripts/onoff.py:186-213)
//   k_kron_pointwise / k_kron_head_pointwise (zigp_kron.hip)                       (probit moments, likelihood, reverse pass)
//   k_kf_backward recomputes the forward tile, hand-derived reverse pass; the sums over points (dAlpha, dS2, dP_p and the
//                 kernel-cotangent moments) accumulate in MFMA accumulators across the wave's tiles -> one partial per wave
//   k_kf_reduce   fixed-order sum of the per-wave partials (+ one workgroup for the sums of the point-wise block partials;
//                 value-only steps launch k_kron_pw_reduce for those instead)
//   k_kf_finish   one workgroup per latent: the M x M reverse pass (dU, dP_p -> dK_p -> dZ, dell, dvar; KL gradient)
// plus ONE staged host->device copy (parameters + minibatch) and ONE device->host copy (results).
#include "zigp_host.h"

namespace zigp {

constexpr int KF_NBMAX = 2;       // 16-row blocks per factor handled by the register-resident kernels (M_p <= 32)
constexpr int KF_LD = 18;         // LDS row stride (doubles) of a [rows][16 points] tile: A- and B-fragment reads are conflict free
constexpr int KF_WAVES = 4;       // waves per workgroup (each wave works alone on its own tiles)

struct KfFac {
  int M, nb, D, col0;
  const double* hyp;        // this factor's record of the device hyperparameter block (KH_INV, KH_ZC, KH_VAR; zigp_kernels.h): 1 / ell, the centre
                            // zc of the moment sums (mid-range of Z_p: sum t (x - z)^k is rebuilt from sum t (x - zc)^k), the variance
  const double* Zs;         // [16 nb][D] inducing inputs divided by the lengthscales, zero rows beyond M (written by k_kf_factor)
  const double* PF;         // P_p in A-fragment order [nb][4 nb][16][4]
};
struct KfLat {
  KfFac f[2];
  const double *AlF, *S2F;      // [nb0][4 nb1][16][4]   rows = factor-0 index, k = factor-1 index
  const double *AlTF, *S2TF;    // [nb1][4 nb0][16][4]   the transposes
  double* part;                 // [4][Npad]: q0, q1, mean, S-term
  const double *gm, *gv, *dq0, *dq1;   // [Npad] cotangents from the point-wise kernel
  double* acc;                  // [waves][KF_ACC_BLOCKS][4][64] per-wave partial sums
  double* spill;                // larger grids: per-tile operand records of the sums over points (zigp_kronl.h)
};
struct KfArgs {
  KfLat lat[2];
  const double* X; int64_t N, Npad; int ldx;
  int tpw;        // tiles per wave
  int lat0;       // first latent of this launch (latents with different block counts are launched separately)
  int ntiles;     // Npad / 16
  int tile0, tile1;   // k_kfl_backward: the tile range of this launch (its spill records are numbered from tile0)
};

// accumulator blocks of the backward kernel (16 x 16 each), compile-time layout for KF_NBMAX
constexpr int KF_B_AL = 0;
constexpr int KF_B_S2 = KF_B_AL + KF_NBMAX * KF_NBMAX;
constexpr int KF_B_P0 = KF_B_S2 + KF_NBMAX * KF_NBMAX;
constexpr int KF_B_P1 = KF_B_P0 + KF_NBMAX * KF_NBMAX;
constexpr int KF_B_K0 = KF_B_P1 + KF_NBMAX * KF_NBMAX;
constexpr int KF_B_K1 = KF_B_K0 + KF_NBMAX;
constexpr int KF_ACC_BLOCKS = KF_B_K1 + KF_NBMAX;
constexpr int KF_ACC_DOUBLES = KF_ACC_BLOCKS * 256;

__device__ __forceinline__ double kf_mfma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
// one 16 x 16 x 4 product: c[0..3] (rows 4 r + l / 16 of a 16-row block, point / column l % 16) += A(row l % 16, k l / 16) . B(k l / 16, column l % 16).
// Since round 4 the point kernels issue this form (zigp_gemm.h has the story): one A value per lane and product instead of four.
__device__ __forceinline__ void kf_mfma16(double* c, double a, double b) {
  mfma_d4 v = {c[0], c[1], c[2], c[3]};
  v = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, v, 0, 0, 0);
  c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
}

// order LDS traffic of ONE wave: the stores above must be visible to (and not sink below) the loads that follow
__device__ __forceinline__ void kf_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// out[4 rb + r] += sum_ks F(rb, ks)[r] * in[ks]:   out (16 nba rows) = A (16 nba x 4 ksn) . in (4 ksn rows), A in fragment order.
// k-steps go in chunks of four, double-buffered by hand: the four fragment loads of chunk c + 1 are issued, a compiler barrier pins
// them there, then come the 16 MFMAs of chunk c.  Left alone the compiler either waits for every load right before its MFMAs (runtime
// bounds: one LDS round trip per k-step) or hoists all loads of the tile (compile-time bounds: hundreds of live registers, spills).
template <int NBA, int QK>
__device__ __forceinline__ void kf_frag_mm(double (&out)[4 * NBA], const double* __restrict__ F, int nba, int ksn, const double (&in)[QK], int lane) {
  static_assert(QK % 4 == 0, "k-steps come in multiples of four (16-row blocks)");
#pragma unroll
  for (int rb = 0; rb < NBA; ++rb) {
    if (rb < nba) {
      const double* __restrict__ Fr = F + rb * ksn * 64 + lane;      // fragment order: one value per lane and (row block, k-step)
      double cur[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) cur[u] = Fr[u * 64];
#pragma unroll
      for (int kc = 0; kc < QK / 4; ++kc) {
        if (4 * kc < ksn) {
          double nxt[4];
          if (4 * (kc + 1) < ksn && kc + 1 < QK / 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) nxt[u] = Fr[(4 * (kc + 1) + u) * 64];
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) nxt[u] = cur[u];
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int u = 0; u < 4; ++u) kf_mfma16(&out[4 * rb], cur[u], in[4 * kc + u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
        }
      }
    }
  }
}

// exp(y) of B values at once, y <= 0.  These are the steps of the device library's exp (same constants, same operations in the same
// order: the results are bit-identical to exp()), written ACROSS the values: sixteen calls of exp() in a row compile to sixteen
// chains of ~16 dependent fp64 operations, one after the other (the kernels around this are at their register limit and the scheduler
// orders for pressure).  Measured with in-kernel stamps on the two K tiles of a 32 x 32 grid: 8.0 k -> 7.3 k cycles from this alone
// (a dependent v_fma_f64 issues every ~5 cycles, so the chains cost less than they look); the branches below were the larger part.
template <int B>
__device__ __forceinline__ void kf_exp_neg(double (&y)[B]) {
  const double LOG2E = __longlong_as_double(0x3ff71547652b82feLL);
  const double NLN2_HI = __longlong_as_double(0xbfe62e42fefa39efLL), NLN2_LO = __longlong_as_double(0xbc7abc9e3b39803fLL);
  const double C11 = __longlong_as_double(0x3e5ade156a5dcb37LL), C10 = __longlong_as_double(0x3e928af3fca7ab0cLL);
  const double CJ[8] = {__longlong_as_double(0x3ec71dee623fde64LL), __longlong_as_double(0x3efa01997c89e6b0LL), __longlong_as_double(0x3f2a01a014761f6eLL),
                        __longlong_as_double(0x3f56c16c1852b7b0LL), __longlong_as_double(0x3f81111111122322LL), __longlong_as_double(0x3fa55555555502a1LL),
                        __longlong_as_double(0x3fc5555555555511LL), __longlong_as_double(0x3fe000000000000bLL)};
  double k[B], r[B], p[B];
#pragma unroll
  for (int i = 0; i < B; ++i) k[i] = __builtin_rint(y[i] * LOG2E);
#pragma unroll
  for (int i = 0; i < B; ++i) r[i] = fma(NLN2_HI, k[i], y[i]);
#pragma unroll
  for (int i = 0; i < B; ++i) r[i] = fma(NLN2_LO, k[i], r[i]);
#pragma unroll
  for (int i = 0; i < B; ++i) p[i] = fma(C11, r[i], C10);
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < B; ++i) p[i] = fma(r[i], p[i], CJ[j]);
#pragma unroll
  for (int i = 0; i < B; ++i) p[i] = fma(r[i], p[i], 1.0);
#pragma unroll
  for (int i = 0; i < B; ++i) p[i] = fma(r[i], p[i], 1.0);
#pragma unroll
  for (int i = 0; i < B; ++i) {
    const double e = ldexp(p[i], (int)k[i]);
    y[i] = !(-1075.0 > y[i]) ? e : 0.0;      // below: 2^k is out of range of the conversion; exp underflows long before
  }
}

// -0.5 |z_m / ell - x / ell|^2 for the B rows m = 4 (q0 + i) + g of a batch; DD: the input dimension at compile time, 0: runtime f.D
template <int DD, int B>
__device__ __forceinline__ void kf_r2_batch(double (&y)[B], const KfFac& f, const double* zs, const double (&xs)[MAXD], int q0, int g, int mlast) {
  const int D = DD ? DD : f.D;
#pragma unroll
  for (int i = 0; i < B; ++i) {
    const int m = min(4 * (q0 + i) + g, mlast);
    double r2 = 0.0;
#pragma unroll
    for (int d = 0; d < (DD ? DD : MAXD); ++d)
      if (d < D) { const double t = zs[m * D + d] - xs[d]; r2 = fma(t, t, r2); }
    y[i] = -0.5 * r2;
  }
}
// K_p tile of this wave: K[q] = k_p(z_{4q+g}, x_n) for the lane's point n, 0 for padding rows / points   (kern.K(Z_p, xnew), :199-201).
// zs = Z_p / ell (LDS copy in the small-grid kernels).  The per-lane conditions (padding row, padding point) are SELECTS, not
// branches: with a branch around each row the 16 loads of z and the 16 exp calls of a tile ran one after the other, each behind
// its own memory round trip (in-kernel stamps: 21 k cycles for a forward tile whose MFMAs take 4.4 k).
// The input dimension of a factor is a kernel argument (wave-uniform).  With `if (d < f.D)` around every term the compiler emits a
// scalar branch per (row, dimension) and waits for each LDS read of z behind its own branch -- 140 branches and 32 serialized LDS
// round trips per pair of K tiles, more than half of the 8.0 k cycles the stamps showed.  SPEC: ONE switch per batch of 8 rows picks a
// straight-line body for D = 1, 2, 3 (the reference's factors are space (2) x time (1), scripts/onoff.py:52-53); larger D loops.
// The kernels with runtime block counts keep the loop form (!SPEC): they are at the register limit and the switch made them spill.
template <int Q, bool SPEC>
__device__ __forceinline__ void kf_ktile(double (&K)[Q], const KfFac& f, int nb, const double* zs, const double* __restrict__ xrow, bool valid, int g) {
  double xs[MAXD];
#pragma unroll
  for (int d = 0; d < MAXD; ++d) xs[d] = 0.0;
  const auto inv_ell = KF_CONST(f.hyp) + KH_INV;
  const double var = KF_CONST(f.hyp)[KH_VAR];
  if (SPEC && f.D == 1) xs[0] = xrow[f.col0] * inv_ell[0];
  else if (SPEC && f.D == 2) { xs[0] = xrow[f.col0] * inv_ell[0]; xs[1] = xrow[f.col0 + 1] * inv_ell[1]; }
  else {
#pragma unroll
    for (int d = 0; d < MAXD; ++d) xs[d] = (d < f.D) ? xrow[f.col0 + d] * inv_ell[d] : 0.0;
  }
  constexpr int B = Q < 8 ? Q : 8;
  const int mlast = 16 * nb - 1;             // zs holds 16 nb rows: blocks beyond nb (runtime block counts) read its last row and are zeroed below
#pragma unroll
  for (int q0 = 0; q0 < Q; q0 += B) {
    double y[B];
    if (SPEC) {
      switch (f.D) {
        case 1: kf_r2_batch<1, B>(y, f, zs, xs, q0, g, mlast); break;
        case 2: kf_r2_batch<2, B>(y, f, zs, xs, q0, g, mlast); break;
        case 3: kf_r2_batch<3, B>(y, f, zs, xs, q0, g, mlast); break;
        default: kf_r2_batch<0, B>(y, f, zs, xs, q0, g, mlast);
      }
    } else {
      kf_r2_batch<0, B>(y, f, zs, xs, q0, g, mlast);
    }
    kf_exp_neg<B>(y);
#pragma unroll
    for (int i = 0; i < B; ++i)
      if (q0 + i < Q) {
        const int m = 4 * (q0 + i) + g;
        K[q0 + i] = (m < f.M && m <= mlast && valid) ? var * y[i] : 0.0;
      }
  }
}

// sum over the 4 row groups (lanes l, l^16, l^32, l^48): every lane ends with the column total, fixed order
__device__ __forceinline__ double kf_colsum(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

constexpr int KF_Q = 4 * KF_NBMAX;   // registers per [16 nb][16] tile quantity
constexpr int KF_MQ_ = 16 * KF_NBMAX;

// The workgroup's copy of the fragment images in LDS (8 KB each at 32 x 32): every wave re-reads them for every tile, and an LDS
// read returns in ~100 cycles where an L2 hit takes 500+ (one wave per SIMD has nothing else to hide that behind).
constexpr int KF_FRAG = KF_MQ_ * KF_MQ_;
struct KfFrags { const double *P0, *P1, *Al, *S2, *AlT, *S2T; const double *Z0, *Z1; };   // + the scaled inducing inputs
__device__ __forceinline__ void kf_stage_z(double* dst, const KfFac& f) {
  for (int idx = threadIdx.x; idx < 16 * f.nb * f.D; idx += blockDim.x) dst[idx] = f.Zs[idx];
}
// n is a multiple of 256.  32 bytes per lane and load, eight loads in flight per thread: the larger grids stage 130-160 KB per
// workgroup (P1 alone is 100 KB at 100 points), and with 16-byte loads four at a time the copy was a chain of ~20 L2 round trips in
// front of a workgroup's only tile (minibatch steps: one tile per wave).
__device__ __forceinline__ void kf_stage_frag(double* dst, const double* __restrict__ src, int n) {
  const double4* __restrict__ s4 = reinterpret_cast<const double4*>(src);
  double4* d4 = reinterpret_cast<double4*>(dst);
  const int n4 = n / 4, step = blockDim.x;
  int idx = threadIdx.x;
  for (; idx + 7 * step < n4; idx += 8 * step) {
    double4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s4[idx + u * step];
#pragma unroll
    for (int u = 0; u < 8; ++u) d4[idx + u * step] = v[u];
  }
  for (; idx < n4; idx += step) d4[idx] = s4[idx];
}

// The same copy without registers: global_load_lds moves 1 KB per wave instruction straight into LDS and only counts on vmcnt, so a
// kernel can issue the whole image, compute what does not need it (the K tiles of its first tile), and wait in kf_stage_wait().
// n is a multiple of 256 doubles (2 KB): chunk c = 128 doubles, dealt out over the waves of the workgroup.
__device__ __forceinline__ void kf_stage_frag_async(double* dst, const double* __restrict__ src, int n) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6, lane = threadIdx.x & 63;
  for (int c = wave; c < n / 128; c += nwaves)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)c * 128 + 2 * lane),
                                     (__attribute__((address_space(3))) void*)(dst + c * 128), 16, 0, 0);
}
__device__ __forceinline__ void kf_stage_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Per-tile state of one latent for a grid of NB0 x NB1 16-row blocks.  EXACT kernels are instantiated for the block counts they run
// with: every loop bound is then a compile-time constant, the guards fold away and the compiler issues the fragment loads of a
// product ahead of its MFMAs.  With runtime bounds each 16 x 16 x 4 step sat behind its own LDS round trip (in-kernel stamps: 21 k
// cycles for the 64 products of a forward tile whose MFMAs take 4.4 k).
template <int NB0, int NB1> struct KfTile { double K0[4 * NB0], K1[4 * NB1], A0[4 * NB0], A1[4 * NB1], B0[4 * NB0], C0[4 * NB0]; };

// forward pieces of one tile: the K tiles (need the inducing inputs only), then A_p = P_p K_p, B0 = Alpha K1, C0 = S2 A1^2 (fragment images)
template <int NB0, int NB1, bool EXACT>
__device__ __forceinline__ void kf_forward_ktiles(KfTile<NB0, NB1>& t, const KfLat& L, const double* Z0, const double* Z1, const double* __restrict__ xrow,
                                                  bool valid, int g) {
  const KfFac &f0 = L.f[0], &f1 = L.f[1];
  const int nb0 = EXACT ? NB0 : f0.nb, nb1 = EXACT ? NB1 : f1.nb;
  kf_ktile<4 * NB0, EXACT>(t.K0, f0, nb0, Z0, xrow, valid, g);
  kf_ktile<4 * NB1, EXACT>(t.K1, f1, nb1, Z1, xrow, valid, g);
}
template <int NB0, int NB1, bool EXACT>
__device__ __forceinline__ void kf_forward_products(KfTile<NB0, NB1>& t, const KfLat& L, const KfFrags& F, int slot) {
  const int nb0 = EXACT ? NB0 : L.f[0].nb, nb1 = EXACT ? NB1 : L.f[1].nb;
#pragma unroll
  for (int q = 0; q < 4 * NB0; ++q) { t.A0[q] = 0.0; t.B0[q] = 0.0; t.C0[q] = 0.0; }
#pragma unroll
  for (int q = 0; q < 4 * NB1; ++q) t.A1[q] = 0.0;
  kf_frag_mm<NB0, 4 * NB0>(t.A0, F.P0, nb0, 4 * nb0, t.K0, slot);
  kf_frag_mm<NB1, 4 * NB1>(t.A1, F.P1, nb1, 4 * nb1, t.K1, slot);
  kf_frag_mm<NB0, 4 * NB1>(t.B0, F.Al, nb0, 4 * nb1, t.K1, slot);
  double sq[4 * NB1];
#pragma unroll
  for (int q = 0; q < 4 * NB1; ++q) sq[q] = t.A1[q] * t.A1[q];
  kf_frag_mm<NB0, 4 * NB1>(t.C0, F.S2, nb0, 4 * nb1, sq, slot);
}
template <int NB0, int NB1, bool EXACT>
__device__ __forceinline__ void kf_forward_tile(KfTile<NB0, NB1>& t, const KfLat& L, const KfFrags& F, const double* __restrict__ xrow, bool valid,
                                                int g, int slot) {
  kf_forward_ktiles<NB0, NB1, EXACT>(t, L, F.Z0, F.Z1, xrow, valid, g);
  kf_forward_products<NB0, NB1, EXACT>(t, L, F, slot);
}

// ---- forward: part[0..3][n] = q0 = k0.a0, q1 = k1.a1, mean = k0^T Alpha k1, st = (a0^2)^T S2 (a1^2) -------------------------
template <int NB0, int NB1>
__global__ void __launch_bounds__(64 * KF_WAVES)
k_kf_forward(KfArgs a) {
  __shared__ double sfr[4 * KF_FRAG];
  __shared__ double sz[2][KF_MQ_ * MAXD];
  const KfLat& L = a.lat[a.lat0 + blockIdx.y];
  {
    constexpr int n0 = NB0 * NB0 * 256, n1 = NB1 * NB1 * 256, n01 = NB0 * NB1 * 256;
    // asynchronous copies (kf_stage_frag_async): the K tiles of the wave's first tile are computed underneath them, from the global
    // copy of the scaled inducing inputs (the LDS copy becomes visible with the images, at kf_stage_wait)
    kf_stage_frag_async(sfr, L.f[0].PF, n0); kf_stage_frag_async(sfr + KF_FRAG, L.f[1].PF, n1);
    kf_stage_frag_async(sfr + 2 * KF_FRAG, L.AlF, n01); kf_stage_frag_async(sfr + 3 * KF_FRAG, L.S2F, n01);
    kf_stage_z(sz[0], L.f[0]); kf_stage_z(sz[1], L.f[1]);
  }
  const KfFrags F = {sfr, sfr + KF_FRAG, sfr + 2 * KF_FRAG, sfr + 3 * KF_FRAG, nullptr, nullptr, sz[0], sz[1]};
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15, slot = lane;      // slot: this lane's position in a fragment block
  const int w = blockIdx.x * KF_WAVES + (threadIdx.x >> 6);
  const int t1 = min((w + 1) * a.tpw, a.ntiles);
  KfTile<NB0, NB1> t;
  bool first = true;
  if (w * a.tpw < t1) {
    const int64_t pn = (int64_t)(w * a.tpw) * 16 + n;
    kf_forward_ktiles<NB0, NB1, true>(t, L, L.f[0].Zs, L.f[1].Zs, a.X + (pn < a.N ? pn : 0) * a.ldx, pn < a.N, g);
  }
  kf_stage_wait();
  for (int tile = w * a.tpw; tile < t1; ++tile) {
    const int64_t pn = (int64_t)tile * 16 + n;
    const bool valid = pn < a.N;
    if (!first) kf_forward_ktiles<NB0, NB1, true>(t, L, F.Z0, F.Z1, a.X + (valid ? pn : 0) * a.ldx, valid, g);
    first = false;
    kf_forward_products<NB0, NB1, true>(t, L, F, slot);
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

// tile in registers -> LDS image [row][KF_LD]
template <int Q>
__device__ __forceinline__ void kf_store_tile(double* T, const double (&V)[Q], int g, int n) {
#pragma unroll
  for (int q = 0; q < Q; ++q) T[(4 * q + g) * KF_LD + n] = V[q];
}
// acc[(rb, cb)] += A(rows of rb, 16 points) . B(16 points, cols of cb), A / B from LDS tiles (rows x points), optional squares and
// per-point scale on B;  MODE 0: plain, 1: both squared
template <int NBR, int NBC, int MODE, bool SCALED>
__device__ __forceinline__ void kf_accum(double (&acc)[NBR * NBC][4], const double* TA, const double* TB, const double (&sck)[4], int lane) {
  const int kk = lane >> 4, bj = lane & 15;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    double bf[NBC];
#pragma unroll
    for (int cb = 0; cb < NBC; ++cb) {
      double b = TB[(16 * cb + bj) * KF_LD + 4 * ks + kk];
      if (MODE == 1) b *= b;
      if (SCALED) b *= sck[ks];
      bf[cb] = b;
    }
#pragma unroll
    for (int rb = 0; rb < NBR; ++rb) {
      double v = TA[(16 * rb + bj) * KF_LD + 4 * ks + kk];      // A operand: (row l % 16, point 4 ks + l / 16)
      if (MODE == 1) v *= v;
#pragma unroll
      for (int cb = 0; cb < NBC; ++cb) kf_mfma16(acc[rb * NBC + cb], v, bf[cb]);
    }
  }
}

// ---- backward ---------------------------------------------------------------------------------------------------------------
// per point, with gm / gv the cotangents of mean / var and dq_p = -gv q_other:
//   B1 = Alpha^T K0, C1 = S2^T A0^2 ;  dA_p = 2 gv A_p . C_p ;  dK_p = gm B_p + 2 dq_p A_p + P_p dA_p ;  E_p = dq_p K_p + dA_p
//   dAlpha += K0 diag(gm) K1^T ; dS2 += A0^2 diag(gv) (A1^2)^T ; dP_p += E_p K_p^T
//   moments of t_p = dK_p . K_p against {1, x - zc, (x - zc)^2}  (-> d var_p, d Z_p, d ell_p in k_kf_finish)
template <int NB0, int NB1>
__global__ void __launch_bounds__(64 * KF_WAVES, 1)
k_kf_backward(KfArgs a) {
  extern __shared__ double lds[];
  const KfLat& L = a.lat[a.lat0 + blockIdx.y];
  const KfFac &f0 = L.f[0], &f1 = L.f[1];
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15, slot = lane;      // slot: this lane's position in a fragment block
  const int wib = threadIdx.x >> 6;
  const int w = blockIdx.x * KF_WAVES + wib;
  constexpr int Q0 = 4 * NB0, Q1 = 4 * NB1;
  constexpr int TILE = 16 * KF_NBMAX * KF_LD;
  double* bK0 = lds + wib * 4 * TILE; double* bK1 = bK0 + TILE; double* c0 = bK1 + TILE; double* c1 = c0 + TILE;
  double* sfr = lds + KF_WAVES * 4 * TILE;
  {
    constexpr int n0 = NB0 * NB0 * 256, n1 = NB1 * NB1 * 256, n01 = NB0 * NB1 * 256;
    kf_stage_frag_async(sfr, f0.PF, n0); kf_stage_frag_async(sfr + KF_FRAG, f1.PF, n1);
    kf_stage_frag_async(sfr + 2 * KF_FRAG, L.AlF, n01); kf_stage_frag_async(sfr + 3 * KF_FRAG, L.S2F, n01);
    kf_stage_frag_async(sfr + 4 * KF_FRAG, L.AlTF, n01); kf_stage_frag_async(sfr + 5 * KF_FRAG, L.S2TF, n01);
    kf_stage_z(sfr + 6 * KF_FRAG, f0); kf_stage_z(sfr + 6 * KF_FRAG + KF_MQ_ * MAXD, f1);
  }
  const KfFrags F = {sfr, sfr + KF_FRAG, sfr + 2 * KF_FRAG, sfr + 3 * KF_FRAG, sfr + 4 * KF_FRAG, sfr + 5 * KF_FRAG,
                     sfr + 6 * KF_FRAG, sfr + 6 * KF_FRAG + KF_MQ_ * MAXD};
  double accAl[NB0 * NB1][4], accS2[NB0 * NB1][4], accP0[NB0 * NB0][4], accP1[NB1 * NB1][4], accK0[NB0][4], accK1[NB1][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int b = 0; b < NB0 * NB1; ++b) { accAl[b][r] = 0.0; accS2[b][r] = 0.0; }
#pragma unroll
    for (int b = 0; b < NB0 * NB0; ++b) accP0[b][r] = 0.0;
#pragma unroll
    for (int b = 0; b < NB1 * NB1; ++b) accP1[b][r] = 0.0;
#pragma unroll
    for (int b = 0; b < NB0; ++b) accK0[b][r] = 0.0;
#pragma unroll
    for (int b = 0; b < NB1; ++b) accK1[b][r] = 0.0;
  }
  const double one4[4] = {1.0, 1.0, 1.0, 1.0};
  // column j = n of the moment matrix Psi_p: 0 -> 1, 1..D -> x_d - zc_d, D+1..2D -> (x_d - zc_d)^2, beyond -> 0   (fixed per lane)
  int psi_kind[2], psi_col[2]; double psi_zc[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const KfFac& f = L.f[p];
    const int d = (n == 0) ? 0 : ((n <= f.D) ? n - 1 : ((n <= 2 * f.D) ? n - 1 - f.D : 0));
    psi_kind[p] = (n == 0) ? 0 : ((n <= f.D) ? 1 : ((n <= 2 * f.D) ? 2 : 3));
    psi_col[p] = f.col0 + d; psi_zc[p] = KF_CONST(f.hyp)[KH_ZC + d];
  }

  const int t1 = min((w + 1) * a.tpw, a.ntiles);
  KfTile<NB0, NB1> t;
  bool first = true;
  if (w * a.tpw < t1) {          // K tiles of the first tile under the asynchronous staging copies (global copy of the inducing inputs)
    const int64_t pn = (int64_t)(w * a.tpw) * 16 + n;
    kf_forward_ktiles<NB0, NB1, true>(t, L, f0.Zs, f1.Zs, a.X + (pn < a.N ? pn : 0) * a.ldx, pn < a.N, g);
  }
  kf_stage_wait();
  for (int tile = w * a.tpw; tile < t1; ++tile) {
    const int64_t pn = (int64_t)tile * 16 + n;
    const bool valid = pn < a.N;
    const double* xrow = a.X + (valid ? pn : 0) * a.ldx;
    if (!first) kf_forward_ktiles<NB0, NB1, true>(t, L, F.Z0, F.Z1, xrow, valid, g);
    first = false;
    kf_forward_products<NB0, NB1, true>(t, L, F, slot);
    const double gmn = L.gm[pn], gvn = L.gv[pn], dq0n = L.dq0[pn], dq1n = L.dq1[pn];   // zero for padding points (scale 0 in the point-wise kernel)
    double B1[Q1], C1[Q1], sq[Q0];
#pragma unroll
    for (int q = 0; q < Q1; ++q) { B1[q] = 0.0; C1[q] = 0.0; }
#pragma unroll
    for (int q = 0; q < Q0; ++q) sq[q] = t.A0[q] * t.A0[q];
    kf_frag_mm<NB1, Q0>(B1, F.AlT, NB1, Q0, t.K0, slot);
    kf_frag_mm<NB1, Q0>(C1, F.S2T, NB1, Q0, sq, slot);
    double dA0[Q0], dA1[Q1], PdA0[Q0], PdA1[Q1];
#pragma unroll
    for (int q = 0; q < Q0; ++q) { dA0[q] = 2.0 * gvn * t.A0[q] * t.C0[q]; PdA0[q] = 0.0; }
#pragma unroll
    for (int q = 0; q < Q1; ++q) { dA1[q] = 2.0 * gvn * t.A1[q] * C1[q]; PdA1[q] = 0.0; }
    kf_frag_mm<NB0, Q0>(PdA0, F.P0, NB0, Q0, dA0, slot);
    kf_frag_mm<NB1, Q1>(PdA1, F.P1, NB1, Q1, dA1, slot);
    // ---- LDS images for the sums over points (k index = point): K tiles, E tiles
    kf_store_tile<Q0>(bK0, t.K0, g, n);
    kf_store_tile<Q1>(bK1, t.K1, g, n);
    {
      double E0[Q0], E1[Q1];
#pragma unroll
      for (int q = 0; q < Q0; ++q) E0[q] = fma(dq0n, t.K0[q], dA0[q]);
      kf_store_tile<Q0>(c0, E0, g, n);
#pragma unroll
      for (int q = 0; q < Q1; ++q) E1[q] = fma(dq1n, t.K1[q], dA1[q]);
      kf_store_tile<Q1>(c1, E1, g, n);
    }
    kf_wave_sync();
    double gmk[4], gvk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { gmk[ks] = __shfl(gmn, 4 * ks + g, 64); gvk[ks] = __shfl(gvn, 4 * ks + g, 64); }
    kf_accum<NB0, NB1, 0, true>(accAl, bK0, bK1, gmk, lane);
    kf_accum<NB0, NB0, 0, false>(accP0, c0, bK0, one4, lane);
    kf_accum<NB1, NB1, 0, false>(accP1, c1, bK1, one4, lane);
    kf_wave_sync();
    // B operand of the moment products: lane (kk = g, column j = n) needs psi_j of point 4 ks + g -- loaded here (most of the tile state
    // is dead by now), the round trip hides behind the dS2 products
    double psi[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // selects, not branches: the eight loads go out back to back (one round trip, not eight)
        const int64_t pt = (int64_t)tile * 16 + 4 * ks + g;
        const double xv = a.X[(pt < a.N ? pt : 0) * a.ldx + psi_col[p]] - psi_zc[p];
        const double v = psi_kind[p] == 0 ? 1.0 : (psi_kind[p] == 1 ? xv : (psi_kind[p] == 2 ? xv * xv : 0.0));
        psi[p][ks] = (pt < a.N) ? v : 0.0;
      }
    kf_store_tile<Q0>(c0, t.A0, g, n);
    kf_store_tile<Q1>(c1, t.A1, g, n);
    kf_wave_sync();
    kf_accum<NB0, NB1, 1, true>(accS2, c0, c1, gvk, lane);
    kf_wave_sync();
    {
      double t0[Q0], t1v[Q1];
#pragma unroll
      for (int q = 0; q < Q0; ++q) t0[q] = fma(gmn, t.B0[q], fma(2.0 * dq0n, t.A0[q], PdA0[q])) * t.K0[q];
      kf_store_tile<Q0>(c0, t0, g, n);
#pragma unroll
      for (int q = 0; q < Q1; ++q) t1v[q] = fma(gmn, B1[q], fma(2.0 * dq1n, t.A1[q], PdA1[q])) * t.K1[q];
      kf_store_tile<Q1>(c1, t1v, g, n);
    }
    kf_wave_sync();
    // moments: acc(rows of factor p, col j) += sum_n t_p[row, n] psi_j(x_n),  psi = {1, xc_d, xc_d^2}
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int rb = 0; rb < NB0; ++rb) kf_mfma16(accK0[rb], c0[(16 * rb + n) * KF_LD + 4 * ks + g], psi[0][ks]);
#pragma unroll
      for (int rb = 0; rb < NB1; ++rb) kf_mfma16(accK1[rb], c1[(16 * rb + n) * KF_LD + 4 * ks + g], psi[1][ks]);
    }
    kf_wave_sync();
  }
  // per-wave partials, [block][r][lane]; blocks numbered for the CAPACITY grid (KF_NBMAX x KF_NBMAX) that k_kf_reduce / k_kf_finish index
  double* out = L.acc + (int64_t)w * KF_ACC_DOUBLES;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int rb = 0; rb < NB0; ++rb)
#pragma unroll
      for (int cb = 0; cb < NB1; ++cb) {
        out[((KF_B_AL + rb * KF_NBMAX + cb) * 4 + r) * 64 + lane] = accAl[rb * NB1 + cb][r];
        out[((KF_B_S2 + rb * KF_NBMAX + cb) * 4 + r) * 64 + lane] = accS2[rb * NB1 + cb][r];
      }
#pragma unroll
    for (int rb = 0; rb < NB0; ++rb)
#pragma unroll
      for (int cb = 0; cb < NB0; ++cb) out[((KF_B_P0 + rb * KF_NBMAX + cb) * 4 + r) * 64 + lane] = accP0[rb * NB0 + cb][r];
#pragma unroll
    for (int rb = 0; rb < NB1; ++rb)
#pragma unroll
      for (int cb = 0; cb < NB1; ++cb) out[((KF_B_P1 + rb * KF_NBMAX + cb) * 4 + r) * 64 + lane] = accP1[rb * NB1 + cb][r];
#pragma unroll
    for (int rb = 0; rb < NB0; ++rb) out[((KF_B_K0 + rb) * 4 + r) * 64 + lane] = accK0[rb][r];
#pragma unroll
    for (int rb = 0; rb < NB1; ++rb) out[((KF_B_K1 + rb) * 4 + r) * 64 + lane] = accK1[rb][r];
  }
}

// ---- fixed-order sum of the partial accumulators; un-permutes the accumulator layout into row-major matrices ------------------
// block numbering shared by the accumulators of every variant (capacities nb0c, nb1c): Al | S2 (nb0c x nb1c each) | P0 | P1 | K0 | K1
struct KfBlock { int kind, rb, cb; };   // kind 0 Al, 1 S2, 2 P0, 3 P1, 4 moments of factor 0, 5 of factor 1
__device__ __forceinline__ KfBlock kf_block_decode(int blk, int nb0c, int nb1c) {
  KfBlock b;
  const int n01 = nb0c * nb1c;
  if (blk < n01) { b.kind = 0; b.rb = blk / nb1c; b.cb = blk % nb1c; return b; }
  blk -= n01;
  if (blk < n01) { b.kind = 1; b.rb = blk / nb1c; b.cb = blk % nb1c; return b; }
  blk -= n01;
  if (blk < nb0c * nb0c) { b.kind = 2; b.rb = blk / nb0c; b.cb = blk % nb0c; return b; }
  blk -= nb0c * nb0c;
  if (blk < nb1c * nb1c) { b.kind = 3; b.rb = blk / nb1c; b.cb = blk % nb1c; return b; }
  blk -= nb1c * nb1c;
  if (blk < nb0c) { b.kind = 4; b.rb = blk; b.cb = 0; return b; }
  b.kind = 5; b.rb = blk - nb0c; b.cb = 0;
  return b;
}
__host__ __device__ inline int kf_nblocks(int nb0c, int nb1c) { return 2 * nb0c * nb1c + nb0c * nb0c + nb1c * nb1c + nb0c + nb1c; }

// work layout per latent (doubles), R_p = 16 nb_p capacity rows, ldw = max(R0, R1):
//   dAl [R0][ldw] | dS2 [R0][ldw] | dP0 [R0][ldw] | dP1 [R1][ldw] | Kr0 [R0][16] | Kr1 [R1][16]
struct KfWork { int ldw, S2, P0, P1, K0, K1, total; };
__host__ __device__ inline KfWork kf_work_layout(int nb0c, int nb1c) {
  KfWork w;
  const int R0 = 16 * nb0c, R1 = 16 * nb1c;
  w.ldw = R0 > R1 ? R0 : R1;
  w.S2 = R0 * w.ldw; w.P0 = 2 * R0 * w.ldw; w.P1 = 3 * R0 * w.ldw; w.K0 = w.P1 + R1 * w.ldw; w.K1 = w.K0 + R0 * 16; w.total = w.K1 + R1 * 16;
  return w;
}
constexpr int KF_MQ = 16 * KF_NBMAX;
constexpr int KF_W_AL = 0, KF_W_S2 = KF_MQ * KF_MQ, KF_W_P0 = 2 * KF_MQ * KF_MQ, KF_W_P1 = 3 * KF_MQ * KF_MQ, KF_W_K0 = 4 * KF_MQ * KF_MQ,
              KF_W_K1 = KF_W_K0 + KF_MQ * 16, KF_W_TOTAL = KF_W_K1 + KF_MQ * 16;   // = kf_work_layout(KF_NBMAX, KF_NBMAX)
// sums of the point-wise block partials, one workgroup of 256 threads (see k_kron_pw_reduce)
__device__ __forceinline__ void kf_pw_reduce(const double* __restrict__ acc, int blocks, double kl_counted, double* __restrict__ pws) {
  __shared__ double sh[4];
  if (threadIdx.x == 0) pws[7] = kl_counted;   // 1 on the rank(s) that count the KL: summed over ranks like everything else
#pragma unroll
  for (int q = 0; q < KPW_ACC; ++q) {
    double v = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256) v += acc[(int64_t)KPW_ACC * b + q];
    v = block_sum<4>(v, sh);
    if (threadIdx.x == 0) pws[q] = v;
  }
}
constexpr int KF_RED_GROUPS = 16;
__global__ void __launch_bounds__(256)
k_kf_reduce(const double* __restrict__ acc0, const double* __restrict__ acc1, int nparts, double* __restrict__ work0, double* __restrict__ work1,
            int nb0c, int nb1c, int nb0_f, int nb1_f, int nb0_g, int nb1_g, const double* __restrict__ pwacc, int pw_blocks, double kl_counted,
            double* __restrict__ pws) {
  if (blockIdx.x == gridDim.x - 1) {          // the extra workgroup(s): the point-wise sums (first latent's only)
    if (blockIdx.y == 0 && pwacc) kf_pw_reduce(pwacc, pw_blocks, kl_counted, pws);
    return;
  }
  __shared__ double sh[KF_RED_GROUPS][16];
  const double* acc = blockIdx.y == 0 ? acc0 : acc1;
  double* work = blockIdx.y == 0 ? work0 : work1;
  const int nb0 = blockIdx.y == 0 ? nb0_f : nb0_g, nb1 = blockIdx.y == 0 ? nb1_f : nb1_g;   // this latent's own block counts (<= capacity)
  const int64_t accd = (int64_t)kf_nblocks(nb0c, nb1c) * 256;
  const int e = blockIdx.x * 16 + (threadIdx.x & 15), grp = threadIdx.x >> 4;
  const int per = (nparts + KF_RED_GROUPS - 1) / KF_RED_GROUPS;
  const int w0 = grp * per, w1 = min(w0 + per, nparts);
  // the point kernels write the accumulator blocks of the latent's OWN grid only: capacity blocks beyond it are never written (and
  // the buffer is never cleared), so they are not read either -- their sums are defined as zero
  bool live;
  {
    const KfBlock b0 = kf_block_decode(e >> 8, nb0c, nb1c);
    const int nr = (b0.kind == 0 || b0.kind == 1 || b0.kind == 2 || b0.kind == 4) ? nb0 : nb1;
    const int nc = (b0.kind == 0 || b0.kind == 1 || b0.kind == 3) ? nb1 : (b0.kind == 2 ? nb0 : 1);
    live = b0.rb < nr && b0.cb < nc;
  }
  double s = 0.0;
  if (live)
    for (int w = w0; w < w1; ++w) s += acc[(int64_t)w * accd + e];
  sh[grp][threadIdx.x & 15] = s;
  __syncthreads();
  if (grp != 0) return;
  double tot = 0.0;
#pragma unroll
  for (int q = 0; q < KF_RED_GROUPS; ++q) tot += sh[q][threadIdx.x & 15];
  // e = (block * 4 + r) * 64 + lane  ->  element (16 rb + 4 r + lane / 16, 16 cb + lane % 16) of its matrix
  const int lane = e & 63, r = (e >> 6) & 3, blk = e >> 8;
  const int ri = 4 * r + (lane >> 4), cj = lane & 15;
  const KfBlock b = kf_block_decode(blk, nb0c, nb1c);
  const KfWork w = kf_work_layout(nb0c, nb1c);
  const int base = b.kind == 0 ? 0 : b.kind == 1 ? w.S2 : b.kind == 2 ? w.P0 : b.kind == 3 ? w.P1 : b.kind == 4 ? w.K0 : w.K1;
  const int ld = b.kind <= 3 ? w.ldw : 16;
  work[base + (16 * b.rb + ri) * ld + 16 * b.cb + cj] = tot;
}

// =============================================================================================================================
// M x M stages
// =============================================================================================================================
struct KfFactorJob {
  const double* Z; int M, D, Mq; const double* hyp;   // Mq = 16 nb; hyp: the factor's record of the device hyperparameter block
  double* zc_out;   // = hyp + KH_ZC, written here: centre of the moment sums (mid-range of Z_p), read by the kernels BEHIND this one
  double* K;        // [128][128] identity padded Kuu factor + jitter (kept for the Kuu-gradient reductions)
  double* P;        // [Mq][Mq] row-major K^-1 (zero padded)
  double* PF;       // fragment order
  double* dvec;     // [Mq] diag(P), then dvec[Mq] = logdet K = sum log L_ii^2
  double* Zs;       // [Mq][D] Z / ell, zero rows beyond M (the point kernels' K tiles)
};
// info: Cholesky status word(s).  own_slots = 0: one word shared by every factor, zeroed by the host (predict).  own_slots = 1: word
// info + 2 * job (an 8-byte slot each, inside the result block that is downloaded anyway); the job's workgroup zeroes its own slot, so
// a step needs neither a memset launch nor a separate status copy.
struct KfFactorArgs { KfFactorJob job[4]; double jitter; double piv_rtol; int* info; int own_slots; };

// fragment image of a row-major matrix: F[(rb * ksn + ks) * 64 + lane] = A(16 rb + lane % 16, 4 ks + lane / 16)   (the A operand of v_mfma_f64_16x16x4)
__device__ __forceinline__ void kf_write_frag(double* __restrict__ F, int nbr, int ksn, int t, int nthreads, const double* __restrict__ A, int64_t lda,
                                              bool transposed) {
  const int total = nbr * ksn * 64;
  for (int idx = t; idx < total; idx += nthreads) {
    const int ln = idx & 63, blk = idx >> 6, ks = blk % ksn, rb = blk / ksn;
    const int row = 16 * rb + (ln & 15), k = 4 * ks + (ln >> 4);
    F[idx] = transposed ? A[(int64_t)k * lda + row] : A[(int64_t)row * lda + k];
  }
}


// One workgroup (1024 threads) per factor: K_p = k_p(Z_p) + jitter I (scripts/onoff.py:188-190), L = chol(K_p) and W = L^-1 in LDS
// (tf.cholesky onofftf/main.py:355; the explicit inverse of scripts/onoff.py:192 is formed as P = W^T W), logdet, diag(P).
__global__ void __launch_bounds__(1024)
k_kf_factor(KfFactorArgs a) {
  extern __shared__ double S[];   // [128][129]
  __shared__ PotrfShared psh;
  __shared__ double red[16];
  const KfFactorJob& jb = a.job[blockIdx.x];
  const int t = threadIdx.x, M = jb.M, Mq = jb.Mq, D = jb.D;
  const auto inv_ell = KF_CONST(jb.hyp) + KH_INV;
  const double var = KF_CONST(jb.hyp)[KH_VAR];
  int* const info = a.own_slots ? a.info + 2 * blockIdx.x : a.info;
  if (a.own_slots && t == 0) { info[0] = 0; info[1] = 0; }       // ordered before the factorisation's atomicCAS by the barriers below
  // inducing inputs into LDS first (the T tiles are idle until the factorisation): with Z read from global memory inside the loop
  // every entry of K sat behind two dependent L2 round trips (in-kernel stamps: 23 k cycles for 16 entries per thread).  K is built
  // for j <= i only and mirrored: rows i and nreal - 1 - i fold into one row of nreal + 1 entries.
  double* zl = &psh.T[0][0][0];
  for (int idx = t; idx < Mq * D; idx += 1024) {
    const int m = idx / D, d = idx - m * D;
    const double z = m < M ? jb.Z[idx] : 0.0;
    zl[idx] = z;
    jb.Zs[idx] = z * inv_ell[d];
  }
  __syncthreads();
  if (t < 64) {        // zc_d = mid-range of the inducing inputs (min / max are exact: any order gives the same bits).  One wave, lanes over the
                       // points: as a serial loop in MAXD lanes (100 dependent LDS reads) this held wave 0 back for ~10 k cycles before its share of K
    for (int d = 0; d < MAXD; ++d) {
      double zc = 0.0;
      if (d < D) {
        double lo = zl[d], hi = lo;                        // point 0: every lane starts from a real value
        for (int m = t; m < M; m += 64) { const double z = zl[m * D + d]; lo = fmin(lo, z); hi = fmax(hi, z); }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) { lo = fmin(lo, __shfl_xor(lo, sft, 64)); hi = fmax(hi, __shfl_xor(hi, sft, 64)); }
        zc = 0.5 * (lo + hi);
      }
      if (t == 0) jb.zc_out[d] = zc;
    }
  }
  const int nreal = ((M + PNB - 1) / PNB) * PNB;   // the factorisation touches the 32-column panels that hold real rows only
  for (int idx = t; idx < (nreal / 2) * (nreal + 1); idx += 1024) {
    int i = idx / (nreal + 1), j = idx - i * (nreal + 1);
    if (j > i) { j -= i + 1; i = nreal - 1 - i; }
    double v;
    if (i < M && j < M) {
      double r2 = 0.0;
      for (int d = 0; d < D; ++d) { const double q = (zl[i * D + d] - zl[j * D + d]) * inv_ell[d]; r2 = fma(q, q, r2); }
      v = var * exp(-0.5 * r2) + ((i == j) ? a.jitter : 0.0);     // same expression as k_rbf_matrix
    } else {
      v = (i == j) ? 1.0 : 0.0;
    }
    jb.K[i * PB + j] = v;
    S[i * PBLD + j] = v;
    if (i != j) { jb.K[j * PB + i] = v; S[j * PBLD + i] = 0.0; }
  }
  __syncthreads();
  if (!potrf_diag_lds(S, psh, 0, info, (M + PNB - 1) / PNB, true, a.piv_rtol * 2.220446049250313e-16 * (var + a.jitter))) return;   // info = plain 1-based pivot, as on the panel path; the slot says which factor
  // logdet K = sum log L_ii^2 (fixed order: strided partials, then 16 wave sums in order)
  {
    double ld = 0.0;
    for (int i = t; i < M; i += 1024) { const double l = S[i * PBLD + i]; ld += log(l * l); }
    ld = wave_sum(ld);
    if ((t & 63) == 0) red[t >> 6] = ld;
    __syncthreads();
    if (t == 0) { double q = 0.0; for (int w = 0; w < 16; ++w) q += red[w]; jb.dvec[Mq] = q; }
  }
  // P = W^T W on the MFMA pipe, operands straight from the LDS image: W[k][i] (k > i) is S[i][k], the diagonal is dinv, zero above.
  // A(i, k) = W[k][i], B(k, j) = W[k][j].  P is symmetric: blocks rb >= cb only, which need k >= 16 rb; the first four k-steps of a
  // block cross the diagonal of the A rows (and of the B rows when rb == cb) and take the select-guarded reads, the rest read S
  // directly; four k-steps of loads go out together (in-kernel stamps at M = 100: 57 k cycles with one guarded k-step at a time,
  // every LDS round trip exposed).  Dearest blocks first over the 16 waves.  Each lane ends with P(16 rb + 4 r + g, 16 cb + n),
  // r = 0..3 -- one 32-byte granule of the fragment image PF -- and mirrors it into block (cb, rb).
  {
    // (r4: one 16 x 16 x 4 MFMA per k-step; the A operand of a lane is row 16 rb + n of the block -- the same read pattern as the B operand)
    const int lane = t & 63, g = lane >> 4, n = lane & 15, wave = t >> 6;
    const int nbq = Mq / 16, ksn = Mq / 4;
    for (int blk = wave; blk < nbq * (nbq + 1) / 2; blk += 16) {
      int rb = 0, cb = blk;
      while (cb > rb) { cb -= rb + 1; ++rb; }
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      const int j = 16 * cb + n, i_a = 16 * rb + n;
      const double* Sb = S + j * PBLD + g;
      const double* Sa = S + i_a * PBLD + g;
      {
        double bv[4], av[4];
        const double dj = psh.dinv[j], di = psh.dinv[i_a];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = 16 * rb + 4 * u + g;
          const double sb = Sb[4 * (4 * rb + u)], sa = Sa[4 * (4 * rb + u)];
          bv[u] = k > j ? sb : (k == j ? dj : 0.0);
          av[u] = k > i_a ? sa : (k == i_a ? di : 0.0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) kf_mfma16(acc, av[u], bv[u]);
      }
      for (int ks = 4 * rb + 4; ks < ksn; ks += 4) {
        double bv[4], av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { bv[u] = Sb[4 * (ks + u)]; av[u] = Sa[4 * (ks + u)]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) kf_mfma16(acc, av[u], bv[u]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * rb + 4 * r + g;
        const double v = (i < M && j < M) ? acc[r] : 0.0;
        jb.P[i * Mq + j] = v;
        // fragment image (kf_write_frag): element (row R, column C) -> block (R / 16, C / 4), lane R % 16 + 16 (C % 4)
        jb.PF[(int64_t)(rb * ksn + 4 * cb + (n >> 2)) * 64 + 4 * r + g + 16 * (n & 3)] = v;           // (row i, column j)
        if (rb != cb) {
          jb.P[j * Mq + i] = v;
          jb.PF[(int64_t)(cb * ksn + 4 * rb + r) * 64 + n + 16 * g] = v;                              // (row j, column i): the mirror
        }
        if (i == j) jb.dvec[i] = v;
      }
    }
  }
}

// ---- small dense algebra of the M x M stages: every operand lives in LDS with row stride KF_SLD (odd: column walks are conflict free)
constexpr int KF_SLD = KF_MQ + 1;
constexpr int KF_SMAT = KF_MQ * KF_SLD;
// C (m x n) = op(A) (m x k) * op(B) (k x n) [+ C]; m, n, k multiples of 16; the caller synchronises.  One wave per 16 x 16 block of C
// on the MFMA pipe (operands straight from the row-major LDS images; 8 k-steps of four 4x4x4 products at 32 points).  The first
// version gave every thread one element of C and a scalar loop over k: 64 LDS reads per thread and product, 2-3 us for each of the
// five chained products of k_kf_finish.
template <bool TA, bool TB, bool ACC>
__device__ __forceinline__ void kf_lds_mm(double* C, const double* A, const double* B, int m, int n, int k, int w0 = 0) {
  // w0: first wave of this product (independent products of one phase start on different waves: a 32 x 32 product is four blocks)
  const int nwaves = blockDim.x >> 6, wave = ((threadIdx.x >> 6) + nwaves - w0) % nwaves, lane = threadIdx.x & 63;
  const int kq = lane >> 4, bj = lane & 15, ci = lane >> 4;
  const int nbn = n / 16, nblk = (m / 16) * nbn;
  for (int blk = wave; blk < nblk; blk += nwaves) {
    const int rb = blk / nbn, cb = blk - rb * nbn;
    double acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = ACC ? C[(16 * rb + 4 * r + ci) * KF_SLD + 16 * cb + bj] : 0.0;
    for (int ks = 0; ks < k / 4; ++ks) {      // one 16 x 16 x 4 product per k-step: A (row l % 16, k l / 16), B (k l / 16, column l % 16)
      const int q = 4 * ks + kq;
      const double b = TB ? B[(16 * cb + bj) * KF_SLD + q] : B[q * KF_SLD + 16 * cb + bj];
      const int i = 16 * rb + bj;
      const double a = TA ? A[q * KF_SLD + i] : A[i * KF_SLD + q];
      kf_mfma16(acc, a, b);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(16 * rb + 4 * r + ci) * KF_SLD + 16 * cb + bj] = acc[r];
  }
}
// The same product with one thread per element of C and k summed in ascending order (the first version of these kernels).  Kept
// for the first phase of k_kf_finish (dP_p += dAlpha T^T, Q_p, X, dU): everything there is multiplied by P_p = K_p^-1 on both sides
// afterwards, and at cond(K_s) ~ 5e7 (the pptr initialisation) the summation order of these inputs moves the last digits of
// d var / d ell around the op-order floor of tests/test_gpu_pptr_params.py -- the MFMA order put d var_g at 3.3x the floor (and d ell_g
// at 0.5x instead of 1x), this order keeps the 2.3x the test has always seen.  Neither is more accurate against the 40-digit values.
template <bool TA, bool TB, bool ACC>
__device__ __forceinline__ void kf_lds_mm_seq(double* C, const double* A, const double* B, int m, int n, int k) {
  for (int idx = threadIdx.x; idx < m * n; idx += blockDim.x) {
    const int i = idx / n, j = idx - i * n;
    double v = 0.0;
#pragma unroll 4
    for (int q = 0; q < k; ++q) v = fma(TA ? A[q * KF_SLD + i] : A[i * KF_SLD + q], TB ? B[j * KF_SLD + q] : B[q * KF_SLD + j], v);
    if (ACC) C[i * KF_SLD + j] += v; else C[i * KF_SLD + j] = v;
  }
}
// ---- compensated (two-double) products for the P_p sandwich of the reverse pass -------------------------------------------------
// dK_p = -P_p sym(dP_p) P_p is where the factored reverse pass loses its digits on an ill-conditioned factor: the entries of P_p = K_p^-1
// are ~cond(K_p) / var in size with alternating signs and the products cancel down to the gradient.  CPU experiment with a working
// precision per stage (tools/dd_experiment.py; pptr init, 32 x 32 grid, cond(K_s) = 5e7, largest d ELBO / d Z_s entry): everything in
// float64 1.7e-3; THESE TWO PRODUCTS carried in twice the precision (inputs and result plain float64) 1.1e-8; nothing else matters
// (the inverse itself, Alpha, the sums over points, the reductions against K_p: unchanged results in 106 bits).  Error-free
// transformations: Knuth's two-sum, the FMA two-product; the running sum is (hi, lo), lo collects every rounding error (Ogita, Rump &
// Oishi's Dot2: the result is as accurate as if computed in doubled precision and rounded once).  Contraction must stay off inside.
__device__ __forceinline__ void kf_dd_mac(double& sh, double& sl, double a, double b) {          // (sh, sl) += a b
#pragma clang fp contract(off)
  const double p = a * b;
  const double e = __builtin_fma(a, b, -p);
  const double s = sh + p;
  const double bb = s - sh;
  const double err = (sh - (s - bb)) + (p - bb);
  sh = s;
  sl += err + e;
}
__device__ __forceinline__ void kf_dd_mac2(double& sh, double& sl, double a, double bh, double bl) {   // (sh, sl) += a (bh + bl)
#pragma clang fp contract(off)
  const double p = a * bh;
  const double e = __builtin_fma(a, bl, __builtin_fma(a, bh, -p));
  const double s = sh + p;
  const double bb = s - sh;
  const double err = (sh - (s - bb)) + (p - bb);
  sh = s;
  sl += err + e;
}
// (Ch, Cl) = A B, one thread per element, A / B float64 LDS images (row stride KF_SLD); result normalised (|Cl| <= ulp(Ch) / 2)
__device__ __forceinline__ void kf_lds_mm_dd(double* Ch, double* Cl, const double* A, const double* B, int m, int n, int k) {
  for (int idx = threadIdx.x; idx < m * n; idx += blockDim.x) {
    const int i = idx / n, j = idx - i * n;
    double sh = 0.0, sl = 0.0;
#pragma unroll 4
    for (int q = 0; q < k; ++q) kf_dd_mac(sh, sl, A[i * KF_SLD + q], B[q * KF_SLD + j]);
    const double h = sh + sl;
    Ch[i * KF_SLD + j] = h;
    Cl[i * KF_SLD + j] = sl - (h - sh);
  }
}
// G = -(A (Bh + Bl)) - coef A2, rounded once to float64 (A2 = the matrix whose multiple is subtracted: P_p, KL logdet term)
__device__ __forceinline__ void kf_lds_mm_dd2_neg(double* G, const double* A, const double* Bh, const double* Bl, const double* A2, double coef, int m, int n,
                                                  int k) {
  for (int idx = threadIdx.x; idx < m * n; idx += blockDim.x) {
    const int i = idx / n, j = idx - i * n;
    double sh = 0.0, sl = 0.0;
#pragma unroll 4
    for (int q = 0; q < k; ++q) kf_dd_mac2(sh, sl, A[i * KF_SLD + q], Bh[q * KF_SLD + j], Bl[q * KF_SLD + j]);
    kf_dd_mac(sh, sl, coef, A2[i * KF_SLD + j]);
    G[i * KF_SLD + j] = -(sh + sl);
  }
}
__device__ __forceinline__ void kf_lds_load(double* dst, const double* __restrict__ src, int rows, int cols, int ld) {
  for (int idx = threadIdx.x; idx < rows * cols; idx += blockDim.x) { const int i = idx / cols, j = idx - i * cols; dst[i * KF_SLD + j] = src[(int64_t)i * ld + j]; }
}
__device__ __forceinline__ void kf_lds_store(double* __restrict__ dst, const double* src, int rows, int cols, int ld) {
  for (int idx = threadIdx.x; idx < rows * cols; idx += blockDim.x) { const int i = idx / cols, j = idx - i * cols; dst[(int64_t)i * ld + j] = src[i * KF_SLD + j]; }
}
// fragment image of an LDS matrix (see kf_write_frag)
__device__ __forceinline__ void kf_lds_frag(double* __restrict__ F, int nbr, int ksn, const double* A, bool transposed) {
  const int total = nbr * ksn * 64;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int ln = idx & 63, blk = idx >> 6, ks = blk % ksn, rb = blk / ksn;
    const int row = 16 * rb + (ln & 15), k = 4 * ks + (ln >> 4);
    F[idx] = transposed ? A[k * KF_SLD + row] : A[row * KF_SLD + k];
  }
}

struct KfLatentJob {
  int M0, M1, Mq0, Mq1;
  const double *P0, *P1, *dvec0, *dvec1;     // from k_kf_factor
  const double *PF0, *PF1;                   // their fragment images (larger-grid kernels)
  const double *u, *s;                       // [M0*M1] from the parameter pack
  double *U, *S2, *T0, *T1, *Al;             // [Mq0][Mq1] zero padded: U, s^2, U P1, P0 U, Alpha = P0 U P1
  double *AlF, *S2F, *AlTF, *S2TF;           // fragment images
  double* klv;                               // [8]: sum U.Alpha, sum log s^2, sum d0 d1 s^2, logdet K0, logdet K1
};
struct KfLatentArgs { KfLatentJob job[2]; };

__global__ void __launch_bounds__(1024)
k_kf_latent(KfLatentArgs a) {
  __shared__ double sm[7 * KF_SMAT];
  __shared__ double sh[3][16];
  double *sP0 = sm, *sP1 = sm + KF_SMAT, *sU = sm + 2 * KF_SMAT, *sS2 = sm + 3 * KF_SMAT, *sT = sm + 4 * KF_SMAT, *sAl = sm + 5 * KF_SMAT,
         *sT1 = sm + 6 * KF_SMAT;
  const KfLatentJob& jb = a.job[blockIdx.x];
  const int t = threadIdx.x, M0 = jb.M0, M1 = jb.M1, Mq0 = jb.Mq0, Mq1 = jb.Mq1;
  kf_lds_load(sP0, jb.P0, Mq0, Mq0, Mq0);
  kf_lds_load(sP1, jb.P1, Mq1, Mq1, Mq1);
  for (int idx = t; idx < Mq0 * Mq1; idx += 1024) {
    const int i = idx / Mq1, j = idx - i * Mq1;
    const bool in = i < M0 && j < M1;
    const double sv = in ? jb.s[i * M1 + j] : 0.0, uv = in ? jb.u[i * M1 + j] : 0.0;
    sU[i * KF_SLD + j] = uv; sS2[i * KF_SLD + j] = sv * sv;
    jb.U[idx] = uv; jb.S2[idx] = sv * sv;
  }
  __syncthreads();
  kf_lds_mm<false, false, false>(sT1, sP0, sU, Mq0, Mq1, Mq0);      // T1 = P0 U
  kf_lds_mm<false, false, false>(sT, sU, sP1, Mq0, Mq1, Mq1, 4);    // T0 = U P1   (independent: same phase, own buffer, other waves)
  __syncthreads();
  kf_lds_store(jb.T1, sT1, Mq0, Mq1, Mq1);
  kf_lds_store(jb.T0, sT, Mq0, Mq1, Mq1);
  kf_lds_mm<false, false, false>(sAl, sP0, sT, Mq0, Mq1, Mq0);      // Alpha = P0 (U P1)   (= __kron_mv, scripts/onoff.py:193)
  __syncthreads();
  kf_lds_store(jb.Al, sAl, Mq0, Mq1, Mq1);
  kf_lds_frag(jb.AlF, Mq0 / 16, Mq1 / 4, sAl, false);
  kf_lds_frag(jb.S2F, Mq0 / 16, Mq1 / 4, sS2, false);
  kf_lds_frag(jb.AlTF, Mq1 / 16, Mq0 / 4, sAl, true);
  kf_lds_frag(jb.S2TF, Mq1 / 16, Mq0 / 4, sS2, true);
  // KL scalars (GaussKLkron, onofftf/main.py:350-387, factored): fixed-order sums
  double av = 0.0, bv = 0.0, cv = 0.0;
  for (int idx = t; idx < M0 * M1; idx += 1024) {
    const int i = idx / M1, j = idx - i * M1;
    const double sv = jb.s[idx];
    av = fma(sU[i * KF_SLD + j], sAl[i * KF_SLD + j], av);
    bv += log(sv * sv);
    cv = fma(jb.dvec0[i] * jb.dvec1[j], sv * sv, cv);
  }
  double vals[3] = {av, bv, cv};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const double v = wave_sum(vals[q]);
    if ((t & 63) == 0) sh[q][t >> 6] = v;
  }
  __syncthreads();
  if (t < 3) { double r = 0.0; for (int w = 0; w < 16; ++w) r += sh[t][w]; jb.klv[t] = r; }
  if (t == 0) { jb.klv[3] = jb.dvec0[Mq0]; jb.klv[4] = jb.dvec1[Mq1]; }
}

// One workgroup per latent: the M x M reverse pass.  Inputs: the summed accumulators (k_kf_reduce) and the forward matrices.
//   dU = P0 dAl P1 ;  dP0 += dAl T0^T ;  dP1 += T1^T dAl                                   (Alpha = P0 U P1)
//   KL:  dP_p -= 1/4 (Q_p + Q_p^T) + 1/2 diag(w_p),  Q0 = T0 U^T, Q1 = U^T T1, w0_i = sum_j d1_j s2_ij, w1_j = sum_i d0_i s2_ij
//   dK_p = -P_p sym(dP_p) P_p - kl * 1/2 M_other P_p  ->  Kuu-style reductions against K_p into krow_p; data moments added
//   gu = dU - kl Alpha ;  gs = 2 s dS2 - kl (-1/s + d0_i d1_j s)
struct KfFinishJob {
  int M0, M1, Mq0, Mq1, D0, D1;
  const double *P0, *P1, *dvec0, *dvec1, *K0, *K1, *Z0, *Z1;
  const double *PF0, *PF1;
  const double *hyp0, *hyp1;          // the two factors' records of the device hyperparameter block (zc: centre of the moment sums)
  const double *U, *S2, *T0, *T1, *Al, *s;
  const double* work;  // KF_W_TOTAL summed accumulators
  double *krow0, *krow1, *gu, *gs;     // outputs: [M0][2 + 2 D0], [M1][2 + 2 D1], [M0*M1], [M0*M1]
};
struct KfFinishArgs { KfFinishJob job[2]; double jitter; int with_kl; };
constexpr size_t KF_FIN_LDS = sizeof(double) * 10 * KF_SMAT;

// grid (2, latents): workgroup (p, h) runs the reverse pass of factor p of latent h; workgroup p = 0 also forms dU -> gu, gs
__global__ void __launch_bounds__(1024)
k_kf_finish(KfFinishArgs a) {
  extern __shared__ double sm[];
  double *sP = sm, *sPo = sm + KF_SMAT, *sdAl = sm + 2 * KF_SMAT, *sdP = sm + 3 * KF_SMAT, *sT = sm + 4 * KF_SMAT, *sU = sm + 5 * KF_SMAT,
         *sX = sm + 6 * KF_SMAT, *sG = sm + 7 * KF_SMAT, *sQ = sm + 8 * KF_SMAT, *sK = sm + 9 * KF_SMAT;
  const KfFinishJob& jb = a.job[blockIdx.y];
  const int p = blockIdx.x;
  const int t = threadIdx.x, M0 = jb.M0, M1 = jb.M1, Mq0 = jb.Mq0, Mq1 = jb.Mq1;
  const bool kl = a.with_kl != 0;
  const int M = p == 0 ? M0 : M1, Mq = p == 0 ? Mq0 : Mq1, Mo = p == 0 ? M1 : M0, D = p == 0 ? jb.D0 : jb.D1;
  const double* Z = p == 0 ? jb.Z0 : jb.Z1;
  const auto zc = KF_CONST(p == 0 ? jb.hyp0 : jb.hyp1) + KH_ZC;
  const double* Kr = jb.work + (p == 0 ? KF_W_K0 : KF_W_K1);
  double* krow = p == 0 ? jb.krow0 : jb.krow1;
  kf_lds_load(sP, p == 0 ? jb.P0 : jb.P1, Mq, Mq, Mq);
  kf_lds_load(sdAl, jb.work + KF_W_AL, Mq0, Mq1, KF_MQ);
  kf_lds_load(sdP, jb.work + (p == 0 ? KF_W_P0 : KF_W_P1), Mq, Mq, KF_MQ);
  kf_lds_load(sT, p == 0 ? jb.T0 : jb.T1, Mq0, Mq1, Mq1);
  kf_lds_load(sU, jb.U, Mq0, Mq1, Mq1);
  kf_lds_load(sK, p == 0 ? jb.K0 : jb.K1, M, M, PB);
  if (p == 0) kf_lds_load(sPo, jb.P1, Mq1, Mq1, Mq1);
  // small per-row operands of the later phases in LDS: the inducing inputs and (KL) w_i = sum_o d_o S2[i, o] -- read inside serial loops
  // they cost an L2 round trip per element (the diagonal threads of the X phase walked 32 of them one after the other)
  __shared__ double sZ[KF_MQ * MAXD], sW[KF_MQ];
  for (int idx = t; idx < M * D; idx += 1024) sZ[idx] = Z[idx];
  if (kl) {
    const int i = t >> 5, l = t & 31;        // 32 lanes per row
    double w = 0.0;
    if (i < M) {
      if (p == 0) { for (int o = l; o < M1; o += 32) w = fma(jb.dvec1[o], jb.S2[i * Mq1 + o], w); }
      else { for (int o = l; o < M0; o += 32) w = fma(jb.dvec0[o], jb.S2[o * Mq1 + i], w); }
    }
#pragma unroll
    for (int sft = 16; sft >= 1; sft >>= 1) w += __shfl_xor(w, sft, 64);
    if (l == 0 && i < KF_MQ) sW[i] = w;
  }
  __syncthreads();
  if (p == 0) {
    // (rounds 3: these ran one thread per element with k in ascending order, to keep the last digits of d var / d ell where a test's
    // "op-order floor" margin had seen them; with the compensated sandwich below that margin is gone -- section 1 of HISTORY.md -- and
    // the products are back on the MFMA pipe, the independent ones of the phase on different waves)
    kf_lds_mm<false, false, false>(sX, sdAl, sPo, Mq0, Mq1, Mq1);           // X = dAl P1
    kf_lds_mm<false, true, true>(sdP, sdAl, sT, Mq0, Mq0, Mq1, 4);          // dP0 += dAl T0^T
    if (kl) kf_lds_mm<false, true, false>(sQ, sT, sU, Mq0, Mq0, Mq1, 8);    // Q0 = T0 U^T
    __syncthreads();
    kf_lds_mm<false, false, false>(sG, sP, sX, Mq0, Mq1, Mq0);              // dU = P0 X
    __syncthreads();
    for (int idx = t; idx < M0 * M1; idx += 1024) {
      const int i = idx / M1, j = idx - i * M1;
      const double sv = jb.s[idx];
      double gu = sG[i * KF_SLD + j], gs = 2.0 * sv * jb.work[KF_W_S2 + i * KF_MQ + j];
      if (kl) { gu -= jb.Al[i * Mq1 + j]; gs -= (-1.0 / sv + jb.dvec0[i] * jb.dvec1[j] * sv); }
      jb.gu[idx] = gu; jb.gs[idx] = gs;
    }
  } else {
    kf_lds_mm<true, false, true>(sdP, sT, sdAl, Mq1, Mq1, Mq0);             // dP1 += T1^T dAl
    if (kl) kf_lds_mm<true, false, false>(sQ, sU, sT, Mq1, Mq1, Mq0, 4);    // Q1 = U^T T1
  }
  __syncthreads();
  // sX = sym(dP) [- kl pieces]
  for (int idx = t; idx < Mq * Mq; idx += 1024) {
    const int i = idx / Mq, j = idx - i * Mq;
    double v = 0.5 * (sdP[i * KF_SLD + j] + sdP[j * KF_SLD + i]);
    if (kl) {
      v -= 0.25 * (sQ[i * KF_SLD + j] + sQ[j * KF_SLD + i]);
      if (i == j) v -= 0.5 * sW[i];
    }
    sX[i * KF_SLD + j] = v;
  }
  __syncthreads();
  // G = -P sym(dP) P - coef P in compensated arithmetic (kf_dd_mac, above): Q = sym(dP) P as (hi, lo) -- lo in the dead dP buffer --
  // then P (Q_hi + Q_lo) + coef P summed in (hi, lo) and rounded once
  kf_lds_mm_dd(sQ, sdP, sX, sP, Mq, Mq, Mq);
  __syncthreads();
  const double coef = kl ? 0.5 * (double)Mo : 0.0;
  kf_lds_mm_dd2_neg(sG, sP, sQ, sdP, sP, coef, Mq, Mq, Mq);
  __syncthreads();
  // krow[m][c]: Kuu part (as k_kuu_grad, Kz = K_p - jitter I) + data moments rebuilt around z_m
  // eight lanes per (m, c) entry sweep j (fixed-order tree over the eight partial sums)
  const int W = 2 + 2 * D;
  for (int base = 0; base < M * W; base += 128) {
    const int idx = base + (t >> 3), l = t & 7;
    const bool on = idx < M * W;
    const int m = on ? idx / W : 0, c = on ? idx - m * W : 0;
    double v = 0.0;
    const int d = (c == 0) ? 0 : (c - 1) % D;
    const double zm = sZ[m * D + d];
    if (on && c <= 2 * D) {
      for (int j = l; j < M; j += 8) {
        const double kz = sK[m * KF_SLD + j] - ((m == j) ? a.jitter : 0.0);
        const double tt = sG[m * KF_SLD + j] * kz;
        const double df = sZ[j * D + d] - zm;
        v += (c == 0) ? tt : ((c <= D) ? 2.0 * tt * df : tt * df * df);
      }
    }
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    if (on && l == 0) {
      if (c <= 2 * D) {
        const double s0 = Kr[m * 16];
        if (c == 0) v += s0;
        else {
          const double dz = zm - zc[d], s1 = Kr[m * 16 + 1 + d];
          if (c <= D) v += s1 - dz * s0;
          else v += Kr[m * 16 + 1 + D + d] - 2.0 * dz * s1 + dz * dz * s0;
        }
      }
      krow[idx] = v;
    }
  }
}

}  // namespace zigp

namespace zigp {

#include "zigp_kronl.h"

// =============================================================================================================================
// host orchestration
// =============================================================================================================================
constexpr size_t KF_BWD_LDS = sizeof(double) * (KF_WAVES * 4 * 16 * KF_NBMAX * KF_LD + 6 * KF_FRAG + 2 * KF_MQ_ * MAXD);
struct KfState {
  DevBuf in, mat, pts, acc, res, out, spill, fit;
};
static void kf_free(KfState* k) {
  DevBuf* bs[] = {&k->in, &k->mat, &k->pts, &k->acc, &k->res, &k->out, &k->spill, &k->fit};
  for (DevBuf* b : bs) b->release();
  delete k;
}

// Variants of the point-stage kernels by capacity (16-row blocks per factor):
//   small  <2, 2>  accumulators in registers (grids up to 32 x 32: BASELINE cfg5)
//   large  <1, 7>  operands spilled per tile, k_kfl_accum (the reference's [10, 100] grid, scripts/onoff.py:52-53)
// anything else (and more than 7 input columns per factor: 1 + 2 D moment columns <= 16) takes the GEMM-panel path of zigp_kron.hip.
struct KfPlan { bool ok, large; int nb0c, nb1c; };
static KfPlan kf_plan(const zigp_kron_params* p, int nlat) {
  KfPlan pl = {false, false, 0, 0};
  if (p->D0 > 7 || p->D1 > 7) return pl;
  const int m0 = std::max(p->M0f, nlat == 2 ? p->M0g : 0), m1 = std::max(p->M1f, nlat == 2 ? p->M1g : 0);
  if (m0 <= 16 * KF_NBMAX && m1 <= 16 * KF_NBMAX) { pl.ok = true; pl.nb0c = KF_NBMAX; pl.nb1c = KF_NBMAX; return pl; }
  if (m0 <= 16 && m1 <= 112) { pl.ok = true; pl.large = true; pl.nb0c = 1; pl.nb1c = 7; return pl; }
  return pl;
}
static bool kf_eligible(const zigp_kron_params* p, int nlat) { return kf_plan(p, nlat).ok; }

// launch helpers: the small-grid kernels are instantiated for the exact block counts (1 or 2 per factor)
template <int NB0, int NB1>
static int kf_launch_small(zigp_ctx* c, bool backward, dim3 grid, const KfArgs& ka) {
  if (backward) {
    static bool attr = false;
    if (!attr) {
      ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kf_backward<NB0, NB1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KF_BWD_LDS));
      attr = true;
    }
    hipLaunchKernelGGL((k_kf_backward<NB0, NB1>), grid, dim3(64 * KF_WAVES), KF_BWD_LDS, c->stream, ka);
  } else {
    hipLaunchKernelGGL((k_kf_forward<NB0, NB1>), grid, dim3(64 * KF_WAVES), 0, c->stream, ka);
  }
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}
static int kf_launch_small(zigp_ctx* c, int nb0, int nb1, bool backward, dim3 grid, const KfArgs& ka) {
  if (nb0 == 1 && nb1 == 1) return kf_launch_small<1, 1>(c, backward, grid, ka);
  if (nb0 == 1 && nb1 == 2) return kf_launch_small<1, 2>(c, backward, grid, ka);
  if (nb0 == 2 && nb1 == 1) return kf_launch_small<2, 1>(c, backward, grid, ka);
  return kf_launch_small<2, 2>(c, backward, grid, ka);
}
// one launch for both latents when their block counts agree, else one per latent (KfArgs::lat0 selects it)
static int kf_launch_small_latents(zigp_ctx* c, const int (*Mq)[2], int nlat, bool backward, unsigned gx, KfArgs& ka) {
  const bool same = nlat == 1 || (Mq[0][0] == Mq[1][0] && Mq[0][1] == Mq[1][1]);
  if (same) { ka.lat0 = 0; return kf_launch_small(c, Mq[0][0] / 16, Mq[0][1] / 16, backward, dim3(gx, nlat), ka); }
  for (int h = 0; h < nlat; ++h) { ka.lat0 = h; ZIGP_TRY(kf_launch_small(c, Mq[h][0] / 16, Mq[h][1] / 16, backward, dim3(gx, 1), ka)); }
  ka.lat0 = 0;
  return 0;
}

struct KfHostLatent { int M[2]; const double* Z[2]; const double* ell[2]; double var[2]; const double* u; const double* s; };
constexpr int KF_KROW_W = 2 + 2 * MAXD;

// Fixed-order sums of the point-wise block partials -> pws[0..KPW_ACC) (kf_pw_reduce, above): the result block of a step has a size
// that depends on the inducing grid only (never on the shard), so a data-parallel run can sum it over ranks in place (comm_allreduce).
// Value-only steps launch it on its own; gradient steps run it as one extra workgroup of k_kf_reduce (a launch less on a step that is
// launch-bound at minibatch size).
__global__ void __launch_bounds__(256)
k_kron_pw_reduce(const double* __restrict__ acc, int blocks, double kl_counted, double* __restrict__ pws) { kf_pw_reduce(acc, blocks, kl_counted, pws); }

// =============================================================================================================================
// device-resident fit loop (zigp_kron_fit_steps): free state -> parameter image, result block -> Adam update
// =============================================================================================================================
// The reference's fit is 50 000 minibatch steps of one model shape (scripts/onoff.py:375-381), each: gradient of cost = -(scale data - KL)
// (:318,334), chained through the Log1pe transforms of the positive parameters (:88-123), one tf.train.AdamOptimizer per learning
// rate (:325-350; TF defaults beta1 .9, beta2 .999, eps 1e-8, lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t)).  Here the parameter image
// the kernels read is produced ON THE DEVICE from the free state (k_fit_update), so n steps are enqueued back to back on one stream and
// the host synchronises once per call.  Block order of the flat free-state vector (ZIGP_FIT_BLOCKS = 17): for latent f, then g:
// Z0, Z1, u, s, ell0, ell1, var0, var1; then the likelihood variance.
constexpr int KFIT_BLOCKS = ZIGP_FIT_BLOCKS;
struct KfFitDesc {
  int off[KFIT_BLOCKS], n[KFIT_BLOCKS], dst[KFIT_BLOCKS], positive[KFIT_BLOCKS];   // offset in the free vector, size, offset in the parameter image
  double lr[KFIT_BLOCKS];
  int nlat, D0, D1, M[2][2], Mq[2][2];
  int off_z[2][2], off_hyp, off_hyp2;          // parameter image; the two hyperparameter blocks (step i reads block i % 2, its update writes the other)
  int big_blk[8], big_off[9], hyp_blk[9], hyp_off[10];   // thread -> element map of k_fit_update: the big blocks (Z, u, s) in order, then the hyperparameter blocks in a workgroup of their own
  int res_size, res_krow0, res_krow1, res_gu, res_gs, res_pws, res_info;   // result block (doubles)
  double beta1, beta2, eps;
};
struct KfFitArgs {
  KfFitDesc d;
  double *x, *m, *v;            // free state and Adam moments [n_free]
  double* img;                  // parameter image (KfState::in)
  const double* res;            // result block of the step that just ran
  double* hist;                 // [n_steps][2]: data term, KL
  int* fail;                    // [4]: 0 = fine, else {1 + step, factor job, pivot}
  int step;                     // index within this call
  double lr_sq, lr_den;         // sqrt(1 - beta2^t), 1 - beta1^t of this step (computed on the host exactly as zigp.optim.AdamGroups does)
  int update;                   // 0: only free state -> image (first launch of a call)
};

// np.logaddexp(0, x) + 1e-6 (zigp/transforms.py Log1pe.forward; GPflow transforms.positive), branch structure of numpy's logaddexp
__device__ __forceinline__ double kfit_softplus(double x) {
#pragma clang fp contract(off)
  const double sp = x < 0.0 ? log1p(exp(x)) : (x == 0.0 ? 0.6931471805599453 : x + log1p(exp(-x)));
  return sp + 1e-6;
}
__device__ __forceinline__ double kfit_value(const KfFitDesc& d, int b, double x) { return d.positive[b] ? kfit_softplus(x) : x; }

// Thread -> element of the free vector.  Workgroups 0 .. G - 2 walk the big blocks (Z0, Z1, u, s of f, then of g) one element per thread;
// the LAST workgroup owns every hyperparameter element (ell, var of both latents, the noise: <= 33), which need the sums over the
// inducing rows.  Returns the block (or -1) and the index inside it.
__device__ __forceinline__ int kfit_element(const KfFitDesc& d, int& i) {
  const bool hyper_wg = blockIdx.x == gridDim.x - 1;
  const int j = hyper_wg ? (int)threadIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int nb = hyper_wg ? 9 : 8;
  const int* off = hyper_wg ? d.hyp_off : d.big_off;
  const int* blk = hyper_wg ? d.hyp_blk : d.big_blk;
  if (j >= off[nb]) { i = 0; return -1; }
  int k = 0;
  for (int q = 1; q < nb; ++q) k += (j >= off[q]) ? 1 : 0;
  i = j - off[k];
  return blk[k];
}

// value of element (block b, index i) into the parameter image: plain blocks are copied, the hyperparameter blocks fill their records of
// the hyper block Hn the NEXT step reads (the block the current step was evaluated at stays intact: two blocks, used in turn)
__device__ __forceinline__ void kfit_store_value(const KfFitArgs& a, double* Hn, int b, int i, double val) {
#pragma clang fp contract(off)
  const KfFitDesc& d = a.d;
  const int kind = b == KFIT_BLOCKS - 1 ? 8 : b % 8;      // 0 Z0, 1 Z1, 2 u, 3 s, 4 ell0, 5 ell1, 6 var0, 7 var1, 8 noise
  const int h = b / 8;
  if (kind <= 3) a.img[d.dst[b] + i] = val;
  else if (kind <= 5) { double* R = Hn + (2 * h + (kind - 4)) * KH_FAC; R[KH_ELL + i] = val; R[KH_INV + i] = 1.0 / val; }
  else if (kind <= 7) Hn[(2 * h + (kind - 6)) * KH_FAC + KH_VAR] = val;
  else Hn[KH_NOISE] = val;
}

// Grid: KFIT_THREADS-wide workgroups, one element of the free vector per thread (the update is ~300 fp64 instructions per element --
// tanh, log1p, exp, sqrt, divisions: in ONE workgroup it took 20 us of a 105 us step, VALU-bound on a single compute unit).  The blocks
// are laid out so that the few hyperparameter elements (ell, var, noise: those that need the sums over the inducing rows) come LAST
// in the element order handled by the last workgroup, which also writes the history entry and the failure record.
constexpr int KFIT_MROWS = 16 * 7;       // most inducing rows per factor on the fused path (kf_plan)
constexpr int KFIT_THREADS = 256;
__global__ void __launch_bounds__(KFIT_THREADS)
k_fit_update(KfFitArgs a) {
  const KfFitDesc& d = a.d;
  const int t = threadIdx.x;
  const double* Ho = a.img + (a.step & 1 ? d.off_hyp2 : d.off_hyp);      // the block this step's kernels read
  double* Hn = a.img + (a.update ? (a.step & 1 ? d.off_hyp : d.off_hyp2) : d.off_hyp);      // the block the next step will read
  int i = 0;
  const int b = kfit_element(d, i);
  const int e = b >= 0 ? d.off[b] + i : 0;
  if (!a.update) {                       // first launch of a call: free state -> parameter image
    if (b >= 0) kfit_store_value(a, Hn, b, i, kfit_value(d, b, a.x[e]));
    return;
  }
  __shared__ int s_fail;
  __shared__ double s_col[2][2][MAXD + 1][KFIT_MROWS];      // the columns of krow that are summed over the inducing rows
  __shared__ double s_dl[2][2][MAXD], s_dv[2][2];
  const bool hyper_wg = blockIdx.x == gridDim.x - 1;         // owns every ell / var / noise element (see the host side: those blocks are its last elements)
  if (t == 0) {
    int f = a.fail[0];
    if (f == 0) {
      const int* info = reinterpret_cast<const int*>(a.res + d.res_info);
      int bad = -1;
      for (int j = 2 * d.nlat - 1; j >= 0; --j) bad = info[2 * j] != 0 ? j : bad;      // independent loads, first failing job wins
      if (bad >= 0) { f = 1 + a.step; if (hyper_wg) { a.fail[1] = bad; a.fail[2] = info[2 * bad]; a.fail[0] = f; } }
    }
    s_fail = f;
  }
  if (hyper_wg) {   // stage the krow columns whose sums over the rows make d ell / d var (all loads in flight together) ...
    for (int idx = t; idx < 4 * (MAXD + 1) * KFIT_MROWS; idx += KFIT_THREADS) {
      const int mm = idx % KFIT_MROWS, c = (idx / KFIT_MROWS) % (MAXD + 1), q = (idx / (KFIT_MROWS * (MAXD + 1))) & 1, h = idx / (2 * KFIT_MROWS * (MAXD + 1));
      const int D = q == 0 ? d.D0 : d.D1, W = 2 + 2 * D;
      if (h < d.nlat && c <= D && mm < d.M[h][q])
        s_col[h][q][c][mm] = a.res[h * d.res_size + (q == 0 ? d.res_krow0 : d.res_krow1) + mm * W + (c == D ? 0 : 1 + D + c)];
    }
  }
  // this thread's element: every load up front
  int kind = 0, h = 0;
  double x = 0.0, mv = 0.0, vv = 0.0, gnum = 0.0, aux = 1.0;
  const double* pws = a.res + d.res_pws;
  if (b >= 0) {
    kind = b == KFIT_BLOCKS - 1 ? 8 : b % 8; h = b / 8;
    const double* R = a.res + h * d.res_size;
    x = a.x[e]; mv = a.m[e]; vv = a.v[e];
    if (kind <= 1) {
      const int q = kind, D = q == 0 ? d.D0 : d.D1, W = 2 + 2 * D, mm = i / D, dd = i - mm * D;
      aux = Ho[(2 * h + q) * KH_FAC + KH_ELL + dd];
      gnum = R[(q == 0 ? d.res_krow0 : d.res_krow1) + mm * W + 1 + dd];
    } else if (kind == 2) gnum = R[d.res_gu + i];
    else if (kind == 3) gnum = R[d.res_gs + i];
    else if (kind <= 5) aux = Ho[(2 * h + (kind - 4)) * KH_FAC + KH_ELL + i];
    else if (kind <= 7) aux = Ho[(2 * h + (kind - 6)) * KH_FAC + KH_VAR];
    else gnum = pws[1];
  }
  __syncthreads();
  if (s_fail) return;             // a Cholesky failed in this or an earlier step of the call: the state stays as it was before that step
  if (hyper_wg) {
    // ... and sum them in the host's order (m = 0, 1, ...), one thread each, from LDS
    if (t < 4 * (MAXD + 1)) {
      const int hh = t / (2 * (MAXD + 1)), q = (t / (MAXD + 1)) & 1, dd = t % (MAXD + 1), D = q == 0 ? d.D0 : d.D1;
      if (hh < d.nlat && dd <= D) {
        double sm = 0.0;
        for (int mm = 0; mm < d.M[hh][q]; ++mm) sm += s_col[hh][q][dd][mm];
        if (dd == D) s_dv[hh][q] = sm; else s_dl[hh][q][dd] = sm;
      }
    }
    if (t == 64) {    // history of the step: data term, KL from its five scalars per latent (as kronf_run assembles it)
#pragma clang fp contract(off)
      double klsum = 0.0;
      const double kl_ranks = pws[7];
      if (kl_ranks > 0.0)
        for (int hh = 0; hh < d.nlat; ++hh) {
          const double* vk = a.res + hh * d.res_size;
          const int M0 = d.M[hh][0], M1 = d.M[hh][1];
          klsum += 0.5 * (vk[0] - kl_ranks * (double)M0 * M1 - vk[1] + vk[2] + (double)M1 * vk[3] + (double)M0 * vk[4]);
        }
      a.hist[2 * a.step] = pws[0];
      a.hist[2 * a.step + 1] = klsum;
    }
    __syncthreads();
  }
  if (b < 0) return;
  {
#pragma clang fp contract(off)
    double gc;      // d ELBO / d (constrained value)
    if (kind <= 1) gc = gnum / (aux * aux);
    else if (kind <= 3 || kind == 8) gc = gnum;
    else if (kind <= 5) gc = s_dl[h][kind - 4][i] / (aux * aux * aux);
    else gc = s_dv[h][kind - 6] / aux + pws[2 + h] * Ho[(2 * h + 1 - (kind - 6)) * KH_FAC + KH_VAR];   // Knn = var0 var1 enters var_n directly (scripts/onoff.py:196-200)
    // cost = -ELBO; chain through the transform: d value / d x = sigmoid(x) for Log1pe (zigp/transforms.py)
    const double g = -(d.positive[b] ? gc * (0.5 * (1.0 + tanh(0.5 * x))) : gc);
    const double mnew = d.beta1 * mv + (1.0 - d.beta1) * g;
    const double vnew = d.beta2 * vv + (1.0 - d.beta2) * g * g;
    const double lr_t = d.lr[b] * a.lr_sq / a.lr_den;
    const double xnew = x - lr_t * mnew / (sqrt(vnew) + d.eps);
    a.m[e] = mnew; a.v[e] = vnew; a.x[e] = xnew;
    kfit_store_value(a, Hn, b, i, kfit_value(d, b, xnew));
  }
}

struct KfFitCall {          // host side of one zigp_kron_fit_steps call
  const zigp_kron_fit_opts* opts;
  double *x, *m, *v; int64_t n_free;
  int64_t t0; int n_steps; const int64_t* row_begin; int64_t batch;
  const double *Xw, *Yw;     // host batches for steps with row_begin < 0 (batch -(1 + k) of them)
  double *hist_data, *hist_kl;
};

// One ELBO step (or prediction) of the fused Kronecker path.  fit != nullptr: n_steps gradient steps with the Adam update on the device
// (p then carries the sizes only; X / Y are the resident data set).
static int kronf_run(zigp_ctx* c, const zigp_kron_params* p, const double* X, const double* Y, int64_t N, double jitter, double scale,
                     double g_offset, double f_mu, int include_kl, bool predict, double* out9, double* elbo_data, double* kl, zigp_kron_grads* grads,
                     int lik, double* d_offset, bool dev_xy, const KfFitCall* fit = nullptr) {
  // dev_xy: X / Y are DEVICE pointers into the resident data set (zigp_set_data): nothing but the parameters is staged
  const int nlat = (lik == ZIGP_LIK_ONOFF) ? 2 : 1;
  const KfPlan pl = kf_plan(p, nlat);
  if (!c->kronf) { c->kronf = new (std::nothrow) KfState(); c->kronf_free = kf_free; if (!c->kronf) { c->err = "out of memory"; return ZIGP_EHIP; } }
  KfState& ks = *c->kronf;
  ZIGP_TRY(begin_staged_call(c));
  const bool need_grad = (grads != nullptr || fit != nullptr) && !predict;
  const int D0 = p->D0, D1 = p->D1, ldx = D0 + D1;
  const int64_t Npad = std::max<int64_t>(1024, round_up(N, 1024));
  KfHostLatent hl[2] = {{{p->M0f, p->M1f}, {p->Z0f, p->Z1f}, {p->ell0f, p->ell1f}, {p->var0f, p->var1f}, p->u_fm, p->u_fs_sqrt},
                        {{p->M0g, p->M1g}, {p->Z0g, p->Z1g}, {p->ell0g, p->ell1g}, {p->var0g, p->var1g}, p->u_gm, p->u_gs_sqrt}};
  if (nlat == 1) hl[1] = hl[0];
  // ---- sizes of this variant
  const int R0 = 16 * pl.nb0c, R1 = 16 * pl.nb1c;
  const KfWork wl = kf_work_layout(pl.nb0c, pl.nb1c);
  const int nblk = kf_nblocks(pl.nb0c, pl.nb1c);
  const size_t r01 = (size_t)R0 * R1;
  // per-factor region of KfState::mat: K [128][128] | P | PF | dvec ; per-latent: U S2 T0 T1 Al | AlF S2F AlTF S2TF | work | scratch
  const size_t rmax = (size_t)wl.ldw * wl.ldw;
  const size_t FAC_P = (size_t)PB * PB, FAC_PF = FAC_P + rmax, FAC_DV = FAC_PF + rmax, FAC_ZS = FAC_DV + wl.ldw + 8, FAC_SIZE = FAC_ZS + (size_t)wl.ldw * MAXD;
  const size_t LAT_U = 0, LAT_S2 = r01, LAT_T0 = 2 * r01, LAT_T1 = 3 * r01, LAT_AL = 4 * r01, LAT_ALF = 5 * r01, LAT_S2F = 6 * r01, LAT_ALTF = 7 * r01,
               LAT_S2TF = 8 * r01, LAT_WORK = 9 * r01, LAT_SCR = LAT_WORK + wl.total, SCR_SET = 3 * rmax + r01 + wl.ldw, LAT_SIZE = LAT_SCR + 2 * SCR_SET;
  // result block: [latent f][latent g][pws: 8 sums over points][4 Cholesky status slots (8 doubles)]; everything in front of the status
  // slots is summed over ranks in a data-parallel run; behind them, not downloaded: the point-wise block partials
  const size_t RES_KLV = 0, RES_KROW0 = 8, RES_KROW1 = RES_KROW0 + (size_t)R0 * KF_KROW_W, RES_GU = RES_KROW1 + (size_t)R1 * KF_KROW_W,
               RES_GS = RES_GU + r01, RES_SIZE = RES_GS + r01;
  // ---- the staged image (one host -> device copy): the PARAMETER IMAGE -- per latent Z0, Z1, u, s, then the hyperparameter block (KH_*) --
  // followed by X, Y of a host minibatch.  In a fit call the parameter image is written by k_fit_update instead.
  size_t off = 0;
  size_t off_z[2][2], off_u[2], off_s[2];
  for (int h = 0; h < nlat; ++h) {
    off_z[h][0] = off; off += (size_t)hl[h].M[0] * D0;
    off_z[h][1] = off; off += (size_t)hl[h].M[1] * D1;
    off_u[h] = off; off += (size_t)hl[h].M[0] * hl[h].M[1];
    off_s[h] = off; off += (size_t)hl[h].M[0] * hl[h].M[1];
  }
  const size_t off_hyp = off; off += KH_SIZE;
  const size_t off_hyp2 = off; off += KH_SIZE;      // second hyperparameter block (fit calls: step i reads block i % 2, its update writes the other)
  const size_t n_par = off;
  const size_t off_x = n_par, off_y = dev_xy ? off_x : off_x + (size_t)N * ldx;
  const size_t n_in = dev_xy ? n_par : off_y + (size_t)N;
  const size_t n_wrap = fit ? [&] { size_t k = 0; for (int i = 0; i < fit->n_steps; ++i) if (fit->row_begin[i] < 0) k = std::max<size_t>(k, (size_t)(-fit->row_begin[i])); return k; }() : 0;
  ZIGP_ENSURE(c, ks.in, n_in + n_wrap * (size_t)N * (ldx + 1));
  double* d_hyp = ks.in.p + off_hyp;          // the hyperparameter block the next enqueue()'s kernels read
  double* const d_wrap = ks.in.p + n_in;      // fit: host batches [k][N][ldx], then their Y [k][N]
  if (!fit) {
    ZIGP_PINNED(c, hin, n_in);
    if (!dev_xy) {
      memcpy(hin + off_x, X, sizeof(double) * N * ldx);
      if (Y) memcpy(hin + off_y, Y, sizeof(double) * N); else memset(hin + off_y, 0, sizeof(double) * N);
    }
    double* H = hin + off_hyp;
    memset(H, 0, sizeof(double) * KH_SIZE);
    for (int h = 0; h < nlat; ++h) {
      memcpy(hin + off_z[h][0], hl[h].Z[0], sizeof(double) * hl[h].M[0] * D0);
      memcpy(hin + off_z[h][1], hl[h].Z[1], sizeof(double) * hl[h].M[1] * D1);
      memcpy(hin + off_u[h], hl[h].u, sizeof(double) * hl[h].M[0] * hl[h].M[1]);
      memcpy(hin + off_s[h], hl[h].s, sizeof(double) * hl[h].M[0] * hl[h].M[1]);
      for (int q = 0; q < 2; ++q) {
        const int M = hl[h].M[q], D = q == 0 ? D0 : D1;
        double* Rh = H + (2 * h + q) * KH_FAC;
        for (int d = 0; d < MAXD; ++d) {      // (KH_ZC, the centre of the moment sums, is filled in by k_kf_factor)
          Rh[KH_INV + d] = d < D ? 1.0 / hl[h].ell[q][d] : 0.0;
          Rh[KH_ELL + d] = d < D ? hl[h].ell[q][d] : 0.0;
        }
        Rh[KH_VAR] = hl[h].var[q];
        (void)M;
      }
    }
    H[KH_NOISE] = p->noise;
    ZIGP_HIP(c, hipMemcpyAsync(ks.in.p, hin, sizeof(double) * n_in, hipMemcpyHostToDevice, c->stream));
  }
  if (predict) ZIGP_HIP(c, hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream));
  ZIGP_ENSURE(c, ks.mat, (size_t)4 * FAC_SIZE + 2 * LAT_SIZE);
  const int pw_blocks = (int)(Npad / PW_THREADS);
  const size_t pts_lat = (size_t)8 * Npad;   // part[4], gm, gv, dq0, dq1
  ZIGP_ENSURE(c, ks.pts, 2 * pts_lat);
  const size_t RES_PWS = (size_t)2 * RES_SIZE, RES_INFO = RES_PWS + 8, n_res = RES_INFO + 8;
  ZIGP_ENSURE(c, ks.res, n_res + (size_t)pw_blocks * KPW_ACC);
  auto fac = [&](int h, int q) { return ks.mat.p + (size_t)(2 * h + q) * FAC_SIZE; };
  auto lat = [&](int h) { return ks.mat.p + (size_t)4 * FAC_SIZE + (size_t)h * LAT_SIZE; };
  auto pts = [&](int h) { return ks.pts.p + (size_t)h * pts_lat; };
  auto res = [&](int h) { return ks.res.p + (size_t)h * RES_SIZE; };
  double* d_pwacc = ks.res.p + n_res;

  int Mq[2][2];
  for (int h = 0; h < nlat; ++h)
    for (int q = 0; q < 2; ++q) Mq[h][q] = (int)round_up(hl[h].M[q], 16);
  bool large_exact = pl.large;   // every latent has exactly the capacity's block counts: the instantiation with compile-time bounds
  for (int h = 0; h < nlat; ++h) large_exact = large_exact && Mq[h][0] == 16 * pl.nb0c && Mq[h][1] == 16 * pl.nb1c;
  // LDS of the larger-grid point kernels: fragment images packed by the actual block counts (max over the latents)
  size_t lds_fwd = 0, lds_bwd = 0;
  for (int h = 0; h < nlat; ++h) {
    const size_t b0 = Mq[h][0] / 16, b1 = Mq[h][1] / 16;
    lds_fwd = std::max(lds_fwd, sizeof(double) * 256 * (b0 * b0 + b1 * b1 + 2 * b0 * b1));
    lds_bwd = std::max(lds_bwd, sizeof(double) * 256 * (b0 * b0 + b1 * b1 + 4 * b0 * b1));
  }
  static bool attr_set = false;
  if (!attr_set) {
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kf_factor), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * PB * PBLD)));
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kf_finish), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KF_FIN_LDS));
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kfl_forward<1, 7, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kfl_backward<1, 7, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kfl_forward<1, 7, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kfl_backward<1, 7, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kfl_latent), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kfl_latent_lds(16 * 7)));
    attr_set = true;
  }
  int* hinfo = nullptr;

  // ---- the launches of ONE step on rows (Xd, Yd) -- everything else it reads lives on the device (parameter image, hyper block)
  auto enqueue = [&](const double* Xd, const double* Yd) -> int {
    if (c->comm && !predict) ZIGP_HIP(c, hipMemsetAsync(ks.res.p, 0, sizeof(double) * RES_INFO, c->stream));   // parts a value-only / single-latent step leaves unwritten
    // ---- factor stage
    {
      KfFactorArgs fa;
      memset(&fa, 0, sizeof(fa));
      for (int h = 0; h < nlat; ++h)
        for (int q = 0; q < 2; ++q) {
          KfFactorJob& jb = fa.job[2 * h + q];
          jb.Z = ks.in.p + off_z[h][q]; jb.M = hl[h].M[q]; jb.D = q == 0 ? D0 : D1; jb.Mq = Mq[h][q];
          jb.hyp = d_hyp + (2 * h + q) * KH_FAC; jb.zc_out = d_hyp + (2 * h + q) * KH_FAC + KH_ZC;
          jb.K = fac(h, q); jb.P = fac(h, q) + FAC_P; jb.PF = fac(h, q) + FAC_PF; jb.dvec = fac(h, q) + FAC_DV; jb.Zs = fac(h, q) + FAC_ZS;
        }
      fa.jitter = jitter; fa.piv_rtol = c->pivot_rtol;
      if (predict) { fa.info = c->d_info; fa.own_slots = 0; }
      else { fa.info = reinterpret_cast<int*>(ks.res.p + RES_INFO); fa.own_slots = 1; }
      hipLaunchKernelGGL(k_kf_factor, dim3(2 * nlat), dim3(1024), sizeof(double) * PB * PBLD, c->stream, fa);
      ZIGP_HIP(c, hipGetLastError());
    }
    if (predict) ZIGP_TRY(request_info(c, &hinfo));   // value / gradient steps: the status slots come back with the result block
    {
      KfLatentArgs la;
      memset(&la, 0, sizeof(la));
      for (int h = 0; h < nlat; ++h) {
        KfLatentJob& jb = la.job[h];
        jb.M0 = hl[h].M[0]; jb.M1 = hl[h].M[1]; jb.Mq0 = Mq[h][0]; jb.Mq1 = Mq[h][1];
        jb.P0 = fac(h, 0) + FAC_P; jb.P1 = fac(h, 1) + FAC_P; jb.dvec0 = fac(h, 0) + FAC_DV; jb.dvec1 = fac(h, 1) + FAC_DV;
        jb.PF0 = fac(h, 0) + FAC_PF; jb.PF1 = fac(h, 1) + FAC_PF;
        jb.u = ks.in.p + off_u[h]; jb.s = ks.in.p + off_s[h];
        double* L = lat(h);
        jb.U = L + LAT_U; jb.S2 = L + LAT_S2; jb.T0 = L + LAT_T0; jb.T1 = L + LAT_T1; jb.Al = L + LAT_AL;
        jb.AlF = L + LAT_ALF; jb.S2F = L + LAT_S2F; jb.AlTF = L + LAT_ALTF; jb.S2TF = L + LAT_S2TF;
        jb.klv = res(h) + RES_KLV;
      }
      if (pl.large) {
        int mq1 = 0;
        for (int h = 0; h < nlat; ++h) mq1 = std::max(mq1, la.job[h].Mq1);
        hipLaunchKernelGGL(k_kfl_latent, dim3(nlat), dim3(1024), kfl_latent_lds(mq1), c->stream, la);
      }
      else hipLaunchKernelGGL(k_kf_latent, dim3(nlat), dim3(1024), 0, c->stream, la);
      ZIGP_HIP(c, hipGetLastError());
    }
    // ---- point stage
    KfArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.X = Xd; ka.N = N; ka.Npad = Npad; ka.ldx = ldx; ka.ntiles = (int)(Npad / 16);
    for (int h = 0; h < nlat; ++h) {
      KfLat& L = ka.lat[h];
      for (int q = 0; q < 2; ++q) {
        KfFac& f = L.f[q];
        f.M = hl[h].M[q]; f.nb = Mq[h][q] / 16; f.D = q == 0 ? D0 : D1; f.col0 = q == 0 ? 0 : D0;
        f.hyp = d_hyp + (2 * h + q) * KH_FAC;
        f.Zs = fac(h, q) + FAC_ZS; f.PF = fac(h, q) + FAC_PF;
      }
      double* Lm = lat(h);
      L.AlF = Lm + LAT_ALF; L.S2F = Lm + LAT_S2F; L.AlTF = Lm + LAT_ALTF; L.S2TF = Lm + LAT_S2TF;
      double* P = pts(h);
      L.part = P; L.gm = P + 4 * Npad; L.gv = P + 5 * Npad; L.dq0 = P + 6 * Npad; L.dq1 = P + 7 * Npad;
    }
    {
      const int waves = std::min(ka.ntiles, pl.large ? 1024 / nlat : 2048);
      ka.tpw = (ka.ntiles + waves - 1) / waves;
      const int nw = (ka.ntiles + ka.tpw - 1) / ka.tpw;
      const dim3 grid((nw + KF_WAVES - 1) / KF_WAVES, nlat);
      if (pl.large && large_exact) hipLaunchKernelGGL((k_kfl_forward<1, 7, true, true>), grid, dim3(64 * KF_WAVES), lds_fwd, c->stream, ka);
      else if (pl.large) hipLaunchKernelGGL((k_kfl_forward<1, 7, true, false>), grid, dim3(64 * KF_WAVES), lds_fwd, c->stream, ka);
      else ZIGP_TRY(kf_launch_small_latents(c, Mq, nlat, false, grid.x, ka));
      ZIGP_HIP(c, hipGetLastError());
    }
    KronPwArgs a;
    memset(&a, 0, sizeof(a));
    const int gl_ = nlat - 1;   // latent whose buffers stand in for g (unused by the single-latent kernels)
    a.part_f = pts(0); a.part_g = pts(gl_); a.Y = Yd; a.N = N; a.Nc = Npad;
    a.hyp = d_hyp;              // knn_f, knn_g, noise: from the device block (the single-latent heads read the by-value fields below)
    if (!fit) { a.knn_f = p->var0f * p->var1f; a.knn_g = p->var0g * p->var1g; a.noise = p->noise; }
    a.g_offset = g_offset; a.f_offset = f_mu; a.scale = scale;
    a.gm_f = need_grad ? pts(0) + 4 * Npad : nullptr; a.gv_f = pts(0) + 5 * Npad; a.gm_g = pts(gl_) + 4 * Npad; a.gv_g = pts(gl_) + 5 * Npad;
    a.dq0_f = pts(0) + 6 * Npad; a.dq1_f = pts(0) + 7 * Npad; a.dq0_g = pts(gl_) + 6 * Npad; a.dq1_g = pts(gl_) + 7 * Npad;
    a.acc = d_pwacc; a.out9 = nullptr; a.ld9 = N;
    if (predict) {
      const int rows = nlat == 2 ? 9 : 4;
      ZIGP_ENSURE(c, ks.out, (size_t)rows * N);
      a.out9 = ks.out.p;
      if (nlat == 2) hipLaunchKernelGGL(k_kron_pointwise<true>, dim3(pw_blocks), dim3(PW_THREADS), 0, c->stream, a);
      else hipLaunchKernelGGL(k_kron_head_pointwise<true>, dim3(pw_blocks), dim3(PW_THREADS), 0, c->stream, a, lik);
      ZIGP_HIP(c, hipGetLastError());
      return 0;
    }
    if (nlat == 2) hipLaunchKernelGGL(k_kron_pointwise<false>, dim3(pw_blocks), dim3(PW_THREADS), 0, c->stream, a);
    else hipLaunchKernelGGL(k_kron_head_pointwise<false>, dim3(pw_blocks), dim3(PW_THREADS), 0, c->stream, a, lik);
    if (!need_grad) hipLaunchKernelGGL(k_kron_pw_reduce, dim3(1), dim3(256), 0, c->stream, d_pwacc, pw_blocks, include_kl ? 1.0 : 0.0, ks.res.p + RES_PWS);
    ZIGP_HIP(c, hipGetLastError());
    if (need_grad) {
      const int waves = std::min(ka.ntiles, 1024 / nlat);   // one wave per SIMD of the chip: the kernels hold ~400-500 registers per lane
      ka.tpw = (ka.ntiles + waves - 1) / waves;
      const int nw = (ka.ntiles + ka.tpw - 1) / ka.tpw;
      const int nwg = (nw + KF_WAVES - 1) / KF_WAVES;
      int nparts;
      if (!pl.large) {
        nparts = nwg * KF_WAVES;
        ZIGP_ENSURE(c, ks.acc, (size_t)2 * nparts * nblk * 256);
        for (int h = 0; h < nlat; ++h) ka.lat[h].acc = ks.acc.p + (size_t)h * nparts * nblk * 256;
        ZIGP_TRY(kf_launch_small_latents(c, Mq, nlat, true, (unsigned)nwg, ka));
      } else {
        // The backward kernel spills ~4 KB of operands per point and latent (at 10 x 100), which k_kfl_accum then sums over the points.  The
        // rows go through in RANGES of <= 1024 tiles (16 384 rows: <= 128 MB of records for both latents, Infinity-Cache resident), one
        // backward + one accumulate launch per range.  The split of the tiles into accumulation parts of tps tiles does not depend on the
        // ranges (a range is a whole number of parts), so the partial sums -- and k_kf_reduce's fixed-order total -- are the same bits as
        // with one range over everything; memory no longer grows with the row count (rounds 2-3: 0.86 GB for the pptr full batch, and a
        // 64 GB cap with an error beyond it).
        size_t rec[2] = {0, 0};
        for (int h = 0; h < nlat; ++h) rec[h] = (size_t)64 * (Mq[h][0] + Mq[h][1]);
        const int tps = std::max(4, std::min(64, ka.ntiles / 16));     // tiles per accumulation part (4 for a minibatch: the chain of dependent loads is short)
        nparts = (ka.ntiles + tps - 1) / tps;
        const int range_tiles = std::max(1, c->kron_range_tiles / tps) * tps;
        const int spill_tiles = std::min(ka.ntiles, range_tiles);
        ZIGP_ENSURE(c, ks.spill, (rec[0] + (nlat == 2 ? rec[1] : 0)) * spill_tiles);
        ka.lat[0].spill = ks.spill.p;
        if (nlat == 2) ka.lat[1].spill = ks.spill.p + rec[0] * spill_tiles;
        ZIGP_ENSURE(c, ks.acc, (size_t)2 * nparts * nblk * 256);
        KflAccArgs aa;
        memset(&aa, 0, sizeof(aa));
        aa.X = ka.X; aa.N = N; aa.ldx = ldx; aa.ntiles = ka.ntiles; aa.tps = tps; aa.nb0c = pl.nb0c; aa.nb1c = pl.nb1c;
        for (int h = 0; h < nlat; ++h) {
          KflAccLat& L = aa.lat[h];
          L.spill = ka.lat[h].spill; L.gm = ka.lat[h].gm; L.gv = ka.lat[h].gv;
          L.acc = ks.acc.p + (size_t)h * nparts * nblk * 256;
          ka.lat[h].acc = L.acc;
          L.nb0 = Mq[h][0] / 16; L.nb1 = Mq[h][1] / 16; L.D0 = D0; L.D1 = D1;
          L.hyp0 = d_hyp + (2 * h) * KH_FAC; L.hyp1 = d_hyp + (2 * h + 1) * KH_FAC;
        }
        for (int t0 = 0; t0 < ka.ntiles; t0 += range_tiles) {
          const int t1 = std::min(t0 + range_tiles, ka.ntiles), nt = t1 - t0;
          const int rwaves = std::min(nt, 1024 / nlat);
          ka.tile0 = t0; ka.tile1 = t1; ka.tpw = (nt + rwaves - 1) / rwaves;
          const int rnw = (nt + ka.tpw - 1) / ka.tpw, rnwg = (rnw + KF_WAVES - 1) / KF_WAVES;
          if (large_exact) hipLaunchKernelGGL((k_kfl_backward<1, 7, true, true>), dim3(rnwg, nlat), dim3(64 * KF_WAVES), lds_bwd, c->stream, ka);
          else hipLaunchKernelGGL((k_kfl_backward<1, 7, true, false>), dim3(rnwg, nlat), dim3(64 * KF_WAVES), lds_bwd, c->stream, ka);
          ZIGP_HIP(c, hipGetLastError());
          aa.tile0 = t0; aa.tile1 = t1; aa.part0 = t0 / tps;
          hipLaunchKernelGGL(k_kfl_accum, dim3((nblk + 3) / 4, (nt + tps - 1) / tps, nlat), dim3(256), 0, c->stream, aa);
          ZIGP_HIP(c, hipGetLastError());
        }
      }
      hipLaunchKernelGGL(k_kf_reduce, dim3(nblk * 256 / 16 + 1, nlat), dim3(256), 0, c->stream, ka.lat[0].acc, ka.lat[gl_].acc, nparts,
                         lat(0) + LAT_WORK, lat(gl_) + LAT_WORK, pl.nb0c, pl.nb1c, Mq[0][0] / 16, Mq[0][1] / 16, Mq[gl_][0] / 16, Mq[gl_][1] / 16,
                         d_pwacc, pw_blocks, include_kl ? 1.0 : 0.0, ks.res.p + RES_PWS);
      ZIGP_HIP(c, hipGetLastError());
      KflFinishArgs fa;
      memset(&fa, 0, sizeof(fa));
      for (int h = 0; h < nlat; ++h) {
        KfFinishJob& jb = fa.job[h];
        jb.M0 = hl[h].M[0]; jb.M1 = hl[h].M[1]; jb.Mq0 = Mq[h][0]; jb.Mq1 = Mq[h][1]; jb.D0 = D0; jb.D1 = D1;
        jb.P0 = fac(h, 0) + FAC_P; jb.P1 = fac(h, 1) + FAC_P; jb.dvec0 = fac(h, 0) + FAC_DV; jb.dvec1 = fac(h, 1) + FAC_DV;
        jb.PF0 = fac(h, 0) + FAC_PF; jb.PF1 = fac(h, 1) + FAC_PF;
        jb.K0 = fac(h, 0); jb.K1 = fac(h, 1); jb.Z0 = ks.in.p + off_z[h][0]; jb.Z1 = ks.in.p + off_z[h][1];
        jb.hyp0 = d_hyp + (2 * h) * KH_FAC; jb.hyp1 = d_hyp + (2 * h + 1) * KH_FAC;
        double* L = lat(h);
        jb.U = L + LAT_U; jb.S2 = L + LAT_S2; jb.T0 = L + LAT_T0; jb.T1 = L + LAT_T1; jb.Al = L + LAT_AL; jb.s = ks.in.p + off_s[h];
        jb.work = L + LAT_WORK;
        jb.krow0 = res(h) + RES_KROW0; jb.krow1 = res(h) + RES_KROW1; jb.gu = res(h) + RES_GU; jb.gs = res(h) + RES_GS;
      }
      fa.jitter = jitter; fa.with_kl = include_kl ? 1 : 0;
      fa.ldw = wl.ldw; fa.wS2 = wl.S2; fa.wP0 = wl.P0; fa.wP1 = wl.P1; fa.wK0 = wl.K0; fa.wK1 = wl.K1;
      fa.scratch_off = (int64_t)wl.total; fa.scratch_set = (int64_t)SCR_SET;
      if (pl.large) {
        int nbmax = 1;
        for (int h = 0; h < nlat; ++h) nbmax = std::max(nbmax, std::max(Mq[h][0], Mq[h][1]) / 16);
        const dim3 grid(nbmax, 2, nlat);
        hipLaunchKernelGGL(k_kfl_finish<1>, grid, dim3(KFL_FIN_THREADS), 0, c->stream, fa);
        hipLaunchKernelGGL(k_kfl_finish<2>, grid, dim3(KFL_FIN_THREADS), 0, c->stream, fa);
        hipLaunchKernelGGL(k_kfl_finish<3>, grid, dim3(KFL_FIN_THREADS), 0, c->stream, fa);
      } else {
        KfFinishArgs fs;
        memset(&fs, 0, sizeof(fs));
        fs.job[0] = fa.job[0]; fs.job[1] = fa.job[1]; fs.jitter = fa.jitter; fs.with_kl = fa.with_kl;
        hipLaunchKernelGGL(k_kf_finish, dim3(2, nlat), dim3(1024), KF_FIN_LDS, c->stream, fs);
      }
      ZIGP_HIP(c, hipGetLastError());
    }
    if (c->comm) {   // data-parallel run: the block (cleared at the start of the step) is summed over ranks where it lies
      // a rank that does not count the KL (include_kl = 0) must not add its scalars, which the latent kernel writes regardless
      if (!include_kl) for (int h = 0; h < nlat; ++h) ZIGP_HIP(c, hipMemsetAsync(res(h) + RES_KLV, 0, sizeof(double) * 8, c->stream));
      ZIGP_TRY(comm_allreduce(c, ks.res.p, RES_INFO));
    }
    return 0;
  };

  static const char* const fac_names[4] = {"Kronecker factor 0 of Kuu (latent f)", "Kronecker factor 1 of Kuu (latent f)",
                                           "Kronecker factor 0 of Kuu (latent g)", "Kronecker factor 1 of Kuu (latent g)"};
  // =========================================================================================================================
  // fit call: n_steps steps back to back, parameters updated on the device, ONE synchronisation at the end
  // =========================================================================================================================
  if (fit) {
    KfFitArgs fa;
    memset(&fa, 0, sizeof(fa));
    KfFitDesc& d = fa.d;
    d.nlat = nlat; d.D0 = D0; d.D1 = D1;
    size_t o = 0;
    for (int h = 0; h < 2; ++h) {
      const int M0 = hl[h].M[0], M1 = hl[h].M[1];
      const int sizes[8] = {M0 * D0, M1 * D1, M0 * M1, M0 * M1, D0, D1, 1, 1};
      const size_t dsts[8] = {off_z[h][0], off_z[h][1], off_u[h], off_s[h], 0, 0, 0, 0};
      for (int k = 0; k < 8; ++k) {
        const int b = 8 * h + k;
        d.off[b] = (int)o; d.n[b] = sizes[k]; d.dst[b] = (int)dsts[k]; d.positive[b] = fit->opts->positive[b]; d.lr[b] = fit->opts->lr[b];
        o += sizes[k];
      }
      for (int q = 0; q < 2; ++q) { d.M[h][q] = hl[h].M[q]; d.Mq[h][q] = Mq[h][q]; d.off_z[h][q] = (int)off_z[h][q]; }
    }
    d.off[16] = (int)o; d.n[16] = 1; d.dst[16] = 0; d.positive[16] = fit->opts->positive[16]; d.lr[16] = fit->opts->lr[16];
    o += 1;
    if ((int64_t)o != fit->n_free) { c->err = "zigp_kron_fit_steps: n_free does not match the model sizes"; return ZIGP_EARG; }
    d.off_hyp = (int)off_hyp; d.off_hyp2 = (int)off_hyp2;
    {
      int nbig = 0, nhyp = 0, kb = 0, kh = 0;
      for (int b = 0; b < KFIT_BLOCKS; ++b) {
        const bool big = b != 16 && b % 8 <= 3;
        if (big) { d.big_blk[kb] = b; d.big_off[kb++] = nbig; nbig += d.n[b]; }
        else { d.hyp_blk[kh] = b; d.hyp_off[kh++] = nhyp; nhyp += d.n[b]; }
      }
      d.big_off[8] = nbig; d.hyp_off[9] = nhyp;
    }
    d.res_size = (int)RES_SIZE; d.res_krow0 = (int)RES_KROW0; d.res_krow1 = (int)RES_KROW1; d.res_gu = (int)RES_GU; d.res_gs = (int)RES_GS;
    d.res_pws = (int)RES_PWS; d.res_info = (int)RES_INFO;
    d.beta1 = fit->opts->beta1; d.beta2 = fit->opts->beta2; d.eps = fit->opts->eps;
    const size_t nf = (size_t)fit->n_free, n_hist = (size_t)2 * fit->n_steps;
    const size_t n_state = 3 * nf + n_hist + 8;
    ZIGP_ENSURE(c, ks.fit, n_state);
    fa.x = ks.fit.p; fa.m = fa.x + nf; fa.v = fa.m + nf; fa.hist = fa.v + nf; fa.fail = reinterpret_cast<int*>(fa.hist + n_hist);
    fa.img = ks.in.p; fa.res = ks.res.p;
    {
      ZIGP_PINNED(c, hst, n_state);
      memcpy(hst, fit->x, sizeof(double) * nf); memcpy(hst + nf, fit->m, sizeof(double) * nf); memcpy(hst + 2 * nf, fit->v, sizeof(double) * nf);
      memset(hst + 3 * nf, 0, sizeof(double) * (n_hist + 8));
      ZIGP_HIP(c, hipMemcpyAsync(ks.fit.p, hst, sizeof(double) * n_state, hipMemcpyHostToDevice, c->stream));
      if (n_wrap) {
        const size_t nw = n_wrap * (size_t)N * (ldx + 1);
        ZIGP_PINNED(c, hw, nw);
        memcpy(hw, fit->Xw, sizeof(double) * n_wrap * N * ldx);
        memcpy(hw + n_wrap * (size_t)N * ldx, fit->Yw, sizeof(double) * n_wrap * N);
        ZIGP_HIP(c, hipMemcpyAsync(d_wrap, hw, sizeof(double) * nw, hipMemcpyHostToDevice, c->stream));
      }
    }
    ZIGP_HIP(c, hipMemsetAsync(ks.in.p + off_hyp, 0, sizeof(double) * 2 * KH_SIZE, c->stream));
    const dim3 ugrid((d.big_off[8] + KFIT_THREADS - 1) / KFIT_THREADS + 1);
    fa.update = 0; fa.step = 0;
    hipLaunchKernelGGL(k_fit_update, ugrid, dim3(KFIT_THREADS), 0, c->stream, fa);       // free state -> parameter image (hyper block 0)
    ZIGP_HIP(c, hipGetLastError());
    fa.update = 1;
    for (int i = 0; i < fit->n_steps; ++i) {
      const int64_t rb = fit->row_begin[i];
      const double* Xd = rb >= 0 ? X + rb * ldx : d_wrap + (size_t)(-rb - 1) * N * ldx;
      const double* Yd = rb >= 0 ? Y + rb : d_wrap + n_wrap * (size_t)N * ldx + (size_t)(-rb - 1) * N;
      d_hyp = ks.in.p + ((i & 1) ? off_hyp2 : off_hyp);       // written by the previous step's update (step 0: by the launch above)
      ZIGP_TRY(enqueue(Xd, Yd));
      // lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), t counted from 1 (zigp/optim.py AdamGroups.step): the two t-dependent factors from the host
      const double t = (double)(fit->t0 + i + 1);
      fa.step = i; fa.lr_sq = std::sqrt(1.0 - std::pow(d.beta2, t)); fa.lr_den = 1.0 - std::pow(d.beta1, t);
      hipLaunchKernelGGL(k_fit_update, ugrid, dim3(KFIT_THREADS), 0, c->stream, fa);
      ZIGP_HIP(c, hipGetLastError());
    }
    double* hst = nullptr;
    ZIGP_TRY(download(c, ks.fit.p, n_state, &hst));
    ZIGP_HIP(c, hipStreamSynchronize(c->stream));
    const int* hfail = reinterpret_cast<const int*>(hst + 3 * nf + n_hist);
    memcpy(fit->x, hst, sizeof(double) * nf); memcpy(fit->m, hst + nf, sizeof(double) * nf); memcpy(fit->v, hst + 2 * nf, sizeof(double) * nf);
    const int done = hfail[0] ? hfail[0] - 1 : fit->n_steps;       // steps whose update was applied
    c->fit_steps_applied = done;                                   // (zigp_kron_fit_steps_applied: the caller's iteration count advances by this)
    for (int i = 0; i < fit->n_steps; ++i) {
      if (fit->hist_data) fit->hist_data[i] = i < done ? hst[3 * nf + 2 * i] : NAN;
      if (fit->hist_kl) fit->hist_kl[i] = i < done ? hst[3 * nf + 2 * i + 1] : NAN;
    }
    if (hfail[0]) {
      char b[256];
      snprintf(b, sizeof(b), "Cholesky failed in step %d of this zigp_kron_fit_steps call (iteration %lld): %s not positive definite at pivot %d; "
               "the state returned is the one before that step", hfail[0] - 1, (long long)(fit->t0 + hfail[0] - 1), fac_names[hfail[1] & 3], hfail[2]);
      c->err = b; c->info = hfail[2];
      return ZIGP_ENOTPD;
    }
    return 0;
  }

  ZIGP_TRY(enqueue(dev_xy ? X : ks.in.p + off_x, Y ? (dev_xy ? Y : ks.in.p + off_y) : nullptr));
  if (predict) {
    const int rows = nlat == 2 ? 9 : 4;
    ZIGP_HIP(c, hipMemcpyAsync(out9, ks.out.p, sizeof(double) * rows * N, hipMemcpyDeviceToHost, c->stream));
    ZIGP_HIP(c, hipStreamSynchronize(c->stream));
    return info_result(c, hinfo, "a Kronecker factor of Kuu");
  }
  double* hres = nullptr;
  ZIGP_TRY(download(c, ks.res.p, n_res, &hres));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  for (int j = 0; j < 2 * nlat; ++j) ZIGP_TRY(info_result(c, reinterpret_cast<const int*>(hres + RES_INFO) + 2 * j, fac_names[j]));
  const double* pws = hres + RES_PWS;   // var_exp, d noise, sum gv_f, sum gv_g, sum gm_f (= d / d f_mu)
  const double s_ve = pws[0], s_dn = pws[1], s_gv[2] = {pws[2], pws[3]};
  if (elbo_data) *elbo_data = s_ve;
  if (d_offset) *d_offset = pws[4];
  // KL from its five scalars; in a data-parallel run they are the sums over the ranks that count it (pws[7] of them: one, by the
  // contract of zigp_comm_init), and every rank returns the same value
  double klsum = 0.0;
  const double kl_ranks = pws[7];
  if (kl_ranks > 0.0) {
    for (int h = 0; h < nlat; ++h) {
      const double* v = hres + (size_t)h * RES_SIZE + RES_KLV;
      const int M0 = hl[h].M[0], M1 = hl[h].M[1];
      klsum += 0.5 * (v[0] - kl_ranks * (double)M0 * M1 - v[1] + v[2] + (double)M1 * v[3] + (double)M0 * v[4]);
    }
  }
  if (kl) *kl = klsum;
  if (need_grad) {
    double* gZ[2][2] = {{grads->Z0f, grads->Z1f}, {grads->Z0g, grads->Z1g}};
    double* gl[2][2] = {{grads->ell0f, grads->ell1f}, {grads->ell0g, grads->ell1g}};
    double gvar[2][2] = {{0, 0}, {0, 0}};
    double* gu[2] = {grads->u_fm, grads->u_gm};
    double* gs[2] = {grads->u_fs_sqrt, grads->u_gs_sqrt};
    for (int h = 0; h < nlat; ++h) {
      const double* R = hres + (size_t)h * RES_SIZE;
      for (int q = 0; q < 2; ++q) {
        const int D = q == 0 ? D0 : D1, W = 2 + 2 * D, M = hl[h].M[q];
        const double* krow = R + (q == 0 ? RES_KROW0 : RES_KROW1);
        const double* ell = hl[h].ell[q];
        double dv = 0.0;
        std::vector<double> dl(D, 0.0);
        for (int m = 0; m < M; ++m) {
          const double* r = &krow[(size_t)m * W];
          dv += r[0];
          for (int d = 0; d < D; ++d) {
            if (gZ[h][q]) gZ[h][q][m * D + d] = r[1 + d] / (ell[d] * ell[d]);
            dl[d] += r[1 + D + d];
          }
        }
        for (int d = 0; d < D; ++d)
          if (gl[h][q]) gl[h][q][d] = dl[d] / (ell[d] * ell[d] * ell[d]);
        gvar[h][q] = dv / hl[h].var[q] + s_gv[h] * hl[h].var[1 - q];   // Knn = var0 * var1 enters var_n directly (scripts/onoff.py:196-200)
      }
      const size_t ng = (size_t)hl[h].M[0] * hl[h].M[1];
      if (gu[h]) memcpy(gu[h], R + RES_GU, sizeof(double) * ng);
      if (gs[h]) memcpy(gs[h], R + RES_GS, sizeof(double) * ng);
    }
    grads->var0f = gvar[0][0]; grads->var1f = gvar[0][1];
    grads->var0g = nlat == 2 ? gvar[1][0] : 0.0; grads->var1g = nlat == 2 ? gvar[1][1] : 0.0;
    grads->noise = s_dn;
  }
  return 0;
}

}  // namespace zigp
