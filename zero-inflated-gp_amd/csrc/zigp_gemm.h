// fp64 GEMM core for gfx950 (MI355X / CDNA4) -- the contraction engine behind every O(M^2 N) and O(M^3)
// product on the zero-inflated-GP ELBO path (replaces TF's MatMul / MatrixTriangularSolve call sites,
// onofftf/main.py:271,284,287 and their tf.gradients twins, scripts/onoff.py:334).
//
// Design (measured on MI355X, profiles/r01_ubench_mfma_f64_*.log):
//   * v_mfma_f64_16x16x4_f64 sustains only ~50 TFLOP/s (~97 cycles/instr/SIMD);
//     v_mfma_f64_4x4x4_4b_f64 sustains ~75 TFLOP/s (~16.9 cycles/instr/SIMD, 512 flop each).
//     cbsz/abid broadcast is ignored for f64, so a 16x16x4 product is issued as FOUR 4x4x4(4-block)
//     instructions whose A operand is a 4-row block read from LDS with the same address in all four
//     16-lane groups (LDS broadcast is free).  The accumulator layout then equals the 16x16x4 one:
//     acc[r] of lane l = C[4r + l/16][l%16].
//   * workgroup = 256 threads = 4 waves (2x2), tile 128x128, BK = 16; wave tile 64x64 = 16 sub-tiles,
//     64 independent accumulators per lane (128 VGPRs) -> MFMA issue is never dependency-bound.
//   * operand tiles are staged global -> registers -> LDS (padded rows: conflict-free ds_read_b64),
//     next tile's global loads are in flight during the MFMAs of the current one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zigp {

constexpr int BM = 128, BN = 128, BK = 16, GEMM_THREADS = 256;
constexpr int LDK = BK + 2;     // row stride (doubles) of a k-contiguous tile  [128][18]
constexpr int LDMN = BM + 16;   // row stride (doubles) of an m/n-contiguous tile [16][144]
constexpr int TILE_DOUBLES = 128 * LDK;  // == 16*LDMN == 2304

// Operand layouts: element (i,k) of A / (k,j) of B
enum { LAY_KCONTIG = 0,   // A[i*ld + k]   /  B[j*ld + k]
       LAY_MNCONTIG = 1   // A[k*ld + i]   /  B[k*ld + j]
};

// One entry of the host-built work list (sorted by descending k-extent = LPT order).
struct GemmTile {
  int bi, bj;        // output tile coordinates (units of 128)
  int kbeg, kend;    // k range in units of BK, applied to every segment
  int slice;         // split-K slice id (selects the partial-output plane), 0 if unused
  int pad0, pad1, pad2;
};

struct GemmSeg { const double* A; const double* B; int64_t lda, ldb; };

struct GemmArgs {
  GemmSeg seg[2];
  int nseg;
  const GemmTile* tiles;
  double* C; int64_t ldc; int64_t slice_stride;  // C plane stride for split-K partials
  double alpha;
};

// ---- B-operand producers: transform values as they are staged (k = row of B, n = column of B) ----
struct BIdentity {
  __device__ __forceinline__ double2 operator()(int64_t, int64_t, double2 v) const { return v; }
};
// dA2[m,n] = gm[n]*u[m] + 2*gv[n]*s2[m]*A2[m,n]   (cotangent of A2 = L^-T A1: mean = A2^T u, var += sum (s A2)^2;
// onofftf/main.py:287,291,302 differentiated).  B is n-contiguous: v = (A2[k][n], A2[k][n+1]).
struct BProdDA2 {
  const double* gm; const double* gv; const double* u; const double* s2;
  __device__ __forceinline__ double2 operator()(int64_t k, int64_t n, double2 v) const {
    double uk = u[k], sk = 2.0 * s2[k];
    double2 r; r.x = gm[n] * uk + gv[n] * sk * v.x; r.y = gm[n + 1] * uk + gv[n + 1] * sk * v.y; return r;
  }
};

// ---- epilogues: called once per accumulator element with its global (row, col) ----
struct EpiStore {   // C = alpha*acc
  __device__ __forceinline__ void operator()(double* C, int64_t ldc, int64_t i, int64_t j, double v) const { C[i * ldc + j] = v; }
};
struct EpiAccum {   // C += alpha*acc
  __device__ __forceinline__ void operator()(double* C, int64_t ldc, int64_t i, int64_t j, double v) const { C[i * ldc + j] += v; }
};
// E = W dA2 stored to C; dA1 = E - 2 gv[n] A1[m,n] stored to dA1  (fvar = Kdiag - sum A1^2, main.py:278)
struct EpiDA1 {
  const double* A1; double* dA1; const double* gv;
  __device__ __forceinline__ void operator()(double* C, int64_t ldc, int64_t i, int64_t j, double v) const {
    C[i * ldc + j] = v; dA1[i * ldc + j] = v - 2.0 * gv[j] * A1[i * ldc + j];
  }
};
// C = alpha*acc on/below the diagonal, 0 above (used where only tril is meaningful)
struct EpiStoreTril {
  __device__ __forceinline__ void operator()(double* C, int64_t ldc, int64_t i, int64_t j, double v) const { C[i * ldc + j] = (j <= i) ? v : 0.0; }
};

template <int LAY>
__device__ __forceinline__ void load_tile_regs(double2 (&r)[4], const double* __restrict__ P, int64_t ld,
                                               int64_t mn0, int64_t k0, int t) {
  if (LAY == LAY_KCONTIG) {            // 128 rows x 16 k ; thread: row = t/8 + 32p, kk = (t%8)*2
    const int kk = (t & 7) * 2, row = t >> 3;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      r[p] = *reinterpret_cast<const double2*>(P + (mn0 + row + 32 * p) * ld + k0 + kk);
  } else {                              // 16 k-rows x 128 mn ; thread: krow = t/64 + 4p, mm = (t%64)*2
    const int mm = (t & 63) * 2, krow = t >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      r[p] = *reinterpret_cast<const double2*>(P + (k0 + krow + 4 * p) * ld + mn0 + mm);
  }
}
template <int LAY>
__device__ __forceinline__ void store_tile_lds(double* __restrict__ S, const double2 (&r)[4], int t) {
  if (LAY == LAY_KCONTIG) {
    const int kk = (t & 7) * 2, row = t >> 3;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<double2*>(S + (row + 32 * p) * LDK + kk) = r[p];
  } else {
    const int mm = (t & 63) * 2, krow = t >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<double2*>(S + (krow + 4 * p) * LDMN + mm) = r[p];
  }
}

template <int ALAY, int BLAY, class BProd, class Epi>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
gemm_f64_kernel(GemmArgs g, BProd bprod, Epi epi) {
  __shared__ double lds[2 * TILE_DOUBLES];
  double* As = lds;
  double* Bs = lds + TILE_DOUBLES;
  const GemmTile tl = g.tiles[blockIdx.x];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t row0 = (int64_t)tl.bi * BM, col0 = (int64_t)tl.bj * BN;

  double acc[4][4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][b][c] = 0.0;

  const int nk = tl.kend - tl.kbeg;
  const int total = nk * g.nseg;
  double2 ra[4], rb[4];

  auto issue_loads = [&](int it) {
    const int sg = it / nk, kb = tl.kbeg + (it - sg * nk);
    const GemmSeg& s = g.seg[sg];
    const int64_t k0 = (int64_t)kb * BK;
    load_tile_regs<ALAY>(ra, s.A, s.lda, row0, k0, t);
    load_tile_regs<BLAY>(rb, s.B, s.ldb, col0, k0, t);
    // B producer (applied on registers; k / n of this thread's elements)
    if (BLAY == LAY_MNCONTIG) {
      const int mm = (t & 63) * 2, krow = t >> 6;
#pragma unroll
      for (int p = 0; p < 4; ++p) rb[p] = bprod(k0 + krow + 4 * p, col0 + mm, rb[p]);
    }
  };

  if (total > 0) issue_loads(0);
  // per-lane LDS read offsets
  const int a_i = lane & 3, a_k = lane >> 4, b_j = lane & 15;
  for (int it = 0; it < total; ++it) {
    store_tile_lds<ALAY>(As, ra, t);
    store_tile_lds<BLAY>(Bs, rb, t);
    __syncthreads();
    if (it + 1 < total) issue_loads(it + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const int k = ks * 4 + a_k;
      double af[4][4], bf[4];
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int col = wn * 64 + tn * 16 + b_j;
        bf[tn] = (BLAY == LAY_MNCONTIG) ? Bs[k * LDMN + col] : Bs[col * LDK + k];
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * 64 + tm * 16 + 4 * r + a_i;
          af[tm][r] = (ALAY == LAY_KCONTIG) ? As[row * LDK + k] : As[k * LDMN + row];
        }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[tm][tn][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[tm][r], bf[tn], acc[tm][tn][r], 0, 0, 0);
    }
    __syncthreads();
  }

  double* C = g.C + (int64_t)tl.slice * g.slice_stride;
  const int c_i = lane >> 4, c_j = lane & 15;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t gi = row0 + wm * 64 + tm * 16 + 4 * r + c_i;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int64_t gj = col0 + wn * 64 + tn * 16 + c_j;
        epi(C, g.ldc, gi, gj, g.alpha * acc[tm][tn][r]);
      }
    }
}

}  // namespace zigp
