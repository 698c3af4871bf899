/* libzigp -- measurement hooks (bench.py) and diagnostics used by the parity tests.  Not part of the drop-in boundary
 * (include/zigp.h); exported by the same libzigp.so. */
#ifndef ZIGP_DIAG_H
#define ZIGP_DIAG_H
#include "zigp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement hooks (bench.py) ---- */
/* Accumulated HIP-event time (ms), launch count and algorithmic flops per kernel class since the last reset,
 * measured with HIP events on the stream the kernels run on.  Classes (gemm_f64_kernel template arguments are
 * <A layout, B layout, ring stages, k-scale, triangular mode, waves, epilogue>):
 * 0 gemm_A1  A1 = W K                  gemm_f64_kernel<1,1,2,false,1,8,EpiStoreColsum>   (W read through its transpose; fused sum v A1, sum A1^2)
 * 1 gemm_A2  A2 = W^T A1               gemm_f64_kernel<1,1,2,false,2,8,EpiColsum>        (fused sum s^2 A2^2; the panel itself is not stored)
 * 2 gemm_H   (not launched since round 4: H = W diag(s^2) A2 is folded into class 3; the slot keeps the class numbering)
 * 3 gemm_J   J' = Q A2 = (Q W^T) A1    gemm_f64_kernel<1,1,2,false,0,8,EpiStorePanel>    (one full product on the A1 panel; with zigp_set_overlap(1) it is
 *                                      gemm_j_pw_kernel, whose leading workgroups are the point-wise stage)
 * 4 syrk     C1 += A1 G A1^T           gemm_f64_kernel<0,0,2,true,3,4,EpiAccum>
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

/* Clock stamps of the 8 XCDs, taken in stream order on the library's main stream (the call synchronises): out[8][3] =
 * {XCC id, shader-clock counter (s_memtime), constant 100 MHz counter (s_memrealtime)}.  Two calls around a region give the
 * sustained shader clock of that region: (cycles_1 - cycles_0) / ((rt_1 - rt_0) / 1e8), paired by XCC id. */
int zigp_clock_stamp(zigp_ctx* ctx, int64_t* out /*[24]*/);

/* ---- diagnostics used by the parity tests (building blocks through the same kernels) ---- */
/* Kronecker entry points: on != 0 forces the GEMM-panel path (zigp_kron.hip) also for grids the fused register-resident kernels
 * (zigp_kronf.hip) cover -- two independent implementations of the same factored algebra that the tests check against each other. */
int zigp_set_kron_panels(zigp_ctx* ctx, int32_t on);
/* Host only -- no context, no GPU: builds the tile lists the chunk loop launches for its triangular products (lower: A1 = W K, else
 * A2 = W^T A1) of a chunk of Nc rows (a multiple of 128) with Mf / Mg inducing points, exactly as chunk_forward does (paired order, merged
 * launch, LPT tail of a last wave that is not full when tail_on), and checks them: every (row block, column panel) tile exactly once, with
 * the whole k range of its row block.  out[8] = {workgroups of latent f's list, of latent g's, entries per workgroup f, g, tail units f, g,
 * largest tail workgroup in k blocks, paired order (0 / 1)}.  Returns 0, ZIGP_EARG, or -10 ... -13 for a list that is not a partition. */
int zigp_test_trmm_list(int32_t lower, int32_t Mf, int32_t Mg, int64_t Nc, int32_t tail_on, int64_t* out);
/* The gradient step of the larger fused grids (<= 16 x <= 112 points) sends its rows through in ranges of `tiles` 16-point tiles
 * (default 1024 = 16 384 rows: the per-point operand records of a range stay within 128 MB).  Results do not depend on it, bit for bit;
 * the tests lower it to run many ranges on small inputs. */
int zigp_set_kron_range_tiles(zigp_ctx* ctx, int32_t tiles);
/* C (m,n) = op(A) * op(B) with the fp64 MFMA GEMM core; transA/transB as BLAS; all dims padded internally. */
int zigp_test_gemm(zigp_ctx* ctx, int32_t transA, int32_t transB, int64_t m, int64_t n, int64_t k,
                   const double* A, const double* B, double* C);
/* L = chol(A) (lower), W = L^-1, A is (n,n) SPD; either output may be NULL.  split_k != 0: the blocked chain's products run as k slices +
 * an ordered reduction, as in the M x M forward of zigp_elbo (0: one workgroup per tile, as in the Kronecker panel path). */
int zigp_test_potrf_trtri(zigp_ctx* ctx, int64_t n, const double* A, double* L, double* W, int32_t split_k);

/* The chunk loop's cross-covariance kernel on its own: K (M,N) row-major = var * exp(-0.5 |(z_m - x_n) / ell|^2) as k_kuf_build writes a
 * Kuf panel (kern.K(X, Xnew), onofftf/main.py:266; its exponential is hand-written, see csrc/zigp_kernels.h).  X (N,D), Z (M,D), ell (D). */
int zigp_test_kuf(zigp_ctx* ctx, int64_t N, int32_t M, int32_t D, const double* X, const double* Z, const double* ell, double var, double* K);

#ifdef __cplusplus
}
#endif
#endif
