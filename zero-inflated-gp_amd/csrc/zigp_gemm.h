// fp64 GEMM core for gfx950 (MI355X / CDNA4) -- the contraction engine behind every O(M^2 N) and O(M^3)
// product on the zero-inflated-GP ELBO path (replaces TF's MatMul / MatrixTriangularSolve call sites,
// onofftf/main.py:271,284,287 and their tf.gradients twins, scripts/onoff.py:334).
//
// Design (measured on MI355X; the measured-and-dropped alternatives live as patches under tools/, see tools/README.md):
//   * a 16 x 16 x 4 product is ONE v_mfma_f64_16x16x4_f64: 77.5 TFLOP/s chip-wide in the VGPR form these kernels compile to
//     (tools/ubench/mfma_f64_tile.hip, profiles/r04k_ubench_mfma_tile.log); the A operand of a 16-row sub-tile is one ds_read_b64 per
//     lane (row l % 16, k l / 16), the B operand one (k l / 16, column l % 16), the accumulator element r of lane l is
//     C[4 r + l / 16][l % 16].  (Rounds 1-3 issued four v_mfma_f64_4x4x4_4b_f64 instead: tools/r4_experiment_arms.patch.)
//   * workgroup tile 128x128, BK = 16, as 4 waves (2x2, 64x64 wave tiles, 64 accumulators/lane) or 8 waves
//     (4x2, 32x64 wave tiles, 32 accumulators/lane); two workgroups share a CU.  MFMA issue is never
//     dependency-bound (>= 8 independent 16x16 accumulator blocks).
//   * operand tiles go global -> LDS directly (global_load_lds_dwordx4, no staging registers) into a
//     2-deep ring.  One wave-instruction writes 1 KB linearly, so bank conflicts are removed
//     (a) for k-contiguous tiles ([128 rows][16 k], 8 rows per instruction) by an XOR swizzle of the 16-byte
//         granule index with (row>>1)&7, applied on the SOURCE address and again on every ds_read
//         (cdna_hip_programming.md rule 21), plus 16 B of padding in front of every 16-row block (see glds_tile);
//     (b) for m/n-contiguous tiles ([16 k][128], one k-row per instruction) by giving each row its own
//         M0 base with a 144-double stride (odd k rows land 128 B further round the banks).
//     One s_barrier per BK step; counted vmcnt keeps NSTAGE-2 stages in flight across it.
//   * a workgroup runs a host-built UNIT of list entries (GemmArgs::per consecutive tiles, each with its own k range and k
//     direction); the lists are built in zigp_host.h (tiles_trmm / tiles_syr2k / tiles_full_xcd).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace zigp {

constexpr int BM = 128, BN = 128, BK = 16;
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma16(double (&c)[4], double a, double b) {
  mfma_d4 v = {c[0], c[1], c[2], c[3]};
  v = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, v, 0, 0, 0);
  c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
}
// Workgroup shapes (template parameter WAVES of the kernel): WAVES/2 (M) x 2 (N) waves, wave tiles 64 columns wide.
//   WAVES = 4: wave tile 64x64, 64 accumulators/lane, <= 256 VGPRs, 2 waves/SIMD with 2 workgroups/CU
//   WAVES = 8: wave tile 32x64, 32 accumulators/lane, <= 128 VGPRs, 4 waves/SIMD with 2 workgroups/CU; triangular
//              structure is skipped at 16-row granularity (balanced sub-tile pairs, see gemm_tile)
// Measured on MI355X (cfg3): the 8-wave shape is ~7 % faster where it fits 128 VGPRs without spilling.
template <int WAVES> struct Shape {
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int WMW = WAVES / 2;        // waves along M
  static constexpr int WNW = 2;                // waves along N
  static constexpr int RW = BM / WMW;          // rows per wave: 64 or 32
  static constexpr int TMW = RW / 16;          // 16-row sub-tiles per wave: 4 or 2
  static constexpr int WTN = BN / WNW;         // wave tile width: 64
  static constexpr int TNW = WTN / 16;         // 16-column sub-tiles per wave: 4
  static constexpr int CHUNKS = 16 / WAVES;    // 1 KB staging chunks per wave and operand tile
};
constexpr int LDMN = 128 + 16;                    // row stride (doubles) of an m/n-contiguous tile image: odd k rows shift 128 B
constexpr int KSUB = 256 + 2;                     // doubles from one 16-row block of a k-contiguous image to the next (16 B of padding, see glds_tile)
constexpr int TILE_DOUBLES = BK * LDMN;           // 2304 doubles (18 KB) holds either image: [128][16] swizzled + padded, or [16][144]
constexpr int STAGE_DOUBLES = 2 * TILE_DOUBLES + BK;   // A tile + B tile + one BK-slice of the k-scale vector

// Operand layouts: element (i,k) of A / (k,j) of B
enum { LAY_KCONTIG = 0,   // A[i*ld + k]   /  B[j*ld + k]
       LAY_MNCONTIG = 1   // A[k*ld + i]   /  B[k*ld + j]
};

// One entry of the host-built work list.  A workgroup runs GemmArgs::per consecutive entries one after the other (a "unit");
// entries with kend <= kbeg are padding and are skipped.
struct GemmTile {
  int bi, bj;        // output tile coordinates (units of 128)
  int kbeg, kend;    // k range in units of BK, applied to every segment
  int slice;         // split-K slice id (selects the partial-output plane), 0 if unused
  int kdir;          // +1: BK steps run kbeg -> kend-1, -1: kend-1 -> kbeg (lets the tiles of one column panel walk k in lockstep)
  int pad1, pad2;
};

struct GemmSeg { const double* A; const double* B; int64_t lda, ldb; };   // one operand pair

struct GemmArgs {
  GemmSeg seg[1];
  const GemmTile* tiles;
  int per;                // list entries per workgroup
  double* C; int64_t ldc; int64_t slice_stride;  // C plane stride for split-K partials
  double alpha;
  const double* kscale;   // KSCALE kernels: B[k][j] is multiplied by kscale[k] (k = global reduction index)
};

// ---- epilogues --------------------------------------------------------------------------------
// acc[tm][tn][r] of lane l is C[row0 + wm*RW + tm*16 + 4r + l/16][col0 + wn*64 + tn*16 + l%16]  (RW = 64 or 32 rows per wave).
// An epilogue is `template <int TM, int TN> void operator()(const double (&acc)[TM][TN][4], const EpiCtx&) const`.
struct EpiCtx {
  double* C; int64_t ldc; double alpha;
  int64_t row0, col0;   // first row / column of this wave's sub-tile (tm = 0, tn = 0)
  int64_t prow;         // index of this wave's partial row for the fused column sums (one per wave and row block)
  int lane;
  int64_t tm_stride = 16;   // rows from sub-tile tm to tm + 1 (16: contiguous; the balanced triangular kernels own sub-tiles wm and 7 - wm)
};
template <int TM, int TN, class F>
__device__ __forceinline__ void epi_foreach(const double (&acc)[TM][TN][4], const EpiCtx& e, F f) {
  const int c_i = e.lane >> 4, c_j = e.lane & 15;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t gi = e.row0 + tm * e.tm_stride + 4 * r + c_i;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) f(gi, e.col0 + tn * 16 + c_j, e.alpha * acc[tm][tn][r]);
    }
}
struct EpiStore {   // C = alpha*acc
  template <int TM, int TN>
  __device__ __forceinline__ void operator()(const double (&acc)[TM][TN][4], const EpiCtx& e) const {
    double* __restrict__ C = e.C; const int64_t ld = e.ldc;
    epi_foreach(acc, e, [&](int64_t i, int64_t j, double v) { C[i * ld + j] = v; });
  }
};
struct EpiStorePanel : EpiStore {};   // same code under its own kernel name: the chunk loop's J' = Q A2 (profilers aggregate by name, and the M x M stage launches the plain one)
struct EpiAccum {   // C += alpha*acc
  template <int TM, int TN>
  __device__ __forceinline__ void operator()(const double (&acc)[TM][TN][4], const EpiCtx& e) const {
    double* C = e.C; const int64_t ld = e.ldc;
    epi_foreach(acc, e, [&](int64_t i, int64_t j, double v) { C[i * ld + j] += v; });
  }
};

// Store + fused column reductions over this wave's 16 * TM rows (GPConditional's reduce_sum over the inducing index,
// onofftf/main.py:278,287,291,302):   out1[n] = sum_m w1[m] C[m,n]   (skipped if w1 == nullptr)
//                                      out2[n] = sum_m w2[m] C[m,n]^2 (w2 == nullptr -> weight 1)
// written to partial row e.prow (one per wave and row block) of out1/out2 (each [Mp / (16 TM)][ldc]); the point-wise kernel adds the partial
// rows in index order, so the result does not depend on scheduling.
template <bool STORE, int TM, int TN>
__device__ __forceinline__ void epi_colsum(const double (&acc)[TM][TN][4], const EpiCtx& e, const double* __restrict__ w1, const double* __restrict__ w2,
                                           double* __restrict__ out1, double* __restrict__ out2) {
  double* __restrict__ C = e.C; const int64_t ld = e.ldc;
  const int c_i = e.lane >> 4, c_j = e.lane & 15;
  // one partial row per wave tile (16 * TM rows)
  double s1[TN], s2[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) { s1[tn] = 0.0; s2[tn] = 0.0; }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t gi = e.row0 + tm * e.tm_stride + 4 * r + c_i;
      const double a1 = w1 ? w1[gi] : 0.0, a2 = w2 ? w2[gi] : 1.0;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const double v = e.alpha * acc[tm][tn][r];
        if (STORE) C[gi * ld + e.col0 + tn * 16 + c_j] = v;
        s1[tn] = fma(a1, v, s1[tn]);
        s2[tn] = fma(a2 * v, v, s2[tn]);
      }
    }
  const int64_t prow = e.prow;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {   // fixed-order combine of the four 16-lane row groups
    double a = s1[tn], b = s2[tn];
    a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
    if (c_i == 0) {
      const int64_t o = prow * ld + e.col0 + tn * 16 + c_j;
      if (w1) out1[o] = a;
      out2[o] = b;
    }
  }
}
struct EpiStoreColsum {
  const double* __restrict__ w1; const double* __restrict__ w2; double* __restrict__ out1; double* __restrict__ out2;
  template <int TM, int TN>
  __device__ __forceinline__ void operator()(const double (&acc)[TM][TN][4], const EpiCtx& e) const { epi_colsum<true>(acc, e, w1, w2, out1, out2); }
};
// The column sums WITHOUT the panel (round 6): A2 = W^T A1 is needed for  sum_m s^2 A2^2  only -- the reverse pass takes J' = (Q W^T) A1 from
// A1 directly and A2 gm = W^T (A1 gm) is linear -- so the product's 8 Mp Nc bytes are never written: same accumulators, same sums, bit for bit.
// e.ldc is still the row stride of the partial-row planes; e.C is not used.
struct EpiColsum {
  const double* __restrict__ w1; const double* __restrict__ w2; double* __restrict__ out1; double* __restrict__ out2;
  template <int TM, int TN>
  __device__ __forceinline__ void operator()(const double (&acc)[TM][TN][4], const EpiCtx& e) const { epi_colsum<false>(acc, e, w1, w2, out1, out2); }
};

// k-contiguous image swizzle: granule position = (k>>1) ^ kswz(row).  kswz takes 8 distinct values on the even and on the odd rows of
// any aligned group of 16 (the read of 16 rows x 4 k by one wave, both operand roles): bank-conflict free.
__device__ __forceinline__ int kswz(int row) { return (row >> 1) & 7; }

// Staging addresses.  A global_load_lds takes (scalar base) + (32-bit per-lane byte offset).  The per-lane offset of this wave's FIRST
// 1 KB chunk is computed once per tile; its other 16 / WAVES - 1 chunks lie a uniform distance further on (chunk c = WAVES p + wave holds
// rows 8c .. 8c + 7 of a k-contiguous tile -- 8 WAVES rows per p, and the swizzle only sees the row's low four bits -- or k-row c of an
// m/n-contiguous one), which goes into the SCALAR base: one offset register per operand instead of 16 / WAVES.  The scalar base
// advances by a constant each BK step, and the LDS destination (M0) is scalar arithmetic on the wave id -- no per-step vector address
// math (glds_pin below is what makes the compiler keep it that way).
template <int LAY, int WAVES>
__device__ __forceinline__ uint32_t glds_lane_offset(int64_t ld, int wave, int lane) {
  if (LAY == LAY_KCONTIG) {                           // chunk = rows 8c..8c+7, lane -> (row, granule position)
    const int row = 8 * wave + (lane >> 3), gp = lane & 7;
    const int g = gp ^ kswz(row);
    return (uint32_t)((row * ld + 2 * g) * 8);
  }
  return (uint32_t)((wave * ld + 2 * lane) * 8);      // chunk = k-row c, lane -> granule
}
template <int LAY, int WAVES>
__device__ __forceinline__ int64_t glds_chunk_stride(int64_t ld) { return (LAY == LAY_KCONTIG ? 8 * WAVES : WAVES) * ld * 8; }
// The uniform part of a staging address, pinned in a scalar register pair.  Without this the loop optimiser turns (uniform base that
// advances per BK step) + (per-lane offset) into one 64-bit per-lane pointer PER LOAD as its induction variables: 2 VGPRs and two
// v_lshl_add_u64 per load and step (18 VGPRs and 18 64-bit VALU adds per BK step in the 4-wave kernels; fp64 MFMAs do not overlap with
// VALU work on this chip, profiles/r03t_kron_stamps.txt).  readfirstlane of a uniform value costs nothing (it folds to the SGPR it
// already lives in) but is opaque to that transformation, and SGPR base + 32-bit VGPR offset is the load's native address form.
// Measured: +0.5 % on every GEMM class, within box noise per step (profiles/r03x_ab_pin.log); kept for the registers.
__device__ __forceinline__ const char* glds_pin(const char* p) {
  const uint64_t u = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (const char*)(((uint64_t)hi << 32) | lo);
}
// A k-contiguous image gets 2 doubles (16 B) of padding in front of every 16-row block.  The four fragments of a k-step then sit
// 2064 B apart -- beyond the 2040-B reach of ds_read2_b64 and not a multiple of 512 B (ds_read2st64_b64) -- so the compiler must
// issue them as single ds_read_b64, which take the conflict-free 64-bank path (the paired form reads two 16-lane groups per cycle on a
// 32-bank mapping: 49 % of the LDS cycles of the rank-N update were bank conflicts).
template <int LAY, int WAVES>
__device__ __forceinline__ void glds_tile(double* tile, const char* __restrict__ base, uint32_t off, int64_t chunk_stride, int wave) {
  asm volatile("" : "+v"(off));   // keeps the 32 -> 64 bit extension of the offset next to the load (instruction selection works per
                                  // block: hoisted out of the loop, the extension hides that the offset is 32 bits wide)
#pragma unroll
  for (int p = 0; p < 16 / WAVES; ++p) {
    const int c = WAVES * p + wave;
    double* dst = tile + ((LAY == LAY_KCONTIG) ? c * 128 + (c >> 1) * 2 : c * LDMN);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(glds_pin(base + p * chunk_stride) + (uint64_t)off),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- hand-pipelined fragment reads (round 5) ----------------------------------------------------------------------------------------
// Left to the compiler, a k-step of the 16x16x4 form is [its LDS reads] [s_waitcnt lgkmcnt(0)] [its MFMAs]: every wave stalls for an LDS
// round trip four times per staged step.  Here the reads of k-step ks + 1 are issued IN FRONT of the MFMAs of k-step ks into a second
// fragment set and waited for with a counted lgkmcnt (LDS operations return in order): the reads are inline assembly -- the compiler's
// own s_waitcnt insertion does not see them, so the waits are exactly the ones written here -- the MFMAs stay builtins (the compiler keeps
// their hazards) and sched_barrier fences pin the order.  tools/ubench/gemm_lab.hip is where the form was developed and measured
// (profiles/r05*_gemm_lab_*.log: +1.1 % on the full product; wider reads, fewer reads, an early barrier, cross-tile prefetch, staggered
// tile boundaries and a column-strip wave shape were measured there too and do not pay).  Used for the full (J' = Q A2) and the
// lower-triangular (A1 = W K) 8-wave m/n-contiguous products: J' 70.2 -> 71.2, A1 64.5 -> 65.1 TFLOP/s, step -0.55 % same-box; the
// upper-triangular product (A2 = W^T A1) measures 1 % SLOWER with it and keeps the compiler's order (profiles/r05h_ab_pipe.log), and so
// does the 4-wave k-contiguous k-scaled kernel (the symmetric rank-N update: 61.9 vs 62.7 TFLOP/s, profiles/r05i_ab_syrk_pipe.log; the
// change is tools/syrk_pipe_experiment.patch).
template <int N> struct IC { static constexpr int value = N; };
template <int B, int E, class F> __device__ __forceinline__ void sfor(F f) { if constexpr (B < E) { f(IC<B>{}); sfor<B + 1, E>(f); } }
template <int OFF> __device__ __forceinline__ double ds_rd64(uint32_t a) {    // a: LDS byte address of this lane, OFF: compile-time byte offset
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  double v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v;
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// The counted wait of the pipelined reads, TIED to the fragments it releases: the registers written by the ds_read_b64 blocks above are
// read-write operands of the wait, so every use of them -- the MFMAs are builtins with no other dependence on an asm block -- is ordered
// behind it by data flow, whatever a later compiler's scheduler or register allocator would like to do with a pure s_waitcnt
// (the instruction stream is unchanged: same registers in, same registers out).
template <int N> __device__ __forceinline__ void wait_lgkm_for(double& a, double (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lgkm_for(double& a0, double& a1, double (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const double* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)p; }

// TRI: triangular structure exploited at wave granularity inside diagonal blocks.
// r3, measured and removed: peeling the diagonal 128 x 128 block of the triangular factor into a tail of 8 steps with 16-row granularity
// (sub-tile rows dealt out round-robin over the waves along M so that they stay balanced; the wave row a template parameter, the tail
// straight-line code) executes 1.016 instead of 1.03-1.06 x the algorithmic MFMAs and was 3 % SLOWER per step (193.4 vs 187.8 ms, same
// box, profiles/r03c_ab_tail.log): a tail step with one or two live sub-tiles is all barrier, staging wait and LDS latency, and the
// four per-wave-row copies spill 50-150 B/lane outside their main loops.
enum { TRI_NONE = 0,
       TRI_A_LOWER = 1,   // A(i,k) = 0 for k > i   (W * B)
       TRI_A_UPPER = 2,   // A(i,k) = 0 for k < i   (W^T * B)
       TRI_C_LOWER = 3    // only C(i,j), j <= i, is used (rank-N update of a lower-triangular cotangent)
};

// ---- diagonal tile of the symmetric rank-N update (TRI_C_LOWER, bi == bj: the A and the B operand are the SAME 128 rows) -------------
// The generic path computes a diagonal tile as three 64 x 64 wave tiles (the fourth wave idles) = 0.75 of a full tile for 0.5625 of
// useful work, in the time of a full tile.  Here only ONE operand tile is needed per BK step, and the 36 lower 16 x 16 sub-tiles (8 x 8
// grid, diagonal included) are dealt out 9 per wave: wave w owns the sub-tile rows w and 7 - w, i.e. w + 1 and 8 - w column sub-tiles.
// The wave index is a template parameter (the four instantiations sit behind one scalar switch): every loop bound is a compile-time
// constant.  36 accumulators, 2 A and <= 8 B fragments per k-step inside a kernel that holds ~170 VGPRs anyway.
// A ring stage has room for two operand tiles, so a step of this path covers TWO BK slices (the second one in the B slot; the k-scale
// slices ride in the spare 2 KB behind each image): half the barriers and staging waits per MFMA of the generic path, which is what
// brings a diagonal tile down to ~0.6 of a full tile's time.  The host gives diagonal tiles k ranges twice as long as the others'
// (tiles_syr2k): both kinds of workgroup then finish together, and every XCD's tiles stay inside one window of k.
template <int W, int NSTAGE, bool KSCALE, class Epi>
__device__ __forceinline__ void syrk_diag_tile(const GemmArgs& g, const GemmTile& tl, double* lds, int wave, int lane, const Epi& epi) {
  constexpr int WAVES = 4, CHUNKS = Shape<WAVES>::CHUNKS;
  constexpr int R1 = W, R2 = 7 - W, NC1 = W + 1, NC2 = 8 - W;   // this wave's two sub-tile rows and the column sub-tiles 0..NC-1 each needs
  constexpr int SCALE_OFF = 128 * 16 + 16;                       // k-scale slice of an image: behind its 2048 (+ 16 of padding) doubles (TILE_DOUBLES = 2304)
  static_assert(TILE_DOUBLES >= SCALE_OFF + BK, "no room for the k-scale slice behind the operand image");
  const GemmSeg& sg = g.seg[0];
  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int kq = ln >> 4, b_j = ln & 15;
  // element (row, k) of the swizzled, padded [128][16] image: (row / 16) * 2 + row * 16 + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1); one read
  // serves both operand roles (lane -> row b_j of a 16-row block, k = 4 ks + kq)
  int b_base[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) b_base[ks] = b_j * 16 + 2 * (((2 * ks) | (kq >> 1)) ^ kswz(b_j)) + (kq & 1);
  double acc1[1][NC1][4], acc2[1][NC2][4];
#pragma unroll
  for (int c = 0; c < NC1; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc1[0][c][r] = 0.0;
#pragma unroll
  for (int c = 0; c < NC2; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc2[0][c][r] = 0.0;
  const int total = tl.kend - tl.kbeg;          // BK slices of this tile
  const int nsteps = (total + 1) / 2;           // two slices per step; a last odd slice stands alone
  const uint32_t offA = glds_lane_offset<LAY_KCONTIG, WAVES>(sg.lda, wave, ln);
  const int64_t csA = glds_chunk_stride<LAY_KCONTIG, WAVES>(sg.lda);
  const int64_t row0 = (int64_t)tl.bi * BM;
  const int kb0 = (tl.kdir >= 0) ? tl.kbeg : tl.kend - 1;
  const int64_t kfirst = (int64_t)kb0 * BK, kd = (tl.kdir >= 0) ? 1 : -1;
  const char* baseA = (const char*)(sg.A + row0 * sg.lda + kfirst);
  const char* baseS = (const char*)(g.kscale + kfirst);
  const int64_t strideA = kd * BK * 8;
  auto issue_slice = [&](double* img, int sl) {   // BK slice `sl` of the tile's k range into the image at img
    glds_tile<LAY_KCONTIG, WAVES>(img, baseA + sl * strideA, offA, csA, wave);
    if (KSCALE) {
      uint32_t so = 16 * ln;
      asm volatile("" : "+v"(so));
      if (ln < 8)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(glds_pin(baseS + kd * sl * (BK * 8)) + (uint64_t)so),
                                         (__attribute__((address_space(3))) void*)(img + SCALE_OFF), 16, 0, 0);
    }
  };
  // every step issues the same number of loads (uniform vmcnt accounting): the missing second slice of an odd tail re-reads the first
  auto issue = [&](int st_) {
    double* st = lds + (st_ % NSTAGE) * STAGE_DOUBLES;
    issue_slice(st, 2 * st_);
    issue_slice(st + TILE_DOUBLES, 2 * st_ + 1 < total ? 2 * st_ + 1 : 2 * st_);
  };
  constexpr int GLDS_PER_STAGE = 2 * (CHUNKS + (KSCALE ? 1 : 0));
  auto slice_mfma = [&](const double* As) {
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double bf[NC2];
#pragma unroll
      for (int c = 0; c < NC2; ++c) bf[c] = As[b_base[ks] + c * KSUB];
      if (KSCALE) {
        const double sc = As[SCALE_OFF + ks * 4 + kq];
#pragma unroll
        for (int c = 0; c < NC2; ++c) bf[c] *= sc;
      }
      // the A operand of sub-tile row R is the SAME read as the B operand of column sub-tile R (one image, both roles): lane -> (row, k)
      const double a2 = As[b_base[ks] + R2 * KSUB], a1 = As[b_base[ks] + R1 * KSUB];
#pragma unroll
      for (int c = 0; c < NC2; ++c) mfma16(acc2[0][c], a2, bf[c]);
#pragma unroll
      for (int c = 0; c < NC1; ++c) mfma16(acc1[0][c], a1, bf[c]);
    }
  };
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nsteps) issue(s);
  for (int it = 0; it < nsteps; ++it) {
    if (it + NSTAGE - 2 < nsteps) wait_vmcnt<GLDS_PER_STAGE * (NSTAGE - 2)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (it + NSTAGE - 1 < nsteps) issue(it + NSTAGE - 1);
    const double* As = lds + (it % NSTAGE) * STAGE_DOUBLES;
    slice_mfma(As);
    if (2 * it + 1 < total) slice_mfma(As + TILE_DOUBLES);
  }
  EpiCtx e;
  e.C = g.C + (int64_t)tl.slice * g.slice_stride; e.ldc = g.ldc; e.alpha = g.alpha; e.lane = lane; e.prow = 0;
  e.col0 = row0;
  e.row0 = row0 + R2 * 16; epi(acc2, e);
  e.row0 = row0 + R1 * 16; epi(acc1, e);
}

// One output tile of the generic path.
template <int ALAY, int BLAY, int NSTAGE, bool KSCALE, int TRI, int WAVES, class Epi>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const GemmTile& tl, double* lds, int wave, int wm, int wn, int lane, const Epi& epi) {
  constexpr int WTN = Shape<WAVES>::WTN, TNW = Shape<WAVES>::TNW, CHUNKS = Shape<WAVES>::CHUNKS;
  constexpr int RW = Shape<WAVES>::RW, TMW = Shape<WAVES>::TMW, WMW = Shape<WAVES>::WMW;
  const GemmSeg& sg = g.seg[0];
  const int64_t row0 = (int64_t)tl.bi * BM, col0 = (int64_t)tl.bj * BN;
  // Everything derived from the lane id is recomputed per tile from an opaque copy: otherwise the compiler keeps the ~16
  // address registers alive across the epilogue of the previous tile and spills there.
  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int kq = ln >> 4, b_j = ln & 15;
  // Balanced diagonal blocks: in the 8-wave triangular products wave (wm, wn) owns the 16-row sub-tiles wm and 7 - wm of the row block instead
  // of the rows 32 wm .. 32 wm + 31.  Inside the DIAGONAL block of the triangular factor, sub-tile t needs the staged steps s <= t (lower) or
  // s >= t (upper) only; with contiguous rows and skipping per wave, wave row 3 works through all 8 steps of that block while wave row 0
  // idles after 2 -- a diagonal block costs the workgroup 8 step times for 62 % of the MFMAs.  With the pairs (t, 7 - t) every wave has
  // 9 sub-tile steps there and the step times are 2,2,2,2,1,1,1,1 halves = 6 instead of 8 (tiles are 1-8 blocks long, one of them diagonal).
  constexpr bool TRI_BAL = WAVES == 8 && ALAY == LAY_MNCONTIG && (TRI == TRI_A_LOWER || TRI == TRI_A_UPPER);
  const int wrow = TRI_BAL ? 16 * wm : wm * RW;     // row (within the 128-row block) of this wave's sub-tile tm = 0
  const int tm_stride = TRI_BAL ? 16 * (7 - 2 * wm) : 16;   // rows from sub-tile tm = 0 to tm = 1
  // per-lane LDS read bases; the (tm, tn, ks) parts are compile-time offsets (see header comment).  lane -> (row wrow + b_j, k = 4 ks + kq)
  // of the A image and (k = 4 ks + kq, column wn * 64 + b_j) of the B image; one base per k-step for a k-contiguous image, one in all for an
  // m/n-contiguous one; the 16-row / 16-column sub-tile is a compile-time offset
  int a_base_[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    a_base_[ks] = (ALAY == LAY_KCONTIG) ? ((wrow + b_j) * 16 + (wrow / 16) * 2 + 2 * (((2 * ks) | (kq >> 1)) ^ kswz(b_j)) + (kq & 1))
                                       : (kq * LDMN + wrow + b_j + ks * 4 * LDMN);
  int b_base_[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    b_base_[ks] = (BLAY == LAY_KCONTIG) ? ((wn * WTN + b_j) * 16 + wn * (WTN / 16) * 2 + 2 * (((2 * ks) | (kq >> 1)) ^ kswz(b_j)) + (kq & 1))
                                       : (kq * LDMN + wn * WTN + b_j + ks * 4 * LDMN);

  double acc[TMW][TNW][4];
#pragma unroll
  for (int a = 0; a < TMW; ++a)
#pragma unroll
    for (int b = 0; b < TNW; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][b][c] = 0.0;

  const int total = tl.kend - tl.kbeg;       // BK steps of this tile
  const uint32_t offA = glds_lane_offset<ALAY, WAVES>(sg.lda, wave, ln), offB = glds_lane_offset<BLAY, WAVES>(sg.ldb, wave, ln);
  const int64_t csA = glds_chunk_stride<ALAY, WAVES>(sg.lda), csB = glds_chunk_stride<BLAY, WAVES>(sg.ldb);
  // scalar bases of BK step 0 and their per-step strides (bytes)
  const int kb0 = (tl.kdir >= 0) ? tl.kbeg : tl.kend - 1;   // k block (units of BK) of BK step 0; step `it` is kb0 + kdir * it
  const int64_t kfirst = (int64_t)kb0 * BK;
  const char* baseA = (const char*)(sg.A + ((ALAY == LAY_KCONTIG) ? row0 * sg.lda + kfirst : kfirst * sg.lda + row0));
  const char* baseB = (const char*)(sg.B + ((BLAY == LAY_KCONTIG) ? col0 * sg.ldb + kfirst : kfirst * sg.ldb + col0));
  const int64_t kd = (tl.kdir >= 0) ? 1 : -1;
  const int64_t strideA = kd * ((ALAY == LAY_KCONTIG) ? BK * 8 : BK * 8 * sg.lda);
  const int64_t strideB = kd * ((BLAY == LAY_KCONTIG) ? BK * 8 : BK * 8 * sg.ldb);
  const char* baseS = (const char*)(g.kscale + kfirst);

  auto issue = [&](int it) {
    double* st = lds + (it % NSTAGE) * STAGE_DOUBLES;
    glds_tile<ALAY, WAVES>(st, baseA + it * strideA, offA, csA, wave);
    glds_tile<BLAY, WAVES>(st + TILE_DOUBLES, baseB + it * strideB, offB, csB, wave);
    if (KSCALE) {   // 16 doubles of the scale vector; every wave issues the same 128 B (uniform vmcnt accounting)
      uint32_t so = 16 * ln;
      asm volatile("" : "+v"(so));
      if (ln < 8)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(glds_pin(baseS + kd * it * (BK * 8)) + (uint64_t)so),
                                         (__attribute__((address_space(3))) void*)(st + 2 * TILE_DOUBLES), 16, 0, 0);
    }
  };
  constexpr int GLDS_PER_STAGE = 2 * CHUNKS + (KSCALE ? 1 : 0);

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < total) issue(s);

  // stage `it` must have landed (at most NSTAGE-2 younger stages may stay in flight), every wave is done with the stage about to be
  // overwritten, then the next stage is requested: one barrier per BK step
  auto stage_step = [&](int it) -> const double* {
    if (it + NSTAGE - 2 < total) wait_vmcnt<GLDS_PER_STAGE * (NSTAGE - 2)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (it + NSTAGE - 1 < total) issue(it + NSTAGE - 1);
    return lds + (it % NSTAGE) * STAGE_DOUBLES;
  };
  // One BK step of this wave's RW x 64 sub-tile, fully unrolled.  The loop has ONE body per statement (runtime predicates inside the
  // unrolled nest, and a choice of bodies, spill and pessimise its schedule; HISTORY.md section 5).
  auto body = [&](const double* As, auto mask_c) {     // mask_c: which of the wave's sub-tile rows take part (bit tm)
    constexpr int MASK = decltype(mask_c)::value;
    const double* Bs = As + TILE_DOUBLES;
    const int (&a_base)[4] = a_base_; const int (&b_base)[4] = b_base_;
    if constexpr (WAVES == 8 && ALAY == LAY_MNCONTIG && BLAY == LAY_MNCONTIG && !KSCALE && TRI != TRI_A_UPPER) {
      // the chunk loop's full and lower-triangular products (J' = Q A2, A1 = W K, and the predictive ones): fragment reads one k-step
      // ahead of the MFMAs, see ds_rd64 above
      constexpr int NA = (MASK & 1) + ((MASK >> 1) & 1);
      static_assert(TMW == 2, "8-wave shape: two sub-tile rows per wave");
      constexpr int NRD = NA + TNW;
      const uint32_t aa0 = lds_addr(As) + 8u * (uint32_t)a_base[0];                          // sub-tile tm = 0 of k-step 0
      const uint32_t aa1 = aa0 + 8u * (uint32_t)(TRI_BAL ? tm_stride : 16);                  // sub-tile tm = 1
      const uint32_t ba = lds_addr(Bs) + 8u * (uint32_t)b_base[0];
      double af[2][TMW], bf[2][TNW];
      auto load = [&](auto ks_, auto set_) {
        constexpr int ks = decltype(ks_)::value, set = decltype(set_)::value;
        if constexpr ((MASK & 1) != 0) af[set][0] = ds_rd64<ks * 4 * LDMN * 8>(aa0);
        if constexpr ((MASK & 2) != 0) af[set][1] = ds_rd64<ks * 4 * LDMN * 8>(aa1);
        sfor<0, TNW>([&](auto p_) { constexpr int p = decltype(p_)::value; bf[set][p] = ds_rd64<(ks * 4 * LDMN + p * 16) * 8>(ba); });
      };
      load(IC<0>{}, IC<0>{});
      sfor<0, BK / 4>([&](auto ks_) {
        constexpr int ks = decltype(ks_)::value;
        static_assert(TNW == 4, "wait_lgkm_for ties four B fragments");
        auto wait_set = [&](auto n_) {      // wait until at most n_ reads are outstanding: the fragment set ks & 1 has landed
          constexpr int n = decltype(n_)::value;
          if constexpr (MASK == 3) wait_lgkm_for<n>(af[ks & 1][0], af[ks & 1][1], bf[ks & 1]);
          else if constexpr (MASK == 1) wait_lgkm_for<n>(af[ks & 1][0], bf[ks & 1]);
          else wait_lgkm_for<n>(af[ks & 1][1], bf[ks & 1]);
        };
        if constexpr (ks + 1 < BK / 4) { load(IC<ks + 1>{}, IC<(ks + 1) & 1>{}); wait_set(IC<NRD>{}); } else wait_set(IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TMW; ++tm)
          if ((MASK >> tm) & 1) {
#pragma unroll
            for (int tn = 0; tn < TNW; ++tn) mfma16(acc[tm][tn], af[ks & 1][tm], bf[ks & 1][tn]);
          }
        __builtin_amdgcn_sched_barrier(0);
      });
      return;
    }
    // every other kernel: the compiler's order (a k-step's reads, s_waitcnt lgkmcnt(0), its MFMAs; with 4 waves per SIMD the other waves
    // cover the wait)
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double bf[TNW], af[TMW];
#pragma unroll
      for (int tn = 0; tn < TNW; ++tn) bf[tn] = Bs[b_base[ks] + ((BLAY == LAY_KCONTIG) ? tn * KSUB : tn * 16)];
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm)
        if ((MASK >> tm) & 1) af[tm] = As[a_base[ks] + ((ALAY == LAY_KCONTIG) ? tm * KSUB : (TRI_BAL ? tm * tm_stride : tm * 16))];
      if (KSCALE) {
        const double sc = As[2 * TILE_DOUBLES + ks * 4 + kq];
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) bf[tn] *= sc;
      }
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm)
        if ((MASK >> tm) & 1) {
#pragma unroll
          for (int tn = 0; tn < TNW; ++tn) mfma16(acc[tm][tn], af[tm], bf[tn]);
        }
    }
  };
  for (int it = 0; it < total; ++it) {
    const double* As = stage_step(it);
    // Triangular structure at wave granularity: a wave whose rows cannot touch this BK step of a triangular A, or whose whole 64x64
    // output lies above the diagonal of a lower-triangular C, issues no MFMAs (it still takes part in staging and barriers; the
    // co-resident workgroup gets the matrix pipe).
    const int krel = (kb0 + (int)kd * it) * BK - tl.bi * BM;   // k offset of this step relative to the row block
    constexpr std::integral_constant<int, (1 << TMW) - 1> all_rows{};
    if constexpr (TRI_BAL) {
      static_assert(!TRI_BAL || (TMW == 2 && BK == 16), "balanced triangular kernels: two sub-tile rows per wave, one staged step per 16 k");
      const int st = krel >> 4;       // staged step relative to the diagonal block (< 0: left of it, > 7: right of it)
      // two independent `if`s, not if / else-if: with one conditional update of the accumulators per statement the compiler keeps them in
      // place (as for the plain skip); an if / else-if of two different bodies made it copy all 64 accumulator registers per branch and spill
      if (TRI == TRI_A_LOWER) {       // sub-tile t takes part while st <= t;  wm < 7 - wm
        const bool both = st <= wm, one = !both && st <= 7 - wm;
        if (both) body(As, all_rows);
        if (one) body(As, std::integral_constant<int, 2>{});
      } else {                        // upper: sub-tile t takes part once st >= t
        const bool both = st >= 7 - wm, one = !both && st >= wm;
        if (both) body(As, all_rows);
        if (one) body(As, std::integral_constant<int, 1>{});
      }
    } else {
      bool skip = false;
      if (TRI == TRI_A_LOWER) skip = krel > wm * RW + RW - 1;
      if (TRI == TRI_A_UPPER) skip = krel + BK - 1 < wm * RW;
      if (TRI == TRI_C_LOWER) skip = (tl.bi == tl.bj) && (wn * WTN > wm * RW + RW - 1);
      if (!skip) body(As, all_rows);
    }
  }

  EpiCtx e;
  e.C = g.C + (int64_t)tl.slice * g.slice_stride; e.ldc = g.ldc; e.alpha = g.alpha;
  e.row0 = row0 + wrow; e.col0 = col0 + wn * WTN; e.lane = lane; e.prow = (int64_t)tl.bi * WMW + wm; e.tm_stride = tm_stride;
  epi(acc, e);
}

template <int ALAY, int BLAY, int NSTAGE, bool KSCALE, int TRI, int WAVES, class Epi>
__device__ __forceinline__ void gemm_workgroup(GemmArgs g, Epi epi, const GemmArgs& g1, const Epi& epi1, int split, int wg, double* lds) {
  // Two argument sets in one launch (the chunk loop's products of latent f and latent g): workgroups [0, split) run set 0 with
  // list position wg, workgroups [round_up(split, 8), ...) run set 1 with list position wg - round_up(split, 8)
  // (the lists are built for "launch position p runs on XCD p % 8"); the workgroups in between are padding.  A plain launch passes
  // split = gridDim.x.  The choice is made once per workgroup, in scalar registers.
  if (wg >= split) {
    const int s8 = (split + 7) & ~7;
    if (wg < s8) return;
    wg -= s8; g = g1; epi = epi1;
  }
  constexpr int WNW = Shape<WAVES>::WNW;
  static_assert(TILE_DOUBLES >= 128 * 16 + 16, "padded k-contiguous image must fit the stage");
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: LDS destinations and wave-level skips stay on the SALU
  const int wm = wave / WNW, wn = wave % WNW;
  bool ring_used = false;
  for (int u = 0; u < g.per; ++u) {
  const GemmTile tl = g.tiles[(int64_t)wg * g.per + u];
  if (tl.kend <= tl.kbeg) continue;                 // padding entry (uniform over the workgroup)
  if (ring_used) __builtin_amdgcn_s_barrier();      // slower waves may still read the previous tile's last stage
  ring_used = true;
  if constexpr (TRI == TRI_C_LOWER && ALAY == LAY_KCONTIG && BLAY == LAY_KCONTIG && WAVES == 4) {
    if (tl.bi == tl.bj) {   // diagonal tile of the symmetric update: balanced lower-triangle path (uniform over the workgroup)
      switch (wave) {
        case 0: syrk_diag_tile<0, NSTAGE, KSCALE>(g, tl, lds, wave, lane, epi); break;
        case 1: syrk_diag_tile<1, NSTAGE, KSCALE>(g, tl, lds, wave, lane, epi); break;
        case 2: syrk_diag_tile<2, NSTAGE, KSCALE>(g, tl, lds, wave, lane, epi); break;
        default: syrk_diag_tile<3, NSTAGE, KSCALE>(g, tl, lds, wave, lane, epi); break;
      }
      continue;
    }
  }
  gemm_tile<ALAY, BLAY, NSTAGE, KSCALE, TRI, WAVES>(g, tl, lds, wave, wm, wn, lane, epi);
  }
}

template <int ALAY, int BLAY, int NSTAGE, bool KSCALE, int TRI, int WAVES, class Epi>
__global__ void __launch_bounds__(64 * WAVES, ((NSTAGE <= 2) ? 2 : 1) * WAVES / 4)
gemm_f64_kernel(GemmArgs g, Epi epi, GemmArgs g1, Epi epi1, int split) {
  extern __shared__ double lds[];   // NSTAGE * STAGE_DOUBLES
  gemm_workgroup<ALAY, BLAY, NSTAGE, KSCALE, TRI, WAVES, Epi>(g, epi, g1, epi1, split, (int)blockIdx.x, lds);
}

}  // namespace zigp
