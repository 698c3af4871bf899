// Host-side helpers shared by the dense and Kronecker paths: tile lists, GEMM launcher, blocked Cholesky +
// triangular inverse, small utilities.  Everything is `static` (one copy per translation unit).
#pragma once
#include "zigp_ctx.h"
#include "zigp_kernels.h"
#include <algorithm>
#include <cmath>

namespace zigp {

struct EpiPhi {  // Phi: keep strictly-lower, halve the diagonal, zero above
  template <int TM, int TN>
  __device__ __forceinline__ void operator()(const double (&acc)[TM][TN][4], const EpiCtx& e) const {
    double* __restrict__ C = e.C; const int64_t ld = e.ldc;
    epi_foreach(acc, e, [&](int64_t i, int64_t j, double v) { C[i * ld + j] = (j < i) ? v : ((j == i) ? 0.5 * v : 0.0); });
  }
};

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

template <class F>
static int get_tiles(zigp_ctx* c, const std::string& key, F build, TileList& out, int per = 1) {
  auto it = c->tiles.find(key);
  if (it != c->tiles.end()) { out = it->second; return 0; }
  std::vector<GemmTile> v;
  build(v);
  TileList tl;
  tl.n = (int)v.size(); tl.per = per;
  if (tl.n > 0) {
    ZIGP_HIP(c, hipMalloc((void**)&tl.d, sizeof(GemmTile) * v.size()));
    ZIGP_HIP(c, hipMemcpy(tl.d, v.data(), sizeof(GemmTile) * v.size(), hipMemcpyHostToDevice));
  }
  c->tiles[key] = tl;
  out = tl;
  return 0;
}

static inline GemmTile mk_tile(int bi, int bj, int kbeg, int kend, int slice = 0) {
  GemmTile t; t.bi = bi; t.bj = bj; t.kbeg = kbeg; t.kend = kend; t.slice = slice; t.kdir = 1; t.pad1 = t.pad2 = 0; return t;
}

constexpr int NST = 2;   // LDS ring depth of the GEMM core (2 stages = 74 KB per workgroup: two workgroups share a CU)

// Workgroup shape per operand-layout pair (see Shape<> in zigp_gemm.h): 8 waves where the kernel fits 128 VGPRs -- the m/n-contiguous
// products, i.e. all of the chunk loop's but the rank-N update.  The lower-triangular products (A1 = W K) read their factor TRANSPOSED
// (m-contiguous image W^T, written once per step by k_transpose_scale) so that they run that kernel too (r3: A1 56-57 -> 60-61 TFLOP/s,
// profiles/r03c_ab_tail.log).  The 8-wave shape for the rank-N update and the 4-wave shape for the triangular products were measured
// again on the 16x16x4 core and lose (profiles/r04m_ab_shapes.log).
template <int AL, int BL, bool KS> struct WavesFor { static constexpr int value = 4; };
template <> struct WavesFor<LAY_MNCONTIG, LAY_MNCONTIG, false> { static constexpr int value = 8; };

template <int AL, int BL, bool KS, int TRI = TRI_NONE, class EP>
static int run_gemm(zigp_ctx* c, const TileList& tl, GemmArgs g, EP ep) {
  constexpr int WV = WavesFor<AL, BL, KS>::value;
  if (tl.n == 0) return 0;
  g.tiles = tl.d; g.per = tl.per;
  constexpr size_t shm = sizeof(double) * NST * STAGE_DOUBLES;
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<AL, BL, NST, KS, TRI, WV, EP>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_set = true;
  }
  const int nwg = tl.n / tl.per;
  hipLaunchKernelGGL((gemm_f64_kernel<AL, BL, NST, KS, TRI, WV, EP>), dim3(nwg), dim3(64 * WV), shm, c->stream, g, ep, g, ep, nwg);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}
// The same product for two argument sets (latent f and latent g of one chunk) in ONE launch: set 1's workgroups follow set 0's in the
// dispatch order, so the tail of one product is filled by the head of the other and the forward products of a chunk are three
// launches instead of six.
template <int AL, int BL, bool KS, int TRI = TRI_NONE, class EP>
static int run_gemm2(zigp_ctx* c, const TileList& tl0, GemmArgs g0, EP ep0, const TileList& tl1, GemmArgs g1, EP ep1) {
  if (tl0.n == 0) return run_gemm<AL, BL, KS, TRI>(c, tl1, g1, ep1);
  if (tl1.n == 0) return run_gemm<AL, BL, KS, TRI>(c, tl0, g0, ep0);
  constexpr int WV = WavesFor<AL, BL, KS>::value;
  g0.tiles = tl0.d; g0.per = tl0.per; g1.tiles = tl1.d; g1.per = tl1.per;
  constexpr size_t shm = sizeof(double) * NST * STAGE_DOUBLES;
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<AL, BL, NST, KS, TRI, WV, EP>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_set = true;
  }
  const int n0 = tl0.n / tl0.per, n1 = tl1.n / tl1.per;
  hipLaunchKernelGGL((gemm_f64_kernel<AL, BL, NST, KS, TRI, WV, EP>), dim3((unsigned)(round_up(n0, 8) + n1)), dim3(64 * WV), shm, c->stream,
                     g0, ep0, g1, ep1, n0);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

// J' = Q A2 of both latents with the point-wise stage as the LEADING workgroups of the same grid (gradient steps): the point-wise stage
// needs the column sums of A1 / A2 only and J' does not need it, so nothing inside the grid waits for anything; its npw workgroups
// (two 64-point blocks each; npw a multiple of 8 keeps the tile lists' XCD positions) run while the first tiles start, and the step has
// one launch and one launch boundary less per chunk.
template <int NSTAGE_>
__global__ void __launch_bounds__(512, 4)      // second argument: waves per SIMD (two workgroups per CU: <= 128 VGPRs)
gemm_j_pw_kernel(GemmArgs g, GemmArgs g1, int split, PwArgs pw, int npw) {
  extern __shared__ double lds[];
  if ((int)blockIdx.x < npw) {
    double (*grp)[PW_GROUPS][PW_PTS] = reinterpret_cast<double (*)[PW_GROUPS][PW_PTS]>(lds + (threadIdx.x >> 8) * (6 * PW_GROUPS * PW_PTS));
    pw_block<false>(pw, 2 * (int)blockIdx.x + (int)(threadIdx.x >> 8), (int)(threadIdx.x & 255), grp);
    return;
  }
  gemm_workgroup<LAY_MNCONTIG, LAY_MNCONTIG, NSTAGE_, false, TRI_NONE, 8, EpiStorePanel>(g, EpiStorePanel(), g1, EpiStorePanel(), split, (int)blockIdx.x - npw, lds);
}
static int run_gemm_j_pw(zigp_ctx* c, const TileList& tl0, GemmArgs g0, const TileList& tl1, GemmArgs g1, const PwArgs& pw, int pw_blocks) {
  if (tl0.n == 0 || tl1.n == 0 || (pw_blocks % 16) != 0) return fail_arg(c, "run_gemm_j_pw: both latents and a multiple of 1024 points expected");
  g0.tiles = tl0.d; g0.per = tl0.per; g1.tiles = tl1.d; g1.per = tl1.per;
  constexpr size_t shm = sizeof(double) * NST * STAGE_DOUBLES;
  static_assert(shm >= sizeof(double) * 2 * 6 * PW_GROUPS * PW_PTS, "two point-wise blocks' group sums fit the ring");
  static bool attr_set = false;
  if (!attr_set) {
    ZIGP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_j_pw_kernel<NST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_set = true;
  }
  const int n0 = tl0.n / tl0.per, n1 = tl1.n / tl1.per, npw = pw_blocks / 2;
  hipLaunchKernelGGL((gemm_j_pw_kernel<NST>), dim3((unsigned)(npw + round_up(n0, 8) + n1)), dim3(512), shm, c->stream, g0, g1, n0, pw, npw);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

static inline GemmArgs mk_args(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, double alpha = 1.0) {
  GemmArgs g;
  g.seg[0].A = A; g.seg[0].B = B; g.seg[0].lda = lda; g.seg[0].ldb = ldb;
  g.tiles = nullptr; g.per = 1; g.C = C; g.ldc = ldc; g.slice_stride = 0; g.alpha = alpha; g.kscale = nullptr;
  return g;
}

// ------------------------------------------------------------------------------------------------
// tile lists
// ------------------------------------------------------------------------------------------------
// Triangular products C(Mp x Nc) = T * B.  Row block bi of a lower-triangular T needs the k blocks [0, bi], of an upper-triangular
// one (W^T) the k blocks [bi, nbm).  Two tile orders:
// LPT (paired = false): one tile per workgroup, longest tiles first, tiles of one column panel on one XCD (launch position p runs on
//   XCD p % 8).  PMC: the B panel is fetched once per row block (1.45-1.5 GB per launch at M = 1024, Nc = 32768 for 0.54 GB of operand
//   and result) -- the tiles of a panel start at different times, so that XCD's L2 never sees them together.
// paired = true: work unit = one workgroup = two tiles of the same column panel, a short one (u + 1 k blocks) and its complement
//   (nbm - u), so every unit runs nbm + 1 k blocks; the units of a panel are consecutive entries of one XCD's queue and walk k in
//   lockstep (lower: short tiles ascend from k block 0, long tiles descend so that block-step t reads k block nbm - t in every unit;
//   upper: the mirror image): at any time the four units of a panel read at most two different slabs of it.  PMC: 2.2x fewer bytes.
//   r3, same-box A/B on the 8-wave kernels (profiles/r03p_ab_paired.log): A1 / A2 / H at the LPT rate (60.4 / 60.4 / 61.1 vs 60.5 / 59.9 /
//   61.1 TF), J' -3 % (its epilogue loads an A2 tile: with equal-length units all epilogues of a wave of workgroups coincide) -- so
//   A1, A2 and H take the paired order and J' stays LPT.  (Round 1 had the 4-wave kernels 1-2 % slower in every paired variant.)
// r6 -- the TAIL of a paired launch.  Units of equal length run in lockstep waves of the 512 resident workgroups, so a launch of
// W + r units (0 < r < 512) costs ceil as many wave times as if the last wave were full: cfg2 (3136 units of 5 k blocks = 6.125 waves)
// paid 7 x 5 = 35 block steps for 30.6 of work.  With tail_units = T > 0 the LAST T units of the launch order (T / 8 per XCD queue;
// the caller passes r + 512) are taken apart into their 2 T tiles and dealt, longest first, onto the 64 workgroup slots of their XCD
// (longest-processing-time rule; a workgroup then runs up to `per` = 3-4 list entries): the two last waves of cfg2 become one of
// makespan 6 instead of 2 x 5.  Inside the tail the tiles of a panel no longer walk k in lockstep (its B slabs are fetched more than
// once: HBM is not what bounds these products); everything in front of it is the paired order unchanged.
static inline int trmm_tail_makespan(int nbm, int tail_units_per_xcd, int bins) {      // LPT makespan (k blocks) of one XCD's tail
  std::vector<int> len;
  const int U = (nbm + 1) / 2;
  for (int i = 0; i < tail_units_per_xcd; ++i) {       // unit u of a panel = tiles of u + 1 and nbm - u k blocks (the middle one alone when nbm is odd)
    const int u = i % U, lo = u + 1, hi = nbm - u;
    len.push_back(hi);
    if (lo != hi) len.push_back(lo);
  }
  std::sort(len.begin(), len.end(), [](int a, int b) { return a > b; });
  std::vector<int> load(bins, 0);
  for (int l : len) *std::min_element(load.begin(), load.end()) += l;
  return *std::max_element(load.begin(), load.end());
}
// the list itself (host only: zigp_test_trmm_list checks its coverage without a GPU); returns the entries per workgroup
static int build_trmm_list(bool lower, int nbm, int nbn, bool paired, int tail_units, int tail_bins, std::vector<GemmTile>& v) {
  const int kb = BM / BK;
  if (!paired) {
    if (lower) { for (int bi = nbm - 1; bi >= 0; --bi) for (int bj = 0; bj < nbn; ++bj) v.push_back(mk_tile(bi, bj, 0, (bi + 1) * kb)); }
    else { for (int bi = 0; bi < nbm; ++bi) for (int bj = 0; bj < nbn; ++bj) v.push_back(mk_tile(bi, bj, bi * kb, nbm * kb)); }
    return 1;
  }
  // per-XCD queues of units (2 entries each); a tail is re-dealt per queue into <= 64 workgroups of up to `per` entries
  const int U = (nbm + 1) / 2;
  auto tile = [&](int bi, int bj, int dir) {
    GemmTile t = lower ? mk_tile(bi, bj, 0, (bi + 1) * kb) : mk_tile(bi, bj, bi * kb, nbm * kb);
    t.kdir = dir;
    return t;
  };
  std::vector<GemmTile> q[8];
  for (int bj = 0; bj < nbn; ++bj)
    for (int u = 0; u < U; ++u) {
      const int lo = u, hi = nbm - 1 - u;                 // lo has the short k range for lower, hi for upper
      std::vector<GemmTile>& dst = q[bj % 8];
      if (lo == hi) { dst.push_back(tile(lo, bj, lower ? -1 : 1)); dst.push_back(mk_tile(0, 0, 0, 0)); continue; }
      if (lower) { dst.push_back(tile(lo, bj, 1)); dst.push_back(tile(hi, bj, -1)); }
      else { dst.push_back(tile(hi, bj, -1)); dst.push_back(tile(lo, bj, 1)); }
    }
  size_t longest = 0;
  for (int x = 0; x < 8; ++x) longest = std::max(longest, q[x].size());
  int per = 2;
  std::vector<std::vector<GemmTile>> bins[8];      // the re-dealt tail of each queue
  const int tu = std::min<int>(tail_units / 8, (int)(longest / 2));
  if (tu > 0) {
    const int BINS = std::max(1, std::min(64, tail_bins));   // of the 64 resident workgroups per XCD, the share of this list's tail
    for (int x = 0; x < 8; ++x) {
      // the queue's units at launch-order positions >= longest / 2 - tu (shorter queues -- nbn not a multiple of 8 -- end in padding there)
      const size_t first = std::min(q[x].size(), (longest / 2 - (size_t)tu) * 2);
      std::vector<GemmTile> t(q[x].begin() + first, q[x].end());
      q[x].resize(first);
      t.erase(std::remove_if(t.begin(), t.end(), [](const GemmTile& a) { return a.kend <= a.kbeg; }), t.end());
      std::stable_sort(t.begin(), t.end(), [](const GemmTile& a, const GemmTile& b) { return a.kend - a.kbeg > b.kend - b.kbeg; });
      bins[x].assign(std::min<size_t>((size_t)BINS, t.size()), std::vector<GemmTile>());
      std::vector<int> load(bins[x].size(), 0);
      for (const GemmTile& a : t) {
        const size_t b = std::min_element(load.begin(), load.end()) - load.begin();
        bins[x][b].push_back(a);
        load[b] += a.kend - a.kbeg;
      }
      // longest workgroups first in the dispatch order
      std::vector<size_t> order(bins[x].size());
      for (size_t i = 0; i < order.size(); ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return load[a] > load[b]; });
      std::vector<std::vector<GemmTile>> sorted;
      for (size_t i : order) sorted.push_back(bins[x][i]);
      bins[x] = sorted;
      for (const auto& b : bins[x]) per = std::max(per, (int)b.size());
    }
  }
  {
    size_t reg = 0, nb = 0;
    for (int x = 0; x < 8; ++x) { reg = std::max(reg, q[x].size() / 2); nb = std::max(nb, bins[x].size()); }
    for (size_t i = 0; i < reg; ++i)                         // launch position p = 8 * i + x  ->  XCD x
      for (int x = 0; x < 8; ++x)
        for (int e = 0; e < per; ++e) v.push_back(e < 2 && 2 * i + e < q[x].size() ? q[x][2 * i + e] : mk_tile(0, 0, 0, 0));
    for (size_t i = 0; i < nb; ++i)
      for (int x = 0; x < 8; ++x)
        for (int e = 0; e < per; ++e) v.push_back(i < bins[x].size() && e < (int)bins[x][i].size() ? bins[x][i][e] : mk_tile(0, 0, 0, 0));
  }
  return per;
}
static int tiles_trmm(zigp_ctx* c, bool lower, int nbm, int nbn, bool paired, TileList& tl, int tail_units = 0, int tail_bins = 64) {
  const std::string key = std::string(lower ? "trl:" : "tru:") + std::to_string(nbm) + ":" + std::to_string(nbn) + (paired ? ":p" : ":l") +
                          (tail_units > 0 ? ":t" + std::to_string(tail_units) + ":" + std::to_string(tail_bins) : std::string());
  auto it = c->tiles.find(key);
  if (it != c->tiles.end()) { tl = it->second; return 0; }
  std::vector<GemmTile> list;
  const int per = build_trmm_list(lower, nbm, nbn, paired, tail_units, tail_bins, list);
  return get_tiles(c, key, [&](std::vector<GemmTile>& v) { v = list; }, tl, per);
}
// The paired order has nbn * ceil(nbm / 2) units of EQUAL length per latent: it pays (2.2x fewer bytes, +6 % on the triangular
// products) where the units of BOTH latents, launched together (run_gemm2), fill whole waves of the 512 resident workgroups --
// cfg3: 2 x 256 panels x 4 units = 4 waves exactly; cfg2: 2 x 784 x 2 = 3136 units = 6.1 waves, 0.875 full, still -1.3 % against LPT
// (profiles/r05l_ab_merge_fg.log); launched per latent the same units are 3.06 waves and lose (+1.6 %).  Otherwise LPT, one launch
// per latent, whose tiles of mixed length pack the tail.
static inline bool trmm_paired_pays(int units_both_latents) {
  const int slots = 512, waves = (units_both_latents + slots - 1) / slots;
  return units_both_latents >= slots && (double)units_both_latents / ((double)waves * slots) >= 0.85;
}
static int tiles_trmm_lower(zigp_ctx* c, int nbm, int nbn, TileList& tl, bool paired, int tail_units = 0, int tail_bins = 64) { return tiles_trmm(c, true, nbm, nbn, paired, tl, tail_units, tail_bins); }
static int tiles_trmm_upper(zigp_ctx* c, int nbm, int nbn, TileList& tl, bool paired, int tail_units = 0, int tail_bins = 64) { return tiles_trmm(c, false, nbm, nbn, paired, tl, tail_units, tail_bins); }
// Tail plan of a merged launch [set 0's units | set 1's units] (unit counts multiples of 8, nbm + 1 k blocks per unit) over 512 slots:
// with a remainder r the last r + 512 units -- the end of set 1 and, where set 1 is shorter than that, the end of set 0 as well -- become
// ONE wave of 512 LPT workgroups (64 per XCD, shared between the two sets in proportion to their tails' work) whenever that wave is
// shorter than the two it replaces.  All zeros: leave the lists as they are.
struct TrmmTail { int units[2], bins[2]; };
static inline TrmmTail trmm_tail_plan(int units0, int nbm0, int units1, int nbm1) {
  TrmmTail none = {{0, 0}, {64, 64}}, t = none;
  const int slots = 512, total = units0 + units1, r = total % slots;
  if (r == 0 || total <= slots) return none;
  const int T = r + slots;
  t.units[1] = std::min(T, units1);
  t.units[0] = T - t.units[1];
  if (t.units[0] > units0 || t.units[0] % 8 || t.units[1] % 8) return none;
  const double w0 = (double)t.units[0] * (nbm0 + 1), w1 = (double)t.units[1] * (nbm1 + 1);
  t.bins[1] = t.units[0] == 0 ? 64 : std::max(1, std::min(63, (int)std::lround(64.0 * w1 / (w0 + w1))));
  t.bins[0] = 64 - t.bins[1];
  int after = t.units[1] ? trmm_tail_makespan(nbm1, t.units[1] / 8, t.bins[1]) : 0;
  if (t.units[0]) after = std::max(after, trmm_tail_makespan(nbm0, t.units[0] / 8, t.bins[0]));
  const int before = (nbm1 + 1) + (t.units[0] ? nbm0 + 1 : nbm1 + 1);       // the two lockstep waves the tail replaces
  return after < before ? t : none;
}
// Split-K plan of the symmetric rank-N update.  Off-diagonal tiles are cut into So slices, diagonal tiles (the balanced lower-triangle
// path of zigp_gemm.h: 36 of 64 sub-tile products per slice, two slices per barrier -- ~0.6 of a full tile's time) into Sd = So / 2
// slices of twice the length, So a multiple of 16: the k range then falls into 8 windows, one per XCD, each holding So / 8 slices of
// every off-diagonal tile and Sd / 8 of every diagonal one -- an XCD's workgroups all stream the same eighth of the A1 panel through its
// L2 (r3: with 15 and 11 slices, unaligned, the launch moved 0.95 GB instead of 0.60; profiles/r03k_pmc_hbm_traffic.json).
// (n_off + n_diag / 2) So <= 512 resident workgroups: M = 1024: So = 16, Sd = 8, 28 x 16 + 8 x 8 = 512; M = 512: 64 / 32, 6 x 64 + 4 x 32 = 512.
// A diagonal workgroup runs ~1.2 x as long as an off-diagonal one; the list puts the diagonal tiles first in each XCD's queue, so that
// each shares its CU with an off-diagonal workgroup and inherits the whole matrix pipe when that one is done.
// (r3, measured and dropped: the update reading both operands from a transposed copy A1^T written by the A1 epilogue, i.e. on the 8-wave
// m-contiguous kernel: the update itself 53.9 -> 56.0 TF, but the scattered transposed stores cost the A1 product 10 % -- 60.4 -> 54.3 TF --
// and the step 9 ms: profiles/r03f_ab_syrk_a1t.log.)
struct SyrPlan { int So, Sd; int planes() const { return std::max(So, Sd); } };
static inline SyrPlan syr_plan(int nbm) {
  const int n_off = nbm * (nbm - 1) / 2, n_d = nbm, slots = 512;
  int So = (int)(slots / (n_off + 0.5 * n_d)) / 16 * 16;
  So = std::max(16, std::min(64, So));
  while (So > 16 && n_off * So + n_d * (So / 2) > slots) So -= 16;
  return SyrPlan{So, So / 2};
}
// Lower-triangular output tiles x split-K slices over nk k-steps.  Launch position p runs on XCD p % 8 (observed round-robin dispatch;
// speed only): XCD x is handed the tiles whose k range lies in the x-th eighth of the k range (both slice counts multiples of 8), else
// -- no diagonal path -- a contiguous run of the k-major tile order.
static int tiles_syr2k(zigp_ctx* c, int nbm, int nk, SyrPlan sp, TileList& tl) {
  return get_tiles(c, "syr:" + std::to_string(nbm) + ":" + std::to_string(nk) + ":" + std::to_string(sp.So) + ":" + std::to_string(sp.Sd), [&](std::vector<GemmTile>& v) {
    auto entry = [&](int bi, int bj, int s, int S) { return mk_tile(bi, bj, (int)((int64_t)nk * s / S), (int)((int64_t)nk * (s + 1) / S), s); };
    if (sp.So % 8 == 0 && sp.Sd % 8 == 0) {
      std::vector<GemmTile> q[8];
      for (int x = 0; x < 8; ++x) {
        for (int s = x * sp.Sd / 8; s < (x + 1) * sp.Sd / 8; ++s)
          for (int bi = 0; bi < nbm; ++bi) q[x].push_back(entry(bi, bi, s, sp.Sd));
        for (int s = x * sp.So / 8; s < (x + 1) * sp.So / 8; ++s)
          for (int bi = 0; bi < nbm; ++bi)
            for (int bj = 0; bj < bi; ++bj) q[x].push_back(entry(bi, bj, s, sp.So));
      }
      for (size_t e = 0; e < q[0].size(); ++e)          // all eight queues have the same length
        for (int x = 0; x < 8; ++x) v.push_back(q[x][e]);
      return;
    }
    std::vector<GemmTile> t;
    for (int bi = 0; bi < nbm; ++bi)
      for (int bj = 0; bj <= bi; ++bj) {
        const int S = (bi == bj) ? sp.Sd : sp.So;
        for (int s = 0; s < S; ++s) t.push_back(entry(bi, bj, s, S));
      }
    std::stable_sort(t.begin(), t.end(), [](const GemmTile& a, const GemmTile& b) { return a.kbeg < b.kbeg; });
    const int n = (int)t.size(), per = (n + 7) / 8;
    v.reserve(n);
    for (int p = 0; (int)v.size() < n; ++p) {
      const int idx = per * (p % 8) + p / 8;
      if (p / 8 < per && idx < n) v.push_back(t[idx]);
    }
  }, tl);
}
// Full product over column panels, for the chunk loop: panel bj goes to XCD bj % 8 (launch position p runs on XCD p % 8), its nbm row
// blocks are consecutive entries of that XCD's queue -- they start together and walk the panel's k range in lockstep, so a panel's
// slab is fetched into that L2 once.  All tiles have the same length: nbm * nbn tiles over the 512 resident workgroups.
static int tiles_full_xcd(zigp_ctx* c, int nbm, int nbn, int nk, TileList& tl) {
  return get_tiles(c, "fullx:" + std::to_string(nbm) + ":" + std::to_string(nbn) + ":" + std::to_string(nk), [&](std::vector<GemmTile>& v) {
    std::vector<GemmTile> q[8];
    for (int bj = 0; bj < nbn; ++bj)
      for (int bi = 0; bi < nbm; ++bi) q[bj % 8].push_back(mk_tile(bi, bj, 0, nk));
    size_t longest = 0;
    for (int x = 0; x < 8; ++x) longest = std::max(longest, q[x].size());
    for (size_t e = 0; e < longest; ++e)
      for (int x = 0; x < 8; ++x) v.push_back(e < q[x].size() ? q[x][e] : mk_tile(0, 0, 0, 0));
  }, tl);
}
static int tiles_full(zigp_ctx* c, int nbm, int nbn, int nk, TileList& tl) {
  return get_tiles(c, "full:" + std::to_string(nbm) + ":" + std::to_string(nbn) + ":" + std::to_string(nk), [&](std::vector<GemmTile>& v) {
    for (int bi = 0; bi < nbm; ++bi)
      for (int bj = 0; bj < nbn; ++bj) v.push_back(mk_tile(bi, bj, 0, nk));
  }, tl);
}

// Deferred Cholesky status: request_info() stages the device flag with the other results of the call, info_result() reads
// it after the call's final synchronisation.  A failed factorisation leaves L / W unfinished; the kernels that follow then
// compute garbage from them, but all their indexing is data independent, so nothing needs a mid-call round trip.
static int request_info(zigp_ctx* c, int** out) {
  int* h = (int*)c->pinned.alloc(sizeof(int));
  if (!h) { c->err = "hipHostMalloc failed for the staging arena"; return ZIGP_EHIP; }
  ZIGP_HIP(c, hipMemcpyAsync(h, c->d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  *out = h;
  return 0;
}
static int info_result(zigp_ctx* c, const int* hinfo, const char* what) {
  const int h = *hinfo;
  if (h != 0) {
    c->info = h;
    char b[256];
    snprintf(b, sizeof(b), "Cholesky decomposition was not successful for %s: the input might not be positive definite (pivot %d)", what, h);
    c->err = b;
    return ZIGP_ENOTPD;
  }
  return 0;
}
static int check_info(zigp_ctx* c, const char* what) {
  int h = 0;
  ZIGP_HIP(c, hipMemcpyAsync(&h, c->d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  if (h != 0) {
    c->info = h;
    char b[256];
    snprintf(b, sizeof(b), "Cholesky decomposition was not successful for %s: the input might not be positive definite (pivot %d)", what, h);
    c->err = b;
    return ZIGP_ENOTPD;
  }
  return 0;
}

// smallest pivot accepted in a Cholesky of an RBF Kuu (constant diagonal variance + jitter): see potrf_diag_lds
// (rtol = 8 by default; zigp_set_pivot_rtol(ctx, 0) gives the bare `pivot > 0` test of LAPACK / Eigen / tf.cholesky)
static inline double pivot_tol(double var, double jitter, double rtol = 8.0) { return rtol * 2.220446049250313e-16 * (var + jitter); }

// ------------------------------------------------------------------------------------------------
// L = chol(A) in place in `Lb` (which holds a copy of A on entry), W = L^-1.  Mp multiple of 128.
// ------------------------------------------------------------------------------------------------
// Mreal = rows that are not identity padding (diagonal blocks factor only the panels that hold real rows)
// One or two factorisations at a time: the launches of job 0 go to streams[0], those of job 1 to streams[1], ALTERNATING step by
// step.  A chain is ~25 dependent launches of 5-50 us; enqueued one chain after the other, the second stream started ~100-400 us
// late (the host was still enqueueing the first), and the M x M forward of the two latents took that much longer than one chain.
// split-K over an explicit tile set (the panel solve / trailing update of the blocked factorisation and the block products of the triangular
// inverse below: 4-28 tiles of 8-32 staged steps, i.e. 4-28 workgroups that each run 15-70 us at one CU's rate).  Every tile's k range is cut into S slices, slice s writes its partial tile into
// plane s, and k_sk_finish_tiles adds the planes IN ORDER (deterministic) and stores alpha * sum into C (accum: adds it to C) -- at these
// tiles only.
__global__ void __launch_bounds__(256)
k_sk_finish_tiles(const double* __restrict__ planes, const GemmTile* __restrict__ tiles, int S, int64_t Mp, double alpha, int accum, double* __restrict__ out) {
  const GemmTile tl = tiles[blockIdx.x >> 6];                          // the first `count` entries of the list are slice 0 of every tile
  const int e = ((blockIdx.x & 63) << 8) + threadIdx.x;                // element of the 128 x 128 tile
  const int64_t idx = ((int64_t)tl.bi * BM + (e >> 7)) * Mp + (int64_t)tl.bj * BN + (e & 127);
  double v = 0.0;
  for (int s = 0; s < S; ++s) v += planes[(int64_t)s * Mp * Mp + idx];
  out[idx] = accum ? fma(alpha, v, out[idx]) : alpha * v;
}
template <int AL, int BL, class Gen>
static int run_gemm_sk_tiles(zigp_ctx* c, DevBuf& planes, const std::string& key, Gen gen, const double* A, const double* B, double* C, int64_t Mp, double alpha,
                             bool accum = false) {
  std::vector<GemmTile> base;
  gen(base);
  if (base.empty()) return 0;
  int minlen = 1 << 30;
  for (const GemmTile& t : base) minlen = std::min(minlen, t.kend - t.kbeg);
  const int count = (int)base.size(), S = std::max(1, std::min(std::min(8, minlen), 512 / count));
  TileList tl;
  ZIGP_TRY(get_tiles(c, "skt:" + key + ":" + std::to_string(S), [&](std::vector<GemmTile>& v) {
    for (int s = 0; s < S; ++s)
      for (const GemmTile& t : base) {
        const int len = t.kend - t.kbeg;
        v.push_back(mk_tile(t.bi, t.bj, t.kbeg + (int)((int64_t)len * s / S), t.kbeg + (int)((int64_t)len * (s + 1) / S), s));
      }
  }, tl));
  ZIGP_ENSURE(c, planes, (size_t)S * Mp * Mp);
  GemmArgs g = mk_args(A, Mp, B, Mp, planes.p, Mp);
  g.slice_stride = Mp * Mp;
  ZIGP_TRY((run_gemm<AL, BL, false>(c, tl, g, EpiStore())));
  hipLaunchKernelGGL(k_sk_finish_tiles, dim3(64 * count), dim3(256), 0, c->stream, planes.p, tl.d, S, Mp, alpha, accum ? 1 : 0, C);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

// prepared: the caller has zeroed W and the strictly-upper blocks of L already (k_kuu_setup writes both next to Kuu); planes: buffer for
// the split-K block products of the inverse (nullptr: plain launches, one workgroup per tile)
struct PotrfJob { double* L; double* W; double* T; int Mp; bool want_W; int Mreal; double piv_tol; bool prepared; DevBuf* planes; };
static int potrf_trtri_jobs(zigp_ctx* c, int njobs, const PotrfJob* jobs, const hipStream_t* streams) {
  hipStream_t const saved = c->stream;
  struct Restore { zigp_ctx* c; hipStream_t s; ~Restore() { c->stream = s; } } restore{c, saved};
  const int kb = BM / BK;  // k-steps per block
  const size_t shm = sizeof(double) * PB * PBLD;
  int nbmax = 0;
  for (int q = 0; q < njobs; ++q) nbmax = std::max(nbmax, jobs[q].Mp / BM);
  // c->d_info is cleared by the caller (several factorizations may share one check_info)
  for (int q = 0; q < njobs; ++q) {
    if (jobs[q].prepared) continue;
    c->stream = streams[q];
    ZIGP_HIP(c, hipMemsetAsync(jobs[q].W, 0, sizeof(double) * jobs[q].Mp * jobs[q].Mp, c->stream));
  }
  for (int j = 0; j < nbmax; ++j) {
    for (int step = 0; step < 3; ++step)
      for (int q = 0; q < njobs; ++q) {
        const PotrfJob& J = jobs[q];
        const int Mp = J.Mp, nb = Mp / BM, Mreal = (J.Mreal < 0 || J.Mreal > Mp) ? Mp : J.Mreal;
        if (j >= nb) continue;
        c->stream = streams[q];
        double* Lb = J.L; double* Wb = J.W;
        if (step == 0) {
          double* Ajj = Lb + (int64_t)j * BM * Mp + (int64_t)j * BM;
          double* Wjj = Wb + (int64_t)j * BM * Mp + (int64_t)j * BM;
          const int nreal_j = std::max(0, std::min(BM, Mreal - j * BM));
          hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(1024), shm, c->stream, Ajj, Ajj, Wjj, (int64_t)Mp, j * BM, c->d_info, (nreal_j + PNB - 1) / PNB, J.piv_tol);
          ZIGP_HIP(c, hipGetLastError());
        } else if (j + 1 < nb && step == 1) {
          auto genp = [&](std::vector<GemmTile>& v) {
            for (int bi = j + 1; bi < nb; ++bi) v.push_back(mk_tile(bi, j, j * kb, (j + 1) * kb));
          };
          const std::string tag = "po_p:" + std::to_string(nb) + ":" + std::to_string(j);
          // L[bi][j] = A[bi][j] * W_jj^T   (in place: each tile reads only itself and W_jj)
          if (J.planes) { ZIGP_TRY((run_gemm_sk_tiles<LAY_KCONTIG, LAY_KCONTIG>(c, *J.planes, tag, genp, Lb, Wb, Lb, Mp, 1.0))); continue; }
          TileList tp;
          ZIGP_TRY(get_tiles(c, tag, genp, tp));
          ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_KCONTIG, false>(c, tp, mk_args(Lb, Mp, Wb, Mp, Lb, Mp), EpiStore())));
        } else if (j + 1 < nb && step == 2) {
          auto gens = [&](std::vector<GemmTile>& v) {
            for (int bi = j + 1; bi < nb; ++bi)
              for (int bj = j + 1; bj <= bi; ++bj) v.push_back(mk_tile(bi, bj, j * kb, (j + 1) * kb));
          };
          const std::string tag = "po_s:" + std::to_string(nb) + ":" + std::to_string(j);
          // A[bi][bj] -= L[bi][j] L[bj][j]^T
          if (J.planes) { ZIGP_TRY((run_gemm_sk_tiles<LAY_KCONTIG, LAY_KCONTIG>(c, *J.planes, tag, gens, Lb, Lb, Lb, Mp, -1.0, true))); continue; }
          TileList ts;
          ZIGP_TRY(get_tiles(c, tag, gens, ts));
          ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_KCONTIG, false>(c, ts, mk_args(Lb, Mp, Lb, Mp, Lb, Mp, -1.0), EpiAccum())));
        }
      }
  }
  // zero the strictly-upper blocks of L (they still hold the copy of A)
  for (int bi = 0; bi + 1 < nbmax; ++bi)
    for (int q = 0; q < njobs; ++q) {
      const int Mp = jobs[q].Mp, nb = Mp / BM;
      if (bi + 1 >= nb || jobs[q].prepared) continue;
      c->stream = streams[q];
      ZIGP_HIP(c, hipMemset2DAsync(jobs[q].L + (int64_t)bi * BM * Mp + (int64_t)(bi + 1) * BM, sizeof(double) * Mp, 0,
                                   sizeof(double) * (size_t)(Mp - (bi + 1) * BM), BM, c->stream));
    }
  // W by recursive doubling over diagonal-block groups: W21 = -W22 (L21 W11)
  for (int b = 1; b < nbmax; b *= 2)
    for (int step = 0; step < 2; ++step)
      for (int q = 0; q < njobs; ++q) {
        const PotrfJob& J = jobs[q];
        const int Mp = J.Mp, nb = Mp / BM;
        if (!J.want_W || b >= nb) continue;
        c->stream = streams[q];
        auto gen1 = [&](std::vector<GemmTile>& v) {
          for (int lo = 0; lo < nb; lo += 2 * b) {
            const int mid = lo + b, hi = std::min(lo + 2 * b, nb);
            if (mid >= nb) continue;
            for (int bi = mid; bi < hi; ++bi)
              for (int bj = lo; bj < mid; ++bj) v.push_back(mk_tile(bi, bj, bj * kb, mid * kb));
          }
        };
        auto gen2 = [&](std::vector<GemmTile>& v) {
          for (int lo = 0; lo < nb; lo += 2 * b) {
            const int mid = lo + b, hi = std::min(lo + 2 * b, nb);
            if (mid >= nb) continue;
            for (int bi = mid; bi < hi; ++bi)
              for (int bj = lo; bj < mid; ++bj) v.push_back(mk_tile(bi, bj, mid * kb, (bi + 1) * kb));
          }
        };
        const std::string tag = std::to_string(nb) + ":" + std::to_string(b);
        if (step == 0) {          // T = L21 W11
          if (J.planes) { ZIGP_TRY((run_gemm_sk_tiles<LAY_KCONTIG, LAY_MNCONTIG>(c, *J.planes, "tri1:" + tag, gen1, J.L, J.W, J.T, Mp, 1.0))); continue; }
          TileList t1;
          ZIGP_TRY(get_tiles(c, "tri1:" + tag, gen1, t1));
          ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_MNCONTIG, false>(c, t1, mk_args(J.L, Mp, J.W, Mp, J.T, Mp), EpiStore())));
        } else {                  // W21 = -W22 T
          if (J.planes) { ZIGP_TRY((run_gemm_sk_tiles<LAY_KCONTIG, LAY_MNCONTIG>(c, *J.planes, "tri2:" + tag, gen2, J.W, J.T, J.W, Mp, -1.0))); continue; }
          TileList t2;
          ZIGP_TRY(get_tiles(c, "tri2:" + tag, gen2, t2));
          ZIGP_TRY((run_gemm<LAY_KCONTIG, LAY_MNCONTIG, false>(c, t2, mk_args(J.W, Mp, J.T, Mp, J.W, Mp, -1.0), EpiStore())));
        }
      }
  return 0;
}
static int potrf_trtri(zigp_ctx* c, double* Lb, double* Wb, double* Tb, int Mp, bool want_W, int Mreal = -1, double piv_tol = 0.0) {
  const PotrfJob job = {Lb, Wb, Tb, Mp, want_W, Mreal, piv_tol, false, nullptr};
  const hipStream_t st = c->stream;
  return potrf_trtri_jobs(c, 1, &job, &st);
}

// ------------------------------------------------------------------------------------------------
// split-K for the O(M^3) products of the M x M stage.  A 1024^3 product is 64 output tiles of 64 BK steps: 64 workgroups on a
// 256-CU chip, 105-150 us per launch, and the reverse pass chains nine of them per latent.  Each tile's k range is cut into S slices
// (S * tiles ~ the 512 resident workgroups), the slices write their partial tile to plane s, and k_sk_finish adds the planes in
// slice order (fixed order: bit-stable) and applies what used to be the GEMM epilogue.
// ------------------------------------------------------------------------------------------------
enum { SK_STORE = 0, SK_ACCUM = 1, SK_PHI = 2 };
template <int POST>
__global__ void __launch_bounds__(256)
k_sk_finish(const double* __restrict__ planes, int S, int64_t Mp, double alpha, int lower_only, double* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Mp * Mp) return;
  const int64_t i = idx / Mp, j = idx - i * Mp;
  if (lower_only && (j / BN > i / BM)) {            // tiles above the diagonal were not computed: their value is DEFINED as zero
    if (POST != SK_ACCUM) out[idx] = 0.0;           // (what the empty-k tiles of a plain GEMM launch would have stored)
    return;
  }
  double v = 0.0;
  for (int s = 0; s < S; ++s) v += planes[(int64_t)s * Mp * Mp + idx];
  v *= alpha;
  if (POST == SK_STORE) out[idx] = v;
  else if (POST == SK_ACCUM) out[idx] += v;
  else out[idx] = (j < i) ? v : ((j == i) ? 0.5 * v : 0.0);
}

// krange(bi, bj, &kbeg, &kend) in units of BK; kend <= kbeg: the tile is not computed
template <int AL, int BL, class KRange>
static int run_gemm_sk(zigp_ctx* c, DevBuf& planes, const std::string& key, int nb, KRange krange, const double* A, const double* B, double* C,
                       int64_t Mp, int post, double alpha, bool lower_only) {
  int count = 0, minlen = 1 << 30;
  for (int bi = 0; bi < nb; ++bi)
    for (int bj = 0; bj < nb; ++bj) {
      int k0, k1; krange(bi, bj, k0, k1);
      if (k1 > k0) { ++count; minlen = std::min(minlen, k1 - k0); }
    }
  if (count == 0) return 0;
  const int S = std::max(1, std::min(std::min(8, minlen), 512 / count));
  TileList tl;
  ZIGP_TRY(get_tiles(c, "sk:" + key + ":" + std::to_string(nb) + ":" + std::to_string(S), [&](std::vector<GemmTile>& v) {
    for (int s = 0; s < S; ++s)
      for (int bi = 0; bi < nb; ++bi)
        for (int bj = 0; bj < nb; ++bj) {
          int k0, k1; krange(bi, bj, k0, k1);
          if (k1 <= k0) continue;
          const int len = k1 - k0;
          v.push_back(mk_tile(bi, bj, k0 + (int)((int64_t)len * s / S), k0 + (int)((int64_t)len * (s + 1) / S), s));
        }
  }, tl));
  ZIGP_ENSURE(c, planes, (size_t)S * Mp * Mp);
  GemmArgs g = mk_args(A, Mp, B, Mp, planes.p, Mp);
  g.slice_stride = Mp * Mp;
  ZIGP_TRY((run_gemm<AL, BL, false>(c, tl, g, EpiStore())));
  const dim3 grid((unsigned)((Mp * Mp + 255) / 256));
  if (post == SK_STORE) hipLaunchKernelGGL(k_sk_finish<SK_STORE>, grid, dim3(256), 0, c->stream, planes.p, S, Mp, alpha, lower_only ? 1 : 0, C);
  else if (post == SK_ACCUM) hipLaunchKernelGGL(k_sk_finish<SK_ACCUM>, grid, dim3(256), 0, c->stream, planes.p, S, Mp, alpha, lower_only ? 1 : 0, C);
  else hipLaunchKernelGGL(k_sk_finish<SK_PHI>, grid, dim3(256), 0, c->stream, planes.p, S, Mp, alpha, lower_only ? 1 : 0, C);
  ZIGP_HIP(c, hipGetLastError());
  return 0;
}

static inline KernHyp make_hyp(const double* ell, double var, int D) {
  KernHyp h;
  for (int d = 0; d < MAXD; ++d) h.inv_ell[d] = (d < D) ? 1.0 / ell[d] : 0.0;
  h.var = var; h.D = D;
  return h;
}

static inline KufHyp make_kuf_hyp(const double* ell, double var, int D) {
  KufHyp h;
  for (int d = 0; d < MAXD; ++d) h.scale[d] = (d < D) ? KUF_C * (1.0 / ell[d]) : 0.0;
  h.var = var;
  return h;
}

// ---- small transfers through the pinned arena (zigp_ctx.h) ----
// Start of an API call that stages transfers: make sure nothing of an earlier call (one that returned an error before its
// final synchronisation, say) is still reading or writing the arena, then rewind it.
static int begin_staged_call(zigp_ctx* c) {
  ZIGP_HIP(c, hipStreamSynchronize(c->stream_main));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream2));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream3));
  c->pinned.reset();
  return 0;
}
static inline double* pinned_doubles(zigp_ctx* c, size_t n) { return (double*)c->pinned.alloc(sizeof(double) * (n ? n : 1)); }
#define ZIGP_PINNED(ctx, ptr, n)                                                         \
  double* ptr = pinned_doubles((ctx), (n));                                              \
  if (!ptr) { (ctx)->err = "hipHostMalloc failed for the staging arena"; return ZIGP_EHIP; }
// device <- host image of n doubles zero-padded to npad: one staged copy, no memset launch
static int upload_padded(zigp_ctx* c, DevBuf& b, const double* src, size_t n, size_t npad) {
  ZIGP_ENSURE(c, b, npad);
  ZIGP_PINNED(c, h, npad);
  if (n) memcpy(h, src, sizeof(double) * n);
  if (npad > n) memset(h + n, 0, sizeof(double) * (npad - n));
  ZIGP_HIP(c, hipMemcpyAsync(b.p, h, sizeof(double) * npad, hipMemcpyHostToDevice, c->stream));
  return 0;
}
// host <- device: the returned pinned pointer holds the data after the next synchronisation of c->stream
static int download(zigp_ctx* c, const double* dev, size_t n, double** out) {
  ZIGP_PINNED(c, h, n);
  ZIGP_HIP(c, hipMemcpyAsync(h, dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  *out = h;
  return 0;
}


}  // namespace zigp
