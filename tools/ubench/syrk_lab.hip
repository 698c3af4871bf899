// Knock-out ladder of the symmetric rank-N update (round 6): C1 += A1 diag(gv) A1^T on the lower triangle, split-K
// into planes -- gemm_f64_kernel<LAY_KCONTIG, LAY_KCONTIG, 2, KSCALE = true, TRI_C_LOWER, 4 waves, EpiAccum> with the product's tile list
// (zigp_host.h tiles_syr2k / syr_plan), the worst GEMM class of the step (62 TFLOP/s = 0.79 of the fp64 MFMA peak).  Every rung is the
// PRODUCT kernel with one thing taken away (template arguments only; zigp_gemm.h is untouched):
//   k-scale off   KSCALE = false: no 128-byte gv slice per staged step, no v_mul_f64 of the B fragments (4 per 16 MFMAs on the MFMA's ALU)
//   RMW off       EpiStore instead of EpiAccum: the planes are written, not read-added-written (the result is then one chunk's, not the sum)
//   no stores     an empty epilogue
//   8 planes      So = 8, Sd = 4: 256 workgroups, the two latents' planes (2 x 67 MB) would fit the 256 MB Infinity Cache
//   8 planes x 2  the same for TWO panels in one launch (512 workgroups, as run_gemm2 merges the forward products)
//   32 planes     So = 32, Sd = 16: 1024 workgroups
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../zero-inflated-gp_amd/csrc syrk_lab.hip -o syrk_lab        Run: syrk_lab [Nc=32768] [reps=30] [rounds=3]
#include "zigp_gemm.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <string>
#include <algorithm>
#include <functional>
using namespace zigp;
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e_),__LINE__); exit(1);} }while(0)

struct EpiNone {
  template <int TM, int TN> __device__ __forceinline__ void operator()(const double (&acc)[TM][TN][4], const EpiCtx& e) const {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(acc[tm][tn][r]));      // every accumulator is "used": no MFMA is dead code
  }
};

static GemmTile mk(int bi, int bj, int k0, int k1, int slice) {
  GemmTile t; t.bi = bi; t.bj = bj; t.kbeg = k0; t.kend = k1; t.slice = slice; t.kdir = 1; t.pad1 = t.pad2 = 0; return t;
}
// zigp_host.h tiles_syr2k for slice counts that are multiples of 8: XCD x gets the x-th eighth of the k range of every tile, diagonal tiles first
static std::vector<GemmTile> tiles_syr(int nbm, int nk, int So, int Sd) {
  auto entry = [&](int bi, int bj, int s, int S) { return mk(bi, bj, (int)((int64_t)nk * s / S), (int)((int64_t)nk * (s + 1) / S), s); };
  std::vector<GemmTile> q[8], v;
  for (int x = 0; x < 8; ++x) {
    for (int s = x * Sd / 8; s < (x + 1) * Sd / 8; ++s)
      for (int bi = 0; bi < nbm; ++bi) q[x].push_back(entry(bi, bi, s, Sd));
    for (int s = x * So / 8; s < (x + 1) * So / 8; ++s)
      for (int bi = 0; bi < nbm; ++bi)
        for (int bj = 0; bj < bi; ++bj) q[x].push_back(entry(bi, bj, s, So));
  }
  for (size_t e = 0; e < q[0].size(); ++e)
    for (int x = 0; x < 8; ++x) v.push_back(q[x][e]);
  return v;
}
// Sd < 8 (the 8-plane plan: So = 8, Sd = 4): diagonal slices cover two XCD windows each; they go to the queue of their first window
static std::vector<GemmTile> tiles_syr_8(int nbm, int nk) {
  const int So = 8, Sd = 4;
  auto entry = [&](int bi, int bj, int s, int S) { return mk(bi, bj, (int)((int64_t)nk * s / S), (int)((int64_t)nk * (s + 1) / S), s); };
  std::vector<GemmTile> q[8], v;
  for (int x = 0; x < 8; ++x) {
    if (x % 2 == 0) for (int bi = 0; bi < nbm; ++bi) q[x].push_back(entry(bi, bi, x / 2, Sd));
    for (int bi = 0; bi < nbm; ++bi)
      for (int bj = 0; bj < bi; ++bj) q[x].push_back(entry(bi, bj, x, So));
  }
  size_t longest = 0;
  for (int x = 0; x < 8; ++x) longest = std::max(longest, q[x].size());
  for (size_t e = 0; e < longest; ++e)
    for (int x = 0; x < 8; ++x) v.push_back(e < q[x].size() ? q[x][e] : mk(0, 0, 0, 0, 0));
  return v;
}

struct Timer {
  hipEvent_t e0, e1;
  Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
  template <class F> double run(F f, int reps) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  }
};

int main(int argc, char** argv) {
  const int M = 1024, nbm = M / BM;
  const int64_t Nc = (argc > 1) ? atoll(argv[1]) : 32768;
  const int reps = (argc > 2) ? atoi(argv[2]) : 30;
  const int rounds = (argc > 3) ? atoi(argv[3]) : 3;
  const int nk = (int)(Nc / BK);
  printf("syrk_lab: planes[s][%d x %d] (+)= tril(A1 diag(gv) A1^T), A1 [%d][%lld] k-contiguous, %d launches per timing, %d rounds; flops = M^2 Nc = %.3e per launch\n",
         M, M, M, (long long)Nc, reps, rounds, (double)M * M * Nc);
  std::vector<double> hA((size_t)M * Nc), hg(Nc);
  srand(2);
  for (auto& x : hA) x = rand() / (double)RAND_MAX - 0.5;
  for (auto& x : hg) x = rand() / (double)RAND_MAX - 0.7;
  double *dA[2], *dg[2], *dP[2];
  const int PL = 32;                                         // planes allocated per panel
  for (int a = 0; a < 2; ++a) {
    CK(hipMalloc(&dA[a], sizeof(double) * hA.size())); CK(hipMalloc(&dg[a], sizeof(double) * Nc)); CK(hipMalloc(&dP[a], sizeof(double) * PL * M * M));
    CK(hipMemcpy(dA[a], hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dg[a], hg.data(), sizeof(double) * Nc, hipMemcpyHostToDevice));
    CK(hipMemset(dP[a], 0, sizeof(double) * PL * M * M));
  }
  std::vector<GemmTile> lists[3] = {tiles_syr(nbm, nk, 16, 8), tiles_syr_8(nbm, nk), tiles_syr(nbm, nk, 32, 16)};
  GemmTile* dT[3];
  for (int a = 0; a < 3; ++a) { CK(hipMalloc(&dT[a], sizeof(GemmTile) * lists[a].size())); CK(hipMemcpy(dT[a], lists[a].data(), sizeof(GemmTile) * lists[a].size(), hipMemcpyHostToDevice)); }
  auto args = [&](int panel, int list) {
    GemmArgs g; g.seg[0].A = dA[panel]; g.seg[0].B = dA[panel]; g.seg[0].lda = Nc; g.seg[0].ldb = Nc; g.tiles = dT[list]; g.per = 1;
    g.C = dP[panel]; g.ldc = M; g.slice_stride = (int64_t)M * M; g.alpha = 1.0; g.kscale = dg[panel]; return g;
  };
  const size_t shm = sizeof(double) * 2 * STAGE_DOUBLES;
  struct Var { std::string name; std::function<void()> launch; double flops; std::vector<double> ms; };
  std::vector<Var> vars;
  const double fl = (double)M * M * Nc;
#define ADD(KS, WV, EPI, LIST, TWO, NAME)                                                                                            \
  {                                                                                                                                  \
    auto kern = gemm_f64_kernel<LAY_KCONTIG, LAY_KCONTIG, 2, KS, TRI_C_LOWER, WV, EPI>;                                              \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));              \
    const GemmArgs g0 = args(0, LIST), g1 = args(1, LIST);                                                                           \
    const int n = (int)lists[LIST].size();                                                                                           \
    if (TWO) vars.push_back({NAME, [=] { hipLaunchKernelGGL(kern, dim3((n + 7) / 8 * 8 + n), dim3(64 * WV), shm, 0, g0, EPI(), g1, EPI(), n); }, 2 * fl, {}}); \
    else vars.push_back({NAME, [=] { hipLaunchKernelGGL(kern, dim3(n), dim3(64 * WV), shm, 0, g0, EPI(), g0, EPI(), n); }, fl, {}});   \
  }
  ADD(true, 4, EpiAccum, 0, false, "product: k-scale, RMW, 16/8 planes, 512 wg")
  ADD(false, 4, EpiAccum, 0, false, "  k-scale off")
  ADD(true, 4, EpiStore, 0, false, "  RMW off (store)")
  ADD(false, 4, EpiStore, 0, false, "  k-scale off, RMW off")
  ADD(false, 4, EpiNone, 0, false, "  k-scale off, no stores")
  ADD(true, 4, EpiNone, 0, false, "  k-scale on, no stores")
  ADD(true, 4, EpiAccum, 1, false, "8/4 planes, 256 wg")
  ADD(true, 4, EpiAccum, 1, true, "8/4 planes x 2 panels in one launch, 512 wg")
  ADD(true, 4, EpiAccum, 0, true, "16/8 planes x 2 panels in one launch, 1024 wg")
  ADD(true, 4, EpiAccum, 2, false, "32/16 planes, 1024 wg")
  Timer tm;
  for (auto& v : vars) { for (int i = 0; i < 2; ++i) v.launch(); CK(hipGetLastError()); }
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars) v.ms.push_back(tm.run(v.launch, reps));
  for (auto& v : vars) {
    std::vector<double> s = v.ms; std::sort(s.begin(), s.end());
    printf("%-52s best %7.3f ms %6.2f TF | median %6.2f TF\n", v.name.c_str(), s[0], v.flops / s[0] * 1e-9, v.flops / s[s.size() / 2] * 1e-9);
  }
  return 0;
}
