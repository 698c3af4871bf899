// GEMM laboratory (round 5): candidate inner loops and tile schedules for the fp64 GEMM core, measured on the shape of the chunk loop's
// dominant product (J' = Q A2: 1024 x N x K, both operands m/n-contiguous, XCD-aware tile order) next to the product kernel of zigp_gemm.h.
// The candidates ("v2") software-pipeline the LDS fragment reads by hand: the reads are inline assembly with explicit
// s_waitcnt lgkmcnt(N), the MFMAs are builtins (the compiler keeps their hazards), sched_barrier fences pin the order.
//   v2<WAVES, RD, EARLYBAR, KO>: WAVES 4 (64 x 64 wave tiles, 2 waves per SIMD) or 8 (32 x 64, 4 per SIMD);
//                            RD 0: ds_read_b64 per fragment, RD 2: ds_read_b128 per fragment pair (rows / columns interleaved in pairs);
//                            EARLYBAR: the step's one barrier sits in front of the LAST k-step's MFMAs (the next stage's first fragments
//                            and the staging loads are issued behind it and land under those MFMAs)
//                            KO (timing only, results wrong): 1 no barrier, 2 no staging loads after the first two, 4 no stores
// A workgroup runs g.per consecutive list entries; an entry with pad1 & 1 CONTINUES a tile (its accumulators start from what an earlier
// entry stored to C: bit-identical to the unsplit tile) and an entry with pad1 & 2 stores the raw partial accumulators.  With that the
// host can shift the tile boundaries of the two workgroups that share a CU against each other ("stagger"), so that one's epilogue
// stores and prologue loads fall under the other's MFMAs.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../zero-inflated-gp_amd/csrc gemm_lab.hip -o gemm_lab
// Run:   gemm_lab [N=32768] [K=1024] [reps=50] [data: 0 random, 1 zeros, 2 smooth] [rounds=2]
#include "zigp_gemm.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <string>
#include <algorithm>
#include <functional>
#include <map>
using namespace zigp;
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e_),__LINE__); exit(1);} }while(0)

typedef double d2v __attribute__((ext_vector_type(2)));
template <int N> struct IC { static constexpr int value = N; };
template <int B, int E, class F> __device__ __forceinline__ void sfor(F f) { if constexpr (B < E) { f(IC<B>{}); sfor<B + 1, E>(f); } }

template <int OFF> __device__ __forceinline__ double ds_rd64(uint32_t a) {
  double v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v;
}
template <int OFF> __device__ __forceinline__ d2v ds_rd128(uint32_t a) {
  d2v v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v;
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// staging of one m/n-contiguous operand tile ([16 k][128]) with row stride LDM doubles in LDS; as glds_tile of zigp_gemm.h
template <int WAVES, int LDM>
__device__ __forceinline__ void stage_mn(double* tile, const char* __restrict__ base, uint32_t off, int64_t chunk_stride, int wave) {
  asm volatile("" : "+v"(off));
#pragma unroll
  for (int p = 0; p < 16 / WAVES; ++p) {
    const int c = WAVES * p + wave;
    double* dst = tile + c * LDM;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(glds_pin(base + p * chunk_stride) + (uint64_t)off),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

struct Stamp { long long t0, r0, t1, r1; unsigned hwid, xcc, pad0, pad1; };

template <int WAVES, int RD, int EARLYBAR, int KO, int PF>
__global__ void __launch_bounds__(64 * WAVES, 2 * WAVES / 4)
k_v2(GemmArgs g, Stamp* stamps) {
  constexpr bool NOBAR = (KO & 1) != 0, NOGLDS = (KO & 2) != 0, NOEPI = (KO & 4) != 0, HOT = (KO & 8) != 0;
  constexpr int NTOUCH = PF > 0 ? 1 : 0;   // touch loads a wave leaves in flight behind each stage's loads
  constexpr int TMW = Shape<WAVES>::TMW, TNW = 4, RW = Shape<WAVES>::RW;
  constexpr int LDM = (RD == 2) ? 128 : 144;               // b128 reads: rows 1 KB apart are conflict-free; b64 reads want odd k rows 128 B further on
  constexpr int TILE_D = 16 * LDM, STAGE_D = 2 * TILE_D;
  constexpr int NRD = (RD == 2) ? (TMW + TNW) / 2 : (TMW + TNW);   // LDS reads per k-step and wave
  extern __shared__ double lds[];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / 2, wn = wave % 2;
  const GemmSeg& sg = g.seg[0];
  if (stamps && t == 0) {
    Stamp& s = stamps[blockIdx.x];
    s.t0 = (long long)__builtin_amdgcn_s_memtime(); s.r0 = (long long)__builtin_amdgcn_s_memrealtime();
    s.hwid = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);
    s.xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
  }
  bool ring_used = false;
  for (int u = 0; u < g.per; ++u) {
  const GemmTile tl = g.tiles[(int64_t)blockIdx.x * g.per + u];
  if (tl.kend <= tl.kbeg) continue;
  if (ring_used) __builtin_amdgcn_s_barrier();
  ring_used = true;
  const int64_t row0 = (int64_t)tl.bi * BM, col0 = (int64_t)tl.bj * BN;
  int lane = t & 63;
  asm volatile("" : "+v"(lane));
  const int kq = lane >> 4, cj = lane & 15;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
  const uint32_t a_addr0 = lds0 + 8u * (uint32_t)(kq * LDM + wm * RW + (RD == 2 ? 2 * cj : cj));
  const uint32_t b_addr0 = lds0 + 8u * (uint32_t)(TILE_D + kq * LDM + wn * 64 + (RD == 2 ? 2 * cj : cj));
  double* __restrict__ C = g.C;
  const int64_t ld = g.ldc;
  // element (tm, tn, r) of this lane: RD 2 interleaves rows / columns in pairs (sub-tile tm, A-row index i <-> row 32 (tm / 2) + 2 i + (tm & 1))
  auto c_row = [&](int tm, int r) -> int64_t {
    return (RD == 2) ? row0 + wm * RW + 32 * (tm / 2) + 2 * (4 * r + kq) + (tm & 1) : row0 + wm * RW + tm * 16 + 4 * r + kq;
  };
  auto c_col = [&](int tn) -> int64_t {
    return (RD == 2) ? col0 + wn * 64 + 32 * (tn / 2) + 2 * cj + (tn & 1) : col0 + wn * 64 + tn * 16 + cj;
  };

  mfma_d4 acc[TMW][TNW];
  if (tl.pad1 & 1) {   // continuation: the accumulators resume from the partial sums an earlier entry of this workgroup stored
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gi = c_row(tm, r);
        if constexpr (RD == 2) {
#pragma unroll
          for (int q = 0; q < TNW / 2; ++q) {
            const d2v v = *reinterpret_cast<const d2v*>(&C[gi * ld + c_col(2 * q)]);
            acc[tm][2 * q][r] = v[0]; acc[tm][2 * q + 1][r] = v[1];
          }
        } else {
#pragma unroll
          for (int tn = 0; tn < TNW; ++tn) acc[tm][tn][r] = C[gi * ld + c_col(tn)];
        }
      }
  } else {
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
      for (int b = 0; b < TNW; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
  }

  const int total = tl.kend - tl.kbeg;
  const uint32_t offA = glds_lane_offset<LAY_MNCONTIG, WAVES, false>(sg.lda, wave, lane), offB = glds_lane_offset<LAY_MNCONTIG, WAVES, false>(sg.ldb, wave, lane);
  const int64_t csA = glds_chunk_stride<LAY_MNCONTIG, WAVES>(sg.lda), csB = glds_chunk_stride<LAY_MNCONTIG, WAVES>(sg.ldb);
  const int64_t kfirst = (int64_t)tl.kbeg * BK;
  const char* baseA = (const char*)(sg.A + kfirst * sg.lda + row0);
  const char* baseB = (const char*)(sg.B + kfirst * sg.ldb + col0);
  const int64_t strideA = BK * 8 * sg.lda, strideB = BK * 8 * sg.ldb;
  uint32_t touch_v = 0;   // destination of the touch loads: stays reserved until the last of them has landed
  // PF > 0: behind the loads of stage `it` every wave touches (one 4-byte load per 128-byte line, 64 lines per wave) its share of the
  // operand slabs of stage it + PF - 1, so that they are in this XCD's L2 when their turn comes; the touches are the youngest loads in
  // flight and the stage waits leave them there (vmcnt(NTOUCH)).
  const uint32_t touch_off = (uint32_t)(((wave * 64 + (t & 63)) & 127) >> 3) * (uint32_t)(((wave * 64 + (t & 63)) & 128 ? sg.lda : sg.ldb) * 8) + (uint32_t)((wave * 64 + (t & 63)) & 7) * 128u;
  auto issue = [&](int it) {
    if (NOGLDS && it > 1) return;
    const int its = HOT ? (it & 1) : it;
    double* st = lds + (it & 1) * STAGE_D;
    stage_mn<WAVES, LDM>(st, baseA + its * strideA, offA, csA, wave);
    stage_mn<WAVES, LDM>(st + TILE_D, baseB + its * strideB, offB, csB, wave);
    if constexpr (PF > 0) {
      const int itp = (it + PF - 1 < total) ? it + PF - 1 : total - 1;
      // waves 0, 1 (lines 0..127): the B slab; waves 2, 3 (lines 128..255): the A slab; further waves repeat the B slab's lines
      const char* tb = glds_pin((wave & 2) ? baseA + itp * strideA : baseB + itp * strideB);
      // s_nop: an SGPR written by v_readfirstlane needs 5 wait states before a VMEM instruction reads it, and the compiler does not see a VMEM instruction here
      asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "+v"(touch_v) : "v"(touch_off), "s"(tb) : "memory");
    }
  };
  double af[2][TMW], bf[2][TNW];
  auto load = [&](auto ks_, auto set_, uint32_t aa, uint32_t ba) {
    constexpr int ks = decltype(ks_)::value, set = decltype(set_)::value;
    if constexpr (RD == 2) {
      sfor<0, TMW / 2>([&](auto p_) { constexpr int p = decltype(p_)::value;
        const d2v v = ds_rd128<(ks * 4 * LDM + p * 32) * 8>(aa); af[set][2 * p] = v[0]; af[set][2 * p + 1] = v[1]; });
      sfor<0, TNW / 2>([&](auto p_) { constexpr int p = decltype(p_)::value;
        const d2v v = ds_rd128<(ks * 4 * LDM + p * 32) * 8>(ba); bf[set][2 * p] = v[0]; bf[set][2 * p + 1] = v[1]; });
    } else {
      sfor<0, TMW>([&](auto p_) { constexpr int p = decltype(p_)::value; af[set][p] = ds_rd64<(ks * 4 * LDM + p * 16) * 8>(aa); });
      sfor<0, TNW>([&](auto p_) { constexpr int p = decltype(p_)::value; bf[set][p] = ds_rd64<(ks * 4 * LDM + p * 16) * 8>(ba); });
    }
  };
  auto mfmas = [&](auto set_) {
    constexpr int set = decltype(set_)::value;
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
      for (int tn = 0; tn < TNW; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[set][tm], bf[set][tn], acc[tm][tn], 0, 0, 0);
  };

  if constexpr (EARLYBAR) {
    issue(0);
    wait_vmcnt<NTOUCH>();
    __builtin_amdgcn_s_barrier();
    if (total > 1) issue(1);
    load(IC<0>{}, IC<0>{}, a_addr0, b_addr0);
    for (int it = 0; it < total; ++it) {
      const uint32_t so = (uint32_t)(it & 1) * (STAGE_D * 8), sn = (uint32_t)((it + 1) & 1) * (STAGE_D * 8);
      const uint32_t aa = a_addr0 + so, ba = b_addr0 + so;
      sfor<0, 3>([&](auto ks_) { constexpr int ks = decltype(ks_)::value;
        load(IC<ks + 1>{}, IC<(ks + 1) & 1>{}, aa, ba);
        wait_lgkm<NRD>();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(IC<ks & 1>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      // this wave has read all it needs of stage `it`; its own loads of stage it + 1 (issued one step ago) must have landed
      wait_lgkm<0>();
      wait_vmcnt<NTOUCH>();
      if (!NOBAR) __builtin_amdgcn_s_barrier();
      if (it + 2 < total) issue(it + 2);
      if (it + 1 < total) load(IC<0>{}, IC<0>{}, a_addr0 + sn, b_addr0 + sn);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(IC<1>{});
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    issue(0);
    for (int it = 0; it < total; ++it) {
      wait_vmcnt<NTOUCH>();
      if (!NOBAR) __builtin_amdgcn_s_barrier();
      if (it + 1 < total) issue(it + 1);
      const uint32_t so = (uint32_t)(it & 1) * (STAGE_D * 8);
      const uint32_t aa = a_addr0 + so, ba = b_addr0 + so;
      load(IC<0>{}, IC<0>{}, aa, ba);
      sfor<0, 4>([&](auto ks_) { constexpr int ks = decltype(ks_)::value;
        if constexpr (ks < 3) { load(IC<ks + 1>{}, IC<(ks + 1) & 1>{}, aa, ba); wait_lgkm<NRD>(); } else wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(IC<ks & 1>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  }

  if constexpr (PF > 0) { wait_vmcnt<0>(); asm volatile("" ::"v"(touch_v)); }   // the last touch has landed: its register is free again
  // epilogue: plain store (raw partial sums for an entry that a later one continues)
  if (NOEPI && acc[0][0][0] != 1.2345e300) continue;
  const double alpha = (tl.pad1 & 2) ? 1.0 : g.alpha;
#pragma unroll
  for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t gi = c_row(tm, r);
      if constexpr (RD == 2) {
#pragma unroll
        for (int q = 0; q < TNW / 2; ++q) {
          d2v v = {alpha * acc[tm][2 * q][r], alpha * acc[tm][2 * q + 1][r]};
          *reinterpret_cast<d2v*>(&C[gi * ld + c_col(2 * q)]) = v;
        }
      } else {
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) C[gi * ld + c_col(tn)] = alpha * acc[tm][tn][r];
      }
    }
  }
  if (stamps && t == 0) {
    Stamp& s = stamps[blockIdx.x];
    s.t1 = (long long)__builtin_amdgcn_s_memtime(); s.r1 = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

static GemmTile mk(int bi, int bj, int k0, int k1, int flags = 0) {
  GemmTile t; t.bi = bi; t.bj = bj; t.kbeg = k0; t.kend = k1; t.slice = 0; t.kdir = 1; t.pad1 = flags; t.pad2 = 0; return t;
}
// one tile per workgroup, launch position p -> XCD p % 8 (the product's order, zigp_host.h tiles_full_xcd)
static std::vector<GemmTile> tiles_full_xcd_host(int nbm, int nbn, int nk) {
  std::vector<GemmTile> q[8], v;
  for (int bj = 0; bj < nbn; ++bj)
    for (int bi = 0; bi < nbm; ++bi) q[bj % 8].push_back(mk(bi, bj, 0, nk));
  size_t longest = 0;
  for (int x = 0; x < 8; ++x) longest = std::max(longest, q[x].size());
  for (size_t e = 0; e < longest; ++e)
    for (int x = 0; x < 8; ++x) v.push_back(e < q[x].size() ? q[x][e] : mk(0, 0, 0, 0));
  return v;
}
// persistent: 512 workgroups, workgroup w = 8 e + x (XCD x, slot e of 64) runs the queue entries e, e + 64, e + 128, ... of its XCD.
// stagger: 0 none; 1 slots e >= 32 are shifted by half a tile (first half of their first tile, the other tiles, then its second half);
//          2 odd slots shifted; 3 slots with (e / 8) odd shifted; 4 slots with (e / 16) odd shifted
static std::vector<GemmTile> tiles_persistent(int nbm, int nbn, int nk, int stagger, int& per) {
  std::vector<GemmTile> q[8];
  for (int bj = 0; bj < nbn; ++bj)
    for (int bi = 0; bi < nbm; ++bi) q[bj % 8].push_back(mk(bi, bj, 0, nk));
  const int slots = 64;
  size_t longest = 0;
  for (int x = 0; x < 8; ++x) longest = std::max(longest, q[x].size());
  const int nt = (int)((longest + slots - 1) / slots);
  per = nt + 1;
  std::vector<GemmTile> v((size_t)512 * per, mk(0, 0, 0, 0));
  for (int e = 0; e < slots; ++e)
    for (int x = 0; x < 8; ++x) {
      const int w = 8 * e + x;
      std::vector<GemmTile> mine;
      for (int i = 0; i < nt; ++i) if ((size_t)(e + i * slots) < q[x].size()) mine.push_back(q[x][e + i * slots]);
      bool shifted = false;
      if (stagger == 1) shifted = e >= 32;
      if (stagger == 2) shifted = (e & 1) != 0;
      if (stagger == 3) shifted = ((e / 8) & 1) != 0;
      if (stagger == 4) shifted = ((e / 16) & 1) != 0;
      std::vector<GemmTile> seq;
      if (shifted && !mine.empty()) {
        GemmTile h0 = mine[0], h1 = mine[0];
        h0.kend = nk / 2; h0.pad1 = 2;          // raw partial sums
        h1.kbeg = nk / 2; h1.pad1 = 1;          // continues from them
        seq.push_back(h0);
        for (size_t i = 1; i < mine.size(); ++i) seq.push_back(mine[i]);
        seq.push_back(h1);
      } else seq = mine;
      for (size_t i = 0; i < seq.size(); ++i) v[(size_t)w * per + i] = seq[i];
    }
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

static double max_diff(const std::vector<double>& a, const std::vector<double>& b) {
  double m = 0; for (size_t i = 0; i < a.size(); ++i) m = std::max(m, std::fabs(a[i] - b[i])); return m;
}

int main(int argc, char** argv) {
  const int M = 1024;
  const int64_t N = (argc > 1) ? atoll(argv[1]) : 32768;
  const int K = (argc > 2) ? atoi(argv[2]) : 1024;
  const int reps = (argc > 3) ? atoi(argv[3]) : 50;
  const int data = (argc > 4) ? atoi(argv[4]) : 0;
  const int rounds = (argc > 5) ? atoi(argv[5]) : 2;
  printf("gemm_lab: C[%d x %lld] = A^T-image[%d x %d] B[%d x %lld], data %s, %d launches per timing, %d rounds\n", M, (long long)N, K, M, K, (long long)N,
         data == 0 ? "uniform random" : data == 1 ? "zeros" : "smooth", reps, rounds);
  std::vector<double> hA((size_t)K * M), hB((size_t)K * N);
  srand(1);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = data == 0 ? (rand() / (double)RAND_MAX - 0.5) : data == 1 ? 0.0 : 1.0 + 1e-3 * (double)(i % 97);
  for (size_t i = 0; i < hB.size(); ++i) hB[i] = data == 0 ? (rand() / (double)RAND_MAX - 0.5) : data == 1 ? 0.0 : 0.5 + 1e-3 * (double)(i % 89);
  double *dA, *dB, *dC; GemmTile *dT, *dTp[5]; Stamp* dS;
  CK(hipMalloc(&dA, sizeof(double) * hA.size())); CK(hipMalloc(&dB, sizeof(double) * hB.size())); CK(hipMalloc(&dC, sizeof(double) * M * N));
  CK(hipMemcpy(dA, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), sizeof(double) * hB.size(), hipMemcpyHostToDevice));
  const int nbm = M / BM, nbn = (int)(N / BN), nk = K / BK;
  std::vector<GemmTile> tiles = tiles_full_xcd_host(nbm, nbn, nk);
  CK(hipMalloc(&dT, sizeof(GemmTile) * tiles.size()));
  CK(hipMemcpy(dT, tiles.data(), sizeof(GemmTile) * tiles.size(), hipMemcpyHostToDevice));
  int per_p[5];
  for (int s = 0; s < 5; ++s) {
    std::vector<GemmTile> tp = tiles_persistent(nbm, nbn, nk, s, per_p[s]);
    CK(hipMalloc(&dTp[s], sizeof(GemmTile) * tp.size()));
    CK(hipMemcpy(dTp[s], tp.data(), sizeof(GemmTile) * tp.size(), hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&dS, sizeof(Stamp) * tiles.size()));
  CK(hipMemset(dS, 0, sizeof(Stamp) * tiles.size()));
  GemmArgs g; g.seg[0].A = dA; g.seg[0].B = dB; g.seg[0].lda = M; g.seg[0].ldb = N; g.tiles = dT; g.per = 1; g.C = dC; g.ldc = N; g.slice_stride = 0; g.alpha = 1.0; g.kscale = nullptr;
  const double flops = 2.0 * M * (double)N * K;
  std::vector<double> ref((size_t)M * N), out((size_t)M * N);

  struct Var { std::string name; std::function<void()> launch; int grid; bool check; std::vector<double> ms; double diff = -1; double mhz = 0; };
  std::vector<Var> vars;
  {
    auto kern = gemm_f64_kernel<LAY_MNCONTIG, LAY_MNCONTIG, 2, false, TRI_NONE, 8, EpiStore>;
    const size_t shm = sizeof(double) * 2 * STAGE_DOUBLES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    const int grid = (int)tiles.size();
    vars.push_back({"product <1,1,2,false,0,8,EpiStore>", [=] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shm, 0, g, EpiStore()); }, grid, false, {}});
  }
  {
    auto kern = gemm_f64_kernel<LAY_MNCONTIG, LAY_MNCONTIG, 2, false, TRI_NONE, 4, EpiStore>;
    const size_t shm = sizeof(double) * 2 * STAGE_DOUBLES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    const int grid = (int)tiles.size();
    vars.push_back({"product, 4-wave shape", [=] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shm, 0, g, EpiStore()); }, grid, true, {}});
  }
#define ADD(WAVES, RD, EB, KO, SCHED, PF)                                                                                                 \
  {                                                                                                                                   \
    auto kern = k_v2<WAVES, RD, EB, KO, PF>;                                                                                              \
    const size_t shm = sizeof(double) * 2 * 2 * 16 * ((RD) == 2 ? 128 : 144);                                                         \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));               \
    GemmArgs g2 = g;                                                                                                                  \
    int grid = (int)tiles.size();                                                                                                     \
    if ((SCHED) >= 0) { g2.tiles = dTp[SCHED]; g2.per = per_p[SCHED]; grid = 512; }                                                   \
    char nm[160];                                                                                                                     \
    snprintf(nm, sizeof(nm), "v2 w%d %s eb%d ko%d pf%d %s", WAVES, (RD) == 2 ? "b128" : "b64", EB, KO, PF,                                     \
             (SCHED) < 0 ? "tile/wg" : (SCHED) == 0 ? "persistent" : (SCHED) == 1 ? "pers stagger e>=32" : (SCHED) == 2 ? "pers stagger odd e" : (SCHED) == 3 ? "pers stagger e/8 odd" : "pers stagger e/16 odd"); \
    Stamp* st = dS;                                                                                                                   \
    vars.push_back({nm, [=] { hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), shm, 0, g2, st); }, grid, (KO) == 0, {}});       \
  }
  ADD(4, 0, 0, 0, -1, 0)
  ADD(8, 0, 0, 0, -1, 0)
  ADD(8, 0, 1, 0, -1, 0)
  ADD(4, 0, 0, 8, -1, 0)
  ADD(8, 0, 0, 8, -1, 0)
  ADD(8, 0, 0, 2, -1, 0)
  ADD(8, 0, 0, 1, -1, 0)
  ADD(8, 0, 0, 7, -1, 0)
  ADD(4, 0, 0, 0, -1, 2)
  ADD(4, 0, 0, 0, -1, 3)
  ADD(8, 0, 0, 0, -1, 2)
  ADD(8, 0, 0, 0, -1, 3)
  ADD(8, 0, 0, 0, -1, 4)
  ADD(8, 0, 0, 0, 0, 0)
  ADD(8, 0, 0, 0, 0, 3)

  Timer tm;
  // correctness first (one launch each), then interleaved timing rounds
  for (size_t i = 0; i < vars.size(); ++i) {
    CK(hipMemset(dC, 0, sizeof(double) * M * N));
    vars[i].launch();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    if (i == 0) {
      CK(hipMemcpy(ref.data(), dC, sizeof(double) * M * N, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int s = 0; s < 16; ++s) {
        const int ii = (s * 67 + 3) % M; const int64_t j = ((int64_t)s * 2039 + 11) % N;
        double acc = 0; for (int k = 0; k < K; ++k) acc += hA[(size_t)k * M + ii] * hB[(size_t)k * N + j];
        worst = std::max(worst, std::fabs(acc - ref[(size_t)ii * N + j]));
      }
      printf("product kernel vs host dot products (16 entries): max abs diff %.2e\n", worst);
    } else if (vars[i].check) {
      CK(hipMemcpy(out.data(), dC, sizeof(double) * M * N, hipMemcpyDeviceToHost));
      vars[i].diff = max_diff(ref, out);
    }
  }
  // census: which workgroups share a CU (persistent grid of 512)
  {
    std::vector<Stamp> hs(512);
    CK(hipMemset(dS, 0, sizeof(Stamp) * 512));
    for (auto& v : vars) if (v.name.find("persistent") != std::string::npos && v.name.find("w4") != std::string::npos) { v.launch(); break; }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hs.data(), dS, sizeof(Stamp) * 512, hipMemcpyDeviceToHost));
    std::map<unsigned, std::vector<int>> cu;
    int xcc_ok = 0;
    for (int w = 0; w < 512; ++w) {
      const unsigned cuid = (hs[w].hwid >> 8) & 0xf, sh = (hs[w].hwid >> 12) & 1, se = (hs[w].hwid >> 13) & 7, xcc = hs[w].xcc & 0xf;
      cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back(w);
      if ((int)xcc == (int)(hs[0].xcc & 0xf) + 0 && w % 8 == 0) ++xcc_ok;
    }
    printf("census of a 512-workgroup launch: %zu distinct (xcc, se, sh, cu) ids; workgroups w with w %% 8 == 0 on workgroup 0's XCC: %d of 64\n", cu.size(), xcc_ok);
    int shown = 0;
    for (auto& kv : cu) {
      if (shown++ >= 12) break;
      printf("  cu %05x:", kv.first);
      for (int w : kv.second) printf(" w%d(e=%d,x=%d)", w, w / 8, w % 8);
      printf("\n");
    }
    // co-residency statistics: slot difference of the workgroups that share a CU
    std::map<int, int> hist;
    for (auto& kv : cu) if (kv.second.size() == 2) hist[std::abs(kv.second[0] / 8 - kv.second[1] / 8)]++;
    printf("  |e1 - e2| of CU partners:");
    for (auto& h : hist) printf(" %d:%d", h.first, h.second);
    printf("\n");
  }
  for (auto& v : vars) { for (int i = 0; i < 3; ++i) v.launch(); }
  CK(hipDeviceSynchronize());
  std::vector<Stamp> hs(tiles.size());
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars) {
      v.ms.push_back(tm.run(v.launch, reps));
      if (v.name[0] == 'v') {   // in-kernel clock of the last launch
        CK(hipMemcpy(hs.data(), dS, sizeof(Stamp) * v.grid, hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        for (int w = 0; w < v.grid; ++w) if (hs[w].r1 > hs[w].r0) mhz.push_back((double)(hs[w].t1 - hs[w].t0) / (double)(hs[w].r1 - hs[w].r0) * 100.0);
        std::sort(mhz.begin(), mhz.end());
        if (!mhz.empty()) v.mhz = mhz[mhz.size() / 2];
      }
    }
  for (auto& v : vars) {
    std::vector<double> s = v.ms; std::sort(s.begin(), s.end());
    printf("%-42s best %7.3f ms %6.2f TF | median %6.2f TF | clock %4.0f MHz", v.name.c_str(), s[0], flops / s[0] * 1e-9, flops / s[s.size() / 2] * 1e-9, v.mhz);
    if (v.diff >= 0) printf(" | max|diff| %.1e", v.diff);
    printf("\n");
  }
  return 0;
}
