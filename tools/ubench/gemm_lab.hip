// GEMM laboratory (round 5): candidate inner loops and tile schedules for the fp64 GEMM core, measured on the shape of the chunk loop's
// dominant product (J' = Q A2: 1024 x N x K, both operands m/n-contiguous, XCD-aware tile order) next to the product kernel of zigp_gemm.h.
// The candidates ("v2") software-pipeline the LDS fragment reads by hand: the reads are inline assembly with explicit
// s_waitcnt lgkmcnt(N), the MFMAs are builtins (the compiler keeps their hazards), sched_barrier fences pin the order.
// First generation ("v2", commit 2620d88; results in profiles/r05a-c_gemm_lab_*.log): ds_read_b128 fragment pairs, the barrier in front of
// the last k-step, staggered tile boundaries of the two workgroups of a CU, touch-prefetch of the operand slabs, hot staging addresses --
// none of them pays; what the knock-outs show is barrier 2.5 %, staging loads 2.7 %, epilogue stores 1.1-1.5 % of the full product.
// This generation ("v3", below): the wave-tile SHAPE, and triangular skipping that is uniform over the workgroup.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../zero-inflated-gp_amd/csrc gemm_lab.hip -o gemm_lab
// Run:   gemm_lab [N=32768] [reps=50] [rounds=3]
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

template <int OFF> __device__ __forceinline__ d2v ds_rd128(uint32_t a) {
  d2v v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v;
}

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

// the MFMAs of one sub-tile row behind a scalar branch on bit TM of `mask` (uniform), as ONE assembly block: no control flow the compiler can see
template <int TM> __device__ __forceinline__ void mfma_masked(mfma_d4& c, double a, double b, uint32_t mask) {
  asm volatile("s_bitcmp1_b32 %3, %4\n\ts_cbranch_scc0 .Lskip%=\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n.Lskip%=:"
               : "+v"(c) : "v"(a), "v"(b), "s"(mask), "n"(TM) : "scc");
}
template <int TM> __device__ __forceinline__ void mfma_masked2(mfma_d4& c0, mfma_d4& c1, double a, double b0, double b1, uint32_t mask) {
  asm volatile("s_bitcmp1_b32 %5, %6\n\ts_cbranch_scc0 .Lskip%=\n\tv_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %4, %1\n.Lskip%=:"
               : "+v"(c0), "+v"(c1) : "v"(a), "v"(b0), "v"(b1), "s"(mask), "n"(TM) : "scc");
}

struct Stamp { long long t0, r0, t1, r1; unsigned hwid, xcc, pad0, pad1; };

// v3<WAVES, WMW, TRI, KO, XPF>: WMW waves along M (WAVES / WMW along N); wave tile (128 / WMW) rows x (128 / WNW) columns.
//   WMW = 1 is the COLUMN-STRIP shape: every wave owns all 128 rows of its 128 / WAVES columns.  For a triangular A operand that makes
//   the skipping uniform over the workgroup: inside the diagonal 128-block, staged step st needs the 16-row sub-tiles t >= st (lower) or
//   t <= st (upper) -- the same set in every wave.
//   TRI (WMW = 1 only): 0 none, 1 A lower (A(i,k) = 0 for k > i), 2 A upper.   KO (timing only): 1 no barrier, 2 no staging loads, 4 no stores
//   XPF: cross-tile prefetch -- the FIRST stage of a workgroup's next list entry is requested at the top of the current entry's LAST step
//   (its ring slot is free by then), i.e. in front of the epilogue's stores in vmcnt order: the next tile's first step waits with
//   vmcnt(<stores per wave>) for the loads alone, and neither the stores' round trip nor the first loads' latency stands between two tiles.
template <int WAVES, int WMW, int TRI, int KO, int XPF>
__global__ void __launch_bounds__(64 * WAVES, 2 * WAVES / 4)
k_v3(GemmArgs g, Stamp* stamps) {
  constexpr bool NOBAR = (KO & 1) != 0, NOGLDS = (KO & 2) != 0, NOEPI = (KO & 4) != 0;
  constexpr int WNW = WAVES / WMW, TMW = 8 / WMW, TNW = 8 / WNW, RW = 128 / WMW, CW = 128 / WNW;
  static_assert(TRI == 0 || WMW == 1, "triangular skipping: column-strip shape only");
  constexpr int NSTORES = TMW * TNW * 4;     // epilogue stores per wave
  static_assert(!XPF || NSTORES <= 63, "vmcnt is a 6-bit counter");
  constexpr int LDM = 144;
  constexpr int TILE_D = 16 * LDM, STAGE_D = 2 * TILE_D;
  extern __shared__ double lds[];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const GemmSeg& sg = g.seg[0];
  if (stamps && t == 0) {
    Stamp& s = stamps[blockIdx.x];
    s.t0 = (long long)__builtin_amdgcn_s_memtime(); s.r0 = (long long)__builtin_amdgcn_s_memrealtime();
    s.hwid = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);
    s.xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
  }
  int lane = t & 63;
  asm volatile("" : "+v"(lane));
  const int kq = lane >> 4, cj = lane & 15;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
  const uint32_t a_addr0 = lds0 + 8u * (uint32_t)(kq * LDM + wm * RW + cj);
  const uint32_t b_addr0 = lds0 + 8u * (uint32_t)(TILE_D + kq * LDM + wn * CW + cj);
  double* __restrict__ C = g.C;
  const int64_t ld = g.ldc;
  const uint32_t offA = glds_lane_offset<LAY_MNCONTIG, WAVES>(sg.lda, wave, lane), offB = glds_lane_offset<LAY_MNCONTIG, WAVES>(sg.ldb, wave, lane);
  const int64_t csA = glds_chunk_stride<LAY_MNCONTIG, WAVES>(sg.lda), csB = glds_chunk_stride<LAY_MNCONTIG, WAVES>(sg.ldb);

  // scalar staging state of the tile whose stages are being requested (pinned where defined: see glds_pin)
  const char* nextA = nullptr; const char* nextB = nullptr; int64_t strideA = 0, strideB = 0;
  int issued = 0, consumed = 0;      // ring positions, carried across tiles
  auto tile_begin = [&](const GemmTile& tt) {
    const int kb0 = (tt.kdir >= 0) ? tt.kbeg : tt.kend - 1;
    const int64_t kfirst = (int64_t)kb0 * BK, kd = (tt.kdir >= 0) ? 1 : -1;
    nextA = glds_pin((const char*)(sg.A + kfirst * sg.lda + (int64_t)tt.bi * BM));
    nextB = glds_pin((const char*)(sg.B + kfirst * sg.ldb + (int64_t)tt.bj * BN));
    strideA = kd * BK * 8 * sg.lda; strideB = kd * BK * 8 * sg.ldb;
  };
  auto issue = [&]() {
    double* st = lds + (issued & 1) * STAGE_D;
    stage_mn<WAVES, LDM>(st, nextA, offA, csA, wave);
    stage_mn<WAVES, LDM>(st + TILE_D, nextB, offB, csB, wave);
    nextA = glds_pin(nextA + strideA); nextB = glds_pin(nextB + strideB); ++issued;
  };

  mfma_d4 acc[TMW][TNW];
  // one staged step (16 k): fragment reads one k-step ahead of the MFMAs (all fragments are read, masked or not: a read is cheap and the
  // lgkmcnt accounting stays static).  MASKED: the MFMAs of sub-tile row tm are issued only when bit tm of `mask` (an SGPR) is set.  The
  // scalar branch lives INSIDE the assembly block of the row's MFMAs: branches the compiler can see (one per row and k-step) are
  // tail-duplicated and threaded into a control-flow graph that spills 180 B/lane, whichever way they are written.
  auto body = [&](auto masked_, uint32_t mask, uint32_t aa, uint32_t ba) {
    constexpr bool MASKED = decltype(masked_)::value != 0;
    constexpr int NRD = TMW + TNW;
    double af[2][TMW], bf[2][TNW];
    auto load = [&](auto ks_, auto set_) {
      constexpr int ks = decltype(ks_)::value, set = decltype(set_)::value;
      sfor<0, TMW>([&](auto p_) { constexpr int p = decltype(p_)::value; af[set][p] = ds_rd64<(ks * 4 * LDM + p * 16) * 8>(aa); });
      sfor<0, TNW>([&](auto p_) { constexpr int p = decltype(p_)::value; bf[set][p] = ds_rd64<(ks * 4 * LDM + p * 16) * 8>(ba); });
    };
    load(IC<0>{}, IC<0>{});
    sfor<0, 4>([&](auto ks_) { constexpr int ks = decltype(ks_)::value;
      if constexpr (ks < 3) { load(IC<ks + 1>{}, IC<(ks + 1) & 1>{}); wait_lgkm<(NRD < 15 ? NRD : 15)>(); } else wait_lgkm<0>();
      __builtin_amdgcn_sched_barrier(0);
      sfor<0, TMW>([&](auto tm_) { constexpr int tm = decltype(tm_)::value;
        if constexpr (!MASKED) {
#pragma unroll
          for (int tn = 0; tn < TNW; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks & 1][tm], bf[ks & 1][tn], acc[tm][tn], 0, 0, 0);
        } else if constexpr (TNW == 1) {
          mfma_masked<tm>(acc[tm][0], af[ks & 1][tm], bf[ks & 1][0], mask);
        } else {
          static_assert(TNW <= 2, "masked MFMA groups: one or two column sub-tiles per wave");
          mfma_masked2<tm>(acc[tm][0], acc[tm][1], af[ks & 1][tm], bf[ks & 1][0], bf[ks & 1][1], mask);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // list entries come in through the SCALAR cache (s_load: counted on lgkmcnt): a vector load here would make every tile begin with
  // s_waitcnt vmcnt(0), i.e. behind the previous tile's stores
  const GemmTile* my = g.tiles + (int64_t)blockIdx.x * g.per;
  auto load_entry = [&](int idx) -> GemmTile {
    typedef int i8v __attribute__((ext_vector_type(8)));
    static_assert(sizeof(GemmTile) == 32, "one s_load_dwordx8 per list entry");
    i8v d;
    const char* p = glds_pin((const char*)(my + idx));
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(d) : "s"(p) : "memory");
    GemmTile e; e.bi = d[0]; e.bj = d[1]; e.kbeg = d[2]; e.kend = d[3]; e.slice = d[4]; e.kdir = d[5]; e.pad1 = d[6]; e.pad2 = d[7];
    return e;
  };
  bool prefetched = false;        // stage 0 of the entry about to run has been requested already (XPF)
  bool ring_used = false;
  GemmTile cur = load_entry(0);
  for (int u = 0; u < g.per; ++u) {
  const GemmTile tl = cur;
  GemmTile nx; nx.bi = nx.bj = nx.kbeg = nx.kend = nx.slice = nx.pad1 = nx.pad2 = 0; nx.kdir = 1;
  if (u + 1 < g.per) nx = load_entry(u + 1);
  cur = nx;
  if (tl.kend <= tl.kbeg) continue;
  const bool have_next = XPF && nx.kend > nx.kbeg;      // (a persistent list has its empty entries at the end)
  const int64_t row0 = (int64_t)tl.bi * BM, col0 = (int64_t)tl.bj * BN;
  const int total = tl.kend - tl.kbeg;
  const int kb0 = (tl.kdir >= 0) ? tl.kbeg : tl.kend - 1, kdi = (tl.kdir >= 0) ? 1 : -1;
#pragma unroll
  for (int a = 0; a < TMW; ++a)
#pragma unroll
    for (int b = 0; b < TNW; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
  if (!prefetched) {
    if (ring_used) __builtin_amdgcn_s_barrier();     // slower waves may still read the previous tile's last stage
    tile_begin(tl);
    issue();
  }
  ring_used = true;
  for (int it = 0; it < total; ++it) {
    if (XPF && it == 0 && prefetched) wait_vmcnt<(XPF ? NSTORES : 0)>(); else wait_vmcnt<0>();
    if (!NOBAR) __builtin_amdgcn_s_barrier();
    if (it + 1 < total) { if (!(NOGLDS && it > 0)) issue(); }
    else if (have_next) { tile_begin(nx); issue(); }      // the next entry's first stage, in front of this entry's stores
    const uint32_t so = (uint32_t)(consumed & 1) * (STAGE_D * 8);
    ++consumed;
    if (NOGLDS && it + 1 < total && it > 0) ++issued;
    const uint32_t aa = a_addr0 + so, ba = b_addr0 + so;
    if constexpr (TRI == 0) body(IC<0>{}, 0xffu, aa, ba);
    else {
      const int st = kb0 + kdi * it - tl.bi * (BM / BK);   // staged step relative to the diagonal block (one step = 16 k = one sub-tile row)
      const int sc = st < 0 ? 0 : (st > 7 ? 7 : st);
      const uint32_t mask = (TRI == 1) ? ((0xffu << sc) & 0xffu) : ((2u << sc) - 1u);   // lower: sub-tile t takes part while st <= t; upper: once st >= t
      body(IC<1>{}, (uint32_t)__builtin_amdgcn_readfirstlane((int)mask), aa, ba);
    }
  }
  prefetched = have_next;

  if constexpr (TRI != 0) {   // the MFMAs were issued from assembly blocks: the compiler does not know that the accumulators come out of the matrix pipe
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  if (NOEPI && acc[0][0][0] != 1.2345e300) continue;
#pragma unroll
  for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t gi = row0 + wm * RW + tm * 16 + 4 * r + kq;
#pragma unroll
      for (int tn = 0; tn < TNW; ++tn) C[gi * ld + col0 + wn * CW + tn * 16 + cj] = g.alpha * acc[tm][tn][r];
    }
  }
  if (stamps && t == 0) {
    Stamp& s = stamps[blockIdx.x];
    s.t1 = (long long)__builtin_amdgcn_s_memtime(); s.r1 = (long long)__builtin_amdgcn_s_memrealtime();
  }
}


// ---- "dva": the A operand DIRECT to registers (the verdict's suggestion, what the vendor's gfx950 DGEMM kernels do): only the B tile is staged
// through LDS; each wave fetches its own A fragments -- lane (kq, cj) of sub-tile tm and k-step ks holds A(row 32 wm + 16 tm + cj,
// k 4 ks + kq) -- with global_load_dwordx2 (16 lanes = 128 contiguous bytes of the m-contiguous image), one staged step ahead, into a
// second register set.  Per staged step and wave: 8 such loads instead of this wave's share of the A tile's LDS staging (2 x 1 KB), and 4
// instead of 6 ds_read_b64 per k-step; the two waves of a wave row fetch the same A data (L1 / L2 hits).  The main loop is unrolled over
// the two register sets / ring slots.  Result (profiles/r05y_gemm_lab_direct_to_vgpr.log): bit-identical to the product kernel, 68.2 vs 71.5 TFLOP/s (72.7 vs 73.5 without
// the barrier): within 128 VGPRs there is room for ONE step of A prefetch only, every wave then waits for its own ten loads at the top of each
// step and the barrier collects the slowest of eight -- the LDS ring, where any wave's staged chunk serves all, hides the same latency better.
template <int OFF> __device__ __forceinline__ double gld64(const char* base, uint32_t off) {
  double v; asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(v) : "v"(off), "s"(base), "n"(OFF) : "memory"); return v;
}
template <int OFF> __device__ __forceinline__ double gld64_first(const char* base, uint32_t off) {    // behind a v_readfirstlane of `base`: 5 wait states
  double v; asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(v) : "v"(off), "s"(base), "n"(OFF) : "memory"); return v;
}
template <int KO>
__global__ void __launch_bounds__(512, 4)
k_dva(GemmArgs g, Stamp* stamps) {
  constexpr bool NOBAR = (KO & 1) != 0, NOEPI = (KO & 4) != 0;
  constexpr int LDM = 144, TILE_D = 16 * LDM;
  extern __shared__ double lds[];      // two ring slots of the B tile
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const GemmSeg& sg = g.seg[0];
  if (stamps && t == 0) { Stamp& s = stamps[blockIdx.x]; s.t0 = (long long)__builtin_amdgcn_s_memtime(); s.r0 = (long long)__builtin_amdgcn_s_memrealtime(); }
  int lane = t & 63;
  asm volatile("" : "+v"(lane));
  const int kq = lane >> 4, cj = lane & 15;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
  const uint32_t b_addr0 = lds0 + 8u * (uint32_t)(kq * LDM + wn * 64 + cj);
  double* __restrict__ C = g.C;
  const int64_t ld = g.ldc;
  const uint32_t offB = glds_lane_offset<LAY_MNCONTIG, 8>(sg.ldb, wave, lane);
  const int64_t csB = glds_chunk_stride<LAY_MNCONTIG, 8>(sg.ldb);
  uint32_t aoff = (uint32_t)((kq * sg.lda + wm * 32 + cj) * 8);     // k-step ks adds 4 ks lda doubles to the SCALAR base
  asm volatile("" : "+v"(aoff));
  const int64_t kstepA = 4 * 8 * sg.lda;
  const char* nextA = nullptr; const char* nextB = nullptr; int64_t strideA = 0, strideB = 0;
  double ag[2][4][2];
  auto issueB = [&](int slot) {
    stage_mn<8, LDM>(lds + slot * TILE_D, nextB, offB, csB, wave);
    nextB = glds_pin(nextB + strideB);
  };
  auto issueA = [&](auto set_) {
    constexpr int set = decltype(set_)::value;
    ag[set][0][0] = gld64_first<0>(nextA, aoff); ag[set][0][1] = gld64<128>(nextA, aoff);
    sfor<1, 4>([&](auto ks_) { constexpr int ks = decltype(ks_)::value; const char* b = nextA + ks * kstepA; ag[set][ks][0] = gld64<0>(b, aoff); ag[set][ks][1] = gld64<128>(b, aoff); });
    nextA = glds_pin(nextA + strideA);
  };
  mfma_d4 acc[2][4];
  const GemmTile* my = g.tiles + (int64_t)blockIdx.x * g.per;
  bool ring_used = false;
  for (int u = 0; u < g.per; ++u) {
  const GemmTile tl = my[u];
  if (tl.kend <= tl.kbeg) continue;
  const int64_t row0 = (int64_t)tl.bi * BM, col0 = (int64_t)tl.bj * BN;
  const int total = tl.kend - tl.kbeg;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
  if (ring_used) __builtin_amdgcn_s_barrier();
  ring_used = true;
  {
    const int64_t kfirst = (int64_t)tl.kbeg * BK;
    nextA = glds_pin((const char*)(sg.A + kfirst * sg.lda + row0));
    nextB = glds_pin((const char*)(sg.B + kfirst * sg.ldb + col0));
    strideA = (int64_t)BK * 8 * sg.lda; strideB = (int64_t)BK * 8 * sg.ldb;
  }
  issueB(0); issueA(IC<0>{});
  auto step = [&](int it, auto par_) {
    constexpr int P = decltype(par_)::value;
    wait_vmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    if (!NOBAR) __builtin_amdgcn_s_barrier();
    if (it + 1 < total) { issueB(P ^ 1); issueA(IC<P ^ 1>{}); }
    const uint32_t ba = b_addr0 + (uint32_t)(P * TILE_D * 8);
    double bf[2][4];
    auto loadB = [&](auto ks_, auto set_) {
      constexpr int ks = decltype(ks_)::value, set = decltype(set_)::value;
      sfor<0, 4>([&](auto p_) { constexpr int p = decltype(p_)::value; bf[set][p] = ds_rd64<(ks * 4 * LDM + p * 16) * 8>(ba); });
    };
    loadB(IC<0>{}, IC<0>{});
    sfor<0, 4>([&](auto ks_) { constexpr int ks = decltype(ks_)::value;
      if constexpr (ks < 3) { loadB(IC<ks + 1>{}, IC<(ks + 1) & 1>{}); wait_lgkm<4>(); } else wait_lgkm<0>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[P][ks][tm], bf[ks & 1][tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  int it = 0;
  for (; it + 1 < total; it += 2) { step(it, IC<0>{}); step(it + 1, IC<1>{}); }
  if (it < total) step(it, IC<0>{});
  if (NOEPI && acc[0][0][0] != 1.2345e300) continue;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t gi = row0 + wm * 32 + tm * 16 + 4 * r + kq;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) C[gi * ld + col0 + wn * 64 + tn * 16 + cj] = g.alpha * acc[tm][tn][r];
    }
  }
  if (stamps && t == 0) { Stamp& s = stamps[blockIdx.x]; s.t1 = (long long)__builtin_amdgcn_s_memtime(); s.r1 = (long long)__builtin_amdgcn_s_memrealtime(); }
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
// the product's paired order of the triangular products (zigp_host.h tiles_trmm, paired = true): two entries per workgroup
static std::vector<GemmTile> tiles_trmm_paired_host(bool lower, int nbm, int nbn) {
  const int kb = BM / BK, U = (nbm + 1) / 2;
  auto tile = [&](int bi, int bj, int dir) { GemmTile t = lower ? mk(bi, bj, 0, (bi + 1) * kb) : mk(bi, bj, bi * kb, nbm * kb); t.kdir = dir; return t; };
  std::vector<GemmTile> q[8], v;
  for (int bj = 0; bj < nbn; ++bj)
    for (int u = 0; u < U; ++u) {
      const int lo = u, hi = nbm - 1 - u;
      std::vector<GemmTile>& dst = q[bj % 8];
      if (lo == hi) { dst.push_back(tile(lo, bj, lower ? -1 : 1)); dst.push_back(mk(0, 0, 0, 0)); continue; }
      if (lower) { dst.push_back(tile(lo, bj, 1)); dst.push_back(tile(hi, bj, -1)); }
      else { dst.push_back(tile(hi, bj, -1)); dst.push_back(tile(lo, bj, 1)); }
    }
  size_t longest = 0;
  for (int x = 0; x < 8; ++x) longest = std::max(longest, q[x].size());
  for (size_t e0 = 0; e0 < longest; e0 += 2)
    for (int x = 0; x < 8; ++x)
      for (int e = 0; e < 2; ++e) v.push_back(e0 + e < q[x].size() ? q[x][e0 + e] : mk(0, 0, 0, 0));
  return v;
}

// persistent form of a list: 512 workgroups, workgroup w runs the units w, w + 512, w + 1024, ... of the one-unit-per-workgroup list
// (the order the hardware would start them in); `per` entries per unit -> per * ceil(units / 512) entries per workgroup
static std::vector<GemmTile> persistent(const std::vector<GemmTile>& v, int per, int& per_out) {
  const int units = (int)v.size() / per, rounds = (units + 511) / 512;
  per_out = per * rounds;
  std::vector<GemmTile> o((size_t)512 * per_out, mk(0, 0, 0, 0));
  for (int w = 0; w < 512; ++w)
    for (int r = 0; r < rounds; ++r) {
      const int unit = w + 512 * r;
      if (unit >= units) continue;
      for (int e = 0; e < per; ++e) o[(size_t)w * per_out + r * per + e] = v[(size_t)unit * per + e];
    }
  return o;
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
  const int M = 1024, K = 1024;
  const int64_t N = (argc > 1) ? atoll(argv[1]) : 32768;
  const int reps = (argc > 2) ? atoi(argv[2]) : 50;
  const int rounds = (argc > 3) ? atoi(argv[3]) : 3;
  printf("gemm_lab: C[%d x %lld] = A(m-contiguous image)[%d x %d] B[%d x %lld], uniform random data, %d launches per timing, %d rounds\n", M, (long long)N, K, M, K, (long long)N, reps, rounds);
  // three A images: full; lower (A(i,k) = 0 for k > i); upper (A(i,k) = 0 for k < i).  Image element (k, i) at k * M + i.
  std::vector<double> hA((size_t)K * M), hAl, hAu, hB((size_t)K * N);
  srand(1);
  for (auto& x : hA) x = rand() / (double)RAND_MAX - 0.5;
  for (auto& x : hB) x = rand() / (double)RAND_MAX - 0.5;
  hAl = hA; hAu = hA;
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < M; ++i) { if (k > i) hAl[(size_t)k * M + i] = 0.0; if (k < i) hAu[(size_t)k * M + i] = 0.0; }
  double *dA[3], *dB, *dC; GemmTile* dT[6]; Stamp* dS;
  const std::vector<double>* hAs[3] = {&hA, &hAl, &hAu};
  for (int a = 0; a < 3; ++a) { CK(hipMalloc(&dA[a], sizeof(double) * hA.size())); CK(hipMemcpy(dA[a], hAs[a]->data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice)); }
  CK(hipMalloc(&dB, sizeof(double) * hB.size())); CK(hipMalloc(&dC, sizeof(double) * M * N));
  CK(hipMemcpy(dB, hB.data(), sizeof(double) * hB.size(), hipMemcpyHostToDevice));
  const int nbm = M / BM, nbn = (int)(N / BN), nk = K / BK;
  std::vector<GemmTile> tl[6] = {tiles_full_xcd_host(nbm, nbn, nk), tiles_trmm_paired_host(true, nbm, nbn), tiles_trmm_paired_host(false, nbm, nbn)};
  int per[6] = {1, 2, 2, 0, 0, 0};
  for (int a = 0; a < 3; ++a) tl[3 + a] = persistent(tl[a], per[a], per[3 + a]);     // lists 3..5: the persistent forms of 0..2
  for (int a = 0; a < 6; ++a) { CK(hipMalloc(&dT[a], sizeof(GemmTile) * tl[a].size())); CK(hipMemcpy(dT[a], tl[a].data(), sizeof(GemmTile) * tl[a].size(), hipMemcpyHostToDevice)); }
  CK(hipMalloc(&dS, sizeof(Stamp) * 4096));
  CK(hipMemset(dS, 0, sizeof(Stamp) * 4096));
  auto args = [&](int a) { GemmArgs g; g.seg[0].A = dA[a % 3]; g.seg[0].B = dB; g.seg[0].lda = M; g.seg[0].ldb = N; g.tiles = dT[a]; g.per = per[a]; g.C = dC; g.ldc = N; g.slice_stride = 0; g.alpha = 1.0; g.kscale = nullptr; return g; };
  const double flops[3] = {2.0 * M * (double)N * K, 1.0 * M * (double)N * K, 1.0 * M * (double)N * K};   // triangle-aware for the triangular products
  std::vector<double> ref[3], out((size_t)M * N);

  struct Var { std::string name; std::function<void()> launch; int grid; int a; bool is_ref; bool check; std::vector<double> ms; double diff = -1; double mhz = 0; };
  std::vector<Var> vars;
  const size_t shm_p = sizeof(double) * 2 * STAGE_DOUBLES;
#define ADDP(TRIK, A, NAME)                                                                                                            \
  {                                                                                                                                   \
    auto kern = gemm_f64_kernel<LAY_MNCONTIG, LAY_MNCONTIG, 2, false, TRIK, 8, EpiStore>;                                             \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_p));             \
    const GemmArgs g = args(A);                                                                                                       \
    const int grid = (int)tl[A].size() / per[A];                                                                                      \
    vars.push_back({NAME, [=] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shm_p, 0, g, EpiStore(), g, EpiStore(), grid); }, grid, A, true, false, {}}); \
  }
  ADDP(TRI_NONE, 0, "full  product kernel (8 waves, 2x4 sub-tiles)")
  ADDP(TRI_A_LOWER, 1, "lower product kernel (balanced pairs)")
  ADDP(TRI_A_UPPER, 2, "upper product kernel (balanced pairs)")
#define ADD(WAVES, WMW, TRIK, KO, A, XPF)                                                                                                  \
  {                                                                                                                                   \
    auto kern = k_v3<WAVES, WMW, TRIK, KO, XPF>;                                                                                           \
    const size_t shm = sizeof(double) * 2 * 2 * 16 * 144;                                                                             \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));               \
    const GemmArgs g = args(A);                                                                                                       \
    const int grid = (int)tl[A].size() / per[A];                                                                                      \
    char nm[160];                                                                                                                     \
    snprintf(nm, sizeof(nm), "%s v3 %dw %dx%d sub %dx%d ko%d%s%s", (A) % 3 == 0 ? "full " : (A) % 3 == 1 ? "lower" : "upper", WAVES, WMW, WAVES / WMW, 8 / WMW, 8 / (WAVES / WMW), KO, (A) >= 3 ? " persistent" : "", XPF ? " xpf" : ""); \
    Stamp* st = dS;                                                                                                                   \
    vars.push_back({nm, [=] { hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), shm, 0, g, st); }, grid, (A) % 3, false, (KO) == 0, {}}); \
  }
  ADD(8, 4, 0, 0, 0, 0)
#define ADDD(KO, NAME)                                                                                                                \
  {                                                                                                                                   \
    auto kern = k_dva<KO>;                                                                                                            \
    const size_t shm = sizeof(double) * 2 * 16 * 144;                                                                                 \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));               \
    const GemmArgs g = args(0);                                                                                                       \
    const int grid = (int)tl[0].size() / per[0];                                                                                      \
    Stamp* st = dS;                                                                                                                   \
    vars.push_back({NAME, [=] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shm, 0, g, st); }, grid, 0, false, (KO) == 0, {}});   \
  }
  ADDD(0, "full  dva: A operand direct to registers")
  ADDD(1, "full  dva ko1 (no barrier; timing only)")
  ADDD(4, "full  dva ko4 (no stores; timing only)")

  Timer tm;
  for (size_t i = 0; i < vars.size(); ++i) {
    CK(hipMemset(dC, 0, sizeof(double) * M * N));
    vars[i].launch();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    const int a = vars[i].a;
    if (vars[i].is_ref) {
      ref[a].resize((size_t)M * N);
      CK(hipMemcpy(ref[a].data(), dC, sizeof(double) * M * N, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int s = 0; s < 16; ++s) {
        const int ii = (s * 67 + 3) % M; const int64_t j = ((int64_t)s * 2039 + 11) % N;
        double acc = 0; for (int k = 0; k < K; ++k) acc += (*hAs[a])[(size_t)k * M + ii] * hB[(size_t)k * N + j];
        worst = std::max(worst, std::fabs(acc - ref[a][(size_t)ii * N + j]));
      }
      printf("%s vs host dot products (16 entries): max abs diff %.2e\n", vars[i].name.c_str(), worst);
    } else if (vars[i].check) {
      CK(hipMemcpy(out.data(), dC, sizeof(double) * M * N, hipMemcpyDeviceToHost));
      vars[i].diff = max_diff(ref[a], out);
    }
  }
  for (auto& v : vars) { for (int i = 0; i < 3; ++i) v.launch(); }
  CK(hipDeviceSynchronize());
  std::vector<Stamp> hs(4096);
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars) {
      v.ms.push_back(tm.run(v.launch, reps));
      if (!v.is_ref) {   // in-kernel clock of the last launch
        CK(hipMemcpy(hs.data(), dS, sizeof(Stamp) * v.grid, hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        for (int w = 0; w < v.grid; ++w) if (hs[w].r1 > hs[w].r0) mhz.push_back((double)(hs[w].t1 - hs[w].t0) / (double)(hs[w].r1 - hs[w].r0) * 100.0);
        std::sort(mhz.begin(), mhz.end());
        if (!mhz.empty()) v.mhz = mhz[mhz.size() / 2];
      }
    }
  for (auto& v : vars) {
    std::vector<double> s = v.ms; std::sort(s.begin(), s.end());
    printf("%-50s best %7.3f ms %6.2f TF | median %6.2f TF | clock %4.0f MHz", v.name.c_str(), s[0], flops[v.a] / s[0] * 1e-9, flops[v.a] / s[s.size() / 2] * 1e-9, v.mhz);
    if (v.diff >= 0) printf(" | max|diff| %.1e", v.diff);
    printf("\n");
  }
  return 0;
}
