// Context, error handling, device-buffer and profiling helpers shared by the dense and Kronecker paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>
#include "../../include/zigp.h"
#include "../../include/zigp_diag.h"
#include "zigp_gemm.h"

namespace zigp {

struct DevBuf {
  double* p = nullptr;
  size_t cap = 0;  // doubles
  bool owned = true;   // false: p points into another buffer (alias)
  int ensure(size_t n) {
    if (!owned) { p = nullptr; cap = 0; owned = true; }
    if (n <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipMalloc((void**)&p, n * sizeof(double));
    if (e != hipSuccess) return -1;
    cap = n;
    return 0;
  }
  void alias(double* q, size_t n) { release(); p = q; cap = n; owned = false; }   // a view of n doubles at q (owned by someone else)
  void release() { if (p && owned) (void)hipFree(p); p = nullptr; cap = 0; owned = true; }
};

enum ProfClass { PC_GEMM_A1 = 0, PC_GEMM_A2 = 1, PC_GEMM_H = 2, PC_GEMM_J = 3, PC_SYR2K = 4, PC_KUF = 5, PC_POINT = 6, PC_RED = 7, PC_MXM = 8, PC_OTHER = 9 };

struct TileList {
  GemmTile* d = nullptr;
  int n = 0;     // list entries
  int per = 1;   // entries per workgroup (n is a multiple of it)
};

// Per-latent (f or g) device state of the dense path
struct Latent {
  int M = 0, Mp = 0;
  DevBuf Z, ell, u, s, s2;              // Z (Mp,D) zero padded; u,s,s2 (Mp) zero padded
  double zc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // mean inducing input (host copy): centre of k_kgrad's moment sums
  bool kg_exact = false;                     // inducing inputs spread over > KG_EXACT_SPREAD lengthscales: k_kgrad forms x - z_m per row (no centre shift)
  DevBuf Zs;                            // Z scaled by KUF_C / ell_d (k_kuf_build's units), same padding
  double var = 1.0;
  DevBuf Kuu, L, W;                      // (Mp,Mp)
  DevBuf K, A1, Jp;                      // chunk panels [Mp][Nc]: Kuf, A1 = W K, J' = Q W^T A1 (A2 = W^T A1 is reduced to its column sums and never stored: r6)
  DevBuf Wp, a1gm;                       // W diag(s^2) (Mp,Mp); running sum of A1 gm [Mp]
  DevBuf Wt, Wpt;                        // W^T, (W diag(s^2))^T (Mp,Mp): the m-contiguous images the lower-triangular products read
  DevBuf P, Qt, Rt;                      // gradient steps: P = W^T W = Kuu^-1, Qt = diag(s^2) P - I = Q^T, Rt = W Qt = (Q W^T)^T: J' = Q A2 = (Q W^T) A1 (zigp_dense.hip, chunk_forward)
  bool P_ready = false;                  // P of THIS call's parameters is in `P` (the reverse M x M stage takes it from there)
  DevBuf part;                           // [3][Mp/32][Nc] partial rows of the fused column sums: v^T A1, sum A1^2, sum s^2 A2^2
  DevBuf gm, gv;                         // cotangents of mean / var per column [Nc]
  DevBuf du, dsq, krow;                  // row accumulators: du[Mp], dsq[Mp], krow[Mp][1+2D]
  DevBuf dLpart;                         // [S][Mp*Mp] split-K partials of the rank-N updates
  DevBuf T1, T2, T3, G;                  // MxM scratch
  DevBuf sk;                             // [S][Mp*Mp] split-K planes of the O(M^3) products of the reverse pass
  DevBuf vec;                            // small vectors: v=W u [Mp], alpha [Mp], dkinv [Mp], scal[8]
};

struct KronState;   // Kronecker-path buffers (zigp_kron.hip)
struct KfState;     // fused Kronecker path (zigp_kronf.hip)

// Page-locked host staging for the small per-step transfers (parameters in, sums and gradients out).  A copy between
// device and PAGEABLE host memory makes the runtime stage and wait; through this arena the H2D copies are truly
// asynchronous and all D2H results of a call are collected by its single final stream synchronisation.  Blocks are never
// moved or freed while a call is in flight: alloc() only appends, reset() (start of every API call -- the previous
// call ended with a synchronisation) rewinds.
struct PinnedArena {
  struct Block { char* p; size_t cap, used; };
  std::vector<Block> blocks;
  void reset() { for (auto& b : blocks) b.used = 0; }
  void* alloc(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    for (auto& b : blocks)
      if (b.cap - b.used >= bytes) { void* r = b.p + b.used; b.used += bytes; return r; }
    Block nb; nb.cap = bytes > ((size_t)1 << 20) ? bytes : ((size_t)1 << 20); nb.used = bytes; nb.p = nullptr;
    if (hipHostMalloc((void**)&nb.p, nb.cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    blocks.push_back(nb);
    return nb.p;
  }
  void release() { for (auto& b : blocks) (void)hipHostFree(b.p); blocks.clear(); }
};

}  // namespace zigp

struct zigp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;    // stream every launch helper enqueues on (swapped to stream2 inside a TwoStream section)
  hipStream_t stream_main = nullptr, stream2 = nullptr;
  hipStream_t stream3 = nullptr;   // dense path: buffers, zeroed accumulators and the first chunk's Kuf panels, under the M x M forward of both latents
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_prep_fork = nullptr, ev_prep = nullptr;
  std::string err;
  int info = 0;
  int64_t fit_steps_applied = 0;       // zigp_kron_fit_steps: updates applied by the LAST call (all of them, or the ones before a failing step)
  int64_t chunk = 32768;
  bool chunk_auto = true;                // no zigp_set_chunk yet: the chunk follows M (32768 rows at M = 1024, more for smaller M)
  // data
  const double* dX = nullptr; const double* dY = nullptr;
  zigp::DevBuf ownX, ownY;
  int64_t N = 0; int D = 0;
  // zigp_select_rows: the active data (dX, dY, N) is a gathered batch of the resident set (fullX, fullY, fullN)
  const double* fullX = nullptr; const double* fullY = nullptr; int64_t fullN = 0;
  zigp::DevBuf selX, selY, selIdx;
  // dense path state
  zigp::Latent lat[2];
  zigp::DevBuf pw_part;                 // pointwise block partials
  // mean function of f, m(x) = mean_b + mean_a . x (zigp_set_mean_function), and its gradient from the last zigp_elbo
  bool fwd_kuf_side = true;             // value-only / predict passes: the next chunk's Kuf panels on the side stream behind this chunk's A1 (env ZIGP_FWD_KUF_SIDE=0: off)
  bool trmm_tail = true;                // merged triangular launches: re-deal the last, partly filled wave (tiles_trmm, zigp_host.h); env ZIGP_TRMM_TAIL=0 turns it off
  int overlap = 1;                      // zigp_set_overlap: 1 (default) = HBM-bound side kernels of a chunk on stream2 under its SYRKs
  bool mean_on = false;
  double mean_a[8] = {0}, mean_b = 0.0, mean_da[8] = {0}, mean_db = 0.0;   // 8 = zigp::MAXD (zigp_kernels.h)
  zigp::DevBuf out9;                    // predict outputs (9,Nc)
  zigp::DevBuf scratch, scratch2;       // misc
  int* d_info = nullptr;
  zigp::PinnedArena pinned;             // host staging of the per-step transfers
  zigp::KronState* kron = nullptr;
  void (*kron_free)(zigp::KronState*) = nullptr;
  zigp::KfState* kronf = nullptr;
  void (*kronf_free)(zigp::KfState*) = nullptr;
  bool kron_panels = false;             // zigp_set_kron_panels: force the panel (GEMM-core) Kronecker path
  int kron_range_tiles = 1024;          // larger-grid fused backward: rows go through in ranges of this many 16-point tiles (bounded operand spill; zigp_set_kron_range_tiles)
  std::map<std::string, zigp::TileList> tiles;
  // data-parallel exchange (zigp_comm_init): RCCL communicator, one rank per context / GPU
  void* comm = nullptr; int comm_rank = 0, comm_nranks = 1; int64_t comm_calls = 0;
  double comm_timeout_s = 120.0;         // zigp_comm_set_timeout: how long zigp_comm_init waits for its peers
  zigp::DevBuf packed;                  // result vector of the dense path (k_dense_pack)
  zigp::DevBuf parm;                    // parameters of both latents, one staged image (latents_upload); lat[h].Z / ell / u / s are views into it
  double pivot_rtol = 8.0;              // zigp_set_pivot_rtol: a Cholesky pivot <= pivot_rtol * eps * (variance + jitter) is ZIGP_ENOTPD
  // profiling
  bool prof_on = false;
  int prof_every = 8;                    // chunk-loop launches are timed on every prof_every-th chunk (event pairs cost ~10 us)
  bool prof_skip = false;                // set by the chunk loop for the chunks that are not sampled
  int64_t prof_total[ZIGP_NCLASS] = {0}; // all launches per class, sampled or not
  double prof_ms[ZIGP_NCLASS] = {0};
  int64_t prof_n[ZIGP_NCLASS] = {0};
  double prof_flops[ZIGP_NCLASS] = {0};
  struct PendingEv { hipEvent_t a, b; int cls; };
  std::vector<PendingEv> pending;
  std::vector<hipEvent_t> ev_pool;
};

namespace zigp {

#define ZIGP_HIP(ctx, expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      char _b[512];                                                                      \
      snprintf(_b, sizeof(_b), "HIP error '%s' at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
      (ctx)->err = _b;                                                                   \
      return ZIGP_EHIP;                                                                  \
    }                                                                                    \
  } while (0)

#define ZIGP_TRY(expr)            \
  do {                            \
    int _rc = (expr);             \
    if (_rc != 0) return _rc;     \
  } while (0)

#define ZIGP_ENSURE(ctx, buf, n)                                        \
  do {                                                                  \
    if ((buf).ensure((size_t)(n)) != 0) {                               \
      (ctx)->err = "hipMalloc failed for " #buf;                        \
      return ZIGP_EHIP;                                                 \
    }                                                                   \
  } while (0)

inline int fail_arg(zigp_ctx* c, const char* msg) { c->err = msg; return ZIGP_EARG; }

// Two independent launch chains (the MxM stages of latents f and g) on two HIP streams: fork() after the work both
// depend on, second() switches the helpers to stream2, join() makes the main stream wait for it.  The destructor
// restores the main stream on every exit path.
struct TwoStream {
  zigp_ctx* c;
  explicit TwoStream(zigp_ctx* c_) : c(c_) {}
  ~TwoStream() { c->stream = c->stream_main; }
  int fork() {
    ZIGP_HIP(c, hipEventRecord(c->ev_fork, c->stream_main));
    ZIGP_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    return 0;
  }
  void second() { c->stream = c->stream2; }
  void first() { c->stream = c->stream_main; }
  int join() {
    c->stream = c->stream_main;
    ZIGP_HIP(c, hipEventRecord(c->ev_join, c->stream2));
    ZIGP_HIP(c, hipStreamWaitEvent(c->stream_main, c->ev_join, 0));
    return 0;
  }
};

// RAII-less profiling bracket: call prof_begin before and prof_end after a group of launches.
struct ProfScope {
  zigp_ctx* c; hipEvent_t a = nullptr, b = nullptr; int cls; bool on;
  static hipEvent_t get_ev(zigp_ctx* c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; return e;
  }
  ProfScope(zigp_ctx* c_, int cls_, double flops = 0.0) : c(c_), cls(cls_), on(c_->prof_on) {
    if (!on) return;
    c->prof_total[cls] += 1;
    if (c->prof_skip) { on = false; return; }
    a = get_ev(c); b = get_ev(c);
    if (!a || !b) { on = false; return; }
    (void)hipEventRecord(a, c->stream);
    c->prof_flops[cls] += flops;
    c->prof_n[cls] += 1;
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, c->stream);
    c->pending.push_back({a, b, cls});
  }
};

inline void prof_collect(zigp_ctx* c) {
  for (auto& p : c->pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->prof_ms[p.cls] += ms;
    c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b);
  }
  c->pending.clear();
}

}  // namespace zigp
