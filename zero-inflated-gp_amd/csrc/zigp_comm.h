// Data-parallel exchange inside the library: one ncclAllReduce(sum, f64) of a call's packed result
// vector, on the device, on the stream the call's kernels ran on.  The reference is single-process (no collective to replace).
// RCCL is bound at run time (dlopen of librccl.so.1 on the first zigp_comm_* call): the library carries no link-time dependency
// on it, and a process that already holds PyTorch's copy of RCCL (same soname) shares it.
#pragma once
#include "zigp_ctx.h"
#include <dlfcn.h>
// The six RCCL entry points this unit binds with dlsym.  With the rccl development headers installed their declarations (and the
// version the ABI below was compiled against) come from <rccl/rccl.h>; without them the library still builds -- single-GPU use needs
// no RCCL at all -- from the minimal declarations below (the NCCL 2.x C API these functions have had since 2.0: a 128-byte opaque id,
// an opaque communicator handle, int-valued enums).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define ZIGP_RCCL_HEADER_VERSION NCCL_VERSION_CODE
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;      // ncclFloat64
ncclResult_t ncclGetVersion(int* version);
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char* ncclGetErrorString(ncclResult_t result);
}
#define ZIGP_RCCL_HEADER_VERSION 0
#endif
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

namespace zigp {

struct RcclApi {
  void* handle = nullptr;
  int version = 0;            // ncclGetVersion of the library dlopen found
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

struct RcclLoad { RcclApi api; std::string err; };
inline RcclLoad rccl_load() {
  RcclLoad r;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { const char* e = dlerror(); r.err = std::string("dlopen(librccl.so.1) failed: ") + (e ? e : "unknown error"); return r; }
  RcclApi& api = r.api;
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(h, "ncclGetVersion"));
  if (!api.GetVersion || !api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy || !api.GetErrorString) {
    r.err = "librccl.so.1 lacks one of ncclGetVersion / ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclGetErrorString";
    dlclose(h);
    return r;
  }
  // The by-value 128-byte id and the enum values are an ABI promise of the NCCL 2.x line: refuse anything else, and a library of
  // another MAJOR version than the headers this unit was compiled against (the run-time library is whatever dlopen finds first).
  int ver = 0;
  if (api.GetVersion(&ver) != ncclSuccess || ver < 20000 || ver >= 30000 ||
      (ZIGP_RCCL_HEADER_VERSION != 0 && ver / 10000 != ZIGP_RCCL_HEADER_VERSION / 10000)) {
    char b[200];
    snprintf(b, sizeof(b), "librccl.so.1 reports version code %d: not the NCCL 2.x ABI this library binds (headers: %d)", ver, (int)ZIGP_RCCL_HEADER_VERSION);
    r.err = b;
    dlclose(h);
    return r;
  }
  api.version = ver;
  api.handle = h;
  return r;
}
// loaded once per process (function-local static: initialised exactly once also when contexts live on several threads)
inline RcclApi* rccl_api(std::string* err) {
  static RcclLoad l = rccl_load();
  if (!l.api.handle) { if (err) *err = l.err; return nullptr; }
  return &l.api;
}

inline int fail_comm(zigp_ctx* c, RcclApi* api, const char* what, ncclResult_t r) {
  char b[384];
  snprintf(b, sizeof(b), "RCCL error in %s: %s", what, api ? api->GetErrorString(r) : "library not loaded");
  c->err = b;
  return ZIGP_ECOMM;
}

// Sum `n` doubles at `dev` over the ranks of the context's communicator, in place, on c->stream.  No-op without a communicator.
// Every rank must call it with the same n (the result vectors are sized by the model, never by the shard).
inline int comm_allreduce(zigp_ctx* c, double* dev, size_t n) {
  if (!c->comm) return 0;
  RcclApi* api = rccl_api(&c->err);
  if (!api) return ZIGP_ECOMM;
  const ncclResult_t r = api->AllReduce(dev, dev, n, ncclDouble, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream);
  if (r != ncclSuccess) return fail_comm(c, api, "ncclAllReduce", r);
  c->comm_calls += 1;
  return 0;
}

}  // namespace zigp
