// Data-parallel exchange inside the library: one ncclAllReduce(sum, f64) of a call's packed result
// vector, on the device, on the stream the call's kernels ran on.  The reference is single-process (no collective to replace).
// RCCL is bound at run time (dlopen of librccl.so.1 on the first zigp_comm_* call): the library carries no link-time dependency
// on it, and a process that already holds PyTorch's copy of RCCL (same soname) shares it.
#pragma once
#include "zigp_ctx.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace zigp {

struct RcclApi {
  void* handle = nullptr;
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
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy || !api.GetErrorString) {
    r.err = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclGetErrorString";
    dlclose(h);
    return r;
  }
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
