// C-ABI of the data-parallel exchange (include/zigp.h, "data-parallel exchange"): communicator set-up and the host-vector all-reduce.
// The per-step reduction itself is comm_allreduce (zigp_comm.h), called by the dense and Kronecker drivers on their packed device vectors.
#include "zigp_host.h"
#include "zigp_comm.h"

using namespace zigp;

extern "C" {

int zigp_comm_unique_id(void* id128) {
  if (!id128) return ZIGP_EARG;
  RcclApi* api = rccl_api(nullptr);
  if (!api) return ZIGP_ECOMM;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return ZIGP_ECOMM;
  static_assert(sizeof(ncclUniqueId) == ZIGP_COMM_ID_BYTES, "zigp.h promises a 128-byte id");
  memcpy(id128, &id, sizeof(id));
  return ZIGP_OK;
}

int zigp_comm_init(zigp_ctx* c, int32_t rank, int32_t nranks, const void* id128) {
  if (!c) return ZIGP_EARG;
  if (!id128 || nranks <= 0 || rank < 0 || rank >= nranks) return fail_arg(c, "zigp_comm_init: need 0 <= rank < nranks and the 128-byte id of zigp_comm_unique_id");
  if (c->comm) return fail_arg(c, "zigp_comm_init: this context already has a communicator (zigp_comm_destroy first)");
  ZIGP_HIP(c, hipSetDevice(c->device));
  RcclApi* api = rccl_api(&c->err);
  if (!api) return ZIGP_ECOMM;
  // ncclCommInitRank is collective and has no timeout of its own: a peer that never arrives (crashed, RCCL not loadable there, another
  // id) would block this rank forever.  It runs on a helper thread; if it has not returned after comm_timeout_s (zigp_comm_set_timeout,
  // default 120 s, env ZIGP_COMM_TIMEOUT_S) the call gives up with ZIGP_ECOMM and leaves the helper behind (detached; the state it
  // writes to is shared-owned).  If the helper's ncclCommInitRank does return later, nobody will ever use that communicator: the helper
  // destroys it itself (`abandoned`).  While it is still blocked it holds RCCL's bootstrap sockets and a thread: after a timeout the
  // process should fall back to the torch.distributed exchange or exit (os._exit or a fresh child process) -- never retry on this id.
  struct InitJob {
    std::mutex m; std::condition_variable cv; bool done = false, abandoned = false;
    ncclResult_t r = ncclSuccess; hipError_t he = hipSuccess; ncclComm_t comm = nullptr; ncclUniqueId id;
  };
  auto job = std::make_shared<InitJob>();
  memcpy(&job->id, id128, sizeof(job->id));
  const int device = c->device;
  std::thread([job, api, device, rank, nranks]() {
    ncclComm_t comm = nullptr;
    ncclResult_t r = ncclSuccess;
    const hipError_t he = hipSetDevice(device);          // the current device is per thread
    if (he == hipSuccess) r = api->CommInitRank(&comm, nranks, job->id, rank);
    bool late;
    {
      std::lock_guard<std::mutex> g(job->m);
      job->he = he; job->r = r; job->comm = comm; job->done = true;
      late = job->abandoned;
      job->cv.notify_all();
    }
    if (late && he == hipSuccess && r == ncclSuccess && comm) (void)api->CommDestroy(comm);   // the caller gave up on it: do not leak it
  }).detach();
  {
    std::unique_lock<std::mutex> g(job->m);
    if (!job->cv.wait_for(g, std::chrono::duration<double>(c->comm_timeout_s), [&] { return job->done; })) {
      job->abandoned = true;        // (under the lock: the helper sees it when it finishes)
      char b[256];
      snprintf(b, sizeof(b), "zigp_comm_init: ncclCommInitRank(rank %d of %d) did not return within %.0f s -- a peer never joined; "
               "giving up on this communicator", (int)rank, (int)nranks, c->comm_timeout_s);
      c->err = b;
      return ZIGP_ECOMM;
    }
  }
  if (job->he != hipSuccess) { c->err = "zigp_comm_init: hipSetDevice failed on the helper thread"; return ZIGP_EHIP; }
  if (job->r != ncclSuccess) return fail_comm(c, api, "ncclCommInitRank", job->r);
  c->comm = job->comm; c->comm_rank = rank; c->comm_nranks = nranks; c->comm_calls = 0;
  return ZIGP_OK;
}

int zigp_comm_available(int32_t* version) {
  std::string err;
  RcclApi* api = rccl_api(&err);
  if (version) *version = api ? api->version : 0;
  return api ? ZIGP_OK : ZIGP_ECOMM;
}

int zigp_comm_set_timeout(zigp_ctx* c, double seconds) {
  if (!c) return ZIGP_EARG;
  if (!(seconds > 0)) return fail_arg(c, "zigp_comm_set_timeout: seconds must be > 0");
  c->comm_timeout_s = seconds;
  return ZIGP_OK;
}

int zigp_comm_destroy(zigp_ctx* c) {
  if (!c) return ZIGP_EARG;
  if (!c->comm) return ZIGP_OK;
  ZIGP_HIP(c, hipSetDevice(c->device));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream_main));
  RcclApi* api = rccl_api(&c->err);
  if (!api) return ZIGP_ECOMM;
  const ncclResult_t r = api->CommDestroy(static_cast<ncclComm_t>(c->comm));
  c->comm = nullptr; c->comm_rank = 0; c->comm_nranks = 1;
  if (r != ncclSuccess) return fail_comm(c, api, "ncclCommDestroy", r);
  return ZIGP_OK;
}

int zigp_comm_allreduce_host(zigp_ctx* c, double* inout, int64_t n) {
  if (!c) return ZIGP_EARG;
  if (!inout || n <= 0) return fail_arg(c, "zigp_comm_allreduce_host: bad arguments");
  if (!c->comm) return fail_arg(c, "zigp_comm_allreduce_host: no communicator (zigp_comm_init first)");
  ZIGP_HIP(c, hipSetDevice(c->device));
  ZIGP_TRY(begin_staged_call(c));
  ZIGP_ENSURE(c, c->packed, (size_t)n);
  ZIGP_PINNED(c, h, (size_t)n);
  memcpy(h, inout, sizeof(double) * n);
  ZIGP_HIP(c, hipMemcpyAsync(c->packed.p, h, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  ZIGP_TRY(comm_allreduce(c, c->packed.p, (size_t)n));
  double* out = nullptr;
  ZIGP_TRY(download(c, c->packed.p, (size_t)n, &out));
  ZIGP_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(inout, out, sizeof(double) * n);
  return ZIGP_OK;
}

int zigp_comm_info(zigp_ctx* c, int32_t* rank, int32_t* nranks, int64_t* allreduce_calls) {
  if (!c) return ZIGP_EARG;
  if (rank) *rank = c->comm ? c->comm_rank : 0;
  if (nranks) *nranks = c->comm ? c->comm_nranks : 0;
  if (allreduce_calls) *allreduce_calls = c->comm_calls;
  return ZIGP_OK;
}

}  // extern "C"
