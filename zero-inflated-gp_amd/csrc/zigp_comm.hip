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
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = api->CommInitRank(&comm, nranks, id, rank);
  if (r != ncclSuccess) return fail_comm(c, api, "ncclCommInitRank", r);
  c->comm = comm; c->comm_rank = rank; c->comm_nranks = nranks; c->comm_calls = 0;
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
