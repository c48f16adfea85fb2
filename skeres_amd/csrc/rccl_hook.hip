// A native sk_allreduce_fn over RCCL, so that a C, C++ or JVM caller of the multi-GPU path needs neither PyTorch nor a
// collective of its own.  librccl.so is opened at run time (dlopen) the first time one of these entry
// points is used: libskeres_amd.so has no link-time dependency on it and loads on machines without RCCL.
//
//   rank 0:            sk_rccl_unique_id(id)            -> 128 bytes, handed to the other ranks by the caller's own means
//   every rank:        h = sk_allreduce_rccl_init(rank, world, id)      (ncclCommInitRank on the current device)
//        or            h = sk_allreduce_rccl_create(existing ncclComm_t)
//                      sk_options_set_distributed(o, rank, world, sk_allreduce_rccl_fn(), h)
//                      ... sk_solve / sk_solver_* ...
//                      sk_allreduce_rccl_free(h)
// The collective is an in-place ncclAllReduce(double, sum) on the stream the solver hands the hook.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/skeres_amd.h"
#include "common.hpp"

namespace {

typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId_ { char internal[128]; };
typedef int (*allreduce_t)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*unique_id_t)(ncclUniqueId_*);
typedef int (*init_rank_t)(ncclComm_t*, int, ncclUniqueId_, int);
typedef int (*destroy_t)(ncclComm_t);
typedef const char* (*error_string_t)(int);

struct Rccl {
  void* lib = nullptr;
  allreduce_t all_reduce = nullptr;
  unique_id_t get_unique_id = nullptr;
  init_rank_t comm_init_rank = nullptr;
  destroy_t comm_destroy = nullptr;
  error_string_t error_string = nullptr;
};

// nullptr (and sk_last_error) when RCCL cannot be opened
const Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.all_reduce = reinterpret_cast<allreduce_t>(dlsym(r.lib, "ncclAllReduce"));
    r.get_unique_id = reinterpret_cast<unique_id_t>(dlsym(r.lib, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<init_rank_t>(dlsym(r.lib, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<destroy_t>(dlsym(r.lib, "ncclCommDestroy"));
    r.error_string = reinterpret_cast<error_string_t>(dlsym(r.lib, "ncclGetErrorString"));
  });
  if (!r.lib || !r.all_reduce || !r.get_unique_id || !r.comm_init_rank || !r.comm_destroy) {
    sk::set_error("RCCL is not available: librccl.so could not be opened (or lacks ncclAllReduce / ncclCommInitRank)");
    return nullptr;
  }
  return &r;
}

constexpr int kNcclDouble = 8, kNcclSum = 0;  // rccl.h: ncclFloat64 = ncclDouble = 8, ncclSum = 0

}  // namespace

struct sk_rccl {
  ncclComm_t comm = nullptr;
  bool owned = false;  // created by sk_allreduce_rccl_init: destroyed by sk_allreduce_rccl_free
  long calls = 0;
};

extern "C" {

int sk_rccl_unique_id(void* id128) {
  if (!id128) { sk::set_error("sk_rccl_unique_id: null buffer"); return SK_ERR_INVALID_ARGUMENT; }
  const Rccl* r = rccl();
  if (!r) return SK_ERR_COMM;
  ncclUniqueId_ id;
  const int rc = r->get_unique_id(&id);
  if (rc != 0) { sk::set_error("ncclGetUniqueId failed: %s", r->error_string ? r->error_string(rc) : "?"); return SK_ERR_COMM; }
  std::memcpy(id128, id.internal, sizeof(id.internal));
  return SK_OK;
}

sk_rccl* sk_allreduce_rccl_init(int rank, int world, const void* id128) {
  // (a NULL handle carries a typed status — sk_last_status — as sk_solver_create's does: an argument, or the communicator)
  sk::set_status(SK_OK);
  if (!id128 || world < 1 || rank < 0 || rank >= world) { sk::set_error("sk_allreduce_rccl_init: invalid argument"); sk::set_status(SK_ERR_INVALID_ARGUMENT); return nullptr; }
  sk::set_status(SK_ERR_COMM);  // (every failure below is the communicator's: librccl missing, no device, ncclCommInitRank)
  const Rccl* r = rccl();
  if (!r) return nullptr;
  ncclUniqueId_ id;
  std::memcpy(id.internal, id128, sizeof(id.internal));
  ncclComm_t comm = nullptr;
  const int rc = r->comm_init_rank(&comm, world, id, rank);
  if (rc != 0 || !comm) { sk::set_error("ncclCommInitRank failed: %s", r->error_string ? r->error_string(rc) : "?"); return nullptr; }
  sk_rccl* h = new (std::nothrow) sk_rccl();
  if (!h) { (void)r->comm_destroy(comm); return nullptr; }
  h->comm = comm; h->owned = true;
  sk::set_status(SK_OK);
  return h;
}

sk_rccl* sk_allreduce_rccl_create(void* nccl_comm) {
  if (!nccl_comm) { sk::set_error("sk_allreduce_rccl_create: null communicator"); return nullptr; }
  if (!rccl()) return nullptr;
  sk_rccl* h = new (std::nothrow) sk_rccl();
  if (h) h->comm = static_cast<ncclComm_t>(nccl_comm);
  return h;
}

void sk_allreduce_rccl_free(sk_rccl* h) {
  if (!h) return;
  if (h->owned && h->comm) { const Rccl* r = rccl(); if (r) (void)r->comm_destroy(h->comm); }
  delete h;
}

long sk_allreduce_rccl_calls(const sk_rccl* h) { return h ? h->calls : 0; }

static int rccl_allreduce_hook(void* user, double* device_buffer, size_t count, void* hip_stream) {
  sk_rccl* h = static_cast<sk_rccl*>(user);
  const Rccl* r = rccl();
  if (!h || !h->comm || !r) return 1;
  ++h->calls;
  if (count == 0) return 0;
  return r->all_reduce(device_buffer, device_buffer, count, kNcclDouble, kNcclSum, h->comm, static_cast<hipStream_t>(hip_stream)) == 0 ? 0 : 1;
}

sk_allreduce_fn sk_allreduce_rccl_fn(void) { return rccl_allreduce_hook; }

}  // extern "C"
