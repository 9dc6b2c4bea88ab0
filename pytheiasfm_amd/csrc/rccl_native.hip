// rccl_native.hip -- the sharded BA solve calling RCCL itself (tracks sharded over one process per GPU,
// the packed reduced camera system all-reduced over xGMI every LM iteration).
//
// The library keeps no link-time dependency on RCCL: the entry points are looked up at run time in the copy of
// librccl the process already loaded (torch.distributed's "nccl" backend on ROCm IS RCCL), else in /opt/rocm/lib.
// The host only has to carry the 128-byte ncclUniqueId from rank 0 to the other ranks (any channel: MPI, a file,
// torch.distributed.broadcast_object_list); after that every all-reduce is `ncclAllReduce` on the solve's own HIP
// stream -- no interpreter, no callback trampoline inside the LM iteration.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "theia_hip.h"
#include "theia_hip_internal.h"

namespace {

// the few declarations of rccl.h this file needs (ABI of NCCL 2.x / RCCL)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclSum = 0, ncclMax = 2 };
enum { ncclFloat64 = 8 };
typedef int (*fn_get_id)(ncclUniqueId*);
typedef int (*fn_init_rank)(ncclComm_t*, int, ncclUniqueId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*fn_destroy)(ncclComm_t);
typedef const char* (*fn_err)(int);
typedef int (*fn_count)(const ncclComm_t, int*);

struct Rccl {
  void* lib = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_err err = nullptr;
  fn_count count = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.lib) break; }   // the copy already in the process
    if (!r.lib) for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
    if (!r.lib) return;
    r.get_id = (fn_get_id)dlsym(r.lib, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(r.lib, "ncclCommInitRank");
    r.all_reduce = (fn_all_reduce)dlsym(r.lib, "ncclAllReduce");
    r.destroy = (fn_destroy)dlsym(r.lib, "ncclCommDestroy");
    r.err = (fn_err)dlsym(r.lib, "ncclGetErrorString");
    r.count = (fn_count)dlsym(r.lib, "ncclCommCount");
    r.ok = r.get_id && r.init_rank && r.all_reduce && r.destroy;
  });
  return r;
}

struct NativeCtx { ncclComm_t comm; };

int native_allreduce(void* ctx, void* buf, size_t count, int op, void* stream) {
  NativeCtx* c = static_cast<NativeCtx*>(ctx);
  const int rc = rccl().all_reduce(buf, buf, count, ncclFloat64, op == THEIA_REDUCE_MAX ? ncclMax : ncclSum, c->comm, (hipStream_t)stream);
  return rc == ncclSuccess ? 0 : rc;
}

}  // namespace

extern "C" {

int theia_hip_rccl_unique_id(void* out128) {
  if (!out128) return thip::set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null output");
  if (!rccl().ok) return thip::set_error(THEIA_HIP_ERR_UNSUPPORTED, "librccl not found in the process nor under /opt/rocm/lib");
  ncclUniqueId id;
  const int rc = rccl().get_id(&id);
  if (rc != ncclSuccess) return thip::set_error(THEIA_HIP_ERR_INTERNAL, "ncclGetUniqueId failed (%d)", rc);
  std::memcpy(out128, &id, sizeof(id));
  return 0;
}

int theia_hip_rccl_comm_create(const void* id128, int32_t rank, int32_t world_size, void** comm_out) {
  if (!id128 || !comm_out) return thip::set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (world_size < 1 || rank < 0 || rank >= world_size) return thip::set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "rank / world_size out of range");
  if (!rccl().ok) return thip::set_error(THEIA_HIP_ERR_UNSUPPORTED, "librccl not found in the process nor under /opt/rocm/lib");
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  const int rc = rccl().init_rank(&comm, world_size, id, rank);
  if (rc != ncclSuccess) return thip::set_error(THEIA_HIP_ERR_INTERNAL, "ncclCommInitRank failed (%d: %s)", rc, rccl().err ? rccl().err(rc) : "?");
  NativeCtx* c = new NativeCtx{comm};
  *comm_out = c;
  return 0;
}

int theia_hip_rccl_comm_destroy(void* comm) {
  if (!comm) return 0;
  NativeCtx* c = static_cast<NativeCtx*>(comm);
  if (rccl().ok && c->comm) (void)rccl().destroy(c->comm);
  delete c;
  return 0;
}

int theia_hip_rccl_comm_count(void* comm, int32_t* count_out) {
  if (!comm || !count_out) return thip::set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  NativeCtx* c = static_cast<NativeCtx*>(comm);
  if (!rccl().ok || !rccl().count) return thip::set_error(THEIA_HIP_ERR_UNSUPPORTED, "ncclCommCount not found in librccl");
  int n = 0;
  const int rc = rccl().count(c->comm, &n);
  if (rc != ncclSuccess) return thip::set_error(THEIA_HIP_ERR_INTERNAL, "ncclCommCount failed (%d)", rc);
  *count_out = n;
  return 0;
}

int theia_hip_ba_set_rccl(theia_ba_handle h, void* comm, int32_t rank, int32_t world_size) {
  if (!h || !comm) return thip::set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  int rc = theia_hip_ba_set_allreduce(h, native_allreduce, comm);
  if (!rc) rc = theia_hip_ba_set_shard(h, rank, world_size);
  return rc;
}

}  // extern "C"
