// The one exchange of the path: an all-gather of each rank's finished samples over RCCL / xGMI,
// as a C-ABI entry point (SURVEY.md 8b lists ld_allgather in the minimum export set).
//
// RCCL is reached through dlopen, not linked: the library must load on a box without RCCL (every other entry point
// works there), and inside a PyTorch process it must use the librccl.so.1 that torch already mapped rather than pull a
// second copy of the runtime into the process -- dlopen by SONAME with RTLD_NOLOAD finds the mapped one first.
// The default product path keeps the collective in torch.distributed (INTEGRATION.md says why); this is the same
// call for a caller that has no process group: ld_comm_unique_id on rank 0, the 128 bytes handed to every rank by
// whatever channel the caller has (file, socket, MPI, torch's TCPStore), ld_comm_init everywhere, ld_allgather on the
// compute stream, ld_comm_destroy.
#include "common.hip.h"
#include <mutex>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

namespace {
typedef struct { char internal[128]; } UniqueId;                       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*ErrStrFn)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_id = nullptr;
  CommInitRankFn init = nullptr;
  AllGatherFn allgather = nullptr;
  CommDestroyFn destroy = nullptr;
  ErrStrFn errstr = nullptr;
  char why[256] = "";
};

void rccl_load(Rccl& r) {
  const char* names[4] = {getenv("LD_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  const char* last_err = nullptr;
  for (int pass = 0; pass < 2 && !r.handle; ++pass)                     // pass 0: a copy the process already mapped
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (r.handle) break;
      const char* e = dlerror();                                        // ONE read: dlerror() clears the message
      if (e && pass == 1) last_err = e;
    }
  if (!r.handle) {
    snprintf(r.why, sizeof(r.why), "librccl.so.1 not found (%s); set LD_RCCL_PATH", last_err ? last_err : "dlopen failed");
    return;
  }
  r.get_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
  r.init = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
  r.allgather = (AllGatherFn)dlsym(r.handle, "ncclAllGather");
  r.destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
  r.errstr = (ErrStrFn)dlsym(r.handle, "ncclGetErrorString");
  if (!r.get_id || !r.init || !r.allgather || !r.destroy) {
    snprintf(r.why, sizeof(r.why), "the loaded librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
    dlclose(r.handle);
    r.handle = nullptr;
  }
}

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;                                           // callable from any host thread
  std::call_once(once, [] { rccl_load(r); });
  return &r;
}

int rccl_fail(const char* what, int rc) {
  Rccl* r = rccl();
  return ld_fail(LD_EHIP, "%s: %s (ncclResult %d)", what, r->errstr ? r->errstr(rc) : "?", rc);
}
}  // namespace

#define LD_RCCL_OR_FAIL(r)                                                      \
  Rccl* r = rccl();                                                             \
  if (!r->handle) return ld_fail(LD_EHIP, "RCCL unavailable: %s", r->why)

extern "C" int ld_comm_unique_id(void* id_out_128) {
  LD_REQUIRE(id_out_128, "ld_comm_unique_id: null");
  LD_RCCL_OR_FAIL(r);
  UniqueId id;
  const int rc = r->get_id(&id);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out_128, &id, sizeof(id));
  return LD_OK;
}

extern "C" int ld_comm_init(void** comm_out, const void* id_128, int world, int rank) {
  LD_REQUIRE(comm_out && id_128 && world >= 1 && rank >= 0 && rank < world, "ld_comm_init: bad arguments (world %d, rank %d)", world, rank);
  LD_RCCL_OR_FAIL(r);
  UniqueId id;
  memcpy(&id, id_128, sizeof(id));
  void* comm = nullptr;
  const int rc = r->init(&comm, world, id, rank);       // uses the calling thread's current HIP device
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  *comm_out = comm;
  return LD_OK;
}

extern "C" int ld_allgather(const void* send, void* recv, size_t bytes_per_rank, void* comm, void* stream) {
  LD_REQUIRE(send && recv && comm && bytes_per_rank > 0, "ld_allgather: bad arguments");
  LD_RCCL_OR_FAIL(r);
  const int rc = r->allgather(send, recv, bytes_per_rank, /*ncclInt8*/ 0, comm, reinterpret_cast<hipStream_t>(stream));
  if (rc != 0) return rccl_fail("ncclAllGather", rc);
  return LD_OK;
}

extern "C" int ld_comm_destroy(void* comm) {
  if (!comm) return LD_OK;
  LD_RCCL_OR_FAIL(r);
  const int rc = r->destroy(comm);
  if (rc != 0) return rccl_fail("ncclCommDestroy", rc);
  return LD_OK;
}
