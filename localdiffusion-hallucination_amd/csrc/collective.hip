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
//
// Round 6: a communicator can be brought up with a DEADLINE (ld_comm_init_timeout).  ncclCommInitRank is collective and
// blocks until every rank has arrived: a stale unique id (a rendezvous file left by a run that died) or a peer that never
// comes is a hang with nothing to report.  With a deadline the communicator is created non-blocking
// (ncclCommInitRankConfig, config.blocking = 0), its state is polled through ncclCommGetAsyncError, and on expiry it is
// torn down with ncclCommAbort and the call returns LD_ETIMEOUT -- an error the caller can act on.  A non-blocking
// communicator's later calls may return ncclInProgress: ld_allgather / ld_comm_destroy poll the same way.
#include "common.hip.h"
#include <rccl/rccl.h>            // TYPES only (ncclConfig_t and its initializer); every function is resolved by dlsym
#include <mutex>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

namespace {
typedef ncclUniqueId UniqueId;                                         // NCCL_UNIQUE_ID_BYTES = 128
static_assert(sizeof(UniqueId) == 128, "the C ABI hands the unique id over as 128 bytes");
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommInitRankConfigFn)(void**, int, UniqueId, int, ncclConfig_t*);
typedef int (*CommGetAsyncErrorFn)(void*, int*);
typedef int (*CommAbortFn)(void*);
typedef int (*CommFinalizeFn)(void*);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*ErrStrFn)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_id = nullptr;
  CommInitRankFn init = nullptr;
  CommInitRankConfigFn init_config = nullptr;      // the three below may be absent in an old RCCL: then there is no deadline
  CommGetAsyncErrorFn async_error = nullptr;
  CommAbortFn abort_ = nullptr;
  CommFinalizeFn finalize = nullptr;
  AllGatherFn allgather = nullptr;
  CommDestroyFn destroy = nullptr;
  ErrStrFn errstr = nullptr;
  char why[256] = "";
};

// What ld_comm_init* hands out: the RCCL communicator plus how it was made.
struct LdCommBox {
  void* comm;
  bool nonblocking;
  double timeout_s;
};

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void rccl_load(Rccl& r) {
  const char* names[4] = {getenv("LD_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  const char* last_err = nullptr;
  if (names[0] && *names[0]) {                                          // an explicit override wins over a copy the process has mapped
    r.handle = dlopen(names[0], RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) last_err = dlerror();
  }
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
  r.init_config = (CommInitRankConfigFn)dlsym(r.handle, "ncclCommInitRankConfig");
  r.async_error = (CommGetAsyncErrorFn)dlsym(r.handle, "ncclCommGetAsyncError");
  r.abort_ = (CommAbortFn)dlsym(r.handle, "ncclCommAbort");
  r.finalize = (CommFinalizeFn)dlsym(r.handle, "ncclCommFinalize");
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

// Poll a non-blocking communicator until its pending operation has finished (ncclSuccess), failed, or `deadline` (absolute,
// seconds on the monotonic clock) has passed.  Returns LD_OK, LD_EHIP (message set) or LD_ETIMEOUT (message set; the
// communicator is NOT touched: the caller decides whether to abort it).
int rccl_wait(Rccl* r, void* comm, double deadline, const char* what) {
  for (;;) {
    int state = ncclSuccess;
    const int rc = r->async_error(comm, &state);
    if (rc != ncclSuccess) return rccl_fail("ncclCommGetAsyncError", rc);
    if (state == ncclSuccess) return LD_OK;
    if (state != ncclInProgress) return rccl_fail(what, state);
    if (now_s() > deadline) return ld_fail(LD_ETIMEOUT, "%s: still in progress at the deadline", what);
    timespec nap = {0, 1000000};                                       // 1 ms
    nanosleep(&nap, nullptr);
  }
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

extern "C" int ld_comm_init_timeout(void** comm_out, const void* id_128, int world, int rank, double timeout_s) {
  LD_REQUIRE(comm_out && id_128 && world >= 1 && rank >= 0 && rank < world, "ld_comm_init: bad arguments (world %d, rank %d)", world, rank);
  LD_RCCL_OR_FAIL(r);
  UniqueId id;
  memcpy(&id, id_128, sizeof(id));
  void* comm = nullptr;
  const bool deadline = timeout_s > 0.0 && r->init_config && r->async_error && r->abort_;
  if (!deadline) {
    const int rc = r->init(&comm, world, id, rank);     // blocking; uses the calling thread's current HIP device
    if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  } else {
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.blocking = 0;
    const int rc = r->init_config(&comm, world, id, rank, &cfg);
    if (rc != ncclSuccess && rc != ncclInProgress) return rccl_fail("ncclCommInitRankConfig", rc);
    const int w = rccl_wait(r, comm, now_s() + timeout_s, "ncclCommInitRankConfig");
    if (w != LD_OK) {
      if (comm) r->abort_(comm);                         // frees the half-built communicator; a dead peer or a stale id is an error now
      if (w == LD_ETIMEOUT)
        return ld_fail(LD_ETIMEOUT, "ld_comm_init: rank %d of %d did not come up within %.1f s (a peer is missing, or the unique id is stale); "
                                    "communicator aborted", rank, world, timeout_s);
      return w;
    }
  }
  LdCommBox* box = new LdCommBox{comm, deadline, timeout_s};
  *comm_out = box;
  return LD_OK;
}

extern "C" int ld_comm_init(void** comm_out, const void* id_128, int world, int rank) {
  return ld_comm_init_timeout(comm_out, id_128, world, rank, 0.0);
}

extern "C" int ld_allgather(const void* send, void* recv, size_t bytes_per_rank, void* comm, void* stream) {
  LD_REQUIRE(send && recv && comm && bytes_per_rank > 0, "ld_allgather: bad arguments");
  LD_RCCL_OR_FAIL(r);
  LdCommBox* box = static_cast<LdCommBox*>(comm);
  const int rc = r->allgather(send, recv, bytes_per_rank, /*ncclInt8*/ 0, box->comm, reinterpret_cast<hipStream_t>(stream));
  if (rc == ncclInProgress && box->nonblocking) return rccl_wait(r, box->comm, now_s() + box->timeout_s, "ncclAllGather");   // (enqueue, not completion)
  if (rc != 0) return rccl_fail("ncclAllGather", rc);
  return LD_OK;
}

extern "C" int ld_comm_destroy(void* comm) {
  if (!comm) return LD_OK;
  LD_RCCL_OR_FAIL(r);
  LdCommBox* box = static_cast<LdCommBox*>(comm);
  int rc = LD_OK;
  if (box->nonblocking && r->finalize) {
    // a non-blocking communicator is finalised first (flushes what was issued); if that does not finish in time it is aborted
    const int f = r->finalize(box->comm);
    int w = (f == ncclSuccess || f == ncclInProgress) ? rccl_wait(r, box->comm, now_s() + box->timeout_s, "ncclCommFinalize") : rccl_fail("ncclCommFinalize", f);
    if (w != LD_OK) { r->abort_(box->comm); delete box; return w; }
  }
  const int d = r->destroy(box->comm);
  if (d != 0 && !(d == ncclInProgress && box->nonblocking)) rc = rccl_fail("ncclCommDestroy", d);
  delete box;
  return rc;
}
