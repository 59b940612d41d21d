// The exchange step of the column-sharded fit inside the C ABI: one RCCL communicator per context, the
// all-reduce of the exchange buffer [t~ (n x B) | ||w~_b||^2 | ||p_b||^2 | ...] is enqueued on the context's stream
// right behind the chunk reduction of a pass and right in front of the step kernel that consumes it (resident.hip),
// so a sharded fit needs neither torch nor a host synchronisation per pass.  One process per GPU; the 128-byte
// unique id travels from rank 0 to the others through whatever the launcher offers (bench.py: torch.distributed's
// store).  librccl is loaded at run time (dlopen) -- a single-GPU user never touches it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>

#include "common.h"

namespace mbpls {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};

struct RcclState {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  bool dead = false;            // aborted after an asynchronous error or a missed deadline: every later call fails
  int timeout_ms = 600000;      // deadline of one wait for the stream (mbpls_hip_comm_set_timeout)
  // fault injection (mbpls_hip_comm_inject_fault; tests at world size 1): the next all-reduce is a kernel that waits for
  // a word in host-mapped memory as a collective waits for a peer that has died; the communicator's asynchronous error
  // turns up after `fault_polls` queries (mode 1) or never (mode 2: the deadline has to catch it); the abort releases it
  int fault_mode = 0, fault_polls = 0, fault_seen = 0;
  unsigned* hang_h = nullptr;   // host-mapped release word of the stand-in kernel
  unsigned* hang_d = nullptr;
  bool hanging = false;
};

// stand-in for a collective whose peer never arrives: spins until the host stores to the release word -- or, should
// nobody abort it, for at most ~8 s of the 100 MHz real-time counter (it must never outlive a test)
__global__ void k_wait_for_peer(volatile unsigned* release) {
  const unsigned long long t0 = wall_clock64();
  while (*release == 0u && wall_clock64() - t0 < 800000000ull) __builtin_amdgcn_s_sleep(64);
}

static RcclApi g_rccl;

static int load_rccl() {
  if (g_rccl.handle) return MBPLS_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    set_error("librccl.so could not be loaded: %s", dlerror());
    return MBPLS_ERR_UNSUPPORTED;
  }
#define MB_SYM(field, name)                                       \
  do {                                                            \
    *(void**)(&g_rccl.field) = dlsym(h, name);                    \
    if (!g_rccl.field) {                                          \
      set_error("librccl.so has no symbol %s", name);             \
      dlclose(h);                                                 \
      return MBPLS_ERR_UNSUPPORTED;                               \
    }                                                             \
  } while (0)
  MB_SYM(GetUniqueId, "ncclGetUniqueId");
  MB_SYM(CommInitRank, "ncclCommInitRank");
  MB_SYM(CommDestroy, "ncclCommDestroy");
  MB_SYM(AllReduce, "ncclAllReduce");
  MB_SYM(GetErrorString, "ncclGetErrorString");
  MB_SYM(GetVersion, "ncclGetVersion");
  MB_SYM(CommCount, "ncclCommCount");
  MB_SYM(CommUserRank, "ncclCommUserRank");
  MB_SYM(CommCuDevice, "ncclCommCuDevice");
  MB_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
  MB_SYM(CommAbort, "ncclCommAbort");
#undef MB_SYM
  g_rccl.handle = h;
  return MBPLS_OK;
}

#define MB_NCCL(call)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r_ = (call);                                                                           \
    if (r_ != ncclSuccess) {                                                                            \
      set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__);         \
      return MBPLS_ERR_HIP;                                                                             \
    }                                                                                                   \
  } while (0)

// Abort the communicator (ncclCommAbort frees the stream of a collective that can no longer complete) and mark it dead.
static void comm_abort(mbpls_ctx* ctx, RcclState* st) {
  if (st->dead) return;
  st->dead = true;
  if (st->hanging && st->hang_h) {          // (the injected stand-in: what the real abort does to a stuck kernel)
    __atomic_store_n(st->hang_h, 1u, __ATOMIC_SEQ_CST);
    st->hanging = false;
  }
  if (st->comm) (void)g_rccl.CommAbort(st->comm);
  st->comm = nullptr;
  (void)hipStreamSynchronize(ctx->stream);   // released now; what was queued behind the collective runs on garbage and ends
}

int comm_guard_wait(mbpls_ctx* ctx) {
  RcclState* st = (RcclState*)ctx->rccl;
  // (aborted earlier: the stream was drained then and no collective can be queued any more -- plain waits are safe, and
  // the clean-up of the failed fit must be able to free its buffers)
  if (!st || st->dead) return MBPLS_OK;
  if (st->world <= 1 && !st->fault_mode) return MBPLS_OK;   // nothing on the stream depends on another process
  using clock = std::chrono::steady_clock;
  const auto t0 = clock::now();
  auto next_check = t0 + std::chrono::microseconds(500);
  for (unsigned spins = 0;; ++spins) {
    const hipError_t q = hipStreamQuery(ctx->stream);
    if (q == hipSuccess) return MBPLS_OK;
    if (q != hipErrorNotReady) {
      set_error("hipStreamQuery failed while waiting for the compute stream: %s", hipGetErrorString(q));
      return MBPLS_ERR_HIP;
    }
    if (spins > 2000) usleep(50);             // ~ the first 200 us busy (a pass of a sharded fit is shorter than that)
    const auto now = clock::now();
    if (now < next_check) continue;
    next_check = now + std::chrono::milliseconds(1);
    ncclResult_t async = ncclSuccess;
    if (st->fault_mode == 1 && st->hanging && ++st->fault_seen > st->fault_polls) async = ncclSystemError;
    else if (st->comm && !st->fault_mode) (void)g_rccl.CommGetAsyncError(st->comm, &async);
    const double waited_ms = std::chrono::duration<double, std::milli>(now - t0).count();
    if (async != ncclSuccess && async != ncclInProgress) {
      comm_abort(ctx, st);
      set_error("RCCL reports an asynchronous error on rank %d of %d (%s) after %.0f ms of waiting for the compute "
                "stream: a peer has gone; the communicator was aborted", st->rank, st->world,
                g_rccl.GetErrorString(async), waited_ms);
      return MBPLS_ERR_COMM;
    }
    if (waited_ms > (double)st->timeout_ms) {
      comm_abort(ctx, st);
      set_error("rank %d of %d waited %.0f ms for the compute stream (deadline %d ms, mbpls_hip_comm_set_timeout) with a "
                "collective in flight: a peer is not taking part; the communicator was aborted", st->rank, st->world,
                waited_ms, st->timeout_ms);
      return MBPLS_ERR_COMM;
    }
  }
}

}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_comm_unique_id(void* id_out) {
  MB_ARG(id_out, "id_out is NULL");
  static_assert(sizeof(ncclUniqueId) == MBPLS_COMM_ID_BYTES, "unique id size");
  MB_TRY(load_rccl());
  ncclUniqueId id;
  MB_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return MBPLS_OK;
}

int mbpls_hip_comm_init_rank(mbpls_ctx* ctx, int world, int rank, const void* id_in) {
  MB_ARG(ctx && id_in, "NULL argument");
  MB_ARG(world >= 1 && rank >= 0 && rank < world, "bad rank %d of %d", rank, world);
  MB_ARG(ctx->rccl == nullptr, "this context already has a communicator");
  MB_TRY(load_rccl());
  MB_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  RcclState* st = new RcclState();
  ncclResult_t r = g_rccl.CommInitRank(&st->comm, world, id, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(world=%d, rank=%d) failed: %s", world, rank, g_rccl.GetErrorString(r));
    delete st;
    return MBPLS_ERR_HIP;
  }
  st->world = world;
  st->rank = rank;
  ctx->rccl = st;
  return MBPLS_OK;
}

int mbpls_hip_comm_info(mbpls_ctx* ctx, int* world, int* rank, int* rccl_version) {
  MB_ARG(ctx, "ctx is NULL");
  RcclState* st = (RcclState*)ctx->rccl;
  if (world) *world = st ? st->world : 1;
  if (rank) *rank = st ? st->rank : 0;
  if (rccl_version) {
    *rccl_version = 0;
    if (g_rccl.handle) (void)g_rccl.GetVersion(rccl_version);
  }
  return MBPLS_OK;
}

int mbpls_hip_comm_query(mbpls_ctx* ctx, int* rccl_ranks, int* rccl_user_rank, int* rccl_device) {
  MB_ARG(ctx, "ctx is NULL");
  RcclState* st = (RcclState*)ctx->rccl;
  MB_ARG(st != nullptr, "no communicator on this context (mbpls_hip_comm_init_rank)");
  // what the communicator itself reports -- not what the caller passed to mbpls_hip_comm_init_rank
  int v = 0;
  if (rccl_ranks) {
    MB_NCCL(g_rccl.CommCount(st->comm, &v));
    *rccl_ranks = v;
  }
  if (rccl_user_rank) {
    MB_NCCL(g_rccl.CommUserRank(st->comm, &v));
    *rccl_user_rank = v;
  }
  if (rccl_device) {
    MB_NCCL(g_rccl.CommCuDevice(st->comm, &v));
    *rccl_device = v;
  }
  return MBPLS_OK;
}

int mbpls_hip_comm_destroy(mbpls_ctx* ctx) {
  MB_ARG(ctx, "ctx is NULL");
  RcclState* st = (RcclState*)ctx->rccl;
  if (!st) return MBPLS_OK;
  if (st->hanging && st->hang_h) __atomic_store_n(st->hang_h, 1u, __ATOMIC_SEQ_CST);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->rccl = nullptr;
  ncclResult_t r = st->comm ? g_rccl.CommDestroy(st->comm) : ncclSuccess;     // (aborted: nothing left to destroy)
  if (st->hang_h) (void)hipHostFree(st->hang_h);
  delete st;
  if (r != ncclSuccess) {
    set_error("ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
    return MBPLS_ERR_HIP;
  }
  return MBPLS_OK;
}

int mbpls_hip_allreduce(mbpls_ctx* ctx, double* d_buf, size_t count) {
  MB_ARG(ctx && (count == 0 || d_buf), "NULL argument");
  RcclState* st = (RcclState*)ctx->rccl;
  MB_ARG(st != nullptr, "no communicator on this context (mbpls_hip_comm_init_rank)");
  if (count == 0) return MBPLS_OK;
  if (st->dead) {
    set_error("the communicator of this context was aborted earlier (dead peer or missed deadline); create a new context");
    return MBPLS_ERR_COMM;
  }
  if (st->fault_mode && !st->hanging) {
    // injected fault: this collective never completes by itself
    if (!st->hang_h) {
      MB_HIP(hipHostMalloc((void**)&st->hang_h, 64, hipHostMallocMapped | hipHostMallocCoherent));
      MB_HIP(hipHostGetDevicePointer((void**)&st->hang_d, st->hang_h, 0));
    }
    __atomic_store_n(st->hang_h, 0u, __ATOMIC_SEQ_CST);
    st->hanging = true;
    st->fault_seen = 0;
    hipLaunchKernelGGL(k_wait_for_peer, dim3(1), dim3(64), 0, ctx->stream, (volatile unsigned*)st->hang_d);
    MB_HIP(hipGetLastError());
    return MBPLS_OK;
  }
  // in place, float64 sum, ordered on the context's stream: every rank receives bit-identical sums, so every rank
  // takes the same loop-exit decision (the step kernel's stop test, mbpls.py:841)
  MB_NCCL(g_rccl.AllReduce(d_buf, d_buf, count, ncclFloat64, ncclSum, st->comm, ctx->stream));
  return MBPLS_OK;
}

int mbpls_hip_comm_set_timeout(mbpls_ctx* ctx, int timeout_ms) {
  MB_ARG(ctx, "ctx is NULL");
  RcclState* st = (RcclState*)ctx->rccl;
  MB_ARG(st != nullptr, "no communicator on this context (mbpls_hip_comm_init_rank)");
  MB_ARG(timeout_ms >= 1, "timeout must be >= 1 ms");
  st->timeout_ms = timeout_ms;
  return MBPLS_OK;
}

int mbpls_hip_comm_inject_fault(mbpls_ctx* ctx, int mode, int polls) {
  MB_ARG(ctx, "ctx is NULL");
  RcclState* st = (RcclState*)ctx->rccl;
  MB_ARG(st != nullptr, "no communicator on this context (mbpls_hip_comm_init_rank)");
  MB_ARG(mode >= 0 && mode <= 2 && polls >= 0, "mode 0 (off), 1 (asynchronous error after `polls` queries) or 2 (silent peer)");
  st->fault_mode = mode;
  st->fault_polls = polls;
  return MBPLS_OK;
}

}  // extern "C"
