// The exchange step of the column-sharded fit inside the C ABI: one RCCL communicator per context, the
// all-reduce of the exchange buffer [t~ (n x B) | ||w~_b||^2 | ||p_b||^2 | ...] is enqueued on the context's stream
// right behind the chunk reduction of a pass and right in front of the step kernel that consumes it (resident.hip),
// so a sharded fit needs neither torch nor a host synchronisation per pass.  One process per GPU; the 128-byte
// unique id travels from rank 0 to the others through whatever the launcher offers (bench.py: torch.distributed's
// store).  librccl is loaded at run time (dlopen) -- a single-GPU user never touches it.
#include <dlfcn.h>
#include <rccl/rccl.h>

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
};

struct RcclState {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
};

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
  (void)hipStreamSynchronize(ctx->stream);
  ctx->rccl = nullptr;
  ncclResult_t r = g_rccl.CommDestroy(st->comm);
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
  // in place, float64 sum, ordered on the context's stream: every rank receives bit-identical sums, so every rank
  // takes the same loop-exit decision (the step kernel's stop test, mbpls.py:841)
  MB_NCCL(g_rccl.AllReduce(d_buf, d_buf, count, ncclFloat64, ncclSum, st->comm, ctx->stream));
  return MBPLS_OK;
}

}  // extern "C"
