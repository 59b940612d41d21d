// Internal definitions shared by the HIP translation units of libmbpls_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <cmath>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mbpls_hip.h"

namespace mbpls {

void set_error(const char* fmt, ...);
}  // namespace mbpls
struct mbpls_ctx;
namespace mbpls {
int comm_guard_wait(mbpls_ctx* ctx);

#define MB_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      ::mbpls::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                       \
      /* the runtime remembers the error: left there, the hipGetLastError() behind the NEXT launch of a healthy call */ \
      /* would report it again (an allocation that did not fit made the following fit fail with "out of memory") */     \
      (void)hipGetLastError();                                                            \
      return e_ == hipErrorOutOfMemory ? MBPLS_ERR_OOM : MBPLS_ERR_HIP;                   \
    }                                                                                     \
  } while (0)

#define MB_ARG(cond, ...)              \
  do {                                 \
    if (!(cond)) {                     \
      ::mbpls::set_error(__VA_ARGS__); \
      return MBPLS_ERR_ARG;            \
    }                                  \
  } while (0)

// More than 64 KB of dynamic LDS has to be allowed per kernel AND per device (hipFuncSetAttribute acts on the current
// device's copy of the code object): one bit per device ordinal remembers where it has been done.
#define MB_ALLOW_LDS(ctx_, fn_, bytes_)                                                                            \
  do {                                                                                                             \
    static std::atomic<uint64_t> done_{0};                                                                         \
    const uint64_t bit_ = 1ull << ((unsigned)(ctx_)->device & 63u);                                                \
    if (!(done_.load(std::memory_order_relaxed) & bit_) || (ctx_)->device >= 64) {                                 \
      MB_HIP(hipFuncSetAttribute((const void*)(fn_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes_)));  \
      done_.fetch_or(bit_, std::memory_order_relaxed);                                                             \
    }                                                                                                              \
  } while (0)

#define MB_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != MBPLS_OK) return rc_; \
  } while (0)

// Every wait of the host for the compute stream goes through this: with a communicator of more than one rank on the
// context a collective queued on the stream only completes when every peer takes part -- a dead peer would leave
// hipStreamSynchronize waiting for ever.  comm_guard_wait (rccl_comm.hip) polls the stream instead, asks the
// communicator for its asynchronous error state and enforces a deadline; on either it aborts the communicator (which
// releases the stream) and the call returns MBPLS_ERR_COMM.  Without such a communicator it returns at once.
#define MB_SYNC_STREAM(ctx_)                               \
  do {                                                     \
    if ((ctx_)->rccl) MB_TRY(::mbpls::comm_guard_wait(ctx_)); \
    MB_HIP(hipStreamSynchronize((ctx_)->stream));          \
  } while (0)

// One chunk table: contiguous column ranges that never straddle a block.
struct ChunkTable {
  int nchunks = 0;
  std::vector<int64_t> h_c0, h_c1;   // [nchunks]
  std::vector<int> h_blk;            // [nchunks]
  std::vector<int> h_blk_start;      // [nblocks+1] chunk range of each block
  int64_t* d_c0 = nullptr;
  int64_t* d_c1 = nullptr;
  int* d_blk_start = nullptr;
};

}  // namespace mbpls

constexpr int PIN_SLOTS = 4;
constexpr int SMALL_PIN_SLOTS = 64;
constexpr size_t SMALL_PIN_BYTES = 256u << 10;   // (what went to the runtime's own staging before: copies below 256 KB)

// A queued column of R_ = W~ C^-1 (mbpls_hip_queue_r_update): taken along by the next chunk reduction as extra workgroups
struct RUpdate {
  int64_t p = 0, ldr = 0;
  const double* w = nullptr;
  double* R = nullptr;
  const double* Ccol = nullptr;
  const double* vk = nullptr;
  double* beta = nullptr;
  int k = -1;              // -1: nothing queued
};

struct mbpls_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int cus = 256;
  // tunables
  int chunks_per_cu = 4;
  int chunks_adaptive = 1;   // (cleared when chunks_per_cu is set by hand) narrow matrices get half as many chunks, see mat_create
  int rows_per_thread = 0;   // 0 = auto
  int prefetch = 2;          // columns in flight per workgroup (register slots)
  int xw_chunks_per_cu = 1;
  int xw_mfma_from = 8;      // x_w with more right-hand sides than this runs on the matrix cores
  int64_t gram_slice_features = 131072;   // Gram: features per split-K slice for long feature ranges (0 = off)
  int sub_columns = 1;       // n <= 256: several columns per wave (colpass_sub.hip); 0 = the one-column-per-workgroup kernels
  int64_t long_columns_from = 8192;    // n above this: streamed-column kernels (colpass_long.hip)
  int64_t nan_long_from = 8192;        // missing-value passes: n above this streams the columns (colpass_nan.hip)
  int nan_long_two_phase = 1;          // ... as two streaming phases with rows as a parallel dimension (colpass_nan_long.hip); 0: one workgroup per chunk re-visits its columns
  // ... and among those, the never-writes-X passes run on CLUSTERS of workgroups that share a column and read X once
  // (colpass_cluster.hip); 0 = the two-phase passes of colpass_long.hip for everything
  int cluster_passes = 1;
  int64_t cluster_slab_rows = 8192;    // rows of a column held by one workgroup of a cluster (tests lower it)
  int cluster_prefetch = 0;            // shared exchange wave: loads of the next column's partial sums issued ahead
  int cluster_exchange = 0;            // 0: dedicated exchange wave unless sharing saves workgroups; 1: always; 2: never
  unsigned cluster_spin_limit = 1u << 22;   // polls before a cluster hand-off gives up
  int cluster_cap = -1;                // workgroups of a cluster pass the device holds at once (-1: not asked yet; 0: none)
  void* cl_ws = nullptr;               // fine-grained: [error word | exchange slots] of the cluster passes
  unsigned* cl_flag_h = nullptr;       // host-mapped word a pass sets when a hand-off gave up (read by the host without a copy)
  size_t cl_ws_bytes = 0;
  // workspace for chunk partials (grown on demand, never shrunk)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // staging for transposing uploads
  void* stage = nullptr;
  size_t stage_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t copy_stream = nullptr;   // non-blocking: D2H of finished results while kernels run
  hipEvent_t fence_ev = nullptr;       // mbpls_hip_fence: what was queued before it is finished before the next overlapped copy reads
  // ring of small K x K matrices for right_multiply
  void* small_m = nullptr;
  unsigned small_m_slot = 0;
  // device-resident pass parameters (mbpls_hip_set_pass_params): [0] alpha, [1] inv_tt, [2] skip flag -- the
  // column passes read them from HBM instead of taking host scalars, so a whole fit can be queued without a host
  // round trip (resident.hip writes them between the passes)
  const double* dpar = nullptr;
  // optional per-column variance output of the standardise passes (mbpls_hip_set_variance_output): sklearn's
  // StandardScaler.var_ (0 for a constant column, whose scale_ is 1)
  double* var_out = nullptr;
  // true-space outputs of the raw-space passes (mbpls_hip_set_true_space_outputs): 1/scale_j and the destination of
  // the true-space loadings of the next loadings pass
  const double* ts_rscale = nullptr;
  double* ts_p_true = nullptr;
  void* pin_d2h = nullptr;             // pinned bounce buffer of the small blocking device-to-host copies
  // pinned bounce buffers of the large host-to-device copies (a ring of slots, one event each): a pageable
  // hipMemcpyAsync of a few MB sporadically takes 8-25 ms inside the runtime right after a fit
  void* pin = nullptr;
  size_t pin_half = 0;     // bytes per slot
  hipEvent_t pin_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned pin_next = 0;
  void* pin_small = nullptr;           // ... and of the small ones (vectors, parameter blocks): SMALL_PIN_SLOTS x 256 KB
  hipEvent_t pin_small_ev[SMALL_PIN_SLOTS];
  unsigned pin_small_next = 0;
  std::mutex pin_mutex;
  std::vector<void*> host_blocks;      // mbpls_hip_host_alloc'd memory not yet freed (released by mbpls_hip_destroy AFTER the copier thread, which may be polling a flag in it, has stopped)
  void* copier = nullptr;    // CopierState* (copier.hip): the thread that carries finished columns to the host
  RUpdate r_update;          // queued by mbpls_hip_queue_r_update, consumed by reduce_chunks / mbpls_hip_flush_r_update
  int step_reg = 1;          // device-resident step: the register-resident kernel where it applies (step_reg.hip); 0 = general kernel
  unsigned step_spin_limit = 1u << 28;   // polls before a grid barrier of the multi-workgroup step gives up (tests lower it)
  void* step_ws = nullptr;   // grid barrier + per-workgroup totals of the multi-workgroup step kernel (resident.hip)
  void* rccl = nullptr;      // RcclState* (rccl_comm.hip) when a communicator was initialised on this context
  // events of the ordered D2H copies are recycled (created by the queueing thread, returned by the copier thread)
  std::mutex user_event_mutex;
  std::vector<hipEvent_t> user_event_pool;
  // per-kernel profile (HIP events around the main kernel of every hot-path call)
  bool prof = false;
  struct Pending { hipEvent_t a, b; int kind; double bytes; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
  int64_t prof_count[MBPLS_K_COUNT] = {0};
  double prof_ms[MBPLS_K_COUNT] = {0};
  double prof_bytes[MBPLS_K_COUNT] = {0};
};

namespace mbpls {
void copier_shutdown(mbpls_ctx* ctx);   // copier.hip
int copier_warmup(mbpls_ctx* ctx);
}  // namespace mbpls

struct mbpls_mat {
  mbpls_ctx* ctx = nullptr;
  int64_t n = 0, p = 0, ld = 0;
  int nblocks = 1;
  std::vector<int64_t> block_offsets;
  int64_t* d_block_offsets = nullptr;   // the same on the device (nblocks + 1), uploaded once at creation
  double* data = nullptr;
  bool owns = false;
  mbpls::ChunkTable fine;    // column-pass kernels (one workgroup per chunk, full columns)
  mbpls::ChunkTable coarse;  // row-slab kernels (x_w)
  mbpls::ChunkTable cluster; // cluster passes (colpass_cluster.hip): one chunk per cluster of workgroups, built on demand
  int cluster_G = 0;         // ... for this many clusters
};

namespace mbpls {
int ensure_ws(mbpls_ctx* ctx, size_t bytes);
int ensure_stage(mbpls_ctx* ctx, size_t bytes);
int build_chunks(mbpls_ctx* ctx, const std::vector<int64_t>& offs, int target, ChunkTable* t);
void free_chunks(ChunkTable* t);
// xtv_mfma.hip: G = alpha X^T V on the f64 matrix cores (any n, any c)
int xtv_mfma(mbpls_ctx* ctx, const mbpls_mat* m, const double* d_V, int64_t ldv, int c, double alpha, double* d_G,
             int64_t ldg, double* d_ssq, const double* d_mean = nullptr, const double* d_rscale = nullptr, int square = 0);

// xw_mfma.hip: OUT = X W on the f64 matrix cores, X read once per 64 right-hand sides
int xw_mfma(mbpls_ctx* ctx, const mbpls_mat* m, const double* d_W, int64_t ldw, int c, int per_block,
            const double* d_mean, const double* d_scale, int nanm, double* d_out, int64_t ldo);

// axpy_cols.hip: OUT = X g for one right-hand side as a full-column streaming pass (MBPLS_ERR_UNSUPPORTED when the
// shape is not handled there and the caller should use the row-slab kernel)
int axpy_cols(mbpls_ctx* ctx, const mbpls_mat* m, const double* d_g, int per_block, double* d_out);

// step_reg.hip: the device-resident step with a thread's rows in registers (one response, <= 4 blocks, n <= 4096);
// MBPLS_ERR_UNSUPPORTED for other shapes (resident.hip has the general kernel)
int launch_step_reg(mbpls_ctx* ctx, const mbpls_nipals_step* s);

// RAII bracket: records an event pair around a kernel launch when profiling is on.
struct ProfScope {
  mbpls_ctx* ctx;
  int kind;
  double bytes;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(mbpls_ctx* c, int k, double by);
  ~ProfScope();
};

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__
// One DPP lane exchange of a double (two v_mov_b32_dpp: the data-parallel-primitive path runs on the VALU, unlike
// __shfl_xor, which goes through the LDS crossbar as ds_bpermute_b32 pairs -- 36 LDS instructions per wave for
// three sums, a measurable part of a column's critical path).  Lanes outside ROW_MASK receive 0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  constexpr bool ALL = ROW_MASK == 0xf;   // every lane has a source: no `old` value to materialise
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, ALL);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, ALL);
  return __hiloint2double(hi, lo);
}

// 64-lane sum in a fixed order; the total is valid in lanes 48..63 (the last row) only.
__device__ __forceinline__ double wave_sum_last_row(double v) {
  v += dpp_move<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]: pairs
  v += dpp_move<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]: quads
  v += dpp_move<0x141, 0xf>(v);   // row_half_mirror: 8 lanes
  v += dpp_move<0x140, 0xf>(v);   // row_mirror: the 16-lane row
  v += dpp_move<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3: 32 lanes
  v += dpp_move<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3: row 3 holds all 64
  return v;
}

// Every lane ends with the same, order-fixed sum.
__device__ __forceinline__ double wave_sum(double v) {
  v = wave_sum_last_row(v);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// Sum NV values over the whole workgroup; every thread receives the same result.
// red: LDS scratch double[2][16][4]; buf alternates so one barrier per call suffices.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red, int& buf) {
  const int nw = blockDim.x >> 6;
  if (nw == 1) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    return;
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum_last_row(v[k]);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double* r = red + buf * 64;
  if (lane == 63) {               // the wave total lives in the last DPP row
#pragma unroll
    for (int k = 0; k < NV; ++k) r[w * 4 + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double s = r[k];
    for (int i = 1; i < nw; ++i) s += r[i * 4 + k];
    v[k] = s;
  }
  buf ^= 1;
}

// Sum the first m of M <= 16 values over a 1024-thread workgroup: wave butterflies, the 16 wave partials through a
// few hundred bytes of LDS, thread k < m adds them in wave order, everyone reads the totals back.  Fixed order
// (deterministic), two barriers per call whatever m is.  (The first version parked every thread's partials in LDS --
// 128 KB of traffic per 16 values -- and cost 5-7 us per call; the step kernels make seven to ten calls.)
// part: at least 16*M doubles, tot: at least M doubles.
constexpr int WG1024 = 1024;
template <int M>
__device__ __forceinline__ void reduce_many(double (&v)[M], int m, double* part, double* tot) {
  static_assert(M <= 16, "tot[] of the callers holds 16 values");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int k = 0; k < M; ++k)
    if (k < m) v[k] = wave_sum_last_row(v[k]);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < M; ++k)
      if (k < m) part[w * M + k] = v[k];
  }
  __syncthreads();
  if (tid < m) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < WG1024 / 64; ++i) s += part[i * M + tid];
    tot[tid] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < M; ++k)
    if (k < m) v[k] = tot[k];
}
#endif

}  // namespace mbpls
