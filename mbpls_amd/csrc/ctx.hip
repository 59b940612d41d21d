// Context, device memory, matrices (feature-major X in HBM), transposing upload/download.
#include <algorithm>
#include <cmath>

#include "common.h"

namespace mbpls {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int ensure_ws(mbpls_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return MBPLS_OK;
  if (ctx->ws) {
    MB_SYNC_STREAM(ctx);
    MB_HIP(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  size_t want = std::max(bytes, (size_t)1 << 20);
  hipError_t e = hipMalloc(&ctx->ws, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ctx->ws = nullptr;
    size_t f = 0, t = 0;
    (void)hipMemGetInfo(&f, &t);
    set_error("workspace of %.2f GB (chunk partials of a pass) does not fit the device: %.2f GB free of %.2f GB (%s)",
              (double)want / 1e9, (double)f / 1e9, (double)t / 1e9, hipGetErrorString(e));
    return MBPLS_ERR_OOM;
  }
  ctx->ws_bytes = want;
  return MBPLS_OK;
}

int ensure_stage(mbpls_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->stage_bytes) return MBPLS_OK;
  if (ctx->stage) {
    MB_SYNC_STREAM(ctx);
    MB_HIP(hipFree(ctx->stage));
    ctx->stage = nullptr;
    ctx->stage_bytes = 0;
  }
  hipError_t e = hipMalloc(&ctx->stage, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ctx->stage = nullptr;
    size_t f = 0, t = 0;
    (void)hipMemGetInfo(&f, &t);
    set_error("scratch of %.2f GB (staging / per-slab partials of a pass) does not fit the device: %.2f GB free of %.2f GB (%s)",
              (double)bytes / 1e9, (double)f / 1e9, (double)t / 1e9, hipGetErrorString(e));
    return MBPLS_ERR_OOM;
  }
  ctx->stage_bytes = bytes;
  return MBPLS_OK;
}

static hipEvent_t take_event(mbpls_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

ProfScope::ProfScope(mbpls_ctx* c, int k, double by) : ctx(c), kind(k), bytes(by) {
  if (!ctx->prof) return;
  a = take_event(ctx);
  b = take_event(ctx);
  if (a) (void)hipEventRecord(a, ctx->stream);
}

ProfScope::~ProfScope() {
  if (!ctx->prof || !a || !b) return;
  (void)hipEventRecord(b, ctx->stream);
  ctx->pending.push_back({a, b, kind, bytes});
}

static int drain_profile(mbpls_ctx* ctx) {
  if (ctx->pending.empty()) return MBPLS_OK;
  MB_SYNC_STREAM(ctx);
  for (auto& pd : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pd.a, pd.b) == hipSuccess) {
      ctx->prof_count[pd.kind] += 1;
      ctx->prof_ms[pd.kind] += (double)ms;
      ctx->prof_bytes[pd.kind] += pd.bytes;
    }
    ctx->event_pool.push_back(pd.a);
    ctx->event_pool.push_back(pd.b);
  }
  ctx->pending.clear();
  return MBPLS_OK;
}

void free_chunks(ChunkTable* t) {
  if (t->d_c0) (void)hipFree(t->d_c0);
  if (t->d_c1) (void)hipFree(t->d_c1);
  if (t->d_blk_start) (void)hipFree(t->d_blk_start);
  *t = ChunkTable();
}

// Split every block into ~target * p_b / p contiguous chunks of near-equal width.
int build_chunks(mbpls_ctx* ctx, const std::vector<int64_t>& offs, int target, ChunkTable* t) {
  const int B = (int)offs.size() - 1;
  const int64_t p = offs[B];
  t->h_c0.clear(); t->h_c1.clear(); t->h_blk.clear();
  t->h_blk_start.assign(B + 1, 0);
  for (int b = 0; b < B; ++b) {
    const int64_t pb = offs[b + 1] - offs[b];
    t->h_blk_start[b] = (int)t->h_c0.size();
    if (pb == 0) continue;
    int64_t nb = (int64_t)std::llround((double)target * (double)pb / (double)std::max<int64_t>(p, 1));
    nb = std::max<int64_t>(1, std::min<int64_t>(nb, pb));
    const int64_t base = pb / nb, rem = pb % nb;
    int64_t c = offs[b];
    for (int64_t k = 0; k < nb; ++k) {
      const int64_t w = base + (k < rem ? 1 : 0);
      t->h_c0.push_back(c);
      t->h_c1.push_back(c + w);
      t->h_blk.push_back(b);
      c += w;
    }
  }
  t->h_blk_start[B] = (int)t->h_c0.size();
  t->nchunks = (int)t->h_c0.size();
  if (t->nchunks == 0) return MBPLS_OK;
  MB_HIP(hipMalloc(&t->d_c0, sizeof(int64_t) * t->nchunks));
  MB_HIP(hipMalloc(&t->d_c1, sizeof(int64_t) * t->nchunks));
  MB_HIP(hipMalloc(&t->d_blk_start, sizeof(int) * (B + 1)));
  MB_HIP(hipMemcpyAsync(t->d_c0, t->h_c0.data(), sizeof(int64_t) * t->nchunks, hipMemcpyHostToDevice, ctx->stream));
  MB_HIP(hipMemcpyAsync(t->d_c1, t->h_c1.data(), sizeof(int64_t) * t->nchunks, hipMemcpyHostToDevice, ctx->stream));
  MB_HIP(hipMemcpyAsync(t->d_blk_start, t->h_blk_start.data(), sizeof(int) * (B + 1), hipMemcpyHostToDevice, ctx->stream));
  MB_SYNC_STREAM(ctx);  // host vectors may be reused by the caller
  return MBPLS_OK;
}

// ---- tiled transposes between a row-major staging tile [n][w] and feature-major [w][ld] ----
constexpr int TT = 32;

// stage[i*w + j] (i<n, j<w)  ->  dst[j*ld + i]
__global__ __launch_bounds__(256) void k_transpose_in(const double* __restrict__ stage, int64_t n, int64_t w,
                                                      double* __restrict__ dst, int64_t ld) {
  __shared__ double tile[TT][TT + 1];
  const int64_t j0 = (int64_t)blockIdx.x * TT, i0 = (int64_t)blockIdx.y * TT;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ty; r < TT; r += 8) {
    const int64_t i = i0 + r, j = j0 + tx;
    tile[r][tx] = (i < n && j < w) ? stage[i * w + j] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < TT; r += 8) {
    const int64_t j = j0 + r, i = i0 + tx;
    if (j < w && i < ld) dst[j * ld + i] = (i < n) ? tile[tx][r] : 0.0;
  }
}


// dst[i*ldd + j] = src[j*lds + i] (i < n, j < p): the feature-major matrix as its own transpose (rows beyond p up to ldd: 0)
__global__ __launch_bounds__(256) void k_transpose_mat(const double* __restrict__ src, int64_t lds, int64_t n, int64_t p,
                                                       double* __restrict__ dst, int64_t ldd) {
  __shared__ double tile[TT][TT + 1];
  const int64_t i0 = (int64_t)blockIdx.x * TT, j0 = (int64_t)blockIdx.y * TT;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ty; r < TT; r += 8) {
    const int64_t j = j0 + r, i = i0 + tx;
    tile[r][tx] = (i < n && j < p) ? src[j * lds + i] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < TT; r += 8) {
    const int64_t i = i0 + r, j = j0 + tx;
    if (i < n && j < ldd) dst[i * ldd + j] = (j < p) ? tile[tx][r] : 0.0;
  }
}

// src[j*ld + i]  ->  stage[i*w + j]
__global__ __launch_bounds__(256) void k_transpose_out(const double* __restrict__ src, int64_t ld, int64_t n,
                                                       int64_t w, double* __restrict__ stage) {
  __shared__ double tile[TT][TT + 1];
  const int64_t j0 = (int64_t)blockIdx.x * TT, i0 = (int64_t)blockIdx.y * TT;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < TT; r += 8) {
    const int64_t j = j0 + r, i = i0 + tx;
    tile[r][tx] = (j < w && i < n) ? src[j * ld + i] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < TT; r += 8) {
    const int64_t i = i0 + r, j = j0 + tx;
    if (i < n && j < w) stage[i * w + j] = tile[tx][r];
  }
}

// zero the padding rows [n, ld) of columns [c0, c0+w)
__global__ void k_zero_pad(double* __restrict__ X, int64_t ld, int64_t n, int64_t c0, int64_t w) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < w)
    for (int64_t i = n; i < ld; ++i) X[(c0 + j) * ld + i] = 0.0;
}

}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_abi_version(void) { return MBPLS_HIP_ABI_VERSION; }

const char* mbpls_hip_last_error(void) { return g_err.c_str(); }

int mbpls_hip_device_count(int* count) {
  MB_ARG(count != nullptr, "count is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    c = 0;
  }
  *count = c;
  return MBPLS_OK;
}

int mbpls_hip_create(int device, void* stream, mbpls_ctx** out) {
  MB_ARG(out != nullptr, "out is NULL");
  *out = nullptr;
  int cnt = 0;
  MB_HIP(hipGetDeviceCount(&cnt));
  MB_ARG(device >= 0 && device < cnt, "device %d out of range (%d visible)", device, cnt);
  MB_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  MB_HIP(hipGetDeviceProperties(&prop, device));
  mbpls_ctx* ctx = new mbpls_ctx();
  ctx->device = device;
  ctx->stream = (hipStream_t)stream;
  ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("MBPLS_HIP_CHUNKS_PER_CU")) {
    ctx->chunks_per_cu = std::max(1, atoi(e));
    ctx->chunks_adaptive = 0;
  }
  if (const char* e = getenv("MBPLS_HIP_ROWS_PER_THREAD")) ctx->rows_per_thread = atoi(e);
  if (const char* e = getenv("MBPLS_HIP_PREFETCH")) ctx->prefetch = atoi(e);
  if (const char* e = getenv("MBPLS_HIP_XW_CHUNKS_PER_CU")) ctx->xw_chunks_per_cu = std::max(1, atoi(e));
  if (const char* e = getenv("MBPLS_HIP_CLUSTER_PASSES")) ctx->cluster_passes = atoi(e) ? 1 : 0;
  hipError_t e0 = hipEventCreate(&ctx->ev0), e1 = hipEventCreate(&ctx->ev1);
  // the copy stream and the fence event are used from helper threads (mbpls_hip_memcpy_d2h_overlap) beside
  // mbpls_hip_fence on the fitting thread: created here, never lazily
  hipError_t e2 = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
  hipError_t e3 = hipEventCreateWithFlags(&ctx->fence_ev, hipEventDisableTiming);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
    set_error("hipEventCreate / hipStreamCreate failed");
    return MBPLS_ERR_HIP;
  }
  *out = ctx;
  return MBPLS_OK;
}

int mbpls_hip_destroy(mbpls_ctx* ctx) {
  if (!ctx) return MBPLS_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  mbpls::copier_shutdown(ctx);
  for (void* hb : ctx->host_blocks) (void)hipHostFree(hb);
  ctx->host_blocks.clear();
  if (ctx->rccl) (void)mbpls_hip_comm_destroy(ctx);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->stage) (void)hipFree(ctx->stage);
  if (ctx->small_m) (void)hipFree(ctx->small_m);
  if (ctx->step_ws) (void)hipFree(ctx->step_ws);
  if (ctx->cl_ws) (void)hipFree(ctx->cl_ws);
  if (ctx->cl_flag_h) (void)hipHostFree(ctx->cl_flag_h);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  for (hipEvent_t e : ctx->pin_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->fence_ev) (void)hipEventDestroy(ctx->fence_ev);
  if (ctx->pin_d2h) (void)hipHostFree(ctx->pin_d2h);
  if (ctx->pin_small) {
    for (hipEvent_t e : ctx->pin_small_ev)
      if (e) (void)hipEventDestroy(e);
    (void)hipHostFree(ctx->pin_small);
  }
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (auto& pd : ctx->pending) {
    (void)hipEventDestroy(pd.a);
    (void)hipEventDestroy(pd.b);
  }
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->user_event_pool) (void)hipEventDestroy(e);
  delete ctx;
  return MBPLS_OK;
}

int mbpls_hip_sync(mbpls_ctx* ctx) {
  MB_ARG(ctx, "ctx is NULL");
  MB_SYNC_STREAM(ctx);
  return MBPLS_OK;
}

int mbpls_hip_device_info(mbpls_ctx* ctx, char* name, size_t name_len, int* compute_units,
                          size_t* total_bytes, size_t* free_bytes) {
  MB_ARG(ctx, "ctx is NULL");
  hipDeviceProp_t prop;
  MB_HIP(hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_len) {
    std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    strncpy(name, s.c_str(), name_len - 1);
    name[name_len - 1] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  size_t f = 0, t = 0;
  MB_HIP(hipSetDevice(ctx->device));
  MB_HIP(hipMemGetInfo(&f, &t));
  if (total_bytes) *total_bytes = t;
  if (free_bytes) *free_bytes = f;
  return MBPLS_OK;
}

int mbpls_hip_device_pci_bus_id(mbpls_ctx* ctx, char* out, size_t out_len) {
  MB_ARG(ctx && out && out_len >= 16, "need a buffer of at least 16 bytes");
  MB_HIP(hipDeviceGetPCIBusId(out, (int)out_len, ctx->device));
  return MBPLS_OK;
}

int mbpls_hip_set_tunable(mbpls_ctx* ctx, const char* key, int value) {
  MB_ARG(ctx && key, "NULL argument");
  if (!strcmp(key, "chunks_per_cu")) {
    ctx->chunks_per_cu = std::max(1, value);
    ctx->chunks_adaptive = 0;
  }
  else if (!strcmp(key, "rows_per_thread")) ctx->rows_per_thread = value;
  else if (!strcmp(key, "xw_chunks_per_cu")) ctx->xw_chunks_per_cu = std::max(1, value);
  else if (!strcmp(key, "prefetch")) ctx->prefetch = value;
  else if (!strcmp(key, "xw_mfma_from")) ctx->xw_mfma_from = std::max(0, value);
  else if (!strcmp(key, "gram_slice_features")) ctx->gram_slice_features = std::max(0, value);
  else if (!strcmp(key, "sub_columns")) ctx->sub_columns = value ? 1 : 0;
  else if (!strcmp(key, "step_reg")) ctx->step_reg = value ? 1 : 0;
  else if (!strcmp(key, "step_spin_limit")) ctx->step_spin_limit = (unsigned)std::max(1, value);
  else if (!strcmp(key, "long_columns_from")) ctx->long_columns_from = std::max(0, std::min(value, 65536));
  else if (!strcmp(key, "nan_long_from")) ctx->nan_long_from = std::max(0, std::min(value, 65536));
  else if (!strcmp(key, "nan_long_two_phase")) ctx->nan_long_two_phase = value ? 1 : 0;
  else if (!strcmp(key, "cluster_passes")) ctx->cluster_passes = value ? 1 : 0;
  else if (!strcmp(key, "cluster_cap")) ctx->cluster_cap = value;   // (tests: -1 = ask the runtime again)
  else if (!strcmp(key, "cluster_slab_rows")) ctx->cluster_slab_rows = std::max(128, std::min(value, 8192)) / 128 * 128;
  else if (!strcmp(key, "cluster_prefetch")) ctx->cluster_prefetch = value ? 1 : 0;
  else if (!strcmp(key, "cluster_exchange")) ctx->cluster_exchange = std::max(0, std::min(value, 2));
  else if (!strcmp(key, "cluster_spin_limit")) ctx->cluster_spin_limit = (unsigned)std::max(1, value);
  else MB_ARG(false, "unknown tunable '%s'", key);
  return MBPLS_OK;
}

int mbpls_hip_profile_enable(mbpls_ctx* ctx, int on) {
  MB_ARG(ctx, "ctx is NULL");
  MB_TRY(drain_profile(ctx));
  ctx->prof = on != 0;
  return MBPLS_OK;
}

int mbpls_hip_profile_reset(mbpls_ctx* ctx) {
  MB_ARG(ctx, "ctx is NULL");
  MB_TRY(drain_profile(ctx));
  for (int k = 0; k < MBPLS_K_COUNT; ++k) {
    ctx->prof_count[k] = 0;
    ctx->prof_ms[k] = 0;
    ctx->prof_bytes[k] = 0;
  }
  return MBPLS_OK;
}

int mbpls_hip_profile_get(mbpls_ctx* ctx, int kind, int64_t* launches, double* total_ms, double* total_bytes) {
  MB_ARG(ctx && kind >= 0 && kind < MBPLS_K_COUNT, "bad argument");
  MB_TRY(drain_profile(ctx));
  if (launches) *launches = ctx->prof_count[kind];
  if (total_ms) *total_ms = ctx->prof_ms[kind];
  if (total_bytes) *total_bytes = ctx->prof_bytes[kind];
  return MBPLS_OK;
}

int mbpls_hip_malloc(mbpls_ctx* ctx, size_t bytes, void** d_ptr) {
  MB_ARG(ctx && d_ptr, "NULL argument");
  MB_HIP(hipSetDevice(ctx->device));
  *d_ptr = nullptr;
  if (bytes == 0) bytes = 8;
  MB_HIP(hipMalloc(d_ptr, bytes));
  return MBPLS_OK;
}

int mbpls_hip_free(mbpls_ctx* ctx, void* d_ptr) {
  MB_ARG(ctx, "ctx is NULL");
  if (!d_ptr) return MBPLS_OK;
  MB_SYNC_STREAM(ctx);
  MB_HIP(hipFree(d_ptr));
  return MBPLS_OK;
}

int mbpls_hip_host_alloc(mbpls_ctx* ctx, size_t bytes, void** h_ptr, void** d_ptr) {
  MB_ARG(ctx && h_ptr && d_ptr && bytes > 0, "bad argument");
  MB_HIP(hipSetDevice(ctx->device));
  *h_ptr = *d_ptr = nullptr;
  MB_HIP(hipHostMalloc(h_ptr, bytes, hipHostMallocMapped | hipHostMallocCoherent));
  memset(*h_ptr, 0, bytes);
  MB_HIP(hipHostGetDevicePointer(d_ptr, *h_ptr, 0));
  {
    std::lock_guard<std::mutex> lock(ctx->pin_mutex);
    ctx->host_blocks.push_back(*h_ptr);
  }
  return MBPLS_OK;
}

int mbpls_hip_host_free(mbpls_ctx* ctx, void* h_ptr) {
  MB_ARG(ctx, "ctx is NULL");
  if (!h_ptr) return MBPLS_OK;
  MB_SYNC_STREAM(ctx);
  {
    std::lock_guard<std::mutex> lock(ctx->pin_mutex);
    auto it = std::find(ctx->host_blocks.begin(), ctx->host_blocks.end(), h_ptr);
    if (it != ctx->host_blocks.end()) ctx->host_blocks.erase(it);
  }
  MB_HIP(hipHostFree(h_ptr));
  return MBPLS_OK;
}

int mbpls_hip_memcpy_h2d(mbpls_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  MB_ARG(ctx && (bytes == 0 || (d_dst && h_src)), "NULL argument");
  if (bytes == 0) return MBPLS_OK;
  // Small copies (n-vectors, parameter blocks; < 256 KB) bounce through a ring of 64 pinned slots of our own.  Handing the
  // pageable source to hipMemcpyAsync was correct (the runtime stages it before returning) but every third fit of a
  // shape with 48-56 KB vectors (n = 6000, 7000) the copy then EXECUTED 26-36 ms late on an idle GPU (rocprofv3
  // memory-copy trace, profiles/r04_stall_n6000.txt): a 10 ms fit took 40.
  if (bytes <= SMALL_PIN_BYTES) {
    std::lock_guard<std::mutex> lock(ctx->pin_mutex);
    if (!ctx->pin_small) {
      MB_HIP(hipHostMalloc(&ctx->pin_small, (size_t)SMALL_PIN_SLOTS * SMALL_PIN_BYTES, hipHostMallocDefault));
      for (int i = 0; i < SMALL_PIN_SLOTS; ++i) ctx->pin_small_ev[i] = nullptr;
    }
    const unsigned h = ctx->pin_small_next++ % SMALL_PIN_SLOTS;
    if (!ctx->pin_small_ev[h]) MB_HIP(hipEventCreateWithFlags(&ctx->pin_small_ev[h], hipEventDisableTiming));
    else MB_HIP(hipEventSynchronize(ctx->pin_small_ev[h]));      // 64 copies ago: done long since
    char* buf = (char*)ctx->pin_small + (size_t)h * SMALL_PIN_BYTES;
    memcpy(buf, h_src, bytes);
    MB_HIP(hipMemcpyAsync(d_dst, buf, bytes, hipMemcpyHostToDevice, ctx->stream));
    MB_HIP(hipEventRecord(ctx->pin_small_ev[h], ctx->stream));
    return MBPLS_OK;
  }
  // larger copies bounce through our own ring of pinned slots (8 MB each): the host memcpy of a chunk overlaps the DMA of
  // the ones before it
  std::lock_guard<std::mutex> lock(ctx->pin_mutex);
  if (!ctx->pin) {
    ctx->pin_half = (size_t)8 << 20;
    MB_HIP(hipHostMalloc(&ctx->pin, PIN_SLOTS * ctx->pin_half, hipHostMallocDefault));
    for (int i = 0; i < PIN_SLOTS; ++i) {
      MB_HIP(hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
      MB_HIP(hipEventRecord(ctx->pin_ev[i], ctx->stream));
    }
  }
  for (size_t off = 0; off < bytes; off += ctx->pin_half) {
    const size_t len = std::min(ctx->pin_half, bytes - off);
    const unsigned h = ctx->pin_next++ % PIN_SLOTS;
    char* buf = (char*)ctx->pin + (size_t)h * ctx->pin_half;
    // the DMA that last read this slot has finished (four copies ago: normally long done -- waiting on an event
    // recorded in front of queued kernels was seen to take until those kernels had run)
    MB_HIP(hipEventSynchronize(ctx->pin_ev[h]));
    memcpy(buf, (const char*)h_src + off, len);
    MB_HIP(hipMemcpyAsync((char*)d_dst + off, buf, len, hipMemcpyHostToDevice, ctx->stream));
    MB_HIP(hipEventRecord(ctx->pin_ev[h], ctx->stream));
  }
  return MBPLS_OK;
}

int mbpls_hip_memcpy_d2h(mbpls_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  MB_ARG(ctx && (bytes == 0 || (h_dst && d_src)), "NULL argument");
  if (bytes == 0) return MBPLS_OK;
  constexpr size_t BOUNCE = (size_t)4 << 20;
  if (bytes <= BOUNCE) {
    // small results (scores, K x K matrices, parameter blocks): DMA into a pinned buffer of the context, then a host
    // memcpy -- a blocking copy straight into pageable memory took 30-250 us of runtime bookkeeping per call
    std::lock_guard<std::mutex> lock(ctx->pin_mutex);
    if (!ctx->pin_d2h) MB_HIP(hipHostMalloc(&ctx->pin_d2h, BOUNCE, hipHostMallocDefault));
    MB_HIP(hipMemcpyAsync(ctx->pin_d2h, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MB_SYNC_STREAM(ctx);
    memcpy(h_dst, ctx->pin_d2h, bytes);
    return MBPLS_OK;
  }
  MB_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  MB_SYNC_STREAM(ctx);
  return MBPLS_OK;
}

int mbpls_hip_fence(mbpls_ctx* ctx) {
  MB_ARG(ctx, "ctx is NULL");
  MB_HIP(hipEventRecord(ctx->fence_ev, ctx->stream));
  return MBPLS_OK;
}

int mbpls_hip_memcpy_d2h_overlap(mbpls_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  MB_ARG(ctx && (bytes == 0 || (h_dst && d_src)), "NULL argument");
  if (bytes == 0) return MBPLS_OK;
  MB_HIP(hipSetDevice(ctx->device));   // may be called from a helper thread (the current device is per thread)
  // not ordered behind the pass that is running -- but behind the last fence: the small kernels that produced a finished
  // column (a scaling, an upload) were queued BEFORE the pass and must be through before the copy reads their output
  if (ctx->fence_ev) MB_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->fence_ev, 0));
  MB_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
  MB_HIP(hipStreamSynchronize(ctx->copy_stream));
  return MBPLS_OK;
}

int mbpls_hip_memcpy_d2d(mbpls_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
  MB_ARG(ctx && (bytes == 0 || (d_dst && d_src)), "NULL argument");
  if (bytes == 0) return MBPLS_OK;
  MB_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return MBPLS_OK;
}

int mbpls_hip_memset(mbpls_ctx* ctx, void* d_ptr, int value, size_t bytes) {
  MB_ARG(ctx && (bytes == 0 || d_ptr), "NULL argument");
  if (bytes == 0) return MBPLS_OK;
  MB_HIP(hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
  return MBPLS_OK;
}

int mbpls_hip_timer_start(mbpls_ctx* ctx) {
  MB_ARG(ctx, "ctx is NULL");
  MB_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  return MBPLS_OK;
}

int mbpls_hip_timer_stop(mbpls_ctx* ctx, double* elapsed_ms) {
  MB_ARG(ctx && elapsed_ms, "NULL argument");
  MB_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  MB_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  MB_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *elapsed_ms = (double)ms;
  return MBPLS_OK;
}

int mbpls_hip_mat_create(mbpls_ctx* ctx, int64_t n, int64_t p, int nblocks, const int64_t* block_offsets,
                         void* d_storage, mbpls_mat** out) {
  MB_ARG(ctx && out, "NULL argument");
  *out = nullptr;
  MB_ARG(n >= 1 && p >= 1, "empty matrix (n=%lld, p=%lld)", (long long)n, (long long)p);
  MB_ARG(nblocks >= 1 && block_offsets, "need at least one block");
  MB_ARG(block_offsets[0] == 0 && block_offsets[nblocks] == p, "block_offsets must run from 0 to p");
  for (int b = 0; b < nblocks; ++b)
    MB_ARG(block_offsets[b + 1] > block_offsets[b], "block %d is empty or offsets not increasing", b);
  MB_HIP(hipSetDevice(ctx->device));
  mbpls_mat* m = new mbpls_mat();
  m->ctx = ctx;
  m->n = n;
  m->p = p;
  m->ld = (n + 1) & ~(int64_t)1;
  m->nblocks = nblocks;
  m->block_offsets.assign(block_offsets, block_offsets + nblocks + 1);
  if (d_storage) {
    m->data = (double*)d_storage;
    m->owns = false;
  } else {
    hipError_t e = hipMalloc((void**)&m->data, sizeof(double) * (size_t)m->ld * (size_t)p);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("hipMalloc of %.3f GB for a %lld x %lld matrix failed: %s",
                (double)m->ld * p * 8 / 1e9, (long long)n, (long long)p, hipGetErrorString(e));
      delete m;
      return MBPLS_ERR_OOM;
    }
    m->owns = true;
  }
  {
    hipError_t e = hipMalloc((void**)&m->d_block_offsets, sizeof(int64_t) * (size_t)(nblocks + 1));
    if (e == hipSuccess)
      e = hipMemcpyAsync(m->d_block_offsets, m->block_offsets.data(), sizeof(int64_t) * (size_t)(nblocks + 1),
                         hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("block offsets: %s", hipGetErrorString(e));
      mbpls_hip_mat_destroy(m);
      return MBPLS_ERR_HIP;
    }
  }
  // One workgroup per chunk, one partial n-vector per workgroup.  Four chunks per CU let fast and slow CUs even out on a
  // wide matrix; on a narrow one (a column shard: 125 000 columns = 122 per chunk) the per-chunk costs -- prologue, the
  // 32 KB partial, its share of the chunk reduction -- outweigh that: with two per CU (one round of equal workgroups at
  // two resident per CU) the passes of a shard-sized fit run 3 % faster and the reduction 13 -> 9 us; at 10^6 columns
  // (976 per chunk) four stay 0.3-0.7 % ahead (rocprofv3, round 4).
  int cpc = ctx->chunks_per_cu;
  // Few samples (n <= 128, colpass_sub.hip): the fused pass runs three 4-wave workgroups per CU (136-168 VGPRs), so FOUR
  // chunks per CU are a round of three and a round of one; six are two full rounds -- loadings + deflation + sweep
  // + 6 % (n = 40), + 8-12 % (n = 64), + 7 % (n = 128), - 3 % at n = 256 (profiles/r06_ab_sub_lm1.txt and the chunk sweep
  // beside it): six up to 128 samples.
  if (ctx->chunks_adaptive && cpc == 4 && n <= 128) cpc = 6;
  if (ctx->chunks_adaptive && cpc >= 4 && p / ((int64_t)ctx->cus * cpc) < 512) cpc /= 2;
  int rc = build_chunks(ctx, m->block_offsets, ctx->cus * cpc, &m->fine);
  if (rc == MBPLS_OK) {
    // row-slab kernels: slabs of 512 rows already multiply the workgroup count
    const int slabs = (int)((n + 511) / 512);
    const int target = std::max(1, ctx->cus * 4 * ctx->xw_chunks_per_cu / slabs);
    rc = build_chunks(ctx, m->block_offsets, target, &m->coarse);
  }
  if (rc != MBPLS_OK) {
    mbpls_hip_mat_destroy(m);
    return rc;
  }
  *out = m;
  return MBPLS_OK;
}


int mbpls_hip_mat_transpose(const mbpls_mat* src, mbpls_mat* dst) {
  MB_ARG(src && dst && src != dst, "NULL argument");
  MB_ARG(dst->n == src->p && dst->p == src->n, "dst must be %lld x %lld (got %lld x %lld)", (long long)src->p,
         (long long)src->n, (long long)dst->n, (long long)dst->p);
  mbpls_ctx* ctx = src->ctx;
  dim3 grid((unsigned)((src->n + TT - 1) / TT), (unsigned)((std::max(src->p, dst->ld) + TT - 1) / TT));
  hipLaunchKernelGGL(k_transpose_mat, grid, dim3(256), 0, ctx->stream, (const double*)src->data, src->ld, src->n, src->p,
                     dst->data, dst->ld);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

int mbpls_hip_mat_destroy(mbpls_mat* m) {
  if (!m) return MBPLS_OK;
  (void)hipStreamSynchronize(m->ctx->stream);
  free_chunks(&m->fine);
  free_chunks(&m->coarse);
  free_chunks(&m->cluster);
  if (m->d_block_offsets) (void)hipFree(m->d_block_offsets);
  if (m->owns && m->data) (void)hipFree(m->data);
  delete m;
  return MBPLS_OK;
}

int mbpls_hip_mat_info(const mbpls_mat* m, int64_t* n, int64_t* p, int64_t* ld, void** d_data, int* nblocks,
                       int* nchunks) {
  MB_ARG(m, "matrix is NULL");
  if (n) *n = m->n;
  if (p) *p = m->p;
  if (ld) *ld = m->ld;
  if (d_data) *d_data = m->data;
  if (nblocks) *nblocks = m->nblocks;
  if (nchunks) *nchunks = m->fine.nchunks;
  return MBPLS_OK;
}

int mbpls_hip_mat_upload(mbpls_mat* m, const double* h_src, int64_t ld_host, int layout, int64_t col0,
                         int64_t ncols) {
  MB_ARG(m && h_src, "NULL argument");
  MB_ARG(col0 >= 0 && ncols >= 0 && col0 + ncols <= m->p, "column range [%lld,+%lld) outside matrix",
         (long long)col0, (long long)ncols);
  if (ncols == 0) return MBPLS_OK;
  mbpls_ctx* ctx = m->ctx;
  const int64_t n = m->n, ld = m->ld;
  if (layout == MBPLS_COL_MAJOR) {
    MB_ARG(ld_host >= n, "ld_host < n");
    MB_HIP(hipMemcpy2DAsync(m->data + col0 * ld, ld * sizeof(double), h_src, ld_host * sizeof(double),
                            n * sizeof(double), ncols, hipMemcpyHostToDevice, ctx->stream));
    if (ld > n) {
      hipLaunchKernelGGL(k_zero_pad, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, ctx->stream, m->data, ld,
                         n, col0, ncols);
      MB_HIP(hipGetLastError());
    }
    MB_SYNC_STREAM(ctx);
    return MBPLS_OK;
  }
  MB_ARG(layout == MBPLS_ROW_MAJOR, "unknown layout %d", layout);
  MB_ARG(ld_host >= ncols, "ld_host < ncols");
  // chunked: strided 2-D copy of [n][w] into staging, transpose on the GPU
  const size_t budget = (size_t)64 << 20;
  int64_t w = std::max<int64_t>(TT, (int64_t)(budget / (sizeof(double) * (size_t)n)) / TT * TT);
  w = std::min(w, (ncols + TT - 1) / TT * TT);
  MB_TRY(ensure_stage(ctx, sizeof(double) * (size_t)n * (size_t)w));
  for (int64_t c = 0; c < ncols; c += w) {
    const int64_t cw = std::min(w, ncols - c);
    MB_HIP(hipMemcpy2DAsync(ctx->stage, cw * sizeof(double), h_src + c, ld_host * sizeof(double),
                            cw * sizeof(double), n, hipMemcpyHostToDevice, ctx->stream));
    dim3 grid((unsigned)((cw + TT - 1) / TT), (unsigned)((ld + TT - 1) / TT));
    hipLaunchKernelGGL(k_transpose_in, grid, dim3(256), 0, ctx->stream, (const double*)ctx->stage, n, cw,
                       m->data + (col0 + c) * ld, ld);
    MB_HIP(hipGetLastError());
    MB_SYNC_STREAM(ctx);  // staging is reused
  }
  return MBPLS_OK;
}

int mbpls_hip_mat_download(const mbpls_mat* m, double* h_dst, int64_t ld_host, int layout, int64_t col0,
                           int64_t ncols) {
  MB_ARG(m && h_dst, "NULL argument");
  MB_ARG(col0 >= 0 && ncols >= 0 && col0 + ncols <= m->p, "column range outside matrix");
  if (ncols == 0) return MBPLS_OK;
  mbpls_ctx* ctx = m->ctx;
  const int64_t n = m->n, ld = m->ld;
  if (layout == MBPLS_COL_MAJOR) {
    MB_ARG(ld_host >= n, "ld_host < n");
    MB_HIP(hipMemcpy2DAsync(h_dst, ld_host * sizeof(double), m->data + col0 * ld, ld * sizeof(double),
                            n * sizeof(double), ncols, hipMemcpyDeviceToHost, ctx->stream));
    MB_SYNC_STREAM(ctx);
    return MBPLS_OK;
  }
  MB_ARG(layout == MBPLS_ROW_MAJOR, "unknown layout %d", layout);
  MB_ARG(ld_host >= ncols, "ld_host < ncols");
  const size_t budget = (size_t)64 << 20;
  int64_t w = std::max<int64_t>(TT, (int64_t)(budget / (sizeof(double) * (size_t)n)) / TT * TT);
  w = std::min(w, (ncols + TT - 1) / TT * TT);
  MB_TRY(ensure_stage(ctx, sizeof(double) * (size_t)n * (size_t)w));
  for (int64_t c = 0; c < ncols; c += w) {
    const int64_t cw = std::min(w, ncols - c);
    dim3 grid((unsigned)((cw + TT - 1) / TT), (unsigned)((n + TT - 1) / TT));
    hipLaunchKernelGGL(k_transpose_out, grid, dim3(256), 0, ctx->stream, (const double*)(m->data + (col0 + c) * ld),
                       ld, n, cw, (double*)ctx->stage);
    MB_HIP(hipGetLastError());
    MB_HIP(hipMemcpy2DAsync(h_dst + c, ld_host * sizeof(double), ctx->stage, cw * sizeof(double),
                            cw * sizeof(double), n, hipMemcpyDeviceToHost, ctx->stream));
    MB_SYNC_STREAM(ctx);
  }
  return MBPLS_OK;
}

int mbpls_hip_mat_copy(mbpls_mat* dst, const mbpls_mat* src) {
  MB_ARG(dst && src, "NULL argument");
  MB_ARG(dst->n == src->n && dst->p == src->p && dst->ld == src->ld, "shape mismatch");
  MB_HIP(hipMemcpyAsync(dst->data, src->data, sizeof(double) * (size_t)src->ld * (size_t)src->p,
                        hipMemcpyDeviceToDevice, dst->ctx->stream));
  return MBPLS_OK;
}

}  // extern "C"

namespace mbpls {
const void* anchor_colpass();
const void* anchor_colpass_cluster();
const void* anchor_colpass_long();
const void* anchor_axpy_cols();
const void* anchor_cv_batch();
const void* anchor_gemv();
const void* anchor_gen();
const void* anchor_gram();
const void* anchor_resident();
const void* anchor_small();
const void* anchor_step_reg();
const void* anchor_xtv_mfma();
const void* anchor_xw_mfma();
}  // namespace mbpls

extern "C" int mbpls_hip_warmup(mbpls_ctx* ctx) {
  MB_ARG(ctx, "ctx is NULL");
  MB_HIP(hipSetDevice(ctx->device));
  // code objects are loaded lazily, one per translation unit, on the first use of one of their kernels: ~13 ms of
  // the first fit of a process went there
  const void* (*anchors[])() = {mbpls::anchor_colpass, mbpls::anchor_colpass_cluster, mbpls::anchor_colpass_long, mbpls::anchor_axpy_cols, mbpls::anchor_cv_batch, mbpls::anchor_gemv, mbpls::anchor_gen, mbpls::anchor_gram, mbpls::anchor_resident, mbpls::anchor_small, mbpls::anchor_step_reg, mbpls::anchor_xtv_mfma, mbpls::anchor_xw_mfma};
  for (auto fn : anchors) {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, fn());
  }
  (void)hipGetLastError();
  // the pinned bounce buffers of the small copies, an event (the first creation sets up runtime state)
  {
    std::lock_guard<std::mutex> lock(ctx->pin_mutex);
    if (!ctx->pin) {
      ctx->pin_half = (size_t)8 << 20;
      MB_HIP(hipHostMalloc(&ctx->pin, PIN_SLOTS * ctx->pin_half, hipHostMallocDefault));
      for (int i = 0; i < PIN_SLOTS; ++i) {
        MB_HIP(hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
        MB_HIP(hipEventRecord(ctx->pin_ev[i], ctx->stream));
      }
    }
    if (!ctx->pin_small) {
      MB_HIP(hipHostMalloc(&ctx->pin_small, (size_t)SMALL_PIN_SLOTS * SMALL_PIN_BYTES, hipHostMallocDefault));
      for (int i = 0; i < SMALL_PIN_SLOTS; ++i) ctx->pin_small_ev[i] = nullptr;
    }
  }
  MB_TRY(mbpls::copier_warmup(ctx));          // the copier thread, its stream and bounce buffers
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) (void)hipEventDestroy(e);
  return MBPLS_OK;
}
