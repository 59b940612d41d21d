// Copier: ONE native thread per context that carries finished results to the host behind events while the compute stream
// keeps running (the reference has no counterpart: its attributes are numpy arrays from the start).
//
// A device-resident fit (engine._ResidentQueue) hands out its p-length attribute columns as they become final: W~[:, k]
// after the last sweep of component k, P[:, k] and R_[:, k] after its loadings pass, the scaler statistics after the first
// pass.  A job = an event recorded on the compute stream at submission + a list of device-to-host copies + (optionally)
// the block normalisation w_b = w~_b / ||w~_b|| (mbpls.py:860 / :510-512) of a column that has just arrived.  Jobs run in
// submission order; mbpls_hip_copier_wait(ticket) returns when that job and every earlier one are complete.
//
// Rounds 2-4 ran this loop on a Python thread.  At the tail of a fit -- the last loadings pass has finished, three columns
// are still to come, the fitting thread is busy with its own bookkeeping -- that thread needed the interpreter lock
// between any two copies and took 200-320 us per 1 MB column that costs 60 us by itself (tools/host_timeline.py, shard
// sized fit: 0.45 of the 1.5 ms it spent outside its passes).  Here nothing takes a lock the fitting thread holds, copies
// of up to 4 MB are double-buffered through two pinned bounce buffers (the DMA of copy i+1 runs while copy i is moved to
// its destination), and the waiter spins briefly before it sleeps.
//
// Why small copies bounce at all: for a pageable destination the runtime pins the caller's pages for the transfer, and with
// 1.3 MB columns (n = 6000, p = 166 666: the scaler statistics sent out behind the first pass) every later fit's first
// host-to-device copy then EXECUTED 26-36 ms late on an idle GPU -- a 10 ms fit took 40 (rocprofv3 copy trace; gone with
// MBPLS_STREAM_STATS=0, profiles/r04_stall_n6000.txt).  Large columns (8 MB at BASELINE config[1]) keep the direct path:
// 55 GB/s into pages the host side has mapped.
//
// The move out of the bounce buffer is what a small copy costs: one core moves ~7 GB/s into fresh pages (MBPLS_COPIER_TRACE=1:
// 3 MB behind the last pass = 25 us of waiting for the DMA and 410 us of memcpy; the block normalisation as a second sweep
// over the column 50-270 us more).  So a landed copy is cut into 64 KB pieces that the copier thread and COPIER_HELPERS
// helper threads take from a shared counter, and the normalisation is done on the piece while it is in cache (one read of
// the bounce buffer, both destinations written).  The helpers sleep between jobs (woken when a job's event has completed,
// i.e. while its first DMA is in flight) and poll for ~200 us after their last piece, so they stay awake through the
// copies of one job.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <thread>

#include "common.h"

namespace mbpls {

struct CopierJob {
  uint64_t id = 0;
  hipEvent_t ev = nullptr;           // NULL: wait for *flag >= flag_min instead
  const volatile double* flag = nullptr;
  double flag_min = 0.0;
  std::vector<mbpls_copy> copies;
  bool has_scale = false;
  const double* s_src = nullptr;
  double* s_dst = nullptr;
  const double* s_nu2 = nullptr;
  std::vector<int64_t> s_offs;
};

struct CopierState {
  mbpls_ctx* ctx = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<CopierJob> jobs;
  uint64_t submitted = 0;
  std::atomic<uint64_t> done{0};
  bool stop = false;
  int error = MBPLS_OK;              // first failure since the last wait that reported one
  std::string error_text;
  hipStream_t stream = nullptr;      // the copier's own non-blocking stream
  void* bounce[2] = {nullptr, nullptr};
  hipEvent_t bev[2] = {nullptr, nullptr};
  // the landed copy being moved out of its bounce buffer, shared with the helper threads
  struct Land {
    const char* src = nullptr;
    char* dst = nullptr;             // NULL: only the scaled copy is written (normalisation in place)
    size_t bytes = 0;
    double* sdst = nullptr;          // != NULL: sdst[j] = src[j] / sqrt(nu2[block of j]) as well
    const double* nu2 = nullptr;
    const int64_t* offs = nullptr;
    int nblocks = 0;
  } land;
  // (epoch << 32) | next piece.  `land` belongs to the epoch in the upper half.  The copier thread rewrites it only after
  // every piece is finished, the lower half has been parked at LAND_CLOSED and no helper is copying it (land_readers);
  // a helper copies it only while it is registered as a reader AND the ticket shows an open epoch, and claims pieces by
  // compare-and-swap on that unchanged epoch.
  std::atomic<uint64_t> land_ticket{0xFFFFFFFFull};
  std::atomic<int> land_readers{0};        // helper threads copying `land` right now: it is not rewritten meanwhile
  std::atomic<int64_t> land_left{0};       // pieces not finished yet
  uint64_t land_epoch = 0;                 // (copier thread only)
  std::vector<std::thread> helpers;
  std::mutex hmu;
  std::condition_variable cv_help;
  std::atomic<bool> hstop{false};
  std::atomic<bool> stopping{false};       // the context is being destroyed: a job still waiting for its flag gives up
};

constexpr int COPIER_HELPERS = 3;
constexpr size_t COPIER_PIECE = (size_t)64 << 10;

// one piece of the landed copy: bytes [a, b) of the bounce buffer
static void land_piece(const CopierState::Land& L, size_t a, size_t b) {
  if (L.dst) memcpy(L.dst + a, L.src + a, b - a);
  if (L.sdst) {
    int64_t j = (int64_t)(a / 8);
    const int64_t j1 = (int64_t)(b / 8);
    const bool in_place = (const void*)L.src == (const void*)L.sdst;   // (a large column, normalised where it arrived)
    for (int blk = 0; blk < L.nblocks && j < j1; ++blk) {
      const int64_t e = L.offs[blk + 1] < j1 ? L.offs[blk + 1] : j1;
      if (e <= j) continue;
      const double d = sqrt(L.nu2[blk]);
      if (in_place) {
        double* out = L.sdst;
        for (; j < e; ++j) out[j] = out[j] / d;
      } else {
        const double* __restrict__ src = (const double*)L.src;
        double* __restrict__ out = L.sdst;
        for (; j < e; ++j) out[j] = src[j] / d;
      }
    }
  }
}

constexpr uint64_t LAND_CLOSED = 0xFFFFFFFFull;

// take pieces of epoch `epoch` (whose description is L) until none are left (others may still be working on theirs
// when this returns)
static void land_work(CopierState* cs, uint64_t epoch, const CopierState::Land& L) {
  const uint64_t npieces = (L.bytes + COPIER_PIECE - 1) / COPIER_PIECE;
  for (;;) {
    uint64_t t = cs->land_ticket.load(std::memory_order_acquire);
    for (;;) {
      if ((t >> 32) != epoch || (t & 0xFFFFFFFFull) >= npieces) return;
      if (cs->land_ticket.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel, std::memory_order_acquire)) break;
    }
    const size_t a = (size_t)(t & 0xFFFFFFFFull) * COPIER_PIECE;
    land_piece(L, a, std::min(L.bytes, a + COPIER_PIECE));
    cs->land_left.fetch_sub(1, std::memory_order_acq_rel);
  }
}

static void helper_main(CopierState* cs) {
  uint64_t seen = cs->land_ticket.load(std::memory_order_acquire) >> 32;
  for (;;) {
    // poll for a while (the copies of one job follow each other within tens of microseconds), then sleep
    bool got = false;
    for (int spin = 0; spin < 12000; ++spin) {
      if ((cs->land_ticket.load(std::memory_order_acquire) >> 32) != seen) {
        got = true;
        break;
      }
      if (cs->hstop.load(std::memory_order_relaxed)) return;
      __builtin_ia32_pause();
    }
    if (!got) {
      std::unique_lock<std::mutex> lock(cs->hmu);
      if (cs->hstop.load(std::memory_order_relaxed)) return;
      cs->cv_help.wait_for(lock, std::chrono::milliseconds(50));   // (woken by the copier thread when a job's event is done)
      if (cs->hstop.load(std::memory_order_relaxed)) return;
      continue;
    }
    // register as a reader, then look again: an open epoch cannot be rewritten while we are registered
    // (store-then-load on two variables against the writer's store-then-load below: both sides sequentially
    // consistent, or the reader could see the old open ticket while the writer sees no reader)
    cs->land_readers.fetch_add(1, std::memory_order_seq_cst);
    const uint64_t t = cs->land_ticket.load(std::memory_order_seq_cst);
    const bool open = (t & 0xFFFFFFFFull) != LAND_CLOSED;
    CopierState::Land L;
    if (open) L = cs->land;
    cs->land_readers.fetch_sub(1, std::memory_order_acq_rel);
    seen = t >> 32;
    if (open) land_work(cs, seen, L);             // (parked: the pieces of that epoch are done; wait for the next one)
  }
}

constexpr size_t COPIER_BOUNCE_BYTES = (size_t)4 << 20;

// MBPLS_COPIER_TRACE=1: one line per job on stderr (microseconds since the job's event completed)
static bool copier_trace() {
  static const bool on = [] {
    const char* e = getenv("MBPLS_COPIER_TRACE");
    return e && e[0] == '1';
  }();
  return on;
}
static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int copier_run(CopierState* cs, CopierJob& job) {
#define CP_HIP(call)                                                            \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      cs->error_text = std::string(#call) + ": " + hipGetErrorString(e_);       \
      return MBPLS_ERR_HIP;                                                     \
    }                                                                           \
  } while (0)
  const bool trace = copier_trace();
  const double t_in = trace ? now_us() : 0.0;
  if (job.ev) {
    CP_HIP(hipEventSynchronize(job.ev));
  } else {
    // a kernel behind the producers of this job's data stores the value (host-mapped memory): poll, politely after a while
    for (unsigned spins = 0; !(*job.flag >= job.flag_min); ++spins) {
      if (spins < 4000) __builtin_ia32_pause();
      else std::this_thread::sleep_for(std::chrono::microseconds(20));
      if (cs->stopping.load(std::memory_order_relaxed)) {
        cs->error_text = "the context was destroyed while a copy job was waiting for its flag";
        return MBPLS_ERR_HIP;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  const double t_ev = trace ? now_us() : 0.0;
  cs->cv_help.notify_all();          // the helpers wake up while the first DMA is in flight
  double t_dma = 0.0, t_cpy = 0.0, t_scale = 0.0;
  int prev = -1, prev_slot = 0, nsmall = 0;
  bool scaled = !job.has_scale;      // the normalisation has been done with the copy that delivered its source
  auto share_out = [&](CopierState::Land& L) {     // cs->land is filled in: its pieces go to this thread and the helpers
    const int64_t npieces = (int64_t)((L.bytes + COPIER_PIECE - 1) / COPIER_PIECE);
    cs->land_left.store(npieces, std::memory_order_relaxed);
    const uint64_t epoch = ++cs->land_epoch;
    cs->land_ticket.store(epoch << 32, std::memory_order_release);       // open: pieces 0 .. npieces - 1
    land_work(cs, epoch, L);
    while (cs->land_left.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
    cs->land_ticket.store((epoch << 32) | LAND_CLOSED, std::memory_order_seq_cst);
  };
  // The normalisation of a LARGE column, which arrived straight in the caller's pages: one thread divides 20 GB/s worth of
  // column, 4.0 ms for the 80 MB of p = 10^7 -- as long as the DMA of the job's other two columns (55 GB/s), and with few
  // samples the passes are shorter than the copies, so the fit waits for this thread (n = 100, p = 10^7, K = 4: 42 ms, 16
  // of them this loop; MBPLS_COPIER_TRACE=1).  The column is therefore shared out in pieces like a landed small copy --
  // and while the DMA of the job's NEXT large column is in flight (w~ is the first copy of its job, engine.nipals_resident).
  // (A copy into pageable memory returns when it is complete, whatever its name says: the division is opened for the
  // helpers BEFORE the call and this thread joins them after it.)
  bool pending_scale = false;        // the source column of the normalisation has arrived, the division is still to do
  uint64_t scale_epoch = 0;          // != 0: the division is open, the helpers are at it
  auto scale_open = [&]() {
    while (cs->land_readers.load(std::memory_order_seq_cst) > 0) __builtin_ia32_pause();
    CopierState::Land& L = cs->land;
    L = CopierState::Land{};
    L.src = (const char*)job.s_src;
    L.dst = nullptr;
    L.bytes = (size_t)job.s_offs.back() * sizeof(double);
    L.sdst = job.s_dst;
    L.nu2 = job.s_nu2;
    L.offs = job.s_offs.data();
    L.nblocks = (int)job.s_offs.size() - 1;
    cs->land_left.store((int64_t)((L.bytes + COPIER_PIECE - 1) / COPIER_PIECE), std::memory_order_relaxed);
    scale_epoch = ++cs->land_epoch;
    cs->land_ticket.store(scale_epoch << 32, std::memory_order_release);
    cs->cv_help.notify_all();        // (a large DMA outlasts the helpers' polling)
    pending_scale = false;
  };
  auto scale_finish = [&]() {
    const double t0 = trace ? now_us() : 0.0;
    land_work(cs, scale_epoch, cs->land);
    while (cs->land_left.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
    cs->land_ticket.store((scale_epoch << 32) | LAND_CLOSED, std::memory_order_seq_cst);
    scale_epoch = 0;
    scaled = true;
    if (trace) t_scale += now_us() - t0;
  };
  auto land = [&]() -> int {         // the copy in flight in bounce[prev_slot] has arrived: move it to its destination
    if (prev < 0) return MBPLS_OK;
    const double a = trace ? now_us() : 0.0;
    CP_HIP(hipEventSynchronize(cs->bev[prev_slot]));
    const double b = trace ? now_us() : 0.0;
    const mbpls_copy& c = job.copies[prev];
    while (cs->land_readers.load(std::memory_order_seq_cst) > 0) __builtin_ia32_pause();   // (ticket is parked: no new readers copy)
    CopierState::Land& L = cs->land;
    L = CopierState::Land{};
    L.src = (const char*)cs->bounce[prev_slot];
    L.dst = (char*)c.h_dst;
    L.bytes = c.bytes;
    if (job.has_scale && !scaled && (const void*)job.s_src == (const void*)c.h_dst &&
        c.bytes == (size_t)job.s_offs.back() * sizeof(double) && c.bytes % 8 == 0) {
      L.sdst = job.s_dst;
      L.nu2 = job.s_nu2;
      L.offs = job.s_offs.data();
      L.nblocks = (int)job.s_offs.size() - 1;
      if ((const void*)job.s_dst == (const void*)c.h_dst) L.dst = nullptr;    // in place: the scaled values only
      scaled = true;
    }
    share_out(L);
    if (trace) {
      t_dma += b - a;
      t_cpy += now_us() - b;
    }
    prev = -1;
    return MBPLS_OK;
  };
  for (int i = 0; i < (int)job.copies.size(); ++i) {
    const mbpls_copy& c = job.copies[i];
    if (c.bytes == 0) continue;
    if (c.bytes <= COPIER_BOUNCE_BYTES) {
      const int slot = nsmall++ & 1;
      CP_HIP(hipMemcpyAsync(cs->bounce[slot], c.d_src, c.bytes, hipMemcpyDeviceToHost, cs->stream));
      CP_HIP(hipEventRecord(cs->bev[slot], cs->stream));
      const int rc = land();
      if (rc != MBPLS_OK) return rc;
      prev = i;
      prev_slot = slot;
    } else {
      // large columns (8 MB at BASELINE config[1]): straight into the caller's pages (mapped ahead by the host side)
      const int rc = land();
      if (rc != MBPLS_OK) return rc;
      if (pending_scale) scale_open();             // (of an earlier column of this job, beside this column's DMA)
      const hipError_t e_copy = hipMemcpyAsync(c.h_dst, c.d_src, c.bytes, hipMemcpyDeviceToHost, cs->stream);
      const hipError_t e_sync = e_copy == hipSuccess ? hipStreamSynchronize(cs->stream) : e_copy;
      if (scale_epoch) scale_finish();             // (also when the copy failed: the epoch must not stay open)
      CP_HIP(e_sync);
      if (job.has_scale && !scaled && (const void*)job.s_src == (const void*)c.h_dst &&
          c.bytes == (size_t)job.s_offs.back() * sizeof(double))
        pending_scale = true;
    }
  }
  {
    const int rc = land();
    if (rc != MBPLS_OK) return rc;
  }
  // Everything on the copier's stream has completed (each copy was waited for through its event) -- but the runtime only
  // releases the command objects of a stream when the STREAM is synchronised: without this call every copy of more than
  // 16 KB left 2.2-3 KB of host memory behind (a fit with no column above 4 MB and no vector below 16 KB: 130 KB per fit,
  // found with an LD_PRELOAD allocation counter; smaller copies take a path of the runtime that synchronises by itself).
  if (nsmall > 0) CP_HIP(hipStreamSynchronize(cs->stream));
  if (!scaled && (size_t)job.s_offs.back() * sizeof(double) > COPIER_BOUNCE_BYTES) {
    scale_open();
    scale_finish();
  }
  const double t_s0 = trace ? now_us() : 0.0;
  if (!scaled) {                     // (its source did not come with one of this job's small copies)
    const int B = (int)job.s_offs.size() - 1;
    for (int b = 0; b < B; ++b) {
      const double d = sqrt(job.s_nu2[b]);
      const double* __restrict__ src = job.s_src;
      double* __restrict__ dst = job.s_dst;
      if (src == dst) {
        for (int64_t j = job.s_offs[b]; j < job.s_offs[b + 1]; ++j) job.s_dst[j] = job.s_dst[j] / d;
      } else {
        for (int64_t j = job.s_offs[b]; j < job.s_offs[b + 1]; ++j) dst[j] = src[j] / d;
      }
    }
  }
  if (trace) {
    size_t bytes = 0;
    for (auto& c : job.copies) bytes += c.bytes;
    t_scale += now_us() - t_s0;
    fprintf(stderr, "[copier] job %llu: %d copies %.3f MB | waited for its event %.0f us | then: total %.0f us (waiting for DMA %.0f, memcpy %.0f, scaling %.0f)\n",
            (unsigned long long)job.id, (int)job.copies.size(), bytes / 1e6, t_ev - t_in, now_us() - t_ev, t_dma, t_cpy, t_scale);
  }
  return MBPLS_OK;
#undef CP_HIP
}

static void copier_main(CopierState* cs) {
  (void)hipSetDevice(cs->ctx->device);
  for (;;) {
    CopierJob job;
    {
      std::unique_lock<std::mutex> lock(cs->mu);
      cs->cv_job.wait(lock, [&] { return cs->stop || !cs->jobs.empty(); });
      if (cs->jobs.empty()) return;          // (stop: the queue is drained first)
      job = std::move(cs->jobs.front());
      cs->jobs.pop_front();
    }
    const int rc = copier_run(cs, job);
    if (job.ev) {
      std::lock_guard<std::mutex> lock(cs->ctx->user_event_mutex);
      cs->ctx->user_event_pool.push_back(job.ev);
    }
    {
      std::lock_guard<std::mutex> lock(cs->mu);
      if (rc != MBPLS_OK && cs->error == MBPLS_OK) cs->error = rc;
      cs->done.store(job.id, std::memory_order_release);
    }
    cs->cv_done.notify_all();
  }
}

static int copier_get(mbpls_ctx* ctx, CopierState** out) {
  if (!ctx->copier) {
    auto* cs = new CopierState();
    cs->ctx = ctx;
    MB_HIP(hipStreamCreateWithFlags(&cs->stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      MB_HIP(hipHostMalloc(&cs->bounce[i], COPIER_BOUNCE_BYTES, hipHostMallocDefault));
      MB_HIP(hipEventCreateWithFlags(&cs->bev[i], hipEventDisableTiming));
    }
    cs->th = std::thread(copier_main, cs);
    for (int i = 0; i < COPIER_HELPERS; ++i) cs->helpers.emplace_back(helper_main, cs);
    ctx->copier = cs;
  }
  *out = (CopierState*)ctx->copier;
  return MBPLS_OK;
}

void copier_shutdown(mbpls_ctx* ctx) {
  auto* cs = (CopierState*)ctx->copier;
  if (!cs) return;
  cs->stopping.store(true);
  {
    std::lock_guard<std::mutex> lock(cs->mu);
    cs->stop = true;
  }
  cs->cv_job.notify_all();
  if (cs->th.joinable()) cs->th.join();
  cs->hstop.store(true);
  {
    std::lock_guard<std::mutex> lock(cs->hmu);
  }
  cs->cv_help.notify_all();
  for (auto& h : cs->helpers)
    if (h.joinable()) h.join();
  for (int i = 0; i < 2; ++i) {
    if (cs->bounce[i]) (void)hipHostFree(cs->bounce[i]);
    if (cs->bev[i]) (void)hipEventDestroy(cs->bev[i]);
  }
  if (cs->stream) (void)hipStreamDestroy(cs->stream);
  delete cs;
  ctx->copier = nullptr;
}

int copier_warmup(mbpls_ctx* ctx) {
  CopierState* cs = nullptr;
  return copier_get(ctx, &cs);
}

}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_copier_submit(mbpls_ctx* ctx, int ncopies, const mbpls_copy* copies, const mbpls_unit_scale* scale,
                            const double* h_flag, double flag_at_least, uint64_t* ticket) {
  MB_ARG(ctx && ticket && ncopies >= 0 && (ncopies == 0 || copies), "bad argument");
  for (int i = 0; i < ncopies; ++i)
    MB_ARG(copies[i].bytes == 0 || (copies[i].h_dst && copies[i].d_src), "copy %d: NULL pointer", i);
  MB_ARG(!scale || (scale->h_src && scale->h_dst && scale->offsets && scale->h_nu2 && scale->nblocks >= 1),
         "bad mbpls_unit_scale");
  CopierState* cs = nullptr;
  MB_TRY(copier_get(ctx, &cs));
  CopierJob job;
  if (h_flag) {
    job.flag = h_flag;
    job.flag_min = flag_at_least;
  } else {
    {
      std::lock_guard<std::mutex> lock(ctx->user_event_mutex);
      if (!ctx->user_event_pool.empty()) {
        job.ev = ctx->user_event_pool.back();
        ctx->user_event_pool.pop_back();
      }
    }
    if (!job.ev) MB_HIP(hipEventCreateWithFlags(&job.ev, hipEventDisableTiming));
    MB_HIP(hipEventRecord(job.ev, ctx->stream));
  }
  job.copies.assign(copies, copies + ncopies);
  if (scale) {
    job.has_scale = true;
    job.s_src = scale->h_src;
    job.s_dst = scale->h_dst;
    job.s_nu2 = scale->h_nu2;
    job.s_offs.assign(scale->offsets, scale->offsets + scale->nblocks + 1);
  }
  {
    std::lock_guard<std::mutex> lock(cs->mu);
    job.id = ++cs->submitted;
    *ticket = job.id;
    cs->jobs.push_back(std::move(job));
  }
  cs->cv_job.notify_one();
  return MBPLS_OK;
}

int mbpls_hip_copier_wait(mbpls_ctx* ctx, uint64_t ticket) {
  MB_ARG(ctx, "ctx is NULL");
  auto* cs = (CopierState*)ctx->copier;
  if (!cs || ticket == 0) return MBPLS_OK;
  // the last columns of a fit are on their way when the fitting thread gets here: poll before sleeping (a wake-up through
  // the condition variable costs 30-60 us)
  for (int spin = 0; spin < 20000 && cs->done.load(std::memory_order_acquire) < ticket; ++spin) __builtin_ia32_pause();
  std::unique_lock<std::mutex> lock(cs->mu);
  cs->cv_done.wait(lock, [&] { return cs->done.load(std::memory_order_acquire) >= ticket; });
  if (cs->error != MBPLS_OK) {
    const int rc = cs->error;
    set_error("copier thread: %s", cs->error_text.c_str());
    cs->error = MBPLS_OK;
    return rc;
  }
  return MBPLS_OK;
}

}  // extern "C"
