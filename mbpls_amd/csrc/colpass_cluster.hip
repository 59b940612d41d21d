// Column passes for LONG columns with ONE read of X per pass: a CLUSTER of workgroups shares a column.
//
// colpass.hip holds a column in the registers of one workgroup (n <= 8192).  A longer column does not fit one
// workgroup, and colpass_long.hip therefore makes two streaming phases per pass (dot products, then the
// accumulation t_b = X_b g_b): two reads of X, 37 % of the HBM peak in effect.  Here W = ceil(n / 8192) workgroups --
// a cluster -- walk the SAME chunk of columns together: workgroup w keeps the rows [w S, (w+1) S) of the current
// columns in registers (16 rows per thread, 512 threads = 8 waves, 2 per SIMD: 256 registers per lane), so a pass
// is again one visit of every element.  What a column needs from the whole cluster -- its dot products with the sweep
// vector / the score vector, or the three shifted sums of the statistics pass -- is 1..3 doubles per workgroup:
//
//   * every workgroup reduces its slab (DPP wave sums, wave partials through LDS) and PUBLISHES the partial sums as
//     8-byte agent-scope stores into a per-column slot  xch[column][value][w]  of fine-grained device memory that a
//     small kernel filled with a sentinel (a NaN payload no arithmetic produces) before the pass -- the datum is its
//     own flag: no fence, no counter, no ordering between values (MI355X_MICROARCH.md, "granule" hand-off);
//   * wave 0 of every workgroup POLLS the W partials of the column (agent-scope loads), adds them in slab order --
//     every workgroup adds the same numbers in the same order and so takes the same decisions -- and leaves the
//     totals in LDS for the other waves.
//
// The hand-off costs 1-3 us under streaming load, about one column-slab of load time (64 KB at ~27 GB/s per CU), so
// the loop is software-pipelined over a ring of D = 4 register-resident columns: in iteration i the partial sums of
// column i + L are produced and published, then (one workgroup barrier) wave 0 polls the totals of column i + 1 while
// the other waves consume the totals of column i (weights, loadings, the accumulation acc += c_j x_j) and refill the
// slot with column i + D.  One barrier per column; loads run D - L columns ahead of their first use, the exchange L - 1.
//
// Flavours: STATS (statistics-only first pass + first sweep, implicit standardisation), SWEEP, DEFL + SWEEP (loadings,
// lazy deflation, next sweep), DEFL (loadings only), each optionally on centred columns (CEN, offset data) -- the
// never-writes-X forms of colpass.hip with the same arguments, the same per-chunk outputs (second stage: reduce_chunks)
// and the same device-resident pass parameters, so nothing above the C ABI changes.  The storing forms (explicit
// deflation, stored standardisation, row masks) stay with the two-phase passes.
//
// Clusters are persistent: G = floor(CUs / W) of them, cluster c walks the chunks c, c + G, ... of a chunk table built
// for exactly G chunks (more only when there are more blocks than clusters).  All W G <= CUs workgroups are co-resident
// (one per CU: 128 KB of LDS each), and the members of a cluster have adjacent block indices; every poll is bounded
// (spin limit -> the pass gives up, flags it, and the host raises) so a broken queue cannot hang the device.
//
// Bytes (algorithmic): 8np per pass, read once.  The exchange moves (p + chunks) x 32 W bytes -- < 0.1 % of X.
#include <algorithm>

#include "common.h"
#include "colpass_util.h"

namespace mbpls {

// 16 rows per thread, a ring of 3 columns, partial sums 2 columns ahead of their consumption; 512 threads = 8 waves,
// 2 per SIMD: 256 registers per lane.  What bounds these passes at 5.2-5.6 TB/s is the data a CU can hold while a
// column waits: about two column times for the hand-off (a store and a load through the fabric while every CU streams)
// plus one for the refill = 3 x 64 KB, all the registers there are (a ring of 4 columns spills 40-90 registers; 4
// columns of 12 rows measured 10-20 % slower: smaller slabs need proportionally more columns in flight).
constexpr int CL_TMAX = 512;
constexpr int CL_NV = 4;       // doubles per (column, workgroup) exchange slot
constexpr unsigned long long CL_SENT = 0x7FF8C0DEC0DEC0DEull;   // "not yet published"

struct ClArgs {
  PassArgs a;
  int W, G, nchunks;
  int64_t S;             // rows per slab (multiple of 128)
  int64_t p;
  double* xch;           // [(p + nchunks)][CL_NV][W]
  unsigned spin_limit;
  int prefetch;          // shared exchange wave: issue the loads of the next slot one iteration ahead
  unsigned* err;         // device word: polls that gave up
  unsigned* flag;        // host-mapped word: set when a poll gave up (the host sees it without a copy)
  double* par;           // device-resident pass parameters (writable view of a.dpar) or NULL
};

__global__ __launch_bounds__(256) void k_cl_fill(int64_t count, unsigned long long* __restrict__ out,
                                                 const double* __restrict__ dpar) {
  if (dpar && dpar[2] != 0.0) return;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i + 1 < count) {
    ulonglong2 q;
    q.x = CL_SENT;
    q.y = CL_SENT;
    *reinterpret_cast<ulonglong2*>(out + i) = q;
  } else if (i < count) {
    out[i] = CL_SENT;
  }
}

// The exchange wave's view of one slot: the NV x W partial sums, flattened (value-major, as they lie in memory) over the
// lanes in rounds of 64.  Loads of up to CL_PRE rounds can be issued AHEAD (prefetch) and looked at one iteration later,
// so the 1-2 us round trip of a fine-grained load is not on the critical path of a column; the loads of all rounds of
// a group are in flight together (a slot of W = 128 workgroups x 3 values is six rounds).
constexpr int CL_PRE = 4;
constexpr int CL_RND = 6;

template <int NVX>
struct ClPoll {
  double pre[CL_PRE];
  __device__ __forceinline__ void prefetch(const ClArgs& c, int64_t slot, int lane) {
    const int W = c.W, nflat = NVX * W;
    const double* base = c.xch + slot * (int64_t)(CL_NV * W);
    if (W <= 64) {                 // one lane per workgroup, one load per value (the layout collect() takes directly)
#pragma unroll
      for (int k = 0; k < NVX; ++k)
        pre[k] = (lane < W) ? __hip_atomic_load(base + k * W + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      return;
    }
#pragma unroll
    for (int r = 0; r < CL_PRE; ++r) {
      const int idx = r * 64 + lane;
      pre[r] = (idx < nflat) ? __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
  }
  __device__ __forceinline__ void gave_up(const ClArgs& c, int lane, volatile unsigned* s_dead) {
    if (lane == 0) {
      *s_dead = 1u;
      atomicAdd(c.err, 1u);
      __hip_atomic_store(c.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (c.par) {
        c.par[7] = 1.0;
        c.par[2] = 1.0;            // everything queued behind becomes a no-op
      }
    }
  }
  // totals of the slot (fixed order: slab order within 64-lane rounds); have_pre: prefetch() was called for this slot
  __device__ __forceinline__ void collect(const ClArgs& c, int64_t slot, int lane, bool have_pre, double* s_xv,
                                          double (&tot)[NVX], volatile unsigned* s_dead) {
    const int W = c.W, nflat = NVX * W;
    const double* base = c.xch + slot * (int64_t)(CL_NV * W);
    if (W <= 64) {
      // the everyday case (n up to half a million rows): lane w' holds workgroup w's partial of every value
      double val[NVX];
      bool first = true;
      unsigned spins = 0;
      while (true) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
          if (first && have_pre) val[k] = pre[k];
          else if (first || (unsigned long long)__double_as_longlong(val[k]) == CL_SENT)
            val[k] = (lane < W) ? __hip_atomic_load(base + k * W + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NVX; ++k) ok = ok && (unsigned long long)__double_as_longlong(val[k]) != CL_SENT;
        first = false;
        if (__all(ok ? 1 : 0) || *s_dead != 0u) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > c.spin_limit) {
          gave_up(c, lane, s_dead);
          break;
        }
      }
#pragma unroll
      for (int k = 0; k < NVX; ++k) tot[k] = wave_sum(val[k]);
      return;
    }
    for (int g0 = 0; g0 < nflat; g0 += 64 * CL_RND) {
      double val[CL_RND];
      bool first = true;
      unsigned spins = 0;
      while (true) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < CL_RND; ++r) {
          const int idx = g0 + r * 64 + lane;
          if (idx < nflat) {
            if (first && have_pre && g0 == 0 && r < CL_PRE) val[r] = pre[r];
            else if (first || (unsigned long long)__double_as_longlong(val[r]) == CL_SENT)
              val[r] = __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            val[r] = 0.0;
          }
        }
#pragma unroll
        for (int r = 0; r < CL_RND; ++r) ok = ok && (unsigned long long)__double_as_longlong(val[r]) != CL_SENT;
        first = false;
        if (__all(ok ? 1 : 0) || *s_dead != 0u) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > c.spin_limit) {
          gave_up(c, lane, s_dead);
          break;
        }
      }
#pragma unroll
      for (int r = 0; r < CL_RND; ++r) {
        const int idx = g0 + r * 64 + lane;
        if (idx < nflat) s_xv[idx] = val[r];
      }
    }
    // (one wave: the LDS writes above are visible to its own reads below in program order)
#pragma unroll
    for (int k = 0; k < NVX; ++k) {
      double t = 0.0;
      for (int w0 = 0; w0 < W; w0 += 64) t += wave_sum((w0 + lane < W) ? s_xv[k * W + w0 + lane] : 0.0);
      tot[k] = t;
    }
  }
};

// R rows per thread, TMAX threads.  DED: TMAX / 64 - 1 COMPUTE waves hold the columns and one dedicated EXCHANGE wave
// (the last) publishes the workgroup's partial sums and collects the cluster's totals while the others work (measured
// at n = 20 000, three workgroups per column: sweep 4.6 -> 5.6 TB/s against wave 0 doing both).  !DED: every wave
// computes and wave 0 also does the exchange -- 8192 instead of 7168 rows per workgroup, which pays where it saves a
// workgroup per column (n = 16 384: two instead of three) or lets a second cluster run (n = 10^6: 2 x 128 workgroups).
template <int R, int TMAX, int D, int L, bool DED, bool STATS, bool DEFL, bool SWEEP, bool CEN>
__global__ __launch_bounds__(TMAX, TMAX / 256) void k_colpass_cl(ClArgs c) {
  static_assert(L >= 1 && L < D, "lookahead");
  static_assert(!STATS || (SWEEP && !DEFL && !CEN), "the statistics pass carries the first sweep");
  constexpr int NV = STATS ? 3 : ((DEFL && SWEEP) ? 2 : 1);
  PassArgs& a = c.a;
  if (a.dpar) {
    if (a.dpar[2] != 0.0) return;
    a.alpha = uniform_f64(a.dpar[0]);
    a.inv_tt = uniform_f64(a.dpar[1]);
  }
  constexpr int Q = 4;                       // ring of per-column staging slots in LDS
  __shared__ double s_part[Q][TMAX / 64][4];
  __shared__ double s_tot[Q][4];
  __shared__ unsigned s_seq[TMAX / 64];      // DED: columns whose partial sums compute wave q has left in s_part
  __shared__ unsigned s_rdy;                 // DED: columns whose totals the exchange wave has left in s_tot
  __shared__ double s_out[4][64];
  __shared__ double s_go[2][64];
  __shared__ double s_xv[3 * 256];
  __shared__ unsigned s_dead;
  extern __shared__ double2 s_dyn[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = (int)blockDim.x >> 6;
  const int ncw = DED ? nw - 1 : nw, T = ncw * 64;                     // compute waves / threads
  constexpr int TC = DED ? TMAX - 64 : TMAX;                           // row-pair stride of the vectors in LDS
  const bool xw = DED ? wv == ncw : wv == 0;                           // this wave does the exchange
  const bool cw = DED ? !xw : true;                                    // this wave holds rows of the columns
  const int W = c.W;
  const int cluster = (int)blockIdx.x / W, w = (int)blockIdx.x - cluster * W;
  const int64_t n = a.n, ld = a.ld;
  const int64_t slab0 = (int64_t)w * c.S;
  const int64_t ns = (n - slab0 < c.S) ? n - slab0 : c.S;             // rows of this workgroup (>= 1)
  RowMap<R> rm;
  {
    const int64_t ld_slab = (ns + 1) & ~(int64_t)1;                    // never load beyond the slab: one read of X
    rm.init(ns, (ld - slab0 < ld_slab) ? ld - slab0 : ld_slab, cw ? tid : 0, T);
  }
  double2* s_v = s_dyn;                                                // [R/2][TC] row pairs of the sweep vector
  double2* s_ts = s_dyn + (SWEEP ? (R / 2) * TC : 0);                  // ... of the score vector
  if (tid == 0) {
    s_dead = 0u;
    s_rdy = 0u;
  }
  if (tid < TMAX / 64) s_seq[tid] = 0u;
  unsigned cbase = 0u;                                                 // columns of the chunks walked so far
  if (cw) {
    if constexpr (SWEEP) {
      double tmp[R];
      load_vec<R>(tmp, a.v + slab0, ns, tid, T);
#pragma unroll
      for (int k = 0; k < R / 2; ++k) s_v[k * TC + tid] = make_double2(tmp[2 * k], tmp[2 * k + 1]);
    }
    if constexpr (DEFL) {
      double tmp[R];
      load_vec<R>(tmp, a.ts + slab0, ns, tid, T);
#pragma unroll
      for (int k = 0; k < R / 2; ++k) s_ts[k * TC + tid] = make_double2(tmp[2 * k], tmp[2 * k + 1]);
    }
  }
  __syncthreads();
  const bool fit = STATS && a.mode == MBPLS_SCALE_FIT;
  const double dn = (double)n;
  const bool lazy_c = a.lazy != 0 && tid < a.kprev;                   // thread i < kprev: component i's correction
  const bool corr = lazy_c && w == 0;                                  // ... entered into the dot products once per cluster
  const double cu_i = (lazy_c && a.cu) ? a.cu[tid] : 0.0;
  const double ce_i = (lazy_c && a.ce) ? a.ce[tid] : 0.0;
  const double* prow = lazy_c ? a.Pm + (int64_t)tid * a.ldp : nullptr;
  const bool outs = !STATS && a.ir2 != nullptr && a.rs != nullptr;     // true-space g / p written here
  const bool writer = w == 0 && wv == ((DED || nw == 1) ? 0 : 1);      // the wave that writes per-column outputs
  ClPoll<NV> poll;
#pragma unroll
  for (int r = 0; r < CL_PRE; ++r) poll.pre[r] = 0.0;

  for (int chunk = cluster; chunk < c.nchunks; chunk += c.G) {
    const int64_t j0 = uniform64(a.c0[chunk]), j1 = uniform64(a.c1[chunk]);
    const int64_t ncols = j1 - j0;
    double acc[SWEEP ? R : 1];
#pragma unroll
    for (int k = 0; k < (SWEEP ? R : 1); ++k) acc[k] = 0.0;
    double ssq_g = 0.0, ssq_p = 0.0, ssq_x = 0.0, amp = 0.0, dacc = 0.0, dk = 0.0;

    // <ts, v> of the whole column (the sweep of a deflating pass sees X - ts p^T): one un-pipelined exchange
    double tsv = 0.0;
    if constexpr (DEFL && SWEEP) {
      if (a.deflate) {
        if (cw) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < R / 2; ++k) {
            const double2 t2 = s_ts[k * TC + tid], v2 = s_v[k * TC + tid];
            s = fma(t2.x, v2.x, s);
            s = fma(t2.y, v2.y, s);
          }
          s = wave_sum_last_row(s);
          if (lane == 63) s_part[0][wv][0] = s;
        }
        __syncthreads();
        if (xw) {
          const int64_t slot = c.p + chunk;
          if (lane == 0) {
            double t = s_part[0][0][0];
            for (int i = 1; i < ncw; ++i) t += s_part[0][i][0];
            __hip_atomic_store(c.xch + slot * (int64_t)(CL_NV * W) + w, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          ClPoll<1> p1;
          double tot[1];
          p1.collect(c, slot, lane, false, s_xv, tot, &s_dead);
          if (lane == 0) s_tot[0][0] = tot[0];
        }
        __syncthreads();
        tsv = s_tot[0][0];
        __syncthreads();
      }
    }

    double x[D][R];
    double pp[D];                    // P[column, tid] of the column in slot d (0 for threads >= kprev)
    double k0v[STATS ? D : 1];       // statistics pass: the column's first sample (the shift)
    double muv[CEN ? D : 1];         // centring flavour: the column's mean
    double r2v[STATS ? 1 : D];       // raw-space passes: 1 / scale_j^2
    auto fetch = [&](int d, int64_t jj) {
      load_col<R, true>(x[d], a.src + jj * ld + slab0, rm);
      pp[d] = lazy_c ? prow[jj] : 0.0;
      if constexpr (STATS) k0v[d] = a.src[jj * ld];
      if constexpr (CEN) muv[d] = a.mu[jj];
      if constexpr (!STATS) r2v[d] = a.ir2 ? a.ir2[jj] : 1.0;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {
      pp[d] = 0.0;
      if (cw && d < ncols) fetch(d, j0 + d);
    }
    bool have_pre = false;           // exchange wave: loads of the next slot's partial sums are in flight

    // LDS hand-offs between the compute waves and a dedicated exchange wave: plain stores / loads of one LDS word,
    // in program order per wave (the LDS executes a wave's instructions in order); bounded like the global polls.
    volatile unsigned* v_seq = s_seq;
    volatile unsigned* v_rdy = &s_rdy;
    volatile unsigned* v_dead = &s_dead;
    auto give_up = [&]() {
      *v_dead = 1u;
      atomicAdd(c.err, 1u);
      __hip_atomic_store(c.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (c.par) {
        c.par[7] = 1.0;
        c.par[2] = 1.0;
      }
    };
    if (DED && xw) {
      // ---- dedicated exchange wave: its own loop, no workgroup barrier.  Step s: publish column s as soon as every
      // compute wave has left its partial sums, then collect the totals of column s - (L - 1) and release them.
      for (int64_t st = 0; st < ncols + (L - 1); ++st) {
        if (st < ncols) {
          const unsigned want = cbase + (unsigned)st + 1u;
          unsigned spins = 0;
          while (true) {
            const bool ok = lane >= ncw || (int)(v_seq[lane < ncw ? lane : 0] - want) >= 0;
            if (__all(ok ? 1 : 0) || *v_dead != 0u) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 26)) {
              if (lane == 0) give_up();
              break;
            }
          }
          if (lane < NV) {
            const int pb = (int)(st & (Q - 1));
            double t = ((volatile double*)&s_part[pb][0][0])[lane];
            for (int q = 1; q < ncw; ++q) t += ((volatile double*)&s_part[pb][q][0])[lane];
            __hip_atomic_store(c.xch + (j0 + st) * (int64_t)(CL_NV * W) + (int64_t)lane * W + w, t, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        const int64_t cc = st - (L - 1);
        if (cc >= 0) {
          double tot[NV];
          poll.collect(c, j0 + cc, lane, false, s_xv, tot, &s_dead);
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) ((volatile double*)&s_tot[cc & (Q - 1)][0])[k] = tot[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            *v_rdy = cbase + (unsigned)cc + 1u;
          }
        }
      }
    }

    // produce: the partial sums of column ip (ring slot sp) over this workgroup's rows, left in LDS for the exchange
    auto produce = [&](const int sp, const int64_t ip) {
      const int pb = (int)(ip & (DED ? Q - 1 : 1));
      double(&xp)[R] = x[sp];
      double s[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) s[k] = 0.0;
      if constexpr (STATS) {
        const double K0 = fit ? uniform_f64(k0v[sp]) : 0.0;
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
          const double2 v2 = s_v[k * TC + tid];
          const bool r0 = (rm.in_n0 >> k) & 1u, r1 = (rm.in_n1 >> k) & 1u;
          const double e0 = r0 ? xp[2 * k] - K0 : 0.0, e1 = r1 ? xp[2 * k + 1] - K0 : 0.0;
          xp[2 * k] = e0;                  // the shifted column serves the accumulation too
          xp[2 * k + 1] = e1;
          s[0] += e0;
          s[0] += e1;
          s[1] = fma(e0, e0, s[1]);
          s[1] = fma(e1, e1, s[1]);
          s[2] = fma(e0, v2.x, s[2]);
          s[2] = fma(e1, v2.y, s[2]);
        }
      } else {
        const double muj = CEN ? uniform_f64(muv[sp]) : 0.0;
        if (corr) {
          if constexpr (DEFL) s[0] = -pp[sp] * ce_i;
          if constexpr (SWEEP) s[DEFL ? 1 : 0] = -pp[sp] * cu_i;
        }
        // rows beyond the slab hold a re-read of its first pair (clamped offsets): their entries of v / ts are
        // zero and their accumulators are never stored, so nothing is masked here
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
          // (keeps the LDS reads of at most two row pairs in flight: hoisted to the top of the loop they hold
          // 64 registers and the x ring spills)
          if ((DEFL && SWEEP) && (k & 1) == 0 && k) __builtin_amdgcn_sched_barrier(0);
          double e0 = xp[2 * k], e1 = xp[2 * k + 1];
          if constexpr (CEN) {
            e0 -= muj;
            e1 -= muj;
            xp[2 * k] = e0;
            xp[2 * k + 1] = e1;
          }
          if constexpr (DEFL) {
            const double2 t2 = s_ts[k * TC + tid];
            s[0] = fma(e0, t2.x, s[0]);
            s[0] = fma(e1, t2.y, s[0]);
          }
          if constexpr (SWEEP) {
            const double2 v2 = s_v[k * TC + tid];
            s[DEFL ? 1 : 0] = fma(e0, v2.x, s[DEFL ? 1 : 0]);
            s[DEFL ? 1 : 0] = fma(e1, v2.y, s[DEFL ? 1 : 0]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) s[k] = wave_sum_last_row(s[k]);
      if (lane == 63) {
#pragma unroll
        for (int k = 0; k < NV; ++k) ((volatile double*)&s_part[pb][wv][0])[k] = s[k];
        if constexpr (DED) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          v_seq[wv] = cbase + (unsigned)ip + 1u;
        }
      }
    };
    // consume: the totals of column i (ring slot sc) -> weights / loadings, the accumulation, the slot's refill
    auto consume = [&](const int sc, const int64_t i) {
      // ---- compute waves, consume: totals of column i (left in LDS by the exchange wave one iteration ago)
      const int64_t jj = j0 + i;
      double(&xc)[R] = x[sc];
      const int cb = (int)(i & (DED ? Q - 1 : 1));
      if constexpr (DED) {
        const unsigned want = cbase + (unsigned)i + 1u;
        unsigned spins = 0;
        while ((int)(*v_rdy - want) < 0 && *v_dead == 0u) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 26)) {
            if (lane == 0) give_up();
            break;
          }
        }
      }
      const int oslot = (int)(i & 63);
      double cj = 0.0;
      if constexpr (STATS) {
        const volatile double* tq = &s_tot[cb][0];
        const double S1 = tq[0], S2 = tq[1], Sv = tq[2];
        if (fit) {
          const double K0 = uniform_f64(k0v[sc]);
          const double inv_n = 1.0 / dn;
          const double m1 = S1 * inv_n;
          const double mean = K0 + m1;
          const double var = fmax((S2 - S1 * m1) * inv_n, 0.0);
          const double eps = 2.220446049250313e-16;
          const double nme = dn * mean * eps;
          const bool constant = var <= dn * eps * var + nme * nme;
          double rv = __builtin_amdgcn_rcp(var);
          rv = fma(fma(-var, rv, 1.0), rv, rv);
          rv = fma(fma(-var, rv, 1.0), rv, rv);
          const double r2 = constant ? 0.0 : rv;
          const double gt = a.alpha * Sv;             // raw-space weight g~_j
          cj = gt * r2;
          ssq_g = fma(gt, cj, ssq_g);
          amp = fmax(amp, mean * mean * r2);
          ssq_x = fma(dn * var, r2, ssq_x);
          ssq_x += S2 - S2;              // non-finite samples poison the sum (fmax() made their variance 0): colpass.hip
          if (writer) {
            if (lane == 0) {
              s_out[0][oslot] = mean;
              s_out[1][oslot] = var;
              s_out[2][oslot] = r2;
              s_out[3][oslot] = gt;
            }
            if (oslot == 63 || i == ncols - 1) {
              const int64_t jb = jj - oslot;
              if (lane <= oslot) {
                a.mean[jb + lane] = s_out[0][lane];
                a.scale[jb + lane] = s_out[1][lane];
                if (a.r2_out) a.r2_out[jb + lane] = s_out[2][lane];
                if (a.g) a.g[jb + lane] = s_out[3][lane];
              }
            }
          }
        } else {                                      // standardize=False: sums of squares and the sweep only
          const double gj = a.alpha * Sv;
          cj = gj;
          ssq_g = fma(gj, gj, ssq_g);
          ssq_x += S2;
          if (writer) {
            if (lane == 0) s_out[3][oslot] = gj;
            if ((oslot == 63 || i == ncols - 1) && lane <= oslot && a.g) a.g[jj - oslot + lane] = s_out[3][lane];
          }
        }
      } else {
        const volatile double* tq = &s_tot[cb][0];
        const double r2c = uniform_f64(r2v[sc]);
        double gj = 0.0, pj = 0.0, pjk = 0.0;
        if constexpr (DEFL) {
          pj = tq[0] * a.inv_tt;               // raw-space loading p~_j
          ssq_p = fma(pj * pj, r2c, ssq_p);
          if constexpr (SWEEP) {
            const double b = tq[1];
            if (a.deflate) {
              gj = a.alpha * fma(-pj, tsv, b);        // <x - p ts, v> = <x, v> - p <ts, v>
              pjk = pj;
            } else {
              gj = a.alpha * b;
            }
          }
        } else {
          gj = a.alpha * tq[0];
        }
        if constexpr (SWEEP) {
          cj = gj * r2c;
          ssq_g = fma(gj, cj, ssq_g);
          dacc = fma(cj, pp[sc], dacc);
          if constexpr (DEFL) dk = fma(cj, pjk, dk);
        }
        if (writer) {
          if (lane == 0) {
            s_go[0][oslot] = gj;
            s_go[1][oslot] = pj;
          }
          if (oslot == 63 || i == ncols - 1) {
            const int64_t jb = jj - oslot;
            if (lane <= oslot) {
              const double r = outs ? a.rs[jb + lane] : 1.0;
              if (SWEEP && a.g) a.g[jb + lane] = s_go[0][lane] * r;
              if (DEFL && a.pvec) {
                a.pvec[jb + lane] = s_go[1][lane];
                if (outs && a.pvec_true) a.pvec_true[jb + lane] = s_go[1][lane] * r;
              }
            }
          }
        }
      }
      if constexpr (SWEEP) {
#pragma unroll
        for (int k = 0; k < R; ++k) acc[k] = fma(cj, xc[k], acc[k]);
      }
      if (i + D < ncols) fetch(sc, jj + D);
    };

    for (int64_t base = -L; (!DED || cw) && base < ncols; base += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int64_t i = base + d;           // the column consumed in this iteration (from iteration 0 on)
        const int64_t ip = i + L;             // the column whose partial sums are produced: slot d (base + L = 0 mod D)
        if (i >= ncols) break;
        const int sp = d;
        const int sc = (d + D - (L % D)) % D;
        if constexpr (DED) {
          // no barrier: the exchange wave runs its own loop.  Produce first: the hand-off between the workgroups (a
          // store and a load through the fabric under full streaming load) takes about two column times, more than
          // the refill of a slot (consume first measured 5.5 -> 3.6 TB/s at n = 20 000).
          if (ip < ncols) produce(sp, ip);
          if (i >= 0) consume(sc, i);
        } else {
          const int pb = (int)(ip & 1);
          if (ip < ncols) produce(sp, ip);
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
          if (xw) {
            // ---- exchange wave: publish column ip, collect the totals of column i + 1 (consumed in the next iteration;
            // its loads were issued one iteration ago), issue the loads for column i + 2
            if (ip < ncols && lane < NV) {
              double t = s_part[pb][0][lane];
              for (int q = 1; q < ncw; ++q) t += s_part[pb][q][lane];
              __hip_atomic_store(c.xch + (j0 + ip) * (int64_t)(CL_NV * W) + (int64_t)lane * W + w, t, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            }
            if (i + 1 >= 0 && i + 1 < ncols) {
              double tot[NV];
              poll.collect(c, j0 + i + 1, lane, have_pre, s_xv, tot, &s_dead);
              if (lane == 0) {
  #pragma unroll
                for (int k = 0; k < NV; ++k) s_tot[(i + 1) & 1][k] = tot[k];
              }
            }
            have_pre = false;
            if (c.prefetch && L >= 2 && i + 2 >= 0 && i + 2 < ncols) {      // (published by every workgroup about now)
              poll.prefetch(c, j0 + i + 2, lane);
              have_pre = true;
            }
          }
          if (i >= 0) consume(sc, i);
        }
      }
    }

    // ---- the chunk's outputs: this workgroup's rows of the partial score vector, the scalars from workgroup 0
    if constexpr (SWEEP) {
      if (cw) {
        double* out = a.part_t + (int64_t)chunk * n + slab0;
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));     // (the store addresses are computed here, not carried through the loop)
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
          const int64_t r0 = 2 * ((int64_t)k * T + tid_e);
          if (r0 + 1 < ns) *reinterpret_cast<double2*>(out + r0) = make_double2(acc[2 * k], acc[2 * k + 1]);
          else if (r0 < ns) out[r0] = acc[2 * k];
        }
      }
    }
    if (w == 0) {
      if (tid == 0) {
        double* ps = a.part_s + (int64_t)chunk * 4;
        ps[0] = ssq_g;
        ps[1] = ssq_p;
        ps[2] = ssq_x;
        ps[3] = amp;
      }
      if (a.lazy && a.part_d) {
        double* pd = a.part_d + (int64_t)chunk * KD;
        if (lazy_c) pd[tid] = dacc;
        if (tid == 0) pd[a.kprev] = dk;
      }
    }
    cbase += (unsigned)ncols;
    __syncthreads();          // LDS staging (s_part / s_tot / s_go) is reused by the next chunk
  }
}

// ------------------------------------------------------------------------------------------------- host side

struct ClPlan {
  int W, G, T;
  int64_t S;
  bool ded;      // dedicated exchange wave (7 compute waves) or wave 0 doing both (8 compute waves)
};

constexpr int CL_R = 16;

// How many workgroups of a cluster pass the device holds AT ONCE: the members of a cluster wait for each other inside
// the launch, so a pass may never be larger than that.  Asked from the runtime for the fattest instantiation (512
// threads, 128 KB of dynamic LDS) instead of assumed from the CU count; 0 when the device is partitioned by a CU mask
// the occupancy query does not see (the passes then stay with the two-phase kernels of colpass_long.hip).  The launch
// itself is cooperative (cl_launch): the runtime refuses a grid it cannot make co-resident, and on top of that every
// wait inside the kernel is bounded -- a pass that shared the device with another tenant's resident workgroups gives
// up, flags it in host-mapped memory, and the host re-runs the fit on the two-phase passes (MBPLS._fit).
static int cl_capacity(mbpls_ctx* ctx) {
  if (ctx->cluster_cap >= 0) return ctx->cluster_cap;
  int cap = 0;
  if (!getenv("HSA_CU_MASK") && !getenv("ROC_GLOBAL_CU_MASK")) {
    auto fn = k_colpass_cl<16, CL_TMAX, 3, 2, false, false, true, true, false>;
    const size_t lds = 2 * CL_TMAX * 8 * 16;
    int per_cu = 0;
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, CL_TMAX, lds) == hipSuccess)
      cap = std::max(0, std::min(per_cu, 1)) * ctx->cus;
    else
      (void)hipGetLastError();
  }
  ctx->cluster_cap = cap;
  return cap;
}

static bool cl_plan_for(const mbpls_ctx* ctx, int64_t n, bool ded, ClPlan* pl) {
  const int tc = ded ? CL_TMAX - 64 : CL_TMAX;
  const int64_t slab = std::max<int64_t>(128, std::min<int64_t>(ctx->cluster_slab_rows, (int64_t)CL_R * tc)) / 128 * 128;
  const int cap = ctx->cluster_cap;                           // co-resident workgroups (one per CU: the vectors fill the LDS)
  const int64_t wmin = (n + slab - 1) / slab;
  if (wmin > cap || wmin > 256 || wmin < 1) return false;
  pl->ded = ded;
  pl->G = (int)(cap / wmin);
  const int64_t wuse = cap / pl->G;                           // spread the rows over every CU the clusters can use
  pl->S = ((n + wuse - 1) / wuse + 127) / 128 * 128;
  pl->W = (int)((n + pl->S - 1) / pl->S);
  pl->T = (int)(((pl->S + CL_R - 1) / CL_R + 63) / 64 * 64) + (ded ? 64 : 0);
  return pl->T <= CL_TMAX;
}

// Dedicated exchange wave unless giving its rows to an eighth compute wave saves a workgroup per column or lets
// another cluster run beside the others (ctx->cluster_exchange: 0 = this rule, 1 = always dedicated, 2 = never).
static bool cl_plan(const mbpls_ctx* ctx, int64_t n, ClPlan* pl) {
  if (cl_capacity(const_cast<mbpls_ctx*>(ctx)) <= 0) return false;
  ClPlan d, s;
  const bool okd = cl_plan_for(ctx, n, true, &d), oks = cl_plan_for(ctx, n, false, &s);
  if (!okd && !oks) return false;
  bool ded = okd;
  if (okd && oks && (s.W < d.W || s.G > d.G)) ded = false;
  if (ctx->cluster_exchange == 1 && okd) ded = true;
  if (ctx->cluster_exchange == 2 && oks) ded = false;
  *pl = ded ? d : s;
  return true;
}

bool cluster_columns(const mbpls_ctx* ctx, int64_t n) {
  ClPlan pl;
  return ctx->cluster_passes && long_columns(ctx, n) && cl_plan(ctx, n, &pl);
}

// exactly min(G, p) chunks when there are at most G blocks (a cluster walks ONE chunk: G + 1 chunks would double the
// time of the pass), one chunk per block otherwise
static int cl_chunks(mbpls_ctx* ctx, mbpls_mat* m, int G) {
  if (m->cluster_G == G && m->cluster.nchunks) return MBPLS_OK;
  free_chunks(&m->cluster);
  m->cluster_G = 0;
  const std::vector<int64_t>& offs = m->block_offsets;
  const int B = m->nblocks;
  std::vector<int64_t> nb(B, 0);
  int64_t used = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t pb = offs[b + 1] - offs[b];
    if (pb > 0) nb[b] = 1;
    used += nb[b];
  }
  // hand the remaining clusters to the block with the widest chunks, one at a time
  while (used < G) {
    int best = -1;
    double width = 1.0;
    for (int b = 0; b < B; ++b) {
      const int64_t pb = offs[b + 1] - offs[b];
      if (nb[b] == 0 || nb[b] >= pb) continue;
      const double wd = (double)pb / (double)nb[b];
      if (wd > width) {
        width = wd;
        best = b;
      }
    }
    if (best < 0) break;
    nb[best] += 1;
    used += 1;
  }
  ChunkTable* t = &m->cluster;
  t->h_c0.clear(); t->h_c1.clear(); t->h_blk.clear();
  t->h_blk_start.assign(B + 1, 0);
  for (int b = 0; b < B; ++b) {
    t->h_blk_start[b] = (int)t->h_c0.size();
    const int64_t pb = offs[b + 1] - offs[b];
    if (pb == 0) continue;
    const int64_t base = pb / nb[b], rem = pb % nb[b];
    int64_t col = offs[b];
    for (int64_t k = 0; k < nb[b]; ++k) {
      const int64_t wd = base + (k < rem ? 1 : 0);
      t->h_c0.push_back(col);
      t->h_c1.push_back(col + wd);
      t->h_blk.push_back(b);
      col += wd;
    }
  }
  t->h_blk_start[B] = (int)t->h_c0.size();
  t->nchunks = (int)t->h_c0.size();
  MB_HIP(hipMalloc(&t->d_c0, sizeof(int64_t) * t->nchunks));
  MB_HIP(hipMalloc(&t->d_c1, sizeof(int64_t) * t->nchunks));
  MB_HIP(hipMalloc(&t->d_blk_start, sizeof(int) * (B + 1)));
  MB_HIP(hipMemcpyAsync(t->d_c0, t->h_c0.data(), sizeof(int64_t) * t->nchunks, hipMemcpyHostToDevice, ctx->stream));
  MB_HIP(hipMemcpyAsync(t->d_c1, t->h_c1.data(), sizeof(int64_t) * t->nchunks, hipMemcpyHostToDevice, ctx->stream));
  MB_HIP(hipMemcpyAsync(t->d_blk_start, t->h_blk_start.data(), sizeof(int) * (B + 1), hipMemcpyHostToDevice, ctx->stream));
  MB_SYNC_STREAM(ctx);
  m->cluster_G = G;
  return MBPLS_OK;
}

static int cl_exchange_buffer(mbpls_ctx* ctx, size_t doubles, double** xch, unsigned** err) {
  const size_t need = 256 + sizeof(double) * doubles;
  if (ctx->cl_ws_bytes < need) {
    if (ctx->cl_ws) {
      MB_SYNC_STREAM(ctx);
      (void)hipFree(ctx->cl_ws);
      ctx->cl_ws = nullptr;
      ctx->cl_ws_bytes = 0;
    }
    const size_t grown = need + need / 4;
    MB_HIP(hipExtMallocWithFlags(&ctx->cl_ws, grown, hipDeviceMallocFinegrained));
    ctx->cl_ws_bytes = grown;
    MB_HIP(hipMemsetAsync(ctx->cl_ws, 0, 256, ctx->stream));
  }
  if (!ctx->cl_flag_h) {
    MB_HIP(hipHostMalloc((void**)&ctx->cl_flag_h, 64, hipHostMallocMapped));
    *ctx->cl_flag_h = 0u;
  }
  *err = (unsigned*)ctx->cl_ws;
  *xch = (double*)((char*)ctx->cl_ws + 256);
  return MBPLS_OK;
}

template <bool STATS, bool DEFL, bool SWEEP>
static int cl_launch(mbpls_ctx* ctx, const mbpls_mat* mc, PassArgs a, bool cen) {
  mbpls_mat* m = const_cast<mbpls_mat*>(mc);
  ClPlan pl;
  if (!cl_plan(ctx, m->n, &pl)) {
    set_error("cluster passes: n=%lld does not fit %d workgroups", (long long)m->n, ctx->cus);
    return MBPLS_ERR_UNSUPPORTED;
  }
  MB_TRY(cl_chunks(ctx, m, pl.G));
  const ChunkTable& ct = m->cluster;
  const bool writes = false;
  const int kind = STATS ? MBPLS_K_STANDARDIZE : (DEFL ? (SWEEP ? MBPLS_K_DEFLATE_SWEEP : MBPLS_K_DEFLATE) : MBPLS_K_SWEEP);
  // partial outputs: [nchunks][n] score partials, [nchunks][4] scalars, [nchunks][KD] lazy-deflation sums
  const size_t nt = (size_t)ct.nchunks * (size_t)m->n, nsc = (size_t)ct.nchunks * 4, nd = (size_t)ct.nchunks * KD;
  MB_TRY(ensure_ws(ctx, sizeof(double) * (nt + nsc + nd)));
  a.part_t = (double*)ctx->ws;
  a.part_s = a.part_t + nt;
  a.part_d = a.part_s + nsc;
  a.c0 = ct.d_c0;
  a.c1 = ct.d_c1;
  a.dpar = ctx->dpar;
  a.var_out = ctx->var_out;
  if (!STATS && a.ir2 && ctx->ts_rscale) {       // raw-space pass: true-space g / p written by the kernel itself
    a.rs = ctx->ts_rscale;
    a.pvec_true = DEFL ? ctx->ts_p_true : nullptr;
  }
  ClArgs c{};
  c.a = a;
  c.W = pl.W; c.G = std::min(pl.G, ct.nchunks); c.nchunks = ct.nchunks; c.S = pl.S; c.p = m->p;
  c.spin_limit = ctx->cluster_spin_limit;
  c.prefetch = ctx->cluster_prefetch;
  c.par = const_cast<double*>(ctx->dpar);
  const size_t slots = (size_t)m->p + (size_t)ct.nchunks;
  const size_t xd = slots * (size_t)CL_NV * (size_t)pl.W;
  MB_TRY(cl_exchange_buffer(ctx, xd, &c.xch, &c.err));
  MB_HIP(hipHostGetDevicePointer((void**)&c.flag, ctx->cl_flag_h, 0));
  ProfScope prof(ctx, kind, 8.0 * (double)m->n * (double)m->p * (writes ? 2.0 : 1.0));
  hipLaunchKernelGGL(k_cl_fill, dim3((unsigned)((xd / 2 + 256) / 256)), dim3(256), 0, ctx->stream, (int64_t)xd,
                     (unsigned long long*)c.xch, ctx->dpar);
  MB_HIP(hipGetLastError());
  const int tc = pl.ded ? CL_TMAX - 64 : CL_TMAX;
  const unsigned dyn = (unsigned)((SWEEP ? 1 : 0) + (DEFL ? 1 : 0)) * (unsigned)tc * (unsigned)(CL_R / 2) * 16u;
  dim3 grid((unsigned)(c.G * c.W)), block((unsigned)pl.T);
#define MB_CL(DED_, CEN_)                                                                                   \
  do {                                                                                                      \
    auto fn = k_colpass_cl<CL_R, CL_TMAX, 3, 2, DED_, STATS, DEFL, SWEEP, CEN_>;                            \
    MB_ALLOW_LDS(ctx, fn, 2 * CL_TMAX * (CL_R / 2) * 16);                                                   \
    void* args[] = {(void*)&c};                                                                             \
    launched = hipLaunchCooperativeKernel((const void*)fn, grid, block, args, dyn, ctx->stream);            \
  } while (0)
#define MB_CL_CEN(DED_)                           \
  do {                                            \
    if constexpr (STATS) {                        \
      MB_CL(DED_, false);                         \
    } else {                                      \
      if (cen) MB_CL(DED_, true);                 \
      else MB_CL(DED_, false);                    \
    }                                             \
  } while (0)
  hipError_t launched = hipSuccess;
  if (pl.ded) MB_CL_CEN(true);
  else MB_CL_CEN(false);
#undef MB_CL_CEN
#undef MB_CL
  if (launched != hipSuccess) {
    // (hipErrorCooperativeLaunchTooLarge: the device holds fewer workgroups than the plan assumed -- a partition or
    // another limit the occupancy query did not see; the caller falls back to the two-phase passes)
    (void)hipGetLastError();
    set_error("cluster pass: cooperative launch of %u workgroups refused: %s", grid.x, hipGetErrorString(launched));
    ctx->cluster_cap = 0;
    return MBPLS_ERR_UNSUPPORTED;
  }
  return MBPLS_OK;
}

static int cl_reduce(mbpls_ctx* ctx, const mbpls_mat* m, double* d_t, double* d_ssq_g, double* d_ssq_p,
                     double* d_ssq_x, int nd, double* d_d, double* d_amp) {
  const ChunkTable& ct = m->cluster;
  const size_t nt = (size_t)ct.nchunks * (size_t)m->n, nsc = (size_t)ct.nchunks * 4;
  const double* part_t = (const double*)ctx->ws;
  const double* part_s = part_t + nt;
  const double* part_d = part_s + nsc;
  return reduce_chunks(ctx, m, part_t, part_s, d_t, d_ssq_g, d_ssq_p, d_ssq_x, part_d, nd, nd ? d_d : nullptr, d_amp,
                       &ct);
}

int cluster_sweep_lazy(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                       const double* d_P, int64_t ldp, int kprev, const double* d_cu, double* d_d, const double* d_r2,
                       const double* d_mean) {
  mbpls_ctx* ctx = m->ctx;
  PassArgs a{};
  a.src = m->data; a.dst = nullptr; a.ld = m->ld; a.n = m->n;
  a.v = d_v; a.alpha = alpha; a.g = d_g;
  a.lazy = 1; a.kprev = kprev; a.Pm = d_P; a.ldp = ldp; a.cu = d_cu; a.ce = nullptr;
  a.ir2 = d_r2; a.mu = d_mean;
  MB_TRY((cl_launch<false, false, true>(ctx, m, a, d_r2 && d_mean)));
  return cl_reduce(ctx, m, d_t, d_ssq_g, nullptr, nullptr, kprev, d_d, nullptr);
}

int cluster_loadings_sweep(const mbpls_mat* m, const double* d_ts, double inv_tt, int deflate, double* d_p,
                           double* d_ssq_p, const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                           const double* d_P, int64_t ldp, int kprev, const double* d_cu, const double* d_ce,
                           double* d_d, const double* d_r2, const double* d_mean) {
  mbpls_ctx* ctx = m->ctx;
  PassArgs a{};
  a.src = m->data; a.dst = nullptr; a.ld = m->ld; a.n = m->n;
  a.ts = d_ts; a.inv_tt = inv_tt; a.deflate = (d_v && deflate) ? 1 : 0; a.pvec = d_p;
  a.v = d_v; a.alpha = alpha; a.g = d_g;
  a.lazy = 1; a.kprev = kprev; a.Pm = d_P; a.ldp = ldp; a.cu = d_cu; a.ce = d_ce;
  a.ir2 = d_r2; a.mu = d_mean;
  const bool cen = d_r2 && d_mean;
  if (d_v) {
    MB_TRY((cl_launch<false, true, true>(ctx, m, a, cen)));
    const int nd = d_d ? kprev + (a.deflate ? 1 : 0) : 0;
    return cl_reduce(ctx, m, d_t, d_ssq_g, d_ssq_p, nullptr, nd, d_d, nullptr);
  }
  MB_TRY((cl_launch<false, true, false>(ctx, m, a, cen)));
  if (d_ssq_p) return cl_reduce(ctx, m, nullptr, nullptr, d_ssq_p, nullptr, 0, nullptr, nullptr);
  return MBPLS_OK;
}

int cluster_standardize_stats_sweep(const mbpls_mat* src, int mode, double* d_mean, double* d_scale, double* d_r2,
                                    double* d_ssq_x, double* d_amp, const double* d_v, double alpha, double* d_g,
                                    double* d_t, double* d_ssq_g) {
  mbpls_ctx* ctx = src->ctx;
  PassArgs a{};
  a.src = src->data; a.dst = nullptr; a.ld = src->ld; a.n = src->n;
  a.mode = mode; a.mean = d_mean; a.scale = d_scale; a.r2_out = d_r2;
  a.v = d_v; a.alpha = alpha; a.g = d_g;
  MB_TRY((cl_launch<true, false, true>(ctx, src, a, false)));
  return MBPLS_OK;
}

int cluster_stats_reduce(const mbpls_mat* src, double* d_t, double* d_ssq_g, double* d_ssq_x, double* d_amp) {
  return cl_reduce(src->ctx, src, d_t, d_ssq_g, nullptr, d_ssq_x, 0, nullptr, d_amp);
}

}  // namespace mbpls

using namespace mbpls;

extern "C" int mbpls_hip_cluster_columns(mbpls_ctx* ctx, int64_t n, int* yes) {
  MB_ARG(ctx && yes, "NULL argument");
  *yes = cluster_columns(ctx, n) ? 1 : 0;
  return MBPLS_OK;
}

extern "C" int mbpls_hip_cluster_status(mbpls_ctx* ctx, int* timeouts) {
  MB_ARG(ctx && timeouts, "NULL argument");
  *timeouts = 0;
  if (!ctx->cl_ws || !ctx->cl_flag_h) return MBPLS_OK;
  MB_SYNC_STREAM(ctx);
  if (__atomic_load_n(ctx->cl_flag_h, __ATOMIC_ACQUIRE) == 0u) return MBPLS_OK;   // the everyday answer: no copy at all
  unsigned v = 0;
  MB_HIP(hipMemcpyAsync(&v, ctx->cl_ws, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  MB_SYNC_STREAM(ctx);
  MB_HIP(hipMemsetAsync(ctx->cl_ws, 0, sizeof(unsigned), ctx->stream));
  __atomic_store_n(ctx->cl_flag_h, 0u, __ATOMIC_RELEASE);
  *timeouts = (int)std::max(1u, v);
  return MBPLS_OK;
}

namespace mbpls {
__global__ void k_cl_status_into(const unsigned* __restrict__ err, double* __restrict__ out) {
  *out = err ? (double)*err : 0.0;
}
}  // namespace mbpls

// Column shards: whether a fit is run again after a hand-off time-out must be decided by ALL ranks alike (the re-run
// issues collectives).  The count of this context's time-outs so far goes into a word of the exchange buffer on the
// stream, in front of the all-reduce of that buffer: every rank then reads the SUM over the ranks behind the pass
// and raises -- or does not -- together.  Does not reset the count (mbpls_hip_cluster_status does, on each rank).
extern "C" int mbpls_hip_cluster_status_into(mbpls_ctx* ctx, double* d_word) {
  MB_ARG(ctx && d_word, "NULL argument");
  hipLaunchKernelGGL(k_cl_status_into, dim3(1), dim3(1), 0, ctx->stream, (const unsigned*)ctx->cl_ws, d_word);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

// one kernel of this translation unit, for mbpls_hip_warmup (its code object is loaded on first use otherwise)
namespace mbpls {
const void* anchor_colpass_cluster() { return (const void*)&k_cl_fill; }
}  // namespace mbpls
