// Device-resident NIPALS step: everything the reference does between two passes over X (mbpls.py:860-913 -- block
// weights' norms, block scores, super weights, super score, the stop rule, Y loadings, next u) plus the bookkeeping of
// lazy deflation and implicit standardisation, in ONE small kernel that reads the exchange buffer of the pass that
// just finished and writes the inputs of the pass that follows.  With it a one-response fit is queued as
//   stats+sweep, step, [sweep, step, loadings+sweep, step] x K
// without a single host round trip (the passes take alpha / inv_tt / a skip flag from the same device block,
// ctx->dpar): the host reads the results once, at the end.  When the column shards of several GPUs take part, the
// all-reduce of the exchange buffer (rccl_comm.hip) sits on the same stream between the pass and the step.
//
// One response converges in exactly two iterations per component (u = Y v / ||Y v|| does not depend on ts), so the
// queue is built SPECULATIVELY for two; a step that finds ||ts_old - ts|| above the tolerance raises the skip flag,
// everything queued behind it becomes a no-op, and the host resumes that component with its ordinary loop.
//
// All of it is n x (B + q + k) work in one 1024-thread workgroup, L2-resident, latency-bound: ~20 us against the
// ~170 us a blocking D2H + numpy + H2D round trip costs per pass.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace mbpls {

constexpr int RS_BMAX = MBPLS_STEP_MAX_BLOCKS;
constexpr int RS_QMAX = MBPLS_STEP_MAX_RESPONSES;
constexpr int RS_KMAX = 40;   // == KD of colpass.hip (lazily deflated components + 1)

// Long score vectors (tall data, n up to millions): the step runs on G workgroups, each owning a fixed set of rows
// (thread t of workgroup g: rows g*1024 + t + k*G*1024 in EVERY loop, so a thread only ever re-reads rows it wrote
// itself); the reductions become two-level -- per-workgroup totals through HBM, a grid barrier, every workgroup adds the
// G partials in the same order and so takes the same branches.  One workgroup (n up to a few ten thousand) needs none
// of it.  The G <= 128 workgroups of 1024 threads are co-resident on the 256 CUs (nothing else runs on the stream).
// workgroups of a step: one per STEP_ROWS_PER_WG rows, co-resident (at most one per CU).  Measured at n = 10^6:
// 355 us per step with 61 or 128 workgroups, 790 us with 244 (barrier contention); 334 us once the barrier stopped
// fencing the L2 -- what remains is the latency of ~ten dependent row loops (16 rows per thread) plus ten barriers.
constexpr int STEP_MAX_WG = 128;
constexpr int STEP_ROWS_PER_WG = 16384;

struct StepGrid {
  int G;                  // workgroups
  unsigned* bar;          // grid barrier counter, zeroed before the launch
  double* gpart;          // [2][G][16] per-workgroup totals, two alternating sets
  double* par;            // par[7] = 1: a barrier timed out (the host finishes the fit with its own loop)
  unsigned spin_limit;    // polls of the counter before a barrier gives up
  unsigned* dead;         // LDS word of the workgroup (set by the kernel): a barrier of this launch gave up
};

__device__ __forceinline__ void grid_barrier(const StepGrid& sg, unsigned& epoch) {
  // The only data that crosses workgroups are the per-workgroup totals and this counter.  They live in FINE-GRAINED
  // device memory (uncached: hipDeviceMallocFinegrained).  The totals are stored by threads tid < m, then
  // __syncthreads(), then thread 0 signals with a RELEASE increment; a waiter spins on relaxed loads and finishes with
  // one ACQUIRE load before anybody reads the totals -- the formally required ordering across XCDs (a stale partial
  // would silently change ts and the stop decision, possibly differently on different ranks).  Costs the L2
  // write-back / invalidate of agent-scope release / acquire per barrier (334 -> 355 us per step at n = 10^6, ten
  // barriers, against 5 ms passes).
  // The launch is cooperative (all workgroups co-resident by construction); the poll limit only keeps a broken queue
  // from hanging the device: once a barrier gave up, the later ones of the launch return at once (epoch bit 31).
  const bool dead = *sg.dead != 0u;     // (written before the previous barrier's second __syncthreads: uniform)
  epoch += 1;               // every thread counts: the alternating partial sets are chosen from it
  __syncthreads();
  if (threadIdx.x == 0 && !dead) {
    const unsigned target = epoch * (unsigned)sg.G;
    __hip_atomic_fetch_add(sg.bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(sg.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > sg.spin_limit) {
        sg.par[7] = 1.0;
        sg.par[2] = 1.0;                 // everything queued behind becomes a no-op
        *sg.dead = 1u;
        break;
      }
    }
    (void)__hip_atomic_load(sg.bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}

// reduce_many over the whole grid: the first m values of v summed over every thread of every workgroup, fixed order
template <int M>
__device__ __forceinline__ void grid_sum(double (&v)[M], int m, double* part, double* tot, const StepGrid& sg,
                                         unsigned& epoch) {
  reduce_many<M>(v, m, part, tot);
  if (sg.G == 1) return;
  const int tid = threadIdx.x;
  double* mine = sg.gpart + ((size_t)(epoch & 1u) * sg.G + blockIdx.x) * 16;
  const double* all = sg.gpart + (size_t)(epoch & 1u) * sg.G * 16;
  if (tid < m) __hip_atomic_store(mine + tid, tot[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  grid_barrier(sg, epoch);
  if (tid < m) {
    double s = 0.0;
    for (int g = 0; g < sg.G; ++g) s += __hip_atomic_load(all + (size_t)g * 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot[tid] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < M; ++k)
    if (k < m) v[k] = tot[k];
  __syncthreads();           // tot[] is reused by the next reduction
}

// BM / QM: compile-time bounds of the per-block / per-response register arrays (the common few-blocks, one-response
// case must not carry 6 x 16 doubles per thread: at 1024 threads that is the whole register file)
template <int BM, int QM>
__global__ __launch_bounds__(WG1024) void k_nipals_step(mbpls_nipals_step s, StepGrid sg) {
  __shared__ double part[16 * 16];       // wave partials of reduce_many
  __shared__ double tot[16];
  __shared__ double s_mx[WG1024 / 64], s_mn[WG1024 / 64];
  __shared__ unsigned s_dead;
  double* par = s.par;
  if (s.host_seq && threadIdx.x == 0 && blockIdx.x == 0)
    __hip_atomic_store(s.host_seq, s.seq_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (par[2] != 0.0) return;             // an earlier component needed more iterations than were queued
  const int tid = threadIdx.x;
  if (tid == 0) s_dead = 0u;
  sg.dead = &s_dead;
  __syncthreads();
  const bool lead = tid == 0 && blockIdx.x == 0;       // the one thread that records scalars
  const int64_t gtid = (int64_t)blockIdx.x * WG1024 + tid, gstride = (int64_t)sg.G * WG1024;
  unsigned epoch = 0;
  const int64_t n = s.n;
  const int B = s.nblocks, q = s.q, kc = s.kc, nd = s.nd;
  const bool istd = s.centre != 0;
  // mode 1: UNIPALS, p > n, one response (mbpls.py:481-572): the pass delivered X_b (X_b^T y) and ||X_b^T y||^2; a
  // component needs no iteration: ts = X (X^T y) / ||.|| (:489-493 with one response), w = X^T u = c X^T y (:503)
  const bool uni = s.mode == 1;
  const double* tr = s.xchg;             // [B][n] raw accumulations of the pass
  const double* ssq_g = s.xchg + n * B;
  const double* ssq_p = ssq_g + B;
  const double* ssq_x = ssq_g + 2 * B;
  const double* dd = ssq_g + 3 * B;      // [B][nd] lazy-deflation sums
  // The n-vectors of a step are distinct buffers: the few-blocks instantiation says so (__restrict__) and unrolls its
  // row loops four times, so several rows' loads are in flight (238 -> 221 us per step at n = 10^6); the general one
  // has no registers to spare (restrict alone took its spills from 28 to 70).
  constexpr int UNR = (BM <= 4 && QM == 1) ? 4 : 1;
  using cptr = std::conditional_t<(UNR > 1), const double* __restrict__, const double*>;
  using mptr = std::conditional_t<(UNR > 1), double* __restrict__, double*>;
  mptr tb = s.Tb + (int64_t)kc * B * n;   // block scores of this component: scratch now, final at convergence
  cptr r_tr = tr;
  cptr r_Tsc = s.Tsc;
  cptr r_Y = s.Y;
  mptr r_u = s.u_cur;
  mptr r_vp = s.vpass;
  mptr r_to = s.ts_old;
  mptr r_tp = s.tspass;

  if (s.run == 1 && kc == 0 && s.centre) {
    // the statistics pass left max_j (mean_j/scale_j)^2 per block where the lazy sums of later passes go: raw-space
    // dot products without centring are only good for modestly offset data -- hand back to the caller otherwise
    // (the measure itself goes back with every fit: a caller that started with centring passes learns from it when the
    // data no longer need them)
    double worst = 0.0;
    for (int b = 0; b < B; ++b) worst = fmax(worst, dd[b]);
    if (lead) par[6] = worst;
    if (s.amp2_max > 0.0 && !(worst <= s.amp2_max)) {
      if (lead) {
        par[2] = 1.0;
        par[3] = 0.0;
        par[5] = 1.0;      // reason: offset data, queue again with the centring passes
        par[6] = worst;
      }
      return;
    }
  }
  if (s.run == 1 && tid < B && blockIdx.x == 0) {
    if (kc == 0) s.ssq_x[tid] = ssq_x[tid];
    else s.scal[(int64_t)(kc - 1) * 4 * B + 2 * B + tid] = ssq_p[tid];   // ||p_b||^2 of the component just deflated
  }

  // ---- t~_b = A_b - Ts d_b (lazy deflation), its mean (implicit standardisation), t~_b^T u, u^T u
  double sum_t[BM], tu[BM], mu[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int b = 0; b < BM; ++b) sum_t[b] = tu[b] = 0.0;
#pragma unroll UNR
  for (int64_t i = gtid; i < n; i += gstride) {
    const double ui = r_u[i];
    mu[0] = fma(ui, ui, mu[0]);
    mu[1] += ui;
#pragma unroll
    for (int b = 0; b < BM; ++b)
      if (b < B) {
        double t = r_tr[(int64_t)b * n + i];
        for (int j = 0; j < nd; ++j) t = fma(-r_Tsc[(int64_t)j * n + i], dd[b * nd + j], t);
        tb[(int64_t)b * n + i] = t;
        sum_t[b] += t;
        tu[b] = fma(t, ui, tu[b]);
      }
  }
  if constexpr (2 * BM + 2 <= 16) {
    // few blocks: the three reductions as one (a grid-wide reduction is a barrier on every workgroup)
    double c3[2 * BM + 2];
#pragma unroll
    for (int b = 0; b < BM; ++b) {
      c3[b] = sum_t[b];
      c3[BM + b] = tu[b];
    }
    c3[2 * BM] = mu[0];
    c3[2 * BM + 1] = mu[1];
    grid_sum<2 * BM + 2>(c3, 2 * BM + 2, part, tot, sg, epoch);
#pragma unroll
    for (int b = 0; b < BM; ++b) {
      sum_t[b] = c3[b];
      tu[b] = c3[BM + b];
    }
    mu[0] = c3[2 * BM];
    mu[1] = c3[2 * BM + 1];
  } else {
    if (istd) grid_sum<BM>(sum_t, B, part, tot, sg, epoch);
    grid_sum<BM>(tu, B, part, tot, sg, epoch);
    grid_sum<4>(mu, 2, part, tot, sg, epoch);
  }
  const double uu = mu[0], su = mu[1];

  // ---- t_b = t~_b / ||w~_b|| (:860, :875-877);  sw = T^T u / u^T u, unit length (:879-880)
  double sw[BM], mean_t[BM], nu[BM];
  double sw2 = 0.0;
#pragma unroll
  for (int b = 0; b < BM; ++b) {
    sw[b] = mean_t[b] = 0.0;
    nu[b] = 1.0;
    if (b < B) {
      mean_t[b] = istd ? sum_t[b] / (double)n : 0.0;
      nu[b] = sqrt(ssq_g[b]);
      sw[b] = (tu[b] - mean_t[b] * su) / nu[b] / uu;
      sw2 = fma(sw[b], sw[b], sw2);
    }
  }
  const double swn = sqrt(sw2);
#pragma unroll
  for (int b = 0; b < BM; ++b) sw[b] = sw[b] / swn;

  // ---- ts = T sw, unit length (:882-883); scratch in vpass (the pass that read it has finished)
  //      UNIPALS: ts = sum_b X_b (X_b^T y), unit length; the block scores are scaled once c is known
  double m2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll UNR
  for (int64_t i = gtid; i < n; i += gstride) {
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < BM; ++b)
      if (b < B) {
        if (uni) {
          acc += tb[(int64_t)b * n + i] - mean_t[b];
        } else {
          const double tm = (tb[(int64_t)b * n + i] - mean_t[b]) / nu[b];
          tb[(int64_t)b * n + i] = tm;
          acc = fma(tm, sw[b], acc);
        }
      }
    r_vp[i] = acc;
    m2[0] = fma(acc, acc, m2[0]);
  }
  grid_sum<4>(m2, 1, part, tot, sg, epoch);
  const double tsn = sqrt(m2[0]);

  // ---- diff = ||ts_old - ts|| (:887), Y^T ts, ts^T ts
  double d4[4] = {0.0, 0.0, 0.0, 0.0};
  double x4[4] = {0.0, 0.0, 0.0, 0.0};     // [0]: ts^T (X_k w~), the diagonal entry of column kc of P^T W~ (s.C)
  double dmax = 0.0, dmin = 1.0e300;
  double yv[QM];
#pragma unroll
  for (int j = 0; j < QM; ++j) yv[j] = 0.0;
#pragma unroll UNR
  for (int64_t i = gtid; i < n; i += gstride) {
    const double t = r_vp[i] / tsn;
    const double d = r_to[i] - t;
    r_to[i] = t;
    if (s.C) {
      double xw = 0.0;                       // row i of X_k w~ = sum_b ||w~_b|| t_b  (UNIPALS: the unnormalised ts)
      if (uni) xw = r_vp[i];
      else {
#pragma unroll
        for (int b = 0; b < BM; ++b)
          if (b < B) xw = fma(nu[b], tb[(int64_t)b * n + i], xw);
      }
      x4[0] = fma(t, xw, x4[0]);
    }
    d4[0] = fma(d, d, d4[0]);
    d4[1] += fabs(d);
    d4[2] = fma(t, t, d4[2]);
    d4[3] += t;
    dmax = fmax(dmax, fabs(d));
    dmin = fmin(dmin, fabs(d));
#pragma unroll
    for (int j = 0; j < QM; ++j)
      if (j < q) yv[j] = fma(r_Y[(int64_t)j * n + i], t, yv[j]);
  }
  if constexpr (QM + 5 <= 16) {
    double c2[QM + 5];
#pragma unroll
    for (int j = 0; j < 4; ++j) c2[j] = d4[j];
#pragma unroll
    for (int j = 0; j < QM; ++j) c2[4 + j] = yv[j];
    c2[QM + 4] = x4[0];
    grid_sum<QM + 5>(c2, QM + 5, part, tot, sg, epoch);
#pragma unroll
    for (int j = 0; j < 4; ++j) d4[j] = c2[j];
#pragma unroll
    for (int j = 0; j < QM; ++j) yv[j] = c2[4 + j];
    x4[0] = c2[QM + 4];
  } else {
    grid_sum<4>(d4, 4, part, tot, sg, epoch);
    grid_sum<QM>(yv, q, part, tot, sg, epoch);
    if (s.C) grid_sum<4>(x4, 1, part, tot, sg, epoch);
  }
  double diff;
  if (s.norm_kind == 0) diff = sqrt(d4[0]);
  else if (s.norm_kind == 1) diff = d4[1];
  else {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
      dmin = fmin(dmin, __shfl_xor(dmin, off, 64));
    }
    if ((tid & 63) == 0) {
      s_mx[tid >> 6] = dmax;
      s_mn[tid >> 6] = dmin;
    }
    __syncthreads();
    for (int w = 0; w < WG1024 / 64; ++w) {
      dmax = fmax(dmax, s_mx[w]);
      dmin = fmin(dmin, s_mn[w]);
    }
    if (sg.G > 1) {
      double* mine = sg.gpart + ((size_t)(epoch & 1u) * sg.G + blockIdx.x) * 16;
      const double* all = sg.gpart + (size_t)(epoch & 1u) * sg.G * 16;
      if (tid == 0) {
        __hip_atomic_store(mine, dmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, dmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      grid_barrier(sg, epoch);
      for (int g = 0; g < sg.G; ++g) {
        dmax = fmax(dmax, __hip_atomic_load(all + (size_t)g * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        dmin = fmin(dmin, __hip_atomic_load(all + (size_t)g * 16 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
    }
    diff = (s.norm_kind == 2) ? dmax : dmin;
  }
  const bool converged = uni || (s.run > 1 && diff <= s.tol);
  if (s.run > 1 && !converged) {
    // the speculation (two iterations per component) does not hold: stop here, the host takes over at component kc
    if (lead) {
      par[2] = 1.0;
      par[3] = (double)kc;
      par[4] = diff;
    }
    return;
  }

  // ---- v = Y^T ts / ts^T ts (:899);  u = Y v / v^T v, unit length (:911-913); scratch in tspass
  double vv = 0.0;
#pragma unroll
  for (int j = 0; j < QM; ++j)
    if (j < q) {
      yv[j] = yv[j] / d4[2];
      vv = fma(yv[j], yv[j], vv);
    }
  double un[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll UNR
  for (int64_t i = gtid; i < n; i += gstride) {
    double w = 0.0;
#pragma unroll
    for (int j = 0; j < QM; ++j)
      if (j < q) w = fma(r_Y[(int64_t)j * n + i], yv[j], w);
    w = w / vv;
    r_tp[i] = w;
    un[0] = fma(w, w, un[0]);
    un[1] = fma(w, r_u[i], un[1]);     // UNIPALS: u^T y (u_cur is y, the start vector)
    un[2] += w;                            // sum of the new u (its mean, for the centred copy of the next sweep)
  }
  grid_sum<4>(un, 3, part, tot, sg, epoch);
  const double unrm = sqrt(un[0]);

  // out[j] = <Tsc_j, vec> for j < kc, 16 at a time (register budget: 1024 threads leave 128 VGPRs each)
  auto dots = [&](const double* vec, double* out) {
    for (int j0 = 0; j0 < kc; j0 += 16) {
      double c[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) c[jj] = 0.0;
      for (int64_t i = gtid; i < n; i += gstride) {
        const double vi = vec[i];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
          if (j0 + jj < kc) c[jj] = fma(s.Tsc[(int64_t)(j0 + jj) * n + i], vi, c[jj]);
      }
      grid_sum<16>(c, min(16, kc - j0), part, tot, sg, epoch);
#pragma unroll
      for (int jj = 0; jj < 16; ++jj)
        if (j0 + jj < kc && lead) out[j0 + jj] = c[jj];
    }
  };

  // Re-orthogonalised sweep vector (see step_reg.hip): vpass <- vpass - sum_{j < kk} <Tsc_j, vpass> Tsc_j.  X_k^T v is
  // the same for both (X_k = (I - Ts Ts^T) X_0), but the lazy pass takes it as X_0^T v - sum_j P[:,j] <ts_j, v>: with the
  // projected vector the two terms are no longer large against their difference.  The scores are unit length and
  // mutually orthogonal to rounding; the pass still gets the exact coefficients of what it reads (dots afterwards).
  auto project = [&](int kk) {
    for (int j0 = 0; j0 < kk; j0 += 16) {
      double c[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) c[jj] = 0.0;
      for (int64_t i = gtid; i < n; i += gstride) {
        const double vi = s.vpass[i];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
          if (j0 + jj < kk) c[jj] = fma(s.Tsc[(int64_t)(j0 + jj) * n + i], vi, c[jj]);
      }
      grid_sum<16>(c, min(16, kk - j0), part, tot, sg, epoch);
      for (int64_t i = gtid; i < n; i += gstride) {
        double vi = s.vpass[i];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
          if (j0 + jj < kk) vi = fma(-c[jj], s.Tsc[(int64_t)(j0 + jj) * n + i], vi);
        s.vpass[i] = vi;
      }
    }
  };

  if (!converged) {
    // ---- next pass: sweep with the new u (centred copy for the raw-space passes), alpha = 1 / u^T u, cu_j = <ts_j, u>
    // (sum u and u^T u follow from the reduction just taken: one loop and one grid-wide reduction less)
    const double mean_u = istd ? (un[2] / unrm) / (double)n : 0.0;
    for (int64_t i = gtid; i < n; i += gstride) {
      const double ui = s.tspass[i] / unrm;
      s.u_cur[i] = ui;
      s.vpass[i] = ui - mean_u;
    }
    project(kc);
    dots(s.vpass, s.cu);
    if (lead) {
      par[0] = (unrm * unrm) / un[0];      // 1 / u^T u of the unit-length u
      par[4] = diff;
    }
    return;
  }

  // ---- UNIPALS: w_raw = X^T u = c X^T y with c = u^T y / y^T y, so ||w_raw_b||^2 = c^2 ||X_b^T y||^2 and
  //      t_b = X_b w_raw_b / ||w_raw_b|| = c t~_b / sqrt(c^2 ||g_b||^2)                         (:503-517)
  const double c_uni = uni ? (un[1] / unrm) / uu : 1.0;
  if (uni) {
    for (int64_t i = gtid; i < n; i += gstride)
#pragma unroll
      for (int b = 0; b < BM; ++b)
        if (b < B) tb[(int64_t)b * n + i] = c_uni * (tb[(int64_t)b * n + i] - mean_t[b]) / sqrt(c_uni * c_uni * ssq_g[b]);
  }

  // ---- converged: record component kc, then prepare loadings (+ lazy deflation) + first sweep of component kc+1
  const double* u0 = s.Y;                      // start vector of every component: Y[:, 0] (:838)
  double a4[4] = {0.0, 0.0, 0.0, 0.0};         // sum u0, u0^T u0
  for (int64_t i = gtid; i < n; i += gstride) {
    const double ui = s.tspass[i] / unrm;
    s.U[(int64_t)kc * n + i] = ui;
    s.Ts[(int64_t)kc * n + i] = s.ts_old[i];
    const double y0 = u0[i];
    s.u_cur[i] = y0;
    a4[0] += y0;
    a4[1] = fma(y0, y0, a4[1]);
  }
  grid_sum<4>(a4, 2, part, tot, sg, epoch);
  const double mean_ts = istd ? d4[3] / (double)n : 0.0, mean_u0 = istd ? a4[0] / (double)n : 0.0;
  for (int64_t i = gtid; i < n; i += gstride) {
    const double ti = s.ts_old[i] - mean_ts, vi = u0[i] - mean_u0;
    s.tspass[i] = ti;
    s.Tsc[(int64_t)kc * n + i] = ti;
    s.vpass[i] = vi;
  }
  project(kc + 1);         // ... against the score being deflated as well (Tsc[:, kc], just written by this thread)
  dots(s.vpass, s.cu);     // j < kc only: the component being deflated is handled inside the pass (its <ts, v>)
  dots(s.tspass, s.ce);
  if (s.C && blockIdx.x == 0) {
    // column kc of C = P^T W~ (see step_reg.hip / mbpls_nipals_step.C)
    const double f = uni ? c_uni : 1.0;
    double* cc = s.C + (int64_t)kc * s.ldc;
    if (tid < nd && nd == kc) {
      double a = 0.0;
      for (int b = 0; b < B; ++b) a += dd[b * nd + tid];
      cc[tid] = f * a;
    }
    if (tid == 0) cc[kc] = (nd == kc) ? f * (uni ? x4[0] / d4[2] : x4[0]) : 0.0;
  }
#pragma unroll
  for (int j = 0; j < QM; ++j)
    if (j < q && lead) s.V[(int64_t)kc * q + j] = yv[j];
#pragma unroll
  for (int b = 0; b < BM; ++b)
    if (b < B && lead) {
      if (uni) {
        double tot_w = 0.0;
        for (int bb = 0; bb < B; ++bb) tot_w += c_uni * c_uni * ssq_g[bb];
        s.scal[(int64_t)kc * 4 * B + b] = c_uni * c_uni * ssq_g[b];                // ||w_raw_b||^2
        s.scal[(int64_t)kc * 4 * B + B + b] = c_uni * c_uni * ssq_g[b] / tot_w;    // a_b (:512)
      } else {
        s.scal[(int64_t)kc * 4 * B + b] = ssq_g[b];               // ||w~_b||^2 of the last iteration
        s.scal[(int64_t)kc * 4 * B + B + b] = sw[b] * sw[b];      // a_b (:949)
      }
      if (s.nu2_host) s.nu2_host[(int64_t)kc * B + b] = s.scal[(int64_t)kc * 4 * B + b];
    }
  if (lead) {
    s.iters[kc] = (double)s.run;
    if (uni) {
      s.scal[(int64_t)kc * 4 * B + 3 * B] = c_uni;   // w_raw = c X^T y: the caller scales the column the pass left
      par[0] = 1.0;               // the fused products of the next component are taken with y itself (:489)
      par[1] = 1.0 / d4[2];       // p = X^T ts / ts^T ts (:520)
    } else {
      par[0] = 1.0 / a4[1];     // alpha of the fused first sweep of the next component
      par[1] = 1.0;             // inv_tt: ts is unit length (:928 has no division)
    }
    par[4] = diff;
  }
}

// dst[j] = src[j] / sqrt(nu2[b(j)]) for one p-length column: the block weights w_b = w~_b / ||w~_b|| (:860) with the
// norms still on the device
__global__ __launch_bounds__(256) void k_block_unit_col(int64_t p, const double* __restrict__ src, double* __restrict__ dst,
                                                        const int64_t* __restrict__ offs, int nblocks,
                                                        const double* __restrict__ nu2, const double* __restrict__ dpar) {
  if (dpar && dpar[2] != 0.0) return;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= p) return;
  int b = 0;
  while (b + 1 < nblocks && j >= offs[b + 1]) ++b;
  dst[j] = src[j] / sqrt(nu2[b]);
}

// dst[j] = src[j] * (*scalar): a p-length column times a scalar that lives on the device
__global__ __launch_bounds__(256) void k_scale_col_by(int64_t p, const double* __restrict__ src, double* __restrict__ dst,
                                                      const double* __restrict__ scalar, const double* __restrict__ dpar) {
  if (dpar && dpar[2] != 0.0) return;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < p) dst[j] = src[j] * scalar[0];
}

}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_set_pass_params(mbpls_ctx* ctx, const double* d_par) {
  MB_ARG(ctx, "ctx is NULL");
  ctx->dpar = d_par;
  return MBPLS_OK;
}

int mbpls_hip_set_true_space_outputs(mbpls_ctx* ctx, const double* d_rscale, double* d_p_true) {
  MB_ARG(ctx, "ctx is NULL");
  MB_ARG(d_rscale || !d_p_true, "d_p_true needs d_rscale");
  ctx->ts_rscale = d_rscale;
  ctx->ts_p_true = d_p_true;
  return MBPLS_OK;
}

int mbpls_hip_set_variance_output(mbpls_ctx* ctx, double* d_var) {
  MB_ARG(ctx, "ctx is NULL");
  ctx->var_out = d_var;
  return MBPLS_OK;
}

int mbpls_hip_nipals_step(mbpls_ctx* ctx, const mbpls_nipals_step* s) {
  MB_ARG(ctx && s, "NULL argument");
  MB_ARG(s->n >= 1 && s->nblocks >= 1 && s->nblocks <= RS_BMAX && s->q >= 1 && s->q <= RS_QMAX,
         "step kernel covers nblocks <= %d, q <= %d (got %d, %d)", RS_BMAX, RS_QMAX, s->nblocks, s->q);
  MB_ARG(s->kc >= 0 && s->kc < RS_KMAX - 1 && s->nd >= 0 && s->nd <= s->kc && s->run >= 1, "bad component / run");
  MB_ARG(s->norm_kind >= 0 && s->norm_kind <= 3, "bad norm kind %d", s->norm_kind);
  MB_ARG(s->mode == 0 || (s->mode == 1 && s->q == 1 && s->run == 1), "mode 1 (UNIPALS) is one response, one step per component");
  MB_ARG(s->xchg && s->Y && s->Ts && s->Tsc && s->U && s->V && s->Tb && s->scal && s->ssq_x && s->iters && s->u_cur &&
             s->ts_old && s->vpass && s->tspass && s->cu && s->ce && s->par,
         "NULL pointer in mbpls_nipals_step");
  // one workgroup up to 32768 rows (latency-bound); beyond that the n-vector traffic dominates (4 ms at n = 10^6 on one
  // CU): spread the rows over up to 128 workgroups with grid-wide reductions
  StepGrid sg{};
  sg.G = (int)std::max<int64_t>(1, std::min<int64_t>(STEP_MAX_WG, s->n / STEP_ROWS_PER_WG));
  if (sg.G == 1 && ctx->step_reg) {
    // the everyday shape (one response, <= 4 blocks, n <= 4096): rows in registers (step_reg.hip)
    const int rc = launch_step_reg(ctx, s);
    if (rc != MBPLS_ERR_UNSUPPORTED) return rc;
  }
  const bool few = s->nblocks <= 4 && s->q == 1;
  const void* fn = few ? (const void*)k_nipals_step<4, 1> : (const void*)k_nipals_step<RS_BMAX, RS_QMAX>;
  if (sg.G > 1) {
    // the grid barrier needs every workgroup resident at once: never more than the device can hold (a partitioned or
    // smaller part), and launched cooperatively so the runtime guarantees it whatever else the GPU is doing
    int per_cu = 0;
    MB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, WG1024, 0));
    sg.G = std::max(1, std::min(sg.G, per_cu * ctx->cus));
  }
  if (sg.G > 1) {
    const size_t need = 256 + sizeof(double) * 2 * STEP_MAX_WG * 16;
    if (!ctx->step_ws) MB_HIP(hipExtMallocWithFlags(&ctx->step_ws, need, hipDeviceMallocFinegrained));
    sg.bar = (unsigned*)ctx->step_ws;
    sg.gpart = (double*)((char*)ctx->step_ws + 256);
    sg.par = s->par;
    sg.spin_limit = ctx->step_spin_limit;
    MB_HIP(hipMemsetAsync(sg.bar, 0, 256, ctx->stream));
    mbpls_nipals_step sv = *s;
    void* args[] = {(void*)&sv, (void*)&sg};
    MB_HIP(hipLaunchCooperativeKernel(fn, dim3((unsigned)sg.G), dim3(WG1024), args, 0, ctx->stream));
    return MBPLS_OK;
  }
  if (few)
    hipLaunchKernelGGL((k_nipals_step<4, 1>), dim3(1), dim3(WG1024), 0, ctx->stream, *s, sg);
  else
    hipLaunchKernelGGL((k_nipals_step<RS_BMAX, RS_QMAX>), dim3(1), dim3(WG1024), 0, ctx->stream, *s, sg);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

int mbpls_hip_block_unit_col(const mbpls_mat* m, const double* d_src, double* d_dst, const double* d_nu2) {
  MB_ARG(m && d_src && d_dst && d_nu2, "NULL argument");
  mbpls_ctx* ctx = m->ctx;
  const int B = m->nblocks;
  const int64_t* d_offs = m->d_block_offsets;      // uploaded when the matrix was created: no copy inside the queue
  hipLaunchKernelGGL(k_block_unit_col, dim3((unsigned)((m->p + 255) / 256)), dim3(256), 0, ctx->stream, m->p, d_src,
                     d_dst, (const int64_t*)d_offs, B, d_nu2, ctx->dpar);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

int mbpls_hip_scale_col_by(mbpls_ctx* ctx, int64_t p, const double* d_src, double* d_dst, const double* d_scalar) {
  MB_ARG(ctx && d_src && d_dst && d_scalar && p >= 1, "bad argument");
  hipLaunchKernelGGL(k_scale_col_by, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, ctx->stream, p, d_src, d_dst,
                     d_scalar, ctx->dpar);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

}  // extern "C"

// one kernel of this translation unit, for mbpls_hip_warmup (its code object is loaded on first use otherwise)
namespace mbpls {
const void* anchor_resident() { return (const void*)&k_block_unit_col; }
}  // namespace mbpls
