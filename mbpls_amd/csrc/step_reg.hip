// Register-resident form of the device-resident NIPALS / UNIPALS step (resident.hip has the general one) for the
// everyday shape: one response, up to four blocks, n <= 4096 samples.  Same inputs, same outputs, same decisions
// (mbpls.py:860-913 between two passes over X; :489-517 for UNIPALS with one response) -- but a thread keeps ITS rows of
// every n-vector in registers from the first load to the last store: the general kernel walks ten dependent row loops
// through L2 (25-55 us, growing with the component index), this one issues all its loads up front -- the first eight
// earlier scores included -- and is bounded by its six to eight workgroup reductions (one barrier each, partials
// double-buffered in LDS).  256 threads, one wave per SIMD: up to 512 registers per thread.
//
// Re-orthogonalised sweep vectors (lazy deflation).  The passes that follow take X_k^T v as X_0^T v - sum_i P[:,i] <ts_i, v>.
// For a deep component <x_j^0, v> and the correction are both large against their difference (y is explained by the
// earlier components: <ts_i, y> is big, <x_j^k, y> tiny), so the rounding of the raw dot product -- relative to X_0 -- was
// amplified a hundredfold and more in g_j, and ||ts_old - ts|| <= 1e-14 (mbpls.py:841/:887) went out of reach beyond
// ~25 components.  X_k = (I - Ts Ts^T) X_0, hence X_k^T v = X_k^T v' with v' = v - sum_i ts_i <ts_i, v>: the step hands
// the pass v' (and the coefficients <ts_i, v'>, now at rounding level, so the identity stays exact); alpha = 1/u^T u is
// still taken with the reference's u.  Rounding is then relative to ||x_j^0|| ||v'||, a small multiple of what the
// explicitly deflated matrix gives.
#include "common.h"

namespace mbpls {

// Launch shapes: n <= 2048: 256 threads (one wave per SIMD, up to 512 registers each), the first 8 earlier scores in
// registers; n <= 4096: 512 threads (256 registers each), the first 4.

// Sum M <= 16 values over a workgroup of W = 4 or 8 waves; every thread gets the totals.  One barrier (alternating
// partial sets); the W wave partials of value k are read by W neighbouring lanes, summed with two / three DPP steps and
// handed to every lane with v_readlane -- one or two LDS loads per thread whatever M is (reading all W M partials per
// thread cost 256 registers at M = 16).  Fixed order.
template <int M, int W>
__device__ __forceinline__ void sr_sum(double (&v)[M], double* part, int& buf) {
  static_assert(M <= 16 && (W == 4 || W == 8), "4 or 8 wave partials of up to 16 values");
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < M; ++k) v[k] = wave_sum_last_row(v[k]);
  double* r = part + buf * (8 * 16);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < M; ++k) r[w * 16 + k] = v[k];
  }
  __syncthreads();
  if constexpr (W == 4) {
    double a = r[(lane & 3) * 16 + (lane >> 2)];
    a += dpp_move<0xB1, 0xf>(a);    // quad_perm [1,0,3,2]
    a += dpp_move<0x4E, 0xf>(a);    // quad_perm [2,3,0,1]: the 4 lanes of a quad hold the total of value lane / 4
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(a), 4 * k);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(a), 4 * k);
      v[k] = __hiloint2double(hi, lo);
    }
  } else {
    double a = r[(lane & 7) * 16 + (lane >> 3)];
    double b = (M > 8) ? r[(lane & 7) * 16 + 8 + (lane >> 3)] : 0.0;
    a += dpp_move<0xB1, 0xf>(a);
    a += dpp_move<0x4E, 0xf>(a);
    a += dpp_move<0x141, 0xf>(a);   // row_half_mirror: the 8 lanes of a group hold the total
    if constexpr (M > 8) {
      b += dpp_move<0xB1, 0xf>(b);
      b += dpp_move<0x4E, 0xf>(b);
      b += dpp_move<0x141, 0xf>(b);
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const double src = (k < 8) ? a : b;
      const int lo = __builtin_amdgcn_readlane(__double2loint(src), 8 * (k & 7));
      const int hi = __builtin_amdgcn_readlane(__double2hiint(src), 8 * (k & 7));
      v[k] = __hiloint2double(hi, lo);
    }
  }
  buf ^= 1;
}

// NL: earlier scores SR_JC .. SR_JC + NL - 1 are parked in LDS (dynamic, NL * SR_T * RPT doubles): requested with the other
// loads at the top, used by the lazy-deflation fix from the registers they arrive in, then written to the thread's OWN
// slots (a thread only ever touches its rows: no barrier) and re-read from there by the re-orthogonalisation and the
// coefficient dots.  Walking them through L2 once per phase made the step grow with the component index (13 us at
// component 0, 28 us at component 9: the fixed cost of a fit that does not shrink with the column shard).
template <int SR_T, int RPT, int SR_JC, int BM, bool FULL, int NL>
__global__ __launch_bounds__(SR_T) void k_step_reg(mbpls_nipals_step s) {
  constexpr int SR_W = SR_T / 64;
  constexpr int NJ = SR_JC < 7 ? SR_JC : 7;       // scores per re-orthogonalisation round
  extern __shared__ double s_sc[];                // [NL][RPT][SR_T]
  __shared__ double part[2 * 8 * 16];
  __shared__ double s_mx[SR_W], s_mn[SR_W];
  int buf = 0;
  double* par = s.par;
  // (first of all: a host thread may be waiting to learn that the kernels queued before this one are done)
  if (s.host_seq && threadIdx.x == 0) __hip_atomic_store(s.host_seq, s.seq_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (par[2] != 0.0) return;             // an earlier component needed more iterations than were queued
  const int tid = threadIdx.x;
  const bool lead = tid == 0;
  const int n = (int)s.n;
  const int B = s.nblocks, kc = s.kc, nd = s.nd;
  const bool istd = s.centre != 0;
  const bool uni = s.mode == 1;
  const double* tr = s.xchg;             // [B][n] raw accumulations of the pass
  const double* ssq_g = s.xchg + (int64_t)n * B;
  const double* ssq_p = ssq_g + B;
  const double* ssq_x = ssq_g + 2 * B;
  const double* dd = ssq_g + 3 * B;      // [B][nd] lazy-deflation sums
  const double dn = (double)n;

  if (s.run == 1 && kc == 0 && s.centre) {
    // offset data: raw-space dot products without centring are only good for modestly offset columns (resident.hip)
    // (the measure itself goes back with every fit: a caller that started with centring passes learns from it when the
    // data no longer need them)
    double worst = 0.0;
    for (int b = 0; b < B; ++b) worst = fmax(worst, dd[b]);
    if (lead) par[6] = worst;
    if (s.amp2_max > 0.0 && !(worst <= s.amp2_max)) {
      if (lead) {
        par[2] = 1.0;
        par[3] = 0.0;
        par[5] = 1.0;
        par[6] = worst;
      }
      return;
    }
  }
  if (s.run == 1 && tid < B) {
    if (kc == 0) s.ssq_x[tid] = ssq_x[tid];
    else s.scal[(int64_t)(kc - 1) * 4 * B + 2 * B + tid] = ssq_p[tid];   // ||p_b||^2 of the component just deflated
  }

  // ---- this thread's rows: tid + 256 i (a wave instruction covers 512 contiguous bytes).  Every access is a
  // wave-uniform base + a 32-bit lane offset shared by all arrays (64-bit addresses per row and array overflowed the
  // register file).  FULL: n = 256 RPT, no masks.  EVERYTHING the step reads is requested now, the first SR_JC earlier
  // scores included: they serve the lazy-deflation fix, the re-orthogonalisation and the coefficients of the next pass
  // from registers (walking them through L2 once per phase cost ~3 us each, four times per step).
  unsigned ro[RPT];
  bool ok[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    ro[i] = (unsigned)tid + (unsigned)(i * SR_T);
    ok[i] = FULL || (int)ro[i] < n;
  }
  auto ld = [&](const double* base, int i) -> double {      // base: wave-uniform
    if constexpr (FULL) return base[ro[i]];
    else return ok[i] ? base[ro[i]] : 0.0;
  };
  double u[RPT], y[RPT], to[RPT], t[BM][RPT], tc[SR_JC][RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    u[i] = ld(s.u_cur, i);
    y[i] = ld(s.Y, i);
    to[i] = ld(s.ts_old, i);
  }
#pragma unroll
  for (int b = 0; b < BM; ++b) {
    const double* trb = tr + (int64_t)(b < B ? b : 0) * n;
#pragma unroll
    for (int i = 0; i < RPT; ++i) t[b][i] = (b < B) ? ld(trb, i) : 0.0;
  }
#pragma unroll
  for (int j = 0; j < SR_JC; ++j) {
    const double* tcj = s.Tsc + (int64_t)(j < kc ? j : 0) * n;
#pragma unroll
    for (int i = 0; i < RPT; ++i) tc[j][i] = (j < kc) ? ld(tcj, i) : 0.0;
  }
  double tl[NL > 0 ? NL : 1][RPT];                // the scores that go to LDS
  if constexpr (NL > 0) {
    if (kc > SR_JC) {
#pragma unroll
      for (int jl = 0; jl < NL; ++jl) {
        const double* tcj = s.Tsc + (int64_t)(SR_JC + jl < kc ? SR_JC + jl : 0) * n;
#pragma unroll
        for (int i = 0; i < RPT; ++i) tl[jl][i] = (SR_JC + jl < kc) ? ld(tcj, i) : 0.0;
      }
    }
  }
  auto lds_at = [&](int jl, int i) -> double& { return s_sc[(jl * RPT + i) * SR_T + tid]; };
  // t~_b = A_b - Ts d_b (lazy deflation): nd is 0 or kc
#pragma unroll
  for (int j = 0; j < SR_JC; ++j)
    if (j < nd) {
#pragma unroll
      for (int b = 0; b < BM; ++b)
        if (b < B) {
          const double dbj = dd[b * nd + j];
#pragma unroll
          for (int i = 0; i < RPT; ++i) t[b][i] = fma(-tc[j][i], dbj, t[b][i]);
        }
    }
  if constexpr (NL > 0) {
    if (kc > SR_JC) {
#pragma unroll
      for (int jl = 0; jl < NL; ++jl) {
        if (SR_JC + jl < nd) {
#pragma unroll
          for (int b = 0; b < BM; ++b)
            if (b < B) {
              const double dbj = dd[b * nd + SR_JC + jl];
#pragma unroll
              for (int i = 0; i < RPT; ++i) t[b][i] = fma(-tl[jl][i], dbj, t[b][i]);
            }
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) lds_at(jl, i) = tl[jl][i];
      }
    }
  }
  for (int j = SR_JC + NL; j < nd; ++j) {
    double tj[RPT];
    const double* tcj = s.Tsc + (int64_t)j * n;
#pragma unroll
    for (int i = 0; i < RPT; ++i) tj[i] = ld(tcj, i);
#pragma unroll
    for (int b = 0; b < BM; ++b)
      if (b < B) {
        const double dbj = dd[b * nd + j];
#pragma unroll
        for (int i = 0; i < RPT; ++i) t[b][i] = fma(-tj[i], dbj, t[b][i]);
      }
  }

  // ---- sums of t~_b, t~_b^T u, u^T u, sum u, sum y, y^T y: ONE reduction
  double c1[2 * BM + 4];
#pragma unroll
  for (int k = 0; k < 2 * BM + 4; ++k) c1[k] = 0.0;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
#pragma unroll
    for (int b = 0; b < BM; ++b) {
      c1[b] += t[b][i];
      c1[BM + b] = fma(t[b][i], u[i], c1[BM + b]);
    }
    c1[2 * BM] = fma(u[i], u[i], c1[2 * BM]);
    c1[2 * BM + 1] += u[i];
    c1[2 * BM + 2] += y[i];
    c1[2 * BM + 3] = fma(y[i], y[i], c1[2 * BM + 3]);
  }
  sr_sum<2 * BM + 4, SR_W>(c1, part, buf);
  const double uu = c1[2 * BM], su = c1[2 * BM + 1], sy = c1[2 * BM + 2], yy = c1[2 * BM + 3];

  // ---- t_b = t~_b / ||w~_b|| (:860, :875-877);  sw = T^T u / u^T u, unit length (:879-880)
  double sw[BM], mean_t[BM], nu[BM];
  double sw2 = 0.0;
#pragma unroll
  for (int b = 0; b < BM; ++b) {
    sw[b] = mean_t[b] = 0.0;
    nu[b] = 1.0;
    if (b < B) {
      mean_t[b] = istd ? c1[b] / dn : 0.0;
      nu[b] = sqrt(ssq_g[b]);
      sw[b] = (c1[BM + b] - mean_t[b] * su) / nu[b] / uu;
      sw2 = fma(sw[b], sw[b], sw2);
    }
  }
  const double swn = sqrt(sw2);
#pragma unroll
  for (int b = 0; b < BM; ++b) sw[b] = sw[b] / swn;

  // ---- ts = T sw, unit length (:882-883).  UNIPALS: ts = sum_b X_b (X_b^T y), unit length
  double ts[RPT];
  double m2[1] = {0.0};
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < BM; ++b)
      if (b < B) {
        if (uni) {
          t[b][i] = ok[i] ? t[b][i] - mean_t[b] : 0.0;
          acc += t[b][i];
        } else {
          const double tm = ok[i] ? (t[b][i] - mean_t[b]) / nu[b] : 0.0;
          t[b][i] = tm;
          acc = fma(tm, sw[b], acc);
        }
      }
    ts[i] = acc;
    m2[0] = fma(acc, acc, m2[0]);
  }
  sr_sum<1, SR_W>(m2, part, buf);
  const double tsn = sqrt(m2[0]);

  // ---- diff = ||ts_old - ts|| (:887), Y^T ts, ts^T ts, sum ts
  // (c2[5] = ts^T (X_k w~): the diagonal entry of column kc of P^T W~, see mbpls_nipals_step.C)
  double c2[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double dmax = 0.0, dmin = 1.0e300;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const double tt = ts[i] / tsn;
    ts[i] = tt;
    {
      double xw = 0.0;                       // row i of X_k w~ = sum_b ||w~_b|| t_b  (UNIPALS: sum_b X_b (X_b^T y))
#pragma unroll
      for (int b = 0; b < BM; ++b)
        if (b < B) xw = uni ? xw + t[b][i] : fma(nu[b], t[b][i], xw);
      c2[5] = fma(tt, xw, c2[5]);
    }
    if (ok[i]) {
      const double d = to[i] - tt;
      c2[0] = fma(d, d, c2[0]);
      c2[1] += fabs(d);
      dmax = fmax(dmax, fabs(d));
      dmin = fmin(dmin, fabs(d));
      s.ts_old[ro[i]] = tt;
    }
    c2[2] = fma(tt, tt, c2[2]);
    c2[3] += tt;
    c2[4] = fma(y[i], tt, c2[4]);
  }
  sr_sum<6, SR_W>(c2, part, buf);
  double diff;
  if (s.norm_kind == 0) diff = sqrt(c2[0]);
  else if (s.norm_kind == 1) diff = c2[1];
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
    for (int w = 0; w < SR_W; ++w) {
      dmax = fmax(dmax, s_mx[w]);
      dmin = fmin(dmin, s_mn[w]);
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

  // ---- v = Y^T ts / ts^T ts (:899);  u = Y v / v^T v, unit length (:911-913)
  const double tt_ = c2[2];
  const double yv = c2[4] / tt_;
  const double vv = yv * yv;
  double w_[RPT];
  double c3[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const double w = (y[i] * yv) / vv;
    w_[i] = w;
    c3[0] = fma(w, w, c3[0]);
    c3[1] = fma(w, u[i], c3[1]);         // UNIPALS: u^T y (u_cur is y, the start vector)
    c3[2] += w;
  }
  sr_sum<3, SR_W>(c3, part, buf);
  const double unrm = sqrt(c3[0]);

  // ---- converged: record component kc now (its registers are free for what follows)
  // UNIPALS: w_raw = X^T u = c X^T y with c = u^T y / y^T y, so ||w_raw_b||^2 = c^2 ||X_b^T y||^2 and
  //          t_b = X_b w_raw_b / ||w_raw_b|| = c t~_b / sqrt(c^2 ||g_b||^2)                         (:503-517)
  const double c_uni = uni ? (c3[1] / unrm) / uu : 1.0;
  if (converged) {
#pragma unroll
    for (int b = 0; b < BM; ++b)
      if (b < B) {
        const double f = uni ? c_uni / sqrt(c_uni * c_uni * ssq_g[b]) : 1.0;
        double* tbo = s.Tb + ((int64_t)kc * B + b) * n;
#pragma unroll
        for (int i = 0; i < RPT; ++i)
          if (ok[i]) tbo[ro[i]] = uni ? f * t[b][i] : t[b][i];
      }
    if (s.C) {
      // column kc of C = P^T W~: the lazy-deflation sums of the pass above the diagonal, ts^T (X_k w~) / ts^T ts on it
      // (UNIPALS: w = c X^T y).  Without the sums (nd != kc) the diagonal is zeroed: the caller then takes the other route
      const double f = uni ? c_uni : 1.0;
      double* cc = s.C + (int64_t)kc * s.ldc;
      if (tid < nd && nd == kc) {
        double a = 0.0;
        for (int b = 0; b < B; ++b) a += dd[b * nd + tid];
        cc[tid] = f * a;
      }
      if (lead) cc[kc] = (nd == kc) ? f * (uni ? c2[5] / tt_ : c2[5]) : 0.0;
    }
    if (lead) {
      s.V[kc] = yv;
      double tot_w = 0.0;
      for (int b = 0; b < B; ++b) tot_w += c_uni * c_uni * ssq_g[b];
      for (int b = 0; b < B; ++b) {
        if (uni) {
          s.scal[(int64_t)kc * 4 * B + b] = c_uni * c_uni * ssq_g[b];                // ||w_raw_b||^2
          s.scal[(int64_t)kc * 4 * B + B + b] = c_uni * c_uni * ssq_g[b] / tot_w;    // a_b (:512)
        } else {
          s.scal[(int64_t)kc * 4 * B + b] = ssq_g[b];               // ||w~_b||^2 of the last iteration
          s.scal[(int64_t)kc * 4 * B + B + b] = sw[b] * sw[b];      // a_b (:949)
        }
        if (s.nu2_host) s.nu2_host[(int64_t)kc * B + b] = s.scal[(int64_t)kc * 4 * B + b];
      }
      s.iters[kc] = (double)s.run;
      if (uni) {
        s.scal[(int64_t)kc * 4 * B + 3 * B] = c_uni;   // w_raw = c X^T y: the caller scales the column the pass left
        par[0] = 1.0;               // the fused products of the next component are taken with y itself (:489)
        par[1] = 1.0 / tt_;         // p = X^T ts / ts^T ts (:520)
      } else {
        par[0] = 1.0 / yy;          // alpha of the fused first sweep of the next component: 1 / y^T y (:856)
        par[1] = 1.0;               // inv_tt: ts is unit length (:928 has no division)
      }
      par[4] = diff;
    }
  } else if (lead) {
    par[0] = (unrm * unrm) / c3[0];      // 1 / u^T u of the unit-length u
    par[4] = diff;
  }

  // ---- the vectors of the next pass
  // vp: its sweep vector (run 1: the new u; converged: the start vector y of the next component, :838), zero-sum copy
  // when the passes run in raw-column space.  tsp (converged only): the score vector of the loadings pass.
  double vp[RPT], tsp[RPT];
  if (!converged) {
    const double mean_u = istd ? (c3[2] / unrm) / dn : 0.0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const double ui = w_[i] / unrm;
      if (ok[i]) s.u_cur[ro[i]] = ui;
      vp[i] = ok[i] ? ui - mean_u : 0.0;
      tsp[i] = 0.0;
    }
  } else {
    const double mean_ts = istd ? c2[3] / dn : 0.0, mean_y = istd ? sy / dn : 0.0;
    double* uo = s.U + (int64_t)kc * n;
    double* tso = s.Ts + (int64_t)kc * n;
    double* tco = s.Tsc + (int64_t)kc * n;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      tsp[i] = ok[i] ? ts[i] - mean_ts : 0.0;
      vp[i] = ok[i] ? y[i] - mean_y : 0.0;
      if (ok[i]) {
        uo[ro[i]] = w_[i] / unrm;
        tso[ro[i]] = ts[i];
        s.u_cur[ro[i]] = y[i];                               // start vector of every component: Y[:, 0] (:838)
        s.tspass[ro[i]] = tsp[i];
        tco[ro[i]] = tsp[i];
      }
    }
  }

  // ---- re-orthogonalisation: a_j = <Tsc_j, vp>, e_j = <Tsc_j, tsp> (= ce_j), seven scores per reduction; the
  // component being deflated (converged: tsp itself) joins with <tsp, vp> / <tsp, tsp>.  The scores are unit length and
  // mutually orthogonal to rounding: one Gram-Schmidt sweep with the coefficients of the ORIGINAL vector.
  double cur[2] = {0.0, 0.0};
  if (converged) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      cur[0] = fma(tsp[i], vp[i], cur[0]);
      cur[1] = fma(tsp[i], tsp[i], cur[1]);
    }
  }
  double vq[RPT];                                    // vp minus its components along the scores
#pragma unroll
  for (int i = 0; i < RPT; ++i) vq[i] = vp[i];
  for (int j0 = 0; j0 < kc || (j0 == 0 && converged); j0 += NJ) {
    double c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = 0.0;
    const int m = min(NJ, kc - j0);
    double tj[NJ][RPT];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const bool cached = j0 == 0;                   // (j0 + jj < SR_JC whenever j0 == 0)
      const int jl = j0 + jj - SR_JC;                // its LDS slot, if it has one (NJ < SR_JC: a round can start in the registers' range)
      const double* tcj = s.Tsc + (int64_t)(jj < m ? j0 + jj : 0) * n;
#pragma unroll
      for (int i = 0; i < RPT; ++i)
        tj[jj][i] = cached ? tc[jj][i] : (jj >= m ? 0.0 : ((NL > 0 && jl >= 0 && jl < NL) ? lds_at(jl, i) : ld(tcj, i)));
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        c[2 * jj] = fma(tj[jj][i], vp[i], c[2 * jj]);
        c[2 * jj + 1] = fma(tj[jj][i], tsp[i], c[2 * jj + 1]);
      }
    c[14] = cur[0];
    c[15] = cur[1];
    sr_sum<16, SR_W>(c, part, buf);
    if (j0 == 0 && converged) {
      const double f = c[14] / c[15];
#pragma unroll
      for (int i = 0; i < RPT; ++i) vq[i] = fma(-f, tsp[i], vq[i]);
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int i = 0; i < RPT; ++i) vq[i] = fma(-c[2 * jj], tj[jj][i], vq[i]);   // (c = 0 beyond m)
    if (lead && converged) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        if (jj < m) s.ce[j0 + jj] = c[2 * jj + 1];
    }
  }
  // ---- the pass reads vq; its exact coefficients cu_j = <Tsc_j, vq> (rounding level now): the first SR_JC from the
  // registers, sixteen per reduction
#pragma unroll
  for (int i = 0; i < RPT; ++i)
    if (ok[i]) s.vpass[ro[i]] = vq[i];
  for (int j0 = 0; j0 < kc; j0 += 16) {
    double c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = 0.0;
    const int m = min(16, kc - j0);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      if (j0 == 0 && jj < SR_JC) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) c[jj] = fma(tc[jj][i], vq[i], c[jj]);      // (tc = 0 beyond kc)
      } else if (jj < m) {
        const int jl = j0 + jj - SR_JC;
        if (NL > 0 && jl < NL) {
#pragma unroll
          for (int i = 0; i < RPT; ++i) c[jj] = fma(lds_at(jl, i), vq[i], c[jj]);
        } else {
          const double* tcj = s.Tsc + (int64_t)(j0 + jj) * n;
#pragma unroll
          for (int i = 0; i < RPT; ++i) c[jj] = fma(ld(tcj, i), vq[i], c[jj]);
        }
      }
    }
    sr_sum<16, SR_W>(c, part, buf);
    if (lead) {
#pragma unroll
      for (int jj = 0; jj < 16; ++jj)
        if (jj < m) s.cu[j0 + jj] = c[jj];
    }
  }
}

// MBPLS_ERR_UNSUPPORTED: shape outside this kernel (the caller uses the general one)
int launch_step_reg(mbpls_ctx* ctx, const mbpls_nipals_step* s) {
  if (s->q != 1 || s->nblocks > 4 || s->n > 4096) return MBPLS_ERR_UNSUPPORTED;
#define MB_STEP_K(T_, R_, J_, B_, F_, L_)                                                                      \
  do {                                                                                                       \
    auto fn = k_step_reg<T_, R_, J_, B_, F_, L_>;                                                            \
    constexpr size_t lds = sizeof(double) * (size_t)(L_) * (T_) * (R_);                                      \
    if (lds > 0) MB_ALLOW_LDS(ctx, fn, lds);                                                                 \
    hipLaunchKernelGGL(fn, dim3(1), dim3(T_), lds, ctx->stream, *s);                                         \
  } while (0)
#define MB_STEP(T_, R_, J_, L_)                                                                              \
  do {                                                                                                       \
    const bool full = s->n == (int64_t)(R_) * (T_);                                                          \
    if (s->nblocks == 1 && full) MB_STEP_K(T_, R_, J_, 1, true, L_);                                         \
    else if (s->nblocks == 1) MB_STEP_K(T_, R_, J_, 1, false, L_);                                           \
    else if (full) MB_STEP_K(T_, R_, J_, 4, true, 0);                                                        \
    else MB_STEP_K(T_, R_, J_, 4, false, 0);                                                                 \
  } while (0)
  const int64_t n = s->n;
  // (LDS slots: 256 threads x RPT rows = 2-16 KB per score -> 8 of them; 512 x 8 rows = 32 KB -> 4.  The four-block
  // instantiations have no registers to receive them in.)
  if (n <= 256) MB_STEP(256, 1, 8, 8);
  else if (n <= 512) MB_STEP(256, 2, 8, 8);
  else if (n <= 1024) MB_STEP(256, 4, 8, 8);
  else if (n <= 2048) MB_STEP(256, 8, 8, 8);
  else MB_STEP(512, 8, 4, 4);
#undef MB_STEP
#undef MB_STEP_K
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

}  // namespace mbpls

// one kernel of this translation unit, for mbpls_hip_warmup (its code object is loaded on first use otherwise)
namespace mbpls {
const void* anchor_step_reg() { return (const void*)&k_step_reg<512, 8, 4, 1, true, 4>; }
}  // namespace mbpls
