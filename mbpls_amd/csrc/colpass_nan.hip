// Column passes for data with missing values (sparse_data=True, NaN-aware NIPALS of Martens & Martens;
// reference mbpls/mbpls.py:846-854 weights, :865-873 block scores, :919-926 loadings, :969 deflation).
//
// Same ownership as colpass.hip: one workgroup per chunk of contiguous feature columns, a column lives in
// the registers of the whole workgroup.  A NaN marks a missing entry; it stays NaN in HBM through
// standardisation and deflation (as in the reference, where NaN - x = NaN), and every sum simply skips it:
//   weights   g_j = sum_{i present} x_ij u_i / sum_{i present} u_i^2                 (:848-852)
//   scores    A_i = sum_{j present} x_ij g_j,   D_i = sum_{j present} g_j^2           (:866-872, the host forms
//             t_i = nu A_i / D_i with nu = ||g_b||: the score over the present columns of the NORMALISED weights)
//   loadings  p_j = sum_{i present} x_ij ts_i / sum_{i present} ts_i^2              (:921-925)
// For a column (row) without missing entries the masked denominator equals the full one, so the reference's
// separate dense branch (u^T u, ||w|| = 1) is the same number up to the rounding of one summation.
// Lazy deflation does not apply (the correction terms would depend on every row's own mask): X is rewritten.
//
// Bytes per column: SWEEP 8n read; STD / DEFL (+SWEEP) 8n read + 8n write.
#include <algorithm>

#include "colpass_util.h"

namespace mbpls {


// Workgroup sum of NV <= 8 values (LDS scratch double[2][16][8]; one barrier per call).
template <int NV>
__device__ __forceinline__ void block_sum8(double (&v)[NV], double* red, int& buf) {
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
  const int nw = blockDim.x >> 6;
  if (nw == 1) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double* r = red + buf * 128;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) r[w * 8 + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double s = r[k];
    for (int i = 1; i < nw; ++i) s += r[i * 8 + k];
    v[k] = s;
  }
  buf ^= 1;
}

// FULL: n == T*R == ld -- no row masks and, more important, no select right behind the column load that would
// force an s_waitcnt before the prefetched column is needed (see colpass.hip).
template <int R, int TMAX, bool STD, bool DEFL, bool SWEEP, bool FULL>
__global__ __launch_bounds__(TMAX, (R == 8 && TMAX == 512) ? 4 : 1) void k_colpass_nan(NanArgs a) {
  constexpr int D = 2;   // columns in flight
  __shared__ double red[2 * 128];
  int buf = 0;
  const int tid = threadIdx.x, T = blockDim.x;
  const int64_t n = a.n, ld = a.ld;
  const int chunk = blockIdx.x;
  const int64_t j0 = uniform64(a.c0[chunk]), j1 = uniform64(a.c1[chunk]);
  RowMap<R> rm;
  rm.init(n, ld, tid, T, STD ? a.rowmask : nullptr);   // row-subset fit: excluded samples load as 0 and count nowhere
  auto real = [&](int k) -> bool { return FULL || ((((k & 1) ? rm.in_n1 : rm.in_n0) >> (k >> 1)) & 1u); };

  // The n = 4096 configuration (R = 8, 512 threads) must stay at <= 128 VGPRs for two workgroups per CU (see
  // colpass.hip); its fused variants keep the n-vectors u / ts in LDS (each thread re-reads its own row pairs).
  constexpr bool VEC_LDS = (R == 8 && TMAX == 512) && (STD || DEFL);
  __shared__ double2 s_v[(VEC_LDS && SWEEP) ? (TMAX * R / 2) : 1];
  __shared__ double2 s_ts[(VEC_LDS && DEFL) ? (TMAX * R / 2) : 1];
  double v_reg[(SWEEP && !VEC_LDS) ? R : 1], ts_reg[(DEFL && !VEC_LDS) ? R : 1], acc[SWEEP ? R : 1], dac[SWEEP ? R : 1];
  if constexpr (SWEEP) {
    if constexpr (VEC_LDS) {
      double tmp[R];
      load_vec<R>(tmp, a.v, n, tid, T);
#pragma unroll
      for (int k = 0; k < R / 2; ++k) s_v[k * T + tid] = make_double2(tmp[2 * k], tmp[2 * k + 1]);
    } else {
      load_vec<R>(v_reg, a.v, n, tid, T);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = dac[k] = 0.0;
  }
  if constexpr (DEFL) {
    if constexpr (VEC_LDS) {
      double tmp[R];
      load_vec<R>(tmp, a.ts, n, tid, T);
#pragma unroll
      for (int k = 0; k < R / 2; ++k) s_ts[k * T + tid] = make_double2(tmp[2 * k], tmp[2 * k + 1]);
    } else {
      load_vec<R>(ts_reg, a.ts, n, tid, T);
    }
  }
  auto v_pair = [&](int k) -> double2 {
    if constexpr (VEC_LDS) return s_v[k * T + tid];
    else return make_double2(v_reg[2 * k], v_reg[2 * k + 1]);
  };
  auto ts_pair = [&](int k) -> double2 {
    if constexpr (VEC_LDS) return s_ts[k * T + tid];
    else return make_double2(ts_reg[2 * k], ts_reg[2 * k + 1]);
  };
  double ssq_g = 0.0, ssq_p = 0.0, ssq_x = 0.0;

  double x[D][R];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (j0 + d < j1) load_col<R, true>(x[d], a.src + (j0 + d) * ld, rm);   // select-free; masked at the point of use

  for (int64_t j = j0; j < j1; j += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int64_t jj = j + d;
      if (jj >= j1) break;
      double(&xc)[R] = x[d];
      if constexpr (!FULL) {
        // rows beyond n / excluded samples become 0 now that the column is needed (a select right behind the load would
        // stall the prefetch): present, and without weight in any sum
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const bool real = (((k & 1) ? rm.in_n1 : rm.in_n0) >> (k >> 1)) & 1u;
          xc[k] = real ? xc[k] : 0.0;
        }
      }
      // one bit per row of the thread: 64 rows per thread (32768 < n <= 65536) need a 64-bit mask -- with 32 bits the
      // rows 32..63 of every thread took the flags of rows 0..31 and a missing value there poisoned the sums (found
      // by the long-column fuzz sweep of round 5; n <= 32768 was never affected)
      using pm_t = typename std::conditional<(R > 32), unsigned long long, unsigned>::type;
      pm_t pm = 0;
#pragma unroll
      for (int k = 0; k < R; ++k) pm |= (pm_t)((xc[k] == xc[k]) ? 1u : 0u) << k;
      auto present = [&](int k) -> bool { return (pm >> k) & (pm_t)1; };

      if constexpr (STD) {
        if (a.mode != MBPLS_SCALE_FIT && a.count) {      // present samples per column (FIT counts them anyway)
          double c1[1] = {0.0};
#pragma unroll
          for (int k = 0; k < R; ++k) c1[0] += (present(k) && real(k)) ? 1.0 : 0.0;
          block_sum8<1>(c1, red, buf);
          if (tid == 0) a.count[jj] = c1[0];
        }
        if (a.mode != MBPLS_SCALE_NONE) {
          double mean, scale;
          if (a.mode == MBPLS_SCALE_FIT) {
            // sklearn StandardScaler on data with NaN (_incremental_mean_and_var with per-column counts)
            double sc[2] = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < R; ++k) {
              const bool use = present(k) && real(k);
              sc[0] += use ? xc[k] : 0.0;
              sc[1] += use ? 1.0 : 0.0;
            }
            block_sum8<2>(sc, red, buf);
            const double cnt = sc[1];
            mean = sc[0] / cnt;
            double cq[2] = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < R; ++k) {
              const bool use = present(k) && real(k);
              const double dlt = use ? xc[k] - mean : 0.0;
              if (real(k)) xc[k] = xc[k] - mean;    // NaN stays NaN
              cq[0] += dlt;
              cq[1] = fma(dlt, dlt, cq[1]);
            }
            block_sum8<2>(cq, red, buf);
            const double var = (cq[1] - cq[0] * cq[0] / cnt) / cnt;
            const double eps = 2.220446049250313e-16;
            const double nme = cnt * mean * eps;
            const double bound = cnt * eps * var + nme * nme;
            scale = (var <= bound) ? 1.0 : sqrt(var);
            if (tid == 0) {
              a.mean[jj] = mean;
              a.scale[jj] = scale;
              if (a.count) a.count[jj] = cnt;
              if (a.var_out) a.var_out[jj] = (var <= bound) ? 0.0 : var;
            }
          } else {
            mean = a.mean[jj];
            scale = a.scale[jj];
#pragma unroll
            for (int k = 0; k < R; ++k)
              if (real(k)) xc[k] = xc[k] - mean;
          }
#pragma unroll
          for (int k = 0; k < R; ++k) xc[k] = xc[k] / scale;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const double z = present(k) ? xc[k] : 0.0;
          ssq_x = fma(z, z, ssq_x);                     // nansum(X^2), mbpls.py:826
        }
        store_col<R, FULL>(xc, a.dst + jj * ld, rm);
      }

      double gj = 0.0;
      if constexpr (DEFL) {
        // one reduction: <x,ts>_present, <ts,ts>_present and, for the fused sweep, <x,u>, <u,u>, <ts,u>
        double r5[SWEEP ? 5 : 2];
#pragma unroll
        for (int q = 0; q < (SWEEP ? 5 : 2); ++q) r5[q] = 0.0;
#pragma unroll
        for (int kp = 0; kp < R / 2; ++kp) {
          const double2 t2 = ts_pair(kp);
          double2 v2 = make_double2(0.0, 0.0);
          if constexpr (SWEEP) v2 = v_pair(kp);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int k = 2 * kp + h;
            const double tk = h ? t2.y : t2.x, vk = h ? v2.y : v2.x;
            const bool pr = present(k);
            const double xz = pr ? xc[k] : 0.0;
            const double tz = pr ? tk : 0.0;
            r5[0] = fma(xz, tk, r5[0]);
            r5[1] = fma(tz, tk, r5[1]);
            if constexpr (SWEEP) {
              const double vz = pr ? vk : 0.0;
              r5[2] = fma(xz, vk, r5[2]);
              r5[3] = fma(vz, vk, r5[3]);
              r5[4] = fma(tz, vk, r5[4]);
            }
          }
        }
        block_sum8<(SWEEP ? 5 : 2)>(r5, red, buf);
        const double pj = (r5[1] != 0.0) ? r5[0] / r5[1] : 0.0;
        ssq_p = fma(pj, pj, ssq_p);
        if (tid == 0 && a.pvec) a.pvec[jj] = pj;
        if (a.deflate) {
#pragma unroll
          for (int kp = 0; kp < R / 2; ++kp) {                            // NaN stays NaN (:969)
            const double2 t2 = ts_pair(kp);
            xc[2 * kp] = fma(-pj, t2.x, xc[2 * kp]);
            xc[2 * kp + 1] = fma(-pj, t2.y, xc[2 * kp + 1]);
          }
          store_col<R, FULL>(xc, a.dst + jj * ld, rm);
          if constexpr (SWEEP) gj = (r5[3] != 0.0) ? fma(-pj, r5[4], r5[2]) / r5[3] : 0.0;
        } else {
          if constexpr (SWEEP) gj = (r5[3] != 0.0) ? r5[2] / r5[3] : 0.0;
        }
      } else if constexpr (SWEEP) {
        double r2[2] = {0.0, 0.0};
#pragma unroll
        for (int kp = 0; kp < R / 2; ++kp) {
          const double2 v2 = v_pair(kp);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int k = 2 * kp + h;
            const double vk = h ? v2.y : v2.x;
            const bool pr = present(k);
            r2[0] = fma(pr ? xc[k] : 0.0, vk, r2[0]);
            r2[1] = fma(pr ? vk : 0.0, vk, r2[1]);
          }
        }
        block_sum8<2>(r2, red, buf);
        gj = (r2[1] != 0.0) ? r2[0] / r2[1] : 0.0;
      }
      if constexpr (SWEEP) {
        ssq_g = fma(gj, gj, ssq_g);
        if (tid == 0 && a.g) a.g[jj] = gj;
        const double g2 = gj * gj;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const bool pr = present(k);
          acc[k] = fma(gj, pr ? xc[k] : 0.0, acc[k]);
          dac[k] += pr ? g2 : 0.0;
        }
      }
      if (jj + D < j1) load_col<R, true>(xc, a.src + (jj + D) * ld, rm);
    }
  }

  if constexpr (SWEEP) {
    double* out = a.part_t + (int64_t)chunk * n;
    double* outd = a.part_d + (int64_t)chunk * n;
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      const int64_t r0 = 2 * ((int64_t)k * T + tid);
      if (FULL || r0 < n) {
        out[r0] = acc[2 * k];
        outd[r0] = dac[2 * k];
      }
      if (FULL || r0 + 1 < n) {
        out[r0 + 1] = acc[2 * k + 1];
        outd[r0 + 1] = dac[2 * k + 1];
      }
    }
  }
  if constexpr (STD) {
    double s[1] = {ssq_x};
    block_sum8<1>(s, red, buf);
    ssq_x = s[0];
  }
  if (tid == 0) {
    double* ps = a.part_s + (int64_t)chunk * 4;
    ps[0] = ssq_g;
    ps[1] = ssq_p;
    ps[2] = ssq_x;
    ps[3] = 0.0;
  }
}

// Columns too long for the registers of one workgroup (n > ctx->nan_long_from, default 8192: the R = 32 / 64 shapes
// above cover up to 65536 rows but spill -- measured 2x slower than this kernel at every n above 8192, round 5).
// Same ownership -- one workgroup per chunk of columns, same sums in the same places -- but a column is STREAMED:
// thread t owns rows t, t + T, t + 2T, ... of every column and of the chunk's partial vectors A / D (which live in
// their [nchunks][n] slots of the workspace instead of registers), and a column is visited once per dependent
// reduction (statistics: count + sum, centred squares, store; sweep: dot products, then the update of A / D).  The
// re-visits hit L2; HBM sees the column once or twice.  Columns go in groups of NL_CB: the row loops that read the
// n-vectors u / ts and the one that updates A / D (and rewrites X when it deflates) run once per group, which cuts the
// vector traffic per column to a quarter; the sums are formed column by column in the same order as without groups.
// The reference's own loop over rows and columns with missing entries (mbpls.py:846-873) is Python -- nobody waits for
// it at this size.
constexpr int NL_CB = 4;

template <bool STD, bool DEFL, bool SWEEP>
__global__ __launch_bounds__(1024) void k_colpass_nan_long(NanArgs a) {
  // (the loadings-only pass has no A / D to update and reads ts twice per column either way: groups only cost it
  // 4.3 -> 3.2 TB/s, measured)
  constexpr int CB = (DEFL && !SWEEP) ? 1 : NL_CB;
  __shared__ double red[2 * 128];
  int buf = 0;
  const int tid = threadIdx.x, T = blockDim.x;
  const int64_t n = a.n, ld = a.ld;
  const int chunk = blockIdx.x;
  const int64_t j0 = uniform64(a.c0[chunk]), j1 = uniform64(a.c1[chunk]);
  double* __restrict__ acc = SWEEP ? a.part_t + (int64_t)chunk * n : nullptr;
  double* __restrict__ dac = SWEEP ? a.part_d + (int64_t)chunk * n : nullptr;
  if constexpr (SWEEP)
    for (int64_t i = tid; i < n; i += T) acc[i] = dac[i] = 0.0;
  const double* __restrict__ rowmask = STD ? a.rowmask : nullptr;
  // a sample excluded from the fit (cross-validation fold) loads as 0: present, and without weight in any sum
  auto excluded = [&](int64_t i) -> bool { return rowmask && rowmask[i] == 0.0; };
  double ssq_g = 0.0, ssq_p = 0.0, ssq_x = 0.0;

  for (int64_t jg = j0; jg < j1; jg += CB) {
    const int cnt = (int)((j1 - jg) < CB ? (j1 - jg) : CB);   // workgroup-uniform
    double gj[CB], pj[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) gj[c] = pj[c] = 0.0;

    if constexpr (STD) {
      // statistics and the stored standardised column, column by column (every step depends on the one before)
      for (int c = 0; c < cnt; ++c) {
        const int64_t jj = jg + c;
        const double* col = a.src + jj * ld;     // (src may be dst: no restrict)
        double* out = a.dst + jj * ld;
        if (a.mode != MBPLS_SCALE_FIT && a.count) {
          double c1[1] = {0.0};
          for (int64_t i = tid; i < n; i += T) {
            const double x = col[i];
            c1[0] += (x == x && !excluded(i)) ? 1.0 : 0.0;
          }
          block_sum8<1>(c1, red, buf);
          if (tid == 0) a.count[jj] = c1[0];
        }
        double mean = 0.0, scale = 1.0;
        if (a.mode == MBPLS_SCALE_FIT) {
          // sklearn StandardScaler on data with NaN (_incremental_mean_and_var with per-column counts)
          double sc[2] = {0.0, 0.0};
          for (int64_t i = tid; i < n; i += T) {
            const double x = col[i];
            const bool use = x == x && !excluded(i);
            sc[0] += use ? x : 0.0;
            sc[1] += use ? 1.0 : 0.0;
          }
          block_sum8<2>(sc, red, buf);
          const double cntp = sc[1];
          mean = sc[0] / cntp;
          double cq[2] = {0.0, 0.0};
          for (int64_t i = tid; i < n; i += T) {
            const double x = col[i];
            const double dlt = (x == x && !excluded(i)) ? x - mean : 0.0;
            cq[0] += dlt;
            cq[1] = fma(dlt, dlt, cq[1]);
          }
          block_sum8<2>(cq, red, buf);
          const double var = (cq[1] - cq[0] * cq[0] / cntp) / cntp;
          const double eps = 2.220446049250313e-16;
          const double nme = cntp * mean * eps;
          const double bound = cntp * eps * var + nme * nme;
          scale = (var <= bound) ? 1.0 : sqrt(var);
          if (tid == 0) {
            a.mean[jj] = mean;
            a.scale[jj] = scale;
            if (a.count) a.count[jj] = cntp;
            if (a.var_out) a.var_out[jj] = (var <= bound) ? 0.0 : var;
          }
        } else if (a.mode == MBPLS_SCALE_APPLY) {
          mean = a.mean[jj];
          scale = a.scale[jj];
        }
        double r2[2] = {0.0, 0.0};
        for (int64_t i = tid; i < ld; i += T) {
          double z = 0.0;
          if (i < n && !excluded(i)) {
            z = col[i];
            if (a.mode != MBPLS_SCALE_NONE) z = z - mean;          // NaN stays NaN
          }
          if (a.mode != MBPLS_SCALE_NONE) z = z / scale;
          out[i] = z;
          const bool pr = z == z;
          ssq_x = fma(pr ? z : 0.0, pr ? z : 0.0, ssq_x);          // nansum(X^2), mbpls.py:826
          if constexpr (SWEEP) {
            if (i < n) {
              const double vk = a.v[i];
              r2[0] = fma(pr ? z : 0.0, vk, r2[0]);
              r2[1] = fma(pr ? vk : 0.0, vk, r2[1]);
            }
          }
        }
        if constexpr (SWEEP) {
          block_sum8<2>(r2, red, buf);
          gj[c] = (r2[1] != 0.0) ? r2[0] / r2[1] : 0.0;
        }
      }
    } else {
      // one visit of the group for all its dot products: <x,ts>_present, <ts,ts>_present and, for the sweep,
      // <x,u>, <u,u>, <ts,u> (DEFL + SWEEP: five per column; DEFL: the first two; SWEEP: <x,u>, <u,u>)
      constexpr int NS = DEFL ? (SWEEP ? 5 : 2) : 2;
      const double* base = a.src + jg * ld;
      double r[CB][NS];
#pragma unroll
      for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int q = 0; q < NS; ++q) r[c][q] = 0.0;
      for (int64_t i = tid; i < n; i += T) {
        double tk = 0.0, vk = 0.0;
        if constexpr (DEFL) tk = a.ts[i];
        if constexpr (SWEEP) vk = a.v[i];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          if (c < cnt) {
            const double x = base[(int64_t)c * ld + i];
            const bool pr = x == x;
            const double xz = pr ? x : 0.0;
            if constexpr (DEFL) {
              const double tz = pr ? tk : 0.0;
              r[c][0] = fma(xz, tk, r[c][0]);
              r[c][1] = fma(tz, tk, r[c][1]);
              if constexpr (SWEEP) {
                const double vz = pr ? vk : 0.0;
                r[c][2] = fma(xz, vk, r[c][2]);
                r[c][3] = fma(vz, vk, r[c][3]);
                r[c][4] = fma(tz, vk, r[c][4]);
              }
            } else {
              r[c][0] = fma(xz, vk, r[c][0]);
              r[c][1] = fma(pr ? vk : 0.0, vk, r[c][1]);
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        if (c < cnt) {
          block_sum8<NS>(r[c], red, buf);
          if constexpr (DEFL) {
            pj[c] = (r[c][1] != 0.0) ? r[c][0] / r[c][1] : 0.0;
            ssq_p = fma(pj[c], pj[c], ssq_p);
            if (tid == 0 && a.pvec) a.pvec[jg + c] = pj[c];
            if constexpr (SWEEP) {
              if (a.deflate) gj[c] = (r[c][3] != 0.0) ? fma(-pj[c], r[c][4], r[c][2]) / r[c][3] : 0.0;
              else gj[c] = (r[c][3] != 0.0) ? r[c][2] / r[c][3] : 0.0;
            }
          } else {
            gj[c] = (r[c][1] != 0.0) ? r[c][0] / r[c][1] : 0.0;
          }
        }
      }
    }

    // second visit of the group: rewrite X where it is deflated, add the group's columns to A / D (column by column,
    // the order of the sums without groups)
    const bool defl = DEFL && a.deflate;
    if (SWEEP || defl) {
      double* cols = (STD || DEFL) ? a.dst + jg * ld : nullptr;       // (STD: this thread's own stores above)
      const double* rcols = (STD || DEFL) ? cols : a.src + jg * ld;
      for (int64_t i = tid; i < n; i += T) {
        double a_ = 0.0, d_ = 0.0, tk = 0.0;
        if constexpr (SWEEP) {
          a_ = acc[i];
          d_ = dac[i];
        }
        if (defl) tk = a.ts[i];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          if (c < cnt) {
            double x = rcols[(int64_t)c * ld + i];
            const bool pr = x == x;
            if (defl) {
              x = fma(-pj[c], tk, x);                               // NaN stays NaN (:969)
              cols[(int64_t)c * ld + i] = x;
            }
            if constexpr (SWEEP) {
              a_ = fma(gj[c], pr ? x : 0.0, a_);
              d_ += pr ? gj[c] * gj[c] : 0.0;
            }
          }
        }
        if constexpr (SWEEP) {
          acc[i] = a_;
          dac[i] = d_;
        }
      }
    }
    if constexpr (SWEEP) {
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        if (c < cnt) {
          ssq_g = fma(gj[c], gj[c], ssq_g);
          if (tid == 0 && a.g) a.g[jg + c] = gj[c];
        }
      }
    }
  }

  if constexpr (STD) {
    double s[1] = {ssq_x};
    block_sum8<1>(s, red, buf);
    ssq_x = s[0];
  }
  if (tid == 0) {
    double* ps = a.part_s + (int64_t)chunk * 4;
    ps[0] = ssq_g;
    ps[1] = ssq_p;
    ps[2] = ssq_x;
    ps[3] = 0.0;
  }
}

template <bool STD, bool DEFL, bool SWEEP>
static int launch_nan(mbpls_ctx* ctx, const ChunkTable& ct, const NanArgs& a, int64_t p) {
  const bool writes = STD || (DEFL && a.deflate);
  const int kind = STD ? MBPLS_K_STANDARDIZE : (DEFL ? (SWEEP ? MBPLS_K_DEFLATE_SWEEP : MBPLS_K_DEFLATE) : MBPLS_K_SWEEP);
  ProfScope prof(ctx, kind, 8.0 * (double)a.n * (double)p * (writes ? 2.0 : 1.0));
  // few samples: several columns per wave (colpass_sub.hip)
  if (a.n <= SUB_MAX_N && ctx->sub_columns) return launch_nan_sub(ctx, ct, a, STD, DEFL, SWEEP);
  Cfg cfg;
  int T = 0;
  if (a.n > ctx->nan_long_from || !pick_cfg(ctx, a.n, &cfg, &T, false)) {
    // streamed columns: as many threads as there are rows, up to 1024
    const int TL = (int)std::min<int64_t>(1024, (a.n + 63) / 64 * 64);
    hipLaunchKernelGGL((k_colpass_nan_long<STD, DEFL, SWEEP>), dim3((unsigned)ct.nchunks), dim3((unsigned)TL), 0,
                       ctx->stream, a);
    MB_HIP(hipGetLastError());
    return MBPLS_OK;
  }
  if (cfg.R == 8 && cfg.TMAX == 512 && T < 256 && (STD || DEFL)) {
    // the fused R=8 variants stage u / ts in LDS arrays sized for 512 threads: small workgroups would be capped at two
    // per CU -- take the register-only R=4 shape instead
    cfg = Cfg{4, 512};
    T = (int)((((a.n + 3) / 4) + 63) / 64 * 64);
  }
  dim3 grid((unsigned)ct.nchunks), block((unsigned)T);
  const bool full = (a.n == (int64_t)T * cfg.R) && (a.ld == a.n) && cfg.R >= 8 && !(STD && a.rowmask);
#define MB_LAUNCH(R_, TM_)                                                                                       \
  do {                                                                                                           \
    if (full && (R_) >= 8 && (R_) <= 16)                                                                         \
      hipLaunchKernelGGL((k_colpass_nan<R_, TM_, STD, DEFL, SWEEP, ((R_) >= 8 && (R_) <= 16)>), grid, block, 0,  \
                         ctx->stream, a);                                                                        \
    else                                                                                                         \
      hipLaunchKernelGGL((k_colpass_nan<R_, TM_, STD, DEFL, SWEEP, false>), grid, block, 0, ctx->stream, a);     \
  } while (0)
  if (cfg.R == 2) MB_LAUNCH(2, 256);
  else if (cfg.R == 4 && cfg.TMAX == 512) MB_LAUNCH(4, 512);
  else if (cfg.R == 4) MB_LAUNCH(4, 1024);
  else if (cfg.R == 8 && cfg.TMAX == 512) MB_LAUNCH(8, 512);
  else if (cfg.R == 8) MB_LAUNCH(8, 1024);
  else if (cfg.R == 16) MB_LAUNCH(16, 512);
  else if (cfg.R == 32) MB_LAUNCH(32, 1024);
  else if (cfg.R == 64) MB_LAUNCH(64, 1024);
  else {
    set_error("no kernel for R=%d TMAX=%d", cfg.R, cfg.TMAX);
    return MBPLS_ERR_UNSUPPORTED;
  }
#undef MB_LAUNCH
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

static int nan_buffers(mbpls_ctx* ctx, const mbpls_mat* m, NanArgs* a) {
  const size_t nt = (size_t)m->fine.nchunks * (size_t)m->n, ns = (size_t)m->fine.nchunks * 4;
  MB_TRY(ensure_ws(ctx, sizeof(double) * (2 * nt + ns)));
  a->part_t = (double*)ctx->ws;
  a->part_d = a->part_t + nt;
  a->part_s = a->part_d + nt;
  return MBPLS_OK;
}

// A and the scalars first, then D through the same fixed-order reduction
static int nan_reduce(mbpls_ctx* ctx, const mbpls_mat* m, const NanArgs& a, bool sweep, double* d_t, double* d_d,
                      double* d_ssq_g, double* d_ssq_p, double* d_ssq_x) {
  if (!sweep) {
    if (d_ssq_p || d_ssq_x) return reduce_chunks(ctx, m, a.part_t, a.part_s, nullptr, nullptr, d_ssq_p, d_ssq_x);
    return MBPLS_OK;
  }
  MB_TRY(reduce_chunks(ctx, m, a.part_t, a.part_s, d_t, d_ssq_g, d_ssq_p, d_ssq_x));
  return reduce_chunks(ctx, m, a.part_d, a.part_s, d_d, nullptr, nullptr, nullptr);
}

// long columns (beyond the few-samples kernels and the register shapes): the two-phase passes of colpass_nan_long.hip
static bool nan_two_phase(const mbpls_ctx* ctx, int64_t n) {
  return ctx->nan_long_two_phase && n > ctx->nan_long_from && !(n <= SUB_MAX_N && ctx->sub_columns);
}

}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_nan_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode, double* d_mean, double* d_scale,
                                    double* d_count, double* d_ssq_x, const double* d_v, double* d_g, double* d_t,
                                    double* d_d, double* d_ssq_g) {
  return mbpls_hip_nan_standardize_sweep_masked(src, dst, mode, nullptr, d_mean, d_scale, d_count, d_ssq_x, d_v, d_g,
                                                d_t, d_d, d_ssq_g);
}

int mbpls_hip_nan_standardize_sweep_masked(const mbpls_mat* src, mbpls_mat* dst, int mode, const double* d_rowmask,
                                           double* d_mean, double* d_scale, double* d_count, double* d_ssq_x,
                                           const double* d_v, double* d_g, double* d_t, double* d_d,
                                           double* d_ssq_g) {
  MB_ARG(src && dst, "NULL argument");
  MB_ARG(src->n == dst->n && src->p == dst->p && src->ld == dst->ld, "src/dst shape mismatch");
  MB_ARG(mode == MBPLS_SCALE_FIT || mode == MBPLS_SCALE_APPLY || mode == MBPLS_SCALE_NONE, "bad mode %d", mode);
  MB_ARG(mode == MBPLS_SCALE_NONE || (d_mean && d_scale), "mean/scale buffers required");
  MB_ARG(!d_v || (d_t && d_d), "the fused sweep needs d_t and d_d");
  mbpls_ctx* ctx = dst->ctx;
  if (nan_two_phase(ctx, dst->n))
    return nan_long_standardize_sweep(src, dst, mode, d_rowmask, d_mean, d_scale, d_count, d_ssq_x, d_v, d_g, d_t, d_d,
                                      d_ssq_g);
  NanArgs a{};
  MB_TRY(nan_buffers(ctx, dst, &a));
  a.src = src->data; a.dst = dst->data; a.ld = dst->ld; a.n = dst->n;
  a.c0 = dst->fine.d_c0; a.c1 = dst->fine.d_c1;
  a.mode = mode; a.mean = d_mean; a.scale = d_scale; a.count = d_count;
  a.var_out = ctx->var_out;
  a.rowmask = d_rowmask;
  a.v = d_v; a.g = d_g;
  if (d_v) MB_TRY((launch_nan<true, false, true>(ctx, dst->fine, a, dst->p)));
  else MB_TRY((launch_nan<true, false, false>(ctx, dst->fine, a, dst->p)));
  return nan_reduce(ctx, dst, a, d_v != nullptr, d_t, d_d, d_ssq_g, nullptr, d_ssq_x);
}

int mbpls_hip_nan_sweep(const mbpls_mat* m, const double* d_v, double* d_g, double* d_t, double* d_d,
                        double* d_ssq_g) {
  MB_ARG(m && d_v && d_t && d_d, "NULL argument");
  mbpls_ctx* ctx = m->ctx;
  if (nan_two_phase(ctx, m->n)) return nan_long_sweep(m, d_v, d_g, d_t, d_d, d_ssq_g);
  NanArgs a{};
  MB_TRY(nan_buffers(ctx, m, &a));
  a.src = m->data; a.dst = nullptr; a.ld = m->ld; a.n = m->n;
  a.c0 = m->fine.d_c0; a.c1 = m->fine.d_c1;
  a.v = d_v; a.g = d_g;
  MB_TRY((launch_nan<false, false, true>(ctx, m->fine, a, m->p)));
  return nan_reduce(ctx, m, a, true, d_t, d_d, d_ssq_g, nullptr, nullptr);
}

int mbpls_hip_nan_deflate_sweep(mbpls_mat* m, const double* d_ts, int deflate, double* d_p, double* d_ssq_p,
                                const double* d_v, double* d_g, double* d_t, double* d_d, double* d_ssq_g) {
  MB_ARG(m && d_ts, "NULL argument");
  MB_ARG(!d_v || (d_t && d_d), "the fused sweep needs d_t and d_d");
  mbpls_ctx* ctx = m->ctx;
  if (nan_two_phase(ctx, m->n)) return nan_long_deflate_sweep(m, d_ts, deflate, d_p, d_ssq_p, d_v, d_g, d_t, d_d, d_ssq_g);
  NanArgs a{};
  MB_TRY(nan_buffers(ctx, m, &a));
  a.src = m->data; a.dst = m->data; a.ld = m->ld; a.n = m->n;
  a.c0 = m->fine.d_c0; a.c1 = m->fine.d_c1;
  a.ts = d_ts; a.deflate = deflate ? 1 : 0; a.pvec = d_p;
  a.v = d_v; a.g = d_g;
  if (d_v) MB_TRY((launch_nan<false, true, true>(ctx, m->fine, a, m->p)));
  else MB_TRY((launch_nan<false, true, false>(ctx, m->fine, a, m->p)));
  return nan_reduce(ctx, m, a, d_v != nullptr, d_t, d_d, d_ssq_g, d_ssq_p, nullptr);
}

}  // extern "C"
