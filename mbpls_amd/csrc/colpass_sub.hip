// Column passes for FEW samples (n <= 1024) -- the everyday shape of multiblock PLS: tens to hundreds of samples,
// up to millions of features.  colpass.hip gives a column to a workgroup; with a 0.5-2 KB column its per-column
// reduction (six DPP steps, a barrier, the scalar algebra) is all latency and the pass runs at 0.8-2.7 TB/s.
//
// Here a WAVE works on four columns at a time, one per 16-lane DPP row: lane s of a row owns the row pairs
// s + 16k (k < KR, n <= 32 KR), so a load instruction still covers 256 contiguous bytes per column and a column's
// reduction is four row-local DPP steps (quad_perm x2, row_half_mirror, row_mirror -- every lane of the row ends with
// the total, no LDS, no barrier, no cross-row traffic).  The four waves of a workgroup are independent until the end
// of the chunk, where the 16 row groups add their score accumulators, sums of squares and lazy-deflation sums in a
// fixed order through LDS.  Same PassArgs, same outputs, same flavours (storing / statistics-only standardise,
// explicit / lazy deflation, implicit standardisation, centring) as k_colpass, which documents the algebra.
#include <algorithm>

#include "colpass_util.h"

namespace mbpls {

// sum over the 16 lanes of a DPP row; every lane of the row receives it
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_move<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_move<0x140, 0xf>(v);   // row_mirror
  return v;
}

// ... and over a 32-lane half wave (256 < n <= 512: two columns per wave): the row sums, then one xor-16 exchange
// between the two rows of the half through the LDS crossbar (ds_swizzle, no memory access)
__device__ __forceinline__ double half32_sum(double v) {
  v = row16_sum(v);
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);   // bit mode: and 0x1f, or 0, xor 0x10
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
  return v + __hiloint2double(hi, lo);
}
// sum over 8 lanes (half a DPP row: n <= 64, eight columns per wave -- the per-column scalar algebra costs the same
// whatever the column length, so the shortest columns are VALU-issue bound and want the most columns per instruction)
__device__ __forceinline__ double oct8_sum(double v) {
  v += dpp_move<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xf>(v);   // row_half_mirror
  return v;
}
template <int LG>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (LG == 8) return oct8_sum(v);
  else if constexpr (LG == 16) return row16_sum(v);
  else if constexpr (LG == 32) return half32_sum(v);
  else return wave_sum(v);      // one column per wave (512 < n <= 1024): still no barrier, no LDS per column
}

// LG lanes per column (8: eight columns per wave, n <= 16 KR; 16: four, n <= 32 KR; 32: two, n <= 64 KR; 64: one,
// n <= 128 KR)
// LM1: at most LG earlier components to correct for (or none: every pass that is not lazy) -- ONE correction slot per
// lane instead of ceil(39 / LG).  The slots cost the 8-lane shapes ~80 VGPRs (pointers, coefficients, prefetched
// loadings, accumulators for five of them): with them the fused pass sits at 194-218 VGPRs = two waves per SIMD, and
// few-sample columns are short on bytes in flight, not on arithmetic (rocprofv3: waves parked 0.73-0.76 of their cycles).
// (The column loop still waits with s_waitcnt vmcnt(0) -- four causes, all removable, and the loop with counted waits is
// 5-13 % SLOWER below n = 128: what it saves in latency it spends in instructions, profiles/r06_sub_waitcnt_experiment.txt.)
template <int LG, int KR, int D, bool STD, bool DEFL, bool SWEEP, bool ISTD, bool CEN, bool LM1 = false>
__global__ __launch_bounds__(256) void k_colpass_sub(PassArgs a) {
  constexpr int R = 2 * KR;
  constexpr int NG = 256 / LG;                       // column groups per workgroup
  constexpr int SUB_LM = LM1 ? 1 : (KD - 1 + LG - 1) / LG;     // lazy corrections per lane: components s, s + LG, ...
  __shared__ double s_t[SUB_MAX_N];
  __shared__ double s_s[NG][4];
  __shared__ double s_d[NG][SUB_LM * LG + 1];
  if (a.dpar) {
    if (a.dpar[2] != 0.0) return;
    a.alpha = uniform_f64(a.dpar[0]);
    a.inv_tt = uniform_f64(a.dpar[1]);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int sub = lane & (LG - 1), grp = (tid >> 6) * (64 / LG) + lane / LG;   // lane inside its group; group of the workgroup
  const int64_t n = a.n, ld = a.ld;
  const int chunk = blockIdx.x;
  const int64_t j0 = uniform64(a.c0[chunk]), j1 = uniform64(a.c1[chunk]);

  // rows of this lane: pair k covers 2 (sub + LG k), + 1
  unsigned keep = 0u;          // bit 2k + h: a real, included sample
  int64_t off[KR];             // element offset of the pair inside a column (clamped inside the allocation)
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int64_t r = 2 * ((int64_t)sub + LG * k);
    off[k] = (r < ld) ? r : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bool m = r + h < n;
      if (STD && m && a.rowmask) m = a.rowmask[r + h] != 0.0;
      keep |= (m ? 1u : 0u) << (2 * k + h);
    }
  }
  auto real = [&](int i) { return ((keep >> i) & 1u) != 0u; };
  auto vec = [&](double(&o)[R], const double* p) {
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t r = 2 * ((int64_t)sub + LG * k) + h;
        o[2 * k + h] = (r < n) ? p[r] : 0.0;
      }
  };
  // The fused loadings + sweep flavour with many row pairs per lane would hold x[2], ts, v and acc in registers
  // (290 VGPRs at KR = 8: one wave per SIMD); its two n-vectors live in LDS instead, as (ts_r, v_r) pairs -- every
  // lane re-reads its own rows each round (the groups of a wave read the same addresses: broadcast).
  constexpr bool VEC_LDS = DEFL && SWEEP && KR >= 6;
  __shared__ double2 s_tv[VEC_LDS ? SUB_MAX_N : 1];
  double v[(SWEEP && !VEC_LDS) ? R : 1], ts[(DEFL && !VEC_LDS) ? R : 1], acc[SWEEP ? R : 1];
  if constexpr (VEC_LDS) {
    for (int r = tid; r < SUB_MAX_N; r += 256) s_tv[r] = (r < n) ? make_double2(a.ts[r], a.v[r]) : make_double2(0.0, 0.0);
    __syncthreads();
  } else {
    if constexpr (SWEEP) vec(v, a.v);
    if constexpr (DEFL) vec(ts, a.ts);
  }
  if constexpr (SWEEP) {
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.0;
  }
  // (ts, v) of this lane's i-th row
  auto tv = [&](int i) -> double2 {
    if constexpr (VEC_LDS) return s_tv[2 * (sub + LG * (i >> 1)) + (i & 1)];
    else return make_double2(DEFL ? ts[DEFL ? i : 0] : 0.0, SWEEP ? v[SWEEP ? i : 0] : 0.0);
  };
  double tsv = 0.0;
  if constexpr (DEFL && SWEEP) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const double2 q = tv(i);
      tsv = fma(q.x, q.y, tsv);
    }
    tsv = group_sum<LG>(tsv);
  }
  const double dn = (STD && a.rowmask) ? a.n_eff : (double)n;
  // lazy deflation: lane sub carries the corrections of the previous components sub, sub + 16, sub + 32
  const bool lazy = !STD && a.lazy != 0;
  double cu[SUB_LM], ce[SUB_LM], dacc[SUB_LM];
  const double* prow[SUB_LM];
  bool mine[SUB_LM];
#pragma unroll
  for (int m = 0; m < SUB_LM; ++m) {
    const int i = sub + LG * m;
    mine[m] = lazy && i < a.kprev;
    cu[m] = (mine[m] && a.cu) ? a.cu[i] : 0.0;
    ce[m] = (mine[m] && a.ce) ? a.ce[i] : 0.0;
    prow[m] = mine[m] ? a.Pm + (int64_t)i * a.ldp : nullptr;
    dacc[m] = 0.0;
  }
  double dk = 0.0, ssq_g = 0.0, ssq_p = 0.0, ssq_x = 0.0, amp = 0.0;

  // every row group walks the same number of rounds; a group past the end of the chunk re-reads the last column
  // and contributes nothing (no divergence inside the wave, the DPP rows stay whole)
  double x[D][R], pp[D][SUB_LM], k0v[D], muv[D], r2v[D];
  auto fetch = [&](int d, int64_t jr) {
    const int64_t j = (jr < j1) ? jr : j1 - 1;
    const double* col = a.src + j * ld;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const v2d_t q = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(col + off[k]));
      x[d][2 * k] = q.x;
      x[d][2 * k + 1] = q.y;
    }
#pragma unroll
    for (int m = 0; m < SUB_LM; ++m) pp[d][m] = mine[m] ? prow[m][j] : 0.0;
    k0v[d] = (STD && ISTD) ? col[0] : 0.0;
    muv[d] = CEN ? a.mu[j] : 0.0;
    // 1 / scale_j^2 of the raw-space passes travels WITH the column: a few dozen samples are reduced in ~50 cycles, far
    // less than the latency of a load issued when the round begins -- every round then stalled on this one scalar
    // (rocprofv3 --pmc, n = 40 / 128: SQ_WAIT_ANY 0.73-0.76 of the wave cycles, SQ_ACTIVE_INST_ANY 0.21-0.24)
    r2v[d] = (ISTD && !STD) ? a.ir2[j] : 1.0;
  };
  const int64_t rounds = (j1 - j0 + NG - 1) / NG;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < rounds) fetch(d, j0 + grp + NG * d);

  for (int64_t rd = 0; rd < rounds; rd += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (rd + d >= rounds) break;
      const int64_t jj = j0 + grp + NG * (rd + d);
      const bool valid = jj < j1;                 // uniform over the group
      const bool writer = valid && sub == 0;
      double(&xc)[R] = x[d];
      constexpr bool NOSTORE = STD && ISTD && SWEEP;
      bool handled = false;
      double gj = 0.0, pjk = 0.0, cj = 0.0;

      if constexpr (NOSTORE) {
        if (a.mode == MBPLS_SCALE_FIT) {
          // statistics-only first pass: shifted one-pass sums (shift = first sample), nothing stored
          const double K0 = k0v[d];
          double s1 = 0.0, s2 = 0.0, sv = 0.0;
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const double e = real(i) ? xc[i] - K0 : 0.0;
            xc[i] = e;
            s1 += e;
            s2 = fma(e, e, s2);
            sv = fma(e, v[i], sv);
          }
          s1 = group_sum<LG>(s1);
          s2 = group_sum<LG>(s2);
          sv = group_sum<LG>(sv);
          const double inv_n = 1.0 / dn;
          const double m1 = s1 * inv_n;
          const double mean = K0 + m1;
          const double var = fmax((s2 - s1 * m1) * inv_n, 0.0);
          const double eps = 2.220446049250313e-16;
          const double nme = dn * mean * eps;
          const bool constant = var <= dn * eps * var + nme * nme;
          double rv = __builtin_amdgcn_rcp(var);
          rv = fma(fma(-var, rv, 1.0), rv, rv);
          rv = fma(fma(-var, rv, 1.0), rv, rv);
          const double r2 = constant ? 0.0 : rv;
          const double gt = a.alpha * sv;
          cj = valid ? gt * r2 : 0.0;
          ssq_g = fma(gt, cj, ssq_g);
          if (valid) amp = fmax(amp, mean * mean * r2);
          if (writer) {
            ssq_x = fma(dn * var, r2, ssq_x);     // sum z^2 of the column, counted by one lane of the row
            ssq_x += s2 - s2;                     // non-finite samples poison the sum (fmax() made their variance 0)
            a.mean[jj] = mean;
            a.scale[jj] = var;                    // k_finish_stats turns var into scale, g~ into g
            if (a.r2_out) a.r2_out[jj] = r2;
            if (a.g) a.g[jj] = gt;
          }
#pragma unroll
          for (int i = 0; i < R; ++i) acc[i] = fma(cj, xc[i], acc[i]);
          handled = true;
        }
      }

      if constexpr (STD) {
        if (!handled && a.mode != MBPLS_SCALE_NONE) {
          double mean, scale;
          if (a.mode == MBPLS_SCALE_FIT) {
            // sklearn _incremental_mean_and_var: mean, then corrected two-pass variance
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) s += real(i) ? xc[i] : 0.0;
            s = group_sum<LG>(s);
            mean = s / dn;
            double c0 = 0.0, c1 = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
              const double dlt = real(i) ? xc[i] - mean : 0.0;
              xc[i] = dlt;
              c0 += dlt;
              c1 = fma(dlt, dlt, c1);
            }
            c0 = group_sum<LG>(c0);
            c1 = group_sum<LG>(c1);
            const double var = (c1 - c0 * c0 / dn) / dn;
            const double eps = 2.220446049250313e-16;
            const double nme = dn * mean * eps;
            const double bound = dn * eps * var + nme * nme;
            scale = (var <= bound) ? 1.0 : sqrt(var);
            if (valid) amp = fmax(amp, fabs(mean) / scale);
            if (writer) {
              a.mean[jj] = mean;
              a.scale[jj] = scale;
              if (a.var_out) a.var_out[jj] = (var <= bound) ? 0.0 : var;
            }
          } else {
            mean = a.mean[valid ? jj : j1 - 1];
            scale = a.scale[valid ? jj : j1 - 1];
#pragma unroll
            for (int i = 0; i < R; ++i) xc[i] = real(i) ? xc[i] - mean : 0.0;
          }
#pragma unroll
          for (int i = 0; i < R; ++i) xc[i] = xc[i] / scale;
        } else if (!handled) {
#pragma unroll
          for (int i = 0; i < R; ++i) xc[i] = real(i) ? xc[i] : 0.0;
        }
        if (!handled) {
          double s = 0.0;
#pragma unroll
          for (int i = 0; i < R; ++i) s = fma(xc[i], xc[i], s);
          if (valid) ssq_x += s;                  // per-lane partial, summed over the row at the end
        }
        if constexpr (!NOSTORE) {
          if (valid && a.dst) {
            double* out = a.dst + jj * ld;
#pragma unroll
            for (int k = 0; k < KR; ++k) {
              const int64_t r = 2 * ((int64_t)sub + LG * k);
              if (r + 1 < ld) {
                v2d_t q;
                q.x = xc[2 * k];
                q.y = xc[2 * k + 1];
                __builtin_nontemporal_store(q, reinterpret_cast<v2d_t*>(out + r));
              } else if (r < ld) {
                out[r] = xc[2 * k];
              }
            }
          }
        }
      } else {
        // rows beyond n read as 0 (an explicitly deflated column stores zeros back into its padding)
#pragma unroll
        for (int i = 0; i < R; ++i) xc[i] = real(i) ? xc[i] : 0.0;
      }

      double r2c = 1.0;
      if constexpr (ISTD && !STD) {
        r2c = r2v[d];
        if constexpr (CEN) {
          const double muj = muv[d];
#pragma unroll
          for (int i = 0; i < R; ++i) xc[i] -= muj;
        }
      }
      if constexpr (DEFL) {
        double ab0 = 0.0, ab1 = 0.0;
#pragma unroll
        for (int m = 0; m < SUB_LM; ++m) {
          ab0 = fma(-pp[d][m], ce[m], ab0);
          ab1 = fma(-pp[d][m], cu[m], ab1);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
          // LDS form: fetch the vectors four rows at a time, just before use (a compiler barrier keeps it from
          // hoisting all 2R values into registers again, which is what the LDS copy is there to avoid)
          if constexpr (VEC_LDS)
            if ((i & 3) == 0) asm volatile("" ::: "memory");
          const double2 q = tv(i);
          ab0 = fma(xc[i], q.x, ab0);
          if constexpr (SWEEP) ab1 = fma(xc[i], q.y, ab1);
        }
        ab0 = group_sum<LG>(ab0);
        if constexpr (SWEEP) ab1 = group_sum<LG>(ab1);
        const double pj = ab0 * a.inv_tt;
        if (valid) ssq_p = fma(pj * pj, (ISTD ? r2c : 1.0), ssq_p);
        if (writer && a.pvec) a.pvec[jj] = pj;
        if (a.deflate) {
          if (!lazy) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
              if constexpr (VEC_LDS)
                if ((i & 3) == 0) asm volatile("" ::: "memory");
              xc[i] = fma(-pj, tv(i).x, xc[i]);
            }
            if (valid) {
              double* out = a.dst + jj * ld;
#pragma unroll
              for (int k = 0; k < KR; ++k) {
                const int64_t r = 2 * ((int64_t)sub + LG * k);
                if (r + 1 < ld) {
                  v2d_t q;
                  q.x = xc[2 * k];
                  q.y = xc[2 * k + 1];
                  __builtin_nontemporal_store(q, reinterpret_cast<v2d_t*>(out + r));
                } else if (r < ld) {
                  out[r] = xc[2 * k];
                }
              }
            }
          }
          if constexpr (SWEEP) gj = a.alpha * fma(-pj, tsv, ab1);
          pjk = pj;
        } else {
          if constexpr (SWEEP) gj = a.alpha * ab1;
        }
      } else if constexpr (SWEEP) {
        if (!handled) {
          double b = 0.0;
#pragma unroll
          for (int m = 0; m < SUB_LM; ++m) b = fma(-pp[d][m], cu[m], b);
#pragma unroll
          for (int i = 0; i < R; ++i) b = fma(xc[i], v[i], b);
          b = group_sum<LG>(b);
          gj = a.alpha * b;
        }
      }
      if constexpr (SWEEP) {
        if (!handled) {
          cj = gj;
          if constexpr (ISTD && !STD) cj = gj * r2c;
          if (!valid) cj = 0.0;
          ssq_g = fma(gj, cj, ssq_g);
          if (writer && a.g) a.g[jj] = gj;
#pragma unroll
          for (int i = 0; i < R; ++i) acc[i] = fma(cj, xc[i], acc[i]);
        }
#pragma unroll
        for (int m = 0; m < SUB_LM; ++m) dacc[m] = fma(cj, pp[d][m], dacc[m]);
        if constexpr (DEFL) dk = fma(cj, pjk, dk);
      }
      if (rd + d + D < rounds) fetch(d, jj + NG * D);
    }
  }

  // ---- the 16 row groups of the workgroup combine in a fixed order
  if constexpr (STD) ssq_x = group_sum<LG>(ssq_x);
  if (sub == 0) {
    s_s[grp][0] = ssq_g;
    s_s[grp][1] = ssq_p;
    s_s[grp][2] = ssq_x;
    s_s[grp][3] = amp;
    s_d[grp][SUB_LM * LG] = dk;
  }
#pragma unroll
  for (int m = 0; m < SUB_LM; ++m) s_d[grp][sub + LG * m] = dacc[m];
  if constexpr (SWEEP) {
    for (int i = tid; i < SUB_MAX_N; i += 256) s_t[i] = 0.0;
    __syncthreads();
    for (int g = 0; g < NG; ++g) {
      if (grp == g) {
#pragma unroll
        for (int k = 0; k < KR; ++k)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = 2 * (sub + LG * k) + h;
            if (r < n) s_t[r] += acc[2 * k + h];
          }
      }
      __syncthreads();
    }
    double* out = a.part_t + (int64_t)chunk * n;
    for (int i = tid; i < n; i += 256) out[i] = s_t[i];
  } else {
    __syncthreads();
  }
  if (tid == 0) {
    double sg = 0.0, sp = 0.0, sx = 0.0, am = 0.0;
    for (int g = 0; g < NG; ++g) {
      sg += s_s[g][0];
      sp += s_s[g][1];
      sx += s_s[g][2];
      am = fmax(am, s_s[g][3]);
    }
    double* ps = a.part_s + (int64_t)chunk * 4;
    ps[0] = sg;
    ps[1] = sp;
    ps[2] = sx;
    ps[3] = am;
  }
  if (lazy && a.part_d) {
    double* pd = a.part_d + (int64_t)chunk * KD;
    if (tid < a.kprev) {
      double s = 0.0;
      for (int g = 0; g < NG; ++g) s += s_d[g][tid];
      pd[tid] = s;
    }
    if (tid == 0) {
      double s = 0.0;
      for (int g = 0; g < NG; ++g) s += s_d[g][SUB_LM * LG];
      pd[a.kprev] = s;
    }
  }
}

template <int LG, int KR, int D, bool STD, bool DEFL, bool SWEEP>
static void launch_sub_kr(mbpls_ctx* ctx, const ChunkTable& ct, const PassArgs& a) {
  dim3 grid((unsigned)ct.nchunks), block(256);
  const bool istd = STD ? (SWEEP && a.dst == nullptr) : (a.lazy && a.ir2 != nullptr);
  const bool cen = !STD && istd && a.mu != nullptr;
  constexpr bool IS_ = !STD || SWEEP;
  constexpr bool CE_ = !STD;
  // one correction slot per lane covers every pass that is not lazy and every lazy one with at most LG earlier components
  const bool lm1 = STD || !a.lazy || a.kprev <= LG;
  if (lm1) {
    if (cen) hipLaunchKernelGGL((k_colpass_sub<LG, KR, D, STD, DEFL, SWEEP, IS_, CE_, true>), grid, block, 0, ctx->stream, a);
    else if (istd) hipLaunchKernelGGL((k_colpass_sub<LG, KR, D, STD, DEFL, SWEEP, IS_, false, true>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_colpass_sub<LG, KR, D, STD, DEFL, SWEEP, false, false, true>), grid, block, 0, ctx->stream, a);
    return;
  }
  if constexpr (!STD) {
    if (cen) hipLaunchKernelGGL((k_colpass_sub<LG, KR, D, STD, DEFL, SWEEP, IS_, CE_>), grid, block, 0, ctx->stream, a);
    else if (istd) hipLaunchKernelGGL((k_colpass_sub<LG, KR, D, STD, DEFL, SWEEP, IS_, false>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_colpass_sub<LG, KR, D, STD, DEFL, SWEEP, false, false>), grid, block, 0, ctx->stream, a);
  }
}

// (lanes per column, row pairs per lane) by n: the tightest cover of the column -- lanes past the end of a column
// still issue their (clamped, cached) loads and their share of every reduction.  64 < n <= 128: the statistics and
// sweep-only flavours gain 5-19 % from 8-lane groups, the fused loadings + sweep (ts and v in registers) loses 10-15 %.
template <bool STD, bool DEFL, bool SWEEP>
static void launch_sub(mbpls_ctx* ctx, const ChunkTable& ct, const PassArgs& a) {
  const int64_t n = a.n;
  constexpr bool NARROW = !DEFL;
  if (n <= 32) launch_sub_kr<8, 2, 4, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 48) launch_sub_kr<8, 3, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 64) launch_sub_kr<8, 4, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 96 && NARROW) launch_sub_kr<8, 6, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 96) launch_sub_kr<16, 3, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 128 && NARROW) launch_sub_kr<8, 8, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 128) launch_sub_kr<16, 4, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 192) launch_sub_kr<16, 6, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 256) launch_sub_kr<16, 8, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 384) launch_sub_kr<32, 6, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 512) launch_sub_kr<32, 8, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else if (n <= 768) launch_sub_kr<64, 6, 2, STD, DEFL, SWEEP>(ctx, ct, a);
  else launch_sub_kr<64, 8, 2, STD, DEFL, SWEEP>(ctx, ct, a);
}

int launch_pass_sub(mbpls_ctx* ctx, const ChunkTable& ct, const PassArgs& a, bool STD, bool DEFL, bool SWEEP) {
  if (STD && !DEFL && SWEEP) launch_sub<true, false, true>(ctx, ct, a);
  else if (STD && !DEFL) launch_sub<true, false, false>(ctx, ct, a);
  else if (DEFL && SWEEP) launch_sub<false, true, true>(ctx, ct, a);
  else if (DEFL) launch_sub<false, true, false>(ctx, ct, a);
  else launch_sub<false, false, true>(ctx, ct, a);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

// ------------------------------------------------------------------------------------------- missing values
// k_colpass_nan with the same column groups: a NaN marks a missing entry, stays NaN in HBM, and every sum skips it.
template <int LG, int KR, int D, bool STD, bool DEFL, bool SWEEP>
__global__ __launch_bounds__(256) void k_colpass_nan_sub(NanArgs a) {
  constexpr int R = 2 * KR;
  constexpr int NG = 256 / LG;
  __shared__ double s_t[SWEEP ? SUB_MAX_N : 1], s_dd[SWEEP ? SUB_MAX_N : 1];
  __shared__ double s_s[NG][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int sub = lane & (LG - 1), grp = (tid >> 6) * (64 / LG) + lane / LG;
  const int64_t n = a.n, ld = a.ld;
  const int chunk = blockIdx.x;
  const int64_t j0 = uniform64(a.c0[chunk]), j1 = uniform64(a.c1[chunk]);
  unsigned keep = 0u;
  int64_t off[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int64_t r = 2 * ((int64_t)sub + LG * k);
    off[k] = (r < ld) ? r : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bool m = r + h < n;
      if (STD && m && a.rowmask) m = a.rowmask[r + h] != 0.0;   // row-subset fit: excluded samples read as 0
      keep |= (m ? 1u : 0u) << (2 * k + h);
    }
  }
  auto real = [&](int i) { return ((keep >> i) & 1u) != 0u; };
  auto vec = [&](double(&o)[R], const double* p) {
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t r = 2 * ((int64_t)sub + LG * k) + h;
        o[2 * k + h] = (r < n) ? p[r] : 0.0;
      }
  };
  // many row pairs per lane: the n-vectors live in LDS as (ts_r, v_r) pairs and are fetched just before use (see
  // k_colpass_sub): x[2], acc, dac and both vectors in registers cost 280-306 VGPRs, one wave per SIMD
  constexpr bool VEC_LDS = KR >= 6 && (SWEEP || DEFL);
  __shared__ double2 s_tv[VEC_LDS ? SUB_MAX_N : 1];
  double v[(SWEEP && !VEC_LDS) ? R : 1], ts[(DEFL && !VEC_LDS) ? R : 1], acc[SWEEP ? R : 1], dac[SWEEP ? R : 1];
  if constexpr (VEC_LDS) {
    for (int r = tid; r < SUB_MAX_N; r += 256)
      s_tv[r] = (r < n) ? make_double2(DEFL ? a.ts[r] : 0.0, SWEEP ? a.v[r] : 0.0) : make_double2(0.0, 0.0);
    __syncthreads();
  } else {
    if constexpr (SWEEP) vec(v, a.v);
    if constexpr (DEFL) vec(ts, a.ts);
  }
  if constexpr (SWEEP) {
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = dac[i] = 0.0;
  }
  auto tv = [&](int i) -> double2 {
    if constexpr (VEC_LDS) {
      if ((i & 3) == 0) asm volatile("" ::: "memory");      // keeps the compiler from hoisting all 2R values again
      return s_tv[2 * (sub + LG * (i >> 1)) + (i & 1)];
    } else {
      return make_double2(DEFL ? ts[DEFL ? i : 0] : 0.0, SWEEP ? v[SWEEP ? i : 0] : 0.0);
    }
  };
  double ssq_g = 0.0, ssq_p = 0.0, ssq_x = 0.0;

  double x[D][R];
  auto fetch = [&](int d, int64_t jr) {
    const double* col = a.src + ((jr < j1) ? jr : j1 - 1) * ld;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const v2d_t q = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(col + off[k]));
      x[d][2 * k] = q.x;
      x[d][2 * k + 1] = q.y;
    }
  };
  auto store = [&](const double(&xc)[R], int64_t jj) {
    double* out = a.dst + jj * ld;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int64_t r = 2 * ((int64_t)sub + LG * k);
      if (r + 1 < ld) {
        v2d_t q;
        q.x = xc[2 * k];
        q.y = xc[2 * k + 1];
        __builtin_nontemporal_store(q, reinterpret_cast<v2d_t*>(out + r));
      } else if (r < ld) {
        out[r] = xc[2 * k];
      }
    }
  };
  const int64_t rounds = (j1 - j0 + NG - 1) / NG;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < rounds) fetch(d, j0 + grp + NG * d);

  for (int64_t rd = 0; rd < rounds; rd += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (rd + d >= rounds) break;
      const int64_t jj = j0 + grp + NG * (rd + d);
      const bool valid = jj < j1;
      const bool writer = valid && sub == 0;
      const int64_t jc = valid ? jj : j1 - 1;
      double(&xc)[R] = x[d];
      // rows beyond n (and excluded samples) read as 0: present, and without weight in any sum
      unsigned pm = 0u;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        xc[i] = real(i) ? xc[i] : 0.0;
        pm |= ((xc[i] == xc[i]) ? 1u : 0u) << i;
      }
      auto present = [&](int i) -> bool { return (pm >> i) & 1u; };

      if constexpr (STD) {
        if (a.mode != MBPLS_SCALE_FIT && a.count) {
          double c = 0.0;
#pragma unroll
          for (int i = 0; i < R; ++i) c += (present(i) && real(i)) ? 1.0 : 0.0;
          c = group_sum<LG>(c);
          if (writer) a.count[jj] = c;
        }
        if (a.mode != MBPLS_SCALE_NONE) {
          double mean, scale;
          if (a.mode == MBPLS_SCALE_FIT) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
              const bool use = present(i) && real(i);
              s0 += use ? xc[i] : 0.0;
              s1 += use ? 1.0 : 0.0;
            }
            s0 = group_sum<LG>(s0);
            s1 = group_sum<LG>(s1);
            const double cnt = s1;
            mean = s0 / cnt;
            double c0 = 0.0, c1 = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
              const bool use = present(i) && real(i);
              const double dlt = use ? xc[i] - mean : 0.0;
              if (real(i)) xc[i] = xc[i] - mean;    // NaN stays NaN
              c0 += dlt;
              c1 = fma(dlt, dlt, c1);
            }
            c0 = group_sum<LG>(c0);
            c1 = group_sum<LG>(c1);
            const double var = (c1 - c0 * c0 / cnt) / cnt;
            const double eps = 2.220446049250313e-16;
            const double nme = cnt * mean * eps;
            const double bound = cnt * eps * var + nme * nme;
            scale = (var <= bound) ? 1.0 : sqrt(var);
            if (writer) {
              a.mean[jj] = mean;
              a.scale[jj] = scale;
              if (a.count) a.count[jj] = cnt;
              if (a.var_out) a.var_out[jj] = (var <= bound) ? 0.0 : var;
            }
          } else {
            mean = a.mean[jc];
            scale = a.scale[jc];
#pragma unroll
            for (int i = 0; i < R; ++i)
              if (real(i)) xc[i] = xc[i] - mean;
          }
#pragma unroll
          for (int i = 0; i < R; ++i) xc[i] = xc[i] / scale;
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const double z = present(i) ? xc[i] : 0.0;
          s = fma(z, z, s);                          // nansum(X^2), mbpls.py:826
        }
        if (valid) {
          ssq_x += s;
          store(xc, jj);
        }
      }

      double gj = 0.0;
      if constexpr (DEFL) {
        double r5[SWEEP ? 5 : 2];
#pragma unroll
        for (int q = 0; q < (SWEEP ? 5 : 2); ++q) r5[q] = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const bool pr = present(i);
          const double2 q2 = tv(i);
          const double xz = pr ? xc[i] : 0.0;
          const double tz = pr ? q2.x : 0.0;
          r5[0] = fma(xz, q2.x, r5[0]);
          r5[1] = fma(tz, q2.x, r5[1]);
          if constexpr (SWEEP) {
            const double vz = pr ? q2.y : 0.0;
            r5[2] = fma(xz, q2.y, r5[2]);
            r5[3] = fma(vz, q2.y, r5[3]);
            r5[4] = fma(tz, q2.y, r5[4]);
          }
        }
#pragma unroll
        for (int q = 0; q < (SWEEP ? 5 : 2); ++q) r5[q] = group_sum<LG>(r5[q]);
        const double pj = (r5[1] != 0.0) ? r5[0] / r5[1] : 0.0;
        if (valid) ssq_p = fma(pj, pj, ssq_p);
        if (writer && a.pvec) a.pvec[jj] = pj;
        if (a.deflate) {
#pragma unroll
          for (int i = 0; i < R; ++i) xc[i] = fma(-pj, tv(i).x, xc[i]);   // NaN stays NaN (:969)
          if (valid) store(xc, jj);
          if constexpr (SWEEP) gj = (r5[3] != 0.0) ? fma(-pj, r5[4], r5[2]) / r5[3] : 0.0;
        } else {
          if constexpr (SWEEP) gj = (r5[3] != 0.0) ? r5[2] / r5[3] : 0.0;
        }
      } else if constexpr (SWEEP) {
        double r0 = 0.0, r1 = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const bool pr = present(i);
          const double vi = tv(i).y;
          r0 = fma(pr ? xc[i] : 0.0, vi, r0);
          r1 = fma(pr ? vi : 0.0, vi, r1);
        }
        r0 = group_sum<LG>(r0);
        r1 = group_sum<LG>(r1);
        gj = (r1 != 0.0) ? r0 / r1 : 0.0;
      }
      if constexpr (SWEEP) {
        if (!valid) gj = 0.0;
        ssq_g = fma(gj, gj, ssq_g);
        if (writer && a.g) a.g[jj] = gj;
        const double g2 = gj * gj;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const bool pr = present(i);
          acc[i] = fma(gj, pr ? xc[i] : 0.0, acc[i]);
          dac[i] += pr ? g2 : 0.0;
        }
      }
      if (rd + d + D < rounds) fetch(d, jj + NG * D);
    }
  }

  if constexpr (STD) ssq_x = group_sum<LG>(ssq_x);
  if (sub == 0) {
    s_s[grp][0] = ssq_g;
    s_s[grp][1] = ssq_p;
    s_s[grp][2] = ssq_x;
  }
  if constexpr (SWEEP) {
    for (int i = tid; i < SUB_MAX_N; i += 256) s_t[i] = s_dd[i] = 0.0;
    __syncthreads();
    for (int g = 0; g < NG; ++g) {
      if (grp == g) {
#pragma unroll
        for (int k = 0; k < KR; ++k)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = 2 * (sub + LG * k) + h;
            if (r < n) {
              s_t[r] += acc[2 * k + h];
              s_dd[r] += dac[2 * k + h];
            }
          }
      }
      __syncthreads();
    }
    double* out = a.part_t + (int64_t)chunk * n;
    double* outd = a.part_d + (int64_t)chunk * n;
    for (int i = tid; i < n; i += 256) {
      out[i] = s_t[i];
      outd[i] = s_dd[i];
    }
  } else {
    __syncthreads();
  }
  if (tid == 0) {
    double sg = 0.0, sp = 0.0, sx = 0.0;
    for (int g = 0; g < NG; ++g) {
      sg += s_s[g][0];
      sp += s_s[g][1];
      sx += s_s[g][2];
    }
    double* ps = a.part_s + (int64_t)chunk * 4;
    ps[0] = sg;
    ps[1] = sp;
    ps[2] = sx;
    ps[3] = 0.0;
  }
}

template <bool STD, bool DEFL, bool SWEEP>
static void launch_nan_sub_t(mbpls_ctx* ctx, const ChunkTable& ct, const NanArgs& a) {
  dim3 grid((unsigned)ct.nchunks), block(256);
  const int64_t n = a.n;
#define MB_NS(LG_, KR_, D_) \
  hipLaunchKernelGGL((k_colpass_nan_sub<LG_, KR_, D_, STD, DEFL, SWEEP>), grid, block, 0, ctx->stream, a)
  if (n <= 32) MB_NS(8, 2, 4);
  else if (n <= 48) MB_NS(8, 3, 2);
  else if (n <= 64) MB_NS(8, 4, 2);
  else if (n <= 96) MB_NS(16, 3, 2);
  else if (n <= 128) MB_NS(16, 4, 2);
  else if (n <= 192) MB_NS(16, 6, 2);
  else if (n <= 256) MB_NS(16, 8, 2);
  else if (n <= 384) MB_NS(32, 6, 2);
  else if (n <= 512) MB_NS(32, 8, 2);
  else if (n <= 768) MB_NS(64, 6, 2);
  else MB_NS(64, 8, 2);
#undef MB_NS
}

int launch_nan_sub(mbpls_ctx* ctx, const ChunkTable& ct, const NanArgs& a, bool STD, bool DEFL, bool SWEEP) {
  if (STD && SWEEP) launch_nan_sub_t<true, false, true>(ctx, ct, a);
  else if (STD) launch_nan_sub_t<true, false, false>(ctx, ct, a);
  else if (DEFL && SWEEP) launch_nan_sub_t<false, true, true>(ctx, ct, a);
  else if (DEFL) launch_nan_sub_t<false, true, false>(ctx, ct, a);
  else launch_nan_sub_t<false, false, true>(ctx, ct, a);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

}  // namespace mbpls
