// Missing-value column passes for LONG columns (n > ctx->nan_long_from, default 8192) as TWO STREAMING PHASES around a
// p-sized finishing step -- the missing-value counterpart of colpass_long.hip (reference: mbpls/mbpls.py:846-854 weights,
// :865-873 block scores, :919-926 loadings, :969 deflation; sums skip NaN, NaN stays NaN in HBM).
//
// A column of more than 8192 samples does not fit the registers of a workgroup; the first version of these passes
// (k_colpass_nan_long in colpass_nan.hip, kept behind the tunable "nan_long_two_phase" = 0) let ONE workgroup stream a
// column several times -- once per dependent reduction -- with the chunk's partial score vectors A / D in the workspace:
// from 16 384 samples on the re-visits miss L2 (two 1024-thread workgroups per CU x four columns x 128 KB in flight) and
// the passes ran at 1.0-2.2 TB/s of algorithmic bytes.  Here rows are the parallel dimension as well:
//
//   phase A  k_nl_slab   one WAVE per 1024-row slab and column group: the masked partial sums a column needs
//                        (statistics: count, shifted sums; dot products with ts and v over the present rows), 16-byte
//                        loads, DPP reductions, no barrier, no LDS                                     -- one read
//   finish   k_nl_fin    one thread per column: fixed-order sum over the slabs, then mean / scale / p_j / g_j and the
//                        per-chunk sums of squares (second stage: reduce_chunks, as everywhere)        -- p-sized
//   phase B  k_nl_rows   two rows per thread, all columns of a coarse chunk: the stored column (standardised or
//                        deflated: NaN stays NaN) and A_i += g_j x_ij, D_i += g_j^2 over the present columns of the row,
//                        A / D in registers                                              -- one read (+ one write)
//
// Transfers of 8np bytes: loadings only 1; sweep 2 (algorithmic 1); loadings + deflation + sweep 2 reads + 1 write
// (algorithmic 2); standardise + sweep 2 reads + 1 write (algorithmic 2) -- every transfer a coalesced stream.
#include <algorithm>

#include "colpass_util.h"

namespace mbpls {

void launch_reduce_xw(mbpls_ctx* ctx, const mbpls_mat* m, const double* part, int CB, int cb, int per_block,
                      double* d_out, int64_t ldo, int c_total, int k0);

constexpr int NL_SLAB = 1024;   // rows per wave in phase A (16 per lane)
constexpr int NL_NV = 8;        // doubles per (slab, column) partial

enum { NL_STD = 0, NL_DEFL = 1, NL_DEFL_SWEEP = 2, NL_SWEEP = 3 };

// ------------------------------------------------------------------------------------------------ the shift
// Statistics in ONE read need shifted sums (S1 = sum (x - K), S2 = sum (x - K)^2 with K near the data); with missing
// values the column's first sample may be the missing one: K = its first PRESENT sample (0 for an empty column).
__global__ __launch_bounds__(256) void k_nl_first_present(const double* __restrict__ X, int64_t ld, int64_t n, int64_t p,
                                                          double* __restrict__ shift) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= p) return;
  const double* col = X + j * ld;
  double K = 0.0;
  for (int64_t i0 = 0; i0 < n; i0 += 64) {
    const int64_t i = i0 + lane;
    const double x = (i < n) ? col[i] : __longlong_as_double(0x7ff8000000000000ll);
    const unsigned long long present = __ballot(x == x);
    if (present) {
      K = __shfl(x, (int)__ffsll((long long)present) - 1, 64);
      break;
    }
  }
  if (lane == 0) shift[j] = K;
}

// ------------------------------------------------------------------------------------------------ phase A
struct NlSlabArgs {
  const double* src;
  int64_t ld, n, p;
  const double* v;         // sweep vector (NULL: none)
  const double* ts;        // score vector (DEFL kinds)
  const double* rowmask;   // standardise pass: excluded samples (0) take no part
  const double* shift;     // NL_STD: per-column shift K (NULL: 0)
  int cols_per_group, nslab;
  double* part;            // [nslab][p][NL_NV]
};

// NL_STD       : [cnt, S1, S2, Sv, Sv1, Svv]   e = x - K over the present, kept rows; Sv = sum e v, Sv1 = sum v, Svv = sum v^2
// NL_DEFL      : [<x,ts>, <ts,ts>]              over the present rows
// NL_DEFL_SWEEP: [<x,ts>, <ts,ts>, <x,v>, <v,v>, <ts,v>]
// NL_SWEEP     : [<x,v>, <v,v>]
template <int KIND, int D>
__global__ __launch_bounds__(256) void k_nl_slab(NlSlabArgs a) {
  constexpr bool HAS_TS = KIND == NL_DEFL || KIND == NL_DEFL_SWEEP;
  constexpr int NV = KIND == NL_STD ? 6 : (KIND == NL_DEFL_SWEEP ? 5 : 2);
  const int lane = threadIdx.x & 63;
  const int slab = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (slab >= a.nslab) return;                   // whole waves leave: no workgroup barrier below
  const int64_t n = a.n, ld = a.ld;
  const int64_t rb = (int64_t)slab * NL_SLAB + 2 * lane;
  const int64_t j0 = (int64_t)blockIdx.y * a.cols_per_group;
  const int64_t j1 = (j0 + a.cols_per_group < a.p) ? j0 + a.cols_per_group : a.p;
  double v[16], ts[HAS_TS ? 16 : 1];
  unsigned keep = 0u;
  int64_t off[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t r = rb + 128 * k;
    off[k] = (r < ld) ? r : 0;                   // out-of-allocation pairs re-read pair 0 (masked by `keep`)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bool m = r + h < n;
      if (KIND == NL_STD && m && a.rowmask) m = a.rowmask[r + h] != 0.0;
      v[2 * k + h] = (m && a.v) ? a.v[r + h] : 0.0;
      if constexpr (HAS_TS) ts[2 * k + h] = m ? a.ts[r + h] : 0.0;
      keep |= (m ? 1u : 0u) << (2 * k + h);
    }
  }
  double x[D][16];
  auto load = [&](double(&xc)[16], int64_t j) {
    const double* col = a.src + j * ld;          // wave-uniform
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const v2d_t q = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(col + off[k]));
      xc[2 * k] = q.x;
      xc[2 * k + 1] = q.y;
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (j0 + d < j1) load(x[d], j0 + d);
  for (int64_t j = j0; j < j1; j += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int64_t jj = j + d;
      if (jj >= j1) break;
      double s[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) s[q] = 0.0;
      if constexpr (KIND == NL_STD) {
        const double K0 = a.shift ? uniform_f64(a.shift[jj]) : 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const double xi = x[d][i];
          const bool pr = ((keep >> i) & 1u) && xi == xi;
          const double e = pr ? xi - K0 : 0.0;
          const double vz = pr ? v[i] : 0.0;
          s[0] += pr ? 1.0 : 0.0;
          s[1] += e;
          s[2] = fma(e, e, s[2]);
          s[3] = fma(e, v[i], s[3]);
          s[4] += vz;
          s[5] = fma(vz, v[i], s[5]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const double xi = x[d][i];
          const bool pr = ((keep >> i) & 1u) && xi == xi;
          const double xz = pr ? xi : 0.0;
          if constexpr (HAS_TS) {
            const double tz = pr ? ts[i] : 0.0;
            s[0] = fma(xz, ts[i], s[0]);
            s[1] = fma(tz, ts[i], s[1]);
            if constexpr (KIND == NL_DEFL_SWEEP) {
              const double vz = pr ? v[i] : 0.0;
              s[2] = fma(xz, v[i], s[2]);
              s[3] = fma(vz, v[i], s[3]);
              s[4] = fma(tz, v[i], s[4]);
            }
          } else {
            s[0] = fma(xz, v[i], s[0]);
            s[1] = fma(pr ? v[i] : 0.0, v[i], s[1]);
          }
        }
      }
      if (jj + D < j1) load(x[d], jj + D);       // refill before the reductions: the loads fly during them
#pragma unroll
      for (int q = 0; q < NV; ++q) s[q] = wave_sum_last_row(s[q]);
      if (lane == 63) {
        double2* o = reinterpret_cast<double2*>(a.part + ((int64_t)slab * a.p + jj) * NL_NV);
        o[0] = make_double2(s[0], s[1]);
        if constexpr (NV > 2) o[1] = make_double2(s[2], s[3]);
        if constexpr (NV == 5) o[2] = make_double2(s[4], 0.0);
        if constexpr (NV == 6) o[2] = make_double2(s[4], s[5]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ finish
struct NlFinArgs {
  const int64_t* c0;       // fine chunks
  const int64_t* c1;
  const double* part;      // [nslab][p][NL_NV]
  int nslab;
  int64_t p;
  int mode;                // NL_STD: MBPLS_SCALE_*
  int has_v, deflate;
  const double* shift;
  double* mean;
  double* scale;
  double* count;
  double* var_out;
  double* g;               // weights (optional output)
  double* pvec;            // loadings (DEFL kinds: always written -- the storing row pass reads them)
  double* wrow;            // coefficient of column j in the row pass (= g_j)
  double* part_s;          // [nchunks][4]: ssq_g, ssq_p, ssq_x, -
};

template <int KIND>
__global__ __launch_bounds__(256) void k_nl_fin(NlFinArgs a) {
  constexpr int NV = KIND == NL_STD ? 6 : (KIND == NL_DEFL_SWEEP ? 5 : 2);
  __shared__ double red[2 * 64];
  int buf = 0;
  const int chunk = blockIdx.x, tid = threadIdx.x;
  double acc[3] = {0.0, 0.0, 0.0};   // ssq_g, ssq_p, ssq_x
  for (int64_t j = a.c0[chunk] + tid; j < a.c1[chunk]; j += 256) {
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int sl = 0; sl < a.nslab; ++sl) {       // fixed order: repeated fits agree bit for bit
      const double2* q = reinterpret_cast<const double2*>(a.part + ((int64_t)sl * a.p + j) * NL_NV);
      const double2 q0 = q[0];
      s[0] += q0.x;
      s[1] += q0.y;
      if constexpr (NV > 2) {
        const double2 q1 = q[1];
        s[2] += q1.x;
        s[3] += q1.y;
      }
      if constexpr (NV > 4) {
        const double2 q2 = q[2];
        s[4] += q2.x;
        s[5] += q2.y;
      }
    }
    double gj = 0.0;
    if constexpr (KIND == NL_STD) {
      const double cnt = s[0], S1 = s[1], S2 = s[2], Sv = s[3], Sv1 = s[4], Svv = s[5];
      double scale = 1.0, dv = Sv, ssd = S2;
      if (a.mode == MBPLS_SCALE_FIT) {
        // sklearn StandardScaler on data with NaN (per-column counts), from the shifted one-pass sums
        const double K0 = a.shift[j];
        const double m1 = cnt > 0.0 ? S1 / cnt : 0.0;
        const double mean = K0 + m1;
        ssd = fmax(S2 - S1 * m1, 0.0);                       // sum (x - mean)^2 over the present rows
        const double var = cnt > 0.0 ? ssd / cnt : 0.0;
        const double eps = 2.220446049250313e-16;
        const double nme = cnt * mean * eps;
        const bool constant = var <= cnt * eps * var + nme * nme;
        scale = constant ? 1.0 : sqrt(var);
        dv = Sv - m1 * Sv1;                                   // <x - mean, v> over the present rows
        a.mean[j] = mean;
        a.scale[j] = scale;
        if (a.count) a.count[j] = cnt;
        if (a.var_out) a.var_out[j] = constant ? 0.0 : var;
      } else if (a.mode == MBPLS_SCALE_APPLY) {
        scale = a.scale[j];                                   // the shift was the given mean
        if (a.count) a.count[j] = cnt;
      }
      acc[2] += (ssd / scale) / scale;                        // nansum(z^2), mbpls.py:826
      if (a.has_v) gj = (Svv != 0.0) ? (dv / scale) / Svv : 0.0;
    } else if constexpr (KIND == NL_SWEEP) {
      gj = (s[1] != 0.0) ? s[0] / s[1] : 0.0;
    } else {
      const double pj = (s[1] != 0.0) ? s[0] / s[1] : 0.0;
      a.pvec[j] = pj;
      acc[1] = fma(pj, pj, acc[1]);
      if constexpr (KIND == NL_DEFL_SWEEP)
        gj = (s[3] != 0.0) ? (a.deflate ? fma(-pj, s[4], s[2]) : s[2]) / s[3] : 0.0;
    }
    if (KIND != NL_DEFL && (KIND != NL_STD || a.has_v)) {
      if (a.g) a.g[j] = gj;
      a.wrow[j] = gj;
      acc[0] = fma(gj, gj, acc[0]);
    }
  }
  block_sum<3>(acc, red, buf);
  if (tid == 0) {
    double* ps = a.part_s + (int64_t)chunk * 4;
    ps[0] = acc[0];
    ps[1] = acc[1];
    ps[2] = acc[2];
    ps[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------------ phase B
struct NlRowArgs {
  const double* src;
  double* dst;
  int64_t ld, n;
  const int64_t* c0;      // coarse chunks
  const int64_t* c1;
  const double* w;        // g_j
  const double* mu;       // per-column mean  (NR_STD_STORE, NULL: no centring)
  const double* sc;       // per-column scale
  const double* pv;       // per-column loading (NR_DEFL_STORE)
  const double* ts;       // per-row score
  const double* rowmask;
  double* part_t;         // [nchunks][n]  A
  double* part_d;         // [nchunks][n]  D
};

enum { NR_ACC = 0, NR_STD_STORE = 1, NR_DEFL_STORE = 2 };

// grid (coarse chunks, ceil(n/512)), 256 threads, two rows per thread: one 16-byte load per column and thread
template <int MODE, bool ACC>
__global__ __launch_bounds__(256) void k_nl_rows(NlRowArgs a) {
  const int chunk = blockIdx.x;
  const int64_t n = a.n, ld = a.ld;
  const int64_t r0 = (int64_t)blockIdx.y * 512 + 2 * threadIdx.x;
  const int64_t j0 = a.c0[chunk], j1 = a.c1[chunk];
  const bool in0 = r0 < n, in1 = r0 + 1 < n, inld = r0 < ld;
  bool k0 = in0, k1 = in1;                         // a sample excluded from the fit is stored as 0: present, without weight
  if (MODE == NR_STD_STORE && a.rowmask) {
    k0 = in0 && a.rowmask[r0] != 0.0;
    k1 = in1 && a.rowmask[r0 + 1] != 0.0;
  }
  double t0 = 0.0, t1 = 0.0;
  if constexpr (MODE == NR_DEFL_STORE) {
    t0 = in0 ? a.ts[r0] : 0.0;
    t1 = in1 ? a.ts[r0 + 1] : 0.0;
  }
  double acc0 = 0.0, acc1 = 0.0, d0 = 0.0, d1 = 0.0;
  const double* col = a.src + j0 * ld + r0;
  double* out = (MODE != NR_ACC) ? a.dst + j0 * ld + r0 : nullptr;
#pragma unroll 4
  for (int64_t j = j0; j < j1; ++j, col += ld) {
    v2d_t q = {0.0, 0.0};
    if (inld) q = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(col));
    double x0 = in0 ? q.x : 0.0, x1 = in1 ? q.y : 0.0;
    if constexpr (MODE == NR_STD_STORE) {
      if (a.mu) {
        const double mu = a.mu[j], sc = a.sc[j];      // workgroup-uniform -> scalar loads
        x0 = k0 ? (x0 - mu) / sc : 0.0;                // NaN stays NaN
        x1 = k1 ? (x1 - mu) / sc : 0.0;
      } else {
        x0 = k0 ? x0 : 0.0;
        x1 = k1 ? x1 : 0.0;
      }
    } else if constexpr (MODE == NR_DEFL_STORE) {
      const double pj = a.pv[j];
      x0 = fma(-pj, t0, x0);                           // NaN stays NaN (:969)
      x1 = fma(-pj, t1, x1);
    }
    if constexpr (MODE != NR_ACC) {
      if (inld) {
        v2d_t o;
        o.x = x0;
        o.y = x1;                                      // (the padding row of an odd n is stored as 0)
        __builtin_nontemporal_store(o, reinterpret_cast<v2d_t*>(out));
      }
      out += ld;
    }
    if constexpr (ACC) {
      const double gj = a.w[j], g2 = gj * gj;
      const bool p0 = x0 == x0, p1 = x1 == x1;
      acc0 = fma(gj, p0 ? x0 : 0.0, acc0);
      acc1 = fma(gj, p1 ? x1 : 0.0, acc1);
      d0 += p0 ? g2 : 0.0;
      d1 += p1 ? g2 : 0.0;
    }
  }
  if constexpr (ACC) {
    double* o = a.part_t + (int64_t)chunk * n;
    double* od = a.part_d + (int64_t)chunk * n;
    if (in0) {
      o[r0] = acc0;
      od[r0] = d0;
    }
    if (in1) {
      o[r0 + 1] = acc1;
      od[r0 + 1] = d1;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
struct NlScratch {
  double *part, *shift, *wrow, *ptmp, *part_s;
  int nslab;
};

static int nl_scratch(mbpls_ctx* ctx, const mbpls_mat* m, NlScratch* s) {
  const size_t p = (size_t)m->p, nc = (size_t)m->fine.nchunks;
  s->nslab = (int)((m->n + NL_SLAB - 1) / NL_SLAB);
  auto up = [](size_t v) { return (v + 31) / 32 * 32; };   // 256-byte granules
  size_t o = 0;
  const size_t oshift = o; o += up(p);
  const size_t owrow = o; o += up(p);
  const size_t optmp = o; o += up(p);
  const size_t ops = o; o += up(nc * 4);
  const size_t opart = o; o += up(p * (size_t)s->nslab * NL_NV);
  MB_TRY(ensure_stage(ctx, sizeof(double) * o));
  double* b = (double*)ctx->stage;
  s->shift = b + oshift; s->wrow = b + owrow; s->ptmp = b + optmp; s->part_s = b + ops; s->part = b + opart;
  return MBPLS_OK;
}

template <int KIND>
static int nl_phase_a(mbpls_ctx* ctx, const mbpls_mat* m, const NlScratch& s, const double* d_v, const double* d_ts,
                      const double* d_rowmask, const double* d_shift) {
  NlSlabArgs a{};
  a.src = m->data; a.ld = m->ld; a.n = m->n; a.p = m->p;
  a.v = d_v; a.ts = d_ts; a.rowmask = d_rowmask; a.shift = d_shift;
  a.nslab = s.nslab; a.part = s.part;
  const int64_t want_groups = std::max<int64_t>(1, ((int64_t)ctx->cus * 32 + s.nslab - 1) / s.nslab);
  const int64_t groups0 = std::min<int64_t>({want_groups, m->p, (int64_t)65535});
  a.cols_per_group = (int)((m->p + groups0 - 1) / groups0);
  const unsigned groups = (unsigned)((m->p + a.cols_per_group - 1) / a.cols_per_group);
  hipLaunchKernelGGL((k_nl_slab<KIND, 2>), dim3((unsigned)((s.nslab + 3) / 4), groups), dim3(256), 0, ctx->stream, a);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

// phase B; with d_t / d_d the per-block sums A, D land there ([nblocks][n])
static int nl_rowpass(mbpls_ctx* ctx, const mbpls_mat* src, mbpls_mat* dst, int mode, const double* w, const double* mu,
                      const double* sc, const double* pv, const double* ts, const double* rowmask, double* d_t,
                      double* d_d) {
  const mbpls_mat* m = src;
  const ChunkTable& ct = m->coarse;
  NlRowArgs a{};
  a.src = src->data; a.dst = dst ? dst->data : nullptr; a.ld = m->ld; a.n = m->n;
  a.c0 = ct.d_c0; a.c1 = ct.d_c1;
  a.w = w; a.mu = mu; a.sc = sc; a.pv = pv; a.ts = ts; a.rowmask = rowmask;
  const bool acc = d_t != nullptr;
  const bool direct = ct.nchunks == m->nblocks;   // one chunk per block: the partial IS the result
  if (acc) {
    if (direct) {
      a.part_t = d_t;
      a.part_d = d_d;
    } else {
      const size_t nt = (size_t)ct.nchunks * (size_t)m->n;
      MB_TRY(ensure_ws(ctx, sizeof(double) * 2 * nt));
      a.part_t = (double*)ctx->ws;
      a.part_d = a.part_t + nt;
    }
  }
  dim3 grid((unsigned)ct.nchunks, (unsigned)((m->n + 511) / 512)), block(256);
#define MB_NR(M_)                                                                        \
  do {                                                                                   \
    if (acc) hipLaunchKernelGGL((k_nl_rows<M_, true>), grid, block, 0, ctx->stream, a);  \
    else hipLaunchKernelGGL((k_nl_rows<M_, false>), grid, block, 0, ctx->stream, a);     \
  } while (0)
  if (mode == NR_STD_STORE) MB_NR(NR_STD_STORE);
  else if (mode == NR_DEFL_STORE) MB_NR(NR_DEFL_STORE);
  else MB_NR(NR_ACC);
#undef MB_NR
  MB_HIP(hipGetLastError());
  if (acc && !direct) {
    launch_reduce_xw(ctx, m, a.part_t, 1, 1, 1, d_t, m->n, 1, 0);
    MB_HIP(hipGetLastError());
    launch_reduce_xw(ctx, m, a.part_d, 1, 1, 1, d_d, m->n, 1, 0);
    MB_HIP(hipGetLastError());
  }
  return MBPLS_OK;
}

template <int KIND>
static int nl_finish(mbpls_ctx* ctx, const mbpls_mat* m, const NlScratch& s, NlFinArgs f) {
  f.c0 = m->fine.d_c0; f.c1 = m->fine.d_c1; f.part = s.part; f.nslab = s.nslab; f.p = m->p;
  f.wrow = s.wrow; f.part_s = s.part_s;
  hipLaunchKernelGGL((k_nl_fin<KIND>), dim3((unsigned)m->fine.nchunks), dim3(256), 0, ctx->stream, f);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

int nan_long_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode, const double* d_rowmask, double* d_mean,
                               double* d_scale, double* d_count, double* d_ssq_x, const double* d_v, double* d_g,
                               double* d_t, double* d_d, double* d_ssq_g) {
  mbpls_ctx* ctx = dst->ctx;
  NlScratch s;
  MB_TRY(nl_scratch(ctx, src, &s));
  ProfScope prof(ctx, MBPLS_K_STANDARDIZE, 16.0 * (double)dst->n * (double)dst->p);
  const double* shift = nullptr;
  if (mode == MBPLS_SCALE_FIT) {
    hipLaunchKernelGGL(k_nl_first_present, dim3((unsigned)((src->p + 3) / 4)), dim3(256), 0, ctx->stream,
                       (const double*)src->data, src->ld, src->n, src->p, s.shift);
    MB_HIP(hipGetLastError());
    shift = s.shift;
  } else if (mode == MBPLS_SCALE_APPLY) {
    shift = d_mean;
  }
  MB_TRY((nl_phase_a<NL_STD>(ctx, src, s, d_v, nullptr, d_rowmask, shift)));
  NlFinArgs f{};
  f.mode = mode; f.has_v = d_v ? 1 : 0; f.shift = shift;
  f.mean = d_mean; f.scale = d_scale; f.count = d_count; f.var_out = ctx->var_out; f.g = d_g;
  MB_TRY((nl_finish<NL_STD>(ctx, src, s, f)));
  if (d_ssq_x || (d_v && d_ssq_g))
    MB_TRY(reduce_chunks(ctx, src, nullptr, s.part_s, nullptr, d_v ? d_ssq_g : nullptr, nullptr, d_ssq_x));
  const bool acc = d_v != nullptr;
  const bool scaled = mode != MBPLS_SCALE_NONE;
  // (standardize=False without a row mask and in place: nothing to store -- the accumulation alone)
  if (!scaled && !d_rowmask && src->data == dst->data)
    return acc ? nl_rowpass(ctx, src, nullptr, NR_ACC, s.wrow, nullptr, nullptr, nullptr, nullptr, nullptr, d_t, d_d)
               : MBPLS_OK;
  return nl_rowpass(ctx, src, dst, NR_STD_STORE, s.wrow, scaled ? d_mean : nullptr, scaled ? d_scale : nullptr, nullptr,
                    nullptr, d_rowmask, acc ? d_t : nullptr, acc ? d_d : nullptr);
}

int nan_long_sweep(const mbpls_mat* m, const double* d_v, double* d_g, double* d_t, double* d_d, double* d_ssq_g) {
  mbpls_ctx* ctx = m->ctx;
  NlScratch s;
  MB_TRY(nl_scratch(ctx, m, &s));
  ProfScope prof(ctx, MBPLS_K_SWEEP, 8.0 * (double)m->n * (double)m->p);
  MB_TRY((nl_phase_a<NL_SWEEP>(ctx, m, s, d_v, nullptr, nullptr, nullptr)));
  NlFinArgs f{};
  f.g = d_g;
  MB_TRY((nl_finish<NL_SWEEP>(ctx, m, s, f)));
  if (d_ssq_g) MB_TRY(reduce_chunks(ctx, m, nullptr, s.part_s, nullptr, d_ssq_g, nullptr, nullptr));
  return nl_rowpass(ctx, m, nullptr, NR_ACC, s.wrow, nullptr, nullptr, nullptr, nullptr, nullptr, d_t, d_d);
}

int nan_long_deflate_sweep(mbpls_mat* m, const double* d_ts, int deflate, double* d_p, double* d_ssq_p,
                           const double* d_v, double* d_g, double* d_t, double* d_d, double* d_ssq_g) {
  mbpls_ctx* ctx = m->ctx;
  NlScratch s;
  MB_TRY(nl_scratch(ctx, m, &s));
  ProfScope prof(ctx, d_v ? MBPLS_K_DEFLATE_SWEEP : MBPLS_K_DEFLATE,
                 8.0 * (double)m->n * (double)m->p * (deflate ? 2.0 : 1.0));
  NlFinArgs f{};
  f.deflate = deflate ? 1 : 0; f.g = d_g;
  f.pvec = d_p ? d_p : s.ptmp;
  if (d_v) {
    MB_TRY((nl_phase_a<NL_DEFL_SWEEP>(ctx, m, s, d_v, d_ts, nullptr, nullptr)));
    MB_TRY((nl_finish<NL_DEFL_SWEEP>(ctx, m, s, f)));
  } else {
    MB_TRY((nl_phase_a<NL_DEFL>(ctx, m, s, nullptr, d_ts, nullptr, nullptr)));
    MB_TRY((nl_finish<NL_DEFL>(ctx, m, s, f)));
  }
  if (d_ssq_p || (d_v && d_ssq_g))
    MB_TRY(reduce_chunks(ctx, m, nullptr, s.part_s, nullptr, d_v ? d_ssq_g : nullptr, d_ssq_p, nullptr));
  const bool acc = d_v != nullptr;
  if (deflate)
    return nl_rowpass(ctx, m, m, NR_DEFL_STORE, s.wrow, nullptr, nullptr, f.pvec, d_ts, nullptr, acc ? d_t : nullptr,
                      acc ? d_d : nullptr);
  if (acc) return nl_rowpass(ctx, m, nullptr, NR_ACC, s.wrow, nullptr, nullptr, nullptr, nullptr, nullptr, d_t, d_d);
  return MBPLS_OK;
}

}  // namespace mbpls
