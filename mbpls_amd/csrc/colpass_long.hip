// Column passes for LONG columns: n beyond what a workgroup can hold in registers (colpass.hip covers
// n <= 65536, its register-resident configurations spill above 8192).  Tall data (n >= p: many samples,
// few features -- the n >= p branches of UNIPALS / KERNEL, mbpls.py:388-479, :580-692) lands here.
//
// One workgroup per chunk of feature columns as before, but a column is streamed from HBM / L2 in row
// strides, once per reduction it needs:
//   STD   : sum -> mean ; corrected two-pass variance -> scale ; write z (+ <z, v> for a fused first sweep)
//   DEFL  : <x, ts> (+ <x, v>) -> p_j ; write x - p_j ts
// (a column of n = 10^6 samples is 8 MB: the re-reads come from L2 / MALL, not HBM).  The block scores
// t_b = X_b g_b of a sweep are a second pass (k_xw, rows are the parallel dimension there), i.e. a sweep costs
// two reads of X instead of one -- the price of not holding the column.  Reductions are fixed-order.
//
// Bytes (algorithmic, as the register-resident passes): SWEEP 8np, STD / DEFL 16np.
#include <algorithm>

#include "colpass_util.h"

namespace mbpls {

struct LongArgs {
  double* var_out;
  const double* src;
  double* dst;
  int64_t ld, n;
  const int64_t* c0;
  const int64_t* c1;
  const double* v;
  double alpha;
  double* g;
  const double* ts;
  double inv_tt;
  int deflate;
  double* pvec;
  int mode;
  double* mean;
  double* scale;
  const double* rowmask;
  double n_eff;
  double* part_s;   // [nchunks][4]: ssq_g, ssq_p, ssq_x, -
};

// rows r, r+1 of this thread's stride; the pair is one 16-byte load (ld is even, columns 16-byte aligned)
#define MB_ROWS(r) for (int64_t r = 2 * (int64_t)threadIdx.x; r < n; r += 2 * (int64_t)blockDim.x)

__device__ __forceinline__ double2 ld2(const double* __restrict__ p, int64_t r, int64_t n) {
  if (r + 1 < n) return *reinterpret_cast<const double2*>(p + r);
  return make_double2(p[r], 0.0);
}
// n-vectors (u, ts, rowmask) may be unpadded and only 8-byte aligned
__device__ __forceinline__ double2 ldv(const double* __restrict__ p, int64_t r, int64_t n) {
  return make_double2(p[r], (r + 1 < n) ? p[r + 1] : 0.0);
}
__device__ __forceinline__ void st2(double* __restrict__ p, int64_t r, int64_t n, double2 q) {
  if (r + 1 < n) *reinterpret_cast<double2*>(p + r) = q;
  else p[r] = q.x;
}

template <bool STD, bool DEFL, bool SWEEP>
__global__ __launch_bounds__(256) void k_colpass_long(LongArgs a) {
  __shared__ double red[2 * 64];
  int buf = 0;
  const int tid = threadIdx.x;
  const int64_t n = a.n, ld = a.ld;
  const int chunk = blockIdx.x;
  const int64_t j0 = a.c0[chunk], j1 = a.c1[chunk];
  const bool masked = STD && a.rowmask != nullptr;
  const double dn = masked ? a.n_eff : (double)n;
  double ssq_g = 0.0, ssq_p = 0.0, ssq_x = 0.0;
  double tsv = 0.0;
  if constexpr (DEFL && SWEEP) {
    double s[1] = {0.0};
    MB_ROWS(r) {
      const double2 t2 = ldv(a.ts, r, n), v2 = ldv(a.v, r, n);
      s[0] = fma(t2.x, v2.x, s[0]);
      s[0] = fma(t2.y, v2.y, s[0]);
    }
    block_sum<1>(s, red, buf);
    tsv = s[0];
  }

  for (int64_t j = j0; j < j1; ++j) {
    const double* col = a.src + j * ld;
    double* out = a.dst ? a.dst + j * ld : nullptr;
    double gj = 0.0;

    if constexpr (STD) {
      double mean = 0.0, scale = 1.0;
      if (a.mode == MBPLS_SCALE_FIT) {
        double s[1] = {0.0};
        MB_ROWS(r) {
          double2 x = ld2(col, r, n);
          if (masked) {
            const double2 m = ldv(a.rowmask, r, n);
            x.x = (m.x != 0.0) ? x.x : 0.0;
            x.y = (m.y != 0.0) ? x.y : 0.0;
          }
          s[0] += x.x;
          s[0] += x.y;
        }
        block_sum<1>(s, red, buf);
        mean = s[0] / dn;
        double cq[2] = {0.0, 0.0};
        MB_ROWS(r) {
          const double2 x = ld2(col, r, n);
          bool k0 = true, k1 = r + 1 < n;
          if (masked) {
            const double2 m = ldv(a.rowmask, r, n);
            k0 = m.x != 0.0;
            k1 = k1 && m.y != 0.0;
          }
          const double d0 = k0 ? x.x - mean : 0.0, d1 = k1 ? x.y - mean : 0.0;
          cq[0] += d0;
          cq[1] = fma(d0, d0, cq[1]);
          cq[0] += d1;
          cq[1] = fma(d1, d1, cq[1]);
        }
        block_sum<2>(cq, red, buf);
        const double var = (cq[1] - cq[0] * cq[0] / dn) / dn;
        const double eps = 2.220446049250313e-16;
        const double nme = dn * mean * eps;
        const double bound = dn * eps * var + nme * nme;
        scale = (var <= bound) ? 1.0 : sqrt(var);
        if (tid == 0) {
          a.mean[j] = mean;
          a.scale[j] = scale;
          if (a.var_out) a.var_out[j] = (var <= bound) ? 0.0 : var;
        }
      } else if (a.mode == MBPLS_SCALE_APPLY) {
        mean = a.mean[j];
        scale = a.scale[j];
      }
      const bool scaled = a.mode != MBPLS_SCALE_NONE;
      double sv[2] = {0.0, 0.0};   // ssq of the written column, <z, v>
      MB_ROWS(r) {
        double2 x = ld2(col, r, n);
        bool k0 = true, k1 = r + 1 < n;
        if (masked) {
          const double2 m = ldv(a.rowmask, r, n);
          k0 = m.x != 0.0;
          k1 = k1 && m.y != 0.0;
        }
        if (scaled) {
          x.x = (x.x - mean) / scale;
          x.y = (x.y - mean) / scale;
        }
        x.x = k0 ? x.x : 0.0;
        x.y = k1 ? x.y : 0.0;
        sv[0] = fma(x.x, x.x, sv[0]);
        sv[0] = fma(x.y, x.y, sv[0]);
        if constexpr (SWEEP) {
          const double2 v2 = ldv(a.v, r, n);
          sv[1] = fma(x.x, v2.x, sv[1]);
          sv[1] = fma(x.y, v2.y, sv[1]);
        }
        st2(out, r, n, x);
      }
      block_sum<2>(sv, red, buf);
      if (tid == 0) ssq_x += sv[0];
      gj = a.alpha * sv[1];
    }

    if constexpr (DEFL) {
      double ab[2] = {0.0, 0.0};
      MB_ROWS(r) {
        const double2 x = ld2(col, r, n), t2 = ldv(a.ts, r, n);
        ab[0] = fma(x.x, t2.x, ab[0]);
        ab[0] = fma(x.y, t2.y, ab[0]);
        if constexpr (SWEEP) {
          const double2 v2 = ldv(a.v, r, n);
          ab[1] = fma(x.x, v2.x, ab[1]);
          ab[1] = fma(x.y, v2.y, ab[1]);
        }
      }
      block_sum<2>(ab, red, buf);
      const double pj = ab[0] * a.inv_tt;
      ssq_p = fma(pj, pj, ssq_p);
      if (tid == 0 && a.pvec) a.pvec[j] = pj;
      if (a.deflate) {
        MB_ROWS(r) {
          double2 x = ld2(col, r, n);
          const double2 t2 = ldv(a.ts, r, n);
          x.x = fma(-pj, t2.x, x.x);
          x.y = fma(-pj, t2.y, x.y);
          st2(out, r, n, x);
        }
        gj = a.alpha * fma(-pj, tsv, ab[1]);
      } else {
        gj = a.alpha * ab[1];
      }
    }

    if constexpr (SWEEP) {
      ssq_g = fma(gj, gj, ssq_g);
      if (tid == 0) a.g[j] = gj;
    }
  }
  if (tid == 0) {
    double* ps = a.part_s + (int64_t)chunk * 4;
    ps[0] = ssq_g;
    ps[1] = ssq_p;
    ps[2] = ssq_x;
    ps[3] = 0.0;
  }
}
#undef MB_ROWS

bool long_columns(const mbpls_ctx* ctx, int64_t n) { return n > ctx->long_columns_from; }

static int long_part_s(mbpls_ctx* ctx, const mbpls_mat* m, double** part_s) {
  MB_TRY(ensure_ws(ctx, sizeof(double) * (size_t)m->fine.nchunks * 4));
  *part_s = (double*)ctx->ws;
  return MBPLS_OK;
}

// scratch weights when the caller does not want them back (UNIPALS products pass d_g = NULL)
static int long_g(mbpls_ctx* ctx, const mbpls_mat* m, double* d_g, double** g) {
  if (d_g) {
    *g = d_g;
    return MBPLS_OK;
  }
  MB_TRY(ensure_stage(ctx, sizeof(double) * (size_t)m->p));
  *g = (double*)ctx->stage;
  return MBPLS_OK;
}

// t~_b = X_b g_b for every block: the row-parallel product kernel (d_t is [nblocks][n])
static int long_scores(const mbpls_mat* m, const double* g, double* d_t) {
  return mbpls_hip_x_w(m, g, m->p, 1, 1, nullptr, nullptr, d_t, m->n);
}

int long_sweep(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g) {
  double* g = nullptr;
  MB_TRY(long_g(m->ctx, m, d_g, &g));
  MB_TRY(mbpls_hip_xt_v(m, d_v, m->n, 1, alpha, g, m->p, d_ssq_g));   // matrix-core path, any n
  return d_t ? long_scores(m, g, d_t) : MBPLS_OK;
}

int long_deflate_sweep(mbpls_mat* m, const double* d_ts, double inv_tt, int deflate, double* d_p, double* d_ssq_p,
                       const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g) {
  mbpls_ctx* ctx = m->ctx;
  LongArgs a{};
  MB_TRY(long_part_s(ctx, m, &a.part_s));
  double* g = nullptr;
  if (d_v) MB_TRY(long_g(ctx, m, d_g, &g));
  a.src = m->data; a.dst = m->data; a.ld = m->ld; a.n = m->n;
  a.c0 = m->fine.d_c0; a.c1 = m->fine.d_c1;
  a.ts = d_ts; a.inv_tt = inv_tt; a.deflate = deflate ? 1 : 0; a.pvec = d_p;
  a.v = d_v; a.alpha = alpha; a.g = g;
  dim3 grid((unsigned)m->fine.nchunks), block(256);
  {
    ProfScope prof(ctx, d_v ? MBPLS_K_DEFLATE_SWEEP : MBPLS_K_DEFLATE,
                   8.0 * (double)m->n * (double)m->p * (deflate ? 2.0 : 1.0));
    if (d_v) hipLaunchKernelGGL((k_colpass_long<false, true, true>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_colpass_long<false, true, false>), grid, block, 0, ctx->stream, a);
  }
  MB_HIP(hipGetLastError());
  if (d_ssq_p || (d_v && d_ssq_g))
    MB_TRY(reduce_chunks(ctx, m, nullptr, a.part_s, nullptr, d_v ? d_ssq_g : nullptr, d_ssq_p, nullptr));
  return (d_v && d_t) ? long_scores(m, g, d_t) : MBPLS_OK;
}

int long_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode, const double* d_rowmask, double n_eff,
                           double* d_mean, double* d_scale, double* d_ssq_x, const double* d_v, double alpha,
                           double* d_g, double* d_t, double* d_ssq_g) {
  mbpls_ctx* ctx = dst->ctx;
  LongArgs a{};
  MB_TRY(long_part_s(ctx, dst, &a.part_s));
  double* g = nullptr;
  if (d_v) MB_TRY(long_g(ctx, dst, d_g, &g));
  a.src = src->data; a.dst = dst->data; a.ld = dst->ld; a.n = dst->n;
  a.c0 = dst->fine.d_c0; a.c1 = dst->fine.d_c1;
  a.mode = mode; a.mean = d_mean; a.scale = d_scale; a.rowmask = d_rowmask; a.n_eff = n_eff;
  a.var_out = ctx->var_out;
  a.v = d_v; a.alpha = alpha; a.g = g;
  dim3 grid((unsigned)dst->fine.nchunks), block(256);
  {
    ProfScope prof(ctx, MBPLS_K_STANDARDIZE, 16.0 * (double)dst->n * (double)dst->p);
    if (d_v) hipLaunchKernelGGL((k_colpass_long<true, false, true>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_colpass_long<true, false, false>), grid, block, 0, ctx->stream, a);
  }
  MB_HIP(hipGetLastError());
  if (d_ssq_x || (d_v && d_ssq_g))
    MB_TRY(reduce_chunks(ctx, dst, nullptr, a.part_s, nullptr, d_v ? d_ssq_g : nullptr, nullptr, d_ssq_x));
  return (d_v && d_t) ? long_scores(dst, g, d_t) : MBPLS_OK;
}

}  // namespace mbpls
