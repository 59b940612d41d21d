// Fused column-pass kernels: the HBM-bound heart of fit().
//
// One workgroup owns one chunk of contiguous feature columns.  A column (n doubles, contiguous
// in the feature-major layout) is held in registers by the whole workgroup: thread t owns the
// row pairs (k*T + t), k < R/2, loaded as 16-byte double2 -> every wave-instruction covers 1 KiB
// of contiguous HBM.  Per column the workgroup does (depending on the template flags)
//   STD   : mean / corrected two-pass variance / scale, z = (x - mean)/scale, store z      (a1)
//   DEFL  : p_j = inv_tt * <x, ts>,  x <- x - p_j ts, store x                                (a5+a7)
//   SWEEP : g_j = alpha * <x, v>,    acc += g_j x                                            (a2+a3)
// with wave-butterfly + LDS cross-wave reductions (deterministic, identical every call), and
// at the end writes its n-vector partial `acc`; k_reduce_chunks then sums the chunks of every
// block in fixed order.  D columns are kept in flight per workgroup (register double-buffering)
// so HBM latency is covered by the loads of the following columns and by co-resident workgroups.
//
// Bytes per column: SWEEP 8n read; DEFL(+SWEEP) 8n read + 8n write; STD(+SWEEP) 8n read + 8n write.
#include <algorithm>

#include "common.h"

namespace mbpls {

struct PassArgs {
  const double* src;
  double* dst;
  int64_t ld;
  int64_t n;
  const int64_t* c0;
  const int64_t* c1;
  // sweep
  const double* v;
  double alpha;
  double* g;
  // deflate
  const double* ts;
  double inv_tt;
  int deflate;
  double* pvec;
  // standardise
  int mode;
  double* mean;
  double* scale;
  // partial outputs
  double* part_t;  // [nchunks][n]
  double* part_s;  // [nchunks][4]: ssq_g, ssq_p, ssq_x, -
  int nt;          // bit 0: non-temporal loads of X, bit 1: non-temporal stores of X
};

template <int R>
__device__ __forceinline__ void load_col(double (&x)[R], const double* __restrict__ col, int64_t n, int64_t ld,
                                         int tid, int T, bool nt = false) {
  typedef double v2d __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < R / 2; ++k) {
    const int64_t r0 = 2 * ((int64_t)k * T + tid);
    double2 t = make_double2(0.0, 0.0);
    if (r0 < ld) {
      if (nt) {   // workgroup-uniform: streamed once, do not keep in L2/MALL
        const v2d q = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(col + r0));
        t = make_double2(q.x, q.y);
      } else {
        t = *reinterpret_cast<const double2*>(col + r0);
      }
    }
    x[2 * k] = (r0 < n) ? t.x : 0.0;
    x[2 * k + 1] = (r0 + 1 < n) ? t.y : 0.0;
  }
}

// plain n-vector (u, ts): may be unpadded / only 8-byte aligned -> scalar loads, done once per workgroup
template <int R>
__device__ __forceinline__ void load_vec(double (&x)[R], const double* __restrict__ vec, int64_t n, int tid, int T) {
#pragma unroll
  for (int k = 0; k < R / 2; ++k) {
    const int64_t r0 = 2 * ((int64_t)k * T + tid);
    x[2 * k] = (r0 < n) ? vec[r0] : 0.0;
    x[2 * k + 1] = (r0 + 1 < n) ? vec[r0 + 1] : 0.0;
  }
}

template <int R>
__device__ __forceinline__ void store_col(const double (&x)[R], double* __restrict__ col, int64_t ld, int tid,
                                          int T, bool nt = false) {
  typedef double v2d __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < R / 2; ++k) {
    const int64_t r0 = 2 * ((int64_t)k * T + tid);
    if (r0 < ld) {
      if (nt) {
        v2d q;
        q.x = x[2 * k];
        q.y = x[2 * k + 1];
        __builtin_nontemporal_store(q, reinterpret_cast<v2d*>(col + r0));
      } else {
        *reinterpret_cast<double2*>(col + r0) = make_double2(x[2 * k], x[2 * k + 1]);
      }
    }
  }
}

template <int R, int TMAX, int D, bool STD, bool DEFL, bool SWEEP>
__global__ __launch_bounds__(TMAX) void k_colpass(PassArgs a) {
  __shared__ double red[2 * 64];
  int buf = 0;
  const int tid = threadIdx.x, T = blockDim.x;
  const int64_t n = a.n, ld = a.ld;
  const int chunk = blockIdx.x;
  const int64_t j0 = a.c0[chunk], j1 = a.c1[chunk];
  const bool ntl = (a.nt & 1) != 0, nts = (a.nt & 2) != 0;

  double v[SWEEP ? R : 1], ts[DEFL ? R : 1], acc[SWEEP ? R : 1];
  double tsv = 0.0;
  if constexpr (SWEEP) {
    load_vec<R>(v, a.v, n, tid, T);
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = 0.0;
  }
  if constexpr (DEFL) load_vec<R>(ts, a.ts, n, tid, T);
  if constexpr (DEFL && SWEEP) {
    double s[1] = {0.0};
#pragma unroll
    for (int k = 0; k < R; ++k) s[0] = fma(ts[k], v[k], s[0]);
    block_sum<1>(s, red, buf);
    tsv = s[0];
  }
  double ssq_g = 0.0, ssq_p = 0.0, ssq_x = 0.0;
  const double dn = (double)n;

  double x[D][R];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (j0 + d < j1) load_col<R>(x[d], a.src + (j0 + d) * ld, n, ld, tid, T, ntl);

  for (int64_t j = j0; j < j1; j += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int64_t jj = j + d;
      if (jj >= j1) break;
      double(&xc)[R] = x[d];

      if constexpr (STD) {
        if (a.mode != MBPLS_SCALE_NONE) {
          double mean, scale;
          if (a.mode == MBPLS_SCALE_FIT) {
            // sklearn _incremental_mean_and_var: mean, then corrected two-pass variance
            double s[1] = {0.0};
#pragma unroll
            for (int k = 0; k < R; ++k) s[0] += xc[k];
            block_sum<1>(s, red, buf);
            mean = s[0] / dn;
            double cq[2] = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < R; ++k) {
              const int64_t r = 2 * ((int64_t)(k >> 1) * T + tid) + (k & 1);
              const double dlt = (r < n) ? xc[k] - mean : 0.0;
              xc[k] = dlt;
              cq[0] += dlt;
              cq[1] = fma(dlt, dlt, cq[1]);
            }
            block_sum<2>(cq, red, buf);
            const double var = (cq[1] - cq[0] * cq[0] / dn) / dn;
            const double eps = 2.220446049250313e-16;
            const double nme = dn * mean * eps;
            const double bound = dn * eps * var + nme * nme;
            scale = (var <= bound) ? 1.0 : sqrt(var);
            if (tid == 0) {
              a.mean[jj] = mean;
              a.scale[jj] = scale;
            }
          } else {
            mean = a.mean[jj];
            scale = a.scale[jj];
#pragma unroll
            for (int k = 0; k < R; ++k) {
              const int64_t r = 2 * ((int64_t)(k >> 1) * T + tid) + (k & 1);
              xc[k] = (r < n) ? xc[k] - mean : 0.0;
            }
          }
#pragma unroll
          for (int k = 0; k < R; ++k) xc[k] = xc[k] / scale;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) ssq_x = fma(xc[k], xc[k], ssq_x);
        store_col<R>(xc, a.dst + jj * ld, ld, tid, T, nts);
      }

      double gj = 0.0;
      if constexpr (DEFL) {
        double ab[2] = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < R; ++k) {
          ab[0] = fma(xc[k], ts[k], ab[0]);
          if constexpr (SWEEP) ab[1] = fma(xc[k], v[k], ab[1]);
        }
        if constexpr (SWEEP) block_sum<2>(ab, red, buf);
        else {
          double a1[1] = {ab[0]};
          block_sum<1>(a1, red, buf);
          ab[0] = a1[0];
        }
        const double pj = ab[0] * a.inv_tt;
        ssq_p = fma(pj, pj, ssq_p);
        if (tid == 0 && a.pvec) a.pvec[jj] = pj;
        if (a.deflate) {
#pragma unroll
          for (int k = 0; k < R; ++k) xc[k] = fma(-pj, ts[k], xc[k]);
          store_col<R>(xc, a.dst + jj * ld, ld, tid, T, nts);
          if constexpr (SWEEP) gj = a.alpha * fma(-pj, tsv, ab[1]);  // <x - p ts, v> = <x,v> - p <ts,v>
        } else {
          if constexpr (SWEEP) gj = a.alpha * ab[1];
        }
      } else if constexpr (SWEEP) {
        double b[1] = {0.0};
#pragma unroll
        for (int k = 0; k < R; ++k) b[0] = fma(xc[k], v[k], b[0]);
        block_sum<1>(b, red, buf);
        gj = a.alpha * b[0];
      }
      if constexpr (SWEEP) {
        ssq_g = fma(gj, gj, ssq_g);
        if (tid == 0 && a.g) a.g[jj] = gj;
#pragma unroll
        for (int k = 0; k < R; ++k) acc[k] = fma(gj, xc[k], acc[k]);
      }
      // refill this register slot with the column D ahead (in flight while the next ones are processed)
      if (jj + D < j1) load_col<R>(xc, a.src + (jj + D) * ld, n, ld, tid, T, ntl);
    }
  }

  if constexpr (SWEEP) {
    double* out = a.part_t + (int64_t)chunk * n;
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      const int64_t r0 = 2 * ((int64_t)k * T + tid);
      if (r0 + 1 < n) *reinterpret_cast<double2*>(out + r0) = make_double2(acc[2 * k], acc[2 * k + 1]);
      else if (r0 < n) out[r0] = acc[2 * k];
    }
  }
  if constexpr (STD) {
    double s[1] = {ssq_x};
    block_sum<1>(s, red, buf);
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

// Second stage: fixed-order sum of the chunk partials of each block.
//   grid (ceil(n/64), nblocks), 256 threads: wave w takes chunks cs+w, cs+w+4, ...; LDS combine in wave order.
//   The workgroup with blockIdx.x == 0 also reduces the per-chunk scalars of its block.
__global__ __launch_bounds__(256) void k_reduce_chunks(const double* __restrict__ part_t,
                                                       const double* __restrict__ part_s,
                                                       const int* __restrict__ blk_start, int64_t n,
                                                       double* __restrict__ out_t, double* __restrict__ out_ssq_g,
                                                       double* __restrict__ out_ssq_p, double* __restrict__ out_ssq_x) {
  __shared__ double comb[4][64];
  const int b = blockIdx.y;
  const int cs = blk_start[b], ce = blk_start[b + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (out_t) {
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    double s = 0.0;
    if (i < n) {
      int c = cs + w;
      // 4 independent accumulators keep several loads in flight; combined in fixed order
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (; c + 12 < ce; c += 16) {
        s0 += part_t[(int64_t)c * n + i];
        s1 += part_t[(int64_t)(c + 4) * n + i];
        s2 += part_t[(int64_t)(c + 8) * n + i];
        s3 += part_t[(int64_t)(c + 12) * n + i];
      }
      for (; c < ce; c += 4) s0 += part_t[(int64_t)c * n + i];
      s = (s0 + s1) + (s2 + s3);
    }
    comb[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < n) out_t[(int64_t)b * n + i] = (comb[0][lane] + comb[1][lane]) + (comb[2][lane] + comb[3][lane]);
  }
  if (blockIdx.x == 0 && w < 3) {
    double* o = (w == 0) ? out_ssq_g : (w == 1 ? out_ssq_p : out_ssq_x);
    if (o) {  // wave-uniform
      double s = 0.0;
      for (int c = cs + lane; c < ce; c += 64) s += part_s[(int64_t)c * 4 + w];
      s = wave_sum(s);
      if (lane == 0) o[b] = s;
    }
  }
}

// ------------------------------------------------------------------------------- dispatch
struct Cfg {
  int R, TMAX;
};

// Choose rows-per-thread R and the workgroup size T = roundup(ceil(n/R), 64) <= TMAX.
static bool pick_cfg(const mbpls_ctx* ctx, int64_t n, Cfg* cfg, int* T, bool std_pass = false) {
  // preference order: smallest register footprint that still covers n with <= TMAX threads
  static const Cfg table[] = {{2, 256}, {4, 512}, {8, 512}, {8, 1024}, {16, 512}, {4, 1024}, {16, 1024}};
  auto fits = [&](const Cfg& c, int* t) {
    int64_t need = (n + c.R - 1) / c.R;
    need = (need + 63) / 64 * 64;
    if (need > c.TMAX) return false;
    *t = (int)need;
    return true;
  };
  // the standardise pass has three reductions + divisions per column: fewer, fatter threads win
  // (measured at n=4096: R=16/T=256 5.46 TB/s vs R=8/T=512 4.77 TB/s)
  if (std_pass && ctx->rows_per_thread <= 0 && n > 2048) {
    const Cfg c16 = {16, 512};
    if (fits(c16, T)) {
      *cfg = c16;
      return true;
    }
  }
  if (ctx->rows_per_thread > 0) {
    for (const Cfg& c : table)
      if (c.R == ctx->rows_per_thread && fits(c, T)) {
        *cfg = c;
        return true;
      }
  }
  for (const Cfg& c : table)
    if (c.TMAX * c.R <= 8192 && fits(c, T)) {
      *cfg = c;
      return true;
    }
  return false;
}

template <bool STD, bool DEFL, bool SWEEP>
static int launch_pass(mbpls_ctx* ctx, const ChunkTable& ct, const PassArgs& a, int64_t p) {
  // algorithmic bytes: read 8np, plus 8np write when the pass stores X
  const bool writes = STD || (DEFL && a.deflate);
  const int kind = STD ? MBPLS_K_STANDARDIZE : (DEFL ? (SWEEP ? MBPLS_K_DEFLATE_SWEEP : MBPLS_K_DEFLATE) : MBPLS_K_SWEEP);
  ProfScope prof(ctx, kind, 8.0 * (double)a.n * (double)p * (writes ? 2.0 : 1.0));
  Cfg cfg;
  int T = 0;
  if (!pick_cfg(ctx, a.n, &cfg, &T, STD)) {
    set_error("fused column pass supports n <= 8192 (got n=%lld)", (long long)a.n);
    return MBPLS_ERR_UNSUPPORTED;
  }
  PassArgs an = a;
  an.nt = ctx->nontemporal;
  dim3 grid((unsigned)ct.nchunks), block((unsigned)T);
  const int D = ctx->prefetch >= 1 && ctx->prefetch <= 3 ? ctx->prefetch : 2;
#define MB_LAUNCH(R_, TM_, D_)                                                                          \
  hipLaunchKernelGGL((k_colpass<R_, TM_, D_, STD, DEFL, SWEEP>), grid, block, 0, ctx->stream, an)
#define MB_LAUNCH_D(R_, TM_)            \
  do {                                  \
    if (D == 1) MB_LAUNCH(R_, TM_, 1);  \
    else if (D == 3) MB_LAUNCH(R_, TM_, 3); \
    else MB_LAUNCH(R_, TM_, 2);         \
  } while (0)
  if (cfg.R == 2) MB_LAUNCH_D(2, 256);
  else if (cfg.R == 4 && cfg.TMAX == 512) MB_LAUNCH_D(4, 512);
  else if (cfg.R == 4) MB_LAUNCH_D(4, 1024);
  else if (cfg.R == 8 && cfg.TMAX == 512) MB_LAUNCH_D(8, 512);
  else if (cfg.R == 8) MB_LAUNCH_D(8, 1024);
  else if (cfg.R == 16 && cfg.TMAX == 512) MB_LAUNCH_D(16, 512);
  else {
    set_error("no kernel for R=%d TMAX=%d", cfg.R, cfg.TMAX);
    return MBPLS_ERR_UNSUPPORTED;
  }
#undef MB_LAUNCH_D
#undef MB_LAUNCH
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

static int reduce_chunks(mbpls_ctx* ctx, const mbpls_mat* m, const double* part_t, const double* part_s,
                         double* d_t, double* d_ssq_g, double* d_ssq_p, double* d_ssq_x) {
  dim3 grid(d_t ? (unsigned)((m->n + 63) / 64) : 1u, (unsigned)m->nblocks);
  hipLaunchKernelGGL(k_reduce_chunks, grid, dim3(256), 0, ctx->stream, part_t, part_s,
                     (const int*)m->fine.d_blk_start, m->n, d_t, d_ssq_g, d_ssq_p, d_ssq_x);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

static int partial_buffers(mbpls_ctx* ctx, const mbpls_mat* m, double** part_t, double** part_s) {
  const size_t nt = (size_t)m->fine.nchunks * (size_t)m->n, ns = (size_t)m->fine.nchunks * 4;
  MB_TRY(ensure_ws(ctx, sizeof(double) * (nt + ns)));
  *part_t = (double*)ctx->ws;
  *part_s = *part_t + nt;
  return MBPLS_OK;
}

}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_sweep(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t,
                    double* d_ssq_g) {
  MB_ARG(m && d_v, "NULL argument");
  mbpls_ctx* ctx = m->ctx;
  PassArgs a{};
  MB_TRY(partial_buffers(ctx, m, &a.part_t, &a.part_s));
  a.src = m->data; a.dst = nullptr; a.ld = m->ld; a.n = m->n;
  a.c0 = m->fine.d_c0; a.c1 = m->fine.d_c1;
  a.v = d_v; a.alpha = alpha; a.g = d_g;
  MB_TRY((launch_pass<false, false, true>(ctx, m->fine, a, m->p)));
  return reduce_chunks(ctx, m, a.part_t, a.part_s, d_t, d_ssq_g, nullptr, nullptr);
}

int mbpls_hip_deflate_sweep(mbpls_mat* m, const double* d_ts, double inv_tt, int deflate, double* d_p,
                            double* d_ssq_p, const double* d_v, double alpha, double* d_g, double* d_t,
                            double* d_ssq_g) {
  MB_ARG(m && d_ts, "NULL argument");
  mbpls_ctx* ctx = m->ctx;
  PassArgs a{};
  MB_TRY(partial_buffers(ctx, m, &a.part_t, &a.part_s));
  a.src = m->data; a.dst = m->data; a.ld = m->ld; a.n = m->n;
  a.c0 = m->fine.d_c0; a.c1 = m->fine.d_c1;
  a.ts = d_ts; a.inv_tt = inv_tt; a.deflate = deflate ? 1 : 0; a.pvec = d_p;
  a.v = d_v; a.alpha = alpha; a.g = d_g;
  if (d_v) {
    MB_TRY((launch_pass<false, true, true>(ctx, m->fine, a, m->p)));
    return reduce_chunks(ctx, m, a.part_t, a.part_s, d_t, d_ssq_g, d_ssq_p, nullptr);
  }
  MB_TRY((launch_pass<false, true, false>(ctx, m->fine, a, m->p)));
  if (d_ssq_p) return reduce_chunks(ctx, m, a.part_t, a.part_s, nullptr, nullptr, d_ssq_p, nullptr);
  return MBPLS_OK;
}

int mbpls_hip_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode, double* d_mean, double* d_scale,
                                double* d_ssq_x, const double* d_v, double alpha, double* d_g, double* d_t,
                                double* d_ssq_g) {
  MB_ARG(src && dst, "NULL argument");
  MB_ARG(src->n == dst->n && src->p == dst->p && src->ld == dst->ld, "src/dst shape mismatch");
  MB_ARG(mode == MBPLS_SCALE_FIT || mode == MBPLS_SCALE_APPLY || mode == MBPLS_SCALE_NONE, "bad mode %d", mode);
  MB_ARG(mode == MBPLS_SCALE_NONE || (d_mean && d_scale), "mean/scale buffers required");
  mbpls_ctx* ctx = dst->ctx;
  PassArgs a{};
  MB_TRY(partial_buffers(ctx, dst, &a.part_t, &a.part_s));
  a.src = src->data; a.dst = dst->data; a.ld = dst->ld; a.n = dst->n;
  a.c0 = dst->fine.d_c0; a.c1 = dst->fine.d_c1;
  a.mode = mode; a.mean = d_mean; a.scale = d_scale;
  a.v = d_v; a.alpha = alpha; a.g = d_g;
  if (d_v) {
    MB_TRY((launch_pass<true, false, true>(ctx, dst->fine, a, dst->p)));
    return reduce_chunks(ctx, dst, a.part_t, a.part_s, d_t, d_ssq_g, nullptr, d_ssq_x);
  }
  MB_TRY((launch_pass<true, false, false>(ctx, dst->fine, a, dst->p)));
  if (d_ssq_x) return reduce_chunks(ctx, dst, a.part_t, a.part_s, nullptr, nullptr, nullptr, d_ssq_x);
  return MBPLS_OK;
}

}  // extern "C"
