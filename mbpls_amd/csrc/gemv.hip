// Multi-right-hand-side products with the feature-major X:
//   xt_v : G = alpha * X^T V   (p x c from n x c)   -- column held by the workgroup, c dots per column
//   x_w  : OUT = X W           (n x c from p x c)   -- row slabs x column chunks, register accumulators
// Both are HBM-bound (<= c/4 flop per byte); right-hand sides are processed in groups so X is
// read ceil(c/CB) times.
#include <algorithm>

#include "common.h"

namespace mbpls {

// ------------------------------------------------------------------------------------ xt_v
template <int R, int TMAX, int CB>
__global__ __launch_bounds__(TMAX) void k_xtv(const double* __restrict__ X, int64_t ld, int64_t n,
                                              const int64_t* __restrict__ c0, const int64_t* __restrict__ c1,
                                              const double* __restrict__ V, int64_t ldv, int cb, double alpha,
                                              double* __restrict__ G, int64_t ldg, double* __restrict__ part_s) {
  __shared__ double red[2 * 64];
  int buf = 0;
  const int tid = threadIdx.x, T = blockDim.x;
  const int chunk = blockIdx.x;
  const int64_t j0 = c0[chunk], j1 = c1[chunk];
  double v[CB][R];
#pragma unroll
  for (int q = 0; q < CB; ++q)
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      const int64_t r0 = 2 * ((int64_t)k * T + tid);
      v[q][2 * k] = (q < cb && r0 < n) ? V[q * ldv + r0] : 0.0;
      v[q][2 * k + 1] = (q < cb && r0 + 1 < n) ? V[q * ldv + r0 + 1] : 0.0;
    }
  double ssq[CB];
#pragma unroll
  for (int q = 0; q < CB; ++q) ssq[q] = 0.0;

  auto load = [&](double(&x)[R], int64_t j) {
    const double* col = X + j * ld;
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      const int64_t r0 = 2 * ((int64_t)k * T + tid);
      double2 t = make_double2(0.0, 0.0);
      if (r0 < ld) t = *reinterpret_cast<const double2*>(col + r0);
      x[2 * k] = (r0 < n) ? t.x : 0.0;
      x[2 * k + 1] = (r0 + 1 < n) ? t.y : 0.0;
    }
  };
  double x[2][R];
  if (j0 < j1) load(x[0], j0);
  if (j0 + 1 < j1) load(x[1], j0 + 1);
  for (int64_t j = j0; j < j1; j += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int64_t jj = j + d;
      if (jj >= j1) break;
      double dot[CB];
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        dot[q] = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) dot[q] = fma(x[d][k], v[q][k], dot[q]);
      }
      block_sum<CB>(dot, red, buf);
      if (tid == 0) {
#pragma unroll
        for (int q = 0; q < CB; ++q)
          if (q < cb) G[q * ldg + jj] = alpha * dot[q];
      }
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const double gq = alpha * dot[q];
        ssq[q] = fma(gq, gq, ssq[q]);
      }
      if (jj + 2 < j1) load(x[d], jj + 2);
    }
  }
  if (tid == 0 && part_s) {
#pragma unroll
    for (int q = 0; q < CB; ++q) part_s[(int64_t)chunk * 4 + q] = ssq[q];
  }
}

// d_ssq[(k0+q)*nblocks + b] = sum over the chunks of block b of part_s[chunk][q]
__global__ __launch_bounds__(64) void k_reduce_ssq(const double* __restrict__ part_s, const int* __restrict__ blk_start,
                                                   int nblocks, int cb, double* __restrict__ out) {
  const int b = blockIdx.x, q = blockIdx.y, lane = threadIdx.x;
  if (q >= cb) return;
  double s = 0.0;
  for (int c = blk_start[b] + lane; c < blk_start[b + 1]; c += 64) s += part_s[(int64_t)c * 4 + q];
  s = wave_sum(s);
  if (lane == 0) out[(int64_t)q * nblocks + b] = s;
}

// ------------------------------------------------------------------------------------ x_w
// grid (nchunks, nslabs); a workgroup owns rows [slab*512, slab*512+512) of the columns of its chunk.
// NANM: 0 plain; 1 missing entries (NaN) contribute nothing (mbpls.py:1379-1383, :1108-1113);
//       2 the PRESENCE indicator takes the place of x (sums of W over the present columns of every row)
template <int CB, bool SCALE, int NANM = 0>
__global__ __launch_bounds__(256) void k_xw(const double* __restrict__ X, int64_t ld, int64_t n,
                                            const int64_t* __restrict__ c0, const int64_t* __restrict__ c1,
                                            const double* __restrict__ W, int64_t ldw, int cb,
                                            const double* __restrict__ mean, const double* __restrict__ scale,
                                            double* __restrict__ part /* [nchunks][CB][n] */) {
  const int chunk = blockIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * 512 + 2 * threadIdx.x;
  const int64_t j0 = c0[chunk], j1 = c1[chunk];
  const bool in0 = r0 < n, in1 = r0 + 1 < n, inld = r0 < ld;
  double acc[CB][2];
#pragma unroll
  for (int q = 0; q < CB; ++q) acc[q][0] = acc[q][1] = 0.0;
  const double* col = X + j0 * ld + r0;
#pragma unroll 4
  for (int64_t j = j0; j < j1; ++j, col += ld) {
    double2 t = make_double2(0.0, 0.0);
    if (inld) t = *reinterpret_cast<const double2*>(col);
    double x0 = in0 ? t.x : 0.0, x1 = in1 ? t.y : 0.0;
    if constexpr (SCALE) {
      const double mu = mean[j], rsc = scale[j];     // `scale` holds the RECIPROCALS here (k_recip, x_w_impl)
      x0 = in0 ? (x0 - mu) * rsc : 0.0;
      x1 = in1 ? (x1 - mu) * rsc : 0.0;
    }
    if constexpr (NANM == 1) {
      x0 = (x0 == x0) ? x0 : 0.0;
      x1 = (x1 == x1) ? x1 : 0.0;
    } else if constexpr (NANM == 2) {
      x0 = (in0 && x0 == x0) ? 1.0 : 0.0;
      x1 = (in1 && x1 == x1) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      const double w = (q < cb) ? W[q * ldw + j] : 0.0;  // workgroup-uniform -> scalar load
      acc[q][0] = fma(w, x0, acc[q][0]);
      acc[q][1] = fma(w, x1, acc[q][1]);
    }
  }
  double* out = part + (int64_t)chunk * CB * n;
#pragma unroll
  for (int q = 0; q < CB; ++q) {
    if (in0) out[q * n + r0] = acc[q][0];
    if (in1) out[q * n + r0 + 1] = acc[q][1];
  }
}

// The same product for FEW samples (n <= 256): a 512-row slab would leave most of the workgroup idle (n = 64: 32 of
// 256 threads).  Here N2 = pow2 >= n/2 threads cover the row pairs and the other 256 / N2 "column lanes" take every
// (256/N2)-th column of the chunk; the lanes' accumulators are added through LDS in lane order.
template <int CB, bool SCALE, int NANM = 0>
__global__ __launch_bounds__(256) void k_xw_small(const double* __restrict__ X, int64_t ld, int64_t n, int log2_n2,
                                                  const int64_t* __restrict__ c0, const int64_t* __restrict__ c1,
                                                  const double* __restrict__ W, int64_t ldw, int cb,
                                                  const double* __restrict__ mean, const double* __restrict__ scale,
                                                  double* __restrict__ part /* [nchunks][CB][n] */) {
  __shared__ double s_acc[256 * CB * 2];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int n2 = 1 << log2_n2, CL = 256 >> log2_n2;
  const int rp = tid & (n2 - 1), cl = tid >> log2_n2;
  const int64_t r0 = 2 * (int64_t)rp;
  const int64_t j0 = c0[chunk], j1 = c1[chunk];
  const bool in0 = r0 < n, in1 = r0 + 1 < n, inld = r0 < ld;
  double acc[CB][2];
#pragma unroll
  for (int q = 0; q < CB; ++q) acc[q][0] = acc[q][1] = 0.0;
#pragma unroll 4
  for (int64_t j = j0 + cl; j < j1; j += CL) {
    double2 t = make_double2(0.0, 0.0);
    if (inld) t = *reinterpret_cast<const double2*>(X + j * ld + r0);
    double x0 = in0 ? t.x : 0.0, x1 = in1 ? t.y : 0.0;
    if constexpr (SCALE) {
      const double mu = mean[j], rsc = scale[j];     // reciprocals (k_recip)
      x0 = in0 ? (x0 - mu) * rsc : 0.0;
      x1 = in1 ? (x1 - mu) * rsc : 0.0;
    }
    if constexpr (NANM == 1) {
      x0 = (x0 == x0) ? x0 : 0.0;
      x1 = (x1 == x1) ? x1 : 0.0;
    } else if constexpr (NANM == 2) {
      x0 = (in0 && x0 == x0) ? 1.0 : 0.0;
      x1 = (in1 && x1 == x1) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      const double w = (q < cb) ? W[q * ldw + j] : 0.0;
      acc[q][0] = fma(w, x0, acc[q][0]);
      acc[q][1] = fma(w, x1, acc[q][1]);
    }
  }
  // s_acc[cl][q][row]: 256 * CB * 2 doubles whatever n2 is
#pragma unroll
  for (int q = 0; q < CB; ++q) {
    s_acc[((cl * CB + q) << (log2_n2 + 1)) + 2 * rp] = acc[q][0];
    s_acc[((cl * CB + q) << (log2_n2 + 1)) + 2 * rp + 1] = acc[q][1];
  }
  __syncthreads();
  double* out = part + (int64_t)chunk * CB * n;
  for (int e = tid; e < CB * 2 * n2; e += 256) {
    const int q = e >> (log2_n2 + 1), row = e & (2 * n2 - 1);
    if (row < n) {
      double sum = 0.0;
      for (int l = 0; l < CL; ++l) sum += s_acc[((l * CB + q) << (log2_n2 + 1)) + row];
      out[(int64_t)q * n + row] = sum;
    }
  }
}

// out[(b or 0)][q][i] = sum over chunks (of block b, or all) of part[chunk][q][i], fixed order.
//   grid (ceil(n/64), cb, blocks), 256 threads: lane = row, wave w takes chunks cs+w, cs+w+4, ... with four independent
//   accumulators (16 loads in flight per row -- the sum is latency-bound), LDS combine in wave order.
__global__ __launch_bounds__(256) void k_reduce_xw(const double* __restrict__ part, int CB, int cb, int64_t n,
                                                   const int* __restrict__ blk_start, int nblocks, int per_block,
                                                   double* __restrict__ out, int64_t ldo, int c_total, int k0) {
  __shared__ double comb[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int q = blockIdx.y;
  const int b = blockIdx.z;
  const int cs = per_block ? blk_start[b] : blk_start[0];
  const int ce = per_block ? blk_start[b + 1] : blk_start[nblocks];
  double s = 0.0;
  if (i < n && q < cb) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int c = cs + w;
    for (; c + 12 < ce; c += 16) {
      s0 += part[((int64_t)c * CB + q) * n + i];
      s1 += part[((int64_t)(c + 4) * CB + q) * n + i];
      s2 += part[((int64_t)(c + 8) * CB + q) * n + i];
      s3 += part[((int64_t)(c + 12) * CB + q) * n + i];
    }
    for (; c < ce; c += 4) s0 += part[((int64_t)c * CB + q) * n + i];
    s = (s0 + s1) + (s2 + s3);
  }
  comb[w][lane] = s;
  __syncthreads();
  if (w == 0 && i < n && q < cb)
    out[(int64_t)b * ldo * c_total + (int64_t)(k0 + q) * ldo + i] = (comb[0][lane] + comb[1][lane]) + (comb[2][lane] + comb[3][lane]);
}

struct Cfg {
  int R, TMAX;
};
static bool pick_cfg(const mbpls_ctx* ctx, int64_t n, Cfg* cfg, int* T) {
  // fewer, fatter threads first (the largest R that still fills a wave), as in the column passes: the per-column
  // reductions are issue-bound
  static const Cfg table[] = {{16, 512}, {8, 512}, {4, 512}, {2, 256}, {8, 1024}};
  for (const Cfg& c : table) {
    if (c.TMAX != 1024 && (n + c.R - 1) / c.R < 64 && c.R > 2) continue;
    int64_t need = (n + c.R - 1) / c.R;
    need = (need + 63) / 64 * 64;
    if (need <= c.TMAX) {
      *cfg = c;
      *T = (int)need;
      return true;
    }
  }
  return false;
}

}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_xt_v_scaled(const mbpls_mat* m, const double* d_mean, const double* d_rscale, const double* d_V,
                          int64_t ldv, int c, double alpha, double* d_G, int64_t ldg, double* d_ssq) {
  MB_ARG(m && d_V && d_G && d_mean && d_rscale, "NULL argument");
  MB_ARG(c >= 1 && ldv >= m->n && ldg >= m->p, "bad shape (c=%d ldv=%lld ldg=%lld)", c, (long long)ldv,
         (long long)ldg);
  return xtv_mfma(m->ctx, m, d_V, ldv, c, alpha, d_G, ldg, d_ssq, d_mean, d_rscale);
}

int mbpls_hip_xt_v(const mbpls_mat* m, const double* d_V, int64_t ldv, int c, double alpha, double* d_G,
                   int64_t ldg, double* d_ssq) {
  MB_ARG(m && d_V && d_G, "NULL argument");
  MB_ARG(c >= 1 && ldv >= m->n && ldg >= m->p, "bad shape (c=%d ldv=%lld ldg=%lld)", c, (long long)ldv,
         (long long)ldg);
  mbpls_ctx* ctx = m->ctx;
  Cfg cfg;
  int T = 0;
  // one right-hand side is exactly the loadings-only column pass (p = alpha X^T ts, per-block ||p_b||^2), the
  // best-tuned single-dot kernel here (4.6 vs 5.3 ms at n=4096, p=1e6)
  if (c == 1 && m->n <= 8192 && m->n <= ctx->long_columns_from)
    return mbpls_hip_deflate_sweep(const_cast<mbpls_mat*>(m), d_V, alpha, 0, d_G, d_ssq, nullptr, 1.0, nullptr, nullptr,
                                   nullptr);
  // two right-hand sides: the fused loadings+sweep pass computes exactly these two dot products per column (its
  // score accumulation goes to a scratch vector): 4.7 vs 6.1 ms for the two-dot kernel below
  if (c == 2 && m->n <= 8192 && m->n <= ctx->long_columns_from) {
    MB_TRY(ensure_stage(ctx, sizeof(double) * (size_t)m->n * (size_t)m->nblocks));
    return mbpls_hip_deflate_sweep(const_cast<mbpls_mat*>(m), d_V, alpha, 0, d_G, d_ssq, d_V + ldv, alpha, d_G + ldg,
                                   (double*)ctx->stage, d_ssq ? d_ssq + m->nblocks : nullptr);
  }
  // many right-hand sides (or columns too long for the register-resident kernel): matrix-core path, X read once
  if (c >= 3 || !pick_cfg(ctx, m->n, &cfg, &T)) return xtv_mfma(ctx, m, d_V, ldv, c, alpha, d_G, ldg, d_ssq);
  const ChunkTable& ct = m->fine;
  MB_TRY(ensure_ws(ctx, sizeof(double) * (size_t)ct.nchunks * 4));
  double* part_s = (double*)ctx->ws;
  constexpr int CB = 2;
  for (int k0 = 0; k0 < c; k0 += CB) {
    const int cb = std::min(CB, c - k0);
    const double* V = d_V + (int64_t)k0 * ldv;
    double* G = d_G + (int64_t)k0 * ldg;
    dim3 grid((unsigned)ct.nchunks), block((unsigned)T);
    ProfScope prof(ctx, MBPLS_K_XT_V, 8.0 * (double)m->n * (double)m->p);
#define MB_LAUNCH(R_, TM_)                                                                                       \
  hipLaunchKernelGGL((k_xtv<R_, TM_, CB>), grid, block, 0, ctx->stream, (const double*)m->data, m->ld, m->n,     \
                     (const int64_t*)ct.d_c0, (const int64_t*)ct.d_c1, V, ldv, cb, alpha, G, ldg,                \
                     d_ssq ? part_s : (double*)nullptr)
    if (cfg.R == 2) MB_LAUNCH(2, 256);
    else if (cfg.R == 4) MB_LAUNCH(4, 512);
    else if (cfg.R == 8 && cfg.TMAX == 512) MB_LAUNCH(8, 512);
    else if (cfg.R == 8) MB_LAUNCH(8, 1024);
    else MB_LAUNCH(16, 512);
#undef MB_LAUNCH
    MB_HIP(hipGetLastError());
    if (d_ssq) {
      hipLaunchKernelGGL(k_reduce_ssq, dim3((unsigned)m->nblocks, (unsigned)cb), dim3(64), 0, ctx->stream,
                         (const double*)part_s, (const int*)ct.d_blk_start, m->nblocks, cb,
                         d_ssq + (int64_t)k0 * m->nblocks);
      MB_HIP(hipGetLastError());
    }
  }
  return MBPLS_OK;
}

}  // extern "C"

namespace mbpls {
void launch_reduce_xw(mbpls_ctx* ctx, const mbpls_mat* m, const double* part, int CB, int cb, int per_block,
                      double* d_out, int64_t ldo, int c_total, int k0) {
  dim3 rgrid((unsigned)((m->n + 63) / 64), (unsigned)cb, per_block ? (unsigned)m->nblocks : 1u);
  hipLaunchKernelGGL(k_reduce_xw, rgrid, dim3(256), 0, ctx->stream, part, CB, cb, m->n,
                     (const int*)m->coarse.d_blk_start, m->nblocks, per_block ? 1 : 0, d_out, ldo, c_total, k0);
}
}  // namespace mbpls

// r[j] = 1 / s[j]: the on-the-fly standardisation multiplies (a division per element cost more than the loads)
__global__ __launch_bounds__(256) void k_recip(int64_t p, const double* __restrict__ s, double* __restrict__ r) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < p) r[j] = 1.0 / s[j];
}

static int x_w_impl(const mbpls_mat* m, const double* d_W, int64_t ldw, int c, int per_block, const double* d_mean,
                    const double* d_scale, int nanm, double* d_out, int64_t ldo) {
  MB_ARG(m && d_W && d_out, "NULL argument");
  MB_ARG(c >= 1 && ldw >= m->p && ldo >= m->n, "bad shape (c=%d ldw=%lld ldo=%lld)", c, (long long)ldw,
         (long long)ldo);
  MB_ARG((d_mean == nullptr) == (d_scale == nullptr), "mean and scale must be given together");
  mbpls_ctx* ctx = m->ctx;
  // many right-hand sides: matrix-core path, X read once per 64 of them (this kernel: once per 8)
  if (c > ctx->xw_mfma_from) return xw_mfma(ctx, m, d_W, ldw, c, per_block, d_mean, d_scale, nanm, d_out, ldo);
  // one plain right-hand side: full-column streaming pass (d_out blocks are ldo apart: only the dense layout; the
  // per-chunk partial n-vectors must stay a small fraction of X: >= 128 columns per chunk)
  if (c == 1 && !d_mean && nanm == 0 && ldo == m->n && m->p >= 128 * (int64_t)m->fine.nchunks) {
    const int rc = axpy_cols(ctx, m, d_W, per_block, d_out);
    if (rc != MBPLS_ERR_UNSUPPORTED) return rc;
  }
  const ChunkTable& ct = m->coarse;
  constexpr int CB = 8;
  const size_t part_bytes = ((sizeof(double) * (size_t)ct.nchunks * CB * (size_t)m->n + 255) / 256) * 256;
  MB_TRY(ensure_ws(ctx, part_bytes + (d_scale ? sizeof(double) * (size_t)m->p : 0)));
  double* part = (double*)ctx->ws;
  if (d_scale && nanm != 2) {
    double* rs = (double*)((char*)ctx->ws + part_bytes);
    hipLaunchKernelGGL(k_recip, dim3((unsigned)((m->p + 255) / 256)), dim3(256), 0, ctx->stream, m->p, d_scale, rs);
    MB_HIP(hipGetLastError());
    d_scale = rs;
  }
  const unsigned slabs = (unsigned)((m->n + 511) / 512);
  for (int k0 = 0; k0 < c; k0 += CB) {
    const int cb = std::min(CB, c - k0);
    const double* W = d_W + (int64_t)k0 * ldw;
    dim3 grid((unsigned)ct.nchunks, slabs);
    const bool small = m->n <= 256 && ctx->sub_columns;
    int log2_n2 = 4;                                      // row-pair threads: 16 .. 128
    while ((2 << log2_n2) < m->n) ++log2_n2;
    {
    ProfScope prof(ctx, MBPLS_K_X_W, 8.0 * (double)m->n * (double)m->p);
#define MB_XW(SC_, NM_)                                                                                            \
  do {                                                                                                             \
    if (small)                                                                                                     \
      hipLaunchKernelGGL((k_xw_small<CB, SC_, NM_>), dim3((unsigned)ct.nchunks), dim3(256), 0, ctx->stream,        \
                         (const double*)m->data, m->ld, m->n, log2_n2, (const int64_t*)ct.d_c0,                    \
                         (const int64_t*)ct.d_c1, W, ldw, cb, d_mean, d_scale, part);                              \
    else                                                                                                           \
      hipLaunchKernelGGL((k_xw<CB, SC_, NM_>), grid, dim3(256), 0, ctx->stream, (const double*)m->data, m->ld,     \
                         m->n, (const int64_t*)ct.d_c0, (const int64_t*)ct.d_c1, W, ldw, cb, d_mean, d_scale,      \
                         part);                                                                                    \
  } while (0)
    if (nanm == 2) MB_XW(false, 2);                       // the indicator needs no scaling
    else if (nanm == 1 && d_mean) MB_XW(true, 1);
    else if (nanm == 1) MB_XW(false, 1);
    else if (d_mean) MB_XW(true, 0);
    else MB_XW(false, 0);
#undef MB_XW
    }
    MB_HIP(hipGetLastError());
    launch_reduce_xw(ctx, m, part, CB, cb, per_block, d_out, ldo, c, k0);
    MB_HIP(hipGetLastError());
  }
  return MBPLS_OK;
}

extern "C" {

int mbpls_hip_x_w(const mbpls_mat* m, const double* d_W, int64_t ldw, int c, int per_block, const double* d_mean,
                  const double* d_scale, double* d_out, int64_t ldo) {
  return x_w_impl(m, d_W, ldw, c, per_block, d_mean, d_scale, 0, d_out, ldo);
}

int mbpls_hip_x_w_nan(const mbpls_mat* m, const double* d_W, int64_t ldw, int c, int per_block, const double* d_mean,
                      const double* d_scale, int indicator, double* d_out, int64_t ldo) {
  return x_w_impl(m, d_W, ldw, c, per_block, d_mean, d_scale, indicator ? 2 : 1, d_out, ldo);
}

}  // extern "C"

// one kernel of this translation unit, for mbpls_hip_warmup (its code object is loaded on first use otherwise)
namespace mbpls {
const void* anchor_gemv() { return (const void*)&k_reduce_ssq; }
}  // namespace mbpls
