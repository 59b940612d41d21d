// Cross-validation folds batched per read of X (mbpls_amd/cv_batch.py): the p x F algebra between the two
// many-right-hand-side products of a NIPALS sweep, for F folds at once.  Fold f has its own column statistics mu_f,
// 1/sigma_f (p x F, column-major like the p x F product G = X^T U it starts from):
//
//     g_jf = alpha_f (G_jf - mu_jf su_f) rs_jf          weights of fold f on the standardised columns   (mbpls.py:856)
//     C_jf = g_jf rs_jf                                 what the raw matrix is multiplied with           (mbpls.py:875)
//     ssq_bf = sum_{j in block b} g_jf^2,   cst_bf = sum_{j in block b} mu_jf C_jf      (block norms :860; the means' share
//                                                                                        of X_b C_f, removed per fold)
//
// One workgroup per (column chunk, fold); chunks never straddle a block, a second kernel adds the chunks of a block in
// fixed order.  Element-wise and HBM-bound: 4 p F doubles read, 2 p F written.
#include "common.h"

namespace mbpls {

__global__ __launch_bounds__(256) void k_cv_weights(int64_t p, const int64_t* __restrict__ c0, const int64_t* __restrict__ c1,
                                                    const double* __restrict__ G, const double* __restrict__ mu,
                                                    const double* __restrict__ rs, const double* __restrict__ su,
                                                    const double* __restrict__ alpha, double* __restrict__ g,
                                                    double* __restrict__ C, double* __restrict__ part, int F, int predict) {
  __shared__ double red[2 * 64];
  int buf = 0;
  const int f = blockIdx.y, chunk = blockIdx.x;
  const int64_t base = (int64_t)f * p;
  const double s = su[f], a = alpha[f];
  double v[2] = {0.0, 0.0};
  for (int64_t j = c0[chunk] + threadIdx.x; j < c1[chunk]; j += 256) {
    const double r = rs[base + j], m = mu[base + j];
    const double gj = predict ? G[base + j] * r : a * ((G[base + j] - m * s) * r);
    const double cj = predict ? gj : gj * r;       // predict: G = beta (standardised space), X is multiplied with beta rs
    g[base + j] = gj;
    C[base + j] = cj;
    v[0] = fma(gj, gj, v[0]);
    v[1] = fma(m, cj, v[1]);
  }
  block_sum<2>(v, red, buf);
  if (threadIdx.x == 0) {
    part[((int64_t)chunk * F + f) * 2] = v[0];
    part[((int64_t)chunk * F + f) * 2 + 1] = v[1];
  }
}

// out[(b * F + f) * 2 + {0, 1}] = sum over the chunks of block b, fixed order; one wave per (block, fold)
__global__ __launch_bounds__(64) void k_cv_block_sums(const double* __restrict__ part, const int* __restrict__ blk_start,
                                                      int F, double* __restrict__ out) {
  const int b = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int c = blk_start[b] + lane; c < blk_start[b + 1]; c += 64) {
    s0 += part[((int64_t)c * F + f) * 2];
    s1 += part[((int64_t)c * F + f) * 2 + 1];
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) {
    out[((int64_t)b * F + f) * 2] = s0;
    out[((int64_t)b * F + f) * 2 + 1] = s1;
  }
}

// Column statistics of the training rows of every fold from the full-data statistics and the held-out rows' centred
// sums h1 = sum_held (x - mu0), h2 = sum_held (x - mu0)^2 (two many-right-hand-side products with the held-out
// indicators): mean_f = mu0 - h1 / n_f, n_f var_f = n var0 - h2 - h1^2 / n_f (corrected two-pass form around mu0),
// sklearn's constant-column rule (scale 1).  Entries whose variance is lost to cancellation (< 1e-9 of the full-data sum
// of squares: a column that is all but constant on one fold's training rows) get rs = -1 and are counted: the host takes
// those few from the rows themselves.
__global__ __launch_bounds__(256) void k_cv_fold_stats(int64_t p, int F, double n, const double* __restrict__ mu0,
                                                       const double* __restrict__ var0, const double* __restrict__ h1,
                                                       const double* __restrict__ h2, const double* __restrict__ nf,
                                                       double* __restrict__ mean, double* __restrict__ rs,
                                                       unsigned long long* __restrict__ risky) {
  const int f = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= p) return;
  const int64_t e = (int64_t)f * p + j;
  const double n_f = nf[f], ss0 = n * var0[j];
  const double dm = -h1[e] / n_f;
  const double ssq = ss0 - h2[e] - n_f * dm * dm;
  const double m = mu0[j] + dm;
  mean[e] = m;
  if (ss0 > 0.0 && ssq <= 1e-9 * ss0) {
    rs[e] = -1.0;
    atomicAdd(risky, 1ull);
    return;
  }
  const double var = fmax(ssq, 0.0) / n_f;
  const double eps = 2.220446049250313e-16;
  const double nme = n_f * m * eps;
  rs[e] = (var <= n_f * eps * var + nme * nme) ? 1.0 : 1.0 / sqrt(var);
}


// The NIPALS inner loop (mbpls.py:841-913) of F cross-validation folds at once, each in the q-dimensional space of its
// iterates (nipals_step.hip k_nipals_subspace_qloop for ONE problem): one WAVE per fold, lane j owns row j of the fold's
// three q x q moment matrices M1 = G^T G, M2 = Y^T G, M3 = Y^T Y; an iteration is a handful of wave reductions.
// out[f] = [c~ (q) | c of the last sweep vector (q) | flag (1 converged, 2 capped) | iterations | last diff]
constexpr int CVQ_MAX = 16;
__device__ __forceinline__ double cv_lane_value(double x, int k) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k), __builtin_amdgcn_readlane(__double2loint(x), k));
}
__global__ __launch_bounds__(64) void k_cv_qloops(int q, int max_iter, double tol, const double* __restrict__ Mall,
                                                   double* __restrict__ out) {
  const int f = blockIdx.x, j = threadIdx.x;
  const double* M = Mall + (int64_t)f * 3 * q * q;
  double m1[CVQ_MAX], m2[CVQ_MAX], m3[CVQ_MAX];
#pragma unroll
  for (int k = 0; k < CVQ_MAX; ++k) {
    const bool in = j < q && k < q;
    m1[k] = in ? M[j * q + k] : 0.0;
    m2[k] = in ? M[q * q + j * q + k] : 0.0;
    m3[k] = in ? M[2 * q * q + j * q + k] : 0.0;
  }
  auto matvec = [&](const double (&m)[CVQ_MAX], double x) {
    double y = 0.0;
#pragma unroll
    for (int k = 0; k < CVQ_MAX; ++k) y = fma(m[k], cv_lane_value(x, k), y);
    return y;
  };
  double c = (j == 0) ? 1.0 : 0.0, cu = c, ct = 0.0, ct_old = 0.0, v = 0.0, vv = 1.0, diff = 1.0, flag = 0.0;   // start: Y[:, 0] (:838)
  int iters = 0;
  while (true) {
    ++iters;
    cu = c;
    ct = c / sqrt(wave_sum(c * matvec(m1, c)));                       // ts = G c~, unit length (:882-883)
    if (iters > 1) {
      const double d = ct_old - ct;
      diff = sqrt(fmax(wave_sum(d * matvec(m1, d)), 0.0));            // (:887)
    }
    ct_old = ct;
    v = matvec(m2, ct);                                               // (:899)
    vv = wave_sum(v * v);
    const double un = sqrt(wave_sum(v * matvec(m3, v))) / vv;         // ||Y v / v^T v|| (:911-913)
    c = v / vv / un;
    if (iters >= 2 && diff <= tol) {
      flag = 1.0;
      break;
    }
    if (iters >= max_iter || !(diff == diff)) {                       // (a NaN anywhere: give up, the caller falls back)
      flag = 2.0;
      break;
    }
  }
  double* o = out + (int64_t)f * (2 * q + 3);
  if (j < q) {
    o[j] = ct;
    o[q + j] = cu;
  }
  if (j == 0) {
    o[2 * q] = flag;
    o[2 * q + 1] = (double)iters;
    o[2 * q + 2] = diff;
  }
}

}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_cv_weights(const mbpls_mat* m, int F, int predict, const double* d_G, const double* d_mu, const double* d_rs,
                         const double* d_su, const double* d_alpha, double* d_g, double* d_C, double* d_sums) {
  MB_ARG(m && d_G && d_mu && d_rs && d_g && d_C && d_sums && (predict || (d_su && d_alpha)), "NULL argument");
  MB_ARG(F >= 1 && F <= 65535, "bad fold count %d", F);
  mbpls_ctx* ctx = m->ctx;
  const int nch = m->fine.nchunks;
  MB_TRY(ensure_ws(ctx, sizeof(double) * (size_t)nch * (size_t)F * 2));
  double* part = (double*)ctx->ws;
  hipLaunchKernelGGL(k_cv_weights, dim3((unsigned)nch, (unsigned)F), dim3(256), 0, ctx->stream, m->p,
                     (const int64_t*)m->fine.d_c0, (const int64_t*)m->fine.d_c1, d_G, d_mu, d_rs, predict ? d_rs : d_su,
                     predict ? d_rs : d_alpha, d_g, d_C, part, F, predict);
  MB_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_cv_block_sums, dim3((unsigned)m->nblocks, (unsigned)F), dim3(64), 0, ctx->stream,
                     (const double*)part, (const int*)m->fine.d_blk_start, F, d_sums);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

int mbpls_hip_xt_v_centred(const mbpls_mat* m, const double* d_mean, int square, const double* d_V, int64_t ldv, int c,
                           double* d_G, int64_t ldg) {
  MB_ARG(m && d_mean && d_V && d_G, "NULL argument");
  MB_ARG(c >= 1 && ldv >= m->n && ldg >= m->p, "bad shape (c=%d ldv=%lld ldg=%lld)", c, (long long)ldv, (long long)ldg);
  return xtv_mfma(m->ctx, m, d_V, ldv, c, 1.0, d_G, ldg, nullptr, d_mean, nullptr, square ? 1 : 0);
}

int mbpls_hip_cv_fold_stats(const mbpls_mat* m, int F, const double* d_mu0, const double* d_var0, const double* d_h1,
                            const double* d_h2, const double* d_nf, double* d_mean, double* d_rs, double* d_risky) {
  MB_ARG(m && d_mu0 && d_var0 && d_h1 && d_h2 && d_nf && d_mean && d_rs && d_risky, "NULL argument");
  MB_ARG(F >= 1 && F <= 65535, "bad fold count %d", F);
  mbpls_ctx* ctx = m->ctx;
  MB_HIP(hipMemsetAsync(d_risky, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_cv_fold_stats, dim3((unsigned)((m->p + 255) / 256), (unsigned)F), dim3(256), 0, ctx->stream, m->p, F,
                     (double)m->n, d_mu0, d_var0, d_h1, d_h2, d_nf, d_mean, d_rs, (unsigned long long*)d_risky);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

int mbpls_hip_cv_qloops(mbpls_ctx* ctx, int folds, int q, const double* d_moments, double tol, int max_iter, double* d_out) {
  MB_ARG(ctx && d_moments && d_out, "NULL argument");
  MB_ARG(folds >= 1 && q >= 1 && q <= CVQ_MAX && max_iter >= 1, "q-space loops cover q <= %d", CVQ_MAX);
  hipLaunchKernelGGL(k_cv_qloops, dim3((unsigned)folds), dim3(64), 0, ctx->stream, q, max_iter, tol, d_moments, d_out);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

}  // extern "C"

// one kernel of this translation unit, for mbpls_hip_warmup (its code object is loaded on first use otherwise)
namespace mbpls {
const void* anchor_cv_batch() { return (const void*)&k_cv_block_sums; }
}  // namespace mbpls
