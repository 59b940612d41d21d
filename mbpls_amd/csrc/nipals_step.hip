// One iteration of the multi-response NIPALS inner loop (mbpls.py:841-913) in association-matrix space, entirely on the
// device, so that a whole batch of iterations can be queued without a host round trip per iteration.
//
// With A = sum_b X_b X_b^T (the per-block Grams side by side, see engine.gram_blocks), P = I - Ts Ts^T (the components
// found so far; deflation :969 is P A P) and z = P u:
//   t_b = X_b w_b = P A_b z / sqrt(z^T A_b z),   sw_b ~ t_b^T u = sqrt(z^T A_b z)      (P symmetric, idempotent)
//   => ts ~ T sw = P (sum_b A_b) z = P A z                                              (:875-883)
// i.e. the superscore iteration does not need the per-block products at all (they are taken once, at convergence,
// for T_b and the block importances).  Given a = A z from the row-slab product kernel this step does
//   ts = P a / ||P a||,  diff = ||ts_old - ts||  (:887),  v = Y^T ts / (ts^T ts)  (:899),
//   u = Y v / (v^T v), normalised  (:911-913),  z_next = P u
// in ONE workgroup (all of it is n x (k + q) work, L2-resident).  A device flag stops the updates once
// diff <= tol (never on the first pass, :884): iterations queued behind the converged one are no-ops, so the state
// the host reads back is exactly the state at the reference's loop exit.
#include "common.h"

namespace mbpls {

constexpr int NS_T = WG1024;   // one workgroup: the whole step is n x (k + q) L2-resident work, latency-bound -> many waves

struct StepArgs {
  int64_t n;
  const double* a;      // A z
  const double* Tp;     // n x k, column-major (ld n): superscores of the previous components
  int k;
  const double* Y;      // n x q, column-major (ld n)
  int q;
  int norm_kind;        // 0: 2-norm, 1: sum |.|, 2: max |.|, 3: min |.|   (matrix norms of the n x 1 difference)
  double tol;
  int max_iter;
  double* ts;           // in: ts of the previous iteration, out: this one
  double* u;
  double* v;
  double* z_next;       // ntile copies of P u, n apart (the right-hand side of the next product with the B Grams)
  int ntile;
  double* state;        // [0] done (1 converged, 2 iteration cap), [1] iterations, [2] last diff
};

constexpr int NS_KMAX = 16;   // previous components / responses (register arrays of the workgroup-wide dot products)

__global__ __launch_bounds__(NS_T) void k_nipals_gram_step(StepArgs s) {
  static_assert(NS_T == 1024, "reduce_many maps value g to wave g of 16");
  __shared__ double part[16 * NS_T];   // 128 KB: one workgroup owns the CU
  __shared__ double tot[16];
  __shared__ double s_max[NS_T / 64], s_min[NS_T / 64];
  if (s.state[0] != 0.0) return;   // converged earlier in this batch: keep the state as it is
  const int tid = threadIdx.x;
  const int64_t n = s.n;
  const int k = s.k, q = s.q;
  const int iters = (int)s.state[1] + 1;

  // ---- ts = P a / ||P a||
  double c[NS_KMAX];
#pragma unroll
  for (int j = 0; j < NS_KMAX; ++j) c[j] = 0.0;
  for (int64_t i = tid; i < n; i += NS_T) {
    const double ai = s.a[i];
#pragma unroll
    for (int j = 0; j < NS_KMAX; ++j)
      if (j < k) c[j] = fma(s.Tp[(int64_t)j * n + i], ai, c[j]);
  }
  reduce_many<NS_KMAX>(c, k, part, tot);
  double nrm[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = tid; i < n; i += NS_T) {
    double ai = s.a[i];
#pragma unroll
    for (int j = 0; j < NS_KMAX; ++j)
      if (j < k) ai = fma(-s.Tp[(int64_t)j * n + i], c[j], ai);
    s.z_next[i] = ai;              // scratch: P a (z_next is overwritten at the end)
    nrm[0] = fma(ai, ai, nrm[0]);
  }
  reduce_many<4>(nrm, 1, part, tot);
  const double inv = 1.0 / sqrt(nrm[0]);

  // ---- diff = ||ts_old - ts||,  Y^T ts,  ts^T ts
  double d4[4] = {0.0, 0.0, 0.0, 0.0};   // sum d^2, sum |d|, ts^T ts, -
  double dmax = 0.0, dmin = 1.0e300;
  double yv[NS_KMAX];
#pragma unroll
  for (int j = 0; j < NS_KMAX; ++j) yv[j] = 0.0;
  for (int64_t i = tid; i < n; i += NS_T) {
    const double t = s.z_next[i] * inv;
    const double d = s.ts[i] - t;
    s.ts[i] = t;
    d4[0] = fma(d, d, d4[0]);
    d4[1] += fabs(d);
    d4[2] = fma(t, t, d4[2]);
    dmax = fmax(dmax, fabs(d));
    dmin = fmin(dmin, fabs(d));
#pragma unroll
    for (int j = 0; j < NS_KMAX; ++j)
      if (j < q) yv[j] = fma(s.Y[(int64_t)j * n + i], t, yv[j]);
  }
  reduce_many<4>(d4, 3, part, tot);
  reduce_many<NS_KMAX>(yv, q, part, tot);
  double diff;
  if (s.norm_kind == 0) diff = sqrt(d4[0]);
  else if (s.norm_kind == 1) diff = d4[1];
  else {
    // max / min over the workgroup (order-independent)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
      dmin = fmin(dmin, __shfl_xor(dmin, off, 64));
    }
    if ((tid & 63) == 0) {
      s_max[tid >> 6] = dmax;
      s_min[tid >> 6] = dmin;
    }
    __syncthreads();
    for (int w = 0; w < NS_T / 64; ++w) {
      dmax = fmax(dmax, s_max[w]);
      dmin = fmin(dmin, s_min[w]);
    }
    diff = (s.norm_kind == 2) ? dmax : dmin;
  }

  // ---- v = Y^T ts / (ts^T ts);  u = Y v / (v^T v), normalised;  z_next = P u
  double vv = 0.0;
#pragma unroll
  for (int j = 0; j < NS_KMAX; ++j)
    if (j < q) {
      yv[j] = yv[j] / d4[2];
      vv = fma(yv[j], yv[j], vv);
    }
  double e[NS_KMAX];
#pragma unroll
  for (int j = 0; j < NS_KMAX; ++j) e[j] = 0.0;
  double un[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = tid; i < n; i += NS_T) {
    double w = 0.0;
#pragma unroll
    for (int j = 0; j < NS_KMAX; ++j)
      if (j < q) w = fma(s.Y[(int64_t)j * n + i], yv[j], w);
    w = w / vv;
    s.u[i] = w;                    // un-normalised for the moment
    un[0] = fma(w, w, un[0]);
#pragma unroll
    for (int j = 0; j < NS_KMAX; ++j)
      if (j < k) e[j] = fma(s.Tp[(int64_t)j * n + i], w, e[j]);
  }
  reduce_many<4>(un, 1, part, tot);
  reduce_many<NS_KMAX>(e, k, part, tot);
  const double iu = 1.0 / sqrt(un[0]);
  for (int64_t i = tid; i < n; i += NS_T) {
    const double ui = s.u[i] * iu;
    s.u[i] = ui;
    double zi = ui;
#pragma unroll
    for (int j = 0; j < NS_KMAX; ++j)
      if (j < k) zi = fma(-s.Tp[(int64_t)j * n + i], e[j] * iu, zi);
    for (int b = 0; b < s.ntile; ++b) s.z_next[(int64_t)b * n + i] = zi;
  }
#pragma unroll
  for (int j = 0; j < NS_KMAX; ++j)
    if (j < q && tid == 0) s.v[j] = yv[j];
  if (tid == 0) {
    s.state[1] = (double)iters;
    s.state[2] = diff;
    if (iters >= 2 && diff <= s.tol) s.state[0] = 1.0;
    else if (iters >= s.max_iter) s.state[0] = 2.0;
  }
}

}  // namespace mbpls

using namespace mbpls;

extern "C" int mbpls_hip_nipals_gram_step(mbpls_ctx* ctx, int64_t n, const double* d_a, const double* d_Tp, int k,
                                          const double* d_Y, int q, int norm_kind, double tol, int max_iter,
                                          double* d_ts, double* d_u, double* d_v, double* d_z_next, int ntile,
                                          double* d_state) {
  MB_ARG(ctx && d_a && d_Y && d_ts && d_u && d_v && d_z_next && d_state, "NULL argument");
  MB_ARG(n >= 1 && k >= 0 && k <= NS_KMAX && q >= 1 && q <= NS_KMAX && (k == 0 || d_Tp), "bad shape (n=%lld k=%d q=%d)",
         (long long)n, k, q);
  MB_ARG(norm_kind >= 0 && norm_kind <= 3 && ntile >= 1, "bad norm kind %d / ntile %d", norm_kind, ntile);
  StepArgs s{n, d_a, d_Tp, k, d_Y, q, norm_kind, tol, max_iter, d_ts, d_u, d_v, d_z_next, ntile, d_state};
  hipLaunchKernelGGL(k_nipals_gram_step, dim3(1), dim3(NS_T), 0, ctx->stream, s);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}
