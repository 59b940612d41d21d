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
  __shared__ double part[16 * 16];     // wave partials of reduce_many
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

// ------------------------------------------------------------------------------------------------------------------
// The multi-response NIPALS inner loop (mbpls.py:841-913) in the q-dimensional space it actually lives in.
//
// Every sweep vector of the loop is u = Y c (c = e_1 for the start vector Y[:, 0], :838; afterwards u = Y v / v^T v,
// normalised, :911-913), and the super score it produces is ts ~ T sw ~ P (sum_b X_b X_b^T) P u  (see the Gram form
// above; P = I - Ts Ts^T carries the deflation :969).  With the n x q matrix
//     G = P X (X^T (P Y))                       -- two many-right-hand-side products over X per COMPONENT
// that is ts = G c / ||G c||: an iteration costs n x q work instead of two passes over X (or one product with an
// n x n Gram), and the whole loop of a component -- 50-500 iterations with several responses -- runs inside ONE kernel
// launch: one 1024-thread workgroup, the rows of `a`, `ts`, `u` in registers, G and Y (n x q each, L2-resident)
// re-read per iteration, three workgroup reductions per iteration (~10 us).  Same iterates as the reference's loop up
// to rounding; the stop rule is evaluated on the n-vector ts exactly as there.
namespace mbpls {

constexpr int SS_T = 512;      // one workgroup of 8 waves: 256 VGPRs per thread for the register-resident n-vectors
constexpr int SS_QMAX = 16;

// sum m <= M values over the workgroup: wave butterflies, then the first m threads combine the wave partials
template <int M>
__device__ __forceinline__ void ss_reduce(double (&v)[M], int m, double* red /* [SS_T/64][M] */, double* tot /* [M] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < M; ++k)
    if (k < m) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < M; ++k)
      if (k < m) red[w * M + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < m) {
    double s = 0.0;
    for (int i = 0; i < SS_T / 64; ++i) s += red[i * M + threadIdx.x];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < M; ++k)
    if (k < m) v[k] = tot[k];
}

struct SubspaceArgs {
  int64_t n;
  int q, norm_kind, max_iter;
  double tol;
  const double* G;      // n x q
  const double* Y;      // n x q
  const double* c0;     // q: coefficients of the start vector (u = Y c0)
  double* ts;           // n
  double* u;            // n: the updated, normalised u (U_ of the component)
  double* v;            // q
  double* c_used;       // q: coefficients of the sweep vector the LAST iteration was taken with
  double* state;        // [0] 1 converged / 2 iteration cap, [1] iterations, [2] last diff
};

// ts and the work vector (a = G c, then u = Y v / v^T v) live in LDS for the whole loop (2 x 8n bytes: n <= 8192);
// the q-vectors c, v are workgroup-uniform and sit in LDS as well.  Plain strided loops, nothing unrolled over rows:
// the kernel is latency-bound (three reductions per iteration), not throughput-bound.
__global__ __launch_bounds__(SS_T) void k_nipals_subspace_loop(SubspaceArgs s) {
  constexpr int M = 4 + SS_QMAX;
  extern __shared__ double s_vec[];      // [2][n]: ts, work
  __shared__ double red[(SS_T / 64) * M];
  __shared__ double tot[M];
  __shared__ double s_mx[SS_T / 64], s_mn[SS_T / 64];
  __shared__ double s_c[SS_QMAX], s_cu[SS_QMAX], s_v[SS_QMAX];
  const int tid = threadIdx.x;
  const int64_t n = s.n;
  const int q = s.q;
  double* s_ts = s_vec;
  double* s_wk = s_vec + n;
  if (tid < SS_QMAX) {
    s_c[tid] = (tid < q) ? s.c0[tid] : 0.0;
    s_cu[tid] = s_c[tid];
    s_v[tid] = 0.0;
  }
  for (int64_t i = tid; i < n; i += SS_T) s_ts[i] = 0.0;
  __syncthreads();
  int iters = 0;
  double diff = 0.0, flag = 0.0, un = 1.0;
  double rv[M];
  while (true) {
    ++iters;
    if (tid < SS_QMAX) s_cu[tid] = s_c[tid];
    // ---- a = G c;  ts = a / ||a||                                                        (:856-883 in Gram form)
#pragma unroll
    for (int k = 0; k < M; ++k) rv[k] = 0.0;
    for (int64_t i = tid; i < n; i += SS_T) {
      double acc = 0.0;
      for (int j = 0; j < q; ++j) acc = fma(s.G[(int64_t)j * n + i], s_c[j], acc);
      s_wk[i] = acc;
      rv[0] = fma(acc, acc, rv[0]);
    }
    ss_reduce<M>(rv, 1, red, tot);
    const double nrm = sqrt(rv[0]);
    // ---- diff = ||ts_old - ts|| (:887),  Y^T ts,  ts^T ts
#pragma unroll
    for (int k = 0; k < M; ++k) rv[k] = 0.0;
    double dmax = 0.0, dmin = 1.0e300;
    for (int64_t i = tid; i < n; i += SS_T) {
      const double t = s_wk[i] / nrm;
      const double d = s_ts[i] - t;
      s_ts[i] = t;
      rv[0] = fma(d, d, rv[0]);
      rv[1] += fabs(d);
      rv[2] = fma(t, t, rv[2]);
      dmax = fmax(dmax, fabs(d));
      dmin = fmin(dmin, fabs(d));
#pragma unroll
      for (int j = 0; j < SS_QMAX; ++j)
        if (j < q) rv[4 + j] = fma(s.Y[(int64_t)j * n + i], t, rv[4 + j]);
    }
    ss_reduce<M>(rv, 4 + q, red, tot);
    if (s.norm_kind == 0) diff = sqrt(rv[0]);
    else if (s.norm_kind == 1) diff = rv[1];
    else {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
        dmin = fmin(dmin, __shfl_xor(dmin, off, 64));
      }
      __syncthreads();
      if ((tid & 63) == 0) {
        s_mx[tid >> 6] = dmax;
        s_mn[tid >> 6] = dmin;
      }
      __syncthreads();
      for (int k = 0; k < SS_T / 64; ++k) {
        dmax = fmax(dmax, s_mx[k]);
        dmin = fmin(dmin, s_mn[k]);
      }
      diff = (s.norm_kind == 2) ? dmax : dmin;
    }
    // ---- v = Y^T ts / ts^T ts (:899);  u = Y v / v^T v, unit length (:911-913)  =>  c = v / (v^T v) / ||Y v / v^T v||
    double vv = 0.0;
#pragma unroll
    for (int j = 0; j < SS_QMAX; ++j) {
      const double vj = (j < q) ? rv[4 + j] / rv[2] : 0.0;
      vv = fma(vj, vj, vv);
      if (tid == j) s_v[j] = vj;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < M; ++k) rv[k] = 0.0;
    for (int64_t i = tid; i < n; i += SS_T) {
      double acc = 0.0;
      for (int j = 0; j < q; ++j) acc = fma(s.Y[(int64_t)j * n + i], s_v[j], acc);
      acc = acc / vv;
      s_wk[i] = acc;
      rv[0] = fma(acc, acc, rv[0]);
    }
    ss_reduce<M>(rv, 1, red, tot);
    un = sqrt(rv[0]);
    if (tid < SS_QMAX) s_c[tid] = s_v[tid] / vv / un;
    __syncthreads();
    if (iters >= 2 && diff <= s.tol) {
      flag = 1.0;
      break;
    }
    if (iters >= s.max_iter) {
      flag = 2.0;
      break;
    }
  }
  for (int64_t i = tid; i < n; i += SS_T) {
    s.ts[i] = s_ts[i];
    s.u[i] = s_wk[i] / un;
  }
  if (tid < q) {
    s.v[tid] = s_v[tid];
    s.c_used[tid] = s_cu[tid];
  }
  if (tid == 0) {
    s.state[0] = flag;
    s.state[1] = (double)iters;
    s.state[2] = diff;
  }
}


// The same loop for the 2-norm stop rule in q x q algebra (any n): with M1 = G^T G, M2 = Y^T G, M3 = Y^T Y every quantity
// of an iteration is a function of the coefficient vector c -- ||G c||^2 = c^T M1 c, v = M2 c~ (c~ = c / ||G c||, so
// ts = G c~ has unit length), ||Y v||^2 = v^T M3 v -- and the stop rule ||ts_old - ts|| = ||G (c~_old - c~)|| is taken on
// the DIFFERENCE of the coefficient vectors, sqrt(d^T M1 d) (formed in q dimensions it resolves 1e-14, where 2 - 2 cos
// would stop at 1e-8).  One wave iterates -- lane j owns row j of the three matrices, a mat-vec is q lane broadcasts
// and q fused multiply-adds, a quadratic form one DPP wave sum: ~1 us per iteration against ~34 us for the n-space loop
// above -- then the whole workgroup forms ts = G c~ (:882-883), u = Y v / v^T v (:911-913) and v = Y^T ts / ts^T ts (:899).
struct QLoopArgs {
  int64_t n;
  int q, max_iter;
  double tol;
  const double* G;      // n x q
  const double* Y;      // n x q
  const double* M;      // [3][q][q] row-major: G^T G, Y^T G, Y^T Y
  const double* c0;
  double* ts;
  double* u;
  double* v;
  double* c_used;
  double* state;
};

__device__ __forceinline__ double lane_value(double x, int k) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k), __builtin_amdgcn_readlane(__double2loint(x), k));
}

__global__ __launch_bounds__(SS_T) void k_nipals_subspace_qloop(QLoopArgs s) {
  constexpr int M = 2 + SS_QMAX;
  __shared__ double red[(SS_T / 64) * M];
  __shared__ double tot[M];
  __shared__ double s_ct[SS_QMAX], s_v[SS_QMAX], s_sc[4];
  const int tid = threadIdx.x;
  const int64_t n = s.n;
  const int q = s.q;
  if (tid < 64) {
    const int j = tid;
    double m1[SS_QMAX], m2[SS_QMAX], m3[SS_QMAX];
#pragma unroll
    for (int k = 0; k < SS_QMAX; ++k) {
      const bool in = j < q && k < q;
      m1[k] = in ? s.M[j * q + k] : 0.0;
      m2[k] = in ? s.M[q * q + j * q + k] : 0.0;
      m3[k] = in ? s.M[2 * q * q + j * q + k] : 0.0;
    }
    auto matvec = [&](const double (&m)[SS_QMAX], double x) {
      double y = 0.0;
#pragma unroll
      for (int k = 0; k < SS_QMAX; ++k) y = fma(m[k], lane_value(x, k), y);
      return y;
    };
    double c = j < q ? s.c0[j] : 0.0, cu = c, ct = 0.0, ct_old = 0.0, v = 0.0, vv = 1.0, diff = 1.0, flag = 0.0;
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
      if (iters >= 2 && diff <= s.tol) {
        flag = 1.0;
        break;
      }
      if (iters >= s.max_iter) {
        flag = 2.0;
        break;
      }
    }
    if (j < SS_QMAX) {
      s_ct[j] = ct;
      s_v[j] = v;
    }
    if (j < q) s.c_used[j] = cu;
    if (j == 0) {
      s_sc[0] = vv;
      s.state[0] = flag;
      s.state[1] = (double)iters;
      s.state[2] = diff;
    }
  }
  __syncthreads();
  // ts = G c~ / ||G c~||,  u = (Y v / v^T v) / || . ||: the norms first, the vectors in a second pass over G and Y
  double rv[M];
#pragma unroll
  for (int k = 0; k < M; ++k) rv[k] = 0.0;
  const double vv = s_sc[0];
  for (int64_t i = tid; i < n; i += SS_T) {
    double a = 0.0, b = 0.0;
    for (int j = 0; j < q; ++j) {
      a = fma(s.G[(int64_t)j * n + i], s_ct[j], a);
      b = fma(s.Y[(int64_t)j * n + i], s_v[j], b);
    }
    b = b / vv;
    rv[0] = fma(a, a, rv[0]);
    rv[1] = fma(b, b, rv[1]);
  }
  ss_reduce<M>(rv, 2, red, tot);
  const double na = sqrt(rv[0]), nb = sqrt(rv[1]);
#pragma unroll
  for (int k = 0; k < M; ++k) rv[k] = 0.0;
  for (int64_t i = tid; i < n; i += SS_T) {
    double a = 0.0, b = 0.0;
    for (int j = 0; j < q; ++j) {
      a = fma(s.G[(int64_t)j * n + i], s_ct[j], a);
      b = fma(s.Y[(int64_t)j * n + i], s_v[j], b);
    }
    const double t = a / na;
    s.ts[i] = t;
    s.u[i] = b / vv / nb;
    rv[0] = fma(t, t, rv[0]);
#pragma unroll
    for (int j = 0; j < SS_QMAX; ++j)
      if (j < q) rv[2 + j] = fma(s.Y[(int64_t)j * n + i], t, rv[2 + j]);
  }
  ss_reduce<M>(rv, 2 + q, red, tot);
  if (tid < q) s.v[tid] = tot[2 + tid] / tot[0];
}

}  // namespace mbpls

extern "C" int mbpls_hip_nipals_subspace_loop(mbpls_ctx* ctx, int64_t n, int q, const double* d_G, const double* d_Y,
                                              const double* d_c0, int norm_kind, double tol, int max_iter, double* d_ts,
                                              double* d_u, double* d_v, double* d_c_used, double* d_state) {
  using namespace mbpls;
  MB_ARG(ctx && d_G && d_Y && d_c0 && d_ts && d_u && d_v && d_c_used && d_state, "NULL argument");
  MB_ARG(n >= 1 && n <= 8192 && q >= 1 && q <= SS_QMAX, "subspace loop covers n <= 8192, q <= %d", SS_QMAX);
  MB_ARG(norm_kind >= 0 && norm_kind <= 3 && max_iter >= 1, "bad norm kind %d / max_iter %d", norm_kind, max_iter);
  SubspaceArgs s{n, q, norm_kind, max_iter, tol, d_G, d_Y, d_c0, d_ts, d_u, d_v, d_c_used, d_state};
  MB_ALLOW_LDS(ctx, k_nipals_subspace_loop, 16 * 8192);   // more than 64 KB of dynamic LDS: allowed once per device
  hipLaunchKernelGGL(k_nipals_subspace_loop, dim3(1), dim3(SS_T), (size_t)(16 * n), ctx->stream, s);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

extern "C" int mbpls_hip_nipals_subspace_qloop(mbpls_ctx* ctx, int64_t n, int q, const double* d_G, const double* d_Y,
                                               const double* d_c0, double tol, int max_iter, double* d_ts, double* d_u,
                                               double* d_v, double* d_c_used, double* d_state, double* d_moments) {
  using namespace mbpls;
  MB_ARG(ctx && d_G && d_Y && d_c0 && d_ts && d_u && d_v && d_c_used && d_state && d_moments, "NULL argument");
  MB_ARG(n >= 1 && q >= 1 && q <= SS_QMAX && max_iter >= 1, "q-space loop covers q <= %d", SS_QMAX);
  MB_TRY(mbpls_hip_cross_product_dev(ctx, n, d_G, n, q, d_G, n, q, d_moments));
  MB_TRY(mbpls_hip_cross_product_dev(ctx, n, d_Y, n, q, d_G, n, q, d_moments + q * q));
  MB_TRY(mbpls_hip_cross_product_dev(ctx, n, d_Y, n, q, d_Y, n, q, d_moments + 2 * q * q));
  QLoopArgs s{n, q, max_iter, tol, d_G, d_Y, d_moments, d_c0, d_ts, d_u, d_v, d_c_used, d_state};
  hipLaunchKernelGGL(k_nipals_subspace_qloop, dim3(1), dim3(SS_T), 0, ctx->stream, s);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}
