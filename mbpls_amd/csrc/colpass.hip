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
#include "colpass_util.h"

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
  const double* rowmask;   // standardise pass only: samples excluded from the fit (cross-validation folds)
  double n_eff;            // number of included samples (used instead of n for mean / variance)
  // lazy (implicit) deflation: X in HBM stays X_0; X_k = X_0 - sum_{i<kprev} ts_i P[:,i]^T is applied algebraically
  int lazy;            // 0: explicit in-place deflation, 1: implicit
  int kprev;           // components already deflated (<= KD-1)
  const double* Pm;    // p x K loadings of the previous components (column-major, leading dim ldp)
  int64_t ldp;
  const double* cu;    // [kprev]  <ts_i, v>   corrections of the sweep dot
  const double* ce;    // [kprev]  <ts_i, ts>  corrections of the loadings dot
  double* part_d;      // [nchunks][KD]  sum_j g_j P[j,i] (i < kprev) and sum_j g_j p_j (i = kprev)
};
constexpr int KD = 40;


// The n=4096 workhorse (R=8, 512 threads = 8 waves) must stay at <= 128 VGPRs: 4 waves/SIMD = two
// workgroups per CU.  At 129+ VGPRs only ONE 8-wave workgroup fits a CU and the pass runs ~2x slower
// (measured: the lazy loadings+sweep variant at 157 VGPRs took 9.3 ms instead of 4.8 ms).
template <int R, int TMAX, int D, bool STD, bool DEFL, bool SWEEP, bool FULL>
__global__ __launch_bounds__(TMAX, (R == 8 && TMAX == 512) ? 4 : 1) void k_colpass(PassArgs a) {
  __shared__ double red[2 * 64];
  int buf = 0;
  const int tid = threadIdx.x, T = blockDim.x;
  const int64_t n = a.n, ld = a.ld;
  const int chunk = blockIdx.x;
  const int64_t j0 = uniform64(a.c0[chunk]), j1 = uniform64(a.c1[chunk]);
  RowMap<R> rm;
  rm.init(n, ld, tid, T, STD ? a.rowmask : nullptr);

  // The fused loadings+sweep variant of the n=4096 configuration needs x[2], ts, v, acc = 80 VGPRs of
  // column data; with the lazy-deflation state that no longer fits 128.  Its score vector ts therefore
  // lives in LDS (32 KB, each thread re-reads only its own row pairs: conflict-free ds_read_b128).
  constexpr bool TS_LDS = DEFL && SWEEP && R == 8 && TMAX == 512;
  __shared__ double2 s_ts[TS_LDS ? (TMAX * R / 2) : 1];
  __shared__ double2 s_v[TS_LDS ? (TMAX * R / 2) : 1];   // same treatment for the sweep vector there
  double v[(SWEEP && !TS_LDS) ? R : 1], ts[(DEFL && !TS_LDS) ? R : 1], acc[SWEEP ? R : 1];
  double tsv = 0.0;
  if constexpr (SWEEP) {
    if constexpr (TS_LDS) {
      double tmp[R];
      load_vec<R>(tmp, a.v, n, tid, T);
#pragma unroll
      for (int k = 0; k < R / 2; ++k) s_v[k * T + tid] = make_double2(tmp[2 * k], tmp[2 * k + 1]);
    } else {
      load_vec<R>(v, a.v, n, tid, T);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = 0.0;
  }
  auto v_pair = [&](int k) -> double2 {
    if constexpr (TS_LDS) return s_v[k * T + tid];
    else return make_double2(v[2 * k], v[2 * k + 1]);
  };
  if constexpr (DEFL && !TS_LDS) load_vec<R>(ts, a.ts, n, tid, T);
  if constexpr (TS_LDS) {
    double tmp[R];
    load_vec<R>(tmp, a.ts, n, tid, T);
#pragma unroll
    for (int k = 0; k < R / 2; ++k) s_ts[k * T + tid] = make_double2(tmp[2 * k], tmp[2 * k + 1]);
  }
  // row pair k of the score vector, from registers or LDS (a thread only ever reads what it wrote)
  auto ts_pair = [&](int k) -> double2 {
    if constexpr (TS_LDS) return s_ts[k * T + tid];
    else return make_double2(ts[2 * k], ts[2 * k + 1]);
  };
  if constexpr (DEFL && SWEEP) {
    double s[1] = {0.0};
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      const double2 t2 = ts_pair(k);
      const double2 v2 = v_pair(k);
      s[0] = fma(t2.x, v2.x, s[0]);
      s[0] = fma(t2.y, v2.y, s[0]);
    }
    block_sum<1>(s, red, buf);
    tsv = s[0];
  }
  double ssq_g = 0.0, ssq_p = 0.0, ssq_x = 0.0;
  const double dn = (STD && a.rowmask) ? a.n_eff : (double)n;   // samples that take part in the statistics
  // lazy deflation: thread i < kprev carries the correction of previous component i
  const bool lazy = !STD && a.lazy != 0;
  const bool mine = lazy && tid < a.kprev;
  const double cu_i = (mine && a.cu) ? a.cu[tid] : 0.0;
  const double ce_i = (mine && a.ce) ? a.ce[tid] : 0.0;
  const double* prow = mine ? a.Pm + (int64_t)tid * a.ldp : nullptr;
  double dacc = 0.0, dk = 0.0;

  double x[D][R];
  double pp[D];   // P[column, tid] of the column held in slot d (0 for threads >= kprev)
#pragma unroll
  for (int d = 0; d < D; ++d) {
    pp[d] = 0.0;
    if (j0 + d < j1) {
      load_col<R, FULL>(x[d], a.src + (j0 + d) * ld, rm);
      if (mine) pp[d] = prow[j0 + d];
    }
  }

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
              const bool real = FULL || ((((k & 1) ? rm.in_n1 : rm.in_n0) >> (k >> 1)) & 1u);
              const double dlt = real ? xc[k] - mean : 0.0;
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
              const bool real = FULL || ((((k & 1) ? rm.in_n1 : rm.in_n0) >> (k >> 1)) & 1u);
              xc[k] = real ? xc[k] - mean : 0.0;
            }
          }
#pragma unroll
          for (int k = 0; k < R; ++k) xc[k] = xc[k] / scale;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) ssq_x = fma(xc[k], xc[k], ssq_x);
        store_col<R, FULL>(xc, a.dst + jj * ld, rm);
      }

      double gj = 0.0;
      if constexpr (DEFL) {
        // lazy: <x_k, ts> = <x_0, ts> - sum_i P[j,i] <ts_i, ts>, folded into the same reduction
        double ab[2] = {-pp[d] * ce_i, -pp[d] * cu_i};
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
          const double2 t2 = ts_pair(k);
          ab[0] = fma(xc[2 * k], t2.x, ab[0]);
          ab[0] = fma(xc[2 * k + 1], t2.y, ab[0]);
          if constexpr (SWEEP) {
            const double2 v2 = v_pair(k);
            ab[1] = fma(xc[2 * k], v2.x, ab[1]);
            ab[1] = fma(xc[2 * k + 1], v2.y, ab[1]);
          }
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
          if (!lazy) {
#pragma unroll
            for (int k = 0; k < R / 2; ++k) {
              const double2 t2 = ts_pair(k);
              xc[2 * k] = fma(-pj, t2.x, xc[2 * k]);
              xc[2 * k + 1] = fma(-pj, t2.y, xc[2 * k + 1]);
            }
            store_col<R, FULL>(xc, a.dst + jj * ld, rm);
          }
          if constexpr (SWEEP) gj = a.alpha * fma(-pj, tsv, ab[1]);  // <x - p ts, v> = <x,v> - p <ts,v>
          if constexpr (SWEEP) dk = fma(gj, pj, dk);
        } else {
          if constexpr (SWEEP) gj = a.alpha * ab[1];
        }
      } else if constexpr (SWEEP) {
        double b[1] = {-pp[d] * cu_i};
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
        dacc = fma(gj, pp[d], dacc);
      }
      // refill this register slot with the column D ahead (in flight while the next ones are processed)
      if (jj + D < j1) {
        load_col<R, FULL>(xc, a.src + (jj + D) * ld, rm);
        if (mine) pp[d] = prow[jj + D];
      }
    }
  }

  if constexpr (SWEEP) {
    double* out = a.part_t + (int64_t)chunk * n;
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      const int64_t r0 = 2 * ((int64_t)k * T + tid);
      if (FULL || r0 + 1 < n) *reinterpret_cast<double2*>(out + r0) = make_double2(acc[2 * k], acc[2 * k + 1]);
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
  if (lazy && a.part_d) {
    double* pd = a.part_d + (int64_t)chunk * KD;
    if (mine) pd[tid] = dacc;
    if (tid == 0) pd[a.kprev] = dk;
  }
}

// Second stage: fixed-order sum of the chunk partials of each block.
//   grid (ceil(n/64), nblocks), 256 threads: wave w takes chunks cs+w, cs+w+4, ...; LDS combine in wave order.
//   The workgroup with blockIdx.x == 0 also reduces the per-chunk scalars of its block.
__global__ __launch_bounds__(256) void k_reduce_chunks(const double* __restrict__ part_t,
                                                       const double* __restrict__ part_s,
                                                       const int* __restrict__ blk_start, int64_t n,
                                                       double* __restrict__ out_t, double* __restrict__ out_ssq_g,
                                                       double* __restrict__ out_ssq_p, double* __restrict__ out_ssq_x,
                                                       const double* __restrict__ part_d, int nd,
                                                       double* __restrict__ out_d /* [nblocks][nd] */) {
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
  if (blockIdx.x == 0 && w == 3 && out_d) {
    for (int i = 0; i < nd; ++i) {
      double s = 0.0;
      for (int c = cs + lane; c < ce; c += 64) s += part_d[(int64_t)c * KD + i];
      s = wave_sum(s);
      if (lane == 0) out_d[(int64_t)b * nd + i] = s;
    }
  }
}

// ------------------------------------------------------------------------------- dispatch

template <bool STD, bool DEFL, bool SWEEP>
static int launch_pass(mbpls_ctx* ctx, const ChunkTable& ct, const PassArgs& a, int64_t p) {
  // algorithmic bytes: read 8np, plus 8np write when the pass stores X
  const bool writes = STD || (DEFL && a.deflate && !a.lazy);
  const int kind = STD ? MBPLS_K_STANDARDIZE : (DEFL ? (SWEEP ? MBPLS_K_DEFLATE_SWEEP : MBPLS_K_DEFLATE) : MBPLS_K_SWEEP);
  ProfScope prof(ctx, kind, 8.0 * (double)a.n * (double)p * (writes ? 2.0 : 1.0));
  Cfg cfg;
  int T = 0;
  if (!pick_cfg(ctx, a.n, &cfg, &T, STD)) {
    set_error("column passes support n <= 65536 (got n=%lld)", (long long)a.n);
    return MBPLS_ERR_UNSUPPORTED;
  }
  const PassArgs& an = a;
  dim3 grid((unsigned)ct.nchunks), block((unsigned)T);
  const int D = ctx->prefetch >= 1 && ctx->prefetch <= 3 ? ctx->prefetch : 2;
  // FULL: every thread's every row pair is a real sample and columns are unpadded (n=4096, 8192, ...)
  const bool full = (a.n == (int64_t)T * cfg.R) && (a.ld == a.n) && cfg.R >= 8 && !(STD && a.rowmask);
#define MB_LAUNCH(R_, TM_, D_)                                                                          \
  do {                                                                                                  \
    if (full && (R_) >= 8)                                                                              \
      hipLaunchKernelGGL((k_colpass<R_, TM_, D_, STD, DEFL, SWEEP, ((R_) >= 8)>), grid, block, 0, ctx->stream, an); \
    else                                                                                                \
      hipLaunchKernelGGL((k_colpass<R_, TM_, D_, STD, DEFL, SWEEP, false>), grid, block, 0, ctx->stream, an);       \
  } while (0)
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
  else if (cfg.R == 32) hipLaunchKernelGGL((k_colpass<32, 1024, 1, STD, DEFL, SWEEP, false>), grid, block, 0, ctx->stream, an);
  else if (cfg.R == 64) hipLaunchKernelGGL((k_colpass<64, 1024, 1, STD, DEFL, SWEEP, false>), grid, block, 0, ctx->stream, an);
  else {
    set_error("no kernel for R=%d TMAX=%d", cfg.R, cfg.TMAX);
    return MBPLS_ERR_UNSUPPORTED;
  }
#undef MB_LAUNCH_D
#undef MB_LAUNCH
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

int reduce_chunks(mbpls_ctx* ctx, const mbpls_mat* m, const double* part_t, const double* part_s,
                  double* d_t, double* d_ssq_g, double* d_ssq_p, double* d_ssq_x,
                  const double* part_d, int nd, double* d_d) {
  dim3 grid(d_t ? (unsigned)((m->n + 63) / 64) : 1u, (unsigned)m->nblocks);
  hipLaunchKernelGGL(k_reduce_chunks, grid, dim3(256), 0, ctx->stream, part_t, part_s,
                     (const int*)m->fine.d_blk_start, m->n, d_t, d_ssq_g, d_ssq_p, d_ssq_x, part_d, nd, d_d);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

static int partial_buffers(mbpls_ctx* ctx, const mbpls_mat* m, double** part_t, double** part_s,
                           double** part_d = nullptr) {
  const size_t nt = (size_t)m->fine.nchunks * (size_t)m->n, ns = (size_t)m->fine.nchunks * 4;
  const size_t nd = (size_t)m->fine.nchunks * KD;
  MB_TRY(ensure_ws(ctx, sizeof(double) * (nt + ns + nd)));
  *part_t = (double*)ctx->ws;
  *part_s = *part_t + nt;
  if (part_d) *part_d = *part_s + ns;
  return MBPLS_OK;
}

}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_sweep(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t,
                    double* d_ssq_g) {
  MB_ARG(m && d_v, "NULL argument");
  mbpls_ctx* ctx = m->ctx;
  if (long_columns(ctx, m->n)) return long_sweep(m, d_v, alpha, d_g, d_t, d_ssq_g);
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
  if (long_columns(ctx, m->n))
    return long_deflate_sweep(m, d_ts, inv_tt, deflate, d_p, d_ssq_p, d_v, alpha, d_g, d_t, d_ssq_g);
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

int mbpls_hip_sweep_lazy(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t,
                         double* d_ssq_g, const double* d_P, int64_t ldp, int kprev, const double* d_cu,
                         double* d_d) {
  MB_ARG(m && d_v, "NULL argument");
  MB_ARG(kprev >= 0 && kprev < KD && (kprev == 0 || (d_P && d_cu && d_d)), "bad lazy-deflation arguments (kprev=%d)", kprev);
  mbpls_ctx* ctx = m->ctx;
  if (long_columns(ctx, m->n)) {
    set_error("lazy deflation needs register-resident columns (n=%lld): use the explicit passes", (long long)m->n);
    return MBPLS_ERR_UNSUPPORTED;
  }
  PassArgs a{};
  MB_TRY(partial_buffers(ctx, m, &a.part_t, &a.part_s, &a.part_d));
  a.src = m->data; a.dst = nullptr; a.ld = m->ld; a.n = m->n;
  a.c0 = m->fine.d_c0; a.c1 = m->fine.d_c1;
  a.v = d_v; a.alpha = alpha; a.g = d_g;
  a.lazy = 1; a.kprev = kprev; a.Pm = d_P; a.ldp = ldp; a.cu = d_cu; a.ce = nullptr;
  MB_TRY((launch_pass<false, false, true>(ctx, m->fine, a, m->p)));
  return reduce_chunks(ctx, m, a.part_t, a.part_s, d_t, d_ssq_g, nullptr, nullptr, a.part_d, kprev, kprev ? d_d : nullptr);
}

int mbpls_hip_loadings_sweep_lazy(const mbpls_mat* m, const double* d_ts, double inv_tt, double* d_p,
                                  double* d_ssq_p, const double* d_v, double alpha, double* d_g, double* d_t,
                                  double* d_ssq_g, const double* d_P, int64_t ldp, int kprev, const double* d_cu,
                                  const double* d_ce, double* d_d) {
  MB_ARG(m && d_ts, "NULL argument");
  MB_ARG(kprev >= 0 && kprev < KD - 1 && (kprev == 0 || (d_P && d_ce)), "bad lazy-deflation arguments (kprev=%d)", kprev);
  MB_ARG(!d_v || (d_d && (kprev == 0 || d_cu)), "sweep part needs d_cu / d_d");
  mbpls_ctx* ctx = m->ctx;
  if (long_columns(ctx, m->n)) {
    set_error("lazy deflation needs register-resident columns (n=%lld): use the explicit passes", (long long)m->n);
    return MBPLS_ERR_UNSUPPORTED;
  }
  PassArgs a{};
  MB_TRY(partial_buffers(ctx, m, &a.part_t, &a.part_s, &a.part_d));
  a.src = m->data; a.dst = m->data; a.ld = m->ld; a.n = m->n;
  a.c0 = m->fine.d_c0; a.c1 = m->fine.d_c1;
  a.ts = d_ts; a.inv_tt = inv_tt; a.deflate = d_v ? 1 : 0; a.pvec = d_p;   // deflate=1: the fused sweep sees X_{k+1}
  a.v = d_v; a.alpha = alpha; a.g = d_g;
  a.lazy = 1; a.kprev = kprev; a.Pm = d_P; a.ldp = ldp; a.cu = d_cu; a.ce = d_ce;
  if (d_v) {
    MB_TRY((launch_pass<false, true, true>(ctx, m->fine, a, m->p)));
    return reduce_chunks(ctx, m, a.part_t, a.part_s, d_t, d_ssq_g, d_ssq_p, nullptr, a.part_d, kprev + 1, d_d);
  }
  MB_TRY((launch_pass<false, true, false>(ctx, m->fine, a, m->p)));
  if (d_ssq_p) return reduce_chunks(ctx, m, a.part_t, a.part_s, nullptr, nullptr, d_ssq_p, nullptr);
  return MBPLS_OK;
}

int mbpls_hip_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode, double* d_mean, double* d_scale,
                                double* d_ssq_x, const double* d_v, double alpha, double* d_g, double* d_t,
                                double* d_ssq_g) {
  return mbpls_hip_standardize_sweep_masked(src, dst, mode, nullptr, 0.0, d_mean, d_scale, d_ssq_x, d_v, alpha, d_g,
                                            d_t, d_ssq_g);
}

int mbpls_hip_standardize_sweep_masked(const mbpls_mat* src, mbpls_mat* dst, int mode, const double* d_rowmask,
                                       double n_eff, double* d_mean, double* d_scale, double* d_ssq_x,
                                       const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g) {
  MB_ARG(src && dst, "NULL argument");
  MB_ARG(!d_rowmask || n_eff >= 1.0, "n_eff must be the number of included samples");
  MB_ARG(src->n == dst->n && src->p == dst->p && src->ld == dst->ld, "src/dst shape mismatch");
  MB_ARG(mode == MBPLS_SCALE_FIT || mode == MBPLS_SCALE_APPLY || mode == MBPLS_SCALE_NONE, "bad mode %d", mode);
  MB_ARG(mode == MBPLS_SCALE_NONE || (d_mean && d_scale), "mean/scale buffers required");
  mbpls_ctx* ctx = dst->ctx;
  if (long_columns(ctx, dst->n))
    return long_standardize_sweep(src, dst, mode, d_rowmask, n_eff, d_mean, d_scale, d_ssq_x, d_v, alpha, d_g, d_t,
                                  d_ssq_g);
  PassArgs a{};
  MB_TRY(partial_buffers(ctx, dst, &a.part_t, &a.part_s));
  a.src = src->data; a.dst = dst->data; a.ld = dst->ld; a.n = dst->n;
  a.c0 = dst->fine.d_c0; a.c1 = dst->fine.d_c1;
  a.mode = mode; a.mean = d_mean; a.scale = d_scale;
  a.rowmask = d_rowmask; a.n_eff = n_eff;
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
