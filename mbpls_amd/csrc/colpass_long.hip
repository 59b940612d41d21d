// Column passes for LONG columns: n beyond what a workgroup can hold in registers (colpass.hip covers
// n <= 65536, its register-resident configurations spill above 8192).  Tall data (n >= p: many samples,
// few features -- the n >= p branches of UNIPALS / KERNEL, mbpls.py:388-479, :580-692) lands here.
//
// A column no longer fits a workgroup, so a quantity that needs the WHOLE column (a dot product, the column
// statistics) and a quantity that needs ALL columns of a row (the block scores t_b = X_b g_b) cannot come out of the
// same visit of an element.  Every pass is therefore two streaming phases around a p-sized finishing step:
//
//   phase A (one read, columns reduced over row slabs)
//       statistics : k_slab_stats   -- one WAVE per 1024-row slab and column group, shifted one-pass sums
//                                      S1 = sum (x - K), S2 = sum (x - K)^2, Sv = sum (x - K) v  (K: the column's first
//                                      sample), per-slab partials, fixed-order sum over the slabs (k_slab_colreduce)
//       dot products: the matrix-core X^T V kernel (xtv_mfma.hip) with V = [ts v]
//   finish (p-sized) : k_fin_std / k_fin_lazy -- mean, scale, weights g_j, loadings p_j, the lazy-deflation corrections,
//                      per-chunk sums of squares (second stage: reduce_chunks of colpass.hip, fixed order)
//   phase B (one read, or read + write when the pass stores X; rows are the parallel dimension)
//       k_rowpass   -- t_b += w_j x_j for the rows of the workgroup, fused with the store of the standardised or the
//                      deflated column when there is one.
//
// Transfers of 8np bytes:              explicit            never-writes-X (implicit standardisation, lazy deflation)
//   standardise (+ first sweep)        read, read+write    read, read
//   loadings + deflation + next sweep  read, read+write    read, read
//   sweep                              read, read          read, read
// (the first version streamed every column two or three times through one workgroup: 5 transfers for the first two.)
// The never-writes-X entry points take the same arguments as the register-resident ones, device-resident pass
// parameters (ctx->dpar) included, so a one-response fit on tall data is one device queue as well (engine.py).
//
// Bytes (algorithmic, as the register-resident passes): SWEEP 8np, STD / DEFL 16np (8np when nothing is stored).
#include <algorithm>

#include "colpass_util.h"

namespace mbpls {

void launch_reduce_xw(mbpls_ctx* ctx, const mbpls_mat* m, const double* part, int CB, int cb, int per_block,
                      double* d_out, int64_t ldo, int c_total, int k0);

constexpr int SLAB = 1024;   // rows per wave in k_slab_stats (16 per lane)

__device__ __forceinline__ bool pass_skipped(const double* dpar) { return dpar && dpar[2] != 0.0; }

// ------------------------------------------------------------------------------------------- phase A: statistics
struct SlabArgs {
  const double* src;
  int64_t ld, n, p;
  const double* v;         // optional: fused first sweep
  const double* rowmask;   // optional: excluded samples (0) take no part
  const double* shift;     // per-column shift K (MBPLS_SCALE_APPLY: the given means), or NULL
  int shift_first;         // MBPLS_SCALE_FIT: K = first sample of the column
  int cols_per_group, nslab;
  double* part;            // [p][nslab][4]: S1, S2, Sv, -
  const double* dpar;
};

template <int D>
__global__ __launch_bounds__(256) void k_slab_stats(SlabArgs a) {
  if (pass_skipped(a.dpar)) return;
  const int lane = threadIdx.x & 63;
  const int slab = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (slab >= a.nslab) return;                   // whole waves leave: no workgroup barrier below
  const int64_t n = a.n, ld = a.ld;
  const int64_t rb = (int64_t)slab * SLAB + 2 * lane;
  const int64_t j0 = (int64_t)blockIdx.y * a.cols_per_group;
  const int64_t j1 = (j0 + a.cols_per_group < a.p) ? j0 + a.cols_per_group : a.p;
  double v[16];
  unsigned keep = 0u;
  int64_t off[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t r = rb + 128 * k;
    off[k] = (r < ld) ? r : 0;                   // out-of-allocation pairs re-read pair 0 (masked by `keep`)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bool m = r + h < n;
      if (m && a.rowmask) m = a.rowmask[r + h] != 0.0;
      v[2 * k + h] = (m && a.v) ? a.v[r + h] : 0.0;
      keep |= (m ? 1u : 0u) << (2 * k + h);
    }
  }
  double x[D][16], k0[D];
  auto load = [&](double(&xc)[16], double& kc, int64_t j) {
    const double* col = a.src + j * ld;          // wave-uniform
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const v2d_t q = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(col + off[k]));
      xc[2 * k] = q.x;
      xc[2 * k + 1] = q.y;
    }
    kc = a.shift_first ? col[0] : (a.shift ? a.shift[j] : 0.0);
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (j0 + d < j1) load(x[d], k0[d], j0 + d);
  for (int64_t j = j0; j < j1; j += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int64_t jj = j + d;
      if (jj >= j1) break;
      const double K0 = uniform_f64(k0[d]);
      double s1 = 0.0, s2 = 0.0, sv = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double e = ((keep >> i) & 1u) ? x[d][i] - K0 : 0.0;
        s1 += e;
        s2 = fma(e, e, s2);
        sv = fma(e, v[i], sv);
      }
      if (jj + D < j1) load(x[d], k0[d], jj + D);   // refill before the reduction: the loads fly during it
      s1 = wave_sum_last_row(s1);
      s2 = wave_sum_last_row(s2);
      sv = wave_sum_last_row(sv);
      if (lane == 63) {
        double2* o = reinterpret_cast<double2*>(a.part + ((int64_t)jj * a.nslab + slab) * 4);
        o[0] = make_double2(s1, s2);
        o[1] = make_double2(sv, 0.0);
      }
    }
  }
}

// S[j][0..2] = sum over the slabs (fixed order), one wave per column
__global__ __launch_bounds__(256) void k_slab_colreduce(int64_t p, int nslab, const double* __restrict__ part,
                                                        double* __restrict__ S, const double* __restrict__ dpar) {
  if (pass_skipped(dpar)) return;
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= p) return;
  double s1 = 0.0, s2 = 0.0, sv = 0.0;
  for (int s = lane; s < nslab; s += 64) {
    const double2* q = reinterpret_cast<const double2*>(part + ((int64_t)j * nslab + s) * 4);
    const double2 a = q[0], b = q[1];
    s1 += a.x;
    s2 += a.y;
    sv += b.x;
  }
  s1 = wave_sum_last_row(s1);
  s2 = wave_sum_last_row(s2);
  sv = wave_sum_last_row(sv);
  if (lane == 63) {
    double2* o = reinterpret_cast<double2*>(S + j * 4);
    o[0] = make_double2(s1, s2);
    o[1] = make_double2(sv, 0.0);
  }
}

// sum_i a_i (b_i) over the rows the mask keeps (<ts, v>, the sum of v): up to 64 workgroups leave partials, the last
// launch adds them in order
constexpr int VD_WG = 64;
__global__ __launch_bounds__(1024) void k_vec_dot(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                  const double* __restrict__ mask, double* __restrict__ part,
                                                  const double* __restrict__ dpar) {
  __shared__ double red[2 * 64];
  int buf = 0;
  if (pass_skipped(dpar)) return;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  int u = 0;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024, u = (u + 1) & 3) {
    double x = a[i];
    if (b) x *= b[i];
    if (mask && mask[i] == 0.0) x = 0.0;
    s[u] += x;
  }
  double t[1] = {(s[0] + s[1]) + (s[2] + s[3])};
  block_sum<1>(t, red, buf);
  if (threadIdx.x == 0) part[blockIdx.x] = t[0];
}
__global__ __launch_bounds__(64) void k_vec_dot_final(int g, const double* __restrict__ part, double* __restrict__ out,
                                                      const double* __restrict__ dpar) {
  if (pass_skipped(dpar)) return;
  const double s = wave_sum((int)threadIdx.x < g ? part[threadIdx.x] : 0.0);
  if (threadIdx.x == 0) out[0] = s;
}

// ------------------------------------------------------------------------------------------- finishing steps
// After the statistics phase.  One workgroup per (fine) chunk of columns, one thread per column.
struct FinStdArgs {
  const int64_t* c0;
  const int64_t* c1;
  const double* S;        // [p][4]
  const double* svsum;    // device scalar: sum of v over the kept rows (NULL without a fused sweep)
  int mode;
  double dn, alpha;
  const double* dpar;
  int implicit;           // 1: nothing is stored, the row pass runs on raw columns (weights c_j = g~_j / scale_j^2)
  const double* src;      // to recover the shift K of MBPLS_SCALE_FIT
  int64_t ld;
  double* mean;
  double* scale;
  double* var_out;
  double* rs;
  double* r2;
  double* g;              // true weights g_j (optional)
  double* wrow;           // coefficient of column j in the row pass
  double* part_s;         // [nchunks][4]: ssq_g, -, ssq_x, amp
};

__global__ __launch_bounds__(256) void k_fin_std(FinStdArgs a) {
  __shared__ double red[2 * 64];
  __shared__ double s_mx[4];
  int buf = 0;
  if (pass_skipped(a.dpar)) return;
  const double alpha = a.dpar ? a.dpar[0] : a.alpha;
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const double dn = a.dn;
  const double svsum = a.svsum ? a.svsum[0] : 0.0;
  double acc[2] = {0.0, 0.0};   // ssq_g, ssq_x
  double amp = 0.0;
  for (int64_t j = a.c0[chunk] + tid; j < a.c1[chunk]; j += 256) {
    const double S1 = a.S[j * 4], S2 = a.S[j * 4 + 1], Sv = a.S[j * 4 + 2];
    double mean = 0.0, scale = 1.0, r2 = 1.0, rs = 1.0, ssqx, dotc;
    if (a.mode == MBPLS_SCALE_FIT) {
      const double K0 = a.src[j * a.ld];
      const double m1 = S1 / dn;
      mean = K0 + m1;
      const double var = fmax((S2 - S1 * m1) / dn, 0.0);
      const double eps = 2.220446049250313e-16;
      const double nme = dn * mean * eps;
      const bool constant = var <= dn * eps * var + nme * nme;
      scale = constant ? 1.0 : sqrt(var);
      r2 = constant ? 0.0 : 1.0 / var;
      rs = constant ? 0.0 : 1.0 / scale;
      ssqx = (constant ? 0.0 : dn * var * r2) + (S2 - S2);   // (non-finite samples poison the sum: colpass.hip)
      dotc = constant ? 0.0 : Sv - m1 * svsum;          // <x - mean, v> over the kept rows
      a.mean[j] = mean;
      a.scale[j] = scale;
      if (a.var_out) a.var_out[j] = constant ? 0.0 : var;
      amp = fmax(amp, mean * mean * r2);
    } else if (a.mode == MBPLS_SCALE_APPLY) {
      mean = a.mean[j];
      scale = a.scale[j];
      rs = 1.0 / scale;
      r2 = rs * rs;
      ssqx = S2 * r2;                                    // the shift was the given mean
      dotc = Sv;
    } else {
      ssqx = S2;                                         // no shift
      dotc = Sv;
    }
    if (a.rs) a.rs[j] = rs;
    if (a.r2) a.r2[j] = r2;
    const double gt = alpha * dotc;                      // raw-space weight g~_j
    const double gj = gt * rs;                           // g_j = alpha <z_j, v>
    if (a.g) a.g[j] = gj;
    if (a.implicit) {
      const double cj = gt * r2;
      a.wrow[j] = cj;
      acc[0] = fma(gt, cj, acc[0]);
    } else {
      a.wrow[j] = gj;
      acc[0] = fma(gj, gj, acc[0]);
    }
    acc[1] += ssqx;
  }
  block_sum<2>(acc, red, buf);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amp = fmax(amp, __shfl_xor(amp, off, 64));
  if ((tid & 63) == 0) s_mx[tid >> 6] = amp;
  __syncthreads();
  if (tid == 0) {
    double* ps = a.part_s + (int64_t)chunk * 4;
    ps[0] = acc[0];
    ps[1] = 0.0;
    ps[2] = acc[1];
    ps[3] = fmax(fmax(s_mx[0], s_mx[1]), fmax(s_mx[2], s_mx[3]));
  }
}

// After the dot-product phase: loadings, weights, lazy-deflation corrections (the algebra of colpass.hip's DEFL / SWEEP
// branches on whole-column dot products).  Explicit deflation is kprev = 0, r2 = NULL.
struct FinLazyArgs {
  const int64_t* c0;
  const int64_t* c1;
  const double* da;      // <x_j, ts>  (NULL: sweep only)
  const double* db;      // <x_j, v>   (NULL: loadings only)
  const double* Pm;      // loadings of the components deflated so far (p x kprev, leading dimension ldp)
  int64_t ldp;
  int kprev;
  const double* cu;      // <ts_i, v>
  const double* ce;      // <ts_i, ts>
  const double* r2;      // 1 / scale_j^2 (NULL: 1)
  const double* tsv;     // device scalar <ts, v>
  int deflate;
  double alpha, inv_tt;
  const double* dpar;
  double* pvec;
  double* g;
  double* wrow;
  double* part_s;        // [nchunks][4]: ssq_g, ssq_p, -, -
  double* part_d;        // [nchunks][KD]
};

__global__ __launch_bounds__(256) void k_fin_lazy(FinLazyArgs a) {
  __shared__ double red[2 * 64];
  __shared__ double s_cu[KD], s_ce[KD];
  int buf = 0;
  if (pass_skipped(a.dpar)) return;
  const double alpha = a.dpar ? a.dpar[0] : a.alpha;
  const double inv_tt = a.dpar ? a.dpar[1] : a.inv_tt;
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int kprev = a.kprev;
  if (tid < kprev) {
    s_cu[tid] = (a.cu && a.db) ? a.cu[tid] : 0.0;
    s_ce[tid] = (a.ce && a.da) ? a.ce[tid] : 0.0;
  }
  __syncthreads();
  const double tsv = (a.da && a.db && a.deflate) ? a.tsv[0] : 0.0;
  const int64_t j0 = a.c0[chunk], j1 = a.c1[chunk];
  double acc[2] = {0.0, 0.0};   // ssq_g, ssq_p
  for (int64_t j = j0 + tid; j < j1; j += 256) {
    double ce = 0.0, cu = 0.0;
    for (int i = 0; i < kprev; ++i) {
      const double pji = a.Pm[(int64_t)i * a.ldp + j];
      ce = fma(pji, s_ce[i], ce);
      cu = fma(pji, s_cu[i], cu);
    }
    const double r2 = a.r2 ? a.r2[j] : 1.0;
    double pj = 0.0;
    if (a.da) {
      pj = (a.da[j] - ce) * inv_tt;
      if (a.pvec) a.pvec[j] = pj;
      acc[1] = fma(pj * pj, r2, acc[1]);
    }
    if (a.db) {
      const double bj = a.db[j] - cu;
      const double gj = alpha * ((a.da && a.deflate) ? fma(-pj, tsv, bj) : bj);
      const double cj = gj * r2;
      if (a.g) a.g[j] = gj;
      a.wrow[j] = cj;
      acc[0] = fma(gj, cj, acc[0]);
    }
  }
  block_sum<2>(acc, red, buf);
  if (tid == 0) {
    double* ps = a.part_s + (int64_t)chunk * 4;
    ps[0] = acc[0];
    ps[1] = acc[1];
    ps[2] = 0.0;
    ps[3] = 0.0;
  }
  if (a.db && a.part_d) {
    // d_i = sum_j c_j P[j,i] (i < kprev), d_kprev = sum_j c_j p_j: every thread re-reads the c_j it has just written
    const int nd = kprev + ((a.da && a.deflate) ? 1 : 0);
    for (int i = 0; i < nd; ++i) {
      const double* col = (i < kprev) ? a.Pm + (int64_t)i * a.ldp : a.pvec;
      double s[1] = {0.0};
      for (int64_t j = j0 + tid; j < j1; j += 256) s[0] = fma(a.wrow[j], col[j], s[0]);
      block_sum<1>(s, red, buf);
      if (tid == 0) a.part_d[(int64_t)chunk * KD + i] = s[0];
    }
  }
}

__global__ __launch_bounds__(256) void k_fill(int64_t p, double value, double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < p) out[j] = value;
}

// ------------------------------------------------------------------------------------------- phase B: row pass
struct RowArgs {
  const double* src;
  double* dst;
  int64_t ld, n;
  const int64_t* c0;      // coarse chunks (row-slab kernels)
  const int64_t* c1;
  const double* w;        // coefficient of column j in the accumulation
  const double* mu;       // per-column mean
  const double* sc;       // per-column scale
  const double* pv;       // per-column loading (deflation)
  const double* ts;       // per-row score (deflation)
  const double* rowmask;
  double* part;           // [nchunks][n]
  const double* dpar;
};

enum { RP_ACC = 0, RP_ACC_CENTRED = 1, RP_STD_STORE = 2, RP_DEFL_STORE = 3, RP_COPY_STORE = 4 };

// grid (chunks, ceil(n/512)), 256 threads, two rows per thread: one 16-byte load per column and thread.
template <int MODE, bool ACC>
__global__ __launch_bounds__(256) void k_rowpass(RowArgs a) {
  if (pass_skipped(a.dpar)) return;
  const int chunk = blockIdx.x;
  const int64_t n = a.n, ld = a.ld;
  const int64_t r0 = (int64_t)blockIdx.y * 512 + 2 * threadIdx.x;
  const int64_t j0 = a.c0[chunk], j1 = a.c1[chunk];
  const bool in0 = r0 < n, in1 = r0 + 1 < n, inld = r0 < ld;
  bool k0 = in0, k1 = in1;
  if ((MODE == RP_STD_STORE || MODE == RP_COPY_STORE) && a.rowmask) {
    k0 = in0 && a.rowmask[r0] != 0.0;
    k1 = in1 && a.rowmask[r0 + 1] != 0.0;
  }
  double t0 = 0.0, t1 = 0.0;
  if constexpr (MODE == RP_DEFL_STORE) {
    t0 = in0 ? a.ts[r0] : 0.0;
    t1 = in1 ? a.ts[r0 + 1] : 0.0;
  }
  double acc0 = 0.0, acc1 = 0.0;
  const double* col = a.src + j0 * ld + r0;
  double* out = (MODE >= RP_STD_STORE) ? a.dst + j0 * ld + r0 : nullptr;
#pragma unroll 4
  for (int64_t j = j0; j < j1; ++j, col += ld) {
    v2d_t q = {0.0, 0.0};
    if (inld) q = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(col));
    double x0 = in0 ? q.x : 0.0, x1 = in1 ? q.y : 0.0;
    if constexpr (MODE == RP_ACC_CENTRED) {
      const double mu = a.mu[j];                     // workgroup-uniform -> scalar load
      x0 = in0 ? x0 - mu : 0.0;
      x1 = in1 ? x1 - mu : 0.0;
    } else if constexpr (MODE == RP_STD_STORE) {
      const double mu = a.mu[j], sc = a.sc[j];
      x0 = k0 ? (x0 - mu) / sc : 0.0;
      x1 = k1 ? (x1 - mu) / sc : 0.0;
    } else if constexpr (MODE == RP_COPY_STORE) {
      x0 = k0 ? x0 : 0.0;
      x1 = k1 ? x1 : 0.0;
    } else if constexpr (MODE == RP_DEFL_STORE) {
      const double pj = a.pv[j];
      x0 = fma(-pj, t0, x0);
      x1 = fma(-pj, t1, x1);
    }
    if constexpr (MODE >= RP_STD_STORE) {
      if (in1) {
        v2d_t o;
        o.x = x0;
        o.y = x1;
        __builtin_nontemporal_store(o, reinterpret_cast<v2d_t*>(out));
      } else if (in0) {
        out[0] = x0;
      }
      out += ld;
    }
    if constexpr (ACC) {
      const double wj = a.w[j];
      acc0 = fma(wj, x0, acc0);
      acc1 = fma(wj, x1, acc1);
    }
  }
  if constexpr (ACC) {
    double* o = a.part + (int64_t)chunk * n;
    if (in0) o[r0] = acc0;
    if (in1) o[r0 + 1] = acc1;
  }
}

bool long_columns(const mbpls_ctx* ctx, int64_t n) { return n > ctx->long_columns_from; }

// ------------------------------------------------------------------------------------------- host side
// Scratch of a composite pass lives in ctx->stage (xtv_mfma and the row-pass partials use ctx->ws).
struct LongScratch {
  double *V2, *dots, *wrow, *ones, *gtmp, *scal, *part_s, *part_d, *S, *slabpart;
  int nslab;
};

static int long_scratch(mbpls_ctx* ctx, const mbpls_mat* m, bool stats, LongScratch* s) {
  const size_t n = (size_t)m->n, p = (size_t)m->p, nc = (size_t)m->fine.nchunks;
  s->nslab = (int)((m->n + SLAB - 1) / SLAB);
  auto up = [](size_t v) { return (v + 31) / 32 * 32; };   // 256-byte granules
  size_t o = 0;
  const size_t oV2 = o; o += up(2 * n);
  const size_t odots = o; o += up(2 * p);
  const size_t owrow = o; o += up(p);
  const size_t oones = o; o += up(p);
  const size_t ogtmp = o; o += up(p);
  const size_t oscal = o; o += up(8 + VD_WG);
  const size_t ops = o; o += up(nc * 4);
  const size_t opd = o; o += up(nc * KD);
  const size_t oS = o; o += up(p * 4);
  const size_t oslab = o; o += stats ? up(p * (size_t)s->nslab * 4) : 0;
  MB_TRY(ensure_stage(ctx, sizeof(double) * o));
  double* b = (double*)ctx->stage;
  s->V2 = b + oV2; s->dots = b + odots; s->wrow = b + owrow; s->ones = b + oones; s->gtmp = b + ogtmp;
  s->scal = b + oscal; s->part_s = b + ops; s->part_d = b + opd; s->S = b + oS; s->slabpart = b + oslab;
  return MBPLS_OK;
}

// s.scal[0] = sum a_i b_i over the kept rows
static int long_vec_dot(mbpls_ctx* ctx, const LongScratch& s, int64_t n, const double* a, const double* b,
                        const double* mask) {
  const int g = (int)std::max<int64_t>(1, std::min<int64_t>(VD_WG, (n + 16383) / 16384));
  hipLaunchKernelGGL(k_vec_dot, dim3((unsigned)g), dim3(1024), 0, ctx->stream, n, a, b, mask, s.scal + 8, ctx->dpar);
  MB_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_vec_dot_final, dim3(1), dim3(64), 0, ctx->stream, g, (const double*)(s.scal + 8), s.scal,
                     ctx->dpar);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

// phase B: launches the row pass in `mode`; with d_t the per-block scores t_b = sum_j w_j x'_j land in d_t [nblocks][n]
static int long_rowpass(mbpls_ctx* ctx, const mbpls_mat* src, mbpls_mat* dst, int mode, const double* w,
                        const double* mu, const double* sc, const double* pv, const double* ts, const double* rowmask,
                        double* d_t) {
  const mbpls_mat* m = src;
  const ChunkTable& ct = m->coarse;
  RowArgs a{};
  a.src = src->data; a.dst = dst ? dst->data : nullptr; a.ld = m->ld; a.n = m->n;
  a.c0 = ct.d_c0; a.c1 = ct.d_c1;
  a.w = w; a.mu = mu; a.sc = sc; a.pv = pv; a.ts = ts; a.rowmask = rowmask; a.dpar = ctx->dpar;
  const bool acc = d_t != nullptr;
  // one chunk per block (tall data): chunk index == block index, the partial IS the result
  const bool direct = ct.nchunks == m->nblocks;
  if (acc) {
    if (direct) a.part = d_t;
    else {
      MB_TRY(ensure_ws(ctx, sizeof(double) * (size_t)ct.nchunks * (size_t)m->n));
      a.part = (double*)ctx->ws;
    }
  }
  dim3 grid((unsigned)ct.nchunks, (unsigned)((m->n + 511) / 512)), block(256);
#define MB_RP(M_)                                                                       \
  do {                                                                                  \
    if (acc) hipLaunchKernelGGL((k_rowpass<M_, true>), grid, block, 0, ctx->stream, a); \
    else hipLaunchKernelGGL((k_rowpass<M_, false>), grid, block, 0, ctx->stream, a);    \
  } while (0)
  switch (mode) {
    case RP_ACC: MB_RP(RP_ACC); break;
    case RP_ACC_CENTRED: MB_RP(RP_ACC_CENTRED); break;
    case RP_STD_STORE: MB_RP(RP_STD_STORE); break;
    case RP_DEFL_STORE: MB_RP(RP_DEFL_STORE); break;
    default: MB_RP(RP_COPY_STORE); break;
  }
#undef MB_RP
  MB_HIP(hipGetLastError());
  if (acc && !direct) {
    launch_reduce_xw(ctx, m, a.part, 1, 1, 1, d_t, m->n, 1, 0);
    MB_HIP(hipGetLastError());
  }
  return MBPLS_OK;
}

// phase A + finish of every standardising pass.  implicit = nothing stored (the statistics-only first pass).
static int long_stats(mbpls_ctx* ctx, const mbpls_mat* src, const LongScratch& s, int mode, const double* d_rowmask,
                      double n_eff, double* d_mean, double* d_scale, double* d_rs, double* d_r2, const double* d_v,
                      double alpha, double* d_g, int implicit) {
  const mbpls_mat* m = src;
  if (d_v) MB_TRY(long_vec_dot(ctx, s, m->n, d_v, nullptr, d_rowmask));
  SlabArgs a{};
  a.src = m->data; a.ld = m->ld; a.n = m->n; a.p = m->p;
  a.v = d_v; a.rowmask = d_rowmask;
  a.shift = (mode == MBPLS_SCALE_APPLY) ? d_mean : nullptr;
  a.shift_first = mode == MBPLS_SCALE_FIT;
  a.nslab = s.nslab; a.part = s.slabpart; a.dpar = ctx->dpar;
  const int64_t want_groups = std::max<int64_t>(1, ((int64_t)ctx->cus * 32 + s.nslab - 1) / s.nslab);
  const int64_t groups0 = std::min<int64_t>({want_groups, m->p, (int64_t)65535});
  a.cols_per_group = (int)((m->p + groups0 - 1) / groups0);
  const unsigned groups = (unsigned)((m->p + a.cols_per_group - 1) / a.cols_per_group);
  hipLaunchKernelGGL((k_slab_stats<2>), dim3((unsigned)((s.nslab + 3) / 4), groups), dim3(256), 0, ctx->stream, a);
  MB_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_slab_colreduce, dim3((unsigned)((m->p + 3) / 4)), dim3(256), 0, ctx->stream, m->p, s.nslab,
                     (const double*)s.slabpart, s.S, ctx->dpar);
  MB_HIP(hipGetLastError());
  FinStdArgs f{};
  f.c0 = m->fine.d_c0; f.c1 = m->fine.d_c1; f.S = s.S; f.svsum = d_v ? s.scal : nullptr;
  f.mode = mode; f.dn = d_rowmask ? n_eff : (double)m->n; f.alpha = alpha; f.dpar = ctx->dpar;
  f.implicit = implicit; f.src = m->data; f.ld = m->ld;
  f.mean = d_mean; f.scale = d_scale; f.var_out = ctx->var_out; f.rs = d_rs; f.r2 = d_r2;
  f.g = d_g; f.wrow = s.wrow; f.part_s = s.part_s;
  hipLaunchKernelGGL(k_fin_std, dim3((unsigned)m->fine.nchunks), dim3(256), 0, ctx->stream, f);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

int long_sweep(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g) {
  mbpls_ctx* ctx = m->ctx;
  LongScratch s;
  MB_TRY(long_scratch(ctx, m, false, &s));
  double* g = d_g ? d_g : s.gtmp;
  MB_TRY(mbpls_hip_xt_v(m, d_v, m->n, 1, alpha, g, m->p, d_ssq_g));   // matrix-core path, any n
  if (!d_t) return MBPLS_OK;
  ProfScope prof(ctx, MBPLS_K_X_W, 8.0 * (double)m->n * (double)m->p);
  return long_rowpass(ctx, m, nullptr, RP_ACC, g, nullptr, nullptr, nullptr, nullptr, nullptr, d_t);
}

// dot products of every column with ts (and v): da = s.dots, db = s.dots + p (or s.dots when there is no ts)
static int long_dots(mbpls_ctx* ctx, const mbpls_mat* m, const LongScratch& s, const double* d_ts, const double* d_v,
                     const double* d_mean) {
  const size_t nb = sizeof(double) * (size_t)m->n;
  int c = 0;
  if (d_ts) {
    MB_HIP(hipMemcpyAsync(s.V2 + (size_t)m->n * c, d_ts, nb, hipMemcpyDeviceToDevice, ctx->stream));
    ++c;
  }
  if (d_v) {
    MB_HIP(hipMemcpyAsync(s.V2 + (size_t)m->n * c, d_v, nb, hipMemcpyDeviceToDevice, ctx->stream));
    ++c;
  }
  if (d_mean) {   // offset data: the dot products run on x - mean (scale 1)
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((m->p + 255) / 256)), dim3(256), 0, ctx->stream, m->p, 1.0, s.ones);
    MB_HIP(hipGetLastError());
  }
  return xtv_mfma(ctx, m, s.V2, m->n, c, 1.0, s.dots, m->p, nullptr, d_mean, d_mean ? s.ones : nullptr);
}

// loadings (+ deflation) (+ the next sweep), explicit or lazy: the shared composite
static int long_loadings_sweep(mbpls_mat* m, const double* d_ts, double inv_tt, int deflate, int store, double* d_p,
                               double* d_ssq_p, const double* d_v, double alpha, double* d_g, double* d_t,
                               double* d_ssq_g, const double* d_P, int64_t ldp, int kprev, const double* d_cu,
                               const double* d_ce, double* d_d, const double* d_r2, const double* d_mean) {
  mbpls_ctx* ctx = m->ctx;
  LongScratch s;
  MB_TRY(long_scratch(ctx, m, false, &s));
  const bool writes = store != 0;
  ProfScope prof(ctx, d_ts ? (d_v ? MBPLS_K_DEFLATE_SWEEP : MBPLS_K_DEFLATE) : MBPLS_K_SWEEP,
                 8.0 * (double)m->n * (double)m->p * (writes ? 2.0 : 1.0));
  MB_TRY(long_dots(ctx, m, s, d_ts, d_v, d_mean));
  if (d_ts && d_v && deflate) MB_TRY(long_vec_dot(ctx, s, m->n, d_ts, d_v, nullptr));
  FinLazyArgs f{};
  f.c0 = m->fine.d_c0; f.c1 = m->fine.d_c1;
  f.da = d_ts ? s.dots : nullptr;
  f.db = d_v ? (d_ts ? s.dots + m->p : s.dots) : nullptr;
  f.Pm = d_P; f.ldp = ldp; f.kprev = kprev; f.cu = d_cu; f.ce = d_ce; f.r2 = d_r2; f.tsv = s.scal;
  f.deflate = deflate; f.alpha = alpha; f.inv_tt = inv_tt; f.dpar = ctx->dpar;
  // the loadings are needed by the storing row pass and by the lazy sums even when the caller does not want them
  f.pvec = d_ts ? (d_p ? d_p : s.gtmp) : nullptr;
  f.g = d_g; f.wrow = s.wrow; f.part_s = s.part_s; f.part_d = d_d ? s.part_d : nullptr;
  hipLaunchKernelGGL(k_fin_lazy, dim3((unsigned)m->fine.nchunks), dim3(256), 0, ctx->stream, f);
  MB_HIP(hipGetLastError());
  const int nd = d_d ? kprev + ((d_ts && d_v && deflate) ? 1 : 0) : 0;
  if ((d_ts && d_ssq_p) || (d_v && d_ssq_g) || nd)
    MB_TRY(reduce_chunks(ctx, m, nullptr, s.part_s, nullptr, d_v ? d_ssq_g : nullptr, d_ts ? d_ssq_p : nullptr, nullptr,
                         s.part_d, nd, nd ? d_d : nullptr));
  const bool acc = d_v && d_t;
  if (writes)
    return long_rowpass(ctx, m, m, RP_DEFL_STORE, s.wrow, nullptr, nullptr, f.pvec, d_ts, nullptr, acc ? d_t : nullptr);
  if (acc)
    return long_rowpass(ctx, m, nullptr, d_mean ? RP_ACC_CENTRED : RP_ACC, s.wrow, d_mean, nullptr, nullptr, nullptr,
                        nullptr, d_t);
  return MBPLS_OK;
}

int long_deflate_sweep(mbpls_mat* m, const double* d_ts, double inv_tt, int deflate, double* d_p, double* d_ssq_p,
                       const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g) {
  return long_loadings_sweep(m, d_ts, inv_tt, deflate, deflate, d_p, d_ssq_p, d_v, alpha, d_g, d_t, d_ssq_g, nullptr, 0,
                             0, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int long_sweep_lazy(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                    const double* d_P, int64_t ldp, int kprev, const double* d_cu, double* d_d, const double* d_r2,
                    const double* d_mean) {
  return long_loadings_sweep(const_cast<mbpls_mat*>(m), nullptr, 1.0, 0, 0, nullptr, nullptr, d_v, alpha, d_g, d_t,
                             d_ssq_g, d_P, ldp, kprev, d_cu, nullptr, kprev ? d_d : nullptr, d_r2, d_mean);
}

int long_loadings_sweep_lazy(const mbpls_mat* m, const double* d_ts, double inv_tt, double* d_p, double* d_ssq_p,
                             const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                             const double* d_P, int64_t ldp, int kprev, const double* d_cu, const double* d_ce,
                             double* d_d, const double* d_r2, const double* d_mean) {
  // deflate = 1 with a fused sweep: the sweep sees X_{k+1} (colpass.hip, mbpls_hip_loadings_sweep_lazy_scaled)
  return long_loadings_sweep(const_cast<mbpls_mat*>(m), d_ts, inv_tt, d_v ? 1 : 0, 0, d_p, d_ssq_p, d_v, alpha, d_g, d_t,
                             d_ssq_g, d_P, ldp, kprev, d_cu, d_ce, d_v ? d_d : nullptr, d_r2, d_mean);
}

int long_loadings_sweep_raw(const mbpls_mat* m, const double* d_ts, double inv_tt, double* d_p, double* d_ssq_p,
                            const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                            const double* d_r2, const double* d_mean) {
  return long_loadings_sweep(const_cast<mbpls_mat*>(m), d_ts, inv_tt, 0, 0, d_p, d_ssq_p, d_v, alpha, d_g, d_t, d_ssq_g,
                             nullptr, 0, 0, nullptr, nullptr, nullptr, d_r2, d_mean);
}

int long_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode, const double* d_rowmask, double n_eff,
                           double* d_mean, double* d_scale, double* d_ssq_x, const double* d_v, double alpha,
                           double* d_g, double* d_t, double* d_ssq_g) {
  mbpls_ctx* ctx = dst->ctx;
  LongScratch s;
  MB_TRY(long_scratch(ctx, src, true, &s));
  ProfScope prof(ctx, MBPLS_K_STANDARDIZE, 16.0 * (double)dst->n * (double)dst->p);
  MB_TRY(long_stats(ctx, src, s, mode, d_rowmask, n_eff, d_mean, d_scale, nullptr, nullptr, d_v, alpha, d_g, 0));
  if (d_ssq_x || (d_v && d_ssq_g))
    MB_TRY(reduce_chunks(ctx, src, nullptr, s.part_s, nullptr, d_v ? d_ssq_g : nullptr, nullptr, d_ssq_x));
  double* t = (d_v && d_t) ? d_t : nullptr;
  if (mode != MBPLS_SCALE_NONE)
    return long_rowpass(ctx, src, dst, RP_STD_STORE, s.wrow, d_mean, d_scale, nullptr, nullptr, d_rowmask, t);
  if (d_rowmask || src->data != dst->data)
    return long_rowpass(ctx, src, dst, RP_COPY_STORE, s.wrow, nullptr, nullptr, nullptr, nullptr, d_rowmask, t);
  return t ? long_rowpass(ctx, src, nullptr, RP_ACC, s.wrow, nullptr, nullptr, nullptr, nullptr, nullptr, t) : MBPLS_OK;
}

int long_standardize_stats_sweep(const mbpls_mat* src, int mode, double* d_mean, double* d_scale, double* d_rscale,
                                 double* d_r2, double* d_ssq_x, double* d_amp, const double* d_v, double alpha,
                                 double* d_g, double* d_t, double* d_ssq_g) {
  mbpls_ctx* ctx = src->ctx;
  LongScratch s;
  MB_TRY(long_scratch(ctx, src, true, &s));
  ProfScope prof(ctx, MBPLS_K_STANDARDIZE, 8.0 * (double)src->n * (double)src->p);
  MB_TRY(long_stats(ctx, src, s, mode, nullptr, 0.0, d_mean, d_scale, d_rscale, d_r2, d_v, alpha, d_g, 1));
  MB_TRY(reduce_chunks(ctx, src, nullptr, s.part_s, nullptr, d_ssq_g, nullptr, d_ssq_x, nullptr, 0, nullptr, d_amp));
  // raw columns, centred in registers once the means are known
  const bool fit = mode == MBPLS_SCALE_FIT;
  return long_rowpass(ctx, src, nullptr, fit ? RP_ACC_CENTRED : RP_ACC, s.wrow, fit ? d_mean : nullptr, nullptr, nullptr,
                      nullptr, nullptr, d_t);
}

}  // namespace mbpls

// one kernel of this translation unit, for mbpls_hip_warmup (its code object is loaded on first use otherwise)
namespace mbpls {
const void* anchor_colpass_long() { return (const void*)&k_slab_colreduce; }
}  // namespace mbpls
