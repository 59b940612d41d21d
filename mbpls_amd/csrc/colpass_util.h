// Column-pass building blocks shared by colpass.hip (dense) and colpass_nan.hip (missing values):
// row ownership of a thread, register-resident column loads / stores, launch-shape selection.
#pragma once
#include "common.h"

namespace mbpls {

// Row ownership of one thread: pair k covers rows 2(kT + tid), 2(kT + tid) + 1.  Byte offsets are 32-bit
// and identical for every column, so a column load is  global_load_dwordx4 v, v_off, s[col_base]  (SGPR base
// + VGPR offset): no 64-bit address arithmetic and no address registers per load in flight.
template <int R>
struct RowMap {
  unsigned off[R / 2];   // byte offset of the pair inside a column
  unsigned in_ld;        // bit k: pair k lies inside the column allocation (r0 < ld)
  unsigned in_n0, in_n1; // bit k: row r0 / r0+1 is a real sample (< n)
  // rowmask (optional, n doubles, 0 = sample excluded): excluded rows are treated exactly like rows
  // beyond n -- they load as 0 -- which is all a fit on a ROW SUBSET needs (cross-validation folds on a
  // matrix that stays resident in HBM).
  __device__ __forceinline__ void init(int64_t n, int64_t ld, int tid, int T, const double* rowmask = nullptr) {
    in_ld = in_n0 = in_n1 = 0u;
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      // 32-bit arithmetic on purpose: the offset must reach the load as zext(i32) for the SGPR-base form
      const unsigned r0 = 2u * ((unsigned)k * (unsigned)T + (unsigned)tid);
      const bool inside = (int64_t)r0 < ld;
      off[k] = inside ? r0 * 8u : 0u;   // out-of-allocation pairs re-read pair 0 (masked below)
      in_ld |= (inside ? 1u : 0u) << k;
      bool k0 = (int64_t)r0 < n, k1 = (int64_t)r0 + 1 < n;
      if (rowmask) {
        if (k0) k0 = rowmask[r0] != 0.0;
        if (k1) k1 = rowmask[r0 + 1] != 0.0;
      }
      in_n0 |= (k0 ? 1u : 0u) << k;
      in_n1 |= (k1 ? 1u : 0u) << k;
    }
  }
};

typedef double v2d_t __attribute__((ext_vector_type(2)));

// force a wave-uniform 64-bit value into SGPRs (values fetched through possibly-aliased pointers land in
// VGPRs, which would turn every column base address into per-lane 64-bit address arithmetic)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// same for a wave-uniform double (per-column statistics fetched through global pointers)
__device__ __forceinline__ double uniform_f64(double v) {
  return __longlong_as_double(uniform64(__double_as_longlong(v)));
}

// X is streamed exactly once per pass: non-temporal loads/stores keep it out of L2/MALL
// (measured on MI355X: read-only sweep 6.2 -> 6.8 TB/s, read+write pass 5.5 -> 5.7 TB/s).
// FULL: n == T*R == ld, every pair of every thread is a real sample -> no masks, and (more important) no
// select right behind the load that would force an s_waitcnt before the prefetched column is needed.
template <int R, bool FULL>
__device__ __forceinline__ void load_col(double (&x)[R], const double* __restrict__ col, const RowMap<R>& m) {
  const char* base = reinterpret_cast<const char*>(col);   // workgroup-uniform
#pragma unroll
  for (int k = 0; k < R / 2; ++k) {
    // unconditional (clamped offset): no exec-mask branch per load, masking happens on the value
    const v2d_t q = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(base + m.off[k]));
    if constexpr (FULL) {
      x[2 * k] = q.x;
      x[2 * k + 1] = q.y;
    } else {
      x[2 * k] = ((m.in_n0 >> k) & 1u) ? q.x : 0.0;
      x[2 * k + 1] = ((m.in_n1 >> k) & 1u) ? q.y : 0.0;
    }
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

template <int R, bool FULL>
__device__ __forceinline__ void store_col(const double (&x)[R], double* __restrict__ col, const RowMap<R>& m) {
  char* base = reinterpret_cast<char*>(col);
#pragma unroll
  for (int k = 0; k < R / 2; ++k) {
    if (FULL || ((m.in_ld >> k) & 1u)) {
      v2d_t q;
      q.x = x[2 * k];
      q.y = x[2 * k + 1];
      __builtin_nontemporal_store(q, reinterpret_cast<v2d_t*>(base + m.off[k]));
    }
  }
}

// ------------------------------------------------------------------------------- launch shape
struct Cfg {
  int R, TMAX;
};

// Choose rows-per-thread R and the workgroup size T = roundup(ceil(n/R), 64) <= TMAX.
static bool pick_cfg(const mbpls_ctx* ctx, int64_t n, Cfg* cfg, int* T, bool std_pass = false, bool has_r10 = false,
                     bool has_r20 = false) {
  // preference order: smallest register footprint that still covers n with <= TMAX threads
  // the last two are correctness fallbacks for very long columns (8192 < n <= 65536): the column no longer
  // fits the register budget of 1024 threads, the compiler spills to scratch -> slow but exact
  static const Cfg table[] = {{2, 256}, {4, 512}, {8, 512}, {8, 1024}, {16, 512}, {10, 512}, {20, 256}, {24, 256}, {4, 1024}, {32, 1024}, {64, 1024}};
  auto fits = [&](const Cfg& c, int* t) {
    int64_t need = (n + c.R - 1) / c.R;
    need = (need + 63) / 64 * 64;
    if (need > c.TMAX) return false;
    *t = (int)need;
    return true;
  };
  // Passes that reduce several values per column (standardise: statistics; loadings + sweep: two dot products) are
  // instruction-issue bound, not bandwidth bound: fewer, fatter threads win -- the per-column reduction overhead is
  // spread over twice the elements (measured at n=4096, R=16/T=256 vs R=8/T=512: storing standardise 5.46 vs
  // 4.77 TB/s, statistics-only 5.8 vs 7.9 ms, lazy loadings+sweep 4.73 vs 5.70 ms).  Single-reduction passes
  // (sweep, loadings only) keep R=8/T=512.
  // The same holds for the single-reduction passes away from the tuned n = 4096 = 8 x 512 shape (measured, default
  // table vs R=16: n=2048 sweep 4.99 vs 3.78 ms, n=6000 5.09 vs 4.30 ms; n=4096 loadings-only 4.64 vs 5.05 ms).
  // Rule: the largest R in {16, 8, 4, 2} that still gives a full wave of real threads.
  const bool tuned8 = !std_pass && n > 3584 && n <= 4096;
  // just above 4096 the single-reduction passes do better with ten rows per thread -- eight full waves -- than with sixteen
  // on five (measured, 8 GB matrices, R = 10 vs 16: n = 4500 sweep 5.80 vs 5.48 TB/s, loadings only 6.73 vs 6.04; n = 5000
  // 6.08 vs 5.78, 6.62 vs 6.30; from n = 6000 on there is nothing in it; the two-reduction passes lose 8 % with R = 10)
  if (has_r10 && ctx->rows_per_thread <= 0 && !std_pass && n > 4096 && n <= 5120) {   // (colpass.hip only)
    *cfg = Cfg{10, 512};
    return fits(*cfg, T);
  }
  // ... and the statistics pass up to n = 6144 with TWENTY (TWENTY-FOUR) rows per thread on four waves.  With sixteen rows
  // the several-reduction kernels hold 206-254 VGPRs (tools/kernel_resources.py): two waves per SIMD, eight per CU -- two
  // workgroups of four waves at n = 4096, but ONE of five at n = 5000 (one of six at 6000, one of eight at 8192), alone
  // on its CU with nothing to run while it reduces a column: the statistics pass falls from 6.8 TB/s at n = 4096 to 3.9
  // at 5000.  On 256 threads two workgroups share a CU: 20 rows (241-254 VGPRs, no scratch) 3.9 -> 6.3 TB/s at n = 5000,
  // 24 rows (108 bytes of scratch per lane) 4.3 -> 6.0 at n = 6000; 28 / 32 rows spill 280 / 430 bytes and LOSE (3.5 / 2.7
  // against 4.5 / 5.0 at n = 7000 / 8192), and so does the fused loadings + lazy deflation + sweep pass from twenty rows on
  // (300 bytes of scratch: 2.7-3.0 against 4.7-5.2 TB/s) -- profiles/r06_stats_pass_r20.txt.
  // Below 4096 the same kernels serve ragged n better than sixteen rows wherever they need fewer wave-instructions per
  // column (waves per workgroup x rows per thread: the pass is issue bound) or a wave count that divides the eight slots
  // of a CU: n = 1200 one wave of twenty rows instead of two of sixteen (5.0 -> 5.7 TB/s), n = 2500 two waves of twenty
  // instead of three of sixteen (4.9 -> 6.4), n = 3000 two of twenty-four (5.7 -> 6.0).
  if (has_r20 && ctx->rows_per_thread <= 0 && n > 1024 && n <= 6144) {
    int best_r = 0;
    double best = 0.0;
    for (int r : {16, 20, 24}) {
      const int64_t threads = ((n + r - 1) / r + 63) / 64 * 64;
      if (threads > (r == 16 ? 512 : 256)) continue;
      const int waves = (int)(threads / 64);
      if (r == 16 && n == threads * 16 && waves <= 4) {      // sixteen rows fill their waves exactly: mask-free, 6.3 TB/s
        best_r = 16;
        break;
      }
      double cost = (double)waves * r;
      if (waves != 1 && waves != 2 && waves != 4 && waves != 8) cost *= 1.15;   // does not divide the eight wave slots
      if (best_r == 0 || cost < best - 1e-9) {
        best_r = r;
        best = cost;
      }
    }
    if (best_r > 16) {
      *cfg = Cfg{best_r, 256};
      return fits(*cfg, T);
    }
  }
  if (ctx->rows_per_thread <= 0 && !tuned8) {
    static const Cfg fat[] = {{16, 512}, {8, 512}, {4, 512}, {2, 256}};
    for (const Cfg& c : fat)
      if ((n + c.R - 1) / c.R >= 64 && fits(c, T)) {
        *cfg = c;
        return true;
      }
  }
  if (ctx->rows_per_thread > 0) {
    for (const Cfg& c : table)
      if (c.R == ctx->rows_per_thread && (has_r10 || c.R != 10) && (has_r20 || c.TMAX != 256 || c.R < 20) && fits(c, T)) {
        *cfg = c;
        return true;
      }
  }
  for (const Cfg& c : table)
    if ((has_r10 || c.R != 10) && !(c.TMAX == 256 && c.R >= 20) && fits(c, T)) {
      *cfg = c;
      return true;
    }
  return false;
}

// colpass.hip: second stage, fixed-order sum of the chunk partials of each block (n-vectors into d_t,
// the per-chunk scalars [ssq_g, ssq_p, ssq_x, -] into the d_ssq_* outputs, lazy-deflation sums into d_d)
int reduce_chunks(mbpls_ctx* ctx, const mbpls_mat* m, const double* part_t, const double* part_s,
                  double* d_t, double* d_ssq_g, double* d_ssq_p, double* d_ssq_x,
                  const double* part_d = nullptr, int nd = 0, double* d_d = nullptr, double* d_amp = nullptr,
                  const ChunkTable* table = nullptr /* default: the matrix's fine table */);


// Arguments of a column pass (colpass.hip; colpass_sub.hip for few samples).
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
  double* part_d;      // [nchunks][KD]  sum_j c_j P[j,i] (i < kprev) and sum_j c_j p_j (i = kprev)
  // implicit standardisation (lazy passes only, template flag ISTD): X in HBM is the RAW matrix and
  // Z = (X - mean)/scale is applied algebraically, everything computed in RAW-column space.  Every vector v / ts
  // of a fit on standardised data sums to zero (Y is centred, scores of centred columns are centred; the caller
  // re-centres to rounding), so the column means drop out of the dot products, <x_j - mean_j 1, v> = <x_j, v>.
  // With raw-space weights g~_j = scale_j g_j and loadings p~_j = scale_j p_j (what the slots then hold):
  //   scale_j z_j^(k) = x_j - mean_j 1 - sum_i ts_i P~[j,i],   g~_j = alpha (<x_j, v> - sum_i P~[j,i] cu_i),
  //   p~_j = inv_tt (<x_j, ts> - sum_i P~[j,i] ce_i),
  //   sum_j g_j z_j = sum_j c_j x_j - const 1 - sum_i ts_i (sum_j c_j P~[j,i]),   c_j = g~_j / scale_j^2
  // so the kernel is the plain lazy pass plus ONE per-column scalar ir2_j = 1/scale_j^2 in the accumulations
  // (||g_b||^2 = sum g~_j c_j, ||p_b||^2 = sum p~_j^2 ir2_j); the constant is the mean of the accumulated vector
  // (scores are centred), removed by the caller, who also rescales finished g~ / p~ columns by 1/scale.
  const double* ir2;
  // ... and for OFFSET data (mu != NULL, template flag CEN) the dot products / accumulations run on x_j - mean_j 1
  // (one subtraction per element in registers): the rounding of a raw-space dot product is amplified by
  // |mean_j| / scale_j, enough to keep the NIPALS stop rule ||ts_old - ts|| <= 1e-14 (mbpls.py:841) out of reach
  // once that ratio is in the hundreds.  The passes are partly issue-bound (8 % / 13 % slower with the subtraction),
  // so it is a separate instantiation, chosen from the offset measure of the statistics pass -- which itself, not
  // knowing the mean yet, always shifts by the column's first sample.
  const double* mu;
  // device-resident parameters (ctx->dpar): when given, alpha = dpar[0], inv_tt = dpar[1] and the whole pass is a
  // no-op when dpar[2] != 0 (a speculatively queued fit that turned out to need more NIPALS iterations)
  const double* dpar;
  double* var_out;     // standardise passes, MBPLS_SCALE_FIT: per-column variance (optional; ctx->var_out)
  double* rs_out;      // standardise pass: 1 / scale_j   (optional)
  double* r2_out;      // standardise pass: 1 / scale_j^2 (optional)
  // raw-space passes (ir2 given), mbpls_hip_set_true_space_outputs: rs = 1/scale_j -> g holds the TRUE-space weights
  // g~_j rs_j, and pvec_true (optional) the true-space loadings p~_j rs_j beside the raw-space ones in pvec
  const double* rs;
  double* pvec_true;
};

// kernels that do not scale their outputs themselves: the same with a p-sized kernel behind the pass (colpass.hip)
int true_space_cols(mbpls_ctx* ctx, int64_t p, double* g, const double* p_raw);

// colpass_sub.hip: the same passes for n <= SUB_MAX_N samples, several columns per wave (KD is used there: keep it above)
constexpr int SUB_MAX_N = 1024;
int launch_pass_sub(mbpls_ctx* ctx, const ChunkTable& ct, const PassArgs& a, bool STD, bool DEFL, bool SWEEP);

// Arguments of a missing-value column pass (colpass_nan.hip; colpass_sub.hip for few samples).
struct NanArgs {
  const double* rowmask;   // standardise pass only: samples excluded from the fit (cross-validation folds)
  double* var_out;
  const double* src;
  double* dst;
  int64_t ld, n;
  const int64_t* c0;
  const int64_t* c1;
  const double* v;    // sweep vector (u)
  double* g;          // [p] un-normalised weights
  const double* ts;   // superscore
  int deflate;
  double* pvec;       // [p] loadings
  int mode;
  double* mean;
  double* scale;
  double* count;      // [p] present samples per column (standardise pass, optional)
  double* part_t;     // [nchunks][n]  A
  double* part_d;     // [nchunks][n]  D
  double* part_s;     // [nchunks][4]  ssq_g, ssq_p, ssq_x, -
};
int launch_nan_sub(mbpls_ctx* ctx, const ChunkTable& ct, const NanArgs& a, bool STD, bool DEFL, bool SWEEP);

// colpass_nan_long.hip: the missing-value passes for long columns as two streaming phases (same arguments as the
// mbpls_hip_nan_* entry points)
int nan_long_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode, const double* d_rowmask, double* d_mean,
                               double* d_scale, double* d_count, double* d_ssq_x, const double* d_v, double* d_g,
                               double* d_t, double* d_d, double* d_ssq_g);
int nan_long_sweep(const mbpls_mat* m, const double* d_v, double* d_g, double* d_t, double* d_d, double* d_ssq_g);
int nan_long_deflate_sweep(mbpls_mat* m, const double* d_ts, int deflate, double* d_p, double* d_ssq_p,
                           const double* d_v, double* d_g, double* d_t, double* d_d, double* d_ssq_g);

// lazy deflation: at most KD - 1 previous components are corrected for inside a pass
constexpr int KD = 40;

// colpass_long.hip: the same passes for columns too long to hold in registers (n > ctx->long_columns_from)
bool long_columns(const mbpls_ctx* ctx, int64_t n);
int long_sweep(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g);
int long_deflate_sweep(mbpls_mat* m, const double* d_ts, double inv_tt, int deflate, double* d_p, double* d_ssq_p,
                       const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g);
int long_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode, const double* d_rowmask, double n_eff,
                           double* d_mean, double* d_scale, double* d_ssq_x, const double* d_v, double alpha,
                           double* d_g, double* d_t, double* d_ssq_g);
// ... and their never-writes-X forms (same arguments as the mbpls_hip_*_lazy_scaled / _stats_sweep entry points)
int long_sweep_lazy(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                    const double* d_P, int64_t ldp, int kprev, const double* d_cu, double* d_d, const double* d_r2,
                    const double* d_mean);
int long_loadings_sweep_lazy(const mbpls_mat* m, const double* d_ts, double inv_tt, double* d_p, double* d_ssq_p,
                             const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                             const double* d_P, int64_t ldp, int kprev, const double* d_cu, const double* d_ce,
                             double* d_d, const double* d_r2, const double* d_mean);
int long_loadings_sweep_raw(const mbpls_mat* m, const double* d_ts, double inv_tt, double* d_p, double* d_ssq_p,
                            const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                            const double* d_r2, const double* d_mean);
int long_standardize_stats_sweep(const mbpls_mat* src, int mode, double* d_mean, double* d_scale, double* d_rscale,
                                 double* d_r2, double* d_ssq_x, double* d_amp, const double* d_v, double alpha,
                                 double* d_g, double* d_t, double* d_ssq_g);


// colpass_cluster.hip: the never-writes-X passes for long columns with ONE read of X -- a cluster of workgroups shares
// a column (same arguments as the mbpls_hip_*_lazy_scaled / _stats_sweep entry points; true-space outputs written by
// the kernel itself)
bool cluster_columns(const mbpls_ctx* ctx, int64_t n);
int cluster_sweep_lazy(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                       const double* d_P, int64_t ldp, int kprev, const double* d_cu, double* d_d, const double* d_r2,
                       const double* d_mean);
int cluster_loadings_sweep(const mbpls_mat* m, const double* d_ts, double inv_tt, int deflate, double* d_p,
                           double* d_ssq_p, const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                           const double* d_P, int64_t ldp, int kprev, const double* d_cu, const double* d_ce,
                           double* d_d, const double* d_r2, const double* d_mean);
int cluster_standardize_stats_sweep(const mbpls_mat* src, int mode, double* d_mean, double* d_scale, double* d_r2,
                                    double* d_ssq_x, double* d_amp, const double* d_v, double alpha, double* d_g,
                                    double* d_t, double* d_ssq_g);
int cluster_stats_reduce(const mbpls_mat* src, double* d_t, double* d_ssq_g, double* d_ssq_x, double* d_amp);

}  // namespace mbpls
