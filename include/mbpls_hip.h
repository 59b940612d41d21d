/*
 * mbpls_hip.h -- C ABI of the MI355X (gfx950) hot path of MBPLS.fit()/predict()/transform().
 *
 * The reference (DTUComputeStatisticsAndDataAnalysis/MBPLS v1.0.4) is one pure-Python module
 * that delegates every X-sized operation to numpy (mbpls/mbpls.py).  It has no FFI of its own,
 * so this header defines the boundary a maintainer would bind with ctypes:
 * each entry point below names the reference statement(s) it replaces.  Everything is plain C:
 * opaque handles, raw pointers, 64-bit sizes, int status (0 = ok, <0 = error, text via
 * mbpls_hip_last_error()).  No exceptions cross the boundary, no torch types appear in it.
 *
 * Data model
 *   X lives in HBM FEATURE-MAJOR: feature (column) j of the n x p data matrix is the contiguous
 *   run  X[j*ld .. j*ld+n)  of doubles (ld >= n, even; = numpy's asfortranarray(X) memory).
 *   Blocks of a multiblock model are column ranges [block_offsets[b], block_offsets[b+1]) of the
 *   one hstacked matrix (mbpls.py:351-357,379).  A matrix carries a CHUNK TABLE: contiguous
 *   column ranges that never straddle a block; one workgroup owns one chunk, writes one partial
 *   n-vector, and a fixed-order second stage sums the chunks of each block -- so every reduction
 *   is deterministic and bit-identical between NIPALS iterations (the loop exit at mbpls.py:841
 *   compares successive score vectors against 1e-14).
 *
 *   Pointers named d_* are DEVICE pointers (hipMalloc / mbpls_hip_malloc / torch data_ptr()).
 *   Pointers named h_* are HOST pointers.  All floating point is float64, as in the reference
 *   (mbpls.py:293,310,318).  All calls are asynchronous on the context's stream unless they
 *   return data to the host; mbpls_hip_sync() drains the stream.
 */
#ifndef MBPLS_HIP_H
#define MBPLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBPLS_HIP_ABI_VERSION 9   /* 9: MBPLS_ERR_COMM, mbpls_hip_comm_set_timeout / _inject_fault (bounded waits under a communicator); 8: mbpls_hip_fence (overlapped copies ordered behind what was queued before the pass they run beside); 7: mbpls_hip_cluster_columns (cluster passes launched cooperatively, capacity from the occupancy query, time-outs flagged in host-mapped memory); 6: mbpls_hip_mat_transpose; mbpls_nipals_step.C / ldc / host_seq, mbpls_hip_queue_r_update, mbpls_hip_flush_r_update (R_ / beta_ column by column during the fit), mbpls_hip_copier_submit / _wait (native copier thread; replaces mbpls_hip_event_record / _memcpy_d2h_after / _event_destroy); 5: mbpls_hip_cluster_status (one-read passes for long columns), mbpls_hip_warmup, mbpls_hip_right_multiply_folds, mbpls_hip_cv_qloops; 2: device-resident step, RCCL communicator, scaled Gram / xt_v, masked NaN standardise; 3: mbpls_hip_loadings_sweep_raw; 4: mbpls_hip_comm_query, mbpls_hip_set_true_space_outputs, mbpls_hip_cross_product_dev, mbpls_hip_host_alloc, nu2_host, mbpls_hip_cv_weights / _cv_fold_stats / _xt_v_centred, mbpls_hip_finalise_small, mbpls_hip_right_multiply_dev, mbpls_hip_nipals_subspace_qloop */

enum {
  MBPLS_OK = 0,
  MBPLS_ERR_ARG = -1,          /* bad argument / shape */
  MBPLS_ERR_HIP = -2,          /* a HIP runtime call failed */
  MBPLS_ERR_OOM = -3,          /* device allocation failed */
  MBPLS_ERR_UNSUPPORTED = -4,  /* shape outside what the kernels cover (e.g. n > 16384 fused) */
  MBPLS_ERR_COMM = -5          /* a peer of the RCCL communicator has gone (asynchronous error or missed deadline); the
                                  communicator was aborted, the context's communicator is dead */
};

enum { MBPLS_ROW_MAJOR = 0,    /* host array is n x p C-order (numpy default), ld = row stride  */
       MBPLS_COL_MAJOR = 1 };  /* host array is feature-major / Fortran order, ld = col stride */

/* scaling mode of mbpls_hip_standardize_sweep */
enum { MBPLS_SCALE_FIT = 0,    /* compute mean/scale per column (StandardScaler.fit_transform)  */
       MBPLS_SCALE_APPLY = 1,  /* apply given mean/scale (StandardScaler.transform)             */
       MBPLS_SCALE_NONE = 2 }; /* standardize=False: copy only, still emits column sums of squares */

typedef struct mbpls_ctx mbpls_ctx;   /* device + stream + workspace */
typedef struct mbpls_mat mbpls_mat;   /* feature-major matrix in HBM + block offsets + chunk table */

/* ---- library / context ------------------------------------------------------------------ */
int         mbpls_hip_abi_version(void);
const char* mbpls_hip_last_error(void);
/* Number of visible HIP devices; does not create a context. */
int mbpls_hip_device_count(int* count);
/* stream: a hipStream_t, or NULL for the default (null) stream -- which is also PyTorch's
 * default stream, so RCCL collectives issued through torch.distributed order correctly
 * against these kernels. */
int mbpls_hip_create(int device, void* stream, mbpls_ctx** out);
int mbpls_hip_destroy(mbpls_ctx* ctx);
/* Optional: pay the process-wide first-use costs now instead of inside the first fit -- the code objects of the kernel
 * translation units (loaded lazily otherwise), the pinned bounce buffers, the first event.  The Python package calls it
 * right after creating a context (on the calling thread: from a helper thread the code-object loads deadlocked against
 * a dlopen of the main thread under rocprofv3). */
int mbpls_hip_warmup(mbpls_ctx* ctx);
int mbpls_hip_sync(mbpls_ctx* ctx);
int mbpls_hip_device_info(mbpls_ctx* ctx, char* name, size_t name_len, int* compute_units,
                          size_t* total_bytes, size_t* free_bytes);
/* PCI bus id of the context's device ("0000:c1:00.0"): a multi-GPU launcher proves with it that its N ranks sit on N
 * distinct devices (bench.py prints it per rank beside what RCCL reports).                                            */
int mbpls_hip_device_pci_bus_id(mbpls_ctx* ctx, char* out, size_t out_len);
/* Tunables (all optional): "chunks_per_cu", "rows_per_thread", "prefetch", "xw_chunks_per_cu",
 * "xw_mfma_from" (x_w with more right-hand sides than this uses the matrix-core kernel, default 8),
 * "gram_slice_features" (features per split-K slice of the Gram kernel for long feature ranges, 0 = off),
 * "long_columns_from" (0..65536, default 8192: matrices with more samples than this use the two-phase long-column
 * passes instead of the register-resident ones, which spill above 8192 and have no lazy / implicit flavour there),
 * "nan_long_from" (0..65536, default 8192: the missing-value passes stream columns with more samples than this
 * instead of holding them in registers -- any n; 0 = every n above the several-columns-per-wave range; 65536 = the
 * register shapes wherever they exist, which spill above 8192),
 * "sub_columns" (default 1: matrices with n <= 1024 samples run the column passes with several columns per wave;
 * 0 = one column per workgroup at every n -- same results, kept for A/B tests),
 * "cluster_passes" (default 1: the never-writes-X passes of long columns -- statistics + sweep, sweep, loadings + sweep
 * on the raw matrix -- run on clusters of workgroups that share a column and read X ONCE, csrc/colpass_cluster.hip;
 * 0 = the two-phase passes, two reads), "cluster_slab_rows" (128..8192, default 8192: rows of a column held by one
 * workgroup of a cluster), "cluster_spin_limit" (polls before a hand-off between the workgroups of a cluster gives up),
 * "cluster_cap" (workgroups of a cluster pass the device holds at once; -1 = ask the runtime, the default; tests).
 * Returns MBPLS_ERR_ARG for unknown keys. */
int mbpls_hip_set_tunable(mbpls_ctx* ctx, const char* key, int value);
/* Cluster passes (long columns, one read of X; replaces the two visits numpy's X.T.dot(u) :856 and X.dot(w) :875 make):
 * the workgroups of a cluster hand partial column sums to each other inside the launch, every wait bounded by
 * "cluster_spin_limit".  Drains the stream and returns in *timeouts how many waits gave up since the last call
 * (0 on a healthy device; results of a pass that gave up are garbage: the caller switches "cluster_passes" off and runs
 * the fit again -- these passes never write X).  Without a time-out the call is a stream synchronisation and a read
 * of one host-mapped word. */
int mbpls_hip_cluster_status(mbpls_ctx* ctx, int* timeouts);
/* Column shards: this context's count of timed-out hand-offs so far as a double into *d_word (a word of the exchange
 * buffer), on the stream -- queued in front of the all-reduce of that buffer, so that every rank reads the sum over the
 * ranks behind the pass and all of them take the same decision (run the fit again on the two-phase passes, or not).
 * Does not reset the count.                                                                                          */
int mbpls_hip_cluster_status_into(mbpls_ctx* ctx, double* d_word);
/* *yes = 1 when the never-writes-X passes of a matrix with n samples run on clusters of workgroups on this context:
 * "cluster_passes" is on, n is above "long_columns_from", and the device holds a whole pass at once (asked from the
 * runtime's occupancy query; the launches are cooperative, and a refused one switches the context to the two-phase
 * passes for good). */
int mbpls_hip_cluster_columns(mbpls_ctx* ctx, int64_t n, int* yes);

/* ---- raw device memory (so a host needs nothing but this library) ------------------------- */
int mbpls_hip_malloc(mbpls_ctx* ctx, size_t bytes, void** d_ptr);
int mbpls_hip_free(mbpls_ctx* ctx, void* d_ptr);
int mbpls_hip_memcpy_h2d(mbpls_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int mbpls_hip_memcpy_d2h(mbpls_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);  /* synchronous */
/* D2H on a separate non-blocking stream: does NOT wait for the kernels queued on the context's stream behind the last
 * mbpls_hip_fence, so results that are already final (e.g. loadings column k-1) reach the host while the next pass
 * runs.  It DOES wait for what was queued before that fence: call mbpls_hip_fence right before launching the pass
 * the copy is to run beside -- the small kernels that produced the column (a scaling, an upload) are then through.
 * Synchronous w.r.t. the host on return. */
int mbpls_hip_fence(mbpls_ctx* ctx);
int mbpls_hip_memcpy_d2h_overlap(mbpls_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int mbpls_hip_memcpy_d2d(mbpls_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
int mbpls_hip_memset(mbpls_ctx* ctx, void* d_ptr, int value, size_t bytes);

/* ---- timing (HIP events on the context's stream; used by bench.py for the roofline) ------- */
int mbpls_hip_timer_start(mbpls_ctx* ctx);
int mbpls_hip_timer_stop(mbpls_ctx* ctx, double* elapsed_ms);   /* synchronous */

/* Per-kernel profile: when enabled every hot-path launch is bracketed by HIP events on the
 * context's stream (main kernel only, not the small chunk-reduction that follows it).
 * kind: one of MBPLS_K_*.  profile_get drains the stream and returns launches / summed ms /
 * summed algorithmic bytes (SURVEY 8d per-kernel figures) since the last reset. */
enum { MBPLS_K_STANDARDIZE = 0, MBPLS_K_SWEEP = 1, MBPLS_K_DEFLATE_SWEEP = 2, MBPLS_K_DEFLATE = 3,
       MBPLS_K_XT_V = 4, MBPLS_K_X_W = 5, MBPLS_K_GRAM = 6, MBPLS_K_COUNT = 7 };
int mbpls_hip_profile_enable(mbpls_ctx* ctx, int on);
int mbpls_hip_profile_reset(mbpls_ctx* ctx);
int mbpls_hip_profile_get(mbpls_ctx* ctx, int kind, int64_t* launches, double* total_ms, double* total_bytes);

/* ---- matrices -------------------------------------------------------------------------------
 * Replaces the host-side copies of fit(): deepcopy (mbpls.py:303), check_array(copy=True) (:310),
 * np.hstack (:379).  block_offsets has nblocks+1 entries, block_offsets[0]=0, last = p.
 * d_storage: NULL (library allocates p*ld doubles) or caller-owned device memory of that size. */
int mbpls_hip_mat_create(mbpls_ctx* ctx, int64_t n, int64_t p, int nblocks,
                         const int64_t* block_offsets, void* d_storage, mbpls_mat** out);
/* dst = src^T on the device: src is n x p (feature-major: p columns of n samples), dst must have been created as p x n
 * (n columns of p "samples": the row-major image of src).  With it the p x p product X^T X of the n >= p branches
 * (mbpls.py:586, VAR = X^T X) is the association matrix of the transposed matrix -- mbpls_hip_gram(dst) -- on the
 * symmetric matrix-core kernel, instead of p/48 many-right-hand-side products that each read all of X.              */
int mbpls_hip_mat_transpose(const mbpls_mat* src, mbpls_mat* dst);
int mbpls_hip_mat_destroy(mbpls_mat* m);
int mbpls_hip_mat_info(const mbpls_mat* m, int64_t* n, int64_t* p, int64_t* ld, void** d_data,
                       int* nblocks, int* nchunks);
/* Host -> device, columns [col0, col0+ncols) of the matrix.  h_src points at element (0, 0) of
 * the n x ncols source; layout/ld_host describe it.  Row-major sources are transposed on the GPU. */
int mbpls_hip_mat_upload(mbpls_mat* m, const double* h_src, int64_t ld_host, int layout,
                         int64_t col0, int64_t ncols);
/* Device -> host, same conventions (used for samples / debugging, not on the fit path). */
int mbpls_hip_mat_download(const mbpls_mat* m, double* h_dst, int64_t ld_host, int layout,
                           int64_t col0, int64_t ncols);
int mbpls_hip_mat_copy(mbpls_mat* dst, const mbpls_mat* src);
/* Synthetic wide matrix with planted latent structure, generated in HBM (SURVEY 8d make_latent
 * recipe with a counter-based RNG): X[:, j] = (T * s) @ L[:, j] + noise * N(0,1), for global
 * column index col_base + j.  d_T is n x r column-major (T already scaled by s). */
int mbpls_hip_mat_fill_latent(mbpls_mat* m, const double* d_T, int r, double noise, uint64_t seed,
                              int64_t col_base);

/* Measurement aid (bench.py): the best read-only bandwidth a kernel reaches on THIS device over this matrix's
 * storage -- slab-wise non-temporal 16-byte loads, nothing stored; best of `reps` launches, GB/s.  The HBM-bound
 * passes are priced against the nominal 8 TB/s AND against this measured ceiling.                                */
int mbpls_hip_probe_read_bandwidth(const mbpls_mat* m, int reps, double* best_gbps);

/* ---- the hot path -----------------------------------------------------------------------------
 * Common conventions: d_v / d_ts are n-vectors; d_g / d_p are p-vectors (may point into a column
 * of a p x K device array); d_t is n*nblocks (block b at d_t + b*n); d_ssq_* are nblocks doubles.
 * Any output pointer may be NULL to skip that output.                                            */

/* fit() pre-processing, fused with the first NIPALS/UNIPALS sweep.
 *   replaces StandardScaler().fit_transform per block (mbpls.py:314,:323), (X**2).sum() per block
 *   (:829,:945,:431,:439,:523,:531,:756,:776) and -- when d_v is given -- the first
 *   X_b.T.dot(u)/u.T.dot(u) (:856) and X_b.dot(w_b) (:875) of component 0.
 *   src -> dst (may be the same matrix for in-place = copy=False).
 *   mode FIT:   d_mean/d_scale (p each) are OUTPUTS;  APPLY: inputs;  NONE: ignored.
 *   d_ssq_x[b] = sum of squares of the standardised block b.
 *   sweep part as mbpls_hip_sweep (skipped when d_v == NULL).                                     */
int mbpls_hip_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode,
                                double* d_mean, double* d_scale, double* d_ssq_x,
                                const double* d_v, double alpha,
                                double* d_g, double* d_t, double* d_ssq_g);

/* Same, for a fit on a ROW SUBSET of a resident matrix (cross-validation folds, mbpls users call
 * cross_val_predict(MBPLS(k), X, y, cv=n)): d_rowmask has n doubles, 0 = sample excluded.  Excluded rows
 * are written as 0 into dst -- every later pass then ignores them for free -- and mean / variance are
 * taken over the n_eff included samples.  d_rowmask == NULL is the plain call. */
int mbpls_hip_standardize_sweep_masked(const mbpls_mat* src, mbpls_mat* dst, int mode,
                                       const double* d_rowmask, double n_eff,
                                       double* d_mean, double* d_scale, double* d_ssq_x,
                                       const double* d_v, double alpha,
                                       double* d_g, double* d_t, double* d_ssq_g);

/* d_var != NULL: every MBPLS_SCALE_FIT standardise pass on this context also writes the per-column variance
 * (sklearn StandardScaler.var_, mbpls.py:307,:316; 0 for a constant column, whose scale_ is 1).  NULL switches it off. */
int mbpls_hip_set_variance_output(mbpls_ctx* ctx, double* d_var);

/* One NIPALS inner iteration over X in ONE read (mbpls.py:856 + :875, a2+a3 of SURVEY 8a):
 *   g = alpha * X^T v                (w~_b = X_b^T u / (u^T u), all blocks at once)
 *   t_b = X_b g_b                    (UN-normalised block scores; caller divides by ||g_b||)
 *   ssq_g[b] = ||g_b||^2
 * Also UNIPALS p>n: X (X^T y) (:489) and X^T u + X_b w_b (:503,:517).                            */
int mbpls_hip_sweep(const mbpls_mat* m, const double* d_v, double alpha,
                    double* d_g, double* d_t, double* d_ssq_g);

/* Loadings + rank-1 deflation in ONE read + ONE write (mbpls.py:928 + :969, a5+a7), optionally
 * fused with the next component's first sweep on the deflated matrix:
 *   p = inv_tt * X^T ts ;  X <- X - ts p^T ;  ssq_p[b] = ||p_b||^2
 *   if d_v: g = alpha * X_new^T v ; t_b = X_new,b g_b ; ssq_g[b]
 * deflate = 0 computes p (and the sweep on the un-deflated X) without writing X.                   */
int mbpls_hip_deflate_sweep(mbpls_mat* m, const double* d_ts, double inv_tt, int deflate,
                            double* d_p, double* d_ssq_p,
                            const double* d_v, double alpha,
                            double* d_g, double* d_t, double* d_ssq_g);

/* LAZY (implicit) deflation: X in HBM stays the standardised X_0; the deflated matrix
 *   X_k = X_0 - sum_{i<kprev} ts_i P[:,i]^T           (mbpls.py:969 applied algebraically)
 * is never written.  With d_cu[i] = <ts_i, v>, d_ce[i] = <ts_i, ts> (host-computed, kprev doubles):
 *   g_j  = alpha (<x_j, v> - sum_i P[j,i] cu_i),   p_j = inv_tt (<x_j, ts> - sum_i P[j,i] ce_i)
 *   t_b  = sum_{j in b} g_j x_j   (RAW accumulation; the caller subtracts  Ts d_b,  d_b[i] = sum_j g_j P[j,i])
 * so loadings + deflation + next sweep is ONE READ of X (8np bytes) instead of read + write.
 * d_P: the p x K loadings array (column-major, ldp), columns [0, kprev) final.  d_d: nblocks x kprev
 * (sweep) or nblocks x (kprev+1) (loadings_sweep: entry kprev belongs to the component being extracted,
 * whose deflation the fused sweep already sees).  d_v == NULL in loadings_sweep: loadings only. */
int mbpls_hip_sweep_lazy(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t,
                         double* d_ssq_g, const double* d_P, int64_t ldp, int kprev, const double* d_cu,
                         double* d_d);
int mbpls_hip_loadings_sweep_lazy(const mbpls_mat* m, const double* d_ts, double inv_tt, double* d_p,
                                  double* d_ssq_p, const double* d_v, double alpha, double* d_g, double* d_t,
                                  double* d_ssq_g, const double* d_P, int64_t ldp, int kprev,
                                  const double* d_cu, const double* d_ce, double* d_d);

/* ---- implicit standardisation: the fit never writes X -------------------------------------------------------
 * With lazy deflation no pass after the first needs the standardised matrix in HBM either: Z = (X - mean)/scale is
 * applied algebraically to the RAW columns.  Every vector of a fit on standardised data (u, ts, the columns of the
 * centred Y) sums to zero, so the column means drop out of the dot products, <x_j - mean_j 1, v> = <x_j, v>.
 * Everything is computed in raw-column space, with raw-space weights g~_j = scale_j g_j and loadings
 * p~_j = scale_j p_j:
 *   scale_j z_j^(k) = x_j - mean_j 1 - sum_i ts_i P~[j,i],
 *   g~_j = alpha (<x_j, v> - sum_i P~[j,i] <ts_i, v>),      p~_j = inv_tt (<x_j, ts> - sum_i P~[j,i] <ts_i, ts>),
 *   sum_j g_j z_j = sum_j c_j x_j - const 1 - sum_i ts_i (sum_j c_j P~[j,i]),      c_j = g~_j / scale_j^2,
 * i.e. the plain lazy pass plus one per-column scalar 1/scale_j^2 in the accumulations.  The constant is the mean
 * of the accumulated vector (block scores of centred columns are centred): the caller subtracts it, and rescales
 * finished g~ / p~ columns by 1/scale (mbpls_hip_scale_rows).  The first pass only takes the statistics (no store:
 * 8np instead of 16np bytes) and X stays the caller's matrix (copy=True costs no second 8np-byte buffer).
 * The statistics pass shifts every element by the column's first sample before it is used, so neither its variance
 * (amplification ((x_0j - mean_j)/scale_j)^2 <= n) nor its dot products suffer from offsets, and it reports
 * max_j |mean_j|/scale_j per block: above a few units the later passes should be given d_mean (see below).
 * The caller must pass zero-sum vectors (re-centre to rounding).
 *
 * StandardScaler statistics (MBPLS_SCALE_FIT; MBPLS_SCALE_NONE: sums of squares only) + the first sweep on the
 * raw columns held in registers; nothing is written to X.  d_rscale / d_r2 (required with MBPLS_SCALE_FIT):
 * 1/scale_j and 1/scale_j^2 (0 for a constant column); d_amp[nblocks] (optional) = the SQUARE of
 * max_j |mean_j|/scale_j, the rounding amplification of un-centred raw-space dot products.                       */
int mbpls_hip_standardize_stats_sweep(const mbpls_mat* src, int mode, double* d_mean, double* d_scale,
                                      double* d_rscale, double* d_r2, double* d_ssq_x, double* d_amp,
                                      const double* d_v, double alpha, double* d_g, double* d_t,
                                      double* d_ssq_g);
/* mbpls_hip_sweep_lazy / mbpls_hip_loadings_sweep_lazy on a RAW matrix with implicit standardisation (d_r2: p
 * values 1/scale_j^2; NULL = the plain lazy passes).  d_g / d_p receive and d_P holds the RAW-SPACE columns g~,
 * p~; d_ssq_g / d_ssq_p are the squared norms of the TRUE g_b, p_b; d_d as in the plain variants with c_j in
 * place of g_j.  d_mean (p values, optional, needs d_r2): the column means, subtracted from every element in
 * registers before the dot products and the accumulation -- a raw-space dot product carries rounding noise of
 * |mean_j|/scale_j ulps, which keeps badly offset data from ever meeting the NIPALS stop rule (mbpls.py:841).
 * The subtraction costs 8-13 % of the pass, so callers give d_mean only when the statistics pass reported
 * max_j |mean_j|/scale_j above a few units.                                                                     */
int mbpls_hip_sweep_lazy_scaled(const mbpls_mat* m, const double* d_v, double alpha, double* d_g, double* d_t,
                                double* d_ssq_g, const double* d_P, int64_t ldp, int kprev, const double* d_cu,
                                double* d_d, const double* d_r2, const double* d_mean);
int mbpls_hip_loadings_sweep_lazy_scaled(const mbpls_mat* m, const double* d_ts, double inv_tt, double* d_p,
                                         double* d_ssq_p, const double* d_v, double alpha, double* d_g,
                                         double* d_t, double* d_ssq_g, const double* d_P, int64_t ldp, int kprev,
                                         const double* d_cu, const double* d_ce, double* d_d,
                                         const double* d_r2, const double* d_mean);
/* The same pass WITHOUT deflating: p~ = inv_tt X^T ts and, from the same read, the sweep g~ = alpha X^T v,
 * t_b = sum_j c_j x_j of the UNDEFLATED raw matrix (several responses: X is standardised implicitly and never
 * deflated, mbpls_amd/solvers.py _fit_nipals_subspace).  Arguments as above with kprev = 0.                      */
int mbpls_hip_loadings_sweep_raw(const mbpls_mat* m, const double* d_ts, double inv_tt, double* d_p, double* d_ssq_p,
                                 const double* d_v, double alpha, double* d_g, double* d_t, double* d_ssq_g,
                                 const double* d_r2, const double* d_mean);
/* d_C[j, k] = d_A[j, k] * d_r[j] for a p x c column-major array (row scaling; raw-space -> true loadings).      */
int mbpls_hip_scale_rows(mbpls_ctx* ctx, int64_t p, int c, const double* d_A, int64_t lda, const double* d_r,
                         double* d_C, int64_t ldc);

/* G = alpha * X^T V for c right-hand sides (mbpls.py:503,:731,:734,:998,:1011).
 *   d_V: n x c column-major (leading dim ldv), d_G: p x c column-major (leading dim ldg).
 *   d_ssq (optional): nblocks x c, d_ssq[k*nblocks + b] = ||G[block b, k]||^2.                     */
int mbpls_hip_xt_v(const mbpls_mat* m, const double* d_V, int64_t ldv, int c, double alpha,
                   double* d_G, int64_t ldg, double* d_ssq);

/* The same product on the RAW matrix with the standardisation applied to the operand in registers,
 * G = alpha ((X - 1 mean^T) diag(rscale))^T V  (rscale_j = 1/scale_j, 0 for a constant column): the KERNEL post-pass
 * X^T [U Ts] (mbpls.py:731,:734) without a standardised copy of X.  Matrix cores for every c.                     */
int mbpls_hip_xt_v_scaled(const mbpls_mat* m, const double* d_mean, const double* d_rscale, const double* d_V,
                          int64_t ldv, int c, double alpha, double* d_G, int64_t ldg, double* d_ssq);

/* OUT = X W for c right-hand sides (mbpls.py:413,:517,:661,:752,:1006,:1117,:1386,:1408).
 *   d_W: p x c column-major (ldw).  per_block = 0: d_out is n x c column-major (ldo);
 *   per_block = 1: d_out holds nblocks matrices, block b at d_out + b*ldo*c, each X_b W_b.
 *   d_mean/d_scale (optional, p each): standardise X on the fly, x <- (x - mean_j)/scale_j
 *   (predict/transform of new data, mbpls.py:1369,:1372,:1097,:1103) without a second pass.       */
int mbpls_hip_x_w(const mbpls_mat* m, const double* d_W, int64_t ldw, int c, int per_block,
                  const double* d_mean, const double* d_scale, double* d_out, int64_t ldo);

/* ---- missing values: sparse_data=True, NaN-aware NIPALS (mbpls.py:255-271, :846-854, :865-873, :919-926) -----
 * A NaN in X marks a missing entry.  It stays NaN in HBM through standardisation and deflation (as in the
 * reference) and every sum skips it:
 *   g_j = sum_{i present} x_ij v_i / sum_{i present} v_i^2            un-normalised block weights (:848-852)
 *   t_i = sum_{j present} x_ij g_j ,  d_i = sum_{j present} g_j^2     per block; the block score over the
 *         normalised weights w = g/||g|| is  t_i ||g|| / d_i  for a row with missing entries (:869-872) and
 *         t_i / ||g|| without (where d_i = ||g||^2) -- the caller forms it
 *   p_j = sum_{i present} x_ij ts_i / sum_{i present} ts_i^2          loadings (:921-925)
 * Buffers as in the dense passes: d_g / d_p p-vectors, d_t / d_d n x nblocks column-major, d_ssq_* nblocks.
 *
 * Standardise with the statistics of the present entries of every column (sklearn StandardScaler on NaN data;
 * d_count, optional: present samples per column), d_ssq_x = nansum(X_b^2) (:826,:943); fused first sweep when
 * d_v is given.                                                                                           */
int mbpls_hip_nan_standardize_sweep(const mbpls_mat* src, mbpls_mat* dst, int mode, double* d_mean,
                                    double* d_scale, double* d_count, double* d_ssq_x, const double* d_v,
                                    double* d_g, double* d_t, double* d_d, double* d_ssq_g);
/* The same for a fit on a ROW SUBSET (cross-validation folds of data with missing values): d_rowmask has n doubles,
 * 0 = sample excluded.  Excluded rows count in no statistic and are written as 0 into dst, where -- their entries of
 * u and ts being 0 as well -- every later masked pass ignores them.  NULL = the plain call.                        */
int mbpls_hip_nan_standardize_sweep_masked(const mbpls_mat* src, mbpls_mat* dst, int mode, const double* d_rowmask,
                                           double* d_mean, double* d_scale, double* d_count, double* d_ssq_x,
                                           const double* d_v, double* d_g, double* d_t, double* d_d,
                                           double* d_ssq_g);
/* One NIPALS sweep over data with missing values (one read of X).                                          */
int mbpls_hip_nan_sweep(const mbpls_mat* m, const double* d_v, double* d_g, double* d_t, double* d_d,
                        double* d_ssq_g);
/* Loadings, X <- X - ts p^T on the present entries (:969; deflate = 0: loadings only) and, when d_v is
 * given, the first sweep of the next component on the deflated columns.  One read + one write of X.       */
int mbpls_hip_nan_deflate_sweep(mbpls_mat* m, const double* d_ts, int deflate, double* d_p, double* d_ssq_p,
                                const double* d_v, double* d_g, double* d_t, double* d_d, double* d_ssq_g);
/* mbpls_hip_x_w on data with missing values (predict :1376-1384, transform :1106-1115, :1133-1153).
 * indicator = 0: missing entries contribute nothing to X W.  indicator = 1: OUT[i,c] = sum over the PRESENT
 * columns j of row i of W[j,c] (the masked sums the sequential-deflation block scores of new data need).   */
int mbpls_hip_x_w_nan(const mbpls_mat* m, const double* d_W, int64_t ldw, int c, int per_block,
                      const double* d_mean, const double* d_scale, int indicator, double* d_out, int64_t ldo);

/* Association matrix of the KERNEL method: d_G (n x n, leading dim ldg) = X X^T (mbpls.py:704),
 * f64 MFMA SYRK; both triangles are written (bitwise symmetric, like numpy's syrk dispatch).      */
int mbpls_hip_gram(const mbpls_mat* m, double* d_G, int64_t ldg);
/* The association matrix of the STANDARDISED data from the RAW matrix: z = (x - mean_j) * rscale_j is applied to
 * both MFMA operands in registers (rscale_j = 1/scale_j, 0 for a constant column; both NULL = mbpls_hip_gram), so
 * the KERNEL method needs no standardised copy of X (mbpls.py:314 + :704 in one read).  centre_first = 1: subtract,
 * then multiply (any offset); 0: one fma(x, rscale, -mean rscale) per value -- its rounding grows with
 * |mean_j|/scale_j, for data whose offset measure (mbpls_hip_standardize_stats_sweep d_amp) is small.             */
int mbpls_hip_gram_scaled(const mbpls_mat* m, const double* d_mean, const double* d_rscale, int centre_first,
                          double* d_G, int64_t ldg);

/* ---- multi-response NIPALS in association-matrix space ----------------------------------------------------
 * One iteration of the inner loop (mbpls.py:841-913) given d_a = A z, A = sum_b X_b X_b^T (mbpls_hip_gram per block,
 * mbpls_hip_x_w on the Grams), z = P u, P = I - Ts Ts^T of the k previous components (d_Tp: n x k, column-major):
 *   ts = P a/||P a||, diff = ||ts_old - ts|| (:887; norm_kind 0: 2-norm, 1: sum|.|, 2: max|.|, 3: min|.|),
 *   v = Y^T ts/(ts^T ts) (:899), u = Y v/(v^T v) normalised (:911-913), z_next = P u,
 * because T sw = P A z for the reference's t_b = X_b w_b and sw ~ T^T u (the per-block products are only needed
 * once, at convergence).  d_state = [done, iterations, diff]: the step is a no-op once done != 0 (1: diff <= tol on a
 * pass after the first, :884; 2: max_iter reached), so a batch of steps can be queued without a host round trip and
 * the state read back is the state at the reference's loop exit.  d_ts holds the previous ts on entry.
 * k, q <= 16; d_Y: n x q column-major; d_z_next receives ntile copies of z_next, n apart (one per Gram).       */
int mbpls_hip_nipals_gram_step(mbpls_ctx* ctx, int64_t n, const double* d_a, const double* d_Tp, int k,
                               const double* d_Y, int q, int norm_kind, double tol, int max_iter, double* d_ts,
                               double* d_u, double* d_v, double* d_z_next, int ntile, double* d_state);

/* ---- multi-response NIPALS in the q-dimensional space of its iterates ------------------------------------------
 * Every sweep vector of the inner loop (mbpls.py:841-913) is u = Y c, and the super score it yields is
 * ts ~ P (sum_b X_b X_b^T) P u (P = I - Ts Ts^T: the deflation :969).  With d_G = P X (X^T (P Y)) (n x q; two
 * many-right-hand-side products over the standardised, never-deflated X per component) an iteration is
 * ts = G c / ||G c||, diff = ||ts_old - ts|| (:887), v = Y^T ts / ts^T ts (:899), u = Y v / v^T v normalised
 * (:911-913) -- n x q work -- and the WHOLE loop of a component runs inside this one launch (one workgroup).
 * d_c0: coefficients of the start vector (e_1 for Y[:, 0], :838).  Outputs: d_ts (n), d_u (n, the updated u = U_),
 * d_v (q), d_c_used (q: the sweep vector of the last iteration is Y c_used -- the weights are taken with it),
 * d_state = [1 converged | 2 iteration cap, iterations, last diff].  n <= 8192, q <= 16.                             */
int mbpls_hip_nipals_subspace_loop(mbpls_ctx* ctx, int64_t n, int q, const double* d_G, const double* d_Y,
                                   const double* d_c0, int norm_kind, double tol, int max_iter, double* d_ts,
                                   double* d_u, double* d_v, double* d_c_used, double* d_state);

/* The same loop for the 2-norm stop rule (nipals_convergence_norm = 2, the default) and ANY n, in q x q algebra: with
 * M1 = G^T G, M2 = Y^T G, M3 = Y^T Y (formed here, three small products into d_moments[3 q q]) an iteration is
 * c~ = c / sqrt(c^T M1 c) (ts = G c~, :882-883), diff = sqrt(d^T M1 d) with d = c~_old - c~ (:887), v = M2 c~ (:899),
 * c = v / v^T v / (sqrt(v^T M3 v) / v^T v) (:911-913) -- ~1 us on one wave instead of ~34 us of n x q work -- and ts, u, v
 * are formed from G and Y once the loop has stopped.  Same outputs as mbpls_hip_nipals_subspace_loop.  q <= 16.          */
int mbpls_hip_nipals_subspace_qloop(mbpls_ctx* ctx, int64_t n, int q, const double* d_G, const double* d_Y,
                                    const double* d_c0, double tol, int max_iter, double* d_ts, double* d_u,
                                    double* d_v, double* d_c_used, double* d_state, double* d_moments);

/* ---- device-resident NIPALS: a whole one-response fit queued without a host round trip -----------------------
 * Between two passes over X the reference does n x (B + q) algebra on the host (mbpls.py:860-913: norms of the
 * block weights, block scores t_b, super weights sw, super score ts, the stop rule ||ts_old - ts|| <= max_tol,
 * Y loadings v, next u).  mbpls_hip_nipals_step does exactly that on the device, reading the exchange buffer the
 * pass (mbpls_hip_sweep*, mbpls_hip_loadings_sweep_lazy*, mbpls_hip_standardize_*sweep) left behind -- layout
 * [t~: nblocks x n | ssq_g: nblocks | ssq_p: nblocks | ssq_x: nblocks | d: nblocks x nd] -- after it was all-reduced
 * over the column shards, and writing the inputs of the next pass: the n-vectors vpass / tspass (centred copies when
 * `centre`), the lazy-deflation coefficients cu / ce (kc doubles each) and par[0] = alpha, par[1] = inv_tt.
 * With mbpls_hip_set_pass_params(ctx, par) the passes take alpha / inv_tt from par instead of their host arguments
 * and turn into no-ops once par[2] != 0.
 *
 * run == 1: the pass was the first sweep of component kc (start vector Y[:, 0], :838): no stop test (:884), the next
 * pass is a sweep with the new u.  run >= 2: stop test; converged -> component kc is recorded (Ts, U, V, Tb, scal,
 * iters) and the next pass is loadings (+ lazy deflation) + first sweep of component kc + 1; not converged -> the
 * queue was built for exactly `run` iterations, so par[2] = 1 (everything queued behind becomes a no-op),
 * par[3] = kc, par[4] = the difference, and the caller resumes component kc with its own loop.
 * Shapes: nblocks <= MBPLS_STEP_MAX_BLOCKS, q <= MBPLS_STEP_MAX_RESPONSES, kc < 39; nd = number of lazily deflated
 * components the pass corrected for (0 or kc).  All arrays column-major.                                          */
#define MBPLS_STEP_MAX_BLOCKS 16
#define MBPLS_STEP_MAX_RESPONSES 16
typedef struct mbpls_nipals_step {
  int64_t n;
  int nblocks, q, kc, run, nd, centre, norm_kind;   /* norm_kind: 0 ||.||_2, 1 sum|.|, 2 max|.|, 3 min|.| (:887) */
  int mode;                /* 0 NIPALS; 1 UNIPALS p > n with one response (:481-572): the pass delivered X_b (X_b^T y),
                              one step per component (run = 1) forms ts, v, u, c (w = c X^T y, scal[kc][3][0]), the
                              block norms and scores and prepares the loadings pass; no iteration                   */
  double tol;              /* max_tol (:841)                                                                       */
  double amp2_max;         /* > 0, component 0 / run 1 only: the statistics pass left max_j (mean_j/scale_j)^2 per block
                              in the d area; above this value the step stops the queue with par[5] = 1 (the caller
                              queues again with the centring flavour of the passes, d_mean given).  With `centre` the
                              measure itself is reported in par[6] whatever amp2_max is: a caller that started with
                              centring passes sees when the data no longer ask for them                             */
  const double* xchg;      /* exchange buffer of the pass that just finished                                       */
  const double* Y;         /* n x q standardised responses                                                         */
  double* Ts;              /* n x K  super scores (column kc written at convergence)                               */
  double* Tsc;             /* n x K  the same, re-centred when `centre` (what the lazy corrections are taken with)   */
  double* U;               /* n x K                                                                                */
  double* V;               /* q x K                                                                                */
  double* Tb;              /* n x nblocks x K: block scores, block b of component k at ((k * nblocks) + b) * n       */
  double* scal;            /* K x 4 x nblocks: [||w~_b||^2 | a_b = sw_b^2 | ||p_b||^2 | -]                           */
  double* ssq_x;           /* nblocks: sums of squares of the standardised blocks (run 1 of component 0)           */
  double* iters;           /* K: NIPALS iterations per component                                                   */
  double* u_cur;           /* n: the sweep vector the finished pass was taken with (in), the next one (out)        */
  double* ts_old;          /* n: super score of the previous iteration (in/out)                                    */
  double* vpass;           /* n: sweep vector of the next pass (out)                                               */
  double* tspass;          /* n: score vector of the next loadings pass (out)                                      */
  double* cu;              /* >= kc: <ts_i, vpass>                                                                 */
  double* ce;              /* >= kc: <ts_i, tspass>                                                                */
  double* par;             /* >= 8: [alpha, inv_tt, skip flag, failed component, last diff, reason (1 = offset data),
                              offset measure, -]                                                                    */
  double* nu2_host;        /* optional (NULL: unused): K x nblocks doubles in HOST-MAPPED memory (mbpls_hip_host_alloc):
                              the step that records component kc also stores scal[kc][0][:] there, so a host thread
                              waiting on an event behind the step can read the block norms without a copy            */
  double* C;               /* optional (NULL: unused): K x K, column-major, leading dimension ldc.  The step that records
                              component kc (nd == kc) writes column kc of C = P^T W~ (mbpls.py:988) above and on the
                              diagonal: C[i][kc] = p_i^T w~_kc = sum_b d[b][i] for i < kc -- the lazy-deflation sums the
                              pass has just delivered ARE these dot products -- and C[kc][kc] = p_kc^T w~_kc =
                              ts^T (X_kc w~_kc) from the score vectors in hand.  Below the diagonal P^T W~ vanishes
                              (X_(i) w~_k = 0 for i > k), so R_ = W~ C^-1 can be built column by column while the fit
                              runs: mbpls_hip_queue_r_update                                                        */
  int64_t ldc;
  double* host_seq;        /* optional (NULL: unused): one double in HOST-MAPPED memory.  Every launch stores seq_value there
                              FIRST (before the skip test), so a host thread can learn that everything queued on the stream
                              before this step has completed without an event in the stream (an event record between two
                              launches idles the queue for ~6 us): mbpls_hip_copier_submit with h_flag                  */
  double seq_value;
} mbpls_nipals_step;
int mbpls_hip_nipals_step(mbpls_ctx* ctx, const mbpls_nipals_step* s);
/* R_ = W (P^T W)^+ (mbpls.py:988 / :569) one column per component, while the fit is still running.  P^T W~ is upper
 * triangular (see mbpls_nipals_step.C) and the column scaling of W cancels, so R C = W~ gives
 *   R[:, k] = (W~[:, k] - sum_{i<k} R[:, i] C[i][k]) / C[k][k],      beta += R[:, k] v_k  (one response, :989 / :570)
 * with d_Ccol = column k of C (k + 1 doubles) and d_vk = V[0][k], both written by the step that recorded component k.
 * The update is QUEUED: it rides as extra workgroups on the next chunk reduction of a column pass on this context (no
 * launch of its own between two passes); mbpls_hip_flush_r_update launches it by itself if no pass has taken it.
 * d_beta: p doubles, set (not accumulated) at k = 0; may be NULL.  A no-op when the device skip flag is raised.
 * The caller checks the conditioning of C afterwards (a vanishing diagonal entry = an exhausted matrix, where the
 * reference's pinv truncates) and falls back to mbpls_hip_finalise_small + mbpls_hip_right_multiply_dev.             */
int mbpls_hip_queue_r_update(mbpls_ctx* ctx, int64_t p, const double* d_w, double* d_R, int64_t ldr, int k,
                             const double* d_Ccol, const double* d_vk, double* d_beta);
int mbpls_hip_flush_r_update(mbpls_ctx* ctx);
/* d_par != NULL: every column pass on this context reads alpha = d_par[0], inv_tt = d_par[1] from the device and
 * does nothing when d_par[2] != 0.  NULL restores the host-scalar behaviour.                                        */
int mbpls_hip_set_pass_params(mbpls_ctx* ctx, const double* d_par);
/* d_dst[j] = d_src[j] / sqrt(d_nu2[block of j]): block weights w_b = w~_b / ||w~_b|| (mbpls.py:860) with the norms
 * still on the device (d_nu2: nblocks squared norms).                                                              */
int mbpls_hip_block_unit_col(const mbpls_mat* m, const double* d_src, double* d_dst, const double* d_nu2);
/* True-space outputs of the raw-space passes (those given d_r2 = 1/scale_j^2: mbpls_hip_sweep_lazy_scaled,
 * mbpls_hip_loadings_sweep_lazy_scaled, mbpls_hip_loadings_sweep_raw).  With d_rscale (1/scale_j, p doubles) set, the
 * weights they store into d_g are the TRUE-space g_j = g~_j / scale_j (mbpls.py:856 on the standardised matrix), and
 * d_p_true (optional, p doubles) receives the true-space loadings p_j = p~_j / scale_j (:928) beside the raw-space
 * ones in d_p (which the lazy corrections of later passes keep using) -- no p-sized rescaling kernel behind the pass.
 * Stays in force until changed; NULL restores raw-space outputs.                                                    */
int mbpls_hip_set_true_space_outputs(mbpls_ctx* ctx, const double* d_rscale, double* d_p_true);
/* d_dst[j] = d_src[j] * d_scalar[0] for a p-length column (UNIPALS: w = X^T u = c X^T y, mbpls.py:503).              */
int mbpls_hip_scale_col_by(mbpls_ctx* ctx, int64_t p, const double* d_src, double* d_dst, const double* d_scalar);
/* Pinned host memory the device can write directly (zero copy): *h_ptr for the host, *d_ptr for kernels.  For a few
 * scalars a kernel hands to a waiting host thread -- a blocking copy of 8 bytes into pageable memory is a blit kernel
 * that queues up behind (and slows) whatever is running.                                                           */
int mbpls_hip_host_alloc(mbpls_ctx* ctx, size_t bytes, void** h_ptr, void** d_ptr);
int mbpls_hip_host_free(mbpls_ctx* ctx, void* h_ptr);   /* (blocks still allocated are released by mbpls_hip_destroy) */
/* The copier: one native thread per context that carries finished results to the host while the compute stream keeps
 * running (csrc/copier.hip).  A job = an event recorded on the compute stream NOW + device-to-host copies that run once
 * it has completed + optionally the block normalisation of a column that has just arrived,
 *   h_dst[j] = h_src[j] / sqrt(h_nu2[b])  for offsets[b] <= j < offsets[b + 1]      (w_b = w~_b / ||w~_b||, mbpls.py:860)
 * with h_nu2 in host(-mapped) memory that is read when the job runs (mbpls_nipals_step.nu2_host).  Jobs run in submission
 * order.  *ticket identifies the job; mbpls_hip_copier_wait returns when that job and all earlier ones are complete (and
 * reports the first failure of any of them).  The host arrays must stay alive until then.                             */
typedef struct mbpls_copy {
  void* h_dst;
  const void* d_src;
  size_t bytes;
} mbpls_copy;
typedef struct mbpls_unit_scale {
  const double* h_src;
  double* h_dst;               /* may be h_src */
  int nblocks;
  const int64_t* offsets;      /* nblocks + 1 (copied at submission) */
  const double* h_nu2;         /* nblocks squared norms */
} mbpls_unit_scale;
int mbpls_hip_copier_submit(mbpls_ctx* ctx, int ncopies, const mbpls_copy* copies, const mbpls_unit_scale* scale,
                            const double* h_flag, double flag_at_least, uint64_t* ticket);
/* h_flag != NULL: no event is recorded; the job starts once *h_flag >= flag_at_least instead (h_flag: host-mapped memory
 * that a kernel queued BEHIND the producers of the copied data writes, mbpls_nipals_step.host_seq).  The caller must make
 * sure that kernel is launched -- or store the value itself when it gives up.                                          */
int mbpls_hip_copier_wait(mbpls_ctx* ctx, uint64_t ticket);

/* ---- exchange step of the column-sharded fit: RCCL over xGMI, inside the C ABI ---------------------
 * Feature columns are split over one process per GPU; every per-iteration quantity that is a sum over features (the
 * exchange buffer of a pass: block scores t~ (n x B), ||w~_b||^2, ||p_b||^2, lazy-deflation sums; X W products; the
 * K x K P^T W; the n x n association matrix of the KERNEL method) is summed over the ranks by ONE ncclAllReduce
 * (float64, in place) enqueued on the context's stream -- between the chunk reduction of the pass and the step kernel
 * that consumes it, no host synchronisation.  The reference has no counterpart (single process, numpy).
 * Bootstrap: rank 0 calls mbpls_hip_comm_unique_id, ships the MBPLS_COMM_ID_BYTES bytes to the other ranks by any
 * means (bench.py: torch.distributed's store), then every rank calls mbpls_hip_comm_init_rank (collective).
 * librccl.so is loaded with dlopen on first use.                                                                    */
#define MBPLS_COMM_ID_BYTES 128
int mbpls_hip_comm_unique_id(void* id_out);
int mbpls_hip_comm_init_rank(mbpls_ctx* ctx, int world_size, int rank, const void* id_in);
int mbpls_hip_comm_info(mbpls_ctx* ctx, int* world_size, int* rank, int* rccl_version);
/* What the communicator ITSELF reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) -- mbpls_hip_comm_info
 * returns the world size / rank the caller passed in; a launcher proves that RCCL really spans N ranks with this.   */
int mbpls_hip_comm_query(mbpls_ctx* ctx, int* rccl_ranks, int* rccl_user_rank, int* rccl_device);
int mbpls_hip_comm_destroy(mbpls_ctx* ctx);
int mbpls_hip_allreduce(mbpls_ctx* ctx, double* d_buf, size_t count);   /* sum, in place, asynchronous */
/* A dead peer must not hang the caller inside a device queue.  With a communicator of more than one rank EVERY wait of
 * this library for the compute stream (mbpls_hip_sync, the blocking copies, ...) polls the stream instead of blocking,
 * queries ncclCommGetAsyncError about once a millisecond and enforces a deadline (default 600 s; the ranks of a fit may
 * be skewed by a host-side upload, never by minutes of it).  On an asynchronous error or a missed deadline the
 * communicator is aborted (ncclCommAbort: the stuck collective leaves the stream), the call returns MBPLS_ERR_COMM with a
 * message naming the rank, and every later collective on the context fails at once -- the caller exits non-zero (it
 * never re-executes itself) and the launcher tears the job down.                                                      */
int mbpls_hip_comm_set_timeout(mbpls_ctx* ctx, int timeout_ms);
/* Test hook (tests/test_gpu_rccl.py, world size 1): the next mbpls_hip_allreduce becomes a kernel that waits as a
 * collective waits for a peer that has died; mode 1: the asynchronous error shows after `polls` queries, mode 2: never
 * (the deadline has to catch it); mode 0: off.                                                                        */
int mbpls_hip_comm_inject_fault(mbpls_ctx* ctx, int mode, int polls);

/* ---- cross-validation folds batched per read of X (mbpls_amd/cv_batch.py; the reference's callers run
 * sklearn.model_selection.cross_val_predict(MBPLS(k), X, y, cv=n): n complete refits) ---------------------------------
 * The p x F algebra of a NIPALS sweep for F folds at once, between the two many-right-hand-side products with the ONE
 * raw matrix in HBM (mbpls_hip_xt_v, mbpls_hip_x_w with F columns).  Fold f has its own column statistics d_mu / d_rs
 * (mean and 1/scale of its training rows, p x F column-major).  From d_G = X^T U (U: the folds' sweep vectors, zero on
 * their held-out rows; d_su[f] = sum_i U_if):
 *     d_g[j,f] = d_alpha[f] (d_G[j,f] - d_mu[j,f] d_su[f]) d_rs[j,f]        (mbpls.py:856 on the fold's standardised X)
 *     d_C[j,f] = d_g[j,f] d_rs[j,f]                                          (X_b d_C_b - const = X_b,std g_b, :875)
 *     d_sums[(b F + f) 2 + 0] = sum_{j in block b} d_g[j,f]^2               (block norms, :860)
 *     d_sums[(b F + f) 2 + 1] = sum_{j in block b} d_mu[j,f] d_C[j,f]       (that const)
 * predict != 0: d_G holds the folds' regression vectors beta_f on the standardised columns; d_g = d_C = beta rs is what the
 * raw matrix is multiplied with and d_sums[..1] = sum mu (beta rs) the constant to remove (mbpls.py:1369-1386); d_su /
 * d_alpha unused.  Asynchronous on the context's stream; fixed-order sums.                                          */
int mbpls_hip_cv_weights(const mbpls_mat* m, int F, int predict, const double* d_G, const double* d_mu, const double* d_rs,
                         const double* d_su, const double* d_alpha, double* d_g, double* d_C, double* d_sums);
/* Several responses: the NIPALS inner loop (mbpls.py:841-913) of `folds` cross-validation folds at once, each in the
 * q-dimensional space of its iterates (u = Y c, ts ~ G c; mbpls_hip_nipals_subspace_qloop for one problem): one wave per
 * fold.  d_moments [folds][3][q][q]: G^T G, Y^T G, Y^T Y of every fold (row-major).  d_out [folds][2 q + 3]:
 * c~ (ts = G c~ / ||.||), the c of the last sweep vector, flag (1 converged, 2 iteration cap), iterations, last diff. */
int mbpls_hip_cv_qloops(mbpls_ctx* ctx, int folds, int q, const double* d_moments, double tol, int max_iter, double* d_out);
/* d_G (p x c) = ((X - 1 d_mean^T) [o (X - 1 d_mean^T) when square])^T d_V: centred sums / sums of squares over row sets
 * given as indicator columns (the held-out rows of cross-validation folds); matrix cores, X read once.               */
int mbpls_hip_xt_v_centred(const mbpls_mat* m, const double* d_mean, int square, const double* d_V, int64_t ldv, int c,
                           double* d_G, int64_t ldg);
/* Column statistics of the training rows of F folds (StandardScaler on X[train_f], mbpls.py:307-314) from the full-data
 * mean / variance (d_mu0, d_var0: p) and the held-out rows' centred sums d_h1, d_h2 (p x F, mbpls_hip_xt_v_centred with
 * the held-out indicators; d_nf[f] = training rows of fold f): d_mean, d_rs = 1 / scale (p x F; scale 1 for a constant
 * column).  Entries whose variance is lost to cancellation get d_rs = -1 and are counted in *d_risky (uint64): the
 * caller takes those from the rows themselves.                                                                     */
int mbpls_hip_cv_fold_stats(const mbpls_mat* m, int F, const double* d_mu0, const double* d_var0, const double* d_h1,
                            const double* d_h2, const double* d_nf, double* d_mean, double* d_rs, double* d_risky);

/* ---- end-of-fit algebra on p x K device arrays (any K; 32-column tiles inside), column-major ---
 * h_out (c1 x c2 row-major, HOST) = A^T B for p x c1 / p x c2 device arrays: P_^T W (mbpls.py:988,
 * :569, :737).  Deterministic two-stage reduction; synchronous (the result is on the host on return). */
int mbpls_hip_cross_product(mbpls_ctx* ctx, int64_t p, const double* d_A, int64_t lda, int c1,
                            const double* d_B, int64_t ldb, int c2, double* h_out);
/* The same with the result left on the DEVICE (d_out: c1 x c2 row-major, c1, c2 <= 32) and no synchronisation: the
 * product can sit in a queue of launches, e.g. behind the last pass of a device-resident fit, and come back to the host
 * together with its other results.                                                                                  */
int mbpls_hip_cross_product_dev(mbpls_ctx* ctx, int64_t p, const double* d_A, int64_t lda, int c1,
                                const double* d_B, int64_t ldb, int c2, double* d_out);
/* The K x K end-of-fit algebra on the device (no host round trip between the last pass and the p x K products):
 * d_M (K x K row-major) = D (d_C D)^+ and d_MV (K x q row-major) = d_M d_V^T, with d_C = P^T W~ (K x K row-major),
 * D = diag(1 / ||W~[:, k]||), ||W~[:, k]||^2 = sum_b d_scal[(k 4 + 0) nblocks + b], d_V q x K column-major -- so that
 * R_ = W~ d_M (mbpls.py:988 / :569) and beta_ = W~ d_MV (:989 / :570).  The pseudo-inverse follows numpy.linalg.pinv
 * (singular values <= 1e-15 sigma_max dropped; one-sided Jacobi SVD on one wave).  K <= 32.                          */
int mbpls_hip_finalise_small(mbpls_ctx* ctx, int K, int nblocks, int q, const double* d_C, const double* d_scal,
                             const double* d_V, double* d_M, double* d_MV);
/* mbpls_hip_right_multiply with the c1 x c2 matrix (row-major, c1, c2 <= 32) already on the device.                  */
int mbpls_hip_right_multiply_dev(mbpls_ctx* ctx, int64_t p, const double* d_A, int64_t lda, int c1, const double* d_M,
                                 int c2, double* d_C, int64_t ldc);
/* The same for `folds` matrices in one launch (cross-validation folds batched per read of X, mbpls_amd/cv_batch.py):
 * fold f works on d_A + f a_stride, the c1 x c2 row-major matrix d_M + f c1 c2, and writes d_C + f c_stride.  Replaces
 * the per-fold R_ = W (P^T W)^+ / beta_ = R_ V_^T products (mbpls.py:988-989) of a loop of fits. */
int mbpls_hip_right_multiply_folds(mbpls_ctx* ctx, int64_t p, int folds, const double* d_A, int64_t lda, int64_t a_stride,
                                   int c1, const double* d_M, int c2, double* d_C, int64_t ldc, int64_t c_stride);
/* d_C (p x c2, ldc) = d_A (p x c1, lda) @ h_M (c1 x c2 row-major HOST matrix):
 * R_ = W (P^T W)^+ and beta_ = R_ V^T (mbpls.py:988-989), column scalings W_b = W~_b diag(1/||w~_b||)
 * (:860) with a diagonal M on the row range of block b. */
int mbpls_hip_right_multiply(mbpls_ctx* ctx, int64_t p, const double* d_A, int64_t lda, int c1,
                             const double* h_M, int c2, double* d_C, int64_t ldc);

#ifdef __cplusplus
}
#endif
#endif /* MBPLS_HIP_H */
