// Association matrix of the KERNEL method: G = X X^T (n x n) from the feature-major X (p x n).
//
// The only true dense GEMM on the path (n(n+1)p flop), so it runs on the f64 matrix cores:
// v_mfma_f64_16x16x4_f64, D(16x16) += A(16x4) B(4x16) with the reduction index k running over
// FEATURES.  In the feature-major layout both operands of one k-step are rows of X:
//   A[i][k] = X[j+k][a0+i],  B[k][c] = X[j+k][b0+c]
// and the MFMA lane map (lane l feeds A[l&15][l>>4], B[l>>4][l&15]) turns into four 128-byte
// contiguous segments per wave-load -- the reduction index is the slow one, perfectly streaming.
//
// Work decomposition: lower-triangle 128x128 tiles (SYRK), 4 waves per tile as 2x2, each wave a
// 64x64 sub-tile = 4x4 MFMA tiles (64 f64 accumulators per lane).  Tiles are ordered in 8x8
// super-tiles and dealt to XCDs in contiguous runs so the workgroups that share an XCD L2 read
// the same few 128-sample panels of each feature.  Small n: split-K over features with a
// fixed-order slice reduction (deterministic).  A final pass mirrors the lower triangle.
#include <algorithm>

#include "common.h"

namespace mbpls {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int GT = 128;  // tile edge

struct GramArgs {
  const double* ms;   // optional [p][2] = (mean_j, 1/scale_j): the Gram of Z = (X - 1 mean^T) diag(1/scale) from the RAW X
  const double* X;
  int64_t ld, n, p;
  const int* tile_i;  // [ntiles]
  const int* tile_j;
  int ntiles, nslices;
  int64_t slice_len;  // features per slice (multiple of 4)
  double* out;        // [nslices][n*ldg]
  int64_t ldg;
};

// SCALE: the operands are standardised on the fly, z = (x - mean_j) / scale_j (one subtraction and one multiplication
// per loaded value -- VALU work next to 16 MFMAs per k-step, the kernel stays matrix-core bound): the KERNEL method
// then needs neither a standardised copy of X nor the read+write pass that makes it.
// SCALE 1: z = (x - mean) * rscale (two operations per value, safe for any column offset); SCALE 2: z = fma(x, rscale,
// -mean * rscale) (one operation; its rounding is amplified by |mean_j|/scale_j, so callers use it when the
// statistics pass reported a small offset measure).
template <int SCALE>
__global__ __launch_bounds__(256, 2) void k_gram(GramArgs a) {
  // XCD-aware remap: blocks b, b+8, ... share an XCD -> give each XCD a contiguous run of work items
  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tile = wid % a.ntiles, slice = wid / a.ntiles;
  const int ti = a.tile_i[tile], tj = a.tile_j[tile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t a0 = (int64_t)ti * GT + wr * 64, b0 = (int64_t)tj * GT + wc * 64;
  const int lk = lane >> 4, li = lane & 15;

  double4_t acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int64_t jbeg = (int64_t)slice * a.slice_len;
  const int64_t jend = std::min<int64_t>(a.p, jbeg + a.slice_len);
  // per-lane BYTE offsets inside a group of 4 features: (lane>>4) selects the feature, (lane&15) the
  // sample; out-of-range samples are clamped (their rows/columns are never stored).  32-bit offsets +
  // a wave-uniform group base -> global_load with an SGPR base, no 64-bit address VALU work.
  // every lane loads TWO consecutive samples of its feature (16 bytes): half as many load instructions per k-step;
  // the even samples feed one MFMA sub-tile and the odd ones the next, i.e. sub-tile m covers the samples
  // base + 32 (m >> 1) + 2 i + (m & 1), i = 0..15 -- the rows / columns of the output are permuted accordingly
  unsigned oa[2], ob[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int64_t ra = a0 + t * 32 + 2 * li, rb = b0 + t * 32 + 2 * li;
    if (ra >= a.n) ra = 0;
    if (rb >= a.n) rb = 0;
    oa[t] = (unsigned)((lk * a.ld + ra) * 8);
    ob[t] = (unsigned)((lk * a.ld + rb) * 8);
  }
  // Software pipeline over k-steps (4 features each): a ring of PD register slots; at step s the group
  // PD-1 steps ahead is loaded into the slot consumed one step earlier, then the 16 MFMAs of the current
  // group issue.  A load therefore has (PD-1) x 16 MFMAs (~3000 cycles) to land: HBM/L2 latency is
  // covered without relying on the second wave of the SIMD.  No select and no branch sits between a load
  // and its MFMA (prefetch index clamped, not guarded), sched_barrier pins the staggered order.
  constexpr int PD = 4;
  const int64_t nsteps = (jend - jbeg) / 4;           // complete groups of 4 features
  const int64_t nmain = (nsteps / PD) * PD;
  const char* xb = (const char*)(a.X + jbeg * a.ld);
  const int64_t gstride = 4 * a.ld * 8;               // bytes between consecutive groups
  typedef double gv2_t __attribute__((ext_vector_type(2)));
  gv2_t av[PD][2], bv[PD][2];
  gv2_t msv[SCALE != 0 ? PD : 1];                           // (mean, 1/scale) of this lane's feature in each group
  const unsigned oms = (unsigned)(lk * 16);
  const char* msb = SCALE != 0 ? (const char*)(a.ms + 2 * jbeg) : nullptr;
  // The ring's loads are inline asm on purpose: hipcc's own waitcnt insertion loses the exact queue depth
  // across the loop back-edge and emits vmcnt(8) -- i.e. it waits for the groups issued one and two steps
  // ago as well, which leaves ONE step of latency cover and 35 % of the requests miss L2 here.  With the
  // loads invisible to that pass the wait is counted by hand: after issuing group g+3 there are 4 groups
  // (16 16-byte loads) in flight and vmcnt(12) retires exactly the oldest one, the group about to be consumed.
  // Rules kept (cdna_hip_programming.md 5.7): no asm output is read before its wait; sched_barrier pins
  // the order; vmcnt(0) right behind the loop so no late load lands in a re-used register.
#define MB_GLD(dst, off, base) \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory")
  if (nmain > 0) {
#pragma unroll
    for (int s = 0; s < PD - 1; ++s) {
      const char* g = xb + s * gstride;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        MB_GLD(av[s][t], oa[t], g);
        MB_GLD(bv[s][t], ob[t], g);
      }
      if constexpr (SCALE != 0) MB_GLD(msv[s], oms, msb + s * 64);
    }
    static_assert(PD == 4, "the step macro below is written for a ring of 4 slots");
#define MB_GRAM_STEP(S, SL)                                                                                 \
  {                                                                                                         \
    int64_t nx = it + (S) + PD - 1;               /* group to prefetch */                                   \
    nx = nx < nmain ? nx : nmain - 1;             /* clamp: a valid group that is never consumed */         \
    const char* g = xb + nx * gstride;                                                                      \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                         \
      MB_GLD(av[SL][t], oa[t], g);                                                                          \
      MB_GLD(bv[SL][t], ob[t], g);                                                                          \
    }                                                                                                       \
    if constexpr (SCALE != 0) {                                                                             \
      MB_GLD(msv[SL], oms, msb + nx * 64);                                                                  \
      asm volatile("s_waitcnt vmcnt(15)" ::: "memory");   /* 4 groups x 5 loads in flight, oldest retired */ \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      /* z = (x - mean) * rscale on the 8 operand values, INTERLEAVED with the 16 MFMAs: an MFMA occupies the */ \
      /* matrix pipe for 16 cycles but the issue port for 4, so two VALU ops ride along with each one        */ \
      const double mu_ = msv[S].x, rs_ = msv[S].y;                                                          \
      auto Z_ = [&](double x) { return SCALE == 2 ? fma(x, rs_, mu_) : (x - mu_) * rs_; };                  \
      double za[4], zb[4];                                                                                  \
      za[0] = Z_(av[S][0].x);                                                                               \
      zb[0] = Z_(bv[S][0].x);                                                                               \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) {   \
        /* operands needed by a later MFMA are prepared while this one runs */                              \
        if (mi == 0 && ni < 3) zb[ni + 1] = Z_(bv[S][(ni + 1) >> 1][(ni + 1) & 1]);                         \
        if (ni == 3 && mi < 3) za[mi + 1] = Z_(av[S][(mi + 1) >> 1][(mi + 1) & 1]);                         \
        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[mi], zb[ni], acc[mi][ni], 0, 0, 0);           \
      }                                                                                                     \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                      \
        __builtin_amdgcn_sched_group_barrier(0x002, SCALE == 2 ? 1 : 2, 0);   /* VALU */                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    /* 1 MFMA */                  \
      }                                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    } else {                                                                                                \
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)     \
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[S][mi >> 1][mi & 1], bv[S][ni >> 1][ni & 1], \
                                                             acc[mi][ni], 0, 0, 0);                         \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
  }
    for (int64_t it = 0; it < nmain; it += PD) {
      MB_GRAM_STEP(0, 3)
      MB_GRAM_STEP(1, 0)
      MB_GRAM_STEP(2, 1)
      MB_GRAM_STEP(3, 2)
    }
#undef MB_GRAM_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#undef MB_GLD
  // remaining complete groups (< PD of them) and the ragged feature tail (p % 4), unpipelined
  for (int64_t st = nmain; st * 4 < jend - jbeg; ++st) {
    const int64_t j = jbeg + 4 * st;
    const bool vj = j + lk < jend;
    const char* g = (const char*)(a.X + (vj ? j : jbeg) * a.ld);
    double ta[4], tb[4];
    double mu = 0.0, rs = 1.0;
    if constexpr (SCALE != 0) {
      if (vj) {
        mu = a.ms[2 * (j + lk)];
        rs = a.ms[2 * (j + lk) + 1];
      }
    }
    auto Zt = [&](double x) { return SCALE == 2 ? fma(x, rs, mu) : (x - mu) * rs; };
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double2 xa = vj ? *(const double2*)(g + oa[t]) : make_double2(0.0, 0.0);
      const double2 xq = vj ? *(const double2*)(g + ob[t]) : make_double2(0.0, 0.0);
      ta[2 * t] = vj ? Zt(xa.x) : 0.0; ta[2 * t + 1] = vj ? Zt(xa.y) : 0.0;
      tb[2 * t] = vj ? Zt(xq.x) : 0.0; tb[2 * t + 1] = vj ? Zt(xq.y) : 0.0;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[mi], tb[ni], acc[mi][ni], 0, 0, 0);
  }
  // D layout (f64): col = lane & 15, row = (lane >> 4) + 4 * reg
  double* out = a.out + (int64_t)slice * a.n * a.ldg;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int64_t row = a0 + 32 * (mi >> 1) + 2 * (lk + 4 * rg) + (mi & 1);
        const int64_t col = b0 + 32 * (ni >> 1) + 2 * li + (ni & 1);
        if (row < a.n && col < a.n) out[row * a.ldg + col] = acc[mi][ni][rg];
      }
}

// G[i][j] (j <= i) = sum_s part[s][i][j] in slice order; G[j][i] = G[i][j].
__global__ __launch_bounds__(256) void k_gram_finish(const double* __restrict__ part, int nslices, int64_t n,
                                                     int64_t ldg, double* __restrict__ G) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j > i || i >= n) return;
  double s = part[i * ldg + j];
  for (int k = 1; k < nslices; ++k) s += part[(int64_t)k * n * ldg + i * ldg + j];
  G[i * ldg + j] = s;
  G[j * ldg + i] = s;
}

}  // namespace mbpls

using namespace mbpls;

// ms[j] = (mean_j, 1/scale_j) -- or (-mean_j/scale_j, 1/scale_j) for the one-fma form -- interleaved, so a lane
// fetches both with one 16-byte load
__global__ __launch_bounds__(256) void k_pack_mean_rscale(int64_t p, const double* __restrict__ mean,
                                                          const double* __restrict__ rscale, double* __restrict__ ms,
                                                          int fused) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= p) return;
  ms[2 * j] = fused ? -mean[j] * rscale[j] : mean[j];
  ms[2 * j + 1] = rscale[j];
}

extern "C" int mbpls_hip_gram(const mbpls_mat* m, double* d_G, int64_t ldg) {
  return mbpls_hip_gram_scaled(m, nullptr, nullptr, 1, d_G, ldg);
}

extern "C" int mbpls_hip_gram_scaled(const mbpls_mat* m, const double* d_mean, const double* d_rscale, int centre_first,
                                     double* d_G, int64_t ldg) {
  MB_ARG(m && d_G, "NULL argument");
  MB_ARG((d_mean == nullptr) == (d_rscale == nullptr), "d_mean and d_rscale must be given together");
  MB_ARG(ldg >= m->n, "ldg < n");
  mbpls_ctx* ctx = m->ctx;
  const int64_t n = m->n, p = m->p;
  const int nt = (int)((n + GT - 1) / GT);
  // lower-triangle tiles in 8x8 super-tile order
  std::vector<int> ti, tj;
  for (int SI = 0; SI < nt; SI += 8)
    for (int SJ = 0; SJ <= SI; SJ += 8)
      for (int i = SI; i < std::min(nt, SI + 8); ++i)
        for (int j = SJ; j < std::min(nt, SJ + 8); ++j)
          if (j <= i) {
            ti.push_back(i);
            tj.push_back(j);
          }
  const int ntiles = (int)ti.size();
  int nslices = 1;
  const int want = ctx->cus * 4;
  if (ntiles < want) {
    nslices = (want + ntiles - 1) / ntiles;
    const int64_t max_slices = std::max<int64_t>(1, p / 256);
    nslices = (int)std::min<int64_t>(nslices, max_slices);
    // keep the slice partials within 1 GiB
    const int64_t cap = std::max<int64_t>(1, ((int64_t)1 << 30) / std::max<int64_t>(1, n * ldg * 8));
    nslices = (int)std::min<int64_t>(nslices, cap);
  }
  // Long feature ranges: workgroups that share operand panels through L2 start together but drift apart as
  // they stream (measured at n=8192: 57 TFLOP/s at p = 2e6 unsliced, 65.7 with 131072-feature slices); cutting the range into
  // slices re-aligns them at every slice boundary.  Partials: nslices x n x ldg doubles (<= 16 GiB).
  // Slices also smooth the launch quantisation: 2080 tiles on 512 resident workgroups are 4.06 rounds, the
  // last one nearly empty (the 62 TFLOP/s plateau at mid p); aim for >= 48 rounds of >= 2048 features (16 rounds
  // still leave a quarter-round tail: 68.3 -> 71 TFLOP/s at mid p).
  if (ctx->gram_slice_features > 0) {
    const int64_t by_len = (p + ctx->gram_slice_features - 1) / ctx->gram_slice_features;
    const int64_t by_rounds = std::min<int64_t>(((int64_t)ctx->cus * 2 * 48 + ntiles - 1) / ntiles, p / 2048);
    const int64_t cap = std::max<int64_t>(1, ((int64_t)16 << 30) / std::max<int64_t>(1, n * ldg * 8));
    nslices = (int)std::max<int64_t>(nslices, std::min(std::max(by_len, by_rounds), cap));
  }
  int64_t slice_len = (p + nslices - 1) / nslices;
  slice_len = (slice_len + 3) / 4 * 4;
  nslices = (int)((p + slice_len - 1) / slice_len);
  const size_t tile_bytes = sizeof(int) * (size_t)ntiles * 2;
  const size_t part_bytes = sizeof(double) * (size_t)nslices * (size_t)n * (size_t)ldg;
  const size_t ms_bytes = d_mean ? ((sizeof(double) * 2 * (size_t)p + 255) / 256) * 256 : 0;
  MB_TRY(ensure_ws(ctx, tile_bytes + 256 + part_bytes + ms_bytes));
  int* d_ti = (int*)ctx->ws;
  int* d_tj = d_ti + ntiles;
  double* part = (double*)((char*)ctx->ws + ((tile_bytes + 255) / 256) * 256);
  double* ms = d_mean ? (double*)((char*)part + ((part_bytes + 255) / 256) * 256) : nullptr;
  if (ms) {
    hipLaunchKernelGGL(k_pack_mean_rscale, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, ctx->stream, p, d_mean,
                       d_rscale, ms, centre_first ? 0 : 1);
    MB_HIP(hipGetLastError());
  }
  MB_HIP(hipMemcpyAsync(d_ti, ti.data(), sizeof(int) * ntiles, hipMemcpyHostToDevice, ctx->stream));
  MB_HIP(hipMemcpyAsync(d_tj, tj.data(), sizeof(int) * ntiles, hipMemcpyHostToDevice, ctx->stream));
  MB_SYNC_STREAM(ctx);  // ti/tj are stack-owned
  GramArgs a{};
  a.X = m->data; a.ld = m->ld; a.n = n; a.p = p; a.ms = ms;
  a.tile_i = d_ti; a.tile_j = d_tj; a.ntiles = ntiles; a.nslices = nslices; a.slice_len = slice_len;
  a.out = part; a.ldg = ldg;
  {
    ProfScope prof(ctx, MBPLS_K_GRAM, 8.0 * (double)n * (double)p);
    if (ms && centre_first) hipLaunchKernelGGL(k_gram<1>, dim3((unsigned)(ntiles * nslices)), dim3(256), 0, ctx->stream, a);
    else if (ms) hipLaunchKernelGGL(k_gram<2>, dim3((unsigned)(ntiles * nslices)), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_gram<0>, dim3((unsigned)(ntiles * nslices)), dim3(256), 0, ctx->stream, a);
  }
  MB_HIP(hipGetLastError());
  dim3 fgrid((unsigned)((n + 255) / 256), (unsigned)n);
  hipLaunchKernelGGL(k_gram_finish, fgrid, dim3(256), 0, ctx->stream, (const double*)part, nslices, n, ldg, d_G);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

// one kernel of this translation unit, for mbpls_hip_warmup (its code object is loaded on first use otherwise)
namespace mbpls {
const void* anchor_gram() { return (const void*)&k_gram_finish; }
}  // namespace mbpls
