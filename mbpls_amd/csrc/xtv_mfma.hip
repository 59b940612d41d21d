// G = alpha * X^T V for MANY right-hand sides (c >= 3: KERNEL post-pass X^T [U Ts] with c = 2K,
// SIMPLS/UNIPALS with several responses, n >= p branches) on the f64 matrix cores, X read ONCE.
//
// Tall-skinny GEMM  G (p x c) = Xfm (p x n) . V (n x c)  with M = features, N = right-hand sides,
// K = samples.  v_mfma_f64_16x16x4_f64 wants A[i][k] with the 16 lanes of a group spread over 16
// FEATURES (rows 8n bytes apart) -- un-coalesced if each lane fetched one sample.  Instead every lane
// loads FOUR consecutive samples of its feature (two 16-byte loads): the wave then touches 16 features
// x 128 contiguous bytes = whole cache lines, and the four samples feed four successive MFMAs.  The
// k-index of MFMA #m is therefore the sample set {i0 + 4g + m : g = 0..3}; V is pre-packed once into
// exactly that fragment order (k_pack_v), so a B fragment is one contiguous 512-byte wave load from L2.
// Each wave owns FT = 2 feature tiles (32 features) so a V fragment is reused twice.  No LDS, no
// atomics; accumulators live in 16*NT VGPRs, so occupancy hides the HBM latency.
// Tall data (few features, many samples) has too few feature tiles to fill the chip: the sample range is
// then cut into slices (grid.y), each slice writes its partial G and k_sum_slices adds them in fixed order.
#include <algorithm>

#include "common.h"

namespace mbpls {

typedef double double4_t __attribute__((ext_vector_type(4)));

// Vp[((ks*4 + m)*NT + nt)*64 + lane] = V[(16 nt + (lane&15)) * ldv + 16 ks + 4 (lane>>4) + m]  (0 outside)
__global__ __launch_bounds__(256) void k_pack_v(const double* __restrict__ V, int64_t ldv, int64_t n, int c, int NT,
                                                int64_t total, double* __restrict__ Vp) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  int64_t r = e >> 6;
  const int nt = (int)(r % NT);
  r /= NT;
  const int m = (int)(r & 3);
  const int64_t ks = r >> 2;
  const int64_t i = 16 * ks + 4 * (lane >> 4) + m;
  const int k = 16 * nt + (lane & 15);
  Vp[e] = (i < n && k < c) ? V[(int64_t)k * ldv + i] : 0.0;
}

// feature tiles (of 16) per wave: every wave re-reads the packed V for its features, so more tiles per wave mean
// proportionally less L2 traffic for V (at c = 40, n = 8192 it exceeded the HBM traffic of X with two tiles);
// bounded by the 8 FT NT accumulator VGPRs
__host__ __device__ constexpr int xtv_ft(int NT) { return NT <= 2 ? 4 : (NT == 3 ? 3 : 2); }

// k-steps (16 samples each) of the packed V staged in LDS at a time: the four waves of a workgroup work on different
// feature tiles but on the SAME samples, so they share every B fragment.  Staging cuts the L2 -> L1 traffic of V by
// four (it equalled the HBM traffic of X at c = 40) and turns 12 global loads per k-step and wave into LDS reads.
constexpr int XTV_KC = 4;

// SCALE: X is the RAW matrix and G = alpha Z^T V with z = (x - mean_j) / scale_j applied to the operand in registers
// (each lane owns one feature of a tile, so mean / 1/scale are two values per tile and lane).  Rows beyond n need no
// mask: the packed V is zero there.
//
// Pipeline per workgroup: V chunk c+1 travels global -> registers while chunk c (in LDS) is consumed; the X operand
// of k-step s+1 is loaded (two 16-byte loads per tile and lane) before the MFMAs of k-step s issue.
template <int NT, bool SCALE>
__global__ __launch_bounds__(256, 2) void k_xtv_mfma(const double* __restrict__ X, int64_t ld, int64_t n, int64_t p,
                                                  const double* __restrict__ Vp, int c, double alpha,
                                                  double* __restrict__ G, int64_t ldg, int64_t ks_per_slice,
                                                  int64_t slice_stride, const double* __restrict__ mean,
                                                  const double* __restrict__ rscale, int square) {
  constexpr int FT = xtv_ft(NT);
  constexpr int CH = XTV_KC * 4 * NT * 64;      // doubles of V per chunk
  constexpr int NV2 = CH / 2 / 256;             // double2 per thread and chunk
  static_assert(CH % 512 == 0, "chunk must split evenly over 256 threads x 16 bytes");
  __shared__ double s_v[CH];
  G += (int64_t)blockIdx.y * slice_stride;
  const int64_t ks_begin = (int64_t)blockIdx.y * ks_per_slice;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = lane & 15, g = lane >> 4;
  const int64_t ntiles = (p + 16 * FT - 1) / (16 * FT);
  const int64_t nsuper = (ntiles + 3) / 4;
  const int64_t nk_all = (n + 15) / 16;
  const int64_t nk = std::min(nk_all, ks_begin + ks_per_slice);
  const int64_t nk_full = std::min(ld / 16, nk);   // k-steps whose 16 samples lie inside the row allocation
  const int64_t vtotal2 = nk_all * 4 * NT * 32;    // packed V, in double2 units
  const double2* Vp2 = reinterpret_cast<const double2*>(Vp);
  double2* s_v2 = reinterpret_cast<double2*>(s_v);

  for (int64_t st = blockIdx.x; st < nsuper; st += gridDim.x) {
    const int64_t tile = st * 4 + wave;            // beyond ntiles: clamped rows, computed but never stored
    const double* xrow[FT];
    double mu[SCALE ? FT : 1], rs[SCALE ? FT : 1];
#pragma unroll
    for (int t = 0; t < FT; ++t) {
      int64_t j = tile * 16 * FT + t * 16 + f;
      if (j >= p) j = p - 1;
      xrow[t] = X + j * ld + 4 * g;
      if constexpr (SCALE) {
        mu[t] = mean[j];
        rs[t] = rscale ? rscale[j] : 1.0;           // (NULL: centring only)
      }
    }
    double4_t acc[FT][NT];
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (double4_t){0.0, 0.0, 0.0, 0.0};

    double2 vreg[NV2];
    auto fetch_chunk = [&](int64_t kc0) {          // global -> registers
      const int64_t base2 = kc0 * 4 * NT * 32;
#pragma unroll
      for (int i = 0; i < NV2; ++i) {
        const int64_t e = base2 + i * 256 + tid;
        vreg[i] = e < vtotal2 ? Vp2[e] : make_double2(0.0, 0.0);
      }
    };
    auto park_chunk = [&]() {                      // registers -> LDS
#pragma unroll
      for (int i = 0; i < NV2; ++i) s_v2[i * 256 + tid] = vreg[i];
    };
    auto load_x = [&](double (&x)[FT][4], int64_t ks) {
      if (ks < nk_full) {
#pragma unroll
        for (int t = 0; t < FT; ++t) {
          const double2 lo = *reinterpret_cast<const double2*>(xrow[t] + 16 * ks);
          const double2 hi = *reinterpret_cast<const double2*>(xrow[t] + 16 * ks + 2);
          x[t][0] = lo.x; x[t][1] = lo.y; x[t][2] = hi.x; x[t][3] = hi.y;
        }
      } else {                                     // ragged end of the rows: guarded scalar loads
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int64_t i = 16 * ks + 4 * g + m;
            x[t][m] = (i < n) ? xrow[t][16 * ks + m] : 0.0;
          }
      }
    };
    auto step = [&](double (&cur)[FT][4], double (&nxt)[FT][4], int64_t ks, int lk) {
      if (ks + 1 < nk) load_x(nxt, ks + 1);
      if constexpr (SCALE) {
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int m = 0; m < 4; ++m) cur[t][m] = (cur[t][m] - mu[t]) * rs[t];
        if (square) {                                // wave-uniform: G = alpha (Z o Z)^T V (sums of squares over row sets)
#pragma unroll
          for (int t = 0; t < FT; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) cur[t][m] = cur[t][m] * cur[t][m];
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const double b = s_v[((lk * 4 + m) * NT + nt) * 64 + lane];
#pragma unroll
          for (int t = 0; t < FT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[t][m], b, acc[t][nt], 0, 0, 0);
        }
    };

    double xa[FT][4], xb[FT][4];
    __syncthreads();                               // the previous super tile's last chunk is no longer read
    if (ks_begin < nk) {
      fetch_chunk(ks_begin);
      load_x(xa, ks_begin);
      park_chunk();
    }
    __syncthreads();
    for (int64_t kc0 = ks_begin; kc0 < nk; kc0 += XTV_KC) {
      const bool more = kc0 + XTV_KC < nk;
      if (more) fetch_chunk(kc0 + XTV_KC);
      static_assert(XTV_KC % 2 == 0, "the operand double buffer alternates per k-step");
#pragma unroll
      for (int lk = 0; lk < XTV_KC; lk += 2) {
        if (kc0 + lk < nk) step(xa, xb, kc0 + lk, lk);
        if (kc0 + lk + 1 < nk) step(xb, xa, kc0 + lk + 1, lk + 1);
      }
      __syncthreads();                             // every wave has consumed this chunk
      if (more) {
        park_chunk();
        __syncthreads();
      }
    }
    // D layout (f64 16x16x4): column (rhs) = lane & 15, row (feature) = (lane >> 4) + 4 * reg
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int64_t j = tile * 16 * FT + t * 16 + g + 4 * rg;
          const int k = 16 * nt + f;
          if (j < p && k < c) G[(int64_t)k * ldg + j] = alpha * acc[t][nt][rg];
        }
  }
}

// G[k, j] = sum over slices (fixed order) of part[s][k][j]
__global__ __launch_bounds__(256) void k_sum_slices(const double* __restrict__ part, int S, int64_t slice_stride,
                                                    int64_t p, double* __restrict__ G, int64_t ldg) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (j >= p) return;
  double s = 0.0;
  for (int i = 0; i < S; ++i) s += part[(int64_t)i * slice_stride + (int64_t)k * p + j];
  G[(int64_t)k * ldg + j] = s;
}

// ssq[k * nblocks + b] = sum_{j in block b} G[j, k]^2, fixed-order (one workgroup per (b, k))
__global__ __launch_bounds__(256) void k_block_colssq(const double* __restrict__ G, int64_t ldg,
                                                      const int64_t* __restrict__ blk_off, int nblocks,
                                                      double* __restrict__ ssq) {
  __shared__ double red[2 * 64];
  int buf = 0;
  const int b = blockIdx.x, k = blockIdx.y;
  const double* col = G + (int64_t)k * ldg;
  double s[1] = {0.0};
  for (int64_t j = blk_off[b] + threadIdx.x; j < blk_off[b + 1]; j += 256) s[0] = fma(col[j], col[j], s[0]);
  block_sum<1>(s, red, buf);
  if (threadIdx.x == 0) ssq[(int64_t)k * nblocks + b] = s[0];
}

int xtv_mfma(mbpls_ctx* ctx, const mbpls_mat* m, const double* d_V, int64_t ldv, int c, double alpha, double* d_G,
             int64_t ldg, double* d_ssq, const double* d_mean, const double* d_rscale, int square) {
  const int NT = (c + 15) / 16;
  if (NT > 4) {   // more than 64 right-hand sides: process in groups of 64
    for (int k0 = 0; k0 < c; k0 += 64) {
      const int cb = std::min(64, c - k0);
      MB_TRY(xtv_mfma(ctx, m, d_V + (int64_t)k0 * ldv, ldv, cb, alpha, d_G + (int64_t)k0 * ldg, ldg,
                      d_ssq ? d_ssq + (int64_t)k0 * m->nblocks : nullptr, d_mean, d_rscale, square));
    }
    return MBPLS_OK;
  }
  const int64_t nk = (m->n + 15) / 16;
  const int64_t total = nk * 4 * NT * 64;
  const size_t off_bytes = sizeof(int64_t) * (size_t)(m->nblocks + 1);
  const int ft = xtv_ft(NT);
  const int64_t ntiles = (m->p + 16 * ft - 1) / (16 * ft);
  const int64_t wgs = (ntiles + 3) / 4;
  // sample slices: only when the feature tiles alone leave most of the chip idle; >= 1024 samples per slice
  int S = 1;
  if (wgs < 2 * (int64_t)ctx->cus)
    S = (int)std::max<int64_t>(1, std::min<int64_t>({(2 * (int64_t)ctx->cus + wgs - 1) / wgs, nk / 64, 256}));
  const int64_t ks_per_slice = (nk + S - 1) / S;
  const int64_t slice_stride = (int64_t)c * m->p;
  const size_t part_off = ((sizeof(double) * (size_t)total + off_bytes + 511) / 256) * 256;
  MB_TRY(ensure_ws(ctx, part_off + (S > 1 ? sizeof(double) * (size_t)S * (size_t)slice_stride : 0)));
  double* Vp = (double*)ctx->ws;
  double* part = (double*)((char*)ctx->ws + part_off);
  hipLaunchKernelGGL(k_pack_v, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_V, ldv, m->n, c, NT,
                     total, Vp);
  MB_HIP(hipGetLastError());
  const unsigned grid = (unsigned)std::min<int64_t>(wgs, (int64_t)ctx->cus * 8);
  {
    ProfScope prof(ctx, MBPLS_K_XT_V, 8.0 * (double)m->n * (double)m->p);
#define MB_XTV_(NT_, SC_)                                                                                              \
  hipLaunchKernelGGL((k_xtv_mfma<NT_, SC_>), dim3(grid, (unsigned)S), dim3(256), 0, ctx->stream, (const double*)m->data, \
                     m->ld, m->n, m->p, (const double*)Vp, c, alpha, S > 1 ? part : d_G, S > 1 ? m->p : ldg,           \
                     ks_per_slice, S > 1 ? slice_stride : (int64_t)0, d_mean, d_rscale, square)
#define MB_XTV(NT_)                 \
  do {                              \
    if (d_mean) MB_XTV_(NT_, true); \
    else MB_XTV_(NT_, false);       \
  } while (0)
    if (NT == 1) MB_XTV(1);
    else if (NT == 2) MB_XTV(2);
    else if (NT == 3) MB_XTV(3);
    else MB_XTV(4);
#undef MB_XTV
#undef MB_XTV_
  }
  MB_HIP(hipGetLastError());
  if (S > 1) {
    hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((m->p + 255) / 256), (unsigned)c), dim3(256), 0, ctx->stream,
                       (const double*)part, S, slice_stride, m->p, d_G, ldg);
    MB_HIP(hipGetLastError());
  }
  if (d_ssq) {
    // (the matrix keeps its block offsets on the device: no pageable host-to-device copy in the middle of a pass -- such
    // copies are staged by the runtime and have been seen to execute tens of milliseconds late, small.hip)
    hipLaunchKernelGGL(k_block_colssq, dim3((unsigned)m->nblocks, (unsigned)c), dim3(256), 0, ctx->stream,
                       (const double*)d_G, ldg, (const int64_t*)m->d_block_offsets, m->nblocks, d_ssq);
    MB_HIP(hipGetLastError());
  }
  return MBPLS_OK;
}

}  // namespace mbpls

// one kernel of this translation unit, for mbpls_hip_warmup (its code object is loaded on first use otherwise)
namespace mbpls {
const void* anchor_xtv_mfma() { return (const void*)&k_pack_v; }
}  // namespace mbpls
