// Small dense helpers on p-length device data (p x K slots, K <= 32), so that the end-of-fit
// algebra  R_ = W (P^T W)^+,  beta_ = R_ V^T,  W_b = W~_b / ||w~_b||  (mbpls.py:986-989, :569-570,
// :733-738) never moves p x K arrays through the host more than the one fetch of each attribute.
#include <algorithm>

#include "common.h"

namespace mbpls {

constexpr int CMAX = 32;

// partial[wg][i][j] = sum over this workgroup's rows of A[r,i] * B[r,j]; TI A-columns per pass over the TJ B-columns
// (<2, 32>: any tile; <4, 16>: up to 16 B-columns -- K = 10 reads B three times instead of five: 222 -> 90 us at p = 10^6).
template <int TI, int TJ>
__global__ __launch_bounds__(256) void k_cross_partial(int64_t p, const double* __restrict__ A, int64_t lda, int c1,
                                                       const double* __restrict__ B, int64_t ldb, int c2,
                                                       double* __restrict__ partial) {
  __shared__ double red[2 * 64];
  int buf = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int i0 = 0; i0 < c1; i0 += TI) {
    double acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p; r += stride) {
      double a[TI];
#pragma unroll
      for (int i = 0; i < TI; ++i) a[i] = (i0 + i < c1) ? A[(int64_t)(i0 + i) * lda + r] : 0.0;
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        if (j < c2) {
          const double b = B[(int64_t)j * ldb + r];
#pragma unroll
          for (int i = 0; i < TI; ++i) acc[i][j] = fma(a[i], b, acc[i][j]);
        }
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j)
      if (j < c2) {
        double v[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) v[i] = acc[i][j];
        block_sum<TI>(v, red, buf);
        if (threadIdx.x == 0) {
          double* out = partial + (int64_t)blockIdx.x * c1 * c2;
#pragma unroll
          for (int i = 0; i < TI; ++i)
            if (i0 + i < c1) out[(i0 + i) * c2 + j] = v[i];
        }
      }
  }
}

static void launch_cross_partial(mbpls_ctx* ctx, int nwg, int64_t p, const double* A, int64_t lda, int c1, const double* B,
                                 int64_t ldb, int c2, double* partial) {
  if (c2 <= 16)
    hipLaunchKernelGGL((k_cross_partial<4, 16>), dim3(nwg), dim3(256), 0, ctx->stream, p, A, lda, c1, B, ldb, c2, partial);
  else
    hipLaunchKernelGGL((k_cross_partial<2, CMAX>), dim3(nwg), dim3(256), 0, ctx->stream, p, A, lda, c1, B, ldb, c2, partial);
}

__global__ __launch_bounds__(64) void k_cross_final(const double* __restrict__ partial, int nwg, int cc,
                                                    double* __restrict__ out) {
  const int e = blockIdx.x;  // one wave per output element
  double s = 0.0;
  for (int w = threadIdx.x; w < nwg; w += 64) s += partial[(int64_t)w * cc + e];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[e] = s;
}

// C[r, j] (+)= sum_i A[r, i] * M[i, j]   (M: c1 x c2 row-major in device memory)
__global__ __launch_bounds__(256) void k_right_multiply(int64_t p, const double* __restrict__ A, int64_t lda, int c1,
                                                        const double* __restrict__ M, int c2,
                                                        double* __restrict__ Cm, int64_t ldc, int accumulate) {
  __shared__ double sM[CMAX * CMAX];
  for (int e = threadIdx.x; e < c1 * c2; e += blockDim.x) sM[e] = M[e];
  __syncthreads();
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p; r += (int64_t)gridDim.x * blockDim.x) {
    double a[CMAX];
#pragma unroll
    for (int i = 0; i < CMAX; ++i) a[i] = (i < c1) ? A[(int64_t)i * lda + r] : 0.0;
    for (int j = 0; j < c2; ++j) {
      double s = accumulate ? Cm[(int64_t)j * ldc + r] : 0.0;
#pragma unroll
      for (int i = 0; i < CMAX; ++i)
        if (i < c1) s = fma(a[i], sM[i * c2 + j], s);
      Cm[(int64_t)j * ldc + r] = s;
    }
  }
}

}  // namespace mbpls

namespace mbpls {
__global__ __launch_bounds__(256) void k_scale_rows(int64_t p, const double* __restrict__ A, int64_t lda,
                                                    const double* __restrict__ r, double* __restrict__ C,
                                                    int64_t ldc) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < p) C[(int64_t)blockIdx.y * ldc + j] = A[(int64_t)blockIdx.y * lda + j] * r[j];
}
}  // namespace mbpls

using namespace mbpls;

extern "C" {

int mbpls_hip_cross_product_dev(mbpls_ctx* ctx, int64_t p, const double* d_A, int64_t lda, int c1, const double* d_B,
                                int64_t ldb, int c2, double* d_out) {
  MB_ARG(ctx && d_A && d_B && d_out, "NULL argument");
  MB_ARG(p >= 1 && c1 >= 1 && c2 >= 1 && c1 <= CMAX && c2 <= CMAX, "need 1 <= c1, c2 <= %d", CMAX);
  MB_ARG(lda >= p && ldb >= p, "leading dimension < p");
  const int nwg = (int)std::min<int64_t>((p + 255) / 256, (int64_t)ctx->cus * 4);
  const int cc = c1 * c2;
  MB_TRY(ensure_ws(ctx, sizeof(double) * ((size_t)nwg * cc)));
  double* partial = (double*)ctx->ws;
  launch_cross_partial(ctx, nwg, p, d_A, lda, c1, d_B, ldb, c2, partial);
  MB_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_cross_final, dim3(cc), dim3(64), 0, ctx->stream, (const double*)partial, nwg, cc, d_out);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

int mbpls_hip_cross_product(mbpls_ctx* ctx, int64_t p, const double* d_A, int64_t lda, int c1, const double* d_B,
                            int64_t ldb, int c2, double* h_out) {
  MB_ARG(ctx && d_A && d_B && h_out, "NULL argument");
  MB_ARG(p >= 1 && c1 >= 1 && c2 >= 1, "need p, c1, c2 >= 1");
  MB_ARG(lda >= p && ldb >= p, "leading dimension < p");
  const int nwg = (int)std::min<int64_t>((p + 255) / 256, (int64_t)ctx->cus * 4);
  // any number of columns: CMAX x CMAX tiles of the result, one pair of launches each
  std::vector<double> tile((size_t)CMAX * CMAX);
  for (int i0 = 0; i0 < c1; i0 += CMAX)
    for (int j0 = 0; j0 < c2; j0 += CMAX) {
      const int t1 = std::min(CMAX, c1 - i0), t2 = std::min(CMAX, c2 - j0), cc = t1 * t2;
      MB_TRY(ensure_ws(ctx, sizeof(double) * ((size_t)nwg * cc + cc)));
      double* partial = (double*)ctx->ws;
      double* d_out = partial + (size_t)nwg * cc;
      launch_cross_partial(ctx, nwg, p, d_A + (int64_t)i0 * lda, lda, t1, d_B + (int64_t)j0 * ldb, ldb, t2, partial);
      MB_HIP(hipGetLastError());
      hipLaunchKernelGGL(k_cross_final, dim3(cc), dim3(64), 0, ctx->stream, (const double*)partial, nwg, cc, d_out);
      MB_HIP(hipGetLastError());
      double* dst = (c1 <= CMAX && c2 <= CMAX) ? h_out : tile.data();
      MB_HIP(hipMemcpyAsync(dst, d_out, sizeof(double) * cc, hipMemcpyDeviceToHost, ctx->stream));
      MB_HIP(hipStreamSynchronize(ctx->stream));
      if (dst != h_out)
        for (int i = 0; i < t1; ++i)
          for (int j = 0; j < t2; ++j) h_out[(size_t)(i0 + i) * c2 + (j0 + j)] = tile[(size_t)i * t2 + j];
    }
  return MBPLS_OK;
}

int mbpls_hip_right_multiply(mbpls_ctx* ctx, int64_t p, const double* d_A, int64_t lda, int c1, const double* h_M,
                             int c2, double* d_C, int64_t ldc) {
  MB_ARG(ctx && d_A && h_M && d_C, "NULL argument");
  MB_ARG(p >= 1 && c1 >= 1 && c2 >= 1, "need p, c1, c2 >= 1");
  MB_ARG(lda >= p && ldc >= p, "leading dimension < p");
  // M travels through a small dedicated device buffer (not the workspace: calls may be queued back to back; the
  // copies are ordered on the stream behind the kernels that still read a slot of the ring)
  constexpr int RING = 16;
  if (!ctx->small_m) MB_HIP(hipMalloc(&ctx->small_m, sizeof(double) * CMAX * CMAX * RING));
  const int nwg = (int)std::min<int64_t>((p + 255) / 256, (int64_t)ctx->cus * 8);
  // any c1, c2: CMAX x CMAX tiles of M; the tiles along c1 accumulate into C (so with c1 > CMAX the output must not
  // alias the input columns)
  const bool tiled = c1 > CMAX || c2 > CMAX;
  std::vector<std::vector<double>> tiles;     // packed sub-blocks of M: alive until the copies have run (sync below)
  for (int j0 = 0; j0 < c2; j0 += CMAX)
    for (int i0 = 0; i0 < c1; i0 += CMAX) {
      const int t1 = std::min(CMAX, c1 - i0), t2 = std::min(CMAX, c2 - j0);
      const double* hM = h_M;
      if (tiled) {
        tiles.emplace_back((size_t)t1 * t2);
        std::vector<double>& tile = tiles.back();
        for (int i = 0; i < t1; ++i)
          for (int j = 0; j < t2; ++j) tile[(size_t)i * t2 + j] = h_M[(size_t)(i0 + i) * c2 + (j0 + j)];
        hM = tile.data();
      }
      double* dM = (double*)ctx->small_m + (size_t)(ctx->small_m_slot++ % RING) * CMAX * CMAX;
      MB_HIP(hipMemcpyAsync(dM, hM, sizeof(double) * t1 * t2, hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(k_right_multiply, dim3(nwg), dim3(256), 0, ctx->stream, p, d_A + (int64_t)i0 * lda, lda, t1,
                         (const double*)dM, t2, d_C + (int64_t)j0 * ldc, ldc, i0 > 0 ? 1 : 0);
      MB_HIP(hipGetLastError());
    }
  if (tiled) MB_HIP(hipStreamSynchronize(ctx->stream));
  return MBPLS_OK;
}

int mbpls_hip_scale_rows(mbpls_ctx* ctx, int64_t p, int c, const double* d_A, int64_t lda, const double* d_r,
                         double* d_C, int64_t ldc) {
  MB_ARG(ctx && d_A && d_r && d_C, "NULL argument");
  MB_ARG(p >= 0 && c >= 1 && lda >= p && ldc >= p, "bad shape");
  if (p == 0) return MBPLS_OK;
  hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((p + 255) / 256), (unsigned)c), dim3(256), 0, ctx->stream, p, d_A,
                     lda, d_r, d_C, ldc);
  MB_HIP(hipGetLastError());
  return MBPLS_OK;
}

}  // extern "C"
