// OUT = X W for MANY right-hand sides (c > 8: transform with K components, KERNEL block scores with c = K per
// block, the masked pair sums of sparse transform, predict with many responses) on the f64 matrix cores,
// X read ONCE for up to 64 right-hand sides (the row-slab kernel k_xw takes 8 per pass).
//
// Skinny GEMM  OUT (n x c) = X (n x p) . W (p x c), reduction over FEATURES.  v_mfma_f64_16x16x4_f64 with
//   A[i][k] = X[sample i, feature j+k]   lane (i = lane & 15, k = lane >> 4)
//   B[k][r] = W[feature j+k, rhs r]      lane (r = lane & 15, k = lane >> 4)
// In the feature-major layout a lane loads TWO consecutive samples of its feature (one 16-byte load; a wave
// covers 32 samples x 4 features = 4 x 256 contiguous bytes): the even samples feed one MFMA, the odd ones a
// second, so a wave owns a 32-sample x (16 NT)-rhs output tile in 16 NT accumulator VGPR pairs.  W is pre-packed
// once per call into fragment order (a B fragment = one contiguous 512-byte wave load from L2; the eight waves
// of a workgroup share it through L1); standardisation on the fly ((x - mean_j)/scale_j, exactly as k_xw) and
// the missing-value modes are applied to the A operand in registers.  A workgroup = 8 waves = 256 samples of one
// feature chunk; the per-chunk partial tiles are summed in fixed order by k_reduce_xw (gemv.hip) -- no atomics.
#include <algorithm>

#include "common.h"

namespace mbpls {

typedef double double4_t __attribute__((ext_vector_type(4)));

// Wp[(g * NT + nt) * 64 + lane] = W[(16 nt + (lane & 15)) * ldw + 4 g + (lane >> 4)]   (0 outside p / c)
__global__ __launch_bounds__(256) void k_pack_w(const double* __restrict__ W, int64_t ldw, int64_t p, int c, int NT,
                                                int64_t total, double* __restrict__ Wp) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  int64_t r = e >> 6;
  const int nt = (int)(r % NT);
  const int64_t g = r / NT;
  const int64_t j = 4 * g + (lane >> 4);
  const int k = 16 * nt + (lane & 15);
  Wp[e] = (j < p && k < c) ? W[(int64_t)k * ldw + j] : 0.0;
}

// NANM as in k_xw: 0 plain, 1 NaN contributes nothing, 2 presence indicator in place of x
template <int NT, bool SCALE, int NANM>
__global__ __launch_bounds__(512) void k_xw_mfma(const double* __restrict__ X, int64_t ld, int64_t n,
                                                 const int64_t* __restrict__ c0, const int64_t* __restrict__ c1,
                                                 const double* __restrict__ Wp,
                                                 const double2* __restrict__ ms /* (mean_j, 1/scale_j) */,
                                                 double* __restrict__ part /* [nchunks][16 NT][n] */) {
  // launch order: the row slabs of one column chunk are consecutive workgroups -- they run side by side and read
  // neighbouring 2 KB pieces of the same columns (DRAM page locality), instead of 16 distant chunks at a time
  const int chunk = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int64_t j0 = c0[chunk], j1 = c1[chunk];
  const int64_t i0 = (int64_t)blockIdx.x * 256 + wave * 32;   // first sample of this wave's tile
  if (i0 >= n) return;                                         // wave-uniform
  const int64_t s0 = i0 + 2 * li;                              // this lane's sample pair
  const bool in_ld = s0 < ld, ok0 = s0 < n, ok1 = s0 + 1 < n;
  const double* xcol = X + (in_ld ? s0 : 0);

  double4_t acc[2][NT];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[s][nt] = (double4_t){0.0, 0.0, 0.0, 0.0};

  // feature groups are aligned to absolute multiples of 4 (the packing of W); features of the first / last group
  // outside [j0, j1) belong to a neighbouring chunk and are masked out
  const int64_t g0 = j0 >> 2, g1 = (j1 + 3) >> 2;
  // UG feature groups per round: all their X loads (and scaling pairs) are issued before the first MFMA -- with one
  // group per iteration a wave had a single 1 KB load in flight and the kernel was latency-bound (5.7 ms where a
  // plain sweep takes 4.7)
  constexpr int UG = NT <= 2 ? 8 : 4;
  for (int64_t g = g0; g < g1; g += UG) {
    double2 xr[UG], mr[UG];
    bool vr[UG];
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      const int64_t j = 4 * (g + u) + lk;
      vr[u] = (g + u < g1) && j >= j0 && j < j1;
      const int64_t jc = vr[u] ? j : j0;
      xr[u] = *reinterpret_cast<const double2*>(xcol + jc * ld);
      if constexpr (SCALE) mr[u] = ms[jc];      // (mean_j, 1/scale_j): one 16-byte load
    }
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      if (g + u >= g1) break;                   // wave-uniform
      double2 x = xr[u];
      const bool vj = vr[u];
      if constexpr (SCALE) {
        x.x = (x.x - mr[u].x) * mr[u].y;
        x.y = (x.y - mr[u].x) * mr[u].y;
      }
      if constexpr (NANM == 1) {
        x.x = (x.x == x.x) ? x.x : 0.0;
        x.y = (x.y == x.y) ? x.y : 0.0;
      } else if constexpr (NANM == 2) {
        x.x = (x.x == x.x) ? 1.0 : 0.0;
        x.y = (x.y == x.y) ? 1.0 : 0.0;
      }
      const double a0 = (vj && ok0) ? x.x : 0.0, a1 = (vj && ok1) ? x.y : 0.0;
      const double* wp = Wp + ((g + u) * NT) * 64 + lane;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const double b = wp[nt * 64];
        acc[0][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][nt], 0, 0, 0);
      }
    }
  }
  // D layout (f64 16x16x4): column (rhs) = lane & 15, row = (lane >> 4) + 4 * reg; row r of sub-tile s is sample
  // i0 + 2 r + s
  double* out = part + (int64_t)chunk * (16 * NT) * n;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int64_t smp = i0 + 2 * (lk + 4 * rg) + s;
        if (smp < n) out[(int64_t)(16 * nt + li) * n + smp] = acc[s][nt][rg];
      }
}

// gemv.hip
void launch_reduce_xw(mbpls_ctx* ctx, const mbpls_mat* m, const double* part, int CB, int cb, int per_block,
                      double* d_out, int64_t ldo, int c_total, int k0);

// ms[j] = (mean_j, 1/scale_j)
__global__ __launch_bounds__(256) void k_pack_mean_recip(int64_t p, const double* __restrict__ mean,
                                                         const double* __restrict__ scale, double2* __restrict__ ms) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < p) ms[j] = make_double2(mean[j], 1.0 / scale[j]);
}

int xw_mfma(mbpls_ctx* ctx, const mbpls_mat* m, const double* d_W, int64_t ldw, int c, int per_block,
            const double* d_mean, const double* d_scale, int nanm, double* d_out, int64_t ldo) {
  const ChunkTable& ct = m->coarse;
  for (int k0 = 0; k0 < c; k0 += 64) {
    const int cb = std::min(64, c - k0);
    const int NT = (cb + 15) / 16;
    const int64_t ngroups = (m->p + 3) / 4;
    const int64_t total = ngroups * NT * 64;
    const size_t ms_off = ((sizeof(double) * (size_t)total + 255) / 256) * 256;
    const size_t part_off = ms_off + (d_mean ? ((sizeof(double2) * (size_t)m->p + 255) / 256) * 256 : 0);
    MB_TRY(ensure_ws(ctx, part_off + sizeof(double) * (size_t)ct.nchunks * 16 * NT * (size_t)m->n));
    double* Wp = (double*)ctx->ws;
    double2* ms = d_mean ? (double2*)((char*)ctx->ws + ms_off) : nullptr;
    double* part = (double*)((char*)ctx->ws + part_off);
    if (ms) {
      hipLaunchKernelGGL(k_pack_mean_recip, dim3((unsigned)((m->p + 255) / 256)), dim3(256), 0, ctx->stream, m->p, d_mean,
                         d_scale, ms);
      MB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_W + (int64_t)k0 * ldw, ldw, m->p, cb, NT, total, Wp);
    MB_HIP(hipGetLastError());
    dim3 grid((unsigned)((m->n + 255) / 256), (unsigned)ct.nchunks);
    {
      ProfScope prof(ctx, MBPLS_K_X_W, 8.0 * (double)m->n * (double)m->p);
#define MB_XWM(NT_, SC_, NM_)                                                                                       \
  hipLaunchKernelGGL((k_xw_mfma<NT_, SC_, NM_>), grid, dim3(512), 0, ctx->stream, (const double*)m->data, m->ld,    \
                     m->n, (const int64_t*)ct.d_c0, (const int64_t*)ct.d_c1, (const double*)Wp, (const double2*)ms, part)
#define MB_XWM_NT(SC_, NM_)           \
  do {                                \
    if (NT == 1) MB_XWM(1, SC_, NM_); \
    else if (NT == 2) MB_XWM(2, SC_, NM_); \
    else if (NT == 3) MB_XWM(3, SC_, NM_); \
    else MB_XWM(4, SC_, NM_);         \
  } while (0)
      if (nanm == 2) MB_XWM_NT(false, 2);
      else if (nanm == 1 && d_mean) MB_XWM_NT(true, 1);
      else if (nanm == 1) MB_XWM_NT(false, 1);
      else if (d_mean) MB_XWM_NT(true, 0);
      else MB_XWM_NT(false, 0);
#undef MB_XWM_NT
#undef MB_XWM
    }
    MB_HIP(hipGetLastError());
    launch_reduce_xw(ctx, m, part, 16 * NT, cb, per_block, d_out, ldo, c, k0);
    MB_HIP(hipGetLastError());
  }
  return MBPLS_OK;
}

}  // namespace mbpls

// one kernel of this translation unit, for mbpls_hip_warmup (its code object is loaded on first use otherwise)
namespace mbpls {
const void* anchor_xw_mfma() { return (const void*)&k_pack_w; }
}  // namespace mbpls
