// Does the 256 MiB Infinity Cache turn the second read of a two-phase pass (reduce a column, then re-read it to store
// it deflated) into an on-die read, if the two phases run over BATCHES of columns instead of the whole matrix?
// A 2 GiB buffer; phase A reads a batch (sum), phase B reads the batch again and writes it back (x*a+b); batches of
// 2048 (no reuse), 256, 128, 96, 64, 32, 16 MiB; phase A's loads default or nontemporal.  Prints the time of the whole
// sweep and the "algorithmic" rate (one read + one write of the buffer / time).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mall_reuse tools/micro/mall_reuse.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void k_a(const v2d* __restrict__ x, int64_t n2, double* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    const v2d q = NT ? __builtin_nontemporal_load(x + i) : x[i];
    s += q.x + q.y;
  }
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_b(v2d* __restrict__ x, int64_t n2, double a, double b) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    v2d q = NTL ? __builtin_nontemporal_load(x + i) : x[i];
    q.x = q.x * a + b;
    q.y = q.y * a + b;
    if (NTS) __builtin_nontemporal_store(q, x + i);
    else x[i] = q;
  }
}

int main() {
  const int64_t bytes = 2ll << 30;
  v2d* d;
  double* out;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  hipMalloc(&out, sizeof(double) * 4 * 65536);
  hipMemset(d, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * 8;
  for (int variant = 0; variant < 4; ++variant) {
    const bool a_nt = variant & 1, b_nt = variant & 2;
    printf("phase A loads %s, phase B loads+stores %s\n", a_nt ? "nontemporal" : "default", b_nt ? "nontemporal" : "default");
    for (int64_t mb : {2048, 512, 256, 192, 128, 96, 64, 32, 16}) {
      const int64_t bb = mb << 20, nb = bytes / bb, n2 = bb / 16;
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int64_t k = 0; k < nb; ++k) {
          v2d* x = d + k * n2;
          if (a_nt) k_a<true><<<grid, 256>>>(x, n2, out);
          else k_a<false><<<grid, 256>>>(x, n2, out);
          if (b_nt) k_b<true, true><<<grid, 256>>>(x, n2, 1.0, 0.0);
          else k_b<false, false><<<grid, 256>>>(x, n2, 1.0, 0.0);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("  batch %5lld MiB x %4lld: %7.3f ms   %6.0f GB/s algorithmic (1R+1W)   %6.0f GB/s if all 2R+1W came from HBM\n",
             (long long)mb, (long long)nb, best, 2.0 * bytes / best / 1e6, 3.0 * bytes / best / 1e6);
    }
  }
  // two streams: phase A of batch k + 1 runs beside phase B of batch k (HBM reads beside HBM writes, B's reads on-die)
  {
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    std::vector<hipEvent_t> ea(2048), eb(2048);
    for (auto& e : ea) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto& e : eb) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    printf("two streams: phase A of batch k+1 beside phase B of batch k (A default loads, B nontemporal)\n");
    for (int64_t mb : {256, 192, 128, 96, 64, 48, 32, 16}) {
      const int64_t bb = mb << 20, nb = bytes / bb, n2 = bb / 16;
      for (int g : {grid, grid / 2}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
          hipDeviceSynchronize();
          hipEventRecord(e0, s1);
          hipStreamWaitEvent(s2, e0, 0);
          for (int64_t k = 0; k < nb; ++k) {
            // A(k) on s2 must not run ahead of B(k - 2): at most two batches in the cache
            if (k >= 2) hipStreamWaitEvent(s2, eb[k - 2], 0);
            k_a<false><<<g, 256, 0, s2>>>(d + k * n2, n2, out);
            hipEventRecord(ea[k], s2);
            hipStreamWaitEvent(s1, ea[k], 0);
            k_b<true, true><<<g, 256, 0, s1>>>(d + k * n2, n2, 1.0, 0.0);
            hipEventRecord(eb[k], s1);
          }
          hipEventRecord(e1, s1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        printf("  batch %5lld MiB x %4lld, %4d workgroups per kernel: %7.3f ms   %6.0f GB/s algorithmic (1R+1W)\n",
               (long long)mb, (long long)nb, g, best, 2.0 * bytes / best / 1e6);
      }
    }
  }
  // reference points: phase A alone and phase B alone over the whole buffer
  for (int w = 0; w < 2; ++w) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (w == 0) k_a<true><<<grid, 256>>>(d, bytes / 16, out);
      else k_b<true, true><<<grid, 256>>>(d, bytes / 16, 1.0, 0.0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s alone over 2 GiB: %.3f ms  %.0f GB/s of traffic\n", w ? "phase B (1R+1W)" : "phase A (1R)", best,
           (w ? 2.0 : 1.0) * bytes / best / 1e6);
  }
  return 0;
}
