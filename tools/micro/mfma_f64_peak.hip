// Pure-MFMA ceiling of v_mfma_f64_16x16x4_f64 on this device (no memory traffic): what clock the chip
// holds under f64 matrix load decides the practical peak the Gram kernel can be compared with.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k(double* out, int iters, double a0, double b0) {
  double4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (double4_t){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 2048);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg : {256, 512, 1024}) {
    const int iters = 200000;
    k<<<wg, 256>>>(d, 1000, 1.0, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<<<wg, 256>>>(d, iters, 1.0, 2.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)wg * 4 * iters * 16 * 2048.0;
    printf("workgroups %4d (x4 waves): %.3f ms  %.2f TFLOP/s f64 MFMA\n", wg, ms, flop / ms / 1e9);
  }
  return 0;
}
