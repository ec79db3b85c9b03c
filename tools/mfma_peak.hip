// Sustained rate of v_mfma_f32_32x32x2_f32 from registers only (no memory traffic): the practical ceiling
// of the f32 matrix pipe at the clock the part holds under this load.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k_peak(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d;
  hipMalloc(&d, 4096 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wgs : {256, 512, 1024, 2048}) {
    for (int iters : {2000, 20000}) {
      k_peak<<<wgs, 256>>>(d, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k_peak<<<wgs, 256>>>(d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)wgs * 4 * iters * 32 * 4096.0;
      printf("wgs %5d iters %6d  %.3f ms  %.1f TFLOP/s\n", wgs, iters, ms, flop / ms / 1e9);
    }
  }
  // sustained: 1 s of back-to-back launches, rate of each 100 ms window
  for (int rep = 0; rep < 12; ++rep) {
    hipEventRecord(e0);
    k_peak<<<1024, 256>>>(d, 20000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("sustained window %2d: %.1f TFLOP/s\n", rep, 1024.0 * 4 * 20000 * 32 * 4096.0 / ms / 1e9);
  }
  return 0;
}
