// Does a 1-workgroup chain kernel get a CU while a persistent tile-GEMM grid is running?  (debug harness)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_dense.hpp"
using namespace ekf;
int main() {
  const int n = 6144, ldy = 2048, ld = 6144;
  float *V, *S, *A, *D; int *st, *cnt, *tm;
  hipMalloc(&V, (size_t)n * ldy * 4); hipMalloc(&S, (size_t)n * ld * 4); hipMalloc(&A, 128 * 2048 * 4); hipMalloc(&D, 128 * 128 * 4);
  hipMalloc(&st, 16); hipMalloc(&cnt, 64 * 4);
  hipMemset(V, 0, (size_t)n * ldy * 4); hipMemset(S, 0, (size_t)n * ld * 4); hipMemset(st, 0, 16);
  std::vector<float> h(128 * 2048, 0.f);
  for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j) h[i * 2048 + j] = (i == j ? 140.f : 0.f) + std::cos(0.37f * i * j + i + j);
  for (int i = 0; i < 128; ++i) for (int j = 0; j < i; ++j) h[j * 2048 + i] = h[i * 2048 + j];
  std::vector<int> tmv; for (int i = 0; i < 48; ++i) for (int j = 0; j <= i; ++j) { tmv.push_back(i); tmv.push_back(j); }
  hipMalloc(&tm, tmv.size() * 4); hipMemcpy(tm, tmv.data(), tmv.size() * 4, hipMemcpyHostToDevice);
  hipStream_t sa, sb; int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi);
  {
    uint32_t mask[8]; for (auto& m : mask) m = 0xffffffffu;
    const char* env = getenv("RESERVE");
    int reserve = env ? atoi(env) : 32;      // CUs kept free of the GEMM stream: bit i of the 256-bit mask
    for (int i = 0; i < reserve; ++i) mask[i / 32] &= ~(1u << (i % 32));
    hipError_t e = hipExtStreamCreateWithCUMask(&sb, 8, mask);
    printf("hipExtStreamCreateWithCUMask(reserve=%d): %s\n", reserve, hipGetErrorString(e));
  }
  hipEvent_t e0, e1, g0, g1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&g0); hipEventCreate(&g1);
  for (int wgs : {0, 448, 512}) {
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(cnt, 0, 256);
    hipDeviceSynchronize();
    if (wgs) {
      GemmArgs g{V, ldy, V, ldy, S, ld, 2048, -1.0, 1.0, 2, 0, 0, 0, 0, tm, (int)tmv.size() / 2, cnt};
      hipEventRecord(g0, sb);
      k_gemm_mfma<ROLE_DOWNDATE, false><<<dim3(wgs), 256, 0, sb>>>(g);
      hipEventRecord(g1, sb);
    }
    // give the GEMM a head start, then time the diag kernel on the other stream
    for (volatile int spin = 0; spin < 2000000; ++spin) {}
    hipEventRecord(e0, sa);
    k_chol_diag_packed<><<<1, 1024, 0, sa>>>(A, 2048, D, st);
    hipEventRecord(e1, sa);
    hipDeviceSynchronize();
    float md, mg = 0; hipEventElapsedTime(&md, e0, e1); if (wgs) hipEventElapsedTime(&mg, g0, g1);
    printf("persistent GEMM wgs=%3d (%.3f ms): diag kernel latency %.1f us\n", wgs, mg, md * 1e3);
  }
  return 0;
}
