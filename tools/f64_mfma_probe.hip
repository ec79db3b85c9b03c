// Layout probe of v_mfma_f64_16x16x4_f64 and check of k_gemm_mfma_f64 against the host (debug harness).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_dense.hpp"
using namespace ekf;
// lane l supplies A[i_of(l)][k_of(l)] and B[k_of(l)][j_of(l)] under the ASSUMED mapping; D is read back raw
__global__ void probe(const double* A, const double* B, double* Draw) {   // A 16x4, B 4x16 row-major
  const int l = threadIdx.x;
  const int i = l & 15, k = l >> 4;
  f64x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[i * 4 + k], B[k * 16 + i], acc, 0, 0, 0);
  for (int e = 0; e < 4; ++e) Draw[l * 4 + e] = acc[e];
}
int main() {
  std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
  for (int i = 0; i < 64; ++i) { A[i] = (rand() % 1009) * 0.5 + 1; B[i] = (rand() % 997) * 0.25 + 3; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  // find for lane l, reg e which (i, j) it holds
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    const double v = D[l * 4 + e];
    if (v != ref[(4 * e + (l >> 4)) * 16 + (l & 15)]) ok1 = 0;          // row 4 e + l / 16, col l % 16
    if (v != ref[(l & 15) * 16 + 4 * (l >> 4) + e]) ok2 = 0;            // row l % 16, col 4 (l/16) + e
  }
  printf("D layout: row=4*e+l/16,col=l%%16: %d    row=l%%16,col=4*(l/16)+e: %d\n", ok1, ok2);
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) for (int e = 0; e < 4; ++e) {
    int cnt = 0, fi = -1, fj = -1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (ref[i * 16 + j] == D[l * 4 + e]) { ++cnt; fi = i; fj = j; }
    printf("lane %2d reg %d -> value %8.1f matches %d cell(s), last (%d, %d)\n", l, e, D[l * 4 + e], cnt, fi, fj);
  }
  // full kernel check, NT and NN
  const int n = 128, K = 96, ld = 128;
  std::vector<double> hA((size_t)n * ld), hB((size_t)n * ld), hC((size_t)n * ld), hR((size_t)n * ld);
  for (auto& x : hA) x = (rand() % 2001 - 1000) * 1e-3; for (auto& x : hB) x = (rand() % 2001 - 1000) * 1e-3;
  for (auto& x : hC) x = (rand() % 2001 - 1000) * 1e-3;
  double *gA, *gB, *gC; hipMalloc(&gA, hA.size() * 8); hipMalloc(&gB, hB.size() * 8); hipMalloc(&gC, hC.size() * 8);
  hipMemcpy(gA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); hipMemcpy(gB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
  for (int bt = 0; bt < 2; ++bt) {
    hipMemcpy(gC, hC.data(), hC.size() * 8, hipMemcpyHostToDevice);
    GemmArgs g{gA, ld, gB, ld, gC, ld, K, -1.0, 1.0, 0, 0, 0, 0, 0, nullptr, 0, nullptr};
    if (bt) k_gemm_mfma_f64<ROLE_GAIN, true><<<dim3(n / 64, n / 64), 256>>>(g);
    else k_gemm_mfma_f64<ROLE_GAIN, false><<<dim3(n / 64, n / 64), 256>>>(g);
    hipMemcpy(hR.data(), gC, hR.size() * 8, hipMemcpyDeviceToHost);
    double emax = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
      double s = hC[(size_t)i * ld + j];
      for (int k = 0; k < K; ++k) s -= hA[(size_t)i * ld + k] * (bt ? hB[(size_t)k * ld + j] : hB[(size_t)j * ld + k]);
      emax = std::fmax(emax, std::fabs(s - hR[(size_t)i * ld + j]));
    }
    printf("%s: max error %.3e\n", bt ? "NN" : "NT", emax);
  }
  return 0;
}
