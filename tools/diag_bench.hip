// Standalone timing + cross-check of the diagonal-block kernel (debug harness; not part of the library):
// k_chol_diag_packed against a host fp64 factorisation (L and L^-1), its time, its phases alone, and the s_memtime
// stamps of wave 0 per 16-column block.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o diag_bench diag_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define EKF_DIAG_STAMPS 1
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_dense.hpp"
using namespace ekf;
template <typename F> float run(F launch, float* dA, float* dA0, int ld, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float tot = 0;
  for (int r = 0; r < reps; ++r) {
    hipMemcpy(dA, dA0, (size_t)128 * ld * 4, hipMemcpyDeviceToDevice);
    hipEventRecord(a);
    launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (r) tot += ms;
  }
  return tot / (reps - 1) * 1e3f;
}
// back-to-back launches on one stream (what the chain sees: no event pair per kernel)
template <typename F> float run_b2b(F launch, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3f;
}
int main() {
  const int ld = 2048;
  for (int real16 : {8, 5, 1}) {
    const int nreal = real16 * 16;
    std::vector<float> h((size_t)128 * ld, 0.f);
    for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j)
      h[(size_t)i * ld + j] = (i < nreal && j < nreal) ? ((i == j ? 140.f : 0.f) + std::cos(0.37f * i * j + i + j)) : (i == j ? 1.f : 0.f);
    for (int i = 0; i < 128; ++i) for (int j = 0; j < i; ++j) h[(size_t)j * ld + i] = h[(size_t)i * ld + j];
    // host reference in double
    std::vector<double> L(128 * 128, 0.0), Li(128 * 128, 0.0);
    for (int j = 0; j < 128; ++j) {
      double d = h[(size_t)j * ld + j];
      for (int k = 0; k < j; ++k) d -= L[j * 128 + k] * L[j * 128 + k];
      L[j * 128 + j] = std::sqrt(d);
      for (int i = j + 1; i < 128; ++i) {
        double v = h[(size_t)i * ld + j];
        for (int k = 0; k < j; ++k) v -= L[i * 128 + k] * L[j * 128 + k];
        L[i * 128 + j] = v / L[j * 128 + j];
      }
    }
    for (int c = 0; c < 128; ++c)
      for (int i = c; i < 128; ++i) {
        double v = (i == c) ? 1.0 : 0.0;
        for (int k = c; k < i; ++k) v -= L[i * 128 + k] * Li[k * 128 + c];
        Li[i * 128 + c] = v / L[i * 128 + i];
      }
    float *dA, *dA0, *dD; int* dst;
    hipMalloc(&dA, h.size() * 4); hipMalloc(&dA0, h.size() * 4); hipMalloc(&dD, 128 * 128 * 4); hipMalloc(&dst, 16);
    hipMemcpy(dA0, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(dst, 0, 16);
    std::vector<float> outL((size_t)128 * ld), outD(128 * 128);
    for (int which = 0; which < 2; ++which) {
      hipMemcpy(dA, dA0, h.size() * 4, hipMemcpyDeviceToDevice);
      hipMemset(dD, 0xff, 128 * 128 * 4);
      if (which == 0) k_chol_diag_packed<><<<1, 1024>>>(dA, ld, dD, dst, real16);
      else continue;
      hipDeviceSynchronize();
      hipMemcpy(outL.data(), dA, h.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(outD.data(), dD, 128 * 128 * 4, hipMemcpyDeviceToHost);
      double eL = 0, eD = 0, nL = 0, nD = 0; int st[4];
      hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost);
      for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j) {
        const double l = outL[(size_t)i * ld + j], d = outD[i * 128 + j];
        eL += (l - L[i * 128 + j]) * (l - L[i * 128 + j]); nL += L[i * 128 + j] * L[i * 128 + j];
        eD += (d - Li[i * 128 + j]) * (d - Li[i * 128 + j]); nD += Li[i * 128 + j] * Li[i * 128 + j];
      }
      if (real16 == 8 && which == 0) {      // stamps of THIS launch (every phase on): wave 0 of the workgroup
        unsigned long long st[64];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(ekf::ekf_diag_stamps), sizeof(st));
        for (int b = 0; b < 7; ++b)
          printf("block %d (wave 0, cycles): own update + factor of block %d: %llu  then waiting for the helper waves' tiles: %llu  panel + barriers: %llu\n", b, b + 1,
                 st[8 + 4 * b + 2] - st[8 + 4 * b + 1], st[8 + 4 * b + 3] - st[8 + 4 * b + 2],
                 b < 6 ? st[8 + 4 * (b + 1)] - st[8 + 4 * b + 3] : 0ull /* no stamp after the last block */);
        printf("diag_factor16 (block 1, wave 0) cycles: prologue incl. the block's own rank-16 update %llu  16 columns %llu  epilogue %llu\n", st[1] - st[0], st[2] - st[1], st[3] - st[2]);
      }
      printf("real16=%d %-22s rel|L - L64| %.2e  rel|Linv - Linv64| %.2e  status %d\n", real16,
             "k_chol_diag_packed", std::sqrt(eL / nL), std::sqrt(eD / nD), st[0]);
    }
    if (real16 == 8) {
      printf("k_chol_diag_packed  event pair %.2f us   back to back %.2f us\n",
             run([&] { k_chol_diag_packed<><<<1, 1024>>>(dA, ld, dD, dst, 8); }, dA, dA0, ld, 50),
             run_b2b([&] { k_chol_diag_packed<><<<1, 1024>>>(dA0, ld, dD, dst, 8); }, 200));
      printf("phases alone, back to back (us; the harness build carries s_memtime stamps + waits: ~2 us): none %.2f  factor only %.2f  panel only %.2f  tile updates only %.2f\n",
             run_b2b([&] { k_chol_diag_packed<0><<<1, 1024>>>(dA0, ld, dD, dst, 8); }, 200),
             run_b2b([&] { k_chol_diag_packed<1><<<1, 1024>>>(dA0, ld, dD, dst, 8); }, 200),
             run_b2b([&] { k_chol_diag_packed<2><<<1, 1024>>>(dA0, ld, dD, dst, 8); }, 200),
             run_b2b([&] { k_chol_diag_packed<4><<<1, 1024>>>(dA0, ld, dD, dst, 8); }, 200));
    }
    hipFree(dA); hipFree(dA0); hipFree(dD); hipFree(dst);
  }
  return 0;
}
