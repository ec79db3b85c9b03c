// Where do the cycles of a K step go?  Diagnostic build of the downdate tile kernel with s_memtime stamps INSIDE the K loop
// (wave 0 of every workgroup): [group 0 MFMAs + stores of the staged tile] [groups 1-2 + global loads] [barrier] [group 3].
// Ideal: 1024 cycles per MFMA group when the wave has the SIMD to itself, 2048 when two workgroups share the CU.
#define EKF_GEMM_STAMP 1
#define EKF_GEMM_LOOPSTAMP 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_dense.hpp"
using namespace ekf;
int main() {
  const int n = 6016, ldy = 2048, ld = 6144;
  float *V, *S;
  hipMalloc(&V, (size_t)(n + 128) * ldy * 4); hipMalloc(&S, (size_t)(n + 128) * ld * 4);
  std::vector<float> h((size_t)(n + 128) * ldy); for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(V, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(S, 0, (size_t)(n + 128) * ld * 4);
  std::vector<int> tm; const int nt128 = n / 128, SB = 8, ns = (nt128 + SB - 1) / SB;
  for (int si = 0; si < ns; ++si) for (int sj = 0; sj <= si; ++sj)
    for (int i = si * SB; i < std::min(nt128, (si + 1) * SB); ++i)
      for (int j = sj * SB; j < std::min(nt128, (sj + 1) * SB); ++j) if (j <= i) { tm.push_back(i); tm.push_back(j); }
  int* dtm; hipMalloc(&dtm, tm.size() * 4); hipMemcpy(dtm, tm.data(), tm.size() * 4, hipMemcpyHostToDevice);
  int* counters; hipMalloc(&counters, 4096 * 4); hipMemset(counters, 0, 4096 * 4);
  unsigned long long* dl; hipGetSymbolAddress((void**)&dl, HIP_SYMBOL(ekf_loop_buf));
  std::vector<unsigned long long> lb(8 * 1024);
  int cn = 0;
  for (int K : {1024, 384}) for (int wgs : {256, 512}) for (int ntiles : {wgs, 1128}) {
    for (int r = 0; r < 5; ++r) {
      GemmArgs a{V, ldy, V, ldy, S, ld, K, -1.0, 1.0, 2, 0, 0, 0, 0, dtm, ntiles, counters + (cn++), 0, 0, 1};
      k_gemm_mfma<ROLE_DOWNDATE, false><<<std::min(ntiles, wgs), 256>>>(a);
    }
    hipDeviceSynchronize();
    hipMemcpy(lb.data(), dl, lb.size() * 8, hipMemcpyDeviceToHost);
    double sum[4] = {0, 0, 0, 0}, steps = 0;
    for (int w = 0; w < std::min(ntiles, wgs); ++w) { for (int q = 0; q < 4; ++q) sum[q] += (double)lb[8 * w + q]; steps += (double)lb[8 * w + 4]; }
    printf("K=%4d workgroups=%3d tiles=%4d: per steady-state K step (wave 0, mean of %.0f steps): group0+stores %.0f  groups1-2+loads %.0f  barrier %.0f  group3 %.0f  total %.0f cycles\n",
           K, wgs, ntiles, steps, sum[0] / steps, sum[1] / steps, sum[2] / steps, sum[3] / steps, (sum[0] + sum[1] + sum[2] + sum[3]) / steps);
  }
  return 0;
}
