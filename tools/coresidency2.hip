// Does a 64 x 64 chain tile (k_gemm_mfma<ROLE_TRAILING, false, 64, 64>: 80 VGPRs, 32 KiB LDS) get a slot on a CU that holds
// two persistent 128 x 128 downdate workgroups (2 x 216 VGPRs, 2 x 64 KiB)?  The downdate grid fills ALL 256 CUs here (no CU
// mask), so the trailing launch only runs early if it fits beside them.  (debug harness)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_dense.hpp"
using namespace ekf;
int main() {
  const int n = 6144, ldy = 2048, ld = 6144;
  float *V, *S, *Y; int *cnt, *tm;
  hipMalloc(&V, (size_t)n * ldy * 4); hipMalloc(&S, (size_t)n * ld * 4); hipMalloc(&Y, (size_t)4096 * ldy * 4);
  hipMalloc(&cnt, 4096 * 4);
  hipMemset(V, 0, (size_t)n * ldy * 4); hipMemset(S, 0, (size_t)n * ld * 4); hipMemset(Y, 0, (size_t)4096 * ldy * 4);
  std::vector<int> tmv; for (int i = 0; i < 48; ++i) for (int j = 0; j <= i; ++j) { tmv.push_back(i); tmv.push_back(j); }
  hipMalloc(&tm, tmv.size() * 4); hipMemcpy(tm, tmv.data(), tmv.size() * 4, hipMemcpyHostToDevice);
  hipStream_t sa, sb; int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t e0, e1, g0, g1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&g0); hipEventCreate(&g1);
  int cn = 0;
  for (int wgs : {0, 512}) for (int rep = 0; rep < 3; ++rep) {
    hipMemset(cnt, 0, 4096 * 4);
    hipDeviceSynchronize();
    if (wgs) {
      GemmArgs g{V, ldy, V, ldy, S, ld, 2048, -1.0, 1.0, 2, 0, 0, 0, 0, tm, (int)tmv.size() / 2, cnt + 8 * (cn++)};
      hipEventRecord(g0, sb);
      k_gemm_mfma<ROLE_DOWNDATE, false><<<dim3(wgs), 256, 0, sb>>>(g);
      hipEventRecord(g1, sb);
    }
    for (volatile int spin = 0; spin < 3000000; ++spin) {}        // head start for the persistent grid
    // trailing update of a 1664-row remainder: 26 x 26 tiles of 64 x 64, K = 128, lower triangle
    GemmArgs t{Y, ldy, Y, ldy, Y + 128, ldy, 128, -1.0, 1.0, 1, 0, 0, 0, 0, nullptr, 0, nullptr};
    hipEventRecord(e0, sa);
    k_gemm_mfma<ROLE_TRAILING, false, 64, 64><<<dim3(26, 26), 256, 0, sa>>>(t);
    hipEventRecord(e1, sa);
    hipDeviceSynchronize();
    float md, mg = 0; hipEventElapsedTime(&md, e0, e1); if (wgs) hipEventElapsedTime(&mg, g0, g1);
    printf("persistent downdate grid %3d workgroups (%.3f ms): trailing launch (351 tiles of 64 x 64) takes %.1f us\n", wgs, mg, md * 1e3);
  }
  return 0;
}
