// Standalone timing of the split-bf16 symmetric update (debug harness; not part of the library).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DSPLIT_DBG=n] tools/split_bench.hip -o tools/split_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_split.hpp"
using namespace ekf;
int main() {
  const int n = 6144, ldy = 2048, ld = 6144;
  float *V, *S; __bf16* P[3]; int *dtm, *cnt;
  hipMalloc(&V, (size_t)n * ldy * 4); hipMalloc(&S, (size_t)n * ld * 4);
  for (auto& p : P) hipMalloc(&p, (size_t)n * ldy * 2);
  std::vector<float> h((size_t)n * ldy); for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(V, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(S, 0, (size_t)n * ld * 4);
  std::vector<int> tm; const int nt = n / 128, SB = 8, ns = (nt + SB - 1) / SB;
  for (int si = 0; si < ns; ++si) for (int sj = 0; sj <= si; ++sj)
    for (int i = si * SB; i < std::min(nt, (si + 1) * SB); ++i)
      for (int j = sj * SB; j < std::min(nt, (sj + 1) * SB); ++j) if (j <= i) { tm.push_back(i); tm.push_back(j); }
  hipMalloc(&dtm, tm.size() * 4); hipMemcpy(dtm, tm.data(), tm.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&cnt, 64 * 4);
  const int ntiles = (int)tm.size() / 2;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int K : {2048, 896, 512}) {
    dim3 g((K + 255) / 256, n);
    k_split_bf16<<<g, 256>>>(V, ldy, n, 0, K, P[0], P[1], P[2]);
    hipMemset(cnt, 0, 64 * 4);
    SplitArgs args{{P[0], P[1], P[2]}, ldy, S, ld, K, dtm, ntiles, cnt};
    k_syrk_bf16x3<<<std::min(ntiles, 512), 256>>>(args);
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) { args.counter = cnt + 1 + r; k_syrk_bf16x3<<<std::min(ntiles, 512), 256>>>(args); }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    printf("split SYRK K=%4d: %.3f ms  %.1f TF (fp32-equivalent n^2 K)\n", K, ms, (double)n * n * K / ms / 1e9);
  }
  return 0;
}
