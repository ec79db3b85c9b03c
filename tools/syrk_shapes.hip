// Downdate (SYRK, lower tiles + mirror) with 128x128 vs 64x64 tiles, queued super-tile lists. Debug harness.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <algorithm>
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_dense.hpp"
using namespace ekf;
static std::vector<int> tilemap(int nt, int SB) {
  std::vector<int> tm; const int ns = (nt + SB - 1) / SB;
  for (int si = 0; si < ns; ++si) for (int sj = 0; sj <= si; ++sj)
    for (int i = si * SB; i < std::min(nt, (si + 1) * SB); ++i)
      for (int j = sj * SB; j < std::min(nt, (sj + 1) * SB); ++j) if (j <= i) { tm.push_back(i); tm.push_back(j); }
  return tm;
}
static int* g_counters = nullptr;
template <int TM, int TN> float run(GemmArgs g, const std::vector<int>& tm, int wgs, int reps) {
  int* d; hipMalloc(&d, tm.size() * 4); hipMemcpy(d, tm.data(), tm.size() * 4, hipMemcpyHostToDevice);
  if (!g_counters) hipMalloc(&g_counters, 64 * 4);
  hipMemset(g_counters, 0, 256);
  g.tile_map = d; g.ntiles = (int)tm.size() / 2; g.counter = g_counters;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_gemm_mfma<ROLE_DOWNDATE, false, TM, TN><<<dim3(std::min(g.ntiles, wgs)), 256>>>(g); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) { g.counter = g_counters + 1 + r; k_gemm_mfma<ROLE_DOWNDATE, false, TM, TN><<<dim3(std::min(g.ntiles, wgs)), 256>>>(g); }
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); hipFree(d); return ms / reps;
}
int main() {
  const int n = 6144, ldy = 2048, ld = 6144;
  float *V, *S;
  hipMalloc(&V, (size_t)n * ldy * 4); hipMalloc(&S, (size_t)n * ld * 4);
  std::vector<float> h((size_t)n * ldy); for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(V, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(S, 0, (size_t)n * ld * 4);
  for (int K : {2048, 512}) {
    GemmArgs g{V, ldy, V, ldy, S, ld, K, -1.0, 1.0, 2, 0, 0, 0, 0, nullptr, 0, nullptr};
    double flop = (double)n * n * K;
    float t = run<128, 128>(g, tilemap(n / 128, 8), 512, 10);
    printf("SYRK K=%4d 128x128 (512 wgs): %.3f ms %.1f TF\n", K, t, flop / t / 1e9);
    for (int wgs : {512, 768, 1024}) {
      t = run<64, 64>(g, tilemap(n / 64, 16), wgs, 10);
      printf("SYRK K=%4d  64x64  (%4d wgs): %.3f ms %.1f TF\n", K, wgs, t, flop / t / 1e9);
    }
  }
  return 0;
}
