// k_syrk_bf16x6 (ekf_syrk6.hpp): sampled entries against an fp64 host sum, exact symmetry, a rank of a sharded filter (camera
// rows + a ragged range of own rows valid, everything else poisoned with NaN) bit-identical to the plain result, and time per
// launch alone on the chip at the shapes of the N = 1000 update, with ablations.  (The round-1 kernel k_syrk_bf16x3 this one
// was first checked bitwise against -- the same six products in the same order -- is gone; profiles/r5_syrk6_probe.txt.)  (debug harness; hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/syrk6_probe.hip -o tools/syrk6_probe)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <vector>
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_dense.hpp"
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_syrk6.hpp"
using namespace ekf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static std::vector<int> tri_list(int nt128, int nsplit) {
  std::vector<int> tm; const int SB = 8, ns = (nt128 + SB - 1) / SB;
  for (int si = 0; si < ns; ++si) for (int sj = 0; sj <= si; ++sj)
    for (int i = si * SB; i < std::min(nt128, (si + 1) * SB); ++i)
      for (int j = sj * SB; j < std::min(nt128, (sj + 1) * SB); ++j) if (j <= i) { tm.push_back(i); tm.push_back(j); }
  const int nt = (int)tm.size() / 2;
  std::vector<int> tl(tm.begin(), tm.begin() + 2 * (nt - nsplit));
  for (int t = nt - nsplit; t < nt; ++t)
    for (int s2 = 0; s2 < 2; ++s2) { tl.push_back((2 * tm[2 * t] + s2) | kHalfTile); tl.push_back(tm[2 * t + 1]); }
  return tl;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 6144, ldy = 2048, ld = n;
  const int nkc_total = ldy / 16;
  float *V, *S, *S2; s6_u32x4* img;
  const size_t velems = (size_t)(n + 128) * ldy;
  CK(hipMalloc(&V, velems * 4)); CK(hipMalloc(&S, (size_t)n * ld * 4)); CK(hipMalloc(&S2, (size_t)n * ld * 4));
  CK(hipMalloc(&img, velems * 6));
  std::vector<float> hv(velems); for (auto& x : hv) x = (rand() % 200001 - 100000) * 1e-5f * ((rand() & 7) ? 1.f : 37.f);
  CK(hipMemcpy(V, hv.data(), velems * 4, hipMemcpyHostToDevice));
  std::vector<float> hs((size_t)n * ld);
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { const float x = (rand() % 2001 - 1000) * 1e-1f; hs[(size_t)i * ld + j] = x; hs[(size_t)j * ld + i] = x; }
  int* counters; CK(hipMalloc(&counters, 65536 * 4)); CK(hipMemset(counters, 0, 65536 * 4));
  int cn = 0;
  auto up = [&](const std::vector<int>& v) { int* d; hipMalloc(&d, v.size() * 4); hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice); return d; };
  hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
  const int nt128 = n / 128;
  // planes (old kernel) and image (new kernel) of the whole V
  { dim3 grid((n + 128) / 128, ldy / 16); k_split_image<<<grid, 256>>>(V, ldy, n + 128, 0, ldy, img, nkc_total); }
  CK(hipDeviceSynchronize());
  std::vector<float> r1((size_t)n * ld), r2((size_t)n * ld);
  std::vector<int> plain = tri_list(nt128, 0), halft = tri_list(nt128, std::min(384, (int)plain.size() / 6));
  int* dplain = up(plain); int* dhalf = up(halft);
  for (int K : {128, 384}) for (int c0 : {0, 512}) {
    CK(hipMemcpy(S, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(S2, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    Syrk6Args b{img, nkc_total, c0 / 16, K / 16, S2, ld, dplain, (int)plain.size() / 2, counters + (cn++), 0, 0, INT_MAX};
    k_syrk_bf16x6<0><<<512, 256>>>(b);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r2.data(), S2, r2.size() * 4, hipMemcpyDeviceToHost));
    size_t asym = 0, changed = 0;
    for (size_t i = 0; i < r2.size(); ++i) changed += (r2[i] != hs[i]);
    for (int i = 0; i < n; i += 1) for (int j = 0; j < i; j += 1) asym += (r2[(size_t)i * ld + j] != r2[(size_t)j * ld + i]);
    // fp64 samples
    double emax = 0, eref = 0;
    for (int s = 0; s < 2000; ++s) {
      const int i = rand() % n, j = rand() % n;
      double acc = 0; for (int k = 0; k < K; ++k) acc += (double)hv[(size_t)i * ldy + c0 + k] * (double)hv[(size_t)j * ldy + c0 + k];
      const double want = (double)hs[(size_t)i * ld + j] - acc;
      emax = std::max(emax, std::fabs(want - r2[(size_t)i * ld + j])); eref = std::max(eref, std::fabs(want));
    }
    printf("K=%4d c0=%4d: %zu of %zu entries changed, asymmetric pairs %zu, max err vs fp64 %.3e (max |ref| %.3e)\n",
           K, c0, changed, r2.size(), asym, emax, eref);
  }
  // ---- a rank of a sharded filter: camera rows + own rows [v_lo, v_hi) valid, everything else poisoned -------------------------
  {
    const int K = 384, c0 = 512, cam = 14;
    CK(hipMemcpy(S, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    Syrk6Args a{img, nkc_total, c0 / 16, K / 16, S, ld, dplain, (int)plain.size() / 2, counters + (cn++), 0, 0, INT_MAX};
    k_syrk_bf16x6<0><<<512, 256>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r1.data(), S, r1.size() * 4, hipMemcpyDeviceToHost));
    for (auto range : {std::pair<int, int>{cam + 6 * 300, cam + 6 * 650}, std::pair<int, int>{cam, cam + 6 * 333}, std::pair<int, int>{cam + 6 * 700, n - 130}}) {
      const int v_lo = range.first, v_hi = range.second;
      auto valid = [&](int r) { return r < cam || (r >= v_lo && r < v_hi); };
      std::vector<float> hp = hs;
      const float nanv = std::nanf("");
      for (int r = 0; r < n; ++r) if (!valid(r)) for (int c = 0; c < n; ++c) hp[(size_t)r * ld + c] = nanv;
      CK(hipMemcpy(S2, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
      auto touched = [&](int b) { for (int r = b * 128; r < b * 128 + 128; ++r) if (valid(r)) return true; return false; };
      std::vector<int> tl;
      for (size_t t = 0; t < plain.size() / 2; ++t) if (touched(plain[2 * t]) || touched(plain[2 * t + 1])) { tl.push_back(plain[2 * t]); tl.push_back(plain[2 * t + 1]); }
      int* dtl = up(tl);
      Syrk6Args b{img, nkc_total, c0 / 16, K / 16, S2, ld, dtl, (int)tl.size() / 2, counters + (cn++), cam, v_lo, v_hi};
      k_syrk_bf16x6<0><<<448, 256>>>(b);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(r2.data(), S2, r2.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0, cnt = 0;
      for (int r = 0; r < n; ++r) if (valid(r)) for (int c = 0; c < n; ++c) { ++cnt; bad += (memcmp(&r1[(size_t)r * ld + c], &r2[(size_t)r * ld + c], 4) != 0); }
      printf("rank rows [0,%d) + [%d,%d): %zu tiles of %zu; %zu of %zu valid-row elements differ from the plain result\n", cam, v_lo, v_hi,
             tl.size() / 2, plain.size() / 2, bad, cnt);
      hipFree(dtl);
    }
  }
  // timing
  CK(hipMemset(S, 0, (size_t)n * ld * 4));
  for (int K : {384, 512, 1152}) for (int wgs : {448, 512}) for (int var = 0; var < 7; ++var) {
    float best = 1e9;
    for (int pass = 0; pass < 3; ++pass) {
      const int reps = 20;
      hipDeviceSynchronize(); hipEventRecord(ea);
      for (int r = 0; r < reps; ++r) {
        if (var == 0) {
          continue;
        } else {
          Syrk6Args b{img, nkc_total, 0, K / 16, S, ld, dplain, (int)plain.size() / 2, counters + (cn++ % 60000), 0, 0, INT_MAX};
          if (var <= 2) k_syrk_bf16x6<0><<<wgs, 256>>>(b);
          else if (var == 3) k_syrk_bf16x6<1><<<wgs, 256>>>(b);
          else if (var == 4) k_syrk_bf16x6<2><<<wgs, 256>>>(b);
          else if (var == 5) k_syrk_bf16x6<3><<<wgs, 256>>>(b);
          else k_syrk_bf16x6<5><<<wgs, 256>>>(b);
        }
      }
      hipEventRecord(eb); hipEventSynchronize(eb);
      float ms; hipEventElapsedTime(&ms, ea, eb); best = std::min(best, ms / reps);
    }
    printf("K=%4d wgs=%3d %-22s %.1f us  %.1f TF fp32-equivalent\n", K, wgs, var == 0 ? "syrk3 (round 1)" : (var == 1 ? "syrk6 plain list" : (var == 2 ? "syrk6 half-tile tail" : (var == 3 ? "syrk6 half, no C" : (var == 4 ? "syrk6 half, no DMA" : (var == 5 ? "syrk6 half, no C no DMA" : "syrk6 half, no C, DMA hits"))))),
           best * 1e3, 2.0 * (plain.size() / 2) * 128 * 128 * K / best / 1e9);
  }
  // the image builder
  for (int K : {384, 1152}) {
    float best = 1e9;
    for (int pass = 0; pass < 3; ++pass) {
      hipDeviceSynchronize(); hipEventRecord(ea);
      for (int r = 0; r < 10; ++r) { dim3 grid((n + 128) / 128, K / 16); k_split_image<<<grid, 256>>>(V, ldy, n + 128, 0, K, img, nkc_total); }
      hipEventRecord(eb); hipEventSynchronize(eb);
      float ms; hipEventElapsedTime(&ms, ea, eb); best = std::min(best, ms / 10);
    }
    printf("k_split_image K=%d: %.1f us (%.2f TB/s of 10 B per element)\n", K, best * 1e3, 10.0 * (n + 128) * K / best / 1e9);
  }
  return 0;
}
