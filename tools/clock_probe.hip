// In-kernel clock of the symmetric-update kernel under sustained load (diagnostic build; not part of the library).
// Delta s_memtime / delta s_memrealtime x 100 MHz per workgroup around its tile loop, after ~2 s of back-to-back
// launches on random data; median over workgroups.  Compared with a register-only MFMA loop on the same data.
#define EKF_GEMM_STAMP 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_dense.hpp"
using namespace ekf;

__global__ void __launch_bounds__(256, 2) k_reg_loop(const float* __restrict__ src, float* __restrict__ out, int iters,
                                                      unsigned long long* stamps) {
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float a[2][4], b[2][4];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) {
    a[i][e] = src[(threadIdx.x * 8 + i * 4 + e) & 4095];
    b[i][e] = src[(threadIdx.x * 8 + i * 4 + e + 2048) & 4095];
  }
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    stamps[4 * blockIdx.x] = t0; stamps[4 * blockIdx.x + 1] = r0;
    stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime(); stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

static double median_clock(const std::vector<unsigned long long>& st, int nwg, double* span_us) {
  std::vector<double> ghz, us;
  for (int w = 0; w < nwg; ++w) {
    const double dc = (double)(st[4 * w + 2] - st[4 * w]), dr = (double)(st[4 * w + 3] - st[4 * w + 1]);
    if (dr > 0) { ghz.push_back(dc / dr * 0.1); us.push_back(dr / 100.0); }
  }
  std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
  *span_us = us[us.size() / 2];
  return ghz[ghz.size() / 2];
}

static int run(int n) {
  const int ldy = 2048, ld = n;
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
  const int nt = (int)tm.size() / 2;
  int* counters; hipMalloc(&counters, 16384 * 4);
  unsigned long long* dst; hipGetSymbolAddress((void**)&dst, HIP_SYMBOL(ekf_stamp_buf));
  std::vector<unsigned long long> st(4 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int K : {2048, 896, 512}) {
    GemmArgs g{V, ldy, V, ldy, S, ld, K, -1.0, 1.0, 2, 0, 0, 0, 0, dtm, nt, nullptr, 0, 0, 1};
    const double flop = (double)n * n * K;
    const int reps = std::min(16000, (int)(2.0 / (flop / 110e12)));
    hipMemset(counters, 0, 16384 * 4);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) { g.counter = counters + r; k_gemm_mfma<ROLE_DOWNDATE, false><<<512, 256>>>(g); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
    double span; const double ghz = median_clock(st, 512, &span);
    {
      unsigned long long* dph; hipGetSymbolAddress((void**)&dph, HIP_SYMBOL(ekf_phase_buf));
      std::vector<unsigned long long> ph(8 * 1024);
      hipMemcpy(ph.data(), dph, ph.size() * 8, hipMemcpyDeviceToHost);
      double sum[4] = {0, 0, 0, 0}, tiles = 0;
      for (int w = 0; w < 512; ++w) { for (int q = 0; q < 4; ++q) sum[q] += (double)ph[8 * w + q]; tiles += (double)ph[8 * w + 4]; }
      {
        unsigned long long s0 = ~0ull; for (int w = 0; w < 512; ++w) s0 = std::min(s0, st[4 * w + 1]);
        std::vector<double> starts, e2, e3, eo;
        for (int w = 0; w < 512; ++w) {
          starts.push_back((st[4 * w + 1] - s0) / 100.0);
          const double e = (st[4 * w + 3] - s0) / 100.0;
          const int nt_ = (int)ph[8 * w + 4];
          (nt_ == 2 ? e2 : nt_ == 3 ? e3 : eo).push_back(e);
        }
        auto pct = [](std::vector<double>& v, const char* name) {
          if (v.empty()) return;
          std::sort(v.begin(), v.end());
          printf("   %s (%zu workgroups): min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us\n", name, v.size(), v.front(), v[v.size() / 10],
                 v[v.size() / 2], v[v.size() * 9 / 10], v.back());
        };
        pct(starts, "start"); pct(e2, "end, 2 tiles"); pct(e3, "end, 3 tiles"); pct(eo, "end, other");
      }
      printf("   per tile (wave 0 cycles, mean of %.0f tiles of the last launch): fetch %.0f  prologue %.0f  K loop %.0f (MFMA issue %d)  epilogue %.0f\n",
             tiles, sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, K / 2 * 4 * 64, sum[3] / tiles);
    }
    printf("SYRK n=%d K=%4d: %d launches, %.4f ms each, %.1f TF; in-kernel clock %.3f GHz (median of 512 workgroups, span %.0f us); "
           "peak at that clock %.1f TF -> %.1f%%\n", n, K, reps, ms / reps, flop / (ms / reps) / 1e9, ghz, span,
           256.0 * 256 * ghz / 1e3, 100.0 * flop / (ms / reps) / 1e9 / (256.0 * 256 * ghz / 1e3));
  }
  hipFree(V); hipFree(S); hipFree(dtm); hipFree(counters);
  return 0;
}

int main() {
  for (int n : {6016}) run(n);
  float* V; hipMalloc(&V, 1 << 20); hipMemset(V, 0x3c, 1 << 20);
  std::vector<unsigned long long> st(4 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  {
    unsigned long long* stamps; hipMalloc(&stamps, 4 * 1024 * 8);
    float* out; hipMalloc(&out, 4096);
    const int iters = 4000;     // 16 MFMAs each
    hipEventRecord(e0);
    const int reps = 600;
    for (int r = 0; r < reps; ++r) k_reg_loop<<<512, 256>>>(V, out, iters, stamps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
    double span; const double ghz = median_clock(st, 512, &span);
    const double flop = 512.0 * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
    printf("register-only MFMA loop: %.1f TF; in-kernel clock %.3f GHz; peak at that clock %.1f TF -> %.1f%%\n",
           flop / (ms / reps) / 1e9, ghz, 256.0 * 256 * ghz / 1e3, 100.0 * flop / (ms / reps) / 1e9 / (256.0 * 256 * ghz / 1e3));
  }
  return 0;
}
