// k_gemm_pipe (cross-tile pipelined persistent tile GEMM) against k_gemm_mfma on the shapes of the N = 1000 update:
// bit-identity of the results and time per launch alone on the chip.  (debug harness; build: see tools/README or
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/pipe_probe.hip -o tools/pipe_probe)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_dense.hpp"
#include "../ekf-monoslam_for_3d-reconstruction_amd/csrc/ekf_gemm_pipe.hpp"
using namespace ekf;

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
// de-phased head: the first `wgs` entries are what workgroup b starts on (first_static): full tiles for b < wgs / 2, half tiles
// for the others (the second workgroup of each CU), so that the two workgroups of a CU reach their epilogues half a tile apart
static std::vector<int> dephased_list(int nt128, int nsplit_tail, int wgs) {
  std::vector<int> base = tri_list(nt128, 0);
  const int nt = (int)base.size() / 2;
  const int nhead_half = wgs / 4;                 // tiles that become the wgs / 2 half tiles of the head
  std::vector<int> tl;
  int t = 0;
  for (int b = 0; b < wgs / 2; ++b, ++t) { tl.push_back(base[2 * t]); tl.push_back(base[2 * t + 1]); }
  for (int b = 0; b < nhead_half; ++b, ++t)
    for (int s2 = 0; s2 < 2; ++s2) { tl.push_back((2 * base[2 * t] + s2) | kHalfTile); tl.push_back(base[2 * t + 1]); }
  for (; t < nt - nsplit_tail; ++t) { tl.push_back(base[2 * t]); tl.push_back(base[2 * t + 1]); }
  for (; t < nt; ++t)
    for (int s2 = 0; s2 < 2; ++s2) { tl.push_back((2 * base[2 * t] + s2) | kHalfTile); tl.push_back(base[2 * t + 1]); }
  return tl;
}

int main(int argc, char** argv) {
  const int n = 6144, ldy = 2048, ld = 6144;
  float *V, *S, *S2, *Z, *O1, *O2;
  hipMalloc(&V, (size_t)(n + 128) * ldy * 4); hipMalloc(&S, (size_t)n * ld * 4); hipMalloc(&S2, (size_t)n * ld * 4);
  hipMalloc(&Z, (size_t)ldy * ldy * 4); hipMalloc(&O1, (size_t)(n + 128) * ldy * 4); hipMalloc(&O2, (size_t)(n + 128) * ldy * 4);
  std::vector<float> hv((size_t)(n + 128) * ldy); for (auto& x : hv) x = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(V, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hz((size_t)ldy * ldy); for (size_t i = 0; i < hz.size(); ++i) { const int r = i / ldy, c = i % ldy; hz[i] = (r <= c) ? (rand() % 2001 - 1000) * 1e-3f : 0.f; }
  hipMemcpy(Z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hs((size_t)n * ld);
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { const float x = (rand() % 2001 - 1000) * 1e-2f; hs[(size_t)i * ld + j] = x; hs[(size_t)j * ld + i] = x; }
  int* counters; hipMalloc(&counters, 65536 * 4); hipMemset(counters, 0, 65536 * 4);
  int cn = 0;
  auto up = [&](const std::vector<int>& v) { int* d; hipMalloc(&d, v.size() * 4); hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice); return d; };
  hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
  const int nt128 = n / 128;
  std::vector<float> r1((size_t)n * ld), r2((size_t)n * ld);
  // ---- bit identity: downdate with half tiles ---------------------------------------------------------------
  {
    std::vector<int> tl = tri_list(nt128, 384);
    int* dtl = up(tl); const int ntl = (int)tl.size() / 2;
    for (int K : {128, 384, 512}) {
      hipMemcpy(S, hs.data(), hs.size() * 4, hipMemcpyHostToDevice); hipMemcpy(S2, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
      GemmArgs a{V, ldy, V, ldy, S, ld, K, -1.0, 1.0, 2, 0, 0, 0, 0, dtl, ntl, counters + (cn++), 0, 0, 1};
      k_gemm_mfma<ROLE_DOWNDATE, false><<<512, 256>>>(a);
      GemmArgs b{V, ldy, V, ldy, S2, ld, K, -1.0, 1.0, 2, 0, 0, 0, 0, dtl, ntl, counters + (cn++), 0, 0, 0};
      k_gemm_pipe<ROLE_DOWNDATE, false><<<512, 256>>>(b);
      hipMemcpy(r1.data(), S, r1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), S2, r2.size() * 4, hipMemcpyDeviceToHost);
      size_t bad = 0, changed = 0; for (size_t i = 0; i < r1.size(); ++i) { bad += (memcmp(&r1[i], &r2[i], 4) != 0); changed += (r1[i] != hs[i]); }
      printf("downdate K=%d pipe vs mfma: %zu differing of %zu (changed by the launch: %zu)\n", K, bad, r1.size(), changed);
      // de-phased list, first_static
      std::vector<int> dl = dephased_list(nt128, 384, 448);
      int* ddl = up(dl);
      hipMemcpy(S2, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
      GemmArgs c{V, ldy, V, ldy, S2, ld, K, -1.0, 1.0, 2, 0, 0, 0, 0, ddl, (int)dl.size() / 2, counters + (cn++), 0, 0, 0};
      c.first_static = 1;
      k_gemm_pipe<ROLE_DOWNDATE, false><<<448, 256>>>(c);
      hipMemcpy(r2.data(), S2, r2.size() * 4, hipMemcpyDeviceToHost);
      bad = 0; for (size_t i = 0; i < r1.size(); ++i) bad += (memcmp(&r1[i], &r2[i], 4) != 0);
      printf("downdate K=%d pipe de-phased/static vs mfma: %zu differing\n", K, bad);
      hipFree(ddl);
    }
    hipFree(dtl);
  }
  // ---- bit identity: triangular solve (NN, ktri), last chunk of the N = 1000 step: 9 column tiles x 49 row tiles -------------
  {
    const int ntr = (n + 128) / 128, wt = 9;
    std::vector<int> tl; for (int j = wt - 1; j >= 0; --j) for (int i = 0; i < ntr; ++i) { tl.push_back(i); tl.push_back(j); }
    int* dtl = up(tl); const int ntl = (int)tl.size() / 2;
    hipMemset(O1, 0, (size_t)(n + 128) * ldy * 4); hipMemset(O2, 0, (size_t)(n + 128) * ldy * 4);
    GemmArgs a{V, ldy, Z, ldy, O1, ldy, wt * 128, 1.0, 0.0, 0, 0, 0, 1, 0, dtl, ntl, counters + (cn++), 0, 0, 0};
    k_gemm_mfma<ROLE_SOLVE, true><<<256, 256>>>(a);
    GemmArgs b = a; b.C = O2; b.counter = counters + (cn++);
    k_gemm_pipe<ROLE_SOLVE, true><<<256, 256>>>(b);
    std::vector<float> o1((size_t)(n + 128) * ldy), o2(o1.size());
    hipMemcpy(o1.data(), O1, o1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), O2, o2.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, nz = 0; for (size_t i = 0; i < o1.size(); ++i) { bad += (memcmp(&o1[i], &o2[i], 4) != 0); nz += (o1[i] != 0.f); }
    printf("solve 9 x %d tiles pipe vs mfma: %zu differing of %zu (nonzero %zu)\n", ntr, bad, o1.size(), nz);
    for (int wgs : {256, 512}) for (int var = 0; var < 2; ++var) {
      float best = 1e9;
      for (int pass = 0; pass < 3; ++pass) {
        const int reps = 20;
        hipDeviceSynchronize(); hipEventRecord(ea);
        for (int r = 0; r < reps; ++r) {
          GemmArgs c = a; c.C = O2; c.counter = counters + (cn++ % 60000);
          if (var) k_gemm_pipe<ROLE_SOLVE, true><<<wgs, 256>>>(c); else k_gemm_mfma<ROLE_SOLVE, true><<<wgs, 256>>>(c);
        }
        hipEventRecord(eb); hipEventSynchronize(eb);
        float ms; hipEventElapsedTime(&ms, ea, eb); best = std::min(best, ms / reps);
      }
      printf("solve last chunk wgs=%d %s: %.1f us\n", wgs, var ? "pipe" : "mfma", best * 1e3);
    }
    hipFree(dtl);
  }
  // ---- timing: downdate alone on the chip ---------------------------------------------------------------------------
  hipMemset(S, 0, (size_t)n * ld * 4);
  for (int K : {384, 512, 1152}) for (int wgs : {448, 512}) {
    const int nsplit = 384;
    std::vector<int> tl = tri_list(nt128, nsplit), dl = dephased_list(nt128, nsplit, wgs), dl0 = dephased_list(nt128, 0, wgs), pl = tri_list(nt128, 0);
    int* dtl = up(tl); int* ddl = up(dl); int* ddl0 = up(dl0); int* dpl = up(pl);
    struct Var { const char* name; int kern; int* list; int ntl; int fs; double beta; int tri; };
    std::vector<Var> vars = {
      {"mfma  halftail", 0, dtl, (int)tl.size() / 2, 0, 1.0, 2},
      {"pipe  halftail", 1, dtl, (int)tl.size() / 2, 0, 1.0, 2},
      {"pipe  plain   ", 1, dpl, (int)pl.size() / 2, 0, 1.0, 2},
      {"pipe  dephased+halftail", 1, ddl, (int)dl.size() / 2, 1, 1.0, 2},
      {"pipe  dephased", 1, ddl0, (int)dl0.size() / 2, 1, 1.0, 2},
      {"mfma  halftail noCread", 0, dtl, (int)tl.size() / 2, 0, 0.0, 2},
      {"mfma  halftail noMirror", 0, dtl, (int)tl.size() / 2, 0, 1.0, 1},
      {"mfma  halftail noCread noMirror", 0, dtl, (int)tl.size() / 2, 0, 0.0, 1},
      {"pipe  halftail noCread noMirror", 1, dtl, (int)tl.size() / 2, 0, 0.0, 0},
    };
    for (auto& v : vars) {
      float best = 1e9;
      for (int pass = 0; pass < 3; ++pass) {
        const int reps = 20;
        hipDeviceSynchronize(); hipEventRecord(ea);
        for (int r = 0; r < reps; ++r) {
          GemmArgs a{V, ldy, V, ldy, S, ld, K, -1.0, v.beta, v.tri, 0, 0, 0, 0, v.list, v.ntl, counters + (cn++ % 60000), 0, 0, 1};
          a.first_static = v.fs;
          if (v.kern) k_gemm_pipe<ROLE_DOWNDATE, false><<<std::min(v.ntl, wgs), 256>>>(a);
          else k_gemm_mfma<ROLE_DOWNDATE, false><<<std::min(v.ntl, wgs), 256>>>(a);
        }
        hipEventRecord(eb); hipEventSynchronize(eb);
        float ms; hipEventElapsedTime(&ms, ea, eb); best = std::min(best, ms / reps);
      }
      printf("K=%4d wgs=%3d %-34s %.1f us  %.1f TF\n", K, wgs, v.name, best * 1e3, 2.0 * (pl.size() / 2) * 128 * 128 * K / best / 1e9);
    }
    hipFree(dtl); hipFree(ddl); hipFree(ddl0); hipFree(dpl);
  }
  return 0;
}
