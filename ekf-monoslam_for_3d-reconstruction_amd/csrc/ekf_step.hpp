// One block step of the Cholesky chain -- diagonal factor, panel, trailing update -- as ONE launch, for maps whose chain is a
// handful of steps with a few dozen tiles each (one column chunk: N up to ~230 features; BASELINE configs[1], N = 200: four
// steps that were 16.6 + 5.6 + 5.7 us as three launches, each of the two short ones little more than its own ramp and drain).
//
// The structure is k_update_small_onelaunch's (ekf_small.hpp): nothing is handed over inside the launch, every workgroup
// re-derives what it needs.  A workgroup owns one 64 x 64 tile (I, K) of the trailing update (or, where a row block has no tile,
// just its 64 rows of the panel).  It
//   * loads the diagonal block (j, j) into the factor's LDS image and its own 64 + 64 rows of column block j into registers,
//   * factors the block itself (diag_factor_lds: the same instructions in every workgroup, the same bits),
//   * forms P_I = Y[I, j] Linv^T and P_K (k_panel_direct's product, Linv read from the image where the factor left it),
//   * applies Y[I, K] -= P_I P_K^T (k_gemm_mfma<TRAILING, 64, 64>'s sums: v_mfma_f32_32x32x2_f32, k pairs {e, e + 4} of every
//     group of eight in ascending order, C' = fma(-1, acc, C)).
// Designated workgroups also write what the launches this one replaces leave behind: L and Linv (one workgroup), the rows
// of the panel (the diagonal tile of every row block; a strip row block's first tile, or a panel-only workgroup when the
// step has no tile for it).  The column block j is read by everybody and overwritten in place by the designated writers:
// as in ekf_small.hpp one arrival counter orders "nobody writes before everybody has read" (every workgroup is resident:
// the host only takes this path when the step has fewer workgroups than the stream has CUs; the wait is bounded, status[3]).
// Every sum is the sum of the launch it replaces: bit-identical (EKF_STEP_FUSED=0; tests/test_gpu_parity.py).
// Measured (N = 200, profiles/r6_step_fused_n200.txt): a step 26.5-27 us against 27.9 for the three launches, the last (short)
// step 11.5 against 11.2: 0.1533 -> 0.1500 ms per filter step.  The factor's 16.6 us stay; what the fusion removes is two
// ramps and drains, what it adds is the factor's image load in front and two barriers behind it.
#pragma once
#include "ekf_small.hpp"

namespace ekf {

enum : int { SF_TILE = 1, SF_WRITE_PANEL = 2, SF_WRITE_DIAG = 4 };

struct StepFusedArgs {
  float* Y; int ldy;
  float* Dj;                                     // Linv of this step (128 x 128, row-major)
  int* status;
  int m, j;                                      // real rows of S; the block step (diagonal block at row / column 128 j)
  const int* wl; int nwg;                        // per workgroup: 64-row block I, 64-row block K (= the tile's column block), flags
  unsigned* gate; unsigned gate_target;
};

__global__ void __launch_bounds__(1024, 1) k_chain_step_fused(StepFusedArgs g) {
  using namespace oneblock;                      // NB = 128, PITCH = 132, f4
  constexpr int LDA = PITCH;
  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  __shared__ __attribute__((aligned(16))) float big[NB * LDA + 2 * 64 * PITCH];
  __shared__ __attribute__((aligned(16))) float x16[2][16 * 20];
  __shared__ __attribute__((aligned(16))) float rinv[2][16];
  __shared__ float junk16[64 * 16];
  __shared__ float sdinv[NB];
  float* const a = big;                          // the factor's image: L below, Z = L^-T above the diagonal
  float* const sVi = big + NB * LDA;             // P_I | P_K, 64 x 132 each
  float* const sVj = sVi + 64 * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int i64 = g.wl[3 * blockIdx.x], k64 = g.wl[3 * blockIdx.x + 1], flags = g.wl[3 * blockIdx.x + 2];
  const bool tile = (flags & SF_TILE) != 0, same = !tile || k64 == i64;
  const int half = wave >> 3, ww = wave & 7, rg = ww & 3;
  const bool second = ww >= 4;                   // waves 4..7 of each half: rows of P_K
  const bool v_active = !second || !same;
  const int vrow0 = 64 * (second ? k64 : i64) + rg * 16;
  const int jc = g.j * 128;
  const size_t ldy = (size_t)g.ldy;
  // ---- everything this workgroup reads of column block j and of its tile, requested up front --------------------------
  f4 fa[8];
  if (v_active) {
    const float* yrow = g.Y + (size_t)(vrow0 + lr) * ldy + jc;
#pragma unroll
    for (int u = 0; u < 8; ++u) fa[u] = *reinterpret_cast<const f4*>(yrow + 16 * u + 4 * lq);
  }
  const int wr = (wave >> 1) & 1, wc = wave & 1, h = lane >> 5, l31 = lane & 31;   // waves 0..3: the 32 x 32 blocks of the tile
  float cv[16];
  if (tile && wave < 4) {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      cv[e] = g.Y[(size_t)(64 * i64 + wr * 32 + 4 * h + (e & 3) + 8 * (e >> 2)) * ldy + 64 * k64 + wc * 32 + l31];
  }
  {
    const float* D = g.Y + (size_t)jc * ldy + jc;
    diag_load_lds([&](int i, int j0) { return *reinterpret_cast<const f4*>(D + (size_t)i * ldy + j0); }, a);
  }
  __syncthreads();                               // (waits for every load above) this workgroup has read all it will read:
  if (tid == 0) __hip_atomic_fetch_add(g.gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {
    const DiagLds L{a, x16, rinv, junk16};
    diag_factor_lds<7>(g.status, max(1, min(8, (g.m - jc + 15) / 16)), L);
  }
  __builtin_amdgcn_s_setprio(0);
  __syncthreads();
  if (tid < NB) sdinv[tid] = 1.f / a[tid * LDA + tid];
  if (tid == 0) {                                // nobody writes column block j or a tile before every workgroup has read them
    bool ok = false;
    for (int spin = 0; spin < (1 << 20) && !ok; ++spin) {
      ok = (int)(__hip_atomic_load(g.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.gate_target) >= 0;
      if (!ok) __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) g.status[3] = 1;
  }
  __syncthreads();
  if (flags & SF_WRITE_DIAG) {                   // what k_chol_diag_packed leaves: L (zeros above) in place, Linv in Dinv
    float* Ab = g.Y + (size_t)jc * ldy + jc;
    float* Db = g.Dj;
    diag_store_lds([&](int i, int j0, const f4& x) { *reinterpret_cast<f4*>(Ab + (size_t)i * ldy + j0) = x; },
                   [&](int i, int jj, float x) { Db[(size_t)i * 128 + jj] = x; }, a);
  }
  // ---- P = Y[rows, j] Linv^T: k_panel_direct's product, column tiles dealt over the two halves (A B B A A B B A) -----------
  if (v_active) {
    float* svp = (second ? sVj : sVi) + (rg * 16) * PITCH;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const int back = 7 - ct;
      const bool mine = (((back & 3) == 0) || ((back & 3) == 3)) ? half == 0 : half == 1;
      if (!mine) continue;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const int c = 16 * ct + lr;                // fb[e] = Linv[c][k], k = 16 u + 4 lq + e
      const float dc = sdinv[c];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u <= ct) {
          f4 fb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = 16 * u + 4 * lq + e;
            const float zv = a[k * LDA + c];
            fb[e] = (u < ct || k < c) ? zv : ((k == c) ? dc : 0.f);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][e], fb[e], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) svp[(4 * lq + e) * PITCH + 16 * ct + lr] = acc[e];
    }
  }
  __syncthreads();                               // P_I, P_K complete
  if (flags & SF_WRITE_PANEL) {                  // the panel rows of block I, in place, whole lines
    for (int q = tid; q < 64 * 32; q += 1024) {
      const int r = q >> 5, c0 = 4 * (q & 31);
      *reinterpret_cast<f4*>(g.Y + (size_t)(64 * i64 + r) * ldy + jc + c0) = *reinterpret_cast<const f4*>(sVi + r * PITCH + c0);
    }
  }
  if (!tile || wave >= 4) return;
  // ---- Y[I, K] -= P_I P_K^T: the sums of k_gemm_mfma<TRAILING, 64, 64> (lane (l31, h) of MFMA e of group s supplies k = 8 s + 4 h + e)
  f32x16_t acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* ap = sVi + (wr * 32 + l31) * PITCH + 4 * h;
  const float* bp = (same ? sVi : sVj) + (wc * 32 + l31) * PITCH + 4 * h;
#pragma unroll
  for (int s8 = 0; s8 < 16; ++s8) {
    const f4 fav = *reinterpret_cast<const f4*>(ap + 8 * s8);
    const f4 fbv = *reinterpret_cast<const f4*>(bp + 8 * s8);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fav[e], fbv[e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float v = 1.f * cv[e];
    v = __builtin_fmaf(-1.f, acc[e], v);
    g.Y[(size_t)(64 * i64 + wr * 32 + 4 * h + (e & 3) + 8 * (e >> 2)) * ldy + 64 * k64 + wc * 32 + l31] = v;
  }
}

}  // namespace ekf
