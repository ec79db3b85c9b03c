// One persistent launch for ALL tile-GEMM work of an update that runs beside the serial Cholesky chain:
// the triangular solves V_g = [W_g; nu_g^T] Z_gg, the right-looking updates W[:, c1:] -= V_g L[c1:, g]^T and the
// downdates Sigma -= V_g V_g^T of every column chunk g.  The separate launches of Filter::update each pay a ramp
// and a partially filled last round of tiles (a third of a launch at 2.2 tiles per workgroup, tools/clock_probe.hip);
// here a persistent grid (2 workgroups per CU of the second stream's CU mask) draws tiles from ONE host-ordered task
// list and every tile waits only for ITS OWN inputs:
//
//   task            reads                                   waits for                                   signals
//   solve(g,i,j)    W[i, chunk g], Z_gg                     chain_done >= g+1;  W row tile i has received  vdone[g][i] += 1
//                                                           every update of the earlier chunks (wdone)
//   wupdate(g,i,c)  V_g[i], L[c, chunk g]; r/w W[i, c]      vdone[g][i] complete; W tile (i, c) has had   wdone[t(c)][i] += 1,
//                                                           the updates of chunks < g (wver)              wver[i][c] += 1
//   downdate(g,I,J) V_g[I], V_g[J]; r/w Sigma[I, J]         vdone[g][I], vdone[g][J] complete; tile has   sver[I, J] += 1
//                                                           had the downdates of chunks < g (sver)
//
// The list is in an order in which every dependency of a task comes EARLIER in the list (or is the chain, which runs
// on its own stream and reserved CUs), and a workgroup holds at most one task: whatever a waiting workgroup waits for
// has been drawn by a resident workgroup that is not waiting on anything later -> no deadlock.  Waits are bounded
// (status[3] is raised and the tile skipped after ~2 s: a bug shows up as an error code, not as a hung GPU).
// Cross-workgroup visibility follows the release / acquire recipe of the CDNA4 guide: storing waves drain
// (s_waitcnt vmcnt(0)), workgroup barrier, lane 0 agent-scope release fence + drained, relaxed agent atomic add;
// the consumer polls with relaxed agent loads, then one agent-scope acquire fence, a drain, a workgroup barrier.
// Same arithmetic per tile as k_gemm_mfma (same K order): results are bit-identical to the launch-per-phase path.
#pragma once
#include "ekf_dense.hpp"

namespace ekf {

enum : int { FLOW_SOLVE = 0, FLOW_WUPDATE = 1, FLOW_DOWNDATE = 2 };

struct FlowTask {              // 64 bytes
  int type, K;
  long long a_off, b_off, c_off;    // element offsets of the tile origins in their base arrays
  int bi, bj;                  // tile indices (downdate: mirror target and diagonal test)
  int dep[3], need[3];         // counters[dep[k]] >= need[k]; dep < 0: none
  int sig[2];                  // counters[sig[k]] += 1 when the tile is stored; sig < 0: none
};

struct FlowArgs {
  const float* W; float* V; const float* Y; const float* Zs; float* Sigma;   // W is also written (W update), via Wm
  float* Wm;
  int ldy, ld;
  const FlowTask* tasks;
  int ntasks;
  int* head;                   // work-queue head
  int* counters;               // dependency counters, zeroed before the launch
  int* status;                 // status[3]: a wait timed out
  int stagger;
  unsigned long long* trace;   // optional (EKF_FLOW_TRACE): 4 words per task: fetch, ready, done (100 MHz ticks), block | type << 32
};

__device__ __forceinline__ bool flow_wait(const int* counters, int dep, int need) {
  if (dep < 0) return true;
  // back-off: hundreds of workgroups polling one line cut the bandwidth of everything else (the chain's kernels
  // beside them): ~1 us between polls at first, ~5 us after a few misses
  for (int spin = 0; spin < (1 << 20); ++spin) {
    if (__hip_atomic_load(counters + dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    __builtin_amdgcn_s_sleep(40);
    if (spin > 4) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
  }
  return false;
}

__global__ void __launch_bounds__(256, 2) k_gemm_flow(FlowArgs g) {
  constexpr int TM = 128, TN = 128, BK = 32, NQ = BK / 4, MI = 2, NJ = 2, PA = 4, PB = 4;
  __shared__ f32x4 lds[2 * NQ * (TM + TN)];
  __shared__ int s_task;
  __shared__ int s_ok;
  constexpr int STAGE = NQ * (TM + TN);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  if (g.stagger && blockIdx.x >= gridDim.x / 2)
    for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  for (;;) {
    __syncthreads();                               // everyone is done with the previous tile (LDS, s_task)
    if (tid == 0) {
      const int t = atomicAdd(g.head, 1);
      int ok = 1;
      if (g.trace && t < g.ntasks) g.trace[4 * (size_t)t] = __builtin_amdgcn_s_memrealtime();
      if (t < g.ntasks) {
        const FlowTask* tk = g.tasks + t;
        for (int k = 0; k < 3; ++k)
          if (!flow_wait(g.counters, tk->dep[k], tk->need[k])) ok = 0;
        if (!ok) g.status[3] = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (g.trace && t < g.ntasks) g.trace[4 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
      s_task = t;
      s_ok = ok;
    }
    __syncthreads();
    const int t = s_task;
    if (t >= g.ntasks) break;
    const FlowTask tk = g.tasks[t];
    const bool ok = s_ok != 0;
    const int type = tk.type;
    const bool BT = (type == FLOW_SOLVE);
    const float* A = (type == FLOW_SOLVE ? g.W : g.V) + tk.a_off;
    const float* B = (type == FLOW_SOLVE ? g.Zs : (type == FLOW_WUPDATE ? g.Y : g.V)) + tk.b_off;
    float* C = (type == FLOW_SOLVE ? g.V : (type == FLOW_WUPDATE ? g.Wm : g.Sigma)) + tk.c_off;
    const int lda = g.ldy, ldb = g.ldy, ldc = (type == FLOW_DOWNDATE) ? g.ld : g.ldy;
    const int K = tk.K;
    const float alpha = (type == FLOW_SOLVE) ? 1.f : -1.f;
    const bool has_beta = (type != FLOW_SOLVE);
    if (ok) {
      const float* Ag[PA];
      const float* Bg[4];
      int aslot[PA], bslot[4];
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        const int idx = tid + 256 * p;
        const int row = idx >> 3, q = idx & 7;
        Ag[p] = A + (size_t)row * lda + q * 4;
        aslot[p] = q * TM + (row ^ q);
      }
      const int qk = tid / (TN / 4), cq = tid % (TN / 4);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (!BT) {
          const int idx = tid + 256 * p;
          const int row = idx >> 3, q = idx & 7;
          Bg[p] = B + (size_t)row * ldb + q * 4;
          bslot[p] = q * TN + (row ^ q);
        } else {
          Bg[p] = B + (size_t)(4 * qk + p) * ldb + 4 * cq;
          bslot[p] = qk * TN + ((4 * cq + p) ^ qk);
        }
      }
      f32x16 acc[MI][NJ];                          // C enters in the epilogue, as in k_gemm_mfma (bit-identical to it)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      f32x4 ra[PA], rb[4];
      auto load_tile = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const f32x4*>(Ag[p] + k0);
#pragma unroll
        for (int p = 0; p < 4; ++p)
          rb[p] = BT ? *reinterpret_cast<const f32x4*>(Bg[p] + (size_t)k0 * ldb) : *reinterpret_cast<const f32x4*>(Bg[p] + k0);
      };
      auto store_tile = [&](int stage) {
        f32x4* As = lds + stage * STAGE;
        f32x4* Bs = As + NQ * TM;
#pragma unroll
        for (int p = 0; p < PA; ++p) As[aslot[p]] = ra[p];
        if (!BT) {
#pragma unroll
          for (int p = 0; p < PB; ++p) Bs[bslot[p]] = rb[p];
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p) {            // column 4cq+p gets (k0..k3) of that column
            f32x4 tq = {rb[0][p], rb[1][p], rb[2][p], rb[3][p]};
            Bs[bslot[p]] = tq;
          }
        }
      };
      constexpr int NG = BK / 8;
      f32x4 fa[2][MI], fb[2][NJ];
      auto read_frag = [&](int stage_, int s, int buf) {
        const f32x4* As = lds + stage_ * STAGE;
        const f32x4* Bs = As + NQ * TM;
        const int q = 2 * s + h;
#pragma unroll
        for (int tt = 0; tt < MI; ++tt) {
          const int ar = wr * (TM / 2) + tt * 32 + l31;
          fa[buf][tt] = As[q * TM + (ar ^ q)];
        }
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
          const int br = wc * (TN / 2) + tt * 32 + l31;
          fb[buf][tt] = Bs[q * TN + (br ^ q)];
        }
      };
      auto mfma_group = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i][e], fb[buf][j][e], acc[i][j], 0, 0, 0);
      };
      load_tile(0);
      store_tile(0);
      if (BK < K) load_tile(BK);
      __syncthreads();
      read_frag(0, 0, 0);
      int stage = 0;
      for (int k0 = 0; k0 < K; k0 += BK, stage ^= 1) {
        const bool more = k0 + BK < K;
#pragma unroll
        for (int s = 0; s < NG; ++s) {
          if (s + 1 < NG) {
            read_frag(stage, s + 1, (s + 1) & 1);
          } else {
            __syncthreads();
            if (more) read_frag(stage ^ 1, 0, 0);
          }
          mfma_group(s & 1);
          if (s == 0 && more) {
            store_tile(stage ^ 1);
            if (k0 + 2 * BK < K) load_tile(k0 + 2 * BK);
          }
        }
      }
      const bool mirror = (type == FLOW_DOWNDATE) && (tk.bi > tk.bj);
      float* Ct = g.Sigma + (size_t)(tk.bj * TN) * g.ld + (size_t)tk.bi * TM;       // origin of the mirrored tile (J, I)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        f32x16 cin[NJ];
        if (has_beta) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const float* Cp = C + (size_t)(wr * (TM / 2) + i * 32 + 4 * h) * ldc + wc * (TN / 2) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) cin[j][e] = Cp[(size_t)((e & 3) + 8 * (e >> 2)) * ldc];
          }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int rbase = wr * (TM / 2) + i * 32;
          const int c = wc * (TN / 2) + j * 32 + l31;
          float v[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int r = rbase + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float x = has_beta ? __builtin_fmaf(alpha, acc[i][j][e], cin[j][e]) : alpha * acc[i][j][e];
            v[e] = x;
            C[(size_t)r * ldc + c] = x;
          }
          if (mirror) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              f32x4 o = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
              *reinterpret_cast<f32x4*>(Ct + (size_t)c * g.ld + rbase + 8 * gq + 4 * h) = o;
            }
          }
        }
      }
    }
    // publish: every storing wave drains, the workgroup meets, lane 0 releases and counts
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int k = 0; k < 2; ++k)
        if (tk.sig[k] >= 0) __hip_atomic_fetch_add(g.counters + tk.sig[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g.trace) {
        g.trace[4 * (size_t)t + 2] = __builtin_amdgcn_s_memrealtime();
        g.trace[4 * (size_t)t + 3] = (unsigned long long)blockIdx.x | ((unsigned long long)tk.type << 32) | ((unsigned long long)(gridDim.x) << 40);
      }
    }
  }
}

// chain -> flow: "chunk g of the factorisation is done" (one lane; launched on the chain's stream after the chunk)
__global__ void k_flow_chain_done(int* counter, int value) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(counter, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace ekf
