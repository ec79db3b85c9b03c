// Optional fp32-accurate contraction on the bf16 matrix pipe (EKF_OPT_SPLIT_BF16, off by default).
//
// An fp32 value is the exact sum of three bf16 values (8 + 8 + 8 significand bits): a = a1 + a2 + a3
// with a1 = bf16(a), a2 = bf16(a - a1), a3 = a - a1 - a2.  A product a*b then is the sum of nine
// bf16 x bf16 products, each exact in fp32; the six with i + j <= 4 carry everything above
// 2^-25 |a||b| -- below half an fp32 ulp of the product -- so
//     a*b ~= a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1,   accumulated in fp32 by the MFMA,
// is as accurate as an fp32 multiply-add chain.  v_mfma_f32_32x32x16_bf16 retires 16 k per 32
// cycles against 2 k per 64 cycles of v_mfma_f32_32x32x2_f32: six of them cost 192 cycles per 16 k
// where the fp32 instruction needs 512.  This first kernel is far from that bound (tools/split_bench.hip:
// 131 TF fp32-equivalent at K = 2048 against 111-116 for the fp32 kernel; 197 with the staging removed): its
// fragment reads (12 ds_read_b128 per 24 MFMAs) are not overlapped with the MFMAs yet.
// Tried and dropped (bit-identical results, no gain): a two-wave 128 x 64 shape (125 TF: the accumulators spill
// into AGPR copies), an 8-wave ping-pong workgroup with the halves held half a K step apart by the barrier (89 TF:
// 64 + 96 + 48 registers of accumulators, fragments and staging do not fit the 256 a wave gets at two waves per
// SIMD; the spills to scratch cost more than the lockstep they remove).  Per-phase accounting says the two
// workgroups of a CU run in lockstep here (MFMA 3072 + LDS 3000 + L2->L1 1536 cycles per K step add up instead of
// overlapping): the next form is BK = 16 stages so that the ping-pong fits in registers.
//
// Used for the dominant contraction only, the symmetric downdate Sigma -= V_g V_g^T:
//   k_split_bf16      V (fp32) -> three bf16 planes, once per chunk (the planes are then read ~48 times)
//   k_syrk_bf16x3     128 x 128 tiles from the same work queue / mirror convention as k_gemm_mfma
#pragma once
#include <hip/hip_runtime.h>
#include "ekf_dense.hpp"

namespace ekf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// rows x width block of V (leading dimension ld, starting at column c0) -> planes P0, P1, P2 (same ld).
__global__ void k_split_bf16(const float* __restrict__ V, int ld, int rows, int c0, int width,
                             __bf16* __restrict__ P0, __bf16* __restrict__ P1, __bf16* __restrict__ P2) {
  const int r = blockIdx.y;
  if (r >= rows) return;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < width; c += gridDim.x * blockDim.x) {
    const size_t o = (size_t)r * ld + c0 + c;
    const float a = V[o];
    const __bf16 a1 = (__bf16)a;
    const float r1 = a - (float)a1;
    const __bf16 a2 = (__bf16)r1;
    const float r2 = r1 - (float)a2;
    P0[o] = a1;
    P1[o] = a2;
    P2[o] = (__bf16)r2;
  }
}

struct SplitArgs {
  const __bf16* P[3];    // planes of V, already offset to the first column of the chunk
  int ld;                // elements per row of the planes
  float* C; int ldc;     // Sigma
  int K;                 // chunk width (multiple of 32)
  const int* tile_map; int ntiles; int* counter;
};

// C -= V V^T over the lower-triangular 128 x 128 tiles of tile_map, strictly-lower tiles mirrored.
// 4 waves as 2 x 2, each 64 x 64 = 2 x 2 accumulators of 32 x 32.  LDS image per plane and operand:
// [kg = k / 8][row] slots of 8 bf16 (16 bytes), slot = kg * 128 + (row ^ 4 kg): the ds_write_b128 of
// 4 lanes that share a row and the ds_read_b128 of 32 lanes that share kg are bank-conflict free.
// One LDS stage of 48 KiB; the next K step waits in registers (12 x 16 bytes per lane) under the MFMAs.
__global__ void __launch_bounds__(256, 2) k_syrk_bf16x3(SplitArgs g) {
  constexpr int T = 128, BK = 32, NKG = BK / 8;
  constexpr int PLANE = NKG * T;                 // slots per plane and operand
  __shared__ f32x4 lds[2 * 3 * PLANE];           // A planes, then B planes
  __shared__ int s_tile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  float* C = g.C;
  const int ldc = g.ldc, ld = g.ld;
  if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(127);   // de-phase the two workgroups of a CU
  for (;;) {
    __syncthreads();
    if (tid == 0) s_tile = atomicAdd(g.counter, 1);
    __syncthreads();
    const int t = s_tile;
    if (t >= g.ntiles) break;
    const int bi = g.tile_map[2 * t], bj = g.tile_map[2 * t + 1];
    // staging: per plane and operand 128 rows x 4 slots = 512 slots, 2 per lane
    const __bf16* gp[2][3][2];
    int slot[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + 256 * p;
      const int row = idx >> 2, kg = idx & 3;
      slot[p] = kg * T + (row ^ (4 * kg));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        gp[0][pl][p] = g.P[pl] + (size_t)(bi * T + row) * ld + 8 * kg;
        gp[1][pl][p] = g.P[pl] + (size_t)(bj * T + row) * ld + 8 * kg;
      }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;       // C' = C - acc in the epilogue: one rounding at |C|
    f32x4 rg[2][3][2];
    auto load_regs = [&](int k0) {
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int p = 0; p < 2; ++p) rg[o][pl][p] = *reinterpret_cast<const f32x4*>(gp[o][pl][p] + k0);
    };
    load_regs(0);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
      __syncthreads();                           // everybody is done reading the previous K step
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int p = 0; p < 2; ++p) lds[(o * 3 + pl) * PLANE + slot[p]] = rg[o][pl][p];
      __syncthreads();
      if (k0 + BK < g.K) load_regs(k0 + BK);
      // both k16 halves are fetched up front: the reads of the second half land under the MFMAs of the first
      bf16x8 fa[2][3][2], fb[2][3][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kg = 2 * s + h;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int ar = wr * 64 + i * 32 + l31, br = wc * 64 + i * 32 + l31;
            f32x4 va = lds[pl * PLANE + kg * T + (ar ^ (4 * kg))];
            f32x4 vb = lds[(3 + pl) * PLANE + kg * T + (br ^ (4 * kg))];
            fa[s][pl][i] = __builtin_bit_cast(bf16x8, va);
            fb[s][pl][i] = __builtin_bit_cast(bf16x8, vb);
          }
      }
      // smallest products first: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1
      constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][PA_[q]][i], fb[s][PB_[q]][j], acc[i][j], 0, 0, 0);
    }
    const bool mirror = bi > bj;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rbase = bi * T + wr * 64 + i * 32;
        const int c = bj * T + wc * 64 + j * 32 + l31;
        float v[16];
        if (bi == bj) {
          // diagonal tile: (r, c) and (c, r) add the same six products in a different order; keep Sigma exactly
          // symmetric by storing the lower triangle and its mirror
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int r = rbase + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float x = C[(size_t)r * ldc + c] - acc[i][j][e];
            if (r >= c) {
              C[(size_t)r * ldc + c] = x;
              C[(size_t)c * ldc + r] = x;
            }
          }
          continue;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = rbase + (e & 3) + 8 * (e >> 2) + 4 * h;
          v[e] = C[(size_t)r * ldc + c] - acc[i][j][e];
          C[(size_t)r * ldc + c] = v[e];
        }
        if (mirror) {
          float* Ct = C + (size_t)c * ldc;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            f32x4 o = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
            *reinterpret_cast<f32x4*>(Ct + rbase + 8 * gq + 4 * h) = o;
          }
        }
      }
  }
}

}  // namespace ekf
