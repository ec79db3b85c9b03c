// Covariance downdate Sigma -= V_g V_g^T (a10, vR.cpp:1279) on the bf16 matrix pipe at fp32 accuracy (round 5).
//
// Arithmetic (the opt-in experiment of round 1, now the default for large maps): an fp32 value is the exact sum of three bf16 values, a = a1 + a2 + a3;
// of the nine bf16 x bf16 products of a * b six are accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest first:
// a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1; the three that are dropped (a2 b3, a3 b2, a3 b3) add up to <= 2^-24 |a||b| in
// the worst case and 2^-28 |a||b| on average (measured over 2 M random pairs) -- the size of the fp32 product's own rounding.  Six of those
// instructions retire 16 k in 192 cycles where v_mfma_f32_32x32x2_f32 needs 512: the fp32-equivalent peak of the scheme
// is 2.5 PF / 6 = 417 TF against 157 TF of the fp32 instruction.
//
// What makes it run is the data path, which is built around the LDS-DMA (global_load_lds_dwordx4):
//   * V_g is kept a second time as a PLANE IMAGE (k_split_image): per 128-row block and 16-column chunk one contiguous
//     12 KB record [plane 3][k half 2][row 128][8 bf16] -- exactly the LDS image one operand of one K step needs, so a
//     stage is filled by 1 KiB wave-instructions that are contiguous on both sides (no staging registers, no ds_write,
//     full cache lines from L2);
//   * a ring of three 24 KB stages (A record + B record): the loads of chunk s + 2 are issued right behind the barrier of
//     step s and stay in flight across the next barrier (raw s_barrier + counted s_waitcnt vmcnt, never __syncthreads);
//   * fragments by ds_read_b128 straight out of the record (lane = row, lane half = k half: conflict-free, no swizzle);
//   * persistent grid on a host-ordered list of CANONICAL 128 x 128 tiles (block of the row >= block of the column; diagonal
//     tiles first, then 8 x 8 super-tiles), two workgroups per CU; the ring runs ACROSS tiles: the next list entry is drawn
//     while the tile computes, and the first two chunks of the next tile are requested during the last two steps of this one
//     and travel under the epilogue.
// Every element pair {r, c} of Sigma is ONE sum (k_syrk_bf16x6 below), so Sigma stays exactly symmetric and the rows a rank
// of a sharded filter holds are bit-identical to the plain filter's.
// Measured (tools/syrk6_probe.hip, profiles/r5_syrk6_probe.txt): LDS reads + MFMAs alone 250 TF fp32-equivalent (1.5 PF
// executed: the clock the chip holds under this load, not the issue stream, is the ceiling); + LDS-DMA 200-215; + the C
// tile 145-190 (K = 384 .. 1152) against 100-118 for k_gemm_mfma on the same launches.
#pragma once
#include <hip/hip_runtime.h>
#include "ekf_dense.hpp"

namespace ekf {

typedef __bf16 s6_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int s6_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kS6Rec = 768;                    // 16-byte slots of one record (128 rows x 16 k x 3 planes x 2 B = 12 KB)

// fp32 rows x width block of V (row-major, leading dimension ld, columns [c0, c0 + width)) -> plane image.
// Image slot index of (row r, column c, plane p): ((r / 128) * nkc_total + c / 16) * 768 + p * 256 + ((c % 16) / 8) * 128 + r % 128.
// One thread per (row, 8 columns): 32 B read, three 16-byte slots written (consecutive lanes = consecutive rows).
__global__ void __launch_bounds__(256) k_split_image(const float* __restrict__ V, int ld, int rows, int c0, int width,
                                                    s6_u32x4* __restrict__ img, int nkc_total) {
  const int r = blockIdx.x * 128 + (threadIdx.x & 127);
  const int o0 = blockIdx.y * 2 + (threadIdx.x >> 7);          // octet (8 columns) inside the launch
  if (r >= rows || o0 * 8 >= width) return;
  const int c = c0 + o0 * 8;
  const float4 x0 = *reinterpret_cast<const float4*>(V + (size_t)r * ld + c);
  const float4 x1 = *reinterpret_cast<const float4*>(V + (size_t)r * ld + c + 4);
  const float a[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  unsigned short p0[8], p1[8], p2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 a1 = (__bf16)a[e];
    const float r1 = a[e] - (float)a1;
    const __bf16 a2 = (__bf16)r1;
    const float r2 = r1 - (float)a2;
    const __bf16 a3 = (__bf16)r2;
    p0[e] = __builtin_bit_cast(unsigned short, a1);
    p1[e] = __builtin_bit_cast(unsigned short, a2);
    p2[e] = __builtin_bit_cast(unsigned short, a3);
  }
  auto pack = [](const unsigned short* p) {
    s6_u32x4 v = {(unsigned)p[0] | ((unsigned)p[1] << 16), (unsigned)p[2] | ((unsigned)p[3] << 16),
                  (unsigned)p[4] | ((unsigned)p[5] << 16), (unsigned)p[6] | ((unsigned)p[7] << 16)};
    return v;
  };
  s6_u32x4* rec = img + ((size_t)(r >> 7) * nkc_total + (c >> 4)) * kS6Rec + ((c >> 3) & 1) * 128 + (r & 127);
  rec[0] = pack(p0);
  rec[256] = pack(p1);
  rec[512] = pack(p2);
}

struct Syrk6Args {
  const s6_u32x4* img;   // plane image of V (every row block of the state: the B side of a tile may be any block)
  int nkc_total;         // 16-column chunks per row block of the image (ldy / 16)
  int kc0;               // first chunk of this launch (c0 / 16)
  int nk;                // chunks of this launch (chunk width / 16): even, >= 8
  float* C; int ldc;     // Sigma (the whole matrix: tiles are addressed by global block index)
  const int* tile_map; int ntiles; int* counter;   // CANONICAL tiles (bi >= bj, 128 x 128 blocks), host-ordered
  // Rows of Sigma that are VALID in this address space: [0, cam) and [v_lo, v_hi).  The plain filter: every row
  // (cam = 0, v_lo = 0, v_hi = INT_MAX).  A rank of a sharded filter: the camera rows and its own rows.
  int cam, v_lo, v_hi;
  // Rider (nrider > 0): the first nrider workgroups of the launch do the right-looking update of the innovation row for
  // the chunk whose V_g this launch downdates with, nu^T[c1:] -= y_g^T L[c1:, g]^T (innov_row_column, one lane per column,
  // 256 columns per workgroup), and leave.  A chain of K fused multiply-adds per lane is 10-25 us of latency wherever it
  // runs; here it runs beside 100+ us of tiles on <= 7 of the 448+ workgroup slots (round 5 first had it in k_split_image,
  // in FRONT of the downdate: 7 -> 17 us on the second stream's critical path, three times per step).
  const float* ry = nullptr; const float* rL = nullptr; int rldl = 0; float* rnu = nullptr; int rcols = 0, rK = 0, nrider = 0;
  // De-phasing (round 6): every tile of a launch has the same K, so workgroups that start together reach their epilogues
  // together -- 448 x 192 KB of C traffic inside a few microseconds, with the matrix pipes idle meanwhile.  The second
  // workgroup of a CU (the second half of the grid) starts stag_half x 0.85 us late, and workgroup b another (b & 3) x
  // stag_mod4 x 0.85 us: the two workgroups of a CU alternate between K loop and epilogue, and the C traffic of the chip is spread.
  int stag_half = 0, stag_mod4 = 0;
  // Tail filler (round 6, the LAST downdate of an update): a workgroup that finds no tile left takes rows of the state update
  // mu += V y (16 rows per ticket of a second counter, one wave per row: k_state_update's sums) instead of leaving; ticket 0
  // holds the quaternion rows and normalises them (Qn -> su_qn).  The launch no longer ends with idle CUs waiting for its
  // slowest tiles, and the state update needs neither a launch nor a second stream.
  float* su_mu = nullptr; const float* su_V = nullptr; int su_ldy = 0, su_n = 0; const float* su_y = nullptr; int su_mpad = 0;
  float* su_qn = nullptr; int* su_counter = nullptr;
};

// Every element pair {r, c}, r >= c, is computed ONCE, as element (r, c) of its canonical tile (A block = the block of r,
// B block = the block of c), from the value of Sigma(r, c) -- read at (r, c) if row r is valid here, else at (c, r) -- and
// stored at (r, c) if row r is valid and at (c, r) if row c is valid.  So Sigma is exactly symmetric, and what a rank of a
// sharded filter holds in its rows is bit-identical to what the plain filter computes: the list of a rank is simply every
// canonical tile that touches one of its row blocks.
// ABL (tools/syrk6_probe.hip only): 1 = no C traffic (the accumulators are kept live), 2 = no LDS-DMA (the ring keeps what it
// has), 4 = every LDS-DMA reads the first record (always cache hits)
template <int ABL = 0>
__global__ void __launch_bounds__(256, 2) k_syrk_bf16x6(Syrk6Args g) {
  constexpr int NS = 3, STG = 2 * kS6Rec;                      // ring stages; slots per stage (A record + B record)
  __shared__ s6_u32x4 lds[NS * STG + 1];                       // ONE LDS object (the last slot: the queue's hand-over words)
  int* const s_next = reinterpret_cast<int*>(&lds[NS * STG]);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1, h = lane >> 5, l31 = lane & 31;
  float* const C = g.C;
  const int ldc = g.ldc;
  const int nk = g.nk;

  // validity of the 128 rows of block b: 0 none, 1 all, 2 mixed
  auto block_class = [&](int b) {
    const int lo = b * 128, hi = lo + 128;
    const int n_cam = max(0, min(hi, g.cam) - lo);
    const int o_lo = max(g.v_lo, g.cam);                       // own rows that are not camera rows
    const int n_own = max(0, min(hi, g.v_hi) - max(lo, o_lo));
    const int cnt = n_cam + n_own;
    return cnt >= 128 ? 1 : (cnt > 0 ? 2 : 0);
  };
  auto valid = [&](int r) { return r < g.cam || (r >= g.v_lo && r < g.v_hi); };
  struct Tile { int bi, bj, va, vb; };
  auto decode = [&](int rbi, int rbj) {
    Tile t;
    t.bi = rbi & 0xffff;
    t.bj = rbj & 0xffff;
    t.va = block_class(t.bi);
    t.vb = block_class(t.bj);
    return t;
  };
  // chunk c of tile t -> ring stage st: 24 pieces of 1 KiB, six per wave (pieces 0-11: A record, 12-23: B record)
  auto issue = [&](const Tile& t, int c, int st) {
    if (ABL & 2) return;
    const s6_u32x4* ra = g.img + ((ABL & 4) ? (size_t)0 : ((size_t)t.bi * g.nkc_total + g.kc0 + c) * kS6Rec);
    const s6_u32x4* rb = g.img + ((ABL & 4) ? (size_t)0 : ((size_t)t.bj * g.nkc_total + g.kc0 + c) * kS6Rec);
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int pc = wave + 4 * u;                             // u < 3: A pieces, u >= 3: B pieces
      const s6_u32x4* src = (u < 3 ? ra + pc * 64 : rb + (pc - 12) * 64) + lane;
      s6_u32x4* dst = lds + st * STG + pc * 64;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  if (g.nrider > 0 && (int)blockIdx.x < g.nrider) {
    const int c = blockIdx.x * 256 + tid;
    if (c < g.rcols) innov_row_column(g.ry, g.rL + (size_t)c * g.rldl, g.rK, g.rnu + c);
    return;
  }
  // ---- first tile -----------------------------------------------------------------------------------------------------
  if (g.stag_half | g.stag_mod4) {
    const int b = (int)blockIdx.x - g.nrider, nb = (int)gridDim.x - g.nrider;
    const int units = (b >= nb / 2 ? g.stag_half : 0) + (b & 3) * g.stag_mod4;
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(32);
  }
  if (tid == 0) {
    const int t = atomicAdd(g.counter, 1);
    s_next[0] = t < g.ntiles ? g.tile_map[2 * t] : -1;
    s_next[1] = t < g.ntiles ? g.tile_map[2 * t + 1] : 0;
  }
  __syncthreads();
  int raw_i = __builtin_amdgcn_readfirstlane(s_next[0]), raw_j = __builtin_amdgcn_readfirstlane(s_next[1]);
  if (raw_i < 0) return;
  Tile cur = decode(raw_i, raw_j);
  int st = 0;                                                  // ring stage of the current chunk
  issue(cur, 0, 0);
  issue(cur, 1, 1);
  bool have = true;
  int epi_stores = -1;                                         // stores per lane the previous tile's epilogue is KNOWN to have issued (-1: no previous tile)

  while (have) {
    const Tile t = cur;
    Tile nxt = t;
    bool have_next = false;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int ticket = 0;
    // The next list entry is drawn while this tile computes: thread 0 takes a ticket at step claim_at, the entry is looked up
    // and handed over through LDS at step pub_at, everybody reads it at step nk - 3 (the barriers of the steps in between
    // order it) and requests the next tile's first two chunks in the last two steps.
    const int claim_at = nk > 14 ? nk - 14 : 0;
    const int pub_at = min(claim_at + 5, nk - 5);
    for (int s = 0; s < nk; ++s) {
      // chunk s has landed (this wave's pieces); chunk s + 1 may be in flight; behind a tile's epilogue its stores, younger
      // than the two chunks that were already under way: the wait may leave as many of them outstanding as are KNOWN to
      // have been issued (predicated stores of diagonal / partly valid tiles are not counted: the compiler may skip them)
      const bool more1 = (s + 1 < nk) || have_next;
      if (s < 2 && epi_stores >= 57) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
      else if (s < 2 && epi_stores >= 16) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
      else if (more1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                            // everybody's pieces of chunk s; everybody is out of stage st - 1
      __builtin_amdgcn_sched_barrier(0);
      const int st2 = (st == 0) ? 2 : st - 1;                  // (st + 2) % 3
      if (s + 2 < nk) issue(t, s + 2, st2);
      else if (have_next) issue(nxt, s + 2 - nk, st2);
      // fragments of this chunk: lane = row, lane half = k half (conflict-free ds_read_b128, no swizzle)
      const s6_u32x4* Sa = lds + st * STG + h * 128 + wr * 64 + l31;
      const s6_u32x4* Sb = lds + st * STG + kS6Rec + h * 128 + wc * 64 + l31;
      s6_bf16x8 fa[3][2], fb[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[p][i] = __builtin_bit_cast(s6_bf16x8, Sa[p * 256 + i * 32]);
          fb[p][i] = __builtin_bit_cast(s6_bf16x8, Sb[p * 256 + i * 32]);
        }
      // smallest products first: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1
      constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA_[q]][i], fb[PB_[q]][j], acc[i][j], 0, 0, 0);
      // queue hand-over (thread 0)
      if (s == claim_at) {
        if (tid == 0) ticket = atomicAdd(g.counter, 1);
      } else if (s == pub_at) {
        if (tid == 0) {
          s_next[0] = ticket < g.ntiles ? g.tile_map[2 * ticket] : -1;
          s_next[1] = ticket < g.ntiles ? g.tile_map[2 * ticket + 1] : 0;
        }
      } else if (s == nk - 3) {
        raw_i = __builtin_amdgcn_readfirstlane(s_next[0]);
        raw_j = __builtin_amdgcn_readfirstlane(s_next[1]);
        have_next = raw_i >= 0;
        if (have_next) nxt = decode(raw_i, raw_j);
      }
      st = (st + 1 == NS) ? 0 : st + 1;
    }
    // ---- epilogue: Sigma' = Sigma - acc (one rounding at the magnitude of Sigma) -------------------------------------------
    const bool diag = t.bi == t.bj;
    epi_stores = (diag || (ABL & 1)) ? 0 : ((t.va == 1 ? 64 : 0) + (t.vb == 1 ? 16 : 0));
    const int path = (ABL & 1) ? 3 : ((t.va == 1 && !diag) ? 0 : ((t.va == 0 && !diag) ? 1 : 2));
    const int rb0 = t.bi * 128 + wr * 64 + 4 * h, cb0 = t.bj * 128 + wc * 64 + l31;
    if (path == 0) {
      // every row of the A block is valid here (the plain filter; the interior blocks of a rank): the whole C tile is
      // requested at once (64 loads in flight), then direct stores, and mirror stores where the column's row is valid
      float v[2][2][16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float* Cd = C + (size_t)(rb0 + i * 32) * ldc + cb0 + j * 32;
#pragma unroll
          for (int e = 0; e < 16; ++e) v[i][j][e] = Cd[(size_t)((e & 3) + 8 * (e >> 2)) * ldc];
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = cb0 + j * 32;
          float* Cd = C + (size_t)(rb0 + i * 32) * ldc + c;
          float* Ct = C + (size_t)c * ldc + rb0 + i * 32;
#pragma unroll
          for (int e = 0; e < 16; ++e) v[i][j][e] = v[i][j][e] - acc[i][j][e];
#pragma unroll
          for (int e = 0; e < 16; ++e) Cd[(size_t)((e & 3) + 8 * (e >> 2)) * ldc] = v[i][j][e];
          if (t.vb == 1 || (t.vb == 2 && valid(c))) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              f32x4 o = {v[i][j][4 * gq], v[i][j][4 * gq + 1], v[i][j][4 * gq + 2], v[i][j][4 * gq + 3]};
              *reinterpret_cast<f32x4*>(Ct + 8 * gq) = o;
            }
          }
        }
    } else if (path == 1) {
      // no row of the A block is valid here: the tile only feeds the rows of its B block, read and written transposed
      f32x4 o[2][2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = cb0 + j * 32;
          const float* Ct = C + (size_t)c * ldc + rb0 + i * 32;
          if (t.vb == 1 || valid(c)) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) o[i][j][gq] = *reinterpret_cast<const f32x4*>(Ct + 8 * gq);
          }
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = cb0 + j * 32;
          float* Ct = C + (size_t)c * ldc + rb0 + i * 32;
          if (t.vb == 1 || valid(c)) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
              for (int u = 0; u < 4; ++u) o[i][j][gq][u] = o[i][j][gq][u] - acc[i][j][4 * gq + u];
              *reinterpret_cast<f32x4*>(Ct + 8 * gq) = o[i][j][gq];
            }
          }
        }
    } else if (path == 2) {
      // diagonal tiles ((r, c) and (c, r) add the same six products in a different order: only r >= c is used) and partly
      // valid A blocks (the ragged ends of a rank's rows, the camera block on another rank): both images of a 32 x 32
      // block are requested at once, then element by element
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int r0 = rb0 + i * 32, c = cb0 + j * 32;
          float* Cd = C + (size_t)r0 * ldc + c;
          float* Ct = C + (size_t)c * ldc + r0;
          const bool cvalid = t.vb == 1 || (t.vb == 2 && valid(c));
          float vd[16];
          f32x4 ot[4];
#pragma unroll
          for (int e = 0; e < 16; ++e) vd[e] = Cd[(size_t)((e & 3) + 8 * (e >> 2)) * ldc];
          if (t.va != 1) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) ot[gq] = *reinterpret_cast<const f32x4*>(Ct + 8 * gq);
          }
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int r = r0 + 8 * gq + u;
              const bool rvalid = t.va == 1 || (t.va == 2 && valid(r));
              const float cin = (t.va == 1 || rvalid) ? vd[4 * gq + u] : ot[gq][u];
              const float x = cin - acc[i][j][4 * gq + u];
              if (!diag || r >= c) {
                if (rvalid) Cd[(size_t)(8 * gq + u) * ldc] = x;
                if (cvalid && r != c) Ct[8 * gq + u] = x;
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[i][j][e]));
    }
    have = have_next;
    cur = nxt;
  }
  if (g.su_mu != nullptr) {
    const int nsu = (g.su_n + 15) / 16;
    for (;;) {
      __syncthreads();
      if (tid == 0) s_next[0] = atomicAdd(g.su_counter, 1);
      __syncthreads();
      const int tk = __builtin_amdgcn_readfirstlane(s_next[0]);
      if (tk >= nsu) break;
#pragma unroll 1
      for (int r4 = 0; r4 < 4; ++r4) {
        const int row = 16 * tk + 4 * wave + r4;
        if (row < g.su_n) state_update_row(g.su_mu, g.su_V, g.su_ldy, row, g.su_y, g.su_mpad, lane);
      }
      if (tk == 0 && g.su_qn != nullptr) {
        __threadfence_block();
        __syncthreads();                           // rows 3 .. 6 are written (this workgroup holds rows 0 .. 15)
        if (tid == 0) state_update_normalise(g.su_mu, g.su_qn);
      }
    }
  }
}

}  // namespace ekf
