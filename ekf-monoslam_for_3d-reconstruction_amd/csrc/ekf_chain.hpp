// The serial chain of the blocked Cholesky of S as ONE launch per column chunk (round 6): a look-ahead, dataflow
// scheduled form of the per-block-step launches of Filter::chain_steps (diagonal factor + panel + trailing update,
// three launch boundaries per block step: vR.cpp:1276's St.inverse() was 16 x 3 launches at N = 1000).
//
// Every unit of work is a TASK on 128 x 128 blocks of Y (S on top, the chunk's identity strip under it):
//   D(j)        diagonal block j: factor + inverse                        (the body of k_chol_diag_packed)
//   P(j; I)     panel: rows of block I of column block j  <-  P Linv_jj^T  (the arithmetic of k_panel_direct)
//   T(j; I, K)  trailing: block (I, K) -= P(j; I) P(j; K)^T               (the arithmetic of k_gemm_mfma<TRAILING, 64 x 64>)
// Each element is the same sum of the same products in the same order as in the per-step launches -- the result is
// BIT-IDENTICAL (tests/test_gpu_parity.py::test_launch_structure_knobs_are_bit_identical[EKF_CHAIN_PERSISTENT=0]).
//
// Scheduling.  The host writes two task lists per launch (Filter::ensure_chain_lists):
//   * the CRITICAL list -- D(j), P(j; j+1), T(j; j+1, j+1), D(j+1), ... -- executed in order by ONE workgroup (the first
//     one to arrive): the next diagonal block is updated and factored while the rest of step j is still running;
//   * the BULK list -- every other panel and tile, in an order in which what the next step needs comes first --
//     drawn ticket by ticket by all other workgroups.
// A task carries up to three (flag, value) pairs it waits for and one it publishes.  A task only ever waits for tasks that
// come EARLIER in its own list or for tasks of the other list whose own waits are satisfiable the same way (the host
// proves it by running the two lists with one worker each: Filter::ensure_chain_lists), so the launch makes progress as
// soon as TWO of its workgroups are resident, whatever else shares the device: no co-residency of the whole grid is
// assumed.  Every wait is bounded (status[2] is raised and every workgroup leaves).
//
// Hand-overs (MI355X_MICROARCH.md, inter-workgroup visibility, the form measured for `sc1` loads in place of the acquire):
// EVERY store of Y / Dinv in this kernel is write-through (`sc1`), every storing wave drains `vmcnt(0)`, the workgroup
// passes a barrier, ONE lane stores the flag `sc1`; the consumer polls that word with `sc1` loads from one lane, the
// workgroup passes a barrier, and EVERY load of Y / Dinv is an `sc1` load to registers (buffer_load ... sc1).  One
// workgroup per CU (128 KB of LDS).  Flags carry the update's epoch in their upper bits, so they are never cleared between
// updates.
#pragma once
#include "ekf_dense.hpp"

namespace ekf {

enum : int { CT_DIAG = 0, CT_PANEL = 1, CT_TRAIL = 2, CT_TRAILQ = 4 };   // CT_TRAILQ + q: ONE 64 x 64 quarter (q = 2 gr + gc) of a block
struct ChainTask {                 // 48 bytes
  int type, j, I, K;               // block step; row block; column block (units of 128)
  int dep[3];                      // flag index to wait for (-1: none) ...
  int need[3];                     // ... until its value (below the epoch bits) is >= need
  int pub, val;                    // flag published when the task is done (-1: none), and its value
};
constexpr int kChainEpochShift = 12;             // flag = epoch << 12 | value; value <= block steps + 1 < 4096
constexpr unsigned kChainPimg = 75776;           // LDS: the critical workgroup's image of P(j; j+1) (64 KB) behind the diagonal block's arrays
constexpr unsigned kChainCtl = 141312;           // LDS: the control words
constexpr unsigned kChainLds = kChainCtl + 256;  // dynamic LDS of k_chain_persistent
constexpr unsigned long long kChainTimeoutTicks = 300000000ull;   // 3 s of the 100 MHz wall clock

struct ChainArgs {
  float* Y; int ldy; unsigned y_bytes;
  float* Dinv; unsigned dinv_bytes;
  int* status;
  int m;                                         // real rows of S
  int s0, s1, deferred;                          // block steps of this launch; 1: the trailing update of step s0 - 1 comes first
  int nblk, rb;                                  // block steps of S; row blocks incl. the widest strip (flag indexing)
  const ChainTask* bulk; int nbulk;
  unsigned* flags; int abort_word;               // hand-over words; flags[abort_word] != 0: every workgroup leaves
  unsigned epoch;                                // (update sequence number) << kChainEpochShift
  int* counters;                                 // [0] bulk ticket, [1] role ticket (zero at launch)
  unsigned* trace; int trace_cap;                // EKF_CHAIN_TRACE: [0] = records written, then 8 words per task:
                                                 // type | workgroup << 8 | critical << 24, j, I, K, wall clock (10 ns) at draw / dependencies met / computed / published
};

namespace chain {
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(1))) unsigned gu32;       // every shared word: a GLOBAL agent-scope access, never flat
typedef __attribute__((address_space(1))) const int gci32;

// agent-coherent accesses: aux = 16 is sc1 (loads: served by L2, never by this CU's L1; stores: write-through)
// (AUX = 0: ordinary accesses, for the kernels whose hand-overs are launch boundaries)
template <int AUX = 16>
__device__ __forceinline__ f4 ld16(rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX));
}
template <int AUX = 16>
__device__ __forceinline__ void st16(rsrc_t r, unsigned byte_off, const f4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, (int)byte_off, 0, AUX);
}
template <int AUX = 16>
__device__ __forceinline__ float ld4(rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, AUX));
}
template <int AUX = 16>
__device__ __forceinline__ void st4(rsrc_t r, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)byte_off, 0, AUX);
}

// One lane: wait until flags[idx] >= want (same epoch, value reached), the abort word is raised or the bound is hit.
__device__ __forceinline__ bool wait_flag(gu32* flags, int abort_word, int* status, int idx, unsigned want) {
  const unsigned long long t0 = wall_clock64();
  for (int spin = 0;; ++spin) {
    const unsigned v = __hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= want) return true;
    if ((spin & 15) == 15) {
      if (__hip_atomic_load(flags + abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      if (wall_clock64() - t0 > kChainTimeoutTicks) {
        __hip_atomic_store(flags + abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        status[2] = 1;
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// ---- T(j; I, K): block (I, K) -= P(j; I) P(j; K)^T ------------------------------------------------------------
// Four groups of four waves, group (gr, gc) = the 64 x 64 tile at (64 gr, 64 gc) of the block with its own two LDS
// stages -- the tile body of k_gemm_mfma<ROLE_TRAILING, false, 64, 64> (LDS image [k / 4][row][4] with slot = q 64 +
// (row ^ q), one barrier per K step, fragment ping-pong; lane half h of MFMA e of group s multiplies k = 8 s + 4 h + e;
// accumulators from zero, C enters once in the epilogue: C' = fma(-1, acc, C)).  On a diagonal block the tile (0, 1)
// is not computed (the per-step launch skips tiles above the diagonal too).
template <int AUX = 16>
__device__ __attribute__((noinline)) void trail(rsrc_t ry, unsigned ldy, int j, int I, int K, int only, f32x4* lds_all, int tid,
                                                int lane, int wave) {
  constexpr int NQ = 8, TS = 64, STAGE = NQ * (TS + TS), BK = 32, NG = BK / 8;
  const int grp = wave >> 2, gr = grp >> 1, gc = grp & 1;
  // only >= 0: a quarter task (the two blocks the next step's critical path waits for are split over four workgroups,
  // one wave per SIMD each: a quarter is done in half the time of a whole block)
  const bool active = !(I == K && gr == 0 && gc == 1) && (only < 0 || only == grp);
  f32x4* lds = lds_all + grp * 2 * STAGE;
  const int t = tid & 255, w4 = wave & 3, wr = w4 >> 1, wc = w4 & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int arow0 = I * 128 + gr * 64, brow0 = K * 128 + gc * 64, kcol0 = j * 128;
  unsigned aoff[2], boff[2];
  int aslot[2], bslot[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int idx = t + 256 * p;
    const int row = idx >> 3, q = idx & 7;
    aoff[p] = (((unsigned)(arow0 + row)) * ldy + (unsigned)(kcol0 + q * 4)) * 4u;
    boff[p] = (((unsigned)(brow0 + row)) * ldy + (unsigned)(kcol0 + q * 4)) * 4u;
    aslot[p] = q * TS + (row ^ q);
    bslot[p] = q * TS + (row ^ q);
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  f4 ra[2], rb[2];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 2; ++p) { ra[p] = ld16<AUX>(ry, aoff[p] + 4u * k0); rb[p] = ld16<AUX>(ry, boff[p] + 4u * k0); }
  };
  auto store_tile = [&](int stage) {
    f32x4* As = lds + stage * STAGE;
    f32x4* Bs = As + NQ * TS;
#pragma unroll
    for (int p = 0; p < 2; ++p) { As[aslot[p]] = ra[p]; Bs[bslot[p]] = rb[p]; }
  };
  f32x4 fa[2], fb[2];
  auto read_frag = [&](int stage, int s, int buf) {
    const f32x4* As = lds + stage * STAGE;
    const f32x4* Bs = As + NQ * TS;
    const int q = 2 * s + h;
    fa[buf] = As[q * TS + ((wr * 32 + l31) ^ q)];
    fb[buf] = Bs[q * TS + ((wc * 32 + l31) ^ q)];
  };
  auto mfma_group = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][e], fb[buf][e], acc, 0, 0, 0);
  };
  // (the caller's barrier separates the previous task's LDS use from these stores)
  if (active) { load_tile(0); store_tile(0); load_tile(BK); }
  __syncthreads();
  if (active) read_frag(0, 0, 0);
  int stage = 0;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bool more = ks < 3, more2 = ks < 2;
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      if (s + 1 < NG) {
        if (active) read_frag(stage, s + 1, (s + 1) & 1);
      } else {
        __syncthreads();                         // stage ^ 1 is complete, everybody has read this stage
        if (active && more) read_frag(stage ^ 1, 0, 0);
      }
      if (active) mfma_group(s & 1);
      if (s == 0 && more && active) {
        store_tile(stage ^ 1);
        if (more2) load_tile(BK * (ks + 2));
      }
    }
    stage ^= 1;
  }
  if (!active) return;
  // epilogue: acc register e of a lane = row (e & 3) + 8 (e >> 2) + 4 h, column l31 of the wave's 32 x 32 block
  const unsigned c = (unsigned)(brow0 + wc * 32 + l31);
  const unsigned rbase = (unsigned)(arow0 + wr * 32 + 4 * h);
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = 1.f * ld4<AUX>(ry, ((rbase + (e & 3) + 8 * (e >> 2)) * ldy + c) * 4u);
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = __builtin_fmaf(-1.f, acc[e], v[e]);
#pragma unroll
  for (int e = 0; e < 16; ++e) st4<AUX>(ry, ((rbase + (e & 3) + 8 * (e >> 2)) * ldy + c) * 4u, v[e]);
}

// ---- P(j; I): rows of block I of column block j  <-  P Linv_jj^T, in place -------------------------------------
// k_panel_direct's arithmetic (v_mfma_f32_16x16x4_f32; lane (lr, lq) of step (u, e) multiplies k = 16 u + 4 lq + e on
// both operands; column tile ct needs k < 16 (ct + 1)).  Under the persistent chain the Dinv buffer holds Linv_jj
// TRANSPOSED (ZT[k][n] = Linv[n][k]: the rows of Z the critical workgroup has in its image, 1 / l_kk on the diagonal, zeros
// below -- written with whole-line 16-byte stores), staged as it is and read one scalar per MFMA operand.
// Wave w: rows 16 (w & 7) .., column tiles {0, 3, 4, 7} (w < 8) or {1, 2, 5, 6}: 18 k blocks each.  Two waves share a
// row range: every A fragment is in registers before the barrier that precedes the first store.  The result goes
// through an LDS image ([k / 4][row][4]) so that the rows leave as whole lines.
__device__ __forceinline__ int panel_ct(int ch, int c4) { return ch == 0 ? (c4 == 0 ? 0 : (c4 == 1 ? 3 : (c4 == 2 ? 4 : 7))) : (c4 == 0 ? 1 : (c4 == 1 ? 2 : (c4 == 2 ? 5 : 6))); }
__device__ __forceinline__ void panel_image_and_rows(rsrc_t ry, unsigned ldy, int j, int I, const f4 (&acc)[4], float* pimgf,
                                                     int rs, int ch, int lr, int lq, int tid) {
  const f32x4* pimg = reinterpret_cast<const f32x4*>(pimgf);
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = rs * 16 + 4 * lq + e, k = 16 * panel_ct(ch, c4) + lr, kq = k >> 2;
      pimgf[(kq * 128 + (rl ^ (kq & 7))) * 4 + (k & 3)] = acc[c4][e];
    }
  __syncthreads();                                   // the image is complete
#pragma unroll
  for (int p = 0; p < 4; ++p) {                      // 16 bytes of a row per lane, whole lines per wave
    const int idx = tid + 1024 * p, row = idx >> 5, kq = idx & 31;
    st16(ry, (((unsigned)(I * 128 + row)) * ldy + (unsigned)(j * 128 + 4 * kq)) * 4u, pimg[kq * 128 + (row ^ (kq & 7))]);
  }
}
__device__ __attribute__((noinline)) void panel(rsrc_t ry, rsrc_t rd, unsigned ldy, int j, int I, unsigned char* smem, int tid,
                                                int lane, int wave) {
  constexpr int NB = 128, PITCH = 132;
  float* sl = reinterpret_cast<float*>(smem);
  const int lr = lane & 15, lq = lane >> 4;
  const int rs = wave & 7, ch = wave >> 3;
  const unsigned row0 = (unsigned)(I * 128 + rs * 16);
  f4 fa[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) fa[u] = ld16(ry, ((row0 + lr) * ldy + (unsigned)(j * 128 + 16 * u + 4 * lq)) * 4u);
  f4 v[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int q = tid + 1024 * p;                // row k = q / 32, columns n = 4 (q % 32) ..
    v[p] = ld16(rd, ((unsigned)j * NB * NB + (unsigned)(q >> 5) * NB + 4u * (q & 31)) * 4u);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int q = tid + 1024 * p;
    *reinterpret_cast<f4*>(sl + (q >> 5) * PITCH + 4 * (q & 31)) = v[p];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's rows are in registers
  __syncthreads();
  f4 acc[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) acc[c4] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 8; ++u) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const int ct = panel_ct(ch, c4);
      if (u <= ct) {
        f4 fb;
#pragma unroll
        for (int e = 0; e < 4; ++e) fb[e] = sl[(16 * u + 4 * lq + e) * PITCH + 16 * ct + lr];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c4] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][e], fb[e], acc[c4], 0, 0, 0);
      }
    }
  }
  panel_image_and_rows(ry, ldy, j, I, acc, reinterpret_cast<float*>(smem + kChainPimg), rs, ch, lr, lq, tid);
}

// ---- the critical path: D(j) -> P(j; j+1) -> T(j; j+1, j+1) -> D(j+1) ... inside ONE workgroup, in LDS -------------------
// The diagonal block never leaves the LDS image between its trailing update and its factorisation; P(j; j+1) takes
// Linv_jj straight from the image (Z = Linv^T in the strict upper triangle: the values diag_store_lds writes to Dinv) and
// leaves its result in a second image (the [k / 4][row][4] layout of the tile GEMM), from which T(j; j+1, j+1) takes both
// operands; only the 10 lower 32 x 32 blocks of the diagonal block are computed (the factorisation reads nothing else).
// The sums are those of k_chol_diag_packed / k_panel_direct / k_gemm_mfma<TRAILING>: same bits.  What other workgroups
// need -- L_jj, Dinv_j, the rows of P(j; j+1) -- goes to global memory write-through and is published behind one drain.
// (Three functions, not inlined into each other: each gets its own register allocation -- the factorisation's inner
// loop must not carry the spills of the panel's fragments.)
struct CritEnv {                                 // what the pieces share (passed by value: scalars)
  unsigned ldy; int m; int* status; gu32* flags; int abort_word; unsigned epoch; int nblk, rb, uflag0;
  unsigned* trace; int trace_cap;
};
__device__ __forceinline__ int crit_pflag(const CritEnv& c, int j, int I) { return c.nblk + j * c.rb + I; }
__device__ __forceinline__ int crit_tflag(const CritEnv& c, int I, int K) { return c.nblk + c.nblk * c.rb + I * c.nblk + K; }
__device__ __forceinline__ void crit_rec(const CritEnv& c, int type, int j, int I, int K, const unsigned* ts) {
  if (c.trace && threadIdx.x == 0) {
    const unsigned r = atomicAdd(c.trace, 1u);
    if ((int)r < c.trace_cap) {
      unsigned* o = c.trace + 8 + 8 * (size_t)r;
      o[0] = (unsigned)type | (blockIdx.x << 8) | (1u << 24);
      o[1] = (unsigned)j; o[2] = (unsigned)I; o[3] = (unsigned)K;
      o[4] = ts[0]; o[5] = ts[1]; o[6] = ts[2]; o[7] = (unsigned)wall_clock64();
    }
  }
}
// one time stamp of wave 0 (trace only): type 16 + code
__device__ __forceinline__ void crit_mark(const CritEnv& c, int code, int j) {
  if (c.trace && threadIdx.x == 0) {
    const unsigned t = (unsigned)wall_clock64();
    const unsigned r = atomicAdd(c.trace, 1u);
    if ((int)r < c.trace_cap) {
      unsigned* o = c.trace + 8 + 8 * (size_t)r;
      o[0] = (unsigned)(16 + code) | (blockIdx.x << 8) | (1u << 24);
      o[1] = (unsigned)j; o[2] = 0; o[3] = 0; o[4] = o[5] = o[6] = o[7] = t;
    }
  }
}
// two polls by one lane, the verdict through an LDS word (the caller's barrier follows)
__device__ __forceinline__ void crit_poll2(const CritEnv& c, int* ctl, int i0, unsigned n0, int i1, unsigned n1) {
  if (threadIdx.x == 0) {
    int ok = 1;
    if (i0 >= 0 && !wait_flag(c.flags, c.abort_word, c.status, i0, c.epoch | n0)) ok = 0;
    if (ok && i1 >= 0 && !wait_flag(c.flags, c.abort_word, c.status, i1, c.epoch | n1)) ok = 0;
    ctl[14] = ok;
  }
}
// T(.; blk, blk): the lower 32 x 32 blocks, wave w < 10 = block (bi, bj); both operands from the image of the panel rows,
// C in `cv` (requested earlier with crit_load_c); the result goes into the diagonal block's LDS image, zeros above the diagonal
template <int AUX = 16>
__device__ __forceinline__ void crit_load_c(rsrc_t ry, unsigned ldy, int blk, int wave, int lane, float (&cv)[16]) {
  const int h = lane >> 5, l31 = lane & 31;
  const int bi = wave >= 6 ? 3 : (wave >= 3 ? 2 : (wave >= 1 ? 1 : 0)), bj = wave - bi * (bi + 1) / 2;
  if (wave < 10) {
    const unsigned c = (unsigned)(blk * 128 + 32 * bj + l31), r0 = (unsigned)(blk * 128 + 32 * bi + 4 * h);
#pragma unroll
    for (int e = 0; e < 16; ++e) cv[e] = ld4<AUX>(ry, ((r0 + (e & 3) + 8 * (e >> 2)) * ldy + c) * 4u);
  }
}
__device__ __forceinline__ void crit_trail_diag(const f32x4* pimg, float* a, int wave, int lane, const float (&cv)[16]) {
  constexpr int LDA = 132;
  const int h = lane >> 5, l31 = lane & 31;
  if (wave < 10) {
    const int bi = wave >= 6 ? 3 : (wave >= 3 ? 2 : (wave >= 1 ? 1 : 0)), bj = wave - bi * (bi + 1) / 2;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        const int q = 2 * sg + h, kq = ks * 8 + q;
        const f32x4 fa = pimg[kq * 128 + ((32 * bi + l31) ^ q)];
        const f32x4 fb = pimg[kq * 128 + ((32 * bj + l31) ^ q)];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
      }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = 32 * bi + (e & 3) + 8 * (e >> 2) + 4 * h, col = 32 * bj + l31;
      const float v = __builtin_fmaf(-1.f, acc[e], 1.f * cv[e]);
      a[row * LDA + col] = (col <= row) ? v : 0.f;
    }
  } else {
    // the six blocks above the diagonal: zeros (the factorisation builds Z there)
    const int u = wave - 10;
    const int zi = u < 3 ? 0 : (u < 5 ? 1 : 2), zj = u < 3 ? u + 1 : (u < 5 ? u - 1 : 3);
#pragma unroll
    for (int r = 0; r < 16; ++r) a[(32 * zi + 16 * h + r) * LDA + 32 * zj + l31] = 0.f;
  }
}

// Launch start: block (s0, s0) into the image -- from global memory, or (deferred) as T(s0 - 1; s0, s0) from the panel rows
// the previous launch published.
__device__ __attribute__((noinline)) bool crit_head(rsrc_t ry, CritEnv c, int s0, int deferred, unsigned char* smem) {
  float* a = reinterpret_cast<float*>(smem);
  f32x4* pimg = reinterpret_cast<f32x4*>(smem + kChainPimg);     // slot = kq 128 + (row ^ (kq & 7)), kq = k / 4
  int* ctl = reinterpret_cast<int*>(smem + kChainCtl);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned ts[3] = {0, 0, 0};
  if (c.trace && tid == 0) ts[0] = (unsigned)wall_clock64();
  if (deferred) {
    // (its inputs -- the rows of P(s0 - 1; s0), block (s0, s0) with the update of step s0 - 2 -- are complete: the
    // previous launch has ended)
    if (c.trace && tid == 0) ts[1] = (unsigned)wall_clock64();
    f4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = tid + 1024 * p, row = idx >> 5, kq = idx & 31;
      v[p] = ld16(ry, (((unsigned)(s0 * 128 + row)) * c.ldy + (unsigned)((s0 - 1) * 128 + 4 * kq)) * 4u);
    }
    float cv[16];
    crit_load_c(ry, c.ldy, s0, wave, lane, cv);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = tid + 1024 * p, row = idx >> 5, kq = idx & 31;
      pimg[kq * 128 + (row ^ (kq & 7))] = v[p];
    }
    __syncthreads();
    crit_trail_diag(pimg, a, wave, lane, cv);
    if (c.trace && tid == 0) ts[2] = (unsigned)wall_clock64();
    crit_rec(c, CT_TRAIL, s0 - 1, s0, s0, ts);
  } else {
    const unsigned base = ((unsigned)s0 * 128u * c.ldy + (unsigned)s0 * 128u) * 4u;
    diag_load_lds([&](int i, int j0) { return ld16(ry, base + ((unsigned)i * c.ldy + (unsigned)j0) * 4u); }, a);
  }
  return true;
}

// D(j) inside the image (starts with a barrier: the image is complete).
__device__ __attribute__((noinline)) void crit_factor(int* status, int nblk_real, unsigned char* smem) {
  float* a = reinterpret_cast<float*>(smem);
  const DiagLds L{a, reinterpret_cast<float(*)[16 * 20]>(a + 128 * 132),
                  reinterpret_cast<float(*)[16]>(a + 128 * 132 + 2 * 16 * 20), a + 128 * 132 + 2 * 16 * 20 + 2 * 16};
  diag_factor_lds<7>(status, nblk_real, L);
  __builtin_amdgcn_s_setprio(2);
}

// Behind D(j): L_jj and Linv_jj^T out, P(j; j+1), T(j; j+1, j+1) into the image.  Returns false when a wait gave up.
// Nothing on this path waits for a store it has just issued: L_jj / ZT go out first and are drained behind the MFMAs of
// P, in front of the barrier the image needs anyway (dflag(j) is published there: the other panels of step j start
// while this workgroup is still in T); the rows of P go out behind that barrier and are drained behind the MFMAs of T, in
// front of the barrier the next factorisation needs anyway (pflag(j, j+1)).
__device__ __attribute__((noinline)) bool crit_tail(rsrc_t ry, rsrc_t rd, CritEnv c, int j, int has_next, int do_t,
                                                    unsigned char* smem, unsigned t_start) {
  constexpr int LDA = 132, NT = 1024;
  float* a = reinterpret_cast<float*>(smem);
  f32x4* pimg = reinterpret_cast<f32x4*>(smem + kChainPimg);
  float* pimgf = reinterpret_cast<float*>(smem + kChainPimg);
  int* ctl = reinterpret_cast<int*>(smem + kChainCtl);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const unsigned ldy = c.ldy;
  unsigned ts[3] = {t_start, 0, 0};
  if (c.trace && tid == 0) ts[1] = (unsigned)wall_clock64();
  {
    // L_jj (lower triangle, zeros above) into Y and ZT = Linv_jj^T (Z above the diagonal, 1 / l_kk on it, zeros below)
    // into the Dinv slot of this step: one 16-byte LDS read of a row piece gives both, each a coalesced 16-byte store
    const unsigned base = ((unsigned)j * 128u * ldy + (unsigned)j * 128u) * 4u, dbase = (unsigned)j * 128u * 128u * 4u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int q = tid + NT * p;
      const int i = q >> 5, j0 = 4 * (q & 31);
      const f4 l = *reinterpret_cast<const f4*>(a + i * LDA + j0);
      f4 lo, zt;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo[e] = (j0 + e <= i) ? l[e] : 0.f;
        zt[e] = (j0 + e > i) ? l[e] : ((j0 + e == i) ? 1.f / l[e] : 0.f);
      }
      st16(ry, base + ((unsigned)i * ldy + (unsigned)j0) * 4u, lo);
      st16(rd, dbase + ((unsigned)i * 128u + (unsigned)j0) * 4u, zt);
    }
  }
  crit_mark(c, 0, j);
  // P(j; j+1) works on block (j+1, j), T(j; j+1, j+1) on block (j+1, j+1): both must carry the update of step j - 1, which
  // other workgroups apply quarter by quarter (CT_TRAILQ): seven words, seven lanes
  f4 fa[8];
  float cv[16];
  const int rs = wave & 7, ch = wave >> 3;
  const unsigned row0 = (unsigned)((j + 1) * 128 + rs * 16);
  if (has_next && j > 0) {
    if (tid < 8) {
      int ok = 1;
      const bool mine = tid < 4 || (do_t && tid != 5);
      if (mine) ok = wait_flag(c.flags, c.abort_word, c.status, c.uflag0 + (j - 1) * 8 + tid, c.epoch | 1u) ? 1 : 0;
      ctl[16 + tid] = ok;
    }
    __syncthreads();
    int ok = 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) ok &= ctl[16 + i];
    if (!ok) return false;
  }
  crit_mark(c, 1, j);
  if (has_next) {
#pragma unroll
    for (int u = 0; u < 8; ++u) fa[u] = ld16(ry, ((row0 + lr) * ldy + (unsigned)(j * 128 + 16 * u + 4 * lq)) * 4u);
    if (do_t) crit_load_c(ry, ldy, j + 1, wave, lane, cv);
  }
  if (c.trace && tid == 0) ts[2] = (unsigned)wall_clock64();
  crit_rec(c, CT_DIAG, j, j, j, ts);
  if (has_next) {
    if (c.trace && tid == 0) ts[0] = ts[1] = (unsigned)wall_clock64();
    // P(j; j+1): k_panel_direct's products with Linv[n][k] read from the image: Z[k][n] above the diagonal, 1 / l_nn on
    // it.  The four column tiles of a wave advance together (four independent accumulator chains); per element the
    // order is still u ascending, e ascending.
    f4 acc[4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) acc[c4] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const int ct = panel_ct(ch, c4);
        if (u <= ct) {
          const int n = 16 * ct + lr;
          f4 fb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = 16 * u + 4 * lq + e;
            float x = a[k * LDA + n];
            if (u == ct) x = (k < n) ? x : ((k == n) ? 1.f / x : 0.f);
            fb[e] = x;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[c4] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][e], fb[e], acc[c4], 0, 0, 0);
        }
      }
    }
    crit_mark(c, 2, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // L_jj / ZT are out (and C is in)
    // image, barrier (nobody reads Z any more), rows out; dflag(j) right behind the barrier
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rl = rs * 16 + 4 * lq + e, k = 16 * panel_ct(ch, c4) + lr, kq = k >> 2;
        pimgf[(kq * 128 + (rl ^ (kq & 7))) * 4 + (k & 3)] = acc[c4][e];
      }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(c.flags + j, c.epoch | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // dflag(j)
    crit_mark(c, 3, j);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = tid + 1024 * p, row = idx >> 5, kq = idx & 31;
      st16(ry, (((unsigned)((j + 1) * 128 + row)) * ldy + (unsigned)(j * 128 + 4 * kq)) * 4u, pimg[kq * 128 + (row ^ (kq & 7))]);
    }
    if (c.trace && tid == 0) ts[2] = (unsigned)wall_clock64();
    crit_rec(c, CT_PANEL, j, j + 1, j, ts);
    if (do_t) {
      if (c.trace && tid == 0) ts[0] = ts[1] = (unsigned)wall_clock64();
      crit_trail_diag(pimg, a, wave, lane, cv);
      if (c.trace && tid == 0) ts[2] = (unsigned)wall_clock64();
      crit_rec(c, CT_TRAIL, j, j + 1, j + 1, ts);
    }
  }
  crit_mark(c, 4, j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every store of this step is out
  __syncthreads();
  crit_mark(c, 5, j);
  if (tid == 0) {
    if (has_next)
      __hip_atomic_store(c.flags + crit_pflag(c, j, j + 1), c.epoch | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      __hip_atomic_store(c.flags + j, c.epoch | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // dflag(j)
  }
  return true;
}
}  // namespace chain

__global__ void __launch_bounds__(1024) k_chain_persistent(ChainArgs g) {
  using namespace chain;
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  int* ctl = reinterpret_cast<int*>(chain_smem + kChainCtl);     // [0 .. 11] the task, [12] go, [13] role ticket, [14] verdict of a poll
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(g.Y, 0, (int)g.y_bytes, 0x27000);
  const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(g.Dinv, 0, (int)g.dinv_bytes, 0x27000);
  gu32* const flags = (gu32*)g.flags;
  const unsigned ldy = (unsigned)g.ldy;
  __builtin_amdgcn_s_setprio(2);
  if (tid == 0) ctl[13] = atomicAdd(&g.counters[1], 1);
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(ctl[13]) == 0) {            // the first workgroup to arrive walks the critical path ...
    const CritEnv c{ldy, g.m, g.status, flags, g.abort_word, g.epoch, g.nblk, g.rb, g.nblk + 2 * g.nblk * g.rb, g.trace, g.trace_cap};
    if (!crit_head(ry, c, g.s0, g.deferred, chain_smem)) return;
    for (int j = g.s0; j < g.s1; ++j) {
      unsigned t0 = 0;
      if (g.trace && tid == 0) t0 = (unsigned)wall_clock64();
      crit_factor(g.status, max(1, min(8, (g.m - j * 128 + 15) / 16)), chain_smem);
      const int has_next = j + 1 < g.nblk, do_t = has_next && j + 1 < g.s1;
      if (!crit_tail(ry, rd, c, j, has_next, do_t, chain_smem, t0)) return;
    }
  }
  for (;;) {                                                      // ... then joins the others on the bulk list
    __syncthreads();                             // everybody is done with ctl and with the LDS of the previous task
    unsigned tr0 = 0, tr1 = 0, tr2 = 0;
    if (g.trace && tid == 0) tr0 = (unsigned)wall_clock64();
    if (tid == 0) {
      const ChainTask* src = nullptr;
      const int k = atomicAdd(&g.counters[0], 1);
      if (k < g.nbulk) src = g.bulk + k;
      int ok = src ? 1 : 0;
      if (src) {
        gci32* s4 = (gci32*)reinterpret_cast<const int*>(src);     // (the lists are written by the host before the first launch)
        int tv[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) tv[i] = s4[i];
#pragma unroll
        for (int i = 0; i < 12; ++i) ctl[i] = tv[i];
#pragma unroll
        for (int d = 0; d < 3; ++d)
          if (ok && tv[4 + d] >= 0 && !wait_flag(flags, g.abort_word, g.status, tv[4 + d], g.epoch | (unsigned)tv[7 + d])) ok = 0;
      }
      ctl[12] = ok;
    }
    __syncthreads();
    if (!__builtin_amdgcn_readfirstlane(ctl[12])) break;
    const int type = __builtin_amdgcn_readfirstlane(ctl[0]), j = __builtin_amdgcn_readfirstlane(ctl[1]);
    const int I = __builtin_amdgcn_readfirstlane(ctl[2]), K = __builtin_amdgcn_readfirstlane(ctl[3]);
    const int pub = __builtin_amdgcn_readfirstlane(ctl[10]), val = __builtin_amdgcn_readfirstlane(ctl[11]);
    if (g.trace && tid == 0) tr1 = (unsigned)wall_clock64();
    if (type == CT_PANEL) {
      panel(ry, rd, ldy, j, I, chain_smem, tid, lane, wave);
    } else {
      trail(ry, ldy, j, I, K, type >= CT_TRAILQ ? type - CT_TRAILQ : -1, reinterpret_cast<f32x4*>(chain_smem), tid, lane, wave);
    }
    if (g.trace && tid == 0) tr2 = (unsigned)wall_clock64();
    // publish: every storing wave drains its stores, the workgroup meets, ONE lane stores the flag (sc1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && pub >= 0)
      __hip_atomic_store(flags + pub, g.epoch | (unsigned)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g.trace && tid == 0) {
      const unsigned r = atomicAdd(g.trace, 1u);
      if ((int)r < g.trace_cap) {
        unsigned* o = g.trace + 8 + 8 * (size_t)r;
        o[0] = (unsigned)type | (blockIdx.x << 8);
        o[1] = (unsigned)j; o[2] = (unsigned)I; o[3] = (unsigned)K;
        o[4] = tr0; o[5] = tr1; o[6] = tr2; o[7] = (unsigned)wall_clock64();
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Trailing update of block step j AND the diagonal factor of step j + 1 as ONE launch (round 6; the default chain): the
// per-step sequence was diag(j) -> panel(j) -> trailing(j) -> diag(j + 1) ..., three launches of which the factor -- one
// workgroup, 17 us -- had the chip to itself.  Here workgroup 0 first applies step j's update to the diagonal block
// (j + 1, j + 1) ITSELF -- its operands, the rows of P(j; j + 1), are final before the launch: no hand-over inside it --
// straight into the factor's LDS image, factors it and writes L_(j+1) and Dinv_(j+1), while every other workgroup takes one
// 128 x 128 block of the trailing update (four 64 x 64 tiles of k_gemm_mfma<TRAILING>'s arithmetic).  Same sums, same
// bits (chain::trail, chain::crit_trail_diag: the pieces of the persistent kernel above, with ordinary loads and stores:
// the hand-overs are launch boundaries).  A step costs panel + max(update + factor, trailing) instead of their sum.
// ---------------------------------------------------------------------------------------------------------------------
struct TrailDiagArgs {
  float* Y; int ldy; unsigned y_bytes;
  float* Dinv; unsigned dinv_bytes;
  int* status;
  int m;                                         // real rows of S
  int j;                                         // the step whose trailing update this is
  const int* blocks; int nblocks;                // (I, K) pairs: the blocks of the update EXCEPT (j + 1, j + 1)
  int do_diag;                                   // 1: workgroup 0 = block (j + 1, j + 1) + the factor of step j + 1
};

__global__ void __launch_bounds__(1024) k_trail_diag(TrailDiagArgs g) {
  using namespace chain;
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(g.Y, 0, (int)g.y_bytes, 0x27000);
  const unsigned ldy = (unsigned)g.ldy;
  __builtin_amdgcn_s_setprio(2);
  if (g.do_diag && blockIdx.x == 0) {
    const int j = g.j, blk = g.j + 1;
    float* a = reinterpret_cast<float*>(chain_smem);
    f32x4* pimg = reinterpret_cast<f32x4*>(chain_smem + kChainPimg);
    // the rows of P(j; j + 1) -> the operand image (slot = kq 128 + (row ^ (kq & 7)), kq = k / 4); C = block (blk, blk)
    f4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = tid + 1024 * p, row = idx >> 5, kq = idx & 31;
      v[p] = ld16<0>(ry, (((unsigned)(blk * 128 + row)) * ldy + (unsigned)(j * 128 + 4 * kq)) * 4u);
    }
    float cv[16];
    crit_load_c<0>(ry, ldy, blk, wave, lane, cv);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = tid + 1024 * p, row = idx >> 5, kq = idx & 31;
      pimg[kq * 128 + (row ^ (kq & 7))] = v[p];
    }
    __syncthreads();
    crit_trail_diag(pimg, a, wave, lane, cv);
    const DiagLds L{a, reinterpret_cast<float(*)[16 * 20]>(a + 128 * 132),
                    reinterpret_cast<float(*)[16]>(a + 128 * 132 + 2 * 16 * 20), a + 128 * 132 + 2 * 16 * 20 + 2 * 16};
    diag_factor_lds<7>(g.status, max(1, min(8, (g.m - blk * 128 + 15) / 16)), L);      // (starts with a barrier)
    float* Ab = g.Y + (size_t)blk * 128 * g.ldy + (size_t)blk * 128;
    float* Db = g.Dinv + (size_t)blk * 128 * 128;
    const int ld_ = g.ldy;
    diag_store_lds([&](int i, int j0, const f4& x) { *reinterpret_cast<f4*>(Ab + (size_t)i * ld_ + j0) = x; },
                   [&](int i, int jj, float x) { Db[(size_t)i * 128 + jj] = x; }, a);
    return;
  }
  const int t = (int)blockIdx.x - g.do_diag;
  if (t >= g.nblocks) return;
  const int I = g.blocks[2 * t], K = g.blocks[2 * t + 1];
  trail<0>(ry, ldy, g.j, I, K, -1, reinterpret_cast<f32x4*>(chain_smem), tid, lane, wave);
}

// ---- host: the task lists of every launch of a chunk plan -------------------------------------------------------
struct ChainPlan {
  int nblk = 0, nchunks = 0, cend[8] = {};
  int rb = 0;                                    // row blocks incl. the widest strip
  int nflags = 0;                                // hand-over words (+ 1: the abort word)
  std::vector<ChainTask> tasks;                  // the bulk lists, launch after launch
  struct Launch { int bulk_off, nbulk; } launch[8] = {};
  // what the critical workgroup does (crit_head / crit_factor / crit_tail), as (waits, publishes) per step: only for
  // the host-side proof that the lists complete
  struct CritStep { std::vector<int> wait; std::vector<int> pub; };
  std::vector<CritStep> crit[8];
  int dflag(int j) const { return j; }
  int pflag(int j, int I) const { return nblk + j * rb + I; }
  int tflag(int I, int K) const { return nblk + nblk * rb + I * nblk + K; }
  int uflag(int j, int t, int q) const { return nblk + 2 * nblk * rb + j * 8 + t * 4 + q; }   // quarter q of urgent block t of step j
};

// Chunk g = block steps [s0, s1).  Launch g runs the trailing update of step s0 - 1 first (the last step of chunk g - 1:
// it only touches columns >= s0, so chunk g - 1 is complete -- and its event can be recorded -- without it), then for
// every step of the chunk the diagonal factor, the panels and -- except for the chunk's last step -- the trailing update.
// Per step j the critical workgroup owns D(j), P(j; j+1), T(j; j+1, j+1); the bulk list holds the rest, in this order:
// P(j; j+2); the two blocks the next step's critical tasks wait for -- (j+2, j+1) and (j+2, j+2) -- as seven quarter
// tasks; the other panels; the other blocks column by column (column j+1 feeds the next step's panels), the strip blocks
// of a column behind its S blocks.
inline void build_chain_plan(ChainPlan& P, int nblk, int nchunks, const int* cend) {
  P.nblk = nblk;
  P.nchunks = nchunks;
  int widest = 0;
  for (int g = 0; g < nchunks; ++g) { P.cend[g] = cend[g]; widest = std::max(widest, cend[g] - (g ? cend[g - 1] : 0)); }
  P.rb = nblk + widest;
  P.nflags = nblk + 2 * nblk * P.rb + 8 * nblk + 1;
  P.tasks.clear();
  auto mk = [&](int type, int j, int I, int K) {
    ChainTask t{type, j, I, K, {-1, -1, -1}, {0, 0, 0}, -1, 0};
    return t;
  };
  for (int g = 0; g < nchunks; ++g) {
    const int s0 = g ? cend[g - 1] : 0, s1 = cend[g];
    std::vector<ChainTask> bulk;
    P.crit[g].clear();
    // strip row block t of this chunk (rows m_pad + 128 t ..) is the identity of column block s0 + t: first touched at step s0 + t
    auto pan = [&](int j, int I) {
      ChainTask t = mk(CT_PANEL, j, I, j);
      t.dep[0] = P.dflag(j); t.need[0] = 1;
      const int first = (I >= nblk) ? s0 + (I - nblk) : 0;          // step before which block (I, j) has never been updated
      if (j > first) { t.dep[1] = P.tflag(I, j); t.need[1] = j; }
      t.pub = P.pflag(j, I); t.val = 1;
      return t;
    };
    auto trl = [&](int j, int I, int K) {
      ChainTask t = mk(CT_TRAIL, j, I, K);
      t.dep[0] = P.pflag(j, I); t.need[0] = 1;
      if (K != I) { t.dep[1] = P.pflag(j, K); t.need[1] = 1; }
      const int first = (I >= nblk) ? s0 + (I - nblk) : 0;
      if (j > first) { t.dep[2] = P.tflag(I, K); t.need[2] = j; }
      t.pub = P.tflag(I, K); t.val = j + 1;
      return t;
    };
    // the two urgent blocks of step j (row block j + 2): quarter tasks, each publishing a word of its own (only the
    // critical workgroup reads these blocks again)
    auto urgent = [&](int j, std::vector<ChainTask>& out) {
      if (j + 2 >= nblk) return;
      for (int t = 0; t < 2; ++t)
        for (int q = 0; q < 4; ++q) {
          if (t == 1 && q == 1) continue;                             // diagonal block: nothing above the diagonal
          ChainTask x = trl(j, j + 2, j + 1 + t);
          x.type = CT_TRAILQ + q;
          x.pub = P.uflag(j, t, q); x.val = 1;
          out.push_back(x);
        }
    };
    auto others = [&](int j, int strip_hi, std::vector<ChainTask>& out) {
      for (int K = j + 1; K < nblk; ++K) {
        for (int I = K; I < nblk; ++I) {
          if (I == j + 1) continue;                                   // (j+1, j+1): the critical workgroup's
          if (I == j + 2 && K <= j + 2) continue;                     // the urgent blocks
          out.push_back(trl(j, I, K));
        }
        if (K < strip_hi)
          for (int t = 0; t <= j - s0; ++t) out.push_back(trl(j, nblk + t, K));
      }
    };
    ChainPlan::CritStep head;
    if (g > 0 && s0 < nblk) {                                          // deferred from chunk g - 1: S blocks only
      urgent(s0 - 1, bulk);
      others(s0 - 1, 0, bulk);
    }
    P.crit[g].push_back(head);                                         // (the head waits for nothing inside the launch)
    for (int j = s0; j < s1; ++j) {
      ChainPlan::CritStep cs;
      const bool has_next = j + 1 < nblk, do_t = has_next && j + 1 < s1;
      if (has_next && j > 0) {
        for (int q = 0; q < 4; ++q) cs.wait.push_back(P.uflag(j - 1, 0, q));
        if (do_t) for (int q = 0; q < 4; ++q) if (q != 1) cs.wait.push_back(P.uflag(j - 1, 1, q));
      }
      cs.pub.push_back(P.dflag(j));
      if (has_next) cs.pub.push_back(P.pflag(j, j + 1));
      P.crit[g].push_back(cs);
      if (j + 2 < nblk) bulk.push_back(pan(j, j + 2));
      if (do_t) urgent(j, bulk);
      for (int I = j + 3; I < nblk; ++I) bulk.push_back(pan(j, I));
      for (int t = 0; t <= j - s0; ++t) bulk.push_back(pan(j, nblk + t));
      if (do_t) others(j, s1, bulk);
    }
    P.launch[g].bulk_off = (int)P.tasks.size();
    P.launch[g].nbulk = (int)bulk.size();
    P.tasks.insert(P.tasks.end(), bulk.begin(), bulk.end());
  }
}

// Runs every launch of the plan with ONE worker per list, each strictly in list order (the most restrictive schedule a
// launch can meet: two resident workgroups).  Returns false if a task waits for something that is never published.
inline bool validate_chain_plan(const ChainPlan& P) {
  std::vector<int> flag(P.nflags, 0);
  for (int g = 0; g < P.nchunks; ++g) {
    const ChainPlan::Launch& L = P.launch[g];
    const std::vector<ChainPlan::CritStep>& C = P.crit[g];
    size_t ic = 0;
    int ib = 0;
    auto ready = [&](const ChainTask& t) {
      for (int d = 0; d < 3; ++d)
        if (t.dep[d] >= 0 && flag[t.dep[d]] < t.need[d]) return false;
      return true;
    };
    for (;;) {
      bool progress = false;
      if (ic < C.size()) {
        bool ok = true;
        for (int w : C[ic].wait) ok = ok && flag[w] >= 1;
        if (ok) { for (int f : C[ic].pub) flag[f] = std::max(flag[f], 1); ++ic; progress = true; }
      }
      if (ib < L.nbulk && ready(P.tasks[L.bulk_off + ib])) {
        const ChainTask& t = P.tasks[L.bulk_off + ib];
        if (t.pub >= 0) flag[t.pub] = std::max(flag[t.pub], t.val);
        ++ib;
        progress = true;
      }
      if (ic == C.size() && ib == L.nbulk) break;
      if (!progress) return false;
    }
  }
  return true;
}

}  // namespace ekf
