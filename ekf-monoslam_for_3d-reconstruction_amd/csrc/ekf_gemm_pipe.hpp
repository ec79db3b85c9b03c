// Persistent tile GEMM with a CROSS-TILE software pipeline (round 5): the queued launches of the EKF update --
// the covariance downdate Sigma -= V_g V_g^T (a10, vR.cpp:1279) and the triangular solve V_g = W_g Z_gg (a9,
// vR.cpp:1276-1278) -- drawn tile by tile from a host-ordered list by two workgroups per CU.
//
// Same arithmetic as k_gemm_mfma (ekf_dense.hpp): 128 x 128 x 32 tile (64 x 128 half tiles at the end of a downdate list), four
// waves as 2 x 2, v_mfma_f32_32x32x2_f32, the XOR-swizzled LDS images, two LDS stages with one barrier per K step,
// fragment ping-pong, accumulators from zero and C entering once in the epilogue -- every output element is the
// same chain of fp32 operations in the same order, so the results are BIT-IDENTICAL to k_gemm_mfma's.
//
// What is different is what happens BETWEEN tiles.  k_gemm_mfma drains its pipeline per tile: two barriers and an
// atomic round trip for the next tile index, a dependent load of the list entry, the first operand stage fetched
// with nothing else in flight, and an epilogue that requests C only after the last MFMA.  With 12-16 K steps per
// tile (column chunks of 384 / 512) that fixed part is a third of the tile.  Here
//   * the next ticket is drawn by thread 0 during K step 0 of the CURRENT tile, the list entry looked up during step 1
//     and handed to the other waves through two LDS words during step 2 (the barriers of the K loop order it);
//   * the first operand stage of the NEXT tile is requested in the last K step of this one (the staging registers are
//     free by then) and travels under the epilogue;
//   * the C tile is requested 32 x 32 block by block, two blocks ahead: block 0 before the second-to-last K step,
//     block 1 before the last one, block b + 2 when block b has been stored;
//   * no barrier between tiles: after the barrier inside the last K step no wave reads the LDS stages again.
// Contract: queued launches only (tile_map != nullptr), K >= 128 (four K steps: the hand-over above needs them), every
// listed tile is computed (no skipping of upper tiles, no zrow), tri in {0, 2, 3}.
#pragma once
#include "ekf_dense.hpp"

namespace ekf {

template <int ROLE, bool BT>
__global__ void __launch_bounds__(256, 2) k_gemm_pipe(GemmArgs g) {
  constexpr int TN = 128, NT = 256, BK = 32, NQ = BK / 4, NJ = 2, NG = BK / 8;
  constexpr bool NTD = (ROLE == ROLE_DOWNDATE) && !BT;          // half tiles, the second product, mirrored tiles
  __shared__ f32x4 lds[2 * NQ * (128 + TN)];                     // two stages of {A image, B image}
  __shared__ int s_next[2];                                      // hand-over of the next list entry (raw bi, bj)
  constexpr int STAGE = NQ * (128 + TN);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, h = lane >> 5, l31 = lane & 31;
  const float alpha = float(g.alpha), beta = float(g.beta);
  const float* const A = static_cast<const float*>(g.A);
  const int lda = g.lda;
  const int row0 = tid >> 3, q0 = tid & 7;                       // NT staging: 8 lanes cover 128 B of a row
  const int qk = tid >> 5, cq = tid & 31;                        // NN staging of B: k quad (8) x column quad (32)
  const unsigned voffA = unsigned(row0 * lda + q0 * 4) * 4u;     // byte offset of this lane's first A quad inside a tile

  // One list entry, decoded.  Everything here is workgroup-uniform (scalar registers).
  struct Tile {
    int arow;              // first row of the tile in A and C
    int half;              // 64 x 128 half tile
    int bj;                // column tile
    const float* B; int ldb;
    float* C; int ldc;
    int mirror;            // also store the transposed tile
    int nk;                // K steps
  };
  auto decode = [&](int rbi, int rbj) {
    Tile t;
    const bool second = NTD && (rbj & kSecondProduct);
    t.bj = rbj & 0xffff;
    const bool lm = NTD && (rbi & kMirrorTile);
    t.half = (NTD && (rbi & kHalfTile)) ? 1 : 0;
    const int bi = rbi & 0xffff;
    t.arow = t.half ? bi * 64 : bi * 128;
    t.B = second ? static_cast<const float*>(g.B2) : static_cast<const float*>(g.B);
    t.ldb = second ? g.ldb2 : g.ldb;
    t.C = second ? static_cast<float*>(g.C2) : static_cast<float*>(g.C);
    t.ldc = second ? g.ldc2 : g.ldc;
    const int tri = second ? 0 : g.tri;
    const int grow0 = g.row_off + t.arow, gcol0 = g.col_off + t.bj * TN;
    t.mirror = ((tri == 2 && grow0 >= gcol0 + TN) || (tri == 3 && lm)) ? 1 : 0;
    const int K = g.ktri ? min(g.K, (t.bj + g.ktile_off + 1) * TN) : g.K;
    t.nk = K / BK;
    return t;
  };
  // list entry of ticket t (thread 0 only)
  auto lookup = [&](int t, int& rbi, int& rbj) {
    if (t >= g.ntiles) {
      rbi = -1;
      rbj = 0;
    } else if (t < g.n2) {
      rbi = g.row2 + t % g.nr2;
      rbj = (t / g.nr2) | kSecondProduct;
    } else {
      rbi = g.tile_map[2 * (t - g.n2)];
      rbj = g.tile_map[2 * (t - g.n2) + 1];
    }
  };
  // first_static: workgroup b starts on list entry b (the host orders the head of the list by workgroup: which CU
  // slot gets a half tile first), the queue hands out the entries behind the grid
  const int ticket0 = g.first_static ? int(gridDim.x) : 0;

  f32x4 ra[4], rb[4];                                            // operand stage in flight: global -> registers -> LDS
  // K slice k0 of tile t -> ra / rb
  auto load_stage = [&](const Tile& t, int k0) {
    const char* ab = reinterpret_cast<const char*>(A + (size_t)t.arow * lda + k0);
#pragma unroll
    for (int p = 0; p < 2; ++p) ra[p] = *reinterpret_cast<const f32x4*>(ab + (size_t)(32 * p) * lda * 4 + voffA);
    if (!t.half) {
#pragma unroll
      for (int p = 2; p < 4; ++p) ra[p] = *reinterpret_cast<const f32x4*>(ab + (size_t)(32 * p) * lda * 4 + voffA);
    }
    if (!BT) {
      const char* bb = reinterpret_cast<const char*>(t.B + (size_t)(t.bj * TN) * t.ldb + k0);
      const unsigned voffB = unsigned(row0 * t.ldb + q0 * 4) * 4u;
#pragma unroll
      for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const f32x4*>(bb + (size_t)(32 * p) * t.ldb * 4 + voffB);
    } else {
      const char* bb = reinterpret_cast<const char*>(t.B + (size_t)k0 * t.ldb + t.bj * TN);
      const unsigned voffB = unsigned(4 * qk * t.ldb + 4 * cq) * 4u;
#pragma unroll
      for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const f32x4*>(bb + (size_t)p * t.ldb * 4 + voffB);
    }
  };

  // ---- first tile --------------------------------------------------------------------------------------------
  {
    if (tid == 0) {
      const int t = g.first_static ? int(blockIdx.x) : atomicAdd(g.counter, 1);
      int rbi, rbj;
      lookup(t, rbi, rbj);
      s_next[0] = rbi;
      s_next[1] = rbj;
    }
    __syncthreads();
  }
  int raw_i = __builtin_amdgcn_readfirstlane(s_next[0]), raw_j = __builtin_amdgcn_readfirstlane(s_next[1]);
  if (raw_i < 0) return;
  Tile cur = decode(raw_i, raw_j);
  load_stage(cur, 0);
  bool have = true;

  auto tile_body = [&](auto tm_tag) {
    constexpr int TMb = decltype(tm_tag)::value, MI = TMb / 64, PA = TMb / 32, NB = MI * NJ;
    const Tile t = cur;
    int aslot[PA], bslot[4];
#pragma unroll
    for (int p = 0; p < PA; ++p) aslot[p] = q0 * TMb + ((row0 + 32 * p) ^ q0);
#pragma unroll
    for (int p = 0; p < 4; ++p) bslot[p] = BT ? qk * TN + ((4 * cq + p) ^ qk) : q0 * TN + ((row0 + 32 * p) ^ q0);
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto store_tile = [&](int stage) {
      f32x4* As = lds + stage * STAGE;
      f32x4* Bs = As + NQ * TMb;
#pragma unroll
      for (int p = 0; p < PA; ++p) As[aslot[p]] = ra[p];
      if (!BT) {
#pragma unroll
        for (int p = 0; p < 4; ++p) Bs[bslot[p]] = rb[p];
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {              // column 4 cq + p gets (k0 .. k3) of that column
          f32x4 tq = {rb[0][p], rb[1][p], rb[2][p], rb[3][p]};
          Bs[bslot[p]] = tq;
        }
      }
    };
    f32x4 fa[2][MI], fb[2][NJ];
    auto read_frag = [&](int stage_, int s, int buf) {
      const f32x4* As = lds + stage_ * STAGE;
      const f32x4* Bs = As + NQ * TMb;
      const int q = 2 * s + h;
#pragma unroll
      for (int u = 0; u < MI; ++u) {
        const int ar = wr * (TMb / 2) + u * 32 + l31;
        fa[buf][u] = As[q * TMb + (ar ^ q)];
      }
#pragma unroll
      for (int u = 0; u < NJ; ++u) {
        const int br = wc * (TN / 2) + u * 32 + l31;
        fb[buf][u] = Bs[q * TN + (br ^ q)];
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
    int stage = 0;
    // one K step (k_gemm_mfma's): `more`: another step of THIS tile follows (stage its slice from the registers),
    // `more2`: one more after that (request it); `nextp`: the last step, the next tile's first slice is requested
    Tile nxt = t;
    auto kstep = [&](auto more_t, auto more2_t, auto nextp_t, int k0) {
      constexpr bool more = decltype(more_t)::value, more2 = decltype(more2_t)::value, nextp = decltype(nextp_t)::value;
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        if (s + 1 < NG) {
          read_frag(stage, s + 1, (s + 1) & 1);
        } else {
          __syncthreads();                       // stage ^ 1 is complete, everybody has read this stage
          if (more) read_frag(stage ^ 1, 0, 0);
        }
        mfma_group(s & 1);
        if (s == 0 && more) {
          store_tile(stage ^ 1);
          if (more2) load_stage(t, k0 + 2 * BK);
          if constexpr (more && more2 && !BT) {
            constexpr int NWRITE = PA + 4, NMFMA = 4 * MI * NJ, R = NMFMA / NWRITE > 0 ? NMFMA / NWRITE : 1;
#pragma unroll
            for (int i = 0; i < NWRITE; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, R, 0);    // R MFMAs
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // one ds_write
            }
          }
        }
        if (s == 0 && nextp) load_stage(nxt, 0);
      }
      stage ^= 1;
    };
    using yes = std::true_type;
    using no = std::false_type;

    // ---- epilogue pieces: C enters block by block, two 32 x 32 blocks ahead of its use ----------------------
    float cb[2][16];
    const unsigned voffC = unsigned(4 * h * t.ldc + l31) * 4u;
    auto c_ptr = [&](int b) {
      const int i = b / NJ, j = b % NJ;
      const int rbase = t.arow + wr * (TMb / 2) + i * 32;
      const int c = t.bj * TN + wc * (TN / 2) + j * 32;
      return reinterpret_cast<char*>(t.C + (size_t)rbase * t.ldc + c);
    };
    auto c_issue = [&](int b, int buf) {
      if (beta != 0.f) {
        const char* cp = c_ptr(b);
#pragma unroll
        for (int e = 0; e < 16; ++e)
          cb[buf][e] = *reinterpret_cast<const float*>(cp + (size_t)((e & 3) + 8 * (e >> 2)) * t.ldc * 4 + voffC);
      }
    };
    auto c_finish = [&](int b, int buf) {
      const int i = b / NJ, j = b % NJ;
      float v[16];
      if (beta != 0.f) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = beta * cb[buf][e];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = __builtin_fmaf(alpha, acc[i][j][e], v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = alpha * acc[i][j][e];
      }
      char* cp = c_ptr(b);
#pragma unroll
      for (int e = 0; e < 16; ++e) *reinterpret_cast<float*>(cp + (size_t)((e & 3) + 8 * (e >> 2)) * t.ldc * 4 + voffC) = v[e];
      if (NTD && t.mirror) {
        // 4 consecutive registers are 4 consecutive rows -> one 16-byte store into the transposed tile
        const int rbase = t.arow + wr * (TMb / 2) + i * 32;
        const int c = t.bj * TN + wc * (TN / 2) + j * 32 + l31;
        float* Ct = t.C + (size_t)(c + g.col_off - g.row_off) * t.ldc + (g.row_off - g.col_off);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          f32x4 o = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
          *reinterpret_cast<f32x4*>(Ct + rbase + 8 * gq + 4 * h) = o;
        }
      }
    };

    // ---- the K loop ----------------------------------------------------------------------------------------------
    const int nk = t.nk;
    store_tile(0);                               // this tile's first slice was requested by the previous tile (or above)
    load_stage(t, BK);
    int ticket = 0, nbi = -1, nbj = 0;
    if (tid == 0) ticket = ticket0 + atomicAdd(g.counter, 1);
    __syncthreads();
    read_frag(0, 0, 0);
    kstep(yes{}, yes{}, no{}, 0);
    if (tid == 0) lookup(ticket, nbi, nbj);
    kstep(yes{}, yes{}, no{}, BK);
    if (tid == 0) {
      s_next[0] = nbi;
      s_next[1] = nbj;
    }
    int k0 = 2 * BK;
    for (int s = 2; s + 3 <= nk; ++s, k0 += BK) kstep(yes{}, yes{}, no{}, k0);
    c_issue(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    kstep(yes{}, no{}, no{}, k0);                // step nk - 2 (>= 2): its barrier also publishes s_next
    raw_i = __builtin_amdgcn_readfirstlane(s_next[0]);
    raw_j = __builtin_amdgcn_readfirstlane(s_next[1]);
    have = raw_i >= 0;
    if (have) nxt = decode(raw_i, raw_j);        // (no next tile: the prefetch below re-reads this tile's first slice)
    c_issue(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    kstep(no{}, no{}, yes{}, k0 + BK);           // step nk - 1
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      c_finish(b, b & 1);
      if (b + 2 < NB) c_issue(b + 2, b & 1);
    }
    cur = nxt;
  };

  while (have) {
    if (NTD && cur.half) tile_body(std::integral_constant<int, 64>{});
    else tile_body(std::integral_constant<int, 128>{});
  }
}

}  // namespace ekf
