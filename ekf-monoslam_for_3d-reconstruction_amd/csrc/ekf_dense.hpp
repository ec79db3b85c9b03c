// Dense kernels of the EKF update (a8-a10): W = Sigma H^T from the compact Jacobian,
// S = H W + R, the tall blocked Cholesky that turns [S; I] into [L; Z = L^-T], the solve
// [V; y^T] = [W; nu^T] Z, the state update mu += V y and the downdate Sigma -= V V^T.
//
// Workspaces (row-major, ldy per row):
//   Y  rows [0, m_pad)            S  (m = 2M (+3) live, identity on the padded diagonal)
//      rows [m_pad, 2 m_pad)      I -> Z = L^-T (upper triangular)
//   W  rows [0, n_pad)            Sigma H^T (pads zero);  row n_pad: nu^T = (z - h)^T
//   V  same shape as W            W L^-T;                 row n_pad: y^T = (L^-1 nu)^T
// All pads are multiples of the GEMM tile, and everything outside the live region is kept
// zero, so the tile kernels carry no edge guards.
#pragma once
#include <type_traits>
#include "ekf_math.hpp"

namespace ekf {

// ---------------------------------------------------------------------------------------
// nu = z - h for the measured list (+ plane rows: 0 - mu[{1,4,6}], vR.cpp:1257-1260).
// ---------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kQueueCounters = 256;              // work-queue heads of the queued launches of one update (8 ints apart)

// A measured list read from DEVICE memory (ekf_update_device) cannot be checked on the host: an entry outside
// [0, nfeat) or a list that is not strictly ascending raises status[1] (the next synchronising call returns
// EKF_ERR_ARG) and every kernel clamps the index it uses, so nothing is read out of bounds meanwhile.
__device__ __forceinline__ int clamp_feature(int fi, int nfeat) { return min(max(fi, 0), nfeat - 1); }
__device__ __forceinline__ void check_measured_entry(const int* __restrict__ midx, int k, int nfeat,
                                                     int* __restrict__ status) {
  const int fi = midx[k];
  if (status && (fi < 0 || fi >= nfeat || (k > 0 && midx[k - 1] >= fi))) status[1] = 1;
}

template <typename T>
__global__ void k_innovation(const T* __restrict__ z, const T* __restrict__ h, const int* __restrict__ midx,
                             int M, int plane, const T* __restrict__ mu, T* __restrict__ nu, int m_pad,
                             int* __restrict__ counters, int nfeat, int* __restrict__ status) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (counters)                                          // work-queue heads / dependency counters of this update's queued GEMMs
    for (int c = t; c < kQueueCounters; c += gridDim.x * blockDim.x) counters[c] = 0;
  if (t >= m_pad) return;
  T v = T(0);
  if (t < 2 * M) {
    if ((t & 1) == 0) check_measured_entry(midx, t >> 1, nfeat, status);
    v = z[t] - h[2 * clamp_feature(midx[t >> 1], nfeat) + (t & 1)];
  } else if (plane && t < 2 * M + 3) {
    const int e = t - 2 * M;
    v = -mu[e == 0 ? 1 : (e == 1 ? 4 : 6)];
  }
  nu[t] = v;
}

// ---------------------------------------------------------------------------------------
// W = Sigma H^T.  Lane = measurement slot k (feature midx[k]); the lane keeps that feature's
// 2x13 compact Jacobian in registers and walks RB rows of Sigma: per row it needs the 7
// camera entries (block-uniform -> scalar loads) and its own 6 (3) feature entries, which are
// contiguous across lanes.  Sigma is read exactly once; W is written as float2 per lane.
// Slots k in [M, m_pad/2) write zeros / the plane columns (Sigma[:,1], [:,4], [:,6]).
// ---------------------------------------------------------------------------------------
template <typename T, int RB>
__global__ void k_sigma_ht(const T* __restrict__ S, int ld, int n,
                           const T* __restrict__ Hc, const T* __restrict__ Hf,
                           const int* __restrict__ pos, const int* __restrict__ coding,
                           const int* __restrict__ midx, int M, int plane,
                           T* __restrict__ W, int ldy, int m_pad, int row_begin, int row_end,
                           int nfeat,
                           const T* __restrict__ z = nullptr, const T* __restrict__ h = nullptr,
                           const T* __restrict__ mu = nullptr, T* __restrict__ nu = nullptr,
                           int* __restrict__ counters = nullptr, int* __restrict__ status = nullptr,
                           T* __restrict__ q_old = nullptr, int slot0 = 0, int slot_end = 0) {
  // slots [slot0, slot_end) only (slot_end = 0: all of them): the columns of ONE column chunk of the factorisation,
  // re-evaluated from the downdated Sigma (Filter::update, EKF_W_RECOMPUTE)
  const int k = slot0 + blockIdx.x * blockDim.x + threadIdx.x;    // measurement slot
  const int nslots = slot_end > 0 ? slot_end : m_pad / 2;
  if (nu != nullptr && blockIdx.y == gridDim.y - 1) {
    if (q_old && k < 4) q_old[k] = mu[3 + k];             // the quaternion before this update (k_update_oneblock_small)
    // one more slab of workgroups than the rows need: the innovation nu = z - h (k_innovation folded into this
    // launch: one launch less in front of the chain) and the reset of the work-queue heads of this update
    if (counters)
      for (int c = k; c < kQueueCounters; c += gridDim.x * blockDim.x) counters[c] = 0;
    if (k >= nslots) return;
    if (k < M) check_measured_entry(midx, k, nfeat, status);
#pragma unroll
    for (int t = 2 * k; t < 2 * k + 2; ++t) {
      T v = T(0);
      if (t < 2 * M) {
        v = z[t] - h[2 * clamp_feature(midx[t >> 1], nfeat) + (t & 1)];
      } else if (plane && t < 2 * M + 3) {
        const int e = t - 2 * M;
        v = -mu[e == 0 ? 1 : (e == 1 ? 4 : 6)];
      }
      nu[t] = v;
    }
    return;
  }
  const int row0 = row_begin + blockIdx.y * RB;           // rows [row_begin, row_end) of Sigma
  if (k >= nslots) return;
  const int row1 = min(row0 + RB, min(row_end, n));
  if (k < M) {
    const int fi = clamp_feature(midx[k], nfeat);
    const int p = pos[fi];
    const int fs = coding[fi] ? 3 : 6;
    T hc[14], hf[12];
#pragma unroll
    for (int t = 0; t < 14; ++t) hc[t] = Hc[(size_t)fi * 14 + t];
#pragma unroll
    for (int t = 0; t < 12; ++t) hf[t] = Hf[(size_t)fi * 12 + t];
    for (int i = row0; i < row1; ++i) {
      const T* srow = S + (size_t)i * ld;
      T a0 = T(0), a1 = T(0);
#pragma unroll
      for (int t = 0; t < 7; ++t) { const T v = srow[t]; a0 += v * hc[t]; a1 += v * hc[7 + t]; }
      if (fs == 6) {
#pragma unroll
        for (int t = 0; t < 6; ++t) { const T v = srow[p + t]; a0 += v * hf[t]; a1 += v * hf[6 + t]; }
      } else {
#pragma unroll
        for (int t = 0; t < 3; ++t) { const T v = srow[p + t]; a0 += v * hf[t]; a1 += v * hf[6 + t]; }
      }
      T* w = W + (size_t)i * ldy + 2 * k;
      w[0] = a0; w[1] = a1;
    }
  } else {
    // pad slots: zeros, except the three plane columns that start at column 2M
    for (int i = row0; i < row1; ++i) {
      T* w = W + (size_t)i * ldy;
      for (int c = 2 * k; c < 2 * k + 2; ++c) {
        T v = T(0);
        if (plane && c >= 2 * M && c < 2 * M + 3) {
          const int e = c - 2 * M;
          v = S[(size_t)i * ld + (e == 0 ? 1 : (e == 1 ? 4 : 6))];
        }
        w[c] = v;
      }
    }
  }
}

// One column of the right-looking update of the innovation row, nu[c] -= sum_k y[k] L[c][k] over a chunk's K columns: a serial
// fmaf chain IN THE ORDER THE TILE GEMM ADDS THEM (k_gemm_mfma: within every group of eight k the four
// v_mfma_f32_32x32x2_f32 of a fragment take k = {0, 4}, {1, 5}, {2, 6}, {3, 7}), then C' = fma(-1, acc, C): the same bits as
// the 64 x 128 tile launch it replaced in round 5.  Used by k_innov_row_update and by the workgroups that ride in k_syrk_bf16x6.
__device__ __forceinline__ void innov_row_column(const float* __restrict__ y, const float* __restrict__ Lr, int K,
                                                 float* __restrict__ nu_c) {
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 32) {
    f32x4 b[8], a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      b[u] = *reinterpret_cast<const f32x4*>(Lr + k0 + 4 * u);
      a[u] = *reinterpret_cast<const f32x4*>(y + k0 + 4 * u);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = __builtin_fmaf(a[2 * s][e], b[2 * s][e], acc);            // k = 8 s + e
        acc = __builtin_fmaf(a[2 * s + 1][e], b[2 * s + 1][e], acc);    // k = 8 s + 4 + e
      }
  }
  *nu_c = __builtin_fmaf(-1.f, acc, 1.f * *nu_c);
}

// ---------------------------------------------------------------------------------------
// W = Sigma H^T for measured lists whose features sit next to each other in the state (the usual case: every visible
// feature measured, all of them inverse-depth), fp32.  One workgroup = 128 slots x RB rows: the RB row segments of
// Sigma the 128 features need (768 contiguous floats each, + 2 in front so that the 16-byte loads are aligned) and the 7
// camera entries of each row are staged in LDS with every load in flight before the first LDS write; lane (slot, half)
// then walks RB / 2 rows: 3 ds_read_b64 + the broadcast camera entries per row, the same sums in the same order as
// k_sigma_ht (bit-identical), one float2 store per lane and row (1 KB contiguous per row).  A workgroup whose 128 slots
// are not such a run (a measured subset, XYZ features, the padding / plane slots at the end of the list) takes
// k_sigma_ht's path for its slots.  Used for the whole W in front of the factorisation and for the re-evaluation of
// one column chunk from the downdated Sigma (slots [slot0, slot_end), EKF_OPT_W_RECOMPUTE).
// ---------------------------------------------------------------------------------------
template <int RB>
__global__ void __launch_bounds__(256, RB <= 8 ? 6 : 3)
k_sigma_ht_fast(const float* __restrict__ S, int ld, int n,
                const float* __restrict__ Hc, const float* __restrict__ Hf,
                const int* __restrict__ pos, const int* __restrict__ coding,
                const int* __restrict__ midx, int M, int plane,
                float* __restrict__ W, int ldy, int m_pad, int nfeat, int slot0, int slot_end,
                const float* __restrict__ z, const float* __restrict__ h, const float* __restrict__ mu,
                float* __restrict__ nu, int* __restrict__ counters, int* __restrict__ status, float* __restrict__ q_old,
                int row_lo = 0) {     // rows [row_lo, n) of W (a rank of a sharded filter: its camera rows, its own rows)
  constexpr int SL = 128, SEG = 6 * SL + 4;              // floats of a row segment: fp <= 3 in front, 768, the rest behind
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float seg[RB * SEG];
  __shared__ __attribute__((aligned(16))) float cam[RB * 8];
  const int tid = threadIdx.x;
  if (nu != nullptr && blockIdx.y == gridDim.y - 1) {
    // one more slab of workgroups than the rows need: nu = z - h, the work-queue heads, the quaternion before the update
    // (exactly k_sigma_ht's slab)
    const int k = blockIdx.x * blockDim.x + tid;
    const int nslots = m_pad / 2;
    if (q_old && k < 4) q_old[k] = mu[3 + k];
    if (counters)
      for (int c = k; c < kQueueCounters; c += gridDim.x * blockDim.x) counters[c] = 0;
    if (k >= nslots) return;
    if (k < M) check_measured_entry(midx, k, nfeat, status);
#pragma unroll
    for (int t = 2 * k; t < 2 * k + 2; ++t) {
      float v = 0.f;
      if (t < 2 * M) {
        v = z[t] - h[2 * clamp_feature(midx[t >> 1], nfeat) + (t & 1)];
      } else if (plane && t < 2 * M + 3) {
        const int e = t - 2 * M;
        v = -mu[e == 0 ? 1 : (e == 1 ? 4 : 6)];
      }
      nu[t] = v;
    }
    return;
  }
  const int s = tid & (SL - 1), half = tid >> 7;
  const int kbase = slot0 + blockIdx.x * SL;
  const int k = kbase + s;
  const int nslots = slot_end > 0 ? slot_end : m_pad / 2;
  const int row0 = row_lo + blockIdx.y * RB, row1 = min(row0 + RB, n);
  // is this workgroup's stretch of the list a run of inverse-depth features that are neighbours in the state?
  int fi = -1, p = 0, fs = 6;
  if (k < M && k < nslots) {
    fi = clamp_feature(midx[k], nfeat);
    p = pos[fi];
    fs = coding[fi] ? 3 : 6;
  }
  int p_first = 0;
  if (kbase < M) p_first = pos[clamp_feature(midx[kbase], nfeat)];
  // the compact Jacobian of the lane's feature: requested together with pos / coding (one level of the dependent chain
  // midx -> {pos, coding, Hc, Hf} -> row segments, not two)
  float hc[14], hf[12];
  {
    const int fj = max(fi, 0);
#pragma unroll
    for (int t = 0; t < 14; ++t) hc[t] = Hc[(size_t)fj * 14 + t];
#pragma unroll
    for (int t = 0; t < 12; ++t) hf[t] = Hf[(size_t)fj * 12 + t];
  }
  // the staged segment starts at the 16-byte boundary below the first feature: fp floats of front pad (camera_dim 14:
  // p = 14 + 6 s, fp = 2; the literal 13 + 6 N layout: fp = 1 or 3 -- round 4 only took fp = 2 and sent every odd layout
  // down the scalar path)
  const int fp = p_first & 3;
  const bool lane_ok = fi >= 0 && fs == 6 && p == p_first + 6 * s;
  const bool fast = __syncthreads_and(lane_ok ? 1 : 0) != 0;
  if (!fast) {
    // k_sigma_ht's path for (slot k, this half of the rows)
    if (k >= nslots) return;
    const int ra = row0 + half * (RB / 2), rb = min(ra + RB / 2, row1);
    if (k < M) {
      for (int i = ra; i < rb; ++i) {
        const float* srow = S + (size_t)i * ld;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int t = 0; t < 7; ++t) { const float v = srow[t]; a0 += v * hc[t]; a1 += v * hc[7 + t]; }
        if (fs == 6) {
#pragma unroll
          for (int t = 0; t < 6; ++t) { const float v = srow[p + t]; a0 += v * hf[t]; a1 += v * hf[6 + t]; }
        } else {
#pragma unroll
          for (int t = 0; t < 3; ++t) { const float v = srow[p + t]; a0 += v * hf[t]; a1 += v * hf[6 + t]; }
        }
        float* w = W + (size_t)i * ldy + 2 * k;
        w[0] = a0; w[1] = a1;
      }
    } else {
      for (int i = ra; i < rb; ++i) {
        float* w = W + (size_t)i * ldy;
        for (int c = 2 * k; c < 2 * k + 2; ++c) {
          float v = 0.f;
          if (plane && c >= 2 * M && c < 2 * M + 3) {
            const int e = c - 2 * M;
            v = S[(size_t)i * ld + (e == 0 ? 1 : (e == 1 ? 4 : 6))];
          }
          w[c] = v;
        }
      }
    }
    return;
  }
  // stage: RB x (SEG / 4 = 193) float4 of the feature segments + RB x 2 float4 of the camera columns
  constexpr int Q = SEG / 4, TOT = RB * Q, PER = (TOT + 255) / 256;
  const float* sbase = S + (p_first - fp);
  // every load unconditional (indices clamped to the last slot / the last live row: a guarded load makes the compiler
  // branch around and wait for each one), all of them in flight before the first LDS write
  f4 v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = min(tid + 256 * u, TOT - 1);
    const int r = idx / Q, q = idx - r * Q;
    v[u] = *reinterpret_cast<const f4*>(sbase + (size_t)min(row0 + r, n - 1) * ld + 4 * q);
  }
  const f4 cv = *reinterpret_cast<const f4*>(S + (size_t)min(row0 + ((tid >> 1) & (RB - 1)), n - 1) * ld + 4 * (tid & 1));
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = tid + 256 * u;
    if (idx < TOT) *reinterpret_cast<f4*>(seg + 4 * idx) = v[u];      // seg[r][4 q ..]: idx = r Q + q, SEG = 4 Q
  }
  if (tid < 2 * RB) *reinterpret_cast<f4*>(cam + 4 * tid) = cv;
  __syncthreads();
  const int ra = half * (RB / 2);
#pragma unroll
  for (int r = ra; r < ra + RB / 2; ++r) {
    const int i = row0 + r;
    if (i < n) {
      const float* sr = seg + r * SEG + fp + 6 * s;
      float fv[6];
      if ((fp & 1) == 0) {                                 // 8-byte aligned: three ds_read_b64
        const f2 x0 = *reinterpret_cast<const f2*>(sr), x1 = *reinterpret_cast<const f2*>(sr + 2), x2 = *reinterpret_cast<const f2*>(sr + 4);
        fv[0] = x0[0]; fv[1] = x0[1]; fv[2] = x1[0]; fv[3] = x1[1]; fv[4] = x2[0]; fv[5] = x2[1];
      } else {
#pragma unroll
        for (int t = 0; t < 6; ++t) fv[t] = sr[t];
      }
      const f4 c0 = *reinterpret_cast<const f4*>(cam + 8 * r), c1 = *reinterpret_cast<const f4*>(cam + 8 * r + 4);
      const float cvv[7] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2]};
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int t = 0; t < 7; ++t) { a0 += cvv[t] * hc[t]; a1 += cvv[t] * hc[7 + t]; }
#pragma unroll
      for (int t = 0; t < 6; ++t) { a0 += fv[t] * hf[t]; a1 += fv[t] * hf[6 + t]; }
      *reinterpret_cast<f2*>(W + (size_t)i * ldy + 2 * k) = f2{a0, a1};
    }
  }
}

// ---------------------------------------------------------------------------------------
// S = H W + R.  Lane = column c of S (coalesced along W rows); a block handles KB measured
// features.  The 7 camera rows of W stay in registers across the features of the block.
// Blocks past the feature blocks write the plane rows and the identity padding, 8 rows each.
// ---------------------------------------------------------------------------------------
// Column chunks of the factorisation (see Filter::update): chunk g covers columns [end[g-1], end[g]).
struct ChunkTab {
  int n;
  int end[8];
};
// Identity strip under S: chunk g keeps its own inverse Z_gg = L_gg^-T in rows [0, width_g) of the
// strip, columns of the chunk: strip[i][c] = 1 where c - (first column of c's chunk) == i.
__device__ __forceinline__ bool strip_is_one(const ChunkTab& t, int i, int c) {
  int start = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g)
    if (g < t.n && c >= t.end[g]) start = t.end[g];
  return c - start == i;
}

template <typename T, int KB>
__global__ void k_innovation_cov(const T* __restrict__ W, int ldy,
                                 const T* __restrict__ Hc, const T* __restrict__ Hf,
                                 const int* __restrict__ pos, const int* __restrict__ coding,
                                 const int* __restrict__ midx, int M, int plane, T r_pix, T r_plane,
                                 T* __restrict__ Sm, int m_pad, int k_begin, int k_end, T* __restrict__ Zid,
                                 int nfeat, ChunkTab tab = ChunkTab{0, {}}, int strip_rows = 0) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m_pad) return;
  const int m = 2 * M + (plane ? 3 : 0);
  T wc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) wc[t] = W[(size_t)t * ldy + c];
  // blockIdx.y < nfb: KB measured features each; the blocks after them: 8 rows each of the plane / padding rows
  const int nfb = max(1, (k_end - k_begin + KB - 1) / KB);
  const int k0 = k_begin + blockIdx.y * KB;               // measured features [k_begin, k_end)
  if (blockIdx.y < nfb) {
  // the KB features of the block: their (uniform) list entries first, then EVERY row of W they need requested before
  // the first sum (round 4: one feature at a time the loop was a chain of KB dependent scalar -> vector load rounds,
  // 27 us for 75 MB; the rows of a 3-entry feature are requested at its last valid row and dropped by a select, so
  // that no load sits behind a branch)
  int pk[KB], fsk[KB], fik[KB];
#pragma unroll
  for (int u = 0; u < KB; ++u) {
    const int k = min(k0 + u, k_end - 1);
    fik[u] = clamp_feature(midx[max(k, 0)], nfeat);
    pk[u] = pos[fik[u]];
    fsk[u] = coding[fik[u]] ? 3 : 6;
  }
  T wv[KB][6];
#pragma unroll
  for (int u = 0; u < KB; ++u)
#pragma unroll
    for (int t = 0; t < 6; ++t) wv[u][t] = W[(size_t)(pk[u] + min(t, fsk[u] - 1)) * ldy + c];
#pragma unroll
  for (int u = 0; u < KB; ++u) {
    const int k = k0 + u;
    if (k >= k_end) break;
    const int fi = fik[u];
    const int fs = fsk[u];
    const T* hc = Hc + (size_t)fi * 14;
    const T* hf = Hf + (size_t)fi * 12;
    T a0 = T(0), a1 = T(0);
#pragma unroll
    for (int t = 0; t < 7; ++t) { a0 += hc[t] * wc[t]; a1 += hc[7 + t] * wc[t]; }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      if (t < fs) {                       // (uniform: fs is a per-feature scalar)
        const T v = wv[u][t];
        a0 += hf[t] * v; a1 += hf[6 + t] * v;
      }
    }
    if (c == 2 * k) a0 += r_pix;
    if (c == 2 * k + 1) a1 += r_pix;
    if (c >= m) { a0 = T(0); a1 = T(0); }
    Sm[(size_t)(2 * k) * ldy + c] = a0;
    Sm[(size_t)(2 * k + 1) * ldy + c] = a1;
    if (Zid && 2 * k < strip_rows) {    // identity strip under S (rows m_pad + i), same sweep
      Zid[(size_t)(2 * k) * ldy + c] = strip_is_one(tab, 2 * k, c) ? T(1) : T(0);
      Zid[(size_t)(2 * k + 1) * ldy + c] = strip_is_one(tab, 2 * k + 1, c) ? T(1) : T(0);
    }
  }
  }
  if (blockIdx.y >= nfb) {
    const int rb = 2 * M + (blockIdx.y - nfb) * 8;
    for (int r = rb; r < min(rb + 8, m_pad); ++r) {
      T v = T(0);
      if (r < m) {                      // plane rows: H = e1, e4, e6  -> rows 1, 4, 6 of W
        const int e = r - 2 * M;
        if (c < m) v = W[(size_t)(e == 0 ? 1 : (e == 1 ? 4 : 6)) * ldy + c];
        if (c == r) v += r_plane;
      } else if (c == r) {
        v = T(1);                       // identity padding keeps the padded factorisation regular
      }
      Sm[(size_t)r * ldy + c] = v;
      if (Zid && r < strip_rows) Zid[(size_t)r * ldy + c] = strip_is_one(tab, r, c) ? T(1) : T(0);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Tile GEMMs  C = beta C + alpha A op(B)   (A: rows x K row-major, K-contiguous)
//   BT = false ("NT"): B is cols x K row-major, C[i][j] = sum_k A[i][k] B[j][k]
//   BT = true  ("NN"): B is K x cols row-major, C[i][j] = sum_k A[i][k] B[k][j]
//   tri: 0 = every tile; 1 = skip tiles strictly above the diagonal, the diagonal being
//        (row_off + i == col_off + j); 2 = as 1 and mirror every strictly-lower tile into
//        C^T (symmetric rank-K update); 3 = (queued 128 x 128 downdate launches only) every listed tile is computed, the
//        ones flagged kMirrorTile are mirrored -- the row panel of a rank whose own x own block is symmetric.
//   ktri: 1 = op(B) is upper-triangular in (k, j) (zero for k > j): the K loop of column tile
//        bj stops at (bj + ktile_off + 1)*TS.
// ROLE only tags the instantiation so that rocprofv3 lists each use of the tile kernel under
// its own name (0 panel, 1 trailing, 2 downdate, 3 solve, 4 gain, 5 right-looking update of W).
//   zrow / zcol_end: tiles whose first row is >= zrow (the inverse strip under S) stop at column
//        zcol_end (the end of the current chunk); zrow = 0 disables the test.
// Dimensions are multiples of the tile (K of the K-step): no guards.
// ---------------------------------------------------------------------------------------
enum : int { ROLE_PANEL = 0, ROLE_TRAILING = 1, ROLE_DOWNDATE = 2, ROLE_SOLVE = 3, ROLE_GAIN = 4, ROLE_WUPDATE = 5 };

struct GemmArgs {
  const void* A; int lda;
  const void* B; int ldb;
  void* C; int ldc;
  int K;
  double alpha, beta;
  int tri, row_off, col_off, ktri, ktile_off;   // ktile_off: global column-tile index of bj = 0
  // Optional work queue: `ntiles` (bi, bj) pairs in tile_map, drawn through *counter by a
  // persistent grid (2 workgroups per CU).  The host orders the list heaviest-first (triangular
  // solve) or in 8x8 super-tiles (downdate: the tiles in flight share their panels in L2), so
  // the makespan does not depend on how the dispatcher places workgroups.
  const int* tile_map; int ntiles; int* counter;
  int zrow, zcol_end;
  int stagger;                                  // persistent grid: the second half of the workgroups starts `stagger` x 3.4 us late
  // Optional second product in the same queued launch (k_gemm_mfma, ROLE_DOWNDATE): C2 -= A B2^T over a plain
  // nr2 x (n2 / nr2) grid of tiles, same A, K, alpha and beta.  Its n2 tiles are drawn first (queue indices
  // 0 .. n2-1, column-major), the tile_map entries after them: the W update of a chunk and its downdate share one
  // ramp and one tail instead of two.
  const void* B2; int ldb2;
  void* C2; int ldc2;
  int n2, nr2;
  int row2;                                     // first row tile of the second product (its tiles are rows row2 .. row2 + nr2 - 1)
  // Optional (k_gemm_mfma<ROLE_SOLVE, .., 64, 128>): the tile also goes into the PLANE IMAGE of V the bf16x6 downdate reads
  // (ekf_syrk6.hpp: three bf16 planes, 12 KB records), through LDS, with k_split_image's arithmetic -- the separate image
  // launch behind every solve is gone (round 6; round 5 had measured it when the chain, not the second stream, bounded the step)
  void* img = nullptr; int img_nkc = 0, img_c0 = 0;
};
constexpr int kSecondProduct = 0x10000;         // flag on bj for a tile of the second product
constexpr int kHalfTile = 0x20000;              // flag on bi: 64-row half tile, bi & 0xffff in 64-row units (k_gemm_mfma, downdate)
constexpr int kMirrorTile = 0x40000;            // flag on bi, tri == 3 (tiles taken as listed): also store the transposed tile

// Next tile of this workgroup: plain 2-D grid (one tile, then done) or the work queue.
__device__ __forceinline__ bool gemm_next_tile(const GemmArgs& g, int* s_tile, int& iter, int& bi, int& bj) {
  if (!g.tile_map) {
    if (iter++) return false;
    bi = blockIdx.y;
    bj = blockIdx.x;
    return true;
  }
  __syncthreads();                       // everyone is done with the previous tile (and s_tile)
  if (threadIdx.x == 0) {
    *s_tile = atomicAdd(g.counter, 1);
  }
  __syncthreads();
  const int t = __builtin_amdgcn_readfirstlane(*s_tile);   // workgroup-uniform: the tile's address arithmetic stays scalar
  if (t >= g.ntiles) return false;
  if (t < g.n2) {
    bi = g.row2 + t % g.nr2;
    bj = (t / g.nr2) | kSecondProduct;
  } else {
    bi = g.tile_map[2 * (t - g.n2)];
    bj = g.tile_map[2 * (t - g.n2) + 1];
  }
  ++iter;
  return true;
}

template <typename T, int ROLE, bool BT>
__global__ void __launch_bounds__(256) k_gemm_valu(GemmArgs g) {
  constexpr int TS = 64, BK = 16;
  const T* A = static_cast<const T*>(g.A);
  const T* B = static_cast<const T*>(g.B);
  T* C = static_cast<T*>(g.C);
  const int lda = g.lda, ldb = g.ldb, ldc = g.ldc;
  const T alpha = T(g.alpha), beta = T(g.beta);
  __shared__ T As[BK][TS + 4];
  __shared__ T Bs[BK][TS + 4];
  __shared__ int s_tile;
  const int tid = threadIdx.x;
  int bi, bj, iter = 0;
  while (gemm_next_tile(g, &s_tile, iter, bi, bj)) {
  const int grow0 = g.row_off + bi * TS, gcol0 = g.col_off + bj * TS;
  if (g.tri && grow0 + TS <= gcol0) continue;
  if (g.zrow && grow0 >= g.zrow && gcol0 >= g.zcol_end) continue;
  const int K = g.ktri ? min(g.K, (bj + g.ktile_off + 1) * TS) : g.K;
  const int tx = tid & 15, ty = tid >> 4;
  const int lr = tid >> 2;            // 0..63 row within tile
  const int lk = (tid & 3) * 4;       // 0,4,8,12
  const T* Ap = A + (size_t)(bi * TS + lr) * lda + lk;
  const T* Bp = BT ? (B + (size_t)(tid >> 4) * ldb + bj * TS + (tid & 15) * 4)
                   : (B + (size_t)(bj * TS + lr) * ldb + lk);
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = T(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    T av[4], bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      av[e] = Ap[k0 + e];
      bv[e] = BT ? Bp[(size_t)k0 * ldb + e] : Bp[k0 + e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      As[lk + e][lr] = av[e];
      if (BT) Bs[tid >> 4][(tid & 15) * 4 + e] = bv[e];
      else Bs[lk + e][lr] = bv[e];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      T a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
  }
  const bool mirror = (g.tri == 2) && (grow0 >= gcol0 + TS);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = bi * TS + ty * 4 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = bj * TS + tx * 4 + j;
      T v = alpha * acc[i][j];
      if (beta != T(0)) v += beta * C[(size_t)r * ldc + c];
      C[(size_t)r * ldc + c] = v;
      if (mirror) C[(size_t)(c + g.col_off - g.row_off) * ldc + (r + g.row_off - g.col_off)] = v;
    }
  }
  }  // tile loop
}

// ---------------------------------------------------------------------------------------
// f32 MFMA tile GEMM, same contract with a 128x128x32 tile:
// 256 lanes = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x2_f32 accumulators.
// LDS image per operand: [q = k/4][row][4] 16-byte slots, slot = q*128 + (row ^ q): the
// ds_write_b128 of 8 lanes that share a row and the ds_read_b128 of 32 lanes that share q
// are both bank-conflict free.  One ds_read_b128 per operand feeds 4 MFMAs: lane half h of
// MFMA e multiplies k = 8s + 4h + e, the same permutation on A and B, so the sum is exact.
// NN mode stages B by 4x4 register transposes of row-major [k][col] quads.
// ---------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ROLE, bool BT, int TM = 128, int TN = 128, bool S2 = false>
// Register budget: two 128 x 128 workgroups per CU (two waves per SIMD each); the 64 x 64 chain tiles (ROLE_TRAILING)
// are held to 80 registers so that one of them fits on a CU BESIDE two downdate workgroups (2 x 216 + 80 <= 512, LDS
// 2 x 64 + 32 KiB): the trailing update of the chain then runs on every CU, not only on the ones the second stream leaves.
__global__ void __launch_bounds__(S2 ? 512 : 256, S2 ? 1 : ((ROLE == ROLE_TRAILING && TM == 64 && TN == 64) ? 6 : 2)) k_gemm_mfma(GemmArgs g) {
  // TM x TN output tile (64 or 128 each), 4 waves as 2 x 2, each wave (TM/2) x (TN/2) = MI x NJ
  // accumulators of 32x32.  The small shapes exist for the latency-bound launches (chain tiles, tail of
  // the triangular solve): same flop, 2-4x the workgroups.
  // (eight-wave workgroups, 2 x 4 waves of (TM/2) x (TN/4), were 3-5 % faster per launch alone and nothing in the step:
  // removed in round 3)
  constexpr int NT = 256, WC = 2, BK = 32, NQ = BK / 4, NJ = TN / (32 * WC), PB = TN * 8 / NT;
  // a queued downdate launch may carry HALF tiles (64 x 128, kHalfTile on bi, bi then in 64-row units) at the end of
  // its list: the last jobs of the slower workgroups are half as long, and the launch ends more evenly
  constexpr bool SPLIT = (ROLE == ROLE_DOWNDATE) && !BT && TM == 128 && TN == 128;
  static_assert((TM == 64 || TM == 128) && (TN == 64 || TN == 128), "tile shape");
  const float* A = static_cast<const float*>(g.A);
  const float* B0 = static_cast<const float*>(g.B);
  float* C0 = static_cast<float*>(g.C);
  const int lda = g.lda;
  constexpr bool DUAL = (ROLE == ROLE_DOWNDATE) && !BT && TM == 128 && TN == 128;   // GemmArgs::B2 / C2
  const float alpha = float(g.alpha), beta = float(g.beta);
  // S2 (the latency-bound launches of the triangular solve: under one round of tiles, each bounded by the K steps of its
  // own loop): TWO groups of four waves per workgroup, group g takes every second K step (parity g) of the tile's K range
  // with its own LDS stages; both run the same number of steps (K is a multiple of 64), so the workgroup barriers of the
  // loop stay in step; group 1 then hands its accumulators over through LDS and group 0 stores the sum.
  __shared__ f32x4 lds[(S2 ? 2 : 1) * 2 * NQ * (TM + TN)];   // per group: two stages of {A image, B image}: one barrier per K step
  // the queue's hand-over word lives in the first LDS slot (free between tiles: the body's first barrier separates its
  // last reader from the first stage store): the workgroup then takes exactly 64 KiB (32 KiB for 64 x 64), and a
  // 64 x 64 chain tile fits beside two 128 x 128 workgroups on a CU
  int* const s_tile_p = reinterpret_cast<int*>(lds);
  constexpr int STAGE = NQ * (TM + TN);
  const int tid = S2 ? (threadIdx.x & 255) : threadIdx.x;
  const int grp = S2 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;
  const int lbase = S2 ? grp * 2 * STAGE : 0;     // this group's stages
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  if (ROLE == ROLE_PANEL || ROLE == ROLE_TRAILING) __builtin_amdgcn_s_setprio(2);   // part of the serial chain
  if (g.stagger && g.tile_map && blockIdx.x >= gridDim.x / 2)
    for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  int bi, bj, iter = 0;
  while (gemm_next_tile(g, s_tile_p, iter, bi, bj)) {
  bool listed_mirror = false;
  auto tile_body = [&](auto tm_tag) {
  constexpr int TMb = decltype(tm_tag)::value, MI = TMb / 64, PA = TMb * 8 / NT;
  const bool second = DUAL && (bj & kSecondProduct);
  if (DUAL) bj &= kSecondProduct - 1;
  const float* B = second ? static_cast<const float*>(g.B2) : B0;
  float* C = second ? static_cast<float*>(g.C2) : C0;
  const int ldb = second ? g.ldb2 : g.ldb, ldc = second ? g.ldc2 : g.ldc;
  const int tri = second ? 0 : g.tri;
  const int grow0 = g.row_off + bi * TMb, gcol0 = g.col_off + bj * TN;
  if (tri && tri != 3 && grow0 + TMb <= gcol0) return;
  if (g.zrow && grow0 >= g.zrow && gcol0 >= g.zcol_end) return;
  const int Kall = g.ktri ? min(g.K, (bj + g.ktile_off + 1) * TN) : g.K;
  // S2: group g takes the K steps of parity g (k in [32 (2 s + g), 32 (2 s + g + 1))): both groups run Kall / 64 steps, and
  // which group adds a given k does not depend on the tile shape (the K range of a triangular tile does)
  const int K = S2 ? Kall / 2 : Kall;
  const int koff = S2 ? grp * BK : 0;
  // A staging: TMb*8 float4 per tile, PA per lane; 8 consecutive lanes cover 128 B of a row
  const float* Ag[PA];
  const float* Bg[4];
  int aslot[PA], bslot[4];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int idx = tid + NT * p;
    const int row = idx >> 3, q = idx & 7;
    Ag[p] = A + (size_t)(bi * TMb + row) * lda + q * 4 + koff;
    aslot[p] = q * TMb + (row ^ q);
  }
  // NN mode: lane owns k-quad qk and column quad cq: rows 4qk+p (p < 4), 4 columns; TN = 64 uses lanes < 128
  const int qk = tid / (TN / 4), cq = tid % (TN / 4);
  const bool bt_active = !BT || qk < NQ;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (!BT) {
      const int idx = tid + NT * p;
      const int row = idx >> 3, q = idx & 7;
      Bg[p] = (p < PB) ? B + (size_t)(bj * TN + row) * ldb + q * 4 + koff : B;
      bslot[p] = q * TN + (row ^ q);
    } else {
      Bg[p] = B + (size_t)(koff + 4 * (bt_active ? qk : 0) + p) * ldb + bj * TN + 4 * cq;
      bslot[p] = qk * TN + ((4 * cq + p) ^ qk);
    }
  }
  const int h = lane >> 5, l31 = lane & 31;
  // The accumulators start at zero and C enters once, in the epilogue (C' = beta C + alpha acc, one rounding at
  // the magnitude of C).  Starting them at (beta / alpha) C instead -- the C tile read up front, the epilogue a pure
  // store -- rounds every one of the K partial sums at the magnitude of C: the products below half an ulp of C are
  // dropped one by one, and the covariance downdate then loses positivity after ~1700 all-measured frames at N = 200
  // where this form holds it (tools/drift_hybrid.py, profiles/r2_drift_hybrid.txt).
  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int PBL = BT ? 4 : PB;             // B float4 loads per lane
  f32x4 ra[PA], rb[4];
  auto load_tile = [&](int k0_) {
    const int k0 = S2 ? 2 * k0_ : k0_;
#pragma unroll
    for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const f32x4*>(Ag[p] + k0);
#pragma unroll
    for (int p = 0; p < PBL; ++p)
      rb[p] = BT ? *reinterpret_cast<const f32x4*>(Bg[p] + (size_t)k0 * ldb) : *reinterpret_cast<const f32x4*>(Bg[p] + k0);
  };
  auto store_tile = [&](int stage) {
    f32x4* As = S2 ? lds + lbase + stage * STAGE : lds + stage * STAGE;
    f32x4* Bs = As + NQ * TMb;
#pragma unroll
    for (int p = 0; p < PA; ++p) As[aslot[p]] = ra[p];
    if (!BT) {
#pragma unroll
      for (int p = 0; p < PB; ++p) Bs[bslot[p]] = rb[p];
    } else if (bt_active) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {          // column 4cq+p gets (k0..k3) of that column
        f32x4 t = {rb[0][p], rb[1][p], rb[2][p], rb[3][p]};
        Bs[bslot[p]] = t;
      }
    }
  };
  // software pipeline: tile k computes from stage k&1 while tile k+1 is written to the other stage and
  // tile k+2 is in flight from global memory; one barrier per K step.  The A / B fragments are double
  // buffered too: the ds_reads of MFMA group g+1 go out before the 16 MFMAs of group g, and the first
  // group of the NEXT stage is fetched right after the barrier, under the last group of this one, so the
  // matrix pipe never waits for an LDS round trip.
  constexpr int NG = BK / 8;                   // MFMA groups per K step (each: one b128 per fragment, 4 k-pairs)
  static_assert(NG % 2 == 0, "fragment ping-pong assumes an even number of groups");
  f32x4 fa[2][MI], fb[2][NJ];
  auto read_frag = [&](int stage_, int s, int buf) {
    const f32x4* As = S2 ? lds + lbase + stage_ * STAGE : lds + stage_ * STAGE;
    const f32x4* Bs = As + NQ * TMb;
    const int q = 2 * s + h;
#pragma unroll
    for (int t = 0; t < MI; ++t) {
      const int ar = wr * (TMb / 2) + t * 32 + l31;
      fa[buf][t] = As[q * TMb + (ar ^ q)];
    }
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
      const int br = wc * (TN / WC) + t * 32 + l31;
      fb[buf][t] = Bs[q * TN + (br ^ q)];
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
  __syncthreads();                             // the previous tile of this workgroup is done with the LDS
  load_tile(0);
  store_tile(0);
  if (BK < K) load_tile(BK);
  __syncthreads();
  read_frag(0, 0, 0);
  int stage = 0;
  // one K step; `more` (another step follows: stage the tile in registers) and `more2` (one more after that: fetch it)
  // are compile-time, so the steady-state step is ONE basic block and the scheduler may place the ds_writes and the
  // global loads between the MFMAs instead of behind a branch after them
  auto kstep = [&](auto more_t, auto more2_t, int k0) {
    constexpr bool more = decltype(more_t)::value, more2 = decltype(more2_t)::value;
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      if (s + 1 < NG) {
        read_frag(stage, s + 1, (s + 1) & 1);
      } else {
        __syncthreads();                       // stage^1 is complete, everybody has read this stage
        if (more) read_frag(stage ^ 1, 0, 0);
      }
      mfma_group(s & 1);
      if (s == 0 && more) {
        store_tile(stage ^ 1);
        if (more2) load_tile(k0 + 2 * BK);
        if constexpr (more && more2 && !BT && TN == 128) {
          // steady state: one ds_write after every few MFMAs of this group instead of all of them in a row behind it
          // (the stores wait for their global loads one by one; in a row they leave the matrix pipe with one
          // instruction in flight).  128 x 128 NT tile: alone on a CU a K = 1024 tile 85.8 -> 79.1 us, the 1128-tile
          // launch 339 -> 331 us, the N = 1000 step 1.320 -> 1.305 ms
          constexpr int NWRITE = BT ? 4 : PA + PB, NMFMA = 4 * MI * NJ, R = NMFMA / NWRITE > 0 ? NMFMA / NWRITE : 1;
#pragma unroll
          for (int i = 0; i < NWRITE; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, R, 0);    // R MFMAs
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // one ds_write
          }
          // (also tried: one write per MFMA; the global loads spread the same way after the writes, one per MFMA or
          // per two: each 2 % slower on the 1128-tile launch.  Shapes: NT tiles with 128 columns -- 128 x 128 and the
          // 64 x 128 of the W update and of the half tiles -- gain; the 64 x 64 chain tiles and the NN solve tiles lose)
        }
        // (left alone, the scheduler sinks these loads to the end of the step, ~600 cycles before the stores that
        // consume them; pinning them here with sched_barrier(0), a whole step ahead, measured SLOWER: a 2-round
        // K = 1024 launch 337 against 292 us, the step 1.332 against 1.320 ms)
      }
    }
    stage ^= 1;
  };
  {
    using yes = std::true_type;
    using no = std::false_type;
    int k0 = 0;
    for (; k0 + 2 * BK < K; k0 += BK) kstep(yes{}, yes{}, k0);
    if (k0 + BK < K) { kstep(yes{}, no{}, k0); k0 += BK; }
    kstep(no{}, no{}, k0);
  }
  if constexpr (S2) {
    // group 1's accumulators -> its own (now idle) LDS stages -> added into group 0's, which stores the tile
    __syncthreads();                               // both groups are out of their loops
    f32x4* xch = lds + 2 * STAGE;
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            f32x4 o = {acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
            xch[((i * NJ + j) * 4 + qd) * NT + tid] = o;
          }
    }
    __syncthreads();
    if (grp == 1) {
      // (the first group meets once more when it also writes the plane image of the tile: every wave of the workgroup has to
      // pass the same barriers before the next tile is drawn)
      if ((ROLE == ROLE_SOLVE) && TMb == 64 && TN == 128 && g.img != nullptr) __syncthreads();
      return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x4 o = xch[((i * NJ + j) * 4 + qd) * NT + tid];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * qd + e] += o[e];
        }
  }
  // epilogue: acc reg e of lane -> row (e&3) + 8*(e>>2) + 4*h, col l31 of the 32x32 tile
  constexpr bool IMG = (ROLE == ROLE_SOLVE) && TMb == 64 && TN == 128;      // may also write the plane image of its tile
  constexpr int SP = 132;                                                   // pitch of the staged tile (floats)
  float* const stg = reinterpret_cast<float*>(lds);                         // 64 x 132 x 4 B = 33 KB of the (now idle) first group's stages
  const bool to_img = IMG && g.img != nullptr;
  if (to_img && !S2) __syncthreads();                                       // everybody is out of the K loop's stages (S2: the exchange's barriers)
  const bool mirror = (TMb == TN || SPLIT) && ((tri == 2 && grow0 >= gcol0 + TN) || (tri == 3 && listed_mirror));
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int rbase = bi * TMb + wr * (TMb / 2) + i * 32;
      const int c = bj * TN + wc * (TN / WC) + j * 32 + l31;
      float v[16];
      if (beta != 0.f) {                       // the 16 C loads of a 32 x 32 block go out together, ahead of their use
        const float* Cp = C + (size_t)(rbase + 4 * h) * ldc + c;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = beta * Cp[(size_t)((e & 3) + 8 * (e >> 2)) * ldc];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = __builtin_fmaf(alpha, acc[i][j][e], v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = alpha * acc[i][j][e];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) C[(size_t)(rbase + (e & 3) + 8 * (e >> 2) + 4 * h) * ldc + c] = v[e];
      if constexpr (IMG) {
        if (to_img) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            stg[(wr * (TMb / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * SP + wc * (TN / WC) + j * 32 + l31] = v[e];
        }
      }
      if (mirror) {
        // 4 consecutive regs are 4 consecutive rows -> one 16-byte store into the transposed tile
        float* Ct = C + (size_t)(c + g.col_off - g.row_off) * ldc + (g.row_off - g.col_off);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          f32x4 o = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
          *reinterpret_cast<f32x4*>(Ct + rbase + 8 * gq + 4 * h) = o;
        }
      }
    }
  if constexpr (IMG) {
    if (to_img) {
      // item = (row of the tile, octet of columns); consecutive lanes = consecutive rows: 1 KiB contiguous per plane
      __syncthreads();                                   // (S2: the second group has left; the barrier counts live waves)
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4* const img = static_cast<u32x4*>(g.img);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = tid & 63, o = (tid >> 6) + 4 * it;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(stg + r * SP + 8 * o), x1 = *reinterpret_cast<const f32x4*>(stg + r * SP + 8 * o + 4);
        const float a[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
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
          u32x4 v = {(unsigned)p[0] | ((unsigned)p[1] << 16), (unsigned)p[2] | ((unsigned)p[3] << 16),
                     (unsigned)p[4] | ((unsigned)p[5] << 16), (unsigned)p[6] | ((unsigned)p[7] << 16)};
          return v;
        };
        const int grow = g.row_off + bi * TMb + r, gc = g.img_c0 + bj * TN + 8 * o;
        u32x4* rec = img + ((size_t)(grow >> 7) * g.img_nkc + (gc >> 4)) * 768 + ((gc >> 3) & 1) * 128 + (grow & 127);
        rec[0] = pack(p0);
        rec[256] = pack(p1);
        rec[512] = pack(p2);
      }
    }
  }
  };
  if constexpr (SPLIT) {
    listed_mirror = (bi & kMirrorTile) != 0;
    bi &= ~kMirrorTile;
    if (bi & kHalfTile) {
      bi &= ~kHalfTile;
      tile_body(std::integral_constant<int, 64>{});
    } else {
      tile_body(std::integral_constant<int, TM>{});
    }
  } else {
    tile_body(std::integral_constant<int, TM>{});
  }
  }  // tile loop
}

// ---------------------------------------------------------------------------------------
// fp64 MFMA tile GEMM (v_mfma_f64_16x16x4_f64), same contract as k_gemm_valu<double> with its 64 x 64 tile:
// 256 lanes = 4 waves (2 x 2), each wave 32 x 32 = 2 x 2 accumulators of 16 x 16 (4 doubles per lane:
// register e holds row 4 e + lane / 16, column lane % 16; tools/f64_mfma_probe.hip).  BK = 16.  LDS image per operand: [q = k / 2][row] slots of two
// doubles (16 bytes), slot = q * 64 + (row ^ q): one ds_read_b128 feeds two MFMAs -- lane group kq = lane / 16
// of read u holds k = 8 u + 2 kq + {0, 1}, the same k permutation on A and B, so every product is there once.
// NN mode stages B by 2 x 2 register transposes of row-major [k][col] pairs.  Two LDS stages, one barrier per
// K step, as in the fp32 kernel.
// ---------------------------------------------------------------------------------------
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int ROLE, bool BT>
__global__ void __launch_bounds__(256) k_gemm_mfma_f64(GemmArgs g) {
  constexpr int TS = 64, BK = 16, NQ = BK / 2;
  const double* A = static_cast<const double*>(g.A);
  const double* B = static_cast<const double*>(g.B);
  double* C = static_cast<double*>(g.C);
  const int lda = g.lda, ldb = g.ldb, ldc = g.ldc;
  const double alpha = g.alpha, beta = g.beta;
  __shared__ f64x2 lds[2 * NQ * 2 * TS];         // two stages of {A image, B image}: 32 KiB
  __shared__ int s_tile;
  constexpr int STAGE = NQ * 2 * TS;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int kq = lane >> 4, l15 = lane & 15;
  if (ROLE == ROLE_PANEL || ROLE == ROLE_TRAILING) __builtin_amdgcn_s_setprio(2);
  int bi, bj, iter = 0;
  while (gemm_next_tile(g, &s_tile, iter, bi, bj)) {
  const int grow0 = g.row_off + bi * TS, gcol0 = g.col_off + bj * TS;
  if (g.tri && grow0 + TS <= gcol0) continue;
  if (g.zrow && grow0 >= g.zrow && gcol0 >= g.zcol_end) continue;
  const int K = g.ktri ? min(g.K, (bj + g.ktile_off + 1) * TS) : g.K;
  // staging: 64 rows x 8 slots per operand = 512 slots, 2 per lane (NT); NN: one (k pair, column pair) per lane
  const double* Ag[2];
  const double* Bg[2];
  int aslot[2], bslot[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int idx = tid + 256 * p;
    const int row = idx >> 3, q = idx & 7;
    Ag[p] = A + (size_t)(bi * TS + row) * lda + 2 * q;
    aslot[p] = q * TS + (row ^ q);
    if (!BT) {
      Bg[p] = B + (size_t)(bj * TS + row) * ldb + 2 * q;
      bslot[p] = aslot[p];
    }
  }
  if (BT) {
    const int q = tid >> 5, cp = tid & 31;       // k pair q (8), column pair cp (32)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      Bg[p] = B + (size_t)(2 * q + p) * ldb + bj * TS + 2 * cp;     // row k = 2q + p, columns 2cp, 2cp + 1
      bslot[p] = q * TS + ((2 * cp + p) ^ q);                        // slot of column 2cp + p
    }
  }
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)                  // C enters in the epilogue: one rounding at its magnitude (see k_gemm_mfma)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
  f64x2 ra[2], rb[2];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      ra[p] = *reinterpret_cast<const f64x2*>(Ag[p] + k0);
      rb[p] = BT ? *reinterpret_cast<const f64x2*>(Bg[p] + (size_t)k0 * ldb) : *reinterpret_cast<const f64x2*>(Bg[p] + k0);
    }
  };
  auto store_tile = [&](int stage) {
    f64x2* As = lds + stage * STAGE;
    f64x2* Bs = As + NQ * TS;
#pragma unroll
    for (int p = 0; p < 2; ++p) As[aslot[p]] = ra[p];
    if (!BT) {
#pragma unroll
      for (int p = 0; p < 2; ++p) Bs[bslot[p]] = rb[p];
    } else {
      // rb[0] = B[2q][c, c+1], rb[1] = B[2q+1][c, c+1]  ->  column c: (B[2q][c], B[2q+1][c]), column c+1 likewise
      f64x2 t0 = {rb[0][0], rb[1][0]}, t1 = {rb[0][1], rb[1][1]};
      Bs[bslot[0]] = t0;
      Bs[bslot[1]] = t1;
    }
  };
  __syncthreads();
  load_tile(0);
  store_tile(0);
  if (BK < K) load_tile(BK);
  __syncthreads();
  int stage = 0;
  for (int k0 = 0; k0 < K; k0 += BK, stage ^= 1) {
    if (k0 + BK < K) {
      store_tile(stage ^ 1);
      if (k0 + 2 * BK < K) load_tile(k0 + 2 * BK);
    }
    const f64x2* As = lds + stage * STAGE;
    const f64x2* Bs = As + NQ * TS;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = 4 * u + kq;
      f64x2 fa[2], fb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ar = wr * 32 + t * 16 + l15, br = wc * 32 + t * 16 + l15;
        fa[t] = As[q * TS + (ar ^ q)];
        fb[t] = Bs[q * TS + (br ^ q)];
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  const bool mirror = (g.tri == 2) && (grow0 >= gcol0 + TS);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = bj * TS + wc * 32 + j * 16 + l15;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = bi * TS + wr * 32 + i * 16 + 4 * e + kq;
        const double x = (beta != 0.0) ? __builtin_fma(alpha, acc[i][j][e], beta * C[(size_t)r * ldc + c]) : alpha * acc[i][j][e];
        C[(size_t)r * ldc + c] = x;
        if (mirror) C[(size_t)(c + g.col_off - g.row_off) * ldc + (r + g.row_off - g.col_off)] = x;
      }
    }
  }  // tile loop
}

// Identity strip of the chunked factorisation (rows = widest chunk): see ChunkTab / strip_is_one.
template <typename T>
__global__ void k_set_identity_strip(T* __restrict__ Z, int ldz, int m_pad, ChunkTab tab) {
  const int r = blockIdx.y;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < m_pad; c += gridDim.x * blockDim.x)
    Z[(size_t)r * ldz + c] = strip_is_one(tab, r, c) ? T(1) : T(0);
}

// ---------------------------------------------------------------------------------------
// Diagonal block of the blocked Cholesky: factor the NB x NB block A = L L^T in LDS and
// produce L^-1 in the same sweep by carrying an identity block under A (the tall matrix
// [A; I] turns into [L; L^-T]).  Inner blocking 16:
//   (1) 16x16 factor in the registers of one wave, rows broadcast with v_readlane;
//   (2) the 16-wide panel by forward substitution, one logical row per lane;
//   (3) rank-16 update of the trailing columns: v_mfma_f32_16x16x4_f32 tiles for f32
//       (4 MFMAs per 16x16 tile), register-tiled VALU for f64.
// At inner step K0 the rows that can change are A rows below the 16-block and I rows
// 0..K0+15: always NB "logical" rows.  One workgroup of 512 lanes; `status[0]` is raised when
// a pivot is not positive.  Writes L (lower, zeros above) back to A and L^-1 to Dinv.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <typename T, int NB, int MASK = 7>
__global__ void __launch_bounds__(512)
k_chol_diag(T* __restrict__ Aglob, int ld, T* __restrict__ Dinv, int* __restrict__ status) {
  constexpr int LDA = NB + 1;
  constexpr int NT = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* a = reinterpret_cast<T*>(smem_raw);           // [2*NB][LDA]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  for (int idx = tid; idx < NB * NB; idx += NT) {
    const int i = idx / NB, j = idx % NB;
    a[i * LDA + j] = Aglob[(size_t)i * ld + j];
    a[(NB + i) * LDA + j] = (i == j) ? T(1) : T(0);
  }
  __syncthreads();
  for (int K0 = 0; K0 < NB; K0 += 16) {
    // (1) 16x16 diagonal factor: lane i (< 16) holds row i; column k is broadcast lane by lane
    if ((MASK & 1) && wave == 0) {
      const int i = lane & 15;
      T r[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) r[j] = a[(K0 + i) * LDA + K0 + j];
      bool bad = false;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const T pk = lane_bcast(r[k], k);
        if (!(pk > T(0))) bad = true;
        const T sq = t_sqrt(pk > T(0) ? pk : T(1));
        const T lik = (i == k) ? sq : r[k] / sq;
        r[k] = lik;
#pragma unroll
        for (int j = k + 1; j < 16; ++j) r[j] -= lik * lane_bcast(lik, j);
      }
      if (bad && lane == 0) status[0] = 1;
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[(K0 + i) * LDA + K0 + j] = (j <= i) ? r[j] : T(0);
      }
    }
    __syncthreads();
    // logical rows: top rows K0+16..NB-1, then bottom rows NB+0..NB+K0+15  (NB - 16 + 16 = NB)
    const int ntop = NB - K0 - 16;
    // (2) panel: p L16^T = y by forward substitution, one logical row per lane
    if ((MASK & 2) && tid < NB) {
      const int prow = (tid < ntop) ? (K0 + 16 + tid) : (NB + (tid - ntop));
      T y[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) y[c] = a[prow * LDA + K0 + c];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        T acc = y[c];
#pragma unroll
        for (int k = 0; k < c; ++k) acc -= y[k] * a[(K0 + c) * LDA + K0 + k];
        y[c] = acc / a[(K0 + c) * LDA + K0 + c];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) a[prow * LDA + K0 + c] = y[c];
    }
    __syncthreads();
    // (3) rank-16 update of columns K0+16.. for every logical row
    if constexpr (!(MASK & 4)) {
    } else if constexpr (sizeof(T) == 4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const int ncb = ntop / 16;                   // 16-column blocks to update
      const int ntiles = (NB / 16) * ncb;
      const int lr = lane & 15, lq = lane >> 4;
      for (int t = wave; t < ntiles; t += NT / 64) {
        const int rb = t / ncb, cb = t % ncb;
        const int lrow0 = rb * 16;                 // logical row block; entirely top or bottom
        const int prow0 = (lrow0 < ntop) ? (K0 + 16 + lrow0) : (NB + (lrow0 - ntop));
        const int c0 = K0 + 16 + cb * 16;
        f4 acc;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = a[(prow0 + 4 * lq + e) * LDA + c0 + lr];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const float av = -a[(prow0 + lr) * LDA + K0 + 4 * s4 + lq];   // A[row lr][k lq], negated
          const float bv = a[(c0 + lr) * LDA + K0 + 4 * s4 + lq];       // B[k lq][col lr] = P[c0+lr][k]
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) a[(prow0 + 4 * lq + e) * LDA + c0 + lr] = acc[e];
      }
    } else if constexpr (sizeof(T) == 8) {
      // fp64: the same tiles on v_mfma_f64_16x16x4_f64 (register e = row 4 e + lane / 16, column lane % 16)
      typedef double d4 __attribute__((ext_vector_type(4)));
      const int ncb = ntop / 16;
      const int ntiles = (NB / 16) * ncb;
      const int lr = lane & 15, lq = lane >> 4;
      for (int t = wave; t < ntiles; t += NT / 64) {
        const int rb = t / ncb, cb = t % ncb;
        const int lrow0 = rb * 16;
        const int prow0 = (lrow0 < ntop) ? (K0 + 16 + lrow0) : (NB + (lrow0 - ntop));
        const int c0 = K0 + 16 + cb * 16;
        d4 acc;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = a[(prow0 + 4 * e + lq) * LDA + c0 + lr];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const double av = -a[(prow0 + lr) * LDA + K0 + 4 * s4 + lq];
          const double bv = a[(c0 + lr) * LDA + K0 + 4 * s4 + lq];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) a[(prow0 + 4 * e + lq) * LDA + c0 + lr] = acc[e];
      }
    } else {
      constexpr int RT = NB / 32;                  // logical rows per lane
      const int ncg = ntop / 4;                    // column groups of 4
      const int l32 = tid & 31, cg = tid >> 5;
      if (cg < ncg) {
        int prow[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const int lrw = l32 + 32 * r;
          prow[r] = (lrw < ntop) ? (K0 + 16 + lrw) : (NB + (lrw - ntop));
        }
        const int c0 = K0 + 16 + cg * 4;
        T acc[RT][4];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = T(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          T pr[RT], pc[4];
#pragma unroll
          for (int r = 0; r < RT; ++r) pr[r] = a[prow[r] * LDA + K0 + k];
#pragma unroll
          for (int c = 0; c < 4; ++c) pc[c] = a[(c0 + c) * LDA + K0 + k];
#pragma unroll
          for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] += pr[r] * pc[c];
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) a[prow[r] * LDA + c0 + c] -= acc[r][c];
      }
    }
    __syncthreads();
  }
  for (int idx = tid; idx < NB * NB; idx += NT) {
    const int i = idx / NB, j = idx % NB;
    Aglob[(size_t)i * ld + j] = (j <= i) ? a[i * LDA + j] : T(0);
    // bottom block holds Z = L^-T (upper): Linv[i][j] = Z[j][i]
    Dinv[(size_t)i * NB + j] = (j <= i) ? a[(NB + j) * LDA + i] : T(0);
  }
}

// ---------------------------------------------------------------------------------------
// f32 diagonal block, packed: ONE 128x129 LDS image (66 KiB, so the kernel can share a CU with
// two tile-GEMM workgroups) holds L in the lower triangle and Z = L^-T in the strict upper
// triangle; diag(Z) = 1/diag(L) is implied.  Inner blocking 16, 8 waves:
//   (1) wave 0: 16x16 factor (v_rsq + v_readlane broadcasts) and its inverse X16, both in
//       registers; writes L16 (lower), Z16 = X16^T (strict upper), 1/diag, X16;
//   (2) 7 row blocks (rows below the block + Z rows of earlier blocks): P = Y X16^T, one
//       16x16x16 product (4 v_mfma_f32_16x16x4_f32) per wave;
//   (3) rank-16 update: lower tiles of the rows below, and Z tiles right of the block (the Z rows
//       of the CURRENT block read their operand through a mask: 0 below, 1/diag on, Z16 above
//       the diagonal).
// ---------------------------------------------------------------------------------------
// One rank-16 tile update a[prow0.., c0..] -= P[prow0..][K0..] P[c0..][K0..]^T (4 MFMA 16x16x4).
// `masked`: the operand rows are the Z rows of the CURRENT 16-block: 0 below, 1/diag on, Z16 above
// the diagonal of the block.
__device__ __forceinline__ void diag_tile_update(float* a, int LDA, int prow0, int c0, int K0, bool masked,
                                                 const float* rinv, int lr, int lq) {
  // LDA = 132 (round 3; was 129): rows are 16-byte aligned and LDA = 4 (mod 64 banks), so (a) the operand rows come as
  // ONE ds_read_b128 each -- lane (lr, lq) of MFMA step e multiplies k = 4 lq + e on both operands, the same k
  // permutation on A and B --, (b) the accumulator accesses (bank 16 lq + 4 e + lr) and the column stores of the factor
  // (bank 4 lr) are conflict-free; with 129 every access of this routine met a 4-way bank conflict, and the update of a
  // block, not its factor, was what the first blocks waited for.
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 acc;
  const f4 av4 = *reinterpret_cast<const f4*>(a + (prow0 + lr) * LDA + K0 + 4 * lq);
  const f4 bv4 = *reinterpret_cast<const f4*>(a + (c0 + lr) * LDA + K0 + 4 * lq);
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = a[(prow0 + 4 * lq + e) * LDA + c0 + lr];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = 4 * lq + e;
    float av = av4[e];
    if (masked) av = (k > lr) ? av : ((k == lr) ? rinv[lr] : 0.f);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-av, bv4[e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) a[(prow0 + 4 * lq + e) * LDA + c0 + lr] = acc[e];
}

// 16x16 factor + inverse of the diagonal block at K0 in ONE wave, on the matrix pipe: writes L16 (lower),
// Z16 = X16^T (strict upper), 1/diag and X16 = L16^-1.
// The block lives in the C layout of v_mfma_f32_16x16x4_f32 (lane (lr, lq), register e = element [4 lq + e][lr]),
// BOTH triangles, and so does X (identity at the start).  Column k (slot s = k / 4, e = k % 4): row k of the
// symmetric block sits in the 16 lanes of group lq = s, register e -- exactly where an MFMA reads k-slot s of its
// A operand (A[i = lr][s]) and of its B operand (B[s][j = lr]).  With pk = a[k][k] (one v_readlane), inv = rsq(pk):
//   v = row k * inv for lr >= k, else 0            = column k of L
//   block -= v v^T                                  one MFMA (right-looking elimination of the whole block)
//   X     -= w (row k of X),  w = (v - e_k) * inv   one MFMA (X <- L_k^-1 X, L_k = I + (v - e_k) e_k^T)
// i.e. 2 MFMAs + ~8 VALU per column instead of 15 broadcasts + 15 FMAs for L and as many again for X: the
// dependent chain readlane -> rsq -> scale -> MFMA is what a column costs.
// (A substitution panel that needs no X16 was measured slower: 1.7 us per step against 0.2 us for
// the MFMA panel + 1.15 us for the inverse.)
#ifdef EKF_DIAG_STAMPS
// tools/diag_bench.hip only: shader-clock stamps of wave 0 (the library is never built with this macro)
__device__ unsigned long long ekf_diag_stamps[64];
#define EKF_DIAG_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (K0 == 16) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (lane == 0) ekf_diag_stamps[i] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#define EKF_KSTAMP(b, i) do { __builtin_amdgcn_sched_barrier(0); if (wave == 0) { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (lane == 0) ekf_diag_stamps[8 + 4 * (b) + (i)] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define EKF_DIAG_STAMP(i) do { } while (0)
#define EKF_KSTAMP(b, i) do { } while (0)
#endif
__device__ __forceinline__ void diag_factor16(float* a, int LDA, int K0, float* x16, float* rinv, int lane,
                                              int* status, float* junk16, int Kp = -1) {
  // Round 3: a column was ~190 cycles of in-order ISSUE (24 instructions), not of latency.  Measured on the MI355X
  // (tools/chain_latency.hip): MFMA -> v_cndmask -> v_mul -> MFMA 128 cycles, MFMA -> MFMA 40, a dependent VALU op ~16;
  // the old column carried two VALU ops on the chain plus masks, selects and address arithmetic around it.  Now:
  //  * block -= (row k)^T (row k) / pivot needs mask and scale on ONE operand: A = the accumulator register itself
  //    (row k where the MFMA reads k-slot s, other rows in the other slots), B = t = that register times a per-lane
  //    scale (-1 / pivot where lane group == s and column > k, else 0: the other slots of A multiply zeros).  What
  //    the unmasked A adds lands in rows <= k of the block, which are never read again;
  //  * the inverse is carried as X~ = L~^-1 of the UNIT-lower factor L~ = L D^-1 (L~[:, k] = row k / pivot): its column
  //    update X~ -= (L~[:, k] - e_k) (row k of X~) has A = the SAME t and B = the X accumulator register unmasked
  //    (t is zero outside lane group s): one v_mul feeds both MFMAs of a column, no select for the diagonal lane;
  //    L^-1 = D^-1 X~ is formed once per block (rows scaled by 1 / l_kk);
  //  * the pivot of column k+1 is known a column ahead (a[k+1][k+1] - a[k][k+1]^2 / pivot_k, two v_readlane), so rsq
  //    and the scale vector are off the chain; L goes to LDS through per-lane-group addresses prepared once
  //    (immediate column offset; the other lane groups aim at 16 scratch words of their own, junk16), the rows < k of a
  //    column land in the strict upper triangle, which Z16 overwrites;
  //  * in program order a column is: v_mul, MFMA (block), MFMA (inverse), THEN the readlanes / fma / rsq / select for the
  //    next column -- they read the accumulator as it was before the MFMA and run in its shadow (tools/chain_latency.hip:
  //    59 cycles for MFMA -> v_mul -> MFMA, 87 with the inverse's MFMA, 145 when the pivot path is issued in front).
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lr = lane & 15, lq = lane >> 4;
  f4 acc, xac;
  EKF_DIAG_STAMP(0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = 4 * lq + e;
    acc[e] = a[(K0 + max(r, lr)) * LDA + K0 + min(r, lr)];     // the LDS image holds the lower triangle
    xac[e] = (r == lr) ? 1.f : 0.f;
  }
  if (Kp >= 0) {
    // the block's own share of the rank-16 update that follows the panel of columns Kp .. Kp + 15: block -= P P^T with
    // P = rows K0 .. of that panel, applied straight to the registers (one LDS round trip for the block AND its update
    // instead of tile update -> LDS -> reload); lane (lr, lq) of step s4 supplies P[lr][4 lq + s4] to both operands
    const f4 pv = *reinterpret_cast<const f4*>(a + (K0 + lr) * LDA + Kp + 4 * lq);   // k = 4 lq + step
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-pv[s4], pv[s4], acc, 0, 0, 0);
  }
  int key[4];                                    // lr in lane group s, -1 elsewhere: "column > k of group s" is one compare
  float* colp[4];                                // L16 column store: row lr of the image in lane group s, scratch elsewhere
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    key[s] = (lq == s) ? lr : -1;
    colp[s] = (lq == s) ? a + (K0 + lr) * LDA + K0 : junk16 + lane * 16;
  }
  // epilogue stores without exec-mask branches: lanes that hold no element aim at their scratch words
  float* lp[4];                                  // L16[i = 4 lq + e][k = lr], i >= k
  float* zp[4];                                  // Z16[lr][r = 4 lq + e] = X[r][lr], r > lr
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = 4 * lq + e;
    lp[e] = (r >= lr) ? a + (K0 + r) * LDA + K0 + lr : junk16 + lane * 16 + e;
    zp[e] = (r > lr) ? a + (K0 + lr) * LDA + K0 + r : junk16 + lane * 16 + 4 + e;
  }
  // pivot and -1 / pivot of column 0; inside the loop those of column k+1 are formed AFTER the MFMAs of column k have
  // been issued, from the accumulator as it was before them.  One wave issues an instruction every ~8 cycles whatever
  // it is (tools/chain_latency.hip: 178 cycles for a 20-instruction column with or without its readlanes / reciprocal),
  // so a column is kept to 12 instructions: the rows go to LDS RAW and are scaled by 1 / sqrt(pivot) once per block.
  bool below[16];                                // lane holds column lr > k of row k: 16 lane masks, formed once (SGPR pairs)
#pragma unroll
  for (int k = 0; k < 16; ++k) below[k] = key[k >> 2] > k;
  float pk = lane_bcast(acc[0], 0);
  float nipk = -__builtin_amdgcn_rcpf(pk);       // -1 / pivot (a pivot <= 0: caught below from the stored diagonal)
  __builtin_amdgcn_sched_barrier(0);
  EKF_DIAG_STAMP(1);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int s = k >> 2, e = k & 3;
    // Program order = issue order (one wave issues in order, ~8-10 cycles per instruction): the reciprocal of the NEXT
    // pivot is started right behind this column's elimination MFMA and everything that does not feed it (the inverse's
    // MFMA, the store) is placed between it and its consumer, so that neither the ~44 cycles of MFMA -> VALU nor the ~30
    // of the reciprocal are ever waited for: 10 instructions per column (was 24).
    const float rk = acc[e];                     // row k of the block in lane group s, other rows elsewhere
    const f4 prev = acc;
    const float t = (below[k] ? rk : 0.f) * nipk;   // -L~[lr][k] for lr > k in lane group s, 0 elsewhere
    float a10 = 0.f, a11 = 0.f;
    if (k < 15) {
      const int s1 = (k + 1) >> 2, e1 = (k + 1) & 3;
      a10 = lane_bcast(rk, 16 * s + k + 1);                        // a[k][k+1] = a[k+1][k], before this column's update
      a11 = lane_bcast(prev[e1], 16 * s1 + k + 1);                 // a[k+1][k+1], likewise
    }
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rk, t, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (k < 15) {
      pk = __builtin_fmaf(a10 * nipk, a10, a11);
      nipk = -__builtin_amdgcn_rcpf(pk);
    }
    __builtin_amdgcn_sched_barrier(0);
    xac = __builtin_amdgcn_mfma_f32_16x16x4f32(t, xac[e], xac, 0, 0, 0);
    colp[s][k] = rk;                             // raw row k = unscaled column k of L (lr < k: scratch in the upper triangle)
    __builtin_amdgcn_sched_barrier(0);
  }
  EKF_DIAG_STAMP(2);
  // once per block: the pivots are the diagonal of the raw rows; L[i][k] = raw[i][k] / sqrt(pivot_k), 1 / l_kk -> rinv,
  // X = D^-1 X~ (row r scaled by 1 / l_rr).  (LDS operations of one wave execute in order.)
  const float dk = a[(K0 + lr) * LDA + K0 + lr];
  float raw[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) raw[e] = a[(K0 + 4 * lq + e) * LDA + K0 + lr];   // rows < lr: scratch, stored to scratch again
  const float ivk = __frsqrt_rn(dk);
  if (!(dk > 0.f)) status[0] = 1;                // a pivot <= 0 or NaN: the caller fails
  rinv[lr] = ivk;                                // (the four lane groups store the same 16 values)
#pragma unroll
  for (int e = 0; e < 4; ++e) *lp[e] = raw[e] * ivk;
  const f4 irv = *reinterpret_cast<const f4*>(rinv + 4 * lq);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = 4 * lq + e;                                                         // xac[e] = X~[r][lr]
    const float x = xac[e] * irv[e];
    *zp[e] = x;
    x16[r * 20 + lr] = x;                                                             // pitch 20: rows 16-byte aligned
  }
  EKF_DIAG_STAMP(3);
}

// The body is shared by the stand-alone launch (k_chol_diag_packed, plain loads and stores) and by the persistent chain
// kernel (ekf_chain.hpp: write-through stores, L1-bypassing loads): ldA(i, j0) = A[i][j0 .. j0 + 3], stA(i, j0, v) stores
// them, stD(i, j, x) stores Linv[i][j].  1024 lanes; the LDS arrays are the caller's.
struct DiagLds {
  float* a;                                      // [128 * 132]
  float (*x16)[16 * 20];                         // [2]
  float (*rinv)[16];                             // [2]
  float* junk16;                                 // [64 * 16]
};
// Three pieces (the persistent chain kernel's critical workgroup keeps the block in LDS between them and skips the load):
//   diag_load_lds    global -> LDS image (lower triangle, zeros above)
//   diag_factor_lds  the factorisation inside the image: L in the lower triangle, Z = L^-T in the strict upper one
//   diag_store_lds   L -> global, L^-1 = Z^T -> Dinv
template <typename LdA>
__device__ __forceinline__ void diag_load_lds(LdA ldA, float* a) {
  constexpr int NB = 128, LDA = NB + 4, NT = 1024, NLD = NB * NB / 4 / NT;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  // 4096 float4 of the block, all loads in flight before the first LDS write
  f4 v[NLD];
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    const int q = tid + NT * p;                  // float4 index: row q / 32, columns 4 (q % 32) ..
    v[p] = ldA(q >> 5, 4 * (q & 31));
  }
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    const int q = tid + NT * p;
    const int i = q >> 5, j0 = 4 * (q & 31);
    f4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (j0 + e <= i) ? v[p][e] : 0.f;
    *reinterpret_cast<f4*>(a + i * LDA + j0) = w;
  }
}

template <int MASK>
__device__ __forceinline__ void diag_factor_lds(int* __restrict__ status, int nblk_real, const DiagLds& L) {
  // nblk_real: 16-column blocks that hold real rows of S; the rest of the 128 block is the identity padding of
  // the last step (L = Z = I there, decoupled from the real part) and is written back untouched.
  // Round 3: SIXTEEN waves.  Measured per 16-column block (tools/diag_bench.hip, s_memtime stamps of wave 0): the factoring
  // wave needs ~2900 cycles, the seven other waves needed 3950 / 3230 / 3090 cycles for the tiles of updates 0 / 1 / 2
  // (a 16 x 16 tile is ~30 instructions around 4 MFMAs, and one wave issues an instruction every ~8-10 cycles): the first
  // blocks waited for the UPDATE, not for the factor.  Twelve waves share the tiles now; the waves 4, 8, 12 sit on the
  // factoring wave's SIMD and take none (they would share its issue slots and its matrix pipe).
  constexpr int NB = 128, LDA = NB + 4, NBLK = NB / 16;
  typedef float f4 __attribute__((ext_vector_type(4)));
  float* const a = L.a;
  float (*const x16)[16 * 20] = L.x16;           // double-buffered: block b+1 is factored while block b's
  float (*const rinv)[16] = L.rinv;              // trailing update is still being applied
  float* const junk16 = L.junk16;                // store target of the lanes that hold no element of an L16 column
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: tile indices and row offsets stay in SGPRs
  const int lr = lane & 15, lq = lane >> 4;
  const int helper = ((wave & 3) == 0) ? -1 : (wave >> 2) * 3 + (wave & 3) - 1;   // 0 .. 11 over the waves off SIMD 0
  __builtin_amdgcn_s_setprio(3);    // serial chain: win issue arbitration against co-resident tile-GEMM waves
  __syncthreads();
  if ((MASK & 1) && wave == 0) diag_factor16(a, LDA, 0, x16[0], rinv[0], lane, status, junk16);
  __syncthreads();
  for (int b = 0; b < NBLK; ++b) {
    const int K0 = b * 16;
    const float* xb = x16[b & 1];
    const float* rb_inv = rinv[b & 1];
    const int nbelow = NBLK - 1 - b;                 // row blocks below the diagonal block
    // (2) panel: row blocks {below} + {Z rows of earlier blocks}: NBLK - 1 of them, P = Y X16^T
    if ((MASK & 2) && wave >= 1 && wave < NBLK) {
      const int w = wave - 1;
      const int prow0 = (w < nbelow) ? (K0 + 16 + w * 16) : ((w - nbelow) * 16);
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const f4 av = *reinterpret_cast<const f4*>(a + (prow0 + lr) * LDA + K0 + 4 * lq);   // k = 4 lq + step on both operands
      const f4 bv = *reinterpret_cast<const f4*>(xb + lr * 20 + 4 * lq);                  // B[k][col] = X16[col][k]
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[(prow0 + 4 * lq + e) * LDA + K0 + lr] = acc[e];
    }
    __syncthreads();
    if (nbelow == 0 || b + 1 >= nblk_real) break;
    EKF_KSTAMP(b, 0);
    EKF_KSTAMP(b, 1);
    // wave 0: the next diagonal block -- its share of update b inside the registers, then elimination + inverse;
    // the helpers: every other tile of update b meanwhile:
    //   lower tiles (rb >= cb) of the rows below, except the diagonal block's; the Z tiles of row blocks 0 .. b (the current
    //   block's through the mask), block column b+1 first (the next panel reads it)
    if (wave == 0) {
      if (MASK & 1) diag_factor16(a, LDA, K0 + 16, x16[(b + 1) & 1], rinv[(b + 1) & 1], lane, status, junk16, K0);
    } else if ((MASK & 4) && helper >= 0) {
      const int ntri = nbelow * (nbelow + 1) / 2 - 1;   // lower tiles relative to block b+1, (0, 0) is wave 0's
      const int nz = (b + 1) * nbelow;                  // Z tiles: row blocks 0 .. b, block columns b+1 ..
      for (int t = helper; t < ntri + nz; t += 12) {
        if (t < ntri) {
          int rb = 0, rem = t + 1;
          while (rem > rb) { rem -= rb + 1; ++rb; }     // t + 1 -> (rb, cb = rem), cb <= rb
          diag_tile_update(a, LDA, K0 + 16 + rb * 16, K0 + 16 + rem * 16, K0, false, rb_inv, lr, lq);
        } else {
          const int u = t - ntri;
          const int cb = u / (b + 1), tb = u % (b + 1);   // column-major: block column b+1 first
          diag_tile_update(a, LDA, tb * 16, K0 + 16 + cb * 16, K0, tb == b, rb_inv, lr, lq);
        }
      }
    }
    EKF_KSTAMP(b, 2);
    __syncthreads();
    EKF_KSTAMP(b, 3);
  }
}

template <typename StA, typename StD>
__device__ __forceinline__ void diag_store_lds(StA stA, StD stD, const float* a) {
  constexpr int NB = 128, LDA = NB + 4, NT = 1024, NLD = NB * NB / 4 / NT;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  // L: row i, columns 4 jq ..: one 16-byte LDS read, one coalesced 16-byte store.  L^-1 = Z^T: row j of Z (strict upper
  // storage), columns 4 iq .. as one 16-byte LDS read, stored as Dinv[4 iq + e][j] -- the lanes of a wave walk j, so
  // each of the four stores is a coalesced 256-byte row segment (a transposed LDS read would meet 8-way conflicts).
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    const int q = tid + NT * p;
    const int i = q >> 5, j0 = 4 * (q & 31);
    const f4 l = *reinterpret_cast<const f4*>(a + i * LDA + j0);
    f4 lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) lo[e] = (j0 + e <= i) ? l[e] : 0.f;
    stA(i, j0, lo);
  }
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    const int q = tid + NT * p;
    const int j = q & 127, i0 = 4 * (q >> 7);
    const f4 z = *reinterpret_cast<const f4*>(a + j * LDA + i0);
    const float ljj = a[j * LDA + j];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e;
      stD(i, j, (j < i) ? z[e] : ((j == i) ? 1.f / ljj : 0.f));   // Linv[i][j] = Z[j][i]
    }
  }
}

template <int MASK, typename LdA, typename StA, typename StD>
__device__ __forceinline__ void chol_diag_packed_body(LdA ldA, StA stA, StD stD, int* __restrict__ status, int nblk_real,
                                                      const DiagLds& L) {
  diag_load_lds(ldA, L.a);
  diag_factor_lds<MASK>(status, nblk_real, L);
  diag_store_lds(stA, stD, L.a);
}

template <int MASK = 7>
__global__ void __launch_bounds__(1024)
k_chol_diag_packed(float* __restrict__ Aglob, int ld, float* __restrict__ Dinv, int* __restrict__ status,
                   int nblk_real = 8) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float a[128 * 132];
  __shared__ __attribute__((aligned(16))) float x16[2][16 * 20];
  __shared__ __attribute__((aligned(16))) float rinv[2][16];
  __shared__ float junk16[64 * 16];
  const DiagLds L{a, x16, rinv, junk16};
  chol_diag_packed_body<MASK>(
      [&](int i, int j0) { return *reinterpret_cast<const f4*>(Aglob + (size_t)i * ld + j0); },
      [&](int i, int j0, const f4& v) { *reinterpret_cast<f4*>(Aglob + (size_t)i * ld + j0) = v; },
      [&](int i, int j, float x) { Dinv[(size_t)i * 128 + j] = x; }, status, nblk_real, L);
}

// ---------------------------------------------------------------------------------------
// Panel of a chain step (f32, block 128), in place: P <- P Linv^T for `rows` rows of P (row stride ldp), Linv the
// 128 x 128 lower-triangular inverse k_chol_diag_packed left in Dinv (row-major, stride 128).
// One workgroup = 64 rows, one wave = 16 rows x all 128 columns: the wave's A fragments (its own rows, all of K)
// go from global memory into registers before its first store, so the update is in place without a barrier
// between reading and writing; Linv is staged once per workgroup in LDS (row pitch 33 x 16 bytes: the 16 lanes of
// a ds_read_b128 group hit 16 different bank groups).  v_mfma_f32_16x16x4_f32; lane (lr, lq) of MFMA step (u, e)
// multiplies k = 16 u + 4 lq + e on both operands; column tile ct only needs k < 16 (ct + 1) (Linv is lower
// triangular): 144 MFMAs per wave instead of 256.  Replaces the 64 x 128 tile-GEMM launch of the general kernel
// (K = 128 is four of its K steps: prologue, epilogue and barriers dominated).
// ---------------------------------------------------------------------------------------
// (the body: workgroup `wg` = rows 64 wg .. of P; shared by the plain launch and by the launch over a LIST of 128-row blocks)
__device__ __forceinline__ void panel_direct_body(float* __restrict__ P, int ldp, const float* __restrict__ Dinv, int rows, int wg,
                                                  float* sl) {
  constexpr int NB = 128, PITCH = 132;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  __builtin_amdgcn_s_setprio(2);                 // part of the serial chain
  const int row0 = wg * 64 + wave * 16;
  const bool live = row0 < rows;
  // the wave's rows: A[row0 + lr][16 u + 4 lq ..]
  f4 fa[8];
  float* Prow = P + (size_t)(row0 + lr) * ldp;
  if (live) {
#pragma unroll
    for (int u = 0; u < 8; ++u) fa[u] = *reinterpret_cast<const f4*>(Prow + 16 * u + 4 * lq);
  }
  // Linv in two halves: rows 0..63 (their columns 0..63: 1024 float4) are staged first and feed column tiles 0..3;
  // rows 64..127 (2048 float4) travel under those MFMAs and feed column tiles 4..7
  f4 v0[4], v1[8];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int q = tid + 256 * p;                 // row q / 16, columns 4 (q % 16) ..
    v0[p] = *reinterpret_cast<const f4*>(Dinv + (size_t)(q >> 4) * NB + 4 * (q & 15));
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int q = tid + 256 * p;                 // row 64 + q / 32, columns 4 (q % 32) ..
    v1[p] = *reinterpret_cast<const f4*>(Dinv + (size_t)(64 + (q >> 5)) * NB + 4 * (q & 31));
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int q = tid + 256 * p;
    *reinterpret_cast<f4*>(sl + (q >> 4) * PITCH + 4 * (q & 15)) = v0[p];
  }
  __syncthreads();
  auto column_tiles = [&](int ct0) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const int ct = ct0 + c4;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* bl = sl + (16 * ct + lr) * PITCH + 4 * lq;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u <= ct) {
          const f4 fb = *reinterpret_cast<const f4*>(bl + 16 * u);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][e], fb[e], acc, 0, 0, 0);
        }
      }
      // acc[e] = P'[row0 + 4 lq + e][16 ct + lr]
#pragma unroll
      for (int e = 0; e < 4; ++e) P[(size_t)(row0 + 4 * lq + e) * ldp + 16 * ct + lr] = acc[e];
    }
  };
  if (live) column_tiles(0);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int q = tid + 256 * p;
    *reinterpret_cast<f4*>(sl + (64 + (q >> 5)) * PITCH + 4 * (q & 31)) = v1[p];
  }
  __syncthreads();
  if (live) column_tiles(4);
}
__global__ void __launch_bounds__(256, 2) k_panel_direct(float* __restrict__ P, int ldp, const float* __restrict__ Dinv, int rows) {
  __shared__ float sl[128 * 132];
  panel_direct_body(P, ldp, Dinv, rows, blockIdx.x, sl);
}
// The same product for a LIST of 128-row blocks of a column block (the distributed chain of the sharded step: a rank's own row
// blocks are not contiguous): workgroups 2 b, 2 b + 1 = the two halves of block blocks[b]; Pcol = row 0 of the column block.
__global__ void __launch_bounds__(256, 2) k_panel_direct_blocks(float* __restrict__ Pcol, int ldp, const float* __restrict__ Dinv,
                                                               const int* __restrict__ blocks) {
  __shared__ float sl[128 * 132];
  const int blk = blocks[blockIdx.x >> 1];
  panel_direct_body(Pcol + (size_t)blk * 128 * ldp, ldp, Dinv, 128, blockIdx.x & 1, sl);
}

// ---------------------------------------------------------------------------------------
// One-block update (f32, the innovation fits one 128-column block: 2 M + 3 <= 128, the reference's operating point).
// With a single diagonal block the chunk inverse IS the transposed Linv k_chol_diag_packed leaves in Dinv, so the
// panel step is not needed and
//     V = [W; nu^T] Linv^T   (the k_panel_direct product out of place; row `rows` of W is nu^T, of V then y^T = (Linv nu)^T)
//     mu += V y              (y = Linv nu formed by every workgroup for itself from the staged Linv: 8 k MACs)
// The pieces (one workgroup of 256 lanes; one wave = 16 rows x 128 columns, v_mfma_f32_16x16x4_f32, column tile ct
// needs k < 16 (ct + 1)):
// ---------------------------------------------------------------------------------------
namespace oneblock {
constexpr int NB = 128, PITCH = 132;
typedef float f4 __attribute__((ext_vector_type(4)));

// Linv (128 x 128, row-major) -> LDS rows of pitch 132; nu -> snu.  The caller synchronises.
__device__ __forceinline__ void stage(const float* __restrict__ Dinv, const float* __restrict__ nu, float* sl, float* snu, int tid) {
  f4 v[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int q = tid + 256 * p;                   // row q / 32, columns 4 (q % 32) ..
    v[p] = *reinterpret_cast<const f4*>(Dinv + (size_t)(q >> 5) * NB + 4 * (q & 31));
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int q = tid + 256 * p;
    *reinterpret_cast<f4*>(sl + (q >> 5) * PITCH + 4 * (q & 31)) = v[p];
  }
  if (tid < NB) snu[tid] = nu[tid];
}

// y = Linv nu: wave w takes rows 32 w .. 32 w + 31, lane = column (two sweeps), butterfly sum.  The same instructions
// in every workgroup: the same bits.
__device__ __forceinline__ void form_y(const float* sl, const float* snu, float* sy, int wave, int lane) {
  for (int i = 0; i < 32; ++i) {
    const int r = 32 * wave + i;
    float acc = sl[r * PITCH + lane] * snu[lane];
    acc = __builtin_fmaf(sl[r * PITCH + 64 + lane], snu[64 + lane], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) sy[r] = acc;
  }
}

// The wave's 16 rows [row0, row0 + 16) of V = W Linv^T; out(ct, acc): acc[e] = V[row0 + 4 lq + e][16 ct + lr].
template <typename Out>
__device__ __forceinline__ void v_rows(const float* __restrict__ W, int ldw, int row0, const float* sl, int lane, Out out) {
  const int lr = lane & 15, lq = lane >> 4;
  f4 fa[8];
  const float* Wrow = W + (size_t)(row0 + lr) * ldw;
#pragma unroll
  for (int u = 0; u < 8; ++u) fa[u] = *reinterpret_cast<const f4*>(Wrow + 16 * u + 4 * lq);
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* bl = sl + (16 * ct + lr) * PITCH + 4 * lq;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u <= ct) {
        const f4 fb = *reinterpret_cast<const f4*>(bl + 16 * u);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][e], fb[e], acc, 0, 0, 0);
      }
    }
    out(ct, acc);
  }
}

// Quaternion normalisation of q (4): q <- q / |q|, Qn = (|q|^2 I - q q^T) / |q|^3 (vR.cpp:1625-1642)
__device__ __forceinline__ void normalise(float* q, float* Qn) {
  const float nn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const float norma = t_sqrt(nn);
  const float inv3 = 1.f / (norma * norma * norma);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) Qn[i * 4 + j] = ((i == j ? norma * norma : 0.f) - q[i] * q[j]) * inv3;
  for (int i = 0; i < 4; ++i) q[i] = q[i] / norma;
}
}  // namespace oneblock

// Gain solve + state update as ONE launch (any map size; the downdate and the normalisation congruence follow as
// their own launches).  Workgroup b = rows 64 b .. 64 b + 63 of [W; nu^T] (b = rows / 64 holds the nu row and also
// writes the strip Zs = Linv^T the general path's panel launch would have left: ekf_get_gain and the 1-point RANSAC
// read it); workgroup 0 holds the quaternion rows: it normalises them and leaves Qn at scr_qn, as k_state_update does.
__global__ void __launch_bounds__(256, 2)
k_solve_state_oneblock(const float* __restrict__ W, int ldw, const float* __restrict__ Dinv, float* __restrict__ V, int ldv,
                       int rows, float* __restrict__ mu, int n, float* __restrict__ scr_qn, float* __restrict__ Zs, int ldz) {
  using namespace oneblock;
  __shared__ float sl[NB * PITCH];
  __shared__ float snu[NB];
  __shared__ float sy[NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int row0 = blockIdx.x * 64 + wave * 16;
  stage(Dinv, W + (size_t)rows * ldw, sl, snu, tid);
  __syncthreads();
  form_y(sl, snu, sy, wave, lane);
  __syncthreads();
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  v_rows(W, ldw, row0, sl, lane, [&](int ct, const f4& acc) {
    const float yc = sy[16 * ct + lr];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      V[(size_t)(row0 + 4 * lq + e) * ldv + 16 * ct + lr] = acc[e];
      part[e] = __builtin_fmaf(acc[e], yc, part[e]);
    }
  });
#pragma unroll
  for (int off = 1; off < 16; off <<= 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[e] += __shfl_xor(part[e], off, 64);
  if (lr == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = row0 + 4 * lq + e;
      if (r < n) mu[r] += part[e];
    }
  }
  if ((int)blockIdx.x == rows / 64) {
    for (int q = tid; q < NB * NB; q += 256) {
      const int k = q >> 7, c = q & 127;
      Zs[(size_t)k * ldz + c] = sl[c * PITCH + k];
    }
  }
  if (blockIdx.x == 0) {
    __threadfence_block();
    __syncthreads();                               // rows 3..6 are written (this workgroup holds rows 0..63)
    if (tid == 0) {
      float q[4] = {mu[3], mu[4], mu[5], mu[6]}, Qn[16];
      normalise(q, Qn);
      for (int i = 0; i < 4; ++i) mu[3 + i] = q[i];
      for (int i = 0; i < 16; ++i) scr_qn[i] = Qn[i];
    }
  }
}

// Small maps (every 64 x 64 lower tile of Sigma gets its own workgroup, all of them side by side): the WHOLE rest of the
// update after the diagonal factor as one launch, with no hand-over between workgroups.  Workgroup t = tile (i, j),
// j <= i, 8 waves: waves 0..3 form the row block V_i, waves 4..7 V_j (2 x 2 MFLOP per workgroup, all tiles in
// parallel; the W rows are requested before Linv is staged), then
//     Sigma_ij -= V_i V_j^T                         (v_mfma_f32_16x16x4_f32 from LDS, K = 128, two 16 x 16 blocks per wave;
//                                                    the Sigma values are requested before the products)
// The tiles (i, 0) own the rest of the state update for rows 64 i ..: they store V_i, form y = Linv nu, add V_i y to
// mu, and -- the quaternion rows / columns 3..6 live in block 0 -- apply the normalisation congruence
// Sigma <- Jn Sigma Jn^T (k_strip_congruence<4>'s formulas) to their tile before it is stored: each forms the updated
// quaternion q' = q_old + V[3:7] y with the same instructions (q_old: the quaternion before the update, left there by
// k_sigma_ht's nu slab, because tile (0, 0) overwrites mu[3:7] while others may not have started), hence the same Qn.
// Tile (0, 0) stores its lower half and mirrors it; a diagonal tile (i, i) updates its lower half and mirrors it; the
// others store the tile and its mirror: Sigma stays exactly symmetric.  One more workgroup (t = ntiles) holds the nu
// row: y^T into V, and the strip Zs = Linv^T.
__global__ void __launch_bounds__(512, 1)
k_update_oneblock_small(const float* __restrict__ W, int ldw, const float* __restrict__ Dinv, float* __restrict__ V, int ldv,
                        int rows, float* __restrict__ mu, int n, const float* __restrict__ q_old, float* __restrict__ scr_qn,
                        float* __restrict__ Zs, int ldz, float* __restrict__ S, int lds_, int ntiles) {
  using namespace oneblock;
  constexpr int TP = 65;                           // pitch of the 64 x 64 tile image
  __shared__ float sl[NB * PITCH];                 // Linv; afterwards the tile image (64 x 65)
  __shared__ float sVi[64 * PITCH];
  __shared__ float sVj[64 * PITCH];
  __shared__ float snu[NB];
  __shared__ float sy[NB];
  __shared__ float sq[4], sJ[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const bool nu_wg = (int)blockIdx.x == ntiles;
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= (int)blockIdx.x) ++i;       // tile t -> (i, j), row-major over the lower triangle
  const int j = blockIdx.x - i * (i + 1) / 2;
  const bool owner = !nu_wg && (j == 0);
  const bool second = wave >= 4;                   // waves 4..7: V_j
  // this wave's 16 rows of [W; nu^T]
  const int vrow0 = nu_wg ? rows + (wave & 3) * 16 : 64 * (second ? j : i) + (wave & 3) * 16;
  const bool v_active = nu_wg ? !second : (!second || j != i);
  f4 fa[8];
  if (v_active) {
    const float* Wrow = W + (size_t)(vrow0 + lr) * ldw;
#pragma unroll
    for (int u = 0; u < 8; ++u) fa[u] = *reinterpret_cast<const f4*>(Wrow + 16 * u + 4 * lq);
  }
  {
    f4 v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int q = tid + 512 * p;                 // row q / 32, columns 4 (q % 32) ..
      v[p] = *reinterpret_cast<const f4*>(Dinv + (size_t)(q >> 5) * NB + 4 * (q & 31));
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int q = tid + 512 * p;
      *reinterpret_cast<f4*>(sl + (q >> 5) * PITCH + 4 * (q & 31)) = v[p];
    }
    if (tid < NB) snu[tid] = W[(size_t)rows * ldw + tid];
  }
  __syncthreads();
  if (owner && tid < 2 * NB) {
    // y = Linv nu: two lanes per row (halves of k), four running sums each: the same instructions in every owner
    const int r = tid >> 1, k0 = 64 * (tid & 1);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      a0 = __builtin_fmaf(sl[r * PITCH + k0 + k], snu[k0 + k], a0);
      a1 = __builtin_fmaf(sl[r * PITCH + k0 + k + 1], snu[k0 + k + 1], a1);
      a2 = __builtin_fmaf(sl[r * PITCH + k0 + k + 2], snu[k0 + k + 2], a2);
      a3 = __builtin_fmaf(sl[r * PITCH + k0 + k + 3], snu[k0 + k + 3], a3);
    }
    float acc = (a0 + a1) + (a2 + a3);
    acc += __shfl_xor(acc, 1, 64);
    if ((tid & 1) == 0) sy[r] = acc;
  }
  if (owner) __syncthreads();
  // V rows of this wave: acc[e] = V[vrow0 + 4 lq + e][16 ct + lr]
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  if (v_active) {
    float* sv = (second ? sVj : sVi) + ((wave & 3) * 16) * PITCH;
    const bool to_global = nu_wg || (owner && !second);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* bl = sl + (16 * ct + lr) * PITCH + 4 * lq;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u <= ct) {
          const f4 fb = *reinterpret_cast<const f4*>(bl + 16 * u);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][e], fb[e], acc, 0, 0, 0);
        }
      }
      const float yc = owner ? sy[16 * ct + lr] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sv[(4 * lq + e) * PITCH + 16 * ct + lr] = acc[e];
        if (to_global) V[(size_t)(vrow0 + 4 * lq + e) * ldv + 16 * ct + lr] = acc[e];
        part[e] = __builtin_fmaf(acc[e], yc, part[e]);
      }
    }
  }
  if (nu_wg) {                                     // the strip the panel launch would have left
    for (int q = tid; q < NB * NB; q += 512) {
      const int k = q >> 7, c = q & 127;
      Zs[(size_t)k * ldz + c] = sl[c * PITCH + k];
    }
    return;
  }
  if (owner && !second) {                          // mu += V_i y (the quaternion rows: below, from q')
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) part[e] += __shfl_xor(part[e], off, 64);
    if (lr == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = vrow0 + 4 * lq + e;
        if (r < n && !(r >= 3 && r < 7)) mu[r] += part[e];
      }
    }
  }
  // the Sigma values of this wave's two 16 x 16 blocks: requested now, used after the products
  const int rb = wave >> 1, cb0 = 2 * (wave & 1);
  float cin[2][4];
#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      cin[b2][e] = S[(size_t)(64 * i + 16 * rb + 4 * lq + e) * lds_ + 64 * j + 16 * (cb0 + b2) + lr];
  __syncthreads();                                 // V_i, V_j complete; nobody reads Linv any more
  const float* vj = (j != i) ? sVj : sVi;
  if (owner && wave == 0) {
    // q' = q_old + V[3:7] y (rows 3..6 of block 0), lane = column (two sweeps), butterfly sum
    for (int a = 0; a < 4; ++a) {
      float acc = vj[(3 + a) * PITCH + lane] * sy[lane];
      acc = __builtin_fmaf(vj[(3 + a) * PITCH + 64 + lane], sy[64 + lane], acc);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) sq[a] = q_old[a] + acc;
    }
    if (lane == 0) {
      float q[4] = {sq[0], sq[1], sq[2], sq[3]}, Qn[16];
      normalise(q, Qn);
      for (int k = 0; k < 16; ++k) sJ[k] = Qn[k];
      if (i == 0) {
        for (int k = 0; k < 4; ++k) mu[3 + k] = q[k];
        for (int k = 0; k < 16; ++k) scr_qn[k] = Qn[k];
      }
    }
  }
  f4 acc2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  {
    const float* ap = sVi + (16 * rb + lr) * PITCH + 4 * lq;
    const float* bp0 = vj + (16 * cb0 + lr) * PITCH + 4 * lq;
    const float* bp1 = bp0 + 16 * PITCH;
#pragma unroll
    for (int u = 0; u < 8; ++u) {                  // lane (lr, lq) of MFMA (u, e) multiplies k = 16 u + 4 lq + e on both operands
      const f4 fav = *reinterpret_cast<const f4*>(ap + 16 * u);
      const f4 fb0 = *reinterpret_cast<const f4*>(bp0 + 16 * u);
      const f4 fb1 = *reinterpret_cast<const f4*>(bp1 + 16 * u);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fav[e], fb0[e], acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fav[e], fb1[e], acc2[1], 0, 0, 0);
      }
    }
  }
  // acc2[b][e] = (V_i V_j^T)[16 rb + 4 lq + e][16 (cb0 + b) + lr]
  if (!owner) {
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const size_t gr = 64 * i + 16 * rb + 4 * lq + e, gc = 64 * j + 16 * (cb0 + b2) + lr;
        if (gr >= gc) {                            // a diagonal tile: its lower half, mirrored (the upper lanes must not touch it)
          const float v = cin[b2][e] - acc2[b2][e];
          S[gr * lds_ + gc] = v;
          S[gc * lds_ + gr] = v;
        }
      }
    return;
  }
  // tiles (i, 0): the downdated tile through LDS, the normalisation congruence on columns 3..6 (and, tile (0, 0),
  // rows 3..6 and the corner), then the tile and its mirror
  float* sT = sl;
#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
    for (int e = 0; e < 4; ++e) sT[(16 * rb + 4 * lq + e) * TP + 16 * (cb0 + b2) + lr] = cin[b2][e] - acc2[b2][e];
  __syncthreads();                                 // (also: sJ is there)
  if (i == 0) {
    // the lower half is what counts: complete the image symmetrically first, so that rows and columns read the same values
    for (int q = tid; q < 64 * 64; q += 512) {
      const int r = q >> 6, c = q & 63;
      if (c > r) sT[r * TP + c] = sT[c * TP + r];
    }
    __syncthreads();
  }
  if (tid < 64) {                                  // column strip: row r, columns 3..6 (rows outside the block)
    const int r = tid;
    if (i > 0 || r < 3 || r >= 7) {
      float x[4], yv[4];
      for (int k = 0; k < 4; ++k) x[k] = sT[r * TP + 3 + k];
      for (int c = 0; c < 4; ++c) {
        float a = 0.f;
        for (int k = 0; k < 4; ++k) a = __builtin_fmaf(x[k], sJ[c * 4 + k], a);
        yv[c] = a;
      }
      for (int c = 0; c < 4; ++c) sT[r * TP + 3 + c] = yv[c];
    }
  } else if (i == 0 && tid < 128) {                // row strip of tile (0, 0): column c, rows 3..6
    const int c = tid - 64;
    if (c < 3 || c >= 7) {
      float x[4], yv[4];
      for (int k = 0; k < 4; ++k) x[k] = sT[(3 + k) * TP + c];
      for (int r = 0; r < 4; ++r) {
        float a = 0.f;
        for (int k = 0; k < 4; ++k) a = __builtin_fmaf(sJ[r * 4 + k], x[k], a);
        yv[r] = a;
      }
      for (int r = 0; r < 4; ++r) sT[(3 + r) * TP + c] = yv[r];
    }
  } else if (i == 0 && tid == 128) {               // corner: lower half of J C J^T, mirrored
    float C[16], A[16];
    for (int k = 0; k < 16; ++k) C[k] = sT[(3 + k / 4) * TP + 3 + k % 4];
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        float a = 0.f;
        for (int k = 0; k < 4; ++k) a += sJ[r * 4 + k] * C[k * 4 + c];
        A[r * 4 + c] = a;
      }
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c <= r; ++c) {
        float a = 0.f;
        for (int k = 0; k < 4; ++k) a += A[r * 4 + k] * sJ[c * 4 + k];
        sT[(3 + r) * TP + 3 + c] = a;
        sT[(3 + c) * TP + 3 + r] = a;
      }
  }
  __syncthreads();
  for (int q = tid; q < 64 * 64; q += 512) {
    const int r = q >> 6, c = q & 63;
    if (i == 0) {
      S[(size_t)r * lds_ + c] = (r >= c) ? sT[r * TP + c] : sT[c * TP + r];
    } else {
      S[(size_t)(64 * i + r) * lds_ + c] = sT[r * TP + c];
      S[(size_t)r * lds_ + 64 * i + c] = sT[c * TP + r];         // the mirror tile (0, i): row r, column 64 i + c
    }
  }
}

// ---------------------------------------------------------------------------------------
// Right-looking update of the innovation row alone, nu^T[c1:] -= y_g^T L[c1:, g]^T (the sequential form of the chunked
// update keeps W by re-evaluation; of [W; nu^T] only this row is still updated by the factor): one lane per column
// (innov_row_column above) -- the same bits as the 64 x 128 tile launch it replaced (26 tiles, one K loop each: 20-29 us of
// latency at N = 1000, against ~5 us here).  L rows are walked by 16-byte loads that stay in L1 (64 rows x 128 B per
// workgroup).  The stand-alone launch: chunks whose downdate does not run in k_syrk_bf16x6 (whose launch carries it).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_innov_row_update(const float* __restrict__ y, const float* __restrict__ L, int ldl,
                                                        float* __restrict__ nu, int cols, int K) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= cols) return;
  innov_row_column(y, L + (size_t)c * ldl, K, nu + c);
}

// ---------------------------------------------------------------------------------------
// mu[i] += sum_c V[i][c] y[c]   (K nu = V (L^-1 nu)): a GEMV that streams V once (n x m_pad, 49 MB at N = M = 1000).
// One wave per row, 16-byte loads, the loads of up to eight 64-lane sweeps of the row in flight before the first
// multiply (a 2048-column row is 8 KiB = 8 loads per lane, all outstanding at once); y comes through L1 / L2 (every
// wave reads the same 8 KiB).  m_pad is a multiple of 64, so a sweep never runs past the padded row.
// ---------------------------------------------------------------------------------------
// One row of mu += V y by one wave (lane = 16 bytes of the row per sweep), and the quaternion normalisation behind it:
// shared by k_state_update and by the workgroups of the last k_syrk_bf16x6 launch that run out of tiles (ekf_syrk6.hpp).
template <typename T>
__device__ __forceinline__ void state_update_row(T* __restrict__ mu, const T* __restrict__ V, int ldy, int row,
                                                 const T* __restrict__ y, int m_pad, int lane) {
  constexpr int VEC = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const T* v = V + (size_t)row * ldy;
  T acc = T(0);
  constexpr int U = 8, SWEEP = 64 * VEC;
  for (int c0 = lane * VEC; c0 < m_pad; c0 += U * SWEEP) {
    vec_t a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * SWEEP;
      if (c < m_pad) {
        a[u] = *reinterpret_cast<const vec_t*>(v + c);
        b[u] = *reinterpret_cast<const vec_t*>(y + c);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u * SWEEP < m_pad) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += a[u][e] * b[u][e];
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) mu[row] += acc;
}
template <typename T>
__device__ __forceinline__ void state_update_normalise(T* __restrict__ mu, T* __restrict__ scr_qn) {
  const T q[4] = {mu[3], mu[4], mu[5], mu[6]};
  const T nn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const T norma = t_sqrt(nn);
  const T inv3 = T(1) / (norma * norma * norma);
  for (int i = 0; i < 4; ++i) mu[3 + i] = q[i] / norma;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      scr_qn[i * 4 + j] = ((i == j ? norma * norma : T(0)) - q[i] * q[j]) * inv3;
}

template <typename T>
__global__ void __launch_bounds__(512)
k_state_update(T* __restrict__ mu, const T* __restrict__ V, int ldy, int n,
               const T* __restrict__ y, int m_pad, T* __restrict__ scr_qn = nullptr) {
  // scr_qn != nullptr (launched with 512 lanes: rows 0..7 are workgroup 0): the workgroup that owns the quaternion
  // rows also normalises it and leaves Qn = (|q|^2 I - q q^T) / |q|^3 (4 x 4) at scr_qn (k_normalize_quat folded in).
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row < n) state_update_row(mu, V, ldy, row, y, m_pad, lane);
  if (scr_qn != nullptr && blockIdx.x == 0) {
    __threadfence_block();
    __syncthreads();                             // rows 3..6 are written (workgroup 0 holds rows 0..7)
    if (threadIdx.x == 0) state_update_normalise(mu, scr_qn);
  }
}

template <typename T>
__global__ void k_copy2d(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, int rows, int cols) {
  const int r = blockIdx.y;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x)
    if (r < rows) dst[(size_t)r * ldd + c] = src[(size_t)r * lds_ + c];
}

// ---------------------------------------------------------------------------------------
// fp64 diagonal block, packed, NB = 64: the structure of k_chol_diag_packed (one LDS image with L in the lower and
// Z = L^-T in the strict upper triangle, 16 x 16 factor + inverse in the registers of wave 0, panel and rank-16
// updates on the matrix pipe, look-ahead factor) on v_mfma_f64_16x16x4_f64.  Register e of an accumulator holds
// row 4 e + lane / 16, column lane % 16.
// ---------------------------------------------------------------------------------------
typedef double f64x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void diag_tile_update_f64(double* a, int LDA, int prow0, int c0, int K0, bool masked,
                                                     const double* rinv, int lr, int lq) {
  f64x4_t acc;
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = a[(prow0 + 4 * e + lq) * LDA + c0 + lr];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int k = 4 * s4 + lq;
    double av = a[(prow0 + lr) * LDA + K0 + k];
    if (masked) av = (k > lr) ? av : ((k == lr) ? rinv[lr] : 0.0);
    const double bv = a[(c0 + lr) * LDA + K0 + k];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) a[(prow0 + 4 * e + lq) * LDA + c0 + lr] = acc[e];
}

// fp64 twin of diag_factor16 on v_mfma_f64_16x16x4_f64, whose C layout differs: lane (lr, lq), register e = element
// [4 e + lq][lr].  Row k of the symmetric block therefore sits in the 16 lanes of group lq = k % 4, register k / 4 --
// again where an MFMA reads one k-slot of its A and of its B operand.
__device__ __forceinline__ void diag_factor16_f64(double* a, int LDA, int K0, double* x16, double* rinv, int lane,
                                                  int* status) {
  const int lr = lane & 15, lq = lane >> 4;
  f64x4_t acc, xac;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = 4 * e + lq;
    acc[e] = a[(K0 + max(r, lr)) * LDA + K0 + min(r, lr)];     // the LDS image holds the lower triangle
    xac[e] = (r == lr) ? 1.0 : 0.0;
  }
  bool bad = false;
  double lcol[16], inv[16];
  double pk = lane_bcast(acc[0], 0);
  constexpr int XLAG = 3;
  auto x_update = [&](int k) {
    const int s = k & 3, e = k >> 2;
    const double nw = ((lq == s) && (lr == k)) ? inv[k] - 1.0 : -lcol[k] * inv[k];   // -(v - e_k) / l_kk
    const double xr = (lq == s) ? xac[e] : 0.0;
    xac = __builtin_amdgcn_mfma_f64_16x16x4f64(nw, xr, xac, 0, 0, 0);
  };
  const int junk = lane * LDA + 64;              // pad word of a row of the 64 x 65 image: never read
  double my_inv = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int s = k & 3, e = k >> 2;
    const bool low = (lq == s) && (lr >= k);
    const double am = low ? acc[e] : 0.0;                  // row k of the block where it is column k of L
    // 1 / sqrt(pk): v_rsq_f64 (about 26 bits) + two Newton steps y <- y (1.5 - (pk / 2) y^2): ~1e-16 relative, a
    // tenth of the instructions of 1.0 / sqrt(pk)  (a pivot <= 0 gives NaN / inf: flagged, the caller fails)
    double iv = __builtin_amdgcn_rsq(pk);
    {
      const double hp = 0.5 * pk;
      iv = iv * fma(-hp * iv, iv, 1.5);
      iv = iv * fma(-hp * iv, iv, 1.5);
    }
    if (!(pk > 0.0)) bad = true;
    const double v = am * iv, nv = -am * iv;               // L[lr][k]
    if (k < 15) {
      const int s1 = (k + 1) & 3, e1 = (k + 1) >> 2;
      const double l10 = lane_bcast(v, 16 * s + k + 1);            // L[k+1][k]
      const double a11 = lane_bcast(acc[e1], 16 * s1 + k + 1);     // a[k+1][k+1] before this column's update
      pk = fma(-l10, l10, a11);
    }
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(nv, v, acc, 0, 0, 0);
    lcol[k] = v;
    inv[k] = iv;
    a[low ? (K0 + lr) * LDA + K0 + k : junk] = v;          // L16
    my_inv = (lr == k) ? iv : my_inv;
    if (k >= XLAG) x_update(k - XLAG);
  }
#pragma unroll
  for (int k = 16 - XLAG; k < 16; ++k) x_update(k);
  if (bad && lane == 0) status[0] = 1;
  if (lane < 16) rinv[lane] = my_inv;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = 4 * e + lq;                                                         // xac[e] = X[r][lr]
    if (r > lr) a[(K0 + lr) * LDA + K0 + r] = xac[e];                                 // Z16[lr][r] = X[r][lr]
    x16[r * 17 + lr] = xac[e];
  }
}

// fp64 twin of k_panel_direct (block 64): P <- P Linv^T in place for `rows` rows, Linv the 64 x 64 lower-triangular
// inverse k_chol_diag_packed_f64 left in Dinv (row-major, stride 64).  One wave = 16 rows x all 64 columns, its A
// fragments (pairs of doubles: k = 8 u + 2 lq + e) in registers before its first store; Linv staged in LDS, row pitch
// 33 x 16 bytes; v_mfma_f64_16x16x4_f64 (C layout: register e = row 4 e + lq); column tile ct needs k < 16 (ct + 1).
__global__ void __launch_bounds__(256, 2) k_panel_direct_f64(double* __restrict__ P, int ldp, const double* __restrict__ Dinv,
                                                            int rows) {
  constexpr int NB = 64, PITCH = 66;
  __shared__ double sl[NB * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  __builtin_amdgcn_s_setprio(2);                 // part of the serial chain
  const int row0 = blockIdx.x * 64 + wave * 16;
  const bool live = row0 < rows;
  f64x2 fa[8];
  double* Prow = P + (size_t)(row0 + lr) * ldp;
  if (live) {
#pragma unroll
    for (int u = 0; u < 8; ++u) fa[u] = *reinterpret_cast<const f64x2*>(Prow + 8 * u + 2 * lq);
  }
  {
    f64x2 v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int q = tid + 256 * p;               // pair index: row q / 32, columns 2 (q % 32) ..
      v[p] = *reinterpret_cast<const f64x2*>(Dinv + (size_t)(q >> 5) * NB + 2 * (q & 31));
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int q = tid + 256 * p;
      *reinterpret_cast<f64x2*>(sl + (q >> 5) * PITCH + 2 * (q & 31)) = v[p];
    }
  }
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
    const double* bl = sl + (16 * ct + lr) * PITCH + 2 * lq;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u <= 2 * ct + 1) {
        const f64x2 fb = *reinterpret_cast<const f64x2*>(bl + 8 * u);
#pragma unroll
        for (int e = 0; e < 2; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[u][e], fb[e], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) P[(size_t)(row0 + 4 * e + lq) * ldp + 16 * ct + lr] = acc[e];
  }
}

__global__ void __launch_bounds__(512)
k_chol_diag_packed_f64(double* __restrict__ Aglob, int ld, double* __restrict__ Dinv, int* __restrict__ status,
                       int nblk_real = 4) {
  constexpr int NB = 64, LDA = NB + 1, NT = 512, NBLK = NB / 16;
  __shared__ double a[NB * LDA];
  __shared__ double x16[2][16 * 17];
  __shared__ double rinv[2][16];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  __builtin_amdgcn_s_setprio(3);
  for (int idx = tid; idx < NB * NB; idx += NT) {
    const int i = idx / NB, j = idx % NB;
    a[i * LDA + j] = (j <= i) ? Aglob[(size_t)i * ld + j] : 0.0;
  }
  __syncthreads();
  if (wave == 0) diag_factor16_f64(a, LDA, 0, x16[0], rinv[0], lane, status);
  __syncthreads();
  for (int b = 0; b < NBLK; ++b) {
    const int K0 = b * 16;
    const double* xb = x16[b & 1];
    const double* rb_inv = rinv[b & 1];
    const int nbelow = NBLK - 1 - b;
    // panel: row blocks {below} + {Z rows of earlier blocks}: NBLK - 1 of them, P = Y X16^T
    if (wave < NBLK - 1) {
      const int prow0 = (wave < nbelow) ? (K0 + 16 + wave * 16) : ((wave - nbelow) * 16);
      f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const double av = a[(prow0 + lr) * LDA + K0 + 4 * s4 + lq];
        const double bv = xb[lr * 17 + 4 * s4 + lq];                 // B[k][col] = X16[col][k]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) a[(prow0 + 4 * e + lq) * LDA + K0 + lr] = acc[e];
    }
    __syncthreads();
    if (nbelow == 0 || b + 1 >= nblk_real) break;
    // urgent tiles: block column b + 1 (the nbelow tiles at / below the diagonal + the Z tiles of row blocks 0..b)
    if (wave < NBLK) {
      const bool top = wave < nbelow;
      const int prow0 = top ? (K0 + 16 + wave * 16) : ((wave - nbelow) * 16);
      diag_tile_update_f64(a, LDA, prow0, K0 + 16, K0, !top && (wave - nbelow) == b, rb_inv, lr, lq);
    }
    // (no barrier: wave 0's urgent tile is the next diagonal tile, which only wave 0 goes on to read; see the f32 kernel)
    // wave 0 factors block b + 1 while the other waves apply the rest of update b (block columns b + 2 ..)
    if (wave == 0) {
      diag_factor16_f64(a, LDA, K0 + 16, x16[(b + 1) & 1], rinv[(b + 1) & 1], lane, status);
    } else {
      const int nb1 = nbelow - 1;
      const int ntri = nb1 * (nb1 + 1) / 2;
      const int nz = (b + 1) * nb1;
      for (int t = wave - 1; t < ntri + nz; t += NT / 64 - 1) {
        if (t < ntri) {
          int rb = 0, rem = t;
          while (rem > rb) { rem -= rb + 1; ++rb; }
          diag_tile_update_f64(a, LDA, K0 + 32 + rb * 16, K0 + 32 + rem * 16, K0, false, rb_inv, lr, lq);
        } else {
          const int u = t - ntri;
          const int tb = u / nb1, cb = u % nb1;
          diag_tile_update_f64(a, LDA, tb * 16, K0 + 32 + cb * 16, K0, tb == b, rb_inv, lr, lq);
        }
      }
    }
    __syncthreads();
  }
  for (int idx = tid; idx < NB * NB; idx += NT) {
    const int i = idx / NB, j = idx % NB;
    const double l = a[i * LDA + j];
    Aglob[(size_t)i * ld + j] = (j <= i) ? l : 0.0;
    Dinv[(size_t)i * NB + j] = (j < i) ? a[j * LDA + i] : ((j == i) ? 1.0 / l : 0.0);
  }
}

}  // namespace ekf
