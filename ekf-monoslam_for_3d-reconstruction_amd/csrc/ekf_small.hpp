// The whole EKF update of a SMALL map as ONE launch (f32; the reference's operating point, conf_sim.cfg: 20-35 features).
//
// What it replaces: k_sigma_ht (W = Sigma H^T, nu = z - h), k_innovation_cov (S = H W + R), k_chol_diag_packed
// (S = L L^T, L^-1) and k_update_oneblock_small (V = W L^-T, mu += V y, Sigma -= V V^T, quaternion normalisation): four
// launches that run back to back for 5.5 + 5.0 + 10.4 + 15.4 = 36.3 us at N = 32 (profiles/r6_step_timeline_n32.txt),
// each paying its own ramp, its own chain of dependent global loads (~1.3 us per round trip) and its own drain.
//
// How: k_update_oneblock_small already gives every 64 x 64 lower tile of Sigma its own workgroup and lets each of them
// re-derive what it needs (V_i, V_j, y, the new quaternion) instead of handing it over.  This kernel extends that to the
// front of the update: EVERY workgroup forms all of W (n x m: 0.5 MFLOP) from Sigma, S from W, and factors S itself --
// the same instructions in every workgroup, hence the same bits -- and then continues with its own tile.  Sigma is read
// by all of them while none has written yet; the one thing that has to be ordered across workgroups is "nobody writes
// Sigma or mu before everybody has read them": one arrival counter (`gate`), added to once per workgroup after its last
// read and polled once, in front of the first write -- by then (S, the factor and V lie in between) everybody has
// arrived long ago (0.2 us measured for poll + barrier).  The wait is bounded (status[3]); workgroups that are not
// resident yet have not arrived, and become resident because the grid (<= 11 workgroups at n <= 256) is far smaller than
// the chip.
//
// Arithmetic: every sum is formed by the instruction sequence of the kernel it replaces (W: k_sigma_ht's two chains per
// slot; S: k_innovation_cov's; the factor: diag_factor_lds itself; V, y, the products and the normalisation:
// k_update_oneblock_small's), so the results are BIT-IDENTICAL to the four-launch path (EKF_SMALL_ONELAUNCH=0;
// tests/test_gpu_parity.py asserts it for mu, Sigma, the gain and the workspaces, together with the launch counts).
// What is left out or added are exact zeros only: column tiles of V and K steps beyond the m live columns of the
// innovation; the three undefined Jacobian entries behind a 3-entry (XYZ) feature, multiplied as zeros.
//
// LDS (one workgroup per CU, 1024 lanes): `a` 128 x 132 (staging of Sigma rows, then S -> L / Z) and the image region
// (W as n x WP at its end, then V_i | V_j as 64 x 132 each at its front), 143 KB, + 15 KB of small arrays.
//
// Measured (tools/small_stamps.py, profiles/r6_small_onelaunch.txt; N = 32, plane rows, n = 206, m = 67): the launch takes
// 29.7 us (front loads 2.3, W 7.4, S 2.4, factor 9.2, y + operands 1.7, V 1.8, products 1.7, tile epilogue 2.7); predict +
// update 39.6 us against 42.9 with the four launches (N = 20: 32.8 against 40.4).  VERDICT r5's "<= 25 us per step" is out
// of reach for this structure: the factor of the 67 x 67 innovation (five 16-column blocks of the LDS-resident blocked
// Cholesky, 1.85 us each -- the same rate as in the chain of large maps) and the 7 us predict launch in front are 16 us
// by themselves.
#pragma once
#include "ekf_dense.hpp"

namespace ekf {

constexpr int kSmallImg = 18944;                 // floats of the W image (>= 2 * 64 * 132 for V_i | V_j)
constexpr int kSmallMaxRows = 256;               // n_pad of the maps this kernel takes

struct SmallUpdateArgs {
  float* S; int ld; int n; int npad;             // Sigma; live rows; rows padded to 128 (the nu row of W / V)
  const float* Hc; const float* Hf; const int* pos; const int* coding; const int* midx;
  int M, plane, nfeat;
  const float* z; const float* h; float* mu;
  float r_pix, r_plane;
  float* W; int ldw;                             // workspace images the other entry points read (ekf_peek_workspace, ekf_get_gain)
  float* Y; int ldy;                             // L in rows [0, 128), Zs = L^-T in rows [128, 256)
  float* Dinv;
  float* V; int ldv;
  float* scr_qn;
  int* status;
  unsigned* gate; unsigned gate_target;
  int ntiles;                                    // 64 x 64 lower tiles; workgroup ntiles holds the nu row
  int wp;                                        // pitch of the W image: m rounded up to 4
  int rc, nchunk;                                // rows of Sigma staged at a time (<= 64), number of such chunks (<= 4): small_chunking()
  unsigned long long* stamps;                    // diagnostics (EKF_SMALL_STAMPS=1): 100 MHz clock of workgroup 0 at the phase boundaries
};

// phase boundaries of workgroup 0 / wave 0 (tools/small_stamps.py); nothing but a uniform branch when off
// (kept in LDS until the end: a global store would be waited for at the next barrier and lengthen the phase it follows)
#define EKF_SMALL_STAMP(i) do { if (g.stamps && blockIdx.x == 0 && tid == 0) sstamp[i] = wall_clock64(); } while (0)

// Rows of Sigma staged per chunk: the staging area is `a` plus what the W image (kept at the END of the image region)
// leaves of that region; as few chunks as fit (every chunk costs two barriers and a pass of LDS latency).  false: no fit.
inline bool small_chunking(int n, int wp, int* rc, int* nchunk) {
  const int sp = 4 * ((n + 3) / 4);
  const long cap = 128L * 132 + ((long)kSmallImg - (long)n * wp) - 4;
  if ((long)n * wp > kSmallImg) return false;
  for (int c = 1; c <= 4; ++c) {
    const int r = (n + c - 1) / c;
    if (r <= 64 && (long)r * sp <= cap) { *rc = r; *nchunk = c; return true; }
  }
  return false;
}

__global__ void __launch_bounds__(1024, 1) k_update_small_onelaunch(SmallUpdateArgs g) {
  using namespace oneblock;                      // NB = 128, PITCH = 132, f4, normalise
  constexpr int LDA = PITCH, TP = 65;
  __shared__ __attribute__((aligned(16))) float big[NB * LDA + kSmallImg];
  __shared__ __attribute__((aligned(16))) float x16[2][16 * 20];
  __shared__ __attribute__((aligned(16))) float rinv[2][16];
  __shared__ float junk16[64 * 16];
  __shared__ __attribute__((aligned(16))) float sHc[64 * 16];      // 14 per slot, pitch 16: read as four 16-byte broadcasts
  __shared__ __attribute__((aligned(16))) float sHf[64 * 12];
  __shared__ float sdinv[NB];                    // 1 / l_cc
  __shared__ int sp[64], sfs[64];
  __shared__ __attribute__((aligned(16))) float snu[NB];
  __shared__ __attribute__((aligned(16))) float sy[NB];
  __shared__ float sq[4], sJ[16], sqold[4];
  __shared__ unsigned long long sstamp[16];
  float* const a = big;                          // staging of Sigma rows, then S -> the factor's image (L below, Z = L^-T above the diagonal), then the tile image
  float* const sVi = big + NB * LDA;             // V_i | V_j, 64 x 132 each
  float* const sVj = sVi + 64 * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int n = g.n, M = g.M, plane = g.plane, rows = g.npad;
  const int m = 2 * M + (plane ? 3 : 0);
  const int WP = g.wp, ns = WP >> 1;
  float* const img = big + NB * LDA + (kSmallImg - n * WP);    // W image (n x WP) at the end of the region: the staging of Sigma rows may run into its front
  const int nct = (m + 15) >> 4;                 // column tiles of 16 that hold live columns of the innovation
  const bool nu_wg = (int)blockIdx.x == g.ntiles;
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= (int)blockIdx.x) ++i;       // tile t -> (i, j), row-major over the lower triangle
  const int j = blockIdx.x - i * (i + 1) / 2;
  const bool owner = !nu_wg && (j == 0);
  // wave = (half, ww): ww < 4: 16 rows of V_i, ww >= 4: of V_j; the two halves share the column tiles of those rows
  const int half = wave >> 3, ww = wave & 7, rg = ww & 3;
  const bool second = ww >= 4;
  const int vrow0 = nu_wg ? rows + rg * 16 : 64 * (second ? j : i) + rg * 16;
  const bool v_active = nu_wg ? !second : (!second || j != i);
  const int rb = wave >> 2, cb = wave & 3;       // the wave's 16 x 16 block of the tile

  EKF_SMALL_STAMP(0);
  // ---- the rows of Sigma, RC at a time: wave w takes rows w, w + 16, ... of a chunk (up to four), lane = 16 bytes of the
  // row (coalesced).  Three chunks are requested at once (a request made one chunk ahead arrives too late: a round trip is
  // ~1.8 us, a chunk's work less); a fourth follows chunk 0 in its registers, and the barriers of this phase wait for LDS
  // only, never for that load -------------------------------------------------------------------------------------------
  const int nq = (n + 3) >> 2;
  const int RC = g.rc, nchunk = g.nchunk;
  f4 svA[4], svB[4], svC[4];
  auto load_chunk = [&](f4 (&buf)[4], int ch) {
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int rr = wave + 16 * pp, row = RC * ch + rr;
      buf[pp] = f4{0.f, 0.f, 0.f, 0.f};
      if (rr < RC && row < n && lane < nq) buf[pp] = *reinterpret_cast<const f4*>(g.S + (size_t)row * g.ld + 4 * lane);
    }
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
  load_chunk(svA, 0);
  if (nchunk > 1) load_chunk(svB, 1);
  if (nchunk > 2) load_chunk(svC, 2);
  // what this workgroup alone will overwrite, requested with everything else (a load issued later would be waited for
  // at the next barrier: ~1.3 us each time): the Sigma values of the wave's block of the tile, the owner's rows of mu
  float cin[4] = {0.f, 0.f, 0.f, 0.f};
  float mu_old[4] = {0.f, 0.f, 0.f, 0.f};
  if (!nu_wg) {
#pragma unroll
    for (int e = 0; e < 4; ++e) cin[e] = g.S[(size_t)(64 * i + 16 * rb + 4 * lq + e) * g.ld + 64 * j + 16 * cb + lr];
    if (owner && half == 0 && !second && lr == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = vrow0 + 4 * lq + e;
        if (r < n) mu_old[e] = g.mu[r];
      }
    }
  }
  // ---- the measured list, its Jacobians, the innovation, the quaternion before the update ----------------------------
  if (tid < M) {
    const int fi = clamp_feature(g.midx[tid], g.nfeat);
    sp[tid] = g.pos[fi];
    sfs[tid] = g.coding[fi] ? 3 : 6;
    if (nu_wg) check_measured_entry(g.midx, tid, g.nfeat, g.status);
  }
  for (int e = tid; e < M * 14; e += 1024) {
    const int k = e / 14;
    sHc[16 * k + (e - 14 * k)] = g.Hc[(size_t)clamp_feature(g.midx[k], g.nfeat) * 14 + (e - 14 * k)];
  }
  for (int e = tid; e < M * 12; e += 1024) {     // (the entries behind a 3-entry feature's Jacobian are not defined: exact zeros
    const int k = e / 12, t = e - 12 * k;        //  here, so that no sum below needs to skip them)
    const int fi = clamp_feature(g.midx[k], g.nfeat);
    const float v = g.Hf[(size_t)fi * 12 + t];
    sHf[e] = ((t % 6) < (g.coding[fi] ? 3 : 6)) ? v : 0.f;
  }
  if (tid >= 896) {                              // (a wave that holds no list entry)
    const int t = tid - 896;
    float v = 0.f;
    if (t < 2 * M) {
      v = g.z[t] - g.h[2 * clamp_feature(g.midx[t >> 1], g.nfeat) + (t & 1)];
    } else if (plane && t < 2 * M + 3) {
      const int e = t - 2 * M;
      v = -g.mu[e == 0 ? 1 : (e == 1 ? 4 : 6)];
    }
    snu[t] = v;
    if (t < 4) sqold[t] = g.mu[3 + t];
  }

  // ---- W = Sigma H^T into the image, RC rows of Sigma at a time through `a` (k_sigma_ht's sums) -------------------------
  EKF_SMALL_STAMP(1);
  const int SP = 4 * nq;                         // pitch of the staged rows (a 3-entry feature at the end of a row reads on into
                                                 // the next row's first words, or the four zero words behind the last one: unused, finite)
  // lane = measurement slot k (two columns of W): its Jacobian rows stay in registers for all rows of Sigma.  Every lane
  // runs the same two chains (k_sigma_ht's order: 7 camera terms, 6 feature terms): a 3-entry feature has exact zeros in the
  // last three factors; a slot behind the list has a zero Jacobian -- or, for the three plane columns, the unit row
  // e_1 / e_4 / e_6, which makes the chain return Sigma[r][1 | 4 | 6] itself (sum of that entry and exact zeros)
  const int kslot = lane;
  const bool k_meas = kslot < M, k_live = kslot < ns;
  float hc[14], hf[12];
  int kp = 0;
  bool all_even = true;
  auto w_chunk = [&](auto CH) {
    constexpr int ch = decltype(CH)::value;
    f4 (&buf)[4] = (ch == 1) ? svB : ((ch == 2) ? svC : svA);   // (chunk 3, if there is one, follows chunk 0 in svA)
    if (ch == 0) __syncthreads(); else lds_barrier();          // the rows of the chunk before are consumed (first pass: the lists are in LDS)
    if (ch == 0) {
#pragma unroll
      for (int t = 0; t < 12; ++t) hf[t] = 0.f;
      if (k_meas) {
        kp = sp[kslot];
#pragma unroll
        for (int t = 0; t < 14; ++t) hc[t] = sHc[16 * kslot + t];
#pragma unroll
        for (int t = 0; t < 12; ++t) hf[t] = sHf[12 * kslot + t];
      } else {
        const int pc0 = 2 * kslot - 2 * M;       // plane column index of the lane's first column (0..2 when it is one)
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          const int e = (t == 1) ? 0 : ((t == 4) ? 1 : ((t == 6) ? 2 : -1));
          hc[t] = (plane && e >= 0 && e == pc0) ? 1.f : 0.f;
          hc[7 + t] = (plane && e >= 0 && e == pc0 + 1) ? 1.f : 0.f;
        }
      }
      all_even = __all((kp & 1) == 0) != 0;
    }
    if (lane < nq) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp)
        if (wave + 16 * pp < RC) *reinterpret_cast<f4*>(a + (wave + 16 * pp) * SP + 4 * lane) = buf[pp];
    }
    if (tid == 0) *reinterpret_cast<f4*>(a + RC * SP) = f4{0.f, 0.f, 0.f, 0.f};
    if (ch == 0 && nchunk > 3) load_chunk(buf, 3);
    lds_barrier();
    EKF_SMALL_STAMP(11 + ch);                    // chunk ch is staged (chunk 0: the loads have arrived)
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {             // two rows (two independent pairs of chains) at a time
      if (wave + 32 * j2 >= RC) break;
      f4 c0[2], c1[2];
      float fv[2][6];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float* srow = a + (wave + 16 * (2 * j2 + jj)) * SP;
        c0[jj] = *reinterpret_cast<const f4*>(srow);           // (broadcasts)
        c1[jj] = *reinterpret_cast<const f4*>(srow + 4);
        if (all_even) {                          // (the usual layout, 14 + 6 f)
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const f2 v2 = *reinterpret_cast<const f2*>(srow + kp + 2 * t);
            fv[jj][2 * t] = v2[0];
            fv[jj][2 * t + 1] = v2[1];
          }
        } else {
#pragma unroll
          for (int t = 0; t < 6; ++t) fv[jj][t] = srow[kp + t];
        }
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float cv[7] = {c0[jj][0], c0[jj][1], c0[jj][2], c0[jj][3], c1[jj][0], c1[jj][1], c1[jj][2]};
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          a0 = __builtin_fmaf(cv[t], hc[t], a0);
          a1 = __builtin_fmaf(cv[t], hc[7 + t], a1);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          a0 = __builtin_fmaf(fv[jj][t], hf[t], a0);
          a1 = __builtin_fmaf(fv[jj][t], hf[6 + t], a1);
        }
        const int r = wave + 16 * (2 * j2 + jj);
        if (r < RC && RC * ch + r < n && k_live) *reinterpret_cast<f2*>(img + (RC * ch + r) * WP + 2 * kslot) = f2{a0, a1};
      }
    }
  };
  w_chunk(std::integral_constant<int, 0>{});
  if (nchunk > 1) w_chunk(std::integral_constant<int, 1>{});
  if (nchunk > 2) w_chunk(std::integral_constant<int, 2>{});
  if (nchunk > 3) w_chunk(std::integral_constant<int, 3>{});
  __syncthreads();                               // the image is complete; Sigma and mu are not read again:
  EKF_SMALL_STAMP(2);
  if (tid == 0) __hip_atomic_fetch_add(g.gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- S = H W + R into `a`: lower triangle, zeros above, identity padding (k_innovation_cov's sums) ---------------------
  {
    const int c = tid & 127, gsel = tid >> 7;
    const bool cin_img = c < WP;
    float wc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) wc[t] = cin_img ? img[t * WP + c] : 0.f;
    for (int kb = gsel; kb < M; kb += 16) {      // two features in flight (k is uniform in the wave: 16-byte broadcasts of H)
      float r0[2], r1[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = min(kb + 8 * u, M - 1);
        const int p = sp[k], fs = sfs[k];
        float hc[16], hf[12];
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f4*>(hc + 4 * t) = *reinterpret_cast<const f4*>(sHc + 16 * k + 4 * t);
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<f4*>(hf + 4 * t) = *reinterpret_cast<const f4*>(sHf + 12 * k + 4 * t);
        float wv[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) wv[t] = cin_img ? img[(p + min(t, fs - 1)) * WP + c] : 0.f;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          a0 = __builtin_fmaf(hc[t], wc[t], a0);
          a1 = __builtin_fmaf(hc[7 + t], wc[t], a1);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {            // (a 3-entry feature: three exact zeros in hf)
          a0 = __builtin_fmaf(hf[t], wv[t], a0);
          a1 = __builtin_fmaf(hf[6 + t], wv[t], a1);
        }
        if (c == 2 * k) a0 += g.r_pix;
        if (c == 2 * k + 1) a1 += g.r_pix;
        if (c >= m) { a0 = 0.f; a1 = 0.f; }
        r0[u] = (c <= 2 * k) ? a0 : 0.f;
        r1[u] = (c <= 2 * k + 1) ? a1 : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = kb + 8 * u;
        if (k < M) {
          a[(2 * k) * LDA + c] = r0[u];
          a[(2 * k + 1) * LDA + c] = r1[u];
        }
      }
    }
    for (int r = 2 * M + gsel; r < NB; r += 8) {
      float v = 0.f;
      if (r < m) {                               // plane rows: H = e1, e4, e6 -> rows 1, 4, 6 of W
        const int e = r - 2 * M;
        if (c < m) v = cin_img ? img[(e == 0 ? 1 : (e == 1 ? 4 : 6)) * WP + c] : 0.f;
        if (c == r) v += g.r_plane;
      } else if (c == r) {
        v = 1.f;
      }
      a[r * LDA + c] = (c <= r) ? v : 0.f;
    }
  }
  EKF_SMALL_STAMP(3);
  // ---- S = L L^T, Z = L^-T in the strict upper triangle (the body of k_chol_diag_packed; it starts with a barrier) -------
  {
    const DiagLds L{a, x16, rinv, junk16};
    diag_factor_lds<7>(g.status, max(1, min(8, nct)), L);
  }
  __builtin_amdgcn_s_setprio(0);
  __syncthreads();
  EKF_SMALL_STAMP(4);
  if (nu_wg) {
    // what the launches this one replaces leave in the workspaces: L, L^-1, the strip Zs = L^-T, W and the nu row
    for (int q = tid; q < NB * 32; q += 1024) {
      const int r = q >> 5, c0 = 4 * (q & 31);
      const f4 l = *reinterpret_cast<const f4*>(a + r * LDA + c0);
      const float lrr = a[r * LDA + r];
      f4 lo, zz;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo[e] = (c0 + e <= r) ? l[e] : 0.f;
        zz[e] = (c0 + e > r) ? l[e] : ((c0 + e == r) ? 1.f / lrr : 0.f);
      }
      *reinterpret_cast<f4*>(g.Y + (size_t)r * g.ldy + c0) = lo;
      *reinterpret_cast<f4*>(g.Y + (size_t)(NB + r) * g.ldy + c0) = zz;      // Zs[k][c] = Linv[c][k] = Z[k][c]
    }
    for (int q = tid; q < NB * NB; q += 1024) {
      const int jj = q & 127, ii = q >> 7;       // Linv[ii][jj] = Z[jj][ii]
      g.Dinv[(size_t)ii * NB + jj] = (jj < ii) ? a[jj * LDA + ii] : ((jj == ii) ? 1.f / a[jj * LDA + jj] : 0.f);
    }
    for (int q = tid; q < n * 32; q += 1024) {
      const int r = q >> 5, c0 = 4 * (q & 31);
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (c0 < WP) v = *reinterpret_cast<const f4*>(img + r * WP + c0);
      *reinterpret_cast<f4*>(g.W + (size_t)r * g.ldw + c0) = v;
    }
    if (tid < 32) *reinterpret_cast<f4*>(g.W + (size_t)rows * g.ldw + 4 * tid) = *reinterpret_cast<const f4*>(snu + 4 * tid);
    __syncthreads();
  }
  // L^-1 is read where the factor left it: Linv[r][k] = Z[k][r] = a[k][r] for k < r, 1 / l_rr on the diagonal, 0 above
  if (tid < NB) sdinv[tid] = 1.f / a[tid * LDA + tid];
  EKF_SMALL_STAMP(5);

  // ---- from here on: k_update_oneblock_small, on 16 waves --------------------------------------------------------------
  if (owner) {
    // y = Linv nu.  k_update_oneblock_small: two lanes per row (halves of k), four running sums each, (a0 + a1) + (a2 + a3),
    // then the two halves added.  Here every one of the eight chains of a row has its own lane (the same chains, the same
    // additions: the same bits), 128 rows x 8 = all 1024 lanes
    const int r = tid >> 3, k0 = 64 * ((tid >> 2) & 1) + (tid & 3);
    const float dr = 1.f / a[r * LDA + r];
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int k = k0 + 4 * t;
      const float zv = a[k * LDA + r];
      acc = __builtin_fmaf((k < r) ? zv : ((k == r) ? dr : 0.f), snu[k], acc);
    }
    acc += __shfl_xor(acc, 1, 64);               // a0 + a1 | a2 + a3
    acc += __shfl_xor(acc, 2, 64);               // (a0 + a1) + (a2 + a3)
    acc += __shfl_xor(acc, 4, 64);               // + the other half of k
    if ((tid & 7) == 0) sy[r] = acc;
  }
  f4 fa[8];
  if (v_active) {
    const int row = vrow0 + lr;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int col = 16 * u + 4 * lq;
      fa[u] = f4{0.f, 0.f, 0.f, 0.f};
      if (nu_wg) {
        if (row == rows) fa[u] = *reinterpret_cast<const f4*>(snu + col);
      } else if (row < n && col < WP) {
        fa[u] = *reinterpret_cast<const f4*>(img + row * WP + col);
      }
    }
  }
  __syncthreads();                               // every wave holds its rows of W: the image becomes V_i | V_j; y is there
  EKF_SMALL_STAMP(6);
  if (v_active) {
    float* svp = (second ? sVj : sVi) + (rg * 16) * PITCH;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const int back = nct - 1 - ct;             // live tiles are dealt heaviest-first: A B B A A B B A
      const bool mine = (ct < nct) ? ((((back & 3) == 0) || ((back & 3) == 3)) ? half == 0 : half == 1) : (half == (ct & 1));
      if (!mine) continue;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      if (ct < nct) {
        const int c = 16 * ct + lr;              // fb[e] = Linv[c][k], k = 16 u + 4 lq + e
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
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        svp[(4 * lq + e) * PITCH + 16 * ct + lr] = acc[e];
      }
    }
  }
  // (V_i goes to the workspace from LDS at the very end, in whole rows: nothing waits for those stores)
  auto store_v_rows = [&](int row0) {
    for (int q = tid; q < 64 * 32; q += 1024) {
      const int r = q >> 5, c0 = 4 * (q & 31);
      *reinterpret_cast<f4*>(g.V + (size_t)(row0 + r) * g.ldv + c0) = *reinterpret_cast<const f4*>(sVi + r * PITCH + c0);
    }
  };
  if (nu_wg) {
    __syncthreads();
    store_v_rows(rows);
    return;
  }
  EKF_SMALL_STAMP(7);
  if (tid == 0) {                                // nobody writes Sigma or mu before every workgroup has read them
    bool ok = false;
    for (int spin = 0; spin < (1 << 20) && !ok; ++spin) {
      ok = (int)(__hip_atomic_load(g.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.gate_target) >= 0;
      if (!ok) __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) g.status[3] = 1;
  }
  __syncthreads();                               // V_i, V_j complete; nobody reads Linv any more; the gate is open
  EKF_SMALL_STAMP(8);
  const float* vj = (j != i) ? sVj : sVi;
  if (owner && half == 0 && !second) {           // mu += V_i y (the quaternion rows: below, from q')
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const float yc = sy[16 * ct + lr];
#pragma unroll
      for (int e = 0; e < 4; ++e) part[e] = __builtin_fmaf(sVi[(rg * 16 + 4 * lq + e) * PITCH + 16 * ct + lr], yc, part[e]);
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) part[e] += __shfl_xor(part[e], off, 64);
    if (lr == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = vrow0 + 4 * lq + e;
        if (r < n && !(r >= 3 && r < 7)) g.mu[r] = mu_old[e] + part[e];
      }
    }
  }
  if (owner && wave == 8) {                      // (a wave without rows of the state update)
    // q' = q_old + V[3:7] y (rows 3..6 of block 0), lane = column (two sweeps), butterfly sum
    for (int q4 = 0; q4 < 4; ++q4) {
      float acc = vj[(3 + q4) * PITCH + lane] * sy[lane];
      acc = __builtin_fmaf(vj[(3 + q4) * PITCH + 64 + lane], sy[64 + lane], acc);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) sq[q4] = sqold[q4] + acc;
    }
    if (lane == 0) {
      float q[4] = {sq[0], sq[1], sq[2], sq[3]}, Qn[16];
      normalise(q, Qn);
      for (int k = 0; k < 16; ++k) sJ[k] = Qn[k];
      if (i == 0) {
        for (int k = 0; k < 4; ++k) g.mu[3 + k] = q[k];
        for (int k = 0; k < 16; ++k) g.scr_qn[k] = Qn[k];
      }
    }
  }
  f4 acc2 = {0.f, 0.f, 0.f, 0.f};
  {
    const float* ap = sVi + (16 * rb + lr) * PITCH + 4 * lq;
    const float* bp = vj + (16 * cb + lr) * PITCH + 4 * lq;
#pragma unroll
    for (int u = 0; u < 8; ++u) {                // lane (lr, lq) of MFMA (u, e) multiplies k = 16 u + 4 lq + e on both operands
      if (u < nct) {
        const f4 fav = *reinterpret_cast<const f4*>(ap + 16 * u);
        const f4 fbv = *reinterpret_cast<const f4*>(bp + 16 * u);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fav[e], fbv[e], acc2, 0, 0, 0);
      }
    }
  }
  EKF_SMALL_STAMP(9);
  // acc2[e] = (V_i V_j^T)[16 rb + 4 lq + e][16 cb + lr]
  if (!owner) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t gr = 64 * i + 16 * rb + 4 * lq + e, gc = 64 * j + 16 * cb + lr;
      if (gr >= gc) {                            // a diagonal tile: its lower half, mirrored (the upper lanes must not touch it)
        const float v = cin[e] - acc2[e];
        g.S[gr * g.ld + gc] = v;
        g.S[gc * g.ld + gr] = v;
      }
    }
    return;
  }
  // tiles (i, 0): the downdated tile through LDS, the normalisation congruence on columns 3..6 (and, tile (0, 0),
  // rows 3..6 and the corner), then the tile and its mirror
  float* sT = a;
#pragma unroll
  for (int e = 0; e < 4; ++e) sT[(16 * rb + 4 * lq + e) * TP + 16 * cb + lr] = cin[e] - acc2[e];
  __syncthreads();                               // (also: sJ is there)
  if (i == 0) {
    // the lower half is what counts: complete the image symmetrically first, so that rows and columns read the same values
    for (int q = tid; q < 64 * 64; q += 1024) {
      const int r = q >> 6, c = q & 63;
      if (c > r) sT[r * TP + c] = sT[c * TP + r];
    }
    __syncthreads();
  }
  if (tid < 64) {                                // column strip: row r, columns 3..6 (rows outside the block)
    const int r = tid;
    if (i > 0 || r < 3 || r >= 7) {
      float x[4], yv[4];
      for (int k = 0; k < 4; ++k) x[k] = sT[r * TP + 3 + k];
      for (int c = 0; c < 4; ++c) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s = __builtin_fmaf(x[k], sJ[c * 4 + k], s);
        yv[c] = s;
      }
      for (int c = 0; c < 4; ++c) sT[r * TP + 3 + c] = yv[c];
    }
  } else if (i == 0 && tid < 128) {              // row strip of tile (0, 0): column c, rows 3..6
    const int c = tid - 64;
    if (c < 3 || c >= 7) {
      float x[4], yv[4];
      for (int k = 0; k < 4; ++k) x[k] = sT[(3 + k) * TP + c];
      for (int r = 0; r < 4; ++r) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s = __builtin_fmaf(sJ[r * 4 + k], x[k], s);
        yv[r] = s;
      }
      for (int r = 0; r < 4; ++r) sT[(3 + r) * TP + c] = yv[r];
    }
  } else if (i == 0 && tid == 128) {             // corner: lower half of J C J^T, mirrored
    float C[16], A[16];
    for (int k = 0; k < 16; ++k) C[k] = sT[(3 + k / 4) * TP + 3 + k % 4];
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s += sJ[r * 4 + k] * C[k * 4 + c];
        A[r * 4 + c] = s;
      }
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c <= r; ++c) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * sJ[c * 4 + k];
        sT[(3 + r) * TP + 3 + c] = s;
        sT[(3 + c) * TP + 3 + r] = s;
      }
  }
  __syncthreads();
  for (int q = tid; q < 64 * 64; q += 1024) {
    const int r = q >> 6, c = q & 63;
    if (i == 0) {
      g.S[(size_t)r * g.ld + c] = (r >= c) ? sT[r * TP + c] : sT[c * TP + r];
    } else {
      g.S[(size_t)(64 * i + r) * g.ld + c] = sT[r * TP + c];
      g.S[(size_t)r * g.ld + 64 * i + c] = sT[c * TP + r];         // the mirror tile (0, i): row r, column 64 i + c
    }
  }
  store_v_rows(64 * i);
  EKF_SMALL_STAMP(10);
  if (g.stamps && blockIdx.x == 0 && tid == 0)
    for (int k = 0; k < 16; ++k) g.stamps[k] = sstamp[k];
}

}  // namespace ekf
