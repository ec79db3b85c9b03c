// HBM-/latency-bound kernels of the EKF hot path: camera motion step, covariance strip
// propagation (in place and streaming), per-feature measurement + Jacobian, quaternion
// normalisation strips, add-feature border and the remove/convert compaction pass.
// Device storage: Sigma row-major, `ld` elements per row, zero outside the live n x n.
#pragma once
#include "ekf_math.hpp"

namespace ekf {

// Scratch block (device, T-typed) shared between the small camera kernels.
enum : int {
  SCR_FT = 0,         // 13x13 motion Jacobian, row-major
  SCR_Q = 169,        // 13x13 process noise
  SCR_QN = 338,       // 4x4 quaternion-normalisation Jacobian
  SCR_G = 354,        // add-feature: 6x7 d f / d[r,q]
  SCR_C = 396,        // add-feature: 6x6 corner block
  SCR_QOLD = 432,     // the quaternion before the update in flight (k_sigma_ht's nu slab -> k_update_oneblock_small)
  SCR_SIZE = 512
};

struct MotionArgs {
  double dT;
  double t_ctl[3];
  double r_ctl[3];
  double vdiag[6];    // Vmax or Vmax_n diagonal (vR.cpp:194-202, 463-473)
};

// ---------------------------------------------------------------------------------------
// a2 + a3: Ft, Q and the camera state step.  One lane; ~1k flops.   (vR.cpp:1492-1535, 1575-1589)
// ---------------------------------------------------------------------------------------
// The one-lane core: sx = camera state (13), sF = Ft in LDS, preset to the identity (nullptr: state step only),
// xn = the predicted camera state (13).
template <typename T>
__device__ __forceinline__ void camera_step(const T* sx, const MotionArgs& a, T* sF, T* xn) {
  const T dTt = T(a.dT);
  const T q[4] = {sx[3], sx[4], sx[5], sx[6]};
  if (sF) {
    const T wc[3] = {sx[10] + T(a.r_ctl[0]), sx[11] + T(a.r_ctl[1]), sx[12] + T(a.r_ctl[2])};
    const T hv[3] = {dTt * wc[0], dTt * wc[1], dTt * wc[2]};
    T h[4];
    vec2quat(hv, h);
    // Ft[3:7,3:7] = Upsilon-bar(h)
    const T hb[16] = {h[0], -h[1], -h[2], -h[3],
                      h[1],  h[0],  h[3], -h[2],
                      h[2], -h[3],  h[0],  h[1],
                      h[3],  h[2], -h[1],  h[0]};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) sF[(3 + r) * 13 + 3 + c] = hb[r * 4 + c];
    // Ft[3:7,10:13] = Upsilon(q) * t2(w, dT); trigonometry in double like the reference (dT is double)
    const T nw = t_sqrt(wc[0] * wc[0] + wc[1] * wc[1] + wc[2] * wc[2]);
    const double ang = a.dT * double(nw) / 2.0;
    const T s = T(sin(ang));
    const T c = T(cos(ang));
    const T sinc = (nw == T(0)) ? T(1) : T(2.0 * sin(ang) / (a.dT * double(nw)));
    T n_w[3] = {T(0), T(0), T(0)};
    if (nw > T(0)) { n_w[0] = wc[0] / nw; n_w[1] = wc[1] / nw; n_w[2] = wc[2] / nw; }
    T t2[12];
    const T a0 = T(-a.dT * 0.5 * double(s));
    const T half = T(a.dT * 0.5);
#pragma unroll
    for (int j = 0; j < 3; ++j) t2[j] = a0 * n_w[j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        t2[(1 + i) * 3 + j] = half * ((i == j ? sinc : T(0)) + (c - sinc) * n_w[i] * n_w[j]);
    const T up[16] = {q[0], -q[1], -q[2], -q[3],
                      q[1],  q[0], -q[3],  q[2],
                      q[2],  q[3],  q[0], -q[1],
                      q[3], -q[2],  q[1],  q[0]};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += up[r * 4 + k] * t2[k * 3 + j];
        sF[(3 + r) * 13 + 10 + j] = acc;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i) sF[i * 13 + 7 + i] = dTt;
  }
  // Predict_State
  const T v[3] = {sx[7] + T(a.t_ctl[0]), sx[8] + T(a.t_ctl[1]), sx[9] + T(a.t_ctl[2])};
  const T w[3] = {sx[10] + T(a.r_ctl[0]), sx[11] + T(a.r_ctl[1]), sx[12] + T(a.r_ctl[2])};
  T dq[4], qn[4];
  const T wv[3] = {dTt * w[0], dTt * w[1], dTt * w[2]};
  vec2quat(wv, dq);
  quat_mul(q, dq, qn);
#pragma unroll
  for (int i = 0; i < 3; ++i) xn[i] = sx[i] + dTt * v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) xn[3 + i] = qn[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) { xn[7 + i] = v[i]; xn[10 + i] = w[i]; }
}

// Q = Ft[:,7:13] diag(V / dT / dT) Ft[:,7:13]^T, entry e = 13 i + j
template <typename T>
__device__ __forceinline__ T process_noise_entry(const T* sF, const MotionArgs& a, int e) {
  const int i = e / 13, j = e % 13;
  const T dTt = T(a.dT);
  T acc = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) acc += sF[i * 13 + 7 + k] * (T(a.vdiag[k]) / dTt / dTt) * sF[j * 13 + 7 + k];
  return acc;
}

template <typename T>
__global__ void k_predict_camera(T* __restrict__ mu, T* __restrict__ scr, MotionArgs a) {
  // one workgroup of 64 lanes: lane 0 forms the two non-trivial blocks of Ft, all lanes fill Ft / Q
  __shared__ T sF[169];
  __shared__ T sx[13];
  __shared__ T sxn[13];
  const int tid = threadIdx.x;
  if (blockIdx.x != 0) return;
  if (tid < 13) sx[tid] = mu[tid];
  for (int i = tid; i < 169; i += blockDim.x) sF[i] = (i / 13 == i % 13) ? T(1) : T(0);
  __syncthreads();
  if (tid == 0) camera_step(sx, a, sF, sxn);
  __syncthreads();
  if (tid < 13) mu[tid] = sxn[tid];
  // Ft and Q, one entry per lane-iteration
  for (int e = tid; e < 169; e += blockDim.x) {
    scr[SCR_FT + e] = sF[e];
    scr[SCR_Q + e] = process_noise_entry(sF, a, e);
  }
}

// ---------------------------------------------------------------------------------------
// a1 / a11 in place: Sigma <- J Sigma J^T (+Q) for J = identity except a KxK block at (o,o).
// Only rows/cols o..o+K-1 change: 2 K n elements.  One launch; three disjoint regions.
//   lane t <  n : column t of the row strip   (rows o..o+K-1, any column outside the block)
//   lane t >= n : row (t-n) of the column strip
//   block 0 additionally owns the KxK corner.
// ---------------------------------------------------------------------------------------
// The body for workgroup `bid` of `nthreads` lanes; sJ (K x K) and, for bid 0, Qs (K x K or nullptr) may live in LDS or
// in global memory; sC / sA: two K x K LDS scratch blocks (used by bid 0 only).
template <typename T, int K>
__device__ __forceinline__ void strip_congruence_body(T* __restrict__ S, int ld, int n, int o, const T* sJ, const T* Qs,
                                                      T* sC, T* sA, int bid, int tid, int nthreads) {
  const int t = bid * nthreads + tid;
  if (t < n) {
    if (t < o || t >= o + K) {             // row strip, column t
      T x[K], y[K];
#pragma unroll
      for (int k = 0; k < K; ++k) x[k] = S[(size_t)(o + k) * ld + t];
#pragma unroll
      for (int r = 0; r < K; ++r) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < K; ++k) acc = t_fma(sJ[r * K + k], x[k], acc);
        y[r] = acc;
      }
#pragma unroll
      for (int r = 0; r < K; ++r) S[(size_t)(o + r) * ld + t] = y[r];
    }
  } else if (t < 2 * n) {
    const int i = t - n;
    if (i < o || i >= o + K) {             // column strip, row i
      T x[K], y[K];
#pragma unroll
      for (int k = 0; k < K; ++k) x[k] = S[(size_t)i * ld + o + k];
#pragma unroll
      for (int c = 0; c < K; ++c) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < K; ++k) acc = t_fma(x[k], sJ[c * K + k], acc);
        y[c] = acc;
      }
#pragma unroll
      for (int c = 0; c < K; ++c) S[(size_t)i * ld + o + c] = y[c];
    }
  }
  if (bid == 0) {                          // corner: J C J^T + Q through LDS
    for (int i = tid; i < K * K; i += nthreads)
      sC[i] = S[(size_t)(o + i / K) * ld + o + i % K];
    __syncthreads();
    for (int i = tid; i < K * K; i += nthreads) {
      const int r = i / K, c = i % K;
      T acc = T(0);
      for (int k = 0; k < K; ++k) acc += sJ[r * K + k] * sC[k * K + c];
      sA[i] = acc;
    }
    __syncthreads();
    // lower triangle only, mirrored: (J C) J^T is symmetric in exact arithmetic, not in its rounding -- and the
    // rest of the update keeps Sigma EXACTLY symmetric (the downdate mirrors its lower tiles)
    for (int i = tid; i < K * K; i += nthreads) {
      const int r = i / K, c = i % K;
      if (c > r) continue;
      T acc = T(0);
      for (int k = 0; k < K; ++k) acc += sA[r * K + k] * sJ[c * K + k];
      if (Qs) acc += Qs[i];
      S[(size_t)(o + r) * ld + o + c] = acc;
      S[(size_t)(o + c) * ld + o + r] = acc;
    }
  }
}

template <typename T, int K>
__global__ void k_strip_congruence(T* __restrict__ S, int ld, int n, int o,
                                   const T* __restrict__ Jm, const T* __restrict__ Qm) {
  __shared__ T sJ[K * K];
  __shared__ T sC[K * K];
  __shared__ T sA[K * K];
  for (int i = threadIdx.x; i < K * K; i += blockDim.x) sJ[i] = Jm[i];
  __syncthreads();
  strip_congruence_body<T, K>(S, ld, n, o, sJ, Qm, sC, sA, blockIdx.x, threadIdx.x, blockDim.x);
}

// ---------------------------------------------------------------------------------------
// a1 streaming: dst = F src F^T + Q out of place, F = blkdiag(Ft13, I).  Reads n^2, writes n^2
// (the HBM-roofline formulation of P-propagate).  Rows >= 13 are a float4 copy except their
// first 13 columns; rows < 13 are the row strip.  grid.x = rows, 256 lanes sweep the row.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void k_propagate_streaming(const T* __restrict__ src, T* __restrict__ dst, int ld, int n,
                                      const T* __restrict__ Ft, const T* __restrict__ Qm) {
  constexpr int K = 13;
  constexpr int V = 16 / sizeof(T);            // elements per 16-byte vector
  typedef T vec_t __attribute__((ext_vector_type(V)));
  __shared__ T sF[K * K];
  for (int i = threadIdx.x; i < K * K; i += blockDim.x) sF[i] = Ft[i];
  __syncthreads();
  for (int row = blockIdx.x; row < n; row += gridDim.x) {
    if (row >= K) {
      const T* s = src + (size_t)row * ld;
      T* d = dst + (size_t)row * ld;
      // vector chunks [16, n): pure copy (16 is the first vector boundary past column 12)
      const int nv = (n + V - 1) / V;            // ld is padded, reading to the vector end is in-bounds
      // (eight loads in flight per lane before the first store)
      for (int cv0 = 16 / V + threadIdx.x; cv0 < nv; cv0 += 8 * blockDim.x) {
        vec_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int cv = cv0 + u * blockDim.x;
          if (cv < nv) v[u] = *reinterpret_cast<const vec_t*>(s + (size_t)cv * V);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int cv = cv0 + u * blockDim.x;
          if (cv < nv) *reinterpret_cast<vec_t*>(d + (size_t)cv * V) = v[u];
        }
      }
      if (threadIdx.x < 16) {                   // columns 0..15: 13 transformed + 3 copied
        const int c = threadIdx.x;
        T acc;
        if (c < K) {
          acc = T(0);
#pragma unroll
          for (int k = 0; k < K; ++k) acc = t_fma(s[k], sF[c * K + k], acc);
        } else {
          acc = s[c];
        }
        d[c] = acc;
      }
    }
  }
  // rows 0..12: column-parallel strip + corner, handled by the first blocks
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int t = t0; t < n; t += gridDim.x * blockDim.x) {
    T x[K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = src[(size_t)k * ld + t];
    if (t >= K) {
#pragma unroll
      for (int r = 0; r < K; ++r) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < K; ++k) acc = t_fma(sF[r * K + k], x[k], acc);
        dst[(size_t)r * ld + t] = acc;
      }
    }
  }
  if (blockIdx.x == 0) {
    __shared__ T sC[K * K];
    __shared__ T sA[K * K];
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) sC[i] = src[(size_t)(i / K) * ld + i % K];
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) {
      const int r = i / K, c = i % K;
      T acc = T(0);
      for (int k = 0; k < K; ++k) acc += sF[r * K + k] * sC[k * K + c];
      sA[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) {       // lower triangle, mirrored (see k_strip_congruence)
      const int r = i / K, c = i % K;
      if (c > r) continue;
      T acc = T(0);
      for (int k = 0; k < K; ++k) acc += sA[r * K + k] * sF[c * K + k];
      acc += Qm[i];
      dst[(size_t)r * ld + c] = acc;
      dst[(size_t)c * ld + r] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------
// a4 / a5 / a6: one lane per feature: h, compact H (2x7 + 2x6), flags.  flags bit0 = visible (vR.cpp:529), bit1 = rho <= 0 (vR.cpp:517-521).
// Hc: [N][2][7], Hf: [N][2][6] row-major (XYZ features: last 3 columns zero).
// ---------------------------------------------------------------------------------------
// `list` (count entries) restricts the sweep to the listed features; `cam_pose` (r, q: 7 scalars)
// overrides the camera part of mu -- the high-innovation rescue linearises the features of the UPDATED
// state around the camera pose of the state BEFORE the first update (vR.cpp:1069-1072, 1085).
template <typename T>
__device__ __forceinline__ void measure_feature(int i, const T* __restrict__ mu, const int* __restrict__ pos,
                                                const int* __restrict__ coding, const CamParams& cam,
                                                T* __restrict__ h_out, T* __restrict__ Hc, T* __restrict__ Hf,
                                                unsigned char* __restrict__ flags, const T* cam_pose);

template <typename T>
__global__ void k_measure(const T* __restrict__ mu,
                          const int* __restrict__ pos, const int* __restrict__ coding, int f_begin, int N,
                          CamParams cam,
                          T* __restrict__ h_out, T* __restrict__ Hc, T* __restrict__ Hf,
                          unsigned char* __restrict__ flags,
                          const int* __restrict__ list = nullptr, int count = 0,
                          const T* __restrict__ cam_pose = nullptr) {
  int i = f_begin + blockIdx.x * blockDim.x + threadIdx.x;   // features [f_begin, N) or list[0..count)
  if (list) {
    if (i >= count) return;
    i = list[i];
  }
  if (i >= N) return;
  measure_feature<T>(i, mu, pos, coding, cam, h_out, Hc, Hf, flags, cam_pose);
}

template <typename T>
__device__ __forceinline__ void measure_feature(int i, const T* __restrict__ mu, const int* __restrict__ pos,
                                                const int* __restrict__ coding, const CamParams& cam,
                                                T* __restrict__ h_out, T* __restrict__ Hc, T* __restrict__ Hf,
                                                unsigned char* __restrict__ flags, const T* cam_pose) {
  const int p = pos[i];
  const bool xyz = coding[i] != 0;
  const int fs = xyz ? 3 : 6;
  const T* cp7 = cam_pose ? cam_pose : mu;
  const T r[3] = {cp7[0], cp7[1], cp7[2]};
  const T qc[4] = {cp7[3], -cp7[4], -cp7[5], -cp7[6]};   // complement, vR.cpp:493
  T R[9];
  quat2rot(qc, R);
  T f[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  for (int k = 0; k < fs; ++k) f[k] = mu[p + k];
  T d[3];
  T Jf[18];                                          // d(d)/d(feature), 3x6 row-major
  T scale_r;
  T hc[14], hf[12];
  for (int k = 0; k < 14; ++k) hc[k] = T(0);
  for (int k = 0; k < 12; ++k) hf[k] = T(0);
  unsigned char fl = 0;
  if (!xyz) {
    const T theta = f[3], phi = f[4], ro = f[5];
    // rho <= 0: flagged for removal and never "visible" (vR.cpp:517-522).  h / H are still
    // evaluated (the formulas are regular there) so a caller that forces the feature into an
    // update list gets a consistent linearisation instead of stale values.
    if (ro <= T(0)) fl |= 2;
    const T st = t_sin(theta), ct = t_cos(theta), sp = t_sin(phi), cp = t_cos(phi);
    const T m[3] = {st * cp, -sp, ct * cp};
    const T ar[3] = {f[0] - r[0], f[1] - r[1], f[2] - r[2]};
    for (int k = 0; k < 3; ++k) d[k] = ro * ar[k] + m[k];
    for (int k = 0; k < 18; ++k) Jf[k] = T(0);
    Jf[0] = ro; Jf[7] = ro; Jf[14] = ro;
    Jf[3] = ct * cp;   Jf[9] = T(0);  Jf[15] = -st * cp;
    Jf[4] = -st * sp;  Jf[10] = -cp;  Jf[16] = -ct * sp;
    Jf[5] = ar[0];     Jf[11] = ar[1]; Jf[17] = ar[2];
    scale_r = -ro;
  } else {
    for (int k = 0; k < 3; ++k) d[k] = f[k] - r[k];
    for (int k = 0; k < 18; ++k) Jf[k] = T(0);
    Jf[0] = T(1); Jf[7] = T(1); Jf[14] = T(1);
    scale_r = T(-1);
  }
  T hC[3];
  mat3_vec(R, d, hC);
  T hd[2], Jp[6];
  project_distort(cam, hC, hd, Jp);
  if (!(fl & 2) && inside_image(cam, hd[0], hd[1]) && hC[2] >= T(0)) fl |= 1;
  // JR = Jp * R (2x3)
  T JR[6];
  for (int a = 0; a < 2; ++a)
    for (int c = 0; c < 3; ++c)
      JR[a * 3 + c] = Jp[a * 3 + 0] * R[0 * 3 + c] + Jp[a * 3 + 1] * R[1 * 3 + c] + Jp[a * 3 + 2] * R[2 * 3 + c];
  T Jq[12];
  drot_dq_times(qc, d, Jq);                         // d(R(qc) d)/d qc, then * diag(1,-1,-1,-1)
  for (int a = 0; a < 2; ++a) {
    for (int c = 0; c < 3; ++c) hc[a * 7 + c] = scale_r * JR[a * 3 + c];
    for (int c = 0; c < 4; ++c) {
      T acc = Jp[a * 3 + 0] * Jq[0 * 4 + c] + Jp[a * 3 + 1] * Jq[1 * 4 + c] + Jp[a * 3 + 2] * Jq[2 * 4 + c];
      hc[a * 7 + 3 + c] = (c == 0) ? acc : -acc;
    }
    for (int c = 0; c < fs; ++c)
      hf[a * 6 + c] = JR[a * 3 + 0] * Jf[0 * 6 + c] + JR[a * 3 + 1] * Jf[1 * 6 + c] + JR[a * 3 + 2] * Jf[2 * 6 + c];
  }
  h_out[2 * i] = hd[0];
  h_out[2 * i + 1] = hd[1];
  for (int k = 0; k < 14; ++k) Hc[(size_t)i * 14 + k] = hc[k];
  for (int k = 0; k < 12; ++k) Hf[(size_t)i * 12 + k] = hf[k];
  flags[i] = fl;
}

// ---------------------------------------------------------------------------------------
// a1 + a2 + a3 + a4 in ONE launch (the in-place propagate; small maps are launch-bound): workgroups [0, nstrip) are
// the strip congruence of a1, the rest one lane per feature of a4 / a5.  Every workgroup reads the OLD camera state
// and repeats the one-lane step of a2 / a3 for itself (the same instructions as k_predict_camera: bit-identical); workgroup 0
// commits the new camera state, Ft and Q once every other workgroup has read the old one (`gate`: one arrival per
// workgroup; nobody but workgroup 0 waits, so the order in which workgroups become resident does not matter).
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_predict_fused(T* __restrict__ mu, T* __restrict__ scr, MotionArgs a,
                                                       T* __restrict__ S, int ld, int n, int nstrip,
                                                       const int* __restrict__ pos, const int* __restrict__ coding, int N,
                                                       CamParams cam, T* __restrict__ h_out, T* __restrict__ Hc,
                                                       T* __restrict__ Hf, unsigned char* __restrict__ flags,
                                                       int* __restrict__ gate, int* __restrict__ status) {
  __shared__ T sF[169];
  __shared__ T sQ[169];
  __shared__ T sC[169];
  __shared__ T sA[169];
  __shared__ T sx[13];
  __shared__ T sxn[13];
  const int tid = threadIdx.x, bid = blockIdx.x;
  const bool strip = bid < nstrip;
  if (tid < 13) sx[tid] = mu[tid];
  for (int i = tid; i < 169; i += 256) sF[i] = (i / 13 == i % 13) ? T(1) : T(0);
  __syncthreads();                                   // the old camera state is in LDS: this workgroup is done with mu[0:13]
  if (tid == 0) {
    if (bid != 0) __hip_atomic_fetch_add(gate, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    camera_step(sx, a, strip ? sF : static_cast<T*>(nullptr), sxn);
  }
  __syncthreads();
  if (!strip) {
    const int i = (bid - nstrip) * 256 + tid;
    if (i < N) measure_feature<T>(i, mu, pos, coding, cam, h_out, Hc, Hf, flags, sxn);
    return;
  }
  if (bid == 0) {
    for (int e = tid; e < 169; e += 256) {
      const T qv = process_noise_entry(sF, a, e);
      sQ[e] = qv;
      scr[SCR_FT + e] = sF[e];
      scr[SCR_Q + e] = qv;
    }
    __syncthreads();
  }
  strip_congruence_body<T, 13>(S, ld, n, 0, sF, sQ, sC, sA, bid, tid, 256);
  if (bid == 0 && tid == 0) {
    bool ok = false;
    for (int spin = 0; spin < (1 << 20) && !ok; ++spin) {
      ok = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (int)gridDim.x - 1;
      if (!ok) __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) status[3] = 1;
    for (int k = 0; k < 13; ++k) mu[k] = sxn[k];
    __hip_atomic_store(gate, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// What a drop-in caller reads back after every predict (ekf_get_predictions: h, flags, the 2x2 St blocks), packed into
// ONE buffer -- host-mapped pinned memory, written straight from the device -- so that the read-back is one launch and
// one synchronisation instead of three staged copies to pageable memory.  Layout: [N][2] h, [N][4] Sd (optional), N flag bytes.
template <typename T>
__global__ void k_pack_predictions(const T* __restrict__ h, const T* __restrict__ Sd, const unsigned char* __restrict__ flags,
                                   int N, T* __restrict__ out_h, T* __restrict__ out_sd, unsigned char* __restrict__ out_fl) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 2 * N) out_h[t] = h[t];
  if (Sd && t < 4 * N) out_sd[t] = Sd[t];
  if (t < N) out_fl[t] = flags[t];
}

// 2x2 diagonal block of St per feature: Hrow P Hrow^T + r_pix I with P = Sigma[idx, idx],
// idx = [0..6, p..p+fs).  Only host-facing consumers need it (the gate of Patch::findMatch, the search
// ellipses, the RANSAC hypotheses), so it is evaluated on demand, 16 lanes per feature: lane a owns row
// a of P (13 + 3 idle), and the 2x2 sums are reduced with shuffles.
template <typename T>
__global__ void k_measure_sd(const T* __restrict__ S, int ld, const int* __restrict__ pos,
                             const int* __restrict__ coding, int f_begin, int N, T r_pix,
                             const T* __restrict__ Hc, const T* __restrict__ Hf, T* __restrict__ Sd,
                             const int* __restrict__ list = nullptr, int count = 0,
                             const T* __restrict__ h = nullptr, const unsigned char* __restrict__ flags = nullptr,
                             T* __restrict__ out_h = nullptr, T* __restrict__ out_sd = nullptr,
                             unsigned char* __restrict__ out_fl = nullptr) {
  // out_h / out_sd / out_fl (ekf_get_predictions): the read-back buffer of k_pack_predictions filled in the same launch
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int i = f_begin + (gid >> 4);
  const int a = gid & 15;
  bool live = i < N;
  if (list) {
    live = (gid >> 4) < count;
    i = live ? list[gid >> 4] : f_begin;
  }
  const int fi = live ? i : f_begin;
  const int p = pos[fi];
  const int fs = coding[fi] ? 3 : 6;
  const T* hc = Hc + (size_t)fi * 14;
  const T* hf = Hf + (size_t)fi * 12;
  T acc0 = T(0), acc1 = T(0);
  T h0 = T(0), h1 = T(0);                    // Hrow[0][a], Hrow[1][a]
  if (live && a < 7 + fs) {
    const int ra = (a < 7) ? a : p + (a - 7);
    const T* row = S + (size_t)ra * ld;
    for (int b = 0; b < 7; ++b) { const T v = row[b]; acc0 += v * hc[b]; acc1 += v * hc[7 + b]; }
    for (int b = 0; b < fs; ++b) { const T v = row[p + b]; acc0 += v * hf[b]; acc1 += v * hf[6 + b]; }
    h0 = (a < 7) ? hc[a] : hf[a - 7];
    h1 = (a < 7) ? hc[7 + a] : hf[6 + a - 7];
  }
  T s00 = h0 * acc0, s01 = h0 * acc1, s10 = h1 * acc0, s11 = h1 * acc1;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    s00 += __shfl_down(s00, off, 16); s01 += __shfl_down(s01, off, 16);
    s10 += __shfl_down(s10, off, 16); s11 += __shfl_down(s11, off, 16);
  }
  if (live && a == 0) {
    Sd[(size_t)i * 4 + 0] = s00 + r_pix;
    Sd[(size_t)i * 4 + 1] = s01;
    Sd[(size_t)i * 4 + 2] = s10;
    Sd[(size_t)i * 4 + 3] = s11 + r_pix;
    if (out_sd) {
      out_sd[(size_t)i * 4 + 0] = s00 + r_pix;
      out_sd[(size_t)i * 4 + 1] = s01;
      out_sd[(size_t)i * 4 + 2] = s10;
      out_sd[(size_t)i * 4 + 3] = s11 + r_pix;
      out_h[2 * i] = h[2 * i];
      out_h[2 * i + 1] = h[2 * i + 1];
      out_fl[i] = flags[i];
    }
  }
}

// ---------------------------------------------------------------------------------------
// a11: quaternion normalisation: mu[3:7] /= |q|, Qn = (|q|^2 I - q q^T)/|q|^3 -> scratch.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void k_normalize_quat(T* __restrict__ mu, T* __restrict__ scr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const T q[4] = {mu[3], mu[4], mu[5], mu[6]};
  const T nn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const T norma = t_sqrt(nn);
  const T inv3 = T(1) / (norma * norma * norma);
  for (int i = 0; i < 4; ++i) mu[3 + i] = q[i] / norma;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      scr[SCR_QN + i * 4 + j] = ((i == j ? norma * norma : T(0)) - q[i] * q[j]) * inv3;
}

// ---------------------------------------------------------------------------------------
// a12 add feature, step 1 (one lane): new 6-vector into mu[n..n+6), G = d f/d[r,q] (6x7),
// corner C = G Scc G^T + s_pix2 Jp Jp^T + e6 e6^T sigma_rho_0 (6x6).
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void k_add_prepare(T* __restrict__ mu, const T* __restrict__ S, int ld, int n,
                              CamParams cam, T u, T v, T rho0, T s_pix2, T sigma_rho0,
                              T* __restrict__ scr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const T q[4] = {mu[3], mu[4], mu[5], mu[6]};
  T hC[3], Jn[4];
  undistort_deproject(cam, u, v, hC, Jn);
  T R[9];
  quat2rot(q, R);
  T hW[3];
  mat3_vec(R, hC, hW);
  const T hx = hW[0], hy = hW[1], hz = hW[2];
  const T theta = t_atan2(hx, hz);
  const T phi = t_atan2(-hy, t_sqrt(hx * hx + hz * hz));
  mu[n + 0] = mu[0]; mu[n + 1] = mu[1]; mu[n + 2] = mu[2];
  mu[n + 3] = theta; mu[n + 4] = phi; mu[n + 5] = rho0;
  // J_f_hW rows 3 (theta) and 4 (phi)                             (vR.cpp:1599-1623)
  const T normal = hx * hx + hz * hz, normal2 = normal + hy * hy, sn = t_sqrt(normal);
  const T jt[3] = {hz / normal, T(0), -hx / normal};
  const T jp[3] = {hx * hy / sn / normal2, -sn / normal2, hz * hy / sn / normal2};
  T Jq[12];
  drot_dq_times(q, hC, Jq);                                          // d(R(q) hC)/dq
  T G[42];
  for (int i = 0; i < 42; ++i) G[i] = T(0);
  G[0 * 7 + 0] = T(1); G[1 * 7 + 1] = T(1); G[2 * 7 + 2] = T(1);
  for (int c = 0; c < 4; ++c) {
    G[3 * 7 + 3 + c] = jt[0] * Jq[0 * 4 + c] + jt[1] * Jq[1 * 4 + c] + jt[2] * Jq[2 * 4 + c];
    G[4 * 7 + 3 + c] = jp[0] * Jq[0 * 4 + c] + jp[1] * Jq[1 * 4 + c] + jp[2] * Jq[2 * 4 + c];
  }
  // Jpix = J_f_hW * R * J_undist (6x2); J_undist = [Jn; 0 0]
  T RJ[6];                                                           // R * J_undist (3x2)
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 2; ++c) RJ[r * 2 + c] = R[r * 3 + 0] * Jn[0 * 2 + c] + R[r * 3 + 1] * Jn[1 * 2 + c];
  T Jx[12];
  for (int i = 0; i < 12; ++i) Jx[i] = T(0);
  for (int c = 0; c < 2; ++c) {
    Jx[3 * 2 + c] = jt[0] * RJ[0 * 2 + c] + jt[1] * RJ[1 * 2 + c] + jt[2] * RJ[2 * 2 + c];
    Jx[4 * 2 + c] = jp[0] * RJ[0 * 2 + c] + jp[1] * RJ[1 * 2 + c] + jp[2] * RJ[2 * 2 + c];
  }
  for (int i = 0; i < 42; ++i) scr[SCR_G + i] = G[i];
  // corner
  T GS[42];                                                          // G * Scc (6x7)
  for (int a = 0; a < 6; ++a)
    for (int c = 0; c < 7; ++c) {
      T acc = T(0);
      for (int k = 0; k < 7; ++k) acc += G[a * 7 + k] * S[(size_t)k * ld + c];
      GS[a * 7 + c] = acc;
    }
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b <= a; ++b) {                                     // lower triangle, mirrored: exactly symmetric
      T acc = T(0);
      for (int k = 0; k < 7; ++k) acc += GS[a * 7 + k] * G[b * 7 + k];
      acc += s_pix2 * (Jx[a * 2 + 0] * Jx[b * 2 + 0] + Jx[a * 2 + 1] * Jx[b * 2 + 1]);
      if (a == 5 && b == 5) acc += sigma_rho0;                       // unsquared, vR.cpp:365
      scr[SCR_C + a * 6 + b] = acc;
      scr[SCR_C + b * 6 + a] = acc;
    }
}

// a12 step 2: border rows/cols.  lane j < n: Sigma[n+a, j] = sum_t G[a,t] Sigma[t, j],
// Sigma[j, n+a] = sum_t Sigma[j, t] G[a,t]; block 0 writes the 6x6 corner.
template <typename T>
__global__ void k_add_border(T* __restrict__ S, int ld, int n, const T* __restrict__ scr) {
  __shared__ T sG[42];
  for (int i = threadIdx.x; i < 42; i += blockDim.x) sG[i] = scr[SCR_G + i];
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    T col[7], row[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) { col[t] = S[(size_t)t * ld + j]; row[t] = S[(size_t)j * ld + t]; }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      T accr = T(0), accc = T(0);
#pragma unroll
      for (int t = 0; t < 7; ++t) { accr = t_fma(sG[a * 7 + t], col[t], accr); accc = t_fma(row[t], sG[a * 7 + t], accc); }
      S[(size_t)(n + a) * ld + j] = accr;
      S[(size_t)j * ld + n + a] = accc;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 36)
    S[(size_t)(n + threadIdx.x / 6) * ld + n + threadIdx.x % 6] = scr[SCR_C + threadIdx.x];
}

// ---------------------------------------------------------------------------------------
// a13 / a14: remove + convert in one out-of-place pass  dst = J src J^T.
// Each NEW index i' has a descriptor: src[i'] = first old index, conv[i'] = -1 for a
// pass-through, else (converted feature slot * 3 + component e): the new entry is
// sum_b Jy[slot][e][b] * old[src + b], b < 6.  Lanes sweep j' (coalesced), blockIdx.y = i'.
// ---------------------------------------------------------------------------------------
constexpr int kCompactRows = 32;                 // rows of the new matrix per workgroup of k_compact_transform
template <typename T>
__global__ void k_compact_transform(const T* __restrict__ src, T* __restrict__ dst, int ld,
                                    int n_new, const int* __restrict__ map_src,
                                    const int* __restrict__ map_conv, const T* __restrict__ Jy) {
  // Round 5: a lane takes FOUR consecutive output columns.  Where they are a pass-through run of the source row (the common
  // case: a removal deletes 6 or 3 columns here and there, everything between them is a shifted copy) the lane moves 16
  // bytes per instruction -- a dword-aligned 16-byte load (the shift is a multiple of 3 floats), a 16-byte aligned store --
  // instead of four 4-byte round trips: the pass is an HBM stream (2 n^2 s).
  // Round 6: the column descriptors of a lane do not depend on the row, so a workgroup keeps its 1024 columns and walks
  // kCompactRows rows with them -- one load and one store per 16 bytes moved where round 5 issued two descriptor loads
  // beside them (0.37-0.39 of the HBM peak, bound by load issue) --, four rows in flight per lane.
  typedef T vec4u __attribute__((ext_vector_type(4), aligned(sizeof(T))));
  typedef T vec4a __attribute__((ext_vector_type(4)));
  typedef int int4a __attribute__((ext_vector_type(4)));
  const int jq = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int row0 = blockIdx.y * kCompactRows, row1 = min(row0 + kCompactRows, n_new);
  bool run = false;                               // the lane's four columns are a pass-through run of the source row
  int ms0 = 0;
  if (jq + 3 < n_new) {
    const int4a ms = *reinterpret_cast<const int4a*>(map_src + jq);
    const int4a mc = *reinterpret_cast<const int4a*>(map_conv + jq);
    run = (mc[0] & mc[1] & mc[2] & mc[3]) < 0 && ms[3] - ms[0] == 3;
    ms0 = ms[0];
  }
  for (int ipb = row0; ipb < row1; ipb += 8) {
    int si[8], ci[8];
    bool plain = true;                            // (uniform) none of the four rows is a converted entry
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int ip = min(ipb + u, n_new - 1);
      si[u] = map_src[ip];
      ci[u] = map_conv[ip];
      plain = plain && ci[u] < 0;
    }
    if (jq >= n_new) continue;
    if (run && plain) {
      vec4u v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const vec4u*>(src + (size_t)si[u] * ld + ms0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (ipb + u < row1) *reinterpret_cast<vec4a*>(dst + (size_t)(ipb + u) * ld + jq) = vec4a{v[u][0], v[u][1], v[u][2], v[u][3]};
      continue;
    }
#pragma unroll 1
    for (int u = 0; u < 8; ++u) {
      const int ip = ipb + u;
      if (ip >= row1) break;
      const int si0 = si[u], ci0 = ci[u];
      if (run && ci0 < 0) {
        const vec4u v = *reinterpret_cast<const vec4u*>(src + (size_t)si0 * ld + ms0);
        *reinterpret_cast<vec4a*>(dst + (size_t)ip * ld + jq) = vec4a{v[0], v[1], v[2], v[3]};
        continue;
      }
#pragma unroll 1
      for (int jp = jq; jp < min(jq + 4, n_new); ++jp) {
        const int sj = map_src[jp];
        const int cj = map_conv[jp];
        T acc;
        if (ci0 < 0 && cj < 0) {
          acc = src[(size_t)si0 * ld + sj];
        } else if (ci0 < 0) {
          acc = T(0);
          for (int b = 0; b < 6; ++b) acc = t_fma(src[(size_t)si0 * ld + sj + b], Jy[cj * 6 + b], acc);
        } else if (cj < 0) {
          acc = T(0);
          for (int a = 0; a < 6; ++a) acc = t_fma(Jy[ci0 * 6 + a], src[(size_t)(si0 + a) * ld + sj], acc);
        } else if (jp <= ip) {
          // 3 x 3 block of two converted entries, lower triangle: sum_a Jy_i[a] (sum_b S[i+a][j+b] Jy_j[b])
          acc = T(0);
          for (int a = 0; a < 6; ++a) {
            T inner = T(0);
            for (int b = 0; b < 6; ++b) inner += src[(size_t)(si0 + a) * ld + sj + b] * Jy[cj * 6 + b];
            acc += Jy[ci0 * 6 + a] * inner;
          }
        } else {
          // ... upper triangle: Jy (S Jy^T) is symmetric only up to rounding, so the element is evaluated in the ORDER of its
          // mirror element -- sum_b Jy_j[b] (sum_a S[i+a][j+b] Jy_i[a]), on an exactly symmetric source the same products in
          // the same sequence -- but from the rows of THIS entry's feature: exactly symmetric, and under row-panel sharding
          // only rows the rank owns are read (round 2 swapped the roles and read the mirror feature's rows, which another
          // rank may own: found by tools/rccl_smoke.py in round 3)
          acc = T(0);
          for (int b = 0; b < 6; ++b) {
            T inner = T(0);
            for (int a = 0; a < 6; ++a) inner += src[(size_t)(si0 + a) * ld + sj + b] * Jy[ci0 * 6 + a];
            acc += Jy[cj * 6 + b] * inner;
          }
        }
        dst[(size_t)ip * ld + jp] = acc;
      }
    }
  }
}

// mu side of the same pass: pass-through copies or y = a + m / rho.
template <typename T>
__global__ void k_compact_mu(const T* __restrict__ src, T* __restrict__ dst, int n_new,
                             const int* __restrict__ map_src, const int* __restrict__ map_conv,
                             const T* __restrict__ Yxyz) {
  const int ip = blockIdx.x * blockDim.x + threadIdx.x;
  if (ip >= n_new) return;
  const int ci = map_conv[ip];
  dst[ip] = (ci < 0) ? src[map_src[ip]] : Yxyz[ci];
}

// a14 linearity test per inverse-depth feature (vR.cpp:713-721) + the 3x6 Jacobian and the
// XYZ point of every feature that passes.  out_flag[i] = 1 -> convert.
template <typename T>
__global__ void k_linearity(const T* __restrict__ mu, const T* __restrict__ S, int ld,
                            const int* __restrict__ pos, const int* __restrict__ coding, int N,
                            unsigned char* __restrict__ out_flag, T* __restrict__ Jy, T* __restrict__ Yxyz,
                            int force_jacobian) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (coding[i] != 0) { out_flag[i] = 0; return; }
  const int p = pos[i];
  const T theta = mu[p + 3], phi = mu[p + 4], ro = mu[p + 5];
  const T st = t_sin(theta), ct = t_cos(theta), sp = t_sin(phi), cp = t_cos(phi);
  const T m[3] = {st * cp, -sp, ct * cp};
  T y[3], d[3];
  for (int k = 0; k < 3; ++k) { y[k] = mu[p + k] + m[k] / ro; d[k] = y[k] - mu[k]; }
  const T sigma_rho = S[(size_t)(p + 5) * ld + p + 5];            // variance, vR.cpp:717
  const T tt = d[0] * m[0] + d[1] * m[1] + d[2] * m[2];
  const T Ld = T(4) * sigma_rho * t_abs(tt) / (ro * ro * (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
  const bool conv = force_jacobian || (Ld < T(0.01));
  out_flag[i] = conv ? 1 : 0;
  T* J = Jy + (size_t)i * 18;
  for (int k = 0; k < 18; ++k) J[k] = T(0);
  J[0] = T(1); J[7] = T(1); J[14] = T(1);
  J[3] = ct * cp / ro;   J[9] = T(0);      J[15] = -st * cp / ro;
  J[4] = -st * sp / ro;  J[10] = -cp / ro; J[16] = -ct * sp / ro;
  J[5] = -m[0] / (ro * ro); J[11] = -m[1] / (ro * ro); J[17] = -m[2] / (ro * ro);
  for (int k = 0; k < 3; ++k) Yxyz[(size_t)i * 3 + k] = y[k];
}

// f3: world point + 3x3 covariance of one feature (map export).
template <typename T>
__global__ void k_feature_xyz(const T* __restrict__ mu, const T* __restrict__ S, int ld,
                              int p, int is_xyz, T* __restrict__ out /*3 + 9*/) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (is_xyz) {
    for (int k = 0; k < 3; ++k) out[k] = mu[p + k];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) out[3 + a * 3 + b] = S[(size_t)(p + a) * ld + p + b];
    return;
  }
  const T theta = mu[p + 3], phi = mu[p + 4], ro = mu[p + 5];
  const T st = t_sin(theta), ct = t_cos(theta), sp = t_sin(phi), cp = t_cos(phi);
  const T m[3] = {st * cp, -sp, ct * cp};
  T J[18];
  for (int k = 0; k < 18; ++k) J[k] = T(0);
  J[0] = T(1); J[7] = T(1); J[14] = T(1);
  J[3] = ct * cp / ro;   J[9] = T(0);      J[15] = -st * cp / ro;
  J[4] = -st * sp / ro;  J[10] = -cp / ro; J[16] = -ct * sp / ro;
  J[5] = -m[0] / (ro * ro); J[11] = -m[1] / (ro * ro); J[17] = -m[2] / (ro * ro);
  for (int k = 0; k < 3; ++k) out[k] = mu[p + k] + m[k] / ro;
  T JS[18];
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 6; ++c) {
      T acc = T(0);
      for (int k = 0; k < 6; ++k) acc += J[a * 6 + k] * S[(size_t)(p + k) * ld + p + c];
      JS[a * 6 + c] = acc;
    }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      T acc = T(0);
      for (int k = 0; k < 6; ++k) acc += JS[a * 6 + k] * J[b * 6 + k];
      out[3 + a * 3 + b] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// f2: search-ellipse parameters of every feature from its 2x2 St block
// (computeEllipsoidParameters, vR.cpp:1368-1382): semi-axes (int)(sigma_size*sqrt(eig)) (1 when the
// eigenvalue is not positive), angle (int)(180/3.14*atan2(v1, v0)) of the eigenvector of the SMALLER
// eigenvalue, taken with v0 >= 0 (an eigenvector's sign is arbitrary; the ellipse is the same).
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void k_search_ellipses(const T* __restrict__ Sd, int N, int sigma_size, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double a = double(Sd[4 * i]), b = 0.5 * (double(Sd[4 * i + 1]) + double(Sd[4 * i + 2])), c = double(Sd[4 * i + 3]);
  const double mean = 0.5 * (a + c), dif = 0.5 * (a - c);
  const double rad = sqrt(dif * dif + b * b);
  const double e0 = mean - rad, e1 = mean + rad;
  // eigenvector of e0: (b, e0 - a) or (e0 - c, b), whichever is better conditioned
  double vx, vy;
  if (fabs(e0 - a) > fabs(e0 - c)) { vx = b; vy = e0 - a; } else { vx = e0 - c; vy = b; }
  if (vx == 0.0 && vy == 0.0) { vx = 1.0; vy = 0.0; }       // isotropic block
  if (vx < 0.0 || (vx == 0.0 && vy < 0.0)) { vx = -vx; vy = -vy; }
  out[3 * i + 0] = (e0 > 0.0) ? (int)(sigma_size * sqrt(e0)) : 1;
  out[3 * i + 1] = (e1 > 0.0) ? (int)(sigma_size * sqrt(e1)) : 1;
  out[3 * i + 2] = (int)(180.0 / 3.14 * atan2(vy, vx));
}

// ---------------------------------------------------------------------------------------
// f1: 1-point RANSAC hypotheses (vR.cpp:986-1034), all of them at once.  Hypothesis k = measured
// feature midx[k]: S_k = its 2x2 St block, K_k = Sigma H_k^T S_k^-1 = W[:, 2k:2k+2] S_k^-1,
// mu_k = mu + K_k (z_k - h_k); every measured feature j is re-projected with mu_k (camera
// quaternion re-normalised, :999) and is an inlier when |z_j - h_j(mu_k)| <= thr.
// Lane = hypothesis k (coalesced along the columns of W), blockIdx.y = feature j.
// Quirk kept: an XYZ feature is re-projected from the un-updated mu (:1016).
// mask[j * M + k] = 1 / 0.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void k_ransac_eval(const T* __restrict__ mu, const T* __restrict__ W, int ldy,
                              const T* __restrict__ Sd, const T* __restrict__ h, const T* __restrict__ z,
                              const int* __restrict__ pos, const int* __restrict__ coding,
                              const int* __restrict__ midx, int M, CamParams cam, T thr,
                              unsigned char* __restrict__ mask, int j0 = 0) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = j0 + blockIdx.y;                 // j0 > 0: a rank evaluates the listed features it owns (their rows of W)
  if (k >= M) return;
  const int fk = midx[k], fj = midx[j];
  // g = S_k^-1 nu_k
  const T s00 = Sd[4 * fk], s01 = Sd[4 * fk + 1], s10 = Sd[4 * fk + 2], s11 = Sd[4 * fk + 3];
  const T det = s00 * s11 - s01 * s10;
  const T n0 = z[2 * k] - h[2 * fk], n1 = z[2 * k + 1] - h[2 * fk + 1];
  const T g0 = (s11 * n0 - s01 * n1) / det, g1 = (-s10 * n0 + s00 * n1) / det;
  T c[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) c[t] = mu[t] + W[(size_t)t * ldy + 2 * k] * g0 + W[(size_t)t * ldy + 2 * k + 1] * g1;
  const T qn = t_sqrt(c[3] * c[3] + c[4] * c[4] + c[5] * c[5] + c[6] * c[6]);
  const T qc[4] = {c[3] / qn, -c[4] / qn, -c[5] / qn, -c[6] / qn};
  T R[9];
  quat2rot(qc, R);
  const int p = pos[fj];
  T d[3];
  if (coding[fj] == 0) {
    T f[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
      f[t] = mu[p + t] + W[(size_t)(p + t) * ldy + 2 * k] * g0 + W[(size_t)(p + t) * ldy + 2 * k + 1] * g1;
    const T st = t_sin(f[3]), ct = t_cos(f[3]), sp = t_sin(f[4]), cp = t_cos(f[4]);
    d[0] = f[5] * (f[0] - c[0]) + st * cp;
    d[1] = f[5] * (f[1] - c[1]) - sp;
    d[2] = f[5] * (f[2] - c[2]) + ct * cp;
  } else {
    for (int t = 0; t < 3; ++t) d[t] = mu[p + t] - c[t];
  }
  T hC[3], hd[2], Jp[6];
  mat3_vec(R, d, hC);
  project_distort(cam, hC, hd, Jp);
  const T e0 = z[2 * j] - hd[0], e1 = z[2 * j + 1] - hd[1];
  mask[(size_t)j * M + k] = (t_sqrt(e0 * e0 + e1 * e1) <= thr) ? 1 : 0;
}

__global__ void k_ransac_count(const unsigned char* __restrict__ mask, int M, int* __restrict__ counts, int j0 = 0,
                               int j1 = -1) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  if (j1 < 0) j1 = M;
  int c = 0;
  for (int j = j0; j < j1; ++j) c += mask[(size_t)j * M + k];    // (a rank: the features it owns -> partial counts)
  counts[k] = c;
}

// What the host needs back after the hypotheses are evaluated (ekf_ransac_1point / ekf_update_two_stage), in ONE
// host-mapped pinned buffer: the counts, the camera pose (r, q: the "state before the first update" the rescue
// linearises around) and -- while it is small -- the whole inlier mask, so that picking a hypothesis' column needs no
// second round trip.  Layout: [M] int counts, [8] T (7 used), then M x M bytes (with_mask).
template <typename T>
__global__ void k_pack_ransac(const int* __restrict__ counts, int M, const T* __restrict__ mu,
                              const unsigned char* __restrict__ mask, int with_mask, int* __restrict__ out_counts,
                              T* __restrict__ out_cam, unsigned char* __restrict__ out_mask) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < M) out_counts[t] = counts[t];
  if (t < 7) out_cam[t] = mu[t];
  if (with_mask)
    for (size_t q = t; q < (size_t)M * M; q += (size_t)gridDim.x * blockDim.x) out_mask[q] = mask[q];
}

// f3: map export, one lane per feature: the N x 12 table of RosVSLAM::getPointsFeatures
// (RosVSLAMRansac.cpp:340-418): [X Y Z] * map_scale, then the 3x3 covariance block row by row.
// The reference fills rows of XYZ features only (inverse-depth rows stay zero, :360-375); with
// `convert_inverse_depth` those rows get inverseDepth2XyzWorld(f) and Jf Sigma_ff Jf^T instead
// (what its marker code does, :171-183).
// row_of != nullptr: feature i goes to row row_of[i] of a table of `rows` rows (Patch::real_index order,
// RosVSLAMRansac.cpp:361, 388); rows outside the table are skipped.
template <typename T>
__global__ void k_export_points(const T* __restrict__ mu, const T* __restrict__ S, int ld,
                                const int* __restrict__ pos, const int* __restrict__ coding, int N,
                                T map_scale, int convert_inverse_depth, T* __restrict__ out,
                                const int* __restrict__ row_of = nullptr, int rows = 0,
                                const T* __restrict__ scale_ptr = nullptr) {
  if (scale_ptr) map_scale = *scale_ptr;               // the scale state read where it lives: no host round trip for it
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int row = row_of ? row_of[i] : i;
  if (row_of && (row < 0 || row >= rows)) return;
  T* o = out + (size_t)row * 12;
  const int p = pos[i];
  if (coding[i] != 0) {
    for (int k = 0; k < 3; ++k) o[k] = mu[p + k] * map_scale;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) o[3 + a * 3 + b] = S[(size_t)(p + a) * ld + p + b];
    return;
  }
  if (!convert_inverse_depth) {
    for (int k = 0; k < 12; ++k) o[k] = T(0);
    return;
  }
  const T theta = mu[p + 3], phi = mu[p + 4], ro = mu[p + 5];
  const T st = t_sin(theta), ct = t_cos(theta), sp = t_sin(phi), cp = t_cos(phi);
  const T m[3] = {st * cp, -sp, ct * cp};
  T J[18];
  for (int k = 0; k < 18; ++k) J[k] = T(0);
  J[0] = T(1); J[7] = T(1); J[14] = T(1);
  J[3] = ct * cp / ro;   J[9] = T(0);      J[15] = -st * cp / ro;
  J[4] = -st * sp / ro;  J[10] = -cp / ro; J[16] = -ct * sp / ro;
  J[5] = -m[0] / (ro * ro); J[11] = -m[1] / (ro * ro); J[17] = -m[2] / (ro * ro);
  for (int k = 0; k < 3; ++k) o[k] = (mu[p + k] + m[k] / ro) * map_scale;
  T JS[18];
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 6; ++c) {
      T acc = T(0);
      for (int k = 0; k < 6; ++k) acc += J[a * 6 + k] * S[(size_t)(p + k) * ld + p + c];
      JS[a * 6 + c] = acc;
    }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      T acc = T(0);
      for (int k = 0; k < 6; ++k) acc += JS[a * 6 + k] * J[b * 6 + k];
      o[3 + a * 3 + b] = acc;
    }
}

// removeFeature's "segment to save good features" (vR.cpp:394-404): XYZ_pos = mu[pos:pos+3] and cov_4_delete = the 3x3
// block of Sigma row by row, of the listed state positions, captured BEFORE the compaction pass drops them.
template <typename T>
__global__ void k_archive_points(const T* __restrict__ mu, const T* __restrict__ S, int ld,
                                 const int* __restrict__ state_pos, int count, T* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / 12, e = t % 12;
  if (i >= count) return;
  const int p = state_pos[i];
  out[(size_t)i * 12 + e] = (e < 3) ? mu[p + e] : S[(size_t)(p + (e - 3) / 3) * ld + p + (e - 3) % 3];
}

// The archived patches written over their rows of the points table (RosVSLAMRansac.cpp:406-414): XYZ_pos * map_scale
// and cov_4_delete.
template <typename T>
__global__ void k_export_archived(const T* __restrict__ arch, const int* __restrict__ row_of, int count, T map_scale_,
                                  T* __restrict__ out, int rows, const T* __restrict__ scale_ptr = nullptr) {
  const T map_scale = scale_ptr ? *scale_ptr : map_scale_;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / 12, e = t % 12;
  if (i >= count) return;
  const int row = row_of[i];
  if (row < 0 || row >= rows) return;
  const T v = arch[(size_t)i * 12 + e];
  out[(size_t)row * 12 + e] = (e < 3) ? v * map_scale : v;
}

// High-innovation gate (vR.cpp:1113-1114): (h - z)^T S_hi^-1 (h - z) <= thr with S_hi = H Sigma H^T
// (no measurement noise), one lane per listed feature.
template <typename T>
__global__ void k_chi2_gate(const T* __restrict__ h, const T* __restrict__ Sd, const T* __restrict__ z,
                            const int* __restrict__ list, int count, T thr, unsigned char* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const int i = list[k];
  const T e0 = h[2 * i] - z[2 * k], e1 = h[2 * i + 1] - z[2 * k + 1];
  const T s00 = Sd[4 * i], s01 = Sd[4 * i + 1], s10 = Sd[4 * i + 2], s11 = Sd[4 * i + 3];
  const T det = s00 * s11 - s01 * s10;
  const T q = (e0 * (s11 * e0 - s01 * e1) + e1 * (-s10 * e0 + s00 * e1)) / det;
  out[k] = (q <= thr) ? 1 : 0;
}

// Invariants of the device covariance (test / debug entry ekf_check_invariants): out[0] = max |S[i][j]| outside the
// live n x n within the n_pad x ld buffer (must be 0: the tile kernels carry no edge guards), out[1] = max
// |S[i][j] - S[j][i]| over the live block, out[2] = max |S[i][j]| over it.  Non-negative floats order like their
// bit patterns, so the three maxima are integer atomicMax on the float bits of the (double -> float) values.
template <typename T>
__global__ void k_check_invariants(const T* __restrict__ S, int ld, int n, int n_pad, unsigned int* __restrict__ out) {
  const int i = blockIdx.y;
  float pad = 0.f, asym = 0.f, big = 0.f;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ld; j += gridDim.x * blockDim.x) {
    const T v = S[(size_t)i * ld + j];
    if (i >= n || j >= n) {
      pad = fmaxf(pad, fabsf(float(v)));
    } else {
      big = fmaxf(big, fabsf(float(v)));
      if (j < i) asym = fmaxf(asym, fabsf(float(v - S[(size_t)j * ld + i])));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    pad = fmaxf(pad, __shfl_down(pad, off, 64));
    asym = fmaxf(asym, __shfl_down(asym, off, 64));
    big = fmaxf(big, __shfl_down(big, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    if (pad > 0.f) atomicMax(out + 0, __float_as_uint(pad));
    if (asym > 0.f) atomicMax(out + 1, __float_as_uint(asym));
    if (big > 0.f) atomicMax(out + 2, __float_as_uint(big));
  }
}

// EKF_OPT_FEATURE_NOISE (opt-in): Sigma[i][i] += delta for the feature states (i >= camera_dim), once per predict.
template <typename T>
__global__ void k_inflate_diagonal(T* __restrict__ S, int ld, int first, int n, T delta) {
  const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) S[(size_t)i * ld + i] += delta;
}

template <typename T>
__global__ void k_fill(T* __restrict__ p, size_t count, T v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}

}  // namespace ekf
