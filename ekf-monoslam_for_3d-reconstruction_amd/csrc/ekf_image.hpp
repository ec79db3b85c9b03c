// Image side of the filter: patch templates, predicted motion blur, NCC search.
//   Patch::Patch template capture           vslamRansac.cpp:318, Patch.cpp:76-105
//   predicted blur of the template          vslamRansac.cpp:496-500, 546-548, 575-576; libblur.cpp:17-79
//   Patch::findMatch + computeCorrelation   Patch.cpp:215-329
// One workgroup per feature; the frame is an 8-bit single-channel image resident on the device.
// The image arithmetic of the reference is float / double regardless of the filter's scalar type:
// the kernels that restate it switch FMA contraction off (the reference is built with -msse4: no FMA)
// and spell the float / double operation order out.
#pragma once
#include <hip/hip_runtime.h>
#include "ekf_math.hpp"

namespace ekf {

constexpr int kMaxWindow = 32;        // window_size (template edge), reference configs: 15, 21, 30
constexpr int kMaxSearch = 20;        // findMatch clamps the half search range (Patch.cpp:240-242)
constexpr int kMaxTaps = 1024;        // longest blur line (pixels) the kernel builder accepts

// window x window pixels of the frame at (x0, y0) -> template slots (vR.cpp:318: cv::Rect(pf.x - w/2, ...)).
__global__ void k_capture_patch(const unsigned char* __restrict__ frame, int fw, int fh, int x0, int y0, int w,
                                unsigned char* __restrict__ patch, unsigned char* __restrict__ mpatch) {
  for (int t = threadIdx.x; t < w * w; t += blockDim.x) {
    const int yy = min(max(y0 + t / w, 0), fh - 1), xx = min(max(x0 + t % w, 0), fw - 1);
    const unsigned char v = frame[(size_t)yy * fw + xx];
    patch[t] = v;
    mpatch[t] = v;
  }
}

// dst[k] = src[keep[k]] for the template stores after removeFeature (vR.cpp:373-421 erases patches[index]).
__global__ void k_gather_patches(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                 const int* __restrict__ keep, int w2) {
  const int k = blockIdx.x;
  const unsigned char* s = src + (size_t)keep[k] * w2;
  unsigned char* d = dst + (size_t)k * w2;
  for (int t = threadIdx.x; t < w2; t += blockDim.x) d[t] = s[t];
}

// Prediction of every feature at the blur pose r + v T_camera dT, q (x) quat(w T_camera dT) (vR.cpp:496-500, 546, 575).
template <typename T>
__global__ void k_blur_points(const T* __restrict__ mu, const int* __restrict__ pos, const int* __restrict__ coding,
                              int N, CamParams cam, T tcam, T dT, T* __restrict__ hb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const T q[4] = {mu[3], mu[4], mu[5], mu[6]};
  T wv[3], rb[3];
  for (int k = 0; k < 3; ++k) {
    wv[k] = mu[10 + k] * tcam * dT;
    rb[k] = mu[k] + mu[7 + k] * tcam * dT;
  }
  T hq[4], qb[4];
  vec2quat(wv, hq);
  quat_mul(q, hq, qb);
  const T qbc[4] = {qb[0], -qb[1], -qb[2], -qb[3]};
  T R[9];
  quat2rot(qbc, R);
  const int p = pos[i];
  T d[3];
  if (coding[i] == 0) {
    const T theta = mu[p + 3], phi = mu[p + 4], ro = mu[p + 5];
    const T st = t_sin(theta), ct = t_cos(theta), sp = t_sin(phi), cp = t_cos(phi);
    const T m[3] = {st * cp, -sp, ct * cp};
    for (int k = 0; k < 3; ++k) d[k] = ro * (mu[p + k] - rb[k]) + m[k];
  } else {
    for (int k = 0; k < 3; ++k) d[k] = mu[p + k] - rb[k];
  }
  T hC[3], hd[2], Jp[6];
  mat3_vec(R, d, hC);
  project_distort(cam, hC, hd, Jp);
  hb[2 * i] = hd[0];
  hb[2 * i + 1] = hd[1];
}

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

__device__ __forceinline__ unsigned char saturate_u8(double v) {     // cv::saturate_cast<uchar>(double) = cvRound + clamp
  const double r = rint(v);                                           // round half to even
  return (unsigned char)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
}

// Patch::blur (Patch.cpp:50-57): matching template = template filtered with the line kernel of
// evaluateKernel (libblur.cpp:17-52) when the predicted motion is longer than kernel_size, else a copy.
// filter2D (libblur.cpp:73): correlation in double, anchor = kernel centre, BORDER_REFLECT_101, then CV_8U.
template <typename T>
__global__ void __launch_bounds__(256)
k_blur_templates(const T* __restrict__ h, const T* __restrict__ hb, const unsigned char* __restrict__ flags, int w,
                 int kernel_min_size, const unsigned char* __restrict__ patch, unsigned char* __restrict__ mpatch) {
#pragma clang fp contract(off)     // the reference's arithmetic has no fused multiply-adds (and HIP's __fmul_rn-style intrinsics are inlined plain operators that fuse anyway)
  const int i = blockIdx.x;
  if (!(flags[i] & 1)) return;                              // blur() is only reached by visible features
  __shared__ unsigned char src[kMaxWindow * kMaxWindow];
  __shared__ int taps[kMaxTaps];                            // (row << 16) | col, distinct, sorted
  __shared__ int ntaps, krows, kcols, do_blur;
  const int w2 = w * w;
  const unsigned char* P = patch + (size_t)i * w2;
  unsigned char* M = mpatch + (size_t)i * w2;
  for (int t = threadIdx.x; t < w2; t += blockDim.x) src[t] = P[t];
  if (threadIdx.x == 0) {
    const float ox = (float)h[2 * i], oy = (float)h[2 * i + 1];          // cv::Point2f one, two
    const float tx = (float)hb[2 * i], ty = (float)hb[2 * i + 1];
    const float dx = ((ox) - (tx)), dy = ((oy) - (ty));
    const float nrm = sqrtf(((((dx) * (dx))) + (((dy) * (dy)))));   // Eigen (p1 - p2).norm()
    int blur = nrm > (float)kernel_min_size;
    int cnt = 0, rows = 1, cols = 1;
    if (blur) {
      cols = (int)((fabsf(dx)) + (1.f));                 // "height" = |dx| + 1 = kernel columns
      rows = (int)((fabsf(dy)) + (1.f));                 // "width"  = |dy| + 1 = kernel rows
      const double theta = (double)atan2f(dy, dx);
      const double length = sqrt((double)dx * dx + (double)dy * dy);   // cv::norm(Point2f)
      const double c = cos(theta), s = sin(theta);
      const int x0 = (int)(s < 0 ? -s * length : 0.0);
      const int y0 = (int)(c < 0 ? -c * length : 0.0);
      if (length >= (double)kMaxTaps) {
        blur = 0;                                            // not representable here: keep the sharp template
      } else {
        int prev = -1;
        for (int k = 0; (double)k < length; ++k) {
          int x = (int)(k * s + x0), y = (int)(k * c + y0);
          x = min(max(x, 0), rows - 1);                      // the reference writes inside the kernel matrix
          y = min(max(y, 0), cols - 1);
          const int key = (x << 16) | y;
          if (key != prev) { taps[cnt++] = key; prev = key; }
        }
        // the cells are generated along a line: monotone in both coordinates; order them (row, col) ascending
        for (int a = 1; a < cnt; ++a) {
          const int v = taps[a];
          int b = a - 1;
          while (b >= 0 && taps[b] > v) { taps[b + 1] = taps[b]; --b; }
          taps[b + 1] = v;
        }
        int u = 0;
        for (int a = 0; a < cnt; ++a)
          if (a == 0 || taps[a] != taps[u - 1]) taps[u++] = taps[a];
        cnt = u;
      }
    }
    ntaps = cnt; krows = rows; kcols = cols; do_blur = blur;
  }
  __syncthreads();
  if (!do_blur) {
    for (int t = threadIdx.x; t < w2; t += blockDim.x) M[t] = src[t];
    return;
  }
  const double coef = 1.0 / (double)ntaps;                  // kernel / sum(kernel)
  const int ay = krows / 2, ax = kcols / 2;                 // anchor (-1,-1) = kernel centre
  for (int t = threadIdx.x; t < w2; t += blockDim.x) {
    const int y = t / w, x = t % w;
    double acc = 0.0;
    for (int k = 0; k < ntaps; ++k) {
      const int r = taps[k] >> 16, cc = taps[k] & 0xffff;
      const int sy = reflect101(y + r - ay, w), sx = reflect101(x + cc - ax, w);
      const double term = coef * (double)src[sy * w + sx];
      acc = acc + term;
    }
    M[t] = saturate_u8(acc);
  }
}

// Patch::findMatch (Patch.cpp:215-293) for every visible feature: the candidate positions inside the
// clamped sigma_size box and the Mahalanobis ellipse of the 2x2 innovation block are scored with the NCC of
// computeCorrelation (Patch.cpp:295-329); the first maximum in scan order (u outer, v inner) wins;
// found iff score >= threshold.  The sums of the NCC are exact integers.
template <typename T>
__global__ void __launch_bounds__(256)
k_ncc_search(const unsigned char* __restrict__ frame, int fw, int fh, const T* __restrict__ h,
             const T* __restrict__ Sd, const unsigned char* __restrict__ flags, int w, float sigma_size,
             float threshold, unsigned char* __restrict__ mpatch, T* __restrict__ z_out,
             unsigned char* __restrict__ found, float* __restrict__ score) {
#pragma clang fp contract(off)
  const int i = blockIdx.x;
  const int tid = threadIdx.x;
  if (!(flags[i] & 1)) {
    if (tid == 0) { found[i] = 0; score[i] = -1.f; z_out[2 * i] = T(-1); z_out[2 * i + 1] = T(-1); }
    return;
  }
  constexpr int REG = kMaxWindow + 2 * kMaxSearch + 1;      // 73: widest / tallest region
  constexpr int RSMAX = (REG + 3 + 3) & ~3;                 // row stride of the staged region, multiple of 4
  // template rows padded with zeros to whole 32-bit words; the region as bytes with a word-aligned row stride:
  // a candidate row is wq + 1 aligned words, realigned with v_alignbyte and multiplied 4 pixels at a time
  // (v_dot4_u32_u8) -- all sums stay exact integers
  __shared__ unsigned int tplw[kMaxWindow * (kMaxWindow / 4)];
  __shared__ __attribute__((aligned(16))) unsigned char reg[REG * RSMAX + 16];
  __shared__ float s_val[256];
  __shared__ int s_idx[256];
  __shared__ int s_st, s_stt;
  const int w2 = w * w, hw = w / 2;
  unsigned char* M = mpatch + (size_t)i * w2;
  // covariance block (row-major s00 s01 s10 s11) as the reference's MatrixXf
  const float s00 = (float)Sd[4 * i], s01 = (float)Sd[4 * i + 1], s10 = (float)Sd[4 * i + 2], s11 = (float)Sd[4 * i + 3];
  // MatrixXf::inverse() of a dynamic matrix = PartialPivLU: row pivot, unit-lower / upper solves of P
  float inv00, inv10, inv11;
  {
    const bool swap = fabsf(s10) > fabsf(s00);
    const float p = swap ? s10 : s00, q = swap ? s11 : s01, r = swap ? s00 : s10, s = swap ? s01 : s11;
    const float l = ((r) / (p));
    const float u11 = ((s) - (((l) * (q))));
    const float ip = ((1.f) / (p)), iu = ((1.f) / (u11));
    // columns of P: no swap -> identity; swap -> [e1 e0]
    float x0[2], x1[2];
    for (int c = 0; c < 2; ++c) {
      const float b0 = (swap ? (c == 1) : (c == 0)) ? 1.f : 0.f;
      const float b1 = (swap ? (c == 0) : (c == 1)) ? 1.f : 0.f;
      const float y1 = ((b1) - (((l) * (b0))));
      x1[c] = ((y1) * (iu));
      x0[c] = (b0 - q * x1[c]) * ip;
    }
    inv00 = x0[0]; inv10 = x1[0]; inv11 = x1[1];
  }
  const float x2c = inv00, y2c = inv11, yxc = ((2.f) * (inv10));
  const float sigma2 = ((sigma_size) * (sigma_size));
  const int uc = (int)(float)h[2 * i], vc = (int)(float)h[2 * i + 1];
  float du = (float)((double)sigma_size * sqrt((double)s00));
  float dv = (float)((double)sigma_size * sqrt((double)s11));
  if (du > 20.f) du = 20.f;
  if (dv > 20.f) dv = 20.f;
  const int i0 = (int)(((float)uc) - (du)), i1 = (int)floorf((((float)uc) + (du)));
  const int j0 = (int)(((float)vc) - (dv)), j1 = (int)floorf((((float)vc) + (dv)));
  const int ni = i1 - i0 + 1, nj = j1 - j0 + 1;
  // stage the template and the image region every candidate window can touch
  int st = 0, stt = 0;
  const int wq = (w + 3) >> 2;                               // words per template row
  for (int t = tid; t < w * wq; t += 256) {
    const int y = t / wq, k = t % wq;
    unsigned int v = 0;
    for (int b = 0; b < 4; ++b)
      if (4 * k + b < w) v |= (unsigned int)M[y * w + 4 * k + b] << (8 * b);
    tplw[t] = v;
  }
  const int rx0 = i0 - hw, ry0 = j0 - hw, rw = ni + w, rh = nj + w;
  const int RS = (rw + 3 + 3) & ~3;
  const bool region_ok = ni > 0 && nj > 0 && rw <= REG && rh <= REG;
  if (region_ok)
    for (int t = tid; t < RS * rh + 16; t += 256) {
      const int ly = t / RS, lx = t % RS;
      const int yy = ry0 + ly, xx = rx0 + lx;
      reg[t] = (ly < rh && lx < rw && yy >= 0 && yy < fh && xx >= 0 && xx < fw) ? frame[(size_t)yy * fw + xx] : 0;
    }
  if (tid == 0) { s_st = 0; s_stt = 0; }
  __syncthreads();
  {
    int a = 0, b = 0;
    for (int t = tid; t < w * wq; t += 256) {
      const unsigned int v = tplw[t];
      a = __builtin_amdgcn_udot4(v, 0x01010101u, a, false);
      b = __builtin_amdgcn_udot4(v, v, b, false);
    }
    if (a | b) { atomicAdd(&s_st, a); atomicAdd(&s_stt, b); }
  }
  __syncthreads();
  st = s_st; stt = s_stt;
  const unsigned int lastmask = (w & 3) ? ((1u << (8 * (w & 3))) - 1u) : 0xffffffffu;
  const unsigned int* regw = reinterpret_cast<const unsigned int*>(reg);
  float best = -1.f;                                         // "float max = -1"
  int best_c = -1;
  if (region_ok) {
    const long long nn = w2;
    const long long d1 = nn * stt - (long long)st * st;
    for (int c = tid; c < ni * nj; c += 256) {
      const int ci = i0 + c / nj, cj = j0 + c % nj;
      if (!(ci > hw && cj > hw && ci < fw - hw && cj < fh - hw)) continue;
      const float fi = (float)(ci - uc), fj = (float)(cj - vc);
      const float g = x2c * fi * fi + y2c * fj * fj + yxc * fi * fj;      // the reference's expression, left to right
      if (!(g <= sigma2)) continue;
      int ss = 0, sss = 0, sts = 0;
      const int o0 = (cj - hw - ry0) * RS + (ci - hw - rx0);     // byte offset of the window's first pixel
      const int sh = o0 & 3;
      const unsigned int* rp = regw + (o0 >> 2);
      for (int y = 0; y < w; ++y) {
        const unsigned int* rr = rp + y * (RS >> 2);
        unsigned int lo = rr[0];
        for (int k = 0; k < wq; ++k) {
          const unsigned int hi = rr[k + 1];
          unsigned int bw = __builtin_amdgcn_alignbyte(hi, lo, sh);
          lo = hi;
          if (k == wq - 1) bw &= lastmask;
          const unsigned int aw = tplw[y * wq + k];
          ss = __builtin_amdgcn_udot4(bw, 0x01010101u, ss, false);
          sss = __builtin_amdgcn_udot4(bw, bw, sss, false);
          sts = __builtin_amdgcn_udot4(aw, bw, sts, false);
        }
      }
      const long long num = nn * sts - (long long)st * ss;
      const long long d2 = nn * sss - (long long)ss * ss;
      const float val = (float)((double)num / sqrt((double)d1 * (double)d2));   // 0/0 -> NaN: never chosen
      if (val > best) { best = val; best_c = c; }
    }
  }
  s_val[tid] = best;
  s_idx[tid] = best_c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      const float v2 = s_val[tid + off];
      const int c2 = s_idx[tid + off];
      // larger score wins; equal scores: the earlier candidate in scan order (strict > in the reference loop)
      if (c2 >= 0 && (s_idx[tid] < 0 || v2 > s_val[tid] || (v2 == s_val[tid] && c2 < s_idx[tid]))) {
        s_val[tid] = v2;
        s_idx[tid] = c2;
      }
    }
    __syncthreads();
  }
  const float mx = s_val[0];
  const int mc = s_idx[0];
  const bool ok = (mc >= 0) && !(mx < threshold);
  if (ok) {
    const int ci = i0 + mc / nj, cj = j0 + mc % nj;
    if (tid == 0) { found[i] = 1; score[i] = mx; z_out[2 * i] = T(ci); z_out[2 * i + 1] = T(cj); }
    // "this->matching_patch = newPatch" (Patch.cpp:286): the matched window replaces the matching template
    const unsigned char* rp = reg + (cj - hw - ry0) * RS + (ci - hw - rx0);
    for (int t = tid; t < w2; t += 256) M[t] = rp[(t / w) * RS + (t % w)];
  } else if (tid == 0) {
    found[i] = 0;
    score[i] = (mc >= 0) ? mx : -1.f;
    z_out[2 * i] = T(-1);
    z_out[2 * i + 1] = T(-1);
  }
}

}  // namespace ekf
