// Device-side scalar math of the EKF-MonoSLAM hot path: quaternion helpers, the
// pinhole + radial/tangential camera model and the inverse-depth parametrisation.
// Written from the equations (reference: mono-slam/src/vslamRansac.cpp:1388-1661,
// mono-slam/src/camModel.cpp:18-192); every routine is a plain inline function on
// registers so that one lane evaluates one feature.
#pragma once
#include <hip/hip_runtime.h>

namespace ekf {

struct CamParams {
  float fx, fy, u0, v0, k1, k2, k3, p1, p2;
  int width, height, half_window;  // half_window = window_size / 2 (integer division)
};

template <typename T> __device__ __forceinline__ T t_sin(T x);
template <> __device__ __forceinline__ float t_sin<float>(float x) { return sinf(x); }
template <> __device__ __forceinline__ double t_sin<double>(double x) { return sin(x); }
template <typename T> __device__ __forceinline__ T t_cos(T x);
template <> __device__ __forceinline__ float t_cos<float>(float x) { return cosf(x); }
template <> __device__ __forceinline__ double t_cos<double>(double x) { return cos(x); }
template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_atan2(T y, T x);
template <> __device__ __forceinline__ float t_atan2<float>(float y, float x) { return atan2f(y, x); }
template <> __device__ __forceinline__ double t_atan2<double>(double y, double x) { return atan2(y, x); }
template <typename T> __device__ __forceinline__ T t_abs(T x) { return x < T(0) ? -x : x; }
// a * b + c with ONE rounding, spelled out: fma(a, b, c) == fma(b, a, c) bit for bit, so a row strip and the column
// strip that mirrors it come out exactly symmetric whatever the compiler would have contracted on its own
__device__ __forceinline__ float t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// q = quat(angle-axis vec)                                     (vR.cpp:1388-1406)
template <typename T>
__device__ __forceinline__ void vec2quat(const T v[3], T q[4]) {
  T alpha = t_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (alpha != T(0)) {
    T s = t_sin(alpha / T(2)) / alpha;
    q[0] = t_cos(alpha / T(2));
    q[1] = v[0] * s; q[2] = v[1] * s; q[3] = v[2] * s;
  } else {
    q[0] = T(1); q[1] = q[2] = q[3] = T(0);
  }
}

// R(q), row-major 3x3                                           (vR.cpp:1408-1421)
template <typename T>
__device__ __forceinline__ void quat2rot(const T q[4], T R[9]) {
  const T r = q[0], i = q[1], j = q[2], k = q[3];
  R[0] = r * r + i * i - j * j - k * k; R[1] = T(2) * (i * j - r * k);        R[2] = T(2) * (r * j + i * k);
  R[3] = T(2) * (r * k + i * j);        R[4] = r * r - i * i + j * j - k * k; R[5] = T(2) * (j * k - r * i);
  R[6] = T(2) * (i * k - r * j);        R[7] = T(2) * (r * i + j * k);        R[8] = r * r - i * i - j * j + k * k;
}

// out = Upsilon(a) * b  (Hamilton product a (x) b)              (vR.cpp:1423-1460)
template <typename T>
__device__ __forceinline__ void quat_mul(const T a[4], const T b[4], T out[4]) {
  out[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  out[1] = a[1] * b[0] + a[0] * b[1] - a[3] * b[2] + a[2] * b[3];
  out[2] = a[2] * b[0] + a[3] * b[1] + a[0] * b[2] - a[1] * b[3];
  out[3] = a[3] * b[0] - a[2] * b[1] + a[1] * b[2] + a[0] * b[3];
}

// J[3x4] = d(R(q) d)/dq, row-major                              (vR.cpp:1537-1566, 1654-1661)
template <typename T>
__device__ __forceinline__ void drot_dq_times(const T q[4], const T d[3], T J[12]) {
  const T q0 = T(2) * q[0], qx = T(2) * q[1], qy = T(2) * q[2], qz = T(2) * q[3];
  const T x = d[0], y = d[1], z = d[2];
  // column 0: dR/dq0 * d
  J[0] = q0 * x - qz * y + qy * z;  J[4] = qz * x + q0 * y - qx * z;  J[8]  = -qy * x + qx * y + q0 * z;
  // column 1: dR/dqx * d
  J[1] = qx * x + qy * y + qz * z;  J[5] = qy * x - qx * y - q0 * z;  J[9]  = qz * x + q0 * y - qx * z;
  // column 2: dR/dqy * d
  J[2] = -qy * x + qx * y + q0 * z; J[6] = qx * x + qy * y + qz * z;  J[10] = -q0 * x + qz * y - qy * z;
  // column 3: dR/dqz * d
  J[3] = -qz * x - q0 * y + qx * z; J[7] = q0 * x - qz * y + qy * z;  J[11] = qx * x + qy * y + qz * z;
}

// Distortion Jacobian D(hn), row-major 2x2                       (cam.cpp:18-47)
template <typename T>
__device__ __forceinline__ void distort_jac(const CamParams& c, T x, T y, T D[4]) {
  const T k1 = T(c.k1), k2 = T(c.k2), k3 = T(c.k3), p1 = T(c.p1), p2 = T(c.p2);
  const T r2 = x * x + y * y;
  const T L = T(1) + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
  const T f = k1 + T(2) * k2 * r2 + T(3) * k3 * r2 * r2;
  // L I + 2 f hn hn^T + 2 [p1;p2][y x] + 2 [p2;p1][x y] + 4 diag(p2 x, p1 y)
  D[0] = L + T(2) * f * x * x + T(2) * p1 * y + T(2) * p2 * x + T(4) * p2 * x;
  D[1] =     T(2) * f * x * y + T(2) * p1 * x + T(2) * p2 * y;
  D[2] =     T(2) * f * y * x + T(2) * p2 * y + T(2) * p1 * x;
  D[3] = L + T(2) * f * y * y + T(2) * p2 * x + T(2) * p1 * y + T(4) * p1 * y;
}

// Pinhole projection with distortion; hd[2] pixel, J[2x3] row-major  (cam.cpp:68-111)
template <typename T>
__device__ __forceinline__ void project_distort(const CamParams& c, const T hC[3], T hd[2], T J[6]) {
  const T k1 = T(c.k1), k2 = T(c.k2), k3 = T(c.k3), p1 = T(c.p1), p2 = T(c.p2);
  const T x = hC[0], y = hC[1], z = hC[2];
  const T x1 = x / z, y1 = y / z;
  const T r2 = x1 * x1 + y1 * y1;
  const T l = T(1) + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
  const T x2 = x1 * l + T(2) * p1 * x1 * y1 + p2 * (r2 + T(2) * x1 * x1);
  const T y2 = y1 * l + T(2) * p2 * x1 * y1 + p1 * (r2 + T(2) * y1 * y1);
  hd[0] = T(c.fx) * x2 + T(c.u0);
  hd[1] = T(c.fy) * y2 + T(c.v0);
  T D[4];
  distort_jac(c, x1, y1, D);
  const T iz = T(1) / z;
  const T n02 = -x / z / z, n12 = -y / z / z;   // d(x/z)/dz, d(y/z)/dz
  // J = diag(fx,fy) * D * [[1/z,0,-x/z^2],[0,1/z,-y/z^2]]
  J[0] = T(c.fx) * (D[0] * iz); J[1] = T(c.fx) * (D[1] * iz); J[2] = T(c.fx) * (D[0] * n02 + D[1] * n12);
  J[3] = T(c.fy) * (D[2] * iz); J[4] = T(c.fy) * (D[3] * iz); J[5] = T(c.fy) * (D[2] * n02 + D[3] * n12);
}

// Pixel -> normalised ray by 50 fixed-point iterations; Jn = D^-1 diag(1/fx,1/fy), 2x2 row-major
// (third row of the 3x2 Jacobian is zero)                          (cam.cpp:140-192)
template <typename T>
__device__ __forceinline__ void undistort_deproject(const CamParams& c, T u, T v, T hC[3], T Jn[4]) {
  const T k1 = T(c.k1), k2 = T(c.k2), k3 = T(c.k3), p1 = T(c.p1), p2 = T(c.p2);
  const T x2 = (u - T(c.u0)) / T(c.fx);
  const T y2 = (v - T(c.v0)) / T(c.fy);
  T x1 = x2, y1 = y2;
  for (int it = 0; it < 50; ++it) {
    const T r2 = x1 * x1 + y1 * y1;
    const T l = T(1) + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
    const T dx = T(2) * p1 * x1 * y1 + p2 * (r2 + T(2) * x1 * x1);
    const T dy = T(2) * p2 * x1 * y1 + p1 * (r2 + T(2) * y1 * y1);
    x1 = (x2 - dx) / l;
    y1 = (y2 - dy) / l;
  }
  hC[0] = x1; hC[1] = y1; hC[2] = T(1);
  T D[4];
  distort_jac(c, x1, y1, D);
  const T det = D[0] * D[3] - D[1] * D[2];
  Jn[0] = (D[3] / det) / T(c.fx);  Jn[1] = (-D[1] / det) / T(c.fy);
  Jn[2] = (-D[2] / det) / T(c.fx); Jn[3] = (D[0] / det) / T(c.fy);
}

template <typename T>
__device__ __forceinline__ bool inside_image(const CamParams& c, T u, T v) {   // vR.cpp:1644-1652
  const T hw = T(c.half_window);
  return (u > hw) && (v > hw) && (u < T(c.width) - hw) && (v < T(c.height) - hw);
}

// 3x3 (row-major) times 3-vector
template <typename T>
__device__ __forceinline__ void mat3_vec(const T R[9], const T d[3], T o[3]) {
  o[0] = R[0] * d[0] + R[1] * d[1] + R[2] * d[2];
  o[1] = R[3] * d[0] + R[4] * d[1] + R[5] * d[2];
  o[2] = R[6] * d[0] + R[7] * d[1] + R[8] * d[2];
}

}  // namespace ekf
