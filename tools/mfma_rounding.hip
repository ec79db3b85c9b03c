// What does v_mfma_f32_32x32x2_f32 round like?  Known-answer cases for D = C + A B (one instruction, k = 2) and the
// signed error statistics of a K = 1024 accumulation against fp64, beside the same sum in plain v_fma_f32.
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_rounding.hip -o tools/mfma_rounding
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));

// A[i][k] = lane (i = l % 32, k = l / 32); B[k][j] = lane (k = l / 32, j = l % 32); D: vgpr v of lane l is
// row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32.
__global__ void k_one(const float* A, const float* B, const float* Cin, float* D) {
  const int l = threadIdx.x;
  f16v acc;
  for (int v = 0; v < 16; ++v) acc[v] = Cin[(8 * (v / 4) + 4 * (l / 32) + v % 4) * 32 + l % 32];
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l % 32) * 2 + l / 32], B[(l / 32) * 32 + l % 32], acc, 0, 0, 0);
  for (int v = 0; v < 16; ++v) D[(8 * (v / 4) + 4 * (l / 32) + v % 4) * 32 + l % 32] = acc[v];
}

// D (32 x 32) = sum over K of A (32 x K, row-major) B (K x 32), accumulated by K / 2 MFMAs; Dv the same by v_fma.
__global__ void k_acc(const float* A, const float* B, int K, float* D, float* Dv) {
  const int l = threadIdx.x;
  f16v acc = {0};
  for (int k = 0; k < K; k += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l % 32) * K + k + l / 32], B[(k + l / 32) * 32 + l % 32], acc, 0, 0, 0);
  for (int v = 0; v < 16; ++v) D[(8 * (v / 4) + 4 * (l / 32) + v % 4) * 32 + l % 32] = acc[v];
  for (int e = l; e < 1024; e += 64) {
    const int i = e / 32, j = e % 32;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[i * K + k], B[k * 32 + j], s);
    Dv[e] = s;
  }
}

int main() {
  float *dA, *dB, *dC, *dD, *dV;
  const int K = 1024;
  hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, K * 32 * 4); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096); hipMalloc(&dV, 4096);
  struct Case { const char* what; float c, x, y, b; };
  const float e24 = ldexpf(1, -24), e23 = ldexpf(1, -23), e40 = ldexpf(1, -40), e12 = ldexpf(1, -12), e11 = ldexpf(1, -11);
  std::vector<Case> cs = {
    {"1 + (2^-24 + 2^-40): nearest 1+2^-23, toward zero 1", 1.f, e24 + e40, 0.f, 1.f},
    {"1 + 2^-24 (tie): even 1, half-up 1+2^-23", 1.f, e24, 0.f, 1.f},
    {"(1+2^-23) + 2^-24 (tie): even 1+2^-22", 1.f + e23, e24, 0.f, 1.f},
    {"-1 - (2^-24 + 2^-40): nearest -(1+2^-23), toward zero -1", -1.f, -(e24 + e40), 0.f, 1.f},
    {"1 - (2^-25 + 2^-40): nearest 1-2^-24, toward +inf 1", 1.f, -(ldexpf(1, -25) + e40), 0.f, 1.f},
    {"1 + 2^-24 + 2^-40 over k=0,1: one rounding 1+2^-23, two roundings 1", 1.f, e24, e40, 1.f},
    {"-(1+2^-11) + (1+2^-12)^2: fused product 2^-24, rounded product 0", -(1.f + e11), 1.f + e12, 0.f, 1.f + e12},
    {"2^-140 * 2^-5 + 0 (denormal result): kept 2^-145, flushed 0", 0.f, ldexpf(1, -140), 0.f, ldexpf(1, -5)},
    {"2^-130 (denormal C) + 0", ldexpf(1, -130), 0.f, 0.f, 1.f},
  };
  std::vector<float> A(64, 0.f), B(64, 0.f), C(1024, 0.f), D(1024);
  for (size_t t = 0; t < cs.size(); ++t) {
    A[t * 2] = cs[t].x; A[t * 2 + 1] = cs[t].y;
    for (int j = 0; j < 32; ++j) C[t * 32 + j] = cs[t].c;
  }
  // column t of B carries case t's multiplier (row i = t is read at column j = t)
  for (size_t t = 0; t < cs.size(); ++t) { B[t] = cs[t].b; B[32 + t] = cs[t].b; }
  hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
  k_one<<<1, 64>>>(dA, dB, dC, dD);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  for (size_t t = 0; t < cs.size(); ++t) {
    const float d = D[t * 32 + t];
    const double exact = double(cs[t].c) + double(cs[t].x) * cs[t].b + double(cs[t].y) * cs[t].b;
    printf("%-78s  got %.10e  (%a)  host RNE of exact %a\n", cs[t].what, d, d, float(exact));
  }
  // accumulation statistics
  for (int trial = 0; trial < 3; ++trial) {
    std::vector<float> a(32 * K), b(K * 32), d(1024), dv(1024);
    srand(7 + trial);
    for (auto& v : a) v = (rand() / float(RAND_MAX)) * (trial == 1 ? 1.f : 2.f) - (trial == 1 ? 0.f : 1.f);
    for (auto& v : b) v = (rand() / float(RAND_MAX)) * (trial == 1 ? 1.f : 2.f) - (trial == 1 ? 0.f : 1.f);
    hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    k_acc<<<1, 64>>>(dA, dB, K, dD, dV);
    hipMemcpy(d.data(), dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(dv.data(), dV, 4096, hipMemcpyDeviceToHost);
    double sm = 0, sv = 0, am = 0, av = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ex = 0;
        for (int k = 0; k < K; ++k) ex += double(a[i * K + k]) * b[k * 32 + j];
        const double sg = ex >= 0 ? 1 : -1, ulp = ldexp(1.0, ilogb(fabs(ex)) - 23);
        sm += sg * (d[i * 32 + j] - ex) / ulp; am += fabs(d[i * 32 + j] - ex) / ulp;
        sv += sg * (dv[i * 32 + j] - ex) / ulp; av += fabs(dv[i * 32 + j] - ex) / ulp;
      }
    printf("K=%d %s operands: mean error toward larger magnitude, ulp of result: mfma %+.3f (|.| %.3f)   v_fma %+.3f (|.| %.3f)\n",
           K, trial == 1 ? "positive" : "signed", sm / 1024, am / 1024, sv / 1024, av / 1024);
  }
  return 0;
}
