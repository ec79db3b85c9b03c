// Ground truth for the column chain of the diagonal factor: cycles per iteration of dependent
// MFMA -> VALU select -> VALU multiply -> MFMA loops (one workgroup, one wave active), for the two MFMA shapes,
// and the shader clock while such a latency-bound kernel runs (s_memtime ticks per s_memrealtime tick, 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k_probe(unsigned long long* out, float* sink, int iters, int mode) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if (mode == 0) {            // 16x16x4: mfma -> cndmask -> mul -> mfma
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f};
    for (int i = 0; i < iters; ++i) {
      const float am = (lane >= (i & 15)) ? acc[i & 3] : 0.f;
      const float v = am * 0.999f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-v, v, acc, 0, 0, 0);
    }
    s = acc[0] + acc[1] + acc[2] + acc[3];
  } else if (mode == 1) {     // 32x32x2
    f16v acc;
    for (int e = 0; e < 16; ++e) acc[e] = 1.f + lane + e;
    for (int i = 0; i < iters; ++i) {
      const float am = (lane >= (i & 31)) ? acc[i & 15] : 0.f;
      const float v = am * 0.999f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-v, v, acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) s += acc[e];
  } else if (mode == 2) {     // 16x16x4 with the readlane + rsq path of the real kernel
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f};
    float pk = 2.f;
    for (int i = 0; i < iters; ++i) {
      const float iv = __frsqrt_rn(pk);
      const float am = (lane >= (i & 15)) ? acc[i & 3] : 0.f;
      const float v = am * iv;
      const float l10 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (i + 1) & 63));
      const float a11 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[(i + 1) & 3]), (i + 1) & 63));
      pk = __builtin_fmaf(-l10, l10, a11) * 1e-30f + 2.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-v, v, acc, 0, 0, 0);
    }
    s = acc[0] + acc[1] + acc[2] + acc[3];
  } else if (mode == 3) {     // back-to-back dependent 16x16x4 MFMAs only
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f};
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(0.5f, 0.5f, acc, 0, 0, 0);
    s = acc[0];
  } else if (mode == 4) {     // dependent VALU chain only (fma)
    float x = 1.f + lane;
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 0.999f, 0.001f);
    s = x;
  } else if (mode == 5) {     // 4x4x1 (16 blocks): mfma -> cndmask -> mul -> mfma
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f};
    for (int i = 0; i < iters; ++i) {
      const float am = (lane >= (i & 15)) ? acc[i & 3] : 0.f;
      const float v = am * 0.999f;
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(-v, v, acc, 0, 0, 0);
    }
    s = acc[0] + acc[1] + acc[2] + acc[3];
  } else if (mode == 6) {     // 16x16x4: mfma -> ONE v_mul -> mfma (A = accumulator register itself)
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f};
    const float sc = (lane & 16) ? -1e-3f : 0.f;
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float rk = acc[e]; acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rk, rk * sc, acc, 0, 0, 0); }
    }
    s = acc[0] + acc[1] + acc[2] + acc[3];
  } else if (mode == 7) {     // mode 6 + a second, independent MFMA chain fed by the same product (the inverse)
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f}, xac = {1.f, 0.f, 0.f, 0.f};
    const float sc = (lane & 16) ? -1e-3f : 0.f;
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float rk = acc[e], t = rk * sc;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rk, t, acc, 0, 0, 0);
        xac = __builtin_amdgcn_mfma_f32_16x16x4f32(t, xac[e], xac, 0, 0, 0);
      }
    }
    s = acc[0] + acc[1] + acc[2] + acc[3] + xac[0] + xac[1] + xac[2] + xac[3];
  } else if (mode == 8) {     // mode 7 + the pivot path: two v_readlane of the accumulator, fma, rsq, scale vector
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f}, xac = {1.f, 0.f, 0.f, 0.f};
    float pk = 2.f;
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float iv = __frsqrt_rn(pk), ipk = iv * iv;
        const float sc = ((lane & 15) > e + 3) ? -ipk * 1e-3f : 0.f;
        const float rk = acc[e];
        const float a10 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rk), 17 + e));
        const float a11 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[(e + 1) & 3]), 18 + e));
        pk = __builtin_fmaf(-a10 * ipk, a10, a11) * 1e-30f + 2.f;
        const float t = rk * sc;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rk, t, acc, 0, 0, 0);
        xac = __builtin_amdgcn_mfma_f32_16x16x4f32(t, xac[e], xac, 0, 0, 0);
      }
    }
    s = acc[0] + acc[1] + acc[2] + acc[3] + xac[0] + xac[1] + xac[2] + xac[3];
  } else if (mode == 9) {     // mode 8 without the inverse chain
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f};
    float pk = 2.f;
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float iv = __frsqrt_rn(pk), ipk = iv * iv;
        const float sc = ((lane & 15) > e + 3) ? -ipk * 1e-3f : 0.f;
        const float rk = acc[e];
        const float a10 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rk), 17 + e));
        const float a11 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[(e + 1) & 3]), 18 + e));
        pk = __builtin_fmaf(-a10 * ipk, a10, a11) * 1e-30f + 2.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rk, rk * sc, acc, 0, 0, 0);
      }
    }
    s = acc[0] + acc[1] + acc[2] + acc[3];
  } else if (mode == 10) {    // mode 6 with an LDS store of the scaled row per column
    __shared__ float buf[64 * 20];
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f};
    const float sc = (lane & 16) ? -1e-3f : 0.f;
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float rk = acc[e]; acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rk, rk * sc, acc, 0, 0, 0); buf[lane * 20 + e] = rk * 0.5f; }
    }
    s = acc[0] + acc[1] + acc[2] + acc[3] + buf[(lane * 7) % 1280];
  } else if (mode >= 11 && mode <= 16) {
    // the column of diag_factor16 as written (MFMAs first, pivot path behind them) with pieces removed:
    // 11 all; 12 no readlanes (values from a VALU op); 13 no reciprocal (a multiply); 14 neither; 15 all, without the inverse MFMA; 16 all + rsq + 2 LDS stores
    __shared__ float buf[64 * 20];
    f4 acc = {1.f + lane, 2.f, 3.f, 4.f}, xac = {1.f, 0.f, 0.f, 0.f};
    float pk = 2.f, nipk = -0.5f;
    const int key = lane & 15;
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float rk = acc[e];
        const f4 prev = acc;
        const float t = ((key > e + 3) ? rk : 0.f) * nipk;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rk, t, acc, 0, 0, 0);
        if (mode != 15) xac = __builtin_amdgcn_mfma_f32_16x16x4f32(t, xac[e], xac, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const float pk_k = pk;
        float a10, a11;
        if (mode == 12 || mode == 14) { a10 = rk * 1e-3f; a11 = prev[(e + 1) & 3] * 1e-3f; }
        else {
          a10 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rk), 17 + e));
          a11 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prev[(e + 1) & 3]), 18 + e));
        }
        pk = __builtin_fmaf(a10 * nipk, a10, a11) * 1e-30f + 2.f;
        if (mode == 13 || mode == 14) nipk = -pk * 0.25f; else nipk = -__builtin_amdgcn_rcpf(pk);
        if (mode == 16) { const float iv = __frsqrt_rn(pk_k); buf[lane * 20 + e] = rk * iv; buf[lane * 20 + 8 + e] = iv; }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    s = acc[0] + acc[1] + acc[2] + acc[3] + xac[0] + xac[1] + xac[2] + xac[3] + buf[(lane * 7) % 1280];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  sink[lane] = s;
}
int main() {
  unsigned long long* d; float* sink; hipMalloc(&d, 16); hipMalloc(&sink, 256);
  const char* names[] = {"16x16x4 mfma->cndmask->mul->mfma", "32x32x2 mfma->cndmask->mul->mfma", "16x16x4 + rsq + 2 readlanes (real column)",
                         "16x16x4 dependent mfma only", "dependent v_fma chain", "4x4x1 mfma->cndmask->mul->mfma",
                         "16x16x4 mfma->ONE mul->mfma (A = acc reg)", "... + independent inverse MFMA chain", "... + pivot path (2 readlane, fma, rsq, scale)",
                         "pivot path without the inverse chain", "mfma->mul->mfma + LDS store per column",
                         "column as written: all", "  - readlanes", "  - reciprocal", "  - both", "  all, no inverse MFMA", "  all + rsq + 2 LDS stores"};
  for (int rep = 0; rep < 1; ++rep)
    for (int mode = 6; mode < 17; ++mode) {
      const int iters = 20000;
      k_probe<<<1, 512>>>(d, sink, iters, mode);
      hipDeviceSynchronize();
      unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("%-44s %7.1f memtime ticks/iter  %7.2f ns/iter  (memtime/realtime ratio %.2f -> %.0f MHz if memtime counts shader clocks)\n", names[mode],
             (double)h[0] / iters, (double)h[1] * 10.0 / iters, (double)h[0] / h[1], (double)h[0] / h[1] * 100.0);
    }
  return 0;
}
