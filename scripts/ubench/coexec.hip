// Micro-experiment: can f32 MFMA and packed-f32 VALU FMAs run concurrently on one CU at (near) their standalone rates?
// mode 0: all 8 waves/WG MFMA; mode 1: all VALU pk_fma; mode 2: waves 0-3 MFMA, waves 4-7 VALU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, float seed) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) || (mode == 2 && wave < 4);
  float r = 0.f;
  if (do_mfma) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f + 1.0f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j) r += a0[j] + a1[j] + a2[j] + a3[j];
  } else {
    f32x2 c[16];
    for (int j = 0; j < 16; ++j) c[j] = f32x2{seed + j, seed - j};
    f32x2 a = {seed * 1.0001f, seed * 0.9999f}, b = {0.5f, 0.25f};
    // per loop trip: 64 pk_fma = 128 lanes-FMAs... each v_pk_fma_f32 = 2 FMA/lane = 256 FLOP/wave
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = __builtin_elementwise_fma(c[j], a, b);
    }
    for (int j = 0; j < 16; ++j) r += c[j][0] + c[j][1];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  float* out; hipMalloc(&out, 1024 * 512 * 4);
  const int iters = 20000, grid = 256;
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, 100, mode, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, iters, mode, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // FLOPs: MFMA wave per iter: 4 * 4096 = 16384; VALU wave per iter: 64 pk_fma * 64 lanes * 4 flop = 16384
    double mf = 0, vf = 0;
    int nm = mode == 0 ? 8 : (mode == 2 ? 4 : 0), nv = 8 - nm;
    mf = (double)grid * nm * iters * 16384.0; vf = (double)grid * nv * iters * 16384.0;
    printf("mode %d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  total %.1f TF\n", mode, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
  }
  return 0;
}
