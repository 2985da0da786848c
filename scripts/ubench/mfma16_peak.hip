// Micro-experiment: sustained rate of v_mfma_f32_32x32x16_f16 on MI355X as a function of waves per SIMD and operand data
// (zeros vs random: switching activity changes the power draw and with it the clock the chip sustains).
// Loop body = what one tap of the split-f16 convolution issues: 24 MFMAs over 4 independent accumulators, no memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(float* out, const f16x8* in, int iters, unsigned long long* ticks) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  f16x8 a[4], b[4];
  for (int j = 0; j < 4; ++j) { a[j] = in[(threadIdx.x + 64 * j) & 1023]; b[j] = in[(threadIdx.x * 3 + 17 * j) & 1023]; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u + 1) & 3], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 1) & 3], b[u & 3], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 2) & 3], b[(u + 3) & 3], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 3) & 3], b[(u + 2) & 3], acc[3], 0, 0, 0);
    }
  }
  float r = 0.f;
  for (int j = 0; j < 16; ++j) r += acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (ticks && blockIdx.x == 0 && threadIdx.x == 0) *ticks = __builtin_readcyclecounter() - t0;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  _Float16* h = (_Float16*)malloc(1024 * 16);
  f16x8* in; hipMalloc(&in, 1024 * 16);
  unsigned long long* dticks; hipMalloc(&dticks, 8);
  for (int data = 0; data < 2; ++data) {
    for (int i = 0; i < 8192; ++i) h[i] = data ? (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f) : (_Float16)0.f;
    hipMemcpy(in, h, 1024 * 16, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 4; ++wps) {          // waves per SIMD = workgroups (4 waves) per CU
      const int grid = 256 * wps, iters = 40000;
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, in, 100, nullptr);
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, in, iters, dticks);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)grid * 4 * iters * 24 * 32768.0;
      // cycles per MFMA if the pipe were saturated: 1024 SIMDs
      unsigned long long ht = 0; hipMemcpy(&ht, dticks, 8, hipMemcpyDeviceToHost);
      printf("  [s_memtime: %.1f ticks per MFMA slot of this SIMD (%d waves x 24 MFMAs per trip); tick rate %.3f GHz]\n", (double)ht / ((double)iters * 24 * wps), wps, ht / (ms * 1e6));
      printf("data %s  waves/SIMD %d: %.2f ms  %.0f TFLOP/s f16 MFMA  (=%.0f TFLOP/s of 3-product f32)  implied clock at 32 cyc/MFMA: %.2f GHz\n",
             data ? "random" : "zeros", wps, ms, fl / ms / 1e9, fl / ms / 1e9 / 3, (double)grid * 4 * iters * 24 * 32 / 1024 / (ms * 1e6));
    }
  }
  return 0;
}
