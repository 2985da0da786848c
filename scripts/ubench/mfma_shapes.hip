// Micro-experiment: sustained rate (random operands, power-limited) of the two f16 MFMA shapes on MI355X:
// v_mfma_f32_32x32x16_f16 (8 passes, 16 accumulator VGPRs) vs v_mfma_f32_16x16x32_f16 (4 passes, 4 accumulator VGPRs).
// Same FLOPs per pass; the 16x16 shape moves half the accumulator bytes per FLOP through the register file.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, const f16x8* in, int iters, unsigned long long* ticks) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  f16x8 a[4], b[4];
  for (int j = 0; j < 4; ++j) { a[j] = in[(threadIdx.x + 64 * j) & 1023]; b[j] = in[(threadIdx.x * 3 + 17 * j) & 1023]; }
  float r = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + u) & 3], b[(j * 3 + u) & 3], acc[j], 0, 0, 0);
    for (int j = 0; j < 16; ++j) r += acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j];
  } else {
    f32x4 acc[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(j + u) & 3], b[(j * 3 + u) & 3], acc[j], 0, 0, 0);
    for (int j = 0; j < 8; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = __builtin_readcyclecounter() - t0;
}

template <int SHAPE>
static void run(float* out, const f16x8* in, unsigned long long* dt, const char* name) {
  const int grid = 512, iters = 30000;
  hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, out, in, 100, dt);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, out, in, iters, dt);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long ht = 0; (void)hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
  const double fl = (double)grid * 4 * iters * 24 * 32768.0;     // both shapes: 24 x 32768 FLOP per wave and trip (48 x 16384)
  printf("%-28s %.2f ms  %.0f TFLOP/s  shader clock %.3f GHz\n", name, ms, fl / ms / 1e9, ht / (ms * 1e6));
}

int main() {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  _Float16* h = (_Float16*)malloc(1024 * 16);
  for (int i = 0; i < 8192; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  f16x8* in; (void)hipMalloc(&in, 1024 * 16);
  (void)hipMemcpy(in, h, 1024 * 16, hipMemcpyHostToDevice);
  unsigned long long* dt; (void)hipMalloc(&dt, 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<32>(out, in, dt, "32x32x16 f16 (random data)");
    run<16>(out, in, dt, "16x16x32 f16 (random data)");
  }
  return 0;
}
