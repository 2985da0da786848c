// Micro-experiment: MFMA pipe occupancy of a "load phase / MFMA phase" loop (the shape of one tap of the halo conv kernel:
// 8 LDS fragment reads + 8 L2 fragment loads + some VALU, then 24 x v_mfma_f32_32x32x16_f16 on the operands fetched one
// iteration earlier) under three schedules:
//   mode 0: 2 workgroups x 4 waves per CU, free running (what the kernels do today: two unsynchronised waves per SIMD)
//   mode 1: 1 workgroup x 8 waves per CU, ping-pong: waves 0-3 issue MFMAs while waves 4-7 fetch, swapped at every s_barrier
//   mode 2: as mode 1 plus s_setprio 1 around the MFMA phase
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512, 1) void k(float* out, const u32x4* __restrict__ gsrc, int iters) {
  __shared__ u32x4 lds[2048];                       // 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2048; i += blockDim.x) lds[i] = gsrc[i];
  __syncthreads();
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  u32x4 a[2][4], b[2][4];
  auto fetch = [&](int buf, int it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[buf][j] = lds[(lane * 4 + j + it * 7 + wave * 64) & 2047];
      b[buf][j] = gsrc[(lane + 64 * j + it * 256 + wave * 1024) & 16383];
    }
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[buf][(j + u) & 3]),
                                                        __builtin_bit_cast(f16x8, b[buf][(j * 3 + u) & 3]), acc[j], 0, 0, 0);
  };
  const int grp = (MODE == 0) ? 0 : (wave >> 2);
  fetch(0, 0);
  if (MODE != 0 && grp == 1) __builtin_amdgcn_s_barrier();      // phase shift of the second wave group
  for (int it = 0; it < iters; it += 2) {
    fetch(1, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 0) __builtin_amdgcn_s_barrier();
    if (MODE == 2) __builtin_amdgcn_s_setprio(1);
    mma(0);
    if (MODE == 2) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 0) __builtin_amdgcn_s_barrier();
    fetch(0, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 0) __builtin_amdgcn_s_barrier();
    if (MODE == 2) __builtin_amdgcn_s_setprio(1);
    mma(1);
    if (MODE == 2) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 0) __builtin_amdgcn_s_barrier();
  }
  if (MODE != 0 && grp == 0) __builtin_amdgcn_s_barrier();
  float r = 0.f;
  for (int j = 0; j < 16; ++j) r += acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j];
  out[blockIdx.x * blockDim.x + tid] = r;
}

template <int MODE>
static void run(float* out, const u32x4* src, const char* name) {
  const int iters = 4000;
  const dim3 grid(MODE == 0 ? 512 : 256), blk(MODE == 0 ? 256 : 512);
  hipLaunchKernelGGL(k<MODE>, grid, blk, 0, 0, out, src, 20);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, grid, blk, 0, 0, out, src, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double fl = 2048.0 * iters * 24 * 32768.0;       // 2048 waves on the chip in every mode
  printf("%-34s %.2f ms  %.0f TFLOP/s f16 MFMA (= %.0f TFLOP/s of 3-product f32)\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 3);
}

int main() {
  float* out; (void)hipMalloc(&out, 512 * 512 * 4);
  _Float16* h = (_Float16*)malloc(16384 * 16);
  for (int i = 0; i < 16384 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  u32x4* src; (void)hipMalloc(&src, 16384 * 16);
  (void)hipMemcpy(src, h, 16384 * 16, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(out, src, "free running (2 WG x 4 waves)");
    run<1>(out, src, "ping-pong (1 WG x 8 waves)");
    run<2>(out, src, "ping-pong + setprio");
  }
  return 0;
}
