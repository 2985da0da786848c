#include <hip/hip_runtime.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* o) {
  __shared__ _Float16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)i;
  __syncthreads();
  const int lane = threadIdx.x;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  // block: rows = k (4), cols = 16 channels; row stride 128 halves
  auto* ptr = (__attribute__((address_space(3))) s16x4*)(lds + (q) * 128 + g * 16 + 4 * p);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  for (int e = 0; e < 4; ++e) o[lane * 4 + e] = (float)v[e];
}
int main() {
  float* o; hipMalloc(&o, 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  float h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) { int c = (l >> 4) * 16 + (l & 15); if (h[l * 4 + e] != e * 128 + c) ++bad; }
  printf("lane0: %g %g %g %g  lane17: %g %g %g %g  lane63: %g %g %g %g  bad=%d\n", h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71], h[252], h[253], h[254], h[255], bad);
  return 0;
}
