// Does a workgroup get more than 64 KB of (static) LDS on gfx950?  Writes a pattern through 100 KB / 150 KB of LDS and reads it back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N>
__global__ __launch_bounds__(256) void k(unsigned* o) {
  __shared__ unsigned buf[N];
  for (int i = threadIdx.x; i < N; i += 256) buf[i] = i * 2654435761u + blockIdx.x;
  __syncthreads();
  unsigned bad = 0;
  for (int i = threadIdx.x; i < N; i += 256) bad += buf[(i * 97 + 13) % N] != (unsigned)((i * 97 + 13) % N) * 2654435761u + blockIdx.x;
  atomicAdd(o, bad);
}
int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  sharedMemPerBlockOptin %zu\n", pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor, pr.sharedMemPerBlockOptin);
  unsigned* d;
  hipMalloc(&d, 4);
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL(k<25600>, dim3(1024), dim3(256), 0, 0, d);   // 100 KB
  hipError_t e1 = hipDeviceSynchronize();
  unsigned h1 = 123;
  hipMemcpy(&h1, d, 4, hipMemcpyDeviceToHost);
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL(k<38400>, dim3(1024), dim3(256), 0, 0, d);   // 150 KB
  hipError_t e2 = hipDeviceSynchronize();
  unsigned h2 = 123;
  hipMemcpy(&h2, d, 4, hipMemcpyDeviceToHost);
  printf("100 KB: %s mismatches %u | 150 KB: %s mismatches %u\n", hipGetErrorString(e1), h1, hipGetErrorString(e2), h2);
  return 0;
}
