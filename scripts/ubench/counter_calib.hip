// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns the resident convolution kernels use
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Every kernel below moves each byte of a [64,112,112,128] f32 tensor (411,041,792 B) exactly ONCE:
//   reads   r_stream_global   global_load_dwordx4, lane l of a wave reads base + 16 l (the guide's baseline: FETCH_SIZE = 1/2 bytes)
//           r_stream_dma      buffer_load_dwordx4 ... lds, lane l fetches base + 16 l (contiguous 1 KiB per instruction)
//           r_gather_dma      buffer_load_dwordx4 ... lds with per-lane offsets as the producer / consumer 3x3 kernel's patch DMA
//                             issues them (conv_halo_pc.hip, DMAP): one instruction = 16 consecutive pixels x the 64 B of ONE plane
//                             of one 32-channel group of a pre-split tensor (4 lanes x 16 B per pixel; the pixel pitch is 512 B),
//                             the other plane of the same 128-byte groups by the next instruction
//   writes  w_stream_nt       16-byte nontemporal stores, lane l writes base + 16 l
//           w_stream_plain    the same with default-policy stores
//           w_quad_nt         the producer / consumer kernel's epilogue: a wave owns 64 pixels (an 8x8 block) x 64 channels of a
//                             128-channel NHWC tensor; per store instruction 4 lanes cover 64 contiguous bytes of one pixel (16
//                             channels), 16 pixels per instruction; the neighbouring 64 bytes of the same 128-byte line follow
//                             three instructions later (conv_halo_pc.hip epilogue: sgg_quad_transpose4 + sgg_out_store4)
//           w_quad_plain      the same with default-policy stores
//   rocprofv3 --pmc FETCH_SIZE -- ./counter_calib.bin     and     rocprofv3 --pmc WRITE_SIZE -- ./counter_calib.bin
// (separate passes; scripts/gpu_calib.sh); factors = counter KiB * 1024 / 411041792 per kernel -> profiles/r05_counter_calibration.log
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define B_ 64
#define H_ 112
#define W_ 112
#define N_ 128
static const size_t BYTES = (size_t)B_ * H_ * W_ * N_ * 4;

__device__ __forceinline__ void dma16(v4i rs, unsigned lds_addr, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
               : "memory");
}
__device__ __forceinline__ v4i rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  return v4i{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

// grid-stride over 1-KiB pieces; `sink` keeps the loads alive
__global__ __launch_bounds__(256) void r_stream_global(const f4* __restrict__ src, size_t n16, float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(src + i);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) *sink = 1.f;
}

__global__ __launch_bounds__(256) void r_stream_dma(const float* __restrict__ src, unsigned bytes, float* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 4096];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const v4i rs = rsrc(src, bytes);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds + wave * 4096);
  const unsigned npieces = bytes >> 10;
  for (unsigned p = (blockIdx.x * 4 + wave) * 4; p < npieces; p += gridDim.x * 16) {
#pragma unroll
    for (int k = 0; k < 4; ++k) dma16(rs, base + k * 1024, (p + k) * 1024u + lane * 16u, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  if (reinterpret_cast<float*>(lds)[threadIdx.x] == 123.456f) *sink = 1.f;
}

// pre-split layout: pixel = 128 channels = 4 groups of 128 B, each group = 64 B leading + 64 B residual pieces.  One instruction:
// 16 consecutive pixels (of a row-major pixel stream) x 4 lanes x 16 B of plane pp of group cg; the (pixel block, cg) pairs are
// dealt to the waves, both planes by consecutive instructions of the same wave.
__global__ __launch_bounds__(256) void r_gather_dma(const float* __restrict__ src, unsigned bytes, float* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 4096];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const v4i rs = rsrc(src, bytes);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds + wave * 4096);
  const unsigned npix16 = (unsigned)((size_t)B_ * H_ * W_ / 16);
  const unsigned lane_off = (unsigned)(lane >> 2) * (N_ * 4) + (unsigned)(lane & 3) * 16u;
  for (unsigned it = blockIdx.x * 4 + wave; it < npix16 * 4; it += gridDim.x * 4) {
    const unsigned pb = it >> 2, cg = it & 3;
    const unsigned off = pb * 16u * (N_ * 4) + cg * 128u + lane_off;
    dma16(rs, base, off, 0);            // leading pieces
    dma16(rs, base + 1024, off, 64);    // residual pieces (the plane's byte offset in the SCALAR offset, as the kernels do)
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  if (reinterpret_cast<float*>(lds)[threadIdx.x] == 123.456f) *sink = 1.f;
}

__global__ __launch_bounds__(256) void w_stream_nt(f4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, 4.f}, dst + i);
}
__global__ __launch_bounds__(256) void w_stream_plain(f4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = f4{1.f, 2.f, 3.f, 4.f};
}

// a wave = one 8x8 block x one 64-column half; 16 store instructions (i = row pair, j = 16-column group) as the epilogue issues them
template <bool NT>
__global__ __launch_bounds__(256) void w_quad(float* __restrict__ dst) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l16 = lane & 15, c4 = lane >> 4;
  const int nblk = B_ * (H_ / 8) * (W_ / 8);
  const unsigned o_lane = (unsigned)((c4 >> 1) * W_ * N_ + (4 * (c4 & 1) + (lane & 3)) * N_ + (l16 >> 2) * 4);
  for (int item = blockIdx.x * 4 + wave; item < nblk * 2; item += gridDim.x * 4) {
    const int blk = item >> 1, half = item & 1;
    const int b = blk / ((H_ / 8) * (W_ / 8)), r = blk % ((H_ / 8) * (W_ / 8));
    const int by = r / (W_ / 8), bx = r % (W_ / 8);
    float* ob = dst + (((size_t)b * H_ + by * 8) * W_ + bx * 8) * N_ + half * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f4* p = reinterpret_cast<f4*>(ob + (size_t)(2 * i) * W_ * N_ + j * 16 + o_lane);
        if (NT) __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)j}, p);
        else *p = f4{1.f, 2.f, 3.f, (float)j};
      }
  }
}

int main() {
  float *a, *sink;
  if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 0, BYTES);
  const size_t n16 = BYTES / 16;
  const dim3 grid(2048), blk(256);
  for (int rep = 0; rep < 3; ++rep) {
    // (every kernel streams 411 MB, more than the 256 MB Infinity Cache: nothing of `a` is left on-die for the next one)
    hipLaunchKernelGGL(r_stream_global, grid, blk, 0, 0, (const f4*)a, n16, sink);
    hipLaunchKernelGGL(r_stream_dma, grid, blk, 0, 0, a, (unsigned)BYTES, sink);
    hipLaunchKernelGGL(r_gather_dma, grid, blk, 0, 0, a, (unsigned)BYTES, sink);
    hipLaunchKernelGGL(w_stream_nt, grid, blk, 0, 0, (f4*)a, n16);
    hipLaunchKernelGGL(w_stream_plain, grid, blk, 0, 0, (f4*)a, n16);
    hipLaunchKernelGGL(w_quad<true>, grid, blk, 0, 0, a);
    hipLaunchKernelGGL(w_quad<false>, grid, blk, 0, 0, a);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("run failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  // every element must have been written by w_quad: spot check on the host
  float h[4];
  hipMemcpy(h, a + (BYTES / 4 - 4), 16, hipMemcpyDeviceToHost);
  printf("bytes per kernel %zu; last 4 floats %.0f %.0f %.0f %.0f (expect 1 2 3 3)\n", BYTES, h[0], h[1], h[2], h[3]);
  return 0;
}
