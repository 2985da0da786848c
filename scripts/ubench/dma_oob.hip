// What does an LDS-DMA lane with an OUT-OF-RANGE buffer offset do to its 16 bytes of LDS?  (conv_wgrad_dma.hip / conv_halo_pc.hip
// stage halo pixels outside the image that way and need zeros.)  LDS is pre-filled with a sentinel; half of the lanes fetch in range,
// half out of range (0x80000000 + immediate offsets), through the builtin and through the inline-assembly form, below and above 64 KiB
// of LDS; the host prints what each class of lanes left behind.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int IMM>
__device__ __forceinline__ void dma16_asm(v4i rs, unsigned lds_addr, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:%4 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff),
               "n"(IMM)
               : "memory");
}

__global__ __launch_bounds__(64) void probe(const unsigned* src, unsigned nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[96 * 1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 96 * 1024 / 4; i += 64) reinterpret_cast<unsigned*>(lds)[i] = 0xDEADBEEFu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
  const unsigned long long a = reinterpret_cast<unsigned long long>(src);
  const v4i rsv = v4i{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)nbytes, 0x00020000};
  const unsigned off = (lane & 1) ? 0x80000000u : (unsigned)lane * 16u;       // odd lanes out of range
  const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;
  // region 0 (LDS 0): builtin, imm 0;  region 1 (LDS 1 KiB): builtin, imm 64;  region 2 (LDS 80 KiB): builtin imm 0
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds), 16, off, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 1024), 16, off, 0, 64, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 80 * 1024), 16, off, 0, 0, 0);
  // region 3 (LDS 2 KiB): asm imm 0;  region 4 (LDS 3 KiB): asm imm 192, soffset 128;  region 5 (LDS 90 KiB): asm imm 64
  dma16_asm<0>(rsv, base + 2048, off, 0);
  dma16_asm<192>(rsv, base + 3072, off, 128);
  dma16_asm<64>(rsv, base + 90 * 1024, off, 0);
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  const int regs[6] = {0, 1024, 80 * 1024, 2048, 3072, 90 * 1024};
  for (int r = 0; r < 6; ++r)
    for (int q = 0; q < 4; ++q) out[(r * 64 + lane) * 4 + q] = reinterpret_cast<unsigned*>(lds + regs[r] + lane * 16)[q];
}

int main() {
  const int n = 4096;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0x1000u + i;
  unsigned *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 6 * 64 * 4 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, (unsigned)(n * 4), o);
  std::vector<unsigned> r(6 * 64 * 4);
  if (hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
  const char* name[6] = {"builtin imm 0 @0", "builtin imm 64 @1K", "builtin imm 0 @80K", "asm imm 0 @2K", "asm imm 192 soff 128 @3K", "asm imm 64 @90K"};
  const int shift[6] = {0, 16, 0, 0, 80, 16};      // dwords the in-range lanes are displaced by (imm + soffset) / 4
  for (int reg = 0; reg < 6; ++reg) {
    int ok_in = 0, zero_oob = 0, sentinel_oob = 0, other = 0;
    for (int lane = 0; lane < 64; ++lane) {
      const unsigned* v = &r[(reg * 64 + lane) * 4];
      if (lane & 1) {
        if (v[0] == 0 && v[1] == 0 && v[2] == 0 && v[3] == 0) ++zero_oob;
        else if (v[0] == 0xDEADBEEFu) ++sentinel_oob;
        else ++other;
      } else {
        ok_in += v[0] == 0x1000u + lane * 4 + shift[reg] && v[3] == 0x1000u + lane * 4 + shift[reg] + 3;
      }
    }
    printf("%-28s in-range lanes correct %2d/32   out-of-range lanes: zero %2d  untouched %2d  other %2d\n", name[reg], ok_in, zero_oob, sentinel_oob, other);
  }
  return 0;
}
