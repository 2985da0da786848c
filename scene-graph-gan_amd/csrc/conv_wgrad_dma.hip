// Conv2DBackpropFilter on PRE-SPLIT operands, staged by LDS-DMA (gfx950):
//   dW[kh][kw][ci][co] = sum_{b,y,x} x[b, y+kh-1, x+kw-1, ci] * dy[b, y, x, co]
// Reference: autodiff of tf.layers.conv2d(kernel_size=3 | 5, padding="same") (architectures/generator_with_attention.py:41-57)
// under optimizer.minimize (train.py:265-266).
//
// conv_wgrad_halo3_kernel splits BOTH operands from f32 into two fp16 planes in every launch - on the issue port of the waves that
// feed the matrix pipe: without that arithmetic it runs 10 % faster, without any staging work 33 % faster
// (profiles/r04_wgrad_staging_ablation.log).  When x and dy come pre-split from the LayerNorm kernels (split16.h: every aligned
// 32-channel group = 64 B of leading + 64 B of residual fp16 pieces) the staging is a pure copy, and `buffer_load ... lds` does it
// without a register or a vector instruction:
//   * one 8-wave workgroup per CU owns a (64 input) x (32 NT output) channel tile and a contiguous range of 8x8 pixel blocks;
//     NT = 4: waves = 2 x 4 tiles of 32 x 32 channels, every wave contracts the block's 64 pixels for all nine taps (144
//     accumulator registers); NT = 2 (64-column layers): 2 x 2 tiles x 2 halves of the block's pixels;
//   * per stage (one block) 62 (46) DMA instructions of 1 KiB move the 10 x 10 patch and the 8 x 8 dy block into the LDS image the
//     transposing reads of conv_wgrad_halo3_kernel expect ([plane][32-channel group][pixel slot][64 B]): lane l of an instruction
//     fetches 16 B (8 channels of one piece plane) of pixel slot 16 q + (l >> 2); halo pixels outside the image and the
//     padding slots are out-of-range offsets (the DMA writes zeros);
//   * two LDS buffers: the DMAs of block s + 1 are issued before the MFMAs of block s and waited for (vmcnt(0)) after them; ONE raw
//     s_barrier per block; the waves never split, convert or write LDS, and hold no staging registers.
// Partial sums go to f32 slabs [slab][taps][Cin][Cout] summed in fixed order by slab_reduce_kernel (deterministic), as before.
#include "split16.h"
#include "conv_halo.h"
#include <type_traits>

typedef short wd_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned wd_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* wd_lds_ptr;

// One LDS-DMA instruction (64 lanes x 16 B -> 1 KiB of LDS at `lds_addr`) as INLINE ASSEMBLY: the compiler's wait-count pass treats
// a `buffer_load ... lds` it knows about as a pending write to ALL of LDS and puts s_waitcnt vmcnt(0) in front of the next LDS read
// of the wave - here the MFMA operand reads of the OTHER buffer, i.e. the whole DMA latency would be exposed in every stage (seen in
// the ISA of the builtin form).  Invisible to that pass, the DMAs are waited for by this file's own s_waitcnt at the end of a stage.
typedef int wd_v4i __attribute__((ext_vector_type(4)));
// The byte offset that selects the (channel group, piece plane) image goes into the SCALAR offset: the instruction's immediate offset
// field is added to the LDS address as well (scripts/ubench/dma_oob.hip: with offset:64 the data lands 64 bytes further in LDS).
// Lanes whose vector offset is out of range write ZEROS (same probe) - the halo pixels outside the image and the padding slots.
__device__ __forceinline__ void wd_dma16(wd_v4i rs, unsigned lds_addr, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
               : "memory");      // (M0 is a RESERVED register of this target: the compiler keeps no value in it across statements and warns
                                 //  about it in a clobber list (-Winline-asm); that this kernel has no other M0 user - movrel indexing,
                                 //  a readlane lane select, the LDS-DMA builtin - is verified on the linked code objects by
                                 //  build.check_m0_users)
}
__device__ __forceinline__ wd_v4i wd_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  return wd_v4i{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ unsigned wd_lds_addr(const unsigned char* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const unsigned char*)p;
}

__device__ __forceinline__ wd_u32x2 wd_tr16(const unsigned char* p) {
  return __builtin_bit_cast(wd_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wd_s16x4*)p));
}

struct WgradDmaParams {
  const void* x;          // pre-split [B, Hx, Wx, C]
  const void* dy;         // pre-split [B, H, W, N]
  float* slabs;
  const float* amax_x;    // the bound words the producers scaled with
  const float* amax_dy;
  int B, H, W, C, N;      // dy grid; C = Cin, N = Cout
  int Hx, Wx, sxy, cy, cx, a0y, a0x, kh0, kw0, kstep, KWt, taps_total;      // as WgradHaloParams (parity classes of a stride-2 kernel)
  int bh, bw, nblk;
  int pairs_n;            // Cout tiles
  int stages;             // blocks per workgroup
  unsigned x_bytes, dy_bytes;
};

#define WD_XSUB 8192                 // 128 patch slots (120 used: 10 rows of pitch 12) x 64 B: one 32-channel group of one plane
#define WD_XPLANE (2 * WD_XSUB)
#define WD_DSUB 4096                 // 64 pixels x 64 B

template <int NT, int NKH, int NKW>
__global__ __launch_bounds__(512, 2) void conv_wgrad_dma_kernel(WgradDmaParams p) {
  static_assert(NT == 4 || NT == 2, "64 x 128 or 64 x 64 channel tiles");
  constexpr int NTAP = NKH * NKW;
  constexpr int DPLANE = NT * WD_DSUB;
  constexpr int BUF = 2 * WD_XPLANE + 2 * DPLANE;        // one stage: x planes, then dy planes
  constexpr int KSW = NT == 4 ? 4 : 2;                   // 16-pixel MFMA steps per wave and block
  constexpr int SPW = NT == 4 ? 1 : 2;                   // slabs per workgroup (waves that split the pixels of a tile)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, pair = blockIdx.y;
  const int c0 = (pair / p.pairs_n) * 64, n0 = (pair % p.pairs_n) * 32 * NT;
  const int ci_t = NT == 4 ? wave >> 2 : (wave >> 1) & 1;
  const int co_t = NT == 4 ? wave & 3 : wave & 1;
  const int kg = NT == 4 ? 0 : wave >> 2;                // pixel half of the block (NT = 2)
  const int ks0 = 2 * kg;

  const wd_v4i rs_x = wd_rsrc(p.x, p.x_bytes), rs_dy = wd_rsrc(p.dy, p.dy_bytes);
  const int ea = scale_exp_from_amax(*p.amax_x), eb = scale_exp_from_amax(*p.amax_dy);

  // ---- DMA plan of this lane: x instruction q = wave covers patch slots 16 q .. 16 q + 15 for the four (group, plane) images; dy
  // instruction q = wave & 3 covers pixels 16 q .. + 15 for this wave's share of the (group, plane) images --------------------------
  const int piece = lane & 3;
  const int xslot = 16 * wave + (lane >> 2);
  const int xry = xslot / 12, xrx = xslot - xry * 12;
  const bool xvalid = xslot < 120 && xrx < 10;
  const unsigned xrel = (unsigned)(((xry * p.sxy * p.Wx + xrx * p.sxy) * p.C) * 4 + piece * 16);
  const int xbits = (xry == 0) | ((xry == 9) << 1) | ((xrx == 0) << 2) | ((xrx == 9) << 3);
  const int dq = wave & 3;
  const int dpx = 16 * dq + (lane >> 2);
  const unsigned drel = (unsigned)((((dpx >> 3) * p.W + (dpx & 7)) * p.N) * 4 + piece * 16);
  const int dgrp = wave >> 2;                            // NT = 4: output groups 2 dgrp, 2 dgrp + 1; NT = 2: group dgrp

  const int bpi = p.bh * p.bw;
  const int blk_begin = split * p.stages;
  auto issue = [&](int s, unsigned char* buf) __attribute__((always_inline)) {
    const int beta = blk_begin + s;
    const bool dead = (s >= p.stages) | (beta >= p.nblk);
    const int b = beta / bpi, rem = beta - b * bpi;
    const int by = rem / p.bw, bx = rem - by * p.bw;
    const unsigned xbase = (unsigned)((((b * p.Hx + (by * 8 - 1) * p.sxy + p.cy) * p.Wx + (bx * 8 - 1) * p.sxy + p.cx) * p.C + c0) * 4);
    const unsigned dbase = (unsigned)((((b * p.H + by * 8) * p.W + bx * 8) * p.N + n0) * 4);
    const int bbits = (by == 0) | ((by == p.bh - 1) << 1) | ((bx == 0) << 2) | ((bx == p.bw - 1) << 3);
    const unsigned xo = (dead | !xvalid | ((xbits & bbits) != 0)) ? SGG_OOB : xbase + xrel;
    const unsigned dof = dead ? SGG_OOB : dbase + drel;
    const unsigned xd = wd_lds_addr(buf) + wave * 1024;
    // (group, plane) images of the patch: byte offset of the piece in HBM = group * 128 + plane * 64
    wd_dma16(rs_x, xd, xo, 0);
    wd_dma16(rs_x, xd + WD_XSUB, xo, 128);
    wd_dma16(rs_x, xd + WD_XPLANE, xo, 64);
    wd_dma16(rs_x, xd + WD_XPLANE + WD_XSUB, xo, 192);
    unsigned dd = wd_lds_addr(buf) + 2 * WD_XPLANE + dq * 1024;
    if constexpr (NT == 4) {
      const int soff = dgrp * 256;
      dd += dgrp * 2 * WD_DSUB;
      wd_dma16(rs_dy, dd, dof, soff);
      wd_dma16(rs_dy, dd + WD_DSUB, dof, soff + 128);
      wd_dma16(rs_dy, dd + DPLANE, dof, soff + 64);
      wd_dma16(rs_dy, dd + DPLANE + WD_DSUB, dof, soff + 192);
    } else {
      const int soff = dgrp * 128;
      dd += dgrp * WD_DSUB;
      wd_dma16(rs_dy, dd, dof, soff);
      wd_dma16(rs_dy, dd + DPLANE, dof, soff + 64);
    }
  };

  // ---- transposing-read lane roles (conv_wgrad_halo3_kernel): 16-lane group g -> k half (g >> 1), channel half (g & 1); lane
  // 4 q + pch of the group addresses pixel q (lo) / q + 4 (hi) of the k half's row and channels 4 pch .. 4 pch + 3 ----------------------
  const int g = lane >> 4, q = (lane >> 2) & 3, pch = lane & 3;
  const int choff = ((g & 1) * 2 + (pch >> 1)) * 16 + (pch & 1) * 8;
  const int a_off = ci_t * WD_XSUB + ((2 * ks0 + (g >> 1) + p.a0y + 1) * 12 + q + p.a0x + 1) * 64 + choff;
  const int b_off = 2 * WD_XPLANE + co_t * WD_DSUB + ((2 * ks0 + (g >> 1)) * 8 + q) * 64 + choff;

  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto compute = [&](const unsigned char* buf) __attribute__((always_inline)) {
    const unsigned char* a_base = buf + a_off;
    const unsigned char* b_base = buf + b_off;
#pragma unroll
    for (int ksi = 0; ksi < KSW; ++ksi) {
      u32x4 b[2];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const wd_u32x2 lo = wd_tr16(b_base + (2 * ksi * 8) * 64 + pp * DPLANE);
        const wd_u32x2 hi = wd_tr16(b_base + (2 * ksi * 8 + 4) * 64 + pp * DPLANE);
        b[pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap) {
        const int kh = tap / NKW, kw = tap % NKW;
        u32x4 a[2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const wd_u32x2 lo = wd_tr16(a_base + ((2 * ksi + kh) * 12 + kw) * 64 + pp * WD_XPLANE);
          const wd_u32x2 hi = wd_tr16(a_base + ((2 * ksi + kh) * 12 + kw + 4) * 64 + pp * WD_XPLANE);
          a[pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        f32x16 d = acc[tap];
        d = mfma16<true>(a[1], b[0], d);
        d = mfma16<true>(a[0], b[1], d);
        d = mfma16<true>(a[0], b[0], d);
        acc[tap] = d;
      }
    }
  };

  issue(0, lds);
  __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0) lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < p.stages; ++s) {
    unsigned char* cur = lds + (s & 1) * BUF;
    issue(s + 1, lds + ((s + 1) & 1) * BUF);  // (past the last block: out-of-range offsets, zeros, no traffic)
    __builtin_amdgcn_sched_barrier(0);
    SGG_PRIO_HI();
    compute(cur);
    SGG_PRIO_LO();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0070);      // this wave's DMAs of block s + 1 have landed; its LDS reads of block s are done
    __builtin_amdgcn_s_barrier();
  }

  // ---- partial slab: [slab][tap][Cin][Cout] -------------------------------------------------------------------
  float* o = p.slabs + (size_t)(split * SPW + kg) * p.taps_total * p.C * p.N;
#pragma unroll
  for (int tap = 0; tap < NTAP; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = c0 + ci_t * 32 + acc_row(r, lane);
      const int co = n0 + co_t * 32 + acc_col(lane);
      const int ktap = (p.kh0 + p.kstep * (tap / NKW)) * p.KWt + p.kw0 + p.kstep * (tap % NKW);
      o[((size_t)ktap * p.C + ci) * p.N + co] = ldexpf(ldexpf(acc[tap][r], -ea), -eb);
    }
}

// ---- row bands (grids that 8x8 blocks do not tile: 28x28, 14x14 - conv3_5, `downsampled`) -----------------------------------------
// Geometry of conv_wgrad_halo3_kernel's GEO 1: a block is R full-width rows of one image (R * W <= 112 pixels = seven 16-pixel MFMA
// steps), the patch is (R + 2) rows of pitch W + 1 - slot 0 of a row is the zero column that serves as right halo of the row above
// and left halo of this one.  64 x 64 channel tiles, the eight waves = 2 x 2 tiles x 2 groups of k-steps (4 + 3); 147 KB of LDS
// (two buffers of 176 patch slots + 112 dy pixels, 2 groups x 2 planes each).
struct WgradDmaRbParams {
  WgradDmaParams d;
  int R, pc, xslots, npx;
  unsigned magic_w, magic_pc;       // ceil(2^32 / W), ceil(2^32 / pc)
};
#define WDR_XSLOTS 176
#define WDR_XSUB (WDR_XSLOTS * 64)
#define WDR_XPLANE (2 * WDR_XSUB)
#define WDR_DPX 112
#define WDR_DSUB (WDR_DPX * 64)
#define WDR_DPLANE (2 * WDR_DSUB)
#define WDR_BUF (2 * WDR_XPLANE + 2 * WDR_DPLANE)

template <int NKH, int NKW>
__global__ __launch_bounds__(512, 2) void conv_wgrad_dma_rb_kernel(WgradDmaRbParams pr) {
  const WgradDmaParams& p = pr.d;
  constexpr int NTAP = NKH * NKW;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * WDR_BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, pair = blockIdx.y;
  const int c0 = (pair / p.pairs_n) * 64, n0 = (pair % p.pairs_n) * 64;
  const int ci_t = (wave >> 1) & 1, co_t = wave & 1, kg = wave >> 2;
  const int ks_begin = kg ? 4 : 0, ks_end = kg ? 7 : 4;          // k-steps of this wave
  const wd_v4i rs_x = wd_rsrc(p.x, p.x_bytes), rs_dy = wd_rsrc(p.dy, p.dy_bytes);
  const int ea = scale_exp_from_amax(*p.amax_x), eb = scale_exp_from_amax(*p.amax_dy);

  // ---- DMA plan: x instructions q = wave and wave + 8 (11 of them cover the 176 slots), each for the four (group, plane) images;
  // dy instruction q = wave (7 cover the 112 pixels), for the two groups x two planes -----------------------------------------------
  const int piece = lane & 3;
  unsigned xrel[2];
  int xmeta[2];                    // bits 0..7 patch row, 8 zero column, 9 valid
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int slot = 16 * (wave + 8 * k) + (lane >> 2);
    const int ry = (int)__umulhi((unsigned)slot, pr.magic_pc), rx = slot - ry * pr.pc;
    xrel[k] = (unsigned)(((ry * p.sxy * p.Wx + rx * p.sxy) * p.C) * 4 + piece * 16);
    xmeta[k] = ry | ((rx == 0) << 8) | ((int)(slot < pr.xslots && wave + 8 * k < 11) << 9);
  }
  const int dpx = 16 * wave + (lane >> 2);
  const unsigned drel = (unsigned)((dpx * p.N) * 4 + piece * 16);

  const int blk_begin = split * p.stages;
  auto issue = [&](int s, unsigned char* buf) __attribute__((always_inline)) {
    const int beta = blk_begin + s;
    const bool dead = (s >= p.stages) | (beta >= p.nblk);
    const int b = beta / p.bh, band = beta - b * p.bh;
    const int r0 = band * pr.R;
    const int rows_left = p.H - r0;
    const int npx_valid = (rows_left < pr.R ? rows_left : pr.R) * p.W;
    const unsigned xbase = (unsigned)((((b * p.Hx + (r0 - 1) * p.sxy + p.cy) * p.Wx - p.sxy + p.cx) * p.C + c0) * 4);
    const unsigned dbase = (unsigned)((((b * p.H + r0) * p.W) * p.N + n0) * 4);
    const unsigned la = wd_lds_addr(buf);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (wave + 8 * k < 11) {        // (uniform)
        const int ry = xmeta[k] & 255;
        const bool bad = dead | !((xmeta[k] >> 9) & 1) | ((xmeta[k] >> 8) & 1) | ((ry == 0) & (r0 == 0)) | (ry > rows_left);
        const unsigned xo = bad ? SGG_OOB : xbase + xrel[k];
        const unsigned xd = la + (wave + 8 * k) * 1024;
        wd_dma16(rs_x, xd, xo, 0);
        wd_dma16(rs_x, xd + WDR_XSUB, xo, 128);
        wd_dma16(rs_x, xd + WDR_XPLANE, xo, 64);
        wd_dma16(rs_x, xd + WDR_XPLANE + WDR_XSUB, xo, 192);
      }
    }
    if (wave < 7) {
      const unsigned dof = (dead | (dpx >= npx_valid)) ? SGG_OOB : dbase + drel;
      const unsigned dd = la + 2 * WDR_XPLANE + wave * 1024;
      wd_dma16(rs_dy, dd, dof, 0);
      wd_dma16(rs_dy, dd + WDR_DSUB, dof, 128);
      wd_dma16(rs_dy, dd + WDR_DPLANE, dof, 64);
      wd_dma16(rs_dy, dd + WDR_DPLANE + WDR_DSUB, dof, 192);
    }
  };

  const int g = lane >> 4, q = (lane >> 2) & 3, pch = lane & 3;
  const int choff = ((g & 1) * 2 + (pch >> 1)) * 16 + (pch & 1) * 8;
  const int a_off = ci_t * WDR_XSUB + ((p.a0y + 1) * pr.pc + p.a0x + 1) * 64 + choff;
  const int b_off = 2 * WDR_XPLANE + co_t * WDR_DSUB + ((g >> 1) * 8 + q) * 64 + choff;
  auto rb_slot = [&](int pid) __attribute__((always_inline)) {      // byte offset of pixel pid's patch slot (past the band: clamped, dy is zero there)
    pid = pid < pr.npx ? pid : pr.npx - 1;
    const int row = (int)__umulhi((unsigned)pid, pr.magic_w);
    return (row * pr.pc + (pid - row * p.W)) * 64;
  };

  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto compute = [&](const unsigned char* buf) __attribute__((always_inline)) {
    const unsigned char* a_base = buf + a_off;
    const unsigned char* b_base = buf + b_off;
    const int pc64 = pr.pc * 64;
    // (not unrolled over the k-steps: the per-step patch addresses are run-time values; unrolled they are hoisted and spilled)
#pragma unroll 1
    for (int ksi = ks_begin; ksi < ks_end; ++ksi) {
      u32x4 b[2];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const wd_u32x2 lo = wd_tr16(b_base + ksi * 1024 + pp * WDR_DPLANE);
        const wd_u32x2 hi = wd_tr16(b_base + ksi * 1024 + 256 + pp * WDR_DPLANE);
        b[pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
      const int pid = 16 * ksi + 8 * (g >> 1) + q;
      const unsigned char* alo = a_base + rb_slot(pid);
      const unsigned char* ahi = a_base + rb_slot(pid + 4);
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap) {
        const int kh = tap / NKW, kw = tap % NKW;
        u32x4 a[2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const wd_u32x2 lo = wd_tr16(alo + kh * pc64 + kw * 64 + pp * WDR_XPLANE);
          const wd_u32x2 hi = wd_tr16(ahi + kh * pc64 + kw * 64 + pp * WDR_XPLANE);
          a[pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        f32x16 d = acc[tap];
        d = mfma16<true>(a[1], b[0], d);
        d = mfma16<true>(a[0], b[1], d);
        d = mfma16<true>(a[0], b[0], d);
        acc[tap] = d;
      }
    }
  };

  issue(0, lds);
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < p.stages; ++s) {
    unsigned char* cur = lds + (s & 1) * WDR_BUF;
    issue(s + 1, lds + ((s + 1) & 1) * WDR_BUF);
    __builtin_amdgcn_sched_barrier(0);
    SGG_PRIO_HI();
    compute(cur);
    SGG_PRIO_LO();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
  }

  float* o = p.slabs + (size_t)(split * 2 + kg) * p.taps_total * p.C * p.N;
#pragma unroll
  for (int tap = 0; tap < NTAP; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = c0 + ci_t * 32 + acc_row(r, lane);
      const int co = n0 + co_t * 32 + acc_col(lane);
      const int ktap = (p.kh0 + p.kstep * (tap / NKW)) * p.KWt + p.kw0 + p.kstep * (tap % NKW);
      o[((size_t)ktap * p.C + ci) * p.N + co] = ldexpf(ldexpf(acc[tap][r], -ea), -eb);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------
// H, W: the dy grid (both divisible by 8); 3x3 stride 1 or 5x5 stride 2 with an even x grid; Cin, Cout % 64 == 0.
int sgg_wgrad_dma_plan(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, WgradDmaPlan* pl) {
  const bool k3 = KH == 3 && KW == 3 && stride == 1, k5 = KH == 5 && KW == 5 && stride == 2;
  if (!((k3 || k5) && B > 0 && H > 0 && W > 0 && Cin % 64 == 0 && Cout % 64 == 0)) return 0;
  if ((size_t)B * H * W * stride * stride * Cin * sizeof(float) >= 0x80000000ull || (size_t)B * H * W * Cout * sizeof(float) >= 0x80000000ull)
    return 0;
  pl->geo = 0; pl->R = 8; pl->pc = 12; pl->xslots = 120;
  if (H % 8 != 0 || W % 8 != 0) {
    // row bands (conv_wgrad_halo.hip: sgg_wgrad_halo_plan): R full-width rows, R * W <= 112 pixels, (R + 2) * (W + 1) + 1 <= 176 slots
    int R = 112 / W;
    if (R > H) R = H;
    while (R > 0 && (R + 2) * (W + 1) + 1 > 176) --R;
    if (R < 1) return 0;
    pl->geo = 1; pl->R = R; pl->pc = W + 1; pl->xslots = (R + 2) * (W + 1) + 1;
    pl->nt = 2; pl->spw = 2;
    pl->pairs_n = Cout / 64;
    pl->pairs = (Cin / 64) * pl->pairs_n;
    const int nb = B * ((H + R - 1) / R);
    int nsr = 256 / pl->pairs;
    if (nsr > nb / 4) nsr = nb / 4;
    if (nsr < 1) nsr = 1;
    pl->stages = (nb + nsr - 1) / nsr;
    pl->nsplit = (nb + pl->stages - 1) / pl->stages;
    pl->nslabs = pl->nsplit * pl->spw;
    pl->ws_bytes = (size_t)pl->nslabs * KH * KW * Cin * Cout * sizeof(float);
    return 1;
  }
  pl->nt = Cout % 128 == 0 ? 4 : 2;
  pl->spw = pl->nt == 4 ? 1 : 2;
  pl->pairs_n = Cout / (32 * pl->nt);
  pl->pairs = (Cin / 64) * pl->pairs_n;
  const int nblk = B * (H / 8) * (W / 8);
  int ns = 256 / pl->pairs;                       // one workgroup per CU
  if (ns > nblk / 4) ns = nblk / 4;
  if (ns < 1) ns = 1;
  pl->stages = (nblk + ns - 1) / ns;
  pl->nsplit = (nblk + pl->stages - 1) / pl->stages;
  pl->nslabs = pl->nsplit * pl->spw;
  pl->ws_bytes = (size_t)pl->nslabs * KH * KW * Cin * Cout * sizeof(float);
  return 1;
}

template <int NKH, int NKW>
static void wgrad_dma_launch_class(const WgradDmaParams& p, const WgradDmaPlan& pl, hipStream_t st) {
  const dim3 grid(pl.nsplit, pl.pairs);
  if (pl.geo == 1) {
    WgradDmaRbParams pr;
    pr.d = p;
    pr.d.bh = (p.H + pl.R - 1) / pl.R; pr.d.bw = 1; pr.d.nblk = p.B * pr.d.bh;
    pr.R = pl.R; pr.pc = pl.pc; pr.xslots = pl.xslots; pr.npx = pl.R * p.W;
    pr.magic_w = (unsigned)((0x100000000ull + p.W - 1) / p.W);
    pr.magic_pc = (unsigned)((0x100000000ull + pl.pc - 1) / pl.pc);
    hipLaunchKernelGGL((conv_wgrad_dma_rb_kernel<NKH, NKW>), grid, dim3(512), 0, st, pr);
    return;
  }
  if (pl.nt == 4) hipLaunchKernelGGL((conv_wgrad_dma_kernel<4, NKH, NKW>), grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL((conv_wgrad_dma_kernel<2, NKH, NKW>), grid, dim3(512), 0, st, p);
}

void sgg_wgrad_dma_launch(const void* x, const void* dy, float* slabs, int B, int H, int W, int Cin, int Cout, int stride, int pad_t,
                          int pad_l, const float* amax_x, const float* amax_dy, const WgradDmaPlan& pl, hipStream_t st) {
  WgradDmaParams p;
  p.x = x; p.dy = dy; p.slabs = slabs; p.amax_x = amax_x; p.amax_dy = amax_dy;
  p.B = B; p.H = H; p.W = W; p.C = Cin; p.N = Cout; p.bh = H / 8; p.bw = W / 8; p.nblk = B * p.bh * p.bw;
  p.Hx = H * stride; p.Wx = W * stride; p.sxy = stride;
  p.pairs_n = pl.pairs_n; p.stages = pl.stages;
  p.x_bytes = (unsigned)((size_t)B * p.Hx * p.Wx * Cin * sizeof(float));
  p.dy_bytes = (unsigned)((size_t)B * H * W * Cout * sizeof(float));
  if (stride == 1) {
    p.cy = p.cx = 0; p.a0y = p.a0x = -1; p.kh0 = p.kw0 = 0; p.kstep = 1; p.KWt = 3; p.taps_total = 9;
    wgrad_dma_launch_class<3, 3>(p, pl, st);
    return;
  }
  p.kstep = 2; p.KWt = 5; p.taps_total = 25;
  for (int cy = 0; cy < 2; ++cy)
    for (int cx = 0; cx < 2; ++cx) {      // the four parity classes of the taps (conv_wgrad_halo.hip: sgg_wgrad_halo_launch)
      const int kh0 = (pad_t + cy) % 2, kw0 = (pad_l + cx) % 2;
      p.cy = cy; p.cx = cx; p.kh0 = kh0; p.kw0 = kw0;
      p.a0y = (kh0 - pad_t - cy) / 2;
      p.a0x = (kw0 - pad_l - cx) / 2;
      const int nkh = (5 - kh0 + 1) / 2, nkw = (5 - kw0 + 1) / 2;
      if (nkh == 3 && nkw == 3) wgrad_dma_launch_class<3, 3>(p, pl, st);
      else if (nkh == 3) wgrad_dma_launch_class<3, 2>(p, pl, st);
      else if (nkw == 3) wgrad_dma_launch_class<2, 3>(p, pl, st);
      else wgrad_dma_launch_class<2, 2>(p, pl, st);
    }
}
