// Halo-resident Conv2DBackpropFilter for the 3x3 / stride-1 layers in the split 16-bit modes (gfx950):
//   dW[kh][kw][ci][co] = sum_{b,y,x} x[b, y+kh-1, x+kw-1, ci] * dy[b, y, x, co]
// Reference: autodiff of tf.layers.conv2d(kernel_size=3, padding="same") (architectures/generator_with_attention.py:31-57)
// under optimizer.minimize (train.py:265-266).
//
// The per-tap GEMM kernels (conv_wgrad.hip) re-read x and dy for each of the nine taps and, for the 32/64-channel
// layers, run one 32x32 MFMA tile per workgroup and tap (8 FLOP per byte fetched).  Here a workgroup owns a
// (32*CT input) x (32*NT output) channel chunk and a contiguous range of 8x8 pixel blocks; per stage the 10x10 input
// patch (halo included) and the 8x8 dy block are split into two 16-bit planes and written to LDS ONCE, and every wave
// contracts its 32x32 channel tile for all nine taps out of the resident patch: 9 accumulator tiles per wave
// (144 VGPRs), contraction over the block's 64 pixels (four 16-pixel MFMA steps).  Both MFMA operands need 8
// consecutive PIXELS of one channel per lane; they are fetched from the [pixel][channel] planes with the gfx950
// transposing LDS read ds_read_b64_tr_b16, each lane addressing its own pixel, so that a tap shift is an immediate
// offset on one base register (no per-tap address arithmetic at all).  Rows are 64 B (32 channels) and a 32-lane
// phase of the read covers 4 consecutive pixel slots = 256 contiguous bytes: conflict free without a swizzle.
// Partial sums go to f32 slabs [slab][9][Cin][Cout] summed in fixed order by slab_reduce_kernel (deterministic).
//
// GEO = 1 (row bands): 28x28 and 14x14 grids (conv3_5, `downsampled`) are not tiled by 8x8 blocks.  There a block is R FULL-WIDTH
// rows of one image with R * W <= 112 pixels (seven 16-pixel MFMA steps: 4 x 28, 8 x 14), the patch is (R + 2) rows of pitch W + 1:
// both halo columns of a full-width band are zero padding, so one zero slot per row serves as the right halo of row r and the left
// halo of row r + 1.  A lane's pixel of a k-step is pid = 16 ks + 8 (g >> 1) + q (+ 4), its patch slot ((pid / W) * (W + 1) +
// pid % W), tap shifts are run-time slot offsets.  73.5 KB of LDS per workgroup (gfx950 gives a workgroup up to 160 KB).
#include "split16.h"
#include "conv_halo.h"
#include <type_traits>

// Timing-only ablations (wrong results by design; scripts/build_variant_one.sh): SGG_WABL_NOSPLIT = the staged bytes go to LDS as
// they are (what pre-split 16-bit operand planes in HBM would leave of the staging: loads + LDS writes, no arithmetic);
// SGG_WABL_NOSTAGE = nothing is loaded or written after the first stage (MFMAs, LDS reads and barriers only).
#ifndef SGG_WABL_NOSPLIT
#define SGG_WABL_NOSPLIT 0
#endif
#ifndef SGG_WABL_NOSTAGE
#define SGG_WABL_NOSTAGE 0
#endif
typedef short s16x4h __attribute__((ext_vector_type(4)));
typedef unsigned u32x2h __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x2h lds_tr16(const unsigned char* p) {
  return __builtin_bit_cast(u32x2h, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4h*)p));
}

struct WgradHaloParams {
  const float* x;
  const float* dy;
  float* slabs;
  const float* amax_x;
  const float* amax_dy;
  const float* ln_stats;  // LN prologue (optional): x holds the producing layer's pre-LayerNorm output y; see HaloParams
  const float* ln_gamma;
  const float* ln_beta;
  int B;
  int H, W, C, N;         // dy grid (= output grid of the convolution); C = Cin, N = Cout
  int Hx, Wx;             // x grid (= H, W for stride 1; 2H, 2W for the stride-2 classes)
  int sxy;                // stride (1 or 2)
  int cy, cx;             // this launch's parity class: x row = sxy * i + cy for sub-grid row i (0 for stride 1)
  int a0y, a0x;           // first sub-grid offset of the class's taps (-1 or 0): tap (ia, ib) reads sub-grid pixel (oy + a0y + ia, ox + a0x + ib)
  int kh0, kw0, kstep;    // kernel tap of (ia, ib) = (kh0 + kstep*ia, kw0 + kstep*ib)
  int KWt, taps_total;    // kernel width (3 / 5) and KH*KW: slab layout [slab][taps_total][Cin][Cout]
  int bh, bw, nblk;       // 8x8 blocks per image (rows, cols) and in total (GEO 1: bands per image, 1)
  int R, pc, xslots, npx; // GEO 1: rows per band, patch pitch, patch slots, pixels of a full band (R * W)
  unsigned magic_w;       // GEO 1: ceil(2^32 / W) for pid / W (pid < 112)
  unsigned magic_pc;      // GEO 1: ceil(2^32 / pc) for slot / pc (slot < 176)
  int pairs_n;            // Cout chunks
  int stages;             // stages per workgroup (NBS blocks each)
  unsigned x_bytes, dy_bytes;
  int x_s16, dy_s16;      // 1: the operand is a pre-split ("S16") tensor of the LayerNorm kernels (split16.h)
};

// NKH x NKW: taps of the launch (3x3 for the stride-1 kernel; 3x3 / 3x2 / 2x3 / 2x2 for the four parity classes of a 5x5
// stride-2 kernel, each a stride-1 problem on the sub-sampled x grid).
// ONE: single-piece mode (precision 1 / 4): one 16-bit plane per operand, one MFMA per product.
template <int CT, int NT, bool HALF, bool PREF, int NKH, int NKW, bool LNP, int GEO = 0, bool ONE = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_halo3_kernel(WgradHaloParams p) {
  static_assert(GEO == 0 || (CT == 2 && NT == 2 && !LNP), "row bands: 64 x 64 channel chunks only");
  static_assert(!(ONE && LNP), "the LN prologue exists in the two-piece modes only");
  constexpr int NTAP = NKH * NKW;
  constexpr int P = ONE ? 1 : 2;
  constexpr int NBS = (CT == 2) ? 1 : 2;                 // blocks per stage
  constexpr int KSW = GEO ? 7 : ((CT == 1 && NT == 1) ? 2 : 4);      // 16-pixel MFMA steps per wave and block
  constexpr int SPW = 4 / (CT * NT) ;                    // waves that share a channel tile (own slabs)
  constexpr int XSLOTS = GEO ? 176 : 120;                // patch slots of a block (GEO 0: 10 x pitch 12)
  constexpr int XSUB = XSLOTS * 64;                      // bytes of one 32-channel sub-plane of a block's patch
  constexpr int XPLANE = NBS * CT * XSUB;
  constexpr int DPX = GEO ? 112 : 64;                    // dy pixels of a block
  constexpr int DSUB = DPX * 64;
  constexpr int DPLANE = NBS * NT * DSUB;
  constexpr int XI = (GEO ? XSLOTS * 4 : 400) * CT, XP = (XI + 255) / 256;    // x staging items (patch pixel, 8 channels) per block, passes
  constexpr int DI = DPX * 4 * NT, DP = (DI + 255) / 256;
  constexpr int NPASS = NBS * (XP + DP);
  // GEO 1 with nine taps: 144 accumulator registers leave room for the x prefetch only; dy is loaded after the MFMAs of a stage
  // (the other resident workgroup computes meanwhile)
  constexpr bool DY_LATE = GEO == 1 && NTAP > 6;
  __shared__ __attribute__((aligned(16))) unsigned char lds[P * (XPLANE + DPLANE)];
  __shared__ __attribute__((aligned(16))) float lnp_s[LNP ? 64 * CT : 4];   // gamma, beta of this workgroup's 32*CT input channels
  unsigned char* x_s = lds;
  unsigned char* d_s = lds + P * XPLANE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x, pair = blockIdx.y;
  const int c0 = (pair / p.pairs_n) * 32 * CT, n0 = (pair % p.pairs_n) * 32 * NT;
  // wave roles
  int ci_t, co_t, wblk, ks0, sw;
  if constexpr (CT == 2) { ci_t = wave >> 1; co_t = wave & 1; wblk = 0; ks0 = 0; sw = 0; }
  else if constexpr (NT == 2) { ci_t = 0; co_t = wave & 1; wblk = wave >> 1; ks0 = 0; sw = wave >> 1; }
  else { ci_t = 0; co_t = 0; wblk = wave >> 1; ks0 = 2 * (wave & 1); sw = wave; }

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
  int ea = 0, eb = 0;
  if constexpr (HALF) {
    ea = scale_exp_from_amax(*p.amax_x);
    eb = scale_exp_from_amax(*p.amax_dy);
  }
  const float sa = ldexpf(1.f, ea), sb = ldexpf(1.f, eb);

  // ---- staging plan: pass -> (kind, block); item -> offset relative to the block origin, border bits, LDS offset ----
  // GEO 1: the plan of an item is recomputed where it is used (20 registers less than keeping it)
  //   meta bits 0..19 LDS byte offset, 20..24 patch row (x) / 20..26 pixel (dy), 27 zero column, 28 valid
  auto rb_plan = [&](int jj, unsigned& rel, int& meta) __attribute__((always_inline)) {
    if (jj < XP) {
      const int i = tid + 256 * jj;
      const int csub = i >= XSLOTS * 4 ? 1 : 0, r = i - csub * (XSLOTS * 4);
      const int slot = r >> 2, c4 = r & 3;
      const int ry = (int)__umulhi((unsigned)slot, p.magic_pc), rx = slot - ry * p.pc;
      rel = (unsigned)(((ry * p.sxy * p.Wx + rx * p.sxy) * p.C + csub * 32 + c4 * 8) * 4);
      meta = (csub * XSUB + slot * 64 + c4 * 16) | (ry << 20) | ((rx == 0) << 27) | ((i < XI && slot < p.xslots) << 28);
    } else {
      const int i = tid + 256 * (jj - XP);
      const int nsub = i >= DPX * 4 ? 1 : 0, r = i - nsub * (DPX * 4);
      const int px = r >> 2, c4 = r & 3;
      rel = (unsigned)((px * p.N + nsub * 32 + c4 * 8) * 4);
      meta = (nsub * DSUB + px * 64 + c4 * 16) | (px << 20) | ((i < DI) << 28);
    }
  };
  unsigned it_rel[GEO ? 1 : NPASS];
  int it_lds[GEO ? 1 : NPASS];        // bits 0..19 LDS byte offset inside a plane, bits 20..23 border bits, bit 24 valid
#pragma unroll
  for (int j = 0; j < (GEO ? 0 : NPASS); ++j) {
    const int blk = j / (XP + DP), jj = j % (XP + DP);
    if (jj < XP) {
      const int i = tid + 256 * jj;
      const int csub = i / 400, r = i % 400;
      const int px = r >> 2, c4 = r & 3;
      const int ry = px / 10, rx = px % 10;
      it_rel[j] = (unsigned)(((ry * p.sxy * p.Wx + rx * p.sxy) * p.C + csub * 32 + c4 * 8) * 4);
      const int bits = (ry == 0) | ((ry == 9) << 1) | ((rx == 0) << 2) | ((rx == 9) << 3);
      it_lds[j] = ((blk * CT + csub) * XSUB + (ry * 12 + rx) * 64 + c4 * 16) | (bits << 20) | ((i < XI) << 24);
    } else {
      const int i = tid + 256 * (jj - XP);
      const int nsub = i >> 8, r = i & 255;
      const int px = r >> 2, c4 = r & 3;
      it_rel[j] = (unsigned)((((px >> 3) * p.W + (px & 7)) * p.N + nsub * 32 + c4 * 8) * 4);
      it_lds[j] = ((blk * NT + nsub) * DSUB + px * 64 + c4 * 16) | (1 << 24);
    }
  }

  // block coordinates of the NBS staged blocks, advanced by NBS per stage
  const int bpi = p.bh * p.bw;
  const int blk_begin = split * p.stages * NBS;
  int cb[NBS], cy[NBS], cx[NBS];
#pragma unroll
  for (int j = 0; j < NBS; ++j) {
    const int beta = blk_begin + j;
    cb[j] = beta / bpi;
    const int rem = beta % bpi;
    cy[j] = rem / p.bw;
    cx[j] = rem % p.bw;
  }
  int next_beta = blk_begin;       // first block of the stage the next stage_load fetches
  if constexpr (LNP) {
    if (tid < 32 * CT) {
      lnp_s[tid] = p.ln_gamma[c0 + tid];
      lnp_s[32 * CT + tid] = p.ln_beta[c0 + tid];
    }
    __syncthreads();
  }
  float ld_mu[NBS], ld_rs[NBS];    // LN prologue: (mean, rstd) of the staged blocks' samples, padding items of the patch in flight
  int ld_bad = 0;

  f32x4 pre[DY_LATE ? XP : NPASS][2];
  // GEO 1 staging (one band per stage).  part 0: x patch and dy band; 1: x patch only; 2: dy band only (into pre[0..DP)).  The
  // block cursor advances once both parts of a band have been issued.
  auto stage_load_rb = [&](auto part_c) __attribute__((always_inline)) {
    constexpr int part = decltype(part_c)::value;
    const bool dead = (next_beta >= p.nblk) | (next_beta >= blk_begin + p.stages);
    const int r0 = cy[0] * p.R;
    const int rows_left = p.H - r0;                                  // sub-grid rows from r0 to the end of the image
    const int npx_valid = (rows_left < p.R ? rows_left : p.R) * p.W;
    const unsigned xbase = (unsigned)((((cb[0] * p.Hx + (r0 - 1) * p.sxy + p.cy) * p.Wx - p.sxy + p.cx) * p.C + c0) * 4);
    const unsigned dbase = (unsigned)((((cb[0] * p.H + r0) * p.W) * p.N + n0) * 4);
#pragma unroll
    for (int jj = 0; jj < XP + DP; ++jj) {
      const bool isx = jj < XP;
      if ((part == 1 && !isx) || (part == 2 && isx)) continue;
      unsigned rel;
      int meta;
      rb_plan(jj, rel, meta);
      bool bad = dead | !((meta >> 28) & 1);
      if (isx) {
        const int ry = (meta >> 20) & 31;
        bad |= ((meta >> 27) & 1) | ((ry == 0) & (r0 == 0)) | (ry > rows_left);
      } else {
        bad |= ((meta >> 20) & 127) >= npx_valid;
      }
      const unsigned off = bad ? SGG_OOB : (isx ? xbase : dbase) + rel;
      const int slot = (DY_LATE && !isx) ? jj - XP : jj;
      const int fmt = isx ? p.x_s16 : p.dy_s16;
      const unsigned o0 = stage_off0(off, fmt), o1 = stage_off1(o0, fmt);
      if (isx) {
        pre[slot][0] = buf_load4_aux<SGG_WGRAD_LOAD_AUX>(rs_x, o0);
        pre[slot][1] = buf_load4_aux<SGG_WGRAD_LOAD_AUX>(rs_x, o1);
      } else {
        pre[slot][0] = buf_load4_aux<SGG_WGRAD_LOAD_AUX>(rs_dy, o0);
        pre[slot][1] = buf_load4_aux<SGG_WGRAD_LOAD_AUX>(rs_dy, o1);
      }
    }
    if constexpr (part != 1) {
      if (++cy[0] >= p.bh) { cy[0] = 0; ++cb[0]; }
      next_beta += 1;
    }
  };
  auto stage_write_rb = [&](auto part_c) __attribute__((always_inline)) {
    constexpr int part = decltype(part_c)::value;
#pragma unroll
    for (int jj = 0; jj < XP + DP; ++jj) {
      const bool isx = jj < XP;
      if ((part == 1 && !isx) || (part == 2 && isx)) continue;
      const int slot = (DY_LATE && !isx) ? jj - XP : jj;
      u32x4 pl[P];
      if constexpr (HALF) stage_planes<P, HALF>(pre[slot][0], pre[slot][1], isx ? sa : sb, isx ? p.x_s16 : p.dy_s16, pl);
      else split8<P, HALF>(pre[slot][0], pre[slot][1], isx ? sa : sb, pl);
      unsigned rel;
      int meta;
      rb_plan(jj, rel, meta);
      if ((meta >> 28) & 1) {
        unsigned char* dst = (isx ? x_s : d_s) + (meta & 0xfffff);
#pragma unroll
        for (int pp = 0; pp < P; ++pp) *reinterpret_cast<u32x4*>(dst + pp * (isx ? XPLANE : DPLANE)) = pl[pp];
      }
    }
  };
  auto stage_load = [&]() {
#pragma unroll
    for (int blk = 0; blk < NBS; ++blk) {
      const bool dead = (next_beta + blk >= p.nblk) | (next_beta + blk >= blk_begin + p.stages * NBS);
      const int by = cy[blk], bx = cx[blk];
      const unsigned xbase = (unsigned)((((cb[blk] * p.Hx + (by * 8 - 1) * p.sxy + p.cy) * p.Wx + (bx * 8 - 1) * p.sxy + p.cx) * p.C + c0) * 4);
      const unsigned dbase = (unsigned)((((cb[blk] * p.H + by * 8) * p.W + bx * 8) * p.N + n0) * 4);
      const int bbits = (by == 0) | ((by == p.bh - 1) << 1) | ((bx == 0) << 2) | ((bx == p.bw - 1) << 3);
      if constexpr (LNP) {
        const int b = cb[blk] < p.B ? cb[blk] : p.B - 1;
        ld_mu[blk] = p.ln_stats[2 * b];
        ld_rs[blk] = p.ln_stats[2 * b + 1];
        if (blk == 0) ld_bad = 0;
      }
#pragma unroll
      for (int jj = 0; jj < XP + DP; ++jj) {
        const int j = blk * (XP + DP) + jj;
        const bool isx = jj < XP;
        const bool bad = dead | !((it_lds[j] >> 24) & 1) | (isx && (((it_lds[j] >> 20) & 15 & bbits) != 0));
        if constexpr (LNP) ld_bad |= (int)bad << j;
        const unsigned off = bad ? SGG_OOB : (isx ? xbase : dbase) + it_rel[j];
        const int fmt = isx ? (LNP ? 0 : p.x_s16) : p.dy_s16;
        const unsigned o0 = stage_off0(off, fmt), o1 = stage_off1(o0, fmt);
        if (isx) {
          pre[j][0] = buf_load4_aux<SGG_WGRAD_LOAD_AUX>(rs_x, o0);
          pre[j][1] = buf_load4_aux<SGG_WGRAD_LOAD_AUX>(rs_x, o1);
        } else {
          pre[j][0] = buf_load4_aux<SGG_WGRAD_LOAD_AUX>(rs_dy, o0);
          pre[j][1] = buf_load4_aux<SGG_WGRAD_LOAD_AUX>(rs_dy, o1);
        }
      }
      // advance this slot to the next stage's block
      cx[blk] += NBS;
      while (cx[blk] >= p.bw) {
        cx[blk] -= p.bw;
        if (++cy[blk] >= p.bh) { cy[blk] = 0; ++cb[blk]; }
      }
    }
    next_beta += NBS;
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const bool isx = (j % (XP + DP)) < XP;
      if constexpr (LNP) {
        if (isx) {
          const int blk = j / (XP + DP), i = tid + 256 * (j % (XP + DP));
          const int cbx = (i / 400) * 32 + ((i % 400) & 3) * 8;
          ln_elu8(pre[j][0], pre[j][1], lnp_s + cbx, lnp_s + 32 * CT + cbx, ld_mu[blk], ld_rs[blk], (ld_bad >> j) & 1);
        }
      }
      u32x4 pl[P];
#if SGG_WABL_NOSPLIT
      pl[0] = __builtin_bit_cast(u32x4, pre[j][0]);
      if constexpr (P == 2) pl[1] = __builtin_bit_cast(u32x4, pre[j][1]);
#else
      if constexpr (HALF) stage_planes<P, HALF>(pre[j][0], pre[j][1], isx ? sa : sb, isx ? (LNP ? 0 : p.x_s16) : p.dy_s16, pl);
      else split8<P, HALF>(pre[j][0], pre[j][1], isx ? sa : sb, pl);
#endif
      if ((it_lds[j] >> 24) & 1) {
        unsigned char* dst = (isx ? x_s : d_s) + (it_lds[j] & 0xfffff);
#pragma unroll
        for (int pp = 0; pp < P; ++pp) *reinterpret_cast<u32x4*>(dst + pp * (isx ? XPLANE : DPLANE)) = pl[pp];
      }
    }
  };

  // transposing-read lane roles (see conv_wgrad_tr_kernel): 16-lane group g -> k half (g >> 1), channel half (g & 1);
  // lane 4q+pch of the group addresses pixel q (lo) / q+4 (hi) of the k half's row and channels 4pch..4pch+3
  const int g = lane >> 4, q = (lane >> 2) & 3, pch = lane & 3;
  const int choff = ((g & 1) * 2 + (pch >> 1)) * 16 + (pch & 1) * 8;
  const unsigned char* a_base = x_s + (wblk * CT + ci_t) * XSUB + (GEO ? ((p.a0y + 1) * p.pc + p.a0x + 1) * 64 : ((2 * ks0 + (g >> 1) + p.a0y + 1) * 12 + q + p.a0x + 1) * 64) + choff;
  const unsigned char* b_base = d_s + (wblk * NT + co_t) * DSUB + ((2 * ks0 + (g >> 1)) * 8 + q) * 64 + choff;
  // GEO 1: byte offset of pixel pid's patch slot relative to a_base (pixels past the band: clamped, their dy is zero)
  auto rb_slot = [&](int pid) __attribute__((always_inline)) {
    pid = pid < p.npx ? pid : p.npx - 1;
    const int row = (int)__umulhi((unsigned)pid, p.magic_w);
    return (row * p.pc + (pid - row * p.W)) * 64;
  };

  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto compute = [&]() __attribute__((always_inline)) {
    if constexpr (GEO == 1) {
      // (not unrolled over the seven k-steps: the per-step patch addresses are run-time values; unrolled they are hoisted and spilled)
      const int pc64 = p.pc * 64;
#pragma unroll 1
      for (int ksi = 0; ksi < KSW; ++ksi) {
        u32x4 b[P];
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
          const u32x2h lo = lds_tr16(b_base + ksi * 1024 + pp * DPLANE);
          const u32x2h hi = lds_tr16(b_base + ksi * 1024 + 256 + pp * DPLANE);
          b[pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        const int pid = 16 * ksi + 8 * (g >> 1) + q;
        const unsigned char* alo = a_base + rb_slot(pid);
        const unsigned char* ahi = a_base + rb_slot(pid + 4);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
          const int kh = tap / NKW, kw = tap % NKW;
          u32x4 a[P];
#pragma unroll
          for (int pp = 0; pp < P; ++pp) {
            const u32x2h lo = lds_tr16(alo + kh * pc64 + kw * 64 + pp * XPLANE);
            const u32x2h hi = lds_tr16(ahi + kh * pc64 + kw * 64 + pp * XPLANE);
            a[pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
          }
          f32x16 d = acc[tap];
          if constexpr (P == 2) {
            d = mfma16<HALF>(a[1], b[0], d);
            d = mfma16<HALF>(a[0], b[1], d);
          }
          d = mfma16<HALF>(a[0], b[0], d);
          acc[tap] = d;
        }
      }
      return;
    }
#pragma unroll
    for (int ksi = 0; ksi < KSW; ++ksi) {
      u32x4 b[P];
#pragma unroll
      for (int pp = 0; pp < P; ++pp) {
        const u32x2h lo = lds_tr16(b_base + (2 * ksi * 8) * 64 + pp * DPLANE);
        const u32x2h hi = lds_tr16(b_base + (2 * ksi * 8 + 4) * 64 + pp * DPLANE);
        b[pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap) {
        const int kh = tap / NKW, kw = tap % NKW;
        u32x4 a[P];
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
          const u32x2h lo = lds_tr16(a_base + ((2 * ksi + kh) * 12 + kw) * 64 + pp * XPLANE);
          const u32x2h hi = lds_tr16(a_base + ((2 * ksi + kh) * 12 + kw + 4) * 64 + pp * XPLANE);
          a[pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        f32x16 d = acc[tap];
        if constexpr (P == 2) {
          d = mfma16<HALF>(a[1], b[0], d);
          d = mfma16<HALF>(a[0], b[1], d);
        }
        d = mfma16<HALF>(a[0], b[0], d);
        acc[tap] = d;
      }
    }
  };

  if constexpr (GEO == 1) {
    // x patch of the next band prefetched under the MFMAs; with late dy its band is loaded and written after them
    auto load1 = [&]() __attribute__((always_inline)) { stage_load_rb(std::integral_constant<int, DY_LATE ? 1 : 0>{}); };
    auto write1 = [&]() __attribute__((always_inline)) {
      if constexpr (DY_LATE) {
        stage_write_rb(std::integral_constant<int, 1>{});
        stage_load_rb(std::integral_constant<int, 2>{});
        stage_write_rb(std::integral_constant<int, 2>{});
      } else {
        stage_write_rb(std::integral_constant<int, 0>{});
      }
    };
    load1();
    write1();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < p.stages; ++s) {
      load1();
      __builtin_amdgcn_sched_barrier(0);
      SGG_PRIO_HI();
      compute();
      SGG_PRIO_LO();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      write1();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
  stage_load();
  stage_write();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < p.stages; ++s) {
#if !SGG_WABL_NOSTAGE
    if constexpr (PREF) stage_load();        // next stage's blocks (out-of-range offsets past the end: zeros, no traffic)
#endif
    __builtin_amdgcn_sched_barrier(0);
    SGG_PRIO_HI();
    compute();
    SGG_PRIO_LO();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#if !SGG_WABL_NOSTAGE
    if constexpr (!PREF) stage_load();
    stage_write();
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  }

  // ---- partial slab: [slab][tap][Cin][Cout] -------------------------------------------------------------------
  float* o = p.slabs + (size_t)(split * SPW + sw) * p.taps_total * p.C * p.N;
#pragma unroll
  for (int tap = 0; tap < NTAP; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = c0 + ci_t * 32 + acc_row(r, lane);
      const int co = n0 + co_t * 32 + acc_col(lane);
      const float v = acc[tap][r];
      const int ktap = (p.kh0 + p.kstep * (tap / NKW)) * p.KWt + p.kw0 + p.kstep * (tap % NKW);
      o[((size_t)ktap * p.C + ci) * p.N + co] = HALF ? ldexpf(ldexpf(v, -ea), -eb) : v;
    }
}

// ---- host ---------------------------------------------------------------------------------------------------
// H, W: the dy grid.  Served: 3x3 stride 1 (x grid = dy grid) and 5x5 stride 2 with an even x grid (= 2H x 2W, SAME pads
// (1, 2)); H % 8 == W % 8 == 0, channels % 32 == 0.
int sgg_wgrad_halo_plan(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, WgradHaloPlan* pl) {
  const bool k3 = KH == 3 && KW == 3 && stride == 1, k5 = KH == 5 && KW == 5 && stride == 2;
  if (!((k3 || k5) && H > 0 && W > 0 && Cin % 32 == 0 && Cout % 32 == 0 && B > 0)) return 0;
  if ((size_t)B * H * W * stride * stride * Cin * sizeof(float) >= 0x80000000ull || (size_t)B * H * W * Cout * sizeof(float) >= 0x80000000ull)
    return 0;
  pl->geo = 0; pl->R = 8; pl->pc = 12; pl->xslots = 120;
  if (H % 8 != 0 || W % 8 != 0) {
    // row bands: R full-width rows with R * W <= 112 pixels and a patch of (R + 2) * (W + 1) + 1 <= 176 slots; 64 x 64 channel chunks
    int R = 112 / W;
    if (R > H) R = H;
    while (R > 0 && (R + 2) * (W + 1) + 1 > 176) --R;
    if (R < 1 || Cin % 64 != 0 || Cout % 64 != 0) return 0;
    pl->geo = 1; pl->R = R; pl->pc = W + 1; pl->xslots = (R + 2) * (W + 1) + 1;
  }
  if (Cin % 64 == 0 && Cout % 64 == 0) { pl->ct = 2; pl->nt = 2; }
  else if (Cout % 64 == 0) { pl->ct = 1; pl->nt = 2; }
  else { pl->ct = 1; pl->nt = 1; }
  pl->nbs = pl->ct == 2 ? 1 : 2;
  pl->spw = 4 / (pl->ct * pl->nt);
  pl->pairs_n = Cout / (32 * pl->nt);
  pl->pairs = (Cin / (32 * pl->ct)) * pl->pairs_n;
  const int nblk = pl->geo ? B * ((H + pl->R - 1) / pl->R) : B * (H / 8) * (W / 8);
  const int total_stages = (nblk + pl->nbs - 1) / pl->nbs;
  int ns = 512 / pl->pairs;                       // one resident round of 2 workgroups per CU
  if (ns > total_stages / 8) ns = total_stages / 8;
  if (ns < 1) ns = 1;
  pl->stages = (total_stages + ns - 1) / ns;
  pl->nsplit = (total_stages + pl->stages - 1) / pl->stages;
  pl->nslabs = pl->nsplit * pl->spw;
  pl->ws_bytes = (size_t)pl->nslabs * KH * KW * Cin * Cout * sizeof(float);
  return 1;
}

template <int NKH, int NKW>
static void wgrad_halo_launch_class(const WgradHaloParams& p, const WgradHaloPlan& pl, bool half, bool one, hipStream_t st) {
  const dim3 grid(pl.nsplit, pl.pairs);
  if (pl.geo == 1) {   // row bands (64 x 64 channel chunks, no LN prologue: host checks)
    if (one) {
      if (half) hipLaunchKernelGGL((conv_wgrad_halo3_kernel<2, 2, true, true, NKH, NKW, false, 1, true>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((conv_wgrad_halo3_kernel<2, 2, false, true, NKH, NKW, false, 1, true>), grid, dim3(256), 0, st, p);
    } else if (half) hipLaunchKernelGGL((conv_wgrad_halo3_kernel<2, 2, true, true, NKH, NKW, false, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_halo3_kernel<2, 2, false, true, NKH, NKW, false, 1>), grid, dim3(256), 0, st, p);
    return;
  }
#define SGG_WH(CT, NT, PF)                                                                                          \
  do {                                                                                                              \
    if (one) {                                                                                                      \
      if (half) hipLaunchKernelGGL((conv_wgrad_halo3_kernel<CT, NT, true, PF, NKH, NKW, false, 0, true>), grid, dim3(256), 0, st, p);   \
      else hipLaunchKernelGGL((conv_wgrad_halo3_kernel<CT, NT, false, PF, NKH, NKW, false, 0, true>), grid, dim3(256), 0, st, p);       \
    } else if (p.ln_stats) {                                                                                        \
      if (half) hipLaunchKernelGGL((conv_wgrad_halo3_kernel<CT, NT, true, PF, NKH, NKW, true>), grid, dim3(256), 0, st, p);   \
      else hipLaunchKernelGGL((conv_wgrad_halo3_kernel<CT, NT, false, PF, NKH, NKW, true>), grid, dim3(256), 0, st, p);       \
    } else if (half) hipLaunchKernelGGL((conv_wgrad_halo3_kernel<CT, NT, true, PF, NKH, NKW, false>), grid, dim3(256), 0, st, p);   \
    else hipLaunchKernelGGL((conv_wgrad_halo3_kernel<CT, NT, false, PF, NKH, NKW, false>), grid, dim3(256), 0, st, p);       \
  } while (0)
  if (pl.ct == 2) SGG_WH(2, 2, true);
  else if (pl.nt == 2) SGG_WH(1, 2, false);
  else SGG_WH(1, 1, true);
#undef SGG_WH
}

// stride: 1 (3x3) or 2 (5x5, one launch per parity class of the taps); pad_t / pad_l: SAME padding before
void sgg_wgrad_halo_launch(const float* x, const float* dy, float* slabs, int B, int H, int W, int Cin, int Cout, int stride,
                           int pad_t, int pad_l, int precision, const float* amax_x, const float* amax_dy, const WgradHaloPlan& pl,
                           hipStream_t st, const float* ln_stats, const float* ln_gamma, const float* ln_beta, int operand_format) {
  WgradHaloParams p;
  p.x_s16 = operand_format & 1; p.dy_s16 = (operand_format >> 1) & 1;
  p.x = x; p.dy = dy; p.slabs = slabs; p.amax_x = amax_x; p.amax_dy = amax_dy;
  p.ln_stats = ln_stats; p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.B = B;
  p.H = H; p.W = W; p.C = Cin; p.N = Cout; p.bh = H / 8; p.bw = W / 8; p.nblk = B * p.bh * p.bw;
  p.R = pl.R; p.pc = pl.pc; p.xslots = pl.xslots; p.npx = pl.R * W; p.magic_w = (unsigned)((0x100000000ull + W - 1) / W);
  p.magic_pc = (unsigned)((0x100000000ull + pl.pc - 1) / pl.pc);
  if (pl.geo == 1) { p.bh = (H + pl.R - 1) / pl.R; p.bw = 1; p.nblk = B * p.bh; }
  p.Hx = H * stride; p.Wx = W * stride; p.sxy = stride;
  p.pairs_n = pl.pairs_n; p.stages = pl.stages;
  p.x_bytes = (unsigned)((size_t)B * p.Hx * p.Wx * Cin * sizeof(float));
  p.dy_bytes = (unsigned)((size_t)B * H * W * Cout * sizeof(float));
  const bool half = sgg_prec_half(precision), one = sgg_prec_one(precision);
  if (stride == 1) {
    p.cy = p.cx = 0; p.a0y = p.a0x = -1; p.kh0 = p.kw0 = 0; p.kstep = 1; p.KWt = 3; p.taps_total = 9;
    wgrad_halo_launch_class<3, 3>(p, pl, half, one, st);
    return;
  }
  p.kstep = 2; p.KWt = 5; p.taps_total = 25;
  for (int cy = 0; cy < 2; ++cy)
    for (int cx = 0; cx < 2; ++cx) {
      // taps kh with (kh - pad_t) mod 2 == cy: kh0 = first such tap, sub-grid offset a0 = floor((kh0 - pad_t - cy) / 2) + ... = (kh0 - pad_t - cy) / 2
      const int kh0 = (pad_t + cy) % 2, kw0 = (pad_l + cx) % 2;
      p.cy = cy; p.cx = cx; p.kh0 = kh0; p.kw0 = kw0;
      p.a0y = (kh0 - pad_t - cy) / 2;           // exact: kh0 - pad_t - cy is even (and <= 0)
      p.a0x = (kw0 - pad_l - cx) / 2;
      const int nkh = (5 - kh0 + 1) / 2, nkw = (5 - kw0 + 1) / 2;
      if (nkh == 3 && nkw == 3) wgrad_halo_launch_class<3, 3>(p, pl, half, one, st);
      else if (nkh == 3) wgrad_halo_launch_class<3, 2>(p, pl, half, one, st);
      else if (nkw == 3) wgrad_halo_launch_class<2, 3>(p, pl, half, one, st);
      else wgrad_halo_launch_class<2, 2>(p, pl, half, one, st);
    }
}
