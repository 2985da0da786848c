// Halo-resident 3x3 stride-1 convolution (split 16-bit modes): internal interface between conv_halo.hip (kernel, launch)
// and conv_gather.hip (the sgg_conv2d_nhwc_fwd / _dgrad entry points that dispatch to it).
#pragma once
#include "sgg_common.h"

struct HaloParams {
  const float* src;       // [B, H, W, C] f32 NHWC (forward: x; dgrad: dy)
  const void* wfrag;      // weights as MFMA B fragments (sgg_conv_split_weights_frag)
  const float* bias;      // may be null
  float* out;             // [B, H, W, N]
  const float* amax_src;  // precision 2: device words with max|src| and max|w|
  const float* amax_w;
  float* tile_stats;      // optional: per (8x8 block, wave column range) (count, mean, M2, max dev) for the following LayerNorm
  // LN prologue (forward only, optional): src holds the PRE-LayerNorm convolution output y of the producing layer; the patch staging
  // applies a = ELU((y - mean_b) * rstd_b * gamma_c + beta_c) on the fly (ln_stats [B][2] from sgg_layernorm_hwc_finalize), so the
  // LayerNorm apply pass and the materialised activation are not needed (generator_with_attention.py:30..56)
  const float* ln_stats;
  const float* ln_gamma;
  const float* ln_beta;
  int B, H, W, C, N;
  int bh, bw, nblk;       // 8x8 blocks per image (rows, cols) and in total
  int flip;               // 0: forward (correlation); 1: dgrad (taps mirrored)
  unsigned src_bytes, w_bytes;
  int gx;                 // workgroups per XCD (set by sgg_halo_launch)
  // Addressing in floats (sgg_halo_dense_strides fills the NHWC defaults).  Source: grid row / pixel strides and the offset of
  // 32-channel chunk cc = (cc >> 1) * in_cA + (cc & 1) * in_cB; output: the same for 32-column group g.  A 5x5 stride-2
  // convolution over 32 channels runs here as a 3x3 stride-1 convolution over the SPACE-TO-DEPTH view of x (chunk = pixel parity
  // (qy, qx): row stride 2*Wx*32, pixel stride 64, cA = Wx*32, cB = 32), its dgrad writes dx through the same view.
  int in_rs, in_ps, in_cA, in_cB;
  int out_rs, out_ps, out_nA, out_nB;
  int ln_nc;              // LN prologue: real channels of the source (a power of two: C, or 32 for the space-to-depth view)
  int frag16;             // 1: wfrag holds the fragments of the K = 32 MFMA shape (w_split_layout 4): the producer / consumer kernel
  int src_s16;            // 1: src is a pre-split ("S16") tensor of the LayerNorm kernels (split16.h): staged without arithmetic
  int cu_cap;             // > 0: the persistent workgroups occupy at most this many of an XCD's 32 CUs (launch hint of the forward entry point)
};
// CUs of an XCD a persistent launch may occupy
inline int sgg_persist_cus(int cu_cap) { return (cu_cap > 0 && cu_cap < SGG_PERSIST_CUS_PER_XCD) ? cu_cap : SGG_PERSIST_CUS_PER_XCD; }
inline void sgg_halo_dense_strides(HaloParams& h) {
  h.in_rs = h.W * h.C; h.in_ps = h.C; h.in_cA = 64; h.in_cB = 32;
  h.out_rs = h.W * h.N; h.out_ps = h.N; h.out_nA = 64; h.out_nB = 32;
  h.ln_nc = h.C;
  h.frag16 = 0;
  h.src_s16 = 0;
  h.cu_cap = 0;
}

// 1 if the halo kernel serves a 3x3 / stride-1 convolution over an H x W grid in this precision
int sgg_halo_applicable(int KH, int KW, int stride, int H, int W, int C, int N, int precision);
int sgg_s2d_applicable(int KH, int KW, int stride, int Hi, int Wi, int Cin, int Cout, int precision);
// columns covered by one (count, mean, M2) partial of the halo kernel for N output channels
int sgg_halo_stats_cols(int N);
void sgg_halo_launch(const HaloParams& p, int precision, hipStream_t st);
// producer / consumer form for 128-column tiles in the two-piece modes (conv_halo_pc.hip; weights in w_split_layout 4);
// sgg_halo_launch dispatches to it when HaloParams::frag16 is set
int sgg_halo_pc_applicable(int C, int N, int precision);
// its four-block form for 64-column tiles (pre-split sources only)
int sgg_halo_pc64_applicable(int C, int N, int precision);
void sgg_halo_pc_launch(const HaloParams& p, int precision, hipStream_t st);

// ---- halo-resident 3x3 stride-1 wgrad (conv_wgrad_halo.hip) ---------------------------------------------------
struct WgradHaloPlan {
  int ct, nt;           // channel chunk of a workgroup: 32*ct input x 32*nt output channels
  int nbs;              // 8x8 blocks staged together
  int spw;              // partial slabs a workgroup writes (waves that split the pixels of a chunk)
  int pairs, pairs_n;   // (Cin chunk, Cout chunk) pairs; Cout chunks
  int nsplit;           // workgroups along the pixel dimension
  int stages;           // stages per workgroup
  int nslabs;           // nsplit * spw
  int geo;              // 0: 8x8 pixel blocks;  1: row bands (R full-width rows, up to 112 pixels) for grids that 8x8 blocks do not tile
  int R, pc, xslots;    // geo 1: rows per band, patch pitch (W + 1: one shared zero column), patch slots ((R + 2) * pc + 1)
  size_t ws_bytes;
};
// returns 1 and fills the plan if the shape is served (H, W = the dy grid; 3x3 stride 1 or 5x5 stride 2; channels % 32 == 0;
// H % 8 == W % 8 == 0, or - channels % 64 == 0 - any H with W <= 28: row bands)
int sgg_wgrad_halo_plan(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, WgradHaloPlan* pl);
// writes pl.nslabs partial dW slabs [slab][9][Cin][Cout] (unscaled f32) into `slabs`
// operand_format: bit 0 = x, bit 1 = dy is a pre-split ("S16") tensor (split16.h)
void sgg_wgrad_halo_launch(const float* x, const float* dy, float* slabs, int B, int H, int W, int Cin, int Cout, int stride,
                           int pad_t, int pad_l, int precision, const float* amax_x, const float* amax_dy, const WgradHaloPlan& pl,
                           hipStream_t st, const float* ln_stats = nullptr, const float* ln_gamma = nullptr,
                           const float* ln_beta = nullptr, int operand_format = 0);

// ---- filter gradient on pre-split operands staged by LDS-DMA (conv_wgrad_dma.hip) ---------------------------------------------
struct WgradDmaPlan {
  int nt;               // 32-column output tiles of a workgroup: 4 (64 x 128 channel tile) or 2 (64 x 64, two pixel halves)
  int spw;              // partial slabs a workgroup writes
  int pairs, pairs_n;   // channel tiles; Cout tiles
  int nsplit, stages, nslabs;
  int geo, R, pc, xslots; // geo 1: row bands as WgradHaloPlan (grids that 8x8 blocks do not tile; 64 x 64 tiles)
  size_t ws_bytes;
};
// returns 1 and fills the plan if the shape is served (H, W = the dy grid: divisible by 8, or row bands of up to 112 pixels; 3x3 stride 1 or 5x5 stride 2; channels % 64 == 0)
int sgg_wgrad_dma_plan(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, WgradDmaPlan* pl);
// x, dy: pre-split ("S16") tensors; writes pl.nslabs partial dW slabs [slab][taps][Cin][Cout] (unscaled f32) into `slabs`
void sgg_wgrad_dma_launch(const void* x, const void* dy, float* slabs, int B, int H, int W, int Cin, int Cout, int stride, int pad_t,
                          int pad_l, const float* amax_x, const float* amax_dy, const WgradDmaPlan& pl, hipStream_t st);

// ---- band-resident 5x5 stride-2 convolution, forward and dgrad (conv_s2.hip) -----------------------------------------
struct S2Params {
  const float* src;       // forward: x [B, 2*Ho, 2*Wo, C];  dgrad: dy [B, Ho, Wo, C]
  const void* wfrag;      // 25 taps as MFMA B fragments (sgg_conv_split_weights_frag with taps = 25)
  const float* bias;      // forward only, may be null
  float* out;             // forward: y [B, Ho, Wo, N];  dgrad: dx [B, 2*Ho, 2*Wo, N]
  const float* amax_src;
  const float* amax_w;
  float* tile_stats;      // forward, optional: (count, mean, M2) per (band, 32 output channels)
  // LN prologue (forward, two-piece modes, optional): src is the producing layer's PRE-LayerNorm output; the patch staging applies
  // ELU((y - mean_b) * rstd_b * gamma_c + beta_c) (see HaloParams)
  const float* ln_stats;
  const float* ln_gamma;
  const float* ln_beta;
  int B, Ho, Wo;          // the half-resolution grid (forward: output positions; dgrad: dy positions)
  int C, N;               // contraction channels, output channels
  int M;                  // B * Ho * Wo
  int nbands;             // ceil(M / 224)
  int pitch;              // Wo: slots per patch row (no halo columns: edge lanes read a zero slot)
  unsigned src_bytes, w_bytes;
  int gx;                 // workgroups per XCD (set by sgg_s2_launch)
  int src_s16;            // 1: src is a pre-split ("S16") tensor (split16.h)
  int ksplit;             // 1, or 2: two workgroups per (band, n-tile), each contracting half of the channel chunks and ADDING its
                          // partial into the zeroed output (a + b = b + a: still deterministic); set by sgg_s2_launch
  int cu_cap;             // as HaloParams::cu_cap
};
// 1 if the band-resident kernel serves this 5x5 / stride-2 / SAME convolution (Hi, Wi = the full-resolution grid, both even;
// C = contraction channels, N = output channels of the direction asked for)
int sgg_s2_applicable(int KH, int KW, int stride, int B, int Hi, int Wi, int C, int N, int precision);
// (count, mean, M2) partials per sample the forward emits, 0 if bands do not align with samples
int sgg_s2_stats_per_sample(int Ho, int Wo, int N);
void sgg_s2_launch(const S2Params& p, int dgrad, int precision, hipStream_t st);
