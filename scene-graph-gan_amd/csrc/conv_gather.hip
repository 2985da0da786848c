// Implicit-GEMM "gather" convolution on f32 MFMA for NHWC tensors (gfx950).
//
// One kernel serves both
//   * forward   tf.layers.conv2d(padding="same")          reference: architectures/generator_with_attention.py:29-68
//   * dgrad     Conv2DBackpropInput (autodiff of the same) reference: train.py:265-266 (optimizer.minimize)
// through the parameterisation
//   out[b, y*osy+ooy, x*osx+oox, n] = bias[n] + sum_{th,tw} sum_c src[b, y*sy+oy+th*dy, x*sx+ox+tw*dx, c] * wm[tap(th,tw)][n][c]
// over a "virtual" output grid [B, Hm, Wm] per class (stride-2 dgrad = 4 parity classes, each a stride-1
// correlation with a sub-sampled kernel; forward = 1 class).  Out-of-range source pixels contribute zero
// (TF SAME padding, incl. the asymmetric (1,2) case).  Weights are [tap][n][c] with c contiguous:
// forward uses the HWOI transpose made by sgg_hwio_to_hwoi, dgrad uses the HWIO tensor as it is.
//
// Tiling: workgroup = 256 threads = 4 waves; tile BM x BN x 32(k); A (gathered pixels) and B (weights) are
// staged global -> registers -> LDS (KC layout, see mma_f32.h) with the next slab's global loads in flight
// while the current slab is contracted (register prefetch, one LDS buffer).
#include "split16.h"
#include "mma_f32.h"
#include "conv_halo.h"
#include <stdlib.h>

struct GatherClass {
  int Hm, Wm, M;        // virtual grid and B*Hm*Wm
  int nth, ntw;         // taps in this class
  int oy, ox;           // source offset
  int kh0, kw0, kstep;  // weight tap index = (kh0 + kstep*th)*KW + (kw0 + kstep*tw)
  int ooy, oox;         // output offset
  int mtiles;
};

struct GatherParams {
  const float* src;
  const float* wm;
  const float* bias;  // may be null
  float* out;
  const void* w_split;  // optional: the weights pre-split into P 16-bit planes [P][taps*N*C] (sgg_conv_split_weights), else null
  const float* amax_src;  // f16x3 mode: device words holding max|src| and max|w| (power-of-two scaling into fp16 range)
  const float* amax_w;
  int B, Hs, Ws, C;  // source grid, channels (= contraction length per tap), C % 32 == 0
  int N;             // output channels, N % 32 == 0
  int Ho, Wo;        // full output grid
  int sy, sx, dy, dx, osy, osx, KW;
  float* tile_stats;   // optional (forward, HW % BM == 0): per output tile (count, mean, M2) of y for the following LayerNorm
  int hw;              // output pixels per sample (tile_stats indexing)
  int ncls;
  unsigned w_bytes;    // size of the weight tensor (bounds the buffer descriptor of the v2 kernel)
  GatherClass cls[4];
};

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_gather_kernel(GatherParams p) {
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NPA = BM * 8 / 256, NPB = (BN * 8 + 255) / 256;
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tile");

  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * SGG_LDK + BM];
  float* A_s = lds;
  float* B_s = lds + BM * SGG_LDK;
  int* out_off_s = reinterpret_cast<int*>(lds + (BM + BN) * SGG_LDK);

  const GatherClass& c = p.cls[blockIdx.y];
  const int ntiles_n = p.N / BN;
  const int nwg = c.mtiles * ntiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int mt = lid / ntiles_n, nt = lid % ntiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;

  // ---- per-thread gather rows -----------------------------------------------------------------
  const int col4 = tid & 7;
  int a_ys[NPA], a_xs[NPA], a_base[NPA];
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    const int m = m0 + (tid >> 3) + 32 * j;
    if (m < c.M) {
      const int x = m % c.Wm;
      const int t = m / c.Wm;
      const int y = t % c.Hm;
      const int b = t / c.Hm;
      a_ys[j] = y * p.sy + c.oy;
      a_xs[j] = x * p.sx + c.ox;
      a_base[j] = b * p.Hs * p.Ws * p.C + col4 * 4;
    } else {
      a_ys[j] = -(1 << 28);  // never in range
      a_xs[j] = 0;
      a_base[j] = 0;
    }
  }
  // output row offsets (element offset of channel 0, or -1) for the epilogue
  for (int r = tid; r < BM; r += 256) {
    const int m = m0 + r;
    int off = -1;
    if (m < c.M) {
      const int x = m % c.Wm;
      const int t = m / c.Wm;
      const int y = t % c.Hm;
      const int b = t / c.Hm;
      off = ((b * p.Ho + y * p.osy + c.ooy) * p.Wo + x * p.osx + c.oox) * p.N;
    }
    out_off_s[r] = off;
  }

  f32x16 acc[TM][TN];
  acc_zero<TM, TN>(acc);

  const int cchunks = p.C >> 5;
  const int n_iters = c.nth * c.ntw * cchunks;

  f32x4 ra[NPA], rb[NPB];
  auto issue_loads = [&](int it) {
    const int tap = it / cchunks;
    const int c0 = (it - tap * cchunks) << 5;
    const int th = tap / c.ntw, tw = tap - th * c.ntw;
    const int ty = th * p.dy, tx = tw * p.dx;
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int yy = a_ys[j] + ty, xx = a_xs[j] + tx;
      const bool ok = (unsigned)yy < (unsigned)p.Hs && (unsigned)xx < (unsigned)p.Ws;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(p.src + (size_t)(a_base[j] + (yy * p.Ws + xx) * p.C + c0));
      ra[j] = v;
    }
    const float* wtap = p.wm + (size_t)((c.kh0 + c.kstep * th) * p.KW + (c.kw0 + c.kstep * tw)) * p.N * p.C;
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const int row = (tid >> 3) + 32 * j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < BN) v = *reinterpret_cast<const f32x4*>(wtap + (size_t)(n0 + row) * p.C + c0 + col4 * 4);
      rb[j] = v;
    }
  };

  issue_loads(0);
  for (int it = 0; it < n_iters; ++it) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPA; ++j)
      *reinterpret_cast<f32x4*>(A_s + ((tid >> 3) + 32 * j) * SGG_LDK + col4 * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const int row = (tid >> 3) + 32 * j;
      if (row < BN) *reinterpret_cast<f32x4*>(B_s + row * SGG_LDK + col4 * 4) = rb[j];
    }
    __syncthreads();
    if (it + 1 < n_iters) issue_loads(it + 1);
    mma_slab_kc_kc<TM, TN>(A_s, B_s, wm0, wn0, lane, acc);
  }

  // ---- epilogue: + bias, scatter rows ----------------------------------------------------------
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 32 + acc_col(lane);
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + tm * 32 + acc_row(r, lane);
        const int off = out_off_s[row];
        if (off >= 0) p.out[(size_t)off + n] = acc[tm][tn][r] + bv;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// v3 of the same contraction for the MFMA-bound layers (N % 128 == 0 and C % 64 == 0): 64-deep slabs, so each
// wave issues 128 MFMAs (8192 pipe cycles) per pair of barriers instead of 64, and a branch-free fetch.
//   * measured on v1 (rocprofv3 PMC, conv2_4 forward): MFMA pipe busy 75 % at 2.14 GHz; per wave and slab 4096
//     cycles of own MFMA, ~1000 of instruction issue (half of it address arithmetic with integer division) and
//     ~1300 parked at s_waitcnt / s_barrier -> halve the barriers per MFMA and shrink the fetch;
//   * gather through raw buffer loads (SRSRC descriptor): out-of-image taps, rows past M and slabs past the last
//     get an out-of-range offset and the hardware returns zeros (TF SAME padding, no branches, no traffic);
//   * the (tap, channel-block) walk is scalar counters; the slab fetch is one straight-line block issued right
//     after the second barrier and in flight during the whole MFMA burst (register prefetch, one LDS buffer of
//     (128+128) x 68 floats = 69.6 KB -> two workgroups per CU).
// ---------------------------------------------------------------------------------------------------
#define SGG_LDK64 68

template <int BM, int BN, int WGM, int WGN, int BK>
__global__ __launch_bounds__(256, (BK == 32 ? 3 : 2)) void conv_gather3_kernel(GatherParams p, unsigned src_bytes) {
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int CPR = BK / 4;                 // float4 pieces per tile row
  constexpr int RPP = 256 / CPR;              // tile rows covered by one pass of the 256 threads
  constexpr int LDKK = BK + 4;
  constexpr int NPA = BM / RPP, NPB = BN / RPP;
  static_assert(WGM * WGN == 4 && BM % RPP == 0 && BN % RPP == 0 && (BK == 32 || BK == 64), "tile");

  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDKK + BM];
  float* A_s = lds;
  float* B_s = lds + BM * LDKK;
  int* out_off_s = reinterpret_cast<int*>(lds + (BM + BN) * LDKK);

  const GatherClass& c = p.cls[blockIdx.y];
  const int ntiles_n = p.N / BN;
  const int nwg = c.mtiles * ntiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int mt = lid / ntiles_n, nt = lid % ntiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int col4 = tid % CPR, prow = tid / CPR;

  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wm), 0, p.w_bytes, 0x00020000);

  int a_ys[NPA], a_xs[NPA], a_off[NPA];
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    const int m = m0 + prow + RPP * j;
    if (m < c.M) {
      const int x = m % c.Wm;
      const int t = m / c.Wm;
      const int y = t % c.Hm;
      const int b = t / c.Hm;
      a_ys[j] = y * p.sy + c.oy;
      a_xs[j] = x * p.sx + c.ox;
      a_off[j] = ((b * p.Hs + a_ys[j]) * p.Ws + a_xs[j]) * p.C + col4 * 4;
    } else {
      a_ys[j] = -(1 << 28);
      a_xs[j] = 0;
      a_off[j] = 0;
    }
  }
  const int b_off0 = (n0 + prow) * p.C + col4 * 4;      // row j adds RPP*j*C
  for (int r = tid; r < BM; r += 256) {
    const int m = m0 + r;
    int off = -1;
    if (m < c.M) {
      const int x = m % c.Wm;
      const int t = m / c.Wm;
      const int y = t % c.Hm;
      const int b = t / c.Hm;
      off = ((b * p.Ho + y * p.osy + c.ooy) * p.Wo + x * p.osx + c.oox) * p.N;
    }
    out_off_s[r] = off;
  }

  f32x16 acc[TM][TN];
  acc_zero<TM, TN>(acc);

  const int n_iters = c.nth * c.ntw * (p.C / BK);
  int s_th = 0, s_tw = 0, s_c0 = 0, s_it = 0;      // scalar walk over (th, tw, c0)
  f32x4 ra[NPA], rb[NPB];
  auto issue_loads = [&]() {
    const unsigned dead = (unsigned)(s_it >= n_iters);   // past the last slab: all offsets out of range -> zeros
    const int ty = s_th * p.dy, tx = s_tw * p.dx;
    const int tapoff = (ty * p.Ws + tx) * p.C + s_c0;
    const int woff = ((c.kh0 + c.kstep * s_th) * p.KW + (c.kw0 + c.kstep * s_tw)) * p.N * p.C + s_c0 + b_off0;
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int yy = a_ys[j] + ty, xx = a_xs[j] + tx;
      // bitwise (not short-circuit) so the whole slab fetch stays one straight-line block
      const unsigned bad = (unsigned)((unsigned)yy >= (unsigned)p.Hs) | (unsigned)((unsigned)xx >= (unsigned)p.Ws) | dead;
      ra[j] = buf_load4(rs_src, ((unsigned)(a_off[j] + tapoff) * 4u) | ((0u - bad) & SGG_OOB));
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j)
      rb[j] = buf_load4(rs_w, ((unsigned)(woff + RPP * j * p.C) * 4u) | ((0u - dead) & SGG_OOB));
    ++s_it;
    s_c0 += BK;
    const bool wrap_c = s_c0 >= p.C;
    s_c0 = wrap_c ? 0 : s_c0;
    s_tw += wrap_c ? 1 : 0;
    const bool wrap_w = s_tw >= c.ntw;
    s_tw = wrap_w ? 0 : s_tw;
    s_th += wrap_w ? 1 : 0;
  };

  issue_loads();
  for (int it = 0; it < n_iters; ++it) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // every wave is done reading the previous slab
#pragma unroll
    for (int j = 0; j < NPA; ++j) *reinterpret_cast<f32x4*>(A_s + (prow + RPP * j) * LDKK + col4 * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < NPB; ++j) *reinterpret_cast<f32x4*>(B_s + (prow + RPP * j) * LDKK + col4 * 4) = rb[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // slab visible to every wave
    issue_loads();                                        // next slab: in flight during the whole burst below
    __builtin_amdgcn_sched_barrier(0);                    // (hipcc otherwise sinks the fetch below the MFMAs)
    mma_slab_kc_kc<TM, TN, LDKK, BK / 8>(A_s, B_s, wm0, wn0, lane, acc);
  }

#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 32 + acc_col(lane);
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + tm * 32 + acc_row(r, lane);
        const int off = out_off_s[row];
        if (off >= 0) p.out[(size_t)off + n] = acc[tm][tn][r] + bv;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Split-bf16 variant: the same contraction on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate) with every f32
// operand split into P bf16 pieces x = x0 + x1 (+ x2) at the LDS-write and the cross terms accumulated in f32:
//   P = 2 (3 products: x0w0 + x0w1 + x1w0):  drops terms of relative size 2^-17: logits err 2.0e-5 vs fp64
//   P = 3 (6 products: + x1w1 + x0w2 + x2w0): drops terms of relative size 2^-25: logits err 1.9e-6 vs fp64,
//          the same as the native f32 MFMA path (2.7e-6)   [tests/test_split_numerics.py: CPU emulation of the piece arithmetic]
// bf16 x bf16 products are exact in f32, the accumulator is f32, so only the dropped cross terms differ from f32.
// Same skeleton as conv_gather3_kernel (buffer-load gather, scalar tap walk, register prefetch, one LDS buffer);
// LDS planes are [row][32 k] bf16 (64 B rows) with the 16-B chunk index XOR-swizzled by (row >> 2) & 3, which makes
// both the ds_write_b128 of the split pieces and the ds_read_b128 of the MFMA fragments conflict free.
// ---------------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, int P, bool WS, bool HALF, int BK>
__global__ __launch_bounds__(256, (P == 2 && BK == 32 && BN <= 128 ? 3 : 2)) void conv_gather_bf16s_kernel(GatherParams p, unsigned src_bytes) {
  static_assert(!HALF || P == 2, "f16 mode uses two pieces");
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int CPR = BK / 8;                           // 16-B chunks (8 k) per plane row
  constexpr int RPP = 256 / CPR;                        // rows staged per pass of the 256 threads
  constexpr int NPA = BM / RPP, NPB = (BN + RPP - 1) / RPP;
  constexpr int ROWB = BK * 2;                          // bytes per plane row (BK sixteen-bit pieces)
  static_assert(BK == 32 || BK == 64, "slab depth");
  static_assert(WGM * WGN == 4 && BM % 64 == 0 && BN % 32 == 0 && (P == 2 || P == 3), "tile");

  __shared__ __attribute__((aligned(16))) unsigned char lds[(BM + BN) * ROWB * P + BM * 4];
  unsigned char* A_s = lds;                             // plane pp at + pp * BM * ROWB
  unsigned char* B_s = lds + P * BM * ROWB;
  int* out_off_s = reinterpret_cast<int*>(lds + (BM + BN) * ROWB * P);

  const GatherClass& c = p.cls[blockIdx.y];
  const int ntiles_n = p.N / BN;
  const int nwg = c.mtiles * ntiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int mt = lid / ntiles_n, nt = lid % ntiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int chunk = tid % CPR, prow = tid / CPR;        // this thread stages k = 8*chunk .. 8*chunk+7 of rows prow + RPP*j

  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, src_bytes, 0x00020000);
  // WS: weights arrive pre-split (P bf16 planes, written once per optimiser step) - no split VALU for the B operand
  const __amdgpu_buffer_rsrc_t rs_w =
      WS ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_split), 0, (p.w_bytes / 2) * P, 0x00020000)
         : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wm), 0, p.w_bytes, 0x00020000);
  const unsigned w_plane_bytes = p.w_bytes / 2;
  int ea = 0, eb = 0;
  if constexpr (HALF) {
    ea = scale_exp_from_amax(*p.amax_src);
    eb = scale_exp_from_amax(*p.amax_w);
  }
  const float sa = ldexpf(1.f, ea), sb = ldexpf(1.f, eb);

  int a_ys[NPA], a_xs[NPA], a_off[NPA];
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    const int m = m0 + prow + RPP * j;
    if (m < c.M) {
      const int x = m % c.Wm;
      const int t = m / c.Wm;
      const int y = t % c.Hm;
      const int b = t / c.Hm;
      a_ys[j] = y * p.sy + c.oy;
      a_xs[j] = x * p.sx + c.ox;
      a_off[j] = ((b * p.Hs + a_ys[j]) * p.Ws + a_xs[j]) * p.C + chunk * 8;
    } else {
      a_ys[j] = -(1 << 28);
      a_xs[j] = 0;
      a_off[j] = 0;
    }
  }
  const int b_off0 = (n0 + prow) * p.C + chunk * 8;
  for (int r = tid; r < BM; r += 256) {
    const int m = m0 + r;
    int off = -1;
    if (m < c.M) {
      const int x = m % c.Wm;
      const int t = m / c.Wm;
      const int y = t % c.Hm;
      const int b = t / c.Hm;
      off = ((b * p.Ho + y * p.osy + c.ooy) * p.Wo + x * p.osx + c.oox) * p.N;
    }
    out_off_s[r] = off;
  }

  f32x16 acc[TM][TN];
  acc_zero<TM, TN>(acc);

  const int n_iters = c.nth * c.ntw * (p.C / BK);
  int s_th = 0, s_tw = 0, s_c0 = 0, s_it = 0;
  f32x4 ra[NPA][2], rb[NPB][WS ? P : 2];
  auto issue_loads = [&]() {
    const unsigned dead = (unsigned)(s_it >= n_iters);
    const int ty = s_th * p.dy, tx = s_tw * p.dx;
    const int tapoff = (ty * p.Ws + tx) * p.C + s_c0;
    const int woff = ((c.kh0 + c.kstep * s_th) * p.KW + (c.kw0 + c.kstep * s_tw)) * p.N * p.C + s_c0 + b_off0;
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int yy = a_ys[j] + ty, xx = a_xs[j] + tx;
      const unsigned bad = (unsigned)((unsigned)yy >= (unsigned)p.Hs) | (unsigned)((unsigned)xx >= (unsigned)p.Ws) | dead;
      const unsigned off = ((unsigned)(a_off[j] + tapoff) * 4u) | ((0u - bad) & SGG_OOB);
      ra[j][0] = buf_load4(rs_src, off);
      ra[j][1] = buf_load4(rs_src, off + 16u);
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const unsigned deadb = dead | (unsigned)(prow + RPP * j >= BN);    // (BN < RPP: some threads stage no weights)
      if constexpr (WS) {
        const unsigned off = ((unsigned)(woff + RPP * j * p.C) * 2u) | ((0u - deadb) & SGG_OOB);
#pragma unroll
        for (int pp = 0; pp < P; ++pp) rb[j][pp] = buf_load4(rs_w, off + pp * w_plane_bytes);
      } else {
        const unsigned off = ((unsigned)(woff + RPP * j * p.C) * 4u) | ((0u - deadb) & SGG_OOB);
        rb[j][0] = buf_load4(rs_w, off);
        rb[j][1] = buf_load4(rs_w, off + 16u);
      }
    }
    ++s_it;
    s_c0 += BK;
    const bool wrap_c = s_c0 >= p.C;
    s_c0 = wrap_c ? 0 : s_c0;
    s_tw += wrap_c ? 1 : 0;
    const bool wrap_w = s_tw >= c.ntw;
    s_tw = wrap_w ? 0 : s_tw;
    s_th += wrap_w ? 1 : 0;
  };
  // swizzled byte offset of (row, logical 16-B chunk q) inside one plane
  // (64-B rows: 4 rows share a 256-B bank row -> XOR with (row>>2)&3; 128-B rows: 2 rows -> XOR with (row>>1)&7)
  auto sw = [](int row, int q) { return row * ROWB + ((q ^ (BK == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7))) << 4); };

  issue_loads();
  for (int it = 0; it < n_iters; ++it) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      u32x4 pl[P];
      split8<P, HALF>(ra[j][0], ra[j][1], sa, pl);
#pragma unroll
      for (int pp = 0; pp < P; ++pp) *reinterpret_cast<u32x4*>(A_s + pp * BM * ROWB + sw(prow + RPP * j, chunk)) = pl[pp];
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      u32x4 pl[P];
      if constexpr (WS) {
#pragma unroll
        for (int pp = 0; pp < P; ++pp) pl[pp] = __builtin_bit_cast(u32x4, rb[j][pp]);
      } else {
        split8<P, HALF>(rb[j][0], rb[j][1], sb, pl);
      }
      if (prow + RPP * j < BN) {
#pragma unroll
        for (int pp = 0; pp < P; ++pp) *reinterpret_cast<u32x4*>(B_s + pp * BN * ROWB + sw(prow + RPP * j, chunk)) = pl[pp];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_loads();
    __builtin_amdgcn_sched_barrier(0);
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      u32x4 a[TM][P], b[TN][P];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
          a[tm][pp] = *reinterpret_cast<const u32x4*>(A_s + pp * BM * ROWB + sw(wm0 + tm * 32 + i, 2 * ks + h));
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
          b[tn][pp] = *reinterpret_cast<const u32x4*>(B_s + pp * BN * ROWB + sw(wn0 + tn * 32 + i, 2 * ks + h));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          f32x16 d = acc[tm][tn];
          if constexpr (P == 3) {      // smallest terms first
            d = mfma16<HALF>(a[tm][2], b[tn][0], d);
            d = mfma16<HALF>(a[tm][0], b[tn][2], d);
            d = mfma16<HALF>(a[tm][1], b[tn][1], d);
          }
          d = mfma16<HALF>(a[tm][1], b[tn][0], d);
          d = mfma16<HALF>(a[tm][0], b[tn][1], d);
          d = mfma16<HALF>(a[tm][0], b[tn][0], d);
          acc[tm][tn] = d;
        }
    }
  }

  // ---- epilogue: unscale, + bias, store; optionally the tile's LayerNorm partial statistics ---------------
  float lsum = 0.f;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 32 + acc_col(lane);
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + tm * 32 + acc_row(r, lane);
        const int off = out_off_s[row];
        const float v = HALF ? ldexpf(ldexpf(acc[tm][tn][r], -ea), -eb) + bv : acc[tm][tn][r] + bv;
        acc[tm][tn][r] = v;
        lsum += v;
        if (off >= 0) p.out[(size_t)off + n] = v;
      }
    }
  }
  if (p.tile_stats) {
    // (count, mean, M2) of the BM x BN outputs of this tile; the tile lies inside one sample (host guarantees hw % BM == 0).
    // Merged per sample with Chan's formula by ln_apply_elu_kernel -> the separate statistics pass over y is skipped.
    float* red = reinterpret_cast<float*>(lds);          // the operand planes are dead now
    __syncthreads();
    lsum = wave_sum(lsum);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    const float mean_t = (red[0] + red[1] + red[2] + red[3]) * (1.f / (float)(BM * BN));
    float q = 0.f, dm = 0.f;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[tm][tn][r] - mean_t;
          q += d * d;
          dm = fmaxf(dm, fabsf(d));
        }
    q = wave_sum(q);
    dm = wave_max(dm);
    if (lane == 0) { red[4 + wave] = q; red[8 + wave] = dm; }
    __syncthreads();
    if (tid == 0) {
      const int b = m0 / p.hw, t_in = (m0 - b * p.hw) / BM;
      const int tps = p.hw / BM;
      float* o = p.tile_stats + ((size_t)(b * tps + t_in) * ntiles_n + nt) * SGG_TS;
      o[0] = (float)(BM * BN);
      o[1] = mean_t;
      o[2] = red[4] + red[5] + red[6] + red[7];
      o[3] = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Cin = 3 forward (conv1_1, generator_with_attention.py:29): K = 27, HBM-bound on the 32-channel output.
// Weights are HWIO [3][3][3][Cout].
// ---------------------------------------------------------------------------------------------------
// Tile = 8 rows x 32 columns of output pixels.  The zero-padded 10 x 34 input patch sits in LDS as three channel PLANES
// [ci][row][pitch 36]; a wave owns two rows of the tile = two 32-pixel MFMA row tiles and contracts K = 27 (+ 1 zero) with
// fourteen v_mfma_f32_32x32x2_f32 per row tile: A[pixel][k = 2 s + h] is ONE conflict-free ds_read_b32 (32 consecutive floats
// per half-wave) at a per-lane tap offset, B[k][output channel] fourteen registers loaded once per workgroup.  28 MFMAs of 64
// cycles per wave and tile = 1792 cycles where the VALU form of rounds 1 - 3 (108 FMAs per pixel and 4 output channels, a rolling
// 3-row window) took 3456: the kernel was VALU-bound at 0.42 of the HBM peak (135 us), and is HBM-bound now.
// Sum order per output: bias, then (kh, kw, ci) ascending - the f32 MFMA is an exact, k-ordered fmaf chain (mma_f32.h), so the
// results are those of the VALU form bit for bit.
// tile_stats (optional): (count, mean, M2, max dev) of every wave's two rows of a tile for the following LayerNorm,
// [B][tiles_y * tiles_x * 4][4].
// workgroup barrier that orders LDS accesses only (no wait for outstanding global loads / stores)
__device__ __forceinline__ void c3_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// patch offset of tap k = (kh * 3 + kw) * 3 + ci in the planar 10 x 36 image (k = 27: the zero column of the contraction)
__host__ __device__ constexpr int c3_tap_off(int k) { return k < 27 ? (k % 3) * 360 + (k / 9) * 36 + (k / 3) % 3 : 0; }

// conv1_1's output stores: the kernel is a 91 % WRITE stream (411 MB per launch at batch 64) of whole 128-byte lines.  Measured with
// scripts/ubench/counter_calib.hip (profiles/r05_counter_calibration.log): a pure stream of 16-byte NONTEMPORAL stores sustains
// 4.9 - 5.0 TB/s on this chip, default-policy stores 6.7 TB/s and more - the nontemporal hint that pays in the MFMA-bound epilogues
// (SGG_CONV_NT_STORE) caps this kernel at its own store rate.  -DSGG_C3_NT_STORE=1 restores the hint.
#ifndef SGG_C3_NT_STORE
#define SGG_C3_NT_STORE 0
#endif
__device__ __forceinline__ void c3_out_store4(float* p, const f32x4& v) {     // p 16-byte aligned
#if SGG_C3_NT_STORE
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}

__global__ __launch_bounds__(256, 4) void conv_c3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ tile_stats, int H, int W, int pt, int pl,
                                                          int tiles_x, int tiles_y, int ntiles) {
  constexpr int COUT = 32, PITCH = 36, PLANE = 10 * PITCH;
  __shared__ float patch[3 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  // B operand (this lane: output channel i, k = 2 s + h), for every tile
  float wb[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + h;
    wb[s] = k < 27 ? w[k * COUT + i] : 0.f;
  }
  const float bv = bias[i];
  // the patch of the NEXT tile is fetched into registers while this tile is computed (two pixels per thread)
  float pv[2][3];
  auto load_patch = [&](int tile_) __attribute__((always_inline)) {
    const int tx_ = tile_ % tiles_x, t2_ = tile_ / tiles_x;
    const int ty_ = t2_ % tiles_y, b_ = t2_ / tiles_y;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx / 34, c = idx % 34;
      const int yy = ty_ * 8 - pt + r, xx = tx_ * 32 - pl + c;
      pv[k][0] = pv[k][1] = pv[k][2] = 0.f;
      if (tile_ < ntiles && idx < 10 * 34 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
        const float* px = x + ((size_t)(b_ * H + yy) * W + xx) * 3;
        pv[k][0] = px[0]; pv[k][1] = px[1]; pv[k][2] = px[2];
      }
    }
  };
  load_patch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int tx = tile % tiles_x, t2 = tile / tiles_x;
  const int ty = t2 % tiles_y, b = t2 / tiles_y;
  const int y0 = ty * 8, x0 = tx * 32;
  // (raw barriers with an LDS-only wait: __syncthreads() also drains vmcnt, i.e. waits for the previous tile's output stores)
  c3_lds_barrier();            // the previous tile's patch is free
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + 256 * k;
    if (idx < 10 * 34) {
      const int o = (idx / 34) * PITCH + idx % 34;
      patch[o] = pv[k][0];
      patch[PLANE + o] = pv[k][1];
      patch[2 * PLANE + o] = pv[k][2];
    }
  }
  c3_lds_barrier();
  load_patch(tile + gridDim.x);

  // rows 2 wave, 2 wave + 1 of the tile; accumulator register r of lane (i, h): pixel column acc_row(r, lane), output channel i
  f32x16 out[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[m][r] = bv;
  const float* prow = patch + 2 * wave * PITCH + i;
#ifndef C3_ABL_NOMFMA      // (timing-only ablations, wrong results: -DC3_ABL_NOMFMA no contraction, -DC3_ABL_NOSTORE no output stores)
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int ao = h ? c3_tap_off(2 * s + 1) : c3_tap_off(2 * s);
    out[0] = mfma32(prow[ao], wb[s], out[0]);
    out[1] = mfma32(prow[ao + PITCH], wb[s], out[1]);
  }
#else
  out[0][0] += prow[0];
  out[1][0] += prow[PITCH];
#endif
  // 16-byte stores through the in-register quad transpose (sgg_common.h): afterwards lane (h, g = i >> 2, k = i & 3) holds pixel
  // column 8 q + 4 h + k and output channels 4 g .. 4 g + 3
  const bool full = y0 + 8 <= H && x0 + 32 <= W;       // (uniform; edge tiles mask per element)
  const int cmax = W - x0 - 4 * h;                      // column offsets 8 q + (0 .. 3) below this are inside the image
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int yy = y0 + 2 * wave + m;
    float* ybase = y + ((size_t)(b * H + yy) * W + x0 + 4 * h + (i & 3)) * COUT + (i & ~3);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v0 = out[m][4 * q], v1 = out[m][4 * q + 1], v2 = out[m][4 * q + 2], v3 = out[m][4 * q + 3];
      sgg_quad_transpose4(v0, v1, v2, v3, lane);
#ifdef C3_ABL_NOSTORE
      if (v0 == 12345.678f)
#endif
      if (full || (yy < H && 8 * q + (i & 3) < cmax)) c3_out_store4(ybase + 8 * q * COUT, f32x4{v0, v1, v2, v3});
    }
  }
  if (tile_stats) {
    // one record per WAVE (its two rows of the tile): no workgroup reduction, no barrier.  (Measured alone at batch 64: 90 us
    // without the partials, 109 - 114 us with them in this form, 125 us with one record per tile - two barriers and an LDS
    // reduction -, 122 us with one record per row.)
    const int rows_ok = max(0, min(2, H - (y0 + 2 * wave))), cols_ok = min(32, W - x0);
    const float cnt = (float)(rows_ok * cols_ok * COUT);
    auto ok = [&](int m, int r) { return y0 + 2 * wave + m < H && (r & 3) + 8 * (r >> 2) < cmax; };
    float sm = 0.f;
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sm += out[0][r] + out[1][r];
    } else {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ok(m, r)) sm += out[m][r];
    }
    sm = wave_sum_dpp(sm);
    const float mean_w = cnt > 0.f ? sm / cnt : 0.f;
    float q = 0.f, dm = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (full || ok(m, r)) {
          const float d = out[m][r] - mean_w;
          q += d * d;
          dm = fmaxf(dm, fabsf(d));
        }
    q = wave_sum_dpp(q);
    dm = wave_max_dpp(dm);
    if (lane == 0) {
      float* o = tile_stats + ((((size_t)b * tiles_y + ty) * tiles_x + tx) * 4 + wave) * SGG_TS;
      o[0] = cnt;
      o[1] = mean_w;
      o[2] = q;
      o[3] = dm;
    }
  }
  }
}

// HWIO [taps][cin][cout] -> HWOI [taps][cout][cin]
__global__ void hwio_to_hwoi_kernel(const float* __restrict__ w, float* __restrict__ wt, int taps, int cin, int cout) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    tile[r][tx] = (ci < cin && co < cout) ? w[((size_t)tap * cin + ci) * cout + co] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    if (ci < cin && co < cout) wt[((size_t)tap * cout + co) * cin + ci] = tile[tx][r];
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN>
static void launch_gather(const GatherParams& p, hipStream_t st) {
  int maxwg = 0;
  GatherParams q = p;
  for (int i = 0; i < q.ncls; ++i) {
    q.cls[i].mtiles = sgg_cdiv(q.cls[i].M, BM);
    const int nwg = q.cls[i].mtiles * (q.N / BN);
    if (nwg > maxwg) maxwg = nwg;
  }
  dim3 grid(maxwg, q.ncls, 1);
  hipLaunchKernelGGL((conv_gather_kernel<BM, BN, WGM, WGN>), grid, dim3(256), 0, st, q);
}

template <int BM, int BN, int WGM, int WGN, int BK>
static void launch_gather3(const GatherParams& p, hipStream_t st) {
  int maxwg = 0;
  GatherParams q = p;
  for (int i = 0; i < q.ncls; ++i) {
    q.cls[i].mtiles = sgg_cdiv(q.cls[i].M, BM);
    const int nwg = q.cls[i].mtiles * (q.N / BN);
    if (nwg > maxwg) maxwg = nwg;
  }
  const unsigned src_bytes = (unsigned)((size_t)q.B * q.Hs * q.Ws * q.C * sizeof(float));
  dim3 grid(maxwg, q.ncls, 1);
  hipLaunchKernelGGL((conv_gather3_kernel<BM, BN, WGM, WGN, BK>), grid, dim3(256), 0, st, q, src_bytes);
}

template <int BM, int BN, int WGM, int WGN, int P, bool HALF, int BK = 32>
static void launch_gather_bf16s(const GatherParams& p, hipStream_t st) {
  int maxwg = 0;
  GatherParams q = p;
  for (int i = 0; i < q.ncls; ++i) {
    q.cls[i].mtiles = sgg_cdiv(q.cls[i].M, BM);
    const int nwg = q.cls[i].mtiles * (q.N / BN);
    if (nwg > maxwg) maxwg = nwg;
  }
  const unsigned src_bytes = (unsigned)((size_t)q.B * q.Hs * q.Ws * q.C * sizeof(float));
  dim3 grid(maxwg, q.ncls, 1);
  if (q.w_split && (size_t)(q.w_bytes / 2) * P < 0x80000000ull)
    hipLaunchKernelGGL((conv_gather_bf16s_kernel<BM, BN, WGM, WGN, P, true, HALF, BK>), grid, dim3(256), 0, st, q, src_bytes);
  else
    hipLaunchKernelGGL((conv_gather_bf16s_kernel<BM, BN, WGM, WGN, P, false, HALF, BK>), grid, dim3(256), 0, st, q, src_bytes);
}

// precision: 0 = native f32 MFMA; 2 = scaled f16 pieces, 3 products; 3 / 6 = bf16 pieces, 3 / 6 products
static int dispatch_gather(const GatherParams& p, hipStream_t st, int precision) {
  precision = sgg_prec_general(precision);      // (no single-piece variant of the gather kernel: modes 1 / 4 run as 2 / 3 here)
  const bool small_ = (size_t)p.B * p.Hs * p.Ws * p.C * sizeof(float) < 0x80000000ull && p.w_bytes < 0x80000000u;
  if (precision != 0 && small_) {
#define SGG_GB(BM, BN, WGM, WGN)                                                        \
  do {                                                                                  \
    if (precision == 2) launch_gather_bf16s<BM, BN, WGM, WGN, 2, true>(p, st);          \
    else if (precision == 3) launch_gather_bf16s<BM, BN, WGM, WGN, 2, false>(p, st);    \
    else launch_gather_bf16s<BM, BN, WGM, WGN, 3, false>(p, st);                        \
  } while (0)
    // (64-deep slabs, BK = 64: 2 instead of 3 waves per SIMD, measured 3 % slower in every mode)
    // (a 128 x 256 tile for N >= 256: 242 VGPRs, 2 waves per SIMD, measured +-3 % - not dispatched)
    if (p.N % 128 == 0) SGG_GB(128, 128, 2, 2);
    else if (p.N % 64 == 0) SGG_GB(256, 64, 4, 1);
    else SGG_GB(256, 32, 4, 1);
#undef SGG_GB
    return SGG_OK;
  }
  // buffer-load path: byte offsets must stay below the out-of-range marker 2^31
  const bool small = (size_t)p.B * p.Hs * p.Ws * p.C * sizeof(float) < 0x80000000ull && p.w_bytes < 0x80000000u;
  if (p.N % 128 == 0 && small)
    launch_gather3<128, 128, 2, 2, 32>(p, st);   // 64-deep slabs measured 3 % slower (2 instead of 3 waves per SIMD)
  else if (p.N % 64 != 0 && small)
    launch_gather3<256, 32, 4, 1, 32>(p, st);     // (N = 64: the v1 kernel measured 95 vs 89 TFLOP/s, kept below)
  else if (p.N % 128 == 0)
    launch_gather<128, 128, 2, 2>(p, st);
  else if (p.N % 64 == 0)
    launch_gather<256, 64, 4, 1>(p, st);
  else
    launch_gather<256, 32, 4, 1>(p, st);
  return SGG_OK;
}

// f32 [n] -> P planes of 16-bit pieces [P][n]: the operand format of conv_gather_bf16s_kernel<.., WS = true>
template <int P, bool HALF>
__global__ void split_weights_kernel(const float* __restrict__ in, unsigned* __restrict__ out, long long n8,
                                     const float* __restrict__ amax) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float scale = HALF ? ldexpf(1.f, scale_exp_from_amax(*amax)) : 1.f;
  const f32x4 v0 = reinterpret_cast<const f32x4*>(in)[2 * i], v1 = reinterpret_cast<const f32x4*>(in)[2 * i + 1];
  u32x4 pl[P];
  split8<P, HALF>(v0, v1, scale, pl);
#pragma unroll
  for (int pp = 0; pp < P; ++pp) reinterpret_cast<u32x4*>(out)[(long long)pp * n8 + i] = pl[pp];
}

// `amax` (device word with max|w|, see sgg_absmax) is required for precision 2 and ignored otherwise
extern "C" int sgg_conv_split_weights(const float* in, void* out, long long n, int precision, const float* amax, void* stream) {
  precision = sgg_prec_general(precision);
  SGG_CHECK_ARG(in && out && n > 0 && n % 8 == 0 && (precision == 2 || precision == 3 || precision == 6) && (precision != 2 || amax),
                "sgg_conv_split_weights: bad argument");
  const long long n8 = n / 8;
  const dim3 grid(sgg_cdiv(n8, 256)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (precision == 2) hipLaunchKernelGGL((split_weights_kernel<2, true>), grid, blk, 0, st, in, (unsigned*)out, n8, amax);
  else if (precision == 3) hipLaunchKernelGGL((split_weights_kernel<2, false>), grid, blk, 0, st, in, (unsigned*)out, n8, amax);
  else hipLaunchKernelGGL((split_weights_kernel<3, false>), grid, blk, 0, st, in, (unsigned*)out, n8, amax);
  SGG_LAUNCH_CHECK("sgg_conv_split_weights");
  return SGG_OK;
}

extern "C" int sgg_hwio_to_hwoi(const float* w, float* wt, int taps, int cin, int cout, void* stream) {
  SGG_CHECK_ARG(w && wt && taps > 0 && cin > 0 && cout > 0, "sgg_hwio_to_hwoi: bad argument");
  dim3 grid(sgg_cdiv(cout, 32), sgg_cdiv(cin, 32), taps);
  hipLaunchKernelGGL(hwio_to_hwoi_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, taps, cin, cout);
  SGG_LAUNCH_CHECK("sgg_hwio_to_hwoi");
  return SGG_OK;
}

// Number of (count, mean, M2) triples per sample the forward conv can emit for the following LayerNorm
// (0 = not available for this shape / precision: use the LayerNorm's own statistics pass).
extern "C" int sgg_conv2d_nhwc_fwd_tile_stats(int Ho, int Wo, int Cin, int Cout, int KH, int KW, int stride, int precision,
                                              int w_split_layout) {
  if (Cin == 3) return (KH == 3 && KW == 3 && stride == 1 && Cout == 32) ? 4 * sgg_cdiv(Ho, 8) * sgg_cdiv(Wo, 32) : 0;   // any precision: one per wave
  if (precision == 0 || Cin % 32 != 0 || Cout % 32 != 0) return 0;
  if (w_split_layout == 1 || w_split_layout == 4) {
    if (!sgg_halo_applicable(KH, KW, stride, Ho, Wo, Cin, Cout, precision)) return 0;
    // (the producer / consumer kernel of layout 4 always emits one partial per block and 64 columns, whatever tiling the
    // four-wave kernel was built with)
    return (Ho * Wo / 64) * (Cout / (w_split_layout == 4 ? 64 : sgg_halo_stats_cols(Cout)));
  }
  if (w_split_layout == 2) {
    if (!sgg_s2_applicable(KH, KW, stride, 1, 2 * Ho, 2 * Wo, Cin, Cout, precision)) return 0;
    return sgg_s2_stats_per_sample(Ho, Wo, Cout);
  }
  if (w_split_layout == 3) {
    if (!sgg_s2d_applicable(KH, KW, stride, 2 * Ho, 2 * Wo, Cin, Cout, precision)) return 0;
    return (Ho * Wo / 64) * (Cout / sgg_halo_stats_cols(Cout));
  }
  const int bm = (Cout % 128 == 0) ? 128 : 256;
  const int bn = (Cout % 128 == 0) ? 128 : (Cout % 64 == 0 ? 64 : 32);
  if ((Ho * Wo) % bm != 0) return 0;
  return (Ho * Wo / bm) * (Cout / bn);
}

// Forward. `w` is the HWIO kernel for Cin == 3 and the HWOI transpose (sgg_hwio_to_hwoi) otherwise.
extern "C" int sgg_conv2d_nhwc_fwd(const float* x, const float* w, const void* w_split, const float* bias, float* y, int B, int Hi, int Wi,
                                   int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l,
                                   int precision, int w_split_layout, const float* amax_x, const float* amax_w, float* tile_stats,
                                   const float* ln_stats, const float* ln_gamma, const float* ln_beta, int operand_format, void* stream) {
  SGG_CHECK_ARG(x && w && bias && y, "sgg_conv2d_nhwc_fwd: null pointer");
  // bits 8 .. 13: launch hint for the persistent kernels - occupy at most this many of an XCD's 32 CUs (0 = all; include/sgg_hip.h)
  const int cu_cap = (operand_format >> 8) & 63;
  SGG_CHECK_ARG((operand_format & ~0x3f01) == 0 && cu_cap <= 32, "sgg_conv2d_nhwc_fwd: operand_format: bit 0 = pre-split x, bits 8 .. 13 = CUs per XCD (<= 32)");
  operand_format &= 1;
  SGG_CHECK_ARG(operand_format == 0 || (operand_format == 1 && sgg_prec_half(precision) && w_split_layout >= 1 && w_split_layout <= 4 &&
                                        !ln_stats && Cin != 3),
                "sgg_conv2d_nhwc_fwd: a pre-split (S16) x needs precision 1 / 2, a resident kernel (w_split_layout 1 .. 4) and no LN prologue");
  SGG_CHECK_ARG(!ln_stats || (w_split_layout >= 1 && w_split_layout <= 4 && ln_gamma && ln_beta && Cin <= 512),
                "sgg_conv2d_nhwc_fwd: the LN prologue needs w_split_layout 1 .. 4 (resident kernels), gamma, beta and Cin <= 512");
  SGG_CHECK_ARG(precision == 0 || (precision >= 1 && precision <= 4) || precision == 6, "sgg_conv2d_nhwc_fwd: precision must be 0, 1, 2, 3, 4 or 6");
  SGG_CHECK_ARG(!sgg_prec_half(precision) || Cin == 3 || (amax_x && amax_w), "sgg_conv2d_nhwc_fwd: precision 1 / 2 need the amax words");
  SGG_CHECK_ARG(!ln_stats || !sgg_prec_one(precision), "sgg_conv2d_nhwc_fwd: the LN prologue exists in the two-piece modes (2, 3) only");
  SGG_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && stride >= 1 && stride <= 2, "sgg_conv2d_nhwc_fwd: bad dims");
  SGG_CHECK_ARG(Ho == (Hi + stride - 1) / stride && Wo == (Wi + stride - 1) / stride,
                "sgg_conv2d_nhwc_fwd: Ho/Wo must be ceil(in/stride) (SAME padding)");
  SGG_CHECK_ARG((long long)B * Hi * Wi * Cin < (1LL << 31) && (long long)B * Ho * Wo * Cout < (1LL << 31),
                "sgg_conv2d_nhwc_fwd: tensor exceeds 2^31 elements");
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 3) {
    SGG_CHECK_ARG(KH == 3 && KW == 3 && stride == 1 && Cout == 32, "sgg_conv2d_nhwc_fwd: Cin=3 path needs 3x3 s1 Cout=32");
    const int tiles_x = sgg_cdiv(Wo, 32), tiles_y = sgg_cdiv(Ho, 8);
    const int ntiles = B * tiles_x * tiles_y;
    hipLaunchKernelGGL(conv_c3_fwd_kernel, dim3((unsigned)(ntiles < 2048 ? ntiles : 2048)), dim3(256), 0, st, x, w, bias, y, tile_stats,
                       Hi, Wi, pad_t, pad_l, tiles_x, tiles_y, ntiles);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_fwd(c3)");
    return SGG_OK;
  }
  SGG_CHECK_ARG(Cin % 32 == 0 && Cout % 32 == 0, "sgg_conv2d_nhwc_fwd: Cin and Cout must be multiples of 32 (or Cin == 3)");
  if (w_split_layout == 1 || w_split_layout == 4) {        // halo-resident 3x3 stride-1 kernels, weights in MFMA fragment order
    SGG_CHECK_ARG(w_split && sgg_halo_applicable(KH, KW, stride, Hi, Wi, Cin, Cout, precision) && pad_t == 1 && pad_l == 1,
                  "sgg_conv2d_nhwc_fwd: w_split_layout 1 / 4 needs 3x3 stride 1, H %% 8 == W %% 8 == 0, precision 2 or 3 (sgg_conv_wsplit_layout)");
    SGG_CHECK_ARG(w_split_layout != 4 || sgg_halo_pc_applicable(Cin, Cout, precision) ||
                      (sgg_halo_pc64_applicable(Cin, Cout, precision) && (operand_format & 1) && !ln_stats),
                  "sgg_conv2d_nhwc_fwd: w_split_layout 4 needs Cout %% 128 == 0, Cin %% 64 == 0, precision 2 or 3 (sgg_conv_wsplit_layout) - or "
                  "Cout %% 64 == 0 with a pre-split x in precision 2 (sgg_conv_wsplit_layout_presplit)");
    HaloParams h;
    h.src = x; h.wfrag = w_split; h.bias = bias; h.out = y; h.amax_src = amax_x; h.amax_w = amax_w; h.tile_stats = tile_stats;
    h.ln_stats = ln_stats; h.ln_gamma = ln_gamma; h.ln_beta = ln_beta;
    h.B = B; h.H = Hi; h.W = Wi; h.C = Cin; h.N = Cout; h.bh = Hi / 8; h.bw = Wi / 8; h.nblk = B * h.bh * h.bw; h.flip = 0;
    h.src_bytes = (unsigned)((size_t)B * Hi * Wi * Cin * sizeof(float));
    h.w_bytes = (unsigned)((size_t)9 * Cin * Cout * sizeof(float));
    sgg_halo_dense_strides(h);
    h.frag16 = w_split_layout == 4;
    h.src_s16 = operand_format & 1;
    h.cu_cap = cu_cap;
    SGG_CHECK_ARG((size_t)B * Hi * Wi * Cin * sizeof(float) < 0x80000000ull, "sgg_conv2d_nhwc_fwd: input exceeds 2 GiB");
    sgg_halo_launch(h, precision, st);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_fwd(halo)");
    return SGG_OK;
  }
  if (w_split_layout == 3) {        // 5x5 stride 2 over 32 channels = 3x3 stride 1 over the space-to-depth view of x (halo-resident kernel)
    SGG_CHECK_ARG(w_split && sgg_s2d_applicable(KH, KW, stride, Hi, Wi, Cin, Cout, precision) && pad_t == 1 && pad_l == 1,
                  "sgg_conv2d_nhwc_fwd: w_split_layout 3 needs 5x5 stride 2, 32 -> 32 channels, H %% 16 == W %% 16 == 0, precision 2 or 3 "
                  "(sgg_conv_wsplit_layout)");
    SGG_CHECK_ARG((size_t)B * Hi * Wi * Cin * sizeof(float) < 0x80000000ull, "sgg_conv2d_nhwc_fwd: input exceeds 2 GiB");
    HaloParams h;
    h.src = x; h.wfrag = w_split; h.bias = bias; h.out = y; h.amax_src = amax_x; h.amax_w = amax_w; h.tile_stats = tile_stats;
    h.ln_stats = ln_stats; h.ln_gamma = ln_gamma; h.ln_beta = ln_beta;
    h.B = B; h.H = Ho; h.W = Wo; h.C = 4 * Cin; h.N = Cout; h.bh = Ho / 8; h.bw = Wo / 8; h.nblk = B * h.bh * h.bw; h.flip = 0;
    h.src_bytes = (unsigned)((size_t)B * Hi * Wi * Cin * sizeof(float));
    h.w_bytes = (unsigned)((size_t)9 * 4 * Cin * Cout * sizeof(float));
    sgg_halo_dense_strides(h);
    h.in_rs = 2 * Wi * Cin; h.in_ps = 2 * Cin; h.in_cA = Wi * Cin; h.in_cB = Cin;     // chunk (qy, qx): x[2a + qy][2c + qx][0..32)
    h.ln_nc = Cin;
    h.src_s16 = operand_format & 1;
    h.cu_cap = cu_cap;
    sgg_halo_launch(h, precision, st);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_fwd(s2d)");
    return SGG_OK;
  }
  if (w_split_layout == 2) {        // band-resident 5x5 stride-2 kernel (conv_s2.hip), weights in MFMA fragment order (25 taps)
    SGG_CHECK_ARG(w_split && sgg_s2_applicable(KH, KW, stride, B, Hi, Wi, Cin, Cout, precision) && pad_t == 1 && pad_l == 1,
                  "sgg_conv2d_nhwc_fwd: w_split_layout 2 needs 5x5 stride 2 on an even grid, Cout %% 128 == 0, precision 2 or 3 "
                  "(sgg_conv_wsplit_layout)");
    SGG_CHECK_ARG((size_t)B * Hi * Wi * Cin * sizeof(float) < 0x80000000ull && (size_t)B * Ho * Wo * Cout * sizeof(float) < 0x80000000ull,
                  "sgg_conv2d_nhwc_fwd: tensor exceeds 2 GiB");
    SGG_CHECK_ARG(!tile_stats || sgg_s2_stats_per_sample(Ho, Wo, Cout) > 0, "sgg_conv2d_nhwc_fwd: tile_stats need Ho*Wo %% 224 == 0 here");
    S2Params q;
    q.src = x; q.wfrag = w_split; q.bias = bias; q.out = y; q.amax_src = amax_x; q.amax_w = amax_w; q.tile_stats = tile_stats;
    q.ln_stats = ln_stats; q.ln_gamma = ln_gamma; q.ln_beta = ln_beta;
    q.B = B; q.Ho = Ho; q.Wo = Wo; q.C = Cin; q.N = Cout; q.M = B * Ho * Wo; q.nbands = sgg_cdiv(q.M, 224); q.pitch = Wo;
    q.src_bytes = (unsigned)((size_t)B * Hi * Wi * Cin * sizeof(float));
    q.w_bytes = (unsigned)((size_t)25 * Cin * Cout * sizeof(float));
    q.src_s16 = operand_format & 1;
    q.cu_cap = cu_cap;
    sgg_s2_launch(q, 0, precision, st);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_fwd(s2)");
    return SGG_OK;
  }
  GatherParams p;
  p.src = x; p.wm = w; p.bias = bias; p.out = y; p.w_split = (precision != 0) ? w_split : nullptr;
  p.amax_src = amax_x; p.amax_w = amax_w;
  p.hw = Ho * Wo;
  p.tile_stats = nullptr;
  if (tile_stats && precision != 0) {
    const int bm = (Cout % 128 == 0) ? 128 : 256;
    SGG_CHECK_ARG((Ho * Wo) % bm == 0, "sgg_conv2d_nhwc_fwd: tile_stats needs Ho*Wo %% %d == 0 (see sgg_conv2d_nhwc_fwd_tile_stats)", bm);
    p.tile_stats = tile_stats;
  }
  p.B = B; p.Hs = Hi; p.Ws = Wi; p.C = Cin; p.N = Cout; p.Ho = Ho; p.Wo = Wo;
  p.sy = stride; p.sx = stride; p.dy = 1; p.dx = 1; p.osy = 1; p.osx = 1; p.KW = KW;
  p.ncls = 1;
  p.w_bytes = (unsigned)((size_t)KH * KW * Cin * Cout * sizeof(float));
  GatherClass& c = p.cls[0];
  c.Hm = Ho; c.Wm = Wo; c.M = B * Ho * Wo; c.nth = KH; c.ntw = KW; c.oy = -pad_t; c.ox = -pad_l;
  c.kh0 = 0; c.kw0 = 0; c.kstep = 1; c.ooy = 0; c.oox = 0; c.mtiles = 0;
  dispatch_gather(p, st, precision);
  SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_fwd");
  return SGG_OK;
}

// dgrad: dx[B,Hi,Wi,Cin] = conv-transpose of dy[B,Ho,Wo,Cout] with the HWIO kernel w (no bias).
extern "C" int sgg_conv2d_nhwc_dgrad(const float* dy, const float* w, const void* w_split, float* dx, int B, int Hi, int Wi, int Cin, int Ho,
                                     int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int precision,
                                     int w_split_layout, const float* amax_dy, const float* amax_w, int operand_format, void* stream) {
  SGG_CHECK_ARG(dy && w && dx, "sgg_conv2d_nhwc_dgrad: null pointer");
  SGG_CHECK_ARG(operand_format == 0 || (operand_format == 1 && sgg_prec_half(precision) && w_split_layout >= 1 && w_split_layout <= 4),
                "sgg_conv2d_nhwc_dgrad: a pre-split (S16) dy needs precision 1 / 2 and a resident kernel (w_split_layout 1 .. 4)");
  SGG_CHECK_ARG(precision == 0 || (precision >= 1 && precision <= 4) || precision == 6, "sgg_conv2d_nhwc_dgrad: precision must be 0, 1, 2, 3, 4 or 6");
  SGG_CHECK_ARG(!sgg_prec_half(precision) || (amax_dy && amax_w), "sgg_conv2d_nhwc_dgrad: precision 1 / 2 need the amax words");
  SGG_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && stride >= 1 && stride <= 2, "sgg_conv2d_nhwc_dgrad: bad dims");
  SGG_CHECK_ARG(Cin % 32 == 0 && Cout % 32 == 0, "sgg_conv2d_nhwc_dgrad: Cin and Cout must be multiples of 32");
  SGG_CHECK_ARG((long long)B * Hi * Wi * Cin < (1LL << 31) && (long long)B * Ho * Wo * Cout < (1LL << 31),
                "sgg_conv2d_nhwc_dgrad: tensor exceeds 2^31 elements");
  if (w_split_layout == 1 || w_split_layout == 4) {        // halo-resident 3x3 stride-1 kernels: dx = correlation of dy with the mirrored taps
    SGG_CHECK_ARG(w_split && sgg_halo_applicable(KH, KW, stride, Hi, Wi, Cout, Cin, precision) && pad_t == 1 && pad_l == 1 &&
                      Ho == Hi && Wo == Wi,
                  "sgg_conv2d_nhwc_dgrad: w_split_layout 1 / 4 needs 3x3 stride 1, H %% 8 == W %% 8 == 0, precision 2 or 3 (sgg_conv_wsplit_layout)");
    SGG_CHECK_ARG(w_split_layout != 4 || sgg_halo_pc_applicable(Cout, Cin, precision) ||
                      (sgg_halo_pc64_applicable(Cout, Cin, precision) && (operand_format & 1)),
                  "sgg_conv2d_nhwc_dgrad: w_split_layout 4 needs Cin %% 128 == 0, Cout %% 64 == 0, precision 2 or 3 (sgg_conv_wsplit_layout)");
    SGG_CHECK_ARG((size_t)B * Ho * Wo * Cout * sizeof(float) < 0x80000000ull, "sgg_conv2d_nhwc_dgrad: dy exceeds 2 GiB");
    HaloParams h;
    h.src = dy; h.wfrag = w_split; h.bias = nullptr; h.out = dx; h.amax_src = amax_dy; h.amax_w = amax_w; h.tile_stats = nullptr;
    h.ln_stats = nullptr; h.ln_gamma = nullptr; h.ln_beta = nullptr;
    h.B = B; h.H = Hi; h.W = Wi; h.C = Cout; h.N = Cin; h.bh = Hi / 8; h.bw = Wi / 8; h.nblk = B * h.bh * h.bw; h.flip = 1;
    h.src_bytes = (unsigned)((size_t)B * Ho * Wo * Cout * sizeof(float));
    h.w_bytes = (unsigned)((size_t)9 * Cin * Cout * sizeof(float));
    sgg_halo_dense_strides(h);
    h.frag16 = w_split_layout == 4;
    h.src_s16 = operand_format & 1;
    sgg_halo_launch(h, precision, (hipStream_t)stream);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_dgrad(halo)");
    return SGG_OK;
  }
  if (w_split_layout == 3) {        // dx through the space-to-depth view: 3x3 correlation of dy with the mirrored 9-tap kernel, 4*Cin outputs
    SGG_CHECK_ARG(w_split && sgg_s2d_applicable(KH, KW, stride, Hi, Wi, Cin, Cout, precision) && pad_t == 1 && pad_l == 1 &&
                      Hi == 2 * Ho && Wi == 2 * Wo,
                  "sgg_conv2d_nhwc_dgrad: w_split_layout 3 needs 5x5 stride 2, 32 -> 32 channels, H %% 16 == W %% 16 == 0, precision 2 or 3 "
                  "(sgg_conv_wsplit_layout)");
    HaloParams h;
    h.src = dy; h.wfrag = w_split; h.bias = nullptr; h.out = dx; h.amax_src = amax_dy; h.amax_w = amax_w; h.tile_stats = nullptr;
    h.ln_stats = nullptr; h.ln_gamma = nullptr; h.ln_beta = nullptr;
    h.B = B; h.H = Ho; h.W = Wo; h.C = Cout; h.N = 4 * Cin; h.bh = Ho / 8; h.bw = Wo / 8; h.nblk = B * h.bh * h.bw; h.flip = 1;
    h.src_bytes = (unsigned)((size_t)B * Ho * Wo * Cout * sizeof(float));
    h.w_bytes = (unsigned)((size_t)9 * 4 * Cin * Cout * sizeof(float));
    sgg_halo_dense_strides(h);
    h.out_rs = 2 * Wi * Cin; h.out_ps = 2 * Cin; h.out_nA = Wi * Cin; h.out_nB = Cin;  // group (qy, qx) -> dx[2a + qy][2c + qx][0..32)
    h.src_s16 = operand_format & 1;
    sgg_halo_launch(h, precision, (hipStream_t)stream);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_dgrad(s2d)");
    return SGG_OK;
  }
  if (w_split_layout == 2) {        // band-resident 5x5 stride-2 kernel: the four pixel parities of dx from resident dy patches
    SGG_CHECK_ARG(w_split && sgg_s2_applicable(KH, KW, stride, B, Hi, Wi, Cout, Cin, precision) && pad_t == 1 && pad_l == 1 &&
                      Hi == 2 * Ho && Wi == 2 * Wo,
                  "sgg_conv2d_nhwc_dgrad: w_split_layout 2 needs 5x5 stride 2 on an even grid, Cin %% 128 == 0, precision 2 or 3 "
                  "(sgg_conv_wsplit_layout)");
    SGG_CHECK_ARG((size_t)B * Hi * Wi * Cin * sizeof(float) < 0x80000000ull && (size_t)B * Ho * Wo * Cout * sizeof(float) < 0x80000000ull,
                  "sgg_conv2d_nhwc_dgrad: tensor exceeds 2 GiB");
    S2Params q;
    q.src = dy; q.wfrag = w_split; q.bias = nullptr; q.out = dx; q.amax_src = amax_dy; q.amax_w = amax_w; q.tile_stats = nullptr;
    q.ln_stats = nullptr; q.ln_gamma = nullptr; q.ln_beta = nullptr;
    q.B = B; q.Ho = Ho; q.Wo = Wo; q.C = Cout; q.N = Cin; q.M = B * Ho * Wo; q.nbands = sgg_cdiv(q.M, 224); q.pitch = Wo;
    q.src_bytes = (unsigned)((size_t)B * Ho * Wo * Cout * sizeof(float));
    q.w_bytes = (unsigned)((size_t)25 * Cin * Cout * sizeof(float));
    q.src_s16 = operand_format & 1;
    q.cu_cap = 0;
    sgg_s2_launch(q, 1, precision, (hipStream_t)stream);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_dgrad(s2)");
    return SGG_OK;
  }
  GatherParams p;
  p.src = dy; p.wm = w; p.bias = nullptr; p.out = dx; p.w_split = (precision != 0) ? w_split : nullptr;
  p.amax_src = amax_dy; p.amax_w = amax_w;
  p.hw = Hi * Wi;
  p.tile_stats = nullptr;
  p.B = B; p.Hs = Ho; p.Ws = Wo; p.C = Cout; p.N = Cin; p.Ho = Hi; p.Wo = Wi;
  p.sy = 1; p.sx = 1; p.dy = -1; p.dx = -1; p.osy = stride; p.osx = stride; p.KW = KW;
  p.ncls = stride * stride;
  p.w_bytes = (unsigned)((size_t)KH * KW * Cin * Cout * sizeof(float));
  for (int ph = 0; ph < stride; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      GatherClass& c = p.cls[ph * stride + pw];
      const int kh0 = (ph + pad_t) % stride, kw0 = (pw + pad_l) % stride;
      c.Hm = (Hi - ph + stride - 1) / stride;
      c.Wm = (Wi - pw + stride - 1) / stride;
      c.M = B * c.Hm * c.Wm;
      c.nth = (KH - kh0 + stride - 1) / stride;
      c.ntw = (KW - kw0 + stride - 1) / stride;
      c.oy = (ph + pad_t - kh0) / stride;
      c.ox = (pw + pad_l - kw0) / stride;
      c.kh0 = kh0; c.kw0 = kw0; c.kstep = stride; c.ooy = ph; c.oox = pw; c.mtiles = 0;
    }
  dispatch_gather(p, (hipStream_t)stream, precision);
  SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_dgrad");
  return SGG_OK;
}
