// Halo-resident 3x3 / stride-1 convolution for the split 16-bit modes (gfx950): forward and dgrad of
//   tf.layers.conv2d(kernel_size=3, strides=1, padding="same")   reference: architectures/generator_with_attention.py:31-57
// (conv1_2, conv2_1..conv2_4, conv3_1, conv3_2 and their Conv2DBackpropInput, train.py:265-266).
//
// The gather kernel (conv_gather.hip) re-stages the shifted input tile for every tap: 9x the global loads, 9x the
// f32 -> 16-bit split VALU work and two barriers per 32-deep slab; it is staging-bound (MFMA pipe 47 % busy).  Here
//   * a workgroup owns NB 8x8 output blocks; per 32-channel chunk the 10x10 input patch (halo included) of each
//     block is loaded, split into two 16-bit planes and written to LDS ONCE; the nine taps read their MFMA A
//     fragments from the resident patch at shifted pixel slots: no re-staging, no barrier inside a chunk;
//   * the weights are pre-arranged as MFMA B fragments (sgg_conv_split_weights_frag: one coalesced 1 KiB
//     buffer_load_b128 per fragment) and go L2 -> registers directly, prefetched one tap ahead: no LDS traffic
//     and no barrier for the B operand.
// Patch layout: plane[pp][block][slot = ry*12 + rx][32 k] with 64-B rows; the 16-B chunk index is XOR-swizzled by
// ((rx>>2)&1) | ((ry&1)<<1): the 16 lanes of one ds_read_b128 phase (two patch rows x 8 pixels) hit 16 distinct
// bank groups for every tap shift (row pitch 12 slots = 768 B aliases rows, the swizzle separates them).
// Out-of-image patch pixels are raw-buffer out-of-range loads -> zeros (TF SAME padding).
#include "split16.h"
#include "conv_halo.h"
#include <type_traits>
#include <stdlib.h>

#define HALO_PITCH 12
#define HALO_BLKB (10 * HALO_PITCH * 64)   // bytes of one plane of one block's patch

__device__ __forceinline__ int halo_sw(int ry, int rx) { return ((rx >> 2) & 1) | ((ry & 1) << 1); }

template <int NB, int BN, int WGM, int WGN, bool HALF, bool PREFETCH>
__global__ __launch_bounds__(256, 2) void conv_halo3_kernel(HaloParams p) {
  constexpr int P = 2;
  constexpr int BM = NB * 64, WN = BN / WGN, TM = 2, TN = WN / 32;
  static_assert(WGM * WGN == 4 && BM / WGM == 64 && WN % 32 == 0 && TN >= 1, "a wave owns one 8x8 block x WN columns");
  constexpr int PLANEB = NB * HALO_BLKB;
  constexpr int ITEMS = NB * 400;                       // (block, patch pixel, 8-channel group)
  constexpr int NPASS = (ITEMS + 255) / 256;
  // (PREFETCH: the staging offsets live in LDS - in registers they are spilled, and a scratch reload in the tap loop
  //  waits for every load in flight)
  __shared__ __attribute__((aligned(16))) unsigned char lds[P * PLANEB + BM * 4 + (PREFETCH ? NPASS * 1024 : 0)];
  int* out_off_s = reinterpret_cast<int*>(lds + P * PLANEB);
  unsigned* it_off_s = reinterpret_cast<unsigned*>(lds + P * PLANEB + BM * 4);

  const int ntiles_n = p.N / BN;
  const int mtiles = (p.nblk + NB - 1) / NB;
  const int nwg = mtiles * ntiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int mt = lid / ntiles_n, nt = lid % ntiles_n;
  const int n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wblk = wave / WGN, wn0 = (wave % WGN) * WN;
  const int bpi = p.bh * p.bw;

  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, p.w_bytes, 0x00020000);
  int ea = 0, eb = 0;
  if constexpr (HALF) {
    ea = scale_exp_from_amax(*p.amax_src);
    eb = scale_exp_from_amax(*p.amax_w);
  }
  const float sa = ldexpf(1.f, ea);

  // ---- staging plan: item -> (global byte offset of 8 channels of a patch pixel, LDS byte offset inside a plane) ----
  unsigned it_off[NPASS];
  int it_lds[NPASS];
#pragma unroll
  for (int j = 0; j < NPASS; ++j) {
    const int it = tid + 256 * j;
    const int blk = it / 400, r = it % 400;
    const int px = r >> 2, ch8 = r & 3;
    const int ry = px / 10, rx = px % 10;
    const int beta = mt * NB + blk;
    const bool valid = it < ITEMS;
    unsigned off = SGG_OOB;
    if (valid && beta < p.nblk) {
      const int b = beta / bpi, rem = beta % bpi;
      const int by = rem / p.bw, bx = rem % p.bw;
      const int yy = by * 8 - 1 + ry, xx = bx * 8 - 1 + rx;
      if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
        off = (unsigned)(((b * p.H + yy) * p.W + xx) * p.C + ch8 * 8) * 4u;
    }
    it_off[j] = off;
    if constexpr (PREFETCH) it_off_s[j * 256 + tid] = off;
    it_lds[j] = valid ? blk * HALO_BLKB + (ry * HALO_PITCH + rx) * 64 + ((ch8 ^ halo_sw(ry, rx)) << 4) : -1;
  }
  for (int r = tid; r < BM; r += 256) {
    const int beta = mt * NB + (r >> 6), ml = r & 63;
    int off = -1;
    if (beta < p.nblk) {
      const int b = beta / bpi, rem = beta % bpi;
      const int by = rem / p.bw, bx = rem % p.bw;
      off = ((b * p.H + by * 8 + (ml >> 3)) * p.W + bx * 8 + (ml & 7)) * p.N;
    }
    out_off_s[r] = off;
  }

  f32x4 pre[NPASS][2];
  auto stage_load = [&](int cc, bool dead) {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const unsigned o0 = (PREFETCH && cc > 0) ? it_off_s[j * 256 + tid] : it_off[j];
      const unsigned off = (o0 + (unsigned)cc * 128u) | (dead ? SGG_OOB : 0u);   // (the marker stays out of range)
      pre[j][0] = buf_load4(rs_src, off);
      pre[j][1] = buf_load4(rs_src, off + 16u);
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      u32x4 pl[P];
      split8<P, HALF>(pre[j][0], pre[j][1], sa, pl);
      if (it_lds[j] >= 0) {
#pragma unroll
        for (int pp = 0; pp < P; ++pp) *reinterpret_cast<u32x4*>(lds + pp * PLANEB + it_lds[j]) = pl[pp];
      }
    }
  };

  // ---- weights: B fragments straight from L2, layout [tap][chunk][n-tile of 32][k-step][plane][lane] x 16 B --------
  const int nch = p.C >> 5;
  const unsigned w_lane = (unsigned)((n0 + wn0) >> 5) * 4096u + (unsigned)lane * 16u;
  const unsigned w_slab = (unsigned)(p.N >> 5) * 4096u;
  u32x4 rb[2][TN][2][P];
  auto load_b = [&](auto par_c, int cc, int tap) {
    constexpr int par = decltype(par_c)::value;
    const unsigned base = (unsigned)(tap * nch + cc) * w_slab + w_lane;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
          rb[par][tn][ks][pp] = __builtin_bit_cast(u32x4, buf_load4(rs_w, base + (unsigned)(tn * 4096 + ks * 2048 + pp * 1024)));
  };

  f32x16 acc[TM][TN];
  acc_zero<TM, TN>(acc);

  const int i = lane & 31, h = lane >> 5;
  const int pyl = i >> 3, pxl = i & 7;
  const unsigned char* patch_w = lds + wblk * HALO_BLKB;

  // One 32-channel chunk = nine taps, statically unrolled and branch free, so that the compiler's s_waitcnt counts are
  // exact: every tap issues the next tap's eight B-fragment loads and waits only for its own (issued one tap earlier).
  // (With the loads under a uniform branch the counter analysis had to assume the fewest loads in flight and every tap
  // waited for the loads it had just issued: the whole L2 latency exposed per tap.)  Past the end the B prefetch
  // re-reads a valid fragment and the patch prefetch uses out-of-range offsets (zeros, no memory traffic).
  // The A fragments of tap t+1 are read from LDS between the two k-steps of tap t (software pipelining inside the wave:
  // with two waves per SIMD the other wave alone does not cover the LDS latency).
  u32x4 a[2][TM][2][P];
  auto read_a = [&](auto buf_c, int tap) {
    constexpr int buf = decltype(buf_c)::value;
    const int kh = tap / 3, kw = tap % 3;
    const int dyy = p.flip ? 2 - kh : kh, dxx = p.flip ? 2 - kw : kw;
    // (opaque to the optimiser: otherwise the 36 per-tap LDS addresses are hoisted out of the chunk loop and spilled,
    //  and each scratch reload drags a vmcnt wait for the B prefetch in flight)
    int pyv = pyl, pxv = pxl;
    asm volatile("" : "+v"(pyv), "+v"(pxv));
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int ry = tm * 4 + pyv + dyy, rx = pxv + dxx;
      const unsigned char* row = patch_w + (ry * HALO_PITCH + rx) * 64;
      const int hs = halo_sw(ry, rx);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
          a[buf][tm][ks][pp] = *reinterpret_cast<const u32x4*>(row + pp * PLANEB + (((2 * ks + h) ^ hs) << 4));
    }
  };
  auto mma_kstep = [&](auto par_c, auto ks_c) {
    constexpr int par = decltype(par_c)::value, ks = decltype(ks_c)::value;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        f32x16 d = acc[tm][tn];
        d = mfma16<HALF>(a[par][tm][ks][1], rb[par][tn][ks][0], d);
        d = mfma16<HALF>(a[par][tm][ks][0], rb[par][tn][ks][1], d);
        d = mfma16<HALF>(a[par][tm][ks][0], rb[par][tn][ks][0], d);
        acc[tm][tn] = d;
      }
  };
  auto tap_body = [&](auto par_c, auto tap_c, int cc, bool more) {
    constexpr int par = decltype(par_c)::value, tap = decltype(tap_c)::value;
    const int ntap = tap == 8 ? (more ? 0 : 8) : tap + 1;
    const int ncc = tap == 8 ? (more ? cc + 1 : cc) : cc;
    load_b(std::integral_constant<int, par ^ 1>{}, ncc, ntap);
    if constexpr (PREFETCH && tap == 6) stage_load(cc + 1, !more);
    __builtin_amdgcn_sched_barrier(0);
    mma_kstep(par_c, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (tap < 8) read_a(std::integral_constant<int, par ^ 1>{}, tap + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_kstep(par_c, std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  auto chunk = [&](auto par0_c, int cc) {
    constexpr int par0 = decltype(par0_c)::value;
    const bool more = cc + 1 < nch;
    read_a(par0_c, 0);
#define SGG_TAP(T) tap_body(std::integral_constant<int, (par0 + T) & 1>{}, std::integral_constant<int, T>{}, cc, more)
    SGG_TAP(0); SGG_TAP(1); SGG_TAP(2); SGG_TAP(3); SGG_TAP(4); SGG_TAP(5); SGG_TAP(6); SGG_TAP(7); SGG_TAP(8);
#undef SGG_TAP
    // next chunk: replace the resident patch (zeros after the last chunk; nobody reads them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (!PREFETCH) stage_load(cc + 1, !more);
    stage_write();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  stage_load(0, false);
  load_b(std::integral_constant<int, 0>{}, 0, 0);
  stage_write();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int cc = 0;
  for (; cc + 1 < nch; cc += 2) {
    chunk(std::integral_constant<int, 0>{}, cc);
    chunk(std::integral_constant<int, 1>{}, cc + 1);
  }
  if (cc < nch) chunk(std::integral_constant<int, 0>{}, cc);

  // ---- epilogue: unscale, + bias, store; optionally this wave's LayerNorm partial statistics --------------------
  float lsum = 0.f;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 32 + acc_col(lane);
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wblk * 64 + tm * 32 + acc_row(r, lane);
        const int off = out_off_s[row];
        const float v = HALF ? ldexpf(ldexpf(acc[tm][tn][r], -ea), -eb) + bv : acc[tm][tn][r] + bv;
        acc[tm][tn][r] = v;
        lsum += v;
        if (off >= 0) p.out[(size_t)off + n] = v;
      }
    }
  }
  if (p.tile_stats) {
    // (count, mean, M2) of this wave's 64 pixels x WN channels (one 8x8 block: inside one sample); merged per sample
    // with Chan's formula by ln_apply_elu_kernel
    const float mean_w = wave_sum(lsum) * (1.f / (float)(64 * WN));
    float q = 0.f;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[tm][tn][r] - mean_w;
          q += d * d;
        }
    q = wave_sum(q);
    const int beta = mt * NB + wblk;
    if (lane == 0 && beta < p.nblk) {
      float* o = p.tile_stats + ((size_t)beta * (p.N / WN) + (n0 + wn0) / WN) * 3;
      o[0] = (float)(64 * WN);
      o[1] = mean_w;
      o[2] = q;
    }
  }
}

// f32 [taps][N][C] -> two 16-bit planes in MFMA B-fragment order [tap][C/32][N/32][k-step][plane][lane] x 16 B:
// lane l of fragment (tap, chunk, n-tile, k-step) holds w[tap][n-tile*32 + (l&31)][chunk*32 + k-step*16 + 8*(l>>5) .. +8]
template <bool HALF>
__global__ void split_weights_frag_kernel(const float* __restrict__ in, u32x4* __restrict__ out, int taps, int N, int C,
                                          const float* __restrict__ amax) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nch = C >> 5, ntl = N >> 5;
  if (idx >= (long long)taps * nch * ntl * 128) return;
  const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 1);
  long long r = idx >> 7;
  const int ntile = (int)(r % ntl); r /= ntl;
  const int cc = (int)(r % nch);
  const int tap = (int)(r / nch);
  const int n = ntile * 32 + (lane & 31), k = cc * 32 + ks * 16 + 8 * (lane >> 5);
  const float* src = in + ((size_t)tap * N + n) * C + k;
  const float scale = HALF ? ldexpf(1.f, scale_exp_from_amax(*amax)) : 1.f;
  u32x4 pl[2];
  split8<2, HALF>(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), scale, pl);
  const size_t o = ((((size_t)(tap * nch + cc) * ntl + ntile) * 2 + ks) * 2) * 64 + lane;
  out[o] = pl[0];
  out[o + 64] = pl[1];
}

// ---- host ---------------------------------------------------------------------------------------------------
int sgg_halo_applicable(int KH, int KW, int stride, int H, int W, int C, int N, int precision) {
  return KH == 3 && KW == 3 && stride == 1 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && C % 32 == 0 && N % 32 == 0 &&
         (precision == 2 || precision == 3);
}

int sgg_halo_stats_cols(int N) { return (N % 64 == 0) ? 64 : 32; }

void sgg_halo_launch(const HaloParams& p, int precision, hipStream_t st) {
  const bool half = precision == 2;
#define SGG_HALO(NB, BN, WGM, WGN, PF)                                                                       \
  do {                                                                                                       \
    const dim3 grid((unsigned)(sgg_cdiv(p.nblk, NB) * (p.N / BN)));                                          \
    if (half) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, true, PF>), grid, dim3(256), 0, st, p);  \
    else hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, false, PF>), grid, dim3(256), 0, st, p);      \
  } while (0)
  if (p.N % 128 == 0) SGG_HALO(2, 128, 2, 2, true);
  else if (p.N % 64 == 0) SGG_HALO(4, 64, 4, 1, false);
  else SGG_HALO(4, 32, 4, 1, false);
#undef SGG_HALO
}

// Which operand format sgg_conv2d_nhwc_fwd / _dgrad want for the pre-split weights of this convolution:
// 0 = planes [P][taps*N*C] (sgg_conv_split_weights), 1 = MFMA fragment order (sgg_conv_split_weights_frag; the
// halo-resident 3x3 stride-1 kernel).  H, W: the (identical) input and output grid of a stride-1 convolution.
extern "C" int sgg_conv_wsplit_layout(int KH, int KW, int stride, int H, int W, int Cin, int Cout, int precision) {
  return sgg_halo_applicable(KH, KW, stride, H, W, Cin, Cout, precision);
}

// in: f32 [taps][N][C] (forward: the HWOI transpose, N = Cout, C = Cin; dgrad: the HWIO kernel, N = Cin, C = Cout)
// out: taps*N*C*4 bytes.  precision 2 needs `amax` (device word with max|w|).
extern "C" int sgg_conv_split_weights_frag(const float* in, void* out, int taps, int N, int C, int precision, const float* amax,
                                           void* stream) {
  SGG_CHECK_ARG(in && out && taps > 0 && N > 0 && C > 0 && N % 32 == 0 && C % 32 == 0 && (precision == 2 || precision == 3) &&
                    (precision != 2 || amax),
                "sgg_conv_split_weights_frag: bad argument");
  const long long n = (long long)taps * (C / 32) * (N / 32) * 128;
  const dim3 grid(sgg_cdiv(n, 256)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (precision == 2) hipLaunchKernelGGL(split_weights_frag_kernel<true>, grid, blk, 0, st, in, (u32x4*)out, taps, N, C, amax);
  else hipLaunchKernelGGL(split_weights_frag_kernel<false>, grid, blk, 0, st, in, (u32x4*)out, taps, N, C, amax);
  SGG_LAUNCH_CHECK("sgg_conv_split_weights_frag");
  return SGG_OK;
}
