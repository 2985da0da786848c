// Conv2DBackpropFilter on f32 MFMA (gfx950): dW[kh][kw][ci][co] = sum_{b,ho,wo} x[b, ho*s-pt+kh, wo*s-pl+kw, ci] * dy[b,ho,wo,co]
// Reference: autodiff of tf.layers.conv2d (architectures/generator_with_attention.py:29-68) under
// optimizer.minimize (train.py:265-266).
//
// Per tap this is a GEMM  dW_tap[ci][co] = X_tap^T[ci][pix] * dY[pix][co]  whose contraction runs over
// B*Ho*Wo pixels (up to 3.2 M).  Decomposition: grid = (ci-tile x co-tile, tap, pixel split); each
// workgroup streams its pixel range in 32-pixel slabs (both operands are rows of contiguous channels, so
// the LDS tiles are in MC layout, mma_f32.h) and writes an f32 partial slab; a second kernel sums the
// slabs in a fixed order (deterministic, no atomics).
#include "mma_f32.h"
#include "conv_halo.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

// 8 floats -> P sixteen-bit pieces: bf16 (x = x0 + x1 (+ x2)) or, HALF, fp16 pieces of x*scale (see conv_gather_bf16s_kernel)
template <int P, bool HALF>
__device__ __forceinline__ void split8_regs(const float (&x)[8], float scale, u32x4 (&pl)[P]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a = x[2 * q], b = x[2 * q + 1];
    if constexpr (HALF) {
      a *= scale; b *= scale;
      unsigned hi, lo;
      f16_split2(a, b, hi, lo);
      pl[0][q] = hi; pl[1][q] = lo;
    } else {
#pragma unroll
      for (int pp = 0; pp < P; ++pp) {
        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2));
        pl[pp][q] = pk;
        if (pp + 1 < P) {
          a -= __builtin_bit_cast(float, pk << 16);
          b -= __builtin_bit_cast(float, pk & 0xffff0000u);
        }
      }
    }
  }
}

template <bool HALF>
__device__ __forceinline__ f32x16 mfma16w(const u32x4& a, const u32x4& b, f32x16 c) {
  if constexpr (HALF)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Split-bf16 contraction of MC tiles (f32 in LDS, [pixel][channel]): each wave gathers its fragments with the pixel
// (= k) index running down the LDS rows (ds_read_b32, conflict free), splits them in registers and issues
// 3 (P = 2) or 6 (P = 3) v_mfma_f32_32x32x16_bf16 per tile and 16-pixel step.  k_count must be a multiple of 16.
template <int TM, int TN, int P, bool HALF>
__device__ __forceinline__ void mma_slab_mc_mc_split(const float* __restrict__ A_s, int lda_s, const float* __restrict__ B_s,
                                                     int ldb_s, int wm0, int wn0, int k_begin, int k_count, int lane,
                                                     float sa, float sb, f32x16 (&acc)[TM][TN]) {
  const int i = lane & 31, h = lane >> 5;
  for (int k = k_begin; k < k_begin + k_count; k += 16) {
    u32x4 a[TM][P], b[TN][P];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = A_s[(k + 8 * h + j) * lda_s + wm0 + tm * 32 + i];
      split8_regs<P, HALF>(x, sa, a[tm]);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = B_s[(k + 8 * h + j) * ldb_s + wn0 + tn * 32 + i];
      split8_regs<P, HALF>(x, sb, b[tn]);
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        f32x16 d = acc[tm][tn];
        if constexpr (P == 3) {
          d = mfma16w<HALF>(a[tm][2], b[tn][0], d);
          d = mfma16w<HALF>(a[tm][0], b[tn][2], d);
          d = mfma16w<HALF>(a[tm][1], b[tn][1], d);
        }
        d = mfma16w<HALF>(a[tm][1], b[tn][0], d);
        d = mfma16w<HALF>(a[tm][0], b[tn][1], d);
        d = mfma16w<HALF>(a[tm][0], b[tn][0], d);
        acc[tm][tn] = d;
      }
  }
}

struct WgradParams {
  const float* x;
  const float* dy;
  float* out;  // [nsplit][taps][Cin][Cout]
  int B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l;
  int Mpix, chunk, ntile_n;
  unsigned magic_wo, magic_ho;   // ceil(2^32 / Wo), ceil(2^32 / Ho): exact division by __umulhi for our ranges
  unsigned x_bytes, dy_bytes;    // buffer descriptor bounds
  const float* amax_x;           // f16x3 mode: device words with max|x|, max|dy|
  const float* amax_dy;
};

template <int BMC, int BNC, int WM, int WN, int P, bool HALF>
__global__ __launch_bounds__(256, 3) void conv_wgrad_kernel(WgradParams p) {
  constexpr int WGM = BMC / WM, WGN = BNC / WN, WGK = 4 / (WGM * WGN);
  constexpr int TM = WM / 32, TN = WN / 32;
  // pixels per slab: a bf16 MFMA step is 16 pixels deep, so the 4-way k-split tiles (32x32 channels) take 64-pixel slabs
  constexpr int SLAB = (P != 0 && WGK == 4) ? 64 : 32;
  constexpr int NPA = SLAB * BMC / 1024, NPB = SLAB * BNC / 1024;
  constexpr int KROWS = SLAB / WGK;
  constexpr int TILE_FLOATS = SLAB * (BMC + BNC);
  constexpr int RED_FLOATS = (WGK > 1) ? 4 * TM * TN * 16 * 64 : 0;
  constexpr int LDS_FLOATS = TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS;
  static_assert(WGM * WGN * WGK == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* A_s = lds;
  float* B_s = lds + SLAB * BMC;

  const int tile = blockIdx.x;
  const int ci0 = (tile / p.ntile_n) * BMC, co0 = (tile % p.ntile_n) * BNC;
  const int tap = blockIdx.y;
  const int kh = tap / p.KW, kw = tap % p.KW;
  const int split = blockIdx.z;
  const int pix_begin = split * p.chunk;
  const int pix_end = min(pix_begin + p.chunk, p.Mpix);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WGM * WGN);
  const int wmi = (wave / WGN) % WGM, wni = wave % WGN;
  const int wm0 = wmi * WM, wn0 = wni * WN;

  f32x16 acc[TM][TN];
  acc_zero<TM, TN>(acc);
  int ea = 0, eb = 0;
  if constexpr (HALF) {
    ea = scale_exp_from_amax(*p.amax_x);
    eb = scale_exp_from_amax(*p.amax_dy);
  }
  const float sa = ldexpf(1.f, ea), sb = ldexpf(1.f, eb);

  constexpr int A_CPR = BMC / 4, B_CPR = BNC / 4;  // float4 per row
  // Branch-free slab fetch (same recipe as conv_gather3_kernel): raw buffer loads, rows past the pixel range or
  // outside the image get an out-of-range offset and read as zeros; pixel -> (b, ho, wo) by multiply-high.
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
  f32x4 ra[NPA], rb[NPB];
  auto issue_loads = [&](int pix0) {
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int pi = tid + 256 * j;
      const int row = pi / A_CPR, c4 = pi % A_CPR;
      const unsigned pix = (unsigned)(pix0 + row);
      const unsigned t = __umulhi(pix, p.magic_wo);          // pix / Wo
      const int wo = (int)(pix - t * (unsigned)p.Wo);
      const unsigned b = __umulhi(t, p.magic_ho);            // t / Ho
      const int ho = (int)(t - b * (unsigned)p.Ho);
      const int yy = ho * p.stride - p.pad_t + kh, xx = wo * p.stride - p.pad_l + kw;
      const unsigned bad = (unsigned)((int)pix >= pix_end) | (unsigned)((unsigned)yy >= (unsigned)p.Hi) |
                           (unsigned)((unsigned)xx >= (unsigned)p.Wi);
      const unsigned off = (unsigned)((((int)b * p.Hi + yy) * p.Wi + xx) * p.Cin + ci0 + c4 * 4) * 4u;
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off | ((0u - bad) & 0x80000000u), 0, 0));
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const int pi = tid + 256 * j;
      const int row = pi / B_CPR, c4 = pi % B_CPR;
      const int pix = pix0 + row;
      const unsigned bad = (unsigned)(pix >= pix_end);
      const unsigned off = (unsigned)(pix * p.Cout + co0 + c4 * 4) * 4u;
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off | ((0u - bad) & 0x80000000u), 0, 0));
    }
  };

  issue_loads(pix_begin);
  for (int pix0 = pix_begin; pix0 < pix_end; pix0 += SLAB) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < NPA; ++j) *reinterpret_cast<f32x4*>(A_s + (tid + 256 * j) * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < NPB; ++j) *reinterpret_cast<f32x4*>(B_s + (tid + 256 * j) * 4) = rb[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_loads(pix0 + SLAB);                             // past pix_end: zeros, no traffic
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P == 0) {
      mma_slab_mc_mc<TM, TN>(A_s, BMC, B_s, BNC, wm0, wn0, wk * KROWS, KROWS, lane, acc);
    } else {
      static_assert(P == 0 || KROWS % 16 == 0, "bf16 MFMA step is 16 pixels deep");
      mma_slab_mc_mc_split<TM, TN, P, HALF>(A_s, BMC, B_s, BNC, wm0, wn0, wk * KROWS, KROWS, lane, sa, sb, acc);
    }
  }

  if constexpr (HALF) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = ldexpf(ldexpf(acc[tm][tn][r], -ea), -eb);
  }
  if constexpr (WGK > 1) {
    // reduce the k-split waves through LDS: red[wave][slot][lane]
    __syncthreads();
    float* red = lds;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * TM * TN * 16 + (tm * TN + tn) * 16 + r) * 64 + lane] = acc[tm][tn][r];
    __syncthreads();
    if (wk != 0) return;
#pragma unroll
    for (int k = 1; k < WGK; ++k) {
      const int w2 = wave + k * WGM * WGN;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tm][tn][r] += red[(w2 * TM * TN * 16 + (tm * TN + tn) * 16 + r) * 64 + lane];
    }
  }

  float* o = p.out + ((size_t)split * p.KH * p.KW + tap) * p.Cin * p.Cout;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wm0 + tm * 32 + acc_row(r, lane);
        const int co = co0 + wn0 + tn * 32 + acc_col(lane);
        o[(size_t)ci * p.Cout + co] = acc[tm][tn][r];
      }
}

// ---------------------------------------------------------------------------------------------------
// 128 x 128 channel tiles in the split modes: split ONCE at the LDS write, transpose in the LDS read.
// The register-split kernel above converts every element in two waves (the 2x2 wave grid shares operand rows) and is
// VALU-bound (256 conversions + 64 ds_read_b32 per lane and slab against 768 cycles of MFMA).  Here each staged
// float4 (4 channels of one pixel) is split into P x 4 sixteen-bit pieces and written with ds_write_b64 into planes
// [32 pixels][128 channels] (256-B rows), and the MFMA fragments - 8 consecutive PIXELS of one channel per lane -
// are fetched with ds_read_b64_tr_b16 (gfx950 transposing LDS read: a 16-lane group reads a 4-pixel x 16-channel
// block and each lane receives one channel's 4 pixels; semantics verified by scripts/ubench/tr16_probe.hip).
// 16-B chunks are XOR-swizzled with f(row) = ((row&3)<<2) | ((row>>2)&3) (cdna_hip_programming.md T10, image (b)).
// ---------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define TR_ROWB 256

__device__ __forceinline__ int tr_off(int row, int chunk) {      // byte offset of 16-B chunk `chunk` of pixel row `row`
  return row * TR_ROWB + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}
__device__ __forceinline__ u32x2 ds_read_tr16(const unsigned char* p) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}

template <int P, bool HALF>
__global__ __launch_bounds__(256, 3) void conv_wgrad_tr_kernel(WgradParams p) {
  constexpr int TM = 2, TN = 2, NPA = 4, NPB = 4;
  constexpr int PLANE = 32 * TR_ROWB;            // bytes per plane (32 pixels x 128 channels x 2 B)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * P * PLANE];
  unsigned char* A_s = lds;
  unsigned char* B_s = lds + P * PLANE;

  const int tile = blockIdx.x;
  const int ci0 = (tile / p.ntile_n) * 128, co0 = (tile % p.ntile_n) * 128;
  const int tap = blockIdx.y;
  const int kh = tap / p.KW, kw = tap % p.KW;
  const int split = blockIdx.z;
  const int pix_begin = split * p.chunk;
  const int pix_end = min(pix_begin + p.chunk, p.Mpix);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

  f32x16 acc[TM][TN];
  acc_zero<TM, TN>(acc);
  int ea = 0, eb = 0;
  if constexpr (HALF) {
    ea = scale_exp_from_amax(*p.amax_x);
    eb = scale_exp_from_amax(*p.amax_dy);
  }
  const float sa = ldexpf(1.f, ea), sb = ldexpf(1.f, eb);

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
  f32x4 ra[NPA], rb[NPB];
  auto issue_loads = [&](int pix0) {
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int pi = tid + 256 * j;
      const int row = pi >> 5, c4 = pi & 31;
      const unsigned pix = (unsigned)(pix0 + row);
      const unsigned t = __umulhi(pix, p.magic_wo);
      const int wo = (int)(pix - t * (unsigned)p.Wo);
      const unsigned b = __umulhi(t, p.magic_ho);
      const int ho = (int)(t - b * (unsigned)p.Ho);
      const int yy = ho * p.stride - p.pad_t + kh, xx = wo * p.stride - p.pad_l + kw;
      const unsigned bad = (unsigned)((int)pix >= pix_end) | (unsigned)((unsigned)yy >= (unsigned)p.Hi) |
                           (unsigned)((unsigned)xx >= (unsigned)p.Wi);
      const unsigned off = (unsigned)((((int)b * p.Hi + yy) * p.Wi + xx) * p.Cin + ci0 + c4 * 4) * 4u;
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off | ((0u - bad) & 0x80000000u), 0, 0));
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const int pi = tid + 256 * j;
      const int row = pi >> 5, c4 = pi & 31;
      const int pix = pix0 + row;
      const unsigned bad = (unsigned)(pix >= pix_end);
      const unsigned off = (unsigned)(pix * p.Cout + co0 + c4 * 4) * 4u;
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off | ((0u - bad) & 0x80000000u), 0, 0));
    }
  };
  // 4 floats -> P x (4 sixteen-bit pieces = 8 B)
  auto split4 = [&](const f32x4& v, float scale, u32x2 (&pl)[P]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float a = v[2 * q], b = v[2 * q + 1];
      if constexpr (HALF) {
        a *= scale; b *= scale;
        unsigned hi, lo;
      f16_split2(a, b, hi, lo);
      pl[0][q] = hi; pl[1][q] = lo;
      } else {
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
          const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2));
          pl[pp][q] = pk;
          if (pp + 1 < P) {
            a -= __builtin_bit_cast(float, pk << 16);
            b -= __builtin_bit_cast(float, pk & 0xffff0000u);
          }
        }
      }
    }
  };

  issue_loads(pix_begin);
  // transposing-read lane roles: 16-lane group g -> pixel half h = g >> 1, channel half (g & 1); lane 4q+p of the group
  // supplies pixel row q (and q+4), channels 4p..4p+3 of the group's 16
  const int g = lane >> 4, q = (lane >> 2) & 3, pch = lane & 3;
  for (int pix0 = pix_begin; pix0 < pix_end; pix0 += 32) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int pi = tid + 256 * j;
      const int row = pi >> 5, c4 = pi & 31;
      u32x2 pl[P];
      split4(ra[j], sa, pl);
#pragma unroll
      for (int pp = 0; pp < P; ++pp) *reinterpret_cast<u32x2*>(A_s + pp * PLANE + tr_off(row, c4 >> 1) + (c4 & 1) * 8) = pl[pp];
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const int pi = tid + 256 * j;
      const int row = pi >> 5, c4 = pi & 31;
      u32x2 pl[P];
      split4(rb[j], sb, pl);
#pragma unroll
      for (int pp = 0; pp < P; ++pp) *reinterpret_cast<u32x2*>(B_s + pp * PLANE + tr_off(row, c4 >> 1) + (c4 & 1) * 8) = pl[pp];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_loads(pix0 + 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 a[TM][P], b[TN][P];
      const int r1 = 16 * ks + 8 * (g >> 1) + q, r2 = r1 + 4;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int ch = wm0 + tm * 32 + (g & 1) * 16 + 4 * pch;       // first of this lane's 4 address channels
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
          const u32x2 lo = ds_read_tr16(A_s + pp * PLANE + tr_off(r1, ch >> 3) + (ch & 7) * 2);
          const u32x2 hi = ds_read_tr16(A_s + pp * PLANE + tr_off(r2, ch >> 3) + (ch & 7) * 2);
          a[tm][pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int ch = wn0 + tn * 32 + (g & 1) * 16 + 4 * pch;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
          const u32x2 lo = ds_read_tr16(B_s + pp * PLANE + tr_off(r1, ch >> 3) + (ch & 7) * 2);
          const u32x2 hi = ds_read_tr16(B_s + pp * PLANE + tr_off(r2, ch >> 3) + (ch & 7) * 2);
          b[tn][pp] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          f32x16 d = acc[tm][tn];
          if constexpr (P == 3) {
            d = mfma16w<HALF>(a[tm][2], b[tn][0], d);
            d = mfma16w<HALF>(a[tm][0], b[tn][2], d);
            d = mfma16w<HALF>(a[tm][1], b[tn][1], d);
          }
          d = mfma16w<HALF>(a[tm][1], b[tn][0], d);
          d = mfma16w<HALF>(a[tm][0], b[tn][1], d);
          d = mfma16w<HALF>(a[tm][0], b[tn][0], d);
          acc[tm][tn] = d;
        }
    }
  }

  float* o = p.out + ((size_t)split * p.KH * p.KW + tap) * p.Cin * p.Cout;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int ci = ci0 + wm0 + tm * 32 + acc_row(rr, lane);
        const int co = co0 + wn0 + tn * 32 + acc_col(lane);
        const float v = acc[tm][tn][rr];
        o[(size_t)ci * p.Cout + co] = HALF ? ldexpf(ldexpf(v, -ea), -eb) : v;
      }
}

// out[e] = sum_s slabs[s][e]   (fixed order -> deterministic)
// A workgroup owns EPB float4 elements and splits the slabs over SG = 256 / EPB thread groups (group g sums slabs
// g, g + SG, ...), combined through LDS in group order: small dW tensors (conv1_2: 2304 float4 x 2048 slabs) would
// otherwise run on 9 workgroups with 2048 serial loads per thread.
template <int EPB>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n4,
                                                          int nsplit) {
  constexpr int SG = 256 / EPB;
  __shared__ f32x4 red[256];
  const int e = threadIdx.x % EPB, g = threadIdx.x / EPB;
  const long long i = (long long)blockIdx.x * EPB + e;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + i;
    int k = g;
    for (; k + 3 * SG < nsplit; k += 4 * SG) {      // four independent loads in flight
      const f32x4 v0 = src[(long long)k * n4], v1 = src[(long long)(k + SG) * n4];
      const f32x4 v2 = src[(long long)(k + 2 * SG) * n4], v3 = src[(long long)(k + 3 * SG) * n4];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < nsplit; k += SG) s += src[(long long)k * n4];
  }
  if constexpr (SG > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
#pragma unroll
      for (int j = 1; j < SG; ++j) s += red[j * EPB + e];
      reinterpret_cast<f32x4*>(out)[i] = s;
    }
  } else {
    if (i < n4) reinterpret_cast<f32x4*>(out)[i] = s;
  }
}

static void launch_slab_reduce(const float* slabs, float* out, long long n4, int nsplit, hipStream_t st) {
  // enough workgroups to fill the chip, and every thread group still sums several slabs
  if (n4 >= 256 * 1024 || nsplit < 8)
    hipLaunchKernelGGL(slab_reduce_kernel<256>, dim3(sgg_cdiv(n4, 256)), dim3(256), 0, st, slabs, out, n4, nsplit);
  else if (n4 >= 32 * 1024 || nsplit < 64)
    hipLaunchKernelGGL(slab_reduce_kernel<64>, dim3(sgg_cdiv(n4, 64)), dim3(256), 0, st, slabs, out, n4, nsplit);
  else
    hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3(sgg_cdiv(n4, 16)), dim3(256), 0, st, slabs, out, n4, nsplit);
}

// ---------------------------------------------------------------------------------------------------
// Cin = 3 (conv1_1): dW[27][32] over up to 3.2 M pixels; HBM-bound on dy (411 MB at batch 64 / 224^2).
// Same tiling as conv_c3_fwd_kernel (8 rows x 32 columns, the zero-padded 10 x 34 input patch in LDS as three channel planes of
// pitch 36); a wave owns two rows = 64 pixels and contracts them with 32 v_mfma_f32_32x32x2_f32 into ONE 32 x 32 accumulator tile
// dW[tap k][output channel] (16 registers, kept over all tiles of the workgroup): A[k][pixel] is a conflict-free ds_read_b32 at a
// per-lane tap offset (the 27 offsets fall into 27 different banks), B[pixel][channel] the wave's dy values - fetched a tile ahead
// with 16-byte loads and brought into the operand layout by the in-register quad transpose (sgg_common.h).  2048 MFMA cycles per wave and tile where the VALU form of rounds 1 - 3
// (108 FMAs per pixel and 4 channels) took 3456.  One LDS reduction of the four waves per workgroup at the end, then the
// deterministic slab reduce.  Sums: pixels in tile order per wave (exact f32 fmaf chains, mma_f32.h).
// ---------------------------------------------------------------------------------------------------
// workgroup barrier that orders LDS accesses only (__syncthreads() would also drain the prefetched global loads)
__device__ __forceinline__ void c3w_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// LNB (round 5): dy is not read - it is COMPUTED here, from the operands of the LayerNorm backward of conv1_1's output (the
// pre-LayerNorm y, the gradient da of the activation, gamma, beta, the per-sample (mean, rstd) and the two per-sample means m1 =
// mean(dxhat), m2 = mean(dxhat * xhat) of sgg_layernorm_hwc_elu_bwd_sums), with the arithmetic of ln_bwd_apply_kernel
// (csrc/layernorm.hip):  dy = rstd * (da * ELU'(n) * gamma - m1 - xhat * m2),  xhat = (y - mean) * rstd,  n = xhat * gamma + beta.
// conv1_1's filter gradient is the ONLY consumer of that dy (no image gradient is needed, the bias gradient comes from the
// LayerNorm reductions), so the LayerNorm backward's apply pass - 411 MB read twice and 411 MB written at batch 64, on the tail of
// every encoder backward where nothing else can run - and this kernel's 411 MB read of dy become ONE read of y and da.
struct C3LnArgs {
  const float* y;
  const float* da;
  const float* gamma;
  const float* beta;
  const float* stats;      // [B][2] (mean, rstd)
  const float* means;      // [B][2] (m1, m2)
};
// (LNB: 64 more prefetch registers - y beside da - do not fit three workgroups per CU without spilling: two)
template <bool LNB>
__global__ __launch_bounds__(256, LNB ? 2 : 3) void conv_c3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ slabs, int H, int W, int pt, int pl,
                                                               int tiles_x, int tiles_y, int ntiles, int tiles_per_wg, C3LnArgs ln) {
  constexpr int COUT = 32, PITCH = 36, PLANE = 10 * PITCH;
  __shared__ float patch[3 * PLANE];
  __shared__ float red[4 * 32 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  // A operand of this lane: tap k = i = (kh * 3 + kw) * 3 + ci (rows 27 .. 31 of the tile are never stored: any finite value)
  const int kk = i < 27 ? i : 0;
  const int abase = (kk % 3) * PLANE + (kk / 9) * PITCH + (kk / 3) % 3 + 2 * wave * PITCH + 4 * h;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(ntiles, t_begin + tiles_per_wg);
  // the NEXT tile's patch pixels and this wave's dy values are fetched into registers while this tile is computed
  float pv[2][3], dvn[2][16];
  // LNB: the prefetched y values (dvn then holds da), the validity of the lane's eight (row, column group) items, the tile's sample
  // constants, and this lane's four channels of gamma / beta
  float yvn[LNB ? 2 : 1][LNB ? 16 : 1];
  int okn = 0;
  float cstn[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 gmv = {0.f, 0.f, 0.f, 0.f}, btv = gmv;
  if constexpr (LNB) {
    gmv = *reinterpret_cast<const f32x4*>(ln.gamma + (i & ~3));
    btv = *reinterpret_cast<const f32x4*>(ln.beta + (i & ~3));
  }
  auto load_tile = [&](int t_) __attribute__((always_inline)) {
    const int tx_ = t_ % tiles_x, t2_ = t_ / tiles_x;
    const int ty_ = t2_ % tiles_y, b_ = t2_ / tiles_y;
    const int y0_ = ty_ * 8, x0_ = tx_ * 32;
    const bool live = t_ < t_end;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx / 34, c = idx % 34;
      const int yy = y0_ - pt + r, xx = x0_ - pl + c;
      pv[k][0] = pv[k][1] = pv[k][2] = 0.f;
      if (live && idx < 10 * 34 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
        const float* px = x + ((size_t)(b_ * H + yy) * W + xx) * 3;
        pv[k][0] = px[0]; pv[k][1] = px[1]; pv[k][2] = px[2];
      }
    }
    // 16-byte loads: lane (h, g = i >> 2, k = i & 3) fetches dy of pixel (row 2 wave + m, column 8 q + 4 h + k), channels 4 g .. 4 g + 3
    // (zeros outside the image: they contribute nothing); the quad transpose at the point of use turns them into this lane's
    // channel i at columns 8 q + 4 h + (0 .. 3) - the B operands of MFMAs (q, j), whose two pixels are columns 8 q + j and 8 q + 4 + j
    if constexpr (LNB) {
      okn = 0;
      const int bb = live ? b_ : 0;
      cstn[0] = ln.stats[2 * bb]; cstn[1] = ln.stats[2 * bb + 1]; cstn[2] = ln.means[2 * bb]; cstn[3] = ln.means[2 * bb + 1];
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int yy = y0_ + 2 * wave + m;
      const size_t roff = ((size_t)(b_ * H + yy) * W + x0_ + 4 * h + (i & 3)) * COUT + (i & ~3);
      const float* drow = (LNB ? ln.da : dy) + roff;
      const int cmax = W - x0_ - 4 * h - (i & 3);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const bool ok = live && yy < H && 8 * q < cmax;
        if (ok) v = *reinterpret_cast<const f32x4*>(drow + (size_t)(8 * q) * COUT);
        dvn[m][4 * q] = v[0]; dvn[m][4 * q + 1] = v[1]; dvn[m][4 * q + 2] = v[2]; dvn[m][4 * q + 3] = v[3];
        if constexpr (LNB) {
          f32x4 u = {0.f, 0.f, 0.f, 0.f};
          if (ok) u = *reinterpret_cast<const f32x4*>(ln.y + roff + (size_t)(8 * q) * COUT);
          yvn[m][4 * q] = u[0]; yvn[m][4 * q + 1] = u[1]; yvn[m][4 * q + 2] = u[2]; yvn[m][4 * q + 3] = u[3];
          okn |= (int)ok << (4 * m + q);
        }
      }
    }
  };
  load_tile(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    c3w_lds_barrier();
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
    float dv[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (LNB) {
          // dy of this lane's pixel (row 2 wave + m, column 8 q + 4 h + (i & 3)), channels 4 (i >> 2) .. + 3: ln_bwd_apply_kernel's arithmetic
          const float mean = cstn[0], rstd = cstn[1], m1 = cstn[2], m2 = cstn[3];
          const bool ok = (okn >> (4 * m + q)) & 1;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float xh = (yvn[m][4 * q + c] - mean) * rstd;
            const float n = xh * gmv[c] + btv[c];
            const float dn = dvn[m][4 * q + c] * (n > 0.f ? 1.f : __expf(n));
            const float o = rstd * (dn * gmv[c] - m1 - xh * m2);
            dv[m][4 * q + c] = ok ? o : 0.f;          // (outside the image: no pixel, no contribution)
          }
        } else {
          dv[m][4 * q] = dvn[m][4 * q]; dv[m][4 * q + 1] = dvn[m][4 * q + 1]; dv[m][4 * q + 2] = dvn[m][4 * q + 2]; dv[m][4 * q + 3] = dvn[m][4 * q + 3];
        }
        sgg_quad_transpose4(dv[m][4 * q], dv[m][4 * q + 1], dv[m][4 * q + 2], dv[m][4 * q + 3], lane);
      }
    c3w_lds_barrier();
    load_tile(t + 1);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma32(patch[abase + m * PITCH + 8 * q + j], dv[m][4 * q + j], acc);
  }
  // the four waves' tiles -> one [27][32] slab of the workgroup (register r of lane (i, h): tap acc_row(r, lane), channel i)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 32 + acc_row(r, lane)) * 32 + i] = acc[r];
  c3w_lds_barrier();
  for (int e = tid; e < 27 * COUT; e += 256)
    slabs[(size_t)blockIdx.x * 27 * COUT + e] = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
}

// ---------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------
struct WgradPlan {
  int bmc, bnc, tiles, nsplit, chunk;
  size_t ws_bytes;
};

#ifndef SGG_WGRAD_WGS
#define SGG_WGRAD_WGS 1536   // workgroups aimed at by the pixel split of the per-tap wgrad kernels
#endif
static WgradPlan wgrad_plan(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW) {
  WgradPlan pl;
  const long long mpix = (long long)B * Ho * Wo;
  if (Cin == 3) {        // tiles of 8 x 32 pixels, `chunk` tiles per workgroup, one slab per workgroup
    pl.bmc = 27; pl.bnc = 32;
    pl.tiles = B * sgg_cdiv(Ho, 8) * sgg_cdiv(Wo, 32);
    pl.chunk = sgg_cdiv(pl.tiles, 1024);
    pl.nsplit = sgg_cdiv(pl.tiles, pl.chunk);
    pl.ws_bytes = (size_t)pl.nsplit * 27 * 32 * sizeof(float);
    return pl;
  }
  pl.bmc = Cin >= 128 ? 128 : Cin;
  pl.bnc = Cout >= 128 ? 128 : Cout;
  pl.tiles = (Cin / pl.bmc) * (Cout / pl.bnc);
  const int base = pl.tiles * KH * KW;
  int ns = (SGG_WGRAD_WGS + base - 1) / base;
  const long long max_ns = mpix / 512 > 0 ? mpix / 512 : 1;
  if (ns > max_ns) ns = (int)max_ns;
  if (ns < 1) ns = 1;
  pl.chunk = (int)(((mpix + ns - 1) / ns + 31) / 32 * 32);
  pl.nsplit = (int)((mpix + pl.chunk - 1) / pl.chunk);
  pl.ws_bytes = pl.nsplit > 1 ? (size_t)pl.nsplit * KH * KW * Cin * Cout * sizeof(float) : 0;
  return pl;
}

extern "C" size_t sgg_conv2d_nhwc_wgrad_workspace_bytes(int B, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int KH,
                                                        int KW) {
  size_t need = wgrad_plan(B, Ho, Wo, Cin, Cout, KH, KW).ws_bytes;
  WgradHaloPlan hp;     // (precision and stride are not known here: upper bound over both kernels)
  const int st_guess = (Hi == Ho && Wi == Wo) ? 1 : ((Hi == 2 * Ho && Wi == 2 * Wo) ? 2 : 0);
  if (st_guess && Cin != 3 && sgg_wgrad_halo_plan(B, Ho, Wo, Cin, Cout, KH, KW, st_guess, &hp) && hp.ws_bytes > need) need = hp.ws_bytes;
  WgradDmaPlan dp;
  if (st_guess && Cin != 3 && sgg_wgrad_dma_plan(B, Ho, Wo, Cin, Cout, KH, KW, st_guess, &dp) && dp.ws_bytes > need) need = dp.ws_bytes;
  return need;
}

// 1: filter gradients whose operands are both pre-split run on the LDS-DMA kernel (conv_wgrad_dma.hip); 0 (-DSGG_WGRAD_DMA=0): always
// the halo-resident kernel (which stages pre-split operands through registers without arithmetic)
#ifndef SGG_WGRAD_DMA
#define SGG_WGRAD_DMA 1
#endif
// 0: per-tap kernels (or conv1_1's own); 1: the halo-resident kernel (takes pre-split operands, stages them through registers);
// 2: with BOTH operands pre-split in precision 2 the LDS-DMA kernel runs instead (conv_wgrad_dma.hip)
extern "C" int sgg_conv2d_nhwc_wgrad_resident(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW, int stride, int precision) {
  WgradHaloPlan hp;
  if (!(sgg_prec_resident(precision) && Cin != 3 && B > 0 && sgg_wgrad_halo_plan(B, Ho, Wo, Cin, Cout, KH, KW, stride, &hp))) return 0;
  WgradDmaPlan dp;
  return (SGG_WGRAD_DMA && precision == 2 && sgg_wgrad_dma_plan(B, Ho, Wo, Cin, Cout, KH, KW, stride, &dp)) ? 2 : 1;
}

extern "C" int sgg_conv2d_nhwc_wgrad(const float* x, const float* dy, float* dw, int B, int Hi, int Wi, int Cin, int Ho,
                                     int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int precision, int algo,
                                     const float* amax_x, const float* amax_dy, const float* ln_stats, const float* ln_gamma,
                                     const float* ln_beta, int operand_format, void* workspace, size_t workspace_bytes, void* stream) {
  SGG_CHECK_ARG(x && dy && dw, "sgg_conv2d_nhwc_wgrad: null pointer");
  SGG_CHECK_ARG(operand_format >= 0 && operand_format <= 3 && (operand_format == 0 || (sgg_prec_half(precision) && Cin != 3 && algo == 0)) &&
                    !((operand_format & 1) && ln_stats),
                "sgg_conv2d_nhwc_wgrad: pre-split (S16) operands need precision 1 / 2 and the halo-resident kernel (algo 0); x with an LN "
                "prologue is the f32 pre-LayerNorm tensor");
  SGG_CHECK_ARG(!ln_stats || (ln_gamma && ln_beta), "sgg_conv2d_nhwc_wgrad: the LN prologue needs stats, gamma and beta");
  SGG_CHECK_ARG(algo == 0 || algo == 1, "sgg_conv2d_nhwc_wgrad: algo must be 0 (auto) or 1 (per-tap kernels only)");
  SGG_CHECK_ARG(precision == 0 || (precision >= 1 && precision <= 4) || precision == 6,
                "sgg_conv2d_nhwc_wgrad: precision must be 0, 1, 2, 3, 4 or 6");
  SGG_CHECK_ARG(!sgg_prec_half(precision) || Cin == 3 || (amax_x && amax_dy), "sgg_conv2d_nhwc_wgrad: precision 1 / 2 need the amax words");
  SGG_CHECK_ARG(!ln_stats || !sgg_prec_one(precision), "sgg_conv2d_nhwc_wgrad: the LN prologue exists in the two-piece modes (2, 3) only");
  SGG_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && stride >= 1 && stride <= 2, "sgg_conv2d_nhwc_wgrad: bad dims");
  SGG_CHECK_ARG((long long)B * Hi * Wi * Cin < (1LL << 31) && (long long)B * Ho * Wo * Cout < (1LL << 31),
                "sgg_conv2d_nhwc_wgrad: tensor exceeds 2^31 elements");
  hipStream_t st = (hipStream_t)stream;
  const WgradPlan pl = wgrad_plan(B, Ho, Wo, Cin, Cout, KH, KW);
  if (pl.ws_bytes > 0) {
    if (!workspace || workspace_bytes < pl.ws_bytes) {
      sgg_set_error("sgg_conv2d_nhwc_wgrad: workspace too small (%zu < %zu)", workspace_bytes, pl.ws_bytes);
      return SGG_ERR_WORKSPACE;
    }
  }
  const long long nout = (long long)KH * KW * Cin * Cout;
  WgradDmaPlan dp;
  if (SGG_WGRAD_DMA && operand_format == 3 && precision == 2 && !ln_stats && algo == 0 && pad_t == 1 && pad_l == 1 && Hi == Ho * stride &&
      Wi == Wo * stride && sgg_wgrad_dma_plan(B, Ho, Wo, Cin, Cout, KH, KW, stride, &dp)) {
    // both operands pre-split: staged by LDS-DMA, no staging arithmetic (conv_wgrad_dma.hip)
    if (!workspace || workspace_bytes < dp.ws_bytes) {
      sgg_set_error("sgg_conv2d_nhwc_wgrad: workspace too small (%zu < %zu)", workspace_bytes, dp.ws_bytes);
      return SGG_ERR_WORKSPACE;
    }
    sgg_wgrad_dma_launch(x, dy, (float*)workspace, B, Ho, Wo, Cin, Cout, stride, pad_t, pad_l, amax_x, amax_dy, dp, st);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad(dma)");
    launch_slab_reduce((const float*)workspace, dw, nout / 4, dp.nslabs, st);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad(dma reduce)");
    return SGG_OK;
  }
  WgradHaloPlan hp;
  if (sgg_prec_resident(precision) && Cin != 3 && pad_t == 1 && pad_l == 1 && Hi == Ho * stride && Wi == Wo * stride &&
      algo == 0 && sgg_wgrad_halo_plan(B, Ho, Wo, Cin, Cout, KH, KW, stride, &hp) && !(hp.geo == 1 && ln_stats)) {
    // halo-resident kernel: the nine taps of a channel chunk from one LDS-resident patch (conv_wgrad_halo.hip)
    if (!workspace || workspace_bytes < hp.ws_bytes) {
      sgg_set_error("sgg_conv2d_nhwc_wgrad: workspace too small (%zu < %zu)", workspace_bytes, hp.ws_bytes);
      return SGG_ERR_WORKSPACE;
    }
    sgg_wgrad_halo_launch(x, dy, (float*)workspace, B, Ho, Wo, Cin, Cout, stride, pad_t, pad_l, precision, amax_x, amax_dy, hp, st,
                          ln_stats, ln_gamma, ln_beta, operand_format);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad(halo)");
    launch_slab_reduce((const float*)workspace, dw, nout / 4, hp.nslabs, st);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad(halo reduce)");
    return SGG_OK;
  }
  SGG_CHECK_ARG(operand_format == 0, "sgg_conv2d_nhwc_wgrad: pre-split (S16) operands are served by the halo-resident kernel only");
  SGG_CHECK_ARG(!ln_stats, "sgg_conv2d_nhwc_wgrad: the LN prologue is served by the halo-resident kernel only (3x3 stride 1 or 5x5 "
                           "stride 2 on grids divisible by 8, precision 2 or 3, algo 0)");
  if (Cin == 3) {
    SGG_CHECK_ARG(KH == 3 && KW == 3 && stride == 1 && Cout == 32, "sgg_conv2d_nhwc_wgrad: Cin=3 path needs 3x3 s1 Cout=32");
    hipLaunchKernelGGL(conv_c3_wgrad_kernel<false>, dim3(pl.nsplit), dim3(256), 0, st, x, dy, (float*)workspace, Hi, Wi, pad_t, pad_l,
                       sgg_cdiv(Wo, 32), sgg_cdiv(Ho, 8), pl.tiles, pl.chunk, C3LnArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad(c3)");
    launch_slab_reduce((const float*)workspace, dw, nout / 4, pl.nsplit, st);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad(c3 reduce)");
    return SGG_OK;
  }
  SGG_CHECK_ARG(Cin % 32 == 0 && Cout % 32 == 0, "sgg_conv2d_nhwc_wgrad: Cin and Cout must be multiples of 32 (or Cin == 3)");
  SGG_CHECK_ARG(pl.bmc == 32 || pl.bmc == 64 || pl.bmc == 128, "sgg_conv2d_nhwc_wgrad: unsupported Cin tile");
  SGG_CHECK_ARG(pl.bnc == 32 || pl.bnc == 64 || pl.bnc == 128, "sgg_conv2d_nhwc_wgrad: unsupported Cout tile");
  WgradParams p;
  p.x = x; p.dy = dy; p.out = pl.nsplit > 1 ? (float*)workspace : dw;
  p.B = B; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.Mpix = B * Ho * Wo; p.chunk = pl.chunk; p.ntile_n = Cout / pl.bnc;
  p.magic_wo = (unsigned)((0x100000000ull + Wo - 1) / Wo);
  p.magic_ho = (unsigned)((0x100000000ull + Ho - 1) / Ho);
  p.amax_x = amax_x; p.amax_dy = amax_dy;
  p.x_bytes = (unsigned)((size_t)B * Hi * Wi * Cin * sizeof(float));
  p.dy_bytes = (unsigned)((size_t)B * Ho * Wo * Cout * sizeof(float));
  // multiply-high division is exact while n * d < 2^32 (n = pixel index + 31, d = Wo; then n / Wo and Ho)
  SGG_CHECK_ARG(Ho >= 2 && Wo >= 2, "sgg_conv2d_nhwc_wgrad: output grid must be at least 2x2");
  SGG_CHECK_ARG(((unsigned long long)B * Ho * Wo + 64) * (unsigned long long)Wo < 0x100000000ull &&
                    (size_t)B * Hi * Wi * Cin * sizeof(float) < 0x80000000ull && (size_t)B * Ho * Wo * Cout * sizeof(float) < 0x80000000ull,
                "sgg_conv2d_nhwc_wgrad: tensor too large for the 32-bit offset path");
  dim3 grid(pl.tiles, KH * KW, pl.nsplit);
  precision = sgg_prec_general(precision);      // (the per-tap kernels have no single-piece variant: modes 1 / 4 run as 2 / 3 here)
#define SGG_WG(BMC, BNC, WM, WN)                                                                        \
  do {                                                                                                \
    if (precision == 0) hipLaunchKernelGGL((conv_wgrad_kernel<BMC, BNC, WM, WN, 0, false>), grid, dim3(256), 0, st, p);      \
    else if (precision == 2) hipLaunchKernelGGL((conv_wgrad_kernel<BMC, BNC, WM, WN, 2, true>), grid, dim3(256), 0, st, p);  \
    else if (precision == 3) hipLaunchKernelGGL((conv_wgrad_kernel<BMC, BNC, WM, WN, 2, false>), grid, dim3(256), 0, st, p); \
    else hipLaunchKernelGGL((conv_wgrad_kernel<BMC, BNC, WM, WN, 3, false>), grid, dim3(256), 0, st, p);                     \
  } while (0)
  if (pl.bmc == 32 && pl.bnc == 32) SGG_WG(32, 32, 32, 32);
  else if (pl.bmc == 32 && pl.bnc == 64) SGG_WG(32, 64, 32, 32);
  else if (pl.bmc == 64 && pl.bnc == 32) SGG_WG(64, 32, 32, 32);
  else if (pl.bmc == 64 && pl.bnc == 64) SGG_WG(64, 64, 32, 32);
  else if (pl.bmc == 32 && pl.bnc == 128) SGG_WG(32, 128, 32, 32);
  else if (pl.bmc == 128 && pl.bnc == 32) SGG_WG(128, 32, 32, 32);
  else if (pl.bmc == 64 && pl.bnc == 128) SGG_WG(64, 128, 32, 64);
  else if (pl.bmc == 128 && pl.bnc == 64) SGG_WG(128, 64, 64, 32);
  else if (precision == 2) hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, true>), grid, dim3(256), 0, st, p);
  else if (precision == 3) hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, false>), grid, dim3(256), 0, st, p);
  else if (precision == 6) hipLaunchKernelGGL((conv_wgrad_tr_kernel<3, false>), grid, dim3(256), 0, st, p);
  else SGG_WG(128, 128, 64, 64);
#undef SGG_WG
  SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad");
  if (pl.nsplit > 1) {
    launch_slab_reduce((const float*)workspace, dw, nout / 4, pl.nsplit, st);
    SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad(reduce)");
  }
  return SGG_OK;
}

// conv1_1's filter gradient FUSED with the apply half of the LayerNorm backward of its output (conv_c3_wgrad_kernel<true>):
//   dW = Conv2DBackpropFilter(x, dy),  dy = LayerNormBackward(y, da) of tf.contrib.layers.layer_norm(activation_fn=elu)
// (architectures/generator_with_attention.py:29-30 under optimizer.minimize, train.py:265-266).  means [B][2] comes from
// sgg_layernorm_hwc_elu_bwd_sums on the same (y, da); dy itself is never materialised.
extern "C" int sgg_conv2d_nhwc_wgrad_c3_ln(const float* x, const float* y, const float* da, const float* gamma, const float* beta,
                                           const float* stats, const float* means, float* dw, int B, int H, int W, int pad_t, int pad_l,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  SGG_CHECK_ARG(x && y && da && gamma && beta && stats && means && dw, "sgg_conv2d_nhwc_wgrad_c3_ln: null pointer");
  SGG_CHECK_ARG(B > 0 && H > 0 && W > 0 && pad_t == 1 && pad_l == 1, "sgg_conv2d_nhwc_wgrad_c3_ln: bad dims (3x3 stride 1, SAME padding)");
  SGG_CHECK_ARG((long long)B * H * W * 32 < (1LL << 31), "sgg_conv2d_nhwc_wgrad_c3_ln: tensor exceeds 2^31 elements");
  const WgradPlan pl = wgrad_plan(B, H, W, 3, 32, 3, 3);
  if (!workspace || workspace_bytes < pl.ws_bytes) {
    sgg_set_error("sgg_conv2d_nhwc_wgrad_c3_ln: workspace too small (%zu < %zu)", workspace_bytes, pl.ws_bytes);
    return SGG_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv_c3_wgrad_kernel<true>, dim3(pl.nsplit), dim3(256), 0, st, x, (const float*)nullptr, (float*)workspace, H, W, pad_t,
                     pad_l, sgg_cdiv(W, 32), sgg_cdiv(H, 8), pl.tiles, pl.chunk, C3LnArgs{y, da, gamma, beta, stats, means});
  SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad_c3_ln");
  launch_slab_reduce((const float*)workspace, dw, 27LL * 32 / 4, pl.nsplit, st);
  SGG_LAUNCH_CHECK("sgg_conv2d_nhwc_wgrad_c3_ln(reduce)");
  return SGG_OK;
}
