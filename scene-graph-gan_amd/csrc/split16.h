// Shared pieces of the split 16-bit convolution kernels (gfx950): raw buffer loads with an out-of-range marker,
// the f32 -> 16-bit piece split and the 32x32x16 MFMA wrappers.  Used by conv_gather.hip and conv_halo.hip.
#pragma once
#include "mma_f32.h"

#define SGG_OOB 0x80000000u

// activation patch loads of the resident convolution kernels: -DSGG_PATCH_LOAD_AUX=2 adds the nontemporal hint (measured: DESIGN.md section 8)
#ifndef SGG_PATCH_LOAD_AUX
#define SGG_PATCH_LOAD_AUX 0
#endif
#ifndef SGG_WGRAD_LOAD_AUX
#define SGG_WGRAD_LOAD_AUX 0
#endif
template <int AUX>
__device__ __forceinline__ f32x4 buf_load4_aux(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, AUX));
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

// 8 consecutive floats -> P planes of 8 sixteen-bit pieces (16 B each), x*scale = x0 + x1 (+ x2).
//   HALF = false: bf16 pieces (RNE), scale unused.   HALF = true: fp16 pieces (P = 2) of the pre-scaled value; the scale
//   is a power of two chosen from the tensor's max|x| so that |x*scale| <= 2^14 (no overflow, and the second piece
//   only reaches fp16 subnormals 17 binades below the tensor's maximum): both pieces round to nearest even (f16_split2), so an
//   element within 2^-16 of the maximum keeps 23 significant bits, one 2^-d below it min(23, 39 - d).
template <int P, bool HALF>
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, float scale, u32x4 (&pl)[P]) {
  float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a = x[2 * q], b = x[2 * q + 1];
    if constexpr (HALF && P == 1) {     // single-piece mode: the operand rounded to fp16 (11 significant bits)
      pl[0][q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a * scale, b * scale}, fp16x2));
    } else if constexpr (HALF) {
      a *= scale; b *= scale;
      unsigned hi, lo;
      f16_split2(a, b, hi, lo);
      pl[0][q] = hi; pl[1][q] = lo;
    } else {
#pragma unroll
      for (int pp = 0; pp < P; ++pp) {
        const unsigned pk = cvt_pk_bf16(a, b);
        pl[pp][q] = pk;
        if (pp + 1 < P) {
          a -= __builtin_bit_cast(float, pk << 16);
          b -= __builtin_bit_cast(float, pk & 0xffff0000u);
        }
      }
    }
  }
}

// ---- pre-split activations ("S16" tensors) -------------------------------------------------------------------------------
// The LayerNorm kernels can write their outputs (a = ELU(LN(y)); dy of the LayerNorm backward) already split: the tensor keeps
// its NHWC shape and byte size, but every aligned group of 32 channels (128 B) holds the 32 LEADING fp16 pieces (64 B) followed
// by the 32 RESIDUAL pieces (64 B) of x * 2^e = hi + lo - exactly what split8<2, true> produces, with e = scale_exp_from_amax of
// the tensor's amax word, which then holds an upper BOUND of max|x| fixed before the tensor is written (sgg_layernorm_hwc_*).
// A consumer stages such an operand without arithmetic: two 16-byte loads (pieces of 8 channels) go to the two LDS planes as they
// are.  Given the byte offset `off` of 8 consecutive channels in the f32 layout (32-byte aligned, inside one 128-byte group):
__device__ __forceinline__ unsigned s16_hi_off(unsigned off) { return (off & ~127u) | ((off & 127u) >> 1); }     // lo pieces: + 64
// (SGG_OOB stays out of range: its low seven bits are zero.)
// second load of an 8-channel item: the next 4 floats, or the residual pieces
__device__ __forceinline__ unsigned stage_off0(unsigned off, int s16) { return s16 ? s16_hi_off(off) : off; }
__device__ __forceinline__ unsigned stage_off1(unsigned off0, int s16) { return off0 + (s16 ? 64u : 16u); }
// 8 staged channels -> P planes: the split of f32 data, or the two loaded registers as they are
template <int P, bool HALF>
__device__ __forceinline__ void stage_planes(const f32x4& v0, const f32x4& v1, float scale, int s16, u32x4 (&pl)[P]) {
  if (s16) {
    pl[0] = __builtin_bit_cast(u32x4, v0);
    if constexpr (P == 2) pl[1] = __builtin_bit_cast(u32x4, v1);
  } else {
    split8<P, HALF>(v0, v1, scale, pl);
  }
}

template <bool HALF>
__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, f32x16 c) {
  if constexpr (HALF)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}


// the K = 32 shape (16 x 16 output tile, 4 accumulator registers, 16 cycles): same FLOPs per cycle; the chip holds a higher clock on it
typedef float f32x4_acc __attribute__((ext_vector_type(4)));
template <bool HALF>
__device__ __forceinline__ f32x4_acc mfma16k32(const u32x4& a, const u32x4& b, f32x4_acc c) {
  if constexpr (HALF)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// half of ln_elu8: 4 channels (the producer waves of conv_halo_pc.hip spread the prologue of an 8-channel item over two taps)
__device__ __forceinline__ void ln_elu4(f32x4& v0, const float* __restrict__ gam, const float* __restrict__ bet, float mean, float rstd,
                                        bool zero) {
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam), b0 = *reinterpret_cast<const f32x4*>(bet);
  const f32x4 i0 = g0 * rstd;
  const f32x4 s0 = b0 - i0 * mean;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float z0 = fmaf(v0[q], i0[q], s0[q]);
    const float e0 = __expf(z0) - 1.f;
    v0[q] = zero ? 0.f : (z0 > 0.f ? z0 : e0);
  }
}

// LN prologue of the patch staging: 8 channels of y -> ELU(y * inv_c + shift_c), inv_c = gamma_c * rstd, shift_c = beta_c - mean * inv_c
// (the arithmetic of ln_apply_elu_kernel; ELU's negative branch through v_exp_f32: |error| <= 1.2e-7 absolute).
// `zero`: the pixel lies outside the image - the padding is a zero ACTIVATION.
__device__ __forceinline__ void ln_elu8(f32x4& v0, f32x4& v1, const float* __restrict__ gam, const float* __restrict__ bet, float mean,
                                        float rstd, bool zero) {
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam), g1 = *reinterpret_cast<const f32x4*>(gam + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(bet), b1 = *reinterpret_cast<const f32x4*>(bet + 4);
  const f32x4 i0 = g0 * rstd, i1 = g1 * rstd;
  const f32x4 s0 = b0 - i0 * mean, s1 = b1 - i1 * mean;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float z0 = fmaf(v0[q], i0[q], s0[q]), z1 = fmaf(v1[q], i1[q], s1[q]);
    const float e0 = __expf(z0) - 1.f, e1 = __expf(z1) - 1.f;
    v0[q] = zero ? 0.f : (z0 > 0.f ? z0 : e0);
    v1[q] = zero ? 0.f : (z1 > 0.f ? z1 : e1);
  }
}
