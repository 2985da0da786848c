// f32-input MFMA tile core for gfx950: v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fmaf chain).
//
// Operand lane maps (cdna_hip_programming.md section 3): lane l, i = l & 31, h = l >> 5
//   A operand: A[i][k = h]      B operand: B[k = h][j = l & 31]
//   C/D: reg r (0..15) of lane l  ->  row (r&3) + 8*(r>>2) + 4*h, col l & 31
//
// Two LDS tile layouts are used by every contraction kernel in this library:
//   KC ("k contiguous"):  tile[row][LDK], LDK = 36 floats (32 k + 4 pad).  A lane reads a float4 at
//        [row][kb*8 + 4*h .. +3] (ds_read_b128, conflict free for stride 36: 36*i mod 64 = 4*(9i mod 16))
//        and feeds element jj to MFMA jj of the 8-k block, so MFMA jj contracts k = {kb*8+jj, kb*8+4+jj}.
//        The k order inside the block is permuted identically for A and B, which leaves the sum unchanged.
//   MC ("m/n contiguous"): tile[k][LD], a lane reads one float at [k0 + 2*jj + h][col0 + i] (ds_read_b32,
//        32 consecutive floats per half-wave: conflict free for any LD).
#pragma once
#include "sgg_common.h"

#define SGG_LDK 36

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row / col of accumulator register r for this lane inside a 32x32 tile
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

// One (8*KB)-deep k-slab of KC x KC tiles with row stride LDK floats:
// acc[TM][TN] += A_s[wm0 + ..][0 .. 8*KB) * B_s[wn0 + ..][0 .. 8*KB)^T
// (LDK = 36 for 32-deep slabs, 68 for 64-deep slabs: 68*i mod 64 = 4*i, also conflict free for ds_read_b128)
template <int TM, int TN, int LDK = SGG_LDK, int KB = 4>
__device__ __forceinline__ void mma_slab_kc_kc(const float* __restrict__ A_s, const float* __restrict__ B_s,
                                               int wm0, int wn0, int lane, f32x16 (&acc)[TM][TN]) {
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    f32x4 a[TM], b[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      a[tm] = *reinterpret_cast<const f32x4*>(A_s + (wm0 + tm * 32 + i) * LDK + kb * 8 + 4 * h);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
      b[tn] = *reinterpret_cast<const f32x4*>(B_s + (wn0 + tn * 32 + i) * LDK + kb * 8 + 4 * h);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(a[tm][jj], b[tn][jj], acc[tm][tn]);
  }
}

// k-slab of MC x MC tiles, k rows [k_begin, k_begin + k_count) of the slab (k_count even):
// acc += A_s[k][wm0 + ..]^T * B_s[k][wn0 + ..]
template <int TM, int TN>
__device__ __forceinline__ void mma_slab_mc_mc(const float* __restrict__ A_s, int lda_s, const float* __restrict__ B_s,
                                               int ldb_s, int wm0, int wn0, int k_begin, int k_count, int lane,
                                               f32x16 (&acc)[TM][TN]) {
  const int i = lane & 31, h = lane >> 5;
  for (int k = k_begin; k < k_begin + k_count; k += 2) {
    float a[TM], b[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = A_s[(k + h) * lda_s + wm0 + tm * 32 + i];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = B_s[(k + h) * ldb_s + wn0 + tn * 32 + i];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(a[tm], b[tn], acc[tm][tn]);
  }
}

// k-slab with A in KC layout and B in MC layout (row-major A[M,K] times row-major B[K,N]).
// MFMA jj of 8-k block kb contracts k = {kb*8+jj, kb*8+4+jj}; B is read at exactly those k rows.
template <int TM, int TN>
__device__ __forceinline__ void mma_slab_kc_mc(const float* __restrict__ A_s, const float* __restrict__ B_s, int ldb_s,
                                               int wm0, int wn0, int lane, f32x16 (&acc)[TM][TN]) {
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    f32x4 a[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      a[tm] = *reinterpret_cast<const f32x4*>(A_s + (wm0 + tm * 32 + i) * SGG_LDK + kb * 8 + 4 * h);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = B_s[(kb * 8 + 4 * h + jj) * ldb_s + wn0 + tn * 32 + i];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(a[tm][jj], b[tn], acc[tm][tn]);
    }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
}
