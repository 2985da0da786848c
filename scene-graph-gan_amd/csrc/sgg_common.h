// Common host/device helpers for libsgg_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define SGG_OK 0
#define SGG_ERR_ARG (-1)
#define SGG_ERR_LAUNCH (-2)
#define SGG_ERR_WORKSPACE (-3)

// library-internal (not part of the C ABI of include/sgg_hip.h: hidden, so that the exported sgg_* set equals the declared one)
extern "C" __attribute__((visibility("hidden"))) void sgg_set_error(const char* fmt, ...);

#define SGG_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      sgg_set_error(__VA_ARGS__);                \
      return SGG_ERR_ARG;                        \
    }                                            \
  } while (0)

#define SGG_LAUNCH_CHECK(name)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      sgg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return SGG_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

// LayerNorm partial statistics of a tile of y, as the convolution epilogues emit them and the LayerNorm kernels merge them:
// (count, mean, M2 = sum (y - mean)^2, max |y - mean|) -> SGG_TS floats per tile
#define SGG_TS 4

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int sgg_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// ---- wave (64 lanes) reductions through DPP/ds_swizzle-backed shuffles -------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// The same reductions without the LDS crossbar: four DPP steps inside a row of 16 lanes (quad permutes, half-row and row mirror),
// then the four row sums through v_readlane.  wave_sum / wave_max above cost six ds_bpermute round trips each (~100 cycles apiece,
// dependent): nothing behind a long K loop, but a third of a tile's time in the conv1_1 forward (28 MFMAs per tile).  Another
// summation order than wave_sum (results differ in the last bits); every lane returns the same value.
#define SGG_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += SGG_DPP(v, 0xB1);       // quad_perm [1,0,3,2]: lane ^ 1
  v += SGG_DPP(v, 0x4E);       // quad_perm [2,3,0,1]: lane ^ 2
  v += SGG_DPP(v, 0x141);      // row_half_mirror: the other quad of the 8
  v += SGG_DPP(v, 0x140);      // row_mirror: the other half of the 16
  const int b = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, SGG_DPP(v, 0xB1));
  v = fmaxf(v, SGG_DPP(v, 0x4E));
  v = fmaxf(v, SGG_DPP(v, 0x141));
  v = fmaxf(v, SGG_DPP(v, 0x140));
  const int b = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48))));
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS. All threads get the sum.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- per-tensor power-of-two scaling for the f16x3 convolution mode --------------------------------------
// amax words hold max|x| of a tensor as float bits (non-negative floats order like unsigned ints -> atomicMax).
__device__ __forceinline__ void atomic_amax(float* p, float v) {
  // thousands of workgroups publish into one word (a single address sustains ~88 atomics/us): skip the atomic when the
  // word already holds a larger value, which is the case for all but the first few arrivals
  const unsigned bits = __float_as_uint(v);
  if (bits > __hip_atomic_load(reinterpret_cast<unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(reinterpret_cast<unsigned*>(p), bits);
}
// block-wide (256 threads) max published once per workgroup; `red` is >= 4 floats of LDS
__device__ __forceinline__ void block_publish_amax(float* p, float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomic_amax(p, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}
// exponent e with |x| * 2^e <= 2^14 for every |x| <= amax (fp16 max is 65504); 0 for an all-zero tensor
__device__ __forceinline__ int scale_exp_from_amax(float amax) {
  if (!(amax > 0.f)) return 0;
  int k;
  frexpf(amax, &k);           // amax = m * 2^k, m in [0.5, 1)
  const int e = 14 - k;
  return e > 100 ? 100 : (e < -100 ? -100 : e);
}

// f16x3 mode: two floats (already scaled into fp16 range) -> packed hi and lo fp16 pieces, a = hi + lo up to 2^-23 relative.
// Both conversions round to nearest even (v_cvt_pk_f16_f32): the split error is zero-mean, unlike a truncating split.
typedef _Float16 sgg_h2 __attribute__((ext_vector_type(2)));
typedef float sgg_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void f16_split2(float a, float b, unsigned& hi, unsigned& lo) {
  const sgg_h2 h = __builtin_convertvector(sgg_f2{a, b}, sgg_h2);
  hi = __builtin_bit_cast(unsigned, h);
#ifdef SGG_EXPERIMENT_NOSPLIT      // timing experiment only (scripts/build_prof_lib.sh nosplit): what the staging costs without the residual piece
  lo = hi;
  return;
#endif
  const sgg_h2 l = __builtin_convertvector(sgg_f2{a - (float)h[0], b - (float)h[1]}, sgg_h2);
  lo = __builtin_bit_cast(unsigned, l);
}

// precision code of the C ABI -> (HALF, pieces) of the resident split kernels: 2 = fp16 x 2 pieces (3 products), 3 = bf16 x 2,
// 1 = ONE fp16 piece (one product: mixed-precision arithmetic, tolerance 2e-3), 4 = one bf16 piece (1.5e-2)
static inline bool sgg_prec_resident(int precision) { return precision >= 1 && precision <= 4; }
static inline bool sgg_prec_half(int precision) { return precision == 1 || precision == 2; }
static inline bool sgg_prec_one(int precision) { return precision == 1 || precision == 4; }
// kernels without a single-piece variant run the two-piece mode of the same number format
static inline int sgg_prec_general(int precision) { return precision == 1 ? 2 : (precision == 4 ? 3 : precision); }

// XCD-aware bijective block remap (cdna_hip_programming.md T1): consecutive logical ids share an XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}
#endif

// CUs per XCD that the PERSISTENT forward / dgrad convolution kernels occupy (one 8-wave workgroup, or two 4-wave ones, per CU; an XCD
// has 32).  Fewer than 32 leaves whole CUs - registers, wave slots - to the HBM-bound kernels of the other HIP streams (LayerNorm
// passes of the other network, Adam, the heads), which cannot co-reside with a workgroup that holds a CU's whole register file.
// Experiment of round 5 (profiles/r05_persistent_cu_cap_ab.log); 32 = every CU.
#ifndef SGG_PERSIST_CUS_PER_XCD
#define SGG_PERSIST_CUS_PER_XCD 32
#endif

// The resident convolution kernels raise the wave's issue priority over their MFMA clusters (s_setprio), so that the other resident
// workgroup's wave in its staging phase does not take issue slots from the wave feeding the matrix pipe (-0.25 ms per step, same box,
// two repetitions; raising it before the tap's B-fragment loads instead: the same).  -DSGG_MFMA_PRIO=0 builds without.
#ifndef SGG_MFMA_PRIO
#define SGG_MFMA_PRIO 1
#endif
#if SGG_MFMA_PRIO
#define SGG_PRIO_HI() __builtin_amdgcn_s_setprio(SGG_MFMA_PRIO)
#define SGG_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define SGG_PRIO_HI()
#define SGG_PRIO_LO()
#endif

// The convolution epilogues store their output tiles with the nontemporal hint (the tensors are far larger than the L2 of an XCD and
// are next read by another kernel): -0.2 ms per step, four same-box repetitions.  -DSGG_CONV_NT_STORE=0 builds without.
#ifndef SGG_CONV_NT_STORE
#define SGG_CONV_NT_STORE 1
#endif
__device__ __forceinline__ void sgg_out_store(float* p, float v) {
#if SGG_CONV_NT_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void sgg_out_store4(float* p, const f32x4& v) {     // p 16-byte aligned
#if SGG_CONV_NT_STORE
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}

// 4x4 transpose inside every quad of lanes (lanes 4g .. 4g+3), in registers (DPP quad permutes, no LDS):
//     before: register j of lane 4g+k holds M[j][k]        after: register j of lane 4g+k holds M[k][j]
// A 32x32 MFMA accumulator has its COLUMN on the lane and four consecutive ROWS in registers 4q .. 4q+3, so a row-major output
// (row = pixel, column = channel) can only be stored 4 bytes per lane as it stands: 64 `global_store_dword` per 64x64 tile and
// wave, and the epilogue is bound by the store ISSUE rate of the CU (~8 B/clk), not by bandwidth (cdna_hip_programming.md T21).
// Transposed, lane 4g+k holds row k, columns 4g .. 4g+3 in four registers: ONE 16-byte store where there were four 4-byte ones.
__device__ __forceinline__ float sgg_quad_perm_1032(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // lane ^ 1
}
__device__ __forceinline__ float sgg_quad_perm_2301(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // lane ^ 2
}
__device__ __forceinline__ void sgg_quad_transpose4(float& a0, float& a1, float& a2, float& a3, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  // (every permute is evaluated by ALL lanes before anything is selected: a DPP read of a lane that sits in the untaken arm of a
  //  conditional expression returns zero)
  // stage 1: element (j, k) with (j ^ k) & 1 comes from (j ^ 1, k ^ 1)
  const float x0 = sgg_quad_perm_1032(a0), x1 = sgg_quad_perm_1032(a1), x2 = sgg_quad_perm_1032(a2), x3 = sgg_quad_perm_1032(a3);
  const float t0 = b0 ? x1 : a0;
  const float t1 = b0 ? a1 : x0;
  const float t2 = b0 ? x3 : a2;
  const float t3 = b0 ? a3 : x2;
  // stage 2: element (j, k) with (j ^ k) & 2 comes from (j ^ 2, k ^ 2)
  const float y0 = sgg_quad_perm_2301(t0), y1 = sgg_quad_perm_2301(t1), y2 = sgg_quad_perm_2301(t2), y3 = sgg_quad_perm_2301(t3);
  a0 = b1 ? y2 : t0;
  a1 = b1 ? y3 : t1;
  a2 = b1 ? t2 : y0;
  a3 = b1 ? t3 : y1;
}
