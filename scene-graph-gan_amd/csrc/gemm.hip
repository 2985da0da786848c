// General f32 GEMM on MFMA for the recurrent heads (gfx950):  C[M,N] (+)= op(A) * op(B) (+ bias)
// Reference call sites: tf.layers.dense (generator_with_attention.py:15,88; discriminator_with_attention.py:15,90),
// the LayerNormBasicLSTMCell kernel matmul (generator_with_attention.py:87), tf.matmul(indices, W)
// (discriminator_with_attention.py:87) and their gradients under optimizer.minimize (train.py:265-266).
//
//   NN  C = A[M,K] * B[K,N]      forward  ("skinny": small M, weight streaming)
//   NT  C = A[M,K] * B[N,K]^T    dgrad
//   TN  C = A[K,M]^T * B[K,N]    wgrad    (K = rows of the batch)
// 64x64x32 tiles, 4 waves (2x2, one 32x32 MFMA tile each), optional split-K with f32 partial slabs summed
// in a fixed order by a second kernel (deterministic). Arbitrary M, N, K, leading dimensions and alignment
// (unaligned operands take a scalar load path).
#include "mma_f32.h"
#include <stdlib.h>
#include <type_traits>

// 1: the head GEMMs whose 64 x 64 tiling would split K over workgroups run on gemm_wk_kernel (split over the waves of one
// workgroup, no slabs, one launch); 0 (-DSGG_GEMM_WK=0): always the slab kernels
#ifndef SGG_GEMM_WK
#define SGG_GEMM_WK 1
#endif

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  float* slabs;
  int M, N, K, lda, ldb, ldc;
  int kchunk, nsplit, accumulate, avec, bvec;
};

// load 4 consecutive elements p[0..3] of a row, elements >= nvalid are zero
__device__ __forceinline__ f32x4 load4(const float* p, int nvalid, bool vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (nvalid >= 4 && vec) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
    if (nvalid > 0) v[0] = p[0];
    if (nvalid > 1) v[1] = p[1];
    if (nvalid > 2) v[2] = p[2];
    if (nvalid > 3) v[3] = p[3];
  }
  return v;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int BM = 64, BN = 64;
  constexpr int A_FLOATS = TA ? 32 * BM : BM * SGG_LDK;
  constexpr int B_FLOATS = TB ? BN * SGG_LDK : 32 * BN;
  __shared__ __attribute__((aligned(16))) float lds[A_FLOATS + B_FLOATS];
  float* A_s = lds;
  float* B_s = lds + A_FLOATS;

  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const int split = blockIdx.z;
  const int k_begin = split * p.kchunk, k_end = min(k_begin + p.kchunk, p.K);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;

  f32x16 acc[1][1];
  acc_zero<1, 1>(acc);
  f32x4 ra[2], rb[2];

  auto issue_loads = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pi = tid + 256 * j;
      if (!TA) {  // A[M,K] row major -> KC tile rows m, 8 float4 per row
        const int row = pi >> 3, c4 = pi & 7;
        const int m = m0 + row, k = k0 + c4 * 4;
        ra[j] = (m < p.M) ? load4(p.A + (size_t)m * p.lda + k, k_end - k, p.avec) : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {  // A[K,M] row major -> MC tile rows k, 16 float4 per row
        const int row = pi >> 4, c4 = pi & 15;
        const int k = k0 + row, m = m0 + c4 * 4;
        ra[j] = (k < k_end) ? load4(p.A + (size_t)k * p.lda + m, p.M - m, p.avec) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (TB) {  // B[N,K] row major -> KC tile rows n
        const int row = pi >> 3, c4 = pi & 7;
        const int n = n0 + row, k = k0 + c4 * 4;
        rb[j] = (n < p.N) ? load4(p.B + (size_t)n * p.ldb + k, k_end - k, p.bvec) : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {  // B[K,N] row major -> MC tile rows k
        const int row = pi >> 4, c4 = pi & 15;
        const int k = k0 + row, n = n0 + c4 * 4;
        rb[j] = (k < k_end) ? load4(p.B + (size_t)k * p.ldb + n, p.N - n, p.bvec) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  if (k_begin < k_end) issue_loads(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += 32) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pi = tid + 256 * j;
      if (!TA) *reinterpret_cast<f32x4*>(A_s + (pi >> 3) * SGG_LDK + (pi & 7) * 4) = ra[j];
      else *reinterpret_cast<f32x4*>(A_s + pi * 4) = ra[j];
      if (TB) *reinterpret_cast<f32x4*>(B_s + (pi >> 3) * SGG_LDK + (pi & 7) * 4) = rb[j];
      else *reinterpret_cast<f32x4*>(B_s + pi * 4) = rb[j];
    }
    __syncthreads();
    if (k0 + 32 < k_end) issue_loads(k0 + 32);
    if (!TA && TB) mma_slab_kc_kc<1, 1>(A_s, B_s, wm0, wn0, lane, acc);
    else if (!TA && !TB) mma_slab_kc_mc<1, 1>(A_s, B_s, BN, wm0, wn0, lane, acc);
    else mma_slab_mc_mc<1, 1>(A_s, BM, B_s, BN, wm0, wn0, 0, 32, lane, acc);
  }

  const int n = n0 + wn0 + acc_col(lane);
  if (n >= p.N) return;
  if (p.nsplit > 1) {
    float* o = p.slabs + (size_t)split * p.M * p.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm0 + acc_row(r, lane);
      if (m < p.M) o[(size_t)m * p.N + n] = acc[0][0][r];
    }
  } else {
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm0 + acc_row(r, lane);
      if (m < p.M) {
        float* c = p.C + (size_t)m * p.ldc + n;
        float v = acc[0][0][r] + bv;
        if (p.accumulate) v += *c;
        *c = v;
      }
    }
  }
}

__global__ void gemm_slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, const float* __restrict__ bias,
                                        int M, int N, int ldc, int nsplit, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float s = bias ? bias[n] : 0.f;
  for (int k = 0; k < nsplit; ++k) s += slabs[(long long)k * M * N + i];
  float* c = C + (size_t)m * ldc + n;
  if (accumulate) s += *c;
  *c = s;
}

// Many slabs (the attention product: 128 slabs of 64 x 196): 32 outputs per workgroup, 8 thread groups each summing every 8th
// slab, combined through LDS in group order (deterministic).  One thread per output walked 128 dependent 50-KB-strided loads.
__global__ __launch_bounds__(256) void gemm_slab_reduce_wide_kernel(const float* __restrict__ slabs, float* __restrict__ C,
                                                                    const float* __restrict__ bias, int M, int N, int ldc, int nsplit,
                                                                    int accumulate) {
  __shared__ float part[8][32];
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long long i = (long long)blockIdx.x * 32 + e, mn = (long long)M * N;
  float s = 0.f;
  if (i < mn)
    for (int k = g; k < nsplit; k += 8) s += slabs[(long long)k * mn + i];
  part[g][e] = s;
  __syncthreads();
  if (g == 0 && i < mn) {
    const int m = (int)(i / N), n = (int)(i % N);
    float t = bias ? bias[n] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][e];
    float* c = C + (size_t)m * ldc + n;
    if (accumulate) t += *c;
    *c = t;
  }
}

// ---- split-K INSIDE one workgroup (NN and NT of the recurrent heads) --------------------------------------------------------
// The head GEMMs have M = 64 .. 192 rows and K = 196 .. 2048: with 64 x 64 tiles they are a handful of tiles, so the kernel above
// splits K over workgroups, writes f32 partial slabs and a second launch sums them - two dependent launches and a slab round trip
// per GEMM, ~100 reduce launches per G+D step, on the critical path of the LSTM step chains.  Here a workgroup OWNS its output tile
// for the whole of K and the split is over its 8 WAVES:
//   * tile = (16 TM) rows x 16 columns on v_mfma_f32_16x16x4_f32 (same FLOPs per cycle as 32x32x2; the 16-column tile is what gives
//     N / 16 x M / (16 TM) >= 200 workgroups for these shapes: TM is chosen per call so that the grid fills the 256 CUs);
//   * wave w takes every k in [16 g_w, 16 g_{w+1}) (a balanced partition of the 16-deep k groups); operands go global -> registers
//     directly, 16 bytes per lane along k where k is the contiguous dimension (A always; B in NT), PD groups in flight per wave;
//     lane l = (q, i) = (l >> 4, l & 15) loads k = kb + 4 q .. + 3 and feeds element jj to MFMA jj, which then contracts
//     k = {kb + jj, kb + 4 + jj, kb + 8 + jj, kb + 12 + jj} - the same permutation for A and B, so the sum is unchanged;
//   * the 8 partial accumulators are summed through LDS in wave order (deterministic), bias / accumulate applied, ONE store.
// Rows / columns past M / N are clamped on load (finite values, never stored); k past K loads zeros.
#define WK_WAVES 8
template <int TM, bool TB, int PD>
__global__ __launch_bounds__(64 * WK_WAVES, 4) void gemm_wk_kernel(GemmParams p) {
  __shared__ float red[WK_WAVES * TM * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int mtiles = (p.M + 16 * TM - 1) / (16 * TM);
  const int ntiles = (p.N + 15) / 16;
  const int bid = xcd_remap(blockIdx.x, mtiles * ntiles);       // consecutive ids share an XCD: the m-tiles of one weight slab
  const int m0 = (bid % mtiles) * 16 * TM, n0 = (bid / mtiles) * 16;
  const int ngroups = (p.K + 15) >> 4;
  const int g0 = (wave * ngroups) / WK_WAVES, g1 = ((wave + 1) * ngroups) / WK_WAVES;

  const float* arow[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) arow[tm] = p.A + (size_t)min(m0 + tm * 16 + i, p.M - 1) * p.lda;
  const int ncol = min(n0 + i, p.N - 1);
  const float* bptr = TB ? p.B + (size_t)ncol * p.ldb : p.B + ncol;

  f32x4 ra[PD][TM];
  f32x4 rb[PD];
  auto load = [&](auto s_c, int g) __attribute__((always_inline)) {
    constexpr int s = decltype(s_c)::value;
    const int k = 16 * g + 4 * q;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) ra[s][tm] = load4(arow[tm] + k, p.K - k, p.avec);
    if constexpr (TB) {
      rb[s] = load4(bptr + k, p.K - k, p.bvec);
    } else {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) rb[s][jj] = (k + jj < p.K) ? bptr[(size_t)(k + jj) * p.ldb] : 0.f;
    }
  };
  f32x4 acc[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) acc[tm] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](auto s_c) __attribute__((always_inline)) {
    constexpr int s = decltype(s_c)::value;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[s][tm][jj], rb[s][jj], acc[tm], 0, 0, 0);
  };
  auto step = [&](auto s_c, int g) __attribute__((always_inline)) {
    if (g < g1) {
      compute(s_c);
      if (g + PD < g1) load(s_c, g + PD);
    }
  };
  if (g0 < g1) load(std::integral_constant<int, 0>{}, g0);
  if (g0 + 1 < g1) load(std::integral_constant<int, 1>{}, g0 + 1);
  if constexpr (PD == 4) {
    if (g0 + 2 < g1) load(std::integral_constant<int, 2>{}, g0 + 2);
    if (g0 + 3 < g1) load(std::integral_constant<int, 3>{}, g0 + 3);
  }
  for (int g = g0; g < g1; g += PD) {
    step(std::integral_constant<int, 0>{}, g);
    step(std::integral_constant<int, 1>{}, g + 1);
    if constexpr (PD == 4) {
      step(std::integral_constant<int, 2>{}, g + 2);
      step(std::integral_constant<int, 3>{}, g + 3);
    }
  }
  // ---- the waves' partial tiles through LDS, summed in wave order ----
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * TM + tm) * 256 + r * 64 + lane] = acc[tm][r];
  __syncthreads();
  for (int o = tid; o < TM * 256; o += 64 * WK_WAVES) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WK_WAVES; ++w) v += red[w * TM * 256 + o];
    const int tm = o >> 8, r = (o >> 6) & 3, l = o & 63;
    const int m = m0 + tm * 16 + (l >> 4) * 4 + r, n = n0 + (l & 15);
    if (m < p.M && n < p.N) {
      float* c = p.C + (size_t)m * p.ldc + n;
      if (p.bias) v += p.bias[n];
      if (p.accumulate) v += *c;
      *c = v;
    }
  }
}

// rows per tile = 16 TM: the largest TM of {6, 4, 2, 1} that still gives >= 224 workgroups (one per CU with some slack), else 1
static int gemm_wk_tm(int M, int N) {
  const int ntn = sgg_cdiv(N, 16);
  const int cand[4] = {6, 4, 2, 1};
  for (int c = 0; c < 4; ++c)
    if (cand[c] * 16 <= ((M + 15) / 16) * 16 && sgg_cdiv(M, 16 * cand[c]) * ntn >= 224) return cand[c];
  return 1;
}
template <bool TB>
static void gemm_wk_launch(const GemmParams& p, hipStream_t st) {
  const int tm = gemm_wk_tm(p.M, p.N);
  const dim3 grid(sgg_cdiv(p.M, 16 * tm) * sgg_cdiv(p.N, 16));
  if (tm == 6) hipLaunchKernelGGL((gemm_wk_kernel<6, TB, 2>), grid, dim3(64 * WK_WAVES), 0, st, p);
  else if (tm == 4) hipLaunchKernelGGL((gemm_wk_kernel<4, TB, 2>), grid, dim3(64 * WK_WAVES), 0, st, p);
  else if (tm == 2) hipLaunchKernelGGL((gemm_wk_kernel<2, TB, 4>), grid, dim3(64 * WK_WAVES), 0, st, p);
  else hipLaunchKernelGGL((gemm_wk_kernel<1, TB, 4>), grid, dim3(64 * WK_WAVES), 0, st, p);
}
// the shapes that take it: NN / NT whose 64 x 64 tiling would split K over workgroups, M and K of the heads' size
static bool gemm_wk_applicable(int mode, int M, int N, int K, int nsplit_old) {
  // measured per shape (profiles/r04_gemm_wk_per_shape.log): 2 - 3 x faster on the small products (3.6 against 8.5 us), 20 % on
  // M = 64 and M = 192 gate products; with 65 .. 128 rows and a large weight matrix its 16-column tiles re-read A once too often
  // (20.0 against 16.2 us): those stay on the slab kernels
  if (M > 64 && M <= 128 && (long long)N * K >= (1LL << 21)) return false;
  return SGG_GEMM_WK && mode != 2 && nsplit_old > 1 && M <= 512 && K <= 8192 && N >= 16;
}

// smallest K range one workgroup of a split-K head GEMM takes.  These GEMMs are chains of dependent 32-deep slabs (f32 MFMA: 0.43 us
// per slab per wave) on a handful of tiles: deeper splits shorten the chain (measured 256 -> 64: 53.16 -> 52.82 ms per G+D step)
#ifndef SGG_GEMM_MIN_KCHUNK
#define SGG_GEMM_MIN_KCHUNK 64
#endif
static void gemm_plan(int M, int N, int K, int* nsplit, int* kchunk) {
  const int tiles = sgg_cdiv(M, 64) * sgg_cdiv(N, 64);
  int ns = 1;
  if (tiles < 256 && K >= 2048) {
    ns = (512 + tiles - 1) / tiles;
    const int maxns = K / 512 > 0 ? K / 512 : 1;
    if (ns > maxns) ns = maxns;
    if (ns < 1) ns = 1;
  } else if (tiles < 128 && K >= 512) {
    // the LSTM gate / decoder GEMMs (M = 64..384 rows, K = 512..1536): 32..96 tiles on 256 CUs, weight-streaming bound
    // (measured on one box, same call: 56.97 -> 55.9 ms per G+D step)
    ns = (256 + tiles - 1) / tiles;
    const int maxns = K / SGG_GEMM_MIN_KCHUNK;
    if (ns > maxns) ns = maxns;
    if (ns < 1) ns = 1;
  } else if (tiles < 64 && K >= 128) {
    ns = K / SGG_GEMM_MIN_KCHUNK;
    if (ns > 64 / tiles) ns = 64 / tiles;
    if (ns < 1) ns = 1;
  }
  int kc = ((K + ns - 1) / ns + 31) / 32 * 32;
  if (kc < 32) kc = 32;
  *kchunk = kc;
  *nsplit = (K + kc - 1) / kc;
  if (*nsplit < 1) *nsplit = 1;
}

extern "C" size_t sgg_gemm_workspace_bytes(int M, int N, int K) {
  int ns, kc;
  gemm_plan(M, N, K, &ns, &kc);
  return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 0;
}

// mode: 0 = NN (fwd), 1 = NT (dgrad), 2 = TN (wgrad)
static int gemm_run(int mode, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    const float* bias, int accumulate, void* workspace, size_t workspace_bytes, void* stream, const char* name) {
  SGG_CHECK_ARG(A && B && C, "%s: null pointer", name);
  SGG_CHECK_ARG(M > 0 && N > 0 && K > 0, "%s: M, N, K must be positive (got %d, %d, %d)", name, M, N, K);
  const int a_cols = (mode == 2) ? M : K, b_cols = (mode == 1) ? K : N;
  SGG_CHECK_ARG(lda >= a_cols && ldb >= b_cols && ldc >= N, "%s: leading dimension too small", name);
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.slabs = (float*)workspace;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.accumulate = accumulate;
  p.avec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
  p.bvec = (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
  gemm_plan(M, N, K, &p.nsplit, &p.kchunk);
  hipStream_t st = (hipStream_t)stream;
  if (gemm_wk_applicable(mode, M, N, K, p.nsplit)) {
    if (mode == 0) gemm_wk_launch<false>(p, st);
    else gemm_wk_launch<true>(p, st);
    SGG_LAUNCH_CHECK(name);
    return SGG_OK;
  }
  if (p.nsplit > 1) {
    const size_t need = (size_t)p.nsplit * M * N * sizeof(float);
    if (!workspace || workspace_bytes < need) {
      sgg_set_error("%s: workspace too small (%zu < %zu)", name, workspace_bytes, need);
      return SGG_ERR_WORKSPACE;
    }
  }
  dim3 grid(sgg_cdiv(M, 64) * sgg_cdiv(N, 64), 1, p.nsplit);
  if (mode == 0) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, st, p);
  else if (mode == 1) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, st, p);
  SGG_LAUNCH_CHECK(name);
  if (p.nsplit > 1) {
    const long long n = (long long)M * N;
    if (p.nsplit >= 16)
      hipLaunchKernelGGL(gemm_slab_reduce_wide_kernel, dim3(sgg_cdiv(n, 32)), dim3(256), 0, st, (const float*)workspace, C, bias, M,
                         N, ldc, p.nsplit, accumulate);
    else
      hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3(sgg_cdiv(n, 256)), dim3(256), 0, st, (const float*)workspace, C, bias, M,
                         N, ldc, p.nsplit, accumulate);
    SGG_LAUNCH_CHECK(name);
  }
  return SGG_OK;
}

extern "C" int sgg_gemm_skinny_fwd(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                   const float* bias, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return gemm_run(0, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, ws, ws_bytes, stream, "sgg_gemm_skinny_fwd");
}
extern "C" int sgg_gemm_skinny_dgrad(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                     int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return gemm_run(1, M, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, ws, ws_bytes, stream, "sgg_gemm_skinny_dgrad");
}
extern "C" int sgg_gemm_skinny_wgrad(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                     int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return gemm_run(2, M, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, ws, ws_bytes, stream, "sgg_gemm_skinny_wgrad");
}

// Step-invariant part of the attention perceptron (generator_with_attention.py:15): P = ctx_flat * W_ctx + b.
// W_ctx = the first L*C rows of attention_perceptron/kernel; hoisted out of the 3-step loop.
extern "C" int sgg_attn_ctx_gemm_fwd(int B, int L, int LC, const float* ctx_flat, const float* w_ctx, const float* bias, float* P,
                                     void* ws, size_t ws_bytes, void* stream) {
  return gemm_run(0, B, L, LC, ctx_flat, LC, w_ctx, L, P, L, bias, 0, ws, ws_bytes, stream, "sgg_attn_ctx_gemm_fwd");
}
extern "C" int sgg_attn_ctx_gemm_dgrad(int B, int L, int LC, const float* dP, const float* w_ctx, float* dctx_flat, int accumulate,
                                       void* ws, size_t ws_bytes, void* stream) {
  return gemm_run(1, B, LC, L, dP, L, w_ctx, L, dctx_flat, LC, nullptr, accumulate, ws, ws_bytes, stream, "sgg_attn_ctx_gemm_dgrad");
}
extern "C" int sgg_attn_ctx_gemm_wgrad(int B, int L, int LC, const float* ctx_flat, const float* dP, float* dw_ctx, int accumulate,
                                       void* ws, size_t ws_bytes, void* stream) {
  return gemm_run(2, LC, L, B, ctx_flat, LC, dP, L, dw_ctx, L, nullptr, accumulate, ws, ws_bytes, stream, "sgg_attn_ctx_gemm_wgrad");
}
