// Recurrent-head kernels of generator_with_attention / discriminator_with_attention (gfx950), all templated
// over float / Dual (dual.h) so the gradient-penalty second-order path reuses the first-order code.
//
//  attn_step_{fwd,bwd}   attentionMechanism: e = P[b] + c*W_c, alpha = softmax_L(e), z = sum_l alpha_l ctx[b,l,:]
//                        reference: architectures/generator_with_attention.py:13-18 (discriminator :13-18)
//  lnlstm_gates_{fwd,bwd} tf.contrib.rnn.LayerNormBasicLSTMCell(512) pointwise part (gate order i,j,f,o, LN per gate,
//                        forget_bias 1, LN on the new cell state), generator_with_attention.py:79,87; Appendix A.5.
//                        One wave per batch row; the five 512-wide LayerNorms are wave reductions (no LDS, no MFMA).
//  spatial_mean_{fwd,bwd} initial state c0 = h0 = mean_{h,w} downsampled, generator_with_attention.py:76-77
//  colsum                 bias / LN-parameter gradient row sums
//
// Dual tensors are two planes (real, dual). For cotangent tensors: real plane = cotangent of the tangent,
// dual plane = cotangent of the primal (dual.h).
#include "dual.h"
#include <stdlib.h>

#define HEAD_LN_EPS 1e-12f
#define NUM_UNITS 512

__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ Dual block_sum_256(Dual v, float* red) {
  const float r = block_sum_256(v.r, red);
  const float d = block_sum_256(v.d, red);
  return mk(r, d);
}

// ---------------------------------------------------------------------------------------------------
// attention step forward: one workgroup per row r (image b = r % B)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_step_fwd_kernel(const float* __restrict__ P, const float* __restrict__ ec_r,
                                                            const float* __restrict__ ec_d, int ldec,
                                                            const float* __restrict__ ctx, float* __restrict__ al_r,
                                                            float* __restrict__ al_d, float* __restrict__ z_r,
                                                            float* __restrict__ z_d, int ldz, int B, int L, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[4];
  T* al = reinterpret_cast<T*>(sm);
  const int r = blockIdx.x, b = r % B, tid = threadIdx.x;
  float mx = -3.0e38f;
  for (int l = tid; l < L; l += 256) {
    const T e = Sc<T>::lift(P[(size_t)b * L + l]) + Sc<T>::ld(ec_r, ec_d, (size_t)r * ldec + l);
    al[l] = e;
    mx = fmaxf(mx, Sc<T>::re(e));
  }
  mx = block_max_256(mx, red);
  T s = Sc<T>::zero();
  for (int l = tid; l < L; l += 256) {
    const T ex = exp_(al[l] - mx);
    al[l] = ex;
    s += ex;
  }
  s = block_sum_256(s, red);
  const T inv = recip_(s);
  for (int l = tid; l < L; l += 256) {
    const T a = al[l] * inv;
    al[l] = a;
    if (blockIdx.y == 0) Sc<T>::st(al_r, al_d, (size_t)r * L + l, a);
  }
  __syncthreads();
  // z = sum_l alpha_l ctx_l: the four waves split the locations, the lanes the channels (4 each, 256 per pass); workgroup
  // (r, y) of the gridDim.y workgroups of a row takes the y-th part of the channels.  (One thread per channel pair walked all L
  // locations serially on R = 64 workgroups: 20 us for 26 MB.)
  const float* cb = ctx + (size_t)b * L * C;
  T* part = al + L;                                   // [4 waves][256 channels]
  const int lane = tid & 63, wave = tid >> 6;
  const int cper = C / gridDim.y, cbeg = blockIdx.y * cper;
  for (int c0 = cbeg; c0 < cbeg + cper; c0 += 256) {
    const int c = c0 + lane * 4;
    T z[4] = {Sc<T>::zero(), Sc<T>::zero(), Sc<T>::zero(), Sc<T>::zero()};
#pragma unroll 4
    for (int l = wave; l < L; l += 4) {
      const f32x4 cv = *reinterpret_cast<const f32x4*>(cb + (size_t)l * C + c);
      const T a = al[l];
      z[0] += a * cv[0]; z[1] += a * cv[1]; z[2] += a * cv[2]; z[3] += a * cv[3];
    }
    __syncthreads();                                  // (the previous pass's partials have been consumed)
#pragma unroll
    for (int q = 0; q < 4; ++q) part[wave * 256 + lane * 4 + q] = z[q];
    __syncthreads();
    const T zz = (part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]);
    Sc<T>::st(z_r, z_d, (size_t)r * ldz + c0 + tid, zz);
  }
}

// ---------------------------------------------------------------------------------------------------
// attention step backward: one workgroup per image b, loops over the rows r = b + k*B that share it.
//   dalpha_l = <dz, ctx_l>;  de = alpha * (dalpha - <dalpha, alpha>)
//   dctx[b,l,:] (+)= pcot(alpha_l * dz)      dP[b,l] (+)= sum_rows pcot(de_l)
// ---------------------------------------------------------------------------------------------------
#define ATTN_MAX_PASS 4
template <typename T>
__global__ __launch_bounds__(256) void attn_step_bwd_kernel(const float* __restrict__ ctx, const float* __restrict__ al_r,
                                                            const float* __restrict__ al_d, const float* __restrict__ dz_r,
                                                            const float* __restrict__ dz_d, int lddz, float* __restrict__ de_r,
                                                            float* __restrict__ de_d, float* __restrict__ dP,
                                                            float* __restrict__ dctx, int R, int B, int L, int C,
                                                            int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[4];
  const int npass = R / B;
  T* dz_s = reinterpret_cast<T*>(sm);         // [npass][C]
  T* al_s = dz_s + (size_t)npass * C;         // [npass][L]
  T* da_s = al_s + (size_t)npass * L;         // [npass][L]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = 0; k < npass; ++k) {
    const int r = b + k * B;
    for (int c = tid; c < C; c += 256) dz_s[k * C + c] = Sc<T>::ld(dz_r, dz_d, (size_t)r * lddz + c);
    for (int l = tid; l < L; l += 256) al_s[k * L + l] = Sc<T>::ld(al_r, al_d, (size_t)r * L + l);
  }
  __syncthreads();
  const float* cb = ctx + (size_t)b * L * C;
  float* db = dctx + (size_t)b * L * C;
  for (int l = wave; l < L; l += 4) {
    T dot[ATTN_MAX_PASS];
#pragma unroll
    for (int k = 0; k < ATTN_MAX_PASS; ++k) dot[k] = Sc<T>::zero();
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 cv = *reinterpret_cast<const f32x4*>(cb + (size_t)l * C + c);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (accumulate) acc = *reinterpret_cast<const f32x4*>(db + (size_t)l * C + c);
#pragma unroll
      for (int k = 0; k < ATTN_MAX_PASS; ++k) {
        if (k < npass) {
          const T a = al_s[k * L + l];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const T dzv = dz_s[k * C + c + q];
            dot[k] += dzv * cv[q];
            acc[q] += Sc<T>::pcot(a * dzv);
          }
        }
      }
      *reinterpret_cast<f32x4*>(db + (size_t)l * C + c) = acc;
    }
#pragma unroll
    for (int k = 0; k < ATTN_MAX_PASS; ++k) {
      if (k < npass) {
        const T d = wave_sum(dot[k]);
        if (lane == 0) da_s[k * L + l] = d;
      }
    }
  }
  __syncthreads();
  for (int k = 0; k < npass; ++k) {
    T s = Sc<T>::zero();
    for (int l = tid; l < L; l += 256) s += da_s[k * L + l] * al_s[k * L + l];
    s = block_sum_256(s, red);
    const int r = b + k * B;
    for (int l = tid; l < L; l += 256) {
      const T de = al_s[k * L + l] * (da_s[k * L + l] - s);
      Sc<T>::st(de_r, de_d, (size_t)r * L + l, de);
      da_s[k * L + l] = de;   // reuse for the dP sum below (same thread reads it back)
    }
  }
  for (int l = tid; l < L; l += 256) {
    float s = accumulate ? dP[(size_t)b * L + l] : 0.f;
    for (int k = 0; k < npass; ++k) s += Sc<T>::pcot(da_s[k * L + l]);
    dP[(size_t)b * L + l] = s;
  }
}

// Split form of the same backward for full-size feature maps: with one workgroup per image only B (= 64) workgroups
// stream ctx / dctx (1.2 TB/s).  Phase A, grid (B, LS): workgroup (b, j) owns the locations l = j, j + LS, ... of image b:
// dctx rows and the raw dalpha_l = <dz, ctx_l> (parked in `de`).  Phase B, grid B: softmax backward over the L
// locations of each row and dP.  No atomics; same summation order as the fused kernel.
template <typename T>
__global__ __launch_bounds__(256) void attn_step_bwd_ctx_kernel(const float* __restrict__ ctx, const float* __restrict__ al_r,
                                                                const float* __restrict__ al_d, const float* __restrict__ dz_r,
                                                                const float* __restrict__ dz_d, int lddz, float* __restrict__ de_r,
                                                                float* __restrict__ de_d, float* __restrict__ dctx, int R, int B,
                                                                int L, int C, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int npass = R / B;
  T* dz_s = reinterpret_cast<T*>(sm);         // [npass][C]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = 0; k < npass; ++k) {
    const int r = b + k * B;
    for (int c = tid; c < C; c += 256) dz_s[k * C + c] = Sc<T>::ld(dz_r, dz_d, (size_t)r * lddz + c);
  }
  __syncthreads();
  const float* cb = ctx + (size_t)b * L * C;
  float* db = dctx + (size_t)b * L * C;
  for (int l = blockIdx.y * 4 + wave; l < L; l += 4 * gridDim.y) {
    T dot[ATTN_MAX_PASS], a[ATTN_MAX_PASS];
#pragma unroll
    for (int k = 0; k < ATTN_MAX_PASS; ++k) {
      dot[k] = Sc<T>::zero();
      a[k] = k < npass ? Sc<T>::ld(al_r, al_d, (size_t)(b + k * B) * L + l) : Sc<T>::zero();
    }
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 cv = *reinterpret_cast<const f32x4*>(cb + (size_t)l * C + c);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (accumulate) acc = *reinterpret_cast<const f32x4*>(db + (size_t)l * C + c);
#pragma unroll
      for (int k = 0; k < ATTN_MAX_PASS; ++k) {
        if (k < npass) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const T dzv = dz_s[k * C + c + q];
            dot[k] += dzv * cv[q];
            acc[q] += Sc<T>::pcot(a[k] * dzv);
          }
        }
      }
      *reinterpret_cast<f32x4*>(db + (size_t)l * C + c) = acc;
    }
#pragma unroll
    for (int k = 0; k < ATTN_MAX_PASS; ++k) {
      if (k < npass) {
        const T d = wave_sum(dot[k]);
        if (lane == 0) Sc<T>::st(de_r, de_d, (size_t)(b + k * B) * L + l, d);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_step_bwd_softmax_kernel(const float* __restrict__ al_r, const float* __restrict__ al_d,
                                                                    float* __restrict__ de_r, float* __restrict__ de_d,
                                                                    float* __restrict__ dP, int R, int B, int L, int accumulate) {
  __shared__ float red[4];
  const int npass = R / B;
  const int b = blockIdx.x, tid = threadIdx.x;
  float dp[8];                                // locations tid, tid + 256, ... (L <= 2048)
#pragma unroll
  for (int j = 0; j < 8; ++j) dp[j] = 0.f;
  for (int k = 0; k < npass; ++k) {
    const int r = b + k * B;
    T s = Sc<T>::zero();
    for (int l = tid; l < L; l += 256) s += Sc<T>::ld(de_r, de_d, (size_t)r * L + l) * Sc<T>::ld(al_r, al_d, (size_t)r * L + l);
    s = block_sum_256(s, red);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int l = tid + 256 * j;
      if (l < L) {
        const T de = Sc<T>::ld(al_r, al_d, (size_t)r * L + l) * (Sc<T>::ld(de_r, de_d, (size_t)r * L + l) - s);
        Sc<T>::st(de_r, de_d, (size_t)r * L + l, de);
        dp[j] += Sc<T>::pcot(de);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int l = tid + 256 * j;
    if (l < L) dP[(size_t)b * L + l] = (accumulate ? dP[(size_t)b * L + l] : 0.f) + dp[j];
  }
}

// ---------------------------------------------------------------------------------------------------
// LN-LSTM gates: one wave per row, lane owns elements (lane*4 + q) + 256*k of each 512-vector
// ---------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void ld8(const float* pr, const float* pd, size_t base, int lane, T (&x)[8]) {
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) x[k * 4 + q] = Sc<T>::ld(pr, pd, base + lane * 4 + 256 * k + q);
}
template <typename T>
__device__ __forceinline__ void st8(float* pr, float* pd, size_t base, int lane, const T (&x)[8]) {
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) Sc<T>::st(pr, pd, base + lane * 4 + 256 * k + q, x[k * 4 + q]);
}
__device__ __forceinline__ void ldp8(const float* p, int lane, float (&x)[8]) {
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) x[k * 4 + q] = p[lane * 4 + 256 * k + q];
}

// xhat = (x - mean) * rstd over the 512 elements held by the wave
template <typename T>
__device__ __forceinline__ void ln_fwd8(const T (&x)[8], T (&xhat)[8], T& rstd) {
  T s = Sc<T>::zero();
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  const T mean = wave_sum(s) * (1.f / NUM_UNITS);
  T q = Sc<T>::zero();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const T d = x[i] - mean;
    q += d * d;
  }
  const T var = wave_sum(q) * (1.f / NUM_UNITS);
  rstd = rsqrt_(var + HEAD_LN_EPS);
#pragma unroll
  for (int i = 0; i < 8; ++i) xhat[i] = (x[i] - mean) * rstd;
}
// given dn (cotangent of n = xhat*gamma + beta): dx, and the per-row gamma/beta gradient contributions
template <typename T>
__device__ __forceinline__ void ln_bwd8(const T (&dn)[8], const T (&xhat)[8], T rstd, const float (&gamma)[8], T (&dx)[8],
                                        float (&dgam)[8], float (&dbet)[8]) {
  T s1 = Sc<T>::zero(), s2 = Sc<T>::zero();
  T dxh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    dgam[i] = Sc<T>::pcot(dn[i] * xhat[i]);
    dbet[i] = Sc<T>::pcot(dn[i]);
    dxh[i] = dn[i] * gamma[i];
    s1 += dxh[i];
    s2 += dxh[i] * xhat[i];
  }
  const T m1 = wave_sum(s1) * (1.f / NUM_UNITS), m2 = wave_sum(s2) * (1.f / NUM_UNITS);
#pragma unroll
  for (int i = 0; i < 8; ++i) dx[i] = rstd * (dxh[i] - m1 - xhat[i] * m2);
}

template <typename T>
struct GateFwd {
  T xh_i[8], xh_j[8], xh_f[8], xh_o[8], xh_s[8];
  T r_i, r_j, r_f, r_o, r_s;
  T si[8], tj[8], sf[8], so[8], cn[8], th[8];
};

// ln: [10][512] = gamma_i, beta_i, gamma_j, beta_j, gamma_f, beta_f, gamma_o, beta_o, gamma_s, beta_s
template <typename T>
__device__ __forceinline__ void gates_forward(const float* g_r, const float* g_d, const float* c_r, const float* c_d,
                                              const float* __restrict__ ln, int row, int lane, T (&cprev)[8], GateFwd<T>& F) {
  T gi[8], gj[8], gf[8], go[8];
  const size_t gb = (size_t)row * 4 * NUM_UNITS;
  ld8<T>(g_r, g_d, gb, lane, gi);
  ld8<T>(g_r, g_d, gb + NUM_UNITS, lane, gj);
  ld8<T>(g_r, g_d, gb + 2 * NUM_UNITS, lane, gf);
  ld8<T>(g_r, g_d, gb + 3 * NUM_UNITS, lane, go);
  ld8<T>(c_r, c_d, (size_t)row * NUM_UNITS, lane, cprev);
  ln_fwd8<T>(gi, F.xh_i, F.r_i);
  ln_fwd8<T>(gj, F.xh_j, F.r_j);
  ln_fwd8<T>(gf, F.xh_f, F.r_f);
  ln_fwd8<T>(go, F.xh_o, F.r_o);
  float gm[8], bt[8];
  T cp[8];
  ldp8(ln + 0 * NUM_UNITS, lane, gm); ldp8(ln + 1 * NUM_UNITS, lane, bt);
#pragma unroll
  for (int i = 0; i < 8; ++i) F.si[i] = sigmoid_(F.xh_i[i] * gm[i] + bt[i]);
  ldp8(ln + 2 * NUM_UNITS, lane, gm); ldp8(ln + 3 * NUM_UNITS, lane, bt);
#pragma unroll
  for (int i = 0; i < 8; ++i) F.tj[i] = tanh_(F.xh_j[i] * gm[i] + bt[i]);
  ldp8(ln + 4 * NUM_UNITS, lane, gm); ldp8(ln + 5 * NUM_UNITS, lane, bt);
#pragma unroll
  for (int i = 0; i < 8; ++i) F.sf[i] = sigmoid_(F.xh_f[i] * gm[i] + bt[i] + 1.0f);
  ldp8(ln + 6 * NUM_UNITS, lane, gm); ldp8(ln + 7 * NUM_UNITS, lane, bt);
#pragma unroll
  for (int i = 0; i < 8; ++i) F.so[i] = sigmoid_(F.xh_o[i] * gm[i] + bt[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) cp[i] = cprev[i] * F.sf[i] + F.si[i] * F.tj[i];
  ln_fwd8<T>(cp, F.xh_s, F.r_s);
  ldp8(ln + 8 * NUM_UNITS, lane, gm); ldp8(ln + 9 * NUM_UNITS, lane, bt);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    F.cn[i] = F.xh_s[i] * gm[i] + bt[i];
    F.th[i] = tanh_(F.cn[i]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void lnlstm_gates_fwd_kernel(const float* __restrict__ g_r, const float* __restrict__ g_d,
                                                               const float* __restrict__ c_r, const float* __restrict__ c_d,
                                                               const float* __restrict__ ln, float* __restrict__ cn_r,
                                                               float* __restrict__ cn_d, float* __restrict__ h_r,
                                                               float* __restrict__ h_d, int ldh, int R) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  T cprev[8];
  GateFwd<T> F;
  gates_forward<T>(g_r, g_d, c_r, c_d, ln, row, lane, cprev, F);
  T h[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = F.th[i] * F.so[i];
  st8<T>(cn_r, cn_d, (size_t)row * NUM_UNITS, lane, F.cn);
  st8<T>(h_r, h_d, (size_t)row * ldh, lane, h);
}

template <typename T>
__global__ __launch_bounds__(256) void lnlstm_gates_bwd_kernel(const float* __restrict__ g_r, const float* __restrict__ g_d,
                                                               const float* __restrict__ c_r, const float* __restrict__ c_d,
                                                               const float* __restrict__ ln, const float* __restrict__ dh_r,
                                                               const float* __restrict__ dh_d, int lddh,
                                                               const float* __restrict__ dcn_r, const float* __restrict__ dcn_d,
                                                               float* __restrict__ dg_r, float* __restrict__ dg_d,
                                                               float* __restrict__ dcp_r, float* __restrict__ dcp_d,
                                                               float* __restrict__ pgrad, int R) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  T cprev[8];
  GateFwd<T> F;
  gates_forward<T>(g_r, g_d, c_r, c_d, ln, row, lane, cprev, F);
  T dh[8], dcn[8];
  ld8<T>(dh_r, dh_d, (size_t)row * lddh, lane, dh);
  if (dcn_r) ld8<T>(dcn_r, dcn_d, (size_t)row * NUM_UNITS, lane, dcn);
  else {
#pragma unroll
    for (int i = 0; i < 8; ++i) dcn[i] = Sc<T>::zero();
  }
  float gm[8], dgam[8], dbet[8];
  float* pg = pgrad + (size_t)row * 10 * NUM_UNITS;
  auto st_pg = [&](int slot, const float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) pg[slot * NUM_UNITS + lane * 4 + 256 * k + q] = v[k * 4 + q];
  };
  // h = th * so ; th = tanh(cn)
  T d_on[8], d_cn[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const T d_so = dh[i] * F.th[i];
    const T d_th = dh[i] * F.so[i];
    d_on[i] = d_so * F.so[i] * (1.f - F.so[i]);
    d_cn[i] = dcn[i] + d_th * (1.f - F.th[i] * F.th[i]);
  }
  // state LN
  T d_cp[8];
  ldp8(ln + 8 * NUM_UNITS, lane, gm);
  ln_bwd8<T>(d_cn, F.xh_s, F.r_s, gm, d_cp, dgam, dbet);
  st_pg(8, dgam); st_pg(9, dbet);
  // cp = cprev*sf + si*tj
  T d_in[8], d_jn[8], d_fn[8], d_cprev[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    d_cprev[i] = d_cp[i] * F.sf[i];
    const T d_sf = d_cp[i] * cprev[i];
    const T d_si = d_cp[i] * F.tj[i];
    const T d_tj = d_cp[i] * F.si[i];
    d_fn[i] = d_sf * F.sf[i] * (1.f - F.sf[i]);
    d_in[i] = d_si * F.si[i] * (1.f - F.si[i]);
    d_jn[i] = d_tj * (1.f - F.tj[i] * F.tj[i]);
  }
  st8<T>(dcp_r, dcp_d, (size_t)row * NUM_UNITS, lane, d_cprev);
  const size_t gb = (size_t)row * 4 * NUM_UNITS;
  T dx[8];
  ldp8(ln + 0 * NUM_UNITS, lane, gm);
  ln_bwd8<T>(d_in, F.xh_i, F.r_i, gm, dx, dgam, dbet);
  st_pg(0, dgam); st_pg(1, dbet);
  st8<T>(dg_r, dg_d, gb, lane, dx);
  ldp8(ln + 2 * NUM_UNITS, lane, gm);
  ln_bwd8<T>(d_jn, F.xh_j, F.r_j, gm, dx, dgam, dbet);
  st_pg(2, dgam); st_pg(3, dbet);
  st8<T>(dg_r, dg_d, gb + NUM_UNITS, lane, dx);
  ldp8(ln + 4 * NUM_UNITS, lane, gm);
  ln_bwd8<T>(d_fn, F.xh_f, F.r_f, gm, dx, dgam, dbet);
  st_pg(4, dgam); st_pg(5, dbet);
  st8<T>(dg_r, dg_d, gb + 2 * NUM_UNITS, lane, dx);
  ldp8(ln + 6 * NUM_UNITS, lane, gm);
  ln_bwd8<T>(d_on, F.xh_o, F.r_o, gm, dx, dgam, dbet);
  st_pg(6, dgam); st_pg(7, dbet);
  st8<T>(dg_r, dg_d, gb + 3 * NUM_UNITS, lane, dx);
}

// ---------------------------------------------------------------------------------------------------
// spatial mean + column sums
// ---------------------------------------------------------------------------------------------------
// out_c[r, :] = out_h[r, :] = mean_l ctx[r % B, l, :]   for r in [0, R)
// grid (B, C/128): 256 threads = 32 lanes x float4 (128 channels) x 8 row groups over L
__global__ __launch_bounds__(256) void spatial_mean_fwd_kernel(const float* __restrict__ ctx, float* __restrict__ out_c, int ldc,
                                                               float* __restrict__ out_h, int ldh, int R, int B, int L, int C) {
  __shared__ f32x4 part[8][32];
  const int b = blockIdx.x, c = blockIdx.y * 128 + (threadIdx.x & 31) * 4, rg = threadIdx.x >> 5;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int l = rg; l < L; l += 8) s += *reinterpret_cast<const f32x4*>(ctx + ((size_t)b * L + l) * C + c);
  part[rg][threadIdx.x & 31] = s;
  __syncthreads();
  if (rg == 0 && c < C) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += part[k][threadIdx.x & 31];
    s *= 1.f / (float)L;
    for (int r = b; r < R; r += B) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        out_c[(size_t)r * ldc + c + q] = s[q];
        out_h[(size_t)r * ldh + c + q] = s[q];
      }
    }
  }
}
// dctx[b,l,c] (+)= (1/L) * sum_{r = b mod B} (dc0[r,c] + dh0[r,c])
__global__ __launch_bounds__(256) void spatial_mean_bwd_kernel(const float* __restrict__ dc0, int ldc, const float* __restrict__ dh0,
                                                               int ldh, float* __restrict__ dctx, int R, int B, int L, int C,
                                                               int accumulate) {
  const int b = blockIdx.x;
  const float invL = 1.f / (float)L;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int r = b; r < R; r += B) s += dc0[(size_t)r * ldc + c] + dh0[(size_t)r * ldh + c];
    s *= invL;
    for (int l = blockIdx.y; l < L; l += gridDim.y) {
      float* p = dctx + ((size_t)b * L + l) * C + c;
      *p = accumulate ? *p + s : s;
    }
  }
}
// rows [r0, r1) of X summed per column: grid (ceil(cols/256), nchunks); chunk c covers rows [c*rpc, (c+1)*rpc)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int rows, int cols, int ld, float* __restrict__ out,
                                                     int out_ld, int rpc, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rpc, r1 = min(r0 + rpc, rows);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += X[(size_t)r * ld + c];
    s1 += X[(size_t)(r + 1) * ld + c];
    s2 += X[(size_t)(r + 2) * ld + c];
    s3 += X[(size_t)(r + 3) * ld + c];
  }
  for (; r < r1; ++r) s0 += X[(size_t)r * ld + c];
  const float s = (s0 + s1) + (s2 + s3);
  float* o = out + (size_t)blockIdx.y * out_ld + c;
  *o = accumulate ? *o + s : s;
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
static int attn_check(const char* name, int R, int B, int L, int C) {
  SGG_CHECK_ARG(R > 0 && B > 0 && L > 0 && C > 0 && R % B == 0, "%s: need R %% B == 0 (R=%d, B=%d)", name, R, B);
  SGG_CHECK_ARG(C % 256 == 0, "%s: C must be a multiple of 256 (got %d)", name, C);
  return SGG_OK;
}

extern "C" int sgg_attn_step_fwd(const float* P, const float* ec, const float* ec_dual, int ldec, const float* ctx, float* alpha,
                                 float* alpha_dual, float* z, float* z_dual, int ldz, int R, int B, int L, int C, void* stream) {
  SGG_CHECK_ARG(P && ec && ctx && alpha && z, "sgg_attn_step_fwd: null pointer");
  int rc = attn_check("sgg_attn_step_fwd", R, B, L, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int cs = (R < 256 && C % 512 == 0) ? 2 : 1;      // two workgroups per row (256 channels each) while the rows do not fill the chip
  if (ec_dual) {
    SGG_CHECK_ARG(alpha_dual && z_dual, "sgg_attn_step_fwd: dual outputs missing");
    hipLaunchKernelGGL(attn_step_fwd_kernel<Dual>, dim3(R, cs), dim3(256), (size_t)(L + 1024) * sizeof(Dual), st, P, ec, ec_dual, ldec,
                       ctx, alpha, alpha_dual, z, z_dual, ldz, B, L, C);
  } else {
    hipLaunchKernelGGL(attn_step_fwd_kernel<float>, dim3(R, cs), dim3(256), (size_t)(L + 1024) * sizeof(float), st, P, ec, nullptr, ldec,
                       ctx, alpha, nullptr, z, nullptr, ldz, B, L, C);
  }
  SGG_LAUNCH_CHECK("sgg_attn_step_fwd");
  return SGG_OK;
}

// Float mode (alpha_dual == NULL): first-order backward. Dual mode ("bwd2"): all *_dual pointers required.
extern "C" int sgg_attn_step_bwd(const float* ctx, const float* alpha, const float* alpha_dual, const float* dz,
                                 const float* dz_dual, int lddz, float* de, float* de_dual, float* dP, float* dctx, int R, int B,
                                 int L, int C, int accumulate, void* stream) {
  SGG_CHECK_ARG(ctx && alpha && dz && de && dP && dctx, "sgg_attn_step_bwd: null pointer");
  int rc = attn_check("sgg_attn_step_bwd", R, B, L, C);
  if (rc) return rc;
  const int npass = R / B;
  SGG_CHECK_ARG(npass <= ATTN_MAX_PASS, "sgg_attn_step_bwd: at most %d rows per image (got %d)", ATTN_MAX_PASS, npass);
  hipStream_t st = (hipStream_t)stream;
  // full-size feature maps: split the locations of an image over several workgroups (see attn_step_bwd_ctx_kernel)
  const int ls = (B < 256 && L >= 64 && L <= 2048) ? (512 / B < L / 8 ? 512 / B : L / 8) : 1;
  if (ls > 1) {
    if (alpha_dual) {
      SGG_CHECK_ARG(dz_dual && de_dual, "sgg_attn_step_bwd: dual pointers missing");
      hipLaunchKernelGGL(attn_step_bwd_ctx_kernel<Dual>, dim3(B, ls), dim3(256), (size_t)npass * C * sizeof(Dual), st, ctx, alpha,
                         alpha_dual, dz, dz_dual, lddz, de, de_dual, dctx, R, B, L, C, accumulate);
      hipLaunchKernelGGL(attn_step_bwd_softmax_kernel<Dual>, dim3(B), dim3(256), 0, st, alpha, alpha_dual, de, de_dual, dP, R, B, L,
                         accumulate);
    } else {
      hipLaunchKernelGGL(attn_step_bwd_ctx_kernel<float>, dim3(B, ls), dim3(256), (size_t)npass * C * sizeof(float), st, ctx, alpha,
                         nullptr, dz, nullptr, lddz, de, nullptr, dctx, R, B, L, C, accumulate);
      hipLaunchKernelGGL(attn_step_bwd_softmax_kernel<float>, dim3(B), dim3(256), 0, st, alpha, nullptr, de, nullptr, dP, R, B, L,
                         accumulate);
    }
    SGG_LAUNCH_CHECK("sgg_attn_step_bwd");
    return SGG_OK;
  }
  if (alpha_dual) {
    SGG_CHECK_ARG(dz_dual && de_dual, "sgg_attn_step_bwd: dual pointers missing");
    const size_t smb = (size_t)npass * (C + 2 * L) * sizeof(Dual);
    SGG_CHECK_ARG(smb <= 150 * 1024, "sgg_attn_step_bwd: L too large for LDS (%zu bytes)", smb);
    hipLaunchKernelGGL(attn_step_bwd_kernel<Dual>, dim3(B), dim3(256), smb, st, ctx, alpha, alpha_dual, dz, dz_dual, lddz, de,
                       de_dual, dP, dctx, R, B, L, C, accumulate);
  } else {
    const size_t smb = (size_t)npass * (C + 2 * L) * sizeof(float);
    hipLaunchKernelGGL(attn_step_bwd_kernel<float>, dim3(B), dim3(256), smb, st, ctx, alpha, nullptr, dz, nullptr, lddz, de,
                       nullptr, dP, dctx, R, B, L, C, accumulate);
  }
  SGG_LAUNCH_CHECK("sgg_attn_step_bwd");
  return SGG_OK;
}

extern "C" int sgg_lnlstm_gates_fwd(const float* gates, const float* gates_dual, const float* c_prev, const float* c_prev_dual,
                                    const float* ln_params, float* c_new, float* c_new_dual, float* h_new, float* h_new_dual,
                                    int ldh, int R, void* stream) {
  SGG_CHECK_ARG(gates && c_prev && ln_params && c_new && h_new && R > 0 && ldh >= NUM_UNITS, "sgg_lnlstm_gates_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (gates_dual) {
    SGG_CHECK_ARG(c_prev_dual && c_new_dual && h_new_dual, "sgg_lnlstm_gates_fwd: dual pointers missing");
    hipLaunchKernelGGL(lnlstm_gates_fwd_kernel<Dual>, dim3(sgg_cdiv(R, 4)), dim3(256), 0, st, gates, gates_dual, c_prev,
                       c_prev_dual, ln_params, c_new, c_new_dual, h_new, h_new_dual, ldh, R);
  } else {
    hipLaunchKernelGGL(lnlstm_gates_fwd_kernel<float>, dim3(sgg_cdiv(R, 4)), dim3(256), 0, st, gates, nullptr, c_prev, nullptr,
                       ln_params, c_new, nullptr, h_new, nullptr, ldh, R);
  }
  SGG_LAUNCH_CHECK("sgg_lnlstm_gates_fwd");
  return SGG_OK;
}

// dc_new may be NULL (no cotangent on the new cell state). pgrad: [R][10][512] per-row LN parameter gradients.
extern "C" int sgg_lnlstm_gates_bwd(const float* gates, const float* gates_dual, const float* c_prev, const float* c_prev_dual,
                                    const float* ln_params, const float* dh, const float* dh_dual, int lddh, const float* dc_new,
                                    const float* dc_new_dual, float* dgates, float* dgates_dual, float* dc_prev,
                                    float* dc_prev_dual, float* pgrad, int R, void* stream) {
  SGG_CHECK_ARG(gates && c_prev && ln_params && dh && dgates && dc_prev && pgrad && R > 0 && lddh >= NUM_UNITS,
                "sgg_lnlstm_gates_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (gates_dual) {
    SGG_CHECK_ARG(c_prev_dual && dh_dual && dgates_dual && dc_prev_dual && (!dc_new || dc_new_dual),
                  "sgg_lnlstm_gates_bwd: dual pointers missing");
    hipLaunchKernelGGL(lnlstm_gates_bwd_kernel<Dual>, dim3(sgg_cdiv(R, 4)), dim3(256), 0, st, gates, gates_dual, c_prev,
                       c_prev_dual, ln_params, dh, dh_dual, lddh, dc_new, dc_new_dual, dgates, dgates_dual, dc_prev, dc_prev_dual,
                       pgrad, R);
  } else {
    hipLaunchKernelGGL(lnlstm_gates_bwd_kernel<float>, dim3(sgg_cdiv(R, 4)), dim3(256), 0, st, gates, nullptr, c_prev, nullptr,
                       ln_params, dh, nullptr, lddh, dc_new, nullptr, dgates, nullptr, dc_prev, nullptr, pgrad, R);
  }
  SGG_LAUNCH_CHECK("sgg_lnlstm_gates_bwd");
  return SGG_OK;
}

extern "C" int sgg_spatial_mean_fwd(const float* ctx, float* out_c, int ldc, float* out_h, int ldh, int R, int B, int L, int C,
                                    void* stream) {
  SGG_CHECK_ARG(ctx && out_c && out_h && R > 0 && B > 0 && R % B == 0 && L > 0 && C > 0, "sgg_spatial_mean_fwd: bad argument");
  SGG_CHECK_ARG(C % 4 == 0, "sgg_spatial_mean_fwd: C must be a multiple of 4");
  hipLaunchKernelGGL(spatial_mean_fwd_kernel, dim3(B, sgg_cdiv(C, 128)), dim3(256), 0, (hipStream_t)stream, ctx, out_c, ldc, out_h,
                     ldh, R, B, L, C);
  SGG_LAUNCH_CHECK("sgg_spatial_mean_fwd");
  return SGG_OK;
}

extern "C" int sgg_spatial_mean_bwd(const float* dc0, int ldc, const float* dh0, int ldh, float* dctx, int R, int B, int L, int C,
                                    int accumulate, void* stream) {
  SGG_CHECK_ARG(dc0 && dh0 && dctx && R > 0 && B > 0 && R % B == 0 && L > 0 && C > 0, "sgg_spatial_mean_bwd: bad argument");
  const int gy = L < 16 ? L : 16;
  hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(B, gy), dim3(256), 0, (hipStream_t)stream, dc0, ldc, dh0, ldh, dctx, R, B, L, C,
                     accumulate);
  SGG_LAUNCH_CHECK("sgg_spatial_mean_bwd");
  return SGG_OK;
}

// tall inputs are reduced in two deterministic stages through `workspace` (sgg_colsum_workspace_bytes)
static int colsum_chunks(int rows) { return rows > 512 ? (rows + 127) / 128 : 1; }
extern "C" size_t sgg_colsum_workspace_bytes(int rows, int cols) {
  const int nc = colsum_chunks(rows);
  return nc > 1 ? (size_t)nc * cols * sizeof(float) : 0;
}
extern "C" int sgg_colsum(const float* X, int rows, int cols, int ld, float* out, int accumulate, void* workspace,
                          size_t workspace_bytes, void* stream) {
  SGG_CHECK_ARG(X && out && rows > 0 && cols > 0 && ld >= cols, "sgg_colsum: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nc = colsum_chunks(rows);
  if (nc == 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(sgg_cdiv(cols, 256), 1), dim3(256), 0, st, X, rows, cols, ld, out, 0, rows, accumulate);
  } else {
    if (!workspace || workspace_bytes < (size_t)nc * cols * sizeof(float)) {
      sgg_set_error("sgg_colsum: workspace too small");
      return SGG_ERR_WORKSPACE;
    }
    float* part = (float*)workspace;
    hipLaunchKernelGGL(colsum_kernel, dim3(sgg_cdiv(cols, 256), nc), dim3(256), 0, st, X, rows, cols, ld, part, cols, 128, 0);
    hipLaunchKernelGGL(colsum_kernel, dim3(sgg_cdiv(cols, 256), 1), dim3(256), 0, st, (const float*)part, nc, cols, cols, out, 0, nc,
                       accumulate);
  }
  SGG_LAUNCH_CHECK("sgg_colsum");
  return SGG_OK;
}
