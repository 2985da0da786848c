// tf.contrib.layers.layer_norm(x, activation_fn=tf.nn.elu) over (H,W,C) per sample, gamma/beta [C], eps = 1e-12
// Reference: architectures/generator_with_attention.py:30..66 (13 call sites, 11 live), same lines in the
// discriminator; TF semantics SURVEY.md Appendix A.2 (biased two-pass variance, batch_normalization form).
//
// HBM-bound streaming kernels (float4 per lane, grid-stride chunks of 4096 elements). The per-sample
// reduction spans many workgroups; partial (count, mean, M2) triples are merged with Chan's formula by
// every consumer workgroup (no atomics, deterministic, numerically equal to the two-pass variance).
//
//   forward : ln_stats_partial  -> ln_apply_elu
//   backward: ln_bwd_partial    -> ln_bwd_finalize (dgamma, dbeta, bias grad of the producing conv)
//                               -> ln_bwd_apply    (dy)
//
// Pre-split outputs (out_format 1, "S16": split16.h).  Both outputs of this file are the next convolutions' MFMA operands (a: forward
// and filter gradient of the next layer; dy: input and filter gradient of this one), and every consumer used to split the same f32
// values into two fp16 pieces again while staging them - on the issue port of its MFMA waves.  These kernels are HBM-bound with the
// VALU idle, so they can write the pieces instead: same bytes, each aligned 32-channel group as 32 leading + 32 residual pieces of
// x * 2^e.  The scale has to be fixed BEFORE the tensor is written, so the tensor's amax word holds an upper BOUND of max|x| instead
// of the maximum itself:
//   a  : max|gamma| * max|y - mean| * rstd + max|beta| over the samples (ln_finalize_kernel, from the statistics partials);
//   dy : max(rstd * max|dxhat|) * (2 + max|xhat|) (|mean dxhat| <= max|dxhat|, |mean dxhat xhat| <= max|dxhat| because
//        mean xhat^2 <= 1): ln_bwd_partial publishes the two maxima atomically into two caller-zeroed words (pq), every workgroup
//        of ln_bwd_apply derives the same bound from them, one of them publishes it.
// A bound that is 2^k too large costs k of the 16 binades between the tensor's maximum and the point where the residual piece turns
// subnormal (split16.h: an element 2^-d below the maximum keeps min(23, 39 - d) bits): nothing at the sizes that occur (k <= 4).
#include "split16.h"

#define LN_EPS 1e-12f
#ifndef SGG_LN_WGS
#define SGG_LN_WGS 1536      // workgroups per launch of the streaming LayerNorm kernels (six per CU)
#endif
#define LN_CHUNK 4096  // elements per workgroup iteration: 256 threads x 4 x float4

// Streaming accesses (SGG_LN_NT bit 0: nontemporal stores, bit 1: nontemporal loads).  y and da are read once per pass and not
// again before 400+ MB of other traffic, while a / dy are the next convolution's input: nontemporal LOADS keep the streamed-once
// bytes from displacing the outputs in the L2 / Infinity Cache (whole step 52.16 -> 51.48 ms, two repetitions; nontemporal
// stores on top give that back: 52.15 ms; stores alone 53.0 ms).
#ifndef SGG_LN_NT
#define SGG_LN_NT 2
#endif
__device__ __forceinline__ f32x4 ln_ld(const float* p) {
#if SGG_LN_NT & 2
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
  return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ void ln_st(float* p, f32x4 v) {
#if SGG_LN_NT & 1
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}

struct LnGeom {
  int B, C;
  long long N;   // elements per sample = HW * C
  int HW;
  int G;         // workgroups per sample
  int cpg;       // chunks per workgroup
};

// Valid region of an H x W canvas (odd image sizes run on an even canvas, trunk.py): statistics and gradients cover the
// region only, outputs outside it are written as zeros (= the zero padding the next convolution expects).
struct LnMask {
  int W, y0, x0, Hv, Wv, logC;
};
__device__ __forceinline__ bool ln_valid(const LnMask& m, long long e) {
  const int pix = (int)(e >> m.logC);
  const int py = pix / m.W, px = pix - py * m.W;
  return (unsigned)(py - m.y0) < (unsigned)m.Hv && (unsigned)(px - m.x0) < (unsigned)m.Wv;
}

static LnGeom ln_geom(int B, int HW, int C) {
  LnGeom g;
  g.B = B; g.C = C; g.HW = HW; g.N = (long long)HW * C;
  const int nchunks = (int)((g.N + LN_CHUNK - 1) / LN_CHUNK);
  int G = (SGG_LN_WGS + B - 1) / B;
  if (G > nchunks) G = nchunks;
  if (G < 1) G = 1;
  g.cpg = (nchunks + G - 1) / G;
  g.G = (nchunks + g.cpg - 1) / g.cpg;
  return g;
}

// ---- forward -----------------------------------------------------------------------------------------
template <bool MASK>
__global__ __launch_bounds__(256) void ln_stats_partial_kernel(const float* __restrict__ y, float* __restrict__ part,
                                                               long long N, int G, int cpg, LnMask mk) {
  __shared__ float red[4];
  const int b = blockIdx.y, g = blockIdx.x;
  const float* yb = y + (size_t)b * N;
  float n_run = 0.f, mean_run = 0.f, m2_run = 0.f, dev_run = 0.f;   // dev_run: max |y - running mean| bound
  for (int c = 0; c < cpg; ++c) {
    const long long base = ((long long)g * cpg + c) * LN_CHUNK;
    if (base >= N) break;
    f32x4 v[4];
    float s = 0.f, cnt = 0.f;
    int okm = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = base + (threadIdx.x + 256 * j) * 4;
      if (e < N && (!MASK || ln_valid(mk, e))) {
        v[j] = ln_ld(yb + e);
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        cnt += 4.f;
        okm |= 1 << j;
      } else {
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    float n_c = (float)((N - base) < LN_CHUNK ? (N - base) : LN_CHUNK);
    if constexpr (MASK) {
      n_c = block_sum_256(cnt, red);
      if (n_c == 0.f) continue;          // (uniform: the whole chunk lies outside the valid region)
    }
    const float mean_c = block_sum_256(s, red) / n_c;
    float q = 0.f, dm = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((okm >> j) & 1) {
        const f32x4 d = v[j] - mean_c;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        dm = fmaxf(fmaxf(dm, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
      }
    }
    const float m2_c = block_sum_256(q, red);
    const float n_new = n_run + n_c;
    const float delta = mean_c - mean_run;
    const float mean_new = mean_run + delta * (n_c / n_new);
    // |y - mean_new| <= max|y - own chunk mean| + |own chunk mean - mean_new| for the elements of either part
    dm = wave_max(dm);
    dev_run = fmaxf(dev_run + fabsf(mean_run - mean_new), dm + fabsf(mean_c - mean_new));
    mean_run = mean_new;
    m2_run += m2_c + delta * delta * (n_run * n_c / n_new);
    n_run = n_new;
  }
  // (dev_run holds a per-wave maximum: combine the four waves)
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dev_run;
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + ((size_t)b * G + g) * SGG_TS;
    o[0] = n_run; o[1] = mean_run; o[2] = m2_run;
    o[3] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
}

// merge G partials of sample b -> (mean, rstd); every thread returns the same values
__device__ __forceinline__ void ln_merge(const float* __restrict__ part, int G, float N, float* red, float& mean,
                                         float& rstd) {
  float s = 0.f;
  for (int i = threadIdx.x; i < G; i += 256) s += part[i * SGG_TS + 0] * part[i * SGG_TS + 1];
  mean = block_sum_256(s, red) / N;
  float q = 0.f;
  for (int i = threadIdx.x; i < G; i += 256) {
    const float d = part[i * SGG_TS + 1] - mean;
    q += part[i * SGG_TS + 2] + part[i * SGG_TS + 0] * d * d;
  }
  const float var = block_sum_256(q, red) / N;
  rstd = rsqrtf(var + LN_EPS);
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

template <bool MASK>
__global__ __launch_bounds__(256) void ln_apply_elu_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ part,
                                                           float* __restrict__ a, float* __restrict__ stats, long long N,
                                                           int C, int G, int cpg, float* __restrict__ amax_out, LnMask mk,
                                                           float nvalid) {
  __shared__ float red[4];
  const int b = blockIdx.y, g = blockIdx.x;
  float mean, rstd;
  float amax = 0.f;
  ln_merge(part + (size_t)b * G * SGG_TS, G, MASK ? nvalid : (float)N, red, mean, rstd);
  if (g == 0 && threadIdx.x == 0) {
    stats[b * 2 + 0] = mean;
    stats[b * 2 + 1] = rstd;
  }
  const float* yb = y + (size_t)b * N;
  float* ab = a + (size_t)b * N;
  const int ch = (threadIdx.x * 4) % C;   // constant across chunks: LN_CHUNK and 1024 are multiples of C
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + ch);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + ch);
  const f32x4 inv = gm * rstd;
  const f32x4 shift = bt - inv * mean;
  for (int c = 0; c < cpg; ++c) {
    const long long base = ((long long)g * cpg + c) * LN_CHUNK;
    if (base >= N) break;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = base + (threadIdx.x + 256 * j) * 4;
      if (e < N) {
        if (MASK && !ln_valid(mk, e)) {
          ln_st(ab + e, f32x4{0.f, 0.f, 0.f, 0.f});
          continue;
        }
        const f32x4 v = ln_ld(yb + e);
        f32x4 o = v * inv + shift;
        o[0] = elu1(o[0]); o[1] = elu1(o[1]); o[2] = elu1(o[2]); o[3] = elu1(o[3]);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        ln_st(ab + e, o);
      }
    }
  }
  if (amax_out) block_publish_amax(amax_out, amax, red);   // max |a| (the consumer convolution's f16 scaling)
}

// Statistics only (no apply pass): stats[b] = (mean, rstd) merged from the partials, and an UPPER BOUND of max |ELU(LN(y))|
// published into amax_out: |a| <= |z| = |gamma_c * xhat + beta_c| <= max|gamma| * max|y - mean| * rstd + max|beta|, with
// max|y - mean| <= max over tiles of (tile deviation + |tile mean - mean|).  Consumers that normalise y on the fly
// (sgg_conv2d_nhwc_fwd / _wgrad with an LN prologue) scale their fp16 pieces with it.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, int G, float N, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int C, float* __restrict__ stats,
                                                          float* __restrict__ amax_out) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* pb = part + (size_t)b * G * SGG_TS;
  float mean, rstd;
  ln_merge(pb, G, N, red, mean, rstd);
  float dev = 0.f, gb = 0.f, bb = 0.f;
  for (int i = threadIdx.x; i < G; i += 256) dev = fmaxf(dev, pb[i * SGG_TS + 3] + fabsf(pb[i * SGG_TS + 1] - mean));
  for (int c = threadIdx.x; c < C; c += 256) {
    gb = fmaxf(gb, fabsf(gamma[c]));
    bb = fmaxf(bb, fabsf(beta[c]));
  }
  dev = wave_max(dev); gb = wave_max(gb); bb = wave_max(bb);
  __shared__ float mx[3][4];
  if ((threadIdx.x & 63) == 0) { mx[0][threadIdx.x >> 6] = dev; mx[1][threadIdx.x >> 6] = gb; mx[2][threadIdx.x >> 6] = bb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[b * 2 + 0] = mean;
    stats[b * 2 + 1] = rstd;
    if (amax_out) {
      const float d = fmaxf(fmaxf(mx[0][0], mx[0][1]), fmaxf(mx[0][2], mx[0][3]));
      const float g = fmaxf(fmaxf(mx[1][0], mx[1][1]), fmaxf(mx[1][2], mx[1][3]));
      const float bt = fmaxf(fmaxf(mx[2][0], mx[2][1]), fmaxf(mx[2][2], mx[2][3]));
      atomic_amax(amax_out, g * d * rstd + bt);
    }
  }
}

// ---- backward ----------------------------------------------------------------------------------------
// per workgroup: sspart[b][g] = (sum dxhat, sum dxhat*xhat); chpart[b][g][3][C] = (sum dn, sum dn*xhat, sum xhat)
template <bool MASK>
__global__ __launch_bounds__(256) void ln_bwd_partial_kernel(const float* __restrict__ y, const float* __restrict__ da,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ stats, float* __restrict__ sspart,
                                                             float* __restrict__ chpart, float* __restrict__ pq, long long N, int C,
                                                             int G, int cpg, LnMask mk) {
  __shared__ float red[4];
  __shared__ float chs[256 * 12];
  float mxd = 0.f, mxx = 0.f;          // max |dxhat|, max |xhat| of this workgroup's elements (the bound of max|dy|: file header)
  // (computed unconditionally: two v_max per element in an HBM-bound kernel)
  const int b = blockIdx.y, g = blockIdx.x;
  const float mean = stats[b * 2 + 0], rstd = stats[b * 2 + 1];
  const float* yb = y + (size_t)b * N;
  const float* db = da + (size_t)b * N;
  const int ch = (threadIdx.x * 4) % C;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + ch);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + ch);
  f32x4 sA = {0.f, 0.f, 0.f, 0.f}, sB = sA, sX = sA;
  float s1 = 0.f, s2 = 0.f;
  for (int c = 0; c < cpg; ++c) {
    const long long base = ((long long)g * cpg + c) * LN_CHUNK;
    if (base >= N) break;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = base + (threadIdx.x + 256 * j) * 4;
      if (e < N && (!MASK || ln_valid(mk, e))) {
        const f32x4 v = ln_ld(yb + e);
        const f32x4 d = ln_ld(db + e);
        const f32x4 xh = (v - mean) * rstd;
        const f32x4 n = xh * gm + bt;
        f32x4 dn;
#pragma unroll
        for (int q = 0; q < 4; ++q) dn[q] = d[q] * (n[q] > 0.f ? 1.f : __expf(n[q]));
        const f32x4 dxh = dn * gm;
        sA += dn; sB += dn * xh; sX += xh;
        s1 += (dxh[0] + dxh[1]) + (dxh[2] + dxh[3]);
        const f32x4 t = dxh * xh;
        s2 += (t[0] + t[1]) + (t[2] + t[3]);
        mxd = fmaxf(fmaxf(mxd, fmaxf(fabsf(dxh[0]), fabsf(dxh[1]))), fmaxf(fabsf(dxh[2]), fabsf(dxh[3])));
        mxx = fmaxf(fmaxf(mxx, fmaxf(fabsf(xh[0]), fabsf(xh[1]))), fmaxf(fabsf(xh[2]), fabsf(xh[3])));
      }
    }
  }
  const float S1 = block_sum_256(s1, red);
  const float S2 = block_sum_256(s2, red);
  mxd = wave_max(mxd);
  mxx = wave_max(mxx);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { chs[threadIdx.x >> 6] = mxd; chs[4 + (threadIdx.x >> 6)] = mxx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sspart[((size_t)b * G + g) * 2 + 0] = S1;
    sspart[((size_t)b * G + g) * 2 + 1] = S2;
    if (pq) {      // (pre-split dy: the two maxima its bound is made of)
      atomic_amax(pq, rstd * fmaxf(fmaxf(chs[0], chs[1]), fmaxf(chs[2], chs[3])));
      atomic_amax(pq + 1, fmaxf(fmaxf(chs[4], chs[5]), fmaxf(chs[6], chs[7])));
    }
  }
  // per-channel reduce across threads with equal (tid*4) % C
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    chs[threadIdx.x * 12 + q] = sA[q];
    chs[threadIdx.x * 12 + 4 + q] = sB[q];
    chs[threadIdx.x * 12 + 8 + q] = sX[q];
  }
  __syncthreads();
  const int tpc = C / 4;                 // distinct thread classes (C <= 1024)
  const int reps = 256 / tpc;            // threads per class (>= 1 when C <= 1024)
  float* o = chpart + ((size_t)b * G + g) * 3 * C;
  for (int item = threadIdx.x; item < 3 * C; item += 256) {
    const int which = item / C, cc = item % C;
    const int t0 = cc / 4, q = cc % 4;
    float s = 0.f;
    if (reps >= 1)
      for (int r = 0; r < reps; ++r) s += chs[(t0 + r * tpc) * 12 + which * 4 + q];
    o[item] = s;
  }
}

// grid = C/32 workgroups; 1024 threads = 32 channels x 32 lanes over the samples
#define LNF_BL 32
__device__ __forceinline__ void ln_bwd_finalize_body(const float* __restrict__ sspart, const float* __restrict__ chpart,
                                                     const float* __restrict__ gamma, const float* __restrict__ stats,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dbias, int B, int C, int G, int HW, int cgroup, float* sm) {
  // sm: [B][2] per-sample (s1/N, s2/N), then [LNF_BL][32][3] reduce
  float* ssb = sm;
  float* redc = sm + 2 * B;
  const float invN = 1.f / ((float)HW * (float)C);
  for (int b = threadIdx.x; b < B; b += 1024) {
    float a1 = 0.f, a2 = 0.f;
    for (int g = 0; g < G; ++g) {
      a1 += sspart[((size_t)b * G + g) * 2 + 0];
      a2 += sspart[((size_t)b * G + g) * 2 + 1];
    }
    ssb[b * 2 + 0] = a1 * invN;
    ssb[b * 2 + 1] = a2 * invN;
  }
  __syncthreads();
  const int cl = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int c = cgroup * 32 + cl;
  float dg = 0.f, dbt = 0.f, dbs = 0.f;
  if (c < C) {
    const float gm = gamma[c];
    for (int b = bl; b < B; b += LNF_BL) {
      float A = 0.f, Bc = 0.f, X = 0.f;
      for (int g = 0; g < G; ++g) {
        const float* o = chpart + ((size_t)b * G + g) * 3 * C;
        A += o[c]; Bc += o[C + c]; X += o[2 * C + c];
      }
      dbt += A; dg += Bc;
      dbs += stats[b * 2 + 1] * (gm * A - (float)HW * ssb[b * 2 + 0] - X * ssb[b * 2 + 1]);
    }
  }
  redc[(bl * 32 + cl) * 3 + 0] = dg;
  redc[(bl * 32 + cl) * 3 + 1] = dbt;
  redc[(bl * 32 + cl) * 3 + 2] = dbs;
  __syncthreads();
  if (bl == 0 && c < C) {
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    for (int k = 0; k < LNF_BL; ++k) {
      r0 += redc[(k * 32 + cl) * 3 + 0];
      r1 += redc[(k * 32 + cl) * 3 + 1];
      r2 += redc[(k * 32 + cl) * 3 + 2];
    }
    dgamma[c] = r0; dbeta[c] = r1;
    if (dbias) dbias[c] = r2;
  }
}
__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const float* __restrict__ sspart, const float* __restrict__ chpart,
                                                               const float* __restrict__ gamma, const float* __restrict__ stats,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dbias, int B, int C, int G, int HW) {
  extern __shared__ float sm[];
  ln_bwd_finalize_body(sspart, chpart, gamma, stats, dgamma, dbeta, dbias, B, C, G, HW, blockIdx.x, sm);
}

// The parameter-gradient reductions of SEVERAL LayerNorms in one launch (sgg_layernorm_hwc_bwd_finalize): an encoder backward
// defers them (dgamma == NULL in sgg_layernorm_hwc_elu_bwd, one workspace per layer) - as launches of their own they are eleven
// 16-workgroup kernels of 20 us each on the critical path of the backward.
#define SGG_LNF_MAX 16
struct LnfLayer {
  const float* ws;         // [B][G][2] sums (sspart) as sgg_layernorm_hwc_elu_bwd left them in the layer's workspace; the
                           // [B][G][3][C] channel sums follow
  const float* gamma;
  const float* stats;
  float* dgamma;
  float* dbeta;
  float* dbias;
  int B, C, G, HW;         // HW: pixels of the valid window
};
struct LnfArgs {
  LnfLayer L[SGG_LNF_MAX];
  int first[SGG_LNF_MAX + 1];
  int nl;
};
__global__ __launch_bounds__(1024) void ln_bwd_finalize_batch_kernel(LnfArgs a) {
  extern __shared__ float sm[];
  int l = 0;
  while (l + 1 < a.nl && (int)blockIdx.x >= a.first[l + 1]) ++l;
  const LnfLayer& L = a.L[l];
  const float* sspart = L.ws;
  const float* chpart = sspart + (size_t)L.B * L.G * 2;
  ln_bwd_finalize_body(sspart, chpart, L.gamma, L.stats, L.dgamma, L.dbeta, L.dbias, L.B, L.C, L.G, L.HW, blockIdx.x - a.first[l], sm);
}

template <bool MASK>
__global__ __launch_bounds__(256) void ln_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ da,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ stats, const float* __restrict__ sspart,
                                                           float* __restrict__ dy, long long N, int C, int G, int cpg,
                                                           float* __restrict__ amax_out, LnMask mk, float nvalid) {
  __shared__ float red[4];
  const int b = blockIdx.y, g = blockIdx.x;
  const float mean = stats[b * 2 + 0], rstd = stats[b * 2 + 1];
  float amax = 0.f;
  float a1 = 0.f, a2 = 0.f;
  for (int i = threadIdx.x; i < G; i += 256) {
    a1 += sspart[((size_t)b * G + i) * 2 + 0];
    a2 += sspart[((size_t)b * G + i) * 2 + 1];
  }
  const float m1 = block_sum_256(a1, red) / (MASK ? nvalid : (float)N);
  const float m2 = block_sum_256(a2, red) / (MASK ? nvalid : (float)N);
  const float* yb = y + (size_t)b * N;
  const float* db = da + (size_t)b * N;
  float* ob = dy + (size_t)b * N;
  const int ch = (threadIdx.x * 4) % C;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + ch);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + ch);
  for (int c = 0; c < cpg; ++c) {
    const long long base = ((long long)g * cpg + c) * LN_CHUNK;
    if (base >= N) break;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = base + (threadIdx.x + 256 * j) * 4;
      if (e < N) {
        if (MASK && !ln_valid(mk, e)) {
          ln_st(ob + e, f32x4{0.f, 0.f, 0.f, 0.f});
          continue;
        }
        const f32x4 v = ln_ld(yb + e);
        const f32x4 d = ln_ld(db + e);
        const f32x4 xh = (v - mean) * rstd;
        const f32x4 n = xh * gm + bt;
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float dn = d[q] * (n[q] > 0.f ? 1.f : __expf(n[q]));
          o[q] = rstd * (dn * gm[q] - m1 - xh[q] * m2);
          amax = fmaxf(amax, fabsf(o[q]));
        }
        ln_st(ob + e, o);
      }
    }
  }
  if (amax_out) block_publish_amax(amax_out, amax, red);
}

// ---- pre-split ("S16") outputs -------------------------------------------------------------------------------------------
// Same thread <-> element map as the f32 kernels (lane t holds 4 consecutive channels: perfectly coalesced 16-byte loads).  A lane's
// four values make 8 bytes of leading and 8 bytes of residual pieces; lanes t and t ^ 1 hold the two halves of an 8-channel item, so
// they swap one half each through a DPP quad permute (no LDS): the even lane then stores the item's 16 bytes of LEADING pieces, the odd
// lane its 16 bytes of RESIDUAL pieces - one 16-byte store per lane, and the eight lanes of a 32-channel group write its whole
// 128-byte line with one instruction.  EVERY lane of the wave calls this; `in`: the lane's channels exist (e < N; N % 8 == 0, so a
// pair is in or out together).
__device__ __forceinline__ unsigned s16_swap1(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]: lane ^ 1
}
__device__ __forceinline__ void s16_store4(float* tensor_b, long long e, const f32x4& o, float scale, bool in) {
  unsigned hi0, lo0, hi1, lo1;
  f16_split2(o[0] * scale, o[1] * scale, hi0, lo0);
  f16_split2(o[2] * scale, o[3] * scale, hi1, lo1);
  const bool odd = (threadIdx.x & 1) != 0;
  const unsigned r0 = s16_swap1(odd ? hi0 : lo0), r1 = s16_swap1(odd ? hi1 : lo1);
  const u32x4 v = odd ? u32x4{r0, r1, lo0, lo1} : u32x4{hi0, hi1, r0, r1};
  const long long eb = e * 4;
  unsigned char* g = reinterpret_cast<unsigned char*>(tensor_b) + (eb & ~127LL) + ((eb & 127LL) >> 5) * 16 + (odd ? 64 : 0);
  if (in) *reinterpret_cast<u32x4*>(g) = v;
}

// a = ELU(LN(y)) written pre-split; stats [B][2] and the bound in *amax are final (ln_finalize_kernel ran before)
template <bool MASK>
__global__ __launch_bounds__(256) void ln_apply_elu_s16_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ stats,
                                                               float* __restrict__ a, long long N, int C, int cpg,
                                                               const float* __restrict__ amax, LnMask mk) {
  const int b = blockIdx.y, g = blockIdx.x;
  const float mean = stats[b * 2 + 0], rstd = stats[b * 2 + 1];
  const float scale = ldexpf(1.f, scale_exp_from_amax(*amax));
  const float* yb = y + (size_t)b * N;
  float* ab = a + (size_t)b * N;
  const int ch = (threadIdx.x * 4) % C;   // constant across chunks: LN_CHUNK and 1024 are multiples of C
  const f32x4 inv = *reinterpret_cast<const f32x4*>(gamma + ch) * rstd;
  const f32x4 shift = *reinterpret_cast<const f32x4*>(beta + ch) - inv * mean;
  for (int c = 0; c < cpg; ++c) {
    const long long base = ((long long)g * cpg + c) * LN_CHUNK;
    if (base >= N) break;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = base + (threadIdx.x + 256 * j) * 4;
      const bool in = e < N;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (in && (!MASK || ln_valid(mk, e))) {
        const f32x4 v = ln_ld(yb + e);
        o = v * inv + shift;
        o[0] = elu1(o[0]); o[1] = elu1(o[1]); o[2] = elu1(o[2]); o[3] = elu1(o[3]);
      }
      s16_store4(ab, e, o, scale, in);
    }
  }
}

// dy of the LayerNorm backward written pre-split.  pq[0] = max over all workgroups of ln_bwd_partial of rstd * max|dxhat|, pq[1] =
// their max|xhat| (atomic maxima, the words zeroed by the caller): bound = pq[0] * (2 + pq[1]) >= max|dy| (file header; the two
// maxima may come from different samples: only looser).  Workgroup (0, 0) publishes the bound in *amax_out for the consumers.
template <bool MASK>
__global__ __launch_bounds__(256) void ln_bwd_apply_s16_kernel(const float* __restrict__ y, const float* __restrict__ da,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ stats, const float* __restrict__ sspart,
                                                               const float* __restrict__ pq, float* __restrict__ dy, long long N,
                                                               int C, int G, int cpg, float* __restrict__ amax_out, LnMask mk,
                                                               float nvalid) {
  __shared__ float red[4];
  const int b = blockIdx.y, g = blockIdx.x;
  const float bound = 1.0001f * pq[0] * (2.f + pq[1]);
  if (amax_out && b == 0 && g == 0 && threadIdx.x == 0) *amax_out = bound;
  const float scale = ldexpf(1.f, scale_exp_from_amax(bound));
  const float mean = stats[b * 2 + 0], rstd = stats[b * 2 + 1];
  float a1 = 0.f, a2 = 0.f;
  for (int i = threadIdx.x; i < G; i += 256) {
    a1 += sspart[((size_t)b * G + i) * 2 + 0];
    a2 += sspart[((size_t)b * G + i) * 2 + 1];
  }
  const float m1 = block_sum_256(a1, red) / (MASK ? nvalid : (float)N);
  const float m2 = block_sum_256(a2, red) / (MASK ? nvalid : (float)N);
  const float* yb = y + (size_t)b * N;
  const float* db = da + (size_t)b * N;
  float* ob = dy + (size_t)b * N;
  const int ch = (threadIdx.x * 4) % C;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + ch);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + ch);
  for (int c = 0; c < cpg; ++c) {
    const long long base = ((long long)g * cpg + c) * LN_CHUNK;
    if (base >= N) break;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = base + (threadIdx.x + 256 * j) * 4;
      const bool in = e < N;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (in && (!MASK || ln_valid(mk, e))) {
        const f32x4 v = ln_ld(yb + e);
        const f32x4 d = ln_ld(db + e);
        const f32x4 xh = (v - mean) * rstd;
        const f32x4 n = xh * gm + bt;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float dn = d[q] * (n[q] > 0.f ? 1.f : __expf(n[q]));
          o[q] = rstd * (dn * gm[q] - m1 - xh[q] * m2);
        }
      }
      s16_store4(ob, e, o, scale, in);
    }
  }
}

// ---- host ----------------------------------------------------------------------------------------------
extern "C" size_t sgg_layernorm_hwc_elu_workspace_bytes(int B, int HW, int C) {
  const LnGeom g = ln_geom(B, HW, C);
  // fwd: part[B][G][SGG_TS]; bwd: sspart[B][G][2] + chpart[B][G][3][C]
  return ((size_t)B * g.G * SGG_TS + (size_t)B * g.G * 2 + (size_t)B * g.G * 3 * C) * sizeof(float) + 256;
}

static int ln_check(const char* name, int B, int HW, int C, size_t ws_bytes, void* ws) {
  SGG_CHECK_ARG(B > 0 && HW > 0 && C >= 4, "%s: bad dims", name);
  SGG_CHECK_ARG(C <= 1024 && (LN_CHUNK % C) == 0, "%s: C must be a power of two in [4, 1024] (got %d)", name, C);
  if (!ws || ws_bytes < sgg_layernorm_hwc_elu_workspace_bytes(B, HW, C)) {
    sgg_set_error("%s: workspace too small", name);
    return SGG_ERR_WORKSPACE;
  }
  return SGG_OK;
}

// Valid-region argument of the two entry points below: W == 0 -> every pixel of the HW plane is valid; otherwise the plane is
// an (HW / W) x W canvas whose valid region is rows [y0, y0 + Hv) x columns [x0, x0 + Wv).
static int ln_mask(const char* name, int HW, int C, int W, int y0, int x0, int Hv, int Wv, LnMask& mk, float& nvalid) {
  mk = LnMask{W, y0, x0, Hv, Wv, 0};
  nvalid = (float)((long long)HW * C);
  if (W == 0) return SGG_OK;
  SGG_CHECK_ARG(W > 0 && HW % W == 0 && y0 >= 0 && x0 >= 0 && Hv > 0 && Wv > 0 && y0 + Hv <= HW / W && x0 + Wv <= W,
                "%s: valid region (%d,%d)+(%dx%d) outside the %dx%d canvas", name, y0, x0, Hv, Wv, HW / W, W);
  while ((1 << mk.logC) < C) ++mk.logC;
  nvalid = (float)((long long)Hv * Wv * C);
  return SGG_OK;
}

// tile_stats / n_tile_stats (optional): [B][n_tile_stats][4] (count, mean, M2, max |y - mean|) partials of y already produced by the
// convolution's epilogue (sgg_conv2d_nhwc_fwd); the statistics pass over y is then skipped.  (Not with a valid region: the
// convolution's partials cover the whole canvas.)
extern "C" int sgg_layernorm_hwc_elu_fwd(const float* y, const float* gamma, const float* beta, float* a, float* stats,
                                         float* amax_out, const float* tile_stats, int n_tile_stats, int B, int HW, int C,
                                         int W, int y0, int x0, int Hv, int Wv, int out_format, void* ws, size_t ws_bytes, void* stream) {
  SGG_CHECK_ARG(y && gamma && beta && a && stats, "sgg_layernorm_hwc_elu_fwd: null pointer");
  SGG_CHECK_ARG(out_format == 0 || (out_format == 1 && amax_out && C % 32 == 0),
                "sgg_layernorm_hwc_elu_fwd: out_format 1 (pre-split output) needs the amax word (zeroed by the caller) and C %% 32 == 0");
  int rc = ln_check("sgg_layernorm_hwc_elu_fwd", B, HW, C, ws_bytes, ws);
  if (rc) return rc;
  LnMask mk;
  float nvalid;
  if ((rc = ln_mask("sgg_layernorm_hwc_elu_fwd", HW, C, W, y0, x0, Hv, Wv, mk, nvalid))) return rc;
  const LnGeom g = ln_geom(B, HW, C);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  if (out_format == 1) {
    // statistics (from the convolution's partials, or a pass of our own) -> stats + the bound of max|a| in *amax_out -> apply
    const float* parts = tile_stats;
    int nparts = n_tile_stats;
    if (!(tile_stats && n_tile_stats > 0)) {
      if (W == 0) hipLaunchKernelGGL(ln_stats_partial_kernel<false>, dim3(g.G, B), dim3(256), 0, st, y, part, g.N, g.G, g.cpg, mk);
      else hipLaunchKernelGGL(ln_stats_partial_kernel<true>, dim3(g.G, B), dim3(256), 0, st, y, part, g.N, g.G, g.cpg, mk);
      parts = part;
      nparts = g.G;
    } else {
      SGG_CHECK_ARG(W == 0, "sgg_layernorm_hwc_elu_fwd: tile_stats cannot be combined with a valid region");
    }
    hipLaunchKernelGGL(ln_finalize_kernel, dim3(B), dim3(256), 0, st, parts, nparts, nvalid, gamma, beta, C, stats, amax_out);
    if (W == 0)
      hipLaunchKernelGGL(ln_apply_elu_s16_kernel<false>, dim3(g.G, B), dim3(256), 0, st, y, gamma, beta, (const float*)stats, a, g.N, C,
                         g.cpg, (const float*)amax_out, mk);
    else
      hipLaunchKernelGGL(ln_apply_elu_s16_kernel<true>, dim3(g.G, B), dim3(256), 0, st, y, gamma, beta, (const float*)stats, a, g.N, C,
                         g.cpg, (const float*)amax_out, mk);
    SGG_LAUNCH_CHECK("sgg_layernorm_hwc_elu_fwd");
    return SGG_OK;
  }
  if (tile_stats && n_tile_stats > 0) {
    SGG_CHECK_ARG(W == 0, "sgg_layernorm_hwc_elu_fwd: tile_stats cannot be combined with a valid region");
    hipLaunchKernelGGL(ln_apply_elu_kernel<false>, dim3(g.G, B), dim3(256), 0, st, y, gamma, beta, tile_stats, a, stats, g.N, C,
                       n_tile_stats, g.cpg, amax_out, mk, nvalid);
    SGG_LAUNCH_CHECK("sgg_layernorm_hwc_elu_fwd");
    return SGG_OK;
  }
  if (W == 0) {
    hipLaunchKernelGGL(ln_stats_partial_kernel<false>, dim3(g.G, B), dim3(256), 0, st, y, part, g.N, g.G, g.cpg, mk);
    hipLaunchKernelGGL(ln_apply_elu_kernel<false>, dim3(g.G, B), dim3(256), 0, st, y, gamma, beta, (const float*)part, a, stats,
                       g.N, C, g.G, g.cpg, amax_out, mk, nvalid);
  } else {
    hipLaunchKernelGGL(ln_stats_partial_kernel<true>, dim3(g.G, B), dim3(256), 0, st, y, part, g.N, g.G, g.cpg, mk);
    hipLaunchKernelGGL(ln_apply_elu_kernel<true>, dim3(g.G, B), dim3(256), 0, st, y, gamma, beta, (const float*)part, a, stats,
                       g.N, C, g.G, g.cpg, amax_out, mk, nvalid);
  }
  SGG_LAUNCH_CHECK("sgg_layernorm_hwc_elu_fwd");
  return SGG_OK;
}

// An f32 tensor that no LayerNorm kernel produces (the gradient the attention head hands to `downsampled`,
// architectures/generator_with_attention.py:68,74) converted to the pre-split format, so that its consumers stage it by DMA too.
__global__ __launch_bounds__(256) void presplit16_kernel(const float* __restrict__ x, float* __restrict__ out, long long n,
                                                         const float* __restrict__ amax) {
  const float scale = ldexpf(1.f, scale_exp_from_amax(*amax));
  for (long long base = (long long)blockIdx.x * 1024; base < n; base += (long long)gridDim.x * 1024) {      // (uniform per workgroup:
    const long long e = base + threadIdx.x * 4;                                                            //  every lane exchanges)
    const bool in = e < n;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (in) v = ln_ld(x + e);
    s16_store4(out, e, v, scale, in);
  }
}

extern "C" int sgg_presplit16(const float* x, float* out, long long n, const float* amax, void* stream) {
  SGG_CHECK_ARG(x && out && amax && n > 0 && n % 32 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0,
                "sgg_presplit16: needs whole 32-channel groups (n %% 32 == 0) and 16-byte aligned tensors");
  long long wgs = (n + 1023) / 1024;
  if (wgs > 2048) wgs = 2048;
  hipLaunchKernelGGL(presplit16_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, x, out, n, amax);
  SGG_LAUNCH_CHECK("sgg_presplit16");
  return SGG_OK;
}

// Statistics without the apply pass, for consumers that normalise y on the fly (LN prologue of sgg_conv2d_nhwc_fwd / _wgrad):
// stats[b] = (mean, rstd) from the convolution's tile partials, and amax_out max-ed with an upper bound of max|ELU(LN(y))|.
extern "C" int sgg_layernorm_hwc_finalize(const float* tile_stats, int n_tile_stats, const float* gamma, const float* beta, float* stats,
                                          float* amax_out, int B, int HW, int C, void* stream) {
  SGG_CHECK_ARG(tile_stats && n_tile_stats > 0 && gamma && beta && stats && B > 0 && HW > 0 && C > 0, "sgg_layernorm_hwc_finalize: bad argument");
  hipLaunchKernelGGL(ln_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, tile_stats, n_tile_stats, (float)((long long)HW * C),
                     gamma, beta, C, stats, amax_out);
  SGG_LAUNCH_CHECK("sgg_layernorm_hwc_finalize");
  return SGG_OK;
}

extern "C" int sgg_layernorm_hwc_elu_bwd(const float* y, const float* da, const float* gamma, const float* beta,
                                         const float* stats, float* dy, float* dgamma, float* dbeta, float* dbias_prev,
                                         float* amax_out, int B, int HW, int C, int W, int y0, int x0, int Hv, int Wv, int out_format,
                                         float* pq, void* ws, size_t ws_bytes, void* stream) {
  SGG_CHECK_ARG(y && da && gamma && beta && stats && dy && (!dgamma == !dbeta), "sgg_layernorm_hwc_elu_bwd: null pointer");
  SGG_CHECK_ARG(out_format == 0 || (out_format == 1 && amax_out && pq && C % 32 == 0),
                "sgg_layernorm_hwc_elu_bwd: out_format 1 (pre-split output) needs the amax word, two zeroed words pq and C %% 32 == 0");
  if (out_format == 0) pq = nullptr;
  int rc = ln_check("sgg_layernorm_hwc_elu_bwd", B, HW, C, ws_bytes, ws);
  if (rc) return rc;
  LnMask mk;
  float nvalid;
  if ((rc = ln_mask("sgg_layernorm_hwc_elu_bwd", HW, C, W, y0, x0, Hv, Wv, mk, nvalid))) return rc;
  const LnGeom g = ln_geom(B, HW, C);
  hipStream_t st = (hipStream_t)stream;
  float* sspart = (float*)ws + (size_t)B * g.G * SGG_TS;
  float* chpart = sspart + (size_t)B * g.G * 2;
  const size_t sm = (size_t)(2 * B + LNF_BL * 32 * 3) * sizeof(float);
  const int hw_valid = W == 0 ? HW : Hv * Wv;
  if (W == 0)
    hipLaunchKernelGGL(ln_bwd_partial_kernel<false>, dim3(g.G, B), dim3(256), 0, st, y, da, gamma, beta, stats, sspart, chpart,
                       pq, g.N, C, g.G, g.cpg, mk);
  else
    hipLaunchKernelGGL(ln_bwd_partial_kernel<true>, dim3(g.G, B), dim3(256), 0, st, y, da, gamma, beta, stats, sspart, chpart,
                       pq, g.N, C, g.G, g.cpg, mk);
  if (dgamma)      // (NULL: the caller reduces the partials later, several layers at once: sgg_layernorm_hwc_bwd_finalize)
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3(sgg_cdiv(C, 32)), dim3(1024), sm, st, (const float*)sspart,
                       (const float*)chpart, gamma, stats, dgamma, dbeta, dbias_prev, B, C, g.G, hw_valid);
  if (out_format == 1) {
    if (W == 0)
      hipLaunchKernelGGL(ln_bwd_apply_s16_kernel<false>, dim3(g.G, B), dim3(256), 0, st, y, da, gamma, beta, stats, (const float*)sspart,
                         (const float*)pq, dy, g.N, C, g.G, g.cpg, amax_out, mk, nvalid);
    else
      hipLaunchKernelGGL(ln_bwd_apply_s16_kernel<true>, dim3(g.G, B), dim3(256), 0, st, y, da, gamma, beta, stats, (const float*)sspart,
                         (const float*)pq, dy, g.N, C, g.G, g.cpg, amax_out, mk, nvalid);
    SGG_LAUNCH_CHECK("sgg_layernorm_hwc_elu_bwd");
    return SGG_OK;
  }
  if (W == 0)
    hipLaunchKernelGGL(ln_bwd_apply_kernel<false>, dim3(g.G, B), dim3(256), 0, st, y, da, gamma, beta, stats,
                       (const float*)sspart, dy, g.N, C, g.G, g.cpg, amax_out, mk, nvalid);
  else
    hipLaunchKernelGGL(ln_bwd_apply_kernel<true>, dim3(g.G, B), dim3(256), 0, st, y, da, gamma, beta, stats,
                       (const float*)sspart, dy, g.N, C, g.G, g.cpg, amax_out, mk, nvalid);
  SGG_LAUNCH_CHECK("sgg_layernorm_hwc_elu_bwd");
  return SGG_OK;
}

// The two per-sample means of the backward, as every workgroup of ln_bwd_apply_kernel derives them from the partial sums (the same
// strided per-thread sums, the same block reduction, the same division: bit-identical values): means[b] = (mean dxhat, mean dxhat * xhat).
__global__ __launch_bounds__(256) void ln_bwd_means_kernel(const float* __restrict__ sspart, float* __restrict__ means, int G, float n) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float a1 = 0.f, a2 = 0.f;
  for (int i = threadIdx.x; i < G; i += 256) {
    a1 += sspart[((size_t)b * G + i) * 2 + 0];
    a2 += sspart[((size_t)b * G + i) * 2 + 1];
  }
  const float m1 = block_sum_256(a1, red) / n;
  const float m2 = block_sum_256(a2, red) / n;
  if (threadIdx.x == 0) {
    means[2 * b + 0] = m1;
    means[2 * b + 1] = m2;
  }
}

// The REDUCTION half of sgg_layernorm_hwc_elu_bwd without its apply pass, for a consumer that computes dy itself
// (sgg_conv2d_nhwc_wgrad_c3_ln: conv1_1's filter gradient, the only consumer of the first LayerNorm's dy): the partial sums go to
// `ws` exactly as sgg_layernorm_hwc_elu_bwd leaves them (dgamma / dbeta / dbias_prev: pass NULL and reduce them later with
// sgg_layernorm_hwc_bwd_finalize, or pass all of them), and means [B][2] receives (mean dxhat, mean dxhat * xhat) per sample.
// Whole planes only (no valid region).
extern "C" int sgg_layernorm_hwc_elu_bwd_sums(const float* y, const float* da, const float* gamma, const float* beta, const float* stats,
                                              float* means, float* dgamma, float* dbeta, float* dbias_prev, int B, int HW, int C, void* ws,
                                              size_t ws_bytes, void* stream) {
  SGG_CHECK_ARG(y && da && gamma && beta && stats && means && (!dgamma == !dbeta), "sgg_layernorm_hwc_elu_bwd_sums: null pointer");
  int rc = ln_check("sgg_layernorm_hwc_elu_bwd_sums", B, HW, C, ws_bytes, ws);
  if (rc) return rc;
  LnMask mk;
  float nvalid;
  if ((rc = ln_mask("sgg_layernorm_hwc_elu_bwd_sums", HW, C, 0, 0, 0, 0, 0, mk, nvalid))) return rc;
  const LnGeom g = ln_geom(B, HW, C);
  hipStream_t st = (hipStream_t)stream;
  float* sspart = (float*)ws + (size_t)B * g.G * SGG_TS;
  float* chpart = sspart + (size_t)B * g.G * 2;
  const size_t sm = (size_t)(2 * B + LNF_BL * 32 * 3) * sizeof(float);
  hipLaunchKernelGGL(ln_bwd_partial_kernel<false>, dim3(g.G, B), dim3(256), 0, st, y, da, gamma, beta, stats, sspart, chpart,
                     (float*)nullptr, g.N, C, g.G, g.cpg, mk);
  if (dgamma)
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3(sgg_cdiv(C, 32)), dim3(1024), sm, st, (const float*)sspart,
                       (const float*)chpart, gamma, stats, dgamma, dbeta, dbias_prev, B, C, g.G, HW);
  hipLaunchKernelGGL(ln_bwd_means_kernel, dim3(B), dim3(256), 0, st, (const float*)sspart, means, g.G, (float)g.N);
  SGG_LAUNCH_CHECK("sgg_layernorm_hwc_elu_bwd_sums");
  return SGG_OK;
}

struct sgg_ln_finalize_desc {      // mirrors include/sgg_hip.h
  const void* workspace;
  const float* gamma;
  const float* stats;
  float* dgamma;
  float* dbeta;
  float* dbias_prev;
  int B, HW, C, HW_valid;
};
extern "C" int sgg_layernorm_hwc_bwd_finalize(const sgg_ln_finalize_desc* layers, int n, void* stream) {
  SGG_CHECK_ARG(layers && n > 0 && n <= SGG_LNF_MAX, "sgg_layernorm_hwc_bwd_finalize: 1..%d layers", SGG_LNF_MAX);
  LnfArgs a;
  a.nl = n;
  int tot = 0, maxB = 0;
  for (int i = 0; i < n; ++i) {
    const sgg_ln_finalize_desc& d = layers[i];
    SGG_CHECK_ARG(d.workspace && d.gamma && d.stats && d.dgamma && d.dbeta && d.B > 0 && d.HW > 0 && d.C >= 4 && d.HW_valid > 0 &&
                      d.HW_valid <= d.HW, "sgg_layernorm_hwc_bwd_finalize: layer %d: bad argument", i);
    const LnGeom g = ln_geom(d.B, d.HW, d.C);
    LnfLayer& L = a.L[i];
    L.ws = (const float*)d.workspace + (size_t)d.B * g.G * SGG_TS;      // (behind the forward's partials)
    L.G = g.G;
    L.gamma = d.gamma; L.stats = d.stats; L.dgamma = d.dgamma; L.dbeta = d.dbeta; L.dbias = d.dbias_prev;
    L.B = d.B; L.C = d.C; L.HW = d.HW_valid;
    a.first[i] = tot;
    tot += sgg_cdiv(d.C, 32);
    if (d.B > maxB) maxB = d.B;
  }
  a.first[n] = tot;
  const size_t sm = (size_t)(2 * maxB + LNF_BL * 32 * 3) * sizeof(float);
  hipLaunchKernelGGL(ln_bwd_finalize_batch_kernel, dim3(tot), dim3(1024), sm, (hipStream_t)stream, a);
  SGG_LAUNCH_CHECK("sgg_layernorm_hwc_bwd_finalize");
  return SGG_OK;
}
