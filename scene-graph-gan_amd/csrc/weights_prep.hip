// All per-layer weight re-layouts of one encoder after an optimiser step, in three launches instead of ~45:
//   HWIO -> HWOI transpose, max|w| (the fp16 scaling of precision 1 / 2), the 9-tap space-to-depth kernel of conv1_3, and both
//   pre-split 16-bit copies (planes or MFMA B-fragment order) for the forward and the dgrad direction.
// Reference: the variables of tf.layers.conv2d (architectures/generator_with_attention.py:29-68) as updated by
// optimizer.minimize (train.py:265-266); the re-layouts are this implementation's operand formats (include/sgg_hip.h).
// Per layer these were sgg_hwio_to_hwoi + sgg_absmax + sgg_conv_s2d_weights + 2 x sgg_conv_split_weights(_frag): 5 us kernels, two
// networks x 11 layers x 4-6 launches = 0.6 ms of a 50 ms step.  Same arithmetic, same outputs (the single-layer entry points stay).
#include "split16.h"

#define SGG_WP_MAX 16

struct WpLayer {
  const float* w;      // HWIO [taps][cin][cout]
  float* w_t;          // HWOI [taps][cout][cin]
  float* w3;           // layout 3: [9][4*cin][cout] (space-to-depth kernel), else null
  float* w3_t;         //           [9][cout][4*cin]
  void* ws_fwd;        // pre-split forward operand (source: w_t or w3_t), may be null
  void* ws_bwd;        // pre-split dgrad operand (source: w or w3), may be null
  float* amax;         // device word, max|w| (precision 1 / 2), may be null
  int taps, cin, cout, lay_f, lay_b;
};
struct WpArgs {
  WpLayer L[SGG_WP_MAX];
  int first[SGG_WP_MAX + 1];      // first workgroup of each layer in the launch at hand
  int nl;
};

__device__ __forceinline__ int wp_layer_of(const WpArgs& a, int blk) {
  int l = 0;
  while (l + 1 < a.nl && blk >= a.first[l + 1]) ++l;
  return l;
}

__global__ void wp_zero_kernel(WpArgs a) {
  if ((int)threadIdx.x < a.nl && a.L[threadIdx.x].amax) *a.L[threadIdx.x].amax = 0.f;
}

// phase A: 32 x 32 tile transposes + max|w|; for a layout-3 layer also w3 / w3_t (gathered element-wise from w).
// WP_TPB tiles per workgroup with all of their loads in flight at once: one tile per workgroup (11 200 workgroups of four loads and
// four stores per thread for the 45.7 MB of an encoder) ran at 1.2 TB/s - 77 us on the tail of every update (round 5).
#define WP_TPB 4
__global__ __launch_bounds__(256) void wp_transpose_kernel(WpArgs a) {
  __shared__ float tile[WP_TPB][32][33];
  __shared__ float red[4];
  const int l = wp_layer_of(a, blockIdx.x);
  const WpLayer& L = a.L[l];
  int lb = blockIdx.x - a.first[l];
  const int tci = (L.cin + 31) >> 5, tco = (L.cout + 31) >> 5;
  const int ntr = L.taps * tci * tco;
  const int ntrb = (ntr + WP_TPB - 1) / WP_TPB;
  if (lb < ntrb) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float v[WP_TPB][4];
    int tap[WP_TPB], ci0[WP_TPB], co0[WP_TPB];
#pragma unroll
    for (int t = 0; t < WP_TPB; ++t) {
      const int ti = lb * WP_TPB + t;
      const int r = ti % (tci * tco);
      tap[t] = ti / (tci * tco); ci0[t] = (r / tco) * 32; co0[t] = (r % tco) * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ci = ci0[t] + ty + 8 * k, co = co0[t] + tx;
        v[t][k] = (ti < ntr && ci < L.cin && co < L.cout) ? L.w[((size_t)tap[t] * L.cin + ci) * L.cout + co] : 0.f;
      }
    }
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < WP_TPB; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        tile[t][ty + 8 * k][tx] = v[t][k];
        m = fmaxf(m, fabsf(v[t][k]));
      }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < WP_TPB; ++t) {
      if (lb * WP_TPB + t >= ntr) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int co = co0[t] + ty + 8 * k, ci = ci0[t] + tx;
        if (ci < L.cin && co < L.cout) L.w_t[((size_t)tap[t] * L.cout + co) * L.cin + ci] = tile[t][tx][ty + 8 * k];
      }
    }
    if (L.amax) {
      m = wave_max(m);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
      __syncthreads();
      if (threadIdx.x == 0) atomic_amax(L.amax, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    }
    return;
  }
  // space-to-depth kernel (sgg_conv_s2d_weights) and its transpose: element idx of w3 [u][v][(q, ci)][co]
  lb -= ntrb;
  const long long idx = (long long)lb * 256 + threadIdx.x;
  const long long n = 36LL * L.cin * L.cout;
  if (idx >= n) return;
  const int co = (int)(idx % L.cout);
  long long r = idx / L.cout;
  const int ci = (int)(r % L.cin); r /= L.cin;
  const int q = (int)(r % 4); r /= 4;
  const int v = (int)(r % 3), u = (int)(r / 3);
  const int kh = 2 * u + (q >> 1) - 1, kw = 2 * v + (q & 1) - 1;
  const float val = (kh >= 0 && kh < 5 && kw >= 0 && kw < 5) ? L.w[(((size_t)kh * 5 + kw) * L.cin + ci) * L.cout + co] : 0.f;
  L.w3[idx] = val;
  L.w3_t[(((size_t)(u * 3 + v)) * L.cout + co) * (4 * L.cin) + q * L.cin + ci] = val;
}

// phase B: the pre-split copies.  Per layer the forward operand's items come first, then the dgrad operand's.
//   layout 0: planes [P][n] (split_weights_kernel); layouts 1..3: MFMA B fragments [tap][C/32][N/32][k-step][plane][lane] x 16 B;
//   layout 4: the fragments of the K = 32 MFMA shape [tap][C/32][N/16][plane][lane] x 16 B (same size, same number of items)
template <int P, bool HALF>
__global__ __launch_bounds__(256) void wp_split_kernel(WpArgs a) {
  const int l = wp_layer_of(a, blockIdx.x);
  const WpLayer& L = a.L[l];
  int lb = blockIdx.x - a.first[l];
  const float scale = (HALF && L.amax) ? ldexpf(1.f, scale_exp_from_amax(*L.amax)) : 1.f;
#pragma unroll 1
  for (int dir = 0; dir < 2; ++dir) {
    void* out = dir ? L.ws_bwd : L.ws_fwd;
    const int lay = dir ? L.lay_b : L.lay_f;
    const bool s2d = lay == 3;
    const int taps = s2d ? 9 : L.taps;
    const int cin = s2d ? 4 * L.cin : L.cin;
    const int N = dir ? cin : L.cout, C = dir ? L.cout : cin;          // the operand is [taps][N][C]
    const float* src = dir ? (s2d ? L.w3 : L.w) : (s2d ? L.w3_t : L.w_t);
    const long long items = !out ? 0 : (lay == 0 ? (long long)taps * N * C / 8 : (long long)taps * (C >> 5) * (N >> 5) * 128);
    const int nb = (int)((items + 255) / 256);
    if (lb >= nb) { lb -= nb; continue; }
    const long long idx = (long long)lb * 256 + threadIdx.x;
    if (idx >= items) return;
    u32x4 pl[P];
    if (lay == 0) {
      split8<P, HALF>(reinterpret_cast<const f32x4*>(src)[2 * idx], reinterpret_cast<const f32x4*>(src)[2 * idx + 1], scale, pl);
#pragma unroll
      for (int pp = 0; pp < P; ++pp) reinterpret_cast<u32x4*>(out)[(long long)pp * items + idx] = pl[pp];
    } else if (lay == 4) {
      // fragments of v_mfma_f32_16x16x32 (conv_halo_pc.hip): [tap][C/32][N/16][plane][lane] x 16 B, lane l = column 16 t + (l & 15),
      // channels 32 c + 8 (l >> 4) .. + 7 (sgg_conv_split_weights_frag16)
      if constexpr (P == 2) {
        const int nch = C >> 5, ntl = N >> 4;
        const int lane = (int)(idx & 63);
        long long r = idx >> 6;
        const int ntile = (int)(r % ntl); r /= ntl;
        const int cc = (int)(r % nch);
        const int tap = (int)(r / nch);
        const int n = ntile * 16 + (lane & 15), k = cc * 32 + 8 * (lane >> 4);
        const float* s = src + ((size_t)tap * N + n) * C + k;
        split8<2, HALF>(*reinterpret_cast<const f32x4*>(s), *reinterpret_cast<const f32x4*>(s + 4), scale, pl);
        const size_t o = (((size_t)(tap * nch + cc) * ntl + ntile) * 2) * 64 + lane;
        reinterpret_cast<u32x4*>(out)[o] = pl[0];
        reinterpret_cast<u32x4*>(out)[o + 64] = pl[1];
      }
    } else if constexpr (P == 2) {
      const int nch = C >> 5, ntl = N >> 5;
      const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 1);
      long long r = idx >> 7;
      const int ntile = (int)(r % ntl); r /= ntl;
      const int cc = (int)(r % nch);
      const int tap = (int)(r / nch);
      const int n = ntile * 32 + (lane & 31), k = cc * 32 + ks * 16 + 8 * (lane >> 5);
      const float* s = src + ((size_t)tap * N + n) * C + k;
      split8<2, HALF>(*reinterpret_cast<const f32x4*>(s), *reinterpret_cast<const f32x4*>(s + 4), scale, pl);
      const size_t o = ((((size_t)(tap * nch + cc) * ntl + ntile) * 2 + ks) * 2) * 64 + lane;
      reinterpret_cast<u32x4*>(out)[o] = pl[0];
      reinterpret_cast<u32x4*>(out)[o + 64] = pl[1];
    }
    return;
  }
}

// ---- host ---------------------------------------------------------------------------------------------------
struct sgg_conv_weight_desc {      // mirrors include/sgg_hip.h
  const float* w;
  float* w_hwoi;
  float* w3;
  float* w3_hwoi;
  void* ws_fwd;
  void* ws_bwd;
  float* amax;
  int taps, cin, cout, layout_fwd, layout_bwd;
};

extern "C" int sgg_conv_prepare_weights(const sgg_conv_weight_desc* layers, int n, int precision, void* stream) {
  SGG_CHECK_ARG(layers && n > 0 && n <= SGG_WP_MAX, "sgg_conv_prepare_weights: 1..%d layers", SGG_WP_MAX);
  SGG_CHECK_ARG(precision == 0 || (precision >= 1 && precision <= 4) || precision == 6, "sgg_conv_prepare_weights: bad precision");
  const int pg = sgg_prec_general(precision);
  WpArgs a;
  a.nl = n;
  for (int i = 0; i < n; ++i) {
    const sgg_conv_weight_desc& d = layers[i];
    SGG_CHECK_ARG(d.w && d.w_hwoi && d.taps > 0 && d.cin > 0 && d.cout > 0, "sgg_conv_prepare_weights: layer %d: null pointer / bad dims", i);
    const bool s2d = d.layout_fwd == 3 || d.layout_bwd == 3;
    SGG_CHECK_ARG(!s2d || (d.w3 && d.w3_hwoi && d.taps == 25), "sgg_conv_prepare_weights: layer %d: layout 3 needs w3 / w3_hwoi and 25 taps", i);
    SGG_CHECK_ARG(!(d.ws_fwd || d.ws_bwd) || ((long long)d.taps * d.cin * d.cout) % 8 == 0, "sgg_conv_prepare_weights: layer %d: size", i);
    SGG_CHECK_ARG(!(d.ws_fwd || d.ws_bwd) || pg != 2 || d.amax, "sgg_conv_prepare_weights: layer %d: precision 1 / 2 need the amax word", i);
    SGG_CHECK_ARG(d.layout_fwd >= 0 && d.layout_fwd <= 4 && d.layout_bwd >= 0 && d.layout_bwd <= 4 &&
                      ((d.layout_fwd == 0 && d.layout_bwd == 0) || (d.cin % 32 == 0 && d.cout % 32 == 0)),
                  "sgg_conv_prepare_weights: layer %d: fragment layouts need channels %% 32 == 0", i);
    SGG_CHECK_ARG(pg != 6 || (d.layout_fwd == 0 && d.layout_bwd == 0), "sgg_conv_prepare_weights: precision 6 has the plane layout only");
    WpLayer& L = a.L[i];
    L.w = d.w; L.w_t = d.w_hwoi; L.w3 = d.w3; L.w3_t = d.w3_hwoi; L.ws_fwd = pg ? d.ws_fwd : nullptr; L.ws_bwd = pg ? d.ws_bwd : nullptr;
    L.amax = (pg == 2) ? d.amax : nullptr;
    L.taps = d.taps; L.cin = d.cin; L.cout = d.cout; L.lay_f = d.layout_fwd; L.lay_b = d.layout_bwd;
  }
  hipStream_t st = (hipStream_t)stream;
  if (pg == 2) hipLaunchKernelGGL(wp_zero_kernel, dim3(1), dim3(64), 0, st, a);
  int tot = 0;
  for (int i = 0; i < n; ++i) {
    const WpLayer& L = a.L[i];
    a.first[i] = tot;
    tot += sgg_cdiv(L.taps * sgg_cdiv(L.cin, 32) * sgg_cdiv(L.cout, 32), WP_TPB);
    if (L.lay_f == 3 || L.lay_b == 3) tot += (int)sgg_cdiv(36LL * L.cin * L.cout, 256);
  }
  a.first[n] = tot;
  hipLaunchKernelGGL(wp_transpose_kernel, dim3(tot), dim3(256), 0, st, a);
  SGG_LAUNCH_CHECK("sgg_conv_prepare_weights(transpose)");
  if (pg == 0) return SGG_OK;
  tot = 0;
  for (int i = 0; i < n; ++i) {
    const WpLayer& L = a.L[i];
    a.first[i] = tot;
    for (int dir = 0; dir < 2; ++dir) {
      const int lay = dir ? L.lay_b : L.lay_f;
      if (!(dir ? L.ws_bwd : L.ws_fwd)) continue;
      const long long taps = lay == 3 ? 9 : L.taps, cin = lay == 3 ? 4 * L.cin : L.cin;
      const long long N = dir ? cin : L.cout, C = dir ? L.cout : cin;
      const long long items = lay == 0 ? taps * N * C / 8 : taps * (C / 32) * (N / 32) * 128;
      tot += (int)sgg_cdiv(items, 256);
    }
  }
  a.first[n] = tot;
  if (tot == 0) return SGG_OK;
  if (pg == 2) hipLaunchKernelGGL((wp_split_kernel<2, true>), dim3(tot), dim3(256), 0, st, a);
  else if (pg == 3) hipLaunchKernelGGL((wp_split_kernel<2, false>), dim3(tot), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wp_split_kernel<3, false>), dim3(tot), dim3(256), 0, st, a);
  SGG_LAUNCH_CHECK("sgg_conv_prepare_weights(split)");
  return SGG_OK;
}
