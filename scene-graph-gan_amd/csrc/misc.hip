// Loss, optimiser and utility kernels of the WGAN-GP step (gfx950).
//   wgan_gp_loss_{fwd,bwd}  tfgan wasserstein_gradient_penalty(one_sided=True, epsilon=1e-10), train.py:249-250 (Appendix A.7)
//   wgan_losses             wasserstein_{generator,discriminator}_loss, train.py:247-248 (Appendix A.6)
//   adam_tf_multi           tf.train.AdamOptimizer(1e-4, beta1=0.5, beta2=0.9), train.py:258-259 (Appendix A.8):
//                           theta -= lr_t * m / (sqrt(v) + eps), eps OUTSIDE the bias correction
//   argmax_rows             tf.argmax(x, -1), train.py:270-271: int64, first index on ties
#include "sgg_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

// ---- error / info (host) ----------------------------------------------------------------------------
static thread_local char g_err[512] = "";
extern "C" __attribute__((visibility("hidden"))) void sgg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* sgg_last_error(void) { return g_err; }
extern "C" int sgg_version(void) { return 100; }  // 0.1.0
extern "C" int sgg_device_info(int* cu_count, size_t* lds_bytes_per_cu, size_t* hbm_bytes, char* arch, int arch_len) {
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    sgg_set_error("sgg_device_info: no HIP device");
    return SGG_ERR_LAUNCH;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = prop.maxSharedMemoryPerMultiProcessor;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return SGG_OK;
}

// ---- kernels ----------------------------------------------------------------------------------------
__global__ void fill_kernel(float* __restrict__ p, long long n, float v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
// 16-byte stores (p 16-byte aligned, n4 = n / 4): the gradient arenas (138 MB, zeroed at the start of every update on the chain of
// that update's first encoder forward) took 86 us with 4-byte stores
__global__ void fill4_kernel(f32x4* __restrict__ p, long long n4, float v) {
  const f32x4 vv = f32x4{v, v, v, v};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) p[i] = vv;
}

// out[b, :] = real[b, :] + alpha[b] * (fake[b, :] - real[b, :])
__global__ void interpolate_kernel(const float* __restrict__ real, const float* __restrict__ fake, const float* __restrict__ alpha,
                                   float* __restrict__ out, int n) {
  const int b = blockIdx.y;
  const float a = alpha[b];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const size_t e = (size_t)b * n + i;
    const float r = real[e];
    out[e] = r + a * (fake[e] - r);
  }
}

__global__ void onehot_kernel(const long long* __restrict__ labels, float* __restrict__ out, int V) {
  const int row = blockIdx.x;
  const long long lab = labels[row];
  for (int i = threadIdx.x; i < V; i += blockDim.x) out[(size_t)row * V + i] = (i == lab) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void gp_fwd_kernel(const float* __restrict__ g, float* __restrict__ slopes, float* __restrict__ pen,
                                                     int n) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = g[(size_t)b * n + i];
    s += v * v;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    const float sl = sqrtf(s + 1e-10f);
    slopes[b] = sl;
    pen[b] = fmaxf(sl - 1.f, 0.f);
  }
}
// v = scale * (2/B) * pen_b / slopes_b * g      (= scale * dGP/dg)
__global__ void gp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ slopes, const float* __restrict__ pen,
                              float* __restrict__ v, int B, int n, float scale) {
  const int b = blockIdx.y;
  const float coef = scale * (2.f / (float)B) * pen[b] / slopes[b];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    v[(size_t)b * n + i] = coef * g[(size_t)b * n + i];
}

// d_out rows: [0,B) fake, [B,2B) real (optional), T outputs per row. out[0]=disc_cost, out[1]=wdist, out[2]=gp, out[3]=mean_fake
__global__ __launch_bounds__(256) void wgan_losses_kernel(const float* __restrict__ d_out, const float* __restrict__ pen, float lam,
                                                          int B, int T, int has_real, float* __restrict__ out) {
  __shared__ float red[4];
  float sf = 0.f, sr = 0.f, sp = 0.f;
  for (int i = threadIdx.x; i < B * T; i += 256) {
    sf += d_out[i];
    if (has_real) sr += d_out[B * T + i];
  }
  if (pen)
    for (int i = threadIdx.x; i < B; i += 256) sp += pen[i] * pen[i];
  sf = block_sum_256(sf, red);
  sr = block_sum_256(sr, red);
  sp = block_sum_256(sp, red);
  if (threadIdx.x == 0) {
    const float mf = sf / (float)(B * T), mr = sr / (float)(B * T), gp = sp / (float)B;
    out[0] = (mf - mr) + lam * gp;
    out[1] = mf - mr;
    out[2] = gp;
    out[3] = mf;
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long n, float lr_t, float b1, float b2, float eps, float gscale) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * gscale;
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], pv = reinterpret_cast<f32x4*>(p)[i];
    mv = mv * b1 + gv * (1.f - b1);
    vv = vv * b2 + gv * gv * (1.f - b2);
#pragma unroll
    for (int q = 0; q < 4; ++q) pv[q] -= lr_t * mv[q] / (sqrtf(vv[q]) + eps);
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float gv = g[i] * gscale;
    const float mv = m[i] * b1 + gv * (1.f - b1);
    const float vv = v[i] * b2 + gv * gv * (1.f - b2);
    m[i] = mv; v[i] = vv;
    p[i] -= lr_t * mv / (sqrtf(vv) + eps);
  }
}

// one wave per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long long* __restrict__ out, int rows, int V,
                                                          int ld) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < V; i += 64) {
    const float v = x[(size_t)row * ld + i];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[row] = (long long)bi;
}

// amax[0] = max(amax[0], max |x|)   (caller zeroes amax first); used to scale tensors for the f16x3 conv mode
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, float* __restrict__ amax) {
  __shared__ float red[4];
  float m = 0.f;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomic_amax(amax, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// tf.matmul(indices, W) for one-hot rows (discriminator_with_attention.py:86-87; real triples are one-hot floats, train.py:173):
// the product degenerates to a row gather.  A label outside [0, V) gives a zero row, as tf.one_hot does.
__global__ void embed_gather_fwd_kernel(const long long* __restrict__ labels, int lstride, const float* __restrict__ W, int V, int E,
                                        float* __restrict__ out, int ldo) {
  const int r = blockIdx.x;
  const long long lab = labels[(size_t)r * lstride];
  const bool ok = lab >= 0 && lab < V;
  for (int i = threadIdx.x; i < E; i += blockDim.x) out[(size_t)r * ldo + i] = ok ? W[(size_t)lab * E + i] : 0.f;
}

// dW[labels[r], :] += dY[r, :].  Rows that share a label are summed in row order by the workgroup of the FIRST such row:
// deterministic, no atomics (R is a few hundred at most).
__global__ void embed_gather_bwd_kernel(const long long* __restrict__ labels, int lstride, const float* __restrict__ dY, int lddy,
                                        float* __restrict__ dW, int V, int E, int R) {
  extern __shared__ long long lab_s[];
  for (int i = threadIdx.x; i < R; i += blockDim.x) lab_s[i] = labels[(size_t)i * lstride];
  __syncthreads();
  const int r = blockIdx.x;
  const long long lab = lab_s[r];
  if (lab < 0 || lab >= V) return;
  for (int q = 0; q < r; ++q)
    if (lab_s[q] == lab) return;            // an earlier row owns this label
  for (int i = threadIdx.x; i < E; i += blockDim.x) {
    float acc = dW[(size_t)lab * E + i];
    for (int q = r; q < R; ++q)
      if (lab_s[q] == lab) acc += dY[(size_t)q * lddy + i];
    dW[(size_t)lab * E + i] = acc;
  }
}

// ---- C ABI ------------------------------------------------------------------------------------------
static inline int grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int sgg_absmax(const float* x, long long n, float* amax, void* stream) {
  SGG_CHECK_ARG(x && amax && n > 0 && (((uintptr_t)x) & 15) == 0, "sgg_absmax: bad argument");
  hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n / 4 + 1, 256) > 1024 ? 1024 : grid_for(n / 4 + 1, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, n, amax);
  SGG_LAUNCH_CHECK("sgg_absmax");
  return SGG_OK;
}

extern "C" int sgg_fill(float* p, long long n, float value, void* stream) {
  SGG_CHECK_ARG(p && n >= 0, "sgg_fill: bad argument");
  if (n == 0) return SGG_OK;
  if (n >= 4096 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const long long n4 = n / 4;
    hipLaunchKernelGGL(fill4_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<f32x4*>(p), n4, value);
    if (n & 3) hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p + 4 * n4, n & 3, value);
  } else {
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, n, value);
  }
  SGG_LAUNCH_CHECK("sgg_fill");
  return SGG_OK;
}

extern "C" int sgg_interpolate(const float* real, const float* fake, const float* alpha, float* out, int B, int n, void* stream) {
  SGG_CHECK_ARG(real && fake && alpha && out && B > 0 && n > 0, "sgg_interpolate: bad argument");
  hipLaunchKernelGGL(interpolate_kernel, dim3(grid_for(n, 256) > 64 ? 64 : grid_for(n, 256), B), dim3(256), 0, (hipStream_t)stream,
                     real, fake, alpha, out, n);
  SGG_LAUNCH_CHECK("sgg_interpolate");
  return SGG_OK;
}

// tf.image.resize_images(image, [oh, ow]) of TF 1.x (bilinear, align_corners=False: the LEGACY grid src = dst * in/out, no half-pixel
// offset, no antialiasing) followed by (x - mean) / std: train.py:171-172.  One launch per batch: image b is RGB uint8 [H_b][W_b][3]
// at src + offsets[b] (variable sizes, packed by the host loader); dst [B][oh][ow][3] fp32.  Arithmetic order as TF's kernel
// (top / bottom row interpolated in x, then in y; every product rounded before it is added: no fused multiply-add, see R below).
__global__ void resize_bilinear_tf1_kernel(const unsigned char* __restrict__ src, const long long* __restrict__ offsets,
                                           const int* __restrict__ heights, const int* __restrict__ widths, float* __restrict__ dst,
                                           int oh, int ow, const float* __restrict__ means, const float* __restrict__ stds) {
  // TF's CPU kernel rounds every product and difference separately.  The library is built with -ffp-contract=fast, which fuses in
  // the backend whatever the source says (__fmul_rn / __fsub_rn and `#pragma clang fp contract(off)` included: fx - x0 became
  // fma(x, sx, -x0), up to 5e-4 on the 0..255 scale against the oracle), so every product passes through an opaque register
  // copy (R) before it is added or subtracted.
#define R(x) ({ float r__ = (x); asm volatile("" : "+v"(r__)); r__; })
  const int b = blockIdx.y, y = blockIdx.x;
  const int H = heights[b], W = widths[b];
  const unsigned char* im = src + offsets[b];
  const float sy = (float)H / (float)oh, sx = (float)W / (float)ow;
  const float fy = R((float)y * sy);
  const int y0 = (int)floorf(fy), y1 = min(y0 + 1, H - 1);
  const float wy = fy - (float)y0;
  for (int i = threadIdx.x; i < ow * 3; i += blockDim.x) {
    const int x = i / 3, c = i - 3 * x;
    const float fx = R((float)x * sx);
    const int x0 = (int)floorf(fx), x1 = min(x0 + 1, W - 1);
    const float wx = fx - (float)x0;
    const float tl = (float)im[((size_t)y0 * W + x0) * 3 + c], tr = (float)im[((size_t)y0 * W + x1) * 3 + c];
    const float bl = (float)im[((size_t)y1 * W + x0) * 3 + c], br = (float)im[((size_t)y1 * W + x1) * 3 + c];
    const float dt = tr - tl, db = br - bl;
    const float pt = R(dt * wx), pb = R(db * wx);
    const float top = tl + pt, bot = bl + pb;
    const float dv = bot - top;
    const float pv = R(dv * wy);
    const float v = top + pv;
    const float num = v - means[c];
    dst[(((size_t)b * oh + y) * ow + x) * 3 + c] = num / stds[c];
  }
#undef R
}

extern "C" int sgg_onehot(const long long* labels, float* out, int rows, int V, void* stream) {
  SGG_CHECK_ARG(labels && out && rows > 0 && V > 0, "sgg_onehot: bad argument");
  hipLaunchKernelGGL(onehot_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, labels, out, V);
  SGG_LAUNCH_CHECK("sgg_onehot");
  return SGG_OK;
}

extern "C" int sgg_embed_gather_fwd(const long long* labels, int label_stride, const float* W, int V, int E, float* out, int ldo,
                                    int R, void* stream) {
  SGG_CHECK_ARG(labels && W && out && R > 0 && V > 0 && E > 0 && label_stride >= 1 && ldo >= E, "sgg_embed_gather_fwd: bad argument");
  hipLaunchKernelGGL(embed_gather_fwd_kernel, dim3(R), dim3(128), 0, (hipStream_t)stream, labels, label_stride, W, V, E, out, ldo);
  SGG_LAUNCH_CHECK("sgg_embed_gather_fwd");
  return SGG_OK;
}

extern "C" int sgg_embed_gather_bwd(const long long* labels, int label_stride, const float* dY, int lddy, float* dW, int V, int E,
                                    int R, void* stream) {
  SGG_CHECK_ARG(labels && dY && dW && R > 0 && V > 0 && E > 0 && label_stride >= 1 && lddy >= E, "sgg_embed_gather_bwd: bad argument");
  SGG_CHECK_ARG(R <= 4096, "sgg_embed_gather_bwd: at most 4096 rows per call (got %d)", R);
  hipLaunchKernelGGL(embed_gather_bwd_kernel, dim3(R), dim3(128), (size_t)R * sizeof(long long), (hipStream_t)stream, labels,
                     label_stride, dY, lddy, dW, V, E, R);
  SGG_LAUNCH_CHECK("sgg_embed_gather_bwd");
  return SGG_OK;
}

extern "C" int sgg_resize_bilinear_tf1(const unsigned char* src, const long long* offsets, const int* heights, const int* widths,
                                       float* dst, int B, int out_h, int out_w, const float* means, const float* stds, void* stream) {
  SGG_CHECK_ARG(src && offsets && heights && widths && dst && means && stds && B > 0 && out_h > 0 && out_w > 0,
                "sgg_resize_bilinear_tf1: bad argument");
  hipLaunchKernelGGL(resize_bilinear_tf1_kernel, dim3(out_h, B), dim3(256), 0, (hipStream_t)stream, src, offsets, heights, widths, dst,
                     out_h, out_w, means, stds);
  SGG_LAUNCH_CHECK("sgg_resize_bilinear_tf1");
  return SGG_OK;
}

extern "C" int sgg_wgan_gp_loss_fwd(const float* g, float* slopes, float* pen, int B, int n, void* stream) {
  SGG_CHECK_ARG(g && slopes && pen && B > 0 && n > 0, "sgg_wgan_gp_loss_fwd: bad argument");
  hipLaunchKernelGGL(gp_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, g, slopes, pen, n);
  SGG_LAUNCH_CHECK("sgg_wgan_gp_loss_fwd");
  return SGG_OK;
}

extern "C" int sgg_wgan_gp_loss_bwd(const float* g, const float* slopes, const float* pen, float* v, int B, int n, float scale,
                                    void* stream) {
  SGG_CHECK_ARG(g && slopes && pen && v && B > 0 && n > 0, "sgg_wgan_gp_loss_bwd: bad argument");
  hipLaunchKernelGGL(gp_bwd_kernel, dim3(grid_for(n, 256) > 64 ? 64 : grid_for(n, 256), B), dim3(256), 0, (hipStream_t)stream, g,
                     slopes, pen, v, B, n, scale);
  SGG_LAUNCH_CHECK("sgg_wgan_gp_loss_bwd");
  return SGG_OK;
}

extern "C" int sgg_wgan_losses(const float* d_out, const float* pen, float lam, int B, int T, int has_real, float* out4,
                               void* stream) {
  SGG_CHECK_ARG(d_out && out4 && B > 0 && T > 0, "sgg_wgan_losses: bad argument");
  hipLaunchKernelGGL(wgan_losses_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_out, pen, lam, B, T, has_real, out4);
  SGG_LAUNCH_CHECK("sgg_wgan_losses");
  return SGG_OK;
}

// "multi-tensor": every trainable tensor of a network lives in one arena, so one flat range covers them all.
// grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
extern "C" int sgg_adam_tf_multi(float* params, const float* grads, float* m, float* v, long long n, float lr_t, float beta1,
                                 float beta2, float eps, float grad_scale, void* stream) {
  SGG_CHECK_ARG(params && grads && m && v && n > 0, "sgg_adam_tf_multi: bad argument");
  SGG_CHECK_ARG((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                "sgg_adam_tf_multi: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, lr_t,
                     beta1, beta2, eps, grad_scale);
  SGG_LAUNCH_CHECK("sgg_adam_tf_multi");
  return SGG_OK;
}

extern "C" int sgg_argmax_rows(const float* x, long long* out, int rows, int V, int ld, void* stream) {
  SGG_CHECK_ARG(x && out && rows > 0 && V > 0 && ld >= V, "sgg_argmax_rows: bad argument");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(sgg_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, out, rows, V, ld);
  SGG_LAUNCH_CHECK("sgg_argmax_rows");
  return SGG_OK;
}
