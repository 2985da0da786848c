// Dual numbers a + eps*b (eps^2 = 0) for the second-order path of the WGAN gradient penalty
// (reference: tf.contrib.gan gan_loss(gradient_penalty_weight=LAMBDA, one_sided=True), train.py:245-250).
//
// d/dtheta [ lambda * GP(g(theta)) ],  g = d sum(D(x_hat)) / d x_hat,  equals the gradient w.r.t. theta of the
// directional derivative of sum(D) along v = dGP/dg held constant.  With every head kernel templated over
// the scalar type, the forward pass evaluated on duals (x + eps*x_dot) is that directional derivative (JVP),
// and the ordinary first-order backward evaluated on duals with cotangent (y_tilde + eps*y_bar) yields
//   real part = cotangent of the tangent input, dual part = cotangent of the primal input / parameter.
// So no hand-derived "backward of backward" formulas exist in this library.
#pragma once
#include "sgg_common.h"

struct Dual {
  float r, d;
};

__device__ __forceinline__ Dual mk(float r, float d) { return Dual{r, d}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return mk(a.r + b.r, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return mk(a.r - b.r, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return mk(-a.r, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return mk(a.r * b.r, a.r * b.d + a.d * b.r); }
__device__ __forceinline__ Dual operator+(Dual a, float b) { return mk(a.r + b, a.d); }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return mk(a.r - b, a.d); }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return mk(a - b.r, -b.d); }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return mk(a.r * b, a.d * b); }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return mk(a * b.r, a * b.d); }
__device__ __forceinline__ Dual& operator+=(Dual& a, Dual b) { a.r += b.r; a.d += b.d; return a; }

template <typename T> struct Sc;
template <> struct Sc<float> {
  static __device__ __forceinline__ float ld(const float* pr, const float*, size_t i) { return pr[i]; }
  static __device__ __forceinline__ void st(float* pr, float*, size_t i, float v) { pr[i] = v; }
  static __device__ __forceinline__ float zero() { return 0.f; }
  static __device__ __forceinline__ float re(float v) { return v; }
  // part that is the cotangent of a real (tangent-free) quantity: the value itself for floats
  static __device__ __forceinline__ float pcot(float v) { return v; }
  static __device__ __forceinline__ float lift(float v) { return v; }
};
template <> struct Sc<Dual> {
  static __device__ __forceinline__ Dual ld(const float* pr, const float* pd, size_t i) { return mk(pr[i], pd[i]); }
  static __device__ __forceinline__ void st(float* pr, float* pd, size_t i, Dual v) { pr[i] = v.r; pd[i] = v.d; }
  static __device__ __forceinline__ Dual zero() { return mk(0.f, 0.f); }
  static __device__ __forceinline__ float re(Dual v) { return v.r; }
  static __device__ __forceinline__ float pcot(Dual v) { return v.d; }
  static __device__ __forceinline__ Dual lift(float v) { return mk(v, 0.f); }
};

__device__ __forceinline__ float exp_(float x) { return expf(x); }
__device__ __forceinline__ Dual exp_(Dual x) { const float e = expf(x.r); return mk(e, e * x.d); }
__device__ __forceinline__ float tanh_(float x) { return tanhf(x); }
__device__ __forceinline__ Dual tanh_(Dual x) { const float t = tanhf(x.r); return mk(t, (1.f - t * t) * x.d); }
__device__ __forceinline__ float sigmoid_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ Dual sigmoid_(Dual x) { const float s = 1.f / (1.f + expf(-x.r)); return mk(s, s * (1.f - s) * x.d); }
__device__ __forceinline__ float rsqrt_(float x) { return 1.f / sqrtf(x); }
__device__ __forceinline__ Dual rsqrt_(Dual x) { const float r = 1.f / sqrtf(x.r); return mk(r, -0.5f * r * r * r * x.d); }
__device__ __forceinline__ float recip_(float x) { return 1.f / x; }
__device__ __forceinline__ Dual recip_(Dual x) { const float r = 1.f / x.r; return mk(r, -r * r * x.d); }

__device__ __forceinline__ Dual wave_sum(Dual v) { return mk(wave_sum(v.r), wave_sum(v.d)); }
