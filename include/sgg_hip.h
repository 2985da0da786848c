/* libsgg_hip.so — C ABI of the MI355X-native (gfx950) Scene-Graph-GAN training hot path.
 *
 * The reference (mklawonn/Scene-Graph-GAN, Python 2 + TensorFlow 1.x) has NO FFI / plugin boundary: every
 * arithmetic op is a stock TensorFlow kernel composed by architectures/*.py and train.py.  This header is the
 * boundary a maintainer would bind instead of those TF kernels; each entry point cites the reference call
 * site (file:line, relative to the reference root) whose TF op (and its autodiff gradient under
 * optimizer.minimize, train.py:265-266) it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (sgg_last_error() gives the text); nothing throws,
 *     allocates, or synchronises; work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - all pointers are DEVICE pointers to fp32 unless stated; tensors are dense row-major; activations are
 *     NHWC, conv kernels HWIO [kh][kw][cin][cout] (TF layouts), dense kernels [in][out].
 *   - `*_workspace_bytes` functions size the scratch buffer the caller must pass.
 *   - "dual" pointer pairs: the second-order path of the gradient penalty runs the same kernels on dual
 *     numbers (real plane, dual plane).  Passing NULL for the *_dual inputs selects plain fp32.
 */
#ifndef SGG_HIP_H
#define SGG_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int sgg_version(void);
const char* sgg_last_error(void);
int sgg_device_info(int* cu_count, size_t* lds_bytes_per_cu, size_t* hbm_bytes, char* arch, int arch_len);

/* ---- convolutional encoder: tf.layers.conv2d(padding="same") --------------------------------------------
 * generator_with_attention.py:29-68, discriminator_with_attention.py:29-68 (12 live layers each).
 * pad_t / pad_l are the TF SAME "before" pads (pad_total // 2); the "after" pad is implied by Ho/Wo. */
int sgg_hwio_to_hwoi(const float* w_hwio, float* w_hwoi, int taps, int cin, int cout, void* stream);
/* forward: `w` = HWIO kernel when Cin == 3, else its HWOI transpose (sgg_hwio_to_hwoi). y = conv(x) + bias */
/* precision: 0 = native f32 MFMA (v_mfma_f32_32x32x2_f32, exact f32);
 *            2 = f32 operands scaled by a per-tensor power of two (from amax_* = device words holding max|tensor|, see
 *                sgg_absmax / the amax_out of the LayerNorm kernels) and split into two fp16 pieces, 3 fp16 MFMAs, f32
 *                accumulate: the two pieces are taken with round-to-nearest-even (v_cvt_pk_f16_f32; csrc/sgg_common.h f16_split2),
 *                i.e. 23 significant bits per operand for elements within 2^-16 of the tensor maximum, min(23, 39 - d) bits for an
 *                element 2^-d below it (DESIGN.md "Error model and its floor"); error vs fp64 within 2x native f32 + 3e-7 on every
 *                configs[1] layer (tests/test_fullsize_conv_gpu.py);
 *            3 / 6 = split into 2 / 3 bf16 pieces, 3 / 6 bf16 MFMAs (6: f32-equivalent, 3: 2^-17 cross terms dropped);
 *            1 / 4 = MIXED PRECISION (not the reference's arithmetic): operands rounded to ONE fp16 (scaled
 *                like precision 2) / ONE bf16 piece, one MFMA per product, f32 accumulate - error vs fp64 2e-3 / 1.5e-2 of the
 *                output's maximum.  Served by the resident kernels (w_split_layout 1 / 2 / 3, the halo-resident wgrad); other
 *                shapes run these codes in the two-piece arithmetic (2 / 3).  The pre-split weights are the ones of precision 2 / 3.
 * amax pointers are only read for precision 1 / 2 (may be NULL otherwise). */
/* w_split (optional, may be NULL): the same weights pre-split by sgg_split_bf16 into precision/2 bf16 planes [P][n];
 * saves the per-workgroup split of the weight operand in the split-bf16 modes. */
int sgg_conv_split_weights(const float* in, void* out, long long n, int precision, const float* amax, void* stream);
int sgg_absmax(const float* x, long long n, float* amax /* atomically max-ed; zero it first */, void* stream);
/* Halo-resident kernel for the 3x3 stride-1 layers (generator_with_attention.py:31-57: conv1_2, conv2_1..2_4, conv3_1,
 * conv3_2) in precision 2 / 3: sgg_conv_wsplit_layout returns 1 where it applies (H % 8 == W % 8 == 0); the pre-split weights
 * must then be in MFMA fragment order (sgg_conv_split_weights_frag over the [taps][N][C] tensor: the HWOI transpose for the
 * forward, the HWIO kernel for dgrad) and w_split_layout = 1 is passed to sgg_conv2d_nhwc_fwd / _dgrad.  Layout 0 = planes.
 * Band-resident kernel for the 5x5 stride-2 layers (generator_with_attention.py:35,50,65,68: conv2_5, conv3_5, `downsampled`;
 * conv1_3 has too few channels for its 128-column tile) on even grids: sgg_conv_wsplit_layout returns 2 (H, W = the full-resolution
 * grid; ask with (Cin, Cout) swapped for the dgrad direction); weights in the same fragment order with taps = 25.
 * conv1_3 (generator_with_attention.py:35: 5x5 stride 2 over 32 -> 32 channels, H % 16 == W % 16 == 0): sgg_conv_wsplit_layout
 * returns 3 - the convolution runs on the halo-resident kernel as a 3x3 stride-1 convolution over the space-to-depth view of x
 * (x[2a + qy][2c + qx][ci] = channel (qy, qx, ci) of pixel (a, c); no copy, strides only) with the 9-tap kernel
 * [3][3][4 * Cin][Cout] of sgg_conv_s2d_weights (11 of its 36 (tap, parity) slots are zero); weights in fragment order with
 * taps = 9 (forward: the HWOI transpose of that kernel, N = Cout, C = 4 * Cin; dgrad: the kernel itself, N = 4 * Cin, C = Cout).
 * Producer / consumer kernel for the 3x3 stride-1 layers with 128-column tiles (generator_with_attention.py:44-57: conv2_3,
 * conv2_4, conv3_1, conv3_2 forward; conv2_4, conv3_1, conv3_2 dgrad): where a layout-1 layer also has Cout % 128 == 0 and
 * Cin % 64 == 0 in precision 2 / 3, sgg_conv_wsplit_layout returns 4 instead - one 8-wave workgroup per CU (4 MFMA waves on
 * v_mfma_f32_16x16x32_*, 4 producer waves that each DMA a quarter of every tap's weight fragments and stage a quarter of the patch; csrc/conv_halo_pc.hip); the pre-split weights are then the
 * fragments of THAT MFMA shape (sgg_conv_split_weights_frag16) and w_split_layout = 4 is passed to sgg_conv2d_nhwc_fwd / _dgrad.
 * Same arithmetic as layout 1 (same pieces, same three products per f32 product, f32 accumulation; the order of the sum over the
 * 32 channels of a chunk differs). */
int sgg_conv_s2d_weights(const float* w5 /* [5][5][Cin][Cout] */, float* w3 /* [3][3][4*Cin][Cout] */, int Cin, int Cout, void* stream);
/* All of the above for every layer of one encoder after an optimiser step, in three launches (the per-layer entry points are
 * ~45 five-microsecond launches per network): w_hwoi = the HWOI transpose; *amax = max|w| (precision 1 / 2; may be NULL
 * otherwise); layout 3 (in either direction): w3 / w3_hwoi = the 9-tap space-to-depth kernel and its transpose; ws_fwd / ws_bwd
 * (optional) = the pre-split operands of the forward / dgrad direction in layout_fwd / layout_bwd (sgg_conv_wsplit_layout; 0 =
 * planes of sgg_conv_split_weights).  At most 16 layers per call; Cin == 3 layers need no preparation. */
typedef struct {
  const float* w;        /* HWIO [taps][Cin][Cout] */
  float* w_hwoi;         /* [taps][Cout][Cin] */
  float* w3;             /* layout 3 only, else NULL: [9][4*Cin][Cout] */
  float* w3_hwoi;        /* layout 3 only, else NULL: [9][Cout][4*Cin] */
  void* ws_fwd;          /* may be NULL */
  void* ws_bwd;          /* may be NULL */
  float* amax;
  int taps, cin, cout, layout_fwd, layout_bwd;
} sgg_conv_weight_desc;
int sgg_conv_prepare_weights(const sgg_conv_weight_desc* layers, int n_layers, int precision, void* stream);
/* The library reads no environment variables and keeps no mutable global state: every kernel choice is a function of the
 * arguments (w_split_layout here, `algo` of sgg_conv2d_nhwc_wgrad). */
int sgg_conv_wsplit_layout(int KH, int KW, int stride, int H, int W, int Cin, int Cout, int precision);
/* The same question for a launch whose source operand (x of the forward, dy of the dgrad) WILL arrive pre-split (operand_format 1):
 * 3x3 stride-1 layers with Cout % 64 == 0 (not 128) and Cin % 64 == 0 in precision 2 then also run on the producer / consumer kernel,
 * in its four-block form (4 blocks x 64 columns per workgroup; the patch by LDS-DMA only): 4 is returned for them as well, and such a
 * launch MUST pass operand_format 1 (an f32 source with this layout is refused).  generator_with_attention.py:41,44: the
 * Conv2DBackpropInput of conv2_2 and conv2_3. */
int sgg_conv_wsplit_layout_presplit(int KH, int KW, int stride, int H, int W, int Cin, int Cout, int precision);
int sgg_conv_split_weights_frag(const float* in, void* out, int taps, int N, int C, int precision, const float* amax, void* stream);
int sgg_conv_split_weights_frag16(const float* in, void* out, int taps, int N, int C, int precision, const float* amax, void* stream);
int sgg_conv2d_nhwc_fwd(const float* x, const float* w, const void* w_split, const float* bias, float* y, int B, int Hi, int Wi, int Cin,
                        int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int precision, int w_split_layout,
                        const float* amax_x, const float* amax_w, float* tile_stats, const float* ln_stats, const float* ln_gamma,
                        const float* ln_beta, int operand_format, void* stream);
/* operand_format of sgg_conv2d_nhwc_fwd, bits 8 .. 13 (launch hint, 0 = none): the persistent resident kernels (w_split_layout
 * 1 .. 4) occupy at most that many of an XCD's 32 CUs instead of all of them - for a forward pass that runs on its own HIP stream
 * beside another stream's chain of short, latency-critical launches (the recurrent heads), which then find free CUs at once.  The
 * work decomposition (tiles, products, summation orders) does not depend on it: results are bit-identical.
 * operand_format bit 0 (sgg_conv2d_nhwc_fwd / _dgrad: 0 or 1; _wgrad: bit 0 = x, bit 1 = dy): 1 = the operand is a PRE-SPLIT tensor as
 * sgg_layernorm_hwc_elu_fwd / _bwd write it with out_format 1 - same shape and bytes as the f32 tensor, every aligned group of 32
 * channels (128 B) holding the 32 leading fp16 pieces (64 B) then the 32 residual pieces of x * 2^e, e from the tensor's amax word
 * (which must be the word the producer used).  The kernel then stages the operand without splitting it again (bit-identical
 * products).  Precision 1 / 2 and the resident kernels only (w_split_layout 1 .. 4; wgrad: where sgg_conv2d_nhwc_wgrad_resident
 * says 1); not together with an LN prologue on that operand.
 * tile_stats (optional): the epilogue also writes (count, mean, M2, max |y - mean|) of every output tile, [B][n][4] with
 * n = sgg_conv2d_nhwc_fwd_tile_stats(...) > 0; pass them to sgg_layernorm_hwc_elu_fwd to skip its statistics pass, or to
 * sgg_layernorm_hwc_finalize when the consumer applies the LayerNorm itself:
 * LN prologue (ln_stats / ln_gamma / ln_beta, optional, w_split_layout 1 only; also on sgg_conv2d_nhwc_wgrad where its halo-resident
 * kernel applies): x is the PRE-LayerNorm output y of the producing convolution and the kernel computes
 * ELU((y - mean_b) * rstd_b * gamma_c + beta_c) while it stages its input patch (tf.contrib.layers.layer_norm(activation_fn=elu),
 * generator_with_attention.py:30..56), so the LayerNorm apply pass and the activation tensor are never written.  ln_stats [B][2] =
 * (mean, rstd) from sgg_layernorm_hwc_finalize; amax_x must then be the word that call published (an upper bound of max|a|). */
int sgg_conv2d_nhwc_fwd_tile_stats(int Ho, int Wo, int Cin, int Cout, int KH, int KW, int stride, int precision, int w_split_layout);
/* Conv2DBackpropInput: dx from dy and the HWIO kernel */
int sgg_conv2d_nhwc_dgrad(const float* dy, const float* w_hwio, const void* w_split, float* dx, int B, int Hi, int Wi, int Cin, int Ho,
                          int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int precision, int w_split_layout,
                          const float* amax_dy, const float* amax_w, int operand_format, void* stream);
/* Conv2DBackpropFilter: dw (HWIO) from x and dy.  algo: 0 = automatic (halo-resident kernel where it applies), 1 = per-tap
 * kernels only (A/B measurements). */
size_t sgg_conv2d_nhwc_wgrad_workspace_bytes(int B, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int KH, int KW);
int sgg_conv2d_nhwc_wgrad(const float* x, const float* dy, float* dw_hwio, int B, int Hi, int Wi, int Cin, int Ho,
                          int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int precision, int algo,
                          const float* amax_x, const float* amax_dy, const float* ln_stats, const float* ln_gamma, const float* ln_beta,
                          int operand_format, void* workspace, size_t workspace_bytes, void* stream);
/* 0: the per-tap kernels serve this shape; 1: the halo-resident kernel does with algo 0 (it takes pre-split operands and stages them
 * through registers, without arithmetic); 2: as 1, and when BOTH operands are pre-split (operand_format 3, precision 2, no LN prologue)
 * the LDS-DMA kernel runs instead: 64 x 128 / 64 x 64 channel tiles, both operands copied HBM -> LDS by `buffer_load ... lds`, no
 * staging registers or arithmetic (csrc/conv_wgrad_dma.hip; channels % 64 == 0, 8-divisible grids or row bands of <= 112 pixels) */
int sgg_conv2d_nhwc_wgrad_resident(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW, int stride, int precision);

/* ---- tf.contrib.layers.layer_norm(activation_fn=tf.nn.elu) over (H,W,C) per sample ------------------------
 * generator_with_attention.py:30..66 / discriminator_with_attention.py:30..66.  C: power of two in [4,1024].
 * fwd writes a = ELU(LN(y)) and stats[b] = (mean, rstd).  bwd writes dy, dgamma, dbeta and (optional, may be
 * NULL) dbias_prev = sum_{b,h,w} dy = the BiasAddGrad of the convolution that produced y. */
size_t sgg_layernorm_hwc_elu_workspace_bytes(int B, int HW, int C);
/* amax_out (optional): device word atomically max-ed with max|output| (the consumer convolution's f16 scaling).
 * Valid region (fwd and bwd): W == 0 -> the whole HW plane is normalised.  W > 0 -> the plane is an (HW / W) x W canvas that
 * holds the layer's true (Hv x Wv) map at rows [y0, y0 + Hv), columns [x0, x0 + Wv) (odd image sizes such as the reference's
 * 221 x 221, train.py:171, run on even canvases so that the tiled convolution kernels apply): statistics, gradients and
 * parameter gradients cover the region only; a / dy are written as ZEROS outside it (the next convolution's zero padding).
 * tile_stats must be NULL with W > 0. */
int sgg_layernorm_hwc_elu_fwd(const float* y, const float* gamma, const float* beta, float* a, float* stats,
                              float* amax_out, const float* tile_stats, int n_tile_stats, int B, int HW, int C,
                              int W, int y0, int x0, int Hv, int Wv, int out_format, void* workspace, size_t workspace_bytes, void* stream);
/* out_format 1 (both directions): the output tensor (a / dy) is written PRE-SPLIT for the convolutions that consume it (see
 * operand_format of sgg_conv2d_nhwc_fwd): every aligned 32-channel group as 32 leading + 32 residual fp16 pieces of x * 2^e.  The
 * scale must be fixed before the tensor is written, so *amax_out (required, C % 32 == 0) then receives an upper BOUND of max|x|
 * instead of the maximum: forward - the caller zeroes the word, the call merges the statistics, publishes max over the samples of
 * max|gamma| * max|y - mean| * rstd + max|beta| and applies (three launches; two with tile_stats); backward - max(rstd * max|dxhat|) *
 * (2 + max|xhat|), the two maxima published atomically by the reduction pass into the caller-zeroed words pq. */
/* An f32 NHWC tensor (n elements, channel count a multiple of 32) -> the PRE-SPLIT format under the scale of *amax (its maximum from
 * sgg_absmax, or any upper bound): for operands that no LayerNorm kernel produces - the gradient the attention head hands to
 * `downsampled` (generator_with_attention.py:68,74), whose Conv2DBackpropInput / Conv2DBackpropFilter then stage it by LDS-DMA like
 * every other layer's.  out may be x (in place). */
int sgg_presplit16(const float* x, float* out, long long n, const float* amax, void* stream);
/* statistics only: stats[b] = (mean, rstd) merged from the convolution's tile partials; amax_out max-ed with an upper bound of
 * max|ELU(LN(y))| (for the fp16 scaling of a consumer with an LN prologue) */
int sgg_layernorm_hwc_finalize(const float* tile_stats, int n_tile_stats, const float* gamma, const float* beta, float* stats,
                               float* amax_out, int B, int HW, int C, void* stream);
int sgg_layernorm_hwc_elu_bwd(const float* y, const float* da, const float* gamma, const float* beta,
                              const float* stats, float* dy, float* dgamma, float* dbeta, float* dbias_prev,
                              float* amax_out, int B, int HW, int C, int W, int y0, int x0, int Hv, int Wv, int out_format,
                              float* pq /* out_format 1: two words, zeroed by the caller, for the maxima the bound is made of */,
                              void* workspace, size_t workspace_bytes, void* stream);

/* The REDUCTION half of sgg_layernorm_hwc_elu_bwd without its apply pass, for a consumer that computes dy itself
 * (sgg_conv2d_nhwc_wgrad_c3_ln below): the partial sums go to `workspace` exactly as sgg_layernorm_hwc_elu_bwd leaves them (dgamma /
 * dbeta / dbias_prev: all NULL and sgg_layernorm_hwc_bwd_finalize later, or all given), and means [B][2] receives the two per-sample
 * means of the backward, (mean dxhat, mean dxhat * xhat), bit-identical to the values the apply pass derives.  Whole planes only. */
int sgg_layernorm_hwc_elu_bwd_sums(const float* y, const float* da, const float* gamma, const float* beta, const float* stats,
                                   float* means, float* dgamma, float* dbeta, float* dbias_prev, int B, int HW, int C, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* conv1_1's filter gradient FUSED with the apply half of the LayerNorm backward of its output (generator_with_attention.py:29-30
 * under optimizer.minimize, train.py:265-266): dW [3][3][3][32] = Conv2DBackpropFilter(x, dy) with
 *   dy = rstd * (da * ELU'(n) * gamma - m1 - xhat * m2),  xhat = (y - mean) * rstd,  n = xhat * gamma + beta
 * computed from (y, da) [B][H][W][32] inside the kernel - conv1_1's filter gradient is the only consumer of that dy (no image
 * gradient is taken, the bias gradient comes from the LayerNorm reductions), so it is never written: one read of y and da replaces
 * the apply pass (read y, read da, write dy) and the filter gradient's read of dy.  stats [B][2] = (mean, rstd) of the forward,
 * means [B][2] from sgg_layernorm_hwc_elu_bwd_sums; exact f32 MFMA arithmetic as sgg_conv2d_nhwc_wgrad with Cin == 3; workspace as
 * sgg_conv2d_nhwc_wgrad_workspace_bytes(B, H, W, 3, H, W, 32, 3, 3). */
int sgg_conv2d_nhwc_wgrad_c3_ln(const float* x, const float* y, const float* da, const float* gamma, const float* beta,
                                const float* stats, const float* means, float* dw, int B, int H, int W, int pad_t, int pad_l,
                                void* workspace, size_t workspace_bytes, void* stream);

/* dgamma == dbeta == NULL in sgg_layernorm_hwc_elu_bwd DEFERS the parameter-gradient reductions (dgamma, dbeta, dbias_prev): the
 * partial sums stay in `workspace` (give every layer its own), and one launch of sgg_layernorm_hwc_bwd_finalize reduces up to 16
 * layers at once (an encoder backward: eleven 20-microsecond launches off its critical path). */
typedef struct {
  const void* workspace;   /* as left by sgg_layernorm_hwc_elu_bwd(…, dgamma = NULL, …) with the same B, HW, C */
  const float* gamma;
  const float* stats;
  float* dgamma;
  float* dbeta;
  float* dbias_prev;       /* may be NULL */
  int B, HW, C, HW_valid;  /* HW_valid: pixels of the valid window (= HW without one) */
} sgg_ln_finalize_desc;
int sgg_layernorm_hwc_bwd_finalize(const sgg_ln_finalize_desc* layers, int n_layers, void* stream);

/* ---- initial LSTM state: tf.reduce_mean(downsampled, axis=(1,2)) -------------------------------------------
 * generator_with_attention.py:76-77.  Rows r in [0,R) use image r % B (R/B passes share one feature map). */
int sgg_spatial_mean_fwd(const float* ctx, float* out_c, int ldc, float* out_h, int ldh, int R, int B, int L, int C,
                         void* stream);
int sgg_spatial_mean_bwd(const float* dc0, int ldc, const float* dh0, int ldh, float* dctx, int R, int B, int L, int C,
                         int accumulate, void* stream);

/* ---- dense contractions on f32 MFMA: tf.layers.dense / LSTM kernel matmul / tf.matmul(indices, W) ----------
 * generator_with_attention.py:15,87,88; discriminator_with_attention.py:15,87,89,90.
 *   fwd   C[M,N] (+)= A[M,K] B[K,N] (+ bias[N])
 *   dgrad C[M,N] (+)= A[M,K] B[N,K]^T
 *   wgrad C[M,N] (+)= A[K,M]^T B[K,N]
 * Split-K partial slabs go to `workspace` (sgg_gemm_workspace_bytes). */
size_t sgg_gemm_workspace_bytes(int M, int N, int K);
int sgg_gemm_skinny_fwd(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        const float* bias, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
int sgg_gemm_skinny_dgrad(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          int accumulate, void* workspace, size_t workspace_bytes, void* stream);
int sgg_gemm_skinny_wgrad(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* the step-invariant part of the attention perceptron, ctx_flat[B, L*C] x W_ctx[L*C, L] (same kernels, named
 * for the call site generator_with_attention.py:15) */
int sgg_attn_ctx_gemm_fwd(int B, int L, int LC, const float* ctx_flat, const float* w_ctx, const float* bias, float* P,
                          void* workspace, size_t workspace_bytes, void* stream);
int sgg_attn_ctx_gemm_dgrad(int B, int L, int LC, const float* dP, const float* w_ctx, float* dctx_flat, int accumulate,
                            void* workspace, size_t workspace_bytes, void* stream);
int sgg_attn_ctx_gemm_wgrad(int B, int L, int LC, const float* ctx_flat, const float* dP, float* dw_ctx, int accumulate,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ---- tf.matmul(indices, W) with one-hot indices = embedding row gather ----------------------------------------
 * discriminator_with_attention.py:86-87 on the real triples (float one-hots, train.py:173).
 *   fwd  out[r, 0:E] = W[labels[r * label_stride], :]   (zero row for a label outside [0, V), as tf.one_hot)
 *   bwd  dW[labels[r * label_stride], :] += dY[r, 0:E]  (rows sharing a label summed in row order: deterministic) */
int sgg_embed_gather_fwd(const long long* labels, int label_stride, const float* W, int V, int E, float* out, int ldo, int R,
                         void* stream);
int sgg_embed_gather_bwd(const long long* labels, int label_stride, const float* dY, int lddy, float* dW, int V, int E, int R,
                         void* stream);

/* ---- attentionMechanism: score add -> softmax over L -> weighted sum of feature rows ------------------------
 * generator_with_attention.py:13-18 / discriminator_with_attention.py:13-18.
 *   e[r,:] = P[r % B,:] + ec[r,:]   (ec = c @ W_c, by sgg_gemm_skinny_fwd)
 *   alpha = softmax(e);  z[r,:] = sum_l alpha[r,l] * ctx[r % B, l, :]
 * bwd: de (cotangent of ec), dP[b,:] (+)= sum over its rows, dctx[b] (+)= alpha (x) dz.  With the *_dual
 * pointers it is the dual-number evaluation used for the gradient-penalty term ("bwd2"). */
int sgg_attn_step_fwd(const float* P, const float* ec, const float* ec_dual, int ldec, const float* ctx, float* alpha,
                      float* alpha_dual, float* z, float* z_dual, int ldz, int R, int B, int L, int C, void* stream);
int sgg_attn_step_bwd(const float* ctx, const float* alpha, const float* alpha_dual, const float* dz,
                      const float* dz_dual, int lddz, float* de, float* de_dual, float* dP, float* dctx, int R, int B,
                      int L, int C, int accumulate, void* stream);

/* ---- tf.contrib.rnn.LayerNormBasicLSTMCell(512) pointwise part ---------------------------------------------
 * generator_with_attention.py:79,87 / discriminator_with_attention.py:81,89.  gates = [x,h] @ kernel, [R,2048] in
 * the order i, j, f, o.  ln_params = [10][512]: gamma,beta of scopes input, transform, forget, output, state.
 * One wave per row; wave-level reductions for the five LayerNorms. pgrad: [R][10][512] per-row LN-parameter
 * gradients (sum the rows with sgg_colsum). */
int sgg_lnlstm_gates_fwd(const float* gates, const float* gates_dual, const float* c_prev, const float* c_prev_dual,
                         const float* ln_params, float* c_new, float* c_new_dual, float* h_new, float* h_new_dual,
                         int ldh, int R, void* stream);
int sgg_lnlstm_gates_bwd(const float* gates, const float* gates_dual, const float* c_prev, const float* c_prev_dual,
                         const float* ln_params, const float* dh, const float* dh_dual, int lddh, const float* dc_new,
                         const float* dc_new_dual, float* dgates, float* dgates_dual, float* dc_prev,
                         float* dc_prev_dual, float* pgrad, int R, void* stream);
size_t sgg_colsum_workspace_bytes(int rows, int cols);
int sgg_colsum(const float* X, int rows, int cols, int ld, float* out, int accumulate, void* workspace,
               size_t workspace_bytes, void* stream);

/* ---- WGAN-GP loss: tf.contrib.gan.gan_loss(wasserstein_*, gradient_penalty_weight, one_sided) ---------------
 * train.py:245-250 */
int sgg_onehot(const long long* labels, float* out, int rows, int V, void* stream);            /* train.py:173 */
int sgg_interpolate(const float* real, const float* fake, const float* alpha, float* out, int B, int n, void* stream);
int sgg_wgan_gp_loss_fwd(const float* g, float* slopes, float* pen, int B, int n, void* stream);
int sgg_wgan_gp_loss_bwd(const float* g, const float* slopes, const float* pen, float* v, int B, int n, float scale,
                         void* stream);
/* out4 = { disc_cost, wasserstein distance term, gradient penalty, mean D(fake) }  (gen_cost = -out4[3]) */
int sgg_wgan_losses(const float* d_out, const float* pen, float lam, int B, int T, int has_real, float* out4, void* stream);

/* ---- input pipeline: tf.image.resize_images(decoded, [221, 221]) + standardisation: train.py:171-172 ---------------------
 * TF 1.x bilinear resize with its defaults (align_corners=False: legacy grid src = dst * in/out, no antialiasing) of B RGB
 * uint8 images of different sizes ([H_b][W_b][3] at src + offsets[b], device memory) into dst [B][out_h][out_w][3] fp32,
 * then (x - means[c]) / stds[c]. */
int sgg_resize_bilinear_tf1(const unsigned char* src, const long long* offsets, const int* heights, const int* widths, float* dst,
                            int B, int out_h, int out_w, const float* means, const float* stds, void* stream);

/* ---- tf.train.AdamOptimizer(1e-4, beta1=0.5, beta2=0.9).minimize: train.py:258-266 ---------------------------
 * lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) is computed by the caller; theta -= lr_t*m/(sqrt(v)+eps). */
int sgg_adam_tf_multi(float* params, const float* grads, float* m, float* v, long long n, float lr_t, float beta1,
                      float beta2, float eps, float grad_scale, void* stream);

/* ---- tf.argmax(x, axis=-1): train.py:270-271 ---------------------------------------------------------------- */
int sgg_argmax_rows(const float* x, long long* out, int rows, int V, int ld, void* stream);

int sgg_fill(float* p, long long n, float value, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGG_HIP_H */
