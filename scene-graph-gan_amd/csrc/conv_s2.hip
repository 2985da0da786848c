// Band-resident 5x5 / stride-2 convolution for the split 16-bit modes (gfx950): forward and dgrad of
//   tf.layers.conv2d(kernel_size=5, strides=2, padding="same")   reference: architectures/generator_with_attention.py:35,50,65,68
// (conv1_3, conv2_5, conv3_5, `downsampled` and their Conv2DBackpropInput, train.py:265-266) on even input sizes
// (SAME pads (1,2)).
//
// The gather kernel (conv_gather.hip) re-stages a shifted, strided input tile for each of the 25 taps: 25/4 global reads
// and f32 -> fp16 splits per input element.  Here the stride-2 convolution is taken apart by the PARITY of the input pixel:
// with X_q(a, c) = x(2a + qy, 2c + qx) (four half-resolution sub-images),
//     y(oy, ox) = sum over classes q, sum over (u, v) in {-1,0,1}^2 or a subset:  X_q(oy + u, ox + v) . w(kh(u,qy), kw(v,qx))
// which is a 3x3 / 3x2 / 2x3 / 2x2 stride-1 correlation per class (9 + 6 + 6 + 4 = 25 taps).  The gradient w.r.t. x has the
// same shape read the other way: the pixels of dx with parity q are a stride-1 correlation of dy with the taps of class q.
//
//   * Output positions are the FLAT index p over [B][Ho][Wo] of the half-resolution grid, cut into bands of 224 = 7 MFMA
//     row tiles of 32: no tile is wasted on 28x28 or 14x14 grids (bands may cross rows and images).
//   * Per (16-channel chunk, class) the rows of X_q (forward) or dy (dgrad) that a band touches, plus one halo row / column
//     on each side, are loaded, split into two fp16 planes and written to LDS ONCE (double buffered); every tap of the class
//     reads its MFMA A fragments from that patch at a shifted slot.  Patch rows live in a "padded row space"
//     (image b, row a  ->  b*(Ho+2) + a + 1), so the zero rows above / below an image are ordinary patch rows and a band
//     that crosses from one image into the next needs no special case.
//   * Weights are MFMA B fragments in L2 (sgg_conv_split_weights_frag, 25 taps), prefetched one tap ahead into registers.
//   * A workgroup = 4 waves = one band x 128 output channels; every wave holds the band's 7 row tiles for its 32 channels
//     (112 accumulator registers); the four waves share the patch.
// LDS image: plane[pp][slot][16 channels] = 32-B slots; the two 16-B halves of a slot are swapped when bit 3 of the slot
// index is set, which makes the ds_read_b128 of 32 consecutive slots conflict free (lane groups of MI355X_MICROARCH.md, LDS).
// Patch rows are stored WITHOUT halo columns (pitch = Wo), so the 32 positions of a row tile are 32 consecutive slots even
// where the tile runs over the end of an image row (with a two-slot gap per row 42 % of the LDS cycles were bank conflicts,
// SQ_LDS_BANK_CONFLICT); the lanes whose tap falls off the left / right image edge read a reserved all-zero slot instead.
#include "split16.h"
#include "conv_halo.h"
#include <type_traits>

// Timing-only ablations (wrong results by design; scripts/build_variant_one.sh): S2_ABL_NOSTAGE = no patch staging after the first
// stage, S2_ABL_NOB = the weight fragments are loaded once, S2_ABL_NOEPI = no output stores / statistics, S2_ABL_NOA = the A
// fragments are read once per stage (no LDS reads in the tap loop).
#ifndef S2_ABL_NOSTAGE
#define S2_ABL_NOSTAGE 0
#endif
#ifndef S2_ABL_NOB
#define S2_ABL_NOB 0
#endif
#ifndef S2_ABL_NOEPI
#define S2_ABL_NOEPI 0
#endif
#ifndef SGG_WIDE_STORE
#define SGG_WIDE_STORE 1        // 16-byte output stores through an in-register quad transpose (sgg_common.h); 0: 4-byte stores
#endif
#define S2_BAND 224                   // positions per band in the 7-tile variant (what LayerNorm partials are defined on)
#define S2_MAXSLOTS 496               // 2 buffers x 2 planes x 496 x 32 B + the row tables stay inside 64 KB of static LDS
#define S2_ZSLOT (S2_MAXSLOTS - 1)    // never part of a patch: staged as zeros (out-of-range loads), read by edge lanes
#define S2_BN 128
#ifndef S2_SMALL_ITEMS
#define S2_SMALL_ITEMS 256            // at most this many 224-position work items: use 128-position bands instead
#endif
// Few work items (`downsampled` at batch 64: 56 bands x 4 n-tiles on 256 CUs): 0 = 128-position bands (392 items); 1 = 224-position
// bands with the channel chunks split over two workgroups that ADD their partials into the zeroed output (448 items; a + b = b + a:
// deterministic); 2 = both.  Measured 49.34 / 48.92 / 49.60 ms per G+D step (same box, two repetitions).
#ifndef S2_KSPLIT_MODE
#define S2_KSPLIT_MODE 1
#endif
#ifndef S2_DMA
#define S2_DMA 1              // 0: a pre-split source is staged through registers (no arithmetic) like an f32 one
#endif
#ifndef S2_WIDE
#define S2_WIDE 1             // 0: always 128-column workgroups
#endif
#ifndef S2_SWZ
#define S2_SWZ(slot) (((slot) >> 3) & 1)      // which 16-B half of a slot holds channels 0..7
#endif

namespace {

// taps of a parity class along one axis: kernel index and patch shift (0..2; the band's own row / column is shift 1)
//   forward, parity 1: k = 0, 2, 4 at shifts 0, 1, 2      parity 0: k = 1, 3 at shifts 1, 2
//   dgrad,   parity 1: k = 0, 2, 4 at shifts 2, 1, 0      parity 0: k = 1, 3 at shifts 1, 0
template <bool DGRAD>
struct Axis {
  static constexpr int count(int par) { return par ? 3 : 2; }
  static constexpr int k(int par, int j) { return par ? 2 * j : 2 * j + 1; }
  static constexpr int shift(int par, int j) { return DGRAD ? (par ? 2 - j : 1 - j) : (par ? j : j + 1); }
};
// class 0 = (qy 1, qx 1): 9 taps, 1 = (1, 0): 6, 2 = (0, 1): 6, 3 = (0, 0): 4
constexpr int cls_qy(int cls) { return cls < 2 ? 1 : 0; }
constexpr int cls_qx(int cls) { return (cls & 1) ? 0 : 1; }
constexpr int cls_ntaps(int cls) { return (cls_qy(cls) ? 3 : 2) * (cls_qx(cls) ? 3 : 2); }

}  // namespace

// MT = MFMA row tiles per band: 7 (224 positions), or 4 (128 positions) where 224-position bands would give fewer
// (band, n-tile) work items than the chip has CUs (`downsampled` at batch 64: 56 bands x 4 n-tiles).
// ONE: single-piece mode (precision 1 / 4): one 16-bit plane, one MFMA per product.
// LNP: LN prologue (forward): src is the producing layer's pre-LayerNorm output, normalised + ELU'd while the patch is staged.
// DMAP: src is a PRE-SPLIT tensor (split16.h; HALF, two pieces, no prologue): the patch goes HBM -> LDS by LDS-DMA (inline assembly:
// invisible to the compiler's wait-count pass, waited for by an explicit s_waitcnt at the end of the stage), issued at the stage's
// FIRST tap into the buffer the workgroup left at the last barrier - no staging registers, loads, conversions or LDS writes.
// NW: waves per workgroup = 32-column slices of the band it owns: 4 (128 output columns, two workgroups per CU) or 8 (256 columns, ONE
// workgroup per CU: the band's patch is staged once per 256 instead of once per 128 output columns - half the x traffic of the layers
// with 256+ output channels; DMAP only).
template <bool DGRAD, bool HALF, int MT, bool ONE = false, bool LNP = false, bool DMAP = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void conv_s2_kernel(S2Params p) {
  static_assert(NW == 4 || (NW == 8 && DMAP), "256-column workgroups: patch by DMA only");
  constexpr int NT = 64 * NW;                 // threads
  constexpr int BN = 32 * NW;                 // output columns per workgroup
  constexpr int NPASS = 1024 / NT;            // (slot, 8-channel half) items per thread: 992 -> 4 passes of 256 / 2 of 512
  static_assert(!LNP || (!DGRAD && !ONE), "the LN prologue exists for the forward of the two-piece modes");
  static_assert(!DMAP || (HALF && !ONE && !LNP), "patch DMA: pre-split fp16 pieces, no prologue");
  constexpr int P = ONE ? 1 : 2;
  // bytes of one plane of one patch buffer; DMAP: 512 slots, because the 32nd DMA instruction of a plane covers slots 480 .. 511
  constexpr int S2_PLB = DMAP ? 16384 : S2_MAXSLOTS * 32;
  __shared__ __attribute__((aligned(16))) float lnp_s[LNP ? 1024 : 4];      // gamma[0..511], beta at +512 (C <= 512: host check)
  constexpr int BAND = 32 * MT;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * P * S2_PLB];
  __shared__ __attribute__((aligned(16))) int rowtab[2][BAND];      // byte offset of each band row in `out` (-1: past the end)

  // ---- persistent workgroup: XCD k owns a contiguous eighth of the bands; its p.gx workgroups walk (band, n-tile) pairs -----
  const int ntn = (p.N / BN) * p.ksplit;                     // (n-tile, channel half) pairs: static per workgroup
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int ntk = jx % ntn;
  const int nt = ntk % (p.N / BN), khalf = ntk / (p.N / BN);
  const int bstride = p.gx / ntn;
  const int band_begin = (int)(((long long)xcd * p.nbands) >> 3) + jx / ntn;
  const int band_end = (int)(((long long)(xcd + 1) * p.nbands) >> 3);
  if (band_begin >= band_end) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = nt * BN + wave * 32;                     // this wave's 32 output channels
  const int i = lane & 31, h = lane >> 5;

  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, p.w_bytes, 0x00020000);
  int ea = 0, eb = 0;
  if constexpr (HALF) {
    ea = scale_exp_from_amax(*p.amax_src);
    eb = scale_exp_from_amax(*p.amax_w);
  }
  const float sa = ldexpf(1.f, ea);
  const float us_a = ldexpf(1.f, -ea), us_b = ldexpf(1.f, -eb);
  const int Hp = p.Ho + 2;                                   // padded rows per image
  const int nch = (p.C >> 4) / p.ksplit;                     // 16-channel chunks this workgroup contracts (even: host check)
  const int cbeg = khalf * nch;                              // its first chunk
  const int nstage = 4 * nch;

  // ---- staging plan: item = (slot, half); static per thread: patch row / column and LDS byte offset -----------------------
  // (patch row / column are recomputed once per band and the LDS offset per write: cheaper than 12 live registers)
  // staging state (runs ahead of the compute state): band, stage inside the band, per-item source offsets of that band
  int s_band = band_begin, s_stage = 0;
  unsigned s_base[NPASS];
  int s_b[LNP ? NPASS : 1];              // LN prologue: sample of each item of the band being staged
  float ld_mu[LNP ? NPASS : 1], ld_rs[LNP ? NPASS : 1];     // ... and the (mean, rstd) / chunk of the patch in flight
  int ld_cc = 0, ld_bad = 0;
  auto band_geometry = [&](int band, int& pg_first, int& nrows) __attribute__((always_inline)) {
    const int p0 = band * BAND;
    const int p1 = min(p0 + BAND, p.M) - 1;
    const int g0 = p0 / p.Wo, g1 = p1 / p.Wo;                // first / last global output row of the band
    const int b0 = g0 / p.Ho, b1 = g1 / p.Ho;
    pg_first = b0 * Hp + (g0 - b0 * p.Ho);                   // padded row of the halo row above the first row
    nrows = (b1 * Hp + (g1 - b1 * p.Ho) + 2) - pg_first + 1;
  };
  auto stage_band = [&](int band) __attribute__((always_inline)) {
    if (band >= band_end) {
#pragma unroll
      for (int j = 0; j < NPASS; ++j) s_base[j] = SGG_OOB;
      return;
    }
    int pg_first, nrows;
    band_geometry(band, pg_first, nrows);
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const int slot = (tid + NT * j) >> 1;
      // (DMAP: lane writes LDS position tid & 1 of its slot, so it fetches the half the swizzle puts there)
      const int half = DMAP ? ((tid & 1) ^ S2_SWZ(slot)) : (tid & 1);
      const int lr = slot / p.pitch, lc = slot - lr * p.pitch;
      const int pg = pg_first + lr;
      const int b = pg / Hp;
      const int a = pg - b * Hp - 1, c = lc;
      const bool ok = (lr < nrows) & (b < p.B) & (a >= 0) & (a < p.Ho) & (c >= 0) & (c < p.Wo);
      const unsigned off = DGRAD ? (unsigned)((((b * p.Ho + a) * p.Wo + c) * p.C + half * 8) * 4)
                                 : (unsigned)((((b * 2 * p.Ho + 2 * a) * 2 * p.Wo + 2 * c) * p.C + half * 8) * 4);
      s_base[j] = ok ? off : SGG_OOB;
      if constexpr (LNP) s_b[j] = b < p.B ? b : p.B - 1;
    }
  };
  f32x4 pre[NPASS][2];
  // issue the global loads of the next (band, stage) patch; then advance the staging state
  auto stage_load = [&]() __attribute__((always_inline)) {
    int cls, cc;
    if constexpr (DGRAD) { cls = s_stage / nch; cc = s_stage - cls * nch; }
    else { cc = s_stage >> 2; cls = s_stage & 3; }
    (void)cls;
    cc += cbeg;
    unsigned uni = (unsigned)(cc * 64);
    if constexpr (!DGRAD) uni += (unsigned)((((cls < 2 ? 1 : 0) * 2 * p.Wo + ((cls & 1) ? 0 : 1)) * p.C) * 4);
    if constexpr (LNP) { ld_cc = cc; ld_bad = 0; }
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const unsigned off = s_base[j] == SGG_OOB ? SGG_OOB : s_base[j] + uni;
      const unsigned o0 = LNP ? off : stage_off0(off, p.src_s16);
      pre[j][0] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, o0);
      pre[j][1] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, LNP ? off + 16u : stage_off1(o0, p.src_s16));
      if constexpr (LNP) {
        ld_mu[j] = p.ln_stats[2 * s_b[j]];
        ld_rs[j] = p.ln_stats[2 * s_b[j] + 1];
        ld_bad |= (int)(s_base[j] == SGG_OOB) << j;
      }
    }
    if (++s_stage == nstage) {
      s_stage = 0;
      s_band += bstride;
      stage_band(s_band);
    }
  };
  auto stage_write = [&](unsigned char* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      if constexpr (LNP) {
        const int ch0 = ld_cc * 16 + (tid & 1) * 8;
        ln_elu8(pre[j][0], pre[j][1], lnp_s + ch0, lnp_s + 512 + ch0, ld_mu[j], ld_rs[j], (ld_bad >> j) & 1);
      }
      u32x4 pl[P];
      if constexpr (LNP || !HALF) split8<P, HALF>(pre[j][0], pre[j][1], sa, pl);
      else stage_planes<P, HALF>(pre[j][0], pre[j][1], sa, p.src_s16, pl);
      if (tid + NT * j < 2 * S2_MAXSLOTS) {
        const int slot = (tid + NT * j) >> 1;
        const int lds_off = slot * 32 + (((tid & 1) ^ S2_SWZ(slot)) << 4);
#pragma unroll
        for (int pp = 0; pp < P; ++pp) *reinterpret_cast<u32x4*>(dst + pp * S2_PLB + lds_off) = pl[pp];
      }
    }
  };

  // DMAP: the four (slot, half) items of this lane as four DMA instructions per plane (instruction j of wave w fills LDS bytes
  // [1024 w + 16 NT j, + 1024) of a plane: items 64 w + NT j .. + 63, lane l at + 16 l); then the staging state advances like stage_load
  const unsigned s2_rs[4] = {(unsigned)reinterpret_cast<unsigned long long>(p.src), (unsigned)(reinterpret_cast<unsigned long long>(p.src) >> 32) & 0xffffu,
                             p.src_bytes, 0x00020000u};
  auto stage_dma = [&](unsigned char* dst) __attribute__((always_inline)) {
    typedef int s2_v4i __attribute__((ext_vector_type(4)));
    const s2_v4i rs = s2_v4i{(int)s2_rs[0], (int)s2_rs[1], (int)s2_rs[2], (int)s2_rs[3]};
    int cls, cc;
    if constexpr (DGRAD) { cls = s_stage / nch; cc = s_stage - cls * nch; }
    else { cc = s_stage >> 2; cls = s_stage & 3; }
    (void)cls;
    cc += cbeg;
    unsigned uni = (unsigned)(cc * 64);
    if constexpr (!DGRAD) uni += (unsigned)((((cls < 2 ? 1 : 0) * 2 * p.Wo + ((cls & 1) ? 0 : 1)) * p.C) * 4);
    const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)dst + (unsigned)wave * 1024u;
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const unsigned off = s_base[j] == SGG_OOB ? SGG_OOB : s16_hi_off(s_base[j] + uni);
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :
                     : "s"(la + (unsigned)(j * (NT * 16) + pp * S2_PLB)), "v"(off), "s"(rs), "s"(pp * 64)
                     : "memory");      // (M0: a RESERVED register of this target - the compiler keeps no value in it across statements and
                                       //  rejects it in a clobber list (-Winline-asm); build.check_m0_users verifies on the linked code
                                       //  objects that this kernel has no other M0 user)
    }
    if (++s_stage == nstage) {
      s_stage = 0;
      s_band += bstride;
      stage_band(s_band);
    }
  };

  // ---- weights: B fragments from L2, layout [tap][C/32][N/32][k-step][plane][lane] x 16 B (16-channel chunk cc = 2*c32 + k-step)
  const unsigned w_lane = (unsigned)(n0 >> 5) * 4096u + (unsigned)lane * 16u;
  const unsigned w_slab = (unsigned)(p.N >> 5) * 4096u;
  const int nch32 = p.C >> 5;
  u32x4 rb[2][P];
  auto load_b = [&](auto par_c, int cc, int tap) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value;
    const unsigned base = (unsigned)(tap * nch32 + (cc >> 1)) * w_slab + (unsigned)((cc & 1) * 2048) + w_lane;
#pragma unroll
    for (int pp = 0; pp < P; ++pp) rb[par][pp] = __builtin_bit_cast(u32x4, buf_load4(rs_w, base + (unsigned)(pp * 1024)));
  };

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  int slot0[MT];          // patch slot of this lane's output position in row tile t (the centre of its 3x3 neighbourhood)
  bool edge_l[MT], edge_r[MT];   // this lane's position is in the first / last column of its image row (lane masks)
  int cur = 0;               // patch buffer the current stage reads
  float bias_v = 0.f;
  if constexpr (!DGRAD) bias_v = (p.bias && khalf == 0) ? p.bias[n0 + i] : 0.f;

  // A fragments go through a ring of three register sets: the fragments of row tile k+2 (of this tap or the next one) are
  // read from the resident patch while the MFMAs of tile k issue, so an LDS latency is exposed only at the start of a stage.
  u32x4 ra[3][P];
  auto read_a = [&](auto ring_c, auto t_c, auto dx_c, int shift) __attribute__((always_inline)) {
    constexpr int ring = decltype(ring_c)::value, t = decltype(t_c)::value, dxx = decltype(dx_c)::value;
    // (opaque to the optimiser: otherwise the 175 per-tap LDS addresses, invariant across the chunk loop, are hoisted and spilled)
    int sc = slot0[t];
    asm volatile("" : "+v"(sc));
    const int s = sc + shift;
    int off = s * 32 + ((h ^ S2_SWZ(s)) << 4);
    if constexpr (dxx == 0) off = edge_l[t] ? S2_ZSLOT * 32 : off;      // column -1 of the image: zero padding
    if constexpr (dxx == 2) off = edge_r[t] ? S2_ZSLOT * 32 : off;      // column Wo
    const unsigned char* a_ptr = lds + cur * (P * S2_PLB) + off;
    ra[ring][0] = *reinterpret_cast<const u32x4*>(a_ptr);
    if constexpr (P == 2) ra[ring][1] = *reinterpret_cast<const u32x4*>(a_ptr + S2_PLB);
  };
  auto mma_tile = [&](auto ring_c, auto t_c, auto par_c) __attribute__((always_inline)) {
    constexpr int ring = decltype(ring_c)::value, t = decltype(t_c)::value, par = decltype(par_c)::value;
    f32x16 d = acc[t];
    if constexpr (P == 2) {
      d = mfma16<HALF>(ra[ring][1], rb[par][0], d);
      d = mfma16<HALF>(ra[ring][0], rb[par][1], d);
    }
    d = mfma16<HALF>(ra[ring][0], rb[par][0], d);
    acc[t] = d;
  };

  // One stage = one (16-channel chunk, class): its taps statically unrolled.  Every tap prefetches the next tap's B fragments
  // (next_tap / next_cc: first tap of the following stage); the next stage's patch is loaded three taps before the end and
  // split + written to the other buffer during the last tap.
  auto stage = [&](auto cls_c, auto par0_c, int cc, int next_cc, int next_tap) __attribute__((always_inline)) {
    constexpr int cls = decltype(cls_c)::value, par0 = decltype(par0_c)::value;
    constexpr int qy = cls_qy(cls), qx = cls_qx(cls);
    constexpr int ny = Axis<DGRAD>::count(qy), nx = Axis<DGRAD>::count(qx), ntaps = ny * nx;
    auto tap_shift = [&](auto ti_c) __attribute__((always_inline)) {
      constexpr int ti = decltype(ti_c)::value;
      return (Axis<DGRAD>::shift(qy, ti / nx) - 1) * p.pitch + (Axis<DGRAD>::shift(qx, ti % nx) - 1);
    };
    constexpr int dx0 = Axis<DGRAD>::shift(qx, 0);
    read_a(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, dx0>{}, tap_shift(std::integral_constant<int, 0>{}));
    read_a(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, dx0>{}, tap_shift(std::integral_constant<int, 0>{}));
    auto body = [&](auto ti_c) __attribute__((always_inline)) {
      constexpr int ti = decltype(ti_c)::value;
      constexpr int par = (par0 + ti) & 1;
#if !S2_ABL_NOB
        if constexpr (ti + 1 < ntaps) {
        constexpr int njy = (ti + 1) / nx, njx = (ti + 1) % nx;
        load_b(std::integral_constant<int, par ^ 1>{}, cc, Axis<DGRAD>::k(qy, njy) * 5 + Axis<DGRAD>::k(qx, njx));
      } else {
        load_b(std::integral_constant<int, par ^ 1>{}, next_cc, next_tap);
      }
#endif
#if !S2_ABL_NOSTAGE
      if constexpr (DMAP) {
        if constexpr (ti == 0) stage_dma(lds + (cur ^ 1) * (P * S2_PLB));
      } else if constexpr (ti == (ntaps >= 6 ? ntaps - 3 : 1)) stage_load();
#endif
      __builtin_amdgcn_sched_barrier(0);
#if !S2_ABL_NOSTAGE
      if constexpr (!DMAP && ti == ntaps - 1) stage_write(lds + (cur ^ 1) * (P * S2_PLB));
#endif
      auto tile = [&](auto t_c) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        constexpr int k = ti * MT + t;
        if constexpr (k + 2 < ntaps * MT) {
          constexpr int nti = (k + 2) / MT, nt2 = (k + 2) % MT;
          read_a(std::integral_constant<int, (k + 2) % 3>{}, std::integral_constant<int, nt2>{},
                 std::integral_constant<int, Axis<DGRAD>::shift(qx, nti % nx)>{}, tap_shift(std::integral_constant<int, nti>{}));
        }
        mma_tile(std::integral_constant<int, k % 3>{}, t_c, std::integral_constant<int, par>{});
      };
      SGG_PRIO_HI();
      tile(std::integral_constant<int, 0>{}); tile(std::integral_constant<int, 1>{}); tile(std::integral_constant<int, 2>{});
      tile(std::integral_constant<int, 3>{});
      if constexpr (MT > 4) { tile(std::integral_constant<int, 4>{}); tile(std::integral_constant<int, 5>{}); tile(std::integral_constant<int, 6>{}); }
      static_assert(MT == 4 || MT == 7, "row tiles per band");
      SGG_PRIO_LO();
      __builtin_amdgcn_sched_barrier(0);
    };
    body(std::integral_constant<int, 0>{}); body(std::integral_constant<int, 1>{});
    body(std::integral_constant<int, 2>{}); body(std::integral_constant<int, 3>{});
    if constexpr (ntaps > 4) { body(std::integral_constant<int, 4>{}); body(std::integral_constant<int, 5>{}); }
    if constexpr (ntaps > 6) { body(std::integral_constant<int, 6>{}); body(std::integral_constant<int, 7>{}); body(std::integral_constant<int, 8>{}); }
    cur ^= 1;
    if constexpr (DMAP) __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): this wave's patch DMAs have landed
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  constexpr int TAP0[4] = {Axis<DGRAD>::k(1, 0) * 5 + Axis<DGRAD>::k(1, 0), Axis<DGRAD>::k(1, 0) * 5 + Axis<DGRAD>::k(0, 0),
                           Axis<DGRAD>::k(0, 0) * 5 + Axis<DGRAD>::k(1, 0), Axis<DGRAD>::k(0, 0) * 5 + Axis<DGRAD>::k(0, 0)};

  // ---- epilogue: unscale (+ bias), store through the band's row table; forward: optional LayerNorm partials ----------------
  auto epilogue = [&](int band, int tabsel, int cls_off_bytes) __attribute__((always_inline)) {
    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = HALF ? fmaf(acc[t][r] * us_a, us_b, bias_v) : acc[t][r] + bias_v;
        acc[t][r] = v;
        lsum += v;
      }
    char* ob = reinterpret_cast<char*>(p.out) + cls_off_bytes + (size_t)(n0 + i) * 4;
#if SGG_WIDE_STORE
    if (p.ksplit <= 1) {
      // 16-byte stores: after the quad transpose lane (h, g, k) = (lane >> 5, i >> 2, i & 3) holds position rq * 8 + 4 h + k of the
      // row tile and channels 4g .. 4g+3 (sgg_common.h: sgg_quad_transpose4)
      char* ow = reinterpret_cast<char*>(p.out) + cls_off_bytes + (size_t)(n0 + (i & ~3)) * 4;
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          float v0 = acc[t][rq * 4], v1 = acc[t][rq * 4 + 1], v2 = acc[t][rq * 4 + 2], v3 = acc[t][rq * 4 + 3];
          sgg_quad_transpose4(v0, v1, v2, v3, lane);
          const int o = rowtab[tabsel][t * 32 + rq * 8 + 4 * h + (i & 3)];
          if (o >= 0) sgg_out_store4(reinterpret_cast<float*>(ow + o), f32x4{v0, v1, v2, v3});
        }
    }
    const bool narrow = p.ksplit > 1;
#else
    const bool narrow = true;
#endif
    if (narrow) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int4 o4 = *reinterpret_cast<const int4*>(&rowtab[tabsel][t * 32 + rq * 8 + 4 * h]);
        if (p.ksplit > 1) {
          if (o4.x >= 0) atomicAdd(reinterpret_cast<float*>(ob + o4.x), acc[t][rq * 4 + 0]);
          if (o4.y >= 0) atomicAdd(reinterpret_cast<float*>(ob + o4.y), acc[t][rq * 4 + 1]);
          if (o4.z >= 0) atomicAdd(reinterpret_cast<float*>(ob + o4.z), acc[t][rq * 4 + 2]);
          if (o4.w >= 0) atomicAdd(reinterpret_cast<float*>(ob + o4.w), acc[t][rq * 4 + 3]);
        } else {
          if (o4.x >= 0) sgg_out_store(reinterpret_cast<float*>(ob + o4.x), acc[t][rq * 4 + 0]);
          if (o4.y >= 0) sgg_out_store(reinterpret_cast<float*>(ob + o4.y), acc[t][rq * 4 + 1]);
          if (o4.z >= 0) sgg_out_store(reinterpret_cast<float*>(ob + o4.z), acc[t][rq * 4 + 2]);
          if (o4.w >= 0) sgg_out_store(reinterpret_cast<float*>(ob + o4.w), acc[t][rq * 4 + 3]);
        }
      }
    }
    if constexpr (!DGRAD) {
      if (p.tile_stats) {
        // (count, mean, M2) of this wave's 224 positions x 32 channels (bands align with samples: host check)
        const float cnt = (float)(BAND * 32);
        const float mean_w = wave_sum(lsum) / cnt;
        float q = 0.f, dm = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[t][r] - mean_w;
            q += d * d;
            dm = fmaxf(dm, fabsf(d));
          }
        q = wave_sum(q);
        dm = wave_max(dm);
        if (lane == 0) {
          float* o = p.tile_stats + ((size_t)band * (p.N >> 5) + (n0 >> 5)) * SGG_TS;
          o[0] = cnt;
          o[1] = mean_w;
          o[2] = q;
          o[3] = dm;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  };

  // ---- prologue: first patch, first B fragments ----------------------------------------------------------------------------
  if constexpr (LNP) {
    for (int c = tid; c < p.C; c += NT) {
      lnp_s[c] = p.ln_gamma[c];
      lnp_s[512 + c] = p.ln_beta[c];
    }
    __syncthreads();
  }
  stage_band(s_band);
  if constexpr (DMAP) {
    stage_dma(lds);
    load_b(std::integral_constant<int, 0>{}, cbeg, TAP0[0]);
    __builtin_amdgcn_s_waitcnt(0x0070);
  } else {
    stage_load();
    load_b(std::integral_constant<int, 0>{}, cbeg, TAP0[0]);
    stage_write(lds);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

#if S2_ABL_NOB
  load_b(std::integral_constant<int, 1>{}, cbeg, TAP0[0]);
#endif
  int tabsel = 0;
  for (int band = band_begin; band < band_end; band += bstride, tabsel ^= 1) {
    // compute state of this band: row table (out offsets) and the per-lane patch slots of the 7 row tiles
    int pg_first, nrows;
    band_geometry(band, pg_first, nrows);
    const int p0 = band * BAND;
    if (tid < BAND) {
      const int pp = p0 + tid;
      int off = -1;
      if (pp < p.M) {
        if constexpr (DGRAD) {
          const int g = pp / p.Wo, ox = pp - g * p.Wo;
          const int b = g / p.Ho, oy = g - b * p.Ho;
          off = (((b * 2 * p.Ho + 2 * oy) * 2 * p.Wo + 2 * ox) * p.N) * 4;
        } else {
          off = pp * p.N * 4;
        }
      }
      rowtab[tabsel][tid] = off;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int pp = min(p0 + t * 32 + i, p.M - 1);
      const int g = pp / p.Wo, ox = pp - g * p.Wo;
      const int b = g / p.Ho, oy = g - b * p.Ho;
      slot0[t] = (b * Hp + oy + 1 - pg_first) * p.pitch + ox;
      edge_l[t] = ox == 0;
      edge_r[t] = ox == p.Wo - 1;
    }
    // (the row table is read after at least one workgroup barrier: every stage ends with one)
    if constexpr (!DGRAD) {
      for (int cc = cbeg; cc < cbeg + nch; cc += 2) {
        const bool last = cc + 2 >= cbeg + nch;
        stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, cc, cc, TAP0[1]);
        stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, cc, cc, TAP0[2]);
        stage(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, cc, cc, TAP0[3]);
        stage(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, cc, cc + 1, TAP0[0]);
        stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, cc + 1, cc + 1, TAP0[1]);
        stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, cc + 1, cc + 1, TAP0[2]);
        stage(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, cc + 1, cc + 1, TAP0[3]);
        stage(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, cc + 1, last ? cbeg : cc + 2, TAP0[0]);
      }
#if !S2_ABL_NOEPI
      epilogue(band, tabsel, 0);
#endif
    } else {
      auto class_loop = [&](auto cls_c) __attribute__((always_inline)) {
        constexpr int cls = decltype(cls_c)::value;
        constexpr int odd = cls_ntaps(cls) & 1;
        constexpr int ncls = (cls + 1) & 3;
        for (int cc = cbeg; cc < cbeg + nch; cc += 2) {
          const bool last = cc + 2 >= cbeg + nch;
          stage(cls_c, std::integral_constant<int, 0>{}, cc, cc + 1, TAP0[cls]);
          stage(cls_c, std::integral_constant<int, odd>{}, cc + 1, last ? cbeg : cc + 2, last ? TAP0[ncls] : TAP0[cls]);
        }
#if !S2_ABL_NOEPI
        epilogue(band, tabsel, ((cls_qy(cls) * 2 * p.Wo + cls_qx(cls)) * p.N) * 4);
#endif
      };
      class_loop(std::integral_constant<int, 0>{});
      class_loop(std::integral_constant<int, 1>{});
      class_loop(std::integral_constant<int, 2>{});
      class_loop(std::integral_constant<int, 3>{});
    }
  }
#if S2_ABL_NOEPI
  epilogue(band_begin, 0, 0);      // (keeps the accumulators alive: once per workgroup)
#endif
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static int s2_gcd(int a, int b) { return b ? s2_gcd(b, a % b) : a; }

// largest number of patch rows any band needs (the band pattern repeats every 224 / gcd(224, Ho*Wo) images)
static int s2_max_rows(int Ho, int Wo, int band = S2_BAND) {
  const int period = band / s2_gcd(band, Ho * Wo);
  const long long M = (long long)period * Ho * Wo;
  int mx = 0;
  for (long long p0 = 0; p0 < M; p0 += band) {
    const long long p1 = (p0 + band < M ? p0 + band : M) - 1;
    const int g0 = (int)(p0 / Wo), g1 = (int)(p1 / Wo);
    const int b0 = g0 / Ho, b1 = g1 / Ho;
    const int rows = (b1 * (Ho + 2) + (g1 - b1 * Ho) + 2) - (b0 * (Ho + 2) + (g0 - b0 * Ho)) + 1;
    if (rows > mx) mx = rows;
  }
  return mx;
}

int sgg_s2_applicable(int KH, int KW, int stride, int B, int Hi, int Wi, int C, int N, int precision) {
  (void)B;
  if (!(KH == 5 && KW == 5 && stride == 2 && Hi > 0 && Wi > 0 && Hi % 2 == 0 && Wi % 2 == 0 && C % 32 == 0 && N % S2_BN == 0 &&
        sgg_prec_resident(precision)))
    return 0;
  const int Ho = Hi / 2, Wo = Wi / 2;
  return s2_max_rows(Ho, Wo) * Wo <= S2_ZSLOT;      // the patch (pitch Wo, no halo columns) must leave the zero slot free
}

int sgg_s2_stats_per_sample(int Ho, int Wo, int N) { return ((Ho * Wo) % S2_BAND == 0) ? (Ho * Wo / S2_BAND) * (N / 32) : 0; }

void sgg_s2_launch(const S2Params& p_, int dgrad, int precision, hipStream_t st) {
  S2Params p = p_;
  const bool half = sgg_prec_half(precision), one = sgg_prec_one(precision);
  const bool dmap = S2_DMA && p.src_s16 && half && !one && !p.ln_stats;      // pre-split source: the patch by LDS-DMA
  // 256-column workgroups (eight waves, one workgroup per CU) where the layer has 256+ output columns and the patch comes by DMA
  // (not where that leaves fewer work items than half the CUs - `downsampled` at batch 64: measured equal forward, 7 % slower dgrad)
  const bool wide = S2_WIDE && dmap && p.N % 256 == 0 && sgg_cdiv(p.M, S2_BAND) * (p.N / 256) > S2_SMALL_ITEMS / 2;
  const int bn = wide ? 256 : S2_BN;
  // 224-position bands unless they give fewer work items than CUs (and no LayerNorm partials are asked for): then 128-position
  // bands (S2_KSPLIT_MODE 0), or 224-position bands with the channel chunks split over two workgroups (1), or both (2)
  const bool small_ = !wide && !p.tile_stats && !p.ln_stats && sgg_cdiv(p.M, S2_BAND) * (p.N / bn) <= S2_SMALL_ITEMS;
  const int mt = (small_ && S2_KSPLIT_MODE != 1) ? 4 : 7;
  p.ksplit = (small_ && S2_KSPLIT_MODE != 0 && (p.C >> 4) % 4 == 0) ? 2 : 1;
  if (p.ksplit > 1)
    (void)hipMemsetAsync(p.out, 0, (size_t)(dgrad ? 4 : 1) * p.M * p.N * sizeof(float), st);
  const int ntn = (p.N / bn) * p.ksplit;
  p.nbands = sgg_cdiv(p.M, 32 * mt);
  const int slots = (wide ? 1 : 2) * sgg_persist_cus(p.cu_cap);      // resident workgroups per XCD (32 CUs)
  int per_xcd = sgg_cdiv(p.nbands, 8) * ntn;       // (band, n-tile, channel half) items an XCD owns
  int gx = per_xcd < slots ? per_xcd : slots;
  gx = sgg_cdiv(gx, ntn) * ntn;
  p.gx = gx;
  const dim3 grid((unsigned)(8 * gx));
  if (wide) {
    if (dgrad) hipLaunchKernelGGL((conv_s2_kernel<true, true, 7, false, false, true, 8>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_s2_kernel<false, true, 7, false, false, true, 8>), grid, dim3(512), 0, st, p);
    return;
  }
  if (dmap) {
    if (mt == 4) {
      if (dgrad) hipLaunchKernelGGL((conv_s2_kernel<true, true, 4, false, false, true>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((conv_s2_kernel<false, true, 4, false, false, true>), grid, dim3(256), 0, st, p);
    } else {
      if (dgrad) hipLaunchKernelGGL((conv_s2_kernel<true, true, 7, false, false, true>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((conv_s2_kernel<false, true, 7, false, false, true>), grid, dim3(256), 0, st, p);
    }
    return;
  }
  if (p.ln_stats) {       // LN prologue: forward, two-piece modes, 224-position bands (host checks in sgg_conv2d_nhwc_fwd)
    if (half) hipLaunchKernelGGL((conv_s2_kernel<false, true, 7, false, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_s2_kernel<false, false, 7, false, true>), grid, dim3(256), 0, st, p);
    return;
  }
#define SGG_S2(MT)                                                                                      \
  do {                                                                                                  \
    if (one) {                                                                                          \
      if (dgrad) {                                                                                      \
        if (half) hipLaunchKernelGGL((conv_s2_kernel<true, true, MT, true>), grid, dim3(256), 0, st, p);  \
        else hipLaunchKernelGGL((conv_s2_kernel<true, false, MT, true>), grid, dim3(256), 0, st, p);      \
      } else {                                                                                          \
        if (half) hipLaunchKernelGGL((conv_s2_kernel<false, true, MT, true>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((conv_s2_kernel<false, false, MT, true>), grid, dim3(256), 0, st, p);     \
      }                                                                                                 \
    } else if (dgrad) {                                                                                 \
      if (half) hipLaunchKernelGGL((conv_s2_kernel<true, true, MT>), grid, dim3(256), 0, st, p);        \
      else hipLaunchKernelGGL((conv_s2_kernel<true, false, MT>), grid, dim3(256), 0, st, p);            \
    } else {                                                                                            \
      if (half) hipLaunchKernelGGL((conv_s2_kernel<false, true, MT>), grid, dim3(256), 0, st, p);       \
      else hipLaunchKernelGGL((conv_s2_kernel<false, false, MT>), grid, dim3(256), 0, st, p);           \
    }                                                                                                   \
  } while (0)
  if (mt == 4) SGG_S2(4);
  else SGG_S2(7);
#undef SGG_S2
}
