// Halo-resident 3x3 / stride-1 convolution for the split 16-bit modes (gfx950): forward and dgrad of
//   tf.layers.conv2d(kernel_size=3, strides=1, padding="same")   reference: architectures/generator_with_attention.py:31-57
// (conv1_2, conv2_1..conv2_4, conv3_1, conv3_2 and their Conv2DBackpropInput, train.py:265-266).
//
// The gather kernel (conv_gather.hip) re-stages the shifted input tile for every tap: 9x the global loads, 9x the
// f32 -> 16-bit split VALU work and two barriers per 32-deep slab; it is staging-bound (MFMA pipe 47 % busy).  Here
//   * a workgroup owns NB 8x8 output blocks; per 32-channel chunk the 10x10 input patch (halo included) of each
//     block is loaded, split into two 16-bit planes and written to LDS ONCE; the nine taps read their MFMA A
//     fragments from the resident patch at shifted pixel slots: no re-staging, no barrier inside a chunk;
//   * the weights are pre-arranged as MFMA B fragments (sgg_conv_split_weights_frag: one coalesced 1 KiB
//     buffer_load_b128 per fragment) and go L2 -> registers directly, prefetched one tap ahead: no LDS traffic
//     and no barrier for the B operand.
// Patch layout: plane[pp][block][slot = ry*12 + rx][32 k] with 64-B rows; the 16-B chunk index is XOR-swizzled by
// ((rx>>2)&1) | ((ry&1)<<1): the 16 lanes of one ds_read_b128 phase (two patch rows x 8 pixels) hit 16 distinct
// bank groups for every tap shift (row pitch 12 slots = 768 B aliases rows, the swizzle separates them).
// Out-of-image patch pixels are raw-buffer out-of-range loads -> zeros (TF SAME padding).
#include "split16.h"
#include "conv_halo.h"
#include <type_traits>
#include <stdlib.h>

// -DSGG_HALO_PROFILE (scripts/build_prof_lib.sh): per-wave cycle accounting of the tap loop with explicit waits, summed into
// sgg_halo_prof[] by wave 0 of every workgroup: {total, wait for B fragments (vmcnt), A fragment reads (issue + lgkmcnt),
// MFMA issue, chunk boundary (patch write + barrier), epilogue, waves}.
#ifdef SGG_HALO_PROFILE
__device__ unsigned long long sgg_halo_prof[8];
extern "C" int sgg_halo_prof_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sgg_halo_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(sgg_halo_prof), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#define PROF(...) __VA_ARGS__
#else
#define PROF(...)
#endif

#ifndef SGG_HALO_DB_MAX
#define SGG_HALO_DB_MAX 65536   // two patch buffers when they fit in this many bytes of LDS (a gfx950 workgroup may use up to 160 KB; measured: see DESIGN.md)
#endif
#ifndef SGG_WIDE_STORE
#define SGG_WIDE_STORE 1        // 16-byte output stores through an in-register quad transpose (sgg_common.h); 0: 4-byte stores
#endif
#define HALO_PITCH 12
#define HALO_BLKB (10 * HALO_PITCH * 64)   // bytes of one plane of one block's patch

__device__ __forceinline__ int halo_sw(int ry, int rx) { return ((rx >> 2) & 1) | ((ry & 1) << 1); }

// ONECH: C == 32 (one chunk per tile); otherwise C % 64 == 0 (an even number of chunks): the B-fragment double buffer
// flips once per chunk (nine taps), so chunks are unrolled in pairs - a run-time parity branch between two chunk bodies
// costs ~110 spilled VGPRs at the merge.
// LNP: LN prologue - src is the producing layer's pre-LayerNorm output, normalised + ELU'd while the patch is staged.
// ONE: single-piece mode (precision 1 / 4): one 16-bit plane, one MFMA per product.
// WB: 8x8 blocks per wave.  1: a wave owns one block x WN columns (TM = 2 row tiles); 2: two blocks (TM = 4) - with WGM = 1, WGN = 4
// the four waves of a workgroup take 32 output columns each of the SAME 128 pixels: a B fragment (L2 -> L1 -> registers, 64 B/clk
// per CU) then feeds four row tiles instead of two, the A fragments (LDS, 256 B/clk per CU) are read by all four waves: the two
// operand paths carry 4 KiB / 16 KiB per wave and tap instead of 8 / 8.  A fragments are then double-buffered per k-step
// (a[k-step][tm]) instead of per tap, which keeps them at 64 registers.
template <int NB, int BN, int WGM, int WGN, bool HALF, bool PREFETCH, bool ONECH, bool LNP, bool ONE = false, int WB = 1>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void conv_halo3_kernel(HaloParams p) {
  constexpr int P = ONE ? 1 : 2;
  constexpr int WN = BN / WGN, TM = 2 * WB, TN = WN / 32;
  constexpr int THREADS = 64 * WGM * WGN;       // four waves (two workgroups per CU) or two (four workgroups per CU, one patch buffer)
  static_assert((WGM * WGN == 4 || WGM * WGN == 2) && NB * 64 / WGM == 64 * WB && WN % 32 == 0 && TN >= 1 && (WB == 1 || WB == 2),
                "a wave owns WB 8x8 blocks x WN columns");
  constexpr int PLANEB = NB * HALO_BLKB;
  constexpr int ITEMS = NB * 400;                       // (block, patch pixel, 8-channel group)
  constexpr int NPASS = (ITEMS + THREADS - 1) / THREADS;
  constexpr bool DB = (2 * P * PLANEB <= SGG_HALO_DB_MAX) && THREADS == 256;        // two patch buffers: one barrier per chunk instead of two
  __shared__ __attribute__((aligned(16))) unsigned char lds[(DB ? 2 : 1) * P * PLANEB];
  __shared__ __attribute__((aligned(16))) float lnp_s[LNP ? 1024 : 4];      // gamma[0..511], beta at +512 (C <= 512: host check)

  // ---- persistent workgroup ----------------------------------------------------------------------------------------
  // XCD k (workgroup ids are dealt round-robin to the 8 XCDs) owns a contiguous eighth of the M-tiles (NB blocks each);
  // its p.gx workgroups walk that range interleaved (tile = first + j + i * stride), so the workgroups resident on an XCD
  // always work on a contiguous window of tiles: halo rows and the patch shared by the n-tiles of one pixel range hit
  // in that XCD's L2.
  const int ntiles_n = p.N / BN;
  const int mtiles = (p.nblk + NB - 1) / NB;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int nt = jx % ntiles_n;
  const int tstride = p.gx / ntiles_n;
  const int mt_begin = (int)(((long long)xcd * mtiles) >> 3) + jx / ntiles_n;
  const int mt_end = (int)(((long long)(xcd + 1) * mtiles) >> 3);
  if (mt_begin >= mt_end) return;
  const int n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wblk = (wave / WGN) * WB, wn0 = (wave % WGN) * WN;      // first block of this wave inside the tile

  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, p.w_bytes, 0x00020000);
  int ea = 0, eb = 0;
  if constexpr (HALF) {
    ea = scale_exp_from_amax(*p.amax_src);
    eb = scale_exp_from_amax(*p.amax_w);
  }
  const float sa = ldexpf(1.f, ea);
  PROF(unsigned long long pc_vm = 0, pc_lds = 0, pc_mfma = 0, pc_chunk = 0, pc_epi = 0; const unsigned long long pc_t0 = __builtin_readcyclecounter();)

  // ---- staging plan (static per thread): item -> offset relative to its block's patch origin, border bits, LDS offset ----
  unsigned it_rel[NPASS];
  int it_meta[NPASS];      // bits 0..19 LDS byte offset inside a plane, 20..23 border bits, 24..25 block, 28 valid
#pragma unroll
  for (int j = 0; j < NPASS; ++j) {
    const int it = tid + THREADS * j;
    const int blk = it / 400, r = it % 400;
    const int px = r >> 2, ch8 = r & 3;
    const int ry = px / 10, rx = px % 10;
    it_rel[j] = (unsigned)((ry * p.in_rs + rx * p.in_ps + ch8 * 8) * 4);
    const int bits = (ry == 0) | ((ry == 9) << 1) | ((rx == 0) << 2) | ((rx == 9) << 3);
    it_meta[j] = (blk * HALO_BLKB + (ry * HALO_PITCH + rx) * 64 + ((ch8 ^ halo_sw(ry, rx)) << 4)) | (bits << 20) | ((blk & 3) << 24) |
                 (ch8 << 26) | ((it < ITEMS) << 28);
  }
  if constexpr (LNP) {
    for (int c = tid; c < p.ln_nc; c += THREADS) {      // (ln_nc = C, or 32 for the space-to-depth view: its four chunks are the same channels)
      lnp_s[c] = p.ln_gamma[c];
      lnp_s[512 + c] = p.ln_beta[c];
    }
    __syncthreads();
  }
  // LN prologue state of the patch in flight (between stage_load and stage_write): per block the sample's (mean, rstd), the
  // channel chunk, and which items are padding (must stay a zero activation)
  float ld_mu[NB], ld_rs[NB];
  int ld_cc = 0, ld_bad = 0;

  // block coordinates of the tile being STAGED (uniform: scalar registers): global block row (b*bh + by), by, bx
  int s_grow[NB], s_by[NB], s_bx[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int beta = mt_begin * NB + j;
    s_grow[j] = beta / p.bw;
    s_bx[j] = beta % p.bw;
    s_by[j] = s_grow[j] % p.bh;
  }
  int s_tile = mt_begin, s_cc = 0;
  const int nch = p.C >> 5;
  const int adv_rows = (tstride * NB) / p.bw, adv_cols = (tstride * NB) % p.bw;   // block advance between this workgroup's tiles

  f32x4 pre[NPASS][2];
  // issue the global loads of the next (tile, chunk) patch in flat order; past the last tile: out-of-range offsets (zeros)
  auto stage_load = [&]() __attribute__((always_inline)) {
    unsigned base[NB];
    int bbits[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool dead = (s_tile >= mt_end) | (s_tile * NB + j >= p.nblk);
      base[j] = (unsigned)(((s_grow[j] * 8 - 1) * p.in_rs + (s_bx[j] * 8 - 1) * p.in_ps + (s_cc >> 1) * p.in_cA + (s_cc & 1) * p.in_cB) * 4);
      bbits[j] = dead ? 15 : ((s_by[j] == 0) | ((s_by[j] == p.bh - 1) << 1) | ((s_bx[j] == 0) << 2) | ((s_bx[j] == p.bw - 1) << 3));
      // a dead block: every item masked (bits 0 -> use the valid flag below)
      if (dead) base[j] = SGG_OOB;
    }
    if constexpr (LNP) {
      ld_cc = s_cc;
      ld_bad = 0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        int b = s_grow[j] / p.bh;
        b = b < p.B ? b : p.B - 1;
        ld_mu[j] = p.ln_stats[2 * b];
        ld_rs[j] = p.ln_stats[2 * b + 1];
      }
    }
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const int blk = (it_meta[j] >> 24) & 3;
      unsigned b0 = base[0];
      int bb = bbits[0];
#pragma unroll
      for (int k = 1; k < NB; ++k) {
        b0 = blk == k ? base[k] : b0;
        bb = blk == k ? bbits[k] : bb;
      }
      const bool bad = !((it_meta[j] >> 28) & 1) | ((((it_meta[j] >> 20) & 15) & bb) != 0) | (b0 == SGG_OOB);
      const unsigned off = bad ? SGG_OOB : b0 + it_rel[j];
      if constexpr (LNP) ld_bad |= (int)bad << j;
      const unsigned o0 = LNP ? off : stage_off0(off, p.src_s16);
      pre[j][0] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, o0);
      pre[j][1] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, LNP ? off + 16u : stage_off1(o0, p.src_s16));
    }
    if (++s_cc == nch) {        // advance to this workgroup's next tile
      s_cc = 0;
      s_tile += tstride;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        s_bx[j] += adv_cols;
        s_grow[j] += adv_rows;
        s_by[j] += adv_rows;
        if (s_bx[j] >= p.bw) {
          s_bx[j] -= p.bw;
          ++s_grow[j];
          ++s_by[j];
        }
        while (s_by[j] >= p.bh) s_by[j] -= p.bh;
      }
    }
  };
  auto stage_write = [&](unsigned char* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      if constexpr (LNP) {
        const int blk = (it_meta[j] >> 24) & 3;
        float mu = ld_mu[0], rs = ld_rs[0];
#pragma unroll
        for (int k = 1; k < NB; ++k) {
          mu = blk == k ? ld_mu[k] : mu;
          rs = blk == k ? ld_rs[k] : rs;
        }
        const int cb = ((ld_cc * 32) & (p.ln_nc - 1)) + ((it_meta[j] >> 26) & 3) * 8;
        ln_elu8(pre[j][0], pre[j][1], lnp_s + cb, lnp_s + 512 + cb, mu, rs, (ld_bad >> j) & 1);
      }
      u32x4 pl[P];
      if constexpr (LNP || !HALF) split8<P, HALF>(pre[j][0], pre[j][1], sa, pl);
      else stage_planes<P, HALF>(pre[j][0], pre[j][1], sa, p.src_s16, pl);
      if ((it_meta[j] >> 28) & 1) {
#pragma unroll
        for (int pp = 0; pp < P; ++pp) *reinterpret_cast<u32x4*>(dst + pp * PLANEB + (it_meta[j] & 0xfffff)) = pl[pp];
      }
    }
  };

  // ---- weights: B fragments straight from L2, layout [tap][chunk][n-tile of 32][k-step][plane][lane] x 16 B --------
  const unsigned w_lane = (unsigned)((n0 + wn0) >> 5) * 4096u + (unsigned)lane * 16u;
  const unsigned w_slab = (unsigned)(p.N >> 5) * 4096u;
  u32x4 rb[2][TN][2][P];
  auto load_b = [&](auto par_c, int cc, int tap) {
    constexpr int par = decltype(par_c)::value;
    const unsigned base = (unsigned)(tap * nch + cc) * w_slab + w_lane;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
          rb[par][tn][ks][pp] = __builtin_bit_cast(u32x4, buf_load4(rs_w, base + (unsigned)(tn * 4096 + ks * 2048 + pp * 1024)));
  };

  f32x16 acc[TM][TN];
  acc_zero<TM, TN>(acc);

  const int i = lane & 31, h = lane >> 5;
  const int pyl = i >> 3, pxl = i & 7;
  int cur = 0;                  // patch buffer the current chunk reads (DB)

  // One 32-channel chunk = nine taps, statically unrolled and branch free, so that the compiler's s_waitcnt counts are
  // exact: every tap issues the next tap's eight B-fragment loads and waits only for its own (issued one tap earlier).
  // (With the loads under a uniform branch the counter analysis had to assume the fewest loads in flight and every tap
  // waited for the loads it had just issued: the whole L2 latency exposed per tap.)
  // (Reading the A fragments of tap t+1 between the two k-steps of tap t - software pipelining inside the wave - was
  //  measured: no gain, 32 more VGPRs.)
  // WB == 1: a[tap parity][tm][k-step][plane] (the fragments of tap t+1 are read between the two k-steps of tap t);
  // WB == 2: a[k-step][tm][0][plane] (the fragments of k-step k+1 are read in front of the MFMAs of k-step k)
  constexpr int AKS = WB == 1 ? 2 : 1;
  u32x4 a[2][TM][AKS][P];
  // A fragments of one tap from the resident patch: k-steps [ks0, ks0 + nks) into a[buf][tm][ks - (WB == 1 ? 0 : ks0)]
  auto read_a_ks = [&](auto buf_c, int tap, auto ks0_c, auto nks_c) {
    constexpr int buf = decltype(buf_c)::value, ks0 = decltype(ks0_c)::value, nks = decltype(nks_c)::value;
    const int kh = tap / 3, kw = tap % 3;
    const int dyy = p.flip ? 2 - kh : kh, dxx = p.flip ? 2 - kw : kw;
    // (opaque to the optimiser: otherwise the 36 per-tap LDS addresses are hoisted out of the chunk loop and spilled,
    //  and each scratch reload drags a vmcnt wait for the B prefetch in flight)
    int pyv = pyl, pxv = pxl;
    asm volatile("" : "+v"(pyv), "+v"(pxv));
    const unsigned char* patch_w = lds + (DB ? cur * (P * PLANEB) : 0) + wblk * HALO_BLKB;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int ry = (tm & 1) * 4 + pyv + dyy, rx = pxv + dxx;           // (tm >> 1: which of the wave's blocks)
      const unsigned char* row = patch_w + (tm >> 1) * HALO_BLKB + (ry * HALO_PITCH + rx) * 64;
      const int hs = halo_sw(ry, rx);
#pragma unroll
      for (int ks = ks0; ks < ks0 + nks; ++ks)
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
          a[buf][tm][WB == 1 ? ks : 0][pp] = *reinterpret_cast<const u32x4*>(row + pp * PLANEB + (((2 * ks + h) ^ hs) << 4));
    }
  };
  auto read_a = [&](auto buf_c, int tap) { read_a_ks(buf_c, tap, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}); };
  auto mma_kstep = [&](auto par_c, auto ks_c) {
    constexpr int par = decltype(par_c)::value, ks = decltype(ks_c)::value;
    constexpr int ab = WB == 1 ? par : ks, ak = WB == 1 ? ks : 0;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        f32x16 d = acc[tm][tn];
        if constexpr (P == 2) {
          d = mfma16<HALF>(a[ab][tm][ak][1], rb[par][tn][ks][0], d);
          d = mfma16<HALF>(a[ab][tm][ak][0], rb[par][tn][ks][1], d);
        }
        d = mfma16<HALF>(a[ab][tm][ak][0], rb[par][tn][ks][0], d);
        acc[tm][tn] = d;
      }
  };
  auto tap_body = [&](auto par_c, auto tap_c, int cc) {
    constexpr int par = decltype(par_c)::value, tap = decltype(tap_c)::value;
    const int ncc = (cc + 1 == nch) ? 0 : cc + 1;         // (the chunk after the last one re-reads valid weights)
#ifndef SGG_ABL_NOB
    load_b(std::integral_constant<int, par ^ 1>{}, tap == 8 ? ncc : cc, tap == 8 ? 0 : tap + 1);
#endif
#ifndef SGG_ABL_NOSTAGE
    if constexpr (PREFETCH && tap == 6) stage_load();
#endif
    __builtin_amdgcn_sched_barrier(0);
    // DB: the next chunk's patch (loads issued at tap 6) is split and written to the other buffer inside the last tap's
    // scheduling region, so its ~170 VALU instructions issue in the shadow of this tap's MFMAs
#ifndef SGG_ABL_NOSTAGE
    if constexpr (DB && PREFETCH && tap == 8) stage_write(lds + (cur ^ 1) * (P * PLANEB));
#endif
#ifdef SGG_HALO_PROFILE
    const unsigned long long q0 = __builtin_readcyclecounter();
    if constexpr (PREFETCH && (tap == 6 || tap == 7)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    const unsigned long long q1 = __builtin_readcyclecounter();
    pc_vm += q1 - q0;
    __builtin_amdgcn_sched_barrier(0);
#endif
    // A fragments of tap t+1 are read between the two k-steps of tap t (the in-kernel profile shows 11 % of a wave's time
    // in issue + wait of these reads when they sit in front of the MFMAs)
    PROF(const unsigned long long q2 = __builtin_readcyclecounter();)
    if constexpr (WB == 2) {
      // k-step granular: [read k-step 1 of this tap] MFMAs of k-step 0 [read k-step 0 of the next tap] MFMAs of k-step 1
#ifndef SGG_ABL_NOA
      read_a_ks(std::integral_constant<int, 1>{}, tap, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
#endif
      __builtin_amdgcn_sched_barrier(0);
      SGG_PRIO_HI();
      mma_kstep(par_c, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
#ifndef SGG_ABL_NOA
      if constexpr (tap < 8) read_a_ks(std::integral_constant<int, 0>{}, tap + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
#endif
      __builtin_amdgcn_sched_barrier(0);
      mma_kstep(par_c, std::integral_constant<int, 1>{});
      SGG_PRIO_LO();
      __builtin_amdgcn_sched_barrier(0);
    } else {
    SGG_PRIO_HI();
    mma_kstep(par_c, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    PROF(const unsigned long long q3 = __builtin_readcyclecounter();)
#ifndef SGG_ABL_NOA
    if constexpr (tap < 8) read_a(std::integral_constant<int, par ^ 1>{}, tap + 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    PROF(const unsigned long long q4 = __builtin_readcyclecounter(); pc_lds += q4 - q3;)
    mma_kstep(par_c, std::integral_constant<int, 1>{});
    SGG_PRIO_LO();
    __builtin_amdgcn_sched_barrier(0);
    PROF(pc_mfma += (__builtin_readcyclecounter() - q4) + (q3 - q2);)
    }
  };
  auto chunk = [&](auto par0_c, int cc) {
    constexpr int par0 = decltype(par0_c)::value;
#ifndef SGG_ABL_NOA
    if constexpr (WB == 2) read_a_ks(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    else read_a(par0_c, 0);
#endif
#define SGG_TAP(T) tap_body(std::integral_constant<int, (par0 + T) & 1>{}, std::integral_constant<int, T>{}, cc)
    SGG_TAP(0); SGG_TAP(1); SGG_TAP(2); SGG_TAP(3); SGG_TAP(4); SGG_TAP(5); SGG_TAP(6); SGG_TAP(7); SGG_TAP(8);
#undef SGG_TAP
    // next chunk (of this tile or the first of the next tile): replace the resident patch
    PROF(const unsigned long long qc = __builtin_readcyclecounter();)
    if constexpr (DB) {
      if constexpr (!PREFETCH) stage_write(lds + (cur ^ 1) * (P * PLANEB));   // nobody reads the other buffer since the previous barrier
#ifndef SGG_ABL_NOSTAGE
      cur ^= 1;
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (!PREFETCH) stage_load();
      stage_write(lds);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    PROF(pc_chunk += __builtin_readcyclecounter() - qc;)
  };

  // this wave's output block: global block row and column, advanced by NB per tile
  int o_grow[WB], o_bx[WB];
#pragma unroll
  for (int wb = 0; wb < WB; ++wb) {
    const int beta = mt_begin * NB + wblk + wb;
    o_grow[wb] = beta / p.bw;
    o_bx[wb] = beta % p.bw;
  }
  // per-lane byte offset inside a block.  After the quad transpose of the accumulators (sgg_quad_transpose4) lane (h, g, k) =
  // (lane >> 5, (lane & 31) >> 2, lane & 3) holds pixel column 4h + k of a block row and output channels 4g .. 4g+3 of its 32-column
  // group: one 16-byte store per four accumulator registers (SGG_WIDE_STORE 0: the accumulators as they stand, 4 bytes per lane)
#if SGG_WIDE_STORE
  const unsigned o_lane_b = (unsigned)(((4 * h + (lane & 3)) * p.out_ps) + ((lane & 31) >> 2) * 4) * 4u;
#else
  const unsigned o_lane_b = (unsigned)((4 * h * p.out_ps) + (lane & 31)) * 4u;
#endif
  int o_goff[TN];                                                                   // float offset of this wave's 32-column groups
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int gi = ((n0 + wn0) >> 5) + tn;
    o_goff[tn] = (gi >> 1) * p.out_nA + (gi & 1) * p.out_nB;
  }
  const float us_a = ldexpf(1.f, -ea), us_b = ldexpf(1.f, -eb);
  // (loaded once: a bias load inside the tile epilogue would wait (vmcnt(0)) for every prefetch in flight)
  float bias_v[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) bias_v[tn] = p.bias ? p.bias[n0 + wn0 + tn * 32 + acc_col(lane)] : 0.f;
  const int wn = p.out_rs;

  stage_load();
  load_b(std::integral_constant<int, 0>{}, 0, 0);
  stage_write(lds);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // Timing-only ablation builds (scripts/build_variant_lib.sh <name> -DSGG_ABL_...; results are WRONG): the tap loop without its
  // B-fragment loads (NOB), its A-fragment reads (NOA), its patch staging (NOSTAGE) or the tile epilogue's stores / statistics
  // (NOEPI) - the operands are fetched once here and stay in registers, so the loop's MFMA stream is unchanged.
#ifdef SGG_ABL_NOB
  load_b(std::integral_constant<int, 1>{}, 0, 1);
#endif
#ifdef SGG_ABL_NOA
  read_a(std::integral_constant<int, 0>{}, 0);
  read_a(std::integral_constant<int, 1>{}, 1);
#endif

  auto epilogue = [&](int tile) {
    PROF(const unsigned long long qe = __builtin_readcyclecounter();)
    // ---- tile epilogue: unscale, + bias, store; optionally this wave's LayerNorm partial statistics; clear ----------
    // (per block of the wave: wb = 0 .. WB-1, row tiles 2 wb and 2 wb + 1)
#pragma unroll
    for (int wb = 0; wb < WB; ++wb) {
      const int beta = tile * NB + wblk + wb;
      const bool live = beta < p.nblk;
      // uniform (scalar) pointer to the block's first pixel; per store: scalar row/pixel offset + one per-lane byte offset
      const char* ob = reinterpret_cast<const char*>(p.out + (size_t)(o_grow[wb] * 8) * wn + (size_t)(o_bx[wb] * 8) * p.out_ps);
      float lsum = 0.f;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            // (two exact power-of-two factors: |ea|, |eb| <= 100 keeps each one a normal float)
            const int tm = 2 * wb + t2;
            const float v = HALF ? fmaf(acc[tm][tn][r] * us_a, us_b, bias_v[tn]) : acc[tm][tn][r] + bias_v[tn];
            acc[tm][tn][r] = v;
            lsum += v;
          }
#if defined(SGG_ABL_NOEPI) || defined(SGG_ABL_NOSTORE)
      if (p.B < 0)       // (never true: keeps the code, skips the stores and the statistics)
#else
      if (live)          // one uniform branch around all stores (a branch per store costs ~64 jumps per tile)
#endif
      {
#if SGG_WIDE_STORE
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int tm = 2 * wb + t2;
              float v0 = acc[tm][tn][4 * q], v1 = acc[tm][tn][4 * q + 1], v2 = acc[tm][tn][4 * q + 2], v3 = acc[tm][tn][4 * q + 3];
              sgg_quad_transpose4(v0, v1, v2, v3, lane);
              const size_t so = ((size_t)(t2 * 4 + q) * wn + o_goff[tn]) * sizeof(float);   // scalar: block row 4 t2 + q
              sgg_out_store4(reinterpret_cast<float*>(const_cast<char*>(ob) + so + o_lane_b), f32x4{v0, v1, v2, v3});
            }
#else
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const size_t so = ((size_t)(t2 * 4 + (r >> 2)) * wn + (size_t)(r & 3) * p.out_ps + o_goff[tn]) * sizeof(float);   // scalar
              sgg_out_store(reinterpret_cast<float*>(const_cast<char*>(ob) + so + o_lane_b), acc[2 * wb + t2][tn][r]);
            }
#endif
      }
#if defined(SGG_ABL_NOEPI) || defined(SGG_ABL_NOSTATS)
      if (p.B < 0) {
#else
      if (p.tile_stats) {
#endif
        // (count, mean, M2) of this wave's 64 pixels x WN channels (one 8x8 block: inside one sample); merged per sample
        // with Chan's formula by ln_apply_elu_kernel
        const float mean_w = wave_sum(lsum) * (1.f / (float)(64 * WN));
        float q = 0.f, dm = 0.f;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = acc[2 * wb + t2][tn][r] - mean_w;
              q += d * d;
              dm = fmaxf(dm, fabsf(d));
            }
        q = wave_sum(q);
        dm = wave_max(dm);
        if (lane == 0 && live) {
          float* o = p.tile_stats + ((size_t)beta * (p.N / WN) + (n0 + wn0) / WN) * SGG_TS;
          o[0] = (float)(64 * WN);
          o[1] = mean_w;
          o[2] = q;
          o[3] = dm;
        }
      }
      o_bx[wb] += adv_cols;
      o_grow[wb] += adv_rows;
      if (o_bx[wb] >= p.bw) { o_bx[wb] -= p.bw; ++o_grow[wb]; }
    }
    acc_zero<TM, TN>(acc);
    PROF(pc_epi += __builtin_readcyclecounter() - qe;)
  };

  if constexpr (ONECH) {
    for (int tile = mt_begin; tile < mt_end; tile += 2 * tstride) {
      if constexpr (!PREFETCH && DB) stage_load();
      chunk(std::integral_constant<int, 0>{}, 0);
      epilogue(tile);
      if (tile + tstride < mt_end) {
        if constexpr (!PREFETCH && DB) stage_load();
        chunk(std::integral_constant<int, 1>{}, 0);
        epilogue(tile + tstride);
      }
    }
  } else {
    for (int tile = mt_begin; tile < mt_end; tile += tstride) {
      for (int cc = 0; cc < nch; cc += 2) {
        if constexpr (!PREFETCH && DB) stage_load();
        chunk(std::integral_constant<int, 0>{}, cc);
        if constexpr (!PREFETCH && DB) stage_load();
        chunk(std::integral_constant<int, 1>{}, cc + 1);
      }
      epilogue(tile);
    }
  }
#ifdef SGG_HALO_PROFILE
  if (tid == 0) {
    atomicAdd(&sgg_halo_prof[0], __builtin_readcyclecounter() - pc_t0);
    atomicAdd(&sgg_halo_prof[1], pc_vm);
    atomicAdd(&sgg_halo_prof[2], pc_lds);
    atomicAdd(&sgg_halo_prof[3], pc_mfma);
    atomicAdd(&sgg_halo_prof[4], pc_chunk);
    atomicAdd(&sgg_halo_prof[5], pc_epi);
    atomicAdd(&sgg_halo_prof[6], 1ull);
  }
#endif
}

// f32 [taps][N][C] -> two 16-bit planes in MFMA B-fragment order [tap][C/32][N/32][k-step][plane][lane] x 16 B:
// lane l of fragment (tap, chunk, n-tile, k-step) holds w[tap][n-tile*32 + (l&31)][chunk*32 + k-step*16 + 8*(l>>5) .. +8]
template <bool HALF>
__global__ void split_weights_frag_kernel(const float* __restrict__ in, u32x4* __restrict__ out, int taps, int N, int C,
                                          const float* __restrict__ amax) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nch = C >> 5, ntl = N >> 5;
  if (idx >= (long long)taps * nch * ntl * 128) return;
  const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 1);
  long long r = idx >> 7;
  const int ntile = (int)(r % ntl); r /= ntl;
  const int cc = (int)(r % nch);
  const int tap = (int)(r / nch);
  const int n = ntile * 32 + (lane & 31), k = cc * 32 + ks * 16 + 8 * (lane >> 5);
  const float* src = in + ((size_t)tap * N + n) * C + k;
  const float scale = HALF ? ldexpf(1.f, scale_exp_from_amax(*amax)) : 1.f;
  u32x4 pl[2];
  split8<2, HALF>(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), scale, pl);
  const size_t o = ((((size_t)(tap * nch + cc) * ntl + ntile) * 2 + ks) * 2) * 64 + lane;
  out[o] = pl[0];
  out[o + 64] = pl[1];
}

// 5x5 stride-2 HWIO kernel [5][5][Ci][Co] -> the 3x3 kernel of the same convolution over the space-to-depth view of x:
// out[u][v][(qy, qx, ci)][co] = w[2u + qy - 1][2v + qx - 1][ci][co], zero where that tap does not exist (11 of the 36 (u, qy) x (v, qx)
// combinations).  SAME padding (1, 2) of the stride-2 convolution on an even grid = padding (1, 1) of the 3x3 one.
__global__ void s2d_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int Ci, int Co) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = 9LL * 4 * Ci * Co;
  if (idx >= n) return;
  const int co = (int)(idx % Co);
  long long r = idx / Co;
  const int ci = (int)(r % Ci); r /= Ci;
  const int q = (int)(r % 4); r /= 4;
  const int v = (int)(r % 3), u = (int)(r / 3);
  const int kh = 2 * u + (q >> 1) - 1, kw = 2 * v + (q & 1) - 1;
  out[idx] = (kh >= 0 && kh < 5 && kw >= 0 && kw < 5) ? w[(((size_t)kh * 5 + kw) * Ci + ci) * Co + co] : 0.f;
}
extern "C" int sgg_conv_s2d_weights(const float* w5, float* w3, int Cin, int Cout, void* stream) {
  SGG_CHECK_ARG(w5 && w3 && Cin > 0 && Cout > 0, "sgg_conv_s2d_weights: bad argument");
  const long long n = 9LL * 4 * Cin * Cout;
  hipLaunchKernelGGL(s2d_weights_kernel, dim3((unsigned)sgg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w5, w3, Cin, Cout);
  SGG_LAUNCH_CHECK("sgg_conv_s2d_weights");
  return SGG_OK;
}

// 1 if a 5x5 stride-2 convolution runs as a 3x3 convolution over the space-to-depth view (weights: sgg_conv_s2d_weights, then
// sgg_conv_split_weights_frag with 9 taps; w_split_layout 3).  Forward: Cin = 32 (a chunk = one pixel parity), any Cout the 3x3
// kernel takes; dgrad (arguments swapped like sgg_conv_wsplit_layout's): contraction over Cout, 4*Cin = 128 virtual outputs.
int sgg_s2d_applicable(int KH, int KW, int stride, int Hi, int Wi, int Cin, int Cout, int precision) {
  return KH == 5 && KW == 5 && stride == 2 && Hi > 0 && Wi > 0 && Hi % 16 == 0 && Wi % 16 == 0 && Cin == 32 && Cout == 32 &&
         sgg_prec_resident(precision);
}

// ---- host ---------------------------------------------------------------------------------------------------
int sgg_halo_applicable(int KH, int KW, int stride, int H, int W, int C, int N, int precision) {
  return KH == 3 && KW == 3 && stride == 1 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && (C == 32 || C % 64 == 0) && N % 32 == 0 &&
         sgg_prec_resident(precision);
}

// 64-column layers run on two-block workgroups (2 x 2 waves of 64 pixels x 32 columns, two patch buffers, prefetch); -DSGG_HALO_N64_NB2=0:
// four-block ones (4 waves of 64 x 64, one buffer, no prefetch: 0.1 ms per step slower)
// 32-column layers run on two-wave workgroups (two blocks, one 30 KB patch buffer, four workgroups per CU: four independent phases
// instead of two hide each other's chunk boundaries and store drains); -DSGG_HALO_N32_W2=0: four-wave workgroups of four blocks
// (0.1 ms per step slower)
#ifndef SGG_HALO_N32_W2
#define SGG_HALO_N32_W2 1
#endif
#ifndef SGG_HALO_N64_NB2
#define SGG_HALO_N64_NB2 1
#endif
// 128-column layers: -DSGG_HALO_N128_WB2=1 = four waves of (two blocks x 32 columns) instead of 2 x 2 waves of (one block x 64 columns)
#ifndef SGG_HALO_N128_WB2
#define SGG_HALO_N128_WB2 0
#endif
int sgg_halo_stats_cols(int N) {
  if (N % 128 == 0) return SGG_HALO_N128_WB2 ? 32 : 64;
  return (N % (SGG_HALO_N64_NB2 ? 128 : 64) == 0) ? 64 : 32;
}

void sgg_halo_launch(const HaloParams& p_, int precision, hipStream_t st) {
  if (p_.frag16) {       // w_split_layout 4 (128-column tiles, two-piece modes): producer / consumer workgroups, K = 32 MFMA shape
    sgg_halo_pc_launch(p_, precision, st);
    return;
  }
  HaloParams p = p_;
  const bool half = sgg_prec_half(precision), one = sgg_prec_one(precision);   // (the LN prologue exists in the two-piece modes only: callers check)
#define SGG_HALO(NB, BN, WGM, WGN, PF, WB)                                                                   \
  do {                                                                                                       \
    const int mtiles = sgg_cdiv(p.nblk, NB), ntn = p.N / BN;                                                 \
    int per_xcd = sgg_cdiv(mtiles, 8) * ntn;       /* (tile, n-tile) pairs an XCD owns */                    \
    const int cap = sgg_persist_cus(p.cu_cap) * 8 / (WGM * WGN);   /* eight resident waves on each of its (32) CUs */   \
    int gx = per_xcd < cap ? per_xcd : cap;                                                                  \
    gx = sgg_cdiv(gx, ntn) * ntn;                                                                            \
    p.gx = gx;                                                                                               \
    const dim3 grid((unsigned)(8 * gx));                                                                     \
    const dim3 blk(64 * WGM * WGN);                                                                          \
    if (one) {                                                                                                                          \
      if (half && p.C == 32) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, true, PF, true, false, true, WB>), grid, blk, 0, st, p);   \
      else if (half) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, true, PF, false, false, true, WB>), grid, blk, 0, st, p);      \
      else if (p.C == 32) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, false, PF, true, false, true, WB>), grid, blk, 0, st, p); \
      else hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, false, PF, false, false, true, WB>), grid, blk, 0, st, p);               \
    } else if (p.ln_stats) {                                                                                                            \
      if (half && p.C == 32) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, true, PF, true, true, false, WB>), grid, blk, 0, st, p);   \
      else if (half) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, true, PF, false, true, false, WB>), grid, blk, 0, st, p);      \
      else if (p.C == 32) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, false, PF, true, true, false, WB>), grid, blk, 0, st, p); \
      else hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, false, PF, false, true, false, WB>), grid, blk, 0, st, p);               \
    } else if (half && p.C == 32) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, true, PF, true, false, false, WB>), grid, blk, 0, st, p); \
    else if (half) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, true, PF, false, false, false, WB>), grid, blk, 0, st, p);       \
    else if (p.C == 32) hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, false, PF, true, false, false, WB>), grid, blk, 0, st, p);  \
    else hipLaunchKernelGGL((conv_halo3_kernel<NB, BN, WGM, WGN, false, PF, false, false, false, WB>), grid, blk, 0, st, p);                \
  } while (0)
  if (p.N % 128 == 0 && SGG_HALO_N128_WB2) SGG_HALO(2, 128, 1, 4, true, 2);      // four waves x (two blocks x 32 columns)
  else if (p.N % 128 == 0) SGG_HALO(2, 128, 2, 2, true, 1);
  else if (p.N % 64 == 0 && SGG_HALO_N64_NB2) SGG_HALO(2, 64, 2, 2, true, 1);
  else if (p.N % 64 == 0) SGG_HALO(4, 64, 4, 1, false, 1);      // (with PREFETCH: 35 spilled VGPRs at the 256-register budget of two waves per SIMD)
  else if (SGG_HALO_N32_W2) SGG_HALO(2, 32, 2, 1, true, 1);
  else SGG_HALO(4, 32, 4, 1, true, 1);
#undef SGG_HALO
}

// Which operand format sgg_conv2d_nhwc_fwd / _dgrad want for the pre-split weights of this convolution:
// 0 = planes [P][taps*N*C] (sgg_conv_split_weights), 1 = MFMA fragment order (sgg_conv_split_weights_frag; the
// halo-resident 3x3 stride-1 kernel).  H, W: the (identical) input and output grid of a stride-1 convolution.
// 2 = MFMA fragment order with 25 taps for the band-resident 5x5 stride-2 kernel (conv_s2.hip); H, W: the full-resolution grid
// (forward input / dgrad output); for the dgrad direction pass (Cin, Cout) swapped, as for layout 1.
// 3 = 5x5 stride 2 over 32 -> 32 channels as a 3x3 convolution over the space-to-depth view: the 9-tap kernel of
// sgg_conv_s2d_weights ([3][3][128][32]) in fragment order (forward: its HWOI transpose with N = 32, C = 128; dgrad: N = 128, C = 32).
// 4 = layout 1's layers that run on the producer / consumer kernel (conv_halo_pc.hip: 128-column tiles, Cin % 64 == 0, precision 2 / 3):
// the fragments of the K = 32 MFMA shape, sgg_conv_split_weights_frag16.
extern "C" int sgg_conv_wsplit_layout(int KH, int KW, int stride, int H, int W, int Cin, int Cout, int precision) {
  if (sgg_halo_applicable(KH, KW, stride, H, W, Cin, Cout, precision)) return sgg_halo_pc_applicable(Cin, Cout, precision) ? 4 : 1;
  if (sgg_s2_applicable(KH, KW, stride, 1, H, W, Cin, Cout, precision)) return 2;
  if (sgg_s2d_applicable(KH, KW, stride, H, W, Cin, Cout, precision)) return 3;
  return 0;
}

// The same question for a launch whose SOURCE operand (x of the forward, dy of the dgrad) will arrive PRE-SPLIT (operand_format 1):
// the layers with 64 output columns (Cout % 64 == 0, not 128; Cin % 64 == 0; precision 2) then also run on the producer / consumer
// kernel - its four-block form, which stages the patch by LDS-DMA only (conv_halo_pc.hip) -, so 4 is returned for them as well.
extern "C" int sgg_conv_wsplit_layout_presplit(int KH, int KW, int stride, int H, int W, int Cin, int Cout, int precision) {
  if (sgg_halo_applicable(KH, KW, stride, H, W, Cin, Cout, precision) && sgg_halo_pc64_applicable(Cin, Cout, precision)) return 4;
  return sgg_conv_wsplit_layout(KH, KW, stride, H, W, Cin, Cout, precision);
}

// f32 [taps][N][C] -> two 16-bit planes in the B-fragment order of v_mfma_f32_16x16x32: [tap][C/32][N/16][plane][lane] x 16 B,
// lane l of fragment (tap, chunk, n-tile, plane) holds w[tap][n-tile*16 + (l&15)][chunk*32 + 8*(l>>4) .. +8]  (w_split_layout 4)
template <bool HALF>
__global__ void split_weights_frag16_kernel(const float* __restrict__ in, u32x4* __restrict__ out, int taps, int N, int C,
                                            const float* __restrict__ amax) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nch = C >> 5, ntl = N >> 4;
  if (idx >= (long long)taps * nch * ntl * 64) return;
  const int lane = (int)(idx & 63);
  long long r = idx >> 6;
  const int ntile = (int)(r % ntl); r /= ntl;
  const int cc = (int)(r % nch);
  const int tap = (int)(r / nch);
  const int n = ntile * 16 + (lane & 15), k = cc * 32 + 8 * (lane >> 4);
  const float* src = in + ((size_t)tap * N + n) * C + k;
  const float scale = HALF ? ldexpf(1.f, scale_exp_from_amax(*amax)) : 1.f;
  u32x4 pl[2];
  split8<2, HALF>(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), scale, pl);
  const size_t o = (((size_t)(tap * nch + cc) * ntl + ntile) * 2) * 64 + lane;
  out[o] = pl[0];
  out[o + 64] = pl[1];
}
extern "C" int sgg_conv_split_weights_frag16(const float* in, void* out, int taps, int N, int C, int precision, const float* amax,
                                             void* stream) {
  SGG_CHECK_ARG(in && out && taps > 0 && N > 0 && C > 0 && N % 16 == 0 && C % 32 == 0 && (precision == 2 || precision == 3) &&
                    (precision != 2 || amax),
                "sgg_conv_split_weights_frag16: bad argument");
  const long long n = (long long)taps * (C / 32) * (N / 16) * 64;
  const dim3 grid(sgg_cdiv(n, 256)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (precision == 2) hipLaunchKernelGGL(split_weights_frag16_kernel<true>, grid, blk, 0, st, in, (u32x4*)out, taps, N, C, amax);
  else hipLaunchKernelGGL(split_weights_frag16_kernel<false>, grid, blk, 0, st, in, (u32x4*)out, taps, N, C, amax);
  SGG_LAUNCH_CHECK("sgg_conv_split_weights_frag16");
  return SGG_OK;
}

// in: f32 [taps][N][C] (forward: the HWOI transpose, N = Cout, C = Cin; dgrad: the HWIO kernel, N = Cin, C = Cout)
// out: taps*N*C*4 bytes.  precision 2 needs `amax` (device word with max|w|).
extern "C" int sgg_conv_split_weights_frag(const float* in, void* out, int taps, int N, int C, int precision, const float* amax,
                                           void* stream) {
  // (the single-piece modes 1 / 4 read plane 0 of the same fragments: the first piece of the split IS the rounded operand)
  precision = sgg_prec_general(precision);
  SGG_CHECK_ARG(in && out && taps > 0 && N > 0 && C > 0 && N % 32 == 0 && C % 32 == 0 && (precision == 2 || precision == 3) &&
                    (precision != 2 || amax),
                "sgg_conv_split_weights_frag: bad argument");
  const long long n = (long long)taps * (C / 32) * (N / 32) * 128;
  const dim3 grid(sgg_cdiv(n, 256)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (precision == 2) hipLaunchKernelGGL(split_weights_frag_kernel<true>, grid, blk, 0, st, in, (u32x4*)out, taps, N, C, amax);
  else hipLaunchKernelGGL(split_weights_frag_kernel<false>, grid, blk, 0, st, in, (u32x4*)out, taps, N, C, amax);
  SGG_LAUNCH_CHECK("sgg_conv_split_weights_frag");
  return SGG_OK;
}
