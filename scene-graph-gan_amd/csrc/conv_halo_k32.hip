// The four-wave halo-resident 3x3 / stride-1 kernel of conv_halo.hip on the K = 32 MFMA shape (gfx950), 128-column tiles, two-piece modes:
//   tf.layers.conv2d(kernel_size=3, strides=1, padding="same")   reference: architectures/generator_with_attention.py:31-57
// for the launches that apply the producing layer's LayerNorm + ELU while staging (LN prologue, generator_with_attention.py:30..56).
//
// conv_halo_pc.hip (producer / consumer waves) wins on the 128-column launches WITHOUT the prologue and loses with it: its four
// producer waves carry the prologue's arithmetic beside ONE MFMA wave per SIMD.  Here every wave does everything, two workgroups per
// CU, as in conv_halo3_kernel - eight waves share the prologue and hide it behind each other's MFMAs - but the MFMAs are
// v_mfma_f32_16x16x32_{f16,bf16} (a whole 32-channel chunk per instruction, 4 x 4 tiles of 16 pixels x 16 columns per wave) instead of
// 32x32x16: the same FLOPs per cycle, a higher clock on the power-limited chip (profiles/r03_halo_pc_mfma_shape.log).  Weights in the
// fragment order of that shape (w_split_layout 4, sgg_conv_split_weights_frag16), patch swizzle as in conv_halo_pc.hip (conflict free
// for the shape's 16-lane read groups).
//
// Per workgroup: 2 blocks x 128 columns; wave w: block w >> 1, column half (w & 1) * 64.  Per 32-channel chunk: nine statically
// unrolled taps; tap t issues tap t + 1's eight weight-fragment loads (L2 -> registers) and alternates the MFMAs of two row tiles with
// the LDS reads of tap t + 1's A fragments for the same two row tiles (A is single-buffered: 256 VGPRs hold two waves per SIMD).  The next chunk's
// patch is staged in two halves (loads at taps 1 / 5, split / normalised / written to the other LDS buffer at taps 4 / 8); one barrier
// per chunk.
#include "split16.h"
#include "conv_halo.h"
#include <type_traits>

#define K32_PITCH 12
#define K32_BLKB (10 * K32_PITCH * 64)      // bytes of one plane of one block's patch
#define K32_NB 2
#define K32_PLANEB (K32_NB * K32_BLKB)
#define K32_P 2
#define K32_PATCHB (K32_P * K32_PLANEB)
#define K32_ITEMS (K32_NB * 400)
#define K32_NPASS 4                         // 800 (block, patch pixel, 8-channel group) items over 256 threads

// (conv_halo_pc.hip: pc_sw) XOR swizzle of the 16-byte chunk index inside a pixel's 64-byte row
__device__ __forceinline__ int k32_sw(int ry, int rx) { return ((rx ^ (ry >> 1)) & 1) | ((ry & 1) << 1); }

template <bool HALF, bool LNP>
__global__ __launch_bounds__(256, 2) void conv_halo3_k32_kernel(HaloParams p) {
  // (ONE __shared__ object: cdna_hip_programming.md on a second array beside a staging buffer)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * K32_PATCHB + (LNP ? 4096 : 0)];
  float* const lnp_s = reinterpret_cast<float*>(lds + 2 * K32_PATCHB);      // gamma[0..511], beta at +512 (C <= 512: host check)

  // ---- persistent workgroup: as conv_halo3_kernel (XCD k owns a contiguous eighth of the M-tiles, its workgroups walk it interleaved)
  const int ntiles_n = p.N / 128;
  const int mtiles = (p.nblk + K32_NB - 1) / K32_NB;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int nt = jx % ntiles_n;
  const int tstride = p.gx / ntiles_n;
  const int mt_begin = (int)(((long long)xcd * mtiles) >> 3) + jx / ntiles_n;
  const int mt_end = (int)(((long long)(xcd + 1) * mtiles) >> 3);
  if (mt_begin >= mt_end) return;
  const int n0 = nt * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wblk = wave >> 1, wn0 = (wave & 1) * 64;
  const int nch = p.C >> 5;
  const int adv_rows = (tstride * K32_NB) / p.bw, adv_cols = (tstride * K32_NB) % p.bw;   // block advance between this workgroup's tiles

  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, p.w_bytes, 0x00020000);
  int ea = 0, eb = 0;
  if constexpr (HALF) {
    ea = scale_exp_from_amax(*p.amax_src);
    eb = scale_exp_from_amax(*p.amax_w);
  }
  const float sa = ldexpf(1.f, ea);

  // ---- staging plan: item (block, patch pixel, 8-channel group) it = tid + 256 j, j = 0 .. 3 -> offset relative to its block's patch
  // origin, border bits, LDS offset.  Derived from an opaque copy of the thread id wherever it is used (stage_load / stage_write):
  // as loop invariants these eight registers are spilled, and a scratch reload in front of a patch load waits (vmcnt is in order) for
  // every load issued before it.  Items are staged in two halves (j = 0, 1 and j = 2, 3) so that 16, not 32, registers hold a patch
  // in flight.
  auto item_plan = [&](int j, unsigned& rel, int& meta) __attribute__((always_inline)) {
    int t = tid;
    asm volatile("" : "+v"(t));
    const int it = t + 256 * j;
    const int blk = it >= 400 ? 1 : 0, r = it - 400 * blk;      // (it < 1024: blocks 0, 1 and the invalid tail 800 .. 1023 in "block 1")
    const int px = r >> 2, ch8 = r & 3;
    const int ry = (px * 205) >> 11, rx = px - 10 * ry;         // px / 10 for px < 1029
    rel = (unsigned)((ry * p.in_rs + rx * p.in_ps + ch8 * 8) * 4);
    const int bits = (ry == 0) | ((ry == 9) << 1) | ((rx == 0) << 2) | ((rx == 9) << 3);
    meta = (blk * K32_BLKB + (ry * K32_PITCH + rx) * 64 + ((ch8 ^ k32_sw(ry, rx)) << 4)) | (bits << 20) | (blk << 24) |
           (ch8 << 26) | ((it < K32_ITEMS) << 28);
  };
  if constexpr (LNP) {
    for (int c = tid; c < p.ln_nc; c += 256) {
      lnp_s[c] = p.ln_gamma[c];
      lnp_s[512 + c] = p.ln_beta[c];
    }
    __syncthreads();
  }
  // LN prologue state of the patch in flight (between stage_load and stage_write)
  float ld_mu[K32_NB], ld_rs[K32_NB];
  int ld_cc = 0, ld_bad = 0;
  int s_grow[K32_NB], s_by[K32_NB], s_bx[K32_NB];
#pragma unroll
  for (int j = 0; j < K32_NB; ++j) {
    const int beta = mt_begin * K32_NB + j;
    s_grow[j] = beta / p.bw;
    s_bx[j] = beta % p.bw;
    s_by[j] = s_grow[j] % p.bh;
  }
  int s_tile = mt_begin, s_cc = 0;
  f32x4 pre[2][2];
  unsigned base_s[K32_NB];     // (uniform) per block: byte offset of the patch origin of the chunk being staged, its border bits
  int bbits_s[K32_NB];
  // first half: fix the (tile, chunk) being staged, issue the loads of items j = 0, 1; second half: items j = 2, 3, then advance
  auto stage_load = [&](auto half_c) __attribute__((always_inline)) {
    constexpr int half = decltype(half_c)::value;
    if constexpr (half == 0) {
#pragma unroll
      for (int j = 0; j < K32_NB; ++j) {
        const bool dead = (s_tile >= mt_end) | (s_tile * K32_NB + j >= p.nblk);
        base_s[j] = (unsigned)(((s_grow[j] * 8 - 1) * p.in_rs + (s_bx[j] * 8 - 1) * p.in_ps + (s_cc >> 1) * p.in_cA + (s_cc & 1) * p.in_cB) * 4);
        bbits_s[j] = dead ? 15 : ((s_by[j] == 0) | ((s_by[j] == p.bh - 1) << 1) | ((s_bx[j] == 0) << 2) | ((s_bx[j] == p.bw - 1) << 3));
        if (dead) base_s[j] = SGG_OOB;
      }
      if constexpr (LNP) {
        ld_cc = s_cc;
        ld_bad = 0;
#pragma unroll
        for (int j = 0; j < K32_NB; ++j) {
          int b = s_grow[j] / p.bh;
          b = b < p.B ? b : p.B - 1;
          ld_mu[j] = p.ln_stats[2 * b];
          ld_rs[j] = p.ln_stats[2 * b + 1];
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      unsigned rel;
      int meta;
      item_plan(2 * half + jj, rel, meta);
      const int blk = (meta >> 24) & 1;
      const unsigned b0 = blk ? base_s[1] : base_s[0];
      const int bb = blk ? bbits_s[1] : bbits_s[0];
      const bool bad = !((meta >> 28) & 1) | ((((meta >> 20) & 15) & bb) != 0) | (b0 == SGG_OOB);
      const unsigned off = bad ? SGG_OOB : b0 + rel;
      if constexpr (LNP) ld_bad |= (int)bad << (2 * half + jj);
      pre[jj][0] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, off);
      pre[jj][1] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, off + 16u);
    }
    if constexpr (half == 1) {
      if (++s_cc == nch) {        // advance to this workgroup's next tile
        s_cc = 0;
        s_tile += tstride;
#pragma unroll
        for (int j = 0; j < K32_NB; ++j) {
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
    }
  };
  auto stage_write = [&](auto half_c, unsigned char* dst) __attribute__((always_inline)) {
    constexpr int half = decltype(half_c)::value;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      unsigned rel;
      int meta;
      item_plan(2 * half + jj, rel, meta);
      if constexpr (LNP) {
        const int blk = (meta >> 24) & 1;
        const float mu = blk ? ld_mu[1] : ld_mu[0], rs = blk ? ld_rs[1] : ld_rs[0];
        const int cb = ((ld_cc * 32) & (p.ln_nc - 1)) + ((meta >> 26) & 3) * 8;
        ln_elu8(pre[jj][0], pre[jj][1], lnp_s + cb, lnp_s + 512 + cb, mu, rs, (ld_bad >> (2 * half + jj)) & 1);
      }
      u32x4 pl[K32_P];
      split8<K32_P, HALF>(pre[jj][0], pre[jj][1], sa, pl);
      if ((meta >> 28) & 1) {
#pragma unroll
        for (int pp = 0; pp < K32_P; ++pp) *reinterpret_cast<u32x4*>(dst + pp * K32_PLANEB + (meta & 0xfffff)) = pl[pp];
      }
    }
  };

  // ---- weights: B fragments straight from L2, layout [tap][chunk][16-column group][plane][lane] x 16 B ----------------------------
  constexpr int TI = 4, TJ = 4;
  const unsigned w_lane = (unsigned)((n0 + wn0) >> 4) * 2048u + (unsigned)lane * 16u;
  const unsigned w_slab = (unsigned)(p.N >> 4) * 2048u;
  u32x4 rb[2][TJ][K32_P];
  auto load_b = [&](auto par_c, int cc, int tap) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value;
    const unsigned base = (unsigned)(tap * nch + cc) * w_slab + w_lane;
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int pp = 0; pp < K32_P; ++pp) rb[par][j][pp] = __builtin_bit_cast(u32x4, buf_load4(rs_w, base + (unsigned)(j * 2048 + pp * 1024)));
  };

  // A operand of tile i: lane l = (c4 = l >> 4, q = l & 15) holds pixel 16 i + q (block row 2 i + (q >> 3), column q & 7), channels
  // 8 c4 .. 8 c4 + 7; B operand of tile j: column 16 j + q, the same channels; D: column q, pixels 16 i + 4 c4 + r in register r.
  const int l16 = lane & 15, c4 = lane >> 4;
  const int pyl = l16 >> 3, pxl = l16 & 7;
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[TI][K32_P];           // single-buffered, refilled by halves: row tiles 0-1 of tap t + 1 behind the first half of tap t's MFMAs
  int cur = 0;                  // patch buffer the current chunk reads

  // A fragments of row tiles 2 * half, 2 * half + 1 of one tap from the resident patch
  auto read_a_half = [&](int tap, auto half_c) __attribute__((always_inline)) {
    constexpr int half = decltype(half_c)::value;
    const int kh = tap / 3, kw = tap % 3;
    const int dyy = p.flip ? 2 - kh : kh, dxx = p.flip ? 2 - kw : kw;
    int pyv = pyl, pxv = pxl;
    asm volatile("" : "+v"(pyv), "+v"(pxv));       // (keeps the per-tap addresses out of the loop-invariant hoisting, conv_halo.hip)
    const unsigned char* patch_w = lds + cur * K32_PATCHB + wblk * K32_BLKB;
#pragma unroll
    for (int i = 2 * half; i < 2 * half + 2; ++i) {
      const int ry = 2 * i + pyv + dyy, rx = pxv + dxx;
      const unsigned char* row = patch_w + (ry * K32_PITCH + rx) * 64 + ((c4 ^ k32_sw(ry, rx)) << 4);
#pragma unroll
      for (int pp = 0; pp < K32_P; ++pp) a[i][pp] = *reinterpret_cast<const u32x4*>(row + pp * K32_PLANEB);
    }
  };
  // MFMAs of row tiles 2 * half and 2 * half + 1 (24 instructions): hi * lo + lo * hi + hi * hi per (i, j)
  auto mma_half = [&](auto par_c, auto half_c) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value, half = decltype(half_c)::value;
#pragma unroll
    for (int i = 2 * half; i < 2 * half + 2; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        f32x4 d = acc[i][j];
        d = mfma16k32<HALF>(a[i][1], rb[par][j][0], d);
        d = mfma16k32<HALF>(a[i][0], rb[par][j][1], d);
        d = mfma16k32<HALF>(a[i][0], rb[par][j][0], d);
        acc[i][j] = d;
      }
  };
  // tap t: [B loads of tap t + 1] MFMAs of row tiles 0-1 | A reads of tap t + 1's row tiles 0-1 | MFMAs of row tiles 2-3 | A reads of
  // tap t + 1's row tiles 2-3 (they land behind the next tap's first MFMA half)
  auto tap_body = [&](auto par_c, auto tap_c, int cc) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value, tap = decltype(tap_c)::value;
    const int ncc = (cc + 1 == nch) ? 0 : cc + 1;         // (the chunk after the last one re-reads valid weights)
    load_b(std::integral_constant<int, par ^ 1>{}, tap == 8 ? ncc : cc, tap == 8 ? 0 : tap + 1);
    // the next chunk's patch -> the buffer nobody reads in this chunk, in two halves: loads at taps 1 / 5, split (+ LN prologue) and
    // LDS writes three taps later, inside that tap's scheduling region so that the VALU work issues in the shadow of its MFMAs
    if constexpr (tap == 1) stage_load(std::integral_constant<int, 0>{});
    if constexpr (tap == 5) stage_load(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (tap == 4) stage_write(std::integral_constant<int, 0>{}, lds + (cur ^ 1) * K32_PATCHB);
    if constexpr (tap == 8) stage_write(std::integral_constant<int, 1>{}, lds + (cur ^ 1) * K32_PATCHB);
    SGG_PRIO_HI();
    mma_half(par_c, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (tap < 8) read_a_half(tap + 1, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    mma_half(par_c, std::integral_constant<int, 1>{});
    SGG_PRIO_LO();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (tap < 8) read_a_half(tap + 1, std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  auto chunk = [&](auto par0_c, int cc) __attribute__((always_inline)) {
    constexpr int par0 = decltype(par0_c)::value;
    read_a_half(0, std::integral_constant<int, 0>{});
    read_a_half(0, std::integral_constant<int, 1>{});
#define K32_TAP(T) tap_body(std::integral_constant<int, (par0 + T) & 1>{}, std::integral_constant<int, T>{}, cc)
    K32_TAP(0); K32_TAP(1); K32_TAP(2); K32_TAP(3); K32_TAP(4); K32_TAP(5); K32_TAP(6); K32_TAP(7); K32_TAP(8);
#undef K32_TAP
    cur ^= 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // ---- output addressing: 16-byte stores.  After the quad transpose of a tile's four accumulator registers lane (c4, g = l16 >> 2,
  // k = l16 & 3) holds pixel (block row 2 i + (c4 >> 1), column 4 (c4 & 1) + k), channels 16 j + 4 g .. + 3  (conv_halo_pc.hip)
  int o_grow, o_bx;
  {
    const int beta = mt_begin * K32_NB + wblk;
    o_grow = beta / p.bw;
    o_bx = beta % p.bw;
  }
  const int wn = p.out_rs;
  const unsigned o_lane_b = (unsigned)((c4 >> 1) * wn + (4 * (c4 & 1) + (lane & 3)) * p.out_ps + (l16 >> 2) * 4) * 4u;
  int o_goff[TJ];       // float offset of this wave's 16-column groups (two per 32-column group of the output's addressing)
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int gi = ((n0 + wn0) >> 5) + (j >> 1);
    o_goff[j] = (gi >> 1) * p.out_nA + (gi & 1) * p.out_nB + (j & 1) * 16;
  }
  const float us_a = ldexpf(1.f, -ea), us_b = ldexpf(1.f, -eb);
  // (loaded once: a bias load inside the tile epilogue would wait (vmcnt(0)) for every prefetch in flight)
  float bias_v[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j) bias_v[j] = p.bias ? p.bias[n0 + wn0 + j * 16 + l16] : 0.f;
  constexpr int WN = 64;

  auto epilogue = [&](int tile) __attribute__((always_inline)) {
    const int beta = tile * K32_NB + wblk;
    const bool live = beta < p.nblk;
    const char* ob = reinterpret_cast<const char*>(p.out + (size_t)(o_grow * 8) * wn + (size_t)(o_bx * 8) * p.out_ps);
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // (two exact power-of-two factors: |ea|, |eb| <= 100 keeps each one a normal float)
          const float v = HALF ? fmaf(acc[i][j][r] * us_a, us_b, bias_v[j]) : acc[i][j][r] + bias_v[j];
          acc[i][j][r] = v;
          lsum += v;
        }
    if (live) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
          sgg_quad_transpose4(v0, v1, v2, v3, lane);
          const size_t so = ((size_t)(2 * i) * wn + o_goff[j]) * sizeof(float);   // scalar: block rows 2 i, 2 i + 1
          sgg_out_store4(reinterpret_cast<float*>(const_cast<char*>(ob) + so + o_lane_b), f32x4{v0, v1, v2, v3});
        }
    }
    if (p.tile_stats) {
      // (count, mean, M2, max dev) of this wave's 64 pixels x 64 channels (one 8x8 block: inside one sample)
      const float mean_w = wave_sum(lsum) * (1.f / (float)(64 * WN));
      float q = 0.f, dm = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = acc[i][j][r] - mean_w;
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
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    o_bx += adv_cols;
    o_grow += adv_rows;
    if (o_bx >= p.bw) { o_bx -= p.bw; ++o_grow; }
  };

  stage_load(std::integral_constant<int, 0>{});
  load_b(std::integral_constant<int, 0>{}, 0, 0);
  stage_write(std::integral_constant<int, 0>{}, lds);
  stage_load(std::integral_constant<int, 1>{});
  stage_write(std::integral_constant<int, 1>{}, lds);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // (C % 64 == 0: an even number of chunks; the B-fragment double buffer flips once per chunk of nine taps)
  for (int tile = mt_begin; tile < mt_end; tile += tstride) {
    for (int cc = 0; cc < nch; cc += 2) {
      chunk(std::integral_constant<int, 0>{}, cc);
      chunk(std::integral_constant<int, 1>{}, cc + 1);
    }
    epilogue(tile);
  }
}

// ---- host --------------------------------------------------------------------------------------------------------------------
void sgg_halo_k32_launch(const HaloParams& p_, int precision, hipStream_t st) {
  HaloParams p = p_;
  const int mtiles = sgg_cdiv(p.nblk, K32_NB), ntn = p.N / 128;
  int per_xcd = sgg_cdiv(mtiles, 8) * ntn;       // (tile, n-tile) pairs an XCD owns
  int gx = per_xcd < 64 ? per_xcd : 64;          // two workgroups on each of its 32 CUs
  gx = sgg_cdiv(gx, ntn) * ntn;
  p.gx = gx;
  const dim3 grid((unsigned)(8 * gx)), blk(256);
  const bool half = precision == 2;
  if (p.ln_stats) {
    if (half) hipLaunchKernelGGL((conv_halo3_k32_kernel<true, true>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((conv_halo3_k32_kernel<false, true>), grid, blk, 0, st, p);
  } else {
    if (half) hipLaunchKernelGGL((conv_halo3_k32_kernel<true, false>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((conv_halo3_k32_kernel<false, false>), grid, blk, 0, st, p);
  }
}
