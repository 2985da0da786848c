// Producer / consumer form of the halo-resident 3x3 / stride-1 convolution (gfx950), 128-column tiles, the two-piece 16-bit modes:
//   tf.layers.conv2d(kernel_size=3, strides=1, padding="same")   reference: architectures/generator_with_attention.py:31-57
// (conv2_3, conv2_4, conv3_1, conv3_2 and their Conv2DBackpropInput, train.py:265-266).
//
// conv_halo.hip runs two 4-wave workgroups per CU, every wave fetching, splitting, multiplying and storing for itself.  Its ablation
// builds (profiles/r03_halo_ablation*.log) put 10 % of the kernel on the output stores - not their count, their completion: vmcnt
// counts loads and stores in order, so the first wait for a weight fragment behind a tile's stores is a wait for the stores -,
// 10 % on the per-wave weight-fragment loads (8 KiB per wave and tap through the CU's 64 B/clk vector-memory return path, the
// same bytes in every wave of a column half) and 8.5 % on the patch staging.  Here ONE 8-wave workgroup owns the CU:
//   * waves 0-3, the CONSUMERS (2 blocks x 2 column halves as before, 64 pixels x 64 columns each), issue nothing but LDS reads,
//     MFMAs and - never waited for - the output stores: no vector-memory load, hence no vmcnt wait, anywhere in their loop;
//   * waves 4-7, the PRODUCERS, own the vector-memory queue: the weight fragments of the next taps go L2 -> LDS by LDS-DMA
//     (buffer_load ... lds, no VGPRs, no VALU) into a ring of four 16-KiB tap slots shared by the four consumers (the L1 traffic of
//     the weight operand halves, the consumers read it at LDS bandwidth), and the activation patch is staged two 32-channel chunks
//     ahead (load, f32 -> 2 x fp16 split or the LayerNorm + ELU prologue, LDS write).
// One raw s_barrier per tap (48 MFMAs of 16x16x32 per consumer) is the only synchronisation: behind barrier g the producers guarantee that the
// fragments of tap g+1 (and, at a chunk's last tap, the next patch) have landed, the consumers that they no longer read slot g.
// Both roles run the SAME loop nest (tile, chunk, nine statically unrolled taps) with exactly one barrier per tap.
// LDS: 2 patch buffers (61,440 B) + ring (65,536 B) (+ LayerNorm parameters, 4 KB) of the 160 KB a gfx950 workgroup may use.
// Where it is used: every dgrad and every forward WITHOUT LN prologue of the 128-column layers; forwards with the prologue stay on
// conv_halo3_kernel (sgg_amd/trunk.py: _query_layouts; measured in DESIGN.md section 3, "Round 3").
#include "split16.h"
#include "conv_halo.h"
#include <type_traits>

#define PC_PITCH 12
#define PC_BLKB (10 * PC_PITCH * 64)      // bytes of one plane of one block's patch
#define PC_P 2
#define PC_D 4                            // ring slots: the DMA of tap g + 4 is issued behind barrier g and must have landed by barrier g + 3
#ifndef PC_INTERLEAVE
#define PC_INTERLEAVE 1
#endif
// 1 (round 5): with the patch DMA (DMAP) the four producer waves split by ROLE - waves 4, 5 move the weight fragments of every tap,
// waves 6, 7 the patch (one plane each), a few DMAs per tap over taps 0 .. 7 of the chunk before, waited for ONCE in front of that
// chunk's last barrier.  vmcnt retires in order: with all four waves doing both, the first wait for a weight fragment behind a
// patch DMA (tap 3) was a wait for the whole patch, i.e. the patch had three taps (~1.2 us) to arrive - a burst of 30 / 60 KB per CU
// that every CU issues at the same tap; split, it has eight taps and is issued evenly.  0: every producer wave does both (round 4).
#ifndef PC_SPLIT_PRODUCERS
#define PC_SPLIT_PRODUCERS 1
#endif
// ... and the same split for the register-staged patch (f32 sources, the LN prologue): 0 = four waves doing both (round 3)
#ifndef PC_SPLIT_PRODUCERS_REG
#define PC_SPLIT_PRODUCERS_REG 1
#endif
// s_waitcnt immediate of gfx9 for vmcnt(n) alone (expcnt, lgkmcnt: no wait): vmcnt low 4 bits in [3:0], its high 2 bits in [15:14]
constexpr int pc_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

// XOR swizzle of the 16-byte chunk index inside a pixel's 64-byte row (write and read side).  With the 16x16x32 operand read (16
// lanes = two patch rows x 8 pixels take one chunk each, four such groups per ds_read_b128) every read of every tap is conflict free
// with chunk ^= ((rx ^ (ry >> 1)) & 1) | ((ry & 1) << 1)  (brute force over the instruction's lane groups, MI355X_MICROARCH.md LDS;
// conv_halo.hip's ((rx >> 2) & 1) | ((ry & 1) << 1) is 2-way conflicted on this access pattern).
__device__ __forceinline__ int pc_sw(int ry, int rx) { return ((rx ^ (ry >> 1)) & 1) | ((ry & 1) << 1); }

typedef __attribute__((address_space(3))) void* pc_lds_ptr;

// DMAP: the source is a PRE-SPLIT tensor (split16.h; HALF, no LN prologue): the producers stage the patch by LDS-DMA as well - no
// staging registers, no vector arithmetic at all on the SIMDs the MFMA waves run on.
// NB (round 5): 8x8 blocks of a workgroup tile.  2: 2 blocks x 128 output columns (a consumer wave = one block x one 64-column half);
// 4: FOUR blocks x 64 columns for the 64-column launches (conv2_2's and conv2_3's dgrad: a consumer wave = one block x all 64
// columns - the same 64 pixels x 64 columns per wave, the same operand reads and MFMAs per tap; the ring slot of a tap is 8 KiB, the
// patch of a chunk 60 KiB: 153 of the 160 KiB of LDS).  With the patch DMA only.
template <bool HALF, bool LNP, bool DMAP = false, int NB = 2>
__global__ __launch_bounds__(512, 2) void conv_halo3_pc_kernel(HaloParams p) {
  static_assert(!DMAP || (HALF && !LNP), "patch DMA: pre-split fp16 pieces, no prologue");
  static_assert(NB == 2 || (NB == 4 && DMAP), "four-block tiles: pre-split sources only");
  constexpr int PC_NB = NB;
  constexpr int BN = 256 / NB;                          // output columns of a workgroup tile
  constexpr int PC_PLANEB = PC_NB * PC_BLKB;
  constexpr int PC_PATCHB = PC_P * PC_PLANEB;           // one patch buffer
  constexpr int PC_SLOTB = (BN / 16) * 2048;            // one tap of weight fragments for BN columns: [n-tile BN / 16][plane 2][lane 64] x 16 B
  constexpr int PC_ITEMS = PC_NB * 400;
  constexpr int PC_NPASS = (PC_ITEMS + 255) / 256;      // (block, patch pixel, 8-channel group) items over the 256 producer threads
  constexpr int WQ = PC_SLOTB / 4096;                   // weight DMAs per producer wave and tap (1 KiB each)
  constexpr int DPP = PC_NB * 120 / 16;                 // patch DMAs per plane and chunk (16 slots of 64 B each): 15 / 30
  constexpr int DPW = (DPP + 1) / 2;                    // ... of which a producer wave issues up to 8 / 15
  constexpr int PQ = NB == 2 ? 8 : 16;                  // patch DMAs in a producer wave's queue per chunk (the rest: fillers)
  constexpr int PLQ = DMAP ? PQ : 2 * PC_NPASS;         // patch loads / DMAs per chunk in the vmcnt queue of a producer wave
  // ONE __shared__ object: with the LayerNorm parameters in an array of their own the compiler waits vmcnt(0) - for every weight DMA
  // in flight - in front of each patch write (cdna_hip_programming.md, "a second __shared__ object beside the glds staging array")
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PC_PATCHB + PC_D * PC_SLOTB + (LNP ? 4096 : 0) + (DMAP ? 1024 : 0)];
  unsigned char* const ring = lds + 2 * PC_PATCHB;
  float* const lnp_s = reinterpret_cast<float*>(lds + 2 * PC_PATCHB + PC_D * PC_SLOTB);      // gamma[0..511], beta at +512 (C <= 512: host check)

  // ---- persistent workgroup: as conv_halo3_kernel (XCD k owns a contiguous eighth of the M-tiles, its workgroups walk it interleaved)
  const int ntiles_n = p.N / BN;
  const int mtiles = (p.nblk + PC_NB - 1) / PC_NB;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int nt = jx % ntiles_n;
  const int tstride = p.gx / ntiles_n;
  const int mt_begin = (int)(((long long)xcd * mtiles) >> 3) + jx / ntiles_n;
  const int mt_end = (int)(((long long)(xcd + 1) * mtiles) >> 3);
  if (mt_begin >= mt_end) return;             // (whole workgroup: no barrier is ever executed by anyone)
  const int n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nch = p.C >> 5;
  const int adv_rows = (tstride * PC_NB) / p.bw, adv_cols = (tstride * PC_NB) % p.bw;   // block advance between this workgroup's tiles
  int ea = 0, eb = 0;
  if constexpr (HALF) {
    ea = scale_exp_from_amax(*p.amax_src);
    eb = scale_exp_from_amax(*p.amax_w);
  }

  if (wave >= 4) {
    // =================================================================================================================
    // PRODUCERS (waves 4-7, pt = 0 .. 255).  Patch: the patch of chunk c + 2 is LOADED at tap 0 of chunk c (two register sets), the
    // patch of chunk c + 1 is split and WRITTEN at taps 1 .. 4 of chunk c: a load has at least ten taps to arrive before its first
    // use, and three before the first weight-DMA wait that covers it (vmcnt retires in order).  Weights: tap (d_cc, d_tap) of the
    // fragment stream -> ring slot by LDS-DMA, four of the tap's sixteen 1-KiB pieces per wave.
    // That is the form every wave doing BOTH jobs takes (the code below the role branches; -DPC_SPLIT_PRODUCERS=0 /
    // -DPC_SPLIT_PRODUCERS_REG=0).  Round 5 splits the four waves by ROLE instead - waves 4, 5 the weight fragments, waves 6, 7 the
    // patch - in every variant: with both jobs in one queue the first wait for a weight fragment issued behind the patch loads is a
    // wait for the whole patch.  (Round 3 had measured such a split 4 % slower with the LayerNorm prologue; with the weight waves'
    // hand-counted vmcnt and the patch written one pass per tap it is the faster form there too: 346 -> 353 TFLOP/s, the step -0.17 ms,
    // profiles/r05_split_producers_reg_ab.log.)
    // =================================================================================================================
    const int pt = tid - 256, pw = wave - 4;
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, p.w_bytes, 0x00020000);
    if constexpr (DMAP && PC_SPLIT_PRODUCERS) {
      const unsigned w_slab2 = (unsigned)(p.N >> 5) * 4096u;
      if (pw < 2) {
        // ---- weight waves: half of every tap's slot each (W2 DMAs of 1 KiB per tap); in their queue nothing but these ----------------
        constexpr int W2 = PC_SLOTB / 2048;
        const unsigned w_lane2 = (unsigned)(n0 >> 5) * 4096u + (unsigned)pw * (unsigned)(PC_SLOTB / 2) + (unsigned)lane * 16u;
        int d_cc = 0, d_tap = 0, d_slot = 0;
        auto issue = [&]() __attribute__((always_inline)) {
          const unsigned base = (unsigned)(d_tap * nch + d_cc) * w_slab2 + w_lane2;
          unsigned char* dst = ring + d_slot * PC_SLOTB + pw * (PC_SLOTB / 2);
#pragma unroll
          for (int q = 0; q < W2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (pc_lds_ptr)(dst + q * 1024), 16, base + (unsigned)(q * 1024), 0, 0, 0);
          if (++d_tap == 9) {
            d_tap = 0;
            if (++d_cc == nch) d_cc = 0;
          }
          d_slot = (d_slot + 1) & (PC_D - 1);
        };
#pragma unroll
        for (int k = 0; k < PC_D; ++k) issue();
        __builtin_amdgcn_s_waitcnt(pc_vmcnt(0));
        __builtin_amdgcn_s_barrier();             // barrier "-1": the fragments of taps 0 .. 3 are in LDS
        for (int tile = mt_begin; tile < mt_end; tile += tstride)
          for (int cc = 0; cc < nch; ++cc) {
#pragma unroll
            for (int T = 0; T < 9; ++T) {
              // the fragments of tap g + 1 (issued behind barrier g - 3) must have landed; younger: those of taps g + 2, g + 3
              __builtin_amdgcn_s_waitcnt(pc_vmcnt(2 * W2));
              __builtin_amdgcn_s_barrier();       // barrier g
              issue();                            // tap g + 4 into slot g % 4 (the consumers have finished reading tap g)
            }
          }
        __builtin_amdgcn_s_waitcnt(pc_vmcnt(0));  // (the last DMAs target this workgroup's LDS: they must not outlive it)
        return;
      }
      // ---- patch waves: plane pl of every chunk, DPP instructions of 16 slots, PT per tap over taps 0 .. 7 of the chunk before -------
      const int pl = pw - 2;
      constexpr int PT = (DPP + 7) / 8;
      unsigned pm_rel[DPP];
      int pm_meta[DPP];            // bits 0..3 border bits, 4..5 block, 6 valid
#pragma unroll
      for (int k = 0; k < DPP; ++k) {
        const int slot = 16 * k + (lane >> 2);      // slot inside the plane: [block][row 0..9][pitch 12]
        const int blk = slot / 120, r = slot - 120 * blk;
        const int ry = r / PC_PITCH, rx = r - ry * PC_PITCH;
        const int piece = (lane & 3) ^ pc_sw(ry, rx);
        pm_rel[k] = (unsigned)((ry * p.in_rs + rx * p.in_ps) * 4 + piece * 16);
        pm_meta[k] = ((ry == 0) | ((ry == 9) << 1) | ((rx == 0) << 2) | ((rx == 9) << 3)) | (blk << 4) | ((int)(rx < 10) << 6);
      }
      int s_grow[PC_NB], s_by[PC_NB], s_bx[PC_NB];
#pragma unroll
      for (int j = 0; j < PC_NB; ++j) {
        const int beta = mt_begin * PC_NB + j;
        s_grow[j] = beta / p.bw;
        s_bx[j] = beta % p.bw;
        s_by[j] = s_grow[j] % p.bh;
      }
      int s_tile = mt_begin, s_cc = 0;
      unsigned base[PC_NB];
      int bbits[PC_NB];
      // block origins and border bits of the next (tile, chunk) of the stream -> base / bbits; past the last tile: out of range (zeros)
      auto next_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PC_NB; ++j) {
          const bool dead = (s_tile >= mt_end) | (s_tile * PC_NB + j >= p.nblk);
          base[j] = (unsigned)(((s_grow[j] * 8 - 1) * p.in_rs + (s_bx[j] * 8 - 1) * p.in_ps + (s_cc >> 1) * p.in_cA + (s_cc & 1) * p.in_cB) * 4);
          bbits[j] = dead ? 15 : ((s_by[j] == 0) | ((s_by[j] == p.bh - 1) << 1) | ((s_bx[j] == 0) << 2) | ((s_bx[j] == p.bw - 1) << 3));
          if (dead) base[j] = SGG_OOB;
        }
        if (++s_cc == nch) {        // advance to this workgroup's next tile
          s_cc = 0;
          s_tile += tstride;
#pragma unroll
          for (int j = 0; j < PC_NB; ++j) {
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
      // instructions [k0, k0 + n) of the plane into patch buffer dstbuf
      auto patch_part = [&](unsigned char* dstbuf, auto k0_c, auto n_c) __attribute__((always_inline)) {
        constexpr int k0 = decltype(k0_c)::value, n = decltype(n_c)::value;
#pragma unroll
        for (int k = k0; k < k0 + n && k < DPP; ++k) {
          const int blk = (pm_meta[k] >> 4) & 3;
          unsigned b0 = base[0];
          int bb = bbits[0];
#pragma unroll
          for (int j = 1; j < PC_NB; ++j) {
            b0 = blk == j ? base[j] : b0;
            bb = blk == j ? bbits[j] : bb;
          }
          const bool bad = !((pm_meta[k] >> 6) & 1) | (((pm_meta[k] & 15) & bb) != 0) | (b0 == SGG_OOB);
          const unsigned off = bad ? SGG_OOB : b0 + pm_rel[k];
          // (the plane's byte offset as SCALAR offset: the immediate offset field would move the LDS address too, scripts/ubench/dma_oob.hip)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (pc_lds_ptr)(dstbuf + pl * PC_PLANEB + k * 1024), 16, off, pl * 64, 0, 0);
        }
      };
      next_chunk();
      patch_part(lds, std::integral_constant<int, 0>{}, std::integral_constant<int, DPP>{});      // chunk 0, whole
      __builtin_amdgcn_s_waitcnt(pc_vmcnt(0));
      __builtin_amdgcn_s_barrier();               // barrier "-1": patch 0 is in LDS
      int cur = 0;
      for (int tile = mt_begin; tile < mt_end; tile += tstride)
        for (int cc = 0; cc < nch; ++cc) {
          unsigned char* nb = lds + (cur ^ 1) * PC_PATCHB;      // the buffer the consumers left at the last barrier
          next_chunk();
#define PC_PTAP(T)                                                                                            \
          patch_part(nb, std::integral_constant<int, (T) * PT>{}, std::integral_constant<int, PT>{});          \
          __builtin_amdgcn_s_barrier();
          PC_PTAP(0) PC_PTAP(1) PC_PTAP(2) PC_PTAP(3) PC_PTAP(4) PC_PTAP(5) PC_PTAP(6) PC_PTAP(7)
#undef PC_PTAP
          __builtin_amdgcn_s_waitcnt(pc_vmcnt(0));            // the next patch has landed ...
          __builtin_amdgcn_s_barrier();                        // ... behind barrier 8
          cur ^= 1;
        }
      return;
    }
    if constexpr (!DMAP && PC_SPLIT_PRODUCERS_REG) {
      // ---- register-staged patch (f32 source, or the LN prologue on the producing layer's pre-LayerNorm y), roles split as above:
      // waves 4, 5 the weight fragments; waves 6, 7 (128 threads) the patch - chunk c + 2 is LOADED at tap 0 of chunk c (two register
      // sets), chunk c + 1 normalised / split and WRITTEN at taps 1 .. 7, one pass of 128 items per tap.  Their queue holds nothing but
      // their own patch loads, which the compiler counts exactly: no wait for a weight fragment is a wait for a patch any more, and
      // the prologue's arithmetic (v_exp, selects, the two-piece split) sits on two SIMDs whose consumer waves it delays by issue
      // slots only.  (Round 5: this variant serves the forward-only passes that fuse LN4 / LN5, trunk.pc_ln_fusion_pays.)
      static_assert(NB == 2, "register-staged patch: two-block tiles");
      const unsigned w_slab2 = (unsigned)(p.N >> 5) * 4096u;
      const float sa2 = ldexpf(1.f, ea);
      if (pw < 2) {
        constexpr int W2 = PC_SLOTB / 2048;
        const unsigned w_lane2 = (unsigned)(n0 >> 5) * 4096u + (unsigned)pw * (unsigned)(PC_SLOTB / 2) + (unsigned)lane * 16u;
        int d_cc = 0, d_tap = 0, d_slot = 0;
        auto issue = [&]() __attribute__((always_inline)) {
          const unsigned base = (unsigned)(d_tap * nch + d_cc) * w_slab2 + w_lane2;
          unsigned char* dst = ring + d_slot * PC_SLOTB + pw * (PC_SLOTB / 2);
#pragma unroll
          for (int q = 0; q < W2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (pc_lds_ptr)(dst + q * 1024), 16, base + (unsigned)(q * 1024), 0, 0, 0);
          if (++d_tap == 9) {
            d_tap = 0;
            if (++d_cc == nch) d_cc = 0;
          }
          d_slot = (d_slot + 1) & (PC_D - 1);
        };
#pragma unroll
        for (int k = 0; k < PC_D; ++k) issue();
        if constexpr (LNP) __builtin_amdgcn_s_barrier();      // (the patch waves' barrier behind the lnp_s fill)
        __builtin_amdgcn_s_waitcnt(pc_vmcnt(0));
        __builtin_amdgcn_s_barrier();             // barrier "-1"
        for (int tile = mt_begin; tile < mt_end; tile += tstride)
          for (int cc = 0; cc < nch; ++cc) {
#pragma unroll
            for (int T = 0; T < 9; ++T) {
              __builtin_amdgcn_s_waitcnt(pc_vmcnt(2 * W2));
              __builtin_amdgcn_s_barrier();
              issue();
            }
          }
        __builtin_amdgcn_s_waitcnt(pc_vmcnt(0));
        return;
      }
      // ---- patch waves ----
      const int pt2 = tid - 384;                  // 0 .. 127
      constexpr int NP2 = (PC_ITEMS + 127) / 128; // 7 passes of 128 (block, patch pixel, 8-channel group) items
      if constexpr (LNP) {
        for (int c = pt2; c < p.ln_nc; c += 128) {
          lnp_s[c] = p.ln_gamma[c];
          lnp_s[512 + c] = p.ln_beta[c];
        }
      }
      unsigned q_rel[NP2];
      int q_meta[NP2];      // bits 0..19 LDS byte offset inside a plane, 20..23 border bits, 24 block, 26..27 channel group, 28 valid
#pragma unroll
      for (int j = 0; j < NP2; ++j) {
        const int it = pt2 + 128 * j;
        const int blk = (it / 400) & 1, r = it % 400;
        const int px = r >> 2, ch8 = r & 3;
        const int ry = px / 10, rx = px % 10;
        q_rel[j] = (unsigned)((ry * p.in_rs + rx * p.in_ps + ch8 * 8) * 4);
        const int bits = (ry == 0) | ((ry == 9) << 1) | ((rx == 0) << 2) | ((rx == 9) << 3);
        q_meta[j] = (blk * PC_BLKB + (ry * PC_PITCH + rx) * 64 + ((ch8 ^ pc_sw(ry, rx)) << 4)) | (bits << 20) | (blk << 24) |
                    (ch8 << 26) | ((it < PC_ITEMS) << 28);
      }
      float q_mu[2][PC_NB], q_rs[2][PC_NB];
      int q_cc[2] = {0, 0}, q_bad[2] = {0, 0};
      int s_grow[PC_NB], s_by[PC_NB], s_bx[PC_NB];
#pragma unroll
      for (int j = 0; j < PC_NB; ++j) {
        const int beta = mt_begin * PC_NB + j;
        s_grow[j] = beta / p.bw;
        s_bx[j] = beta % p.bw;
        s_by[j] = s_grow[j] % p.bh;
      }
      int s_tile = mt_begin, s_cc = 0;
      f32x4 qre[2][NP2][2];
      auto q_load = [&](auto s_c) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value;
        unsigned base[PC_NB];
        int bbits[PC_NB];
#pragma unroll
        for (int j = 0; j < PC_NB; ++j) {
          const bool dead = (s_tile >= mt_end) | (s_tile * PC_NB + j >= p.nblk);
          base[j] = (unsigned)(((s_grow[j] * 8 - 1) * p.in_rs + (s_bx[j] * 8 - 1) * p.in_ps + (s_cc >> 1) * p.in_cA + (s_cc & 1) * p.in_cB) * 4);
          bbits[j] = dead ? 15 : ((s_by[j] == 0) | ((s_by[j] == p.bh - 1) << 1) | ((s_bx[j] == 0) << 2) | ((s_bx[j] == p.bw - 1) << 3));
          if (dead) base[j] = SGG_OOB;
        }
        if constexpr (LNP) {
          q_cc[S] = s_cc;
          q_bad[S] = 0;
#pragma unroll
          for (int j = 0; j < PC_NB; ++j) {
            int b = s_grow[j] / p.bh;
            b = b < p.B ? b : p.B - 1;
            q_mu[S][j] = p.ln_stats[2 * b];
            q_rs[S][j] = p.ln_stats[2 * b + 1];
          }
        }
#pragma unroll
        for (int j = 0; j < NP2; ++j) {
          const int blk = (q_meta[j] >> 24) & 1;
          const unsigned b0 = blk ? base[1] : base[0];
          const int bb = blk ? bbits[1] : bbits[0];
          const bool bad = !((q_meta[j] >> 28) & 1) | ((((q_meta[j] >> 20) & 15) & bb) != 0) | (b0 == SGG_OOB);
          const unsigned off = bad ? SGG_OOB : b0 + q_rel[j];
          if constexpr (LNP) q_bad[S] |= (int)bad << j;
          const unsigned o0 = LNP ? off : stage_off0(off, p.src_s16);
          qre[S][j][0] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, o0);
          qre[S][j][1] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, LNP ? off + 16u : stage_off1(o0, p.src_s16));
        }
        if (++s_cc == nch) {        // advance to this workgroup's next tile
          s_cc = 0;
          s_tile += tstride;
#pragma unroll
          for (int j = 0; j < PC_NB; ++j) {
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
      // pass j of register set S: LN prologue (LNP), split, write into patch buffer dst
      auto q_pass = [&](auto s_c, auto j_c, unsigned char* dst) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value, j = decltype(j_c)::value;
        if constexpr (LNP) {
          const int blk = (q_meta[j] >> 24) & 1;
          const float mu = blk ? q_mu[S][1] : q_mu[S][0], rs = blk ? q_rs[S][1] : q_rs[S][0];
          const int cb = ((q_cc[S] * 32) & (p.ln_nc - 1)) + ((q_meta[j] >> 26) & 3) * 8;
          ln_elu8(qre[S][j][0], qre[S][j][1], lnp_s + cb, lnp_s + 512 + cb, mu, rs, (q_bad[S] >> j) & 1);
        }
        u32x4 pl[PC_P];
        if constexpr (LNP || !HALF) split8<PC_P, HALF>(qre[S][j][0], qre[S][j][1], sa2, pl);
        else stage_planes<PC_P, HALF>(qre[S][j][0], qre[S][j][1], sa2, p.src_s16, pl);
        if (j < NP2 - 1 || ((q_meta[j] >> 28) & 1)) {
#pragma unroll
          for (int pp = 0; pp < PC_P; ++pp) *reinterpret_cast<u32x4*>(dst + pp * PC_PLANEB + (q_meta[j] & 0xfffff)) = pl[pp];
        }
      };
      if constexpr (LNP) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // lnp_s is filled (matched by every other wave)
      }
      q_load(std::integral_constant<int, 0>{});   // chunk 0
#define PC_QP(S, J, DST) q_pass(std::integral_constant<int, S>{}, std::integral_constant<int, J>{}, DST);
      PC_QP(0, 0, lds) PC_QP(0, 1, lds) PC_QP(0, 2, lds) PC_QP(0, 3, lds) PC_QP(0, 4, lds) PC_QP(0, 5, lds) PC_QP(0, 6, lds)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      q_load(std::integral_constant<int, 1>{});   // chunk 1: in flight across the prologue barrier
      __builtin_amdgcn_s_barrier();               // barrier "-1": patch 0 is in LDS
      int cur = 0;
      // one chunk c (register-set parity S = c & 1): chunk c + 2 -> set S at tap 0, chunk c + 1 (set S ^ 1) written at taps 1 .. 7
      auto q_chunk = [&](auto s_c) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value;
        unsigned char* nb = lds + (cur ^ 1) * PC_PATCHB;
        q_load(s_c);
        __builtin_amdgcn_s_barrier();             // tap 0
        PC_QP(S ^ 1, 0, nb) __builtin_amdgcn_s_barrier();
        PC_QP(S ^ 1, 1, nb) __builtin_amdgcn_s_barrier();
        PC_QP(S ^ 1, 2, nb) __builtin_amdgcn_s_barrier();
        PC_QP(S ^ 1, 3, nb) __builtin_amdgcn_s_barrier();
        PC_QP(S ^ 1, 4, nb) __builtin_amdgcn_s_barrier();
        PC_QP(S ^ 1, 5, nb) __builtin_amdgcn_s_barrier();
        PC_QP(S ^ 1, 6, nb) __builtin_amdgcn_s_barrier();      // tap 7
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next patch is written ...
        __builtin_amdgcn_s_barrier();                          // ... behind barrier 8
        cur ^= 1;
      };
#undef PC_QP_UNUSED
      for (int tile = mt_begin; tile < mt_end; tile += tstride)
        for (int cc = 0; cc < nch; cc += 2) {
          q_chunk(std::integral_constant<int, 0>{});
          q_chunk(std::integral_constant<int, 1>{});
        }
#undef PC_QP
      return;
    }
    const float sa = ldexpf(1.f, ea);
    if constexpr (LNP) {
      for (int c = pt; c < p.ln_nc; c += 256) {
        lnp_s[c] = p.ln_gamma[c];
        lnp_s[512 + c] = p.ln_beta[c];
      }
    }
    // ---- staging plan (static per thread): item -> offset relative to its block's patch origin, border bits, LDS offset ----
    unsigned it_rel[PC_NPASS];
    int it_meta[PC_NPASS];      // bits 0..19 LDS byte offset inside a plane, 20..23 border bits, 24 block, 26..27 channel group, 28 valid
#pragma unroll
    for (int j = 0; j < PC_NPASS; ++j) {
      const int it = pt + 256 * j;
      const int blk = (it / 400) % PC_NB, r = it % 400;
      const int px = r >> 2, ch8 = r & 3;
      const int ry = px / 10, rx = px % 10;
      it_rel[j] = (unsigned)((ry * p.in_rs + rx * p.in_ps + ch8 * 8) * 4);
      const int bits = (ry == 0) | ((ry == 9) << 1) | ((rx == 0) << 2) | ((rx == 9) << 3);
      // (the register-staged patch - one block bit in it_meta - serves two-block tiles only: NB == 4 implies DMAP, static_assert above)
      it_meta[j] = (blk * PC_BLKB + (ry * PC_PITCH + rx) * 64 + ((ch8 ^ pc_sw(ry, rx)) << 4)) | (bits << 20) | (blk << 24) |
                   (ch8 << 26) | ((it < PC_ITEMS) << 28);
    }
    float ld_mu[2][PC_NB], ld_rs[2][PC_NB];
    int ld_cc[2] = {0, 0}, ld_bad[2] = {0, 0};
    int s_grow[PC_NB], s_by[PC_NB], s_bx[PC_NB];
#pragma unroll
    for (int j = 0; j < PC_NB; ++j) {
      const int beta = mt_begin * PC_NB + j;
      s_grow[j] = beta / p.bw;
      s_bx[j] = beta % p.bw;
      s_by[j] = s_grow[j] % p.bh;
    }
    int s_tile = mt_begin, s_cc = 0;
    f32x4 pre[2][PC_NPASS][2];
    // issue the global loads of the next (tile, chunk) patch of the stream into register set S; past the last tile: out-of-range offsets (zeros)
    auto stage_load = [&](auto s_c) __attribute__((always_inline)) {
      constexpr int S = decltype(s_c)::value;
      unsigned base[PC_NB];
      int bbits[PC_NB];
#pragma unroll
      for (int j = 0; j < PC_NB; ++j) {
        const bool dead = (s_tile >= mt_end) | (s_tile * PC_NB + j >= p.nblk);
        base[j] = (unsigned)(((s_grow[j] * 8 - 1) * p.in_rs + (s_bx[j] * 8 - 1) * p.in_ps + (s_cc >> 1) * p.in_cA + (s_cc & 1) * p.in_cB) * 4);
        bbits[j] = dead ? 15 : ((s_by[j] == 0) | ((s_by[j] == p.bh - 1) << 1) | ((s_bx[j] == 0) << 2) | ((s_bx[j] == p.bw - 1) << 3));
        if (dead) base[j] = SGG_OOB;
      }
      if constexpr (LNP) {
        ld_cc[S] = s_cc;
        ld_bad[S] = 0;
#pragma unroll
        for (int j = 0; j < PC_NB; ++j) {
          int b = s_grow[j] / p.bh;
          b = b < p.B ? b : p.B - 1;
          ld_mu[S][j] = p.ln_stats[2 * b];
          ld_rs[S][j] = p.ln_stats[2 * b + 1];
        }
      }
#pragma unroll
      for (int j = 0; j < PC_NPASS; ++j) {
        const int blk = (it_meta[j] >> 24) & 1;
        const unsigned b0 = blk ? base[1] : base[0];
        const int bb = blk ? bbits[1] : bbits[0];
        const bool bad = !((it_meta[j] >> 28) & 1) | ((((it_meta[j] >> 20) & 15) & bb) != 0) | (b0 == SGG_OOB);
        const unsigned off = bad ? SGG_OOB : b0 + it_rel[j];
        if constexpr (LNP) ld_bad[S] |= (int)bad << j;
        const unsigned o0 = LNP ? off : stage_off0(off, p.src_s16);
        pre[S][j][0] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, o0);
        pre[S][j][1] = buf_load4_aux<SGG_PATCH_LOAD_AUX>(rs_src, LNP ? off + 16u : stage_off1(o0, p.src_s16));
      }
      if (++s_cc == nch) {        // advance to this workgroup's next tile
        s_cc = 0;
        s_tile += tstride;
#pragma unroll
        for (int j = 0; j < PC_NB; ++j) {
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
    // LN prologue of half `hf` (4 channels) of pass j of register set S, in place
    auto stage_ln_half = [&](auto s_c, auto j_c, auto hf_c) __attribute__((always_inline)) {
      constexpr int S = decltype(s_c)::value, j = decltype(j_c)::value, hf = decltype(hf_c)::value;
      const int blk = (it_meta[j] >> 24) & 1;
      const float mu = blk ? ld_mu[S][1] : ld_mu[S][0], rs = blk ? ld_rs[S][1] : ld_rs[S][0];
      const int cb = ((ld_cc[S] * 32) & (p.ln_nc - 1)) + ((it_meta[j] >> 26) & 3) * 8 + 4 * hf;
      ln_elu4(pre[S][j][hf], lnp_s + cb, lnp_s + 512 + cb, mu, rs, (ld_bad[S] >> j) & 1);
    };
    // split pass j of register set S (LNP: after both stage_ln_half) and write it into patch buffer `dst`
    auto stage_write_pass = [&](auto s_c, auto j_c, unsigned char* dst) __attribute__((always_inline)) {
      constexpr int S = decltype(s_c)::value, j = decltype(j_c)::value;
      u32x4 pl[PC_P];
      if constexpr (LNP || !HALF) split8<PC_P, HALF>(pre[S][j][0], pre[S][j][1], sa, pl);
      else stage_planes<PC_P, HALF>(pre[S][j][0], pre[S][j][1], sa, p.src_s16, pl);
      // (passes 0 .. 2 cover items 0 .. 767: always valid; the last pass holds 32 items)
      if (j < PC_NPASS - 1 || ((it_meta[j] >> 28) & 1)) {
#pragma unroll
        for (int pp = 0; pp < PC_P; ++pp) *reinterpret_cast<u32x4*>(dst + pp * PC_PLANEB + (it_meta[j] & 0xfffff)) = pl[pp];
      }
    };
    // ---- DMAP: the patch of one chunk = 2 planes x (NB * 120) slots (NB blocks of 10 rows x pitch 12) x 64 B = 2 x DPP DMA instructions of
    // 16 slots; the two producer waves of a plane issue instructions [DPW h, DPW h + DPW) of it (h = pw & 1), padded with fillers to PQ
    // per wave (a filler goes to a scratch KiB with out-of-range offsets: every wave issues exactly PQ, so one vmcnt protocol serves all
    // four).  Lane l of an instruction writes LDS position l & 3 of slot 16 q + (l >> 2), so it FETCHES the piece the swizzle puts
    // there: (l & 3) ^ pc_sw(ry, rx).
    unsigned dm_rel[PQ];
    int dm_meta[PQ];             // bits 0..3 border bits, 4..5 block, 6 valid
    auto patch_dma = [&](unsigned char* dstbuf) __attribute__((always_inline)) {
      unsigned base[PC_NB];
      int bbits[PC_NB];
#pragma unroll
      for (int j = 0; j < PC_NB; ++j) {
        const bool dead = (s_tile >= mt_end) | (s_tile * PC_NB + j >= p.nblk);
        base[j] = (unsigned)(((s_grow[j] * 8 - 1) * p.in_rs + (s_bx[j] * 8 - 1) * p.in_ps + (s_cc >> 1) * p.in_cA + (s_cc & 1) * p.in_cB) * 4);
        bbits[j] = dead ? 15 : ((s_by[j] == 0) | ((s_by[j] == p.bh - 1) << 1) | ((s_bx[j] == 0) << 2) | ((s_bx[j] == p.bw - 1) << 3));
        if (dead) base[j] = SGG_OOB;
      }
      const int plane = pw >> 1, hh = pw & 1;
      unsigned char* dst = dstbuf + plane * PC_PLANEB + hh * DPW * 1024;
#pragma unroll
      for (int k = 0; k < PQ; ++k) {
        const int blk = (dm_meta[k] >> 4) & 3;
        unsigned b0 = base[0];
        int bb = bbits[0];
#pragma unroll
        for (int j = 1; j < PC_NB; ++j) {
          b0 = blk == j ? base[j] : b0;
          bb = blk == j ? bbits[j] : bb;
        }
        const bool bad = !((dm_meta[k] >> 6) & 1) | (((dm_meta[k] & 15) & bb) != 0) | (b0 == SGG_OOB);
        const unsigned off = bad ? SGG_OOB : b0 + dm_rel[k];
        // (instructions past the plane's last one, and past this wave's share, do not exist: all lanes invalid, sent to the scratch)
        const bool real = k < DPW && hh * DPW + k < DPP;
        unsigned char* d = real ? dst + k * 1024 : lds + sizeof(lds) - 1024;
        // (the plane's byte offset as SCALAR offset: the immediate offset field would move the LDS address too, scripts/ubench/dma_oob.hip)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (pc_lds_ptr)d, 16, off, plane * 64, 0, 0);
      }
      if (++s_cc == nch) {        // advance to this workgroup's next tile
        s_cc = 0;
        s_tile += tstride;
#pragma unroll
        for (int j = 0; j < PC_NB; ++j) {
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
    if constexpr (DMAP) {
#pragma unroll
      for (int k = 0; k < PQ; ++k) {
        const int slot = 16 * (DPW * (pw & 1) + k) + (lane >> 2);      // slot inside the plane: [block][row 0..9][pitch 12]
        const int blk = slot / 120 < PC_NB ? slot / 120 : PC_NB - 1, r = slot - 120 * blk;
        const int ry = r / PC_PITCH, rx = r - ry * PC_PITCH;
        const bool valid = k < DPW && slot < PC_NB * 120 && rx < 10;
        const int piece = (lane & 3) ^ pc_sw(ry, rx);
        dm_rel[k] = (unsigned)((ry * p.in_rs + rx * p.in_ps) * 4 + piece * 16);
        dm_meta[k] = ((ry == 0) | ((ry == 9) << 1) | ((rx == 0) << 2) | ((rx == 9) << 3)) | (blk << 4) | ((int)valid << 6);
      }
    }
    // ---- weight fragments: this wave moves a quarter (WQ 1-KiB pieces) of the tap's [n-tile 16][plane] pieces
    const unsigned w_slab = (unsigned)(p.N >> 5) * 4096u;
    const unsigned w_lane = (unsigned)(n0 >> 5) * 4096u + (unsigned)pw * (unsigned)(PC_SLOTB / 4) + (unsigned)lane * 16u;
    int d_cc = 0, d_tap = 0, d_slot = 0;
    auto dma_issue = [&]() __attribute__((always_inline)) {
      const unsigned base = (unsigned)(d_tap * nch + d_cc) * w_slab + w_lane;
      unsigned char* dst = ring + d_slot * PC_SLOTB + pw * (PC_SLOTB / 4);
#pragma unroll
      for (int q = 0; q < WQ; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (pc_lds_ptr)(dst + q * 1024), 16, base + (unsigned)(q * 1024), 0, 0, 0);
      if (++d_tap == 9) {
        d_tap = 0;
        if (++d_cc == nch) d_cc = 0;
      }
      d_slot = (d_slot + 1) & (PC_D - 1);
    };

    // ---- prologue: the first PC_D taps of weights, the patches of chunks 0 (written) and 1 (in flight) ---------------------------
#pragma unroll
    for (int k = 0; k < PC_D; ++k) dma_issue();
    int cur = 0;                                // patch buffer the consumers read in the current chunk
    if constexpr (LNP) {
      // lnp_s is filled and read by the 256 producer threads: one extra barrier (matched by the consumers)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    auto full_pass = [&](auto s_c, auto j_c, unsigned char* dst) __attribute__((always_inline)) {
      if constexpr (LNP) {
        stage_ln_half(s_c, j_c, std::integral_constant<int, 0>{});
        stage_ln_half(s_c, j_c, std::integral_constant<int, 1>{});
      }
      stage_write_pass(s_c, j_c, dst);
    };
    if constexpr (DMAP) {
      patch_dma(lds);                                        // chunk 0 (chunk c + 1 follows at tap 0 of chunk c, into the other buffer)
      __builtin_amdgcn_s_waitcnt(pc_vmcnt(0));               // vmcnt(0)
    } else {
      stage_load(std::integral_constant<int, 0>{});            // chunk 0
      full_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, lds);
      full_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, lds);
      full_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, lds);
      full_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, lds);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      stage_load(std::integral_constant<int, 1>{});            // chunk 1: in flight across the prologue barrier
    }
    __builtin_amdgcn_s_barrier();               // barrier "-1": patch 0 and the fragments of taps 0 .. 3 are in LDS

    // one tap of chunk c (register-set parity S = c & 1): T = 0 .. 8
    auto tap = [&](auto s_c, auto t_c) __attribute__((always_inline)) {
      constexpr int S = decltype(s_c)::value, T = decltype(t_c)::value;
      // (timing-only ablation builds, wrong results: -DPC_ABL_NOSTAGE no patch staging after the first, -DPC_ABL_NODMA no weight DMA
      //  after the prologue's, -DPC_ABL_NOEPI no output stores / statistics)
#ifndef PC_ABL_NOSTAGE
      if constexpr (DMAP) {
        // the patch of chunk c + 1 straight into the buffer the consumers left at the last barrier: eight DMAs per wave, older than
        // the weight DMAs of taps g + 3 .. (the same eight entries in the vmcnt queue as the register path's loads: same waits below;
        // the vmcnt(8) of tap 3 retires them, five barriers before the consumers read that buffer)
        if constexpr (T == 0) patch_dma(lds + (cur ^ 1) * PC_PATCHB);
      } else
      if constexpr (T == 0) stage_load(s_c);                                    // chunk c + 2 -> set S (chunk c's data left it a chunk ago)
      // chunk c + 1 (set S ^ 1) -> the buffer the consumers do not read.  Without the LN prologue one pass (25 VALU) per tap at taps
      // 1 .. 4; with it half a pass per tap at taps 1 .. 8: the vector issue port of a SIMD is shared with the consumer wave, whose
      // 16x16x32 MFMAs alone hold it half of the time, and a whole pass of the prologue (16 v_exp) in one tap makes the barrier late
      if constexpr (DMAP) {
      } else if constexpr (LNP) {
        if constexpr (T >= 1) {
          constexpr int J = (T - 1) >> 1, HF = (T - 1) & 1;
          stage_ln_half(std::integral_constant<int, S ^ 1>{}, std::integral_constant<int, J>{}, std::integral_constant<int, HF>{});
          if constexpr (HF == 1) stage_write_pass(std::integral_constant<int, S ^ 1>{}, std::integral_constant<int, J>{}, lds + (cur ^ 1) * PC_PATCHB);
        }
      } else if constexpr (T >= 1 && T <= 4) {
        stage_write_pass(std::integral_constant<int, S ^ 1>{}, std::integral_constant<int, T - 1>{}, lds + (cur ^ 1) * PC_PATCHB);
      }
#endif
      // fragments of tap g + 1 (issued behind barrier g - 3) must have landed.  Younger than them in this wave's vmcnt queue: the
      // DMAs of taps g + 2, g + 3 (8 instructions) and - at taps 0 .. 2 only, later the patch loads of tap 0 are older - 8 patch loads
      // (the BUILTIN, not inline asm: the compiler's own wait insertion must see these waits - with asm it believes every DMA since
      //  the kernel's start is still in flight, its count outgrows the 6-bit counter and it falls back to vmcnt(0) at the chunk loop's
      //  head and in front of the patch writes: a full L2 round trip in the producers once per chunk, the barrier late by as much)
      // (NB = 2: vmcnt(16) / vmcnt(8); NB = 4: 2 weight DMAs per tap and 16 patch DMAs per chunk: vmcnt(20) / vmcnt(4))
      if constexpr (T <= 2) __builtin_amdgcn_s_waitcnt(pc_vmcnt(2 * WQ + PLQ));
      else __builtin_amdgcn_s_waitcnt(pc_vmcnt(2 * WQ));
      if constexpr (T == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next patch is written
      __builtin_amdgcn_s_barrier();             // barrier g
#ifndef PC_ABL_NODMA
      dma_issue();                              // tap g + 4 into slot g % 4 (the consumers have finished reading tap g)
#endif
    };
    auto chunk = [&](auto s_c) __attribute__((always_inline)) {
      tap(s_c, std::integral_constant<int, 0>{}); tap(s_c, std::integral_constant<int, 1>{}); tap(s_c, std::integral_constant<int, 2>{});
      tap(s_c, std::integral_constant<int, 3>{}); tap(s_c, std::integral_constant<int, 4>{}); tap(s_c, std::integral_constant<int, 5>{});
      tap(s_c, std::integral_constant<int, 6>{}); tap(s_c, std::integral_constant<int, 7>{}); tap(s_c, std::integral_constant<int, 8>{});
      cur ^= 1;
    };
    for (int tile = mt_begin; tile < mt_end; tile += tstride) {
      for (int cc = 0; cc < nch; cc += 2) {
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the last DMAs target this workgroup's LDS: they must not outlive it)
    return;
  }

  // ===================================================================================================================
  // CONSUMERS (waves 0-3): NB = 2: block wblk = wave >> 1, column half wn0 = (wave & 1) * 64;  NB = 4: block wblk = wave, all 64 columns.
  // v_mfma_f32_16x16x32_f16 / _bf16: a whole 32-channel chunk per instruction (K = 32), 4 x 4 tiles of 16 pixels x 16 columns per wave,
  // 48 MFMAs of 16 cycles per tap.  Same FLOPs per cycle as the 32x32x16 shape, but the power-limited chip holds a higher clock on
  // it: +6 % measured on this kernel with both shapes issued on the same operands (profiles/r03_halo_pc_mfma_shape.log).
  // A operand of tile i: lane l = (c4 = l >> 4, p = l & 15) holds pixel 16 i + p (block row 2 i + (p >> 3), column p & 7), channels
  // 8 c4 .. 8 c4 + 7; B operand of tile j: column 16 j + p, the same channels; D: column p, pixels 16 i + 4 c4 + r in register r.
  // ===================================================================================================================
  constexpr int TI = 4, TJ = 4;
  const int wblk = NB == 2 ? wave >> 1 : wave, wn0 = NB == 2 ? (wave & 1) * 64 : 0;
  const int l16 = lane & 15, c4 = lane >> 4;
  const int pyl = l16 >> 3, pxl = l16 & 7;
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[2][TI][PC_P], rb[2][TJ][PC_P];
  int cur = 0, slot = 0;       // patch buffer / ring slot of the tap whose operands were read LAST

  // operands of one tap -> register set `buf`: A from the resident patch (shifted slots), B from the ring slot.  The addresses are
  // computed BEFORE the tap's barrier (addr_ops: VALU in the shadow of the first half's MFMAs), the reads issued behind it.
  struct OpAddr {
    const unsigned char* row[TI];
    const unsigned char* bw;
  };
  auto addr_ops = [&](int tp, int patch_buf, int ring_slot) __attribute__((always_inline)) {
    OpAddr o;
    const int kh = tp / 3, kw = tp % 3;
    const int dyy = p.flip ? 2 - kh : kh, dxx = p.flip ? 2 - kw : kw;
    int pyv = pyl, pxv = pxl;
    asm volatile("" : "+v"(pyv), "+v"(pxv));       // (keeps the per-tap addresses out of the loop-invariant hoisting, conv_halo.hip)
    const unsigned char* patch_w = lds + patch_buf * PC_PATCHB + wblk * PC_BLKB;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int ry = 2 * i + pyv + dyy, rx = pxv + dxx;
      o.row[i] = patch_w + (ry * PC_PITCH + rx) * 64 + ((c4 ^ pc_sw(ry, rx)) << 4);
    }
    int lv = lane;
    asm volatile("" : "+v"(lv));
    o.bw = ring + ring_slot * PC_SLOTB + (wn0 >> 4) * 2048 + lv * 16;
    return o;
  };
  auto read_ops = [&](auto buf_c, const OpAddr& o) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_c)::value;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int pp = 0; pp < PC_P; ++pp) a[buf][i][pp] = *reinterpret_cast<const u32x4*>(o.row[i] + pp * PC_PLANEB);
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int pp = 0; pp < PC_P; ++pp) rb[buf][j][pp] = *reinterpret_cast<const u32x4*>(o.bw + j * 2048 + pp * 1024);
  };
  // MFMAs of row tiles 2 * half and 2 * half + 1 (2 x 4 tiles x 3 = 24 instructions): hi * lo + lo * hi + hi * hi per (i, j)
  auto mma_half = [&](auto par_c, auto half_c) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value, half = decltype(half_c)::value;
#pragma unroll
    for (int i = 2 * half; i < 2 * half + 2; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        f32x4 d = acc[i][j];
        d = mfma16k32<HALF>(a[par][i][1], rb[par][j][0], d);
        d = mfma16k32<HALF>(a[par][i][0], rb[par][j][1], d);
        d = mfma16k32<HALF>(a[par][i][0], rb[par][j][0], d);
        acc[i][j] = d;
      }
  };
  // one tap: [addresses of the next tap's operands | MFMAs of the first two row tiles]  barrier g  [16 LDS reads | MFMAs of the others].
  // One MFMA wave per SIMD: nothing else feeds the matrix pipe while this wave issues other instructions, so the address arithmetic
  // is spread over the first 24 MFMAs and the reads over the second 24 (which use the OTHER register set).
  auto tap = [&](auto par_c, auto t_c) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value, T = decltype(t_c)::value;
    SGG_PRIO_HI();
    const int nslot = (slot + 1) & (PC_D - 1), ncur = T == 8 ? cur ^ 1 : cur;
    const OpAddr o = addr_ops(T == 8 ? 0 : T + 1, ncur, nslot);
    mma_half(par_c, std::integral_constant<int, 0>{});
#if PC_INTERLEAVE
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);       // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);       // VALU (addresses)
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();               // barrier g: tap g + 1 (and, at T == 8, the next patch) is in LDS
    slot = nslot;
    cur = ncur;
    read_ops(std::integral_constant<int, par ^ 1>{}, o);
#if PC_INTERLEAVE
    mma_half(par_c, std::integral_constant<int, 1>{});
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);       // MFMA
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#else
    __builtin_amdgcn_sched_barrier(0);
    mma_half(par_c, std::integral_constant<int, 1>{});
#endif
    SGG_PRIO_LO();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto chunk = [&](auto par0_c) __attribute__((always_inline)) {
    constexpr int par0 = decltype(par0_c)::value;
#define PC_TAP(T) tap(std::integral_constant<int, (par0 + T) & 1>{}, std::integral_constant<int, T>{})
    PC_TAP(0); PC_TAP(1); PC_TAP(2); PC_TAP(3); PC_TAP(4); PC_TAP(5); PC_TAP(6); PC_TAP(7); PC_TAP(8);
#undef PC_TAP
  };

  // ---- output addressing: 16-byte stores.  After the quad transpose of a tile's four accumulator registers lane (c4, g = l16 >> 2,
  // k = l16 & 3) holds pixel (block row 2 i + (c4 >> 1), column 4 (c4 & 1) + k), channels 16 j + 4 g .. + 3
  int o_grow, o_bx;
  {
    const int beta = mt_begin * PC_NB + wblk;
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
  float bias_v[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j) bias_v[j] = p.bias ? p.bias[n0 + wn0 + j * 16 + l16] : 0.f;
  constexpr int WN = 64;

  auto epilogue = [&](int tile) __attribute__((always_inline)) {
    const int beta = tile * PC_NB + wblk;
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
#ifdef PC_ABL_NOEPI
    if (p.B < 0) {
#else
    if (live) {
#endif
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
#ifdef PC_ABL_NOEPI
    if (p.B < 0) {
#else
    if (p.tile_stats) {
#endif
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

  if constexpr (LNP) __builtin_amdgcn_s_barrier();      // (matches the producers' barrier behind the lnp_s fill)
  __builtin_amdgcn_s_barrier();                          // barrier "-1"
  read_ops(std::integral_constant<int, 0>{}, addr_ops(0, 0, 0));         // tap 0: patch buffer 0, ring slot 0
  // (C % 64 == 0: an even number of chunks; the register-set parity flips once per chunk of nine taps)
  for (int tile = mt_begin; tile < mt_end; tile += tstride) {
    for (int cc = 0; cc < nch; cc += 2) {
      chunk(std::integral_constant<int, 0>{});
      chunk(std::integral_constant<int, 1>{});
    }
    epilogue(tile);
  }
}

// ---- host --------------------------------------------------------------------------------------------------------------------
// 1 if the producer / consumer kernel serves this launch: the two-piece modes (precision 2 / 3), 128-column tiles, an even number of
// 32-channel chunks (C % 64 == 0).  -DSGG_HALO_PC=0 builds never use it.
#ifndef SGG_HALO_PC
#define SGG_HALO_PC 1
#endif
#ifndef SGG_HALO_PC_DMA
#define SGG_HALO_PC_DMA 1      // 0: a pre-split source is staged through registers (no arithmetic) like an f32 one
#endif
#ifndef SGG_HALO_PC64
#define SGG_HALO_PC64 1        // 0: no four-block (64-column) tiles
#endif
int sgg_halo_pc_applicable(int C, int N, int precision) {
  return SGG_HALO_PC && (precision == 2 || precision == 3) && N % 128 == 0 && C % 64 == 0 && C <= 512;
}
// ... and the four-block form (64-column tiles, round 5): N % 64 == 0 but not 128, C % 64 == 0, precision 2, and only with a PRE-SPLIT
// source (the patch comes by LDS-DMA: there is no register-staged variant of it)
int sgg_halo_pc64_applicable(int C, int N, int precision) {
  return SGG_HALO_PC && SGG_HALO_PC64 && SGG_HALO_PC_DMA && precision == 2 && N % 64 == 0 && N % 128 != 0 && C % 64 == 0 && C <= 512;
}

void sgg_halo_pc_launch(const HaloParams& p_, int precision, hipStream_t st) {
  HaloParams p = p_;
  const bool four = p.N % 128 != 0;            // (callers checked sgg_halo_pc64_applicable and the pre-split source)
  const int nb = four ? 4 : 2, bn = 256 / nb;
  const int mtiles = sgg_cdiv(p.nblk, nb), ntn = p.N / bn;
  int per_xcd = sgg_cdiv(mtiles, 8) * ntn;       // (tile, n-tile) pairs an XCD owns
  const int cus = sgg_persist_cus(p.cu_cap);
  int gx = per_xcd < cus ? per_xcd : cus;          // one workgroup on each of its (32) CUs
  gx = sgg_cdiv(gx, ntn) * ntn;
  p.gx = gx;
  const dim3 grid((unsigned)(8 * gx)), blk(512);
  const bool half = precision == 2;
  if (four) {
    hipLaunchKernelGGL((conv_halo3_pc_kernel<true, false, true, 4>), grid, blk, 0, st, p);
    return;
  }
  if (p.src_s16 && half && !p.ln_stats && SGG_HALO_PC_DMA) {      // pre-split source: the patch by LDS-DMA too
    hipLaunchKernelGGL((conv_halo3_pc_kernel<true, false, true>), grid, blk, 0, st, p);
    return;
  }
  if (p.ln_stats) {
    if (half) hipLaunchKernelGGL((conv_halo3_pc_kernel<true, true>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((conv_halo3_pc_kernel<false, true>), grid, blk, 0, st, p);
  } else {
    if (half) hipLaunchKernelGGL((conv_halo3_pc_kernel<true, false>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((conv_halo3_pc_kernel<false, false>), grid, blk, 0, st, p);
  }
}
