// Winograd F(4,3) (1-D, along W) 3x3 / stride 1 / pad 1 convolution of G lock-step experts as a ROW-BLOCK kernel: the dominant kernel of
// the TRBA x 6 router phase (sixteen 512 -> 512 layers on 4 x 65 maps per expert; reference modules/feature_extraction.py:165-199,262-294)
// and of loop A's trained convolutions (forward and data gradient; il_modules/mrn.py:260-261).
//
// Same operands and arithmetic as conv_x3_kernel<2,4,2,1,false,3,4> (conv_x3.hip, which stays the fallback for H % 4 != 0):
//   v  [G][B][H][Wq][6][Cin/32][128 B]   B^T d per group of 4 output columns (Wq = ceil(W/4)), HL32 lines [hi fp16 x 32 | lo fp16 x 32]
//   u  [G][Cout][6][Cin/32][3 ky][128 B] G g per kernel row, power-of-two prescaled
//   y  [G][B][H][W][Cout] fp32 = A^T (sum over Cin, ky of v (.) u) + bias, products as split-fp16 x3 (lo*hi + hi*lo + hi*hi) on
//      v_mfma_f32_32x32x16_f16 with fp32 accumulation, component sums T_m folded into the four output accumulators in registers.
//
// What is different is WHO shares WHAT.  The x3 kernel treats the layer as a 3 x 1 convolution over GEMM rows (one output row of the image
// per tile): every (component, channel block, ky) K-step stages 128 activation lines + 128 weight lines for 12 MFMAs per wave -- one
// ds_read_b128 per MFMA, 1.33 KiB of LDS traffic per MFMA, and on zero operands it sits on that staging floor (round 3: 2.16 ms without
// MFMAs vs 2.18 with).  Here a workgroup owns 64 POSITIONS (b, column group) x FOUR output rows x 64 output channels, and a
// (component, channel block) step stages the <= 6 input rows those four output rows touch ONCE (4 for the 4-row maps) plus the three
// kernel rows of the weights:
//   * an activation line serves up to three (ky, output row) products from LDS instead of being staged three times;
//   * a wave (4 output rows x 32 positions x 32 channels: one 32 x 32 accumulator per output row + 16 output accumulators) reads 2 A
//     fragments per input row and 6 B fragments per half step: 28 ds_read_b128 for 60 MFMAs (0.47 per MFMA instead of 1);
//   * the kernel rows that fall into the vertical padding are simply absent (10 of 12 block products on 4-row maps): no interior /
//     border tile classes, the transformed activation is read once.
// The price is registers: 4 x 4 output accumulators + 4 component accumulators = 320 per lane, so a workgroup is four waves, ONE per SIMD
// (the 512-register pattern), with the MFMA stream skewed across the step barrier (the last input row's products of step s issue after
// the barrier, covering the DMA issue and the first fragment reads of step s + 1).  LDS: 2 stages x (6 x 64 + 3 x 64) lines = 144 KiB.
#include "common.hpp"
#include "conv_wino.hpp"
#include <stdlib.h>

namespace {

typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int NPOS = 64, NCH = 64;                       // positions / output channels per workgroup
constexpr int LDS_BYTES = 144 * 1024;                   // activation ring + weight ring (sized per row-block class in wino_rows_tile)

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t r, unsigned char* l, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
#ifdef MRN_WPROBE_NO_MFMA
  f32x16 r = c;                 // (what-if probes MRN_WPROBE_*: never in the product build)
  r[0] += __builtin_bit_cast(float, a[0] ^ b[0]);
  return r;
#endif
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16v8*>(&a), *reinterpret_cast<const f16v8*>(&b), c, 0, 0, 0);
}

// DENSE = 2: the plain 16-bit operands are bfloat16 (the literal "bf16" of BASELINE config 2: a comparison instantiation of the reduced mode)
__device__ __forceinline__ f32x16 mma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16v8*>(&a), *reinterpret_cast<const bf16v8*>(&b), c, 0, 0, 0);
}

// A^T of F(4,3) with the inverse row scales of the packed weight transform (conv_x3.hip: WinoAT<4>, pack_weight_wino_hl32_kernel)
__constant__ float kWinoAT[6][4] = {{0.25f, 0.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f, -0.5f},
                                    {0.5f, 1.f, 2.f, 4.f}, {0.5f, -1.f, 2.f, -4.f}, {0.f, 0.f, 0.f, 1.f}};

// scheduling pattern of one row-op (see the main loop): groups the scheduler fills in this order, empty ones are skipped
#define WINO_SGB(mask) __builtin_amdgcn_sched_group_barrier(mask, 1, 0);
#define WINO_INTERLEAVE_1 WINO_SGB(0x008) WINO_SGB(0x100) WINO_SGB(0x020)      // one MFMA, one DS read, one VMEM read (LDS-DMA piece)
#define WINO_INTERLEAVE_3 WINO_INTERLEAVE_1 WINO_INTERLEAVE_1 WINO_INTERLEAVE_1

#ifdef MRN_WPROBE_TIMING
// what-if / timing probes (never in the product build): per-wave shader-clock totals of the main loop's phases, summed over all waves
__device__ unsigned long long g_wino_dbg[8];
extern "C" __attribute__((visibility("default"))) int mrn_wino_dbg_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_dbg), sizeof(g_wino_dbg)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wino_dbg), z, sizeof(z));
  }
  return 0;
}
#define WTICK(var) const long var = __builtin_readcyclecounter()
#else
#define WTICK(var)
#endif

// TOP / BOT: the row block has an input row above / below its four output rows inside the image (slot 0 / slot 5)
// DENSE: the reduced-precision operand layout (one fp16 product per term): a 128-byte line holds 64 CHANNELS as plain fp16 (p.Cb = Cin / 64)
// instead of [hi | lo] of 32.  The staging geometry is unchanged -- the "hi" half-line is channels 0 .. 31, the "lo" half-line channels
// 32 .. 63 -- and a row-op multiplies matching halves (Bh.Ah + Bl.Al: two MFMAs per (kernel row, output row) for 64 channels) where the
// split form takes the three cross products of 32: a third of the MFMAs on half the bytes.
template <int TOP, int BOT, int DENSE>
__device__ __forceinline__ void wino_rows_tile(const WinoRowsParams& p, unsigned char* lds, int g, int pos0, int oy0, int n0, int tile_m) {
  constexpr int LO = 1 - TOP, HI = 4 + BOT, NR = HI - LO + 1;      // input-row slots present: slot = iy - oy0 + 1
#ifdef MRN_WPROBE_TIMING
  const long dbg_tile0 = __builtin_readcyclecounter();
#endif
  // LDS: the activation lines run through a ring of ADEPTH stages (4-row maps: three -- the lines come from HBM, two steps of look-ahead),
  // the weight lines (L2 / Infinity Cache hits) through two
  constexpr int ADEPTH = NR == 4 ? 3 : 2;
  constexpr int A_STAGE = NR * NPOS * 128, B_STAGE = 3 * NCH * 128, B_BASE = ADEPTH * A_STAGE;
  static_assert(B_BASE + 2 * B_STAGE <= LDS_BYTES, "LDS budget");
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int Cb = p.Cb, nsteps = 6 * Cb;                             // step = component * Cb + channel block
  const int BWq = p.B * p.Wq;
  const int a_pitch = 6 * Cb * 128;                                 // bytes between two positions' line blocks
  const int w_pitch = 18 * Cb * 128;                                // bytes per output channel

  // ---- DMA geometry: an instruction moves 8 lines (64 lanes x 16 B); wave w owns line groups j = w and w + 4 of every input-row slot
  // and of every kernel row.  LDS is written lane-linearly, so the XOR swizzle of the 16-byte chunk is applied to the SOURCE chunk.
  const int lrow = lane >> 3, lch = lane & 7;
  int apos[2], brow[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int r = (wave + 4 * jj) * 8 + lrow;                       // line inside the 64-line slot
    const int coff = (lch ^ ((r >> 1) & 7)) << 4;
    const int pos = pos0 + r;
    const int b = pos / p.Wq, q = pos - b * p.Wq;
    apos[jj] = pos < BWq ? (b * p.H * p.Wq + q) * a_pitch + coff : (int)0x80000000;      // (beyond the descriptor: zeros)
    const int n = min(n0 + r, p.N - 1);                             // channels beyond N re-read the last one (never stored)
    brow[jj] = n * w_pitch + coff;
  }
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v + (long)g * p.v_gstride), 0, p.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.u + (long)g * p.u_gstride), 0, (int)p.u_gstride, 0x00020000);
  const int row_bytes = p.Wq * a_pitch;                             // one image row of one sample
  // The DMAs of a step are a list of NDMA items -- the six weight pieces first, then two per input-row slot -- and every row-op of the
  // main loop issues DMA_PER_OP of them: all four waves issuing a step's pieces at once right after the barrier queue up in front of
  // the CU's one address unit (measured: 14 pieces in a burst cost 1500 cycles of issue per wave, a third of the step).
  constexpr int NDMA = 6 + 2 * NR, DMA_PER_OP = 2;
  static_assert((NDMA + DMA_PER_OP - 1) / DMA_PER_OP <= 2 * NR - 1, "a step's DMAs are issued before its last row-op");
  auto issue_item = [&](int item, int sa, unsigned char* a_st, int sb, unsigned char* b_st) {
    if (item < 6) {                                                 // weights of step sb: (ky, jj)
      const int ky = item >> 1, jj = item & 1;
#ifdef MRN_WPROBE_NO_DMA_B
      if (sb < 2)
#endif
      dma16(wr, b_st + (ky * NCH + (wave + 4 * jj) * 8) * 128, brow[jj], (sb * 3 + ky) * 128);
    } else {                                                        // activation lines of step sa: (slot, jj)
      const int si = (item - 6) >> 1, jj = item & 1;
      const int so = (oy0 - 1 + LO + si) * row_bytes + sa * 128;
#ifdef MRN_WPROBE_NO_DMA_A
      if (sa < 3)
#endif
      dma16(xr, a_st + (si * NPOS + (wave + 4 * jj) * 8) * 128, apos[jj] + so, 0);
    }
  };

  // ---- fragment offsets: row = lane & 31 of a 32-line block, logical chunk = plane * 4 + ks * 2 + (lane >> 5)
  const int key = (lane >> 1) & 7, rb = (lane & 31) * 128;
  int foff[2][2];                                                   // [plane][ks]
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[pl][ks] = rb + (((pl * 4 + ks * 2 + (lane >> 5)) ^ key) << 4);
  const int abase = wm * 32 * 128, bbase = wn * 32 * 128;          // (relative to the activation / weight stage)

  // accumulators: MFMA rows = output channels (weights are the A operand), columns = positions -- a lane then holds 4 consecutive
  // channels of ONE position per register quad, and the epilogue stores 16 bytes per lane
  f32x16 acc[4], Y[4][4];                                           // acc[oy]: component sum T_m; Y[r][oy]: output column 4 q + r
#pragma unroll
  for (int o = 0; o < 4; ++o) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int e = 0; e < 16; ++e) Y[r][o][e] = 0.f;
  }
  u32x4 Ah[2], Al[2], Bh[2][3], Bl[2][3];
  auto read_A = [&](const unsigned char* a_st, int slot, int ks, int par) {
    Al[par] = *reinterpret_cast<const u32x4*>(a_st + abase + (slot - LO) * NPOS * 128 + foff[1][ks]);
    Ah[par] = *reinterpret_cast<const u32x4*>(a_st + abase + (slot - LO) * NPOS * 128 + foff[0][ks]);
  };
  auto read_B = [&](const unsigned char* b_st, int ks) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      Bh[ks][ky] = *reinterpret_cast<const u32x4*>(b_st + bbase + ky * NCH * 128 + foff[0][ks]);
      Bl[ks][ky] = *reinterpret_cast<const u32x4*>(b_st + bbase + ky * NCH * 128 + foff[1][ks]);
    }
  };
  // the products of one input row (slot = iy - oy0 + 1, iy = oy + ky - 1  =>  block-local output row o = slot - ky); consecutive MFMAs
  // target different accumulators
  auto row_products = [&](int slot, int ks, int par) {
    if constexpr (DENSE) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int o = slot - ky;
        if (o >= 0 && o <= 3) acc[o] = DENSE == 2 ? mma_bf16(Bh[ks][ky], Ah[par], acc[o]) : mma(Bh[ks][ky], Ah[par], acc[o]);
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int o = slot - ky;
        if (o >= 0 && o <= 3) acc[o] = DENSE == 2 ? mma_bf16(Bl[ks][ky], Al[par], acc[o]) : mma(Bl[ks][ky], Al[par], acc[o]);
      }
      return;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int o = slot - ky;
      if (o >= 0 && o <= 3) acc[o] = mma(Bh[ks][ky], Al[par], acc[o]);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int o = slot - ky;
      if (o >= 0 && o <= 3) acc[o] = mma(Bl[ks][ky], Ah[par], acc[o]);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int o = slot - ky;
      if (o >= 0 && o <= 3) acc[o] = mma(Bh[ks][ky], Ah[par], acc[o]);
    }
  };

  // ---- pipeline: the row-ops (ks, slot) of a step run with the fragment reads of row-op r + 1 issued before the MFMAs of row-op r; the
  // last row-op of a step issues its MFMAs AFTER the step barrier.  During step s the weights of step s + 1 and the activation lines
  // of step s + ADEPTH - 1 are fetched (past the end the last step is re-fetched: every step issues the same DMAs, so the counted
  // s_waitcnt below is a compile-time immediate).
  constexpr int A_AHEAD = ADEPTH - 1;
  constexpr int WAIT_BOUNDARY = ADEPTH == 3 ? (((2 * NR) & 15) | (7 << 4) | (0 << 8) | (((2 * NR) >> 4) << 14))   // vmcnt(2 NR): the next-but-one step's lines may fly
                                            : 0;                                                                    // vmcnt(0) lgkmcnt(0)
#pragma unroll
  for (int pre = 0; pre < A_AHEAD; ++pre)
#pragma unroll
    for (int item = 6; item < NDMA; ++item) issue_item(item, min(pre, nsteps - 1), lds + pre * A_STAGE, 0, nullptr);
#pragma unroll
  for (int item = 0; item < 6; ++item) issue_item(item, 0, nullptr, 0, lds + B_BASE);
  __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));                 // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  read_B(lds + B_BASE, 0);
  read_A(lds, LO, 0, 0);
  int cbi = 0, comp = 0;
  int a_cur = 0;                                                    // ring position of the current step's activation lines
#ifdef MRN_WPROBE_TIMING
  long dbg_wait = 0, dbg_bar = 0, dbg_fold = 0;
  const long dbg_t0 = __builtin_readcyclecounter();
#endif
  for (int s = 0; s < nsteps; ++s) {
    const unsigned char* a_st = lds + a_cur * A_STAGE;
    const unsigned char* b_st = lds + B_BASE + (s & 1) * B_STAGE;
    const int a_fill = a_cur + A_AHEAD >= ADEPTH ? a_cur + A_AHEAD - ADEPTH : a_cur + A_AHEAD;      // stage of step s + A_AHEAD
    unsigned char* a_dst = lds + a_fill * A_STAGE;
    unsigned char* b_dst = lds + B_BASE + ((s + 1) & 1) * B_STAGE;
    const int sa = min(s + A_AHEAD, nsteps - 1), sb = min(s + 1, nsteps - 1);
    const int a_next = a_cur + 1 == ADEPTH ? 0 : a_cur + 1;
#pragma unroll
    for (int r = 0; r < 2 * NR; ++r) {
      const int ks = r / NR, slot = LO + r % NR, par = r & 1;
#pragma unroll
      for (int item = r * DMA_PER_OP; item < (r + 1) * DMA_PER_OP; ++item)
        if (item < NDMA) issue_item(item, sa, a_dst, sb, b_dst);
      if (r + 1 < 2 * NR) {
        const int ks2 = (r + 1) / NR, slot2 = LO + (r + 1) % NR;
        if (ks2 != ks) read_B(b_st, ks2);
        read_A(a_st, slot2, ks2, par ^ 1);
      } else {
        // step boundary (also after the last step, where the reads below fetch a stale stage that nothing uses: a branch here makes the
        // compiler wait for ALL of them before the MFMAs that follow -- its wait counts are merged over both paths): own fragment reads of this step are complete, own DMAs of the next step's operands have landed (the younger
        // ones -- activation lines of the step after it -- may still fly); after the barrier everyone's have, and this step's stages are free
        WTICK(tk0);
        __builtin_amdgcn_s_waitcnt(WAIT_BOUNDARY);
        WTICK(tk1);
#ifndef MRN_WPROBE_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        WTICK(tk2);
#ifdef MRN_WPROBE_TIMING
        dbg_wait += tk1 - tk0;
        dbg_bar += tk2 - tk1;
#endif
        read_B(b_dst, 0);
        read_A(lds + a_next * A_STAGE, LO, 0, par ^ 1);
      }
      // Software pipeline, pinned: left alone the scheduler sinks every fragment read to just in front of its first MFMA (one register
      // set, the LDS latency exposed once per input row).  And a wave that is ALONE on its SIMD only hides what sits BETWEEN its
      // MFMAs: the DMA pieces, fragment reads and address arithmetic of row-op r + 1 issued as one clump in front of the MFMA group of
      // row-op r ran with the matrix pipe idle (in-kernel clocks: 2700 cycles per step against 1920 of MFMAs, barrier and counted
      // waits 100).  So one row-op is one scheduling region whose order is given as [1 MFMA, 1 LDS read, 1 DMA piece] x 9.
#ifdef MRN_WPROBE_CLUMPED
      __builtin_amdgcn_sched_barrier(0);
#endif
      row_products(slot, ks, par);
#ifndef MRN_WPROBE_CLUMPED
      if (r + 1 < 2 * NR) {
        WINO_INTERLEAVE_3 WINO_INTERLEAVE_3
        if constexpr (!DENSE) { WINO_INTERLEAVE_3 }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    a_cur = a_next;
#ifdef MRN_WPROBE_NO_FOLD
    if (s == nsteps - 1) {
#else
    if (++cbi == Cb) {
#endif
      // component `comp` is complete: Y_r += A^T[r][comp] * T, T = 0
      WTICK(tf0);
      float cf[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) cf[r] = kWinoAT[comp][r];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (cf[r] != 0.f) {
#pragma unroll
          for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[r][o][e] = fmaf(cf[r], acc[o][e], Y[r][o][e]);
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;
      cbi = 0;
      ++comp;
#ifdef MRN_WPROBE_TIMING
      asm volatile("s_nop 0" ::: "memory");
      dbg_fold += __builtin_readcyclecounter() - tf0;
#endif
    }
  }
  __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));                 // vmcnt(0): the re-fetches of the tail are not left in flight
#ifdef MRN_WPROBE_TIMING
  const long dbg_t1 = __builtin_readcyclecounter();
#endif

  // ---- epilogue: scale, bias, BatchNorm partial statistics, activation, store.  Lane = one position (b, column group q); register quad
  // j4 of an accumulator = channels c0 + 8 j4 .. + 3 with c0 = n0 + 32 wn + 4 (lane >> 5)
  const float osc = (p.out_scale ? p.out_scale[g * 2 + 1] : 1.f) * (p.x_scale ? p.x_scale[1] : 1.f);
  const int c0 = n0 + wn * 32 + 4 * (lane >> 5);
  f32x4 bias4[4];
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4) {
    bias4[j4] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && c0 + 8 * j4 < p.N) bias4[j4] = *reinterpret_cast<const f32x4*>(p.bias + (long)g * p.N + c0 + 8 * j4);
  }
  float* yg = p.y + (long)g * p.y_gstride;
  const int pos = pos0 + wm * 32 + (lane & 31);
  const bool pos_ok = pos < BWq;
  const int b = pos / p.Wq, q = pos - b * p.Wq;
  f32x4 csum[4], csq[4];
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4) csum[j4] = csq[j4] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (pos_ok && !p.pool) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const long pix0 = ((long)b * p.H + oy0 + o) * p.W + 4 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * q + r >= p.W) continue;
        float* dst = yg + (pix0 + r) * p.N + c0;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          if (c0 + 8 * j4 >= p.N) continue;
          f32x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = Y[r][o][4 * j4 + i] * osc + bias4[j4][i];
          csum[j4] += v;
          csq[j4] += v * v;
          if (p.act == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu_nan(v[i]);
          }
          *reinterpret_cast<f32x4*>(dst + 8 * j4) = v;
        }
      }
    }
  } else if (pos_ok) {
    // POOLED form (p.pool: the 2x2 / stride-2 max-pool behind BatchNorm + ReLU taken here, see conv_patch.hip): a lane holds the 4 x 4
    // pixels of its position -- four whole windows per channel.  Per window and channel the maximum of the raw output where the
    // BatchNorm weight is >= 0, the minimum (= -max(-v), exact) where it is negative; the statistics cover every unpooled value.
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    float* ypg = p.y + (long)g * p.B * Ho * Wo * p.N;
    const float* gm = p.gamma ? reinterpret_cast<const float*>(p.gamma[g]) : nullptr;
    const float relu_floor = p.act == 1 ? 0.f : -INFINITY;
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      if (c0 + 8 * j4 >= p.N) continue;
      f32x4 sg = {1.f, 1.f, 1.f, 1.f};
      if (gm) {
        const f32x4 gq = *reinterpret_cast<const f32x4*>(gm + c0 + 8 * j4);
#pragma unroll
        for (int i = 0; i < 4; ++i) sg[i] = gq[i] < 0.f ? -1.f : 1.f;
      }
#pragma unroll
      for (int po = 0; po < 2; ++po) {
        f32x4 mx[2];                             // the two windows of this row pair (the statistics see the values in the unpooled form's order: same bits)
#pragma unroll
        for (int oo = 0; oo < 2; ++oo)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 2 * po + oo;
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = Y[r][o][4 * j4 + i] * osc + bias4[j4][i];
            if (4 * q + r < p.W) {                // (columns beyond the row exist only in the padded last group: W even => whole windows)
              csum[j4] += v;
              csq[j4] += v * v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) mx[r >> 1][i] = (oo | (r & 1)) ? max_nan(mx[r >> 1][i], v[i] * sg[i]) : v[i] * sg[i];
          }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int ox = 2 * q + pr;
          if (ox < Wo) {
            f32x4 out;
#pragma unroll
            for (int i = 0; i < 4; ++i) out[i] = relu_nan(mx[pr][i] * sg[i], relu_floor);
            *reinterpret_cast<f32x4*>(ypg + (((long)b * Ho + (oy0 >> 1) + po) * Wo + ox) * p.N + c0 + 8 * j4) = out;
          }
        }
      }
    }
  }
#ifdef MRN_WPROBE_TIMING
  if (lane == 0) {
    const long dbg_t2 = __builtin_readcyclecounter();
    atomicAdd(&g_wino_dbg[0], (unsigned long long)(dbg_t1 - dbg_t0));      // main loop
    atomicAdd(&g_wino_dbg[1], (unsigned long long)dbg_wait);               // s_waitcnt at the step boundary
    atomicAdd(&g_wino_dbg[2], (unsigned long long)dbg_bar);                // s_barrier
    atomicAdd(&g_wino_dbg[3], (unsigned long long)dbg_fold);               // component folds
    atomicAdd(&g_wino_dbg[4], (unsigned long long)(dbg_t2 - dbg_t1));      // epilogue stores
    atomicAdd(&g_wino_dbg[5], (unsigned long long)(dbg_t0 - dbg_tile0));   // prologue
    atomicAdd(&g_wino_dbg[6], 1ull);
  }
#endif
  if (p.stats) {
    // per channel: sum over the 32 positions of each lane half, then over the two position waves through LDS
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s_ = csum[j4][i], q_ = csq[j4][i];
#pragma unroll
        for (int ofs = 16; ofs > 0; ofs >>= 1) {
          s_ += __shfl_xor(s_, ofs);
          q_ += __shfl_xor(q_, ofs);
        }
        csum[j4][i] = s_;
        csq[j4][i] = q_;
      }
    __syncthreads();                                                // (all DMAs were drained above; LDS is free)
    float* red = reinterpret_cast<float*>(lds);                    // [2 wm][2][NCH]
    if ((lane & 31) == 0) {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = wn * 32 + 8 * j4 + 4 * (lane >> 5) + i;
          red[(wm * 2 + 0) * NCH + c] = csum[j4][i];
          red[(wm * 2 + 1) * NCH + c] = csq[j4][i];
        }
    }
    __syncthreads();
    if (t < NCH && n0 + t < p.N) {
      float* sg = p.stats + ((long)g * p.stats_blocks + tile_m) * 2 * p.N;
      sg[n0 + t] = red[t] + red[2 * NCH + t];
      sg[p.N + n0 + t] = red[NCH + t] + red[3 * NCH + t];
      if (tile_m + p.tiles_m < p.stats_blocks) {                    // the buffer is sized for the x3 kernel's tiling: clear the unused slots
        float* sz = p.stats + ((long)g * p.stats_blocks + tile_m + p.tiles_m) * 2 * p.N;
        sz[n0 + t] = 0.f;
        sz[p.N + n0 + t] = 0.f;
      }
    }
  }
}

template <int DENSE>
__global__ __launch_bounds__(256) void wino_rows_kernel(const WinoRowsParams p) {
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
#ifdef MRN_WPROBE_SKEW
  // what-if probe: every other workgroup of the first round starts MRN_WPROBE_SKEW x 3.4 us late, so that the halves of the chip reach their
  // store epilogues (a 64 MB burst per round when all 256 workgroups arrive together) at different times
  if (blockIdx.x < 256u && ((blockIdx.x >> 3) & 1u))
    for (int i = 0; i < MRN_WPROBE_SKEW; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  // tile order: output-channel tile fastest, so the workgroups that share an XCD's L2 at one time read the same activation lines
  // (block b runs on XCD b % 8; xcd_remap makes consecutive logical tiles share an XCD)
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = lid % p.tiles_n;
  int t2 = lid / p.tiles_n;
  const int rb = t2 % p.row_blocks;
  t2 /= p.row_blocks;
  const int tp = t2 % p.tiles_p;
  const int g = t2 / p.tiles_p;
  const int oy0 = rb * 4, pos0 = tp * NPOS, n0 = tn * NCH;
  const int tile_m = tp * p.row_blocks + rb;
  const bool top = oy0 > 0, bot = oy0 + 4 < p.H;
  if (top) {
    if (bot) wino_rows_tile<1, 1, DENSE>(p, lds, g, pos0, oy0, n0, tile_m);
    else wino_rows_tile<1, 0, DENSE>(p, lds, g, pos0, oy0, n0, tile_m);
  } else {
    if (bot) wino_rows_tile<0, 1, DENSE>(p, lds, g, pos0, oy0, n0, tile_m);
    else wino_rows_tile<0, 0, DENSE>(p, lds, g, pos0, oy0, n0, tile_m);
  }
}

}  // namespace

static int g_rows_mode = -1;      // -1: MRN_WINO_ROWS from the environment (default on); 0 / 1: set by mrn_conv2d_x3_wino_select (A/B runs, tests)
void mrn_wino_rows_select(int mode) { g_rows_mode = mode; }

bool mrn_wino_rows_supported(int H, int R, int Cout) {
  static const bool env_off = getenv("MRN_WINO_ROWS") && atoi(getenv("MRN_WINO_ROWS")) == 0;   // A/B switch: 0 = the x3 kernel's Winograd form
  const bool off = g_rows_mode >= 0 ? g_rows_mode == 0 : env_off;
  return !off && R == 4 && H % 4 == 0 && H >= 4 && Cout >= 32 && Cout % 4 == 0;
}

int mrn_launch_wino_rows(const WinoRowsParams& p0, void* stream) {
  WinoRowsParams p = p0;
  p.tiles_p = ceil_div((long)p.B * p.Wq, NPOS);
  p.row_blocks = p.H / 4;
  p.tiles_n = ceil_div(p.N, NCH);
  p.tiles_m = p.tiles_p * p.row_blocks;
  if (p.stats && p.tiles_m > p.stats_blocks) {
    mrn_set_error("wino_rows: statistics buffer has %d row blocks, the tiling needs %d", p.stats_blocks, p.tiles_m);
    return MRN_ERR_WORKSPACE;
  }
  const long tiles = (long)p.G * p.tiles_m * p.tiles_n;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wino_rows_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)wino_rows_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)wino_rows_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  if (p.dense == 2) hipLaunchKernelGGL(wino_rows_kernel<2>, dim3((unsigned)tiles), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  else if (p.dense) hipLaunchKernelGGL(wino_rows_kernel<1>, dim3((unsigned)tiles), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(wino_rows_kernel<0>, dim3((unsigned)tiles), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  MRN_LAUNCH_CHECK("conv2d_wino_rows");
  return MRN_OK;
}
