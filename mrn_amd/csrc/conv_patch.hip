// Patch-resident, weight-stationary 3x3 / stride 1 / pad 1 convolution for the NARROW early layers of the frozen experts' stacks
// (Cin = 32 -> Cout = 64: ResNet conv0_2, modules/feature_extraction.py:216-218; Cin = 64 -> Cout = 128: layer1[0].conv1 :171-199
// and conv 2 of the TPS localisation network, modules/transformation.py:63-66), G lock-step experts in one launch, products as
// split-fp16 x3 (lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate) like every other frozen convolution.
//
// Why another kernel.  On these shapes the tiled implicit-GEMM kernel (conv_x3.hip) is neither MFMA- nor HBM-bound but STAGING-bound:
// K is only 288 / 576, so a K-step is one (tap, 32 channels) slice and every 128-byte activation line travels L2 -> LDS nine times,
// once per tap, next to a weight slice of the same size (32 -> 64 at 32 x 256: 2.3 ms for 4.8 GB = 2.1 TB/s, 14.5 GB staged).  Here
//   * the WEIGHTS live in registers for the whole launch: a wave owns 32 output channels, and its A-operand fragments of all 9 taps
//     (144 VGPRs per 32 input channels) are loaded once; workgroups are persistent (gridDim.x per expert, tiles strided), so the
//     weight traffic of the launch is one pass over 74 / 295 KB per workgroup;
//   * the ACTIVATION patch of a tile -- (TH + 2) x 34 pixels, 128 bytes per (pixel, 32 channels) -- is staged ONCE by
//     buffer_load ... lds into a two-stage ring (the next tile's patch flies under the current tile's MFMAs), and the nine taps are
//     nine shifted ds_read_b128 views of it (column-keyed XOR swizzle: conflict-free for every shift);
//   * the epilogue can POOL: a 2 x 2 / stride 2 max-pool behind train-mode BatchNorm + ReLU commutes with them up to the sign of the
//     BatchNorm weight (y -> gamma (y - mean) / sigma + beta is monotone: increasing for gamma >= 0, decreasing for gamma < 0, and so
//     is every rounding step of its fp32 evaluation), and gamma is a PARAMETER, known before the batch statistics are: the kernel
//     writes, per channel, the maximum (gamma >= 0) or the minimum (gamma < 0) of each window of the raw convolution output -- a
//     quarter of the bytes -- while the partial sums for the statistics still cover every unpooled value.  The BatchNorm-apply pass
//     then runs on the pooled map and produces bit for bit what apply -> ReLU -> max-pool produced.
// One wave per SIMD (4 waves, up to 512 registers); LDS: 2 x 43 KiB (Cin 32, 8-row tiles) or 2 x 51 KiB (Cin 64, 4-row tiles) + 32 KiB
// of per-lane statistics accumulators (one deterministic row of partial sums per workgroup for the whole launch).
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct ConvPatchParams {
  const unsigned char* x;      // HL32 [Gx][B][H][W][Cin/32][128 B]
  const unsigned char* w;      // HL32 [G][Cout][Cin/32][9][128 B], power-of-two prescaled
  const float* w_scale;        // [G][2] = {s, 1/s}
  const float* bias;           // [G][Cout] or null
  const long long* gamma;      // [G] device addresses of the BatchNorm weights that follow (pooled form: which extreme to keep), or null
  float* y;                    // [G][B][H][W][Cout], pooled form [G][B][H/2][W/2][Cout]
  float* stats;                // [G][rows * RG][2][Cout] or null (RG = wave row groups: 2 for Cin 32, 1 for Cin 64)
  long x_gstride;              // bytes between the experts' inputs (0: shared)
  int x_bytes;                 // bytes of one expert's input
  int B, H, W, act, pool;
  int tiles_x, tiles_y, tiles; // per expert
  int rows;                    // statistics rows per expert: tile t belongs to row t % rows, whatever workgroup computes it
};

template <int CB, int NCO, int NST>
struct Cfg {
  static constexpr int RG = 4 / NCO;             // row groups of waves (the other factor of the four waves: 32-channel blocks)
  static constexpr int TH = 4 * RG, TW = 32;     // output tile
  static constexpr int PR = TH + 2, PC = TW + 2; // patch rows / columns
  static constexpr int NL = CB * PR * PC;        // 128-byte lines of a patch, [cb][row][column]
  static constexpr int NDMA = (NL + 7) / 8;      // 1-KiB DMA instructions per patch
  static constexpr int NJ = (NDMA + 3) / 4;      // per wave
  static constexpr int STAGE = NJ * 4 * 1024;    // (padded: every wave issues exactly NJ pieces, the surplus ones write zeros behind the patch)
  static constexpr bool LEAN = CB == 2;          // 288 weight registers per wave: statistics in LDS, bias / BatchNorm signs re-derived, no second accumulator set
  static constexpr int STATS = LEAN ? 4 * 8 * 64 * 16 : 0;      // [wave][chunk][lane][16 B]
  static constexpr int LDS = NST * STAGE + STATS;
  static constexpr int NS = CB * 9 * 2;          // (channel block, tap, 16-channel half) steps of a tile
};

__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
#ifdef MRN_PPROBE_NO_MFMA
  f32x16 r = c;                 // (what-if probes MRN_PPROBE_*: never in the product build; tools/build_probe.sh)
  r[0] += __builtin_bit_cast(float, a[0] ^ b[0]);
  return r;
#endif
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16v8*>(&a), *reinterpret_cast<const f16v8*>(&b), c, 0, 0, 0);
}

#ifdef MRN_PPROBE_TIMING
// timing probe (never in the product build): per-wave shader-clock totals of a tile's phases, summed over all waves
__device__ unsigned long long g_patch_dbg[8];
extern "C" __attribute__((visibility("default"))) int mrn_patch_dbg_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_patch_dbg), sizeof(g_patch_dbg)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_patch_dbg), z, sizeof(z));
  }
  return 0;
}
#define PTICK(var) const long var = __builtin_readcyclecounter()
#define PADD(slot, a, b) dbg_acc[slot] += (b) - (a)
#else
#define PTICK(var)
#define PADD(slot, a, b)
#endif

// value of the lane whose index differs in bit 0 (quad_perm [1,0,3,2])
__device__ __forceinline__ float swap1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// NST = 2 stages: one workgroup per CU, the next tile's patch flies under this tile's MFMAs.  (Two single-stage workgroups per CU at
// 256 registers were measured: the patch fetch then sits exposed -- 11 k cycles per tile -- and the register budget spills.)
// The main loop of a pass is ONE basic block -- no branch on anything the launch decides at run time: the pooled form is a template
// parameter, ReLU is a floor of 0 or -inf, invalid lanes store and fetch at an offset beyond their buffer descriptor (dropped / zeros)
// -- because the scheduler interleaves the epilogue slices with the MFMAs only inside a block.
// NPROD: 3 = split-fp16 x3 (parity mode); 1 = hi * hi only (reduced-precision mode: the lo halves are neither held nor read)
template <int CB, int NCO, int NST, bool POOL, int NPROD = 3>
__global__ __launch_bounds__(256) void conv_patch_x3_kernel(const ConvPatchParams p) {
  using C = Cfg<CB, NCO, NST>;
  constexpr int PR = C::PR, PC = C::PC, TH = C::TH, STAGE = C::STAGE;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = lane & 31, kh = lane >> 5;
  const int g = blockIdx.y;
  const int cob = wave % NCO, rg = wave / NCO;
  const int co0 = cob * 32;
  constexpr int Cout = NCO * 32;

  // ---- weights: the A-operand fragments of all taps, once.  Lane (m = n, kh): output channel co0 + m, input channels 16 ks + 8 kh .. + 7
  u32x4 Wh[CB][9][2], Wl[CB][9][2];
  {
    const unsigned char* wg = p.w + ((long)g * Cout + co0 + n) * (CB * 9 * 128) + kh * 16;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          Wh[cb][tap][ks] = *reinterpret_cast<const u32x4*>(wg + (cb * 9 + tap) * 128 + ks * 32);
          if constexpr (NPROD == 3) Wl[cb][tap][ks] = *reinterpret_cast<const u32x4*>(wg + (cb * 9 + tap) * 128 + 64 + ks * 32);
        }
  }
  const float osc = p.w_scale[g * 2 + 1];
  // which extreme of a pooling window survives BatchNorm + ReLU + max-pool: bit e of the mask = gamma of accumulator register e's channel < 0
  unsigned negmask = 0;
  if (POOL && p.gamma) {
    const float* gm = reinterpret_cast<const float*>(p.gamma[g]);
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (gm[co0 + (e & 3) + 8 * (e >> 2) + 4 * kh] < 0.f) negmask |= 1u << e;
  }

  // ---- fragment read offsets: lane (pixel column n + dx, kh); logical 16-byte chunk = 4 * plane + 2 * ks + kh, stored at chunk ^ key,
  // key = (patch column >> 1) & 7 -- with 34-column rows (4352 bytes = 17 x 256) every row starts on bank 0, so the 16 lanes a
  // ds_read_b128 services per cycle (patch columns distinct mod 16) hit 16 distinct 4-bank groups for every shift dx
  int offv[3][2][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int pc = n + dx, key = (pc >> 1) & 7;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) offv[dx][pl][ks] = (rg * 4 * PC + pc) * 128 + (((pl * 4 + ks * 2 + kh) ^ key) << 4);
  }

  // ---- DMA geometry: instruction j = wave + 4 i moves lines 8 j .. 8 j + 7; lane -> line L = 8 j + (lane >> 3), physical chunk
  // lane & 7 (LDS is written lane-linearly: the swizzle is applied to the SOURCE chunk).  Recomputed per piece (a dozen VALU
  // operations) rather than held in 2 x NJ registers next to the stationary weights.
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long)g * p.x_gstride), 0, p.x_bytes, 0x00020000);
  const int tiles_img = p.tiles_x * p.tiles_y;
  struct TileAt { int b, y0, x0, base; };        // sample, top-left output pixel, byte offset of the patch's top-left pixel (y0 - 1, x0 - 1)
  auto locate = [&](int tile) {
    TileAt a;
    a.b = tile / tiles_img;
    const int r2 = tile - a.b * tiles_img, ty = r2 / p.tiles_x;
    a.y0 = ty * TH;
    a.x0 = (r2 - ty * p.tiles_x) * C::TW;
    a.base = (((a.b * p.H + a.y0 - 1) * p.W + a.x0 - 1) * CB) * 128;
    return a;
  };
  auto issue_dma = [&](int i, const TileAt& a, int stage, bool live) {
    const int L = 8 * (wave + 4 * i) + (lane >> 3);                 // (lines >= NL: the padding of the stage, zeros)
    const int cb = CB == 1 ? 0 : (L >= PR * PC ? 1 : 0), rem = L - cb * (PR * PC);
    const int prow = (rem * 1928) >> 16, pcol = rem - prow * PC;   // rem / 34 for rem < 1024
    static_assert(PC == 34 && C::NJ * 32 < 1024, "the division by the patch width is a multiply-shift");
    const int y = a.y0 - 1 + prow, x = a.x0 - 1 + pcol;
    const bool ok = live && L < C::NL && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    const int rel = ((prow * p.W + pcol) * CB + cb) * 128 + (((lane & 7) ^ ((pcol >> 1) & 7)) << 4);
    const int okm = -(int)ok;                                       // (mask arithmetic, not a select: the compiler turned the select into an exec-masked BLOCK)
    const int voff = ((a.base + rel) & okm) | ((int)0x80000000 & ~okm);      // beyond the descriptor: zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(lds + stage * STAGE + (wave + 4 * i) * 1024), 16, voff, 0, 0, 0);
  };

  constexpr int RP = 2, NPASS = 4 / RP;          // a wave's four rows in passes of two: two accumulators (= one pooled row) per pass
  // ---- software pipeline.  A wave alone on its SIMD hides only what sits BETWEEN its MFMAs, and a pass's epilogue (scale + bias,
  // statistics, the pooling window, stores: ~230 VALU operations) measured 3.6 k cycles of idle matrix pipe per pass when it ran behind
  // the pass (in-kernel clocks: 7.2 k of a 16.4 k-cycle tile).  So the epilogue of pass k runs in SLICES under the MFMAs of pass k + 1:
  // pass 0 accumulates into accA, pass 1 into accB (static assignment), slice j of the pending epilogue sits in step j of the next
  // main loop, and the last pass's epilogue crosses the tile boundary (its coordinates travel in `pend`).
  f32x16 accA[RP], accB[RP];
#pragma unroll
  for (int r = 0; r < RP; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) accA[r][e] = accB[r][e] = 0.f;      // (the first pass's slices run on accB before anything was accumulated)
  f32x4 hold = {0.f, 0.f, 0.f, 0.f};              // pooled form: the first quad of a lane's pair, until the second is ready
  struct Pending { int off, ok0, ok1, okp, live; };  // element offset of the lane's first output; the pass's two rows / its pooling window exist; anything pending
  Pending pend = {0, 0, 0, 0, 0};
  constexpr bool LEAN = C::LEAN, PIPE = !LEAN;
  // the lane's running partial statistics (register quad q: channels co0 + 8 q + 4 kh .. + 3): registers, or (LEAN) LDS [wave][chunk][lane][16 B]
  f32x4 st_s[LEAN ? 1 : 4], st_q[LEAN ? 1 : 4];
  f32x4 bias4[LEAN ? 1 : 4];
  unsigned char* const st_base = lds + NST * STAGE + (wave * 8 * 64 + lane) * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (LEAN) {
      *reinterpret_cast<f32x4*>(st_base + q * 1024) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(st_base + (4 + q) * 1024) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      st_s[q] = st_q[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      bias4[q] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + (long)g * Cout + co0 + 8 * q + 4 * kh) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // sign of the BatchNorm weight of accumulator register e's channel: the minimum of a window is -max(-v), i.e. negate, max, negate
  // (multiplications by +-1 are exact, and their results are canonical: the maxima below then need no quieting v_max v, v, v each)
  float sgn[LEAN ? 1 : 16];
  if (!LEAN) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sgn[e] = (negmask >> e) & 1u ? -1.f : 1.f;
  }
  auto sign_of = [&](int e) { return LEAN ? ((negmask >> e) & 1u ? -1.f : 1.f) : sgn[LEAN ? 0 : e]; };
  const int Ho = p.H >> 1, Wo = p.W >> 1;
  const long y_elems = (long)p.B * (POOL ? Ho * (long)Wo : p.H * (long)p.W) * Cout;      // one expert's output (< 2^30 elements: checked by the launcher)
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long)g * y_elems), 0, (int)(y_elems * 4), 0x00020000);
  const float relu_floor = p.act == 1 ? 0.f : -INFINITY;
  auto store16 = [&](const f32x4 v, int elem_off, bool ok) {      // (a lane without an output stores beyond the descriptor: dropped, no branch)
    const int okm = -(int)ok;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, ((elem_off * 4) & okm) | ((int)0x80000000 & ~okm), 0, 0);
  };

  // slice j (0 .. 7) of the epilogue of the pass whose accumulators are `acc`: even j = 2 q: scale, bias, statistics of quad q;
  // odd j = 2 q + 1: its output (the pooling window's extreme or the two rows) and, for the pooled form, the store of a quad pair
  auto epilogue_slice = [&](f32x16 (&acc)[RP], int j) {
    const int q = j >> 1;
    if ((j & 1) == 0) {
      f32x4 bq;
      if (LEAN) bq = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + (long)g * Cout + co0 + 8 * q + 4 * kh) : f32x4{0.f, 0.f, 0.f, 0.f};
      else bq = bias4[LEAN ? 0 : q];
      f32x4 cs, cq;
      if (LEAN) cs = cq = f32x4{0.f, 0.f, 0.f, 0.f};
      else { cs = st_s[LEAN ? 0 : q]; cq = st_q[LEAN ? 0 : q]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v0 = fmaf(acc[0][4 * q + i], osc, bq[i]), v1 = fmaf(acc[1][4 * q + i], osc, bq[i]);
        acc[0][4 * q + i] = v0;
        acc[1][4 * q + i] = v1;
        const float m0 = pend.ok0 ? v0 : 0.f, m1 = pend.ok1 ? v1 : 0.f;
        cs[i] += m0 + m1;
        cq[i] = fmaf(m1, m1, fmaf(m0, m0, cq[i]));
      }
      if (LEAN) {
        if (p.stats) {
          f32x4* ps_ = reinterpret_cast<f32x4*>(st_base + q * 1024);
          f32x4* pq_ = reinterpret_cast<f32x4*>(st_base + (4 + q) * 1024);
          *ps_ = *ps_ + cs;
          *pq_ = *pq_ + cq;
        }
      } else { st_s[LEAN ? 0 : q] = cs; st_q[LEAN ? 0 : q] = cq; }
      return;
    }
#ifdef MRN_PPROBE_NO_STORE
    return;
#endif
    if (!POOL) {
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = relu_nan(acc[r][4 * q + i], relu_floor);
        store16(v, pend.off + r * p.W * Cout + 8 * q, r == 0 ? pend.ok0 : pend.ok1);
      }
    } else {
      // the 2 x 2 window: the pass's two rows, columns (n, n ^ 1).  Even lanes keep quads 0-1 of the window, odd lanes quads 2-3.
      f32x4 sel;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sg = sign_of(4 * q + i);
        const float f0 = acc[0][4 * q + i], f1 = acc[1][4 * q + i];      // (copies: __builtin_bit_cast applied to a vector-element lvalue read element 0)
        float mx = max_nan(f0 * sg, f1 * sg);
        mx = max_nan(mx, swap1(mx));
        sel[i] = relu_nan(mx * sg, relu_floor);
      }
      const bool odd = n & 1;
      if (q == 0 || q == 2) hold = sel;          // (even lanes have stored quads 0-1 by the time quad 2 overwrites it)
      else {
        const bool mine = pend.okp && odd == (q == 3);
        store16(hold, pend.off, mine);
        store16(sel, pend.off + 8, mine);
      }
    }
  };
  // what the epilogue of (tile position a, pass ps) needs to know, computed when the pass is issued
  auto make_pending = [&](const TileAt& a, int ps) {
    Pending e;
    const int y0 = a.y0 + rg * 4 + ps * RP, x = a.x0 + n;
    const bool xok = x < p.W;
    e.ok0 = xok && y0 < p.H;
    e.ok1 = xok && y0 + 1 < p.H;
    e.live = 1;
    e.okp = 0;
    if (!POOL) e.off = ((a.b * p.H + y0) * p.W + x) * Cout + co0 + 4 * kh;
    else {
      const int oy = y0 >> 1, ox = x >> 1;
      e.okp = oy < Ho && ox < Wo;
      e.off = ((a.b * Ho + oy) * Wo + ox) * Cout + co0 + 4 * kh + (n & 1) * 16;
    }
    return e;
  };

  // ---- tile order.  The partial BatchNorm statistics must not depend on how many workgroups the launch has (the experts run as one
  // lock-step group of six or as sub-groups of two or three: bit-identical results either way), so tiles are summed per ROW -- tile t
  // belongs to row t % rows, rows = min(tiles, 1008) whatever the grid -- and a workgroup walks whole rows: row = blockIdx.x,
  // blockIdx.x + gridDim.x, ...; inside a row tiles row, row + rows, ...  A row's sums leave (flush_stats) when its last epilogue has run.
  auto flush_stats = [&](int row) {
    f32x4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v[c] = LEAN ? *reinterpret_cast<const f32x4*>(st_base + c * 1024) : (c < 4 ? st_s[LEAN ? 0 : c] : st_q[LEAN ? 0 : c - 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float sum = v[c][i];
#pragma unroll
        for (int ofs = 16; ofs > 0; ofs >>= 1) sum += __shfl_xor(sum, ofs);      // over the 32 pixel columns of a kh half
        v[c][i] = sum;
      }
      if (LEAN) *reinterpret_cast<f32x4*>(st_base + c * 1024) = f32x4{0.f, 0.f, 0.f, 0.f};
      else if (c < 4) st_s[LEAN ? 0 : c] = f32x4{0.f, 0.f, 0.f, 0.f};
      else st_q[LEAN ? 0 : c - 4] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (n == 0 && p.stats) {                      // lanes 0 and 32: channels co0 + 8 q + 4 kh .. + 3 of the wave's row group
      float* dst = p.stats + (((long)g * p.rows + row) * C::RG + rg) * 2 * Cout + co0 + 4 * kh;
#pragma unroll
      for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(dst + (c >> 2) * Cout + 8 * (c & 3)) = v[c];
    }
  };
  int row = blockIdx.x, tile = row, stage = 0;
  int flush_row = -1;                             // row whose sums are complete once the pending epilogue has run
  TileAt cur = locate(tile < p.tiles ? tile : 0);
  if (tile < p.tiles) {
#pragma unroll
    for (int i = 0; i < C::NJ; ++i) issue_dma(i, cur, 0, true);
  }
  __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));                 // vmcnt(0): weights and the first patch
#ifdef MRN_PPROBE_TIMING
  long dbg_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  while (tile < p.tiles) {
    PTICK(tk0);
    // NST = 2: everyone's pieces of this tile's patch have landed (each wave waited for its own at the end of its last tile / above)
    // and everyone has left the other stage (tile - gridDim.x)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PTICK(tk1);
    PADD(0, tk0, tk1);                                              // loop-top barrier (+ the single-stage form's patch fetch)
    int nxt = tile + p.rows, nrow = row;
    if (nxt >= p.tiles) {                         // the row is done: on to the workgroup's next row
      nrow = row + gridDim.x;
      nxt = nrow < p.rows ? nrow : p.tiles;
    }
    const bool more = nxt < p.tiles;
    const TileAt nx = locate(more ? nxt : tile);

#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      f32x16 (&acc)[RP] = (!PIPE || ps == 0) ? accA : accB;      // this pass accumulates here ...
      f32x16 (&accp)[RP] = ps == 0 ? accB : accA;                // ... while (PIPE) the previous pass's results leave from there
#pragma unroll
      for (int r = 0; r < RP; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
      // ---- main loop: steps (cb, tap, ks); the fragment reads of step s + 1 are issued in front of the MFMAs of step s
      u32x4 Xh[2][RP], Xl[2][RP];
      auto read_X = [&](int s, int set) {
        const int cb = s / 18, tap = (s % 18) >> 1, ks = s & 1, dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int r = 0; r < RP; ++r) {
          const int imm = (cb * PR + ps * RP + r + dy) * PC * 128;
          if constexpr (NPROD == 3) Xl[set][r] = *reinterpret_cast<const u32x4*>(lds + offv[dx][1][ks] + imm);
          Xh[set][r] = *reinterpret_cast<const u32x4*>(lds + offv[dx][0][ks] + imm);
        }
      };
      PTICK(tm0);
      read_X(0, 0);
      constexpr int DMA_PER_STEP = (C::NJ + C::NS - 3) / (C::NS - 2);      // the next patch's pieces spread over the steps of pass 0
      constexpr int EPI0 = 1;                                              // first step that carries an epilogue slice
#pragma unroll
      for (int s = 0; s < C::NS; ++s) {
        const int cb = s / 18, tap = (s % 18) >> 1, ks = s & 1, set = s & 1;
        if (s + 1 < C::NS) read_X(s + 1, set ^ 1);
        if (ps == 0) {                             // (no further tile: the pieces fetch zeros -- no branch inside the block)
#pragma unroll
          for (int i = s * DMA_PER_STEP; i < (s + 1) * DMA_PER_STEP; ++i)
            if (i < C::NJ) issue_dma(i, nx, stage ^ 1, more);
        }
        if (PIPE && s >= EPI0 && s < EPI0 + 8) epilogue_slice(accp, s - EPI0);      // (nothing pending: every validity flag is 0)
        if constexpr (NPROD == 3) {
#pragma unroll
          for (int r = 0; r < RP; ++r) acc[r] = mma(Wh[cb][tap][ks], Xl[set][r], acc[r]);
#pragma unroll
          for (int r = 0; r < RP; ++r) acc[r] = mma(Wl[cb][tap][ks], Xh[set][r], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < RP; ++r) acc[r] = mma(Wh[cb][tap][ks], Xh[set][r], acc[r]);
        // one scheduling region per step (the scheduler otherwise hoists the reads of several steps: live fragments spill next to the
        // weights), ordered [MFMA, a few VALU operations of the pending epilogue, a fragment read] so that the fillers sit BETWEEN the MFMAs
#define PATCH_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define PATCH_INTERLEAVE PATCH_SGB(0x008, 1) PATCH_SGB(0x002, 6) PATCH_SGB(0x100, 1) PATCH_SGB(0x020, 1) PATCH_SGB(0x040, 1)
        PATCH_INTERLEAVE PATCH_INTERLEAVE
        if constexpr (NPROD == 3) { PATCH_INTERLEAVE PATCH_INTERLEAVE PATCH_INTERLEAVE PATCH_INTERLEAVE }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ps == 0 && flush_row >= 0) {            // (the pending epilogue of the previous row's last tile ran under this pass)
        flush_stats(flush_row);
        flush_row = -1;
      }
      pend = make_pending(cur, ps);
      if (!PIPE) {                                 // (LEAN: no second accumulator set -- the epilogue runs behind its pass)
#pragma unroll
        for (int j = 0; j < 8; ++j) epilogue_slice(acc, j);
        pend.live = 0;
      }
      PTICK(tm1);
      PADD(1, tm0, tm1);                                            // main loop of the pass
      // own pieces of the next patch have landed (they had the whole tile to do so; so have the stores of the slices above)
      if (ps == NPASS - 1) __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));      // vmcnt(0)
      PTICK(tm2);
      PADD(2, tm1, tm2);                                            // wait for the next patch
    }
    PTICK(tk9);
    PADD(3, tk1, tk9);                                              // whole tile behind the barrier
    PADD(4, tk0, tk9);
#ifdef MRN_PPROBE_TIMING
    dbg_acc[5] += 1;
#endif
    if (nrow != row) flush_row = row;
    cur = nx;
    tile = nxt;
    row = nrow;
    stage ^= 1;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offv[dx][pl][ks] += stage ? STAGE : -STAGE;
  }
  // the last pass's epilogue (pass NPASS - 1 accumulated into accB)
  if (PIPE && pend.live) {
#pragma unroll
    for (int j = 0; j < 8; ++j) epilogue_slice(accB, j);
  }

#ifdef MRN_PPROBE_TIMING
  if (lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_patch_dbg[i], (unsigned long long)dbg_acc[i]);
#endif
  if (flush_row >= 0) flush_stats(flush_row);
}

int patch_wgs(int G, long tiles, int per_cu) {
  // persistent workgroups, per_cu per CU: split the slots evenly over the experts
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    else
      cus = 256;
  }
  long per = (long)cus * per_cu / G;       // (never more workgroups than slots: a workgroup that has to wait for one doubles the launch)
  if (per > tiles) per = tiles;
  return (int)(per < 1 ? 1 : per);
}

}  // namespace

// does the patch-resident kernel take this layer?  (3 x 3, stride 1, padding 1 is implied by the entry point)
MRN_EXPORT int64_t mrn_conv3x3_patch_supported(int Cin, int Cout) { return (Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 128); }

static int patch_rows(long tiles) { return (int)(tiles < 1008 ? tiles : 1008); }      // (1008 = 42 x 24: no imbalance for six experts on 256 CUs, <= 1.6 % for 1 / 2 / 3)

// rows of partial statistics per expert that mrn_conv3x3_patch_x3_hl32 writes: a function of the layer only, NOT of G or the device --
// tile t is summed into row t % rows whatever workgroup computes it, so every grouping of the experts yields the same bits
MRN_EXPORT int64_t mrn_conv3x3_patch_stats_blocks(int G, int B, int H, int W, int Cin) {
  (void)G;
  const int TH = Cin == 32 ? 8 : 4, RG = Cin == 32 ? 2 : 1;
  return (int64_t)patch_rows((long)B * ((H + TH - 1) / TH) * ((W + 31) / 32)) * RG;
}

// y[g] = act(conv3x3(x[g or shared], w[g]) / w_scale[g] + bias[g]) for G experts, stride 1, padding 1, (Cin, Cout) = (32, 64) or (64, 128).
// x_hl: HL32 lines [Gx][B][H][W][Cin/32][128 B] (x_group_stride_bytes 0: one shared input); w_hl / w_scale from mrn_pack_weight_hl32 /
// mrn_pow2_scale_f32; stats [G][mrn_conv3x3_patch_stats_blocks][2][Cout] (sums / sums of squares of the pre-activation result) or NULL.
// pool = 1 (H, W even): y is [G][B][H/2][W/2][Cout] and holds, per 2 x 2 window and channel, the window's MAXIMUM where the
// BatchNorm weight that follows (bn_gamma_ptrs: device table of G device addresses, NULL = no BatchNorm: all maxima) is >= 0 and
// its MINIMUM where it is negative -- applying scale / shift / ReLU to that map equals max-pooling the applied full map bit for bit;
// the statistics still cover the full map.
static int patch_launch(const void* x_hl, const void* w_hl, const float* w_scale, const float* bias,
                        const void* bn_gamma_ptrs, float* y, float* stats, int G, int64_t x_group_stride_bytes, int B,
                        int H, int W, int Cin, int Cout, int act, int pool, int products, void* stream) {
  MRN_CHECK_ARG(x_hl && w_hl && w_scale && y && G >= 1 && mrn_conv3x3_patch_supported(Cin, Cout),
                "mrn_conv3x3_patch_x3_hl32: bad operands (Cin=%d Cout=%d)", Cin, Cout);
  MRN_CHECK_ARG(!pool || (H % 2 == 0 && W % 2 == 0), "mrn_conv3x3_patch_x3_hl32: the pooled form needs even H, W (%d x %d)", H, W);
  MRN_CHECK_ARG((long)B * H * W * Cin * 4 < 2147483647L && (long)B * H * W * Cout * 4 < 2147483647L && (uintptr_t)x_hl % 16 == 0 &&
                    (uintptr_t)w_hl % 16 == 0 && (uintptr_t)y % 16 == 0,
                "mrn_conv3x3_patch_x3_hl32: one expert's input and output must stay below 2 GiB each and be 16-byte aligned");
  if (B == 0 || H == 0 || W == 0) return MRN_OK;
  ConvPatchParams p;
  p.x = (const unsigned char*)x_hl; p.w = (const unsigned char*)w_hl; p.w_scale = w_scale; p.bias = bias;
  p.gamma = (const long long*)bn_gamma_ptrs; p.y = y; p.stats = stats; p.x_gstride = x_group_stride_bytes;
  p.x_bytes = (int)((long)B * H * W * Cin * 4);
  p.B = B; p.H = H; p.W = W; p.act = act; p.pool = pool;
  const int TH = Cin == 32 ? 8 : 4;
  p.tiles_x = (W + 31) / 32; p.tiles_y = (H + TH - 1) / TH; p.tiles = B * p.tiles_x * p.tiles_y;
  p.rows = patch_rows(p.tiles);
  int wgs = patch_wgs(G, p.tiles, 1);
  if (wgs > p.rows) wgs = p.rows;
#define PATCH_LAUNCH(CB_, NCO_, POOL_, NP_)                                                                                              \
  do {                                                                                                                                   \
    using C = Cfg<CB_, NCO_, 2>;                                                                                                          \
    static bool set = false;                                                                                                             \
    if (!set) {                                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv_patch_x3_kernel<CB_, NCO_, 2, POOL_, NP_>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); \
      set = true;                                                                                                                        \
    }                                                                                                                                    \
    hipLaunchKernelGGL((conv_patch_x3_kernel<CB_, NCO_, 2, POOL_, NP_>), dim3(wgs, G), dim3(256), C::LDS, (hipStream_t)stream, p);      \
  } while (0)
#define PATCH_LAUNCH_P(CB_, NCO_, POOL_) do { if (products == 1) PATCH_LAUNCH(CB_, NCO_, POOL_, 1); else PATCH_LAUNCH(CB_, NCO_, POOL_, 3); } while (0)
  if (Cin == 32) {
    if (pool) PATCH_LAUNCH_P(1, 2, true);
    else PATCH_LAUNCH_P(1, 2, false);
  } else {
    if (pool) PATCH_LAUNCH_P(2, 4, true);
    else PATCH_LAUNCH_P(2, 4, false);
  }
#undef PATCH_LAUNCH_P
#undef PATCH_LAUNCH
  MRN_LAUNCH_CHECK("conv3x3_patch_x3");
  return MRN_OK;
}

MRN_EXPORT int mrn_conv3x3_patch_x3_hl32(const void* x_hl, const void* w_hl, const float* w_scale, const float* bias,
                                         const void* bn_gamma_ptrs, float* y, float* stats, int G, int64_t x_group_stride_bytes, int B,
                                         int H, int W, int Cin, int Cout, int act, int pool, void* stream) {
  return patch_launch(x_hl, w_hl, w_scale, bias, bn_gamma_ptrs, y, stats, G, x_group_stride_bytes, B, H, W, Cin, Cout, act, pool, 3, stream);
}

// the same convolution with ONE fp16 product per term (hi * hi; the lo halves of both operands stay unread): the reduced-precision mode
MRN_EXPORT int mrn_conv3x3_patch_x1_hl32(const void* x_hl, const void* w_hl, const float* w_scale, const float* bias,
                                         const void* bn_gamma_ptrs, float* y, float* stats, int G, int64_t x_group_stride_bytes, int B,
                                         int H, int W, int Cin, int Cout, int act, int pool, void* stream) {
  return patch_launch(x_hl, w_hl, w_scale, bias, bn_gamma_ptrs, y, stats, G, x_group_stride_bytes, B, H, W, Cin, Cout, act, pool, 1, stream);
}
