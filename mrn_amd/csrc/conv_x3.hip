// Grouped split-fp16 x3 implicit-GEMM convolution, 256-wide tiles, both operands staged by direct-to-LDS DMA.
//
// Arithmetic: every fp32 operand is x = hi + lo (two fp16, 22 significand bits); product = lo*hi + hi*lo + hi*hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation (same numerics as conv_bf16_dma_kernel<3,true> in gemm_bf16.hip).
//
// Operand layout "HL32" (written by the producer pass, mrn_split_hl32_f32 / mrn_pack_weight_hl32):
//   activation [pixel][Cin/32][ hi[32] | lo[32] ] fp16  -- one 128-byte line per (pixel, 32-channel block)
//   weight     [Cout][Cin/32][tap][ hi[32] | lo[32] ] fp16
// so that one K-step of the GEMM (32 channels of one tap) is exactly one 128-byte line per tile row for BOTH
// operands: a DMA lane-group of 8 lanes fetches a full cache line, and the reduction runs channel-block-outer /
// tap-inner: the kh*kw taps of one channel block re-read the same ~BM pixel lines back to back (L2 hits) instead
// of sweeping the whole activation once per tap (the earlier tap-outer kernel fetched 8x the activation from beyond L2;
// this one 2x its algorithmic bytes, the excess being the weight matrix re-streamed per round of tiles: profiles/r01_pmc.json).
//
// Tile BM x BN x 32; 256x256 runs 16 waves of 64x64 (four per SIMD, <= 128 VGPRs), the smaller tiles 8 waves and two
// workgroups per CU.  LDS stage = (BM + BN) rows x 128 B, two stages; one barrier per K-step:
//   wait own DMAs (tile kt) -> barrier -> issue DMAs (tile kt+1 -> other stage) -> 2 x (WM*WN*3) MFMAs on tile kt.
// Tiles are visited image-row-major and class-ordered (see the kernel prologue): border image rows skip the kernel rows
// that lie in the padding, and every XCD gets an equal share of both classes.
// LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 so that the 16-lane groups of
// ds_read_b128 (16 rows, same logical chunk) touch 16 distinct 16-byte slots of the 256-byte bank row.  The DMA
// writes LDS lane-linearly, so the swizzle is applied to the SOURCE chunk each lane fetches.
//
// Groups: G independent convolutions of identical geometry (the frozen experts of MRN's router phase run their
// backbones in lock-step): weights / bias / scales / outputs / statistics are [G]-strided, the input is either
// [G]-strided or shared (x_gstride = 0).  Tiles never straddle a group.
#include "common.hpp"
#include "conv_wino.hpp"
#include <stdlib.h>

namespace {

typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// K-window of one group of a windowed GEMM (mrn_gemm_x3_windows_hl32): byte offsets of the window's first line inside the two
// operand matrices and its length in K-steps (128-byte lines); rows are `a_pitch` / `w_pitch` lines apart
struct X3Window {
  long a_off, w_off;
  int nk, pad;
};

struct ConvX3Params {
  const unsigned char* x;     // HL32 activation
  const unsigned char* w;     // HL32 weight
  const unsigned char* zero;  // >= 128 bytes of zeros
  const float* bias;          // [G][N] or null
  const float* out_scale;     // [G][os_stride] = {s, 1/s} (epilogue multiplies by [1]) or null; os_stride 2, or 0: one pair for all groups
  int os_stride;
  unsigned* amax_ws;          // optional: max|y| of the stored result folded into 64 words (slot = block % 64), see mrn_pow2_finalize_f32
  const float* x_scale;       // [2] = {s, 1/s} of the activation operand (epilogue multiplies by [1]) or null
  const float* res;           // optional residual, same layout as y, added before the activation
  float* y;                   // [G][M][N], or null when only y_hl is wanted
  unsigned char* y_hl;        // the same result as HL32 lines [G][M][N/32][128 B] (dense rows, N % 32 == 0), or null
  float* stats;               // [G][tilesM][2][N] or null
  const float* ch_scale;      // [G][N] per-channel affine applied after bias (eval-mode BatchNorm folded into the epilogue) or null
  const float* ch_shift;      // [G][N] (with ch_scale)
  const unsigned char* res_hl; // residual as HL32 lines (same geometry as y_hl), added before the activation, or null
  long x_gstride, w_gstride;  // bytes
  int x_group_div;            // activation group = g / x_group_div (weight-gradient GEMMs: one dy^T chunk serves all taps)
  long y_gstride, y_ld;       // output group stride / row pitch in floats
  int x_bytes;                // bytes of one group's activation
  int G, M, N, Cb, taps, nk;  // nk = Cb * taps = K-steps of the full reduction (weight row length in 128-byte lines)
  int H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, act;
  int tilesM, tilesN;
  int oy_major, BWo;           // image-row-major GEMM row order: m = (oy * B + b) * Wo + ox
  int tiles_per_row;           // > 0: B * Wo is a multiple of the tile height and tiles are visited row-of-image fastest
  int class_order;             // interior-row tiles before border-row tiles inside every XCD's share
  unsigned wo_magic, bw_magic; // fast_div constants for Wo and B * Wo
  int wo_shift, bw_shift;
  unsigned kw_magic;          // ... and for kw (tap -> (ky, kx) once per K-step, on the scalar unit)
  int kw_shift;
  const X3Window* win;        // windowed 1x1 GEMM: per-group operand windows (null: the regular grouped conv)
  int a_pitch, w_pitch;       // ... row pitch of the two operand matrices in lines
  int w_bytes;                // ... bytes of the whole weight-side matrix (x_bytes = the activation-side one)
  int wino_W;                 // Winograd F(R,3) along W (WINO = R instantiations): real output width; GEMM rows are groups of R pixels
};

// 16 bytes per lane, global (buffer descriptor + per-lane byte offset + wave-uniform offset) -> LDS (wave-uniform base + 16*lane)
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t r, unsigned char* l, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const unsigned char* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
#ifdef MRN_PROBE_NO_MFMA
  f32x16 r = c;                 // (what-if probe, never in the product build: the staging floor without matrix work)
  r[0] += __builtin_bit_cast(float, a[0] ^ b[0]);
  return r;
#endif
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16v8*>(&a), *reinterpret_cast<const f16v8*>(&b), c, 0, 0, 0);
}

// n / d for 0 <= n < 2^31 with host-computed (magic, shift): q = umulhi(n, magic) >> shift
__device__ __forceinline__ int fast_div(int n, unsigned magic, int shift) {
  return magic ? (int)(__umulhi((unsigned)n, magic) >> shift) : n;      // magic 0 encodes d == 1
}

// GEMM row m -> output pixel index (b*Ho + oy)*Wo + ox.  Row-major order: identity.  Image-row-major order
// (p.oy_major): m = (oy * B + b) * Wo + ox, so a tile holds pixels of (nearly) one output row of the images and a tap
// whose input row falls into the padding for that output row can be skipped for the whole tile.
__device__ __forceinline__ int row_to_pixel(const ConvX3Params& p, int m, int& oy, int& ox, int& b) {
  if (p.oy_major) {
    oy = fast_div(m, p.bw_magic, p.bw_shift);
    const int r = m - oy * p.BWo;
    b = fast_div(r, p.wo_magic, p.wo_shift);
    ox = r - b * p.Wo;
  } else {
    const int hw = p.Ho * p.Wo;
    b = m / hw;
    const int rem = m - b * hw;
    oy = fast_div(rem, p.wo_magic, p.wo_shift);
    ox = rem - oy * p.Wo;
  }
  return (b * p.Ho + oy) * p.Wo + ox;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int NPROD>
constexpr bool x3_hi_only() {
  return NPROD == 1 && (WAVES_M * WM * 32) % (16 * WAVES_M * WAVES_N) == 0 && (WAVES_N * WN * 32) % (16 * WAVES_M * WAVES_N) == 0;
}
// LDS ring depth of the Winograd instantiations (128 x 128 tiles: 32 KiB per stage).  Measured on the dominant shape, same box
// (tools/bench_wino.py, what-if builds of tools/build_probe.sh): 2 stages 3.49 ms, 3 stages 2.89, 4 stages 2.99, 5 stages 3.07
#if defined(MRN_WINO_STAGES_4)
constexpr int X3_WINO_STAGES = 4;
#elif defined(MRN_WINO_STAGES_5)
constexpr int X3_WINO_STAGES = 5;
#else
constexpr int X3_WINO_STAGES = 3;
#endif
template <int WAVES_M, int WAVES_N, int WM, int WN, int NPROD, int WINO = 0>
constexpr size_t x3_lds_bytes() {
  return (size_t)(WAVES_M * WM * 32 + WAVES_N * WN * 32) *
         (x3_hi_only<WAVES_M, WAVES_N, WM, WN, NPROD>() ? 64 * 4 : 128 * (WINO > 0 ? X3_WINO_STAGES : 2));
}

// WAVES_M x WAVES_N waves; wave tile = (WM*32) x (WN*32); HL_OUT: the epilogue can also write the HL32 result (p.y_hl)
// NPROD: 3 = split-fp16 x3 (lo*hi + hi*lo + hi*hi, 22-bit products: the parity mode); 1 = hi*hi only (plain fp16 products with fp32
// accumulation -- the reduced-precision mode of BASELINE configs 2 and 5; the lo halves of the staged lines are not read)
// Winograd F(R,3) output transforms A^T [R][R+2] with the power-of-two row scales of the packed weight transform folded in
// (pack_weight_wino_hl32_kernel multiplies row m of G by wino_gscale(R, m); column m here carries the inverse)
template <int R> struct WinoAT;
template <> struct WinoAT<2> {
  static constexpr float at[2][4] = {{1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, -1.f}};
};
template <> struct WinoAT<4> {
  static constexpr float at[4][6] = {{0.25f, 0.5f, 0.5f, 0.5f, 0.5f, 0.f},
                                     {0.f, 0.5f, -0.5f, 1.f, -1.f, 0.f},
                                     {0.f, 0.5f, 0.5f, 2.f, 2.f, 0.f},
                                     {0.f, 0.5f, -0.5f, 4.f, -4.f, 1.f}};
};

// WINO = R > 0: 1-D Winograd F(R,3) along W for 3x3 / stride 1 / pad 1 convolutions.  The producer pass has already applied
// the input transform B^T to every group of R output columns (mrn_bn_apply_wino_grouped_f32: NC = R + 2 components per group,
// layout [b][y][group][component][Cin/32][128 B]) and the packed weights hold G g per kernel row ([Cout][component][Cin/32][ky]
// lines), so to this kernel the layer IS a 3x1 convolution with NC * Cin input channels on a [H][ceil(W/R)] map -- the staging,
// the tile order and the border-row skipping are unchanged -- except that the reduction is cut into NC component segments:
// each segment's sum T_m is folded into the R output accumulators Y_r += A^T[r][m] * T_m in registers, and the epilogue writes
// R pixels per GEMM row.  F(4,3): 6 products per 4 outputs instead of 12 -- half the MFMA work (and MFMA energy) of the direct
// form; measured error against float64 on post-ReLU data 1.7x the direct x3 product's (8.9e-7 vs 5.4e-7 rms, tools/wino_numerics.py).
template <int WAVES_M, int WAVES_N, int WM, int WN, bool HL_OUT = false, int NPROD = 3, int WINO = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_x3_kernel(const ConvX3Params p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
  // HI_ONLY (one product, tiles whose rows divide evenly over the waves): only the hi half of every line is staged -- 64-byte LDS
  // rows, 16 rows per DMA instruction, half the staging traffic and LDS footprint, which buys a four-deep ring instead of two stages
  constexpr bool HI_ONLY = x3_hi_only<WAVES_M, WAVES_N, WM, WN, NPROD>();
  constexpr int ROWB = HI_ONLY ? 64 : 128;                // LDS bytes per tile row
  constexpr int RPI = HI_ONLY ? 16 : 8;                   // tile rows per DMA instruction (64 lanes x 16 bytes)
  constexpr int NA = BM / (RPI * NW), NB = BN / (RPI * NW);   // DMA instructions per wave per K-step
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int NSTAGE = HI_ONLY ? 4 : 2;            // (the Winograd instantiations run their own ring: X3_WINO_STAGES)
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  int tile_n, tile_m, g;
  if (p.tiles_per_row > 0 && p.class_order) {
    // Image-row-major order with whole tiles per image row, two tile classes: INTERIOR rows run the full reduction,
    // BORDER rows (oy = 0, Ho-1) skip the kernel rows that fall into the padding (2/3 of the K-steps for 3x3, pad 1).
    // Every XCD's contiguous share [a, b) of the tile list gets its proportional part of both classes, interior tiles
    // first: the XCDs finish together (equal work), and the workgroups of an XCD walk the weight matrix in lock-step
    // (same class => same K-step sequence), which keeps the weight stream an L2 hit instead of three phases thrashing it.
    const int nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
    const int a = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int b = a + (xcd < r8 ? q8 + 1 : q8);
    const int n_int = p.Ho - 2;                                   // interior rows per image (Ho >= 3 guaranteed by the host)
    const long T = nwg, nl = (long)p.G * p.tiles_per_row * n_int * p.tilesN;
    const int li0 = (int)((long)a * nl / T), li1 = (int)((long)b * nl / T);
    const int u = lid - a;
    const bool interior = u < li1 - li0;
    const int idx = interior ? li0 + u : (a - li0) + (u - (li1 - li0));
    const int rows_in_class = interior ? n_int : 2;
    int row_sel;
    tile_n = idx % p.tilesN;
    int t2 = idx / p.tilesN;
    row_sel = t2 % rows_in_class;
    t2 /= rows_in_class;
    const int j = t2 % p.tiles_per_row;
    g = t2 / p.tiles_per_row;
    const int oy = interior ? 1 + row_sel : (row_sel == 0 ? 0 : p.Ho - 1);
    tile_m = oy * p.tiles_per_row + j;
  } else {
    tile_n = lid % p.tilesN;
    const int t2 = lid / p.tilesN;
    tile_m = t2 % p.tilesM;
    g = t2 / p.tilesM;
    // whole tiles per image row: consecutive workgroups take the SAME image block at successive output rows
    if (p.tiles_per_row > 0) tile_m = (tile_m % p.Ho) * p.tiles_per_row + tile_m / p.Ho;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- taps this tile needs (wave-uniform): a tap (ky, kx) is dropped when its input row iy = oy*sh - ph + ky lies in
  // the padding for EVERY output row oy the tile touches (only the image-row-major order makes that happen)
  unsigned active = p.taps >= 32 ? 0xffffffffu : (1u << p.taps) - 1u;
  if (p.oy_major) {
    const int oy_lo = m0 / p.BWo, oy_hi = (min(m0 + BM, p.M) - 1) / p.BWo;
    active = 0;
    for (int tp = 0; tp < p.taps; ++tp) {
      const int ky = tp / p.kw;
      // smallest / largest input row over the tile's output rows
      const int lo = oy_lo * p.sh - p.ph + ky, hi = oy_hi * p.sh - p.ph + ky;
      if (hi >= 0 && lo < p.H) active |= 1u << tp;
    }
  }
  active = (unsigned)__builtin_amdgcn_readfirstlane((int)active);   // wave-uniform by construction: keeps the K-step iterator
  const int nact = __builtin_popcount(active);                      // (tap, channel block) and the DMA's scalar offset in SGPRs
  // operand bases / pitches / reduction length: the group's own tensors, or (windowed GEMM) a K-window of two shared matrices
  int Cb = p.Cb, a_pitch = p.Cb, w_pitch = p.nk, a_bytes = p.x_bytes, w_bytes = (int)p.w_gstride;
  long a_base = (long)(g / p.x_group_div) * p.x_gstride, w_base = (long)g * p.w_gstride;
  if (p.win) {
    const X3Window e = p.win[g];
    Cb = e.nk; a_base = e.a_off; w_base = e.w_off;
    a_pitch = p.a_pitch; w_pitch = p.w_pitch;
    a_bytes = p.x_bytes - (int)e.a_off; w_bytes = p.w_bytes - (int)e.w_off;
  }
  const int nk = Cb * nact;

  // ---- DMA geometry: instruction j = i*8 + wave covers tile rows 8j .. 8j+7; lane -> row 8j + lane/8, LDS chunk lane&7.
  // Both operands go through buffer descriptors: per-lane 32-bit byte offset (VGPR) + wave-uniform K-step offset; an
  // offset beyond the descriptor's range returns zeros, which is how padded taps / rows beyond M are produced.
  const int lrow = HI_ONLY ? lane >> 2 : lane >> 3, lch = HI_ONLY ? lane & 3 : lane & 7;
  int arow[NA];                     // byte offset of (pixel0 line + swizzled chunk) inside this group's activation (may be < 0)
  unsigned amask[NA];               // bit tap = this row's tap is inside the image
  int brow[NB];                     // byte offset of (weight row + swizzled chunk); rows beyond N re-read row N-1 (never stored)
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + a_base, a_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w + w_base, w_bytes);
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (i * NW + wave) * RPI + lrow;
    const int coff = HI_ONLY ? (lch ^ ((row >> 2) & 3)) << 4 : (lch ^ ((row >> 1) & 7)) << 4;
    const int m = m0 + row;
    const bool ok = m < p.M;
    int oy, ox, b;
    row_to_pixel(p, ok ? m : 0, oy, ox, b);
    const int iy0 = oy * p.sh - p.ph, ix0 = ox * p.sw - p.pw;
    unsigned mask = 0;
    if (ok) {
      for (int tp = 0; tp < p.taps; ++tp) {
        const int ky = tp / p.kw, kx = tp - ky * p.kw;
        if ((unsigned)(iy0 + ky) < (unsigned)p.H && (unsigned)(ix0 + kx) < (unsigned)p.W) mask |= 1u << tp;
      }
    }
    amask[i] = mask;
    arow[i] = ((b * p.H + iy0) * p.W + ix0) * a_pitch * 128 + coff;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (i * NW + wave) * RPI + lrow;
    const int coff = HI_ONLY ? (lch ^ ((row >> 2) & 3)) << 4 : (lch ^ ((row >> 1) & 7)) << 4;
    const int n = min(n0 + row, p.N - 1);
    brow[i] = n * w_pitch * 128 + coff;
  }

  // K-steps run channel-block outer, active-tap inner; (cb, tap) of the next tile to fetch is a wave-uniform iterator
  int it_cb = 0;
  unsigned it_rem = active;
  int it_tap = 0;
  auto issue_next = [&](unsigned char* st) {
    if (it_rem == 0 && it_cb + 1 < Cb) {       // next channel block (after the last K-step the final tile is re-fetched
      it_rem = active;                            //  into the idle stage, which keeps the loop body one basic block)
      ++it_cb;
    }
    if (it_rem != 0) {
      it_tap = __builtin_ctz(it_rem);
      it_rem &= it_rem - 1;
    }
    const int tap = it_tap, cb = it_cb;
    const int ky = fast_div(tap, p.kw_magic, p.kw_shift), kx = tap - ky * p.kw;
    const int aoff = ((ky * p.W + kx) * a_pitch + cb) * 128;
#ifdef MRN_PROBE_NO_DMA_A
    if (it_cb == 0 && tap == __builtin_ctz(active))      // (what-if probe, never in the product build: activations staged once)
#endif
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int voff = ((amask[i] >> tap) & 1u) ? arow[i] + aoff : (int)0x80000000;
      dma16(xr, st + (i * NW + wave) * 1024, voff, 0);
    }
    const int boff = (cb * p.taps + tap) * 128;
#ifdef MRN_PROBE_NO_DMA_B
    if (it_cb == 0 && tap == __builtin_ctz(active))      // (what-if probe, never in the product build: weights staged once)
#endif
#pragma unroll
    for (int i = 0; i < NB; ++i) dma16(wr, st + BM * ROWB + (i * NW + wave) * 1024, brow[i], boff);
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets: row = lane & 31 (+32 per fragment), logical chunk = plane*4 + ks*2 + (lane >> 5);
  // tile rows start at multiples of 32 so the swizzle key depends on the lane only
  // (HI_ONLY: 64-byte rows, four chunks, key = (row >> 2) & 3 -- the 16 lanes of one ds_read_b128 phase cover all 64 banks once)
  const int key = HI_ONLY ? (lane >> 2) & 3 : (lane >> 1) & 7;
  const int rbase = (lane & 31) * ROWB;
  int foff[2][2];   // [plane][ks]
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[pl][ks] = rbase + ((((HI_ONLY ? 0 : pl * 4) + ks * 2 + (lane >> 5)) ^ key) << 4);
  constexpr int FRAG = 32 * ROWB;   // bytes of one 32-row fragment
  const int abase = wm * WM * FRAG, bbase = BM * ROWB + wn * WN * FRAG;

  u32x4 ah[2][WM], al[2][WM], bh[2][WN], bl[2][WN];
  auto read_frags = [&](const unsigned char* cur, int ks) {
#ifdef MRN_PROBE_NO_DSREAD
    if (it_cb > 0 || it_rem != active) return;          // (what-if probe, never in the product build: fragments read once)
#endif
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      if constexpr (NPROD == 3) al[ks][i] = *reinterpret_cast<const u32x4*>(cur + abase + i * FRAG + foff[1][ks]);
      ah[ks][i] = *reinterpret_cast<const u32x4*>(cur + abase + i * FRAG + foff[0][ks]);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      bh[ks][j] = *reinterpret_cast<const u32x4*>(cur + bbase + j * FRAG + foff[0][ks]);
      if constexpr (NPROD == 3) bl[ks][j] = *reinterpret_cast<const u32x4*>(cur + bbase + j * FRAG + foff[1][ks]);
    }
  };
  auto mmas = [&](int ks) {
    // consecutive MFMAs target different accumulators
    if constexpr (NPROD == 3) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = mma(al[ks][i], bh[ks][j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = mma(ah[ks][i], bl[ks][j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = mma(ah[ks][i], bh[ks][j], acc[i][j]);
  };

  // ---- main loop: every wave loads and computes every K-step; one barrier per K-step.  (Measured alternatives: a
  // ping-pong schedule -- half the waves compute while the other half fetch fragments, 4 barriers per K-step -- 371 vs
  // 410 TFLOP/s; a 3-stage ring with counted vmcnt + raw s_barrier on the 256x128 tile +2.7 %, which does not fit the
  // 160 KiB LDS at 256x256; single-stage 256x128 tiles with 4 waves and two workgroups per CU (occupancy instead of
  // software pipelining) 451 vs 478 TFLOP/s for the 256x256 double-buffered tile.)
  constexpr int WR = WINO > 0 ? WINO : 1;           // output pixels per GEMM row
  f32x16 Y[WR][WM][WN];                              // Winograd: the R output accumulators (acc holds the component sum T_m)
  if constexpr (WINO > 0) {
#pragma unroll
    for (int r = 0; r < WR; ++r)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) Y[r][i][j][e] = 0.f;
  }
  if constexpr (WINO > 0) {
    static_assert(NW == 8 && NPROD == 3 && !HL_OUT, "Winograd instantiations: 8 waves, x3 products, fp32 result");
    constexpr int NC = WINO + 2;
    // the skewed 8-wave schedule of the plain kernel; a component's last MFMAs (second half of its final K-step) retire right
    // after the barrier of the next component's first tile, then T is folded into the outputs and cleared
    const int KC = (Cb / NC) * nact;                 // K-steps per component (Cb = NC * Cin/32 here)
    auto combine = [&](int m) {
      float cf[WR];
#pragma unroll
      for (int r = 0; r < WR; ++r) cf[r] = WinoAT<WR>::at[r][m];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
#pragma unroll
          for (int r = 0; r < WR; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[r][i][j][e] = fmaf(cf[r], acc[i][j][e], Y[r][i][j][e]);
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
    };
    // With half the MFMAs per staged byte of the direct form a K-step is too short to cover a DMA round trip (the weight stream
    // of 6 x Cin lines per output channel comes from beyond L2), so the tiles run through a ring of X3_WINO_STAGES stages: NST - 2 K-steps
    // of DMA stay in flight across the barrier, own arrivals counted with s_waitcnt vmcnt (compile-time count: the iterator re-fetches
    // the final tile once the reduction is exhausted, so every iteration issues the same DMAs).
    constexpr int NST = X3_WINO_STAGES;
    constexpr int INFLIGHT = (NA + NB) * (NST - 2);
    constexpr int WAITIMM = (INFLIGHT & 15) | (7 << 4) | (15 << 8) | ((INFLIGHT >> 4) << 14);   // vmcnt(INFLIGHT), others untouched
    static_assert(INFLIGHT < 64, "vmcnt is six bits");
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_) issue_next(lds + s_ * STAGE);
    __builtin_amdgcn_s_waitcnt(WAITIMM);
    __builtin_amdgcn_s_barrier();
    read_frags(lds, 0);
    issue_next(lds + (NST - 1) * STAGE);
    read_frags(lds, 1);
    mmas(0);
    int kc = 1, comp = 0;                            // K-steps of the current component whose first half is issued
    int st_cur = 1, st_free = 0;                     // ring positions of tile kt and of the stage tile kt-1 has left
    for (int kt = 1; kt < nk; ++kt) {
      __builtin_amdgcn_s_waitcnt(WAITIMM);           // own DMAs of tile kt have landed (the younger tiles may still fly)
      __builtin_amdgcn_s_barrier();                  // ... everyone's have, and everyone holds tile kt-1's fragments in registers
      const unsigned char* cur = lds + st_cur * STAGE;
      read_frags(cur, 0);
      issue_next(lds + st_free * STAGE);
      st_free = st_cur;
      st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
      mmas(1);              // tile kt-1, second half
      if (kc == KC) {       // ... which closed component `comp`
        combine(comp);
        ++comp;
        kc = 0;
      }
      ++kc;
      read_frags(cur, 1);
      mmas(0);
    }
    mmas(1);
    combine(NC - 1);
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // vmcnt(0): the re-fetches of the tail are not left in flight
  } else if constexpr (HI_ONLY) {
    // one product per term: 8 MFMAs per wave per K-step cannot hide a barrier-to-barrier DMA round trip, so the tiles run
    // through a four-deep ring -- three K-steps of DMA in flight, own arrivals counted with s_waitcnt vmcnt (a compile-time
    // count: the iterator re-fetches the final tile once the reduction is exhausted, so every iteration issues the same DMAs)
    constexpr int INFLIGHT = (NA + NB) * (NSTAGE - 2);
    constexpr int WAITIMM = (INFLIGHT & 15) | (7 << 4) | (15 << 8) | ((INFLIGHT >> 4) << 14);   // vmcnt(INFLIGHT), others untouched
    static_assert(INFLIGHT < 64, "vmcnt is six bits");
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue_next(lds + s * STAGE);
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_waitcnt(WAITIMM);   // own DMAs of tile kt have landed (the NSTAGE-2 younger tiles may still fly)
      __builtin_amdgcn_s_barrier();          // ... everyone's have, and everyone has finished reading tile kt-1's stage
      issue_next(lds + ((kt + NSTAGE - 1) % NSTAGE) * STAGE);
      const unsigned char* cur = lds + (kt % NSTAGE) * STAGE;
      read_frags(cur, 0);
      read_frags(cur, 1);
      mmas(0);
      mmas(1);
    }
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // vmcnt(0): the re-fetches of the tail are not left in flight
  } else if constexpr (NW == 16 && NPROD == 3) {
    // 16 waves x 64x64 (four waves per SIMD: 128 VGPRs each): only the hi x hi third of a K-step's second half is carried
    // across the barrier (16 fragment registers) -- the full skew of the 8-wave variants would need both fragment sets live
    // (36 spills inside the loop: 185 TFLOP/s).
    auto mmas_cross = [&](int ks) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = mma(al[ks][i], bh[ks][j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = mma(ah[ks][i], bl[ks][j], acc[i][j]);
    };
    auto mmas_hh = [&](int ks) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = mma(ah[ks][i], bh[ks][j], acc[i][j]);
    };
    issue_next(lds);
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // vmcnt(0)
    __syncthreads();
    read_frags(lds, 0);
    issue_next(lds + STAGE);
    read_frags(lds, 1);
    mmas(0);
    mmas_cross(1);
    for (int kt = 1; kt < nk; ++kt) {
      __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // own DMAs of tile kt retired: vmcnt(0), spelled out (hipcc may drop it from __syncthreads())
      __syncthreads();      // ... + everyone has read both halves of tile kt-1's stage
      const unsigned char* cur = lds + (kt & 1) * STAGE;
      read_frags(cur, 0);
      issue_next(lds + ((kt + 1) & 1) * STAGE);
      mmas_hh(1);           // tile kt-1: the last four MFMAs cover the fragment reads issued above
      read_frags(cur, 1);
      mmas(0);
      mmas_cross(1);
    }
    mmas_hh(1);
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // vmcnt(0): the re-fetched tail tile has landed before the epilogue reuses LDS
  } else {
    // 8-wave variants: the MFMAs are skewed by half a K-step against the barrier -- when a wave leaves the barrier of tile
    // kt it still owes the second half (ks = 1) of tile kt-1, whose fragments are already in registers, so the burst of
    // fragment reads that all waves issue right after the barrier is covered by MFMAs instead of starving the matrix pipes
    // (128x128 tile 311 -> 335 TFLOP/s, 256x64 186 -> 194, 8-wave 256x256 438 -> 454).
    issue_next(lds);
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // vmcnt(0)
    __syncthreads();
    read_frags(lds, 0);
    issue_next(lds + STAGE);
    read_frags(lds, 1);
    mmas(0);
    for (int kt = 1; kt < nk; ++kt) {
      __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // own DMAs of tile kt retired: vmcnt(0), spelled out (hipcc may drop it from __syncthreads())
      __syncthreads();      // ... + everyone has read both halves of tile kt-1's stage
      const unsigned char* cur = lds + (kt & 1) * STAGE;
      read_frags(cur, 0);
      issue_next(lds + ((kt + 1) & 1) * STAGE);
      mmas(1);              // tile kt-1, second half
      read_frags(cur, 1);
      mmas(0);
    }
    mmas(1);
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // vmcnt(0): the re-fetched tail tile has landed before the epilogue reuses LDS
  }

  // ---- epilogue: scale, bias, BatchNorm partial statistics, activation, store ------------------------------
  float* yg = p.y + (long)g * p.y_gstride;
  const float osc = (p.out_scale ? p.out_scale[g * p.os_stride + 1] : 1.f) * (p.x_scale ? p.x_scale[1] : 1.f);
  const float* rg = p.res ? p.res + (long)g * p.y_gstride : nullptr;
  float csum[WN], csq[WN], bn[WN], cs[WN], sh[WN];
  float amx = 0.f;
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    csum[j] = csq[j] = 0.f;
    const int n = n0 + (wn * WN + j) * 32 + (lane & 31);
    bn[j] = (p.bias && n < p.N) ? p.bias[(long)g * p.N + n] : 0.f;
    cs[j] = (p.ch_scale && n < p.N) ? p.ch_scale[(long)g * p.N + n] : 1.f;
    sh[j] = (p.ch_scale && n < p.N) ? p.ch_shift[(long)g * p.N + n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (m >= p.M) continue;
      int oy, ox, b;
      const long pix0 = (p.oy_major || WINO > 0) ? row_to_pixel(p, m, oy, ox, b) : m;
#pragma unroll
      for (int wr = 0; wr < WR; ++wr) {
      long pix = pix0;
      if constexpr (WINO > 0) {      // GEMM row = group of R output columns: this pass stores column R * group + wr
        if (ox * WR + wr >= p.wino_W) continue;
        pix = ((long)b * p.Ho + oy) * p.wino_W + ox * WR + wr;
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = n0 + (wn * WN + j) * 32 + (lane & 31);
        if (n < p.N) {
          float v = (WINO > 0 ? Y[wr][i][j][e] : acc[i][j][e]) * osc + bn[j];
          if (p.ch_scale) v = v * cs[j] + sh[j];             // eval-mode BatchNorm (running statistics): same expression as bn_apply
          if (rg) v += rg[pix * p.y_ld + n];
          if (HL_OUT && p.res_hl) {
            const unsigned char* rl = p.res_hl + ((long)g * p.y_gstride + pix * p.y_ld + (n & ~31)) * 4 + (n & 31) * 2;
            v += (float)*reinterpret_cast<const _Float16*>(rl) + (float)*reinterpret_cast<const _Float16*>(rl + 64);
          }
          csum[j] += v;
          csq[j] += v * v;
          if (p.act == 1) v = relu_nan(v);
          else if (p.act == 2) v = gelu_fast(v);                                             // GELU (erf), SVTR Mlp
          amx = fmaxf(amx, fabsf(v));
          if (!HL_OUT || p.y) yg[pix * p.y_ld + n] = v;
          if (HL_OUT && p.y_hl) {       // operand of the next GEMM: 32 lanes fill the hi half and the lo half of one 128-byte line
            _Float16 h, l;               // (pairing neighbouring lanes' halves into 4-byte stores measured no better: SVTR x 6 -2 %)
            split_f16(v, h, l);
            unsigned char* line = p.y_hl + ((long)g * p.y_gstride + pix * p.y_ld + (n & ~31)) * 4 + (n & 31) * 2;
            *reinterpret_cast<_Float16*>(line) = h;
            *reinterpret_cast<_Float16*>(line + 64) = l;
          }
        }
      }
      }
    }
  }
  if (p.amax_ws) {        // the range of the result for whoever turns it into the next operand (one atomic per wave, spread over 64 words)
    amx = wave_max(amx);
    if (lane == 0 && !(amx <= 0.f)) atomicMax(p.amax_ws + ((blockIdx.x * (WAVES_M * WAVES_N) + wave) & 63), __float_as_uint(amx));
  }
  if (p.stats) {
    __syncthreads();      // (every main-loop variant has drained its DMAs with an explicit vmcnt(0) before the epilogue)
    float* red = reinterpret_cast<float*>(lds);   // [WAVES_M][2][BN]
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const float s = csum[j] + __shfl_xor(csum[j], 32);
      const float q = csq[j] + __shfl_xor(csq[j], 32);
      if (lane < 32) {
        const int c = (wn * WN + j) * 32 + lane;
        red[(wm * 2 + 0) * BN + c] = s;
        red[(wm * 2 + 1) * BN + c] = q;
      }
    }
    __syncthreads();
    if (t < BN) {
      const int n = n0 + t;
      if (n < p.N) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < WAVES_M; ++r) {
          s += red[(r * 2 + 0) * BN + t];
          q += red[(r * 2 + 1) * BN + t];
        }
        float* sg = p.stats + ((long)g * p.tilesM + tile_m) * 2 * p.N;
        sg[n] = s;
        sg[p.N + n] = q;
      }
    }
  }
}

template __global__ void conv_x3_kernel<2, 4, 4, 2>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 2, 2, 2>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 2, 1, 2>(const ConvX3Params);
template __global__ void conv_x3_kernel<8, 1, 1, 2>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 4, 2, 2>(const ConvX3Params);
template __global__ void conv_x3_kernel<2, 2, 1, 1>(const ConvX3Params);     // 64 x 64: weight-gradient GEMMs of the first layers (Cout, Cin <= 64)
template __global__ void conv_x3_kernel<4, 1, 1, 2>(const ConvX3Params);     // 128 x 64: ... of the Cout = 128, Cin <= 64 layers
template __global__ void conv_x3_kernel<4, 2, 2, 2, true>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 2, 1, 2, true>(const ConvX3Params);
template __global__ void conv_x3_kernel<8, 1, 1, 2, true>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 4, 2, 2, true>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 2, 2, 2, false, 1>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 2, 1, 2, false, 1>(const ConvX3Params);
template __global__ void conv_x3_kernel<8, 1, 1, 2, false, 1>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 4, 2, 2, false, 1>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 2, 2, 2, true, 1>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 2, 1, 2, true, 1>(const ConvX3Params);
template __global__ void conv_x3_kernel<8, 1, 1, 2, true, 1>(const ConvX3Params);
template __global__ void conv_x3_kernel<4, 4, 2, 2, true, 1>(const ConvX3Params);
template __global__ void conv_x3_kernel<2, 4, 2, 1, false, 3, 4>(const ConvX3Params);     // Winograd F(4,3): 128 groups x 128 channels
template __global__ void conv_x3_kernel<2, 4, 2, 1, false, 3, 2>(const ConvX3Params);     // Winograd F(2,3)

// ---- producers of the HL32 layout -------------------------------------------------------------------------------
__device__ __forceinline__ void split_h(float v, _Float16& h, _Float16& l) {
  split_f16(v, h, l);
}

// fp32 [rows][C] -> HL32 [rows][C/32][hi 32 | lo 32]; one thread = 8 channels (32 B in, 2 x 16 B out)
__global__ __launch_bounds__(256) void split_hl32_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, long n8,
                                                         const float* __restrict__ scale) {
  const float sc = scale ? scale[0] : 1.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8) * sc, b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4) * sc;
    f16v8 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 hh, ll;
      split_h(a[e], hh, ll);
      h[e] = hh; l[e] = ll;
      split_h(b[e], hh, ll);
      h[4 + e] = hh; l[4 + e] = ll;
    }
    const long blk = i >> 2;            // 32-channel block index (4 threads per block)
    unsigned char* o = out + blk * 128 + (i & 3) * 16;
    *reinterpret_cast<f16v8*>(o) = h;
    *reinterpret_cast<f16v8*>(o + 64) = l;
  }
}

// Transposed split for weight-gradient GEMMs (dW = dy^T x reduces over ROWS): fp32 x[rows][C] -> `splits` HL32 matrices
// out[s][c][rps/32][hi 32 | lo 32] with rps = rows / splits, element (c, r) = scale * x[s*rps + r][c].  One block moves a
// 32-row x 32-column tile through LDS: coalesced 128-byte row reads, one 128-byte HL32 line per column written.
__global__ __launch_bounds__(256) void split_hl32_t_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int C,
                                                           long rows, long rps32, long tiles_c, long ntiles,
                                                           const float* __restrict__ scale) {
  __shared__ float tile[32][33];
  const float sc = scale ? scale[0] : 1.f;
  const int t = threadIdx.x;
  for (long id = blockIdx.x; id < ntiles; id += gridDim.x) {
    const long rb = id / tiles_c;                 // 32-row block (global)
    const int c0 = (int)(id - rb * tiles_c) * 32;
    {
      const int r = t >> 3, c4 = (t & 7) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c0 + c4 < C && rb * 32 + r < rows) v = *reinterpret_cast<const f32x4*>(x + (rb * 32 + r) * C + c0 + c4);
      tile[r][c4 + 0] = v[0] * sc; tile[r][c4 + 1] = v[1] * sc; tile[r][c4 + 2] = v[2] * sc; tile[r][c4 + 3] = v[3] * sc;
    }
    __syncthreads();
    {
      const int c = t >> 3, seg = t & 7;          // column c, rows seg*4 .. +3
      if (c0 + c < C) {
        const long s_ = rb / rps32, rl = rb - s_ * rps32;
        unsigned char* o = out + ((s_ * C + c0 + c) * rps32 + rl) * 128 + seg * 8;
        typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
        f16v4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 hh, ll;
          split_h(tile[seg * 4 + e][c], hh, ll);
          h[e] = hh; l[e] = ll;
        }
        *reinterpret_cast<f16v4*>(o) = h;
        *reinterpret_cast<f16v4*>(o + 64) = l;
      }
    }
    __syncthreads();
  }
}

// The same transposed split WITH the column sums of x (the bias gradient of a Linear layer is the column sum of the very dy this pass
// transposes for the weight gradient: two more passes over dy, 100 launches and 2.5 ms of an SVTR loop-A step's weight-gradient stream).
// grid (column tiles, chunks): a block keeps ONE 32-column tile and walks the 32-row blocks of its chunk four at a time (four 16-byte
// loads in flight per lane); its column sums (of the scaled values: the scale is a power of two, so 1/s * sum(s x) == sum(x) in the
// same order) go to partial[chunk][C] (one chunk: straight into colsum); a column-sum pass over those <= 256 rows finishes.  [Measured:
// the finish inside this kernel -- last block to arrive per column tile, __threadfence() + ticket -- made it 8-15 x slower, 39 -> 324 us
// at 131072 x 256: every block's agent-scope release writes back an L2 that 2048 blocks keep dirtying with 134 MB of operand lines.]
__global__ __launch_bounds__(256) void split_hl32_t_colsum_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int C,
                                                                  long rows, long rps32, long rblocks, long rb_per_chunk,
                                                                  const float* __restrict__ scale, float* __restrict__ partial,
                                                                  float* __restrict__ colsum, int accumulate) {
  __shared__ float tile[4][32][33];
  const float sc = scale ? scale[0] : 1.f, inv = scale ? scale[1] : 1.f;
  const int t = threadIdx.x;
  const int c0 = blockIdx.x * 32, chunk = blockIdx.y, nchunks = gridDim.y;
  const long rb0 = chunk * rb_per_chunk, rb1 = min(rblocks, rb0 + rb_per_chunk);
  const int r = t >> 3, c4 = (t & 7) * 4;          // load role: row r of a row block, columns c4 .. +3
  const int c = t >> 3, seg = t & 7;               // store role: column c, rows seg*4 .. +3
  float acc = 0.f;
  for (long rb = rb0; rb < rb1; rb += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      const long row = (rb + u) * 32 + r;
      if (rb + u < rb1 && c0 + c4 < C && row < rows) v[u] = *reinterpret_cast<const f32x4*>(x + row * C + c0 + c4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      tile[u][r][c4 + 0] = v[u][0] * sc; tile[u][r][c4 + 1] = v[u][1] * sc;
      tile[u][r][c4 + 2] = v[u][2] * sc; tile[u][r][c4 + 3] = v[u][3] * sc;
    }
    __syncthreads();
    if (c0 + c < C) {
      typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (rb + u >= rb1) break;
        const long s_ = (rb + u) / rps32, rl = (rb + u) - s_ * rps32;
        unsigned char* o = out + ((s_ * C + c0 + c) * rps32 + rl) * 128 + seg * 8;
        f16v4 h, l;
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f = tile[u][seg * 4 + e][c];
          _Float16 hh, ll;
          split_h(f, hh, ll);
          h[e] = hh; l[e] = ll;
          a += f;
        }
        acc += a;
        *reinterpret_cast<f16v4*>(o) = h;
        *reinterpret_cast<f16v4*>(o + 64) = l;
      }
    }
    __syncthreads();
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  acc *= inv;
  if (nchunks == 1) {
    if (seg == 0 && c0 + c < C) colsum[c0 + c] = accumulate ? colsum[c0 + c] + acc : acc;
    return;
  }
  if (seg == 0 && c0 + c < C) partial[(long)chunk * C + c0 + c] = acc;
}

struct Im2colT {
  const float* x; unsigned char* out; const float* scale;
  int B, H, W, C, Ho, Wo, kw, taps, sh, sw, ph, pw;
  long P, rps32, tiles_c, ntiles;
};

// one block = (32 output pixels) x (32 channels) of ONE tap; the tap index runs fastest over the grid so the nine taps of a
// pixel block re-read the same input lines from L2
__global__ __launch_bounds__(256) void im2col_t_hl32_kernel(const Im2colT p) {
  __shared__ float tile[32][33];
  const float sc = p.scale ? p.scale[0] : 1.f;
  const int t = threadIdx.x;
  for (long id = blockIdx.x; id < p.ntiles; id += gridDim.x) {
    const int tap = (int)(id % p.taps);
    const long id2 = id / p.taps;
    const long rb = id2 / p.tiles_c;                 // 32-pixel block (global over the padded pixel axis)
    const int c0 = (int)(id2 - rb * p.tiles_c) * 32;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    {
      const int r = t >> 3, c4 = (t & 7) * 4;
      const long pix = rb * 32 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pix < p.P && c0 + c4 < p.C) {
        const int hw = p.Ho * p.Wo;
        const int b = (int)(pix / hw);
        const int rem = (int)(pix - (long)b * hw);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy = oy * p.sh - p.ph + ky, ix = ox * p.sw - p.pw + kx;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          v = *reinterpret_cast<const f32x4*>(p.x + (((long)b * p.H + iy) * p.W + ix) * p.C + c0 + c4);
      }
      tile[r][c4 + 0] = v[0] * sc; tile[r][c4 + 1] = v[1] * sc; tile[r][c4 + 2] = v[2] * sc; tile[r][c4 + 3] = v[3] * sc;
    }
    __syncthreads();
    {
      const int c = t >> 3, seg = t & 7;
      if (c0 + c < p.C) {
        const long s_ = rb / p.rps32, rl = rb - s_ * p.rps32;
        unsigned char* o = p.out + ((((s_ * p.taps + tap) * p.C) + c0 + c) * p.rps32 + rl) * 128 + seg * 8;
        typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
        f16v4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 hh, ll;
          split_h(tile[seg * 4 + e][c], hh, ll);
          h[e] = hh; l[e] = ll;
        }
        *reinterpret_cast<f16v4*>(o) = h;
        *reinterpret_cast<f16v4*>(o + 64) = l;
      }
    }
    __syncthreads();
  }
}

// Transposed split in IMAGE-ROW-MAJOR pixel order, optionally shifted along x: NHWC fp32 x[b][y][xx][c] -> out[c][k / 32][hi 32 | lo 32]
// with k = (y * B + b) * W + xx and element (c, k) = scale * x[b][y][xx + shift][c] (0 outside the row; k >= P: 0).  The operand layout
// of the convolution weight gradient without an im2col (mrn_gemm_x3_windows_hl32): in this order a kernel-row offset ky - 1 is
// a whole number of lines (B * W pixels) and only the THREE horizontal shifts need their own copy.  One block = 32 pixels x 32 channels.
__global__ __launch_bounds__(256) void transpose_oy_hl32_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int B,
                                                                int H, int W, int C, int shift, long P, long lines, long tiles_c,
                                                                long ntiles, const float* __restrict__ scale) {
  __shared__ float tile[32][33];
  const float sc = scale ? scale[0] : 1.f;
  const int t = threadIdx.x;
  for (long id = blockIdx.x; id < ntiles; id += gridDim.x) {
    const long kb = id / tiles_c;                 // 32-pixel block along k
    const int c0 = (int)(id - kb * tiles_c) * 32;
    {
      const int r = t >> 3, c4 = (t & 7) * 4;
      const long k = kb * 32 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k < P && c0 + c4 < C) {
        const long row = k / W;                   // = y * B + b
        const int xx = (int)(k - row * W) + shift;
        const int y = (int)(row / B), b = (int)(row - (long)y * B);
        if ((unsigned)xx < (unsigned)W) v = *reinterpret_cast<const f32x4*>(x + (((long)b * H + y) * W + xx) * C + c0 + c4);
      }
      tile[r][c4 + 0] = v[0] * sc; tile[r][c4 + 1] = v[1] * sc; tile[r][c4 + 2] = v[2] * sc; tile[r][c4 + 3] = v[3] * sc;
    }
    __syncthreads();
    {
      const int c = t >> 3, seg = t & 7;          // column c, pixels seg*4 .. +3
      if (c0 + c < C) {
        unsigned char* o = out + ((long)(c0 + c) * lines + kb) * 128 + seg * 8;
        typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
        f16v4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 hh, ll;
          split_h(tile[seg * 4 + e][c], hh, ll);
          h[e] = hh; l[e] = ll;
        }
        *reinterpret_cast<f16v4*>(o) = h;
        *reinterpret_cast<f16v4*>(o + 64) = l;
      }
    }
    __syncthreads();
  }
}

// The three horizontal shifts (-1, 0, +1) of transpose_oy_hl32_kernel from ONE read of the activation: out[s][c][k / 32][...] for
// s = 0, 1, 2 (copy stride `copy_bytes`).  A block stages pixels k0 - 1 .. k0 + 32 (34 rows) x 32 channels.
__global__ __launch_bounds__(256) void transpose_oy3_hl32_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int B,
                                                                 int H, int W, int C, long P, long lines, long tiles_c, long ntiles,
                                                                 long copy_bytes, const float* __restrict__ scale) {
  __shared__ float tile[34][33];
  __shared__ int xpos[34];
  const float sc = scale ? scale[0] : 1.f;
  const int t = threadIdx.x;
  for (long id = blockIdx.x; id < ntiles; id += gridDim.x) {
    const long kb = id / tiles_c;
    const int c0 = (int)(id - kb * tiles_c) * 32;
    for (int i = t; i < 34 * 8; i += 256) {
      const int r = i >> 3, c4 = (i & 7) * 4;
      const long k = kb * 32 + r - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      int xx = -1000;
      if (k >= 0 && k < P) {
        const long row = k / W;
        xx = (int)(k - row * W);
        const int y = (int)(row / B), b = (int)(row - (long)y * B);
        if (c0 + c4 < C) v = *reinterpret_cast<const f32x4*>(x + (((long)b * H + y) * W + xx) * C + c0 + c4);
      }
      if (c4 == 0) xpos[r] = xx;
      tile[r][c4 + 0] = v[0] * sc; tile[r][c4 + 1] = v[1] * sc; tile[r][c4 + 2] = v[2] * sc; tile[r][c4 + 3] = v[3] * sc;
    }
    __syncthreads();
    {
      const int c = t >> 3, seg = t & 7;
      if (c0 + c < C) {
        typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int s = 0; s < 3; ++s) {              // shift = s - 1: destination pixel j takes source pixel j + shift of the SAME image row
          f16v4 h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = seg * 4 + e;             // destination row of the tile is j + 1, its source j + s
            const int xd = xpos[j + 1];
            const bool ok = xd >= 0 && (unsigned)(xd + s - 1) < (unsigned)W;
            _Float16 hh, ll;
            split_h(ok ? tile[j + s][c] : 0.f, hh, ll);
            h[e] = hh; l[e] = ll;
          }
          unsigned char* o = out + s * copy_bytes + ((long)(c0 + c) * lines + kb) * 128 + seg * 8;
          *reinterpret_cast<f16v4*>(o) = h;
          *reinterpret_cast<f16v4*>(o + 64) = l;
        }
      }
    }
    __syncthreads();
  }
}

// Transposed operands of the Winograd-domain weight gradient (mrn_conv_wgrad_wino...): NHWC fp32 t[b][y][xx][c] -> out[m][c][kq / 32][hi 32 | lo 32]
// over GROUPS of 4 columns in image-row-major order kq = (y * B + b) * Wq + q (Wq = ceil(W / 4)), six components m per group:
//   MODE 0 (the layer input x):       V_m  = sum_j B^T[m][j] * x[.., 4q - 1 + j]   (j = 0..5, zero outside the row)
//   MODE 1 (the output gradient dy):  dY_m = sum_r A^T[r][m] * dy[.., 4q + r]      (r = 0..3, zero beyond W)
// times scale[0].  With these, dU_m[ky] = sum over groups of dY_m (x) V_m (row-shifted by ky - 1) and dW[ky][kx] = sum_m G[m][kx] dU_m[ky]:
// 18 K-windows of length P/4 instead of 9 of length P -- half the matrix work -- and ONE 1.5x copy of x^T instead of three shifted copies.
// One block = 32 groups x 32 channels through LDS.
template <int MODE>
__global__ __launch_bounds__(256) void transpose_oy_wino_hl32_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int B, int H,
                                                                     int W, int Wq, int C, long Pq, long lines, long tiles_c, long ntiles,
                                                                     const float* __restrict__ scale) {
  constexpr int NCOL = MODE == 0 ? 6 : 4;
  __shared__ float tile[32 * NCOL][33];
  const float sc = scale ? scale[0] : 1.f;
  const int t = threadIdx.x;
  for (long id = blockIdx.x; id < ntiles; id += gridDim.x) {
    const long kb = id / tiles_c;
    const int c0 = (int)(id - kb * tiles_c) * 32;
    for (int i = t; i < 32 * NCOL * 8; i += 256) {
      const int rr = i >> 3, c4 = (i & 7) * 4;          // rr = group-in-block * NCOL + column
      const int gq = rr / NCOL, j = rr - gq * NCOL;
      const long kq = kb * 32 + gq;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kq < Pq && c0 + c4 < C) {
        const long row = kq / Wq;                        // = y * B + b
        const int q = (int)(kq - row * Wq);
        const int xx = 4 * q + j - (MODE == 0 ? 1 : 0);
        const int y = (int)(row / B), b = (int)(row - (long)y * B);
        if ((unsigned)xx < (unsigned)W) v = *reinterpret_cast<const f32x4*>(x + (((long)b * H + y) * W + xx) * C + c0 + c4);
      }
      tile[rr][c4 + 0] = v[0]; tile[rr][c4 + 1] = v[1]; tile[rr][c4 + 2] = v[2]; tile[rr][c4 + 3] = v[3];
    }
    __syncthreads();
    {
      const int c = t >> 3, seg = t & 7;                 // channel c, groups seg*4 .. +3 of the block
      if (c0 + c < C) {
        typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
        f16v4 h[6], l[6];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int gq = seg * 4 + e;
          float m_[6];
          if constexpr (MODE == 0) {
            const float d0 = tile[gq * 6 + 0][c], d1 = tile[gq * 6 + 1][c], d2 = tile[gq * 6 + 2][c], d3 = tile[gq * 6 + 3][c],
                        d4 = tile[gq * 6 + 4][c], d5 = tile[gq * 6 + 5][c];
            m_[0] = 4.f * d0 - 5.f * d2 + d4;
            m_[1] = -4.f * (d1 + d2) + d3 + d4;
            m_[2] = 4.f * (d1 - d2) - d3 + d4;
            m_[3] = 2.f * (d3 - d1) - d2 + d4;
            m_[4] = 2.f * (d1 - d3) - d2 + d4;
            m_[5] = 4.f * d1 - 5.f * d3 + d5;
          } else {
            const float y0 = tile[gq * 4 + 0][c], y1 = tile[gq * 4 + 1][c], y2 = tile[gq * 4 + 2][c], y3 = tile[gq * 4 + 3][c];
            m_[0] = y0;                                  // columns of A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
            m_[1] = y0 + y1 + y2 + y3;
            m_[2] = y0 - y1 + y2 - y3;
            m_[3] = y0 + 2.f * y1 + 4.f * y2 + 8.f * y3;
            m_[4] = y0 - 2.f * y1 + 4.f * y2 - 8.f * y3;
            m_[5] = y3;
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            _Float16 hh, ll;
            split_h(m_[k] * sc, hh, ll);
            h[k][e] = hh; l[k][e] = ll;
          }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          unsigned char* o = out + (((long)k * C + c0 + c) * lines + kb) * 128 + seg * 8;
          *reinterpret_cast<f16v4*>(o) = h[k];
          *reinterpret_cast<f16v4*>(o + 64) = l[k];
        }
      }
    }
    __syncthreads();
  }
}

// dW [Cout][3][3][Cin] = sum over split-K chunks s and components m of G[m][kx] * part[s][m * 3 + ky][co][ci]  (G of F(4,3), unfolded);
// one lane = 4 consecutive input channels (Cin % 4 == 0) of ONE kernel row (blockIdx.y = ky: the three rows are independent, and a
// 512 x 512 layer has only 65536 channel quads -- one wave per SIMD without the split): 6 accumulators, two split-K slabs in flight
// OIHW_ACC: dW is ADDED into the parameter's own [Cout][Cin][3][3] gradient (the flat-gradient slice)
template <bool OIHW_ACC>
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, long CC4, int Cin4) {
  const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
  const int ky = blockIdx.y;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < CC4; i += (long)gridDim.x * 256) {      // i = (co * Cin + ci) / 4
    const long co = i / Cin4;
    const int c4 = (int)(i - co * Cin4);
    f32x4 u[6], w[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) u[m] = w[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    int s_ = 0;
    for (; s_ + 1 < S; s_ += 2) {
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        u[m] += p4[((long)s_ * 18 + m * 3 + ky) * CC4 + i];
        w[m] += p4[((long)(s_ + 1) * 18 + m * 3 + ky) * CC4 + i];
      }
    }
    if (s_ < S) {
#pragma unroll
      for (int m = 0; m < 6; ++m) u[m] += p4[((long)s_ * 18 + m * 3 + ky) * CC4 + i];
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) u[m] += w[m];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 6; ++m) v += G[m][kx] * u[m];
      if constexpr (OIHW_ACC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dw[((co * Cin4 + c4) * 4 + j) * 9 + ky * 3 + kx] += v[j];
      } else {
        reinterpret_cast<f32x4*>(dw)[((co * 3 + ky) * 3 + kx) * Cin4 + c4] = v;
      }
    }
  }
}

// w [Cout][taps][Cin] fp32 (x scale[0]) -> [Cout][Cin/32][taps][hi 32 | lo 32]; one thread = 8 channels
__global__ __launch_bounds__(256) void pack_weight_hl32_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                               int Cout, int taps, int Cin, const float* __restrict__ scale) {
  const float sc = scale ? scale[0] : 1.f;
  const int Cb = Cin >> 5;
  const long n8 = (long)Cout * taps * (Cin >> 3);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % (Cin >> 3));
    const long r = i / (Cin >> 3);
    const int tap = (int)(r % taps);
    const long o_ = r / taps;
    const float* src = w + (o_ * taps + tap) * Cin + c8 * 8;
    f16v8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 hh, ll;
      split_h(src[e] * sc, hh, ll);
      h[e] = hh; l[e] = ll;
    }
    const int cb = c8 >> 2;
    unsigned char* o = out + ((o_ * Cb + cb) * taps + tap) * 128 + (c8 & 3) * 16;
    *reinterpret_cast<f16v8*>(o) = h;
    *reinterpret_cast<f16v8*>(o + 64) = l;
  }
}

// (magic, shift) with n / d == umulhi(n, magic) >> shift for every 0 <= n < 2^31
void magic_div(unsigned d, unsigned& magic, int& shift) {
  if (d <= 1) {            // encoded as magic 0 (fast_div returns n)
    magic = 0;
    shift = 0;
    return;
  }
  int s = 0;
  while ((1u << s) < d) ++s;                       // s = ceil(log2 d) >= 1
  // ceil(2^(31+s) / d) is a 32-bit value in [2^31, 2^32); with shift s-1 the quotient is exact for n < 2^31
  magic = (unsigned)(((1ull << (31 + s)) + d - 1) / d);
  shift = s - 1;
}

// A/B switches of the tile order, read from the environment ONCE (not per launch)
struct EnvFlags { bool no_interleave, no_class_order, row_major, w8; };
const EnvFlags& env_flags() {
  static const EnvFlags f = {getenv("MRN_X3_NO_INTERLEAVE") != nullptr, getenv("MRN_X3_NO_CLASS_ORDER") != nullptr,
                             getenv("MRN_X3_ROW_MAJOR") != nullptr, getenv("MRN_X3_W8") != nullptr};
  return f;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, bool HL_OUT = false, int NPROD = 3, int WINO = 0>
int launch_x3(const ConvX3Params& p0, hipStream_t st) {
  constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
  ConvX3Params p = p0;
  p.tilesM = ceil_div(p.M, BM);
  p.tilesN = ceil_div(p.N, BN);
  p.tiles_per_row = (p.oy_major && p.BWo % BM == 0 && !env_flags().no_interleave) ? p.BWo / BM : 0;
  p.class_order = (p.tiles_per_row > 0 && p.Ho >= 3 && p.kh == 3 && p.ph == 1 && p.sh == 1 && !env_flags().no_class_order) ? 1 : 0;
  const size_t ldsz = x3_lds_bytes<WAVES_M, WAVES_N, WM, WN, NPROD, WINO>();
  const long tiles = (long)p.G * p.tilesM * p.tilesN;
  static bool attr_set = false;      // (per instantiation: the attribute is sticky, one driver call instead of one per launch)
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_x3_kernel<WAVES_M, WAVES_N, WM, WN, HL_OUT, NPROD, WINO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_x3_kernel<WAVES_M, WAVES_N, WM, WN, HL_OUT, NPROD, WINO>), dim3((unsigned)tiles), dim3(WAVES_M * WAVES_N * 64), ldsz, st, p);
  MRN_LAUNCH_CHECK("conv2d_x3_hl32");
  return MRN_OK;
}

}  // namespace

MRN_EXPORT int64_t mrn_conv2d_x3_stats_floats(int G, int B, int Ho, int Wo, int Cout, int tile_m) {
  return (int64_t)G * ceil_div((long)B * Ho * Wo, tile_m) * 2 * Cout;
}

// tile_m x tile_n: 256x256 (Cout >= 256), 256x128, or 128x128 (two workgroups per CU: short reductions)
MRN_EXPORT int mrn_conv2d_x3_hl32(const void* x_hl, const void* w_hl, const void* zero_page, const float* bias,
                                  const float* residual, float* y, float* stats, const float* out_scale,
                                  const float* x_scale, int G, int64_t x_group_stride_bytes, int B, int H,
                                  int W, int Cin, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int act,
                                  int tile_m, int tile_n, int64_t y_row_stride, int64_t y_group_stride, int x_group_div,
                                  void* y_hl32, int products, const float* ch_scale, const float* ch_shift,
                                  const void* residual_hl32, void* amax_ws, void* stream) {
  MRN_CHECK_ARG(x_hl && w_hl && zero_page && (y || y_hl32) && G >= 1, "mrn_conv2d_x3_hl32: bad operands");
  MRN_CHECK_ARG(!y_hl32 || (Cout % 32 == 0 && y_row_stride <= 0 && y_group_stride <= 0 && (uintptr_t)y_hl32 % 128 == 0),
                "mrn_conv2d_x3_hl32: the HL32 result needs Cout %% 32 == 0 and dense rows");
  MRN_CHECK_ARG(Cin % 32 == 0 && kh * kw <= 32, "mrn_conv2d_x3_hl32: unsupported Cin=%d kernel=%dx%d", Cin, kh, kw);
  MRN_CHECK_ARG((!ch_scale == !ch_shift) && (!residual_hl32 || (y_hl32 && (uintptr_t)residual_hl32 % 128 == 0)),
                "mrn_conv2d_x3_hl32: ch_scale / ch_shift come as a pair; an HL32 residual needs the HL32 result (same geometry)");
  MRN_CHECK_ARG(products == 3 || products == 1, "mrn_conv2d_x3_hl32: products must be 3 (split-fp16 x3) or 1 (hi x hi), got %d", products);
  MRN_CHECK_ARG(((uintptr_t)x_hl % 128 == 0) && ((uintptr_t)w_hl % 128 == 0) && ((uintptr_t)zero_page % 16 == 0) &&
                    (x_group_stride_bytes % 128 == 0), "mrn_conv2d_x3_hl32: HL32 operands must be 128-byte aligned");
  MRN_CHECK_ARG((tile_m == 256 && (tile_n == 256 || tile_n == 128 || tile_n == 64)) || (tile_m == 128 && tile_n == 128),
                "mrn_conv2d_x3_hl32: tile must be 256x256, 256x128, 256x64 or 128x128");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  MRN_CHECK_ARG(Ho > 0 && Wo > 0 && B >= 0, "mrn_conv2d_x3_hl32: empty output");
  MRN_CHECK_ARG((long)B * H * W * Cin * 4 < (1L << 31) && (long)Cout * kh * kw * Cin * 4 < (1L << 31),
                "mrn_conv2d_x3_hl32: one group's activation / weight must stay below 2 GiB (32-bit buffer offsets)");
  ConvX3Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const unsigned char*)x_hl; p.w = (const unsigned char*)w_hl; p.zero = (const unsigned char*)zero_page;
  p.amax_ws = (unsigned*)amax_ws;
  p.bias = bias; p.out_scale = out_scale; p.os_stride = 2; p.x_scale = x_scale; p.res = residual; p.y = y; p.y_hl = (unsigned char*)y_hl32; p.stats = stats;
  p.ch_scale = ch_scale; p.ch_shift = ch_shift; p.res_hl = (const unsigned char*)residual_hl32;
  p.Cb = Cin / 32; p.taps = kh * kw; p.nk = p.Cb * p.taps;
  p.x_gstride = x_group_stride_bytes; p.w_gstride = (long)Cout * p.nk * 128;
  p.x_group_div = x_group_div > 1 ? x_group_div : 1;
  p.x_bytes = (int)((long)B * H * W * Cin * 4);
  p.G = G; p.M = B * Ho * Wo; p.N = Cout;
  p.y_ld = y_row_stride > 0 ? y_row_stride : Cout;
  p.y_gstride = y_group_stride > 0 ? y_group_stride : (long)p.M * p.y_ld;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.act = act;
  if (p.M == 0) return MRN_OK;
  // image-row-major tiles pay off when the kernel is padded vertically and the maps are only a few rows high (the TRBA
  // backbone's 4 x 65 maps: the two border rows skip one of three kernel rows)
  p.oy_major = (ph > 0 && kh > 1 && Ho <= 8) ? 1 : 0;   // (taller maps lose more L2 reuse across kernel rows than they skip)
  if (env_flags().row_major) p.oy_major = 0;
  p.BWo = B * Wo;
  magic_div((unsigned)Wo, p.wo_magic, p.wo_shift);
  magic_div((unsigned)p.BWo, p.bw_magic, p.bw_shift);
  magic_div((unsigned)kw, p.kw_magic, p.kw_shift);
  // 256x256: 16 waves (64x64 wave tiles, four waves per SIMD) hide the per-K-step LDS / barrier stalls better than 8 waves
  // with 128x64 tiles: 464 vs 438 TFLOP/s on the dominant shape (MRN_X3_W8=1 selects the 8-wave variant for A/B runs)
  if (products == 1) {   // reduced-precision mode: one fp16 product per term, the lo halves stay unread
    hipStream_t st1 = (hipStream_t)stream;
    if (y_hl32) {
      if (tile_n == 256) return launch_x3<4, 4, 2, 2, true, 1>(p, st1);
      if (tile_m == 256 && tile_n == 64) return launch_x3<8, 1, 1, 2, true, 1>(p, st1);
      if (tile_m == 256) return launch_x3<4, 2, 2, 2, true, 1>(p, st1);
      return launch_x3<4, 2, 1, 2, true, 1>(p, st1);
    }
    if (tile_n == 256) return launch_x3<4, 4, 2, 2, false, 1>(p, st1);
    if (tile_m == 256 && tile_n == 64) return launch_x3<8, 1, 1, 2, false, 1>(p, st1);
    if (tile_m == 256) return launch_x3<4, 2, 2, 2, false, 1>(p, st1);
    return launch_x3<4, 2, 1, 2, false, 1>(p, st1);
  }
  if (y_hl32) {     // separate instantiations: the store path of the plain kernels stays as it was
    if (tile_n == 256) return launch_x3<4, 4, 2, 2, true>(p, (hipStream_t)stream);
    if (tile_m == 256 && tile_n == 64) return launch_x3<8, 1, 1, 2, true>(p, (hipStream_t)stream);
    if (tile_m == 256) return launch_x3<4, 2, 2, 2, true>(p, (hipStream_t)stream);
    return launch_x3<4, 2, 1, 2, true>(p, (hipStream_t)stream);
  }
  if (tile_n == 256 && env_flags().w8) return launch_x3<2, 4, 4, 2>(p, (hipStream_t)stream);
  if (tile_n == 256) return launch_x3<4, 4, 2, 2>(p, (hipStream_t)stream);
  if (tile_m == 256 && tile_n == 64) return launch_x3<8, 1, 1, 2>(p, (hipStream_t)stream);
  if (tile_m == 256) return launch_x3<4, 2, 2, 2>(p, (hipStream_t)stream);
  return launch_x3<4, 2, 1, 2>(p, (hipStream_t)stream);
}

// fp32 [rows][C] (C % 32 == 0) -> HL32 of scale[0] * x (scale: device float[2] from mrn_pow2_scale_f32, or NULL)
MRN_EXPORT int mrn_split_hl32_f32(const float* x, void* out, int64_t rows, int C, const float* scale, void* stream) {
  MRN_CHECK_ARG(x && out && C % 32 == 0, "mrn_split_hl32_f32: bad operands (C=%d)", C);
  const long n8 = rows * (C / 8);
  if (n8 == 0) return MRN_OK;
  long grid = (n8 + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(split_hl32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned char*)out, n8, scale);
  MRN_LAUNCH_CHECK("split_hl32");
  return MRN_OK;
}

// w [Cout][kh*kw][Cin] fp32 -> HL32 weight [Cout][Cin/32][kh*kw][128 B], multiplied by scale[0] (device) when given
MRN_EXPORT int mrn_pack_weight_hl32(const float* w_ohwi, void* out, int Cout, int taps, int Cin, const float* scale,
                                    void* stream) {
  MRN_CHECK_ARG(w_ohwi && out && Cin % 32 == 0, "mrn_pack_weight_hl32: bad operands (Cin=%d)", Cin);
  const long n8 = (long)Cout * taps * (Cin / 8);
  if (n8 == 0) return MRN_OK;
  long grid = (n8 + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(pack_weight_hl32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w_ohwi,
                     (unsigned char*)out, Cout, taps, Cin, scale);
  MRN_LAUNCH_CHECK("pack_weight_hl32");
  return MRN_OK;
}

// Every trained Linear weight's operands for the coming step in ONE call (3 stream operations): the per-weight sequence max|W| ->
// power-of-two prescale -> HL32 pack [-> the same for W^T, the data gradient's operand] was 6 launches per layer, ~310 per SVTR loop-A
// step, issued by the host while the forward pass waited (a host-bound step: 1287 dispatches in 27 ms of host time).
// desc: n x 8 int64 on the device {W (fp32 [N][K]), out, scale (float[2]), N, K, transposed, first_tile, tiles_i}: the operand is
// L = W (O = N, I = K) or L = W^T (O = K, I = N), out [O][I/32][128 B] = HL32 of s * L, s = 2^floor(log2(target / max|W|)) exactly as
// mrn_pow2_scale_f32 + mrn_pack_weight_hl32 produce it.  amax: n zeroed-by-this-call words.  One block = one 32 x 32 tile of L.
namespace {
struct MultiPackDesc { const float* w; unsigned char* out; float* scale; long N, K, transposed, first_tile, tiles_i; };

__device__ __forceinline__ int multi_pack_find(const MultiPackDesc* __restrict__ d, int n, long tile) {
  int lo = 0, hi = n - 1;                       // last descriptor with first_tile <= tile
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].first_tile <= tile) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void multi_pack_amax_kernel(const MultiPackDesc* __restrict__ desc, int n, unsigned* __restrict__ amax) {
  __shared__ float scratch[4];
  const int di = multi_pack_find(desc, n, blockIdx.x);
  const MultiPackDesc d = desc[di];
  const long tl = blockIdx.x - d.first_tile;
  const long to = tl / d.tiles_i, ti = tl - to * d.tiles_i;
  // the tile of L as a 32 x 32 region of the SOURCE matrix [N][K]: rows = o (plain) or i (transposed), columns the other index
  const long r0 = (d.transposed ? ti : to) * 32, c0 = (d.transposed ? to : ti) * 32;
  const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  float m = 0.f;
  if (r0 + r < d.N && c0 + c4 < d.K) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(d.w + (r0 + r) * d.K + c0 + c4);
    m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  m = block_max<256>(m, scratch);
  if (threadIdx.x == 0 && !(m <= 0.f)) atomicMax(amax + di, __float_as_uint(m));      // NaN / inf pass through
}

__global__ __launch_bounds__(256) void multi_pack_kernel(const MultiPackDesc* __restrict__ desc, int n, const unsigned* __restrict__ amax,
                                                         float target) {
  __shared__ float tile[32][33];
  const int di = multi_pack_find(desc, n, blockIdx.x);
  const MultiPackDesc d = desc[di];
  const long tl = blockIdx.x - d.first_tile;
  const long to = tl / d.tiles_i, ti = tl - to * d.tiles_i;
  const float mm = __uint_as_float(amax[di]);
  float sc = 1.f;
  if (mm > 0.f && isfinite(mm)) sc = exp2f(floorf(log2f(target / mm)));
  if (tl == 0 && threadIdx.x == 0) {
    d.scale[0] = sc;
    d.scale[1] = 1.f / sc;
  }
  const long O = d.transposed ? d.K : d.N, Cb = (d.transposed ? d.N : d.K) >> 5;
  const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!d.transposed) {
    if (to * 32 + r < O) v = *reinterpret_cast<const f32x4*>(d.w + (to * 32 + r) * d.K + ti * 32 + c4);
    tile[r][c4 + 0] = v[0] * sc; tile[r][c4 + 1] = v[1] * sc; tile[r][c4 + 2] = v[2] * sc; tile[r][c4 + 3] = v[3] * sc;
  } else {        // source row = i, source columns = o: lands transposed
    if (to * 32 + c4 < O) v = *reinterpret_cast<const f32x4*>(d.w + (ti * 32 + r) * d.K + to * 32 + c4);
    tile[c4 + 0][r] = v[0] * sc; tile[c4 + 1][r] = v[1] * sc; tile[c4 + 2][r] = v[2] * sc; tile[c4 + 3][r] = v[3] * sc;
  }
  __syncthreads();
  const int o = threadIdx.x >> 3, seg = threadIdx.x & 7;          // row o of the tile, elements seg*4 .. +3 of its 32-channel line
  if (to * 32 + o < O) {
    typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
    f16v4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 hh, ll;
      split_h(tile[o][seg * 4 + e], hh, ll);
      h[e] = hh; l[e] = ll;
    }
    unsigned char* dst = d.out + ((to * 32 + o) * Cb + ti) * 128 + seg * 8;
    *reinterpret_cast<f16v4*>(dst) = h;
    *reinterpret_cast<f16v4*>(dst + 64) = l;
  }
}
}  // namespace

MRN_EXPORT int mrn_multi_pack_linear_hl32(const void* desc, int n, int64_t tiles, void* amax, float target, void* stream) {
  MRN_CHECK_ARG(desc && amax && n > 0 && tiles > 0 && tiles < (1L << 31) && target > 0.f, "mrn_multi_pack_linear_hl32: bad operands (n=%d)", n);
  const hipStream_t st = (hipStream_t)stream;
  const hipError_t e = hipMemsetAsync(amax, 0, sizeof(unsigned) * (size_t)n, st);
  MRN_CHECK_ARG(e == hipSuccess, "mrn_multi_pack_linear_hl32: memset failed (%s)", hipGetErrorString(e));
  hipLaunchKernelGGL(multi_pack_amax_kernel, dim3((unsigned)tiles), dim3(256), 0, st, (const MultiPackDesc*)desc, n, (unsigned*)amax);
  hipLaunchKernelGGL(multi_pack_kernel, dim3((unsigned)tiles), dim3(256), 0, st, (const MultiPackDesc*)desc, n, (const unsigned*)amax, target);
  MRN_LAUNCH_CHECK("multi_pack_linear_hl32");
  return MRN_OK;
}

// fp32 x[rows][C] (C % 4 == 0) -> `splits` transposed HL32 matrices [splits][C][rows_padded/splits/32][128 B] of scale[0] * x
// (rows_padded % (32 * splits) == 0, rows beyond `rows` are zero): the operand layout of a weight-gradient GEMM that
// reduces over rows, split-K = groups.
MRN_EXPORT int mrn_split_hl32_t_f32(const float* x, void* out, int64_t rows, int64_t rows_padded, int C, int splits,
                                    const float* scale, void* stream) {
  MRN_CHECK_ARG(x && out && splits >= 1 && C % 4 == 0 && rows_padded >= rows && rows_padded % (32L * splits) == 0,
                "mrn_split_hl32_t_f32: bad operands (rows=%ld padded=%ld C=%d splits=%d)", (long)rows, (long)rows_padded, C, splits);
  if (rows_padded == 0 || C == 0) return MRN_OK;
  const long tiles_c = (C + 31) / 32, ntiles = rows_padded / 32 * tiles_c;
  long grid = ntiles > 65536 ? 65536 : ntiles;
  hipLaunchKernelGGL(split_hl32_t_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned char*)out, C,
                     (long)rows, (long)(rows_padded / splits / 32), tiles_c, ntiles, scale);
  MRN_LAUNCH_CHECK("split_hl32_t");
  return MRN_OK;
}

// mrn_split_hl32_t_f32 that also leaves colsum[c] (+)= sum over rows of x[r][c].  partial: mrn_split_hl32_t_colsum_chunks(rows_padded, C) * C
// floats of scratch (unused with one chunk).
extern "C" int mrn_colsum_f32(const float* in, int64_t ld, float* out, float* workspace, int64_t rows, int C, int accumulate, void* stream);

MRN_EXPORT int64_t mrn_split_hl32_t_colsum_chunks(int64_t rows_padded, int C) {
  const long rblocks = rows_padded / 32, tiles_c = (C + 31) / 32;
  long want = 4096 / (tiles_c < 1 ? 1 : tiles_c);
  want = want < 1 ? 1 : (want > 256 ? 256 : want);
  long per = (rblocks + want - 1) / want;
  per = (per + 3) / 4 * 4;                             // (four row blocks per iteration)
  const long chunks = per > 0 ? (rblocks + per - 1) / per : 1;
  return chunks < 1 ? 1 : chunks;
}

MRN_EXPORT int mrn_split_hl32_t_colsum_f32(const float* x, void* out, int64_t rows, int64_t rows_padded, int C, int splits,
                                           const float* scale, float* colsum, int accumulate, float* partial, void* stream) {
  MRN_CHECK_ARG(x && out && colsum && splits >= 1 && C % 4 == 0 && rows_padded >= rows && rows_padded % (32L * splits) == 0,
                "mrn_split_hl32_t_colsum_f32: bad operands (rows=%ld padded=%ld C=%d splits=%d)", (long)rows, (long)rows_padded, C, splits);
  if (rows_padded == 0 || C == 0) return MRN_OK;
  const long rblocks = rows_padded / 32, tiles_c = (C + 31) / 32;
  const long chunks = mrn_split_hl32_t_colsum_chunks(rows_padded, C);
  const long per = (((rblocks + chunks - 1) / chunks) + 3) / 4 * 4;
  MRN_CHECK_ARG(chunks == 1 || partial, "mrn_split_hl32_t_colsum_f32: %ld chunks need the scratch rows", chunks);
  MRN_CHECK_ARG(per * chunks >= rblocks && chunks <= 512, "mrn_split_hl32_t_colsum_f32: bad chunking (%ld x %ld < %ld)", per, chunks, rblocks);
  hipLaunchKernelGGL(split_hl32_t_colsum_kernel, dim3((unsigned)tiles_c, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned char*)out, C, (long)rows, (long)(rows_padded / splits / 32), rblocks, per, scale, partial, colsum, accumulate);
  MRN_LAUNCH_CHECK("split_hl32_t_colsum");
  if (chunks > 1) return mrn_colsum_f32(partial, C, colsum, nullptr, chunks, C, accumulate, stream);     // (<= 512 rows: one pass)
  return MRN_OK;
}

// Transposed im2col for the convolution weight gradient dW[co][tap][ci] = sum_p dy[p][co] * x[pixel(p, tap)][ci]:
//   out[s][tap][ci][rps/32][hi 32 | lo 32], element (tap, ci, p) = scale * x[b][oy*sh + ky - ph][ox*sw + kx - pw][ci]
// (zero outside the image / beyond P = B*Ho*Wo), p = s*rps + r the output pixel.  x NHWC [B][H][W][C], C % 4 == 0.
MRN_EXPORT int mrn_im2col_t_hl32_f32(const float* x, void* out, int B, int H, int W, int C, int kh, int kw, int sh, int sw,
                                     int ph, int pw, int64_t rows_padded, int splits, const float* scale, void* stream) {
  MRN_CHECK_ARG(x && out && splits >= 1 && C % 4 == 0 && rows_padded % (32L * splits) == 0, "mrn_im2col_t_hl32_f32: bad operands");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  MRN_CHECK_ARG(Ho > 0 && Wo > 0 && rows_padded >= (long)B * Ho * Wo, "mrn_im2col_t_hl32_f32: bad geometry");
  if (rows_padded == 0 || C == 0) return MRN_OK;
  Im2colT p;
  p.x = x; p.out = (unsigned char*)out; p.scale = scale;
  p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.kw = kw; p.taps = kh * kw;
  p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.P = (long)B * Ho * Wo; p.rps32 = rows_padded / splits / 32; p.tiles_c = (C + 31) / 32;
  p.ntiles = rows_padded / 32 * p.tiles_c * p.taps;
  long grid = p.ntiles > 262144 ? 262144 : p.ntiles;
  hipLaunchKernelGGL(im2col_t_hl32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
  MRN_LAUNCH_CHECK("im2col_t_hl32");
  return MRN_OK;
}

// x NHWC [B][H][W][C] fp32 -> transposed HL32 matrix [C][lines][128 B] (lines = ceil(B*H*W / 32)) in image-row-major pixel order
// k = (y*B + b)*W + xx, shifted by shift_x pixels along x with zero fill: see transpose_oy_hl32_kernel
MRN_EXPORT int mrn_transpose_oy_hl32_f32(const float* x, void* out, int B, int H, int W, int C, int shift_x, const float* scale,
                                         void* stream) {
  MRN_CHECK_ARG(x && out && C % 4 == 0 && B > 0 && H > 0 && W > 0, "mrn_transpose_oy_hl32_f32: bad operands");
  const long P = (long)B * H * W, lines = (P + 31) / 32, tiles_c = (C + 31) / 32, ntiles = lines * tiles_c;
  long grid = ntiles > 65536 ? 65536 : ntiles;
  hipLaunchKernelGGL(transpose_oy_hl32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned char*)out, B, H, W,
                     C, shift_x, P, lines, tiles_c, ntiles, scale);
  MRN_LAUNCH_CHECK("transpose_oy_hl32");
  return MRN_OK;
}

// the three copies shift_x = -1, 0, +1 of mrn_transpose_oy_hl32_f32 in one pass: out [3][C][lines][128 B]
MRN_EXPORT int mrn_transpose_oy3_hl32_f32(const float* x, void* out, int B, int H, int W, int C, const float* scale, void* stream) {
  MRN_CHECK_ARG(x && out && C % 4 == 0 && B > 0 && H > 0 && W > 0, "mrn_transpose_oy3_hl32_f32: bad operands");
  const long P = (long)B * H * W, lines = (P + 31) / 32, tiles_c = (C + 31) / 32, ntiles = lines * tiles_c;
  long grid = ntiles > 65536 ? 65536 : ntiles;
  hipLaunchKernelGGL(transpose_oy3_hl32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned char*)out, B, H,
                     W, C, P, lines, tiles_c, ntiles, (long)C * lines * 128, scale);
  MRN_LAUNCH_CHECK("transpose_oy3_hl32");
  return MRN_OK;
}

// The two operand passes and the final reduction of the Winograd-domain weight gradient of a 3x3 / stride 1 / pad 1 convolution
// (transpose_oy_wino_hl32_kernel, wino_wgrad_finish_kernel): t NHWC [B][H][W][C] fp32 -> out [6][C][ceil(B*H*ceil(W/4) / 32)][128 B];
// mode 0 = layer input (B^T over 6 columns), mode 1 = output gradient (A over 4 columns).
MRN_EXPORT int mrn_transpose_oy_wino_hl32_f32(const float* t, void* out, int B, int H, int W, int C, int mode, const float* scale,
                                              void* stream) {
  MRN_CHECK_ARG(t && out && C % 4 == 0 && B > 0 && H > 0 && W > 0 && (mode == 0 || mode == 1), "mrn_transpose_oy_wino_hl32_f32: bad operands");
  const int Wq = (W + 3) / 4;
  const long Pq = (long)B * H * Wq, lines = (Pq + 31) / 32, tiles_c = (C + 31) / 32, ntiles = lines * tiles_c;
  long grid = ntiles > 65536 ? 65536 : ntiles;
  if (mode == 0)
    hipLaunchKernelGGL(transpose_oy_wino_hl32_kernel<0>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, t, (unsigned char*)out, B,
                       H, W, Wq, C, Pq, lines, tiles_c, ntiles, scale);
  else
    hipLaunchKernelGGL(transpose_oy_wino_hl32_kernel<1>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, t, (unsigned char*)out, B,
                       H, W, Wq, C, Pq, lines, tiles_c, ntiles, scale);
  MRN_LAUNCH_CHECK("transpose_oy_wino_hl32");
  return MRN_OK;
}

// part [S][18][Cout][Cin] (group (m, ky) = m * 3 + ky of split-K chunk s, from mrn_gemm_x3_windows_hl32) -> dW [Cout][3][3][Cin]
// (oihw_accumulate 0), or ADDED into dW [Cout][Cin][3][3], the parameter's own layout (oihw_accumulate 1)
MRN_EXPORT int mrn_wino_wgrad_finish_f32(const float* part, float* dw, int S, int Cout, int Cin, int oihw_accumulate, void* stream) {
  MRN_CHECK_ARG(part && dw && S >= 1 && Cout >= 1 && Cin >= 4 && Cin % 4 == 0, "mrn_wino_wgrad_finish_f32: bad operands (Cin %% 4 == 0)");
  const long CC4 = (long)Cout * Cin / 4;
  long grid = (CC4 + 255) / 256;
  if (grid > 16384) grid = 16384;
  if (oihw_accumulate)
    hipLaunchKernelGGL(wino_wgrad_finish_kernel<true>, dim3((unsigned)grid, 3), dim3(256), 0, (hipStream_t)stream, part, dw, S, CC4, Cin / 4);
  else
    hipLaunchKernelGGL(wino_wgrad_finish_kernel<false>, dim3((unsigned)grid, 3), dim3(256), 0, (hipStream_t)stream, part, dw, S, CC4, Cin / 4);
  MRN_LAUNCH_CHECK("wino_wgrad_finish");
  return MRN_OK;
}

// Grouped GEMM over K-WINDOWS of two shared HL32 matrices on the conv_x3 kernel: for group g,
//   y[g][m][n] = out_scale * sum over the window's K of A[m][a_off_g + k] * Wm[n][w_off_g + k]
// a_hl: [M][a_pitch lines][128 B], w_hl: [N][w_pitch lines][128 B] (a_bytes / w_bytes their sizes); windows: DEVICE array of G
// records {int64 a_off_bytes, int64 w_off_bytes, int32 n_lines, int32 pad}: where the window starts inside a row of each matrix
// and how many 128-byte lines (32 K each) it spans (>= 1).  y [G][M][N] fp32.  out_scale / x_scale: ONE {s, 1/s} pair per operand
// (w_hl's and a_hl's), shared by all groups.
// Used by the convolution weight gradient: A = dy^T, Wm = one of three x-shifted copies of x^T, one group per (split-K chunk, tap).
MRN_EXPORT int mrn_gemm_x3_windows_hl32(const void* a_hl, int64_t a_bytes, int a_pitch_lines, const void* w_hl, int64_t w_bytes,
                                        int w_pitch_lines, const void* windows, int G, int M, int N, const void* zero_page,
                                        const float* out_scale, const float* x_scale, float* y, int tile_m, int tile_n, int products,
                                        void* stream) {
  MRN_CHECK_ARG(a_hl && w_hl && windows && zero_page && y && G >= 1 && M >= 1 && N >= 1, "mrn_gemm_x3_windows_hl32: bad operands");
  MRN_CHECK_ARG(products == 3 || products == 1, "mrn_gemm_x3_windows_hl32: products must be 3 (split-fp16 x3) or 1 (hi x hi), got %d", products);
  MRN_CHECK_ARG(a_bytes < (1L << 31) && w_bytes < (1L << 31) && (uintptr_t)a_hl % 128 == 0 && (uintptr_t)w_hl % 128 == 0,
                "mrn_gemm_x3_windows_hl32: operand matrices must be 128-byte aligned and below 2 GiB");
  MRN_CHECK_ARG((tile_m == 256 && (tile_n == 256 || tile_n == 128 || tile_n == 64)) || (tile_m == 128 && tile_n == 128) ||
                    ((tile_m == 64 || tile_m == 128) && tile_n == 64 && products == 3),
                "mrn_gemm_x3_windows_hl32: tile must be 256x256, 256x128, 256x64, 128x128 or (x3 products) 128x64 / 64x64");
  ConvX3Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const unsigned char*)a_hl; p.w = (const unsigned char*)w_hl; p.zero = (const unsigned char*)zero_page;
  p.out_scale = nullptr; p.x_scale = x_scale; p.y = y;
  p.win = (const X3Window*)windows; p.a_pitch = a_pitch_lines; p.w_pitch = w_pitch_lines;
  p.x_bytes = (int)a_bytes; p.w_bytes = (int)w_bytes;
  p.Cb = 1; p.taps = 1; p.nk = 1; p.x_group_div = 1;
  p.G = G; p.M = M; p.N = N; p.y_ld = N; p.y_gstride = (long)M * N;
  p.H = 1; p.W = 1; p.Ho = 1; p.Wo = 1; p.kh = 1; p.kw = 1; p.sh = 1; p.sw = 1;
  p.BWo = M;
  magic_div(1u, p.wo_magic, p.wo_shift);
  magic_div((unsigned)p.BWo, p.bw_magic, p.bw_shift);
  magic_div(1u, p.kw_magic, p.kw_shift);
  // both operands carry their own power-of-two prescale: the epilogue multiplies by out_scale[1] * x_scale[1]; ONE pair serves every group
  p.out_scale = out_scale; p.os_stride = 0;
  hipStream_t st = (hipStream_t)stream;
  if (products == 1) {
    if (tile_n == 256) return launch_x3<4, 4, 2, 2, false, 1>(p, st);
    if (tile_m == 256 && tile_n == 64) return launch_x3<8, 1, 1, 2, false, 1>(p, st);
    if (tile_m == 256) return launch_x3<4, 2, 2, 2, false, 1>(p, st);
    return launch_x3<4, 2, 1, 2, false, 1>(p, st);
  }
  if (tile_m == 64) return launch_x3<2, 2, 1, 1>(p, st);
  if (tile_m == 128 && tile_n == 64) return launch_x3<4, 1, 1, 2>(p, st);
  if (tile_n == 256) return launch_x3<4, 4, 2, 2>(p, st);
  if (tile_m == 256 && tile_n == 64) return launch_x3<8, 1, 1, 2>(p, st);
  if (tile_m == 256) return launch_x3<4, 2, 2, 2>(p, st);
  return launch_x3<4, 2, 1, 2>(p, st);
}

// ---- Winograd F(R,3) along W: weight transform + the convolution entry point ----------------------------------------------------
namespace {

// rows of G (F(4,3): interpolation points 0, +-1, +-2, inf; F(2,3): 0, +-1, inf) times the power-of-two row scale whose inverse
// WinoAT folds into A^T, so that every transformed row sum of |.| stays <= 1.5 and ONE per-tensor prescale serves all components
__device__ __forceinline__ double wino_g(int R, int m, int kx) {
  if (R == 2) {
    const double g[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    return g[m][kx];
  }
  const double g[6][3] = {{1, 0, 0},
                          {-1. / 3, -1. / 3, -1. / 3},
                          {-1. / 3, 1. / 3, -1. / 3},
                          {1. / 12, 1. / 6, 1. / 3},
                          {1. / 12, -1. / 6, 1. / 3},
                          {0, 0, 1}};
  return g[m][kx];
}

// w [Cout][3][3][Cin] fp32 (OHWI) -> U [Cout][NC][Cin/32][3 (ky)][hi 32 | lo 32] with U_m = scale * sum_kx G[m][kx] * w[.][ky][kx][.],
// the sum in double, split to hi + lo from the double; one thread = 8 channels of one (cout, component, ky)
__global__ __launch_bounds__(256) void pack_weight_wino_hl32_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                                    int Cout, int Cin, int R, const float* __restrict__ scale, int dense) {
  const double sc = scale ? (double)scale[0] : 1.0;
  const int Cb = Cin >> 5, NC = R + 2;
  const long n8 = (long)Cout * NC * 3 * (Cin >> 3);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % (Cin >> 3));
    long r = i / (Cin >> 3);
    const int ky = (int)(r % 3);
    r /= 3;
    const int m = (int)(r % NC);
    const long o_ = r / NC;
    const float* src = w + ((o_ * 3 + ky) * 3) * Cin + c8 * 8;       // [kx][Cin]
    f16v8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      double u = 0.0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) u += wino_g(R, m, kx) * (double)src[kx * Cin + e];
      u *= sc;
      const _Float16 hh = (_Float16)u;
      const _Float16 ll = (_Float16)(u - (double)hh);
      h[e] = hh; l[e] = ll;
    }
    if (dense) {           // plain fp16 (dense == 2: bfloat16), 64 channels per line: [Cout][NC][Cin/64][3][128 B]
      unsigned char* dst = out + (((o_ * NC + m) * (Cin >> 6) + (c8 >> 3)) * 3 + ky) * 128 + (c8 & 7) * 16;
      if (dense == 2) {
        typedef unsigned short u16v8 __attribute__((ext_vector_type(8)));
        u16v8 b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          double u = 0.0;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) u += wino_g(R, m, kx) * (double)src[kx * Cin + e];
          const unsigned bits = __float_as_uint((float)(u * sc));
          b[e] = (unsigned short)((bits + 0x7fffu + ((bits >> 16) & 1u)) >> 16);
        }
        *reinterpret_cast<u16v8*>(dst) = b;
      } else {
        *reinterpret_cast<f16v8*>(dst) = h;
      }
      continue;
    }
    const int cb = c8 >> 2;
    unsigned char* o = out + (((o_ * NC + m) * Cb + cb) * 3 + ky) * 128 + (c8 & 3) * 16;
    *reinterpret_cast<f16v8*>(o) = h;
    *reinterpret_cast<f16v8*>(o + 64) = l;
  }
}

}  // namespace

// w [Cout][3][3][Cin] fp32 -> Winograd-domain HL32 weight [Cout][R+2][Cin/32][3][128 B] of scale[0] * (G w) (R = 2 or 4)
MRN_EXPORT int mrn_pack_weight_wino_hl32(const float* w_ohwi, void* out, int Cout, int Cin, int R, const float* scale, void* stream) {
  MRN_CHECK_ARG(w_ohwi && out && Cin % 32 == 0 && (R == 2 || R == 4), "mrn_pack_weight_wino_hl32: bad operands (Cin=%d R=%d)", Cin, R);
  const long n8 = (long)Cout * (R + 2) * 3 * (Cin / 8);
  if (n8 == 0) return MRN_OK;
  long grid = (n8 + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(pack_weight_wino_hl32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w_ohwi, (unsigned char*)out,
                     Cout, Cin, R, scale, 0);
  MRN_LAUNCH_CHECK("pack_weight_wino_hl32");
  return MRN_OK;
}

// the same transform as PLAIN fp16 for the reduced-precision mode: w [Cout][3][3][Cin] fp32 -> [Cout][6][Cin/64][3][128 B] of
// scale[0] * (G w), F(4,3), 64 channels per line (mrn_conv2d_x3_wino_d16's weight operand); Cin % 64 == 0
MRN_EXPORT int mrn_pack_weight_wino_d16(const float* w_ohwi, void* out, int Cout, int Cin, const float* scale, int bf16, void* stream) {
  MRN_CHECK_ARG(w_ohwi && out && Cin % 64 == 0, "mrn_pack_weight_wino_d16: bad operands (Cin=%d)", Cin);
  const long n8 = (long)Cout * 6 * 3 * (Cin / 8);
  if (n8 == 0) return MRN_OK;
  long grid = (n8 + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(pack_weight_wino_hl32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w_ohwi, (unsigned char*)out,
                     Cout, Cin, 4, scale, bf16 ? 2 : 1);
  MRN_LAUNCH_CHECK("pack_weight_wino_d16");
  return MRN_OK;
}

// row blocks of the BatchNorm partial statistics of one group: the x3 kernel writes one per 128 GEMM rows; the row-block kernel
// (conv_wino.hip, H % 4 == 0) one per (64 positions, 4 output rows) and clears the rest
static int wino_stats_blocks(int B, int H, int W, int Cout, int R) {
  const int Wq = ceil_div(W, R);
  int blocks = ceil_div((long)B * H * Wq, 128);
  if (mrn_wino_rows_supported(H, R, Cout)) {
    const int rows = ceil_div((long)B * Wq, 64) * (H / 4);
    blocks = blocks > rows ? blocks : rows;
  }
  return blocks;
}

// 1 when mrn_conv2d_x3_wino_hl32 runs this geometry on the row-block kernel (conv_wino.hip), 0 for the x3 kernel's Winograd form
// A/B switch of the same choice for the whole process: -1 = default (row-block kernel unless MRN_WINO_ROWS=0), 0 = always the x3
// kernel's Winograd form, 1 = the row-block kernel wherever it applies.  Statistics buffers must be sized AFTER the call.
MRN_EXPORT int mrn_conv2d_x3_wino_select(int mode) {
  mrn_wino_rows_select(mode);
  return MRN_OK;
}

MRN_EXPORT int64_t mrn_conv2d_x3_wino_rows(int H, int R, int Cout) { return mrn_wino_rows_supported(H, R, Cout) ? 1 : 0; }

MRN_EXPORT int64_t mrn_conv2d_x3_wino_stats_floats(int G, int B, int H, int W, int Cout, int R) {
  return (int64_t)G * wino_stats_blocks(B, H, W, Cout, R) * 2 * Cout;
}

// Grouped 3x3 / stride 1 / pad 1 convolution as 1-D Winograd F(R,3) along W on the x3 kernel (see conv_x3_kernel, WINO):
//   v_hl  [Gx][B][H][ceil(W/R)][R+2][Cin/32][128 B]   transformed activation (mrn_bn_apply_wino_grouped_f32)
//   u_hl  [G][Cout][R+2][Cin/32][3][128 B]            transformed weight (mrn_pack_weight_wino_hl32), out_scale [G][2] its {s, 1/s}
//   y     [G][B][H][W][Cout] fp32 = conv + bias (act 0 / 1), stats [G][mrn_conv2d_x3_wino_stats_floats / G] BatchNorm partials or NULL
//   x_scale {s, 1/s} of an activation operand that was transformed as s * B^T(x) (trained layers), or NULL
static int wino_conv_launch(const void* v_hl, const void* u_hl, const void* zero_page, const float* bias, float* y, float* stats,
                            const float* out_scale, const float* x_scale, int G, int64_t v_group_stride_bytes, int B, int H, int W, int Cin,
                            int Cout, int R, int act, int pool, const void* bn_gamma_ptrs, void* stream, int dense = 0) {
  MRN_CHECK_ARG(v_hl && u_hl && (zero_page || dense) && y && G >= 1 && (R == 2 || R == 4), "mrn_conv2d_x3_wino_hl32: bad operands");
  if (dense && !(mrn_wino_rows_supported(H, R, Cout) && Cin % 64 == 0)) {
    mrn_set_error("mrn_conv2d_x3_wino_d16: the plain-fp16 form runs on the row-block kernel only (H %% 4 == 0, F(4,3), Cin %% 64 == 0; H=%d Cin=%d)", H, Cin);
    return MRN_ERR_UNSUPPORTED;
  }
  if (pool && !(mrn_wino_rows_supported(H, R, Cout) && H % 2 == 0 && W % 2 == 0)) {
    mrn_set_error("mrn_conv2d_x3_wino_pool_hl32: the pooled form needs the row-block kernel (mrn_conv2d_x3_wino_rows) and even H, W (%d x %d)", H, W);
    return MRN_ERR_UNSUPPORTED;
  }
  MRN_CHECK_ARG(Cin % 32 == 0 && ((uintptr_t)v_hl % 128 == 0) && ((uintptr_t)u_hl % 128 == 0) && v_group_stride_bytes % 128 == 0,
                "mrn_conv2d_x3_wino_hl32: HL32 operands must be 128-byte aligned, Cin %% 32 == 0 (Cin=%d)", Cin);
  const int NC = R + 2, Wq = ceil_div(W, R);
  MRN_CHECK_ARG(H > 0 && W > 0 && B >= 0, "mrn_conv2d_x3_wino_hl32: empty output");
  MRN_CHECK_ARG((long)B * H * Wq * NC * Cin * 4 < (1L << 31) && (long)Cout * 3 * NC * Cin * 4 < (1L << 31),
                "mrn_conv2d_x3_wino_hl32: one group's activation / weight must stay below 2 GiB (32-bit buffer offsets)");
  if (mrn_wino_rows_supported(H, R, Cout)) {      // the row-block kernel (conv_wino.hip): 4-row maps and multiples
    if ((long)B * H * Wq == 0) return MRN_OK;
    WinoRowsParams q;
    memset(&q, 0, sizeof(q));
    q.v = (const unsigned char*)v_hl; q.u = (const unsigned char*)u_hl; q.bias = bias; q.out_scale = out_scale; q.x_scale = x_scale;
    q.y = y; q.stats = stats;
    const int eb = dense ? 2 : 4;                 // operand bytes per (component, channel)
    q.v_gstride = v_group_stride_bytes; q.u_gstride = (long)Cout * 3 * NC * Cin * eb; q.y_gstride = (long)B * H * W * Cout;
    q.v_bytes = (int)((long)B * H * Wq * NC * Cin * eb);
    q.G = G; q.B = B; q.H = H; q.W = W; q.Wq = Wq; q.Cb = Cin / (dense ? 64 : 32); q.N = Cout; q.act = act;
    q.dense = dense;
    q.stats_blocks = wino_stats_blocks(B, H, W, Cout, R);
    q.pool = pool; q.gamma = (const long long*)bn_gamma_ptrs;
    return mrn_launch_wino_rows(q, stream);
  }
  ConvX3Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const unsigned char*)v_hl; p.w = (const unsigned char*)u_hl; p.zero = (const unsigned char*)zero_page;
  p.bias = bias; p.out_scale = out_scale; p.os_stride = 2; p.x_scale = x_scale; p.y = y; p.stats = stats;
  p.Cb = NC * (Cin / 32); p.taps = 3; p.nk = p.Cb * 3;
  p.x_gstride = v_group_stride_bytes; p.w_gstride = (long)Cout * p.nk * 128;
  p.x_group_div = 1;
  p.x_bytes = (int)((long)B * H * Wq * NC * Cin * 4);
  p.G = G; p.M = B * H * Wq; p.N = Cout;
  p.y_ld = Cout;
  p.y_gstride = (long)B * H * W * Cout;
  p.H = H; p.W = Wq; p.Ho = H; p.Wo = Wq; p.kh = 3; p.kw = 1; p.sh = 1; p.sw = 1; p.ph = 1; p.pw = 0; p.act = act;
  p.wino_W = W;
  if (p.M == 0) return MRN_OK;
  p.oy_major = (H <= 8) ? 1 : 0;
  if (env_flags().row_major) p.oy_major = 0;
  p.BWo = B * Wq;
  magic_div((unsigned)Wq, p.wo_magic, p.wo_shift);
  magic_div((unsigned)p.BWo, p.bw_magic, p.bw_shift);
  magic_div(1u, p.kw_magic, p.kw_shift);
  if (R == 4) return launch_x3<2, 4, 2, 1, false, 3, 4>(p, (hipStream_t)stream);
  return launch_x3<2, 4, 2, 1, false, 3, 2>(p, (hipStream_t)stream);
}

MRN_EXPORT int mrn_conv2d_x3_wino_hl32(const void* v_hl, const void* u_hl, const void* zero_page, const float* bias, float* y,
                                       float* stats, const float* out_scale, const float* x_scale, int G, int64_t v_group_stride_bytes,
                                       int B, int H, int W, int Cin, int Cout, int R, int act, void* stream) {
  return wino_conv_launch(v_hl, u_hl, zero_page, bias, y, stats, out_scale, x_scale, G, v_group_stride_bytes, B, H, W, Cin, Cout, R, act, 0,
                          nullptr, stream);
}

// The reduced-precision form (ONE fp16 product per term, fp32 accumulate; BASELINE configs 2 / 5): the same F(4,3) row-block kernel on
// PLAIN fp16 operands, 64 channels per 128-byte line -- a third of the MFMAs on half the bytes of the split form:
//   v_d16 [Gx][B][H][ceil(W/4)][6][Cin/64][128 B]  (mrn_bn_apply_wino_grouped_d16_f32 / mrn_maxpool_wino_grouped_d16_f32)
//   u_d16 [G][Cout][6][Cin/64][3][128 B]           (mrn_pack_weight_wino_d16), out_scale [G][2]
// pool != 0: the pooled epilogue of mrn_conv2d_x3_wino_pool_hl32 (bn_gamma_ptrs as there).  H % 4 == 0, Cin % 64 == 0, else
// MRN_ERR_UNSUPPORTED.  Statistics buffer: mrn_conv2d_x3_wino_stats_floats(G, B, H, W, Cout, 4).
MRN_EXPORT int mrn_conv2d_x3_wino_d16(const void* v_d16, const void* u_d16, const float* bias, float* y, float* stats,
                                      const float* out_scale, const float* x_scale, int G, int64_t v_group_stride_bytes, int B, int H,
                                      int W, int Cin, int Cout, int act, int pool, const void* bn_gamma_ptrs, int bf16, void* stream) {
  return wino_conv_launch(v_d16, u_d16, nullptr, bias, y, stats, out_scale, x_scale, G, v_group_stride_bytes, B, H, W, Cin, Cout, 4, act,
                          pool, bn_gamma_ptrs, stream, bf16 ? 2 : 1);
}

// mrn_conv2d_x3_wino_hl32 with the 2x2 / stride-2 max-pool that follows BatchNorm + ReLU taken in the epilogue (row-block kernel only:
// mrn_conv2d_x3_wino_rows(H, R, Cout) != 0, H and W even): y is [G][B][H/2][W/2][Cout] and holds, per window and channel, the maximum
// of the raw output where the BatchNorm weight (bn_gamma_ptrs: device table of G device addresses; NULL: all maxima) is >= 0 and the
// minimum where it is negative; the statistics cover the full map (see mrn_conv3x3_patch_x3_hl32).
MRN_EXPORT int mrn_conv2d_x3_wino_pool_hl32(const void* v_hl, const void* u_hl, const void* zero_page, const float* bias, float* y,
                                            float* stats, const float* out_scale, const float* x_scale, int G,
                                            int64_t v_group_stride_bytes, int B, int H, int W, int Cin, int Cout, int R, int act,
                                            const void* bn_gamma_ptrs, void* stream) {
  return wino_conv_launch(v_hl, u_hl, zero_page, bias, y, stats, out_scale, x_scale, G, v_group_stride_bytes, B, H, W, Cin, Cout, R, act, 1,
                          bn_gamma_ptrs, stream);
}
