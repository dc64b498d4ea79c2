// Split-bf16 implicit-GEMM convolution on the bf16 MFMA pipe: fp32 operands, fp32-class results.
//
// Every fp32 operand x is split into two bf16 values, x = hi + lo (+ 2^-17 relative residue), and the product is
// evaluated as hi*hi + hi*lo + lo*hi with v_mfma_f32_32x32x16_bf16 and fp32 accumulation ("bf16x3").  The dropped
// lo*lo term is 2^-16 relative, so a K=4608 reduction carries ~1e-5 relative error -- measured end to end on the
// reference (TPS + 29-conv ResNet + BiLSTM + decoder): features within 7e-5, logits within 3e-6 of the fp32 path,
// i.e. inside the 1e-4 parity band -- at 3/16 of the cost of the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
// nsplit = 1 keeps only hi*hi (plain bf16 operands, fp32 accumulate; 2e-2 features / 1e-3 logits on the same test).
//
// HALF variant ("fp16x3"): the same three-product scheme with fp16 halves on v_mfma_f32_32x32x16_f16.  fp16 carries 11
// significand bits, so hi + lo represents 22 bits and each product is good to ~2^-22 -- fp32-rounding class -- at the
// same MFMA cost.  fp16's narrow exponent is handled by a per-weight-tensor power-of-two prescale (exact; chosen on the
// device from max|w| so the scaled weights peak near 2^14, which keeps the lo halves in the normal range) whose inverse
// is folded into the epilogue; activations (|x| << 65504 after BatchNorm) are split unscaled.
//
// Activations stay fp32 NHWC in HBM (the split happens while staging a tile into LDS); weights are pre-split once
// per weight version into two bf16 [Cout][K] arrays (mrn_split_weight_bf16).  Same epilogue contract as gemm.hip
// (bias, activation, BatchNorm partial statistics).
//
// Tile 128x128x32, 4 waves (2x2), wave tile 64x64 = 2x2 MFMA tiles; LDS rows are k-contiguous (64 B of bf16) with
// the 16-byte chunk index XOR-swizzled by (row >> 2) & 3, so the 16-lane groups of ds_read_b128 (16 rows, same
// chunk) hit 16 distinct 16-byte slots; double-buffered, 64 KB + tap table -> two workgroups per CU.
#include "common.hpp"
#include <stdlib.h>
#include "gemm.hpp"

namespace {

typedef __bf16 bf16_t;
typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
typedef bf16_t bf16v8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BKB = 32, NT = 256;
constexpr int ROWB = 64;                      // bytes per LDS row (32 bf16)
constexpr int PLANE = BM * ROWB;              // one 128-row plane (hi or lo) = 8192 B
constexpr int STAGE = 4 * PLANE;              // A_hi, A_lo, B_hi, B_lo
constexpr int MAX_TAPS = 2048;

__device__ __forceinline__ unsigned pack2(float a, float b) {   // two fp32 -> packed bf16 pair (RNE), a in the low half
  f32x2 v = {a, b};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<unsigned*>(&r);
}
__device__ __forceinline__ float lo_of(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_of(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

__device__ __forceinline__ unsigned pack2h(float a, float b) {  // two fp32 -> packed fp16 pair (RNE)
  f32x2 v = {a, b};
  f16x2_t r = __builtin_convertvector(v, f16x2_t);
  return *reinterpret_cast<unsigned*>(&r);
}
__device__ __forceinline__ f32x2 unpack2h(unsigned pk) {
  f16x2_t r = *reinterpret_cast<f16x2_t*>(&pk);
  return __builtin_convertvector(r, f32x2);
}

// split 4 floats -> (hi packed x2, lo packed x2)
template <bool HALF>
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
  if (HALF) {
    hi[0] = pack2h(v[0], v[1]);
    hi[1] = pack2h(v[2], v[3]);
    const f32x2 h0 = unpack2h(hi[0]), h1 = unpack2h(hi[1]);
    lo[0] = pack2h(v[0] - h0[0], v[1] - h0[1]);
    lo[1] = pack2h(v[2] - h1[0], v[3] - h1[1]);
  } else {
    hi[0] = pack2(v[0], v[1]);
    hi[1] = pack2(v[2], v[3]);
    lo[0] = pack2(v[0] - lo_of(hi[0]), v[1] - hi_of(hi[0]));
    lo[1] = pack2(v[2] - lo_of(hi[1]), v[3] - hi_of(hi[1]));
  }
}

template <bool HALF>
__device__ __forceinline__ f32x16 mma16(const u32x4 a, const u32x4 b, const f32x16 c) {
  if (HALF) return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16v8*>(&a), *reinterpret_cast<const f16v8*>(&b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16v8*>(&a), *reinterpret_cast<const bf16v8*>(&b), c, 0, 0, 0);
}

// byte offset of 16-byte chunk c (0..3) of LDS row `row`
__device__ __forceinline__ int swz(int row, int c) { return row * ROWB + ((c ^ ((row >> 2) & 3)) << 4); }

struct ConvRowB {
  long base;
  int iy0, ix0;
  bool ok;
};

__device__ __forceinline__ int pack_tap(int k, int Cin, int kw) {
  const int tap = k / Cin;
  const int ci = k - tap * Cin;
  const int ky = tap / kw;
  const int kx = tap - ky * kw;
  return (ky << 26) | (kx << 20) | ci;
}

template <int NSPLIT, bool HALF>
__global__ __launch_bounds__(NT) void conv_bf16_kernel(const GemmParams p, const unsigned short* __restrict__ w_hi,
                                                       const unsigned short* __restrict__ w_lo,
                                                       const float* __restrict__ out_scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 stages + tap table
  int* const taps = reinterpret_cast<int*>(lds + 2 * STAGE);

  const int tilesN = (p.N + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tilesN, tile_n = lid - tile_m * tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- per-thread staging coordinates -----------------------------------------------------------
  // A: 128 rows x 8 float4 per row; thread -> kq = t & 7, rows (t >> 3) + 32 p
  const int akq = t & 7;
  ConvRowB crow[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + (t >> 3) + 32 * q;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int hw = p.Ho * p.Wo;
    const int b = mm / hw;
    const int rem = mm - b * hw;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    crow[q].base = (long)b * p.H * p.Wd * p.Cin;
    crow[q].iy0 = oy * p.sh - p.ph;
    crow[q].ix0 = ox * p.sw - p.pw;
    crow[q].ok = ok;
  }
  // B: 128 rows x 4 chunks of 8 bf16; thread -> chunk = t & 3, rows (t >> 2) + 64 p
  const int bch = t & 3;
  for (int i = t; i < p.K / 4; i += NT) taps[i] = pack_tap(i * 4, p.Cin, p.kw);
  __syncthreads();

  f32x4 ra[4];
  u32x4 rbh[2], rbl[2];

  auto load_tile = [&](int k0) {
    const int info = taps[(k0 >> 2) + akq];
    const int ky = info >> 26, kx = (info >> 20) & 63, ci = info & 0xfffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int iy = crow[q].iy0 + ky, ix = crow[q].ix0 + kx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (crow[q].ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd)
        v = *reinterpret_cast<const f32x4*>(p.A + crow[q].base + ((long)iy * p.Wd + ix) * p.Cin + ci);
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = n0 + (t >> 2) + 64 * q;
      const u32x4 z = {0u, 0u, 0u, 0u};
      rbh[q] = z;
      rbl[q] = z;
      if (n < p.N) {
        const long off = (long)n * p.K + k0 + bch * 8;
        rbh[q] = *reinterpret_cast<const u32x4*>(w_hi + off);
        if (NSPLIT > 1) rbl[q] = *reinterpret_cast<const u32x4*>(w_lo + off);
      }
    }
  };
  auto store_tile = [&](unsigned char* st) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x2 hi, lo;
      split4<HALF>(ra[q], hi, lo);
      const int off = swz((t >> 3) + 32 * q, akq >> 1) + (akq & 1) * 8;
      *reinterpret_cast<u32x2*>(st + off) = hi;
      if (NSPLIT > 1) *reinterpret_cast<u32x2*>(st + PLANE + off) = lo;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int off = swz((t >> 2) + 64 * q, bch);
      *reinterpret_cast<u32x4*>(st + 2 * PLANE + off) = rbh[q];
      if (NSPLIT > 1) *reinterpret_cast<u32x4*>(st + 3 * PLANE + off) = rbl[q];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K / BKB;
  load_tile(0);
  store_tile(lds);
  __syncthreads();

  // fragment reads: row = lane & 31 (+32 i), 8 bf16 (16 B) = chunk (lane >> 5) + 2 ks; the swizzle key depends
  // only on (lane & 31) because tile rows start at multiples of 32
  const int ra_ = wm * 64 + (lane & 31), rb_ = wn * 64 + (lane & 31);

  for (int kt = 0; kt < nk; ++kt) {
    unsigned char* cur = lds + (kt & 1) * STAGE;
    if (kt + 1 < nk) load_tile((kt + 1) * BKB);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = swz(ra_ + i * 32, (lane >> 5) + 2 * ks);
        ah[i] = *reinterpret_cast<const u32x4*>(cur + o);
        if (NSPLIT > 1) al[i] = *reinterpret_cast<const u32x4*>(cur + PLANE + o);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int o = swz(rb_ + j * 32, (lane >> 5) + 2 * ks);
        bh[j] = *reinterpret_cast<const u32x4*>(cur + 2 * PLANE + o);
        if (NSPLIT > 1) bl[j] = *reinterpret_cast<const u32x4*>(cur + 3 * PLANE + o);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (NSPLIT > 1) {
            acc[i][j] = mma16<HALF>(al[i], bh[j], acc[i][j]);
            acc[i][j] = mma16<HALF>(ah[i], bl[j], acc[i][j]);
          }
          acc[i][j] = mma16<HALF>(ah[i], bh[j], acc[i][j]);
        }
    }
    if (kt + 1 < nk) store_tile(lds + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }

  // ---- epilogue (same contract as gemm_f32_kernel) ---------------------------------------------
  float csum[2], csq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    const bool nok = n < p.N;
    const float bn = (p.bias && nok) ? p.bias[n] : 0.f;
    const float osc = out_scale ? out_scale[1] : 1.f;      // inverse of the weight prescale (exact power of two)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M && nok) {
          float v = acc[i][j][e] * osc + bn;
          csum[j] += v;
          csq[j] += v * v;
          if (p.act == 1) v = fmaxf(v, 0.f);
          p.C[(long)m * p.sCm + n] = v;
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);   // [2][2][BN]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float s = csum[j] + __shfl_xor(csum[j], 32);
      const float q = csq[j] + __shfl_xor(csq[j], 32);
      if (lane < 32) {
        const int c = wn * 64 + j * 32 + lane;
        red[(wm * 2 + 0) * BN + c] = s;
        red[(wm * 2 + 1) * BN + c] = q;
      }
    }
    __syncthreads();
    if (t < BN) {
      const int n = n0 + t;
      if (n < p.N) {
        p.stats[((long)tile_m * 2 + 0) * p.N + n] = red[0 * BN + t] + red[2 * BN + t];
        p.stats[((long)tile_m * 2 + 1) * p.N + n] = red[1 * BN + t] + red[3 * BN + t];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Pre-split variant: the activation arrives as two bf16 NHWC planes (hi, lo; mrn_split_weight_bf16 on the fp32 tensor)
// and BOTH operands are staged with direct-to-LDS DMA (global_load_lds_dwordx4): no staging registers, no VALU
// conversion, no ds_write.  One DMA instruction moves 16 rows x 64 B = 1 KiB per wave; the LDS image is lane-linear,
// so the XOR swizzle is applied to the SOURCE chunk a lane fetches (chunk_global = chunk_lds ^ key(row)) and the
// fragment reads use the same involution (swz()).  Out-of-image taps / rows fetch from a 64-byte zero page.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

template <int NSPLIT, bool HALF>
__global__ __launch_bounds__(NT) void conv_bf16_dma_kernel(const GemmParams p, const unsigned short* __restrict__ x_hi,
                                                           const unsigned short* __restrict__ x_lo,
                                                           const unsigned short* __restrict__ w_hi,
                                                           const unsigned short* __restrict__ w_lo,
                                                           const unsigned short* __restrict__ zero_page,
                                                           const float* __restrict__ out_scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int* const taps = reinterpret_cast<int*>(lds + 2 * STAGE);

  const int tilesN = (p.N + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tilesN, tile_n = lid - tile_m * tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // DMA geometry: wave w, instruction i (0,1) covers rows (i*4 + w)*16 .. +15 of a 128-row plane;
  // lane -> row = +lane/4, LDS chunk = lane & 3, global chunk = LDS chunk ^ ((row >> 2) & 3)
  const int lrow = lane >> 2, lch = lane & 3;
  long abase[2];      // element offset of the image of A row (i), -1 if the row is beyond M
  int aiy0[2], aix0[2], gch[2];
  long brow[2];       // element offset of weight row, -1 if beyond N
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 4 + wave) * 16 + lrow;
    gch[i] = lch ^ ((row >> 2) & 3);
    const int m = m0 + row;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int hw = p.Ho * p.Wo;
    const int b = mm / hw;
    const int rem = mm - b * hw;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    abase[i] = ok ? (long)b * p.H * p.Wd * p.Cin : -1;
    aiy0[i] = oy * p.sh - p.ph;
    aix0[i] = ox * p.sw - p.pw;
    const int n = n0 + row;
    brow[i] = n < p.N ? (long)n * p.K : -1;
  }
  for (int i = t; i < p.K / 4; i += NT) taps[i] = pack_tap(i * 4, p.Cin, p.kw);
  __syncthreads();

  auto issue_tile = [&](int k0, unsigned char* st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned char* dst = st + (i * 4 + wave) * 16 * ROWB;            // wave-uniform; lane l lands at +16 l
      const int k = k0 + gch[i] * 8;
      const int info = taps[k >> 2];
      const int ky = info >> 26, kx = (info >> 20) & 63, ci = info & 0xfffff;
      const int iy = aiy0[i] + ky, ix = aix0[i] + kx;
      const bool ok = abase[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
      const long off = ok ? abase[i] + ((long)iy * p.Wd + ix) * p.Cin + ci : 0;
      dma16(ok ? (const void*)(x_hi + off) : (const void*)zero_page, dst);
      if (NSPLIT > 1) dma16(ok ? (const void*)(x_lo + off) : (const void*)zero_page, dst + PLANE);
      const bool bok = brow[i] >= 0;
      const long boff = bok ? brow[i] + k : 0;
      dma16(bok ? (const void*)(w_hi + boff) : (const void*)zero_page, dst + 2 * PLANE);
      if (NSPLIT > 1) dma16(bok ? (const void*)(w_lo + boff) : (const void*)zero_page, dst + 3 * PLANE);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K / BKB;
  issue_tile(0, lds);
  __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // vmcnt(0), spelled out: hipcc may drop the DMA wait from __syncthreads()
  __syncthreads();

  const int ra_ = wm * 64 + (lane & 31), rb_ = wn * 64 + (lane & 31);
  for (int kt = 0; kt < nk; ++kt) {
    unsigned char* cur = lds + (kt & 1) * STAGE;
    if (kt + 1 < nk) issue_tile((kt + 1) * BKB, lds + ((kt + 1) & 1) * STAGE);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = swz(ra_ + i * 32, (lane >> 5) + 2 * ks);
        ah[i] = *reinterpret_cast<const u32x4*>(cur + o);
        if (NSPLIT > 1) al[i] = *reinterpret_cast<const u32x4*>(cur + PLANE + o);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int o = swz(rb_ + j * 32, (lane >> 5) + 2 * ks);
        bh[j] = *reinterpret_cast<const u32x4*>(cur + 2 * PLANE + o);
        if (NSPLIT > 1) bl[j] = *reinterpret_cast<const u32x4*>(cur + 3 * PLANE + o);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (NSPLIT > 1) {
            acc[i][j] = mma16<HALF>(al[i], bh[j], acc[i][j]);
            acc[i][j] = mma16<HALF>(ah[i], bl[j], acc[i][j]);
          }
          acc[i][j] = mma16<HALF>(ah[i], bh[j], acc[i][j]);
        }
    }
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // own DMAs of the next tile have landed (vmcnt(0))
    __syncthreads();
  }

  float csum[2], csq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    const bool nok = n < p.N;
    const float bn = (p.bias && nok) ? p.bias[n] : 0.f;
    const float osc = out_scale ? out_scale[1] : 1.f;      // inverse of the weight prescale (exact power of two)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M && nok) {
          float v = acc[i][j][e] * osc + bn;
          csum[j] += v;
          csq[j] += v * v;
          if (p.act == 1) v = fmaxf(v, 0.f);
          p.C[(long)m * p.sCm + n] = v;
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float s = csum[j] + __shfl_xor(csum[j], 32);
      const float q = csq[j] + __shfl_xor(csq[j], 32);
      if (lane < 32) {
        const int c = wn * 64 + j * 32 + lane;
        red[(wm * 2 + 0) * BN + c] = s;
        red[(wm * 2 + 1) * BN + c] = q;
      }
    }
    __syncthreads();
    if (t < BN) {
      const int n = n0 + t;
      if (n < p.N) {
        p.stats[((long)tile_m * 2 + 0) * p.N + n] = red[0 * BN + t] + red[2 * BN + t];
        p.stats[((long)tile_m * 2 + 1) * p.N + n] = red[1 * BN + t] + red[3 * BN + t];
      }
    }
  }
}

// fp32 [rows] -> bf16 hi / lo planes
__global__ void split_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                  long n, int half, const float* __restrict__ scale) {
  const float sc = scale ? scale[0] : 1.f;
  for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 2; i < n; i += (long)gridDim.x * blockDim.x * 2) {
    const float a = w[i] * sc, b = (i + 1 < n) ? w[i + 1] * sc : 0.f;
    unsigned h, l;
    if (half) {
      h = pack2h(a, b);
      const f32x2 hf = unpack2h(h);
      l = pack2h(a - hf[0], b - hf[1]);
    } else {
      h = pack2(a, b);
      l = pack2(a - lo_of(h), b - hi_of(h));
    }
    hi[i] = (unsigned short)(h & 0xffff);
    lo[i] = (unsigned short)(l & 0xffff);
    if (i + 1 < n) {
      hi[i + 1] = (unsigned short)(h >> 16);
      lo[i + 1] = (unsigned short)(l >> 16);
    }
  }
}

// scale[0] = 2^floor(log2(target / max|w|)), scale[1] = 1 / scale[0].  Three tiny launches, no workspace: scale[1] is
// cleared, receives max|w| through atomicMax on its bit pattern (non-negative floats order like unsigned ints), and the
// finalise kernel turns it into the pair.
// ws: two words the caller zeroed ONCE ([0] = running max|w| as uint bits, [1] = arrival ticket), reusable by every later call on the
// same stream: every block folds its maximum into ws[0] and takes a ticket; the LAST block to arrive turns the maximum into the
// power-of-two pair {s, 1/s} and puts the two words back to zero (one launch instead of clear + max + finalise).
__global__ __launch_bounds__(256) void pow2_scale_kernel(const float* __restrict__ w, long n, float target, float* __restrict__ scale,
                                                         unsigned* __restrict__ ws) {
  __shared__ float scratch[4];
  float m = 0.f;
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * 256;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {          // four independent 16-byte loads in flight per lane
    const f32x4 a = reinterpret_cast<const f32x4*>(w)[i], b = reinterpret_cast<const f32x4*>(w)[i + stride];
    const f32x4 c = reinterpret_cast<const f32x4*>(w)[i + 2 * stride], d = reinterpret_cast<const f32x4*>(w)[i + 3 * stride];
#pragma unroll
    for (int e = 0; e < 4; ++e) m = fmaxf(fmaxf(m, fmaxf(fabsf(a[e]), fabsf(b[e]))), fmaxf(fabsf(c[e]), fabsf(d[e])));
  }
  for (; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(w)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0)
    for (long j = (n4 << 2) + threadIdx.x; j < n; j += 256) m = fmaxf(m, fabsf(w[j]));
  m = block_max<256>(m, scratch);
  if (threadIdx.x == 0) {
    if (!(m <= 0.f)) atomicMax(ws, __float_as_uint(m));                // NaN / inf pass through
    __threadfence();
    if (atomicAdd(ws + 1, 1u) == gridDim.x - 1) {                      // last arrival: every block's maximum is in
      const float mm = __uint_as_float(atomicExch(ws, 0u));
      atomicExch(ws + 1, 0u);
      float sc = 1.f;
      if (mm > 0.f && isfinite(mm)) sc = exp2f(floorf(log2f(target / mm)));
      scale[0] = sc;
      scale[1] = 1.f / sc;
    }
  }
}

// the three-launch form (clear, maximum, finalise): the default; A/B partner: the single launch above (MRN_POW2_LAUNCHES=1)
__global__ void pow2_clear_kernel(unsigned* __restrict__ ws) { ws[0] = 0u; }

__global__ __launch_bounds__(256) void pow2_amax_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ ws) {
  __shared__ float scratch[4];
  float m = 0.f;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(w)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
  m = block_max<256>(m, scratch);
  if (threadIdx.x == 0 && !(m <= 0.f)) atomicMax(ws, __float_as_uint(m));
}

__global__ void pow2_finalize_kernel(float target, float* __restrict__ scale, unsigned* __restrict__ ws) {
  const float m = __uint_as_float(ws[0]);
  ws[0] = 0u;
  float s = 1.f;
  if (m > 0.f && isfinite(m)) s = exp2f(floorf(log2f(target / m)));
  scale[0] = s;
  scale[1] = 1.f / s;
}

// the producers that fold max|.| into their own pass spread their per-block atomics over 64 slots: one wave folds and clears them
__global__ __launch_bounds__(64) void pow2_finalize64_kernel(float target, float* __restrict__ scale, unsigned* __restrict__ ws) {
  float m = __uint_as_float(ws[threadIdx.x]);
  ws[threadIdx.x] = 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (threadIdx.x == 0) {
    float s = 1.f;
    if (m > 0.f && isfinite(m)) s = exp2f(floorf(log2f(target / m)));
    scale[0] = s;
    scale[1] = 1.f / s;
  }
}

}  // namespace

// scale[0] = largest power of two with scale*max|w| <= target, scale[1] = its inverse (both on the device); workspace: two 32-bit
// words zeroed once by the caller and then reusable by every call issued on the same stream.  One launch, no host sync.
MRN_EXPORT int mrn_pow2_scale_f32(const float* w, int64_t n, float target, float* scale, void* workspace, void* stream) {
  MRN_CHECK_ARG(w && scale && workspace && target > 0.f, "mrn_pow2_scale_f32: bad operands");
  MRN_CHECK_ARG((uintptr_t)w % 16 == 0, "mrn_pow2_scale_f32: operand must be 16-byte aligned");
  // default: clear + maximum + finalise.  The single self-resetting launch (MRN_POW2_LAUNCHES=1) measured no better on the same box
  // (CRNN x 3 loop B 13.28-13.35 vs 13.04-13.26 ms/step, SVTR x 6 30.6 vs 30.1-30.3, TRBA loop A equal): its per-block ticket
  // serialises on one address, the two extra launches hide behind neighbouring kernels.
  static const bool three = !(getenv("MRN_POW2_LAUNCHES") && atoi(getenv("MRN_POW2_LAUNCHES")) == 1);
  if (three) {
    long g3 = (n / 4 + 255) / 256;
    g3 = g3 < 1 ? 1 : (g3 > 1024 ? 1024 : g3);
    hipLaunchKernelGGL(pow2_clear_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned*)workspace);
    hipLaunchKernelGGL(pow2_amax_kernel, dim3((unsigned)g3), dim3(256), 0, (hipStream_t)stream, w, (long)n, (unsigned*)workspace);
    hipLaunchKernelGGL(pow2_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, target, scale, (unsigned*)workspace);
    MRN_LAUNCH_CHECK("pow2_scale");
    return MRN_OK;
  }
  long grid = (n / 16 + 255) / 256;                      // (each lane keeps four float4 loads in flight)
  grid = grid < 1 ? 1 : (grid > 512 ? 512 : grid);
  hipLaunchKernelGGL(pow2_scale_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w, (long)n, target, scale,
                     (unsigned*)workspace);
  MRN_LAUNCH_CHECK("pow2_scale");
  return MRN_OK;
}

// second half of mrn_pow2_scale_f32 for producers that folded max|.| into their own pass (mrn_scale_shift_act_f32 / mrn_bn_bwd_apply_f32
// with amax_ws = a 64-word workspace zeroed once: per-block maxima land in slot blockIdx % 64): scale = {s, 1/s} from the maximum over
// the 64 slots, which are put back to zero
MRN_EXPORT int mrn_pow2_finalize_f32(float target, float* scale, void* workspace, void* stream) {
  MRN_CHECK_ARG(scale && workspace && target > 0.f, "mrn_pow2_finalize_f32: bad operands");
  hipLaunchKernelGGL(pow2_finalize64_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, target, scale, (unsigned*)workspace);
  MRN_LAUNCH_CHECK("pow2_finalize");
  return MRN_OK;
}

MRN_EXPORT int mrn_split_weight_bf16(const float* w, void* hi, void* lo, int64_t n, int half, const float* scale,
                                     void* stream) {
  MRN_CHECK_ARG(w && hi && lo, "mrn_split_weight_bf16: null operand");
  if (n == 0) return MRN_OK;
  long grid = (n / 2 + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)hi,
                     (unsigned short*)lo, (long)n, half, scale);
  MRN_LAUNCH_CHECK("split_weight_bf16");
  return MRN_OK;
}

// Same contract as mrn_conv2d_nhwc_f32 with the weight given as two bf16 [Cout][kh*kw*Cin] planes (hi, lo).
// nsplit 3: hi*hi + hi*lo + lo*hi (fp32-class accuracy); nsplit 1: hi*hi only (plain bf16 operands).
// Requires Cin % 4 == 0 and (kh*kw*Cin) % 32 == 0.
MRN_EXPORT int mrn_conv2d_nhwc_bf16split(const float* x, const void* w_hi, const void* w_lo, const float* bias, float* y,
                                         float* stats, int B, int H, int Wd, int Cin, int Cout, int kh, int kw, int sh,
                                         int sw, int ph, int pw, int act, int nsplit, int half, const float* out_scale,
                                         void* stream) {
  MRN_CHECK_ARG(x && w_hi && y && (nsplit == 1 || (nsplit == 3 && w_lo)), "mrn_conv2d_nhwc_bf16split: bad operands");
  const int K = kh * kw * Cin;
  MRN_CHECK_ARG(Cin % 4 == 0 && K % 32 == 0 && K <= MAX_TAPS * 4, "mrn_conv2d_nhwc_bf16split: unsupported K=%d Cin=%d", K, Cin);
  MRN_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)w_hi % 16 == 0) && ((uintptr_t)w_lo % 16 == 0),
                "mrn_conv2d_nhwc_bf16split: operands must be 16-byte aligned");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (Wd + 2 * pw - kw) / sw + 1;
  MRN_CHECK_ARG(Ho > 0 && Wo > 0 && B >= 0, "mrn_conv2d_nhwc_bf16split: empty output");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.bias = bias; p.C = y; p.stats = stats;
  p.M = B * Ho * Wo; p.N = Cout; p.K = K; p.batch = 1;
  p.sCm = Cout; p.sCn = 1;
  p.H = H; p.Wd = Wd; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo;
  p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.act = act; p.alpha = 1.f;
  if (p.M == 0) return MRN_OK;
  const size_t ldsz = 2 * STAGE + (size_t)((K / 4 + 3) / 4 * 4) * sizeof(int);
  const int ldmax = 2 * STAGE + MAX_TAPS * (int)sizeof(int);
  const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)conv_bf16_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax);
    hipFuncSetAttribute((const void*)conv_bf16_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax);
    hipFuncSetAttribute((const void*)conv_bf16_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax);
    hipFuncSetAttribute((const void*)conv_bf16_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax);
    attr_done = true;
  }
  const unsigned short* wh = (const unsigned short*)w_hi;
  const unsigned short* wl = (const unsigned short*)w_lo;
  hipStream_t st = (hipStream_t)stream;
  if (nsplit == 3 && !half) hipLaunchKernelGGL((conv_bf16_kernel<3, false>), dim3(tiles), dim3(NT), ldsz, st, p, wh, wl, out_scale);
  else if (nsplit == 3) hipLaunchKernelGGL((conv_bf16_kernel<3, true>), dim3(tiles), dim3(NT), ldsz, st, p, wh, wl, out_scale);
  else if (!half) hipLaunchKernelGGL((conv_bf16_kernel<1, false>), dim3(tiles), dim3(NT), ldsz, st, p, wh, wl, out_scale);
  else hipLaunchKernelGGL((conv_bf16_kernel<1, true>), dim3(tiles), dim3(NT), ldsz, st, p, wh, wl, out_scale);
  MRN_LAUNCH_CHECK("conv_bf16split");
  return MRN_OK;
}

// Pre-split activation variant: x_hi / x_lo are the bf16 planes of the NHWC activation (same element order as the fp32
// tensor); zero_page: >= 64 bytes of device zeros.  Requires Cin % 8 == 0 (a 16-byte chunk never straddles a tap).
MRN_EXPORT int mrn_conv2d_nhwc_bf16split_dma(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo,
                                             const void* zero_page, const float* bias, float* y, float* stats, int B, int H,
                                             int Wd, int Cin, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int act,
                                             int nsplit, int half, const float* out_scale, void* stream) {
  MRN_CHECK_ARG(x_hi && w_hi && y && zero_page && (nsplit == 1 || (nsplit == 3 && w_lo && x_lo)), "mrn_conv2d_nhwc_bf16split_dma: bad operands");
  const int K = kh * kw * Cin;
  MRN_CHECK_ARG(Cin % 8 == 0 && K % 32 == 0 && K <= MAX_TAPS * 4, "mrn_conv2d_nhwc_bf16split_dma: unsupported K=%d Cin=%d", K, Cin);
  MRN_CHECK_ARG(((uintptr_t)x_hi % 16 == 0) && ((uintptr_t)x_lo % 16 == 0) && ((uintptr_t)w_hi % 16 == 0) &&
                    ((uintptr_t)w_lo % 16 == 0) && ((uintptr_t)zero_page % 16 == 0), "mrn_conv2d_nhwc_bf16split_dma: operands must be 16-byte aligned");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (Wd + 2 * pw - kw) / sw + 1;
  MRN_CHECK_ARG(Ho > 0 && Wo > 0 && B >= 0, "mrn_conv2d_nhwc_bf16split_dma: empty output");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.bias = bias; p.C = y; p.stats = stats;
  p.M = B * Ho * Wo; p.N = Cout; p.K = K; p.batch = 1;
  p.sCm = Cout; p.sCn = 1;
  p.H = H; p.Wd = Wd; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo;
  p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.act = act; p.alpha = 1.f;
  if (p.M == 0) return MRN_OK;
  const size_t ldsz = 2 * STAGE + (size_t)((K / 4 + 3) / 4 * 4) * sizeof(int);
  const int ldmax = 2 * STAGE + MAX_TAPS * (int)sizeof(int);
  const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)conv_bf16_dma_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax);
    hipFuncSetAttribute((const void*)conv_bf16_dma_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax);
    hipFuncSetAttribute((const void*)conv_bf16_dma_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax);
    hipFuncSetAttribute((const void*)conv_bf16_dma_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax);
    attr_done = true;
  }
  const unsigned short *xh = (const unsigned short*)x_hi, *xl = (const unsigned short*)x_lo;
  const unsigned short *wh = (const unsigned short*)w_hi, *wl = (const unsigned short*)w_lo, *zp = (const unsigned short*)zero_page;
  hipStream_t st = (hipStream_t)stream;
  if (nsplit == 3 && !half) hipLaunchKernelGGL((conv_bf16_dma_kernel<3, false>), dim3(tiles), dim3(NT), ldsz, st, p, xh, xl, wh, wl, zp, out_scale);
  else if (nsplit == 3) hipLaunchKernelGGL((conv_bf16_dma_kernel<3, true>), dim3(tiles), dim3(NT), ldsz, st, p, xh, xl, wh, wl, zp, out_scale);
  else if (!half) hipLaunchKernelGGL((conv_bf16_dma_kernel<1, false>), dim3(tiles), dim3(NT), ldsz, st, p, xh, xl, wh, wl, zp, out_scale);
  else hipLaunchKernelGGL((conv_bf16_dma_kernel<1, true>), dim3(tiles), dim3(NT), ldsz, st, p, xh, xl, wh, wl, zp, out_scale);
  MRN_LAUNCH_CHECK("conv_bf16split_dma");
  return MRN_OK;
}
