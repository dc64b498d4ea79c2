// Split-bf16 implicit-GEMM convolution on the bf16 MFMA pipe: fp32 operands, fp32-class results.
//
// Every fp32 operand x is split into two bf16 values, x = hi + lo (+ 2^-17 relative residue), and the product is
// evaluated as hi*hi + hi*lo + lo*hi with v_mfma_f32_32x32x16_bf16 and fp32 accumulation ("bf16x3").  The dropped
// lo*lo term is 2^-16 relative, so a K=4608 reduction carries ~1e-5 relative error -- measured end to end on the
// reference (TPS + 29-conv ResNet + BiLSTM + decoder): features within 7e-5, logits within 3e-6 of the fp32 path,
// i.e. inside the 1e-4 parity band -- at 3/16 of the cost of the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
// nsplit = 1 keeps only hi*hi (plain bf16 operands, fp32 accumulate; 2e-2 features / 1e-3 logits on the same test).
//
// Activations stay fp32 NHWC in HBM (the split happens while staging a tile into LDS); weights are pre-split once
// per weight version into two bf16 [Cout][K] arrays (mrn_split_weight_bf16).  Same epilogue contract as gemm.hip
// (bias, activation, BatchNorm partial statistics).
//
// Tile 128x128x32, 4 waves (2x2), wave tile 64x64 = 2x2 MFMA tiles; LDS rows are k-contiguous (64 B of bf16) with
// the 16-byte chunk index XOR-swizzled by (row >> 2) & 3, so the 16-lane groups of ds_read_b128 (16 rows, same
// chunk) hit 16 distinct 16-byte slots; double-buffered, 64 KB + tap table -> two workgroups per CU.
#include "common.hpp"
#include "gemm.hpp"

namespace {

typedef __bf16 bf16_t;
typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
typedef bf16_t bf16v8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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

// split 4 floats -> (hi packed x2, lo packed x2)
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
  hi[0] = pack2(v[0], v[1]);
  hi[1] = pack2(v[2], v[3]);
  lo[0] = pack2(v[0] - lo_of(hi[0]), v[1] - hi_of(hi[0]));
  lo[1] = pack2(v[2] - lo_of(hi[1]), v[3] - hi_of(hi[1]));
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

template <int NSPLIT>
__global__ __launch_bounds__(NT) void conv_bf16_kernel(const GemmParams p, const unsigned short* __restrict__ w_hi,
                                                       const unsigned short* __restrict__ w_lo) {
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
      split4(ra[q], hi, lo);
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
      bf16v8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = swz(ra_ + i * 32, (lane >> 5) + 2 * ks);
        ah[i] = *reinterpret_cast<const bf16v8*>(cur + o);
        if (NSPLIT > 1) al[i] = *reinterpret_cast<const bf16v8*>(cur + PLANE + o);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int o = swz(rb_ + j * 32, (lane >> 5) + 2 * ks);
        bh[j] = *reinterpret_cast<const bf16v8*>(cur + 2 * PLANE + o);
        if (NSPLIT > 1) bl[j] = *reinterpret_cast<const bf16v8*>(cur + 3 * PLANE + o);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (NSPLIT > 1) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
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
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M && nok) {
          float v = acc[i][j][e] + bn;
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

// fp32 [rows] -> bf16 hi / lo planes
__global__ void split_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                  long n) {
  for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 2; i < n; i += (long)gridDim.x * blockDim.x * 2) {
    const float a = w[i], b = (i + 1 < n) ? w[i + 1] : 0.f;
    const unsigned h = pack2(a, b);
    const unsigned l = pack2(a - lo_of(h), b - hi_of(h));
    hi[i] = (unsigned short)(h & 0xffff);
    lo[i] = (unsigned short)(l & 0xffff);
    if (i + 1 < n) {
      hi[i + 1] = (unsigned short)(h >> 16);
      lo[i + 1] = (unsigned short)(l >> 16);
    }
  }
}

}  // namespace

MRN_EXPORT int mrn_split_weight_bf16(const float* w, void* hi, void* lo, int64_t n, void* stream) {
  MRN_CHECK_ARG(w && hi && lo, "mrn_split_weight_bf16: null operand");
  if (n == 0) return MRN_OK;
  long grid = (n / 2 + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)hi,
                     (unsigned short*)lo, (long)n);
  MRN_LAUNCH_CHECK("split_weight_bf16");
  return MRN_OK;
}

// Same contract as mrn_conv2d_nhwc_f32 with the weight given as two bf16 [Cout][kh*kw*Cin] planes (hi, lo).
// nsplit 3: hi*hi + hi*lo + lo*hi (fp32-class accuracy); nsplit 1: hi*hi only (plain bf16 operands).
// Requires Cin % 4 == 0 and (kh*kw*Cin) % 32 == 0.
MRN_EXPORT int mrn_conv2d_nhwc_bf16split(const float* x, const void* w_hi, const void* w_lo, const float* bias, float* y,
                                         float* stats, int B, int H, int Wd, int Cin, int Cout, int kh, int kw, int sh,
                                         int sw, int ph, int pw, int act, int nsplit, void* stream) {
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
  if (nsplit == 3) {
    static bool attr3 = false;
    if (!attr3) { hipFuncSetAttribute((const void*)conv_bf16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax); attr3 = true; }
    hipLaunchKernelGGL(conv_bf16_kernel<3>, dim3(tiles), dim3(NT), ldsz, (hipStream_t)stream, p, (const unsigned short*)w_hi,
                       (const unsigned short*)w_lo);
  } else {
    static bool attr1 = false;
    if (!attr1) { hipFuncSetAttribute((const void*)conv_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldmax); attr1 = true; }
    hipLaunchKernelGGL(conv_bf16_kernel<1>, dim3(tiles), dim3(NT), ldsz, (hipStream_t)stream, p, (const unsigned short*)w_hi,
                       (const unsigned short*)w_lo);
  }
  MRN_LAUNCH_CHECK("conv_bf16split");
  return MRN_OK;
}
