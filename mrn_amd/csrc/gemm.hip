// fp32 MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[b][m][n] = act( sum_k A[b][m][k] * W[b][n][k] + bias )      (+ optional per-column sum / sumsq partials)
//
// A is either a strided matrix (any of the two dims may be the contiguous one) or an NHWC activation
// gathered on the fly (implicit im2col: m = (img, oy, ox), k = (ky, kx, ci)).  The math is
// v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate -- bitwise a k-ordered fmaf chain -- so
// results stay inside the reference's 1e-4 fp32 parity band (north star).
//
// Replaces the reference's conv2d / linear / bmm op sites:
//   modules/feature_extraction.py:19-44,214-294   modules/transformation.py:60-87
//   modules/sequence_modeling.py:10  modules/prediction.py:104-107  modules/dm_router.py:41-46
//   modules/model.py:151,437-438
//
// Tiling: block = WM x WN waves (64 lanes each), each wave owns TM x TN tiles of 32x32; BK = 16.
// LDS image is k-major ([k][row], row contiguous) so the MFMA operand read (lane -> row = lane&31,
// k = lane>>5) is a conflict-free ds_read_b32 of 32 consecutive floats per half-wave.
#include "common.hpp"
#include "gemm.hpp"
#include <stdlib.h>

namespace {

// floats staged per thread for a ROWS x BK tile (at least one float4)
constexpr int stage_regs(int rows, int nt, int bk) { return rows * bk / nt < 4 ? 4 : rows * bk / nt; }

// ---- strided operand: element (row, k) at base[row*sr + k*sk] --------------------------------
// mode 0: scalar, k fastest    1: scalar, row fastest    2: float4 along k    3: float4 along row
template <int ROWS, int NT, int BK>
struct Stage {
  static constexpr int KQ = BK / 4;                                          // float4 groups along k
  static constexpr int RPP2 = NT / KQ;                                        // rows per pass, float4 along k
  static constexpr int NP2 = ROWS / RPP2 > 0 ? ROWS / RPP2 : 1;
  static constexpr int RQ = ROWS / 4;                                      // row quads per k
  static constexpr int KPP3 = (NT / RQ) < BK ? (NT / RQ) : BK;
  static constexpr int NP3 = BK / KPP3;
  static constexpr int NP0 = ROWS / (NT / BK) > 0 ? ROWS / (NT / BK) : 1;
  static constexpr int KPP1 = (NT / ROWS) < BK ? (NT / ROWS) : BK;
  static constexpr int NP1 = BK / KPP1;
  static constexpr int LD = ROWS + 4;
};

template <int ROWS, int NT, int BK>
__device__ __forceinline__ void load_strided(float (&r)[stage_regs(ROWS, NT, BK)], const float* __restrict__ base,
                                             long sr, long sk, int row0, int k0, int nrows, int K, int mode) {
  using S = Stage<ROWS, NT, BK>;
  const int t = threadIdx.x;
  if (mode == 2) {
    const int k = k0 + (t % S::KQ) * 4;
#pragma unroll
    for (int p = 0; p < S::NP2; ++p) {
      const int rl = (t / S::KQ) + p * S::RPP2;
      const int row = row0 + rl;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (rl < ROWS && row < nrows && k < K) v = *reinterpret_cast<const f32x4*>(base + (long)row * sr + k);
      r[p * 4 + 0] = v[0]; r[p * 4 + 1] = v[1]; r[p * 4 + 2] = v[2]; r[p * 4 + 3] = v[3];
    }
  } else if (mode == 3) {
    const int rq = t % S::RQ, kk = t / S::RQ;
    const int row = row0 + rq * 4;
#pragma unroll
    for (int p = 0; p < S::NP3; ++p) {
      const int k = k0 + kk + p * S::KPP3;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kk < S::KPP3 && row < nrows && k < K) v = *reinterpret_cast<const f32x4*>(base + (long)k * sk + row);
      r[p * 4 + 0] = v[0]; r[p * 4 + 1] = v[1]; r[p * 4 + 2] = v[2]; r[p * 4 + 3] = v[3];
    }
  } else if (mode == 0) {
    const int k = k0 + (t % BK);
#pragma unroll
    for (int p = 0; p < S::NP0; ++p) {
      const int rl = (t / BK) + p * (NT / BK);
      const int row = row0 + rl;
      r[p] = (rl < ROWS && row < nrows && k < K) ? base[(long)row * sr + (long)k * sk] : 0.f;
    }
  } else {
    const int rr = t % ROWS, kk = t / ROWS;
    const int row = row0 + rr;
#pragma unroll
    for (int p = 0; p < S::NP1; ++p) {
      const int k = k0 + kk + p * S::KPP1;
      r[p] = (kk < S::KPP1 && row < nrows && k < K) ? base[(long)row * sr + (long)k * sk] : 0.f;
    }
  }
}

template <int ROWS, int NT, int BK>
__device__ __forceinline__ void store_lds(const float (&r)[stage_regs(ROWS, NT, BK)], float* __restrict__ s, int mode) {
  using S = Stage<ROWS, NT, BK>;
  constexpr int LD = S::LD;
  const int t = threadIdx.x;
  if (mode == 2) {
    const int kq = (t % S::KQ) * 4;
#pragma unroll
    for (int p = 0; p < S::NP2; ++p) {
      const int rl = (t / S::KQ) + p * S::RPP2;
      if (rl < ROWS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[(kq + j) * LD + rl] = r[p * 4 + j];
      }
    }
  } else if (mode == 3) {
    const int rq = t % S::RQ, kk = t / S::RQ;
#pragma unroll
    for (int p = 0; p < S::NP3; ++p) {
      if (kk < S::KPP3) {
        f32x4 v = {r[p * 4 + 0], r[p * 4 + 1], r[p * 4 + 2], r[p * 4 + 3]};
        *reinterpret_cast<f32x4*>(s + (kk + p * S::KPP3) * LD + rq * 4) = v;
      }
    }
  } else if (mode == 0) {
    const int kk = t % BK;
#pragma unroll
    for (int p = 0; p < S::NP0; ++p) {
      const int rl = (t / BK) + p * (NT / BK);
      if (rl < ROWS) s[kk * LD + rl] = r[p];
    }
  } else {
    const int rr = t % ROWS, kk = t / ROWS;
#pragma unroll
    for (int p = 0; p < S::NP1; ++p)
      if (kk < S::KPP1) s[(kk + p * S::KPP1) * LD + rr] = r[p];
  }
}

// ---- implicit-GEMM gather from an NHWC activation (Cin % 4 == 0) -------------------------------
struct ConvRow {
  long base;  // element offset of image b
  int iy0, ix0;
  bool ok;
};

// tap table entry for a 4-float group of the K axis: (ky, kx, ci) packed; built once per block in LDS so the
// K loop carries no integer division
__device__ __forceinline__ int pack_tap(int k, int Cin, int kw) {
  const int tap = k / Cin;
  const int ci = k - tap * Cin;
  const int ky = tap / kw;
  const int kx = tap - ky * kw;
  return (ky << 26) | (kx << 20) | ci;
}

template <int ROWS, int NT, int BK>
__device__ __forceinline__ void load_conv(float (&r)[stage_regs(ROWS, NT, BK)], const GemmParams& p,
                                          const ConvRow (&rows)[Stage<ROWS, NT, BK>::NP2], const int* __restrict__ taps,
                                          int k0) {
  using S = Stage<ROWS, NT, BK>;
  const int t = threadIdx.x;
  const int k = k0 + (t % S::KQ) * 4;
  const bool kok = k < p.K;
  const int info = taps[kok ? (k >> 2) : 0];
  const int ky = info >> 26, kx = (info >> 20) & 63, ci = info & 0xfffff;
#pragma unroll
  for (int q = 0; q < S::NP2; ++q) {
    const int iy = rows[q].iy0 + ky, ix = rows[q].ix0 + kx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kok && rows[q].ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd)
      v = *reinterpret_cast<const f32x4*>(p.A + rows[q].base + ((long)iy * p.Wd + ix) * p.Cin + ci);
    r[q * 4 + 0] = v[0]; r[q * 4 + 1] = v[1]; r[q * 4 + 2] = v[2]; r[q * 4 + 3] = v[3];
  }
}

// ---- weight-gradient gather (W operand, wmode 4): element (n = (tap, ci), k = output pixel) is the input pixel the
// tap sees, x[b][oy*sh-ph+ky][ox*sw-pw+kx][ci] (0 outside).  n is contiguous in ci, so the LDS image is written as
// float4 along the row exactly like mode 3.  pix0 = first pixel of this block's split-K chunk.
template <int ROWS, int NT, int BK>
__device__ __forceinline__ void load_wgrad(float (&r)[stage_regs(ROWS, NT, BK)], const GemmParams& p, int n0, int k0,
                                           long pix0, int kchunk) {
  using S = Stage<ROWS, NT, BK>;
  const int t = threadIdx.x;
  const int rq = t % S::RQ, kk = t / S::RQ;
  const int n = n0 + rq * 4;
  const bool nok = n < p.N;
  const int tap = nok ? n / p.Cin : 0;
  const int ci = n - tap * p.Cin;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
  for (int q = 0; q < S::NP3; ++q) {
    const int k = k0 + kk + q * S::KPP3;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kk < S::KPP3 && nok && k < kchunk) {
      const long pix = pix0 + k;
      const int hw = p.Ho * p.Wo;
      const int b = (int)(pix / hw);
      const int rem = (int)(pix - (long)b * hw);
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int iy = oy * p.sh - p.ph + ky, ix = ox * p.sw - p.pw + kx;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd)
        v = *reinterpret_cast<const f32x4*>(p.W + (((long)b * p.H + iy) * p.Wd + ix) * p.Cin + ci);
    }
    r[q * 4 + 0] = v[0]; r[q * 4 + 1] = v[1]; r[q * 4 + 2] = v[2]; r[q * 4 + 3] = v[3];
  }
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return relu_nan(v);
  if (act == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}

constexpr int MAX_TAPS = 2048;   // K/4 entries of the tap table (K <= 8192)

template <int WM, int WN, int TM, int TN, int BK, bool CONV>
__global__ __launch_bounds__(WM* WN * 64) void gemm_f32_kernel(const GemmParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int PA = stage_regs(BM, NT, BK), PB = stage_regs(BN, NT, BK);
  constexpr int NPA = Stage<BM, NT, BK>::NP2;
  constexpr int RPP = Stage<BM, NT, BK>::RPP2, KQ = Stage<BM, NT, BK>::KQ;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB) + (CONV ? MAX_TAPS : 0)];
  float* const As = smem;                 // two buffers of BK*LDA
  float* const Bs = smem + 2 * BK * LDA;  // two buffers of BK*LDB
  int* const taps = reinterpret_cast<int*>(smem + 2 * BK * (LDA + LDB));

  const int tilesN = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int tile_m = lid / tilesN, tile_n = lid - tile_m * tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int bz = blockIdx.z;

  const float* A = p.A + (CONV ? 0 : (long)bz * p.sAb);
  const float* W = p.W + (p.wmode == 4 ? 0 : (long)bz * p.sWb);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;

  ConvRow crow[NPA];
  if (CONV) {
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int rl = (t / KQ) + q * RPP;
      const int m = m0 + rl;
      const bool ok = rl < BM && m < p.M;
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
    for (int i = t; i < (p.K + 3) / 4; i += NT) taps[i] = pack_tap(i * 4, p.Cin, p.kw);
    __syncthreads();
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float ra[PA], rb[PB];
  const int nk = (p.K + BK - 1) / BK;

  if (CONV) load_conv<BM, NT, BK>(ra, p, crow, taps, 0);
  else load_strided<BM, NT, BK>(ra, A, p.sAm, p.sAk, m0, 0, p.M, p.K, p.amode);
  const bool wg = !CONV && p.wmode == 4;      // weight-gradient gather on the W operand
  const long pix0 = (long)bz * p.K;
  if (wg) load_wgrad<BN, NT, BK>(rb, p, n0, 0, pix0, p.K);
  else load_strided<BN, NT, BK>(rb, W, p.sWn, p.sWk, n0, 0, p.N, p.K, p.wmode);
  store_lds<BM, NT, BK>(ra, As, CONV ? 2 : p.amode);
  store_lds<BN, NT, BK>(rb, Bs, wg ? 3 : p.wmode);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      if (CONV) load_conv<BM, NT, BK>(ra, p, crow, taps, (kt + 1) * BK);
      else load_strided<BM, NT, BK>(ra, A, p.sAm, p.sAk, m0, (kt + 1) * BK, p.M, p.K, p.amode);
      if (wg) load_wgrad<BN, NT, BK>(rb, p, n0, (kt + 1) * BK, pix0, p.K);
      else load_strided<BN, NT, BK>(rb, W, p.sWn, p.sWk, n0, (kt + 1) * BK, p.N, p.K, p.wmode);
    }
    const float* as = As + cur * BK * LDA + wm * TM * 32 + (lane & 31) + (lane >> 5) * LDA;
    const float* bs = Bs + cur * BK * LDB + wn * TN * 32 + (lane & 31) + (lane >> 5) * LDB;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[kk * 2 * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[kk * 2 * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_lds<BM, NT, BK>(ra, As + (cur ^ 1) * BK * LDA, CONV ? 2 : p.amode);
      store_lds<BN, NT, BK>(rb, Bs + (cur ^ 1) * BK * LDB, wg ? 3 : p.wmode);
    }
    __syncthreads();
  }

  // ---- epilogue: bias, activation, store, optional per-column statistics ------------------------
  float* C = p.C + (long)bz * p.sCb;
  const float* R = p.res ? p.res + (long)bz * p.sCb : nullptr;
  const float* bias = p.bias ? p.bias + (long)bz * p.sBiasB : nullptr;
  float csum[TN], csq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { csum[j] = 0.f; csq[j] = 0.f; }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * TN * 32 + j * 32 + (lane & 31);
    const bool nok = n < p.N;
    const float bn = (bias && p.bias_axis == 0 && nok) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M && nok) {
          float v = acc[i][j][e] * p.alpha + bn;
          if (bias && p.bias_axis == 1) v += bias[m];
          if (R) v += R[(long)m * p.sCm + (long)n * p.sCn];
          csum[j] += v;
          csq[j] += v * v;
          v = apply_act(v, p.act);
          float* dst = C + (long)m * p.sCm + (long)n * p.sCn;
          if (p.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }

  if (p.stats) {
    // rows of this block: combine the two half-waves, then the WM waves through LDS (deterministic)
    __syncthreads();
    float* red = smem;  // [WM][2][BN]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = csum[j] + __shfl_xor(csum[j], 32);
      float q = csq[j] + __shfl_xor(csq[j], 32);
      if (lane < 32) {
        const int c = wn * TN * 32 + j * 32 + lane;
        red[(wm * 2 + 0) * BN + c] = s;
        red[(wm * 2 + 1) * BN + c] = q;
      }
    }
    __syncthreads();
    for (int c = t; c < BN; c += NT) {
      const int n = n0 + c;
      if (n < p.N) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s += red[(w * 2 + 0) * BN + c]; q += red[(w * 2 + 1) * BN + c]; }
        float* st = p.stats;
        st[((long)tile_m * 2 + 0) * p.N + n] = s;
        st[((long)tile_m * 2 + 1) * p.N + n] = q;
      }
    }
  }
}

template <int WM, int WN, int TM, int TN, int BK>
int launch_cfg(const GemmParams& p, bool conv, hipStream_t st, size_t pad_lds = 0) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
  dim3 grid(tiles, 1, p.batch), block(WM * WN * 64);
  if (conv) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, BK, true>), grid, block, pad_lds, st, p);
  else hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, BK, false>), grid, block, pad_lds, st, p);
  MRN_LAUNCH_CHECK("gemm_f32");
  return MRN_OK;
}

}  // namespace

int mrn_gemm_tile_m(int M, int N) {
  (void)M;
  (void)N;
  return 128;
}

int mrn_gemm_launch(const GemmParams& p, bool conv, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || p.batch <= 0) return MRN_OK;
  // every configuration keeps BM = 128 so the statistics workspace is always ceil(M/128) row blocks
  if (conv && p.K > MAX_TAPS * 4) {
    mrn_set_error("conv K=%d exceeds the tap table (%d)", p.K, MAX_TAPS * 4);
    return MRN_ERR_UNSUPPORTED;
  }
  static const int variant = getenv("MRN_GEMM_VARIANT") ? atoi(getenv("MRN_GEMM_VARIANT")) : 0;   // tuning experiments only
  if (p.N <= 32) return launch_cfg<4, 1, 1, 1, 16>(p, conv, st);
  if (p.N <= 64) return launch_cfg<4, 1, 1, 2, 16>(p, conv, st);
  if (variant == 1) return launch_cfg<2, 2, 2, 2, 32>(p, conv, st);   // measured slower (4-way LDS store conflicts, 2 blocks/CU)
  return launch_cfg<2, 2, 2, 2, 16>(p, conv, st);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static int pick_mode(const void* ptr, long sb, long sr, long sk, int nrows, int K) {
  const bool al = ((uintptr_t)ptr % 16 == 0) && (sb % 4 == 0);
  if (sk == 1 && al && sr % 4 == 0 && K % 4 == 0) return 2;
  if (sr == 1 && al && sk % 4 == 0 && nrows % 4 == 0) return 3;
  if (sr == 1 && sk != 1) return 1;
  return 0;
}

MRN_EXPORT int mrn_gemm_f32(const float* A, const float* W, const float* bias, const float* residual, float* C,
                            int M, int N, int K,
                            int batch, int64_t sAb, int64_t sAm, int64_t sAk, int64_t sWb, int64_t sWn,
                            int64_t sWk, int64_t sCb, int64_t sCm, int64_t sCn, int64_t sBiasB, int bias_axis,
                            int act, int accumulate, float alpha, void* stream) {
  MRN_CHECK_ARG(A && W && C, "mrn_gemm_f32: null operand");
  MRN_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && batch >= 0, "mrn_gemm_f32: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  MRN_CHECK_ARG(act >= 0 && act <= 2 && (bias_axis == 0 || bias_axis == 1), "mrn_gemm_f32: bad act/bias_axis");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.W = W; p.bias = bias; p.res = residual; p.C = C;
  p.M = M; p.N = N; p.K = K; p.batch = batch;
  p.sAb = sAb; p.sAm = sAm; p.sAk = sAk;
  p.sWb = sWb; p.sWn = sWn; p.sWk = sWk;
  p.sCb = sCb; p.sCm = sCm; p.sCn = sCn;
  p.sBiasB = sBiasB; p.bias_axis = bias_axis;
  p.amode = pick_mode(A, sAb, sAm, sAk, M, K);
  p.wmode = pick_mode(W, sWb, sWn, sWk, N, K);
  p.act = act; p.accumulate = accumulate; p.alpha = alpha;
  return mrn_gemm_launch(p, false, (hipStream_t)stream);
}

MRN_EXPORT int64_t mrn_conv2d_stats_floats(int B, int Ho, int Wo, int Cout) {
  return (int64_t)ceil_div((long)B * Ho * Wo, 128) * 2 * Cout;
}

MRN_EXPORT int mrn_conv2d_nhwc_f32(const float* x, const float* w_ohwi, const float* bias, float* y, float* stats,
                                   int B, int H, int Wd, int Cin, int Cout, int kh, int kw, int sh, int sw,
                                   int ph, int pw, int act, void* stream) {
  MRN_CHECK_ARG(x && w_ohwi && y, "mrn_conv2d_nhwc_f32: null operand");
  MRN_CHECK_ARG(Cin % 4 == 0, "mrn_conv2d_nhwc_f32: Cin=%d must be a multiple of 4", Cin);
  MRN_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)w_ohwi % 16 == 0), "mrn_conv2d_nhwc_f32: operands must be 16-byte aligned");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (Wd + 2 * pw - kw) / sw + 1;
  MRN_CHECK_ARG(Ho > 0 && Wo > 0 && B >= 0, "mrn_conv2d_nhwc_f32: empty output");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.W = w_ohwi; p.bias = bias; p.C = y; p.stats = stats;
  p.M = B * Ho * Wo; p.N = Cout; p.K = kh * kw * Cin; p.batch = 1;
  p.sWn = p.K; p.sWk = 1; p.wmode = 2;
  p.sCm = Cout; p.sCn = 1;
  p.H = H; p.Wd = Wd; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo;
  p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.act = act; p.alpha = 1.f;
  return mrn_gemm_launch(p, true, (hipStream_t)stream);
}

// dW[s][co][(ky,kx,ci)] = sum over the s-th chunk of output pixels of dy[pix][co] * x[pix shifted by the tap][ci]
// (split-K partials; the caller column-sums the `splits` slabs).  dy: [B*Ho*Wo][Cout] (NHWC), x: NHWC input.
MRN_EXPORT int mrn_conv2d_wgrad_f32(const float* dy, const float* x, float* dw_partial, int B, int H, int Wd, int Cin,
                                    int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int splits, void* stream) {
  MRN_CHECK_ARG(dy && x && dw_partial && splits >= 1, "mrn_conv2d_wgrad_f32: bad operands");
  MRN_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0, "mrn_conv2d_wgrad_f32: Cin=%d, Cout=%d must be multiples of 4", Cin, Cout);
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (Wd + 2 * pw - kw) / sw + 1;
  const long pixels = (long)B * Ho * Wo;
  MRN_CHECK_ARG(pixels % splits == 0, "mrn_conv2d_wgrad_f32: %ld output pixels not divisible by splits=%d", pixels, splits);
  if (pixels == 0) return MRN_OK;
  GemmParams p;
  memset(&p, 0, sizeof(p));
  const int Kc = (int)(pixels / splits);
  p.A = dy; p.W = x; p.C = dw_partial;
  p.M = Cout; p.N = kh * kw * Cin; p.K = Kc; p.batch = splits;
  p.sAb = (long)Kc * Cout; p.sAm = 1; p.sAk = Cout; p.amode = 3;     // A[m = co][k = pixel] = dy[pixel][co]
  p.wmode = 4;
  p.sCb = (long)Cout * p.N; p.sCm = p.N; p.sCn = 1;
  p.H = H; p.Wd = Wd; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo;
  p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.alpha = 1.f;
  return mrn_gemm_launch(p, false, (hipStream_t)stream);
}
