// Internal descriptor for the fp32 MFMA GEMM / implicit-GEMM conv kernel (gemm.hip).
#pragma once
#include <hip/hip_runtime.h>

struct GemmParams {
  const float* A;     // strided matrix, or NHWC activation in conv mode
  const float* W;     // W[n][k] (general strides)
  const float* bias;  // nullptr, per-column (bias_axis 0) or per-row (bias_axis 1)
  const float* res;   // optional residual, same strides as C, added before the activation
  float* C;
  int M, N, K, batch;
  long sAb, sAm, sAk;
  long sWb, sWn, sWk;
  long sCb, sCm, sCn;
  long sBiasB;
  int bias_axis;
  int amode, wmode;  // staging mode per operand: 0 scalar k-fast, 1 scalar row-fast, 2 float4 along k, 3 float4 along row
  // conv mode: A is [B][H][Wd][Cin], M = B*Ho*Wo, K = kh*kw*Cin
  int H, Wd, Cin, Ho, Wo, kh, kw, sh, sw, ph, pw;
  int act;         // 0 none, 1 relu, 2 gelu(erf)
  int accumulate;  // C += result
  float alpha;     // scale on the product
  float* stats;    // [ceil(M/128)][2][N] per-row-block column sums / sums of squares of (alpha*AB + bias), or nullptr
};

int mrn_gemm_launch(const GemmParams& p, bool conv, hipStream_t st);
