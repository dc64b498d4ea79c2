// Row-wise HBM-bound operators used by the DM-Router and the heads: LayerNorm (over the contiguous dim and
// over a strided "patch" dim), GELU, gating products, column sums (bias / affine gradients, split-K combine),
// index gathers and argmax.  Rows have a contiguous last dim and an arbitrary row stride (ld), so chunk()/
// rearrange() views of the reference (modules/dm_router.py:12,58-65) never need a copy.
//
// Reference op sites: nn.LayerNorm modules/dm_router.py:8,23,40 (eps 1e-5); nn.GELU :42; u*v :17, x*v :33.
#include "common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm over the contiguous dim: one wave per row, C <= 1024 (C % 4 == 0), two-pass in registers.
// ---------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 4;  // float4 per lane -> C <= 64*4*4 = 1024

// HL: the result also as the range-scaled HL32 operand of the Linear layer that follows (y_hl [rows][C/32][hi 32 | lo 32], C % 32 == 0),
// with a scale that needs no pass over the data: |xhat| < sqrt(C), so |y| <= sqrt(C) max|gamma| + max|beta| -- every wave derives
// the same power of two s (s * bound <= target) from the parameters it reads anyway; scale_out = {s, 1/s} for the GEMM's epilogue
template <bool HL>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, long ldy, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, long rows, int C, float eps,
                                                            unsigned char* __restrict__ y_hl, float* __restrict__ scale_out, float target) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int C4 = C >> 2;
  float sc = 1.f;
  if constexpr (HL) {
    float mg = 0.f, mb = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c4 = lane + i * 64;
      if (c4 < C4) {
        const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[c4];
        const f32x4 b = reinterpret_cast<const f32x4*>(beta)[c4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { mg = fmaxf(mg, fabsf(g[j])); mb = fmaxf(mb, fabsf(b[j])); }
      }
    }
    const float bound = sqrtf((float)C) * wave_max(mg) + wave_max(mb);
    if (bound > 0.f && isfinite(bound)) sc = exp2f(floorf(log2f(target / bound)));
    if (row == 0 && lane == 0) { scale_out[0] = sc; scale_out[1] = 1.f / sc; }
  }
  f32x4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + i * 64;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c4 < C4) v[i] = reinterpret_cast<const f32x4*>(x + row * ldx)[c4];
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + i * 64;
    if (c4 < C4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + i * 64;
    if (c4 < C4) {
      const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[c4];
      const f32x4 b = reinterpret_cast<const f32x4*>(beta)[c4];
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      reinterpret_cast<f32x4*>(y + row * ldy)[c4] = o;
      if constexpr (HL) {
        typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
        f16v4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          _Float16 hh, ll;
          split_f16_sat(o[j] * sc, hh, ll);
          hi[j] = hh; lo[j] = ll;
        }
        unsigned char* line = y_hl + (row * (C >> 5) + (c4 >> 3)) * 128 + (c4 & 7) * 8;
        *reinterpret_cast<f16v4*>(line) = hi;
        *reinterpret_cast<f16v4*>(line + 64) = lo;
      }
    }
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); per-block partial sums of dgamma / dbeta.
// Each block walks ROWS_PER_BLOCK rows, one wave per row at a time; partials land in part[blk][2][C].
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, long lddy,
                                                            const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx, long lddx,
                                                            int accumulate, float* __restrict__ part, long rows, int C,
                                                            int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  f32x4 dg[LN_MAXV], db[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) { dg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; db[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(rows, r0 + rows_per_block);
  for (long row = r0 + wave; row < r1; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[LN_MAXV], gy[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c4 = lane + i * 64;
      xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      gy[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c4 < C4) {
        const f32x4 xv = reinterpret_cast<const f32x4*>(x + row * ldx)[c4];
        const f32x4 dv = reinterpret_cast<const f32x4*>(dy + row * lddy)[c4];
        const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[c4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = (xv[j] - mu) * rs;
          gy[i][j] = dv[j] * g[j];
          s1 += gy[i][j];
          s2 += gy[i][j] * xh[i][j];
          dg[i][j] += dv[j] * xh[i][j];
          db[i][j] += dv[j];
        }
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c4 = lane + i * 64;
      if (c4 < C4) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rs * (gy[i][j] - s1 - xh[i][j] * s2);
        f32x4* dst = reinterpret_cast<f32x4*>(dx + row * lddx) + c4;
        if (accumulate) {
          const f32x4 old = *dst;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += old[j];
        }
        *dst = o;
      }
    }
  }
  if (!part) return;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + i * 64;
    if (c4 < C4) {
      reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * C)[c4] = dg[i];
      reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * C)[c4] = db[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int which = c / C, cc = c - which * C;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * C + cc];
    part[((long)blockIdx.x * 2 + which) * C + cc] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over a strided axis: x[b][p][w] (w contiguous, W columns), normalise every (b, w) column over
// the P entries; affine parameters are indexed by p.  This is ChannelDomainGating's LayerNorm(patch) applied
// to the 'b (d c) p' rearrangement (modules/dm_router.py:23,29,63) without materialising the transpose.
// One thread per column; loads are coalesced across w.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          int P, int Wd, float eps) {
  const int b = blockIdx.y;
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= Wd) return;
  const float* xb = x + (long)b * P * Wd + w;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += xb[(long)p * Wd];
  const float mean = s / (float)P;
  float q = 0.f;
  for (int p = 0; p < P; ++p) { const float d = xb[(long)p * Wd] - mean; q += d * d; }
  const float rstd = 1.f / sqrtf(q / (float)P + eps);
  mean_out[(long)b * Wd + w] = mean;
  rstd_out[(long)b * Wd + w] = rstd;
  float* yb = y + (long)b * P * Wd + w;
  for (int p = 0; p < P; ++p) yb[(long)p * Wd] = (xb[(long)p * Wd] - mean) * rstd * gamma[p] + beta[p];
}

// backward of colnorm; per-block partial sums of dgamma[p], dbeta[p] into part[blk][2][P]
__global__ __launch_bounds__(256) void colnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ dx,
                                                          int accumulate, float* __restrict__ part, int P, int Wd) {
  __shared__ float scratch[4];
  const int b = blockIdx.y;
  const int w = blockIdx.x * 256 + threadIdx.x;
  const bool ok = w < Wd;
  const long base = (long)b * P * Wd + (ok ? w : 0);
  const float mu = ok ? mean[(long)b * Wd + w] : 0.f, rs = ok ? rstd[(long)b * Wd + w] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok) {
    for (int p = 0; p < P; ++p) {
      const float xh = (x[base + (long)p * Wd] - mu) * rs;
      const float gy = dy[base + (long)p * Wd] * gamma[p];
      s1 += gy;
      s2 += gy * xh;
    }
  }
  s1 /= (float)P;
  s2 /= (float)P;
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  for (int p = 0; p < P; ++p) {
    float dgp = 0.f, dbp = 0.f;
    if (ok) {
      const float xh = (x[base + (long)p * Wd] - mu) * rs;
      const float d = dy[base + (long)p * Wd];
      float o = rs * (d * gamma[p] - s1 - xh * s2);
      if (accumulate) o += dx[base + (long)p * Wd];
      dx[base + (long)p * Wd] = o;
      dgp = d * xh;
      dbp = d;
    }
    const float tg = block_sum<256>(dgp, scratch);
    const float tb = block_sum<256>(dbp, scratch);
    if (threadIdx.x == 0) {
      part[((long)blk * 2 + 0) * P + p] = tg;
      part[((long)blk * 2 + 1) * P + p] = tb;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// elementwise over strided rows (C % 4 == 0)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}

// op 0: y = gelu(a)   1: y = b * gelu'(a)   2: y = a * b   3: y = a + b   4: y = b * (a > 0)
//    5: y = sigmoid(a)   6: y = b * a * (1 - a)  (a = sigmoid output)   7: y = relu(a + b)      (5-7: the gated recurrent conv
//    layer of the RCNN extractor, modules/feature_extraction.py:146-161)
__global__ __launch_bounds__(256) void ew_rows_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b,
                                                      long ldb, float* __restrict__ y, long ldy, long rows, int C4, int op) {
  const long n = rows * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const int c4 = (int)(i - r * C4);
    const f32x4 av = reinterpret_cast<const f32x4*>(a + r * lda)[c4];
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (b) bv = reinterpret_cast<const f32x4*>(b + r * ldb)[c4];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (op == 0) o[j] = gelu_f(av[j]);
      else if (op == 1) o[j] = bv[j] * gelu_grad_f(av[j]);
      else if (op == 2) o[j] = av[j] * bv[j];
      else if (op == 3) o[j] = av[j] + bv[j];
      else if (op == 4) o[j] = av[j] > 0.f ? bv[j] : 0.f;
      else if (op == 5) o[j] = 1.f / (1.f + expf(-av[j]));
      else if (op == 6) o[j] = bv[j] * av[j] * (1.f - av[j]);
      else o[j] = relu_nan(av[j] + bv[j]);
    }
    reinterpret_cast<f32x4*>(y + r * ldy)[c4] = o;
  }
}

// Elementwise producer of a TRAINED Linear layer's operand (SVTR blocks in loop A, contiguous rows): y = op(a, b) in fp32 and, in the same pass,
//   y_hl   the range-scaled HL32 operand split(scale[0] * y) ([rows][C/32][hi 32 | lo 32], C % 32 == 0), for a scale the caller already has
//          (a bound: |gelu(f)| <= |f|, so max|f| from the fc1 GEMM's epilogue serves gelu(f));
//   amax_ws  max|y| folded into 64 words (mrn_pow2_finalize_f32) for a consumer that needs the exact range (gradients).
// op 0: gelu(a)   1: b * gelu'(a)   3: a + b   8: a + b * d[row / rows_per_d] (residual add with the per-sample DropPath multiplier)
__global__ __launch_bounds__(256) void ew_operand_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ d,
                                                         long rows_per_d, float* __restrict__ y, unsigned char* __restrict__ y_hl,
                                                         const float* __restrict__ scale, unsigned* __restrict__ amax_ws, long n4, int C4, int op) {
  typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
  const float sc = (y_hl && scale) ? scale[0] : 1.f;
  float mx = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 av = reinterpret_cast<const f32x4*>(a)[i];
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (b) bv = reinterpret_cast<const f32x4*>(b)[i];
    f32x4 o;
    if (op == 8) {
      const float dv = d[(i / C4) / rows_per_d];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaf(dv, bv[j], av[j]);      // (as mrn_residual_scale_rows_f32)
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (op == 0) o[j] = gelu_f(av[j]);
        else if (op == 1) o[j] = bv[j] * gelu_grad_f(av[j]);
        else o[j] = av[j] + bv[j];
      }
    }
    if (y) reinterpret_cast<f32x4*>(y)[i] = o;
    if (y_hl) {
      f16v4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        _Float16 hh, ll;
        split_f16_sat(o[j] * sc, hh, ll);
        hi[j] = hh; lo[j] = ll;
      }
      const long r = i / C4;
      const int c4 = (int)(i - r * C4);
      unsigned char* line = y_hl + (r * (C4 >> 3) + (c4 >> 3)) * 128 + (c4 & 7) * 8;
      *reinterpret_cast<f16v4*>(line) = hi;
      *reinterpret_cast<f16v4*>(line + 64) = lo;
    }
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (amax_ws) {
    __shared__ float wmax[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      if (!(mx <= 0.f)) atomicMax(amax_ws + (blockIdx.x & 63), __float_as_uint(mx));
    }
  }
}

// BatchNorm batch statistics of a tensor that is NOT a conv output (the RCNN extractor normalises sums and gated products):
// part[blk][0][c] = sum over the block's rows of x[r][c], part[blk][1][c] = sum of squares -- the partial-sum layout that
// mrn_bn_finalize_f32 consumes (same as the conv epilogues').  One block = BN_STATS_ROWS rows, thread = column.
constexpr int BN_STATS_ROWS = 256;
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, long rows, int C, float* __restrict__ part) {
  const long r0 = (long)blockIdx.x * BN_STATS_ROWS, r1 = min(rows, r0 + BN_STATS_ROWS);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    long r = r0;
    for (; r + 1 < r1; r += 2) {
      const float a = x[r * C + c], b = x[(r + 1) * C + c];
      s0 += a; q0 += a * a;
      s1 += b; q1 += b * b;
    }
    if (r < r1) { const float a = x[r * C + c]; s0 += a; q0 += a * a; }
    part[((long)blockIdx.x * 2 + 0) * C + c] = s0 + s1;
    part[((long)blockIdx.x * 2 + 1) * C + c] = q0 + q1;
  }
}

// out[c] (+)= sum_r in[r*ld + c]; two-stage: grid (colblocks, rowchunks) -> part, then a final pass when chunks > 1
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, long ld, float* __restrict__ out,
                                                     long rows, int C, long rows_per_chunk, int accumulate, int direct) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += in[r * ld + c];
    s1 += in[(r + 1) * ld + c];
    s2 += in[(r + 2) * ld + c];
    s3 += in[(r + 3) * ld + c];
  }
  for (; r < r1; ++r) s0 += in[r * ld + c];
  const float s = (s0 + s1) + (s2 + s3);
  if (direct) out[c] = accumulate ? out[c] + s : s;
  else out[(long)blockIdx.y * C + c] = s;
}

// the same sums with 16-byte loads: a thread owns 4 consecutive columns, a block is TX column quads x TY row lanes (TX * TY = 256,
// TX = min(C / 4, 256) rounded up to a power of two), every row lane walks its rows with four loads in flight, the row lanes meet in
// LDS.  Narrow matrices (SVTR's C = 64 ... 256 bias / LayerNorm gradients over 131 k rows: 262 launches, 18 % of a loop-A step with the
// one-column-per-thread kernel above, a quarter of whose lanes were idle at C = 64) keep every lane busy.  C % 4 == 0, ld % 4 == 0, 16-byte
// aligned base.
template <int TX>
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ in, long ld, float* __restrict__ out, long rows, int C,
                                                      long rows_per_chunk, int accumulate, int direct) {
  constexpr int TY = 256 / TX;
  __shared__ f32x4 red[TY][TX];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int c = (blockIdx.x * TX + tx) * 4;
  const long r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (c < C) {
    long r = r0 + ty;
    for (; r + 3 * TY < r1; r += 4 * TY) {
      s0 += *reinterpret_cast<const f32x4*>(in + r * ld + c);
      s1 += *reinterpret_cast<const f32x4*>(in + (r + TY) * ld + c);
      s2 += *reinterpret_cast<const f32x4*>(in + (r + 2 * TY) * ld + c);
      s3 += *reinterpret_cast<const f32x4*>(in + (r + 3 * TY) * ld + c);
    }
    for (; r < r1; r += TY) s0 += *reinterpret_cast<const f32x4*>(in + r * ld + c);
  }
  f32x4 s = (s0 + s1) + (s2 + s3);
  if constexpr (TY > 1) {
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0) {
#pragma unroll
      for (int j = 1; j < TY; ++j) s += red[j][tx];
    }
  }
  if (ty == 0 && c < C) {
    float* o = direct ? out + c : out + (long)blockIdx.y * C + c;
    if (direct && accumulate) s += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = s;
  }
}

// out[i][j] = in[ridx[i]][cidx[j]] (index arrays may be null = identity); out row stride ld_out, padding cols zeroed
__global__ void gather2d_kernel(const float* __restrict__ in, long ld_in, const int* __restrict__ ridx,
                                const int* __restrict__ cidx, float* __restrict__ out, long ld_out, int R, int C, int Cpad) {
  const long n = (long)R * Cpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / Cpad), c = (int)(i - (long)r * Cpad);
    float v = 0.f;
    if (c < C) v = in[(long)(ridx ? ridx[r] : r) * ld_in + (cidx ? cidx[c] : c)];
    out[(long)r * ld_out + c] = v;
  }
}

// first index of the maximum along the last dim (torch.max semantics), one wave per row
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ x, long ld, int64_t* __restrict__ out,
                                                     long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = x[row * ld + c];
    if (v > best || (v != v && best == best)) { best = v; bi = c; }  // NaN propagates like torch
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    const bool take = (ov > best) || (ov == best && oi < bi) || (ov != ov && best == best);
    if (take) { best = ov; bi = oi; }
  }
  if (lane == 0) out[row] = bi;
}

// greedy decoding + confidence in one pass (reference test.py:211,218-219: preds.max(2) and F.softmax(preds, 2).max(2)):
// idx[row] = first argmax, prob[row] = softmax(x[row])[argmax] = 1 / sum_c exp(x[c] - max).  One wave per row.
__global__ __launch_bounds__(256) void argmax_prob_kernel(const float* __restrict__ x, long ld, int64_t* __restrict__ idx,
                                                          float* __restrict__ prob, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = x[row * ld + c];
    if (v > best || (v != v && best == best)) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    const bool take = (ov > best) || (ov == best && oi < bi) || (ov != ov && best == best);
    if (take) { best = ov; bi = oi; }
  }
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(x[row * ld + c] - best);
  s = wave_sum(s);
  if (lane == 0) {
    idx[row] = bi;
    prob[row] = 1.f / s;
  }
}

// In-place row softmax of s[b][i][:] (N columns) with an optional additive mask[i][:] shared by all b:
// softmax(q k^T + mask) of SVTR's Local mixing (modules/svtr.py:140-146).  One wave per row, N <= 1024.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, const float* __restrict__ mask, long rows,
                                                          int N, int rows_per_mask) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* sr = s + row * N;
  const float* mr = mask ? mask + (row % rows_per_mask) * (long)N : nullptr;
  float v[16];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    v[i] = -INFINITY;
    if (c < N) v[i] = sr[c] + (mr ? mr[c] : 0.f);
    m = fmaxf(m, v[i]);
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    v[i] = (c < N) ? expf(v[i] - m) : 0.f;
    sum += v[i];
  }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    if (c < N) sr[c] = v[i] * inv;
  }
}

// Backward of the row softmax, in place on dp: ds = p * (dp - sum_j p_j dp_j).  One wave per row, N <= 1024.
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long rows, int N) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pr = p + row * N;
  float* dr = dp + row * N;
  float pv[16], dv[16];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    pv[i] = dv[i] = 0.f;
    if (c < N) { pv[i] = pr[c]; dv[i] = dr[c]; }
    dot = fmaf(pv[i], dv[i], dot);
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    if (c < N) dr[c] = pv[i] * (dv[i] - dot);
  }
}

// y = x + scale[row / rows_per_group] * branch   (DropPath: per-sample Bernoulli(keep)/keep scale, svtr.py:7-22,202-203)
__global__ __launch_bounds__(256) void residual_scale_kernel(const float* __restrict__ x, const float* __restrict__ br,
                                                            const float* __restrict__ scale, float* __restrict__ y, long rows,
                                                            int C4, long rows_per_group) {
  const long n = rows * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const float sc = scale[r / rows_per_group];
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 b = reinterpret_cast<const f32x4*>(br)[i];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaf(sc, b[j], a[j]);
    reinterpret_cast<f32x4*>(y)[i] = o;
  }
}

}  // namespace

static inline int ew_grid(long n, int per_block) {
  long g = (n + per_block - 1) / per_block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

MRN_EXPORT int mrn_layernorm_fwd_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                                     int64_t ldy, float* mean, float* rstd, int64_t rows, int C, float eps, void* stream) {
  MRN_CHECK_ARG(x && gamma && beta && y, "mrn_layernorm_fwd_f32: null operand");
  MRN_CHECK_ARG(C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ldy % 4 == 0, "mrn_layernorm_fwd_f32: C=%d must be a multiple of 4, <= 1024", C);
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(layernorm_fwd_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx,
                     gamma, beta, y, (long)ldy, mean, rstd, (long)rows, C, eps, (unsigned char*)nullptr, (float*)nullptr, 1.f);
  MRN_LAUNCH_CHECK("layernorm_fwd");
  return MRN_OK;
}

// the same, and the result a second time as the range-scaled split-fp16 operand of the Linear layer it feeds in an expert being TRAINED
// (svtr.py:200-204 norm1 -> qkv, norm2 -> fc1): y_hl [rows][C/32][128 B] = split(s * y), scale_out = {s, 1/s}, s the largest power of two
// with s * (sqrt(C) max|gamma| + max|beta|) <= target -- a bound from the parameters, so no max|y| pass and no separate split pass
MRN_EXPORT int mrn_layernorm_fwd_hl32_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                                          float* mean, float* rstd, int64_t rows, int C, float eps, void* y_hl, float* scale_out,
                                          float target, void* stream) {
  MRN_CHECK_ARG(x && gamma && beta && y && y_hl && scale_out && target > 0.f, "mrn_layernorm_fwd_hl32_f32: null operand");
  MRN_CHECK_ARG(C % 32 == 0 && C <= 1024 && ldx % 4 == 0 && ldy % 4 == 0, "mrn_layernorm_fwd_hl32_f32: C=%d must be a multiple of 32, <= 1024", C);
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(layernorm_fwd_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx,
                     gamma, beta, y, (long)ldy, mean, rstd, (long)rows, C, eps, (unsigned char*)y_hl, scale_out, target);
  MRN_LAUNCH_CHECK("layernorm_fwd_hl32");
  return MRN_OK;
}

MRN_EXPORT int64_t mrn_layernorm_bwd_blocks(int64_t rows) {
  int64_t b = (rows + 63) / 64;
  if (b > 1024) b = 1024;
  return b < 1 ? 1 : b;
}

MRN_EXPORT int mrn_layernorm_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                     const float* mean, const float* rstd, float* dx, int64_t lddx, int accumulate,
                                     float* partials, int64_t rows, int C, void* stream) {
  MRN_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "mrn_layernorm_bwd_f32: null operand");
  MRN_CHECK_ARG(C % 4 == 0 && C <= 1024, "mrn_layernorm_bwd_f32: C=%d", C);
  if (rows == 0) return MRN_OK;
  const long nblk = mrn_layernorm_bwd_blocks(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)nblk), dim3(256), sizeof(float) * 8 * C, (hipStream_t)stream, dy,
                     (long)lddy, x, (long)ldx, gamma, mean, rstd, dx, (long)lddx, accumulate, partials, (long)rows, C, rpb);
  MRN_LAUNCH_CHECK("layernorm_bwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_colnorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                   float* rstd, int B, int P, int Wd, float eps, void* stream) {
  MRN_CHECK_ARG(x && gamma && beta && y && mean && rstd, "mrn_colnorm_fwd_f32: null operand");
  if (B == 0) return MRN_OK;
  hipLaunchKernelGGL(colnorm_fwd_kernel, dim3(ceil_div(Wd, 256), B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                     mean, rstd, P, Wd, eps);
  MRN_LAUNCH_CHECK("colnorm_fwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_colnorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                   const float* rstd, float* dx, int accumulate, float* partials, int B, int P, int Wd,
                                   void* stream) {
  MRN_CHECK_ARG(dy && x && gamma && mean && rstd && dx && partials, "mrn_colnorm_bwd_f32: null operand");
  if (B == 0) return MRN_OK;
  hipLaunchKernelGGL(colnorm_bwd_kernel, dim3(ceil_div(Wd, 256), B), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean,
                     rstd, dx, accumulate, partials, P, Wd);
  MRN_LAUNCH_CHECK("colnorm_bwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_ew_rows_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy,
                               int64_t rows, int C, int op, void* stream) {
  MRN_CHECK_ARG(a && y && op >= 0 && op <= 7 && (op == 0 || op == 5 || b), "mrn_ew_rows_f32: bad operands for op %d", op);
  MRN_CHECK_ARG(C % 4 == 0 && lda % 4 == 0 && ldy % 4 == 0 && ldb % 4 == 0, "mrn_ew_rows_f32: C and row strides must be multiples of 4");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(ew_rows_kernel, dim3(ew_grid(rows * (C / 4), 1024)), dim3(256), 0, (hipStream_t)stream, a, (long)lda, b,
                     (long)ldb, y, (long)ldy, (long)rows, C / 4, op);
  MRN_LAUNCH_CHECK("ew_rows");
  return MRN_OK;
}

// see ew_operand_kernel: a, b, y contiguous [rows][C]; d [rows / rows_per_d] (op 8); y and / or y_hl (needs C % 32 == 0) and / or amax_ws
MRN_EXPORT int mrn_ew_operand_f32(const float* a, const float* b, const float* d, int64_t rows_per_d, float* y, void* y_hl,
                                  const float* scale, void* amax_ws, int64_t rows, int C, int op, void* stream) {
  MRN_CHECK_ARG(a && (y || y_hl) && (op == 0 || op == 1 || op == 3 || op == 8) && (op == 0 || b) && (op != 8 || (d && rows_per_d >= 1)),
                "mrn_ew_operand_f32: bad operands for op %d", op);
  MRN_CHECK_ARG(C % 4 == 0 && (!y_hl || (C % 32 == 0 && (uintptr_t)y_hl % 16 == 0)), "mrn_ew_operand_f32: C %% 4 == 0 (C %% 32 == 0 with an HL32 result), got %d", C);
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(ew_operand_kernel, dim3(ew_grid(rows * (C / 4), 1024)), dim3(256), 0, (hipStream_t)stream, a, b, d, (long)rows_per_d, y,
                     (unsigned char*)y_hl, scale, (unsigned*)amax_ws, (long)(rows * (C / 4)), C / 4, op);
  MRN_LAUNCH_CHECK("ew_operand");
  return MRN_OK;
}

MRN_EXPORT int64_t mrn_bn_stats_blocks(int64_t rows) { return (rows + BN_STATS_ROWS - 1) / BN_STATS_ROWS; }

// part [mrn_bn_stats_blocks(rows)][2][C]: per-block column sums / sums of squares of x [rows][C] (input of mrn_bn_finalize_f32)
MRN_EXPORT int mrn_bn_stats_f32(const float* x, int64_t rows, int C, float* part, void* stream) {
  MRN_CHECK_ARG(x && part && C > 0, "mrn_bn_stats_f32: bad operands");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((unsigned)mrn_bn_stats_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)rows, C, part);
  MRN_LAUNCH_CHECK("bn_stats");
  return MRN_OK;
}

static int colsum4_tx(int C) {      // column quads per block of the vectorised kernel: C / 4 rounded up to a power of two, 4 ... 256
  int tx = 4;
  while (tx < 256 && tx * 4 < C) tx *= 2;
  return tx;
}

// column quads per block for a pass over `rows` rows that ends in ONE row of sums: as many row lanes (TY = 256 / TX, up to 64) as keep
// 16 rows each -- a [2048][1024] second pass on one 256-quad block walked 2048 dependent rows per lane (190 us for a 134 MB bias gradient)
static int colsum4_tx_final(long rows, int C) {
  int ty = 1;
  while (ty < 64 && ty * 16L < rows) ty *= 2;
  const int tx = 256 / ty, need = colsum4_tx(C);
  return tx < need ? tx : need;
}

// (chunks, TX of the first pass): few rows -> one pass; otherwise blocks of at most 256 columns with >= 4 row lanes, up to 256 chunks and
// about 4096 blocks, and a second pass planned by colsum4_tx_final
static void colsum_plan(long rows, int C, long& chunks, int& tx1) {
  if (rows <= 512) {
    chunks = 1;
    tx1 = colsum4_tx_final(rows, C);
    return;
  }
  tx1 = colsum4_tx(C);
  if (tx1 > 64) tx1 = 64;
  const int ty1 = 256 / tx1;
  const long colblocks = ceil_div(C, tx1 * 4);
  chunks = 4096 / colblocks;
  chunks = chunks < 1 ? 1 : (chunks > 256 ? 256 : chunks);
  const long min_rows = 16L * ty1;                    // at least 16 rows per row lane and chunk
  if (chunks > (rows + min_rows - 1) / min_rows) chunks = (rows + min_rows - 1) / min_rows;
  if (chunks < 1) chunks = 1;
}

MRN_EXPORT int64_t mrn_colsum_chunks(int64_t rows, int C) {
  long chunks;
  int tx1;
  colsum_plan((long)rows, C, chunks, tx1);
  return chunks;
}

template <int TX>
static void launch_colsum4(const float* in, long ld, float* out, long rows, int C, long rpc, long chunks, int accumulate, int direct,
                           hipStream_t st) {
  hipLaunchKernelGGL(colsum4_kernel<TX>, dim3((unsigned)ceil_div(C, TX * 4), (unsigned)chunks), dim3(256), 0, st, in, ld, out, rows, C, rpc,
                     accumulate, direct);
}

static void colsum4(int tx, const float* in, long ld, float* out, long rows, int C, long rpc, long chunks, int accumulate, int direct,
                    hipStream_t st) {
  switch (tx) {
    case 4: launch_colsum4<4>(in, ld, out, rows, C, rpc, chunks, accumulate, direct, st); break;
    case 8: launch_colsum4<8>(in, ld, out, rows, C, rpc, chunks, accumulate, direct, st); break;
    case 16: launch_colsum4<16>(in, ld, out, rows, C, rpc, chunks, accumulate, direct, st); break;
    case 32: launch_colsum4<32>(in, ld, out, rows, C, rpc, chunks, accumulate, direct, st); break;
    case 64: launch_colsum4<64>(in, ld, out, rows, C, rpc, chunks, accumulate, direct, st); break;
    case 128: launch_colsum4<128>(in, ld, out, rows, C, rpc, chunks, accumulate, direct, st); break;
    default: launch_colsum4<256>(in, ld, out, rows, C, rpc, chunks, accumulate, direct, st); break;
  }
}

// out[c] (+)= sum over rows; workspace must hold mrn_colsum_chunks(rows, C) * C floats (unused when chunks == 1)
MRN_EXPORT int mrn_colsum_f32(const float* in, int64_t ld, float* out, float* workspace, int64_t rows, int C,
                              int accumulate, void* stream) {
  MRN_CHECK_ARG(in && out, "mrn_colsum_f32: null operand");
  if (C == 0) return MRN_OK;
  long chunks;
  int tx1;
  colsum_plan((long)rows, C, chunks, tx1);
  const int cb = ceil_div(C, 256);
  const hipStream_t st = (hipStream_t)stream;
  if (C % 4 == 0 && ld % 4 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0 && (chunks == 1 || (uintptr_t)workspace % 16 == 0)) {
    if (chunks == 1) {
      colsum4(tx1, in, (long)ld, out, (long)rows, C, (long)rows, 1, accumulate, 1, st);
    } else {
      MRN_CHECK_ARG(workspace, "mrn_colsum_f32: workspace required for %ld chunks", chunks);
      const long rpc = (rows + chunks - 1) / chunks;
      colsum4(tx1, in, (long)ld, workspace, (long)rows, C, rpc, chunks, 0, 0, st);
      colsum4(colsum4_tx_final(chunks, C), workspace, (long)C, out, chunks, C, chunks, 1, accumulate, 1, st);
    }
    MRN_LAUNCH_CHECK("colsum");
    return MRN_OK;
  }
  if (chunks == 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(cb, 1), dim3(256), 0, (hipStream_t)stream, in, (long)ld, out, (long)rows, C,
                       (long)rows, accumulate, 1);
  } else {
    MRN_CHECK_ARG(workspace, "mrn_colsum_f32: workspace required for %ld chunks", chunks);
    const long rpc = (rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_kernel, dim3(cb, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, in, (long)ld, workspace,
                       (long)rows, C, rpc, 0, 0);
    hipLaunchKernelGGL(colsum_kernel, dim3(cb, 1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (long)C, out,
                       chunks, C, chunks, accumulate, 1);
  }
  MRN_LAUNCH_CHECK("colsum");
  return MRN_OK;
}

MRN_EXPORT int mrn_gather2d_f32(const float* in, int64_t ld_in, const int* row_idx, const int* col_idx, float* out,
                                int64_t ld_out, int R, int C, void* stream) {
  MRN_CHECK_ARG(in && out && ld_out >= C, "mrn_gather2d_f32: bad operands");
  const long n = (long)R * ld_out;
  if (n == 0) return MRN_OK;
  hipLaunchKernelGGL(gather2d_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, row_idx,
                     col_idx, out, (long)ld_out, R, C, (int)ld_out);
  MRN_LAUNCH_CHECK("gather2d");
  return MRN_OK;
}

MRN_EXPORT int mrn_argmax_f32(const float* x, int64_t ld, int64_t* out, int64_t rows, int C, void* stream) {
  MRN_CHECK_ARG(x && out && C > 0, "mrn_argmax_f32: bad operands");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (long)ld, out,
                     (long)rows, C);
  MRN_LAUNCH_CHECK("argmax");
  return MRN_OK;
}

MRN_EXPORT int mrn_argmax_prob_f32(const float* x, int64_t ld, int64_t* idx, float* prob, int64_t rows, int C, void* stream) {
  MRN_CHECK_ARG(x && idx && prob && C > 0, "mrn_argmax_prob_f32: bad operands");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(argmax_prob_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (long)ld, idx,
                     prob, (long)rows, C);
  MRN_LAUNCH_CHECK("argmax_prob");
  return MRN_OK;
}

// ds = p * (dp - rowsum(p * dp)), in place on dp: backward of mrn_softmax_rows_f32 (SVTR attention, expert training)
MRN_EXPORT int mrn_softmax_rows_bwd_f32(const float* p, float* dp, int64_t rows, int N, void* stream) {
  MRN_CHECK_ARG(p && dp && N > 0 && N <= 1024, "mrn_softmax_rows_bwd_f32: bad operands (N=%d)", N);
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp,
                     (long)rows, N);
  MRN_LAUNCH_CHECK("softmax_rows_bwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_softmax_rows_f32(float* s, const float* mask, int64_t rows, int N, int rows_per_mask, void* stream) {
  MRN_CHECK_ARG(s && N > 0 && N <= 1024 && rows_per_mask > 0, "mrn_softmax_rows_f32: bad operands (N=%d)", N);
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, mask, (long)rows,
                     N, rows_per_mask);
  MRN_LAUNCH_CHECK("softmax_rows");
  return MRN_OK;
}

MRN_EXPORT int mrn_residual_scale_rows_f32(const float* x, const float* branch, const float* scale, float* y, int64_t rows,
                                           int C, int64_t rows_per_group, void* stream) {
  MRN_CHECK_ARG(x && branch && scale && y && C % 4 == 0 && rows_per_group > 0, "mrn_residual_scale_rows_f32: bad operands");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(residual_scale_kernel, dim3(ew_grid(rows * (C / 4), 1024)), dim3(256), 0, (hipStream_t)stream, x, branch,
                     scale, y, (long)rows, C / 4, (long)rows_per_group);
  MRN_LAUNCH_CHECK("residual_scale_rows");
  return MRN_OK;
}
