// Backward kernels of the expert stages (loop A, reference il_modules/mrn.py:260-261 `loss.backward()` through
// modules/feature_extraction.py, modules/sequence_modeling.py): data-gradient weight packing, BatchNorm2d (training)
// backward fused with the ReLU mask, max-pool backward, LSTM backward-through-time.
// Convolution data / weight gradients themselves are implicit GEMMs on the forward kernels (gemm.hip):
//   dgrad = conv(dy, flipped-transposed weight), wgrad = mrn_conv2d_wgrad_f32.
#include "common.hpp"

namespace {

// w [O][kh][kw][I] -> wt [I][kh][kw][O] with both taps flipped (the weight of the data-gradient convolution)
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int O, int I, int kh, int kw) {
  const long n = (long)O * I * kh * kw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % O);
    long r = i / O;
    const int kx = (int)(r % kw);
    r /= kw;
    const int ky = (int)(r % kh);
    const int ci = (int)(r / kh);
    wt[i] = w[(((long)co * kh + (kh - 1 - ky)) * kw + (kw - 1 - kx)) * I + ci];
  }
}

// [O][kh][kw][I] -> [O][I][kh][kw] (gradient back to the parameter's layout), optionally accumulating
__global__ void unpack_ohwi_oihw_kernel(const float* __restrict__ g, float* __restrict__ out, int O, int I, int khw,
                                        int accumulate) {
  const long n = (long)O * I * khw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % khw);
    const long r = i / khw;
    const int ci = (int)(r % I);
    const int oc = (int)(r / I);
    const float v = g[((long)oc * khw + tap) * I + ci];
    out[i] = accumulate ? out[i] + v : v;
  }
}

// zero-insertion: dy [B][Ho][Wo][C] -> out [B][(Ho-1)*sh+1][(Wo-1)*sw+1][C]  (data gradient of a strided conv)
__global__ void dilate_nhwc_kernel(const float* __restrict__ dy, float* __restrict__ out, int B, int Ho, int Wo, int C4,
                                   int sh, int sw, int extra_h, int extra_w) {
  const int Hd = (Ho - 1) * sh + 1 + extra_h, Wd = (Wo - 1) * sw + 1 + extra_w;
  const long n = (long)B * Hd * Wd * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int x = (int)(r % Wd);
    r /= Wd;
    const int y = (int)(r % Hd);
    const int b = (int)(r / Hd);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y % sh == 0 && x % sw == 0 && y / sh < Ho && x / sw < Wo) v = reinterpret_cast<const f32x4*>(dy)[(((long)b * Ho + y / sh) * Wo + x / sw) * C4 + c4];
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

// ---- BatchNorm2d (training) backward, ReLU mask fused -----------------------------------------------------
// g = dz * (z > 0 when relu);  xhat = (y - mean) * invstd;  partial sums of g and g*xhat per channel.
// thread -> (column quad, row lane); block covers rows_per_block rows; part[blk][2][C].
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                            const float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ part,
                                                            long rows, int C, int relu, int rows_per_block,
                                                            const unsigned char* __restrict__ zmask) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [lanes][2][C]
  const int C4 = C >> 2;
  const int cq = threadIdx.x % C4, rl = threadIdx.x / C4, lanes = 256 / C4;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cq];
  const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[cq];
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  for (long r = r0 + rl; r < r1; r += lanes) {
    f32x4 g = reinterpret_cast<const f32x4*>(dz)[r * C4 + cq];
    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[r * C4 + cq];
    if (relu) {
      if (zmask) {                       // 4 bits per quad from the forward pass (mrn_scale_shift_act_f32 pos_mask) instead of z itself
        const unsigned m = zmask[r * C4 + cq];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!((m >> j) & 1u)) g[j] = 0.f;
      } else {
        const f32x4 zv = reinterpret_cast<const f32x4*>(z)[r * C4 + cq];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(zv[j] > 0.f)) g[j] = 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1[j] += g[j];
      s2[j] += g[j] * (yv[j] - mu[j]) * is[j];
    }
  }
  reinterpret_cast<f32x4*>(red + (rl * 2 + 0) * C)[cq] = s1;
  reinterpret_cast<f32x4*>(red + (rl * 2 + 1) * C)[cq] = s2;
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int which = c / C, cc = c - which * C;
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[(l * 2 + which) * C + cc];
    part[((long)blockIdx.x * 2 + which) * C + cc] = s;
  }
}

// dy = gamma*invstd*(g - sum_g/N - xhat*sum_gx/N);  dres = g (masked upstream gradient, for the residual branch)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sums,  // [2][C]: sum g, sum g*xhat
                                                           float* __restrict__ dy, float* __restrict__ dres, long rows, int C,
                                                           int relu, float inv_n, unsigned* __restrict__ amax_ws,
                                                           const unsigned char* __restrict__ zmask) {
  const int C4 = C >> 2;
  const long n4 = rows * C4;
  float mx = 0.f;                                      // max |dy| of this lane (amax_ws: the range scale of the data / weight gradients)
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int cq = (int)(i % C4);
    f32x4 g = reinterpret_cast<const f32x4*>(dz)[i];
    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
    if (relu) {
      if (zmask) {
        const unsigned m = zmask[i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!((m >> j) & 1u)) g[j] = 0.f;
      } else {
        const f32x4 zv = reinterpret_cast<const f32x4*>(z)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(zv[j] > 0.f)) g[j] = 0.f;
      }
    }
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cq];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[cq];
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[cq];
    const f32x4 a1 = reinterpret_cast<const f32x4*>(sums)[cq];
    const f32x4 a2 = reinterpret_cast<const f32x4*>(sums + C)[cq];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (yv[j] - mu[j]) * is[j];
      o[j] = ga[j] * is[j] * (g[j] - a1[j] * inv_n - xh * a2[j] * inv_n);
    }
    reinterpret_cast<f32x4*>(dy)[i] = o;
    if (dres) reinterpret_cast<f32x4*>(dres)[i] = g;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (amax_ws) {                       // one atomic per BLOCK, spread over 64 slots (thousands of same-address atomics would serialise)
    __shared__ float wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      if (!(mx <= 0.f)) atomicMax(amax_ws + (blockIdx.x & 63), __float_as_uint(mx));
    }      // (NaN / inf pass through, as mrn_pow2_scale_f32)
  }
}

// dx[argmax window position] += dy ; first maximum in (ky, kx) scan order, as torch.  dx must be zero-initialised.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo,
                                                          int kh, int kw, int sh, int sw, int ph, int pw) {
  const long n = (long)B * Ho * Wo * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long r = i / C;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float best = -INFINITY;
    long bi = -1;
    for (int ky = 0; ky < kh; ++ky) {
      const int iy = oy * sh - ph + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int ix = ox * sw - pw + kx;
        if (ix < 0 || ix >= W) continue;
        const long idx = (((long)b * H + iy) * W + ix) * C + c;
        const float v = x[idx];
        if (v > best || bi < 0) { best = v; bi = idx; }
      }
    }
    if (bi >= 0) atomicAdd(dx + bi, dy[i]);
  }
}

// Non-overlapping windows that tile the map exactly (kernel == stride, no padding, H % kh == 0, W % kw == 0: the 2 x 2 / 2 pools of the
// ResNet and TPS stacks): every input element belongs to exactly one window, so a thread owning (window, 4 channels) WRITES its kh x kw
// input positions -- dy at the first maximum, 0 elsewhere -- and neither the zero fill of dx nor atomics are needed.
__global__ __launch_bounds__(256) void maxpool_bwd_disjoint_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   float* __restrict__ dx, int B, int H, int W, int C, int kh, int kw) {
  const int Ho = H / kh, Wo = W / kw, C4 = C >> 2;
  const long n = (long)B * Ho * Wo * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    const long base = (((long)b * H + (long)oy * kh) * W + (long)ox * kw) * C + c4 * 4;
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int at[4] = {0, 0, 0, 0};
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + base + ((long)ky * W + kx) * C);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[j] > best[j] || (ky == 0 && kx == 0)) { best[j] = v[j]; at[j] = ky * kw + kx; }     // first maximum in scan order, as torch
      }
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) {
        const int k = ky * kw + kx;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = at[j] == k ? g[j] : 0.f;
        *reinterpret_cast<f32x4*>(dx + base + ((long)ky * W + kx) * C) = o;
      }
  }
}

// ---- LSTM backward through time ------------------------------------------------------------------------
constexpr int HID = 256, BT = 16, NW = 16, NTH = NW * 64;
constexpr int GLD = 4 * HID + 4;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// gates [B][T][ndir][4H] (post-activation i,f,g,o), cseq [B][T][ndir][H], dout [B][T][ndir*H]
// w_hhT: fragment-major W_hh^T (rows = hidden unit j, K = 4H gate index), one gate group
// dgates [B][T][ndir][4H]: gradient with respect to the gate pre-activations
__global__ __launch_bounds__(NTH) void lstm_layer_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ gates,
                                                             const float* __restrict__ cseq, const float* __restrict__ w_hhT,
                                                             float* __restrict__ dgates, int B, int T, int ndir) {
  extern __shared__ __attribute__((aligned(16))) float dg_lds[];   // [BT][GLD]
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * BT;
  const int t_ = threadIdx.x, lane = t_ & 63, wave = t_ >> 6;
  const int col = lane & 15, rbase = (lane >> 4) * 4;
  const int j = wave * 16 + col;
  const int Q = 4 * HID / 16;
  const f32x4* wp = reinterpret_cast<const f32x4*>(w_hhT) + (long)dir * (HID * 4 * HID / 4) + (long)wave * Q * 64 + lane;
  float dh_rec[4] = {0.f, 0.f, 0.f, 0.f}, dc_next[4] = {0.f, 0.f, 0.f, 0.f};

  for (int step = T - 1; step >= 0; --step) {
    const int t = dir == 0 ? step : T - 1 - step;           // time index processed at forward step `step`
    const int tp = dir == 0 ? t - 1 : t + 1;                // previous time in the direction's order
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + r, b = b0 + row;
      float di = 0.f, df = 0.f, dg = 0.f, dob = 0.f;
      if (b < B) {
        const long base = ((long)b * T + t) * ndir + dir;
        const float* gp = gates + base * 4 * HID + j;
        const float ig = gp[0], fg = gp[HID], gg = gp[2 * HID], og = gp[3 * HID];
        const float ct = cseq[base * HID + j];
        const float cp = step > 0 ? cseq[(((long)b * T + tp) * ndir + dir) * HID + j] : 0.f;
        const float dh = dout[((long)b * T + t) * (ndir * HID) + dir * HID + j] + dh_rec[r];
        const float tc = tanhf(ct);
        dob = dh * tc * og * (1.f - og);
        const float dc = dc_next[r] + dh * og * (1.f - tc * tc);
        di = dc * gg * ig * (1.f - ig);
        df = dc * cp * fg * (1.f - fg);
        dg = dc * ig * (1.f - gg * gg);
        dc_next[r] = dc * fg;
        float* dp = dgates + base * 4 * HID + j;
        dp[0] = di; dp[HID] = df; dp[2 * HID] = dg; dp[3 * HID] = dob;
      }
      float* l = dg_lds + row * GLD + j;
      l[0] = di; l[HID] = df; l[2 * HID] = dg; l[3 * HID] = dob;
    }
    __syncthreads();
    if (step > 0) {
      // dh_rec[b][j] = sum_n dgate[b][n] * W_hh[n][j]
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* ap = dg_lds + col * GLD + (lane >> 4) * 4;
      f32x4 wv = wp[0];
#pragma unroll 1
      for (int q = 0; q < Q; ++q) {
        const f32x4 wn = wp[(long)((q + 1 < Q) ? q + 1 : q) * 64];
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + q * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma4(av[r], wv[r], acc);
        wv = wn;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh_rec[r] = acc[r];
    }
    __syncthreads();
  }
}

// ---- the same backward pass with the recurrent product dh_rec = dgates . W_hh as split-fp16 x3 on v_mfma_f32_16x16x32_f16: the gate
// gradients are written to LDS as two fp16 planes of gscale * dgate (gscale: a power of two from max|dout|, the gradients are 1e-3 ..
// 1e-8-sized), W_hh^T comes as the fragment-major hi / lo stream of ops.pack_fragment_major_h with its own prescale.  96 MFMAs of 16
// cycles per wave and step instead of 256 of 32 on the exact-fp32 pipe: the step was bound by them (4 waves per SIMD).
typedef _Float16 f16v8h __attribute__((ext_vector_type(8)));
constexpr int GLDH = 4 * HID + 8;     // fp16 LDS row (halves)

__global__ __launch_bounds__(NTH) void lstm_layer_bwd_x3_kernel(const float* __restrict__ dout, const float* __restrict__ gates,
                                                                const float* __restrict__ cseq, const unsigned char* __restrict__ w_hhT,
                                                                const float* __restrict__ w_inv, const float* __restrict__ gscale,
                                                                float* __restrict__ dgates, int B, int T, int ndir) {
  extern __shared__ __attribute__((aligned(16))) _Float16 dgh_lds[];   // [2 planes][BT][GLDH]
  _Float16* dg_hi = dgh_lds;
  _Float16* dg_lo = dgh_lds + BT * GLDH;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * BT;
  const int t_ = threadIdx.x, lane = t_ & 63, wave = t_ >> 6;
  const int col = lane & 15, rbase = (lane >> 4) * 4;
  const int j = wave * 16 + col;
  constexpr int Q = 4 * HID / 32;
  const f16v8h* wp = reinterpret_cast<const f16v8h*>(w_hhT + (long)dir * HID * 4 * HID * 4) + ((long)wave * Q * 64 + lane) * 2;
  const float gs = gscale[0];
  const float unscale = w_inv[dir] * gscale[1];
  float dh_rec[4] = {0.f, 0.f, 0.f, 0.f}, dc_next[4] = {0.f, 0.f, 0.f, 0.f};

  for (int step = T - 1; step >= 0; --step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + r, b = b0 + row;
      float di = 0.f, df = 0.f, dg = 0.f, dob = 0.f;
      if (b < B) {
        const long base = ((long)b * T + t) * ndir + dir;
        const float* gp = gates + base * 4 * HID + j;
        const float ig = gp[0], fg = gp[HID], gg = gp[2 * HID], og = gp[3 * HID];
        const float ct = cseq[base * HID + j];
        const float cp = step > 0 ? cseq[(((long)b * T + tp) * ndir + dir) * HID + j] : 0.f;
        const float dh = dout[((long)b * T + t) * (ndir * HID) + dir * HID + j] + dh_rec[r];
        const float tc = tanhf(ct);
        dob = dh * tc * og * (1.f - og);
        const float dc = dc_next[r] + dh * og * (1.f - tc * tc);
        di = dc * gg * ig * (1.f - ig);
        df = dc * cp * fg * (1.f - fg);
        dg = dc * ig * (1.f - gg * gg);
        dc_next[r] = dc * fg;
        float* dp = dgates + base * 4 * HID + j;
        dp[0] = di; dp[HID] = df; dp[2 * HID] = dg; dp[3 * HID] = dob;
      }
      const float v4[4] = {di * gs, df * gs, dg * gs, dob * gs};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        _Float16 hh, ll;
        split_f16_sat(v4[g], hh, ll);
        dg_hi[row * GLDH + g * HID + j] = hh;
        dg_lo[row * GLDH + g * HID + j] = ll;
      }
    }
    __syncthreads();
    if (step > 0) {
      // dh_rec[b][j] = sum_n dgate[b][n] * W_hh[n][j]
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int n = lane & 15, kg = lane >> 4;
      const _Float16* ah = dg_hi + n * GLDH + kg * 8;
      const _Float16* al = dg_lo + n * GLDH + kg * 8;
      f16v8h wh = wp[0], wl = wp[1];
#pragma unroll 1
      for (int q = 0; q < Q; ++q) {
        const int qn = (q + 1 < Q) ? q + 1 : q;
        const f16v8h nh = wp[(long)qn * 128], nl = wp[(long)qn * 128 + 1];
        const f16v8h xh = *reinterpret_cast<const f16v8h*>(ah + q * 32), xl = *reinterpret_cast<const f16v8h*>(al + q * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh, acc, 0, 0, 0);
        wh = nh;
        wl = nl;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh_rec[r] = acc[r] * unscale;
    }
    __syncthreads();
  }
}

}  // namespace

static inline int ew_grid(long n, int per_block) {
  long g = (n + per_block - 1) / per_block;
  if (g > 256 * 16) g = 256 * 16;
  return g < 1 ? 1 : (int)g;
}

MRN_EXPORT int mrn_pack_dgrad_weight_f32(const float* w_ohwi, float* wt_ihwo, int O, int I, int kh, int kw, void* stream) {
  MRN_CHECK_ARG(w_ohwi && wt_ihwo, "mrn_pack_dgrad_weight_f32: null operand");
  hipLaunchKernelGGL(pack_dgrad_weight_kernel, dim3(ew_grid((long)O * I * kh * kw, 256)), dim3(256), 0, (hipStream_t)stream,
                     w_ohwi, wt_ihwo, O, I, kh, kw);
  MRN_LAUNCH_CHECK("pack_dgrad_weight");
  return MRN_OK;
}

MRN_EXPORT int mrn_unpack_conv_weight_f32(const float* g_ohwi, float* g_oihw, int O, int I, int kh, int kw, int accumulate,
                                          void* stream) {
  MRN_CHECK_ARG(g_ohwi && g_oihw, "mrn_unpack_conv_weight_f32: null operand");
  hipLaunchKernelGGL(unpack_ohwi_oihw_kernel, dim3(ew_grid((long)O * I * kh * kw, 256)), dim3(256), 0, (hipStream_t)stream,
                     g_ohwi, g_oihw, O, I, kh * kw, accumulate);
  MRN_LAUNCH_CHECK("unpack_conv_weight");
  return MRN_OK;
}

MRN_EXPORT int mrn_dilate_nhwc_f32(const float* dy, float* out, int B, int Ho, int Wo, int C, int sh, int sw, int extra_h,
                                   int extra_w, void* stream) {
  MRN_CHECK_ARG(dy && out && C % 4 == 0 && sh >= 1 && sw >= 1 && extra_h >= 0 && extra_w >= 0, "mrn_dilate_nhwc_f32: bad operands");
  const long n = (long)B * ((Ho - 1) * sh + 1 + extra_h) * ((Wo - 1) * sw + 1 + extra_w) * (C / 4);
  if (n <= 0) return MRN_OK;
  hipLaunchKernelGGL(dilate_nhwc_kernel, dim3(ew_grid(n, 512)), dim3(256), 0, (hipStream_t)stream, dy, out, B, Ho, Wo, C / 4,
                     sh, sw, extra_h, extra_w);
  MRN_LAUNCH_CHECK("dilate_nhwc");
  return MRN_OK;
}

MRN_EXPORT int64_t mrn_bn_bwd_blocks(int64_t rows) {
  int64_t b = (rows + 255) / 256;
  if (b > 2048) b = 2048;
  return b < 1 ? 1 : b;
}

// partials: mrn_bn_bwd_blocks(rows) * 2 * C floats; the ReLU mask comes from z (z > 0) or from zmask (4 bits per 4 channels, bit j = element j > 0)
MRN_EXPORT int mrn_bn_bwd_reduce_f32(const float* dz, const float* z, const void* zmask, const float* y, const float* mean,
                                     const float* invstd, float* partials, int64_t rows, int C, int relu, void* stream) {
  MRN_CHECK_ARG(dz && y && mean && invstd && partials && (!relu || z || zmask), "mrn_bn_bwd_reduce_f32: null operand");
  MRN_CHECK_ARG(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0, "mrn_bn_bwd_reduce_f32: unsupported C=%d", C);
  if (rows == 0) return MRN_OK;
  const long nblk = mrn_bn_bwd_blocks(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  const int lanes = 256 / (C / 4);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)nblk), dim3(256), sizeof(float) * lanes * 2 * C, (hipStream_t)stream,
                     dz, z, y, mean, invstd, partials, (long)rows, C, relu, rpb, (const unsigned char*)zmask);
  MRN_LAUNCH_CHECK("bn_bwd_reduce");
  return MRN_OK;
}

// sums [2][C] = column sums of the partials [nblk][2][C] (sum g | sum g * xhat: what the apply pass needs); with dgamma_acc / dbeta_acc the
// two parameter gradients are ADDED there in the same launch (dbeta += sum g, dgamma += sum g * xhat) -- the flat-gradient slices of the
// BatchNorm weight / bias -- instead of two reduction launches and two accumulation launches
// (a block = 32 columns x 32 row lanes, two independent loads in flight per lane: the partials are few -- up to 2048 rows -- and the pass is
// latency-bound; it sits on the backward chain between the reduce and the apply pass)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ sums,
                                                               float* __restrict__ dgamma_acc, float* __restrict__ dbeta_acc) {
  __shared__ float red[32][33];
  const int col = blockIdx.x * 32 + (threadIdx.x & 31), lane = threadIdx.x >> 5;
  float s0 = 0.f, s1 = 0.f;
  if (col < 2 * C) {
    int r = lane;
    for (; r + 32 < nblk; r += 64) {
      s0 += part[(long)r * 2 * C + col];
      s1 += part[(long)(r + 32) * 2 * C + col];
    }
    if (r < nblk) s0 += part[(long)r * 2 * C + col];
  }
  red[lane][threadIdx.x & 31] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < 32 && col < 2 * C) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 32; ++l) t += red[l][threadIdx.x];
    sums[col] = t;
    if (col < C) {
      if (dbeta_acc) dbeta_acc[col] += t;
    } else if (dgamma_acc) {
      dgamma_acc[col - C] += t;
    }
  }
}

MRN_EXPORT int mrn_bn_bwd_finalize_f32(const float* partials, int64_t nblk, int C, float* sums, float* dgamma_acc, float* dbeta_acc,
                                       void* stream) {
  MRN_CHECK_ARG(partials && sums && nblk >= 1 && C >= 1, "mrn_bn_bwd_finalize_f32: bad operands");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)((2 * C + 31) / 32)), dim3(1024), 0, (hipStream_t)stream, partials, (int)nblk, C,
                     sums, dgamma_acc, dbeta_acc);
  MRN_LAUNCH_CHECK("bn_bwd_finalize");
  return MRN_OK;
}

MRN_EXPORT int mrn_bn_bwd_apply_f32(const float* dz, const float* z, const void* zmask, const float* y, const float* mean,
                                    const float* invstd, const float* gamma, const float* sums, float* dy, float* dres, int64_t rows,
                                    int C, int relu, void* amax_ws, void* stream) {
  MRN_CHECK_ARG(dz && y && mean && invstd && gamma && sums && dy && (!relu || z || zmask) && C % 4 == 0, "mrn_bn_bwd_apply_f32: bad operands");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(rows * (C / 4), 1024)), dim3(256), 0, (hipStream_t)stream, dz, z, y, mean,
                     invstd, gamma, sums, dy, dres, (long)rows, C, relu, 1.f / (float)rows, (unsigned*)amax_ws, (const unsigned char*)zmask);
  MRN_LAUNCH_CHECK("bn_bwd_apply");
  return MRN_OK;
}

// 1 when mrn_maxpool_bwd_nhwc_f32 writes EVERY element of dx itself (non-overlapping windows tiling the map): the caller may skip the zero fill
MRN_EXPORT int64_t mrn_maxpool_bwd_writes_all(int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw) {
  return kh == sh && kw == sw && ph == 0 && pw == 0 && H % kh == 0 && W % kw == 0 && C % 4 == 0;
}

MRN_EXPORT int mrn_maxpool_bwd_nhwc_f32(const float* dy, const float* x, float* dx_zeroed, int B, int H, int W, int C, int kh,
                                        int kw, int sh, int sw, int ph, int pw, void* stream) {
  MRN_CHECK_ARG(dy && x && dx_zeroed, "mrn_maxpool_bwd_nhwc_f32: null operand");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  const long n = (long)B * Ho * Wo * C;
  if (n <= 0) return MRN_OK;
  if (mrn_maxpool_bwd_writes_all(H, W, C, kh, kw, sh, sw, ph, pw) && (uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dx_zeroed % 16 == 0) {
    hipLaunchKernelGGL(maxpool_bwd_disjoint_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx_zeroed, B, H, W,
                       C, kh, kw);
    MRN_LAUNCH_CHECK("maxpool_bwd");
    return MRN_OK;
  }
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(n, 512)), dim3(256), 0, (hipStream_t)stream, dy, x, dx_zeroed, B, H, W, C,
                     Ho, Wo, kh, kw, sh, sw, ph, pw);
  MRN_LAUNCH_CHECK("maxpool_bwd");
  return MRN_OK;
}

// mrn_lstm_layer_bwd_f32 with the recurrent product as split-fp16 x3: w_hhT_h = per direction the fragment-major fp16 hi / lo stream of
// W_hh^T [H][4H] (ops.pack_fragment_major_h), w_inv device float[ndir] = 1 / its prescale, gscale device float[2] = {s, 1/s} with a power
// of two s that brings max|dout| to ~16 (mrn_pow2_scale_f32: the gate gradients are split as s * dgate; fp16 keeps a factor 4096 of headroom)
MRN_EXPORT int mrn_lstm_layer_bwd_x3(const float* dout, const float* gates, const float* cseq, const void* w_hhT_h, const float* w_inv,
                                     const float* gscale, float* dgates, int B, int T, int hidden, int ndir, void* stream) {
  MRN_CHECK_ARG(dout && gates && cseq && w_hhT_h && w_inv && gscale && dgates, "mrn_lstm_layer_bwd_x3: null operand");
  MRN_CHECK_ARG(hidden == HID && (ndir == 1 || ndir == 2), "mrn_lstm_layer_bwd_x3: hidden=%d ndir=%d unsupported", hidden, ndir);
  if (B == 0 || T == 0) return MRN_OK;
  const size_t lds = sizeof(_Float16) * 2 * BT * GLDH;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)lstm_layer_bwd_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(lstm_layer_bwd_x3_kernel, dim3(ceil_div(B, BT), ndir), dim3(NTH), lds, (hipStream_t)stream, dout, gates, cseq,
                     (const unsigned char*)w_hhT_h, w_inv, gscale, dgates, B, T, ndir);
  MRN_LAUNCH_CHECK("lstm_layer_bwd_x3");
  return MRN_OK;
}

MRN_EXPORT int mrn_lstm_layer_bwd_f32(const float* dout, const float* gates, const float* cseq, const float* w_hhT,
                                      float* dgates, int B, int T, int hidden, int ndir, void* stream) {
  MRN_CHECK_ARG(dout && gates && cseq && w_hhT && dgates, "mrn_lstm_layer_bwd_f32: null operand");
  MRN_CHECK_ARG(hidden == HID && (ndir == 1 || ndir == 2), "mrn_lstm_layer_bwd_f32: hidden=%d ndir=%d unsupported", hidden, ndir);
  if (B == 0 || T == 0) return MRN_OK;
  const size_t lds = sizeof(float) * BT * GLD;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)lstm_layer_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(lstm_layer_bwd_kernel, dim3(ceil_div(B, BT), ndir), dim3(NTH), lds, (hipStream_t)stream, dout, gates, cseq,
                     w_hhT, dgates, B, T, ndir);
  MRN_LAUNCH_CHECK("lstm_layer_bwd");
  return MRN_OK;
}
