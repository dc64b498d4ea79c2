// HBM-bound NHWC helpers around the conv stack: layout changes, BatchNorm statistics/apply, pooling.
//
// Reference op sites: BatchNorm2d (train-mode batch statistics + running-stat update) and ReLU after
// every ResNet / localization conv (modules/feature_extraction.py:171-197,222-294, modules/transformation.py:69-81),
// MaxPool2d variants (feature_extraction.py:22,25,30,41,234,246,260), AdaptiveAvgPool2d(1)
// (transformation.py:83).  All kernels move each byte once, 16 B per lane, channels innermost.
#include "common.hpp"

namespace {

// [B][C][H][W] -> [B][H][W][C]; one block handles a (b, 64-pixel) strip through LDS so both sides coalesce.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int C, int HW) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [C][65]
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * 64;
  const int np = min(64, HW - p0);
  const float* xb = x + (long)b * C * HW;
  for (int i = threadIdx.x; i < C * 64; i += 256) {
    const int c = i >> 6, p = i & 63;
    if (p < np) tile[c * 65 + p] = xb[(long)c * HW + p0 + p];
  }
  __syncthreads();
  float* yb = y + ((long)b * HW + p0) * C;
  for (int i = threadIdx.x; i < np * C; i += 256) {
    const int p = i / C, c = i - p * C;
    yb[i] = tile[c * 65 + p];
  }
}

// conv weight [O][I][kh][kw] -> [O][kh][kw][I]
__global__ void pack_oihw_ohwi_kernel(const float* __restrict__ w, float* __restrict__ o, int O, int I, int khw) {
  const long n = (long)O * I * khw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ci = i % I;
    const long r = i / I;
    const int tap = r % khw;
    const int oc = r / khw;
    o[i] = w[((long)oc * I + ci) * khw + tap];
  }
}

// Reduce the conv epilogue's per-row-block partials to batch mean / biased variance, fold them with the
// affine parameters into (scale, shift), and update the running statistics exactly like
// torch.nn.BatchNorm2d in training mode (momentum form, unbiased running variance).
// (a block = 32 channels x 32 row lanes: coalesced 128-byte reads of the partial rows, the sums in double)
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int nblk, int C, long count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var,
                                                           float momentum, float eps, float* __restrict__ scale,
                                                           float* __restrict__ shift, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd) {
  __shared__ double red[2][32][33];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int b = lane; b < nblk; b += 32) {
      s += (double)part[((long)b * 2 + 0) * C + c];
      q += (double)part[((long)b * 2 + 1) * C + c];
    }
  red[0][lane][cl] = s;
  red[1][lane][cl] = q;
  __syncthreads();
  if (threadIdx.x < 32 && c < C) {
    s = q = 0.0;
#pragma unroll
    for (int l = 0; l < 32; ++l) {
      s += red[0][l][cl];
      q += red[1][l][cl];
    }
    const double mean = s / (double)count;
    double var = q / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = bt - (float)mean * sc;
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = invstd;
    if (run_mean) {
      const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
  }
}

// eval-mode BatchNorm folded to (scale, shift) from the running statistics
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                      float eps, int C, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(run_var[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * invstd;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - run_mean[c] * sc;
}

// y = act(x * scale[c] + shift[c] (+ res)); C % 4 == 0, 16 B per lane, in place allowed
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                              float* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, long n4, int C4, int relu,
                                                              unsigned* __restrict__ amax_ws, unsigned char* __restrict__ pos_mask) {
  float mx = 0.f;                  // max |y| of this lane (amax_ws: the range scale of the NEXT trained convolution's operand)
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4];
    const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[c4];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] * sc[j] + sh[j];
    if (res) {
      const f32x4 r = reinterpret_cast<const f32x4*>(res)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += r[j];
    }
    if (relu == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = relu_nan(o[j]);
    } else if (relu == 2) {   // GELU (erf), SVTR PatchEmbed (modules/svtr.py:227-233)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = 0.5f * o[j] * (1.f + erff(o[j] * 0.70710678118654752440f));
    }
    reinterpret_cast<f32x4*>(y)[i] = o;
    // the ReLU mask of the backward pass (mrn_bn_bwd_*: g = dz * (y > 0)), 4 bits per lane: 1/32 of the bytes of re-reading y there
    if (pos_mask) pos_mask[i] = (unsigned char)((o[0] > 0.f) | ((o[1] > 0.f) << 1) | ((o[2] > 0.f) << 2) | ((o[3] > 0.f) << 3));
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
    }
  }
}

// NHWC max pooling, padding behaves as -inf (torch semantics); optional fused (scale, shift, relu) on the
// input so BatchNorm-apply + ReLU + pool is one pass over the conv output.
__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int relu, int B, int H, int W, int C4, int Ho, int Wo, int kh,
                                                           int kw, int sh, int sw, int ph, int pw) {
  const long n = (long)B * Ho * Wo * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
    if (scale) {
      sc = reinterpret_cast<const f32x4*>(scale)[c4];
      sf = reinterpret_cast<const f32x4*>(shift)[c4];
    }
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int ky = 0; ky < kh; ++ky) {
      const int iy = oy * sh - ph + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int ix = ox * sw - pw + kx;
        if (ix < 0 || ix >= W) continue;
        f32x4 v = reinterpret_cast<const f32x4*>(x)[(((long)b * H + iy) * W + ix) * C4 + c4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = v[j] * sc[j] + sf[j];
          if (relu) t = relu_nan(t);
          m[j] = max_nan(m[j], t);
        }
      }
    }
    reinterpret_cast<f32x4*>(y)[i] = m;
  }
}

// mean over the HW pixels of each image: [B][HW][C] -> [B][C], optional fused (scale, shift, relu)
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int relu, int HW, int C) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float sc = scale ? scale[c] : 1.f, sf = scale ? shift[c] : 0.f;
  const float* xb = x + (long)b * HW * C + c;
  float s = 0.f;
  for (int p = 0; p < HW; ++p) {
    float t = xb[(long)p * C] * sc + sf;
    if (relu) t = relu_nan(t);
    s += t;
  }
  y[(long)b * C + c] = s / (float)HW;
}

}  // namespace

static inline int ew_grid(long n, int per_block) {
  long g = (n + per_block - 1) / per_block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

MRN_EXPORT int mrn_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int H, int W, void* stream) {
  MRN_CHECK_ARG(x && y && C > 0 && C <= 512, "mrn_nchw_to_nhwc_f32: bad args (C=%d)", C);
  if (B == 0) return MRN_OK;
  const int HW = H * W;
  dim3 grid(ceil_div(HW, 64), B);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), (size_t)C * 65 * sizeof(float), (hipStream_t)stream, x, y, C, HW);
  MRN_LAUNCH_CHECK("nchw_to_nhwc");
  return MRN_OK;
}

MRN_EXPORT int mrn_pack_conv_weight_f32(const float* w_oihw, float* w_ohwi, int O, int I, int kh, int kw, void* stream) {
  MRN_CHECK_ARG(w_oihw && w_ohwi, "mrn_pack_conv_weight_f32: null");
  const long n = (long)O * I * kh * kw;
  hipLaunchKernelGGL(pack_oihw_ohwi_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, w_ohwi, O, I, kh * kw);
  MRN_LAUNCH_CHECK("pack_conv_weight");
  return MRN_OK;
}

MRN_EXPORT int mrn_bn_finalize_f32(const float* partials, int nblk, int C, int64_t count, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum,
                                   float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                                   void* stream) {
  MRN_CHECK_ARG(partials && scale && shift && C > 0 && count > 0, "mrn_bn_finalize_f32: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 32)), dim3(1024), 0, (hipStream_t)stream, partials, nblk, C,
                     (long)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean,
                     save_invstd);
  MRN_LAUNCH_CHECK("bn_finalize");
  return MRN_OK;
}

MRN_EXPORT int mrn_bn_eval_affine_f32(const float* gamma, const float* beta, const float* running_mean,
                                      const float* running_var, float eps, int C, float* scale, float* shift,
                                      void* stream) {
  MRN_CHECK_ARG(running_mean && running_var && scale && shift, "mrn_bn_eval_affine_f32: null");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  MRN_LAUNCH_CHECK("bn_eval_affine");
  return MRN_OK;
}

MRN_EXPORT int mrn_scale_shift_act_f32(const float* x, const float* residual, float* y, const float* scale,
                                       const float* shift, int64_t rows, int C, int relu, void* amax_ws, void* pos_mask, void* stream) {
  MRN_CHECK_ARG(x && y && scale && shift && C % 4 == 0, "mrn_scale_shift_act_f32: bad args (C=%d)", C);
  const long n4 = rows * (C / 4);
  if (n4 == 0) return MRN_OK;
  hipLaunchKernelGGL(scale_shift_act_kernel, dim3(ew_grid(n4, 256 * 4)), dim3(256), 0, (hipStream_t)stream, x, residual,
                     y, scale, shift, n4, C / 4, relu, (unsigned*)amax_ws, (unsigned char*)pos_mask);
  MRN_LAUNCH_CHECK("scale_shift_act");
  return MRN_OK;
}

MRN_EXPORT int mrn_maxpool_nhwc_f32(const float* x, float* y, const float* scale, const float* shift, int relu, int B,
                                    int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, void* stream) {
  MRN_CHECK_ARG(x && y && C % 4 == 0, "mrn_maxpool_nhwc_f32: bad args (C=%d)", C);
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  const long n = (long)B * Ho * Wo * (C / 4);
  if (n <= 0) return MRN_OK;
  hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3(ew_grid(n, 256 * 2)), dim3(256), 0, (hipStream_t)stream, x, y, scale, shift,
                     relu, B, H, W, C / 4, Ho, Wo, kh, kw, sh, sw, ph, pw);
  MRN_LAUNCH_CHECK("maxpool_nhwc");
  return MRN_OK;
}

MRN_EXPORT int mrn_avgpool_nhwc_f32(const float* x, float* y, const float* scale, const float* shift, int relu, int B,
                                    int HW, int C, void* stream) {
  MRN_CHECK_ARG(x && y && HW > 0, "mrn_avgpool_nhwc_f32: bad args");
  if (B == 0) return MRN_OK;
  hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3(ceil_div(C, 256), B), dim3(256), 0, (hipStream_t)stream, x, y, scale, shift,
                     relu, HW, C);
  MRN_LAUNCH_CHECK("avgpool_nhwc");
  return MRN_OK;
}
