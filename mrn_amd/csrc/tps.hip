// TPS rectification: grid generation + bilinear sampling fused in one pass.
//
// Reference: modules/transformation.py:204-216 (build_P_prime: T = inv_delta_C . [C'; 0], P' = P_hat . T) and
// :38-44 (F.grid_sample, padding_mode="border", align_corners=True).  The reference materialises
// inv_delta_C and P_hat per batch element (repeat), two bmm's, the [B, n, 2] grid, then samples; here each
// block derives the 23x2 transform of its image in LDS, evaluates the grid point of a pixel in registers
// and gathers the four neighbours as 16-byte NHWC pixels.  Output is NHWC so the first conv reads it directly.
#include "common.hpp"

namespace {

constexpr int MAXF = 64;  // fiducials + 3

__global__ __launch_bounds__(256) void tps_sample_kernel(const float* __restrict__ img,       // [B][H][W][4]
                                                         const float* __restrict__ cprime,    // [B][F][2]
                                                         const float* __restrict__ inv_delta, // [F+3][F+3]
                                                         const float* __restrict__ p_hat,     // [Hr*Wr][F+3]
                                                         float* __restrict__ out,             // [B][Hr][Wr][4]
                                                         float* __restrict__ grid_out,        // optional [B][Hr*Wr][2]
                                                         int H, int W, int Hr, int Wr, int F) {
  __shared__ float T[MAXF][2];
  const int b = blockIdx.y;
  const int F3 = F + 3;
  // T[r][d] = sum_{j<F} inv_delta[r][j] * C'[b][j][d]   (the three appended rows of C' are zero)
  for (int i = threadIdx.x; i < F3 * 2; i += 256) {
    const int r = i >> 1, d = i & 1;
    float s = 0.f;
    for (int j = 0; j < F; ++j) s = fmaf(inv_delta[r * F3 + j], cprime[((long)b * F + j) * 2 + d], s);
    T[r][d] = s;
  }
  __syncthreads();
  const int n = Hr * Wr;
  const f32x4* im = reinterpret_cast<const f32x4*>(img) + (long)b * H * W;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
    const float* ph = p_hat + (long)p * F3;
    float gx = 0.f, gy = 0.f;
    for (int j = 0; j < F3; ++j) {
      const float w = ph[j];
      gx = fmaf(w, T[j][0], gx);
      gy = fmaf(w, T[j][1], gy);
    }
    if (grid_out) {
      grid_out[((long)b * n + p) * 2 + 0] = gx;
      grid_out[((long)b * n + p) * 2 + 1] = gy;
    }
    // align_corners=True unnormalisation, then border clamp
    float ix = (gx + 1.f) * 0.5f * (float)(W - 1);
    float iy = (gy + 1.f) * 0.5f * (float)(H - 1);
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1;  // weight of x1 / x0
    const float wy1 = iy - fy, wy0 = 1.f - wy1;
    const float w_nw = wx0 * wy0, w_ne = wx1 * wy0, w_sw = wx0 * wy1, w_se = wx1 * wy1;
    const bool xin1 = x1 < W, yin1 = y1 < H;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 nw = im[(long)y0 * W + x0];
    const f32x4 ne = xin1 ? im[(long)y0 * W + x1] : zero;
    const f32x4 sw = yin1 ? im[(long)y1 * W + x0] : zero;
    const f32x4 se = (xin1 && yin1) ? im[(long)y1 * W + x1] : zero;
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = nw[c] * w_nw + ne[c] * w_ne + sw[c] * w_sw + se[c] * w_se;
    reinterpret_cast<f32x4*>(out)[(long)b * n + p] = o;
  }
}

}  // namespace

MRN_EXPORT int mrn_tps_grid_sample_f32(const float* img_nhwc, const float* cprime, const float* inv_delta_c,
                                       const float* p_hat, float* out_nhwc, float* grid_out, int B, int H, int W,
                                       int C, int Hr, int Wr, int F, void* stream) {
  MRN_CHECK_ARG(img_nhwc && cprime && inv_delta_c && p_hat && out_nhwc, "mrn_tps_grid_sample_f32: null operand");
  MRN_CHECK_ARG(C == 4, "mrn_tps_grid_sample_f32: only 4-channel (RGBA) NHWC input is supported, got C=%d", C);
  MRN_CHECK_ARG(F + 3 <= MAXF && F > 0, "mrn_tps_grid_sample_f32: F=%d out of range", F);
  if (B == 0) return MRN_OK;
  dim3 grid(ceil_div(Hr * Wr, 256 * 4), B);
  hipLaunchKernelGGL(tps_sample_kernel, grid, dim3(256), 0, (hipStream_t)stream, img_nhwc, cprime, inv_delta_c, p_hat,
                     out_nhwc, grid_out, H, W, Hr, Wr, F);
  MRN_LAUNCH_CHECK("tps_grid_sample");
  return MRN_OK;
}
