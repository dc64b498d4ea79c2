// First convolution of the frozen experts' stacks: 3x3, stride 1, padding 1 on the Cin = 4 crops (VGG conv 0
// feature_extraction.py:19, ResNet conv0_1 :214, TPS localisation conv 1 transformation.py:60), G experts in one launch.
//
// K = 36 is too short for the staged implicit-GEMM kernels (gemm.hip ran this shape at 0.75 TB/s: three 16-deep K-steps of
// 16-byte gathers per tile).  Here one workgroup owns TWO image rows x 128 output pixels: the 4 x 130 x 4 input patch (8 KB) and the
// expert's whole weight matrix [36][Cout] sit in LDS, a wave computes 32 columns x 2 rows x Cout on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: 18 steps per 32 output channels) with the PIXELS as the A operand: a lane then holds ONE channel of 16
// pixels per row, so a store instruction writes full 128-byte lines (32 consecutive channels of a pixel per half-wave; the other
// orientation -- 16 bytes per lane, 32-byte pieces of 32 different lines per instruction -- measured 1.59 ms against 1.3 on the
// 64-channel layer: partial-line writes), the BatchNorm partial sums are in-register sums over the lane's pixels, and a 2x2 window
// is four registers of one lane.  The epilogue adds the bias, accumulates the partial statistics and either writes the full map or
// POOLS: the 2x2 / stride-2 max-pool that follows BatchNorm + ReLU (VGG, TPS) commutes with them up to the sign of the BatchNorm
// weight, so the kernel keeps, per window and channel, the maximum (weight >= 0) or the minimum (weight < 0) of the raw output -- a
// quarter of the bytes of a kernel that is bound by its output write (see conv_patch.hip for the argument and the bit-exactness test).
#include "common.hpp"

namespace {

constexpr int PX = 128;          // output pixels per workgroup and row (one wave = 32 columns x 2 rows)
constexpr int PW = PX + 2;       // patch width
constexpr int CIN = 4;
constexpr int KTOT = 36;         // 9 taps x 4 channels, k = tap * 4 + c

struct ConvFirstParams {
  const float* x;      // [Gx][B][H][W][4]  (x_gstride = 0: one shared input)
  const float* w;      // [G][Cout][3][3][4]  (OHWI)
  const float* bias;   // [G][Cout] or null
  const long long* gamma;   // [G] device addresses of the BatchNorm weights that follow (pooled form), or null: all maxima
  float* y;            // [G][B][H][W][Cout], pooled form [G][B][H/2][W/2][Cout]
  float* stats;        // [G][nblk][2][Cout] per-workgroup sums / sums of squares (before the activation), or null
  long x_gstride;      // floats between the groups' inputs (0 = shared)
  int G, B, H, W, Cout, act, tiles_w, tiles_h, nblk;
};

template <int NT, bool POOL>     // NT = Cout / 32 (1 or 2)
__global__ __launch_bounds__(256) void conv_first_kernel(const ConvFirstParams p) {
  __shared__ __attribute__((aligned(16))) float patch[4 * PW * CIN];
  __shared__ float wl[KTOT * NT * 32];            // [k][cout]
  __shared__ float red[4 * 2 * NT * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n31 = lane & 31, kg = lane >> 5;
  int bid = blockIdx.x;
  const int tw = bid % p.tiles_w;
  bid /= p.tiles_w;
  const int th = bid % p.tiles_h;
  bid /= p.tiles_h;
  const int b = bid % p.B;
  const int g = bid / p.B;
  const int x0 = tw * PX, oy0 = th * 2;
  constexpr int Cout = NT * 32;

  // ---- stage the input patch (rows oy0-1..oy0+2, columns x0-1..x0+PX) and the weights
  const float* xg = p.x + (long)g * p.x_gstride + (long)b * p.H * p.W * CIN;
  for (int i = tid; i < 4 * PW; i += 256) {
    const int r = i / PW, c = i - r * PW;
    const int iy = oy0 - 1 + r, ix = x0 - 1 + c;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = *reinterpret_cast<const f32x4*>(xg + ((long)iy * p.W + ix) * CIN);
    *reinterpret_cast<f32x4*>(&patch[i * CIN]) = v;
  }
  const float* wg = p.w + (long)g * Cout * KTOT;
  for (int i = tid; i < KTOT * Cout; i += 256) {
    const int co = i / KTOT, k = i - co * KTOT;
    wl[k * Cout + co] = wg[i];
  }
  __syncthreads();

  // ---- 32 pixels x 2 rows x Cout per wave: A lane (pixel n31, kg) = patch element k = 2 s + kg, B lane (cout n31, kg) = w[k][cout]
  f32x16 acc[2][NT];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][j][e] = 0.f;
  const int px = wave * 32 + n31;                 // pixel column inside the tile
#pragma unroll
  for (int s = 0; s < KTOT / 2; ++s) {
    const int tap = s >> 1, ky = tap / 3, kx = tap - ky * 3;      // (compile-time; + kg per lane: channel 2 (s & 1) + kg)
    float wv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wv[j] = wl[(2 * s + kg) * Cout + j * 32 + n31];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float a = patch[((r + ky) * PW + px + kx) * CIN + 2 * (s & 1) + kg];
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wv[j], acc[r][j], 0, 0, 0);
    }
  }

  // ---- epilogue: register e of lane (cout n31, kg) is pixel column x0 + wave * 32 + (e & 3) + 8 (e >> 2) + 4 kg of rows oy0, oy0 + 1
  const float relu_floor = p.act == 1 ? 0.f : -INFINITY;
  const int Ho = p.H >> 1, Wo = p.W >> 1;
  float* const yg = p.y + (long)g * p.B * (POOL ? Ho * (long)Wo : p.H * (long)p.W) * Cout;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = j * 32 + n31;
    const float bv = p.bias ? p.bias[(long)g * Cout + co] : 0.f;
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bool rok = oy0 + r < p.H;
      float* yrow = yg + (((long)b * p.H + oy0 + r) * p.W) * Cout;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ox = x0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
        const float v = acc[r][j][e] + bv;
        acc[r][j][e] = v;
        if (rok && ox < p.W) {
          sm += v;
          sq = fmaf(v, v, sq);
          if (!POOL) yrow[(long)ox * Cout + co] = relu_nan(v, relu_floor);
        }
      }
    }
    if (POOL) {
      // 2 x 2 windows: registers (e, e + 1) of the two rows, e even; the minimum is -max(-v): exact
      const float sg = (p.gamma && reinterpret_cast<const float*>(p.gamma[g])[co] < 0.f) ? -1.f : 1.f;
      const int py = oy0 >> 1;
      float* prow = yg + (((long)b * Ho + py) * Wo) * Cout;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const int pxo = (x0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg) >> 1;
        const float mx = max_nan(max_nan(acc[0][j][e] * sg, acc[0][j][e + 1] * sg), max_nan(acc[1][j][e] * sg, acc[1][j][e + 1] * sg));
        if (py < Ho && pxo < Wo) prow[(long)pxo * Cout + co] = relu_nan(mx * sg, relu_floor);
      }
    }
    if (p.stats) {
      sm += __shfl_xor(sm, 32);
      sq += __shfl_xor(sq, 32);
      if (kg == 0) {
        red[(wave * 2 + 0) * Cout + co] = sm;
        red[(wave * 2 + 1) * Cout + co] = sq;
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    if (tid < 2 * Cout) {
      const int which = tid / Cout, co = tid - which * Cout;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[(w * 2 + which) * Cout + co];
      const long blk = ((long)b * p.tiles_h + th) * p.tiles_w + tw;      // (tile-indexed: the same bits for every grouping of the experts)
      p.stats[(((long)g * p.nblk + blk) * 2 + which) * Cout + co] = v;
    }
  }
}

}  // namespace

// per-group partial statistics blocks of mrn_conv3x3_c4_grouped_f32 (one per workgroup = 2 rows x 128 pixels)
MRN_EXPORT int64_t mrn_conv3x3_c4_stats_blocks(int B, int H, int W) { return (int64_t)B * ((H + 1) / 2) * ((W + PX - 1) / PX); }

// y[g] = act(conv3x3(x[g or shared], w[g]) + bias[g]), stride 1, padding 1, Cin = 4, Cout = 32 or 64; NHWC fp32.
// x_group_stride (floats; 0 = every group reads the same input); stats [G][blocks][2][Cout] or NULL; act: 0 none, 1 ReLU.
// pool = 1 (even H, W): y is [G][B][H/2][W/2][Cout] and holds, per 2 x 2 window and channel, the window's maximum where the
// BatchNorm weight that follows (bn_gamma_ptrs: device table of G device addresses; NULL: all maxima) is >= 0 and its minimum
// elsewhere -- see mrn_conv3x3_patch_x3_hl32; the statistics cover the full map.
MRN_EXPORT int mrn_conv3x3_c4_grouped_f32(const float* x, const float* w_ohwi, const float* bias, float* y, float* stats, int G,
                                          int64_t x_group_stride, int B, int H, int W, int Cout, int act, int pool,
                                          const void* bn_gamma_ptrs, void* stream) {
  MRN_CHECK_ARG(x && w_ohwi && y && G >= 1 && (Cout == 32 || Cout == 64), "mrn_conv3x3_c4_grouped_f32: bad operands (Cout=%d)", Cout);
  MRN_CHECK_ARG(((uintptr_t)x % 16 == 0) && x_group_stride % 4 == 0 && (uintptr_t)y % 16 == 0 && (!bias || (uintptr_t)bias % 16 == 0),
                "mrn_conv3x3_c4_grouped_f32: input / output / bias must be 16-byte aligned");
  MRN_CHECK_ARG(!pool || (H % 2 == 0 && W % 2 == 0), "mrn_conv3x3_c4_grouped_f32: the pooled form needs even H, W (%d x %d)", H, W);
  if (B == 0 || H == 0 || W == 0) return MRN_OK;
  ConvFirstParams p;
  p.x = x; p.w = w_ohwi; p.bias = bias; p.gamma = (const long long*)bn_gamma_ptrs; p.y = y; p.stats = stats; p.x_gstride = x_group_stride;
  p.G = G; p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.act = act;
  p.tiles_w = (W + PX - 1) / PX;
  p.tiles_h = (H + 1) / 2;
  p.nblk = (int)mrn_conv3x3_c4_stats_blocks(B, H, W);
  const dim3 grid((unsigned)((long)G * B * p.tiles_h * p.tiles_w));
  const hipStream_t st = (hipStream_t)stream;
  if (Cout == 32) {
    if (pool) hipLaunchKernelGGL((conv_first_kernel<1, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_first_kernel<1, false>), grid, dim3(256), 0, st, p);
  } else {
    if (pool) hipLaunchKernelGGL((conv_first_kernel<2, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_first_kernel<2, false>), grid, dim3(256), 0, st, p);
  }
  MRN_LAUNCH_CHECK("conv3x3_c4_grouped");
  return MRN_OK;
}
