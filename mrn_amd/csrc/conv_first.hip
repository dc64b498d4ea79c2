// First convolution of the frozen experts' stacks: 3x3, stride 1, padding 1 on the Cin = 4 crops (VGG conv 0
// feature_extraction.py:19, ResNet conv0_1 :214, TPS localisation conv 1 transformation.py:60), G experts in one launch.
//
// K = 36 is too short for the staged implicit-GEMM kernels (gemm.hip ran this shape at 0.75 TB/s: three 16-deep K-steps of
// 16-byte gathers per tile).  Here one workgroup owns 128 consecutive output pixels of one image row: the 3 x 130 x 4 input
// patch (6 KB) and the expert's whole weight matrix [36][Cout] sit in LDS, each wave computes 32 pixels x Cout on the exact
// fp32 MFMA (18 steps of v_mfma_f32_32x32x2_f32 per 32 output channels), and the epilogue adds the bias, accumulates the
// BatchNorm partial statistics and writes full 128-byte lines -- the kernel is bound by the output write.
#include "common.hpp"

namespace {

constexpr int PX = 128;          // output pixels per workgroup (one wave = 32)
constexpr int PW = PX + 2;       // patch width
constexpr int CIN = 4;
constexpr int KTOT = 36;         // 9 taps x 4 channels, k = tap * 4 + c

struct ConvFirstParams {
  const float* x;      // [Gx][B][H][W][4]  (x_gstride = 0: one shared input)
  const float* w;      // [G][Cout][3][3][4]  (OHWI)
  const float* bias;   // [G][Cout] or null
  float* y;            // [G][B][H][W][Cout]
  float* stats;        // [G][nblk][2][Cout] per-workgroup sums / sums of squares (before the activation), or null
  long x_gstride;      // floats between the groups' inputs (0 = shared)
  int G, B, H, W, Cout, act, tiles_w, nblk;
};

template <int NT>     // NT = Cout / 32 (1 or 2)
__global__ __launch_bounds__(256) void conv_first_kernel(const ConvFirstParams p) {
  __shared__ __attribute__((aligned(16))) float patch[3 * PW * CIN];
  __shared__ float wl[KTOT * NT * 32];            // [k][cout]
  __shared__ float red[4 * 2 * NT * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n31 = lane & 31, kg = lane >> 5;
  int bid = blockIdx.x;
  const int tw = bid % p.tiles_w;
  bid /= p.tiles_w;
  const int oy = bid % p.H;
  bid /= p.H;
  const int b = bid % p.B;
  const int g = bid / p.B;
  const int x0 = tw * PX;
  const int Cout = NT * 32;

  // ---- stage the input patch (rows oy-1..oy+1, columns x0-1..x0+PX) and the weights
  const float* xg = p.x + (long)g * p.x_gstride + (long)b * p.H * p.W * CIN;
  for (int i = tid; i < 3 * PW; i += 256) {
    const int r = i / PW, c = i - r * PW;
    const int iy = oy - 1 + r, ix = x0 - 1 + c;
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

  // ---- 32 pixels x Cout per wave: A lane (pixel n31, kg) = patch element k = 2s + kg, B lane (cout n31, kg) = w[k][cout]
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const int px = wave * 32 + n31;                 // pixel inside the tile
#pragma unroll
  for (int s = 0; s < KTOT / 2; ++s) {
    const int k = 2 * s + kg;                     // tap = s / 2 and channel pair 2 * (s & 1) are compile-time, + kg per lane
    const int tap = s >> 1, ky = tap / 3, kx = tap - ky * 3;
    const float a = patch[(ky * PW + px + kx) * CIN + 2 * (s & 1) + kg];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wl[k * Cout + j * 32 + n31], acc[j], 0, 0, 0);
  }

  // ---- epilogue: register e of lane (cout n31, kg) is pixel wave*32 + (e&3) + 8*(e>>2) + 4*kg
  float* yrow = p.y + (((long)g * p.B + b) * p.H + oy) * (long)p.W * Cout;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = j * 32 + n31;
    const float bv = p.bias ? p.bias[(long)g * Cout + co] : 0.f;
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ox = x0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
      if (ox < p.W) {
        float v = acc[j][e] + bv;
        sm += v;
        sq += v * v;
        if (p.act == 1) v = fmaxf(v, 0.f);
        yrow[(long)ox * Cout + co] = v;
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
      const long blk = ((long)b * p.H + oy) * p.tiles_w + tw;
      p.stats[(((long)g * p.nblk + blk) * 2 + which) * Cout + co] = v;
    }
  }
}

}  // namespace

// per-group partial statistics blocks of mrn_conv3x3_c4_grouped_f32 (one per workgroup = 128 pixels of an image row)
MRN_EXPORT int64_t mrn_conv3x3_c4_stats_blocks(int B, int H, int W) { return (int64_t)B * H * ((W + PX - 1) / PX); }

// y[g] = act(conv3x3(x[g or shared], w[g]) + bias[g]), stride 1, padding 1, Cin = 4, Cout = 32 or 64; NHWC fp32.
// x_group_stride (floats; 0 = every group reads the same input); stats [G][blocks][2][Cout] or NULL; act: 0 none, 1 ReLU.
MRN_EXPORT int mrn_conv3x3_c4_grouped_f32(const float* x, const float* w_ohwi, const float* bias, float* y, float* stats, int G,
                                          int64_t x_group_stride, int B, int H, int W, int Cout, int act, void* stream) {
  MRN_CHECK_ARG(x && w_ohwi && y && G >= 1 && (Cout == 32 || Cout == 64), "mrn_conv3x3_c4_grouped_f32: bad operands (Cout=%d)", Cout);
  MRN_CHECK_ARG(((uintptr_t)x % 16 == 0) && x_group_stride % 4 == 0, "mrn_conv3x3_c4_grouped_f32: input must be 16-byte aligned");
  if (B == 0 || H == 0 || W == 0) return MRN_OK;
  ConvFirstParams p;
  p.x = x; p.w = w_ohwi; p.bias = bias; p.y = y; p.stats = stats; p.x_gstride = x_group_stride;
  p.G = G; p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.act = act;
  p.tiles_w = (W + PX - 1) / PX;
  p.nblk = (int)mrn_conv3x3_c4_stats_blocks(B, H, W);
  const long grid = (long)G * B * H * p.tiles_w;
  if (Cout == 32) hipLaunchKernelGGL(conv_first_kernel<1>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(conv_first_kernel<2>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
  MRN_LAUNCH_CHECK("conv3x3_c4_grouped");
  return MRN_OK;
}
