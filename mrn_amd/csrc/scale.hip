// Power-of-two range scales of split-fp16 operands: scale = {s, 1/s} with s the largest power of two such that s * max|x| <= target,
// computed on the device (no host synchronisation).  Weights of the frozen experts are scaled to (2^13, 2^14]; both operands of a
// TRAINED layer (activations, gradients of 1e-6 .. 1e-8) are scaled per call so that hi + lo keeps 22 significand bits.
// Replaces nothing in the reference (its fp32 products need no scaling): part of how nn.Conv2d / nn.Linear products
// (modules/feature_extraction.py:19-44,214-294, il_modules/mrn.py:260-261) run on the f16 MFMA pipe at fp32-class accuracy.
#include "common.hpp"

namespace {

// the three-launch form (clear, maximum, finalise): the default; A/B partner: the single launch above (MRN_POW2_LAUNCHES=1)
// ws: two words the caller zeroed ONCE ([0] = running max|w| as uint bits -- non-negative floats order like unsigned ints --, [1] = arrival
// ticket), reusable by every later call on the same stream: every block folds its maximum into ws[0] and takes a ticket; the LAST block to
// arrive turns the maximum into the power-of-two pair {s, 1/s} and puts the two words back to zero.  One launch per operand (rounds 1-3 ran
// clear + maximum + finalise: 3 x 145 launches per loop-A step).
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
// words zeroed ONCE by the caller, owned by the calling stream (the kernel's last block restores the zeros).  One launch, no host sync.
MRN_EXPORT int mrn_pow2_scale_f32(const float* w, int64_t n, float target, float* scale, void* workspace, void* stream) {
  MRN_CHECK_ARG(w && scale && workspace && target > 0.f, "mrn_pow2_scale_f32: bad operands");
  MRN_CHECK_ARG((uintptr_t)w % 16 == 0, "mrn_pow2_scale_f32: operand must be 16-byte aligned");
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
