// Shared helpers for the MRN gfx950 kernel library (libmrn_hip.so).
// Everything here is internal; the exported surface is include/mrn_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define MRN_EXPORT extern "C" __attribute__((visibility("default")))

// error codes returned across the C ABI (0 = ok, >0 = hipError_t, <0 = library code)
enum {
  MRN_OK = 0,
  MRN_ERR_BAD_ARG = -1,
  MRN_ERR_UNSUPPORTED = -2,
  MRN_ERR_WORKSPACE = -3,
};

void mrn_set_error(const char* fmt, ...);

#define MRN_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      mrn_set_error(__VA_ARGS__);         \
      return MRN_ERR_BAD_ARG;             \
    }                                     \
  } while (0)

// check the launch that just happened (asynchronous errors surface later, as in any HIP code)
#define MRN_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      mrn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware remap of a linear workgroup id so that consecutive logical tiles share one XCD's L2
// (block b is dispatched to XCD b % 8 on gfx950; bijective for any grid size).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// x = hi + lo as two fp16 (22 significand bits).  The operand is pinned in a register first: under -ffp-contract=fast the
// compiler may otherwise fold the multiply / fma that PRODUCED x into the conversion (v_fma_mixlo_f16: one rounding from the
// exact product) for one use of hi and convert the fp32-rounded value (two roundings) for another -- the two differ by one
// fp16 ulp when the fp32 value sits on an fp16 tie, and the pair then misses x by 2^-11 (seen on 2 of 65536 elements).
__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
  asm volatile("" : "+v"(x));
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// the same for range-scaled GRADIENT operands (BPTT gate gradients, scaled by a power of two from max|dout|): saturate at the
// fp16 range instead of hi = inf, lo = -inf (NaN products) when back-propagation through time grows them past the headroom --
// the exact-fp32 kernels give large finite gradients there, which the global-norm clip then handles
__device__ __forceinline__ void split_f16_sat(float x, _Float16& hi, _Float16& lo) {
  x = fminf(fmaxf(x, -65504.f), 65504.f);
  split_f16(x, hi, lo);
}

// torch.relu / max_pool2d propagate NaN; v_max_f32 (fmaxf) returns the OTHER operand.  The pooled epilogues and the separate pooling
// pass must agree on non-finite data too (a NaN in a frozen expert's raw convolution output must reach the features).
__device__ __forceinline__ float relu_nan(float v, float floor_ = 0.f) { return v != v ? v : fmaxf(v, floor_); }
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a || b != b) ? a + b : fmaxf(a, b); }

// GELU(x) = x * Phi(x) with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute on erf, so <= 0.75e-7 * |x| on the
// result: below one fp32 ulp of the activations that matter): one v_rcp, one v_exp and 7 FMAs instead of the branchy ~45
// instruction erff of the device library -- the GELU of the SVTR Mlp runs in a GEMM epilogue (64 elements per thread).
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = pl * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);      // 1 - erf(|x| / sqrt 2)
  const float cdf = x >= 0.f ? fmaf(-0.5f, e, 1.f) : 0.5f * e;                         // Phi(x)
  return x * cdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// block-wide sum for blocks of NT threads (NT multiple of 64); scratch must hold NT/64 floats
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) scratch[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) r += scratch[i];
  return r;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* scratch) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) scratch[w] = v;
  __syncthreads();
  float r = scratch[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.f / (1.f + expf(-x)); }
