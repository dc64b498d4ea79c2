// Optimiser step over flat parameter / gradient buffers: global-norm clip + Adam in two HBM passes.
//
// Reference: torch.nn.utils.clip_grad_norm_(model.parameters(), 5) followed by torch.optim.Adam.step
// (il_modules/base.py:255-262, il_modules/mrn.py:362-367; Adam lr 5e-4, betas (0.9, 0.999), eps 1e-8, no weight
// decay: base.py:85).  The trainable set lives in ONE flat fp32 buffer (views handed to the modules), so the
// norm is one reduction (4 B/param read) and the update one pass (16 B/param read, 12-16 B/param written),
// and the data-parallel gradient all-reduce is a single RCCL call on the same buffer.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void sqsum_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float scratch[4];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(256) void norm_finalize_kernel(const float* __restrict__ part, int nblk, float max_norm,
                                                            float* __restrict__ out) {
  __shared__ float scratch[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    out[0] = norm;
    const float c = max_norm / (norm + 1e-6f);
    out[1] = c < 1.f ? c : 1.f;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const float* __restrict__ norm_coef,
                                                   float step_size, float beta1, float beta2, float bc2_sqrt, float eps) {
  const float coef = norm_coef ? norm_coef[1] : 1.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    g[i] = gi;                                           // clip_grad_norm_ scales .grad in place
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);  // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

}  // namespace

MRN_EXPORT int64_t mrn_grad_norm_workspace_floats(int64_t n) {
  int64_t b = (n / 4 + 255) / 256;
  if (b > 1024) b = 1024;
  return b < 1 ? 1 : b;
}

// norm_coef[0] = ||g||_2, norm_coef[1] = clip coefficient; workspace holds mrn_grad_norm_workspace_floats(n) floats
MRN_EXPORT int mrn_grad_norm_clip_f32(const float* g, int64_t n, float max_norm, float* workspace, float* norm_coef,
                                      void* stream) {
  MRN_CHECK_ARG(g && workspace && norm_coef && ((uintptr_t)g % 16 == 0), "mrn_grad_norm_clip_f32: bad operands");
  const int nblk = (int)mrn_grad_norm_workspace_floats(n);
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, (long)n, workspace);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk, max_norm,
                     norm_coef);
  MRN_LAUNCH_CHECK("grad_norm_clip");
  return MRN_OK;
}

// step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) are computed by the host in double precision
MRN_EXPORT int mrn_adam_step_f32(float* p, float* g, float* m, float* v, int64_t n, const float* norm_coef,
                                 float step_size, float beta1, float beta2, float bc2_sqrt, float eps, void* stream) {
  MRN_CHECK_ARG(p && g && m && v, "mrn_adam_step_f32: null operand");
  if (n == 0) return MRN_OK;
  long grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, norm_coef,
                     step_size, beta1, beta2, bc2_sqrt, eps);
  MRN_LAUNCH_CHECK("adam_step");
  return MRN_OK;
}
