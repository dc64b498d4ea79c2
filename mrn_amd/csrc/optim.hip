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

// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)), out[2] += 1 when the norm is not finite.
// A non-finite norm (an overflowed gradient) makes the update kernels skip the step: torch's clip_grad_norm_ + Adam would write NaN
// into every parameter, from which no later step recovers; parameters and moments stay as they were.  The host's step count -- Adam's
// bias correction, the learning-rate schedule -- advances all the same (the host never waits for the norm), which is why the skip is
// COUNTED: out[2] is the caller's persistent counter (FlatOptimizer.skipped_steps(), the learners' log lines, bench.py's JSON), so a
// persistent overflow shows up as a number instead of as training that quietly stops learning.
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
    if (!(norm <= 3.0e38f)) out[2] += 1.f;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const float* __restrict__ norm_coef,
                                                   float step_size, float beta1, float beta2, float bc2_sqrt, float eps) {
  const float coef = norm_coef ? norm_coef[1] : 1.f;
  if (norm_coef && !(norm_coef[0] <= 3.0e38f)) return;      // non-finite gradient norm: the step is skipped (see norm_finalize_kernel)
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

// torch.optim.SGD (momentum, weight decay; dampening 0, no Nesterov -- il_modules/base.py:74-79): buf = mu * buf + (g + wd * p),
// p -= lr * buf.  With buf zero-initialised the first step equals torch's "buf = clone(g)".
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf, long n,
                                                  const float* __restrict__ norm_coef, float lr, float momentum, float weight_decay) {
  const float coef = norm_coef ? norm_coef[1] : 1.f;
  if (norm_coef && !(norm_coef[0] <= 3.0e38f)) return;      // non-finite gradient norm: the step is skipped (see norm_finalize_kernel)
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gc = g[i] * coef;
    g[i] = gc;
    float d = gc + weight_decay * p[i];
    if (momentum != 0.f) {
      d = momentum * buf[i] + d;
      buf[i] = d;
    }
    p[i] = p[i] - lr * d;
  }
}

// torch.optim.Adadelta (rho, eps, no weight decay -- il_modules/base.py:80-83)
__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ sq,
                                                       float* __restrict__ acc, long n, const float* __restrict__ norm_coef,
                                                       float lr, float rho, float eps) {
  const float coef = norm_coef ? norm_coef[1] : 1.f;
  if (norm_coef && !(norm_coef[0] <= 3.0e38f)) return;      // non-finite gradient norm: the step is skipped (see norm_finalize_kernel)
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gc = g[i] * coef;
    g[i] = gc;
    const float s = sq[i] * rho + (1.f - rho) * gc * gc;
    sq[i] = s;
    const float delta = sqrtf(acc[i] + eps) / sqrtf(s + eps) * gc;
    acc[i] = acc[i] * rho + (1.f - rho) * delta * delta;
    p[i] = p[i] - lr * delta;
  }
}

// ---- EWC (il_modules/ewc.py:120-167) and weight alignment (modules/model.py:166-174) -------------------------------
// mode 0: fisher += g*g        mode 1: fisher = min(fisher * a, b)      mode 2: g += a * fisher * (p - mean)
__global__ __launch_bounds__(256) void ewc_elementwise_kernel(float* __restrict__ fisher, float* __restrict__ g,
                                                              const float* __restrict__ p, const float* __restrict__ mean,
                                                              long n, int mode, float a, float b) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (mode == 0) fisher[i] += g[i] * g[i];
    else if (mode == 1) fisher[i] = fminf(fisher[i] * a, b);
    else g[i] += a * fisher[i] * (p[i] - mean[i]);
  }
}

// part[blk] = sum fisher * (p - mean)^2 / 2
__global__ __launch_bounds__(256) void ewc_penalty_partial_kernel(const float* __restrict__ fisher, const float* __restrict__ p,
                                                                  const float* __restrict__ mean, long n, float* __restrict__ part) {
  __shared__ float scratch[4];
  float s = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = p[i] - mean[i];
    s += fisher[i] * d * d;
  }
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = 0.5f * s;
}

__global__ __launch_bounds__(256) void sum_finalize_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float scratch[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) out[0] = s;
}

// norms[r] = ||w[r][:]||_2   (one wave per row)
__global__ __launch_bounds__(256) void row_l2norm_kernel(const float* __restrict__ w, long ld, float* __restrict__ norms, long rows,
                                                        int C) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = w[row * ld + c]; s += v * v; }
  s = wave_sum(s);
  if (lane == 0) norms[row] = sqrtf(s);
}

// gamma = mean(norms[:n_old]) / mean(norms[n_old:]);  w[n_old:] *= gamma   (single block computes gamma, then all scale)
__global__ __launch_bounds__(256) void weight_align_kernel(float* __restrict__ w, long ld, const float* __restrict__ norms, long rows,
                                                          long n_old, int C, float* __restrict__ gamma_out) {
  __shared__ float scratch[4];
  float so = 0.f, sn = 0.f;
  for (long r = threadIdx.x; r < rows; r += 256) { if (r < n_old) so += norms[r]; else sn += norms[r]; }
  so = block_sum<256>(so, scratch);
  sn = block_sum<256>(sn, scratch);
  const float gamma = (so / (float)n_old) / (sn / (float)(rows - n_old));
  if (blockIdx.x == 0 && threadIdx.x == 0) gamma_out[0] = gamma;
  const long n = (rows - n_old) * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = n_old + i / C;
    const int c = (int)(i % C);
    w[r * ld + c] *= gamma;
  }
}

}  // namespace

static inline int opt_grid(long n) {
  long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  return g < 1 ? 1 : (int)g;
}

MRN_EXPORT int mrn_fisher_accumulate_f32(float* fisher, const float* grad, int64_t n, void* stream) {
  MRN_CHECK_ARG(fisher && grad, "mrn_fisher_accumulate_f32: null operand");
  if (n == 0) return MRN_OK;
  hipLaunchKernelGGL(ewc_elementwise_kernel, dim3(opt_grid(n)), dim3(256), 0, (hipStream_t)stream, fisher, (float*)grad,
                     (const float*)nullptr, (const float*)nullptr, (long)n, 0, 0.f, 0.f);
  MRN_LAUNCH_CHECK("fisher_accumulate");
  return MRN_OK;
}

// fisher = min(fisher / iterations, fisher_max)
MRN_EXPORT int mrn_fisher_finalize_f32(float* fisher, int64_t n, float inv_iterations, float fisher_max, void* stream) {
  MRN_CHECK_ARG(fisher, "mrn_fisher_finalize_f32: null operand");
  if (n == 0) return MRN_OK;
  hipLaunchKernelGGL(ewc_elementwise_kernel, dim3(opt_grid(n)), dim3(256), 0, (hipStream_t)stream, fisher, (float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (long)n, 1, inv_iterations, fisher_max);
  MRN_LAUNCH_CHECK("fisher_finalize");
  return MRN_OK;
}

// penalty[0] = sum fisher * (p - mean)^2 / 2 ; workspace: 2048 floats
MRN_EXPORT int mrn_ewc_penalty_fwd_f32(const float* fisher, const float* p, const float* mean, int64_t n, float* workspace,
                                       float* penalty, void* stream) {
  MRN_CHECK_ARG(fisher && p && mean && workspace && penalty, "mrn_ewc_penalty_fwd_f32: null operand");
  const int nb = opt_grid(n);
  hipLaunchKernelGGL(ewc_penalty_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, fisher, p, mean, (long)n, workspace);
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nb, penalty);
  MRN_LAUNCH_CHECK("ewc_penalty_fwd");
  return MRN_OK;
}

// grad += coef * fisher * (p - mean)
MRN_EXPORT int mrn_ewc_penalty_bwd_f32(const float* fisher, const float* p, const float* mean, float* grad, int64_t n,
                                       float coef, void* stream) {
  MRN_CHECK_ARG(fisher && p && mean && grad, "mrn_ewc_penalty_bwd_f32: null operand");
  if (n == 0) return MRN_OK;
  hipLaunchKernelGGL(ewc_elementwise_kernel, dim3(opt_grid(n)), dim3(256), 0, (hipStream_t)stream, (float*)fisher, grad, p, mean,
                     (long)n, 2, coef, 0.f);
  MRN_LAUNCH_CHECK("ewc_penalty_bwd");
  return MRN_OK;
}

// Model.weight_align: scale the newest `rows - n_old` rows of w by mean||old rows|| / mean||new rows||; workspace: rows floats
MRN_EXPORT int mrn_weight_align_f32(float* w, int64_t ld, int64_t rows, int64_t n_old, int C, float* workspace, float* gamma_out,
                                    void* stream) {
  MRN_CHECK_ARG(w && workspace && gamma_out && n_old > 0 && n_old < rows, "mrn_weight_align_f32: bad operands");
  hipLaunchKernelGGL(row_l2norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                     (long)ld, workspace, (long)rows, C);
  hipLaunchKernelGGL(weight_align_kernel, dim3(opt_grid((rows - n_old) * C)), dim3(256), 0, (hipStream_t)stream, w, (long)ld,
                     (const float*)workspace, (long)rows, (long)n_old, C, gamma_out);
  MRN_LAUNCH_CHECK("weight_align");
  return MRN_OK;
}

MRN_EXPORT int64_t mrn_grad_norm_workspace_floats(int64_t n) {
  int64_t b = (n / 4 + 255) / 256;
  if (b > 1024) b = 1024;
  return b < 1 ? 1 : b;
}

// norm_coef[0] = ||g||_2, norm_coef[1] = clip coefficient, norm_coef[2] += 1 when the norm is not finite (three floats; the caller
// zero-initialises the counter and keeps the buffer across steps); workspace holds mrn_grad_norm_workspace_floats(n) floats
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

// g *= coef (in place), then torch.optim.SGD's update (momentum buffer `buf`, may be NULL when momentum == 0)
MRN_EXPORT int mrn_sgd_step_f32(float* p, float* g, float* buf, int64_t n, const float* norm_coef, float lr, float momentum,
                                float weight_decay, void* stream) {
  MRN_CHECK_ARG(p && g && (buf || momentum == 0.f), "mrn_sgd_step_f32: null operand");
  if (n == 0) return MRN_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3(opt_grid(n) * 2), dim3(256), 0, (hipStream_t)stream, p, g, buf, (long)n, norm_coef, lr,
                     momentum, weight_decay);
  MRN_LAUNCH_CHECK("sgd_step");
  return MRN_OK;
}

// g *= coef (in place), then torch.optim.Adadelta's update (running averages square_avg / acc_delta)
MRN_EXPORT int mrn_adadelta_step_f32(float* p, float* g, float* square_avg, float* acc_delta, int64_t n, const float* norm_coef,
                                     float lr, float rho, float eps, void* stream) {
  MRN_CHECK_ARG(p && g && square_avg && acc_delta, "mrn_adadelta_step_f32: null operand");
  if (n == 0) return MRN_OK;
  hipLaunchKernelGGL(adadelta_kernel, dim3(opt_grid(n) * 2), dim3(256), 0, (hipStream_t)stream, p, g, square_avg, acc_delta, (long)n,
                     norm_coef, lr, rho, eps);
  MRN_LAUNCH_CHECK("adadelta_step");
  return MRN_OK;
}
