// Loss kernels: cross-entropy with ignore_index, and CTC (log-softmax + alpha/beta + gradient).
//
// Reference sites: CrossEntropyLoss(ignore_index=[PAD]) on [B*26, C] (il_modules/base.py:134, mrn.py:254-258),
// CrossEntropyLoss on the router weights (mrn.py:150-152,342,350), and
// preds.log_softmax(2).permute(1,0,2) -> CTCLoss(blank=0, reduction=mean, zero_infinity=True) (base.py:131,
// mrn.py:250-252,345-346).  Rows of logits have a free row stride (padded fan-in output).
#include "common.hpp"

namespace {

// one block per row: lse = logsumexp(row); loss = lse - row[target] (0 when ignored / no target)
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ x, long ld, const int64_t* __restrict__ target,
                                                      long ignore_index, float* __restrict__ lse_out,
                                                      float* __restrict__ loss_out, int C) {
  __shared__ float scratch[4];
  const long row = blockIdx.x;
  const float* xr = x + row * ld;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, xr[c]);
  m = block_max<256>(m, scratch);
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += expf(xr[c] - m);
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) {
    const float lse = m + logf(s);
    lse_out[row] = lse;
    if (loss_out) {
      const long t = target[row];
      loss_out[row] = (t == ignore_index) ? 0.f : lse - xr[t];
    }
  }
}

// loss = sum(loss_row) / n_valid ; inv_count = 1 / n_valid  (single block)
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ loss_row, const int64_t* __restrict__ target,
                                                          long ignore_index, long rows, float* __restrict__ loss,
                                                          float* __restrict__ inv_count) {
  __shared__ float scratch[4];
  float s = 0.f, n = 0.f;
  for (long r = threadIdx.x; r < rows; r += 256) {
    if (target[r] != ignore_index) { s += loss_row[r]; n += 1.f; }
  }
  s = block_sum<256>(s, scratch);
  n = block_sum<256>(n, scratch);
  if (threadIdx.x == 0) {
    loss[0] = s / n;
    inv_count[0] = 1.f / n;
  }
}

// dlogits[row][c] = (softmax - onehot) * upstream * inv_count   (zeros for ignored rows)
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ x, long ld, const int64_t* __restrict__ target,
                                                     long ignore_index, const float* __restrict__ lse,
                                                     const float* __restrict__ upstream, const float* __restrict__ inv_count,
                                                     float* __restrict__ dx, long ldd, int C) {
  const long row = blockIdx.x;
  const long t = target[row];
  const float g = (t == ignore_index) ? 0.f : upstream[0] * inv_count[0];
  const float l = lse[row];
  const float* xr = x + row * ld;
  float* dr = dx + row * ldd;
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = 0.f;
    if (g != 0.f) v = (expf(xr[c] - l) - (c == t ? 1.f : 0.f)) * g;
    dr[c] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// CTC.  One block (one wave) per sample: states s = 0..2L (blank, l1, blank, ...), lane = state.
// ---------------------------------------------------------------------------------------------
constexpr int CTC_MAXS = 64;   // 2*L+1 <= 64  ->  L <= 31 (reference: batch_max_length = 25 -> 51 states)

__device__ __forceinline__ float lse2(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
}

__global__ __launch_bounds__(64) void ctc_alpha_beta_kernel(const float* __restrict__ x, long ld, const float* __restrict__ lse,
                                                            const int64_t* __restrict__ targets, long tstride,
                                                            const int* __restrict__ tlen, float* __restrict__ nll_out,
                                                            float* __restrict__ occ,  // [B][T][CTC_MAXS]: exp(alpha+beta+nll-lp)
                                                            int T, int blank) {
  extern __shared__ float alpha[];  // [T][CTC_MAXS]
  const int b = blockIdx.x, s = threadIdx.x;
  const int L = tlen[b];
  const int S = 2 * L + 1;
  const bool live = s < S;
  int cls = blank;
  if (live && (s & 1)) cls = (int)targets[(long)b * tstride + (s >> 1)];
  int cls_m2 = blank;
  if (live && (s & 1) && s >= 3) cls_m2 = (int)targets[(long)b * tstride + (s >> 1) - 1];
  const bool skip_ok = live && (s & 1) && s >= 3 && cls != cls_m2;  // may come from s-2
  const float* xb = x + (long)b * T * ld;
  const float* lb = lse + (long)b * T;

  // forward
  float a = -INFINITY;
  {
    const float lp = live ? xb[cls] - lb[0] : -INFINITY;
    if (s == 0 || (s == 1 && L > 0)) a = lp;
    alpha[s] = a;
  }
  for (int t = 1; t < T; ++t) {
    float a1 = __shfl_up(a, 1), a2 = __shfl_up(a, 2);
    if (s < 1) a1 = -INFINITY;
    if (s < 2) a2 = -INFINITY;
    float acc = lse2(a, a1);
    if (skip_ok) acc = lse2(acc, a2);
    const float lp = live ? xb[(long)t * ld + cls] - lb[t] : -INFINITY;
    a = live ? acc + lp : -INFINITY;
    alpha[t * CTC_MAXS + s] = a;
  }
  // total log-likelihood
  const float aS1 = __shfl(a, S - 1 < 0 ? 0 : S - 1);
  const float aS2 = S >= 2 ? __shfl(a, S - 2) : -INFINITY;
  const float ll = lse2(aS1, aS2);
  const float nll = -ll;
  if (s == 0) nll_out[b] = nll;

  // backward; skip from s to s+2 allowed when s odd, s+2 < S and cls[s+2] != cls[s]
  int cls_p2 = blank;
  if ((s & 1) && s + 2 < S) cls_p2 = (int)targets[(long)b * tstride + (s >> 1) + 1];
  const bool skip_fw = live && (s & 1) && s + 2 < S && cls_p2 != cls;
  float be = -INFINITY;
  {
    const float lp = live ? xb[(long)(T - 1) * ld + cls] - lb[T - 1] : -INFINITY;
    if (s == S - 1 || (s == S - 2 && S >= 2)) be = lp;
    const float al = alpha[(T - 1) * CTC_MAXS + s];
    occ[((long)b * T + (T - 1)) * CTC_MAXS + s] = live ? expf(al + be + nll - lp) : 0.f;
  }
  for (int t = T - 2; t >= 0; --t) {
    float b1 = __shfl_down(be, 1), b2 = __shfl_down(be, 2);
    if (s + 1 >= S) b1 = -INFINITY;
    if (s + 2 >= S) b2 = -INFINITY;
    float acc = lse2(be, b1);
    if (skip_fw) acc = lse2(acc, b2);
    const float lp = live ? xb[(long)t * ld + cls] - lb[t] : -INFINITY;
    be = live ? acc + lp : -INFINITY;
    const float al = alpha[t * CTC_MAXS + s];
    // alpha and beta both include lp(t, s): posterior = exp(alpha + beta - lp - ll)
    occ[((long)b * T + t) * CTC_MAXS + s] = live ? expf(al + be + nll - lp) : 0.f;
  }
}

// loss = mean_b( zero_inf(nll_b) / max(len_b, 1) )
__global__ __launch_bounds__(256) void ctc_finalize_kernel(const float* __restrict__ nll, const int* __restrict__ tlen, int B,
                                                           float* __restrict__ loss) {
  __shared__ float scratch[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float v = nll[b];
    if (isinf(v)) v = 0.f;
    const int L = tlen[b] < 1 ? 1 : tlen[b];
    s += v / (float)L;
  }
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) loss[0] = s / (float)B;
}

// dlogits[b][t][c] = g_b * (softmax[b][t][c] - sum_{s: cls_s = c} occ[b][t][s]),  g_b = upstream / (B * max(len_b,1))
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ x, long ld, const float* __restrict__ lse,
                                                       const float* __restrict__ occ, const int64_t* __restrict__ targets,
                                                       long tstride, const int* __restrict__ tlen, const float* __restrict__ nll,
                                                       const float* __restrict__ upstream, float* __restrict__ dx, long ldd,
                                                       int B, int T, int C, int blank) {
  const long row = blockIdx.x;
  const int b = (int)(row / T);
  const int L = tlen[b];
  const float nl = nll[b];
  float g = upstream[0] / ((float)B * (float)(L < 1 ? 1 : L));
  if (isinf(nl) || nl != nl) g = 0.f;   // zero_infinity
  const float l = lse[row];
  const float* xr = x + row * ld;
  float* dr = dx + row * ldd;
  for (int c = threadIdx.x; c < C; c += 256) dr[c] = g == 0.f ? 0.f : expf(xr[c] - l) * g;
  __syncthreads();
  const int S = 2 * L + 1;
  if (g != 0.f && threadIdx.x < S) {
    const int s = threadIdx.x;
    const int cls = (s & 1) ? (int)targets[(long)b * tstride + (s >> 1)] : blank;
    atomicAdd(dr + cls, -occ[row * CTC_MAXS + s] * g);
  }
}

// ---- knowledge distillation (LwF): -sum softmax(old/T) * log_softmax(new/T) / rows over the class slice [c0, c1) -------
// reference il_modules/lwf.py:81-87,111-114.  One block per row.  loss_rows[row] = -sum_c p_old * logp_new.
// When dnew != nullptr also writes d loss / d new = (softmax(new/T) - softmax(old/T)) / T * g (zeros outside the slice).
__global__ __launch_bounds__(256) void kd_rows_kernel(const float* __restrict__ xnew, long ldn, const float* __restrict__ xold,
                                                      long ldo, int c0, int c1, float invT, float* __restrict__ loss_rows,
                                                      const float* __restrict__ upstream, float inv_rows,
                                                      float* __restrict__ dnew, long ldd, int C) {
  __shared__ float scratch[4];
  const long row = blockIdx.x;
  const float* a = xnew + row * ldn;
  const float* b = xold + row * ldo;
  float ma = -INFINITY, mb = -INFINITY;
  for (int c = c0 + threadIdx.x; c < c1; c += 256) { ma = fmaxf(ma, a[c] * invT); mb = fmaxf(mb, b[c] * invT); }
  ma = block_max<256>(ma, scratch);
  mb = block_max<256>(mb, scratch);
  float sa = 0.f, sb = 0.f;
  for (int c = c0 + threadIdx.x; c < c1; c += 256) { sa += expf(a[c] * invT - ma); sb += expf(b[c] * invT - mb); }
  sa = block_sum<256>(sa, scratch);
  sb = block_sum<256>(sb, scratch);
  const float lsa = ma + logf(sa), lsb = mb + logf(sb);
  if (loss_rows) {
    float l = 0.f;
    for (int c = c0 + threadIdx.x; c < c1; c += 256) l -= expf(b[c] * invT - lsb) * (a[c] * invT - lsa);
    l = block_sum<256>(l, scratch);
    if (threadIdx.x == 0) loss_rows[row] = l;
  }
  if (dnew) {
    const float g = upstream[0] * inv_rows * invT;
    float* d = dnew + row * ldd;
    for (int c = threadIdx.x; c < C; c += 256)
      d[c] = (c >= c0 && c < c1) ? (expf(a[c] * invT - lsa) - expf(b[c] * invT - lsb)) * g : 0.f;
  }
}

__global__ __launch_bounds__(256) void mean_finalize_kernel(const float* __restrict__ v, long n, float scale, float* __restrict__ out) {
  __shared__ float scratch[4];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) s += v[i];
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) out[0] = s * scale;
}

}  // namespace

MRN_EXPORT int mrn_kd_loss_fwd_f32(const float* xnew, int64_t ldn, const float* xold, int64_t ldo, int c0, int c1, float T,
                                   int64_t rows, float* loss_rows, float* loss, void* stream) {
  MRN_CHECK_ARG(xnew && xold && loss_rows && loss && c1 > c0 && T > 0.f, "mrn_kd_loss_fwd_f32: bad operands");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(kd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, xnew, (long)ldn, xold, (long)ldo, c0,
                     c1, 1.f / T, loss_rows, (const float*)nullptr, 0.f, (float*)nullptr, 0L, 0);
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)loss_rows, (long)rows,
                     1.f / (float)rows, loss);
  MRN_LAUNCH_CHECK("kd_loss_fwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_kd_loss_bwd_f32(const float* xnew, int64_t ldn, const float* xold, int64_t ldo, int c0, int c1, float T,
                                   int64_t rows, const float* upstream, float* dnew, int64_t ldd, int C, void* stream) {
  MRN_CHECK_ARG(xnew && xold && upstream && dnew && c1 > c0 && T > 0.f, "mrn_kd_loss_bwd_f32: bad operands");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(kd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, xnew, (long)ldn, xold, (long)ldo, c0,
                     c1, 1.f / T, (float*)nullptr, upstream, 1.f / (float)rows, dnew, (long)ldd, C);
  MRN_LAUNCH_CHECK("kd_loss_bwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_ce_loss_fwd_f32(const float* logits, int64_t ld, const int64_t* target, int64_t ignore_index,
                                   int64_t rows, int C, float* lse, float* loss_rows, float* loss, float* inv_count,
                                   void* stream) {
  MRN_CHECK_ARG(logits && target && lse && loss_rows && loss && inv_count && C > 0, "mrn_ce_loss_fwd_f32: bad operands");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, target,
                     (long)ignore_index, lse, loss_rows, C);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)loss_rows, target,
                     (long)ignore_index, (long)rows, loss, inv_count);
  MRN_LAUNCH_CHECK("ce_loss_fwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_ce_loss_bwd_f32(const float* logits, int64_t ld, const int64_t* target, int64_t ignore_index,
                                   const float* lse, const float* upstream, const float* inv_count, float* dlogits,
                                   int64_t ldd, int64_t rows, int C, void* stream) {
  MRN_CHECK_ARG(logits && target && lse && upstream && inv_count && dlogits, "mrn_ce_loss_bwd_f32: null operand");
  if (rows == 0) return MRN_OK;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, target,
                     (long)ignore_index, lse, upstream, inv_count, dlogits, (long)ldd, C);
  MRN_LAUNCH_CHECK("ce_loss_bwd");
  return MRN_OK;
}

MRN_EXPORT int64_t mrn_ctc_occ_floats(int B, int T) { return (int64_t)B * T * CTC_MAXS; }

// logits [B][T][C] (row stride ld), targets [B][tstride] int64 (padded), target_len [B] int32; input lengths are all T.
MRN_EXPORT int mrn_ctc_loss_fwd_f32(const float* logits, int64_t ld, const int64_t* targets, int64_t tstride,
                                    const int* target_len, int max_target_len, float* lse, float* nll, float* occ,
                                    float* loss, int B, int T, int C, int blank, void* stream) {
  MRN_CHECK_ARG(logits && targets && target_len && lse && nll && occ && loss, "mrn_ctc_loss_fwd_f32: null operand");
  MRN_CHECK_ARG(2 * max_target_len + 1 <= CTC_MAXS, "mrn_ctc_loss_fwd_f32: target length %d exceeds the %d-state kernel",
                max_target_len, CTC_MAXS);
  if (B == 0) return MRN_OK;
  hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)(B * T)), dim3(256), 0, (hipStream_t)stream, logits, (long)ld,
                     (const int64_t*)nullptr, -1L, lse, (float*)nullptr, C);
  hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(B), dim3(64), sizeof(float) * T * CTC_MAXS, (hipStream_t)stream, logits,
                     (long)ld, (const float*)lse, targets, (long)tstride, target_len, nll, occ, T, blank);
  hipLaunchKernelGGL(ctc_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)nll, target_len, B, loss);
  MRN_LAUNCH_CHECK("ctc_loss_fwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_ctc_loss_bwd_f32(const float* logits, int64_t ld, const float* lse, const float* occ,
                                    const int64_t* targets, int64_t tstride, const int* target_len, const float* nll,
                                    const float* upstream, float* dlogits, int64_t ldd, int B, int T, int C, int blank,
                                    void* stream) {
  MRN_CHECK_ARG(logits && lse && occ && targets && target_len && nll && upstream && dlogits, "mrn_ctc_loss_bwd_f32: null operand");
  if (B == 0) return MRN_OK;
  hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)(B * T)), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, lse, occ,
                     targets, (long)tstride, target_len, nll, upstream, dlogits, (long)ldd, B, T, C, blank);
  MRN_LAUNCH_CHECK("ctc_loss_bwd");
  return MRN_OK;
}
