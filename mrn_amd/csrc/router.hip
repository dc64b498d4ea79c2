// MRN fan-in and gate tail.
//
// Fan-in (reference modules/model.py:361-364,410-423): the reference pads every older expert's logits to the
// newest class count with ONES (torch.ones, despite the name pad_zeros_features), stacks [I,B,T,C], permutes,
// multiplies by the weights, permutes back, makes it contiguous and sums -- about five passes over I*B*T*C.
// Here: one pass; each real logit is read once (16 B per lane), the padding is synthesised.
//   out[b][t][c] = sum_i w[b][i] * (c < C_i ? L_i[b][t][c] : 1)
// Backward: dw[b][i] = sum_{t,c} dOut[b][t][c] * Lpad_i[b][t][c]   (experts are frozen in loop B).
//
// Gate tail (modules/model.py:403-406): route = Linear(P -> 1) over the patch axis of channel_route's output,
// squeeze, softmax(beta * s).
#include "common.hpp"

namespace {

constexpr int MAXI = 8;

struct FaninArgs {
  const float* L[MAXI];
  long ld[MAXI];   // row stride of each expert's logits (multiple of 4)
  int C[MAXI];     // real class count of each expert
  int I;
};

__global__ __launch_bounds__(256) void fanin_fwd_kernel(const FaninArgs a, const float* __restrict__ w, float* __restrict__ out,
                                                        long ldo, int T, int C) {
  // grid: (ceil(C4/256), B*T)
  const long row = blockIdx.y;
  const int b = (int)(row / T);
  const int c4 = blockIdx.x * 256 + threadIdx.x;
  const int c = c4 * 4;
  if (c >= C) return;
  float wi[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) wi[i] = i < a.I ? w[(long)b * a.I + i] : 0.f;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < a.I) {
      f32x4 v = {1.f, 1.f, 1.f, 1.f};
      if (c < a.C[i]) {
        v = *reinterpret_cast<const f32x4*>(a.L[i] + row * a.ld[i] + c);   // ld % 4 == 0: whole quad is in the row
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j >= a.C[i]) v[j] = 1.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaf(wi[i], v[j], o[j]);
    }
  }
  // output rows are padded to a multiple of 4 floats: the lanes beyond C land in the row's padding
  *reinterpret_cast<f32x4*>(out + row * ldo + c) = o;
}

// partial[row][i] = sum_c dOut[row][c] * Lpad_i[row][c]; one block per (b,t) row
__global__ __launch_bounds__(256) void fanin_bwd_kernel(const FaninArgs a, const float* __restrict__ dout, long ldd,
                                                        float* __restrict__ partial, int C) {
  __shared__ float scratch[4];
  const long row = blockIdx.x;
  float acc[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) acc[i] = 0.f;
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    f32x4 d = *reinterpret_cast<const f32x4*>(dout + row * ldd + c);   // rows padded to a multiple of 4
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c + j >= C) d[j] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      if (i < a.I) {
        f32x4 v = {1.f, 1.f, 1.f, 1.f};
        if (c < a.C[i]) {
          v = *reinterpret_cast<const f32x4*>(a.L[i] + row * a.ld[i] + c);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c + j >= a.C[i]) v[j] = 1.f;
        }
        acc[i] += d[0] * v[0] + d[1] * v[1] + d[2] * v[2] + d[3] * v[3];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < a.I) {
      const float s = block_sum<256>(acc[i], scratch);
      if (threadIdx.x == 0) partial[row * a.I + i] = s;
    }
  }
}

// dw[b][i] = sum_t partial[b][t][i]
__global__ void fanin_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int B, int T, int I) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * I) return;
  const int b = idx / I, i = idx - b * I;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += partial[((long)b * T + t) * I + i];
  dw[idx] = s;
}

// hard routing (modules/model.py:377,393): out[b] = Lpad_{index[b]}[b]
__global__ __launch_bounds__(256) void select_expert_kernel(const FaninArgs a, const int64_t* __restrict__ index,
                                                            float* __restrict__ out, long ldo, int T, int C) {
  const long row = blockIdx.y;
  const int b = (int)(row / T);
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int i = (int)index[b];
  float v = 1.f;
#pragma unroll
  for (int k = 0; k < MAXI; ++k)
    if (k == i && c < a.C[k]) v = a.L[k][row * a.ld[k] + c];
  out[row * ldo + c] = v;
}

// s[b][i] = sum_p Wr[p] * r[b][p][i] + br ; w = softmax(beta * s)      (one thread per sample, I <= 8)
__global__ void gate_tail_fwd_kernel(const float* __restrict__ r, const float* __restrict__ Wr, const float* __restrict__ br,
                                     float beta, float* __restrict__ s_out, float* __restrict__ w_out,
                                     int64_t* __restrict__ argmax_out, int B, int P, int I) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) s[i] = 0.f;
  for (int p = 0; p < P; ++p) {
    const float wp = Wr[p];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (i < I) s[i] = fmaf(wp, r[((long)b * P + p) * I + i], s[i]);
  }
  float m = -INFINITY;
  int am = 0;
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (i < I) {
      s[i] += br[0];
      if (s[i] > m) { m = s[i]; am = i; }
      if (s_out) s_out[(long)b * I + i] = s[i];
    }
  if (argmax_out) argmax_out[b] = am;
  if (!w_out) return;
  float e[MAXI], sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (i < I) { e[i] = expf(beta * (s[i] - m)); sum += e[i]; }
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (i < I) w_out[(long)b * I + i] = e[i] / sum;
}

// ds = beta * w * (dw - sum_i dw*w);  dr[b][p][i] = Wr[p] * ds[b][i]
__global__ void gate_tail_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dw, const float* __restrict__ Wr,
                                     float beta, float* __restrict__ ds, float* __restrict__ dr, int B, int P, int I) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float dot = 0.f;
  for (int i = 0; i < I; ++i) dot += dw[(long)b * I + i] * w[(long)b * I + i];
  float d[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (i < I) {
      d[i] = beta * w[(long)b * I + i] * (dw[(long)b * I + i] - dot);
      ds[(long)b * I + i] = d[i];
    }
  for (int p = 0; p < P; ++p) {
    const float wp = Wr[p];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (i < I) dr[((long)b * P + p) * I + i] = wp * d[i];
  }
}

// dWr[p] = sum_{b,i} ds[b][i] * r[b][p][i]  (block per p);  block P: dbr = sum ds
__global__ __launch_bounds__(256) void gate_tail_wgrad_kernel(const float* __restrict__ ds, const float* __restrict__ r,
                                                              float* __restrict__ dWr, float* __restrict__ dbr, int B, int P,
                                                              int I) {
  __shared__ float scratch[4];
  const int p = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < B * I; k += 256) {
    const int b = k / I, i = k - b * I;
    s += ds[k] * (p < P ? r[((long)b * P + p) * I + i] : 1.f);
  }
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) {
    if (p < P) dWr[p] = s;
    else dbr[0] = s;
  }
}

}  // namespace

static int fill_fanin(FaninArgs& a, const void* const* logits, const int64_t* lds, const int* classes, int I,
                      const char* who) {
  MRN_CHECK_ARG(I >= 1 && I <= MAXI, "%s: I=%d out of range (1..%d)", who, I, MAXI);
  a.I = I;
  for (int i = 0; i < MAXI; ++i) { a.L[i] = nullptr; a.ld[i] = 0; a.C[i] = 0; }
  for (int i = 0; i < I; ++i) {
    a.L[i] = (const float*)logits[i];
    a.ld[i] = lds[i];
    a.C[i] = classes[i];
    MRN_CHECK_ARG(a.L[i] && lds[i] % 4 == 0 && lds[i] >= classes[i] && ((uintptr_t)a.L[i] % 16 == 0),
                  "%s: expert %d logits must be 16-byte aligned with a row stride that is a multiple of 4 (ld=%ld C=%d)", who, i,
                  (long)lds[i], classes[i]);
  }
  return MRN_OK;
}

// `logits`, `lds`, `classes` are HOST arrays of length I (device pointers / row strides / class counts per expert)
MRN_EXPORT int mrn_fanin_fwd_f32(const void* const* logits, const int64_t* lds, const int* classes, int I, const float* w,
                                 float* out, int64_t ldo, int B, int T, int C, void* stream) {
  FaninArgs a;
  int rc = fill_fanin(a, logits, lds, classes, I, "mrn_fanin_fwd_f32");
  if (rc) return rc;
  MRN_CHECK_ARG(w && out && ldo >= C && ldo % 4 == 0 && ((uintptr_t)out % 16 == 0),
                "mrn_fanin_fwd_f32: output must be 16-byte aligned with a row stride that is a multiple of 4");
  if (B * T == 0) return MRN_OK;
  hipLaunchKernelGGL(fanin_fwd_kernel, dim3(ceil_div(ceil_div(C, 4), 256), B * T), dim3(256), 0, (hipStream_t)stream, a, w, out,
                     (long)ldo, T, C);
  MRN_LAUNCH_CHECK("fanin_fwd");
  return MRN_OK;
}

// workspace: B*T*I floats
MRN_EXPORT int mrn_fanin_bwd_f32(const void* const* logits, const int64_t* lds, const int* classes, int I,
                                 const float* dout, int64_t ldd, float* dw, float* workspace, int B, int T, int C,
                                 void* stream) {
  FaninArgs a;
  int rc = fill_fanin(a, logits, lds, classes, I, "mrn_fanin_bwd_f32");
  if (rc) return rc;
  MRN_CHECK_ARG(dout && dw && workspace && ldd % 4 == 0 && ((uintptr_t)dout % 16 == 0), "mrn_fanin_bwd_f32: bad operands");
  if (B * T == 0) return MRN_OK;
  hipLaunchKernelGGL(fanin_bwd_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, a, dout, (long)ldd, workspace, C);
  hipLaunchKernelGGL(fanin_bwd_reduce_kernel, dim3(ceil_div(B * I, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, dw, B, T, I);
  MRN_LAUNCH_CHECK("fanin_bwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_select_expert_f32(const void* const* logits, const int64_t* lds, const int* classes, int I,
                                     const int64_t* index, float* out, int64_t ldo, int B, int T, int C, void* stream) {
  FaninArgs a;
  int rc = fill_fanin(a, logits, lds, classes, I, "mrn_select_expert_f32");
  if (rc) return rc;
  MRN_CHECK_ARG(index && out, "mrn_select_expert_f32: null operand");
  if (B * T == 0) return MRN_OK;
  hipLaunchKernelGGL(select_expert_kernel, dim3(ceil_div(C, 256), B * T), dim3(256), 0, (hipStream_t)stream, a, index, out,
                     (long)ldo, T, C);
  MRN_LAUNCH_CHECK("select_expert");
  return MRN_OK;
}

MRN_EXPORT int mrn_gate_tail_fwd_f32(const float* r, const float* w_route, const float* b_route, float beta, float* s_out,
                                     float* w_out, int64_t* argmax_out, int B, int P, int I, void* stream) {
  MRN_CHECK_ARG(r && w_route && b_route && I >= 1 && I <= MAXI, "mrn_gate_tail_fwd_f32: bad operands (I=%d)", I);
  if (B == 0) return MRN_OK;
  hipLaunchKernelGGL(gate_tail_fwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, r, w_route, b_route, beta,
                     s_out, w_out, argmax_out, B, P, I);
  MRN_LAUNCH_CHECK("gate_tail_fwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_gate_tail_bwd_f32(const float* w, const float* dw, const float* r, const float* w_route, float beta,
                                     float* ds, float* dr, float* d_w_route, float* d_b_route, int B, int P, int I,
                                     void* stream) {
  MRN_CHECK_ARG(w && dw && r && w_route && ds && dr && d_w_route && d_b_route && I <= MAXI, "mrn_gate_tail_bwd_f32: bad operands");
  if (B == 0) return MRN_OK;
  hipLaunchKernelGGL(gate_tail_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, w, dw, w_route, beta, ds,
                     dr, B, P, I);
  hipLaunchKernelGGL(gate_tail_wgrad_kernel, dim3(P + 1), dim3(256), 0, (hipStream_t)stream, (const float*)ds, r, d_w_route,
                     d_b_route, B, P, I);
  MRN_LAUNCH_CHECK("gate_tail_bwd");
  return MRN_OK;
}
