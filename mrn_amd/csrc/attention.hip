// Fused multi-head attention of the SVTR mixing blocks (reference modules/svtr.py:90-152: qkv -> q k^T * scale (+ local
// window mask) -> softmax -> attn v), head dimension 32, inference path of the frozen experts.
//
// One workgroup (4 waves) owns 128 query tokens of one (sample, head); each wave owns 32 of them and walks the keys in
// tiles of 32 with an online softmax, so the [B, heads, N, N] score tensor (537 MB per block at B = 256, N = 512) never
// exists in HBM.  The K and V tiles are fetched once per workgroup (one 16-byte load per thread and operand, prefetched
// a tile ahead) into a two-stage LDS ring shared by the four waves.  Products run on the exact fp32 MFMA
// (v_mfma_f32_32x32x2_f32).  Scores are computed TRANSPOSED, S^T = K Q^T: in the 32x32 accumulator layout a lane then
// holds one QUERY (column lane & 31) and its 16 registers hold 16 keys, so the softmax reductions over keys are register
// reductions plus one cross-half shuffle, the running rescale of the output is a per-lane scalar, and the probabilities
// P^T already sit in the B-operand layout of the second product O^T = V^T P^T (register e pairs key
// k0(e) = (e&3) + 8(e>>2) in lanes 0-31 with key k0(e) + 4 in lanes 32-63; V is read from LDS in the same pairing).
// The contraction over d is order-free, so MFMA step t of the first product takes d = 16*(lane>>5) + t: every lane reads
// 16 contiguous floats of its K (and Q) row.  Softmax runs in base 2 (log2 e folded into the q scale and the mask).
// The additive mask must be symmetric (SVTR's local window mask is): mask[key][query] is read row-wise.
#include "common.hpp"

namespace {

constexpr int HD = 32;          // head dimension of every SVTR stage (embed_dim / num_heads = 64/2 = 128/4 = 256/8)
constexpr int AW = 4;           // waves per workgroup: 128 query tokens
constexpr int KS = 36;          // LDS row stride of the K tile in floats (16-byte aligned rows, conflict-free b128 reads)
constexpr int VS = 40;          // LDS row stride of the V tile: rows key and key + 4 land on disjoint bank halves
constexpr float LOG2E = 1.4426950408889634f;
typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));

struct AttnParams {
  const float* qkv;    // [B][N][3*C]: q | k | v, each (head, d) innermost
  const float* mask;   // [N][N] additive, symmetric, or null
  float* out;          // [B][N][C] fp32, or null
  unsigned char* out_hl;   // the same tensor as HL32 lines [B*N][heads][hi 32 | lo 32] (a head is one 32-channel block), or null
  int B, N, C, heads;
  float scale;
};

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(AW * 64) void svtr_attention_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) float lds_k[2][32 * KS];
  __shared__ __attribute__((aligned(16))) float lds_v[2][32 * VS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n31 = lane & 31, half = lane >> 5;
  const int qchunks = (p.N + 32 * AW - 1) / (32 * AW);
  const int qc = blockIdx.x % qchunks;                     // (sample, head, chunk of 128 queries)
  const int bh = blockIdx.x / qchunks;
  const int h = bh % p.heads, b = bh / p.heads;
  const long rs = 3L * p.C;                                // row stride of qkv
  const float* qbase = p.qkv + (long)b * p.N * rs + h * HD;
  const float* kbase = qbase + p.C;
  const float* vbase = qbase + 2 * p.C;

  // B operand of S^T = K Q^T: lane (query n31, half) holds Q[query][16*half + t], t = 0..15, times scale * log2 e
  const int q = (qc * AW + wave) * 32 + n31;
  const bool qok = q < p.N;
  float qreg[16];
  {
    const float* qr = qbase + (long)(qok ? q : 0) * rs + 16 * half;
    const float sc = p.scale * LOG2E;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qr + 4 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) qreg[4 * j + i] = v[i] * sc;
    }
  }

  // tile loader: thread (row tid>>3, 16-byte chunk tid&7) of the 32 x 32 K and V tiles; rows past N read as zeros
  const int lr = tid >> 3, lc = tid & 7;
  f32x4 knext, vnext;
  auto fetch = [&](int k0) {
    const int key = k0 + lr;
    if (key < p.N) {
      knext = *reinterpret_cast<const f32x4*>(kbase + (long)key * rs + 4 * lc);
      vnext = *reinterpret_cast<const f32x4*>(vbase + (long)key * rs + 4 * lc);
    } else {
      knext = f32x4{0.f, 0.f, 0.f, 0.f};
      vnext = knext;
    }
  };
  auto stash = [&](int stage) {
    *reinterpret_cast<f32x4*>(&lds_k[stage][lr * KS + 4 * lc]) = knext;
    *reinterpret_cast<f32x4*>(&lds_v[stage][lr * VS + 4 * lc]) = vnext;
  };
  fetch(0);
  stash(0);
  __syncthreads();

  f32x16 o;                                                // O^T: rows = d, column = this lane's query
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float* mcol = p.mask ? p.mask + (qok ? q : 0) + 4L * half * p.N : nullptr;

  const int ntiles = (p.N + 31) / 32;
  for (int it = 0; it < ntiles; ++it) {
    const int k0 = it * 32, cur = it & 1;
    const bool more = it + 1 < ntiles;
    if (more) fetch(k0 + 32);
    // ---- mask column of this query: register e is key k0 + (e&3) + 8*(e>>2) + 4*half (symmetric: [key][query] coalesces)
    float mreg[16];
    if (mcol) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * half;
        mreg[e] = key < p.N ? mcol[(long)(k0 + (e & 3) + 8 * (e >> 2)) * p.N] : 0.f;
      }
    }
    // ---- S^T tile: A operand lane (key n31, half) = K[k0 + n31][16*half + t]
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    {
      const float* kr = &lds_k[cur][n31 * KS + 16 * half];
      f32x4 kf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const f32x4*>(kr + 4 * j);
#pragma unroll
      for (int t = 0; t < 16; ++t) s = mfma2(kf[t >> 2][t & 3], qreg[t], s);
    }
    // ---- online softmax in base 2
    if (mcol) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = fmaf(mreg[e], LOG2E, s[e]);
    }
    if (k0 + 32 > p.N) {                                   // ragged last tile: keys past N drop out
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (k0 + (e & 3) + 8 * (e >> 2) + 4 * half >= p.N) s[e] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;   // every key so far masked out: all exponentials are 0
    const float corr = __builtin_amdgcn_exp2f(m_run - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = __builtin_amdgcn_exp2f(s[e] - m_safe);
      psum += s[e];
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] *= corr;
    // ---- O^T += V^T P^T: MFMA e pairs keys (k0(e), k0(e) + 4); A operand lane (d = n31, pair member half)
    {
      const float* vr = &lds_v[cur][4 * half * VS + n31];
#pragma unroll
      for (int e = 0; e < 16; ++e) o = mfma2(vr[((e & 3) + 8 * (e >> 2)) * VS], s[e], o);
    }
    if (more) stash(cur ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: register e of this lane is d = (e&3) + 8*(e>>2) + 4*half of query q
  if (qok) {
    const float inv = 1.f / l_run;
    const long row = (long)b * p.N + q;
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      const f32x4 v = {o[e] * inv, o[e + 1] * inv, o[e + 2] * inv, o[e + 3] * inv};
      const int d0 = 8 * (e >> 2) + 4 * half;
      if (p.out) *reinterpret_cast<f32x4*>(p.out + row * p.C + h * HD + d0) = v;
      if (p.out_hl) {
        f16v4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hi[j] = (_Float16)v[j];
          lo[j] = (_Float16)(v[j] - (float)hi[j]);
        }
        unsigned char* line = p.out_hl + (row * p.heads + h) * 128 + d0 * 2;
        *reinterpret_cast<f16v4*>(line) = hi;
        *reinterpret_cast<f16v4*>(line + 64) = lo;
      }
    }
  }
}

}  // namespace

// out[b][n][h*32 + :] = softmax_m(scale * q[b][n][h] . k[b][m][h] + mask[n][m]) @ v[b][m][h]  for every head h;
// qkv [B][N][3*C] (q | k | v, C = heads * 32), mask [N][N] additive and SYMMETRIC or NULL, out [B][N][C] fp32 and / or
// out_hl32 (HL32 operand of the proj Linear).
// Replaces the q k^T / softmax / attn v chain of modules/svtr.py:140-149 without materialising [B][heads][N][N].
MRN_EXPORT int mrn_svtr_attention_f32(const float* qkv, const float* mask, float* out, void* out_hl32, int B, int N, int C,
                                      int heads, float scale, void* stream) {
  MRN_CHECK_ARG(qkv && (out || out_hl32) && heads >= 1 && C == heads * HD, "mrn_svtr_attention_f32: head dimension must be %d (C=%d heads=%d)", HD, C, heads);
  MRN_CHECK_ARG(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0), "mrn_svtr_attention_f32: operands must be 16-byte aligned");
  if (B == 0 || N == 0) return MRN_OK;
  AttnParams p;
  p.qkv = qkv; p.mask = mask; p.out = out; p.out_hl = (unsigned char*)out_hl32; p.B = B; p.N = N; p.C = C; p.heads = heads; p.scale = scale;
  const long groups = (long)B * heads * ((N + 32 * AW - 1) / (32 * AW));
  hipLaunchKernelGGL(svtr_attention_kernel, dim3((unsigned)groups), dim3(AW * 64), 0, (hipStream_t)stream, p);
  MRN_LAUNCH_CHECK("svtr_attention");
  return MRN_OK;
}
