// Fused multi-head attention of the SVTR mixing blocks (reference modules/svtr.py:90-152: qkv -> q k^T * scale (+ local
// window mask) -> softmax -> attn v), head dimension 32, inference path of the frozen experts.
//
// One wave owns 32 query tokens of one (sample, head) and walks the keys in tiles of 32 with an online softmax, so the
// [B, heads, N, N] score tensor (537 MB per block at B = 256, N = 512) never exists in HBM.  Products run on the exact
// fp32 MFMA (v_mfma_f32_32x32x2_f32).  Scores are computed TRANSPOSED, S^T = K Q^T: in the 32x32 accumulator layout a
// lane then holds one QUERY (column lane & 31) and its 16 registers hold 16 keys, so the softmax reductions over keys
// are register reductions plus one cross-half shuffle, the running rescale of the output is a per-lane scalar, and the
// probabilities P^T already sit in the B-operand layout of the second product O^T = V^T P^T (register e pairs key
// k0(e) = (e&3) + 8(e>>2) in lanes 0-31 with key k0(e) + 4 in lanes 32-63; V is fetched in the same pairing).
// The additive mask must be symmetric (SVTR's local window mask is): mask[key][query] is read row-wise.
#include "common.hpp"

namespace {

constexpr int HD = 32;          // head dimension of every SVTR stage (embed_dim / num_heads = 64/2 = 128/4 = 256/8)
constexpr int AW = 4;           // waves per workgroup: 128 query tokens

struct AttnParams {
  const float* qkv;    // [B][N][3*C]: q | k | v, each (head, d) innermost
  const float* mask;   // [N][N] additive, symmetric, or null
  float* out;          // [B][N][C]
  int B, N, C, heads;
  float scale;
};

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(AW * 64) void svtr_attention_kernel(const AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n31 = lane & 31, half = lane >> 5;
  const int qblocks = (p.N + 31) / 32;
  const int wid = blockIdx.x * AW + wave;                  // (sample, head, query block)
  const int qb = wid % qblocks;
  const int bh = wid / qblocks;
  if (bh >= p.B * p.heads) return;
  const int h = bh % p.heads, b = bh / p.heads;
  const long rs = 3L * p.C;                                // row stride of qkv
  const float* qbase = p.qkv + (long)b * p.N * rs + h * HD;
  const float* kbase = qbase + p.C;
  const float* vbase = qbase + 2 * p.C;

  // B operand of S^T = K Q^T: lane (query n31, d parity half) holds Q[query][2s + half], s = 0..15
  const int q = qb * 32 + n31;
  const bool qok = q < p.N;
  float qreg[16];
  {
    const float* qr = qbase + (long)(qok ? q : 0) * rs;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qr + 4 * j);
      qreg[2 * j] = (half ? v[1] : v[0]) * p.scale;
      qreg[2 * j + 1] = (half ? v[3] : v[2]) * p.scale;
    }
  }

  f32x16 o;                                                // O^T: rows = d, column = this lane's query
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < p.N; k0 += 32) {
    // ---- S^T tile: A operand lane (key n31, d parity half) = K[k0 + n31][2s + half]
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    {
      const int key = k0 + n31;
      const float* kr = kbase + (long)(key < p.N ? key : 0) * rs;
      float kreg[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(kr + 4 * j);
        kreg[2 * j] = half ? v[1] : v[0];
        kreg[2 * j + 1] = half ? v[3] : v[2];
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) s = mfma2(kreg[t], qreg[t], s);
    }
    // ---- mask + online softmax: register e of this lane is key k0 + (e&3) + 8*(e>>2) + 4*half for query q
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * half;
      float v = s[e];
      if (key >= p.N) v = -INFINITY;
      else if (p.mask && qok) v += p.mask[(long)key * p.N + q];           // symmetric: [key][query] is coalesced
      s[e] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float corr = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);   // (m_run = -inf: exp(-inf) = 0)
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = (m_new == -INFINITY) ? 0.f : __expf(s[e] - m_new);
      s[e] = pe;
      psum += pe;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] *= corr;
    // ---- O^T += V^T P^T: MFMA e pairs keys (k0(e), k0(e) + 4); A operand lane (d = n31, pair member half)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * half;
      const float vv = key < p.N ? vbase[(long)key * rs + n31] : 0.f;
      o = mfma2(vv, s[e], o);
    }
  }

  // ---- normalise and store: register e of this lane is d = (e&3) + 8*(e>>2) + 4*half of query q
  if (qok) {
    const float inv = 1.f / l_run;
    float* orow = p.out + ((long)b * p.N + q) * p.C + h * HD;
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      f32x4 v = {o[e] * inv, o[e + 1] * inv, o[e + 2] * inv, o[e + 3] * inv};
      *reinterpret_cast<f32x4*>(orow + 8 * (e >> 2) + 4 * half) = v;
    }
  }
}

}  // namespace

// out[b][n][h*32 + :] = softmax_m(scale * q[b][n][h] . k[b][m][h] + mask[n][m]) @ v[b][m][h]  for every head h;
// qkv [B][N][3*C] (q | k | v, C = heads * 32), mask [N][N] additive and SYMMETRIC or NULL, out [B][N][C].
// Replaces the q k^T / softmax / attn v chain of modules/svtr.py:140-149 without materialising [B][heads][N][N].
MRN_EXPORT int mrn_svtr_attention_f32(const float* qkv, const float* mask, float* out, int B, int N, int C, int heads,
                                      float scale, void* stream) {
  MRN_CHECK_ARG(qkv && out && heads >= 1 && C == heads * HD, "mrn_svtr_attention_f32: head dimension must be %d (C=%d heads=%d)", HD, C, heads);
  MRN_CHECK_ARG(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0), "mrn_svtr_attention_f32: operands must be 16-byte aligned");
  if (B == 0 || N == 0) return MRN_OK;
  AttnParams p;
  p.qkv = qkv; p.mask = mask; p.out = out; p.B = B; p.N = N; p.C = C; p.heads = heads; p.scale = scale;
  const long waves = (long)B * heads * ((N + 31) / 32);
  hipLaunchKernelGGL(svtr_attention_kernel, dim3((unsigned)((waves + AW - 1) / AW)), dim3(AW * 64), 0, (hipStream_t)stream, p);
  MRN_LAUNCH_CHECK("svtr_attention");
  return MRN_OK;
}
