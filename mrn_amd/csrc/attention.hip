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
  const unsigned* mask_bits;   // [N][ceil(N/32)] bit k of word t of row q: key 32t + k is visible to query q (a 0 / -inf mask), or null
  float* out;          // [B][N][C] fp32, or null
  unsigned char* out_hl;   // the same tensor as HL32 lines [B*N][heads][hi 32 | lo 32] (a head is one 32-channel block), or null
  float* lse;          // [B][heads][N] base-2 log-sum-exp of the scaled, masked scores (kept for the backward pass), or null
  const float* hl_scale;   // {s, 1/s}: out_hl = split(s * out) -- the range scale of the proj Linear's operand in an expert being trained -- or null
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
  const unsigned* brow = p.mask_bits ? p.mask_bits + (long)(qok ? q : 0) * ((p.N + 31) / 32) : nullptr;

  const int ntiles = (p.N + 31) / 32;
  for (int it = 0; it < ntiles; ++it) {
    const int k0 = it * 32, cur = it & 1;
    const bool more = it + 1 < ntiles;
    if (more) fetch(k0 + 32);
    const unsigned bword = brow ? brow[it] : 0xffffffffu;
    if (__ballot(bword != 0u) == 0) {                      // no query of this wave sees a key of this tile (local window): skip
      if (more) stash(cur ^ 1);
      __syncthreads();
      continue;
    }
    const unsigned bw = bword >> (4 * half);
    // ---- mask column of this query: register e is key k0 + (e&3) + 8*(e>>2) + 4*half (symmetric: [key][query] coalesces)
    float mreg[16];
    if (mcol) {      // rows past N are clamped, not branched around (their scores are dropped below): 16 plain loads
      const int last = p.N - 1 - 4 * half;
#pragma unroll
      for (int e = 0; e < 16; ++e) mreg[e] = mcol[(long)min(k0 + (e & 3) + 8 * (e >> 2), last) * p.N];
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
    if (brow) {                                            // one mask word per (query, key tile) instead of 16 floats
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (!((bw >> ((e & 3) + 8 * (e >> 2))) & 1u)) s[e] = -INFINITY;
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
    const float hs = p.hl_scale ? p.hl_scale[0] : 1.f;
    if (p.lse && half == 0) p.lse[(long)bh * p.N + q] = m_run + __builtin_amdgcn_logf(l_run);     // (v_log_f32 is log2)
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      const f32x4 v = {o[e] * inv, o[e + 1] * inv, o[e + 2] * inv, o[e + 3] * inv};
      const int d0 = 8 * (e >> 2) + 4 * half;
      if (p.out) *reinterpret_cast<f32x4*>(p.out + row * p.C + h * HD + d0) = v;
      if (p.out_hl) {
        f16v4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          _Float16 hh, ll;
          split_f16_sat(v[j] * hs, hh, ll);
          hi[j] = hh;
          lo[j] = ll;
        }
        unsigned char* line = p.out_hl + (row * p.heads + h) * 128 + d0 * 2;
        *reinterpret_cast<f16v4*>(line) = hi;
        *reinterpret_cast<f16v4*>(line + 64) = lo;
      }
    }
  }
}

// ---- split-fp16 x3 variant for the frozen experts ----------------------------------------------------------------------------
// Same schedule as svtr_attention_kernel, both products as hi/lo fp16 pairs on v_mfma_f32_32x32x16_f16 (lo*hi + hi*lo + hi*hi,
// fp32 accumulate: 22-bit products like every other GEMM of the frozen experts) -- 12 MFMAs of 8 passes per key tile instead
// of 32 of 16 passes, which leaves the softmax VALU work as the bound.  K tiles are staged as HL32 lines [hi 32 | lo 32] per
// key, V tiles TRANSPOSED as one line per d with the keys in the order the score accumulator holds them (lane half h, register
// e -> slot 16h + e), so P^T goes from the first product's accumulator into the second product's B operand by a register
// pack.  The probabilities carry a 2^12 bias (exp2(s - m + 12), cancelled by the final 1 / l) to keep small ones out of
// fp16's subnormal range.
typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
constexpr int XS = 144;         // LDS line stride in bytes (128-byte line + 16: conflict-free 16-byte fragment reads)
constexpr float PBIAS = 12.f;
constexpr float OPSCALE = 64.f;  // q, k, v are split as 64 * x: the lo halves of O(1) operands stay out of fp16's subnormal range

__device__ __forceinline__ f32x16 mfma16h(f16v8 a, f16v8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split2(float v, _Float16& h, _Float16& l) {
  split_f16(v, h, l);
}

__global__ __launch_bounds__(AW * 64) void svtr_attention_x3_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_k[2][32 * XS];     // [key][hi d 0..31 | lo d 0..31]
  __shared__ __attribute__((aligned(16))) unsigned char lds_v[2][32 * XS];     // [d][hi slot 0..31 | lo slot 0..31]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n31 = lane & 31, half = lane >> 5;
  const int qchunks = (p.N + 32 * AW - 1) / (32 * AW);
  const int qc = blockIdx.x % qchunks;
  const int bh = blockIdx.x / qchunks;
  const int h = bh % p.heads, b = bh / p.heads;
  const long rs = 3L * p.C;
  const float* qbase = p.qkv + (long)b * p.N * rs + h * HD;
  const float* kbase = qbase + p.C;
  const float* vbase = qbase + 2 * p.C;
  const int q = (qc * AW + wave) * 32 + n31;
  const bool qok = q < p.N;

  // B operand of S^T = K Q^T: MFMA m, lane (query n31, half), slot s holds d = 16m + 8*half + s of scale * log2e * q
  f16v8 qh[2], ql[2];
  {
    const float* qr = qbase + (long)(qok ? q : 0) * rs;
    const float sc = p.scale * LOG2E * OPSCALE;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qr + 16 * m + 8 * half);
      const f32x4 c = *reinterpret_cast<const f32x4*>(qr + 16 * m + 8 * half + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        _Float16 hh, ll;
        split2(a[i] * sc, hh, ll); qh[m][i] = hh; ql[m][i] = ll;
        split2(c[i] * sc, hh, ll); qh[m][4 + i] = hh; ql[m][4 + i] = ll;
      }
    }
  }

  // tile loader: thread (key row tid>>3, d chunk 4*(tid&7) .. +3); V goes in transposed, key -> slot 16*((key>>2)&1) + (key&3) + 4*(key>>3)
  const int lr = tid >> 3, lc = tid & 7;
  const int vslot = 16 * ((lr >> 2) & 1) + (lr & 3) + 4 * (lr >> 3);
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
    f16v4 kh, kl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      _Float16 hh, ll;
      split2(knext[i] * OPSCALE, hh, ll); kh[i] = hh; kl[i] = ll;
      split2(vnext[i] * OPSCALE, hh, ll);
      unsigned char* vline = &lds_v[stage][(4 * lc + i) * XS + vslot * 2];
      *reinterpret_cast<_Float16*>(vline) = hh;
      *reinterpret_cast<_Float16*>(vline + 64) = ll;
    }
    *reinterpret_cast<f16v4*>(&lds_k[stage][lr * XS + lc * 8]) = kh;
    *reinterpret_cast<f16v4*>(&lds_k[stage][lr * XS + 64 + lc * 8]) = kl;
  };
  fetch(0);
  stash(0);
  __syncthreads();

  f32x16 o;
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float* mcol = p.mask ? p.mask + (qok ? q : 0) + 4L * half * p.N : nullptr;
  const unsigned* brow = p.mask_bits ? p.mask_bits + (long)(qok ? q : 0) * ((p.N + 31) / 32) : nullptr;

  const int ntiles = (p.N + 31) / 32;
  for (int it = 0; it < ntiles; ++it) {
    const int k0 = it * 32, cur = it & 1;
    const bool more = it + 1 < ntiles;
    if (more) fetch(k0 + 32);
    const unsigned bword = brow ? brow[it] : 0xffffffffu;
    if (__ballot(bword != 0u) == 0) {                      // no query of this wave sees a key of this tile (local window): skip
      if (more) stash(cur ^ 1);
      __syncthreads();
      continue;
    }
    const unsigned bw = bword >> (4 * half);
    float mreg[16];
    if (mcol) {
      const int last = p.N - 1 - 4 * half;
#pragma unroll
      for (int e = 0; e < 16; ++e) mreg[e] = mcol[(long)min(k0 + (e & 3) + 8 * (e >> 2), last) * p.N];
    }
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    {
      const unsigned char* kl_ = &lds_k[cur][n31 * XS + 16 * half];
      f16v8 kh[2], kl[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        kh[m] = *reinterpret_cast<const f16v8*>(kl_ + 32 * m);
        kl[m] = *reinterpret_cast<const f16v8*>(kl_ + 64 + 32 * m);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) s = mfma16h(kl[m], qh[m], s);
#pragma unroll
      for (int m = 0; m < 2; ++m) s = mfma16h(kh[m], ql[m], s);
#pragma unroll
      for (int m = 0; m < 2; ++m) s = mfma16h(kh[m], qh[m], s);
    }
    if (mcol) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = fmaf(s[e], 1.f / (OPSCALE * OPSCALE), mreg[e] * LOG2E);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] *= 1.f / (OPSCALE * OPSCALE);
    }
    if (brow) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (!((bw >> ((e & 3) + 8 * (e >> 2))) & 1u)) s[e] = -INFINITY;
    }
    if (k0 + 32 > p.N) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (k0 + (e & 3) + 8 * (e >> 2) + 4 * half >= p.N) s[e] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float corr = __builtin_amdgcn_exp2f(m_run - m_safe);
    const float mb = m_safe - PBIAS;
    float psum = 0.f;
    f16v8 ph[2], pl[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = __builtin_amdgcn_exp2f(s[e] - mb);
      psum += pe;
      _Float16 hh, ll;
      split2(pe, hh, ll);
      ph[e >> 3][e & 7] = hh;
      pl[e >> 3][e & 7] = ll;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] *= corr;
    {
      const unsigned char* vl_ = &lds_v[cur][n31 * XS + 32 * half];
      f16v8 vh[2], vl[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        vh[m] = *reinterpret_cast<const f16v8*>(vl_ + 16 * m);
        vl[m] = *reinterpret_cast<const f16v8*>(vl_ + 64 + 16 * m);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) o = mfma16h(vl[m], ph[m], o);
#pragma unroll
      for (int m = 0; m < 2; ++m) o = mfma16h(vh[m], pl[m], o);
#pragma unroll
      for (int m = 0; m < 2; ++m) o = mfma16h(vh[m], ph[m], o);
    }
    if (more) stash(cur ^ 1);
    __syncthreads();
  }

  if (qok) {
    const float inv = 1.f / (l_run * OPSCALE);
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
          _Float16 hh, ll;
          split_f16(v[j], hh, ll);
          hi[j] = hh;
          lo[j] = ll;
        }
        unsigned char* line = p.out_hl + (row * p.heads + h) * 128 + d0 * 2;
        *reinterpret_cast<f16v4*>(line) = hi;
        *reinterpret_cast<f16v4*>(line + 64) = lo;
      }
    }
  }
}

// ---- backward (expert training, loop A) ------------------------------------------------------------------------------------
// Flash-style: nothing of size N x N is kept from the forward pass, only the per-query log-sum-exp.  Two kernels recompute
// the probabilities tile by tile:
//   dq kernel    (workgroup = 128 queries, wave = 32 queries, walks the keys; same orientation as the forward kernel):
//                P^T = exp2(S^T - lse), dP^T = V dO^T, dS^T = P^T o (dP^T - D), dQ^T += K^T dS^T; also writes D = rowsum(dO o O)
//   dk/dv kernel (workgroup = 128 keys, wave = 32 keys, walks the queries; a lane owns one KEY):
//                P = exp2(S - lse), dV^T += dO^T P, dP = dO V^T, dS = P o (dP - D), dK^T += Q^T dS
// Every product is on v_mfma_f32_32x32x2_f32; the probabilities / score gradients always come out of one product in exactly
// the B-operand layout of the next.
struct AttnBwdParams {
  const float* qkv;    // [B][N][3*C]
  const float* mask;   // [N][N] additive, symmetric, or null
  const float* out;    // [B][N][C] forward result
  const float* dout;   // [B][N][C]
  const float* lse;    // [B][heads][N] from the forward pass (base 2)
  float* dsum;         // [B][heads][N] workspace: D = rowsum(dO o O), written by the dq kernel, read by the dk/dv kernel
  float* dqkv;         // [B][N][3*C]
  unsigned* amax_ws;   // optional: max|dqkv| folded into 64 words (the range of the qkv Linear's gradient operand), see mrn_pow2_finalize_f32
  int B, N, C, heads;
  float scale;
};

// one atomic per wave into slot (wave id % 64)
__device__ __forceinline__ void fold_amax(unsigned* ws, float m, int lane, int slot) {
  m = wave_max(m);
  if (lane == 0 && !(m <= 0.f)) atomicMax(ws + (slot & 63), __float_as_uint(m));
}

constexpr int RS = 40;          // LDS row stride of tiles that are read both by rows (b128) and by columns
constexpr int kKeyOf(int e) { return (e & 3) + 8 * (e >> 2); }

__global__ __launch_bounds__(AW * 64) void svtr_attention_dq_kernel(const AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) float lds_k[2][32 * RS];
  __shared__ __attribute__((aligned(16))) float lds_v[2][32 * KS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n31 = lane & 31, half = lane >> 5;
  const int qchunks = (p.N + 32 * AW - 1) / (32 * AW);
  const int qc = blockIdx.x % qchunks;
  const int bh = blockIdx.x / qchunks;
  const int h = bh % p.heads, b = bh / p.heads;
  const long rs = 3L * p.C;
  const float* qbase = p.qkv + (long)b * p.N * rs + h * HD;
  const float* kbase = qbase + p.C;
  const float* vbase = qbase + 2 * p.C;
  const int q = (qc * AW + wave) * 32 + n31;
  const bool qok = q < p.N;
  const int qs = qok ? q : 0;

  float qreg[16], doreg[16];
  float dsum_q;
  {
    const float* qr = qbase + (long)qs * rs + 16 * half;
    const float* dr = p.dout + ((long)b * p.N + qs) * p.C + h * HD + 16 * half;
    const float* orow = p.out + ((long)b * p.N + qs) * p.C + h * HD + 16 * half;
    const float sc = p.scale * LOG2E;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qr + 4 * j);
      const f32x4 g = *reinterpret_cast<const f32x4*>(dr + 4 * j);
      const f32x4 o = *reinterpret_cast<const f32x4*>(orow + 4 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        qreg[4 * j + i] = v[i] * sc;
        doreg[4 * j + i] = g[i];
        part = fmaf(g[i], o[i], part);
      }
    }
    dsum_q = part + __shfl_xor(part, 32);
    if (qok && half == 0) p.dsum[(long)bh * p.N + q] = dsum_q;
  }
  const float lse_q = p.lse[(long)bh * p.N + qs];

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
    *reinterpret_cast<f32x4*>(&lds_k[stage][lr * RS + 4 * lc]) = knext;
    *reinterpret_cast<f32x4*>(&lds_v[stage][lr * KS + 4 * lc]) = vnext;
  };
  fetch(0);
  stash(0);
  __syncthreads();

  f32x16 dq;
#pragma unroll
  for (int e = 0; e < 16; ++e) dq[e] = 0.f;
  const float* mcol = p.mask ? p.mask + qs + 4L * half * p.N : nullptr;

  const int ntiles = (p.N + 31) / 32;
  for (int it = 0; it < ntiles; ++it) {
    const int k0 = it * 32, cur = it & 1;
    const bool more = it + 1 < ntiles;
    if (more) fetch(k0 + 32);
    float mreg[16];
    if (mcol) {
      const int last = p.N - 1 - 4 * half;
#pragma unroll
      for (int e = 0; e < 16; ++e) mreg[e] = mcol[(long)min(k0 + kKeyOf(e), last) * p.N];
    }
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
    {
      const float* kr = &lds_k[cur][n31 * RS + 16 * half];
      const float* vr = &lds_v[cur][n31 * KS + 16 * half];
      f32x4 kf[4], vf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kf[j] = *reinterpret_cast<const f32x4*>(kr + 4 * j);
        vf[j] = *reinterpret_cast<const f32x4*>(vr + 4 * j);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) s = mfma2(kf[t >> 2][t & 3], qreg[t], s);        // S^T = K Q^T
#pragma unroll
      for (int t = 0; t < 16; ++t) dp = mfma2(vf[t >> 2][t & 3], doreg[t], dp);     // dP^T = V dO^T
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = s[e];
      if (mcol) v = fmaf(mreg[e], LOG2E, v);
      if (k0 + kKeyOf(e) + 4 * half >= p.N) v = -INFINITY;
      const float pe = __builtin_amdgcn_exp2f(v - lse_q);
      s[e] = pe * (dp[e] - dsum_q);                                                  // dS^T
    }
    {
      const float* kc = &lds_k[cur][4 * half * RS + n31];
#pragma unroll
      for (int e = 0; e < 16; ++e) dq = mfma2(kc[kKeyOf(e) * RS], s[e], dq);         // dQ^T += K^T dS^T
    }
    if (more) stash(cur ^ 1);
    __syncthreads();
  }
  float amx = 0.f;
  if (qok) {
    float* drow = p.dqkv + ((long)b * p.N + q) * rs + h * HD;
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      const f32x4 v = {dq[e] * p.scale, dq[e + 1] * p.scale, dq[e + 2] * p.scale, dq[e + 3] * p.scale};
      *reinterpret_cast<f32x4*>(drow + 8 * (e >> 2) + 4 * half) = v;
      amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
  }
  if (p.amax_ws) fold_amax(p.amax_ws, amx, lane, blockIdx.x * AW + wave);
}

__global__ __launch_bounds__(AW * 64) void svtr_attention_dkv_kernel(const AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) float lds_q[2][32 * RS];
  __shared__ __attribute__((aligned(16))) float lds_g[2][32 * RS];      // dO tile
  __shared__ float lds_l[2][32], lds_d[2][32];                          // lse, D of the tile's queries
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n31 = lane & 31, half = lane >> 5;
  const int kchunks = (p.N + 32 * AW - 1) / (32 * AW);
  const int kc = blockIdx.x % kchunks;
  const int bh = blockIdx.x / kchunks;
  const int h = bh % p.heads, b = bh / p.heads;
  const long rs = 3L * p.C;
  const float* qbase = p.qkv + (long)b * p.N * rs + h * HD;
  const float* gbase = p.dout + (long)b * p.N * p.C + h * HD;
  const int key = (kc * AW + wave) * 32 + n31;
  const bool kok = key < p.N;
  const int ks = kok ? key : 0;

  float kreg[16], vreg[16];
  {
    const float* kr = qbase + p.C + (long)ks * rs + 16 * half;
    const float* vr = qbase + 2 * p.C + (long)ks * rs + 16 * half;
    const float sc = p.scale * LOG2E;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(kr + 4 * j);
      const f32x4 c = *reinterpret_cast<const f32x4*>(vr + 4 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kreg[4 * j + i] = a[i] * sc;
        vreg[4 * j + i] = c[i];
      }
    }
  }

  const int lr = tid >> 3, lc = tid & 7;
  f32x4 qnext, gnext;
  float lnext = 0.f, dnext = 0.f;
  auto fetch = [&](int q0) {
    const int qq = q0 + lr;
    if (qq < p.N) {
      qnext = *reinterpret_cast<const f32x4*>(qbase + (long)qq * rs + 4 * lc);
      gnext = *reinterpret_cast<const f32x4*>(gbase + (long)qq * p.C + 4 * lc);
    } else {
      qnext = f32x4{0.f, 0.f, 0.f, 0.f};
      gnext = qnext;
    }
    if (tid < 32) {
      const bool ok = q0 + tid < p.N;
      lnext = ok ? p.lse[(long)bh * p.N + q0 + tid] : INFINITY;      // queries past N: probability 0
      dnext = ok ? p.dsum[(long)bh * p.N + q0 + tid] : 0.f;
    }
  };
  auto stash = [&](int stage) {
    *reinterpret_cast<f32x4*>(&lds_q[stage][lr * RS + 4 * lc]) = qnext;
    *reinterpret_cast<f32x4*>(&lds_g[stage][lr * RS + 4 * lc]) = gnext;
    if (tid < 32) { lds_l[stage][tid] = lnext; lds_d[stage][tid] = dnext; }
  };
  fetch(0);
  stash(0);
  __syncthreads();

  f32x16 dk, dv;
#pragma unroll
  for (int e = 0; e < 16; ++e) dk[e] = dv[e] = 0.f;
  const float* mrow = p.mask ? p.mask + ks : nullptr;

  const int ntiles = (p.N + 31) / 32;
  for (int it = 0; it < ntiles; ++it) {
    const int q0 = it * 32, cur = it & 1;
    const bool more = it + 1 < ntiles;
    if (more) fetch(q0 + 32);
    float mreg[16];
    if (mrow) {      // queries past N are clamped (their probability is 0 through lse = +inf)
#pragma unroll
      for (int e = 0; e < 16; ++e) mreg[e] = mrow[(long)min(q0 + kKeyOf(e) + 4 * half, p.N - 1) * p.N];
    }
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
    {
      const float* qr = &lds_q[cur][n31 * RS + 16 * half];
      const float* gr = &lds_g[cur][n31 * RS + 16 * half];
      f32x4 qf[4], gf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qf[j] = *reinterpret_cast<const f32x4*>(qr + 4 * j);
        gf[j] = *reinterpret_cast<const f32x4*>(gr + 4 * j);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) s = mfma2(qf[t >> 2][t & 3], kreg[t], s);         // S = Q K^T   (rows = queries, lane = key)
#pragma unroll
      for (int t = 0; t < 16; ++t) dp = mfma2(gf[t >> 2][t & 3], vreg[t], dp);       // dP = dO V^T
    }
    {
      const float* lq = &lds_l[cur][4 * half];
      const float* dq_ = &lds_d[cur][4 * half];
      const float* gc = &lds_g[cur][4 * half * RS + n31];
      const float* qc = &lds_q[cur][4 * half * RS + n31];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = s[e];
        if (mrow) v = fmaf(mreg[e], LOG2E, v);
        const float pe = __builtin_amdgcn_exp2f(v - lq[kKeyOf(e)]);
        dv = mfma2(gc[kKeyOf(e) * RS], pe, dv);                                      // dV^T += dO^T P
        s[e] = pe * (dp[e] - dq_[kKeyOf(e)]);                                        // dS
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) dk = mfma2(qc[kKeyOf(e) * RS], s[e], dk);         // dK^T += Q^T dS
    }
    if (more) stash(cur ^ 1);
    __syncthreads();
  }
  float amx = 0.f;
  if (kok) {
    float* krow = p.dqkv + ((long)b * p.N + key) * rs + p.C + h * HD;
    float* vrow = krow + p.C;
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      const f32x4 a = {dk[e] * p.scale, dk[e + 1] * p.scale, dk[e + 2] * p.scale, dk[e + 3] * p.scale};
      const f32x4 c = {dv[e], dv[e + 1], dv[e + 2], dv[e + 3]};
      *reinterpret_cast<f32x4*>(krow + 8 * (e >> 2) + 4 * half) = a;
      *reinterpret_cast<f32x4*>(vrow + 8 * (e >> 2) + 4 * half) = c;
#pragma unroll
      for (int j = 0; j < 4; ++j) amx = fmaxf(amx, fmaxf(fabsf(a[j]), fabsf(c[j])));
    }
  }
  if (p.amax_ws) fold_amax(p.amax_ws, amx, lane, blockIdx.x * AW + wave);
}

}  // namespace

// out[b][n][h*32 + :] = softmax_m(scale * q[b][n][h] . k[b][m][h] + mask[n][m]) @ v[b][m][h]  for every head h;
// qkv [B][N][3*C] (q | k | v, C = heads * 32), mask [N][N] additive and SYMMETRIC or NULL, out [B][N][C] fp32 and / or
// out_hl32 (HL32 operand of the proj Linear).
// Replaces the q k^T / softmax / attn v chain of modules/svtr.py:140-149 without materialising [B][heads][N][N].
// x3 != 0: both products as split-fp16 x3 (frozen experts; lse must be NULL), else exact fp32 products.
// mask_bits (optional, instead of mask): [N][ceil(N/32)] visibility bits of a 0 / -inf mask (bit k of word t of row q = key
// 32t + k is visible to query q): one word per (query, key tile) instead of 16 floats per lane.
// hl_scale (optional, fp32 kernel only): {s, 1/s} -- out_hl32 = split(s * out), the range-scaled operand of the proj Linear of an expert
// being trained; any s with s * max|v| <= the fp16 headroom serves (a row of out is a convex combination of rows of v), e.g. the scale
// of max|qkv| from the qkv GEMM's epilogue (mrn_conv2d_x3_hl32 amax_ws)
MRN_EXPORT int mrn_svtr_attention_f32(const float* qkv, const float* mask, const void* mask_bits, float* out, void* out_hl32,
                                      float* lse, int B, int N, int C, int heads, float scale, int x3, const float* hl_scale, void* stream) {
  MRN_CHECK_ARG(!(mask && mask_bits), "mrn_svtr_attention_f32: pass the additive mask or its bit form, not both");
  MRN_CHECK_ARG(!(x3 && hl_scale), "mrn_svtr_attention_f32: the x3 variant (frozen experts) writes the unscaled operand");
  MRN_CHECK_ARG(!(x3 && lse), "mrn_svtr_attention_f32: the x3 variant keeps no log-sum-exp (training uses the fp32 kernel)");
  MRN_CHECK_ARG(qkv && (out || out_hl32) && heads >= 1 && C == heads * HD, "mrn_svtr_attention_f32: head dimension must be %d (C=%d heads=%d)", HD, C, heads);
  MRN_CHECK_ARG(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0), "mrn_svtr_attention_f32: operands must be 16-byte aligned");
  if (B == 0 || N == 0) return MRN_OK;
  AttnParams p;
  p.qkv = qkv; p.mask = mask; p.mask_bits = (const unsigned*)mask_bits; p.out = out; p.out_hl = (unsigned char*)out_hl32; p.lse = lse; p.hl_scale = hl_scale; p.B = B; p.N = N; p.C = C; p.heads = heads; p.scale = scale;
  const long groups = (long)B * heads * ((N + 32 * AW - 1) / (32 * AW));
  if (x3) hipLaunchKernelGGL(svtr_attention_x3_kernel, dim3((unsigned)groups), dim3(AW * 64), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(svtr_attention_kernel, dim3((unsigned)groups), dim3(AW * 64), 0, (hipStream_t)stream, p);
  MRN_LAUNCH_CHECK("svtr_attention");
  return MRN_OK;
}

// Backward of mrn_svtr_attention_f32 for an expert being trained: dqkv [B][N][3*C] from dout, the forward result `out` and the
// forward pass's log-sum-exp `lse` [B][heads][N]; dsum: [B][heads][N] floats of workspace.  No N x N tensor is stored or read.
MRN_EXPORT int mrn_svtr_attention_bwd_f32(const float* qkv, const float* mask, const float* out, const float* dout,
                                          const float* lse, float* dsum, float* dqkv, int B, int N, int C, int heads,
                                          float scale, void* amax_ws, void* stream) {
  MRN_CHECK_ARG(qkv && out && dout && lse && dsum && dqkv && heads >= 1 && C == heads * HD,
                "mrn_svtr_attention_bwd_f32: bad operands (C=%d heads=%d)", C, heads);
  MRN_CHECK_ARG(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)dqkv % 16 == 0),
                "mrn_svtr_attention_bwd_f32: operands must be 16-byte aligned");
  if (B == 0 || N == 0) return MRN_OK;
  AttnBwdParams p;
  p.qkv = qkv; p.mask = mask; p.out = out; p.dout = dout; p.lse = lse; p.dsum = dsum; p.dqkv = dqkv; p.amax_ws = (unsigned*)amax_ws;
  p.B = B; p.N = N; p.C = C; p.heads = heads; p.scale = scale;
  const long groups = (long)B * heads * ((N + 32 * AW - 1) / (32 * AW));
  hipLaunchKernelGGL(svtr_attention_dq_kernel, dim3((unsigned)groups), dim3(AW * 64), 0, (hipStream_t)stream, p);
  hipLaunchKernelGGL(svtr_attention_dkv_kernel, dim3((unsigned)groups), dim3(AW * 64), 0, (hipStream_t)stream, p);
  MRN_LAUNCH_CHECK("svtr_attention_bwd");
  return MRN_OK;
}
