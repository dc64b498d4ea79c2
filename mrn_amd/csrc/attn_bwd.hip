// Backward of the attention decoder (teacher-forced), TPS sampler and small companions.
//
// Reference: loss.backward() (il_modules/mrn.py:260-261) through modules/prediction.py:58-68,102-118 (26 steps of
// additive attention + LSTMCell) and modules/transformation.py:33-44,204-216.  Same workgroup geometry as the
// forward (rnn.hip): 16 samples per workgroup, 16 waves, wave w owns hidden units [16w, 16w+16); the transposed
// recurrent weights stream fragment-major from L2.
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int HID = 256, BT = 16, NW = 16, NTH = NW * 64;
constexpr int HLD = HID + 4, GLD = 4 * HID + 4;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc[g] += A(16 x K in LDS) . Wg^T with fragment-major weights; G weight sets share the A fragment reads
template <int G>
__device__ __forceinline__ void mma_shared_a(f32x4 (&acc)[G], const float* __restrict__ a_lds, int lda,
                                             const f32x4* const (&wp)[G], int Q, int lane) {
  const float* ap = a_lds + (lane & 15) * lda + (lane >> 4) * 4;
  f32x4 wv[G];
#pragma unroll
  for (int g = 0; g < G; ++g) wv[g] = wp[g][0];
#pragma unroll 1
  for (int q = 0; q < Q; ++q) {
    f32x4 wn[G];
    const long qn = (q + 1 < Q) ? q + 1 : q;
#pragma unroll
    for (int g = 0; g < G; ++g) wn[g] = wp[g][qn * 64];
    const f32x4 av = *reinterpret_cast<const f32x4*>(ap + q * 16);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = mfma4(av[r], wv[g][r], acc[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) wv[g] = wn[g];
  }
}

// the same with the A rows as fp16 hi / lo planes (halves, row stride ld) and fragment-major fp16 hi / lo weight streams
// (ops.pack_fragment_major_h): split-fp16 x3 on v_mfma_f32_16x16x32_f16, Q32 = K / 32 steps
typedef _Float16 f16v8b __attribute__((ext_vector_type(8)));
template <int G>
__device__ __forceinline__ void mma_shared_a_h(f32x4 (&acc)[G], const _Float16* __restrict__ a_hi, const _Float16* __restrict__ a_lo, int ld,
                                               const f16v8b* const (&wp)[G], int Q32, int lane) {
  const _Float16* ah = a_hi + (lane & 15) * ld + (lane >> 4) * 8;
  const _Float16* al = a_lo + (lane & 15) * ld + (lane >> 4) * 8;
  f16v8b wh[G], wl[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { wh[g] = wp[g][0]; wl[g] = wp[g][1]; }
#pragma unroll 1
  for (int q = 0; q < Q32; ++q) {
    f16v8b nh[G], nl[G];
    const long qn = (q + 1 < Q32) ? q + 1 : q;
#pragma unroll
    for (int g = 0; g < G; ++g) { nh[g] = wp[g][qn * 128]; nl[g] = wp[g][qn * 128 + 1]; }
    const f16v8b xh = *reinterpret_cast<const f16v8b*>(ah + q * 32), xl = *reinterpret_cast<const f16v8b*>(al + q * 32);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh[g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl[g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh[g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) { wh[g] = nh[g]; wl[g] = nl[g]; }
  }
}

struct AttnBwdParams {
  const float *Hb, *Hproj, *alpha, *gates, *cseq, *ctx, *hp, *dhid, *w_score;
  const float *w_h2hT, *w_ih_ctxT, *w_hhT;
  float *dgates, *dhp, *dHb, *dHproj, *dws_part;
  float *dctx, *de;       // [B][S][D] d context and [B][S][T] d score (pre-softmax) of every step: the operands of the two deferred passes below
  int B, T, D, S;
  int vb;      // samples per workgroup (16, 8 or 4): rows >= vb of the 16-row MFMA tile are treated like rows beyond the batch
  const float* w_inv;     // x3 form: device float[3] = 1 / prescale of W_h2h^T, W_ih_ctx^T, W_hh^T (then fp16 hi / lo fragment-major streams)
  const float* gscale;    // x3 form: device float[2] = {s, 1/s}: the gate / hp gradients are split as s * value
};

__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(2.f * x);
  return 1.f - 2.f / (e + 1.f);
}

// X3: the three transposed products (dgates . W_ih_ctx, dgates . W_hh, dhp . W_h2h) as split-fp16 x3 -- the gate / hp gradients live in
// LDS as fp16 hi / lo planes of gscale * value.  On the exact-fp32 pipe they are 30 us of a step (576 MFMAs of 32 cycles per wave).
constexpr int GLDH = 4 * HID + 8, HLDH = HID + 8;   // fp16 rows (halves)
template <bool X3>
__global__ __launch_bounds__(NTH) void attn_decoder_bwd_kernel(const AttnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int T = p.T;
  float* dg_lds = lds;                       // [BT][GLD]   gate gradients (A operand of the two transposed products)   (x3: 2 fp16 planes [BT][GLDH])
  float* dctx_lds = X3 ? lds + (2 * BT * GLDH) / 2 : dg_lds + BT * GLD;       // [BT][HLD]
  float* dhp_lds = dctx_lds + BT * HLD;      // [BT][HLD]                                                            (x3: 2 fp16 planes [BT][HLDH])
  float* hp_lds = X3 ? dhp_lds + (2 * BT * HLDH) / 2 : dhp_lds + BT * HLD;        // [BT][HLD]
  _Float16* dg_hi = reinterpret_cast<_Float16*>(dg_lds);
  _Float16* dg_lo = dg_hi + BT * GLDH;
  _Float16* dhp_hi = reinterpret_cast<_Float16*>(dhp_lds);
  _Float16* dhp_lo = dhp_hi + BT * HLDH;
  const float gs = X3 ? p.gscale[0] : 1.f, gsi = X3 ? p.gscale[1] : 1.f;
  const float un_h2h = X3 ? p.w_inv[0] * gsi : 1.f, un_ih = X3 ? p.w_inv[1] * gsi : 1.f, un_hh = X3 ? p.w_inv[2] * gsi : 1.f;
  float* de_lds = hp_lds + BT * HLD;         // [BT][T]
  float* al_lds = de_lds + BT * T;           // [BT][T]
  float* sw_lds = al_lds + BT * T;           // [HID]
  float* red = sw_lds + HID;                 // [NW][HID] reduction of d w_score at the end

  // every step re-reads / updates the workgroup's Hproj, Hb, dHproj, dHb slices (66 KB per sample each) through ONE CU's memory
  // pipeline, so a batch that would occupy only a few CUs runs with fewer samples per workgroup (like the forward kernel's vb)
  const int vb = p.vb;
  const int b0 = blockIdx.x * vb;
  const int Bend = min(p.B, b0 + vb);
  const int t_ = threadIdx.x, lane = t_ & 63, wave = t_ >> 6;
  const int col = lane & 15, rbase = (lane >> 4) * 4;
  const int j = wave * 16 + col;
  // pack_fragment_major([D, 4H]) = [wave][D / 256 blocks][4H / 16 steps][64 lanes] float4s
  const f32x4* w_ctx = reinterpret_cast<const f32x4*>(p.w_ih_ctxT) + (long)wave * (p.D / HID) * (4 * HID / 16) * 64 + lane;
  const f32x4* w_hh = reinterpret_cast<const f32x4*>(p.w_hhT) + (long)wave * (4 * HID / 16) * 64 + lane;
  const f32x4* w_h2h = reinterpret_cast<const f32x4*>(p.w_h2hT) + (long)wave * (HID / 16) * 64 + lane;
  // x3 streams: [wave][block][K / 32 steps][64 lanes][hi 8 | lo 8 halves]
  const f16v8b* w_ctx_h = reinterpret_cast<const f16v8b*>(p.w_ih_ctxT) + ((long)wave * (p.D / HID) * (4 * HID / 32) * 64 + lane) * 2;
  const f16v8b* w_hh_h = reinterpret_cast<const f16v8b*>(p.w_hhT) + ((long)wave * (4 * HID / 32) * 64 + lane) * 2;
  const f16v8b* w_h2h_h = reinterpret_cast<const f16v8b*>(p.w_h2hT) + ((long)wave * (HID / 32) * 64 + lane) * 2;

  for (int i = t_; i < HID; i += NTH) sw_lds[i] = p.w_score[i];
  for (int i = t_; i < (X3 ? (2 * BT * HLDH) / 2 : BT * HLD); i += NTH) dhp_lds[i] = 0.f;      // rows >= vb stay zero
  float dh_rec[4] = {0.f, 0.f, 0.f, 0.f}, dc_next[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dws = {0.f, 0.f, 0.f, 0.f};          // d w_score for channels lane*4.. of this wave's sample
  __syncthreads();

  for (int s = p.S - 1; s >= 0; --s) {
    // stage hp[s] and alpha[s] of the 16 samples
    for (int i = t_; i < BT * HID; i += NTH) {
      const int row = i / HID, c = i - row * HID, b = b0 + row;
      hp_lds[row * HLD + c] = b < Bend ? p.hp[((long)b * p.S + s) * HID + c] : 0.f;
    }
    for (int i = t_; i < BT * T; i += NTH) {
      const int row = i / T, t = i - row * T, b = b0 + row;
      al_lds[i] = b < Bend ? p.alpha[((long)b * p.S + s) * T + t] : 0.f;
    }
    // (a) LSTMCell backward for (sample row, unit j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + r, b = b0 + row;
      float di = 0.f, df = 0.f, dg = 0.f, dob = 0.f;
      if (b < Bend) {
        const long base = (long)b * p.S + s;
        const float* gp = p.gates + base * 4 * HID + j;
        const float ig = gp[0], fg = gp[HID], gg = gp[2 * HID], og = gp[3 * HID];
        const float ct = p.cseq[base * HID + j];
        const float cp = s > 0 ? p.cseq[(base - 1) * HID + j] : 0.f;
        const float dh = p.dhid[base * HID + j] + dh_rec[r];
        const float tc = tanhf(ct);
        dob = dh * tc * og * (1.f - og);
        const float dc = dc_next[r] + dh * og * (1.f - tc * tc);
        di = dc * gg * ig * (1.f - ig);
        df = dc * cp * fg * (1.f - fg);
        dg = dc * ig * (1.f - gg * gg);
        dc_next[r] = dc * fg;
        float* dp = p.dgates + base * 4 * HID + j;
        dp[0] = di; dp[HID] = df; dp[2 * HID] = dg; dp[3 * HID] = dob;
      }
      if constexpr (X3) {
        const float v4[4] = {di * gs, df * gs, dg * gs, dob * gs};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          _Float16 hh, ll;
          split_f16_sat(v4[g], hh, ll);
          dg_hi[row * GLDH + g * HID + j] = hh;
          dg_lo[row * GLDH + g * HID + j] = ll;
        }
      } else {
        float* l = dg_lds + row * GLD + j;
        l[0] = di; l[HID] = df; l[2 * HID] = dg; l[3 * HID] = dob;
      }
    }
    __syncthreads();
    // (b) + (c), one 256-column block of the context at a time (D = 256 * G: DERNet's main head attends over the G
    // extractors' concatenated features, modules/model.py:289-291):
    //   dctx[:, blk] = dgates . W_ih[:, blk]   (block 0 shares its A fragments with dh_prev = dgates . W_hh, K = 4H)
    //   dalpha[b][t] += dctx[b, blk] . Hb[b][t][blk]   (wave per (b,t) pair).  dHb = sum_s alpha[s] (x) dctx[s] is NOT accumulated here (a
    //   read-modify-write of the workgroup's whole dHb slice every step, twice the bytes of everything else the step moves): the steps'
    //   dctx go to p.dctx and attn_bwd_dhb_kernel forms the sum afterwards -- one small product per sample, dHb written once
    const int G = p.D / HID;
    for (int blk = 0; blk < G; ++blk) {
      if (blk == 0) {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (X3) {
          const f16v8b* const wp[2] = {w_ctx_h, w_hh_h};
          mma_shared_a_h<2>(acc, dg_hi, dg_lo, GLDH, wp, 4 * HID / 32, lane);
        } else {
          const f32x4* const wp[2] = {w_ctx, w_hh};
          mma_shared_a<2>(acc, dg_lds, GLD, wp, 4 * HID / 16, lane);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[0][r] * un_ih;
          dctx_lds[(rbase + r) * HLD + j] = v;
          if (b0 + rbase + r < Bend) p.dctx[((long)(b0 + rbase + r) * p.S + s) * p.D + j] = v;
          dh_rec[r] = acc[1][r] * un_hh;
        }
      } else {
        f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (X3) {
          const f16v8b* const wp[1] = {w_ctx_h + (long)blk * (4 * HID / 32) * 64 * 2};
          mma_shared_a_h<1>(acc, dg_hi, dg_lo, GLDH, wp, 4 * HID / 32, lane);
        } else {
          const f32x4* const wp[1] = {w_ctx + (long)blk * (4 * HID / 16) * 64};
          mma_shared_a<1>(acc, dg_lds, GLD, wp, 4 * HID / 16, lane);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[0][r] * un_ih;
          dctx_lds[(rbase + r) * HLD + j] = v;
          if (b0 + rbase + r < Bend) p.dctx[((long)(b0 + rbase + r) * p.S + s) * p.D + blk * HID + j] = v;
        }
      }
      __syncthreads();
      for (int pr0 = wave * 4; pr0 < vb * T; pr0 += NW * 4) {
        f32x4 hv[4];
        int rows[4], ts[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pr = pr0 + u;
          rows[u] = pr < vb * T ? pr / T : 0;
          ts[u] = pr < vb * T ? pr - rows[u] * T : 0;
          ok[u] = pr < vb * T && b0 + rows[u] < Bend;
          hv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (ok[u]) hv[u] = *reinterpret_cast<const f32x4*>(p.Hb + ((long)(b0 + rows[u]) * T + ts[u]) * p.D + blk * HID + lane * 4);
        }
        float sacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x4 dv = *reinterpret_cast<const f32x4*>(dctx_lds + rows[u] * HLD + lane * 4);
          sacc[u] = hv[u][0] * dv[0] + hv[u][1] * dv[1] + hv[u][2] * dv[2] + hv[u][3] * dv[3];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) sacc[u] += __shfl_xor(sacc[u], o);
        }
        if (lane < 4 && pr0 + lane < vb * T) {
          const float v = lane == 0 ? sacc[0] : lane == 1 ? sacc[1] : lane == 2 ? sacc[2] : sacc[3];
          de_lds[pr0 + lane] = blk == 0 ? v : de_lds[pr0 + lane] + v;      // (the same wave owns this pair in every block)
        }
      }
      __syncthreads();
    }
    // (d) softmax backward: de = alpha * (dalpha - sum_t alpha*dalpha)   (wave per sample)
    if (wave < vb) {
      const int row = wave;
      float dot = 0.f;
      for (int t = lane; t < T; t += 64) dot += al_lds[row * T + t] * de_lds[row * T + t];
      dot = wave_sum(dot);
      for (int t = lane; t < T; t += 64) {
        const float v = al_lds[row * T + t] * (de_lds[row * T + t] - dot);
        de_lds[row * T + t] = v;
        if (b0 + row < Bend) p.de[((long)(b0 + row) * p.S + s) * T + t] = v;
      }
    }
    __syncthreads();
    // (e) through e = score . tanh(Hproj + hp): NW / vb waves per sample (each a contiguous slice of the T positions), lane = 4
    //     channels; dhp accumulates in registers and the slices of one sample meet in LDS.  dHproj (the same terms summed over the steps
    //     instead of over t) is formed afterwards by attn_bwd_dhproj_kernel from the steps' de -- no read-modify-write of dHproj here
    {
      const int wps = NW / vb;
      const int row = wave % vb, part = wave / vb, b = b0 + row;
      const int tchunk = (T + wps - 1) / wps;
      const int t_end = min(T, (part + 1) * tchunk);
      f32x4 dhp = {0.f, 0.f, 0.f, 0.f};
      if (b < Bend) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(sw_lds + lane * 4);
        const f32x4 pv = *reinterpret_cast<const f32x4*>(hp_lds + row * HLD + lane * 4);
        int t = part * tchunk;
#pragma unroll 1
        for (; t + 4 <= t_end; t += 4) {
          f32x4 hv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) hv[u] = *reinterpret_cast<const f32x4*>(p.Hproj + ((long)b * T + t + u) * HID + lane * 4);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float de = de_lds[row * T + t + u];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float uu = fast_tanh(hv[u][k] + pv[k]);
              const float dpre = de * wv[k] * (1.f - uu * uu);
              dws[k] = fmaf(de, uu, dws[k]);
              dhp[k] += dpre;
            }
          }
        }
        for (; t < t_end; ++t) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(p.Hproj + ((long)b * T + t) * HID + lane * 4);
          const float de = de_lds[row * T + t];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float uu = fast_tanh(hv[k] + pv[k]);
            const float dpre = de * wv[k] * (1.f - uu * uu);
            dws[k] = fmaf(de, uu, dws[k]);
            dhp[k] += dpre;
          }
        }
      }
      if (wps > 1) {                      // (uniform across the workgroup)
        *reinterpret_cast<f32x4*>(red + wave * HID + lane * 4) = dhp;
        __syncthreads();
        if (part == 0) {
          for (int q = 1; q < wps; ++q) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(red + (row + q * vb) * HID + lane * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) dhp[k] += o[k];
          }
        }
      }
      if (part == 0) {
        if (b < Bend) *reinterpret_cast<f32x4*>(p.dhp + ((long)b * p.S + s) * HID + lane * 4) = dhp;
        if constexpr (X3) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            _Float16 hh, ll;
            split_f16_sat(dhp[k] * gs, hh, ll);
            dhp_hi[row * HLDH + lane * 4 + k] = hh;
            dhp_lo[row * HLDH + lane * 4 + k] = ll;
          }
        } else {
          *reinterpret_cast<f32x4*>(dhp_lds + row * HLD + lane * 4) = dhp;
        }
      }
    }
    __syncthreads();
    // (f) dh_prev += dhp . W_h2h
    if (s > 0) {
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      if constexpr (X3) {
        const f16v8b* const wp[1] = {w_h2h_h};
        mma_shared_a_h<1>(acc, dhp_hi, dhp_lo, HLDH, wp, HID / 32, lane);
      } else {
        const f32x4* const wp[1] = {w_h2h};
        mma_shared_a<1>(acc, dhp_lds, HLD, wp, HID / 16, lane);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh_rec[r] += acc[0][r] * un_h2h;
    }
    __syncthreads();
  }
  // d w_score: sum the 16 waves (samples) of this workgroup
  *reinterpret_cast<f32x4*>(red + wave * HID + lane * 4) = dws;
  __syncthreads();
  for (int c = t_; c < HID; c += NTH) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w * HID + c];
    p.dws_part[(long)blockIdx.x * HID + c] = v;
  }
}

__global__ void embed_scatter_add_kernel(const long* __restrict__ idx, long idx_stride, const float* __restrict__ demb,
                                         float* __restrict__ dtable, long n, int E, int num_class, int S) {
  const long i = blockIdx.x;
  if (i >= n) return;
  const long b = i / S, s = i - b * S;
  long k = idx[b * idx_stride + s];
  if (k >= num_class || k < 0) k = 0;
  for (int e = threadIdx.x; e < E; e += blockDim.x) atomicAdd(dtable + k * E + e, demb[i * E + e]);
}

__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int HW, int C, float inv) {
  const int b = blockIdx.y;
  const long n = (long)HW * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[(long)b * n + i] = dy[(long)b * C + (i % C)] * inv;
}

// ---- TPS sampler backward: d C' -------------------------------------------------------------------------
constexpr int MAXF = 64;

__global__ __launch_bounds__(256) void tps_sample_bwd_kernel(const float* __restrict__ img, const float* __restrict__ cprime,
                                                             const float* __restrict__ inv_delta, const float* __restrict__ p_hat,
                                                             const float* __restrict__ dout, float* __restrict__ dcprime, int H,
                                                             int W, int Hr, int Wr, int F) {
  __shared__ float T[MAXF][2];
  __shared__ float dT[MAXF][2];
  __shared__ float scratch[4];
  const int b = blockIdx.x;
  const int F3 = F + 3;
  for (int i = threadIdx.x; i < F3 * 2; i += 256) {
    const int r = i >> 1, d = i & 1;
    float s = 0.f;
    for (int jj = 0; jj < F; ++jj) s = fmaf(inv_delta[r * F3 + jj], cprime[((long)b * F + jj) * 2 + d], s);
    T[r][d] = s;
  }
  __syncthreads();
  const int n = Hr * Wr;
  const f32x4* im = reinterpret_cast<const f32x4*>(img) + (long)b * H * W;
  float acc[MAXF / 2][2];     // F3 <= 32 supported in registers
#pragma unroll
  for (int q = 0; q < MAXF / 2; ++q) { acc[q][0] = 0.f; acc[q][1] = 0.f; }
  for (int pix = threadIdx.x; pix < n; pix += 256) {
    const float* ph = p_hat + (long)pix * F3;
    float gx = 0.f, gy = 0.f;
    for (int q = 0; q < F3; ++q) {
      gx = fmaf(ph[q], T[q][0], gx);
      gy = fmaf(ph[q], T[q][1], gy);
    }
    float ix = (gx + 1.f) * 0.5f * (float)(W - 1), iy = (gy + 1.f) * 0.5f * (float)(H - 1);
    float mx = 1.f, my = 1.f;                    // border clamp kills the gradient (torch clip_coordinates_set_grad)
    if (ix <= 0.f) { ix = 0.f; mx = 0.f; } else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); mx = 0.f; }
    if (iy <= 0.f) { iy = 0.f; my = 0.f; } else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); my = 0.f; }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = ix - fx, ty = iy - fy;
    const bool xin1 = x1 < W, yin1 = y1 < H;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 nw = im[(long)y0 * W + x0];
    const f32x4 ne = xin1 ? im[(long)y0 * W + x1] : zero;
    const f32x4 sw = yin1 ? im[(long)y1 * W + x0] : zero;
    const f32x4 se = (xin1 && yin1) ? im[(long)y1 * W + x1] : zero;
    const f32x4 g = reinterpret_cast<const f32x4*>(dout)[(long)b * n + pix];
    float dix = 0.f, diy = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dix += g[c] * ((ne[c] - nw[c]) * (1.f - ty) + (se[c] - sw[c]) * ty);
      diy += g[c] * ((sw[c] - nw[c]) * (1.f - tx) + (se[c] - ne[c]) * tx);
    }
    const float dgx = dix * mx * 0.5f * (float)(W - 1), dgy = diy * my * 0.5f * (float)(H - 1);
#pragma unroll
    for (int q = 0; q < MAXF / 2; ++q)
      if (q < F3) {
        acc[q][0] = fmaf(ph[q], dgx, acc[q][0]);
        acc[q][1] = fmaf(ph[q], dgy, acc[q][1]);
      }
  }
#pragma unroll
  for (int q = 0; q < MAXF / 2; ++q) {
    if (q < F3) {   // uniform condition
      const float s0 = block_sum<256>(acc[q][0], scratch);
      const float s1 = block_sum<256>(acc[q][1], scratch);
      if (threadIdx.x == 0) { dT[q][0] = s0; dT[q][1] = s1; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < F * 2; i += 256) {
    const int f = i >> 1, d = i & 1;
    float s = 0.f;
    for (int r = 0; r < F3; ++r) s = fmaf(inv_delta[r * F3 + f], dT[r][d], s);
    dcprime[((long)b * F + f) * 2 + d] = s;
  }
}

// ---- the two deferred sums of the decoder backward (one workgroup of 256 threads per sample and 256-column block) ------------------
// dHb[b][t][d] = sum_s alpha[b][s][t] * dctx[b][s][d]: thread = column d; alpha[b] and the block's dctx[b] staged in LDS
__global__ __launch_bounds__(256) void attn_bwd_dhb_kernel(const float* __restrict__ alpha, const float* __restrict__ dctx,
                                                           float* __restrict__ dHb, int T, int D, int S) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* al = sm;                 // [S][T]
  float* dc = sm + S * T;         // [S][256]
  const int b = blockIdx.x, d0 = blockIdx.y * 256, tid = threadIdx.x;
  for (int i = tid; i < S * T; i += 256) al[i] = alpha[(long)b * S * T + i];
  for (int s = 0; s < S; ++s) dc[s * 256 + tid] = dctx[((long)b * S + s) * D + d0 + tid];
  __syncthreads();
  float* out = dHb + (long)b * T * D + d0 + tid;
  for (int t = 0; t < T; ++t) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a = fmaf(al[s * T + t], dc[s * 256 + tid], a);
    out[(long)t * D] = a;
  }
}

// dHproj[b][t][c] = sum_s de[b][s][t] * w_score[c] * (1 - tanh^2(Hproj[b][t][c] + hp[b][s][c])): thread = channel c, blockIdx.y = a slice
// of the T positions; hp[b] and de[b] staged in LDS.  The terms are the main kernel's dpre (same expression, same fast_tanh).
__global__ __launch_bounds__(256) void attn_bwd_dhproj_kernel(const float* __restrict__ Hproj, const float* __restrict__ hp,
                                                              const float* __restrict__ de, const float* __restrict__ w_score,
                                                              float* __restrict__ dHproj, int T, int S) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* hps = sm;                // [S][HID]
  float* des = sm + S * HID;      // [S][T]
  const int b = blockIdx.x, c = threadIdx.x;
  for (int s = 0; s < S; ++s) hps[s * HID + c] = hp[((long)b * S + s) * HID + c];
  for (int i = c; i < S * T; i += 256) des[i] = de[(long)b * S * T + i];
  __syncthreads();
  const float w = w_score[c];
  const int per = (T + gridDim.y - 1) / gridDim.y;
  const int t_end = min(T, (int)(blockIdx.y + 1) * per);
  for (int t = blockIdx.y * per; t < t_end; ++t) {
    const float x = Hproj[((long)b * T + t) * HID + c];
    float a = 0.f;
    for (int s = 0; s < S; ++s) {
      const float uu = fast_tanh(x + hps[s * HID + c]);
      a += des[s * T + t] * w * (1.f - uu * uu);
    }
    dHproj[((long)b * T + t) * HID + c] = a;
  }
}

}  // namespace

// samples per workgroup of the decoder backward: the smallest of {2, 4, 8, 16} that keeps the grid within one workgroup per CU (B = 256,
// D = 256 / 1536, us per launch: 1150 / 3120 at 1, 1195 / 3360 at 2, 1440 / 4040 at 4, 1900 / 5420 at 8, 2840 / 8280 at 16; not 1 for the
// reason given at the forward's rule, rnn.hip: attn_launch)
static int attn_bwd_vb(int B) {
  static const int forced = getenv("MRN_ATTN_BWD_VB") ? atoi(getenv("MRN_ATTN_BWD_VB")) : 0;     // (A/B switch, read once)
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) return forced;
  for (int v = 2; v < 16; v *= 2)
    if (ceil_div(B, v) <= 256) return v;
  return 16;
}

static int attn_bwd_launch(const AttnBwdParams& p, hipStream_t st) {
  const bool x3 = p.w_inv != nullptr;
  const size_t lds = sizeof(float) * (BT * GLD + 3 * BT * HLD + 2 * BT * p.T + HID + NW * HID) + (x3 ? 1024 : 0);
  MRN_CHECK_ARG(lds <= 160 * 1024, "mrn_attn_decoder_bwd: LDS budget exceeded (T=%d)", p.T);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_decoder_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)attn_decoder_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const size_t lds_b = sizeof(float) * (size_t)p.S * (p.T + 256), lds_p = sizeof(float) * (size_t)p.S * (HID + p.T);
  MRN_CHECK_ARG(lds_b <= 160 * 1024 && lds_p <= 160 * 1024, "mrn_attn_decoder_bwd: S=%d, T=%d beyond the deferred sums' LDS staging", p.S, p.T);
  if (lds_b > 64 * 1024 || lds_p > 64 * 1024) {
    static bool big = false;
    if (!big) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_dhb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dhproj_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      big = true;
    }
  }
  if (x3) hipLaunchKernelGGL(attn_decoder_bwd_kernel<true>, dim3(ceil_div(p.B, p.vb)), dim3(NTH), lds, st, p);
  else hipLaunchKernelGGL(attn_decoder_bwd_kernel<false>, dim3(ceil_div(p.B, p.vb)), dim3(NTH), lds, st, p);
  MRN_LAUNCH_CHECK("attn_decoder_bwd");
  hipLaunchKernelGGL(attn_bwd_dhb_kernel, dim3(p.B, p.D / 256), dim3(256), lds_b, st, p.alpha, (const float*)p.dctx, p.dHb, p.T, p.D, p.S);
  MRN_LAUNCH_CHECK("attn_bwd_dhb");
  hipLaunchKernelGGL(attn_bwd_dhproj_kernel, dim3(p.B, 4), dim3(256), lds_p, st, p.Hproj, p.hp, (const float*)p.de, p.w_score, p.dHproj, p.T, p.S);
  MRN_LAUNCH_CHECK("attn_bwd_dhproj");
  return MRN_OK;
}

// rows of the d w_score partial-sum buffer (= workgroups of mrn_attn_decoder_bwd_f32) for a batch of B
MRN_EXPORT int64_t mrn_attn_decoder_bwd_parts(int B) { return B > 0 ? ceil_div(B, attn_bwd_vb(B)) : 0; }

MRN_EXPORT int mrn_attn_decoder_bwd_f32(const float* Hb, const float* Hproj, const float* alpha, const float* gates,
                                        const float* cseq, const float* ctx, const float* hp, const float* dhid,
                                        const float* w_score, const float* w_h2hT, const float* w_ih_ctxT, const float* w_hhT,
                                        float* dgates, float* dhp, float* dHb, float* dHproj, float* dwscore_part, float* dctx,
                                        float* de, int B, int T, int D, int S, int hidden, void* stream) {
  MRN_CHECK_ARG(Hb && Hproj && alpha && gates && cseq && ctx && hp && dhid && w_score && w_h2hT && w_ih_ctxT && w_hhT && dgates &&
                    dhp && dHb && dHproj && dwscore_part && dctx && de, "mrn_attn_decoder_bwd_f32: null operand");
  MRN_CHECK_ARG(hidden == HID && D >= HID && D % HID == 0,
                "mrn_attn_decoder_bwd_f32: needs hidden == %d and D a multiple of it (got D=%d hidden=%d)", HID, D, hidden);
  if (B == 0 || S == 0) return MRN_OK;
  AttnBwdParams p;
  p.Hb = Hb; p.Hproj = Hproj; p.alpha = alpha; p.gates = gates; p.cseq = cseq; p.ctx = ctx; p.hp = hp; p.dhid = dhid;
  p.w_score = w_score; p.w_h2hT = w_h2hT; p.w_ih_ctxT = w_ih_ctxT; p.w_hhT = w_hhT;
  p.dgates = dgates; p.dhp = dhp; p.dHb = dHb; p.dHproj = dHproj; p.dws_part = dwscore_part; p.dctx = dctx; p.de = de;
  p.B = B; p.T = T; p.D = D; p.S = S;
  p.vb = attn_bwd_vb(B);
  p.w_inv = nullptr; p.gscale = nullptr;
  return attn_bwd_launch(p, (hipStream_t)stream);
}

// mrn_attn_decoder_bwd_f32 with the three transposed recurrent products as split-fp16 x3: w_h2hT / w_ih_ctxT / w_hhT are the fragment-major
// fp16 hi / lo streams of W_h2h^T [H][H], W_ih[:, :D]^T [D][4H], W_hh^T [H][4H] (ops.pack_fragment_major_h), w_inv a device float[3] of
// their inverse prescales, gscale a device float[2] = {s, 1/s} with a power of two s bringing max|dhid| to ~16 (the gate and hp gradients
// are split as s * value).
MRN_EXPORT int mrn_attn_decoder_bwd_x3(const float* Hb, const float* Hproj, const float* alpha, const float* gates, const float* cseq,
                                       const float* ctx, const float* hp, const float* dhid, const float* w_score, const void* w_h2hT,
                                       const void* w_ih_ctxT, const void* w_hhT, const float* w_inv, const float* gscale, float* dgates,
                                       float* dhp, float* dHb, float* dHproj, float* dwscore_part, float* dctx, float* de, int B, int T,
                                       int D, int S, int hidden, void* stream) {
  MRN_CHECK_ARG(Hb && Hproj && alpha && gates && cseq && ctx && hp && dhid && w_score && w_h2hT && w_ih_ctxT && w_hhT && w_inv && gscale &&
                    dgates && dhp && dHb && dHproj && dwscore_part && dctx && de, "mrn_attn_decoder_bwd_x3: null operand");
  MRN_CHECK_ARG(hidden == HID && D >= HID && D % HID == 0,
                "mrn_attn_decoder_bwd_x3: needs hidden == %d and D a multiple of it (got D=%d hidden=%d)", HID, D, hidden);
  if (B == 0 || S == 0) return MRN_OK;
  AttnBwdParams p;
  p.Hb = Hb; p.Hproj = Hproj; p.alpha = alpha; p.gates = gates; p.cseq = cseq; p.ctx = ctx; p.hp = hp; p.dhid = dhid;
  p.w_score = w_score; p.w_h2hT = (const float*)w_h2hT; p.w_ih_ctxT = (const float*)w_ih_ctxT; p.w_hhT = (const float*)w_hhT;
  p.dgates = dgates; p.dhp = dhp; p.dHb = dHb; p.dHproj = dHproj; p.dws_part = dwscore_part; p.dctx = dctx; p.de = de;
  p.B = B; p.T = T; p.D = D; p.S = S;
  p.vb = attn_bwd_vb(B);
  p.w_inv = w_inv; p.gscale = gscale;
  return attn_bwd_launch(p, (hipStream_t)stream);
}

MRN_EXPORT int mrn_embed_scatter_add_f32(const int64_t* idx, int64_t idx_stride, const float* demb, float* dtable, int B, int S,
                                         int E, int num_class, void* stream) {
  MRN_CHECK_ARG(idx && demb && dtable, "mrn_embed_scatter_add_f32: null operand");
  const long n = (long)B * S;
  if (n == 0) return MRN_OK;
  hipLaunchKernelGGL(embed_scatter_add_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, (const long*)idx,
                     (long)idx_stride, demb, dtable, n, E, num_class, S);
  MRN_LAUNCH_CHECK("embed_scatter_add");
  return MRN_OK;
}

MRN_EXPORT int mrn_avgpool_bwd_nhwc_f32(const float* dy, float* dx, int B, int HW, int C, void* stream) {
  MRN_CHECK_ARG(dy && dx && HW > 0, "mrn_avgpool_bwd_nhwc_f32: bad operands");
  if (B == 0) return MRN_OK;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ceil_div((long)HW * C, 256), B), dim3(256), 0, (hipStream_t)stream, dy, dx, HW, C,
                     1.f / (float)HW);
  MRN_LAUNCH_CHECK("avgpool_bwd");
  return MRN_OK;
}

MRN_EXPORT int mrn_tps_grid_sample_bwd_f32(const float* img_nhwc, const float* cprime, const float* inv_delta_c,
                                           const float* p_hat, const float* dout_nhwc, float* dcprime, int B, int H, int W, int C,
                                           int Hr, int Wr, int F, void* stream) {
  MRN_CHECK_ARG(img_nhwc && cprime && inv_delta_c && p_hat && dout_nhwc && dcprime, "mrn_tps_grid_sample_bwd_f32: null operand");
  MRN_CHECK_ARG(C == 4 && F > 0 && F + 3 <= MAXF / 2, "mrn_tps_grid_sample_bwd_f32: C=%d F=%d unsupported", C, F);
  if (B == 0) return MRN_OK;
  hipLaunchKernelGGL(tps_sample_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, img_nhwc, cprime, inv_delta_c, p_hat,
                     dout_nhwc, dcprime, H, W, Hr, Wr, F);
  MRN_LAUNCH_CHECK("tps_grid_sample_bwd");
  return MRN_OK;
}
