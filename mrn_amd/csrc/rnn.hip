// Recurrent kernels: bidirectional LSTM layer and the attention decoder (teacher-forced and greedy).
//
// Reference: modules/sequence_modeling.py:7-21 (nn.LSTM bidirectional, gate order i,f,g,o) and
// modules/prediction.py:38-118 (Attention / AttentionCell, 26 decode steps).
//
// Samples are independent along the time recurrence, so one workgroup owns a 16-sample batch tile for the
// whole sequence: hidden state lives in LDS, cell state in registers, and only the recurrent weights stream
// from L2 each step.  Products run on v_mfma_f32_16x16x4_f32 (exact fp32): M = 16 samples, N = gate columns,
// K = hidden.  Wave w owns hidden units [64w, 64w+64) for all four gates, so the LSTM pointwise update is
// lane-local (the i,f,g,o pre-activations of a unit land in the same lane and register index).
// The k index is permuted (lane group g takes k = 16q+4g+r) so every operand read is one 16-byte access.
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int HID = 256;       // hidden size (reference configs: hidden_size=256)
constexpr int BT = 16;         // samples per workgroup
constexpr int HLD = HID + 4;   // padded LDS row
constexpr int NW = 16;         // waves per workgroup: wave w owns hidden units [16w, 16w+16) of every gate
constexpr int NTH = NW * 64;   // 1024 threads -- many waves in flight hide the L2 latency of the weight stream

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc[g] += A(16 x K from LDS rows of stride lda) . W_g^T   for K a multiple of 16.
// W is FRAGMENT-MAJOR: Wp[((wave*NG + g)*Q + q)*64 + lane] is the float4 the lane feeds to the four MFMAs of k-step q,
//   = W[g*HID + 16*wave + (lane&15)][16q + 4*(lane>>4) .. +3]
// so every wave-level load is one contiguous 1 KiB line instead of 16 rows x 64 B (pack_fragment_major on the host).
// Weight fragments of step q+1 are fetched before the MFMAs of step q issue; consecutive MFMAs target different
// accumulators (the 16x16x4 f32 MFMA has a 40-cycle dependent latency vs 32-cycle issue).
template <int NG>
__device__ __forceinline__ void mma_rows(f32x4 (&acc)[NG], const float* __restrict__ a_lds, int lda,
                                         const float* __restrict__ Wp, int K, int wave, int lane) {
  const int n = lane & 15, g4 = (lane >> 4) * 4;
  const float* ap = a_lds + n * lda + g4;
  const int Q = K / 16;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (long)wave * NG * Q * 64 + lane;
  f32x4 wv[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) wv[g] = wp[(long)g * Q * 64];
#pragma unroll 1   // keep the one-step-ahead register pipeline; full unrolling hoists every load and spills
  for (int q = 0; q < Q; ++q) {
    f32x4 wn[NG];
    const int qn = (q + 1 < Q) ? q + 1 : q;
#pragma unroll
    for (int g = 0; g < NG; ++g) wn[g] = wp[((long)g * Q + qn) * 64];
    const f32x4 av = *reinterpret_cast<const f32x4*>(ap + q * 16);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = mfma4(av[r], wv[g][r], acc[g]);
#pragma unroll
    for (int g = 0; g < NG; ++g) wv[g] = wn[g];
  }
}

// act[g][r] receives the post-activation gates (i, f, g, o) for the backward pass
__device__ __forceinline__ void lstm_pointwise(const f32x4 (&acc)[4], const float (&xg)[4][4], const float (&bh)[4],
                                               float (&c)[4], float (&h)[4], float (&act)[4][4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gi = acc[0][r] + xg[0][r] + bh[0];
    const float gf = acc[1][r] + xg[1][r] + bh[1];
    const float gg = acc[2][r] + xg[2][r] + bh[2];
    const float go = acc[3][r] + xg[3][r] + bh[3];
    const float ig = sigmoidf_acc(gi), fg = sigmoidf_acc(gf), og = sigmoidf_acc(go), cg = tanhf(gg);
    const float cn = fg * c[r] + ig * cg;
    c[r] = cn;
    h[r] = og * tanhf(cn);
    act[0][r] = ig; act[1][r] = fg; act[2][r] = cg; act[3][r] = og;
  }
}

// (the form whose accumulators already hold the input projection)
__device__ __forceinline__ void lstm_pointwise0(const f32x4 (&acc)[4], const float (&bh)[4], float (&c)[4], float (&h)[4], float (&act)[4][4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gi = acc[0][r] + bh[0];
    const float gf = acc[1][r] + bh[1];
    const float gg = acc[2][r] + bh[2];
    const float go = acc[3][r] + bh[3];
    const float ig = sigmoidf_acc(gi), fg = sigmoidf_acc(gf), og = sigmoidf_acc(go), cg = tanhf(gg);
    const float cn = fg * c[r] + ig * cg;
    c[r] = cn;
    h[r] = og * tanhf(cn);
    act[0][r] = ig; act[1][r] = fg; act[2][r] = cg; act[3][r] = og;
  }
}

// ---------------------------------------------------------------------------------------------
// Bidirectional LSTM layer.  xproj[b][t][dir*4H + gate*H + j] holds W_ih x + b_ih.
// grid = (ceil(B/16), ndir); out[b][t][dir*H + j].
// ---------------------------------------------------------------------------------------------
// Up to MAX_GROUPS independent layers (the frozen experts of the router phase) share one launch: grid.x = groups * tiles.
constexpr int MAX_GROUPS = 8;
struct LstmParams {
  const float* xproj; const float* w_hh; const float* b_hh; float* out;
  float* gates_out;   // optional [B][T][ndir][4H]
  float* c_out;       // optional [B][T][ndir][H]
};
struct LstmGroup {
  LstmParams g[MAX_GROUPS];
  int tiles, B, T, ndir, nsets, pinned;
};

__global__ __launch_bounds__(NTH) void lstm_layer_kernel(const LstmGroup grp) {
  __shared__ __attribute__((aligned(16))) float h_lds[2][BT * HLD];
  // set = (group, direction) owns one W_hh stream (1 MiB); all tiles of a set run on ONE XCD (block b lands on XCD b % 8)
  // so the stream stays resident in that XCD's 4 MiB L2 instead of every XCD cycling through every set's weights
  // (only while a set's tiles fit the 32 CUs of an XCD; larger batches keep the plain block order)
  const int set = grp.pinned ? (int)(blockIdx.x % 8) + 8 * (int)((blockIdx.x / 8) / grp.tiles) : (int)blockIdx.x / grp.tiles;
  if (set >= grp.nsets) return;
  const int gi = set / grp.ndir;
  const float* __restrict__ xproj = grp.g[gi].xproj;
  const float* __restrict__ w_hh = grp.g[gi].w_hh;
  const float* __restrict__ b_hh = grp.g[gi].b_hh;
  float* __restrict__ out = grp.g[gi].out;
  float* __restrict__ gates_out = grp.g[gi].gates_out;
  float* __restrict__ c_out = grp.g[gi].c_out;
  const int B = grp.B, T = grp.T, ndir = grp.ndir;
  const int dir = set - gi * grp.ndir;
  const int b0 = (grp.pinned ? (int)((blockIdx.x / 8) % grp.tiles) : (int)blockIdx.x % grp.tiles) * BT;
  const int t_ = threadIdx.x, lane = t_ & 63, wave = t_ >> 6;
  const float* W = w_hh + (long)dir * 4 * HID * HID;
  const int col = lane & 15, rbase = (lane >> 4) * 4;
  const int j = wave * 16 + col;               // this lane's hidden unit

  for (int i = t_; i < BT * HLD; i += NTH) h_lds[0][i] = 0.f;
  float c[4] = {0.f, 0.f, 0.f, 0.f}, bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = b_hh ? b_hh[dir * 4 * HID + g * HID + j] : 0.f;
  __syncthreads();

  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int cur = step & 1;
    // input projections of this step: issued first, they land while the recurrent product runs
    float xg[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + rbase + r;
      const float* xp = xproj + ((long)(b < B ? b : 0) * T + t) * (ndir * 4 * HID) + dir * 4 * HID + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) xg[g][r] = xp[g * HID];
    }
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_rows<4>(acc, h_lds[cur], HLD, W, HID, wave, lane);
    float h[4], act[4][4];
    lstm_pointwise(acc, xg, bh, c, h, act);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + r, b = b0 + row;
      if (b < B) {
        out[((long)b * T + t) * (ndir * HID) + dir * HID + j] = h[r];
        if (gates_out) {
          const long base = ((long)b * T + t) * ndir + dir;
#pragma unroll
          for (int g = 0; g < 4; ++g) gates_out[base * 4 * HID + g * HID + j] = act[g][r];
          c_out[base * HID + j] = c[r];
        }
      }
      h_lds[cur ^ 1][row * HLD + j] = b < B ? h[r] : 0.f;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Inference variant of the LSTM layer for the FROZEN experts of the router phase: the recurrent product runs as
// split-fp16 x3 on v_mfma_f32_16x16x32_f16 (h = hi + lo written to LDS as two fp16 planes by the pointwise stage, W_hh
// pre-split with a power-of-two prescale into a fragment-major fp16 stream of the same byte size), 96 MFMAs of 16 cycles
// per wave and step instead of 256 MFMAs of 32 cycles on the exact-fp32 matrix pipe.  Same launch geometry, grouping and
// XCD pinning as lstm_layer_kernel; no training saves.
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = HID + 8;   // fp16 LDS row (halves): 528 B, 16 rows hit 16 distinct 16-byte bank groups

// acc[g] += A(16 x K, fp16 hi / lo planes in LDS) . W_g^T with W FRAGMENT-MAJOR fp16:
//   Wp[(((wave*NG + g)*Q + q)*64 + lane)*2 + {0: hi, 1: lo}] = 8 halves W[g*HID + 16*wave + (lane&15)][32q + 8*(lane>>4) .. +7]
template <int NG>
__device__ __forceinline__ void mma_rows_h(f32x4 (&acc)[NG], const _Float16* __restrict__ a_hi, const _Float16* __restrict__ a_lo,
                                           const unsigned char* __restrict__ Wp, int K, int wave, int lane, int ld = LDH) {
  const int n = lane & 15, kg = lane >> 4;
  const int Q = K / 32;
  const f16v8* wp = reinterpret_cast<const f16v8*>(Wp) + ((long)wave * NG * Q * 64 + lane) * 2;
  const _Float16* ah = a_hi + n * ld + kg * 8;
  const _Float16* al = a_lo + n * ld + kg * 8;
  f16v8 wh[NG], wl[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    wh[g] = wp[(long)g * Q * 128];
    wl[g] = wp[(long)g * Q * 128 + 1];
  }
#pragma unroll 1
  for (int q = 0; q < Q; ++q) {
    f16v8 nh[NG], nl[NG];
    const int qn = (q + 1 < Q) ? q + 1 : q;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      nh[g] = wp[((long)g * Q + qn) * 128];
      nl[g] = wp[((long)g * Q + qn) * 128 + 1];
    }
    const f16v8 xh = *reinterpret_cast<const f16v8*>(ah + q * 32), xl = *reinterpret_cast<const f16v8*>(al + q * 32);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh[g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl[g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh[g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      wh[g] = nh[g];
      wl[g] = nl[g];
    }
  }
}

// mma_rows_h with the weight stream behind a buffer resource: scalar (wave, gate, k-step) offsets + one 32-bit lane offset, no 64-bit
// fragment pointers in the vector file (the pointer form put the decoder kernel 60 registers over its 128)
template <int NG>
__device__ __forceinline__ void mma_rows_hb(f32x4 (&acc)[NG], const _Float16* __restrict__ a_hi, const _Float16* __restrict__ a_lo,
                                            const __amdgpu_buffer_rsrc_t rw, int K, int wave, int lane, int ld = LDH) {
  const int n = lane & 15, kg = lane >> 4;
  const int Q = K / 32;
  const int wwave = __builtin_amdgcn_readfirstlane(wave) * NG * Q * 2048;
  const int wlane = lane * 32;
  auto wfrag = [&](int g, int q, int plane) -> f16v8 {
    return __builtin_bit_cast(f16v8, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, wwave + ((g * Q + q) * 128 + plane) * 16, 0));
  };
  const _Float16* ah = a_hi + n * ld + kg * 8;
  const _Float16* al = a_lo + n * ld + kg * 8;
  f16v8 wh[NG], wl[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    wh[g] = wfrag(g, 0, 0);
    wl[g] = wfrag(g, 0, 1);
  }
#pragma unroll 1
  for (int q = 0; q < Q; ++q) {
    f16v8 nh[NG], nl[NG];
    const int qn = (q + 1 < Q) ? q + 1 : q;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      nh[g] = wfrag(g, qn, 0);
      nl[g] = wfrag(g, qn, 1);
    }
    const f16v8 xh = *reinterpret_cast<const f16v8*>(ah + q * 32), xl = *reinterpret_cast<const f16v8*>(al + q * 32);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh[g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl[g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh[g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      wh[g] = nh[g];
      wl[g] = nl[g];
    }
  }
}

__device__ __forceinline__ void store_h_split(_Float16* hi, _Float16* lo, int idx, float v) {
  _Float16 h, l;
  split_f16(v, h, l);
  hi[idx] = h;
  lo[idx] = l;
}

struct LstmX3Params {
  const float* xproj; const unsigned char* w_hh; const float* w_inv; const float* b_hh; float* out;
  float* gates_out;   // optional training saves, as in LstmParams: [B][T][ndir][4H] post-activation gates
  float* c_out;       //                                          [B][T][ndir][H] cell state
};
struct LstmX3Group {
  LstmX3Params g[MAX_GROUPS];
  int tiles, B, T, ndir, nsets, pinned;
};

// RB: 16-sample row blocks per workgroup.  A workgroup streams the whole W_hh of its (expert, direction) set through one CU every step
// (1 MiB; 64 B / clk / CU from L2 = 7.8 us), with little MFMA work behind each fragment (96 MFMAs of 16 cycles per wave).  RB = 2 -- the same
// stream for twice the samples, half as many workgroups per L2 -- was built and measured SLOWER (round 6, tools/bench_lstm.py, B = 256,
// T = 65: G = 1 17.1 us / step against 11.8, G = 3 21.3 / 14.8, G = 6 22.9 / 15.3): with registers for the next hi fragments only, the
// second row block's MFMAs wait on the lo fragments instead of hiding them.  RB = 1 is the product form; MRN_LSTM_RB=2 selects the other
// for an A/B (bit-identical results).
// The input projection is the accumulators' INITIAL value, scaled by the weights' power-of-two prescale (exact): no registers of its own.
// Every global access is a buffer instruction (resource + scalar offset + a 32-bit lane offset).  Both changes are round 6's: the form
// with 64-bit row and fragment pointers in vector registers ran at the 128-register cap; this one takes 100 and is 12-24 % faster
// (G = 1 13.3 -> 11.8 us / step, G = 3 17.9 -> 14.8, G = 6 20.3 -> 15.3).
template <int RB>
__global__ __launch_bounds__(NTH) void lstm_layer_x3_kernel(const LstmX3Group grp) {
  constexpr int BTX = BT * RB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lstm_lds[];
  _Float16* const h_hi = reinterpret_cast<_Float16*>(lstm_lds);        // [2][BTX * LDH]
  _Float16* const h_lo = h_hi + 2 * BTX * LDH;                         // [2][BTX * LDH]
  // pinned == 1: all tiles of a (expert, direction) set run on XCD (set % 8), whose L2 then holds that set's 1 MiB of W_hh.
  // pinned == 2: the (set, tile) pairs in set-major order are cut into eight equal runs, one per XCD (workgroup b runs on XCD b % 8): every
  // L2 serves the same number of workgroups and at most two or three sets' weights.  The step time follows the number of workgroups
  // streaming through one L2 (13.1 us at 4, 17.4 at 16, 21.9 at 32): 12 sets = 24 per XCD instead of 32 / 16, 6 sets = 12 instead of 16 / 0.
  int set, tile;
  if (grp.pinned == 2) {
    const int per = grp.nsets * grp.tiles / 8;
    const int pair = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    set = pair / grp.tiles;
    tile = pair - set * grp.tiles;
  } else {
    set = grp.pinned ? (int)(blockIdx.x % 8) + 8 * (int)((blockIdx.x / 8) / grp.tiles) : (int)blockIdx.x / grp.tiles;
    tile = grp.pinned ? (int)((blockIdx.x / 8) % grp.tiles) : (int)blockIdx.x % grp.tiles;
  }
  if (set >= grp.nsets) return;
  const int gi = set / grp.ndir;
  const float* __restrict__ xproj = grp.g[gi].xproj;
  const float* __restrict__ b_hh = grp.g[gi].b_hh;
  float* __restrict__ out = grp.g[gi].out;
  float* __restrict__ gates_out = grp.g[gi].gates_out;
  float* __restrict__ c_out = grp.g[gi].c_out;
  const int B = grp.B, T = grp.T, ndir = grp.ndir;
  const int dir = set - gi * grp.ndir;
  const int b0 = tile * BTX;
  const int t_ = threadIdx.x, lane = t_ & 63, wave = t_ >> 6;
  const unsigned char* W = grp.g[gi].w_hh + (long)dir * 4 * HID * HID * 4;     // hi + lo fp16 = 4 bytes per weight
  const float inv = grp.g[gi].w_inv[dir];
  const float pre = 1.f / inv;                                                 // the prescale itself (a power of two: exact)
  const int col = lane & 15, rbase = (lane >> 4) * 4;
  const int j = wave * 16 + col;

  for (int i = t_; i < BTX * LDH; i += NTH) {
    h_hi[i] = (_Float16)0.f;
    h_lo[i] = (_Float16)0.f;
  }
  float c[RB][4], bh[4];
  // per-lane BYTE offsets of this lane's rows in xproj / the gate saves ([B][T][ndir][4H]); the (t, direction) part of an address is
  // wave-uniform and goes into the scalar base, and the [B][T][ndir][H] arrays' offsets follow from the same registers -- eight
  // loop-invariant registers instead of a 64-bit pointer per row and array (which the compiler hoisted and, at RB = 2, spilled).
  // The launcher keeps B * T * ndir * 4H * 4 bytes below 2^32 (larger batches go in several launches)
  unsigned xoff[RB][4];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + rb * BT + rbase + r;
      c[rb][r] = 0.f;
      xoff[rb][r] = ((unsigned)(b < B ? b : 0) * (unsigned)(T * ndir * 4 * HID) + (unsigned)j) * 4u;
    }
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = b_hh ? b_hh[dir * 4 * HID + g * HID + j] : 0.f;
  __syncthreads();

  // fragment-major weight stream of this wave: [g][q][lane][hi 8 | lo 8 halves] (mma_rows_h's layout)
  constexpr int Q = HID / 32;
  // every global access is a buffer instruction -- resource + scalar offset + one 32-bit lane offset: no address lives in the vector file
  // (as 64-bit pointers per gate / row the compiler hoisted them out of the step loop and, at RB = 2, spilled them)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 4 * HID * HID * 4, 0x00020000);
  const int wwave = __builtin_amdgcn_readfirstlane(wave) * (4 * Q * 64 * 32);
  const int wlane = lane * 32;
  auto wfrag = [&](int g, int q, int plane) -> f16v8 {
    return __builtin_bit_cast(f16v8, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, wwave + ((g * Q + q) * 128 + plane) * 16, 0));
  };
  const unsigned xbytes = (unsigned)B * (unsigned)(T * ndir * 4 * HID) * 4u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xproj, 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, xbytes / 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gates_out, 0, gates_out ? xbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)c_out, 0, gates_out ? xbytes / 4 : 0, 0x00020000);
  const int n = lane & 15, kg = lane >> 4;

  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int cur = step & 1;
    f32x4 acc[RB][4];
    const int tq = (t * ndir + dir) * HID * 4;                                     // (wave-uniform) byte offset in a [..][T][ndir][H] row
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[rb][g][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)xoff[rb][r], 4 * tq + g * HID * 4, 0)) * pre;
      }
    {
      const _Float16* ah = h_hi + cur * BTX * LDH + n * LDH + kg * 8;
      const _Float16* al = h_lo + cur * BTX * LDH + n * LDH + kg * 8;
      // RB = 1: the hi and lo fragments of k-step q + 1 are fetched before the MFMAs of step q issue.  RB = 2 has registers for the next hi
      // fragments only: the lo fragments of step q are fetched at its top and used by its LAST product group
      constexpr bool PF_LO = RB == 1;
      f16v8 wh[4], wl[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        wh[g] = wfrag(g, 0, 0);
        if (PF_LO) wl[g] = wfrag(g, 0, 1);
      }
#pragma unroll 1
      for (int q = 0; q < Q; ++q) {
        f16v8 nh[4], nl[4];
        const int qn = (q + 1 < Q) ? q + 1 : q;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (!PF_LO) wl[g] = wfrag(g, q, 1);
          nh[g] = wfrag(g, qn, 0);
          if (PF_LO) nl[g] = wfrag(g, qn, 1);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const f16v8 xh = *reinterpret_cast<const f16v8*>(ah + rb * BT * LDH + q * 32), xl = *reinterpret_cast<const f16v8*>(al + rb * BT * LDH + q * 32);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[rb][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh[g], acc[rb][g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[rb][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh[g], acc[rb][g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[rb][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl[g], acc[rb][g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          wh[g] = nh[g];
          if (PF_LO) wl[g] = nl[g];
        }
      }
    }
    _Float16* const nhi = h_hi + (cur ^ 1) * BTX * LDH;
    _Float16* const nlo = h_lo + (cur ^ 1) * BTX * LDH;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[rb][g] *= inv;        // undo the power-of-two weight prescale (exact): x-projection + W_hh . h
      float h[4], act[4][4];
      lstm_pointwise0(acc[rb], bh, c[rb], h, act);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb * BT + rbase + r, b = b0 + row;
        if (b < B) {
          const int ooff = (int)(((xoff[rb][r] - 4u * (unsigned)j) >> 2) + 4u * (unsigned)j);       // the same row of a [B][T][ndir][H] array
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h[r]), ro, ooff, tq, 0);
          if (gates_out) {
#pragma unroll
            for (int g = 0; g < 4; ++g) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, act[g][r]), rg, (int)xoff[rb][r], 4 * tq + g * HID * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c[rb][r]), rc, ooff, tq, 0);
          }
        }
        store_h_split(nhi, nlo, row * LDH + j, b < B ? h[r] : 0.f);
      }
    }
    __syncthreads();
  }
}

// one launch of a filled group
template <int RB>
static int lstm_x3_launch_rb(LstmX3Group& grp, int n, bool may_pin, hipStream_t stream) {
  constexpr size_t lds = (size_t)4 * BT * RB * LDH * sizeof(_Float16);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)lstm_layer_x3_kernel<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  grp.tiles = ceil_div(grp.B, BT * RB);
  grp.nsets = n * grp.ndir;
  grp.pinned = 0;
  int blocks = grp.nsets * grp.tiles;
  if (may_pin) {
    grp.pinned = grp.nsets > 2 && grp.tiles * ceil_div(grp.nsets, 8) <= 32;
    if (grp.pinned) blocks = 8 * ceil_div(grp.nsets, 8) * grp.tiles;
    static const bool balance = !(getenv("MRN_LSTM_BALANCE") && atoi(getenv("MRN_LSTM_BALANCE")) == 0);     // (A/B switch, read once)
    if (balance && grp.nsets > 2 && (grp.nsets * grp.tiles) % 8 == 0 && grp.nsets * grp.tiles <= 256) {
      grp.pinned = 2;
      blocks = grp.nsets * grp.tiles;
    }
  }
  hipLaunchKernelGGL(lstm_layer_x3_kernel<RB>, dim3(blocks), dim3(NTH), lds, stream, grp);
  return 0;
}
static int lstm_x3_launch(LstmX3Group& all, int n, bool may_pin, hipStream_t stream) {
  static const int forced = getenv("MRN_LSTM_RB") ? atoi(getenv("MRN_LSTM_RB")) : 0;
  // the kernel's row offsets are 32-bit byte offsets into [B][T][ndir][4H] floats: batches beyond that go in chunks of whole tiles
  const long row_bytes = (long)all.T * all.ndir * 4 * HID * 4;
  const long fit = 0xffffffffL / row_bytes / (2 * BT) * (2 * BT);
  MRN_CHECK_ARG(fit >= 2 * BT, "lstm x3 layer: T=%d is beyond the kernel's 32-bit row offsets", all.T);
  for (long s0 = 0; s0 < all.B; s0 += fit) {
    LstmX3Group grp = all;
    grp.B = (int)(all.B - s0 < fit ? all.B - s0 : fit);
    for (int i = 0; i < n; ++i) {
      LstmX3Params& q = grp.g[i];
      q.xproj += s0 * (row_bytes / 4);
      q.out += s0 * (row_bytes / 16);
      if (q.gates_out) q.gates_out += s0 * (row_bytes / 4);
      if (q.c_out) q.c_out += s0 * (row_bytes / 16);
    }
    const int rc = forced == 2 ? lstm_x3_launch_rb<2>(grp, n, may_pin, stream) : lstm_x3_launch_rb<1>(grp, n, may_pin, stream);
    if (rc) return rc;
  }
  return 0;
}

// (A weight-stationary variant -- W_hh slices in the LDS of 16 workgroups per (expert, direction), h exchanged through L2 every step --
// was built, bit-identical and measured slower in round 2: G = 3 22.3 us / step against 17.1 streaming, G = 6 23.9 against 22.4; the
// hand-over of h at agent scope costs more than streaming W_hh.  Removed in round 4; DESIGN.md section 7 keeps the measurements.)

// ---------------------------------------------------------------------------------------------
// Attention decoder, all S steps in one launch (reference recomputes i2h(H) every step and issues ~10
// small launches per step; here i2h(H) and the embedding half of the LSTMCell input projection are hoisted
// into GEMMs by the caller).
//   Hb     [B][T][D]      encoder states (D multiple of 16)
//   Hproj  [B][T][HID]    i2h(Hb)
//   eproj  [B][S][4*HID]  W_ih[:, D:] . emb(text) + b_ih       (teacher forced)
//   hid    [B][S][HID]    decoder hidden states (generator GEMM is applied afterwards)
// Greedy mode (tokens fed back through argmax of the generator) uses the same kernel one step at a time:
//   steps = 1, state carried in h_state / c_state.
// ---------------------------------------------------------------------------------------------
struct AttnDecParams {
  const float* Hb; const float* Hproj; const float* eproj;
  const float* w_h2h; const float* b_h2h; const float* w_score;
  const float* w_ih;                // context part of LSTMCell weight_ih ([4H][D]), fragment-major
  const float* w_hh;                // [4H][HID], fragment-major (w_h2h too)
  const float* b_hh;                // optional [4H] (eproj already carries b_ih)
  float* hid;
  const float* w_inv;               // x3 form only: device float[3] = 1 / prescale of w_h2h, w_ih, w_hh (then fragment-major fp16 hi / lo streams)
  float* h_state; float* c_state;   // optional [B][HID] carried state (nullptr: start from zero, do not store)
  float* alpha_out;                 // optional [B][S][T]
  float* gates_out; float* c_out;   // optional training saves: [B][S][4H] post-activation gates, [B][S][H] cell state
  float* ctx_out; float* hp_out;    //                          [B][S][D] context vectors, [B][S][H] h2h(h)+b
  int B, T, D, S;
  long eproj_stride_b, eproj_stride_s, hid_stride_b, hid_stride_s;
};

__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); saturates correctly through inf / 0
  const float e = __expf(2.f * x);
  return 1.f - 2.f / (e + 1.f);
}

struct AttnDecGroup {
  AttnDecParams g[MAX_GROUPS];
  int tiles, groups, pinned, vb;
};

// X3: the three recurrent products (h2h, W_ih[:, :D] on the context, W_hh) as split-fp16 x3 on v_mfma_f32_16x16x32_f16 -- h and the
// context live in LDS as fp16 hi / lo planes, the weights come as fragment-major hi / lo streams with a power-of-two prescale
// (ops.pack_fragment_major_h).  On the exact-fp32 pipe those products are 30 us of a step (576 MFMAs of 32 cycles per wave, four
// waves per SIMD); as x3 216 MFMAs of 16 cycles.
template <bool X3>
__global__ __launch_bounds__(NTH) void attn_decoder_kernel(const AttnDecGroup grp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // pinned == 1: one expert's recurrent weights (2.3 MiB) stay in one XCD's L2, all tiles of a group run on XCD (group % 8); pinned == 2
  // (six experts: 24 workgroups on every XCD instead of 32 on six of them; two experts' weights per L2 at most)
  int gi, tile_;
  if (grp.pinned == 2) {          // equal group-major runs of (group, tile) pairs per XCD: see lstm_layer_x3_kernel
    const int per = grp.groups * grp.tiles / 8;
    const int pair = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    gi = pair / grp.tiles;
    tile_ = pair - gi * grp.tiles;
  } else {
    gi = grp.pinned ? (int)(blockIdx.x % 8) + 8 * (int)((blockIdx.x / 8) / grp.tiles) : (int)blockIdx.x / grp.tiles;
    tile_ = grp.pinned ? (int)((blockIdx.x / 8) % grp.tiles) : (int)blockIdx.x % grp.tiles;
  }
  if (gi >= grp.groups) return;
  const AttnDecParams p = grp.g[gi];
  const int D = p.D, T = p.T;
  const int CLD = D + 4;
  const int CLDH = D + 8;                  // x3: fp16 row of the context planes
  float* h_lds = lds;                      // [BT][HLD]                     (x3: two fp16 planes [BT][LDH])
  float* hp_lds = X3 ? lds + (2 * BT * LDH) / 2 : h_lds + BT * HLD;        // [BT][HLD]
  float* ctx_lds = hp_lds + BT * HLD;      // [BT][CLD]                     (x3: two fp16 planes [BT][CLDH])
  float* e_lds = X3 ? ctx_lds + (2 * BT * CLDH) / 2 : ctx_lds + BT * CLD;  // [BT][T]
  float* sw_lds = e_lds + BT * T;          // [HID]
  _Float16* h_hi = reinterpret_cast<_Float16*>(h_lds);
  _Float16* h_lo = h_hi + BT * LDH;
  _Float16* c_hi = reinterpret_cast<_Float16*>(ctx_lds);
  _Float16* c_lo = c_hi + BT * CLDH;
  const float inv_h2h = X3 ? p.w_inv[0] : 1.f, inv_ih = X3 ? p.w_inv[1] : 1.f, inv_hh = X3 ? p.w_inv[2] : 1.f;
  // x3: the three fp16 weight streams (4 bytes per weight) behind buffer resources
  const __amdgpu_buffer_rsrc_t r_h2h = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_h2h, 0, X3 ? HID * HID * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_ih = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_ih, 0, X3 ? 4 * HID * D * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_hh = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hh, 0, X3 ? 4 * HID * HID * 4 : 0, 0x00020000);

  // vb = samples per workgroup (2 ... 16, the fewest that keep the launch within 256 workgroups -- attn_launch: every step
  // re-reads the workgroup's Hproj / Hb slices (66 KB per sample each), so more, smaller workgroups shorten the step);
  // rows >= vb of the 16-row MFMA tile are treated like rows beyond the batch
  const int vb = grp.vb;
  const int b0 = tile_ * vb;
  const int Bend = min(p.B, b0 + vb);
  const int t_ = threadIdx.x, lane = t_ & 63, wave = t_ >> 6;
  const int col = lane & 15, rbase = (lane >> 4) * 4;
  const int j = wave * 16 + col;

  if constexpr (X3) {
    for (int i = t_; i < BT * LDH; i += NTH) {
      const int row = i / LDH, jj = i - row * LDH;
      const int b = b0 + row;
      store_h_split(h_hi, h_lo, i, (p.h_state && b < Bend && jj < HID) ? p.h_state[(long)b * HID + jj] : 0.f);
    }
    for (int i = t_; i < BT * HLD + BT * CLDH + BT * T; i += NTH) hp_lds[i] = 0.f;   // hp, ctx planes, e: rows >= vb stay zero
  } else {
    for (int i = t_; i < BT * HLD; i += NTH) {
      const int row = i / HLD, jj = i - row * HLD;
      const int b = b0 + row;
      h_lds[i] = (p.h_state && b < Bend && jj < HID) ? p.h_state[(long)b * HID + jj] : 0.f;
    }
    for (int i = t_; i < BT * (HLD + CLD + T); i += NTH) hp_lds[i] = 0.f;     // hp, ctx, e: rows >= vb stay zero
  }
  for (int i = t_; i < HID; i += NTH) sw_lds[i] = p.w_score[i];
  float bh[4], c[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = p.b_hh ? p.b_hh[g * HID + j] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + rbase + r;
    c[r] = (p.c_state && b < Bend) ? p.c_state[(long)b * HID + j] : 0.f;
  }
  const float bj = p.b_h2h[j];
  float hlast[4] = {0.f, 0.f, 0.f, 0.f};        // h of the last step (the carried state, exact also when LDS holds the fp16 planes)
  __syncthreads();

  for (int step = 0; step < p.S; ++step) {
    // (1) hp = h2h(h) + bias
    {
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      if constexpr (X3) mma_rows_hb<1>(acc, h_hi, h_lo, r_h2h, HID, wave, lane);
      else mma_rows<1>(acc, h_lds, HLD, p.w_h2h, HID, wave, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[0][r] * inv_h2h + bj;
        hp_lds[(rbase + r) * HLD + j] = v;
        const int b = b0 + rbase + r;
        if (p.hp_out && b < Bend) p.hp_out[((long)b * p.S + step) * HID + j] = v;
      }
    }
    __syncthreads();
    // (2) e[b][t] = score . tanh(Hproj[b][t] + hp[b]); one wave per (b, t) pair, 4 channels per lane, four pairs in
    //     flight per wave so the Hproj loads (L2) and the cross-lane reductions of different pairs overlap
    {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(sw_lds + lane * 4);
      const int npair = vb * T;
      constexpr int U = 4;                 // pairs in flight per wave (8 measured the same, round 6)
      for (int pr0 = wave * U; pr0 < npair; pr0 += NW * U) {
        f32x4 hv[U];
        int rows[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int pr = pr0 + u;
          const int row = pr < npair ? pr / T : 0, t = pr < npair ? pr - row * T : 0;
          rows[u] = row;
          const int b = b0 + row;
          hv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (pr < npair && b < Bend) hv[u] = *reinterpret_cast<const f32x4*>(p.Hproj + ((long)b * T + t) * HID + lane * 4);
        }
        float sacc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const f32x4 pv = *reinterpret_cast<const f32x4*>(hp_lds + rows[u] * HLD + lane * 4);
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) s = fmaf(wv[k], fast_tanh(hv[u][k] + pv[k]), s);
          sacc[u] = s;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
          for (int u = 0; u < U; ++u) sacc[u] += __shfl_xor(sacc[u], o);
        }
        if (lane < U && pr0 + lane < npair) {
          float v = sacc[0];
#pragma unroll
          for (int u = 1; u < U; ++u) v = lane == u ? sacc[u] : v;
          e_lds[pr0 + lane] = (b0 + (pr0 + lane) / T < p.B) ? v : 0.f;
        }
      }
    }
    __syncthreads();
    // (3) softmax over t, one wave per sample
    if (wave < vb) {
      const int row = wave;
      float m = -INFINITY;
      for (int t = lane; t < T; t += 64) m = fmaxf(m, e_lds[row * T + t]);
      m = wave_max(m);
      float sum = 0.f;
      for (int t = lane; t < T; t += 64) {
        const float v = expf(e_lds[row * T + t] - m);
        e_lds[row * T + t] = v;
        sum += v;
      }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      for (int t = lane; t < T; t += 64) {
        const float a = e_lds[row * T + t] * inv;
        e_lds[row * T + t] = a;
        const int b = b0 + row;
        if (p.alpha_out && b < Bend) p.alpha_out[((long)b * p.S + step) * T + t] = a;
      }
    }
    __syncthreads();
    // embedding half of the LSTMCell input projection for this step: issued here, it lands during phase (4) and becomes the INITIAL value
    // of the gate accumulators (x the W_ih stream's power-of-two prescale: exact) -- sixteen registers live over one phase, not the step
    f32x4 acc5[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + rbase + r;
      const float* ep = p.eproj + (long)(b < Bend ? b : 0) * p.eproj_stride_b + (long)step * p.eproj_stride_s + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) acc5[g][r] = ep[g * HID];
    }
    // (4) context[b][:] = sum_t alpha[b][t] * Hb[b][t][:]
    for (int it = t_; it < vb * (D / 4); it += NTH) {
      const int row = it / (D / 4), c4 = it - row * (D / 4);
      const int b = b0 + row;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      if (b < Bend) {
        const float* hb = p.Hb + (long)b * T * D + c4 * 4;
        int t = 0;
#pragma unroll 1
        for (; t + 4 <= T; t += 4) {                 // four independent loads in flight per lane, 16 waves per CU (eight: no faster)
          f32x4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(hb + (long)(t + u) * D);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float w = e_lds[row * T + t + u];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = fmaf(w, v[u][k], a[k]);
          }
        }
        for (; t < T; ++t) {
          const float w = e_lds[row * T + t];
          const f32x4 v = *reinterpret_cast<const f32x4*>(hb + (long)t * D);
#pragma unroll
          for (int k = 0; k < 4; ++k) a[k] = fmaf(w, v[k], a[k]);
        }
      }
      if constexpr (X3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) store_h_split(c_hi, c_lo, row * CLDH + c4 * 4 + k, a[k]);
      } else {
        *reinterpret_cast<f32x4*>(ctx_lds + row * CLD + c4 * 4) = a;
      }
      if (p.ctx_out && b < Bend) *reinterpret_cast<f32x4*>(p.ctx_out + ((long)b * p.S + step) * D + c4 * 4) = a;
    }
    __syncthreads();
    // (5) gates = eproj + ctx . W_ih[:, :D]^T + h . W_hh^T ; (6) LSTM cell
    {
      f32x4 acc[4];
      const float pre_ih = 1.f / inv_ih;               // (a power of two)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = acc5[g] * pre_ih;
      if constexpr (X3) {
        mma_rows_hb<4>(acc, c_hi, c_lo, r_ih, D, wave, lane, CLDH);
        const float ratio = inv_ih / inv_hh;            // (powers of two: exact)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] *= ratio;
        mma_rows_hb<4>(acc, h_hi, h_lo, r_hh, HID, wave, lane);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] *= inv_hh;
      } else {
        mma_rows<4>(acc, ctx_lds, CLD, p.w_ih, D, wave, lane);
        mma_rows<4>(acc, h_lds, HLD, p.w_hh, HID, wave, lane);
      }
      __syncthreads();  // every wave has finished reading h_lds
      float h[4], act[4][4];
      lstm_pointwise0(acc, bh, c, h, act);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + r, b = b0 + row;
        if (b < Bend) {
          p.hid[(long)b * p.hid_stride_b + (long)step * p.hid_stride_s + j] = h[r];
          if (p.gates_out) {
            const long base = (long)b * p.S + step;
#pragma unroll
            for (int g = 0; g < 4; ++g) p.gates_out[base * 4 * HID + g * HID + j] = act[g][r];
            p.c_out[base * HID + j] = c[r];
          }
        }
        if constexpr (X3) store_h_split(h_hi, h_lo, row * LDH + j, b < Bend ? h[r] : 0.f);
        else h_lds[row * HLD + j] = b < Bend ? h[r] : 0.f;
        hlast[r] = h[r];
      }
    }
    __syncthreads();
  }

  if (p.h_state) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + r, b = b0 + row;
      if (b < Bend) {
        p.h_state[(long)b * HID + j] = hlast[r];
        p.c_state[(long)b * HID + j] = c[r];
      }
    }
  }
}

// rows[i][:] = table[min-cut(idx[i])][:]   (Attention.cut_unknown: indices >= num_class map to 0)
__global__ void embed_gather_kernel(const long* __restrict__ idx, const float* __restrict__ table,
                                    float* __restrict__ out, long n, int E, int num_class, long idx_stride, int S) {
  const long i = blockIdx.x;
  if (i >= n) return;
  const long b = i / S, s = i - b * S;
  long k = idx[b * idx_stride + s];
  if (k >= num_class || k < 0) k = 0;
  for (int e = threadIdx.x; e < E; e += blockDim.x) out[i * E + e] = table[k * E + e];
}

}  // namespace

static int lstm_launch(LstmGroup& grp, int groups, hipStream_t st) {
  grp.nsets = groups * grp.ndir;
  grp.pinned = grp.nsets > 2 && grp.tiles * ceil_div(grp.nsets, 8) <= 32;   // (a single layer already fits every L2)
  const int blocks = grp.pinned ? 8 * ceil_div(grp.nsets, 8) * grp.tiles : grp.nsets * grp.tiles;
  hipLaunchKernelGGL(lstm_layer_kernel, dim3(blocks), dim3(NTH), 0, st, grp);
  MRN_LAUNCH_CHECK("lstm_layer");
  return MRN_OK;
}

MRN_EXPORT int mrn_lstm_layer_fwd_f32(const float* xproj, const float* w_hh, const float* b_hh, float* out,
                                      float* gates_out, float* c_out, int B, int T, int hidden, int ndir, void* stream) {
  MRN_CHECK_ARG(xproj && w_hh && out, "mrn_lstm_layer_fwd_f32: null operand");
  MRN_CHECK_ARG(hidden == HID, "mrn_lstm_layer_fwd_f32: hidden=%d unsupported (library is built for %d)", hidden, HID);
  MRN_CHECK_ARG(ndir == 1 || ndir == 2, "mrn_lstm_layer_fwd_f32: ndir=%d", ndir);
  MRN_CHECK_ARG((gates_out == nullptr) == (c_out == nullptr), "mrn_lstm_layer_fwd_f32: gates_out / c_out must come together");
  if (B == 0 || T == 0) return MRN_OK;
  LstmGroup grp;
  memset(&grp, 0, sizeof(grp));
  grp.g[0] = LstmParams{xproj, w_hh, b_hh, out, gates_out, c_out};
  grp.tiles = ceil_div(B, BT); grp.B = B; grp.T = T; grp.ndir = ndir;
  return lstm_launch(grp, 1, (hipStream_t)stream);
}

// `groups` independent layers of identical geometry in one launch.  xproj / w_hh / b_hh / out are HOST arrays of
// `groups` device pointers (b_hh may be NULL or hold NULL entries).
MRN_EXPORT int mrn_lstm_layer_fwd_grouped_f32(const void* const* xproj, const void* const* w_hh, const void* const* b_hh,
                                              const void* const* out, int groups, int B, int T, int hidden, int ndir,
                                              void* stream) {
  MRN_CHECK_ARG(xproj && w_hh && out && groups >= 1, "mrn_lstm_layer_fwd_grouped_f32: null operand");
  MRN_CHECK_ARG(hidden == HID, "mrn_lstm_layer_fwd_grouped_f32: hidden=%d unsupported (library is built for %d)", hidden, HID);
  MRN_CHECK_ARG(ndir == 1 || ndir == 2, "mrn_lstm_layer_fwd_grouped_f32: ndir=%d", ndir);
  if (B == 0 || T == 0) return MRN_OK;
  for (int g0 = 0; g0 < groups; g0 += MAX_GROUPS) {
    const int n = groups - g0 < MAX_GROUPS ? groups - g0 : MAX_GROUPS;
    LstmGroup grp;
    memset(&grp, 0, sizeof(grp));
    for (int i = 0; i < n; ++i) {
      MRN_CHECK_ARG(xproj[g0 + i] && w_hh[g0 + i] && out[g0 + i], "mrn_lstm_layer_fwd_grouped_f32: null operand in group %d", g0 + i);
      grp.g[i] = LstmParams{(const float*)xproj[g0 + i], (const float*)w_hh[g0 + i], b_hh ? (const float*)b_hh[g0 + i] : nullptr,
                            (float*)out[g0 + i], nullptr, nullptr};
    }
    grp.tiles = ceil_div(B, BT); grp.B = B; grp.T = T; grp.ndir = ndir;
    const int rc = lstm_launch(grp, n, (hipStream_t)stream);
    if (rc) return rc;
  }
  return MRN_OK;
}

// Inference-only LSTM layers of `groups` frozen experts with the recurrent product on the f16 MFMA (split-fp16 x3).
// w_hh: HOST array of device pointers to the fragment-major fp16 streams ([ndir][16][4][H/32][64][hi 8 | lo 8 halves],
// ops.pack_fragment_major_h), w_inv: HOST array of device float[ndir] = 1 / prescale of each direction's weights.
MRN_EXPORT int mrn_lstm_layer_fwd_x3_grouped(const void* const* xproj, const void* const* w_hh, const void* const* w_inv,
                                             const void* const* b_hh, const void* const* out, int groups, int B, int T,
                                             int hidden, int ndir, void* stream) {
  MRN_CHECK_ARG(xproj && w_hh && w_inv && out && groups >= 1, "mrn_lstm_layer_fwd_x3_grouped: null operand");
  MRN_CHECK_ARG(hidden == HID, "mrn_lstm_layer_fwd_x3_grouped: hidden=%d unsupported (library is built for %d)", hidden, HID);
  MRN_CHECK_ARG(ndir == 1 || ndir == 2, "mrn_lstm_layer_fwd_x3_grouped: ndir=%d", ndir);
  if (B == 0 || T == 0) return MRN_OK;
  for (int g0 = 0; g0 < groups; g0 += MAX_GROUPS) {
    const int n = groups - g0 < MAX_GROUPS ? groups - g0 : MAX_GROUPS;
    LstmX3Group grp;
    memset(&grp, 0, sizeof(grp));
    for (int i = 0; i < n; ++i) {
      MRN_CHECK_ARG(xproj[g0 + i] && w_hh[g0 + i] && w_inv[g0 + i] && out[g0 + i], "mrn_lstm_layer_fwd_x3_grouped: null operand in group %d", g0 + i);
      grp.g[i] = LstmX3Params{(const float*)xproj[g0 + i], (const unsigned char*)w_hh[g0 + i], (const float*)w_inv[g0 + i],
                              b_hh ? (const float*)b_hh[g0 + i] : nullptr, (float*)out[g0 + i]};
    }
    grp.B = B; grp.T = T; grp.ndir = ndir;
    const int rc = lstm_x3_launch(grp, n, true, (hipStream_t)stream);
    if (rc) return rc;
    MRN_LAUNCH_CHECK("lstm_layer_x3");
  }
  return MRN_OK;
}

// One layer being TRAINED with the recurrent product as split-fp16 x3 (the arithmetic of the trained convolutions, fp16x3s): the
// forward of mrn_lstm_layer_fwd_x3_grouped for one network, plus the saves mrn_lstm_layer_bwd_f32 reads (gates [B][T][ndir][4H]
// post-activation, cseq [B][T][ndir][H]).  Half the time per step of the exact-fp32 kernel (96 MFMAs of 16 cycles instead of 256 of 32).
MRN_EXPORT int mrn_lstm_layer_fwd_x3_save(const float* xproj, const void* w_hh, const float* w_inv, const float* b_hh, float* out,
                                          float* gates_out, float* c_out, int B, int T, int hidden, int ndir, void* stream) {
  MRN_CHECK_ARG(xproj && w_hh && w_inv && out && gates_out && c_out, "mrn_lstm_layer_fwd_x3_save: null operand");
  MRN_CHECK_ARG(hidden == HID, "mrn_lstm_layer_fwd_x3_save: hidden=%d unsupported (library is built for %d)", hidden, HID);
  MRN_CHECK_ARG(ndir == 1 || ndir == 2, "mrn_lstm_layer_fwd_x3_save: ndir=%d", ndir);
  if (B == 0 || T == 0) return MRN_OK;
  LstmX3Group grp;
  memset(&grp, 0, sizeof(grp));
  grp.g[0] = LstmX3Params{xproj, (const unsigned char*)w_hh, w_inv, b_hh, out, gates_out, c_out};
  grp.B = B; grp.T = T; grp.ndir = ndir;
  const int rc = lstm_x3_launch(grp, 1, false, (hipStream_t)stream);
  if (rc) return rc;
  MRN_LAUNCH_CHECK("lstm_layer_x3_save");
  return MRN_OK;
}

static int attn_fill(AttnDecParams& p, const float* Hb, const float* Hproj, const float* eproj, int64_t eproj_stride_b,
                     int64_t eproj_stride_s, const float* w_h2h, const float* b_h2h, const float* w_score,
                     const float* w_ih_ctx, const float* w_hh, const float* b_hh, float* hid, int64_t hid_stride_b,
                     int64_t hid_stride_s, int B, int T, int D, int S) {
  memset(&p, 0, sizeof(p));
  p.Hb = Hb; p.Hproj = Hproj; p.eproj = eproj; p.w_h2h = w_h2h; p.b_h2h = b_h2h; p.w_score = w_score;
  p.w_ih = w_ih_ctx; p.w_hh = w_hh; p.b_hh = b_hh; p.hid = hid;
  p.B = B; p.T = T; p.D = D; p.S = S;
  p.eproj_stride_b = eproj_stride_b; p.eproj_stride_s = eproj_stride_s;
  p.hid_stride_b = hid_stride_b; p.hid_stride_s = hid_stride_s;
  return MRN_OK;
}

static int attn_launch(AttnDecGroup& grp, int groups, int D, int T, hipStream_t st) {
  grp.groups = groups;
  const int B = grp.g[0].B;
  // samples per workgroup: the fewest (from two) that keep the launch within one workgroup per CU.  The step is a chain of dependent
  // phases whose length grows with the samples a workgroup walks through (B = 256, one expert, us per launch: 1100 at 1, 1130 at 2, 1315 at
  // 4, 1700 at 8, 2580 at 16 -- tools/bench_decoder.py with MRN_ATTN_VB), and a second round of workgroups doubles it (three experts:
  // 2330 at 2, 1380 at 4).  Not one: a trained network's launch then fills the chip, and the DER step, which runs its two heads'
  // decoders side by side, loses 3 % (loop A gains 1.5 %: tools/probe/vb_sweep.sh)
  grp.vb = BT;
  for (int v = 2; v < BT; v *= 2)
    if ((long)groups * ceil_div(B, v) <= 256) { grp.vb = v; break; }
  static const int forced_vb = getenv("MRN_ATTN_VB") ? atoi(getenv("MRN_ATTN_VB")) : 0;     // (A/B switch, read once)
  if (forced_vb == 1 || forced_vb == 2 || forced_vb == 4 || forced_vb == 8 || forced_vb == 16) grp.vb = forced_vb;
  grp.tiles = ceil_div(B, grp.vb);
  grp.pinned = groups > 1 && grp.tiles * ceil_div(groups, 8) <= 32;
  static const bool balance = !(getenv("MRN_ATTN_BALANCE") && atoi(getenv("MRN_ATTN_BALANCE")) == 0);     // (A/B switch, read once)
  if (balance && groups > 1 && (groups * grp.tiles) % 8 == 0 && groups * grp.tiles <= 256) grp.pinned = 2;
  const bool x3 = grp.g[0].w_inv != nullptr;
  const size_t lds = sizeof(float) * (2 * BT * HLD + BT * (D + 4) + BT * T + HID) + (x3 ? 1024 : 0);
  MRN_CHECK_ARG(lds <= 160 * 1024, "mrn_attn_decoder_fwd: LDS budget exceeded (D=%d T=%d)", D, T);
  MRN_CHECK_ARG(!x3 || D % 32 == 0, "mrn_attn_decoder_fwd (x3): D=%d must be a multiple of 32", D);
  const dim3 grid(grp.pinned == 1 ? 8 * ceil_div(groups, 8) * grp.tiles : groups * grp.tiles);
  if (x3) {
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)attn_decoder_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_decoder_kernel<true>, grid, dim3(NTH), lds, st, grp);
  } else {
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)attn_decoder_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_decoder_kernel<false>, grid, dim3(NTH), lds, st, grp);
  }
  MRN_LAUNCH_CHECK("attn_decoder");
  return MRN_OK;
}

MRN_EXPORT int mrn_attn_decoder_fwd_f32(const float* Hb, const float* Hproj, const float* eproj, int64_t eproj_stride_b,
                                        int64_t eproj_stride_s, const float* w_h2h, const float* b_h2h,
                                        const float* w_score, const float* w_ih_ctx, const float* w_hh,
                                        const float* b_hh, float* hid, int64_t hid_stride_b, int64_t hid_stride_s, float* h_state,
                                        float* c_state, float* alpha_out, float* gates_out, float* c_out, float* ctx_out,
                                        float* hp_out, int B, int T, int D, int S, int hidden, void* stream) {
  MRN_CHECK_ARG(Hb && Hproj && eproj && w_h2h && b_h2h && w_score && w_ih_ctx && w_hh && hid, "mrn_attn_decoder_fwd_f32: null operand");
  MRN_CHECK_ARG(hidden == HID, "mrn_attn_decoder_fwd_f32: hidden=%d unsupported (library is built for %d)", hidden, HID);
  MRN_CHECK_ARG(D % 16 == 0 && D > 0, "mrn_attn_decoder_fwd_f32: D=%d must be a multiple of 16", D);
  MRN_CHECK_ARG((h_state == nullptr) == (c_state == nullptr), "mrn_attn_decoder_fwd_f32: h_state/c_state must come together");
  if (B == 0 || S == 0) return MRN_OK;
  MRN_CHECK_ARG(!gates_out || (c_out && ctx_out && hp_out && alpha_out), "mrn_attn_decoder_fwd_f32: training saves must all be given");
  AttnDecGroup grp;
  memset(&grp, 0, sizeof(grp));
  AttnDecParams& p = grp.g[0];
  attn_fill(p, Hb, Hproj, eproj, eproj_stride_b, eproj_stride_s, w_h2h, b_h2h, w_score, w_ih_ctx, w_hh, b_hh, hid, hid_stride_b,
            hid_stride_s, B, T, D, S);
  p.h_state = h_state; p.c_state = c_state;
  p.alpha_out = alpha_out; p.gates_out = gates_out; p.c_out = c_out; p.ctx_out = ctx_out; p.hp_out = hp_out;
  grp.tiles = ceil_div(B, BT);
  return attn_launch(grp, 1, D, T, (hipStream_t)stream);
}

// Teacher-forced decoders of `groups` experts (identical geometry) in one launch.  Every pointer argument is a HOST
// array of `groups` device pointers; strides are shared.  No carried state / training saves.
MRN_EXPORT int mrn_attn_decoder_fwd_grouped_f32(const void* const* Hb, const void* const* Hproj, const void* const* eproj,
                                                int64_t eproj_stride_b, int64_t eproj_stride_s, const void* const* w_h2h,
                                                const void* const* b_h2h, const void* const* w_score,
                                                const void* const* w_ih_ctx, const void* const* w_hh, const void* const* b_hh,
                                                const void* const* hid, int64_t hid_stride_b, int64_t hid_stride_s, int groups,
                                                int B, int T, int D, int S, int hidden, void* stream) {
  MRN_CHECK_ARG(Hb && Hproj && eproj && w_h2h && b_h2h && w_score && w_ih_ctx && w_hh && hid && groups >= 1,
                "mrn_attn_decoder_fwd_grouped_f32: null operand");
  MRN_CHECK_ARG(hidden == HID, "mrn_attn_decoder_fwd_grouped_f32: hidden=%d unsupported (library is built for %d)", hidden, HID);
  MRN_CHECK_ARG(D % 16 == 0 && D > 0, "mrn_attn_decoder_fwd_grouped_f32: D=%d must be a multiple of 16", D);
  if (B == 0 || S == 0) return MRN_OK;
  for (int g0 = 0; g0 < groups; g0 += MAX_GROUPS) {
    const int n = groups - g0 < MAX_GROUPS ? groups - g0 : MAX_GROUPS;
    AttnDecGroup grp;
    memset(&grp, 0, sizeof(grp));
    for (int i = 0; i < n; ++i) {
      const int g = g0 + i;
      MRN_CHECK_ARG(Hb[g] && Hproj[g] && eproj[g] && w_h2h[g] && b_h2h[g] && w_score[g] && w_ih_ctx[g] && w_hh[g] && hid[g],
                    "mrn_attn_decoder_fwd_grouped_f32: null operand in group %d", g);
      attn_fill(grp.g[i], (const float*)Hb[g], (const float*)Hproj[g], (const float*)eproj[g], eproj_stride_b, eproj_stride_s,
                (const float*)w_h2h[g], (const float*)b_h2h[g], (const float*)w_score[g], (const float*)w_ih_ctx[g],
                (const float*)w_hh[g], b_hh ? (const float*)b_hh[g] : nullptr, (float*)hid[g], hid_stride_b, hid_stride_s, B, T, D, S);
    }
    grp.tiles = ceil_div(B, BT);
    const int rc = attn_launch(grp, n, D, T, (hipStream_t)stream);
    if (rc) return rc;
  }
  return MRN_OK;
}

// The same decoders with the three recurrent products as split-fp16 x3: w_h2h / w_ih_ctx / w_hh are the fragment-major fp16 hi / lo
// streams of ops.pack_fragment_major_h, w_inv a device float[3] = 1 / prescale of each (per group in the grouped form).  D % 32 == 0.
MRN_EXPORT int mrn_attn_decoder_fwd_x3(const float* Hb, const float* Hproj, const float* eproj, int64_t eproj_stride_b,
                                       int64_t eproj_stride_s, const void* w_h2h, const float* b_h2h, const float* w_score,
                                       const void* w_ih_ctx, const void* w_hh, const float* w_inv, const float* b_hh, float* hid,
                                       int64_t hid_stride_b, int64_t hid_stride_s, float* h_state, float* c_state, float* alpha_out,
                                       float* gates_out, float* c_out, float* ctx_out, float* hp_out, int B, int T, int D, int S,
                                       int hidden, void* stream) {
  MRN_CHECK_ARG(Hb && Hproj && eproj && w_h2h && b_h2h && w_score && w_ih_ctx && w_hh && w_inv && hid, "mrn_attn_decoder_fwd_x3: null operand");
  MRN_CHECK_ARG(hidden == HID, "mrn_attn_decoder_fwd_x3: hidden=%d unsupported (library is built for %d)", hidden, HID);
  MRN_CHECK_ARG(D % 32 == 0 && D > 0, "mrn_attn_decoder_fwd_x3: D=%d must be a multiple of 32", D);
  MRN_CHECK_ARG((h_state == nullptr) == (c_state == nullptr), "mrn_attn_decoder_fwd_x3: h_state/c_state must come together");
  if (B == 0 || S == 0) return MRN_OK;
  MRN_CHECK_ARG(!gates_out || (c_out && ctx_out && hp_out && alpha_out), "mrn_attn_decoder_fwd_x3: training saves must all be given");
  AttnDecGroup grp;
  memset(&grp, 0, sizeof(grp));
  AttnDecParams& p = grp.g[0];
  attn_fill(p, Hb, Hproj, eproj, eproj_stride_b, eproj_stride_s, (const float*)w_h2h, b_h2h, w_score, (const float*)w_ih_ctx,
            (const float*)w_hh, b_hh, hid, hid_stride_b, hid_stride_s, B, T, D, S);
  p.w_inv = w_inv;
  p.h_state = h_state; p.c_state = c_state;
  p.alpha_out = alpha_out; p.gates_out = gates_out; p.c_out = c_out; p.ctx_out = ctx_out; p.hp_out = hp_out;
  grp.tiles = ceil_div(B, BT);
  return attn_launch(grp, 1, D, T, (hipStream_t)stream);
}

MRN_EXPORT int mrn_attn_decoder_fwd_x3_grouped(const void* const* Hb, const void* const* Hproj, const void* const* eproj,
                                               int64_t eproj_stride_b, int64_t eproj_stride_s, const void* const* w_h2h,
                                               const void* const* b_h2h, const void* const* w_score, const void* const* w_ih_ctx,
                                               const void* const* w_hh, const void* const* w_inv, const void* const* b_hh,
                                               const void* const* hid, int64_t hid_stride_b, int64_t hid_stride_s, int groups, int B,
                                               int T, int D, int S, int hidden, void* stream) {
  MRN_CHECK_ARG(Hb && Hproj && eproj && w_h2h && b_h2h && w_score && w_ih_ctx && w_hh && w_inv && hid && groups >= 1,
                "mrn_attn_decoder_fwd_x3_grouped: null operand");
  MRN_CHECK_ARG(hidden == HID, "mrn_attn_decoder_fwd_x3_grouped: hidden=%d unsupported (library is built for %d)", hidden, HID);
  MRN_CHECK_ARG(D % 32 == 0 && D > 0, "mrn_attn_decoder_fwd_x3_grouped: D=%d must be a multiple of 32", D);
  if (B == 0 || S == 0) return MRN_OK;
  for (int g0 = 0; g0 < groups; g0 += MAX_GROUPS) {
    const int n = groups - g0 < MAX_GROUPS ? groups - g0 : MAX_GROUPS;
    AttnDecGroup grp;
    memset(&grp, 0, sizeof(grp));
    for (int i = 0; i < n; ++i) {
      const int g = g0 + i;
      MRN_CHECK_ARG(Hb[g] && Hproj[g] && eproj[g] && w_h2h[g] && b_h2h[g] && w_score[g] && w_ih_ctx[g] && w_hh[g] && w_inv[g] && hid[g],
                    "mrn_attn_decoder_fwd_x3_grouped: null operand in group %d", g);
      attn_fill(grp.g[i], (const float*)Hb[g], (const float*)Hproj[g], (const float*)eproj[g], eproj_stride_b, eproj_stride_s,
                (const float*)w_h2h[g], (const float*)b_h2h[g], (const float*)w_score[g], (const float*)w_ih_ctx[g],
                (const float*)w_hh[g], b_hh ? (const float*)b_hh[g] : nullptr, (float*)hid[g], hid_stride_b, hid_stride_s, B, T, D, S);
      grp.g[i].w_inv = (const float*)w_inv[g];
    }
    grp.tiles = ceil_div(B, BT);
    const int rc = attn_launch(grp, n, D, T, (hipStream_t)stream);
    if (rc) return rc;
  }
  return MRN_OK;
}

MRN_EXPORT int mrn_embed_gather_f32(const int64_t* idx, int64_t idx_stride, const float* table, float* out, int B,
                                    int S, int E, int num_class, void* stream) {
  MRN_CHECK_ARG(idx && table && out, "mrn_embed_gather_f32: null operand");
  const long n = (long)B * S;
  if (n == 0) return MRN_OK;
  hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, (const long*)idx, table, out,
                     n, E, num_class, (long)idx_stride, S);
  MRN_LAUNCH_CHECK("embed_gather");
  return MRN_OK;
}
