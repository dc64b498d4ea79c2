// Fused Mlp of an SVTR mixing block for G lock-step frozen experts (reference modules/svtr.py:46-67: fc1 -> GELU -> fc2), one kernel,
// the 4C-wide hidden activation never leaves the registers.
//
// Arithmetic is the grouped Linear path's (conv_x3.hip): split-fp16 x3 products (lo*hi + hi*lo + hi*hi, 22 significand bits) on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation, weights prescaled by a per-tensor power of two, bias and GELU in fp32, the hidden
// activation re-split to hi + lo before the second product.
//
// Formulation: everything is computed TRANSPOSED, tokens on the MFMA N axis.  A wave owns 32 tokens; their C input channels sit in
// registers as the B operand for the whole kernel.  For every block of 32 hidden units
//     H^T[32 hidden x 32 tokens] = W1[32 x C] . X^T[C x 32]          (A = W1 rows from LDS, B = the token fragments)
// lands in the MFMA result layout: lane = token (lane & 31), register e = hidden unit (e & 3) + 8 (e >> 2) + 4 (lane >> 5).  The B
// operand of the second product needs, per lane, 8 consecutive reduction indices -- and registers e = 8s .. 8s+7 ARE a valid set of 8
// if the reduction index of fc2 is permuted accordingly (position p = 16 s + 8 h + j of a 32-unit block <-> unit (j & 3) + 8 (2 s + (j >> 2)) + 4 h).
// So bias + GELU + hi/lo split happen in place and the registers feed
//     Y^T[C x 32 tokens] += W2'[C x 32 hidden] . H^T                  (A = permuted W2 rows from LDS, B = the registers just written)
// without any cross-lane movement or LDS round trip of the hidden tensor: the chained-MFMA form.  The 8 waves of a workgroup (256
// tokens) share the weight slabs of a hidden block (W1: 32 rows x C, W2': C rows x 32) through a double-buffered LDS ring filled by
// direct-to-LDS DMA; the weights of one expert (8 C^2 x 4 bytes: 128 KiB at C = 64, 512 KiB at C = 128) are re-read per 256 tokens from L2.
// HBM traffic: the HL32 input once (4 B / element), the fp32 branch output once -- against 10 x that for fc1 and fc2 as two GEMMs.
// C = 64 and 128 (SVTR stages 1 and 2): eight waves of 32 tokens, two per SIMD.  C = 256 (stage 3): the token fragments (128 registers)
// and the output accumulators (128) need the 512-register form -- four waves, one per SIMD, 128 tokens per workgroup, 2 x 64 KiB of
// slab ring (the round-4 pricing: fc1 + fc2 as two GEMMs 0.89 ms per block against ~0.5 here).
#include "common.hpp"

namespace {

typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct MlpParams {
  const unsigned char* x_hl;      // [rows][C/32][128 B] LayerNorm output (HL32)
  const unsigned char* w1;        // [G][4C][C/32][128 B]
  const unsigned char* w2;        // [G][C][4C/32][128 B], hidden index permuted inside every 32-block (see above)
  const float* b1;                // [G][4C]
  const float* b2;                // [G][C]
  const float* s1;                // [G][2] {s, 1/s} of W1
  const float* s2;                // [G][2]
  float* y;                       // [rows][C]
  long rows, rows_per_group;
  int tiles_per_group;
  // TAIL form (C = 256): x_hl is the attention context; proj -> DropPath-scaled residual add -> LayerNorm2 run in front of fc1
  const unsigned char* wp;        // [G][C][C/32][128 B] proj weights
  const float* sp;                // [G][2]
  const float* bp;                // [G][C]
  const float* drop;              // [rows / rows_per_drop] DropPath multipliers of the attention branch, or null
  const float* gamma;             // [G][C] LayerNorm2
  const float* beta;              // [G][C]
  float* x_res;                   // [rows][C] residual stream, updated in place: x += drop * proj(ctx)
  long rows_per_drop;
  float eps;
};

__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16v8*>(&a), *reinterpret_cast<const f16v8*>(&b), c, 0, 0, 0);
}

// TAIL (C = 256 only): the second half of a stage-3 mixing block in one launch -- modules/svtr.py:196-204 from the attention context on:
//   x <- x + drop * (ctx Wproj^T + bproj);  y = LayerNorm2(x);  branch = fc2(GELU(fc1(y)))
// The proj product is one more link of the chain: A = Wproj rows from the slab ring, B = the context's token fragments; its result
// lands as lane = token, register = channel, so the residual add and LayerNorm2 (sum over a lane's 128 registers + one cross-half
// shuffle) happen in registers, and registers 8 s .. 8 s + 7 of a 32-channel block are again a valid B operand once fc1's INPUT
// channels are permuted inside every 32-block the way fc2's hidden units are (a static repack of W1).
#ifdef MRN_MPROBE_TIMING
// timing probe (never in the product build; bash tools/build_probe.sh MRN_MPROBE_TIMING svtr_mlp.hip): per-wave shader-clock totals
__device__ unsigned long long g_mlp_dbg[8];
extern "C" __attribute__((visibility("default"))) int mrn_mlp_dbg_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_dbg), sizeof(g_mlp_dbg)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_dbg), z, sizeof(z));
  }
  return 0;
}
#define MTICK(var) const long var = __builtin_readcyclecounter()
#define MADD(slot, a, b) dbg_acc[slot] += (b) - (a)
#else
#define MTICK(var)
#define MADD(slot, a, b)
#endif

#ifndef MRN_MLP_WAVES
#define MRN_MLP_WAVES 8       // waves per workgroup at C = 64 / 128 (what-if: 4 = two independent 128-token workgroups per CU)
#endif
template <int C, bool TAIL>
__global__ __launch_bounds__(C == 256 ? 256 : MRN_MLP_WAVES * 64, C == 256 ? 1 : 2) void svtr_mlp_kernel(const MlpParams p) {
#ifdef MRN_MPROBE_TIMING
  long dbg_acc[6] = {0, 0, 0, 0, 0, 0};
  const long dbg_t0 = __builtin_readcyclecounter();
#endif
  static_assert(!TAIL || C == 256, "the tail form is the 512-register form");
  constexpr int NW = C == 256 ? 4 : MRN_MLP_WAVES, CB = C / 32, KB = C / 16, HID = 4 * C, NH = HID / 32, OC = C / 32;
  constexpr int W1_SLAB = CB * 32 * 128;            // 32 hidden rows x C channels, [channel block][row][128 B]
  constexpr int W2_SLAB = C * 128;                  // C output rows x one 32-hidden line
  constexpr int SLAB = W1_SLAB + W2_SLAB;
  constexpr int N1 = W1_SLAB / 1024, N2 = W2_SLAB / 1024;        // 1-KiB DMA instructions per slab part
  static_assert((N1 + N2) % NW == 0, "slab DMA instructions divide over the waves");
  constexpr int NDMA = (N1 + N2) / NW;
  constexpr bool SKEW = C == 256;                   // (the one-wave-per-SIMD form: see the main loop)
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
  float* b1_lds = reinterpret_cast<float*>(lds + 2 * SLAB);     // [HID]
  float* ln_lds = b1_lds + HID;                                  // TAIL: [3][C] = LayerNorm2 gamma, beta, proj bias
  float* b2_lds = ln_lds + (TAIL ? 3 * C : 0);                   // [C] fc2 bias: read in the epilogue between the stores (see there)

  // (block b runs on XCD b % 8: consecutive LOGICAL tiles share an XCD, so an XCD's L2 holds the 0.5 - 2 MB weights of one or two experts
  // instead of all six -- the slabs are re-read from L2 by every workgroup)
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int g = lid / p.tiles_per_group, tile = lid % p.tiles_per_group;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, tok_l = lane & 31;
  const long row0 = (long)g * p.rows_per_group + (long)tile * (NW * 32) + wave * 32;
  const long row_end = min((long)(g + 1) * p.rows_per_group, p.rows);
  const long row = row0 + tok_l;
  const bool ok = row < row_end;

  // ---- this wave's tokens as MFMA B-operand fragments: k-block kb (16 channels) = 16-byte chunk (kb & 1) * 2 + half of the hi / lo
  // half-line of channel block kb >> 1
  u32x4 xh[KB], xl[KB];
  {
    const unsigned char* xr = p.x_hl + row * (long)CB * 128;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int off = (kb >> 1) * 128 + (((kb & 1) * 2 + half) << 4);
      xh[kb] = ok ? *reinterpret_cast<const u32x4*>(xr + off) : u32x4{0u, 0u, 0u, 0u};
      xl[kb] = ok ? *reinterpret_cast<const u32x4*>(xr + off + 64) : u32x4{0u, 0u, 0u, 0u};
    }
  }
  for (int i = t; i < HID; i += NW * 64) b1_lds[i] = p.b1[(long)g * HID + i];
  for (int i = t; i < C; i += NW * 64) b2_lds[i] = p.b2[(long)g * C + i];
  if (TAIL) {
    for (int i = t; i < C; i += NW * 64) {
      ln_lds[i] = p.gamma[(long)g * C + i];
      ln_lds[C + i] = p.beta[(long)g * C + i];
      ln_lds[2 * C + i] = p.bp[(long)g * C + i];
    }
  }

  // ---- weight slabs through LDS: DMA instruction d (0 .. N1+N2-1) moves 8 rows x 128 B; lane -> row 8 d' + lane / 8, chunk lane & 7,
  // source chunk XOR-swizzled with (row >> 1) & 7 as in conv_x3.hip (conflict-free ds_read_b128 fragment reads)
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w1 + (long)g * HID * CB * 128), 0, HID * CB * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w2 + (long)g * C * NH * 128), 0, C * NH * 128, 0x00020000);
  // W1 rows of hidden block h1 -> the W1 region of ring buffer h1 & 1, W2' columns of hidden block h2 -> the W2 region of buffer h2 & 1
  // (either may be -1: nothing); the plain schedule moves both parts of one block, the skewed one (C = 256) W1 one block ahead of W2'
  auto issue2 = [&](int h1, int h2) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int d = i * NW + wave;
      if (d < N1) {                                 // W1: LDS [cb][32 rows]; source row = h1 * 32 + r, line cb
        if (h1 < 0) continue;
        const int cb = d / 4, r = (d % 4) * 8 + (lane >> 3);
        const int coff = ((lane & 7) ^ ((r >> 1) & 7)) << 4;
        const int voff = ((h1 * 32 + r) * CB + cb) * 128 + coff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr_t)(lds + (h1 & 1) * SLAB + d * 1024), 16, voff, 0, 0, 0);
      } else {                                      // W2': LDS [C rows]; source row r, line h2
        if (h2 < 0) continue;
        const int d2 = d - N1, r = d2 * 8 + (lane >> 3);
        const int coff = ((lane & 7) ^ ((r >> 1) & 7)) << 4;
        const int voff = (r * NH + h2) * 128 + coff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr_t)(lds + (h2 & 1) * SLAB + W1_SLAB + d2 * 1024), 16, voff, 0, 0, 0);
      }
    }
  };
  auto issue = [&](int hs, unsigned char*) { issue2(hs, hs); };
  // fragment read offsets inside a 32-row block: row = lane & 31, logical chunk = plane * 4 + ks * 2 + half, swizzled
  const int key = (lane >> 1) & 7;
  int foff[2][2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[pl][ks] = tok_l * 128 + (((pl * 4 + ks * 2 + half) ^ key) << 4);

  f32x16 out[OC];
#pragma unroll
  for (int o = 0; o < OC; ++o)
#pragma unroll
    for (int e = 0; e < 16; ++e) out[o][e] = 0.f;
  const float inv1 = p.s1 ? p.s1[g * 2 + 1] : 1.f, inv2 = p.s2 ? p.s2[g * 2 + 1] : 1.f;

  if (TAIL) {
    // ---- proj: eight slabs of 32 output channels x C input channels through the W1 half of the ring (slab o -> buffer o & 1; the
    // Mlp's first slab follows into buffer 0), then residual + LayerNorm2 in registers
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wp + (long)g * C * CB * 128), 0, C * CB * 128, 0x00020000);
    auto issue_proj = [&](int o, unsigned char* buf) {
#pragma unroll
      for (int i = 0; i < N1 / NW; ++i) {
        const int d = i * NW + wave;
        const int cb = d / 4, r = (d % 4) * 8 + (lane >> 3);
        const int coff = ((lane & 7) ^ ((r >> 1) & 7)) << 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr_t)(buf + d * 1024), 16, ((o * 32 + r) * CB + cb) * 128 + coff, 0, 0, 0);
      }
    };
    static_assert(N1 % NW == 0, "proj slab DMA instructions divide over the waves");
    const float invp = p.sp ? p.sp[g * 2 + 1] : 1.f;
    issue_proj(0, lds);
#pragma unroll
    for (int o = 0; o < OC; ++o) {
      __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));            // vmcnt(0)
      __syncthreads();
      const unsigned char* cur = lds + (o & 1) * SLAB;
      if (o + 1 < OC) issue_proj(o + 1, lds + ((o + 1) & 1) * SLAB);
      f32x16 acc, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = acc1[e] = 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; kb += 2) {
        const unsigned char* blk = cur + (kb >> 1) * 4096;
        const u32x4 wl0 = *reinterpret_cast<const u32x4*>(blk + foff[1][0]), wh0 = *reinterpret_cast<const u32x4*>(blk + foff[0][0]);
        const u32x4 wl1 = *reinterpret_cast<const u32x4*>(blk + foff[1][1]), wh1 = *reinterpret_cast<const u32x4*>(blk + foff[0][1]);
        acc = mma(wh0, xl[kb], acc);
        acc1 = mma(wh1, xl[kb + 1], acc1);
        acc = mma(wl0, xh[kb], acc);
        acc1 = mma(wl1, xh[kb + 1], acc1);
        acc = mma(wh0, xh[kb], acc);
        acc1 = mma(wh1, xh[kb + 1], acc1);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] += acc1[e];
      out[o] = acc;
    }
    // the Mlp's slab 0 into buffer 0 (OC even), flying under the residual add and LayerNorm2 below.  Behind a barrier of its own:
    // issued under the last proj slab's reads (no barrier between) the last 32 channels came out wrong on the GPU, with every wait
    // and barrier that orders the two buffers in place -- the arrangement that is correct by measurement is kept
    __syncthreads();
    if (SKEW) issue2(0, -1);
    else issue(0, lds);
    // x <- x + drop * (proj + bias): registers 4 k .. 4 k + 3 of block o are channels 32 o + 8 k + 4 half + 0 .. 3 of token lane & 31
    const float ds = p.drop ? p.drop[(ok ? row : row0) / p.rows_per_drop] : 1.f;
    float* xr_ = p.x_res + row * C;
    float sum = 0.f;
    // the residual row in batches of MRN_TAIL_BATCH 16-byte pieces in flight (one load -> wait -> add -> store per piece is 32 serialized
    // round trips; but every piece in flight is four more live registers next to 128 accumulators + 128 fragment registers)
#ifndef MRN_TAIL_BATCH
#define MRN_TAIL_BATCH 8
#endif
    constexpr int XB = MRN_TAIL_BATCH;
#pragma unroll
    for (int i0 = 0; i0 < OC * 4; i0 += XB) {
      f32x4 xv[XB];
#pragma unroll
      for (int i = 0; i < XB; ++i)
        xv[i] = ok ? *reinterpret_cast<const f32x4*>(xr_ + ((i0 + i) >> 2) * 32 + 8 * ((i0 + i) & 3) + 4 * half) : f32x4{0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < XB; ++i) {
        const int o = (i0 + i) >> 2, k = (i0 + i) & 3;
        const int c = o * 32 + 8 * k + 4 * half;
        const f32x4 bpv = *reinterpret_cast<const f32x4*>(ln_lds + 2 * C + c);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = fmaf(ds, fmaf(out[o][4 * k + j], invp, bpv[j]), xv[i][j]);
          out[o][4 * k + j] = v[j];
          sum += v[j];
        }
        if (ok) *reinterpret_cast<f32x4*>(xr_ + c) = v;
      }
    }
    // LayerNorm2 (the arithmetic of add_layernorm_grouped_kernel: mean, then the centred sum of squares)
    sum += __shfl_xor(sum, 32);
    const float mean = sum / (float)C;
    float q = 0.f;
#pragma unroll
    for (int o = 0; o < OC; ++o)
#pragma unroll
      for (int e = 0; e < 16; ++e) { const float d = out[o][e] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    const float rstd = 1.f / sqrtf(q / (float)C + p.eps);
    // y -> fc1's B operand: k-block kb = 2 o + s holds registers 8 s .. 8 s + 7 of block o (fc1's input channels are packed in that order)
#pragma unroll
    for (int o = 0; o < OC; ++o)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        f16v8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = s_ * 8 + j, c = o * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
          const float v = (out[o][e] - mean) * rstd * ln_lds[c] + ln_lds[C + c];
          _Float16 a, b;
          split_f16(v, a, b);
          vh[j] = a;
          vl[j] = b;
        }
        xh[2 * o + s_] = __builtin_bit_cast(u32x4, vh);
        xl[2 * o + s_] = __builtin_bit_cast(u32x4, vl);
      }
#pragma unroll
    for (int o = 0; o < OC; ++o)
#pragma unroll
      for (int e = 0; e < 16; ++e) out[o][e] = 0.f;
  } else {
    if (SKEW) issue2(0, -1);
    else issue(0, lds);
  }
  // fc1 of hidden block hs from its W1 slab; GELU + split of an fc1 result into fc2's B operand; fc2 of a hidden block from its W2' slab
  // (a 32x32x16 MFMA that accumulates into the result of the one issued right before it waits for that result: ~64 cycles per MFMA
  //  instead of 32 -- in-kernel clocks, 3249 cycles for the 48 chained MFMAs of fc1 at C = 256.  Two accumulators, alternating.)
  auto fc1 = [&](int hs) {
    const unsigned char* cur = lds + (hs & 1) * SLAB;
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; kb += 2) {
      const unsigned char* blk = cur + (kb >> 1) * 4096;      // (kb and kb + 1: the two 16-channel halves of one channel block)
      const u32x4 wl0 = *reinterpret_cast<const u32x4*>(blk + foff[1][0]), wh0 = *reinterpret_cast<const u32x4*>(blk + foff[0][0]);
      const u32x4 wl1 = *reinterpret_cast<const u32x4*>(blk + foff[1][1]), wh1 = *reinterpret_cast<const u32x4*>(blk + foff[0][1]);
      acc0 = mma(wh0, xl[kb], acc0);                // (x_lo * w_hi, x_hi * w_lo, x_hi * w_hi: the order of conv_x3.hip)
      acc1 = mma(wh1, xl[kb + 1], acc1);
      acc0 = mma(wl0, xh[kb], acc0);
      acc1 = mma(wl1, xh[kb + 1], acc1);
      acc0 = mma(wh0, xh[kb], acc0);
      acc1 = mma(wh1, xh[kb + 1], acc1);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] += acc1[e];
    return acc0;
  };
  auto gelu_split = [&](const f32x16& acc, int hs, u32x4 (&hh)[2], u32x4 (&hl)[2]) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      f16v8 vh, vl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = s_ * 8 + j;
        const int unit = (e & 3) + 8 * (e >> 2) + 4 * half;
        const float v = gelu_fast(acc[e] * inv1 + b1_lds[hs * 32 + unit]);
        _Float16 a, b;
        split_f16(v, a, b);
        vh[j] = a;
        vl[j] = b;
      }
      hh[s_] = __builtin_bit_cast(u32x4, vh);
      hl[s_] = __builtin_bit_cast(u32x4, vl);
    }
  };
  auto fc2 = [&](int hs, const u32x4 (&hh)[2], const u32x4 (&hl)[2]) {
    const unsigned char* w2b = lds + (hs & 1) * SLAB + W1_SLAB;
#pragma unroll
    for (int o = 0; o < OC; o += 2) {              // two output blocks at a time: consecutive MFMAs hit different accumulators
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const u32x4 wla = *reinterpret_cast<const u32x4*>(w2b + o * 4096 + foff[1][s_]);
        const u32x4 wha = *reinterpret_cast<const u32x4*>(w2b + o * 4096 + foff[0][s_]);
        const u32x4 wlb = *reinterpret_cast<const u32x4*>(w2b + (o + 1) * 4096 + foff[1][s_]);
        const u32x4 whb = *reinterpret_cast<const u32x4*>(w2b + (o + 1) * 4096 + foff[0][s_]);
        out[o] = mma(wha, hl[s_], out[o]);
        out[o + 1] = mma(whb, hl[s_], out[o + 1]);
        out[o] = mma(wla, hh[s_], out[o]);
        out[o + 1] = mma(wlb, hh[s_], out[o + 1]);
        out[o] = mma(wha, hh[s_], out[o]);
        out[o + 1] = mma(whb, hh[s_], out[o + 1]);
      }
    }
  };
  if (SKEW) {
    // One wave per SIMD (C = 256): the bias + GELU + split of a hidden block (~500 VALU operations) ran with the matrix pipe idle -- as
    // long as fc1 or fc2 (in-kernel clocks).  Skewed schedule: iteration hs computes fc1(hs), then GELU(hs) INTERLEAVED with fc2(hs - 1),
    // which needs W2' one block later than W1: the two halves of the ring are refilled one block apart (W1(hs + 1) and W2'(hs) at the
    // top of iteration hs), same LDS.
    u32x4 hh[2], hl[2], nh[2], nl[2];
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));              // vmcnt(0): W1(0) has landed
    __syncthreads();
    issue2(1, 0);
    gelu_split(fc1(0), 0, hh, hl);
    for (int hs = 1; hs < NH; ++hs) {
      __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));            // vmcnt(0): W1(hs) and W2'(hs - 1) have landed
      __syncthreads();
      issue2(hs + 1 < NH ? hs + 1 : -1, hs);
      const f32x16 acc = fc1(hs);
      __builtin_amdgcn_sched_barrier(0);
      fc2(hs - 1, hh, hl);
      gelu_split(acc, hs, nh, nl);
#pragma unroll
      for (int i = 0; i < 48; ++i) {               // one region: [1 MFMA of fc2(hs - 1), a dozen VALU operations of GELU(hs), a fragment read]
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      hh[0] = nh[0]; hh[1] = nh[1]; hl[0] = nl[0]; hl[1] = nl[1];
    }
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));
    __syncthreads();
    fc2(NH - 1, hh, hl);
  } else
  for (int hs = 0; hs < NH; ++hs) {
    MTICK(tk0);
    // own DMAs of slab hs retired -- spelled out: hipcc drops the vmcnt wait of __syncthreads() here (LDS-DMA is not a load it
    // orders behind the barrier; seen as stale slabs in 1 of ~10^4 workgroups at two workgroups per CU) -- then everyone's have, and
    // everyone is done with the other buffer
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));              // vmcnt(0)
    __syncthreads();
    const unsigned char* cur = lds + (hs & 1) * SLAB;
    MTICK(tk1);
    MADD(0, tk0, tk1);                               // slab wait + barrier
    if (hs + 1 < NH) issue(hs + 1, lds + ((hs + 1) & 1) * SLAB);
    // ---- fc1 for 32 hidden units
    const f32x16 acc = fc1(hs);
    MTICK(tk2);
    MADD(1, tk1, tk2);                               // slab issue + fc1
    // ---- bias, GELU, split: registers 8 s .. 8 s + 7 become the B operand of k half s of the second product
    u32x4 hh[2], hl[2];
    gelu_split(acc, hs, hh, hl);
    MTICK(tk3);
    MADD(2, tk2, tk3);                               // bias + GELU + split
    // ---- fc2 partial sums over these 32 hidden units
    fc2(hs, hh, hl);
    MTICK(tk4);
    MADD(3, tk3, tk4);                               // fc2
  }
#ifdef MRN_MPROBE_TIMING
  if (lane == 0) {
    dbg_acc[4] = __builtin_readcyclecounter() - dbg_t0;
    for (int i = 0; i < 5; ++i) atomicAdd(&g_mlp_dbg[i], (unsigned long long)dbg_acc[i]);
    atomicAdd(&g_mlp_dbg[5], 1ull);
  }
#endif
  // ---- epilogue: registers 4 k .. 4 k + 3 of an output block are channels 32 o + 8 k + 4 half + 0 .. 3 of token lane & 31
  if (ok) {
    float* yr = p.y + row * C;
#pragma unroll
    for (int o = 0; o < OC; ++o)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = o * 32 + 8 * k + 4 * half;
        // (the bias from LDS: as a global load per store it cost `load, s_waitcnt vmcnt(0), store` sixteen to thirty-two times in a
        //  row -- and vmcnt(0) also waits for the PREVIOUS store: the epilogue was a chain of serialized memory round trips)
        const f32x4 b = *reinterpret_cast<const f32x4*>(b2_lds + c);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = out[o][4 * k + j] * inv2 + b[j];
        *reinterpret_cast<f32x4*>(yr + c) = v;
      }
  }
}

template <int C, bool TAIL>
int launch_mlp(const MlpParams& p, int G, hipStream_t st) {
  constexpr size_t ldsz = 2 * ((C / 32) * 32 * 128 + C * 128) + 4 * C * sizeof(float) + (TAIL ? 3 * C * sizeof(float) : 0) + C * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)svtr_mlp_kernel<C, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    attr_set = true;
  }
  hipLaunchKernelGGL((svtr_mlp_kernel<C, TAIL>), dim3((unsigned)(G * p.tiles_per_group)), dim3(C == 256 ? 256 : MRN_MLP_WAVES * 64), ldsz, st, p);
  MRN_LAUNCH_CHECK("svtr_mlp_x3");
  return MRN_OK;
}

}  // namespace

// y[r] = fc2(GELU(fc1(x[r]))) for the rows of G lock-step experts (group = r / rows_per_group), SVTR Mlp (modules/svtr.py:46-67).
//   x_hl  [rows][C/32][128 B]            HL32 input (the LayerNorm output of mrn_add_layernorm_grouped_f32)
//   w1_hl [G][4C][C/32][128 B]           fc1 weights, mrn_pack_weight_hl32 of [4C][1][C]; s1 [G][2] their {s, 1/s}; b1 [G][4C]
//   w2_hl [G][C][4C/32][128 B]           fc2 weights packed from [C][1][4C] with the hidden index of every 32-block permuted:
//                                        position p = 16 s + 8 h + j holds unit (j & 3) + 8 (2 s + (j >> 2)) + 4 h; s2, b2 [G][C]
//   y     [rows][C] fp32
// C = 64, 128 or 256.
MRN_EXPORT int mrn_svtr_mlp_x3_f32(const void* x_hl, const void* w1_hl, const float* s1, const float* b1, const void* w2_hl,
                                   const float* s2, const float* b2, float* y, int64_t rows, int64_t rows_per_group, int G, int C,
                                   void* stream) {
  MRN_CHECK_ARG(x_hl && w1_hl && w2_hl && b1 && b2 && y && G >= 1 && rows_per_group >= 1 && rows <= (int64_t)G * rows_per_group,
                "mrn_svtr_mlp_x3_f32: bad operands");
  MRN_CHECK_ARG(C == 64 || C == 128 || C == 256, "mrn_svtr_mlp_x3_f32: C must be 64, 128 or 256 (got %d)", C);
  MRN_CHECK_ARG((uintptr_t)x_hl % 128 == 0 && (uintptr_t)w1_hl % 128 == 0 && (uintptr_t)w2_hl % 128 == 0 && (uintptr_t)y % 16 == 0,
                "mrn_svtr_mlp_x3_f32: operands must be 128-byte (HL32) / 16-byte (y) aligned");
  if (rows == 0) return MRN_OK;
  MlpParams p;
  p.x_hl = (const unsigned char*)x_hl; p.w1 = (const unsigned char*)w1_hl; p.w2 = (const unsigned char*)w2_hl;
  p.b1 = b1; p.b2 = b2; p.s1 = s1; p.s2 = s2; p.y = y;
  p.rows = rows; p.rows_per_group = rows_per_group;
  const int tile_rows = C == 256 ? 128 : MRN_MLP_WAVES * 32;
  p.tiles_per_group = (int)((rows_per_group + tile_rows - 1) / tile_rows);
  p.wp = nullptr; p.sp = p.bp = p.drop = p.gamma = p.beta = nullptr; p.x_res = nullptr; p.rows_per_drop = 1; p.eps = 0.f;
  if (C == 256) return launch_mlp<256, false>(p, G, (hipStream_t)stream);
  return C == 64 ? launch_mlp<64, false>(p, G, (hipStream_t)stream) : launch_mlp<128, false>(p, G, (hipStream_t)stream);
}

// The second half of an SVTR stage-3 mixing block (C = 256) for the rows of G lock-step experts, one launch (modules/svtr.py:196-204):
//   x_res[r] += drop[r / rows_per_drop] * (ctx[r] Wproj^T + bproj)        (drop NULL: 1)
//   branch[r] = fc2(GELU(fc1(LayerNorm(x_res[r]; gamma, beta, eps))))
// ctx_hl [rows][8][128 B]: the attention context (mrn_svtr_attention_block_x3_f32); wproj_hl [G][256][8][128 B] from mrn_pack_weight_hl32,
// sproj / bproj its scale pair and bias; gamma, beta [G][256]; w1_hl as for mrn_svtr_mlp_x3_f32 but with fc1's INPUT channels permuted
// inside every 32-block by the same permutation as fc2's hidden units (position p = 16 s + 8 h + j holds channel
// (j & 3) + 8 (2 s + (j >> 2)) + 4 h); w2_hl, s1, b1, s2, b2 as there; branch [rows][256] fp32.
MRN_EXPORT int mrn_svtr_tail_x3_f32(const void* ctx_hl, float* x_res, const void* wproj_hl, const float* sproj, const float* bproj,
                                    const float* drop, int64_t rows_per_drop, const float* gamma, const float* beta, float eps,
                                    const void* w1_hl, const float* s1, const float* b1, const void* w2_hl, const float* s2, const float* b2,
                                    float* branch, int64_t rows, int64_t rows_per_group, int G, int C, void* stream) {
  MRN_CHECK_ARG(ctx_hl && x_res && wproj_hl && bproj && gamma && beta && w1_hl && w2_hl && b1 && b2 && branch && G >= 1 &&
                    rows_per_group >= 1 && rows <= (int64_t)G * rows_per_group && rows_per_drop >= 1,
                "mrn_svtr_tail_x3_f32: bad operands");
  MRN_CHECK_ARG(C == 256, "mrn_svtr_tail_x3_f32: C must be 256 (got %d)", C);
  MRN_CHECK_ARG((uintptr_t)ctx_hl % 128 == 0 && (uintptr_t)wproj_hl % 128 == 0 && (uintptr_t)w1_hl % 128 == 0 && (uintptr_t)w2_hl % 128 == 0 &&
                    (uintptr_t)branch % 16 == 0 && (uintptr_t)x_res % 16 == 0,
                "mrn_svtr_tail_x3_f32: operands must be 128-byte (HL32) / 16-byte (fp32) aligned");
  if (rows == 0) return MRN_OK;
  MlpParams p;
  p.x_hl = (const unsigned char*)ctx_hl; p.w1 = (const unsigned char*)w1_hl; p.w2 = (const unsigned char*)w2_hl;
  p.b1 = b1; p.b2 = b2; p.s1 = s1; p.s2 = s2; p.y = branch;
  p.rows = rows; p.rows_per_group = rows_per_group;
  p.tiles_per_group = (int)((rows_per_group + 127) / 128);
  p.wp = (const unsigned char*)wproj_hl; p.sp = sproj; p.bp = bproj; p.drop = drop; p.gamma = gamma; p.beta = beta; p.x_res = x_res;
  p.rows_per_drop = rows_per_drop; p.eps = eps;
  return launch_mlp<256, true>(p, G, (hipStream_t)stream);
}
