// Fused attention half of an SVTR mixing block for G lock-step frozen experts (reference modules/svtr.py:90-152 Attention, :196-201 the
// first line of Block.forward):
//     t     = x + drop_prev * pending                          (the previous block's Mlp branch, folded in here)
//     x_out = t + drop1 * proj(softmax(q k^T * scale + mask) v),   q | k | v = qkv(LayerNorm1(t))
//     y2    = LayerNorm2(x_out)  as the HL32 operand of the Mlp kernel (svtr_mlp.hip)
// in ONE kernel: neither the LayerNorm output, nor q | k | v, nor the attention context, nor the proj result touch HBM.  The unfused
// chain (add_layernorm_grouped -> grouped Linear -> svtr_attention_x3 -> grouped Linear -> add_layernorm_grouped) moves 13 x the
// residual stream's bytes per block; this kernel reads x (+ pending) twice (the second time from L2) and writes x_out and y2.
//
// Arithmetic is the unfused chain's: LayerNorm in fp32, every product as split-fp16 x3 (lo*hi + hi*lo + hi*hi, fp32 accumulate) on
// v_mfma_f32_32x32x16_f16, weights prescaled by a per-tensor power of two, q / k / v split as 64 * value (q also carries scale * log2 e),
// base-2 online softmax with a 2^12 bias on the probabilities (attention.hip).
//
// Formulation (the chained-MFMA form of svtr_mlp.hip, one step further).  A wave owns 32 tokens of one image for the whole kernel;
// their LayerNorm-ed channels sit in registers as MFMA fragments (lane = token, 8 consecutive channels per k-block half) -- which is
// the B operand of  W . y^T  AND the A operand of  y . W^T.  Per head (head dim 32):
//   K^T[d][token] = Wk . y^T       lane = token, registers = d     -> + bias, split: the A operand rows of S^T = K Q^T, written to LDS
//   V[token][d]   = y . Wv^T       lane = d,     registers = token -> + bias, split: the A operand rows of O^T = V^T P^T, written to LDS
//   Q^T[d][token] = Wq . y^T       lane = token, registers = d     -> stays in registers as the B operand of S^T
//   S^T[key][query] = K Q^T over the key tiles of the image (fragments from LDS), online softmax per lane (= per query), P^T straight
//                     from the score registers into the B operand of O^T[d][query] += V^T P^T
//   branch^T[c][query] += Wproj[:, head] . O^T   (O^T normalised and split in place; reduction index permuted inside the head's 32
//                     columns exactly as the hidden units of svtr_mlp.hip's fc2)
// so the only cross-lane traffic is K / V through LDS (other waves' keys).  The four weight slabs of a head (Wk, Wv, Wq rows: 32 x C;
// Wproj columns: C x 32; C * 128 bytes each in HL32) stream through a two-deep LDS ring by direct-to-LDS DMA, one barrier per slab.
// The epilogue adds the residual, writes x_out, and LayerNorm2 runs on the accumulator registers.
// A workgroup is NT waves = NT * 32 queries of one image (IMG images when they are short).  Images of more than 256 tokens are cut into
// CHUNKS = 2 key chunks of NT * 32: two workgroups share the image, each owns one chunk's queries and walks both chunks' keys -- K and V
// of the other chunk are recomputed from x (LayerNorm1 again, 24 of ~250 MFMAs per tile and head) so that only one chunk's K / V
// (64 KiB) sits in LDS at a time.
// C = 64 with up to 512 tokens (SVTR stage 1: 8 x 25 at 32 x 100 crops, 8 x 64 at 32 x 256) and C = 128 with up to 256 tokens (stage 2);
// C = 256 (stage 3: the token fragments and the accumulators alone are 256 registers) uses the unfused chain.
#include "common.hpp"

namespace {

typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr float LOG2E_F = 1.4426950408889634f;
constexpr float PBIAS = 12.f;       // probabilities carry 2^12 (cancelled by 1 / l): small ones stay out of fp16's subnormal range
constexpr float OPSCALE = 64.f;     // q, k, v are split as 64 * x (attention.hip)
constexpr float SINV = 1.f / (OPSCALE * OPSCALE);

struct MixerParams {
  const float* x;                 // [imgs][N][C] residual stream
  const float* pend;              // [imgs][N][C] branch of the previous block's Mlp, or null
  const float* drop_prev;         // [imgs] its DropPath scale per sample, or null (= 1)
  const float* g1;                // [G][C] LayerNorm1
  const float* b1;
  const unsigned char* wqkv;      // [G][3C][C/32][128 B] HL32
  const float* sqkv;              // [G][2] {s, 1/s}
  const float* bqkv;              // [G][3C] or null
  const unsigned* mask_bits;      // [N][ceil(N/32)] visibility bits of the local mixer, or null
  const unsigned char* wproj;     // [G][C][C/32][128 B] HL32, input channel permuted inside every 32-block (head)
  const float* sproj;             // [G][2]
  const float* bproj;             // [G][C]
  const float* drop1;             // [imgs] or null
  const float* g2;                // [G][C] LayerNorm2
  const float* b2;
  float* x_out;                   // [imgs][N][C]
  unsigned char* y_hl;            // [imgs * N][C/32][128 B]
  int imgs, imgs_per_group, N;
  int perm_h, perm_w;             // > 0: the kernel walks the tokens of an image COLUMN-major (position p = col * perm_h + row is token
                                  // row * perm_w + col; N == perm_h * perm_w) and mask_bits is indexed by positions: a key tile of 32
                                  // positions is then a block of 32 / perm_h whole columns, and the local mixer's window (all rows, 11
                                  // columns) touches 3 tiles of 8 (stage 2) / 5 of 16 (stage 1) instead of every row's half-rows
  float scale, eps1, eps2;
};

__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16v8*>(&a), *reinterpret_cast<const f16v8*>(&b), c, 0, 0, 0);
}

#ifdef MRN_XPROBE_TIMING
// timing probe (never in the product build; bash tools/build_probe.sh MRN_XPROBE_TIMING svtr_mixer.hip; tools/probe/xtiming.py): shader-clock
// totals of a wave's phases, summed over all waves: [0] prologue (x + pending, LayerNorm1), [1] K / V projections, [2] Q projection,
// [3] key-tile loop, [4] proj, [5] epilogue, [6] waves, [7] whole kernel
__device__ unsigned long long g_mixer_dbg[8];
extern "C" __attribute__((visibility("default"))) int mrn_mixer_dbg_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mixer_dbg), sizeof(g_mixer_dbg)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mixer_dbg), z, sizeof(z));
  }
  return 0;
}
#define XTICK(var) const long var = __builtin_readcyclecounter()
#define XADD(slot, a, b) xdbg[slot] += (b) - (a)
#else
#define XTICK(var)
#define XADD(slot, a, b)
#endif

// ATTN: attention only -- LayerNorm1 -> qkv -> attention, the context leaves as the HL32 operand of an unfused proj Linear (p.y_hl) and,
// when a pending branch was folded in, t = x + drop_prev * pending as the residual stream (p.x_out).  For C = 256 (SVTR stage 3), where
// the proj accumulators of the full form do not fit the register file next to the token fragments.
template <int C, int NT, int IMG, int CHUNKS, int RING, bool ATTN = false>
__global__ __launch_bounds__(NT * IMG * 64) void svtr_mixer_kernel(const MixerParams p) {
  static_assert(CHUNKS == 1 || (CHUNKS == 2 && IMG == 1), "key chunks: one image per workgroup");
  static_assert(!ATTN || CHUNKS == 1, "attention-only form: one key chunk");
  constexpr int NW = NT * IMG, CB = C / 32, KB = C / 16, HEADS = C / 32, OC = C / 32;
  constexpr int STEPS = (ATTN ? 1 : 2) + 2 * CHUNKS; // weight slabs per head: (Wk, Wv) per key chunk, Wq, Wproj (not in the attention-only form)
  constexpr int SLAB = C * 128;                      // one weight slab: 32 rows x C channels, or C rows x 32 channels
  constexpr int NDMA = SLAB / 1024;                  // 1-KiB DMA instructions per slab
  constexpr int DMA_ROUNDS = (NDMA + NW - 1) / NW;
  // RING slabs in the LDS ring: RING - 1 in flight ahead of the one being consumed
  constexpr int TOTAL = STEPS * HEADS;
  constexpr int TILE = 32 * 128;                     // one K (or V^T) tile: 32 lines of [hi 32 | lo 32]
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
  unsigned char* lds_k = lds;                        // [IMG * NT] tiles, line = key
  unsigned char* lds_v = lds + IMG * NT * TILE;      // [IMG * NT] tiles, line = d, slots = the tile's keys in score-register order
  unsigned char* slab0 = lds + 2 * IMG * NT * TILE;  // RING slabs
  // the parameter block sits behind BOTH the K / V tiles + slab ring and the epilogue's row-staging tiles (which reuse the LDS from offset 0)
  constexpr int RING_END = 2 * IMG * NT * TILE + RING * SLAB, STAGE_END = ATTN ? 0 : NT * IMG * 32 * (C * 4 + 16);
  constexpr int PARAM_OFF = RING_END > STAGE_END ? RING_END : STAGE_END;
  float* bias_lds = reinterpret_cast<float*>(lds + PARAM_OFF);        // [3C] qkv bias, then [C] each: LayerNorm1 gamma, beta, LayerNorm2 gamma, beta, proj bias
  float* ln_lds = bias_lds + 3 * C;                                  // (per-channel vectors read by every lane: from LDS, not 16 dependent global round trips)

#ifdef MRN_XPROBE_TIMING
  long xdbg[6] = {0, 0, 0, 0, 0, 0};
#endif
  XTICK(x_t0);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wi = wave / NT, ti = wave % NT;
  const int qc = CHUNKS == 1 ? 0 : blockIdx.x % CHUNKS;      // the key chunk whose tokens are this workgroup's queries
  const int img0 = (blockIdx.x / CHUNKS) * IMG;
  const int g = img0 / p.imgs_per_group;
  const int img = img0 + wi;
  // position -> token: everything per token (LayerNorm, projections, residual) is order-free; only the row addresses and the mask see it
  auto token_of = [&](int pos) -> int {
    if (p.perm_h <= 0) return pos;
    const int col = pos / p.perm_h;
    return (pos - col * p.perm_h) * p.perm_w + col;
  };
  const int tok = (qc * NT + ti) * 32 + l31;                  // POSITION of this lane's query
  const bool ok = tok < p.N && img < p.imgs;
  const long row = (long)(img < p.imgs ? img : img0) * p.N + (tok < p.N ? token_of(tok) : 0);

  for (int i = t; i < 3 * C; i += NW * 64) bias_lds[i] = p.bqkv ? p.bqkv[(long)g * 3 * C + i] : 0.f;
  for (int i = t; i < C; i += NW * 64) {
    ln_lds[i] = p.g1[(long)g * C + i];
    ln_lds[C + i] = p.b1[(long)g * C + i];
    if (!ATTN) {
      ln_lds[2 * C + i] = p.g2[(long)g * C + i];
      ln_lds[3 * C + i] = p.b2[(long)g * C + i];
      ln_lds[4 * C + i] = p.bproj[(long)g * C + i];
    }
  }
  bool params_ready = false;          // (the first reader of the parameter block waits for it: load_ln1 below)

  // ---- weight slabs, STEPS per head in the order Wk, Wv (own key chunk), Wq, [Wk, Wv (other chunk),] Wproj.  DMA instruction d moves 8
  // lines of 128 B: lane -> line 8 d + lane / 8, chunk lane & 7, source chunk XOR-swizzled with (line >> 1) & 7 (conflict-free
  // ds_read_b128 fragments)
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wqkv + (long)g * 3 * C * CB * 128), 0, 3 * C * CB * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wproj + (long)g * C * CB * 128), 0, C * CB * 128, 0x00020000);
  auto issue = [&](int step, unsigned char* buf) {
    const int h = step / STEPS, sh = step % STEPS;
    const int ty = (!ATTN && sh == STEPS - 1) ? 3 : sh == 2 ? 2 : sh < 2 ? sh : sh - 3;      // 0: Wk, 1: Wv, 2: Wq, 3: Wproj
#pragma unroll
    for (int i = 0; i < DMA_ROUNDS; ++i) {
      const int d = i * NW + wave;
      if (d < NDMA) {
        if (ty < 3) {                                // LDS [cb][32 rows]; source row = row0 + r, line cb
          const int row0 = (ty == 0 ? C : ty == 1 ? 2 * C : 0) + h * 32;
          const int cb = d / 4, r = (d % 4) * 8 + (lane >> 3);
          const int coff = ((lane & 7) ^ ((r >> 1) & 7)) << 4;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)(buf + d * 1024), 16, ((row0 + r) * CB + cb) * 128 + coff, 0, 0, 0);
        } else {                                     // LDS [C rows]; source row r, line h
          const int r = d * 8 + (lane >> 3);
          const int coff = ((lane & 7) ^ ((r >> 1) & 7)) << 4;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr_t)(buf + d * 1024), 16, (r * CB + h) * 128 + coff, 0, 0, 0);
        }
      }
    }
  };
#pragma unroll
  for (int i = 0; i < RING - 1; ++i)
    if (i < TOTAL) issue(i, slab0 + i * SLAB);
  // DMA instructions of one slab that THIS wave issues (they divide unevenly when NDMA % NW != 0)
  const int my_dma = (NDMA % NW == 0 || wave < NDMA % NW) ? DMA_ROUNDS : DMA_ROUNDS - 1;

  // ---- t = x + drop_prev * pending, LayerNorm1, split: MFMA fragments of one token per lane pair, k-block kb = channels
  // 16 kb + 8 half .. + 7 of token lane & 31
  const float ds = (p.pend && p.drop_prev) ? p.drop_prev[img < p.imgs ? img : img0] : 1.f;
  auto load_ln1 = [&](long r, bool valid, u32x4* fh, u32x4* fl, bool store_t = false) {
    float v[KB][8];
    const float* xr = p.x + r * C;
    const float* pr = p.pend ? p.pend + r * C : nullptr;
    // every piece of the token's row in flight at once (2 KB loads of 16 B, then the pending branch's): issued four at a time behind a
    // full s_waitcnt each -- what the compiler made of the interleaved form -- the prologue was 16 serialized memory round trips, a
    // quarter of the kernel's time with ONE workgroup per CU and nothing to overlap them with (in-kernel clocks, tools/probe/xtiming.py)
    {
      f32x4 xa[KB], xb[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int c = 16 * kb + 8 * half;
        xa[kb] = *reinterpret_cast<const f32x4*>(xr + c);
        xb[kb] = *reinterpret_cast<const f32x4*>(xr + c + 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[kb][j] = xa[kb][j]; v[kb][4 + j] = xb[kb][j]; }
    }
    if (pr) {
      constexpr int PB = KB > 8 ? 8 : KB;          // (C = 256: two batches of 64 registers -- one of 128 next to the row itself spills)
#pragma unroll
      for (int k0 = 0; k0 < KB; k0 += PB) {
        f32x4 pa[PB], pb[PB];
#pragma unroll
        for (int kb = 0; kb < PB; ++kb) {
          const int c = 16 * (k0 + kb) + 8 * half;
          pa[kb] = *reinterpret_cast<const f32x4*>(pr + c);
          pb[kb] = *reinterpret_cast<const f32x4*>(pr + c + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < PB; ++kb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[k0 + kb][j] = fmaf(ds, pa[kb][j], v[k0 + kb][j]);
            v[k0 + kb][4 + j] = fmaf(ds, pb[kb][j], v[k0 + kb][4 + j]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int c = 16 * kb + 8 * half;
      if (store_t && valid) {                      // attention-only form: the folded residual stream t = x + drop_prev * pending
        *reinterpret_cast<f32x4*>(p.x_out + r * C + c) = f32x4{v[kb][0], v[kb][1], v[kb][2], v[kb][3]};
        *reinterpret_cast<f32x4*>(p.x_out + r * C + c + 4) = f32x4{v[kb][4], v[kb][5], v[kb][6], v[kb][7]};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[kb][j];
    }
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[kb][j] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    const float rstd = 1.f / sqrtf(q / (float)C + p.eps1);
    if (!params_ready) {                           // (first call: the parameter block was written by other lanes)
      __syncthreads();
      params_ready = true;
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int c = 16 * kb + 8 * half;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(ln_lds + c), gb = *reinterpret_cast<const f32x4*>(ln_lds + c + 4);
      const f32x4 ba = *reinterpret_cast<const f32x4*>(ln_lds + C + c), bb = *reinterpret_cast<const f32x4*>(ln_lds + C + c + 4);
      f16v8 vh, vl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float y = valid ? (v[kb][j] - mean) * rstd * (j < 4 ? ga[j] : gb[j - 4]) + (j < 4 ? ba[j] : bb[j - 4]) : 0.f;
        _Float16 a, b;
        split_f16(y, a, b);
        vh[j] = a;
        vl[j] = b;
      }
      fh[kb] = __builtin_bit_cast(u32x4, vh);
      fl[kb] = __builtin_bit_cast(u32x4, vl);
    }
  };
  u32x4 xh[KB], xl[KB];
  load_ln1(row, ok, xh, xl, ATTN && p.pend != nullptr && p.x_out != nullptr);
#ifdef MRN_XPROBE_TIMING
  asm volatile("" :: "v"(xh[KB - 1][3]), "v"(xl[KB - 1][3]));
#endif
  XTICK(x_t1);
  XADD(0, x_t0, x_t1);

  // fragment offsets inside a 32-line tile / slab block: line = lane & 31, logical chunk = plane * 4 + ks * 2 + half, swizzled
  const int key = (l31 >> 1) & 7;
  int foff[2][2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[pl][ks] = l31 * 128 + (((pl * 4 + ks * 2 + half) ^ key) << 4);

  f32x16 out[OC];
#pragma unroll
  for (int o = 0; o < OC; ++o)
#pragma unroll
    for (int e = 0; e < 16; ++e) out[o][e] = 0.f;
  const float invq = p.sqkv ? p.sqkv[g * 2 + 1] : 1.f, invp = p.sproj ? p.sproj[g * 2 + 1] : 1.f;
  unsigned char* my_k = lds_k + (wi * NT + ti) * TILE;
  unsigned char* my_v = lds_v + (wi * NT + ti) * TILE;
  const unsigned* brow = p.mask_bits ? p.mask_bits + (long)(tok < p.N ? tok : 0) * ((p.N + 31) / 32) : nullptr;

  // every slab step starts the same way: own DMAs of this slab retired -- a counted wait: the younger slabs of the ring stay in
  // flight (spelled out: hipcc does not order LDS-DMA behind __syncthreads()) -- then everyone's have and everyone is done with the
  // slab consumed one step ago, whose buffer takes the slab RING - 1 steps ahead
  auto next_slab = [&](int step) -> const unsigned char* {
    const int younger = min(RING - 2, TOTAL - 1 - step) * my_dma;      // this wave's DMA instructions issued after slab `step`'s
    if (RING > 2 && younger >= 4) __builtin_amdgcn_s_waitcnt((4 & 15) | (7 << 4) | (15 << 8));
    else if (younger == 3) __builtin_amdgcn_s_waitcnt(3 | (7 << 4) | (15 << 8));
    else if (younger == 2) __builtin_amdgcn_s_waitcnt(2 | (7 << 4) | (15 << 8));
    else if (younger == 1) __builtin_amdgcn_s_waitcnt(1 | (7 << 4) | (15 << 8));
    else __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));
    __syncthreads();
    if (step + RING - 1 < TOTAL) issue(step + RING - 1, slab0 + ((step + RING - 1) % RING) * SLAB);
    return slab0 + (step % RING) * SLAB;
  };
  // W . y^T (weights as the A operand): lane = token, register e = row (e & 3) + 8 (e >> 2) + 4 half of the slab
  auto w_times_y = [&](const unsigned char* cur, const u32x4* fh, const u32x4* fl) -> f32x16 {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const unsigned char* blk = cur + (kb >> 1) * 4096;
      const u32x4 wl = *reinterpret_cast<const u32x4*>(blk + foff[1][kb & 1]);
      const u32x4 wh = *reinterpret_cast<const u32x4*>(blk + foff[0][kb & 1]);
      acc = mma(wh, fl[kb], acc);
      acc = mma(wl, fh[kb], acc);
      acc = mma(wh, fh[kb], acc);
    }
    return acc;
  };
  // K^T and V of the 32 tokens whose fragments are (fh, fl) -> this wave's K / V tiles (two slab steps)
  auto kv_tiles = [&](int step, int h, const u32x4* fh, const u32x4* fl) {
    {   // K^T: lane = token, registers = d -> LDS lines [key][hi d | lo d], d in register order (the order Q^T's registers use)
      const unsigned char* cur = next_slab(step);
      const f32x16 acc = w_times_y(cur, fh, fl);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f16v8 kh, kl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = 8 * m + j;
          const int d = (e & 3) + 8 * (e >> 2) + 4 * half;
          const float v = (acc[e] * invq + bias_lds[C + h * 32 + d]) * OPSCALE;
          _Float16 a, b;
          split_f16(v, a, b);
          kh[j] = a;
          kl[j] = b;
        }
        *reinterpret_cast<f16v8*>(my_k + foff[0][m]) = kh;
        *reinterpret_cast<f16v8*>(my_k + foff[1][m]) = kl;
      }
    }
    {   // V (y as the A operand): lane = d, registers = tokens -> LDS lines [d][hi keys | lo keys]
      const unsigned char* cur = next_slab(step + 1);
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const unsigned char* blk = cur + (kb >> 1) * 4096;
        const u32x4 wl = *reinterpret_cast<const u32x4*>(blk + foff[1][kb & 1]);
        const u32x4 wh = *reinterpret_cast<const u32x4*>(blk + foff[0][kb & 1]);
        acc = mma(fl[kb], wh, acc);
        acc = mma(fh[kb], wl, acc);
        acc = mma(fh[kb], wh, acc);
      }
      const float bv = bias_lds[2 * C + h * 32 + l31];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f16v8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (acc[8 * m + j] * invq + bv) * OPSCALE;
          _Float16 a, b;
          split_f16(v, a, b);
          vh[j] = a;
          vl[j] = b;
        }
        *reinterpret_cast<f16v8*>(my_v + foff[0][m]) = vh;
        *reinterpret_cast<f16v8*>(my_v + foff[1][m]) = vl;
      }
    }
  };

  for (int h = 0; h < HEADS; ++h) {
    u32x4 qh[2], ql[2];
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    float l_run = 0.f, m_run = -INFINITY;
    // the attention of this wave's 32 queries over the key chunk `kc` whose K / V tiles are in LDS
    auto attend = [&](int kc) {
      for (int kt = 0; kt < NT; ++kt) {
        const int k0 = (kc * NT + kt) * 32;
        if (k0 >= p.N) break;
        const unsigned bword = brow ? brow[kc * NT + kt] : 0xffffffffu;
        if (__ballot(bword != 0u) == 0) continue;               // no query of this wave sees a key of this tile (local window)
        const unsigned bw = bword >> (4 * half);
        const unsigned char* kt_ = lds_k + (wi * NT + kt) * TILE;
        const unsigned char* vt_ = lds_v + (wi * NT + kt) * TILE;
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        {
          u32x4 kh[2], kl[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            kh[m] = *reinterpret_cast<const u32x4*>(kt_ + foff[0][m]);
            kl[m] = *reinterpret_cast<const u32x4*>(kt_ + foff[1][m]);
          }
#ifdef MRN_XPROBE_NO_ATTN_MFMA
          // what-if probe: the softmax VALU work without the twelve attention MFMAs of a key tile (scores = a fragment's bits)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[e] = __builtin_bit_cast(float, (kh[e >> 3][e & 3] ^ qh[0][e & 3]) & 0x3fffffffu);
#else
#pragma unroll
          for (int m = 0; m < 2; ++m) s = mma(kl[m], qh[m], s);
#pragma unroll
          for (int m = 0; m < 2; ++m) s = mma(kh[m], ql[m], s);
#pragma unroll
          for (int m = 0; m < 2; ++m) s = mma(kh[m], qh[m], s);
#endif
        }
#ifdef MRN_XPROBE_NO_SOFTMAX
        // what-if probe (never in the product build; bash tools/build_probe.sh MRN_XPROBE_NO_SOFTMAX svtr_mixer.hip): the score registers go
        // straight into the P operand -- no mask, maximum, exponential, split or rescale: what the key-tile loop costs WITHOUT its VALU work
        {
          u32x4 ph_[2], pl_[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            ph_[m] = u32x4{__builtin_bit_cast(unsigned, s[8 * m]), __builtin_bit_cast(unsigned, s[8 * m + 1]), __builtin_bit_cast(unsigned, s[8 * m + 2]), __builtin_bit_cast(unsigned, s[8 * m + 3])};
            pl_[m] = u32x4{__builtin_bit_cast(unsigned, s[8 * m + 4]), __builtin_bit_cast(unsigned, s[8 * m + 5]), __builtin_bit_cast(unsigned, s[8 * m + 6]), __builtin_bit_cast(unsigned, s[8 * m + 7])};
          }
          l_run = 1.f;
          u32x4 vh[2], vl[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            vh[m] = *reinterpret_cast<const u32x4*>(vt_ + foff[0][m]);
            vl[m] = *reinterpret_cast<const u32x4*>(vt_ + foff[1][m]);
          }
#pragma unroll
          for (int m = 0; m < 2; ++m) o = mma(vl[m], ph_[m], o);
#pragma unroll
          for (int m = 0; m < 2; ++m) o = mma(vh[m], pl_[m], o);
#pragma unroll
          for (int m = 0; m < 2; ++m) o = mma(vh[m], ph_[m], o);
          continue;
        }
#endif
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
        const float m_new = fmaxf(m_run, mx * SINV);             // (the scores carry OPSCALE^2: folded into the exponent's FMA)
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float corr = __builtin_amdgcn_exp2f(m_run - m_safe);
        const float mb = m_safe - PBIAS;
        float psum = 0.f;
        u32x4 ph[2], pl[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          f16v8 vh, vl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float pe = __builtin_amdgcn_exp2f(fmaf(s[8 * m + j], SINV, -mb));
            psum += pe;
            _Float16 a, b;
            split_f16(pe, a, b);
            vh[j] = a;
            vl[j] = b;
          }
          ph[m] = __builtin_bit_cast(u32x4, vh);
          pl[m] = __builtin_bit_cast(u32x4, vl);
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * corr + psum;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] *= corr;
        {
          u32x4 vh[2], vl[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            vh[m] = *reinterpret_cast<const u32x4*>(vt_ + foff[0][m]);
            vl[m] = *reinterpret_cast<const u32x4*>(vt_ + foff[1][m]);
          }
#ifdef MRN_XPROBE_NO_ATTN_MFMA
#pragma unroll
          for (int e = 0; e < 16; ++e) o[e] += __builtin_bit_cast(float, (vh[e >> 3][e & 3] ^ ph[e >> 3][e & 3] ^ pl[0][e & 3] ^ vl[1][e & 3]) & 0x3fffffffu);
#else
#pragma unroll
          for (int m = 0; m < 2; ++m) o = mma(vl[m], ph[m], o);
#pragma unroll
          for (int m = 0; m < 2; ++m) o = mma(vh[m], pl[m], o);
#pragma unroll
          for (int m = 0; m < 2; ++m) o = mma(vh[m], ph[m], o);
#endif
        }
      }
    };

    const int step0 = h * STEPS;
    XTICK(x_h0);
    kv_tiles(step0, h, xh, xl);                                  // own chunk's K / V from this wave's own tokens
    XTICK(x_h1);
    XADD(1, x_h0, x_h1);
    {
      // ---- Q^T of this wave's tokens (registers); the slab's barrier also publishes every wave's K and V lines
      const unsigned char* cur = next_slab(step0 + 2);
      const f32x16 acc = w_times_y(cur, xh, xl);
      const float sc = p.scale * LOG2E_F * OPSCALE;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f16v8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = 8 * m + j;
          const int d = (e & 3) + 8 * (e >> 2) + 4 * half;
          const float v = (acc[e] * invq + bias_lds[h * 32 + d]) * sc;
          _Float16 a, b;
          split_f16(v, a, b);
          vh[j] = a;
          vl[j] = b;
        }
        qh[m] = __builtin_bit_cast(u32x4, vh);
        ql[m] = __builtin_bit_cast(u32x4, vl);
      }
      XTICK(x_h2);
      XADD(2, x_h1, x_h2);
      attend(qc);
#ifdef MRN_XPROBE_TIMING
      asm volatile("" :: "v"(o[15]));
#endif
      XTICK(x_h3);
      XADD(3, x_h2, x_h3);
    }
    if constexpr (CHUNKS == 2) {
      // ---- the other chunk: its tokens' LayerNorm1 fragments again (x from L2), K / V into the same tiles -- the first slab barrier
      // inside kv_tiles is the point where every wave is done with the own chunk's tiles
      const int tok2 = ((1 - qc) * NT + ti) * 32 + l31;
      u32x4 th[KB], tl[KB];
      load_ln1((long)(img < p.imgs ? img : img0) * p.N + (tok2 < p.N ? token_of(tok2) : 0), tok2 < p.N && img < p.imgs, th, tl);
      kv_tiles(step0 + 3, h, th, tl);
    }
    if constexpr (ATTN) {
      // ---- attention-only form: the head's context (normalised, split) leaves as line (row, head) of the proj Linear's HL32 operand;
      // the next head's first slab barrier is where every wave is done with this head's K / V tiles
      if (ok) {
        const float inv = l_run > 0.f ? 1.f / (l_run * OPSCALE) : 0.f;
        unsigned char* line = p.y_hl + (row * (long)HEADS + h) * 128;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f16v4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            _Float16 a, b;
            split_f16(o[4 * k + j] * inv, a, b);
            hi[j] = a;
            lo[j] = b;
          }
          *reinterpret_cast<f16v4*>(line + (8 * k + 4 * half) * 2) = hi;
          *reinterpret_cast<f16v4*>(line + 64 + (8 * k + 4 * half) * 2) = lo;
        }
      }
    } else
    // ---- branch^T += Wproj[:, head h] . O^T: the context registers (normalised, split in place) are the B operand
    {
      XTICK(x_p0);
      const unsigned char* cur = next_slab(step0 + STEPS - 1);  // (its barrier publishes the other chunk's tiles / frees the own chunk's)
      if constexpr (CHUNKS == 2) attend(1 - qc);
      const float inv = l_run > 0.f ? 1.f / (l_run * OPSCALE) : 0.f;
      u32x4 oh[2], ol[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f16v8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          _Float16 a, b;
          split_f16(o[8 * m + j] * inv, a, b);
          vh[j] = a;
          vl[j] = b;
        }
        oh[m] = __builtin_bit_cast(u32x4, vh);
        ol[m] = __builtin_bit_cast(u32x4, vl);
      }
#pragma unroll
      for (int oc = 0; oc < OC; ++oc) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const u32x4 wl = *reinterpret_cast<const u32x4*>(cur + oc * 4096 + foff[1][m]);
          const u32x4 wh = *reinterpret_cast<const u32x4*>(cur + oc * 4096 + foff[0][m]);
          out[oc] = mma(wh, ol[m], out[oc]);
          out[oc] = mma(wl, oh[m], out[oc]);
          out[oc] = mma(wh, oh[m], out[oc]);
        }
      }
#ifdef MRN_XPROBE_TIMING
      asm volatile("" :: "v"(out[OC - 1][15]));
      XTICK(x_p1);
      XADD(4, x_p0, x_p1);
#endif
    }
  }
  XTICK(x_e0);

  // ---- epilogue: registers 4 k .. 4 k + 3 of block oc are channels 32 oc + 8 k + 4 half + 0 .. 3 of token lane & 31:
  // x_out = t + drop1 * (branch + bias), LayerNorm2 on the registers, HL32 lines for the Mlp kernel.  A lane holds 16-byte pieces of
  // one token's row: stored directly, every instruction would scatter 64 pieces over 64 rows (measured: a third of the kernel's
  // time).  The wave's 32 rows go through a private LDS tile instead (the K / V tiles and the slab ring are free now) and leave as
  // whole rows, 1 KiB contiguous per store instruction.
  if constexpr (!ATTN) {
    constexpr int ROWB = C * 4 + 16;                 // padded row: conflict-free 16-byte column writes
    constexpr int LPR = C * 4 / 16, RPI = 64 / LPR;  // lanes per row, rows per store instruction
    __syncthreads();                                 // every wave is done with the K / V tiles and the last slab
    unsigned char* stg = lds + wave * (32 * ROWB);
    const int tbase = (qc * NT + ti) * 32;
    const int nvalid = img < p.imgs ? min(32, max(0, p.N - tbase)) : 0;
    const long img_row0 = (long)(img < p.imgs ? img : img0) * p.N;
    const int rr = lane / LPR, cc = (lane % LPR) * 16;
    const float* xr = p.x + row * C;
    const float* pr = p.pend ? p.pend + row * C : nullptr;
    const float d1 = p.drop1 ? p.drop1[img < p.imgs ? img : img0] : 1.f;
    float s = 0.f;
    // the token's row again (L2 / Infinity Cache): all pieces in flight at once, as in the prologue -- interleaved with their uses they
    // were 16 more serialized round trips (40 % of the kernel's time by the in-kernel clocks)
    f32x4 tv[OC * 4];
#pragma unroll
    for (int i = 0; i < OC * 4; ++i) tv[i] = *reinterpret_cast<const f32x4*>(xr + (i >> 2) * 32 + 8 * (i & 3) + 4 * half);
    __builtin_amdgcn_sched_barrier(0);
    if (pr) {
      f32x4 pv[OC * 4];
#pragma unroll
      for (int i = 0; i < OC * 4; ++i) pv[i] = *reinterpret_cast<const f32x4*>(pr + (i >> 2) * 32 + 8 * (i & 3) + 4 * half);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < OC * 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) tv[i][j] = fmaf(ds, pv[i][j], tv[i][j]);
    }
#pragma unroll
    for (int oc = 0; oc < OC; ++oc)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = oc * 32 + 8 * k + 4 * half;
        const f32x4 b = *reinterpret_cast<const f32x4*>(ln_lds + 4 * C + c);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = fmaf(d1, out[oc][4 * k + j] * invp + b[j], tv[oc * 4 + k][j]);
          out[oc][4 * k + j] = v[j];
          s += v[j];
        }
        *reinterpret_cast<f32x4*>(stg + l31 * ROWB + c * 4) = v;
      }
    {
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int r = i * RPI + rr;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * ROWB + cc);
        if (r < nvalid) *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(p.x_out + (img_row0 + token_of(tbase + r)) * C) + cc) = v;
      }
    }
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int oc = 0; oc < OC; ++oc)
#pragma unroll
      for (int e = 0; e < 16; ++e) { const float d = out[oc][e] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    const float rstd = 1.f / sqrtf(q / (float)C + p.eps2);
#pragma unroll
    for (int oc = 0; oc < OC; ++oc)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = oc * 32 + 8 * k + 4 * half;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(ln_lds + 2 * C + c);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(ln_lds + 3 * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[oc][4 * k + j] = (out[oc][4 * k + j] - mean) * rstd * gm[j] + bt[j];
      }
    {
      // ---- half-block form: LayerNorm2(x_out) leaves as the HL32 operand of the Mlp kernel, through the same LDS tile
#pragma unroll
      for (int oc = 0; oc < OC; ++oc)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f16v4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            _Float16 a, b;
            split_f16(out[oc][4 * k + j], a, b);
            hi[j] = a;
            lo[j] = b;
          }
          unsigned char* line = stg + l31 * ROWB + oc * 128 + (8 * k + 4 * half) * 2;      // the row's HL32 lines: [hi 32 | lo 32] per block
          *reinterpret_cast<f16v4*>(line) = hi;
          *reinterpret_cast<f16v4*>(line + 64) = lo;
        }
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int r = i * RPI + rr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + r * ROWB + cc);
        if (r < nvalid) *reinterpret_cast<u32x4*>(p.y_hl + (img_row0 + token_of(tbase + r)) * (long)CB * 128 + cc) = v;
      }
    }
  }
#ifdef MRN_XPROBE_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0) {
    const long x_e1 = __builtin_readcyclecounter();
    xdbg[5] += x_e1 - x_e0;
    for (int i = 0; i < 6; ++i) atomicAdd(&g_mixer_dbg[i], (unsigned long long)xdbg[i]);
    atomicAdd(&g_mixer_dbg[6], 1ull);
    atomicAdd(&g_mixer_dbg[7], (unsigned long long)(x_e1 - x_t0));
  }
#endif
}

template <int C, int NT, int IMG, int CHUNKS, int RING, bool ATTN = false>
int launch_mixer(const MixerParams& p, hipStream_t st) {
  constexpr size_t ring = 2 * IMG * NT * 32 * 128 + RING * C * 128, stage = ATTN ? 0 : (size_t)NT * IMG * 32 * (C * 4 + 16);
  constexpr size_t ldsz = (ring > stage ? ring : stage) + 8 * C * sizeof(float);      // + the parameter block (PARAM_OFF in the kernel)
  static_assert(ldsz <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)svtr_mixer_kernel<C, NT, IMG, CHUNKS, RING, ATTN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    attr_set = true;
  }
  hipLaunchKernelGGL((svtr_mixer_kernel<C, NT, IMG, CHUNKS, RING, ATTN>), dim3((unsigned)((p.imgs + IMG - 1) / IMG * CHUNKS)), dim3(NT * IMG * 64), ldsz, st, p);
  MRN_LAUNCH_CHECK(ATTN ? "svtr_attention_block_x3" : "svtr_mixer_x3");
  return MRN_OK;
}

int dispatch_mixer(const MixerParams& p, int C, hipStream_t st) {
  const int N = p.N;
  if (C == 64) return N <= 224 ? launch_mixer<64, 7, 1, 1, 4>(p, st) : N <= 256 ? launch_mixer<64, 8, 1, 1, 4>(p, st) : launch_mixer<64, 8, 1, 2, 4>(p, st);
  return N <= 128 ? launch_mixer<128, 4, 2, 1, 4>(p, st) : launch_mixer<128, 8, 1, 1, 4>(p, st);
}

bool mixer_shape_ok(int C, int N, int imgs_per_group) {
  return N >= 1 && ((C == 64 && N <= 512) || (C == 128 && N <= 256 && (N > 128 || imgs_per_group % 2 == 0)));
}

}  // namespace

// The attention half of an SVTR mixing block (modules/svtr.py:196-201 first line; Attention :90-152) for the images of G lock-step
// experts (group = image / imgs_per_group), fused:
//   t = x + drop_prev[img] * pending;  x_out = t + drop1[img] * proj(attention(qkv(LayerNorm1(t))));  y_hl = HL32(LayerNorm2(x_out))
//   x, pending, x_out  [imgs][N][C] fp32 (pending, drop_prev, drop1 may be null)
//   g1, b1, g2, b2     [G][C] LayerNorm parameters; eps1, eps2
//   wqkv_hl            [G][3C][C/32][128 B] mrn_pack_weight_hl32 of the qkv weight [3C][1][C]; sqkv [G][2] {s, 1/s}; bqkv [G][3C] or null
//   mask_bits          [N][ceil(N/32)] visibility bits (bit j of word w: key 32 w + j) of the local mixer's 0 / -inf mask, or null
//   wproj_hl           [G][C][C/32][128 B] proj weight packed from [C][1][C] with the input channel of every 32-block (= head) permuted:
//                      position 16 s + 8 h + j holds channel (j & 3) + 8 (2 s + (j >> 2)) + 4 h; sproj [G][2]; bproj [G][C]
//   y_hl               [imgs * N][C/32][128 B]
//   token_h, token_w   0, 0: tokens are walked in memory order.  token_h x token_w == N (local mixers on a token_h x token_w map): the kernel
//                      walks them COLUMN-major -- position p = col * token_h + row is token row * token_w + col -- and mask_bits must be
//                      given in POSITION order (rows and bits permuted alike); the result is the same function of x, token by token, but
//                      a 32-position key tile is a block of whole columns, so the local window leaves most tiles fully masked (skipped)
// Heads have 32 channels.  Supported: C = 64 with N <= 512, C = 128 with N <= 256 (N <= 128: imgs_per_group even); anything else
// returns MRN_ERR_UNSUPPORTED and the caller runs the unfused chain.
MRN_EXPORT int mrn_svtr_mixer_x3_f32(const float* x, const float* pending, const float* drop_prev, const float* g1, const float* b1,
                                     float eps1, const void* wqkv_hl, const float* sqkv, const float* bqkv, const void* mask_bits,
                                     float scale, const void* wproj_hl, const float* sproj, const float* bproj, const float* drop1,
                                     const float* g2, const float* b2, float eps2, float* x_out, void* y_hl, int imgs,
                                     int imgs_per_group, int N, int C, int token_h, int token_w, void* stream) {
  MRN_CHECK_ARG(x && g1 && b1 && wqkv_hl && wproj_hl && bproj && g2 && b2 && x_out && y_hl && imgs >= 0 && imgs_per_group >= 1,
                "mrn_svtr_mixer_x3_f32: bad operands");
  MRN_CHECK_ARG((uintptr_t)wqkv_hl % 128 == 0 && (uintptr_t)wproj_hl % 128 == 0 && (uintptr_t)y_hl % 128 == 0 &&
                    (uintptr_t)x % 16 == 0 && (uintptr_t)x_out % 16 == 0 && (!pending || (uintptr_t)pending % 16 == 0),
                "mrn_svtr_mixer_x3_f32: operands must be 128-byte (HL32) / 16-byte (fp32) aligned");
  if (!mixer_shape_ok(C, N, imgs_per_group)) {
    mrn_set_error("mrn_svtr_mixer_x3_f32: unsupported shape C=%d N=%d imgs_per_group=%d", C, N, imgs_per_group);
    return MRN_ERR_UNSUPPORTED;
  }
  if (imgs == 0) return MRN_OK;
  MRN_CHECK_ARG(token_h == 0 || (token_h > 0 && token_w > 0 && token_h * token_w == N && mask_bits),
                "mrn_svtr_mixer_x3_f32: column-major token walk needs token_h * token_w == N (%d x %d vs %d) and a mask in that order", token_h, token_w, N);
  MixerParams p;
  p.perm_h = token_h; p.perm_w = token_w;
  p.x = x; p.pend = pending; p.drop_prev = drop_prev; p.g1 = g1; p.b1 = b1; p.wqkv = (const unsigned char*)wqkv_hl; p.sqkv = sqkv;
  p.bqkv = bqkv; p.mask_bits = (const unsigned*)mask_bits; p.wproj = (const unsigned char*)wproj_hl; p.sproj = sproj; p.bproj = bproj;
  p.drop1 = drop1; p.g2 = g2; p.b2 = b2; p.x_out = x_out; p.y_hl = (unsigned char*)y_hl;
  p.imgs = imgs; p.imgs_per_group = imgs_per_group; p.N = N; p.scale = scale; p.eps1 = eps1; p.eps2 = eps2;
  return dispatch_mixer(p, C, (hipStream_t)stream);
}

// (The whole Block.forward in one launch -- this kernel followed by the Mlp half on the same registers -- was built, parity-tested and
// measured slower than the two half-block kernels in round 3: C = 128, 256 tokens 1038 vs 965 us per block of 6 x 256 images; the Mlp
// phase inherits the one-workgroup-per-CU occupancy of this kernel.  Removed in round 4.)

// Attention-only form for the wide stage (C = 256, SVTR stage 3; reference modules/svtr.py:130-152 behind :200's norm1):
//   t = x + drop_prev * pending  (written to t_out when pending is given);  ctx = attention(qkv(LayerNorm1(t)))
// ctx_hl [imgs * N][C/32][128 B] is the HL32 operand of the (unfused) proj Linear; the residual add and LayerNorm2 follow as
// mrn_add_layernorm_grouped_f32.  Saves the LayerNorm pass and the qkv round trip of the unfused chain.  C = 256, N <= 128,
// imgs_per_group a multiple of 2 (N > 64) or 4 (N <= 64).  Other arguments as mrn_svtr_mixer_x3_f32.
MRN_EXPORT int mrn_svtr_attention_block_x3_f32(const float* x, const float* pending, const float* drop_prev, const float* g1, const float* b1,
                                               float eps1, const void* wqkv_hl, const float* sqkv, const float* bqkv, const void* mask_bits,
                                               float scale, float* t_out, void* ctx_hl, int imgs, int imgs_per_group, int N, int C,
                                               void* stream) {
  MRN_CHECK_ARG(x && g1 && b1 && wqkv_hl && ctx_hl && imgs >= 0 && imgs_per_group >= 1 && (!pending || t_out),
                "mrn_svtr_attention_block_x3_f32: bad operands");
  MRN_CHECK_ARG((uintptr_t)wqkv_hl % 128 == 0 && (uintptr_t)ctx_hl % 128 == 0 && (uintptr_t)x % 16 == 0 && (!pending || (uintptr_t)pending % 16 == 0) &&
                    (!t_out || (uintptr_t)t_out % 16 == 0), "mrn_svtr_attention_block_x3_f32: operands must be 128-byte (HL32) / 16-byte (fp32) aligned");
  const bool ok = C == 256 && N >= 1 && N <= 128 && imgs_per_group % (N <= 64 ? 4 : 2) == 0;
  if (!ok) {
    mrn_set_error("mrn_svtr_attention_block_x3_f32: unsupported shape C=%d N=%d imgs_per_group=%d", C, N, imgs_per_group);
    return MRN_ERR_UNSUPPORTED;
  }
  if (imgs == 0) return MRN_OK;
  MixerParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.pend = pending; p.drop_prev = drop_prev; p.g1 = g1; p.b1 = b1; p.wqkv = (const unsigned char*)wqkv_hl; p.sqkv = sqkv;
  p.bqkv = bqkv; p.mask_bits = (const unsigned*)mask_bits; p.x_out = t_out; p.y_hl = (unsigned char*)ctx_hl;
  p.imgs = imgs; p.imgs_per_group = imgs_per_group; p.N = N; p.scale = scale; p.eps1 = eps1; p.eps2 = eps1;
  const hipStream_t st = (hipStream_t)stream;
  return N <= 64 ? launch_mixer<256, 2, 4, 1, 2, true>(p, st) : launch_mixer<256, 4, 2, 1, 2, true>(p, st);
}
