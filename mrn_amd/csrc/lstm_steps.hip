// LSTM layers of the frozen experts as ONE SMALL KERNEL PER TIME STEP, replayed from a HIP graph.
//
// Reference: modules/sequence_modeling.py:7-21 (nn.LSTM, bidirectional, gate order i, f, g, o).  Same arithmetic as
// lstm_layer_x3_kernel (rnn.hip): recurrent product as split-fp16 x3 on the f16 MFMA with fp32 accumulation, W_hh prescaled by a
// power of two, pointwise part in fp32.
//
// Why.  The persistent kernel gives a 16-sample tile to one workgroup for the whole sequence; every step that workgroup streams the
// whole W_hh (1 MiB as hi + lo fp16) from L2, and ONE CU cannot pull more than ~56 GB/s (its loads in flight / the L2 latency): 18.6 us
// per step whatever else happens, with 32 .. 192 of the 256 CUs busy.  Here a step is a grid over (expert, direction, 128-sample
// tile, 32-unit tile): a workgroup needs 128 KiB of W_hh and 128 KiB of h, and ALL CUs pull at once (aggregate L2 bandwidth instead
// of one CU's).  What made this form a loser on paper -- T launches per layer -- is what the graph removes: a dependent kernel node
// costs 1.6 us on this machine (tools/launch_gap.py; 6.5 us as a plain launch through the Python binding).  The step kernels read
// their buffer pointers from a small device-side argument block, so ONE instantiated graph per (T, grid) serves every call: a
// one-thread kernel refreshes the block, then the graph is launched on the caller's stream.
//
// Layouts.  h crosses the steps as HL32 lines (per sample 8 lines of [hi 32 units | lo 32 units]) in a ping-pong buffer: exactly the
// A-operand rows of the next step's product.  W_hh is mrn_pack_weight_hl32 of [4H][1][H] per direction (the B operand; a K-slab of 32
// is one line per gate column).  A wave owns 32 samples x 32 units x 4 gates: the four gate pre-activations of a (sample, unit) land
// in the same lane and register of four accumulators, so the pointwise update is lane-local; lanes run along the units, so xproj,
// c and out are read / written as 128-byte segments.
#include "common.hpp"
#include <map>
#include <mutex>
#include <tuple>
#include <stdlib.h>

namespace {

typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int HID = 256;
constexpr int SK_TS = 128;          // samples per workgroup (4 waves x 32)
constexpr int SK_TU = 32;           // hidden units per workgroup (x 4 gates = 128 gate columns)
constexpr int SK_UT = HID / SK_TU;  // unit tiles
constexpr int SK_MAX_GROUPS = 8;
constexpr int SK_SLAB = 4 * SK_TU * 128;      // one K-slab of 32: 128 gate columns x one 128-byte line
constexpr int SK_RING = 4;
constexpr int SK_NCB = HID / 32;

struct StepGroup {
  const float* xproj;              // [B][T][ndir * 4H]  W_ih x + b_ih
  const unsigned char* w_hl;       // [ndir][4H][H/32][128 B] HL32 (prescaled)
  const float* w_inv;              // [ndir] 1 / prescale
  const float* b_hh;               // [ndir * 4H] or null
  float* out;                      // [B][T][ndir * H]
};
struct StepArgs {
  StepGroup g[SK_MAX_GROUPS];
  unsigned char* hbuf;             // [2][nsets][Bpad][H/32][128 B]
  float* cbuf;                     // [nsets][Bpad][H]
  int B, Bpad, T, ndir, nsets, stiles;
};

__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16v8*>(&a), *reinterpret_cast<const f16v8*>(&b), c, 0, 0, 0);
}
// v_exp_f32 + v_rcp_f32 forms (1 ulp each; tanh(x) = 1 - 2 / (exp(2x) + 1) carries an ABSOLUTE error of ~1e-7 near 0, the form the
// attention decoder's score path uses): a lane updates 16 (sample, unit) pairs per step, the library expf / tanhf cost 7 us of it
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

__global__ void lstm_set_args_kernel(const StepArgs v, StepArgs* dst) {
  if (threadIdx.x == 0) *dst = v;
}

__global__ __launch_bounds__(256) void lstm_step_x3_kernel(const StepArgs* __restrict__ ap, const int step) {
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
  const int t_ = threadIdx.x, lane = t_ & 63, wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int B = ap->B, Bpad = ap->Bpad, T = ap->T, ndir = ap->ndir, nsets = ap->nsets, stiles = ap->stiles;
  const int ut = blockIdx.x % SK_UT;
  const int st = (blockIdx.x / SK_UT) % stiles;
  const int set = blockIdx.x / (SK_UT * stiles);
  const int gi = set / ndir, dir = set - gi * ndir;
  const StepGroup grp = ap->g[gi];
  const int t = dir == 0 ? step : T - 1 - step;
  const int j = ut * SK_TU + l31;                      // this lane's hidden unit
  const int sb = st * SK_TS + wave * 32;               // first sample of this wave

  // ---- W_hh K-slabs through an LDS ring: slab cb = lines (gate, unit) x channel block cb; DMA instruction d moves 8 lines
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(grp.w_hl + (long)dir * 4 * HID * SK_NCB * 128), 0,
                                                                      4 * HID * SK_NCB * 128, 0x00020000);
  auto issue = [&](int cb, unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = wave * 4 + i;                      // 16 instructions per slab
      const int L = d * 8 + (lane >> 3);               // line inside the slab: gate = L / 32, unit = L % 32
      const int row = (L >> 5) * HID + ut * SK_TU + (L & 31);
      const int coff = ((lane & 7) ^ ((L >> 1) & 7)) << 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(buf + d * 1024), 16, (row * SK_NCB + cb) * 128 + coff, 0, 0, 0);
    }
  };

  // ---- everything this step reads besides W, issued BEFORE the slab DMAs (the counted waits below assume nothing younger than a DMA
  // but DMAs): pointwise inputs (lanes along the units: 128-byte segments) and the A-operand fragments of h(t-1)
  float xg[4][16], cprev[16];
  const float inv = grp.w_inv[dir];
  float bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = grp.b_hh ? grp.b_hh[dir * 4 * HID + g * HID + j] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int s = sb + 8 * (e >> 2) + (e & 3) + 4 * half;
    const float* xp = grp.xproj + ((long)(s < B ? s : 0) * T + t) * (ndir * 4 * HID) + dir * 4 * HID + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) xg[g][e] = xp[g * HID];
    cprev[e] = step > 0 ? ap->cbuf[((long)set * Bpad + s) * HID + j] : 0.f;
  }
  u32x4 ah[SK_NCB][2], al[SK_NCB][2];
  if (step > 0) {
    const unsigned char* hrow = ap->hbuf + ((((long)(step & 1) * nsets + set) * Bpad + sb + l31) * SK_NCB) * 128;
#pragma unroll
    for (int cb = 0; cb < SK_NCB; ++cb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ah[cb][ks] = *reinterpret_cast<const u32x4*>(hrow + cb * 128 + ((ks * 2 + half) << 4));
        al[cb][ks] = *reinterpret_cast<const u32x4*>(hrow + cb * 128 + 64 + ((ks * 2 + half) << 4));
      }
  }

  f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;

  if (step > 0) {
#pragma unroll
    for (int i = 0; i < SK_RING - 1; ++i) issue(i, lds + i * SK_SLAB);
    const int key = (l31 >> 1) & 7;
    int foff[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) foff[pl][ks] = l31 * 128 + (((pl * 4 + ks * 2 + half) ^ key) << 4);
#pragma unroll
    for (int cb = 0; cb < SK_NCB; ++cb) {
      // own DMAs of slab cb retired (the younger slabs of the ring stay in flight: 4 instructions each), then everyone's
      constexpr int VM0 = (7 << 4) | (15 << 8);
      const int younger = (SK_NCB - 1 - cb) < (SK_RING - 2) ? (SK_NCB - 1 - cb) : (SK_RING - 2);
      if (younger == 2) __builtin_amdgcn_s_waitcnt(8 | VM0);
      else if (younger == 1) __builtin_amdgcn_s_waitcnt(4 | VM0);
      else __builtin_amdgcn_s_waitcnt(VM0);
      __syncthreads();
      if (cb + SK_RING - 1 < SK_NCB) issue(cb + SK_RING - 1, lds + ((cb + SK_RING - 1) % SK_RING) * SK_SLAB);
      const unsigned char* cur = lds + (cb % SK_RING) * SK_SLAB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x4 wl = *reinterpret_cast<const u32x4*>(cur + g * 4096 + foff[1][ks]);
          const u32x4 wh = *reinterpret_cast<const u32x4*>(cur + g * 4096 + foff[0][ks]);
          acc[g] = mma(al[cb][ks], wh, acc[g]);      // (h_lo * w_hi, h_hi * w_lo, h_hi * w_hi: the order of rnn.hip)
          acc[g] = mma(ah[cb][ks], wl, acc[g]);
          acc[g] = mma(ah[cb][ks], wh, acc[g]);
        }
    }
  }

  // ---- pointwise update: register e of this lane is sample sb + 8 (e >> 2) + (e & 3) + 4 half, unit j
  unsigned char* hnext = ap->hbuf + ((((long)((step + 1) & 1) * nsets + set) * Bpad) * SK_NCB + ut) * 128;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int s = sb + 8 * (e >> 2) + (e & 3) + 4 * half;
    const float gi_ = acc[0][e] * inv + xg[0][e] + bh[0];
    const float gf = acc[1][e] * inv + xg[1][e] + bh[1];
    const float gg = acc[2][e] * inv + xg[2][e] + bh[2];
    const float go = acc[3][e] * inv + xg[3][e] + bh[3];
    const float ig = sigm(gi_), fg = sigm(gf), og = sigm(go), cg = tanh_fast(gg);
    const float cn = fg * cprev[e] + ig * cg;
    const float h = og * tanh_fast(cn);
    if (s < B) {
      ap->cbuf[((long)set * Bpad + s) * HID + j] = cn;
      grp.out[((long)s * T + t) * (ndir * HID) + dir * HID + j] = h;
    }
    _Float16 hh, hl;
    split_f16(s < B ? h : 0.f, hh, hl);
    unsigned char* line = hnext + (long)s * SK_NCB * 128;
    *reinterpret_cast<_Float16*>(line + l31 * 2) = hh;
    *reinterpret_cast<_Float16*>(line + 64 + l31 * 2) = hl;
  }
}

// ---- one instantiated graph per (stream, T, grid): T dependent kernel nodes reading one device-side argument block
struct StepGraph {
  hipGraphExec_t exec = nullptr;
  StepArgs* args = nullptr;
};
std::mutex g_mutex;
std::map<std::tuple<void*, int, int>, StepGraph> g_graphs;

int build_graph(StepGraph& sg, int T, int blocks) {
  hipError_t e = hipMalloc((void**)&sg.args, sizeof(StepArgs));
  if (e != hipSuccess) { mrn_set_error("lstm_steps: hipMalloc of the argument block failed: %s", hipGetErrorString(e)); return (int)e; }
  hipStream_t cs;
  e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  if (e != hipSuccess) { mrn_set_error("lstm_steps: stream creation failed: %s", hipGetErrorString(e)); return (int)e; }
  hipGraph_t graph = nullptr;
  e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (e == hipSuccess) {
    for (int s = 0; s < T; ++s)
      hipLaunchKernelGGL(lstm_step_x3_kernel, dim3(blocks), dim3(256), SK_RING * SK_SLAB, cs, (const StepArgs*)sg.args, s);
    e = hipStreamEndCapture(cs, &graph);
  }
  if (e == hipSuccess) e = hipGraphInstantiate(&sg.exec, graph, nullptr, nullptr, 0);
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipStreamDestroy(cs);
  if (e != hipSuccess) { mrn_set_error("lstm_steps: graph capture failed: %s", hipGetErrorString(e)); return (int)e; }
  return MRN_OK;
}

}  // namespace

// Bytes of the workspace of mrn_lstm_layer_fwd_x3_steps: the ping-pong h planes (HL32 lines) and the cell state
MRN_EXPORT int64_t mrn_lstm_steps_workspace_bytes(int groups, int B, int ndir) {
  const long Bpad = (long)ceil_div(B, SK_TS) * SK_TS;
  return 2L * groups * ndir * Bpad * HID * 4 + (long)groups * ndir * Bpad * HID * 4;
}

// Inference-only LSTM layers of `groups` frozen experts (modules/sequence_modeling.py:7-21), one kernel per time step replayed from a
// HIP graph (see the head of this file).  Pointer arguments are HOST arrays of `groups` device pointers:
//   xproj[g] [B][T][ndir*4H] fp32 (W_ih x + b_ih); w_hl[g] [ndir][4H][H/32][128 B] = mrn_pack_weight_hl32 of W_hh per direction;
//   w_inv[g] device float[ndir] = 1 / prescale; b_hh[g] [ndir*4H] (the array or an entry may be NULL); out[g] [B][T][ndir*H] fp32.
// workspace: mrn_lstm_steps_workspace_bytes(groups, B, ndir) bytes, 128-byte aligned, owned by the call (no initialisation needed).
// groups <= 8, hidden == 256.  Same arithmetic as mrn_lstm_layer_fwd_x3_grouped (another summation order inside the recurrent product).
MRN_EXPORT int mrn_lstm_layer_fwd_x3_steps(const void* const* xproj, const void* const* w_hl, const void* const* w_inv,
                                           const void* const* b_hh, const void* const* out, int groups, int B, int T, int hidden,
                                           int ndir, void* workspace, int64_t workspace_bytes, void* stream) {
  MRN_CHECK_ARG(xproj && w_hl && w_inv && out && workspace && groups >= 1 && groups <= SK_MAX_GROUPS, "mrn_lstm_layer_fwd_x3_steps: bad operands");
  MRN_CHECK_ARG(hidden == HID, "mrn_lstm_layer_fwd_x3_steps: hidden=%d unsupported (library is built for %d)", hidden, HID);
  MRN_CHECK_ARG(ndir == 1 || ndir == 2, "mrn_lstm_layer_fwd_x3_steps: ndir=%d", ndir);
  MRN_CHECK_ARG(workspace_bytes >= mrn_lstm_steps_workspace_bytes(groups, B, ndir) && (uintptr_t)workspace % 128 == 0,
                "mrn_lstm_layer_fwd_x3_steps: workspace too small or misaligned");
  if (B == 0 || T == 0) return MRN_OK;
  StepArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < groups; ++i) {
    MRN_CHECK_ARG(xproj[i] && w_hl[i] && w_inv[i] && out[i] && (uintptr_t)w_hl[i] % 128 == 0, "mrn_lstm_layer_fwd_x3_steps: bad operand in group %d", i);
    a.g[i] = StepGroup{(const float*)xproj[i], (const unsigned char*)w_hl[i], (const float*)w_inv[i],
                       (b_hh && b_hh[i]) ? (const float*)b_hh[i] : nullptr, (float*)out[i]};
  }
  a.B = B; a.T = T; a.ndir = ndir; a.nsets = groups * ndir; a.stiles = ceil_div(B, SK_TS); a.Bpad = a.stiles * SK_TS;
  a.hbuf = (unsigned char*)workspace;
  a.cbuf = (float*)((unsigned char*)workspace + 2L * a.nsets * a.Bpad * HID * 4);
  const int blocks = a.nsets * a.stiles * SK_UT;
  const hipStream_t st = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)lstm_step_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SK_RING * SK_SLAB); attr = true; }
  static const bool use_graph = !(getenv("MRN_LSTM_STEPS_GRAPH") && atoi(getenv("MRN_LSTM_STEPS_GRAPH")) == 0);
  StepGraph sg;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    StepGraph& ref = g_graphs[std::make_tuple((void*)st, T, blocks)];
    if (!ref.exec) {
      const int rc = build_graph(ref, T, blocks);
      if (rc) { g_graphs.erase(std::make_tuple((void*)st, T, blocks)); return rc; }
    }
    sg = ref;
  }
  hipLaunchKernelGGL(lstm_set_args_kernel, dim3(1), dim3(64), 0, st, a, sg.args);
  if (use_graph) {
    const hipError_t e = hipGraphLaunch(sg.exec, st);
    if (e != hipSuccess) { mrn_set_error("mrn_lstm_layer_fwd_x3_steps: hipGraphLaunch failed: %s", hipGetErrorString(e)); return (int)e; }
  } else {
    for (int s = 0; s < T; ++s)
      hipLaunchKernelGGL(lstm_step_x3_kernel, dim3(blocks), dim3(256), SK_RING * SK_SLAB, st, (const StepArgs*)sg.args, s);
  }
  MRN_LAUNCH_CHECK("lstm_step_x3");
  return MRN_OK;
}
