// Elementwise passes between the grouped convolutions (conv_x3.hip) when G frozen experts run their backbones in
// lock-step: BatchNorm statistics finalisation for G BatchNorm modules at once, BatchNorm-apply (+ residual + ReLU)
// and BatchNorm-apply + ReLU + MaxPool, each writing the fp32 tensor and / or the HL32 split-fp16 operand the next
// convolution stages by DMA (one 128-byte line [hi x 32 | lo x 32] per (pixel, 32-channel block)).
//
// Reference op sites: BatchNorm2d / ReLU / residual add / MaxPool2d of modules/feature_extraction.py:171-199,222-294
// and modules/transformation.py:69-81, evaluated for every expert of modules/model.py:399-401.
// All kernels are HBM-bound: each activation byte is read once, 32 B per lane, channels innermost.
#include "common.hpp"

namespace {

typedef _Float16 f16v8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16v4 __attribute__((ext_vector_type(4)));

struct F8 {
  f32x4 a, b;
};

__device__ __forceinline__ F8 load8(const float* p) {
  F8 v;
  v.a = *reinterpret_cast<const f32x4*>(p);
  v.b = *reinterpret_cast<const f32x4*>(p + 4);
  return v;
}
__device__ __forceinline__ void store8(float* p, const F8& v) {
  *reinterpret_cast<f32x4*>(p) = v.a;
  *reinterpret_cast<f32x4*>(p + 4) = v.b;
}
// 8 consecutive channels c8*8 .. +7 of row `row` -> their slots in the HL32 image
__device__ __forceinline__ void store_hl(unsigned char* out, long row, int C, int c8, const F8& v) {
  f16v8 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 hh, ll;
    split_f16_sat(v.a[e], hh, ll);       // (saturating: an activation beyond the fp16 range -- 6.5e4 plain, 6.5e3 under B^T of F(4,3) --
    h[e] = hh; l[e] = ll;                //  stays finite instead of hi = inf, lo = -inf -> NaN products)
    split_f16_sat(v.b[e], hh, ll);
    h[4 + e] = hh; l[4 + e] = ll;
  }
  unsigned char* o = out + (row * (C >> 5) + (c8 >> 2)) * 128 + (c8 & 3) * 16;
  *reinterpret_cast<f16v8*>(o) = h;
  *reinterpret_cast<f16v8*>(o + 64) = l;
}

// the reduced-precision operand line: 64 channels of plain fp16 per 128 bytes (conv_wino.hip, DENSE), saturating like store_hl
__device__ __forceinline__ void store_d16(unsigned char* out, long row, int C, int c8, const F8& v, int dense = 1) {
  unsigned char* dst = out + (row * (C >> 6) + (c8 >> 3)) * 128 + (c8 & 7) * 16;
  if (dense == 2) {            // bfloat16, round to nearest even (the comparison instantiation of the reduced mode)
    typedef unsigned short u16v8 __attribute__((ext_vector_type(8)));
    u16v8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned u = __float_as_uint(e < 4 ? v.a[e] : v.b[e - 4]);
      h[e] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    *reinterpret_cast<u16v8*>(dst) = h;
    return;
  }
  f16v8 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (_Float16)fminf(fmaxf(v.a[e], -65504.f), 65504.f);
    h[4 + e] = (_Float16)fminf(fmaxf(v.b[e], -65504.f), 65504.f);
  }
  *reinterpret_cast<f16v8*>(dst) = h;
}

// a block = 32 channels x 32 row lanes of one group over ONE CHUNK of the partial rows (coalesced 128-byte reads, sums in double as
// bn_finalize_kernel, spatial.hip: same arithmetic), parameters through pointer tables.  gridDim.z chunks: the first layers come with up
// to 16384 partial rows per expert (128-pixel blocks of the 32 x 256 maps) for 32 / 64 channels -- one wave per channel walked them in
// 256 dependent steps (300 us per launch, five such launches per loop-B step).  Chunk sums go to `ws` [G][cblocks][Z][2][32] doubles; the
// last block to arrive (ticket per (group, channel block), self-resetting) adds them IN CHUNK ORDER -- deterministic -- and finalises.
__global__ __launch_bounds__(1024) void bn_finalize_grouped_kernel(const float* __restrict__ part, int nblk, int C, long count,
                                                                   const float* const* __restrict__ ptrs, int G, float momentum,
                                                                   float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                                   double* __restrict__ ws, unsigned* __restrict__ tickets) {
  __shared__ double red[2][32][33];
  __shared__ unsigned last;
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int g = blockIdx.y, Z = gridDim.z, z = blockIdx.z;
  const int chunk = (nblk + Z - 1) / Z, b0 = z * chunk, b1 = min(nblk, b0 + chunk);
  const float* pg = part + (long)g * nblk * 2 * C;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int b = b0 + lane; b < b1; b += 32) {
      s += (double)pg[((long)b * 2 + 0) * C + c];
      q += (double)pg[((long)b * 2 + 1) * C + c];
    }
  red[0][lane][cl] = s;
  red[1][lane][cl] = q;
  __syncthreads();
  if (threadIdx.x < 32) {
    s = q = 0.0;
#pragma unroll
    for (int l = 0; l < 32; ++l) {
      s += red[0][l][cl];
      q += red[1][l][cl];
    }
  }
  if (Z > 1) {
    const long slot = (long)g * gridDim.x + blockIdx.x;
    double* w = ws + slot * Z * 64;
    if (threadIdx.x < 32) {
      w[z * 64 + cl] = s;
      w[z * 64 + 32 + cl] = q;
      __threadfence();
    }
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(tickets + slot, 1u) == (unsigned)(Z - 1);
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) tickets[slot] = 0u;
    __threadfence();
    if (threadIdx.x < 32) {
      s = q = 0.0;
      for (int k = 0; k < Z; ++k) {
        s += __builtin_nontemporal_load(w + k * 64 + cl);
        q += __builtin_nontemporal_load(w + k * 64 + 32 + cl);
      }
    }
  }
  if (threadIdx.x < 32 && c < C) {
    const float* gamma = ptrs[0 * G + g];
    const float* beta = ptrs[1 * G + g];
    float* run_mean = const_cast<float*>(ptrs[2 * G + g]);
    float* run_var = const_cast<float*>(ptrs[3 * G + g]);
    const double mean = s / (double)count;
    double var = q / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = gm * invstd;
    scale[(long)g * C + c] = sc;
    shift[(long)g * C + c] = bt - (float)mean * sc;
    if (run_mean) {
      const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
  }
}

// eval-mode BatchNorm of G modules folded to (scale, shift) [G][C] from the CURRENT running statistics, through the same
// pointer table as bn_finalize_grouped_kernel (same arithmetic as bn_eval_affine_kernel, spatial.hip)
__global__ __launch_bounds__(256) void bn_eval_affine_grouped_kernel(const float* const* __restrict__ ptrs, int G, int C, float eps,
                                                                     float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (c >= C) return;
  const float* gamma = ptrs[0 * G + g];
  const float* beta = ptrs[1 * G + g];
  const float* run_mean = ptrs[2 * G + g];
  const float* run_var = ptrs[3 * G + g];
  const float invstd = 1.f / sqrtf(run_var[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * invstd;
  scale[(long)g * C + c] = sc;
  shift[(long)g * C + c] = (beta ? beta[c] : 0.f) - run_mean[c] * sc;
}

// 8 consecutive channels of row `row` from an HL32 image: value = hi + lo
__device__ __forceinline__ F8 load_hl(const unsigned char* hl, long row, int C, int c8) {
  const unsigned char* o = hl + (row * (C >> 5) + (c8 >> 2)) * 128 + (c8 & 3) * 16;
  const f16v8 h = *reinterpret_cast<const f16v8*>(o), l = *reinterpret_cast<const f16v8*>(o + 64);
  F8 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v.a[e] = (float)h[e] + (float)l[e];
    v.b[e] = (float)h[4 + e] + (float)l[4 + e];
  }
  return v;
}

// out = act(y * scale[g][c] + shift[g][c] (+ res)); one lane = 8 channels; the residual comes as fp32 or as an HL32 image
__global__ __launch_bounds__(256) void bn_apply_grouped_kernel(const float* __restrict__ y, const float* __restrict__ res,
                                                               const unsigned char* __restrict__ res_hl,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               float* __restrict__ out, unsigned char* __restrict__ out_hl,
                                                               long rows_per_group, long n8, int C, int relu) {
  const int C8 = C >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const long row = i / C8;
    const int g = (int)(row / rows_per_group);
    F8 v = load8(y + i * 8);
    if (scale) {
      const F8 sc = load8(scale + (long)g * C + c8 * 8), sh = load8(shift + (long)g * C + c8 * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v.a[e] = v.a[e] * sc.a[e] + sh.a[e];
        v.b[e] = v.b[e] * sc.b[e] + sh.b[e];
      }
    }
    if (res || res_hl) {
      const F8 r = res ? load8(res + i * 8) : load_hl(res_hl, row, C, c8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v.a[e] += r.a[e];
        v.b[e] += r.b[e];
      }
    }
    if (relu == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v.a[e] = relu_nan(v.a[e]);
        v.b[e] = relu_nan(v.b[e]);
      }
    } else if (relu == 2) {   // GELU (erf), SVTR PatchEmbed (modules/svtr.py:227-233), as mrn_scale_shift_act_f32
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v.a[e] = 0.5f * v.a[e] * (1.f + erff(v.a[e] * 0.70710678118654752440f));
        v.b[e] = 0.5f * v.b[e] * (1.f + erff(v.b[e] * 0.70710678118654752440f));
      }
    }
    if (out) store8(out + i * 8, v);
    if (out_hl) store_hl(out_hl, row, C, c8, v);
  }
}

// The same pass as the producer of a Winograd F(R,3) convolution (conv_x3.hip, WINO): one lane = 8 channels of one GROUP of R
// output columns of one image row.  It evaluates out = act(y * scale + shift (+ res)) on the group's R + 2 input columns
// R*q - 1 .. R*q + R (zero outside the row: the convolution's padding), applies the input transform B^T in fp32 and writes the
// R + 2 components as HL32 lines [image row][group][component][C/32][128 B]; the plain fp32 / HL32 results of the group's own R
// columns are written too when asked for (identity-shortcut source of the next block).  Neighbouring groups re-read two
// columns each (cache hits: they are processed by neighbouring waves).
template <int R>
__global__ __launch_bounds__(256) void bn_apply_wino_grouped_kernel(const float* __restrict__ y, const float* __restrict__ res,
                                                                    const unsigned char* __restrict__ res_hl,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    float* __restrict__ out, unsigned char* __restrict__ out_hl,
                                                                    unsigned char* __restrict__ out_v, long imgrows_per_group, int W,
                                                                    int Wq, long n8, int C, int relu,
                                                                    const float* __restrict__ prescale, int dense) {
  constexpr int NC = R + 2;
  const float ps = prescale ? prescale[0] : 1.f;     // power-of-two range scale of a trained layer's operand (B^T is linear: applied once, after it)
  const int C8 = C >> 3, Cb = C >> 5;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const long t = i / C8;                       // (image row, group)
    const int q = (int)(t % Wq);
    const long irow = t / Wq;                    // image row index over [G][B][H]
    const int g = (int)(irow / imgrows_per_group);
    F8 sc, sh;
    if (scale) {
      sc = load8(scale + (long)g * C + c8 * 8);
      sh = load8(shift + (long)g * C + c8 * 8);
    }
    F8 d[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int x = R * q - 1 + j;
      if ((unsigned)x < (unsigned)W) {
        const long row = irow * W + x;
        F8 v = load8(y + (row * C8 + c8) * 8);
        if (scale) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v.a[e] = v.a[e] * sc.a[e] + sh.a[e];
            v.b[e] = v.b[e] * sc.b[e] + sh.b[e];
          }
        }
        if (res || res_hl) {
          const F8 r = res ? load8(res + (row * C8 + c8) * 8) : load_hl(res_hl, row, C, c8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v.a[e] += r.a[e];
            v.b[e] += r.b[e];
          }
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v.a[e] = relu_nan(v.a[e]);
            v.b[e] = relu_nan(v.b[e]);
          }
        }
        if (j >= 1 && j <= R) {                  // the group's own columns
          if (out) store8(out + (row * C8 + c8) * 8, v);
          if (out_hl) store_hl(out_hl, row, C, c8, v);
        }
        d[j] = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) d[j].a[e] = d[j].b[e] = 0.f;
      }
    }
    F8 m[NC];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      auto D = [&](int j) { return e < 4 ? d[j].a[e] : d[j].b[e - 4]; };
      float r_[NC];
      if constexpr (R == 2) {
        r_[0] = D(0) - D(2);
        r_[1] = D(1) + D(2);
        r_[2] = D(2) - D(1);
        r_[3] = D(1) - D(3);
      } else {
        r_[0] = 4.f * D(0) - 5.f * D(2) + D(4);
        r_[1] = -4.f * (D(1) + D(2)) + D(3) + D(4);
        r_[2] = 4.f * (D(1) - D(2)) - D(3) + D(4);
        r_[3] = 2.f * (D(3) - D(1)) - D(2) + D(4);
        r_[4] = 2.f * (D(1) - D(3)) - D(2) + D(4);
        r_[5] = 4.f * D(1) - 5.f * D(3) + D(5);
      }
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (e < 4) m[k].a[e] = r_[k] * ps;
        else m[k].b[e - 4] = r_[k] * ps;
      }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (dense) store_d16(out_v, t * NC + k, C, c8, m[k], dense);
      else store_hl(out_v, t * NC + k, C, c8, m[k]);
    }
  }
}

// NHWC max pooling over [G*B] images with the (scale, shift, relu) of group b / B fused on the input; padding = -inf
__global__ __launch_bounds__(256) void maxpool_grouped_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ out,
                                                              unsigned char* __restrict__ out_hl, int relu, int B, long n8,
                                                              int H, int W, int C, int Ho, int Wo, int kh, int kw, int sh,
                                                              int sw, int ph, int pw) {
  const int C8 = C >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const long row = i / C8;
    long r = row;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const long b = r / Ho;
    const int g = (int)(b / B);
    F8 sc, sf;
    if (scale) {
      sc = load8(scale + (long)g * C + c8 * 8);
      sf = load8(shift + (long)g * C + c8 * 8);
    }
    F8 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m.a[e] = m.b[e] = -INFINITY;
    for (int ky = 0; ky < kh; ++ky) {
      const int iy = oy * sh - ph + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int ix = ox * sw - pw + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        F8 v = load8(x + (((b * H + iy) * W + ix) * (long)C8 + c8) * 8);
        if (scale) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v.a[e] = v.a[e] * sc.a[e] + sf.a[e];
            v.b[e] = v.b[e] * sc.b[e] + sf.b[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m.a[e] = max_nan(m.a[e], v.a[e]);
          m.b[e] = max_nan(m.b[e], v.b[e]);
        }
      }
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        m.a[e] = relu_nan(m.a[e]);
        m.b[e] = relu_nan(m.b[e]);
      }
    }
    if (out) store8(out + i * 8, m);
    if (out_hl) store_hl(out_hl, row, C, c8, m);
  }
}

// maxpool_grouped_kernel as the producer of a Winograd F(R,3) convolution: one lane = 8 channels of one group of R POOLED columns of
// one pooled row; the R + 2 pooled values (own columns + one halo column each side, zero outside the row) go through B^T and leave as
// HL32 component lines [pooled row][group][component][C/32][128 B], the plain fp32 / HL32 pooled result of the own columns too if asked
template <int R>
__global__ __launch_bounds__(256) void maxpool_wino_grouped_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, float* __restrict__ out,
                                                                   unsigned char* __restrict__ out_hl, unsigned char* __restrict__ out_v,
                                                                   int relu, int B, long n8, int H, int W, int C, int Ho, int Wo, int Wq,
                                                                   int kh, int kw, int sh, int sw, int ph, int pw, int dense) {
  constexpr int NC = R + 2;
  const int C8 = C >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const long t = i / C8;                       // (pooled row over [G*B][Ho], group)
    const int q = (int)(t % Wq);
    const long prow = t / Wq;
    const int oy = (int)(prow % Ho);
    const long b = prow / Ho;
    const int g = (int)(b / B);
    F8 sc, sf;
    if (scale) {
      sc = load8(scale + (long)g * C + c8 * 8);
      sf = load8(shift + (long)g * C + c8 * 8);
    }
    F8 d[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int ox = R * q - 1 + j;
      if ((unsigned)ox < (unsigned)Wo) {
        F8 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m.a[e] = m.b[e] = -INFINITY;
        for (int ky = 0; ky < kh; ++ky) {
          const int iy = oy * sh - ph + ky;
          if ((unsigned)iy >= (unsigned)H) continue;
          for (int kx = 0; kx < kw; ++kx) {
            const int ix = ox * sw - pw + kx;
            if ((unsigned)ix >= (unsigned)W) continue;
            F8 v = load8(x + (((b * H + iy) * W + ix) * (long)C8 + c8) * 8);
            if (scale) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v.a[e] = v.a[e] * sc.a[e] + sf.a[e];
                v.b[e] = v.b[e] * sc.b[e] + sf.b[e];
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              m.a[e] = max_nan(m.a[e], v.a[e]);
              m.b[e] = max_nan(m.b[e], v.b[e]);
            }
          }
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            m.a[e] = relu_nan(m.a[e]);
            m.b[e] = relu_nan(m.b[e]);
          }
        }
        if (j >= 1 && j <= R) {
          const long row = prow * Wo + ox;
          if (out) store8(out + (row * C8 + c8) * 8, m);
          if (out_hl) store_hl(out_hl, row, C, c8, m);
        }
        d[j] = m;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) d[j].a[e] = d[j].b[e] = 0.f;
      }
    }
    F8 m_[NC];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      auto D = [&](int j) { return e < 4 ? d[j].a[e] : d[j].b[e - 4]; };
      float r_[NC];
      if constexpr (R == 2) {
        r_[0] = D(0) - D(2);
        r_[1] = D(1) + D(2);
        r_[2] = D(2) - D(1);
        r_[3] = D(1) - D(3);
      } else {
        r_[0] = 4.f * D(0) - 5.f * D(2) + D(4);
        r_[1] = -4.f * (D(1) + D(2)) + D(3) + D(4);
        r_[2] = 4.f * (D(1) - D(2)) - D(3) + D(4);
        r_[3] = 2.f * (D(3) - D(1)) - D(2) + D(4);
        r_[4] = 2.f * (D(1) - D(3)) - D(2) + D(4);
        r_[5] = 4.f * D(1) - 5.f * D(3) + D(5);
      }
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (e < 4) m_[k].a[e] = r_[k];
        else m_[k].b[e - 4] = r_[k];
      }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (dense) store_d16(out_v, t * NC + k, C, c8, m_[k], dense);
      else store_hl(out_v, t * NC + k, C, c8, m_[k]);
    }
  }
}

// SVTR mixing blocks of G lock-step experts (modules/svtr.py:200-204, :298-305): t = x + drop[r / rows_per_drop] * branch
// (the DropPath-scaled residual add of the PREVIOUS half block; branch NULL: t = x), optionally written back as the new
// fp32 residual stream, then y = LayerNorm(t; gamma[g], beta[g]) with g = r / rows_per_group (gamma NULL: y = t), written
// as fp32 and / or straight as the HL32 operand of the next grouped Linear.  LPR lanes share one row (a 16-byte chunk
// each per pass), so C = 64 / 128 rows still fill the wave.  Same arithmetic as layernorm_fwd_kernel (rowops.hip).
template <int LPR>
__global__ __launch_bounds__(256) void add_layernorm_grouped_kernel(const float* __restrict__ x, const float* __restrict__ branch,
                                                                   const float* __restrict__ drop, long rows_per_drop,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   long rows_per_group, float* __restrict__ sum_out,
                                                                   float* __restrict__ y_f32, unsigned char* __restrict__ y_hl,
                                                                   long rows, int C, float eps) {
  constexpr int RPW = 64 / LPR;                            // rows per wave
  constexpr int NV = 4;                                    // 16-byte chunks per lane: C <= LPR * 16
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const long row = (blockIdx.x * 4L + (threadIdx.x >> 6)) * RPW + lane / LPR;
  if (row >= rows) return;                                 // (whole sub-rows drop out together: shuffles stay inside LPR lanes)
  const int C4 = C >> 2;
  const float* xr = x + row * C;
  f32x4 v[NV];
  float s = 0.f;
  const float ds = (branch && drop) ? drop[row / rows_per_drop] : 1.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = sub + i * LPR;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c4 < C4) {
      v[i] = reinterpret_cast<const f32x4*>(xr)[c4];
      if (branch) {
        const f32x4 b = reinterpret_cast<const f32x4*>(branch + row * C)[c4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = fmaf(ds, b[j], v[i][j]);
      }
      if (sum_out) reinterpret_cast<f32x4*>(sum_out + row * C)[c4] = v[i];
    }
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  if (gamma) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (sub + i * LPR < C4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q / (float)C + eps);
    const long g = row / rows_per_group;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = sub + i * LPR;
      if (c4 < C4) {
        const f32x4 gm = reinterpret_cast<const f32x4*>(gamma + g * C)[c4];
        const f32x4 bt = reinterpret_cast<const f32x4*>(beta + g * C)[c4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = (v[i][j] - mean) * rstd * gm[j] + bt[j];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = sub + i * LPR;
    if (c4 < C4) {
      if (y_f32) reinterpret_cast<f32x4*>(y_f32 + row * C)[c4] = v[i];
      if (y_hl) {
        f16v4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          _Float16 hh, ll;
          split_f16(v[i][j], hh, ll);
          h[j] = hh; l[j] = ll;
        }
        unsigned char* o = y_hl + (row * (C >> 5) + (c4 >> 3)) * 128 + (c4 & 7) * 8;
        *reinterpret_cast<f16v4*>(o) = h;
        *reinterpret_cast<f16v4*>(o + 64) = l;
      }
    }
  }
}

}  // namespace

// Train-mode BatchNorm2d statistics for G BatchNorm modules of C channels at once.  partials: [G][nblk][2][C] from the
// grouped conv epilogue; ptrs: device table [4][G] of device pointers {gamma, beta, running_mean, running_var} (entries
// may be NULL); scale / shift: [G][C].  Same arithmetic as mrn_bn_finalize_f32.
// row chunks of mrn_bn_finalize_grouped_f32 (1: no workspace needed); workspace = G * ceil(C / 32) * chunks * 64 doubles + G * ceil(C / 32)
// 32-bit tickets that the caller zeroed ONCE (the kernel puts them back to zero)
MRN_EXPORT int64_t mrn_bn_finalize_grouped_chunks(int nblk) {
  int64_t z = (nblk + 511) / 512;
  return z < 1 ? 1 : (z > 32 ? 32 : z);
}

MRN_EXPORT int mrn_bn_finalize_grouped_f32(const float* partials, int G, int nblk, int C, int64_t count,
                                           const void* const* ptrs, float momentum, float eps, float* scale, float* shift,
                                           void* chunk_ws, void* tickets, void* stream) {
  MRN_CHECK_ARG(partials && ptrs && scale && shift && G >= 1 && C >= 1 && count >= 1, "mrn_bn_finalize_grouped_f32: bad operands");
  const int Z = (int)mrn_bn_finalize_grouped_chunks(nblk);
  MRN_CHECK_ARG(Z == 1 || (chunk_ws && tickets && (uintptr_t)chunk_ws % 8 == 0),
                "mrn_bn_finalize_grouped_f32: %d partial rows need the chunk workspace and the zeroed tickets", nblk);
  hipLaunchKernelGGL(bn_finalize_grouped_kernel, dim3((C + 31) / 32, G, Z), dim3(1024), 0, (hipStream_t)stream, partials, nblk, C,
                     (long)count, (const float* const*)ptrs, G, momentum, eps, scale, shift, (double*)chunk_ws, (unsigned*)tickets);
  MRN_LAUNCH_CHECK("bn_finalize_grouped");
  return MRN_OK;
}

// Eval-mode BatchNorm2d of G modules as per-channel affines: scale[g] = gamma / sqrt(running_var + eps), shift[g] = beta -
// running_mean * scale, read from the modules' CURRENT buffers through the [4][G] pointer table of mrn_bn_finalize_grouped_f32
// (gamma / beta entries may be NULL; running_mean / running_var must not).  Same arithmetic as mrn_bn_eval_affine_f32.
MRN_EXPORT int mrn_bn_eval_affine_grouped_f32(const void* const* ptrs, int G, int C, float eps, float* scale, float* shift,
                                              void* stream) {
  MRN_CHECK_ARG(ptrs && scale && shift && G >= 1 && C >= 1, "mrn_bn_eval_affine_grouped_f32: bad operands");
  hipLaunchKernelGGL(bn_eval_affine_grouped_kernel, dim3((C + 255) / 256, G), dim3(256), 0, (hipStream_t)stream,
                     (const float* const*)ptrs, G, C, eps, scale, shift);
  MRN_LAUNCH_CHECK("bn_eval_affine_grouped");
  return MRN_OK;
}

// out = act(y * scale[g] + shift[g] (+ residual)) over [G][rows_per_group][C]; scale/shift may be NULL (identity); the
// residual is given as fp32 (residual) or as an HL32 image (residual_hl32: hi + lo is added, 22 significand bits);
// out_f32 and / or out_hl32 (C % 32 == 0) receive the result; out_f32 may alias y.  relu: 0 / 1.
MRN_EXPORT int mrn_bn_apply_grouped_f32(const float* y, const float* residual, const void* residual_hl32, const float* scale,
                                        const float* shift, float* out_f32, void* out_hl32, int G, int64_t rows_per_group,
                                        int C, int relu, void* stream) {
  MRN_CHECK_ARG(y && (out_f32 || out_hl32) && C % 8 == 0 && (!(out_hl32 || residual_hl32) || C % 32 == 0) &&
                    (!scale == !shift) && !(residual && residual_hl32),
                "mrn_bn_apply_grouped_f32: bad operands (C=%d)", C);
  const long n8 = (long)G * rows_per_group * (C / 8);
  if (n8 == 0) return MRN_OK;
  long grid = (n8 + 255) / 256;
  if (grid > 32768) grid = 32768;
  hipLaunchKernelGGL(bn_apply_grouped_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, y, residual,
                     (const unsigned char*)residual_hl32, scale, shift,
                     out_f32, (unsigned char*)out_hl32, (long)rows_per_group, n8, C, relu);
  MRN_LAUNCH_CHECK("bn_apply_grouped");
  return MRN_OK;
}

// mrn_bn_apply_grouped_f32 as the producer of a Winograd F(R,3) convolution (mrn_conv2d_x3_wino_hl32): y [G][B][H][W][C];
// out_wino [G][B][H][ceil(W/R)][R+2][C/32][128 B] receives B^T applied to out = act(y * scale + shift (+ residual)) per group of R
// columns (zero padding outside the row); out_f32 (must NOT alias y: neighbouring groups re-read y) / out_hl32 optionally
// receive the plain result.
static int bn_apply_wino_launch(const float* y, const float* residual, const void* residual_hl32, const float* scale,
                                const float* shift, float* out_f32, void* out_hl32, void* out_wino, int G, int B, int H,
                                int W, int C, int R, int relu, const float* prescale, int dense, void* stream) {
  MRN_CHECK_ARG(y && out_wino && C % (dense ? 64 : 32) == 0 && (!scale == !shift) && !(residual && residual_hl32) && (R == 2 || R == 4) &&
                    out_f32 != y && (uintptr_t)out_wino % 128 == 0,
                "mrn_bn_apply_wino_grouped_f32: bad operands (C=%d R=%d)", C, R);
  const int Wq = (W + R - 1) / R;
  const long n8 = (long)G * B * H * Wq * (C / 8);
  if (n8 == 0) return MRN_OK;
  long grid = (n8 + 255) / 256;
  if (grid > 32768) grid = 32768;
  if (R == 4)
    hipLaunchKernelGGL(bn_apply_wino_grouped_kernel<4>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, y, residual,
                       (const unsigned char*)residual_hl32, scale, shift, out_f32, (unsigned char*)out_hl32, (unsigned char*)out_wino,
                       (long)B * H, W, Wq, n8, C, relu, prescale, dense);
  else
    hipLaunchKernelGGL(bn_apply_wino_grouped_kernel<2>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, y, residual,
                       (const unsigned char*)residual_hl32, scale, shift, out_f32, (unsigned char*)out_hl32, (unsigned char*)out_wino,
                       (long)B * H, W, Wq, n8, C, relu, prescale, dense);
  MRN_LAUNCH_CHECK("bn_apply_wino_grouped");
  return MRN_OK;
}

MRN_EXPORT int mrn_bn_apply_wino_grouped_f32(const float* y, const float* residual, const void* residual_hl32, const float* scale,
                                             const float* shift, float* out_f32, void* out_hl32, void* out_wino, int G, int B, int H,
                                             int W, int C, int R, int relu, const float* prescale, void* stream) {
  return bn_apply_wino_launch(y, residual, residual_hl32, scale, shift, out_f32, out_hl32, out_wino, G, B, H, W, C, R, relu, prescale, 0, stream);
}

// mrn_bn_apply_wino_grouped_f32 for the reduced-precision mode (one fp16 product per term): out_wino_d16 [G][B][H][ceil(W/4)][6][C/64][128 B]
// receives the F(4,3) components as PLAIN fp16, 64 channels per line (mrn_conv2d_x3_wino_d16's activation operand); C % 64 == 0.  The plain
// results (out_f32 / out_hl32: identity-shortcut sources) keep their full-precision forms.
MRN_EXPORT int mrn_bn_apply_wino_grouped_d16_f32(const float* y, const float* residual, const void* residual_hl32, const float* scale,
                                                 const float* shift, float* out_f32, void* out_hl32, void* out_wino_d16, int G, int B, int H,
                                                 int W, int C, int relu, const float* prescale, int bf16, void* stream) {
  return bn_apply_wino_launch(y, residual, residual_hl32, scale, shift, out_f32, out_hl32, out_wino_d16, G, B, H, W, C, 4, relu, prescale, bf16 ? 2 : 1, stream);
}

// MaxPool2d over x [G][B][H][W][C] with the BatchNorm-apply (+ ReLU) of group g fused on the input (scale/shift [G][C] or NULL)
MRN_EXPORT int mrn_maxpool_grouped_f32(const float* x, const float* scale, const float* shift, int relu, float* out_f32,
                                       void* out_hl32, int G, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph,
                                       int pw, void* stream) {
  MRN_CHECK_ARG(x && (out_f32 || out_hl32) && C % 8 == 0 && (!out_hl32 || C % 32 == 0) && (!scale == !shift),
                "mrn_maxpool_grouped_f32: bad operands (C=%d)", C);
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  MRN_CHECK_ARG(Ho > 0 && Wo > 0, "mrn_maxpool_grouped_f32: empty output");
  const long n8 = (long)G * B * Ho * Wo * (C / 8);
  if (n8 == 0) return MRN_OK;
  long grid = (n8 + 255) / 256;
  if (grid > 32768) grid = 32768;
  hipLaunchKernelGGL(maxpool_grouped_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, scale, shift, out_f32,
                     (unsigned char*)out_hl32, relu, B, n8, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw);
  MRN_LAUNCH_CHECK("maxpool_grouped");
  return MRN_OK;
}

// mrn_maxpool_grouped_f32 as the producer of a Winograd F(R,3) convolution: out_wino [G][B][Ho][ceil(Wo/R)][R+2][C/32][128 B]
// receives B^T applied to the pooled map per group of R pooled columns (zero padding outside the row); out_f32 / out_hl32 optionally
// the plain pooled result (C % 32 == 0)
static int maxpool_wino_launch(const float* x, const float* scale, const float* shift, int relu, float* out_f32,
                               void* out_hl32, void* out_wino, int G, int B, int H, int W, int C, int kh, int kw, int sh,
                               int sw, int ph, int pw, int R, int dense, void* stream) {
  MRN_CHECK_ARG(x && out_wino && C % (dense ? 64 : 32) == 0 && (!scale == !shift) && (R == 2 || R == 4) && (uintptr_t)out_wino % 128 == 0,
                "mrn_maxpool_wino_grouped_f32: bad operands (C=%d R=%d)", C, R);
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  MRN_CHECK_ARG(Ho > 0 && Wo > 0, "mrn_maxpool_wino_grouped_f32: empty output");
  const int Wq = (Wo + R - 1) / R;
  const long n8 = (long)G * B * Ho * Wq * (C / 8);
  if (n8 == 0) return MRN_OK;
  long grid = (n8 + 255) / 256;
  if (grid > 32768) grid = 32768;
  if (R == 4)
    hipLaunchKernelGGL(maxpool_wino_grouped_kernel<4>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, scale, shift, out_f32,
                       (unsigned char*)out_hl32, (unsigned char*)out_wino, relu, B, n8, H, W, C, Ho, Wo, Wq, kh, kw, sh, sw, ph, pw, dense);
  else
    hipLaunchKernelGGL(maxpool_wino_grouped_kernel<2>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, scale, shift, out_f32,
                       (unsigned char*)out_hl32, (unsigned char*)out_wino, relu, B, n8, H, W, C, Ho, Wo, Wq, kh, kw, sh, sw, ph, pw, dense);
  MRN_LAUNCH_CHECK("maxpool_wino_grouped");
  return MRN_OK;
}

MRN_EXPORT int mrn_maxpool_wino_grouped_f32(const float* x, const float* scale, const float* shift, int relu, float* out_f32,
                                            void* out_hl32, void* out_wino, int G, int B, int H, int W, int C, int kh, int kw, int sh,
                                            int sw, int ph, int pw, int R, void* stream) {
  return maxpool_wino_launch(x, scale, shift, relu, out_f32, out_hl32, out_wino, G, B, H, W, C, kh, kw, sh, sw, ph, pw, R, 0, stream);
}

// mrn_maxpool_wino_grouped_f32 for the reduced-precision mode: out_wino_d16 [G][B][Ho][ceil(Wo/4)][6][C/64][128 B] plain fp16 (C % 64 == 0)
MRN_EXPORT int mrn_maxpool_wino_grouped_d16_f32(const float* x, const float* scale, const float* shift, int relu, float* out_f32,
                                                void* out_hl32, void* out_wino_d16, int G, int B, int H, int W, int C, int kh, int kw, int sh,
                                                int sw, int ph, int pw, int bf16, void* stream) {
  return maxpool_wino_launch(x, scale, shift, relu, out_f32, out_hl32, out_wino_d16, G, B, H, W, C, kh, kw, sh, sw, ph, pw, 4, bf16 ? 2 : 1, stream);
}

// t = x + drop[r / rows_per_drop] * branch (branch NULL: t = x; drop NULL: 1) -> sum_out (optional, may alias x);
// y = LayerNorm(t) * gamma[g] + beta[g], g = r / rows_per_group (gamma NULL: y = t) -> y_f32 and / or y_hl32 (C % 32 == 0).
// x, branch, sum_out, y_f32: [rows][C] contiguous; gamma, beta: [G][C].  Replaces the residual add + DropPath scale + LayerNorm
// + operand split chain between two Linear layers of an SVTR block (modules/svtr.py:200-204) by one pass.
MRN_EXPORT int mrn_add_layernorm_grouped_f32(const float* x, const float* branch, const float* drop, int64_t rows_per_drop,
                                             const float* gamma, const float* beta, int64_t rows_per_group, float* sum_out,
                                             float* y_f32, void* y_hl32, int64_t rows, int C, float eps, void* stream) {
  MRN_CHECK_ARG(x && (y_f32 || y_hl32 || sum_out) && C % 4 == 0 && C <= 1024 && (!y_hl32 || C % 32 == 0) && (!gamma == !beta) &&
                    rows_per_group >= 1 && (!drop || rows_per_drop >= 1),
                "mrn_add_layernorm_grouped_f32: bad operands (C=%d)", C);
  if (rows == 0) return MRN_OK;
  const hipStream_t st = (hipStream_t)stream;
  unsigned char* hl = (unsigned char*)y_hl32;
#define MRN_ALN_LAUNCH(LPR)                                                                                                   \
  hipLaunchKernelGGL(add_layernorm_grouped_kernel<LPR>, dim3((unsigned)((rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)))),     \
                     dim3(256), 0, st, x, branch, drop, (long)rows_per_drop, gamma, beta, (long)rows_per_group, sum_out, y_f32,  \
                     hl, (long)rows, C, eps)
  if (C <= 64) MRN_ALN_LAUNCH(16);
  else if (C <= 128) MRN_ALN_LAUNCH(32);
  else MRN_ALN_LAUNCH(64);
#undef MRN_ALN_LAUNCH
  MRN_LAUNCH_CHECK("add_layernorm_grouped");
  return MRN_OK;
}
