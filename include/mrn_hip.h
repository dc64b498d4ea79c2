/*
 * mrn_hip.h -- C ABI of libmrn_hip.so, the MI355X (gfx950) kernel library for MRN's per-step
 * recognition-and-routing training path.
 *
 * The reference (simplify23/MRN) has no FFI: every op on the path is a PyTorch call.  Each entry point below
 * names the reference op site (file:line under the reference tree) it replaces; INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add at that site.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *   - the caller owns every buffer; the library allocates nothing and keeps no mutable global state (one exception: the RCCL
 *     communicator created by an explicit mrn_comm_init)
 *   - return 0 on success, <0 library error, >0 hipError_t; text via mrn_last_error() (thread-local)
 *   - activations are fp32, NHWC (channels innermost); "rows" means all leading dims flattened
 * Compiled-in limits (every shipped config of the reference sits inside them; a call outside returns an error, never a wrong
 * result): recurrent kernels (LSTM layer, attention decoder) hidden_size == 256 (config/*.py: hidden_size=256), decoder context
 * width D a multiple of 16 (forward) / of 256 (backward: DERNet concatenates 256-wide extractors); router fan-in / gate tail at most
 * 8 experts (the reference trains 6 languages); TPS at most 61 fiducials (reference: 20); grouped-conv kernels Cin % 32 == 0 and
 * kh*kw <= 32 (other layers run on the exact-fp32 kernels, Cin % 4 == 0).
 */
#ifndef MRN_HIP_H
#define MRN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int mrn_version(void);
const char* mrn_last_error(void);

/* ---- GEMM / convolution (exact-fp32 MFMA) ------------------------------------------------------------ */

/* C[b][m][n] = act(alpha * sum_k A[b][m][k] * W[b][n][k] + bias) (+ residual) (+ C when accumulate).
 * Any strides; bias_axis 0 = per n, 1 = per m; act 0 none, 1 relu, 2 gelu(erf).
 * Replaces nn.Linear / torch.bmm sites: modules/sequence_modeling.py:10, modules/prediction.py:58-68,104-107,
 * modules/dm_router.py:41-46 and :9,24 (token / channel mixing), modules/model.py:151,437-438,
 * modules/transformation.py:86-87; also every weight/data gradient GEMM of those layers. */
int mrn_gemm_f32(const float* A, const float* W, const float* bias, const float* residual, float* C,
                 int M, int N, int K, int batch,
                 int64_t sAb, int64_t sAm, int64_t sAk, int64_t sWb, int64_t sWn, int64_t sWk,
                 int64_t sCb, int64_t sCm, int64_t sCn, int64_t sBiasB, int bias_axis,
                 int act, int accumulate, float alpha, void* stream);

/* y = act(conv2d(x, w) + bias), x NHWC [B][H][W][Cin] (Cin % 4 == 0), w [Cout][kh][kw][Cin], y NHWC.
 * When stats != NULL it receives mrn_conv2d_stats_floats() floats: per 128-row block, per channel, the sum and
 * sum of squares of the pre-activation output (the BatchNorm batch statistics, fused into the conv epilogue).
 * Replaces nn.Conv2d sites: modules/feature_extraction.py:19-44 (VGG), :214-294 (ResNet),
 * modules/transformation.py:60-84 (TPS localization net). */
int mrn_conv2d_nhwc_f32(const float* x, const float* w_ohwi, const float* bias, float* y, float* stats,
                        int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                        int act, void* stream);
int64_t mrn_conv2d_stats_floats(int B, int Ho, int Wo, int Cout);

/* Power-of-two range scale of a split-fp16 operand: scale = device float[2] = {s, 1/s}, s the largest power of two with
 * s * max|w| <= target (weights of the frozen experts: target 2^14; both operands of a trained layer per call).  Computed on the
 * device, no host synchronisation.  workspace: two 32-bit device words owned by the calling stream. */
int mrn_pow2_scale_f32(const float* w, int64_t n, float target, float* scale, void* workspace, void* stream);
/* the same scale from maxima a producer pass already folded into `workspace` (amax_ws of mrn_scale_shift_act_f32 /
 * mrn_bn_bwd_apply_f32: 64 32-bit words zeroed once, slot = block % 64): saves the extra read of the tensor; the words are put back to zero */
int mrn_pow2_finalize_f32(float target, float* scale, void* workspace, void* stream);

/* Grouped split-fp16 x3 convolution on 256-wide tiles (the router phase runs the G frozen experts' backbones in
 * lock-step: il_modules/mrn.py:323-337 calls every expert on the same batch; modules/model.py:399-401).
 * Operands are in the "HL32" layout: one 128-byte line [hi fp16 x 32 | lo fp16 x 32] per (pixel, 32-channel block) for
 * the activation (mrn_split_hl32_f32, or the fused BatchNorm-apply pass mrn_bn_apply_hl32_f32) and per
 * (Cout row, 32-channel block, tap) for the weight (mrn_pack_weight_hl32; reduction order = channel block outer, tap
 * inner).  x_hl: [G][B][H][W][Cin/32][128 B] with group stride x_group_stride_bytes (0 = all groups read the same
 * input); w_hl: [G][Cout][Cin/32][kh*kw][128 B]; bias [G][Cout] or NULL; out_scale [G][2] = {s, 1/s} per group (the
 * power-of-two weight prescale of mrn_pow2_scale_f32) or NULL; x_scale: {s, 1/s} of an activation that was split as
 * s * x (mrn_split_hl32_f32 with a scale; keeps small-magnitude operands such as gradients inside fp16's normal range) or
 * NULL; residual: optional fp32 tensor with y's layout, added before the activation; y [G][B][Ho][Wo][Cout] fp32; stats
 * [G][ceil(B*Ho*Wo/tile_m)][2][Cout] per-row-block sums / sums of squares (mrn_conv2d_x3_stats_floats) or NULL.
 * tile_m x tile_n = 256x256 (Cout >= 256), 256x128, 256x64 (Cout <= 64) or 128x128.  Requires Cin % 32 == 0; zero_page: >= 128 bytes of device zeros.
 * y_row_stride / y_group_stride (floats; 0 = dense [G][B*Ho*Wo][Cout]) let the result land in a wider buffer, e.g. one
 * expert's slice of the router's [B][P][I][C] feature tensor.  x_group_div > 1: group g reads activation group g / x_group_div.  With H = W = kh = kw = 1 this is a grouped Linear layer
 * (nn.Linear sites of modules/sequence_modeling.py:10,19-22 and modules/prediction.py:58-68,104-107).
 * y_hl32 (optional, Cout % 32 == 0, dense rows): the result also (y != NULL) or only (y == NULL) as the HL32 operand of the
 * next GEMM -- fc1 + GELU -> fc2 of the SVTR Mlp (modules/svtr.py:46-67) without an operand-split pass in between.
 * products: 3 = split-fp16 x3 (22-bit products, the 1e-4 parity mode); 1 = hi x hi only: plain fp16 products with fp32
 * accumulation, the reduced-precision mode BASELINE configs 2 ("bf16") and 5 ("fp16 MFMA") ask for.
 * ch_scale / ch_shift [G][Cout] (a pair, or both NULL): per-channel affine applied after the bias -- BatchNorm2d in EVAL mode
 * (running statistics, mrn_bn_eval_affine_f32) folded into the epilogue, so a frozen eval-mode Conv2d -> BatchNorm2d -> (+ identity)
 * -> ReLU (modules/feature_extraction.py:184-199) is ONE launch; residual_hl32: the identity as HL32 lines (needs y_hl32). */
int mrn_conv2d_x3_hl32(const void* x_hl, const void* w_hl, const void* zero_page, const float* bias,
                       const float* residual, float* y, float* stats, const float* out_scale, const float* x_scale, int G, int64_t x_group_stride_bytes, int B, int H, int W,
                       int Cin, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int act, int tile_m, int tile_n,
                       int64_t y_row_stride, int64_t y_group_stride, int x_group_div, void* y_hl32, int products, const float* ch_scale,
                       const float* ch_shift, const void* residual_hl32, void* amax_ws, void* stream);
/* amax_ws (optional): max|y| of the stored result folded into 64 words (mrn_pow2_finalize_f32 turns them into {s, 1/s}): the range of the
 * NEXT trained layer's operand without a pass over y (qkv -> attention -> proj, fc1 -> GELU -> fc2 in loop A) */
int64_t mrn_conv2d_x3_stats_floats(int G, int B, int Ho, int Wo, int Cout, int tile_m);
int mrn_split_hl32_f32(const float* x, void* out, int64_t rows, int C, const float* scale, void* stream);
/* transposed split for weight-gradient GEMMs (dW = dy^T x reduces over rows): x[rows][C] -> [splits][C][rows/splits/32][128 B],
 * i.e. `splits` HL32 matrices whose reduction axis is a row chunk (split-K = the conv's group dimension) */
int mrn_split_hl32_t_f32(const float* x, void* out, int64_t rows, int64_t rows_padded, int C, int splits, const float* scale,
                         void* stream);
/* mrn_split_hl32_t_f32 of dy that also leaves the bias gradient colsum[c] (+)= sum_r dy[r][c] of the Linear layer whose weight gradient the
 * transposed operand feeds (loss.backward() through nn.Linear, il_modules/mrn.py:260-261; modules/svtr.py:46-152): partial =
 * mrn_split_hl32_t_colsum_chunks(rows_padded, C) * C floats of scratch (<= 256 chunk rows, finished by one column-sum pass in a fixed
 * order: deterministic). */
int64_t mrn_split_hl32_t_colsum_chunks(int64_t rows_padded, int C);
int mrn_split_hl32_t_colsum_f32(const float* x, void* out, int64_t rows, int64_t rows_padded, int C, int splits, const float* scale,
                                float* colsum, int accumulate, float* partial, void* stream);
/* transposed im2col for the convolution weight gradient (loss.backward() through Conv2d, il_modules/mrn.py:260-261):
 * out[s][tap][ci][rows_padded/splits/32][128 B], element (tap, ci, p) = scale * x[pixel(p) shifted by tap][ci] (0 in the
 * padding).  With mrn_split_hl32_t_f32(dy) as the other operand, dW = mrn_conv2d_x3_hl32 with groups = splits * taps and
 * x_group_div = taps (the activation group of the GEMM is the dy^T chunk g / taps). */
int mrn_im2col_t_hl32_f32(const float* x, void* out, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                          int64_t rows_padded, int splits, const float* scale, void* stream);
int mrn_pack_weight_hl32(const float* w_ohwi, void* out, int Cout, int taps, int Cin, const float* scale, void* stream);
/* The weight operands of ALL trained Linear layers for the coming step in one call (what `optimizer.step()` invalidates every iteration,
 * il_modules/mrn.py:262; nn.Linear products of modules/svtr.py:46-152): desc = n x 8 int64 on the device {W fp32 [N][K], out, scale
 * float[2], N, K, transposed, first_tile, tiles_i}; out = HL32 [O][I/32][128 B] of s * W (O = N, I = K) or s * W^T (O = K, I = N), s the
 * power-of-two prescale of mrn_pow2_scale_f32 for `target`, written to scale; tiles = sum of ceil(O/32) * (I/32), first_tile its running
 * sum; amax: n words of scratch.  N % 4 == 0, K % 4 == 0, I % 32 == 0. */
int mrn_multi_pack_linear_hl32(const void* desc, int n, int64_t tiles, void* amax, float target, void* stream);
/* Convolution weight gradient of a 3x3 / stride 1 / pad 1 conv WITHOUT an im2col (loss.backward() through Conv2d, il_modules/mrn.py:260-261):
 * mrn_transpose_oy_hl32_f32 writes x^T (or dy^T) as an HL32 matrix [C][ceil(B*H*W/32)][128 B] in image-row-major pixel order
 * k = (y*B + b)*W + xx, shifted by shift_x along x with zero fill -- in that order a kernel-row offset is a whole number of lines
 * and only the three horizontal shifts need their own copy (3x the activation instead of 9x);
 * mrn_gemm_x3_windows_hl32 is the grouped split-fp16 x3 GEMM over K-WINDOWS of two such matrices: group g multiplies the window
 * {a_off_bytes, w_off_bytes, n_lines} (device array of G records {int64, int64, int32, int32}) of every row pair,
 * y [G][M][N]; one group per (split-K chunk, tap). */
int mrn_transpose_oy_hl32_f32(const float* x, void* out, int B, int H, int W, int C, int shift_x, const float* scale, void* stream);
int mrn_transpose_oy3_hl32_f32(const float* x, void* out, int B, int H, int W, int C, const float* scale, void* stream);   /* shifts -1, 0, +1 in one pass: out [3][C][lines][128 B] */
/* ... and in the Winograd domain (F(4,3) along W): both operands transformed per group of 4 columns while they are transposed
 * (mode 0: the layer input through B^T, mode 1: the output gradient through A; out [6][C][ceil(B*H*ceil(W/4)/32)][128 B] in
 * image-row-major group order), 18 K-windows (component x kernel row) of a quarter of the length on mrn_gemm_x3_windows_hl32, then
 * dW[ky][kx] = sum_m G[m][kx] dU_m[ky] over the split-K slabs (mrn_wino_wgrad_finish_f32): half the matrix work of the 9-window form */
int mrn_transpose_oy_wino_hl32_f32(const float* t, void* out, int B, int H, int W, int C, int mode, const float* scale, void* stream);
int mrn_wino_wgrad_finish_f32(const float* part, float* dw, int S, int Cout, int Cin, int oihw_accumulate, void* stream);
/* oihw_accumulate 0: dw [Cout][3][3][Cin] written; 1: ADDED into dw [Cout][Cin][3][3], the Conv2d parameter's own gradient layout
 * (.grad accumulation of loss.backward() without a layout pass); out_scale / x_scale of the windows GEMM: ONE {s, 1/s} pair per operand */
int mrn_gemm_x3_windows_hl32(const void* a_hl, int64_t a_bytes, int a_pitch_lines, const void* w_hl, int64_t w_bytes,
                             int w_pitch_lines, const void* windows, int G, int M, int N, const void* zero_page,
                             const float* out_scale, const float* x_scale, float* y, int tile_m, int tile_n, int products,
                             void* stream);   /* products: 3 or 1, as in mrn_conv2d_x3_hl32 */

/* The 3x3 / stride 1 / pad 1 convolutions of the frozen experts' deep ResNet stages (16 of TRBA's 29, the 512 -> 512 layers on
 * 4 x 65 maps: modules/feature_extraction.py:165-199,262-294) as 1-D Winograd F(R,3) along W on the same split-fp16 x3 kernel:
 * R + 2 products per R output columns instead of 3R (R = 4: half the matrix work of the direct form; R = 2: two thirds).
 *   mrn_pack_weight_wino_hl32      w [Cout][3][3][Cin] fp32 -> U [Cout][R+2][Cin/32][3 (ky)][128 B] = scale * G g per kernel
 *                                  row (sum in double; the power-of-two row scales of G are undone by the kernel's A^T);
 *   mrn_bn_apply_wino_grouped_f32  BatchNorm-apply (+ residual + ReLU) of the previous layer, as mrn_bn_apply_grouped_f32, writing
 *                                  V [G][B][H][ceil(W/R)][R+2][C/32][128 B] = B^T applied to every group of R columns (+ halo,
 *                                  zero outside the row) and optionally the plain fp32 (not aliasing y) / HL32 result;
 *   mrn_conv2d_x3_wino_hl32        y [G][B][H][W][Cout] = A^T [ sum over (ky, Cin) of U_m V_m ] + bias (act 0 / 1), BatchNorm
 *                                  partial statistics [G][ceil(B*H*ceil(W/R)/128)][2][Cout] (mrn_conv2d_x3_wino_stats_floats) or
 *                                  NULL; out_scale [G][2] = the weight prescale {s, 1/s}; v_group_stride_bytes 0 = shared input.
 * Trained layers (loop A, il_modules/mrn.py:225-279: forward and data-gradient convolutions) run the same three entry points with
 * G = 1 and per-operand power-of-two range scales: `prescale` of the producer pass = `x_scale` of the convolution = {s, 1/s} of the
 * activation (or gradient) operand from mrn_pow2_scale_f32. */
int mrn_pack_weight_wino_hl32(const float* w_ohwi, void* out, int Cout, int Cin, int R, const float* scale, void* stream);
int mrn_bn_apply_wino_grouped_f32(const float* y, const float* residual, const void* residual_hl32, const float* scale,
                                  const float* shift, float* out_f32, void* out_hl32, void* out_wino, int G, int B, int H, int W,
                                  int C, int R, int relu, const float* prescale, void* stream);
/* ... and MaxPool2d (mrn_maxpool_grouped_f32: BatchNorm-apply + ReLU fused on the input) as such a producer: the pooled map's groups of
 * R columns through B^T -> out_wino [G][B][Ho][ceil(Wo/R)][R+2][C/32][128 B]; the plain pooled fp32 / HL32 result optionally */
int mrn_maxpool_wino_grouped_f32(const float* x, const float* scale, const float* shift, int relu, float* out_f32, void* out_hl32,
                                 void* out_wino, int G, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                                 int R, void* stream);
int64_t mrn_conv2d_x3_wino_stats_floats(int G, int B, int H, int W, int Cout, int R);
/* which kernel mrn_conv2d_x3_wino_hl32 runs: 1 = the row-block kernel (csrc/conv_wino.hip: R = 4, H % 4 == 0, Cout % 4 == 0 -- a workgroup
 * owns 64 positions x 4 output rows x 64 channels and stages every input row once per (component, channel block)), 0 = the x3 kernel's
 * Winograd form (one output row per tile).  Same operands and results either way (telemetry, statistics-buffer sizing). */
int64_t mrn_conv2d_x3_wino_rows(int H, int R, int Cout);
/* process-wide A/B switch of that choice: -1 default (row-block kernel unless MRN_WINO_ROWS=0), 0 always the x3 kernel's form, 1 the
 * row-block kernel wherever it applies; size statistics buffers after the call */
int mrn_conv2d_x3_wino_select(int mode);
int mrn_conv2d_x3_wino_hl32(const void* v_hl, const void* u_hl, const void* zero_page, const float* bias, float* y, float* stats,
                            const float* out_scale, const float* x_scale, int G, int64_t v_group_stride_bytes, int B, int H, int W,
                            int Cin, int Cout, int R, int act, void* stream);
/* the same convolution with the 2x2 / stride-2 max-pool that follows BatchNorm + ReLU (feature_extraction.py:229,242; transformation.py:
 * 70-75) taken in the epilogue of the row-block kernel (needs mrn_conv2d_x3_wino_rows(H, R, Cout) != 0 and even H, W; otherwise
 * MRN_ERR_UNSUPPORTED): y [G][B][H/2][W/2][Cout] = per window and channel the maximum of the raw output where the BatchNorm weight
 * (bn_gamma_ptrs: device table of G device addresses, NULL: all maxima) is >= 0, the minimum elsewhere; statistics over the full map */
int mrn_conv2d_x3_wino_pool_hl32(const void* v_hl, const void* u_hl, const void* zero_page, const float* bias, float* y, float* stats,
                                 const float* out_scale, const float* x_scale, int G, int64_t v_group_stride_bytes, int B, int H, int W,
                                 int Cin, int Cout, int R, int act, const void* bn_gamma_ptrs, void* stream);

/* Reduced-precision mode (BASELINE configs 2 "bf16" and 5 "fp16 MFMA"; bench.py --precision fp16; never the parity path): the same
 * F(4,3) row-block convolution of feature_extraction.py:165-199,262-294 with ONE fp16 product per term (fp32 accumulate, fp32 results)
 * on PLAIN fp16 operands, 64 channels per 128-byte line -- a third of the MFMAs on half the operand bytes of the split form:
 *   mrn_pack_weight_wino_d16            w [Cout][3][3][Cin] -> [Cout][6][Cin/64][3][128 B] = fp16(scale * G g)
 *   mrn_bn_apply_wino_grouped_d16_f32   as mrn_bn_apply_wino_grouped_f32 (R = 4), V [G][B][H][ceil(W/4)][6][C/64][128 B] in fp16; the
 *   mrn_maxpool_wino_grouped_d16_f32    plain fp32 / HL32 by-products (identity-shortcut sources) keep full precision
 *   mrn_conv2d_x3_wino_d16              y, stats, out_scale, x_scale, v_group_stride_bytes as mrn_conv2d_x3_wino_hl32; pool != 0: the
 *                                       pooled epilogue of mrn_conv2d_x3_wino_pool_hl32.  H % 4 == 0 and Cin % 64 == 0, else
 *                                       MRN_ERR_UNSUPPORTED (there is no fallback kernel for this layout).
 * bf16 != 0 (all four: producer and consumer must agree): the 16-bit operands are bfloat16 on v_mfma_f32_32x32x16_bf16 -- the literal "bf16" of
 * BASELINE config 2, kept as a comparison instantiation (bench.py --precision bf16): same speed, 8 significand bits instead of fp16's 11. */
int mrn_pack_weight_wino_d16(const float* w_ohwi, void* out, int Cout, int Cin, const float* scale, int bf16, void* stream);
int mrn_bn_apply_wino_grouped_d16_f32(const float* y, const float* residual, const void* residual_hl32, const float* scale,
                                      const float* shift, float* out_f32, void* out_hl32, void* out_wino_d16, int G, int B, int H,
                                      int W, int C, int relu, const float* prescale, int bf16, void* stream);
int mrn_maxpool_wino_grouped_d16_f32(const float* x, const float* scale, const float* shift, int relu, float* out_f32, void* out_hl32,
                                     void* out_wino_d16, int G, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph,
                                     int pw, int bf16, void* stream);
int mrn_conv2d_x3_wino_d16(const void* v_d16, const void* u_d16, const float* bias, float* y, float* stats, const float* out_scale,
                           const float* x_scale, int G, int64_t v_group_stride_bytes, int B, int H, int W, int Cin, int Cout, int act,
                           int pool, const void* bn_gamma_ptrs, int bf16, void* stream);

/* First convolution of the frozen experts' stacks (3x3, stride 1, padding 1, Cin = 4, Cout = 32 or 64: VGG conv 0
 * feature_extraction.py:19, ResNet conv0_1 :214, TPS localisation conv 1 transformation.py:60), G experts in one launch on the
 * exact-fp32 MFMA; x [Gx][B][H][W][4] with x_group_stride floats between groups (0: all experts read the same crops), w
 * [G][Cout][3][3][4], bias [G][Cout] or NULL, y [G][B][H][W][Cout], stats [G][mrn_conv3x3_c4_stats_blocks][2][Cout] (BatchNorm
 * partial sums / sums of squares of the pre-activation result) or NULL; act 0 / 1 (ReLU).  pool = 1 (even H, W): the 2x2 / stride-2
 * max-pool behind BatchNorm + ReLU (feature_extraction.py:20, transformation.py:61-62) is taken in the epilogue as in
 * mrn_conv3x3_patch_x3_hl32 below: y is [G][B][H/2][W/2][Cout], per window and channel the maximum where the BatchNorm weight
 * (bn_gamma_ptrs: device table of G device addresses, NULL: all maxima) is >= 0, the minimum elsewhere. */
int64_t mrn_conv3x3_c4_stats_blocks(int B, int H, int W);
int mrn_conv3x3_c4_grouped_f32(const float* x, const float* w_ohwi, const float* bias, float* y, float* stats, int G,
                               int64_t x_group_stride, int B, int H, int W, int Cout, int act, int pool, const void* bn_gamma_ptrs,
                               void* stream);

/* Patch-resident, weight-stationary 3x3 / stride 1 / pad 1 convolution of the narrow early layers of G lock-step experts
 * ((Cin, Cout) = (32, 64): ResNet conv0_2, feature_extraction.py:216-218; (64, 128): layer1[0].conv1 :171-199 and conv 2 of the TPS
 * localisation network, transformation.py:63-66) as split-fp16 x3 products: x_hl HL32 lines [Gx][B][H][W][Cin/32][128 B]
 * (x_group_stride_bytes 0: one shared input), w_hl / w_scale as for mrn_conv2d_x3_hl32, bias [G][Cout] or NULL, y
 * [G][B][H][W][Cout], stats [G][mrn_conv3x3_patch_stats_blocks][2][Cout] or NULL, act 0 / 1 (ReLU).
 * pool = 1 (even H, W): the 2x2 / stride-2 max-pool that follows BatchNorm + ReLU (feature_extraction.py:219, transformation.py:62-66)
 * is taken in the epilogue: y is [G][B][H/2][W/2][Cout] and holds, per window and channel, the MAXIMUM of the raw convolution output
 * where the BatchNorm weight is >= 0 and the MINIMUM where it is negative (bn_gamma_ptrs: device table of G device addresses of the
 * weights; NULL: every channel keeps its maximum) -- BatchNorm is monotone per channel, so applying scale / shift / ReLU to this map
 * (mrn_bn_apply[_wino]_grouped_f32) equals max-pooling the applied full map bit for bit; the statistics cover the full map. */
int64_t mrn_conv3x3_patch_supported(int Cin, int Cout);
int64_t mrn_conv3x3_patch_stats_blocks(int G, int B, int H, int W, int Cin);
int mrn_conv3x3_patch_x3_hl32(const void* x_hl, const void* w_hl, const float* w_scale, const float* bias, const void* bn_gamma_ptrs,
                              float* y, float* stats, int G, int64_t x_group_stride_bytes, int B, int H, int W, int Cin, int Cout,
                              int act, int pool, void* stream);
/* ... and with ONE fp16 product per term (hi x hi: the reduced-precision mode of BASELINE configs 2 / 5; same operands, the lo halves unread) */
int mrn_conv3x3_patch_x1_hl32(const void* x_hl, const void* w_hl, const float* w_scale, const float* bias, const void* bn_gamma_ptrs,
                              float* y, float* stats, int G, int64_t x_group_stride_bytes, int B, int H, int W, int Cin, int Cout,
                              int act, int pool, void* stream);

/* Elementwise passes between grouped convolutions (G frozen experts in lock-step).
 * mrn_bn_finalize_grouped_f32: train-mode BatchNorm2d statistics for G modules at once; partials [G][nblk][2][C] from
 *   the conv epilogue; ptrs = device table [4][G] of device pointers {gamma, beta, running_mean, running_var} (NULL
 *   entries allowed); scale / shift [G][C].  Same arithmetic as mrn_bn_finalize_f32.
 * mrn_bn_apply_grouped_f32: out = relu?(y * scale[g] + shift[g] (+ residual)) over [G][rows_per_group][C] (residual as
 *   fp32, or as the HL32 image the identity shortcut's source already exists in: hi + lo is added); writes the
 *   fp32 tensor (out_f32, may alias y) and / or the HL32 operand of the next convolution (out_hl32, C % 32 == 0).
 * mrn_maxpool_grouped_f32: the same with MaxPool2d (padding = -inf) applied after the affine + ReLU.
 * modules/feature_extraction.py:171-199,222-294; modules/transformation.py:69-81. */
int mrn_bn_finalize_grouped_f32(const float* partials, int G, int nblk, int C, int64_t count, const void* const* ptrs,
                                float momentum, float eps, float* scale, float* shift, void* chunk_ws, void* tickets, void* stream);
/* the partial rows are reduced in mrn_bn_finalize_grouped_chunks(nblk) chunks by as many workgroups per (expert, 32 channels); with more
 * than one chunk: chunk_ws = G * ceil(C / 32) * chunks * 64 doubles of scratch, tickets = G * ceil(C / 32) 32-bit words zeroed ONCE by the
 * caller (the last workgroup to arrive adds the chunk sums in chunk order -- deterministic -- and puts its ticket back to zero) */
int64_t mrn_bn_finalize_grouped_chunks(int nblk);
/* eval-mode BatchNorm2d (modules/feature_extraction.py:171-197 under model.eval()) of G modules folded to per-channel affines
 * scale / shift [G][C] from the modules' CURRENT running statistics, through the same [4][G] pointer table */
int mrn_bn_eval_affine_grouped_f32(const void* const* ptrs, int G, int C, float eps, float* scale, float* shift, void* stream);
int mrn_bn_apply_grouped_f32(const float* y, const float* residual, const void* residual_hl32, const float* scale,
                             const float* shift, float* out_f32, void* out_hl32, int G, int64_t rows_per_group, int C,
                             int relu, void* stream);
int mrn_maxpool_grouped_f32(const float* x, const float* scale, const float* shift, int relu, float* out_f32,
                            void* out_hl32, int G, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                            void* stream);

/* Convolution backward (loss.backward() through Conv2d, il_modules/mrn.py:260-261):
 *   data gradient  = mrn_conv2d_nhwc_* of dy (zero-dilated by the stride, mrn_dilate_nhwc_f32) with the flipped /
 *                    transposed weight from mrn_pack_dgrad_weight_f32 and padding (k-1-p);
 *   weight gradient = mrn_conv2d_wgrad_f32: split-K partial slabs [splits][Cout][kh*kw*Cin] (column-sum them),
 *                    then mrn_unpack_conv_weight_f32 back to the parameter's [O][I][kh][kw] layout. */
int mrn_conv2d_wgrad_f32(const float* dy, const float* x, float* dw_partial, int B, int H, int W, int Cin, int Cout,
                         int kh, int kw, int sh, int sw, int ph, int pw, int splits, void* stream);
int mrn_pack_dgrad_weight_f32(const float* w_ohwi, float* wt_ihwo, int O, int I, int kh, int kw, void* stream);
int mrn_unpack_conv_weight_f32(const float* g_ohwi, float* g_oihw, int O, int I, int kh, int kw, int accumulate,
                               void* stream);
/* zero-insertion of a strided conv's output gradient: out [B][(Ho-1)*sh+1+extra_h][(Wo-1)*sw+1+extra_w][C]; the extra trailing
 * zero rows / columns cover the input rows the floor in the output-size formula left over (they still sit under kernel taps) */
int mrn_dilate_nhwc_f32(const float* dy, float* out, int B, int Ho, int Wo, int C, int sh, int sw, int extra_h, int extra_w,
                        void* stream);
/* BatchNorm2d (training) backward with the ReLU mask fused: g = dz * (z > 0); partial per-channel sums of g and
 * g*xhat (reduce), then dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) and dres = g (apply). */
int64_t mrn_bn_bwd_blocks(int64_t rows);
int mrn_bn_bwd_reduce_f32(const float* dz, const float* z, const void* zmask, const float* y, const float* mean, const float* invstd,
                          float* partials, int64_t rows, int C, int relu, void* stream);
/* zmask (optional, instead of z): the ReLU mask as 4 bits per 4 consecutive channels (one byte per 16-byte quad, bit j = element j > 0),
 * written by mrn_scale_shift_act_f32(pos_mask) in the forward pass -- 1/32 of the bytes of reading z again in both passes.
 * mrn_bn_bwd_finalize_f32: sums [2][C] = column sums of the partials; dgamma_acc / dbeta_acc (optional): the BatchNorm weight / bias
 * gradients are ADDED there by the same launch (il_modules/mrn.py:260-261 loss.backward(): .grad accumulation). */
int mrn_bn_bwd_finalize_f32(const float* partials, int64_t nblk, int C, float* sums, float* dgamma_acc, float* dbeta_acc, void* stream);
int mrn_bn_bwd_apply_f32(const float* dz, const float* z, const void* zmask, const float* y, const float* mean, const float* invstd,
                         const float* gamma, const float* sums, float* dy, float* dres, int64_t rows, int C, int relu,
                         void* amax_ws, void* stream);   /* amax_ws (optional): max|dy| folded in */
/* MaxPool2d backward: dx (zero-initialised) += dy at the first maximum of each window.  When mrn_maxpool_bwd_writes_all(...) is 1
 * (non-overlapping windows that tile the map: kernel == stride, no padding, H % kh == 0, W % kw == 0, C % 4 == 0) every element of dx is
 * WRITTEN by the call -- no atomics, and the caller may skip the zero fill. */
int64_t mrn_maxpool_bwd_writes_all(int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw);
int mrn_maxpool_bwd_nhwc_f32(const float* dy, const float* x, float* dx_zeroed, int B, int H, int W, int C, int kh, int kw,
                             int sh, int sw, int ph, int pw, void* stream);

/* conv weight repack [O][I][kh][kw] (state_dict layout) -> [O][kh][kw][I] */
int mrn_pack_conv_weight_f32(const float* w_oihw, float* w_ohwi, int O, int I, int kh, int kw, void* stream);

/* ---- layout, BatchNorm, pooling ------------------------------------------------------------------------ */

/* [B][C][H][W] -> [B][H][W][C]; the data loader's tensor contract is NCHW (data/dataset.py:235-246). */
int mrn_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int H, int W, void* stream);

/* Train-mode BatchNorm2d statistics: reduce conv-epilogue partials to mean / biased var, emit the folded
 * (scale, shift), update running stats (momentum form, unbiased var) -- torch.nn.BatchNorm2d semantics.
 * modules/feature_extraction.py:34,39,171-197,222-294; modules/transformation.py:69-81. */
int mrn_bn_finalize_f32(const float* partials, int nblk, int C, int64_t count, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                        float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
/* eval-mode BatchNorm folded to (scale, shift) from running statistics */
int mrn_bn_eval_affine_f32(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, int C, float* scale, float* shift, void* stream);
/* y = act(x * scale[c] + shift[c] + residual): BatchNorm apply + residual add + activation in one pass
 * (BasicBlock tail, modules/feature_extraction.py:184-199; relu = 1 ReLU, 2 GELU for SVTR's PatchEmbed). In place allowed. */
int mrn_scale_shift_act_f32(const float* x, const float* residual, float* y, const float* scale,
                            const float* shift, int64_t rows, int C, int relu, void* amax_ws, void* pos_mask, void* stream);
/* amax_ws (optional): max|y| folded in, see mrn_pow2_finalize_f32; pos_mask (optional, rows * C / 4 bytes): bit j of byte q = element 4q + j
 * of y is > 0 -- the ReLU mask mrn_bn_bwd_* takes as zmask */
/* NHWC max pooling (padding = -inf), optional fused (scale, shift, relu) on the input.
 * modules/feature_extraction.py:22,25,30,41,234,246,260; modules/transformation.py:71-79. */
int mrn_maxpool_nhwc_f32(const float* x, float* y, const float* scale, const float* shift, int relu,
                         int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, void* stream);
/* [B][HW][C] -> [B][C] mean (AdaptiveAvgPool2d(1), modules/transformation.py:83), optional fused affine+relu */
int mrn_avgpool_nhwc_f32(const float* x, float* y, const float* scale, const float* shift, int relu,
                         int B, int HW, int C, void* stream);

/* ---- TPS rectification ---------------------------------------------------------------------------------- */

/* Fused GridGenerator.build_P_prime + F.grid_sample(border, align_corners=True):
 * modules/transformation.py:204-216 and :38-44. img/out NHWC with C == 4; grid_out optional [B][Hr*Wr][2]. */
int mrn_tps_grid_sample_f32(const float* img_nhwc, const float* cprime, const float* inv_delta_c,
                            const float* p_hat, float* out_nhwc, float* grid_out, int B, int H, int W, int C,
                            int Hr, int Wr, int F, void* stream);

/* ---- recurrent ------------------------------------------------------------------------------------------ */

/* One (bi)directional LSTM layer given xproj = x W_ih^T + b_ih laid out [B][T][ndir*4*hidden]
 * (gate order i,f,g,o), b_hh [ndir*4*hidden] or NULL; out [B][T][ndir*hidden].  w_hh is [ndir][4*hidden][hidden] in the
 * FRAGMENT-MAJOR order the kernel streams (one contiguous 1 KiB line per wave load):
 *   packed[dir][w][g][q][lane][r] = W[dir][g*hidden + 16w + (lane&15)][16q + 4(lane>>4) + r],  w<16, g<4, q<hidden/16.
 * modules/sequence_modeling.py:7-21 (nn.LSTM(bidirectional=True, batch_first=True)). */
int mrn_lstm_layer_fwd_f32(const float* xproj, const float* w_hh, const float* b_hh, float* out, float* gates_out,
                           float* c_out, int B, int T, int hidden, int ndir, void* stream);
/* Backward through time of the same layer.  gates [B][T][ndir][4*hidden] (post-activation i,f,g,o) and cseq
 * [B][T][ndir][hidden] are the forward's optional outputs; w_hhT = fragment-major W_hh^T per direction (rows = hidden
 * units, K = 4*hidden); dgates receives the gradient of the gate pre-activations (dW_ih, dW_hh, db, dx follow as GEMMs). */
int mrn_lstm_layer_bwd_f32(const float* dout, const float* gates, const float* cseq, const float* w_hhT, float* dgates,
                           int B, int T, int hidden, int ndir, void* stream);
/* The same backward pass with the recurrent product dh = dgates . W_hh as split-fp16 x3 on the f16 MFMA (the step of the exact-fp32
 * kernel is bound by its 256 MFMAs per wave): w_hhT_h = per direction the fragment-major fp16 hi / lo stream of W_hh^T [H][4H], w_inv
 * device float[ndir] = 1 / its prescale, gscale device float[2] = {s, 1/s}, s a power of two bringing max|dout| to ~16
 * (mrn_pow2_scale_f32(dout, target 16): the gate gradients are split as s * dgate). */
int mrn_lstm_layer_bwd_x3(const float* dout, const float* gates, const float* cseq, const void* w_hhT_h, const float* w_inv,
                          const float* gscale, float* dgates, int B, int T, int hidden, int ndir, void* stream);

/* Attention decoder, S steps in one launch (modules/prediction.py:58-68 teacher forced; :78-86 greedy when
 * called with S = 1 and carried h_state/c_state).  Hb [B][T][D], Hproj = i2h(Hb) [B][T][hidden],
 * eproj = W_ih[:, D:] emb + b_ih (strided [B][S][4*hidden]), b_hh [4*hidden] or NULL, hid out (strided [B][S][hidden]);
 * alpha_out optional [B][S][T].  w_h2h [hidden][hidden], w_ih_ctx = W_ih[:, :D] and w_hh [4*hidden][hidden] are
 * fragment-major (see mrn_lstm_layer_fwd_f32; one gate group for w_h2h, K = D for w_ih_ctx). */
int mrn_attn_decoder_fwd_f32(const float* Hb, const float* Hproj, const float* eproj, int64_t eproj_stride_b,
                             int64_t eproj_stride_s, const float* w_h2h, const float* b_h2h,
                             const float* w_score, const float* w_ih_ctx, const float* w_hh,
                             const float* b_hh, float* hid, int64_t hid_stride_b, int64_t hid_stride_s, float* h_state,
                             float* c_state, float* alpha_out, float* gates_out, float* c_out, float* ctx_out,
                             float* hp_out, int B, int T, int D, int S, int hidden, void* stream);
/* The same two recurrent kernels for `groups` experts of identical geometry in ONE launch (grid = groups x tiles): in
 * the router phase every expert decodes the same batch (modules/model.py:399-401), and one expert alone occupies 16-32
 * of the 256 CUs.  Every pointer argument is a HOST array of `groups` device pointers (b_hh may be NULL). */
int mrn_lstm_layer_fwd_grouped_f32(const void* const* xproj, const void* const* w_hh, const void* const* b_hh,
                                   const void* const* out, int groups, int B, int T, int hidden, int ndir, void* stream);
/* Inference-only variant for the frozen experts: the recurrent product as split-fp16 x3 on v_mfma_f32_16x16x32_f16
 * (h = hi + lo in LDS, W_hh pre-split with a power-of-two prescale into a fragment-major fp16 stream of the same size).
 * w_hh[g]: [ndir][16][4][H/32][64 lanes][hi 8 | lo 8] fp16; w_inv[g]: device float[ndir] = 1 / prescale. */
int mrn_lstm_layer_fwd_x3_grouped(const void* const* xproj, const void* const* w_hh, const void* const* w_inv,
                                  const void* const* b_hh, const void* const* out, int groups, int B, int T, int hidden,
                                  int ndir, void* stream);
/* One layer being TRAINED, forward with the recurrent product as split-fp16 x3 (il_modules/mrn.py:232-269 through
 * modules/sequence_modeling.py:7-21): mrn_lstm_layer_fwd_x3_grouped for one network plus the saves mrn_lstm_layer_bwd_f32 reads
 * (gates_out [B][T][ndir][4H] post-activation gates, c_out [B][T][ndir][H]). */
int mrn_lstm_layer_fwd_x3_save(const float* xproj, const void* w_hh, const float* w_inv, const float* b_hh, float* out,
                               float* gates_out, float* c_out, int B, int T, int hidden, int ndir, void* stream);
int mrn_attn_decoder_fwd_grouped_f32(const void* const* Hb, const void* const* Hproj, const void* const* eproj,
                                     int64_t eproj_stride_b, int64_t eproj_stride_s, const void* const* w_h2h,
                                     const void* const* b_h2h, const void* const* w_score, const void* const* w_ih_ctx,
                                     const void* const* w_hh, const void* const* b_hh, const void* const* hid,
                                     int64_t hid_stride_b, int64_t hid_stride_s, int groups, int B, int T, int D, int S,
                                     int hidden, void* stream);

/* Backward of the teacher-forced decoder (26 steps of BPTT through attention + LSTMCell in one launch).
 * Saved by the forward: alpha [B][S][T], gates [B][S][4H] (post-activation), cseq [B][S][H], ctx [B][S][D], hp [B][S][H].
 * Weights transposed + fragment-major: w_h2hT (rows = input unit, K = H), w_ih_ctxT (rows = D index, K = 4H),
 * w_hhT (rows = hidden unit, K = 4H).  Outputs: dgates [B][S][4H] (gate pre-activations), dhp [B][S][H],
 * dHb [B][T][D] and dHproj [B][T][H] (written, not accumulated: no initialisation needed),
 * dwscore_part [mrn_attn_decoder_bwd_parts(B)][H] (one row per workgroup; the caller sums the rows),
 * dctx [B][S][D] (gradient of every step's context vector) and de [B][S][T] (gradient of every step's pre-softmax scores): the step
 * loop writes them, and two small launches behind it form dHb = sum_s alpha[s] (x) dctx[s] and dHproj = sum_s de[s] * w * (1 - tanh^2)
 * -- instead of a read-modify-write of both arrays in every step.
 * D must be a multiple of hidden (256). */
int64_t mrn_attn_decoder_bwd_parts(int B);
int mrn_attn_decoder_bwd_f32(const float* Hb, const float* Hproj, const float* alpha, const float* gates,
                             const float* cseq, const float* ctx, const float* hp, const float* dhid,
                             const float* w_score, const float* w_h2hT, const float* w_ih_ctxT, const float* w_hhT,
                             float* dgates, float* dhp, float* dHb, float* dHproj, float* dwscore_part, float* dctx, float* de,
                             int B, int T, int D, int S, int hidden, void* stream);
/* The same backward pass with its three transposed recurrent products (dgates . W_ih_ctx, dgates . W_hh, dhp . W_h2h) as split-fp16 x3:
 * w_h2hT / w_ih_ctxT / w_hhT = fragment-major fp16 hi / lo streams of the transposed weights (ops.pack_fragment_major_h), w_inv device
 * float[3] of their inverse prescales, gscale device float[2] = {s, 1/s}, s a power of two bringing max|dhid| to ~16. */
int mrn_attn_decoder_bwd_x3(const float* Hb, const float* Hproj, const float* alpha, const float* gates, const float* cseq,
                            const float* ctx, const float* hp, const float* dhid, const float* w_score, const void* w_h2hT,
                            const void* w_ih_ctxT, const void* w_hhT, const float* w_inv, const float* gscale, float* dgates, float* dhp,
                            float* dHb, float* dHproj, float* dwscore_part, float* dctx, float* de, int B, int T, int D, int S, int hidden,
                            void* stream);
/* dtable[cut_unknown(idx[b][s])][:] += demb[b][s][:]  (nn.Embedding backward, modules/prediction.py:61) */
int mrn_embed_scatter_add_f32(const int64_t* idx, int64_t idx_stride, const float* demb, float* dtable, int B, int S,
                              int E, int num_class, void* stream);
/* TPS backward: gradient of the rectified image with respect to the fiducials C' (through the bilinear sampler and
 * the grid; modules/transformation.py:33-44,204-216).  dout NHWC [B][Hr][Wr][4] -> dcprime [B][F][2]. */
int mrn_tps_grid_sample_bwd_f32(const float* img_nhwc, const float* cprime, const float* inv_delta_c,
                                const float* p_hat, const float* dout_nhwc, float* dcprime, int B, int H, int W, int C,
                                int Hr, int Wr, int F, void* stream);
/* AdaptiveAvgPool2d(1) backward: dx[b][p][c] = dy[b][c] / HW */
int mrn_avgpool_bwd_nhwc_f32(const float* dy, float* dx, int B, int HW, int C, void* stream);
/* The same decoders with the three recurrent products (h2h, W_ih[:, :D] on the context, W_hh) as split-fp16 x3 on the f16 MFMA: on the
 * exact-fp32 pipe they are 30 us of a step.  w_h2h / w_ih_ctx / w_hh: fragment-major fp16 hi / lo streams with a power-of-two prescale
 * (mrn_amd/ops.py::pack_fragment_major_h), w_inv: device float[3] = 1 / prescale of each (HOST array of such pointers in the grouped
 * form).  D % 32 == 0.  Other arguments as mrn_attn_decoder_fwd_f32 / mrn_attn_decoder_fwd_grouped_f32. */
int mrn_attn_decoder_fwd_x3(const float* Hb, const float* Hproj, const float* eproj, int64_t eproj_stride_b, int64_t eproj_stride_s,
                            const void* w_h2h, const float* b_h2h, const float* w_score, const void* w_ih_ctx, const void* w_hh,
                            const float* w_inv, const float* b_hh, float* hid, int64_t hid_stride_b, int64_t hid_stride_s,
                            float* h_state, float* c_state, float* alpha_out, float* gates_out, float* c_out, float* ctx_out,
                            float* hp_out, int B, int T, int D, int S, int hidden, void* stream);
int mrn_attn_decoder_fwd_x3_grouped(const void* const* Hb, const void* const* Hproj, const void* const* eproj, int64_t eproj_stride_b,
                                    int64_t eproj_stride_s, const void* const* w_h2h, const void* const* b_h2h,
                                    const void* const* w_score, const void* const* w_ih_ctx, const void* const* w_hh,
                                    const void* const* w_inv, const void* const* b_hh, const void* const* hid, int64_t hid_stride_b,
                                    int64_t hid_stride_s, int groups, int B, int T, int D, int S, int hidden, void* stream);
/* out[b][s][:] = table[cut_unknown(idx[b][s])][:]  (modules/prediction.py:35-36,61) */
int mrn_embed_gather_f32(const int64_t* idx, int64_t idx_stride, const float* table, float* out, int B, int S,
                         int E, int num_class, void* stream);

/* ---- row-wise operators of the DM-Router (modules/dm_router.py) ------------------------------------------- */

/* LayerNorm over the contiguous dim of strided rows (nn.LayerNorm(channel), dm_router.py:8,40; eps 1e-5);
 * mean / rstd [rows] are saved for the backward. C % 4 == 0, C <= 1024. */
int mrn_layernorm_fwd_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                          float* mean, float* rstd, int64_t rows, int C, float eps, void* stream);
/* ... and, for the Linear layer the norm feeds in an expert being TRAINED (svtr.py:200-204: norm1 -> qkv, norm2 -> fc1), the result a
 * second time as that GEMM's range-scaled HL32 operand: y_hl [rows][C/32][128 B] = split(s * y), scale_out = {s, 1/s} with s the largest
 * power of two such that s * (sqrt(C) max|gamma| + max|beta|) <= target -- a bound from the parameters: no max|y| pass, no split pass */
int mrn_layernorm_fwd_hl32_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy, float* mean,
                               float* rstd, int64_t rows, int C, float eps, void* y_hl, float* scale_out, float target, void* stream);
/* elementwise producers of a trained Linear layer's operand over contiguous [rows][C]: y = op(a, b) (op 0 gelu(a), 1 b * gelu'(a), 3 a + b,
 * 8 a + b * d[row / rows_per_d]: svtr.py:46-67 Mlp activation and its gradient, :17-22,202-203 the DropPath residual add) with, in the same
 * pass, y_hl = split(scale[0] * y) (HL32, C % 32 == 0) for a scale the caller has (a bound of max|y|) and / or max|y| folded into amax_ws
 * (64 words, mrn_pow2_finalize_f32) */
int mrn_ew_operand_f32(const float* a, const float* b, const float* d, int64_t rows_per_d, float* y, void* y_hl, const float* scale,
                       void* amax_ws, int64_t rows, int C, int op, void* stream);
int64_t mrn_layernorm_bwd_blocks(int64_t rows);
/* dx (+)= LayerNorm backward; partials [blocks][2][C] receive per-block sums of (dgamma, dbeta) */
int mrn_layernorm_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                          const float* mean, const float* rstd, float* dx, int64_t lddx, int accumulate,
                          float* partials, int64_t rows, int C, void* stream);
/* LayerNorm over the P axis of x[B][P][Wd] for every (b, w) column -- ChannelDomainGating's LayerNorm(patch) on
 * the 'b (d c) p' rearrangement (dm_router.py:23,29,63) without the transpose; mean / rstd are [B][Wd]. */
int mrn_colnorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int B, int P, int Wd, float eps, void* stream);
/* partials [B*ceil(Wd/256)][2][P] */
int mrn_colnorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                        float* dx, int accumulate, float* partials, int B, int P, int Wd, void* stream);
/* elementwise on strided rows: op 0 y=gelu(a) (dm_router.py:42), 1 y=b*gelu'(a), 2 y=a*b (:17,:33), 3 y=a+b,
 * 4 y=b*(a>0) (ReLU backward) */
int mrn_ew_rows_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t rows,
                    int C, int op, void* stream);
/* in-place softmax over the N columns of every row of s, with an optional additive mask [rows_per_mask][N] shared
 * across the batch (row r uses mask row r % rows_per_mask): SVTR Local/Global mixing, modules/svtr.py:140-146 */
int mrn_softmax_rows_f32(float* s, const float* mask, int64_t rows, int N, int rows_per_mask, void* stream);
/* backward of mrn_softmax_rows_f32, in place on dp: ds = p * (dp - rowsum(p * dp)) (autograd of svtr.py:146 in expert training) */
int mrn_softmax_rows_bwd_f32(const float* p, float* dp, int64_t rows, int N, void* stream);
/* One pass between two Linear layers of the SVTR mixing blocks of G lock-step experts (modules/svtr.py:200-204 Block.forward,
 * :298-305 SubSample): t = x + drop[r / rows_per_drop] * branch (DropPath-scaled residual add; branch NULL: t = x, drop NULL: 1)
 * -> sum_out (optional, may alias x); y = LayerNorm(t) * gamma[g] + beta[g] with g = r / rows_per_group (gamma NULL: y = t)
 * -> y_f32 and / or y_hl32 (HL32 operand of the next grouped Linear, C % 32 == 0).  [rows][C] contiguous, gamma/beta [G][C]. */
int mrn_add_layernorm_grouped_f32(const float* x, const float* branch, const float* drop, int64_t rows_per_drop,
                                  const float* gamma, const float* beta, int64_t rows_per_group, float* sum_out, float* y_f32,
                                  void* y_hl32, int64_t rows, int C, float eps, void* stream);
/* The Mlp of an SVTR mixing block (modules/svtr.py:46-67: fc1 -> GELU -> fc2) of G lock-step experts in ONE kernel, split-fp16 x3
 * products, the 4C-wide hidden activation kept in registers (chained MFMAs on the transposed problem, tokens on the N axis; the fc2
 * weights come with the hidden index of every 32-block permuted to the MFMA result layout: position 16 s + 8 h + j holds unit
 * (j & 3) + 8 (2 s + (j >> 2)) + 4 h).  x_hl [rows][C/32][128 B] = the HL32 LayerNorm output; w1_hl [G][4C][C/32][128 B], w2_hl
 * [G][C][4C/32][128 B] (mrn_pack_weight_hl32), s1 / s2 [G][2] their prescales, b1 [G][4C], b2 [G][C]; y [rows][C] fp32.  C = 64 | 128. */
int mrn_svtr_mlp_x3_f32(const void* x_hl, const void* w1_hl, const float* s1, const float* b1, const void* w2_hl, const float* s2,
                        const float* b2, float* y, int64_t rows, int64_t rows_per_group, int G, int C, void* stream);

/* The second half of an SVTR stage-3 mixing block (C = 256; modules/svtr.py:196-204 from the attention context on) for the rows of G
 * lock-step experts in one launch: x_res[r] += drop[r / rows_per_drop] * (ctx[r] Wproj^T + bproj) (drop NULL: 1, in place);
 * branch[r] = fc2(GELU(fc1(LayerNorm(x_res[r]; gamma[g], beta[g], eps)))).  ctx_hl [rows][8][128 B] from
 * mrn_svtr_attention_block_x3_f32; wproj_hl / sproj / bproj: proj packed by mrn_pack_weight_hl32; w1_hl as for mrn_svtr_mlp_x3_f32 but
 * with fc1's INPUT channels permuted inside every 32-block like fc2's hidden units there; the rest as there. */
int mrn_svtr_tail_x3_f32(const void* ctx_hl, float* x_res, const void* wproj_hl, const float* sproj, const float* bproj, const float* drop,
                         int64_t rows_per_drop, const float* gamma, const float* beta, float eps, const void* w1_hl, const float* s1,
                         const float* b1, const void* w2_hl, const float* s2, const float* b2, float* branch, int64_t rows,
                         int64_t rows_per_group, int G, int C, void* stream);
/* The attention half of an SVTR mixing block (modules/svtr.py:196-201 `x = x + drop_path(mixer(norm1(x)))`, Attention :90-152) of G
 * lock-step frozen experts in ONE kernel (group = image / imgs_per_group):
 *   t = x + drop_prev[img] * pending;  x_out = t + drop1[img] * proj(attention(qkv(LayerNorm1(t))));  y_hl = HL32(LayerNorm2(x_out))
 * x, pending, x_out [imgs][N][C] fp32 (pending / drop_prev / drop1 may be NULL); g1, b1, g2, b2 [G][C]; wqkv_hl [G][3C][C/32][128 B]
 * (mrn_pack_weight_hl32 of [3C][1][C]) with prescale sqkv [G][2] and bias bqkv [G][3C] or NULL; mask_bits [N][ceil(N/32)] visibility
 * bits of the local mixer's 0 / -inf mask or NULL; wproj_hl [G][C][C/32][128 B] packed with the input channel of every 32-block (head)
 * permuted to the MFMA result layout (position 16 s + 8 h + j holds channel (j & 3) + 8 (2 s + (j >> 2)) + 4 h), sproj [G][2], bproj
 * [G][C]; y_hl [imgs * N][C/32][128 B] is the operand of mrn_svtr_mlp_x3_f32.  Split-fp16 x3 products throughout; q | k | v, the
 * probabilities and the context never leave registers / LDS.  Supported: C = 64 with N <= 512, C = 128 with N <= 256 (N <= 128:
 * imgs_per_group even); otherwise MRN_ERR_UNSUPPORTED (the caller runs the unfused chain).
 * token_h, token_w: 0, 0 = tokens walked in memory order.  For a LOCAL mixer on a token_h x token_w map (token_h * token_w == N, the 7 x 11
 * window of svtr.py:110-128 sees every row but only 11 columns) pass the map's shape and mask_bits in COLUMN-major position order
 * (position p = col * token_h + row <-> token row * token_w + col, rows and bits permuted alike): a 32-position key tile is then a block of
 * whole columns and 5 of 8 (stage 2) / 11 of 16 (stage 1) tiles are fully masked and skipped.  Same result per token. */
int mrn_svtr_mixer_x3_f32(const float* x, const float* pending, const float* drop_prev, const float* g1, const float* b1, float eps1,
                          const void* wqkv_hl, const float* sqkv, const float* bqkv, const void* mask_bits, float scale,
                          const void* wproj_hl, const float* sproj, const float* bproj, const float* drop1, const float* g2,
                          const float* b2, float eps2, float* x_out, void* y_hl, int imgs, int imgs_per_group, int N, int C,
                          int token_h, int token_w, void* stream);
/* Attention-only form for the wide stage (C = 256, SVTR stage 3: the proj accumulators of mrn_svtr_mixer_x3_f32 do not fit next to the token
 * fragments): t = x + drop_prev * pending (written to t_out when pending is given); ctx_hl = HL32(attention(qkv(LayerNorm1(t)))), the
 * operand of the (unfused) proj Linear -- the residual add and LayerNorm2 follow as mrn_add_layernorm_grouped_f32.  Saves the LayerNorm
 * pass and the qkv round trip (modules/svtr.py:130-152 behind :200's norm1).  C = 256, N <= 128, imgs_per_group a multiple of 2 (N > 64) or
 * 4; otherwise MRN_ERR_UNSUPPORTED.  Other arguments as mrn_svtr_mixer_x3_f32. */
int mrn_svtr_attention_block_x3_f32(const float* x, const float* pending, const float* drop_prev, const float* g1, const float* b1,
                                    float eps1, const void* wqkv_hl, const float* sqkv, const float* bqkv, const void* mask_bits,
                                    float scale, float* t_out, void* ctx_hl, int imgs, int imgs_per_group, int N, int C, void* stream);
/* Fused multi-head attention of the SVTR mixing blocks (head dimension 32), inference path of the frozen experts:
 * out[b][n][h*32 + :] = softmax_m(scale * q[b][n][h] . k[b][m][h] + mask[n][m]) @ v[b][m][h]; qkv [B][N][3*C] (q | k | v,
 * C = heads * 32), mask [N][N] additive and SYMMETRIC (SVTR's local window mask) or NULL, out [B][N][C] fp32 and / or
 * out_hl32 (the same tensor as the HL32 operand of the proj Linear: a head IS one 32-channel block).  Online softmax
 * on the exact-fp32 MFMA (x3 = 0) or, for frozen experts, with both products as split-fp16 x3 on the f16 MFMA (x3 = 1, 22-bit
 * products like the experts' other GEMMs): the [B][heads][N][N] score tensor of modules/svtr.py:140-149 never reaches HBM.
 * mask_bits: instead of `mask`, the visibility bits [N][ceil(N/32)] of a mask whose entries are 0 or -inf (SVTR's local window
 * mask, svtr.py:117-128): bit k of word t of row q set = key 32t + k is visible to query q. */
int mrn_svtr_attention_f32(const float* qkv, const float* mask, const void* mask_bits, float* out, void* out_hl32, float* lse,
                           int B, int N, int C, int heads, float scale, int x3, const float* hl_scale, void* stream);
/* hl_scale (optional, x3 = 0): {s, 1/s}, out_hl32 = split(s * out): the range-scaled operand of the proj Linear of an expert being trained;
 * rows of out are convex combinations of rows of v, so the scale of max|qkv| (the qkv GEMM's amax_ws) is a valid one */
/* Backward of the above for an expert being trained (autograd of svtr.py:140-149 under loss.backward(), il_modules/mrn.py:260):
 * the forward call also stores lse [B][heads][N] (base-2 log-sum-exp of the scaled, masked scores; pass NULL when frozen);
 * dqkv [B][N][3*C] is recomputed tile by tile from qkv, out, dout and lse -- no [N][N] tensor is kept.  dsum: [B][heads][N]
 * floats of workspace (rowsum(dout o out)). */
int mrn_svtr_attention_bwd_f32(const float* qkv, const float* mask, const float* out, const float* dout, const float* lse,
                               float* dsum, float* dqkv, int B, int N, int C, int heads, float scale, void* amax_ws, void* stream);
/* amax_ws (optional): max|dqkv| folded into 64 words by both kernels (mrn_pow2_finalize_f32): the range of the qkv Linear's gradient operand */
/* y = x + scale[row / rows_per_group] * branch : residual add with the per-sample DropPath scale (svtr.py:7-22,202-203) */
int mrn_residual_scale_rows_f32(const float* x, const float* branch, const float* scale, float* y, int64_t rows, int C,
                                int64_t rows_per_group, void* stream);
/* out[c] (+)= sum_r in[r][c] (bias / affine gradients, split-K combine); workspace: chunks*C floats */
/* BatchNorm batch statistics of a tensor that is not a conv output (RCNN extractor: BN over gated products,
 * modules/feature_extraction.py:146-161): part [mrn_bn_stats_blocks(rows)][2][C] partial sums / sums of squares of x [rows][C] */
int64_t mrn_bn_stats_blocks(int64_t rows);
int mrn_bn_stats_f32(const float* x, int64_t rows, int C, float* part, void* stream);
int64_t mrn_colsum_chunks(int64_t rows, int C);
int mrn_colsum_f32(const float* in, int64_t ld, float* out, float* workspace, int64_t rows, int C, int accumulate,
                   void* stream);
/* out[i][j] = in[row_idx[i]][col_idx[j]] (NULL index = identity), columns [C, ld_out) zero-filled */
int mrn_gather2d_f32(const float* in, int64_t ld_in, const int* row_idx, const int* col_idx, float* out,
                     int64_t ld_out, int R, int C, void* stream);
/* first index of the row maximum (preds.max(2), test.py:211; greedy decode prediction.py:84) */
int mrn_argmax_f32(const float* x, int64_t ld, int64_t* out, int64_t rows, int C, void* stream);
/* greedy decode + confidence in one pass (test.py:211,218-219: preds.max(2); F.softmax(preds, 2).max(2)): idx = first argmax,
   prob = softmax(row)[argmax] */
int mrn_argmax_prob_f32(const float* x, int64_t ld, int64_t* idx, float* prob, int64_t rows, int C, void* stream);

/* ---- MRN fan-in and gate tail (modules/model.py:361-423) --------------------------------------------------- */

/* out[b][t][c] = sum_i w[b][i] * (c < C_i ? L_i[b][t][c] : 1)  -- pad-with-ones + stack + weight + sum in one pass.
 * logits / lds / classes are HOST arrays of length I (device pointers, row strides (multiples of 4), class counts). */
int mrn_fanin_fwd_f32(const void* const* logits, const int64_t* lds, const int* classes, int I, const float* w,
                      float* out, int64_t ldo, int B, int T, int C, void* stream);
/* dw[b][i] = sum_{t,c} dout[b][t][c] * Lpad_i[b][t][c]; workspace B*T*I floats */
int mrn_fanin_bwd_f32(const void* const* logits, const int64_t* lds, const int* classes, int I, const float* dout,
                      int64_t ldd, float* dw, float* workspace, int B, int T, int C, void* stream);
/* eval hard routing (model.py:377,393): out[b] = Lpad_{index[b]}[b] */
int mrn_select_expert_f32(const void* const* logits, const int64_t* lds, const int* classes, int I,
                          const int64_t* index, float* out, int64_t ldo, int B, int T, int C, void* stream);
/* s = route(r) over the patch axis of r[B][P][I], w = softmax(beta*s) (model.py:405-406); argmax for eval (:377) */
int mrn_gate_tail_fwd_f32(const float* r, const float* w_route, const float* b_route, float beta, float* s_out,
                          float* w_out, int64_t* argmax_out, int B, int P, int I, void* stream);
int mrn_gate_tail_bwd_f32(const float* w, const float* dw, const float* r, const float* w_route, float beta,
                          float* ds, float* dr, float* d_w_route, float* d_b_route, int B, int P, int I, void* stream);

/* ---- losses -------------------------------------------------------------------------------------------------- */

/* CrossEntropyLoss(mean, ignore_index) over strided rows (il_modules/base.py:134, mrn.py:150-152,254-258,342) */
int mrn_ce_loss_fwd_f32(const float* logits, int64_t ld, const int64_t* target, int64_t ignore_index, int64_t rows,
                        int C, float* lse, float* loss_rows, float* loss, float* inv_count, void* stream);
int mrn_ce_loss_bwd_f32(const float* logits, int64_t ld, const int64_t* target, int64_t ignore_index,
                        const float* lse, const float* upstream, const float* inv_count, float* dlogits, int64_t ldd,
                        int64_t rows, int C, void* stream);
/* log_softmax + CTCLoss(blank, mean, zero_infinity=True), all input lengths = T (base.py:131, mrn.py:250-252) */
int64_t mrn_ctc_occ_floats(int B, int T);
int mrn_ctc_loss_fwd_f32(const float* logits, int64_t ld, const int64_t* targets, int64_t tstride,
                         const int* target_len, int max_target_len, float* lse, float* nll, float* occ, float* loss,
                         int B, int T, int C, int blank, void* stream);
int mrn_ctc_loss_bwd_f32(const float* logits, int64_t ld, const float* lse, const float* occ, const int64_t* targets,
                         int64_t tstride, const int* target_len, const float* nll, const float* upstream,
                         float* dlogits, int64_t ldd, int B, int T, int C, int blank, void* stream);

/* Knowledge distillation of LwF / WA (il_modules/lwf.py:81-87,111-114): loss = -sum softmax(old/T) log_softmax(new/T) / rows
 * over the class slice [c0, c1); bwd writes d loss / d new over all C columns (zeros outside the slice). */
int mrn_kd_loss_fwd_f32(const float* xnew, int64_t ldn, const float* xold, int64_t ldo, int c0, int c1, float T,
                        int64_t rows, float* loss_rows, float* loss, void* stream);
int mrn_kd_loss_bwd_f32(const float* xnew, int64_t ldn, const float* xold, int64_t ldo, int c0, int c1, float T,
                        int64_t rows, const float* upstream, float* dnew, int64_t ldd, int C, void* stream);

/* ---- EWC and weight alignment (il_modules/ewc.py:120-167, modules/model.py:166-174) ------------------------------ */
int mrn_fisher_accumulate_f32(float* fisher, const float* grad, int64_t n, void* stream);           /* F += g^2 */
int mrn_fisher_finalize_f32(float* fisher, int64_t n, float inv_iterations, float fisher_max, void* stream);
int mrn_ewc_penalty_fwd_f32(const float* fisher, const float* p, const float* mean, int64_t n, float* workspace,
                            float* penalty, void* stream);                                       /* sum F (p-p*)^2 / 2 */
int mrn_ewc_penalty_bwd_f32(const float* fisher, const float* p, const float* mean, float* grad, int64_t n, float coef,
                            void* stream);                                                       /* g += coef F (p-p*) */
int mrn_weight_align_f32(float* w, int64_t ld, int64_t rows, int64_t n_old, int C, float* workspace, float* gamma_out,
                         void* stream);

/* ---- optimiser (il_modules/base.py:85,255-262) ------------------------------------------------------------------ */

int64_t mrn_grad_norm_workspace_floats(int64_t n);
/* norm_coef: THREE floats.  [0] = ||g||_2, [1] = min(1, max_norm/(norm+1e-6))  (clip_grad_norm_); [2] += 1 when the norm is not
   finite -- the update kernels below skip such a step (parameters and optimiser state untouched) -- a counter the caller
   zero-initialises once and keeps across steps */
int mrn_grad_norm_clip_f32(const float* g, int64_t n, float max_norm, float* workspace, float* norm_coef, void* stream);
/* g *= coef (in place), then torch.optim.Adam's update; step_size = lr/(1-beta1^t), bc2_sqrt = sqrt(1-beta2^t) */
int mrn_adam_step_f32(float* p, float* g, float* m, float* v, int64_t n, const float* norm_coef, float step_size,
                      float beta1, float beta2, float bc2_sqrt, float eps, void* stream);
/* torch.optim.SGD (il_modules/base.py:74-79: momentum, weight decay; dampening 0, no Nesterov): g *= coef, buf = mu*buf + g + wd*p,
   p -= lr*buf.  buf zero-initialised by the caller (may be NULL when momentum == 0). */
int mrn_sgd_step_f32(float* p, float* g, float* buf, int64_t n, const float* norm_coef, float lr, float momentum,
                     float weight_decay, void* stream);
/* torch.optim.Adadelta (il_modules/base.py:80-83: rho, eps): running averages square_avg / acc_delta zero-initialised by the caller */
int mrn_adadelta_step_f32(float* p, float* g, float* square_avg, float* acc_delta, int64_t n, const float* norm_coef, float lr,
                          float rho, float eps, void* stream);

/* ---- data-parallel collectives over RCCL / xGMI ------------------------------------------------------------------
 * Replace the per-iteration replicate / scatter / gather / reduce of torch.nn.DataParallel (il_modules/base.py:68,
 * il_modules/mrn.py:106,133) for one-process-per-GPU hosts: rank 0 calls mrn_comm_unique_id (HOST buffer of
 * mrn_comm_unique_id_bytes() bytes) and distributes it, every rank calls mrn_comm_init with its HIP device current, then the flat
 * gradient buckets go through mrn_allreduce_f32 (average = 1: mean over ranks) and freshly built models through mrn_broadcast_f32.
 * RCCL is bound with dlopen at mrn_comm_init time (a process that already carries an RCCL shares it).  Device buffers, in place,
 * asynchronous on `stream`. */
int64_t mrn_comm_unique_id_bytes(void);
int mrn_comm_unique_id(void* id_out);
int mrn_comm_init(int rank, int world, const void* unique_id);
int64_t mrn_comm_world(void);
int64_t mrn_comm_rank(void);
int mrn_allreduce_f32(float* buf, int64_t n, int average, void* stream);
int mrn_broadcast_f32(float* buf, int64_t n, int root, void* stream);
int mrn_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif
