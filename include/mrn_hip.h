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
 *   - the caller owns every buffer; the library allocates nothing and keeps no mutable global state
 *   - return 0 on success, <0 library error, >0 hipError_t; text via mrn_last_error() (thread-local)
 *   - activations are fp32, NHWC (channels innermost); "rows" means all leading dims flattened
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
/* y = [relu](x * scale[c] + shift[c] + residual): BatchNorm apply + residual add + ReLU in one pass
 * (BasicBlock tail, modules/feature_extraction.py:184-199). In place allowed. */
int mrn_scale_shift_act_f32(const float* x, const float* residual, float* y, const float* scale,
                            const float* shift, int64_t rows, int C, int relu, void* stream);
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

/* One (bi)directional LSTM layer given xproj = x W_ih^T + b_ih + b_hh laid out [B][T][ndir*4*hidden]
 * (gate order i,f,g,o), w_hh [ndir][4*hidden][hidden]; out [B][T][ndir*hidden].
 * modules/sequence_modeling.py:7-21 (nn.LSTM(bidirectional=True, batch_first=True)). */
int mrn_lstm_layer_fwd_f32(const float* xproj, const float* w_hh, float* out, int B, int T, int hidden,
                           int ndir, void* stream);

/* Attention decoder, S steps in one launch (modules/prediction.py:58-68 teacher forced; :78-86 greedy when
 * called with S = 1 and carried h_state/c_state).  Hb [B][T][D], Hproj = i2h(Hb) [B][T][hidden],
 * eproj = W_ih[:, D:] emb + b_ih + b_hh (strided [B][S][4*hidden]), w_ih [4*hidden][ld_wih] (context part =
 * first D columns), hid out (strided [B][S][hidden]); alpha_out optional [B][S][T]. */
int mrn_attn_decoder_fwd_f32(const float* Hb, const float* Hproj, const float* eproj, int64_t eproj_stride_b,
                             int64_t eproj_stride_s, const float* w_h2h, const float* b_h2h,
                             const float* w_score, const float* w_ih, int64_t ld_wih, const float* w_hh,
                             float* hid, int64_t hid_stride_b, int64_t hid_stride_s, float* h_state,
                             float* c_state, float* alpha_out, int B, int T, int D, int S, int hidden,
                             void* stream);
/* out[b][s][:] = table[cut_unknown(idx[b][s])][:]  (modules/prediction.py:35-36,61) */
int mrn_embed_gather_f32(const int64_t* idx, int64_t idx_stride, const float* table, float* out, int B, int S,
                         int E, int num_class, void* stream);

#ifdef __cplusplus
}
#endif
#endif
