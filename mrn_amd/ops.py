"""Thin tensor-level wrappers over the C ABI (include/mrn_hip.h).

PyTorch is used here only for device memory (caching allocator), the current HIP stream and shapes; every
arithmetic operation is a call into libmrn_hip.so.  All tensors must be fp32 CUDA tensors.
"""
import torch

from ._lib import call

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mrn_amd ops need CUDA (HIP) tensors; there is no CPU fallback on the product path")
        if t.dtype != torch.float32:
            raise RuntimeError(f"mrn_amd ops are fp32; got {t.dtype}")


# ---------------------------------------------------------------------------------------------------------
# GEMM family
# ---------------------------------------------------------------------------------------------------------
def gemm_raw(A, W, C, M, N, K, batch=1, sA=(0, 0, 1), sW=(0, 0, 1), sC=(0, 0, 1), bias=None, bias_axis=0,
             bias_batch_stride=0, residual=None, act=ACT_NONE, accumulate=False, alpha=1.0):
    """C[b,m,n] = act(alpha * sum_k A[b,m,k] W[b,n,k] + bias + residual); strides are (batch, row, k) in elements."""
    _chk(A, W, C, bias, residual)
    call("mrn_gemm_f32", _p(A), _p(W), _p(bias), _p(residual), _p(C), M, N, K, batch,
         sA[0], sA[1], sA[2], sW[0], sW[1], sW[2], sC[0], sC[1], sC[2], bias_batch_stride, bias_axis,
         act, int(accumulate), float(alpha), _stream())
    return C


def linear(x, weight, bias=None, act=ACT_NONE, residual=None, out=None):
    """y = act(x @ weight.T + bias (+ residual)); x [..., K] with contiguous last dim and uniform row stride."""
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = x.reshape(-1, K) if x.is_contiguous() else x
    if x2.dim() != 2:
        x2 = x.contiguous().view(-1, K)
    M = x2.shape[0]
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    o2 = out.view(-1, N) if out.is_contiguous() else out
    assert o2.dim() == 2 and o2.stride(1) == 1 and x2.stride(1) == 1 and weight.stride(1) == 1
    r2 = None
    if residual is not None:
        r2 = residual.view(-1, N) if residual.is_contiguous() else residual
        assert r2.stride() == o2.stride()
    gemm_raw(x2, weight, o2, M, N, K, 1, (0, x2.stride(0), 1), (0, weight.stride(0), 1), (0, o2.stride(0), 1),
             bias=bias, residual=r2, act=act)
    return out


# ---------------------------------------------------------------------------------------------------------
# convolution / BatchNorm / pooling (NHWC)
# ---------------------------------------------------------------------------------------------------------
def nchw_to_nhwc(x):
    _chk(x)
    B, C, H, W = x.shape
    x = x.contiguous()
    y = torch.empty(B, H, W, C, device=x.device, dtype=torch.float32)
    call("mrn_nchw_to_nhwc_f32", _p(x), _p(y), B, C, H, W, _stream())
    return y


def pack_conv_weight(w):
    _chk(w)
    O, I, kh, kw = w.shape
    w = w.contiguous()
    out = torch.empty(O, kh, kw, I, device=w.device, dtype=torch.float32)
    call("mrn_pack_conv_weight_f32", _p(w), _p(out), O, I, kh, kw, _stream())
    return out


def conv_out_hw(H, W, k, s, p):
    return (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1


def conv2d_nhwc(x, w_ohwi, bias=None, stride=(1, 1), padding=(0, 0), act=ACT_NONE, want_stats=False):
    """x [B,H,W,Cin] -> y [B,Ho,Wo,Cout]; returns (y, stats or None) where stats are per-128-row-block partials."""
    _chk(x, w_ohwi, bias)
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = w_ohwi.shape
    Ho, Wo = conv_out_hw(H, W, (kh, kw), stride, padding)
    y = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    stats = None
    if want_stats:
        n = call("mrn_conv2d_stats_floats", B, Ho, Wo, Cout)
        stats = torch.empty(n, device=x.device, dtype=torch.float32)
    call("mrn_conv2d_nhwc_f32", _p(x), _p(w_ohwi), _p(bias), _p(y), _p(stats), B, H, W, Cin, Cout, kh, kw,
         stride[0], stride[1], padding[0], padding[1], act, _stream())
    return y, stats


def bn_finalize(stats, C, count, gamma, beta, running_mean, running_var, momentum, eps, save=False):
    dev = stats.device
    scale = torch.empty(C, device=dev, dtype=torch.float32)
    shift = torch.empty(C, device=dev, dtype=torch.float32)
    mean = torch.empty(C, device=dev, dtype=torch.float32) if save else None
    invstd = torch.empty(C, device=dev, dtype=torch.float32) if save else None
    nblk = stats.numel() // (2 * C)
    call("mrn_bn_finalize_f32", _p(stats), nblk, C, count, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
         float(momentum), float(eps), _p(scale), _p(shift), _p(mean), _p(invstd), _stream())
    return scale, shift, mean, invstd


def bn_eval_affine(gamma, beta, running_mean, running_var, eps):
    C = running_mean.numel()
    scale = torch.empty(C, device=running_mean.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    call("mrn_bn_eval_affine_f32", _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), C,
         _p(scale), _p(shift), _stream())
    return scale, shift


def scale_shift_act(x, scale, shift, relu=True, residual=None, out=None):
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = x
    call("mrn_scale_shift_act_f32", _p(x), _p(residual), _p(out), _p(scale), _p(shift), rows, C, int(relu), _stream())
    return out


def maxpool_nhwc(x, kernel, stride, padding=(0, 0), scale=None, shift=None, relu=False):
    B, H, W, C = x.shape
    Ho, Wo = conv_out_hw(H, W, kernel, stride, padding)
    y = torch.empty(B, Ho, Wo, C, device=x.device, dtype=torch.float32)
    call("mrn_maxpool_nhwc_f32", _p(x), _p(y), _p(scale), _p(shift), int(relu), B, H, W, C, kernel[0], kernel[1],
         stride[0], stride[1], padding[0], padding[1], _stream())
    return y


def avgpool_nhwc(x, scale=None, shift=None, relu=False):
    B, H, W, C = x.shape
    y = torch.empty(B, C, device=x.device, dtype=torch.float32)
    call("mrn_avgpool_nhwc_f32", _p(x), _p(y), _p(scale), _p(shift), int(relu), B, H * W, C, _stream())
    return y


# ---------------------------------------------------------------------------------------------------------
# TPS
# ---------------------------------------------------------------------------------------------------------
def tps_grid_sample(img_nhwc, cprime, inv_delta_c, p_hat, out_hw, want_grid=False):
    _chk(img_nhwc, cprime, inv_delta_c, p_hat)
    B, H, W, C = img_nhwc.shape
    Hr, Wr = out_hw
    F = cprime.shape[1]
    out = torch.empty(B, Hr, Wr, C, device=img_nhwc.device, dtype=torch.float32)
    grid = torch.empty(B, Hr * Wr, 2, device=img_nhwc.device, dtype=torch.float32) if want_grid else None
    call("mrn_tps_grid_sample_f32", _p(img_nhwc), _p(cprime.contiguous()), _p(inv_delta_c), _p(p_hat), _p(out),
         _p(grid), B, H, W, C, Hr, Wr, F, _stream())
    return (out, grid) if want_grid else out


# ---------------------------------------------------------------------------------------------------------
# recurrent
# ---------------------------------------------------------------------------------------------------------
def lstm_layer(xproj, w_hh, hidden, ndir):
    """xproj [B,T,ndir*4H] (already includes both biases), w_hh [ndir,4H,H] -> [B,T,ndir*H]"""
    _chk(xproj, w_hh)
    B, T, _ = xproj.shape
    out = torch.empty(B, T, ndir * hidden, device=xproj.device, dtype=torch.float32)
    call("mrn_lstm_layer_fwd_f32", _p(xproj), _p(w_hh), _p(out), B, T, hidden, ndir, _stream())
    return out


def embed_gather(idx, table, num_class):
    """idx [B,S] int64 (any row stride), table [C,E] -> [B,S,E] with cut_unknown semantics"""
    assert idx.dtype == torch.int64 and idx.stride(1) == 1
    B, S = idx.shape
    E = table.shape[1]
    out = torch.empty(B, S, E, device=table.device, dtype=torch.float32)
    call("mrn_embed_gather_f32", _p(idx), idx.stride(0), _p(table), _p(out), B, S, E, num_class, _stream())
    return out


def attn_decoder(Hb, Hproj, eproj, w_h2h, b_h2h, w_score, w_ih, w_hh, hidden, hid=None, h_state=None, c_state=None,
                 want_alpha=False):
    _chk(Hb, Hproj, eproj, w_h2h, b_h2h, w_score, w_ih, w_hh)
    B, T, D = Hb.shape
    S = eproj.shape[1]
    if hid is None:
        hid = torch.empty(B, S, hidden, device=Hb.device, dtype=torch.float32)
    alpha = torch.empty(B, S, T, device=Hb.device, dtype=torch.float32) if want_alpha else None
    assert eproj.stride(2) == 1 and hid.stride(2) == 1 and w_ih.stride(1) == 1
    call("mrn_attn_decoder_fwd_f32", _p(Hb), _p(Hproj), _p(eproj), eproj.stride(0), eproj.stride(1), _p(w_h2h),
         _p(b_h2h), _p(w_score), _p(w_ih), w_ih.stride(0), _p(w_hh), _p(hid), hid.stride(0), hid.stride(1),
         _p(h_state), _p(c_state), _p(alpha), B, T, D, S, hidden, _stream())
    return (hid, alpha) if want_alpha else hid
