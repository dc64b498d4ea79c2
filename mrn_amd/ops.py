"""Thin tensor-level wrappers over the C ABI (include/mrn_hip.h).

PyTorch is used here only for device memory (caching allocator), the current HIP stream and shapes; every
arithmetic operation is a call into libmrn_hip.so.  All tensors must be fp32 CUDA tensors.
"""
import os
import sys
import weakref

import torch

from ._lib import call

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


class KernelTimer:
    """Optional HIP-event bracket around the dominant kernel (the 128x128 implicit-GEMM conv) so bench.py can report
    achieved FLOP/s per launch, measured on the stream the kernel runs on."""

    def __init__(self, only=None):
        self.spans = []          # (start_event, end_event, flops)
        self.only = only         # None: every timed call site; else a set of site tags ("wino": the row-block / Winograd convolutions)

    def begin(self, site="other"):
        if self.only is not None and site not in self.only:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, start, flops, kind="f32", nbytes=0.0):
        if start is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.spans.append((start, e, flops, kind, nbytes))

    def summary(self):
        """per kernel kind: launches, summed event time, UNION of the launch intervals (two lock-step half-groups run on
        two streams, so launches of one kernel overlap; the union is the time the kernel actually occupied the GPU),
        total algorithmic flops / bytes"""
        torch.cuda.synchronize()
        if not self.spans:
            return {}
        ref = self.spans[0][0]
        out, ivals = {}, {}
        for s, e, f, kind, nb in self.spans:
            d = out.setdefault(kind, {"launches": 0, "total_ms": 0.0, "total_flops": 0.0, "total_bytes": 0.0})
            d["launches"] += 1
            d["total_ms"] += s.elapsed_time(e)
            d["total_flops"] += f
            d["total_bytes"] += nb
            t0 = ref.elapsed_time(s)
            ivals.setdefault(kind, []).append((t0, t0 + s.elapsed_time(e)))
        for kind, iv in ivals.items():
            iv.sort()
            union, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
            for a, b in iv[1:]:
                if a > cur_e:
                    union += cur_e - cur_s
                    cur_s, cur_e = a, b
                else:
                    cur_e = max(cur_e, b)
            out[kind]["union_ms"] = union + (cur_e - cur_s)
        return out


CONV_TIMER = None            # set to a KernelTimer by bench.py
TIMER_SHAPES = False         # tools/layer_times.py: one timer kind per layer shape
# Linear layers of the TRAINED router (DM-Router proj_1/2/3, channel gating; forward and data-gradient GEMMs):
#   "fp16x3"  split-fp16 x3 on the grouped conv kernel with BOTH operands prescaled by a power of two computed on the
#             device from max|.| (gradients of 1e-6 would otherwise fall into fp16's subnormal range)
#   "f32"     exact fp32 MFMA
ROUTER_GEMM_PRECISION = "fp16x3"
SVTR_FUSED_ATTENTION = os.environ.get("MRN_SVTR_ATTENTION", "fused") == "fused"   # frozen SVTR experts: mrn_svtr_attention_f32
SVTR_ATTENTION_X3 = os.environ.get("MRN_SVTR_ATTENTION_PRECISION", "fp16x3") == "fp16x3"   # frozen experts: x3 products ("f32": exact)
# A/B switches (environment variables, read once at import): the defaults are the measured winners; tools/ and DESIGN.md
# section 4 quote the runs.
RECURRENT_X3 = os.environ.get("MRN_RECURRENT", "fp16x3") == "fp16x3"   # frozen experts' LSTM recurrences on the f16 MFMA
ROUTER_WGRAD_X3 = os.environ.get("MRN_WGRAD", "fp16x3") == "fp16x3"     # weight-gradient GEMMs too (A/B switch)
ROUTER_TOKEN_WGRAD_X3 = os.environ.get("MRN_TOKEN_WGRAD", "fp16x3") == "fp16x3"     # the token-mixing weight gradient of the DM-Router too (A/B switch)

# Arithmetic of a frozen expert's convolutions OUTSIDE a lock-step group (a single network: validation() of a one-network learner,
# LwF's previous network, ...; the lock-step groups of modules/expert_group.py always run the grouped split-fp16 x3 kernels):
#   "auto" / "fp16x3"  (default) split-fp16 x3 on v_mfma_f32_32x32x16_f16 -- 22-bit products, power-of-two weight prescale -- on the
#            grouped kernel with G = 1 for every conv with Cin % 32 == 0, exact fp32 otherwise.  Measured on the golden vectors the router
#            weights / fused logits sit at the SAME distance from the reference as the exact fp32 kernel (TRBA 6.5e-5 / 6.4e-5 vs
#            5.7e-5 / 7.9e-5 -- that floor is the TPS grid's fp32 conditioning; CRNN 5.7e-7 vs 3.6e-7), also with the TPS localisation
#            network on it.
#   "f32"    exact fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere, per expert
# (Rounds 1-3 also carried split-bf16 x3 / plain bf16 / plain fp16 per-expert kernels -- 16-bit products sit at the edge of the 1e-4 band
# on TRBA -- removed in round 4 together with their 128x128x32 kernel family.)
CONV_PRECISION = "auto"
# Convs that are being TRAINED (loop A: forward and data gradients; weight gradients stay on the exact-fp32 kernel):
#   "fp16x3s" (default) split-fp16 x3 on the grouped kernel with BOTH operands prescaled by a device-computed power of two,
#             so gradient operands of 1e-6 keep 22-bit products; every parameter gradient of a CRNN / TRBA expert matches
#             torch autograd on the oracle inside the bands of tests/test_model_gpu.py::test_loop_a_*
#   "f32"     exact fp32 MFMA.  (Unscaled "fp16x3" fails the TRBA gradient test: small gradients fall into fp16 subnormals.)
TRAIN_CONV_PRECISION = os.environ.get("MRN_TRAIN_PRECISION", "fp16x3s")
TRAIN_WGRAD_X3 = os.environ.get("MRN_TRAIN_WGRAD", "fp16x3s") == "fp16x3s"   # weight gradients on the same path
WGRAD_WINDOWS = os.environ.get("MRN_WGRAD_WINDOWS", "1") == "1"   # 3x3 / s1 / p1 weight gradients without an im2col (A/B switch)
LOCNET_CONV_PRECISION = None   # TPS localisation network: None = follow CONV_PRECISION ("f32" to pin it exact)
_ZERO_PAGES = {}


_POW2_WS = {}


def _pow2_ws():
    """two zeroed words per (device, stream): the self-resetting workspace of mrn_pow2_scale_f32 (stream-ordered reuse)"""
    key = _dev_stream()
    ws = _POW2_WS.get(key)
    if ws is None:
        ws = torch.zeros(2, device=torch.device("cuda", key[0]), dtype=torch.int32)
        _POW2_WS[key] = ws
    return ws.data_ptr()


_AMAX_WS = {}


def _amax_ws():
    """64 zeroed words per (device, stream): the slots producers fold per-block maxima into (mrn_pow2_finalize_f32 clears them)"""
    key = _dev_stream()
    ws = _AMAX_WS.get(key)
    if ws is None:
        ws = torch.zeros(64, device=torch.device("cuda", key[0]), dtype=torch.int32)
        _AMAX_WS[key] = ws
    return ws.data_ptr()


def pow2_finalize(target):
    """{s, 1/s} from the maxima a producer just folded into this stream's 64 words (_amax_ws), which are put back to zero"""
    dev, st = _dev_stream()
    sc = torch.empty(2, device=torch.device("cuda", dev), dtype=torch.float32)
    call("mrn_pow2_finalize_f32", float(target), _p(sc), _amax_ws(), st)
    return sc


def _zero_page(device):
    z = _ZERO_PAGES.get(device)
    if z is None:
        z = torch.zeros(64, device=device, dtype=torch.float32)
        _ZERO_PAGES[device] = z
    return z


FP16_WEIGHT_PEAK = 16384.0   # fp16 weight planes are prescaled (power of two) so that max|w| lands in (8192, 16384]


class PackedConvWeight:
    """[O,kh,kw,I] fp32 weight plus its lazily built split-fp16 operand stacks for the grouped x3 kernel (HL32 lines or the Winograd-domain
    form, with their power-of-two prescale); modules/_nn.packed_weight rebuilds the object when the parameter's version changes."""

    def __init__(self, ohwi):
        self.ohwi = ohwi
        self.shape = ohwi.shape
        self._operand = {}

    def x3_operand(self, wino):
        """-> (stack, scale) for conv_x3.hip with G = 1: pack_weights_wino (F(WINO_R,3)) or pack_weights_hl32"""
        key = (bool(wino), WINO_R)
        got = self._operand.get(key)
        if got is None:
            w = self.ohwi.contiguous()
            got = tuple(pack_weights_wino([w], WINO_R)) if wino else tuple(pack_weights_hl32([w]))
            self._operand[key] = got
        return got


# (device index, raw handle of the current stream) through two C calls: torch.cuda.current_stream() builds a Stream object through five
# Python frames, 4-8 us a call, ~1300 calls in a launch-bound SVTR loop-A step.  The two torch._C entry points are private: resolved ONCE
# here, with the public API as the fallback of a torch that renames them.
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)
_GET_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if _GET_DEVICE is None or _GET_RAW_STREAM is None:
    def _dev_stream():
        dev = torch.cuda.current_device()
        return dev, torch.cuda.current_stream(dev).cuda_stream

    def _stream():
        return torch.cuda.current_stream().cuda_stream
else:
    def _dev_stream():
        dev = _GET_DEVICE()
        return dev, _GET_RAW_STREAM(dev)

    def _stream():
        return _GET_RAW_STREAM(_GET_DEVICE())


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
    timed = CONV_TIMER is not None and TIMER_SHAPES
    t0 = CONV_TIMER.begin() if timed else None
    call("mrn_gemm_f32", _p(A), _p(W), _p(bias), _p(residual), _p(C), M, N, K, batch,
         sA[0], sA[1], sA[2], sW[0], sW[1], sW[2], sC[0], sC[1], sC[2], bias_batch_stride, bias_axis,
         act, int(accumulate), float(alpha), _stream())
    if timed:
        CONV_TIMER.end(t0, 2.0 * batch * M * N * K, "gemm_f32|b%d M%d N%d K%d sA%s sW%s sC%s" % (batch, M, N, K, sA[1:], sW[1:], sC[1:]))
    return C


def rows2d(t):
    """View [..., K] (contiguous last dim, uniformly strided leading dims) as a 2-D [rows, K] strided tensor."""
    if t.dim() == 2:
        assert t.stride(1) == 1 or t.shape[1] == 1
        return t
    if t.is_contiguous():
        return t.view(-1, t.shape[-1])
    assert t.stride(-1) == 1 or t.shape[-1] == 1, "last dim must be contiguous"
    rs, n = None, 1
    for d in range(t.dim() - 2, -1, -1):        # every leading dim (size > 1) must continue the same row stride
        if t.shape[d] == 1:
            continue
        if rs is None:
            rs = t.stride(d)
        else:
            assert t.stride(d) == rs * n, f"non-uniform row stride {t.stride()} for shape {tuple(t.shape)}"
        n *= t.shape[d]
    if rs is None:
        rs = t.shape[-1]
    return t.as_strided((n, t.shape[-1]), (rs, 1))


def linear(x, weight, bias=None, act=ACT_NONE, residual=None, out=None):
    """y = act(x @ weight.T + bias (+ residual)); x [..., K], weight [N, K] (row stride free), out may be strided."""
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = rows2d(x)
    M = x2.shape[0]
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    o2 = rows2d(out)
    assert o2.shape == (M, N) and weight.stride(1) == 1
    r2 = None
    if residual is not None:
        r2 = rows2d(residual)
        assert r2.stride() == o2.stride(), "residual must share the output's layout"
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
    return PackedConvWeight(out)


def conv_out_hw(H, W, k, s, p):
    return (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1


def conv2d_nhwc(x, w_ohwi, bias=None, stride=(1, 1), padding=(0, 0), act=ACT_NONE, want_stats=False, precision=None,
                out=None, stats_out=None, x_scale=None):
    """x [B,H,W,Cin] -> y [B,Ho,Wo,Cout]; returns (y, stats or None) where stats are per-128-row-block partials.
    w_ohwi: [O,kh,kw,I] fp32 tensor or a PackedConvWeight."""
    packed = w_ohwi if isinstance(w_ohwi, PackedConvWeight) else None
    if packed is not None:
        w_ohwi = packed.ohwi
    _chk(x, w_ohwi, bias)
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = w_ohwi.shape
    precision = precision or CONV_PRECISION
    if precision == "fp16x3s":
        # range-safe split-fp16 x3 on the grouped kernel (G = 1): both operands prescaled by a device-computed power of
        # two -- the precision of TRAINED convolutions, whose gradient operands can be 1e-6-sized
        if Cin % 32 == 0 and out is None and stats_out is None:
            return conv2d_x3_scaled(x, w_ohwi, bias, stride, padding, act, want_stats, sx=x_scale)
        precision = "f32"
    if precision in ("auto", "fp16x3") and Cin % 32 == 0 and out is None and stats_out is None:
        # a single (not lock-step) frozen expert, e.g. validation() of a one-network learner or LwF's previous network: the grouped
        # split-fp16 x3 kernel with G = 1 -- plain HL32 operands, or the Winograd form where the lock-step path would use it
        return conv2d_x3_frozen(x, packed if packed is not None else PackedConvWeight(w_ohwi), bias, stride, padding, act, want_stats)
    Ho, Wo = conv_out_hw(H, W, (kh, kw), stride, padding)
    y = out if out is not None else torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    assert y.is_contiguous() and y.numel() == B * Ho * Wo * Cout
    stats = None
    if want_stats:
        n = call("mrn_conv2d_stats_floats", B, Ho, Wo, Cout)
        stats = stats_out if stats_out is not None else torch.empty(n, device=x.device, dtype=torch.float32)
        assert stats.is_contiguous() and stats.numel() == n
    timed = CONV_TIMER is not None and Cout > 64          # the 128x128-tile kernel
    Kred = kh * kw * Cin
    kind = "f32"
    t0 = CONV_TIMER.begin() if timed else None
    call("mrn_conv2d_nhwc_f32", _p(x), _p(w_ohwi), _p(bias), _p(y), _p(stats), B, H, W, Cin, Cout, kh, kw,
         stride[0], stride[1], padding[0], padding[1], act, _stream())
    if timed:
        CONV_TIMER.end(t0, 2.0 * B * Ho * Wo * Cout * Kred, kind)
    return y, stats


# ---------------------------------------------------------------------------------------------------------
# grouped split-fp16 x3 convolution on HL32 operands (conv_x3.hip)
# ---------------------------------------------------------------------------------------------------------
# Products per term of the grouped conv / Linear of the FROZEN experts (conv_x3.hip): 3 = split-fp16 x3 (default: the 1e-4 parity
# mode); 1 = hi x hi only, i.e. plain fp16 products with fp32 accumulation -- the reduced-precision mode of BASELINE configs 2
# and 5 (bench.py --precision fp16; measured tolerance and routing agreement: tests/test_model_gpu.py::test_reduced_precision_*).
# The router's and the heads' Linear layers always run 3 products (functional.x3_linear / x3_wgrad); the TRAINED convolutions (forward,
# data gradient, weight gradient -- 95 % of a training step's flops) follow TRAIN_PRODUCTS: fp16 products with fp32 accumulation and
# per-operand power-of-two range scaling, i.e. what mixed-precision training computes (BASELINE config 5 "fp16 MFMA").
X3_PRODUCTS = 1 if os.environ.get("MRN_X3_PRODUCTS") == "1" else 3
TRAIN_PRODUCTS = 1 if os.environ.get("MRN_TRAIN_PRODUCTS") == "1" else 3
X3_SMALL_TILE_MAX_K = 1200     # measured on MI355X (tools/bench_conv_x3.py): +3.5 % at K = 1152, +7.6 % at K = 576


def x3_tile(Cout, K, M=None, G=1):
    """(tile_m, tile_n) of the grouped conv: 256x256 for wide outputs, 256x128 otherwise; 128x128 (two workgroups per CU)
    for short reductions, where the prologue / epilogue of one tile overlaps the main loop of its neighbour.  When the
    tile count is known (M = rows per group), a 256x256 launch that would leave a ragged last round of workgroups on the
    256 CUs (e.g. 520 tiles = 2.03 rounds for ONE expert's 4x65 layers in loop A) falls back to the tile with the best
    (relative kernel speed x round efficiency)."""
    force = os.environ.get("MRN_X3_TILE")
    if force:
        return tuple(int(v) for v in force.split("x"))
    if Cout <= 64:
        return 256, 64
    if Cout < 256:
        return (128, 128) if K <= X3_SMALL_TILE_MAX_K else (256, 128)
    if M is None:
        return 256, 256

    def eff(tm, tn, slots):
        tiles = G * ((M + tm - 1) // tm) * ((Cout + tn - 1) // tn)
        rounds = (tiles + slots - 1) // slots
        return tiles / (rounds * slots)
    cands = [((256, 256), 1.00 * eff(256, 256, 256)), ((256, 128), 0.87 * eff(256, 128, 256)), ((128, 128), 0.88 * eff(128, 128, 512))]
    return max(cands, key=lambda c: c[1])[0]


def split_hl32(x, scale=None):
    """fp32 [..., C] (C % 32 == 0, contiguous) -> HL32 bytes (same byte count), one 128-B line per (row, 32 channels);
    scale: device float[2] {s, 1/s}: the halves hold s * x (pass the same tensor to conv2d_x3 as x_scale)"""
    _chk(x)
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty(x.numel() * 4, device=x.device, dtype=torch.uint8)
    call("mrn_split_hl32_f32", _p(x), _p(out), rows, C, _p(scale), _stream())
    return out


SPLIT_T_COLSUM = os.environ.get("MRN_SPLIT_T_COLSUM", "1") == "1"      # A/B switch: 0 = the bias gradient as its own column-sum passes


def split_hl32_t(x2, splits, scale=None, rows_padded=None, colsum_out=None, accumulate=False):
    """fp32 [rows, C] -> `splits` transposed HL32 matrices [splits][C][rows_padded/splits/32][128 B] of scale[0] * x
    (rows beyond x2's are zero).  colsum_out [C]: the same pass leaves the column sums of x there (added when `accumulate`): the bias
    gradient of the Linear layer whose weight gradient this operand feeds"""
    _chk(x2)
    rows, C = x2.shape
    rows_padded = rows if rows_padded is None else rows_padded
    assert x2.is_contiguous() and rows_padded % (32 * splits) == 0
    out = torch.empty(rows_padded * C * 4, device=x2.device, dtype=torch.uint8)
    if colsum_out is None:
        call("mrn_split_hl32_t_f32", _p(x2), _p(out), rows, rows_padded, C, splits, _p(scale), _stream())
        return out
    assert colsum_out.is_contiguous() and colsum_out.numel() == C and colsum_out.dtype == torch.float32
    chunks = call("mrn_split_hl32_t_colsum_chunks", rows_padded, C)
    part = torch.empty(chunks * C, device=x2.device, dtype=torch.float32) if chunks > 1 else None
    call("mrn_split_hl32_t_colsum_f32", _p(x2), _p(out), rows, rows_padded, C, splits, _p(scale), _p(colsum_out), int(accumulate), _p(part),
         _stream())
    return out


def conv2d_wgrad_x3(dy, x, ksize, stride, padding, dy_scale=None, x_scale=None):
    """conv weight gradient on the split-fp16 x3 path: dW [Cout, kh, kw, Cin] = sum over output pixels of dy (x) im2col(x);
    both operands are transposed-split with device prescales (reduction axis = pixels), the GEMM is ONE grouped launch
    over (split-K chunk, tap); the partial slabs are reduced by a column-sum pass."""
    B, Ho, Wo, Cout = dy.shape
    _, H, W, Cin = x.shape
    kh, kw = ksize
    taps = kh * kw
    P = B * Ho * Wo
    blocks = (P + 31) // 32
    tiles = ((Cout + 255) // 256) * ((Cin + 255) // 256) * taps
    S = max(1, min(512 // tiles if tiles < 512 else 1, max(blocks // 8, 1)))
    rps = ((blocks + S - 1) // S) * 32
    Pp = rps * S
    dy2 = dy.contiguous().view(P, Cout)
    sd = dy_scale if dy_scale is not None else pow2_scale(dy2)
    sx = x_scale if x_scale is not None else pow2_scale(x)
    a_hl = split_hl32_t(dy2, S, sd, rows_padded=Pp)                       # [S][Cout][rps/32][128]
    w_hl = torch.empty(S * taps * Cin * rps * 4, device=x.device, dtype=torch.uint8)
    call("mrn_im2col_t_hl32_f32", _p(x.contiguous()), _p(w_hl), B, H, W, Cin, kh, kw, stride[0], stride[1], padding[0], padding[1],
         Pp, S, _p(sx), _stream())
    G = S * taps
    part = torch.empty(S, taps, Cout, Cin, device=x.device, dtype=torch.float32)
    conv2d_x3(a_hl, G, False, Cout, 1, 1, rps, w_hl, sx.view(1, 2).expand(G, 2).contiguous(), Cin, (1, 1), x_scale=sd,
              out=part, x_group_div=taps, products=TRAIN_PRODUCTS)
    dw = colsum(part.view(S, taps * Cout * Cin)).view(taps, Cout, Cin) if S > 1 else part[0]
    return dw.permute(1, 0, 2).contiguous().view(Cout, kh, kw, Cin)


_WINDOW_TABLES = {}


def wgrad_windows_supported(dy, x, ksize, stride, padding):
    """the im2col-free weight gradient covers 3x3 / stride 1 / pad 1 convs on maps at least two rows high whose image rows fill
    whole 32-pixel lines (B * W % 32 == 0: every batch size that is a multiple of 32)"""
    B, H, W, Cout = dy.shape
    return (tuple(ksize) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1) and H >= 2 and (B * W) % 32 == 0
            and Cout % 4 == 0 and x.shape[-1] % 4 == 0 and tuple(x.shape[:3]) == (B, H, W))


def transpose_oy_hl32(x, shift, scale, out=None):
    """NHWC fp32 [B,H,W,C] -> HL32 matrix [C][B*H*W/32][128 B] in image-row-major pixel order, shifted along x (zero fill)"""
    B, H, W, C = x.shape
    lines = (B * H * W + 31) // 32
    if out is None:
        out = torch.empty(C * lines * 128, device=x.device, dtype=torch.uint8)
    call("mrn_transpose_oy_hl32_f32", _p(x), _p(out), B, H, W, C, int(shift), _p(scale), _stream())
    return out


def conv2d_wgrad_x3_windows(dy, x, dy_scale=None, x_scale=None):
    """dW [Cout,3,3,Cin] of a 3x3 / stride 1 / pad 1 conv on the split-fp16 x3 path WITHOUT an im2col: dy^T and three x-shifted
    copies of x^T in image-row-major pixel order (a kernel-row offset is then a whole number of 128-byte lines), one grouped GEMM
    over (split-K chunk, tap) K-windows of those matrices, partial slabs reduced by a column-sum pass.  3x the activation is
    written instead of the 9x of mrn_im2col_t_hl32_f32."""
    B, H, W, Cout = dy.shape
    Cin = x.shape[-1]
    P = B * H * W
    lines, bwl = P // 32, (B * W) // 32
    dev = x.device
    dy, x = dy.contiguous(), x.contiguous()
    sd = dy_scale if dy_scale is not None else pow2_scale(dy)
    sx = x_scale if x_scale is not None else pow2_scale(x)
    a_hl = transpose_oy_hl32(dy, 0, sd)
    per = Cin * lines * 128
    w_hl = torch.empty(3 * per, device=dev, dtype=torch.uint8)
    call("mrn_transpose_oy3_hl32_f32", _p(x), _p(w_hl), B, H, W, Cin, _p(sx), _stream())       # shifts -1, 0, +1 from one read
    tiles = ((Cout + 255) // 256) * ((Cin + 255) // 256) * 9
    small = Cout <= 128 and Cin <= 64 and TRAIN_PRODUCTS == 3 and WGRAD_SMALL_TILE
    # (64 x 64 tiles: 32 KiB of LDS, several workgroups per CU -- four times the split-K chunks fill them)
    S = max(1, min((2048 if small else 512) // tiles if tiles < 512 else 1, max((lines - bwl) // 8, 1)))
    key = (lines, bwl, Cin, S, dev)
    tab = _WINDOW_TABLES.get(key)
    if tab is None:
        rows = []
        for s_ in range(S):
            for tap in range(9):
                ky, kx = divmod(tap, 3)
                lo = bwl if ky == 0 else 0                      # dy rows y >= 1 pair with x rows y - 1
                hi = lines - bwl if ky == 2 else lines          # dy rows y <= H - 2 pair with x rows y + 1
                L = hi - lo
                start, end = lo + (L * s_) // S, lo + (L * (s_ + 1)) // S
                assert end > start
                rows.append([start * 128, (kx * Cin * lines + start + (ky - 1) * bwl) * 128, end - start])
        tab = torch.tensor(rows, dtype=torch.int64).to(dev)
        _WINDOW_TABLES[key] = tab
    G = S * 9
    part = torch.empty(S, 9, Cout, Cin, device=dev, dtype=torch.float32)
    tile_m, tile_n = x3_tile(Cin, 32 * ((lines + S - 1) // S), M=Cout, G=G)
    if small:
        tile_m, tile_n = (64 if Cout <= 64 else 128), 64      # the first layers: a 256-row tile would stage (and multiply) mostly padding
    timed = CONV_TIMER is not None
    t0 = CONV_TIMER.begin() if timed else None
    call("mrn_gemm_x3_windows_hl32", _p(a_hl), Cout * lines * 128, lines, _p(w_hl), 3 * per, lines, _p(tab), G, Cout, Cin,
         _p(_zero_page(dev)), _p(sx), _p(sd), _p(part), tile_m, tile_n, int(TRAIN_PRODUCTS), _stream())
    if timed:
        CONV_TIMER.end(t0, 2.0 * 9 * Cout * Cin * P, "fp16x3/x3g%dx%d" % (tile_m, tile_n), 4.0 * (4 * P * Cin + P * Cout + G * Cout * Cin))
    dw = colsum(part.view(S, 9 * Cout * Cin)).view(9, Cout, Cin) if S > 1 else part[0]
    return dw.permute(1, 0, 2).contiguous().view(Cout, 3, 3, Cin)


TRAIN_WGRAD_WINO = os.environ.get("MRN_TRAIN_WGRAD_WINO", "1") == "1"     # weight gradients of the Winograd-eligible trained layers in the Winograd domain


def wgrad_wino_supported(dy, x, ksize, stride, padding):
    """the Winograd-domain weight gradient covers what the forward's Winograd form covers, on maps at least two rows high whose image rows
    fill whole 32-group lines (B * ceil(W / 4) % 32 == 0), with operands range-scaled for the transforms (TRAIN_OPERAND_PEAK)"""
    B, H, W, Cout = dy.shape
    Cin = x.shape[-1]
    return (TRAIN_WGRAD_WINO and TRAIN_WINO and WINO_R == 4 and TRAIN_PRODUCTS == 3 and wino_eligible(ksize, stride, padding, Cin, Cout, 3)
            and H >= 2 and (B * ((W + 3) // 4)) % 32 == 0 and Cout % 4 == 0 and tuple(x.shape[:3]) == (B, H, W))


def conv2d_wgrad_x3_wino(dy, x, dy_scale=None, x_scale=None, acc_oihw=None):
    """dW [Cout,3,3,Cin] of a 3x3 / stride 1 / pad 1 conv in the Winograd domain (F(4,3) along W): dU_m[ky] = sum over column groups of
    (A dy)_m (x) (B^T x)_m shifted by ky - 1 image rows -- 18 K-windows of a quarter of the pixel count on the grouped x3 GEMM -- then
    dW[ky][kx] = sum_m G[m][kx] dU_m[ky].  Half the matrix work of conv2d_wgrad_x3_windows and 3x instead of 4x the operand bytes.
    acc_oihw: the parameter's [Cout,Cin,3,3] gradient -- dW is ADDED there by the finish launch and None is returned"""
    B, H, W, Cout = dy.shape
    Cin = x.shape[-1]
    Wq = (W + 3) // 4
    Pq = B * H * Wq
    lines, bwl = Pq // 32, (B * Wq) // 32
    dev = x.device
    dy, x = dy.contiguous(), x.contiguous()
    sd = dy_scale if dy_scale is not None else pow2_scale(dy, TRAIN_OPERAND_PEAK)
    sx = x_scale if x_scale is not None else pow2_scale(x, TRAIN_OPERAND_PEAK)
    a_bytes, w_bytes = 6 * Cout * lines * 128, 6 * Cin * lines * 128
    a_hl = torch.empty(a_bytes, device=dev, dtype=torch.uint8)
    w_hl = torch.empty(w_bytes, device=dev, dtype=torch.uint8)
    call("mrn_transpose_oy_wino_hl32_f32", _p(dy), _p(a_hl), B, H, W, Cout, 1, _p(sd), _stream())
    call("mrn_transpose_oy_wino_hl32_f32", _p(x), _p(w_hl), B, H, W, Cin, 0, _p(sx), _stream())
    tiles = ((Cout + 255) // 256) * ((Cin + 255) // 256) * 18
    S = max(1, min(512 // tiles if tiles < 512 else 1, max((lines - bwl) // 8, 1)))
    key = ("wino", lines, bwl, Cin, Cout, S, dev)
    tab = _WINDOW_TABLES.get(key)
    if tab is None:
        rows = []
        for s_ in range(S):
            for m in range(6):
                for ky in range(3):
                    lo = bwl if ky == 0 else 0                      # dy rows y >= 1 pair with x rows y - 1
                    hi = lines - bwl if ky == 2 else lines          # dy rows y <= H - 2 pair with x rows y + 1
                    L = hi - lo
                    start, end = lo + (L * s_) // S, lo + (L * (s_ + 1)) // S
                    assert end > start
                    rows.append([(m * Cout * lines + start) * 128, (m * Cin * lines + start + (ky - 1) * bwl) * 128, end - start])
        tab = torch.tensor(rows, dtype=torch.int64).to(dev)
        _WINDOW_TABLES[key] = tab
    G = S * 18
    part = torch.empty(S, 18, Cout, Cin, device=dev, dtype=torch.float32)
    tile_m, tile_n = x3_tile(Cin, 32 * ((lines + S - 1) // S), M=Cout, G=G)
    timed = CONV_TIMER is not None
    t0 = CONV_TIMER.begin() if timed else None
    call("mrn_gemm_x3_windows_hl32", _p(a_hl), a_bytes, lines, _p(w_hl), w_bytes, lines, _p(tab), G, Cout, Cin,
         _p(_zero_page(dev)), _p(sx), _p(sd), _p(part), tile_m, tile_n, int(TRAIN_PRODUCTS), _stream())
    if timed:
        CONV_TIMER.end(t0, 2.0 * 9 * Cout * Cin * B * H * W, "fp16x3/x3g%dx%d" % (tile_m, tile_n), 4.0 * (6 * Pq * (Cin + Cout) + G * Cout * Cin))
    if acc_oihw is not None:
        assert acc_oihw.is_contiguous() and tuple(acc_oihw.shape) == (Cout, Cin, 3, 3)
        call("mrn_wino_wgrad_finish_f32", _p(part), _p(acc_oihw), S, Cout, Cin, 1, _stream())
        return None
    dw = torch.empty(Cout, 3, 3, Cin, device=dev, dtype=torch.float32)
    call("mrn_wino_wgrad_finish_f32", _p(part), _p(dw), S, Cout, Cin, 0, _stream())
    return dw


def pow2_scale(x, target=FP16_WEIGHT_PEAK):
    """device float[2] = {s, 1/s}, s = the largest power of two with s * max|x| <= target (no host sync)"""
    _chk(x)
    assert x.is_contiguous()
    scale = torch.empty(2, device=x.device, dtype=torch.float32)
    call("mrn_pow2_scale_f32", _p(x), x.numel(), float(target), _p(scale), _pow2_ws(), _stream())
    return scale


def pack_weights_hl32(ws, scale=None):
    """list of G [O,kh,kw,I] fp32 weights (same shape) -> (HL32 weight stack bytes [G][O][I/32][taps][128], scale [G,2]).
    scale: [G,2] power-of-two prescales already known for these tensors (e.g. of the same weights in another layout)"""
    O, kh, kw, I = ws[0].shape
    G = len(ws)
    dev = ws[0].device
    per = O * kh * kw * I * 4
    out = torch.empty(G * per, device=dev, dtype=torch.uint8)
    known = scale is not None
    if not known:
        scale = torch.empty(G, 2, device=dev, dtype=torch.float32)
    for g, w in enumerate(ws):
        _chk(w)
        assert tuple(w.shape) == (O, kh, kw, I) and w.is_contiguous()
        if not known:
            call("mrn_pow2_scale_f32", _p(w), w.numel(), FP16_WEIGHT_PEAK, scale[g].data_ptr(), _pow2_ws(), _stream())
        call("mrn_pack_weight_hl32", _p(w), out.data_ptr() + g * per, O, kh * kw, I, scale[g].data_ptr(), _stream())
    return out, scale


def conv2d_x3(x_hl, G, shared_input, B, H, W, Cin, w_hl, w_scale, Cout, ksize, stride=(1, 1), padding=(0, 0), bias=None,
              act=ACT_NONE, want_stats=False, out=None, out_row_stride=0, out_group_stride=0, residual=None, x_scale=None,
              x_group_div=1, hl_only=False, products=3, ch_scale=None, ch_shift=None, residual_hl=None, also_hl=False, amax_ws=None):
    """Grouped conv on HL32 operands -> (y [G,B,Ho,Wo,Cout] fp32, stats or None).  With `out` and the two strides (floats)
    the rows of group g land at out.data_ptr + g * out_group_stride + row * out_row_stride.
    hl_only: the result is written ONLY as the HL32 operand of the next GEMM (returned in place of y); also_hl: fp32 AND HL32
    (returned as the pair (y, y_hl)).  ch_scale / ch_shift [G,Cout]: eval-mode BatchNorm folded into the epilogue;
    residual_hl: the identity shortcut as HL32 lines (needs an HL32 result)."""
    kh, kw = ksize
    Ho, Wo = conv_out_hw(H, W, ksize, stride, padding)
    dev = x_hl.device
    y_hl = None
    if hl_only or also_hl:
        assert (out is None or also_hl) and Cout % 32 == 0
        y_hl = torch.empty(G * B * Ho * Wo * Cout * 4, device=dev, dtype=torch.uint8)
    if hl_only:
        y = None
    else:
        y = out if out is not None else torch.empty(G, B, Ho, Wo, Cout, device=dev, dtype=torch.float32)
    tile_m, tile_n = x3_tile(Cout, kh * kw * Cin, M=B * Ho * Wo, G=G)
    stats = None
    if want_stats:
        stats = torch.empty(call("mrn_conv2d_x3_stats_floats", G, B, Ho, Wo, Cout, tile_m), device=dev, dtype=torch.float32)
    gstride = 0 if shared_input else B * H * W * Cin * 4
    timed = CONV_TIMER is not None
    t0 = CONV_TIMER.begin() if timed else None
    call("mrn_conv2d_x3_hl32", _p(x_hl), _p(w_hl), _p(_zero_page(dev)), _p(bias), _p(residual), _p(y), _p(stats), _p(w_scale),
         _p(x_scale), G, gstride,
         B, H, W, Cin, Cout, kh, kw, stride[0], stride[1], padding[0], padding[1], act, tile_m, tile_n, out_row_stride, out_group_stride,
         x_group_div, _p(y_hl), int(products), _p(ch_scale), _p(ch_shift), _p(residual_hl), amax_ws, _stream())
    if timed:
        # algorithmic bytes: every operand element once (HL32 = 4 B / element, like fp32) + the fp32 result
        nbytes = 4.0 * ((1 if shared_input else G) * B * H * W * Cin + G * Cout * kh * kw * Cin + G * B * Ho * Wo * Cout)
        kind = ("fp16x3" if products == 3 else "fp16") + "/x3g%dx%d" % (tile_m, tile_n)
        if TIMER_SHAPES:
            kind += "|G%d B%d %dx%d %d->%d k%dx%d s%d%d" % (G, B, H, W, Cin, Cout, kh, kw, stride[0], stride[1])
        CONV_TIMER.end(t0, 2.0 * G * B * Ho * Wo * Cout * kh * kw * Cin, kind, nbytes)
    return (y_hl if hl_only else ((y, y_hl) if also_hl else y)), stats


# 1-D Winograd F(R,3) along W for the frozen experts' 3x3 / stride 1 / pad 1 convolutions with Cin >= WINO_MIN_CIN whose input
# comes from a train-mode BatchNorm-apply pass (csrc/conv_x3.hip WINO, csrc/group_ops.hip): R = 4 halves the matrix work of
# the dominant 512 -> 512 layers.  MRN_WINO=0 switches it off (A/B), MRN_WINO=2 selects F(2,3).
WINO_R = int(os.environ.get("MRN_WINO", "4"))
WINO_CHECK = os.environ.get("MRN_WINO_CHECK") == "1"      # debug: every row-block Winograd launch is re-run on the x3 kernel and compared
WINO_MIN_CIN = int(os.environ.get("MRN_WINO_MIN_CIN", "128"))
TRAIN_WINO = os.environ.get("MRN_TRAIN_WINO", "1") == "1"      # the trained convolutions of loop A too (forward + data gradient)
# range target of the power-of-two scale of a TRAINED convolution's activation / gradient operand: the Winograd input transform B^T
# amplifies an operand by up to 10x (row sums of |B^T|, F(4,3)), so its prescale leaves a factor 16 of fp16 headroom below the 16384
# the plain split aims at (the weight side needs none: the folded row scales keep |G g| <= 1.17 max|g|)
TRAIN_OPERAND_PEAK = 16384.0 / 16 if TRAIN_WINO else 16384.0


# Reduced-precision mode (X3_PRODUCTS / TRAIN_PRODUCTS == 1): the Winograd layers run the row-block kernel on PLAIN fp16 operands, 64 channels
# per 128-byte line ("d16": a third of the MFMAs on half the operand bytes; csrc/conv_wino.hip DENSE).  MRN_WINO_DENSE=0: the round-5 form of
# the mode (no Winograd, direct one-product convolutions on HL32 lines whose lo halves are not read) -- the A/B partner.
WINO_DENSE = os.environ.get("MRN_WINO_DENSE", "1") == "1"
# the 16-bit type of those operands: fp16 (default: 11 significand bits) or bfloat16 (bench.py --precision bf16: the literal "bf16" of BASELINE
# config 2 as a comparison instantiation -- same kernel, v_mfma_f32_32x32x16_bf16, 8 significand bits)
REDUCED_BF16 = os.environ.get("MRN_REDUCED_BF16", "0") == "1"


def wino_dense(products=None):
    """does a Winograd layer of a `products`-product pass (None: the frozen experts' X3_PRODUCTS) run on plain-fp16 operands"""
    return (X3_PRODUCTS if products is None else products) == 1 and WINO_DENSE and WINO_R == 4


def wino_eligible(ksize, stride, padding, Cin, Cout, products=None):
    """products: of the pass that asks (None: X3_PRODUCTS, the frozen experts; the trained layers pass TRAIN_PRODUCTS)"""
    products = X3_PRODUCTS if products is None else products
    ok = (WINO_R in (2, 4) and tuple(ksize) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
          and Cin % 32 == 0 and Cin >= WINO_MIN_CIN and Cout >= 64)
    if products == 3:
        return ok
    return ok and wino_dense(products) and Cin % 64 == 0 and Cout % 4 == 0          # (the d16 form has no fallback kernel: row-block geometry only)


def pack_weights_wino(ws, R, scale=None, dense=None):
    """list of G [O,3,3,I] fp32 weights -> (Winograd-domain HL32 stack bytes [G][O][R+2][I/32][3][128], scale [G,2]); dense (None:
    wino_dense()): plain fp16 [G][O][6][I/64][3][128]"""
    O, kh, kw, I = ws[0].shape
    assert (kh, kw) == (3, 3)
    G = len(ws)
    dev = ws[0].device
    dense = wino_dense() if dense is None else dense
    per = O * (R + 2) * 3 * I * (2 if dense else 4)
    out = torch.empty(G * per, device=dev, dtype=torch.uint8)
    known = scale is not None
    if not known:
        scale = torch.empty(G, 2, device=dev, dtype=torch.float32)
    for g, w in enumerate(ws):
        _chk(w)
        assert tuple(w.shape) == (O, 3, 3, I) and w.is_contiguous()
        if not known:
            call("mrn_pow2_scale_f32", _p(w), w.numel(), FP16_WEIGHT_PEAK, scale[g].data_ptr(), _pow2_ws(), _stream())
        if dense:
            call("mrn_pack_weight_wino_d16", _p(w), out.data_ptr() + g * per, O, I, scale[g].data_ptr(), int(REDUCED_BF16), _stream())
        else:
            call("mrn_pack_weight_wino_hl32", _p(w), out.data_ptr() + g * per, O, I, R, scale[g].data_ptr(), _stream())
    return out, scale


def bn_apply_wino_grouped(y, scale, shift, R, relu=True, residual=None, residual_hl=None, want_f32=False, want_hl=False, prescale=None,
                          dense=None):
    """y [G,B,H,W,C] fp32 -> (fp32 result (a NEW tensor) or None, HL32 bytes or None, Winograd-domain operand bytes
    [G][B][H][ceil(W/R)][R+2][C/32][128]); dense (None: wino_dense()): the operand as plain fp16 [..][R+2][C/64][128]"""
    G, B, H, W, C = y.shape
    Wq = (W + R - 1) // R
    dense = wino_dense() if dense is None else dense
    eb = 2 if dense else 4
    out = torch.empty_like(y) if want_f32 else None
    out_hl = torch.empty(y.numel() * 4, device=y.device, dtype=torch.uint8) if want_hl else None
    v = torch.empty(G * B * H * Wq * (R + 2) * C * eb, device=y.device, dtype=torch.uint8)
    t0 = CONV_TIMER.begin("bnw") if CONV_TIMER is not None else None
    if dense:
        call("mrn_bn_apply_wino_grouped_d16_f32", _p(y), _p(residual), _p(residual_hl), _p(scale), _p(shift), _p(out), _p(out_hl), _p(v),
             G, B, H, W, C, int(bool(relu)), _p(prescale), int(REDUCED_BF16), _stream())
    else:
        call("mrn_bn_apply_wino_grouped_f32", _p(y), _p(residual), _p(residual_hl), _p(scale), _p(shift), _p(out), _p(out_hl), _p(v),
             G, B, H, W, C, R, int(bool(relu)), _p(prescale), _stream())
    if t0 is not None:       # algorithmic bytes: every input / output element once; the transformed operand is (R+2)/R elements per element
        n_io = 1 + int(residual is not None or residual_hl is not None) + int(want_f32) + int(want_hl)
        CONV_TIMER.end(t0, 0.0, "hbm/bn_apply_wino_grouped", 4.0 * y.numel() * n_io + float(eb) * G * B * H * Wq * (R + 2) * C)
    return out, out_hl, v


def wino_pool_supported(H, W, R, Cout):
    """can the 2x2 / 2 max-pool behind this Winograd convolution be taken in its epilogue (row-block kernel, even map)"""
    return PATCH_CONV and H % 2 == 0 and W % 2 == 0 and bool(call("mrn_conv2d_x3_wino_rows", H, R, Cout))


def conv2d_x3_wino(v_hl, G, shared_input, B, H, W, Cin, u_hl, u_scale, Cout, R, bias=None, act=ACT_NONE, want_stats=False, out=None,
                   x_scale=None, pool=False, gamma_ptrs=None, dense=None):
    """3x3 / stride 1 / pad 1 grouped conv on Winograd-domain operands -> (y [G,B,H,W,Cout] fp32, stats or None).
    pool (wino_pool_supported): y is the [G,B,H/2,W/2,Cout] map of per-window extremes (maxima where the BatchNorm weight that follows is
    >= 0 -- gamma_ptrs: int64 device tensor [G] of the weights' addresses, None: all maxima -- minima elsewhere), see conv3x3_patch_x3"""
    dev = v_hl.device
    y = out if out is not None else torch.empty(G, B, H // 2 if pool else H, W // 2 if pool else W, Cout, device=dev, dtype=torch.float32)
    stats = None
    if want_stats:
        stats = torch.empty(call("mrn_conv2d_x3_wino_stats_floats", G, B, H, W, Cout, R), device=dev, dtype=torch.float32)
    Wq = (W + R - 1) // R
    dense = wino_dense() if dense is None else dense
    eb = 2 if dense else 4
    gstride = 0 if shared_input else B * H * Wq * (R + 2) * Cin * eb
    timed = CONV_TIMER is not None
    t0 = CONV_TIMER.begin("wino") if timed else None
    if dense:           # (reduced-precision mode: plain fp16 operands, one product per term; row-block kernel only -- fails loudly otherwise)
        call("mrn_conv2d_x3_wino_d16", _p(v_hl), _p(u_hl), _p(bias), _p(y), _p(stats), _p(u_scale), _p(x_scale), G, gstride, B, H, W, Cin,
             Cout, act, int(bool(pool)), _p(gamma_ptrs), int(REDUCED_BF16), _stream())
    elif pool:
        call("mrn_conv2d_x3_wino_pool_hl32", _p(v_hl), _p(u_hl), _p(_zero_page(dev)), _p(bias), _p(y), _p(stats), _p(u_scale), _p(x_scale), G,
             gstride, B, H, W, Cin, Cout, R, act, _p(gamma_ptrs), _stream())
    else:
        call("mrn_conv2d_x3_wino_hl32", _p(v_hl), _p(u_hl), _p(_zero_page(dev)), _p(bias), _p(y), _p(stats), _p(u_scale), _p(x_scale), G,
             gstride, B, H, W, Cin, Cout, R, act, _stream())
    if WINO_CHECK and not pool and not dense and call("mrn_conv2d_x3_wino_rows", H, R, Cout):
        # debug (MRN_WINO_CHECK=1): the same call on the x3 kernel's Winograd form, compared element by element
        call("mrn_conv2d_x3_wino_select", 0)
        y2 = torch.empty_like(y)
        st2 = torch.empty(call("mrn_conv2d_x3_wino_stats_floats", G, B, H, W, Cout, R), device=dev, dtype=torch.float32) if want_stats else None
        call("mrn_conv2d_x3_wino_hl32", _p(v_hl), _p(u_hl), _p(_zero_page(dev)), _p(bias), _p(y2), _p(st2), _p(u_scale), _p(x_scale), G, gstride,
             B, H, W, Cin, Cout, R, act, _stream())
        call("mrn_conv2d_x3_wino_select", -1)
        if want_stats:
            a1, a2 = stats.view(G, -1, 2, Cout).sum(1), st2.view(G, -1, 2, Cout).sum(1)
            print(f"[wino check] stats blocks {stats.numel() // (2 * Cout * G)} vs {st2.numel() // (2 * Cout * G)}: sum diff "
                  f"{float((a1[:, 0] - a2[:, 0]).abs().max()):.3e} of {float(a2[:, 0].abs().max()):.3e}, sq diff "
                  f"{float((a1[:, 1] - a2[:, 1]).abs().max()):.3e} of {float(a2[:, 1].abs().max()):.3e}", flush=True, file=sys.__stderr__)
        d = float((y - y2).abs().max())
        sc = float(y2.abs().max())
        bad = int(((y - y2).abs() > 1e-4 * sc + 1e-30).sum())
        print(f"[wino check] G{G} B{B} {H}x{W} {Cin}->{Cout} stats={want_stats} x_scale={x_scale is not None}: max diff {d:.3e} of {sc:.3e}, {bad} bad", flush=True, file=sys.__stderr__)
        if bad:
            idx = ((y - y2).abs() > 1e-4 * sc + 1e-30).nonzero()
            print("   first bad indices [g,b,y,x,c]:", idx[:6].tolist(), "x range", int(idx[:, 3].min()), int(idx[:, 3].max()),
                  "c range", int(idx[:, 4].min()), int(idx[:, 4].max()), "y set", sorted(set(idx[:, 2].tolist())), flush=True, file=sys.__stderr__)
    if timed:
        # algorithmic flops = the convolution's (2 * 9 * Cin per output element); the kernel executes (R+2)/(3R) of them as MFMA products
        nbytes = (float(eb) * ((1 if shared_input else G) * B * H * Wq * (R + 2) * Cin + G * Cout * 3 * (R + 2) * Cin)
                  + 4.0 * G * B * (H // 2 if pool else H) * (W // 2 if pool else W) * Cout)
        kind = (("fp16" if dense else "fp16x3") + "/winorows%d" if call("mrn_conv2d_x3_wino_rows", H, R, Cout) else "fp16x3/wino%dg128x128") % R
        if TIMER_SHAPES:
            kind += "|G%d B%d %dx%d %d->%d k3x3 s11" % (G, B, H, W, Cin, Cout)
        CONV_TIMER.end(t0, 2.0 * G * B * H * W * Cout * 9 * Cin, kind, nbytes)
    return y, stats


# ---- the trained convolutions' weight operands, packed ahead of use -------------------------------------------------------
# A trained layer re-packs its weights every step: [O,I,kh,kw] -> OHWI, max|w| -> power-of-two scale, the HL32 / Winograd-domain stack, and
# the same again for the flipped data-gradient weights -- ~300 short launches per TRBA step, each in front of the convolution that
# needs it.  prepack_trained() (called by the learners right after the optimiser step) runs all of them on the side stream for the
# layers registered by ConvBlockFn, one event per layer; the consumers below find the operands by (source tensor, version) and wait
# on the event.  A miss (first step, a layer that changed since) just packs in place, as before.
WGRAD_SMALL_TILE = os.environ.get("MRN_WGRAD_SMALL_TILE", "1") == "1"      # A/B: 64 x 64 tiles for the weight gradients of Cout, Cin <= 64 layers
TRAIN_PREPACK = os.environ.get("MRN_TRAIN_PREPACK", "1") == "1"
# trained SVTR blocks (loop A): the pass that produces a Linear layer's input also writes its split operand / leaves its range scale
# (LayerNorm -> qkv / fc1, attention -> proj, GELU -> fc2; GELU' and the DropPath residual for the gradients); MRN_TRAIN_OPERAND_FUSION=0: A/B
TRAIN_OPERAND_FUSION = os.environ.get("MRN_TRAIN_OPERAND_FUSION", "1") == "1"
_PREPACKED = {}                     # key -> (value, event, source tensor kept alive)
TRAINED_CONVS = {}                  # id(conv) -> (weakref to the module, stride, padding), filled by functional.ConvBlockFn.forward


def _memo(key, src, build):
    got = _PREPACKED.get(key)
    if got is not None:
        if got[1] is not None:
            torch.cuda.current_stream().wait_event(got[1])
        return got[0]
    return build()


def _wkey(kind, t, *extra):
    return (kind, t.data_ptr(), t._version, tuple(t.shape)) + extra


_PACK_REGISTRY = {}                 # (kind, ids of the parameters) -> (weakrefs to them, build): what prepack_trained() re-runs every step


def train_pack(kind, params, build):
    """build(*params) for the CURRENT values of `params` (Parameters of a layer being trained): a repack of recurrent / decoder weights
    that the step needs in front of a kernel.  Found prepacked on the side stream -> wait for its event; otherwise built in place and
    registered, so that prepack_trained() issues it ahead of use from the next step on."""
    key = (kind,) + tuple((p.data_ptr(), p._version) for p in params)
    got = _PREPACKED.get(key)
    if got is not None:
        if got[1] is not None:
            torch.cuda.current_stream().wait_event(got[1])
        return got[0]
    val = build(*params)
    if TRAIN_PREPACK and all(isinstance(p, torch.nn.Parameter) for p in params):
        _PACK_REGISTRY[(kind,) + tuple(id(p) for p in params)] = (tuple(weakref.ref(p) for p in params), build)
    return val


def trained_weight_operand(w_ohwi, stride, padding, H=None):
    """-> ("wino", Winograd-domain stack, scale) or ("hl32", HL32 stack, scale) of one trained layer's [O,kh,kw,I] weights; H: rows of
    the map it is applied to (the plain-fp16 Winograd form of the reduced mode exists on the row-block kernel only: H % 4 == 0)"""
    Cout, kh, kw, Cin = w_ohwi.shape
    dense = wino_dense(TRAIN_PRODUCTS)
    wino = TRAIN_WINO and wino_eligible((kh, kw), stride, padding, Cin, Cout, TRAIN_PRODUCTS) and (not dense or (H is not None and H % 4 == 0))

    def build():
        w = w_ohwi.contiguous()
        return ("wino",) + tuple(pack_weights_wino([w], WINO_R, dense=dense)) if wino else ("hl32",) + tuple(pack_weights_hl32([w]))
    return _memo(_wkey("op", w_ohwi, wino, WINO_R, dense, REDUCED_BF16), w_ohwi, build)


def trained_dgrad_weight(w_ohwi):
    """pack_dgrad_weight of a trained layer (found prepacked, or built in place)"""
    return _memo(_wkey("dgrad", w_ohwi), w_ohwi, lambda: pack_dgrad_weight(w_ohwi))


def prepack_trained():
    """issue every registered trained layer's weight packing for the coming step on the side stream (see above)"""
    if not (TRAIN_PREPACK and WGRAD_SIDE_STREAM and (TRAINED_CONVS or _PACK_REGISTRY)):
        return
    from .modules._nn import packed_weight
    side = side_stream()
    side.wait_stream(torch.cuda.current_stream())          # the optimiser step wrote the weights on this stream
    with torch.cuda.stream(side):
        _PREPACKED.clear()
        for cid, (ref, stride, padding) in list(TRAINED_CONVS.items()):
            conv = ref()
            if conv is None or not conv.weight.requires_grad:
                del TRAINED_CONVS[cid]
                continue
            w = packed_weight(conv).ohwi
            kh, kw = w.shape[1], w.shape[2]
            new = {}
            if w.shape[-1] % 32 == 0:                        # (what conv2d_nhwc sends to conv2d_x3_scaled)
                Cout, _, _, Cin = w.shape
                dense = wino_dense(TRAIN_PRODUCTS)          # (H unknown here: a layer on a map the d16 form cannot take misses and packs in place)
                wino = TRAIN_WINO and wino_eligible((kh, kw), stride, padding, Cin, Cout, TRAIN_PRODUCTS)
                wc = w.contiguous()
                new[_wkey("op", w, wino, WINO_R, dense, REDUCED_BF16)] = (("wino",) + tuple(pack_weights_wino([wc], WINO_R, dense=dense)) if wino
                                                            else ("hl32",) + tuple(pack_weights_hl32([wc])), w)
            wt = pack_dgrad_weight(w)
            new[_wkey("dgrad", w)] = (wt, w)
            if wt.ohwi.shape[-1] % 32 == 0:
                dpad = (kh - 1 - padding[0], kw - 1 - padding[1])
                dense = wino_dense(TRAIN_PRODUCTS)
                wino = TRAIN_WINO and wino_eligible((kh, kw), (1, 1), dpad, wt.ohwi.shape[-1], wt.ohwi.shape[0], TRAIN_PRODUCTS)
                new[_wkey("op", wt.ohwi, wino, WINO_R, dense, REDUCED_BF16)] = (("wino",) + tuple(pack_weights_wino([wt.ohwi], WINO_R, dense=dense)) if wino
                                                                  else ("hl32",) + tuple(pack_weights_hl32([wt.ohwi])), wt.ohwi)
            ev = torch.cuda.Event()
            ev.record(side)
            for k, (val, src) in new.items():
                _PREPACKED[k] = (val, ev, src)
        batched = []                                            # Linear weights: all of them in one call (multi_pack_linear)
        for rk, (refs, build) in list(_PACK_REGISTRY.items()):
            ps = [r() for r in refs]
            if any(p_ is None or not p_.requires_grad for p_ in ps):
                del _PACK_REGISTRY[rk]
                continue
            if MULTI_PACK and rk[0] in _MULTI_PACK_KINDS and multi_pack_eligible(ps[0], rk[0]):
                batched.append((rk[0], ps[0]))
                continue
            with torch.no_grad():
                val = build(*ps)
            ev = torch.cuda.Event()
            ev.record(side)
            _PREPACKED[(rk[0],) + tuple((p_.data_ptr(), p_._version) for p_ in ps)] = (val, ev, ps)
        if batched:
            vals = multi_pack_linear(batched)
            ev = torch.cuda.Event()
            ev.record(side)
            for (kind, p_), val in zip(batched, vals):
                _PREPACKED[(kind, (p_.data_ptr(), p_._version))] = (val, ev, [p_])


MULTI_PACK = os.environ.get("MRN_MULTI_PACK", "1") == "1"      # A/B switch: 0 = one max / pack launch sequence per Linear weight
_MULTI_PACK_KINDS = {"lin_fwd_x3": 0, "lin_bwd_x3": 1}        # -> transposed flag (functional._pack_linear / _pack_linear_t)
_MULTI_PACK_STATE = {}
MULTI_PACK_STATS = {"calls": 0, "rebuilds": 0, "weights": 0}


def multi_pack_eligible(p, kind):
    if p.dim() != 2 or p.dtype != torch.float32 or not p.is_contiguous() or p.data_ptr() % 16:
        return False
    N, K = p.shape
    return N % 4 == 0 and K % 4 == 0 and (N if _MULTI_PACK_KINDS[kind] else K) % 32 == 0


def multi_pack_linear(entries):
    """entries [(kind, weight [N,K])] -> [(HL32 operand bytes, scale [1,2])] exactly as pack_weights_hl32 builds them one by one
    (kind lin_fwd_x3: of W; lin_bwd_x3: of W^T), in ONE library call on the current stream.
    ALIASING CONTRACT: the returned operands are views into persistent per-device buffers, not fresh tensors.  Two buffers alternate
    call by call, so a view stays valid through the NEXT call (one optimiser step: prepack_trained() rewrites the other buffer) and is
    overwritten in stream order by the call after that; nothing may hold a pack across two optimiser steps (the _PREPACKED / _memo
    keys -- data pointer + version -- only guard the lookup, not a reference somebody kept)."""
    dev = entries[0][1].device
    sig = tuple((kind, p.data_ptr(), tuple(p.shape)) for kind, p in entries)
    st = _MULTI_PACK_STATE.get(dev)
    if st is None or st["sig"] != sig:
        n = len(entries)
        offs, tiles, total = [], 0, 0
        for kind, p in entries:
            N, K = p.shape
            tr = _MULTI_PACK_KINDS[kind]
            O, I = (K, N) if tr else (N, K)
            offs.append((total, tiles, I // 32, tr))
            tiles += ((O + 31) // 32) * (I // 32)
            total += (N * K * 4 + 255) // 256 * 256
        halves = []
        for _ in range(2):
            out = torch.empty(total, device=dev, dtype=torch.uint8)
            scale = torch.empty(n, 2, device=dev, dtype=torch.float32)
            rows = [[p.data_ptr(), out.data_ptr() + off, scale.data_ptr() + 8 * i, p.shape[0], p.shape[1], tr, t0, ib]
                    for i, ((_, p), (off, t0, ib, tr)) in enumerate(zip(entries, offs))]
            views = [(out[offs[i][0]:offs[i][0] + p.numel() * 4], scale[i:i + 1]) for i, (_, p) in enumerate(entries)]
            halves.append({"desc": torch.tensor(rows, dtype=torch.int64).to(dev), "out": out, "scale": scale, "views": views})
        st = {"sig": sig, "halves": halves, "flip": 0, "amax": torch.zeros(n, device=dev, dtype=torch.int32), "tiles": tiles}
        _MULTI_PACK_STATE[dev] = st
        MULTI_PACK_STATS["rebuilds"] += 1
    half = st["halves"][st["flip"]]
    st["flip"] ^= 1
    call("mrn_multi_pack_linear_hl32", half["desc"].data_ptr(), len(entries), st["tiles"], st["amax"].data_ptr(), FP16_WEIGHT_PEAK, _stream())
    MULTI_PACK_STATS["calls"] += 1
    MULTI_PACK_STATS["weights"] += len(entries)
    return half["views"]


def conv2d_x3_scaled(x, w_ohwi, bias, stride, padding, act=ACT_NONE, want_stats=False, sx=None):
    """one convolution on the grouped x3 kernel with per-call operand scaling: x fp32 [B,H,W,Cin] (Cin % 32 == 0),
    w_ohwi fp32 [O,kh,kw,I] -> (y [B,Ho,Wo,O], stats or None)"""
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = w_ohwi.shape
    x = x.contiguous()
    if sx is None:
        sx = pow2_scale(x, TRAIN_OPERAND_PEAK)          # (callers that use x in several GEMMs compute it once: ConvBlockFn)
    kind, w_hl, sw = trained_weight_operand(w_ohwi, stride, padding, H)
    if kind == "wino":
        # Winograd F(R,3) along W, as for the frozen experts (conv_x3.hip WINO): the operand pass applies B^T to sx * x in place of the
        # plain split, the weights are re-transformed with the step's values.  Serves the forward AND the data-gradient convolutions of
        # loop A (conv2d_dgrad arrives here with the flipped weights and the gradient's own range scale).
        R = WINO_R
        dense = wino_dense(TRAIN_PRODUCTS)
        _, _, v = bn_apply_wino_grouped(x.view(1, B, H, W, Cin), None, None, R, relu=False, prescale=sx, dense=dense)
        y, stats = conv2d_x3_wino(v, 1, False, B, H, W, Cin, w_hl, sw, Cout, R, bias=bias, act=act, want_stats=want_stats, x_scale=sx, dense=dense)
        return y[0], stats
    y, stats = conv2d_x3(split_hl32(x, sx), 1, False, B, H, W, Cin, w_hl, sw, Cout, (kh, kw), stride, padding, bias=bias, act=act,
                         want_stats=want_stats, x_scale=sx, products=TRAIN_PRODUCTS)
    return y[0], stats


def conv2d_x3_frozen(x, packed, bias, stride, padding, act=ACT_NONE, want_stats=False):
    """one convolution of a frozen expert outside a lock-step group on the grouped x3 kernel (G = 1): x fp32 [B,H,W,Cin] (Cin % 32 == 0)
    split unscaled (post-BatchNorm activations), weights from the PackedConvWeight's cached operand -> (y [B,Ho,Wo,O], stats or None)"""
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = packed.shape
    x = x.contiguous()
    wino = WINO_R in (2, 4) and X3_PRODUCTS == 3 and wino_eligible((kh, kw), stride, padding, Cin, Cout)
    w_hl, sw = packed.x3_operand(wino)
    if wino:
        _, _, v = bn_apply_wino_grouped(x.view(1, B, H, W, Cin), None, None, WINO_R, relu=False)
        y, stats = conv2d_x3_wino(v, 1, False, B, H, W, Cin, w_hl, sw, Cout, WINO_R, bias=bias, act=act, want_stats=want_stats)
        return y[0], stats
    y, stats = conv2d_x3(split_hl32(x), 1, False, B, H, W, Cin, w_hl, sw, Cout, (kh, kw), stride, padding, bias=bias, act=act,
                         want_stats=want_stats, products=X3_PRODUCTS)
    return y[0], stats


def conv3x3_c4_grouped(x, weights, bias=None, act=ACT_NONE, want_stats=False, out=None, pool=False, gamma_ptrs=None):
    """First conv of G frozen experts (3x3, stride 1, padding 1, Cin = 4, Cout 32 / 64) in one launch.  x: [B,H,W,4] (shared by
    all experts) or [G,B,H,W,4]; weights: [G,Cout,3,3,4] stack; bias [G,Cout] or None -> (y [G,B,H,W,Cout], stats or None).
    pool: y is the [G,B,H/2,W/2,Cout] map of per-window extremes (maxima where the BatchNorm weight that follows is >= 0 -- gamma_ptrs:
    int64 device tensor [G] of the weights' addresses, None: all maxima -- minima elsewhere), see conv3x3_patch_x3"""
    _chk(x, weights, bias)
    G, Cout = weights.shape[0], weights.shape[1]
    shared = x.dim() == 4
    B, H, W, C = x.shape[-4:]
    assert C == 4 and x.is_contiguous() and weights.is_contiguous() and tuple(weights.shape[2:]) == (3, 3, 4)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = out if out is not None else torch.empty(G, B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    stats = None
    if want_stats:
        nblk = call("mrn_conv3x3_c4_stats_blocks", B, H, W)
        stats = torch.empty(G, nblk, 2, Cout, device=x.device, dtype=torch.float32)
    t0 = CONV_TIMER.begin() if CONV_TIMER is not None else None
    call("mrn_conv3x3_c4_grouped_f32", _p(x), _p(weights), _p(bias), _p(y), _p(stats), G, 0 if shared else B * H * W * 4, B, H, W,
         Cout, act, int(bool(pool)), _p(gamma_ptrs), _stream())
    if t0 is not None:
        # (the full-map form is bound by its output write; the pooled form writes a quarter and is bound by the exact-fp32 MFMA)
        CONV_TIMER.end(t0, 2.0 * G * B * H * W * Cout * 36, ("f32mfma/" if pool else "hbm/") + "conv_first_kernel<%d%s>" % (Cout // 32, ", true" if pool else ""),
                       4.0 * ((1 if shared else G) * B * H * W * 4 + G * B * Ho * Wo * Cout))
    return y, stats


PATCH_CONV = os.environ.get("MRN_PATCH_CONV", "1") == "1"       # A/B switch: the narrow early 3x3 layers on the patch-resident kernel


def patch_conv_supported(ksize, stride, padding, Cin, Cout):
    return (PATCH_CONV and tuple(ksize) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and bool(call("mrn_conv3x3_patch_supported", Cin, Cout)))


def conv3x3_patch_x3(x_hl, G, shared_input, B, H, W, Cin, w_hl, w_scale, Cout, bias=None, act=ACT_NONE, want_stats=False, pool=False,
                     gamma_ptrs=None, out=None):
    """3x3 / stride 1 / pad 1 convolution of G lock-step experts on the patch-resident weight-stationary kernel (csrc/conv_patch.hip;
    (Cin, Cout) = (32, 64) / (64, 128)) -> (y, stats or None).  pool: y is the [G,B,H/2,W/2,Cout] map of per-window extremes -- maxima
    where the BatchNorm weight that follows is >= 0 (gamma_ptrs: int64 device tensor [G] of the weights' addresses; None: all maxima),
    minima elsewhere -- on which BatchNorm-apply + ReLU equals apply + ReLU + 2x2 max-pool of the full map."""
    dev = x_hl.device
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = out if out is not None else torch.empty(G, B, Ho, Wo, Cout, device=dev, dtype=torch.float32)
    stats = None
    if want_stats:
        stats = torch.empty(G, call("mrn_conv3x3_patch_stats_blocks", G, B, H, W, Cin), 2, Cout, device=dev, dtype=torch.float32)
    timed = CONV_TIMER is not None
    t0 = CONV_TIMER.begin() if timed else None
    one = X3_PRODUCTS == 1          # (reduced-precision mode: hi x hi only, same operands)
    call("mrn_conv3x3_patch_x1_hl32" if one else "mrn_conv3x3_patch_x3_hl32", _p(x_hl), _p(w_hl), _p(w_scale), _p(bias), _p(gamma_ptrs), _p(y),
         _p(stats), G, 0 if shared_input else B * H * W * Cin * 4, B, H, W, Cin, Cout, act, int(bool(pool)), _stream())
    if timed:
        nbytes = 4.0 * ((1 if shared_input else G) * B * H * W * Cin + G * Cout * 9 * Cin + G * B * Ho * Wo * Cout)
        kind = ("fp16" if one else "fp16x3") + "/patch%d" % Cin + ("pool" if pool else "")
        if TIMER_SHAPES:
            kind += "|G%d B%d %dx%d %d->%d k3x3 s11" % (G, B, H, W, Cin, Cout)
        CONV_TIMER.end(t0, 2.0 * G * B * H * W * Cout * 9 * Cin, kind, nbytes)
    return y, stats


_FINALIZE_TICKETS = {}


def _finalize_tickets(device, slots):
    """zeroed 32-bit tickets per (device, stream) for the chunked BatchNorm finalize (the kernel's last workgroup resets its ticket)"""
    key = _dev_stream()
    t = _FINALIZE_TICKETS.get(key)
    if t is None or t.numel() < slots:
        t = torch.zeros(max(slots, 1024), device=device, dtype=torch.int32)
        _FINALIZE_TICKETS[key] = t
    return t


def bn_finalize_grouped(stats, G, C, count, ptr_table, momentum, eps):
    """stats [G][nblk][2][C]; ptr_table: int64 device tensor [4,G] of {gamma, beta, running_mean, running_var} addresses"""
    scale = torch.empty(G, C, device=stats.device, dtype=torch.float32)
    shift = torch.empty(G, C, device=stats.device, dtype=torch.float32)
    nblk = stats.numel() // (2 * C * G)
    Z = call("mrn_bn_finalize_grouped_chunks", nblk)
    ws = tk = None
    if Z > 1:       # (the first layers: thousands of partial rows per expert -- reduced in Z chunks, combined by the last workgroup to arrive)
        slots = G * ((C + 31) // 32)
        ws = torch.empty(slots * Z * 64, device=stats.device, dtype=torch.float64)
        tk = _finalize_tickets(stats.device, slots)
    call("mrn_bn_finalize_grouped_f32", _p(stats), G, nblk, C, count, _p(ptr_table), float(momentum), float(eps), _p(scale),
         _p(shift), _p(ws), _p(tk), _stream())
    return scale, shift


def bn_eval_affine_grouped(ptr_table, G, C, eps):
    """eval-mode BatchNorm of G modules -> (scale [G,C], shift [G,C]) from their current running statistics (one launch)"""
    scale = torch.empty(G, C, device=ptr_table.device, dtype=torch.float32)
    shift = torch.empty(G, C, device=ptr_table.device, dtype=torch.float32)
    call("mrn_bn_eval_affine_grouped_f32", _p(ptr_table), G, C, float(eps), _p(scale), _p(shift), _stream())
    return scale, shift


def bn_apply_grouped(y, scale, shift, relu=True, residual=None, want_f32=True, want_hl=False, residual_hl=None):
    """y [G,...,C] fp32 -> (fp32 result (in place) or None, HL32 bytes or None)"""
    G, C = y.shape[0], y.shape[-1]
    rows = y.numel() // (G * C)
    out_hl = torch.empty(y.numel() * 4, device=y.device, dtype=torch.uint8) if want_hl else None
    t0 = CONV_TIMER.begin() if CONV_TIMER is not None else None
    call("mrn_bn_apply_grouped_f32", _p(y), _p(residual), _p(residual_hl), _p(scale), _p(shift), _p(y) if want_f32 else None,
         _p(out_hl), G, rows,
         C, 2 if relu == 2 else int(bool(relu)), _stream())          # (relu = 2: GELU)
    if t0 is not None:       # algorithmic bytes: every input / output element once (fp32 and HL32 are both 4 B / element)
        n_io = 1 + int(residual is not None or residual_hl is not None) + int(want_f32) + int(want_hl)
        CONV_TIMER.end(t0, 0.0, "hbm/bn_apply_grouped", 4.0 * y.numel() * n_io)
    return (y if want_f32 else None), out_hl


def maxpool_grouped(x, kernel, stride, padding, scale=None, shift=None, relu=False, want_f32=True, want_hl=False):
    """x [G,B,H,W,C] -> (fp32 [G,B,Ho,Wo,C] or None, HL32 bytes or None)"""
    G, B, H, W, C = x.shape
    Ho, Wo = conv_out_hw(H, W, kernel, stride, padding)
    out = torch.empty(G, B, Ho, Wo, C, device=x.device, dtype=torch.float32) if want_f32 else None
    out_hl = torch.empty(G * B * Ho * Wo * C * 4, device=x.device, dtype=torch.uint8) if want_hl else None
    call("mrn_maxpool_grouped_f32", _p(x), _p(scale), _p(shift), int(bool(relu)), _p(out), _p(out_hl), G, B, H, W, C, kernel[0],
         kernel[1], stride[0], stride[1], padding[0], padding[1], _stream())
    return out, out_hl, (Ho, Wo)


def maxpool_wino_grouped(x, kernel, stride, padding, R, scale=None, shift=None, relu=False, want_f32=False, want_hl=False, dense=None):
    """x [G,B,H,W,C] -> (fp32 [G,B,Ho,Wo,C] or None, HL32 bytes or None, Winograd-domain operand bytes of the pooled map, (Ho, Wo));
    dense (None: wino_dense()): the operand as plain fp16"""
    G, B, H, W, C = x.shape
    Ho, Wo = conv_out_hw(H, W, kernel, stride, padding)
    Wq = (Wo + R - 1) // R
    dense = wino_dense() if dense is None else dense
    out = torch.empty(G, B, Ho, Wo, C, device=x.device, dtype=torch.float32) if want_f32 else None
    out_hl = torch.empty(G * B * Ho * Wo * C * 4, device=x.device, dtype=torch.uint8) if want_hl else None
    v = torch.empty(G * B * Ho * Wq * (R + 2) * C * (2 if dense else 4), device=x.device, dtype=torch.uint8)
    if dense:
        call("mrn_maxpool_wino_grouped_d16_f32", _p(x), _p(scale), _p(shift), int(bool(relu)), _p(out), _p(out_hl), _p(v), G, B, H, W, C,
             kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], int(REDUCED_BF16), _stream())
    else:
        call("mrn_maxpool_wino_grouped_f32", _p(x), _p(scale), _p(shift), int(bool(relu)), _p(out), _p(out_hl), _p(v), G, B, H, W, C,
             kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], R, _stream())
    return out, out_hl, v, (Ho, Wo)


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


# Range scales of tensors whose producer pass folded max|.| into its own sweep: id(tensor) -> (weak reference, device {s, 1/s}).  An
# entry is put by the producer (scale_shift_act(range_target=...)) and popped by the consumer (cached_scale), which takes it only if
# the weak reference still points at the very tensor object it was asked about -- a recycled id or address can never match.
_SCALE_CACHE = {}
FUSED_AMAX = os.environ.get("MRN_FUSED_AMAX", "1") == "1"


def cached_scale(x):
    e = _SCALE_CACHE.pop(id(x), None)
    return e[1] if (e is not None and e[0]() is x) else None


def clear_scale_cache():
    _SCALE_CACHE.clear()


def scale_shift_act(x, scale, shift, relu=True, residual=None, out=None, range_target=None, pos_mask=None):
    """range_target: also fold max|out| into the pass and keep the power-of-two range scale {s, 1/s} (s * max|out| <= range_target)
    for the consumer (cached_scale): saves the extra read of the tensor mrn_pow2_scale_f32 would make.
    pos_mask: uint8 [numel / 4] receiving the ReLU mask of the backward pass (bn_bwd zmask)"""
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = x
    ws = _amax_ws() if (range_target is not None and FUSED_AMAX) else None
    call("mrn_scale_shift_act_f32", _p(x), _p(residual), _p(out), _p(scale), _p(shift), rows, C, int(relu), ws, _p(pos_mask), _stream())
    if ws is not None:
        sc = torch.empty(2, device=x.device, dtype=torch.float32)
        call("mrn_pow2_finalize_f32", float(range_target), _p(sc), ws, _stream())
        if len(_SCALE_CACHE) > 256:
            clear_scale_cache()                      # (entries nobody asked for: eval-mode consumers, non-conv consumers)
        _SCALE_CACHE[id(out)] = (weakref.ref(out), sc)
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
def tps_grid_sample(img_nhwc, cprime, inv_delta_c, p_hat, out_hw, want_grid=False, out=None):
    _chk(img_nhwc, cprime, inv_delta_c, p_hat)
    B, H, W, C = img_nhwc.shape
    Hr, Wr = out_hw
    F = cprime.shape[1]
    if out is None:
        out = torch.empty(B, Hr, Wr, C, device=img_nhwc.device, dtype=torch.float32)
    assert out.is_contiguous() and out.numel() == B * Hr * Wr * C
    grid = torch.empty(B, Hr * Wr, 2, device=img_nhwc.device, dtype=torch.float32) if want_grid else None
    call("mrn_tps_grid_sample_f32", _p(img_nhwc), _p(cprime.contiguous()), _p(inv_delta_c), _p(p_hat), _p(out),
         _p(grid), B, H, W, C, Hr, Wr, F, _stream())
    return (out, grid) if want_grid else out


# ---------------------------------------------------------------------------------------------------------
# recurrent
# ---------------------------------------------------------------------------------------------------------
def pack_fragment_major(w, hidden=256):
    """[G*hidden, K] weight (G gate groups, K % 16 == 0) -> the fragment-major stream order of the recurrent kernels:
    packed[w][g][q][lane = gg*16 + n][r] = W[g*hidden + 16w + n][16q + 4gg + r]  (pure data movement)."""
    G = w.shape[0] // hidden
    K = w.shape[1]
    assert w.shape[0] == G * hidden and hidden == 256 and K % 16 == 0
    v = w.reshape(G, 16, 16, K // 16, 4, 4)              # g, w, n, q, gg, r
    return v.permute(1, 0, 3, 4, 2, 5).contiguous()      # w, g, q, gg, n, r


def lstm_layer(xproj, w_hh, b_hh, hidden, ndir, save=False):
    """xproj [B,T,ndir*4H] (= x W_ih^T + b_ih), w_hh: ndir stacked pack_fragment_major([4H,H]), b_hh [ndir*4H] or None"""
    _chk(xproj, w_hh, b_hh)
    B, T, _ = xproj.shape
    out = torch.empty(B, T, ndir * hidden, device=xproj.device, dtype=torch.float32)
    gates = cseq = None
    if save:
        gates = torch.empty(B, T, ndir, 4 * hidden, device=xproj.device, dtype=torch.float32)
        cseq = torch.empty(B, T, ndir, hidden, device=xproj.device, dtype=torch.float32)
    call("mrn_lstm_layer_fwd_f32", _p(xproj), _p(w_hh), _p(b_hh), _p(out), _p(gates), _p(cseq), B, T, hidden, ndir, _stream())
    return (out, gates, cseq) if save else out


TRAIN_LSTM_X3 = os.environ.get("MRN_TRAIN_LSTM_X3", "1") == "1"   # trained LSTM layers: forward recurrent product as split-fp16 x3 (half the step time)


def lstm_layer_x3_save(xproj, w_hh_h, w_inv, b_hh, hidden, ndir):
    """training forward on the f16 MFMA: xproj [B,T,ndir*4H], w_hh_h [ndir,...] fp16 streams + w_inv [ndir] (pack_fragment_major_h per
    direction), b_hh [ndir*4H] -> (out [B,T,ndir*H], gates [B,T,ndir,4H], cseq [B,T,ndir,H]) as ops.lstm_layer(save=True)"""
    _chk(xproj, b_hh)
    B, T, _ = xproj.shape
    assert xproj.is_contiguous() and w_hh_h.is_contiguous() and w_inv.is_contiguous()
    out = torch.empty(B, T, ndir * hidden, device=xproj.device, dtype=torch.float32)
    gates = torch.empty(B, T, ndir, 4 * hidden, device=xproj.device, dtype=torch.float32)
    cseq = torch.empty(B, T, ndir, hidden, device=xproj.device, dtype=torch.float32)
    call("mrn_lstm_layer_fwd_x3_save", _p(xproj), _p(w_hh_h), _p(w_inv), _p(b_hh), _p(out), _p(gates), _p(cseq), B, T, hidden, ndir, _stream())
    return out, gates, cseq


def _ptr_array(ptrs):
    import ctypes
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def pack_fragment_major_h(w, hidden=256):
    """[G*hidden, K] fp32 weight (K % 32 == 0) -> (fp16 fragment-major split stream for the f16 MFMA recurrent kernels,
    inverse prescale [1]): packed[w][g][q][lane = kg*16 + n][plane][e] = split(s * W[g*hidden + 16w + n][32q + 8kg + e])"""
    G = w.shape[0] // hidden
    K = w.shape[1]
    assert w.shape[0] == G * hidden and hidden == 256 and K % 32 == 0
    sc = pow2_scale(w.contiguous())
    v = (w * sc[0]).reshape(G, 16, 16, K // 32, 4, 8).permute(1, 0, 3, 4, 2, 5)     # w, g, q, kg, n, e
    hi = v.half()
    lo = (v - hi.float()).half()
    return torch.stack([hi, lo], dim=5).contiguous(), sc[1:2].clone()                  # [..., n, plane, e]


def lstm_layer_x3_grouped(xproj, w_hh_h, w_inv, b_hh, hidden, ndir):
    """xproj [G,B,T,ndir*4H]; w_hh_h [G,ndir,...] fp16 streams (pack_fragment_major_h), w_inv [G,ndir], b_hh [G,ndir*4H]"""
    _chk(xproj, b_hh)
    G, B, T, _ = xproj.shape
    assert xproj.is_contiguous() and w_hh_h.is_contiguous() and w_inv.is_contiguous() and b_hh.is_contiguous()
    out = torch.empty(G, B, T, ndir * hidden, device=xproj.device, dtype=torch.float32)
    call("mrn_lstm_layer_fwd_x3_grouped", _ptr_array([xproj[g].data_ptr() for g in range(G)]),
         _ptr_array([w_hh_h[g].data_ptr() for g in range(G)]), _ptr_array([w_inv[g].data_ptr() for g in range(G)]),
         _ptr_array([b_hh[g].data_ptr() for g in range(G)]), _ptr_array([out[g].data_ptr() for g in range(G)]), G, B, T,
         hidden, ndir, _stream())
    return out


# Measured and removed in round 4 (DESIGN.md section 7 keeps the numbers): a weight-stationary LSTM (W_hh slices in the LDS of 16 workgroups per
# (expert, direction), h exchanged every step: 22.3 vs 17.1 us / step at G = 3) and the layers as one small kernel per time step replayed from
# a HIP graph (G = 1 13.0 -> 16.6, G = 3 17.2 -> 18.3, G = 6 22.3 -> 19.9 us / step).  The streaming kernels above are the product path.


def lstm_layer_grouped(xproj, w_hh, b_hh, hidden, ndir):
    """xproj [G,B,T,ndir*4H], w_hh [G,ndir,...] (fragment-major stacks), b_hh [G,ndir*4H] -> [G,B,T,ndir*H], one launch"""
    _chk(xproj, w_hh, b_hh)
    G, B, T, _ = xproj.shape
    assert xproj.is_contiguous() and w_hh.is_contiguous() and b_hh.is_contiguous()
    out = torch.empty(G, B, T, ndir * hidden, device=xproj.device, dtype=torch.float32)
    call("mrn_lstm_layer_fwd_grouped_f32", _ptr_array([xproj[g].data_ptr() for g in range(G)]),
         _ptr_array([w_hh[g].data_ptr() for g in range(G)]), _ptr_array([b_hh[g].data_ptr() for g in range(G)]),
         _ptr_array([out[g].data_ptr() for g in range(G)]), G, B, T, hidden, ndir, _stream())
    return out


def attn_decoder_grouped(Hb, Hproj, eproj, w_h2h, b_h2h, w_score, w_ih, w_hh, b_hh, hidden, w_inv=None):
    """Hb [G,B,T,D], Hproj [G,B,T,H], eproj [G,B,S,4H]; the weight arguments are lists of G tensors -> hid [G,B,S,H]"""
    _chk(Hb, Hproj, eproj)
    G, B, T, D = Hb.shape
    S = eproj.shape[2]
    assert Hb.is_contiguous() and Hproj.is_contiguous() and eproj.is_contiguous()
    hid = torch.empty(G, B, S, hidden, device=Hb.device, dtype=torch.float32)

    def arr(ts):
        return _ptr_array([t.data_ptr() for t in ts])
    if w_inv is not None:          # x3 form: the weight lists hold pack_fragment_major_h streams, w_inv a list of device float[3]
        call("mrn_attn_decoder_fwd_x3_grouped", arr(Hb), arr(Hproj), arr(eproj), eproj.stride(1), eproj.stride(2), arr(w_h2h),
             arr(b_h2h), arr(w_score), arr(w_ih), arr(w_hh), arr(w_inv), arr(b_hh), arr(hid), hid.stride(1), hid.stride(2), G, B, T, D, S,
             hidden, _stream())
        return hid
    call("mrn_attn_decoder_fwd_grouped_f32", arr(Hb), arr(Hproj), arr(eproj), eproj.stride(1), eproj.stride(2), arr(w_h2h),
         arr(b_h2h), arr(w_score), arr(w_ih), arr(w_hh), arr(b_hh), arr(hid), hid.stride(1), hid.stride(2), G, B, T, D, S,
         hidden, _stream())
    return hid


DECODER_X3 = os.environ.get("MRN_DECODER_X3", "1") == "1"     # attention decoder: the three recurrent products as split-fp16 x3


def pack_decoder_x3(w_h2h, w_ih, w_hh, D):
    """(h2h.weight [H,H], rnn.weight_ih [4H, D+E], rnn.weight_hh [4H,H]) -> (three pack_fragment_major_h streams, w_inv float[3])"""
    a, b, c = pack_fragment_major_h(w_h2h.detach()), pack_fragment_major_h(w_ih.detach()[:, :D].contiguous()), pack_fragment_major_h(w_hh.detach())
    return a[0], b[0], c[0], torch.cat([a[1], b[1], c[1]]).contiguous()


def lstm_layer_bwd(dout, gates, cseq, w_hhT, hidden, ndir):
    """-> dgates [B,T,ndir,4H] (gradient of the gate pre-activations); w_hhT: ndir stacked pack_fragment_major(W_hh.t())"""
    B, T, _ = dout.shape
    dout = dout.contiguous()
    dgates = torch.empty(B, T, ndir, 4 * hidden, device=dout.device, dtype=torch.float32)
    call("mrn_lstm_layer_bwd_f32", _p(dout), _p(gates), _p(cseq), _p(w_hhT), _p(dgates), B, T, hidden, ndir, _stream())
    return dgates


def lstm_layer_bwd_x3(dout, gates, cseq, w_hhT_h, w_inv, hidden, ndir):
    """lstm_layer_bwd with the recurrent product as split-fp16 x3: w_hhT_h [ndir,...] fp16 streams + w_inv [ndir] =
    pack_fragment_major_h(W_hh.t()) per direction; the gate gradients are range-scaled by a power of two from max|dout|"""
    B, T, _ = dout.shape
    dout = dout.contiguous()
    assert w_hhT_h.is_contiguous() and w_inv.is_contiguous()
    gs = pow2_scale(dout, 16.0)
    dgates = torch.empty(B, T, ndir, 4 * hidden, device=dout.device, dtype=torch.float32)
    call("mrn_lstm_layer_bwd_x3", _p(dout), _p(gates), _p(cseq), _p(w_hhT_h), _p(w_inv), _p(gs), _p(dgates), B, T, hidden, ndir, _stream())
    return dgates


def embed_gather(idx, table, num_class, out=None):
    """idx [B,S] int64 (any row stride), table [C,E] -> [B,S,E] with cut_unknown semantics"""
    assert idx.dtype == torch.int64 and idx.stride(1) == 1
    B, S = idx.shape
    E = table.shape[1]
    if out is None:
        out = torch.empty(B, S, E, device=table.device, dtype=torch.float32)
    assert out.is_contiguous() and out.numel() == B * S * E
    call("mrn_embed_gather_f32", _p(idx), idx.stride(0), _p(table), _p(out), B, S, E, num_class, _stream())
    return out


def attn_decoder(Hb, Hproj, eproj, w_h2h, b_h2h, w_score, w_ih, w_hh, b_hh, hidden, hid=None, h_state=None,
                 c_state=None, want_alpha=False, w_inv=None):
    """w_h2h, w_ih (= W_ih[:, :D]) and w_hh must be pack_fragment_major()'d -- or, with w_inv, the fp16 streams of pack_decoder_x3"""
    _chk(Hb, Hproj, eproj, b_h2h, w_score, b_hh)
    if w_inv is None:
        _chk(w_h2h, w_ih, w_hh)
    B, T, D = Hb.shape
    S = eproj.shape[1]
    if hid is None:
        hid = torch.empty(B, S, hidden, device=Hb.device, dtype=torch.float32)
    alpha = torch.empty(B, S, T, device=Hb.device, dtype=torch.float32) if want_alpha else None
    assert eproj.stride(2) == 1 and hid.stride(2) == 1 and w_ih.is_contiguous() and w_hh.is_contiguous()
    if w_inv is not None:          # x3 form (pack_decoder_x3)
        call("mrn_attn_decoder_fwd_x3", _p(Hb), _p(Hproj), _p(eproj), eproj.stride(0), eproj.stride(1), _p(w_h2h),
             _p(b_h2h), _p(w_score), _p(w_ih), _p(w_hh), _p(w_inv), _p(b_hh), _p(hid), hid.stride(0), hid.stride(1),
             _p(h_state), _p(c_state), _p(alpha), None, None, None, None, B, T, D, S, hidden, _stream())
        return (hid, alpha) if want_alpha else hid
    call("mrn_attn_decoder_fwd_f32", _p(Hb), _p(Hproj), _p(eproj), eproj.stride(0), eproj.stride(1), _p(w_h2h),
         _p(b_h2h), _p(w_score), _p(w_ih), _p(w_hh), _p(b_hh), _p(hid), hid.stride(0), hid.stride(1),
         _p(h_state), _p(c_state), _p(alpha), None, None, None, None, B, T, D, S, hidden, _stream())
    return (hid, alpha) if want_alpha else hid


# ---------------------------------------------------------------------------------------------------------
# row-wise operators (DM-Router)
# ---------------------------------------------------------------------------------------------------------
def argmax_lastdim(x):
    x2 = rows2d(x)
    out = torch.empty(x2.shape[0], device=x.device, dtype=torch.int64)
    call("mrn_argmax_f32", _p(x2), x2.stride(0), _p(out), x2.shape[0], x2.shape[1], _stream())
    return out.view(x.shape[:-1])


def argmax_prob_lastdim(x):
    """(first argmax, softmax probability of it) along the last dim -- greedy decoding + confidence in one pass"""
    x2 = rows2d(x)
    idx = torch.empty(x2.shape[0], device=x.device, dtype=torch.int64)
    prob = torch.empty(x2.shape[0], device=x.device, dtype=torch.float32)
    call("mrn_argmax_prob_f32", _p(x2), x2.stride(0), _p(idx), _p(prob), x2.shape[0], x2.shape[1], _stream())
    return idx.view(x.shape[:-1]), prob.view(x.shape[:-1])


def layernorm_fwd(x, gamma, beta, eps=1e-5, out=None):
    """LayerNorm over the last dim of (strided) rows -> (y, mean, rstd)"""
    x2 = rows2d(x)
    rows, C = x2.shape
    if out is None:
        out = torch.empty(*x.shape, device=x.device, dtype=torch.float32)
    o2 = rows2d(out)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    call("mrn_layernorm_fwd_f32", _p(x2), x2.stride(0), _p(gamma), _p(beta), _p(o2), o2.stride(0), _p(mean), _p(rstd),
         rows, C, float(eps), _stream())
    return out, mean, rstd


def layernorm_fwd_operand(x, gamma, beta, eps=1e-5, target=None):
    """LayerNorm of contiguous rows, with the result also as the range-scaled HL32 operand of the trained Linear layer it feeds
    -> (y, mean, rstd, y_hl bytes, scale {s, 1/s}); the scale comes from the parameters (sqrt(C) max|gamma| + max|beta| bounds |y|)"""
    assert x.is_contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    hl = torch.empty(x.numel() * 4, device=x.device, dtype=torch.uint8)
    sc = torch.empty(2, device=x.device, dtype=torch.float32)
    call("mrn_layernorm_fwd_hl32_f32", _p(x), C, _p(gamma), _p(beta), _p(out), C, _p(mean), _p(rstd), rows, C, float(eps), _p(hl), _p(sc),
         float(FP16_WEIGHT_PEAK if target is None else target), _stream())
    return out, mean, rstd, hl, sc


def ew_operand(op, a, b=None, drop=None, rows_per_drop=1, scale=None, want_hl=False, want_amax=None, want_f32=True):
    """y = op(a, b) over contiguous [..., C] (EW_GELU, EW_GELU_BWD, EW_ADD, EW_RESIDUAL_SCALE: a + b * drop[row // rows_per_drop]) with the
    trained consumer's needs folded in: want_hl -> the HL32 operand split(scale[0] * y); want_amax = range target -> max|y| folded and the
    exact power-of-two scale {s, 1/s} returned.  -> (y or None, y_hl or None, scale or None)"""
    C = a.shape[-1]
    rows = a.numel() // C
    assert a.is_contiguous() and (b is None or (b.is_contiguous() and b.numel() == a.numel()))
    y = torch.empty_like(a) if want_f32 else None
    hl = torch.empty(a.numel() * 4, device=a.device, dtype=torch.uint8) if want_hl else None
    ws = _amax_ws() if want_amax is not None else None
    call("mrn_ew_operand_f32", _p(a), _p(b), _p(drop), int(rows_per_drop), _p(y), _p(hl), _p(scale), ws, rows, C, int(op), _stream())
    sc = None
    if ws is not None:
        sc = torch.empty(2, device=a.device, dtype=torch.float32)
        call("mrn_pow2_finalize_f32", float(want_amax), _p(sc), ws, _stream())
    return y, hl, sc


EW_RESIDUAL_SCALE = 8


# trained Linear layers find their operand -- already split, with its range scale -- where the producing pass left it (SVTR blocks of
# loop A: LayerNorm -> qkv / fc1, attention -> proj, GELU -> fc2), or the range scale alone (a bound of max|.| for a producer that
# derives the next operand from this tensor: qkv -> attention, fc1 -> GELU); keyed by the tensor object, dropped with it
_OPERANDS = {}
_GRAD_OPERANDS = {}                 # the same for gradients inside a backward pass: strong references (a gradient's Python object lives only
                                    # while autograd hands it on), dropped when the pass ends (join_side_stream) or the table grows


OPERAND_STATS = {"hits": 0, "stale": 0}       # cached_operand: operands served / entries refused because the tensor changed in place


def stash_operand(t, hl, scale, grad=False):
    """the producer of t leaves t's split operand (hl) and / or its range scale for the Linear that consumes t; the entry is valid for
    this tensor OBJECT at its current version counter (an in-place op on t between producer and consumer invalidates it)"""
    if grad:
        if len(_GRAD_OPERANDS) > 512:
            _GRAD_OPERANDS.clear()
        _GRAD_OPERANDS[id(t)] = (t, hl, scale, t._version)
        return
    if _GRAD_OPERANDS:
        _GRAD_OPERANDS.clear()          # a forward pass is running: whatever the last backward pass left (one without a side-stream join) is dead
    if len(_OPERANDS) > 48:         # (entries own the HL32 buffers: dead owners' entries go before they pile up -- a block leaves four per step)
        for k in [k for k, v in _OPERANDS.items() if v[0]() is None]:
            del _OPERANDS[k]
    _OPERANDS[id(t)] = (weakref.ref(t), hl, scale, t._version)


def cached_operand(t):
    """-> (hl or None, scale) left by the producer of t, or None (no entry, another tensor at the same address, or t modified in place
    since the entry was made)"""
    got = _GRAD_OPERANDS.get(id(t))
    if got is not None and got[0] is t:
        if got[3] != t._version:
            OPERAND_STATS["stale"] += 1
            return None
        OPERAND_STATS["hits"] += 1
        return got[1], got[2]
    got = _OPERANDS.get(id(t))
    if got is None or got[0]() is not t:
        return None
    if got[3] != t._version:
        OPERAND_STATS["stale"] += 1
        return None
    OPERAND_STATS["hits"] += 1
    return got[1], got[2]


def layernorm_bwd(dy, x, gamma, mean, rstd, dx=None, accumulate=False, grad_acc=None):
    """-> (dx, dgamma, dbeta); grad_acc: [2C] buffer ([weight.grad | bias.grad], adjacent in the flat gradient) the reduction of the
    partials ADDS into -- (dx, None, None) is returned"""
    dy2, x2 = rows2d(dy), rows2d(x)
    rows, C = x2.shape
    if dx is None:
        dx = torch.empty(*x.shape, device=x.device, dtype=torch.float32)
        accumulate = False
    dx2 = rows2d(dx)
    nblk = call("mrn_layernorm_bwd_blocks", rows)
    part = torch.empty(nblk, 2 * C, device=x.device, dtype=torch.float32)
    call("mrn_layernorm_bwd_f32", _p(dy2), dy2.stride(0), _p(x2), x2.stride(0), _p(gamma), _p(mean), _p(rstd), _p(dx2),
         dx2.stride(0), int(accumulate), _p(part), rows, C, _stream())
    if grad_acc is not None:
        colsum(part, out=grad_acc, accumulate=True)
        return dx, None, None
    dgb = colsum(part)
    return dx, dgb[:C], dgb[C:]


def colnorm_fwd(x, gamma, beta, eps=1e-5):
    """x [B,P,W] contiguous: LayerNorm over P for every (b, w) -> (y, mean[B,W], rstd[B,W])"""
    B, P, Wd = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x)
    mean = torch.empty(B, Wd, device=x.device, dtype=torch.float32)
    rstd = torch.empty(B, Wd, device=x.device, dtype=torch.float32)
    call("mrn_colnorm_fwd_f32", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), B, P, Wd, float(eps), _stream())
    return y, mean, rstd


def colnorm_bwd(dy, x, gamma, mean, rstd, dx=None, accumulate=False):
    B, P, Wd = x.shape
    assert x.is_contiguous() and dy.is_contiguous()
    if dx is None:
        dx = torch.empty_like(x)
        accumulate = False
    nblk = B * ((Wd + 255) // 256)
    part = torch.empty(nblk, 2 * P, device=x.device, dtype=torch.float32)
    call("mrn_colnorm_bwd_f32", _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), int(accumulate), _p(part), B, P, Wd,
         _stream())
    dgb = colsum(part)
    return dx, dgb[:P], dgb[P:]


EW_GELU, EW_GELU_BWD, EW_MUL, EW_ADD = 0, 1, 2, 3
EW_SIGMOID, EW_SIGMOID_BWD, EW_ADD_RELU = 5, 6, 7        # (4 = EW_RELU_BWD, defined with the backward ops)


def bn_stats(x):
    """per-block partial sums / sums of squares of x [..., C] over its rows: the `stats` input of bn_finalize() for tensors
    that are not conv outputs"""
    _chk(x)
    assert x.is_contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    part = torch.empty(call("mrn_bn_stats_blocks", rows), 2, C, device=x.device, dtype=torch.float32)
    call("mrn_bn_stats_f32", _p(x), rows, C, _p(part), _stream())
    return part


def ew_rows(op, a, b=None, out=None):
    a2 = rows2d(a)
    rows, C = a2.shape
    if out is None:
        out = torch.empty(*a.shape, device=a.device, dtype=torch.float32)
    o2 = rows2d(out)
    b2 = rows2d(b) if b is not None else None
    call("mrn_ew_rows_f32", _p(a2), a2.stride(0), _p(b2), b2.stride(0) if b2 is not None else 0, _p(o2), o2.stride(0),
         rows, C, op, _stream())
    return out


def colsum(x, out=None, accumulate=False):
    """sum over rows of a (strided) 2-D view -> [C]"""
    x2 = rows2d(x)
    rows, C = x2.shape
    if out is None:
        out = torch.empty(C, device=x.device, dtype=torch.float32)
        accumulate = False
    chunks = call("mrn_colsum_chunks", rows, C)
    ws = torch.empty(chunks * C, device=x.device, dtype=torch.float32) if chunks > 1 else None
    call("mrn_colsum_f32", _p(x2), x2.stride(0), _p(out), _p(ws), rows, C, int(accumulate), _stream())
    return out


def gather2d(x, row_idx, col_idx, ld_out=None):
    """out[i,j] = x[row_idx[i], col_idx[j]] (int32 device index tensors or None); out has row stride ld_out (zero padded)"""
    R = x.shape[0] if row_idx is None else row_idx.numel()
    C = x.shape[1] if col_idx is None else col_idx.numel()
    ld_out = C if ld_out is None else ld_out
    out = torch.empty(R, ld_out, device=x.device, dtype=torch.float32)
    call("mrn_gather2d_f32", _p(x), x.stride(0), _p(row_idx), _p(col_idx), _p(out), ld_out, R, C, _stream())
    return out[:, :C]


# ---------------------------------------------------------------------------------------------------------
# fan-in / gate tail
# ---------------------------------------------------------------------------------------------------------
import ctypes as _ct  # noqa: E402


def _fanin_args(logits):
    I = len(logits)
    ptrs = (_ct.c_void_p * I)(*[l.data_ptr() for l in logits])
    lds = (_ct.c_int64 * I)(*[l.stride(-2) for l in logits])
    cls = (_ct.c_int * I)(*[l.shape[-1] for l in logits])
    for l in logits:
        assert l.dim() == 3 and l.stride(2) == 1 and l.stride(0) == l.shape[1] * l.stride(1)
    return I, ptrs, lds, cls


def padded_rows(B, T, C, device):
    """[B,T,C] view of a buffer whose rows are padded to a multiple of 4 floats (16-byte aligned rows)."""
    ld = (C + 3) // 4 * 4
    return torch.empty(B, T, ld, device=device, dtype=torch.float32)[:, :, :C]


def fanin_fwd(logits, w):
    I, ptrs, lds, cls = _fanin_args(logits)
    B, T, C = logits[-1].shape
    out = padded_rows(B, T, C, w.device)
    call("mrn_fanin_fwd_f32", ptrs, lds, cls, I, _p(w), _p(out), out.stride(1), B, T, C, _stream())
    return out


def fanin_bwd(logits, dout):
    I, ptrs, lds, cls = _fanin_args(logits)
    B, T, C = logits[-1].shape
    assert dout.stride(2) == 1 and dout.stride(0) == T * dout.stride(1)
    dw = torch.empty(B, I, device=dout.device, dtype=torch.float32)
    ws = torch.empty(B * T * I, device=dout.device, dtype=torch.float32)
    call("mrn_fanin_bwd_f32", ptrs, lds, cls, I, _p(dout), dout.stride(1), _p(dw), _p(ws), B, T, C, _stream())
    return dw


def select_expert(logits, index):
    I, ptrs, lds, cls = _fanin_args(logits)
    B, T, C = logits[-1].shape
    out = torch.empty(B, T, C, device=index.device, dtype=torch.float32)
    call("mrn_select_expert_f32", ptrs, lds, cls, I, _p(index), _p(out), C, B, T, C, _stream())
    return out


def gate_tail_fwd(r, w_route, b_route, beta=1.0, hard=False):
    """r [B,P,I] -> (s, w) or (s, argmax)"""
    B, P, I = r.shape
    s = torch.empty(B, I, device=r.device, dtype=torch.float32)
    if hard:
        am = torch.empty(B, device=r.device, dtype=torch.int64)
        call("mrn_gate_tail_fwd_f32", _p(r), _p(w_route), _p(b_route), float(beta), _p(s), None, _p(am), B, P, I, _stream())
        return s, am
    w = torch.empty(B, I, device=r.device, dtype=torch.float32)
    call("mrn_gate_tail_fwd_f32", _p(r), _p(w_route), _p(b_route), float(beta), _p(s), _p(w), None, B, P, I, _stream())
    return s, w


def gate_tail_bwd(w, dw, r, w_route, beta=1.0):
    B, P, I = r.shape
    ds = torch.empty(B, I, device=r.device, dtype=torch.float32)
    dr = torch.empty(B, P, I, device=r.device, dtype=torch.float32)
    dW = torch.empty(P, device=r.device, dtype=torch.float32)
    db = torch.empty(1, device=r.device, dtype=torch.float32)
    call("mrn_gate_tail_bwd_f32", _p(w), _p(dw.contiguous()), _p(r), _p(w_route), float(beta), _p(ds), _p(dr), _p(dW), _p(db),
         B, P, I, _stream())
    return dr, dW, db


# ---------------------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------------------
def ce_loss_fwd(logits, target, ignore_index=-100):
    """logits [..., C] (strided rows), target [...] int64 -> (loss scalar tensor, ctx)"""
    l2 = rows2d(logits)
    rows, C = l2.shape
    t = target.reshape(-1).contiguous()
    dev = logits.device
    lse = torch.empty(rows, device=dev, dtype=torch.float32)
    lrow = torch.empty(rows, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    inv = torch.empty(1, device=dev, dtype=torch.float32)
    call("mrn_ce_loss_fwd_f32", _p(l2), l2.stride(0), _p(t), ignore_index, rows, C, _p(lse), _p(lrow), _p(loss), _p(inv),
         _stream())
    return loss, (l2, t, ignore_index, lse, inv)


def ce_loss_bwd(ctx, upstream, like):
    l2, t, ignore_index, lse, inv = ctx
    rows, C = l2.shape
    d = torch.empty_strided(like.shape, like.stride(), device=like.device, dtype=torch.float32)
    d2 = rows2d(d)
    call("mrn_ce_loss_bwd_f32", _p(l2), l2.stride(0), _p(t), ignore_index, _p(lse), _p(upstream), _p(inv), _p(d2),
         d2.stride(0), rows, C, _stream())
    return d


def ctc_loss_fwd(logits, targets, target_len, blank=0):
    """logits [B,T,C] (strided rows), targets [B,L] int64, target_len [B] int32 -> (loss, ctx)"""
    B, T, C = logits.shape
    assert logits.stride(2) == 1 and logits.stride(0) == T * logits.stride(1)
    dev = logits.device
    targets = targets.contiguous()
    target_len = target_len.to(torch.int32).contiguous()
    lse = torch.empty(B * T, device=dev, dtype=torch.float32)
    nll = torch.empty(B, device=dev, dtype=torch.float32)
    occ = torch.empty(call("mrn_ctc_occ_floats", B, T), device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    call("mrn_ctc_loss_fwd_f32", _p(logits), logits.stride(1), _p(targets), targets.stride(0), _p(target_len),
         targets.shape[1], _p(lse), _p(nll), _p(occ), _p(loss), B, T, C, blank, _stream())
    return loss, (logits, targets, target_len, lse, nll, occ, blank)


def ctc_loss_bwd(ctx, upstream):
    logits, targets, target_len, lse, nll, occ, blank = ctx
    B, T, C = logits.shape
    d = torch.empty_strided(logits.shape, logits.stride(), device=logits.device, dtype=torch.float32)
    call("mrn_ctc_loss_bwd_f32", _p(logits), logits.stride(1), _p(lse), _p(occ), _p(targets), targets.stride(0),
         _p(target_len), _p(nll), _p(upstream), _p(d), d.stride(1), B, T, C, blank, _stream())
    return d


# ---------------------------------------------------------------------------------------------------------
# optimiser
# ---------------------------------------------------------------------------------------------------------
def grad_norm_clip(gflat, max_norm, out=None):
    """-> device tensor [norm, clip_coef, skipped steps]; `out`: the caller's persistent three-float buffer (zero-initialised once), whose
    third element counts the steps the update kernels skipped because the gradient norm was not finite"""
    n = gflat.numel()
    ws = torch.empty(call("mrn_grad_norm_workspace_floats", n), device=gflat.device, dtype=torch.float32)
    nc = out if out is not None else torch.zeros(3, device=gflat.device, dtype=torch.float32)
    call("mrn_grad_norm_clip_f32", _p(gflat), n, float(max_norm), _p(ws), _p(nc), _stream())
    return nc


def adam_step(p, g, m, v, norm_coef, lr, step, betas=(0.9, 0.999), eps=1e-8):
    bc1 = 1.0 - betas[0] ** step
    bc2 = 1.0 - betas[1] ** step
    call("mrn_adam_step_f32", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(norm_coef), float(lr / bc1), float(betas[0]),
         float(betas[1]), float(bc2 ** 0.5), float(eps), _stream())


def sgd_step(p, g, buf, norm_coef, lr, momentum=0.0, weight_decay=0.0):
    call("mrn_sgd_step_f32", _p(p), _p(g), _p(buf), p.numel(), _p(norm_coef), float(lr), float(momentum), float(weight_decay),
         _stream())


def adadelta_step(p, g, square_avg, acc_delta, norm_coef, lr, rho=0.9, eps=1e-6):
    call("mrn_adadelta_step_f32", _p(p), _p(g), _p(square_avg), _p(acc_delta), p.numel(), _p(norm_coef), float(lr), float(rho),
         float(eps), _stream())


# ---------------------------------------------------------------------------------------------------------
# convolution / BatchNorm / pooling backward
# ---------------------------------------------------------------------------------------------------------
EW_RELU_BWD = 4


def pack_dgrad_weight(w_ohwi):
    """[O,kh,kw,I] -> PackedConvWeight of the data-gradient conv: [I,kh,kw,O], taps flipped"""
    O, kh, kw, I = w_ohwi.shape
    out = torch.empty(I, kh, kw, O, device=w_ohwi.device, dtype=torch.float32)
    call("mrn_pack_dgrad_weight_f32", _p(w_ohwi), _p(out), O, I, kh, kw, _stream())
    return PackedConvWeight(out)


def conv2d_dgrad(dy, wt_packed, x_hw, stride, padding, precision=None, dy_scale=None):
    """dy [B,Ho,Wo,Cout] -> dx [B,H,W,Cin] with wt_packed = pack_dgrad_weight(w)"""
    B, Ho, Wo, Cout = dy.shape
    Cin, kh, kw, _ = wt_packed.shape
    H, W = x_hw
    # full correlation with the flipped kernel.  Input rows / columns the floor of the output-size formula left over still sit
    # under kernel taps of the last output row (they overlap what would be trailing padding), so the dilated gradient gets that
    # many extra trailing zero rows: the correlation then produces exactly [H, W]
    ph, pw = kh - 1 - padding[0], kw - 1 - padding[1]
    Hd, Wd = (Ho - 1) * stride[0] + 1, (Wo - 1) * stride[1] + 1
    eh, ew = H - (Hd + 2 * ph - kh + 1), W - (Wd + 2 * pw - kw + 1)
    assert eh >= 0 and ew >= 0
    if stride != (1, 1) or eh or ew:
        d = torch.empty(B, Hd + eh, Wd + ew, Cout, device=dy.device, dtype=torch.float32)
        call("mrn_dilate_nhwc_f32", _p(dy), _p(d), B, Ho, Wo, Cout, stride[0], stride[1], eh, ew, _stream())
        dy = d
    dx, _ = conv2d_nhwc(dy, wt_packed, None, (1, 1), (ph, pw), precision=precision, x_scale=dy_scale)   # (dilation adds only zeros)
    return dx


def conv2d_wgrad(dy, x, ksize, stride, padding):
    """-> dW [Cout,kh,kw,Cin] (packed layout)"""
    B, Ho, Wo, Cout = dy.shape
    _, H, W, Cin = x.shape
    kh, kw = ksize
    pixels = B * Ho * Wo
    tiles = ((Cout + 127) // 128) * ((kh * kw * Cin + 127) // 128)
    S = 1
    want = max(1, min(1024 // max(tiles, 1), pixels // 256))
    for s_ in range(want, 0, -1):
        if pixels % s_ == 0:
            S = s_
            break
    part = torch.empty(S, Cout, kh * kw * Cin, device=dy.device, dtype=torch.float32)
    call("mrn_conv2d_wgrad_f32", _p(dy), _p(x), _p(part), B, H, W, Cin, Cout, kh, kw, stride[0], stride[1],
         padding[0], padding[1], S, _stream())
    dw = part[0] if S == 1 else colsum(part.view(S, -1))
    return dw.view(Cout, kh, kw, Cin)


def unpack_conv_weight(g_ohwi, out=None, accumulate=False):
    """[O,kh,kw,I] -> the parameter's [O,I,kh,kw] layout; out / accumulate: added into an existing gradient buffer"""
    O, kh, kw, I = g_ohwi.shape
    if out is None:
        out = torch.empty(O, I, kh, kw, device=g_ohwi.device, dtype=torch.float32)
        accumulate = False
    assert out.is_contiguous() and tuple(out.shape) == (O, I, kh, kw)
    call("mrn_unpack_conv_weight_f32", _p(g_ohwi.contiguous()), _p(out), O, I, kh, kw, int(bool(accumulate)), _stream())
    return out


# ---- weight gradients off the critical path --------------------------------------------------------------------------------
# loss.backward() of a trained expert is a chain: BatchNorm backward -> data gradient -> the previous layer's BatchNorm backward ...
# The weight gradient of a layer (two operand transposes, the K-window GEMM, the fold back to [O,I,kh,kw]) hangs off that chain.
# With WGRAD_SIDE_STREAM it is issued on a second HIP stream and ACCUMULATED straight into the parameter's .grad (the flat gradient
# of optim.FlatOptimizer), so its many short launches fill the tails of the chain's kernels; the backward pass's final callback
# joins the two streams before anyone reads a gradient.  Only inside `with ops.direct_gradients():` (the learners' backward_and_step; with
# N > 1 ranks the bucketed all-reduce is told about every completed parameter through direct_gradients(notify=...)).
WGRAD_SIDE_STREAM = os.environ.get("MRN_WGRAD_STREAM", "1") == "1"
GRAD_DIRECT = False                 # set by direct_gradients(): only a caller that runs loss.backward() INTO .grad may skip autograd's accumulation
_GRAD_GENS = [()]                   # direct_gradients(notify=..., roots=...): forward generations of the graph being run backward
_GRAD_NOTIFY = [None]               # direct_gradients(notify=...): called with a Parameter once ALL its side-stream accumulations are issued
# Forward uses (by functions that may accumulate directly) not yet matched by a backward, PER FORWARD GENERATION: a function's forward
# stores the generation it was counted under on its ctx (ctx.use_gen) and its backward ticks the counts of THAT generation, so graphs
# whose backward runs elsewhere (torch.autograd.grad: Fisher passes; grad-enabled validation) or interleaved graphs (fwd A, fwd B,
# bwd A, bwd B) never touch each other's counts.  A generation ends when a backward under direct_gradients exits or a data-parallel
# wrapper starts a forward (new_use_generation); the table keeps the last few.  A parameter that is MISSING from its generation's
# table is never reported complete by direct_done -- the bucketed all-reduce's finish() picks it up after the join.
_PARAM_USES = {0: {}}               # generation -> {id(Parameter): uses}
_USE_GEN = [0]
_USE_GEN_KEEP = 8
_SIDE_STREAMS = {}
_SIDE_PENDING = [False]
_SIDE_KEEP = []                     # tensors the side stream still reads, held until the join (see side_stream_keep)


class direct_gradients:
    """with ops.direct_gradients(): loss.backward() -- inside, backward functions may accumulate a parameter's gradient into its .grad
    themselves (and return None for it).  Not for torch.autograd.grad(), which wants the gradients returned: hence opt-in per call
    (il_modules/base.py backward_and_step).  `notify(param)`: autograd's post-accumulate hooks never see such a gradient, so a bucketed
    all-reduce (parallel.BucketedAllReduce, N > 1) passes its param_ready here; it is called when the LAST forward use of the parameter
    has had its accumulation issued on the side stream (note_param_uses counts the uses).  On exit -- also when backward raised -- the
    streams are joined and the references the side stream held are dropped."""

    def __init__(self, notify=None, roots=()):
        self.notify = notify
        self.roots = roots           # the tensors backward() is about to be called on: with notify, their graph is walked for the
        #                              forward generations it holds (a loss fed by two wrapper forwards spans two)

    def __enter__(self):
        global GRAD_DIRECT
        self.prev, GRAD_DIRECT = GRAD_DIRECT, True
        self.prev_notify, _GRAD_NOTIFY[0] = _GRAD_NOTIFY[0], self.notify
        self.prev_gens, _GRAD_GENS[0] = _GRAD_GENS[0], (graph_generations(self.roots) if self.notify is not None else ())
        return self

    def __exit__(self, *exc):
        global GRAD_DIRECT
        GRAD_DIRECT = self.prev
        _GRAD_NOTIFY[0] = self.prev_notify
        _GRAD_GENS[0] = self.prev_gens
        new_use_generation()
        join_side_stream()           # (normally done by the backward pass's final callback; an exception inside backward skips that)
        return False


def graph_generations(roots):
    """the forward generations (ctx.use_gen of our autograd Functions) found in the graphs behind `roots`"""
    gens, seen = set(), set()
    stack = [t.grad_fn for t in roots if t is not None and t.grad_fn is not None]
    while stack:
        node = stack.pop()
        if node in seen:
            continue
        seen.add(node)
        g = getattr(node, "use_gen", None)
        if g is not None:
            gens.add(g)
        stack.extend(fn for fn, _ in node.next_functions if fn is not None)
    return tuple(gens)


def new_use_generation():
    """later forwards count their parameter uses in a fresh table (called when a backward under direct_gradients has run and by the
    data-parallel wrapper at the start of every grad-enabled forward)"""
    _USE_GEN[0] += 1
    _PARAM_USES[_USE_GEN[0]] = {}
    for old in [g for g in _PARAM_USES if g <= _USE_GEN[0] - _USE_GEN_KEEP]:
        del _PARAM_USES[old]
    return _USE_GEN[0]


def note_param_uses(params, recording=True):
    """forward of a function whose backward may accumulate these parameters' gradients on the side stream: one more use to wait for
    before the parameter counts as complete (a recurrent conv layer applies one weight several times).  `recording`: the call is
    being recorded by autograd (any(ctx.needs_input_grad); grad mode itself is off inside a Function's forward).
    -> the generation the uses were counted under: the caller keeps it on its ctx and hands it to direct_done."""
    gen = _USE_GEN[0]
    if recording:
        table = _PARAM_USES[gen]
        for p in params:
            if p is not None and p.requires_grad:
                table[id(p)] = table.get(id(p), 0) + 1
    return gen


DIRECT_STATS = {"parameters": 0, "notified": 0}    # side-stream accumulations issued / parameters reported complete (tests, bench telemetry)


def direct_done(params, gen=None):
    """the side-stream accumulation of one use of each of `params` has been issued; gen: what note_param_uses returned in the
    forward of the calling function.  A parameter is reported to direct_gradients(notify=...) when the count of ITS generation
    reaches zero AND no other generation of the running backward's graph still holds uses of it; one without a count (generation
    dropped, forward not recorded) is left to the reducer's finish()."""
    notify = _GRAD_NOTIFY[0]
    table = _PARAM_USES.get(gen)
    for p in params:
        if p is None:
            continue
        DIRECT_STATS["parameters"] += 1
        if table is None or id(p) not in table:
            continue
        n = table[id(p)] - 1
        table[id(p)] = n
        # two grad-enabled wrapper forwards that feed ONE backward count the same parameter in two tables: complete only when every
        # generation this backward's graph holds (direct_gradients(roots=...)) has run dry, not just the calling function's own
        if n == 0 and notify is not None and not any(_PARAM_USES.get(g, {}).get(id(p), 0) > 0 for g in _GRAD_GENS[0] if g != gen):
            DIRECT_STATS["notified"] += 1
            notify(p)


def side_stream_pending():
    return _SIDE_PENDING[0]


def side_stream():
    dev = torch.cuda.current_device()
    st = _SIDE_STREAMS.get(dev)
    if st is None:
        st = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return st


_AUX_STREAMS = {}


def aux_stream(i, device=None):
    """the i-th PROCESS-WIDE auxiliary stream of a device (lock-step sub-groups of the frozen experts, the heads, DER's frozen
    extractors, LwF's previous network).  Every model takes the same streams instead of drawing fresh ones: torch hands streams out
    of a pool round-robin and the runtime maps them onto a few hardware queues in creation order, so the third learner of a process
    got streams that shared a queue -- SVTR x 6 (three concurrent sub-groups) ran 7 % slower as the last short line of a default
    bench run than alone (21.5 vs 19.8 ms; host issue time equal), and the reduced-mode DER line 9 %."""
    dev = torch.cuda.current_device() if device is None else (device.index if isinstance(device, torch.device) else int(device))
    if dev is None:
        dev = torch.cuda.current_device()
    pool = _AUX_STREAMS.setdefault(dev, [])
    while len(pool) <= i:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[i]


def join_side_stream():
    """the current stream waits for everything issued on the side stream (called once at the end of a backward pass)"""
    if _SIDE_PENDING[0]:
        _SIDE_PENDING[0] = False
        torch.cuda.current_stream().wait_stream(side_stream())
    _SIDE_KEEP.clear()
    _GRAD_OPERANDS.clear()


def side_stream_keep(tensors):
    """hold references to what the side stream reads until the join.  record_stream() only keeps a FREED block from being reused; a
    gradient that autograd still owns can be ACCUMULATED INTO IN PLACE on the main stream once our reference is gone (InputBuffer adds in
    place when it holds the last reference) -- seen as NaN weights after one SVTR step.  With a reference held, autograd sums out of place."""
    for t in tensors:
        if t is not None:
            _SIDE_KEEP.append(t)
            t.record_stream(side_stream())


def side_stream_begin():
    """-> the side stream, ordered behind everything issued so far on the current stream; the first use inside a backward pass queues
    the join as that pass's final callback"""
    st = side_stream()
    st.wait_stream(torch.cuda.current_stream())
    if not _SIDE_PENDING[0]:
        _SIDE_PENDING[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
    return st


# BatchNorm2d.num_batches_tracked += 1 of every train-mode layer: inside `with batch_counters():` (one expert's backbone forward,
# modules/model.py visual()) the increments are collected and issued as ONE multi-tensor launch at the end (36 launches -> 1 on TRBA)
_NBT_PENDING, _NBT_DEPTH = [], [0]


def count_batch(bn):
    if bn.num_batches_tracked is None:
        return
    if _NBT_DEPTH[0] > 0:
        _NBT_PENDING.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


class batch_counters:
    def __enter__(self):
        _NBT_DEPTH[0] += 1

    def __exit__(self, *exc):
        _NBT_DEPTH[0] -= 1
        if _NBT_DEPTH[0] == 0 and _NBT_PENDING:
            pending = list(_NBT_PENDING)
            _NBT_PENDING.clear()
            while pending:                           # a layer called twice is listed twice: one round per multiplicity (no in-launch races)
                seen, first, rest = set(), [], []
                for t in pending:
                    (rest if id(t) in seen else first).append(t)
                    seen.add(id(t))
                torch._foreach_add_(first, 1)
                pending = rest
        return False


def bn_bwd(dz, z, y, mean, invstd, gamma, relu, want_dres=False, range_target=None, zmask=None, grad_acc=None):
    """BatchNorm2d(train) backward with fused ReLU mask -> (dy, dgamma, dbeta, dres or None[, range scale of dy]); range_target: max|dy|
    is folded into the apply pass and the power-of-two scale {s, 1/s} with s * max|dy| <= range_target is returned as a fifth value.
    zmask: the forward pass's ReLU bit mask (scale_shift_act pos_mask) in place of z.  grad_acc = (weight.grad, bias.grad): dgamma / dbeta
    are ADDED there by the finalize launch and returned as None"""
    C = y.shape[-1]
    rows = y.numel() // C
    nblk = call("mrn_bn_bwd_blocks", rows)
    part = torch.empty(nblk, 2 * C, device=y.device, dtype=torch.float32)
    call("mrn_bn_bwd_reduce_f32", _p(dz), _p(z), _p(zmask), _p(y), _p(mean), _p(invstd), _p(part), rows, C, int(relu), _stream())
    sums = torch.empty(2 * C, device=y.device, dtype=torch.float32)          # sum g | sum g * xhat
    gw, gb = grad_acc if grad_acc is not None else (None, None)
    call("mrn_bn_bwd_finalize_f32", _p(part), nblk, C, _p(sums), _p(gw), _p(gb), _stream())
    dgamma, dbeta = (None, None) if grad_acc is not None else (sums[C:], sums[:C])
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    ws = _amax_ws() if (range_target is not None and FUSED_AMAX) else None
    call("mrn_bn_bwd_apply_f32", _p(dz), _p(z), _p(zmask), _p(y), _p(mean), _p(invstd), _p(gamma), _p(sums), _p(dy), _p(dres), rows, C,
         int(relu), ws, _stream())
    if range_target is None:
        return dy, dgamma, dbeta, dres
    if ws is None:
        return dy, dgamma, dbeta, dres, pow2_scale(dy, range_target)
    sc = torch.empty(2, device=y.device, dtype=torch.float32)
    call("mrn_pow2_finalize_f32", float(range_target), _p(sc), ws, _stream())
    return dy, dgamma, dbeta, dres, sc


def maxpool_bwd(dy, x, kernel, stride, padding):
    B, H, W, C = x.shape
    x, dy = x.contiguous(), dy.contiguous()
    # non-overlapping windows: the kernel writes every element of dx -- but only its 16-byte-aligned form does (an offset view falls
    # back to the accumulating kernel, which needs zeros): the pointers are part of the decision
    writes_all = (call("mrn_maxpool_bwd_writes_all", H, W, C, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1])
                  and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)
    dx = torch.empty_like(x) if writes_all else torch.zeros_like(x)
    call("mrn_maxpool_bwd_nhwc_f32", _p(dy), _p(x), _p(dx), B, H, W, C, kernel[0], kernel[1], stride[0], stride[1],
         padding[0], padding[1], _stream())
    return dx


# ---------------------------------------------------------------------------------------------------------
# attention decoder / TPS backward
# ---------------------------------------------------------------------------------------------------------
def attn_decoder_train(Hb, Hproj, eproj, w_h2h, b_h2h, w_score, w_ih, w_hh, b_hh, hidden, w_inv=None):
    """teacher-forced forward that also returns what the backward needs: (hid, saves); w_inv: the x3 form (pack_decoder_x3)"""
    B, T, D = Hb.shape
    S = eproj.shape[1]
    dev = Hb.device
    hid = torch.empty(B, S, hidden, device=dev, dtype=torch.float32)
    alpha = torch.empty(B, S, T, device=dev, dtype=torch.float32)
    gates = torch.empty(B, S, 4 * hidden, device=dev, dtype=torch.float32)
    cseq = torch.empty(B, S, hidden, device=dev, dtype=torch.float32)
    ctx = torch.empty(B, S, D, device=dev, dtype=torch.float32)
    hp = torch.empty(B, S, hidden, device=dev, dtype=torch.float32)
    if w_inv is not None:
        call("mrn_attn_decoder_fwd_x3", _p(Hb), _p(Hproj), _p(eproj), eproj.stride(0), eproj.stride(1), _p(w_h2h),
             _p(b_h2h), _p(w_score), _p(w_ih), _p(w_hh), _p(w_inv), _p(b_hh), _p(hid), hid.stride(0), hid.stride(1), None, None,
             _p(alpha), _p(gates), _p(cseq), _p(ctx), _p(hp), B, T, D, S, hidden, _stream())
        return hid, (alpha, gates, cseq, ctx, hp)
    call("mrn_attn_decoder_fwd_f32", _p(Hb), _p(Hproj), _p(eproj), eproj.stride(0), eproj.stride(1), _p(w_h2h),
         _p(b_h2h), _p(w_score), _p(w_ih), _p(w_hh), _p(b_hh), _p(hid), hid.stride(0), hid.stride(1), None, None,
         _p(alpha), _p(gates), _p(cseq), _p(ctx), _p(hp), B, T, D, S, hidden, _stream())
    return hid, (alpha, gates, cseq, ctx, hp)


def attn_decoder_bwd(Hb, Hproj, saves, dhid, w_score, w_h2hT, w_ih_ctxT, w_hhT, hidden, w_inv=None):
    """-> dgates [B,S,4H], dhp [B,S,H], dHb [B,T,D], dHproj [B,T,H], dw_score [H]; w_inv: the x3 form (the three weights as
    pack_fragment_major_h streams of the transposes, w_inv float[3])"""
    alpha, gates, cseq, ctx, hp = saves
    B, T, D = Hb.shape
    S = alpha.shape[1]
    dev = Hb.device
    dgates = torch.empty(B, S, 4 * hidden, device=dev, dtype=torch.float32)
    dhp = torch.empty(B, S, hidden, device=dev, dtype=torch.float32)
    dHb = torch.empty(B, T, D, device=dev, dtype=torch.float32)
    dHproj = torch.empty(B, T, hidden, device=dev, dtype=torch.float32)
    dctx = torch.empty(B, S, D, device=dev, dtype=torch.float32)          # per-step d context / d score: the step loop writes them, two
    de = torch.empty(B, S, T, device=dev, dtype=torch.float32)            # launches behind it form dHb / dHproj (written once)
    nwg = call("mrn_attn_decoder_bwd_parts", B)
    dws = torch.empty(nwg, hidden, device=dev, dtype=torch.float32)
    dhid = dhid.contiguous()
    if w_inv is not None:
        gs = pow2_scale(dhid, 16.0)
        call("mrn_attn_decoder_bwd_x3", _p(Hb), _p(Hproj), _p(alpha), _p(gates), _p(cseq), _p(ctx), _p(hp), _p(dhid),
             _p(w_score), _p(w_h2hT), _p(w_ih_ctxT), _p(w_hhT), _p(w_inv), _p(gs), _p(dgates), _p(dhp), _p(dHb), _p(dHproj), _p(dws), _p(dctx), _p(de), B, T, D, S,
             hidden, _stream())
    else:
        call("mrn_attn_decoder_bwd_f32", _p(Hb), _p(Hproj), _p(alpha), _p(gates), _p(cseq), _p(ctx), _p(hp), _p(dhid),
             _p(w_score), _p(w_h2hT), _p(w_ih_ctxT), _p(w_hhT), _p(dgates), _p(dhp), _p(dHb), _p(dHproj), _p(dws), _p(dctx), _p(de), B, T, D, S,
             hidden, _stream())
    return dgates, dhp, dHb, dHproj, (colsum(dws) if nwg > 1 else dws[0])


def embed_scatter_add(idx, demb, num_class):
    B, S = idx.shape
    E = demb.shape[-1]
    dtable = torch.zeros(num_class, E, device=demb.device, dtype=torch.float32)
    call("mrn_embed_scatter_add_f32", _p(idx), idx.stride(0), _p(demb.contiguous()), _p(dtable), B, S, E, num_class, _stream())
    return dtable


def tps_grid_sample_bwd(img_nhwc, cprime, inv_delta_c, p_hat, dout_nhwc):
    B, H, W, C = img_nhwc.shape
    _, Hr, Wr, _ = dout_nhwc.shape
    F = cprime.shape[1]
    d = torch.empty(B, F, 2, device=cprime.device, dtype=torch.float32)
    call("mrn_tps_grid_sample_bwd_f32", _p(img_nhwc), _p(cprime.contiguous()), _p(inv_delta_c), _p(p_hat), _p(dout_nhwc.contiguous()),
         _p(d), B, H, W, C, Hr, Wr, F, _stream())
    return d


def avgpool_bwd(dy, HW):
    B, C = dy.shape
    dx = torch.empty(B, HW, C, device=dy.device, dtype=torch.float32)
    call("mrn_avgpool_bwd_nhwc_f32", _p(dy.contiguous()), _p(dx), B, HW, C, _stream())
    return dx


# ---------------------------------------------------------------------------------------------------------
# SVTR helpers
# ---------------------------------------------------------------------------------------------------------
def softmax_rows_(s, mask=None):
    """in-place softmax over the last dim of s [..., Nq, N]; mask [Nq, N] additive, shared across leading dims"""
    assert s.is_contiguous()
    N = s.shape[-1]
    rows = s.numel() // N
    rpm = mask.shape[0] if mask is not None else 1
    call("mrn_softmax_rows_f32", _p(s), _p(mask), rows, N, rpm, _stream())
    return s


def softmax_rows_bwd_(p, dp):
    """in place on dp: ds = p * (dp - rowsum(p * dp))"""
    assert p.is_contiguous() and dp.is_contiguous() and p.shape == dp.shape
    N = p.shape[-1]
    call("mrn_softmax_rows_bwd_f32", _p(p), _p(dp), p.numel() // N, N, _stream())
    return dp


def _mask_bits(mask):
    """visibility bits [N, ceil(N/32)] (int32) of an additive mask whose entries are all 0 or -inf, cached on the tensor; None
    for any other mask (one device->host check per mask tensor, when the cache is built)"""
    got = getattr(mask, "_mrn_bits", None)
    if got is None or got[0] != mask._version:
        vis = mask == 0
        binary = bool(torch.all(vis | torch.isneginf(mask)))
        bits = None
        if binary:
            N, M = mask.shape
            pad = (-M) % 32
            v = torch.nn.functional.pad(vis, (0, pad)).view(N, (M + pad) // 32, 32).to(torch.int64)
            w = (v << torch.arange(32, device=mask.device, dtype=torch.int64)).sum(-1)
            bits = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
        got = (mask._version, bits)
        mask._mrn_bits = got
    return got[1]


def svtr_attention(qkv, heads, scale, mask=None, want_f32=True, want_hl=False, want_lse=False, x3=False, hl_scale=None):
    """qkv [B,N,3C] (q | k | v, head dim 32), mask [N,N] additive symmetric or None -> [B,N,C]: fused q k^T / softmax / attn v.
    want_hl: also (or only) the HL32 operand of the proj Linear; returns the fp32 tensor, the HL32 bytes, or (fp32, hl).
    want_lse: returns (fp32, lse [B,heads,N]) -- what svtr_attention_bwd needs.  x3: split-fp16 x3 products (frozen experts)"""
    _chk(qkv, mask)
    B, N, C3 = qkv.shape
    C = C3 // 3
    assert qkv.is_contiguous() and C == heads * 32 and (mask is None or (mask.is_contiguous() and tuple(mask.shape) == (N, N)))
    out = torch.empty(B, N, C, device=qkv.device, dtype=torch.float32) if want_f32 else None
    hl = torch.empty(B * N * C * 4, device=qkv.device, dtype=torch.uint8) if want_hl else None
    lse = torch.empty(B, heads, N, device=qkv.device, dtype=torch.float32) if want_lse else None
    bits = _mask_bits(mask) if (mask is not None and not want_lse) else None     # (the backward kernels read the additive mask)
    call("mrn_svtr_attention_f32", _p(qkv), None if bits is not None else _p(mask), _p(bits), _p(out), _p(hl), _p(lse), B, N, C,
         heads, float(scale), int(bool(x3) and not want_lse), _p(hl_scale), _stream())
    if want_lse:
        return (out, lse, hl) if want_hl else (out, lse)
    return (out, hl) if (want_f32 and want_hl) else (hl if want_hl else out)


def svtr_attention_bwd(qkv, mask, out, dout, lse, heads, scale, want_range=None):
    """-> dqkv [B,N,3C]; recomputes the probabilities from lse (nothing of size N x N is stored).  want_range = range target: max|dqkv|
    is folded into both kernels and (dqkv, {s, 1/s}) is returned"""
    _chk(qkv, mask, out, dout, lse)
    B, N, C3 = qkv.shape
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and lse.is_contiguous()
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty_like(lse)
    call("mrn_svtr_attention_bwd_f32", _p(qkv), _p(mask), _p(out), _p(dout), _p(lse), _p(dsum), _p(dqkv), B, N, C3 // 3, heads,
         float(scale), _amax_ws() if want_range is not None else None, _stream())
    if want_range is not None:
        return dqkv, pow2_finalize(want_range)
    return dqkv


def add_layernorm_grouped(x, branch=None, drop=None, rows_per_drop=1, gamma=None, beta=None, rows_per_group=None, eps=1e-6,
                          want_sum=False, want_f32=False, want_hl=True):
    """x, branch [..., C] contiguous: t = x + drop[row // rows_per_drop] * branch; y = LayerNorm(t) * gamma[g] + beta[g]
    (gamma, beta [G,C]; None: y = t) -> (t or None, y fp32 or None, y HL32 bytes or None)"""
    _chk(x, branch, drop, gamma, beta)
    C = x.shape[-1]
    rows = x.numel() // C
    assert x.is_contiguous() and (branch is None or (branch.is_contiguous() and branch.numel() == x.numel()))
    t = torch.empty_like(x) if want_sum else None
    y = torch.empty_like(x) if want_f32 else None
    hl = torch.empty(x.numel() * 4, device=x.device, dtype=torch.uint8) if want_hl else None
    call("mrn_add_layernorm_grouped_f32", _p(x), _p(branch), _p(drop), int(rows_per_drop), _p(gamma), _p(beta),
         int(rows_per_group if rows_per_group is not None else rows), _p(t), _p(y), _p(hl), rows, C, float(eps), _stream())
    return t, y, hl


SVTR_FUSED_MLP = os.environ.get("MRN_SVTR_MLP", "fused") == "fused"     # fc1 -> GELU -> fc2 of the frozen SVTR experts in one kernel (C <= 128)
_MLP_PERM = {}


def mlp_hidden_permutation(hidden, device):
    """index tensor P with packed_w2[:, p] = w2[:, P[p]]: inside every 32-block, position 16 s + 8 h + j holds unit
    (j & 3) + 8 (2 s + (j >> 2)) + 4 h -- the order in which the MFMA result registers of fc1 present the hidden units to fc2"""
    key = (hidden, device)
    got = _MLP_PERM.get(key)
    if got is None:
        idx = []
        for blk in range(hidden // 32):
            for pos in range(32):
                s_, h, j = pos >> 4, (pos >> 3) & 1, pos & 7
                idx.append(blk * 32 + (j & 3) + 8 * (2 * s_ + (j >> 2)) + 4 * h)
        got = torch.tensor(idx, dtype=torch.int64, device=device)
        _MLP_PERM[key] = got
    return got


def svtr_mlp_fused(x_hl, rows, rows_per_group, G, C, w1_hl, s1, b1, w2_hl, s2, b2):
    """y [rows, C] = fc2(GELU(fc1(x))) per group (mrn_svtr_mlp_x3_f32); w2_hl packed from the hidden-permuted fc2 weights"""
    y = torch.empty(rows, C, device=x_hl.device, dtype=torch.float32)
    t0 = CONV_TIMER.begin() if CONV_TIMER is not None else None
    call("mrn_svtr_mlp_x3_f32", _p(x_hl), _p(w1_hl), _p(s1), _p(b1), _p(w2_hl), _p(s2), _p(b2), _p(y), rows, rows_per_group, G, C,
         _stream())
    if t0 is not None:
        CONV_TIMER.end(t0, 2.0 * 2 * rows * C * 4 * C, "fp16x3/svtrmlp", 4.0 * (2 * rows * C + 2 * G * 4 * C * C))
    return y


def svtr_tail_fused(ctx_hl, x_res, rows, rows_per_group, G, C, wp_hl, sp, bp, drop, rows_per_drop, g2, b2, eps2, w1_hl, s1, b1, w2_hl, s2, b2m):
    """the second half of an SVTR stage-3 block in one launch (mrn_svtr_tail_x3_f32): x_res += drop * proj(ctx) IN PLACE, -> branch
    [rows, C] = fc2(GELU(fc1(LayerNorm2(x_res)))); w1_hl packed from fc1's weights with the input channels permuted by
    mlp_hidden_permutation(C), w2_hl from the hidden-permuted fc2 weights"""
    br = torch.empty(rows, C, device=x_res.device, dtype=torch.float32)
    t0 = CONV_TIMER.begin() if CONV_TIMER is not None else None
    call("mrn_svtr_tail_x3_f32", _p(ctx_hl), _p(x_res), _p(wp_hl), _p(sp), _p(bp), _p(drop), rows_per_drop, _p(g2), _p(b2), float(eps2),
         _p(w1_hl), _p(s1), _p(b1), _p(w2_hl), _p(s2), _p(b2m), _p(br), rows, rows_per_group, G, C, _stream())
    if t0 is not None:
        CONV_TIMER.end(t0, 2.0 * rows * C * C + 2.0 * 2 * rows * C * 4 * C, "fp16x3/svtrblock3", 4.0 * (4 * rows * C + G * 9 * C * C))
    return br


SVTR_FUSED_TAIL = os.environ.get("MRN_SVTR_TAIL", "1") == "1"           # stage 3: proj -> residual -> LayerNorm2 -> Mlp in one launch (A/B switch)
SVTR_FUSED_MLP256 = os.environ.get("MRN_SVTR_MLP256", "1") == "1"       # stage 3 (C = 256) on the fused Mlp kernel too (512-register form; A/B switch)
SVTR_FUSED_MIXER = os.environ.get("MRN_SVTR_MIXER", "fused") == "fused"   # LN1 -> qkv -> attention -> proj -> +residual -> LN2 in one kernel


def svtr_mixer_supported(N, C, imgs_per_group, mask):
    """shapes mrn_svtr_mixer_x3_f32 takes (SVTR stages 1 and 2 at 32 x 100 and 32 x 256 crops); mask: None or a 0 / -inf additive mask"""
    if not (SVTR_FUSED_MIXER and X3_PRODUCTS == 3):
        return False
    if not ((C == 64 and 96 < N <= 512) or (C == 128 and 48 < N <= 256 and (N > 128 or imgs_per_group % 2 == 0))):
        return False
    return mask is None or _mask_bits(mask) is not None


SVTR_LOCAL_COLUMNS = os.environ.get("MRN_SVTR_LOCAL_COLUMNS", "1") == "1"     # A/B: 0 = local mixers walk their tokens in memory (row-major) order


def _mask_bits_columns(mask, H, W):
    """_mask_bits of the same mask with queries and keys in COLUMN-major position order (position p = col * H + row <-> token row * W + col)"""
    got = getattr(mask, "_mrn_bits_cols", None)
    if got is None or got[0] != (mask._version, H, W):
        pos = torch.arange(H * W, device=mask.device)
        tok = (pos % H) * W + pos // H
        pm = mask.index_select(0, tok).index_select(1, tok).contiguous()
        got = ((mask._version, H, W), _mask_bits(pm))
        mask._mrn_bits_cols = got
    return got[1]


def svtr_mixer_fused(x, pending, drop_prev, g1, b1, eps1, wqkv_hl, sqkv, bqkv, mask, scale, wproj_hl, sproj, bproj, drop1, g2, b2, eps2,
                     imgs_per_group, hw=None):
    """x [imgs, N, C] -> (x_out [imgs, N, C], y_hl HL32 bytes of LayerNorm2(x_out)): the attention half of a mixing block
    (mrn_svtr_mixer_x3_f32); wproj_hl packed from the input-permuted proj weights (mlp_hidden_permutation(C)).  hw = (H, W) of the token
    map of a LOCAL mixer: the kernel walks the tokens column-major, so that whole key tiles fall outside the 7 x 11 window (svtr.py:110-128)
    and are skipped -- same result per token"""
    _chk(x, pending, drop_prev, g1, b1, bqkv, bproj, drop1, g2, b2)
    imgs, N, C = x.shape
    assert x.is_contiguous() and (pending is None or (pending.is_contiguous() and pending.numel() == x.numel()))
    x_out = torch.empty_like(x)
    y_hl = torch.empty(x.numel() * 4, device=x.device, dtype=torch.uint8)
    th = tw = 0
    bits = None
    if mask is not None:
        if SVTR_LOCAL_COLUMNS and hw is not None and hw[0] * hw[1] == N and N % 32 == 0 and 32 % hw[0] == 0:
            th, tw = int(hw[0]), int(hw[1])
            bits = _mask_bits_columns(mask, th, tw)
        else:
            bits = _mask_bits(mask)
    t0 = CONV_TIMER.begin() if CONV_TIMER is not None else None
    call("mrn_svtr_mixer_x3_f32", _p(x), _p(pending), _p(drop_prev), _p(g1), _p(b1), float(eps1), _p(wqkv_hl), _p(sqkv), _p(bqkv),
         _p(bits), float(scale), _p(wproj_hl), _p(sproj), _p(bproj), _p(drop1), _p(g2), _p(b2), float(eps2), _p(x_out), _p(y_hl),
         imgs, imgs_per_group, N, C, th, tw, _stream())
    if t0 is not None:
        rows = imgs * N
        CONV_TIMER.end(t0, 2.0 * rows * C * 4 * C + 4.0 * rows * N * C, "fp16x3/svtrmixer", 4.0 * rows * C * (4 if pending is not None else 3))
    return x_out, y_hl


SVTR_ATTN_BLOCK = os.environ.get("MRN_SVTR_ATTN_BLOCK", "1") == "1"      # stage 3 (C = 256): LayerNorm1 -> qkv -> attention in one kernel


def svtr_attention_block_supported(N, C, imgs_per_group, mask):
    """shapes mrn_svtr_attention_block_x3_f32 takes (SVTR stage 3: C = 256)"""
    if not (SVTR_FUSED_MIXER and SVTR_ATTN_BLOCK and X3_PRODUCTS == 3 and C == 256 and 16 < N <= 128 and imgs_per_group % (4 if N <= 64 else 2) == 0):
        return False
    return mask is None or _mask_bits(mask) is not None


def svtr_attention_block_fused(x, pending, drop_prev, g1, b1, eps1, wqkv_hl, sqkv, bqkv, mask, scale, imgs_per_group):
    """x [imgs, N, 256] -> (t = x + drop_prev * pending (x itself without a pending branch), HL32 bytes of attention(qkv(LayerNorm1(t)))):
    LayerNorm1 -> qkv -> attention of a mixing block in one kernel (mrn_svtr_attention_block_x3_f32); proj and the rest stay unfused"""
    _chk(x, pending, drop_prev, g1, b1, bqkv)
    imgs, N, C = x.shape
    assert x.is_contiguous() and (pending is None or (pending.is_contiguous() and pending.numel() == x.numel()))
    t = torch.empty_like(x) if pending is not None else None
    ctx_hl = torch.empty(x.numel() * 4, device=x.device, dtype=torch.uint8)
    bits = _mask_bits(mask) if mask is not None else None
    t0 = CONV_TIMER.begin() if CONV_TIMER is not None else None
    call("mrn_svtr_attention_block_x3_f32", _p(x), _p(pending), _p(drop_prev), _p(g1), _p(b1), float(eps1), _p(wqkv_hl), _p(sqkv), _p(bqkv),
         _p(bits), float(scale), _p(t), _p(ctx_hl), imgs, imgs_per_group, N, C, _stream())
    if t0 is not None:
        rows = imgs * N
        CONV_TIMER.end(t0, 2.0 * rows * C * 3 * C + 4.0 * rows * N * C, "fp16x3/svtrattn", 4.0 * rows * C * (4 if pending is not None else 2))
    return (t if t is not None else x), ctx_hl


def residual_scale_rows(x, branch, scale, rows_per_group, out=None):
    """x + scale[group] * branch on [rows, C] (contiguous)"""
    assert x.is_contiguous() and branch.is_contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    call("mrn_residual_scale_rows_f32", _p(x), _p(branch), _p(scale), _p(out), rows, C, rows_per_group, _stream())
    return out


# ---------------------------------------------------------------------------------------------------------
# LwF / EWC / weight alignment
# ---------------------------------------------------------------------------------------------------------
def kd_loss_fwd(xnew, xold, c0, c1, T):
    n2, o2 = rows2d(xnew), rows2d(xold)
    rows = n2.shape[0]
    lrows = torch.empty(rows, device=xnew.device, dtype=torch.float32)
    loss = torch.empty(1, device=xnew.device, dtype=torch.float32)
    call("mrn_kd_loss_fwd_f32", _p(n2), n2.stride(0), _p(o2), o2.stride(0), c0, c1, float(T), rows, _p(lrows), _p(loss), _stream())
    return loss


def kd_loss_bwd(xnew, xold, c0, c1, T, upstream):
    n2, o2 = rows2d(xnew), rows2d(xold)
    rows, C = n2.shape
    d = torch.empty_strided(xnew.shape, xnew.stride(), device=xnew.device, dtype=torch.float32)
    d2 = rows2d(d)
    call("mrn_kd_loss_bwd_f32", _p(n2), n2.stride(0), _p(o2), o2.stride(0), c0, c1, float(T), rows, _p(upstream), _p(d2),
         d2.stride(0), C, _stream())
    return d


def fisher_accumulate(fisher, grad):
    call("mrn_fisher_accumulate_f32", _p(fisher), _p(grad), fisher.numel(), _stream())


def fisher_finalize(fisher, iterations, fisher_max):
    call("mrn_fisher_finalize_f32", _p(fisher), fisher.numel(), 1.0 / iterations, float(fisher_max), _stream())


def ewc_penalty(fisher, p, mean):
    ws = torch.empty(2048, device=p.device, dtype=torch.float32)
    out = torch.empty(1, device=p.device, dtype=torch.float32)
    call("mrn_ewc_penalty_fwd_f32", _p(fisher), _p(p), _p(mean), p.numel(), _p(ws), _p(out), _stream())
    return out


def ewc_penalty_grad_(grad, fisher, p, mean, coef):
    call("mrn_ewc_penalty_bwd_f32", _p(fisher), _p(p), _p(mean), _p(grad), p.numel(), float(coef), _stream())


def weight_align_(weight, increment):
    """scale the last `increment` rows of weight [rows, C] by mean||old||/mean||new|| (in place) -> gamma (device scalar)"""
    rows, C = weight.shape
    ws = torch.empty(rows, device=weight.device, dtype=torch.float32)
    gamma = torch.empty(1, device=weight.device, dtype=torch.float32)
    call("mrn_weight_align_f32", _p(weight), weight.stride(0), rows, rows - increment, C, _p(ws), _p(gamma), _stream())
    return gamma
