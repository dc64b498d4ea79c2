"""Differentiable building blocks of the trainable (router) region, each a torch.autograd.Function whose forward
and backward are HIP kernel launches (mrn_amd.ops).  torch.autograd only sequences them.

Everything here works on the router-internal layout L2 = [B, P, I, C] ('b w h c'): expert i's contextual
features are written straight into slice [:, :, i, :], the reference's three rearranges
(modules/dm_router.py:58,61,63,65 and modules/model.py:402) disappear, and the only token-order dependence --
the spatial-gating weight over the (domain, patch) axis -- is absorbed by permuting that small weight.
"""
import weakref

import torch

from . import ops


# ---------------------------------------------------------------------------------------------------------
# GEMM helpers for backward passes
# ---------------------------------------------------------------------------------------------------------
def x3_eligible(x2, N, K):
    return (ops.ROUTER_GEMM_PRECISION == "fp16x3" and K % 32 == 0 and N >= 64 and x2.is_contiguous()
            and x2.shape[0] * max(N, K) * 4 < 2 ** 31)


def x3_linear(x2, weight, bias=None, residual=None, act=ops.ACT_NONE, sx=None, sw=None, want_sw=False, x_hl=None, w_pack=None,
              amax_ws=None):
    """y = act(x2 @ weight^T + bias (+ residual)) on the split-fp16 x3 MFMA path, both operands prescaled on the device
    (x2 [R,K] contiguous, weight [N,K]); 22-bit products, fp32 accumulation.  x_hl (with sx): the operand as its producer already
    split it; w_pack = (w_hl, sw): the weight operand packed ahead of use; amax_ws: max|y| folded into the epilogue"""
    R, K = x2.shape
    N = weight.shape[0]
    if sx is None:
        sx = ops.pow2_scale(x2)
    w_hl, sw = w_pack if w_pack is not None else ops.pack_weights_hl32([weight.detach().contiguous().view(N, 1, 1, K)], scale=sw)
    y, _ = ops.conv2d_x3(x_hl if x_hl is not None else ops.split_hl32(x2, sx), 1, False, R, 1, 1, K, w_hl, sw, N, (1, 1), bias=bias,
                         act=act, residual=residual, x_scale=sx, amax_ws=amax_ws)
    return (y.view(R, N), sw) if want_sw else y.view(R, N)


def frozen_linear(x, weight, bias=None, act=ops.ACT_NONE, residual=None):
    """Linear layer of a FROZEN module under no_grad (SVTR mixing blocks of the experts): split-fp16 x3 with the weight's
    HL32 pack cached per (storage, version); x [..., K] contiguous.  Falls back to the exact-fp32 GEMM when not eligible."""
    K, N = weight.shape[1], weight.shape[0]
    ok = (not torch.is_grad_enabled() and not weight.requires_grad and ops.CONV_PRECISION in ("auto", "fp16x3") and K % 32 == 0
          and N >= 64 and x.is_contiguous() and (residual is None or residual.is_contiguous()))
    if not ok:
        return ops.linear(x, weight, bias, act=act, residual=residual)
    key = (weight.data_ptr(), weight._version, N, K)
    got = getattr(weight, "_mrn_hl32", None)            # the pack lives (and dies) with the parameter it was built from
    if got is None or got[0] != key:
        got = (key, ops.pack_weights_hl32([weight.detach().contiguous().view(N, 1, 1, K)]))
        weight._mrn_hl32 = got
    w_hl, sw = got[1]
    R = x.numel() // K
    y, _ = ops.conv2d_x3(ops.split_hl32(x), 1, False, R, 1, 1, K, w_hl, sw, N, (1, 1), bias=bias, act=act,
                         residual=residual.view(R, N) if residual is not None else None, products=ops.X3_PRODUCTS)
    return y.view(*x.shape[:-1], N)


def linear_fwd(x2, weight, bias=None, residual=None, act=ops.ACT_NONE, sx=None):
    """router Linear: split-fp16 x3 when eligible, exact fp32 otherwise; sx: the operand's range scale when the caller already has it"""
    if x3_eligible(x2, weight.shape[0], weight.shape[1]) and (residual is None or residual.is_contiguous()):
        return x3_linear(x2, weight, bias, residual, act, sx=sx)
    return ops.linear(x2, weight, bias, act=act, residual=residual)


def operand_scale(x2):
    """the power-of-two range scale of a router GEMM operand, computed ONCE per tensor: the forward Linear, the weight gradient that
    reads the same activation in backward, and the two backward GEMMs that read one gradient (data + weight gradient) all used to run
    their own max|.| pass over it (26 such passes per loop-B step, 8 of them repeats); None off the split-fp16 path"""
    if ops.ROUTER_GEMM_PRECISION != "fp16x3" or not x2.is_contiguous():
        return None
    return ops.pow2_scale(x2)


def linear_dgrad(dy, weight, out=None, accumulate=False, sd=None, sw=None, wt_pack=None, amax_ws=None, dy_hl=None):
    """dx = dy @ weight ; dy rows [R, N], weight [N, K] -> [R, K]; sd / sw: pow2 scales of dy / weight when already known; wt_pack:
    the transposed weight operand (HL32 stack, scale) packed ahead of use"""
    dy2 = ops.rows2d(dy)
    R, N = dy2.shape
    K = weight.shape[1]
    if out is None and x3_eligible(dy2, K, N):
        if wt_pack is not None:
            return x3_linear(dy2, weight.detach().t(), sx=sd, w_pack=wt_pack, amax_ws=amax_ws, x_hl=dy_hl)
        return x3_linear(dy2, weight.detach().t().contiguous(), sx=sd, sw=sw)
    if out is None:
        out = torch.empty(R, K, device=dy.device, dtype=torch.float32)
        accumulate = False
    o2 = ops.rows2d(out)
    # C[r][k] = sum_n dy[r][n] * W[n][k]  ->  "W operand"[k][n] = weight[n][k]
    ops.gemm_raw(dy2, weight, o2, R, K, N, 1, (0, dy2.stride(0), 1), (0, 1, weight.stride(0)), (0, o2.stride(0), 1),
                 accumulate=accumulate)
    return out


def _pick_split(R, tiles, target=1024, min_rows=128):
    want = max(1, min(target // max(tiles, 1), R // min_rows))
    for s in range(want, 0, -1):
        if R % s == 0:
            return s
    return 1


def x3_wgrad(dy2, x2, sd=None, sx=None, out=None, bias_out=None, bias_accumulate=False):
    """dW = dy2^T @ x2 on the split-fp16 x3 path: both operands transposed-split (reduction axis = rows) with device
    prescales; split-K chunks are the grouped conv's groups, partial slabs reduced by a column-sum pass (which ADDS into `out`
    [N,K] when one is given and returns it).  bias_out [N]: the pass that transposes dy also leaves its column sums (the bias
    gradient) there, added when bias_accumulate."""
    R, N = dy2.shape
    K = x2.shape[1]
    blocks = R // 32
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    want = max(1, min(512 // tiles, blocks // 8))
    S = next(s for s in range(want, 0, -1) if blocks % s == 0)
    rps = R // S
    sd = ops.pow2_scale(dy2) if sd is None else sd
    sx = ops.pow2_scale(x2) if sx is None else sx
    a_hl = ops.split_hl32_t(dy2, S, sd, colsum_out=bias_out, accumulate=bias_accumulate)     # [S][N][rps/32][128]: "activation" rows = n
    w_hl = ops.split_hl32_t(x2, S, sx)                       # [S][K][rps/32][128]: "weight" rows = k
    part, _ = ops.conv2d_x3(a_hl, S, False, N, 1, 1, rps, w_hl, sx.view(1, 2).expand(S, 2).contiguous(), K, (1, 1), x_scale=sd)
    part = part.view(S, N * K)
    if S > 1 and out is not None:
        ops.colsum(part, out=out.view(-1), accumulate=True)
        return out
    return (ops.colsum(part) if S > 1 else part[0]).view(N, K)


def linear_wgrad(dy, x, sd=None, sx=None, out=None, bias_out=None, bias_accumulate=False):
    """dW[n][k] = sum_r dy[r][n] * x[r][k] with split-K over r (partials reduced by a column-sum pass; with `out` [N,K] -- a
    parameter's gradient buffer -- that pass ADDS into it and `out` is returned: no separate accumulation launch).
    bias_out [N]: also the bias gradient sum_r dy[r][n] (+= when bias_accumulate) -- out of the transposing pass over dy on the x3
    path, a column-sum pass otherwise."""
    dy2, x2 = ops.rows2d(dy), ops.rows2d(x)
    R, N = dy2.shape
    K = x2.shape[1]
    if (ops.ROUTER_GEMM_PRECISION == "fp16x3" and ops.ROUTER_WGRAD_X3 and R % 32 == 0 and R >= 4096 and K >= 64 and N >= 64 and N % 4 == 0 and K % 4 == 0
            and dy2.is_contiguous() and x2.is_contiguous()):
        if bias_out is not None and not ops.SPLIT_T_COLSUM:
            ops.colsum(dy2, out=bias_out, accumulate=bias_accumulate)
            bias_out = None
        return x3_wgrad(dy2, x2, sd, sx, out, bias_out, bias_accumulate)
    if bias_out is not None:
        ops.colsum(dy2, out=bias_out, accumulate=bias_accumulate)
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    S = _pick_split(R, tiles)
    Rc = R // S
    part = torch.empty(S, N, K, device=dy.device, dtype=torch.float32)
    ops.gemm_raw(dy2, x2, part, N, K, Rc, S, (Rc * dy2.stride(0), 1, dy2.stride(0)), (Rc * x2.stride(0), 1, x2.stride(0)),
                 (N * K, K, 1))
    if S == 1:
        return part[0]
    if out is not None:
        ops.colsum(part.view(S, N * K), out=out.view(-1), accumulate=True)
        return out
    return ops.colsum(part.view(S, N * K)).view(N, K)


def side_param_grads(params, compute, used, into=False, gen=None):
    """params: the Parameters a backward function owes gradients (None entries allowed); compute() -> their gradients in the same
    order.  Inside `with ops.direct_gradients()` (loss.backward() into the flat gradient, N = 1) the gradients are computed on the side
    stream and ADDED into each parameter's .grad there -- off the backward chain, filling the idle CUs next to the recurrent kernels
    (ops.side_stream_begin; the backward pass's final callback joins the streams) -- and None is returned for every parameter.
    `used`: the tensors compute() reads (kept alive for the side stream).  Otherwise: just compute().
    into=True: compute(outs) is handed the parameters' gradient buffers (None outside direct mode) and may ADD a gradient there itself
    (its last reduction pass accumulating), returning that same buffer in the gradient's place.
    gen: the forward generation ops.note_param_uses returned for this function's forward (ctx.use_gen)."""
    ok = (ops.WGRAD_SIDE_STREAM and ops.GRAD_DIRECT and not torch.is_grad_enabled() and
          all(p is None or (p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32) for p in params))
    if not ok:
        return list(compute([None] * len(params)) if into else compute())
    side = ops.side_stream_begin()
    with torch.cuda.stream(side):
        outs = [p.grad if p is not None else None for p in params]
        for p, g, o in zip(params, compute(outs) if into else compute(), outs):
            if p is not None and g is not None and g is not o:
                p.grad.add_(g.reshape(p.grad.shape))
    ops.side_stream_keep(used)
    ops.direct_done(params, gen)
    return [None] * len(params)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b over (strided) rows."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        ctx.use_gen = ops.note_param_uses(ctx.params, any(ctx.needs_input_grad))
        return ops.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = linear_dgrad(dy, weight).view(x.shape) if ctx.needs_input_grad[0] else None
        need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dw, db = side_param_grads([ctx.params[0] if need_w else None, ctx.params[1] if need_b else None],
                                  lambda o: (linear_wgrad(dy, x, out=o[0]) if need_w else None,
                                             ops.colsum(dy, out=o[1], accumulate=True) if need_b else None), (dy, x), into=True, gen=ctx.use_gen)
        return dx, dw, db


# ---------------------------------------------------------------------------------------------------------
# DM-Router block in L2 layout
# ---------------------------------------------------------------------------------------------------------
def token_permutation(P, I, device):
    """perm[p*I + d] = d*P + p : L2 token position -> reference '(d p)' token index; and its inverse."""
    idx = torch.arange(P * I, device=device, dtype=torch.int32)
    perm = (idx % I) * P + idx // I
    inv = torch.empty_like(perm)
    inv[perm.long()] = idx
    return perm, inv


class DMRouterFn(torch.autograd.Function):
    """DM_Router.forward (reference modules/dm_router.py:50-67) on x [B,P,I,C]; returns the same layout."""

    @staticmethod
    def forward(ctx, x, n_w, n_b, w1, b1, sn_w, sn_b, wsp, bsp, w2, b2, cn_w, cn_b, wch, bch, w3, b3):
        B, P, I, C = x.shape
        R, N = B * P * I, P * I
        x = x.contiguous()
        X = x.view(R, C)
        xn, mu1, rs1 = ops.layernorm_fwd(X, n_w, n_b)
        s_xn = operand_scale(xn)
        hpre = linear_fwd(xn, w1, b1, sx=s_xn)
        h = ops.ew_rows(ops.EW_GELU, hpre)
        u, v = h[:, :C], h[:, C:]
        vn, mu2, rs2 = ops.layernorm_fwd(v, sn_w, sn_b)
        perm, inv = token_permutation(P, I, x.device)
        ldw = (N + 3) // 4 * 4
        wsp_p = ops.gather2d(wsp, perm, perm, ld_out=ldw)                    # [N, N] view, row stride ldw
        bsp_p = ops.gather2d(bsp.view(1, N), None, perm).view(N)
        vp = torch.empty(R, C, device=x.device, dtype=torch.float32)
        # vp[b] = Wsp' . vn[b] + bsp'[:, None]
        ops.gemm_raw(wsp_p, vn, vp, N, C, N, B, (0, ldw, 1), (N * C, 1, C), (N * C, C, 1), bias=bsp_p, bias_axis=1)
        g = ops.ew_rows(ops.EW_MUL, u, vp)
        s_g = operand_scale(g)
        y = linear_fwd(g, w2, b2, residual=X, sx=s_g)
        y3 = y.view(B, P, I * C)
        zn, mu3, rs3 = ops.colnorm_fwd(y3, cn_w, cn_b)
        s_zn = operand_scale(zn.view(B * P, I * C))
        zp = linear_fwd(zn.view(B * P, I * C), wch, bch, sx=s_zn).view(R, C)
        z2 = ops.ew_rows(ops.EW_MUL, y, zp)
        s_z2 = operand_scale(z2)
        out = linear_fwd(z2, w3, b3, residual=X, sx=s_z2)
        ctx.scales = (s_xn, s_g, s_zn, s_z2)
        ctx.save_for_backward(X, n_w, w1, sn_w, w2, cn_w, wch, w3, mu1, rs1, xn, hpre, h, mu2, rs2, vn, wsp_p, vp, g, y,
                              mu3, rs3, zn, zp, z2, inv)
        ctx.dims = (B, P, I, C, ldw)
        return out.view(B, P, I, C)

    @staticmethod
    def backward(ctx, dout):
        (X, n_w, w1, sn_w, w2, cn_w, wch, w3, mu1, rs1, xn, hpre, h, mu2, rs2, vn, wsp_p, vp, g, y, mu3, rs3, zn, zp, z2,
         inv) = ctx.saved_tensors
        B, P, I, C, ldw = ctx.dims
        R, N = B * P * I, P * I
        need_x = ctx.needs_input_grad[0]
        dout = dout.contiguous().view(R, C)
        dev = dout.device
        # out = z2 W3^T + b3 + X
        s_xn, s_g, s_zn, s_z2 = ctx.scales
        s_do = operand_scale(dout)
        dw3, db3 = linear_wgrad(dout, z2, sd=s_do, sx=s_z2), ops.colsum(dout)
        dz2 = linear_dgrad(dout, w3, sd=s_do)
        # z2 = y * zp
        dy = ops.ew_rows(ops.EW_MUL, dz2, zp)
        dzp = ops.ew_rows(ops.EW_MUL, dz2, y)
        # zp = zn Wch^T + bch over rows [B*P, I*C]
        dzp2, zn2 = dzp.view(B * P, I * C), zn.view(B * P, I * C)
        s_dzp = operand_scale(dzp2)
        dwch, dbch = linear_wgrad(dzp2, zn2, sd=s_dzp, sx=s_zn), ops.colsum(dzp2)
        dzn = linear_dgrad(dzp2, wch, sd=s_dzp)
        # zn = LayerNorm_P(y)
        _, dcn_w, dcn_b = ops.colnorm_bwd(dzn.view(B, P, I * C), y.view(B, P, I * C), cn_w, mu3, rs3,
                                          dx=dy.view(B, P, I * C), accumulate=True)
        # y = g W2^T + b2 + X
        s_dy = operand_scale(dy)
        dw2, db2 = linear_wgrad(dy, g, sd=s_dy, sx=s_g), ops.colsum(dy)
        dg = linear_dgrad(dy, w2, sd=s_dy)
        # g = u * vp
        u, v = h[:, :C], h[:, C:]
        dh = torch.empty(R, 2 * C, device=dev, dtype=torch.float32)
        ops.ew_rows(ops.EW_MUL, dg, vp, out=dh[:, :C])                      # du
        dvp = ops.ew_rows(ops.EW_MUL, dg, u)
        # vp[b] = Wsp' vn[b] + bsp'
        if ops.ROUTER_GEMM_PRECISION == "fp16x3" and ops.ROUTER_WGRAD_X3 and ops.ROUTER_TOKEN_WGRAD_X3 and C % 32 == 0 and N >= 64:
            # per sample dW_b[n][m] = sum_c dvp[b][n][c] vn[b][m][c]: the reduction runs over the CONTIGUOUS axis of both operands, so the
            # plain HL32 splits are the operands of ONE grouped split-fp16 x3 GEMM with the samples as groups (the exact-fp32 MFMA ran
            # these 256 390 x 390 x 256 products at 37 TF: 0.53 ms of a router phase of 6-7 ms)
            sd, sx = ops.pow2_scale(dvp), ops.pow2_scale(vn)
            part, _ = ops.conv2d_x3(ops.split_hl32(dvp, sd), B, False, N, 1, 1, C, ops.split_hl32(vn, sx),
                                    sx.view(1, 2).expand(B, 2).contiguous(), N, (1, 1), x_scale=sd)
            part = part.view(B, N, N)
        else:
            part = torch.empty(B, N, N, device=dev, dtype=torch.float32)
            ops.gemm_raw(dvp, vn, part, N, N, C, B, (N * C, C, 1), (N * C, C, 1), (N * N, N, 1))
        dwsp_p = ops.colsum(part.view(B, N * N)).view(N, N) if B > 1 else part[0]
        dwsp = ops.gather2d(dwsp_p, inv, inv)
        ones = torch.ones(1, C, device=dev, dtype=torch.float32)
        rowsum = ops.linear(dvp, ones)                                        # [R,1] = sum_c dvp
        dbsp_p = ops.colsum(rowsum.view(B, N)) if B > 1 else rowsum.view(N)
        dbsp = ops.gather2d(dbsp_p.view(1, N), None, inv).view(N)
        dvn = torch.empty(R, C, device=dev, dtype=torch.float32)
        # dvn[b] = Wsp'^T dvp[b]
        ops.gemm_raw(wsp_p, dvp, dvn, N, C, N, B, (0, 1, ldw), (N * C, 1, C), (N * C, C, 1))
        # vn = LayerNorm(v)
        _, dsn_w, dsn_b = ops.layernorm_bwd(dvn, v, sn_w, mu2, rs2, dx=dh[:, C:])
        # h = gelu(hpre)
        dhpre = ops.ew_rows(ops.EW_GELU_BWD, hpre, dh)
        s_dh = operand_scale(dhpre)
        dw1, db1 = linear_wgrad(dhpre, xn, sd=s_dh, sx=s_xn), ops.colsum(dhpre)
        dxn = linear_dgrad(dhpre, w1, sd=s_dh)
        # xn = LayerNorm(X)
        dx_ln, dn_w, dn_b = ops.layernorm_bwd(dxn, X, n_w, mu1, rs1)
        dx = None
        if need_x:
            dx = ops.ew_rows(ops.EW_ADD, dx_ln, dout)
            dx = ops.ew_rows(ops.EW_ADD, dx, dy, out=dx).view(B, P, I, C)
        return (dx, dn_w, dn_b, dw1, db1, dsn_w, dsn_b, dwsp, dbsp, dw2, db2, dcn_w, dcn_b, dwch, dbch, dw3, db3)


class GateTailFn(torch.autograd.Function):
    """route Linear(P->1) over the patch axis + squeeze + softmax(beta * s)  (modules/model.py:405-406,495-496)"""

    @staticmethod
    def forward(ctx, r, w_route, b_route, beta):
        r = r.contiguous()
        _, w = ops.gate_tail_fwd(r, w_route.view(-1), b_route, beta)
        ctx.save_for_backward(r, w_route, w)
        ctx.beta = beta
        return w

    @staticmethod
    def backward(ctx, dw):
        r, w_route, w = ctx.saved_tensors
        dr, dW, db = ops.gate_tail_bwd(w, dw, r, w_route.view(-1), ctx.beta)
        return dr, dW.view_as(w_route), db, None


class FaninFn(torch.autograd.Function):
    """out = sum_i w[:, i] * pad_ones(L_i)   (modules/model.py:410-423); experts' logits carry no gradient here."""

    @staticmethod
    def forward(ctx, w, *logits):
        ctx.logits = logits
        ctx.save_for_backward(w)
        return ops.fanin_fwd(list(logits), w.contiguous())

    @staticmethod
    def backward(ctx, dout):
        if any(ctx.needs_input_grad[1:]):
            raise NotImplementedError("gradient of the fan-in with respect to expert logits is not implemented "
                                      "(experts are frozen in MRN's router phase)")
        if dout.stride(2) != 1 or dout.stride(1) % 4 != 0 or dout.stride(0) != dout.shape[1] * dout.stride(1):
            d = ops.padded_rows(*dout.shape, dout.device)
            d.copy_(dout)
            dout = d
        dw = ops.fanin_bwd(list(ctx.logits), dout)
        return (dw,) + (None,) * len(ctx.logits)


class CrossEntropyFn(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=...) on (strided) rows of logits."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        loss, c = ops.ce_loss_fwd(logits, target, ignore_index)
        ctx.c = c
        ctx.like = logits
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        d = ops.ce_loss_bwd(ctx.c, g.contiguous().view(1), ctx.like)
        return d, None, None


class CTCLossFn(torch.autograd.Function):
    """preds.log_softmax(2).permute(1,0,2) -> CTCLoss(blank=0, mean, zero_infinity=True) with input lengths = T."""

    @staticmethod
    def forward(ctx, logits, targets, target_len):
        loss, c = ops.ctc_loss_fwd(logits, targets, target_len, 0)
        ctx.c = c
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        return ops.ctc_loss_bwd(ctx.c, g.contiguous().view(1)), None, None


def cross_entropy(logits, target, ignore_index=-100):
    return CrossEntropyFn.apply(logits, target, ignore_index)


def ctc_loss(logits, targets, target_len):
    return CTCLossFn.apply(logits, targets, target_len)


# ---------------------------------------------------------------------------------------------------------
# Differentiable expert stages (loop A: `loss.backward()` through the newest expert, il_modules/mrn.py:260-261)
# ---------------------------------------------------------------------------------------------------------
def needs_grad(module, *tensors):
    """True when autograd must record this stage (a trainable parameter or an input that carries gradient)."""
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


class ConvBlockFn(torch.autograd.Function):
    """NHWC conv (+bias) [-> BatchNorm2d(train)] [-> +residual] [-> ReLU]; all math in HIP kernels.
    forward(x, weight[O,I,kh,kw], bias, gamma, beta, residual, cfg)  with cfg = (conv_module, bn_module, relu, precision)"""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, residual, cfg):
        from .modules._nn import packed_weight, _pair
        conv, bn, relu, precision = cfg
        w = packed_weight(conv)
        stride, padding = _pair(conv.stride), _pair(conv.padding)
        ctx.geom = (stride, padding, relu, precision, bn is not None, residual is not None)
        ctx.wpacked = w
        ctx.conv = conv
        ctx.bn = bn
        ctx.use_gen = ops.note_param_uses((conv.weight, conv.bias) + ((bn.weight, bn.bias) if bn is not None else ()), any(ctx.needs_input_grad))
        if id(conv) not in ops.TRAINED_CONVS:
            ops.TRAINED_CONVS[id(conv)] = (weakref.ref(conv), stride, padding)
        # one max|x| pass serves the forward conv and the weight gradient (both split x with the same power-of-two scale)
        x3s = precision == "fp16x3s" and x.shape[-1] % 32 == 0
        # (the producer of x may have folded max|x| into its own pass: ops.scale_shift_act(range_target=...) of the previous block)
        sx = None
        if x3s:
            sx = ops.cached_scale(x) if x.is_contiguous() else None
            if sx is None:
                sx = ops.pow2_scale(x.contiguous(), ops.TRAIN_OPERAND_PEAK)
        ctx.x_scale = sx
        if bn is None:
            y, _ = ops.conv2d_nhwc(x, w, bias, stride, padding, act=ops.ACT_RELU if relu else ops.ACT_NONE, precision=precision,
                                   x_scale=sx)
            ctx.save_for_backward(x, y if relu else None, None, None, None, None)
            return y
        if not bn.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not implemented (experts train in train mode)")
        y, stats = ops.conv2d_nhwc(x, w, bias, stride, padding, act=ops.ACT_NONE, want_stats=True, precision=precision, x_scale=sx)
        count = y.shape[0] * y.shape[1] * y.shape[2]
        mom = 0.1 if bn.momentum is None else bn.momentum
        scale, shift, mean, invstd = ops.bn_finalize(stats, bn.num_features, count, gamma, beta, bn.running_mean,
                                                     bn.running_var, mom, bn.eps, save=True)
        ops.count_batch(bn)
        z = torch.empty_like(y)
        # the backward pass needs only the SIGN of z (ReLU mask): 4 bits per 4 channels written by this pass, read there instead of z
        zmask = torch.empty(z.numel() // 4, device=z.device, dtype=torch.uint8) if relu else None
        # z is (almost always) the operand of the next trained convolution: its range scale comes out of this pass
        ops.scale_shift_act(y, scale, shift, relu=relu, residual=residual, out=z, pos_mask=zmask,
                            range_target=ops.TRAIN_OPERAND_PEAK if (precision == "fp16x3s" and z.shape[-1] % 32 == 0) else None)
        ctx.save_for_backward(x, zmask, y, mean, invstd, gamma)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, z, y, mean, invstd, gamma = ctx.saved_tensors
        stride, padding, relu, precision, has_bn, has_res = ctx.geom
        w = ctx.wpacked
        dz = dz.contiguous()
        dgamma = dbeta = dres = dbias = None
        sd = None
        direct = ops.WGRAD_SIDE_STREAM and ops.GRAD_DIRECT and not torch.is_grad_enabled()
        if has_bn:
            # (z is the bit mask here.)  In direct mode the BatchNorm weight / bias gradients are added to the flat gradient by the
            # finalize launch of the statistics (main stream): no reduction launches, no accumulation launches
            bw, bb = ctx.bn.weight, ctx.bn.bias
            acc = None
            if (direct and ctx.needs_input_grad[3] and ctx.needs_input_grad[4]
                    and all(p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in (bw, bb))):
                acc = (bw.grad, bb.grad)
            if precision == "fp16x3s":      # max|dy| folded into the apply pass: the range scale shared by the data and weight gradients
                dy, dgamma, dbeta, dres, sd = ops.bn_bwd(dz, None, y, mean, invstd, gamma, relu, want_dres=has_res,
                                                         range_target=ops.TRAIN_OPERAND_PEAK, zmask=z, grad_acc=acc)
            else:
                dy, dgamma, dbeta, dres = ops.bn_bwd(dz, None, y, mean, invstd, gamma, relu, want_dres=has_res, zmask=z, grad_acc=acc)
            if acc is not None:
                ops.direct_done((bw, bb), ctx.use_gen)
            if ctx.needs_input_grad[2]:       # a conv bias in front of train-mode BatchNorm (SVTR PatchEmbed) cancels in the
                dbias = torch.zeros(dy.shape[-1], device=dy.device, dtype=torch.float32)     # mean: its gradient is exactly 0
        else:
            dy = ops.ew_rows(ops.EW_RELU_BWD, z, dz) if relu else dz
            if ctx.needs_input_grad[2]:
                dbias = ops.colsum(dy.view(-1, dy.shape[-1]))
        dx = None
        dy = dy.contiguous()
        if sd is None and precision == "fp16x3s":
            sd = ops.pow2_scale(dy, ops.TRAIN_OPERAND_PEAK)                                   # shared by the data and weight gradients

        def weight_gradient(acc=None):                  # -> [O,kh,kw,I], or None when it was ADDED into acc ([O,I,kh,kw])
            kh, kw = w.shape[1], w.shape[2]
            if precision == "fp16x3s" and x.shape[-1] % 4 == 0 and ops.TRAIN_WGRAD_X3:
                if ops.wgrad_wino_supported(dy, x, (kh, kw), stride, padding):
                    return ops.conv2d_wgrad_x3_wino(dy, x, dy_scale=sd, x_scale=ctx.x_scale, acc_oihw=acc)
                if ops.WGRAD_WINDOWS and ops.wgrad_windows_supported(dy, x, (kh, kw), stride, padding):
                    return ops.conv2d_wgrad_x3_windows(dy, x, dy_scale=sd, x_scale=ctx.x_scale)
                return ops.conv2d_wgrad_x3(dy, x, (kh, kw), stride, padding, dy_scale=sd, x_scale=ctx.x_scale)
            return ops.conv2d_wgrad(dy, x, (kh, kw), stride, padding)
        dw = None
        if ctx.needs_input_grad[1]:
            wgrad = ctx.conv.weight.grad
            if (ops.WGRAD_SIDE_STREAM and ops.GRAD_DIRECT and wgrad is not None and wgrad.is_contiguous() and wgrad.dtype == torch.float32
                    and torch.is_grad_enabled() is False):
                # off the critical path: second stream, accumulated straight into the parameter's gradient (ops.side_stream_begin)
                side = ops.side_stream_begin()
                with torch.cuda.stream(side):
                    g = weight_gradient(wgrad)
                    if g is not None:
                        ops.unpack_conv_weight(g, out=wgrad, accumulate=True)
                ops.side_stream_keep((dy, x, sd, ctx.x_scale))
                ops.direct_done((ctx.conv.weight,), ctx.use_gen)
            else:
                dw = ops.unpack_conv_weight(weight_gradient())
        # the small per-channel gradients (conv bias, BatchNorm weight / bias) too: handed to autograd they cost two or three 5-us
        # accumulation launches per block ON the backward chain (72 per TRBA step)
        small = [(ctx.conv.bias, dbias), (ctx.bn.weight if ctx.bn is not None else None, dgamma),
                 (ctx.bn.bias if ctx.bn is not None else None, dbeta)]
        if (ops.WGRAD_SIDE_STREAM and ops.GRAD_DIRECT and not torch.is_grad_enabled()
                and all(g is None or (p is not None and p.grad is not None and p.grad.dtype == torch.float32) for p, g in small)
                and any(g is not None for _, g in small)):
            side = ops.side_stream_begin()
            with torch.cuda.stream(side):
                for p, g in small:
                    if g is not None:
                        p.grad.add_(g.reshape(p.grad.shape))
            ops.side_stream_keep([g for _, g in small])
            ops.direct_done([p for p, g in small if g is not None], ctx.use_gen)
            dbias = dgamma = dbeta = None
        if ctx.needs_input_grad[0]:
            wt = ops.trained_dgrad_weight(w.ohwi)
            dx = ops.conv2d_dgrad(dy, wt, (x.shape[1], x.shape[2]), stride, padding, precision=precision, dy_scale=sd)
        return dx, dw, dbias, dgamma, dbeta, dres, None


class BatchNorm2dFn(torch.autograd.Function):
    """nn.BatchNorm2d on an NHWC tensor that is not a conv output (the gated recurrent conv layer of the RCNN extractor normalises
    the same conv result under several BatchNorms, and a gated product: modules/feature_extraction.py:146-161).  Train mode:
    batch statistics + running-statistics update; backward through the statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, relu=False):
        x = x.contiguous()
        C = x.shape[-1]
        count = x.numel() // C
        mom = 0.1 if bn.momentum is None else bn.momentum
        scale, shift, mean, invstd = ops.bn_finalize(ops.bn_stats(x), C, count, gamma, beta, bn.running_mean, bn.running_var,
                                                     mom, bn.eps, save=True)
        ops.count_batch(bn)
        y = torch.empty_like(x)
        ops.scale_shift_act(x, scale, shift, relu=relu, out=y)
        ctx.save_for_backward(x, y, mean, invstd, gamma)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd, gamma = ctx.saved_tensors
        dx, dgamma, dbeta, _ = ops.bn_bwd(dy.contiguous(), y, x, mean, invstd, gamma, ctx.relu)
        return dx, dgamma, dbeta, None, None


def batch_norm_nhwc(x, bn, relu=False):
    """BatchNorm2d module `bn` applied to an NHWC tensor: autograd Function in train mode with gradients, plain kernels otherwise"""
    if bn.training:
        if needs_grad(bn, x):
            return BatchNorm2dFn.apply(x, bn.weight, bn.bias, bn, relu)
        C = x.shape[-1]
        mom = 0.1 if bn.momentum is None else bn.momentum
        scale, shift, _, _ = ops.bn_finalize(ops.bn_stats(x.contiguous()), C, x.numel() // C, bn.weight, bn.bias, bn.running_mean,
                                             bn.running_var, mom, bn.eps)
        ops.count_batch(bn)
    else:
        if needs_grad(bn, x):
            raise NotImplementedError("backward through eval-mode BatchNorm is not implemented (experts train in train mode)")
        scale, shift = ops.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    return ops.scale_shift_act(x.contiguous(), scale, shift, relu=relu, out=torch.empty_like(x))


class SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.ew_rows(ops.EW_SIGMOID, x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.ew_rows(ops.EW_SIGMOID_BWD, y, dy.contiguous())


class MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.ew_rows(ops.EW_MUL, a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = dy.contiguous()
        return ops.ew_rows(ops.EW_MUL, dy, b), ops.ew_rows(ops.EW_MUL, dy, a)


class AddReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        y = ops.ew_rows(ops.EW_ADD_RELU, a.contiguous(), b.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        d = ops.ew_rows(ops.EW_RELU_BWD, y, dy.contiguous())
        return d, d


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.ew_rows(ops.EW_ADD, a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding):
        ctx.save_for_backward(x)
        ctx.geom = (kernel, stride, padding)
        return ops.maxpool_nhwc(x, kernel, stride, padding)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool_bwd(dy, x, *ctx.geom), None, None, None


def _pack_lstm_h(w_f, w_r):
    packs = [ops.pack_fragment_major_h(w.detach()) for w in (w_f, w_r)]
    return torch.stack([p_[0] for p_ in packs]).contiguous(), torch.cat([p_[1] for p_ in packs]).contiguous()


def _pack_lstm_hT(w_f, w_r):
    packs = [ops.pack_fragment_major_h(w.detach().t().contiguous()) for w in (w_f, w_r)]
    return torch.stack([p_[0] for p_ in packs]).contiguous(), torch.cat([p_[1] for p_ in packs]).contiguous()


class BiLSTMFn(torch.autograd.Function):
    """nn.LSTM(bidirectional=True, batch_first=True) forward + backward through time on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        H = w_hh_f.shape[1]
        # (the repacks of a step's weights come from ops.train_pack: issued on the side stream ahead of use from the second step on)
        w_ih, b_ih, b_hh = ops.train_pack("lstm_cat", (w_ih_f, w_ih_r, b_ih_f, b_ih_r, b_hh_f, b_hh_r),
                                          lambda a, b, c, d, e, f: (torch.cat([a.detach(), b.detach()], 0), torch.cat([c.detach(), d.detach()], 0),
                                                                    torch.cat([e.detach(), f.detach()], 0)))
        x = x.contiguous()
        x2 = x.view(-1, x.shape[-1])
        if ops.TRAIN_LSTM_X3 and x3_eligible(x2, w_ih.shape[0], x2.shape[1]):
            # the input projection of both directions on the split-fp16 x3 GEMM, like its data gradient below and like the frozen experts'
            # (exact fp32: 0.51 ms for 512 -> 2048 over 16640 rows; x3: 0.14 ms)
            w_pack = ops.train_pack("lstm_ih_x3", (w_ih_f, w_ih_r), lambda a, b: _pack_linear(torch.cat([a.detach(), b.detach()], 0)))
            xproj = x3_linear(x2, w_ih, b_ih, w_pack=w_pack).view(*x.shape[:-1], w_ih.shape[0])
        else:
            xproj = ops.linear(x, w_ih, b_ih)
        if ops.TRAIN_LSTM_X3 and H == 256:
            # recurrent product as split-fp16 x3 (the trained convolutions' arithmetic): half the time per step of the exact-fp32 MFMA
            w_h, w_inv = ops.train_pack("lstm_fwd_x3", (w_hh_f, w_hh_r), _pack_lstm_h)
            out, gates, cseq = ops.lstm_layer_x3_save(xproj, w_h, w_inv, b_hh, H, 2)
        else:
            w_hh = torch.stack([ops.pack_fragment_major(w_hh_f), ops.pack_fragment_major(w_hh_r)], 0)
            out, gates, cseq = ops.lstm_layer(xproj, w_hh, b_hh, H, 2, save=True)
        ctx.save_for_backward(x, w_ih, w_hh_f, w_hh_r, out, gates, cseq)
        ctx.H = H
        ctx.params = (w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r)
        ctx.use_gen = ops.note_param_uses(ctx.params, any(ctx.needs_input_grad))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w_ih, w_hh_f, w_hh_r, out, gates, cseq = ctx.saved_tensors
        H = ctx.H
        B, T, _ = out.shape
        if ops.TRAIN_LSTM_X3 and H == 256:
            w_hT, w_invT = ops.train_pack("lstm_bwd_x3", (ctx.params[1], ctx.params[5]), _pack_lstm_hT)
            dg = ops.lstm_layer_bwd_x3(dout, gates, cseq, w_hT, w_invT, H, 2)                                 # [B,T,2,4H]
        else:
            w_hhT = torch.stack([ops.pack_fragment_major(w_hh_f.t().contiguous()), ops.pack_fragment_major(w_hh_r.t().contiguous())], 0)
            dg = ops.lstm_layer_bwd(dout, gates, cseq, w_hhT, H, 2)          # [B,T,2,4H]
        dg2 = dg.view(B * T, 8 * H)
        dx = linear_dgrad(dg2, w_ih).view(x.shape) if ctx.needs_input_grad[0] else None

        def param_grads():
            dw_ih = linear_wgrad(dg2, x.view(B * T, -1))                       # [8H, in]
            db = ops.colsum(dg2)                                               # d b_ih = d b_hh
            # h_{t-1} in each direction's own time order (data movement only)
            hprev = torch.zeros(B, T, 2, H, device=out.device, dtype=torch.float32)
            hprev[:, 1:, 0, :] = out[:, :-1, :H]
            hprev[:, :-1, 1, :] = out[:, 1:, H:]
            dw_hh_f = linear_wgrad(dg[:, :, 0, :], hprev[:, :, 0, :])
            dw_hh_r = linear_wgrad(dg[:, :, 1, :], hprev[:, :, 1, :])
            return (dw_ih[:4 * H], dw_hh_f, db[:4 * H], db[:4 * H], dw_ih[4 * H:], dw_hh_r, db[4 * H:], db[4 * H:])
        # (the parameter gradients hang off the chain: with direct gradients they overlap the next recurrent kernel's idle CUs)
        return (dx, *side_param_grads(list(ctx.params), param_grads, (dg, x, out), gen=ctx.use_gen))


class LinearReluFn(torch.autograd.Function):
    """relu(x W^T + b)"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = ops.linear(x, weight, bias, act=ops.ACT_RELU)
        ctx.save_for_backward(x, weight, y)
        ctx.params = (weight, bias)
        ctx.use_gen = ops.note_param_uses(ctx.params, any(ctx.needs_input_grad))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        g = ops.ew_rows(ops.EW_RELU_BWD, y, dy.contiguous())
        dx = linear_dgrad(g, weight).view(x.shape) if ctx.needs_input_grad[0] else None
        dw, db = side_param_grads(list(ctx.params), lambda: (linear_wgrad(g, x), ops.colsum(g)), (g, x), gen=ctx.use_gen)
        return dx, dw, db


class AvgPoolFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) on NHWC: [B,H,W,C] -> [B,C]"""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return ops.avgpool_nhwc(x)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        return ops.avgpool_bwd(dy, H * W).view(B, H, W, C)


class TPSSampleFn(torch.autograd.Function):
    """grid generation + bilinear sampling; gradient flows to the fiducials C' only (the image is an input)."""

    @staticmethod
    def forward(ctx, img_nhwc, cprime, inv_delta_c, p_hat, out_hw):
        ctx.save_for_backward(img_nhwc, cprime, inv_delta_c, p_hat)
        return ops.tps_grid_sample(img_nhwc, cprime, inv_delta_c, p_hat, out_hw)

    @staticmethod
    def backward(ctx, dout):
        img, cprime, inv_delta_c, p_hat = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient of the TPS sampler with respect to the image is not implemented")
        return None, ops.tps_grid_sample_bwd(img, cprime, inv_delta_c, p_hat, dout), None, None, None


def _pack_decoder_x3(h2h_w, w_ih, w_hh, emb_w):
    """(the context width D of an AttentionCell is rnn.weight_ih's width minus the embedding width)"""
    return ops.pack_decoder_x3(h2h_w, w_ih, w_hh, w_ih.shape[1] - emb_w.shape[1])


def _pack_decoder_T(h2h_w, w_ih, w_hh, emb_w):
    D = w_ih.shape[1] - emb_w.shape[1]
    return (ops.pack_fragment_major(h2h_w.detach().t().contiguous()), ops.pack_fragment_major(w_ih.detach()[:, :D].t().contiguous()),
            ops.pack_fragment_major(w_hh.detach().t().contiguous()))


def _pack_decoder_T_x3(h2h_w, w_ih, w_hh, emb_w):
    D = w_ih.shape[1] - emb_w.shape[1]
    a = ops.pack_fragment_major_h(h2h_w.detach().t().contiguous())
    b = ops.pack_fragment_major_h(w_ih.detach()[:, :D].t().contiguous())
    c = ops.pack_fragment_major_h(w_hh.detach().t().contiguous())
    return a[0], b[0], c[0], torch.cat([a[1], b[1], c[1]]).contiguous()


class AttnDecoderFn(torch.autograd.Function):
    """Teacher-forced Attention.forward (modules/prediction.py:58-68) with its full backward."""

    @staticmethod
    def forward(ctx, batch_H, text, i2h_w, h2h_w, h2h_b, score_w, w_ih, w_hh, b_ih, b_hh, emb_w, gen_w, gen_b, S):
        Hd = h2h_w.shape[0]
        D = i2h_w.shape[1]
        num_class = emb_w.shape[0]
        batch_H = batch_H.contiguous()
        Hproj = ops.linear(batch_H, i2h_w)
        tok = text[:, :S]
        emb = ops.embed_gather(tok, emb_w, num_class)
        eproj = ops.linear(emb, w_ih[:, D:], b_ih)
        if ops.DECODER_X3 and D % 32 == 0 and Hd == 256:
            a, b_, c_, w_inv = ops.train_pack("dec_fwd_x3", (h2h_w, w_ih, w_hh, emb_w), _pack_decoder_x3)
            hid, saves = ops.attn_decoder_train(batch_H, Hproj, eproj, a, h2h_b, score_w, b_, c_, b_hh, Hd, w_inv=w_inv)
        else:
            hid, saves = ops.attn_decoder_train(batch_H, Hproj, eproj, ops.pack_fragment_major(h2h_w), h2h_b, score_w,
                                                ops.pack_fragment_major(w_ih[:, :D]), ops.pack_fragment_major(w_hh), b_hh, Hd)
        probs = ops.linear(hid, gen_w, gen_b)
        ctx.save_for_backward(batch_H, Hproj, emb, hid, i2h_w, h2h_w, score_w, w_ih, w_hh, gen_w, tok, *saves)
        ctx.dims = (Hd, D, num_class, S)
        ctx.params = (i2h_w, h2h_w, h2h_b, score_w, w_ih, w_hh, b_ih, b_hh, emb_w, gen_w, gen_b)
        ctx.use_gen = ops.note_param_uses(ctx.params, any(ctx.needs_input_grad))
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        (batch_H, Hproj, emb, hid, i2h_w, h2h_w, score_w, w_ih, w_hh, gen_w, tok, alpha, gates, cseq, cx, hp) = ctx.saved_tensors
        Hd, D, num_class, S = ctx.dims
        B, T, _ = batch_H.shape
        dprobs = dprobs.contiguous()
        dhid = linear_dgrad(dprobs, gen_w).view(B, S, Hd)
        if ops.DECODER_X3 and Hd == 256:
            pT = ops.train_pack("dec_bwd_x3", (ctx.params[1], ctx.params[4], ctx.params[5], ctx.params[8]), _pack_decoder_T_x3)
            dgates, dhp, dHb, dHproj, dws = ops.attn_decoder_bwd(batch_H, Hproj, (alpha, gates, cseq, cx, hp), dhid, score_w, pT[0], pT[1], pT[2],
                                                                 Hd, w_inv=pT[3])
        else:
            pT = ops.train_pack("dec_bwd", (ctx.params[1], ctx.params[4], ctx.params[5], ctx.params[8]), _pack_decoder_T)
            dgates, dhp, dHb, dHproj, dws = ops.attn_decoder_bwd(batch_H, Hproj, (alpha, gates, cseq, cx, hp), dhid, score_w, pT[0], pT[1], pT[2], Hd)

        def param_grads():            # order of ctx.params: i2h_w, h2h_w, h2h_b, score_w, w_ih, w_hh, b_ih, b_hh, emb_w, gen_w, gen_b
            dgen_w, dgen_b = linear_wgrad(dprobs, hid), ops.colsum(dprobs)
            hprev = torch.zeros(B, S, Hd, device=hid.device, dtype=torch.float32)
            hprev[:, 1:] = hid[:, :-1]                                  # h_{s-1} (data movement)
            dg2 = dgates.view(B * S, 4 * Hd)
            dw_ih = torch.empty_like(w_ih)
            dw_ih[:, :D] = linear_wgrad(dg2, cx)
            dw_ih[:, D:] = linear_wgrad(dg2, emb)
            dw_hh = linear_wgrad(dg2, hprev)
            db = ops.colsum(dg2)
            dh2h_w, dh2h_b = linear_wgrad(dhp, hprev), ops.colsum(dhp.view(B * S, Hd))
            demb = linear_dgrad(dg2, w_ih[:, D:])
            demb_w = ops.embed_scatter_add(tok, demb.view(B, S, -1), num_class)
            di2h_w = linear_wgrad(dHproj, batch_H)
            return (di2h_w, dh2h_w, dh2h_b, dws.view_as(score_w), dw_ih, dw_hh, db, db, demb_w, dgen_w, dgen_b)
        grads = side_param_grads(list(ctx.params), param_grads, (dprobs, hid, dgates, cx, emb, dhp, dHproj, batch_H, dws, w_ih, tok), gen=ctx.use_gen)
        dH = None
        if ctx.needs_input_grad[0]:
            dH = linear_dgrad(dHproj, i2h_w, out=dHb.view(B * T, D), accumulate=True).view(B, T, D)
        return (dH, None, *grads, None)


class KDLossFn(torch.autograd.Function):
    """_KD_loss of LwF/WA (reference il_modules/lwf.py:111-114) on the class slice [c0, c1); the teacher carries no grad."""

    @staticmethod
    def forward(ctx, pred, soft, c0, c1, T):
        ctx.save_for_backward(pred, soft)
        ctx.args = (c0, c1, T)
        return ops.kd_loss_fwd(pred, soft, c0, c1, T).view(())

    @staticmethod
    def backward(ctx, g):
        pred, soft = ctx.saved_tensors
        c0, c1, T = ctx.args
        return ops.kd_loss_bwd(pred, soft, c0, c1, T, g.contiguous().view(1)), None, None, None, None


def kd_loss(pred, soft, c0, c1, T=2.0):
    return KDLossFn.apply(pred, soft, c0, c1, T)


# ---------------------------------------------------------------------------------------------------------
# SVTR mixing blocks in expert training (loop A): autograd of modules/svtr.py Block / Attention / Mlp / SubSample
def _pack_linear(w):
    """(HL32 stack, scale) of a trained Linear weight [N,K] as the x3 GEMM's weight operand"""
    N, K = w.shape
    return ops.pack_weights_hl32([w.detach().contiguous().view(N, 1, 1, K)])


def _pack_linear_t(w):
    """... of its transpose (the data gradient dx = dy W multiplies by W^T as the [K,N] weight operand)"""
    N, K = w.shape
    return ops.pack_weights_hl32([w.detach().t().contiguous().view(K, 1, 1, N)])


class TrainLinearFn(torch.autograd.Function):
    """y = x W^T + b over contiguous rows, Linear layers of an expert being trained: forward, data gradient and weight
    gradient on the range-safe split-fp16 x3 GEMM when the shape is eligible (exact fp32 otherwise)."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_range=False, dx_range=False):
        """want_range: max|y| is folded into the GEMM's epilogue and its power-of-two scale left for the pass that derives the next
        trained operand from y under a bound (|attention(qkv)| <= max|qkv|, |gelu(f)| <= |f|): ops.cached_operand(y).
        dx_range: the same for dx in the backward pass, for a GELU in front of this layer (|dx gelu'(.)| <= 1.13 |dx|)"""
        xin = x
        ctx.dx_range = dx_range
        x = x.contiguous()
        ctx.has_bias = bias is not None
        K, N = x.shape[-1], weight.shape[0]
        x2 = x.view(-1, K)
        # one max|x| pass serves the forward product and the weight gradient; one max|dy| pass both gradients -- and neither pass runs
        # when the producer of x / dy left the operand's range scale (and the split operand itself) behind
        ctx.x3 = x3_eligible(x2, N, K)
        if not ctx.x3:
            y, sx, sw = ops.linear(x2, weight, bias), None, None
        else:
            got = ops.cached_operand(xin)
            x_hl, sx = got if (got is not None and got[0] is not None) else (None, ops.pow2_scale(x2))
            w_pack = ops.train_pack("lin_fwd_x3", (weight,), _pack_linear) if isinstance(weight, torch.nn.Parameter) else None
            ws = ops._amax_ws() if want_range else None
            y, sw = x3_linear(x2, weight, bias, sx=sx, want_sw=True, x_hl=x_hl, w_pack=w_pack, amax_ws=ws)
        ctx.save_for_backward(x, weight, sx, sw)          # (sw: max|W| is the same for W^T in the data gradient)
        ctx.params = (weight, bias)
        ctx.use_gen = ops.note_param_uses(ctx.params, any(ctx.needs_input_grad))
        y = y.view(*x.shape[:-1], N)
        if ctx.x3 and want_range:
            ops.stash_operand(y, None, ops.pow2_finalize(ops.FP16_WEIGHT_PEAK))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, sx, sw = ctx.saved_tensors
        got = ops.cached_operand(dy)
        dy2 = dy.contiguous().view(-1, dy.shape[-1])
        sd = dy_hl = None
        if ctx.x3 or x3_eligible(dy2, x.shape[-1], dy2.shape[1]):
            if got is not None and dy2.data_ptr() == dy.data_ptr():
                dy_hl, sd = got
            else:
                sd = ops.pow2_scale(dy2)
        wt_pack = None
        if ctx.needs_input_grad[0] and sd is not None and isinstance(ctx.params[0], torch.nn.Parameter) and x3_eligible(dy2, x.shape[-1], dy2.shape[1]):
            wt_pack = ops.train_pack("lin_bwd_x3", (ctx.params[0],), _pack_linear_t)
        dx = None
        if ctx.needs_input_grad[0]:
            fold = ctx.dx_range and wt_pack is not None and ops.TRAIN_OPERAND_FUSION
            dx = linear_dgrad(dy2, weight, sd=sd, sw=sw, wt_pack=wt_pack, amax_ws=ops._amax_ws() if fold else None, dy_hl=dy_hl).view(x.shape)
            if fold:        # half the range target: the consumer multiplies by gelu'(.) <= 1.13
                ops.stash_operand(dx, None, ops.pow2_finalize(ops.FP16_WEIGHT_PEAK / 2), grad=True)
        need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        def grads(o):
            if not need_w:
                return None, (ops.colsum(dy2, out=o[1], accumulate=True) if need_b else None)
            db = None
            if need_b:         # the bias gradient rides on the weight gradient's pass over dy
                db = o[1] if o[1] is not None else torch.empty(dy2.shape[1], device=dy2.device, dtype=torch.float32)
            return linear_wgrad(dy2, x.view(-1, x.shape[-1]), sd, sx, out=o[0], bias_out=db, bias_accumulate=o[1] is not None), db
        dw, db = side_param_grads([ctx.params[0] if need_w else None, ctx.params[1] if need_b else None], grads, (dy2, x, sd, sx), into=True,
                                  gen=ctx.use_gen)
        return dx, dw, db, None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, operand=False):
        """operand: the result feeds a trained Linear layer -- the pass also writes that GEMM's range-scaled HL32 operand (the scale is
        a bound from gamma / beta: no max|y| pass, no split pass) and leaves it for TrainLinearFn (ops.cached_operand)"""
        x = x.contiguous()
        hl = sc = None
        if operand and x.shape[-1] % 32 == 0 and ops.ROUTER_GEMM_PRECISION == "fp16x3" and ops.TRAIN_OPERAND_FUSION:
            y, mean, rstd, hl, sc = ops.layernorm_fwd_operand(x, gamma, beta, eps)
        else:
            y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps)
        if hl is not None:
            ops.stash_operand(y, hl, sc)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.use_gen = ops.note_param_uses(ctx.params, any(ctx.needs_input_grad))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        gw, gb = (p.grad for p in ctx.params)
        C = gamma.numel()
        acc = None
        # direct mode: weight.grad and bias.grad are neighbours in the flat gradient -- the reduction of the partials adds into both at once
        if (ops.GRAD_DIRECT and not torch.is_grad_enabled() and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and gw is not None
                and gb is not None and gw.dtype == torch.float32 and gw.is_contiguous() and gb.is_contiguous()
                and gb.data_ptr() == gw.data_ptr() + 4 * C and gw.untyped_storage().data_ptr() == gb.untyped_storage().data_ptr()):
            acc = torch.as_strided(gw, (2 * C,), (1,))
        dx, dgamma, dbeta = ops.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, grad_acc=acc)
        if acc is not None:
            ops.direct_done(ctx.params, ctx.use_gen)
        return dx, dgamma, dbeta, None, None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        got = ops.cached_operand(x) if ops.TRAIN_OPERAND_FUSION else None
        x = x.contiguous()
        ctx.save_for_backward(x)
        if got is not None and x.shape[-1] % 32 == 0:
            # x came out of a trained Linear that left the scale of max|x| (TrainLinearFn want_range): |gelu(x)| <= |x|, so the same scale
            # serves the result -- written here as the next Linear's split operand too
            y, hl, _ = ops.ew_operand(ops.EW_GELU, x, scale=got[1], want_hl=True)
            ops.stash_operand(y, hl, got[1])
            return y
        return ops.ew_rows(ops.EW_GELU, x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if ops.TRAIN_OPERAND_FUSION and ops.ROUTER_GEMM_PRECISION == "fp16x3":
            # the gradient is a trained Linear's dy.  With the range of the incoming gradient at hand (the Linear behind this GELU folded
            # max|dy| into its data-gradient GEMM, at half the target: gelu' <= 1.13) this pass writes the split operand itself;
            # otherwise the result's exact range comes out of it
            got = ops.cached_operand(dy)
            dy = dy.contiguous()
            if got is not None and x.shape[-1] % 32 == 0:
                dx, hl, _ = ops.ew_operand(ops.EW_GELU_BWD, x, dy, scale=got[1], want_hl=True)
                ops.stash_operand(dx, hl, got[1], grad=True)
                return dx
            dx, _, sc = ops.ew_operand(ops.EW_GELU_BWD, x, dy, want_amax=ops.FP16_WEIGHT_PEAK)
            ops.stash_operand(dx, None, sc, grad=True)
            return dx
        return ops.ew_rows(ops.EW_GELU_BWD, x, dy.contiguous())


class AddPosFn(torch.autograd.Function):
    """tokens [B,N,C] + pos_embed [1,N,C] (svtr.py:503): the broadcast operand is read with row stride 0"""

    @staticmethod
    def forward(ctx, t, pos):
        B = t.shape[0]
        NC = t.shape[1] * t.shape[2]
        t = t.contiguous()
        return ops.ew_rows(ops.EW_ADD, t.view(B, NC), pos.view(1, NC).expand(B, NC)).view(t.shape)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dpos = ops.colsum(dy.view(dy.shape[0], -1)).view(1, dy.shape[1], dy.shape[2]) if ctx.needs_input_grad[1] else None
        return dy, dpos


class ResidualScaleFn(torch.autograd.Function):
    """x + drop[b] * branch (residual add with the per-sample DropPath multiplier, svtr.py:17-22,202-203); drop None: x + branch"""

    @staticmethod
    def forward(ctx, x, branch, drop, rows_per_sample):
        x, branch = x.contiguous(), branch.contiguous()
        ctx.drop, ctx.rps = drop, rows_per_sample
        if drop is None:
            return ops.ew_rows(ops.EW_ADD, x, branch)
        return ops.residual_scale_rows(x, branch, drop, rows_per_sample)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if ctx.drop is None:
            return dy, dy, None, None
        if ops.TRAIN_OPERAND_FUSION and ops.ROUTER_GEMM_PRECISION == "fp16x3":
            # drop * dy is the dy of the branch's last trained Linear (proj / fc2): its range comes out of this pass
            db, _, sc = ops.ew_operand(ops.EW_RESIDUAL_SCALE, dy, dy, drop=ctx.drop - 1.0, rows_per_drop=ctx.rps, want_amax=ops.FP16_WEIGHT_PEAK)
            ops.stash_operand(db, None, sc, grad=True)
            return dy, db, None, None
        return dy, ops.residual_scale_rows(dy, dy, ctx.drop - 1.0, ctx.rps), None, None       # dy + (drop - 1) dy = drop * dy


class SvtrAttentionFn(torch.autograd.Function):
    """softmax(scale q k^T + mask) v for every head (svtr.py:140-149).  Head dimension 32: the fused attention kernel forward
    (keeping the log-sum-exp) and the two recomputing backward kernels of csrc/attention.hip.  Otherwise (or with
    MRN_SVTR_ATTENTION=gemm) per-head batched exact-fp32 GEMMs on strided views of qkv [B,N,3C] with the probabilities kept."""

    @staticmethod
    def forward(ctx, qkv, mask, heads, scale):
        qkv_in = qkv
        qkv = qkv.contiguous()
        B, N, C3 = qkv.shape
        C, h = C3 // 3, heads
        d = C // h
        ctx.fused = d == 32 and ops.SVTR_FUSED_ATTENTION
        if ctx.fused:      # flash-style: the fused forward kernel keeps only the log-sum-exp, backward recomputes the tiles
            got = ops.cached_operand(qkv_in) if ops.TRAIN_OPERAND_FUSION else None
            if got is not None:
                # qkv came out of a trained Linear that left the scale of max|qkv|: rows of the result are convex combinations of rows of
                # v, so that scale serves it -- written here as the proj Linear's split operand too
                out, lse, hl = ops.svtr_attention(qkv, h, scale, mask, want_lse=True, want_hl=True, hl_scale=got[1])
                ops.stash_operand(out, hl, got[1])
            else:
                out, lse = ops.svtr_attention(qkv, h, scale, mask, want_lse=True)
            ctx.save_for_backward(qkv, out, lse, mask)
            ctx.cfg = (h, scale)
            return out
        attn = torch.empty(B, h, N, N, device=qkv.device, dtype=torch.float32)
        out = torch.empty(B, N, C, device=qkv.device, dtype=torch.float32)
        sq, sp = (N * C3, C3, 1), (h * N * N, N, 1)
        for hd in range(h):
            q, k = qkv[:, :, hd * d:(hd + 1) * d], qkv[:, :, C + hd * d:C + (hd + 1) * d]
            ops.gemm_raw(q, k, attn[:, hd], N, N, d, B, sq, sq, sp, alpha=scale)
        ops.softmax_rows_(attn, mask)
        for hd in range(h):
            v = qkv[:, :, 2 * C + hd * d:2 * C + (hd + 1) * d]
            ops.gemm_raw(attn[:, hd], v, out[:, :, hd * d:(hd + 1) * d], N, d, N, B, sp, (N * C3, 1, C3), (N * C, C, 1))
        ctx.save_for_backward(qkv, attn)
        ctx.cfg = (h, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, scale = ctx.cfg
        dout = dout.contiguous()
        if ctx.fused:
            qkv, out, lse, mask = ctx.saved_tensors
            if ops.TRAIN_OPERAND_FUSION and ops.ROUTER_GEMM_PRECISION == "fp16x3":      # dqkv is the qkv Linear's dy: its range comes out of the two kernels
                dqkv, sc = ops.svtr_attention_bwd(qkv, mask, out, dout, lse, h, scale, want_range=ops.FP16_WEIGHT_PEAK)
                ops.stash_operand(dqkv, None, sc, grad=True)
                return dqkv, None, None, None
            return ops.svtr_attention_bwd(qkv, mask, out, dout, lse, h, scale), None, None, None
        qkv, attn = ctx.saved_tensors
        B, N, C3 = qkv.shape
        C = C3 // 3
        d = C // h
        dqkv = torch.empty_like(qkv)
        dp = torch.empty_like(attn)
        sq, sp, so = (N * C3, C3, 1), (h * N * N, N, 1), (N * C, C, 1)
        spt, sqt, sot = (h * N * N, 1, N), (N * C3, 1, C3), (N * C, 1, C)
        for hd in range(h):
            do = dout[:, :, hd * d:(hd + 1) * d]
            v = qkv[:, :, 2 * C + hd * d:2 * C + (hd + 1) * d]
            ops.gemm_raw(do, v, dp[:, hd], N, N, d, B, so, sq, sp)                                   # dP = dO V^T
            ops.gemm_raw(attn[:, hd], do, dqkv[:, :, 2 * C + hd * d:2 * C + (hd + 1) * d], N, d, N, B, spt, sot, sq)   # dV = P^T dO
        ops.softmax_rows_bwd_(attn, dp)                                                              # dS (the mask is constant)
        for hd in range(h):
            q, k = qkv[:, :, hd * d:(hd + 1) * d], qkv[:, :, C + hd * d:C + (hd + 1) * d]
            ops.gemm_raw(dp[:, hd], k, dqkv[:, :, hd * d:(hd + 1) * d], N, d, N, B, sp, sqt, sq, alpha=scale)           # dQ = scale dS K
            ops.gemm_raw(dp[:, hd], q, dqkv[:, :, C + hd * d:C + (hd + 1) * d], N, d, N, B, spt, sqt, sq, alpha=scale)  # dK = scale dS^T Q
        return dqkv, None, None, None
