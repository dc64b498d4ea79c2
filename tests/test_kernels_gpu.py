"""Kernel-level parity on the GPU: every C-ABI entry point against the fp32 torch-CPU op it replaces
(these CPU ops are exactly what the oracle restatement is composed of).  Tolerance: 1e-4 (north star), tighter
where the op is a plain reduction."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from mrn_amd import ops as o
    from mrn_amd._lib import LIB
    LIB.load()
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def cu(t):
    return t.cuda()


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(257, 130, 64), (128, 128, 16), (65, 33, 390), (1000, 512, 260), (3, 5, 7), (300, 40, 512)])
def test_linear_shapes(ops, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    for act, ref_act in ((0, lambda v: v), (1, F.relu), (2, F.gelu)):
        y = ops.linear(cu(x), cu(w), cu(b), act=act)
        assert_close(f"linear act{act}", y, ref_act(F.linear(x, w, b)), atol=1e-5 * K ** 0.5, rtol=1e-5)


def test_gemm_all_staging_modes(ops):
    """operands contiguous along k or along the row dim, vectorisable or not; batch + bias per row + residual"""
    Bz = 3
    for (M, N, K) in [(132, 72, 100), (65, 390, 390), (256, 64, 63)]:
        A = rnd(Bz, M, K, seed=4)
        W = rnd(Bz, N, K, seed=5)
        bias = rnd(M, seed=6)
        res = rnd(Bz, M, N, seed=7)
        ref = torch.einsum("bmk,bnk->bmn", A, W) + bias[None, :, None] + res
        for a_t in (False, True):
            for w_t in (False, True):
                Ad = cu(A.transpose(1, 2).contiguous()) if a_t else cu(A)      # [B,K,M] or [B,M,K]
                Wd = cu(W.transpose(1, 2).contiguous()) if w_t else cu(W)
                sA = (M * K, 1, M) if a_t else (M * K, K, 1)
                sW = (N * K, 1, N) if w_t else (N * K, K, 1)
                C = torch.empty(Bz, M, N, device="cuda")
                ops.gemm_raw(Ad, Wd, C, M, N, K, Bz, sA, sW, (M * N, N, 1), bias=cu(bias), bias_axis=1, residual=cu(res))
                assert_close(f"gemm {M}x{N}x{K} aT={a_t} wT={w_t}", C, ref, atol=2e-5 * K ** 0.5, rtol=1e-5)
    # accumulate + alpha + strided output (transposed store)
    M, N, K = 70, 50, 36
    A, W, C0 = rnd(M, K, seed=8), rnd(N, K, seed=9), rnd(N, M, seed=10)
    Cd = cu(C0.clone())
    ops.gemm_raw(cu(A), cu(W), Cd, M, N, K, 1, (0, K, 1), (0, K, 1), (0, 1, M), accumulate=True, alpha=0.5)
    assert_close("gemm accumulate/alpha/transposed C", Cd, C0 + 0.5 * (A @ W.t()).t(), atol=1e-5)


CONVS = [
    # B, H, W, Cin, Cout, k, s, p
    (2, 32, 256, 4, 32, (3, 3), (1, 1), (1, 1)),
    (3, 16, 128, 32, 64, (3, 3), (1, 1), (1, 1)),
    (2, 8, 64, 64, 128, (1, 1), (1, 1), (0, 0)),
    (2, 4, 65, 512, 512, (3, 3), (1, 1), (1, 1)),
    (2, 4, 65, 256, 512, (2, 2), (2, 1), (0, 1)),
    (2, 2, 66, 128, 512, (2, 2), (1, 1), (0, 0)),
    (1, 5, 7, 8, 40, (3, 3), (1, 1), (1, 1)),
]


@pytest.mark.parametrize("cfg", CONVS)
def test_conv2d_and_batch_stats(ops, cfg):
    B, H, W, Cin, Cout, k, s, p = cfg
    x = rnd(B, Cin, H, W, seed=11)
    w = rnd(Cout, Cin, *k, seed=12, scale=(2.0 / (Cin * k[0] * k[1])) ** 0.5)
    b = rnd(Cout, seed=13)
    ref = F.conv2d(x, w, b, s, p)
    xn = ops.nchw_to_nhwc(cu(x))
    assert_close("nchw_to_nhwc", xn, x.permute(0, 2, 3, 1))
    wp = ops.pack_conv_weight(cu(w))
    assert_close("pack", wp.ohwi, w.permute(0, 2, 3, 1), atol=0)
    # split-fp16 x3 with the power-of-two weight prescale on the grouped kernel (G = 1; Cin % 32 == 0, Winograd F(4,3) where eligible;
    # other shapes fall through to the exact fp32 kernel): 22-bit products, same band as the exact fp32 kernel
    yh, sth = ops.conv2d_nhwc(xn, wp, cu(b), s, p, act=0, want_stats=True, precision="fp16x3")
    assert_close("conv fp16x3", yh.permute(0, 3, 1, 2), ref, atol=2e-5, rtol=1e-5)
    assert_close("fp16x3 stats", sth.view(-1, 2, Cout).sum(0)[0], ref.sum((0, 2, 3)), atol=2e-3, rtol=1e-4)
    yh1, _ = ops.conv2d_nhwc(xn, wp, cu(b), s, p, act=1, precision="auto")
    assert_close("conv auto + relu", yh1.permute(0, 3, 1, 2), F.relu(ref), atol=2e-5, rtol=1e-5)
    y, stats = ops.conv2d_nhwc(xn, wp, cu(b), s, p, act=0, want_stats=True, precision="f32")
    assert_close("conv", y.permute(0, 3, 1, 2), ref, atol=2e-5, rtol=1e-5)
    yr, _ = ops.conv2d_nhwc(xn, wp, cu(b), s, p, act=1, precision="f32")
    assert_close("conv+relu", yr.permute(0, 3, 1, 2), F.relu(ref), atol=2e-5, rtol=1e-5)
    # BatchNorm (training): statistics from the conv epilogue, running-stat update, apply + residual + relu
    gamma, beta = rnd(Cout, seed=14) + 1.5, rnd(Cout, seed=15)
    rm, rv = rnd(Cout, seed=16) * 0.1, rnd(Cout, seed=17) * 0.2 + 1.0
    rm_ref, rv_ref = rm.clone(), rv.clone()
    res = rnd(*ref.shape, seed=18)
    bn_ref = F.relu(F.batch_norm(ref, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5) + res)
    rm_d, rv_d = cu(rm), cu(rv)
    count = ref.shape[0] * ref.shape[2] * ref.shape[3]
    scale, shift, mean, invstd = ops.bn_finalize(stats, Cout, count, cu(gamma), cu(beta), rm_d, rv_d, 0.1, 1e-5, save=True)
    out = ops.scale_shift_act(y.clone(), scale, shift, relu=True, residual=cu(res.permute(0, 2, 3, 1).contiguous()))
    assert_close("bn train apply", out.permute(0, 3, 1, 2), bn_ref, atol=5e-5, rtol=1e-5)
    assert_close("running_mean", rm_d, rm_ref, atol=1e-6, rtol=1e-5)
    assert_close("running_var", rv_d, rv_ref, atol=1e-6, rtol=1e-5)
    assert_close("save_mean", mean, ref.mean((0, 2, 3)), atol=1e-6, rtol=1e-5)
    # eval-mode BN
    sc, sh = ops.bn_eval_affine(cu(gamma), cu(beta), cu(rm), cu(rv), 1e-5)
    out = ops.scale_shift_act(y.clone(), sc, sh, relu=False)
    assert_close("bn eval", out.permute(0, 3, 1, 2), F.batch_norm(ref, rm, rv, gamma, beta, False, 0.1, 1e-5), atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize("k,s,p", [((2, 2), (2, 2), (0, 0)), ((2, 1), (2, 1), (0, 0)), ((2, 2), (2, 1), (0, 1))])
def test_maxpool_and_avgpool(ops, k, s, p):
    x = rnd(3, 16, 8, 33, seed=19)
    xn = cu(x.permute(0, 2, 3, 1).contiguous())
    y = ops.maxpool_nhwc(xn, k, s, p)
    assert_close("maxpool", y.permute(0, 3, 1, 2), F.max_pool2d(x, k, s, p), atol=0, rtol=0)
    sc, sh = rnd(16, seed=20), rnd(16, seed=21)
    y = ops.maxpool_nhwc(xn, k, s, p, scale=cu(sc), shift=cu(sh), relu=True)
    ref = F.max_pool2d(F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]), k, s, p)
    assert_close("fused affine+relu+maxpool", y.permute(0, 3, 1, 2), ref, atol=1e-6)
    assert_close("avgpool", ops.avgpool_nhwc(xn), x.mean((2, 3)), atol=1e-6)


@pytest.mark.parametrize("k,s,p,H,W,C", [((2, 2), (2, 2), (0, 0), 16, 8, 32), ((2, 2), (2, 2), (0, 0), 15, 8, 32), ((2, 2), (2, 1), (0, 1), 16, 8, 32),
                                           ((2, 1), (2, 1), (0, 0), 16, 9, 36), ((2, 2), (2, 2), (0, 0), 8, 8, 33)])
def test_maxpool_backward(ops, k, s, p, H, W, C):
    """mrn_maxpool_bwd_nhwc_f32 against torch autograd: the non-atomic form for windows that tile the map (no zero fill: every element
    written, first maximum on ties) and the atomic form for overlapping / ragged ones"""
    x = rnd(3, C, H, W, seed=29)
    x[:, :, 0:2, 0:2] = 0.25                                  # ties: the first maximum in scan order takes the gradient
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, k, s, p)
    dy = rnd(*y.shape, seed=30)
    y.backward(dy)
    dx = ops.maxpool_bwd(cu(dy.permute(0, 2, 3, 1).contiguous()), cu(x.permute(0, 2, 3, 1).contiguous()), k, s, p)
    assert_close("maxpool backward", dx.permute(0, 3, 1, 2), xr.grad, atol=1e-6, rtol=0)


def test_tps_grid_sample(ops):
    from oracle.mrn_oracle import tps_constants
    from mrn_amd.tools.weights import fiducial_bias
    B, H, W = 3, 32, 256
    from mrn_amd.tools.weights import smooth_image
    img = torch.from_numpy(smooth_image("tps_test", (B, 4, H, W), 3))   # see smooth_image: fp32 conditioning of the TPS grid
    cp = torch.from_numpy(fiducial_bias(20)).view(1, 20, 2) + rnd(B, 20, 2, seed=23) * 0.15
    inv, ph = tps_constants(20, (H, W))
    cz = torch.cat([cp, torch.zeros(B, 3, 2)], 1)
    grid = torch.bmm(ph.repeat(B, 1, 1), torch.bmm(inv.repeat(B, 1, 1), cz)).reshape(B, H, W, 2)
    ref = F.grid_sample(img, grid, padding_mode="border", align_corners=True)
    out, g = ops.tps_grid_sample(ops.nchw_to_nhwc(cu(img)), cu(cp), cu(inv), cu(ph), (H, W), want_grid=True)
    assert_close("tps grid", g.view(B, H, W, 2), grid, atol=2e-5)
    # the grid itself is only reproducible to ~1e-5 in fp32 (ill-conditioned RBF sum) = 2e-3 pixel; image slope <= 0.1/pixel
    assert_close("tps sample", out.permute(0, 3, 1, 2), ref, atol=5e-4)


def test_bilstm_layer(ops):
    from oracle.mrn_oracle import _lstm_dir
    B, T, IN, Hd = 19, 13, 64, 256
    x = rnd(B, T, IN, seed=24)
    k = 1 / 16.0
    P = {n: rnd(*s, seed=25 + i, scale=k) for i, (n, s) in enumerate(
        [("wi", (4 * Hd, IN)), ("wh", (4 * Hd, Hd)), ("bi", (4 * Hd,)), ("bh", (4 * Hd,)),
         ("wi_r", (4 * Hd, IN)), ("wh_r", (4 * Hd, Hd)), ("bi_r", (4 * Hd,)), ("bh_r", (4 * Hd,))])}
    ref = torch.cat([_lstm_dir(x, P["wi"], P["wh"], P["bi"], P["bh"], False),
                     _lstm_dir(x, P["wi_r"], P["wh_r"], P["bi_r"], P["bh_r"], True)], 2)
    w_ih = cu(torch.cat([P["wi"], P["wi_r"]], 0))
    xproj = ops.linear(cu(x), w_ih, cu(torch.cat([P["bi"], P["bi_r"]])))
    w_hh = torch.stack([ops.pack_fragment_major(cu(P["wh"])), ops.pack_fragment_major(cu(P["wh_r"]))])
    out = ops.lstm_layer(xproj, w_hh, cu(torch.cat([P["bh"], P["bh_r"]])), Hd, 2)
    assert_close("bilstm", out, ref, atol=2e-5)


@pytest.mark.parametrize("is_train", [True, False])
def test_attention_decoder(ops, is_train):
    from oracle import mrn_oracle as O
    from mrn_amd.modules.prediction import Attention
    import torch.nn as nn
    B, T, D, Hd, C = 21, 65, 256, 256, 97
    fc = nn.Linear(Hd, C)
    att = Attention(D, Hd, C, fc)
    sd = {k: rnd(*v.shape, seed=40 + i, scale=0.08) for i, (k, v) in enumerate(att.state_dict().items())}
    sd["char_embeddings.weight"] = rnd(C, 256, seed=77)
    att.load_state_dict(sd)
    Hb = rnd(B, T, D, seed=41)
    text = torch.randint(0, C + 3, (B, 26), generator=torch.Generator().manual_seed(5))   # includes ids >= C (cut_unknown)
    text[:, 0] = 2
    tin = text if is_train else torch.LongTensor(B).fill_(2)
    osd = {"P." + k: v for k, v in sd.items()}
    ref = O.attention_forward(osd, "P.", Hb, tin, is_train, 25, sd["generator.weight"], sd["generator.bias"])
    att = att.cuda()
    with torch.no_grad():
        out = att(cu(Hb), cu(tin), is_train, 25)
    assert_close("attn decoder", out, ref, atol=1e-4)
    assert np.array_equal(out.argmax(2).cpu().numpy(), ref.argmax(2).numpy())


@pytest.mark.parametrize("B,D,T,S", [(37, 256, 65, 26), (21, 512, 65, 26), (3, 256, 65, 26), (5, 256, 150, 60)])
def test_attention_decoder_backward(ops, B, D, T, S):
    """BPTT through the 26 teacher-forced steps (attn_decoder_bwd_kernel, several workgroups with a ragged last one, D = 256
    and the DERNet-style wider context) against autograd through the oracle's step loop (reference prediction.py:58-68,102-118); the
    last case is a long line (150 positions, 60 steps: the deferred dHb / dHproj sums stage more than 64 KB in LDS)"""
    from oracle import mrn_oracle as O
    from mrn_amd.modules.prediction import Attention
    import torch.nn as nn
    Hd, C = 256, 97
    att = Attention(D, Hd, C, nn.Linear(Hd, C))
    sd = {k: rnd(*v.shape, seed=140 + i, scale=0.08) for i, (k, v) in enumerate(att.state_dict().items())}
    sd["char_embeddings.weight"] = rnd(C, 256, seed=177)
    att.load_state_dict(sd)
    Hb = rnd(B, T, D, seed=141)
    text = torch.randint(0, C + 3, (B, S), generator=torch.Generator().manual_seed(6))
    text[:, 0] = 2
    up = rnd(B, S, C, seed=142)
    osd = {"P." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    Hr = Hb.clone().requires_grad_(True)
    ref = O.attention_forward(osd, "P.", Hr, text, True, S - 1, osd["P.generator.weight"], osd["P.generator.bias"])
    (ref * up).sum().backward()
    att = att.cuda()
    Hc = cu(Hb).requires_grad_(True)
    out = att(Hc, cu(text), True, S - 1)
    (out * cu(up)).sum().backward()
    assert_close("decoder logits", out, ref, atol=1e-4)
    scale = Hr.grad.abs().max().item()
    assert_close("decoder dH", Hc.grad, Hr.grad, atol=2e-4 * max(scale, 1.0))
    for k, prm in att.named_parameters():
        g_ref = osd["P." + k].grad
        if g_ref is None:                      # generator.* aliases fc.* in the module's parameter list
            continue
        tol = 2e-4 * max(g_ref.abs().max().item(), 1.0)
        assert_close("decoder d" + k, prm.grad, g_ref, atol=tol)


def test_rowops(ops):
    R, C = 1000, 256
    x = rnd(R, 2 * C, seed=50)
    g, b = rnd(C, seed=51) + 1.2, rnd(C, seed=52)
    xd = cu(x)
    v = xd[:, C:]                                   # strided rows (chunk view)
    y, mean, rstd = ops.layernorm_fwd(v, cu(g), cu(b))
    xr = x[:, C:].clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br)
    assert_close("layernorm", y, ref, atol=1e-5)
    dy = rnd(R, C, seed=53)
    ref.backward(dy)
    dx, dg, db = ops.layernorm_bwd(cu(dy), v, cu(g), mean, rstd)
    assert_close("layernorm dx", dx, xr.grad, atol=1e-5)
    assert_close("layernorm dgamma", dg, gr.grad, atol=1e-4, rtol=1e-5)
    assert_close("layernorm dbeta", db, br.grad, atol=1e-4, rtol=1e-5)
    # colnorm: LayerNorm over P of [B,P,W]
    B, P, Wd = 5, 65, 768
    x = rnd(B, P, Wd, seed=54)
    g, b = rnd(P, seed=55) + 1.2, rnd(P, seed=56)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr.permute(0, 2, 1), (P,), gr, br).permute(0, 2, 1)
    y, mean, rstd = ops.colnorm_fwd(cu(x), cu(g), cu(b))
    assert_close("colnorm", y, ref, atol=1e-5)
    dy = rnd(B, P, Wd, seed=57)
    ref.backward(dy)
    dx, dg, db = ops.colnorm_bwd(cu(dy), cu(x), cu(g), mean, rstd)
    assert_close("colnorm dx", dx, xr.grad, atol=1e-5)
    assert_close("colnorm dgamma", dg, gr.grad, atol=1e-4, rtol=1e-5)
    assert_close("colnorm dbeta", db, br.grad, atol=1e-4, rtol=1e-5)
    # elementwise + colsum + gather + argmax
    a, bb = rnd(300, 512, seed=58) * 3, rnd(300, 512, seed=59)
    ar = a.clone().requires_grad_(True)
    F.gelu(ar).backward(bb)
    assert_close("gelu", ops.ew_rows(ops.EW_GELU, cu(a)), F.gelu(a), atol=1e-6)
    assert_close("gelu bwd", ops.ew_rows(ops.EW_GELU_BWD, cu(a), cu(bb)), ar.grad, atol=1e-6)
    assert_close("mul", ops.ew_rows(ops.EW_MUL, cu(a)[:, :256], cu(bb)[:, 256:]), a[:, :256] * bb[:, 256:], atol=0)
    big = rnd(100000, 40, seed=60)
    assert_close("colsum", ops.colsum(cu(big)), big.sum(0), atol=2e-3, rtol=1e-5)
    # the vectorised form (4 columns per thread, row lanes reduced through LDS) and its scalar fallback: narrow / wide / odd widths,
    # strided views, one chunk and many, accumulation into an existing vector
    for rows, C, ld in ((131072, 64, 64), (1000, 256, 256), (37, 12, 12), (5000, 1032, 1032), (3, 4, 4), (70000, 128, 384), (999, 30, 30),
                        (257, 2048, 2048), (8, 16384, 16384)):
        src = rnd(rows, ld, seed=63 + C)
        view = cu(src)[:, :C]
        ref_sum = src[:, :C].double().sum(0).float()
        assert_close(f"colsum {rows}x{C} ld {ld}", ops.colsum(view), ref_sum, atol=2e-4 * rows ** 0.5, rtol=1e-5)
        base = cu(rnd(C, seed=64))
        got = ops.colsum(view, out=base.clone(), accumulate=True)
        assert_close(f"colsum accumulate {rows}x{C}", got, ref_sum + base.cpu(), atol=2e-4 * rows ** 0.5, rtol=1e-5)
    m = rnd(30, 30, seed=61)
    perm = torch.randperm(30, generator=torch.Generator().manual_seed(1)).int()
    gat = ops.gather2d(cu(m), cu(perm), cu(perm), ld_out=32)
    assert_close("gather2d", gat, m[perm.long()][:, perm.long()], atol=0)
    lg = rnd(77, 5374, seed=62)
    assert np.array_equal(ops.argmax_lastdim(cu(lg)).cpu().numpy(), lg.argmax(1).numpy())


def test_fanin_gate_tail(ops):
    from oracle import mrn_oracle as O
    B, T = 9, 26
    classes = (41, 71, 98)
    logits = [rnd(B, T, c, seed=70 + i) for i, c in enumerate(classes)]
    w = torch.softmax(rnd(B, 3, seed=73), 1).requires_grad_(True)
    ref = O.fanin(logits, w)
    dl = []
    for l in logits:                      # padded-row device copies
        d = ops.padded_rows(B, T, l.shape[2], "cuda")
        d.copy_(l)
        dl.append(d)
    out = ops.fanin_fwd(dl, cu(w.detach()))
    assert_close("fanin", out, ref, atol=1e-6)
    dout = rnd(B, T, classes[-1], seed=74)
    ref.backward(dout)
    dd = ops.padded_rows(B, T, classes[-1], "cuda")
    dd.copy_(dout)
    assert_close("fanin dw", ops.fanin_bwd(dl, dd), w.grad, atol=1e-4, rtol=1e-5)
    idx = torch.tensor([0, 1, 2, 2, 1, 0, 0, 2, 1])
    assert_close("select_expert", ops.select_expert(dl, cu(idx)), O.select_expert(logits, idx), atol=0)
    # gate tail
    P, I = 65, 3
    r = rnd(B, P, I, seed=75).requires_grad_(True)
    Wr, br = (rnd(1, P, seed=76) * 0.3).requires_grad_(True), rnd(1, seed=77).requires_grad_(True)
    s_ref = F.linear(r.permute(0, 2, 1).contiguous(), Wr, br).squeeze(-1)
    w_ref = F.softmax(s_ref, -1)
    s, wt = ops.gate_tail_fwd(cu(r.detach()), cu(Wr.detach().view(-1)), cu(br.detach()))
    assert_close("gate w", wt, w_ref, atol=1e-6)
    dw = rnd(B, I, seed=78)
    w_ref.backward(dw)
    dr, dW, db = ops.gate_tail_bwd(wt, cu(dw), cu(r.detach()), cu(Wr.detach().view(-1)))
    assert_close("gate dr", dr, r.grad, atol=1e-6)
    assert_close("gate dW", dW, Wr.grad.view(-1), atol=1e-5)
    _, am = ops.gate_tail_fwd(cu(r.detach()), cu(Wr.detach().view(-1)), cu(br.detach()), hard=True)
    assert np.array_equal(am.cpu().numpy(), s_ref.argmax(1).numpy())


def test_ce_and_ctc_losses(ops):
    B, S, C = 12, 26, 98
    logits = (rnd(B, S, C, seed=80) * 4).requires_grad_(True)
    target = torch.randint(0, C, (B, S), generator=torch.Generator().manual_seed(2))
    target[:, 10:] = 1                                           # [PAD] -> ignored
    ref = F.cross_entropy(logits.view(-1, C), target.view(-1), ignore_index=1)
    (15 * ref).backward()
    d = ops.padded_rows(B, S, C, "cuda")
    d.copy_(logits.detach())
    loss, ctx = ops.ce_loss_fwd(d, cu(target), ignore_index=1)
    assert_close("ce loss", loss, ref.detach().view(1), atol=1e-5)
    dl = ops.ce_loss_bwd(ctx, torch.tensor([15.0], device="cuda"), d)
    assert_close("ce grad", dl, logits.grad, atol=1e-6, rtol=1e-4)
    # CTC: varied lengths incl. repeats, empty target, and an infeasible one (len 25 with repeats can exceed T)
    B, T, C = 7, 20, 40
    logits = (rnd(B, T, C, seed=81) * 3).requires_grad_(True)
    tl = torch.tensor([5, 1, 0, 12, 20, 3, 9], dtype=torch.int32)
    targets = torch.randint(2, C, (B, 25), generator=torch.Generator().manual_seed(3))
    targets[3, :12] = 7                                          # all repeats: needs 2*12-1 = 23 > T frames -> inf
    targets[5, :3] = torch.tensor([4, 4, 9])
    for b in range(B):
        targets[b, tl[b]:] = 1
    lp = logits.log_softmax(2).permute(1, 0, 2)
    ref = F.ctc_loss(lp, targets, torch.IntTensor([T] * B), tl, blank=0, reduction="mean", zero_infinity=True)
    (15 * ref).backward()
    d = ops.padded_rows(B, T, C, "cuda")
    d.copy_(logits.detach())
    loss, ctx = ops.ctc_loss_fwd(d, cu(targets), cu(tl))
    assert_close("ctc loss", loss, ref.detach().view(1), atol=1e-5, rtol=1e-5)
    dl = ops.ctc_loss_bwd(ctx, torch.tensor([15.0], device="cuda"))
    assert_close("ctc grad", dl, logits.grad, atol=2e-6, rtol=1e-4)


def test_clip_and_adam(ops):
    from oracle.mrn_oracle import clip_and_adam
    n = 100003
    p, g = rnd(n, seed=90), rnd(n, seed=91) * 0.1
    pr, st = [p.clone()], [{"m": torch.zeros(n), "v": torch.zeros(n)}]
    pd, gd = cu(p), cu(g)
    md, vd = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in (1, 2, 3):
        gg = g * step
        total = clip_and_adam(pr, [gg], st, 5e-4, step)
        gd.copy_(gg)
        nc = ops.grad_norm_clip(gd, 5.0)
        ops.adam_step(pd, gd, md, vd, nc, 5e-4, step)
        assert_close("grad norm", nc[0:1], total.view(1), atol=1e-4, rtol=1e-5)
    assert_close("adam params", pd, pr[0], atol=1e-6, rtol=1e-5)
    assert_close("adam m", md, st[0]["m"], atol=1e-7, rtol=1e-4)


def test_tps_backward_fiducial_gradient(ops):
    from oracle.mrn_oracle import tps_constants
    from mrn_amd.tools.weights import fiducial_bias, smooth_image, uniform
    B, H, W = 3, 32, 256
    img = torch.from_numpy(smooth_image("tps_test", (B, 4, H, W), 3))
    cp = (torch.from_numpy(fiducial_bias(20)).view(1, 20, 2) + torch.from_numpy(uniform("cpn", (B, 20, 2), -0.15, 0.15, 1))).requires_grad_(True)
    dout = torch.from_numpy(uniform("dout", (B, 4, H, W), -1, 1, 2))
    inv, ph = tps_constants(20, (H, W))
    cz = torch.cat([cp, torch.zeros(B, 3, 2)], 1)
    grid = torch.bmm(ph.repeat(B, 1, 1), torch.bmm(inv.repeat(B, 1, 1), cz)).reshape(B, H, W, 2)
    F.grid_sample(img, grid, padding_mode="border", align_corners=True).backward(dout)
    d = ops.tps_grid_sample_bwd(ops.nchw_to_nhwc(cu(img)), cu(cp.detach()), cu(inv), cu(ph), ops.nchw_to_nhwc(cu(dout)))
    # fp32 conditioning of the grid moves a few sampling points across pixel boundaries: ~1e-3 relative (f32 vs f64 on CPU: 6e-4)
    a, b = d.cpu().double().numpy(), cp.grad.double().numpy()
    rel = np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel < 3e-3, f"d C' relative L2 error {rel:.3e}"


# ---------------------------------------------------------------------------------------------------------
# grouped kernels of the lock-step expert path (conv_x3.hip, group_ops.hip, grouped recurrences)
# ---------------------------------------------------------------------------------------------------------
X3_CONVS = [
    # G, B, H, W, Cin, Cout, k, s, p, shared input
    (3, 5, 4, 65, 64, 128, (3, 3), (1, 1), (1, 1), False),      # image-row-major tiles + padded-tap skipping, ragged tiles
    (2, 3, 8, 17, 32, 64, (3, 3), (1, 1), (1, 1), True),        # one input for all groups
    (2, 7, 1, 1, 64, 96, (1, 1), (1, 1), (0, 0), False),        # a grouped Linear layer
    (2, 2, 4, 66, 64, 256, (2, 2), (2, 1), (0, 1), False),      # ResNet conv4_1 geometry
    (1, 3, 6, 10, 96, 320, (3, 3), (2, 2), (1, 1), False),      # strides, N not a tile multiple
    (6, 2, 4, 65, 512, 512, (3, 3), (1, 1), (1, 1), False),     # the dominant TRBA shape
    (3, 8, 4, 64, 64, 256, (3, 3), (1, 1), (1, 1), False),      # whole tiles per image row: class-ordered tile schedule
    (2, 4, 8, 64, 32, 320, (3, 3), (1, 1), (1, 1), False),      #   (interior rows first, border rows skip padded taps)
    (5, 4, 3, 64, 64, 128, (3, 3), (1, 1), (1, 1), False),      #   one interior row, tile count not a multiple of 8
]


@pytest.mark.parametrize("cfg", X3_CONVS)
def test_grouped_x3_conv_bn_pool(ops, cfg):
    """mrn_conv2d_x3_hl32 + mrn_bn_finalize_grouped_f32 + mrn_bn_apply_grouped_f32 + mrn_maxpool_grouped_f32 against
    torch fp32 conv2d / batch_norm / max_pool2d per group"""
    G, B, H, W, Cin, Cout, k, s, p, shared = cfg
    xs = [rnd(B, Cin, H, W, seed=100 + g) for g in range(1 if shared else G)]
    ws = [rnd(Cout, Cin, *k, seed=110 + g, scale=(2.0 / (Cin * k[0] * k[1])) ** 0.5) for g in range(G)]
    bias = rnd(G, Cout, seed=120)
    refs = [F.conv2d(xs[0 if shared else g], ws[g], bias[g], s, p) for g in range(G)]
    xn = torch.stack([x.permute(0, 2, 3, 1) for x in xs]).contiguous()
    x_hl = ops.split_hl32(cu(xn))
    w_hl, w_scale = ops.pack_weights_hl32([cu(w.permute(0, 2, 3, 1).contiguous()) for w in ws])
    y, stats = ops.conv2d_x3(x_hl, G, shared, B, H, W, Cin, w_hl, w_scale, Cout, k, s, p, bias=cu(bias), want_stats=True)
    Ho, Wo = refs[0].shape[2:]
    for g in range(G):
        assert_close(f"x3 conv g{g}", y[g].permute(0, 3, 1, 2), refs[g], atol=2e-5, rtol=1e-5)
    # strided destination (one expert's slice of a wider buffer)
    wide = torch.zeros(B * Ho * Wo, G, Cout + 4, device="cuda")
    ops.conv2d_x3(x_hl, G, shared, B, H, W, Cin, w_hl, w_scale, Cout, k, s, p, bias=cu(bias), act=1, out=wide,
                  out_row_stride=G * (Cout + 4), out_group_stride=Cout + 4)
    for g in range(G):
        assert_close(f"x3 conv strided g{g}", wide[:, g, :Cout], F.relu(refs[g]).permute(0, 2, 3, 1).reshape(-1, Cout), atol=2e-5, rtol=1e-5)
    assert float(wide[:, :, Cout:].abs().max()) == 0.0
    # BatchNorm (training) for all groups at once
    gamma, beta = rnd(G, Cout, seed=121) + 1.5, rnd(G, Cout, seed=122)
    rm, rv = rnd(G, Cout, seed=123) * 0.1, rnd(G, Cout, seed=124) * 0.2 + 1.0
    rm_ref, rv_ref = rm.clone(), rv.clone()
    res = rnd(G, *refs[0].shape, seed=125)
    bn_ref = [F.relu(F.batch_norm(refs[g], rm_ref[g], rv_ref[g], gamma[g], beta[g], True, 0.1, 1e-5) + res[g]) for g in range(G)]
    gd, bd, rmd, rvd = cu(gamma), cu(beta), cu(rm), cu(rv)
    table = torch.tensor([[t[g].data_ptr() for g in range(G)] for t in (gd, bd, rmd, rvd)], dtype=torch.int64, device="cuda")
    scale, shift = ops.bn_finalize_grouped(stats, G, Cout, B * Ho * Wo, table, 0.1, 1e-5)
    assert_close("grouped running_mean", rmd, rm_ref, atol=1e-6, rtol=1e-5)
    assert_close("grouped running_var", rvd, rv_ref, atol=1e-6, rtol=1e-5)
    resn = cu(res.permute(0, 1, 3, 4, 2).contiguous())
    want_hl = Cout % 32 == 0
    f32, hl = ops.bn_apply_grouped(y.clone(), scale, shift, relu=True, residual=resn, want_f32=True, want_hl=want_hl)
    for g in range(G):
        assert_close(f"grouped bn apply g{g}", f32[g].permute(0, 3, 1, 2), bn_ref[g], atol=5e-5, rtol=1e-5)
    if want_hl:
        # identity shortcut read back from an HL32 image (hi + lo, 22 bits) instead of the fp32 tensor
        f2, _ = ops.bn_apply_grouped(y.clone(), scale, shift, relu=True, residual_hl=ops.split_hl32(resn), want_f32=True)
        assert_close("grouped bn apply, HL32 residual", f2, f32, atol=1e-6, rtol=1e-6)
        # the HL32 operand is hi + lo of the same values, laid out [row][C/32][hi x 32 | lo x 32]
        v = hl.view(torch.float16).view(-1, Cout // 32, 2, 32).float()
        assert_close("HL32 image", (v[:, :, 0] + v[:, :, 1]).reshape(f32.shape), f32, atol=1e-6, rtol=2e-7)
    if Ho >= 2 and Wo >= 2:
        pf, _, _ = ops.maxpool_grouped(y, (2, 2), (2, 1), (0, 1), scale, shift, relu=True, want_f32=True)
        for g in range(G):
            ref = F.max_pool2d(F.relu(F.batch_norm(refs[g], None, None, gamma[g], beta[g], True, 0.0, 1e-5)), (2, 2), (2, 1), (0, 1))
            assert_close(f"grouped bn+relu+maxpool g{g}", pf[g].permute(0, 3, 1, 2), ref, atol=5e-5, rtol=1e-5)


def test_grouped_recurrences_match_single_launches(ops):
    """mrn_lstm_layer_fwd_grouped_f32 / mrn_attn_decoder_fwd_grouped_f32: bit-identical to one launch per expert"""
    G, B, T, Hd, S = 3, 19, 13, 256, 7
    xproj = cu(rnd(G, B, T, 8 * Hd, seed=130, scale=0.5))
    w_hh = torch.stack([torch.stack([ops.pack_fragment_major(cu(rnd(4 * Hd, Hd, seed=131 + 2 * g + d, scale=1 / 16.0)))
                                     for d in range(2)]) for g in range(G)]).contiguous()
    b_hh = cu(rnd(G, 8 * Hd, seed=140, scale=1 / 16.0))
    out = ops.lstm_layer_grouped(xproj, w_hh, b_hh, Hd, 2)
    for g in range(G):
        assert torch.equal(out[g], ops.lstm_layer(xproj[g], w_hh[g], b_hh[g], Hd, 2))
    D = 256
    Hb, Hproj = cu(rnd(G, B, T, D, seed=141)), cu(rnd(G, B, T, Hd, seed=142))
    eproj = cu(rnd(G, B, S, 4 * Hd, seed=143, scale=0.5))
    mk = lambda shape, seed: [cu(rnd(*shape, seed=seed + g, scale=1 / 16.0)) for g in range(G)]
    w_h2h = [ops.pack_fragment_major(w) for w in mk((Hd, Hd), 150)]
    w_ih = [ops.pack_fragment_major(w) for w in mk((4 * Hd, D), 160)]
    w_hh2 = [ops.pack_fragment_major(w) for w in mk((4 * Hd, Hd), 170)]
    b_h2h, w_score, b_hh2 = mk((Hd,), 180), mk((1, Hd), 190), mk((4 * Hd,), 200)
    hid = ops.attn_decoder_grouped(Hb, Hproj, eproj, w_h2h, b_h2h, w_score, w_ih, w_hh2, b_hh2, Hd)
    for g in range(G):
        ref = ops.attn_decoder(Hb[g], Hproj[g], eproj[g], w_h2h[g], b_h2h[g], w_score[g], w_ih[g], w_hh2[g], b_hh2[g], Hd)
        assert torch.equal(hid[g], ref)


@pytest.mark.parametrize("magnitude", [1.0, 3e-6, 2e4])
def test_range_safe_x3_linear(ops, magnitude):
    """functional.x3_linear (router Linear layers): both operands prescaled by a device-computed power of two, so
    gradient-sized (1e-6) and large (1e4) operands keep fp32-class accuracy on the fp16 MFMA path"""
    from mrn_amd import functional as Fn
    R, K, N = 1000, 256, 320
    x, w, b, res = rnd(R, K, seed=210) * magnitude, rnd(N, K, seed=211) * 0.07, rnd(N, seed=212) * magnitude, rnd(R, N, seed=213) * magnitude
    ref = (x.double() @ w.double().t() + b.double() + res.double())
    y = Fn.x3_linear(cu(x), cu(w), cu(b), residual=cu(res))
    assert_close("x3 linear", y.cpu().double(), ref, atol=3e-6 * magnitude, rtol=1e-6)
    # data gradient form: dy @ W
    dy = rnd(R, N, seed=214) * magnitude
    dx = Fn.linear_dgrad(cu(dy), cu(w))
    assert_close("x3 dgrad", dx.cpu().double(), dy.double() @ w.double(), atol=3e-6 * magnitude, rtol=1e-6)
    s = ops.pow2_scale(cu(x))
    m = float(x.abs().max())
    assert 8192.0 < float(s[0]) * m <= 16384.0 and abs(float(s[0]) * float(s[1]) - 1.0) < 1e-7


@pytest.mark.parametrize("magnitude", [1.0, 3e-6])
def test_x3_weight_gradient(ops, magnitude):
    """functional.x3_wgrad: dW = dy^T x through mrn_split_hl32_t_f32 (transposed split, split-K = groups)"""
    from mrn_amd import functional as Fn
    R, N, K = 8320, 320, 256
    dy, x = rnd(R, N, seed=220) * magnitude, rnd(R, K, seed=221)
    ref = dy.double().t() @ x.double()
    dw = Fn.linear_wgrad(cu(dy), cu(x))
    assert_close("x3 wgrad", dw.cpu().double(), ref, atol=2e-5 * magnitude, rtol=1e-6)


@pytest.mark.parametrize("cfg", [(3, 4, 65, 64, 128, (3, 3), (1, 1), (1, 1)), (2, 8, 17, 4, 32, (3, 3), (1, 1), (1, 1)),
                                 (2, 4, 66, 32, 64, (2, 2), (2, 1), (0, 1)), (5, 6, 10, 96, 320, (3, 3), (2, 2), (1, 1)),
                                 (2, 5, 9, 64, 48, (1, 1), (1, 1), (0, 0))])
@pytest.mark.parametrize("magnitude", [1.0, 1e-6])
def test_conv_weight_gradient_x3(ops, cfg, magnitude):
    """ops.conv2d_wgrad_x3 (mrn_split_hl32_t_f32 + mrn_im2col_t_hl32_f32 + one grouped x3 GEMM over (split, tap)) against
    torch autograd's conv2d weight gradient, also for gradient-sized dy"""
    B, H, W, Cin, Cout, k, s, p = cfg
    x = rnd(B, Cin, H, W, seed=230)
    w = rnd(Cout, Cin, *k, seed=231, scale=0.1).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = rnd(*y.shape, seed=232) * magnitude
    y.backward(dy)
    dw = ops.conv2d_wgrad_x3(cu(dy.permute(0, 2, 3, 1).contiguous()), cu(x.permute(0, 2, 3, 1).contiguous()), k, s, p)
    assert_close("x3 conv wgrad", dw.permute(0, 3, 1, 2), w.grad, atol=2e-5 * magnitude, rtol=2e-6)


def test_lstm_recurrence_on_f16_mfma(ops):
    """mrn_lstm_layer_fwd_x3_grouped (split-fp16 x3 recurrent product, fragment-major fp16 weight stream) against the
    exact-fp32 recurrent kernel"""
    G, B, T, Hd = 3, 19, 29, 256
    xproj = cu(rnd(G, B, T, 8 * Hd, seed=240, scale=0.7))
    ws = [[cu(rnd(4 * Hd, Hd, seed=241 + 2 * g + d, scale=(1 + g) / 16.0)) for d in range(2)] for g in range(G)]
    b_hh = cu(rnd(G, 8 * Hd, seed=250, scale=1 / 16.0))
    w_f32 = torch.stack([torch.stack([ops.pack_fragment_major(w) for w in p]) for p in ws]).contiguous()
    ref = ops.lstm_layer_grouped(xproj, w_f32, b_hh, Hd, 2)
    packs = [[ops.pack_fragment_major_h(w) for w in p] for p in ws]
    w_h = torch.stack([torch.stack([d[0] for d in p]) for p in packs]).contiguous()
    w_inv = torch.stack([torch.cat([d[1] for d in p]) for p in packs]).contiguous()
    out = ops.lstm_layer_x3_grouped(xproj, w_h, w_inv, b_hh, Hd, 2)
    assert_close("lstm on the f16 MFMA", out, ref, atol=2e-6, rtol=1e-5)


def test_lstm_x3_batch_beyond_the_32bit_row_offsets(ops):
    """the x3 layer kernel addresses its rows with 32-bit byte offsets into [B][T][ndir][4H]; the launcher cuts a batch whose array is
    beyond 4 GiB into chunks of whole tiles (rnn.hip: lstm_x3_launch).  T = 2048: 224 samples fit, B = 256 goes as 224 + 32 -- the same
    bits as the two parts launched separately"""
    G, B, T, Hd = 1, 256, 2048, 256
    gen = torch.Generator(device="cuda").manual_seed(77)
    xproj = torch.randn(G, B, T, 8 * Hd, device="cuda", generator=gen) * 0.7
    ws = [cu(rnd(4 * Hd, Hd, seed=261 + d, scale=1 / 16.0)) for d in range(2)]
    b_hh = cu(rnd(G, 8 * Hd, seed=263, scale=1 / 16.0))
    packs = [ops.pack_fragment_major_h(w) for w in ws]
    w_h = torch.stack([d[0] for d in packs]).unsqueeze(0).contiguous()
    w_inv = torch.cat([d[1] for d in packs]).unsqueeze(0).contiguous()
    out = ops.lstm_layer_x3_grouped(xproj, w_h, w_inv, b_hh, Hd, 2)
    for lo, hi in ((0, 224), (224, 256)):
        part = ops.lstm_layer_x3_grouped(xproj[:, lo:hi].contiguous(), w_h, w_inv, b_hh, Hd, 2)
        assert torch.equal(out[:, lo:hi], part), (lo, hi)
    assert torch.isfinite(out).all() and out.abs().max().item() > 0.1


@pytest.mark.parametrize("B,T,mag", [(19, 29, 1.0), (256, 65, 1e-5), (37, 12, 300.0)])
def test_lstm_training_kernels_on_f16_mfma(ops, B, T, mag):
    """mrn_lstm_layer_fwd_x3_save / mrn_lstm_layer_bwd_x3 (training forward with saves and backward through time, recurrent products as
    split-fp16 x3; the gate gradients range-scaled by a power of two) against the exact-fp32 kernels, for gradient-sized and large dout"""
    Hd, ndir = 256, 2
    xproj = cu(rnd(B, T, ndir * 4 * Hd, seed=440, scale=0.7))
    ws = [cu(rnd(4 * Hd, Hd, seed=441 + d, scale=1 / 16.0)) for d in range(ndir)]
    b_hh = cu(rnd(ndir * 4 * Hd, seed=450, scale=1 / 16.0))
    w_f32 = torch.stack([ops.pack_fragment_major(w) for w in ws]).contiguous()
    ref_out, ref_gates, ref_c = ops.lstm_layer(xproj, w_f32, b_hh, Hd, ndir, save=True)
    packs = [ops.pack_fragment_major_h(w) for w in ws]
    out, gates, cseq = ops.lstm_layer_x3_save(xproj, torch.stack([p[0] for p in packs]).contiguous(), torch.cat([p[1] for p in packs]).contiguous(),
                                             b_hh, Hd, ndir)
    assert_close("x3 training forward: out", out, ref_out, atol=2e-6, rtol=1e-5)
    assert_close("x3 training forward: gates", gates, ref_gates, atol=2e-6, rtol=1e-5)
    assert_close("x3 training forward: cell state", cseq, ref_c, atol=4e-6, rtol=1e-5)
    dout = cu(rnd(B, T, ndir * Hd, seed=460)) * mag
    wT_f32 = torch.stack([ops.pack_fragment_major(w.t().contiguous()) for w in ws]).contiguous()
    ref_dg = ops.lstm_layer_bwd(dout, ref_gates, ref_c, wT_f32, Hd, ndir)
    packsT = [ops.pack_fragment_major_h(w.t().contiguous()) for w in ws]
    dg = ops.lstm_layer_bwd_x3(dout, ref_gates, ref_c, torch.stack([p[0] for p in packsT]).contiguous(),
                               torch.cat([p[1] for p in packsT]).contiguous(), Hd, ndir)
    scale = float(ref_dg.abs().max())
    assert float((dg - ref_dg).abs().max()) <= 2e-6 * scale, (float((dg - ref_dg).abs().max()), scale)


def test_full_size_dominant_conv_properties(ops):
    """BASELINE-size check of the dominant kernel (6 experts x 256 images, 4x65 maps, 512->512 3x3) through properties that
    do not need a CPU reference: agreement with the exact-fp32 MFMA kernel, homogeneity under power-of-two scaling
    (the tap-skipping tile schedule and the hi/lo splits are scale-invariant up to fp16 subnormals), agreement of the fused BatchNorm partial
    statistics with reductions of the output, and independence of the groups."""
    G, B, H, W, C = 6, 256, 4, 65, 512
    torch.manual_seed(3)
    x = torch.rand(G, B, H, W, C, device="cuda") * 2 - 1
    ws = [(torch.rand(C, 3, 3, C, device="cuda") * 2 - 1) * 0.02 for _ in range(G)]
    w_hl, w_scale = ops.pack_weights_hl32(ws)
    y, stats = ops.conv2d_x3(ops.split_hl32(x), G, False, B, H, W, C, w_hl, w_scale, C, (3, 3), (1, 1), (1, 1), want_stats=True)
    for g in (0, 5):
        ref, _ = ops.conv2d_nhwc(x[g], ws[g], None, (1, 1), (1, 1), precision="f32")
        assert_close(f"full-size x3 vs exact fp32, expert {g}", y[g], ref, atol=1e-5, rtol=1e-5)
    y4, _ = ops.conv2d_x3(ops.split_hl32(x * 4.0), G, False, B, H, W, C, w_hl, w_scale, C, (3, 3), (1, 1), (1, 1))
    # homogeneity under a power-of-two scale: exact up to the lo halves that cross fp16's subnormal threshold
    assert_close("homogeneity", y4, y * 4.0, atol=2e-6, rtol=1e-6)
    tot = stats.view(G, -1, 2, C).sum(1)
    assert_close("fused column sums", tot[:, 0], y.sum((1, 2, 3)), atol=2e-2, rtol=2e-5)
    assert_close("fused column sums of squares", tot[:, 1], (y * y).sum((1, 2, 3)), atol=2e-2, rtol=2e-5)
    # groups are independent: expert 2 alone gives bit-identical rows
    w2, s2 = ops.pack_weights_hl32([ws[2]])
    y2, _ = ops.conv2d_x3(ops.split_hl32(x[2:3].contiguous()), 1, False, B, H, W, C, w2, s2, C, (3, 3), (1, 1), (1, 1))
    assert torch.equal(y2[0], y[2])


@pytest.mark.parametrize("x3", [False, True])
@pytest.mark.parametrize("B,N,heads,masked", [(3, 512, 2, True), (2, 256, 4, True), (5, 128, 8, False), (2, 100, 2, False)])
def test_svtr_fused_attention(ops, B, N, heads, masked, x3):
    """mrn_svtr_attention_f32 (online softmax; exact-fp32 MFMA products, or split-fp16 x3 products for the frozen experts)
    against torch softmax attention with the SVTR local mask"""
    from mrn_amd.modules.svtr import local_attention_mask
    C = heads * 32
    qkv = rnd(B, N, 3 * C, seed=260, scale=1.5)
    mask = local_attention_mask(N // 64, 64, 7, 11) if masked else None
    q, k, v = [t.reshape(B, N, heads, 32).permute(0, 2, 1, 3) for t in qkv.split(C, dim=2)]
    s = (q @ k.transpose(-1, -2)) * 32 ** -0.5
    if mask is not None:
        s = s + mask
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, C)
    out = ops.svtr_attention(cu(qkv), heads, 32 ** -0.5, cu(mask) if masked else None, x3=x3)
    assert_close("fused attention", out, ref, atol=4e-6 if x3 else 2e-6, rtol=1e-5)


def _hl32_to_f32(hl, rows, C):
    """HL32 bytes -> fp32 [rows, C] (hi + lo)"""
    v = hl.view(torch.float16).view(rows, C // 32, 2, 32).float()
    return (v[:, :, 0] + v[:, :, 1]).reshape(rows, C)


def test_svtr_fused_attention_hl32_output(ops):
    """the HL32 image the attention kernel writes for the proj Linear is the split of its fp32 output"""
    B, N, heads = 3, 256, 4
    C = heads * 32
    qkv = cu(rnd(B, N, 3 * C, seed=261))
    for x3 in (False, True):
        out, hl = ops.svtr_attention(qkv, heads, 32 ** -0.5, None, want_f32=True, want_hl=True, x3=x3)
        assert torch.equal(hl, ops.split_hl32(out))


@pytest.mark.parametrize("C,G,rows_pg", [(64, 3, 700), (128, 2, 515), (256, 3, 130), (512, 2, 67)])
def test_add_layernorm_grouped(ops, C, G, rows_pg):
    """mrn_add_layernorm_grouped_f32: DropPath-scaled residual add + per-expert LayerNorm + HL32 split in one pass"""
    N = rows_pg // 5 if rows_pg % 5 == 0 else 1                       # rows per DropPath sample
    rows = G * rows_pg
    x, br = rnd(rows, C, seed=270), rnd(rows, C, seed=271)
    drop = (torch.rand(rows // N, generator=torch.Generator().manual_seed(3)) > 0.3).float() / 0.7
    gamma, beta = rnd(G, C, seed=272) + 1.0, rnd(G, C, seed=273)
    t_ref = x + drop.repeat_interleave(N)[:, None] * br
    y_ref = torch.cat([torch.nn.functional.layer_norm(t_ref[g * rows_pg:(g + 1) * rows_pg], (C,), gamma[g], beta[g], 1e-6) for g in range(G)])
    t, y, hl = ops.add_layernorm_grouped(cu(x), cu(br), cu(drop), N, cu(gamma), cu(beta), rows_pg, 1e-6, want_sum=True,
                                         want_f32=True, want_hl=True)
    assert_close("residual sum", t, t_ref, atol=1e-6, rtol=1e-6)
    assert_close("layernorm", y, y_ref, atol=2e-5, rtol=1e-5)
    assert torch.equal(hl, ops.split_hl32(y))
    # no LayerNorm, no branch: plain operand split
    _, _, hl2 = ops.add_layernorm_grouped(cu(x), want_hl=True)
    assert torch.equal(hl2, ops.split_hl32(cu(x)))
    assert_close("hl32 round trip", _hl32_to_f32(hl2, rows, C), x, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("H,W,Cin,Cout,stride", [(32, 64, 4, 32, (2, 2)), (16, 32, 32, 64, (2, 2)), (8, 64, 64, 128, (2, 1)),
                                                 (2, 64, 256, 512, (2, 1)), (7, 9, 32, 64, (2, 2))])
def test_strided_conv_block_gradients(ops, H, W, Cin, Cout, stride):
    """ConvBlockFn (conv 3x3 pad 1, no BatchNorm) with stride > 1 -- SVTR's PatchEmbed / SubSample convs in expert training:
    data, weight and bias gradients against torch autograd, including the even-size case where the floor in the output-size
    formula leaves an input row that still receives gradient"""
    from mrn_amd.modules._nn import conv_block
    B = 3
    conv = torch.nn.Conv2d(Cin, Cout, 3, stride, 1)
    x = rnd(B, Cin, H, W, seed=280)
    xr = x.clone().requires_grad_(True)
    y = conv(xr)
    g = rnd(*y.shape, seed=281)
    y.backward(g)
    convc = torch.nn.Conv2d(Cin, Cout, 3, stride, 1).cuda()
    convc.load_state_dict(conv.state_dict())
    xc = cu(x).permute(0, 2, 3, 1).contiguous().requires_grad_(True)          # NHWC
    yc = conv_block(xc, convc, None, relu=False)
    assert_close("strided conv", yc.permute(0, 3, 1, 2), y, atol=2e-5, rtol=1e-5)
    yc.backward(cu(g).permute(0, 2, 3, 1).contiguous())
    assert_close("strided conv dx", xc.grad.permute(0, 3, 1, 2), xr.grad, atol=2e-5, rtol=1e-4)
    assert_close("strided conv dw", convc.weight.grad, conv.weight.grad, atol=1e-4, rtol=1e-4)
    assert_close("strided conv db", convc.bias.grad, conv.bias.grad, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("G,rows,K,N,act", [(3, 700, 64, 256, 2), (2, 300, 256, 1024, 2), (1, 515, 128, 64, 0), (2, 260, 512, 512, 1)])
def test_x3_linear_hl32_result(ops, G, rows, K, N, act):
    """mrn_conv2d_x3_hl32 with y_hl32: the epilogue writes the result as the next GEMM's HL32 operand -- identical to
    splitting the fp32 result"""
    x = rnd(G, rows, K, seed=290)
    w = [rnd(N, K, seed=291 + g, scale=K ** -0.5) for g in range(G)]
    b = cu(rnd(G, N, seed=295))
    w_hl, sw = ops.pack_weights_hl32([cu(t).view(N, 1, 1, K) for t in w])
    x_hl = ops.split_hl32(cu(x))
    y, _ = ops.conv2d_x3(x_hl, G, False, rows, 1, 1, K, w_hl, sw, N, (1, 1), bias=b, act=act)
    hl, _ = ops.conv2d_x3(x_hl, G, False, rows, 1, 1, K, w_hl, sw, N, (1, 1), bias=b, act=act, hl_only=True)
    assert torch.equal(hl, ops.split_hl32(y))


@pytest.mark.parametrize("shortcut", ["none", "f32", "hl32"])
def test_conv_x3_epilogue_folds_eval_batchnorm_and_shortcut(ops, shortcut):
    """mrn_conv2d_x3_hl32 with ch_scale / ch_shift (+ residual / residual_hl32, ReLU, fp32 and HL32 results): one launch gives
    what conv -> mrn_bn_apply_grouped_f32 (eval-mode affine + identity + ReLU + operand split) gives in two
    (reference modules/feature_extraction.py:184-199 in eval mode)"""
    G, B, H, W, Cin, Cout = 3, 5, 4, 13, 64, 96
    x = cu(rnd(G, B, H, W, Cin, seed=310))
    ws = [cu(rnd(Cout, 3, 3, Cin, seed=311 + g, scale=0.04)) for g in range(G)]
    scale, shift = cu(rnd(G, Cout, seed=315)) + 1.5, cu(rnd(G, Cout, seed=316))
    res = cu(rnd(G, B, H, W, Cout, seed=317))
    x_hl = ops.split_hl32(x)
    w_hl, sw = ops.pack_weights_hl32(ws)
    args = (x_hl, G, False, B, H, W, Cin, w_hl, sw, Cout, (3, 3), (1, 1), (1, 1))
    y, _ = ops.conv2d_x3(*args)
    res_hl = ops.split_hl32(res) if shortcut == "hl32" else None
    ref_f32, ref_hl = ops.bn_apply_grouped(y.clone(), scale, shift, relu=True, residual=res if shortcut == "f32" else None,
                                           want_f32=True, want_hl=True, residual_hl=res_hl)
    (got_f32, got_hl), _ = ops.conv2d_x3(*args, act=ops.ACT_RELU, ch_scale=scale, ch_shift=shift,
                                         residual=res if shortcut == "f32" else None, residual_hl=res_hl, also_hl=True)
    assert_close("fused eval layer", got_f32, ref_f32, atol=1e-6, rtol=1e-6)
    assert torch.equal(got_hl, ops.split_hl32(got_f32))
    only_hl, _ = ops.conv2d_x3(*args, act=ops.ACT_RELU, ch_scale=scale, ch_shift=shift,
                               residual=res if shortcut == "f32" else None, residual_hl=res_hl, hl_only=True)
    assert torch.equal(only_hl, got_hl)


@pytest.mark.parametrize("B,N,heads,H", [(3, 512, 2, 8), (2, 256, 4, 4), (3, 128, 8, 2), (2, 100, 2, 0), (2, 64, 8, 0)])
def test_svtr_attention_backward(ops, B, N, heads, H):
    """SvtrAttentionFn (fused forward keeping the log-sum-exp + the two recomputing backward kernels) against torch autograd
    of softmax(scale q k^T + mask) v, with the SVTR local mask (H > 0) and without, ragged N included"""
    from mrn_amd import functional as Fn
    from mrn_amd.modules.svtr import local_attention_mask
    C = heads * 32
    qkv = rnd(B, N, 3 * C, seed=300, scale=1.2).requires_grad_(True)
    mask = local_attention_mask(H, 64, 7, 11) if H else None
    g = rnd(B, N, C, seed=301)
    q, k, v = [t.reshape(B, N, heads, 32).permute(0, 2, 1, 3) for t in qkv.split(C, dim=2)]
    s = (q @ k.transpose(-1, -2)) * 32 ** -0.5
    if mask is not None:
        s = s + mask
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, C)
    ref.backward(g)
    qc = cu(qkv.detach()).requires_grad_(True)
    out = Fn.SvtrAttentionFn.apply(qc, cu(mask) if mask is not None else None, heads, 32 ** -0.5)
    out.backward(cu(g))
    assert_close("attention", out, ref, atol=2e-6, rtol=1e-5)
    for i, nm in enumerate(("dq", "dk", "dv")):
        assert_close(nm, qc.grad[:, :, i * C:(i + 1) * C], qkv.grad[:, :, i * C:(i + 1) * C], atol=5e-6, rtol=1e-4)


@pytest.mark.parametrize("G,B,H,W,Cout,shared,act", [(3, 5, 32, 256, 32, False, 0), (2, 3, 32, 256, 64, True, 1), (1, 2, 7, 100, 64, True, 0),
                                                     (2, 2, 5, 130, 32, False, 1), (2, 2, 6, 130, 64, False, 0)])
def test_first_conv_c4_grouped(ops, G, B, H, W, Cout, shared, act):
    """mrn_conv3x3_c4_grouped_f32 (first conv of the experts' stacks, Cin = 4) against torch conv2d: outputs, and the
    BatchNorm partial statistics summed over blocks against the pre-activation result; ragged widths included"""
    x = rnd(*( (B, H, W, 4) if shared else (G, B, H, W, 4) ), seed=320)
    w = rnd(G, Cout, 3, 3, 4, seed=321, scale=0.3)
    bias = rnd(G, Cout, seed=322)
    y, stats = ops.conv3x3_c4_grouped(cu(x), cu(w), cu(bias), act=act, want_stats=True)
    for g in range(G):
        xg = (x if shared else x[g]).permute(0, 3, 1, 2)
        pre = torch.nn.functional.conv2d(xg, w[g].permute(0, 3, 1, 2), bias[g], 1, 1)
        ref = torch.relu(pre) if act else pre
        assert_close(f"first conv, expert {g}", y[g].permute(0, 3, 1, 2), ref, atol=2e-5, rtol=1e-5)
        tot = stats[g].double().sum(0).cpu()
        assert_close("sum", tot[0].float(), pre.double().sum((0, 2, 3)).float(), atol=2e-2, rtol=1e-4)
        assert_close("sum of squares", tot[1].float(), (pre.double() ** 2).sum((0, 2, 3)).float(), atol=2e-2, rtol=1e-4)
    if H % 2 or W % 2:
        return
    # pooled epilogue: BatchNorm-apply + ReLU on the map of per-window extremes (mixed-sign BatchNorm weights) equals, bit for bit, the
    # 2 x 2 max-pool of BatchNorm-apply + ReLU on the full map; the statistics still cover the full map
    y0, _ = ops.conv3x3_c4_grouped(cu(x), cu(w), cu(bias), act=0, want_stats=False)
    gammas = [cu(rnd(Cout, seed=323 + g)) for g in range(G)]
    ptrs = torch.tensor([t.data_ptr() for t in gammas], dtype=torch.int64, device="cuda")
    bn_scale = torch.stack([t * (0.5 + 0.1 * g) for g, t in enumerate(gammas)]).contiguous()
    bn_shift = cu(rnd(G, Cout, seed=329, scale=0.3))
    yp, sp = ops.conv3x3_c4_grouped(cu(x), cu(w), cu(bias), act=0, want_stats=True, pool=True, gamma_ptrs=ptrs)
    assert torch.equal(sp, stats)
    got, _ = ops.bn_apply_grouped(yp.clone(), bn_scale, bn_shift, relu=True, want_f32=True, want_hl=False)
    ref, _, _ = ops.maxpool_grouped(y0, (2, 2), (2, 2), (0, 0), bn_scale, bn_shift, relu=True, want_f32=True, want_hl=False)
    assert torch.equal(got, ref)
    ypr, _ = ops.conv3x3_c4_grouped(cu(x), cu(w), cu(bias), act=1, pool=True)            # (VGG: no BatchNorm behind the pool)
    refr, _, _ = ops.maxpool_grouped(torch.relu(y0), (2, 2), (2, 2), (0, 0), None, None, relu=False, want_f32=True, want_hl=False)
    assert torch.equal(ypr, refr)


@pytest.mark.parametrize("G,B,H,W,Cin,Cout,shared", [(3, 3, 32, 256, 32, 64, False), (2, 2, 16, 128, 64, 128, False), (2, 5, 16, 64, 32, 64, True),
                                                     (1, 2, 12, 70, 64, 128, False), (2, 37, 8, 32, 32, 64, False)])
def test_patch_resident_conv(ops, G, B, H, W, Cin, Cout, shared):
    """mrn_conv3x3_patch_x3_hl32 (the narrow early 3x3 layers: weights in registers, activation patch in LDS) against torch conv2d in
    float64 on the split operands' values: outputs at the x3 products' 2^-22 level, BatchNorm partial statistics, the ReLU epilogue,
    ragged tiles (12 x 70: partial tiles in both directions), more tiles than workgroups (persistent loop), and the POOLED epilogue:
    with mixed-sign BatchNorm weights, BatchNorm-apply + ReLU on the pooled-extremes map equals -- bit for bit -- the 2 x 2 max-pool of
    BatchNorm-apply + ReLU on the full map (mrn_maxpool_grouped_f32 on the unpooled result of the same kernel)"""
    x = rnd(*((B, H, W, Cin) if shared else (G, B, H, W, Cin)), seed=330, scale=2.0)
    x = torch.relu(x) if Cin == 64 else x
    w = [rnd(Cout, 3, 3, Cin, seed=331 + g, scale=0.1) for g in range(G)]
    bias = rnd(G, Cout, seed=339)
    x_hl = ops.split_hl32(cu(x))
    w_hl, w_scale = ops.pack_weights_hl32([cu(t) for t in w])
    y, stats = ops.conv3x3_patch_x3(x_hl, G, shared, B, H, W, Cin, w_hl, w_scale, Cout, bias=cu(bias), want_stats=True)
    yr, _ = ops.conv3x3_patch_x3(x_hl, G, shared, B, H, W, Cin, w_hl, w_scale, Cout, bias=cu(bias), act=ops.ACT_RELU)
    for g in range(G):
        xg = (x if shared else x[g]).permute(0, 3, 1, 2).double()
        pre = torch.nn.functional.conv2d(xg, w[g].permute(0, 3, 1, 2).double(), bias[g].double(), 1, 1)
        scale = float(pre.abs().max())
        assert_close(f"patch conv, expert {g}", y[g].permute(0, 3, 1, 2), pre.float(), atol=3e-6 * scale, rtol=0)
        assert torch.equal(yr[g], torch.relu(y[g]))
        tot = stats[g].double().sum(0).cpu()
        n = B * H * W
        assert_close("sum", tot[0].float(), pre.sum((0, 2, 3)).float(), atol=1e-5 * scale * n, rtol=0)
        assert_close("sum of squares", tot[1].float(), (pre ** 2).sum((0, 2, 3)).float(), atol=1e-5 * scale * scale * n, rtol=0)
    # same launch twice: deterministic partial statistics (one row per persistent workgroup, fixed tile order)
    y2, stats2 = ops.conv3x3_patch_x3(x_hl, G, shared, B, H, W, Cin, w_hl, w_scale, Cout, bias=cu(bias), want_stats=True)
    assert torch.equal(y, y2) and torch.equal(stats, stats2)
    # ... and independent of the grouping: expert 0 launched alone (four to six times the workgroups per expert) gives the same bits --
    # tiles are summed into rows by tile index, not by the workgroup that happens to compute them
    if not shared:
        per_x, per_w = x_hl.numel() // G, w_hl.numel() // G
        y1, stats1 = ops.conv3x3_patch_x3(x_hl[:per_x], 1, False, B, H, W, Cin, w_hl[:per_w], w_scale[:1], Cout, bias=cu(bias)[:1], want_stats=True)
        assert torch.equal(y1[0], y[0]) and torch.equal(stats1[0], stats[0])
    if H % 2 or W % 2:
        return
    # pooled epilogue with mixed-sign BatchNorm weights (incl. an exact zero)
    gammas = [cu(rnd(Cout, seed=340 + g)) for g in range(G)]
    gammas[0][3] = 0.0
    ptrs = torch.tensor([t.data_ptr() for t in gammas], dtype=torch.int64, device="cuda")
    bn_scale = torch.stack([t * (0.5 + 0.1 * g) for g, t in enumerate(gammas)]).contiguous()       # sign(scale) = sign(gamma)
    bn_shift = cu(rnd(G, Cout, seed=349, scale=0.3))
    yp, sp = ops.conv3x3_patch_x3(x_hl, G, shared, B, H, W, Cin, w_hl, w_scale, Cout, bias=cu(bias), want_stats=True, pool=True, gamma_ptrs=ptrs)
    assert torch.equal(sp, stats)                                  # the statistics cover the unpooled map
    got, got_hl = ops.bn_apply_grouped(yp.clone(), bn_scale, bn_shift, relu=True, want_f32=True, want_hl=True)
    ref, ref_hl, _ = ops.maxpool_grouped(y, (2, 2), (2, 2), (0, 0), bn_scale, bn_shift, relu=True, want_f32=True, want_hl=True)
    assert torch.equal(got, ref) and torch.equal(got_hl, ref_hl)
    # no BatchNorm behind the pool (VGG): all maxima, ReLU in the epilogue
    ypr, _ = ops.conv3x3_patch_x3(x_hl, G, shared, B, H, W, Cin, w_hl, w_scale, Cout, bias=cu(bias), act=ops.ACT_RELU, pool=True)
    refr, _, _ = ops.maxpool_grouped(yr, (2, 2), (2, 2), (0, 0), None, None, relu=False, want_f32=True, want_hl=False)
    assert torch.equal(ypr, refr)
    # non-finite data: torch's max_pool2d / relu propagate NaN, v_max_f32 would return the finite neighbour.  One NaN
    # activation poisons the 3 x 3 output positions around it (all channels): every pooled window that holds one must be NaN, in the
    # pooled epilogue exactly where the separate pooling pass over the full map has it
    xn = (x if shared else x[0]).clone()
    xn[B - 1, 5, 7, 0] = float("nan")
    xn_hl = ops.split_hl32(cu(xn))
    w1, s1, b1 = w_hl[:w_hl.numel() // G], w_scale[:1], cu(bias)[:1].contiguous()
    yn, _ = ops.conv3x3_patch_x3(xn_hl, 1, False, B, H, W, Cin, w1, s1, Cout, bias=b1, act=ops.ACT_RELU)
    ynp, _ = ops.conv3x3_patch_x3(xn_hl, 1, False, B, H, W, Cin, w1, s1, Cout, bias=b1, act=ops.ACT_RELU, pool=True)
    refn, _, _ = ops.maxpool_grouped(yn, (2, 2), (2, 2), (0, 0), None, None, relu=False, want_f32=True, want_hl=False)
    want = torch.nn.functional.max_pool2d(yn[0].permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.isnan(yn[0, B - 1, 4:7, 6:9]).all() and int(torch.isnan(want).sum()) == 4 * Cout
    assert torch.equal(torch.isnan(ynp[0]), torch.isnan(want)) and torch.equal(torch.isnan(refn[0]), torch.isnan(want))
    assert torch.equal(torch.nan_to_num(ynp), torch.nan_to_num(refn))


@pytest.mark.parametrize("G,B,H,W,Cin,Cout", [(3, 3, 32, 256, 32, 64), (2, 2, 16, 128, 64, 128), (1, 2, 12, 70, 64, 128)])
def test_patch_resident_conv_one_product(ops, G, B, H, W, Cin, Cout):
    """mrn_conv3x3_patch_x1_hl32 (the reduced-precision mode's form of the patch-resident kernel: hi x hi only) against float64 torch on
    the operands' fp16 hi parts -- exact products, fp32 accumulation: 1e-5 of the output scale -- and its distance from the unquantised
    convolution (a few 1e-3: the mode's accuracy); statistics and the pooled epilogue as in the x3 form"""
    x = rnd(G, B, H, W, Cin, seed=330, scale=2.0)
    w = [rnd(Cout, 3, 3, Cin, seed=331 + g, scale=0.1) for g in range(G)]
    bias = rnd(G, Cout, seed=339)
    x_hl = ops.split_hl32(cu(x))
    w_hl, w_scale = ops.pack_weights_hl32([cu(t) for t in w])
    saved = ops.X3_PRODUCTS
    try:
        ops.X3_PRODUCTS = 1
        y, stats = ops.conv3x3_patch_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, bias=cu(bias), want_stats=True)
        if H % 2 == 0 and W % 2 == 0:
            yp, sp = ops.conv3x3_patch_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, bias=cu(bias), want_stats=True, pool=True)
    finally:
        ops.X3_PRODUCTS = saved
    sc = w_scale.cpu().double()
    for g in range(G):
        xq = x[g].half().double().permute(0, 3, 1, 2)
        wq = (w[g].double() * sc[g, 0]).half().double() * sc[g, 1]
        ref_q = torch.nn.functional.conv2d(xq, wq.permute(0, 3, 1, 2), bias[g].double(), 1, 1).permute(0, 2, 3, 1)
        ref = torch.nn.functional.conv2d(x[g].double().permute(0, 3, 1, 2), w[g].double().permute(0, 3, 1, 2), bias[g].double(), 1, 1).permute(0, 2, 3, 1)
        scale = float(ref.abs().max())
        got = y[g].double().cpu()
        assert float((got - ref_q).abs().max()) <= 1e-5 * scale, (float((got - ref_q).abs().max()), scale)
        err = float((got - ref).abs().max())
        assert 1e-5 * scale < err <= 1e-2 * scale, (err, scale)
        tot = stats[g].double().sum(0).cpu()
        assert_close("sum", tot[0].float(), got.sum((0, 1, 2)).float(), atol=1e-5 * scale * B * H * W, rtol=0)
    if H % 2 == 0 and W % 2 == 0:
        assert torch.equal(sp, stats)
        ref_p, _, _ = ops.maxpool_grouped(y, (2, 2), (2, 2), (0, 0), None, None, relu=False, want_f32=True, want_hl=False)
        assert torch.equal(yp, ref_p)


def test_sgd_and_adadelta_steps_vs_torch():
    """FlatSGD / FlatAdadelta (mrn_sgd_step_f32 / mrn_adadelta_step_f32) against torch.optim.SGD(momentum, weight_decay) and
    torch.optim.Adadelta(rho, eps) with clip_grad_norm_ in front -- the other two optimisers of il_modules/base.py:72-85"""
    from mrn_amd.optim import FlatAdadelta, FlatSGD
    g = torch.Generator().manual_seed(5)
    shapes = [(37, 5), (130,), (4, 3, 3, 3)]
    init = [torch.randn(*s, generator=g) for s in shapes]
    grads = [[torch.randn(*s, generator=g) * 3 for s in shapes] for _ in range(3)]
    for kind in ("sgd", "adadelta"):
        ref_p = [torch.nn.Parameter(t.clone()) for t in init]
        mine_p = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        if kind == "sgd":
            ref = torch.optim.SGD(ref_p, lr=0.05, momentum=0.9, weight_decay=1e-3)
            mine = FlatSGD(mine_p, 0.05, momentum=0.9, weight_decay=1e-3)
        else:
            ref = torch.optim.Adadelta(ref_p, lr=1.0, rho=0.95, eps=1e-8)
            mine = FlatAdadelta(mine_p, 1.0, rho=0.95, eps=1e-8)
        for it in range(3):
            for p, q, gr in zip(ref_p, mine_p, grads[it]):
                p.grad = gr.clone()
                q.grad.copy_(gr)
            total = torch.nn.utils.clip_grad_norm_(ref_p, 5.0)
            ref.step()
            v0 = mine_p[0]._version
            nc = mine.step(lr=0.05 if kind == "sgd" else 1.0, max_norm=5.0, momentum=0.8 if (kind == "sgd" and it == 2) else None)
            assert mine_p[0]._version > v0                       # raw-pointer update is announced to torch (repack caches)
            assert abs(float(nc[0]) - float(total)) <= 1e-5 * float(total)
            if kind == "sgd" and it == 2:                        # OneCycle-cycled momentum: torch reads param_groups each step
                break
            for p, q in zip(ref_p, mine_p):
                assert_close(f"{kind} step {it}", q.detach(), p.detach(), atol=1e-6, rtol=1e-5)


def test_non_finite_gradient_norm_skips_the_step_and_is_counted():
    """an inf / NaN in the gradient (an overflowed split-fp16 product chain): clip + update leave parameters and optimiser state as they
    were for all three optimisers, and the skip is COUNTED on the device (FlatOptimizer.skipped_steps(); the learners' log and bench.py's
    JSON show it) -- torch's clip_grad_norm_ + step would have written NaN into every parameter.  The next finite step applies."""
    from mrn_amd.optim import FlatAdadelta, FlatAdam, FlatSGD
    g = torch.Generator().manual_seed(6)
    shapes = [(37, 5), (130,), (4, 3, 3, 3)]
    for make in (lambda ps: FlatAdam(ps, lr=1e-3), lambda ps: FlatSGD(ps, 0.05, momentum=0.9, weight_decay=1e-3),
                 lambda ps: FlatAdadelta(ps, 1.0, rho=0.95, eps=1e-8)):
        ps = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
        opt = make(ps)
        for q in ps:
            q.grad.copy_(torch.randn(*q.shape, generator=g))
        opt.step(lr=opt.lr, max_norm=5.0)
        assert opt.skipped_steps() == 0
        before = (opt.flat.clone(), [t.clone() for t in opt.state])
        for bad in (float("inf"), float("nan")):
            for q in ps:
                q.grad.copy_(torch.randn(*q.shape, generator=g))
            ps[1].grad[17] = bad
            nc = opt.step(lr=opt.lr, max_norm=5.0)
            assert not torch.isfinite(nc[0])
            assert torch.equal(opt.flat, before[0]) and all(torch.equal(a, b) for a, b in zip(opt.state, before[1]))
        assert opt.skipped_steps() == 2
        for q in ps:
            q.grad.copy_(torch.randn(*q.shape, generator=g))
        opt.step(lr=opt.lr, max_norm=5.0)
        assert opt.skipped_steps() == 2 and not torch.equal(opt.flat, before[0]) and bool(torch.isfinite(opt.flat).all())


def test_native_rccl_entry_points_single_rank():
    """include/mrn_hip.h: mrn_comm_unique_id / mrn_comm_init / mrn_allreduce_f32 / mrn_broadcast_f32 / mrn_comm_destroy (RCCL bound
    with dlopen at init time) on a one-rank communicator, through the stream plumbing the learners use with MRN_COMM=native.
    (The multi-GPU collectives themselves can only run on the driver's 8-GPU node.)"""
    from mrn_amd import _lib, parallel
    assert parallel.init_native_comm(rank_=0, world=1) == 1
    assert _lib.call("mrn_comm_world") == 1 and _lib.call("mrn_comm_rank") == 0
    g = torch.randn(100003, device="cuda")
    ref = g.clone()
    parallel.native_all_reduce(g, average=True).wait()
    parallel.native_all_reduce(g, average=False).wait()
    parallel.native_broadcast(g, src=0)
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    with pytest.raises(RuntimeError):
        _lib.call("mrn_comm_init", 0, 1, __import__("ctypes").create_string_buffer(int(_lib.call("mrn_comm_unique_id_bytes"))))   # second communicator
    _lib.call("mrn_comm_destroy")
    parallel._native["world"] = 0
    with pytest.raises(RuntimeError):
        _lib.call("mrn_allreduce_f32", g.data_ptr(), g.numel(), 1, None)        # no communicator any more


def test_softargmax1d_matches_torch():
    """MRNNet.softargmax1d (reference modules/model.py:495-496): softmax(beta * x), forward and gradient"""
    import contextlib
    import io
    import types
    from mrn_amd.modules.model import MRNNet
    opt = types.SimpleNamespace(Transformation="None", FeatureExtraction="VGG", SequenceModeling="BiLSTM", Prediction="CTC",
                                num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512, hidden_size=256, batch_max_length=25)
    with contextlib.redirect_stdout(io.StringIO()):
        net = MRNNet(opt)
    x = torch.randn(7, 6)
    ref_in = x.clone().requires_grad_(True)
    ref = torch.softmax(5 * ref_in, -1)
    up = torch.randn(7, 6)
    (ref * up).sum().backward()
    mine_in = x.clone().cuda().requires_grad_(True)
    out = net.softargmax1d(mine_in, beta=5)
    (out * up.cuda()).sum().backward()
    assert_close("softargmax1d", out, ref.detach(), atol=1e-6, rtol=1e-5)
    assert_close("softargmax1d grad", mine_in.grad, ref_in.grad, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(32, 4, 65, 64, 128), (64, 2, 16, 32, 64), (32, 8, 64, 128, 96), (16, 8, 32, 64, 48), (8, 16, 128, 32, 100)])
def test_wgrad_without_im2col_matches_exact_fp32(B, H, W, Cin, Cout):
    """mrn_transpose_oy_hl32_f32 + mrn_gemm_x3_windows_hl32 (3x3 / s1 / p1 weight gradient over K-windows of dy^T and three
    x-shifted copies of x^T) against torch's conv weight gradient and against the im2col-based x3 path; gradient-sized dy"""
    from mrn_amd import ops
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn(B, H, W, Cin, generator=g)
    dy = torch.randn(B, H, W, Cout, generator=g) * 1e-4          # gradient magnitudes: the operands are prescaled on the device
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(),
                                      stride=1, padding=1).permute(0, 2, 3, 1)                 # [Cout,3,3,Cin]
    xd, dyd = x.cuda(), dy.cuda()
    assert ops.wgrad_windows_supported(dyd, xd, (3, 3), (1, 1), (1, 1))
    dw = ops.conv2d_wgrad_x3_windows(dyd, xd)
    old = ops.conv2d_wgrad_x3(dyd, xd, (3, 3), (1, 1), (1, 1))
    scale = ref.abs().max().item()
    assert (dw.cpu().double() - ref).abs().max().item() <= 2e-6 * scale
    assert (dw - old).abs().max().item() <= 2e-6 * scale
    assert not ops.wgrad_windows_supported(dyd[:, :1], xd[:, :1], (3, 3), (1, 1), (1, 1))        # one-row maps: the im2col path


# ---------------------------------------------------------------------------------------------------------
# Winograd F(R,3) form of the frozen experts' 3x3 convolutions (conv_x3.hip WINO, group_ops.hip producer)
# ---------------------------------------------------------------------------------------------------------
WINO_CONVS = [
    # G, B, H, W, Cin, Cout, shortcut ("none" | "f32" | "hl32"), relu
    (2, 3, 4, 65, 64, 128, "hl32", True),        # W not a multiple of R: the last group is partly outside the row
    (3, 2, 8, 64, 32, 96, "none", True),         # whole groups; Cout not a tile multiple; taller map
    (1, 5, 1, 9, 64, 64, "f32", False),          # one image row: both kernel-row neighbours in the padding; signed inputs
    (2, 32, 4, 65, 64, 160, "hl32", True),       # several row tiles, tiles straddle image rows
    (6, 2, 4, 65, 512, 512, "none", True),       # the dominant TRBA shape
    (2, 64, 4, 64, 32, 128, "none", True),       # B * ceil(W/R) a multiple of the tile height: class-ordered tile schedule
    (3, 16, 3, 32, 64, 128, "hl32", True),       #   (R = 4: 16 * 8 = 128) one interior row
    (2, 2, 16, 32, 32, 64, "none", True),        # four row blocks of the row-block kernel (conv_wino.hip): first / interior / last
    (1, 3, 12, 20, 64, 96, "f32", False),        # three row blocks, signed inputs, Cout and positions not tile multiples
    (2, 5, 4, 9, 32, 32, "none", True),          # one channel block per component, half a channel tile, three column groups
]


@pytest.mark.parametrize("R", [4, 2])
@pytest.mark.parametrize("cfg", WINO_CONVS)
def test_winograd_conv_matches_direct(ops, cfg, R):
    """mrn_bn_apply_wino_grouped_f32 -> mrn_conv2d_x3_wino_hl32 (+ mrn_pack_weight_wino_hl32) against torch:
    relu(y * scale + shift (+ shortcut)) -> conv2d(3x3, pad 1) + bias per group, the fused BatchNorm partial statistics, and
    the plain fp32 / HL32 by-products of the producer pass"""
    G, B, H, W, Cin, Cout, shortcut, relu = cfg
    yprev = rnd(G, B, H, W, Cin, seed=300)
    scale, shift = rnd(G, Cin, seed=301) + 1.5, rnd(G, Cin, seed=302) * 0.5
    res = rnd(G, B, H, W, Cin, seed=303) if shortcut != "none" else None
    ws = [rnd(Cout, Cin, 3, 3, seed=310 + g, scale=(2.0 / (Cin * 9)) ** 0.5) for g in range(G)]
    bias = rnd(G, Cout, seed=320)
    a = yprev.double() * scale.double()[:, None, None, None, :] + shift.double()[:, None, None, None, :]
    if res is not None:
        a = a + res.double()
    if relu:
        a = a.clamp_min(0)
    refs = [F.conv2d(a[g].permute(0, 3, 1, 2), ws[g].double(), bias[g].double(), 1, 1) for g in range(G)]
    resd = cu(res) if res is not None else None
    f32, hl, v = ops.bn_apply_wino_grouped(cu(yprev), cu(scale), cu(shift), R, relu=relu,
                                           residual=resd if shortcut == "f32" else None,
                                           residual_hl=ops.split_hl32(resd) if shortcut == "hl32" else None, want_f32=True, want_hl=True)
    tol = 4e-6 if shortcut == "hl32" else 1e-6       # an HL32 shortcut carries 22 bits
    assert_close("producer fp32 result", f32, a.float(), atol=tol, rtol=tol)
    hv = hl.view(torch.float16).view(-1, Cin // 32, 2, 32).float()
    assert_close("producer HL32 result", (hv[:, :, 0] + hv[:, :, 1]).reshape(f32.shape), f32, atol=1e-6, rtol=2e-7)
    u_hl, u_scale = ops.pack_weights_wino([cu(w.permute(0, 2, 3, 1).contiguous()) for w in ws], R)
    y, stats = ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u_hl, u_scale, Cout, R, bias=cu(bias), want_stats=True)
    for g in range(G):
        assert_close(f"winograd F({R},3) conv g{g}", y[g].permute(0, 3, 1, 2), refs[g].float(), atol=3e-5, rtol=1e-5)
    tot = stats.view(G, -1, 2, Cout).sum(1)
    assert_close("fused column sums", tot[:, 0], y.double().sum((1, 2, 3)).float(), atol=1e-3, rtol=2e-5)
    assert_close("fused column sums of squares", tot[:, 1], (y.double() ** 2).sum((1, 2, 3)).float(), atol=1e-3, rtol=2e-5)
    # ReLU in the epilogue, no statistics
    y1, _ = ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u_hl, u_scale, Cout, R, bias=cu(bias), act=1)
    assert torch.equal(y1, y.clamp_min(0))
    # the 2x2 / 2 max-pool behind BatchNorm + ReLU taken in the row-block kernel's epilogue (mrn_conv2d_x3_wino_pool_hl32): BatchNorm-apply +
    # ReLU on the map of per-window extremes (mixed-sign BatchNorm weights) == max-pool of the applied full map, bit for bit; same statistics
    if ops.wino_pool_supported(H, W, R, Cout):
        gammas = [cu(rnd(Cout, seed=330 + g)) for g in range(G)]
        ptrs = torch.tensor([t.data_ptr() for t in gammas], dtype=torch.int64, device="cuda")
        bn_scale = torch.stack([t * (0.5 + 0.1 * g) for g, t in enumerate(gammas)]).contiguous()
        bn_shift = cu(rnd(G, Cout, seed=339, scale=0.3))
        yp, sp = ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u_hl, u_scale, Cout, R, bias=cu(bias), want_stats=True, pool=True, gamma_ptrs=ptrs)
        assert torch.equal(sp, stats)
        got, _ = ops.bn_apply_grouped(yp.clone(), bn_scale, bn_shift, relu=True, want_f32=True, want_hl=False)
        ref, _, _ = ops.maxpool_grouped(y, (2, 2), (2, 2), (0, 0), bn_scale, bn_shift, relu=True, want_f32=True, want_hl=False)
        assert torch.equal(got, ref)


D16_CONVS = [(6, 2, 4, 65, 512, 512, "none", True), (2, 32, 4, 65, 64, 160, "hl32", True), (2, 2, 16, 32, 64, 64, "none", True),
             (1, 3, 12, 20, 128, 96, "f32", False), (3, 4, 8, 64, 128, 256, "none", True)]


@pytest.mark.parametrize("cfg", D16_CONVS)
def test_winograd_d16_reduced_mode(ops, cfg):
    """The reduced-precision form of the row-block Winograd convolution (mrn_bn_apply_wino_grouped_d16_f32 -> mrn_conv2d_x3_wino_d16 +
    mrn_pack_weight_wino_d16: ONE fp16 product per term on plain-fp16 operands, 64 channels per line) against (a) float64 arithmetic on
    the SAME fp16-quantised Winograd operands -- the kernel's own contract: fp32 accumulation of exact fp16 products, 1e-5 of the output
    scale -- and (b) the float64 convolution of the unquantised data: the fp16 mode's accuracy, a few 1e-3 of the scale (rms 1e-3).
    Plain by-products of the producer keep full precision; statistics, ReLU and the pooled epilogue behave as in the split form."""
    G, B, H, W, Cin, Cout, shortcut, relu = cfg
    R = 4
    yprev = rnd(G, B, H, W, Cin, seed=300)
    scale, shift = rnd(G, Cin, seed=301) + 1.5, rnd(G, Cin, seed=302) * 0.5
    res = rnd(G, B, H, W, Cin, seed=303) if shortcut != "none" else None
    ws = [rnd(Cout, Cin, 3, 3, seed=310 + g, scale=(2.0 / (Cin * 9)) ** 0.5) for g in range(G)]
    bias = rnd(G, Cout, seed=320)
    a = yprev.double() * scale.double()[:, None, None, None, :] + shift.double()[:, None, None, None, :]
    if res is not None:
        a = a + res.double()
    if relu:
        a = a.clamp_min(0)
    refs = torch.stack([F.conv2d(a[g].permute(0, 3, 1, 2), ws[g].double(), bias[g].double(), 1, 1).permute(0, 2, 3, 1) for g in range(G)])
    resd = cu(res) if res is not None else None
    f32, hl, v = ops.bn_apply_wino_grouped(cu(yprev), cu(scale), cu(shift), R, relu=relu, residual=resd if shortcut == "f32" else None,
                                           residual_hl=ops.split_hl32(resd) if shortcut == "hl32" else None, want_f32=True, want_hl=True,
                                           dense=True)
    Wq = (W + 3) // 4
    assert v.numel() == G * B * H * Wq * 6 * Cin * 2                   # half the bytes of the split operand
    assert_close("producer fp32 by-product", f32, a.float(), atol=4e-6 if shortcut == "hl32" else 1e-6, rtol=4e-6)
    hv = hl.view(torch.float16).view(-1, Cin // 32, 2, 32).float()
    assert_close("producer HL32 by-product", (hv[:, :, 0] + hv[:, :, 1]).reshape(f32.shape), f32, atol=1e-6, rtol=2e-7)
    u, u_scale = ops.pack_weights_wino([cu(w.permute(0, 2, 3, 1).contiguous()) for w in ws], R, dense=True)
    assert u.numel() == G * Cout * 6 * 3 * Cin * 2
    y, stats = ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u, u_scale, Cout, R, bias=cu(bias), want_stats=True, dense=True)
    # (a) float64 on the quantised operands: V [G,B,H,Wq,6,Cin] fp16, U [G,Cout,6,Cin/64,3,64] fp16 (scaled), A^T with the row scales of G
    Vq = v.view(torch.float16).view(G, B, H, Wq, 6, Cin).double().cpu()
    Uq = u.view(torch.float16).view(G, Cout, 6, Cin // 64, 3, 64).double().cpu().permute(0, 1, 2, 4, 3, 5).reshape(G, Cout, 6, 3, Cin)
    AT = torch.tensor([[0.25, 0.5, 0.5, 0.5, 0.5, 0.0], [0.0, 0.5, -0.5, 1.0, -1.0, 0.0], [0.0, 0.5, 0.5, 2.0, 2.0, 0.0],
                       [0.0, 0.5, -0.5, 4.0, -4.0, 1.0]], dtype=torch.float64)
    inv = u_scale.double().cpu()[:, 1]
    Vp = torch.nn.functional.pad(Vq, (0, 0, 0, 0, 0, 0, 1, 1))            # one zero image row above and below
    T = torch.zeros(G, B, H, Wq, 6, Cout, dtype=torch.float64)
    for ky in range(3):
        T += torch.einsum("gbhqmc,gomc->gbhqmo", Vp[:, :, ky:ky + H], Uq[:, :, :, ky])
    yq = torch.einsum("rm,gbhqmo->gbhqro", AT, T).reshape(G, B, H, Wq * 4, Cout)[:, :, :, :W] * inv[:, None, None, None, None]
    yq = yq + bias.double()[:, None, None, None, :]
    sc = float(refs.abs().max())
    err_q = float((y.double().cpu() - yq).abs().max())
    assert err_q <= 1e-5 * sc, (err_q, sc)
    # (b) the mode's accuracy against the unquantised convolution
    d = y.double().cpu() - refs
    assert float(d.abs().max()) <= 1.5e-2 * sc and float(d.pow(2).mean().sqrt()) <= 2e-3 * sc, (float(d.abs().max()), float(d.pow(2).mean().sqrt()), sc)
    assert float(d.abs().max()) > 1e-5 * sc                                # (it really is the one-product form)
    tot = stats.view(G, -1, 2, Cout).sum(1)
    assert_close("fused column sums", tot[:, 0], y.double().sum((1, 2, 3)).float(), atol=1e-3, rtol=2e-5)
    assert_close("fused column sums of squares", tot[:, 1], (y.double() ** 2).sum((1, 2, 3)).float(), atol=1e-3, rtol=2e-5)
    y1, _ = ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u, u_scale, Cout, R, bias=cu(bias), act=1, dense=True)
    assert torch.equal(y1, y.clamp_min(0))
    if ops.wino_pool_supported(H, W, R, Cout):
        gammas = [cu(rnd(Cout, seed=330 + g)) for g in range(G)]
        ptrs = torch.tensor([t.data_ptr() for t in gammas], dtype=torch.int64, device="cuda")
        bn_scale = torch.stack([t * (0.5 + 0.1 * g) for g, t in enumerate(gammas)]).contiguous()
        bn_shift = cu(rnd(G, Cout, seed=339, scale=0.3))
        yp, sp = ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u, u_scale, Cout, R, bias=cu(bias), want_stats=True, pool=True, gamma_ptrs=ptrs,
                                    dense=True)
        assert torch.equal(sp, stats)
        got, _ = ops.bn_apply_grouped(yp.clone(), bn_scale, bn_shift, relu=True, want_f32=True, want_hl=False)
        ref, _, _ = ops.maxpool_grouped(y, (2, 2), (2, 2), (0, 0), bn_scale, bn_shift, relu=True, want_f32=True, want_hl=False)
        assert torch.equal(got, ref)
    # the pooling producer in the same layout == the two-pass form, bit for bit
    if H % 2 == 0 and W % 2 == 0 and (H // 2) % 4 == 0:
        pf, _, _ = ops.maxpool_grouped(cu(yprev), (2, 2), (2, 2), (0, 0), cu(scale), cu(shift), relu=True, want_f32=True, want_hl=False)
        _, _, v_ref = ops.bn_apply_wino_grouped(pf, None, None, 4, relu=False, dense=True)
        _, _, v_p, _ = ops.maxpool_wino_grouped(cu(yprev), (2, 2), (2, 2), (0, 0), 4, cu(scale), cu(shift), relu=True, dense=True)
        assert torch.equal(v_p, v_ref)
    # a map the row-block kernel cannot take (H % 4 != 0): refused loudly, there is no fallback for this layout
    with pytest.raises(RuntimeError):
        ops.conv2d_x3_wino(v, G, False, B * H // 2, 2, W, Cin, u, u_scale, Cout, R, dense=True)


def test_winograd_d16_bf16_instantiation(ops):
    """ops.REDUCED_BF16 (bench.py --precision bf16): the d16 operands as bfloat16 on v_mfma_f32_32x32x16_bf16 -- BASELINE config 2 says
    "bf16"; the fp16 form is the mode that ships (11 significand bits against 8 at the same speed), this instantiation exists to compare.
    Equal to float64 arithmetic on ITS quantised operands within 1e-5 of the output scale; against the unquantised convolution its error
    is several times the fp16 form's (both asserted)."""
    G, B, H, W, Cin, Cout, R = 2, 8, 4, 65, 128, 128, 4
    yprev = rnd(G, B, H, W, Cin, seed=300)
    scale, shift = rnd(G, Cin, seed=301) + 1.5, rnd(G, Cin, seed=302) * 0.5
    ws = [rnd(Cout, Cin, 3, 3, seed=310 + g, scale=(2.0 / (Cin * 9)) ** 0.5) for g in range(G)]
    a = (yprev.double() * scale.double()[:, None, None, None, :] + shift.double()[:, None, None, None, :]).clamp_min(0)
    refs = torch.stack([F.conv2d(a[g].permute(0, 3, 1, 2), ws[g].double(), None, 1, 1).permute(0, 2, 3, 1) for g in range(G)])
    sc = float(refs.abs().max())
    AT = torch.tensor([[0.25, 0.5, 0.5, 0.5, 0.5, 0.0], [0.0, 0.5, -0.5, 1.0, -1.0, 0.0], [0.0, 0.5, 0.5, 2.0, 2.0, 0.0],
                       [0.0, 0.5, -0.5, 4.0, -4.0, 1.0]], dtype=torch.float64)
    Wq = (W + 3) // 4
    errs = {}
    saved = ops.REDUCED_BF16
    try:
        for bf16, dt in ((False, torch.float16), (True, torch.bfloat16)):
            ops.REDUCED_BF16 = bf16
            _, _, v = ops.bn_apply_wino_grouped(cu(yprev), cu(scale), cu(shift), R, relu=True, dense=True)
            u, u_scale = ops.pack_weights_wino([cu(w.permute(0, 2, 3, 1).contiguous()) for w in ws], R, dense=True)
            y, _ = ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u, u_scale, Cout, R, dense=True)
            Vq = v.view(dt).view(G, B, H, Wq, 6, Cin).double().cpu()
            Uq = u.view(dt).view(G, Cout, 6, Cin // 64, 3, 64).double().cpu().permute(0, 1, 2, 4, 3, 5).reshape(G, Cout, 6, 3, Cin)
            Vp = torch.nn.functional.pad(Vq, (0, 0, 0, 0, 0, 0, 1, 1))
            T = torch.zeros(G, B, H, Wq, 6, Cout, dtype=torch.float64)
            for ky in range(3):
                T += torch.einsum("gbhqmc,gomc->gbhqmo", Vp[:, :, ky:ky + H], Uq[:, :, :, ky])
            yq = torch.einsum("rm,gbhqmo->gbhqro", AT, T).reshape(G, B, H, Wq * 4, Cout)[:, :, :, :W] * u_scale.double().cpu()[:, 1][:, None, None, None, None]
            assert float((y.double().cpu() - yq).abs().max()) <= 1e-5 * sc, (bf16, float((y.double().cpu() - yq).abs().max()), sc)
            errs[bf16] = float((y.double().cpu() - refs).pow(2).mean().sqrt()) / sc
    finally:
        ops.REDUCED_BF16 = saved
    assert errs[False] < 2e-3 and 3 * errs[False] < errs[True] < 3e-2, errs


def test_winograd_full_size_dominant_shape_properties(ops):
    """BASELINE-size check of the Winograd form of the dominant layer (6 experts x 256 images, 4x65 maps, 512 -> 512) against the
    direct split-fp16 x3 kernel on the same post-ReLU activations, and of the fused statistics"""
    G, B, H, W, C, R = 6, 256, 4, 65, 512, 4
    torch.manual_seed(5)
    ypre = torch.randn(G, B, H, W, C, device="cuda")
    ws = [(torch.rand(C, 3, 3, C, device="cuda") * 2 - 1) * 0.02 for _ in range(G)]
    _, hl, v = ops.bn_apply_wino_grouped(ypre, None, None, R, relu=True, want_hl=True)
    w_hl, w_scale = ops.pack_weights_hl32(ws)
    yd, _ = ops.conv2d_x3(hl, G, False, B, H, W, C, w_hl, w_scale, C, (3, 3), (1, 1), (1, 1))
    u_hl, u_scale = ops.pack_weights_wino(ws, R)
    yw, stats = ops.conv2d_x3_wino(v, G, False, B, H, W, C, u_hl, u_scale, C, R, want_stats=True)
    assert_close("full-size winograd vs direct x3", yw, yd, atol=2e-5, rtol=1e-5)
    tot = stats.view(G, -1, 2, C).sum(1)
    assert_close("fused column sums", tot[:, 0], yw.sum((1, 2, 3)), atol=2e-2, rtol=2e-5)
    assert_close("fused column sums of squares", tot[:, 1], (yw * yw).sum((1, 2, 3)), atol=2e-2, rtol=2e-5)


@pytest.mark.parametrize("magnitude", [1.0, 1e-5, 300.0])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 8, 64, 128, 256), (3, 4, 64, 512, 512), (2, 4, 65, 128, 128)])
def test_trained_conv_winograd_range_safe(ops, B, H, W, Cin, Cout, magnitude):
    """loop A's trained convolutions on the Winograd form (ops.conv2d_x3_scaled with TRAIN_WINO: forward and data gradient): the
    operand's power-of-two range scale leaves the factor 16 of fp16 headroom the input transform B^T needs (row sums of |B^T| up
    to 10), so gradient-sized (1e-5) and large (300) operands keep 22-bit products"""
    assert ops.TRAIN_WINO and ops.TRAIN_OPERAND_PEAK == 1024.0
    x = cu(rnd(B, H, W, Cin, seed=400) * 3 * magnitude)
    w = cu(rnd(Cout, 3, 3, Cin, seed=401, scale=(2.0 / (9 * Cin)) ** 0.5))
    bias = cu(rnd(Cout, seed=402) * magnitude)
    y, stats = ops.conv2d_x3_scaled(x, w, bias, (1, 1), (1, 1), act=ops.ACT_NONE, want_stats=True)
    ref = F.conv2d(x.cpu().double().permute(0, 3, 1, 2), w.cpu().double().permute(0, 3, 1, 2), bias.cpu().double(), 1, 1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    assert float((y.cpu().double() - ref).abs().max()) <= 1e-5 * scale
    tot = stats.view(-1, 2, Cout).sum(0)
    assert_close("fused column sums", tot[0] / scale, (ref.sum((0, 1, 2)) / scale).float(), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("G,B,H,W,C,pool", [(2, 3, 8, 128, 64, ((2, 2), (2, 2), (0, 0))), (3, 2, 8, 64, 32, ((2, 2), (2, 1), (0, 1))),
                                            (1, 2, 4, 18, 64, ((2, 1), (2, 1), (0, 0)))])
def test_maxpool_winograd_producer(ops, G, B, H, W, C, pool):
    """mrn_maxpool_wino_grouped_f32 (BatchNorm-apply + ReLU + MaxPool2d + Winograd input transform in one pass) against the two-pass
    form: mrn_maxpool_grouped_f32, then the transform of the pooled map by mrn_bn_apply_wino_grouped_f32 -- bit-identical operands"""
    x = cu(rnd(G, B, H, W, C, seed=500))
    scale, shift = cu(rnd(G, C, seed=501) + 1.5), cu(rnd(G, C, seed=502) * 0.5)
    pf, ph, _ = ops.maxpool_grouped(x, pool[0], pool[1], pool[2], scale, shift, relu=True, want_f32=True, want_hl=True)
    _, _, v_ref = ops.bn_apply_wino_grouped(pf, None, None, 4, relu=False)
    f32, hl, v, (Ho, Wo) = ops.maxpool_wino_grouped(x, pool[0], pool[1], pool[2], 4, scale, shift, relu=True, want_f32=True, want_hl=True)
    assert (Ho, Wo) == tuple(pf.shape[2:4])
    assert torch.equal(f32, pf) and torch.equal(hl, ph) and torch.equal(v, v_ref)


@pytest.mark.parametrize("C,G,rows_pg", [(64, 3, 700), (128, 2, 515), (64, 1, 256), (128, 6, 31), (256, 2, 515), (256, 3, 128)])
def test_svtr_fused_mlp_matches_two_gemms(ops, C, G, rows_pg):
    """mrn_svtr_mlp_x3_f32 (fc1 -> GELU -> fc2 of G experts in one kernel, hidden activation in registers, chained MFMAs on the
    transposed problem with the hidden-permuted fc2 weights) against torch float64 and against the two grouped x3 GEMMs it replaces"""
    Ch = 4 * C
    x = rnd(G * rows_pg, C, seed=600) * 2
    w1 = [rnd(Ch, C, seed=601 + g, scale=(1.0 / C) ** 0.5) for g in range(G)]
    w2 = [rnd(C, Ch, seed=611 + g, scale=(1.0 / Ch) ** 0.5) for g in range(G)]
    b1, b2 = rnd(G, Ch, seed=620) * 0.2, rnd(G, C, seed=621) * 0.2
    xd = x.double().view(G, rows_pg, C)
    ref = torch.stack([F.gelu(xd[g] @ w1[g].double().t() + b1[g].double()) @ w2[g].double().t() + b2[g].double() for g in range(G)])
    x_hl = ops.split_hl32(cu(x))
    w1_hl, s1 = ops.pack_weights_hl32([cu(w).view(Ch, 1, 1, C).contiguous() for w in w1])
    perm = ops.mlp_hidden_permutation(Ch, torch.device("cuda"))
    w2_hl, s2 = ops.pack_weights_hl32([cu(w).index_select(1, perm).contiguous().view(C, 1, 1, Ch) for w in w2])
    y = ops.svtr_mlp_fused(x_hl, G * rows_pg, rows_pg, G, C, w1_hl, s1, cu(b1), w2_hl, s2, cu(b2))
    assert_close("fused SVTR Mlp vs float64", y.view(G, rows_pg, C), ref.float(), atol=2e-5, rtol=1e-5)
    # the path it replaces: fc1 + GELU -> HL32, fc2 (same products, another summation order inside fc2)
    w2n_hl, s2n = ops.pack_weights_hl32([cu(w).view(C, 1, 1, Ch).contiguous() for w in w2])
    h_hl, _ = ops.conv2d_x3(x_hl, G, False, rows_pg, 1, 1, C, w1_hl, s1, Ch, (1, 1), bias=cu(b1), act=ops.ACT_GELU, hl_only=True)
    y2, _ = ops.conv2d_x3(h_hl, G, False, rows_pg, 1, 1, Ch, w2n_hl, s2n, C, (1, 1), bias=cu(b2))
    assert_close("fused vs two GEMMs", y.view(G, rows_pg, C), y2.view(G, rows_pg, C), atol=2e-6, rtol=2e-6)


@pytest.mark.parametrize("G,imgs_pg,N,with_drop", [(2, 3, 128, True), (3, 1, 100, False), (1, 5, 128, True)])
def test_svtr_fused_tail_c256(ops, G, imgs_pg, N, with_drop):
    """mrn_svtr_tail_x3_f32 (stage 3: proj -> DropPath-scaled residual add -> LayerNorm2 -> fc1 -> GELU -> fc2 of G experts in one launch;
    the LayerNorm output becomes fc1's operand in registers through the input-channel permutation of W1) against torch float64: the
    updated residual stream and the Mlp branch; ragged row counts (100 tokens per image: partial 128-token tiles)"""
    C, Ch = 256, 1024
    rows_pg = imgs_pg * N
    rows = G * rows_pg
    ctx = rnd(rows, C, seed=640) * 1.5
    x = rnd(rows, C, seed=641) * 2
    wp = [rnd(C, C, seed=642 + g, scale=(1.0 / C) ** 0.5) for g in range(G)]
    w1 = [rnd(Ch, C, seed=650 + g, scale=(1.0 / C) ** 0.5) for g in range(G)]
    w2 = [rnd(C, Ch, seed=660 + g, scale=(1.0 / Ch) ** 0.5) for g in range(G)]
    bp, b1, b2 = rnd(G, C, seed=670) * 0.2, rnd(G, Ch, seed=671) * 0.2, rnd(G, C, seed=672) * 0.2
    gam, bet = rnd(G, C, seed=673) * 0.3 + 1.0, rnd(G, C, seed=674) * 0.2
    drop = (torch.tensor([0.0, 1.25] * ((G * imgs_pg + 1) // 2))[:G * imgs_pg]) if with_drop else None
    eps = 1e-6
    xd, cd = x.double().view(G, rows_pg, C), ctx.double().view(G, rows_pg, C)
    xo, br = [], []
    for g in range(G):
        d = drop.double().view(G, imgs_pg)[g].repeat_interleave(N)[:, None] if with_drop else 1.0
        t = xd[g] + d * (cd[g] @ wp[g].double().t() + bp[g].double())
        y = F.layer_norm(t, (C,), gam[g].double(), bet[g].double(), eps)
        xo.append(t)
        br.append(F.gelu(y @ w1[g].double().t() + b1[g].double()) @ w2[g].double().t() + b2[g].double())
    dev = torch.device("cuda")
    wp_hl, sp = ops.pack_weights_hl32([cu(w).view(C, 1, 1, C).contiguous() for w in wp])
    w1_hl, s1 = ops.pack_weights_hl32([cu(w).index_select(1, ops.mlp_hidden_permutation(C, dev)).contiguous().view(Ch, 1, 1, C) for w in w1])
    w2_hl, s2 = ops.pack_weights_hl32([cu(w).index_select(1, ops.mlp_hidden_permutation(Ch, dev)).contiguous().view(C, 1, 1, Ch) for w in w2])
    x_res = cu(x).clone()
    got = ops.svtr_tail_fused(ops.split_hl32(cu(ctx)), x_res, rows, rows_pg, G, C, wp_hl, sp, cu(bp), cu(drop) if with_drop else None, N,
                              cu(gam), cu(bet), eps, w1_hl, s1, cu(b1), w2_hl, s2, cu(b2))
    assert_close("residual stream after proj", x_res.view(G, rows_pg, C), torch.stack(xo).float(), atol=2e-5, rtol=1e-5)
    assert_close("Mlp branch of the fused tail", got.view(G, rows_pg, C), torch.stack(br).float(), atol=4e-5, rtol=2e-5)


def test_svtr_fused_mlp_full_size_no_stale_slabs(ops):
    """mrn_svtr_mlp_x3_f32 at SVTR stage-1 size (two experts x 131072 tokens, C = 64: two workgroups per CU, 1024 of them), several
    launches against the two-GEMM path: every row must agree.  Pins the LDS-DMA ordering inside the kernel's slab ring -- hipcc drops
    the vmcnt wait from a plain __syncthreads() there, which showed as stale weight slabs in a few workgroups per launch."""
    C, G, rpg = 64, 2, 131072
    Ch = 4 * C
    torch.manual_seed(11)
    x = torch.randn(G * rpg, C, device="cuda")
    w1 = [torch.randn(Ch, C, device="cuda") * C ** -0.5 for _ in range(G)]
    w2 = [torch.randn(C, Ch, device="cuda") * Ch ** -0.5 for _ in range(G)]
    b1, b2 = torch.randn(G, Ch, device="cuda") * 0.2, torch.randn(G, C, device="cuda") * 0.2
    x_hl = ops.split_hl32(x)
    w1_hl, s1 = ops.pack_weights_hl32([w.view(Ch, 1, 1, C).contiguous() for w in w1])
    perm = ops.mlp_hidden_permutation(Ch, x.device)
    w2_hl, s2 = ops.pack_weights_hl32([w.index_select(1, perm).contiguous().view(C, 1, 1, Ch) for w in w2])
    w2n_hl, s2n = ops.pack_weights_hl32([w.view(C, 1, 1, Ch).contiguous() for w in w2])
    h_hl, _ = ops.conv2d_x3(x_hl, G, False, rpg, 1, 1, C, w1_hl, s1, Ch, (1, 1), bias=b1, act=ops.ACT_GELU, hl_only=True)
    y2, _ = ops.conv2d_x3(h_hl, G, False, rpg, 1, 1, Ch, w2n_hl, s2n, C, (1, 1), bias=b2)
    for _ in range(6):
        y = ops.svtr_mlp_fused(x_hl, G * rpg, rpg, G, C, w1_hl, s1, b1, w2_hl, s2, b2)
        bad = ((y.view(-1, C) - y2.view(-1, C)).abs().max(1)[0] > 1e-4).sum()
        assert int(bad) == 0, int(bad)


def _mixer_case(C, N, G, B, masked, bias, with_pending, seed=700):
    from mrn_amd.modules.svtr import local_attention_mask
    heads = C // 32
    imgs = G * B
    t = {"x": rnd(imgs, N, C, seed=seed) * 1.5, "pend": rnd(imgs, N, C, seed=seed + 1) if with_pending else None,
         "g1": rnd(G, C, seed=seed + 2) * 0.3 + 1.0, "b1": rnd(G, C, seed=seed + 3) * 0.2,
         "g2": rnd(G, C, seed=seed + 4) * 0.3 + 1.0, "b2": rnd(G, C, seed=seed + 5) * 0.2,
         "wqkv": [rnd(3 * C, C, seed=seed + 10 + g, scale=(1.0 / C) ** 0.5) * 1.5 for g in range(G)],
         "bqkv": rnd(G, 3 * C, seed=seed + 20) * 0.3 if bias else None,
         "wproj": [rnd(C, C, seed=seed + 30 + g, scale=(1.0 / C) ** 0.5) for g in range(G)], "bproj": rnd(G, C, seed=seed + 40) * 0.2}
    gen = torch.Generator().manual_seed(seed)
    t["dprev"] = ((torch.rand(imgs, generator=gen) > 0.3).float() / 0.7) if with_pending else None
    t["d1"] = (torch.rand(imgs, generator=gen) > 0.3).float() / 0.7
    H = {200: 8, 100: 4, 224: 8, 128: 4, 193: 1, 97: 1, 512: 8, 256: 4, 300: 4, 250: 1, 160: 4}[N]
    t["mask"] = local_attention_mask(H, N // H, 7, 11) if masked else None
    return t, heads


def _mixer_reference(t, C, N, G, B, heads):
    """float64: t = x + d_prev * pending; x_out = t + d1 * proj(attention(qkv(LN1(t)))); y2 = LN2(x_out)"""
    x = t["x"].double()
    if t["pend"] is not None:
        x = x + t["dprev"].double()[:, None, None] * t["pend"].double()
    outs, ys = [], []
    for g in range(G):
        xg = x[g * B:(g + 1) * B]
        y = F.layer_norm(xg, (C,), t["g1"][g].double(), t["b1"][g].double(), 1e-6)
        qkv = y @ t["wqkv"][g].double().t()
        if t["bqkv"] is not None:
            qkv = qkv + t["bqkv"][g].double()
        q, k, v = [u.reshape(B, N, heads, 32).permute(0, 2, 1, 3) for u in qkv.split(C, dim=2)]
        s = (q @ k.transpose(-1, -2)) * 32 ** -0.5
        if t["mask"] is not None:
            s = s + t["mask"].double()
        ctx = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, C)
        br = ctx @ t["wproj"][g].double().t() + t["bproj"][g].double()
        xo = xg + t["d1"][g * B:(g + 1) * B].double()[:, None, None] * br
        outs.append(xo)
        ys.append(F.layer_norm(xo, (C,), t["g2"][g].double(), t["b2"][g].double(), 1e-6))
    return torch.cat(outs), torch.cat(ys)


def _mixer_run(ops, t, C, N, G, B, hw=None):
    dev = torch.device("cuda")
    wq, sq = ops.pack_weights_hl32([cu(w).view(3 * C, 1, 1, C).contiguous() for w in t["wqkv"]])
    perm = ops.mlp_hidden_permutation(C, dev)
    wp, sp = ops.pack_weights_hl32([cu(w).index_select(1, perm).contiguous().view(C, 1, 1, C) for w in t["wproj"]])
    o = lambda v: cu(v) if v is not None else None
    return ops.svtr_mixer_fused(cu(t["x"]), o(t["pend"]), o(t["dprev"]), cu(t["g1"]), cu(t["b1"]), 1e-6, wq, sq, o(t["bqkv"]), o(t["mask"]),
                                32 ** -0.5, wp, sp, cu(t["bproj"]), cu(t["d1"]), cu(t["g2"]), cu(t["b2"]), 1e-6, B, hw=hw)


@pytest.mark.parametrize("C,N,H,G,B", [(64, 512, 8, 2, 3), (128, 256, 4, 2, 3), (128, 160, 4, 1, 3), (64, 256, 8, 1, 2), (64, 512, 8, 6, 40)])
def test_svtr_fused_mixer_column_major_walk(ops, C, N, H, G, B):
    """LOCAL mixers walk their tokens column-major (mrn_svtr_mixer_x3_f32 token_h / token_w + position-ordered mask bits): the 7 x 11 window
    (modules/svtr.py:110-128) then leaves whole 32-position key tiles masked, and they are skipped.  Per token the same function: equal to
    float64 torch inside the usual band and to the memory-order walk to 4e-6 (another summation order over the key tiles); the mask bits in
    position order are the permuted mask's; a map the walk does not fit (N % 32 != 0) silently keeps the memory order."""
    t, heads = _mixer_case(C, N, G, B, True, True, True)
    W = N // H
    x_ref, y_ref = _mixer_reference(t, C, N, G, B, heads)
    x_col, y_col = _mixer_run(ops, t, C, N, G, B, hw=(H, W))
    saved = ops.SVTR_LOCAL_COLUMNS
    try:
        ops.SVTR_LOCAL_COLUMNS = False
        x_row, y_row = _mixer_run(ops, t, C, N, G, B, hw=(H, W))
    finally:
        ops.SVTR_LOCAL_COLUMNS = saved
    assert_close("column walk: residual stream vs float64", x_col, x_ref.float(), atol=2e-5, rtol=1e-5)
    assert_close("column walk: LayerNorm2 operand vs float64", _hl32_to_f32(y_col, G * B * N, C).view(G * B, N, C), y_ref.float(), atol=4e-5, rtol=1e-5)
    assert_close("column walk vs memory-order walk", x_col, x_row, atol=4e-6, rtol=2e-6)
    assert not torch.equal(x_col, x_row) or N <= 32          # (it really took the other order)
    m = cu(t["mask"])
    pos = torch.arange(N, device="cuda")
    tok = (pos % H) * W + pos // H
    assert torch.equal(ops._mask_bits_columns(m, H, W), ops._mask_bits(m[tok][:, tok].contiguous()))
    # fully masked tiles per wave of 32 queries (what the walk buys): count the (query tile, key tile) pairs with any visible key
    vis = (m[tok][:, tok] == 0).view(N // 32, 32, N // 32, 32).any(3).any(1).float().mean().item()
    vis_row = (m == 0).view(N // 32, 32, N // 32, 32).any(3).any(1).float().mean().item()
    assert vis < vis_row and (N < 512 or vis < 0.5 * vis_row), (vis, vis_row)


@pytest.mark.parametrize("C,N,G,B,masked,bias,with_pending", [(64, 200, 2, 3, True, True, True), (64, 200, 1, 5, False, False, False),
                                                              (128, 100, 2, 4, True, True, True), (128, 100, 3, 2, False, True, False),
                                                              (64, 224, 1, 2, False, True, True), (64, 193, 1, 2, False, True, False),
                                                              (128, 128, 1, 2, False, False, True), (128, 97, 1, 2, False, True, True),
                                                              (64, 512, 2, 3, True, True, True), (64, 512, 1, 2, False, False, False),
                                                              (64, 300, 1, 3, True, True, True), (64, 250, 1, 2, False, True, True),
                                                              (128, 256, 2, 3, True, True, True), (128, 160, 1, 3, True, False, False)])
def test_svtr_fused_mixer(ops, C, N, G, B, masked, bias, with_pending):
    """mrn_svtr_mixer_x3_f32 (LayerNorm1 -> qkv -> local / global attention -> proj -> DropPath-scaled residual -> LayerNorm2 of G
    experts in one kernel) against float64 torch (modules/svtr.py:90-152, :196-201) and against the unfused chain it replaces"""
    t, heads = _mixer_case(C, N, G, B, masked, bias, with_pending)
    x_ref, y_ref = _mixer_reference(t, C, N, G, B, heads)
    assert ops.svtr_mixer_supported(N, C, B, cu(t["mask"]) if masked else None)
    x_out, y_hl = _mixer_run(ops, t, C, N, G, B)
    assert_close("fused mixer: residual stream vs float64", x_out, x_ref.float(), atol=2e-5, rtol=1e-5)
    assert_close("fused mixer: LayerNorm2 operand vs float64", _hl32_to_f32(y_hl, G * B * N, C).view(G * B, N, C), y_ref.float(), atol=4e-5, rtol=1e-5)
    # the chain it replaces
    o = lambda v: cu(v) if v is not None else None
    rows = B * N
    tt, _, hl = ops.add_layernorm_grouped(cu(t["x"]), o(t["pend"]), o(t["dprev"]), N, cu(t["g1"]), cu(t["b1"]), rows, 1e-6, want_sum=with_pending)
    xs = tt if tt is not None else cu(t["x"])
    wq, sq = ops.pack_weights_hl32([cu(w).view(3 * C, 1, 1, C).contiguous() for w in t["wqkv"]])
    wp, sp = ops.pack_weights_hl32([cu(w).view(C, 1, 1, C).contiguous() for w in t["wproj"]])
    qkv, _ = ops.conv2d_x3(hl, G, False, rows, 1, 1, C, wq, sq, 3 * C, (1, 1), bias=o(t["bqkv"]))
    ctx = ops.svtr_attention(qkv.view(G * B, N, 3 * C), heads, 32 ** -0.5, o(t["mask"]), want_f32=False, want_hl=True, x3=True)
    br, _ = ops.conv2d_x3(ctx, G, False, rows, 1, 1, C, wp, sp, C, (1, 1), bias=cu(t["bproj"]))
    x2, y2, hl2 = ops.add_layernorm_grouped(xs, br.view(G * B, N, C), cu(t["d1"]), N, cu(t["g2"]), cu(t["b2"]), rows, 1e-6, want_sum=True, want_f32=True)
    assert_close("fused mixer vs unfused chain: residual stream", x_out, x2, atol=4e-6, rtol=2e-6)


@pytest.mark.parametrize("N,G,B,masked,bias,with_pending", [(128, 2, 4, False, True, True), (128, 1, 2, True, True, False), (50, 2, 4, False, False, True),
                                                          (100, 1, 6, True, True, True), (64, 1, 4, False, True, False)])
def test_svtr_fused_attention_block_c256(ops, N, G, B, masked, bias, with_pending):
    """mrn_svtr_attention_block_x3_f32 (stage 3, C = 256: LayerNorm1 -> qkv -> attention in one kernel, the context as the HL32 operand of
    the proj Linear, t = x + drop * pending as the residual stream) against float64 torch and the unfused chain"""
    from mrn_amd.modules.svtr import local_attention_mask
    C, heads = 256, 8
    imgs = G * B
    x, pend = rnd(imgs, N, C, seed=1100) * 1.5, rnd(imgs, N, C, seed=1101) if with_pending else None
    g1, b1 = rnd(G, C, seed=1102) * 0.3 + 1.0, rnd(G, C, seed=1103) * 0.2
    wqkv = [rnd(3 * C, C, seed=1110 + g, scale=(1.0 / C) ** 0.5) * 1.5 for g in range(G)]
    bqkv = rnd(G, 3 * C, seed=1120) * 0.3 if bias else None
    dprev = ((torch.rand(imgs, generator=torch.Generator().manual_seed(9)) > 0.3).float() / 0.7) if with_pending else None
    H = {128: 2, 100: 4, 50: 2, 64: 2}[N]
    mask = local_attention_mask(H, N // H, 7, 11) if masked else None
    t_ref = x.double() + (dprev.double()[:, None, None] * pend.double() if with_pending else 0)
    ctx_ref = []
    for g in range(G):
        y = F.layer_norm(t_ref[g * B:(g + 1) * B], (C,), g1[g].double(), b1[g].double(), 1e-6)
        qkv = y @ wqkv[g].double().t() + (bqkv[g].double() if bias else 0)
        q, k, v = [u.reshape(B, N, heads, 32).permute(0, 2, 1, 3) for u in qkv.split(C, dim=2)]
        s_ = (q @ k.transpose(-1, -2)) * 32 ** -0.5
        if masked:
            s_ = s_ + mask.double()
        ctx_ref.append((torch.softmax(s_, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, C))
    ctx_ref = torch.cat(ctx_ref)
    o = lambda v: cu(v) if v is not None else None
    wq, sq = ops.pack_weights_hl32([cu(w).view(3 * C, 1, 1, C).contiguous() for w in wqkv])
    assert ops.svtr_attention_block_supported(N, C, B, o(mask))
    t, ctx_hl = ops.svtr_attention_block_fused(cu(x), o(pend), o(dprev), cu(g1), cu(b1), 1e-6, wq, sq, o(bqkv), o(mask), 32 ** -0.5, B)
    assert_close("attention block: residual stream", t, t_ref.float(), atol=1e-6, rtol=1e-6)
    assert_close("attention block: context vs float64", _hl32_to_f32(ctx_hl, imgs * N, C).view(imgs, N, C), ctx_ref.float(), atol=2e-5, rtol=1e-5)
    # the unfused chain
    tt, _, hl = ops.add_layernorm_grouped(cu(x), o(pend), o(dprev), N, cu(g1), cu(b1), B * N, 1e-6, want_sum=with_pending)
    qkv, _ = ops.conv2d_x3(hl, G, False, B * N, 1, 1, C, wq, sq, 3 * C, (1, 1), bias=o(bqkv))
    ctx2 = ops.svtr_attention(qkv.view(imgs, N, 3 * C), heads, 32 ** -0.5, o(mask), want_f32=False, want_hl=True, x3=True)
    assert_close("attention block vs unfused chain", _hl32_to_f32(ctx_hl, imgs * N, C), _hl32_to_f32(ctx2, imgs * N, C), atol=4e-6, rtol=2e-6)


def test_svtr_fused_mixer_full_size(ops):
    """the supported shapes (32 x 100 and 32 x 256 crops) at the headline's size (6 experts x 256 images), three launches each: every image must agree with the unfused
    chain (pins the slab ring / K-V tile barriers at full occupancy)"""
    for C, N in ((64, 200), (128, 100), (64, 512), (128, 256)):
        G, B = 6, 256
        heads = C // 32
        t, _ = _mixer_case(C, N, G, B, True, True, True, seed=800 + C)
        o = lambda v: cu(v) if v is not None else None
        rows = B * N
        tt, _, hl = ops.add_layernorm_grouped(cu(t["x"]), o(t["pend"]), o(t["dprev"]), N, cu(t["g1"]), cu(t["b1"]), rows, 1e-6, want_sum=True)
        wq, sq = ops.pack_weights_hl32([cu(w).view(3 * C, 1, 1, C).contiguous() for w in t["wqkv"]])
        wp, sp = ops.pack_weights_hl32([cu(w).view(C, 1, 1, C).contiguous() for w in t["wproj"]])
        qkv, _ = ops.conv2d_x3(hl, G, False, rows, 1, 1, C, wq, sq, 3 * C, (1, 1), bias=o(t["bqkv"]))
        ctx = ops.svtr_attention(qkv.view(G * B, N, 3 * C), heads, 32 ** -0.5, o(t["mask"]), want_f32=False, want_hl=True, x3=True)
        br, _ = ops.conv2d_x3(ctx, G, False, rows, 1, 1, C, wp, sp, C, (1, 1), bias=cu(t["bproj"]))
        x2, _, hl2 = ops.add_layernorm_grouped(tt, br.view(G * B, N, C), cu(t["d1"]), N, cu(t["g2"]), cu(t["b2"]), rows, 1e-6, want_sum=True)
        y2 = _hl32_to_f32(hl2, G * rows, C)
        for _ in range(3):
            x_out, y_hl = _mixer_run(ops, t, C, N, G, B)
            bad = ((x_out - x2).abs().amax((1, 2)) > 1e-4).sum()
            assert int(bad) == 0, (C, int(bad))
            bad = ((_hl32_to_f32(y_hl, G * rows, C) - y2).abs().amax(1) > 2e-4).sum()
            assert int(bad) == 0, (C, int(bad))


@pytest.mark.parametrize("magnitude", [1e-4, 30.0])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(32, 4, 65, 128, 128), (64, 2, 16, 128, 160), (32, 8, 64, 256, 128), (256, 4, 65, 128, 256)])
def test_wgrad_in_the_winograd_domain(B, H, W, Cin, Cout, magnitude):
    """mrn_transpose_oy_wino_hl32_f32 (both operands transformed per group of 4 columns while transposed) + 18 K-windows on
    mrn_gemm_x3_windows_hl32 + mrn_wino_wgrad_finish_f32 against torch's conv weight gradient in float64 and against the 9-window
    form; gradient-sized and large dy (range-scaled operands, factor-16 headroom for the transforms)"""
    from mrn_amd import ops
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn(B, H, W, Cin, generator=g)
    dy = torch.randn(B, H, W, Cout, generator=g) * magnitude
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(),
                                      stride=1, padding=1).permute(0, 2, 3, 1)                 # [Cout,3,3,Cin]
    xd, dyd = x.cuda(), dy.cuda()
    assert ops.wgrad_wino_supported(dyd, xd, (3, 3), (1, 1), (1, 1))
    dw = ops.conv2d_wgrad_x3_wino(dyd, xd)
    scale = ref.abs().max().item()
    assert (dw.cpu().double() - ref).abs().max().item() <= 8e-6 * scale
    if ops.wgrad_windows_supported(dyd, xd, (3, 3), (1, 1), (1, 1)):
        assert (dw - ops.conv2d_wgrad_x3_windows(dyd, xd)).abs().max().item() <= 8e-6 * scale
    assert not ops.wgrad_wino_supported(dyd[:5], xd[:5], (3, 3), (1, 1), (1, 1))                  # 5 * ceil(W/4) groups: not whole lines


# ---------------------------------------------------------------------------------------------------------
# producers of a trained Linear layer's operand (SVTR blocks in loop A): the split operand / its range scale out of the producing pass
# ---------------------------------------------------------------------------------------------------------
def _hl_value(hl, rows, C):
    hv = hl.view(torch.float16).view(rows, C // 32, 2, 32).float()
    return (hv[:, :, 0] + hv[:, :, 1]).reshape(rows, C)


@pytest.mark.parametrize("rows,C,mag", [(517, 64, 1.0), (96, 256, 1e-4), (33, 128, 300.0)])
def test_layernorm_operand_bound_scale(ops, rows, C, mag):
    """mrn_layernorm_fwd_hl32_f32: same y / mean / rstd as the plain kernel (bit for bit), the HL32 operand = s * y to 22 bits, and s the
    largest power of two with s * (sqrt(C) max|gamma| + max|beta|) <= 2^14 -- which bounds max|y|"""
    x = cu(rnd(rows, C, seed=900) * 3 + 0.5)
    gamma, beta = cu((rnd(C, seed=901) + 1.5) * mag), cu(rnd(C, seed=902) * 0.3 * mag)
    y0, m0, r0 = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    y, mean, rstd, hl, sc = ops.layernorm_fwd_operand(x, gamma, beta, 1e-6)
    assert torch.equal(y, y0) and torch.equal(mean, m0) and torch.equal(rstd, r0)
    bound = float(C ** 0.5 * gamma.abs().max() + beta.abs().max())
    s = float(sc[0])
    assert s == 2.0 ** np.floor(np.log2(ops.FP16_WEIGHT_PEAK / bound)) and float(sc[1]) == 1.0 / s
    assert float(y.abs().max()) * s <= ops.FP16_WEIGHT_PEAK
    assert_close("HL32 operand", _hl_value(hl, rows, C) / s, y, atol=float(y.abs().max()) * 3e-7, rtol=3e-7)


def test_elementwise_operand_producers(ops):
    """mrn_ew_operand_f32: gelu / gelu' / DropPath residual against the plain kernels (bit for bit), the HL32 form under a given scale, and
    the folded max|y| -> exact power-of-two scale"""
    B, N, C = 6, 40, 96
    a, b = cu(rnd(B, N, C, seed=910) * 2), cu(rnd(B, N, C, seed=911) * 1e-3)
    drop = cu(torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25, 1.25]))
    sc_in = cu(torch.tensor([2048.0, 1 / 2048.0]))
    y, hl, _ = ops.ew_operand(ops.EW_GELU, a, scale=sc_in, want_hl=True)
    assert torch.equal(y, ops.ew_rows(ops.EW_GELU, a))
    assert_close("gelu HL32 operand", _hl_value(hl, B * N, C) / 2048.0, y.view(-1, C), atol=1e-6, rtol=3e-7)
    g, _, sc = ops.ew_operand(ops.EW_GELU_BWD, a, b, want_amax=ops.FP16_WEIGHT_PEAK)
    assert torch.equal(g, ops.ew_rows(ops.EW_GELU_BWD, a, b))
    assert torch.equal(sc, ops.pow2_scale(g))
    r, _, sc = ops.ew_operand(ops.EW_RESIDUAL_SCALE, a, b, drop=drop, rows_per_drop=N, want_amax=ops.FP16_WEIGHT_PEAK)
    assert torch.equal(r, ops.residual_scale_rows(a, b, drop, N))
    assert torch.equal(sc, ops.pow2_scale(r))
    # a second producer on the same stream starts from cleared words
    _, _, sc2 = ops.ew_operand(ops.EW_GELU_BWD, a, b * 1e-3, want_amax=ops.FP16_WEIGHT_PEAK)
    assert torch.equal(sc2, ops.pow2_scale(ops.ew_rows(ops.EW_GELU_BWD, a, b * 1e-3)))


def test_gemm_epilogue_range_and_attention_operand(ops):
    """mrn_conv2d_x3_hl32 amax_ws: the folded max|y| equals the maximum of the stored result (bias and activation included);
    mrn_svtr_attention_f32 hl_scale: the HL32 form of the result under the scale of max|qkv| (a bound of max|out|)"""
    from mrn_amd import functional as Fn
    R, K, N = 700, 64, 192
    x, w, bias = cu(rnd(R, K, seed=920)), cu(rnd(N, K, seed=921) * 0.2), cu(rnd(N, seed=922))
    y = Fn.x3_linear(x, w, bias, amax_ws=ops._amax_ws())
    sc = ops.pow2_finalize(ops.FP16_WEIGHT_PEAK)
    assert torch.equal(sc, ops.pow2_scale(y))
    assert torch.equal(Fn.x3_linear(x, w, bias), y)
    B, Nt, heads = 3, 72, 2
    qkv = cu(rnd(B, Nt, 3 * 32 * heads, seed=930) * 1.7)
    sq = ops.pow2_scale(qkv)
    out0, lse0 = ops.svtr_attention(qkv, heads, 32 ** -0.5, None, want_lse=True)
    out, lse, hl = ops.svtr_attention(qkv, heads, 32 ** -0.5, None, want_lse=True, want_hl=True, hl_scale=sq)
    assert torch.equal(out, out0) and torch.equal(lse, lse0)
    assert float(out.abs().max()) <= float(qkv.abs().max())
    assert_close("attention HL32 operand", _hl_value(hl, B * Nt, 32 * heads) / float(sq[0]), out.view(-1, 32 * heads), atol=1e-6, rtol=3e-7)
    dout = cu(rnd(B, Nt, 32 * heads, seed=931) * 1e-4)
    d0 = ops.svtr_attention_bwd(qkv, None, out, dout, lse, heads, 32 ** -0.5)
    d1, sc = ops.svtr_attention_bwd(qkv, None, out, dout, lse, heads, 32 ** -0.5, want_range=ops.FP16_WEIGHT_PEAK)
    assert torch.equal(d0, d1) and torch.equal(sc, ops.pow2_scale(d0))


@pytest.mark.parametrize("G,nblk,C", [(3, 7, 64), (2, 5000, 32), (6, 16384, 64), (2, 1100, 96)])
def test_bn_finalize_grouped_chunked(ops, G, nblk, C):
    """mrn_bn_finalize_grouped_f32 over few and over thousands of partial rows (the first layers: 128-pixel blocks of 32 x 256 maps; reduced
    in chunks by several workgroups, the last to arrive combines them in chunk order): scale / shift / running statistics against the
    float64 formula, and twice the same result bit for bit (no atomics in the sums)"""
    torch.manual_seed(nblk)
    count = nblk * 128
    part = torch.rand(G, nblk, 2, C, device="cuda") * 4
    part[:, :, 1] += 40.0                                    # sums of squares dominate: var > 0
    gammas = [torch.rand(C, device="cuda") + 0.5 for _ in range(G)]
    betas = [torch.randn(C, device="cuda") for _ in range(G)]
    outs = []
    for _ in range(2):
        rm = [torch.zeros(C, device="cuda") for _ in range(G)]
        rv = [torch.ones(C, device="cuda") for _ in range(G)]
        table = torch.tensor([[t.data_ptr() for t in ts] for ts in (gammas, betas, rm, rv)], dtype=torch.int64).cuda()
        scale, shift = ops.bn_finalize_grouped(part, G, C, count, table, 0.1, 1e-5)
        torch.cuda.synchronize()
        outs.append((scale.clone(), shift.clone(), torch.stack(rm), torch.stack(rv)))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    s = part[:, :, 0].double().sum(1)
    q = part[:, :, 1].double().sum(1)
    mean = s / count
    var = (q / count - mean * mean).clamp_min(0)
    sc = torch.stack(gammas).double() / torch.sqrt(var + 1e-5)
    assert_close("scale", outs[0][0], sc.float(), atol=1e-6, rtol=2e-6)
    assert_close("shift", outs[0][1], (torch.stack(betas).double() - mean * sc).float(), atol=2e-6, rtol=2e-6)
    assert_close("running_mean", outs[0][2], (0.1 * mean).float(), atol=1e-6, rtol=2e-6)
    assert_close("running_var", outs[0][3], (0.9 + 0.1 * var * count / (count - 1)).float(), atol=1e-6, rtol=2e-6)


def test_multi_pack_linear_matches_one_by_one(ops):
    """every trained Linear weight's operand (and its transpose's) from ONE call == the per-weight max / scale / pack sequence, byte for
    byte: ragged row counts, tiny and huge magnitudes, an all-zero weight (scale 1), a non-finite one (scale 1), a changed set of layers"""
    shapes = [(192, 64), (64, 64), (256, 64), (100, 96), (68, 128), (1024, 256), (256, 1024), (512, 32), (36, 160)]
    ws = []
    for i, (N, K) in enumerate(shapes):
        w = cu(rnd(N, K, seed=40 + i, scale=(1e-6, 1.0, 300.0)[i % 3]))
        ws.append(w)
    ws[3].zero_()
    ws[4][5, 7] = float("inf")
    entries = []
    for w in ws:
        N, K = w.shape
        if K % 32 == 0:
            entries.append(("lin_fwd_x3", w))
        if N % 32 == 0:
            entries.append(("lin_bwd_x3", w))
    assert all(ops.multi_pack_eligible(w, kind) for kind, w in entries)
    assert not ops.multi_pack_eligible(ws[3], "lin_bwd_x3") and not ops.multi_pack_eligible(ws[0][:, :32], "lin_fwd_x3")

    def check(entries):
        got = ops.multi_pack_linear(entries)
        for (kind, w), (hl, sc) in zip(entries, got):
            N, K = w.shape
            src = w.t().contiguous().view(K, 1, 1, N) if kind == "lin_bwd_x3" else w.view(N, 1, 1, K)
            ref_hl, ref_sc = ops.pack_weights_hl32([src])
            assert torch.equal(sc, ref_sc), (kind, tuple(w.shape), sc, ref_sc)
            assert torch.equal(hl, ref_hl), (kind, tuple(w.shape))
    check(entries)
    before = dict(ops.MULTI_PACK_STATS)
    for w in ws[:3]:                           # new values, same table: no rebuild, results follow the weights
        w.mul_(-3.7)
    check(entries)
    assert ops.MULTI_PACK_STATS["rebuilds"] == before["rebuilds"]
    check(entries[3:])                         # another set of layers: the table is rebuilt
    assert ops.MULTI_PACK_STATS["rebuilds"] == before["rebuilds"] + 1


@pytest.mark.parametrize("rows,padded,C,S,mag", [(4096, 4096, 64, 8, 1.0), (8192, 8192, 384, 4, 1e-5), (1000, 1024, 100, 1, 300.0),
                                                  (32, 32, 8, 1, 1.0), (65536, 65536, 64, 64, 1.0), (16384, 16384, 1024, 2, 1e-3)])
def test_transposed_split_with_column_sums(ops, rows, padded, C, S, mag):
    """the weight gradient's transposed operand of dy and the bias gradient from ONE pass: same bytes as the plain pass, column sums equal
    to the fp64 sums to fp32 rounding, bit-identical across runs, `accumulate` adds"""
    x = cu(rnd(rows, C, seed=5, scale=mag))
    sc = ops.pow2_scale(x)
    ref = ops.split_hl32_t(x, S, sc, rows_padded=padded)
    out = torch.full((C,), 7.0, device="cuda")
    got = ops.split_hl32_t(x, S, sc, rows_padded=padded, colsum_out=out)
    assert torch.equal(got, ref)
    want = x.double().sum(0)
    err = (out.double() - want).abs().max().item()
    assert err <= 2e-6 * mag * rows ** 0.5 + 1e-30, (err, mag)
    out2 = out.clone()
    again = ops.split_hl32_t(x, S, sc, rows_padded=padded, colsum_out=out2, accumulate=True)
    assert torch.equal(again, ref) and torch.equal(out2, out + out)
    out3 = torch.empty_like(out)
    ops.split_hl32_t(x, S, sc, rows_padded=padded, colsum_out=out3)
    assert torch.equal(out3, out)
    unscaled = torch.empty_like(out)                       # no prescale: plain sums
    assert torch.equal(ops.split_hl32_t(x, S, None, rows_padded=padded, colsum_out=unscaled), ops.split_hl32_t(x, S, None, rows_padded=padded))
    assert (unscaled.double() - want).abs().max().item() <= 2e-6 * mag * rows ** 0.5 + 1e-30


def test_lstm_grouped_tail_sets_spread_over_all_xcds(ops):
    """the (expert, direction, tile) workgroups of a grouped layer are dealt to the eight XCDs in equal set-major runs (rnn.hip: pinned == 2;
    twelve sets of 16 tiles = 24 per XCD) -- every (set, tile) is computed exactly once: the grouped launch equals the per-expert launches
    bit for bit, also with a ragged last tile (B = 250), ten sets, and eighteen sets of four tiles"""
    Hd, T = 256, 7
    for G, B in ((6, 256), (5, 256), (6, 250), (9, 64)):
        xproj = cu(rnd(G, B, T, 8 * Hd, seed=300 + G, scale=0.7))
        ws = [[cu(rnd(4 * Hd, Hd, seed=301 + 2 * g + d, scale=1 / 16.0)) for d in range(2)] for g in range(G)]
        b_hh = cu(rnd(G, 8 * Hd, seed=350, scale=1 / 16.0))
        packs = [[ops.pack_fragment_major_h(w) for w in p] for p in ws]
        w_h = torch.stack([torch.stack([d[0] for d in p]) for p in packs]).contiguous()
        w_inv = torch.stack([torch.cat([d[1] for d in p]) for p in packs]).contiguous()
        out = ops.lstm_layer_x3_grouped(xproj, w_h, w_inv, b_hh, Hd, 2)
        for g in range(G):
            one = ops.lstm_layer_x3_grouped(xproj[g:g + 1].contiguous(), w_h[g:g + 1].contiguous(), w_inv[g:g + 1].contiguous(),
                                            b_hh[g:g + 1].contiguous(), Hd, 2)
            assert torch.equal(out[g], one[0]), (G, B, g)
