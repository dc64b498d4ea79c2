"""Model-level parity on the GPU: the HIP-backed containers (mrn_amd.modules.model) against
 (a) golden vectors produced by the reference itself and (b) the CPU oracle on the same inputs.
Tolerance 1e-4 on features / logits / losses (north star); integer outputs (argmax, routing index, CTC strings)
bit-exact."""
import contextlib
import io
import os
import types

import numpy as np
import pytest
import torch

from tests.helpers import (assert_close, assert_sub_close, assert_sub_l2, det_inputs, drop_masks, golden_state_dict,
                           load_golden)

pytestmark = pytest.mark.gpu

CASES = {"crnn_mrn3": ("crnn", (40, 70, 97), 2, 1), "trba_mrn3": ("trba", (41, 71, 98), 2, 2),
         "svtr_mrn3": ("svtr", (40, 70, 97), 2, 3),
         # the same cases on U(-1,1) white-noise crops, the distribution bench.py runs (no TPS stage: the 1e-4 band holds;
         # TRBA on noise: test_trba_noise_inside_reference_band)
         "crnn_mrn3_noise": ("crnn", (40, 70, 97), 2, 1), "svtr_mrn3_noise": ("svtr", (40, 70, 97), 2, 3)}


def set_drop_masks(net, kind, B, seed, tag, experts):
    """pin the DropPath draws of the given SVTR experts to the ones the golden generator injected into the reference"""
    if kind != "svtr":
        return
    from mrn_amd.modules.svtr import DropPath
    for e, masks in zip(experts, drop_masks(B, seed, tag, len(experts))):
        mods = [m for m in net.model[e].modules() if isinstance(m, DropPath)]      # forward order, 11 per expert
        assert len(mods) == 11
        for j, m in enumerate(mods):
            m.forced_masks = [masks[2 * j], masks[2 * j + 1]]


def make_opt(kind):
    o = types.SimpleNamespace(num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512, hidden_size=256,
                              batch_max_length=25)
    if kind == "crnn":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "VGG", "BiLSTM", "CTC"
    elif kind == "svtr":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "SVTR", "None", "CTC"
    else:
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "TPS", "ResNet", "BiLSTM", "Attn"
    return o


def build_net(kind, classes, g, seed):
    from mrn_amd.modules.model import MRNNet
    opt = make_opt(kind)
    with contextlib.redirect_stdout(io.StringIO()):
        net = MRNNet(opt)
        for c in classes:
            net.update_fc(opt.hidden_size, c)
            net.build_prediction(opt, c)
    net.load_state_dict(golden_state_dict(g, seed), strict=True)
    return opt, net.cuda()


def reload(net, g, seed):
    net.load_state_dict(golden_state_dict(g, seed), strict=True)


def labels_for(kind, words, chars):
    from mrn_amd.tools.utils import AttnLabelConverter, CTCLabelConverter
    with contextlib.redirect_stdout(io.StringIO()):
        conv = CTCLabelConverter(chars) if kind != "trba" else AttnLabelConverter(chars)
    idx, ln = conv.encode(words, batch_max_length=25)
    return conv, idx, ln


@pytest.mark.parametrize("name", list(CASES))
def test_expert_forward_vs_golden(name):
    kind, classes, B, seed = CASES[name]
    g = load_golden(name)
    opt, net = build_net(kind, classes, g, seed)
    image, words, chars, _ = det_inputs(kind, classes, B, seed, noise=name.endswith("_noise"))
    conv, labels_index, labels_length = labels_for(kind, words, chars)
    assert np.array_equal(labels_index.cpu().numpy(), g["labels_index"])
    image = image.cuda()
    ctc = kind != "trba"
    text = None if ctc else labels_index[:, :-1].cuda()
    m0 = net.model[0]
    with torch.no_grad():
        net.train()
        if kind == "trba":
            out, cp, _ = m0.model.Transformation(image, return_aux=True)
            assert_close("cprime", cp, g["e0/tps_cprime"], atol=2e-5)
            assert_sub_close(g, "e0/tps_out", out, atol=5e-4)      # fp32 conditioning of the TPS grid, see weights.smooth_image
            reload(net, g, seed)
        set_drop_masks(net, kind, B, seed, "e0", [0])
        o = m0(image, text, True)
        assert_sub_close(g, "e0/feature", o["feature"], atol=1e-4)
        assert_sub_close(g, "e0/predict", o["predict"], atol=1e-4)
        # BatchNorm running statistics moved exactly like the reference's
        sd = net.state_dict()
        rm = [v for k, v in sd.items() if k.startswith("model.0.") and k.endswith("running_mean")
              and tuple(v.shape) == g["e0/bn_running_mean_after"].shape]
        assert any(np.abs(r.cpu().numpy() - g["e0/bn_running_mean_after"]).max() < 1e-5 for r in rm)
        reload(net, g, seed)
        net.eval()
        sos = None if ctc else torch.LongTensor(B).fill_(2).cuda()
        o = m0(image, sos, False)
        assert_sub_close(g, "e0_eval/feature", o["feature"], atol=1e-4)
        assert_sub_close(g, "e0_eval/predict", o["predict"], atol=1e-4)
        assert np.array_equal(o["predict"].max(2)[1].cpu().numpy(), g["e0_eval/argmax"])
        oe = net(image, True, sos, False)
        assert np.array_equal(oe["index"].cpu().numpy(), g["eval/index"])
        assert_sub_close(g, "eval/logits", oe["logits"], atol=1e-4)
        am = oe["logits"].max(2)[1].cpu().numpy()
        assert np.array_equal(am, g["eval/argmax"])
        if ctc:
            assert conv.decode(am, [am.shape[1]] * B) == [str(s) for s in g["eval/ctc_strings"]]
        # loop A forward + loss
        reload(net, g, seed)
        net.train()
        from mrn_amd import functional as Fn
        set_drop_masks(net, kind, B, seed, "stepA", [len(classes) - 1])
        if ctc:
            preds = net(image, False)["logits"]
            loss = Fn.ctc_loss(preds.contiguous(), labels_index.cuda(), labels_length.cuda())
        else:
            preds = net(image, False, labels_index[:, :-1].cuda())["logits"]
            loss = Fn.cross_entropy(preds, labels_index[:, 1:].cuda(), 1)
        assert_sub_close(g, "stepA/logits", preds, atol=1e-4)
        assert abs(loss.item() - float(g["stepA/loss"])) < 1e-4 * max(1.0, abs(float(g["stepA/loss"])))


@pytest.mark.parametrize("precision,locnet,tol", [("f32", None, 1e-4), ("fp16x3", None, 1e-4)])
def test_loop_b_forward_other_conv_precisions(precision, locnet, tol):
    """The default conv arithmetic is "auto" = split-fp16x3 on every eligible conv (covered by every other test at
    1e-4).  This pins the alternatives: exact fp32 everywhere (the per-expert exact-fp32 MFMA kernel, 1e-4) and fp16x3 forced
    (1e-4).  (The split-bf16 x3 mode of rounds 1-3 -- 16-bit products, 5e-4 -- left the library in round 4.)"""
    from mrn_amd import ops
    kind, classes, B, seed = CASES["trba_mrn3"]
    g = load_golden("trba_mrn3")
    opt, net = build_net(kind, classes, g, seed)
    image, words, chars, _ = det_inputs(kind, classes, B, seed)
    conv, labels_index, _ = labels_for(kind, words, chars)
    old, old_loc = ops.CONV_PRECISION, ops.LOCNET_CONV_PRECISION
    ops.CONV_PRECISION, ops.LOCNET_CONV_PRECISION = precision, locnet
    try:
        net.train()
        with torch.no_grad():
            out = net(image.cuda(), True, labels_index[:, :-1].cuda(), True)
        assert_close("weights", out["index"], g["stepB/weights"], atol=tol, rtol=tol)
        assert_sub_close(g, "stepB/logits", out["logits"], atol=tol, rtol=tol)
        assert np.array_equal(out["index"].argmax(1).cpu().numpy(), g["stepB/weights"].argmax(1))
    finally:
        ops.CONV_PRECISION, ops.LOCNET_CONV_PRECISION = old, old_loc


@pytest.mark.parametrize("name", list(CASES))
def test_loop_b_two_steps_vs_golden(name):
    """il_modules/mrn.py:323-371 -- weights, fused logits, losses, clipped gradients, 2-step Adam deltas"""
    from mrn_amd import functional as Fn
    from mrn_amd.optim import FlatAdam, OneCycle
    kind, classes, B, seed = CASES[name]
    g = load_golden(name)
    opt, net = build_net(kind, classes, g, seed)
    image, words, chars, domain = det_inputs(kind, classes, B, seed, noise=name.endswith("_noise"))
    conv, labels_index, labels_length = labels_for(kind, words, chars)
    image, domain, labels_index, labels_length = image.cuda(), domain.cuda(), labels_index.cuda(), labels_length.cuda()
    net.train()
    for i in range(len(classes)):
        for p in net.model[i].parameters():
            p.requires_grad = False
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    assert names == [str(n) for n in g["router_param_names"]]
    params = [p for p in net.parameters() if p.requires_grad]
    before = [p.detach().clone() for p in params]
    adam = FlatAdam(params, lr=0.0005)
    sched = OneCycle(0.0005, 40)
    for it in range(2):
        set_drop_masks(net, kind, B, seed, f"stepB{it}", list(range(len(classes))))
        if kind != "trba":
            out = net(image, True)
            clf = Fn.ctc_loss(out["logits"], labels_index, labels_length)
        else:
            out = net(image, cross=True, text=labels_index[:, :-1], is_train=True)
            clf = Fn.cross_entropy(out["logits"], labels_index[:, 1:], 1)
        taski = Fn.cross_entropy(out["index"], domain, -100)
        loss = 15 * clf + taski
        adam.zero_grad()
        loss.backward()
        nc = adam.step(lr=sched.lr_at(it), max_norm=5.0)
        if it == 0:
            assert_close("weights", out["index"], g["stepB/weights"], atol=1e-4)
            assert_sub_close(g, "stepB/logits", out["logits"], atol=1e-4)
            assert abs(clf.item() - float(g["stepB/loss_clf"])) < 1e-4 * max(1.0, abs(float(g["stepB/loss_clf"])))
            assert abs(taski.item() - float(g["stepB/loss_taski"])) < 1e-4
            assert abs(nc[0].item() - float(g["stepB/grad_norm"])) <= 1e-3 * float(g["stepB/grad_norm"])
            for n, p in zip(names, params):
                if float(g[f"stepB/grad/{n}/absmean"]) < 1e-6:
                    continue
                assert_sub_close(g, f"stepB/grad/{n}", p.grad, atol=1e-6, rtol=2e-3)
    assert abs(clf.item() - float(g["stepB/loss_clf_1"])) < 1e-4 * max(1.0, abs(float(g["stepB/loss_clf_1"])))
    for n, p, b in zip(names, params, before):
        if float(g[f"stepB/grad/{n}/absmean"]) < 1e-6:
            continue          # route.bias: shift-invariant under softmax, gradient is round-off noise
        # two Adam steps move every element by <= ~5e-5 (2 x lr); where a gradient element is near zero the
        # normalised update m/sqrt(v) has an essentially random sign, so compare in L2 / quantile terms
        assert_sub_l2(g, f"stepB/delta2/{n}", p.detach() - b)


def test_trba_noise_inside_reference_band():
    """TRBA x 3 loop-B forward on U(-1,1) white noise (bench.py's input distribution) against the reference's own output AND
    against exact arithmetic.  The TPS grid is an ill-conditioned fp32 sum (the CPU side of this claim is
    tests/test_oracle_golden.py::test_trba_tps_conditioning_smooth_vs_noise): on noise the REFERENCE's fp32 result is `band`
    (several 1e-3) away from the float64 result, so no fp32 implementation can be asked for 1e-4 against it.  Required here:
    the HIP path is as close to exact arithmetic as the reference is (within 2x its band), therefore within 3x band of the
    reference; the routing argmax is bit-exact (the float64 top-2 margin is far above the band)."""
    from oracle import mrn_oracle as O
    from tests.helpers import oracle_dtype, sub
    kind, classes, B, seed = "trba", (41, 71, 98), 2, 2
    g = load_golden("trba_mrn3_noise")
    opt, net = build_net(kind, classes, g, seed)
    image, words, chars, _ = det_inputs(kind, classes, B, seed, noise=True)
    conv, labels_index, _ = labels_for(kind, words, chars)
    with oracle_dtype(torch.float64) as od, torch.no_grad():
        out64 = O.mrn_forward(od.cast(golden_state_dict(g, seed)), O.Cfg("TPS", "ResNet", "BiLSTM", "Attn"), 3, image.double(), True,
                              labels_index.cpu()[:, :-1], True, training=True)
    w64, l64 = out64["index"].numpy(), sub(out64["logits"])[0].astype(np.float64)
    band_w = np.abs(g["stepB/weights"] - w64).max()
    band_l = np.abs(g["stepB/logits/sub"] - l64).max()
    assert band_w > 1e-3 and band_l > 1e-3           # the premise: the reference itself is > 1e-3 from exact arithmetic here
    net.train()
    with torch.no_grad():
        out = net(image.cuda(), True, labels_index[:, :-1].cuda(), True)
    w = out["index"].cpu().double().numpy()
    lg = sub(out["logits"])[0].astype(np.float64)
    assert np.abs(w - w64).max() <= 2 * band_w, (np.abs(w - w64).max(), band_w)
    assert np.abs(lg - l64).max() <= 2 * band_l, (np.abs(lg - l64).max(), band_l)
    assert np.abs(w - g["stepB/weights"]).max() <= 3 * band_w and np.abs(lg - g["stepB/logits/sub"]).max() <= 3 * band_l
    top2 = np.sort(w64, axis=1)[:, ::-1]
    assert (top2[:, 0] - top2[:, 1]).min() > 10 * band_w
    assert np.array_equal(w.argmax(1), g["stepB/weights"].argmax(1)) and np.array_equal(w.argmax(1), w64.argmax(1))
    # eval routing (hard argmax + per-sample expert selection + greedy decode): indices bit-exact against the reference
    net.load_state_dict(golden_state_dict(g, seed), strict=True)
    net.eval()
    with torch.no_grad():
        oe = net(image.cuda(), True, torch.LongTensor(B).fill_(2).cuda(), False)
    assert np.array_equal(oe["index"].cpu().numpy(), g["eval/index"])
    assert np.array_equal(oe["logits"].max(2)[1].cpu().numpy(), g["eval/argmax"])


def test_full_size_trba6_loop_b_properties():
    """The headline workload itself (bench.py: TRBA x 6 experts, 256 crops, class counts 2091...5374) as a test: the production
    schedule -- six experts in two lock-step half-groups on two HIP streams, split-fp16 x3 convolutions, the frozen-expert
    forward issued ahead through MRNNet.experts_prefetch -- against the plainest one: expert after expert, exact-fp32 MFMA
    everywhere.  Smooth crops: tight agreement; U(-1,1) noise (the bench's distribution): agreement inside the TPS conditioning
    band.  Size-independent properties: routing weights row-stochastic, prefetched == direct bit for bit, BatchNorm running
    statistics advanced identically, one loop-B optimiser step moves router parameters only."""
    from mrn_amd import functional as Fn
    from mrn_amd import ops
    from mrn_amd.modules.model import MRNNet
    from mrn_amd.optim import FlatAdam
    from mrn_amd.tools import weights as W
    opt = make_opt("trba")
    classes = (2091, 2311, 4039, 5199, 5272, 5374)
    B = 256

    def build():
        with contextlib.redirect_stdout(io.StringIO()):
            net = MRNNet(opt)
            for c in classes:
                net.update_fc(256, c)
                net.build_prediction(opt, c)
        W.fill_state_dict(net.state_dict(), seed=31)
        net = net.cuda().train()
        for e in net.model:
            for p in e.parameters():
                p.requires_grad = False
        return net

    text = torch.from_numpy(W.randint("full_text", (B, 27), 4, classes[-1], 3)).cuda()
    text[:, 0] = 2
    for kind, tol_w, tol_l in (("smooth", 5e-4, 2e-3), ("noise", 3e-2, 1e-1)):
        if kind == "smooth":
            image = torch.from_numpy(W.smooth_image("full_trba", (B, 4, 32, 256), 3)).cuda()
        else:
            image = torch.from_numpy(W.uniform("full_trba", (B, 4, 32, 256), -1.0, 1.0, 3)).cuda()
        net = build()
        with torch.no_grad():
            handle = net.experts_prefetch(image, text[:, :-1], True)
            assert handle is not None and len(handle["parts"]) == 1          # six experts: ONE lock-step group on one side stream (the default)
            fast = net(image, True, text[:, :-1], True, experts=handle)
        torch.cuda.synchronize()
        bn_fast = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k}
        w_fast, l_fast = fast["index"].clone(), fast["logits"].clone()
        assert_close("routing weights sum to 1", w_fast.sum(1), torch.ones(B), atol=1e-5)
        assert torch.isfinite(l_fast).all()
        # the same schedule without the look-ahead: bit-identical
        net2 = build()
        with torch.no_grad():
            direct = net2(image, True, text[:, :-1], True)
        assert torch.equal(direct["index"], w_fast) and torch.equal(direct["logits"], l_fast)
        del net2
        # plain schedule, exact fp32
        net3 = build()
        net3.expert_grouping = False
        net3.expert_streams = False
        old = ops.CONV_PRECISION
        try:
            ops.CONV_PRECISION = "f32"
            with torch.no_grad():
                plain = net3(image, True, text[:, :-1], True)
        finally:
            ops.CONV_PRECISION = old
        torch.cuda.synchronize()
        dw = (plain["index"] - w_fast).abs()
        dl = (plain["logits"] - l_fast).abs()
        assert float(dw.max()) <= tol_w and float(dl.max()) <= tol_l, (kind, float(dw.max()), float(dl.max()))
        assert float(dw.mean()) <= tol_w / 10 and float(dl.mean()) <= tol_l / 20, (kind, float(dw.mean()), float(dl.mean()))
        agree = float((plain["index"].argmax(1) == w_fast.argmax(1)).float().mean())
        assert agree >= (1.0 if kind == "smooth" else 0.97), (kind, agree)
        for k, v in net3.state_dict().items():
            if "running_" in k:
                assert_close(k, bn_fast[k], v, atol=1e-5 if kind == "smooth" else 1e-3, rtol=1e-3)
        del net3
        if kind == "noise":
            # one full loop-B step (il_modules/mrn.py:329-371) at full size
            for n, p in net.named_parameters():
                p.requires_grad = not n.startswith("model.")
            before = {n: p.detach().clone() for n, p in net.named_parameters()}
            adam = FlatAdam([p for p in net.parameters() if p.requires_grad], lr=5e-4)
            adam.zero_grad()
            out = net(image, True, text[:, :-1], True)
            domain = torch.from_numpy(W.randint("full_dom", (B,), 0, 2, 3)).cuda()
            loss = 15 * Fn.cross_entropy(out["logits"], text[:, 1:], 1) + Fn.cross_entropy(out["index"], domain, -100)
            loss.backward()
            nc = adam.step(lr=2.5e-5, max_norm=5.0)
            assert torch.isfinite(loss).item() and float(nc[0]) > 0
            for n, p in net.named_parameters():
                moved = not torch.equal(before[n], p.detach())
                assert moved == (not n.startswith("model.") and n != "route.bias" or (n == "route.bias" and moved)), n
        del net


def test_trba6_batch32_full_class_counts_vs_oracle():
    """The headline configuration against the CPU ORACLE (not against another HIP schedule): TRBA x 6 experts with the bench's
    class counts 2091 ... 5374, 32 crops, the production schedule (one lock-step group of the six experts on a side stream, Winograd F(4,3) +
    split-fp16 x3 convolutions).  Smooth crops: routing weights and fused logits within 1e-4 of the fp32 oracle -- or, where the
    fp32 oracle itself is further than that from float64 arithmetic (the TPS grid's conditioning), within 3x that band -- routing
    argmax and eval routing / greedy indices bit-exact, the gradients of all router tensors of one loop-B step within 2e-3.
    U(-1,1) noise (bench.py's distribution): inside the conditioning band, indices bit-exact where the float64 margin clears it."""
    from mrn_amd import functional as Fn
    from mrn_amd.modules.model import MRNNet
    from mrn_amd.tools import weights as W
    from oracle import mrn_oracle as O
    from tests.helpers import oracle_dtype
    opt = make_opt("trba")
    classes = (2091, 2311, 4039, 5199, 5272, 5374)
    B, I = 32, len(classes)
    with contextlib.redirect_stdout(io.StringIO()):
        net = MRNNet(opt)
        for c in classes:
            net.update_fc(256, c)
            net.build_prediction(opt, c)
    W.fill_state_dict(net.state_dict(), seed=37)          # (the deterministic fill of 314 M parameters dominates the test's time: one
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}      #  build serves both crop kinds)
    net = net.cuda()
    for n, p in net.named_parameters():
        p.requires_grad = not n.startswith("model.")
    for crops in ("smooth", "noise"):
        net.load_state_dict(sd, strict=True)
        net.train()
        net.zero_grad(set_to_none=True)
        _trba6_b32_case(net, sd, crops, classes, B, I)


def _trba6_b32_case(net, sd, crops, classes, B, I):
    from mrn_amd import functional as Fn
    from mrn_amd.tools import weights as W
    from oracle import mrn_oracle as O
    from tests.helpers import oracle_dtype
    if crops == "smooth":
        image = torch.from_numpy(W.smooth_image("b32_trba", (B, 4, 32, 256), 5))
    else:
        image = torch.from_numpy(W.uniform("b32_trba", (B, 4, 32, 256), -1.0, 1.0, 5))
    text = torch.from_numpy(W.randint("b32_text", (B, 27), 4, classes[-1], 5))
    text[:, 0] = 2
    domain = torch.from_numpy(W.randint("b32_dom", (B,), 0, 2, 5))
    cfg = O.Cfg("TPS", "ResNet", "BiLSTM", "Attn")
    # ---- oracle: fp32 (the reference's arithmetic) with autograd through the router, and float64 (the conditioning yardstick)
    names = [n for n in sd if not n.startswith("model.") and sd[n].is_floating_point()]
    sd32 = {k: v.clone() for k, v in sd.items()}
    for n in names:
        sd32[n].requires_grad_(True)
    out32 = O.mrn_forward(sd32, cfg, I, image, True, text[:, :-1], True, training=True)
    loss32 = 15 * O.attn_ce_loss(out32["logits"], text) + torch.nn.functional.cross_entropy(out32["index"], domain)
    g32 = torch.autograd.grad(loss32, [sd32[n] for n in names])
    with oracle_dtype(torch.float64) as od, torch.no_grad():
        out64 = O.mrn_forward(od.cast(sd), cfg, I, image.double(), True, text[:, :-1], True, training=True)
    w32, l32 = out32["index"].detach(), out32["logits"].detach()
    band_w = float((w32.double() - out64["index"]).abs().max())
    band_l = float((l32.double() - out64["logits"]).abs().max())
    # ---- HIP, production schedule
    handle = None
    with torch.no_grad():
        handle = net.experts_prefetch(image.cuda(), text[:, :-1].cuda(), True)
    assert handle is not None and len(handle["parts"]) == 1
    out = net(image.cuda(), True, text[:, :-1].cuda(), True, experts=handle)
    loss = 15 * Fn.cross_entropy(out["logits"], text[:, 1:].cuda(), 1) + Fn.cross_entropy(out["index"], domain.cuda(), -100)
    loss.backward()
    w, lg = out["index"].detach().cpu(), out["logits"].detach().cpu()
    ew, el = float((w - w32).abs().max()), float((lg - l32).abs().max())
    ew64, el64 = float((w.double() - out64["index"]).abs().max()), float((lg.double() - out64["logits"]).abs().max())
    scale_l = float(l32.abs().max())
    if crops == "smooth":
        assert ew <= max(1e-4, 3 * band_w), (ew, band_w)
        assert el <= max(1e-4 * max(1.0, scale_l), 3 * band_l), (el, band_l, scale_l)
    else:
        assert band_w > 5e-4           # the premise on noise: the fp32 reference arithmetic itself is far from exact
        assert ew64 <= 2 * band_w and el64 <= 2 * band_l, (ew64, band_w, el64, band_l)
        assert ew <= 3 * band_w and el <= 3 * band_l, (ew, band_w, el, band_l)
    assert abs(float(loss.detach()) - float(loss32.detach())) <= (1e-4 if crops == "smooth" else 3 * band_l) * max(1.0, abs(float(loss32.detach())))
    # routing argmax: bit-exact on every sample whose float64 top-2 margin clears the band
    top2 = out64["index"].sort(1, descending=True)[0]
    clear = (top2[:, 0] - top2[:, 1]) > 10 * max(band_w, 1e-5)
    assert int(clear.sum()) >= (B - 4 if crops == "smooth" else B // 2)
    assert torch.equal(w.argmax(1)[clear], w32.argmax(1)[clear]) and torch.equal(w.argmax(1)[clear], out64["index"].argmax(1)[clear])
    # one loop-B step's gradients of every router tensor
    mine = dict(net.named_parameters())
    for n, gr in zip(names, g32):
        if float(gr.abs().mean()) < 1e-7:
            continue          # route.bias: shift-invariant under softmax, its gradient is round-off noise
        tol = 2e-3 if crops == "smooth" else max(2e-3, 30 * band_w)
        _grad_check(n, mine[n].grad, gr, rel_l2=tol, rel_max=5 * tol)
    # ---- eval routing + greedy decoding (test.py:validation's model call): integer outputs
    with torch.no_grad():      # (sd32: its BatchNorm running statistics took the same one train-mode update as the HIP modules')
        oe32 = O.mrn_forward({k: v.detach() for k, v in sd32.items()}, cfg, I, image, True, torch.LongTensor(B).fill_(2), False,
                             training=False)
    net.eval()
    with torch.no_grad():
        oe = net(image.cuda(), True, torch.LongTensor(B).fill_(2).cuda(), False)
    if crops == "smooth":
        assert torch.equal(oe["index"].cpu(), oe32["index"]), (oe["index"].cpu(), oe32["index"])
        am, am32 = oe["logits"].max(2)[1].cpu(), oe32["logits"].max(2)[1]
        assert torch.equal(am, am32), int((am != am32).sum())
    else:       # on noise a sub-band eval margin may flip a routing decision: demand agreement on a clear majority and exact
        same = oe["index"].cpu() == oe32["index"]          # greedy strings wherever the routing agrees
        assert float(same.float().mean()) >= 0.9
        am, am32 = oe["logits"].max(2)[1].cpu(), oe32["logits"].max(2)[1]
        assert float((am[same] == am32[same]).float().mean()) >= 0.99


def _ctc_family_b32_case(kind, classes, crops, seed):
    """CRNN / SVTR + MRN at the bench's class counts, 32 crops, production lock-step schedule, against the CPU oracle in fp32 (the
    reference's arithmetic, autograd through the router) and float64 (the conditioning yardstick).  No TPS stage in these families, so
    the plain 1e-4 band of north_star has to hold on U(-1,1) noise as well as on smooth crops; indices bit-exact."""
    from mrn_amd import functional as Fn
    from mrn_amd.modules.model import MRNNet
    from mrn_amd.tools import weights as W
    from oracle import mrn_oracle as O
    from tests.helpers import oracle_dtype
    opt = make_opt(kind)
    B, I = 32, len(classes)
    with contextlib.redirect_stdout(io.StringIO()):
        net = MRNNet(opt)
        for c in classes:
            net.update_fc(256, c)
            net.build_prediction(opt, c)
    W.fill_state_dict(net.state_dict(), seed=seed)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().train()
    for n, p in net.named_parameters():
        p.requires_grad = not n.startswith("model.")
    if crops == "smooth":
        image = torch.from_numpy(W.smooth_image(f"b32_{kind}", (B, 4, 32, 256), seed))
    else:
        image = torch.from_numpy(W.uniform(f"b32_{kind}", (B, 4, 32, 256), -1.0, 1.0, seed))
    # CTC targets as CTCLabelConverter.encode lays them out (tools/utils.py:35-58): [B,25] long, [PAD] = 1 behind the label
    lens = torch.from_numpy(W.randint("b32_len", (B,), 1, 26, seed)).int()
    labels = torch.from_numpy(W.randint("b32_ctc", (B, 25), 4, classes[-1], seed))
    labels[torch.arange(25)[None, :] >= lens[:, None]] = 1
    lens[0] = 25
    labels[0] = torch.from_numpy(W.randint("b32_ctc_full", (25,), 4, classes[-1], seed))       # a full-length label
    domain = torch.from_numpy(W.randint("b32_dom", (B,), 0, 2, seed))
    cfg = O.Cfg("None", "VGG" if kind == "crnn" else "SVTR", "BiLSTM" if kind == "crnn" else "None", "CTC")
    masks = drop_masks(B, seed, "b32", I) if kind == "svtr" else None
    names = [n for n in sd if not n.startswith("model.") and sd[n].is_floating_point()]
    sd32 = {k: v.clone() for k, v in sd.items()}
    for n in names:
        sd32[n].requires_grad_(True)
    out32 = O.mrn_forward(sd32, cfg, I, image, True, None, True, training=True, masks=[[m.clone() for m in ms] for ms in masks] if masks else None)
    clf32 = O.ctc_loss(out32["logits"], labels, lens)
    loss32 = 15 * clf32 + torch.nn.functional.cross_entropy(out32["index"], domain)
    g32 = torch.autograd.grad(loss32, [sd32[n] for n in names])
    with oracle_dtype(torch.float64) as od, torch.no_grad():
        out64 = O.mrn_forward(od.cast(sd), cfg, I, image.double(), True, None, True, training=True,
                              masks=[[m.double() for m in ms] for ms in masks] if masks else None)
    w32, l32 = out32["index"].detach(), out32["logits"].detach()
    band_w = float((w32.double() - out64["index"]).abs().max())
    band_l = float((l32.double() - out64["logits"]).abs().max())
    # ---- HIP, production schedule: the lock-step group(s) issued ahead on the side stream(s), as the learner's pipeline does
    set_drop_masks_from(net, masks)
    with torch.no_grad():
        handle = net.experts_prefetch(image.cuda(), None, True)
    assert handle is not None
    out = net(image.cuda(), True, experts=handle)
    clf = Fn.ctc_loss(out["logits"], labels.cuda(), lens.cuda())
    loss = 15 * clf + Fn.cross_entropy(out["index"], domain.cuda(), -100)
    loss.backward()
    w, lg = out["index"].detach().cpu(), out["logits"].detach().cpu()
    ew, el = float((w - w32).abs().max()), float((lg - l32).abs().max())
    scale_l = float(l32.abs().max())
    assert band_w < 1e-4 and band_l < 1e-4 * max(1.0, scale_l), (band_w, band_l)       # the premise: no ill-conditioned stage here
    assert ew <= 1e-4, (ew, band_w)
    assert el <= 1e-4 * max(1.0, scale_l), (el, band_l, scale_l)
    assert abs(float(clf.detach()) - float(clf32.detach())) <= 1e-4 * max(1.0, abs(float(clf32.detach())))
    assert abs(float(loss.detach()) - float(loss32.detach())) <= 1e-4 * max(1.0, abs(float(loss32.detach())))
    top2 = out64["index"].sort(1, descending=True)[0]
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert int(clear.sum()) >= B - 4
    assert torch.equal(w.argmax(1)[clear], w32.argmax(1)[clear]) and torch.equal(w.argmax(1)[clear], out64["index"].argmax(1)[clear])
    # train-mode greedy CTC path (argmax over classes per frame): bit-exact wherever the float64 top-2 logit margin clears 1e-4
    t2 = out64["logits"].topk(2, dim=2)[0]
    clear_l = (t2[..., 0] - t2[..., 1]) > 2e-4 * max(1.0, scale_l)
    assert float(clear_l.float().mean()) > 0.9
    assert torch.equal(lg.argmax(2)[clear_l], l32.argmax(2)[clear_l])
    mine = dict(net.named_parameters())
    for n, gr in zip(names, g32):
        if n == "route.bias":
            assert float(gr.abs().max()) < 1e-5 and float(mine[n].grad.abs().max()) < 1e-5      # shift-invariant under softmax: round-off noise
            continue
        _grad_check(n, mine[n].grad, gr, rel_l2=2e-3, rel_max=1e-2)
    # ---- eval routing + greedy CTC strings (test.py:validation's model call + preds.max(2) + converter.decode): integers, bit-exact
    with torch.no_grad():
        oe32 = O.mrn_forward({k: v.detach() for k, v in sd32.items()}, cfg, I, image, True, None, False, training=False)
    net.eval()
    with torch.no_grad():
        oe = net(image.cuda(), True, None, False)
    assert torch.equal(oe["index"].cpu(), oe32["index"]), (oe["index"].cpu(), oe32["index"])
    am, am32 = oe["logits"].max(2)[1].cpu(), oe32["logits"].max(2)[1]
    assert torch.equal(am, am32), int((am != am32).sum())
    from mrn_amd.data.synthetic import synthetic_characters
    from mrn_amd.tools.utils import CTCLabelConverter
    with contextlib.redirect_stdout(io.StringIO()):
        conv = CTCLabelConverter(synthetic_characters(classes[-1] - 4))
    T = am.shape[1]
    assert conv.decode(am.numpy(), [T] * B) == O.CTCConverter(synthetic_characters(classes[-1] - 4)).decode(am32.numpy(), [T] * B)
    del net


def set_drop_masks_from(net, masks):
    """pin the DropPath draws of the SVTR experts to `masks` (per expert: 22 [B] 0/1 tensors, two per stochastic block)"""
    if masks is None:
        return
    from mrn_amd.modules.svtr import DropPath
    for e, ms in enumerate(masks):
        mods = [m for m in net.model[e].modules() if isinstance(m, DropPath)]
        assert len(mods) == 11 and len(ms) == 22
        for j, m in enumerate(mods):
            m.forced_masks = [ms[2 * j].clone(), ms[2 * j + 1].clone()]


@pytest.mark.parametrize("crops", ["smooth", "noise"])
def test_crnn3_batch32_full_class_counts_vs_oracle(crops):
    """BASELINE config 2 at the bench's size: CRNN x 3 experts, CTC class counts 2090 / 2310 / 4038 (VERDICT r05 missing #3)"""
    _ctc_family_b32_case("crnn", (2090, 2310, 4038), crops, seed=41)


@pytest.mark.parametrize("crops", ["smooth", "noise"])
def test_svtr6_batch32_full_class_counts_vs_oracle(crops):
    """BASELINE config 4 at the bench's size: SVTR x 6 experts, CTC class counts 2090 ... 5373, injected DropPath draws"""
    _ctc_family_b32_case("svtr", (2090, 2310, 4038, 5198, 5271, 5373), crops, seed=43)


def _grad_check(name, mine, ref, rel_l2=2e-3, rel_max=2e-3):
    a = mine.detach().cpu().double().numpy()
    b = ref.detach().double().numpy()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-12)
    l2 = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)
    mx = np.abs(a - b).max() / scale
    assert l2 <= rel_l2 and mx <= rel_max, f"{name}: rel L2 {l2:.2e}, rel max {mx:.2e} (|g|max {scale:.2e})"


def test_loop_a_crnn_gradients_vs_oracle():
    """loop A (il_modules/mrn.py:232-269) on a CRNN expert: every parameter gradient of loss.backward() against torch
    autograd on the CPU oracle, then one clip + Adam step."""
    from mrn_amd import functional as Fn
    from mrn_amd.optim import FlatAdam
    from oracle import mrn_oracle as O
    kind, classes, B, seed = "crnn", (40,), 3, 4
    g = load_golden("crnn_mrn3")
    opt, net = build_net(kind, (40, 70, 97), g, 1)          # key layout from the fixture; only expert 0 is trained
    image, words, chars, _ = det_inputs(kind, classes, B, seed)
    conv, labels_index, labels_length = labels_for(kind, words, chars)
    # oracle side
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    names = [n for n, p in net.named_parameters() if n.startswith("model.0.")]
    params = [sd[n].requires_grad_(True) for n in names]
    cfg = O.Cfg("None", "VGG", "BiLSTM", "CTC")
    ref_out = O.model_forward(sd, "model.0.", cfg, image, None, True, training=True)["predict"]
    ref_loss = O.ctc_loss(ref_out, labels_index.cpu(), labels_length.cpu())
    ref_grads = torch.autograd.grad(ref_loss, params)
    # HIP side
    net.train()
    for n, p in net.named_parameters():
        p.requires_grad = n.startswith("model.0.")
    expert = net.model[0]
    preds = expert(image.cuda(), None, True)["predict"]
    loss = Fn.ctc_loss(preds, labels_index.cuda(), labels_length.cuda())
    assert_close("loop A logits", preds, ref_out, atol=1e-4)
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    loss.backward()
    mine = dict(net.named_parameters())
    for n, rg in zip(names, ref_grads):
        if rg.abs().max() < 1e-9:
            continue
        _grad_check(n, mine[n].grad, rg)
    # BatchNorm running statistics advanced identically
    for k in sd:
        if k.startswith("model.0.") and k.endswith("running_var"):
            assert_close(k, net.state_dict()[k], sd[k], atol=1e-5)
    # one optimiser step on the flat buffers
    tr = [p for p in net.parameters() if p.requires_grad]
    grads = [p.grad.clone() for p in tr]
    adam = FlatAdam(tr, lr=5e-4)
    for p, gr in zip(tr, grads):
        p.grad.copy_(gr)
    before = [p.detach().clone() for p in tr]
    nc = adam.step(lr=2.5e-5, max_norm=5.0)
    total = torch.norm(torch.stack([torch.norm(gg) for gg in ref_grads]))
    assert abs(nc[0].item() - total.item()) <= 2e-3 * total.item()
    moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(tr, before))
    assert moved > 0


def test_loop_a_svtr_gradients_vs_oracle():
    """loop A on an SVTR expert (BASELINE config 4's first task): every parameter gradient of loss.backward() -- LayerNorm,
    qkv / proj / MLP Linear layers, attention with the local mask, DropPath-scaled residuals, PatchEmbed conv + BN + GELU,
    SubSample convs, pos_embed -- against torch autograd on the CPU oracle with the same DropPath draws."""
    from mrn_amd import functional as Fn
    from oracle import mrn_oracle as O
    kind, classes, B, seed = "svtr", (40,), 3, 4
    g = load_golden("svtr_mrn3")
    opt, net = build_net(kind, (40, 70, 97), g, 3)
    image, words, chars, _ = det_inputs(kind, classes, B, seed)
    conv, labels_index, labels_length = labels_for(kind, words, chars)
    masks = drop_masks(B, seed, "loopA", 1)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    names = [n for n, p in net.named_parameters() if n.startswith("model.0.")]
    params = [sd[n].requires_grad_(True) for n in names]
    cfg = O.Cfg("None", "SVTR", "None", "CTC")
    ref_out = O.model_forward(sd, "model.0.", cfg, image, None, True, training=True, masks=masks[0])["predict"]
    ref_loss = O.ctc_loss(ref_out, labels_index.cpu(), labels_length.cpu())
    ref_grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    net.train()
    for n, p in net.named_parameters():
        p.requires_grad = n.startswith("model.0.")
    set_drop_masks(net, kind, B, seed, "loopA", [0])
    expert = net.model[0]
    preds = expert(image.cuda(), None, True)["predict"]
    loss = Fn.ctc_loss(preds, labels_index.cuda(), labels_length.cuda())
    assert_close("loop A logits", preds, ref_out, atol=1e-4)
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    loss.backward()
    mine = dict(net.named_parameters())
    checked = 0
    for n, rg in zip(names, ref_grads):
        if rg is None:                        # the reference's unused parameters (svtr.py:465-479) never receive gradients
            assert mine[n].grad is None or float(mine[n].grad.abs().max()) == 0.0, n
            continue
        if rg.abs().max() < 1e-9:
            continue
        if ".patch_embed.proj." in n and n.endswith(".bias") and n.split(".")[-2] in ("0", "3"):
            # conv bias in front of train-mode BatchNorm: exactly 0 here, fp32 round-off noise in torch autograd
            assert float(mine[n].grad.abs().max()) == 0.0 and float(rg.abs().max()) < 1e-5, n
            continue
        _grad_check(n, mine[n].grad, rg)
        checked += 1
    assert checked > 130          # (blocks whose DropPath draw is 0 for every sample have exactly-zero gradients)
    for k in sd:
        if k.startswith("model.0.") and k.endswith("running_var"):
            assert_close(k, net.state_dict()[k], sd[k], atol=1e-5)


def test_svtr_operand_fusion_matches_separate_passes():
    """loop A on an SVTR expert with the trained Linear layers' operands written by the producing pass (LayerNorm with its parameter bound,
    attention / GELU under the upstream GEMM's epilogue maximum, gradients' ranges folded into GELU', the DropPath residual and the attention
    backward kernels) against the separate max / split passes (ops.TRAIN_OPERAND_FUSION off): only the power-of-two range scales differ
    (a bound instead of the exact maximum), so logits, loss and every gradient agree to fp32 round-off -- measured 1e-6 of the logits'
    range, 3e-5 of a gradient tensor's (both paths are deterministic run to run)"""
    from mrn_amd import functional as Fn
    from mrn_amd import ops
    kind, classes, B, seed = "svtr", (40,), 8, 9
    g = load_golden("svtr_mrn3")
    image, words, chars, _ = det_inputs(kind, classes, B, seed)
    conv, labels_index, labels_length = labels_for(kind, words, chars)
    outs = []
    saved = ops.TRAIN_OPERAND_FUSION
    try:
        hits = {}
        for fused in (True, False):
            ops.TRAIN_OPERAND_FUSION = fused
            h0 = ops.OPERAND_STATS["hits"]
            opt, net = build_net(kind, (40, 70, 97), g, 3)
            net.train()
            for n, p in net.named_parameters():
                p.requires_grad = n.startswith("model.0.")
            set_drop_masks(net, kind, B, seed, "fusion", [0])
            preds = net.model[0](image.cuda(), None, True)["predict"]
            loss = Fn.ctc_loss(preds, labels_index.cuda(), labels_length.cuda())
            loss.backward()
            outs.append((preds.detach().clone(), float(loss.detach()), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
            hits[fused] = ops.OPERAND_STATS["hits"] - h0
    finally:
        ops.TRAIN_OPERAND_FUSION = saved
    # the fused path really was taken (a miss silently falls back to the separate passes: the A/B above could not tell): every trained
    # Linear of the twelve mixing blocks finds its operand or range in the forward pass and again in the backward pass
    assert hits[True] >= 60 and hits[False] == 0, hits
    # an entry is valid for the tensor at the version it was stashed under: an in-place change between producer and consumer is refused
    t = torch.ones(4, 32, device="cuda")
    ops.stash_operand(t, None, torch.ones(2, device="cuda"))
    assert ops.cached_operand(t) is not None
    stale0 = ops.OPERAND_STATS["stale"]
    t.add_(1.0)
    assert ops.cached_operand(t) is None and ops.OPERAND_STATS["stale"] == stale0 + 1
    (p1, l1, g1), (p0, l0, g0) = outs
    assert_close("logits", p1, p0, atol=5e-6 * float(p0.abs().max()), rtol=0)
    assert abs(l1 - l0) <= 5e-6 * max(1.0, abs(l0))
    assert set(g1) == set(g0) and len(g0) > 130
    for n in g0:
        scale = float(g0[n].abs().max())
        assert float((g1[n] - g0[n]).abs().max()) <= 2e-4 * scale + 1e-12, (n, float((g1[n] - g0[n]).abs().max()), scale)


def _oracle_trba_grads(g, image, labels_index, dtype):
    """loss and parameter gradients of a TRBA expert's loop A on the CPU oracle in the given precision"""
    from oracle import mrn_oracle as O
    sd = golden_state_dict(g, 2)
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    names = [k for k in sd if k.startswith("model.0.") and sd[k].is_floating_point() and "running" not in k
             and "generator" not in k]
    params = [sd[n].requires_grad_(True) for n in names]
    for k in list(sd):               # Prediction.generator.* aliases fc.* (one tensor under two keys)
        if k.startswith("model.0.Prediction.generator."):
            sd[k] = sd[k.replace("Prediction.generator.", "fc.")]
    cfg = O.Cfg("TPS", "ResNet", "BiLSTM", "Attn")
    old = O.tps_constants
    O.tps_constants = lambda *a: tuple(t.to(dtype) for t in old(*a))
    try:
        torch.set_default_dtype(dtype)
        out = O.model_forward(sd, "model.0.", cfg, image.to(dtype), labels_index[:, :-1], True, training=True)["predict"]
        loss = O.attn_ce_loss(out, labels_index)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
    finally:
        torch.set_default_dtype(torch.float32)
        O.tps_constants = old
    return names, grads, out.detach(), loss.detach()


@pytest.mark.parametrize("B,factor,floor", [(3, 3.0, 2e-3), (32, 2.0, 1e-3)])
def test_loop_a_trba_gradients_vs_oracle(B, factor, floor):
    """loop A on a TRBA expert (TPS + ResNet + BiLSTM + attention decoder): every parameter gradient of loss.backward().
    B = 32 is BASELINE config 1's batch: BatchNorm statistics over 32 x H x W samples are well conditioned there, so the band is the
    tighter max(2 x the reference's own fp32-vs-f64 error, 1e-3) per tensor.

    This 29-conv / small-batch-BatchNorm case is ill-conditioned in fp32: the oracle's own float32 gradients differ
    from its float64 gradients by ~2.5e-2 (median over parameters).  The HIP gradients are therefore judged against the
    float64 oracle and must be at least as close to it as 3x the reference's own fp32 arithmetic (the recurrent /
    decoder parameters, which are well conditioned, land at 1e-5..3e-4)."""
    from mrn_amd import functional as Fn
    kind, classes, seed = "trba", (41,), 6
    g = load_golden("trba_mrn3")
    opt, net = build_net(kind, (41, 71, 98), g, 2)
    image, words, chars, _ = det_inputs(kind, classes, B, seed)
    conv, labels_index, labels_length = labels_for(kind, words, chars)
    li = labels_index.cpu()
    names, g32, out32, loss32 = _oracle_trba_grads(g, image, li, torch.float32)
    _, g64, _, _ = _oracle_trba_grads(g, image, li, torch.float64)
    net.train()
    for n, p in net.named_parameters():
        p.requires_grad = n.startswith("model.0.")
    preds = net.model[0](image.cuda(), labels_index[:, :-1].cuda(), True)["predict"]
    loss = Fn.cross_entropy(preds, labels_index[:, 1:].cuda(), 1)
    assert_close("loop A logits", preds, out32, atol=1e-4)
    assert abs(loss.item() - loss32.item()) < 1e-4 * max(1.0, abs(loss32.item()))
    loss.backward()
    mine = dict(net.named_parameters())

    def rel(a, b):
        return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)

    for n, a32, a64 in zip(names, g32, g64):
        if a64 is None or a64.abs().max() < 1e-12:
            continue
        ref64 = a64.numpy()
        e_ref = rel(a32.double().numpy(), ref64)
        e_hip = rel(mine[n].grad.detach().cpu().double().numpy(), ref64)
        assert e_hip <= max(factor * e_ref, floor), f"{n}: HIP vs f64 {e_hip:.2e}, torch-f32 vs f64 {e_ref:.2e}"


@pytest.mark.parametrize("kind,B", [("trba", 3), ("crnn", 3), ("svtr", 3), ("svtr", 24)])
def test_loop_a_weight_gradients_on_the_side_stream_match(kind, B):
    """ops.direct_gradients() + ops.WGRAD_SIDE_STREAM: the parameter gradients that hang off the backward chain (weight gradients of the
    trained convolutions, of the Linear / LSTM / decoder layers) issued on a second stream and accumulated straight into the flat
    gradient give the flat gradient of the single-stream autograd path (to its run-to-run noise), also when gradients accumulate over
    two backward passes; the backward pass's final callback has joined the streams before .grad is read.  B = 24 on SVTR puts the
    Linear weight gradients on the split-K x3 GEMM and makes the residual gradients large enough for autograd's in-place accumulation
    to overtake a side-stream read (ops.side_stream_keep: NaN weights after one step before it)."""
    from mrn_amd import functional as Fn
    from mrn_amd import ops
    from mrn_amd.optim import FlatAdam
    classes, seed = {"trba": ((41,), 6), "crnn": ((40,), 4), "svtr": ((40,), 4)}[kind]
    g = load_golden(kind + "_mrn3")
    opt, net = build_net(kind, {"trba": (41, 71, 98)}.get(kind, (40, 70, 97)), g, {"trba": 2, "crnn": 1, "svtr": 3}[kind])
    image, words, chars, _ = det_inputs(kind, classes, B, seed)
    conv, labels_index, labels_length = labels_for(kind, words, chars)
    net.train()
    for n, p in net.named_parameters():
        p.requires_grad = n.startswith("model.0.")
    fo = FlatAdam([p for p in net.parameters() if p.requires_grad], lr=1e-3)
    bn_state = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}

    def flat_grad(side, passes):
        net.load_state_dict(bn_state, strict=False)
        ops.WGRAD_SIDE_STREAM = side
        fo.zero_grad()
        for _ in range(passes):
            if kind == "svtr":
                set_drop_masks(net, kind, B, seed, "loopA", [0])
            if kind == "trba":
                preds = net.model[0](image.cuda(), labels_index[:, :-1].cuda(), True)["predict"]
                loss = Fn.cross_entropy(preds, labels_index[:, 1:].cuda(), 1)
            else:
                preds = net.model[0](image.cuda(), None, True)["predict"]
                loss = Fn.ctc_loss(preds, labels_index.cuda(), labels_length.cuda())
            with ops.direct_gradients():                       # (what il_modules/base.py backward_and_step does when N = 1)
                loss.backward()
        return fo.grad.clone()
    keep = ops.WGRAD_SIDE_STREAM
    try:
        for passes in (1, 2):
            ref, ref2 = flat_grad(False, passes), flat_grad(False, passes)
            got = flat_grad(True, passes)
            assert float(ref.abs().max()) > 0
            noise = float((ref - ref2).abs().max())          # run-to-run (atomic reductions in the recurrent / loss kernels)
            print("side stream %s: max |grad| %.3e, run-to-run %.3e, side vs main %.3e" % (kind, float(ref.abs().max()), noise, float((ref - got).abs().max())))
            assert float((ref - got).abs().max()) <= max(4 * noise, 1e-6 * float(ref.abs().max()))
    finally:
        ops.WGRAD_SIDE_STREAM = keep


def test_side_stream_gradients_full_size_svtr_no_inplace_hazard():
    """SVTR loop A at 32 x 256 crops, batch 256 (the Linear weight gradients on the split-K x3 GEMM, residual gradients of 33 M
    elements): the flat gradient of one backward pass with the parameter gradients on the side stream equals the single-stream one.
    Pins ops.side_stream_keep: without the held references autograd accumulates the residual gradients IN PLACE on the main stream
    while the side stream still reads them (the weights were NaN after one step)."""
    import bench
    from mrn_amd import ops
    from mrn_amd.data.synthetic import SyntheticTextLines, synthetic_characters
    from mrn_amd.il_modules.mrn import MRN
    opt = bench.make_opt("svtr", 256)
    with contextlib.redirect_stdout(io.StringIO()):
        learner = MRN(opt)
        learner.character = synthetic_characters(2086)
        learner.converter = learner.build_converter()
        learner.criterion = learner.build_criterion()
        learner.build_model()
        learner.build_optimizer(learner.count_param())
    data = SyntheticTextLines(opt, seed=111)
    data.set_characters(learner.character)
    image, labels = data.get_batch()
    labels_index, labels_length = learner.converter.encode(labels, batch_max_length=opt.batch_max_length)
    bn_state = {k: v.clone() for k, v in learner.model.state_dict().items() if "running" in k or "num_batches" in k}

    def flat_grad(side):
        learner.model.load_state_dict(bn_state, strict=False)
        ops.WGRAD_SIDE_STREAM = side
        torch.manual_seed(7)                                   # the DropPath draws
        learner.optimizer.zero_grad()
        loss = learner.criterion(learner._forward_train(image, None), labels_index, labels_length)
        with ops.direct_gradients():
            loss.backward()
        return learner.optimizer.grad.clone()
    keep = ops.WGRAD_SIDE_STREAM
    try:
        ref, ref2, got = flat_grad(False), flat_grad(False), flat_grad(True)
        noise = float((ref - ref2).abs().max())
        assert torch.isfinite(got).all()
        assert float((ref - got).abs().max()) <= max(4 * noise, 1e-6 * float(ref.abs().max())), (float((ref - got).abs().max()), noise)
    finally:
        ops.WGRAD_SIDE_STREAM = keep


def test_dernet_vs_golden():
    """DERNet (reference modules/model.py:203-312) forward in DER's training configuration + weight_align"""
    from mrn_amd.modules.model import DERNet
    g = load_golden("crnn_der2")
    classes, B, seed = (40, 70), 2, 4
    opt = make_opt("crnn")
    with contextlib.redirect_stdout(io.StringIO()):
        net = DERNet(opt)
        for c in classes:
            net.update_fc(opt.hidden_size, c)
            net.build_prediction(opt, c)
            net.build_aux_prediction(opt, c)
    ref = {str(k): str(s) for k, s in zip(g["sd_keys"], g["sd_shapes"])}
    assert {k: ",".join(map(str, v.shape)) for k, v in net.state_dict().items()} == ref
    net.load_state_dict(golden_state_dict(g, seed), strict=True)
    net = net.cuda().train()
    net.model[0].eval()
    image, _, _, _ = det_inputs("crnn", classes, B, seed)
    with torch.no_grad():
        out = net(image.cuda())
        assert_sub_close(g, "logits", out["logits"], atol=1e-4)
        assert_sub_close(g, "aux_logits", out["aux_logits"], atol=1e-4)
        assert_sub_close(g, "features", out["features"], atol=1e-4)
        with contextlib.redirect_stdout(io.StringIO()):
            gamma = net.weight_align(30)
        assert abs(float(gamma) - float(g["weight_align_gamma"])) < 1e-5
        assert_sub_close(g, "fc_after_align", net.fc.weight, atol=1e-6)


@pytest.mark.parametrize("arch", ["trba", "crnn", "svtr"])
@pytest.mark.parametrize("train_mode", [True, False])
def test_grouped_backbones_match_per_expert_path(arch, train_mode):
    """modules/expert_group.py (G experts in lock-step on the 256-wide grouped conv) against the per-expert path:
    same features / fused logits / routing weights and the same BatchNorm running statistics."""
    import contextlib
    import io
    import types
    from mrn_amd.modules.model import MRNNet
    from mrn_amd.tools import weights as W
    stages = dict(trba=("TPS", "ResNet", "BiLSTM", "Attn"), crnn=("None", "VGG", "BiLSTM", "CTC"),
                  svtr=("None", "SVTR", "None", "CTC"))[arch]
    opt = types.SimpleNamespace(Transformation=stages[0], FeatureExtraction=stages[1], SequenceModeling=stages[2],
                                Prediction=stages[3], num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512,
                                hidden_size=256, batch_max_length=25)
    classes = (30, 45, 61)

    def build():
        with contextlib.redirect_stdout(io.StringIO()):
            net = MRNNet(opt)
            for c in classes:
                net.update_fc(256, c)
                net.build_prediction(opt, c)
        W.fill_state_dict(net.state_dict(), seed=11)
        net = net.cuda()
        for p in net.parameters():
            p.requires_grad = False
        return net.train() if train_mode else net.eval()

    B = 5
    image = torch.from_numpy(W.smooth_image("grp", (B, 4, 32, 256), 3)).cuda()
    text = torch.from_numpy(W.randint("grp_text", (B, 26), 4, classes[-1], 3)).cuda()
    text[:, 0] = 2
    outs = []
    for grouping in (True, False):
        net = build()
        net.expert_grouping = grouping
        if train_mode:
            set_drop_masks(net, arch, B, 5, "grp", range(len(classes)))      # SVTR: the same DropPath draws on both paths
        with torch.no_grad():
            o = net(image, True, text if stages[3] == "Attn" else None, True)
        assert (net._backbone_group() is not None) == grouping
        bn = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
        outs.append((o["logits"].clone(), o["index"].clone(), bn))
    (la, wa, bna), (lb, wb, bnb) = outs
    # TRBA: the two paths sum the localisation convs in different orders and the TPS grid amplifies that fp32 round-off
    # to the 1e-4 level; CRNN has no such stage and agrees to fp32 round-off
    tol = 2e-4 if arch == "trba" else (2e-5 if arch == "svtr" else 2e-6)   # (SVTR: Seq / CTC Linear x3 grouped vs exact fp32)
    assert_close("grouped logits", la, lb, atol=10 * tol, rtol=1e-4)
    assert_close("grouped routing weights", wa, wb, atol=tol, rtol=1e-4)
    for k in bna:
        if "num_batches" in k:
            assert torch.equal(bna[k], bnb[k]), k
        else:
            assert_close(k, bna[k], bnb[k], atol=1e-6, rtol=2e-5)


@pytest.mark.parametrize("arch", ["crnn", "trba"])
def test_eval_fold_follows_running_statistics_updated_in_train_mode(arch):
    """validation -> train-mode steps -> validation on ONE BackboneGroup (MRN loop B validates at iteration 1 and again later while
    the frozen experts keep running train-mode BatchNorm, reference il_modules/mrn.py:107,379,401): the eval-mode BatchNorm folded
    into the conv epilogue must use the CURRENT running statistics, which kernels update through raw pointers (no version bump).
    Checked against the per-expert path run through the same sequence."""
    import contextlib
    import io
    import types
    from mrn_amd.modules.model import MRNNet
    from mrn_amd.tools import weights as W
    stages = dict(trba=("TPS", "ResNet", "BiLSTM", "Attn"), crnn=("None", "VGG", "BiLSTM", "CTC"))[arch]
    opt = types.SimpleNamespace(Transformation=stages[0], FeatureExtraction=stages[1], SequenceModeling=stages[2],
                                Prediction=stages[3], num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512,
                                hidden_size=256, batch_max_length=25)
    classes = (30, 45, 61)
    B = 4
    images = [torch.from_numpy(W.smooth_image("fold%d" % i, (B, 4, 32, 256), 3 + i)).cuda() * (1.0 + 0.5 * i) for i in range(3)]
    text = torch.from_numpy(W.randint("fold_text", (B, 26), 4, classes[-1], 3)).cuda()
    text[:, 0] = 2
    txt = text if stages[3] == "Attn" else None
    outs = []
    for grouping in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            net = MRNNet(opt)
            for c in classes:
                net.update_fc(256, c)
                net.build_prediction(opt, c)
        W.fill_state_dict(net.state_dict(), seed=13)
        net = net.cuda()
        for p in net.parameters():
            p.requires_grad = False
        net.expert_grouping = grouping
        with torch.no_grad():
            net.eval()
            first = net(images[0], True, txt, True)["logits"].clone()
            net.train()
            for im in images[1:]:
                net(im, True, txt, True)          # running statistics move (momentum 0.1, different input scales)
            net.eval()
            second = net(images[0], True, txt, True)
        outs.append((first, second["logits"].clone(), second["index"].clone()))
    (fa, sa, wa), (fb, sb, wb) = outs
    tol = 2e-4 if arch == "trba" else 2e-6
    assert float((sb - fb).abs().max()) > 100 * tol, "the sequence must move the eval-mode outputs for the test to mean anything"
    assert_close("eval logits before the train-mode steps", fa, fb, atol=10 * tol, rtol=1e-4)
    assert_close("eval logits after the train-mode steps", sa, sb, atol=10 * tol, rtol=1e-4)
    assert_close("eval routing weights after the train-mode steps", wa, wb, atol=tol, rtol=1e-4)


@pytest.mark.parametrize("train_mode", [True, False])
def test_dernet_groups_frozen_extractors(train_mode):
    """DERNet with three frozen extractors + one trainable: the frozen ones run in lock-step (expert_group.py) and give
    the features / logits of the per-extractor path"""
    from mrn_amd.modules.model import DERNet
    from mrn_amd.tools import weights as W
    opt = make_opt("crnn")
    classes = (30, 45, 61, 80)
    outs = []
    for grouping in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            net = DERNet(opt)
            for c in classes:
                net.update_fc(256, c)
            net.build_prediction(opt, classes[-1])
            net.build_aux_prediction(opt, classes[-1])
        W.fill_state_dict(net.state_dict(), seed=13)
        net = net.cuda()
        for ext in list(net.model)[:-1]:
            for p in ext.parameters():
                p.requires_grad = False
        net.train() if train_mode else net.eval()
        net.expert_grouping = grouping
        image = torch.from_numpy(W.smooth_image("der", (4, 4, 32, 256), 3)).cuda()
        with contextlib.nullcontext() if train_mode else torch.no_grad():     # (eval-mode BatchNorm has no backward path)
            o = net(image)
        assert (net._group is not None) == grouping
        outs.append((o["features"].detach().clone(), o["logits"].detach().clone()))
    assert_close("der features", outs[0][0], outs[1][0], atol=2e-5, rtol=1e-4)
    assert_close("der logits", outs[0][1], outs[1][1], atol=2e-5, rtol=1e-4)


def test_dernet_frozen_side_stream_and_prefetch_are_bit_identical():
    """DERNet's three ways to run the frozen lock-step group -- in line, on the side stream next to the trained extractor, and
    ahead of time through frozen_prefetch() (DER._update's look-ahead) -- only change the launch order: identical bits, and the
    trained extractor's gradients too"""
    from mrn_amd.modules.model import DERNet
    from mrn_amd.tools import weights as W
    opt = make_opt("trba")
    classes = (30, 45, 61)
    with contextlib.redirect_stdout(io.StringIO()):
        net = DERNet(opt)
        for c in classes:
            net.update_fc(256, c)
        net.build_prediction(opt, classes[-1])
        net.build_aux_prediction(opt, classes[-1])
    W.fill_state_dict(net.state_dict(), seed=17)
    net = net.cuda().train()
    for ext in list(net.model)[:-1]:
        ext.eval()
        for p in ext.parameters():
            p.requires_grad = False
    image = torch.from_numpy(W.smooth_image("derp", (6, 4, 32, 256), 5)).cuda()
    text = torch.randint(0, classes[-1], (6, 26), generator=torch.Generator().manual_seed(3)).cuda()
    runs = []
    for mode in ("inline", "side", "prefetch"):
        net.zero_grad(set_to_none=True)
        net.frozen_stream = mode != "inline"
        handle = net.frozen_prefetch(image) if mode == "prefetch" else None
        assert (handle is not None) == (mode == "prefetch")
        o = net(image, text, True, frozen=handle)
        (o["logits"].square().mean() + o["aux_logits"].square().mean()).backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.flatten() for p in net.model[-1].parameters()])
        runs.append((o["features"].detach().clone(), o["logits"].detach().clone(), g.clone()))
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)


@pytest.mark.parametrize("arch", ["trba", "crnn", "svtr"])
def test_two_stream_half_groups_are_bit_identical(arch):
    """MRNNet with >= 4 experts splits them into two lock-step half-groups on two HIP streams (so one half's HBM-bound
    passes overlap the other half's convolutions); per-expert arithmetic is unchanged, so results are bit-identical to
    the single group on one stream, including the BatchNorm running statistics"""
    from mrn_amd.modules.model import MRNNet
    from mrn_amd.tools import weights as W
    opt = make_opt(arch)
    classes = (30, 45, 61, 80, 97)
    B = 3
    image = torch.from_numpy(W.smooth_image("halves", (B, 4, 32, 256), 3)).cuda()
    text = torch.from_numpy(W.randint("halves_text", (B, 26), 4, classes[-1], 3)).cuda()
    text[:, 0] = 2
    outs = []
    for parts in (2, 0):
        with contextlib.redirect_stdout(io.StringIO()):
            net = MRNNet(opt)
            for c in classes:
                net.update_fc(256, c)
                net.build_prediction(opt, c)
        W.fill_state_dict(net.state_dict(), seed=17)
        net = net.cuda().train()
        for p in net.parameters():
            p.requires_grad = False
        net.expert_halves = parts
        set_drop_masks(net, arch, B, 6, "halves", range(len(classes)))
        with torch.no_grad():
            o = net(image, True, text if opt.Prediction == "Attn" else None, True)
        torch.cuda.synchronize()
        assert (net._halves is not None) == (parts == 2)
        outs.append((o["logits"].clone(), o["index"].clone(), {k: v.clone() for k, v in net.state_dict().items() if "running_" in k}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k


def test_full_size_svtr_lockstep_matches_per_expert():
    """BASELINE config 4 at full size (6 SVTR experts, 256 images per GPU): the lock-step path (grouped x3 Linear layers, fused
    add + LayerNorm passes, x3 attention over 1536 samples, two half-groups on two streams) against the per-expert path with
    exact-fp32 attention products, same DropPath draws -- a size-independent consistency property, plus the row-stochastic
    routing weights."""
    from mrn_amd import ops
    from mrn_amd.modules.model import MRNNet
    from mrn_amd.tools import weights as W
    opt = make_opt("svtr")
    classes = (40, 70, 97, 120, 150, 181)
    B = 256
    image = torch.from_numpy(W.smooth_image("full_svtr", (B, 4, 32, 256), 3)).cuda()
    outs = []
    for lockstep in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            net = MRNNet(opt)
            for c in classes:
                net.update_fc(256, c)
                net.build_prediction(opt, c)
        W.fill_state_dict(net.state_dict(), seed=23)
        net = net.cuda().train()
        for p in net.parameters():
            p.requires_grad = False
        net.expert_grouping = lockstep
        set_drop_masks(net, "svtr", B, 9, "full", range(len(classes)))
        saved = ops.SVTR_ATTENTION_X3
        try:
            ops.SVTR_ATTENTION_X3 = lockstep
            with torch.no_grad():
                o = net(image, True, None, True)
        finally:
            ops.SVTR_ATTENTION_X3 = saved
        torch.cuda.synchronize()
        outs.append((o["logits"].clone(), o["index"].clone()))
        del net
    assert_close("full-size logits", outs[0][0], outs[1][0], atol=2e-4, rtol=1e-4)
    assert_close("full-size routing weights", outs[0][1], outs[1][1], atol=2e-5, rtol=1e-4)
    assert_close("routing weights sum to 1", outs[0][1].sum(1), torch.ones(B), atol=1e-5)


@pytest.mark.parametrize("arch", ["crnn", "trba"])
def test_reduced_precision_training_mode_gradients(arch):
    """ops.TRAIN_PRODUCTS = 1 (bench.py --precision fp16 --loop a / der): the trained convolutions' forward, data gradient and
    weight gradient on ONE fp16 product per term (range-scaled operands, fp32 accumulate) -- mixed-precision training as BASELINE
    config 5 ("fp16 MFMA") runs it.  Pins what it meets against the parity mode on a full expert (B = 32): loss within 2e-3
    relative, the global gradient norm within 1 %; CRNN: every parameter gradient within 8 % relative L2 (tensors of >= 1 M elements
    2 %), direction cosine >= 0.999; TRBA (30 layers + TPS, every single product accurate to 2.9e-4): per tensor <= 0.5 (small-norm tensors), whole-gradient cosine >= 0.98."""
    from mrn_amd import functional as Fn
    from mrn_amd import ops
    from mrn_amd.modules.model import Model
    from mrn_amd.tools import weights as W
    opt = make_opt(arch)
    C, B = 97, 32
    with contextlib.redirect_stdout(io.StringIO()):
        net = Model(opt)
        net.update_fc(opt.hidden_size, C)
        net.build_prediction(opt, C)
    W.fill_state_dict(net.state_dict(), seed=43)
    net = net.cuda().train()
    # (TRBA on smooth crops: on white noise the TPS rectifier is so ill-conditioned that even the fp32 reference only reproduces
    # itself to 3.5e-3 -- test_trba_tps_conditioning_smooth_vs_noise -- and gradients of two arithmetic variants are not comparable)
    image = torch.from_numpy(W.uniform("redtrain", (B, 4, 32, 256), -1.0, 1.0, 5) if arch == "crnn" else
                             W.smooth_image("redtrain", (B, 4, 32, 256), 5)).cuda()
    text = torch.from_numpy(W.randint("redtrain_text", (B, 27), 4, C, 3)).cuda()
    text[:, 0] = 2
    tlen = torch.full((B,), 12, dtype=torch.int32, device="cuda")
    runs = {}
    for products in (3, 1):
        ops.TRAIN_PRODUCTS = products
        try:
            net.zero_grad(set_to_none=True)
            if arch == "crnn":
                loss = Fn.ctc_loss(net(image)["predict"], text[:, 1:13].contiguous(), tlen)
            else:
                loss = Fn.cross_entropy(net(image, text[:, :-1], True)["predict"], text[:, 1:], 1)
            loss.backward()
            torch.cuda.synchronize()
            runs[products] = (float(loss.detach()), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
        finally:
            ops.TRAIN_PRODUCTS = 3
    (l3, g3), (l1, g1) = runs[3], runs[1]
    assert abs(l1 - l3) <= 2e-3 * abs(l3), (l1, l3)
    n3 = torch.sqrt(sum(g.double().pow(2).sum() for g in g3.values()))
    n1 = torch.sqrt(sum(g.double().pow(2).sum() for g in g1.values()))
    assert abs(float(n1 - n3)) <= 1e-2 * float(n3)
    worst = 0.0
    for n, a in g3.items():
        rel = float((g1[n] - a).norm() / a.norm().clamp_min(1e-30))
        worst = max(worst, rel)
        if os.environ.get("MRN_REPORT"):
            print(f"{n:60s} {a.numel():9d} rel {rel:.3e}")
        elif arch == "crnn":       # measured 1.2e-3 (last conv) ... 5.4e-2 (first conv: seven fp16-product data gradients deep)
            assert rel <= (2e-2 if a.numel() >= 1000000 else 8e-2), (n, rel)
        else:                      # TRBA: 30 layers + the TPS sampler amplify the 3e-4 per-product error to 0.2-0.36 per tensor
            assert rel <= 0.5, (n, rel)
    flat3 = torch.cat([g.flatten() for g in g3.values()]).double()
    flat1 = torch.cat([g1[n].flatten() for n in g3]).double()
    cos = float(torch.dot(flat3, flat1) / (flat3.norm() * flat1.norm()))
    if os.environ.get("MRN_REPORT"):
        print(f"global cosine {cos:.5f} worst rel {worst:.3e}")
    assert cos >= (0.9999 if arch == "crnn" else 0.98), cos          # measured 0.99999 / 0.9917
    assert worst > 1e-6            # (the mode really ran: one product is not the parity arithmetic)


@pytest.mark.parametrize("name", ["crnn_mrn3", "trba_mrn3", "svtr_mrn3", "crnn_mrn3_noise"])
def test_reduced_precision_mode_tolerance_and_agreement(name):
    """ops.X3_PRODUCTS = 1 (bench.py --precision fp16): ONE fp16 product per term instead of the split-fp16 x3 of the parity mode,
    fp32 accumulation and storage unchanged -- the reduced-precision mode BASELINE configs 2 ("bf16") and 5 ("fp16 MFMA") ask
    for (fp16 keeps 11 significand bits where bf16 keeps 8).  It does NOT meet the 1e-4 band and is never the headline; this
    pins what it does meet against the reference's golden outputs: routing weights within 2e-2, fused logits within 5e-2 of
    their scale, the routing argmax and the greedy CTC strings / eval routing indices unchanged on these cases."""
    from mrn_amd import ops
    kind, classes, B, seed = CASES.get(name, ("trba", (41, 71, 98), 2, 2))
    g = load_golden(name)
    opt, net = build_net(kind, classes, g, seed)
    image, words, chars, _ = det_inputs(kind, classes, B, seed, noise=name.endswith("_noise"))
    conv, labels_index, _ = labels_for(kind, words, chars)
    ctc = kind != "trba"
    saved = ops.X3_PRODUCTS
    try:
        ops.X3_PRODUCTS = 1
        net.train()
        set_drop_masks(net, kind, B, seed, "stepB0", list(range(len(classes))))
        with torch.no_grad():
            out = net(image.cuda(), True, None if ctc else labels_index[:, :-1].cuda(), True)
        w = out["index"].cpu().numpy()
        err_w = np.abs(w - g["stepB/weights"]).max()
        from tests.helpers import sub
        ref_l = g["stepB/logits/sub"]
        err_l = np.abs(sub(out["logits"])[0] - ref_l).max() / max(np.abs(ref_l).max(), 1e-6)
        print(f"[reduced precision] {name}: routing weights max err {err_w:.2e}, fused logits max err / scale {err_l:.2e}")
        assert err_w <= 2e-2 and err_l <= 5e-2, (err_w, err_l)
        assert err_w > 1e-6                                   # it really is the reduced mode (x3 sits at ~1e-6 on CRNN)
        assert np.array_equal(w.argmax(1), g["stepB/weights"].argmax(1))
        reload(net, g, seed)
        net.eval()
        with torch.no_grad():
            sos = None if ctc else torch.LongTensor(B).fill_(2).cuda()
            oe = net(image.cuda(), True, sos, False)
        assert np.array_equal(oe["index"].cpu().numpy(), g["eval/index"])
        am = oe["logits"].max(2)[1].cpu().numpy()
        agree = float((am == g["eval/argmax"]).mean())
        print(f"[reduced precision] {name}: eval argmax agreement {agree:.4f}")
        assert agree >= 0.97
    finally:
        ops.X3_PRODUCTS = saved


def test_rcnn_extractor_vs_reference():
    """RCNN_FeatureExtractor / GRCL / GRCL_unit (reference modules/feature_extraction.py:50-162) on the HIP path:
    state_dict layout, train-mode forward (five BatchNorms per recurrent iteration, running statistics), CTC loss, parameter
    gradients through the gated recurrence, eval-mode forward with bit-exact argmax -- all against the reference's outputs"""
    from mrn_amd import functional as Fn
    from mrn_amd.modules.model import Model
    from mrn_amd.tools import weights as W
    g = load_golden("rcnn_model")
    opt = make_opt("crnn")
    opt.FeatureExtraction = "RCNN"
    B, seed = 2, 9
    with contextlib.redirect_stdout(io.StringIO()):
        net = Model(opt)
        net.update_fc(opt.hidden_size, 40)
        net.build_prediction(opt, 40)
    ref = {str(k): str(s) for k, s in zip(g["sd_keys"], g["sd_shapes"])}
    assert {k: ",".join(map(str, v.shape)) for k, v in net.state_dict().items()} == ref
    net.load_state_dict(golden_state_dict(g, seed), strict=True)
    net = net.cuda().train()
    image = torch.from_numpy(W.smooth_image("input:image", (B, 4, 32, 256), seed)).cuda()
    chars = "".join(chr(0x4E00 + i) for i in range(36))
    lens = W.randint("label_len", (B,), 1, 26, seed)
    words = ["".join(chars[i] for i in W.randint(f"label_{b}", (int(lens[b]),), 0, 36, seed)) for b in range(B)]
    conv, labels_index, labels_length = labels_for("crnn", words, chars)
    out = net(image, None, True)
    assert_sub_close(g, "train/feature", out["feature"], atol=1e-4)
    assert_sub_close(g, "train/predict", out["predict"], atol=1e-4)
    loss = Fn.ctc_loss(out["predict"], labels_index.cuda(), labels_length.cuda())
    assert abs(loss.item() - float(g["train/loss"])) < 1e-4 * max(1.0, float(g["train/loss"]))
    loss.backward()
    for k in [str(s) for s in g["grad_keys"]]:
        assert_sub_close(g, f"grad/{k}", net.get_parameter(k).grad, atol=1e-6, rtol=5e-3)
    assert_close("running_var", net.state_dict()["model.FeatureExtraction.ConvNet.5.GRCL.1.BN_rx.running_var"], g["bn_running_var_after"], atol=1e-5)
    net.load_state_dict(golden_state_dict(g, seed), strict=True)
    net.eval()
    with torch.no_grad():
        oe = net(image, None, False)
    assert_sub_close(g, "eval/predict", oe["predict"], atol=1e-4)
    assert np.array_equal(oe["predict"].max(2)[1].cpu().numpy(), g["eval/argmax"])


@pytest.mark.parametrize("arch", ["crnn", "svtr", "trba"])
def test_full_size_loop_a_directional_derivative(arch):
    """Loop A at BASELINE size (one expert, 256 crops, 2090 classes): a size-independent property of loss.backward() -- along the
    normalised gradient direction d = g / |g| the central difference (L(theta + eps d) - L(theta - eps d)) / (2 eps) equals |g|.
    Ties the whole backward (conv dgrad / weight gradient without im2col, BatchNorm, pooling, BiLSTM BPTT or SVTR attention, CTC) to
    the forward at the size the bench runs.  Eval-free: BatchNorm in train mode on the same batch for all three evaluations.
    TRBA (the north-star family: TPS sampler, Winograd forward / data / weight gradients of the ResNet, x3 BPTT, the attention
    decoder's backward at 4 samples per workgroup, side-stream parameter gradients) runs it with the CE loss of the Attn head on a
    fixed teacher-forcing text."""
    from mrn_amd import functional as Fn
    from mrn_amd import ops
    from mrn_amd.modules.model import Model
    from mrn_amd.tools import weights as W
    opt = make_opt(arch)
    C, B = (2091 if arch == "trba" else 2090), 256
    with contextlib.redirect_stdout(io.StringIO()):
        net = Model(opt)
        net.update_fc(opt.hidden_size, C)
        net.build_prediction(opt, C)
    W.fill_state_dict(net.state_dict(), seed=41)
    net = net.cuda().train()
    if arch == "svtr":
        from mrn_amd.modules.svtr import DropPath
        for m in net.modules():
            if isinstance(m, DropPath):
                m.drop_prob = 0.0                      # (the three evaluations must see the same function)
    image = torch.from_numpy(W.uniform("fullA", (B, 4, 32, 256), -1.0, 1.0, 5)).cuda()
    labels = torch.from_numpy(W.randint("fullA_lab", (B, 25), 4, C, 5)).cuda()
    lengths = torch.from_numpy(W.randint("fullA_len", (B,), 1, 26, 5)).int().cuda()
    bn_state = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}

    if arch == "trba":                                 # [SOS] text [EOS] [PAD]...: AttnLabelConverter.encode's layout (tools/utils.py)
        ln = lengths.long().clamp(max=25)
        pos = torch.arange(27, device="cuda")[None, :]
        index = torch.cat([torch.full((B, 1), 2, device="cuda"), labels.long().clamp(min=5), torch.ones(B, 1, dtype=torch.long, device="cuda")], 1)
        index = torch.where(pos == ln[:, None] + 1, torch.full_like(index, 3), index)
        index = torch.where(pos > ln[:, None] + 1, torch.ones_like(index), index)

    def loss_at():
        net.load_state_dict(bn_state, strict=False)
        if arch == "trba":
            return Fn.cross_entropy(net(image, index[:, :-1], True)["predict"], index[:, 1:], 1)
        return Fn.ctc_loss(net(image, None, True)["predict"], labels, lengths)

    params = [p for p in net.parameters() if p.requires_grad]
    from mrn_amd.optim import FlatAdam
    fo = FlatAdam(params, lr=1e-3)                     # the learners' flat parameter / gradient buffers (the side stream adds into .grad)
    fo.zero_grad()
    loss = loss_at()
    with ops.direct_gradients():                       # (the learners' backward_and_step: parameter gradients on the side stream)
        loss.backward()
    grads = [p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    assert all(bool(torch.isfinite(g).all()) for g in grads)
    if arch == "trba":
        # The direction leaves the TPS localisation network out: on U(-1,1) noise crops the loss is piecewise bilinear in the sampling
        # grid with a kink at every pixel boundary, so a finite step along d(loss)/d(grid) measures nothing (fd = 0.008 / -0.045 / 0.13
        # for eps = 2e-3 / 5e-4 / 1e-4 against |g| = 0.81, while every other stage agrees to 1-3 %); those gradients are pinned
        # against the float64 oracle by test_loop_a_trba_gradients_vs_oracle (B = 3, 32).  The backward chain through the ResNet, BiLSTM
        # and decoder that FEEDS the sampler's gradient is what this direction exercises.
        loc = {id(p) for n, p in net.named_parameters() if "Transformation" in n}
        assert loc
        grads = [torch.zeros_like(g) if id(p) in loc else g for p, g in zip(params, grads)]
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
    assert np.isfinite(gnorm) and gnorm > 0
    eps = 2e-3
    vals = []
    with torch.no_grad():
        for sign in (+1.0, -1.0):
            for p, g in zip(params, grads):
                p.add_(g, alpha=sign * eps / gnorm)
            torch.autograd.graph.increment_version(params)
            vals.append(float(loss_at()))
            for p, g in zip(params, grads):
                p.add_(g, alpha=-sign * eps / gnorm)
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - gnorm) <= (3e-2 if arch == "trba" else 2e-2) * gnorm, (fd, gnorm, float(loss))
