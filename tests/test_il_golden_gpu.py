"""BASELINE config 5 on the GPU against fixtures produced by the REFERENCE learners (tests/golden/make_golden_il.py):
LwF / EWC / DER flows over two tasks for the CRNN and the TRBA family, DERNet over two TRBA extractors, and a full-size
property test of DERNet TRBA x 6 at B = 256.  Reference: il_modules/{base,lwf,ewc,der}.py, modules/model.py:203-312."""
import contextlib
import io
import os
import types

import numpy as np
import pytest
import torch

from mrn_amd.tools import weights as W
from tests.helpers import DetLoader, assert_close, assert_sub_close, assert_sub_l2, golden_state_dict, load_golden

pytestmark = pytest.mark.gpu

NCHARS = (36, 30)
SEEDS = {"lwf": (21, 22), "ewc": (23, 24), "der": (25, 26)}


def chars_upto(t):
    return "".join(chr(0x4E00 + i) for i in range(sum(NCHARS[:t + 1])))


def learner_opt(kind, num_iter=2):
    o = types.SimpleNamespace(num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512, hidden_size=256,
                              batch_max_length=25, exp_name="g", il="x", memory=None, memory_num=2000, start_task=0,
                              schedule="super", optimizer="adam", lr=0.00003, batch_size=2, num_iter=num_iter, val_interval=1000,
                              grad_clip=5, lan_list=["A", "B"], NED=True, workers=0, manual_seed=111)
    if kind == "crnn":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "VGG", "BiLSTM", "CTC"
    else:
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "TPS", "ResNet", "BiLSTM", "Attn"
    return o


def hooked(cls, seeds):
    """the same hooks the golden generator put on the reference learners: deterministic weights after (re)building"""
    class Hooked(cls):
        def build_model(self):
            super().build_model()
            W.fill_state_dict(self.model.state_dict(), seeds[0])

        def change_model(self):
            super().change_model()
            W.fill_state_dict(self.model.state_dict(), seeds[1])
    return Hooked


def record(learner, name, out):
    """wrap a step method so that its returned loss tensors are logged (what the reference feeds its Averagers)"""
    fn = getattr(learner, name)

    def wrapped(*a, **k):
        r = fn(*a, **k)
        out.append(tuple(float(v.detach()) for v in (r if isinstance(r, tuple) else (r,))))
        return r
    setattr(learner, name, wrapped)


_F64 = {}
_ALT = {}                        # name -> (movement of the same flow on the alternate convolution kernel, its distance from float64)
_MOVE_MODE = ["assert"]
MOVEMENT_FACTOR, MOVEMENT_FLOOR = 3.0, 1e-2
COND_FACTOR, COND_FLOOR = 3.0, 1e-2
# Tensors whose two-kernel spread is allowed above COND_FACTOR x the reference's own fp32-vs-float64 distance, pinned BY NAME with the
# cap they get (calibrated on MI355X with MRN_MOVEMENT_LOG; DESIGN section 2: LwF task 0 on CRNN has ONE knife-edge ReLU at the last
# convolution -- a pre-activation within 1e-6 of zero under one of the largest upstream gradients -- that flips with the accumulation
# order of either kernel and moves 14 % of the two-step Adam movement of the tensors upstream of it, while the reference's fp32 and
# float64 runs agree to 3e-5 there).  Every other tensor of every flow has to stay inside the reference's own conditioning.
KNIFE_EDGE_CAPS = {("crnn", "lwf/t0/delta/model.FeatureExtraction.ConvNet.14.weight/sub"): 0.2}      # measured 0.139 (r05 calibration: the
# row-block kernel's run flips the ReLU, the x3 kernel's run sits at 3.7e-5 of the float64 movement; 182 other tensors need no entry:
# their spread is <= 1.6 x the reference's own fp32-vs-float64 distance or under the 1e-2 floor)
MOVEMENT_LOG = []
_FLOWS_SEEN, _CAPS_VISITED = set(), set()     # (kind, flow) pairs whose movements were judged / cap entries that were actually used


def _assert_movement(kind, name, k, mine, ref32):
    """parameter movement of a task's optimiser steps (1024-element subsample), judged against the float64 yardstick:
    tests/golden/il_{trba,crnn}_f64.npz hold the SAME reference flows run in float64 arithmetic (make_golden_il_f64.py).  Adam turns the
    fp32 round-off of near-zero gradients into +-lr steps, so the reference's own fp32 run is 12-35 % (relative L2) away from its float64
    run on TRBA's ResNet / TPS tensors and up to 9 % on CRNN's first convolution, while it sits at 5e-5 ... 5e-3 on the recurrent / head
    tensors.  Every flow runs twice: on the x3 kernel's Winograd form (the alternate kernel, first) and on the row-block kernel (the
    default); the two agree to 1e-6 launch by launch, so their movements differ only through the conditioning of the flow itself.
    Three assertions per tensor, none of which can be satisfied by the quantity it bounds:
      1. band = max(MOVEMENT_FACTOR x the reference's own fp32 run, MOVEMENT_FLOOR): the CLOSER of the two HIP runs must be inside it
         (MOVEMENT_FLOOR: the split-fp16 x3 products' distance on well-conditioned tensors, <= 3.4e-3 measured);
      2. the two HIP runs may differ from each other by no more than max(COND_FACTOR x the reference's fp32-vs-float64 distance,
         COND_FLOOR) -- a bug that only one of the two kernels has shows up HERE, however close the other run is to the yardstick --
         except for the tensors pinned by name in KNIFE_EDGE_CAPS, which get the absolute cap written there;
      3. hence (triangle inequality) BOTH runs are within band + that spread of the float64 run; asserted explicitly for the log."""
    if kind not in _F64:
        _F64[kind] = dict(load_golden(f"il_{kind}_f64"))
    r64 = _F64[kind][name].astype(np.float64)
    n64 = max(np.linalg.norm(r64), 1e-30)
    e_ref = np.linalg.norm(ref32 - r64) / n64
    e_hip = np.linalg.norm(mine - r64) / n64
    if _MOVE_MODE[0] == "collect":
        _ALT[name] = (np.array(mine, copy=True), e_hip)
        return
    alt, e_alt = _ALT[name]
    e_cond = np.linalg.norm(mine - alt) / n64
    band = max(MOVEMENT_FACTOR * e_ref, MOVEMENT_FLOOR)
    spread = KNIFE_EDGE_CAPS.get((kind, name), max(COND_FACTOR * e_ref, COND_FLOOR))
    _FLOWS_SEEN.add((kind, name.split("/")[0]))
    if (kind, name) in KNIFE_EDGE_CAPS:
        _CAPS_VISITED.add((kind, name))
    MOVEMENT_LOG.append(f"movement {kind} {name}: HIP vs f64 {e_hip:.3e} (alternate kernel {e_alt:.3e}), reference fp32 vs f64 {e_ref:.3e}, "
                        f"HIP kernel A vs B {e_cond:.3e}; band {band:.3e}, allowed spread {spread:.3e}")
    if os.environ.get("MRN_MOVEMENT_LOG"):          # (stdout is captured by the flows' own redirect: calibration runs log to a file)
        with open(os.environ["MRN_MOVEMENT_LOG"], "a") as f:
            f.write(MOVEMENT_LOG[-1] + "\n")
    if os.environ.get("MRN_MOVEMENT_CALIBRATE"):    # log every tensor of every flow without stopping at the first one out of band
        return
    assert min(e_hip, e_alt) <= band, MOVEMENT_LOG[-1]
    assert e_cond <= spread, MOVEMENT_LOG[-1]
    assert max(e_hip, e_alt) <= band + spread, MOVEMENT_LOG[-1]


def _twice(tmp_path, body):
    """run a flow on the alternate Winograd kernel first (movement collected as the conditioning yardstick, every other assertion of the
    flow applies to it too), then on the default kernel with the movement assertions"""
    from mrn_amd._lib import call
    _ALT.clear()
    _MOVE_MODE[0] = "collect"
    call("mrn_conv2d_x3_wino_select", 0)
    try:
        (tmp_path / "alt").mkdir()
        body(tmp_path / "alt")
    finally:
        call("mrn_conv2d_x3_wino_select", -1)
        _MOVE_MODE[0] = "assert"
    (tmp_path / "main").mkdir()
    body(tmp_path / "main")


def rel_close(a, b, rtol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.all(np.abs(a - b) <= rtol * np.maximum(1.0, np.abs(b))), (a, b)


@pytest.mark.parametrize("kind", ["crnn", "trba"])
@pytest.mark.parametrize("which", ["lwf", "ewc", "der"])
def test_il_flow_vs_reference(tmp_path, kind, which):
    _twice(tmp_path, lambda d: _il_flow(d, kind, which))


def test_il_flow_crnn_ewc_direct_products(tmp_path):
    """the same EWC flow with the trained convolutions on the DIRECT split-fp16 x3 products (MRN_TRAIN_WINO=0): the Fisher diagonal
    after two Adam steps keeps the 2 % band; the Winograd form of the trained convolutions (the default, above) is held to 4 %"""
    from mrn_amd import ops
    saved = (ops.TRAIN_WINO, ops.TRAIN_OPERAND_PEAK)
    ops.TRAIN_WINO, ops.TRAIN_OPERAND_PEAK = False, 16384.0
    try:
        _twice(tmp_path, lambda d: _il_flow(d, "crnn", "ewc", crnn_fisher_tol=0.02))
    finally:
        ops.TRAIN_WINO, ops.TRAIN_OPERAND_PEAK = saved


def _il_flow(tmp_path, kind, which, crnn_fisher_tol=0.04):
    """Two tasks of LwF / EWC / DER driven through incremental_train() / after_task() exactly like the reference learners
    were when the fixture was generated: per-iteration losses, KD terms, Fisher diagonals (incl. the positional blend of
    task 1), the reference's identically-zero EWC penalty, weight_align gamma, parameter movement, checkpoints written."""
    from mrn_amd.il_modules.der import DER
    from mrn_amd.il_modules.ewc import EWC
    from mrn_amd.il_modules.lwf import LwF
    g = load_golden(f"il_{kind}")
    pre = which + "/"
    os.chdir(tmp_path)
    opt = learner_opt(kind)
    cls = {"lwf": LwF, "ewc": EWC, "der": DER}[which]
    seeds = SEEDS[which]
    train = DetLoader(2, f"il:{kind}:{which}", 31)
    valid = DetLoader(2, f"il:{kind}:{which}:val", 32)
    sink = io.StringIO()
    steps, kd_vals, ewc_vals = [], [], []
    with contextlib.redirect_stdout(sink):
        learner = hooked(cls, seeds)(opt)
        if which == "ewc":
            learner.fisher_iterations = 2
        record(learner, "train_step", steps)
        if which == "lwf":
            record(learner, "kd_step", steps)
        if which == "ewc":
            record(learner, "ewc_step", steps)
            orig = learner.compute_ewc

            def compute():
                v = orig()
                ewc_vals.append(float(v))
                return v
            learner.compute_ewc = compute
        if which == "der":
            record(learner, "der_step", steps)
        for taski in range(2):
            chars = chars_upto(taski)
            train.set_characters(chars)
            valid.set_characters(chars)
            n0 = len(steps)
            learner.incremental_train(taski, chars, train, valid)
            got = steps[n0:]
            ref = g[f"{pre}t{taski}/losses"]
            # --- per-iteration losses (iteration 2 has seen one clipped Adam step) ---
            if which == "der" and taski == 1:
                mine = np.array([[clf, clf, aux] for clf, aux in got]).reshape(-1)      # Averagers: loss, loss_clf, loss_aux
            elif which == "lwf" and taski == 1:
                mine = np.array([s[0] for s in got])
                rel_close([s[1] for s in got], g[pre + "t1/kd"], 2e-4)
            else:
                mine = np.array([s[0] for s in got])
            rel_close(mine, ref, 2e-4)
            # --- parameter movement of the task's optimiser steps ---
            sd = {k.replace("module.", ""): v for k, v in learner.model.state_dict().items()}
            for k in [str(s) for s in g[f"{pre}t{taski}/param_keys"]]:
                init = torch.from_numpy(W.det_param(W.canonical_key(k), tuple(sd[k].shape), seeds[taski])).to(sd[k].device)
                name = f"{pre}t{taski}/delta/{k}"
                if float(np.abs(g[name + "/sub"]).max()) == 0.0:
                    assert float((sd[k] - init).abs().max()) == 0.0, k      # frozen (DER's old extractor)
                    continue
                moved = (sd[k] - init).detach().cpu().double().numpy().reshape(-1)
                step_ = max(1, moved.size // 1024)
                s = moved[::step_][:1024]
                r = g[name + "/sub"].astype(np.float64)
                # Adam normalises every element's update to ~lr, so elements whose gradient is ~0 move with an essentially
                # random sign (fp32 conditioning, see tests/helpers.py::assert_sub_l2): compare in L2 terms
                _assert_movement(kind, f"{pre}t{taski}/delta/{k}/sub", k, s, r)
            if which == "ewc":
                fk = [str(s) for s in g[f"{pre}t{taski}/fisher_keys"]]
                assert list(learner.fisher.keys()) == fk
                total = sum(float(v.double().sum()) for v in learner.fisher.values())
                rel_close(total, g[f"{pre}t{taski}/fisher_total"], 5e-2 if kind == "trba" else 5e-3)
                for k in [str(s) for s in g[f"{pre}t{taski}/param_keys"]]:
                    f = learner.fisher["module." + k]
                    assert float(f.max()) <= 1e-4 + 1e-12 and float(f.min()) >= 0.0
                    ref_f = g[f"{pre}t{taski}/fisher/{k}/sub"].astype(np.float64)
                    a = f.detach().cpu().double().numpy().reshape(-1)
                    step_ = max(1, a.size // 4096)
                    a = a[::step_][:4096]
                    l2 = np.linalg.norm(a - ref_f) / max(np.linalg.norm(ref_f), 1e-30)
                    # TRBA backbone / localisation-network gradients are ill-conditioned in fp32 (torch's own fp32-vs-fp64 gradient
                    # error is 2.5e-2 in the median, test_loop_a_trba_gradients_vs_oracle) and this Fisher is grad^2 AFTER two
                    # optimiser steps; the step-free Fisher of the same tensors is pinned at 0.1 by test_fisher_diagonal_vs_reference
                    loose = kind == "trba" and any(t in k for t in ("ConvNet", "Transformation"))
                    # CRNN: 4 % -- the first conv's Fisher sits behind every data-gradient convolution AND two Adam steps, whose
                    # sign-normalised updates amplify product-level differences (below 2 % with the direct x3 products, 2.6 % with the
                    # Winograd form of the trained convolutions; the step-free gradients stay within 2e-3 of the oracle,
                    # test_loop_a_crnn_gradients_vs_oracle, the step-free Fisher within 1 %, test_fisher_diagonal_vs_reference)
                    assert l2 <= (0.6 if loose else 0.15 if kind == "trba" else crnn_fisher_tol), (k, l2)
            learner.after_task()
            assert learner._known_classes == int(g[f"{pre}t{taski}/known_classes"])
            if which == "lwf" and taski == 0:
                W.fill_state_dict(learner._old_network.state_dict(), 27)
                assert learner._old_network.training == bool(g[pre + "old_network_training"])
    if which == "ewc":
        # reference quirk: Fisher keys carry "module.", compute_ewc looks names up without it -> the penalty is exactly 0
        assert ewc_vals == [float(v) for v in g[pre + "t1/compute_ewc"]] == [0.0, 0.0]
    if which == "der":
        gam = [float(l.split("=")[1]) for l in sink.getvalue().splitlines() if l.startswith("alignweights,gamma=")]
        rel_close(gam, g[pre + "t1/weight_align_gamma"], 1e-4)
        ref_keys = {str(k): str(s) for k, s in zip(g[pre + "sd_keys"], g[pre + "sd_shapes"])}
        assert {k: ",".join(map(str, v.shape)) for k, v in learner.model.state_dict().items()} == ref_keys
    assert sink.getvalue().count("Current_score") == int(g[pre + "n_valid_calls"])
    assert sorted(os.listdir(f"./saved_models/{opt.exp_name}")) == [str(s) for s in g[pre + "checkpoints"]]


@pytest.mark.parametrize("kind", ["crnn", "trba"])
def test_fisher_diagonal_vs_reference(tmp_path, kind):
    """EWC.getFisherDiagonal (ewc.py:128-167) on a freshly filled model: mean of squared gradients over 2 batches, clipped at 1e-4
    -- a pure function of the seeds (no optimiser step before it)"""
    from mrn_amd.il_modules.ewc import EWC
    g = load_golden(f"il_{kind}")
    os.chdir(tmp_path)
    opt = learner_opt(kind)
    with contextlib.redirect_stdout(io.StringIO()):
        learner = hooked(EWC, (41, 42))(opt)
        learner.fisher_iterations = 2
        learner.character = chars_upto(0)
        learner.converter = learner.build_converter()
        learner.criterion = learner.build_criterion()
        learner.build_model()
        learner.build_optimizer(learner.count_param())
        loader = DetLoader(2, f"il:{kind}:fisher", 33)
        loader.set_characters(chars_upto(0))
        fisher = learner.getFisherDiagonal(loader)
    assert [k.replace("module.", "") for k in fisher] == [str(s) for s in g["fisher0/all_keys"]]
    total = sum(float(v.double().sum()) for v in fisher.values())
    rel_close(total, g["fisher0/total"], 3e-2 if kind == "trba" else 2e-3)
    for k in [str(s) for s in g["fisher0/keys"]]:
        ref = g[f"fisher0/{k}/sub"].astype(np.float64)
        a = fisher["module." + k].detach().cpu().double().numpy().reshape(-1)
        step_ = max(1, a.size // 4096)
        a = a[::step_][:4096]
        l2 = np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30)
        assert l2 <= (0.1 if kind == "trba" else 0.01), (k, l2)


def build_dernet(opt, classes, seed=None, sd=None):
    from mrn_amd.modules.model import DERNet
    with contextlib.redirect_stdout(io.StringIO()):
        net = DERNet(opt)
        for c in classes:
            net.update_fc(opt.hidden_size, c)
            net.build_prediction(opt, c)
            net.build_aux_prediction(opt, c)
    if sd is not None:
        net.load_state_dict(sd, strict=True)
    else:
        W.fill_state_dict(net.state_dict(), seed)
    return net.cuda()


def test_trba_dernet_vs_reference():
    """DERNet over two TRBA extractors (BASELINE config 5's model at N = 2): state_dict layout, teacher-forced logits /
    auxiliary logits / concatenated features in DER's training configuration (old extractor eval, new one train), the two
    losses of a DER step, the clipped gradient norm and head gradients, greedy eval logits + argmax (bit-exact)"""
    from mrn_amd import functional as Fn
    from mrn_amd import ops
    from mrn_amd.tools.utils import AttnLabelConverter
    g = load_golden("trba_der2")
    opt = learner_opt("trba")
    classes, seed, B = (41, 71), 8, 2
    net = build_dernet(opt, classes, sd=golden_state_dict(g, seed))
    ref = {str(k): str(s) for k, s in zip(g["sd_keys"], g["sd_shapes"])}
    assert {k: ",".join(map(str, v.shape)) for k, v in net.state_dict().items()} == ref
    loader = DetLoader(B, "trba_der2", seed)
    loader.set_characters(chars_upto(1))
    image, words = loader.get_batch()
    with contextlib.redirect_stdout(io.StringIO()):
        conv = AttnLabelConverter(chars_upto(1))
    labels_index, _ = conv.encode(words, batch_max_length=25)
    assert np.array_equal(labels_index.cpu().numpy(), g["labels_index"])
    for ext in list(net.model)[:-1]:
        for p in ext.parameters():
            p.requires_grad = False
    net.train()
    net.model[0].eval()
    out = net(image.cuda(), labels_index[:, :-1])
    assert_sub_close(g, "features", out["features"], atol=1e-4)
    assert_sub_close(g, "logits", out["logits"], atol=1e-4)
    assert_sub_close(g, "aux_logits", out["aux_logits"], atol=1e-4)
    loss_clf = Fn.cross_entropy(out["logits"], labels_index[:, 1:], 1)
    loss_aux = Fn.cross_entropy(out["aux_logits"].detach(), labels_index[:, 1:], 1)
    assert abs(loss_clf.item() - float(g["loss_clf"])) < 1e-4 * max(1.0, float(g["loss_clf"]))
    assert abs(loss_aux.item() - float(g["loss_aux"])) < 1e-4 * max(1.0, float(g["loss_aux"]))
    loss_clf.backward()
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    norm = torch.sqrt(sum((gr.double() ** 2).sum() for gr in grads)).item()
    assert abs(norm - float(g["grad_norm"])) <= 2e-2 * float(g["grad_norm"])       # (TRBA backbone gradients: fp32 conditioning)
    for k in ("fc.weight", "Prediction.attention_cell.i2h.weight", "Prediction.attention_cell.rnn.weight_ih",
              "model.1.SequenceModeling.1.linear.weight"):
        assert_sub_close(g, f"grad/{k}", net.get_parameter(k).grad, atol=1e-7, rtol=5e-3)
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in net.aux_Prediction.attention_cell.parameters())
    net.load_state_dict(golden_state_dict(g, seed), strict=True)
    net.eval()
    with torch.no_grad():
        sos = torch.LongTensor(B).fill_(conv.dict["[SOS]"]).cuda()
        oe = net(image.cuda(), sos, False)
    assert_sub_close(g, "eval/logits", oe["logits"], atol=1e-4)
    assert np.array_equal(ops.argmax_lastdim(oe["logits"]).cpu().numpy(), g["eval/argmax"])


def test_full_size_dernet_trba6_properties():
    """BASELINE config 5 at full size: DERNet over SIX TRBA extractors, 256 images -- five frozen extractors in lock-step
    (grouped x3 convolutions, eval-mode BatchNorm) + the newest one, main attention head over the 1536-wide concatenation.
    Size-independent properties: (1) lock-step frozen extractors == per-extractor path; (2) each extractor's slice of
    `features` equals that extractor run alone; (3) the auxiliary head sees exactly the newest 256 channels; (4) a DER step
    moves only the newest extractor and the heads, and the frozen extractors' BatchNorm statistics do not move."""
    from mrn_amd import functional as Fn
    opt = learner_opt("trba")
    classes = (2091, 2311, 4039, 5199, 5272, 5374)
    B = 256
    image = torch.from_numpy(W.uniform("full_der", (B, 4, 32, 256), -1.0, 1.0, 3)).cuda()
    text = torch.from_numpy(W.randint("full_der_text", (B, 27), 4, classes[-1], 3)).cuda()
    text[:, 0] = 2
    feats = []
    for grouping in (True, False):
        net = build_dernet(opt, classes, seed=29)
        for ext in list(net.model)[:-1]:
            for p in ext.parameters():
                p.requires_grad = False
        net.train()
        for ext in list(net.model)[:-1]:
            ext.eval()
        net.expert_grouping = grouping
        with torch.no_grad():
            o = net(image, text[:, :-1])
        torch.cuda.synchronize()
        assert tuple(o["features"].shape) == (B, 65, 1536) and tuple(o["logits"].shape) == (B, 26, classes[-1])
        feats.append((o["features"].clone(), o["logits"].clone(), o["aux_logits"].clone()))
        if grouping:
            keep = net
        else:
            del net
    assert_close("full-size DER features (lock-step vs per extractor)", feats[0][0], feats[1][0], atol=2e-4, rtol=1e-4)
    assert_close("full-size DER logits", feats[0][1], feats[1][1], atol=2e-4, rtol=1e-4)
    net = keep
    with torch.no_grad():
        alone = net.model[2](image)                                   # one frozen extractor on its own
        assert_close("extractor 2 slice", feats[0][0][:, :, 512:768], alone, atol=2e-4, rtol=1e-4)
        aux = net.aux_Prediction(feats[0][0][:, :, -256:].contiguous(), text[:, :-1], True, batch_max_length=25)
        assert_close("aux head = newest 256 channels", aux, feats[0][2], atol=1e-5, rtol=1e-5)
    # one DER step at full size
    from mrn_amd.optim import FlatAdam
    frozen_before = {k: v.clone() for k, v in net.state_dict().items() if k.startswith(("model.0.", "model.3."))}
    adam = FlatAdam([p for p in net.parameters() if p.requires_grad], lr=5e-4)
    adam.zero_grad()
    out = net(image, text[:, :-1])
    loss = Fn.cross_entropy(out["logits"], text[:, 1:], 1)
    loss.backward()
    nc = adam.step(lr=2.5e-5, max_norm=5.0)
    assert torch.isfinite(loss).item() and torch.isfinite(nc).all().item() and float(nc[0]) > 0
    after = net.state_dict()
    for k, v in frozen_before.items():
        assert torch.equal(v, after[k]), k
    assert float((net.aux_fc.weight.grad if net.aux_fc.weight.grad is not None else torch.zeros(1)).abs().max()) == 0.0


def test_der6_batch32_full_class_counts_vs_oracle():
    """BASELINE config 5 at the bench's size against the CPU ORACLE (not another HIP schedule): DERNet over SIX TRBA extractors, class
    counts 2091 ... 5374, 32 smooth crops, DER's training configuration (five frozen extractors in eval mode on the lock-step path, the
    newest one in train mode, main head over the 1536-wide concatenation, auxiliary head on the newest 256 channels).  Features, logits,
    auxiliary logits and the classification loss within 1e-4 of the fp32 oracle -- or within 3 x the oracle's own distance from float64
    arithmetic where that is larger (the TPS grid's conditioning) --, gradients of the heads within 5e-3, teacher-forced greedy indices
    bit-exact wherever the float64 top-2 margin clears the band."""
    from mrn_amd import functional as Fn
    from oracle import mrn_oracle as O
    from tests.helpers import oracle_dtype
    opt = learner_opt("trba")
    classes = (2091, 2311, 4039, 5199, 5272, 5374)
    B, N = 32, len(classes)
    net = build_dernet(opt, classes, seed=31)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for ext in list(net.model)[:-1]:
        for p in ext.parameters():
            p.requires_grad = False
    net.train()
    for ext in list(net.model)[:-1]:
        ext.eval()
    image = torch.from_numpy(W.smooth_image("b32_der", (B, 4, 32, 256), 7))
    text = torch.from_numpy(W.randint("b32_der_text", (B, 27), 4, classes[-1], 7))
    text[:, 0] = 2
    cfg = O.Cfg("TPS", "ResNet", "BiLSTM", "Attn")
    head_names = ["fc.weight", "fc.bias", "Prediction.attention_cell.i2h.weight", "Prediction.attention_cell.h2h.weight",
                  "Prediction.attention_cell.rnn.weight_ih", "Prediction.attention_cell.rnn.weight_hh",
                  f"model.{N - 1}.SequenceModeling.1.linear.weight"]
    sd32 = {k: v.clone() for k, v in sd.items()}
    for n in head_names:
        sd32[n].requires_grad_(True)
    ref = O.dernet_forward(sd32, cfg, N, image, text[:, :-1], True, training=True)
    loss32 = O.attn_ce_loss(ref["logits"], text)
    g32 = torch.autograd.grad(loss32, [sd32[n] for n in head_names])
    with oracle_dtype(torch.float64) as od, torch.no_grad():
        ref64 = O.dernet_forward(od.cast(sd), cfg, N, image.double(), text[:, :-1], True, training=True)
    out = net(image.cuda(), text[:, :-1].cuda())
    loss = Fn.cross_entropy(out["logits"], text[:, 1:].cuda(), 1)
    loss.backward()
    for key in ("features", "logits", "aux_logits"):
        r32 = ref[key].detach()
        band = float((r32.double() - ref64[key]).abs().max())
        err = float((out[key].detach().cpu() - r32).abs().max())
        scale = float(r32.abs().max())
        assert err <= max(1e-4 * max(1.0, scale), 3 * band), (key, err, band, scale)
    assert abs(float(loss.detach()) - float(loss32.detach())) <= 1e-4 * max(1.0, abs(float(loss32.detach())))
    t2 = ref64["logits"].topk(2, dim=2)[0]
    band_l = float((ref["logits"].detach().double() - ref64["logits"]).abs().max())
    clear = (t2[..., 0] - t2[..., 1]) > 10 * max(band_l, 1e-5)
    assert float(clear.float().mean()) > 0.9
    assert torch.equal(out["logits"].detach().cpu().argmax(2)[clear], ref["logits"].detach().argmax(2)[clear])
    for n, gr in zip(head_names, g32):
        mine = net.get_parameter(n).grad.detach().cpu().double()
        scale = max(float(gr.abs().max()), 1e-12)
        rel = float((mine - gr.double()).norm() / max(float(gr.double().norm()), 1e-12))
        assert rel <= 5e-3, (n, rel, scale)
    # the frozen extractors took the eval path: their BatchNorm statistics did not move
    after = net.state_dict()
    for k in ("model.0.FeatureExtraction.ConvNet.bn0_1.running_mean", "model.4.FeatureExtraction.ConvNet.bn4_2.running_var"):
        assert torch.equal(after[k].cpu(), sd[k]), k


SEEDS2 = {"wa": (51, 52), "joint": (53, 54)}


def _check_movement(g, pre, taski, learner, seeds, kind):
    sd = {k.replace("module.", ""): v for k, v in learner.model.state_dict().items()}
    for k in [str(s) for s in g[f"{pre}t{taski}/param_keys"]]:
        init = torch.from_numpy(W.det_param(W.canonical_key(k), tuple(sd[k].shape), seeds[taski])).to(sd[k].device)
        moved = (sd[k] - init).detach().cpu().double().numpy().reshape(-1)
        step_ = max(1, moved.size // 1024)
        s = moved[::step_][:1024]
        r = g[f"{pre}t{taski}/delta/{k}/sub"].astype(np.float64)
        _assert_movement(kind, f"{pre}t{taski}/delta/{k}/sub", k, s, r)


@pytest.mark.parametrize("kind", ["crnn", "trba"])
def test_wa_flow_vs_reference(tmp_path, kind):
    _twice(tmp_path, lambda d: _wa_flow(d, kind))


def _wa_flow(tmp_path, kind):
    """Two tasks of the WA learner (reference il_modules/wa.py:29-116) driven like the reference class was for
    tests/golden/il2_*.npz (make_golden_il2.py): per-iteration losses (loss_clf + 2 * KD in task 1), the KD terms, BOTH
    weight_align() calls of task 1 (end of _update_representation: gamma and the rescaled classifier rows; after_task(): gamma 1
    on the already aligned rows), parameter movement, class bookkeeping, checkpoints, validations."""
    from mrn_amd.il_modules.wa import WA
    g = load_golden(f"il2_{kind}")
    pre = "wa/"
    os.chdir(tmp_path)
    opt = learner_opt(kind)
    seeds = SEEDS2["wa"]
    train = DetLoader(2, f"il2:{kind}:wa", 61)
    valid = DetLoader(2, f"il2:{kind}:wa:val", 62)
    sink = io.StringIO()
    steps, fc_snaps = [], []
    with contextlib.redirect_stdout(sink):
        learner = hooked(WA, seeds)(opt)
        record(learner, "train_step", steps)
        record(learner, "kd_step", steps)
        for taski in range(2):
            chars = chars_upto(taski)
            train.set_characters(chars)
            valid.set_characters(chars)
            n0 = len(steps)
            if taski == 1:
                net = learner.model.module if hasattr(learner.model, "module") else learner.model
            learner.incremental_train(taski, chars, train, valid)
            got = steps[n0:]
            rel_close([s[0] for s in got], g[f"{pre}t{taski}/losses"], 2e-4)
            if taski == 1:
                rel_close([s[1] for s in got], g[pre + "t1/kd"], 2e-4)
                fc_snaps.append(learner.model.module.fc.weight.detach().cpu().numpy().copy()[::7, ::5])
            _check_movement(g, pre, taski, learner, seeds, kind)
            learner.after_task()
            if taski == 1:
                fc_snaps.append(learner.model.fc.weight.detach().cpu().numpy().copy()[::7, ::5])
            assert learner._known_classes == int(g[f"{pre}t{taski}/known_classes"])
            if taski == 0:
                W.fill_state_dict(learner._old_network.state_dict(), 57)
                assert learner._old_network.training == bool(g[pre + "old_network_training"])
    gam = [float(l.split("=")[1]) for l in sink.getvalue().splitlines() if l.startswith("alignweights,gamma=")]
    rel_close(gam, g[pre + "t1/weight_align_gamma"], 1e-4)
    # classifier rows after each alignment (the trained rows carry the Adam-step conditioning of _check_movement: compare the
    # alignment itself through the ratio after / before, and the aligned matrix loosely)
    for i, snap in enumerate(fc_snaps):
        ref_after = g[f"{pre}t1/align{i}/fc_after"]
        assert snap.shape == ref_after.shape
        assert np.abs(snap - ref_after).max() <= 2e-4 * max(1.0, np.abs(ref_after).max())
    assert sink.getvalue().count("Current_score") == int(g[pre + "n_valid_calls"])
    assert sorted(os.listdir(f"./saved_models/{opt.exp_name}")) == [str(s) for s in g[pre + "checkpoints"]]


@pytest.mark.parametrize("kind", ["crnn", "trba"])
def test_joint_flow_vs_reference(tmp_path, kind):
    _twice(tmp_path, lambda d: _joint_flow(d, kind))


def _joint_flow(tmp_path, kind):
    """Two rounds of JointLearner (reference il_modules/joint.py:9-105; the second after change_model() grew the classifier):
    per-iteration losses, parameter movement, checkpoints, validations, and the empty score lists incremental_train() returns when
    no test interval is reached"""
    from mrn_amd.il_modules.joint import JointLearner
    g = load_golden(f"il2_{kind}")
    pre = "joint/"
    os.chdir(tmp_path)
    opt = learner_opt(kind)
    opt.saved_model = ""
    seeds = SEEDS2["joint"]
    train = DetLoader(2, f"il2:{kind}:joint", 63)
    valid = DetLoader(2, f"il2:{kind}:joint:val", 64)
    sink = io.StringIO()
    steps = []
    with contextlib.redirect_stdout(sink):
        learner = hooked(JointLearner, seeds)(opt)
        record(learner, "train_step", steps)
        for taski in range(2):
            chars = chars_upto(taski)
            train.set_characters(chars)
            valid.set_characters(chars)
            n0 = len(steps)
            best, ned = learner.incremental_train(taski, chars, train, valid, None, None)
            assert [len(best), len(ned)] == list(g[f"{pre}t{taski}/returned_lengths"])
            rel_close([s[0] for s in steps[n0:]], g[f"{pre}t{taski}/losses"], 2e-4)
            _check_movement(g, pre, taski, learner, seeds, kind)
            learner.after_task()
            assert learner._known_classes == int(g[f"{pre}t{taski}/known_classes"])
    assert sink.getvalue().count("Current_score") == int(g[pre + "n_valid_calls"])
    assert sorted(os.listdir(f"./saved_models/{opt.exp_name}")) == [str(s) for s in g[pre + "checkpoints"]]


def test_zz_movement_assertions_were_live():
    """runs last in this file: the calibration switch turns _assert_movement into a logger, so a suite run with it set proves nothing
     -- fail loudly; and every tensor exempted BY NAME in KNIFE_EDGE_CAPS must have been met by the flow it names, or a
    renamed tensor has silently lost (or kept) its exemption"""
    assert not os.environ.get("MRN_MOVEMENT_CALIBRATE"), "MRN_MOVEMENT_CALIBRATE is set: the movement assertions of this run were skipped"
    for kind, name in KNIFE_EDGE_CAPS:
        if (kind, name.split("/")[0]) in _FLOWS_SEEN:
            assert (kind, name) in _CAPS_VISITED, f"KNIFE_EDGE_CAPS entry {(kind, name)} matched no tensor of its flow"
