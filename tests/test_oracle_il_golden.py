"""Pin the CPU oracle's BASELINE-config-5 pieces (DERNet, LwF step loss, EWC Fisher diagonal) against fixtures produced by
the REFERENCE learners (tests/golden/make_golden_il.py).  CPU only; the first iteration of every flow is a pure function
of the seeds, so it is compared tightly."""
import contextlib
import io
import types

import numpy as np
import pytest
import torch

from mrn_amd.tools import weights as W
from oracle import mrn_oracle as O
from tests.helpers import DetLoader, assert_sub_close, golden_state_dict, load_golden

NCHARS = (36, 30)


def chars_upto(t):
    return "".join(chr(0x4E00 + i) for i in range(sum(NCHARS[:t + 1])))


def opt_for(kind):
    o = types.SimpleNamespace(num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512, hidden_size=256,
                              batch_max_length=25)
    if kind == "crnn":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "VGG", "BiLSTM", "CTC"
    else:
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "TPS", "ResNet", "BiLSTM", "Attn"
    return o


def cfg_for(kind):
    return O.Cfg("None", "VGG", "BiLSTM", "CTC") if kind == "crnn" else O.Cfg("TPS", "ResNet", "BiLSTM", "Attn")


def model_state(kind, classes, seed):
    """state_dict of a single recogniser (reference `Model`) with deterministic values; the key layout comes from the product's
    parameter containers, which test_host_cpu pins against the reference's"""
    from mrn_amd.modules.model import Model
    opt = opt_for(kind)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Model(opt)
        net.update_fc(opt.hidden_size, classes)
        net.build_prediction(opt, classes)
    sd = {k: v.clone() for k, v in W.fill_state_dict(net.state_dict(), seed).items()}
    return sd, [n for n, _ in net.named_parameters()]


def encode(kind, words, chars):
    conv = O.CTCConverter(chars) if kind == "crnn" else O.AttnConverter(chars)
    return conv.encode(words, 25)


def first_loss(kind, sd, image, words, chars, training=True):
    li, ll = encode(kind, words, chars)
    out = O.model_forward(sd, "", cfg_for(kind), image, None if kind == "crnn" else li[:, :-1], True, training=training)["predict"]
    loss = O.ctc_loss(out, li, ll) if kind == "crnn" else O.attn_ce_loss(out, li)
    return out, loss, li, ll


@pytest.mark.parametrize("kind", ["crnn", "trba"])
def test_first_iteration_losses_of_the_reference_flows(kind):
    g = load_golden(f"il_{kind}")
    nsp = 4 if kind == "crnn" else 5
    c0, c1 = NCHARS[0] + nsp, NCHARS[0] + NCHARS[1] + nsp
    with torch.no_grad():
        # task 0, iteration 1 of LwF / EWC (BaseLearner._init_train, base.py:226-250) and DER (der.py:151-183)
        for which, seed in (("lwf", 21), ("ewc", 23)):
            sd, _ = model_state(kind, c0, seed)
            loader = DetLoader(2, f"il:{kind}:{which}", 31)
            loader.set_characters(chars_upto(0))
            image, words = loader.get_batch()
            _, loss, _, _ = first_loss(kind, sd, image, words, chars_upto(0))
            ref = float(g[f"{which}/t0/losses"][0])
            assert abs(float(loss) - ref) <= 1e-4 * max(1.0, abs(ref)), (which, float(loss), ref)
        # task 1, iteration 1 of LwF: lamda * KD against the frozen (eval-mode) previous network + loss_clf (lwf.py:63-87)
        sd_new, _ = model_state(kind, c1, 22)
        sd_old, _ = model_state(kind, c0, 27)
        loader = DetLoader(2, f"il:{kind}:lwf", 31)
        loader.set_characters(chars_upto(1))
        loader.count = 2                                     # task 0 consumed two training batches
        image, words = loader.get_batch()
        new, _, li, ll = first_loss(kind, sd_new, image, words, chars_upto(1))
        old = O.model_forward(sd_old, "", cfg_for(kind), image, None if kind == "crnn" else li[:, :-1], True, training=False)["predict"]
        total, kd, _ = O.lwf_step_loss(new, old, li, ll, "CTC" if kind == "crnn" else "Attn", c0)
        assert tuple(g["lwf/kd_shape"]) == (new.shape[0] * new.shape[1], c0 - (0 if kind == "crnn" else 1))
        assert abs(float(kd) - float(g["lwf/t1/kd"][0])) <= 1e-4 * max(1.0, float(g["lwf/t1/kd"][0]))
        assert abs(float(total) - float(g["lwf/t1/losses"][0])) <= 1e-4 * max(1.0, float(g["lwf/t1/losses"][0]))
    assert [float(v) for v in g["ewc/t1/compute_ewc"]] == [0.0, 0.0]       # the reference's penalty is identically zero


@pytest.mark.parametrize("kind", ["crnn", "trba"])
def test_fisher_diagonal(kind):
    """ewc.py:128-167 through autograd on the oracle: two batches, squared gradients averaged, clipped at 1e-4"""
    g = load_golden(f"il_{kind}")
    nsp = 4 if kind == "crnn" else 5
    sd, names = model_state(kind, NCHARS[0] + nsp, 41)
    assert names == [str(s) for s in g["fisher0/all_keys"]]
    for k in list(sd):                                        # Prediction.generator.* aliases fc.* (one tensor, two keys)
        if "Prediction.generator." in k:
            sd[k] = sd[k.replace("Prediction.generator.", "fc.")]
        if k in ("Prediction.weight", "Prediction.bias"):
            sd[k] = sd["fc." + k.split(".")[1]]
    params = [sd[n].requires_grad_(True) for n in names]
    loader = DetLoader(2, f"il:{kind}:fisher", 33)
    loader.set_characters(chars_upto(0))
    grads = []
    for _ in range(2):
        image, words = loader.get_batch()
        _, loss, _, _ = first_loss(kind, sd, image, words, chars_upto(0))
        grads.append([gr if gr is not None else torch.zeros_like(p)
                      for gr, p in zip(torch.autograd.grad(loss, params, allow_unused=True), params)])
    fisher = dict(zip(names, O.fisher_diagonal(grads)))
    total = sum(float(f.double().sum()) for f in fisher.values())
    ref = float(g["fisher0/total"])
    assert abs(total - ref) <= (2e-2 if kind == "trba" else 1e-4) * ref, (total, ref)
    for k in [str(s) for s in g["fisher0/keys"]]:
        r = g[f"fisher0/{k}/sub"].astype(np.float64)
        a = fisher[k].detach().double().numpy().reshape(-1)
        a = a[::max(1, a.size // 4096)][:4096]
        l2 = np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-30)
        assert l2 <= (0.1 if kind == "trba" else 1e-3), (k, l2)      # (TRBA: fp32 summation order through 29 convs, see test_model_gpu)


def test_fisher_blend_and_penalty_formulas():
    old = [torch.tensor([1.0, 2.0]), torch.tensor([[1.0, 1.0]])]
    new = [torch.tensor([3.0, 4.0, 5.0]), torch.tensor([[3.0, 3.0], [7.0, 7.0]])]
    b = O.fisher_blend(old, new)
    assert torch.equal(b[0], torch.tensor([2.0, 3.0, 5.0])) and torch.equal(b[1], torch.tensor([[2.0, 2.0], [7.0, 7.0]]))
    pen = O.ewc_penalty([torch.tensor([2.0, 4.0])], [torch.tensor([1.0, 3.0, 9.0])], [torch.tensor([0.0, 1.0])])
    assert float(pen) == (2.0 * 1.0 + 4.0 * 4.0) / 2


def test_trba_dernet():
    """DERNet over two TRBA extractors, DER's training configuration (der.py:39-44) and greedy evaluation"""
    g = load_golden("trba_der2")
    sd = golden_state_dict(g, 8)
    cfg = cfg_for("trba")
    loader = DetLoader(2, "trba_der2", 8)
    loader.set_characters(chars_upto(1))
    image, words = loader.get_batch()
    li, _ = O.AttnConverter(chars_upto(1)).encode(words, 25)
    assert np.array_equal(li.numpy(), g["labels_index"])
    with torch.no_grad():
        out = O.dernet_forward(dict(sd), cfg, 2, image, li[:, :-1], True, training=True, old_eval=True)
        assert_sub_close(g, "features", out["features"], atol=1e-4)
        assert_sub_close(g, "logits", out["logits"], atol=1e-4)
        assert_sub_close(g, "aux_logits", out["aux_logits"], atol=1e-4)
        for name, key in (("logits", "loss_clf"), ("aux_logits", "loss_aux")):
            loss = O.attn_ce_loss(out[name], li)
            assert abs(float(loss) - float(g[key])) <= 1e-4 * float(g[key])
        oe = O.dernet_forward(dict(golden_state_dict(g, 8)), cfg, 2, image, torch.LongTensor(2).fill_(2), False, training=False)
        assert_sub_close(g, "eval/logits", oe["logits"], atol=1e-4)
        assert np.array_equal(oe["logits"].max(2)[1].numpy(), g["eval/argmax"])


@pytest.mark.parametrize("kind", ["crnn", "trba"])
def test_first_iteration_losses_of_wa_and_joint(kind):
    """tests/golden/il2_*.npz (the reference's WA and JointLearner driven by make_golden_il2.py) against the oracle: iteration 1 of
    every task is a pure function of the seeds.  WA task 1: loss_clf + 2 * KD (wa.py:81-88, the LwF step with lamda = 2); Joint:
    the canonical loss of base.py:242-250 on a freshly grown classifier."""
    g = load_golden(f"il2_{kind}")
    nsp = 4 if kind == "crnn" else 5
    c0, c1 = NCHARS[0] + nsp, NCHARS[0] + NCHARS[1] + nsp
    pred = "CTC" if kind == "crnn" else "Attn"
    with torch.no_grad():
        for which, tag, seeds, lseed in (("wa", f"il2:{kind}:wa", (51, 52), 61), ("joint", f"il2:{kind}:joint", (53, 54), 63)):
            sd, _ = model_state(kind, c0, seeds[0])
            loader = DetLoader(2, tag, lseed)
            loader.set_characters(chars_upto(0))
            image, words = loader.get_batch()
            _, loss, _, _ = first_loss(kind, sd, image, words, chars_upto(0))
            ref = float(g[f"{which}/t0/losses"][0])
            assert abs(float(loss) - ref) <= 1e-4 * max(1.0, abs(ref)), (which, float(loss), ref)
            # task 1, iteration 1 (two training batches consumed by task 0)
            sd_new, _ = model_state(kind, c1, seeds[1])
            loader.set_characters(chars_upto(1))
            loader.count = 2
            image, words = loader.get_batch()
            new, loss1, li, ll = first_loss(kind, sd_new, image, words, chars_upto(1))
            ref1 = float(g[f"{which}/t1/losses"][0])
            if which == "joint":
                assert abs(float(loss1) - ref1) <= 1e-4 * max(1.0, abs(ref1)), (float(loss1), ref1)
                continue
            sd_old, _ = model_state(kind, c0, 57)
            old = O.model_forward(sd_old, "", cfg_for(kind), image, None if kind == "crnn" else li[:, :-1], True, training=False)["predict"]
            total, kd, _ = O.lwf_step_loss(new, old, li, ll, pred, c0, lamda=2.0)
            assert tuple(g["wa/kd_shape"]) == (new.shape[0] * new.shape[1], c0 - (0 if kind == "crnn" else 1))
            assert abs(float(kd) - float(g["wa/t1/kd"][0])) <= 1e-4 * max(1.0, float(g["wa/t1/kd"][0]))
            assert abs(float(total) - ref1) <= 1e-4 * max(1.0, abs(ref1)), (float(total), ref1)
    # weight_align (modules/model.py:166-174): gamma = mean old-row norm / mean new-row norm, reproduced from the stored rows
    for i in range(2):
        before, after = g[f"wa/t1/align{i}/fc_before"], g[f"wa/t1/align{i}/fc_after"]
        inc = int(g[f"wa/t1/align{i}/increment"])
        rows = np.arange(c1)[::7]
        new_rows = rows >= c1 - inc
        ratio = after[new_rows] / np.where(before[new_rows] == 0, 1, before[new_rows])
        assert np.allclose(ratio[before[new_rows] != 0], float(g["wa/t1/weight_align_gamma"][i]), rtol=2e-5)
        assert np.array_equal(after[~new_rows], before[~new_rows])
