"""Generate golden vectors by running the REFERENCE implementation (simplify23/MRN at /root/reference) on CPU.

Run in the build container only (the reference never travels to the GPU box):
    python tests/golden/make_golden.py
Writes tests/golden/*.npz.  Inputs and weights are NOT stored: they come from the deterministic generator
mrn_amd/tools/weights.py (value = f(name, shape, seed)), applied here to the reference modules' state_dict and,
in the tests, to the oracle / HIP-backed modules.  Only outputs of the reference are stored (large tensors as a
fixed strided subsample plus moments).

Harness shims: `timm` is absent -> stub `timm.models.layers.trunc_normal_`.
The optimiser step of loop B is driven with the same torch calls the reference learner makes
(il_modules/mrn.py:338-371: CrossEntropyLoss / CTCLoss, clip_grad_norm_, Adam, OneCycleLR(total=2*num_iter)).
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)

timm = types.ModuleType("timm")
timm.models = types.ModuleType("timm.models")
timm.models.layers = types.ModuleType("timm.models.layers")
timm.models.layers.trunc_normal_ = torch.nn.init.trunc_normal_
sys.modules.update({"timm": timm, "timm.models": timm.models, "timm.models.layers": timm.models.layers})

torch.Tensor.cuda = lambda self, *a, **k: self   # SVTR's local-mask constructor calls .cuda() (svtr.py:119,125)
import modules.svtr as ref_svtr  # noqa: E402  (reference)
from modules.model import MRNNet  # noqa: E402  (reference)

DROP_MASKS = []          # DropPath draws injected into the reference (svtr.py:17-22 uses the global torch RNG)


def _injected_drop_path(x, drop_prob=0., training=False, scale_by_keep=True):
    if drop_prob == 0. or not training:
        return x
    keep = 1 - drop_prob
    m = DROP_MASKS.pop(0).view((x.shape[0],) + (1,) * (x.ndim - 1)).to(x.dtype)
    return x * (m / keep if keep > 0.0 and scale_by_keep else m)


ref_svtr.drop_path = _injected_drop_path


def drop_masks(B, seed, tag, n_experts=1):
    """22 draws per SVTR expert forward (11 blocks with drop_prob > 0, two sites each), Bernoulli(0.5) to exercise both arms"""
    return [[torch.from_numpy(W.randint(f"droppath:{tag}:{e}:{k}", (B,), 0, 2, seed)).float() for k in range(22)]
            for e in range(n_experts)]
from tools.utils import AttnLabelConverter, CTCLabelConverter  # noqa: E402  (reference)

from mrn_amd.tools import weights as W  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
torch.set_num_threads(8)


def sub(t, n=4096):
    """fixed strided subsample + moments of a tensor"""
    a = t.detach().cpu().double().numpy().reshape(-1)
    step = max(1, a.size // n)
    return {"sub": a[::step][:n].astype(np.float32), "mean": np.float64(a.mean()), "absmean": np.float64(np.abs(a).mean()),
            "shape": np.array(t.shape, dtype=np.int64)}


def put(d, name, t, full=False):
    if full:
        d[name] = t.detach().cpu().numpy().copy()
    else:
        for k, v in sub(t).items():
            d[f"{name}/{k}"] = v


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


def words_for(B, nchar, seed):
    """synthetic labels over a character set 'chr(0x4e00+i)'; returns (words, character string)"""
    chars = "".join(chr(0x4E00 + i) for i in range(nchar))
    lens = W.randint("label_len", (B,), 1, 26, seed)
    out = []
    for b in range(B):
        ids = W.randint(f"label_{b}", (int(lens[b]),), 0, nchar, seed)
        out.append("".join(chars[i] for i in ids))
    return out, chars


def build(kind, classes, seed):
    opt = make_opt(kind)
    net = MRNNet(opt)
    for c in classes:
        net.update_fc(opt.hidden_size, c)
        net.build_prediction(opt, c)
    W.fill_state_dict(net.state_dict(), seed)
    return opt, net


def run(kind, classes, B, seed, noise=False):
    """noise: U(-1,1) white-noise crops -- the distribution bench.py runs (mrn_amd/data/synthetic.py) -- instead of smooth ones"""
    d = {}
    opt, net = build(kind, classes, seed)
    I = len(classes)
    sd0 = net.state_dict()
    d["sd_keys"] = np.array(sorted(sd0.keys()))                      # pins the reference's state_dict layout
    d["sd_shapes"] = np.array([",".join(map(str, sd0[k].shape)) for k in sorted(sd0.keys())])
    if noise:
        image = torch.from_numpy(W.uniform("input:image", (B, 4, 32, 256), -1.0, 1.0, seed))
    else:
        image = torch.from_numpy(W.smooth_image("input:image", (B, 4, 32, 256), seed))
    nspecial = 4 if kind in ("crnn", "svtr") else 5
    words, chars = words_for(B, classes[-1] - nspecial, seed)
    ctc = kind in ("crnn", "svtr")
    conv = CTCLabelConverter(chars) if ctc else AttnLabelConverter(chars)
    labels_index, labels_length = conv.encode(words, batch_max_length=25)
    d["labels_index"] = labels_index.numpy()
    d["labels_length"] = labels_length.numpy()
    text = None if ctc else labels_index[:, :-1]

    def inject(tag, n=1):
        DROP_MASKS.clear()
        if kind == "svtr":
            for e in drop_masks(B, seed, tag, n):
                DROP_MASKS.extend(e)
    domain = torch.from_numpy(W.randint("domain", (B,), 0, 2, seed))

    # ---- per-stage outputs of expert 0, train-mode BN (batch statistics) --------------------------------
    net.train()
    m0 = net.model[0]
    with torch.no_grad():
        x = image
        if kind == "trba":
            tps = m0.model.Transformation
            cp = tps.LocalizationNetwork(x)
            put(d, "e0/tps_cprime", cp, full=True)
            put(d, "tps/inv_delta_C", tps.GridGenerator.inv_delta_C, full=True)
            put(d, "tps/P_hat", tps.GridGenerator.P_hat)
            x = tps(image)          # second pass also moves the BN running stats a second time; restore below
            put(d, "e0/tps_out", x)
        inject("featmap")
        fm = m0.model.FeatureExtraction(x)
        put(d, "e0/featmap", fm)
    # reset weights/buffers (running stats were touched) and run the real forward paths
    W.fill_state_dict(net.state_dict(), seed)
    with torch.no_grad():
        inject("e0")
        o = m0(image, text, True)
        put(d, "e0/feature", o["feature"])
        put(d, "e0/predict", o["predict"])
        put(d, "e0/bn_running_mean_after", _first_bn(net, "running_mean"), full=True)
        put(d, "e0/bn_running_var_after", _first_bn(net, "running_var"), full=True)
    W.fill_state_dict(net.state_dict(), seed)

    # ---- eval-mode expert forward (running statistics; Attn decodes greedily) ---------------------------
    net.eval()
    with torch.no_grad():
        sos = None if ctc else torch.LongTensor(B).fill_(2)   # test.py:186-189
        o = m0(image, sos, False)
        put(d, "e0_eval/feature", o["feature"])
        put(d, "e0_eval/predict", o["predict"])
        d["e0_eval/argmax"] = o["predict"].max(2)[1].numpy()
        oe = net(image, True, sos, False)      # cross_forward_expert
        d["eval/index"] = oe["index"].numpy()
        put(d, "eval/logits", oe["logits"])
        d["eval/argmax"] = oe["logits"].max(2)[1].numpy()
        if ctc:
            am = oe["logits"].max(2)[1]
            d["eval/ctc_strings"] = np.array(conv.decode(am.numpy(), [am.shape[1]] * B))

    # ---- loop B (il_modules/mrn.py:323-371): experts in train mode, router trained ---------------------
    net.train()
    for i in range(I):
        for p in net.model[i].parameters():
            p.requires_grad = False
    params = [p for p in net.parameters() if p.requires_grad]
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    d["router_param_names"] = np.array(names)
    opt_ = torch.optim.Adam(params, lr=0.0005)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt_, max_lr=0.0005, cycle_momentum=False, div_factor=20,
                                                final_div_factor=1000, total_steps=20 * 2)
    taski_crit = torch.nn.CrossEntropyLoss(reduction="mean")
    if ctc:
        crit = torch.nn.CTCLoss(reduction="mean", zero_infinity=True)
    else:
        crit = torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=conv.dict["[PAD]"])
    before = {n: p.detach().clone() for n, p in zip(names, params)}
    for it in range(2):
        inject(f"stepB{it}", I)
        if ctc:
            out = net(image, True)
            preds = out["logits"]
            taski = taski_crit(out["index"], domain)
            preds_size = torch.IntTensor([preds.size(1)] * B)
            clf = crit(preds.log_softmax(2).permute(1, 0, 2), labels_index, preds_size, labels_length)
        else:
            out = net(image, cross=True, text=labels_index[:, :-1], is_train=True)
            preds = out["logits"]
            taski = taski_crit(out["index"], domain)
            target = labels_index[:, 1:]
            clf = crit(preds.view(-1, preds.shape[-1]), target.contiguous().view(-1))
        loss = 15 * clf + taski
        net.zero_grad()
        loss.backward()
        total_norm = torch.nn.utils.clip_grad_norm_(net.parameters(), 5)
        if it == 0:
            put(d, "stepB/weights", out["index"], full=True)
            put(d, "stepB/logits", preds)
            d["stepB/loss_clf"] = np.float64(clf.item())
            d["stepB/loss_taski"] = np.float64(taski.item())
            d["stepB/grad_norm"] = np.float64(total_norm.item())
            for n, p in zip(names, params):     # grads AFTER clipping, as the optimiser sees them
                put(d, f"stepB/grad/{n}", p.grad)
        opt_.step()
        sched.step()
        d[f"stepB/lr_after_{it}"] = np.float64(opt_.param_groups[0]["lr"])
    for n, p in zip(names, params):
        put(d, f"stepB/delta2/{n}", p.detach() - before[n])
    d["stepB/loss_clf_1"] = np.float64(clf.item())
    d["stepB/loss_taski_1"] = np.float64(taski.item())

    # ---- loop A forward + loss on the newest expert (il_modules/mrn.py:247-258), no update ---------------
    W.fill_state_dict(net.state_dict(), seed)
    net.train()
    with torch.no_grad():
        inject("stepA")
        if ctc:
            preds = net(image, False)["logits"]
            lossA = crit(preds.log_softmax(2).permute(1, 0, 2), labels_index, torch.IntTensor([preds.size(1)] * B), labels_length)
        else:
            preds = net(image, False, labels_index[:, :-1])["logits"]
            lossA = crit(preds.view(-1, preds.shape[-1]), labels_index[:, 1:].contiguous().view(-1))
        put(d, "stepA/logits", preds)
        d["stepA/loss"] = np.float64(lossA.item())
    return d


def _first_bn(net, leaf):
    for k, v in net.state_dict().items():
        if k.startswith("model.0.") and k.endswith(leaf):
            return v
    raise KeyError(leaf)


def der(kind, classes, B, seed):
    """DERNet with two extractors (reference modules/model.py:203-312): forward in the training configuration of
    DER._update_representation (old extractor eval, new one train), weight_align gamma, LwF KD loss on the logits."""
    from modules.model import DERNet

    def _KD_loss(pred, soft, T):         # the three torch calls of il_modules/lwf.py:111-114 (importing il_modules needs lmdb/cv2/mmcv)
        pred = torch.log_softmax(pred / T, dim=1)
        soft = torch.softmax(soft / T, dim=1)
        return -1 * torch.mul(soft, pred).sum() / pred.shape[0]
    d = {}
    opt = make_opt(kind)
    net = DERNet(opt)
    for c in classes:
        net.update_fc(opt.hidden_size, c)
        net.build_prediction(opt, c)
        net.build_aux_prediction(opt, c)
    sd0 = net.state_dict()
    d["sd_keys"] = np.array(sorted(sd0.keys()))
    d["sd_shapes"] = np.array([",".join(map(str, sd0[k].shape)) for k in sorted(sd0.keys())])
    W.fill_state_dict(net.state_dict(), seed)
    image = torch.from_numpy(W.smooth_image("input:image", (B, 4, 32, 256), seed))
    words, chars = words_for(B, classes[-1] - 4, seed)
    conv = CTCLabelConverter(chars)
    labels_index, labels_length = conv.encode(words, batch_max_length=25)
    net.train()
    net.model[0].eval()
    with torch.no_grad():
        out = net(image)
        put(d, "logits", out["logits"])
        put(d, "aux_logits", out["aux_logits"])
        put(d, "features", out["features"])
        d["kd_loss"] = np.float64(_KD_loss(out["logits"].view(-1, classes[-1])[:, 0:classes[0]],
                                           out["aux_logits"].view(-1, classes[-1])[:, 0:classes[0]], 2).item())
        w = net.fc.weight.data.clone()
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            net.weight_align(classes[-1] - classes[0])
        d["weight_align_gamma"] = np.float64((net.fc.weight.data[-1] / w[-1]).mean().item())
        put(d, "fc_after_align", net.fc.weight.data)
    return d


def rcnn_model(B, seed):
    """a single recogniser with the RCNN (gated recurrent conv) extractor, reference modules/feature_extraction.py:50-162 via
    modules/model.py:44: train-mode forward + CTC loss + parameter gradients, eval-mode forward"""
    from modules.model import Model
    d = {}
    opt = make_opt("crnn")
    opt.FeatureExtraction = "RCNN"
    net = Model(opt)
    net.update_fc(opt.hidden_size, 40)
    net.build_prediction(opt, 40)
    sd0 = net.state_dict()
    d["sd_keys"] = np.array(sorted(sd0.keys()))
    d["sd_shapes"] = np.array([",".join(map(str, sd0[k].shape)) for k in sorted(sd0.keys())])
    W.fill_state_dict(net.state_dict(), seed)
    image = torch.from_numpy(W.smooth_image("input:image", (B, 4, 32, 256), seed))
    words, chars = words_for(B, 36, seed)
    conv = CTCLabelConverter(chars)
    labels_index, labels_length = conv.encode(words, batch_max_length=25)
    net.train()
    out = net(image, None, True)
    put(d, "train/feature", out["feature"])
    put(d, "train/predict", out["predict"])
    preds = out["predict"]
    loss = torch.nn.CTCLoss(zero_infinity=True)(preds.log_softmax(2).permute(1, 0, 2), labels_index, torch.IntTensor([preds.size(1)] * B),
                                                  labels_length)
    d["train/loss"] = np.float64(loss.item())
    net.zero_grad()
    loss.backward()
    named = dict(net.named_parameters())
    keys = ["model.FeatureExtraction.ConvNet.0.weight", "model.FeatureExtraction.ConvNet.3.wgf_u.weight",
            "model.FeatureExtraction.ConvNet.3.wr_x.weight", "model.FeatureExtraction.ConvNet.5.wf_u.weight",
            "model.FeatureExtraction.ConvNet.5.GRCL.2.BN_Gx.weight", "model.FeatureExtraction.ConvNet.7.GRCL.4.BN_grx.bias",
            "model.FeatureExtraction.ConvNet.7.wgr_x.weight", "model.FeatureExtraction.ConvNet.9.weight",
            "model.SequenceModeling.0.rnn.weight_hh_l0", "fc.weight"]
    d["grad_keys"] = np.array(keys)
    for k in keys:
        put(d, f"grad/{k}", named[k].grad)
    put(d, "bn_running_var_after", net.state_dict()["model.FeatureExtraction.ConvNet.5.GRCL.1.BN_rx.running_var"], full=True)
    W.fill_state_dict(net.state_dict(), seed)
    net.eval()
    with torch.no_grad():
        o = net(image, None, False)
        put(d, "eval/predict", o["predict"])
        d["eval/argmax"] = o["predict"].max(2)[1].numpy()
    return d


def converters():
    d = {}
    chars = "abcdefghij klmno"   # includes a space duplicate, as real dictionaries may
    words = ["hello", "", "a b", "zzz", "abcdefghijklmnoabcdefghij", "jjjj"]
    c = CTCLabelConverter(chars)
    idx, ln = c.encode(words, 25)
    d["ctc/encode_idx"], d["ctc/encode_len"] = idx.numpy(), ln.numpy()
    seq = np.array([[0, 5, 5, 0, 5, 6, 6, 1, 0, 2, 3, 3, 3, 0, 0, 7], [4, 4, 4, 4, 0, 0, 0, 0, 9, 9, 8, 8, 0, 1, 1, 2]])
    d["ctc/decode_in"] = seq
    d["ctc/decode_out"] = np.array(c.decode(seq, [16, 16]))
    a = AttnLabelConverter(chars)
    idx, ln = a.encode(words, 25)
    d["attn/encode_idx"], d["attn/encode_len"] = idx.numpy(), ln.numpy()
    d["attn/decode_out"] = np.array(a.decode(idx.numpy()[:, 1:], ln.numpy()))
    d["words"] = np.array(words)
    d["chars"] = np.array(chars)
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    jobs = {
        "crnn_mrn3": lambda: run("crnn", (40, 70, 97), 2, 1),
        "trba_mrn3": lambda: run("trba", (41, 71, 98), 2, 2),
        "svtr_mrn3": lambda: run("svtr", (40, 70, 97), 2, 3),
        "crnn_der2": lambda: der("crnn", (40, 70), 2, 4),
        "crnn_mrn3_noise": lambda: run("crnn", (40, 70, 97), 2, 1, noise=True),
        "trba_mrn3_noise": lambda: run("trba", (41, 71, 98), 2, 2, noise=True),
        "svtr_mrn3_noise": lambda: run("svtr", (40, 70, 97), 2, 3, noise=True),
        "rcnn_model": lambda: rcnn_model(2, 9),
        "converters": converters,
    }
    for name, fn in jobs.items():
        if args.only and args.only != name:
            continue
        d = fn()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, "->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")
