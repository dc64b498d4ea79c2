"""Golden vectors for BASELINE config 5 (TRBA + DER, EWC / LwF auxiliary losses) and for validation(), produced by running
the REFERENCE learners (simplify23/MRN at /root/reference: il_modules/{base,der,ewc,lwf}.py, test.py) on CPU.

Run in the build container only (the reference never travels to the GPU box):
    python tests/golden/make_golden_il.py [--only NAME]
Writes tests/golden/{trba_der2,il_crnn,il_trba,validation}.npz.  As in make_golden.py, weights and inputs are NOT stored: they
come from mrn_amd/tools/weights.py (value = f(name, shape, seed)) and tests/helpers.py::DetLoader (batch = f(tag, seed, n)).

Harness shims, all confined to this script:
  * modules absent from the image and only touched by data / CLI code the synthetic step never calls are stubbed:
    lmdb, natsort, cv2, mmcv.Config, torchvision.transforms, timm.models.layers.trunc_normal_;
    nltk.metrics.distance.edit_distance is a plain Levenshtein distance (unit costs, no transpositions = nltk's default);
  * the reference Dataset_Manager / Val_Dataset (LMDB + the removed iterator.next()) is replaced by DetLoader;
  * the reference learners are driven through their own incremental_train() / after_task(); the only hooks are
    (a) deterministic weights after build_model / change_model, (b) recorders on Averager.add and _KD_loss,
    (c) ewc.num_iter (5000 Fisher iterations) lowered to 2.
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise RuntimeError("stubbed data-side component called")


timm = _stub("timm")
timm.models = _stub("timm.models")
timm.models.layers = _stub("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_)
_stub("lmdb")
_stub("natsort", natsorted=sorted)
_stub("cv2")
_stub("mmcv", Config=_Anything)
nltk = _stub("nltk")
nltk.metrics = _stub("nltk.metrics")
nltk.metrics.distance = _stub("nltk.metrics.distance", edit_distance=levenshtein)
tv = _stub("torchvision")
tv.transforms = _stub("torchvision.transforms", Compose=_Anything, ToTensor=_Anything, Resize=_Anything,
                      RandomApply=_Anything, RandomRotation=_Anything, RandomResizedCrop=_Anything, RandomGrayscale=_Anything,
                      ColorJitter=_Anything, RandomHorizontalFlip=_Anything)
torch.Tensor.cuda = lambda self, *a, **k: self

import il_modules.base as ref_base  # noqa: E402  (reference)
import il_modules.der as ref_der  # noqa: E402
import il_modules.ewc as ref_ewc  # noqa: E402
import il_modules.lwf as ref_lwf  # noqa: E402
import test as ref_test  # noqa: E402  (reference test.py: validation)
from modules.model import DERNet  # noqa: E402

from mrn_amd.tools import weights as W  # noqa: E402
from tests.golden.make_golden import make_opt, put, sub  # noqa: E402
from tests.helpers import DetLoader  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
torch.set_num_threads(8)

NCHARS = (36, 30)          # new characters per task -> Attn classes (41, 71), CTC classes (40, 70)


def learner_opt(kind, num_iter=2):
    o = make_opt(kind)
    o.__dict__.update(exp_name="g", il="x", memory=None, memory_num=2000, start_task=0, schedule="super", optimizer="adam",
                      lr=0.00003, batch_size=2, num_iter=num_iter, val_interval=1000, grad_clip=5, lan_list=["A", "B"], NED=True,
                      workers=0, manual_seed=111)
    return o


class Recorder:
    """collects what the reference loops feed their Averagers (the per-iteration losses)"""

    def __init__(self):
        self.values = []

    def averager(self):
        rec = self

        class Avg(ref_base.Averager):
            def add(self, v):
                rec.values.append(float(v.detach()))
                super().add(v)
        return Avg


def det_fill(learner, seed):
    W.fill_state_dict(learner.model.state_dict(), seed)


def hooked(cls, seeds):
    """reference learner class with deterministic weights after every build_model / change_model"""
    class Hooked(cls):
        def build_model(self):
            super().build_model()
            det_fill(self, seeds[0])

        def change_model(self):
            super().change_model()
            det_fill(self, seeds[1])
    return Hooked


def chars_upto(t):
    return "".join(chr(0x4E00 + i) for i in range(sum(NCHARS[:t + 1])))


def sub_params(d, prefix, named, keys, seed):
    """what the task's optimiser steps changed: trained value - deterministic initial value (1024-element subsample)"""
    named = dict(named)
    for k in keys:
        init = torch.from_numpy(W.det_param(W.canonical_key(k), tuple(named[k].shape), seed))
        for kk, v in sub(named[k].detach() - init, 1024).items():
            d[f"{prefix}/{k}/{kk}"] = v


def pick_keys(named, kind):
    """a few tensors that cover every stage (localisation net, first / deep conv, BN, LSTM, attention cell / head)"""
    names = [n for n, _ in named]
    want = ["conv0_1.weight", "layer3.2.conv1.weight", "layer4.0.bn2.weight", "localization_fc2.weight", "ConvNet.0.weight",
            "ConvNet.14.weight", "ConvNet.15.bias", "SequenceModeling.0.rnn.weight_hh_l0", "SequenceModeling.1.linear.weight",
            "attention_cell.rnn.weight_hh", "attention_cell.i2h.weight", "char_embeddings.weight", "fc.weight", "fc.bias"]
    out = []
    for w in want:
        for n in names:
            if n.endswith(w) and "aux_" not in n and n not in out:
                out.append(n)
                break
    return out


def run_learner(kind, which):
    """two tasks of a reference learner (LwF / EWC / DER) on deterministic batches; returns the fixture dict"""
    d = {}
    seeds = {"lwf": (21, 22), "ewc": (23, 24), "der": (25, 26)}[which]
    opt = learner_opt(kind)
    cls = {"lwf": ref_lwf.LwF, "ewc": ref_ewc.EWC, "der": ref_der.DER}[which]
    mod = {"lwf": ref_lwf, "ewc": ref_ewc, "der": ref_der}[which]
    rec = Recorder()
    saved_avg = mod.Averager
    mod.Averager = rec.averager()
    ref_base.Averager = mod.Averager           # BaseLearner._init_train (task 0 of LwF / EWC)
    kd_values = []
    if which == "lwf":
        orig_kd = ref_lwf._KD_loss

        def kd(pred, soft, T):
            v = orig_kd(pred, soft, T)
            kd_values.append(float(v.detach()))
            d.setdefault("kd_shape", np.array(pred.shape))
            return v
        ref_lwf._KD_loss = kd
    if which == "ewc":
        ref_ewc.num_iter = 2
    gammas = []
    if which == "der":
        orig_align = DERNet.weight_align

        def align(self, increment):
            before = self.fc.weight.data.clone()
            orig_align(self, increment)
            gammas.append(float((self.fc.weight.data[-1] / before[-1]).mean()))
            d["t1/fc_before_align"] = before.numpy().copy()[::7, ::5]
            d["t1/fc_after_align"] = self.fc.weight.data.numpy().copy()[::7, ::5]
        DERNet.weight_align = align
    train = DetLoader(2, f"il:{kind}:{which}", 31)
    valid = DetLoader(2, f"il:{kind}:{which}:val", 32)
    sink = io.StringIO()
    try:
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(io.StringIO()):
            learner = hooked(cls, seeds)(opt)
            for taski in range(2):
                chars = chars_upto(taski)
                train.set_characters(chars)
                valid.set_characters(chars)
                n0 = len(rec.values)
                if which == "ewc" and taski == 1:
                    ewc_vals = []
                    orig = learner.compute_ewc

                    def compute():
                        v = orig()
                        ewc_vals.append(float(v))
                        return v
                    learner.compute_ewc = compute
                learner.incremental_train(taski, chars, train, valid)
                d[f"t{taski}/losses"] = np.array(rec.values[n0:], dtype=np.float64)
                named = [(n.replace("module.", ""), p) for n, p in learner.model.state_dict().items()]
                keys = pick_keys(named, kind)
                d[f"t{taski}/param_keys"] = np.array(keys)
                sub_params(d, f"t{taski}/delta", named, keys, seeds[taski])
                if which == "ewc":
                    fk = list(learner.fisher.keys())
                    d[f"t{taski}/fisher_keys"] = np.array(fk)
                    for k in fk:
                        if k.replace("module.", "") in keys:
                            put(d, f"t{taski}/fisher/{k.replace('module.', '')}", learner.fisher[k])
                    d[f"t{taski}/fisher_total"] = np.float64(sum(float(v.double().sum()) for v in learner.fisher.values()))
                    d[f"t{taski}/fisher_saturated"] = np.float64(
                        sum(int((v >= 1e-4).sum()) for v in learner.fisher.values()) / sum(v.numel() for v in learner.fisher.values()))
                    if taski == 1:
                        d["t1/compute_ewc"] = np.array(ewc_vals, dtype=np.float64)
                learner.after_task()
                d[f"t{taski}/known_classes"] = np.int64(learner._known_classes)
                if which == "lwf" and taski == 0:
                    # the frozen previous network gets its own deterministic weights so that the KD target is a pure
                    # function of seeds (not of task 0's two Adam steps)
                    W.fill_state_dict(learner._old_network.state_dict(), 27)
                    d["old_network_training"] = np.bool_(learner._old_network.training)
    finally:
        mod.Averager = saved_avg
        ref_base.Averager = saved_avg
        if which == "lwf":
            ref_lwf._KD_loss = orig_kd
        if which == "der":
            DERNet.weight_align = orig_align
    if which == "lwf":
        d["t1/kd"] = np.array(kd_values, dtype=np.float64)
    if which == "der":
        d["t1/weight_align_gamma"] = np.array(gammas, dtype=np.float64)
        d["sd_keys"] = np.array(sorted(learner.model.state_dict().keys()))
        d["sd_shapes"] = np.array([",".join(map(str, learner.model.state_dict()[k].shape)) for k in sorted(learner.model.state_dict().keys())])
    d["n_valid_calls"] = np.int64(sink.getvalue().count("Current_score"))
    ck = sorted(os.listdir(f"./saved_models/{opt.exp_name}"))
    d["checkpoints"] = np.array(ck)
    return d


def il_fixture(kind):
    d = {}
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)
    os.makedirs("./saved_models/g", exist_ok=True)
    try:
        for which in ("lwf", "ewc", "der"):
            for k, v in run_learner(kind, which).items():
                d[f"{which}/{k}"] = v
            for f in os.listdir("./saved_models/g"):
                os.remove(os.path.join("./saved_models/g", f))
        # Fisher diagonal of a freshly filled model (no training before it): a pure function of seeds
        opt = learner_opt(kind)
        ref_ewc.num_iter = 2
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            learner = hooked(ref_ewc.EWC, (41, 42))(opt)
            chars = chars_upto(0)
            learner.character = chars
            learner.converter = learner.build_converter()
            learner.criterion = learner.build_criterion()
            learner.build_model()
            learner.build_optimizer(learner.count_param())
            loader = DetLoader(2, f"il:{kind}:fisher", 33)
            loader.set_characters(chars)
            fisher = learner.getFisherDiagonal(loader)
        named = [(n.replace("module.", ""), p) for n, p in learner.model.named_parameters()]
        keys = pick_keys(named, kind)
        d["fisher0/all_keys"] = np.array([n for n, _ in named])
        d["fisher0/keys"] = np.array(keys)
        for k in keys:
            put(d, f"fisher0/{k}", fisher["module." + k])
        d["fisher0/total"] = np.float64(sum(float(v.double().sum()) for v in fisher.values()))
    finally:
        os.chdir(cwd)
    return d


def trba_der2():
    """DERNet over two TRBA extractors (reference modules/model.py:203-312) in DER's training configuration (old extractor eval,
    new one train; der.py:39-44): teacher-forced logits / auxiliary logits / features, greedy eval-mode logits + argmax, one DER
    step's losses (der.py:249-265) and clipped gradient norm."""
    d = {}
    classes = (41, 71)
    opt = make_opt("trba")
    with contextlib.redirect_stdout(io.StringIO()):
        net = DERNet(opt)
        for c in classes:
            net.update_fc(opt.hidden_size, c)
            net.build_prediction(opt, c)
            net.build_aux_prediction(opt, c)
    sd0 = net.state_dict()
    d["sd_keys"] = np.array(sorted(sd0.keys()))
    d["sd_shapes"] = np.array([",".join(map(str, sd0[k].shape)) for k in sorted(sd0.keys())])
    seed, B = 8, 2
    W.fill_state_dict(net.state_dict(), seed)
    loader = DetLoader(B, "trba_der2", seed)
    loader.set_characters(chars_upto(1))
    image, words = loader.get_batch()
    from tools.utils import AttnLabelConverter
    with contextlib.redirect_stdout(io.StringIO()):
        conv = AttnLabelConverter(chars_upto(1))
    labels_index, labels_length = conv.encode(words, batch_max_length=25)
    d["labels_index"] = labels_index.numpy()
    for ext in list(net.model)[:-1]:
        for p in ext.parameters():
            p.requires_grad = False
    net.train()
    net.model[0].eval()
    out = net(image, labels_index[:, :-1])
    put(d, "logits", out["logits"])
    put(d, "aux_logits", out["aux_logits"])
    put(d, "features", out["features"])
    crit = torch.nn.CrossEntropyLoss(ignore_index=conv.dict["[PAD]"])
    target = labels_index[:, 1:]
    loss_clf = crit(out["logits"].view(-1, out["logits"].shape[-1]), target.contiguous().view(-1))
    loss_aux = crit(out["aux_logits"].view(-1, out["aux_logits"].shape[-1]), target.contiguous().view(-1))
    d["loss_clf"], d["loss_aux"] = np.float64(loss_clf.item()), np.float64(loss_aux.item())
    net.zero_grad()
    loss_clf.backward()
    d["grad_norm"] = np.float64(torch.nn.utils.clip_grad_norm_(net.parameters(), 5).item())
    for k in ("fc.weight", "Prediction.attention_cell.i2h.weight", "Prediction.attention_cell.rnn.weight_ih",
              "model.1.SequenceModeling.1.linear.weight"):
        put(d, f"grad/{k}", net.get_parameter(k).grad)
    assert all(p.grad is None for p in net.aux_Prediction.attention_cell.parameters())       # aux loss is not in the loss
    W.fill_state_dict(net.state_dict(), seed)
    net.eval()
    with torch.no_grad():
        sos = torch.LongTensor(B).fill_(conv.dict["[SOS]"])
        oe = net(image, sos, False)
        put(d, "eval/logits", oe["logits"])
        d["eval/argmax"] = oe["logits"].max(2)[1].numpy()
    return d


def validation_fixture():
    """reference test.py:139-279 (validation) on hand-made logits (tests/helpers.py::crafted_validation_case: every scoring
    branch) and on real models (a CRNN and a TRBA recogniser with deterministic weights): loss / accuracy / normalised edit
    distance / decoded strings / confidences."""
    from modules.model import Model
    from tools.utils import AttnLabelConverter, CTCLabelConverter
    from tests.helpers import crafted_validation_case
    d = {}
    for kind in ("crnn", "trba"):
        opt = make_opt(kind)
        opt.NED = True
        chars, batches, logits = crafted_validation_case(kind)
        with contextlib.redirect_stdout(io.StringIO()):
            conv = CTCLabelConverter(chars) if kind == "crnn" else AttnLabelConverter(chars)
        crit = torch.nn.CTCLoss(zero_infinity=True) if kind == "crnn" else torch.nn.CrossEntropyLoss(ignore_index=conv.dict["[PAD]"])
        calls = iter(logits)
        stub = lambda image, *a, **k: {"predict": next(calls), "feature": None}        # noqa: E731
        preds_all, conf_all = [], []
        with torch.no_grad(), contextlib.redirect_stderr(io.StringIO()):
            loss, acc, ned, preds, conf, labels, _, n = ref_test.validation(stub, crit, batches, conv, opt)
            # per-batch results too (the reference returns the strings / confidences of the LAST batch only)
            for i in range(len(batches)):
                one = iter([logits[i]])
                r = ref_test.validation(lambda image, *a, **k: {"predict": next(one), "feature": None}, crit, [batches[i]], conv, opt)
                d[f"crafted/{kind}/batch{i}/accuracy"], d[f"crafted/{kind}/batch{i}/ned"] = np.float64(r[1]), np.float64(r[2])
                d[f"crafted/{kind}/batch{i}/loss"] = np.float64(float(r[0]))
                d[f"crafted/{kind}/batch{i}/preds"] = np.array(list(r[3]))
                d[f"crafted/{kind}/batch{i}/confidence"] = np.array([float(c) for c in r[4]], dtype=np.float64)
        pre = f"crafted/{kind}/"
        d[pre + "valid_loss"], d[pre + "accuracy"], d[pre + "ned"] = np.float64(float(loss)), np.float64(acc), np.float64(ned)
        d[pre + "preds_last_batch"] = np.array(list(preds))
        d[pre + "confidence_last_batch"] = np.array([float(c) for c in conf], dtype=np.float64)
        d[pre + "labels_last_batch"] = np.array(list(labels))
        d[pre + "length"] = np.int64(n)
        print(kind, "crafted: acc", acc, "ned", ned, [str(p)[:30] for p in preds], [float(c) for c in conf])
        # ---- a real model: forward path + decoding + scoring together ----
        classes = 40 if kind == "crnn" else 41
        seed = 51 if kind == "crnn" else 52
        with contextlib.redirect_stdout(io.StringIO()):
            net = Model(opt)
            net.update_fc(opt.hidden_size, classes)
            net.build_prediction(opt, classes)
        W.fill_state_dict(net.state_dict(), seed)
        with torch.no_grad():
            net.fc.weight *= 60.0             # make the argmax depend on the input instead of on the head's bias
        net.eval()
        loader = DetLoader(3, f"validation:{kind}", seed, oov=True, n_valid=2)
        loader.set_characters(chars)
        real = loader.create_dataset()
        with torch.no_grad(), contextlib.redirect_stderr(io.StringIO()):
            loss, acc, ned, preds, conf, labels, _, n = ref_test.validation(net, crit, real, conv, opt)
        pre = f"model/{kind}/"
        d[pre + "valid_loss"], d[pre + "accuracy"], d[pre + "ned"] = np.float64(float(loss)), np.float64(acc), np.float64(ned)
        d[pre + "preds_last_batch"] = np.array(list(preds))
        d[pre + "confidence_last_batch"] = np.array([float(c) for c in conf], dtype=np.float64)
        d[pre + "length"] = np.int64(n)
        print(kind, "model: acc", acc, "ned", ned, "loss", float(loss), [float(c) for c in conf])
    return d


def rehearsal_fixture():
    """index bookkeeping of the rehearsal memory (il_modules/base.py:278-302, il_modules/mrn.py:168-181): the memory_index lists the
    REFERENCE learners build over four tasks from a seeded numpy RNG, and what they hand to train_loader.get_dataset()"""
    import il_modules.mrn as ref_mrn
    d = {}
    for name, cls, memory_num in (("base", ref_base.BaseLearner, 2000), ("mrn", ref_mrn.MRN, 2000), ("mrn_large", ref_mrn.MRN, 6000)):
        opt = learner_opt("crnn")
        opt.memory, opt.memory_num, opt.il = "random", memory_num, "mrn"
        with contextlib.redirect_stdout(io.StringIO()):
            learner = cls(opt)
            loader = DetLoader(2, "rehearsal", 1)
            loader.dataset_len = 9000
            loader.rehearsal_prev_model = lambda taski: (loader, 9000 - 1000 * taski)
            np.random.seed(1234)
            for taski in range(1, 4):
                learner.build_rehearsal_memory(loader, taski)
                d[f"{name}/t{taski}/lengths"] = np.array([len(i) for i in learner.memory_index], dtype=np.int64)
                d[f"{name}/t{taski}/checksums"] = np.array([int(np.asarray(i, dtype=np.int64).sum()) for i in learner.memory_index], dtype=np.int64)
            for i, idx in enumerate(learner.memory_index):
                d[f"{name}/final/memory_index/{i}"] = np.asarray(idx).astype(np.int32)
        calls = [c for c in loader.calls if c[0] == "get_dataset"]
        d[f"{name}/get_dataset_memory_args"] = np.array([str(c[1][1]) for c in calls])
        d[f"{name}/get_dataset_index_checksums"] = np.array([sum(int(i.sum()) for i in c[1][2]) for c in calls], dtype=np.int64)
    return d


def data_manage_fixture():
    """reference data/data_manage.py (Dataset_Manager, Val_Dataset, IndexConcatDataset) + data/dataset.py (AlignCollate[2],
    ResizeNormalize) over in-memory datasets: what get_dataset() builds for every `memory` mode and the first batches it
    yields under fixed numpy / torch seeds.  Shims: LmdbDataset -> in-memory samples (tests/helpers.py::fake_text_samples),
    torchvision's ToTensor -> the same uint8 HWC -> float CHW / 255 conversion, the removed DataLoader-iterator .next()."""
    import data.data_manage as ref_dm
    import data.dataset as ref_ds
    import PIL.Image
    from torch.utils.data import Dataset
    from torch.utils.data.dataloader import _BaseDataLoaderIter
    from tests.helpers import fake_text_samples
    _BaseDataLoaderIter.next = lambda self: self.__next__()

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img, dtype=np.uint8)
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)
    ref_ds.transforms.ToTensor = ToTensor

    class FakeLmdb(Dataset):
        def __init__(self, root, opt, mode="train"):
            images, labels = fake_text_samples(root)
            keep = [i for i, l in enumerate(labels) if len(l) <= opt.batch_max_length]      # LmdbDataset's length filter (:78-83)
            self.items = [(images[i], labels[i]) for i in keep]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return PIL.Image.fromarray(self.items[i][0]).convert("RGBA"), self.items[i][1]
    ref_dm.LmdbDataset = FakeLmdb

    def fake_tree(root, opt, select_data="/", data_type="label", mode="train"):
        from torch.utils.data import ConcatDataset
        return ConcatDataset([FakeLmdb(root, opt, mode)]), "log"
    ref_dm.hierarchical_dataset = fake_tree
    d = {}
    scenarios = [("mrn_random", "mrn", "random"), ("plain", "mrn", None), ("test_ch", "lwf", "test_ch"), ("large", "lwf", "large"),
                 ("total", "lwf", "total"), ("halves", "lwf", "random")]
    for name, il, memory in scenarios:
        opt = learner_opt("crnn")
        opt.__dict__.update(il=il, memory=memory, memory_num=40 if memory != "large" else 12, batch_size=6, workers=0, Aug="None",
                            lan_list=["Chinese", "Latin", "Japanese"], select_data=["rootA", "rootB"])
        np.random.seed(77)
        torch.manual_seed(77)
        with contextlib.redirect_stdout(io.StringIO()):
            dm = ref_dm.Dataset_Manager(opt)
            dm.select_data = opt.select_data
            taski = 2
            index_list = [np.random.choice(range(30), 40 // taski if memory != "large" else 12, replace=False) for _ in range(taski)]
            out_index = dm.get_dataset(taski, memory=memory, index_list=index_list)
            d[f"{name}/n_loaders"] = np.int64(len(dm.data_loader_list))
            d[f"{name}/loader_dataset_lengths"] = np.array([len(l.dataset) for l in dm.data_loader_list], dtype=np.int64)
            d[f"{name}/loader_batch_sizes"] = np.array([l.batch_size for l in dm.data_loader_list], dtype=np.int64)
            mix = isinstance(dm.data_loader_list[0].collate_fn, ref_ds.AlignCollate2)
            d[f"{name}/mix"] = np.bool_(mix)
            for b in range(3):
                got = dm.get_batch2() if mix else dm.get_batch()
                d[f"{name}/batch{b}/labels"] = np.array(list(got[1]))
                d[f"{name}/batch{b}/image_shape"] = np.array(got[0].shape)
                d[f"{name}/batch{b}/image_sum"] = np.float64(got[0].double().sum().item())
                d[f"{name}/batch{b}/image_probe"] = got[0][:, :, ::8, ::32].numpy().copy()
                if mix:
                    d[f"{name}/batch{b}/index"] = np.array([list(t) for t in got[2]], dtype=np.int64)
            loader, n = dm.rehearsal_prev_model(taski)
            d[f"{name}/prev_len"] = np.int64(n)
    # validation side
    opt = learner_opt("crnn")
    opt.__dict__.update(batch_size=5, workers=0, Aug="None", lan_list=["Chinese", "Latin"], NED=True)
    np.random.seed(78)
    torch.manual_seed(78)
    with contextlib.redirect_stdout(io.StringIO()):
        vd = ref_dm.Val_Dataset(["valA/Chinese", "valA/Latin"], opt)
        for name, loader in (("val/current", vd.create_dataset()), ("val/list", vd.create_list_dataset())):
            d[f"{name}/len"] = np.int64(len(loader.dataset))
            images, labels = next(iter(loader))
            d[f"{name}/labels"] = np.array(list(labels))
            d[f"{name}/image_sum"] = np.float64(images.double().sum().item())
    # ResizeNormalize alone (BICUBIC resize to 256 x 32, [-1, 1])
    img, _ = FakeLmdb("rootA/Latin", opt)[3]
    t = ref_ds.ResizeNormalize((256, 32))(img)
    d["resize/shape"] = np.array(t.shape)
    d["resize/probe"] = t[:, ::4, ::16].numpy().copy()
    d["resize/sum"] = np.float64(t.double().sum().item())
    return d


EOS_BIAS = 0.0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--eos-bias", type=float, default=None)
    args = ap.parse_args()
    if args.eos_bias is not None:
        EOS_BIAS = args.eos_bias
    jobs = {
        "trba_der2": trba_der2,
        "il_crnn": lambda: il_fixture("crnn"),
        "il_trba": lambda: il_fixture("trba"),
        "validation": validation_fixture,
        "rehearsal": rehearsal_fixture,
        "data_manage": data_manage_fixture,
    }
    for name, fn in jobs.items():
        if args.only and args.only != name:
            continue
        d = fn()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, "->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")
