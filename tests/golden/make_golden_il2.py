"""Golden vectors for the remaining incremental learners of the reference -- WA (il_modules/wa.py:29-116) and JointLearner
(il_modules/joint.py:9-105) -- produced by driving the REFERENCE classes through incremental_train() / after_task() on CPU,
exactly as make_golden_il.py does for LwF / EWC / DER (same stubs, same deterministic weights and DetLoader batches: this
script imports that harness).

Run in the build container only (the reference never travels to the GPU box):
    python tests/golden/make_golden_il2.py [--only il2_crnn|il2_trba]
Writes tests/golden/il2_{crnn,trba}.npz:
  wa/    two tasks of WA: per-iteration losses (task 0: BaseLearner._init_train; task 1: loss_clf + 2 * KD), the KD terms,
         the two weight_align() calls of task 1 (end of _update_representation and after_task(): gamma, classifier rows before /
         after), parameter movement, known-class bookkeeping, checkpoints written, validations run;
  joint/ two rounds of JointLearner (second one after change_model()): per-iteration losses, parameter movement, checkpoints,
         validations, the (empty) score lists incremental_train() returns when no test interval is reached.
Hooks: deterministic weights after build_model / change_model, recorders on Averager.add / wa._KD_loss / Model.weight_align,
the frozen previous network of WA refilled deterministically after task 0 (as for LwF).
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests.golden import make_golden_il as H  # noqa: E402  (installs the stubs, imports the reference)

import il_modules.joint as ref_joint  # noqa: E402  (reference)
import il_modules.wa as ref_wa  # noqa: E402
from modules.model import Model  # noqa: E402

from mrn_amd.tools import weights as W  # noqa: E402
from tests.helpers import DetLoader  # noqa: E402

SEEDS = {"wa": (51, 52), "joint": (53, 54)}


def run_wa(kind):
    d = {}
    seeds = SEEDS["wa"]
    opt = H.learner_opt(kind)
    rec = H.Recorder()
    saved_avg = ref_wa.Averager
    ref_wa.Averager = rec.averager()
    H.ref_base.Averager = ref_wa.Averager
    kd_values, aligns = [], []
    orig_kd, orig_align = ref_wa._KD_loss, Model.weight_align

    def kd(pred, soft, T):
        v = orig_kd(pred, soft, T)
        kd_values.append(float(v.detach()))
        d.setdefault("kd_shape", np.array(pred.shape))
        return v

    def align(self, increment):
        before = self.fc.weight.data.clone()
        orig_align(self, increment)
        i = len(aligns)
        aligns.append(float((self.fc.weight.data[-1] / before[-1]).mean()))
        d[f"t1/align{i}/increment"] = np.int64(increment)
        d[f"t1/align{i}/fc_before"] = before.numpy().copy()[::7, ::5]
        d[f"t1/align{i}/fc_after"] = self.fc.weight.data.numpy().copy()[::7, ::5]
    ref_wa._KD_loss = kd
    Model.weight_align = align
    train = DetLoader(2, f"il2:{kind}:wa", 61)
    valid = DetLoader(2, f"il2:{kind}:wa:val", 62)
    sink = io.StringIO()
    try:
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(io.StringIO()):
            learner = H.hooked(ref_wa.WA, seeds)(opt)
            for taski in range(2):
                chars = H.chars_upto(taski)
                train.set_characters(chars)
                valid.set_characters(chars)
                n0 = len(rec.values)
                learner.incremental_train(taski, chars, train, valid)
                d[f"t{taski}/losses"] = np.array(rec.values[n0:], dtype=np.float64)
                named = [(n.replace("module.", ""), p) for n, p in learner.model.state_dict().items()]
                keys = H.pick_keys(named, kind)
                d[f"t{taski}/param_keys"] = np.array(keys)
                H.sub_params(d, f"t{taski}/delta", named, keys, seeds[taski])
                learner.after_task()
                d[f"t{taski}/known_classes"] = np.int64(learner._known_classes)
                if taski == 0:
                    W.fill_state_dict(learner._old_network.state_dict(), 57)
                    d["old_network_training"] = np.bool_(learner._old_network.training)
    finally:
        ref_wa.Averager = saved_avg
        H.ref_base.Averager = saved_avg
        ref_wa._KD_loss = orig_kd
        Model.weight_align = orig_align
    d["t1/kd"] = np.array(kd_values, dtype=np.float64)
    d["t1/weight_align_gamma"] = np.array(aligns, dtype=np.float64)
    d["n_valid_calls"] = np.int64(sink.getvalue().count("Current_score"))
    d["checkpoints"] = np.array(sorted(os.listdir(f"./saved_models/{opt.exp_name}")))
    return d


def run_joint(kind):
    d = {}
    seeds = SEEDS["joint"]
    opt = H.learner_opt(kind)
    opt.saved_model = ""
    rec = H.Recorder()
    saved_avg = ref_joint.Averager
    ref_joint.Averager = rec.averager()
    train = DetLoader(2, f"il2:{kind}:joint", 63)
    valid = DetLoader(2, f"il2:{kind}:joint:val", 64)
    sink = io.StringIO()
    try:
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(io.StringIO()):
            learner = H.hooked(ref_joint.JointLearner, seeds)(opt)
            for taski in range(2):
                chars = H.chars_upto(taski)
                train.set_characters(chars)
                valid.set_characters(chars)
                n0 = len(rec.values)
                best, ned = learner.incremental_train(taski, chars, train, valid, None, None)
                d[f"t{taski}/returned_lengths"] = np.array([len(best), len(ned)])
                d[f"t{taski}/losses"] = np.array(rec.values[n0:], dtype=np.float64)
                named = [(n.replace("module.", ""), p) for n, p in learner.model.state_dict().items()]
                keys = H.pick_keys(named, kind)
                d[f"t{taski}/param_keys"] = np.array(keys)
                H.sub_params(d, f"t{taski}/delta", named, keys, seeds[taski])
                learner.after_task()
                d[f"t{taski}/known_classes"] = np.int64(learner._known_classes)
    finally:
        ref_joint.Averager = saved_avg
    d["n_valid_calls"] = np.int64(sink.getvalue().count("Current_score"))
    d["checkpoints"] = np.array(sorted(os.listdir(f"./saved_models/{opt.exp_name}")))
    return d


def fixture(kind):
    d = {}
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)
    os.makedirs("./saved_models/g", exist_ok=True)
    try:
        for which, fn in (("wa", run_wa), ("joint", run_joint)):
            for k, v in fn(kind).items():
                d[f"{which}/{k}"] = v
            for f in os.listdir("./saved_models/g"):
                os.remove(os.path.join("./saved_models/g", f))
    finally:
        os.chdir(cwd)
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    for kind in ("crnn", "trba"):
        name = "il2_" + kind
        if args.only and args.only != name:
            continue
        d = fixture(kind)
        path = os.path.join(H.OUT, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, "->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")
