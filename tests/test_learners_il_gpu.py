"""LwF / EWC / DER learners and their kernels on the GPU (BASELINE config 5)."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_close
from tests.test_learner_gpu import make_opt

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def test_kd_ewc_weight_align_kernels():
    from mrn_amd import functional as Fn
    from mrn_amd import ops
    R, C, known = 50, 97, 70
    new = (rnd(R, C, seed=1) * 3).requires_grad_(True)
    old = rnd(R, C, seed=2) * 3
    # reference _KD_loss on the slice [1:known], T = 2 (il_modules/lwf.py:81-87,111-114)
    ref = -1 * torch.mul(torch.softmax(old[:, 1:known] / 2, 1), torch.log_softmax(new[:, 1:known] / 2, 1)).sum() / R
    (3 * ref).backward()
    nd = new.detach().cuda().requires_grad_(True)
    loss = Fn.kd_loss(nd, old.cuda(), 1, known, 2.0)
    assert_close("kd loss", loss.view(1), ref.detach().view(1), atol=1e-6, rtol=1e-5)
    (3 * loss).backward()
    assert_close("kd grad", nd.grad, new.grad, atol=1e-7, rtol=1e-4)
    # EWC: Fisher accumulate/finalise, penalty and its gradient
    n = 10007
    g1, g2, p, m = rnd(n, seed=3) * 0.02, rnd(n, seed=4) * 0.02, rnd(n, seed=5), rnd(n, seed=6)
    f = torch.zeros(n, device="cuda")
    ops.fisher_accumulate(f, g1.cuda())
    ops.fisher_accumulate(f, g2.cuda())
    ops.fisher_finalize(f, 2, 1e-4)
    fref = torch.min((g1 ** 2 + g2 ** 2) / 2, torch.tensor(1e-4))
    assert_close("fisher", f, fref, atol=1e-10, rtol=1e-5)
    pen = ops.ewc_penalty(f, p.cuda(), m.cuda())
    assert_close("ewc penalty", pen, (fref * (p - m) ** 2).sum().view(1) / 2, atol=1e-9, rtol=1e-4)
    grad = torch.zeros(n, device="cuda")
    ops.ewc_penalty_grad_(grad, f, p.cuda(), m.cuda(), 1000.0)
    assert_close("ewc grad", grad, 1000.0 * fref * (p - m), atol=1e-8, rtol=1e-5)
    # weight_align (modules/model.py:166-174)
    w = rnd(97, 256, seed=7)
    w[70:] *= 3.0
    gamma_ref = torch.norm(w[:70], dim=1).mean() / torch.norm(w[70:], dim=1).mean()
    wd = w.clone().cuda()
    gamma = ops.weight_align_(wd, 27)
    assert_close("gamma", gamma, gamma_ref.view(1), atol=1e-6, rtol=1e-5)
    wref = w.clone()
    wref[70:] *= gamma_ref
    assert_close("aligned", wd, wref, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("il", ["lwf", "ewc", "der", "wa"])
def test_il_learners_two_tasks(tmp_path, il):
    from mrn_amd.data.synthetic import SyntheticTextLines, SyntheticValidation, synthetic_characters
    from mrn_amd.il_modules.der import DER
    from mrn_amd.il_modules.ewc import EWC
    from mrn_amd.il_modules.lwf import LwF
    from mrn_amd.il_modules.wa import WA
    os.chdir(tmp_path)
    opt = make_opt(tmp_path, "crnn")
    opt.il, opt.memory = il, None
    torch.manual_seed(0)
    cls = {"lwf": LwF, "ewc": EWC, "der": DER, "wa": WA}[il]
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        learner = cls(opt)
        if il == "ewc":
            learner.fisher_iterations = 2
            learner.reference_prefix_bug = False      # exercise the real penalty path
        train, valid = SyntheticTextLines(opt), SyntheticValidation(opt)
        chars = ""
        for taski, n_new in enumerate((30, 20)):
            chars = synthetic_characters(len(chars) + n_new)
            train.set_characters(chars)
            valid.set_characters(chars)
            learner.incremental_train(taski, chars, train, valid)
            learner.after_task()
    net = learner.model
    assert all(torch.isfinite(p).all() for p in net.parameters())
    if il == "der":
        assert len(net.model) == 2 and net.fc.in_features == 512 and net.fc.out_features == 54
        assert "alignweights,gamma=" in sink.getvalue()
    if il == "wa":
        # aligned twice (end of the incremental task + after_task, wa.py:34-36,110): the second gamma must be ~1
        gammas = [float(l.split("=")[1]) for l in sink.getvalue().splitlines() if l.startswith("alignweights,gamma=")]
        assert len(gammas) == 2 and abs(gammas[1] - 1.0) < 1e-4
        w = net.fc.weight.detach()
        assert abs(float(w[:34].norm(dim=1).mean() / w[34:].norm(dim=1).mean()) - 1.0) < 1e-4
    if il == "ewc":
        assert learner.fisher is not None and all(float(v.max()) <= 1e-4 + 1e-12 for v in learner.fisher.values())


def test_joint_learner_runs_test_pass(tmp_path):
    """JointLearner (il_modules/joint.py): canonical loop + the per-dataset test pass at every validation interval"""
    from mrn_amd.data.synthetic import SyntheticTextLines, SyntheticValidation, synthetic_characters
    from mrn_amd.il_modules.joint import JointLearner
    os.chdir(tmp_path)
    opt = make_opt(tmp_path, "crnn")
    opt.il, opt.memory = "joint", None
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        learner = JointLearner(opt)
        chars = synthetic_characters(40)
        train, valid = SyntheticTextLines(opt), SyntheticValidation(opt)
        train.set_characters(chars)
        valid.set_characters(chars)
        scores, neds = learner.incremental_train(0, chars, train, valid, None, [valid.create_dataset(), valid.create_dataset()])
    assert len(scores) == 1 and len(neds) == 1 and 0.0 <= scores[0] <= 100.0
    assert all(torch.isfinite(p).all() for p in learner.model.parameters())
