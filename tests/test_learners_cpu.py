"""Host logic of the learners that needs no GPU: rehearsal-memory index bookkeeping (against lists the REFERENCE learners
built, tests/golden/rehearsal.npz), learning-rate schedules, optimiser construction, checkpoint naming / resume control
flow, the TPS buffer keys of reference checkpoints."""
import contextlib
import io
import os
import types

import numpy as np
import pytest
import torch

from tests.helpers import DetLoader, load_golden


def learner_opt(**kw):
    o = types.SimpleNamespace(num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512, hidden_size=256,
                              batch_max_length=25, exp_name="g", il="mrn", memory="random", memory_num=2000, start_task=0,
                              schedule="super", optimizer="adam", lr=0.0005, batch_size=2, num_iter=10, val_interval=1000,
                              grad_clip=5, lan_list=["A", "B", "C", "D"], NED=True, workers=0, manual_seed=111,
                              Transformation="None", FeatureExtraction="VGG", SequenceModeling="BiLSTM", Prediction="CTC",
                              sgd_momentum=0.9, sgd_weight_decay=1e-4, rho=0.95, eps=1e-8, lr_drop_rate=0.1)
    o.__dict__.update(kw)
    return o


@pytest.mark.parametrize("name", ["base", "mrn", "mrn_large"])
def test_rehearsal_memory_bookkeeping_matches_reference(name):
    """base.py:278-302 / mrn.py:168-181: per previous task `memory_num / taski` random indices (6000-sample memories: per task),
    older lists cut back when the memory overflows, the lists handed to get_dataset() -- bit-exact integers under one numpy seed"""
    from mrn_amd.il_modules.base import BaseLearner
    from mrn_amd.il_modules.mrn import MRN
    g = load_golden("rehearsal")
    cls = BaseLearner if name == "base" else MRN
    opt = learner_opt(memory_num=6000 if name == "mrn_large" else 2000)
    with contextlib.redirect_stdout(io.StringIO()):
        learner = cls(opt)
        loader = DetLoader(2, "rehearsal", 1)
        loader.rehearsal_prev_model = lambda taski: (loader, 9000 - 1000 * taski)
        np.random.seed(1234)
        for taski in range(1, 4):
            learner.build_rehearsal_memory(loader, taski)
            assert [len(i) for i in learner.memory_index] == list(g[f"{name}/t{taski}/lengths"])
            assert [int(np.asarray(i, dtype=np.int64).sum()) for i in learner.memory_index] == list(g[f"{name}/t{taski}/checksums"])
    for i, idx in enumerate(learner.memory_index):
        assert np.array_equal(np.asarray(idx), g[f"{name}/final/memory_index/{i}"])
    calls = [c for c in loader.calls if c[0] == "get_dataset"]
    assert [str(c[1][1]) for c in calls] == [str(s) for s in g[f"{name}/get_dataset_memory_args"]]
    assert [sum(int(i.sum()) for i in c[1][2]) for c in calls] == list(g[f"{name}/get_dataset_index_checksums"])


class FakeOptimizer:
    def __init__(self, lr):
        self.param_groups = [{"lr": lr}]
        self.lrs, self.momenta = [], []

    def zero_grad(self):
        pass

    def step(self, lr=None, max_norm=None, momentum=None):
        self.lrs.append(lr)
        self.momenta.append(momentum)
        self.param_groups[0]["lr"] = lr


class FakeLoss:
    def backward(self):
        pass


def test_stepwise_schedule_reaches_the_optimiser():
    """schedule = [0.6, 0.8] (tools/utils.py:169-178): the lr the optimiser STEPS with drops by lr_drop_rate at 60 % and again at
    80 % of num_iter (round 1 only changed the logged value)"""
    from mrn_amd.il_modules.base import BaseLearner
    opt = learner_opt(schedule=[0.6, 0.8], num_iter=10)
    with contextlib.redirect_stdout(io.StringIO()):
        learner = BaseLearner(opt)
    learner.optimizer, learner.scheduler, learner.reducer, learner.opt_step = FakeOptimizer(opt.lr), None, None, 0
    for it in range(1, 11):
        learner.backward_and_step(FakeLoss())
        learner.end_iteration(it)
    want = [5e-4] * 6 + [5e-5] * 2 + [5e-6] * 2        # iteration it steps with what end_iteration(it - 1) set
    assert np.allclose(learner.optimizer.lrs, want, rtol=1e-12), learner.optimizer.lrs


def test_one_cycle_momentum_matches_torch_sgd_cycle():
    from mrn_amd.optim import OneCycle
    p = torch.nn.Parameter(torch.zeros(1))
    sgd = torch.optim.SGD([p], lr=0.1, momentum=0.9)
    ref = torch.optim.lr_scheduler.OneCycleLR(sgd, max_lr=0.1, cycle_momentum=True, div_factor=20, final_div_factor=1000, total_steps=50)
    mine = OneCycle(0.1, 50, cycle_momentum=True)
    for step in range(50):
        assert abs(sgd.param_groups[0]["lr"] - mine.lr_at(step)) < 1e-15
        assert abs(sgd.param_groups[0]["momentum"] - mine.momentum_at(step)) < 1e-15
        sgd.step()
        if step < 49:
            ref.step()
    assert OneCycle(0.1, 50).momentum_at(3) is None


def test_checkpoint_names_follow_the_reference():
    from mrn_amd.il_modules.base import BaseLearner
    from mrn_amd.il_modules.der import DER
    from mrn_amd.il_modules.ewc import EWC
    from mrn_amd.il_modules.lwf import LwF
    from mrn_amd.il_modules.mrn import MRN
    opt = learner_opt()
    with contextlib.redirect_stdout(io.StringIO()):
        for cls in (BaseLearner, LwF, EWC, DER):       # base.py:331, der.py:318: {lan}_{taski}_best_score.pth
            assert cls(opt).checkpoint_path(1, None).endswith("/B_1_best_score.pth")
            assert cls(opt).checkpoint_path(1, 0).endswith("/B_1_best_score.pth")
        m = MRN(opt)                                    # mrn.py:414: {lan}_{taski}_{step}_best_score.pth
    assert m.checkpoint_path(1, 0).endswith("/B_1_0_best_score.pth") and m.checkpoint_path(1, 1).endswith("/B_1_1_best_score.pth")


def test_tps_buffer_keys_of_reference_checkpoints_load_strictly():
    """reference checkpoints written on a multi-GPU host carry GridGenerator.inv_delta_C / P_hat per TPS stage, single-GPU ones do
    not (modules/transformation.py:127-146): both load with strict=True whatever this host is"""
    from mrn_amd.modules.model import Model
    from mrn_amd.modules import transformation as T
    opt = learner_opt(Transformation="TPS", FeatureExtraction="ResNet", Prediction="Attn")
    for persist in (False, True):
        old = T._reference_persists_tps_buffers
        T._reference_persists_tps_buffers = lambda: persist
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                net = Model(opt)
                net.update_fc(256, 30)
                net.build_prediction(opt, 30)
        finally:
            T._reference_persists_tps_buffers = old
        sd = net.state_dict()
        has = any(k.endswith("GridGenerator.P_hat") for k in sd)
        assert has == persist
        base = {k: v for k, v in sd.items() if "GridGenerator" not in k}
        gg = net.model.Transformation.GridGenerator
        with_keys = dict(base, **{"model.Transformation.GridGenerator.inv_delta_C": gg.inv_delta_C.clone(),
                                  "model.Transformation.GridGenerator.P_hat": gg.P_hat.clone()})
        for flavour in (base, with_keys):
            missing, unexpected = net.load_state_dict(dict(flavour), strict=True)
            assert not missing and not unexpected


def test_optimizer_selection_and_unknown_name():
    from mrn_amd.il_modules.base import BaseLearner
    with contextlib.redirect_stdout(io.StringIO()):
        learner = BaseLearner(learner_opt(optimizer="rmsprop"))
    with pytest.raises(ValueError):
        learner.build_optimizer([torch.nn.Parameter(torch.zeros(4))])
