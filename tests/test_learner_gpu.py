"""End-to-end plugin surface on the GPU: the MRN learner driven exactly like tiny_train.train() does
(reference tiny_train.py:232-277): incremental_train -> test -> after_task over two synthetic tasks."""
import contextlib
import io
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def make_opt(tmp, model):
    o = types.SimpleNamespace(
        exp_name="t", il="mrn", memory="random", memory_num=2000, batch_max_length=25, imgH=32, imgW=256, manual_seed=111,
        start_task=0, num_fiducial=20, input_channel=4, output_channel=512, hidden_size=256, schedule="super",
        optimizer="adam", lr=0.0005, batch_size=4, num_iter=4, val_interval=2, grad_clip=5, lan_list=["A", "B"], NED=True,
        workers=0)
    if model == "crnn":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "VGG", "BiLSTM", "CTC"
    elif model == "svtr":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "SVTR", "None", "CTC"
    else:
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "TPS", "ResNet", "BiLSTM", "Attn"
    return o


@pytest.mark.parametrize("model", ["crnn", "trba", "svtr"])
def test_mrn_learner_two_tasks(tmp_path, model):
    from mrn_amd.data.synthetic import SyntheticTextLines, SyntheticValidation, synthetic_characters
    from mrn_amd.il_modules.mrn import MRN
    os.chdir(tmp_path)
    opt = make_opt(tmp_path, model)
    torch.manual_seed(0)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        learner = MRN(opt)
        train, valid = SyntheticTextLines(opt), SyntheticValidation(opt)
        best, ned = [], []
        chars = ""
        for taski, n_new in enumerate((30, 20)):
            chars = synthetic_characters(len(chars) + n_new)
            train.set_characters(chars)
            valid.set_characters(chars)
            if taski == 0:
                train.init_start(taski)
            learner.incremental_train(taski, chars, train, valid)
            if taski == 0:
                first = [p.detach().clone() for p in learner.model.module.model[0].parameters()]
            best, ned = learner.test(None, [valid.create_dataset()], best, ned, taski)
            learner.after_task()
    net = learner.model
    assert len(net.model) == 2 and len(best) == 2
    # task 1: expert 0 stayed frozen through loop A of expert 1 and loop B of the router
    for a, b in zip(first, net.model[0].parameters()):
        assert torch.equal(a, b.detach())
    assert all(torch.isfinite(p).all() for p in net.parameters())
    assert os.path.exists(f"./saved_models/{opt.exp_name}/B_1_1_best_score.pth")
    sd = torch.load(f"./saved_models/{opt.exp_name}/B_1_1_best_score.pth")
    assert all(k.startswith("module.") for k in sd)          # reference checkpoint layout (DataParallel prefix)
    log = sink.getvalue()
    assert "Train_taski_loss" in log and "Current_score" in log


def test_prefetched_expert_forward_is_bit_identical(tmp_path):
    """Loop B with the software pipeline (batch n+1's frozen-expert forward issued on the side streams before batch n's
    router phase, MRN.prefetch_experts / MRNNet.experts_prefetch) against the strictly sequential loop: same losses and
    bit-identical router parameters after several steps."""
    import bench
    from mrn_amd.data.synthetic import SyntheticTextLines
    from mrn_amd.tools.utils import to_device
    results = []
    for pipeline in (True, False):
        torch.manual_seed(111)
        opt = bench.make_opt("crnn", 8)
        learner = bench.build_learner(opt, 4)
        data = SyntheticTextLines(opt, seed=5)
        data.set_characters(learner.character)

        def fetch():
            image, labels, idx = data.get_batch2()
            indexs = to_device(torch.LongTensor(idx).squeeze())
            pre = learner.prefetch_experts(image, labels) if pipeline else None
            assert not pipeline or pre[0] is not None
            return image, labels, indexs, pre
        nxt = fetch()
        losses = []
        for it in range(4):
            image, labels, indexs, pre = nxt
            if it < 3:                  # (as in MRN._update_representation: no look-ahead past the last batch)
                nxt = fetch()
            lc, lt = learner.routing_step(image, labels, indexs, prefetched=pre)
            losses.append((float(lc.detach()), float(lt.detach())))
        torch.cuda.synchronize()
        flat = learner.optimizer.flat.detach().clone()
        bn = torch.cat([v.flatten().float() for k, v in learner.model.state_dict().items() if "running_var" in k])
        results.append((losses, flat, bn))
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1]) and torch.equal(results[0][2], results[1][2])


def test_mrn_resume_from_checkpoints(tmp_path):
    """opt.start_task (reference base.py:178-195, mrn.py:187-203): tasks below it skip training and load the checkpoints the
    earlier run wrote ({lan}_{taski}_{step}_best_score.pth); the resumed learner ends with the trained learner's parameters"""
    from mrn_amd.data.synthetic import SyntheticTextLines, SyntheticValidation, synthetic_characters
    from mrn_amd.il_modules.mrn import MRN
    os.chdir(tmp_path)
    states = []
    for start_task in (0, 2):
        opt = make_opt(tmp_path, "crnn")
        opt.start_task = start_task
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            learner = MRN(opt)
            train, valid = SyntheticTextLines(opt), SyntheticValidation(opt)
            chars = ""
            for taski, n_new in enumerate((30, 20)):
                chars = synthetic_characters(len(chars) + n_new)
                train.set_characters(chars)
                valid.set_characters(chars)
                learner.incremental_train(taski, chars, train, valid)
                if start_task == 2:
                    assert learner.opt_step == 0                  # nothing was trained
                learner.after_task()
        states.append({k: v.detach().clone() for k, v in learner.model.state_dict().items()})
        if start_task == 0:
            assert sorted(f for f in os.listdir("./saved_models/t") if f.endswith(".pth")) == \
                ["A_0_0_best_score.pth", "B_1_0_best_score.pth", "B_1_1_best_score.pth"]
        else:
            assert not any(p.requires_grad for p in learner.model.model[-1].parameters())     # what update_step1 leaves behind
    trained, resumed = states
    assert trained.keys() == resumed.keys()
    for k in trained:
        assert torch.equal(trained[k], resumed[k]), k


def test_driver_runs_the_reference_task_loop_over_the_data_manager(tmp_path):
    """mrn_amd.tiny_train.train() (reference tiny_train.py:195-294) end to end: learner picked by opt.il, the array-backed
    Dataset_Manager with rehearsal-memory mixing and GPU staging, Val_Dataset loaders, per-task test pass -- two MRN tasks"""
    import numpy as np
    from torch.utils.data import ConcatDataset
    from mrn_amd import tiny_train
    from mrn_amd.data.data_manage import Dataset_Manager, Val_Dataset
    from mrn_amd.data.dataset import ArrayDataset
    from tests.helpers import fake_text_samples
    os.chdir(tmp_path)
    opt = make_opt(tmp_path, "crnn")
    opt.__dict__.update(select_data=["rootA"], valid_datas=["valA"], Aug="None", memory_num=20, batch_size=6, num_iter=4, val_interval=2)

    def open_fake(path, o, mode="train"):
        images, labels = fake_text_samples(path)
        return ArrayDataset(images, labels, o, mode)

    np.random.seed(3)
    torch.manual_seed(3)
    dm = Dataset_Manager(opt, open_dataset=open_fake)
    valid = Val_Dataset(["valA/A", "valA/B"], opt, open_tree=lambda root, o, mode: (ConcatDataset([open_fake(root, o, mode)]), "log"))
    alphabet = {0: "abcdefghijklmnopqrstuvwxyz", 1: "abcdefghijklmnopqrstuvwxyzAB"}
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        learner, best, ned = tiny_train.train(opt, io.StringIO(), data=(dm, valid, lambda t: alphabet[t],
                                                                         lambda t: [valid.create_dataset("valA/A")]))
    assert isinstance(learner, tiny_train.MRN) and len(best) == 2 and len(ned) == 2
    assert len(learner.memory_index) == 1 and len(learner.memory_index[0]) == 20        # memory of task 0 for the router phase
    assert len(learner.model.model) == 2
    assert dm.stager.stream is not None                                                  # batches were staged on a side stream
    assert "Train_taski_loss" in sink.getvalue()
    assert sorted(f for f in os.listdir("./saved_models/t") if f.endswith(".pth")) == \
        ["A_0_0_best_score.pth", "B_1_0_best_score.pth", "B_1_1_best_score.pth"]


def test_config_loader_accepts_the_reference_layout(tmp_path):
    from mrn_amd import tiny_train
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    opt = tiny_train.load_config(os.path.join(root, "config", "crnn_mrn_synthetic.py"))
    assert opt.il == "mrn" and opt.FeatureExtraction == "VGG" and opt.lr == 0.0005 and opt.lan_list[0] == "Chinese"
    assert tiny_train.make_learner.__call__ is not None
    (tmp_path / "Latin").mkdir()
    (tmp_path / "Latin" / "dict.txt").write_text("a\nb\na\nc\n")
    character, char = tiny_train.load_dict(str(tmp_path / "Latin"), {"z": 1})
    assert character == ["z", "a", "b", "c"]


@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_learner_step_matches_torch_clip_optimizer_onecycle(optimizer):
    """a17 at learner level: BaseLearner.backward_and_step (global-norm clip to 5 -> flat Adam / SGD kernel -> OneCycle) against the
    reference's own composition (il_modules/base.py:72-114,255-262: clip_grad_norm_ + torch.optim step + OneCycleLR.step) replayed
    on the CPU with the gradients the HIP backward produced -- eight consecutive steps, every parameter within 2e-6"""
    import bench
    from mrn_amd.data.synthetic import SyntheticTextLines
    from mrn_amd.tools import weights as W
    torch.manual_seed(7)
    opt = bench.make_opt("crnn", 8)
    opt.optimizer, opt.num_iter, opt.lr = optimizer, 12, 0.02 if optimizer == "sgd" else 1e-3
    opt.sgd_momentum, opt.sgd_weight_decay = 0.9, 5e-4
    learner = bench.build_loop_a_learner(opt)
    learner.build_optimizer(learner.count_param(), total_steps=opt.num_iter)
    W.fill_state_dict(learner.model.state_dict(), seed=29)
    torch.autograd.graph.increment_version(learner.optimizer.params)
    params = learner.optimizer.params
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in params]
    if optimizer == "adam":
        topt = torch.optim.Adam(ref, lr=opt.lr)
    else:
        topt = torch.optim.SGD(ref, lr=opt.lr, momentum=opt.sgd_momentum, weight_decay=opt.sgd_weight_decay)
    sched = torch.optim.lr_scheduler.OneCycleLR(topt, max_lr=opt.lr, cycle_momentum=(optimizer == "sgd"), div_factor=20,
                                                final_div_factor=1000, total_steps=opt.num_iter)
    data = SyntheticTextLines(opt, seed=5)
    data.set_characters(learner.character)
    captured = {}
    step_fn = learner.optimizer.step

    def spy(*a, **k):                                   # the gradients the update is about to consume (after zero_grad + backward)
        captured["g"] = [p.grad.detach().cpu().clone() for p in params]
        return step_fn(*a, **k)
    learner.optimizer.step = spy
    for it in range(8):
        learner.train_step(*data.get_batch())
        for r, g in zip(ref, captured["g"]):
            r.grad = g
        # clip_grad_norm_ with the norm accumulated in float64: torch's CPU float32 vector_norm is itself off by 4e-5 relative on
        # 9.3 M elements (the HIP norm kernel is within 2e-8 of the float64 value), which momentum SGD would amplify
        total = torch.sqrt(sum(r.grad.double().pow(2).sum() for r in ref))
        coef = min(1.0, opt.grad_clip / (float(total) + 1e-6))
        for r in ref:
            r.grad.mul_(coef)
        topt.step()
        sched.step()
        for i, (p, r) in enumerate(zip(params, ref)):
            err = (p.detach().cpu() - r.detach()).abs().max().item()
            assert err <= 2e-6 * max(1.0, r.detach().abs().max().item()), (it, i, err)


def test_full_size_loop_b_pipeline_is_deterministic_and_order_independent():
    """Soak of the three-stream loop-B schedule at BASELINE size (TRBA x 6, B = 256, ten routing steps): two pipelined runs (frozen
    experts of batch n+1 issued on three side streams before batch n's router step) end in bit-identical losses, router parameters
    and BatchNorm buffers -- no stream race -- and the same losses / parameters as the run that issues the experts inside the step"""
    import bench
    from mrn_amd.data.synthetic import SyntheticTextLines
    from mrn_amd.tools import weights as W
    from mrn_amd.tools.utils import to_device

    def run(prefetch, steps=10):
        torch.manual_seed(3)
        opt = bench.make_opt("trba", 256)
        learner = bench.build_learner(opt, 6)
        W.fill_state_dict(learner.model.state_dict(), seed=11)
        torch.autograd.graph.increment_version(list(learner.model.parameters()))
        learner.prepare_routing(total_steps=10 ** 9)
        data = SyntheticTextLines(opt, seed=9)
        data.set_characters(learner.character)
        pending, losses = [], []

        def fetch():
            image, labels, idx = data.get_batch2()
            ix = to_device(torch.LongTensor(idx).squeeze())
            return image, labels, ix, (learner.prefetch_experts(image, labels) if prefetch else None)
        for _ in range(steps):
            if not pending:
                pending.append(fetch())
            image, labels, ix, pre = pending.pop(0)
            if prefetch:
                pending.append(fetch())
            lc, lt = learner.routing_step(image, labels, ix, prefetched=pre)
            losses.append((float(lc.detach()), float(lt.detach())))
        torch.cuda.synchronize()
        bufs = torch.cat([b.float().reshape(-1) for b in learner.model.buffers()])
        out = (losses, learner.optimizer.flat.clone(), bufs.clone())
        del learner
        torch.cuda.empty_cache()
        return out
    la, fa, ba = run(True)
    lb, fb, bb = run(True)
    assert la == lb and torch.equal(fa, fb) and torch.equal(ba, bb)
    lc, fc, _ = run(False)                      # (its BatchNorm buffers have seen one batch less: the look-ahead batch)
    assert la == lc and torch.equal(fa, fc)
