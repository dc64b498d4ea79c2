"""INTEGRATION.md section A executed: the REFERENCE's own learner code (il_modules/mrn.py, il_modules/base.py) running on top of
this package's operator modules through the documented `sys.modules` aliases -- construction, growth, freezing, optimiser and
checkpoint plumbing, i.e. everything of the drop-in boundary that does not need a GPU.  Runs only where the reference checkout
exists (the build container); the GPU box has no /root/reference and skips it."""
import contextlib
import importlib
import io
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def test_reference_learner_runs_on_aliased_modules(tmp_path, monkeypatch):
    import mrn_amd.modules.model
    import mrn_amd.test
    import mrn_amd.tools.utils
    saved = dict(sys.modules)
    saved_path = list(sys.path)
    try:
        # the aliases of INTEGRATION.md section A
        sys.modules["modules.model"] = mrn_amd.modules.model
        sys.modules["tools.utils"] = mrn_amd.tools.utils
        sys.modules["test"] = mrn_amd.test
        # data-side imports of the reference that this image lacks (only touched by LMDB / CLI code)
        for name, attrs in (("lmdb", {}), ("natsort", {"natsorted": sorted}), ("cv2", {}), ("mmcv", {"Config": object}),
                            ("torchvision", {}), ("torchvision.transforms", {"Compose": object})):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
        sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
        sys.path.insert(0, REF)
        for k in [k for k in sys.modules if k == "il_modules" or k.startswith("il_modules.") or k == "data" or k.startswith("data.")]:
            del sys.modules[k]
        ref_mrn = importlib.import_module("il_modules.mrn")
        assert ref_mrn.MRNNet is mrn_amd.modules.model.MRNNet          # the reference learner now builds THIS package's model
        opt = types.SimpleNamespace(exp_name="g", il="mrn", memory="random", memory_num=2000, batch_max_length=25, imgH=32, imgW=256,
                                    manual_seed=111, start_task=0, num_fiducial=20, input_channel=4, output_channel=512,
                                    hidden_size=256, schedule="super", optimizer="adam", lr=0.0005, batch_size=4, num_iter=4,
                                    val_interval=2, grad_clip=5, lan_list=["A", "B"], NED=True, workers=0, Transformation="TPS",
                                    FeatureExtraction="ResNet", SequenceModeling="BiLSTM", Prediction="Attn")
        os.chdir(tmp_path)
        os.makedirs("saved_models/g")
        with contextlib.redirect_stdout(io.StringIO()):
            learner = ref_mrn.MRN(opt)
            for taski, nchar in enumerate((30, 50)):
                learner.character = "".join(chr(0x4E00 + i) for i in range(nchar))
                learner.converter = learner.build_converter()
                if taski == 0:
                    learner.criterion = learner.build_criterion()
                    learner.build_model()                       # reference code: build_fc, kaiming re-init, DataParallel wrap
                else:
                    learner.change_model()
                for i in range(taski):
                    for p in learner.model.module.model[i].parameters():
                        p.requires_grad = False
                learner.build_optimizer(learner.count_param())   # torch.optim.Adam + OneCycleLR over this package's parameters
                if taski == 0:
                    learner.after_task()                         # unwrap + deep copy through MRNNet.copy() / freeze()
        net = learner.model.module
        assert isinstance(net, mrn_amd.modules.model.MRNNet) and len(net.model) == 2
        assert net.model[1].fc.out_features == 55 and net.channel_route.in_features == 512
        assert not any(p.requires_grad for p in net.model[0].parameters())
        assert isinstance(learner._old_network, mrn_amd.modules.model.MRNNet) and not learner._old_network.training
        path = tmp_path / "ck.pth"
        torch.save(learner.model.state_dict(), path)             # reference checkpoint convention: DataParallel-prefixed keys
        sd = torch.load(path)
        assert all(k.startswith("module.") for k in sd)
        learner.model.load_state_dict(sd, strict=True)
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]
        sys.modules.update(saved)
