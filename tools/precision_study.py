"""GPU study: loop-B parity error (router weights / fused logits vs the golden vectors) per conv-arithmetic policy.
usage: python tools/precision_study.py [case ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from mrn_amd import ops
from test_model_gpu import CASES, build_net, labels_for
from helpers import load_golden, det_inputs

POLICIES = [
    ("f32", dict(CONV_PRECISION="f32", LOCNET_CONV_PRECISION="f32")),
    ("auto bf16x3 K>=2304", dict(CONV_PRECISION="auto", AUTO_SPLIT_KIND="bf16x3", AUTO_SPLIT_MIN_K=2304, LOCNET_CONV_PRECISION="f32")),
    ("auto fp16x3 K>=2304", dict(CONV_PRECISION="auto", AUTO_SPLIT_KIND="fp16x3", AUTO_SPLIT_MIN_K=2304, LOCNET_CONV_PRECISION="f32")),
    ("auto fp16x3 all, loc f32", dict(CONV_PRECISION="auto", AUTO_SPLIT_KIND="fp16x3", AUTO_SPLIT_MIN_K=0, LOCNET_CONV_PRECISION="f32")),
    ("auto fp16x3 all, loc fp16x3", dict(CONV_PRECISION="auto", AUTO_SPLIT_KIND="fp16x3", AUTO_SPLIT_MIN_K=0, LOCNET_CONV_PRECISION="fp16x3")),
    ("bf16x3 all, loc f32", dict(CONV_PRECISION="bf16x3", LOCNET_CONV_PRECISION="f32")),
]

for name in (sys.argv[1:] or ["trba_mrn3", "crnn_mrn3"]):
    kind, classes, B, seed = CASES[name]
    g = load_golden(name)
    image, words, chars, _ = det_inputs(kind, classes, B, seed)
    conv, labels_index, _ = labels_for(kind, words, chars)
    for label, cfg in POLICIES:
        for k, v in cfg.items():
            setattr(ops, k, v)
        opt, net = build_net(kind, classes, g, seed)
        net.train()
        with torch.no_grad():
            if kind == "trba":
                out = net(image.cuda(), True, labels_index[:, :-1].cuda(), True)
            else:
                out = net(image.cuda(), True)
        w = out["index"].cpu().numpy()
        ew = np.abs(w - g["stepB/weights"]).max()
        from helpers import sub
        el = np.abs(sub(out["logits"])[0] - g["stepB/logits/sub"]).max()
        print(f"{name:10s} {label:30s} weights_err={ew:.2e} logits_err={el:.2e}", flush=True)
