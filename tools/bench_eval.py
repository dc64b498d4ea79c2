#!/usr/bin/env python3
"""Inference throughput of the routed ensemble (MRNNet.cross_forward_expert + greedy decoding, the model call of
validation(..., val_choose="TF"), reference test.py:163-201 / modules/model.py:366-395) on synthetic 32x256 crops.
    python tools/bench_eval.py [trba|crnn|svtr] [experts] [batch] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrn_amd.data.synthetic import SyntheticTextLines  # noqa: E402
from mrn_amd.test import _forward  # noqa: E402


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "trba"
    experts = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    torch.cuda.set_device(0)
    torch.manual_seed(111)
    opt = bench.make_opt(model, batch)
    learner = bench.build_learner(opt, experts)
    data = SyntheticTextLines(opt, seed=opt.manual_seed)
    data.set_characters(learner.character)
    net = learner.model
    net.eval()
    image, _ = data.get_batch()
    with torch.no_grad():
        for _ in range(2):
            out = _forward(net, image, opt, learner.converter, "TF")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = _forward(net, image, opt, learner.converter, "TF")
            idx = out.argmax(2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print(f"eval routing {model} x {experts}, B={batch}: {dt * 1e3:.1f} ms/batch, {batch / dt:.0f} images/s "
          f"(logits {tuple(out.shape)}, greedy indices {tuple(idx.shape)})")


if __name__ == "__main__":
    main()
