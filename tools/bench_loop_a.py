#!/usr/bin/env python3
"""Loop A (il_modules/mrn.py:232-271: train the newest expert, cross=False) timing on synthetic crops."""
import contextlib
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrn_amd import ops  # noqa: E402
from mrn_amd.data.synthetic import SyntheticTextLines, synthetic_characters  # noqa: E402
from mrn_amd.il_modules.mrn import MRN  # noqa: E402


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "trba"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    if len(sys.argv) > 4:
        ops.TRAIN_CONV_PRECISION = sys.argv[4]
    torch.cuda.set_device(0)
    opt = bench.make_opt(model, batch)
    with contextlib.redirect_stdout(io.StringIO()):
        learner = MRN(opt)
        learner.character = synthetic_characters(2086)
        learner.converter = learner.build_converter()
        learner.criterion = learner.build_criterion()
        learner.build_model()
        learner.build_optimizer(learner.count_param())
    data = SyntheticTextLines(opt, seed=111)
    data.set_characters(learner.character)
    for _ in range(2):
        image, labels = data.get_batch()
        learner.train_step(image, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        image, labels = data.get_batch()
        loss = learner.train_step(image, labels)
    host = (time.perf_counter() - t0) / steps       # launch-side time: equal to dt when the host is the bottleneck
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"host launch time {host * 1e3:.1f} ms/step")
    print(f"loop A {model} B={batch} train precision {ops.TRAIN_CONV_PRECISION}: {dt * 1e3:.1f} ms/step, {batch / dt:.0f} images/s, loss {float(loss):.4f}")


if __name__ == "__main__":
    main()
