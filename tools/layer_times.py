#!/usr/bin/env python3
"""Per-layer-shape timing of the grouped conv / Linear launches inside one bench step (HIP events on the launch stream)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrn_amd import ops  # noqa: E402
from mrn_amd.data.synthetic import SyntheticTextLines  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.cuda.set_device(0)
    model = os.environ.get("MODEL", "trba")
    opt = bench.make_opt(model, 256)
    learner = bench.build_learner(opt, int(os.environ.get("EXPERTS", "6")))
    if os.environ.get("SERIAL") == "1":          # one lock-step group of all experts on one stream: every launch has the GPU to itself
        learner.model.module.expert_halves = 0
    data = SyntheticTextLines(opt, seed=111)
    data.set_characters(learner.character)

    def step():
        image, labels, idx = data.get_batch2()
        return learner.routing_step(image, labels, torch.LongTensor(idx).squeeze().cuda())
    step()
    step()
    ops.TIMER_SHAPES = True
    ops.CONV_TIMER = ops.KernelTimer()
    for _ in range(steps):
        step()
    summ = ops.CONV_TIMER.summary()
    ops.CONV_TIMER = None
    rows = sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])
    tot = sum(v["total_ms"] for _, v in rows) / steps
    print(f"timed launches: {tot:.2f} ms per step")
    for k, v in rows:
        n = v["launches"] / steps
        ms = v["total_ms"] / v["launches"]
        tf = v["total_flops"] / v["total_ms"] / 1e9
        gbs = v["total_bytes"] / v["total_ms"] / 1e6
        print(f"{v['total_ms'] / steps:7.2f} ms/step  x{n:4.1f}  {ms:7.3f} ms  {tf:6.1f} TF  {gbs:6.0f} GB/s(alg)  {k}")


if __name__ == "__main__":
    main()
