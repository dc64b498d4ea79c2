#!/usr/bin/env python3
"""Per-shape timing of the timed launches (convolutions, GEMMs, operand passes) inside loop A steps (HIP events on the launch streams)."""
import contextlib, io, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrn_amd import ops  # noqa: E402
from mrn_amd.data.synthetic import SyntheticTextLines  # noqa: E402


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "trba"
    steps = 3
    torch.cuda.set_device(0)
    opt = bench.make_opt(model, 256)
    learner = bench.build_loop_a_learner(opt, quiet=True)
    data = SyntheticTextLines(opt, seed=111)
    data.set_characters(learner.character)
    for _ in range(3):
        learner.train_step(*data.get_batch())
    ops.TIMER_SHAPES = True
    ops.CONV_TIMER = ops.KernelTimer()
    for _ in range(steps):
        learner.train_step(*data.get_batch())
    summ = ops.CONV_TIMER.summary()
    ops.CONV_TIMER = None
    rows = sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])
    print(f"timed launches: {sum(v['total_ms'] for _, v in rows) / steps:.2f} ms per step")
    for k, v in rows[:45]:
        n = v["launches"] / steps
        ms = v["total_ms"] / v["launches"]
        print(f"{v['total_ms'] / steps:7.2f} ms/step  x{n:4.1f}  {ms:7.3f} ms  {v['total_flops'] / v['total_ms'] / 1e9:6.1f} TF  {k}")


if __name__ == "__main__":
    main()
