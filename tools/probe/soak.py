"""soak: N training steps of loop A (and loop B) -- finite losses, no growth of allocated memory after warm-up"""
import contextlib, io, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa
from mrn_amd import ops  # noqa
from mrn_amd.data.synthetic import SyntheticTextLines  # noqa

model, steps = sys.argv[1], int(sys.argv[2])
torch.cuda.set_device(0)
opt = bench.make_opt(model, 256)
learner = bench.build_loop_a_learner(opt, quiet=True)
data = SyntheticTextLines(opt, seed=5)
data.set_characters(learner.character)
marks = []
for i in range(steps):
    loss = learner.train_step(*data.get_batch())
    if i % 25 == 24 or i == steps - 1:
        torch.cuda.synchronize()
        marks.append((i + 1, float(loss), torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.max_memory_allocated() / 2 ** 30))
        assert torch.isfinite(loss).item(), marks
for m in marks:
    print("%s step %4d loss %.4f allocated %.2f GiB peak %.2f GiB" % ((model,) + m))
assert marks[-1][2] <= marks[1][2] * 1.02 + 0.05, "allocated memory grows"
print("stash sizes", len(ops._OPERANDS), len(ops._GRAD_OPERANDS), len(ops._SCALE_CACHE))
