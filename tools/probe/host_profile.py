#!/usr/bin/env python3
"""cProfile of the host side of loop-A steps (which Python frames the launch-bound SVTR step spends its time in).
    python tools/probe/host_profile.py [model] [steps] [sort]"""
import contextlib, cProfile, io, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrn_amd.data.synthetic import SyntheticTextLines  # noqa: E402


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "svtr"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    sort = sys.argv[3] if len(sys.argv) > 3 else "tottime"
    torch.cuda.set_device(0)
    opt = bench.make_opt(model, 256)
    with contextlib.redirect_stdout(io.StringIO()):
        learner = bench.build_loop_a_learner(opt, quiet=True)
    data = SyntheticTextLines(opt, seed=111)
    data.set_characters(learner.character)
    for _ in range(3):
        learner.train_step(*data.get_batch())
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    with torch.autograd.set_multithreading_enabled(False):      # backward nodes on this thread, where the profiler sees them
        learner.train_step(*data.get_batch())
        torch.cuda.synchronize()
        pr.enable()
        for _ in range(steps):
            learner.train_step(*data.get_batch())
        pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats(sort).print_stats(45)
    print(s.getvalue().replace(ROOT + "/", ""))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
    print(s.getvalue().replace(ROOT + "/", ""))


if __name__ == "__main__":
    main()
