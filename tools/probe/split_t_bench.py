#!/usr/bin/env python3
"""transposed split of dy with / without the fused column sums vs the separate column-sum passes, SVTR loop-A shapes"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrn_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for R, C in [(131072, 64), (131072, 256), (131072, 192), (65536, 128), (65536, 512), (65536, 384), (32768, 256), (32768, 1024), (32768, 768)]:
    x = torch.randn(R, C, device="cuda")
    sc = ops.pow2_scale(x)
    blocks = R // 32
    S = 64
    out = torch.zeros(C, device="cuda")
    t0 = timeit(lambda: ops.split_hl32_t(x, S, sc))
    t1 = timeit(lambda: ops.split_hl32_t(x, S, sc, colsum_out=out, accumulate=True))
    t2 = timeit(lambda: ops.colsum(x, out=out, accumulate=True))
    print(f"R={R:6d} C={C:4d}  split_t {t0:7.1f} us   +colsum fused {t1:7.1f} us   separate colsum {t2:7.1f} us   ({R * C * 8 / t0 / 1e6:.2f} TB/s plain)")
