#!/usr/bin/env python3
"""Bandwidth of the grouped BatchNorm-apply / pool passes at the TRBA backbone's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    G, B = 6, 256
    for (H, W, C) in [(4, 65, 512), (8, 64, 256), (16, 128, 128), (32, 256, 64)]:
        y = torch.rand(G, B, H, W, C, device="cuda")
        res = torch.rand(G, B, H, W, C, device="cuda")
        scale, shift = torch.rand(G, C, device="cuda"), torch.rand(G, C, device="cuda")
        n = y.numel() * 4
        dst = torch.empty_like(y)
        ms = timeit(lambda: dst.copy_(y))
        print(f"torch copy {H}x{W}x{C} ({n / 1e6:.0f} MB in + out)   {ms:7.3f} ms  {2 * n / ms / 1e6:7.0f} GB/s")
        for name, kw, nb in [("hl only", dict(want_f32=False, want_hl=True), 2 * n),
                             ("res + f32 + hl", dict(residual=res, want_f32=True, want_hl=True), 4 * n),
                             ("res + hl", dict(residual=res, want_f32=False, want_hl=True), 3 * n)]:
            ms = timeit(lambda: ops.bn_apply_grouped(y, scale, shift, relu=True, **kw))
            print(f"bn_apply {H}x{W}x{C} {name:16s} {ms:7.3f} ms  {nb / ms / 1e6:7.0f} GB/s")
        if C >= 128:
            Wq = (W + 3) // 4
            nv = G * B * H * Wq * 6 * C * 4
            res_hl = ops.split_hl32(res)
            for name, kw, nb in [("wino only", dict(), n + nv), ("wino + res(hl) + hl", dict(residual_hl=res_hl, want_hl=True), 3 * n + nv),
                                 ("wino + res(f32) + f32", dict(residual=res, want_f32=True), 3 * n + nv)]:
                ms = timeit(lambda: ops.bn_apply_wino_grouped(y, scale, shift, 4, relu=True, **kw))
                print(f"bn_apply_wino {H}x{W}x{C} {name:22s} {ms:7.3f} ms  {nb / ms / 1e6:7.0f} GB/s")
        if H >= 8:
            ms = timeit(lambda: ops.maxpool_grouped(y, (2, 2), (2, 2), (0, 0), scale, shift, relu=True, want_f32=False, want_hl=True))
            print(f"maxpool  {H}x{W}x{C} -> hl          {ms:7.3f} ms  {(n + n / 4) / ms / 1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()
