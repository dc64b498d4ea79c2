#!/usr/bin/env python3
"""The narrow early 3x3 layers alone at BASELINE sizes (G = 6, B = 256): patch-resident kernel (csrc/conv_patch.hip), plain and pooled,
next to the tiled x3 kernel (csrc/conv_x3.hip) on the same operands.  usage: python tools/bench_patch.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrn_amd import ops  # noqa: E402


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    G, B = 6, 256
    for (H, W, Cin, Cout) in ((32, 256, 32, 64), (16, 128, 64, 128)):
        x = torch.relu(torch.randn(G, B, H, W, Cin, device="cuda"))
        ws = [torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.05 for _ in range(G)]
        x_hl = ops.split_hl32(x)
        w_hl, w_scale = ops.pack_weights_hl32(ws)
        del x
        y = torch.empty(G, B, H, W, Cout, device="cuda")
        yp = torch.empty(G, B, H // 2, W // 2, Cout, device="cuda")
        flops = 2.0 * G * B * H * W * Cout * 9 * Cin
        t_old = timeit(lambda: ops.conv2d_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, (3, 3), (1, 1), (1, 1), want_stats=True, out=y))
        t_new = timeit(lambda: ops.conv3x3_patch_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, want_stats=True, out=y))
        t_pool = timeit(lambda: ops.conv3x3_patch_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, want_stats=True, pool=True, out=yp))
        gb_in, gb_out = x_hl.numel() / 1e9, y.numel() * 4 / 1e9
        print(f"{Cin}->{Cout} {H}x{W}: tiled x3 {t_old:.3f} ms ({flops / t_old / 1e9:.0f} TF alg) | patch {t_new:.3f} ms ({flops / t_new / 1e9:.0f} TF, "
              f"{(gb_in + gb_out) / t_new * 1e3:.0f} GB/s) | patch pooled {t_pool:.3f} ms ({flops / t_pool / 1e9:.0f} TF, {(gb_in + gb_out / 4) / t_pool * 1e3:.0f} GB/s)")


if __name__ == "__main__":
    main()
