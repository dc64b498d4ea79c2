#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernel on the shapes of the TRBA backbone (B=256)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout, k, s, p
    (256, 4, 65, 512, 512, (3, 3), (1, 1), (1, 1)),
    (256, 4, 65, 256, 512, (3, 3), (1, 1), (1, 1)),
    (256, 8, 64, 256, 256, (3, 3), (1, 1), (1, 1)),
    (256, 16, 128, 128, 128, (3, 3), (1, 1), (1, 1)),
    (256, 32, 256, 32, 64, (3, 3), (1, 1), (1, 1)),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    if len(sys.argv) > 2:
        ops.CONV_PRECISION = sys.argv[2]
    if len(sys.argv) > 3:
        ops.USE_DMA_CONV = sys.argv[3] == "dma"
        ops.DMA_MIN_CIN = 0
    print("precision", ops.CONV_PRECISION)
    for (B, H, W, Cin, Cout, k, s, p) in SHAPES:
        x = torch.rand(B, H, W, Cin, device="cuda") * 2 - 1
        w = ops.PackedConvWeight((torch.rand(Cout, k[0], k[1], Cin, device="cuda") * 2 - 1) * 0.05)
        ops.conv2d_nhwc(x, w, None, s, p, want_stats=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y, st = ops.conv2d_nhwc(x, w, None, s, p, want_stats=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * k[0] * k[1] * Cin
        print(f"conv B{B} {H}x{W} {Cin}->{Cout} k{k[0]}: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
