#!/usr/bin/env python3
"""Correctness + speed of the grouped HL32 conv (conv_x3.hip) against the exact-fp32 kernel and the older DMA kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops  # noqa: E402

SHAPES_ALL = [  # B, H, W, Cin, Cout, k, s, p
    (256, 4, 65, 512, 512, (3, 3), (1, 1), (1, 1)),
    (256, 4, 65, 256, 512, (3, 3), (1, 1), (1, 1)),
    (256, 8, 64, 256, 256, (3, 3), (1, 1), (1, 1)),
    (256, 16, 128, 128, 128, (3, 3), (1, 1), (1, 1)),
    (256, 16, 128, 64, 128, (3, 3), (1, 1), (1, 1)),
    (256, 4, 65, 256, 512, (1, 1), (1, 1), (0, 0)),
    (256, 4, 65, 512, 512, (2, 2), (2, 1), (0, 1)),
    (256, 32, 256, 32, 64, (3, 3), (1, 1), (1, 1)),
]


SHAPES = SHAPES_ALL[:int(os.environ.get("NSHAPES", "99"))]


def timeit(fn, reps):
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    groups = [int(g) for g in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 6]
    check = "--no-check" not in sys.argv
    for (B, H, W, Cin, Cout, k, s, p) in SHAPES:
        for G in groups:
            torch.manual_seed(1)
            xs = [torch.rand(B, H, W, Cin, device="cuda") * 2 - 1 for _ in range(G)]
            ws = [(torch.rand(Cout, k[0], k[1], Cin, device="cuda") * 2 - 1) * 0.05 for _ in range(G)]
            if os.environ.get("ZERO_INPUTS"):          # DVFS probe: same instruction stream, no operand toggling
                xs = [t * 0 for t in xs]
                ws = [t * 0 for t in ws] if os.environ["ZERO_INPUTS"] == "2" else ws
            x = torch.stack(xs)
            x_hl = ops.split_hl32(x)
            w_hl, w_scale = ops.pack_weights_hl32(ws)
            y, st = ops.conv2d_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, k, s, p, want_stats=True, products=ops.X3_PRODUCTS)
            err = serr = -1.0
            if check:
                err = serr = 0.0
                for g in range(G if G <= 2 else 2):
                    ops.CONV_PRECISION = "f32"
                    yr, sr = ops.conv2d_nhwc(xs[g], ws[g], None, s, p, want_stats=True)
                    err = max(err, (y[g] - yr).abs().max().item() / yr.abs().max().item())
                    C = Cout
                    tot = st.view(G, -1, 2, C)[g].sum(0)
                    totr = sr.view(-1, 2, C).sum(0)
                    serr = max(serr, ((tot - totr).abs().max() / totr.abs().max()).item())
            flops = 2.0 * G * y.shape[1] * y.shape[2] * y.shape[3] * Cout * k[0] * k[1] * Cin
            ms = timeit(lambda: ops.conv2d_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, k, s, p, want_stats=True, products=ops.X3_PRODUCTS), reps)
            if "--only-x3" in sys.argv:
                print(f"conv G{G} B{B} {H}x{W} {Cin}->{Cout} k{k[0]}x{k[1]}: x3g {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF", flush=True)
                continue
            ops.CONV_PRECISION = "fp16x3"
            pw = [ops.PackedConvWeight(w) for w in ws]

            def old():
                for g in range(G):
                    ops.conv2d_nhwc(xs[g], pw[g], None, s, p, want_stats=True)
            ms_old = timeit(old, reps)
            ms_split = timeit(lambda: ops.split_hl32(x), reps)
            print(f"conv G{G} B{B} {H}x{W} {Cin}->{Cout} k{k[0]}x{k[1]}: x3g {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF | per-expert (G = 1 launches) {ms_old:8.3f} ms "
                  f"{flops / ms_old / 1e9:7.1f} TF | split_hl32 {ms_split:6.3f} ms | rel err {err:.2e} stats {serr:.2e}", flush=True)


if __name__ == "__main__":
    main()
