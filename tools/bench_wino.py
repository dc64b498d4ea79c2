#!/usr/bin/env python3
"""Winograd F(R,3) form of the frozen experts' 3x3 convolutions against the direct split-fp16 x3 kernel: kernel time, producer
pass time (BatchNorm-apply writing the Winograd-domain operand vs the plain HL32 operand) and error against the exact-fp32 kernel.
usage: bench_wino.py [reps] [groups, e.g. 2,6]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout
    (256, 4, 65, 512, 512),
    (256, 8, 64, 256, 256),
    (256, 16, 128, 128, 128),
][:int(os.environ.get("NSHAPES", "99"))]


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
    groups = [int(g) for g in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2][0] != "-" else [2, 6]
    zero = os.environ.get("ZERO_INPUTS")
    for (B, H, W, Cin, Cout) in SHAPES:
        for G in groups:
            torch.manual_seed(1)
            ypre = torch.randn(G, B, H, W, Cin, device="cuda")
            ws = [(torch.rand(Cout, 3, 3, Cin, device="cuda") * 2 - 1) * 0.05 for _ in range(G)]
            if zero:                       # DVFS probe: same instruction stream, no operand toggling
                ypre = ypre * 0
                ws = [t * 0 for t in ws] if zero == "2" else ws
            sc = torch.ones(G, Cin, device="cuda")
            sh = torch.zeros(G, Cin, device="cuda")
            if "--only-wino" in sys.argv:        # PMC passes (tools/pmc_kernel.sh): nothing but F(4,3) launches of this shape
                u_hl, u_scale = ops.pack_weights_wino(ws, 4)
                _, _, v = ops.bn_apply_wino_grouped(ypre, sc, sh, 4, relu=True)
                ms = timeit(lambda: ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u_hl, u_scale, Cout, 4, want_stats=True), reps)
                print(f"G{G} B{B} {H}x{W} {Cin}->{Cout}: F(4,3) {ms:7.3f} ms", flush=True)
                continue
            w_hl, w_scale = ops.pack_weights_hl32(ws)
            _, hl = ops.bn_apply_grouped(ypre.clone(), sc, sh, relu=True, want_f32=False, want_hl=True)
            yd, _ = ops.conv2d_x3(hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, (3, 3), (1, 1), (1, 1), want_stats=True)
            flops = 2.0 * G * B * H * W * Cout * 9 * Cin
            ms_d = timeit(lambda: ops.conv2d_x3(hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, (3, 3), (1, 1), (1, 1), want_stats=True), reps)
            ms_pd = timeit(lambda: ops.bn_apply_grouped(ypre, sc, sh, relu=True, want_f32=False, want_hl=True), reps)
            ops.CONV_PRECISION = "f32"
            act = torch.relu(ypre)
            exact = [ops.conv2d_nhwc(act[g], ws[g], None, (1, 1), (1, 1))[0] for g in range(min(G, 2))]
            ops.CONV_PRECISION = "auto"
            den = max(e.abs().max().item() for e in exact) or 1.0
            err_d = max((yd[g] - exact[g]).abs().max().item() for g in range(len(exact))) / den
            line = (f"G{G} B{B} {H}x{W} {Cin}->{Cout}: direct {ms_d:7.3f} ms {flops / ms_d / 1e9:6.1f} TF (producer {ms_pd:6.3f} ms, err {err_d:.1e})")
            for R in (4, 2):
                u_hl, u_scale = ops.pack_weights_wino(ws, R)
                _, _, v = ops.bn_apply_wino_grouped(ypre, sc, sh, R, relu=True)
                yw, _ = ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u_hl, u_scale, Cout, R, want_stats=True)
                err_w = max((yw[g] - exact[g]).abs().max().item() for g in range(len(exact))) / den
                ms_w = timeit(lambda: ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u_hl, u_scale, Cout, R, want_stats=True), reps)
                ms_pw = timeit(lambda: ops.bn_apply_wino_grouped(ypre, sc, sh, R, relu=True), reps)
                ms_pwh = timeit(lambda: ops.bn_apply_wino_grouped(ypre, sc, sh, R, relu=True, want_hl=True), reps)
                line += (f" | F({R},3) {ms_w:7.3f} ms {flops / ms_w / 1e9:6.1f} TF (producer {ms_pw:6.3f}, +HL32 {ms_pwh:6.3f} ms, err {err_w:.1e})")
            print(line, flush=True)


if __name__ == "__main__":
    main()
