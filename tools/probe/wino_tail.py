#!/usr/bin/env python3
"""G = 1 row-block Winograd launches by batch size: the cost of the ragged last round of workgroups (544 tiles at B = 256 on 4 x 65 maps)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrn_amd import ops


def timeit(fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


H, W, Cin, Cout, G = 4, 65, 512, 512, 1
ws = [(torch.rand(Cout, 3, 3, Cin, device="cuda") * 2 - 1) * 0.05 for _ in range(G)]
u_hl, u_scale = ops.pack_weights_wino(ws, 4)
order = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (120, 240, 256, 360, 376, 480)
for B in order:
    ypre = torch.randn(G, B, H, W, Cin, device="cuda")
    sc, sh = torch.ones(G, Cin, device="cuda"), torch.zeros(G, Cin, device="cuda")
    _, _, v = ops.bn_apply_wino_grouped(ypre, sc, sh, 4, relu=True)
    ms = timeit(lambda: ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u_hl, u_scale, Cout, 4, want_stats=True))
    tiles = -(-B * 17 // 64) * 8
    print(f"B={B}: {tiles} tiles = {tiles / 256:.3f} rounds, {ms * 1e3:.0f} us, {2.0 * B * H * W * Cout * 9 * Cin / ms / 1e9:.0f} TF")
