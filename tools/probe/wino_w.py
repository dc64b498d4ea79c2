#!/usr/bin/env python3
"""Row-block Winograd launch time on 4 x W maps, W = 64 vs 65 (512 -> 512, B = 256): what the 17th column group of the 65-wide maps
(one valid column, a full group's work, and a ragged extra round of workgroups) costs at G = 1 and G = 6"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrn_amd import ops


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


H, Cin, Cout, B = 4, 512, 512, 256
for G in (1, 6):
    ws = [(torch.rand(Cout, 3, 3, Cin, device="cuda") * 2 - 1) * 0.05 for _ in range(G)]
    u_hl, u_scale = ops.pack_weights_wino(ws, 4, dense=False)
    for rnd in range(2):
        for W in (64, 65):
            ypre = torch.randn(G, B, H, W, Cin, device="cuda")
            _, _, v = ops.bn_apply_wino_grouped(ypre, None, None, 4, relu=True, dense=False)
            ms = timeit(lambda: ops.conv2d_x3_wino(v, G, False, B, H, W, Cin, u_hl, u_scale, Cout, 4, want_stats=True, dense=False))
            msb = timeit(lambda: ops.bn_apply_wino_grouped(ypre, None, None, 4, relu=True, dense=False))
            tiles = -(-B * ((W + 3) // 4) // 64) * 8 * G
            print(f"G={G} W={W}: {tiles} tiles = {tiles / 256:.3f} rounds, conv {ms * 1e3:.0f} us, producer pass {msb * 1e3:.0f} us")
