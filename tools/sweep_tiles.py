#!/usr/bin/env python3
"""Tile sweep of the grouped x3 conv over the VGG / ResNet layer shapes at small group counts (CRNN x 3 runs as half-groups of 2 + 1
experts): time per launch for every tile the kernel has, next to the tile ops.x3_tile() picks.   python tools/sweep_tiles.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops  # noqa: E402
from tools.bench_conv_x3 import timeit  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, Cout, k, s, p
    ("vgg1", 256, 16, 128, 64, 128, (3, 3), (1, 1), (1, 1)),
    ("vgg2", 256, 8, 64, 128, 256, (3, 3), (1, 1), (1, 1)),
    ("vgg3", 256, 8, 64, 256, 256, (3, 3), (1, 1), (1, 1)),
    ("vgg4", 256, 4, 64, 256, 512, (3, 3), (1, 1), (1, 1)),
    ("vgg5", 256, 4, 64, 512, 512, (3, 3), (1, 1), (1, 1)),
    ("vgg6", 256, 2, 64, 512, 512, (2, 2), (1, 1), (0, 0)),
    ("res4x65", 256, 4, 65, 512, 512, (3, 3), (1, 1), (1, 1)),
    # SVTR mixing-block Linears (1x1): stage 1 (512 tokens, C = 64), stage 2 (256 tokens, C = 128), stage 3 (128 tokens, C = 256)
    ("s1.qkv", 256, 1, 512, 64, 192, (1, 1), (1, 1), (0, 0)),
    ("s1.fc1", 256, 1, 512, 64, 256, (1, 1), (1, 1), (0, 0)),
    ("s1.fc2", 256, 1, 512, 256, 64, (1, 1), (1, 1), (0, 0)),
    ("s2.qkv", 256, 1, 256, 128, 384, (1, 1), (1, 1), (0, 0)),
    ("s2.fc1", 256, 1, 256, 128, 512, (1, 1), (1, 1), (0, 0)),
    ("s2.fc2", 256, 1, 256, 512, 128, (1, 1), (1, 1), (0, 0)),
    ("s3.qkv", 256, 1, 128, 256, 768, (1, 1), (1, 1), (0, 0)),
    ("s3.fc1", 256, 1, 128, 256, 1024, (1, 1), (1, 1), (0, 0)),
    ("s3.fc2", 256, 1, 128, 1024, 256, (1, 1), (1, 1), (0, 0)),
]
if os.environ.get("SWEEP_ONLY"):
    SHAPES = [s_ for s_ in SHAPES if s_[0].startswith(os.environ["SWEEP_ONLY"])]
GROUPS = [int(g) for g in os.environ.get("SWEEP_G", "1,2,3").split(",")]
TILES = ["256x256", "256x128", "128x128", "256x64"]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for (name, B, H, W, Cin, Cout, k, s, p) in SHAPES:
        for G in GROUPS:
            torch.manual_seed(1)
            x = torch.rand(G, B, H, W, Cin, device="cuda") * 2 - 1
            ws = [(torch.rand(Cout, k[0], k[1], Cin, device="cuda") * 2 - 1) * 0.05 for _ in range(G)]
            x_hl = ops.split_hl32(x)
            w_hl, w_scale = ops.pack_weights_hl32(ws)
            Ho, Wo = ops.conv_out_hw(H, W, k, s, p)
            os.environ.pop("MRN_X3_TILE", None)
            pick = "%dx%d" % ops.x3_tile(Cout, k[0] * k[1] * Cin, M=B * Ho * Wo, G=G)
            res = {}
            for t in TILES:
                if t == "256x64" and Cout > 128:
                    continue
                os.environ["MRN_X3_TILE"] = t
                res[t] = timeit(lambda: ops.conv2d_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, k, s, p, want_stats=True), reps)
            os.environ.pop("MRN_X3_TILE", None)
            best = min(res, key=res.get)
            flag = "" if best == pick or res[pick] <= 1.03 * res[best] else "   <-- heuristic loses %.0f %%" % ((res[pick] / res[best] - 1) * 100)
            flops = 2.0 * G * B * Ho * Wo * Cout * k[0] * k[1] * Cin
            print(f"{name} G{G} [{flops / res[pick] / 1e9:5.0f} TF]: " + "  ".join(f"{t} {v * 1e3:7.1f}us" for t, v in res.items()) + f"  | pick {pick}{flag}", flush=True)


if __name__ == "__main__":
    main()
