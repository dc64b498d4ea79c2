"""Tile sweep for the grouped x3 Linear shapes of the SVTR blocks (forces ops.x3_tile)."""
import sys
import torch
sys.path.insert(0, ".")
from mrn_amd import ops
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
orig = ops.x3_tile
for name, rows, K, N, act, hl in (("qkv s1", 131072, 64, 192, 0, False), ("proj s1", 131072, 64, 64, 0, False), ("fc1 s1", 131072, 64, 256, 2, True),
                                  ("fc2 s1", 131072, 256, 64, 0, False), ("qkv s2", 65536, 128, 384, 0, False), ("proj s2", 65536, 128, 128, 0, False),
                                  ("fc1 s2", 65536, 128, 512, 2, True), ("fc2 s2", 65536, 512, 128, 0, False),
                                  ("qkv s3", 32768, 256, 768, 0, False), ("proj s3", 32768, 256, 256, 0, False),
                                  ("fc1 s3", 32768, 256, 1024, 2, True), ("fc2 s3", 32768, 1024, 256, 0, False)):
    x = torch.randn(G, rows, K, device=dev)
    w = [torch.randn(N, 1, 1, K, device=dev) * K ** -0.5 for _ in range(G)]
    b = torch.randn(G, N, device=dev)
    w_hl, sw = ops.pack_weights_hl32(w)
    x_hl = ops.split_hl32(x)
    res = []
    for tile in ((256, 256), (256, 128), (128, 128), (256, 64)):
        ops.x3_tile = lambda *a, **k: tile
        def run():
            return ops.conv2d_x3(x_hl, G, False, rows, 1, 1, K, w_hl, sw, N, (1, 1), bias=b, act=act, hl_only=hl)
        for _ in range(3):
            run()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            run()
        t1.record()
        torch.cuda.synchronize()
        res.append((t0.elapsed_time(t1) * 1e3 / 20, tile))
    ops.x3_tile = orig
    print(f"{name:8s} K {K:4d} N {N:4d} default {orig(N, K, M=rows, G=G)}: " + "  ".join(f"{t[0]}x{t[1]} {us:6.1f}us" for us, t in res))
