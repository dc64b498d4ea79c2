"""Grouped x3 Linear launches of the SVTR blocks in isolation: us, algorithmic TFLOP/s and HBM GB/s per variant.
usage (GPU box): python tools/bench_linear_x3.py"""
import sys
import torch
sys.path.insert(0, ".")
from mrn_amd import ops

dev = torch.device("cuda:0")
G = 3
for name, rows, K, N, act, hl in (("qkv  s1", 131072, 64, 192, 0, False), ("proj s1", 131072, 64, 64, 0, False),
                                  ("fc1  s1", 131072, 64, 256, 2, True), ("fc1  s1 no gelu", 131072, 64, 256, 0, True),
                                  ("fc1  s1 f32 out", 131072, 64, 256, 2, False), ("fc2  s1", 131072, 256, 64, 0, False),
                                  ("qkv  s2", 65536, 128, 384, 0, False), ("fc1  s2", 65536, 128, 512, 2, True),
                                  ("fc2  s2", 65536, 512, 128, 0, False), ("fc1  s3", 32768, 256, 1024, 2, True),
                                  ("fc2  s3", 32768, 1024, 256, 0, False)):
    x = torch.randn(G, rows, K, device=dev)
    w = [torch.randn(N, 1, 1, K, device=dev) * K ** -0.5 for _ in range(G)]
    b = torch.randn(G, N, device=dev)
    w_hl, sw = ops.pack_weights_hl32(w)
    x_hl = ops.split_hl32(x)

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
    us = t0.elapsed_time(t1) * 1e3 / 20
    flops = 2.0 * G * rows * K * N
    byts = 4.0 * G * rows * (K + N)
    print(f"{name:16s} rows {rows:6d} K {K:4d} N {N:4d} tile {ops.x3_tile(N, K, M=rows, G=G)}: {us:7.1f} us  {flops / us * 1e-6:6.1f} TF  {byts / us * 1e-3:6.0f} GB/s")
