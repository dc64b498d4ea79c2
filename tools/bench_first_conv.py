"""First conv (Cin = 4) of the experts' stacks: dedicated grouped kernel vs the generic fp32 implicit-GEMM kernel per expert."""
import sys
import torch
sys.path.insert(0, ".")
from mrn_amd import ops
dev = torch.device("cuda:0")
G, B, H, W = 3, 256, 32, 256
for Cout, shared in ((64, True), (32, False)):
    x = torch.randn(*((B, H, W, 4) if shared else (G, B, H, W, 4)), device=dev)
    w = torch.randn(G, Cout, 3, 3, 4, device=dev) * 0.2
    bias = torch.randn(G, Cout, device=dev)
    y = torch.empty(G, B, H, W, Cout, device=dev)

    def new():
        ops.conv3x3_c4_grouped(x, w, bias, want_stats=True, out=y)

    def old():
        for g in range(G):
            ops.conv2d_nhwc(x if shared else x[g], w[g], bias[g], (1, 1), (1, 1), want_stats=True, precision="f32", out=y[g])
    for name, fn in (("dedicated", new), ("generic x G", old)):
        for _ in range(3):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            fn()
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 100
        gb = (G * B * H * W * Cout * 4 + (1 if shared else G) * B * H * W * 16) / 1e9
        print(f"Cout {Cout} shared {shared} {name:12s}: {us:8.1f} us  {gb / us * 1e6:6.0f} GB/s")
