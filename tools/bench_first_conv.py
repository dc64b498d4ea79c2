"""First conv (Cin = 4) of the experts' stacks at BASELINE sizes (G = 6, B = 256, 32 x 256): full map and pooled form."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops
dev = torch.device("cuda:0")
G, B, H, W = 6, 256, 32, 256
for Cout, shared in ((64, True), (32, False)):
    x = torch.randn(*((B, H, W, 4) if shared else (G, B, H, W, 4)), device=dev)
    w = torch.randn(G, Cout, 3, 3, 4, device=dev) * 0.2
    y = torch.empty(G, B, H, W, Cout, device=dev)
    yp = torch.empty(G, B, H // 2, W // 2, Cout, device=dev)
    for name, fn, out_gb in (("full map", lambda: ops.conv3x3_c4_grouped(x, w, None, want_stats=True, out=y), y.numel() * 4 / 1e9),
                             ("pooled", lambda: ops.conv3x3_c4_grouped(x, w, None, want_stats=True, out=yp, pool=True), yp.numel() * 4 / 1e9)):
        for _ in range(3):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            fn()
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 10
        gb = out_gb + (1 if shared else G) * B * H * W * 16 / 1e9
        print(f"Cout {Cout} shared {shared} {name:9s}: {ms:6.3f} ms  {gb / ms * 1e3:6.0f} GB/s  {2.0 * G * B * H * W * Cout * 36 / ms / 1e9:5.1f} TF (fp32 MFMA)")
