"""The frozen experts' LSTM layer (split-fp16 x3 recurrent product) at BASELINE sizes: G experts x 2 directions, B = 256, T = 65 (CRNN) /
T = 26-step geometry of TRBA's encoder (T = 65 as well).  MRN_LSTM_RB = 1 / 2 forces the 16- / 32-sample tile form (read once per
process: run the script once per form for an A/B)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops
dev = torch.device("cuda:0")
Hd, T = 256, 65
print("MRN_LSTM_RB =", os.environ.get("MRN_LSTM_RB", "(default)"))
for G, B in ((1, 256), (3, 256), (6, 256), (6, 64), (3, 32)):
    torch.manual_seed(G)
    xproj = torch.randn(G, B, T, 8 * Hd, device=dev) * 0.7
    packs = [[ops.pack_fragment_major_h(torch.randn(4 * Hd, Hd, device=dev) / 16.0) for _ in range(2)] for _ in range(G)]
    w_h = torch.stack([torch.stack([d[0] for d in p]) for p in packs]).contiguous()
    w_inv = torch.stack([torch.cat([d[1] for d in p]) for p in packs]).contiguous()
    b_hh = torch.randn(G, 8 * Hd, device=dev) / 16.0
    fn = lambda: ops.lstm_layer_x3_grouped(xproj, w_h, w_inv, b_hh, Hd, 2)
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        fn()
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 10 * 1e3
    print(f"G {G} B {B:3d} T {T}: {us:7.1f} us / layer  {us / T:5.2f} us / step   checksum {fn().double().sum().item():.6f}")

# one layer being TRAINED: forward with saves, backward through time (B = 256, T = 65, both directions)
print("training forward (with saves) / backward")
B, ndir = 256, 2
torch.manual_seed(7)
xproj = torch.randn(B, T, ndir * 4 * Hd, device=dev) * 0.7
ws = [torch.randn(4 * Hd, Hd, device=dev) / 16.0 for _ in range(ndir)]
b_hh = torch.randn(ndir * 4 * Hd, device=dev) / 16.0
packs = [ops.pack_fragment_major_h(w) for w in ws]
w_h, w_inv = torch.stack([p[0] for p in packs]).contiguous(), torch.cat([p[1] for p in packs]).contiguous()
packsT = [ops.pack_fragment_major_h(w.t().contiguous()) for w in ws]
wT_h, wT_inv = torch.stack([p[0] for p in packsT]).contiguous(), torch.cat([p[1] for p in packsT]).contiguous()
fwd = lambda: ops.lstm_layer_x3_save(xproj, w_h, w_inv, b_hh, Hd, ndir)
out, gates, cseq = fwd()
dout = torch.randn(B, T, ndir * Hd, device=dev) * 1e-3
bwd = lambda: ops.lstm_layer_bwd_x3(dout, gates, cseq, wT_h, wT_inv, Hd, ndir)
for name, fn in (("forward ", fwd), ("backward", bwd)):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        fn()
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 10 * 1e3
    print(f"{name}: {us:7.1f} us / layer  {us / T:5.2f} us / step")
print("   checksum dgates %.8f" % bwd().double().sum().item())
