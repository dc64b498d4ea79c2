#!/usr/bin/env python3
"""time per step of the frozen experts' grouped LSTM layer (split-fp16 x3 recurrence) by number of experts"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrn_amd import ops
Hd, T, B = 256, 65, 256
for G in (1, 2, 3, 4, 6):
    xproj = torch.randn(G, B, T, 8 * Hd, device="cuda") * 0.5
    packs = [[ops.pack_fragment_major_h(torch.randn(4 * Hd, Hd, device="cuda") / 16) for d in range(2)] for g in range(G)]
    w_h = torch.stack([torch.stack([d[0] for d in p]) for p in packs]).contiguous()
    w_inv = torch.stack([torch.cat([d[1] for d in p]) for p in packs]).contiguous()
    b_hh = torch.randn(G, 8 * Hd, device="cuda") / 16
    for _ in range(3):
        ops.lstm_layer_x3_grouped(xproj, w_h, w_inv, b_hh, Hd, 2)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.lstm_layer_x3_grouped(xproj, w_h, w_inv, b_hh, Hd, 2)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"G={G}: {ms:.3f} ms per layer, {ms / T * 1e3:.1f} us per step")
