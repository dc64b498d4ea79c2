import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mrn_amd import ops
Hd, T = 256, 65
for G, B in ((1, 256), (1, 512), (1, 1024), (2, 128), (2, 256), (6, 64), (6, 128), (6, 256)):
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
    print(f"G={G} B={B}: {G * 2 * ((B + 15) // 16)} workgroups, {ms / T * 1e3:.1f} us per step")
