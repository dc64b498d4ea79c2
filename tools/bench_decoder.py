"""The frozen experts' attention decoder (split-fp16 x3 recurrent products) at BASELINE sizes: G experts, B = 256, T = 65 encoder
positions, D = 256, S = 26 decode steps (one launch for all experts and steps)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops
dev = torch.device("cuda:0")
Hd, T, D, S = 256, 65, 256, 26
for G, B in ((1, 256), (3, 256), (6, 256), (6, 32)):
    torch.manual_seed(G)
    Hb, Hproj = torch.randn(G, B, T, D, device=dev), torch.randn(G, B, T, Hd, device=dev)
    eproj = torch.randn(G, B, S, 4 * Hd, device=dev) * 0.5
    mk = lambda *shape: [torch.randn(*shape, device=dev) / 16.0 for _ in range(G)]
    packs = [ops.pack_decoder_x3(a, b, c, D) for a, b, c in zip(mk(Hd, Hd), mk(4 * Hd, D + 4), mk(4 * Hd, Hd))]
    w_h2h, w_ih, w_hh, w_inv = ([p[i] for p in packs] for i in range(4))
    b_h2h, w_score, b_hh = mk(Hd), mk(1, Hd), mk(4 * Hd)
    fn = lambda: ops.attn_decoder_grouped(Hb, Hproj, eproj, w_h2h, b_h2h, w_score, w_ih, w_hh, b_hh, Hd, w_inv=w_inv)
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        fn()
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 10 * 1e3
    print(f"G {G} B {B:3d}: {us:7.1f} us / launch  {us / S:6.2f} us / step   checksum {fn().double().sum().item():.6f}")

# the decoder being TRAINED: forward with saves + backward through the 26 steps (one network; D = 256: a TRBA expert, D = 1536: DERNet's
# main head over six extractors' features)
print("training forward (with saves) / backward, B = 256")
for D in (256, 1536):
    torch.manual_seed(D)
    B = 256
    Hb, Hproj = torch.randn(B, T, D, device=dev), torch.randn(B, T, Hd, device=dev)
    eproj = torch.randn(B, S, 4 * Hd, device=dev) * 0.5
    h2h, w_ih, w_hh = torch.randn(Hd, Hd, device=dev) / 16, torch.randn(4 * Hd, D, device=dev) / 16, torch.randn(4 * Hd, Hd, device=dev) / 16
    b_h2h, w_score, b_hh = torch.randn(Hd, device=dev) / 16, torch.randn(1, Hd, device=dev) / 16, torch.randn(4 * Hd, device=dev) / 16
    f = [ops.pack_fragment_major_h(w) for w in (h2h, w_ih, w_hh)]
    f_inv = torch.cat([x[1] for x in f]).contiguous()
    bT = [ops.pack_fragment_major_h(w.t().contiguous()) for w in (h2h, w_ih, w_hh)]
    b_inv = torch.cat([x[1] for x in bT]).contiguous()
    dhid = torch.randn(B, S, Hd, device=dev) * 0.1
    fwd = lambda: ops.attn_decoder_train(Hb, Hproj, eproj, f[0][0], b_h2h, w_score, f[1][0], f[2][0], b_hh, Hd, w_inv=f_inv)
    hid, saves = fwd()
    bwd = lambda: ops.attn_decoder_bwd(Hb, Hproj, saves, dhid, w_score, bT[0][0], bT[1][0], bT[2][0], Hd, w_inv=b_inv)
    for name, fn in (("forward ", fwd), ("backward", bwd)):
        for _ in range(2):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            fn()
        t1.record()
        torch.cuda.synchronize()
        print(f"D {D:4d} {name}: {t0.elapsed_time(t1) / 5 * 1e3:8.1f} us")
    out = bwd()
    print("   checksums dHb %.5f dHproj %.5f dgates %.5f" % (out[2].double().sum().item(), out[3].double().sum().item(), out[0].double().sum().item()))
