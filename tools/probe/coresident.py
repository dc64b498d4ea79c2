#!/usr/bin/env python3
"""Can an HBM-bound pass share the CUs of the row-block Winograd kernel?  wino_rows_kernel allocates 464 of a SIMD's 512 registers and
144 of 160 KiB of LDS, one wave per SIMD: a kernel with <= 48 VGPRs and no LDS fits next to it.  Two half-groups (G = 3 each) on two
streams: conv(A) alone, pass(B) alone, both concurrently -- if the dispatcher co-schedules them the concurrent time approaches
max(conv, pass) instead of the sum."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrn_amd import ops

G, B, H, W, C = 3, 256, 4, 65, 512
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(3)
ya = torch.randn(G, B, H, W, C, device=dev, generator=gen)
yb = torch.randn(G, B, H, W, C, device=dev, generator=gen)
ws = [(torch.rand(C, 3, 3, C, device=dev, generator=gen) * 2 - 1) * 0.02 for _ in range(G)]
u_hl, u_scale = ops.pack_weights_wino(ws, 4)
sc, sh = torch.ones(G, C, device=dev), torch.zeros(G, C, device=dev)
_, _, v = ops.bn_apply_wino_grouped(ya, sc, sh, 4, relu=True)
which = sys.argv[1] if len(sys.argv) > 1 else "plain"
reps = 40


def conv():
    ops.conv2d_x3_wino(v, G, False, B, H, W, C, u_hl, u_scale, C, 4, want_stats=True)


def passb():
    if which == "plain":          # bn_apply_grouped_kernel: 42 VGPRs -- fits beside the Winograd kernel today
        ops.bn_apply_grouped(yb, sc, sh, relu=True, want_f32=False, want_hl=True)
    else:                         # bn_apply_wino_grouped_kernel: the pass the step actually runs between the convolutions
        ops.bn_apply_wino_grouped(yb, sc, sh, 4, relu=True)


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_stream(torch.cuda.current_stream())
    sb.wait_stream(torch.cuda.current_stream())
    for _ in range(reps):
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(2):
    conv(); passb()
for rnd in range(3):              # (clocks / power state settle over the first rounds: the last round counts)
    tc, tp, tb = timed(conv, None), timed(None, passb), timed(conv, passb)
    tc2 = timed(conv, None)
    print(f"  round {rnd}: conv {tc:.3f} / {tc2:.3f}, pass {tp:.3f}, concurrent {tb:.3f}")
print(f"pass={which}: conv alone {tc:.3f} ms, pass alone {tp:.3f} ms, sum {tc + tp:.3f}, concurrent {tb:.3f} ms "
      f"(overlap {(tc + tp - tb) / min(tc, tp):.2f} of the shorter)")
