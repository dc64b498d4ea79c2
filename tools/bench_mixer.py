#!/usr/bin/env python3
"""The fused SVTR attention half-block (mrn_svtr_mixer_x3_f32) at the headline's sizes against the chain it replaces.
usage: bench_mixer.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops  # noqa: E402
from mrn_amd.modules.svtr import local_attention_mask  # noqa: E402

CASES = [(128, 4, 64, False), (64, 8, 64, True), (128, 4, 64, True), (64, 8, 25, True), (128, 4, 25, True), (128, 4, 25, False)][:int(os.environ.get("NSHAPES", "99"))]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    G, B = 6, 256
    dev = torch.device("cuda")
    only_fused = "--only-fused" in sys.argv
    for C, H, W, local in CASES:
        N = H * W
        heads = C // 32
        torch.manual_seed(0)
        x = torch.randn(G * B, N, C, device=dev)
        pend = torch.randn(G * B, N, C, device=dev)
        dprev = torch.ones(G * B, device=dev)
        d1 = torch.ones(G * B, device=dev)
        g1, b1 = torch.ones(G, C, device=dev), torch.zeros(G, C, device=dev)
        wqkv = [torch.randn(3 * C, C, device=dev) * C ** -0.5 for _ in range(G)]
        wproj = [torch.randn(C, C, device=dev) * C ** -0.5 for _ in range(G)]
        bqkv, bproj = torch.randn(G, 3 * C, device=dev) * 0.1, torch.randn(G, C, device=dev) * 0.1
        mask = local_attention_mask(H, W, 7, 11).to(dev) if local else None
        wq, sq = ops.pack_weights_hl32([w.view(3 * C, 1, 1, C).contiguous() for w in wqkv])
        perm = ops.mlp_hidden_permutation(C, dev)
        wp, sp = ops.pack_weights_hl32([w.index_select(1, perm).contiguous().view(C, 1, 1, C) for w in wproj])
        wpn, spn = ops.pack_weights_hl32([w.view(C, 1, 1, C).contiguous() for w in wproj])
        rows = B * N

        def fused():
            return ops.svtr_mixer_fused(x, pend, dprev, g1, b1, 1e-6, wq, sq, bqkv, mask, 32 ** -0.5, wp, sp, bproj, d1, g1, b1, 1e-6, B,
                                        hw=(H, W) if local else None)      # (MRN_SVTR_LOCAL_COLUMNS=0: the memory-order walk, A/B)

        def chain():
            t, _, hl = ops.add_layernorm_grouped(x, pend, dprev, N, g1, b1, rows, 1e-6, want_sum=True)
            qkv, _ = ops.conv2d_x3(hl, G, False, rows, 1, 1, C, wq, sq, 3 * C, (1, 1), bias=bqkv)
            ctx = ops.svtr_attention(qkv.view(G * B, N, 3 * C), heads, 32 ** -0.5, mask, want_f32=False, want_hl=True, x3=True)
            br, _ = ops.conv2d_x3(ctx, G, False, rows, 1, 1, C, wpn, spn, C, (1, 1), bias=bproj)
            return ops.add_layernorm_grouped(t, br.view(G * B, N, C), d1, N, g1, b1, rows, 1e-6, want_sum=True)

        Ch = 4 * C
        w1 = [torch.randn(Ch, C, device=dev) * C ** -0.5 for _ in range(G)]
        w2 = [torch.randn(C, Ch, device=dev) * Ch ** -0.5 for _ in range(G)]
        bm1, bm2 = torch.randn(G, Ch, device=dev) * 0.1, torch.randn(G, C, device=dev) * 0.1
        ph = ops.mlp_hidden_permutation(Ch, dev)
        w1n, s1n = ops.pack_weights_hl32([w.view(Ch, 1, 1, C).contiguous() for w in w1])
        w2p, s2 = ops.pack_weights_hl32([w.index_select(1, ph).contiguous().view(C, 1, 1, Ch) for w in w2])

        def halves():
            xo, yhl = fused()
            return ops.svtr_mlp_fused(yhl, G * rows, rows, G, C, w1n, s1n, bm1, w2p, s2, bm2)

        ms_h = timeit(halves, reps)
        print(f"C{C} N{N} ({H}x{W}) {'local ' if local else 'global'}: mixer + Mlp kernels {ms_h * 1e3:7.1f} us", flush=True)
        ms_f = timeit(fused, reps)
        ms_c = float("nan") if only_fused else timeit(chain, reps)
        flops = 2.0 * G * rows * C * 4 * C + 4.0 * G * rows * N * C
        print(f"C{C} N{N} ({H}x{W}) {'local ' if local else 'global'}: fused {ms_f * 1e3:7.1f} us  chain {ms_c * 1e3:7.1f} us   "
              f"{flops / ms_f * 1e-9:6.1f} TF (dense count)", flush=True)


if __name__ == "__main__":
    main()
