#!/usr/bin/env python3
"""Per-kernel roofline table of the HBM-bound and latency-bound kernels of the path at BASELINE sizes (B = 256, six experts,
cumulative class counts of SURVEY.md section 8d) -- the companion of bench.py's `roofline` object, which covers the MFMA-bound
dominant kernel.  Each line: algorithmic bytes (every input and output element once), average launch duration over `reps` launches
(HIP events on the launch stream), achieved GB/s and the fraction of the 8 TB/s HBM3E peak.

    python tools/bench_kernels.py [reps] > profiles/rNN_kernel_rooflines.md
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops  # noqa: E402

HBM_PEAK = 8.0e12
B = 256
CLASSES_CTC = [2090, 2310, 4038, 5198, 5271, 5373]      # SURVEY 8d cumulative class counts + 4 CTC tokens
CLASSES_ATTN = [2091, 2311, 4039, 5199, 5272, 5374]     # ... + 5 Attn tokens
dev = "cuda"


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ROWS = []


def line(name, site, nbytes, ms, note=""):
    gbs = nbytes / ms / 1e6
    ROWS.append((name, site, nbytes, ms, gbs, note))
    print(f"| `{name}` | {site} | {nbytes / 1e6:9.1f} | {ms * 1e3:8.1f} | {gbs:7.0f} | {gbs * 1e9 / HBM_PEAK:5.2f} | {note} |", flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.manual_seed(0)
    print(f"<!-- python tools/bench_kernels.py {reps} on {torch.cuda.get_device_name()} -->")
    print("| kernel (entry point) | workload | algorithmic MB | avg us | GB/s | of 8 TB/s | note |")
    print("|---|---|---:|---:|---:|---:|---|")

    # ---- router fan-in (K13): six experts' logits -> fused logits; ones-padding synthesised, never read ------------------------
    T = 26
    logits = [ops.padded_rows(B, T, c, dev).copy_(torch.randn(B, T, c, device=dev)) for c in CLASSES_ATTN]
    w = torch.softmax(torch.randn(B, 6, device=dev), 1)
    nb = 4.0 * B * T * (sum(CLASSES_ATTN) + CLASSES_ATTN[-1])
    line("mrn_fanin_fwd_f32", "TRBA x 6, T = 26", nb, timeit(lambda: ops.fanin_fwd(logits, w), reps))
    dout = ops.padded_rows(B, T, CLASSES_ATTN[-1], dev).copy_(torch.randn(B, T, CLASSES_ATTN[-1], device=dev))
    line("mrn_fanin_bwd_f32", "TRBA x 6, T = 26", nb, timeit(lambda: ops.fanin_bwd(logits, dout), reps),
         "reads the logits and dout once, writes dw [B, 6]")
    T = 63
    logits_c = [ops.padded_rows(B, T, c, dev).copy_(torch.randn(B, T, c, device=dev)) for c in CLASSES_CTC[:3]]
    w3 = torch.softmax(torch.randn(B, 3, device=dev), 1)
    nb = 4.0 * B * T * (sum(CLASSES_CTC[:3]) + CLASSES_CTC[2])
    line("mrn_fanin_fwd_f32", "CRNN x 3, T = 63", nb, timeit(lambda: ops.fanin_fwd(logits_c, w3), reps))
    idx = torch.randint(0, 6, (B,), device=dev)
    nb = 4.0 * B * 26 * 2 * CLASSES_ATTN[-1]
    line("mrn_select_expert_f32", "eval routing, TRBA x 6", nb, timeit(lambda: ops.select_expert(logits, idx), reps),
         "reads only the selected expert's rows")

    # ---- losses ---------------------------------------------------------------------------------------------------------------
    C = CLASSES_CTC[-1]
    lg = ops.padded_rows(B, 63, C, dev).copy_(torch.randn(B, 63, C, device=dev))
    tgt = torch.randint(1, C, (B, 25), device=dev)
    tl = torch.randint(1, 26, (B,), device=dev, dtype=torch.int32)
    loss, ctx = ops.ctc_loss_fwd(lg, tgt, tl)
    up = torch.ones(1, device=dev)
    line("mrn_ctc_loss_fwd_f32", f"B x 63 x {C}", 4.0 * B * 63 * C, timeit(lambda: ops.ctc_loss_fwd(lg, tgt, tl), reps),
         "log-softmax statistics + alpha/beta lattice: logits read once")
    line("mrn_ctc_loss_bwd_f32", f"B x 63 x {C}", 8.0 * B * 63 * C, timeit(lambda: ops.ctc_loss_bwd(ctx, up), reps))
    C = CLASSES_ATTN[-1]
    la = ops.padded_rows(B, 26, C, dev).copy_(torch.randn(B, 26, C, device=dev))
    ta = torch.randint(1, C, (B, 26), device=dev)
    loss, cctx = ops.ce_loss_fwd(la, ta, ignore_index=1)
    line("mrn_ce_loss_fwd_f32", f"B x 26 x {C}", 4.0 * B * 26 * C, timeit(lambda: ops.ce_loss_fwd(la, ta, ignore_index=1), reps))
    line("mrn_ce_loss_bwd_f32", f"B x 26 x {C}", 8.0 * B * 26 * C, timeit(lambda: ops.ce_loss_bwd(cctx, up, la), reps))
    xo = torch.randn(B * 26, C, device=dev)
    xn = torch.randn(B * 26, C, device=dev)
    line("mrn_kd_loss_fwd_f32", f"LwF, B x 26 x {CLASSES_ATTN[-2]} old classes", 8.0 * B * 26 * CLASSES_ATTN[-2],
         timeit(lambda: ops.kd_loss_fwd(xn, xo, 0, CLASSES_ATTN[-2], 2.0), reps))

    # ---- optimiser over config 5's 57.8 M trainable parameters ------------------------------------------------------------------
    n = 57_800_000
    p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    line("mrn_grad_norm_clip_f32", "57.8 M parameters", 4.0 * n, timeit(lambda: ops.grad_norm_clip(g, 5.0), reps))
    nc = ops.grad_norm_clip(g, 5.0)
    line("mrn_adam_step_f32", "57.8 M parameters", 28.0 * n, timeit(lambda: ops.adam_step(p, g, m, v, nc, 1e-4, 3), reps),
         "reads p, g, m, v; writes p, m, v")
    line("mrn_sgd_step_f32", "57.8 M parameters", 20.0 * n, timeit(lambda: ops.sgd_step(p, g, m, nc, 1e-4, 0.9, 5e-4), reps))
    line("mrn_adadelta_step_f32", "57.8 M parameters", 28.0 * n, timeit(lambda: ops.adadelta_step(p, g, m, v, nc, 1.0), reps))
    fisher = torch.rand(n, device=dev)
    mean = torch.randn(n, device=dev) * 0.01
    line("mrn_ewc_penalty_fwd_f32", "57.8 M parameters", 12.0 * n, timeit(lambda: ops.ewc_penalty(fisher, p, mean), reps))
    line("mrn_ewc_penalty_bwd_f32", "57.8 M parameters", 20.0 * n,
         timeit(lambda: ops.ewc_penalty_grad_(g, fisher, p, mean, 0.5), reps), "reads g, F, p, p*; writes g")
    del p, g, m, v, fisher, mean

    # ---- per-layer passes of the lock-step experts (ResNet stage shapes of TRBA, G = 6) -------------------------------------------
    for (H, W, Cc) in ((16, 128, 128), (8, 64, 256), (4, 65, 512)):
        y = torch.randn(6, B, H, W, Cc, device=dev)
        sc, sh = torch.rand(6, Cc, device=dev) + 0.5, torch.randn(6, Cc, device=dev)
        res_hl = ops.split_hl32(torch.randn(6, B, H, W, Cc, device=dev))
        line("mrn_bn_apply_grouped_f32", f"6 x B x {H}x{W}x{Cc}: BN + ReLU -> HL32", 8.0 * y.numel(),
             timeit(lambda: ops.bn_apply_grouped(y, sc, sh, relu=True, want_f32=False, want_hl=True), reps))
        line("mrn_bn_apply_grouped_f32", f"6 x B x {H}x{W}x{Cc}: BN + residual(HL32) + ReLU -> HL32", 12.0 * y.numel(),
             timeit(lambda: ops.bn_apply_grouped(y, sc, sh, relu=True, want_f32=False, want_hl=True, residual_hl=res_hl), reps))
        del res_hl
    y = torch.randn(6, B, 16, 128, 128, device=dev)
    sc, sh = torch.rand(6, 128, device=dev) + 0.5, torch.randn(6, 128, device=dev)
    big = torch.randn(6, B, 32, 256, 128, device=dev)
    line("mrn_maxpool_grouped_f32", "6 x B x 32x256x128 -> 16x128, BN + ReLU fused, HL32 out",
         4.0 * 6 * B * 128 * (32 * 256 + 16 * 128),
         timeit(lambda: ops.maxpool_grouped(big, (2, 2), (2, 2), (0, 0), sc, sh, True, False, True), reps))
    del big
    x = torch.randn(6, B, 8, 64, 256, device=dev)
    line("mrn_split_hl32_f32", "6 x B x 8x64x256 fp32 -> HL32", 8.0 * x.numel(), timeit(lambda: ops.split_hl32(x), reps))
    del x, y

    # ---- TPS sampling, LayerNorm, argmax -------------------------------------------------------------------------------------------
    xl = torch.randn(B * 65 * 6, 256, device=dev)
    gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    line("mrn_layernorm_fwd_f32", "router LN(256) over B x 390 tokens", 8.0 * xl.numel(),
         timeit(lambda: ops.layernorm_fwd(xl, gam, bet), reps))
    lgt = torch.randn(B, 26, CLASSES_ATTN[-1], device=dev)
    line("mrn_argmax_prob_f32", f"validation: B x 26 x {CLASSES_ATTN[-1]} -> index + softmax max", 4.0 * lgt.numel(),
         timeit(lambda: ops.argmax_prob_lastdim(lgt), reps))


if __name__ == "__main__":
    main()
