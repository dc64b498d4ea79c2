#!/usr/bin/env python3
"""isolated timing of the training-path BatchNorm passes (forward apply with mask / range, backward reduce + apply) on TRBA loop-A shapes"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrn_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B, H, W, C in [(256, 4, 65, 512), (256, 8, 65, 256), (256, 16, 128, 128), (256, 32, 256, 64)]:
    y = torch.randn(B, H, W, C, device="cuda")
    dz = torch.randn_like(y)
    res = torch.randn_like(y)
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    mean, invstd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    scale, shift = gamma.clone(), beta.clone()
    z = torch.empty_like(y)
    zmask = torch.empty(z.numel() // 4, device="cuda", dtype=torch.uint8)
    n = y.numel()
    t_f = timeit(lambda: ops.scale_shift_act(y, scale, shift, relu=True, residual=None, out=z, pos_mask=zmask, range_target=ops.TRAIN_OPERAND_PEAK))
    t_fr = timeit(lambda: ops.scale_shift_act(y, scale, shift, relu=True, residual=res, out=z, pos_mask=zmask, range_target=ops.TRAIN_OPERAND_PEAK))
    t_b = timeit(lambda: ops.bn_bwd(dz, None, y, mean, invstd, gamma, True, want_dres=False, range_target=ops.TRAIN_OPERAND_PEAK, zmask=zmask))
    t_br = timeit(lambda: ops.bn_bwd(dz, None, y, mean, invstd, gamma, True, want_dres=True, range_target=ops.TRAIN_OPERAND_PEAK, zmask=zmask))
    sx = ops.pow2_scale(z, ops.TRAIN_OPERAND_PEAK)
    t_w = timeit(lambda: ops.bn_apply_wino_grouped(z.view(1, B, H, W, C), None, None, 4, relu=False, prescale=sx)) if C % 32 == 0 and H % 4 == 0 else float("nan")
    print(f"[{B},{H},{W},{C}] {n * 4 / 1e6:6.0f} MB  fwd apply {t_f:6.1f} us ({n * 8.25 / t_f / 1e6:.2f} TB/s)  +res {t_fr:6.1f} ({n * 12.25 / t_fr / 1e6:.2f})"
          f"  bwd reduce+finalize+apply {t_b:6.1f} us ({n * 20.5 / t_b / 1e6:.2f} TB/s)  +dres {t_br:6.1f} ({n * 24.5 / t_br / 1e6:.2f})  wino operand {t_w:6.1f} us ({n * 10 / t_w / 1e6:.2f})")
