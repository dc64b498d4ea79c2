"""Isolated timing of mrn_svtr_attention_f32 at the three SVTR-tiny stage shapes (B = 256): us per launch and the
fraction of the fp32-MFMA peak (v_mfma_f32_32x32x2_f32: 256 flop/cycle/CU x 256 CUs x 2.4 GHz = 157 TFLOP/s).
usage (GPU box): python tools/bench_attention.py [B]"""
import sys
import torch
sys.path.insert(0, ".")
from mrn_amd import ops
from mrn_amd.modules.svtr import local_attention_mask

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
for N, heads, H in ((512, 2, 8), (256, 4, 4), (128, 8, 2)):
    C = heads * 32
    qkv = torch.randn(B, N, 3 * C, device=dev)
    for masked in (True, False):
      for x3 in (False, True):
        mask = local_attention_mask(H, 64, 7, 11).to(dev) if masked else None
        for _ in range(3):
            ops.svtr_attention(qkv, heads, 32 ** -0.5, mask, x3=x3)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            ops.svtr_attention(qkv, heads, 32 ** -0.5, mask, x3=x3)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / 20
        flops = 4.0 * B * heads * N * N * 32
        kind = "fp16x3" if x3 else "fp32  "
        print(f"N={N} heads={heads} masked={masked} {kind}: {us:8.1f} us  {flops / us * 1e-6:6.1f} TFLOP/s algorithmic"
              + ("" if x3 else f"  ({flops / us * 1e-6 / 157.3:.1%} of fp32 MFMA peak)"))
