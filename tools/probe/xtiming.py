"""timing probe of the fused SVTR mixer kernel (bash tools/build_probe.sh MRN_XPROBE_TIMING svtr_mixer.hip): where a wave's cycles go"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MRN_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmrn_MRN_XPROBE_TIMING.so")
from mrn_amd import ops
from mrn_amd.modules.svtr import local_attention_mask
lib = ctypes.CDLL(os.environ["MRN_LIB_PATH"])
G, B = 6, 256
dev = torch.device("cuda")
for C, H, W, local in ((128, 4, 64, False), (128, 4, 64, True), (64, 8, 64, True)):
    N = H * W
    x = torch.randn(G * B, N, C, device=dev)
    pend = torch.randn(G * B, N, C, device=dev)
    ones = torch.ones(G * B, device=dev)
    g1, b1 = torch.ones(G, C, device=dev), torch.zeros(G, C, device=dev)
    wqkv = [torch.randn(3 * C, C, device=dev) * C ** -0.5 for _ in range(G)]
    wproj = [torch.randn(C, C, device=dev) * C ** -0.5 for _ in range(G)]
    bqkv, bproj = torch.randn(G, 3 * C, device=dev) * 0.1, torch.randn(G, C, device=dev) * 0.1
    mask = local_attention_mask(H, W, 7, 11).to(dev) if local else None
    wq, sq = ops.pack_weights_hl32([w.view(3 * C, 1, 1, C).contiguous() for w in wqkv])
    perm = ops.mlp_hidden_permutation(C, dev)
    wp, sp = ops.pack_weights_hl32([w.index_select(1, perm).contiguous().view(C, 1, 1, C) for w in wproj])
    run = lambda: ops.svtr_mixer_fused(x, pend, ones, g1, b1, 1e-6, wq, sq, bqkv, mask, 32 ** -0.5, wp, sp, bproj, ones, g1, b1, 1e-6, B,
                                       hw=(H, W) if local else None)
    buf = (ctypes.c_ulonglong * 8)()
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    lib.mrn_mixer_dbg_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record(); torch.cuda.synchronize()
    lib.mrn_mixer_dbg_read(buf, 1)
    n = max(buf[6], 1)
    names = ("prologue (x, pending, LayerNorm1)", "K / V projections", "Q projection", "key-tile loop", "proj", "epilogue")
    tot = buf[7] / n
    print(f"C={C} N={N} {'local' if local else 'global'}: {e0.elapsed_time(e1):.3f} ms, {n} waves, {tot:.0f} cycles per wave: "
          + ", ".join(f"{nm} {buf[i] / n:.0f} ({100.0 * buf[i] / n / tot:.0f} %)" for i, nm in enumerate(names)))
