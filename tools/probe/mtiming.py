"""timing probe of the fused SVTR Mlp kernels (bash tools/build_probe.sh MRN_MPROBE_TIMING svtr_mlp.hip): where a wave's cycles go"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MRN_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmrn_MRN_MPROBE_TIMING.so")
from mrn_amd import ops
lib = ctypes.CDLL(os.environ["MRN_LIB_PATH"])
G = 6
dev = torch.device("cuda")
for C, toks in ((64, 256 * 512), (128, 256 * 256), (256, 256 * 128)):
    Ch = 4 * C
    rows = G * toks
    x_hl = ops.split_hl32(torch.randn(rows, C, device=dev))
    w1 = [torch.randn(Ch, C, device=dev) * C ** -0.5 for _ in range(G)]
    w2 = [torch.randn(C, Ch, device=dev) * Ch ** -0.5 for _ in range(G)]
    b1, b2 = torch.randn(G, Ch, device=dev) * 0.2, torch.randn(G, C, device=dev) * 0.2
    w1_hl, s1 = ops.pack_weights_hl32([w.view(Ch, 1, 1, C).contiguous() for w in w1])
    w2_hl, s2 = ops.pack_weights_hl32([w.index_select(1, ops.mlp_hidden_permutation(Ch, dev)).contiguous().view(C, 1, 1, Ch) for w in w2])
    buf = (ctypes.c_ulonglong * 8)()
    for _ in range(2):
        ops.svtr_mlp_fused(x_hl, rows, toks, G, C, w1_hl, s1, b1, w2_hl, s2, b2)
    torch.cuda.synchronize()
    lib.mrn_mlp_dbg_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.svtr_mlp_fused(x_hl, rows, toks, G, C, w1_hl, s1, b1, w2_hl, s2, b2)
    e1.record(); torch.cuda.synchronize()
    lib.mrn_mlp_dbg_read(buf, 1)
    n = max(buf[5], 1)
    nh = Ch // 32
    mf1, mf2 = (C // 16) * 3 * 32, (C // 32) * 2 * 3 * 32
    print(f"C={C}: {e0.elapsed_time(e1):.3f} ms, {n} waves; per wave and hidden block: wait+barrier {buf[0] / n / nh:.0f}, fc1 {buf[1] / n / nh:.0f} (MFMA {mf1}), "
          f"GELU {buf[2] / n / nh:.0f}, fc2 {buf[3] / n / nh:.0f} (MFMA {mf2}); whole wave {buf[4] / n:.0f} cycles")
