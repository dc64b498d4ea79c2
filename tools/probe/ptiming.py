"""timing probe of the patch-resident convolution (bash tools/build_probe.sh MRN_PPROBE_TIMING conv_patch.hip): where a wave's cycles go"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MRN_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmrn_MRN_PPROBE_TIMING.so")
from mrn_amd import ops
lib = ctypes.CDLL(os.environ["MRN_LIB_PATH"])
G, B = 6, 256
for (H, W, Cin, Cout) in ((32, 256, 32, 64), (16, 128, 64, 128)):
    x = torch.relu(torch.randn(G, B, H, W, Cin, device="cuda"))
    ws = [torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.05 for _ in range(G)]
    x_hl = ops.split_hl32(x)
    w_hl, w_scale = ops.pack_weights_hl32(ws)
    del x
    for pool in (False, True):
        y = torch.empty(G, B, H // (2 if pool else 1), W // (2 if pool else 1), Cout, device="cuda")
        buf = (ctypes.c_ulonglong * 8)()
        for _ in range(2):
            ops.conv3x3_patch_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, want_stats=True, pool=pool, out=y)
        torch.cuda.synchronize()
        lib.mrn_patch_dbg_read(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv3x3_patch_x3(x_hl, G, False, B, H, W, Cin, w_hl, w_scale, Cout, want_stats=True, pool=pool, out=y)
        e1.record(); torch.cuda.synchronize()
        lib.mrn_patch_dbg_read(buf, 1)
        n = max(buf[5], 1)
        print(f"{Cin}->{Cout} pool={pool}: {e0.elapsed_time(e1):.3f} ms, {n} wave-tiles; per wave-tile cycles: top barrier/fetch {buf[0] / n:.0f}, "
              f"main loops {buf[1] / n:.0f} (MFMA ideal {216 * (Cin // 32) * 32}), patch wait {buf[2] / n:.0f}, tile behind barrier {buf[3] / n:.0f}, total {buf[4] / n:.0f}")
