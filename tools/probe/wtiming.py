"""timing probe of the row-block Winograd kernel (tools/build_probe.sh MRN_WPROBE_TIMING conv_wino.hip): where a wave's cycles go"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MRN_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmrn_MRN_WPROBE_TIMING.so")
from mrn_amd import ops
from mrn_amd import _lib
lib = ctypes.CDLL(os.environ["MRN_LIB_PATH"])
G, B, H, W, C = int(os.environ.get("G", "6")), 256, 4, 65, 512
zero = os.environ.get("ZERO")
torch.manual_seed(1)
ypre = torch.randn(G, B, H, W, C, device="cuda")
ws = [(torch.rand(C, 3, 3, C, device="cuda") * 2 - 1) * 0.05 for _ in range(G)]
if zero:
    ypre = ypre * 0; ws = [w * 0 for w in ws]
u_hl, u_scale = ops.pack_weights_wino(ws, 4)
_, _, v = ops.bn_apply_wino_grouped(ypre, torch.ones(G, C, device="cuda"), torch.zeros(G, C, device="cuda"), 4, relu=True)
buf = (ctypes.c_ulonglong * 8)()
for rep in range(3):
    ops.conv2d_x3_wino(v, G, False, B, H, W, C, u_hl, u_scale, C, 4, want_stats=True)
torch.cuda.synchronize()
lib.mrn_wino_dbg_read(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.conv2d_x3_wino(v, G, False, B, H, W, C, u_hl, u_scale, C, 4, want_stats=True)
e1.record(); torch.cuda.synchronize()
lib.mrn_wino_dbg_read(buf, 1)
n = buf[6]
names = ["main loop", "boundary waitcnt", "barrier", "folds", "epilogue", "prologue"]
print(f"G{G} zero={zero}: {e0.elapsed_time(e1):.3f} ms, {n} waves")
for i, nm in enumerate(names):
    print(f"  {nm:18s} {buf[i] / n:12.0f} cycles per wave-tile")
steps = 96
print(f"  per step: main {buf[0] / n / steps:.0f}, waitcnt {buf[1] / n / steps:.0f}, barrier {buf[2] / n / steps:.0f}, fold/16 steps {buf[3] / n / 6:.0f} per fold; ideal MFMA 1920")
