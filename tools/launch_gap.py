"""cost of a dependent tiny-kernel launch: plain launches through the Python binding vs nodes of a replayed HIP graph"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops
x = torch.zeros(64, device="cuda")
sc = torch.empty(2, device="cuda")
ws = torch.zeros(64, device="cuda", dtype=torch.int32)
def f():
    ops.call("mrn_pow2_finalize_f32", 1.0, ops._p(sc), ops._p(ws), ops._stream())
for n in (200, 2000):
    for _ in range(50): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): f()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(n, "launches: GPU %.2f us each, host issue %.2f us each" % (e0.elapsed_time(e1) * 1e3 / n, (t1 - t0) * 1e6 / n))
# graph capture
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): f()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(200): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print("graph of 200: %.2f us per kernel" % (e0.elapsed_time(e1) * 1e3 / 1000))
