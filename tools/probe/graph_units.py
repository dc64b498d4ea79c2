"""which autograd unit breaks HIP graph capture: each unit's forward + backward captured on its own (run one unit per process)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrn_amd import ops, functional as Fn  # noqa

unit = sys.argv[1]
torch.cuda.set_device(0)
dev = "cuda"
torch.manual_seed(0)
B, N, C = 64, 128, 128
x = torch.randn(B, N, C, device=dev, requires_grad=True)
w = torch.nn.Parameter(torch.randn(256, C, device=dev) * 0.05)
b = torch.nn.Parameter(torch.zeros(256, device=dev))
g1 = torch.nn.Parameter(torch.ones(C, device=dev)); b1 = torch.nn.Parameter(torch.zeros(C, device=dev))


def run():
    if unit == "linear":
        return Fn.TrainLinearFn.apply(x, w, b).sum()
    if unit == "linear_plain":
        return Fn.LinearFn.apply(x, w, b).sum()
    if unit == "ln":
        return Fn.LayerNormFn.apply(x, g1, b1, 1e-6).sum()
    if unit == "gelu":
        return Fn.GeluFn.apply(x).sum()
    if unit == "attn":
        qkv = torch.randn(B, N, 3 * C, device=dev, requires_grad=True)
        return Fn.SvtrAttentionFn.apply(qkv, None, C // 32, 32 ** -0.5).sum()
    if unit == "resid":
        d = torch.ones(B, device=dev)
        return Fn.ResidualScaleFn.apply(x, x * 2, d, N).sum()
    if unit == "torchonly":
        return (x * 2).sum()
    raise SystemExit("unknown unit")


s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        l = run(); l.backward()
        del l
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    l = run()
    if os.environ.get("FWD_ONLY") != "1":
        l.backward()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print(unit, "ok", float(l))
