"""debug: directional derivative of TRBA loop A at B = 256 per parameter group and eps"""
import contextlib, io, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mrn_amd import functional as Fn, ops
from mrn_amd.modules.model import Model
from mrn_amd.tools import weights as W
from mrn_amd.optim import FlatAdam
from tests.test_model_gpu import make_opt

B = int(os.environ.get("DD_B", "256"))
opt = make_opt("trba")
C = 2091
with contextlib.redirect_stdout(io.StringIO()):
    net = Model(opt); net.update_fc(opt.hidden_size, C); net.build_prediction(opt, C)
W.fill_state_dict(net.state_dict(), seed=41)
net = net.cuda().train()
image = torch.from_numpy(W.uniform("fullA", (B, 4, 32, 256), -1.0, 1.0, 5) if os.environ.get("DD_NOISE", "1") == "1" else W.smooth_image("fullA", (B, 4, 32, 256), 5)).cuda()
labels = torch.from_numpy(W.randint("fullA_lab", (B, 25), 4, C, 5)).cuda()
lengths = torch.from_numpy(W.randint("fullA_len", (B,), 1, 26, 5)).int().cuda()
ln = lengths.long().clamp(max=25)
pos = torch.arange(27, device="cuda")[None, :]
index = torch.cat([torch.full((B, 1), 2, device="cuda"), labels.long().clamp(min=5), torch.ones(B, 1, dtype=torch.long, device="cuda")], 1)
index = torch.where(pos == ln[:, None] + 1, torch.full_like(index, 3), index)
index = torch.where(pos > ln[:, None] + 1, torch.ones_like(index), index)
bn_state = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}

def loss_at():
    net.load_state_dict(bn_state, strict=False)
    return Fn.cross_entropy(net(image, index[:, :-1], True)["predict"], index[:, 1:], 1)

named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
params = [p for _, p in named]
fo = FlatAdam(params, lr=1e-3)
for side, wino in ((True, True), (False, True), (False, False)):
    ops.WGRAD_SIDE_STREAM = side
    ops.TRAIN_WINO = wino
    ops.TRAIN_OPERAND_PEAK = 16384.0 / 16 if wino else 16384.0
    fo.zero_grad()
    loss = loss_at()
    with ops.direct_gradients():
        loss.backward()
    torch.cuda.synchronize()
    grads = [p.grad.detach().clone() for p in params]
    print("side", side, "wino", wino, "loss", float(loss), "gnorm", float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads))))
    if side and wino:
        g0 = grads
    else:
        num = float(torch.sqrt(sum(((a.double() - b.double()) ** 2).sum() for a, b in zip(grads, g0))))
        print("   diff to first", num)
        rel = sorted(((float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30)), float(b.double().norm()), n) for (n, _), a, b in zip(named, grads, g0)), reverse=True)[:8]
        for r, nb, n in rel:
            print(f"      rel diff {r:.3e} |g| {nb:.3e} {n}")
groups = {}
for (n, p), g in zip(named, grads):
    groups.setdefault(n.split(".")[1] if n.startswith("model.") else n.split(".")[0], []).append((n, p, g))
for gname, items in groups.items():
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for _, _, g in items)))
    out = [f"{gname}: |g| {gn:.4e}"]
    for eps in (2e-3, 5e-4, 1e-4):
        vals = []
        with torch.no_grad():
            for sign in (+1.0, -1.0):
                for _, p, g in items:
                    p.add_(g, alpha=sign * eps / gn)
                torch.autograd.graph.increment_version(params)
                vals.append(float(loss_at()))
                for _, p, g in items:
                    p.add_(g, alpha=-sign * eps / gn)
            torch.autograd.graph.increment_version(params)
        out.append(f"eps {eps:g}: fd {(vals[0] - vals[1]) / (2 * eps):.4e}")
    print("  ".join(out), flush=True)
# largest tensors by gradient norm
big = sorted(((float(g.double().norm()), n) for (n, p), g in zip(named, grads)), reverse=True)[:12]
for v, n in big:
    print(f"{v:.4e} {n}")
