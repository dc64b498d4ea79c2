import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mrn_amd import ops
torch.manual_seed(0)
G, imgs, N, C, Ch = 1, 1, 128, 256, 1024
rows = G * imgs * N
dev = torch.device("cuda")
ctx = torch.randn(rows, C, device=dev); x = torch.randn(rows, C, device=dev)
wp = torch.randn(C, C, device=dev) / 16; w1 = torch.randn(Ch, C, device=dev) / 16; w2 = torch.randn(C, Ch, device=dev) / 32
bp = torch.zeros(1, C, device=dev); b1 = torch.zeros(1, Ch, device=dev); b2 = torch.zeros(1, C, device=dev)
gam = torch.ones(1, C, device=dev); bet = torch.zeros(1, C, device=dev)
wp_hl, sp = ops.pack_weights_hl32([wp.view(C, 1, 1, C).contiguous()])
w1_hl, s1 = ops.pack_weights_hl32([w1.index_select(1, ops.mlp_hidden_permutation(C, dev)).contiguous().view(Ch, 1, 1, C)])
w2_hl, s2 = ops.pack_weights_hl32([w2.index_select(1, ops.mlp_hidden_permutation(Ch, dev)).contiguous().view(C, 1, 1, Ch)])
xr = x.clone()
br = ops.svtr_tail_fused(ops.split_hl32(ctx), xr, rows, rows, G, C, wp_hl, sp, bp, None, N, gam, bet, 1e-6, w1_hl, s1, b1, w2_hl, s2, b2)
ref = x.double() + ctx.double() @ wp.double().t()
d = (xr.double() - ref).abs()
print("x_res err max", d.max().item(), "by channel block:", d.view(rows, 8, 32).amax(dim=(0, 2)).cpu().numpy().round(4))
print("by token block:", d.view(4, 32, C).amax(dim=(1, 2)).cpu().numpy().round(4))
proj = (xr - x)            # what the kernel added
pref = (ctx.double() @ wp.double().t()).float()
# is the kernel's proj a permutation of channels?
print("proj[0,:8]", proj[0, :8].cpu().numpy().round(3), "ref", pref[0, :8].cpu().numpy().round(3))
y = torch.nn.functional.layer_norm(ref, (C,), eps=1e-6)
bref = torch.nn.functional.gelu(y @ w1.double().t()) @ w2.double().t()
print("branch err", (br.double() - bref).abs().max().item())
