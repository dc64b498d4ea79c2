import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mrn_amd import ops
torch.manual_seed(0)
G,B,H,W,Cin,Cout=1,1,8,32,32,64
x=torch.randn(G,B,H,W,Cin,device="cuda"); w=[torch.randn(Cout,3,3,Cin,device="cuda")*0.1]
x_hl=ops.split_hl32(x); w_hl,w_scale=ops.pack_weights_hl32(w)
y,_=ops.conv3x3_patch_x3(x_hl,G,False,B,H,W,Cin,w_hl,w_scale,Cout)
gam=torch.randn(Cout,device="cuda"); ptrs=torch.tensor([gam.data_ptr()],dtype=torch.int64,device="cuda")
yp,_=ops.conv3x3_patch_x3(x_hl,G,False,B,H,W,Cin,w_hl,w_scale,Cout,pool=True,gamma_ptrs=ptrs)
yy=y[0,0].view(H//2,2,W//2,2,Cout)
mx=yy.amax(dim=(1,3)); mn=yy.amin(dim=(1,3))
ref=torch.where(gam>=0,mx,mn)
d=(yp[0,0]!=ref)
print("mismatch frac",d.float().mean().item())
print("by channel:",d.float().mean(dim=(0,1)).cpu().numpy().round(2))
print("by oy:",d.float().mean(dim=(1,2)).cpu().numpy().round(2))
print("by ox:",d.float().mean(dim=(0,2)).cpu().numpy().round(2))
print("gamma<0:",(gam<0).int().cpu().numpy())
# is got equal to max or min anywhere
print("eq max frac", (yp[0,0]==mx).float().mean().item(), "eq min", (yp[0,0]==mn).float().mean().item())
