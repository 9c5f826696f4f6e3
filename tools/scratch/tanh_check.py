import sys; sys.path.insert(0,'/root/repo')
import torch
from rna_gan_amd.ops_hip import HipOps
from rna_gan_amd.engine import ConvW
dev=torch.device('cuda:0'); ops=HipOps(torch.bfloat16,'cuda:0')
w=torch.randn(64,3,4,4,device=dev)*0.3; b3=torch.randn(3,device=dev)*0.1; cw=ConvW(w)
a=(torch.randn(4,128,128,64,device=dev)*2).to(torch.bfloat16)
y1=ops.last_up(a,cw,b3,True); y0=ops.last_up(a,cw,b3,False)
ref=torch.tanh(y0.double())
print('max abs err vs tanh(fp64) of the pre-activation:', float((y1.double()-ref).abs().max()), 'range', float(y0.min()), float(y0.max()))
