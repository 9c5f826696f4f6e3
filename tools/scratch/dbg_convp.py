import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rna_gan_amd import _abi
from rna_gan_amd.engine import ConvW
from rna_gan_amd.ops_hip import HipOps
lib = _abi.load()
ops = HipOps(torch.bfloat16, "cuda:0")
N, Ws, O, I = int(sys.argv[1]), int(sys.argv[2]), 128, 64
torch.manual_seed(0)
w = torch.randn(O, 4, 4, I, device="cuda") * 0.05
cw = ConvW(w, None, torch.zeros_like(w), None, "OHWI")
g = torch.randn(N, Ws, Ws, O, device="cuda").to(torch.bfloat16)
outs = []
for on in (0, 1):
    lib.rg_set_option(b"convp", on)
    want = len(sys.argv) > 3
    y = ops.conv_up(g, cw, want_stats=want)
    y = y[0] if want else y
    torch.cuda.synchronize()
    outs.append(y.float().clone())
d = (outs[0] - outs[1]).abs()
print("max diff", d.max().item(), "frac wrong", (d > 1e-3).float().mean().item())
bad = (d > 1e-3)
print("by channel", bad.sum(dim=(0, 1, 2))[:64].tolist())
print("by row parity", bad[:, 0::2].sum().item(), bad[:, 1::2].sum().item(), "col parity", bad[:, :, 0::2].sum().item(), bad[:, :, 1::2].sum().item())
print("by image", bad.sum(dim=(1, 2, 3)).tolist())
print("by out row", bad.sum(dim=(0, 2, 3)).tolist())
