"""Generator-only synthesis (BASELINE configs[4]: 4096 samples, chunks of 512, eval-mode BatchNorm): fp8 with the MX-format MFMA
(unit block scales) vs the non-scaled fp8 MFMA vs bf16, interleaved rounds in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import rna_gan_amd as P
from rna_gan_amd import synth as R, gan_utils as GU, _abi
lib = _abi.load()
dev = torch.device("cuda:0")
G = P.DCGANGenerator(2048, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
R.seeded_fill_(G, 3)
G = G.set_precision("bf16").to(dev).eval()
n = 4096
noise = torch.randn(n, 2048, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
out = torch.empty((n, 3, 256, 256), device=dev)
def run(fp8, mx):
    _abi.check(lib.rg_set_option(b"fp8_mx", mx), "opt")
    GU.synthesize(G, noise[:1024], chunk=512, fp8=fp8, out=out[:1024])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    GU.synthesize(G, noise, chunk=512, fp8=fp8, out=out)
    torch.cuda.synchronize(); return time.perf_counter() - t0
ref = None
for r in range(3):
    for tag, fp8, mx in (("fp8 mx", True, 1), ("fp8 plain", True, 0), ("bf16", False, 1)):
        dt = run(fp8, mx)
        print("%-10s %8.0f imgs/s  %6.1f TFLOP/s" % (tag, n / dt, 5.604e9 * n / dt / 1e12), flush=True)
        if r == 0 and fp8:
            s = out[::97].clone()
            if ref is None: ref = s
            else: print("   max |mx - plain| on sampled images: %.3e" % float((s - ref).abs().max()))
