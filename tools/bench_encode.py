"""Time the frozen betaVAE encode (19198 -> 6000 -> 4000 -> 2048 -> z_mean, batch 64: 303 MB of bf16 weights streamed per call)
and G.0 (2048 -> 2048x4x4, 134 MB)."""
import sys, time, torch
sys.path.insert(0, ".")
import rna_gan_amd as P
from rna_gan_amd.ops_hip import HipOps
from rna_gan_amd.engine import ConvW
from oracle import ref_cpu as R

bv = P.betaVAE(19198, 2048, [6000, 4000, 2048], [4000, 6000], beta=0.005)
R.seeded_fill_(bv, 4)
bv = bv.set_precision("bf16").cuda().eval()
x = R.synthetic_rna(64, 19198, seed=5, distinct=16).cuda()
ops = HipOps(torch.bfloat16, "cuda:0")
w0 = ConvW(torch.randn(2048, 2048, 4, 4, device="cuda") * 0.02, None)
z = torch.randn(64, 2048, device="cuda")

def timeit(fn, rep=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rep): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / rep * 1e6

te = timeit(lambda: bv.encode(x, mean_only=True))
tg = timeit(lambda: ops.g0_fwd(z, w0))
print(f"encode {te:7.1f} us ({303.2e6 / te / 1e6:.2f} TB/s of weights)   g0_fwd {tg:6.1f} us ({134.2e6 / tg / 1e6:.2f} TB/s)")
