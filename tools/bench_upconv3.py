"""Time the resize-convolution block forward (MFMA vs functor path) at the up-generator's batch-64 shapes."""
import sys, time
import torch
sys.path.insert(0, ".")
from rna_gan_amd import _abi
from rna_gan_amd.ops_hip import HipOps
from rna_gan_amd.engine import ConvW

def main():
    N = 64
    for (H, Cin, Cout) in [(4, 1024, 512), (8, 512, 256), (16, 256, 128), (32, 128, 64), (64, 64, 64)]:
        x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * (2.0 / (9 * Cin)) ** 0.5
        b = torch.randn(Cout, device="cuda") * 0.1
        cw = ConvW(w, b, torch.zeros_like(w))
        gy = torch.randn(N, 2 * H, 2 * H, Cout, device="cuda").bfloat16()
        fl = 2.0 * N * 4 * H * H * Cout * 9 * Cin
        calls = {"fwd": lambda o: o.upconv3(x, cw, b), "bwd_data": lambda o: o.upconv3_bwd_data(gy, cw),
                 "wgrad": lambda o: (o.upconv3_wgrad(gy, x, cw, False), cw.dw.clone())[1]}
        for what, fn in calls.items():
            res = {}
            for name, algo in (("mfma", _abi.ALGO_AUTO), ("generic", _abi.ALGO_GENERIC)):
                ops = HipOps(torch.bfloat16, "cuda:0", algo=algo)
                y = fn(ops)
                torch.cuda.synchronize()
                reps = 20 if name == "mfma" else 2
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn(ops)
                torch.cuda.synchronize()
                res[name] = ((time.perf_counter() - t0) / reps * 1e6, y.float())
            err = (res["mfma"][1] - res["generic"][1]).abs().max().item()
            ref = res["generic"][1].abs().max().item()
            print(f"H={H:3d} Cin={Cin:4d} Cout={Cout:4d} {what:9s} mfma {res['mfma'][0]:8.1f} us ({fl / res['mfma'][0] / 1e6:6.1f} TF/s)"
                  f"  generic {res['generic'][0]:9.1f} us   max|diff| {err:.4f} (max|ref| {ref:.2f})", flush=True)

if __name__ == "__main__":
    main()
