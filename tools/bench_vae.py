"""Time one betaVAE training iteration (forward, loss, backward, fused Adam) at the reference's full size
(19198 genes, [6000, 4000, 2048] / [4000, 6000], batch 64 = src/betaVAE_training.py defaults) on the HIP path."""
import os, sys, time
import torch
sys.path.insert(0, ".")
import rna_gan_amd as P
from rna_gan_amd import vae_train as VT
from oracle import ref_cpu as R


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dims = (19198, 2048, [6000, 4000, 2048], [4000, 6000])
    m = P.betaVAE(*dims, beta=2.0)
    R.seeded_fill_(m, 51)
    m = m.set_precision(prec).cuda().train()
    opt = P.Adam(m.parameters(), lr=3e-3, weight_decay=1e-4).bind(m, fuse_linear_wgrad=os.environ.get("VAE_FUSE", "1") != "0")
    x = torch.tanh(torch.randn(N, dims[0], device="cuda"))
    nparam = sum(p.numel() for p in m.parameters())

    def step():
        opt.zero_grad(set_to_none=True)
        out, mu, lv = m(x)
        losses = VT.betaVAEloss(x, out, mu, lv, m.beta, training=True)
        losses["total_loss"].backward()
        opt.step()
        return losses

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        losses = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # floor: Adam 30 B/param + fp32 weight gradient written once (4 B) + weights read 3x as bf16 operands
    floor_bytes = nparam * (28 + 4 + 3 * 2)
    print(f"{prec} batch {N}: {dt * 1e3:.2f} ms/iteration, {N / dt:.0f} samples/s, {nparam / 1e6:.1f} M parameters, "
          f"HBM floor {floor_bytes / 1e9:.1f} GB -> {floor_bytes / dt / 1e12:.2f} TB/s of it; loss {float(losses['total_loss'].detach()):.4f}")


if __name__ == "__main__":
    main()
