"""Generator-only conditioned inference throughput (SURVEY 8f f1 / BASELINE configs[4] in bf16):
generate_images(trainer, gene_exp, sample_size) -- betaVAE-conditioned noise, generator on chunks of 10 (the reference's
chunking, src/gan_utils.py:217-221), un-normalise + NHWC export -- and the raw generator at larger chunk sizes."""
import sys, time
import torch
import torch.nn as nn
sys.path.insert(0, ".")
import rna_gan_amd as P
from rna_gan_amd import gan_utils as GU
from rna_gan_amd import synth as R
from rna_gan_amd import engine as E


class _T:
    pass


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda:0")
    G = P.DCGANGenerator(2048, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    R.seeded_fill_(G, 3)
    G = G.set_precision("bf16").to(dev).train()
    bv = P.betaVAE(19198, 2048, [6000, 4000, 2048], [4000, 6000], beta=0.005)
    R.seeded_fill_(bv, 4)
    bv = bv.set_precision("bf16").to(dev).eval()
    tr = _T(); tr.generator = G; tr.device = dev
    rna = R.synthetic_rna(n, 19198, seed=5, distinct=16)
    for _ in range(2):
        GU.generate_images(tr, rna[:40], 40, bv)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    imgs = GU.generate_images(tr, rna, n, bv)
    dt = time.perf_counter() - t0
    print(f"generate_images (chunks of 10, train-mode BN, host copy included): {n} images in {dt * 1e3:.1f} ms = {n / dt:.0f} imgs/s", flush=True)
    noise = torch.randn(n, 2048, device=dev)
    for chunk in (10, 64, 256):
        with torch.no_grad():
            for c in torch.split(noise[:4 * chunk], chunk):
                G(c.contiguous())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in torch.split(noise, chunk):
                if c.shape[0] == chunk:
                    G(c.contiguous())
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        m = (n // chunk) * chunk
        print(f"generator forward only, chunk {chunk:4d}: {m / dt:.0f} imgs/s ({dt / (n // chunk) * 1e3:.3f} ms per chunk)", flush=True)
    assert imgs.shape == (n, 256, 256, 3)
    # eval-mode BatchNorm (BASELINE configs[4] as SURVEY 8d specifies it: running statistics, stated explicitly), output
    # resident on the device: fused epilogue (one kernel per Conv+BN+LReLU block) against the unfused conv -> bn_act pairs
    G.eval()
    ops, net = G.runtime()
    for chunk in (256, 512):
        with torch.no_grad():
            for c in torch.split(noise[:2 * chunk], chunk):
                E.gen_forward_eval_fp8(ops, net, c.contiguous())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in torch.split(noise, chunk):
                if c.shape[0] == chunk:
                    _, nfp8 = E.gen_forward_eval_fp8(ops, net, c.contiguous())
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        m = (n // chunk) * chunk
        print(f"eval-mode generator, fp8 e4m3 on {nfp8} of 7 layers, chunk {chunk:4d}: {m / dt:.0f} imgs/s "
              f"({dt / (n // chunk) * 1e3:.3f} ms per chunk, {5.604e9 * m / dt / 1e12:.0f} TFLOP/s)", flush=True)
    for fused in (False, True):
        for chunk in (64, 256, 512):
            with torch.no_grad():
                for c in torch.split(noise[:2 * chunk], chunk):
                    E.gen_forward_eval(ops, net, c.contiguous(), fused_epilogue=fused)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for c in torch.split(noise, chunk):
                    if c.shape[0] == chunk:
                        E.gen_forward_eval(ops, net, c.contiguous(), fused_epilogue=fused)
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            m = (n // chunk) * chunk
            print(f"eval-mode generator, fused epilogue {fused}, chunk {chunk:4d}: {m / dt:.0f} imgs/s "
                  f"({dt / (n // chunk) * 1e3:.3f} ms per chunk, {5.604e9 * m / dt / 1e12:.0f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
