#!/usr/bin/env python3
"""Training-quality A/B for the north star's "FID within 2.0 of the CPU reference" clause ("a clearly-labelled
proxy Frechet distance ... for GPU vs CPU-reference generators trained from identical seeds"; src/fid.py:98-163, 312-330).

Pretrained Inception-v3 weights cannot be obtained offline, so a real FID value does not exist here.  What CAN be measured is
whether the bf16 MFMA mode (the benchmarked one) trains generators of the same quality as the fp32 mode (the reference's own
arithmetic, pinned to the CPU oracle by the parity tests at 2e-3) -- although their trajectories diverge step by step
(LeakyReLU kinks amplify bf16-sized perturbations).

Protocol, per loss type (``wgan`` = the stock torchgan losses the CLI selects for --loss_type wgan, clamp (-0.01, 0.01);
``wganvae`` = the three betaVAE-conditioned plugins of src/wgan_loss.py):

  * data: procedurally generated tissue-like tiles (background stain, elliptical nuclei, smooth texture) whose appearance is
    driven by a hidden per-slide attribute vector; a slide's RNA row is a fixed random linear image of those attributes plus
    noise, standardised -- so the conditioning latent carries information about the tile, as in the reference's data.
    Disjoint training / held-out slides, everything seeded (numpy PCG64);
  * three runs of ``iters`` iterations at batch ``batch`` on identical data order and identical draws (weights, uniform /
    normal noise, eps) unless the seed says otherwise:   A = fp32, seed s1;   B = bf16, seed s1;   C = fp32, seed s2
    (different weights and draws: the run-to-run spread of the fp32 mode -- the yardstick);
  * evaluation: ``n_eval`` samples from each trained generator (train-mode BatchNorm on chunks, as the reference's
    generate_images leaves it; every generator sampled in fp32 precision from the SAME evaluation draws / held-out RNA rows) and
    ``n_eval`` held-out real tiles; Frechet distance (rna_gan_amd.fid.frechet_distance = src/fid.py:112-163) on
      (i)  D-trunk features: the trunk of run A's trained discriminator (fid.discriminator_features), one fixed extractor
           for every set;
      (ii) Inception-v3 pool features from the HIP Inception extractor with FIXED SEEDED RANDOM weights (architecture of
           src/fid.py:33-60; not the pretrained network, so these are not FID values);
    reported:  FD(B, A) against FD(C, A)  -- is bf16-vs-fp32 further apart than fp32-vs-fp32? --  and FD(x, held-out real)
    for x in {A, B, C}  -- did any mode train a worse generator? -- plus windowed means of the three loss curves.

Output: a JSON record (--out) and a markdown table on stdout.  tests/test_quality_ab_gpu.py runs a reduced size and asserts
FD(B, A) <= factor x FD(C, A).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# ---------------------------------------------------------------------------------------------- synthetic tissue
def make_slides(n_slides, tiles_per_slide, size, rna_features, seed, mix_seed=777):
    """(images (N, 3, S, S) float32 in [-1, 1], rna (N, F) float32, slide index (N,)).  The attribute -> RNA mixing matrix
    depends on ``mix_seed`` only, so training and held-out slides share the same 'biology'."""
    rng = np.random.default_rng(seed)
    n_attr = 6
    mix = np.random.default_rng(mix_seed).normal(0, 1, size=(n_attr, rna_features)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    imgs = np.empty((n_slides * tiles_per_slide, 3, size, size), dtype=np.float32)
    rna = np.empty((n_slides * tiles_per_slide, rna_features), dtype=np.float32)
    sid = np.empty(n_slides * tiles_per_slide, dtype=np.int64)
    k = 0
    for s in range(n_slides):
        attr = rng.uniform(0, 1, size=n_attr).astype(np.float32)      # density, nucleus size, eosin hue, haematoxylin depth, texture, elongation
        row = (attr - 0.5) @ mix + 0.1 * rng.normal(0, 1, size=rna_features).astype(np.float32)
        for _ in range(tiles_per_slide):
            bg = np.array([0.93 - 0.15 * attr[2], 0.70 + 0.15 * attr[2] - 0.1 * attr[4], 0.85 - 0.1 * attr[2]], dtype=np.float32)
            img = np.broadcast_to(bg[:, None, None], (3, size, size)).copy()
            # smooth stromal texture: a few random low-frequency waves
            for _w in range(3):
                fx, fy = rng.uniform(0.02, 0.12, size=2) * (0.5 + attr[4])
                ph = rng.uniform(0, 2 * np.pi)
                img += (0.04 + 0.05 * attr[4]) * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)[None] * np.array([1.0, 0.6, 0.8], dtype=np.float32)[:, None, None]
            n_nuc = rng.poisson(2 + 22 * attr[0] * (size / 64.0) ** 2)
            nuc = np.array([0.35 - 0.2 * attr[3], 0.20 - 0.1 * attr[3], 0.55 - 0.15 * attr[3]], dtype=np.float32)
            if n_nuc > 0:
                img = img.astype(np.float64)          # (what the whole-image blend with the float64 mask made of it: same values)
            for _n in range(n_nuc):
                cx, cy = rng.uniform(0, size, size=2)
                r = (1.5 + 3.5 * attr[1]) * (size / 64.0) * rng.uniform(0.7, 1.3)
                el = 1.0 + 1.5 * attr[5] * rng.uniform(0, 1)
                th = rng.uniform(0, np.pi)
                # (the blend is the identity where m = 0, i.e. outside the ellipse u^2 + v^2 < 1.5: only its bounding box is touched --
                # bit-identical to the whole-image form, 4-60 x faster at 256 x 256)
                hb = int(np.ceil(1.25 * r * el)) + 2
                y0, y1 = max(0, int(cy) - hb), min(size, int(cy) + hb + 1)
                x0, x1 = max(0, int(cx) - hb), min(size, int(cx) + hb + 1)
                if y0 >= y1 or x0 >= x1:
                    continue
                dx, dy = xx[y0:y1, x0:x1] - cx, yy[y0:y1, x0:x1] - cy
                u = (dx * np.cos(th) + dy * np.sin(th)) / (r * el)
                v = (-dx * np.sin(th) + dy * np.cos(th)) / r
                m = np.clip(1.5 - (u * u + v * v), 0, 1)[None]
                img[:, y0:y1, x0:x1] = img[:, y0:y1, x0:x1] * (1 - m) + nuc[:, None, None] * m
            imgs[k] = np.clip(img, 0, 1)
            rna[k] = row
            sid[k] = s
            k += 1
    rna = (rna - rna.mean(0, keepdims=True)) / (rna.std(0, keepdims=True) + 1e-6)      # StandardScaler (src/histopathology_gan.py:148-151)
    return torch.from_numpy((imgs - 0.5) / 0.5), torch.from_numpy(rna), torch.from_numpy(sid)


# ---------------------------------------------------------------------------------------------- one training run
def build_models(size, step, enc, precision, seed, device):
    import rna_gan_amd as P
    from rna_gan_amd import synth as R
    G = P.DCGANGenerator(enc, size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    D = P.DCGANDiscriminator(size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
    R.seeded_fill_(G, seed); R.seeded_fill_(D, seed + 1)
    G.set_precision(precision); D.set_precision(precision)
    G, D = G.to(device).train(), D.to(device).train()
    og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
    od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
    return G, D, og, od


def build_plugins(loss_type, rna_features, precision, device, vae_seed=4242):
    import rna_gan_amd as P
    from rna_gan_amd import synth as R
    if loss_type == "wgan":
        return [P.WassersteinGeneratorLoss(), P.WassersteinDiscriminatorLoss(clip=(-0.01, 0.01)), P.WassersteinGradientPenalty()]
    lg = P.WassersteinGeneratorLossVAE(checkpoint=None, rna_features=rna_features)
    ld = P.WassersteinDiscriminatorLossVAE(checkpoint=None, rna_features=rna_features)
    lp = P.WassersteinGradientPenaltyVAE(checkpoint=None, rna_features=rna_features)
    R.seeded_fill_(lg.betavae, vae_seed)            # the FROZEN encoder is the same in every run (it is an input, not trained here)
    sd = lg.betavae.state_dict()
    for l in (lg, ld, lp):
        if l is not lg:
            l.betavae.load_state_dict(sd)
        l.betavae.set_precision(precision)
        l.betavae = l.betavae.to(device).eval()
    return [lg, ld, lp]


def train_run(loss_type, precision, seed, data, size, step, enc, iters, batch, device, log=None):
    """One run; returns (G, D, plugins, loss curves (iters, 3))."""
    from rna_gan_amd import losses as PL
    imgs, rna, _ = data
    G, D, og, od = build_models(size, step, enc, precision, seed, device)
    lg, ld, lp = build_plugins(loss_type, rna.shape[1], precision, device)
    gen = torch.Generator(device="cpu").manual_seed(1000 + seed)
    order_rng = np.random.default_rng(5000 + seed)
    n = imgs.shape[0]
    curves = np.zeros((iters, 3), dtype=np.float64)
    perm, pos = order_rng.permutation(n), 0
    t0 = time.perf_counter()
    for it in range(iters):
        if pos + batch > n:
            perm, pos = order_rng.permutation(n), 0
        idx = torch.from_numpy(perm[pos:pos + batch]); pos += batch
        real = imgs[idx].to(device)
        eps = torch.empty(1).uniform_(0.0, 1.0, generator=gen).to(device)
        PL.new_batch()
        if loss_type == "wgan":
            nz = [torch.randn(batch, enc, generator=gen).to(device) for _ in range(3)]
            vals = [lg.step(G, D, og, nz[0]), ld.step(G, D, od, real, nz[1], next_noise=nz[2]), lp.step(G, D, od, real, nz[2], eps)]
        else:
            r = rna[idx].to(device)
            u = [torch.empty(batch, enc).uniform_(-0.3, 0.3, generator=gen).to(device) for _ in range(3)]
            vals = [lg.step(G, D, og, r, u[0]), ld.step(G, D, od, real, r, u[1], next_u=u[2]), lp.step(G, D, od, real, r, u[2], eps)]
        curves[it] = [float(v.item()) for v in vals]
        if not np.isfinite(curves[it]).all():
            raise RuntimeError("%s %s seed %d: non-finite loss at iteration %d: %s" % (loss_type, precision, seed, it, curves[it]))
    PL.new_batch()
    torch.cuda.synchronize(device)
    if log:
        log("  trained %s / %s / seed %d: %d iterations in %.1f s; last losses %s" % (
            loss_type, precision, seed, iters, time.perf_counter() - t0, np.round(curves[-1], 4).tolist()))
    return G, D, (lg, ld, lp), curves


@torch.no_grad()
def _fp32_twin(module, size, step, enc, device):
    """A fresh module of the same architecture on the fp32 kernels holding ``module``'s parameters and buffers."""
    import rna_gan_amd as P
    if isinstance(module, P.DCGANGenerator):
        twin = P.DCGANGenerator(enc, size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    else:
        twin = P.DCGANDiscriminator(size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
    twin.load_state_dict({k: v.detach().cpu().clone() for k, v in module.state_dict().items()})
    twin.set_precision("fp32")
    return twin.to(device)


def sample(G, plugins, loss_type, rna_rows, n_eval, size, step, enc, chunk, device, seed):
    """n_eval images (N, 3, S, S) fp32 in [-1, 1] on the host from a COPY of G on the fp32 kernels, train-mode BatchNorm on
    chunks (the mode the reference's generate_images leaves a loaded trainer in, src/gan_utils.py:217-221), same draws for
    every generator."""
    Gs = _fp32_twin(G, size, step, enc, device).train()
    ops, _ = Gs.runtime()
    gen = torch.Generator(device="cpu").manual_seed(seed)
    outs = []
    for i in range(0, n_eval, chunk):
        m = min(chunk, n_eval - i)
        if loss_type == "wgan":
            nz = torch.randn(m, enc, generator=gen).to(device)
        else:
            u = torch.empty(m, enc).uniform_(-0.3, 0.3, generator=gen).to(device)
            bv = plugins[0].betavae
            z = bv.encode(rna_rows[i:i + m].to(device).float().contiguous(), mean_only=True)[0]
            nz = ops.latent_prep(u.contiguous(), z.detach().float().contiguous())
        outs.append(Gs(nz.float().contiguous()).float().cpu())
    return torch.cat(outs, 0)


# ---------------------------------------------------------------------------------------------- features / distances
def trunk_features(Dc, images):
    from rna_gan_amd import fid as F
    return F.discriminator_features(Dc, images, batch_size=64)


def inception_features(images, device, seed=31337, batch=64):
    """Pool features of the HIP Inception-v3 with fixed seeded random weights; input resized to 299 x 299 as src/fid.py:166-190."""
    from rna_gan_amd.inception import InceptionV3
    from rna_gan_amd import synth as R
    net = _INCEPTION.get(seed)
    if net is None:
        net = InceptionV3()
        R.seeded_fill_(net, seed)
        net = _INCEPTION[seed] = net.to(device).eval()
    feats = []
    with torch.no_grad():
        for i in range(0, images.shape[0], batch):
            x01 = (images[i:i + batch].to(device).float() + 1.0) * 0.5
            x = torch.nn.functional.interpolate(x01, size=(299, 299), mode="bilinear", align_corners=False).clamp_(0, 1)
            feats.append(net.features(x).float().cpu().numpy())
    return np.concatenate(feats, 0)


_INCEPTION = {}


def fd(a, b):
    from rna_gan_amd import fid as F
    m1, s1 = F.activation_statistics(a)
    m2, s2 = F.activation_statistics(b)
    return F.frechet_distance(m1, s1, m2, s2)


def windows(curve, k=5):
    n = curve.shape[0]
    edges = np.linspace(0, n, k + 1).astype(int)
    return [[round(float(curve[a:b, j].mean()), 5) for j in range(3)] for a, b in zip(edges[:-1], edges[1:])]


# ---------------------------------------------------------------------------------------------- the A/B
def run_ab(loss_type, size=64, step=64, enc=2048, iters=500, batch=64, n_eval=2048, rna_features=512, n_slides=64,
           tiles_per_slide=48, seeds=(11, 12), device="cuda:0", inception=True, log=print, b_precision="bf16"):
    device = torch.device(device)
    train = make_slides(n_slides, tiles_per_slide, size, rna_features, seed=1)
    held = make_slides(max(8, n_eval // tiles_per_slide + 1), tiles_per_slide, size, rna_features, seed=2)
    held_imgs, held_rna = held[0][:n_eval], held[1][:n_eval]
    log("%s: %d training tiles of %d slides, %d held-out tiles; %d iterations at batch %d, %d x %d" % (
        loss_type, train[0].shape[0], n_slides, held_imgs.shape[0], iters, batch, size, size))
    runs = {}
    # (arm B: the 16-bit mode under test -- bf16, the benchmarked one, or fp16 with its static loss scale, BASELINE configs[3];
    # the record keeps the key names of the bf16 table and says which mode B was in "b_precision")
    for tag, prec, seed in (("A_fp32_s1", "fp32", seeds[0]), ("B_bf16_s1", b_precision, seeds[0]), ("C_fp32_s2", "fp32", seeds[1])):
        runs[tag] = train_run(loss_type, prec, seed, train, size, step, enc, iters, batch, device, log)
    D_ref = _fp32_twin(runs["A_fp32_s1"][1], size, step, enc, device)
    sets = {tag: sample(G, pl, loss_type, held_rna, n_eval, size, step, enc, 64, device, seed=99)
            for tag, (G, _, pl, _) in runs.items()}
    sets["real_heldout"] = held_imgs
    rec = {"loss_type": loss_type, "size": size, "iters": iters, "batch": batch, "n_eval": int(held_imgs.shape[0]),
           "step_channels": step, "encoding_dims": enc, "seeds": list(seeds), "b_precision": b_precision,
           "loss_curves_windowed_means[g,d,gp]": {tag: windows(r[3]) for tag, r in runs.items()}, "frechet": {}}
    extractors = [("d_trunk(A)", lambda x: trunk_features(D_ref, x))]
    if inception:
        extractors.append(("inception_random_weights", lambda x: inception_features(x, device)))
    for name, fx in extractors:
        t0 = time.perf_counter()
        feats = {tag: fx(x) for tag, x in sets.items()}
        row = {"dims": int(next(iter(feats.values())).shape[1]),
               "FD(B_bf16, A_fp32)": fd(feats["B_bf16_s1"], feats["A_fp32_s1"]),
               "FD(C_fp32_seed2, A_fp32)": fd(feats["C_fp32_s2"], feats["A_fp32_s1"]),
               "FD(A, real)": fd(feats["A_fp32_s1"], feats["real_heldout"]),
               "FD(B, real)": fd(feats["B_bf16_s1"], feats["real_heldout"]),
               "FD(C, real)": fd(feats["C_fp32_s2"], feats["real_heldout"])}
        row["ratio bf16-vs-fp32 / fp32-vs-fp32"] = row["FD(B_bf16, A_fp32)"] / max(row["FD(C_fp32_seed2, A_fp32)"], 1e-30)
        rec["frechet"][name] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in row.items()}
        log("  %s (%d-d, %.1f s): %s" % (name, row["dims"], time.perf_counter() - t0, json.dumps(rec["frechet"][name])))
    return rec


def markdown(recs):
    b = recs[0].get("b_precision", "bf16") if recs else "bf16"
    lines = ["| loss type | features | FD(%s, fp32) same seed | FD(fp32 seed 2, fp32 seed 1) | ratio | FD(fp32 s1, real) | FD(%s s1, real) | FD(fp32 s2, real) |" % (b, b),
             "|---|---|---|---|---|---|---|---|"]
    for r in recs:
        for name, row in r["frechet"].items():
            lines.append("| %s | %s (%d-d) | %.4g | %.4g | %.2f | %.4g | %.4g | %.4g |" % (
                r["loss_type"], name, row["dims"], row["FD(B_bf16, A_fp32)"], row["FD(C_fp32_seed2, A_fp32)"],
                row["ratio bf16-vs-fp32 / fp32-vs-fp32"], row["FD(A, real)"], row["FD(B, real)"], row["FD(C, real)"]))
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loss-types", default="wgan,wganvae")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--n-eval", type=int, default=2048)
    ap.add_argument("--step", type=int, default=64)
    ap.add_argument("--enc", type=int, default=2048)
    ap.add_argument("--n-slides", type=int, default=64)
    ap.add_argument("--tiles-per-slide", type=int, default=48)
    ap.add_argument("--rna-features", type=int, default=512)
    ap.add_argument("--no-inception", action="store_true")
    ap.add_argument("--b-precision", default="bf16", choices=["bf16", "fp16"], help="the 16-bit mode of arm B")
    ap.add_argument("--out", default="gpurun_out/train_quality_ab.json")
    args = ap.parse_args()
    log = lambda *a: print(*a, file=sys.stderr, flush=True)
    recs = [run_ab(lt, size=args.size, step=args.step, enc=args.enc, iters=args.iters, batch=args.batch, n_eval=args.n_eval,
                   rna_features=args.rna_features, n_slides=args.n_slides, tiles_per_slide=args.tiles_per_slide,
                   inception=not args.no_inception, log=log, b_precision=args.b_precision) for lt in args.loss_types.split(",")]
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(recs, f, indent=1)
    print(markdown(recs))


if __name__ == "__main__":
    main()
