#!/usr/bin/env python3
"""Working counterpart of the reference CLI (src/histopathology_gan.py) on the MI355X path.

Same flags (--config --checkpoint --seed --image_dir --model_dir --num_epochs --num_patches
--gan_type --loss_type) and the same JSON keys (path_csv, patch_data_path, save_dir, img_size,
[flag, encoder_checkpoint, bag_size]); the model/optimizer dictionary, loss selection, Trainer call
and checkpoint naming follow src/histopathology_gan.py:175-192,248-278,298-314.

The reference script itself cannot start (absent modules wsi_model/biggan/sagan).  Real data goes
through rna_gan_amd.data (the reference's tile-record format, per-slide sampling and RNA log / StandardScaler
preparation; slide databases as LMDB files when the ``lmdb`` package is installed, or as directory stores);
``--synthetic`` trains on synthetic tiles / RNA rows of the right shapes (what bench.py measures).  Extra flags: --batch_size (reference hard-codes 8),
--precision, --betavae_checkpoint, --steps_per_epoch.
"""
import argparse
import datetime
import json
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this driver stack (before torch initialises HIP)

import numpy as np
import torch
import torch.nn as nn
from torch.optim import Adam
from torch.utils.data import DataLoader, Dataset

import rna_gan_amd as P
from rna_gan_amd import dist as D_


class SyntheticTiles(Dataset):
    """uint8 uniform tiles -> float -> (x-0.5)/0.5 (src/histopathology_gan.py:106-109) and N(0,1) RNA rows
    (StandardScaler output, :148-151) with 16 distinct rows (tiles of a slide share RNA)."""

    def __init__(self, n, img_size, rna_features, with_rna, seed):
        self.n, self.s, self.f, self.with_rna = n, img_size, rna_features, with_rna
        rng = np.random.default_rng(seed)
        self.rows = torch.from_numpy(rng.normal(size=(16, rna_features)).astype(np.float32))
        self.seed = seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rng = np.random.default_rng([self.seed, i])
        img = torch.from_numpy(rng.integers(0, 256, size=(3, self.s, self.s), dtype=np.uint8)).float() / 255.0
        img = (img - 0.5) / 0.5
        if self.with_rna:
            return {"image": img, "rna_data": self.rows[i % 16], "labels": torch.tensor(0.0)}
        return img, torch.tensor(0.0)


# (flag, type, default, help) -- the reference's flag set (src/histopathology_gan.py:54-72) + this build's extras
REFERENCE_FLAGS = [
    ("--config", str, None, "JSON config file"),
    ("--checkpoint", str, None, "File with the checkpoint to start with"),
    ("--seed", int, 99, "Seed for random generation"),
    ("--image_dir", str, "images", "Image dir to save image"),
    ("--model_dir", str, "./model/gan", "Image dir to save model checkpoints"),
    ("--num_epochs", int, None, "Number of epochs to train the model"),
    ("--num_patches", int, 250, "Number of tiles to use per slide"),
    ("--gan_type", str, "dcgan", "Architecture to use"),
    ("--loss_type", str, "wgangp", "Loss type to use"),
]
EXTRA_FLAGS = [
    ("--sync_stats", int, 0, "data parallel: 1 = BatchNorm / latent / penalty statistics over the GLOBAL batch (exact "
                              "single-process semantics, no HIP graphs), 0 = rank-local statistics (plain DDP)"),
    ("--batch_size", int, 8, "per-process batch (the reference hard-codes 8, :94)"),
    ("--precision", str, "bf16", "bf16 (MFMA kernels), fp16 (the same kernels built for IEEE fp16 storage, loss-scaled backward) or fp32 (parity mode)"),
    ("--betavae_checkpoint", str, "checkpoints/betavae_training_tissues/model_dict_best.pt", "frozen betaVAE weights"),
    ("--steps_per_epoch", int, 100, "synthetic dataset length / batch"),
]


def make_collate_fn(with_rna: bool, world: int):
    """The reference's collate_fn (src/histopathology_gan.py:24-34): drop records whose tile could not be read.  Data
    parallel (world > 1): every rank must see the SAME batch size -- the gathered G.0 gradient factors and the captured step
    graphs are sized by it, and each train_op issues a collective -- so the dropped records are replaced by repeating this
    batch's readable ones instead of shrinking the batch."""
    def collate_fn(batch):
        img = (lambda b: b["image"]) if with_rna else (lambda b: b[0])
        want = len(batch)
        batch = [b for b in batch if img(b) is not None]
        if world > 1 and len(batch) < want:
            if not batch:
                raise RuntimeError("data parallel: a batch with no readable tile record cannot be equalised across ranks")
            batch = (batch * ((want + len(batch) - 1) // len(batch)))[:want]
        return torch.utils.data.dataloader.default_collate(batch)
    return collate_fn


def shard_indices(n_items: int, rank: int, world: int, batch_size: int):
    """Indices of rank ``rank``'s shard of a list of n_items: strided (rank, rank + world, ...), truncated so that EVERY rank
    gets the same number of items and that number is a multiple of the batch size (equal batch counts on all ranks)."""
    per_rank = (n_items // world) // batch_size * batch_size
    return list(range(rank, n_items, world))[:per_rank]


def parse_args():
    ap = argparse.ArgumentParser(description="GANs training on histology data (MI355X path)")
    for flag, typ, default, text in REFERENCE_FLAGS + EXTRA_FLAGS:
        ap.add_argument(flag, type=typ, default=default, help=text)
    ap.add_argument("--synthetic", action="store_true", help="train on synthetic tiles / RNA rows")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.precision not in ("bf16", "fp32", "fp16"):
        raise SystemExit("--precision must be bf16, fp16 or fp32")

    D_.set_sync_stats(bool(args.sync_stats))
    D_.init_from_env()
    # every rank builds its replicas (G, D, and -- without a checkpoint file -- the betaVAE copies inside the three
    # loss plugins) from the SAME seed; the per-rank offset is applied after construction, for noise / eps / data only
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    with open(args.config) as f:
        config = json.load(f)
    if D_.rank() == 0:
        print(10 * "-"); print("Config for this experiment \n"); print(config); print(10 * "-")
    args.flag = config.get("flag", "train_{date:%Y-%m-%d %H:%M:%S}".format(date=datetime.datetime.now()))
    img_size = config["img_size"]
    rna_features = config.get("rna_features", 19198)
    with_rna = args.loss_type == "wganvae"
    if args.synthetic:
        ds = SyntheticTiles(args.steps_per_epoch * args.batch_size, img_size, rna_features, with_rna,
                            args.seed + 1000 * D_.rank())
        loader = DataLoader(ds, batch_size=args.batch_size, num_workers=0, pin_memory=True, drop_last=True)
    else:
        # the reference's data path (src/histopathology_gan.py:111-168): slide tables -> [log + StandardScaler on the
        # rna_ columns] -> per-slide tile sampling from the slide databases -> batches; rna_gan_amd.data restates the
        # record format / sampling / preparation (LMDB files need the `lmdb` package, directory stores do not)
        from rna_gan_amd import data as PD
        import random
        # every rank must build the SAME tile list (the datasets draw their per-slide tile sample from the global `random`
        # generator, src/read_data.py:313-316), so that the strided shards below partition one list
        random.seed(args.seed)
        patch_data_path = config["patch_data_path"]
        train_df = PD.load_slide_tables(config["path_csv"], patch_data_path)
        tf = PD.ToFloatNormalize(0.5, 0.5)
        if with_rna:
            train_df, _, _ = PD.log_standardize_rna(train_df)
            ds = PD.PatchRNADataset(patch_data_path, train_df, img_size, max_patches_total=args.num_patches, transforms=tf)
        else:
            ds = PD.PatchDataset(patch_data_path, train_df, img_size, max_patches_total=args.num_patches, transforms=tf)

        world = D_.world_size()
        collate_fn = make_collate_fn(with_rna, world)
        if world > 1:
            # one shard of the (identical) tile list per rank, the SAME number of full batches on every rank: each train_op
            # issues a gradient all-reduce, so a rank with one batch more would pair its collectives with nobody
            ds = torch.utils.data.Subset(ds, shard_indices(len(ds), D_.rank(), world, args.batch_size))
        loader = DataLoader(ds, batch_size=args.batch_size, num_workers=4, pin_memory=True, collate_fn=collate_fn,
                            drop_last=world > 1)   # :163-168

    if args.gan_type not in ("dcgan", "dcgan_up"):
        raise SystemExit("--gan_type dcgan (the reference CLI's path) or dcgan_up (src/dcgan.py's DCGANUpGenerator, "
                         "which the reference defines but never selects); condgan/biggan/sagan sources are absent upstream")
    gan_network = {
        "generator": {"name": P.DCGANUpGenerator if args.gan_type == "dcgan_up" else P.DCGANGenerator,
                      "args": {"encoding_dims": 2048, "out_channels": 3, "step_channels": 64, "out_size": img_size,
                               "nonlinearity": nn.LeakyReLU(0.2), "last_nonlinearity": nn.Tanh()},
                      "optimizer": {"name": Adam, "args": {"lr": 0.0001, "betas": (0.5, 0.999)}}},
        "discriminator": {"name": P.DCGANDiscriminator,
                          "args": {"in_size": img_size, "in_channels": 3, "step_channels": 64,
                                   "nonlinearity": nn.LeakyReLU(0.2), "last_nonlinearity": nn.LeakyReLU(0.2)},
                          "optimizer": {"name": Adam, "args": {"lr": 0.0004, "betas": (0.5, 0.999)}}},
    }
    if args.loss_type == "wgan":
        losses = [P.WassersteinGeneratorLoss(), P.WassersteinDiscriminatorLoss(clip=(-0.01, 0.01)),
                  P.WassersteinGradientPenalty()]
    elif args.loss_type == "wganvae":
        ck = args.betavae_checkpoint if os.path.exists(args.betavae_checkpoint) else None
        if ck is None and D_.rank() == 0:
            print("betaVAE checkpoint not found: using randomly initialised encoder weights")
        losses = [P.WassersteinGeneratorLossVAE(checkpoint=ck, rna_features=rna_features),
                  P.WassersteinDiscriminatorLossVAE(checkpoint=ck, rna_features=rna_features),
                  P.WassersteinGradientPenaltyVAE(checkpoint=ck, rna_features=rna_features)]
    else:
        raise SystemExit(f"Loss type {args.loss_type} not implemented on this path. Choose wgan or wganvae.")

    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RNAGAN_DIST_BACKEND") == "gloo":
        local %= max(torch.cuda.device_count(), 1)       # functional multi-rank runs on fewer devices than ranks (dist.init_from_env)
    device = torch.device("cuda", local)
    epochs = args.num_epochs if args.num_epochs is not None else 5
    print("Device: {}".format(device)); print("Epochs: {}".format(epochs))
    trainer = P.Trainer(gan_network, losses, checkpoints=args.model_dir, sample_size=64, epochs=epochs, devices=[0],
                        recon=args.image_dir, device=device, precision=args.precision)
    if args.checkpoint is not None:
        trainer.load_model(load_path=args.checkpoint)
    for loss in losses:                                   # identical frozen encoders on every rank (rank 0's)
        bv = getattr(loss, "betavae", None)
        if bv is not None and D_.world_size() > 1:
            bv.to(device)
            before = bv.signature()                       # which of this rank's three encoders held the same weights
            for t in list(bv.parameters()) + list(bv.buffers()):
                D_.broadcast_(t.data, 0)
            bv.weights_changed()
            # every rank runs the same construction, so encoders that shared weights before the broadcast (one checkpoint
            # file) share rank 0's afterwards: keep that identity for the per-batch latent cache (losses._LatentCache)
            bv.adopt_weights_token(("broadcast",) + tuple(before))
    torch.manual_seed(args.seed + D_.rank())
    np.random.seed(args.seed + D_.rank())
    trainer(loader)
    D_.flush()                     # data parallel: apply a trailing optimizer step before the process group goes away
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
