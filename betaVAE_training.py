#!/usr/bin/env python3
"""Working counterpart of the reference's src/betaVAE_training.py on the MI355X path.

Same flags (--config --checkpoint --seed --log --parallel) and JSON keys (save_dir, rna_features, batch_size, beta,
optimizer, lr, weights_decay, num_epochs, log_interval); model / optimizer / scheduler construction follows
src/betaVAE_training.py:131-176 (betaVAE(rna_features, 2048, [6000, 4000, 2048], [4000, 6000], beta), xavier init,
Adam(weight_decay, lr), CosineAnnealingLR(500)); the loop is rna_gan_amd.vae_train.train_betaVAE.

The RNA CSV pipeline (pandas merge / log / StandardScaler, src/betaVAE_training.py:60-112, src/read_data.py) is outside
this build's scope: ``--synthetic`` trains on standardised synthetic expression rows of the right width.  The
reference's GradualWarmupScheduler (third-party ``warmup_scheduler``, absent here) is replaced by torch's LinearLR
warm-up chained in front of the same cosine schedule.
"""
import argparse
import json
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

import rna_gan_amd as P
from rna_gan_amd import vae_train as VT


class SyntheticRNA(Dataset):
    """N(0,1) rows squashed into tanh's range (StandardScaler output; the decoder ends in Tanh)."""

    def __init__(self, n, features, seed):
        rng = np.random.default_rng(seed)
        self.rows = torch.from_numpy(np.tanh(rng.normal(size=(n, features)).astype(np.float32)))

    def __len__(self):
        return self.rows.shape[0]

    def __getitem__(self, i):
        return {"rna_data": self.rows[i]}


def init_weights_xavier(m):
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.xavier_uniform_(m.weight)
        m.bias.data.fill_(0.01)


def main():
    ap = argparse.ArgumentParser(description="betaVAE training over RNA-Seq data (MI355X path)")
    ap.add_argument("--config", type=str, help="JSON config file")
    ap.add_argument("--checkpoint", type=str, default=None, help="File with the checkpoint to start with")
    ap.add_argument("--seed", type=int, default=99, help="Seed for random generation")
    ap.add_argument("--log", type=int, default=0, help="Use tensorboard for experiment logging (needs tensorboardX)")
    ap.add_argument("--parallel", type=int, default=None, help="accepted for compatibility; one process drives one GPU")
    ap.add_argument("--synthetic", action="store_true", help="train on synthetic standardised expression rows")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--samples", type=int, default=1024, help="synthetic training rows")
    args = ap.parse_args()
    with open(args.config) as f:
        config = json.load(f)
    print(10 * "-"); print("Config for this experiment \n"); print(config); print(10 * "-")
    if not args.synthetic:
        raise SystemExit("real RNA-Seq tables need the reference's CSV pipeline (out of scope here); use --synthetic")
    torch.manual_seed(args.seed)
    rna_features = config.get("rna_features", 19198)
    batch_size = config.get("batch_size", 64)
    os.makedirs(config["save_dir"], exist_ok=True)
    loaders = {"train": DataLoader(SyntheticRNA(args.samples, rna_features, args.seed), batch_size=batch_size, shuffle=True,
                                   drop_last=True),
               "val": DataLoader(SyntheticRNA(max(batch_size, args.samples // 8), rna_features, args.seed + 1),
                                 batch_size=batch_size)}
    model = P.betaVAE(rna_features, 2048, [6000, 4000, 2048], [4000, 6000], beta=config.get("beta", 2))
    if args.checkpoint is not None:
        model.load_state_dict(torch.load(args.checkpoint, map_location="cpu"))
    else:
        model.apply(init_weights_xavier)
    model = model.set_precision(args.precision).cuda()
    if config.get("optimizer", "Adam") != "Adam":
        raise SystemExit("only Adam has a fused HIP step (the reference's default)")
    optimizer = P.Adam(model.parameters(), weight_decay=config.get("weights_decay", 0.0), lr=config.get("lr", 3e-3)).bind(model)
    cosine = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, 500)
    warm = torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=1e-3, total_iters=config.get("warmup_steps", 1000))
    scheduler = torch.optim.lr_scheduler.SequentialLR(optimizer, [warm, cosine], milestones=[config.get("warmup_steps", 1000)])
    writer = None
    if args.log:
        from tensorboardX import SummaryWriter
        writer = SummaryWriter(config.get("summary_path", os.path.join(config["save_dir"], "tb")))
    model, results = VT.train_betaVAE(model, optimizer, loaders, save_dir=config["save_dir"],
                                      log_interval=config.get("log_interval", 100), summary_writer=writer,
                                      num_epochs=config.get("num_epochs", 5), scheduler=scheduler)
    test_loss, _, _ = VT.evaluate_betaVAE(model, loaders["val"])
    print("best epoch", results["best_epoch"], "best validation loss", results["best_loss"], "final", test_loss)


if __name__ == "__main__":
    main()
