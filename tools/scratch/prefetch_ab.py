#!/usr/bin/env python3
# Trainer.train() on HOST batches (the reference CLI's flow: a DataLoader of pinned dict batches, wganvae plugins) at the
# benchmark's size: ms per iteration with the device prefetcher on / off, interleaved.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import rna_gan_amd as P
from rna_gan_amd import synth as R
args = bench.parse_args([])
dev = torch.device("cuda:0")
N = args.batch
G, Dm, og, od, losses = bench.build(dev, "bf16", N, 19198, args.seed)
rna = R.synthetic_rna(N, 19198, seed=4321, distinct=16)
host = [{"image": (R.synthetic_tiles_u8(N, 256, seed=100 + k).float() / 255 - 0.5) / 0.5, "rna_data": rna.clone(), "labels": None}
        for k in range(4)]
host = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()} for b in host]

class Loader:
    batch_size = N
    def __init__(self, n): self.n = n
    def __len__(self): return self.n
    def __iter__(self):
        for i in range(self.n): yield MODE["src"][i % len(host)]

tr = P.Trainer.__new__(P.Trainer)
tr.device = dev; tr.generator, tr.discriminator = G, Dm
tr.optimizer_generator, tr.optimizer_discriminator = og, od
tr.model_names = ["generator", "discriminator"]; tr.schedulers = []
tr.losses = {type(l).__name__: l for l in losses}
tr.loss_logs = {n: [] for n in tr.losses}
tr.loss_information = {"generator_losses": 0.0, "discriminator_losses": 0.0, "generator_iters": 0, "discriminator_iters": 0}
tr.ncritic = 1; tr.labels = None; tr.start_epoch = 0; tr.epochs = 1; tr.recon = None; tr.sample_size = 4
tr.test_noise = torch.zeros(1); tr.save_model = lambda epoch: None
MODE = {"src": host}
dev_batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in host]
def run(prefetch, n, pipeline=True, resident=False):
    tr.prefetch = prefetch
    tr.pipeline = pipeline
    MODE["src"] = dev_batches if resident else host
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr(Loader(n))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(True, 12); run(False, 12)
TIMELINE_ONLY = os.environ.get("PREFETCH_AB_ONLY") == "timeline"
if os.environ.get("PREFETCH_AB_ONLY") == "resident":       # for rocprofv3: one mode, 40 iterations
    print("device-resident batches, pipelined   %.3f ms/iteration" % run(False, 40, True, True), flush=True)
    sys.exit(0)
for r in range(0 if TIMELINE_ONLY else 2):
    print("prefetch on  %.3f ms/iteration" % run(True, 30), flush=True)
    print("prefetch off %.3f ms/iteration" % run(False, 30), flush=True)
    print("device-resident batches, pipelined   %.3f ms/iteration" % run(False, 30, True, True), flush=True)
    print("device-resident batches, synchronous %.3f ms/iteration" % run(False, 30, False, True), flush=True)
    print("host batches, prefetch, synchronous  %.3f ms/iteration" % run(True, 30, False, False), flush=True)
import cProfile, pstats
if os.environ.get("PREFETCH_AB_ONLY") != "timeline":
    pass
if not TIMELINE_ONLY:
    pr = cProfile.Profile(); pr.enable(); run(True, 30); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

# ---- host-side timeline of the pipelined loop on device-resident batches: per-iteration mean of each launch / wait
import statistics
from rna_gan_amd import losses as L
tr._store_loss_maps()
names = list(tr.losses)
T = {("launch", n): [] for n in names}; T.update({("wait", n): [] for n in names}); T[("total",)] = []
pend = None
for i in range(40):
    t_it = time.perf_counter()
    tr.real_inputs = dev_batches[i % len(dev_batches)]
    L.new_batch()
    for n in names:
        loss = tr.losses[n]
        t0 = time.perf_counter()
        val = tr._post(loss.train_ops_async(**tr._get_arguments(tr._arg_maps[n])))
        t1 = time.perf_counter()
        if pend is not None:
            tr._take(pend[1])
            T[("wait", pend[0])].append(time.perf_counter() - t1)
        T[("launch", n)].append(t1 - t0)
        pend = (n, val)
    T[("total",)].append(time.perf_counter() - t_it)
tr._take(pend[1])
for k, v in T.items():
    v = v[10:]
    print("%-60s mean %.3f ms  max %.3f ms" % (" ".join(k), statistics.mean(v) * 1e3, max(v) * 1e3))
