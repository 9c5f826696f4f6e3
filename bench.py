#!/usr/bin/env python3
"""bench.py -- training imgs/sec of the RNA-GAN WGAN-GP iteration on MI355X.

One "step" = one full hot-loop iteration of the reference (src/histopathology_gan.py:298-314 ->
torchgan Trainer -> the three train_ops of src/wgan_loss.py): generator-loss step, discriminator-loss
step, gradient-penalty step (3 optimizer steps, 3 G forwards, 4 D forwards, all backwards incl. the
second-order GP pass), on synthetic 256x256 RGB tiles with betaVAE-conditioned noise (wganvae path),
reference model size (encoding 2048, step_channels 64, rna_features 19198).

Workload = BASELINE.json configs[1] ("RNA-GAN lung 256x256 bf16 batch 64 on 1xMI355X"); with
--gpus N it is configs[2] (64 per rank, weak scaling, RCCL gradient all-reduce).

Contract: W untimed warm-up steps, K timed steps bracketed by barrier + synchronize, MAX over
ranks, rank 0 prints ONE JSON line.  Extra objects:
  roofline     : dominant kernel (bf16 MFMA implicit-GEMM conv, gather_gemm_kernel), achieved =
                 algorithmic conv FLOPs of its launches / their summed duration, measured with
                 HIP events inside this process on a dedicated pass (see DESIGN.md "Measurement").
  cpu_baseline : the oracle (plain PyTorch fp32 restatement of the reference path, oracle/ref_cpu.py)
                 timed on this box's host cores on a bounded sample (batch 8, the reference's own
                 hard-coded batch size).
  parity       : the FIRST iteration of that batch-8 CPU-oracle run (seeded weights / inputs / draws) repeated through the
                 HIP product path in this process: the three losses side by side with the stated tolerance.
  config.extras: figures measured AFTER the headline timed region in the same process (never part of `value`):
                 generate (BASELINE configs[4]: generator-only synthesis of 4096 samples, fp8 and bf16), api_path (the
                 drop-in Trainer.train_iter() loop with its host syncs and uint8 input), fp32_step (the fp32 parity mode).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); the launchers
# export it, this is the default for a bare `torchrun bench.py` (read by the HSA runtime when HIP initialises: before torch)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 (MI355X_MICROARCH.md)


def conv_flops_per_image(in_size=256, step=64, enc=2048):
    """Algorithmic conv FLOPs of ONE pass (2*MACs), SURVEY 8d: D 5.469 GFLOP, G 5.604 GFLOP."""
    reps = in_size.bit_length() - 4
    d_layers, g_layers = [], []
    c, s = step, in_size // 2
    d_layers.append(2 * s * s * c * 3 * 16)
    for _ in range(reps):
        s //= 2
        d_layers.append(2 * s * s * (2 * c) * c * 16)
        c *= 2
    head = 2 * c * 16
    g0 = 2 * enc * c * 16
    cc, ss = c, 4
    for _ in range(reps):
        g_layers.append(2 * ss * ss * cc * (cc // 2) * 16)
        cc //= 2
        ss *= 2
    g_last = 2 * ss * ss * cc * 3 * 16
    return dict(D=sum(d_layers) + head, G=g0 + sum(g_layers) + g_last, D_mfma=sum(d_layers[1:]),
                G_mfma=sum(g_layers), D_layers=d_layers, G_layers=g_layers)


def build(device, precision, batch, rna_features, seed, gan_type="dcgan", enc=2048):
    import torch.nn as nn
    import rna_gan_amd as P
    from rna_gan_amd import synth as R       # seeded weight / input generators
    gen_cls = P.DCGANUpGenerator if gan_type == "dcgan_up" else P.DCGANGenerator
    G = gen_cls(enc, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    D = P.DCGANDiscriminator(256, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
    R.seeded_fill_(G, seed); R.seeded_fill_(D, seed + 1)
    G.set_precision(precision); D.set_precision(precision)
    G, D = G.to(device).train(), D.to(device).train()
    og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
    od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
    # the three loss plugins of --loss_type wganvae, each with its own frozen betaVAE copy as in the
    # reference (src/histopathology_gan.py:273-278); seeded weights instead of a checkpoint file
    lg = P.WassersteinGeneratorLossVAE(checkpoint=None, rna_features=rna_features)
    ld = P.WassersteinDiscriminatorLossVAE(checkpoint=None, rna_features=rna_features)
    lp = P.WassersteinGradientPenaltyVAE(checkpoint=None, rna_features=rna_features)
    if enc != 2048:
        # the plugins build betaVAE(rna_features, 2048, [6000, 4000, 2048], ...) as src/wgan_loss.py:67 hard-codes it; a
        # generator with another latent width needs the matching z_dim, which is also the encoder's last width
        # (src/betaVAE.py: z_mu = Linear(z_dim, z_dim))
        for l in (lg, ld, lp):
            l.betavae = P.betaVAE(rna_features, enc, [6000, 4000, enc], [4000, 6000], beta=0.005)
    R.seeded_fill_(lg.betavae, seed + 2)
    sd = lg.betavae.state_dict()
    for l in (lg, ld, lp):
        if l is not lg:
            l.betavae.load_state_dict(sd)
        l.betavae.set_precision(precision)
        l.betavae = l.betavae.to(device).eval()
    return G, D, og, od, (lg, ld, lp)


def log(*a):
    print("[bench %.1fs]" % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()
_DEFAULT_THREADS = 1


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--trace-steps", action="store_true", help="log every timed step's duration (diagnostic)")
    ap.add_argument("--sync-stats", action="store_true",
                    help="N > 1: global-batch BatchNorm / latent / penalty statistics (exact single-process semantics, no "
                         "HIP graphs); default: rank-local statistics (plain DDP), which every reported number uses")
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE configs[1])")
    ap.add_argument("--precision", default=os.environ.get("RNAGAN_BENCH_PRECISION", "bf16"), choices=["bf16", "fp32", "fp16"],
                    help="default bf16 (the headline); RNAGAN_BENCH_PRECISION sets the default for launchers that pass no flags")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the on-box ceiling probes (~15 s) of roofline.peak_measured")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary figures measured after the timed region (config.extras) and the parity object")
    ap.add_argument("--seed", type=int, default=99)
    ap.add_argument("--gan-type", default="dcgan", choices=["dcgan", "dcgan_up"],
                    help="dcgan = the reference CLI's generator (the benchmark); dcgan_up = src/dcgan.py's resize-convolution "
                         "generator (diagnostic: implies --no-roofline --no-cpu-baseline)")
    ap.add_argument("--api-path", action="store_true",
                    help="time the drop-in API instead of the kernel pipeline: Trainer.train_iter() -> the three train_ops "
                         "with their .item() host syncs, a fresh uint8 batch per iteration normalised on the device "
                         "(reported under config.api_path; the headline number is the default mode)")
    ap.add_argument("--step-plugin", default=None,
                    help="TEST HOOK module:function -- replaces the HIP workload by function(args, rank, world, device) -> "
                         "(one_step, flush, items_per_rank_step, info); lets tests/ run the launcher and the timing "
                         "protocol on CPU ranks (gloo).  Never used for a reported number.")
    args = ap.parse_args(argv)
    if args.gan_type != "dcgan":
        args.no_roofline = args.no_cpu_baseline = args.no_extras = True
    if args.api_path or args.precision != "bf16" or args.batch != 64:
        args.no_extras = True              # the extras belong to the headline configuration only
    return args


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torchrun: this process becomes the launcher.  It starts N fresh rank
    processes of this script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; it has not touched the GPU
    itself), relays rank 0's JSON line and exits non-zero if any rank fails (the other ranks are then stopped by PID)."""
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    out0 = b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                p = procs[r]
                if r == 0 and p.stdout is not None:
                    try:
                        o, _ = p.communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                code = p.poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    log("rank %d exited with code %d: stopping the other ranks" % (r, code))
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


def hip_workload(args, rank, world, device):
    """The product path: G / D / three loss plugins on `device`, synthetic inputs resident in HBM."""
    from rna_gan_amd import dist as D_
    from rna_gan_amd import synth as R
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py measures the HIP path: an MI355X is required (there is no CPU fallback)")
    N = args.batch
    rna_features = 19198
    G, Dm, og, od, (lg, ld, lp) = build(device, args.precision, N, rna_features, args.seed, args.gan_type)
    for mod in (G, Dm):
        for t in list(mod.parameters()) + list(mod.buffers()):
            D_.broadcast_(t.data, 0)
    from rna_gan_amd import graphed
    # the two resident input tensors are persistent buffers: the step graphs read them in place (no per-call copy)
    real = graphed.mark_static(R.synthetic_images(N, 256, seed=1234 + rank).to(device))
    rna = graphed.mark_static(R.synthetic_rna(N, rna_features, seed=4321 + rank, distinct=16).to(device))
    gen = torch.Generator(device="cpu").manual_seed(args.seed + rank)
    ops, _ = G.runtime()

    class PinnedRing:
        """Fixed set of pinned staging buffers for the per-train_op host draws.  The host runs many steps ahead
        of the GPU here (no .item() per train_op), and a fresh pinned allocation per draw made the first ~15
        iterations 10-25 % slower (hipHostMalloc while the queue is full); a buffer is reused only after the
        event recorded behind its H2D copy has completed."""
        def __init__(self, shape, depth=48):
            self.bufs = [torch.empty(*shape, pin_memory=True) for _ in range(depth)]
            self.evs = [None] * depth
            self.i = 0

        def draw(self, fill, dst):
            """fill a pinned buffer on the host, copy it (asynchronously, stream-ordered) into the PERSISTENT device tensor
            `dst`: the step graphs read that tensor in place (graphed.mark_static), so a draw costs one H2D copy and no
            device-to-device copy into a graph's own input buffer"""
            k = self.i % len(self.bufs)
            self.i += 1
            if self.evs[k] is not None:
                self.evs[k].synchronize()
            fill(self.bufs[k])
            dst.copy_(self.bufs[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.evs[k] = ev
            return dst

    u_ring, e_ring = PinnedRing((N, 2048)), PinnedRing((1,))
    u_dev = [graphed.mark_static(torch.empty(N, 2048, device=device)) for _ in range(3)]      # one per train_op of an iteration
    eps_dev = graphed.mark_static(torch.empty(1, device=device))

    def draw_u(j):
        # U(-0.3, 0.3) on the CPU generator (src/wgan_loss.py:100), written in place into a pinned
        # buffer so that the H2D copy is asynchronous and the host can run ahead of the GPU
        return u_ring.draw(lambda b: b.uniform_(-0.3, 0.3, generator=gen), u_dev[j])

    def draw_eps():
        return e_ring.draw(lambda b: b.uniform_(0.0, 1.0, generator=gen), eps_dev)

    if args.api_path:
        one_step, api_info = api_path_step(args, G, Dm, og, od, (lg, ld, lp), device, rank)
    else:
        api_info = None

        from rna_gan_amd import losses as PL

        def one_step():
            # the three train_ops of src/wgan_loss.py:82-129,181-263,314-389 in Trainer order, through the
            # loss plugins' step() (= train_ops without the final .item() host sync): per train_op a fresh
            # uniform draw on the CPU generator and eps ~ U(0,1) for the penalty; the frozen betaVAE encodes this
            # iteration's RNA rows ONCE (first train_op) and the other two reuse the latent (new_batch() drops it, so
            # every iteration encodes although this benchmark reuses one resident RNA tensor)
            PL.new_batch()
            u_g, u_d, u_p = draw_u(0), draw_u(1), draw_u(2)    # same order of draws as three separate train_ops
            # the D-loss step is told the penalty step's draw: both fake batches (same generator weights) come out of one
            # generator pass over the double batch; the penalty step picks its fake up (losses._FAKE)
            return [lg.step(G, Dm, og, rna, u_g),
                    ld.step(G, Dm, od, real, rna, u_d, next_u=u_p),
                    lp.step(G, Dm, od, real, rna, u_p, draw_eps())]

    info = {"ops": ops, "rna_features": rna_features, "api_path": api_info,
            # the live objects of the workload (tests/test_bench_step_gpu.py compares this very step with the CPU oracle;
            # the extras below reuse the models): never serialised
            "handles": {"G": G, "D": Dm, "og": og, "od": od, "losses": (lg, ld, lp), "real": real, "rna": rna,
                        "host_generator": gen},
            "workload": "RNA-GAN lung (betaVAE-conditioned wganvae path) 256x256, %s enc2048/step64, "
                        % ("DCGAN" if args.gan_type == "dcgan" else "DCGANUpGenerator + DCGAN discriminator") +
                        "per-GPU batch %d, one iteration = G-loss + D-loss + GP steps" % N}
    return one_step, D_.flush, N, info


def api_path_step(args, G, Dm, og, od, losses, device, rank):
    """--api-path: one iteration as a user of the reference CLI gets it (src/histopathology_gan.py:298-314): the
    Trainer's per-batch body -- real_inputs = the loader's dict batch, then every loss's train_ops resolved by
    argument name, each ending in .item() -- fed with a FRESH uint8 tile batch per iteration (pinned host memory ->
    H2D as uint8 -> rg_u8_to_norm on the device, the a15 input contract of :106-109)."""
    import rna_gan_amd as P
    from rna_gan_amd import synth as R
    N = args.batch
    pool = [R.synthetic_tiles_u8(N, 256, seed=1234 + rank + 17 * k).pin_memory() for k in range(4)]
    rna = R.synthetic_rna(N, 19198, seed=4321 + rank, distinct=16).pin_memory()
    trainer = P.Trainer.__new__(P.Trainer)       # the loop body only: models / optimizers are the ones built above
    trainer.device = device
    trainer.generator, trainer.discriminator = G, Dm
    trainer.optimizer_generator, trainer.optimizer_discriminator = og, od
    trainer.losses = {type(l).__name__: l for l in losses}
    trainer.loss_logs = {name: [] for name in trainer.losses}
    trainer.loss_information = {"generator_losses": 0.0, "discriminator_losses": 0.0, "generator_iters": 0,
                                "discriminator_iters": 0}
    trainer.ncritic = 1
    trainer.batch_size = N
    trainer.labels = None
    trainer._store_loss_maps()
    ops, _ = G.runtime()
    k = [0]

    def one_step():
        u8 = pool[k[0] % len(pool)].to(device, non_blocking=True)
        k[0] += 1
        trainer.real_inputs = {"image": ops.u8_to_norm(u8), "rna_data": rna, "labels": None}
        trainer.train_iter()
        return [torch.tensor(trainer.loss_logs[name][-1]) for name in trainer.losses]

    return one_step, {"host_syncs_per_iteration": 3, "input": "uint8 tiles, pinned host -> device, normalised by "
                                                              "rg_u8_to_norm", "loop": "Trainer.train_iter"}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))

    # stdout carries exactly ONE line (the JSON record): everything else this process or the libraries it loads write
    # to file descriptor 1 (RCCL prints a version banner there when a communicator is created) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from rna_gan_amd import dist as D_
    global _DEFAULT_THREADS
    _DEFAULT_THREADS = torch.get_num_threads()
    torch.set_num_threads(min(8, _DEFAULT_THREADS))   # GPU leg: the host only draws noise
    D_.set_sync_stats(args.sync_stats)
    use_cuda = torch.cuda.is_available()
    D_.init_from_env()
    rank, world = D_.rank(), D_.world_size()
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s): launch with torchrun "
                         "--nproc-per-node %d, or run `python bench.py --gpus %d` (it starts the ranks itself)"
                         % (args.gpus, world, args.gpus, args.gpus))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if use_cuda and os.environ.get("RNAGAN_DIST_BACKEND") == "gloo":
        local %= torch.cuda.device_count()        # functional multi-rank runs on fewer devices than ranks (see dist.init_from_env)
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)

    if args.step_plugin:
        mod, fn = args.step_plugin.split(":")
        one_step, flush, N, info = getattr(importlib.import_module(mod), fn)(args, rank, world, device)
        args.no_roofline = args.no_cpu_baseline = args.no_extras = True
    else:
        one_step, flush, N, info = hip_workload(args, rank, world, device)

    def sync():
        if use_cuda:
            torch.cuda.synchronize(device)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        sync()

    # HIP-graph capture phase (part of building the step, like a compiler's first run): the loss plugins run a
    # train_op eagerly twice per launch-sequence variant and capture it on the third call; the very first
    # iteration uses a different variant (all bf16 weight images stale), so the replayed graphs are in place
    # after ~6 iterations (5 graphs: the G step is captured for two staleness variants).  Once back-to-back
    # replay starts, the chip needs another ~10-20 iterations to settle (steps run 15-25 % slower during that
    # transient, see --trace-steps; it moves with the start of continuous load, not with the iteration count), so
    # 40 priming iterations (~0.7 s) are run before the W warm-up steps.
    from rna_gan_amd import graphed as _gr
    prime = 0 if args.step_plugin else (40 if _gr.ENABLED else 1)
    log("workload built; %d priming iterations (graph capture)" % prime)
    for it in range(prime):
        one_step()
        if os.environ.get("RNAGAN_GRAPH_DEBUG"):
            sync(); log("priming iteration %d done" % it)
    barrier()
    log("warm-up (%d steps)" % args.warmup)
    for _ in range(args.warmup):
        one_step()
    barrier()
    log("warm-up done; timing %d steps" % args.steps)
    evs = []
    from rna_gan_amd import losses as _PL
    enc0 = _PL._LATENT.misses
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ls = one_step()
        if args.trace_steps and use_cuda:
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    flush()               # data parallel: the last train_op's all-reduce + optimizer step belong to the timed work
    barrier()
    dt = time.perf_counter() - t0
    if args.trace_steps and rank == 0 and evs:
        log("per-step ms (GPU events): " + " ".join("%.2f" % evs[i - 1].elapsed_time(evs[i]) for i in range(1, len(evs))))
    last_losses = [float(l.item()) for l in ls] if ls is not None else None
    if not args.step_plugin:
        from rna_gan_amd.ops_hip import check_handoffs
        check_handoffs()          # a timed-out in-kernel rendezvous (fused split-K BatchNorm) invalidates the measurement
    out_graphs = bool(_gr.ENABLED) and not args.step_plugin
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = N * world * args.steps / dt
    log("timed region: %.3f ms/step, %.1f imgs/s" % (ms_per_step, value))

    backend = torch.distributed.get_backend() if world > 1 or (torch.distributed.is_available() and
                                                                torch.distributed.is_initialized()) else None
    out = {
        "metric": "training imgs/sec (G+D WGAN-GP step, 256x256)",
        "value": round(value, 2), "unit": "imgs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": info.get("workload", "plug-in step (test hook)"),
                   "global_batch": N * world, "parallelism": "dp%d" % world,
                   "ranks": world, "collective_backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend),
                   "rna_features": info.get("rna_features"),
                   "losses_last_step": last_losses, "hip_graphs": out_graphs and not D_.sync_stats(),
                   "dp_statistics": "global (sync-stats)" if D_.sync_stats() else "rank-local (DDP)"},
    }
    if D_.active():
        ops_h = "f16" if args.precision == "fp16" else "bf16"
        out["config"]["dp_route"] = _PL.DP_ROUTE
        out["config"]["dp_wire"] = "fp32" if (args.precision == "fp32" or not D_.COMPRESS or (ops_h == "f16" and not D_.F16_WIRE)) else ops_h
    if not args.step_plugin:
        # frozen-betaVAE encodes per timed iteration: 1 (the three plugins share one encode per batch, losses._LatentCache;
        # the reference encodes three times, src/wgan_loss.py:96-97, :223-224, :353-354)
        out["config"]["betavae_encodes_per_iteration"] = round((_PL._LATENT.misses - enc0) / max(args.steps, 1), 3)
    if info.get("api_path"):
        out["config"]["api_path"] = info["api_path"]
    if args.step_plugin:
        out["config"]["step_plugin"] = args.step_plugin

    # the instrumented extra iteration contains collectives in a data-parallel run: every rank runs it
    roof = None if args.no_roofline or args.api_path else measure_roofline(info["ops"], device, one_step, ms_per_step,
                                                                              extra_ops=[m.runtime()[0] for m in (info["handles"]["G"], info["handles"]["D"])
                                                                                         if m.runtime()[0] is not info["ops"]])
    flush()
    if roof is not None and rank == 0 and not args.no_ceilings:
        attach_ceilings(roof, device)
    if world > 1:
        torch.distributed.barrier()       # (the other ranks wait while rank 0 measures its ceilings)
    extras = parity_gpu = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = measure_extras(args, device, info)
        parity_gpu = parity_gpu_leg(args, device)
    if rank == 0:
        if not args.step_plugin:
            fl = conv_flops_per_image()
            total_flops_img = 5 * fl["G"] + 14 * fl["D"]        # 104.6 GFLOP / image / iteration
            out["config"]["algorithmic_conv_tflops_whole_step"] = round(total_flops_img * value / world / 1e12, 2)
        # BASELINE.json's metric also names FID@10k; the north star's "FID within 2.0 of the CPU reference" clause
        out["fid"] = {"status": "blocked: no Inception-v3 weights offline",
                      "detail": "src/fid.py:33-63 loads torchvision inception_v3(pretrained=True); the weights are not in this image and "
                                "there is no network.  Built and tested without them: the Frechet distance (tests/golden F8, the reference's "
                                "calculate_frechet_distance), the 299 x 299 preprocessing / protocol, the Inception-v3 extractor on HIP with "
                                "pluggable weights (rna_gan_amd/inception.py vs oracle/inception_ref.py on random weights).  "
                                "rna_gan_amd.fid.fid_protocol(generate_fake, real_images, inception_feature_extractor(<torchvision inception_v3 state_dict>)) "
                                "runs the README protocol (5 generations, mean +- std) once weights exist.",
                      "value": None}
        if roof is not None:
            out["roofline"] = roof
        if extras is not None:
            out["config"]["extras"] = extras
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], first = cpu_baseline(args.seed)
            if parity_gpu is not None:
                out["parity"] = parity_object(parity_gpu, first, args.precision)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    barrier()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def measure_roofline(ops, device, one_step, step_ms, extra_ops=()):
    """Per-launch HIP events (recorded on the stream the kernels are enqueued on) around every conv
    launch of ONE extra, instrumented iteration of the same workload, run right after the timed
    region.  Kernel families:
      conv_fwd_dgrad = gather_gemm_kernel (bf16 MFMA implicit GEMM; every stride-2 conv / transposed
                       conv, forward, data-gradient and GP tangent pass of both networks)
      conv_wgrad     = wgrad_kernel + its slab reduction
      conv_wgrad_adam = the weight-gradient launches that also apply the Adam step of their tensor (rg_conv_wgrad_adam: the two
                       33.5 M-parameter layers; their interval is dominated by the optimizer's 26 bytes per parameter)
    `achieved` = algorithmic FLOPs of the family's launches / their summed duration (SURVEY 8d:
    2*N*Ho*Wo*Cout*Cin*16 per launch); a launch's duration is its event interval minus the interval an empty event
    pair reads (calibrated just before; reported as event_pair_overhead_us) and includes the split-K slab reduction
    where one follows.  The dominant family (largest share of the iteration) is the
    one reported; the other is listed under `others`."""
    from rna_gan_amd import graphed
    was = graphed.ENABLED
    graphed.ENABLED = False            # per-launch events need eager launches (not a graph replay)
    # what an empty (record, record) pair reads on this stream: subtracted from every launch's interval
    stream = torch.cuda.current_stream(device)
    cal = []
    for _ in range(32):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); b.record(stream)
        cal.append((a, b))
    ops.timing = []
    for o in extra_ops:                # (the discriminator's optimizer launches its fused weight-gradient + Adam kernel through the
        o.timing = ops.timing          # discriminator module's own HipOps: same list)
    one_step()
    torch.cuda.synchronize(device)
    graphed.ENABLED = was
    overhead_ms = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2]
    fam, stacks = {}, {}
    for key, flops, e0, e1, owner in ops.timing:
        ms = max(e0.elapsed_time(e1) - overhead_ms, 0.0)
        f = fam.setdefault(key, {"launches": 0, "flops": 0.0, "ms": 0.0})
        f["launches"] += 1
        f["flops"] += flops
        f["ms"] += ms
        if key == "conv_fwd_dgrad" and owner in ("G", "D"):
            # the same launches split by the network whose layer they run: the discriminator's conv stack (forward, data
            # gradient, the penalty's tangent and joint reverse passes: the stack BASELINE.json's north_star names) / the generator's
            g = stacks.setdefault(owner, {"launches": 0, "flops": 0.0, "ms": 0.0})
            g["launches"] += 1
            g["flops"] += flops
            g["ms"] += ms
    ops.timing = None
    for o in extra_ops:
        o.timing = None
    rows = {}
    for k, f in fam.items():
        rows[k] = {"launches": f["launches"], "ms_total": round(f["ms"], 3),
                   "avg_us_per_launch": round(f["ms"] * 1e3 / max(f["launches"], 1), 1),
                   "tflops": round(f["flops"] / (f["ms"] * 1e-3) / 1e12, 1) if f["ms"] > 0 else None,
                   "share_of_step": round(f["ms"] / step_ms, 3)}
    dom = max((k for k in rows if k != "bn_split_fused"), key=lambda k: rows[k]["ms_total"])
    names = {"conv_fwd_dgrad": "conv8_kernel + convp_kernel + convd_kernel + gather_gemm_dma_kernel (bf16 MFMA implicit-GEMM / patch-resident conv: fwd / dgrad / tangent)",
             "conv_wgrad": "wgrad8_kernel / wgrad_dma_kernel (bf16 MFMA weight gradient) + reduce_slabs_kernel"}
    pmc_fam = {"conv_fwd_dgrad": "gather_gemm", "conv_wgrad": "wgrad_dma"}.get(dom)
    traffic, traffic_src = pmc_traffic(pmc_fam)
    util, util_src = pmc_mfma_util(pmc_fam)
    # Where the split-K launches' slab reduction is timed: round 2 reduced the fp32 slabs in a kernel of the conv op itself
    # (inside this family's intervals); now the BatchNorm kernel that consumes the conv output reduces them (slab_bn_*,
    # rg_splitbn.hip: reduction + statistics + hand-off + apply in one launch), listed separately under others.bn_split_fused.
    # achieved_incl_split_bn charges the WHOLE of those fused kernels (their BatchNorm work included) to the conv family: a
    # lower bound of the round-2 way of counting.
    incl = None
    if dom == "conv_fwd_dgrad" and "bn_split_fused" in rows and rows[dom]["ms_total"] > 0:
        fl = fam[dom]["flops"]
        incl = round(fl / ((fam[dom]["ms"] + fam["bn_split_fused"]["ms"]) * 1e-3) / 1e12, 1)
    return {"bound": "mfma", "kernel": names.get(dom, dom), "achieved": rows[dom]["tflops"],
            "achieved_incl_split_bn": incl,
            "split_k_reduction": "in the consuming BatchNorm kernel (others.bn_split_fused), not in this family's intervals",
            "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(rows[dom]["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src,
            "mfma_util": util, "mfma_util_source": util_src,
            "event_pair_overhead_us": round(overhead_ms * 1e3, 2),
            "launches": rows[dom]["launches"], "avg_us_per_launch": rows[dom]["avg_us_per_launch"],
            "share_of_step": rows[dom]["share_of_step"],
            "d_stack": stack_row(stacks.get("D")), "g_stack": stack_row(stacks.get("G")),
            "others": {k: v for k, v in rows.items() if k != dom}}


def attach_ceilings(roof, device):
    """roofline.peak_measured: the on-box ceilings (rna_gan_amd/probe.py, rg_probe.hip) -- bare bf16 MFMA loops on random
    register operands, the product's own 8-wave conv k-loop fed from LDS-resident stages (no operand has to arrive: what the
    schedule can yield), a float4 stream copy -- measured in this process behind the timed region, and the conv fractions
    against the second of them (`frac_of_measured`).  SURVEY 8d / BASELINE.md 4."""
    from rna_gan_amd import probe
    try:
        pm = probe.measure_ceilings(device)
    except Exception as e:                                   # a failed probe never takes the headline down
        roof["peak_measured"] = {"error": repr(e)[:200]}
        return
    roof["peak_measured"] = pm
    # the product's conv launches use v_mfma_f32_16x16x32_bf16 (option conv8_mfma = 16)
    loop = pm.get("conv8_loop_lds_fed_16x16x32_tflops")
    bare = max(v for k, v in pm.items() if k.startswith("mfma_bare_"))
    roof["peak_measured_what"] = ("frac_of_measured = achieved / conv8_loop_lds_fed_16x16x32_tflops (the product's 8-wave k-loop over "
                                  "LDS-resident stages: its schedule's ceiling on this box, random operands); frac_of_bare_mfma = "
                                  "achieved / the best bare MFMA loop")
    if loop:
        roof["frac_of_measured"] = round(roof["achieved"] / loop, 4)
        roof["frac_of_bare_mfma"] = round(roof["achieved"] / bare, 4)
        for k in ("d_stack", "g_stack"):
            if roof.get(k):
                roof[k]["frac_of_measured"] = round(roof[k]["achieved"] / loop, 4)
                roof[k]["frac_of_bare_mfma"] = round(roof[k]["achieved"] / bare, 4)
        wg = roof.get("others", {}).get("conv_wgrad")
        if wg and wg.get("tflops"):
            wg["frac_of_measured"] = round(wg["tflops"] / loop, 4)


def stack_row(g):
    """One network's share of the conv family (measure_roofline): algorithmic FLOPs of its MFMA conv launches / their summed
    HIP-event durations, as a fraction of the bf16 peak."""
    if not g or g["ms"] <= 0:
        return None
    tf = g["flops"] / (g["ms"] * 1e-3) / 1e12
    return {"what": "bf16 MFMA conv launches of this network's layers (forward, data gradient, tangent): algorithmic FLOPs / "
                    "summed HIP-event time", "launches": g["launches"], "ms_total": round(g["ms"], 3),
            "achieved": round(tf, 1), "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4)}


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC summary (two separate --pmc passes,
    FETCH_SIZE doubled per the gfx950 correction; tools/pmc_traffic.py).  Counters cannot be read from inside
    this process, so the figure is the one measured on this workload (batch 64) when the profile was taken."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((p for p in (os.path.join(here, "round%d_pmc_hbm_traffic.json" % r) for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(p)),
                os.path.join(here, "round3_pmc_hbm_traffic.json"))
    try:
        with open(path) as f:
            row = json.load(f)[family]
        return row["hbm_bytes_per_launch"], "profiles/%s (%d launches)" % (os.path.basename(path), row["launches"])
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


def pmc_mfma_util(family):
    """Time-weighted MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / elapsed cycles over the 4 x 256 SIMDs) of a kernel family
    from the committed rocprofv3 --pmc summary of this workload (tools/pmc_bench.sh + tools/pmc_family.py)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((p for p in (os.path.join(here, "round%d_mfma_util.json" % r) for r in (6, 5, 4, 3, 2)) if os.path.exists(p)),
                os.path.join(here, "round3_mfma_util.json"))
    try:
        with open(path) as f:
            return json.load(f)[family]["mfma_util"], "profiles/" + os.path.basename(path)
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


def cpu_baseline(seed):
    """The oracle on the host cores: batch 8 (the reference's hard-coded batch size,
    src/histopathology_gan.py:94), fp32, the wganvae path INCLUDING the three frozen-betaVAE encodes of an iteration
    (src/wgan_loss.py:96-106).  The box's best intra-op thread count is found first (one iteration each over a few
    candidates: torch's default of all SMT threads is far from optimal on this host), then 3 iterations are timed at
    that setting.  Bounded to ~40 s."""
    import torch.nn as nn
    from oracle import ref_cpu as R
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    n, rna_features = 8, 19198
    G = R.seeded_fill_(R.OracleDCGANGenerator(2048, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2),
                                              last_nonlinearity=nn.Tanh()), seed).train()
    D = R.seeded_fill_(R.OracleDCGANDiscriminator(256, 3, 64, nonlinearity=nn.LeakyReLU(0.2),
                                                  last_nonlinearity=nn.LeakyReLU(0.2)), seed + 1).train()
    vae = R.seeded_fill_(R.OracleBetaVAE(rna_features, 2048, [6000, 4000, 2048], [4000, 6000]), seed + 2).eval()
    og, od = R.make_adam(G.parameters(), 1e-4), R.make_adam(D.parameters(), 4e-4)
    real = R.synthetic_images(n, 256, seed=1234)
    rna = R.synthetic_rna(n, rna_features, seed=4321, distinct=16)

    losses = {}

    def one(it):
        us = [R.synthetic_uniform(n, 2048, seed=10 * it + j) for j in range(3)]
        t0 = time.perf_counter()
        with torch.no_grad():
            noises = [R.conditioned_noise(u, R.encode_latent(vae, rna)) for u in us]      # one encode per train_op
        losses[it] = R.train_iteration(G, D, og, od, real, noises, PARITY_EPS)
        return time.perf_counter() - t0

    cands = [c for c in (8, 16, 32, 64) if c <= avail] or [max(1, avail)]
    torch.set_num_threads(cands[0])
    one(0)                                            # warm-up (allocator, oneDNN primitive caches); = the parity iteration
    trial = {}
    for i, c in enumerate(cands):
        torch.set_num_threads(c)
        trial[c] = one(1 + i)
        log("cpu baseline calibration: %d threads -> %.2f s/iteration" % (c, trial[c]))
        if trial[c] > 2.5 * min(trial.values()):
            break
    best = min(trial, key=trial.get)
    torch.set_num_threads(best)
    times = [one(10 + k) for k in range(3)]
    t = sum(times) / len(times)
    log("cpu baseline: %d threads, %.2f s/iteration" % (best, t))
    first = [losses[0]["g"], losses[0]["d"], losses[0]["gp"]]
    # kind: the bench contract's enum is "reference" | "port"; "port" = the ORACLE (oracle/ref_cpu.py, the pinned CPU
    # restatement: one process, this image's torch CPU kernels) -- not the reference's own scripts at its torch 1.10
    return {"value": round(n / t, 3), "unit": "imgs/sec", "cores": best, "kind": "port", "implementation": "oracle",
            "threads": "best of %s intra-op thread counts tried, %d host threads available" % (sorted(trial), avail),
            "sample": "oracle/ref_cpu.py: PyTorch fp32 restatement of the reference path incl. the 3 frozen-betaVAE encodes per "
                      "iteration, batch 8, 3 timed iterations after a warm-up, %.2f s/iteration, torch %s"
                      % (t, torch.__version__)}, first


PARITY_EPS = 0.5          # the interpolation coefficient of the parity iteration (both legs)
PARITY_TOL = {"bf16": 6e-2, "fp32": 2e-3, "fp16": 2e-2}       # |gpu - cpu| <= tol * (|cpu| + 0.1), as tests/test_train_gpu.py states it


def parity_gpu_leg(args, device):
    """The first iteration of cpu_baseline()'s run (batch 8; weights seeded seed / seed+1 / seed+2, tiles seed 1234, RNA
    seed 4321, uniform draws seeds 0..2, eps 0.5) through the HIP product path: the three loss plugins' step() on a fresh
    model set.  Returns the three losses (floats)."""
    from rna_gan_amd import losses as PL
    from rna_gan_amd import synth as R
    n, rna_features = 8, 19198
    G, Dm, og, od, (lg, ld, lp) = build(device, args.precision, n, rna_features, args.seed)
    real = R.synthetic_images(n, 256, seed=1234).to(device)
    rna = R.synthetic_rna(n, rna_features, seed=4321, distinct=16).to(device)
    us = [R.synthetic_uniform(n, 2048, seed=j).to(device) for j in range(3)]
    eps = torch.tensor([PARITY_EPS], dtype=torch.float32, device=device)
    PL.new_batch()
    ls = [lg.step(G, Dm, og, rna, us[0]), ld.step(G, Dm, od, real, rna, us[1], next_u=us[2]),
          lp.step(G, Dm, od, real, rna, us[2], eps)]
    out = [float(l.item()) for l in ls]
    PL.new_batch()
    log("parity (GPU leg, batch 8, first iteration): %s" % out)
    return out


def parity_object(gpu, cpu, precision):
    tol = PARITY_TOL[precision]
    err = [abs(g - c) / (abs(c) + 0.1) for g, c in zip(gpu, cpu)]
    return {"what": "first iteration (G-loss, D-loss, gradient-penalty train_ops) of the batch-8 cpu_baseline run repeated on "
                    "the HIP path: same seeded weights, tiles, RNA rows, uniform draws and eps",
            "losses_gpu": [round(v, 6) for v in gpu], "losses_cpu_oracle": [round(v, 6) for v in cpu],
            "rel_err": [round(e, 5) for e in err], "tolerance": tol, "tolerance_form": "|gpu - cpu| <= tol * (|cpu| + 0.1)",
            "ok": bool(max(err) <= tol), "gpu_precision": precision, "cpu_precision": "fp32"}


def measure_extras(args, device, info):
    """Secondary figures, measured after the headline timed region on the same box in the same process."""
    ex = {}
    for name, fn in (("generate", extra_generate), ("api_path", extra_api_path), ("fp32_step", extra_fp32_step),
                     ("fp16_step", extra_fp16_step), ("dcgan_up", extra_dcgan_up),
                     ("vae_train", extra_vae_train), ("enc200", extra_enc200),
                     ("batch128", lambda a, d, i: extra_batch(a, d, i, 128)),
                     ("batch256", lambda a, d, i: extra_batch(a, d, i, 256))):
        try:
            ex[name] = fn(args, device, info)
            log("extra %s: %s" % (name, json.dumps(ex[name])))
        except Exception as e:                      # an extra never takes the headline record down with it
            ex[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            log("extra %s FAILED: %s" % (name, ex[name]["error"]))
    return ex


def extra_generate(args, device, info, n=4096, chunk=512):
    """BASELINE configs[4]: generator-only synthesis of 4096 samples from random 2048-d latents, eval-mode BatchNorm
    (running statistics, folded into the conv epilogues -- SURVEY 8d states the mode), images written into one device-resident
    (N, 3, 256, 256) fp32 buffer; fp8 e4m3 weights / activations on the layers that have an fp8 kernel, and bf16."""
    from rna_gan_amd import gan_utils as GU
    G = info["handles"]["G"]
    fl = conv_flops_per_image()["G"]
    gen = torch.Generator(device="cpu").manual_seed(args.seed + 7)
    noise = torch.randn(n, 2048, generator=gen).to(device)
    out = torch.empty((n, 3, 256, 256), dtype=torch.float32, device=device)
    res = {"samples": n, "chunk": chunk, "batchnorm": "eval mode (running statistics folded into the conv epilogues)",
           "output": "fp32 NCHW resident on the device"}
    for tag, fp8 in (("fp8", True), ("bf16", False)):
        GU.synthesize(G, noise[:2 * chunk], chunk=chunk, fp8=fp8, out=out[:2 * chunk])        # warm-up (operand images)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        GU.synthesize(G, noise, chunk=chunk, fp8=fp8, out=out)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        res[tag] = {"imgs_per_sec": round(n / dt, 1), "ms_total": round(dt * 1e3, 2),
                    "tflops": round(fl * n / dt / 1e12, 1), "finite": bool(torch.isfinite(out[::257]).all())}
    return res


def extra_api_path(args, device, info, steps=20, warm=12):
    """The drop-in API as a user of the reference CLI gets it (see api_path_step): ms per iteration and imgs/sec."""
    h = info["handles"]
    step, api_info = api_path_step(args, h["G"], h["D"], h["og"], h["od"], h["losses"], device, 0)
    for _ in range(warm):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    return dict(api_info, ms_per_step=round(dt * 1e3, 3), imgs_per_sec=round(args.batch / dt, 1), steps=steps)


def extra_vae_train(args, device, info, steps=10, warm=3):
    """one betaVAE TRAINING iteration at the reference's full size (src/betaVAE_training.py defaults:
    19198 genes, [6000, 4000, 2048] / [4000, 6000], batch 64): forward, loss, backward, fused Adam on the HIP GEMM kernels."""
    import rna_gan_amd as P
    from rna_gan_amd import synth as R
    from rna_gan_amd import vae_train as VT
    N = args.batch
    dims = (19198, 2048, [6000, 4000, 2048], [4000, 6000])
    m = P.betaVAE(*dims, beta=2.0)
    R.seeded_fill_(m, 51)
    m = m.set_precision("bf16").to(device).train()
    opt = P.Adam(m.parameters(), lr=3e-3, weight_decay=1e-4).bind(m, fuse_linear_wgrad=os.environ.get("VAE_FUSE", "1") != "0")
    gen = torch.Generator(device="cpu").manual_seed(args.seed + 13)
    x = torch.tanh(torch.randn(N, dims[0], generator=gen)).to(device)
    nparam = sum(p.numel() for p in m.parameters())

    def step():
        opt.zero_grad(set_to_none=True)
        out, mu, lv = m(x)
        losses = VT.betaVAEloss(x, out, mu, lv, m.beta, training=True)
        losses["total_loss"].backward()
        opt.step()
        return losses
    for _ in range(warm):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    floor_bytes = nparam * (28 + 4 + 3 * 2)      # Adam 28 B + fp32 weight gradient written once + weights read 3x as bf16
    res = {"ms_per_step": round(dt * 1e3, 3), "samples_per_sec": round(N / dt, 1), "parameters_M": round(nparam / 1e6, 1),
           "precision": "bf16", "hbm_floor_TBps_achieved": round(floor_bytes / dt / 1e12, 2), "steps": steps,
           "loss": round(float(losses["total_loss"].detach()), 5)}
    del m, opt
    torch.cuda.empty_cache()
    return res


def extra_enc200(args, device, info, steps=10, warm=12, enc=200):
    """BASELINE.json's north_star speaks of "random 200-d conditioning latents"; the reference's code uses 2048
    (src/histopathology_gan.py:179, src/wgan_loss.py:67), which SURVEY 0.4 makes the benchmark.  This is the 200-d data point:
    the same iteration (wganvae plugins, batch 64, bf16, step graphs) with encoding_dims = z_dim = 200 -- only G.0
    (200 x 2048 x 4 x 4 instead of 2048 x ...) and the betaVAE's last encoder layer / z_mu (4000 -> 200 -> 200) change."""
    from rna_gan_amd import losses as PL
    N = args.batch
    G, Dm, og, od, (lg, ld, lp) = build(device, "bf16", N, 19198, args.seed, enc=enc)
    h = info["handles"]
    gen = torch.Generator(device="cpu").manual_seed(args.seed + 17)

    def it():
        PL.new_batch()
        us = [torch.empty(N, enc).uniform_(-0.3, 0.3, generator=gen).to(device) for _ in range(3)]
        eps = torch.empty(1).uniform_(0.0, 1.0, generator=gen).to(device)
        return [lg.step(G, Dm, og, h["rna"], us[0]), ld.step(G, Dm, od, h["real"], h["rna"], us[1], next_u=us[2]),
                lp.step(G, Dm, od, h["real"], h["rna"], us[2], eps)]
    for _ in range(warm):
        it()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        ls = it()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    PL.new_batch()
    res = {"encoding_dims": enc, "ms_per_step": round(dt * 1e3, 3), "imgs_per_sec": round(N / dt, 1), "steps": steps,
           "generator_parameters_M": round(sum(p.numel() for p in G.parameters()) / 1e6, 1),
           "losses": [round(float(l.item()), 5) for l in ls],
           "note": "host draws not pinned / pre-staged as in the headline loop: an upper bound of the step time"}
    del G, Dm, og, od, lg, ld, lp
    torch.cuda.empty_cache()
    return res


def extra_batch(args, device, info, batch, steps=6, warm=14):
    """The SAME iteration at a larger per-GPU batch, with the same roofline object: separates kernel quality from launch
    granularity.  At batch 64 a conv launch is 68.7 GFLOP = 27 us at peak on 256 CUs -- fill, drain, the split-K slabs of
    the deep layers and the epilogue are a fixed cost per launch; at batch 128 / 256 the same kernels run 2 x / 4 x the work
    per launch (fewer or no split-K launches).  Own models / plug-ins / step graphs; never part of `value`."""
    import copy
    from rna_gan_amd import losses as PL
    a2 = copy.copy(args)
    a2.batch = batch
    one_step, flush, n, inf = hip_workload(a2, 0, 1, device)
    for _ in range(warm):               # graph capture (two eager runs per launch-sequence variant) + settling
        one_step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        ls = one_step()
    flush()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    roof = measure_roofline(inf["ops"], device, one_step, dt * 1e3,
                            extra_ops=[m.runtime()[0] for m in (inf["handles"]["G"], inf["handles"]["D"])
                                       if m.runtime()[0] is not inf["ops"]])
    flush()
    torch.cuda.synchronize(device)
    res = {"batch": batch, "ms_per_step": round(dt * 1e3, 3), "imgs_per_sec": round(batch / dt, 1), "steps": steps,
           "losses": [round(float(l.item()), 5) for l in ls],
           "roofline": {k: roof[k] for k in ("achieved", "achieved_incl_split_bn", "peak", "unit", "frac", "launches",
                                             "avg_us_per_launch", "share_of_step", "d_stack", "g_stack")},
           "conv_wgrad": roof["others"].get("conv_wgrad"), "bn_split_fused": roof["others"].get("bn_split_fused")}
    PL.new_batch()
    del one_step, flush, inf
    torch.cuda.empty_cache()
    return res


F32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X fp32 matrix (= vector) peak, MI355X_MICROARCH.md


def extra_fp16_step(args, device, info, steps=10, warm=14):
    """BASELINE.json configs[3] names fp16 storage: the SAME iteration (same workload, batch, graphs) on the fp16 build of the
    library (librnagan_hip_f16.so: IEEE fp16 activations / operand images / slabs, v_mfma_f32_*_f16, fp32 accumulation and
    masters) with the static loss scale on the backward seeds (default 4096 = 64 x 64: seed scale x tangent scale in the
    penalty step) that rna_gan_amd.optim.Adam removes inside its kernels."""
    from rna_gan_amd import graphed
    from rna_gan_amd import losses as PL
    N = args.batch
    G, Dm, og, od, (lg, ld, lp) = build(device, "fp16", N, 19198, args.seed)
    h = info["handles"]
    gen = torch.Generator(device="cpu").manual_seed(args.seed + 13)

    def it():
        PL.new_batch()
        us = [torch.empty(N, 2048).uniform_(-0.3, 0.3, generator=gen).to(device) for _ in range(3)]
        eps = torch.empty(1).uniform_(0.0, 1.0, generator=gen).to(device)
        return [lg.step(G, Dm, og, h["rna"], us[0]), ld.step(G, Dm, od, h["real"], h["rna"], us[1], next_u=us[2]),
                lp.step(G, Dm, od, h["real"], h["rna"], us[2], eps)]
    for _ in range(warm):
        it()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        ls = it()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    ops, _ = G.runtime()
    res = {"ms_per_step": round(dt * 1e3, 3), "imgs_per_sec": round(N / dt, 1), "steps": steps, "hip_graphs": bool(graphed.ENABLED),
           "losses": [round(float(l.item()), 5) for l in ls], "loss_scale": ops.loss_scale,
           "penalty_seed_scale": ops.gp_seed_scale, "penalty_tangent_scale": ops.gp_tangent_scale,
           "finite": bool(all(torch.isfinite(p).all() for p in list(G.parameters()) + list(Dm.parameters()))),
           "kernels": "librnagan_hip_f16.so: the bf16 step's kernels compiled for IEEE fp16 storage (v_mfma_f32_16x16x32_f16 / "
                      "32x32x16_f16), fp32 accumulation, statistics and masters"}
    PL.new_batch()
    del G, Dm, og, od, lg, ld, lp, it
    torch.cuda.empty_cache()
    return res


def extra_dcgan_up(args, device, info, steps=8, warm=12):
    """The same iteration with the reference's OTHER generator, src/dcgan.py's DCGANUpGenerator (resize-convolution blocks:
    bilinear x2 + ReflectionPad2d(1) + Conv3x3, src/dcgan.py:45-56,76-84; `--gan_type dcgan_up`; the reference CLI never selects
    it): bf16, batch 64.  Its upsample + pad image is materialised once per block (uppad_bf16_kernel) and read by a 9-tap MFMA conv."""
    from rna_gan_amd import graphed
    from rna_gan_amd import losses as PL
    N = args.batch
    G, Dm, og, od, (lg, ld, lp) = build(device, "bf16", N, 19198, args.seed, gan_type="dcgan_up")
    h = info["handles"]
    gen = torch.Generator(device="cpu").manual_seed(args.seed + 17)

    def it():
        PL.new_batch()
        us = [torch.empty(N, 2048).uniform_(-0.3, 0.3, generator=gen).to(device) for _ in range(3)]
        eps = torch.empty(1).uniform_(0.0, 1.0, generator=gen).to(device)
        return [lg.step(G, Dm, og, h["rna"], us[0]), ld.step(G, Dm, od, h["real"], h["rna"], us[1], next_u=us[2]),
                lp.step(G, Dm, od, h["real"], h["rna"], us[2], eps)]
    for _ in range(warm):
        it()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        ls = it()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    res = {"ms_per_step": round(dt * 1e3, 3), "imgs_per_sec": round(N / dt, 1), "steps": steps, "hip_graphs": bool(graphed.ENABLED),
           "losses": [round(float(l.item()), 5) for l in ls], "generator": "DCGANUpGenerator (92.3 M parameters)", "precision": "bf16"}
    PL.new_batch()
    del G, Dm, og, od, lg, ld, lp, it
    torch.cuda.empty_cache()
    return res


def _fp32_variant(args, device, info, planes, f32mma, steps, warm, want_roofline):
    """One timed fp32 workload: planes = RNAGAN_F32_PLANES (6 / 3 bf16 products per fp32 product on planes split once per tensor;
    0 = the per-tile kernels of rg_generic.hip under option f32mma)."""
    from rna_gan_amd import graphed, _abi
    from rna_gan_amd import losses as PL
    N = args.batch
    lib = _abi.load()
    old_env = os.environ.get("RNAGAN_F32_PLANES")
    os.environ["RNAGAN_F32_PLANES"] = str(planes)
    if f32mma is not None:
        _abi.check(lib.rg_set_option(b"f32mma", f32mma), "rg_set_option")
    try:
        G, Dm, og, od, (lg, ld, lp) = build(device, "fp32", N, 19198, args.seed)
        h = info["handles"]
        gen = torch.Generator(device="cpu").manual_seed(args.seed + 11)

        def it():
            PL.new_batch()
            us = [torch.empty(N, 2048).uniform_(-0.3, 0.3, generator=gen).to(device) for _ in range(3)]
            eps = torch.empty(1).uniform_(0.0, 1.0, generator=gen).to(device)
            return [lg.step(G, Dm, og, h["rna"], us[0]), ld.step(G, Dm, od, h["real"], h["rna"], us[1], next_u=us[2]),
                    lp.step(G, Dm, od, h["real"], h["rna"], us[2], eps)]
        for _ in range(warm):           # graph capture: two eager runs per launch-sequence variant, replay from the third on
            it()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            ls = it()
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / steps
        res = {"ms_per_step": round(dt * 1e3, 2), "imgs_per_sec": round(N / dt, 1), "steps": steps,
               "hip_graphs": bool(graphed.ENABLED), "losses": [round(float(l.item()), 5) for l in ls]}
        if want_roofline:
            # per-launch events over one eager iteration (same method as the bf16 roofline)
            ops, _ = G.runtime()
            was = graphed.ENABLED
            graphed.ENABLED = False
            try:
                it(); torch.cuda.synchronize(device)            # eager warm-up (workspace growth)
                stream = torch.cuda.current_stream(device)
                cal = []
                for _ in range(16):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream); b.record(stream)
                    cal.append((a, b))
                ops.timing = []
                it()
                torch.cuda.synchronize(device)
                overhead = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2]
                fam = {}
                for key, flops, e0, e1, _owner in ops.timing:
                    f = fam.setdefault(key, [0, 0.0, 0.0])
                    f[0] += 1; f[1] += flops; f[2] += max(e0.elapsed_time(e1) - overhead, 0.0)
            finally:
                ops.timing = None
                graphed.ENABLED = was
            mm = [v for k, v in fam.items() if k in ("conv_fwd_dgrad", "conv_wgrad")]
            if mm:
                fl, ms = sum(v[1] for v in mm), sum(v[2] for v in mm)
                ach = fl / (ms * 1e-3) / 1e12
                prod = planes if planes else 6
                res["roofline_fp32"] = {
                    "bound": "mfma", "kernel": "conv8_kernel / wgrad8_kernel over K-concatenated bf16 planes (rg_conv8f.hip, rg_wgrad8f.hip): conv fwd / "
                                               "dgrad / tangent + weight gradients, fp32 operands, accumulation and results",
                    "achieved": round(ach, 1), "peak": round(MFMA_BF16_PEAK_TFLOPS / prod, 1), "unit": "TFLOP/s",
                    "peak_what": "nominal bf16 matrix peak / %d bf16 products per fp32 product (the executing unit's ceiling)" % prod,
                    "frac": round(ach / (MFMA_BF16_PEAK_TFLOPS / prod), 4),
                    "bf16_equivalent_tflops": round(ach * prod, 1),
                    "peak_f32_instruction": F32_MATRIX_PEAK_TFLOPS, "frac_of_f32_instruction_peak": round(ach / F32_MATRIX_PEAK_TFLOPS, 4),
                    "launches": sum(v[0] for v in mm), "ms_total": round(ms, 2), "share_of_step": round(ms / (dt * 1e3), 3),
                    "families": {k: {"launches": v[0], "ms": round(v[2], 2),
                                     "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 1) if v[2] > 0 else None} for k, v in fam.items()}}
        PL.new_batch()
        del G, Dm, og, od, lg, ld, lp, it
        return res
    finally:
        if f32mma is not None:
            lib.rg_set_option(b"f32mma", -1)
        if old_env is None:
            os.environ.pop("RNAGAN_F32_PLANES", None)
        else:
            os.environ["RNAGAN_F32_PLANES"] = old_env
        torch.cuda.empty_cache()


def extra_fp32_step(args, device, info, steps=4, warm=10):
    """The fp32 parity mode (--precision fp32: the reference's own arithmetic, src/betaVAE.py:184,223,230-236; the anchor of the
    tight-tolerance parity tests) on the same workload.  Default since round 6: fp32 tensors are split ONCE into three bf16
    planes (v = h + m + l exactly) and the stride-2 convs / transposed convs / weight gradients run the bf16 step's 8-wave
    kernels over K-concatenated plane pairs with fp32 accumulation -- 6 bf16 products per fp32 product (fp32-grade: the tolerance
    statistics of the f32 instruction).  `planes3`: the 3-product tier (2^-16 per product; its own tolerance table in
    profiles/).  `per_tile_kernels`: round 5's default (operands split per tile inside gemm_bf16x3s_kernel)."""
    res = _fp32_variant(args, device, info, 6, None, steps, warm, True)
    res["kernels"] = ("fp32 activations; 4 x 4 convs / weight gradients: conv8_kernel / wgrad8_kernel over bf16 planes split once per tensor, "
                      "6 bf16 matrix-core products per fp32 product (RNAGAN_F32_PLANES=6, the default); G.0, image-side layers, head: "
                      "f32-instruction kernels")
    for name, planes, f32mma in (("planes3", 3, None), ("per_tile_kernels", 0, 2)):
        try:
            res[name] = _fp32_variant(args, device, info, planes, f32mma, steps, warm, name == "planes3")
        except Exception as e:                                   # an extra never takes the headline down
            res[name] = {"error": repr(e)[:200]}
    res["planes3"]["kernels"] = "RNAGAN_F32_PLANES=3: three bf16 products per fp32 product (hh + hm + mh)"
    res["per_tile_kernels"]["kernels"] = "RNAGAN_F32_PLANES=0, f32mma=2: round 5's default (three-way split per tile inside gemm_bf16x3s_kernel)"
    return res


if __name__ == "__main__":
    main()
