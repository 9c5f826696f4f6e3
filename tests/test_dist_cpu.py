"""Data-parallel semantics on CPU (gloo, world_size 2): the product's DP glue (rna_gan_amd.dist:
grad_scale folded into the backward seed + SUM all-reduce of the flat gradient buffer) applied to the
engine reproduces "run the single-process gradient computation on each shard, average the gradients"
(SURVEY 8e parity definition).  The engine runs on the torch twin ops here (no GPU in this container)."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import ref_cpu as R


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _mk():
    G = R.seeded_fill_(R.OracleDCGANGenerator(16, 16, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                                              last_nonlinearity=nn.Tanh()), 3)
    D = R.seeded_fill_(R.OracleDCGANDiscriminator(16, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                                                  last_nonlinearity=nn.LeakyReLU(0.2)), 4)
    return G.double().train(), D.double().train()


def _shard_inputs(rank, n=4):
    real = R.synthetic_images(n, 16, seed=50 + rank).double()
    noise = R.synthetic_normal(n, 16, seed=60 + rank).double()
    return real, noise, 0.2 + 0.5 * rank


def _worker(rank, world, port, q, sync=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      RNAGAN_SYNC_STATS="1" if sync else "0")
    import torch.distributed as dist
    from oracle.ops_ref import RefOps
    from rna_gan_amd import dist as D_, engine as E
    from rna_gan_amd.models import FlatParams
    torch.set_num_threads(1)
    D_.init_from_env(backend="gloo")
    assert D_.world_size() == world and D_.rank() == rank
    G, D = _mk()
    flat_g, flat_d = FlatParams(G), FlatParams(D)
    Gn, Dn = E.build_gen_net(G), E.build_disc_net(D)
    ops = D_.attach_sync(RefOps(torch.float64))
    assert (ops.stat_reduce is not None) == sync
    real, noise, eps = _shard_inputs(rank)
    if sync:
        eps = 0.3                       # one interpolation weight for the whole (global) batch, as in the reference
    out = {}
    E.gen_loss_grads(ops, Gn, Dn, noise, grad_scale=D_.grad_scale())
    D_.allreduce_sum_(flat_g.grad); out["G"] = flat_g.grad.clone()
    E.disc_loss_grads(ops, Gn, Dn, real, noise, grad_scale=D_.grad_scale())
    D_.allreduce_sum_(flat_d.grad); out["D"] = flat_d.grad.clone()
    E.gp_loss_grads(ops, Gn, Dn, real, noise, eps, 10.0, grad_scale=D_.gp_grad_scale())
    D_.allreduce_sum_(flat_d.grad); out["P"] = flat_d.grad.clone()
    out["rm"] = torch.cat([b.reshape(-1).double() for b in D.buffers()])
    q.put((rank, {k: v.numpy() for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _flat_grads(mod):
    return torch.cat([p.grad.reshape(-1) for p in mod.parameters()])


def test_dp2_gloo_matches_shard_average():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # oracle: independent autograd per shard (each on a fresh replica: BN buffers are rank-local), averaged
    acc = {"G": 0, "D": 0, "P": 0}
    for r in range(world):
        G, D = _mk()
        real, noise, eps = _shard_inputs(r)
        R.generator_loss(D(G(noise))).backward(); acc["G"] = acc["G"] + _flat_grads(G) / world
        for p in D.parameters():
            p.grad = None
        R.discriminator_loss(D(real), D(G(noise).detach())).backward(); acc["D"] = acc["D"] + _flat_grads(D) / world
        for p in D.parameters():
            p.grad = None
        xhat = eps * real + (1 - eps) * G(noise)
        (10.0 * R.gradient_penalty(xhat, D(xhat))).backward(); acc["P"] = acc["P"] + _flat_grads(D) / world
    for r in range(world):
        for k in ("G", "D", "P"):
            n = acc[k].numel()
            np.testing.assert_allclose(res[r][k][:n], acc[k].numpy(), rtol=1e-6, atol=1e-9, err_msg=f"rank{r} {k}")
    for k in ("G", "D", "P"):
        np.testing.assert_array_equal(res[0][k], res[1][k])


def test_dp2_sync_stats_matches_single_process_global_batch():
    """--sync-stats: two ranks with 4 samples each and synchronised statistics reproduce the gradients of
    the SINGLE-process reference step on the concatenated batch of 8 -- whole-batch BatchNorm (forward, backward and
    the penalty's second-order pass) and the whole-batch penalty norm -- to fp64 round-off."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    G, D = _mk()
    real = torch.cat([_shard_inputs(r)[0] for r in range(world)])
    noise = torch.cat([_shard_inputs(r)[1] for r in range(world)])
    eps = 0.3
    want = {}
    R.generator_loss(D(G(noise))).backward(); want["G"] = _flat_grads(G)
    for p in D.parameters():
        p.grad = None
    R.discriminator_loss(D(real), D(G(noise).detach())).backward(); want["D"] = _flat_grads(D)
    for p in D.parameters():
        p.grad = None
    xhat = eps * real + (1 - eps) * G(noise)
    (10.0 * R.gradient_penalty(xhat, D(xhat))).backward(); want["P"] = _flat_grads(D)
    want_rm = torch.cat([b.reshape(-1).double() for b in D.buffers()])
    for r in range(world):
        for k in ("G", "D", "P"):
            n = want[k].numel()
            np.testing.assert_allclose(res[r][k][:n], want[k].numpy(), rtol=1e-6, atol=1e-9, err_msg=f"rank{r} {k}")
        np.testing.assert_allclose(res[r]["rm"], want_rm.numpy(), rtol=1e-9, atol=1e-12, err_msg="BN buffers")


def _handoff_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from rna_gan_amd import dist as D_, ops_hip as H
    torch.set_num_threads(1)
    D_.init_from_env(backend="gloo")

    class FakeOps:                       # what check_handoffs reads of a backend: its device and the error word
        device = torch.device("cpu")

        def __init__(self):
            self._sb_sync = torch.zeros(4, dtype=torch.int32)
    ops = FakeOps()
    H._LIVE_OPS.add(ops)
    res = {}
    H.check_handoffs()                   # clean on every rank: no error
    if rank == world - 1:
        ops._sb_sync[0] = 1              # ONE rank's in-kernel rendezvous timed out
    try:
        H.check_handoffs()
        res["raised"] = False
    except RuntimeError as e:
        res["raised"] = "timed out" in str(e)
    res["rearmed"] = int(ops._sb_sync[0]) == 0
    H.check_handoffs()                   # re-armed: clean again (and still collective: no rank hangs)
    # the sync-free poll of the Trainer: same semantics
    h = H.handoffs_poll_start()
    H.handoffs_poll_finish(h)
    if rank == 0:
        ops._sb_sync[0] = 7
    h = H.handoffs_poll_start()
    try:
        H.handoffs_poll_finish(h)
        res["poll_raised"] = False
    except RuntimeError:
        res["poll_raised"] = True
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_check_handoffs_is_rank_collective():
    """the error word of the fused split-K BatchNorm kernels is MAX-all-reduced, so EVERY rank raises when
    one rank's launch timed out (rank 0 must not checkpoint weights that absorbed that rank's gradients, and no rank may be
    left waiting in the next collective); three ranks over gloo, the word set on the last / the first rank only."""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_handoff_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r] == {"raised": True, "rearmed": True, "poll_raised": True}, (r, got[r])


def test_dp_wire_table_tiles_the_travelling_part_of_the_flat_buffer():
    """Host logic of the data-parallel wire-direct route (losses._dp_wire_table): what the weight-gradient launches of a pass left
    on the layer handles -- split-K slabs (buffer, nsplit, dtype) or a slice already written to the wire (None, -1, 0) -- becomes
    a segment table that tiles flat[head:] in order, plain ranges in between; layers that left nothing stay plain; the handles
    are cleared."""
    from rna_gan_amd import losses as PL

    flat = torch.zeros(10000)

    class CW:
        def __init__(self, off, n, pending):
            self.w = flat[off:off + n]
            self.pending_slabs = pending

    class Stepped:
        pass
    st = Stepped()
    st.flat = type("F", (), {"data": flat})()
    slab = torch.zeros(64)
    a, b, c = CW(1000, 2000, (slab, 8, 1)), CW(4000, 1000, None), CW(6000, 3000, (None, -1, 0))
    table = PL._dp_wire_table(st, [(a, None), (b, None), (c, None)], head=400)
    assert table == [(0, 600, 0, 0, 0), (600, 2000, slab.data_ptr(), 8, 1), (2600, 3000, 0, 0, 0), (5600, 3000, 0, -1, 0),
                     (8600, 1000, 0, 0, 0)]
    assert sum(r[1] for r in table) == flat.numel() - 400 and all(x.pending_slabs is None for x in (a, b, c))
    assert PL._dp_wire_table(st, [(a, None), (b, None)], head=0) is None            # nothing pending: the plain cast pass
    a.pending_slabs = (slab, 4, 0)
    with pytest.raises(RuntimeError):
        PL._dp_wire_table(st, [(a, None)], head=2000)                                # a deferred layer inside the factor head
