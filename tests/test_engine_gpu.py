"""GPU parity of the whole explicit training-step gradient computation (rna_gan_amd.engine on
HipOps) against the autograd oracle on CPU (oracle/ref_cpu.py), same seeded weights/inputs."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
from rna_gan_amd import engine as E


def mk(in_size, step, enc, seed=5):
    G = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                              last_nonlinearity=nn.Tanh()), seed)
    D = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                  last_nonlinearity=nn.LeakyReLU(0.2)), seed + 1)
    return G, D


def relerr(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    assert torch.isfinite(a).all()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def cmp_grads(mod_gpu, mod_cpu, tol, what):
    worst = 0.0
    for (k, p), (_, q) in zip(mod_gpu.named_parameters(), mod_cpu.named_parameters()):
        e = relerr(p.grad, q.grad)
        worst = max(worst, e)
        assert e <= tol, f"{what}: grad {k} rel err {e:.3e} > {tol:.1e}"
    return worst


def cmp_bufs(mod_gpu, mod_cpu, tol, what):
    for (k, p), (_, q) in zip(mod_gpu.named_buffers(), mod_cpu.named_buffers()):
        if k.endswith("num_batches_tracked"):
            assert int(p.cpu()) == int(q), f"{what}: {k}"
        else:
            e = relerr(p, q)
            assert e <= tol, f"{what}: buffer {k} rel err {e:.3e}"


CASES = [
    # in_size, step, enc, batch, dtype, tol_g, tol_gp
    (16, 4, 24, 5, torch.float32, 2e-4, 2e-3),
    (32, 4, 16, 3, torch.float32, 2e-4, 2e-3),
    (32, 64, 128, 4, torch.float32, 3e-4, 3e-3),
    (32, 64, 128, 8, torch.bfloat16, 6e-2, 1.5e-1),    # MFMA kernels on the conv stack
    (64, 64, 128, 4, torch.bfloat16, 6e-2, 1.5e-1),
]


@pytest.mark.parametrize("in_size,step,enc,n,dtype,tol,tol_gp", CASES)
def test_three_steps_vs_autograd(in_size, step, enc, n, dtype, tol, tol_gp):
    from rna_gan_amd.ops_hip import HipOps
    ops = HipOps(dtype, "cuda:0")
    G, D = mk(in_size, step, enc)
    Gg, Dg = copy.deepcopy(G).cuda(), copy.deepcopy(D).cuda()
    for m in (G, D, Gg, Dg):
        m.train()
    real = R.synthetic_images(n, in_size, seed=3)
    noise = R.synthetic_normal(n, enc, seed=4)
    Gn, Dn = E.build_gen_net(Gg), E.build_disc_net(Dg)
    real_d, noise_d = real.cuda(), noise.cuda()

    loss_o = R.generator_loss(D(G(noise))); loss_o.backward()
    loss_e = E.gen_loss_grads(ops, Gn, Dn, noise_d)
    assert abs(float(loss_e.cpu()) - float(loss_o)) <= tol * (abs(float(loss_o)) + 1e-3), "G loss"
    cmp_grads(Gg, G, tol, "G step"); cmp_bufs(Gg, G, tol, "G step"); cmp_bufs(Dg, D, tol, "G step")

    for p in D.parameters():
        p.grad = None
    loss_o = R.discriminator_loss(D(real), D(G(noise).detach())); loss_o.backward()
    loss_e = E.disc_loss_grads(ops, Gn, Dn, real_d, noise_d)
    assert abs(float(loss_e.cpu()) - float(loss_o)) <= tol * (abs(float(loss_o)) + 1e-3), "D loss"
    cmp_grads(Dg, D, tol, "D step"); cmp_bufs(Dg, D, tol, "D step")

    for p in D.parameters():
        p.grad = None
    eps = 0.3
    xhat = eps * real + (1 - eps) * G(noise)
    gp = R.gradient_penalty(xhat, D(xhat)); (10.0 * gp).backward()
    loss_e = E.gp_loss_grads(ops, Gn, Dn, real_d, noise_d, eps, 10.0)
    assert abs(float(loss_e.cpu()) - float(gp)) <= tol_gp * (abs(float(gp)) + 1e-3), "GP value"
    cmp_grads(Dg, D, tol_gp, "GP step"); cmp_bufs(Dg, D, tol, "GP step"); cmp_bufs(Gg, G, tol, "GP step")
