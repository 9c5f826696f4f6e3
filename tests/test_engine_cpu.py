"""The explicit (autograd-free) forward/backward/GP sequencing of rna_gan_amd.engine, driven by
the torch twin ops (oracle/ops_ref.py), must reproduce the autograd oracle.  CPU only: this
validates the ALGORITHM (incl. the second-order gradient-penalty pass); the HIP kernels are
checked against the same twins in the gpu tests."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import ref_cpu as R
from oracle.ops_ref import RefOps
from rna_gan_amd import engine as E


def mk(in_size, step, enc, seed=5):
    G = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                              last_nonlinearity=nn.Tanh()), seed)
    D = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                  last_nonlinearity=nn.LeakyReLU(0.2)), seed + 1)
    return G.double(), D.double()


def grads_of(mod):
    return {k: p.grad.clone() for k, p in mod.named_parameters()}


def bufs_of(mod):
    return {k: b.clone() for k, b in mod.named_buffers()}


def assert_close_dict(a, b, rtol, atol):
    assert a.keys() == b.keys()
    for k in a:
        np.testing.assert_allclose(a[k].double().numpy(), b[k].double().numpy(), rtol=rtol, atol=atol, err_msg=k)


@pytest.mark.parametrize("in_size,step,enc,n,tap_major", [(16, 4, 24, 5, False), (32, 4, 16, 3, False),
                                                          (32, 4, 16, 3, True)])
def test_three_steps_match_autograd(in_size, step, enc, n, tap_major):
    torch.manual_seed(0)
    G, D = mk(in_size, step, enc)
    G2, D2 = copy.deepcopy(G), copy.deepcopy(D)
    for m in (G, D, G2, D2):
        m.train()
    if tap_major:      # the product's storage order for the middle conv weights; parameters keep their shape
        E.tap_major_(G2), E.tap_major_(D2)
        assert any(E.is_tap_major(p.data) for p in D2.parameters())
    real = R.synthetic_images(n, in_size, seed=3).double()
    noise = R.synthetic_normal(n, enc, seed=4).double()
    ops = RefOps(torch.float64)
    Gn, Dn = E.build_gen_net(G2), E.build_disc_net(D2)

    # ---- G step gradients
    for p in list(G.parameters()) + list(D.parameters()):
        p.grad = None
    loss_o = R.generator_loss(D(G(noise)))
    loss_o.backward()
    loss_e = E.gen_loss_grads(ops, Gn, Dn, noise)
    np.testing.assert_allclose(float(loss_e), float(loss_o), rtol=1e-9)
    assert_close_dict(grads_of(G2), grads_of(G), 1e-7, 1e-10)
    assert_close_dict(bufs_of(G2), bufs_of(G), 1e-9, 1e-12)
    assert_close_dict(bufs_of(D2), bufs_of(D), 1e-9, 1e-12)

    # ---- D step gradients
    for p in D.parameters():
        p.grad = None
    loss_o = R.discriminator_loss(D(real), D(G(noise).detach()))
    loss_o.backward()
    loss_e = E.disc_loss_grads(ops, Gn, Dn, real, noise)
    np.testing.assert_allclose(float(loss_e), float(loss_o), rtol=1e-9)
    assert_close_dict(grads_of(D2), grads_of(D), 1e-7, 1e-10)
    assert_close_dict(bufs_of(D2), bufs_of(D), 1e-9, 1e-12)
    # the data-parallel split of the same step (prefix = D(real) forward + backward, rest = the fake half accumulated):
    # same gradients; the BatchNorm buffers see the two extra forward calls, which is all that may differ
    want = {k: v.clone() for k, v in grads_of(D2).items()}
    for p in D2.parameters():
        p.grad.zero_()
    out_r = E.disc_loss_prefix_bwd(ops, Dn, real)
    loss_s = E.disc_loss_rest_acc(ops, Gn, Dn, out_r, noise)
    np.testing.assert_allclose(float(loss_s), float(loss_e), rtol=1e-12)
    assert_close_dict(grads_of(D2), want, 1e-9, 1e-12)
    D2.load_state_dict(D.state_dict()); G2.load_state_dict(G.state_dict())      # undo the extra running-statistics updates

    # ---- GP step gradients (second order)
    for p in D.parameters():
        p.grad = None
    eps = 0.3
    xhat = eps * real + (1 - eps) * G(noise)
    gp = R.gradient_penalty(xhat, D(xhat))
    (10.0 * gp).backward()
    loss_e = E.gp_loss_grads(ops, Gn, Dn, real, noise, eps, 10.0)
    np.testing.assert_allclose(float(loss_e), float(gp), rtol=1e-8)
    assert_close_dict(grads_of(D2), grads_of(D), 1e-6, 1e-9)
    assert_close_dict(bufs_of(D2), bufs_of(D), 1e-9, 1e-12)
    assert_close_dict(bufs_of(G2), bufs_of(G), 1e-9, 1e-12)


def test_up_generator_step_matches_autograd():
    """Resize-convolution generator (src/dcgan.py DCGANUpGenerator): the engine's autograd-free G step on the fp64
    twin against torch autograd on the oracle modules (which tests/test_oracle_golden.py pins to the reference)."""
    torch.manual_seed(0)
    G = R.OracleDCGANUpGenerator(16, 32, 3, 4).double()
    D = R.OracleDCGANDiscriminator(32, 3, 4).double()
    R.seeded_fill_(G, 7), R.seeded_fill_(D, 8)
    G2, D2 = copy.deepcopy(G), copy.deepcopy(D)
    for m in (G, D, G2, D2):
        m.train()
    noise = R.synthetic_normal(5, 16, seed=4).double()
    ops = RefOps(torch.float64)
    Gn, Dn = E.build_upgen_net(G2), E.build_disc_net(D2)
    assert isinstance(Gn, E.UpGenNet)
    loss_o = R.generator_loss(D(G(noise)))
    loss_o.backward()
    loss_e = E.gen_loss_grads(ops, Gn, Dn, noise)
    np.testing.assert_allclose(float(loss_e), float(loss_o), rtol=1e-9)
    assert_close_dict(grads_of(G2), grads_of(G), 1e-6, 1e-10)
    assert_close_dict(bufs_of(G2), bufs_of(G), 1e-9, 1e-12)
    # D step / GP step only use the generator's forward
    real = R.synthetic_images(5, 32, seed=3).double()
    for p in D.parameters():
        p.grad = None
    loss_o = R.discriminator_loss(D(real), D(G(noise).detach()))
    loss_o.backward()
    loss_e = E.disc_loss_grads(ops, Gn, Dn, real, noise)
    np.testing.assert_allclose(float(loss_e), float(loss_o), rtol=1e-9)
    assert_close_dict(grads_of(D2), grads_of(D), 1e-6, 1e-10)


def test_penalty_parts_input_gradients_and_eval_discriminator():
    """Pieces behind the public functional forms: (a) engine.disc_gp_first / disc_gp_second with accumulate=True and
    need_input_grad=True reproduce torch's double backward of lambd * (||d sum D(x)/dx|| - 1)^2 with respect to the
    parameters (added to existing gradients) AND to x; (b) the generator's input gradient; (c) eval-mode D."""
    torch.manual_seed(0)
    G, D = mk(32, 4, 16)
    G2, D2 = copy.deepcopy(G), copy.deepcopy(D)
    for m in (G, D, G2, D2):
        m.train()
    ops = RefOps(torch.float64)
    Gn, Dn = E.build_gen_net(G2), E.build_disc_net(D2)
    x = R.synthetic_images(4, 32, seed=3).double().requires_grad_(True)
    # (a) autograd reference: gradient of 7 * penalty wrt parameters and x
    gp = R.gradient_penalty(x, D(x))
    (7.0 * gp).backward()
    want_x = x.grad.clone()
    for p in D2.parameters():
        p.grad.fill_(0.5)                                   # accumulate semantics: gradients are ADDED
    _, ctx = E.disc_forward(ops, Dn, x.detach().clone())
    loss, (g, v) = E.disc_gp_first(ops, Dn, ctx, 1.0)
    gx = E.disc_gp_second(ops, Dn, ctx, (g, 7.0 * v), accumulate=True, need_input_grad=True)
    np.testing.assert_allclose(float(loss), float(gp), rtol=1e-9)
    np.testing.assert_allclose(gx.numpy(), want_x.numpy(), rtol=1e-6, atol=1e-9)
    assert_close_dict({k: v_ - 0.5 for k, v_ in grads_of(D2).items()}, grads_of(D), 1e-6, 1e-9)
    # (b) d/dz of the generator
    z = R.synthetic_normal(4, 16, seed=4).double().requires_grad_(True)
    G(z).square().sum().backward()
    img, gctx = E.gen_forward(ops, Gn, z.detach().clone())
    gin = E.gen_backward(ops, Gn, gctx, 2.0 * img, accumulate=False, need_input_grad=True)
    np.testing.assert_allclose(gin.numpy(), z.grad.numpy(), rtol=1e-7, atol=1e-10)
    # (c) eval-mode discriminator (running statistics)
    D.eval()
    with torch.no_grad():
        want = D(x.detach())
    got = E.disc_forward_eval(ops, Dn, x.detach().clone())
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("in_size,step,enc,n", [(16, 4, 24, 5), (32, 4, 16, 4)])
def test_batched_d_step_matches_autograd(in_size, step, enc, n):
    """engine.disc_loss_grads_batched (D(real) and D(fake) as one double batch through the conv layers, BatchNorm per half)
    against the autograd oracle: loss, every parameter gradient, and the running statistics (updated by the real half
    first, as the reference's call order does)."""
    torch.manual_seed(0)
    G, D = mk(in_size, step, enc)
    G2, D2 = copy.deepcopy(G), copy.deepcopy(D)
    for m in (G, D, G2, D2):
        m.train()
    real = R.synthetic_images(n, in_size, seed=3).double()
    noise = R.synthetic_normal(n, enc, seed=4).double()
    ops = RefOps(torch.float64)
    Gn, Dn = E.build_gen_net(G2), E.build_disc_net(D2)
    for p in D.parameters():
        p.grad = None
    loss_o = R.discriminator_loss(D(real), D(G(noise).detach()))
    loss_o.backward()
    loss_e = E.disc_loss_grads_batched(ops, Gn, Dn, real, noise)
    np.testing.assert_allclose(float(loss_e), float(loss_o), rtol=1e-9)
    assert_close_dict(grads_of(D2), grads_of(D), 1e-7, 1e-10)
    assert_close_dict(bufs_of(D2), bufs_of(D), 1e-9, 1e-12)
    assert_close_dict(bufs_of(G2), bufs_of(G), 1e-9, 1e-12)


def test_generator_forward_pair_equals_two_forwards():
    """engine.gen_forward_pair (two noise batches as one double batch, BatchNorm per half) against two consecutive
    gen_forward calls: images and running statistics (updated by the first half first)."""
    torch.manual_seed(0)
    G, _ = mk(32, 4, 16)
    G2 = copy.deepcopy(G)
    G.train(); G2.train()
    ops = RefOps(torch.float64)
    Gn, Gn2 = E.build_gen_net(G), E.build_gen_net(G2)
    za, zb = R.synthetic_normal(3, 16, seed=4).double(), R.synthetic_normal(3, 16, seed=5).double()
    ia, _ = E.gen_forward(ops, Gn, za, keep=False)
    ib, _ = E.gen_forward(ops, Gn, zb, keep=False)
    pair = E.gen_forward_pair(ops, Gn2, torch.cat([za, zb]))
    np.testing.assert_allclose(pair[:3].numpy(), ia.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(pair[3:].numpy(), ib.numpy(), rtol=1e-10, atol=1e-12)
    assert_close_dict(bufs_of(G2), bufs_of(G), 1e-10, 1e-13)
