"""Full-size checks (BASELINE configs[1]: batch 64, the five 4x4 stride-2 layer shapes of the 256x256 model): the kernel
variants the benchmark actually runs -- 256x256 tiles, parity classes in the block order, XCD remapping, split-K
plans, direct weight-gradient writes, two-level statistics finishing -- only exist at these sizes, so the small
parity cases of test_ops_gpu.py never reach them.  Two independent checks per layer, all through the C ABI:
  * element-wise against the functor (generic) kernels of the same library, which are size-generic and pinned to
    the oracle at small sizes;
  * the size-independent adjoint identities of a convolution, per sample and per output channel:
        <conv_down(x; w), g>_n = <x, conv_up(g; w)>_n        <conv_down(x; w), g>_o = <wgrad(g, x), w>_o
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from rna_gan_amd import _abi
from rna_gan_amd.engine import ConvW
from rna_gan_amd.ops_hip import HipOps

N = 64
LAYERS = [(64, 128, 128), (128, 256, 64), (256, 512, 32), (512, 1024, 16), (1024, 2048, 8)]      # I, O, input size


def relmax(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("I,O,hs", LAYERS)
def test_conv_layers_full_size(I, O, hs):
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(100 + I)
    mf = HipOps(torch.bfloat16, dev, algo=_abi.ALGO_AUTO)
    ge = HipOps(torch.bfloat16, dev, algo=_abi.ALGO_GENERIC)
    w = (torch.randn(O, 4, 4, I, generator=gen) * (2.0 / (I * 16)) ** 0.5).bfloat16().float().to(dev)   # bf16-exact masters
    cw_m = ConvW(w.clone(), None, torch.full_like(w, 3.0), None, "OHWI")
    cw_g = ConvW(w.clone(), None, torch.zeros_like(w), None, "OHWI")
    x = torch.randn(N, hs, hs, I, generator=gen).bfloat16().to(dev)
    g = torch.randn(N, hs // 2, hs // 2, O, generator=gen).bfloat16().to(dev)

    y, st = mf.conv_down(x, cw_m, want_stats=True)
    assert relmax(y, ge.conv_down(x, cw_g)) < 8e-3
    mask = torch.randn(N, hs, hs, I, generator=gen).bfloat16().to(dev)
    u = mf.conv_up(g, cw_m)
    assert relmax(u, ge.conv_up(g, cw_g)) < 8e-3
    assert relmax(mf.conv_up(g, cw_m, mask, 0.2), ge.conv_up(g, cw_g, mask, 0.2)) < 8e-3
    mf.conv_wgrad(g, x, cw_m, False)
    ge.conv_wgrad(g, x, cw_g, False)
    assert relmax(cw_m.dw, cw_g.dw) < 2e-3
    dw1 = cw_m.dw.clone()
    g2, x2 = g.flip(0).contiguous(), x.flip(0).contiguous()          # second segment: the same pairs in another order
    mf.conv_wgrad2(g, x, g2, x2, cw_m, True)
    assert relmax(cw_m.dw, 3 * dw1) < 2e-3

    # INDEPENDENT reference (not this library): torch's fp32 convolutions on the host over the same bf16-rounded
    # operands -- forward / transposed convolution on a slice of samples (first, last and one from the middle, so row
    # tiles at both ends of M and every parity class are covered), weight gradient over the FULL batch for a slice of
    # 8 low-resolution-side channels at both ends of O
    import torch.nn.functional as F
    w_oihw = w.permute(0, 3, 1, 2).contiguous().cpu()
    sel = [0, N // 2 + 1, N - 1]
    xs = x[sel].float().permute(0, 3, 1, 2).cpu()
    gs = g[sel].float().permute(0, 3, 1, 2).cpu()
    y_ref = F.conv2d(xs, w_oihw, stride=2, padding=1).permute(0, 2, 3, 1)
    u_ref = F.conv_transpose2d(gs, w_oihw, stride=2, padding=1).permute(0, 2, 3, 1)
    assert relmax(y[sel].cpu(), y_ref) < 8e-3, "conv_down vs torch fp32 conv2d"
    assert relmax(u[sel].cpu(), u_ref) < 8e-3, "conv_up vs torch fp32 conv_transpose2d"
    m_ref = u_ref * torch.where(mask[sel].float().cpu() > 0, 1.0, 0.2)
    assert relmax(mf.conv_up(g, cw_m, mask, 0.2)[sel].cpu(), m_ref) < 8e-3, "conv_up (fused LeakyReLU mask) vs torch"
    xa = x.float().permute(0, 3, 1, 2).cpu()
    for o0 in (0, O - 8):
        ga = g[..., o0:o0 + 8].float().permute(0, 3, 1, 2).contiguous().cpu()
        dw_ref = torch.nn.grad.conv2d_weight(xa, (8, I, 4, 4), ga, stride=2, padding=1)       # [8][I][4][4]
        got = cw_m.dw[o0:o0 + 8].permute(0, 3, 1, 2).cpu() / 3.0                              # dw holds 3 x wgrad(g, x) now
        assert relmax(got, dw_ref) < 3e-3, "conv_wgrad vs torch fp32 conv2d_weight (channels %d..)" % o0

    # epilogue statistics (when this launch produces them): exact column sums of what was stored
    if st is not None:
        yf = y.float().reshape(-1, O)
        assert relmax(st[:, 0, :].sum(0), yf.sum(0)) < 2e-3 and relmax(st[:, 1, :].sum(0), (yf * yf).sum(0)) < 1e-4

    # adjoint identities in fp64 from the stored tensors
    yd, gd, xd, ud, wd = y.double(), g.double(), x.double(), u.double(), w.double()
    lhs_n = (yd * gd).sum(dim=(1, 2, 3))
    rhs_n = (xd * ud).sum(dim=(1, 2, 3))
    scale_n = float((yd * gd).abs().sum(dim=(1, 2, 3)).mean())
    assert float((lhs_n - rhs_n).abs().max()) < 1e-3 * scale_n, "per-sample <conv_down x, g> = <x, conv_up g>"
    lhs_o = (yd * gd).sum(dim=(0, 1, 2))
    rhs_o = (dw1.double() * wd).sum(dim=(1, 2, 3))
    scale_o = float((yd * gd).abs().sum(dim=(0, 1, 2)).mean())
    assert float((lhs_o - rhs_o).abs().max()) < 1e-3 * scale_o, "per-channel <conv_down x, g> = <wgrad(g, x), w>"


@pytest.mark.parametrize("M,C", [(64 * 128 * 128, 64), (64 * 64 * 64, 128), (64 * 32 * 32, 256), (64 * 4 * 4, 2048)])
def test_batchnorm_full_size(M, C):
    """BatchNorm + LeakyReLU forward / backward on the benchmark's row counts against plain fp32 tensor arithmetic on the
    same device (checker only)."""
    dev = torch.device("cuda:0")
    ops = HipOps(torch.bfloat16, dev)
    gen = torch.Generator(device="cpu").manual_seed(7)
    z = (torch.randn(M, C, generator=gen) * 1.5 + 0.3).bfloat16().to(dev).view(1, M, 1, C)
    ga = torch.randn(M, C, generator=gen).bfloat16().to(dev).view(1, M, 1, C)
    gam, bet = (1 + 0.1 * torch.randn(C, generator=gen)).to(dev), (0.1 * torch.randn(C, generator=gen)).to(dev)
    a, mean, invstd = ops.bn_forward(z, gam, bet, 0.2, 1e-5, 0.1)
    zf = z.float().view(M, C)
    mu, var = zf.mean(0), zf.var(0, unbiased=False)
    assert relmax(mean, mu) < 1e-4 and relmax(invstd, torch.rsqrt(var + 1e-5)) < 1e-4
    xh = (zf - mu) * torch.rsqrt(var + 1e-5)
    pre = xh * gam + bet
    assert relmax(a.view(M, C), torch.where(pre > 0, pre, 0.2 * pre)) < 8e-3
    gz, s_gy, s_gyxh = ops.bn_act_bwd(z, ga, mean, invstd, gam, bet, 0.2)
    gy = ga.float().view(M, C) * torch.where(pre > 0, 1.0, 0.2)
    assert relmax(s_gy, gy.sum(0)) < 2e-3 and relmax(s_gyxh, (gy * xh).sum(0)) < 2e-3
    ref = gam * torch.rsqrt(var + 1e-5) * (gy - gy.mean(0) - xh * (gy * xh).mean(0))
    assert relmax(gz.view(M, C), ref) < 8e-3


def test_image_side_layers_full_size():
    """The three row-staged image-side kernels at 64 x 3 x 256 x 256 (first_down, last_up, skinny weight gradient share
    ONE weight array: Conv2d(3, 64) and ConvTranspose2d(64, 3) are adjoint for the same [64][3][4][4] values):
        <first_down(x), g>_n = <x, last_up(g)>_n        <first_down(x), g>_o = <skinny_wgrad(g, x), w>_o
    plus a direct check of first_down on a crop (border rows/columns and an interior window) in fp64."""
    dev = torch.device("cuda:0")
    ops = HipOps(torch.bfloat16, dev)
    gen = torch.Generator(device="cpu").manual_seed(11)
    O, I, S = 64, 3, 256
    w = (torch.randn(O, I, 4, 4, generator=gen) * 0.2).bfloat16().float().to(dev)
    cw = ConvW(w, None)
    x = torch.randn(N, I, S, S, generator=gen).bfloat16().float().to(dev)            # bf16-exact image
    g = torch.randn(N, S // 2, S // 2, O, generator=gen).bfloat16().to(dev)
    y = ops.first_down(x, cw, None, 1.0)                                             # slope 1: the bare convolution
    u = ops.last_up(g, cw, None, False)
    dw = torch.zeros(O, I, 4, 4, device=dev)
    ops.skinny_wgrad(g, x, dw, False)
    yd, gd = y.double(), g.double()
    lhs_n, rhs_n = (yd * gd).sum(dim=(1, 2, 3)), (x.double() * u.double()).sum(dim=(1, 2, 3))
    scale_n = float((yd * gd).abs().sum(dim=(1, 2, 3)).mean())
    assert float((lhs_n - rhs_n).abs().max()) < 1e-3 * scale_n
    lhs_o, rhs_o = (yd * gd).sum(dim=(0, 1, 2)), (dw.double() * w.double()).sum(dim=(1, 2, 3))
    scale_o = float((yd * gd).abs().sum(dim=(0, 1, 2)).mean())
    assert float((lhs_o - rhs_o).abs().max()) < 1e-3 * scale_o
    # direct fp64 evaluation of y[n, ho, wo, :] at the four corners and an interior point of three samples
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    for n in (0, 31, 63):
        for ho, wo in ((0, 0), (0, 127), (127, 0), (127, 127), (60, 77)):
            patch = xp[n, :, 2 * ho:2 * ho + 4, 2 * wo:2 * wo + 4]                   # [3][4][4]
            ref = (w.double() * patch).sum(dim=(1, 2, 3))
            assert float((y[n, ho, wo].double() - ref).abs().max()) < 8e-3 * float(ref.abs().max() + 1.0)


@pytest.mark.parametrize("n", [64, 5])
def test_image_side_rows128_kernels_equal_the_general_row_kernels(n):
    """The control-flow-free forms of the three image-side kernels for 256 x 256 images (rg_skinny.hip, `skinny128`) against
    the general row-staged kernels they replace there: same LDS images, same MFMA order, so the outputs are BIT-IDENTICAL
    (first_down with / without sign bits and in the masked tangent form; last_up plain, with tanh, with the fused BatchNorm
    input and with the fused consumer pass); the weight gradient differs by the fp32 summation order over workgroups only.
    n = 5: an odd unit count per workgroup (the peeled tail) and fewer strips than workgroups."""
    dev = torch.device("cuda:0")
    ops = HipOps(torch.bfloat16, dev)
    lib = _abi.load()
    gen = torch.Generator(device="cpu").manual_seed(29)
    O, I, S = 64, 3, 256
    w = (torch.randn(O, I, 4, 4, generator=gen) * 0.2).to(dev)
    b64, b3 = (0.1 * torch.randn(O, generator=gen)).to(dev), (0.1 * torch.randn(I, generator=gen)).to(dev)
    cw = ConvW(w, None)
    x = torch.randn(n, I, S, S, generator=gen).to(dev)
    v = torch.randn(n, I, S, S, generator=gen).to(dev)
    g = torch.randn(n, S // 2, S // 2, O, generator=gen).bfloat16().to(dev)
    img = torch.tanh(torch.randn(n, I, S, S, generator=gen)).to(dev)
    mean, invstd = (0.2 * torch.randn(O, generator=gen)).to(dev), (1 + 0.3 * torch.rand(O, generator=gen)).to(dev)
    gam, bet = (1 + 0.1 * torch.randn(O, generator=gen)).to(dev), (0.1 * torch.randn(O, generator=gen)).to(dev)

    def run():
        out = {}
        a = ops.first_down(x, cw, b64, 0.2)
        out["fd_bits"] = a.clone()
        bits = getattr(a, "_rg_sign_bits", None)
        assert bits is not None, "layer 0 of the discriminator writes its sign bits at this shape"
        out["bits"] = bits.clone()
        out["fd_raw"] = ops.first_down(x, cw, None, 1.0).clone()
        out["fd_tan"] = ops.first_down_tangent(v, cw, a, 0.2).clone()
        out["lu"] = ops.last_up(g, cw, None, False).clone()
        out["lu_tanh"] = ops.last_up(g, cw, b3, True).clone()
        y = torch.empty(n, I, S, S, device=dev)
        _abi.check(lib.rg_last_up_pre(g.data_ptr(), w.data_ptr(), b3.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                      invstd.data_ptr(), gam.data_ptr(), bet.data_ptr(), 0.2, n, S // 2, S // 2, O, I, 1,
                                      ops.dt, ops.stream), "rg_last_up_pre")
        out["lu_pre"] = y
        for tag, ti in (("post_tb", img), ("post", None)):
            yy, parts = ops.last_up_post(g, cw, ti)
            out["lu_" + tag], out["parts_" + tag] = yy.clone(), parts.clone()
        dw = torch.zeros(O, I, 4, 4, device=dev)
        ops.skinny_wgrad(g, x, dw, False)
        out["dw"] = dw
        torch.cuda.synchronize()
        return out
    try:
        _abi.check(lib.rg_set_option(b"skinny128", 0), "rg_set_option")
        old = run()
        _abi.check(lib.rg_set_option(b"skinny128", 1), "rg_set_option")
        new = run()
    finally:
        lib.rg_set_option(b"skinny128", -1)
    for k in ("fd_bits", "bits", "fd_raw", "fd_tan", "lu", "lu_tanh", "lu_pre", "lu_post_tb", "lu_post"):
        assert torch.equal(old[k], new[k]), (k, float((old[k].float() - new[k].float()).abs().max()))
    for k in ("parts_post_tb", "parts_post"):
        so, sn = old[k].double().sum(0), new[k].double().sum(0)
        assert float(((so - sn).abs() / (so.abs() + 1e-3)).max()) < 1e-5, k
    scale = float(old["dw"].abs().max())
    assert float((old["dw"] - new["dw"]).abs().max()) < 2e-5 * scale


@pytest.mark.parametrize("n,hw", [(64, 128), (3, 32)])
def test_input_gradient_kernel_with_fused_consumer_pass(n, hw):
    """rg_last_up_post at the benchmark's shape (and a ragged small one): the transposed conv's output multiplied by
    1 - img^2 in the store phase equals rg_last_up followed by rg_tanh_bwd BIT FOR BIT (same fp32 products), the
    per-workgroup partial rows add up to the channel sums (rg_nchw_chan_sum) and to the squared norm (rg_sqnorm) of what
    was written, and rg_gp_coef_parts gives rg_gp_coef's loss / coefficient from them."""
    dev = torch.device("cuda:0")
    ops = HipOps(torch.bfloat16, dev)
    gen = torch.Generator(device="cpu").manual_seed(23)
    w = (torch.randn(64, 3, 4, 4, generator=gen) * 0.2).to(dev)
    cw = ConvW(w, None)
    g = torch.randn(n, hw, hw, 64, generator=gen).bfloat16().to(dev)
    img = torch.tanh(torch.randn(n, 3, 2 * hw, 2 * hw, generator=gen)).to(dev)
    plain = ops.last_up(g, cw, None, False)
    for tanh_img in (img, None):
        fused = ops.last_up_post(g, cw, tanh_img)
        assert fused is not None, "no fused kernel for the benchmark shape"
        y, parts = fused
        want = ops.tanh_bwd(plain, img) if tanh_img is not None else plain
        assert torch.equal(y, want)
        assert parts.shape[1] == 4 and parts.shape[0] == ops.lib.rg_last_up_post_blocks(n, hw, hw, 64, 3, ops.dt)
        wd = want.double()
        cs, sq = wd.sum(dim=(0, 2, 3)), float((wd * wd).sum())
        scale = float(wd.abs().sum(dim=(0, 2, 3)).max())
        assert float((parts[:, :3].double().sum(0) - cs).abs().max()) < 1e-5 * scale
        assert abs(float(parts[:, 3].double().sum()) - sq) < 1e-5 * sq
        db = torch.full((3,), 0.5, device=dev)
        ops.parts_chan_sum(parts, db, True)
        assert float((db.double() - 0.5 - cs).abs().max()) < 1e-5 * scale
        ops.parts_chan_sum(parts, db, False)
        ref = torch.zeros(3, device=dev)
        ops.nchw_chan_sum(want, ref, False)
        assert float((db - ref).abs().max()) < 1e-5 * scale
        loss, coef = ops.gp_coef_parts(parts, 10.0)
        loss0, coef0 = ops.gp_coef(ops.sqnorm(want), 10.0)
        assert abs(float(loss) - float(loss0)) <= 1e-5 * abs(float(loss0)) + 1e-7
        assert abs(float(coef) - float(coef0)) <= 1e-5 * abs(float(coef0)) + 1e-7


def test_g0_and_head_full_size():
    """G.0 (ConvTranspose2d(2048, 2048, 4) on a 1x1 input = a [64 x 2048] . [2048 x 32768] GEMM), its rank-64 weight
    gradient (268 MB fp32 output) and the discriminator head at the benchmark's sizes: element-wise against the generic
    kernels, plus <g0_fwd(z), gy> = <g0_wgrad(z, gy), w> per latent dimension."""
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(13)
    E = C = 2048
    mf = HipOps(torch.bfloat16, dev, algo=_abi.ALGO_AUTO)
    ge = HipOps(torch.bfloat16, dev, algo=_abi.ALGO_GENERIC)
    w = (torch.randn(E, C, 4, 4, generator=gen) * (1.0 / E) ** 0.5).bfloat16().float().to(dev)
    z = torch.randn(N, E, generator=gen).bfloat16().float().to(dev)
    gy = torch.randn(N, 4, 4, C, generator=gen).bfloat16().to(dev)
    y = mf.g0_fwd(z, ConvW(w, None))
    assert relmax(y, ge.g0_fwd(z, ConvW(w, None))) < 8e-3
    dw, dw_g = torch.full_like(w, 2.0), torch.zeros_like(w)
    mf.g0_wgrad(z, gy, dw, False)
    ge.g0_wgrad(z, gy, dw_g, False)
    assert relmax(dw, dw_g) < 2e-3
    lhs = (y.double() * gy.double()).sum()
    rhs = (dw.double() * w.double()).sum()
    assert abs(float(lhs - rhs)) < 1e-3 * float((y.double() * gy.double()).abs().sum()) ** 0.5 * 50
    # head: 4x4 valid conv over [64][4][4][2048] -> (64,)
    a = torch.randn(N, 4, 4, C, generator=gen).bfloat16().to(dev)
    wh = (torch.randn(1, C, 4, 4, generator=gen) * 0.01).to(dev)
    h, out = mf.head_fwd(a, ConvW(wh, None), 0.2)
    ref = (a.double().permute(0, 3, 1, 2) * wh.double()).sum(dim=(1, 2, 3))
    assert relmax(h, ref.float()) < 2e-3 and relmax(out, torch.where(ref > 0, ref, 0.2 * ref).float()) < 2e-3


@pytest.mark.parametrize("H,Cin,Cout", [(4, 1024, 512), (16, 256, 128), (64, 64, 64)])
def test_resize_convolution_full_size(H, Cin, Cout):
    """DCGANUpGenerator's blocks at batch 64: matrix-core forward / data gradient / weight gradient against the functor
    kernels (which interpolate the padded upsampled image inside their operand functors: an independent formulation)."""
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(17 + H)
    mf = HipOps(torch.bfloat16, dev, algo=_abi.ALGO_AUTO)
    ge = HipOps(torch.bfloat16, dev, algo=_abi.ALGO_GENERIC)
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=gen) * 0.1).to(dev)
    x = torch.randn(N, H, H, Cin, generator=gen).bfloat16().to(dev)
    gy = torch.randn(N, 2 * H, 2 * H, Cout, generator=gen).bfloat16().to(dev)
    cm, cg = ConvW(w, b, torch.full_like(w, 5.0)), ConvW(w, b, torch.zeros_like(w))
    assert relmax(mf.upconv3(x, cm, b), ge.upconv3(x, cg, b)) < 1e-2
    assert relmax(mf.upconv3_bwd_data(gy, cm), ge.upconv3_bwd_data(gy, cg)) < 1.5e-2
    mf.upconv3_wgrad(gy, x, cm, False)
    ge.upconv3_wgrad(gy, x, cg, False)
    assert relmax(cm.dw, cg.dw) < 4e-3


def test_resize_convolution_image_block_full_size():
    """the generator's output block (64 -> 3 channels at 128 -> 256, NCHW fp32) at batch 64"""
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(23)
    mf = HipOps(torch.bfloat16, dev, algo=_abi.ALGO_AUTO)
    ge = HipOps(torch.bfloat16, dev, algo=_abi.ALGO_GENERIC)
    H, Cin, Cout = 128, 64, 3
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=gen) * 0.1).to(dev)
    x = torch.randn(N, H, H, Cin, generator=gen).bfloat16().to(dev)
    g_img = torch.randn(N, Cout, 2 * H, 2 * H, generator=gen).to(dev)
    cm, cg = ConvW(w, b, torch.zeros_like(w)), ConvW(w, b, torch.zeros_like(w))
    assert relmax(mf.upconv3(x, cm, b, out_nchw=True), ge.upconv3(x, cg, b, out_nchw=True)) < 1e-2
    mf.upconv3_wgrad(g_img, x, cm, False, gy_nchw=True)
    ge.upconv3_wgrad(g_img, x, cg, False, gy_nchw=True)
    assert relmax(cm.dw, cg.dw) < 4e-3


@pytest.mark.parametrize("in_size,n,step", [(64, 16, 64), (256, 64, 64), (64, 16, 32)])
def test_batched_d_step_matches_two_chains(in_size, n, step):
    """engine.disc_loss_grads_batched (D(real) and D(fake) as one double batch through the conv layers, BatchNorm per half)
    against engine.disc_loss_grads (two forward / backward chains) on the HIP path: same loss, same running statistics, every
    parameter gradient within the bf16 noise of two different tile shapes (the exact equivalence is the CPU test
    test_engine_cpu.py::test_batched_d_step_matches_autograd).  step_channels 32: layer 0 has 32 channels,
    so the packed LeakyReLU sign bits (64 channels, 64 -> 128 second layer only) must not be requested."""
    import torch.nn as nn
    import rna_gan_amd as P
    from rna_gan_amd import engine as E
    from oracle import ref_cpu as R
    enc = 128
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 7)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 8)
    res = []
    for fn in (E.disc_loss_grads, E.disc_loss_grads_batched):
        G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
        D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
        G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
        G, D = G.cuda().train(), D.cuda().train()
        ops, gn = G.runtime()
        _, dn = D.runtime()
        real = R.synthetic_images(n, in_size, seed=100).cuda()
        nz = R.synthetic_normal(n, enc, seed=200).cuda()
        loss = fn(ops, gn, dn, real, nz)
        torch.cuda.synchronize()
        res.append((float(loss), {k: p.grad.detach().float().cpu() for k, p in D.named_parameters()},
                    {k: b.detach().double().cpu() for k, b in D.named_buffers()}))
    (la, ga, ba), (lb, gb, bb) = res
    assert abs(la - lb) <= 0.1 * abs(la) + 1e-3, (la, lb)
    for k in ga:
        cos = float((ga[k] * gb[k]).sum() / (ga[k].norm() * gb[k].norm() + 1e-30))
        ratio = float(gb[k].norm() / (ga[k].norm() + 1e-30))
        assert cos >= 0.97 and 0.85 <= ratio <= 1.15, (k, cos, ratio)
    for k in ba:
        np.testing.assert_allclose(bb[k].numpy(), ba[k].numpy(), rtol=2e-3, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("I,O,hs,n,groups", [(256, 512, 32, 64, 1), (512, 1024, 16, 64, 1), (1024, 2048, 8, 64, 1),
                                              (512, 1024, 16, 128, 2), (1024, 2048, 8, 128, 2)])
def test_split_k_conv_fused_into_batchnorm(I, O, hs, n, groups):
    """The deep layers run split-K at the benchmark's batch; with defer= the conv leaves its fp32 slabs and ONE kernel
    (rg_bn_forward_slabs / rg_bn_act_bwd_slabs: slab reduction + statistics + cross-workgroup hand-off + apply) replaces
    reduce_slabs + statistics pass + finisher + apply.  Against the separate-launch path of the same library (pinned by
    test_conv_layers_full_size / test_batchnorm_full_size): z and ga bit-identical (same summation order of the slabs),
    statistics to fp32 round-off, outputs to one bf16 rounding; and against plain tensor arithmetic.  Both directions
    (stride-2 conv forward + its BatchNorm; transposed conv as the data gradient + BatchNorm backward), one and two batch
    groups (a double batch = two forward calls: own statistics, running statistics updated in order).  Run twice: the
    hand-off words must be left clean for the next launch."""
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(500 + I + n)
    fu, se = HipOps(torch.bfloat16, dev), HipOps(torch.bfloat16, dev)
    se.split_bn = False
    w = (torch.randn(O, 4, 4, I, generator=gen) * (2.0 / (I * 16)) ** 0.5).bfloat16().float().to(dev)
    cw_f, cw_s = ConvW(w.clone(), None, torch.zeros_like(w), None, "OHWI"), ConvW(w.clone(), None, torch.zeros_like(w), None, "OHWI")
    x = torch.randn(n, hs, hs, I, generator=gen).bfloat16().to(dev)
    gam_o, bet_o = (1 + 0.1 * torch.randn(O, generator=gen)).to(dev), (0.1 * torch.randn(O, generator=gen)).to(dev)
    ho = hs // 2

    def bn_fwd(ops, z, run):
        rm, rv, nbt = run
        if groups == 1:
            return ops.bn_forward(z, gam_o, bet_o, 0.2, 1e-5, 0.1, rm, rv, nbt)
        return ops.bn_forward2(z, gam_o, bet_o, 0.2, 1e-5, 0.1, rm, rv, nbt)

    for rep in range(2):
        runs = [(torch.zeros(O, device=dev), torch.ones(O, device=dev), torch.zeros((), dtype=torch.int64, device=dev)) for _ in range(2)]
        zf, _ = fu.conv_down(x, cw_f, want_stats=True, defer=groups)
        assert getattr(zf, "_rg_slabs", None) is not None, "this shape is expected to take the split-K slab path"
        af, mean_f, inv_f = bn_fwd(fu, zf, runs[0])
        assert getattr(zf, "_rg_slabs", None) is None
        zs, st = se.conv_down(x, cw_s, want_stats=True, defer=groups)
        assert getattr(zs, "_rg_slabs", None) is None
        as_, mean_s, inv_s = bn_fwd(se, zs, runs[1])
        torch.cuda.synchronize()
        assert torch.equal(zf.view(torch.int16), zs.view(torch.int16)), "z = bf16(sum of the slabs in split order)"
        assert relmax(mean_f, mean_s) < 2e-6 and relmax(inv_f, inv_s) < 2e-6
        assert relmax(af, as_) < 8e-3 and float((af.float() - as_.float()).abs().mean()) < 1e-5
        assert relmax(runs[0][0], runs[1][0]) < 1e-5 and relmax(runs[0][1], runs[1][1]) < 1e-5
        assert int(runs[0][2]) == int(runs[1][2]) == groups
        # independent arithmetic on the stored z
        for h in range(groups):
            zh = zf.float().view(groups, -1, O)[h]
            mu, var = zh.mean(0), zh.var(0, unbiased=False)
            assert relmax(mean_f.view(groups, O)[h], mu) < 1e-4 and relmax(inv_f.view(groups, O)[h], torch.rsqrt(var + 1e-5)) < 1e-4
            pre = (zh - mu) * torch.rsqrt(var + 1e-5) * gam_o + bet_o
            assert relmax(af.view(groups, -1, O)[h], torch.where(pre > 0, pre, 0.2 * pre)) < 8e-3

        # forward-mode tangent of the same block (the penalty's tangent forward): zt = conv(tangent) arrives as slabs
        if groups == 1:
            xt = torch.randn(n, hs, hs, I, generator=gen).bfloat16().to(dev)
            zt_f = fu.conv_down(xt, cw_f, defer=1)
            assert getattr(zt_f, "_rg_slabs", None) is not None
            at_f, t1_f, t2_f = fu.bn_tangent(zf, zt_f, mean_f, inv_f, gam_o, bet_o, 0.2)
            zt_s = se.conv_down(xt, cw_s, defer=1)
            at_s, t1_s, t2_s = se.bn_tangent(zs, zt_s, mean_s, inv_s, gam_o, bet_o, 0.2)
            torch.cuda.synchronize()
            assert torch.equal(zt_f.view(torch.int16), zt_s.view(torch.int16))
            # (column sums of signed values: compared on the scale of the summands, sqrt(rows) * |zt|)
            scale = float(zt_s.float().abs().mean()) * (zt_s.numel() / O) ** 0.5
            assert float((t1_f - t1_s).abs().max()) < 1e-4 * scale and float((t2_f - t2_s).abs().max()) < 1e-4 * scale
            assert relmax(at_f, at_s) < 8e-3 and float((at_f.float() - at_s.float()).abs().mean()) < 1e-5 * float(at_s.float().abs().mean() + 1)

        # backward: ga = conv_up(g) (data gradient of the Conv2d) arrives as slabs, BatchNorm backward of the layer below
        g = torch.randn(n, ho, ho, O, generator=gen).bfloat16().to(dev)
        zb = (torch.randn(n, hs, hs, I, generator=gen) * 1.3 + 0.2).bfloat16().to(dev)          # the lower layer's z
        gam_i, bet_i = (1 + 0.1 * torch.randn(I, generator=gen)).to(dev), (0.1 * torch.randn(I, generator=gen)).to(dev)
        res = []
        for ops, cw in ((fu, cw_f), (se, cw_s)):
            if groups == 1:
                _, mean, inv = ops.bn_forward(zb.clone(), gam_i, bet_i, 0.2, 1e-5, 0.1)
            else:
                _, mean, inv = ops.bn_forward2(zb.clone(), gam_i, bet_i, 0.2, 1e-5, 0.1)
            ga = ops.conv_up(g, cw, defer=groups)
            if ops is fu:          # (the 512 -> 256 transposed conv fills the chip without split-K at batch 64: plain path)
                split = fu.lib.rg_conv_split(1, n, ho, ho, O, I, fu.dt, fu.algo) > 1
                assert (getattr(ga, "_rg_slabs", None) is not None) == split
            dg, db = torch.full((I,), 2.0, device=dev), torch.full((I,), -1.0, device=dev)
            if groups == 1:
                gz, s1, s2 = ops.bn_act_bwd(zb, ga, mean, inv, gam_i, bet_i, 0.2, dg, db, True, keep_ga=True)
            else:
                gz = ops.bn_act_bwd2(zb, ga, mean, inv, gam_i, bet_i, 0.2, dg, db, True)
                s1 = s2 = None
            torch.cuda.synchronize()
            res.append((ga, gz, s1, s2, dg, db, mean, inv))
        (ga_f, gz_f, s1_f, s2_f, dg_f, db_f, mean_b, inv_b), (ga_s, gz_s, s1_s, s2_s, dg_s, db_s, _, _) = res
        if groups == 1:
            assert torch.equal(ga_f.view(torch.int16), ga_s.view(torch.int16)), "ga = bf16(sum of the slabs)"
            assert relmax(s1_f, s1_s) < 1e-4 and relmax(s2_f, s2_s) < 1e-4
        assert relmax(gz_f, gz_s) < 8e-3 and float((gz_f.float() - gz_s.float()).abs().mean()) < 1e-5 * float(gz_s.float().abs().mean() + 1)
        assert relmax(dg_f - 2.0, dg_s - 2.0) < 1e-4 and relmax(db_f + 1.0, db_s + 1.0) < 1e-4      # accumulated onto 2 / -1
        for h in range(groups):
            zh = zb.float().view(groups, -1, I)[h]
            gah = ga_s.float().view(groups, -1, I)[h]
            mu, rstd = mean_b.view(groups, I)[h], inv_b.view(groups, I)[h]
            xh = (zh - mu) * rstd
            gy = gah * torch.where(xh * gam_i + bet_i > 0, 1.0, 0.2)
            ref = gam_i * rstd * (gy - gy.mean(0) - xh * (gy * xh).mean(0))
            assert relmax(gz_f.view(groups, -1, I)[h], ref) < 8e-3
    assert int(fu._sb_sync.abs().sum()) == int(fu._sb_sync[17::16].abs().sum()), "arrival counters / error word left at zero"
    assert int(fu._sb_sync[0]) == 0, "no hand-off timed out"


@pytest.mark.parametrize("I,O,hs,n,groups", [(128, 256, 64, 64, 1), (256, 512, 32, 64, 1), (128, 256, 64, 128, 2),
                                              (512, 1024, 16, 128, 2), (64, 128, 128, 64, 1)])
def test_bn_backward_sums_in_conv_epilogue(I, O, hs, n, groups):
    """Data-gradient convs that do not split K produce the BatchNorm-backward sums of the block they feed in their own
    epilogue (rg_conv_up_bnbwd / rg_conv_down_bnbwd + rg_bn_act_bwd_partials) instead of a reduction pass over (z, ga).
    Both conv directions against the separate-pass path of the same library: ga bit-identical, sums to fp32 round-off on the
    scale of the summands, gz to one bf16 rounding, dgamma / dbeta accumulated alike; and gz against plain tensor arithmetic."""
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(900 + I + n)
    fu, se = HipOps(torch.bfloat16, dev), HipOps(torch.bfloat16, dev)
    fu.bwd_epilogue, se.bwd_epilogue = True, False          # (opt-in: measured no faster than the separate pass, see ops_hip)
    se.split_bn = False                                     # reference side: conv, slab reduction, reduce, finish, apply
    w = (torch.randn(O, 4, 4, I, generator=gen) * (2.0 / (I * 16)) ** 0.5).bfloat16().float().to(dev)
    cws = [ConvW(w.clone(), None, torch.zeros_like(w), None, "OHWI") for _ in range(2)]
    ho = hs // 2
    for direction in ("up", "down"):
        if direction == "up":       # nn.Conv2d's data gradient: g [n, ho, ho, O] -> ga [n, hs, hs, I], consumer block has I channels
            src = torch.randn(n, ho, ho, O, generator=gen).bfloat16().to(dev)
            C, shape = I, (n, hs, hs, I)
        else:                        # nn.ConvTranspose2d's: x [n, hs, hs, I] -> ga [n, ho, ho, O]
            src = torch.randn(n, hs, hs, I, generator=gen).bfloat16().to(dev)
            C, shape = O, (n, ho, ho, O)
        zb = (torch.randn(*shape, generator=gen) * 1.3 + 0.2).bfloat16().to(dev)
        gam, bet = (1 + 0.1 * torch.randn(C, generator=gen)).to(dev), (0.1 * torch.randn(C, generator=gen)).to(dev)
        res = []
        for ops, cw in ((fu, cws[0]), (se, cws[1])):
            if groups == 1:
                _, mean, inv = ops.bn_forward(zb.clone(), gam, bet, 0.2, 1e-5, 0.1)
            else:
                _, mean, inv = ops.bn_forward2(zb.clone(), gam, bet, 0.2, 1e-5, 0.1)
            bnb = (zb, mean, inv, gam, bet, 0.2, groups)
            ga = ops.conv_up(src, cw, defer=groups, bn_bwd=bnb) if direction == "up" else ops.conv_down(src, cw, defer=groups, bn_bwd=bnb)
            fused = getattr(ga, "_rg_bwd_partials", None) is not None
            if ops is fu:
                rows = fu.lib.rg_conv_bnbwd_rows(1 if direction == "up" else 0, n, ho, ho, O, I, groups, fu.dt, fu.algo)
                split = fu.lib.rg_conv_split(1 if direction == "up" else 0, n, ho, ho, O, I, fu.dt, fu.algo) > 1
                assert fused == (rows > 0 and not split), (direction, rows, split)
            else:
                assert not fused
            dg, db = torch.full((C,), 2.0, device=dev), torch.full((C,), -1.0, device=dev)
            if groups == 1:
                gz, s1, s2 = ops.bn_act_bwd(zb, ga, mean, inv, gam, bet, 0.2, dg, db, True)
            else:
                gz, s1, s2 = ops.bn_act_bwd2(zb, ga, mean, inv, gam, bet, 0.2, dg, db, True), None, None
            torch.cuda.synchronize()
            assert getattr(ga, "_rg_bwd_partials", None) is None and getattr(ga, "_rg_slabs", None) is None
            res.append((ga, gz, s1, s2, dg, db, mean, inv))
        (ga_f, gz_f, s1_f, s2_f, dg_f, db_f, mean, inv), (ga_s, gz_s, s1_s, s2_s, dg_s, db_s, _, _) = res
        if getattr(fu, "_last_ga_written", True) and not (groups == 2 and fu.lib.rg_conv_split(
                1 if direction == "up" else 0, n, ho, ho, O, I, fu.dt, fu.algo) > 1):
            # (a double-batch split-K launch hands its slabs to the fused BatchNorm kernel, which does not write ga)
            assert torch.equal(ga_f.view(torch.int16), ga_s.view(torch.int16)), direction
        rows_per_group = ga_s.numel() // C // groups
        scale = float(ga_s.float().abs().mean()) * rows_per_group ** 0.5 + 1e-12        # |sum| of signed summands ~ sqrt(rows) |ga|
        if groups == 1:
            assert float((s1_f - s1_s).abs().max()) < 2e-4 * scale and float((s2_f - s2_s).abs().max()) < 4e-4 * scale, direction
        assert float(((dg_f - 2.0) - (dg_s - 2.0)).abs().max()) < 8e-4 * scale and float(((db_f + 1.0) - (db_s + 1.0)).abs().max()) < 4e-4 * scale
        assert relmax(gz_f, gz_s) < 8e-3 and float((gz_f.float() - gz_s.float()).abs().mean()) < 2e-5 * float(gz_s.float().abs().mean() + 1)
        for h in range(groups):
            zh, gah = zb.float().view(groups, -1, C)[h], ga_s.float().view(groups, -1, C)[h]
            mu, rstd = mean.view(groups, C)[h], inv.view(groups, C)[h]
            xh = (zh - mu) * rstd
            gy = gah * torch.where(xh * gam + bet > 0, 1.0, 0.2)
            ref = gam * rstd * (gy - gy.mean(0) - xh * (gy * xh).mean(0))
            assert relmax(gz_f.view(groups, -1, C)[h], ref) < 8e-3, (direction, h)


@pytest.mark.parametrize("wslab16", [0, 1])
@pytest.mark.parametrize("I,O,hs,two", [(64, 128, 64, True), (128, 256, 32, False), (256, 512, 16, True), (512, 1024, 8, True),
                                         (1024, 2048, 4, False)])
def test_split_k_weight_gradient_slabs_inside_the_adam_step(I, O, hs, two, wslab16):
    """Round 5: a split-K weight-gradient launch may leave its fp32 slabs unreduced (rg_conv_wgrad_slabs) and the fused Adam
    step sums them itself (rg_adam_step_slabs: one launch over a flat buffer cut into plain and slab segments; 1 / 4 / 16 slab
    lanes per 16-byte column by nsplit).  At the benchmark's five layer shapes (one and two segments): against the reduced
    gradient (rg_conv_wgrad / rg_conv_wgrad2) followed by rg_adam_step_dev on the same flat buffer -- moments to fp32 rounding of
    a different summation order, weights within two steps' bound of the sign-like first update, bf16 shadow = rounded weights.
    wslab16 = 1 (the default): the partial tiles are bf16 -- each partial sum rounded once (2^-9 relative), added in fp32."""
    import ctypes as C
    dev = torch.device("cuda:0")
    ops = HipOps(torch.bfloat16, dev)
    lib = _abi.load()
    _abi.check(lib.rg_set_option(b"wslab16", wslab16), "set_option")
    try:
        _slabs_inside_adam(lib, ops, dev, I, O, hs, two, wslab16)
    finally:
        lib.rg_set_option(b"wslab16", -1)


def _slabs_inside_adam(lib, ops, dev, I, O, hs, two, wslab16):
    import ctypes as C
    gen = torch.Generator(device="cpu").manual_seed(31)
    low0 = torch.randn(N, hs, hs, O, generator=gen).bfloat16().to(dev)
    high0 = torch.randn(N, 2 * hs, 2 * hs, I, generator=gen).bfloat16().to(dev)
    low1 = torch.randn(N, hs, hs, O, generator=gen).bfloat16().to(dev) if two else None
    high1 = torch.randn(N, 2 * hs, 2 * hs, I, generator=gen).bfloat16().to(dev) if two else None
    nw = O * 16 * I
    head, tail = 4096, 1000                                  # plain segments in front of / behind the layer (tail: not a multiple of 4)
    total = head + nw + tail
    pad = (-total) % 4
    p0 = torch.randn(total + pad, generator=gen).to(dev) * 0.05
    g0 = torch.randn(total + pad, generator=gen).to(dev) * 0.01
    m0 = torch.randn(total + pad, generator=gen).to(dev) * 0.01
    v0 = (torch.rand(total + pad, generator=gen).to(dev) * 1e-4)
    hyper = torch.zeros(12, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    _abi.check(lib.rg_adam_hyper_dev(step.data_ptr(), 4e-4, 0.5, 0.999, 1e-8, 0.0, hyper.data_ptr(), ops.stream), "hyper")
    wsb = int(lib.rg_conv_wgrad_workspace_bytes(N, hs, hs, O, I, ops.dt, ops.algo))
    ws = torch.empty(max(wsb, 256) + 4096, dtype=torch.uint8, device=dev)
    ptr = lambda t: 0 if t is None else t.data_ptr()

    # reference: reduced gradient into g[head : head + nw], then the plain fused step over the whole buffer
    pr, gr, mr, vr = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    shr = torch.zeros(total + pad, dtype=torch.bfloat16, device=dev)
    dw = gr[head:head + nw]
    if two:
        _abi.check(lib.rg_conv_wgrad2(ptr(low0), ptr(high0), ptr(low1), ptr(high1), dw.data_ptr(), N, hs, hs, O, I, ops.dt, 0,
                                      ops.algo, ws.data_ptr(), ws.numel(), ops.stream), "rg_conv_wgrad2")
    else:
        _abi.check(lib.rg_conv_wgrad(ptr(low0), ptr(high0), dw.data_ptr(), N, hs, hs, O, I, ops.dt, 0, ops.algo, ws.data_ptr(),
                                     ws.numel(), ops.stream), "rg_conv_wgrad")
    _abi.check(lib.rg_adam_step_dev(pr.data_ptr(), gr.data_ptr(), mr.data_ptr(), vr.data_ptr(), total, hyper.data_ptr(),
                                    shr.data_ptr(), 0, ops.stream), "rg_adam_step_dev")

    # deferred: slabs stay in `slab`, the segmented step sums them
    pd, gd, md, vd = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    shd = torch.zeros(total + pad, dtype=torch.bfloat16, device=dev)
    slab = torch.empty_like(ws)
    ns, sdt = C.c_int(0), C.c_int(-1)
    _abi.check(lib.rg_conv_wgrad_slabs(ptr(low0), ptr(high0), ptr(low1), ptr(high1), gd[head:head + nw].data_ptr(), N, hs, hs,
                                       O, I, ops.dt, ops.algo, slab.data_ptr(), slab.numel(), C.addressof(ns), C.addressof(sdt),
                                       ops.stream), "rg_conv_wgrad_slabs")
    print("layer %d -> %d at %d^2, %d segment(s): nsplit %d, slab dtype %d" % (I, O, hs, 2 if two else 1, ns.value, sdt.value))
    s16 = sdt.value == _abi.RG_BF16
    if ns.value > 1:
        assert s16 == bool(wslab16) or not s16               # (the 128 x 128 kernel's slabs stay fp32 under either setting)
        gd[head:head + nw].fill_(float("nan"))               # the reduced gradient must never be read
        table = [(0, head, 0, 0, 0), (head, nw, slab.data_ptr(), ns.value, sdt.value), (head + nw, tail, 0, 0, 0)]
    else:
        assert sdt.value == _abi.RG_F32
        table = [(0, total, 0, 0, 0)]                        # no split at this shape: dw was written, one plain segment
    k = len(table)
    offs = (C.c_ulonglong * k)(*[t[0] for t in table])
    lens = (C.c_ulonglong * k)(*[t[1] for t in table])
    slabs = (C.c_void_p * k)(*[t[2] or None for t in table])
    nsp = (C.c_int * k)(*[t[3] for t in table])
    sdts = (C.c_int * k)(*[t[4] for t in table])
    _abi.check(lib.rg_adam_step_slabs(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), total, hyper.data_ptr(),
                                      shd.data_ptr(), k, C.addressof(offs), C.addressof(lens), C.addressof(slabs),
                                      C.addressof(nsp), C.addressof(sdts), ops.stream), "rg_adam_step_slabs")
    torch.cuda.synchronize()
    assert torch.isfinite(pd[:total]).all() and torch.isfinite(md[:total]).all() and torch.isfinite(vd[:total]).all()
    # plain segments: the same arithmetic, bit for bit
    for a, b in ((pr, pd), (mr, md), (vr, vd)):
        assert torch.equal(a[:head], b[:head]) and torch.equal(a[head + nw:total], b[head + nw:total])
    assert torch.equal(shr[:head], shd[:head]) and torch.equal(shr[head + nw:total], shd[head + nw:total])
    sl = slice(head, head + nw)
    gscale = float((mr[sl] - 0.5 * m0[sl]).abs().max())      # 0.5 * |g|_max
    # bf16 partial tiles: every partial sum carries a relative 2^-9 rounding, the sum of nsplit of them ~ 2^-9 of the gradient's
    # typical size (independent signs); m moves by half of that, v by 2 g dg / 1000
    assert float((mr[sl] - md[sl]).abs().max()) <= (1e-2 if s16 else 2e-5) * gscale
    assert float((vr[sl] - vd[sl]).abs().max()) <= (2e-2 if s16 else 1e-4) * float(vr[sl].abs().max())
    if s16:
        rel = float((mr[sl] - md[sl]).norm() / (mr[sl] - 0.5 * m0[sl]).norm())
        print("  bf16 partial tiles: relative L2 error of the summed gradient %.2e" % rel)
        assert rel <= 4e-3
    # the step lr * m_hat / (sqrt(v_hat) + eps) of every element: within 2 % of the largest step (a different fp32 summation
    # order of the gradient moves m and v in the 6th digit; where v is tiny the quotient amplifies it)
    if s16:
        # an element whose gradient nearly cancels its old moment may flip the sign of its (sign-like) first step under a 2^-9
        # perturbation of the gradient: no per-element bound -- the update as a whole, and how many elements moved visibly
        du_r, du_d = (pr[sl] - p0[sl]).double(), (pd[sl] - p0[sl]).double()
        cos = float((du_r * du_d).sum() / (du_r.norm() * du_d.norm()))
        moved = float(((du_r - du_d).abs() > 5e-2 * float(du_r.abs().max())).float().mean())
        print("  update cosine %.6f, elements whose step moved by more than 5 %% of the largest step: %.2e" % (cos, moved))
        assert cos >= 0.998 and moved <= 4e-3          # measured: 0.99925 / 1.05e-3 (random gradients against random old moments)
    else:
        assert float((pr[sl] - pd[sl]).abs().max()) <= 2e-2 * float((pr[sl] - p0[sl]).abs().max())
    assert torch.equal(shd[:total], pd[:total].bfloat16())
    # bad tables are refused
    bad_off = (C.c_ulonglong * 1)(4)
    one_len = (C.c_ulonglong * 1)(total)
    nul = (C.c_void_p * 1)(None)
    zero = (C.c_int * 1)(0)
    assert lib.rg_adam_step_slabs(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), total, hyper.data_ptr(), 0, 1,
                                  C.addressof(bad_off), C.addressof(one_len), C.addressof(nul), C.addressof(zero),
                                  C.addressof(zero), ops.stream) != 0


@pytest.mark.parametrize("two", [False, True])
def test_weight_gradient_and_adam_step_in_one_launch(two):
    """Round 5 (b2): the two 33.5 M-parameter layers (D.5, G.1: 1024 <-> 2048 channels at 4 x 4 / 8 x 8) have a weight-gradient
    plan without split-K; rg_conv_wgrad_adam applies the optimizer step to the gradient tile where it sits instead of writing
    it.  Against rg_conv_wgrad[2] + rg_adam_step_dev on the same buffers: the same products in the same order and the one Adam
    expression (rg_common.h) -- bit for bit on p, m, v and the bf16 image; the plain segments around the tensor are stepped by
    rg_adam_step_slabs with the tensor's segment skipped (nsplit = -1)."""
    import ctypes as C
    I, O, hs = 1024, 2048, 4
    dev = torch.device("cuda:0")
    ops = HipOps(torch.bfloat16, dev)
    lib = _abi.load()
    assert lib.rg_conv_wgrad_adam_supported(N, hs, hs, O, I, int(two), ops.dt, ops.algo) == 1
    assert lib.rg_conv_wgrad_adam_supported(N, 64, 64, 128, 64, 1, ops.dt, ops.algo) == 0          # a split-K plan
    gen = torch.Generator(device="cpu").manual_seed(47)
    low0 = torch.randn(N, hs, hs, O, generator=gen).bfloat16().to(dev)
    high0 = torch.randn(N, 2 * hs, 2 * hs, I, generator=gen).bfloat16().to(dev)
    low1 = torch.randn(N, hs, hs, O, generator=gen).bfloat16().to(dev) if two else None
    high1 = torch.randn(N, 2 * hs, 2 * hs, I, generator=gen).bfloat16().to(dev) if two else None
    nw = O * 16 * I
    head, tail = 4096, 1000
    total = head + nw + tail
    pad = (-total) % 4
    p0 = torch.randn(total + pad, generator=gen).to(dev) * 0.05
    g0 = torch.randn(total + pad, generator=gen).to(dev) * 0.01
    m0 = torch.randn(total + pad, generator=gen).to(dev) * 0.01
    v0 = (torch.rand(total + pad, generator=gen).to(dev) * 1e-4)
    hyper = torch.zeros(12, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    _abi.check(lib.rg_adam_hyper_dev(step.data_ptr(), 4e-4, 0.5, 0.999, 1e-8, 0.0, hyper.data_ptr(), ops.stream), "hyper")
    wsb = int(lib.rg_conv_wgrad_workspace_bytes(N, hs, hs, O, I, ops.dt, ops.algo))
    ws = torch.empty(max(wsb, 256) + 4096, dtype=torch.uint8, device=dev)
    ptr = lambda t: 0 if t is None else t.data_ptr()

    pr, gr, mr, vr = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    shr = torch.zeros(total + pad, dtype=torch.bfloat16, device=dev)
    dw = gr[head:head + nw]
    if two:
        _abi.check(lib.rg_conv_wgrad2(ptr(low0), ptr(high0), ptr(low1), ptr(high1), dw.data_ptr(), N, hs, hs, O, I, ops.dt, 0,
                                      ops.algo, ws.data_ptr(), ws.numel(), ops.stream), "rg_conv_wgrad2")
    else:
        _abi.check(lib.rg_conv_wgrad(ptr(low0), ptr(high0), dw.data_ptr(), N, hs, hs, O, I, ops.dt, 0, ops.algo, ws.data_ptr(),
                                     ws.numel(), ops.stream), "rg_conv_wgrad")
    _abi.check(lib.rg_adam_step_dev(pr.data_ptr(), gr.data_ptr(), mr.data_ptr(), vr.data_ptr(), total, hyper.data_ptr(),
                                    shr.data_ptr(), 0, ops.stream), "rg_adam_step_dev")

    pd, gd, md, vd = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    gd[head:head + nw].fill_(float("nan"))                   # never written, never read
    shd = torch.zeros(total + pad, dtype=torch.bfloat16, device=dev)
    _abi.check(lib.rg_conv_wgrad_adam(ptr(low0), ptr(high0), ptr(low1), ptr(high1), pd[head:].data_ptr(), md[head:].data_ptr(),
                                      vd[head:].data_ptr(), hyper.data_ptr(), shd[head:].data_ptr(), N, hs, hs, O, I, ops.dt,
                                      ops.algo, ops.stream), "rg_conv_wgrad_adam")
    table = [(0, head, 0, 0), (head, nw, 0, -1), (head + nw, tail, 0, 0)]
    k = len(table)
    offs = (C.c_ulonglong * k)(*[t[0] for t in table])
    lens = (C.c_ulonglong * k)(*[t[1] for t in table])
    slabs = (C.c_void_p * k)(*[None for t in table])
    nsp = (C.c_int * k)(*[t[3] for t in table])
    sdts = (C.c_int * k)(*[0 for t in table])
    _abi.check(lib.rg_adam_step_slabs(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), total, hyper.data_ptr(),
                                      shd.data_ptr(), k, C.addressof(offs), C.addressof(lens), C.addressof(slabs),
                                      C.addressof(nsp), C.addressof(sdts), ops.stream), "rg_adam_step_slabs")
    torch.cuda.synchronize()
    assert not torch.equal(pr[head:head + nw], p0[head:head + nw])
    for name, a, b in (("p", pr, pd), ("m", mr, md), ("v", vr, vd)):
        assert torch.equal(a[:total], b[:total]), (name, float((a[:total] - b[:total]).abs().max()))
    assert torch.equal(shr[:total], shd[:total])
    # refused: a shape whose plan splits K, unaligned moments
    assert lib.rg_conv_wgrad_adam(ptr(low0), ptr(high0), 0, 0, pd[head:].data_ptr(), md[head:].data_ptr(), vd[head:].data_ptr(),
                                  hyper.data_ptr(), 0, N, hs, hs, O, 64, ops.dt, ops.algo, ops.stream) != 0
    assert lib.rg_conv_wgrad_adam(ptr(low0), ptr(high0), ptr(low1), ptr(high1), pd[head:].data_ptr(), md[head + 1:].data_ptr(),
                                  vd[head:].data_ptr(), hyper.data_ptr(), 0, N, hs, hs, O, I, ops.dt, ops.algo, ops.stream) != 0


def test_weight_gradients_straight_onto_the_data_parallel_wire():
    """Round 5: in a data-parallel bf16 run the 4 x 4 layers' weight gradients reach the all-reduce wire without an fp32 gradient
    in between.  rg_conv_wgrad_wire (a plan without split-K) must equal rg_conv_wgrad + one bf16 rounding bit for bit;
    rg_grad_to_wire must round plain segments like rg_cast_pad, sum slab segments (fp32 and bf16 slabs) to within one bf16 ulp
    of the reduced gradient's rounding, and leave skipped segments alone."""
    import ctypes as C
    dev = torch.device("cuda:0")
    ops = HipOps(torch.bfloat16, dev)
    lib = _abi.load()
    gen = torch.Generator(device="cpu").manual_seed(53)
    ptr = lambda t: 0 if t is None else t.data_ptr()
    # (a) the no-split layer, one and two segments
    I, O, hs = 1024, 2048, 4
    nw = O * 16 * I
    for two in (False, True):
        low0 = torch.randn(N, hs, hs, O, generator=gen).bfloat16().to(dev)
        high0 = torch.randn(N, 2 * hs, 2 * hs, I, generator=gen).bfloat16().to(dev)
        low1 = torch.randn(N, hs, hs, O, generator=gen).bfloat16().to(dev) if two else None
        high1 = torch.randn(N, 2 * hs, 2 * hs, I, generator=gen).bfloat16().to(dev) if two else None
        wsb = int(lib.rg_conv_wgrad_workspace_bytes(N, hs, hs, O, I, ops.dt, ops.algo))
        ws = torch.empty(max(wsb, 256) + 4096, dtype=torch.uint8, device=dev)
        dw = torch.empty(nw, device=dev)
        if two:
            _abi.check(lib.rg_conv_wgrad2(ptr(low0), ptr(high0), ptr(low1), ptr(high1), dw.data_ptr(), N, hs, hs, O, I, ops.dt, 0,
                                          ops.algo, ws.data_ptr(), ws.numel(), ops.stream), "rg_conv_wgrad2")
        else:
            _abi.check(lib.rg_conv_wgrad(ptr(low0), ptr(high0), dw.data_ptr(), N, hs, hs, O, I, ops.dt, 0, ops.algo,
                                         ws.data_ptr(), ws.numel(), ops.stream), "rg_conv_wgrad")
        wire = torch.zeros(nw + 8, dtype=torch.bfloat16, device=dev)
        _abi.check(lib.rg_conv_wgrad_wire(ptr(low0), ptr(high0), ptr(low1), ptr(high1), wire[8:].data_ptr(), N, hs, hs, O, I,
                                          ops.dt, ops.algo, ops.stream), "rg_conv_wgrad_wire")
        torch.cuda.synchronize()
        assert torch.equal(wire[8:], dw.bfloat16()) and float(wire[:8].abs().max()) == 0.0
        assert lib.rg_conv_wgrad_wire(ptr(low0), ptr(high0), ptr(low1), ptr(high1), wire[1:].data_ptr(), N, hs, hs, O, I, ops.dt,
                                      ops.algo, ops.stream) != 0          # unaligned slice
    # (b) the segmented cast: plain / fp32 slabs / bf16 slabs / skipped / plain tail
    I, O, hs = 256, 512, 16
    nw = O * 16 * I
    low0 = torch.randn(N, hs, hs, O, generator=gen).bfloat16().to(dev)
    high0 = torch.randn(N, 2 * hs, 2 * hs, I, generator=gen).bfloat16().to(dev)
    wsb = int(lib.rg_conv_wgrad_workspace_bytes(N, hs, hs, O, I, ops.dt, ops.algo))
    ws = torch.empty(max(wsb, 256) + 4096, dtype=torch.uint8, device=dev)
    head, skip, tail = 4096, 2048, 1002
    total = head + nw + nw + skip + tail
    g = torch.randn(total + 2, generator=gen).to(dev)
    dw = torch.empty(nw, device=dev)
    _abi.check(lib.rg_conv_wgrad(ptr(low0), ptr(high0), dw.data_ptr(), N, hs, hs, O, I, ops.dt, 0, ops.algo, ws.data_ptr(),
                                 ws.numel(), ops.stream), "rg_conv_wgrad")
    slabs = {}
    for w16 in (0, 1):
        _abi.check(lib.rg_set_option(b"wslab16", w16), "set_option")
        try:
            buf = torch.empty_like(ws)
            ns, sdt = C.c_int(0), C.c_int(-1)
            _abi.check(lib.rg_conv_wgrad_slabs(ptr(low0), ptr(high0), 0, 0, g[head:].data_ptr(), N, hs, hs, O, I, ops.dt, ops.algo,
                                               buf.data_ptr(), buf.numel(), C.addressof(ns), C.addressof(sdt), ops.stream), "slabs")
            assert ns.value > 1 and sdt.value == (_abi.RG_BF16 if w16 else _abi.RG_F32)
            slabs[w16] = (buf, ns.value, sdt.value)
        finally:
            lib.rg_set_option(b"wslab16", -1)
    table = [(0, head, 0, 0, 0), (head, nw, slabs[0][0].data_ptr(), slabs[0][1], slabs[0][2]),
             (head + nw, nw, slabs[1][0].data_ptr(), slabs[1][1], slabs[1][2]), (head + 2 * nw, skip, 0, -1, 0),
             (head + 2 * nw + skip, tail, 0, 0, 0)]
    k = len(table)
    offs = (C.c_ulonglong * k)(*[t[0] for t in table])
    lens = (C.c_ulonglong * k)(*[t[1] for t in table])
    sl = (C.c_void_p * k)(*[t[2] or None for t in table])
    nsp = (C.c_int * k)(*[t[3] for t in table])
    sdts = (C.c_int * k)(*[t[4] for t in table])
    wire = torch.full((total + 2,), 7.0, dtype=torch.bfloat16, device=dev)
    _abi.check(lib.rg_grad_to_wire(g.data_ptr(), wire.data_ptr(), total, k, C.addressof(offs), C.addressof(lens), C.addressof(sl),
                                   C.addressof(nsp), C.addressof(sdts), ops.stream), "rg_grad_to_wire")
    torch.cuda.synchronize()
    assert torch.equal(wire[:head], g[:head].bfloat16())
    assert torch.equal(wire[head + 2 * nw + skip:total], g[head + 2 * nw + skip:total].bfloat16())
    assert float((wire[head + 2 * nw:head + 2 * nw + skip].float() - 7.0).abs().max()) == 0.0        # skipped: untouched
    assert float((wire[total:].float() - 7.0).abs().max()) == 0.0
    scale = float(dw.abs().max())
    e32 = (wire[head:head + nw].float() - dw).abs()
    e16 = (wire[head + nw:head + 2 * nw].float() - dw).abs()
    # fp32 slabs: the reduced gradient up to its summation order, rounded once; bf16 slabs: one more rounding per partial sum
    assert float(e32.max()) <= 2.0 ** -8 * scale and float((e32 / (dw.abs() + 1e-3 * scale)).max()) <= 2.0 ** -7
    assert float(e16.norm() / dw.norm()) <= 4e-3


def test_image_side_weight_gradient_partials_inside_the_adam_step():
    """Round 5: the image-side layers' weight gradient (rg_skinny_wgrad: 512 per-workgroup partial gradients + a reduction launch)
    with the partials LEFT for the optimizer step (rg_skinny_wgrad_slabs -> rg_adam_step_slabs), two contributions to one tensor
    (D(real) + D(fake); primal + tangent) as one run of 1024 slabs: against rg_skinny_wgrad twice (accumulate) + rg_adam_step_dev."""
    import ctypes as C
    dev = torch.device("cuda:0")
    ops = HipOps(torch.bfloat16, dev)
    lib = _abi.load()
    gen = torch.Generator(device="cpu").manual_seed(61)
    O, I, S = 64, 3, 256
    xs = [torch.randn(N, I, S, S, generator=gen).to(dev) for _ in range(2)]
    gs = [torch.randn(N, S // 2, S // 2, O, generator=gen).bfloat16().to(dev) for _ in range(2)]
    nw = O * I * 16
    wsb = int(lib.rg_skinny_wgrad_workspace_bytes(N, S // 2, S // 2, O, I))
    ws = torch.empty(wsb + 4096, dtype=torch.uint8, device=dev)
    head, tail = 1024, 500
    total = head + nw + tail
    p0 = torch.randn(total, generator=gen).to(dev) * 0.05
    g0 = torch.randn(total, generator=gen).to(dev) * 0.01
    m0 = torch.randn(total, generator=gen).to(dev) * 0.01
    v0 = torch.rand(total, generator=gen).to(dev) * 1e-4
    hyper = torch.zeros(12, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    _abi.check(lib.rg_adam_hyper_dev(step.data_ptr(), 4e-4, 0.5, 0.999, 1e-8, 0.0, hyper.data_ptr(), ops.stream), "hyper")
    # reference: reduced gradient (two contributions), plain step
    pr, gr, mr, vr = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    shr = torch.zeros(total, dtype=torch.bfloat16, device=dev)
    for k in range(2):
        _abi.check(lib.rg_skinny_wgrad(gs[k].data_ptr(), xs[k].data_ptr(), gr[head:].data_ptr(), N, S // 2, S // 2, O, I, ops.dt, k,
                                       ws.data_ptr(), ws.numel(), ops.stream), "rg_skinny_wgrad")
    _abi.check(lib.rg_adam_step_dev(pr.data_ptr(), gr.data_ptr(), mr.data_ptr(), vr.data_ptr(), total, hyper.data_ptr(),
                                    shr.data_ptr(), 0, ops.stream), "rg_adam_step_dev")
    # deferred
    pd, gd, md, vd = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    gd[head:head + nw].fill_(float("nan"))
    shd = torch.zeros(total, dtype=torch.bfloat16, device=dev)
    slab = torch.empty(2 * wsb + 4096, dtype=torch.uint8, device=dev)
    bias_part = torch.full((2048, O), float("nan"), device=dev)
    have = 0
    for k in range(2):
        ns = C.c_int(0)
        off = have * nw * 4
        bd = C.c_int(0)
        _abi.check(lib.rg_skinny_wgrad_slabs(gs[k].data_ptr(), xs[k].data_ptr(), N, S // 2, S // 2, O, I, ops.dt,
                                             slab.data_ptr() + off, slab.numel() - off, C.addressof(ns),
                                             bias_part[have:].data_ptr(), C.addressof(bd), ops.stream), "slabs")
        assert ns.value > 0 and bd.value == 1
        have += ns.value
    print("image-side weight gradient: %d partial slabs of %d elements" % (have, nw))
    table = [(0, head, 0, 0, 0), (head, nw, slab.data_ptr(), have, _abi.RG_F32), (head + nw, tail, 0, 0, 0)]
    k = len(table)
    offs = (C.c_ulonglong * k)(*[t[0] for t in table])
    lens = (C.c_ulonglong * k)(*[t[1] for t in table])
    sl = (C.c_void_p * k)(*[t[2] or None for t in table])
    nsp = (C.c_int * k)(*[t[3] for t in table])
    sdts = (C.c_int * k)(*[t[4] for t in table])
    _abi.check(lib.rg_adam_step_slabs(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), total, hyper.data_ptr(),
                                      shd.data_ptr(), k, C.addressof(offs), C.addressof(lens), C.addressof(sl), C.addressof(nsp),
                                      C.addressof(sdts), ops.stream), "rg_adam_step_slabs")
    torch.cuda.synchronize()
    for a, b in ((pr, pd), (mr, md), (vr, vd)):
        assert torch.equal(a[:head], b[:head]) and torch.equal(a[head + nw:], b[head + nw:])
    sl_ = slice(head, head + nw)
    gscale = float((mr[sl_] - 0.5 * m0[sl_]).abs().max())
    assert float((mr[sl_] - md[sl_]).abs().max()) <= 2e-5 * gscale
    assert float((vr[sl_] - vd[sl_]).abs().max()) <= 1e-4 * float(vr[sl_].abs().max())
    assert float((pr[sl_] - pd[sl_]).abs().max()) <= 2e-2 * float((pr[sl_] - p0[sl_]).abs().max())
    assert torch.equal(shd, pd.bfloat16())
    # the bias gradient as a by-product of the same pass (a column of ones in the patch operand): the partials' sum, and the
    # reducing entry point, against the column sums of `low` (rg_col_sum and fp64)
    torch.cuda.synchronize()
    want = sum(g.double().sum(dim=(0, 1, 2)) for g in gs)
    got = bias_part[:have].double().sum(0)
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-3
    db = torch.zeros(O, device=dev)
    dw = torch.empty(nw, device=dev)
    for k in range(2):
        bd = C.c_int(0)
        _abi.check(lib.rg_skinny_wgrad_bias(gs[k].data_ptr(), xs[k].data_ptr(), dw.data_ptr(), db.data_ptr(), N, S // 2, S // 2, O,
                                            I, ops.dt, k, k, ws.data_ptr(), ws.numel(), C.addressof(bd), ops.stream), "wgrad_bias")
        assert bd.value == 1
    torch.cuda.synchronize()
    assert torch.equal(dw, gr[head:head + nw])                   # the weight gradient itself is what rg_skinny_wgrad writes
    assert float((db.double() - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-3
    ref = torch.zeros(O, device=dev)
    for k in range(2):
        ops.col_sum(gs[k], ref, bool(k))
    torch.cuda.synchronize()
    assert float((db - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-4
    # the fp32 generic shape has no slab form: nothing launched, 0 reported
    ns, bd = C.c_int(7), C.c_int(7)
    _abi.check(lib.rg_skinny_wgrad_slabs(gs[0].data_ptr(), xs[0].data_ptr(), N, S // 2, S // 2, O, I, _abi.RG_F32, slab.data_ptr(),
                                         slab.numel(), C.addressof(ns), 0, C.addressof(bd), ops.stream), "slabs(f32)")
    assert ns.value == 0 and bd.value == 0
