"""The fp32 mode's convolutions on the bf16 matrix cores from operands split once per tensor into bf16 planes
(rna_gan_amd/csrc/rg_conv8f.hip, rg_wgrad8f.hip; HipOps.f32_planes): v = h + m + l exactly, K-concatenated plane pairs,
fp32 accumulation.  Against fp64 convolutions of the SAME fp32 operands: 6 products per fp32 product are fp32-grade
(the f32 instruction's class), 3 products hold 2^-16 per product."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from rna_gan_amd.engine import ConvW
from rna_gan_amd.ops_hip import HipOps


def _ops(products, monkeypatch):
    monkeypatch.setenv("RNAGAN_F32_PLANES", str(products))
    return HipOps(torch.float32, "cuda:0")


def test_split_planes_is_exact_and_keeps_non_finite_values():
    ops = HipOps(torch.float32, "cuda:0")
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(4096, generator=g) * torch.exp(torch.randn(4096, generator=g) * 8)).cuda()
    x[:8] = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 3.4e38, -3.4e38, 1e-30]).cuda()
    p = ops._planes(x).float()
    s = p[0] + p[1] + p[2]
    fin = torch.isfinite(x)
    fin[5:7] = False                       # |v| beyond the largest finite bf16 rounds to an infinite h: class kept, value not
    fin &= x.abs() >= 1e-30                # (fp32 values whose residual planes would be subnormal are outside the claim)
    fin[:2] = True
    assert torch.equal(s[fin], x[fin])     # h + m + l == v exactly (fp32 adds of the three planes are exact here)
    assert torch.isinf(s[2]) and s[2] > 0 and torch.isinf(s[3]) and s[3] < 0 and torch.isnan(s[4])
    assert torch.isinf(s[5]) and torch.isinf(s[6])
    assert float(p[1][2]) == 0.0 and float(p[2][2]) == 0.0       # an infinity's residual planes are zero, not NaN


@pytest.mark.parametrize("products,tol", [(6, 2e-6), (3, 6e-5)])
@pytest.mark.parametrize("N,H,I,O", [(8, 32, 128, 256),      # 256 x 256 tiles, split-K
                                      (16, 64, 64, 128),      # 512 x 128 tiles
                                      (64, 8, 512, 1024)])    # deep layer: few row tiles, long K
def test_conv_down_up_on_planes(products, tol, N, H, I, O, monkeypatch):
    ops = _ops(products, monkeypatch)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, H, I, generator=g).cuda()
    w = (torch.randn(O, I, 4, 4, generator=g) * 0.05).cuda()
    wt = w.permute(0, 2, 3, 1).contiguous()
    cw = ConvW(wt, None, torch.zeros_like(wt), None, "OHWI")
    assert ops.lib.rg_f32p_conv_supported(0, N, H // 2, H // 2, O, I, products)
    y, st = ops.conv_down(x, cw, want_stats=True)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), stride=2, padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    assert y.dtype == torch.float32 and float((y.double() - ref).abs().max()) <= tol * scale
    if st is not None:      # BatchNorm column sums from the epilogue (unsplit launches)
        s = st.double().sum(0)
        assert float((s[0] - ref.reshape(-1, O).sum(0)).abs().max()) <= 1e-4 * float(ref.reshape(-1, O).abs().sum(0).max())
        assert float((s[1] - (ref * ref).reshape(-1, O).sum(0)).abs().max()) <= 1e-4 * float((ref * ref).reshape(-1, O).sum(0).max())
    gy = torch.randn(N, H // 2, H // 2, O, generator=g).cuda()
    if ops.lib.rg_f32p_conv_supported(1, N, H // 2, H // 2, O, I, products):
        gx = ops.conv_up(gy, cw)
        refx = F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), w.double(), stride=2, padding=1).permute(0, 2, 3, 1)
        assert float((gx.double() - refx).abs().max()) <= tol * float(refx.abs().max())
        # with the consumer's LeakyReLU backward (fused into the 64-column kernel, a pass behind the others)
        a0 = torch.randn(N, H, H, I, generator=g).cuda()
        gm = ops.conv_up(gy, cw, a0, 0.2)
        refm = refx * torch.where(a0 > 0, 1.0, 0.2).double()
        assert float((gm.double() - refm).abs().max()) <= tol * float(refx.abs().max())


def test_cached_planes_follow_in_place_changes(monkeypatch):
    """The planes of an activation are cached on the tensor (split once for the conv and for the weight gradient that reads it
    again); a torch in-place op on the tensor bumps its version counter and the planes are split again."""
    ops = _ops(6, monkeypatch)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, 32, 32, 128, generator=g).cuda()
    wt = (torch.randn(256, 4, 4, 128, generator=g) * 0.05).cuda()
    cw = ConvW(wt, None, torch.zeros_like(wt), None, "OHWI")
    y1 = ops.conv_down(x, cw)
    planes = x._rg_planes[1]
    y1b = ops.conv_down(x, cw)
    assert x._rg_planes[1] is planes and torch.equal(y1, y1b)
    x.mul_(2.0)
    y2 = ops.conv_down(x, cw)
    assert torch.equal(y2, 2.0 * y1)                         # (a power of two: exact)


@pytest.mark.parametrize("products,tol", [(6, 2e-6), (3, 6e-5)])
@pytest.mark.parametrize("N,H,I,O", [(8, 32, 128, 256),      # 256 x 256 tiles (wgrad8_kernel)
                                      (8, 64, 64, 128)])      # 128 x 512 tiles (wgrad8n_kernel: the layers with 128 low-side channels)
def test_weight_gradient_on_planes_one_and_two_segments(products, tol, N, H, I, O, monkeypatch):
    ops = _ops(products, monkeypatch)
    g = torch.Generator().manual_seed(9)
    x0, x1 = torch.randn(N, H, H, I, generator=g).cuda(), torch.randn(N, H, H, I, generator=g).cuda()
    g0, g1 = torch.randn(N, H // 2, H // 2, O, generator=g).cuda(), torch.randn(N, H // 2, H // 2, O, generator=g).cuda()
    wt = torch.zeros(O, 4, 4, I).cuda()
    cw = ConvW(wt, None, torch.zeros_like(wt), None, "OHWI")
    assert ops.lib.rg_f32p_wgrad_supported(N, H // 2, H // 2, O, I, products)

    def ref(gy, x):
        xp = F.pad(x.double(), (0, 0, 1, 1, 1, 1)).unfold(1, 4, 2).unfold(2, 4, 2).permute(0, 1, 2, 4, 5, 3)
        return torch.einsum("nhwo,nhwkli->okli", gy.double(), xp)
    ops.conv_wgrad(g0, x0, cw, False)
    r0 = ref(g0, x0)
    assert float((cw.dw.double() - r0).abs().max()) <= tol * float(r0.abs().max())
    ops.conv_wgrad2(g0, x0, g1, x1, cw, True)                 # accumulate two more segments onto the first result
    r = 2 * r0 + ref(g1, x1)
    assert float((cw.dw.double() - r).abs().max()) <= 2 * tol * float(r.abs().max())


def test_three_product_tier_statistics_on_unselected_inputs(monkeypatch):
    """RNAGAN_F32_PLANES=3 (hh + hm + mh: 2^-16 per product) over 24 unselected seeds at 64 x 64, batch 8, against the CPU
    oracle: what the 100-seed table (profiles/round6_tolerance_fp32_planes3.txt: median 3.4e-4, 90th 2.1e-3, 99th 6.5e-3, update
    cosines >= 0.95) says, asserted with headroom.  Between bf16 (median 6.6e-3) and the 6-product default (3.5e-5)."""
    import test_train_gpu as T
    monkeypatch.setenv("RNAGAN_F32_PLANES", "3")
    seeds = list(range(701, 725))
    errs, coss = T._loss_and_update_statistics(64, seeds, precisions=("fp32",))
    e, c = errs["fp32"], coss["fp32"]
    T._describe("fp32 on 3-product planes 64x64", e, c, seeds)
    flat = np.sort(e.reshape(-1))
    assert np.isfinite(flat).all()
    assert float(np.median(flat)) <= 2e-3 and float(flat[int(0.9 * len(flat))]) <= 1e-2
    # (the tail is the same in every mode: a seed whose penalty sits on a LeakyReLU kink of the critic head -- seed 705 here moves the
    # penalty by 0.54 and the discriminator's update cosine to 0.83; the fp32-grade modes show such seeds at 1e-2)
    assert float(np.median(c)) >= 0.98 and int((c.min(1) < 0.90).sum()) <= 2 and float(c.min()) >= 0.6
