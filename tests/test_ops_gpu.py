"""GPU parity: every C-ABI op (through rna_gan_amd.ops_hip.HipOps) against its torch twin
(oracle/ops_ref.py) on the same seeded inputs, generic (fp32 / bf16 storage) and MFMA (bf16) paths.
Tolerances: fp32 path 2e-5 of the tensor's max magnitude (different summation order only);
bf16 path 1.5e-2 (one bf16 rounding of inputs/outputs, fp32 accumulation)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.ops_ref import RefOps
from rna_gan_amd.engine import ConvW

TOL = {torch.float32: 2e-5, torch.bfloat16: 1.5e-2, torch.float16: 2e-3}


def _hip(dtype):
    from rna_gan_amd.ops_hip import HipOps
    return HipOps(dtype, "cuda:0")


def rnd(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(size=shape) * scale).astype(np.float32))


def relerr(a, b):
    a = a.detach().float().cpu().double()
    b = b.detach().float().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all(), "non-finite output"
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(a, b, tol, what=""):
    e = relerr(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


def dev(t, dtype=None):
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


def cwpair(w, bias=None):
    return ConvW(w.clone(), bias), ConvW(w.cuda(), None if bias is None else bias.cuda())


def cwpair_tm(w):
    """Tap-major handles w[O][4][4][I] (what the HIP conv kernels take) for the twin and the product, each with dw."""
    wt = w.permute(0, 2, 3, 1).contiguous()
    return (ConvW(wt.clone(), None, torch.zeros_like(wt), None, "OHWI"),
            ConvW(wt.cuda(), None, torch.full_like(wt, 7.0).cuda(), None, "OHWI"))


def test_selftest_layouts():
    assert _hip(torch.bfloat16).selftest() == [0, 0]


CONV_CASES = [
    # N, Hi, Wi, I, O, dtype
    (2, 8, 8, 4, 8, torch.float32),
    (2, 8, 8, 4, 8, torch.bfloat16),
    (3, 4, 8, 5, 6, torch.float32),
    (2, 16, 16, 64, 128, torch.float32),
    (2, 16, 16, 64, 128, torch.bfloat16),      # MFMA, I=64
    (3, 8, 8, 128, 256, torch.bfloat16),       # MFMA, M tail (48 rows), 2 column tiles
    (1, 8, 8, 256, 128, torch.bfloat16),       # MFMA, 16 rows only
    (4, 32, 32, 64, 128, torch.bfloat16),      # MFMA, several row tiles
    # fp32 storage on the f32 matrix cores (gemm_mfma32_kernel: 128 x 128 tiles, v_mfma_f32_32x32x2_f32)
    (3, 16, 16, 24, 100, torch.float32),       # down: 192 x 100 (ragged tile), K = 384, non-power-of-two I (division path)
    (2, 32, 32, 136, 40, torch.float32),       # up: 512 x 136 per class (two column tiles, ragged); down stays on the vector kernel
    (2, 16, 16, 128, 136, torch.float32),      # both directions + weight gradient (136 x 2048 outputs, split-K slabs)
    # structured operands (gemm_mfma32s_kernel: power-of-two geometry, padding / tails as out-of-range buffer offsets)
    (2, 64, 64, 32, 72, torch.float32),        # down (ragged 72 columns) + weight gradient with 32-pixel output rows: two k-tiles
                                               # per row (first / last pixel flags), 72 x 512 outputs, split-K; up: general (O = 72)
    (2, 32, 32, 40, 64, torch.float32),        # up structured (O = 64) with 40 ragged columns per class; down general (I = 40)
    (3, 8, 8, 64, 64, torch.float32),          # weight gradient over 4 x 4 outputs: a k-tile = one whole image (rows of 4 pixels)
]


@pytest.mark.parametrize("N,Hi,Wi,I,O,dtype", CONV_CASES)
def test_conv_down_up_wgrad(N, Hi, Wi, I, O, dtype):
    ref, hip = RefOps(dtype), _hip(dtype)
    w = rnd((O, I, 4, 4), 1, (2.0 / (I * 16)) ** 0.5)
    cr, ch = cwpair_tm(w)
    x = rnd((N, Hi, Wi, I), 2).to(dtype)
    y_ref = ref.conv_down(x, cr)
    y = hip.conv_down(dev(x), ch)
    check(y, y_ref, TOL[dtype], "conv_down")
    g = rnd((N, Hi // 2, Wi // 2, O), 3).to(dtype)
    u_ref = ref.conv_up(g, cr)
    u = hip.conv_up(dev(g), ch)
    check(u, u_ref, TOL[dtype], "conv_up")
    # BatchNorm partial sums from the MFMA epilogue (None where the launch cannot produce them): exact column sums of
    # the stored values, and bn_forward(partials=...) == bn_forward()
    for name, (yy, st) in (("conv_down", hip.conv_down(dev(x), ch, want_stats=True)),
                           ("conv_up", hip.conv_up(dev(g), ch, want_stats=True))):
        if st is None:
            continue
        C = yy.shape[-1]
        yf = yy.float().reshape(-1, C)
        check(st[:, 0, :].sum(0), yf.sum(0), 1e-4, name + " epilogue sum")
        check(st[:, 1, :].sum(0), (yf * yf).sum(0), 1e-4, name + " epilogue sumsq")
        gam, bet = dev(1 + 0.1 * rnd((C,), 40)), dev(0.1 * rnd((C,), 41))
        a1, m1, i1 = hip.bn_forward(yy, gam, bet, 0.2, 1e-5, 0.1, partials=st)
        a2, m2, i2 = hip.bn_forward(yy, gam, bet, 0.2, 1e-5, 0.1)
        check(m1, m2, 1e-5, name + " mean from partials"); check(i1, i2, 1e-4, name + " invstd from partials")
        check(a1, a2, 2e-2, name + " bn output from partials")
    m = rnd((N, Hi, Wi, I), 21).to(dtype)       # fused LeakyReLU backward of the consumer layer
    check(hip.conv_up(dev(g), ch, dev(m), 0.2), ref.conv_up(g, cr, m, 0.2), TOL[dtype], "conv_up(masked)")
    ref.conv_wgrad(g, x, cr, False)
    dw_ref = cr.dw.clone()
    hip.conv_wgrad(dev(g), dev(x), ch, False)          # overwrites the 7.0 fill
    check(ch.dw, dw_ref, TOL[dtype] * 2, "conv_wgrad")
    hip.conv_wgrad(dev(g), dev(x), ch, True)
    check(ch.dw, 2 * dw_ref, TOL[dtype] * 2, "conv_wgrad(accumulate)")
    # two-segment form: dw = wgrad(g, x) + wgrad(g2, x2)
    g2, x2 = rnd((N, Hi // 2, Wi // 2, O), 13).to(dtype), rnd((N, Hi, Wi, I), 12).to(dtype)
    ref.conv_wgrad(g, x, cr, False); ref.conv_wgrad(g2, x2, cr, True)
    ch.dw.fill_(-5.0)
    hip.conv_wgrad2(dev(g), dev(x), dev(g2), dev(x2), ch, False)
    check(ch.dw, cr.dw, TOL[dtype] * 2, "conv_wgrad2")
    hip.conv_wgrad2(dev(g), dev(x), dev(g2), dev(x2), ch, True)
    check(ch.dw, 2 * cr.dw, TOL[dtype] * 2, "conv_wgrad2(accumulate)")


@pytest.mark.parametrize("N,Hi,Wi,I,O", [(2, 16, 16, 128, 136), (3, 16, 16, 24, 100), (1, 32, 32, 64, 256),
                                         (3, 64, 64, 32, 72),        # structured, ragged, 3 images
                                         (5, 8, 8, 64, 96),          # weight gradient, 4-pixel rows (k-tile = one image)
                                         (4, 128, 128, 64, 512),     # 128 x 128 tiles: down (512 tiles) and weight gradient (768)
                                         (4, 128, 128, 128, 32)])    # 128 x 128 tiles: up (512 tiles over the four classes)
def test_f32_matrix_core_gemm_matches_vector_kernel(N, Hi, Wi, I, O):
    """fp32 mode: the f32-MFMA GEMM skeleton (option f32mma = 1) against the vector-ALU skeleton (f32mma = 0) through
    the same operand functors -- conv forward, transposed conv, weight gradient, generator layer 0 and a dense layer.  Both
    are fused-multiply-add chains in fp32; only the association of the k-loop differs (two interleaved k per MFMA step).
    f32mma = 2 (the default): the 128 x 128-tile launches form every fp32 product as six bf16 matrix-core products of the operands'
    exact three-way bf16 splits (gemm_bf16x3s_kernel; < 2^-25 relative per product, fp32 accumulation) -- same bound."""
    from rna_gan_amd import _abi
    hip = _hip(torch.float32)
    lib = hip.lib
    w = rnd((O, I, 4, 4), 1, (2.0 / (I * 16)) ** 0.5)
    x, g = dev(rnd((N, Hi, Wi, I), 2)), dev(rnd((N, Hi // 2, Wi // 2, O), 3))
    z, gy0 = dev(rnd((96, 128), 4)), dev(rnd((96, 4, 4, 96), 5))
    w0 = rnd((128, 96, 4, 4), 6, 0.05)
    lx, lw = dev(rnd((200, 300), 7)), dev(rnd((150, 300), 8, 0.05))
    outs = {}
    try:
        for on in (0, 1, 2):
            _abi.check(lib.rg_set_option(b"f32mma", on), "rg_set_option")
            _, ch = cwpair_tm(w)
            _, c0 = cwpair(w0)
            c0.dw = torch.zeros_like(c0.w)
            hip.conv_wgrad(g, x, ch, False)
            hip.g0_wgrad(z, gy0, c0.dw, False)
            outs[on] = [hip.conv_down(x, ch), hip.conv_up(g, ch), ch.dw.clone(), hip.g0_fwd(z, c0), c0.dw.clone(),
                        hip.linear_affine_act(lx, lw, None, None, 1.0)]
    finally:
        lib.rg_set_option(b"f32mma", -1)
    for name, a, b in zip(("conv_down", "conv_up", "conv_wgrad", "g0_fwd", "g0_wgrad", "linear"), outs[1], outs[0]):
        check(a, b, 2e-6, name)
    for name, a, b in zip(("conv_down", "conv_up", "conv_wgrad", "g0_fwd", "g0_wgrad", "linear"), outs[2], outs[0]):
        # (one MFMA adds 16 k positions x 6 plane products into the accumulator: a coarser association than the f32 chain's
        # two k per step -- measured 2.1e-6 at K = 1024, the 128 x 128-tile case)
        check(a, b, 4e-6, name + " (six bf16 products)")


CONV8_CASES = [
    # N, Hi, Wi, I, O     (conv_down: M = N*Hi*Wi/4 rows, O columns, K = 16*I; conv_up: same M per class, I columns, K = 4*O)
    (2, 32, 32, 64, 256),      # down: 256x256 tile, 2 whole row tiles, 16 k-tiles
    (5, 16, 16, 128, 256),     # down: 256x256 tile with a ragged last row tile (320 rows), 32 k-tiles -> split-K when forced
    (3, 32, 32, 64, 128),      # down: 512x128 tile, ragged (768 rows); up: 64 columns (not a conv8 shape)
    (3, 32, 32, 128, 256),     # up: 512x128 tile (I = 128 columns), ragged, 16 k-tiles; down: 256x256
    (2, 32, 32, 256, 128),     # up: 256x256 tile (I = 256 columns), 8 k-tiles; down: 512x128
]


@pytest.mark.parametrize("mfma", [16, 32])
@pytest.mark.parametrize("mode", [1, 5, 7])
@pytest.mark.parametrize("N,Hi,Wi,I,O", CONV8_CASES)
def test_conv8_pingpong_kernel(N, Hi, Wi, I, O, mode, mfma):
    """The 8-wave ping-pong conv kernel (rg_conv8.hip) forced on (conv8 = 5 / 7: also with split-K; 7: parity classes
    fastest) at shapes that cover both tiles, both MFMA shapes, ragged row tiles, all four output-parity classes, the
    fused LeakyReLU mask and the BatchNorm partial sums -- against the torch twin."""
    from rna_gan_amd import _abi
    lib = _abi.load()
    dtype = torch.bfloat16
    ref, hip = RefOps(dtype), _hip(dtype)
    try:
        _abi.check(lib.rg_set_option(b"conv8", mode), "rg_set_option")
        _abi.check(lib.rg_set_option(b"conv8_mfma", mfma), "rg_set_option")
        _abi.check(lib.rg_set_option(b"conv8_blocks", 8), "rg_set_option")      # small grids: split-K already at 8 tiles
        _abi.check(lib.rg_set_option(b"narrow8", 1), "rg_set_option")           # 64-column conv_up: all classes per block
        w = rnd((O, I, 4, 4), 1, (2.0 / (I * 16)) ** 0.5)
        cr, ch = cwpair_tm(w)
        x = rnd((N, Hi, Wi, I), 2).to(dtype)
        g = rnd((N, Hi // 2, Wi // 2, O), 3).to(dtype)
        m = rnd((N, Hi, Wi, I), 21).to(dtype)
        y, st = hip.conv_down(dev(x), ch, want_stats=True)
        check(y, ref.conv_down(x, cr), TOL[dtype], "conv_down")
        u, su = hip.conv_up(dev(g), ch, want_stats=True)
        check(u, ref.conv_up(g, cr), TOL[dtype], "conv_up")
        check(hip.conv_up(dev(g), ch, dev(m), 0.2), ref.conv_up(g, cr, m, 0.2), TOL[dtype], "conv_up(masked)")
        for name, yy, ss in (("conv_down", y, st), ("conv_up", u, su)):
            if ss is None:
                continue
            yf = yy.float().reshape(-1, yy.shape[-1])
            check(ss[:, 0, :].sum(0), yf.sum(0), 1e-4, name + " epilogue sum")
            check(ss[:, 1, :].sum(0), (yf * yf).sum(0), 1e-4, name + " epilogue sumsq")
    finally:
        for k in (b"conv8", b"conv8_mfma", b"conv8_blocks", b"narrow8"):
            lib.rg_set_option(k, -1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W,Cout", [(2, 4, 16, 3),       # ONE tile row: first and last padded row in the same tile
                                        (3, 16, 16, 3), (2, 32, 64, 1), (5, 64, 64, 8), (2, 128, 128, 3)])
def test_image_block_without_the_materialised_upsample(N, H, W, Cout, dtype):
    """rg_upimg.hip: the resize-convolution generator's image block with upsample + reflection pad formed in LDS --
    against the path that materialises the padded image (option upimg = 0): the same roundings in the same order, so the
    two outputs are bit-identical; and against torch's interpolate + pad + conv2d in fp64."""
    hip = _hip(dtype)
    lib = hip.lib
    w = rnd((Cout, 64, 3, 3), 1, (2.0 / (9 * 64)) ** 0.5)
    b = rnd((Cout,), 2, 0.1)
    x = rnd((N, H, W, 64), 3).to(dtype)
    cw = ConvW(dev(w), dev(b), torch.zeros_like(dev(w)))
    outs = {}
    try:
        for on in (1, 0):
            _abi_check = lib.rg_set_option(b"upimg", on)
            assert _abi_check == 0
            outs[on] = hip.upconv3(dev(x), cw, dev(b), out_nchw=True)
    finally:
        lib.rg_set_option(b"upimg", -1)
    assert torch.equal(outs[1], outs[0])
    up = torch.nn.functional.interpolate(x.double().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    pad = torch.nn.functional.pad(up, (1, 1, 1, 1), mode="reflect")
    ref = torch.nn.functional.conv2d(pad, w.to(dtype).double(), b.double())
    check(outs[1], ref, 1.5e-2 if dtype == torch.bfloat16 else 2e-3, "image block")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W,Cout", [(2, 4, 16, 3), (3, 16, 16, 3), (2, 32, 64, 1), (5, 64, 64, 4), (8, 128, 128, 3)])
def test_image_block_weight_gradient_without_the_materialised_upsample(N, H, W, Cout, dtype):
    """rg_upimg.hip, weight gradient: the padded tile formed in LDS and contracted over its pixels through transposed LDS reads
    (ds_read_b64_tr_b16), per-workgroup partial sums reduced in a fixed order -- against fp64 (unfold of torch's interpolate + pad,
    operands rounded as the kernels round them), overwrite and accumulate; and within the same bound as the materialising path."""
    hip = _hip(dtype)
    lib = hip.lib
    w = rnd((Cout, 64, 3, 3), 1, 0.04)
    x = rnd((N, H, W, 64), 3).to(dtype)
    gy = rnd((N, Cout, 2 * H, 2 * W), 4)
    up = torch.nn.functional.interpolate(x.double().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    pad = torch.nn.functional.pad(up.to(dtype).double(), (1, 1, 1, 1), mode="reflect")
    ref = torch.einsum("nohw,nchwyx->ocyx", gy.to(dtype).double(), pad.unfold(2, 3, 1).unfold(3, 3, 1))
    got = {}
    try:
        for on in (1, 0):
            assert lib.rg_set_option(b"upimg", on) == 0
            cw = ConvW(dev(w), None, torch.full_like(dev(w), 5.0))
            hip.upconv3_wgrad(dev(gy), dev(x), cw, False, gy_nchw=True)
            once = cw.dw.clone()
            hip.upconv3_wgrad(dev(gy), dev(x), cw, True, gy_nchw=True)
            got[on] = (once, cw.dw.clone())
    finally:
        lib.rg_set_option(b"upimg", -1)
    for on in (1, 0):
        check(got[on][0], ref, 1e-5, "dw (upimg = %d)" % on)
        check(got[on][1], 2 * ref, 1e-5, "dw accumulated (upimg = %d)" % on)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W,Cout", [(2, 4, 16, 3), (3, 16, 16, 3), (2, 32, 64, 1), (5, 64, 64, 2), (4, 128, 128, 3)])
def test_image_block_data_gradient_without_the_materialised_upsample(N, H, W, Cout, dtype):
    """rg_upimg.hip, data gradient: transposed 3 x 3 conv at the tile's padded positions (one MFMA k-step over (channel, tap)) into
    LDS, then the adjoint of reflection pad o bilinear x2 -- against torch autograd through interpolate + pad + conv2d in fp64
    (image borders included: the one-tile-row shape puts the first and the last padded row into the same tile), and no further
    from it than the path that materialises the padded-grid gradient."""
    hip = _hip(dtype)
    lib = hip.lib
    w = rnd((Cout, 64, 3, 3), 1, 0.04)
    gy = rnd((N, Cout, 2 * H, 2 * W), 4)
    x = torch.zeros(N, 64, H, W, dtype=torch.float64, requires_grad=True)
    up = torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    yy = torch.nn.functional.conv2d(torch.nn.functional.pad(up, (1, 1, 1, 1), mode="reflect"), w.to(dtype).double())
    (ref,) = torch.autograd.grad(yy, x, gy.to(dtype).double())
    ref = ref.permute(0, 2, 3, 1)
    cw = ConvW(dev(w), None, torch.zeros_like(dev(w)))
    errs = {}
    try:
        for on in (1, 0):
            assert lib.rg_set_option(b"upimg", on) == 0
            gx = hip.upconv3_bwd_data(dev(gy), cw, gy_nchw=True)
            assert gx.dtype == dtype and tuple(gx.shape) == (N, H, W, 64)
            errs[on] = relerr(gx, ref)
    finally:
        lib.rg_set_option(b"upimg", -1)
    tol = 6e-3 if dtype == torch.bfloat16 else 8e-4          # the 16-bit rounding of the result (+ of the padded-grid gradient)
    assert errs[1] <= tol and errs[1] <= 1.5 * errs[0] + 1e-4, errs


def test_f32_mode_non_finite_operands_stay_non_finite_and_local():
    """fp32 mode forms products from three-way bf16 splits (f32mma = 2, and the plane kernels).  A non-finite operand keeps its
    class in the split (h = +-inf / NaN, residuals 0 -- not inf - inf), so every output IEEE fp32 would make non-finite is
    non-finite here too and nothing else is touched.  What is NOT kept: +-inf may come out as NaN (inf times a residual plane that
    is exactly zero), as stated in DESIGN 6.3."""
    hip = _hip(torch.float32)
    lx, lw = rnd((256, 512), 7), rnd((256, 512), 8, 0.05)
    lx[3, 7] = float("inf")
    out = hip.linear_affine_act(dev(lx), dev(lw), None, None, 1.0).cpu()
    assert not torch.isfinite(out[3]).any()
    assert torch.isfinite(torch.cat([out[:3], out[4:]])).all()
    x = rnd((4, 128, 128, 64), 2)
    x[1, 5, 9, 3] = float("-inf")
    _, ch = cwpair_tm(rnd((128, 64, 4, 4), 1, 0.05))
    y = hip.conv_down(dev(x), ch).cpu()
    bad = ~torch.isfinite(y)
    assert bad[1, 2:4, 4:6].all() and int(bad.sum()) == 4 * 128          # the 2 x 2 output pixels whose 4 x 4 window holds (5, 9)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("blocks", [256, 3])
@pytest.mark.parametrize("N,Ws", [(1, 16), (3, 16), (2, 32), (1, 64), (5, 64)])
def test_convp_patch_resident_kernel(N, Ws, blocks, dtype):
    """The 128 -> 64 channel transposed conv with the input patch resident in LDS (rg_convp.hip), all three image widths,
    one and several tiles per workgroup (blocks = 3: persistent loop with the cross-tile prefetch): plain, BatchNorm partial
    sums, fused LeakyReLU mask from PACKED sign bits (rg_sign_pack), folded affine -- against the torch twin, and
    bit-identical to the implicit-GEMM kernel it replaces."""
    from rna_gan_amd import _abi
    ref, hip = RefOps(dtype), _hip(dtype)
    lib = hip.lib                                            # (the fp16 build is a library of its own, with its own option table)
    O, I = 128, 64
    assert lib.rg_conv_up_maskbits_supported(N, Ws, Ws, O, I, hip.H16, 0) == 1
    w = rnd((O, I, 4, 4), 1, (2.0 / (O * 4)) ** 0.5)
    cr, ch = cwpair_tm(w)
    g = rnd((N, Ws, Ws, O), 3).to(dtype)
    m = rnd((N, 2 * Ws, 2 * Ws, I), 21).to(dtype)
    m[0, 0, :4, :8] = 0.0                                   # +0 / -0 count as "not positive"
    m[0, 1, :4, :8] = -0.0
    try:
        _abi.check(lib.rg_set_option(b"convp_blocks", blocks), "rg_set_option")
        outs = {}
        for on in (1, 0):
            _abi.check(lib.rg_set_option(b"convp", on), "rg_set_option")
            u, su = hip.conv_up(dev(g), ch, want_stats=True)
            md = dev(m)
            if on:
                md._rg_sign_bits = hip.sign_pack(md)
            um = hip.conv_up(dev(g), ch, md, 0.2)
            outs[on] = (u, su, um)
        u, su, um = outs[1]
        check(u, ref.conv_up(g, cr), TOL[dtype], "conv_up")
        check(um, ref.conv_up(g, cr, m, 0.2), TOL[dtype], "conv_up(packed mask)")
        assert torch.equal(u, outs[0][0]) and torch.equal(um, outs[0][2])
        assert su is not None and su.shape[0] == 4 * (N * Ws * Ws // 256) * 2
        uf = u.float().reshape(-1, I)
        check(su[:, 0, :].sum(0), uf.sum(0), 1e-4, "epilogue sum")
        check(su[:, 1, :].sum(0), (uf * uf).sum(0), 1e-4, "epilogue sumsq")
        # sign bits: bit c of word p = activation[p][c] > 0
        bits = hip.sign_pack(dev(m)).cpu().reshape(-1)
        mm = (m.float().reshape(-1, 64) > 0)
        expect = torch.zeros(mm.shape[0], dtype=torch.int64)
        for c in range(64):
            expect |= mm[:, c].to(torch.int64) << c
        assert torch.equal(bits, expect)
        # folded affine epilogue (generator-only inference)
        sc, sh = rnd((I,), 31).abs() + 0.5, rnd((I,), 32)
        ua = hip.conv_up_affine(dev(g), ch, dev(sc), dev(sh), 0.2)
        _abi.check(lib.rg_set_option(b"convp", 0), "rg_set_option")
        ub = hip.conv_up_affine(dev(g), ch, dev(sc), dev(sh), 0.2)
        assert torch.equal(ua, ub)
    finally:
        for k in (b"convp", b"convp_blocks"):
            lib.rg_set_option(k, -1)


@pytest.mark.parametrize("blocks", [256, 3])
@pytest.mark.parametrize("N,Hs", [(1, 128), (3, 128), (2, 32), (5, 8)])
def test_convd_plane_resident_kernel(N, Hs, blocks):
    """The 64 -> 128 channel stride-2 conv on a 128-pixel-wide input with the input's parity planes resident in LDS
    (rg_convd.hip): one and several tiles per workgroup (blocks = 3: persistent loop with the cross-tile prefetch), image
    heights down to one tile per image (every row of the tile touches the padding), plain and with BatchNorm partial sums --
    against the torch twin and against the implicit-GEMM kernel it replaces (same products, another summation order)."""
    from rna_gan_amd import _abi
    lib = _abi.load()
    dtype = torch.bfloat16
    ref, hip = RefOps(dtype), _hip(dtype)
    O, I, Ws = 128, 64, 128
    w = rnd((O, I, 4, 4), 1, (2.0 / (I * 16)) ** 0.5)
    cr, ch = cwpair_tm(w)
    x = rnd((N, Hs, Ws, I), 2).to(dtype)
    M = N * (Hs // 2) * (Ws // 2)
    try:
        _abi.check(lib.rg_set_option(b"conv8_blocks", 1), "rg_set_option")      # (the kernel is selected from 256 tiles up)
        _abi.check(lib.rg_set_option(b"convd_blocks", blocks), "rg_set_option")
        outs = {}
        for on in (1, 0):
            _abi.check(lib.rg_set_option(b"convd", on), "rg_set_option")
            y, st = hip.conv_down(dev(x), ch, want_stats=True)
            outs[on] = (y, st, hip.conv_down(dev(x), ch))
        y, st, yp = outs[1]
        y_ref = ref.conv_down(x, cr)
        check(y, y_ref, TOL[dtype], "conv_down")
        assert torch.equal(y, yp)
        check(y, outs[0][0], 4e-3, "conv_down against the implicit-GEMM kernel")
        assert st is not None and st.shape[0] == M // 64, (None if st is None else st.shape, M // 64)
        yf = y.float().reshape(-1, O)
        check(st[:, 0, :].sum(0), yf.sum(0), 1e-4, "epilogue sum")
        check(st[:, 1, :].sum(0), (yf * yf).sum(0), 1e-4, "epilogue sumsq")
    finally:
        for k in (b"convd", b"convd_blocks", b"conv8_blocks"):
            lib.rg_set_option(k, -1)


def test_first_down_sign_bits():
    """first_down writes the packed sign bits of its output itself (discriminator layer 0); they equal rg_sign_pack of the
    stored activation, and the data-gradient conv of layer 1 picks them up from the tensor."""
    dtype = torch.bfloat16
    ref, hip = RefOps(dtype), _hip(dtype)
    for N, H in ((2, 64), (1, 128), (1, 256)):
        x = rnd((N, 3, H, H), 5)
        w = rnd((64, 3, 4, 4), 6, 0.2)
        b = rnd((64,), 7, 0.1)
        cr, ch = cwpair(w)
        a = hip.first_down(dev(x), ch, dev(b), 0.2)
        bits = getattr(a, "_rg_sign_bits", None)
        assert bits is not None and bits.shape == (N, H // 2, H // 2)
        assert torch.equal(bits, hip.sign_pack(a))
        check(a, ref.first_down(x, cr, b, 0.2), TOL[dtype], "first_down")
        a1 = hip.first_down(dev(x), ch, dev(b), 1.0)                    # slope 1: no consumer for the bits
        assert getattr(a1, "_rg_sign_bits", None) is None
        w1 = rnd((128, 64, 4, 4), 8, (2.0 / 512) ** 0.5)
        c1r, c1h = cwpair_tm(w1)
        gz = rnd((N, H // 4, H // 4, 128), 9).to(dtype)
        check(hip.conv_up(dev(gz), c1h, a, 0.2), ref.conv_up(gz, c1r, a.cpu(), 0.2), TOL[dtype], "conv_up(mask from first_down)")
        # tangent of layer 0: lrelu'(a) * conv(v) in one kernel (mask from the packed bits) against conv + masking pass
        v = rnd((N, 3, H, H), 10)
        t1 = hip.first_down_tangent(dev(v), ch, a, 0.2)
        a_nobits = a.clone()
        t0 = hip.first_down_tangent(dev(v), ch, a_nobits, 0.2)
        check(t1, t0, 1e-2, "first_down_tangent")
        check(t1, ref.first_down_tangent(v, cr, a.cpu(), 0.2), TOL[dtype], "first_down_tangent vs twin")


@pytest.mark.parametrize("blocks", [1, 8, 256])
@pytest.mark.parametrize("N,Hi,Wi,I,O", [(2, 32, 32, 64, 256),      # 8 k-tiles of 64 pixels, 1 x 4 output tiles
                                         (3, 16, 16, 128, 256),     # 192 pixels per segment: ragged last k-tile
                                         (5, 16, 16, 64, 512),      # 320 pixels: two row tiles of output channels
                                         (1, 32, 32, 256, 256)])    # one tap per 256-column tile
def test_wgrad8_pingpong_kernel(N, Hi, Wi, I, O, blocks):
    """The 8-wave ping-pong weight-gradient kernel (rg_wgrad8.hip): one and two segments, overwrite and accumulate,
    direct write (one split) and split-K slabs, ragged pixel counts -- against the torch twin."""
    from rna_gan_amd import _abi
    lib = _abi.load()
    dtype = torch.bfloat16
    ref, hip = RefOps(dtype), _hip(dtype)
    try:
        _abi.check(lib.rg_set_option(b"wgrad8", 1), "rg_set_option")
        _abi.check(lib.rg_set_option(b"wgrad8_blocks", blocks), "rg_set_option")
        w = rnd((O, I, 4, 4), 1, (2.0 / (I * 16)) ** 0.5)
        cr, ch = cwpair_tm(w)
        x = rnd((N, Hi, Wi, I), 2).to(dtype)
        g = rnd((N, Hi // 2, Wi // 2, O), 3).to(dtype)
        ref.conv_wgrad(g, x, cr, False)
        dw_ref = cr.dw.clone()
        hip.conv_wgrad(dev(g), dev(x), ch, False)          # overwrites the 7.0 fill
        check(ch.dw, dw_ref, TOL[dtype] * 2, "conv_wgrad")
        hip.conv_wgrad(dev(g), dev(x), ch, True)
        check(ch.dw, 2 * dw_ref, TOL[dtype] * 2, "conv_wgrad(accumulate)")
        if (N * Hi * Wi // 4) % 64 == 0:                   # two segments need whole 64-pixel k-tiles per segment
            g2, x2 = rnd((N, Hi // 2, Wi // 2, O), 13).to(dtype), rnd((N, Hi, Wi, I), 12).to(dtype)
            ref.conv_wgrad(g, x, cr, False); ref.conv_wgrad(g2, x2, cr, True)
            ch.dw.fill_(-5.0)
            hip.conv_wgrad2(dev(g), dev(x), dev(g2), dev(x2), ch, False)
            check(ch.dw, cr.dw, TOL[dtype] * 2, "conv_wgrad2")
            hip.conv_wgrad2(dev(g), dev(x), dev(g2), dev(x2), ch, True)
            check(ch.dw, 2 * cr.dw, TOL[dtype] * 2, "conv_wgrad2(accumulate)")
    finally:
        for k in (b"wgrad8", b"wgrad8_blocks"):
            lib.rg_set_option(k, -1)


def test_u8_to_norm_bit_exact():
    """a15 input contract (src/histopathology_gan.py:106-109): device-side uint8 -> (x / 255 - 0.5) / 0.5 equals the
    host transform bit for bit, including a length that is not a multiple of 16."""
    from rna_gan_amd import synth
    hip = _hip(torch.bfloat16)
    u8 = synth.synthetic_tiles_u8(3, 64, 11)
    want = synth.synthetic_images(3, 64, 11)
    got = hip.u8_to_norm(u8.cuda())
    assert got.dtype == torch.float32 and torch.equal(got.cpu(), want)
    allv = torch.arange(256, dtype=torch.uint8).repeat(3)[:-5].contiguous()        # every byte value, ragged length
    assert torch.equal(hip.u8_to_norm(allv.cuda()).cpu(), (allv.float() / 255.0 - 0.5) / 0.5)


@pytest.mark.parametrize("O,dtype,W", [(4, torch.float32, 32), (4, torch.bfloat16, 32), (64, torch.float32, 32),
                                       (64, torch.bfloat16, 32), (128, torch.bfloat16, 32),
                                       (64, torch.bfloat16, 128), (64, torch.bfloat16, 64),    # these two: bf16 MFMA path
                                       # fp32 on the f32 matrix cores with structured operands (as (64, float32, 32) above)
                                       (64, torch.float32, 128), (128, torch.float32, 64)])
def test_image_side_layers(O, dtype, W):
    ref, hip = RefOps(dtype), _hip(dtype)
    N, H, I = 3, 16, 3
    w = rnd((O, I, 4, 4), 4, 0.2)
    b = rnd((O,), 5, 0.1)
    cr, ch = cwpair(w)
    x = rnd((N, I, H, W), 6)
    check(hip.first_down(dev(x), ch, dev(b), 0.2), ref.first_down(x, cr, b, 0.2), TOL[dtype], "first_down")
    check(hip.first_down(dev(x), ch, None, 1.0), ref.first_down(x, cr, None, 1.0), TOL[dtype], "first_down(raw)")
    a = rnd((N, H // 2, W // 2, O), 7).to(dtype)
    b3 = rnd((I,), 8, 0.1)
    check(hip.last_up(dev(a), ch, dev(b3), True), ref.last_up(a, cr, b3, True), TOL[dtype], "last_up(tanh)")
    check(hip.last_up(dev(a), ch, None, False), ref.last_up(a, cr, None, False), TOL[dtype], "last_up(raw)")
    dw_ref = torch.zeros(O, I, 4, 4)
    ref.skinny_wgrad(a, x, dw_ref, False)
    dw = torch.full((O, I, 4, 4), 3.0).cuda()
    hip.skinny_wgrad(dev(a), dev(x), dw, False)
    check(dw, dw_ref, TOL[dtype] * 2, "skinny_wgrad")
    hip.skinny_wgrad(dev(a), dev(x), dw, True)
    check(dw, 2 * dw_ref, TOL[dtype] * 2, "skinny_wgrad(acc)")


@pytest.mark.parametrize("N,E,C,dtype", [(5, 24, 8, torch.float32), (5, 24, 8, torch.bfloat16),
                                         (6, 128, 64, torch.bfloat16), (64, 256, 128, torch.bfloat16)])
def test_g0_and_head(N, E, C, dtype):
    ref, hip = RefOps(dtype), _hip(dtype)
    w = rnd((E, C, 4, 4), 9, (2.0 / (C * 16)) ** 0.5)
    cr, ch = cwpair(w)
    z = rnd((N, E), 10)
    check(hip.g0_fwd(dev(z), ch), ref.g0_fwd(z, cr), TOL[dtype], "g0_fwd")
    gy = rnd((N, 4, 4, C), 11).to(dtype)
    dw_ref = torch.zeros(E, C, 4, 4)
    ref.g0_wgrad(z, gy, dw_ref, False)
    dw = torch.zeros(E, C, 4, 4).cuda()
    hip.g0_wgrad(dev(z), dev(gy), dw, False)
    check(dw, dw_ref, TOL[dtype] * 2, "g0_wgrad")
    # head
    wh = rnd((1, C, 4, 4), 12, 0.1)
    hr, hh = cwpair(wh)
    a = rnd((N, 4, 4, C), 13).to(dtype)
    h_ref, o_ref = ref.head_fwd(a, hr, 0.2)
    h, o = hip.head_fwd(dev(a), hh, 0.2)
    check(h, h_ref, TOL[dtype], "head_fwd.h"); check(o, o_ref, TOL[dtype], "head_fwd.out")
    gh_ref = ref.head_grad(h_ref, -0.25, 0.2)
    gh = hip.head_grad(dev(h_ref), -0.25, 0.2)
    check(gh, gh_ref, 1e-6, "head_grad")
    check(hip.head_bwd_data(gh, hh), ref.head_bwd_data(gh_ref, hr), TOL[dtype], "head_bwd_data")
    dwh_ref = torch.zeros(1, C, 4, 4)
    ref.head_wgrad(gh_ref, a, dwh_ref, False)
    dwh = torch.zeros(1, C, 4, 4).cuda()
    hip.head_wgrad(gh, dev(a), dwh, False)
    check(dwh, dwh_ref, TOL[dtype], "head_wgrad")


@pytest.mark.parametrize("M,C,dtype", [(40, 8, torch.float32), (37, 6, torch.float32), (40, 8, torch.bfloat16),
                                       (4096, 128, torch.float32), (4096, 128, torch.bfloat16),
                                       (70000, 64, torch.bfloat16),
                                       # single-launch (fused) path of the small deep layers
                                       (1024, 2048, torch.bfloat16), (4096, 1024, torch.bfloat16),
                                       (256, 512, torch.float32)])
def test_bn_family(M, C, dtype, monkeypatch):
    if M * C >= 256 * 512 and C >= 512:
        monkeypatch.setenv("RNAGAN_BN_FUSED", "1")      # opt-in single-launch kernels: keep them covered
    ref, hip = RefOps(dtype), _hip(dtype)
    tol = TOL[dtype] if dtype == torch.bfloat16 else 1e-4
    z = (rnd((1, M, 1, C), 20) * 1.5 + 0.3).to(dtype)
    gamma, beta = 1 + 0.1 * rnd((C,), 21), 0.1 * rnd((C,), 22)
    rm, rv = 0.1 * rnd((C,), 23), 1 + 0.1 * rnd((C,), 24).abs()
    nbt = torch.zeros((), dtype=torch.int64)
    s_ref, ss_ref = ref.bn_stats(z)
    s, ss = hip.bn_stats(dev(z))
    check(s, s_ref, 1e-4, "bn_stats.sum"); check(ss, ss_ref, 1e-4, "bn_stats.sumsq")
    rm_d, rv_d, nbt_d = dev(rm.clone()), dev(rv.clone()), dev(nbt.clone())
    mean_ref, inv_ref = ref.bn_finalize(s_ref, ss_ref, M, 1e-5, 0.1, rm, rv, nbt)
    mean, inv = hip.bn_finalize(dev(s_ref), dev(ss_ref), M, 1e-5, 0.1, rm_d, rv_d, nbt_d)
    check(mean, mean_ref, 1e-6, "mean"); check(inv, inv_ref, 1e-5, "invstd")
    check(rm_d, rm, 1e-6, "running_mean"); check(rv_d, rv, 1e-5, "running_var")
    assert int(nbt_d.cpu()) == int(nbt) == 1
    # fused statistics + finalize (what the engine's BatchNorm forward uses): second update of the running stats
    mean_ref2, inv_ref2 = ref.bn_stats_finalize(z, 1e-5, 0.1, rm, rv, nbt)
    mean2, inv2 = hip.bn_stats_finalize(dev(z), 1e-5, 0.1, rm_d, rv_d, nbt_d)
    check(mean2, mean_ref2, 1e-5, "fused mean"); check(inv2, inv_ref2, 1e-4, "fused invstd")
    check(rm_d, rm, 1e-5, "fused running_mean"); check(rv_d, rv, 1e-4, "fused running_var")
    assert int(nbt_d.cpu()) == int(nbt) == 2
    D = lambda t: None if t is None else dev(t)
    # whole forward in one call (one launch for small tensors): third update of the running statistics
    a_ref3, mean_ref3, inv_ref3 = ref.bn_forward(z, gamma, beta, 0.2, 1e-5, 0.1, rm, rv, nbt)
    a3, mean3, inv3 = hip.bn_forward(dev(z), dev(gamma), dev(beta), 0.2, 1e-5, 0.1, rm_d, rv_d, nbt_d)
    check(mean3, mean_ref3, 1e-5, "bn_forward mean"); check(inv3, inv_ref3, 1e-4, "bn_forward invstd")
    check(a3, a_ref3, tol, "bn_forward a"); check(rv_d, rv, 1e-4, "bn_forward running_var")
    assert int(nbt_d.cpu()) == int(nbt) == 3
    a_ref = ref.bn_act(z, mean_ref, inv_ref, gamma, beta, 0.2)
    check(hip.bn_act(D(z), D(mean_ref), D(inv_ref), D(gamma), D(beta), 0.2), a_ref, tol, "bn_act")
    ga = rnd((1, M, 1, C), 25).to(dtype)
    dg_ref, db_ref = torch.zeros(C), torch.zeros(C)
    gz_ref, sgy_ref, sgx_ref = ref.bn_act_bwd(z, ga, mean_ref, inv_ref, gamma, beta, 0.2, dg_ref, db_ref, False)
    dg, db = torch.ones(C).cuda(), torch.ones(C).cuda()
    gz, sgy, sgx = hip.bn_act_bwd(D(z), D(ga), D(mean_ref), D(inv_ref), D(gamma), D(beta), 0.2, dg, db, False)
    check(gz, gz_ref, tol, "bn_act_bwd.gz"); check(sgy, sgy_ref, 2e-4, "s_gy"); check(sgx, sgx_ref, 2e-4, "s_gyxh")
    check(dg, dg_ref, 2e-4, "dgamma"); check(db, db_ref, 2e-4, "dbeta")
    zt = rnd((1, M, 1, C), 26).to(dtype)
    at_ref, szt_ref, sxz_ref = ref.bn_tangent(z, zt, mean_ref, inv_ref, gamma, beta, 0.2)
    at, szt, sxz = hip.bn_tangent(D(z), D(zt), D(mean_ref), D(inv_ref), D(gamma), D(beta), 0.2)
    check(at, at_ref, tol, "bn_tangent.at"); check(szt, szt_ref, 2e-4, "s_zt"); check(sxz, sxz_ref, 2e-4, "s_xhzt")
    for qa in (None, rnd((1, M, 1, C), 27).to(dtype)):
        dg_ref, db_ref = torch.zeros(C), torch.zeros(C)
        pz_ref = ref.bn_double_bwd(z, qa, zt, ga, mean_ref, inv_ref, gamma, beta, 0.2, sgy_ref, sgx_ref, szt_ref,
                                   sxz_ref, dg_ref, db_ref, False)
        dg, db = torch.ones(C).cuda(), torch.ones(C).cuda()
        pz = hip.bn_double_bwd(D(z), D(qa), D(zt), D(ga), D(mean_ref), D(inv_ref), D(gamma), D(beta), 0.2,
                               D(sgy_ref), D(sgx_ref), D(szt_ref), D(sxz_ref), dg, db, False)
        check(pz, pz_ref, 5 * tol, "bn_double_bwd.pz")
        check(dg, dg_ref, 1e-3, "dbl.dgamma"); check(db, db_ref, 1e-3 if qa is not None else 1e-30, "dbl.dbeta")
    check(hip.lrelu_bwd(D(ga), D(a_ref), 0.2), ref.lrelu_bwd(ga, a_ref, 0.2), tol, "lrelu_bwd")
    out_ref, out = torch.ones(C), torch.ones(C).cuda()
    ref.col_sum(ga, out_ref, True); hip.col_sum(D(ga), out, True)
    check(out, out_ref, 2e-4, "col_sum")


def test_pointwise_reductions_adam():
    ref, hip = RefOps(torch.float32), _hip(torch.float32)
    n = (3, 3, 20, 24)
    a, b = rnd(n, 30), torch.tanh(rnd(n, 31))
    check(hip.tanh_bwd(dev(a), dev(b)), ref.tanh_bwd(a, b), 1e-6, "tanh_bwd")
    check(hip.interp(dev(a), dev(b), 0.37), ref.interp(a, b, 0.37), 1e-6, "interp")
    o_ref, o = torch.ones(3), torch.ones(3).cuda()
    ref.nchw_chan_sum(a, o_ref, True); hip.nchw_chan_sum(dev(a), o, True)
    check(o, o_ref, 1e-5, "nchw_chan_sum")
    big = rnd((1, 3, 300, 301), 32)
    sq_ref, sq = ref.sqnorm(big), hip.sqnorm(dev(big))
    check(sq, sq_ref, 1e-5, "sqnorm")
    l_ref, c_ref = ref.gp_coef(sq_ref, 10.0)
    l, c = hip.gp_coef(dev(sq_ref), 10.0)
    check(l, l_ref, 1e-5, "gp loss"); check(c, c_ref, 1e-5, "gp coef")
    check(hip.scale_by(dev(a), dev(c_ref)), ref.scale_by(a, c_ref), 1e-6, "scale_by")
    v1, v2 = rnd((37,), 33), rnd((37,), 34)
    check(hip.mean_diff(dev(v1), dev(v2), 1.0), ref.mean_diff(v1, v2, 1.0), 1e-5, "mean_diff")
    check(hip.mean_diff(dev(v1), None, -1.0), ref.mean_diff(v1, None, -1.0), 1e-5, "mean_diff(neg)")
    u, z = rnd((7, 50), 35, 0.17), rnd((7, 50), 36)
    check(hip.latent_prep(dev(u), dev(z)), ref.latent_prep(u, z), 1e-5, "latent_prep")
    # adam: 3 steps against torch.optim.Adam itself
    p0, g0 = rnd((1003,), 37), rnd((1003,), 38)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=4e-4, betas=(0.5, 0.999))
    p, m, v = dev(p0.clone()), torch.zeros(1003).cuda(), torch.zeros(1003).cuda()
    for step in range(1, 4):
        g = g0 * step
        pt.grad = g.clone(); opt.step()
        hip.adam_step(p, dev(g), m, v, step, 4e-4, 0.5, 0.999, 1e-8)
    check(p, pt.detach(), 2e-6, "adam p")
    check(m, opt.state[pt]["exp_avg"], 2e-6, "adam m"); check(v, opt.state[pt]["exp_avg_sq"], 2e-6, "adam v")
    q = dev(p0.clone()); hip.clamp_(q, -0.01, 0.01)
    check(q, p0.clamp(-0.01, 0.01), 1e-7, "clamp")


@pytest.mark.parametrize("M,K,Nout,packed", [(6, 50, 24, False), (16, 200, 136, True), (64, 19198, 256, True)])
def test_linear(M, K, Nout, packed):
    ref, hip = RefOps(torch.float32), _hip(torch.bfloat16)
    x, w = rnd((M, K), 40), rnd((Nout, K), 41, (1.0 / K) ** 0.5)
    sc, sh = 1 + 0.1 * rnd((Nout,), 42), 0.1 * rnd((Nout,), 43)
    y_ref = ref.linear_affine_act(x, w, sc, sh, 0.01)
    wd = dev(w)
    wp = hip.pack_linear(wd) if packed else None
    y = hip.linear_affine_act(dev(x), wd, dev(sc), dev(sh), 0.01, wp=wp)
    check(y, y_ref, 1.5e-2 if packed else 2e-5, "linear")


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Cin,Cout,dtype", [(2, 4, 4, 8, 4, torch.float32), (3, 5, 3, 6, 7, torch.float32),
                                                  (2, 8, 8, 16, 8, torch.bfloat16), (2, 1, 2, 4, 3, torch.float32),
                                                  # matrix-core forward (Cin % 64 == 0): narrow / 128x128 / split-K tiles
                                                  (2, 8, 8, 64, 64, torch.bfloat16), (2, 4, 8, 128, 128, torch.bfloat16),
                                                  (4, 4, 4, 512, 256, torch.bfloat16),
                                                  # the image block's shape class (Cout = 3 from 64 channels)
                                                  (2, 8, 16, 64, 3, torch.bfloat16)])
def test_upconv3_block(N, H, W, Cin, Cout, dtype):
    """Resize-convolution block of DCGANUpGenerator (src/dcgan.py:45-56,76-84): bilinear x2 + reflection pad + 3x3
    conv, forward / data gradient / weight gradient, NHWC activation and NCHW fp32 image variants."""
    ref, hip = RefOps(dtype), _hip(dtype)
    w = rnd((Cout, Cin, 3, 3), 31, (2.0 / (Cin * 9)) ** 0.5)
    b = rnd((Cout,), 32, 0.1)
    cr = ConvW(w.clone(), b.clone(), torch.zeros_like(w))
    ch = ConvW(w.cuda(), b.cuda(), torch.full_like(w, 3.0).cuda())
    x = rnd((N, H, W, Cin), 33).to(dtype)
    tol = TOL[dtype]
    check(hip.upconv3(dev(x), ch, ch.bias), ref.upconv3(x, cr, cr.bias), tol, "upconv3")
    check(hip.upconv3(dev(x), ch, ch.bias, out_nchw=True), ref.upconv3(x, cr, cr.bias, out_nchw=True), tol,
          "upconv3(nchw)")
    gy = rnd((N, 2 * H, 2 * W, Cout), 34).to(dtype)
    gy_img = rnd((N, Cout, 2 * H, 2 * W), 35)
    check(hip.upconv3_bwd_data(dev(gy), ch), ref.upconv3_bwd_data(gy, cr), tol, "upconv3_bwd_data")
    check(hip.upconv3_bwd_data(dev(gy_img), ch, gy_nchw=True), ref.upconv3_bwd_data(gy_img, cr, gy_nchw=True), tol,
          "upconv3_bwd_data(nchw)")
    ref.upconv3_wgrad(gy, x, cr, False)
    hip.upconv3_wgrad(dev(gy), dev(x), ch, False)
    check(ch.dw, cr.dw, tol * 2, "upconv3_wgrad")
    ref.upconv3_wgrad(gy_img, x, cr, True, gy_nchw=True)
    hip.upconv3_wgrad(dev(gy_img), dev(x), ch, True, gy_nchw=True)
    check(ch.dw, cr.dw, tol * 2, "upconv3_wgrad(nchw, accumulate)")


def _fp8_decode(u8):
    return u8.cpu().view(torch.float8_e4m3fn).float()


def test_fp8_selftest_and_cast():
    hip = _hip(torch.bfloat16)
    assert hip.selftest_fp8() == [0, 0]
    x = rnd((1000,), 5, 3.0)
    q = hip.cast_fp8(x.cuda())
    want = x.to(torch.float8_e4m3fn).float()                 # torch's OCP e4m3 round-to-nearest-even
    assert torch.equal(_fp8_decode(q), want)


@pytest.mark.parametrize("mx", [1, 0])
@pytest.mark.parametrize("N,Ho,O,I,out_fp8", [(2, 16, 128, 128, False), (3, 16, 256, 256, True), (1, 32, 128, 256, False),
                                              (5, 8, 512, 128, True)])
def test_fp8_conv_up_and_gemm(N, Ho, O, I, out_fp8, mx):
    """fp8 operand kernels (conv8_kernel<.., EB = 1>) against fp32 torch arithmetic on the DEQUANTISED fp8 operands:
    transposed conv (all four parity classes, ragged row tiles) and the plain GEMM of the generator's first layer, with the
    affine + LeakyReLU epilogue and bf16 / fp8 output.  mx = 1: the MX-format matrix instruction
    (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales: same products, 2x the matrix rate); mx = 0:
    v_mfma_f32_16x16x32_fp8_fp8."""
    import torch.nn.functional as F
    hip = _hip(torch.bfloat16)
    from rna_gan_amd import _abi
    _abi.check(hip.lib.rg_set_option(b"fp8_mx", mx), "rg_set_option")
    try:
        _fp8_conv_up_and_gemm_body(hip, N, Ho, O, I, out_fp8)
    finally:
        _abi.check(hip.lib.rg_set_option(b"fp8_mx", -1), "rg_set_option")


def _fp8_conv_up_and_gemm_body(hip, N, Ho, O, I, out_fp8):
    import torch.nn.functional as F
    w = rnd((O, I, 4, 4), 1, (2.0 / (I * 16)) ** 0.5)
    _, ch = cwpair_tm(w)
    x8 = hip.cast_fp8(rnd((N, Ho, Ho, O), 2).cuda())
    scale, shift = 1 + 0.1 * rnd((I,), 3), 0.1 * rnd((I,), 4)
    if not hip.fp8_supported(N * Ho * Ho, 4 * O, I, 4):
        pytest.skip("no fp8 kernel for this shape")
    y = hip.conv_up_fp8(x8, ch, scale.cuda(), shift.cuda(), 0.2, out_fp8)
    q, s = hip.fp8_pack_up(ch)
    wq = _fp8_decode(q).reshape(4, 4, I, O) * s.cpu()[None, None, :, None]           # dequantised [kh][kw][i][o]
    ref = F.conv_transpose2d(_fp8_decode(x8).permute(0, 3, 1, 2), wq.permute(3, 2, 0, 1), stride=2, padding=1)
    ref = ref.permute(0, 2, 3, 1) * scale + shift
    ref = torch.where(ref > 0, ref, 0.2 * ref)
    got = _fp8_decode(y) if out_fp8 else y.float().cpu()
    tol = 0.07 if out_fp8 else 1.2e-2                       # one e4m3 (3 mantissa bits) / one bf16 rounding of the output
    assert float((got - ref).abs().max()) <= tol * float(ref.abs().max())
    assert float((wq - w.permute(2, 3, 1, 0)).abs().max()) <= 0.07 * float(w.abs().max())     # weight quantisation itself
    # plain GEMM: M rows x K, B[(tap, c)][E]
    M, E, C = 256, 512, 32
    g0 = ConvW(rnd((E, C, 4, 4), 7, (2.0 / E) ** 0.5).cuda())
    z8 = hip.cast_fp8(rnd((M, E), 8).cuda())
    sc, sh = 1 + 0.1 * rnd((C,), 9), 0.1 * rnd((C,), 10)
    y0 = hip.g0_fwd_fp8(z8, g0, sc.cuda(), sh.cuda(), 0.2, out_fp8)
    q0, s0 = hip.fp8_pack_g0(g0)
    ref0 = _fp8_decode(z8) @ (_fp8_decode(q0) * s0.cpu()[:, None]).t()
    ref0 = ref0 * sc.repeat(16) + sh.repeat(16)
    ref0 = torch.where(ref0 > 0, ref0, 0.2 * ref0).reshape(M, 4, 4, C)
    got0 = _fp8_decode(y0) if out_fp8 else y0.float().cpu()
    assert float((got0 - ref0).abs().max()) <= tol * float(ref0.abs().max())


@pytest.mark.parametrize("N,H,C", [(4, 8, 256), (8, 4, 2048), (16, 32, 128), (64, 64, 128)])
def test_bn_two_batch_groups(N, H, C):
    """bn_forward2 / bn_act_bwd2 (two batch groups in one set of launches) against two separate calls on the halves:
    activations, per-half statistics, running statistics (first half first), data gradient, summed parameter gradients;
    with the statistics taken from conv-epilogue style partial sums and computed by the pass itself."""
    dtype = torch.bfloat16
    hip = _hip(dtype)
    z = dev(rnd((2 * N, H, H, C), 1).to(dtype))
    ga = dev(rnd((2 * N, H, H, C), 2).to(dtype))
    gamma, beta = dev(rnd((C,), 3).abs() + 0.5), dev(rnd((C,), 4))
    M = N * H * H
    for with_partials in (False, True):
        rm1, rv1, nb1 = dev(torch.zeros(C)), dev(torch.ones(C)), torch.zeros((), dtype=torch.int64, device="cuda")
        rm2, rv2, nb2 = rm1.clone(), rv1.clone(), nb1.clone()
        parts = None
        if with_partials:                      # [rows][2][C]: one partial row per 64 tensor rows, first half first
            zf = z.float().reshape(2 * M // 64, 64, C)
            parts = torch.stack([zf.sum(1), (zf * zf).sum(1)], dim=1).contiguous()
        ref = []
        for h in range(2):
            ph = None if parts is None else parts[h * (parts.shape[0] // 2):(h + 1) * (parts.shape[0] // 2)]
            ref.append(hip.bn_forward(z[h * N:(h + 1) * N], gamma, beta, 0.2, 1e-5, 0.1, rm1, rv1, nb1, partials=ph))
        a2, mean2, invstd2 = hip.bn_forward2(z, gamma, beta, 0.2, 1e-5, 0.1, rm2, rv2, nb2, partials=parts)
        for h in range(2):
            check(a2[h * N:(h + 1) * N], ref[h][0], 1e-2, "a")
            check(mean2[h], ref[h][1], 1e-5, "mean")
            check(invstd2[h], ref[h][2], 1e-5, "invstd")
        check(rm2, rm1, 1e-6, "running_mean")
        check(rv2, rv1, 1e-6, "running_var")
        assert int(nb2) == int(nb1) == 2
        dg1, db1 = dev(torch.zeros(C)), dev(torch.zeros(C))
        dg2, db2 = dev(torch.zeros(C)), dev(torch.zeros(C))
        gz_ref = [hip.bn_act_bwd(z[h * N:(h + 1) * N], ga[h * N:(h + 1) * N], ref[h][1], ref[h][2], gamma, beta, 0.2, dg1, db1,
                                 h == 1)[0] for h in range(2)]
        gz2 = hip.bn_act_bwd2(z, ga, mean2, invstd2, gamma, beta, 0.2, dg2, db2, False)
        for h in range(2):
            check(gz2[h * N:(h + 1) * N], gz_ref[h], 1e-2, "gz")
        check(dg2, dg1, 1e-4, "dgamma")
        check(db2, db1, 1e-4, "dbeta")


@pytest.mark.parametrize("E,C", [(128, 8), (192, 24), (64, 6)])
def test_g0_weight_pack_from_bf16_shadow(E, C):
    """The transposed bf16 image of G.0's weight written from the bf16 shadow (16-byte tile kernel when E % 64 == 0 and
    16 C % 128 == 0, the 4-byte kernel otherwise) is bit-identical to the one packed from the fp32 master, and to the
    layout dst[tap * C + c][e] = w[e][c][tap]."""
    from rna_gan_amd import _abi
    lib = _abi.load()
    w = rnd((E, C, 4, 4), 71).cuda()
    shadow = w.to(torch.bfloat16).contiguous()
    a = torch.empty(16 * C, E, dtype=torch.bfloat16, device="cuda")
    b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    _abi.check(lib.rg_pack_g0_weight(w.data_ptr(), a.data_ptr(), E, C, _abi.RG_BF16, st), "rg_pack_g0_weight")
    _abi.check(lib.rg_pack_g0_weight_from_bf16(shadow.data_ptr(), b.data_ptr(), E, C, st), "rg_pack_g0_weight_from_bf16")
    torch.cuda.synchronize()
    want = shadow.view(E, C, 16).permute(2, 1, 0).reshape(16 * C, E)
    assert torch.equal(b.view(torch.int16), want.contiguous().view(torch.int16))
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("M,K,ldd", [(1, 1 << 20, 1 << 20), (1, 4096 + 8, 4096 + 8), (3, 1000, 1024), (1, 4100, 4100), (64, 256, 256)])
def test_cast_pad_matches_torch(M, K, ldd):
    """fp32 -> bf16 cast with optional row padding (rg_cast_pad: the data-parallel gradient compression and the betaVAE
    operand casts): bit-exact against torch's round-to-nearest-even on both the 8-wide and the scalar kernel."""
    from rna_gan_amd import _abi
    lib = _abi.load()
    x = rnd((M, K), 5, 3.0).cuda()
    x.view(-1)[:4] = torch.tensor([0.0, -0.0, 1e-40, 3.3895e38], device="cuda")
    out = torch.full((M, ldd), 7.0, dtype=torch.bfloat16, device="cuda")
    _abi.check(lib.rg_cast_pad(x.data_ptr(), out.data_ptr(), M, K, ldd, _abi.RG_BF16, torch.cuda.current_stream().cuda_stream),
               "rg_cast_pad")
    torch.cuda.synchronize()
    want = torch.zeros(M, ldd, dtype=torch.bfloat16, device="cuda")
    want[:, :K] = x.to(torch.bfloat16)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))


def test_multi_tensor_weight_image_transpose():
    """rg_pack_conv_wup_from_bf16_multi: the transposed-conv weight images wup[16*I][O] of several layers from their bf16
    tap-major images wdn[O][16*I] in one launch == the per-layer call == a plain transpose (exact: data movement)."""
    import ctypes as C
    from rna_gan_amd import _abi
    lib = _abi.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    shapes = [(128, 64), (256, 128), (512, 256), (64, 8), (1024, 512)]          # (O, I)
    src = [torch.randn(O, 16 * I, generator=gen).bfloat16().to(dev) for O, I in shapes]
    dst = [torch.empty(16 * I, O, dtype=torch.bfloat16, device=dev) for O, I in shapes]
    n = len(shapes)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _abi.check(lib.rg_pack_conv_wup_from_bf16_multi(
        n, (C.c_void_p * n)(*[t.data_ptr() for t in src]), (C.c_void_p * n)(*[t.data_ptr() for t in dst]),
        (C.c_int * n)(*[s[0] for s in shapes]), (C.c_int * n)(*[s[1] for s in shapes]), stream), "multi")
    torch.cuda.synchronize()
    for (O, I), s, d in zip(shapes, src, dst):
        assert torch.equal(d, s.t().contiguous()), (O, I)
        one = torch.empty_like(d)
        _abi.check(lib.rg_pack_conv_wup_from_bf16(s.data_ptr(), one.data_ptr(), O, I, stream), "single")
        torch.cuda.synchronize()
        assert torch.equal(one, d)
    # a layer that is not 64 x 128 tileable is refused (the caller uses the per-layer kernel)
    bad = torch.zeros(32, 16 * 8, dtype=torch.bfloat16, device=dev)
    rc = lib.rg_pack_conv_wup_from_bf16_multi(1, (C.c_void_p * 1)(bad.data_ptr()), (C.c_void_p * 1)(bad.data_ptr()),
                                              (C.c_int * 1)(32), (C.c_int * 1)(8), stream)
    assert rc != 0


@pytest.mark.parametrize("N,E,C", [(512, 128, 64), (200, 64, 32), (512, 2048, 2048)])
def test_fused_layer0_gradient_adam_over_gathered_factors(N, E, C):
    """rg_g0_wgrad_adam with K = the samples of ALL ranks (data parallel: the all-gathered factors z [W n, E] and gz0
    [W n, 4, 4, C], dist.G0_FACTORS) -- 8 x 64 samples, a ragged count, and the reference generator's full 2048 x 2048 x 4 x 4
    tensor: the gradient sum_n z[n][e] * gz0[n][tap][c] formed on MFMA from bf16-rounded operands with fp32 accumulation,
    then torch.optim.Adam's update (weight decay included), against the same arithmetic in plain tensor operations."""
    from rna_gan_amd import _abi
    dev = torch.device("cuda:0")
    lib = _abi.load()
    g = torch.Generator().manual_seed(N + E)
    z = torch.randn(N, E, generator=g).to(dev)
    gz0 = (torch.randn(N, 4, 4, C, generator=g) * 0.05).bfloat16().to(dev)
    p = (torch.randn(E, C, 4, 4, generator=g) * 0.02).to(dev)
    m = (torch.randn(E, C, 4, 4, generator=g) * 1e-3).to(dev)
    v = (torch.rand(E, C, 4, 4, generator=g) * 1e-4).to(dev)
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    shadow = torch.zeros(E, C, 4, 4, dtype=torch.bfloat16, device=dev)
    step = torch.full((1,), 4, dtype=torch.int32, device=dev)
    hyper = torch.zeros(12, device=dev)
    lr, b1, b2, eps, wd = 1e-3, 0.5, 0.999, 1e-8, 1e-2
    stream = torch.cuda.current_stream().cuda_stream
    _abi.check(lib.rg_adam_hyper_dev(step.data_ptr(), lr, b1, b2, eps, wd, hyper.data_ptr(), stream), "rg_adam_hyper_dev")
    assert lib.rg_g0_wgrad_adam_supported(N, E, C, _abi.RG_BF16) == 1
    _abi.check(lib.rg_g0_wgrad_adam(z.data_ptr(), gz0.data_ptr(), p.data_ptr(), m.data_ptr(), v.data_ptr(), hyper.data_ptr(),
                                    shadow.data_ptr(), N, E, C, _abi.RG_BF16, stream), "rg_g0_wgrad_adam")
    torch.cuda.synchronize()
    # dw[e][c][kh][kw] = sum_n bf16(z[n][e]) * gz0[n][kh][kw][c], fp32 accumulation
    dw = torch.einsum("ne,nhwc->echw", z.bfloat16().float(), gz0.float())
    t = 5                                              # the step the hyper kernel advanced to
    gg = dw + wd * p0
    m1 = m0 + (1 - b1) * (gg - m0)
    v1 = b2 * v0 + (1 - b2) * gg * gg
    denom = v1.sqrt() / (1 - b2 ** t) ** 0.5 + eps
    p1 = p0 - (lr / (1 - b1 ** t)) * (m1 / denom)
    scale = float(dw.abs().mean())
    assert float((m - m1).abs().max()) <= 2e-4 * (1 - b1) * scale * N ** 0.5 + 1e-6     # order of an N-term fp32 sum
    assert float(((v - v1) / (v1 + 1e-12)).abs().max()) <= 2e-3
    upd, upd_ref = p - p0, p1 - p0
    assert float((upd - upd_ref).abs().max()) <= 2e-3 * float(upd_ref.abs().max())
    assert torch.equal(shadow, p.bfloat16())


@pytest.mark.parametrize("I,O,hs,n,groups", [(64, 128, 16, 8, 1), (128, 256, 8, 8, 1), (64, 128, 16, 16, 2), (128, 256, 8, 16, 2),
                                              (256, 512, 4, 8, 1), (128, 256, 16, 8, 1), (128, 256, 16, 16, 2)])
def test_split_k_batchnorm_fusion_small_shapes(I, O, hs, n, groups):
    """The fused split-K reduction + BatchNorm kernels (rg_bn_forward_slabs / rg_bn_act_bwd_slabs) at SMALL shapes (32 x 32 and
    64 x 64 models, batch 8 / 16) and in all four conv -> BatchNorm pairings the engine uses -- stride-2 conv -> forward
    (discriminator), transposed conv -> forward (generator), transposed conv -> backward (discriminator's data gradient),
    stride-2 conv -> backward (generator's) -- against the separate-launch path of the same library: equal to one bf16
    rounding whether or not a pairing takes the slab path at that shape (tests/test_fullsize_gpu.py covers the benchmark's)."""
    from rna_gan_amd.engine import ConvW
    from rna_gan_amd.ops_hip import HipOps
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7 * I + hs + n)
    fu, se = HipOps(torch.bfloat16, dev), HipOps(torch.bfloat16, dev)
    se.split_bn = False
    w = (torch.randn(O, 4, 4, I, generator=gen) * (2.0 / (I * 16)) ** 0.5).bfloat16().float().to(dev)
    cf, cs = ConvW(w.clone(), None, torch.zeros_like(w), None, "OHWI"), ConvW(w.clone(), None, torch.zeros_like(w), None, "OHWI")
    ho = hs // 2
    x = torch.randn(n, hs, hs, I, generator=gen).bfloat16().to(dev)
    y = torch.randn(n, ho, ho, O, generator=gen).bfloat16().to(dev)
    gO, bO = (1 + 0.1 * torch.randn(O, generator=gen)).to(dev), (0.1 * torch.randn(O, generator=gen)).to(dev)
    gI, bI = (1 + 0.1 * torch.randn(I, generator=gen)).to(dev), (0.1 * torch.randn(I, generator=gen)).to(dev)
    rel = lambda a, b: float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))

    def fwd(ops, z, gam, bet):
        return (ops.bn_forward if groups == 1 else ops.bn_forward2)(z, gam, bet, 0.2, 1e-5, 0.1)
    used = 0
    for conv, src, gam, bet in (("conv_down", x, gO, bO), ("conv_up", y, gI, bI)):
        res = []
        for ops, cw in ((fu, cf), (se, cs)):
            out = getattr(ops, conv)(src, cw, want_stats=True, defer=groups)
            z = out[0] if isinstance(out, tuple) else out
            used += getattr(z, "_rg_slabs", None) is not None
            a, mean, inv = fwd(ops, z, gam, bet)
            res.append((z, a, mean, inv))
        torch.cuda.synchronize()
        assert torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16)), conv
        assert rel(res[0][1], res[1][1]) < 8e-3 and rel(res[0][2], res[1][2]) < 1e-5 and rel(res[0][3], res[1][3]) < 1e-5, conv
    for conv, src, C, gam, bet, zshape in (("conv_up", y, I, gI, bI, (n, hs, hs, I)), ("conv_down", x, O, gO, bO, (n, ho, ho, O))):
        zb = (torch.randn(*zshape, generator=gen) * 1.3 + 0.2).bfloat16().to(dev)
        res = []
        for ops, cw in ((fu, cf), (se, cs)):
            _, mean, inv = fwd(ops, zb.clone(), gam, bet)
            ga = getattr(ops, conv)(src, cw, defer=groups)
            ga = ga[0] if isinstance(ga, tuple) else ga
            used += getattr(ga, "_rg_slabs", None) is not None
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            if groups == 1:
                gz, _, _ = ops.bn_act_bwd(zb, ga, mean, inv, gam, bet, 0.2, dg, db, False, keep_ga=False)
            else:
                gz = ops.bn_act_bwd2(zb, ga, mean, inv, gam, bet, 0.2, dg, db, False)
            res.append((gz, dg, db))
        torch.cuda.synchronize()
        assert rel(res[0][0], res[1][0]) < 8e-3 and rel(res[0][1], res[1][1]) < 1e-4 and rel(res[0][2], res[1][2]) < 1e-4, conv
    assert used >= 1, "none of the four pairings took the slab path at this shape: the case checks nothing"
    assert int(fu._sb_sync[0]) == 0, "no hand-off timed out"


@pytest.mark.parametrize("N,O,I", [(64, 200, 774), (100, 64, 515), (40, 130, 1024), (64, 6000, 2048)])
def test_linear_weight_gradient_inside_adam(N, O, I):
    """rg_linear_wgrad_adam on a weight segment W[O][I] inside a larger buffer (so that its start is only 4- / 8- / 16-byte
    aligned as the case may be): dW = g^T x from the transposed, zero-padded bf16 operand images, torch.optim.Adam's update
    with weight decay; ragged tiles in both directions, odd row pitch, a ragged second 64-sample chunk; the floats around the
    segment stay untouched."""
    from rna_gan_amd import _abi
    dev = torch.device("cuda:0")
    lib = _abi.load()
    gen = torch.Generator().manual_seed(N + O + I)
    lead = {774: 2, 515: 1, 1024: 4, 2048: 0}[I]                 # elements in front of the segment: alignment 8 / 4 / 16 / 16
    buf = [torch.zeros(lead + O * I + 8, device=dev) for _ in range(3)]
    p, m, v = [b[lead:lead + O * I].view(O, I) for b in buf]
    p.copy_((torch.randn(O, I, generator=gen) * 0.05).to(dev)); m.copy_((torch.randn(O, I, generator=gen) * 1e-3).to(dev))
    v.copy_((torch.rand(O, I, generator=gen) * 1e-4).to(dev))
    for b in buf:
        b[:lead] = 7.0; b[lead + O * I:] = 7.0
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    g = (torch.randn(N, O, generator=gen) * 0.1).to(dev)
    x = torch.randn(N, I, generator=gen).to(dev)
    ldn = (N + 63) // 64 * 64
    gT = torch.zeros(O + 3, ldn, dtype=torch.bfloat16, device=dev); gT[:O, :N] = g.t().bfloat16()
    xT = torch.zeros(I + 5, ldn, dtype=torch.bfloat16, device=dev); xT[:I, :N] = x.t().bfloat16()
    step = torch.full((1,), 2, dtype=torch.int32, device=dev)
    hyper = torch.zeros(12, device=dev)
    lr, b1, b2, eps, wd = 3e-3, 0.9, 0.999, 1e-8, 1e-4
    stream = torch.cuda.current_stream().cuda_stream
    _abi.check(lib.rg_adam_hyper_dev(step.data_ptr(), lr, b1, b2, eps, wd, hyper.data_ptr(), stream), "rg_adam_hyper_dev")
    Kp = (I + 63) // 64 * 64
    wpack = torch.full((O + 2, Kp), 3.0, dtype=torch.bfloat16, device=dev)
    _abi.check(lib.rg_linear_wgrad_adam(gT.data_ptr(), xT.data_ptr(), ldn, N, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                        hyper.data_ptr(), O, I, wpack.data_ptr(), Kp, stream), "rg_linear_wgrad_adam")
    torch.cuda.synchronize()
    assert torch.equal(wpack[:O, :I], p.bfloat16()) and bool((wpack[:O, I:] == 3.0).all()) and bool((wpack[O:] == 3.0).all())
    dw = g.bfloat16().float().t() @ x.bfloat16().float()
    t = 3
    gg = dw + wd * p0
    m1 = m0 + (1 - b1) * (gg - m0)
    v1 = b2 * v0 + (1 - b2) * gg * gg
    p1 = p0 - (lr / (1 - b1 ** t)) * (m1 / (v1.sqrt() / (1 - b2 ** t) ** 0.5 + eps))
    scale = float(dw.abs().mean())
    assert float((m - m1).abs().max()) <= 2e-4 * (1 - b1) * scale * N ** 0.5 + 1e-6
    assert float(((v - v1) / (v1 + 1e-12)).abs().max()) <= 2e-3
    assert float(((p - p0) - (p1 - p0)).abs().max()) <= 2e-3 * float((p1 - p0).abs().max())
    for b in buf:
        assert bool((b[:lead] == 7.0).all()) and bool((b[lead + O * I:] == 7.0).all())
