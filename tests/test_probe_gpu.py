"""The on-box ceiling probes behind bench.py's roofline.peak_measured (rna_gan_amd/probe.py, csrc/rg_probe.hip): they run,
their FLOP / byte accounting is the stated one, and the rates they read are physically possible on an MI355X."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ceiling_probes_run_and_read_plausible_rates():
    from rna_gan_amd import probe
    dev = torch.device("cuda", 0)
    pm = probe.measure_ceilings(dev, settle_s=0.05, n_timed=2, copy_mb=256)
    bare = [v for k, v in pm.items() if k.startswith("mfma_bare_")]
    loop = [v for k, v in pm.items() if k.startswith("conv8_loop_lds_fed_")]
    assert len(bare) == 4 and len(loop) == 2
    # nominal dense bf16 peak 2500 TFLOP/s; a bare loop on random data holds well under it (DVFS) and far above a tenth of it
    assert all(300.0 < v < 2600.0 for v in bare), pm
    # the LDS-fed product loop cannot beat the bare loops by more than measurement noise
    assert all(200.0 < v < 1.05 * max(bare) for v in loop), pm
    assert 1000.0 < pm["stream_copy_gbps"] < 8200.0, pm


def test_lds_fed_probe_accounts_its_flops_and_writes_finite_results():
    from rna_gan_amd import _abi
    lib = _abi.load()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    blocks, iters = 8, 64
    a = torch.empty((blocks * 256, 128), dtype=torch.bfloat16, device=dev)
    b = torch.empty((256, 128), dtype=torch.bfloat16, device=dev)
    c = torch.zeros((blocks * 256, 256), dtype=torch.bfloat16, device=dev)
    _abi.check(lib.rg_probe_fill_bf16(a.data_ptr(), a.numel(), 1, st), "fill")
    _abi.check(lib.rg_probe_fill_bf16(b.data_ptr(), b.numel(), 2, st), "fill")
    assert float(a.float().abs().max()) <= 1.0 and float(a.float().std()) > 0.4
    fl = ctypes.c_double(0.0)
    for shape in (16, 32):
        _abi.check(lib.rg_probe_lds_mfma(shape, blocks, iters, a.data_ptr(), b.data_ptr(), c.data_ptr(), ctypes.addressof(fl), st), "probe")
        torch.cuda.synchronize()
        assert fl.value == blocks * iters * 2.0 * 256 * 256 * 64
        # the loop re-reads the two resident 64-deep k-tiles (the operands' 128 columns) iters / 2 times each: c = (iters / 2) a b^T
        ref = (iters / 2) * (a.float() @ b.float().t())
        got = c.float()
        assert torch.isfinite(got).all()
        assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max()) + 1e-3
