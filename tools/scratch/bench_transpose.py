import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rna_gan_amd import _abi
lib = _abi.load()
st = torch.cuda.current_stream().cuda_stream
for O, I in ((2048, 1024), (1024, 512), (512, 256), (256, 128), (128, 64)):
    src = torch.randn(O, 16 * I, device="cuda").to(torch.bfloat16)
    dst = torch.zeros(16 * I, O, dtype=torch.bfloat16, device="cuda")
    fn = lambda: lib.rg_pack_conv_wup_from_bf16(src.data_ptr(), dst.data_ptr(), O, I, st)
    fn(); torch.cuda.synchronize()
    assert torch.equal(dst, src.t().contiguous()), (O, I)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(O, I, "%.1f us  %.2f TB/s" % (us, 4.0 * O * 16 * I / us / 1e6))
