"""The product's synthetic generators (rna_gan_amd.synth: bench.py, CLI --synthetic) and the oracle's own copy
(oracle.ref_cpu) produce the same tensors bit for bit."""
import torch
import torch.nn as nn

from oracle import ref_cpu as R
from rna_gan_amd import synth as S


def test_generators_agree():
    for name, shape in (("model.0.0.weight", (8, 4, 4, 4)), ("model.1.1.running_var", (16,)), ("model.1.1.weight", (16,)),
                        ("model.1.1.bias", (16,)), ("m.num_batches_tracked", ())):
        assert torch.equal(S.seeded_tensor(name, shape, 7), R.seeded_tensor(name, shape, 7)), name
    assert torch.equal(S.synthetic_images(3, 16, 5), R.synthetic_images(3, 16, 5))
    assert torch.equal(S.synthetic_rna(5, 33, 9, distinct=2), R.synthetic_rna(5, 33, 9, distinct=2))
    assert torch.equal(S.synthetic_uniform(4, 9, 11), R.synthetic_uniform(4, 9, 11))
    u8 = S.synthetic_tiles_u8(3, 16, 5)
    assert u8.dtype == torch.uint8 and torch.equal((u8.float() / 255.0 - 0.5) / 0.5, S.synthetic_images(3, 16, 5))
    a, b = nn.Linear(5, 3), nn.Linear(5, 3)
    S.seeded_fill_(a, 3); R.seeded_fill_(b, 3)
    assert torch.equal(a.weight, b.weight) and torch.equal(a.bias, b.bias)
