"""rna_gan_amd/prefetch.py on the CPU: identity wrapper for a CPU device, batch structure mapping."""
import collections

import torch

from rna_gan_amd.prefetch import DevicePrefetcher, _map


def test_cpu_device_is_the_identity_wrapper():
    batches = [{"image": torch.full((2, 3), float(i)), "rna_data": torch.arange(4.) + i, "labels": None} for i in range(3)]
    pf = DevicePrefetcher(batches, "cpu")
    got = list(pf)
    assert len(pf) == 3 and len(got) == 3
    for a, b in zip(got, batches):
        assert a is b


def test_structure_mapping():
    P = collections.namedtuple("P", "a b")
    obj = {"x": torch.ones(2), "y": [torch.zeros(1), (torch.ones(1), "s", 3)], "z": P(torch.ones(1), None)}
    out = _map(obj, lambda t: t + 1)
    assert float(out["x"][0]) == 2 and float(out["y"][0][0]) == 1 and out["y"][1][1:] == ("s", 3)
    assert isinstance(out["z"], P) and float(out["z"].a[0]) == 2 and out["z"].b is None and isinstance(out["y"][1], tuple)
