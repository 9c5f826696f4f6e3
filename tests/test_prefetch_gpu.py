"""DevicePrefetcher on the GPU: same batches in the same order, every tensor on the device, inside Trainer.train the loss
plugins see device tensors (their own .to(device) calls are no-ops)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from rna_gan_amd.prefetch import DevicePrefetcher


def test_batches_arrive_on_the_device_in_order():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    host = [{"image": torch.randn(8, 3, 32, 32, generator=g), "rna_data": torch.randn(8, 50, generator=g), "labels": None,
             "pair": (torch.arange(8), torch.randn(8, generator=g).pin_memory())} for _ in range(5)]
    pf = DevicePrefetcher(host, dev)
    seen = 0
    for k, b in enumerate(pf):
        # work on the consumer's stream between batches, as a train_iter would do
        x = b["image"] * 2.0
        torch.cuda.current_stream().synchronize()
        assert b["image"].device == dev and b["rna_data"].device == dev and b["pair"][0].device == dev and b["labels"] is None
        assert torch.equal(b["image"].cpu(), host[k]["image"]) and torch.equal(b["rna_data"].cpu(), host[k]["rna_data"])
        assert torch.equal(b["pair"][1].cpu(), host[k]["pair"][1]) and torch.equal(x.cpu(), host[k]["image"] * 2.0)
        assert b["image"].to(dev) is b["image"]           # what the plugins' own .to(device) does now
        seen += 1
    assert seen == 5 and len(pf) == 5
    assert list(DevicePrefetcher([], dev)) == []


_SEEN = []


def _probe_class():
    from rna_gan_amd import losses as L

    class Probe(L.WassersteinDiscriminatorLoss):          # module-level name: the Trainer pickles its loss objects
        def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
            _SEEN.append((real_inputs.device, labels.device if labels is not None else None))
            return super().train_ops(generator, discriminator, optimizer_discriminator, real_inputs, device, labels)
    Probe.__qualname__ = "Probe"
    globals()["Probe"] = Probe
    return Probe


def test_trainer_train_feeds_device_batches(tmp_path):
    import rna_gan_amd as P
    dev = torch.device("cuda:0")
    seen = _SEEN
    Probe = _probe_class()

    models = {"generator": {"name": P.DCGANGenerator, "args": {"encoding_dims": 16, "out_size": 32, "out_channels": 3,
                                                               "step_channels": 8},
                            "optimizer": {"name": torch.optim.Adam, "args": {"lr": 1e-4, "betas": (0.5, 0.999)}}},
              "discriminator": {"name": P.DCGANDiscriminator, "args": {"in_size": 32, "in_channels": 3, "step_channels": 8},
                                "optimizer": {"name": torch.optim.Adam, "args": {"lr": 4e-4, "betas": (0.5, 0.999)}}}}
    for prefetch, want in ((True, "cuda"), (False, "cpu")):
        seen.clear()
        tr = P.Trainer(models, [P.WassersteinGeneratorLoss(), Probe()], device=dev, epochs=1, sample_size=4,
                       checkpoints=str(tmp_path / "m"), recon=None, precision="fp32", prefetch=prefetch)
        data = [(torch.randn(4, 3, 32, 32), torch.zeros(4, dtype=torch.long)) for _ in range(3)]

        class Loader(list):
            batch_size = 4
        tr(Loader(data))
        # tuple batches are moved by Trainer.train itself either way; with the prefetcher that .to() is a no-op
        assert len(seen) == 3 and all(d[0].type == "cuda" for d in seen)
        assert tr.prefetch is prefetch
