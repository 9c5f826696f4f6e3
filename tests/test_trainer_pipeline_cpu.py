"""Trainer.train_iter's launch / read ordering on the CPU (no kernels involved): plugins whose ``train_ops`` carries the
library's async marker are launched through ``train_ops_async`` and read one launch later; a plain plugin is called as written;
``carry`` leaves the last value pending for ``flush_pending`` / the next call; values and their order in ``loss_logs`` are those of
the synchronous loop."""
import torch

from rna_gan_amd import losses as L
from rna_gan_amd.trainer import Trainer

LOG = []


class _Val:
    """stands for a device scalar: .item() is the 'read'"""
    def __init__(self, name, v):
        self.name, self.v = name, v

    def item(self):
        LOG.append("read " + self.name)
        return self.v


def _mk(base, name, value, asynchronous=True):
    class Plug(base):
        def __init__(self):
            pass

        def train_ops_async(self, device):
            LOG.append("launch " + name)
            return _Val(name, value)

        def train_ops(self, device):
            if not asynchronous:
                LOG.append("sync " + name)
                return value
            return self.train_ops_async(device).item()
    if asynchronous:
        Plug.train_ops._rg_async = "train_ops_async"
    Plug.__name__ = name
    return Plug()


def _trainer(plugins, pipeline=True):
    tr = Trainer.__new__(Trainer)
    tr.losses = {type(p).__name__: p for p in plugins}
    tr.loss_logs = {n: [] for n in tr.losses}
    tr.loss_information = {"generator_losses": 0.0, "discriminator_losses": 0.0, "generator_iters": 0, "discriminator_iters": 0}
    tr.ncritic, tr.pipeline, tr.device = 1, pipeline, torch.device("cpu")
    tr._arg_maps = {n: {"device": "device"} for n in tr.losses}
    # _take() reads tensors and (slot, event) pairs; the stand-in values go through .item()
    tr._take = staticmethod(lambda v: v.item() if hasattr(v, "item") else v).__func__
    return tr


def test_launch_order_and_values():
    G, D, Pn = _mk(L.GeneratorLoss, "G", 1.5), _mk(L.DiscriminatorLoss, "D", 2.5), _mk(L.DiscriminatorLoss, "P", 4.0)
    LOG.clear()
    tr = _trainer([G, D, Pn])
    assert tr.train_iter() == (1.5, 6.5, 1, 2)
    assert LOG == ["launch G", "launch D", "read G", "launch P", "read D", "read P"]
    assert tr.loss_logs == {"G": [1.5], "D": [2.5], "P": [4.0]}
    # the synchronous loop: same values, read right behind each launch
    LOG.clear()
    ts = _trainer([G, D, Pn], pipeline=False)
    assert ts.train_iter() == (1.5, 6.5, 1, 2)
    assert LOG == ["launch G", "read G", "launch D", "read D", "launch P", "read P"] and ts.loss_logs == tr.loss_logs


def test_carry_and_flush():
    G, D, Pn = _mk(L.GeneratorLoss, "G", 1.0), _mk(L.DiscriminatorLoss, "D", 2.0), _mk(L.DiscriminatorLoss, "P", 3.0)
    LOG.clear()
    tr = _trainer([G, D, Pn])
    assert tr.train_iter(carry=True) == (1.0, 2.0, 1, 2)            # P's value is still pending
    assert LOG[-1] == "read D" and tr.loss_logs["P"] == []
    assert tr.train_iter(carry=True) == (1.0, 5.0, 1, 2)            # ... and is read behind the next call's first launch
    assert LOG[5:8] == ["launch G", "read P", "launch D"] and tr.loss_logs["P"] == [3.0]
    assert tr.flush_pending() == (0.0, 3.0) and tr.loss_logs == {"G": [1.0, 1.0], "D": [2.0, 2.0], "P": [3.0, 3.0]}
    assert tr.flush_pending() == (0.0, 0.0)


def test_foreign_plugin_is_called_as_written():
    G = _mk(L.GeneratorLoss, "G", 1.0)
    U = _mk(L.DiscriminatorLoss, "U", 7.0, asynchronous=False)
    LOG.clear()
    tr = _trainer([G, U])
    assert tr.train_iter() == (1.0, 7.0, 1, 1)
    assert LOG == ["launch G", "sync U", "read G"] and tr.loss_logs == {"G": [1.0], "U": [7.0]}
