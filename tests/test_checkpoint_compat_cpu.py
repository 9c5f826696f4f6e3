"""A checkpoint written by the REFERENCE opens here (SURVEY 8b "Checkpoint"; src/histopathology_gan.py:311-312,
src/gan_utils.py:292-297): torchgan's ``save_model`` pickles the live plugin objects -- ``torchgan.losses.*`` /
``wgan_loss.*LossVAE`` instances holding a ``betaVAE.betaVAE`` module each -- next to the state_dicts, so a plain ``torch.load``
fails with ModuleNotFoundError on a machine without those modules.

The test writes such a file with stand-in modules registered under the reference's module names (the classes are
``nn.Module`` subclasses with parameters, nested modules, a plain function reference and an ``arg_map`` dict, as the real
ones), REMOVES the modules from ``sys.modules``, and loads the file through ``Trainer.load_model``: epoch, logs, model and
optimizer state must be restored, the unimportable classes must have become placeholders, and the trainer's own plugin
objects must still be the live ones."""
import copy
import sys
import types

import pytest
import torch
import torch.nn as nn

import rna_gan_amd as P
from rna_gan_amd import _tolerant_pickle
from rna_gan_amd import losses as L
from rna_gan_amd.optim import Adam
from rna_gan_amd.trainer import Trainer

FAKE = ["torchgan", "torchgan.losses", "torchgan.losses.loss", "wgan_loss", "betaVAE", "torchgan.metrics"]


def _network():
    return {
        "generator": {"name": P.DCGANGenerator,
                      "args": dict(encoding_dims=16, out_size=32, out_channels=3, step_channels=4,
                                   nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()),
                      "optimizer": {"name": torch.optim.Adam, "args": {"lr": 1e-4, "betas": (0.5, 0.999)}}},
        "discriminator": {"name": P.DCGANDiscriminator,
                          "args": dict(in_size=32, in_channels=3, step_channels=4,
                                       nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)),
                          "optimizer": {"name": torch.optim.Adam, "args": {"lr": 4e-4, "betas": (0.5, 0.999)}}}}


def _plugins():
    return [L.WassersteinGeneratorLoss(), L.WassersteinDiscriminatorLoss(clip=(-0.01, 0.01)), L.WassersteinGradientPenalty()]


def _install_fake_reference_modules():
    mods = {n: types.ModuleType(n) for n in FAKE}

    def reduce_vae(x, reduction=None):       # a module-level function the plugin objects refer to
        return x

    class GeneratorLoss(nn.Module):
        def __init__(self, reduction="mean", override_train_ops=None):
            super().__init__()
            self.reduction, self.override_train_ops, self.arg_map = reduction, override_train_ops, {}

    class DiscriminatorLoss(GeneratorLoss):
        pass

    class RNAEncoder(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = nn.Sequential(nn.Dropout(), nn.Sequential(nn.Linear(12, 6), nn.BatchNorm1d(6), nn.LeakyReLU()))

    class betaVAE(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder, self.z_mu, self.beta = RNAEncoder(), nn.Linear(6, 4), 0.005

    class WassersteinGeneratorLossVAE(GeneratorLoss):
        def __init__(self, checkpoint, rna_features, beta=0.005):
            super().__init__(checkpoint, rna_features)        # the reference's ctor quirk (src/wgan_loss.py:63-66)
            self.vae, self.reduce = betaVAE(), reduce_vae

    class WassersteinDiscriminatorLossVAE(DiscriminatorLoss):
        def __init__(self, checkpoint, rna_features, beta=0.005, clip=None):
            super().__init__(checkpoint, rna_features)
            self.vae, self.clip = betaVAE(), clip

    class WassersteinGradientPenaltyVAE(DiscriminatorLoss):
        def __init__(self, checkpoint, rna_features, beta=0.005, lambd=10.0):
            super().__init__(checkpoint, rna_features)
            self.vae, self.lambd = betaVAE(), lambd

    class ClassifierScore:                                    # a torchgan metric object (plain class)
        def __init__(self):
            self.name, self.history = "IS", [1.0, 2.0]

    placed = {"torchgan.losses.loss": [GeneratorLoss, DiscriminatorLoss], "betaVAE": [RNAEncoder, betaVAE],
              "wgan_loss": [WassersteinGeneratorLossVAE, WassersteinDiscriminatorLossVAE, WassersteinGradientPenaltyVAE,
                            reduce_vae],
              "torchgan.metrics": [ClassifierScore]}
    for mname, objs in placed.items():
        for o in objs:
            o.__module__ = mname
            o.__qualname__ = o.__name__
            setattr(mods[mname], o.__name__, o)
    sys.modules.update(mods)
    return mods


def _remove_fake_modules():
    for n in FAKE:
        sys.modules.pop(n, None)


@pytest.fixture
def reference_checkpoint(tmp_path):
    """What torchgan's Trainer.save_model writes: state_dicts of a trained-for-one-step G / D pair and of plain
    torch.optim.Adam instances, plus the pickled plugin / metric objects of the reference's modules."""
    torch.manual_seed(3)
    mods = _install_fake_reference_modules()
    try:
        src = Trainer(_network(), _plugins(), device=torch.device("cpu"), checkpoints=str(tmp_path / "src"), recon=None)
        sd, opt_sd = {}, {}
        for name in ("generator", "discriminator"):
            m = getattr(src, name)
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(0.01 * torch.randn_like(p))
                for b in m.buffers():
                    if b.dtype.is_floating_point:
                        b.add_(0.05 * torch.rand_like(b))
            sd[name] = copy.deepcopy(m.state_dict())
            # optimizer state as torch 1.10's Adam pickles it: python-int step, one entry per parameter index
            plist = list(m.parameters())
            opt_sd[name] = {"state": {i: {"step": 7, "exp_avg": torch.randn_like(p), "exp_avg_sq": torch.rand_like(p)}
                                      for i, p in enumerate(plist)},
                            "param_groups": [{"lr": 1e-4 if name == "generator" else 4e-4, "betas": (0.5, 0.999), "eps": 1e-8,
                                              "weight_decay": 0, "amsgrad": False, "params": list(range(len(plist)))}]}
        wl = mods["wgan_loss"]
        plugins = [wl.WassersteinGeneratorLossVAE("/data/vae.pt", 19198), wl.WassersteinDiscriminatorLossVAE("/data/vae.pt", 19198),
                   wl.WassersteinGradientPenaltyVAE("/data/vae.pt", 19198)]
        ckpt = {"epoch": 12, "loss_information": {"generator_losses": -3.5, "discriminator_losses": 9.25,
                                                  "generator_iters": 40, "discriminator_iters": 80},
                "loss_objects": {type(p).__name__: p for p in plugins},
                "metric_objects": {"ClassifierScore": mods["torchgan.metrics"].ClassifierScore()},
                "loss_logs": {type(p).__name__: [0.5, 0.25] for p in plugins}, "metric_logs": {"ClassifierScore": [1.0]},
                "generator": sd["generator"], "discriminator": sd["discriminator"],
                "optimizer_generator": opt_sd["generator"], "optimizer_discriminator": opt_sd["discriminator"]}
        path = str(tmp_path / "reference_gan0.model")
        torch.save(ckpt, path)
    finally:
        _remove_fake_modules()
    return path, sd, opt_sd


def test_plain_torch_load_fails_without_the_reference_modules(reference_checkpoint):
    path, _, _ = reference_checkpoint
    with pytest.raises((ModuleNotFoundError, AttributeError)):
        torch.load(path, map_location="cpu", weights_only=False)


def test_load_model_opens_a_reference_written_checkpoint(reference_checkpoint, tmp_path):
    path, sd, opt_sd = reference_checkpoint
    assert not any(n in sys.modules for n in FAKE)
    net = _network()
    for cfg in net.values():
        cfg["optimizer"]["name"] = Adam
    plugins = _plugins()
    tr = Trainer(net, plugins, device=torch.device("cpu"), checkpoints=str(tmp_path / "dst"), recon=None)
    tr.load_model(load_path=path)
    assert tr.start_epoch == 12
    assert tr.loss_information["discriminator_iters"] == 80 and tr.loss_information["generator_losses"] == -3.5
    assert tr.loss_logs["WassersteinGeneratorLossVAE"] == [0.5, 0.25]
    assert all(name in tr.loss_logs for name in tr.losses)               # the live plugins keep (get) their own lists
    assert list(tr.losses.values()) == plugins                           # ... and stay the live objects
    for name in ("generator", "discriminator"):
        got = getattr(tr, name).state_dict()
        assert list(got.keys()) == list(sd[name].keys())
        for k, v in sd[name].items():
            assert torch.equal(got[k], v), (name, k)
        osd = getattr(tr, "optimizer_" + name).state_dict()
        assert osd["param_groups"][0]["lr"] == opt_sd[name]["param_groups"][0]["lr"]
        for i, st in opt_sd[name]["state"].items():
            assert torch.equal(osd["state"][i]["exp_avg"], st["exp_avg"])
            assert torch.equal(osd["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
            assert int(float(osd["state"][i]["step"])) == 7
    missing = _tolerant_pickle.missing_globals()
    for want in [("wgan_loss", "WassersteinGeneratorLossVAE"), ("wgan_loss", "WassersteinGradientPenaltyVAE"),
                 ("betaVAE", "betaVAE"), ("betaVAE", "RNAEncoder"), ("wgan_loss", "reduce_vae"),
                 ("torchgan.metrics", "ClassifierScore")]:
        assert want in missing, want


def test_placeholders_keep_state_and_refuse_use(reference_checkpoint):
    path, _, _ = reference_checkpoint
    ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_tolerant_pickle)
    g = ck["loss_objects"]["WassersteinGeneratorLossVAE"]
    assert isinstance(g, _tolerant_pickle.MissingGlobal) and type(g)._rg_missing == ("wgan_loss", "WassersteinGeneratorLossVAE")
    assert g.reduction == "/data/vae.pt" and g.override_train_ops == 19198       # the ctor quirk survives as plain state
    w = g._modules["vae"]._modules["z_mu"]                                        # torch's own classes load as themselves
    assert isinstance(w, nn.Linear) and w.weight.shape == (4, 6)
    with pytest.raises(RuntimeError, match="placeholder"):
        g(1, 2)
    assert ck["metric_objects"]["ClassifierScore"].history == [1.0, 2.0]


def test_product_checkpoint_round_trip_still_works(tmp_path):
    net = _network()
    tr = Trainer(net, _plugins(), device=torch.device("cpu"), checkpoints=str(tmp_path / "gan"), recon=None)
    tr.loss_logs["WassersteinGeneratorLoss"].append(1.25)
    tr.save_model(4)
    tr2 = Trainer(_network(), _plugins(), device=torch.device("cpu"), checkpoints=str(tmp_path / "gan2"), recon=None)
    tr2.load_model(load_path=str(tmp_path / "gan0.model"))
    assert tr2.start_epoch == 5 and tr2.loss_logs["WassersteinGeneratorLoss"] == [1.25]
    for k, v in tr.generator.state_dict().items():
        assert torch.equal(tr2.generator.state_dict()[k], v)


def test_a_corrupt_stream_is_not_papered_over(tmp_path):
    bad = tmp_path / "bad.model"
    torch.save({"epoch": 1, "x": torch.nn.Linear(2, 2)}, str(bad))
    data = bad.read_bytes().replace(b"torch.nn.modules.linear", b"torch.nn.modules.linxxx")
    bad.write_bytes(data)
    tr = Trainer(_network(), _plugins(), device=torch.device("cpu"), checkpoints=str(tmp_path / "g"), recon=None)
    with pytest.raises(RuntimeError, match="could not be loaded"):
        tr.load_model(load_path=str(bad))


def test_own_package_globals_and_placeholder_load_items_are_refused(reference_checkpoint, tmp_path, capsys):
    """(1) a checkpoint that refers to a global of THIS package that no longer exists is an error, not a
    placeholder (a refactor must not be papered over); (2) load_items must not attach a placeholder to the trainer;
    (3) load_model says which globals it replaced."""
    import pickle

    class Gone:                                           # pickled as rna_gan_amd.losses.NoSuchPlugin
        pass
    Gone.__module__, Gone.__qualname__, Gone.__name__ = "rna_gan_amd.losses", "NoSuchPlugin", "NoSuchPlugin"
    L.NoSuchPlugin = Gone
    try:
        blob = pickle.dumps({"epoch": 1, "x": Gone()})
    finally:
        del L.NoSuchPlugin
    with pytest.raises(AttributeError):
        _tolerant_pickle.loads(blob)
    path, _, _ = reference_checkpoint
    net = _network()
    for cfg in net.values():
        cfg["optimizer"]["name"] = Adam
    tr = Trainer(net, _plugins(), device=torch.device("cpu"), checkpoints=str(tmp_path / "dst"), recon=None)
    capsys.readouterr()
    _tolerant_pickle._PLACEHOLDERS.clear()                # as a fresh process: the report lists what THIS load replaced
    tr.load_model(load_path=path)
    out = capsys.readouterr().out
    assert "inert placeholders" in out and "wgan_loss.WassersteinGeneratorLossVAE" in out
    with pytest.raises(RuntimeError, match="not importable"):      # a dict of the reference's metric objects: placeholders
        tr.load_model(load_path=path, load_items="metric_objects")
    assert not hasattr(tr, "metric_objects") or not any(
        _tolerant_pickle.is_placeholder(v) for v in getattr(tr, "metric_objects", {}).values())
    tr.load_model(load_path=path, load_items="metric_logs")        # plain data passes
    assert tr.metric_logs == {"ClassifierScore": [1.0]}
