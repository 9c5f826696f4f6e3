"""losses._LatentCache shares one betaVAE encode between the three loss plugins (src/wgan_loss.py:67-69, :159-161,
:289-291 build three frozen copies of one checkpoint).  Which encoders count as 'the same' is decided by the provenance
of their weights (betaVAE.signature()), not by a fingerprint of the values -- host logic, checked here without a GPU."""
import torch

from rna_gan_amd.betavae import betaVAE


def _vae():
    return betaVAE(40, 16, [32, 24, 16], [24, 32])


def test_signature_is_provenance_not_values(tmp_path):
    a, b, c = _vae(), _vae(), _vae()
    assert a.signature() != b.signature()                       # fresh modules: private
    sd = a.state_dict()
    b.load_state_dict(sd); c.load_state_dict(sd)
    assert a.signature() == b.signature() == c.signature()      # one state_dict loaded into all: shared
    for m in (a, b, c):
        m.to(torch.float64)                                     # moved / converted alike (buffers become new tensors): still shared
    assert a.signature() == b.signature() == c.signature()
    c.to(torch.float32)                                         # ... not alike: different values, different signature
    assert c.signature() != a.signature()
    c.to(torch.float64)
    assert c.signature() == a.signature()
    with torch.no_grad():
        c.z_mu.bias.add_(1.0)                                   # written since: private again
    assert c.signature() != a.signature()
    d = _vae()
    d.load_state_dict({k: v.clone() for k, v in sd.items()})    # equal VALUES from an anonymous dict: not shared
    assert d.signature() != a.signature()
    # a permutation of one weight's rows keeps every sum / norm (what the round-2 fingerprint hashed): still distinct
    e = _vae()
    sd_perm = {k: v.clone() for k, v in sd.items()}
    sd_perm["z_mu.weight"] = sd_perm["z_mu.weight"].flip(0)
    e.load_state_dict(sd_perm)
    assert e.signature() != a.signature()


def test_checkpoint_file_token(tmp_path):
    path = str(tmp_path / "vae.pt")
    torch.save(dict(_vae().state_dict()), path)
    e1, e2 = _vae(), _vae()
    e1.load_checkpoint_file(path); e2.load_checkpoint_file(path)
    assert e1.signature() == e2.signature() and e1.signature()[1][0] == "file"
    e1.eval()                                                   # eval() keeps the token, a training phase drops it
    assert e1.signature() == e2.signature()
    e2.weights_changed()                                        # the fused optimizer wrote the flat buffers
    assert e1.signature() != e2.signature()
    e1.train()
    assert e1.signature()[1][0] == "private"
