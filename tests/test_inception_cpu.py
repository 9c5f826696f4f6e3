"""f2: the FID metric's Inception-v3 (src/fid.py:33-94 uses torchvision inception_v3 up to Mixed_7c).  CPU side:
the product's table-driven module and the oracle's module-per-block restatement describe the SAME network -- identical
state_dict keys, shapes and order (= what a torchvision checkpoint must provide / what load_state_dict accepts) -- and the
oracle's forward has torchvision's published dimensions.  (No compute through the C ABI here: that is test_inception_gpu.)"""
import torch

from oracle.inception_ref import OracleInception3
from rna_gan_amd import inception as PI


def test_manifest_matches_oracle_and_torchvision_facts():
    man = PI.manifest()
    with torch.device("meta"):
        ref = [(k, tuple(v.shape)) for k, v in OracleInception3().state_dict().items()]
    assert man == ref
    keys = dict(man)
    # published facts of torchvision's Inception3: 27 161 264 parameters with the auxiliary head, first / last conv shapes
    n_params = sum(int(torch.tensor(s).prod()) if s else 1 for k, s in man
                   if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")))
    assert n_params == 27161264
    assert keys["Conv2d_1a_3x3.conv.weight"] == (32, 3, 3, 3) and keys["Mixed_7c.branch_pool.conv.weight"] == (192, 2048, 1, 1)
    assert keys["Mixed_6b.branch7x7_2.conv.weight"] == (128, 128, 1, 7) and keys["Mixed_6b.branch7x7_3.conv.weight"] == (192, 128, 7, 1)
    assert keys["AuxLogits.fc.weight"] == (1000, 768) and keys["fc.weight"] == (1000, 2048)
    assert keys["Mixed_5b.branch1x1.bn.running_var"] == (64,)


def test_product_module_loads_oracle_state_dict_strictly():
    torch.manual_seed(0)
    o = OracleInception3()
    p = PI.InceptionV3()
    r = p.load_state_dict(o.state_dict(), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    assert p.Mixed_7b.branch3x3dbl_3b.bn.eps == 0.001 and not p.training


def test_oracle_forward_dimensions():
    torch.manual_seed(0)
    o = OracleInception3().eval()
    with torch.no_grad():
        f = o.features(torch.rand(1, 3, 299, 299))
    assert f.shape == (1, 2048) and torch.isfinite(f).all()
    # torchvision's transform_input on x * 2 - 1 of a mid-grey image (x = 0.5 -> 0): what is left is (mean_c - .5) / .5
    x = OracleInception3.transform_input(torch.full((1, 3, 2, 2), 0.5) * 2 - 1)
    got = x[0, :, 0, 0]                   # 0 * (std / .5) + (mean - .5) / .5  ==  (mean - .5) / .5
    assert torch.allclose(got, (torch.tensor([0.485, 0.456, 0.406]) - 0.5) / 0.5, atol=1e-6)
