import copy, sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn as nn
from oracle import ref_cpu as R
from oracle.ops_ref import RefOps
from rna_gan_amd import engine as E
from rna_gan_amd.ops_hip import HipOps
sys.path.insert(0, "tests")
from test_engine_gpu import mk, oracle64, err
for seed in (11, 12, 13, 14):
    in_size, step, enc, n = 32, 64, 128, 16
    G, D = mk(in_size, step, enc, seed)
    real = R.synthetic_images(n, in_size, seed=3*seed)
    noise = R.synthetic_normal(n, enc, seed=3*seed+1)
    ref, _ = oracle64(G, D, real, noise, 0.3)
    out = {}
    for name, make_ops, dev in (("twin", lambda: RefOps(torch.bfloat16), "cpu"), ("hip", lambda: HipOps(torch.bfloat16, "cuda:0"), "cuda")):
        Gx, Dx = copy.deepcopy(G).to(dev).train(), copy.deepcopy(D).to(dev).train()
        E.tap_major_(Gx), E.tap_major_(Dx)
        Gn, Dn = E.build_gen_net(Gx), E.build_disc_net(Dx)
        ops = make_ops()
        # instrument: capture gzl passed to nchw_chan_sum
        cap = {}
        orig = ops.nchw_chan_sum
        def wrapped(gzl, dst, acc, orig=orig, cap=cap):
            cap["gzl"] = gzl.detach().clone()
            return orig(gzl, dst, acc)
        ops.nchw_chan_sum = wrapped
        E.gen_loss_grads(ops, Gn, Dn, noise.to(dev))
        b = dict(Gx.named_parameters())["model.3.0.bias"].grad.detach().cpu().double()
        exact = cap["gzl"].double().sum(dim=(0, 2, 3)).cpu()
        out[name] = b
        print(seed, name, "bias grad", b.tolist(), "exact sum of its own gzl", exact.tolist(), "abs-sum", cap["gzl"].double().abs().sum(dim=(0,2,3)).cpu().tolist())
    print(seed, "fp64", ref["G"]["model.3.0.bias"].tolist())
