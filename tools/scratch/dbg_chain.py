import copy, sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn as nn
from oracle import ref_cpu as R
from oracle.ops_ref import RefOps
from rna_gan_amd import engine as E
from rna_gan_amd.ops_hip import HipOps
sys.path.insert(0, "tests")
from test_engine_gpu import mk
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12
in_size, step, enc, n = 32, 64, 128, 16
G, D = mk(in_size, step, enc, seed)
real = R.synthetic_images(n, in_size, seed=3*seed)
noise = R.synthetic_normal(n, enc, seed=3*seed+1)
logs = {}
for name, make_ops, dev in (("f64", lambda: RefOps(torch.float64), "cpu"), ("twin", lambda: RefOps(torch.bfloat16), "cpu"), ("hip", lambda: HipOps(torch.bfloat16, "cuda:0"), "cuda")):
    Gx, Dx = copy.deepcopy(G).to(dev).train(), copy.deepcopy(D).to(dev).train()
    if name == "f64":
        Gx, Dx = Gx.double(), Dx.double()
    E.tap_major_(Gx), E.tap_major_(Dx)
    Gn, Dn = E.build_gen_net(Gx), E.build_disc_net(Dx)
    ops = make_ops()
    log = []
    def wrap(fname, fn):
        def w(*a, **k):
            r = fn(*a, **k)
            outs = r if isinstance(r, (tuple, list)) else (r,)
            for i, o in enumerate(outs[:1]):
                if torch.is_tensor(o) and o.numel() > 1:
                    log.append((fname + "#%d" % i, o.detach().double().cpu()))
            return r
        return w
    for fname in ("g0_fwd", "bn_forward", "conv_up", "conv_down", "last_up", "first_down", "head_fwd", "head_grad", "head_bwd_data", "bn_act_bwd", "tanh_bwd", "lrelu_bwd"):
        setattr(ops, fname, wrap(fname, getattr(ops, fname)))
    z = noise.to(dev)
    if name == "f64":
        z = z.double()
    E.gen_loss_grads(ops, Gn, Dn, z)
    logs[name] = log
ref = logs["f64"]
print("%-18s %12s | %10s %10s | %10s" % ("op", "norm(f64)", "twin-f64", "hip-f64", "hip-twin"))
for i, (nm, t) in enumerate(ref):
    a, b = logs["twin"][i][1], logs["hip"][i][1]
    assert logs["twin"][i][0] == nm and logs["hip"][i][0] == nm
    if a.shape != t.shape:
        continue
    n0 = float(t.norm()) + 1e-30
    print("%-18s %12.4e | %10.3e %10.3e | %10.3e   norms twin %.4e hip %.4e" % (nm, n0, float((a - t).norm()) / n0, float((b - t).norm()) / n0, float((b - a).norm()) / n0, float(a.norm()), float(b.norm())))
