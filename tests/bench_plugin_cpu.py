"""Step plug-in for bench.py's TEST HOOK (--step-plugin tests.bench_plugin_cpu:make): the three train_op gradient
bodies of the product engine on the CPU twin ops (oracle.ops_ref.RefOps, fp64) at toy sizes, with the product's
data-parallel glue (grad_scale + flat-buffer SUM all-reduce, rna_gan_amd.dist) and a plain Adam step.  It lets the
launcher, the rank wiring (RANK / WORLD_SIZE / MASTER_*), the barrier + MAX-over-ranks timing protocol and the JSON
line of `bench.py --gpus 2` run end to end on CPU ranks with gloo.  Not a measurement of anything."""
import torch
import torch.nn as nn

from oracle import ref_cpu as R
from oracle.ops_ref import RefOps


def make(args, rank, world, device):
    from rna_gan_amd import dist as D_, engine as E
    from rna_gan_amd.models import FlatParams
    torch.set_num_threads(1)
    n = 4
    G = R.seeded_fill_(R.OracleDCGANGenerator(16, 16, 3, 4, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 3)
    D = R.seeded_fill_(R.OracleDCGANDiscriminator(16, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                                                  last_nonlinearity=nn.LeakyReLU(0.2)), 4)
    G, D = G.double().train(), D.double().train()
    flat_g, flat_d = FlatParams(G), FlatParams(D)
    Gn, Dn = E.build_gen_net(G), E.build_disc_net(D)
    ops = D_.attach_sync(RefOps(torch.float64))
    og = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999))
    real = R.synthetic_images(n, 16, seed=50 + rank).double()
    it = [0]

    def one_step():
        k = it[0]
        it[0] += 1
        nz = [R.synthetic_normal(n, 16, seed=1000 * rank + 3 * k + j).double() for j in range(3)]
        lg = E.gen_loss_grads(ops, Gn, Dn, nz[0], grad_scale=D_.grad_scale())
        D_.allreduce_sum_(flat_g.grad); og.step()
        ld = E.disc_loss_grads(ops, Gn, Dn, real, nz[1], grad_scale=D_.grad_scale())
        D_.allreduce_sum_(flat_d.grad); od.step()
        lp = E.gp_loss_grads(ops, Gn, Dn, real, nz[2], 0.3, 10.0, grad_scale=D_.gp_grad_scale())
        D_.allreduce_sum_(flat_d.grad); od.step()
        return [torch.as_tensor(float(x)) for x in (lg, ld, lp)]

    def flush():
        pass

    # replicas must stay identical across ranks (same seed, all-reduced gradients): checked by the test through the
    # checksum rank 0 prints in the JSON line
    info = {"workload": "toy CPU twin (test hook)", "rna_features": None,
            "param_checksum": lambda: float(sum(p.double().sum() for p in list(G.parameters()) + list(D.parameters())))}
    return one_step, flush, n, info
