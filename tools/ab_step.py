#!/usr/bin/env python3
"""Interleaved A/B of the benchmarked training iteration (bench.py's `one_step`) inside ONE process on one box.

Boxes differ by +-2.5 % and one box drifts by ~1 % over minutes, so a change of 0.1 ms (1 %) cannot be read off two bench.py
runs.  Here every variant gets its OWN workload (models, optimizers, loss plug-ins, step graphs: built and primed under the
variant's settings), and the timed regions alternate  A B C A B C ...  for `--rounds` rounds; reported per variant: every
round's ms per iteration, the mean, and the paired difference to the first variant (mean and range over the rounds).

    python tools/ab_step.py --variants "base:skinny128=0;new:" --rounds 5 --steps 20

The first variant is built twice (`name#control`): two identically configured workloads differ by 0.006 ... 0.12 ms in one
process (each owns its buffers), and that difference -- `noise_floor_ms` in the result -- is what a delta has to exceed.

A variant is  name:key=value,key=value ...  with keys of three kinds
  * kernel-selection options of the C ABI (rg_set_option: conv8, convp, convd, skinny128, ...), re-applied at every switch
    (they only act when a launch is issued eagerly or captured; a captured graph replays what it captured);
  * RNAGAN_* environment variables that the host side reads when a workload is built (HipOps / Adam construction),
    exported before the variant is built and restored afterwards;
  * the module-level switches of rna_gan_amd.losses (LOOKAHEAD, D_BATCHED, G0_ADAM, ...), written as losses.NAME=0/1 and
    re-applied at every switch.
Nothing here is imported by the product or the tests.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402


def parse_variants(spec):
    out = []
    for part in spec.split(";"):
        part = part.strip()
        if not part:
            continue
        name, _, rest = part.partition(":")
        kv = {}
        for item in rest.split(","):
            item = item.strip()
            if item:
                k, _, v = item.partition("=")
                kv[k.strip()] = v.strip()
        out.append((name.strip(), kv))
    return out


class Variant:
    def __init__(self, name, kv, lib):
        self.name, self.lib = name, lib
        self.copts = {k: int(v) for k, v in kv.items() if not k.startswith("RNAGAN_") and not k.startswith("losses.")}
        self.env = {k: v for k, v in kv.items() if k.startswith("RNAGAN_")}
        self.mod = {k.split(".", 1)[1]: v for k, v in kv.items() if k.startswith("losses.")}
        self.step = self.flush = None
        self.times = []

    def apply(self, all_copts):
        from rna_gan_amd import _abi, losses
        for k in all_copts:                                   # every option any variant names: set or cleared
            _abi.check(self.lib.rg_set_option(k.encode(), self.copts.get(k, -1)), "rg_set_option(%s)" % k)
        for k, v in self.mod.items():
            cur = getattr(losses, k)
            setattr(losses, k, type(cur)(int(v)) if not isinstance(cur, str) else v)

    def build(self, bench, args, device, all_copts, base_mod):
        from rna_gan_amd import losses
        saved = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
        for k, v in base_mod.items():
            setattr(losses, k, v)
        self.apply(all_copts)
        try:
            self.step, self.flush, self.n, self.info = bench.hip_workload(args, 0, 1, device)
            for _ in range(args.prime):
                self.step()
            torch.cuda.synchronize(device)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", required=True)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prime", type=int, default=40)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--json", default=None, help="write the result object here as well")
    ap.add_argument("--no-control", action="store_true",
                    help="by default the FIRST variant is built a second time (name + '#control', same settings, own buffers) "
                         "and measured last: its difference to the first is the noise floor of the comparison -- two identically "
                         "configured workloads in one process differ by 0.006 ... 0.12 ms (buffer placement), DESIGN 14.1")
    a = ap.parse_args()
    import bench
    from rna_gan_amd import _abi, losses
    lib = _abi.load()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    bargs = bench.parse_args(["--batch", str(a.batch), "--no-cpu-baseline", "--no-roofline", "--no-extras"])
    bargs.prime = a.prime
    pv = parse_variants(a.variants)
    if not a.no_control and pv:
        pv.append((pv[0][0] + "#control", dict(pv[0][1])))
    vs = [Variant(n, kv, lib) for n, kv in pv]
    assert len(vs) >= 2, "need at least two variants"
    all_copts = sorted({k for v in vs for k in v.copts})
    base_mod = {k: getattr(losses, k) for v in vs for k in v.mod}
    for v in vs:
        t0 = time.perf_counter()
        v.build(bench, bargs, device, all_copts, base_mod)
        print("[ab_step] built + primed %-12s in %.1f s" % (v.name, time.perf_counter() - t0), file=sys.stderr, flush=True)
    for r in range(a.rounds):
        for v in (vs if r % 2 == 0 else vs[::-1]):           # alternate the order inside a round too
            for k, val in base_mod.items():
                setattr(losses, k, val)
            v.apply(all_copts)
            for _ in range(a.warmup):
                v.step()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                ls = v.step()
            v.flush()
            torch.cuda.synchronize(device)
            v.times.append((time.perf_counter() - t0) / a.steps * 1e3)
            v.last = [float(x.item()) for x in ls]
        print("[ab_step] round %d: %s" % (r, "  ".join("%s %.3f" % (v.name, v.times[-1]) for v in vs)), file=sys.stderr,
              flush=True)
    from rna_gan_amd.ops_hip import check_handoffs
    check_handoffs()
    res = {"steps": a.steps, "rounds": a.rounds, "batch": a.batch, "unit": "ms per iteration", "variants": []}
    base = vs[0]
    for v in vs:
        d = [x - y for x, y in zip(v.times, base.times)]
        res["variants"].append({"name": v.name, "options": dict(v.copts, **v.env, **{"losses." + k: x for k, x in v.mod.items()}),
                                "ms": [round(x, 3) for x in v.times], "mean": round(sum(v.times) / len(v.times), 3),
                                "min": round(min(v.times), 3),
                                "delta_vs_first": {"mean": round(sum(d) / len(d), 3), "min": round(min(d), 3),
                                                   "max": round(max(d), 3)},
                                "losses_last_step": [round(x, 5) for x in v.last]})
    if not a.no_control:
        res["noise_floor_ms"] = abs(res["variants"][-1]["delta_vs_first"]["mean"])
    line = json.dumps(res)
    print(line)
    if a.json:
        with open(a.json, "w") as f:
            f.write(line + "\n")
    for v in res["variants"]:
        print("[ab_step] %-12s mean %.3f  min %.3f  delta vs %s: %+.3f (%+.3f .. %+.3f)" %
              (v["name"], v["mean"], v["min"], base.name, v["delta_vs_first"]["mean"], v["delta_vs_first"]["min"],
               v["delta_vs_first"]["max"]), file=sys.stderr)


if __name__ == "__main__":
    main()
