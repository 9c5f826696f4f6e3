"""tools/dp_first_run.sh -- what to run first on a multi-GPU node -- kept working against HEAD: its
self-test mode sends every knob line through the data-parallel route on a one-rank RCCL group on this one GPU.  Checked: the
script ends with status 0, every record holds one bench line with a finite value, the data-parallel lines really took the DP route
(config.ranks / collective backend), and the fp16 lines ran the fp16 build."""
import json
import math
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = ["knob_overlap0", "knob_g0factors0", "knob_splitbn_dp1", "knob_nchannels8", "knob_nchannels16", "knob_prefix_bwd1",
         "knob_route_whole", "knob_route_whole_splitbn", "fp16_wire_fp32", "fp16_wire_f16", "force_dp_1rank",
         "force_dp_1rank_prefix1", "force_dp_1rank_whole"]


def test_first_run_kit_selftest():
    env = dict(os.environ, DP_FIRST_RUN_SELFTEST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RNAGAN_FORCE_DP", "RNAGAN_DP_ROUTE"):
        env.pop(k, None)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "dp_first_run.sh"), "1", "3", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "exited" not in r.stdout, r.stdout[-3000:]
    out = os.path.join(ROOT, "gpurun_out", "dp_first_run")
    for tag in ["scale_1"] + KNOBS:
        p = os.path.join(out, tag + ".json")
        assert os.path.exists(p) and os.path.getsize(p), tag
        rec = json.loads(open(p).read().strip().splitlines()[-1])
        assert math.isfinite(rec["value"]) and rec["value"] > 0 and rec["n_gpus"] == 1, (tag, rec["value"])
        cfg = rec["config"]
        if tag != "scale_1":
            assert str(cfg.get("collective_backend", "")).startswith("rccl") and cfg.get("dp_route") in ("prefix", "whole"), (tag, cfg)
            assert cfg["dp_route"] == ("whole" if "whole" in tag else "prefix"), (tag, cfg["dp_route"])
        if tag.startswith("fp16_"):
            assert rec["dtype"] == "fp16" and cfg["dp_wire"] == ("f16" if tag.endswith("_f16") else "fp32"), (tag, rec["dtype"], cfg.get("dp_wire"))
