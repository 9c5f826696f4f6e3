"""bench.py's own multi-rank launcher (`python bench.py --gpus 2` outside torchrun) end to end on CPU ranks: the parent
starts 2 fresh rank processes, they rendezvous over gloo on 127.0.0.1, run warm-up + timed steps of the toy plug-in
workload (tests/bench_plugin_cpu.py) bracketed by barriers, take the MAX over ranks, and rank 0 prints ONE JSON line
whose n_gpus comes from the live process group."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=300):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT)


def test_gpus2_launcher_runs_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--step-plugin", "tests.bench_plugin_cpu:make"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["collective_backend"] == "gloo"
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 8 and out["value"] > 0
    assert abs(out["value"] - 8 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-2 * out["value"]


def test_gpus_mismatch_is_an_error():
    """Launched under a 1-rank environment but asked for 2 GPUs: refuse instead of printing a 1-GPU number under a 2-GPU label."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--step-plugin", "tests.bench_plugin_cpu:make"],
             env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                        "MASTER_PORT": "29999"})
    assert r.returncode != 0
    assert "process group has 1 rank" in (r.stderr + r.stdout)


def test_failing_rank_fails_the_launch():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--step-plugin", "tests.bench_plugin_cpu:does_not_exist"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
