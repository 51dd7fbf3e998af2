"""CPU: `python bench.py --gpus N` starts N ranks itself (VERDICT r1 missing #5) -- env plumbing rehearsed over gloo -- and
refuses to report an N-GPU line from fewer devices or from a launcher whose WORLD_SIZE disagrees."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_n_spawns_n_ranks_with_the_distributed_env():
    r = run(["--gpus", "3", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["rank_sum"] == 6.0 and out["master"] == "127.0.0.1"


def test_more_gpus_than_the_box_has_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has several GPUs")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_world_size_mismatch_is_refused():
    r = run(["--gpus", "4", "--launch-check"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0",
                                                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
