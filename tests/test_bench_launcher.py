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


def test_workload_label_follows_the_arguments_and_host_is_described():
    """config.workload must name the configuration actually run; cpu_baseline carries the host CPU."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    wl = {"n": 99856}
    for (side, q, cs, miss), tag in {(1000, 1, 25, ""): "config #3", (316, 1, 25, ""): "config #2", (577, 3, 25, ""): "config #4",
                                     (1155, 3, 9, "0.1,0.3,0.5"): "config #5", (200, 2, 16, ""): "custom"}.items():
        a = argparse.Namespace(side=side, q=q, cell_size=cs, missing=miss)
        s = bench.workload_name(a, wl, 5461, 7, 1)
        assert s.startswith(tag) and f"{side}^2" in s and f"cell_size={cs}" in s and (miss in s)
    assert "sharded over 4 GPUs" in bench.workload_name(argparse.Namespace(side=1000, q=1, cell_size=25, missing=""), wl, 1, 1, 4)
    h = bench.host_cpu()
    assert set(h) >= {"model", "physical_cores", "logical_cpus"} and h["logical_cpus"] >= 1
    assert bench.mem_available_gb() > 0.0


def test_a_rank_that_never_arrives_costs_the_deadline_not_the_lease():
    """VERDICT r2 missing #1: rank 1 never reaches the collective (as a rank blocked in ncclCommInitRank would); the launcher
    must end exactly its own children after SPAMTREE_LAUNCH_DEADLINE seconds and exit non-zero without a JSON line."""
    import time
    t0 = time.time()
    r = run(["--gpus", "2", "--launch-check"], env={"SPAMTREE_LAUNCH_CHECK_HANG": "1", "SPAMTREE_LAUNCH_DEADLINE": "8"}, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert "no JSON line after 8 s" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
    assert time.time() - t0 < 60


def test_cpu_thread_plan_follows_the_contract():
    """min(physical cores, cpus allowed) and the README's 10 threads."""
    sys.path.insert(0, ROOT)
    import bench
    host, full, readme = bench.cpu_threads_plan()
    allowed = bench.effective_cpus(host)
    assert 1 <= full <= allowed <= (host.get("cpus_allowed") or os.cpu_count()) and readme == min(10, allowed)
    if host.get("physical_cores"):
        assert full == min(host["physical_cores"], allowed)
