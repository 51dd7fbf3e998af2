#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/s of the HIP hot path on the n=1e6 univariate grid (BASELINE.json metric).

One "step" = one body of the reference's MCMC loop (/root/reference/src/spamtree_fit.cpp:167-391) with all four
samplers on, no prediction, no saving: w sweep (B) + w log-density (C) + Metropolis proposal for theta, i.e. a
full re-factorisation of the proposal slot (A) + tausq and beta conjugate draws (S1, S2).
Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_GLOB = "profiles/r*/*_pmc_hbm.json"   # written by profiles/collect.sh + profiles/summarize.py; the newest one is read
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_PEAK_TFLOPS = 78.6    # FP64 matrix = FP64 vector peak; both run on ONE pipe per SIMD (profiles/r01/micro_f64_pipes.log)


def reference_cost(wl):
    """Flops of the reference algorithm AS WRITTEN per iteration (BASELINE.md section 4 cost law): dense
    (P+m)^3 Gram of the extended inverse Cholesky for blocks with children, H = Kxc' Kxx_inv (2mP^2),
    G = A_u H (2mP^2), Schur complement and m x m factorisations."""
    ip, pp, pidx = wl["indexing"][0], wl["parents"][0], wl["parents"][1]
    cp = wl["children"][0]
    m = np.diff(ip).astype(np.float64)
    P = np.array([m[pidx[pp[u]:pp[u + 1]]].sum() for u in range(m.size)])
    has_ch = np.diff(cp) > 0
    return float(np.sum(4 * m * P * P + 2 * m * m * P + m ** 3 + np.where(has_ch, (P + m) ** 3, 0.0)))


def mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                return float(ln.split()[1]) / 1024.0 ** 2
    except Exception:      # noqa: BLE001
        pass
    return 0.0


def effective_cpus(h):
    """CPUs this process can really use: affinity mask, bounded by the cgroup quota (or, on a gpurun box that reports none,
    by its documented 16-CPU share per GPU)."""
    allowed = h.get("cpus_allowed") or os.cpu_count() or 1
    if h.get("cpu_quota"):
        allowed = min(allowed, h["cpu_quota"])
    elif "GRAFT_REPO_ROOT" in os.environ:
        allowed = min(allowed, 16)
    return max(1, allowed)


def cpu_threads_plan():
    """the reference path timed with OMP_NUM_THREADS = physical cores of the box (bounded by what this
    process may use: affinity mask AND cgroup CPU quota; SPAMTREE_CPU_THREADS overrides) and with the README's
    num_threads = 10 (/root/reference/README.md:69)."""
    h = host_cpu()
    allowed = effective_cpus(h)
    full = min(h.get("physical_cores") or allowed, allowed)
    if os.environ.get("SPAMTREE_CPU_THREADS"):
        full = int(os.environ["SPAMTREE_CPU_THREADS"])
    return h, max(1, full), min(10, allowed)


def cpu_baseline(full_wl, side, seconds_budget=20.0):
    """oracle/refcpu (OpenMP restatement of the reference algorithm as written, kind="port") timed on the host cores on a
    bounded sample: full iterations (B + C + A + statistics).  When the host has the memory for the reference's caches at
    the FULL workload (about 160 GB at n = 1e6: two copies of every dense per-block matrix, profiles/r02/cpu_scaling.json)
    the sample is a few iterations of the full workload itself -- nothing is extrapolated; otherwise a smaller grid of the
    same family, scaled with the reference's own cost law (exponent 0.99 against measurements at five sizes, same file).
    Allocation / page-touch of the caches is not timed.  Two thread counts (cpu_threads_plan): `value` is the faster one."""
    from oracle.refcpu import RefCpu
    from spamtree_amd.synthetic import make_workload
    host, cores, readme_threads = cpu_threads_plan()
    need_gb = 165.0 * full_wl["n"] / 1.0e6 * (full_wl["q"] ** 2 if full_wl["q"] > 1 else 1)
    direct = os.environ.get("SPAMTREE_CPU_DIRECT", "1") != "0" and mem_available_gb() > need_gb + 30.0 and need_gb < 200.0
    wl = full_wl if direct else make_workload(side)
    if direct:
        seconds_budget = 8.0
    rc = RefCpu(wl["y"], wl["X"], wl["coords"], wl["mv_id"], wl["res_is_ref"], wl["parents"], wl["children"],
                wl["block_names"], wl["block_groups"], wl["indexing"], threads=cores)
    rc.set_tausq_inv(10.0)
    rc.set_beta(np.zeros((wl["p"], wl["q"])))
    rng = np.random.default_rng(1)
    rc.factor(0, wl["theta"])
    if direct:
        rc.factor(1, wl["theta"])       # touch the second cache copy before the clock starts

    def timed(threads, budget, min_its):
        rc.set_threads(threads)
        its, t0 = 0, time.perf_counter()
        while True:
            rc.sample_w(rng.standard_normal(wl["n"]))
            rc.loglik_w(0)
            rc.factor(1, wl["theta"] * (1 + 0.01 * rng.standard_normal(wl["theta"].size)))
            rc.stats()
            its += 1
            dt = time.perf_counter() - t0
            if (dt > budget and its >= min_its) or its >= 400:
                return its, dt

    runs = []
    for threads in sorted({cores, readme_threads}, reverse=True):
        its, dt = timed(threads, seconds_budget, 3)
        runs.append({"threads": threads, "iterations": its, "seconds": round(dt, 2), "it_per_s_at_sample": its / dt})
    rc.close()
    ratio = 1.0 if direct else reference_cost(wl) / reference_cost(full_wl)
    for r in runs:
        r["value"] = r["it_per_s_at_sample"] * ratio
    best = max(runs, key=lambda r: r["value"])
    scaling = None
    try:      # measured at several sizes on a GPU box's host by profiles/cpu_scaling.py (committed): fitted exponent vs the law
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cpu_scaling.json")))
        if f:
            scaling = json.load(open(f[-1]))
            scaling = {"source": os.path.relpath(f[-1], ROOT), "threads": scaling.get("threads"),
                       "fitted_exponent_seconds_vs_cost_law": scaling.get("fitted_exponent_seconds_vs_cost_law"),
                       "fitted_exponent_seconds_vs_n": scaling.get("fitted_exponent_seconds_vs_n")}
    except Exception:      # noqa: BLE001
        scaling = None
    what = (f"full iterations (B+C+A+stats) of oracle/refcpu on the FULL workload (n={wl['n']}): measured, not extrapolated"
            if direct else
            f"full iterations (B+C+A+stats) of oracle/refcpu on the {side}^2 grid (n={wl['n']}), scaled by the reference cost "
            f"law (sum (P+m)^3 + 4mP^2 + ...) ratio {ratio:.4f} to n={full_wl['n']}")
    return {"value": best["value"], "unit": "Gibbs iterations/s", "cores": best["threads"], "kind": "port", "host": host,
            "by_threads": runs,
            "threads_note": "SURVEY.md 8(d): min(physical cores, cpus allowed) and the reference README's num_threads = 10; "
                            "`value` / `cores` = the faster of the two",
            "scaling_check": scaling,
            "sample": "; ".join(f"{r['iterations']} {what} in {r['seconds']} s with {r['threads']} threads" for r in runs),
            "extrapolated": not direct,
            "measured_it_per_s_at_sample": best["it_per_s_at_sample"], "sample_n": int(wl["n"])}


def host_cpu():
    """lscpu model name, physical cores and sockets of the host the CPU baseline runs on."""
    info = {"model": None, "physical_cores": None, "logical_cpus": os.cpu_count()}
    try:
        import subprocess
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {ln.split(":", 1)[0].strip(): ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ":" in ln}
        info["model"] = kv.get("Model name")
        info["physical_cores"] = int(kv.get("Core(s) per socket", "0")) * int(kv.get("Socket(s)", "1")) or None
    except Exception:      # noqa: BLE001
        pass
    try:
        info["cpus_allowed"] = len(os.sched_getaffinity(0))
    except Exception:      # noqa: BLE001
        pass
    info["cpu_quota"] = cgroup_cpu_quota()
    return info


def cgroup_cpu_quota():
    """CPUs' worth of run time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None: an affinity mask of 256 CPUs
    says nothing when the scheduler grants 16 of them -- 128 OpenMP threads on such a share run at HALF the rate of 16."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(float(q) / float(per) + 0.5))
        return None
    except Exception:      # noqa: BLE001
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return max(1, int(q / per + 0.5)) if q > 0 else None
    except Exception:      # noqa: BLE001
        return None


def workload_name(args, wl, n_blocks, n_levels, world):
    """config.workload from the ACTUAL arguments."""
    known = {(1000, 1, 25, ""): "config #3", (316, 1, 25, ""): "config #2", (577, 3, 25, ""): "config #4",
             (1155, 3, 9, "0.1,0.3,0.5"): "config #5"}
    tag = known.get((args.side, args.q, args.cell_size, args.missing), "custom")
    if getattr(args, "limited_tree", False):
        tag = "custom (limited_tree = TRUE)"
    cov = "univariate exponential covariance" if args.q == 1 else f"q={args.q} Apanasovich-Genton cross-covariance"
    miss = f", outcomes dropped with probabilities ({args.missing})" if args.missing else ""
    return (f"{tag}: n={wl['n']} rows ({args.side}^2 grid x q={args.q}) {cov}{miss}, tree cell_size={args.cell_size} K=(2,2), "
            f"{n_blocks} blocks on {n_levels} levels, B+C+A+S1+S2 per iteration, theta at the data-generating value, RAM-adaptive MH"
            + (f"; ONE problem sharded over {world} GPUs by subtree, RCCL exchanges" if world > 1 else ""))


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, BEFORE anything touches the GPU (children are
    fresh processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what torch.distributed.run would give them);
    rank 0 prints the JSON line on the inherited stdout.  Fails loudly when the box has fewer GPUs than asked for."""
    import socket
    import subprocess
    if not args.launch_check:
        import torch          # device_count() does not initialise the GPU (no HIP context is created)
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible on this box; refusing to report "
                             f"a {args.gpus}-GPU line from fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    import tempfile
    done_flag = os.path.join(tempfile.gettempdir(), f"spamtree_bench_done_{os.getpid()}_{port}")   # rank 0 touches it after its JSON line
    for r in range(args.gpus):
        env = dict(os.environ, SPAMTREE_BENCH_DONE_FLAG=done_flag, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    # deadline: a rank blocked inside ncclCommInitRank (or any collective) would otherwise hold the box until the driver's own
    # limit.  Rank 0's stdout is relayed line by line; `deadline` seconds without the JSON line end exactly our child PIDs.
    deadline = float(os.environ.get("SPAMTREE_LAUNCH_DEADLINE", "240"))
    t_start = time.time()
    rc = 0
    got_line = False
    try:
        alive = list(procs)
        while alive:
            time.sleep(0.2)
            if os.path.exists(done_flag):
                got_line = True
            if not got_line and time.time() - t_start > deadline:
                print(f"bench.py --gpus {args.gpus}: no JSON line after {deadline:.0f} s (a rank is probably blocked in "
                      "ncclCommInitRank or a collective); terminating the ranks started here", file=sys.stderr, flush=True)
                rc = 124
                for pr in alive:
                    pr.terminate()
                t_kill = time.time() + 10.0
                while time.time() < t_kill and any(pr.poll() is None for pr in alive):
                    time.sleep(0.2)
                break
            for pr in list(alive):
                code = pr.poll()
                if code is None:
                    continue
                alive.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in alive:      # a dead rank leaves the others blocked in a collective: end exactly those PIDs
                        other.terminate()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        try:
            os.unlink(done_flag)
        except OSError:
            pass
    raise SystemExit(rc)


def mark_done():
    """Tell the launcher (launch_ranks) that rank 0 is about to print its JSON line: the deadline no longer applies."""
    flag = os.environ.get("SPAMTREE_BENCH_DONE_FLAG")
    if flag:
        try:
            open(flag, "w").close()
        except OSError:
            pass


class Watchdog:
    """Per-rank deadline for launches through torch.distributed.run (which bypass launch_ranks): a rank that has not
    finished its timed region `seconds` after start-up -- blocked in ncclCommInitRank, a collective or a barrier because
    another rank died or never arrived -- says so and exits 124, so the launcher tears the job down in minutes instead of
    holding the node until the driver's own limit."""

    def __init__(self, seconds, rank):
        import threading
        self.t = threading.Timer(seconds, self._fire, args=(seconds, rank))
        self.t.daemon = True
        self.t.start()

    @staticmethod
    def _fire(seconds, rank):
        print(f"bench.py rank {rank}: no result after {seconds:.0f} s (blocked in ncclCommInitRank / a collective?); exiting 124",
              file=sys.stderr, flush=True)
        os._exit(124)

    def cancel(self):
        self.t.cancel()


def launch_check(rank, world):
    """CPU rehearsal of the launcher's env plumbing (tests/test_bench_launcher.py): gloo group from the env the launcher set."""
    import torch
    import torch.distributed as dist
    if os.environ.get("SPAMTREE_LAUNCH_CHECK_HANG") == str(rank):      # rehearsal of a rank that never reaches the collective
        time.sleep(3600)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        mark_done()
        print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": float(t.item()),
                          "local_rank_env": os.environ.get("LOCAL_RANK"), "master": os.environ.get("MASTER_ADDR")}), flush=True)
    dist.destroy_process_group()


class ExternalChain:
    """Fallback for N > 1 when the library's own RCCL communicator cannot be set up: the same sharded phases with the
    collectives issued through torch.distributed on torch's stream (spamtree_amd/sharded.py) and the Python host driver
    (spamtree_amd/mcmc.py).  Same results; a Python loop instead of the C++ driver."""

    def __init__(self, wl, dist, device, k):
        import ctypes as C
        from spamtree_amd import mcmc
        from spamtree_amd.sharded import ShardedSpamTreeMV
        self._C = C
        self.mt = ShardedSpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"],
                                    wl["res_is_ref"], wl["parents"], wl["children"], False, wl["block_names"],
                                    wl["block_groups"], wl["indexing"], np.zeros(wl["n"]), np.zeros(wl["p"]), wl["theta"],
                                    1.0 / 0.1, device=device, dist=dist)
        self.pc = mcmc.Chain(self.mt, wl["bounds"], 0.01 * np.eye(k), seed=2021, adapting=True)

    def step(self, n=1):
        for _ in range(int(n)):
            self.pc.step()

    def state(self):
        return {"accept_ratio": float(self.pc.adaptivemc.accept_ratio)}

    def profile_levels_all(self):
        C = self._C
        nl = C.c_int32(); ms = np.zeros(128); by = np.zeros(128)
        self.mt.lib.st_profile_levels(self.mt.h, C.byref(nl), ms.ctypes.data_as(C.POINTER(C.c_double)),
                                      by.ctypes.data_as(C.POINTER(C.c_double)), 128)
        kk = nl.value
        return ms[:kk].copy(), by[:kk].copy(), ms[kk: 2 * kk].copy(), by[kk: 2 * kk].copy()

    def factor_ahead_levels(self):
        return 0          # the Python driver does not call st_factor_begin

    def __getattr__(self, name):      # algorithmic_bytes, profile, profile_get, profile_levels, synchronize, shard_info, close
        return getattr(self.mt, name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--side", type=int, default=1000, help="grid side; n = side^2 (1000 -> config #3, 316 -> #2)")
    ap.add_argument("--q", type=int, default=1)
    ap.add_argument("--cell-size", type=int, default=25, help="knots per cell (config #5: 9)")
    ap.add_argument("--missing", type=str, default="", help="per-outcome drop probabilities, e.g. 0.1,0.3,0.5 (config #5)")
    ap.add_argument("--limited-tree", action="store_true", help="limited_tree = TRUE: single-parent edges (tree_dep.cpp:133-186); a timing of that variant, not a BASELINE config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stationary-windows", type=int, default=12,
                    help="at most this many untimed 50-iteration windows after the timed region until one accepts >= 20 %% (0 = skip)")
    ap.add_argument("--cpu-side", type=int, default=316, help="grid side of the bounded CPU-baseline sample")
    ap.add_argument("--external", action="store_true", help="force the torch.distributed + Python-driver fallback path")
    ap.add_argument("--launch-check", action="store_true", help="CPU rehearsal of the rank launcher (gloo, no GPU work)")
    args = ap.parse_args()

    if "RANK" not in os.environ:
        if args.gpus > 1:
            launch_ranks(args, sys.argv[1:])          # does not return
        world_env = 1
    else:
        world_env = int(os.environ.get("WORLD_SIZE", "1"))
        if world_env != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world_env}")
    rank = int(os.environ.get("RANK", "0"))
    world = world_env
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        launch_check(rank, world)
        return
    import torch
    dist = None
    dog = Watchdog(float(os.environ.get("SPAMTREE_LAUNCH_DEADLINE", "240")), rank) if world > 1 else None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")

    from spamtree_amd import fit
    from spamtree_amd.synthetic import make_workload

    t_setup = time.time()
    missing = tuple(float(x) for x in args.missing.split(",")) if args.missing else None
    wl = make_workload(args.side, q=args.q, cell_size=args.cell_size, missing=missing, device=local_rank, limited_tree=args.limited_tree)
    # N > 1: one problem shared by all ranks -- subtrees below a cut level are owned by one GPU, the top is replicated,
    # exchanges are RCCL all-reduces issued by the library on its own stream (include/spamtree_hip.h, multi-GPU section)
    k = wl["theta"].size
    # the C++ host driver (spamtree_amd/csrc/spamtree_fit.cpp) steps the chain: w sweep, log-density, RAM-adaptive MH with a
    # full re-factorisation of the proposal slot, tausq and beta draws.  N > 1: the local part (st_create) first; the ranks
    # then AGREE that it succeeded everywhere before anyone enters the collective ncclCommInitRank (a rank that failed earlier
    # would leave the others blocked in its bootstrap; a failure inside that collective itself cannot be recovered from)
    chain, native_err = None, ""
    if not args.external:
        try:
            chain = fit.Chain(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"],
                              wl["res_is_ref"], wl["parents"], wl["children"], bool(args.limited_tree), wl["block_names"], wl["block_groups"],
                              wl["indexing"], wl["bounds"], wl["theta"], np.zeros(wl["p"]), 0.1, 0.01 * np.eye(k), seed=2021,
                              adapting=True, device=local_rank, rank=rank, world=world, defer_comm=True)
        except Exception as exc:      # noqa: BLE001  (N > 1 only: every rank must agree before falling back)
            if world == 1:
                raise
            native_err = repr(exc)

    def all_ok(flag):
        t = torch.tensor([1 if flag else 0], device="cuda", dtype=torch.int32)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    native = all_ok(chain is not None)
    if native and world > 1:
        box = [fit.make_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        chain.comm_init(box[0])
    if native:
        chain.start()
    host_path = "C++ host driver, collectives issued by the library (RCCL on its stream)" if world > 1 else "C++ host driver"
    if not native:
        if chain is not None:
            chain.close()
        if rank == 0:
            print("bench.py: native path unavailable on some rank (" + (native_err or "another rank failed")
                  + "); falling back to torch.distributed", file=sys.stderr)
        chain = ExternalChain(wl, dist, local_rank, k)
        host_path = "Python host driver, collectives through torch.distributed (fallback path)"
    model = chain
    n_blocks = int(np.asarray(wl["block_names"]).size)
    t_setup = time.time() - t_setup
    alg = model.algorithmic_bytes()

    model.profile(2)          # timed region: HIP events around the phase-A launches only (the roofline measurement)
    chain.step(args.warmup)
    model.profile_get()
    model.profile_levels()

    def fence():
        model.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    chain.step(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    prof = model.profile_get()
    lvl_ms, lvl_bytes = model.profile_levels()
    fac_ms, fac_n = prof["factor"]
    # phase breakdown of the other kernel families: a short untimed pass with every launch bracketed by events
    n_extra = max(1, min(10, args.steps))
    model.profile(1)
    if native:      # every level of phase A on the launch stream for this pass (under the sweep, on the second stream, the top
        from spamtree_amd import _lib as _l      # levels' launch times would not be their own)
        import ctypes as _C
        _l.load().st_factor_ahead_enable(_C.c_void_p(_l.load().stm_handle(chain.c)), 0)
    chain.step(n_extra)
    fence()
    prof_all = model.profile_get()
    lvl_ms, lvl_bytes, smp_ms, smp_bytes = model.profile_levels_all()   # per-level phase-A / phase-B times come from this pass too
    if native:
        _l.load().st_factor_ahead_enable(_C.c_void_p(_l.load().stm_handle(chain.c)), 1)
    # phase P (draws at the rows without an observation, spamtree_model.cpp:1234-1358; the reference runs it on every SAVED
    # iteration, spamtree_fit.cpp:300-306): not part of the timed iteration, timed on its own when the workload has such rows
    predict = None
    n_na = int(np.sum(~np.isfinite(wl["y"])))
    if n_na > 0 and native:
        import ctypes as C
        from spamtree_amd import _lib
        lib = _lib.load()
        for _ in range(4):
            lib.st_predict(C.c_void_p(lib.stm_handle(chain.c)), 1)
        pp = model.profile_get()["predict"]
        predict = {"rows": n_na, "ms_per_call": round(pp[0] / max(1, pp[1]), 4), "calls": pp[1],
                   "note": "st_predict on its own (HIP events), after the timed region; the leaf path of k_factor_quad where the prediction "
                           "blocks are eligible, else the generic kernel"}
    model.profile(0)
    n_levels = max(1, len(lvl_ms))
    avg_launch_ms = fac_ms / max(1, fac_n)
    share = 1.0
    if world > 1:
        info = model.shard_info()
        share = max(info["owned_rows"], 1) / float(wl["n"])        # this rank's part of the level launches (approximate)
    # the launches inside the timed phase-A bracket: all levels, or -- when the driver starts the latency-bound top levels
    # ahead of time on a second stream, under the sweep (st_factor_begin) -- the levels below them (99.6 % of the bytes)
    g_top = model.factor_ahead_levels()
    n_bracket = max(1, n_levels - g_top)
    bytes_bracket = float(np.sum(lvl_bytes[g_top:])) if g_top > 0 and len(lvl_bytes) == n_levels else alg["A"]
    bytes_per_launch = bytes_bracket / n_bracket * share
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    it_s = args.steps / dt
    # HBM bytes per k_factor launch from the PMC pass committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate passes, FETCH_SIZE doubled as the gfx950 guide prescribes); only valid for the workload it was taken on
    traffic, pmc_file, pmc_commit = None, None, None
    try:
        # the counter summary is named EXPLICITLY (profiles/pmc_source.json: file + the commit it was measured at), not
        # picked by sort order
        src = json.load(open(os.path.join(ROOT, "profiles", "pmc_source.json")))
        if world == 1 and args.side == 1000 and args.q == 1 and args.cell_size == 25 and not args.missing:
            pmc_file, pmc_commit = src["file"], src.get("commit")
            pmc = json.load(open(os.path.join(ROOT, pmc_file)))
            # per launch of the timed bracket: the k_factor_quad launches when the top levels run ahead, else all of phase A
            traffic = pmc["summary"]["k_factor_quad" if g_top > 0 else "phase_A"]["hbm_bytes_per_launch"]
    except Exception:      # noqa: BLE001
        traffic = None
    # what the counters say the kernel is bound by: measured HBM bytes over the launch time, next to the
    # contract's algorithmic-bytes rate (`achieved`) and the useful-flop rate of the FP64 pipe
    hbm_measured = None
    if traffic is not None and avg_launch_ms > 0:
        g = traffic / (avg_launch_ms * 1e-3) / 1e9
        hbm_measured = {"bytes_per_launch": traffic, "GBps": g, "frac": g / HBM_PEAK_GBS, "source": pmc_file,
                        "source_commit": pmc_commit,
                        "note": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, same command"}
    fp64_achieved = alg["flops_A"] * share / (avg_launch_ms * n_bracket * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
    all_ms = float(np.sum(lvl_ms))
    all_levels = {"ms": round(all_ms, 4), "launches": int(np.count_nonzero(np.asarray(lvl_ms) > 0)),
                  "GBps": round(float(np.sum(lvl_bytes)) * share / (all_ms * 1e-3) / 1e9, 1) if all_ms > 0 else 0.0,
                  "frac": round(float(np.sum(lvl_bytes)) * share / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if all_ms > 0 else 0.0,
                  "note": "every level of phase A run back to back on one stream (the per-launch pass), algorithmic bytes"}
    # box-measured peaks: a stream copy and the FP64 matrix / vector pipes, ~0.15 s, outside the timed region
    peak_measured = None
    try:
        import ctypes as C
        from spamtree_amd import _lib
        pk = np.zeros(3)
        if _lib.load().st_probe_peaks(local_rank, 1 << 30, 5, pk.ctypes.data_as(C.POINTER(C.c_double))) == 0:
            peak_measured = {"stream_copy_GBps": round(float(pk[0]), 1), "fp64_mfma_TFLOPs": round(float(pk[1]), 2),
                             "fp64_fma_TFLOPs": round(float(pk[2]), 2),
                             "frac_of_stream_copy": round(achieved / pk[0], 4) if pk[0] > 0 else None,
                             "fp64_frac_of_mfma": round(fp64_achieved / pk[1], 4) if pk[1] > 0 else None,
                             "note": "this box, this run: float4-wide copy of 1 GiB (read + written bytes), v_mfma_f64_16x16x4_f64 "
                                     "and v_fma_f64 loops at 2 waves per SIMD (csrc/probe.hip); `peak` stays the vendor figure"}
    except Exception as exc:      # noqa: BLE001
        peak_measured = {"error": repr(exc)}

    # stationary throughput: the timed window above starts at the data-generating theta with a 0.01 I proposal, where the
    # adaptation has not reached its 0.234 target yet, so it under-samples the sweeps that follow an ACCEPTED theta (the
    # records' Gram parts are rebuilt: dearer).  Untimed extra windows until one has accepted >= 20 % of its proposals
    # (bounded), then THAT window's rate; plus the cached / rebuild sweep times for an acceptance-weighted estimate.
    stationary = None
    if world == 1 and args.stationary_windows > 0:
        try:
            win = max(20, min(50, args.steps))
            st0 = chain.state()
            acc0, it0 = st0["accept_ratio"] * st0["iteration"], st0["iteration"]
            hist = []
            for wdx in range(args.stationary_windows):
                fence()
                tw = time.perf_counter()
                chain.step(win)
                fence()
                tw = time.perf_counter() - tw
                st1 = chain.state()
                acc1, it1 = st1["accept_ratio"] * st1["iteration"], st1["iteration"]
                ar = (acc1 - acc0) / max(1, it1 - it0)
                hist.append({"iterations": int(it1 - it0), "accept": round(float(ar), 3), "it_per_s": round(win / tw, 2)})
                acc0, it0 = acc1, it1
                if ar >= 0.2:
                    break
            stationary = {"value": hist[-1]["it_per_s"], "unit": "Gibbs iterations/s", "window_accept_ratio": hist[-1]["accept"],
                          "reached_0.2": bool(hist[-1]["accept"] >= 0.2), "windows": hist,
                          "note": f"windows of {win} iterations after the timed region, not part of `value`; the last one is reported"}
        except Exception as exc:      # noqa: BLE001
            stationary = {"error": repr(exc)}

    info1 = model.shard_info() if world > 1 else None
    out = {
        "metric": "Gibbs iterations/sec + achieved HBM GB/s, n=1e6 grid, 1/2/4/8 MI355X",
        "value": it_s, "unit": "Gibbs iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload_name(args, wl, n_blocks, n_levels, world),
                   "n": int(wl["n"]), "q": args.q, "blocks": int(n_blocks), "levels": int(n_levels), "host_path": host_path,
                   "mh_accept_ratio": float(chain.state()["accept_ratio"]),
                   "algorithmic_bytes_per_iter": alg["total"], "algorithmic_flops_per_iter":
                       alg["flops_A"] + alg["flops_B"] + alg["flops_C"], "setup_s": round(t_setup, 2)},
        "stationary": stationary,
        "multi_gpu": None if world == 1 else {
            "n_gpus": world, "rccl_ranks": int(info1["world"]), "cut_level": int(info1["cut_level"]),
            "owned_rows_rank0": int(info1["owned_rows"]), "native_rccl": bool(native),
            "collective_ms_per_iter": round(prof_all.get("comm", (0.0, 0))[0] / n_extra, 4),
            "collective_launches_per_iter": prof_all.get("comm", (0.0, 0))[1] / n_extra,
            "note": "rank 0's view; collective time = HIP events around the library's RCCL calls on its stream in the untimed "
                    "bracketed pass (includes waiting for the slowest rank)"},
        "roofline": {"bound": "mfma",
                     "bound_note": "what limits the kernel by the counters is the FP64 matrix / vector pipe (ONE pipe per SIMD: "
                                   "fp64_pipe below) and latency, not HBM (hbm_measured ~0.1 of peak).  achieved / peak / frac / unit "
                                   "stay the CONTRACT figure of north_star and SURVEY.md 8(d): algorithmic bytes per launch over the "
                                   "launch time against the 8 TB/s HBM roofline",
                     "contract_bound": "hbm",
                     "peak_measured": peak_measured,
                     "kernel": "k_factor (phase A: covariance build + chain solve + Cholesky)"
                     + (f"; levels {g_top}-{n_levels - 1} (levels 0-{g_top - 1}, 0.4 % of the bytes, run ahead of time on a second "
                        "stream under the sweep and are not in the timed bracket)" if g_top > 0 else ""),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "avg_launch_ms": avg_launch_ms, "launches": fac_n,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "whole_iteration_GBps": alg["total"] / (dt / args.steps) / 1e9,
                     "achieved_is": "ALGORITHMIC bytes (SURVEY.md 8d operand-streaming model) per launch / mean launch time",
                     "hbm_measured": hbm_measured,
                     "limiter": "FP64 pipe + latency, not HBM: measured HBM traffic is ~5x below the algorithmic bytes (chain panels are "
                                "served by L2 / Infinity Cache and shared by the units of a quad); MFMA and VALU FP64 share one pipe per SIMD",
                     "fp64_pipe": {"achieved": fp64_achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": fp64_achieved / FP64_PEAK_TFLOPS,
                                   "note": "algorithmic flops of phase A (no tile padding, covariance / Cholesky arithmetic not counted) "
                                           "over the bracket's time"},
                     "all_levels": all_levels,
                     "by_level_ms": [round(float(x), 4) for x in lvl_ms],
                     "by_level_GBps": [round(float(b / (m * 1e-3) / 1e9), 1) if m > 0 else 0.0
                                       for b, m in zip(lvl_bytes, lvl_ms)],
                     "by_level_frac": [round(float(b / (m * 1e-3) / 1e9 / HBM_PEAK_GBS), 3) if m > 0 else 0.0
                                       for b, m in zip(lvl_bytes, lvl_ms)],   # the same ratio per launch (one kernel per level)
                     "sample_by_level_ms": [round(float(x), 4) for x in smp_ms],
                     "sample_by_level_GBps": [round(float(b / (m * 1e-3) / 1e9), 1) if m > 0 else 0.0
                                              for b, m in zip(smp_bytes, smp_ms)],
                     "phase_ms_per_iter": {kk: round((prof[kk][0] / args.steps) if kk == "factor" else (v[0] / n_extra), 4)
                                           for kk, v in prof_all.items()},
                     "predict": predict,
                     "phase_ms_note": "factor (and avg_launch_ms, achieved): one HIP-event pair around each phase A of the timed "
                                      "steps; by_level_ms and the other families: an untimed pass of "
                                      f"{n_extra} steps with every launch bracketed (event traffic costs 4-6 % of an iteration)"},
    }
    if dog is not None:
        dog.cancel()
    if rank == 0:
        mark_done()
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(wl, args.cpu_side)
            except Exception as exc:          # noqa: BLE001  (a missing checker must not lose the GPU line)
                out["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    model.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
