"""Worker for the multi-process tests (spawned with torch.multiprocessing; gloo backend, 127.0.0.1 rendezvous)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def plan_worker(rank, world, port, case, out_dir):
    """CPU only: the library's pure-host ownership plan + the sum-with-zeros exchange, with the oracle as compute."""
    import torch
    dist = _init(rank, world, port)
    from spamtree_amd import _lib
    from spamtree_amd.model import _lists_to_csr, _i64, _f64, _dp, _ip
    from spamtree_amd.sharded import shard_plan
    from tests.util import make_problem, oracle_model
    pb = make_problem(**case)
    y = _f64(pb["y"]); X = np.asfortranarray(pb["X"]); co = np.asfortranarray(pb["coords"]); mv = _i64(pb["mv_id"])
    ip, ii = _lists_to_csr(pb["indexing"]); pp, pi = _lists_to_csr(pb["parents"]); cp, ci = _lists_to_csr(pb["children"])
    keep = [_i64(pb["res_is_ref"]), _i64(pb["block_names"]), _i64(pb["block_groups"]), ip, ii, pp, pi, cp, ci]
    st = _lib.StProblem(pb["n"], 2, pb["q"], pb["p"], int(keep[0].size), int(keep[1].size), _dp(y), _dp(X), _dp(co), _ip(mv),
                        *[_ip(a) for a in keep])
    owner, cut = shard_plan(st, world, limited_tree=bool(case.get("limited_tree", False)))
    nb = owner.size
    levels = np.unique(pb["block_groups"])
    lev_of = np.searchsorted(levels, pb["block_groups"])
    # ---- plan invariants
    assert np.all(owner[lev_of < cut] == -1), "levels above the cut must be replicated"
    obs_blocks = np.array([np.isfinite(pb["y"][ix]).any() for ix in pb["indexing"]])
    below = (lev_of >= cut)
    assert np.all(owner[below] >= 0) and np.all(owner[below] < world)
    for u in range(nb):                                  # a block below the cut inherits its cut-level ancestor's rank
        if lev_of[u] > cut:
            a = u                                        # walk the direct parents up to the cut level (full lists: the last entry;
            while lev_of[a] > cut:                       # make_edges_limited's lists: the only one)
                a = pb["parents"][a][-1]
            assert lev_of[a] == cut and owner[a] == owner[u]
    if cut < levels.size:
        assert len(set(owner[below].tolist())) == world, "every rank owns at least one subtree"
    # ---- exchange: every rank masks a full oracle result to what it owns; the all-reduce restores it bit for bit
    rng = np.random.default_rng(9)
    om = oracle_model(pb, w=rng.standard_normal(pb["n"]), tausq=0.2)
    assert om.get_loglik_comps_w(om.param_data)
    om.gibbs_sample_w(rng.standard_normal(pb["n"]))
    mine_blk = (owner == rank) | ((owner == -1) & (rank == 0))
    comps = np.concatenate([np.where(mine_blk, om.param_data.logdetCi_comps, 0.0), np.where(mine_blk, om.param_data.loglik_w_comps, 0.0)])
    rowmask = np.zeros(pb["n"], dtype=bool)
    for u in range(nb):
        if mine_blk[u]:
            rowmask[pb["indexing"][u]] = True
    wbuf = np.where(rowmask, om.w, 0.0)
    tc, tw = torch.from_numpy(comps.copy()), torch.from_numpy(wbuf.copy())
    dist.all_reduce(tc); dist.all_reduce(tw)
    full = np.concatenate([om.param_data.logdetCi_comps, om.param_data.loglik_w_comps])
    assert np.array_equal(tc.numpy(), full) and np.array_equal(tw.numpy(), om.w)
    cnt = torch.from_numpy(rowmask.astype(np.int64)); dist.all_reduce(cnt)
    assert np.all(cnt.numpy() == 1), "every row is contributed by exactly one rank"
    np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([cut]))
    dist.destroy_process_group()


def gpu_worker(rank, world, port, side, q, out_dir, steps, limited=False):
    """Needs a GPU: `world` processes share device 0 and one problem; results go to out_dir for the parent to compare."""
    dist = _init(rank, world, port)
    from spamtree_amd.sharded import ShardedSpamTreeMV
    from spamtree_amd.synthetic import make_workload
    wl = make_workload(side, q=q, limited_tree=limited)
    m = ShardedSpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"],
                          wl["res_is_ref"], wl["parents"], wl["children"], bool(limited), wl["block_names"], wl["block_groups"],
                          wl["indexing"], np.zeros(wl["n"]), np.array([-0.5, 0.2, 0.4]), wl["theta"], 1.0 / 0.15,
                          device=0, dist=dist if world > 1 else None, allreduce_w=(world == 3))   # both forms of the w exchange
    rng = np.random.default_rng(3)
    m.set_w(rng.standard_normal(wl["n"]))
    res = {}
    assert m.get_loglik_comps_w(0)
    res["ll_A"] = m.loglik_w[0]
    for it in range(steps):
        if it % 2 == 1 and world > 1:             # the fused order: phase C before the exchange of w, one exchange for both
            res[f"ll_C{it}"] = m.deal_with_w_loglik(0, None, seed=5, it=it)
        else:
            m.deal_with_w(None, seed=5, it=it)
            res[f"ll_C{it}"] = m.get_loglik_w(0)
    th2 = wl["theta"] * 1.03
    m.theta_update(1, th2)
    assert m.get_loglik_comps_w(1)
    res["ll_A2"] = m.loglik_w[1]
    res["err"] = 0.0
    if q == 1:                                   # negative sigma^2: every rank must agree on the failure code
        bad = wl["theta"].copy(); bad[0] = -1.0
        m.theta_update(1, bad)
        assert m.get_loglik_comps_w(1) is False
        res["err"] = float(m.last_errtype)
    xty, ssq = m.stats()
    info = m.shard_info() if world > 1 else {}
    kernels = np.array([{"generic_lds": 0, "generic_scratch": 1, "k_factor_mfma": 2, "k_factor_quad": 3, "k_factor_bigmfma": 4,
                         "k_factor_wide": 5, "k_factor_lchain": 6, "k_factor_lchain+ref_finish": 7}[L["kernel"]] for L in m.level_info()])
    np.savez(os.path.join(out_dir, f"res_{world}_{rank}.npz"), w=m.get_w(), xty=xty, ssq=ssq,
             owned_rows=info.get("owned_rows", wl["n"]), kernels=kernels, **res)
    m.close()
    dist.destroy_process_group()
