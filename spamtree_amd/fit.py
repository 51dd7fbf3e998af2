"""Python face of the C++ host driver (spamtree_amd/csrc/spamtree_fit.cpp, include/spamtree_fit.h): the same
`spamtree_mv_mcmc(...)` argument list and returned names as the reference's Rcpp export
(/root/reference/src/spamtree_fit.cpp:5-54, 403-414), and a steppable `Chain` for benchmarking."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .model import SpamTreeError, _dp, _f64, _i64, _ip, _lists_to_csr


def _problem(y, X, coords, mv_id, res_is_ref, parents, children, block_names, block_groups, indexing):
    y = _f64(np.asarray(y).reshape(-1))
    X = np.asfortranarray(np.asarray(X, dtype=np.float64))
    coords = np.asfortranarray(np.asarray(coords, dtype=np.float64))
    mv_id = _i64(mv_id)
    ip, ii = indexing if isinstance(indexing, tuple) else _lists_to_csr(indexing)
    pp, pi = parents if isinstance(parents, tuple) else _lists_to_csr(parents)
    cp, ci = children if isinstance(children, tuple) else _lists_to_csr(children)
    keep = [y, X, coords, mv_id, _i64(res_is_ref), _i64(block_names), _i64(block_groups), _i64(ip), _i64(ii), _i64(pp),
            _i64(pi), _i64(cp), _i64(ci)]
    n, p = X.shape
    q = int(np.unique(mv_id).size)
    pb = _lib.StProblem(n, coords.shape[1], q, p, int(keep[4].size), int(keep[5].size), _dp(y), _dp(X), _dp(coords),
                        _ip(mv_id), *[_ip(a) for a in keep[4:]])
    return pb, keep, n, p, q


def make_unique_id():
    """128-byte RCCL unique id (rank 0 creates it; broadcast it to the other ranks, e.g. with torch.distributed)."""
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    n = lib.st_comm_unique_id(C.cast(buf, C.c_void_p), 128)
    if n <= 0:
        raise SpamTreeError("st_comm_unique_id failed")
    return bytes(buf.raw[:128])


class Chain:
    """stm_chain: SpamTreeMV + RAMAdapt + loop state, stepped from C++ (one ctypes call per `step(n)`)."""

    def __init__(self, y, X, Z, coords, mv_id, blocking, gix_block, res_is_ref, parents, children, limited_tree,
                 block_names, block_groups, indexing, set_unif_bounds, theta, beta, tausq, mcmcsd, seed=2021,
                 adapting=True, sample_beta=True, sample_tausq=True, sample_theta=True, sample_w=True, device=0,
                 reference_quirks=True, rank=0, world=1, unique_id=None, defer_comm=False):
        self.lib = _lib.load()
        pb, self._keep, self.n, self.p, self.q = _problem(y, X, coords, mv_id, res_is_ref, parents, children,
                                                          block_names, block_groups, indexing)
        theta = _f64(theta)
        self.k = theta.size
        bounds = np.asfortranarray(np.asarray(set_unif_bounds, dtype=np.float64))
        sd = np.asfortranarray(np.asarray(mcmcsd, dtype=np.float64))
        opt = _lib.StOptions(int(device), int(bool(reference_quirks)), int(rank), int(world), 0, 2 if limited_tree else 0)
        fl = _lib.StmFlags(int(adapting), int(sample_beta), int(sample_tausq), int(sample_theta), int(sample_w), 1)
        c = C.c_void_p()
        self.c = None
        if world > 1 and unique_id is None and not defer_comm:
            raise SpamTreeError("world > 1 needs the RCCL unique id of rank 0 (spamtree_amd.fit.make_unique_id)")
        rc = self.lib.stm_create(C.byref(pb), C.byref(opt), _dp(bounds), _dp(sd), _dp(theta), self.k, _dp(_f64(beta)),
                                 float(tausq), int(seed), C.byref(fl), C.byref(c))
        self.c = c
        if rc != 0:
            msg = self.lib.stm_last_error(c).decode() if c else self.lib.st_last_error(None).decode()
            if c:
                self.lib.stm_destroy(c)
                self.c = None
            raise SpamTreeError(f"stm_create failed ({rc}): {msg or self.lib.st_last_error(None).decode()}")
        self.h = C.c_void_p(self.lib.stm_handle(self.c))
        self.rank, self.world = int(rank), int(world)
        # defer_comm: stop after the local part (st_create), so that the ranks can first agree that it succeeded everywhere:
        # ncclCommInitRank is collective and a rank that failed before it would leave the others blocked in the bootstrap.
        # The caller then runs comm_init(unique_id) -- a failure INSIDE that collective cannot be recovered from -- and start().
        if defer_comm:
            return
        if world > 1 or unique_id is not None:   # a unique id with world == 1: the RCCL protocol path on a single rank (tests)
            self.comm_init(unique_id)
        self.start()

    def comm_init(self, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        rc = self.lib.st_comm_init(self.h, C.cast(buf, C.c_void_p))
        if rc != 0:
            raise SpamTreeError(f"st_comm_init failed ({rc}): {self.lib.st_last_error(self.h).decode()}")

    def start(self):
        rc = self.lib.stm_init(self.c)
        if rc != 0:
            raise SpamTreeError(f"stm_init failed ({rc}): {self.lib.stm_last_error(self.c).decode()}")

    def step(self, n=1):
        rc = self.lib.stm_step(self.c, int(n))
        if rc != 0:
            raise SpamTreeError(f"stm_step failed ({rc}): {self.lib.stm_last_error(self.c).decode()}")

    def state(self):
        theta = np.zeros(self.k); B = np.zeros(self.p * self.q); tsq = np.zeros(self.q); sd = np.zeros(self.k * self.k)
        ll, ar, it = C.c_double(), C.c_double(), C.c_int64()
        self.lib.stm_state(self.c, _dp(theta), _dp(B), _dp(tsq), C.byref(ll), C.byref(ar), C.byref(it), _dp(sd))
        return dict(theta=theta, Bcoeff=B.reshape(self.q, self.p).T.copy(), tausq_inv=tsq, loglik=ll.value,
                    accept_ratio=ar.value, iteration=it.value, paramsd=sd.reshape(self.k, self.k).T.copy())

    def get_w(self):
        out = np.zeros(self.n)
        self.lib.st_get_w(self.h, _dp(out))
        return out

    # measurement helpers (same as SpamTreeMV's)
    def algorithmic_bytes(self):
        out, fl = np.zeros(5), np.zeros(3)
        self.lib.st_algorithmic_bytes(self.h, _dp(out), _dp(fl))
        return dict(A=out[0], B=out[1], C=out[2], msg=out[3], S=out[4], total=float(out.sum()), flops_A=fl[0],
                    flops_B=fl[1], flops_C=fl[2])

    def profile(self, enable):
        self.lib.st_profile_enable(self.h, 2 if enable == 2 else int(bool(enable)))

    def profile_get(self):
        ms, n = np.zeros(8), np.zeros(8, dtype=np.int64)
        self.lib.st_profile_get(self.h, _dp(ms), _ip(n))
        names = ["factor", "sample", "loglik", "reduce", "stats", "rng", "predict", "comm"]
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def profile_levels(self):
        nl = C.c_int32(); ms = np.zeros(64); by = np.zeros(64)
        self.lib.st_profile_levels(self.h, C.byref(nl), _dp(ms), _dp(by), 64)
        return ms[: nl.value].copy(), by[: nl.value].copy()

    def profile_levels_all(self):
        """(phase A ms, phase A algorithmic bytes, phase B ms, phase B + message bytes) per level since the last call."""
        nl = C.c_int32(); ms = np.zeros(128); by = np.zeros(128)
        self.lib.st_profile_levels(self.h, C.byref(nl), _dp(ms), _dp(by), 128)
        k = nl.value
        return ms[:k].copy(), by[:k].copy(), ms[k: 2 * k].copy(), by[k: 2 * k].copy()

    def factor_ahead_levels(self):
        """Leading levels whose phase A the driver starts before the sweep (st_factor_begin); 0 = none."""
        return int(self.lib.st_factor_ahead_levels(self.h))

    def synchronize(self):
        self.lib.st_synchronize(self.h)

    def shard_info(self):
        r, w, c = C.c_int32(), C.c_int32(), C.c_int32()
        ob, orow = C.c_int64(), C.c_int64()
        self.lib.st_shard_info(self.h, C.byref(r), C.byref(w), C.byref(c), C.byref(ob), C.byref(orow))
        return dict(rank=r.value, world=w.value, cut_level=c.value, owned_blocks=ob.value, owned_rows=orow.value)

    def close(self):
        if self.c:
            self.lib.stm_destroy(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def spamtree_mv_mcmc(y, X, Z, coords, mv_id, blocking, gix_block, res_is_ref, parents, children, limited_tree,
                     layer_names, layer_gibbs_group, indexing, set_unif_bounds_in, start_w, theta, beta, tausq, mcmcsd,
                     mcmc_keep=100, mcmc_burn=100, mcmc_thin=1, num_threads=1, use_alg="S", adapting=False,
                     main_verbose=True, verbose=False, debug=False, printall=False, sample_beta=True, sample_tausq=True,
                     sample_theta=True, sample_w=True, sample_predicts=True, seed=2021, device=0, reference_quirks=True):
    """spamtree_fit.cpp:5-430 through the C++ driver.  `num_threads`, `use_alg`, the verbosity flags and `start_w` are
    accepted and ignored exactly where the reference ignores them (start_w, :95) or where they do not apply to a GPU."""
    lib = _lib.load()
    pb, keep, n, p, q = _problem(y, X, coords, mv_id, res_is_ref, parents, children, layer_names, layer_gibbs_group, indexing)
    theta = _f64(theta)
    k = theta.size
    bounds = np.asfortranarray(np.asarray(set_unif_bounds_in, dtype=np.float64))
    sd = np.asfortranarray(np.asarray(mcmcsd, dtype=np.float64))
    opt = _lib.StOptions(int(device), int(bool(reference_quirks)), 0, 1, 0, 2 if limited_tree else 0)
    fl = _lib.StmFlags(int(adapting), int(sample_beta), int(sample_tausq), int(sample_theta), int(sample_w), int(sample_predicts))
    w_all = np.zeros((n, mcmc_keep), order="F"); yh_all = np.zeros((n, mcmc_keep), order="F")
    beta_mcmc = np.zeros((p, mcmc_keep, q), order="F"); tausq_mcmc = np.zeros((q, mcmc_keep), order="F")
    theta_mcmc = np.zeros((k, mcmc_keep), order="F"); paramsd = np.zeros((k, k), order="F")
    t = C.c_double()
    rc = lib.spamtree_mv_mcmc_c(C.byref(pb), C.byref(opt), _dp(bounds), _dp(theta), k, _dp(_f64(beta)), float(tausq), _dp(sd),
                                int(mcmc_keep), int(mcmc_burn), int(mcmc_thin), int(seed), C.byref(fl), _dp(w_all), _dp(yh_all),
                                _dp(beta_mcmc), _dp(tausq_mcmc), _dp(theta_mcmc), _dp(paramsd), C.byref(t))
    if rc == -10:
        raise FloatingPointError("At nan loglik: error.")
    if rc != 0:
        if main_verbose:
            print("MCMC has been interrupted.")
        return {"None": np.zeros(0)}
    return dict(w_mcmc=[w_all[:, i].reshape(-1, 1).copy() for i in range(mcmc_keep)],
                yhat_mcmc=[yh_all[:, i].reshape(-1, 1).copy() for i in range(mcmc_keep)], beta_mcmc=beta_mcmc,
                tausq_mcmc=tausq_mcmc, theta_mcmc=theta_mcmc, paramsd=paramsd, mcmc_time=t.value)
