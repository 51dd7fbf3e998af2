"""Host-side mirror of the reference's `SpamTreeMV` (/root/reference/src/spamtree_model.h:22-212) over the C-ABI.

Same constructor arguments, method names and failure behaviour as the reference class, so parity tests read like
calls into the reference; every hot method is one C-ABI call into the HIP library.  Nothing here computes on the
CPU: without the library and a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _dp(a):
    return a.ctypes.data_as(_lib.c_dp)


def _ip(a):
    return a.ctypes.data_as(_lib.c_ip)


def _lists_to_csr(lists):
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(x) for x in lists])
    idx = np.concatenate([np.asarray(x, dtype=np.int64) for x in lists]) if len(lists) and ptr[-1] > 0 \
        else np.zeros(0, dtype=np.int64)
    return ptr, _i64(idx)


class SpamTreeError(RuntimeError):
    pass


class SpamTreeMV:
    """spamtree_model.cpp:8-192.  `param_data` / `alter_data` are slots 0 / 1 of the device handle."""

    PARAM, ALTER = 0, 1

    def __init__(self, y, X, Z, coords, mv_id, blocking, gix_block, res_is_ref, parents, children, limited_tree,
                 block_names, block_groups, indexing, w, beta, theta, tausq_inv,
                 device=0, reference_quirks=True, force_generic=False, rank=0, world=1, cache_gram=True):
        self.lib = _lib.load()
        y = _f64(np.asarray(y).reshape(-1))
        X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        coords = np.asfortranarray(np.asarray(coords, dtype=np.float64))
        mv_id = _i64(mv_id)
        self.n_all, self.p = X.shape
        self.q = int(np.unique(mv_id).size)
        self.dd = coords.shape[1]
        ip, ii = indexing if isinstance(indexing, tuple) else _lists_to_csr(indexing)
        pp, pi = parents if isinstance(parents, tuple) else _lists_to_csr(parents)
        cp, ci = children if isinstance(children, tuple) else _lists_to_csr(children)
        keep = [y, X, coords, mv_id, _i64(res_is_ref), _i64(block_names), _i64(block_groups), _i64(ip), _i64(ii),
                _i64(pp), _i64(pi), _i64(cp), _i64(ci)]
        pb = _lib.StProblem(self.n_all, self.dd, self.q, self.p, int(keep[4].size), int(keep[5].size),
                            _dp(y), _dp(X), _dp(coords), _ip(mv_id), _ip(keep[4]), _ip(keep[5]), _ip(keep[6]),
                            _ip(keep[7]), _ip(keep[8]), _ip(keep[9]), _ip(keep[10]), _ip(keep[11]), _ip(keep[12]))
        opt = _lib.StOptions(int(device), int(bool(reference_quirks)), int(rank), int(world), int(bool(force_generic)),
                              (0 if cache_gram else 1) | (2 if limited_tree else 0))
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        rc = self.lib.st_create(C.byref(pb), C.byref(opt), C.byref(h))
        if rc != 0:
            raise SpamTreeError(f"st_create failed ({rc}): {self.lib.st_last_error(None).decode()}")
        self.h = h
        self.n_blocks = int(keep[5].size)
        self.theta = [np.asarray(theta, dtype=np.float64).copy(), np.asarray(theta, dtype=np.float64).copy()]
        self.loglik_w = [float("nan"), float("nan")]
        self.Bcoeff = np.zeros((self.p, self.q), order="F")
        beta = np.asarray(beta, dtype=np.float64).reshape(-1)
        for j in range(self.q):
            self.Bcoeff[:, j] = beta
        self.beta_update(self.Bcoeff)
        self.tausq_inv = np.ones(self.q) * float(tausq_inv)
        self._check(self.lib.st_set_tausq_inv(self.h, _dp(self.tausq_inv)))
        self.w = np.asarray(w, dtype=np.float64).reshape(-1).copy()
        self._check(self.lib.st_set_w(self.h, _dp(_f64(self.w))))
        xtx = np.zeros(self.p * self.p * self.q)
        self._check(self.lib.st_xtx(self.h, _dp(xtx)))
        self.XtX = [xtx[j * self.p * self.p:(j + 1) * self.p * self.p].reshape(self.p, self.p).T.copy()
                    for j in range(self.q)]
        self.Vi = 0.01 * np.eye(self.p)
        self.Vim = np.zeros(self.p)
        nq = np.zeros(self.q, dtype=np.int64)
        ssq = np.zeros(self.q)
        self._check(self.lib.st_tausq_stats(self.h, _dp(ssq), _ip(nq)))
        self.n_obs_by_q = nq

    def close(self):
        if getattr(self, "h", None):
            self.lib.st_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise SpamTreeError(f"spamtree_hip error {rc}: {self.lib.st_last_error(self.h).decode()}")
        return rc

    # ---- spamtree_model.cpp:1420-1422
    def theta_update(self, slot, new_param):
        self.theta[slot] = np.asarray(new_param, dtype=np.float64).copy()

    # ---- spamtree_model.cpp:829-998 ; returns the reference's bool
    def get_loglik_comps_w(self, slot) -> bool:
        th = _f64(self.theta[slot])
        ll = C.c_double(0.0)
        rc = self._check(self.lib.st_factor(self.h, slot, _dp(th), th.size, C.byref(ll)))
        self.last_errtype = rc if rc > 0 else -1
        if rc > 0:
            return False
        self.loglik_w[slot] = ll.value
        return True

    # ---- spamtree_model.cpp:1000-1226
    def deal_with_w(self, z=None, seed=0, it=0):
        if z is not None:
            z = _f64(z)
            rc = self._check(self.lib.st_sample_w(self.h, _dp(z), 0, 0))
        else:
            rc = self._check(self.lib.st_sample_w(self.h, None, int(seed), int(it)))
        if rc > 0:
            raise SpamTreeError("Error at gibbs_sample_w")           # Rcpp::stop (:1215-1217)

    gibbs_sample_w = deal_with_w

    # ---- spamtree_model.cpp:776-826
    def get_loglik_w(self, slot):
        ll = C.c_double(0.0)
        self._check(self.lib.st_loglik_w(self.h, slot, C.byref(ll)))
        self.loglik_w[slot] = ll.value
        return ll.value

    # ---- spamtree_model.cpp:1229-1358
    def predict(self, theta_update=True):
        self._check(self.lib.st_predict(self.h, int(bool(theta_update))))

    # ---- spamtree_model.cpp:1364-1391 (normals_by_q[j] = arma::randn(p) of :1378)
    def gibbs_sample_beta(self, normals_by_q):
        xty = np.zeros(self.p * self.q)
        self._check(self.lib.st_beta_stats(self.h, _dp(xty)))
        xty = xty.reshape(self.q, self.p).T
        for j in range(self.q):
            Si_chol = np.linalg.cholesky(self.tausq_inv[j] * self.XtX[j] + self.Vi)
            Sc = np.linalg.solve(Si_chol, np.eye(self.p))
            Xprecy_j = self.Vim + self.tausq_inv[j] * xty[:, j]
            self.Bcoeff[:, j] = Sc.T @ (Sc @ Xprecy_j) + Sc.T @ np.asarray(normals_by_q[j], dtype=np.float64)
        self.beta_update(self.Bcoeff)

    deal_with_beta = gibbs_sample_beta

    # ---- spamtree_model.cpp:1393-1417 (gamma_draw(j, shape, scale) = R::rgamma)
    def gibbs_sample_tausq(self, gamma_draw):
        ssq = np.zeros(self.q)
        self._check(self.lib.st_tausq_stats(self.h, _dp(ssq), None))
        for j in range(self.q):
            aparam = 2.01 + self.n_obs_by_q[j] / 2.0
            bparam = 1.0 / (1.0 + 0.5 * ssq[j])
            self.tausq_inv[j] = gamma_draw(j, aparam, bparam)
        self._check(self.lib.st_set_tausq_inv(self.h, _dp(_f64(self.tausq_inv))))

    def tausq_update(self, new_tausq):                                  # :1424-1426
        self.tausq_inv = np.ones(self.q) / float(new_tausq)
        self._check(self.lib.st_set_tausq_inv(self.h, _dp(_f64(self.tausq_inv))))

    def beta_update(self, new_beta):                                    # :1428-1430 (+ XB refresh)
        self.Bcoeff = np.asfortranarray(np.asarray(new_beta, dtype=np.float64).reshape(self.p, self.q))
        self._check(self.lib.st_set_beta(self.h, _dp(self.Bcoeff)))

    def accept_make_change(self):                                       # :1432-1435
        self._check(self.lib.st_swap(self.h))
        self.theta[0], self.theta[1] = self.theta[1], self.theta[0]
        self.loglik_w[0], self.loglik_w[1] = self.loglik_w[1], self.loglik_w[0]

    # ---- public fields of the reference object
    def get_w(self):
        out = np.zeros(self.n_all)
        self._check(self.lib.st_get_w(self.h, _dp(out)))
        self.w = out
        return out

    def set_w(self, w):
        self.w = np.asarray(w, dtype=np.float64).copy()
        self._check(self.lib.st_set_w(self.h, _dp(_f64(self.w))))

    def get_XB(self):
        out = np.zeros(self.n_all)
        self._check(self.lib.st_get_xb(self.h, _dp(out)))
        return out

    def stats(self):
        xty = np.zeros(self.p * self.q)
        ssq = np.zeros(self.q)
        self._check(self.lib.st_beta_stats(self.h, _dp(xty)))
        self._check(self.lib.st_tausq_stats(self.h, _dp(ssq), None))
        return xty.reshape(self.q, self.p).T.copy(), ssq

    def yhat(self, noise=None, seed=0, it=0):
        out = np.zeros(self.n_all)
        if noise is not None:
            noise = _f64(noise)
            self._check(self.lib.st_yhat(self.h, _dp(noise), 0, 0, _dp(out)))
        else:
            self._check(self.lib.st_yhat(self.h, None, int(seed), int(it), _dp(out)))
        return out

    # ---- inspection for parity tests
    def block(self, slot, u):
        """(H, Ri) of block u: H = K_{u,pa} K_{pa,pa}^{-1} (m x P) recovered from the stored panel, Ri = chol(R)^{-1}
        (m x m) for a reference block or the m per-row values 1/sqrt(r_ii) for a non-reference block."""
        m, P = C.c_int64(), C.c_int64()
        isref, nobs = C.c_int32(), C.c_int32()
        self._check(self.lib.st_block_dims(self.h, u, C.byref(m), C.byref(P), C.byref(isref), C.byref(nobs)))
        m, P = m.value, P.value
        N = np.zeros(m * max(P, 1))
        Ri = np.zeros(m * m if isref.value else m)
        self._check(self.lib.st_get_block(self.h, slot, u, _dp(N), _dp(Ri)))
        N = N[: m * P].reshape(P, m).T if P else np.zeros((m, 0))
        if isref.value:
            Ri = Ri.reshape(m, m).T
            H = -np.linalg.solve(Ri, N) if P else N
        else:
            H = -N / Ri[:, None] if P else N
        return H, Ri

    def comps(self, slot):
        a = np.zeros(self.n_blocks)
        b = np.zeros(self.n_blocks)
        self._check(self.lib.st_get_comps(self.h, slot, _dp(a), _dp(b)))
        return a, b

    def algorithmic_bytes(self):
        out = np.zeros(5)
        fl = np.zeros(3)
        self._check(self.lib.st_algorithmic_bytes(self.h, _dp(out), _dp(fl)))
        return dict(A=out[0], B=out[1], C=out[2], msg=out[3], S=out[4], total=float(out.sum()),
                    flops_A=fl[0], flops_B=fl[1], flops_C=fl[2])

    def profile(self, enable):
        self._check(self.lib.st_profile_enable(self.h, 2 if enable == 2 else int(bool(enable))))

    def profile_get(self):
        ms = np.zeros(8)
        n = np.zeros(8, dtype=np.int64)
        self._check(self.lib.st_profile_get(self.h, _dp(ms), _ip(n)))
        names = ["factor", "sample", "loglik", "reduce", "stats", "rng", "predict", "comm"]
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def profile_levels(self):
        nl = C.c_int32()
        ms = np.zeros(64)
        by = np.zeros(64)
        self._check(self.lib.st_profile_levels(self.h, C.byref(nl), _dp(ms), _dp(by), 64))
        return ms[: nl.value].copy(), by[: nl.value].copy()

    def profile_levels_sample(self):
        """Mean k_sample* launch time per level (ms) since the last call."""
        nl = C.c_int32()
        ms = np.zeros(128)
        by = np.zeros(128)
        self._check(self.lib.st_profile_levels(self.h, C.byref(nl), _dp(ms), _dp(by), 128))
        return ms[nl.value: 2 * nl.value].copy(), by[nl.value: 2 * nl.value].copy()

    KERNEL_NAMES = ["generic_lds", "generic_scratch", "k_factor_mfma", "k_factor_quad", "k_factor_bigmfma", "k_factor_wide", "k_factor_lchain", "k_factor_lchain+ref_finish"]

    def level_info(self):
        """Per observed level: dict(kernel=name of the phase-A kernel, max_m, max_P, n_blocks)."""
        nl = C.c_int32()
        arr = [np.zeros(64, dtype=np.int32) for _ in range(4)]
        ptr = [a.ctypes.data_as(C.POINTER(C.c_int32)) for a in arr]
        self._check(self.lib.st_level_info(self.h, C.byref(nl), ptr[0], ptr[1], ptr[2], ptr[3], 64))
        return [dict(kernel=self.KERNEL_NAMES[arr[0][g]], max_m=int(arr[1][g]), max_P=int(arr[2][g]), n_blocks=int(arr[3][g]))
                for g in range(nl.value)]

    def synchronize(self):
        self._check(self.lib.st_synchronize(self.h))
