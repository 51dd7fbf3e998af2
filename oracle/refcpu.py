"""ORACLE (test infrastructure): ctypes wrapper of oracle/refcpu.cpp, the OpenMP restatement of the reference
algorithm as written.  Used by tests/ and by bench.py's cpu_baseline leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "librefcpu.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "refcpu.cpp")):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = C.CDLL(LIB)
        _lib.refcpu_create.restype = C.c_void_p
        _lib.refcpu_create.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_longlong, C.c_int, _dp, _dp, _dp, _ip, _ip, _ip,
                                       _ip, _ip, _ip, _ip, _ip, _ip, _ip, C.c_int, C.c_int, C.c_int]
        _lib.refcpu_destroy.argtypes = [C.c_void_p]
        _lib.refcpu_set_threads.argtypes = [C.c_int]
        _lib.refcpu_factor.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp]
        _lib.refcpu_sample_w.argtypes = [C.c_void_p, _dp]
        _lib.refcpu_loglik_w.restype = C.c_double
        _lib.refcpu_loglik_w.argtypes = [C.c_void_p, C.c_int]
        _lib.refcpu_swap.argtypes = [C.c_void_p]
        for f in ("refcpu_set_w", "refcpu_get_w", "refcpu_set_tausq_inv", "refcpu_set_beta"):
            getattr(_lib, f).argtypes = [C.c_void_p, _dp]
        _lib.refcpu_stats.argtypes = [C.c_void_p, _dp, _dp]
        _lib.refcpu_get_block.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp]
        _lib.refcpu_get_comps.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    return _lib


def _csr(x):
    if isinstance(x, tuple):
        return np.ascontiguousarray(x[0], dtype=np.int64), np.ascontiguousarray(x[1], dtype=np.int64)
    ptr = np.zeros(len(x) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(v) for v in x])
    idx = np.concatenate([np.asarray(v, dtype=np.int64) for v in x]) if ptr[-1] > 0 else np.zeros(0, dtype=np.int64)
    return ptr, np.ascontiguousarray(idx, dtype=np.int64)


class RefCpu:
    def __init__(self, y, X, coords, mv_id, res_is_ref, parents, children, block_names, block_groups, indexing,
                 reference_distance=False, reference_quirks=True, threads=0, limited_tree=False):
        self.lib = load()
        self.y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        self.X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        self.coords = np.asfortranarray(np.asarray(coords, dtype=np.float64))
        self.n, self.p = self.X.shape
        mv = np.ascontiguousarray(mv_id, dtype=np.int64)
        self.q = int(np.unique(mv).size)
        arrs = [mv, np.ascontiguousarray(res_is_ref, dtype=np.int64), np.ascontiguousarray(block_names, dtype=np.int64),
                np.ascontiguousarray(block_groups, dtype=np.int64), *_csr(indexing), *_csr(parents), *_csr(children)]
        self.nb = int(arrs[2].size)
        self._keep = arrs
        ip = lambda a: a.ctypes.data_as(_ip)   # noqa: E731
        self.h = self.lib.refcpu_create(self.n, self.q, self.p, self.nb, int(arrs[1].size), self._d(self.y), self._d(self.X),
                                        self._d(self.coords), *[ip(a) for a in arrs], int(reference_distance),
                                        int(bool(reference_quirks)) | (2 if limited_tree else 0), int(threads))   # bit 1: limited_tree
        self.idx_ptr, self.par_ptr = arrs[4], arrs[6]
        self.isref = None

    @staticmethod
    def _d(a):
        return a.ctypes.data_as(_dp)

    def close(self):
        if self.h:
            self.lib.refcpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_threads(self, threads):
        self.lib.refcpu_set_threads(int(threads))

    def factor(self, slot, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        ll = C.c_double()
        rc = self.lib.refcpu_factor(self.h, slot, self._d(th), th.size, C.byref(ll))
        return rc, ll.value

    def sample_w(self, z):
        z = np.ascontiguousarray(z, dtype=np.float64)
        return self.lib.refcpu_sample_w(self.h, self._d(z))

    def loglik_w(self, slot):
        return self.lib.refcpu_loglik_w(self.h, slot)

    def swap(self):
        self.lib.refcpu_swap(self.h)

    def set_w(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        self.lib.refcpu_set_w(self.h, self._d(w))

    def get_w(self):
        w = np.zeros(self.n)
        self.lib.refcpu_get_w(self.h, self._d(w))
        return w

    def set_tausq_inv(self, t):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.q,)))
        self.lib.refcpu_set_tausq_inv(self.h, self._d(t))

    def set_beta(self, B):
        B = np.asfortranarray(np.asarray(B, dtype=np.float64).reshape(self.p, self.q))
        self.lib.refcpu_set_beta(self.h, self._d(B))

    def stats(self):
        xty = np.zeros(self.p * self.q)
        ssq = np.zeros(self.q)
        self.lib.refcpu_stats(self.h, self._d(xty), self._d(ssq))
        return xty.reshape(self.q, self.p).T.copy(), ssq

    def comps(self, slot):
        a, b = np.zeros(self.nb), np.zeros(self.nb)
        self.lib.refcpu_get_comps(self.h, slot, self._d(a), self._d(b))
        return a, b
