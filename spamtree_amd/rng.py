"""Host-side draws of the MCMC driver: Philox4x32-10 counter streams (the R generator the reference uses through
Rcpp -- arma::randn, R::runif, R::rgamma; /root/reference/src/spamtree_fit.cpp:211, mh_adapt.h:30,
spamtree_model.cpp:1378, 1405 -- is not available outside R, so the stream contract is this build's own).

counter = (index_lo, index_hi | outcome, iteration, stream), key = seed.  Streams: 0 sweep normals (generated on
the device, same contract), 1 theta proposal, 2 MH uniform, 3 gamma, 4 beta normals, 5 yhat noise (device).
"""
import math

import numpy as np

_MASK = np.uint64(0xFFFFFFFF)


def _philox(c, key):
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & _MASK for x in c)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = key
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)), p1 & _MASK, \
            ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)), p0 & _MASK
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _u01(a, b):
    return (((a >> np.uint64(5)) * np.uint64(1 << 26) + (b >> np.uint64(6))).astype(np.float64) + 0.5) * 2.0 ** -53


class HostRng:
    def __init__(self, seed):
        self.seed = int(seed)
        self.key = (self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF)

    def _normal(self, idx, hi, it, stream):
        x = _philox((idx, hi, it, stream), self.key)
        return np.sqrt(-2.0 * np.log(_u01(x[0], x[1]))) * np.cos(2.0 * math.pi * _u01(x[2], x[3]))

    def _uniform(self, idx, hi, it, stream):
        x = _philox((idx, hi, it, stream), self.key)
        return _u01(x[0], x[1])

    def theta_normals(self, it, k):
        return self._normal(np.arange(k), 0, it, 1)

    def mh_uniform(self, it):
        return float(self._uniform(0, 0, it, 2))

    def gamma(self, it, j, shape, scale):
        """Marsaglia & Tsang (2000), shape >= 1."""
        d = shape - 1.0 / 3.0
        c = 1.0 / math.sqrt(9.0 * d)
        t = 0
        while True:
            x = float(self._normal(2 * t, j, it, 3))
            u = float(self._uniform(2 * t + 1, j, it, 3))
            t += 1
            v = 1.0 + c * x
            if v <= 0.0:
                continue
            v = v ** 3
            if math.log(u) < 0.5 * x * x + d - d * v + d * math.log(v):
                return d * v * scale

    def beta_normals(self, it, j, p):
        return self._normal(np.arange(p), j, it, 4)
