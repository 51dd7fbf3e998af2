"""`CrossCovarianceAG10` -- the reference's other public export (/root/reference/NAMESPACE:14,
/root/reference/src/covariance_functions.cpp:301-355, man/CrossCovarianceAG10.Rd) on the GPU."""
import numpy as np

from . import _lib
from .model import SpamTreeError, _dp, _f64, _i64, _ip


def CrossCovarianceAG10(coords1, mv1, coords2, mv2, ai1, ai2, phi_i, thetamv, Dmat, device=0):
    lib = _lib.load()
    c1 = np.asfortranarray(np.asarray(coords1, dtype=np.float64))
    c2 = np.asfortranarray(np.asarray(coords2, dtype=np.float64))
    D = np.asfortranarray(np.atleast_2d(np.asarray(Dmat, dtype=np.float64)))
    q = D.shape[1]
    n1, n2 = c1.shape[0], c2.shape[0]
    out = np.zeros((n1, n2), order="F")
    m1, m2 = _i64(mv1), _i64(mv2)
    a1, a2, ph, tm = _f64(ai1), _f64(ai2), _f64(phi_i), _f64(np.atleast_1d(thetamv))
    rc = lib.st_cross_covariance_ag10(_dp(c1), _ip(m1), n1, _dp(c2), _ip(m2), n2, _dp(a1), _dp(a2), _dp(ph), _dp(tm), _dp(D), q,
                                      int(device), _dp(out))
    if rc != 0:
        raise SpamTreeError(lib.st_last_error(None).decode() or f"st_cross_covariance_ag10 failed ({rc})")
    return out
