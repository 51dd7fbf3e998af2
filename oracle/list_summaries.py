"""ORACLE -- test infrastructure only.  Restatement of the reference's after-the-fact posterior summaries over the `keep`
stored draws (/root/reference/src/list_mean.cpp): list_mean (:10-30) and list_qtile / cqtile / prctile_stl (:62-137).
Parity unpinned (the reference ships no fixtures); checked in tests against numpy order statistics."""
import math

import numpy as np


def list_mean(x):
    """list_mean.cpp:10-30: elementwise mean over the list of equally sized matrices."""
    return np.mean(np.stack([np.asarray(a, dtype=np.float64) for a in x], axis=0), axis=0)


def prctile_stl(values, percent):
    """list_mean.cpp:62-106 (the value it returns in range[1]): order statistics around r = percent / 100 * len picked with
    nth_element / min_element / max_element, then MATLAB-prctile-like linear interpolation."""
    a = np.sort(np.asarray(values, dtype=np.float64))
    n = a.size
    r = (percent / 100.0) * n
    if r >= n / 2.0:
        lo = int(max(r - 1.0, 0.0))
        lower = a[lo]
        upper = a[lo + 1] if lo < n - 1 else lower
    else:
        up = int(math.ceil(max(r - 1.0, 0.0)))
        upper = a[up]
        lower = a[up - 1] if up > 0 else upper
    k = int(r + 0.5)
    r = r - k
    return (0.5 - r) * lower + (0.5 + r) * upper


def list_qtile(x, q):
    """list_mean.cpp:116-137: elementwise cqtile(slices, q) = prctile_stl(slices, q * 100)."""
    st = np.stack([np.asarray(a, dtype=np.float64).reshape(-1) for a in x], axis=0)
    out = np.array([prctile_stl(st[:, j], q * 100.0) for j in range(st.shape[1])])
    return out.reshape(np.asarray(x[0]).shape)
