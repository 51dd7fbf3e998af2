"""Synthetic workloads of BASELINE.md section 3 / SURVEY.md section 8d (grids on [0,1]^2, seed 2021)."""
import numpy as np

from .topology import grid_coords, prepare


def theta_layout(q):
    n_cbase = 3 if q > 2 else 1
    return 3 * q + n_cbase, n_cbase, q * (q - 1) // 2


def default_bounds(q, btmlim=1e-3, toplim=1e3):
    """set_unif_bounds as built in /root/reference/R/spamtree_fit.R:105-133."""
    npars, n_cbase, k = theta_layout(q)
    b = np.zeros((npars, 2))
    b[:, 0], b[:, 1] = btmlim, toplim
    if q > 1:
        b[1:q, 0] = -toplim
    if n_cbase == 3:
        b[npars - 2, :] = (btmlim, 1 - btmlim)
    if q > 1:
        vb = np.zeros((k, 2))
        vb[:, 0], vb[:, 1] = btmlim, toplim - btmlim
        b = np.vstack([b, vb])
    return b


def true_theta(q):
    """q=1: sigma^2 = 2.3, phi = 6 (README.md:39-42 of the reference); q>1: a well-conditioned AG10 setting."""
    if q == 1:
        return np.array([2.3, 1.0, 1.0, 6.0])
    if q == 2:
        return np.array([1.0, 1.5, 0.3, 0.51, 3.0, 4.0, 5.0, 1.0])
    ai1 = np.array(([1.0, -0.8, 1.3] + [1.0] * q)[:q])
    k = q * (q - 1) // 2
    return np.concatenate([ai1, np.linspace(0.3, 0.6, q), np.linspace(3.0, 5.0, q), [1.2, 0.7, 4.0],
                           np.linspace(0.5, 1.5, k)])


def make_workload(side, q=1, p=3, seed=2021, missing=None, cell_size=25, device=None, limited_tree=False):
    """Grid workload: X ~ N(0,1), beta = (-1, .5, 1), tausq = .1, a smooth synthetic latent field of variance ~2.3.

    missing: None or per-outcome drop probabilities (config #5 uses (0.1, 0.3, 0.5)).
    Returns a dict with everything `spamtree_mv_mcmc` takes, rows in the sorted order.
    """
    rng = np.random.default_rng(seed)
    coords, mv_id = grid_coords(side, q)
    n = coords.shape[0]
    X = rng.standard_normal((n, p))
    beta = np.array([-1.0, 0.5, 1.0, 0.25, -0.3][:p])
    f = np.zeros(n)
    for _ in range(8):
        kx, ky, ph = rng.uniform(2, 20), rng.uniform(2, 20), rng.uniform(0, 6.28)
        f += np.sin(kx * coords[:, 0] + ky * coords[:, 1] + ph + 0.9 * mv_id)
    f *= np.sqrt(2.3 / max(f.var(), 1e-12))
    y = X @ beta + f + np.sqrt(0.1) * rng.standard_normal(n)
    if missing is not None:
        pr = np.asarray(missing, dtype=np.float64)[mv_id - 1]
        y = np.where(rng.uniform(size=n) < pr, np.nan, y)
    topo = prepare(y, coords, mv_id, cell_size=cell_size, device=device, limited_tree=limited_tree)   # device: the row-parallel tree-building steps on the GPU
    s = topo.sort_ix
    Z = np.zeros((n, q))
    Z[np.arange(n), topo.mv_id - 1] = 1.0
    return dict(topo=topo, y=y[s], X=X[s], Z=Z, coords=topo.coords, mv_id=topo.mv_id, blocking=topo.blocking,
                gix_block=topo.gix_block, res_is_ref=topo.res_is_ref,
                parents=(topo.parents_ptr, topo.parents_idx), children=(topo.children_ptr, topo.children_idx),
                block_names=topo.block_names, block_groups=topo.block_groups,
                indexing=(topo.indexing_ptr, topo.indexing_idx), n=n, q=q, p=p,
                bounds=default_bounds(q), theta=true_theta(q), beta_true=beta)
