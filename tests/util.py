"""Shared synthetic-problem builder for the tests (inputs only; no oracle or product logic here)."""
import numpy as np

from spamtree_amd.topology import grid_coords, prepare


def csr_to_lists(ptr, idx):
    return [idx[ptr[i]:ptr[i + 1]].copy() for i in range(ptr.size - 1)]


def theta_layout(q):
    n_cbase = 3 if q > 2 else 1
    npars = 3 * q + n_cbase
    k = q * (q - 1) // 2
    return npars, n_cbase, k


def default_bounds(q, btmlim=1e-3, toplim=1e3):
    """set_unif_bounds of R/spamtree_fit.R:105-133."""
    npars, n_cbase, k = theta_layout(q)
    b = np.zeros((npars, 2))
    b[:, 0] = btmlim
    b[:, 1] = toplim
    if q > 1:
        b[1:q, 0] = -toplim
    if n_cbase == 3:
        b[npars - 2, :] = (btmlim, 1 - btmlim)
    if q > 1:
        vb = np.zeros((k, 2))
        vb[:, 0] = btmlim
        vb[:, 1] = toplim - btmlim
        b = np.vstack([b, vb])
    return b


def nice_theta(q):
    """A well-conditioned covariance parameter vector for tests (q=1: sigma^2=2.3, phi=6 as README.md:39-42)."""
    if q == 1:
        return np.array([2.3, 1.0, 1.0, 6.0])
    if q == 2:
        return np.array([1.0, 1.5, 0.3, 0.51, 3.0, 4.0, 5.0, 1.0])
    ai1 = np.array([1.0, -0.8, 1.3][:q] + [1.0] * max(0, q - 3))
    ai2 = np.linspace(0.3, 0.6, q)
    phi = np.linspace(3.0, 5.0, q)
    thetamv = np.array([1.2, 0.7, 4.0])
    k = q * (q - 1) // 2
    dvec = np.linspace(0.5, 1.5, k)
    return np.concatenate([ai1, ai2, phi, thetamv, dvec])


def make_problem(side=25, q=1, seed=0, missing=0.0, coords=None, mv_id=None, p=3, random_coords=False, **tree_kw):
    """Synthetic inputs in the layout spamtree_mv_mcmc receives (R/spamtree_fit.R:327-362)."""
    rng = np.random.default_rng(seed)
    if coords is None:
        if random_coords:
            n0 = side * side
            base = rng.uniform(size=(n0, 2))
            coords = np.tile(base, (q, 1))
            mv_id = np.repeat(np.arange(1, q + 1), n0)
        else:
            coords, mv_id = grid_coords(side, q)
    n = coords.shape[0]
    X = rng.standard_normal((n, p))
    beta = np.array([-1.0, 0.5, 1.0, 0.25, -0.3][:p])
    f = np.zeros(n)
    for _ in range(6):
        kx, ky, ph = rng.uniform(1, 6), rng.uniform(1, 6), rng.uniform(0, 6.28)
        f += rng.normal() * np.sin(kx * coords[:, 0] + ky * coords[:, 1] + ph + 0.7 * mv_id)
    y = X @ beta + f + np.sqrt(0.1) * rng.standard_normal(n)
    if np.ndim(missing) > 0:      # per-outcome drop probabilities (config #5: 0.1, 0.3, 0.5 -- imbalanced)
        y = y.copy()
        y[rng.uniform(size=n) < np.asarray(missing, dtype=np.float64)[np.asarray(mv_id) - 1]] = np.nan
    elif missing > 0:
        y = y.copy()
        y[rng.uniform(size=n) < missing] = np.nan
    limited_tree = bool(tree_kw.get("limited_tree", False))
    topo = prepare(y, coords, mv_id, **tree_kw)
    s = topo.sort_ix
    Z = np.zeros((n, q))
    Z[np.arange(n), topo.mv_id - 1] = 1.0
    return dict(
        topo=topo, y=y[s], X=X[s], Z=Z, coords=topo.coords, mv_id=topo.mv_id, blocking=topo.blocking,
        gix_block=topo.gix_block, res_is_ref=topo.res_is_ref,
        parents=csr_to_lists(topo.parents_ptr, topo.parents_idx),
        children=csr_to_lists(topo.children_ptr, topo.children_idx),
        block_names=topo.block_names, block_groups=topo.block_groups,
        indexing=csr_to_lists(topo.indexing_ptr, topo.indexing_idx),
        q=q, p=p, n=n, beta_true=beta, bounds=default_bounds(q), theta=nice_theta(q), limited_tree=limited_tree)


def oracle_model(pb, theta=None, beta=None, tausq=0.1, w=None, **kw):
    from oracle.spamtree_oracle import SpamTreeMV
    theta = pb["theta"] if theta is None else theta
    beta = np.zeros(pb["p"]) if beta is None else beta
    w = np.zeros(pb["n"]) if w is None else w
    return SpamTreeMV(pb["y"], pb["X"], pb["Z"], pb["coords"], pb["mv_id"], pb["blocking"], pb["gix_block"],
                      pb["res_is_ref"], pb["parents"], pb["children"], pb.get("limited_tree", False), pb["block_names"],
                      pb["block_groups"], pb["indexing"], w, beta, theta, 1.0 / tausq, **kw)


def strip_coords(nx, ny, q, width=0.02):
    """An nx x ny grid on the thin strip [0,1] x [0,width], replicated per outcome: with K = (2, 1) the tree splits one axis
    only, so it gets DEEP (long ancestor chains) with few rows -- chains of config #4 / #5 length at oracle-friendly sizes."""
    xs = np.linspace(0.0, 1.0, nx)
    ys = np.linspace(0.0, width, ny)
    g = np.stack(np.meshgrid(xs, ys, indexing="ij"), axis=-1).reshape(-1, 2)
    return np.tile(g, (q, 1)), np.repeat(np.arange(1, q + 1), nx * ny)
