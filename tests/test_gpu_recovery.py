"""GPU: an end-to-end statistical check that does not involve the oracle at all -- data simulated from an exact Gaussian
process (dense Cholesky of the exponential covariance), fitted with the C++ driver on the GPU: the chain must recover the
regression coefficients, the noise variance and the latent field.  (Not a parity test: it says the sampler targets a sensible
posterior, which no agreement between two implementations of the same formulas can say.)"""
import numpy as np
import pytest

from tests.util import default_bounds

pytestmark = pytest.mark.gpu


def test_chain_recovers_beta_tausq_and_the_latent_field():
    from spamtree_amd import fit
    from spamtree_amd.topology import grid_coords, prepare
    rng = np.random.default_rng(42)
    side = 30
    coords, mv = grid_coords(side, 1)
    n = coords.shape[0]
    sigmasq, phi, tausq = 2.0, 5.0, 0.1
    beta_true = np.array([-1.0, 0.5, 1.0])
    d = np.sqrt(((coords[:, None, :] - coords[None, :, :]) ** 2).sum(-1))
    w_true = np.linalg.cholesky(sigmasq * np.exp(-phi * d) + 1e-10 * np.eye(n)) @ rng.standard_normal(n)
    X = rng.standard_normal((n, 3))
    y = X @ beta_true + w_true + np.sqrt(tausq) * rng.standard_normal(n)
    held = rng.uniform(size=n) < 0.1
    y_fit = y.copy()
    y_fit[held] = np.nan
    topo = prepare(y_fit, coords, mv)
    s = topo.sort_ix
    Z = np.ones((n, 1))
    lists = lambda ptr, idx: [idx[ptr[i]:ptr[i + 1]] for i in range(ptr.size - 1)]
    keep = 300
    out = fit.spamtree_mv_mcmc(y_fit[s], X[s], Z, topo.coords, topo.mv_id, topo.blocking, topo.gix_block, topo.res_is_ref,
                               lists(topo.parents_ptr, topo.parents_idx), lists(topo.children_ptr, topo.children_idx), False,
                               topo.block_names, topo.block_groups, lists(topo.indexing_ptr, topo.indexing_idx), default_bounds(1),
                               np.zeros((n, 1)), np.array([1.0, 1.0, 1.0, 3.0]), np.zeros(3), 0.5, 0.05 * np.eye(4),
                               mcmc_keep=keep, mcmc_burn=700, mcmc_thin=1, adapting=True, main_verbose=False, seed=7)
    assert "None" not in out
    beta_hat = out["beta_mcmc"][:, :, 0].mean(axis=1)
    beta_sd = out["beta_mcmc"][:, :, 0].std(axis=1)
    assert np.all(np.abs(beta_hat - beta_true) < np.maximum(4 * beta_sd, 0.12)), (beta_hat, beta_sd)
    tau_hat = out["tausq_mcmc"][0].mean()
    assert 0.04 < tau_hat < 0.25, tau_hat
    w_hat = np.mean([np.asarray(w).reshape(-1) for w in out["w_mcmc"]], axis=0)
    wt = w_true[s]
    obs = ~held[s]
    assert np.corrcoef(w_hat[obs], wt[obs])[0, 1] > 0.93
    # held-out locations: the predicted latent field beats the trivial predictor by a wide margin
    assert np.corrcoef(w_hat[~obs], wt[~obs])[0, 1] > 0.8
    yhat = np.mean([np.asarray(v).reshape(-1) for v in out["yhat_mcmc"]], axis=0)
    rmse = np.sqrt(np.mean((yhat[~obs] - y[s][~obs]) ** 2))
    assert rmse < 0.75 * np.std(y[s][~obs]), rmse
    # the spatial variance-range product sigma^2 * phi is what the data identify: within a factor 2.5 of the truth
    prod = (out["theta_mcmc"][0] * out["theta_mcmc"][3]).mean()
    assert sigmasq * phi / 2.5 < prod < sigmasq * phi * 2.5, prod
