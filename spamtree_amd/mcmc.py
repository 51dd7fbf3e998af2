"""Host MCMC driver over the HIP model: mirrors `spamtree_mv_mcmc` (/root/reference/src/spamtree_fit.cpp:5-430) and
the adaptive Metropolis helpers of /root/reference/src/mh_adapt.{h,cpp}; same argument list, same returned names.
The per-iteration order of operations and of random draws is the reference's."""
import math
import time

import numpy as np

from .model import SpamTreeMV
from .rng import HostRng


def _logit(x, l, u):
    return -math.log((u - l) / (x - l) - 1.0)


def _logistic(x, l, u):
    return l + (u - l) / (1.0 + math.exp(-x))


def par_huvtransf_fwd(par, bounds):                       # mh_adapt.cpp:3-8
    return np.array([_logit(par[j], bounds[j, 0], bounds[j, 1]) for j in range(len(par))])


def par_huvtransf_back(par, bounds):                      # mh_adapt.cpp:10-15
    return np.array([_logistic(par[j], bounds[j, 0], bounds[j, 1]) for j in range(len(par))])


def unif_bounds(par, bounds):                             # mh_adapt.h:188-202
    out = False
    for i in range(par.size):
        if par[i] < bounds[i, 0]:
            out, par[i] = True, bounds[i, 0] + 1e-10
        if par[i] > bounds[i, 1]:
            out, par[i] = True, bounds[i, 1] - 1e-10
    return out


def calc_jacobian(new_param, param, bounds):              # mh_adapt.h:210-239
    jac = 0.0
    for j in range(param.size):
        l, u = bounds[j]
        jac += (-math.log(u - param[j]) - math.log(param[j] - l)) - \
               (-math.log(u - new_param[j]) - math.log(new_param[j] - l))
    return jac


def do_I_accept(logaccept, u):                            # mh_adapt.h:20-36
    if not math.isfinite(logaccept):
        acceptj = 0.0
    else:
        acceptj = math.exp(logaccept) if logaccept < 0 else 1.0
    return u < acceptj


class RAMAdapt:
    """Robust adaptive Metropolis (Vihola 2012) as configured in mh_adapt.h:78-135 (member g0 = 50)."""

    def __init__(self, npars, metropolis_sd):
        self.p = npars
        self.alpha_star, self.gamma, self.g0 = 0.234, 0.5 + 1e-6, 50
        self.S = np.array(metropolis_sd, dtype=np.float64)
        self.paramsd = np.linalg.cholesky(self.S)
        self.prodparam = self.paramsd / (self.g0 + 1.0)
        self.started = False
        self.propos_count = self.accept_count = self.accept_ratio = 0.0
        self.history = np.zeros(200)
        self.c = 0
        self.flag_accepted = False

    def count_proposal(self):
        self.propos_count += 1
        self.c += 1
        self.flag_accepted = False

    def count_accepted(self):
        self.accept_count += 1
        self.history[self.c % 200] = 1
        self.flag_accepted = True

    def update_ratios(self):
        self.accept_ratio = self.accept_count / self.propos_count
        if not self.flag_accepted:
            self.history[self.c % 200] = 0

    def adapt(self, U, alpha, mc):
        if mc < self.g0:
            self.prodparam = self.prodparam + np.outer(U, U) / (mc + 1.0)
            return
        if not self.started:
            self.paramsd, self.started = self.prodparam, True
        eta = min(1.0, self.p * (mc - self.g0 + 1.0) ** (-self.gamma))
        alpha = 1.0 if math.isnan(alpha) else min(1.0, alpha)
        Sigma = np.eye(self.p) + eta * (alpha - self.alpha_star) * np.outer(U, U) / float(U @ U)
        self.S = self.paramsd @ Sigma @ self.paramsd.T
        self.paramsd = np.linalg.cholesky(self.S)


class Chain:
    """The state `spamtree_mv_mcmc` keeps across iterations, so a caller (bench.py) can step it."""

    def __init__(self, mtree: SpamTreeMV, bounds, mcmcsd, seed=2021, adapting=True, sample_beta=True,
                 sample_tausq=True, sample_theta=True, sample_w=True):
        self.mtree = mtree
        self.bounds = np.asarray(bounds, dtype=np.float64)
        self.rng = HostRng(seed)
        self.seed = int(seed)
        self.adapting = adapting
        self.flags = (sample_beta, sample_tausq, sample_theta, sample_w)
        if not mtree.get_loglik_comps_w(0):
            raise RuntimeError("starting theta is not positive definite")
        mtree.theta_update(1, mtree.theta[0])
        mtree.get_loglik_comps_w(1)                       # spamtree_fit.cpp:110-111
        self.param = mtree.theta[0].copy()
        self.current_loglik = mtree.loglik_w[0]
        self.adaptivemc = RAMAdapt(self.param.size, mcmcsd)
        self.m = 0
        self.last = {}

    def step(self):
        self.step_w_theta()
        self.step_tausq_beta()

    def step_w_theta(self):
        """spamtree_fit.cpp:183-289: w sweep, its log-density, Metropolis step for theta."""
        mt, m = self.mtree, self.m
        sample_beta, sample_tausq, sample_theta, sample_w = self.flags
        if sample_w:
            mt.deal_with_w(None, seed=self.seed, it=m)    # spamtree_fit.cpp:183-187
            self.current_loglik = mt.get_loglik_w(0)
        if sample_theta:
            am = self.adaptivemc
            am.count_proposal()
            U = self.rng.theta_normals(m, self.param.size)
            new_param = par_huvtransf_back(par_huvtransf_fwd(self.param, self.bounds) + am.paramsd @ U, self.bounds)
            unif_bounds(new_param, self.bounds)
            mt.theta_update(1, new_param)
            acceptable = mt.get_loglik_comps_w(1)
            new_loglik = mt.loglik_w[1]
            self.current_loglik = mt.loglik_w[0]
            if math.isnan(self.current_loglik):
                raise FloatingPointError("At nan loglik: error.")
            logaccept = new_loglik - self.current_loglik + calc_jacobian(new_param, self.param, self.bounds)
            accepted = do_I_accept(logaccept, self.rng.mh_uniform(m)) and acceptable
            if accepted:
                am.count_accepted()
                self.current_loglik = new_loglik
                mt.accept_make_change()
                self.param = new_param
            am.update_ratios()
            if self.adapting:
                with np.errstate(all="ignore"):
                    alpha = float(np.float64(1.0 if acceptable else 0.0) * np.exp(np.float64(logaccept)))
                am.adapt(U, alpha, m)
            self.last = dict(accepted=accepted, acceptable=acceptable, logaccept=logaccept)

    def step_tausq_beta(self):
        """spamtree_fit.cpp:308-330, then the iteration counter advances."""
        mt, m = self.mtree, self.m
        sample_beta, sample_tausq, sample_theta, sample_w = self.flags
        if sample_tausq:
            mt.gibbs_sample_tausq(lambda j, a, b: self.rng.gamma(m, j, a, b))
        if sample_beta:
            mt.gibbs_sample_beta([self.rng.beta_normals(m, j, mt.p) for j in range(mt.q)])
        self.m += 1


def spamtree_mv_mcmc(y, X, Z, coords, mv_id, blocking, gix_block, res_is_ref, parents, children, limited_tree,
                     layer_names, layer_gibbs_group, indexing, set_unif_bounds_in, start_w, theta, beta, tausq,
                     mcmcsd, mcmc_keep=100, mcmc_burn=100, mcmc_thin=1, num_threads=1, use_alg="S",
                     adapting=False, main_verbose=True, verbose=False, debug=False, printall=False,
                     sample_beta=True, sample_tausq=True, sample_theta=True, sample_w=True, sample_predicts=True,
                     seed=2021, device=0, reference_quirks=True):
    """spamtree_fit.cpp:5-430.  `num_threads` is accepted and ignored (the parallelism is the GPU's)."""
    n_all = np.asarray(coords).shape[0]
    q = np.asarray(Z).shape[1]
    mtree = SpamTreeMV(y, X, Z, coords, mv_id, blocking, gix_block, res_is_ref, parents, children, limited_tree,
                       layer_names, layer_gibbs_group, indexing, np.zeros(n_all), beta, theta, 1.0 / tausq,
                       device=device, reference_quirks=reference_quirks)      # start_w ignored (:95)
    chain = Chain(mtree, set_unif_bounds_in, mcmcsd, seed, adapting, sample_beta, sample_tausq, sample_theta, sample_w)
    p = mtree.p
    beta_mcmc = np.zeros((p, mcmc_keep, q))
    tausq_mcmc = np.zeros((q, mcmc_keep))
    theta_mcmc = np.zeros((chain.param.size, mcmc_keep))
    w_mcmc, yhat_mcmc = [None] * mcmc_keep, [None] * mcmc_keep
    mcmc = mcmc_thin * mcmc_keep + mcmc_burn
    predict_param = chain.param.copy()
    msaved = 0
    t0 = time.time()
    try:
        for m in range(mcmc):
            mx = m - mcmc_burn
            saving = mx >= 0 and mx % mcmc_thin == 0
            chain.step_w_theta()
            if saving and sample_predicts and sample_w:                       # :300-306
                need_update = bool(np.sum(np.abs(chain.param - predict_param) > 1e-05))
                mtree.predict(need_update)
                predict_param = chain.param.copy()
            chain.step_tausq_beta()
            if saving:
                tausq_mcmc[:, msaved] = 1.0 / mtree.tausq_inv
                beta_mcmc[:, msaved, :] = mtree.Bcoeff
                theta_mcmc[:, msaved] = mtree.theta[0]
                w_mcmc[msaved] = mtree.get_w().reshape(-1, 1)
                yhat_mcmc[msaved] = mtree.yhat(None, seed=seed, it=m).reshape(-1, 1)
                msaved += 1
    except Exception as exc:                                      # spamtree_fit.cpp:416-428
        if main_verbose:
            print(exc)
            print("MCMC has been interrupted.")
        mtree.close()
        return {"None": np.zeros(0)}
    mcmc_time = time.time() - t0
    out = dict(w_mcmc=w_mcmc, yhat_mcmc=yhat_mcmc, beta_mcmc=beta_mcmc, tausq_mcmc=tausq_mcmc,
               theta_mcmc=theta_mcmc, paramsd=chain.adaptivemc.paramsd, mcmc_time=mcmc_time)
    mtree.close()
    return out
