// spamtree_fit.cpp -- C++ host MCMC driver above the C-ABI: the counterpart of the reference's Rcpp-exported
// `spamtree_mv_mcmc` (/root/reference/src/spamtree_fit.cpp:5-430) and of the adaptive-Metropolis helpers in
// /root/reference/src/mh_adapt.h:20-239 and mh_adapt.cpp:3-15.  Everything here is host code (no kernels): the RNG,
// the Metropolis step, the conjugate draws of tausq and beta, the bookkeeping of saved iterations.  Each hot step is
// one call into include/spamtree_hip.h.
//
// R's generator (arma::randn / R::runif / R::rgamma through Rcpp) is not available outside R; the draws come from
// Philox4x32-10 counter streams with the contract documented in include/spamtree_fit.h (same per-iteration ORDER of
// draws as the reference).
#include "spamtree_fit.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// ---- Philox4x32-10 (Salmon et al. 2011) ------------------------------------------------------------------------
struct HostRng {
  uint32_t k0, k1;
  explicit HostRng(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  void block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) const {
    uint32_t a = k0, b = k1;
    for (int r = 0; r < 10; ++r) {
      const uint64_t p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b;
      c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
  static double u01(uint32_t a, uint32_t b) {
    return ((double)(((uint64_t)(a >> 5) << 26) + (uint64_t)(b >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
  }
  double normal(uint64_t idx, uint32_t hi, uint32_t it, uint32_t stream) const {
    uint32_t o[4];
    block((uint32_t)idx, hi, it, stream, o);
    return std::sqrt(-2.0 * std::log(u01(o[0], o[1]))) * std::cos(6.283185307179586476925286766559 * u01(o[2], o[3]));
  }
  double uniform(uint64_t idx, uint32_t hi, uint32_t it, uint32_t stream) const {
    uint32_t o[4];
    block((uint32_t)idx, hi, it, stream, o);
    return u01(o[0], o[1]);
  }
  // Marsaglia & Tsang (2000), shape >= 1; attempt t uses counters 2t (normal) and 2t+1 (uniform) of stream 3
  double gamma(uint32_t it, uint32_t j, double shape, double scale) const {
    const double d = shape - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
    for (uint64_t t = 0;; ++t) {
      const double x = normal(2 * t, j, it, 3), u = uniform(2 * t + 1, j, it, 3);
      double v = 1.0 + c * x;
      if (v <= 0.0) continue;
      v = v * v * v;
      if (std::log(u) < 0.5 * x * x + d - d * v + d * std::log(v)) return d * v * scale;
    }
  }
};

// ---- small dense helpers (column-major k x k, k <= 39) -----------------------------------------------------------
typedef std::vector<double> Mat;
bool chol_lower(const Mat &A, int n, Mat &L) {
  L.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)k * n + j] * L[(size_t)k * n + j];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    L[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)j * n + i];
      for (int k = 0; k < j; ++k) s -= L[(size_t)k * n + i] * L[(size_t)k * n + j];
      L[(size_t)j * n + i] = s / d;
    }
  }
  return true;
}
Mat inv_lower(const Mat &L, int n) {
  Mat X((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = j; k < i; ++k) s -= L[(size_t)k * n + i] * X[(size_t)j * n + k];
      X[(size_t)j * n + i] = s / L[(size_t)i * n + i];
    }
  return X;
}

inline double logit(double x, double l, double u) { return -std::log((u - l) / (x - l) - 1.0); }
inline double logistic(double x, double l, double u) { return l + (u - l) / (1.0 + std::exp(-x)); }

// ---- RAMAdapt (mh_adapt.h:40-135; the member g0 = 50 shadows the file-level 500) -------------------------------
struct RAMAdapt {
  int p = 0, g0 = 50, c = 0;
  double alpha_star = 0.234, gamma = 0.5 + 1e-6;
  Mat S, paramsd, prodparam;
  bool started = false, flag_accepted = false;
  double propos_count = 0, accept_count = 0, accept_ratio = 0;
  std::vector<double> history = std::vector<double>(200, 0.0);
  bool init(int npars, const double *sd) {
    p = npars;
    S.assign(sd, sd + (size_t)p * p);
    if (!chol_lower(S, p, paramsd)) return false;
    prodparam = paramsd;
    for (auto &v : prodparam) v /= (g0 + 1.0);
    return true;
  }
  void count_proposal() { propos_count += 1; c += 1; flag_accepted = false; }
  void count_accepted() { accept_count += 1; history[c % 200] = 1; flag_accepted = true; }
  void update_ratios() { accept_ratio = accept_count / propos_count; if (!flag_accepted) history[c % 200] = 0; }
  void adapt(const std::vector<double> &U, double alpha, int mc) {
    if (mc < g0) {
      for (int j = 0; j < p; ++j)
        for (int i = 0; i < p; ++i) prodparam[(size_t)j * p + i] += U[i] * U[j] / (mc + 1.0);
      return;
    }
    if (!started) { paramsd = prodparam; started = true; }
    const double eta = std::min(1.0, p * std::pow(mc - g0 + 1.0, -gamma));
    alpha = std::isnan(alpha) ? 1.0 : std::min(1.0, alpha);   // std::min(1.0, NaN) == 1.0 in the reference
    double uu = 0;
    for (int i = 0; i < p; ++i) uu += U[i] * U[i];
    Mat Sigma((size_t)p * p, 0.0), T((size_t)p * p, 0.0);
    for (int j = 0; j < p; ++j)
      for (int i = 0; i < p; ++i) Sigma[(size_t)j * p + i] = (i == j ? 1.0 : 0.0) + eta * (alpha - alpha_star) * U[i] * U[j] / uu;
    for (int j = 0; j < p; ++j)          // T = paramsd * Sigma
      for (int i = 0; i < p; ++i) {
        double s = 0;
        for (int k = 0; k < p; ++k) s += paramsd[(size_t)k * p + i] * Sigma[(size_t)j * p + k];
        T[(size_t)j * p + i] = s;
      }
    for (int j = 0; j < p; ++j)          // S = T * paramsd'
      for (int i = 0; i < p; ++i) {
        double s = 0;
        for (int k = 0; k < p; ++k) s += T[(size_t)k * p + i] * paramsd[(size_t)k * p + j];
        S[(size_t)j * p + i] = s;
      }
    Mat L;
    if (chol_lower(S, p, L)) paramsd = L;   // arma::chol throws otherwise; the reference would abort the chain
  }
};

}  // namespace

struct stm_chain_s {
  st_handle h = nullptr;
  std::string err;
  int q = 1, p = 1, k = 0;
  long long n_all = 0;
  uint64_t seed = 0;
  HostRng rng{0};
  RAMAdapt am;
  std::vector<double> bounds;         // k x 2 column-major
  std::vector<double> param, theta_alt, Bcoeff, tausq_inv, xtx, Vi, Vim;
  std::vector<long long> n_obs_q;
  double loglik[2] = {0, 0}, current_loglik = 0;
  int adapting = 1, sample_beta = 1, sample_tausq = 1, sample_theta = 1, sample_w = 1;
  long long m = 0;
  int last_accepted = 0, last_acceptable = 1;
  bool tb_drawn = false;   // this iteration's tausq / beta were drawn under the proposal's factorisation already
  double last_logaccept = 0;
  bool initialised = false;
};

extern "C" const char *stm_last_error(stm_chain c) { return c ? c->err.c_str() : "null chain"; }

extern "C" int stm_destroy(stm_chain c) {
  if (!c) return 0;
  if (c->h) st_destroy(c->h);
  delete c;
  return 0;
}
extern "C" st_handle stm_handle(stm_chain c) { return c ? c->h : nullptr; }

// SpamTreeMV construction + the two initial factorisations of spamtree_fit.cpp:93-120, RAMAdapt of :151
extern "C" int stm_create(const st_problem *pb, const st_options *opt, const double *set_unif_bounds, const double *mcmcsd,
                          const double *theta, int ntheta, const double *beta, double tausq, uint64_t seed, const stm_flags *flags,
                          stm_chain *out) {
  if (!pb || !out || !theta || !beta || !set_unif_bounds || !mcmcsd) return ST_ERR_USAGE;
  *out = nullptr;
  stm_chain c = new stm_chain_s();
  int rc = st_create(pb, opt, &c->h);
  if (rc != 0) { delete c; return rc; }
  c->q = pb->q; c->p = pb->p; c->k = ntheta; c->n_all = pb->n_all; c->seed = seed; c->rng = HostRng(seed);
  if (flags) { c->adapting = flags->adapting; c->sample_beta = flags->sample_beta; c->sample_tausq = flags->sample_tausq;
               c->sample_theta = flags->sample_theta; c->sample_w = flags->sample_w; }
  c->bounds.assign(set_unif_bounds, set_unif_bounds + (size_t)ntheta * 2);
  c->param.assign(theta, theta + ntheta);
  c->Bcoeff.assign((size_t)c->p * c->q, 0.0);
  for (int j = 0; j < c->q; ++j) for (int i = 0; i < c->p; ++i) c->Bcoeff[(size_t)j * c->p + i] = beta[i];   // :124-129
  c->tausq_inv.assign(c->q, 1.0 / tausq);
  c->xtx.assign((size_t)c->p * c->p * c->q, 0.0);
  c->Vi.assign((size_t)c->p * c->p, 0.0);
  for (int i = 0; i < c->p; ++i) c->Vi[(size_t)i * c->p + i] = 0.01;                                           // :157-159
  c->Vim.assign(c->p, 0.0);
  c->n_obs_q.assign(c->q, 0);
  auto fail = [&](int code, const std::string &msg) { c->err = msg; *out = c; return code; };
  *out = c;
  if ((rc = st_set_beta(c->h, c->Bcoeff.data())) != 0) return fail(rc, st_last_error(c->h));
  if ((rc = st_set_tausq_inv(c->h, c->tausq_inv.data())) != 0) return fail(rc, st_last_error(c->h));
  if ((rc = st_xtx(c->h, c->xtx.data())) != 0) return fail(rc, st_last_error(c->h));
  std::vector<double> ssq(c->q);
  std::vector<int64_t> nq(c->q);
  if ((rc = st_tausq_stats(c->h, ssq.data(), nq.data())) != 0) return fail(rc, st_last_error(c->h));
  for (int j = 0; j < c->q; ++j) c->n_obs_q[j] = nq[j];
  c->theta_alt = c->param;
  if (!c->am.init(ntheta, mcmcsd)) return fail(ST_ERR_USAGE, "mcmcsd is not positive definite");
  return ST_OK;
}

// w starts at zero (start_w is ignored, :95); both cache slots are factorised at the starting theta (:110-111).
// Separate from stm_create so that a multi-GPU caller can attach the communicator (st_comm_init on stm_handle) first.
extern "C" int stm_init(stm_chain c) {
  if (!c || !c->h) return ST_ERR_USAGE;
  if (c->initialised) return ST_OK;
  for (int slot = 0; slot < 2; ++slot) {
    const int rc = st_factor(c->h, slot, c->param.data(), c->k, &c->loglik[slot]);
    if (rc < 0) { c->err = st_last_error(c->h); return rc; }
    if (rc > 0) { c->err = "starting theta is not positive definite"; return ST_ERR_USAGE; }
  }
  c->current_loglik = c->loglik[0];
  c->initialised = true;
  return ST_OK;
}

// spamtree_fit.cpp:183-289 : w sweep, its log-density, Metropolis step for theta
static int draw_tausq_beta(stm_chain c);
// early_ok: nothing that changes w (a prediction on a saved iteration: :300-306) will run between this step and the tausq / beta draws
static int step_w_theta(stm_chain c, bool early_ok = true) {
  const uint32_t m = (uint32_t)c->m;
  int rc;
  const int k = c->k;
  RAMAdapt &am = c->am;
  std::vector<double> U(k), np(k);
  if (c->sample_theta) {
    // The proposal (:211-229) needs only theta, the adaptation state and its own normals (counter-based streams: drawing
    // them before the sweep's changes no value) -- so the factorisation of the latency-bound top levels for it can start
    // now and run under the sweep (st_factor_begin; results identical)
    am.count_proposal();
    for (int i = 0; i < k; ++i) U[i] = c->rng.normal(i, 0, m, 1);
    for (int j = 0; j < k; ++j) {                       // par_huvtransf_back(par_huvtransf_fwd(param) + paramsd * U)
      double f = logit(c->param[j], c->bounds[j], c->bounds[k + j]);
      for (int i = 0; i < k; ++i) f += am.paramsd[(size_t)i * k + j] * U[i];
      np[j] = logistic(f, c->bounds[j], c->bounds[k + j]);
    }
    for (int j = 0; j < k; ++j) {                       // unif_bounds (mh_adapt.h:188-202)
      if (np[j] < c->bounds[j]) np[j] = c->bounds[j] + 1e-10;
      if (np[j] > c->bounds[k + j]) np[j] = c->bounds[k + j] - 1e-10;
    }
    if (c->sample_w) {
      rc = st_factor_begin(c->h, 1, np.data(), k);
      if (rc < 0) { c->err = st_last_error(c->h); return rc; }
    }
  }
  // deal_with_w + get_loglik_w (:182-185).  The sweep's log-density is first read in the Metropolis step below: when a
  // proposal follows, sweep and phase C are only ENQUEUED here and their results come along with the synchronisation at the end
  // of the proposal's factorisation (one host round trip and one idle gap of the GPU less per iteration)
  bool sweep_open = false;
  auto finish_sweep = [&]() -> int {
    const int r2 = st_sample_w_loglik_end(c->h, &c->loglik[0]);
    if (r2 > 0) { c->err = "Error at gibbs_sample_w"; return r2; }
    if (r2 < 0) { c->err = st_last_error(c->h); return r2; }
    c->current_loglik = c->loglik[0];
    return 0;
  };
  if (c->sample_w) {
    rc = st_sample_w_loglik_begin(c->h, nullptr, c->seed, m, 0);
    if (rc) { c->err = st_last_error(c->h); return rc; }
    static const bool defer = !(getenv("SPAMTREE_DEFER_SYNC") && getenv("SPAMTREE_DEFER_SYNC")[0] == '0');   // 0: read the sweep's results at once
    if ((!c->sample_theta || !defer) && (rc = finish_sweep()) != 0) return rc;
    sweep_open = c->sample_theta && defer;
  }
  if (c->sample_theta) {
    c->theta_alt = np;
    double new_ll = c->loglik[1];
    // The tausq / beta draws of this iteration (:308-330) need the sweep's statistics only -- not the Metropolis outcome -- and
    // their streams are counter-based: they are made HERE, while the GPU factorises the proposal (the statistics ran on the
    // second stream at its start), and their uploads and the XB update queue behind phase A.  After the Metropolis step the
    // next sweep can be enqueued at once; before, two more host round trips (statistics -> draw -> upload, twice) stood between
    // the end of phase A and the next kernel: ~50 us of idle GPU per iteration (SPAMTREE_EARLY_BETA=0: the old order; same chain).
    // Not on iterations whose prediction step runs in between: with quirk Q3 the beta statistics pair y with w of ALL rows
    // (spamtree_model.cpp:1375), i.e. they see the predicted values.
    static const bool early = !(getenv("SPAMTREE_EARLY_BETA") && getenv("SPAMTREE_EARLY_BETA")[0] == '0');
    if (early && early_ok && c->sample_w && st_factor_is_async(c->h)) {
      rc = st_factor_enqueue(c->h, 1, np.data(), k);
      if (rc == 0) {
        const int r3 = draw_tausq_beta(c);
        c->tb_drawn = true;
        rc = st_factor_finish(c->h, &new_ll);
        if (r3) return r3;
      }
    } else
    rc = st_factor(c->h, 1, np.data(), k, &new_ll);
    if (sweep_open) {
      const int r2 = finish_sweep();   // (a failed sweep outranks whatever the factorisation of the proposal made of its w)
      if (r2) return r2;
    }
    if (rc < 0) { c->err = st_last_error(c->h); return rc; }
    const bool acceptable = rc == 0;
    if (acceptable) c->loglik[1] = new_ll;
    c->current_loglik = c->loglik[0];
    if (std::isnan(c->current_loglik)) { c->err = "At nan loglik: error."; return STM_ERR_NAN; }   // throw 1 (:234-237)
    double jac = 0;                                      // calc_jacobian (mh_adapt.h:210-239)
    for (int j = 0; j < k; ++j) {
      const double l = c->bounds[j], u = c->bounds[k + j];
      jac += (-std::log(u - c->param[j]) - std::log(c->param[j] - l)) - (-std::log(u - np[j]) - std::log(np[j] - l));
    }
    const double logaccept = c->loglik[1] - c->current_loglik + jac;
    double acceptj = 1.0;                                // do_I_accept (mh_adapt.h:20-36); the uniform is always drawn
    if (!std::isfinite(logaccept)) acceptj = 0.0;
    else if (logaccept < 0) acceptj = std::exp(logaccept);
    const double u = c->rng.uniform(0, 0, m, 2);
    const bool accepted = (u < acceptj) && acceptable;
    if (accepted) {
      am.count_accepted();
      c->current_loglik = c->loglik[1];
      st_swap(c->h);                                     // accept_make_change
      std::swap(c->loglik[0], c->loglik[1]);
      std::swap(c->param, c->theta_alt);
    }
    am.update_ratios();
    if (c->adapting) am.adapt(U, (acceptable ? 1.0 : 0.0) * std::exp(logaccept), (int)c->m);
    c->last_accepted = accepted; c->last_acceptable = acceptable; c->last_logaccept = logaccept;
  }
  return ST_OK;
}

// spamtree_fit.cpp:308-330 with the host parts of gibbs_sample_tausq (:1393-1417) and gibbs_sample_beta (:1364-1391)
static int draw_tausq_beta(stm_chain c) {
  const uint32_t m = (uint32_t)c->m;
  const int p = c->p, q = c->q;
  int rc;
  if (c->sample_tausq) {
    std::vector<double> ssq(q);
    if ((rc = st_tausq_stats(c->h, ssq.data(), nullptr)) != 0) { c->err = st_last_error(c->h); return rc; }
    for (int j = 0; j < q; ++j) {
      const double aparam = 2.01 + c->n_obs_q[j] / 2.0, bparam = 1.0 / (1.0 + 0.5 * ssq[j]);
      c->tausq_inv[j] = c->rng.gamma(m, j, aparam, bparam);
    }
    if ((rc = st_set_tausq_inv(c->h, c->tausq_inv.data())) != 0) { c->err = st_last_error(c->h); return rc; }
  }
  if (c->sample_beta) {
    std::vector<double> xty((size_t)p * q);
    if ((rc = st_beta_stats(c->h, xty.data())) != 0) { c->err = st_last_error(c->h); return rc; }
    for (int j = 0; j < q; ++j) {
      Mat Si((size_t)p * p), L;
      for (int i = 0; i < p * p; ++i) Si[i] = c->tausq_inv[j] * c->xtx[(size_t)j * p * p + i] + c->Vi[i];
      if (!chol_lower(Si, p, L)) { c->err = "beta posterior precision is not positive definite"; return ST_ERR_USAGE; }
      const Mat Sc = inv_lower(L, p);
      std::vector<double> b(p), t(p, 0.0), zn(p);
      for (int i = 0; i < p; ++i) { b[i] = c->Vim[i] + c->tausq_inv[j] * xty[(size_t)j * p + i]; zn[i] = c->rng.normal(i, j, m, 4); }
      for (int i = 0; i < p; ++i) { double s = 0; for (int kk = 0; kk <= i; ++kk) s += Sc[(size_t)kk * p + i] * b[kk]; t[i] = s + zn[i]; }
      for (int i = 0; i < p; ++i) {                      // Sc' (Sc b) + Sc' z  =  Sc' (Sc b + z)
        double s = 0;
        for (int kk = i; kk < p; ++kk) s += Sc[(size_t)i * p + kk] * t[kk];
        c->Bcoeff[(size_t)j * p + i] = s;
      }
    }
    if ((rc = st_set_beta(c->h, c->Bcoeff.data())) != 0) { c->err = st_last_error(c->h); return rc; }
  }
  return ST_OK;
}
static int step_tausq_beta(stm_chain c) {
  if (!c->tb_drawn) { const int rc = draw_tausq_beta(c); if (rc) return rc; }   // (else: drawn under the proposal's factorisation, step_w_theta)
  c->tb_drawn = false;
  c->m += 1;
  return ST_OK;
}

extern "C" int stm_step(stm_chain c, int n_iters) {
  if (!c || !c->h) return ST_ERR_USAGE;
  if (!c->initialised) { const int rc0 = stm_init(c); if (rc0) return rc0; }
  for (int it = 0; it < n_iters; ++it) {
    int rc = step_w_theta(c);
    if (rc) return rc;
    rc = step_tausq_beta(c);
    if (rc) return rc;
  }
  return ST_OK;
}

extern "C" int stm_state(stm_chain c, double *theta, double *Bcoeff, double *tausq_inv, double *loglik, double *accept_ratio,
                         int64_t *iteration, double *paramsd) {
  if (!c) return ST_ERR_USAGE;
  if (theta) std::memcpy(theta, c->param.data(), c->param.size() * sizeof(double));
  if (Bcoeff) std::memcpy(Bcoeff, c->Bcoeff.data(), c->Bcoeff.size() * sizeof(double));
  if (tausq_inv) std::memcpy(tausq_inv, c->tausq_inv.data(), c->tausq_inv.size() * sizeof(double));
  if (loglik) *loglik = c->current_loglik;
  if (accept_ratio) *accept_ratio = c->am.accept_ratio;
  if (iteration) *iteration = c->m;
  if (paramsd) std::memcpy(paramsd, c->am.paramsd.data(), c->am.paramsd.size() * sizeof(double));
  return ST_OK;
}

// The whole of spamtree_mv_mcmc (spamtree_fit.cpp:5-430): outputs into caller buffers (column-major):
// beta_mcmc p x keep x q, tausq_mcmc q x keep, theta_mcmc k x keep, w_mcmc / yhat_mcmc n_all x keep (may be NULL).
extern "C" int spamtree_mv_mcmc_c(const st_problem *pb, const st_options *opt, const double *set_unif_bounds, const double *theta,
                                  int ntheta, const double *beta, double tausq, const double *mcmcsd, int mcmc_keep, int mcmc_burn,
                                  int mcmc_thin, uint64_t seed, const stm_flags *flags, double *w_mcmc, double *yhat_mcmc,
                                  double *beta_mcmc, double *tausq_mcmc, double *theta_mcmc, double *paramsd, double *mcmc_time) {
  stm_chain c = nullptr;
  int rc = stm_create(pb, opt, set_unif_bounds, mcmcsd, theta, ntheta, beta, tausq, seed, flags, &c);
  if (rc == 0) rc = stm_init(c);
  if (rc != 0) { stm_destroy(c); return rc; }
  const int sample_predicts = flags ? flags->sample_predicts : 1;
  const auto t0 = std::chrono::steady_clock::now();
  const long long mcmc = (long long)mcmc_thin * mcmc_keep + mcmc_burn;
  std::vector<double> predict_param = c->param;
  int msaved = 0;
  const int p = c->p, q = c->q, k = c->k;
  const long long n = c->n_all;
  for (long long m = 0; m < mcmc && rc == 0; ++m) {
    const long long mx = m - mcmc_burn;
    const bool saving = mx >= 0 && (mx % mcmc_thin == 0);
    rc = step_w_theta(c, !(saving && sample_predicts && c->sample_w));
    if (rc) break;
    if (saving && sample_predicts && c->sample_w) {                    // :300-306
      int need_update = 0;
      for (int j = 0; j < k; ++j) need_update += std::fabs(c->param[j] - predict_param[j]) > 1e-05;
      rc = st_predict(c->h, need_update);
      if (rc) break;
      predict_param = c->param;
    }
    rc = step_tausq_beta(c);
    if (rc) break;
    if (saving && msaved < mcmc_keep) {                                 // :376-389
      if (tausq_mcmc) for (int j = 0; j < q; ++j) tausq_mcmc[(size_t)msaved * q + j] = 1.0 / c->tausq_inv[j];
      if (beta_mcmc) for (int j = 0; j < q; ++j) for (int i = 0; i < p; ++i)
        beta_mcmc[(size_t)j * p * mcmc_keep + (size_t)msaved * p + i] = c->Bcoeff[(size_t)j * p + i];
      if (theta_mcmc) std::memcpy(theta_mcmc + (size_t)msaved * k, c->param.data(), k * sizeof(double));
      if (w_mcmc) rc = st_get_w(c->h, w_mcmc + (size_t)msaved * n);
      if (!rc && yhat_mcmc) rc = st_yhat(c->h, nullptr, seed, (uint32_t)m, yhat_mcmc + (size_t)msaved * n);
      ++msaved;
    }
  }
  if (rc == 0) {
    if (paramsd) std::memcpy(paramsd, c->am.paramsd.data(), (size_t)k * k * sizeof(double));
    if (mcmc_time) *mcmc_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  stm_destroy(c);
  return rc;
}
