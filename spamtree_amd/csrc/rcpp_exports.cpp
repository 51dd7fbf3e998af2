// R-side face of the library: the two functions the reference exports to R (NAMESPACE:14, 17 of /root/reference), with the
// reference's names, argument lists and returned list -- compiled ONLY inside the R package build (SPAMTREE_WITH_RCPP, where
// Rcpp + RcppArmadillo exist; Rcpp::compileAttributes() then regenerates the SEXP wrappers `_spamtree_spamtree_mv_mcmc`,
// 35 arguments, and `_spamtree_CrossCovarianceAG10`, 9 arguments, exactly as /root/reference/src/RcppExports.cpp:20-38,
// 112-154 has them).  Everything below the argument conversion is the C-ABI of include/spamtree_fit.h / spamtree_hip.h.
// R, Rcpp and Armadillo are absent from the build image: tests/test_rcpp_shim.py only checks that this file parses and
// type-checks against a stub of the few Rcpp / arma names it uses (tests/stubs/RcppArmadillo.h).  See INTEGRATION.md.
#ifdef SPAMTREE_WITH_RCPP
#include <RcppArmadillo.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "spamtree_fit.h"
#include "spamtree_hip.h"

namespace {
void field_to_csr(const arma::field<arma::uvec> &f, std::vector<int64_t> &ptr, std::vector<int64_t> &idx) {
  ptr.assign(f.n_elem + 1, 0);
  for (arma::uword i = 0; i < f.n_elem; ++i) ptr[i + 1] = ptr[i] + (int64_t)f(i).n_elem;
  idx.resize((size_t)ptr.back());
  for (arma::uword i = 0; i < f.n_elem; ++i)
    for (arma::uword k = 0; k < f(i).n_elem; ++k) idx[(size_t)ptr[i] + k] = (int64_t)f(i)(k);
}
}   // namespace

// spamtree_mv_mcmc: /root/reference/src/spamtree_fit.cpp:5-54 (arguments), :403-414 (returned names), :416-428 (failure list)
// [[Rcpp::export]]
Rcpp::List spamtree_mv_mcmc(const arma::mat &y, const arma::mat &X, const arma::mat &Z, const arma::mat &coords, const arma::uvec &mv_id,
                            const arma::uvec &blocking, const arma::uvec &gix_block, const arma::uvec &res_is_ref,
                            const arma::field<arma::uvec> &parents, const arma::field<arma::uvec> &children, bool limited_tree,
                            const arma::vec &layer_names, const arma::vec &layer_gibbs_group, const arma::field<arma::uvec> &indexing,
                            const arma::mat &set_unif_bounds_in, const arma::mat &start_w, const arma::vec &theta, const arma::vec &beta,
                            const double &tausq, const arma::mat &mcmcsd, int mcmc_keep = 100, int mcmc_burn = 100, int mcmc_thin = 1,
                            int num_threads = 1, char use_alg = 'S', bool adapting = false, bool main_verbose = true, bool verbose = false,
                            bool debug = false, bool printall = false, bool sample_beta = true, bool sample_tausq = true,
                            bool sample_theta = true, bool sample_w = true, bool sample_predicts = true) {
  (void)blocking; (void)gix_block; (void)start_w; (void)num_threads; (void)use_alg; (void)main_verbose; (void)verbose; (void)debug; (void)printall;
  std::vector<int64_t> ip, ii, pp, pi, cp, ci, mv(mv_id.begin(), mv_id.end()), rr(res_is_ref.begin(), res_is_ref.end()),
      bn(layer_names.n_elem), bg(layer_gibbs_group.n_elem);
  for (size_t i = 0; i < bn.size(); ++i) { bn[i] = (int64_t)layer_names(i); bg[i] = (int64_t)layer_gibbs_group(i); }
  field_to_csr(indexing, ip, ii); field_to_csr(parents, pp, pi); field_to_csr(children, cp, ci);
  const int q = (int)Z.n_cols, p = (int)X.n_cols, k = (int)theta.n_elem;
  st_problem pb = {(int64_t)coords.n_rows, (int32_t)coords.n_cols, q, p, (int32_t)rr.size(), (int64_t)bn.size(), y.memptr(), X.memptr(),
                   coords.memptr(), mv.data(), rr.data(), bn.data(), bg.data(), ip.data(), ii.data(), pp.data(), pi.data(), cp.data(), ci.data()};
  st_options opt = {0, 1, 0, 1, 0, limited_tree ? 2 : 0};
  stm_flags fl = {adapting, sample_beta, sample_tausq, sample_theta, sample_w, sample_predicts};
  arma::cube beta_mcmc(p, mcmc_keep, q, arma::fill::zeros);
  arma::mat tausq_mcmc(q, mcmc_keep, arma::fill::zeros), theta_mcmc(k, mcmc_keep, arma::fill::zeros), paramsd(k, k, arma::fill::zeros);
  arma::mat w_all(coords.n_rows, mcmc_keep, arma::fill::zeros), yhat_all(coords.n_rows, mcmc_keep, arma::fill::zeros);
  double mcmc_time = 0;
  const uint64_t seed = (uint64_t)std::floor(R::runif(0, 1) * 9007199254740992.0);   // chain seed from R's generator (set.seed applies)
  const int rc = spamtree_mv_mcmc_c(&pb, &opt, set_unif_bounds_in.memptr(), theta.memptr(), k, beta.memptr(), tausq, mcmcsd.memptr(), mcmc_keep,
                                    mcmc_burn, mcmc_thin, seed, &fl, w_all.memptr(), yhat_all.memptr(), beta_mcmc.memptr(), tausq_mcmc.memptr(),
                                    theta_mcmc.memptr(), paramsd.memptr(), &mcmc_time);
  if (rc == STM_ERR_NAN) throw 1;                                                  // spamtree_fit.cpp:234-237 ("At nan loglik: error.")
  if (rc != 0) return Rcpp::List::create(Rcpp::Named("None") = arma::zeros(0));   // :416-428
  arma::field<arma::mat> w_mcmc(mcmc_keep), yhat_mcmc(mcmc_keep);
  for (int i = 0; i < mcmc_keep; ++i) { w_mcmc(i) = w_all.col(i); yhat_mcmc(i) = yhat_all.col(i); }
  // model fields the reference hands back untouched and spamtree() passes on to the user (R/spamtree_fit.R:365-368):
  // block_ct_obs (na_study, spamtree_model.cpp:303-313), indexing (:101), parents_indexing (init_indexing, :328-335)
  const arma::uword nb = indexing.n_elem;
  arma::uvec block_ct_obs(nb);
  arma::field<arma::uvec> parents_indexing(nb);
  for (arma::uword u = 0; u < nb; ++u) {
    arma::uword ct = 0;
    for (arma::uword j = 0; j < indexing(u).n_elem; ++j) ct += std::isfinite(y(indexing(u)(j))) ? 1 : 0;
    block_ct_obs(u) = ct;
    arma::uword np = 0;
    for (arma::uword t = 0; t < parents(u).n_elem; ++t) np += indexing(parents(u)(t)).n_elem;
    arma::uvec pix(np);
    arma::uword at = 0;
    for (arma::uword t = 0; t < parents(u).n_elem; ++t)
      for (arma::uword j = 0; j < indexing(parents(u)(t)).n_elem; ++j) pix(at++) = indexing(parents(u)(t))(j);
    parents_indexing(u) = pix;
  }
  return Rcpp::List::create(Rcpp::Named("w_mcmc") = w_mcmc, Rcpp::Named("yhat_mcmc") = yhat_mcmc, Rcpp::Named("beta_mcmc") = beta_mcmc,
                            Rcpp::Named("tausq_mcmc") = tausq_mcmc, Rcpp::Named("theta_mcmc") = theta_mcmc, Rcpp::Named("paramsd") = paramsd,
                            Rcpp::Named("block_ct_obs") = block_ct_obs, Rcpp::Named("indexing") = indexing,
                            Rcpp::Named("parents_indexing") = parents_indexing, Rcpp::Named("mcmc_time") = mcmc_time);
}

// CrossCovarianceAG10: /root/reference/src/covariance_functions.cpp:301-355 (exported, NAMESPACE:14; man/CrossCovarianceAG10.Rd)
// [[Rcpp::export]]
arma::mat CrossCovarianceAG10(arma::mat coords1, arma::uvec mv1, arma::mat coords2, arma::uvec mv2, arma::vec ai1, arma::vec ai2,
                              arma::vec phi_i, arma::vec thetamv, arma::mat Dmat) {
  arma::mat res(coords1.n_rows, coords2.n_rows, arma::fill::zeros);
  if (Dmat.n_cols < 2) Rcpp::stop("Invalid Dmat for multivariate data");           // the reference's message (:338-340)
  std::vector<int64_t> m1(mv1.begin(), mv1.end()), m2(mv2.begin(), mv2.end());     // 1-based, as R passes them
  const int rc = st_cross_covariance_ag10(coords1.memptr(), m1.data(), (int64_t)coords1.n_rows, coords2.memptr(), m2.data(),
                                          (int64_t)coords2.n_rows, ai1.memptr(), ai2.memptr(), phi_i.memptr(), thetamv.memptr(),
                                          Dmat.memptr(), (int32_t)Dmat.n_cols, 0, res.memptr());
  if (rc != 0) Rcpp::stop(st_last_error(nullptr));
  return res;
}
#endif   // SPAMTREE_WITH_RCPP
