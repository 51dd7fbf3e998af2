/* spamtree_fit.h -- C-ABI of the C++ host MCMC driver (spamtree_amd/csrc/spamtree_fit.cpp), the counterpart of the
 * reference's Rcpp-exported `spamtree_mv_mcmc` (/root/reference/src/spamtree_fit.cpp:5-54, registered at
 * /root/reference/src/RcppExports.cpp:112-154, 239).  It sits ABOVE include/spamtree_hip.h and contains no kernels.
 *
 * Random draws: the reference uses R's generator through Rcpp (arma::randn, R::runif, R::rgamma), which does not exist
 * outside R.  Contract here: Philox4x32-10, key = seed, counter = (index_lo, index_hi | outcome, iteration, stream);
 *   stream 0  sweep normals z (device, counter index = row)          spamtree_model.cpp:1018
 *   stream 1  theta proposal normals (index = component)             spamtree_fit.cpp:211
 *   stream 2  Metropolis uniform                                     mh_adapt.h:30
 *   stream 3  gamma draws (Marsaglia-Tsang; index = 2*attempt [+1])  spamtree_model.cpp:1405
 *   stream 4  beta normals (index = component, hi = outcome)         spamtree_model.cpp:1378
 *   stream 5  yhat noise (device)                                    spamtree_fit.cpp:384
 * normal = sqrt(-2 ln u1) cos(2 pi u2), u from 53 bits of two 32-bit words.  Draws are identical for any GPU count.
 */
#ifndef SPAMTREE_FIT_H
#define SPAMTREE_FIT_H

#include "spamtree_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define STM_ERR_NAN (-10) /* "At nan loglik: error." -- the reference's `throw 1` (spamtree_fit.cpp:234-237) */

typedef struct stm_chain_s *stm_chain;

typedef struct stm_flags {   /* the reference's boolean arguments (spamtree_fit.cpp:44-54) */
  int32_t adapting, sample_beta, sample_tausq, sample_theta, sample_w, sample_predicts;
} stm_flags;

/* One chain = SpamTreeMV + RAMAdapt + the loop state of spamtree_fit.cpp:93-165.  set_unif_bounds: k x 2 column-major;
 * mcmcsd: k x k; beta: p (copied to every outcome, spamtree_model.cpp:124-129); w starts at 0 (start_w is ignored, :95). */
int stm_create(const st_problem *pb, const st_options *opt, const double *set_unif_bounds, const double *mcmcsd, const double *theta,
               int ntheta, const double *beta, double tausq, uint64_t seed, const stm_flags *flags, stm_chain *out);
int stm_init(stm_chain c);  /* the two initial factorisations (spamtree_fit.cpp:110-111); stm_step calls it when needed.  With
                              world > 1 attach the communicator first: st_comm_init(stm_handle(c), id) */
int stm_destroy(stm_chain c);
const char *stm_last_error(stm_chain c);
st_handle stm_handle(stm_chain c);
/* n_iters bodies of the loop spamtree_fit.cpp:167-391 without prediction and saving (B, C, theta MH with phase A, tausq, beta) */
int stm_step(stm_chain c, int n_iters);
int stm_state(stm_chain c, double *theta, double *Bcoeff, double *tausq_inv, double *loglik, double *accept_ratio, int64_t *iteration,
              double *paramsd);

/* The whole fit.  Outputs (caller buffers, column-major, any may be NULL): w_mcmc, yhat_mcmc n_all x keep;
 * beta_mcmc p x keep x q; tausq_mcmc q x keep; theta_mcmc k x keep; paramsd k x k; mcmc_time seconds. */
int spamtree_mv_mcmc_c(const st_problem *pb, const st_options *opt, const double *set_unif_bounds, const double *theta, int ntheta,
                       const double *beta, double tausq, const double *mcmcsd, int mcmc_keep, int mcmc_burn, int mcmc_thin, uint64_t seed,
                       const stm_flags *flags, double *w_mcmc, double *yhat_mcmc, double *beta_mcmc, double *tausq_mcmc,
                       double *theta_mcmc, double *paramsd, double *mcmc_time);

#ifdef __cplusplus
}
#endif
#endif
