/* spamtree_hip.h -- C-ABI of the MI355X (gfx950) build of spamtree's per-Gibbs-sweep DAG-node hot path.
 *
 * The boundary sits UNDER the reference's `SpamTreeMV` model object (/root/reference/src/spamtree_model.h:22-212):
 * each entry point replaces one of its hot methods; the host driver above it (spamtree_mv_mcmc,
 * /root/reference/src/spamtree_fit.cpp:5-430) keeps owning the RNG, the Metropolis step and the outputs.
 * Plain pointers and sizes only; no C++ / torch types.  All matrices are column-major doubles (Armadillo's
 * layout), all index vectors int64 and 0-based unless stated, rows are in the order R hands them to C++
 * (sorted by coordinates, /root/reference/R/spamtree_fit.R:267-269).  Pointers are borrowed for the call only;
 * the handle owns every device allocation.  One host thread per handle; the handle is not re-entrant.
 *
 * Return value of every function: 0 = ok; 1/2/3 = Cholesky failed in phase A at the root / a reference block /
 * a non-reference row (the reference's `errtype`, spamtree_model.cpp:876, 919, 958); 10/11 = Cholesky failed in
 * the w sweep (spamtree_model.cpp:1056, 1135); negative = usage / HIP error (st_last_error() has the text).
 * No exception crosses this boundary.
 */
#ifndef SPAMTREE_HIP_H
#define SPAMTREE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST_OK 0
#define ST_ERR_CHOL_ROOT 1
#define ST_ERR_CHOL_REF 2
#define ST_ERR_CHOL_LEAF 3
#define ST_ERR_CHOL_SAMPLE_REF 10
#define ST_ERR_CHOL_SAMPLE_LEAF 11
#define ST_ERR_USAGE (-1)
#define ST_ERR_HIP (-2)
#define ST_ERR_TOPOLOGY (-3)
#define ST_ERR_UNSUPPORTED (-4)

#define ST_MAX_Q 6          /* outcomes (theta has 3q + (q>2?3:1) + q(q-1)/2 <= 39 entries) */
#define ST_MAX_ANCESTORS 24 /* tree depth - 1 */

typedef struct st_handle_s *st_handle;

/* Inputs of the SpamTreeMV constructor (spamtree_model.cpp:8-37) that the hot path needs. */
typedef struct st_problem {
  int64_t n_all;               /* rows (observed + NA)                                   coords.n_rows            */
  int32_t d;                   /* coordinate columns, must be 2                           coords.n_cols            */
  int32_t q;                   /* outcomes                                                unique(mv_id)            */
  int32_t p;                   /* regressors                                              X.n_cols                 */
  int32_t n_groups;            /* length of res_is_ref                                                             */
  int64_t n_blocks;            /*                                                         block_names.n_elem       */
  const double *y;             /* n_all, NaN = NA                                         y_in                     */
  const double *X;             /* n_all x p column-major                                  X_in                     */
  const double *coords;        /* n_all x d column-major                                  coords_in                */
  const int64_t *mv_id;        /* n_all, 1-based outcome id                               mv_id_in                 */
  const int64_t *res_is_ref;   /* n_groups flags, indexed by level rank                   res_is_ref_in            */
  const int64_t *block_names;  /* n_blocks, 1-based ids                                   block_names_in           */
  const int64_t *block_groups; /* n_blocks, level ("res") of block id-1                   block_groups_in          */
  const int64_t *indexing_ptr; /* n_blocks+1 CSR offsets                                  indexing_in (field)      */
  const int64_t *indexing_idx; /* row ids of each block, ascending                                                 */
  const int64_t *parents_ptr;  /* n_blocks+1                                              parents_in (field)       */
  const int64_t *parents_idx;  /* ancestor block ids, ascending (root first)                                       */
  const int64_t *children_ptr; /* n_blocks+1, may be NULL (derived from parents)          children_in (field)      */
  const int64_t *children_idx; /* all non-empty descendants, ascending; may be NULL                                */
} st_problem;

typedef struct st_options {
  int32_t device;              /* HIP device ordinal                                                               */
  int32_t reference_quirks;    /* 1 = reproduce spamtree_model.cpp:1375 (Q3: beta uses subset positions on full w) */
  int32_t rank;                /* multi-GPU: this process' rank ...                                                */
  int32_t world;               /* ... of `world` processes sharing one problem (1 = single GPU); see the end of this file */
  int32_t force_generic;       /* 1 = use the global-scratch kernels even where the LDS kernels fit (testing)      */
  int32_t reserved;            /* bit 0: recompute the theta-only Gram part of the messages on every sweep, as the
                                  reference does (need_update is always true, spamtree_fit.cpp:184); default 0 caches it
                                  per accepted theta -- identical values
                                  bit 1: limited_tree = TRUE (spamtree_fit.cpp:20, spamtree_model.cpp:901-903, 1275-1278):
                                  parents(u) is the single parent of make_edges_limited (tree_dep.cpp:133-186), children(u)
                                  the direct children, and Kxx_inv(u) = inv_sympd(K_uu); sharded like full trees since round 3                  */
} st_options;

/* ---- lifetime: SpamTreeMV::SpamTreeMV (spamtree_model.cpp:8-192) incl. init_indexing/init_finalize/init_model_data */
int st_create(const st_problem *pb, const st_options *opt, st_handle *out);
int st_destroy(st_handle h);
const char *st_last_error(st_handle h); /* h may be NULL: error text of the last failed st_create */

/* ---- state the driver reads / writes directly in the reference (public fields w, Bcoeff, tausq_inv) */
int st_set_w(st_handle h, const double *w);                /* n_all, model row order */
int st_get_w(st_handle h, double *w);
int st_set_beta(st_handle h, const double *Bcoeff);        /* p x q column-major; recomputes XB (spamtree_model.cpp:127, 1382) */
int st_set_tausq_inv(st_handle h, const double *tausq_inv);/* q (spamtree_model.cpp:118, 1405-1407) */
int st_get_xb(st_handle h, double *xb);                    /* n_all */

/* ---- phase A: theta_update + get_loglik_comps_w(data) (spamtree_model.cpp:834-998, 1420-1422).
 * slot 0 = param_data, 1 = alter_data.  Returns 0 (the reference's `true`) or 1/2/3 (`false`, errtype);
 * *loglik = data.loglik_w (undefined on failure).  theta has ntheta = 3q + (q>2?3:1) + q(q-1)/2 entries. */
int st_factor(st_handle h, int slot, const double *theta, int ntheta, double *loglik);
/* st_factor in two halves: _enqueue starts the work (one GPU: every launch and the copy of the results; with a communicator
 * attached: nothing yet), _finish waits and returns what st_factor returns.  In between the caller may issue calls that do not
 * touch the slot -- the C++ driver draws tausq / beta from the sweep's statistics and uploads them (st_tausq_stats, st_beta_stats,
 * st_set_tausq_inv, st_set_beta: none of them waits for the main stream), so that the Metropolis step's host round trip
 * (src/spamtree_fit.cpp:232-262, then :308-330) is the only one of the iteration.  No st_swap / st_sample_w* in between. */
int st_factor_enqueue(st_handle h, int slot, const double *theta, int ntheta);
int st_factor_is_async(st_handle h);   /* 1: _enqueue really starts the work (one GPU, no communicator) */
int st_factor_finish(st_handle h, double *loglik);
/* optional: start phase A of the latency-bound top levels ahead of time -- they depend on theta only (their blocks'
 * quadratic forms are redone with the current w afterwards) -- on a second stream, e.g. before the sweep; the next
 * st_factor / st_factor_local for the same slot and theta picks the result up.  Identical results; a no-op when the tree
 * does not qualify (column-group levels above a k_factor_quad level) or SPAMTREE_ASYNC_TOP=0.  (The proposal of spamtree_fit.cpp:211-229 does not depend on the sweep.) */
/* Contract: between st_factor_begin(slot, theta) and the st_factor / st_factor_local that picks its result up, st_swap is
 * refused (ST_ERR_USAGE: the arena being written would become the accepted slot); readers of the slot (st_get_block,
 * st_get_comps, st_loglik_w(1), st_mg_pack_comps) are ordered behind the launches in flight. */
int st_factor_begin(st_handle h, int slot, const double *theta, int ntheta);
int st_factor_ahead_levels(st_handle h);   /* how many leading levels st_factor_begin runs ahead (0: none) */
/* measurement only: 0 switches the ahead-of-time path off (st_factor_begin becomes a no-op, every level runs inside
 * st_factor on the launch stream), 1 back on.  Results are identical either way. */
int st_factor_ahead_enable(st_handle h, int enable);

/* ---- accept_make_change (spamtree_model.cpp:1432-1435): swap the two cache slots */
int st_swap(st_handle h);

/* ---- phase B: gibbs_sample_w_std(true) on param_data (spamtree_model.cpp:1011-1226).
 * z = the reference's bigrnorm (n_all standard normals, model row order).  z == NULL: generate on device,
 * z_i = normal(Philox4x32-10; key=seed, counter=(row, row>>32, iter, 0)) -- identical for any GPU count. */
int st_sample_w(st_handle h, const double *z, uint64_t seed, uint32_t iter);

/* ---- phase C: get_loglik_w_std(data) (spamtree_model.cpp:781-826) */
int st_loglik_w(st_handle h, int slot, double *loglik);

/* st_sample_w followed by st_loglik_w(slot), same results and return codes (the sweep's 10 / 11 first), with a single
 * host synchronisation: what the MCMC driver calls once per iteration (spamtree_fit.cpp:182-185). */
int st_sample_w_loglik(st_handle h, const double *z, uint64_t seed, uint32_t iter, int slot, double *loglik);
/* The same pair without its host synchronisation (spamtree_fit.cpp:182-185 then :211-289: the sweep's log-density is not read
 * before the Metropolis step): _begin enqueues sweep + phase C and returns; any later synchronising call (st_factor) brings the
 * results along; _end returns what st_sample_w_loglik would have (0 and *loglik, or the sweep's failure code 10 / 11).  One
 * _end per _begin, before the next sweep.  Multi-GPU handles run the synchronous protocol inside _begin. */
int st_sample_w_loglik_begin(st_handle h, const double *z, uint64_t seed, uint32_t iter, int slot);
int st_sample_w_loglik_end(st_handle h, double *loglik);

/* ---- phase P: predict_std(true, theta_changed) on param_data (spamtree_model.cpp:1234-1358); uses the last sweep's z */
int st_predict(st_handle h, int theta_changed);

/* ---- reductions for gibbs_sample_beta / gibbs_sample_tausq (spamtree_model.cpp:1374-1375, 1397-1400).
 * xty: p x q column-major, column j = X_avail_j' (y_avail_j - w[...]);  ssq: q, sum (y - XB - w)^2 over observed rows.
 * n_obs_by_q: q (may be NULL).  The draws themselves (R::rgamma, arma::randn) stay with the host driver. */
int st_beta_stats(st_handle h, double *xty);
int st_tausq_stats(st_handle h, double *ssq, int64_t *n_obs_by_q);
int st_xtx(st_handle h, double *xtx);                      /* p x p x q, XtX(j) of spamtree_model.cpp:151-155 */

/* ---- yhat = XB + w + tausq^{1/2} * normal (spamtree_fit.cpp:384); noise==NULL: device stream 5 */
int st_yhat(st_handle h, const double *noise, uint64_t seed, uint32_t iter, double *yhat);

/* ---- inspection (parity tests): per-block caches of a slot.
 * For block u (0-based id) with m rows and P ancestor rows the build keeps the inverse-Cholesky row panel
 * [ -Ri*H | Ri ] (tree_utils.cpp:204-206) instead of H, Kxx_inv, Kxx_invchol separately.
 * st_block_dims: *m, *P, *is_ref.  st_get_block: negRiH (m x P, column-major) and Ri (m x m column-major for a
 * reference block, m diagonal entries for a non-reference block). */
int st_block_dims(st_handle h, int64_t u, int64_t *m, int64_t *P, int32_t *is_ref, int32_t *n_obs);
int st_get_block(st_handle h, int slot, int64_t u, double *negRiH, double *Ri);
int st_get_comps(st_handle h, int slot, double *logdetCi_comps, double *loglik_w_comps); /* n_blocks each */

/* ---- measurement: algorithmic bytes of one iteration (SURVEY.md section 8d operand-streaming model)
 * out[0..4] = phase A, B, C, messages, S1+S2;  flops[0..2] = A, B, C (may be NULL). */
int st_algorithmic_bytes(st_handle h, double *out5, double *flops3);
/* box-measured peaks for the roofline report (csrc/probe.hip; no handle needed): out3[0] = stream-copy GB/s (read + written
 * bytes, `bytes` per buffer, best of `reps`), out3[1] = FP64 MFMA TFLOP/s (v_mfma_f64_16x16x4_f64), out3[2] = FP64 FMA TFLOP/s
 * (v_fma_f64).  About 0.15 s. */
int st_probe_peaks(int device, int64_t bytes, int reps, double *out3);
/* per-kernel-family device time from HIP events recorded on the launch stream around every launch (enable=1), or around
 * the phase-A launches only (enable=2: the roofline measurement at a third of the event traffic; ~60 event records per
 * iteration cost 4-6 % of the iteration at n = 1e6).  Events are harvested lazily: no host synchronisation is added.
 * families: 0 factor(A) 1 sample(B) 2 loglik(C) 3 reduce 4 stats/xb 5 rng 6 predict 7 comm (the library's RCCL collectives
 * on the launch stream: device time between the events, i.e. including the wait for the slowest rank) */
#define ST_N_KERNEL_FAMILIES 8
int st_profile_enable(st_handle h, int enable);
int st_profile_get(st_handle h, double *ms_total, int64_t *launches); /* ST_N_KERNEL_FAMILIES each; resets */
/* phase-A launches by tree level since the last call: mean ms per launch, algorithmic bytes per launch; resets */
int st_profile_levels(st_handle h, int32_t *n_levels, double *ms_by_level, double *bytes_by_level, int32_t cap);
/* which phase-A kernel each observed level takes (same dispatch as st_factor: a function of the tree only) and the sizes
 * that decide it: per level g < *n_levels (at most cap entries written): kernel[g] = one of ST_KERNEL_*, max_m[g], max_P[g]
 * (largest block / ancestor-row count), n_blocks[g].  Any output pointer but n_levels may be NULL. */
#define ST_KERNEL_GENERIC_LDS 0
#define ST_KERNEL_GENERIC_SCRATCH 1
#define ST_KERNEL_MFMA 2
#define ST_KERNEL_QUAD 3
#define ST_KERNEL_BIGMFMA 4
#define ST_KERNEL_WIDE 5
#define ST_KERNEL_LCHAIN 6
#define ST_KERNEL_LCHAIN_REF 7   /* reference level: k_factor_lchain for the chain pass + k_factor_ref_finish per block */
int st_level_info(st_handle h, int32_t *n_levels, int32_t *kernel, int32_t *max_m, int32_t *max_P, int32_t *n_blocks, int32_t cap);
int st_synchronize(st_handle h);
void *st_stream(st_handle h);                              /* the hipStream_t every kernel is launched on */

/* ---- SURVEY.md section 8f "next" rows -------------------------------------------------------------------------------
 * CrossCovarianceAG10 (/root/reference/src/covariance_functions.cpp:301-355, exported to R, NAMESPACE:14): dense
 * n1 x n2 Apanasovich-Genton cross-covariance, column-major; coords n x 2 column-major, mv 1-based, Dmat q x q.
 * Needs no handle.  q < 2 is refused like the reference ("Invalid Dmat for multivariate data"). */
int st_cross_covariance_ag10(const double *coords1, const int64_t *mv1, int64_t n1, const double *coords2, const int64_t *mv2,
                             int64_t n2, const double *ai1, const double *ai2, const double *phi_i, const double *thetamv,
                             const double *Dmat, int32_t q, int32_t device, double *out);
/* Running posterior means of w and yhat on the device (what list_mean, /root/reference/src/list_mean.cpp:10-40, computes
 * after the fact from `keep` stored copies): accumulate on every saved iteration, read the means once. */
int st_summary_reset(st_handle h);
int st_summary_accumulate(st_handle h, uint64_t seed, uint32_t iter);
int st_summary_get(st_handle h, double *w_mean, double *yhat_mean, int64_t *n_accumulated);
/* Posterior quantiles on the device (list_qtile / prctile_stl, /root/reference/src/list_mean.cpp:62-137: per row, the order
 * statistics around q * keep of the `keep` saved draws, interpolated by the reference's rule).  st_summary_reserve(keep) keeps
 * the draws of the next `keep` st_summary_accumulate calls in HBM (2 x keep x n_all doubles: w and yhat; keep <= 16384;
 * keep = 0 frees them); st_summary_quantile sorts every row's draws in LDS.  Either output may be NULL. */
int st_summary_reserve(st_handle h, int64_t keep);
int st_summary_quantile(st_handle h, double q, double *w_q, double *yhat_q);

int st_set_stream(st_handle h, void *stream);              /* launch on the caller's stream (the one its collectives use) */

/* ---- multi-GPU (st_options.world > 1): one process per GPU shares ONE problem.
 * Ownership: whole subtrees below a cut level belong to one rank, levels above the cut are replicated
 * (st_shard_plan is pure host code: owner[u] = rank or -1 for replicated; no GPU needed).
 * Every exchange is an all-reduce(sum) in which each entry is contributed by exactly one rank and is zero on the
 * others, so the result is bit-identical to the single-GPU arrays for any number of ranks.  The caller owns the
 * collective (RCCL through torch.distributed on the stream given to st_set_stream); the single-call forms
 * (st_factor, st_sample_w, st_loglik_w) are the world == 1 composition of the same steps.
 *   phase A : st_factor_local -> st_mg_pack_comps -> all-reduce(buf) -> st_mg_finish        (code 0/1/2/3, loglik)
 *   phase C : st_loglik_local -> st_mg_pack_comps -> all-reduce(buf) -> st_mg_finish
 *   phase B : st_sample_w_local -> all-reduce(st_mg_top_region) -> st_sample_w_top
 *             -> st_mg_pack_w -> all-reduce(buf) -> st_mg_unpack_w                           (code 0/10/11) */
/* native exchange: rank 0 creates a 128-byte RCCL unique id (returns its size), every rank passes it to st_comm_init;
 * afterwards the single-call forms run the steps above with ncclAllReduce on the library's stream */
int st_comm_unique_id(void *out, int32_t cap);
int st_comm_init(st_handle h, const void *unique_id);
int st_shard_plan(const st_problem *pb, int32_t world, int64_t *owner /* n_blocks */, int32_t *cut_level);
/* the same with options (needed for limited_tree problems, whose single-parent lists only parse with reserved bit 1 set) */
int st_shard_plan_opt(const st_problem *pb, const st_options *opt, int32_t world, int64_t *owner /* n_blocks */, int32_t *cut_level);
int st_shard_info(st_handle h, int32_t *rank, int32_t *world, int32_t *cut_level, int64_t *owned_blocks, int64_t *owned_rows);
int st_factor_local(st_handle h, int slot, const double *theta, int ntheta);
int st_loglik_local(st_handle h, int slot);
int st_mg_pack_comps(st_handle h, int slot, void **dev_ptr, int64_t *len);
int st_mg_finish(st_handle h, double *loglik);
int st_sample_w_local(st_handle h, const double *z, uint64_t seed, uint32_t iter);
int st_mg_top_region(st_handle h, void **dev_ptr, int64_t *len);
int st_sample_w_top(st_handle h);
int st_mg_pack_w(st_handle h, void **dev_ptr, int64_t *len);
int st_mg_unpack_w(st_handle h);
/* all-gather form of the last step of phase B (half the traffic of the all-reduce of n doubles; what the native path uses):
 * every rank's slice of the receive buffer holds its owned rows in device order + its failure word, `count_per_rank`
 * doubles each (the same on all ranks); the replicated top is sampled identically everywhere and does not travel.
 *   st_mg_gather_w_pack -> all-gather(send = recv + rank * count, recv) -> st_mg_gather_w_unpack      (code 0/10/11) */
int st_mg_gather_w_pack(st_handle h, void **send_ptr, void **recv_ptr, int64_t *count_per_rank);
int st_mg_gather_w_unpack(st_handle h);

#ifdef __cplusplus
}
#endif
#endif /* SPAMTREE_HIP_H */
