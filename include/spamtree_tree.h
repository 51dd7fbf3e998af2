/* spamtree_tree.h -- C-ABI of the device parts of the tree builder: the data-parallel steps of
 * `make_tree` (/root/reference/R/make_tree.R) that scale with the number of rows.  Handle-free: plain host pointers in and
 * out, every call moves its operands to the device, computes there and brings the result back.  The host side
 * (spamtree_amd/topology.py) keeps the control flow of make_tree and produces IDENTICAL trees with or without these calls
 * (tests/test_gpu_tree.py); nothing here is used by the MCMC hot path.
 * Return value: 0 ok, negative = usage / HIP error (st_last_error(NULL) has the text). */
#ifndef SPAMTREE_TREE_H
#define SPAMTREE_TREE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ascending sort of n doubles (device radix sort): the order statistics `kthresholds` reads
 * (/root/reference/src/tree_dep.cpp:16-27: threshold i of k = element i*n/k of the sorted sample). */
int st_tb_sort(const double *x, int64_t n, int32_t device, double *sorted_out);

/* one knot per fine cell (make_tree.R:84-92, with this build's deterministic rule instead of R's sample()): for every cell
 * c < ncells, the row r with code[r] == c that minimises (key[r], ix[r]) lexicographically; out_row[c] = r, or -1 for a cell
 * without rows.  key >= 0 (squared distance to the cell centre, scaled by the margin weight when mvbias > 0); ix unique. */
int st_tb_cell_argmin(const int64_t *code, const double *key, const int64_t *ix, int64_t n, int64_t ncells, int32_t device,
                      int64_t *out_row);

/* nearest placed row of the same margin (make_tree.R:236, 256, 345, 367: FNN::get.knnx with k = 1 per margin): for every
 * query the target t minimising (qx-tx)^2 + (qy-ty)^2 among the targets with tmv == qmv -- among ALL targets when no target
 * has the query's margin -- ties to the lowest target index.  Margins are 0-based and < n_margins. */
int st_tb_nearest(const double *tx, const double *ty, const int32_t *tmv, int64_t nt, const double *qx, const double *qy,
                  const int32_t *qmv, int64_t nq, int32_t n_margins, int32_t device, int64_t *out_target);

#ifdef __cplusplus
}
#endif
#endif /* SPAMTREE_TREE_H */
